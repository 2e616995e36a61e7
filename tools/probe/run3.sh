#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/probe3
O=gpurun_out/probe3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
timeout 400 python bench.py --workload C4 --no-legs --no-cpu-baseline > $O/bench_C4.json 2> $O/bench_C4.err
timeout 400 python bench.py --workload M1 --no-legs --no-cpu-baseline --no-rollout > $O/bench_M1.json 2> $O/bench_M1.err
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/pytest.txt; tail -c 1500 $O/bench_C4.json; echo; tail -c 600 $O/bench_M1.json; echo; cat $O/bench_default.time
