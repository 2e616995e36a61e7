#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in C2 C3 C2w C3w C5; do timeout 500 tools/profile_gpu.sh $w trace sq mem; done
timeout 900 tools/profile_gpu.sh C4 trace sq mem
timeout 300 tools/profile_gpu.sh B1 trace
timeout 300 tools/profile_gpu.sh C2 rtrace
for w in C2 C3 C4 C5 C5b B1 K1 C2w C3w; do
  timeout 500 python bench.py --workload $w --no-legs > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
( time timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default.time
python tools/timeline.py C2 > gpurun_out/timeline_C2.txt 2>&1
python tools/big_microbench.py 100 100 1024 > gpurun_out/big_microbench.txt 2>&1
tail -c 400 gpurun_out/bench_default.json
