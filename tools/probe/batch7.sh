set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py -x -q) > gpurun_out/r5b/pytest_parity3.txt 2>&1
(timeout 600 python bench.py --workload B1 --steps 50 --warmup 10 --no-cpu-baseline --no-legs --steady-warmup 0) > gpurun_out/r5b/bench_B1d.json 2> gpurun_out/r5b/bench_B1d.err
(timeout 300 python tools/probe/big_prof.py 50) > gpurun_out/r5b/big_prof4.txt 2>&1
cd /tmp && export TMPDIR=/tmp && (timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 -o b1 -- python /root/repo/bench.py --workload B1 --steps 50 --warmup 10 --no-cpu-baseline --no-legs --steady-warmup 0 --no-rollout) > /root/repo/gpurun_out/r5b/rocprof_b1.txt 2>&1; find /tmp/prof_b1 -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r5b/B1_kernel_stats.csv \;
cd /root/repo
tail -n 8 gpurun_out/r5b/pytest_parity3.txt gpurun_out/r5b/big_prof4.txt; head -8 gpurun_out/r5b/B1_kernel_stats.csv
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5b/bench_B1d.json").read().strip().splitlines()[-1])
print("B1", d["value"], d["ms_per_step"])
PY
