set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 600 python bench.py --workload B1 --steps 50 --warmup 10 --no-cpu-baseline --no-legs --steady-warmup 0) > gpurun_out/r5b/bench_B1b.json 2> gpurun_out/r5b/bench_B1b.err
(timeout 300 python tools/probe/big_prof.py 50) > gpurun_out/r5b/big_prof3.txt 2>&1
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py -x -q) > gpurun_out/r5b/pytest_parity.txt 2>&1
tail -n 6 gpurun_out/r5b/big_prof3.txt gpurun_out/r5b/pytest_parity.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5b/bench_B1b.json").read().strip().splitlines()[-1])
print("B1", d["value"], d["ms_per_step"])
PY
