set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py -x -q) > gpurun_out/r5b/pytest_parity5.txt 2>&1
(timeout 600 python bench.py --workload B1 --steps 50 --warmup 10 --no-cpu-baseline --no-legs --steady-warmup 7000) > gpurun_out/r5b/bench_B1g.json 2> gpurun_out/r5b/bench_B1g.err
(timeout 300 python tools/probe/big_prof.py 50 7000) > gpurun_out/r5b/big_prof8.txt 2>&1
tail -n 5 gpurun_out/r5b/pytest_parity5.txt gpurun_out/r5b/big_prof8.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5b/bench_B1g.json").read().strip().splitlines()[-1])
print("B1", d["value"], d["ms_per_step"], d["steady_state"]["value"], d["steady_state"]["ms_per_step"])
PY
