set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 600 python bench.py --workload B1 --steps 50 --warmup 10 --no-cpu-baseline --no-legs --steady-warmup 0) > gpurun_out/r5b/bench_B1c.json 2> gpurun_out/r5b/bench_B1c.err
(timeout 600 python bench.py --workload B1 --steps 50 --warmup 10 --no-cpu-baseline --no-legs --steady-warmup 0 --tuning big_team=0) > gpurun_out/r5b/bench_B1c0.json 2> gpurun_out/r5b/bench_B1c0.err
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py -x -q) > gpurun_out/r5b/pytest_parity2.txt 2>&1
tail -n 12 gpurun_out/r5b/pytest_parity2.txt
python - <<'PY'
import json
for f in ("c","c0"):
    try:
        d=json.loads(open("gpurun_out/r5b/bench_B1%s.json"%f).read().strip().splitlines()[-1])
        print("B1",f, d["value"], d["ms_per_step"])
    except Exception as ex: print(f, ex, open("gpurun_out/r5b/bench_B1%s.err"%f).read()[-800:])
PY
