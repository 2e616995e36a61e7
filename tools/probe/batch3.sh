set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 900 python -m pytest tests/test_gpu_round5.py -x -q) > gpurun_out/r5b/pytest_round5.txt 2>&1
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "smb") > gpurun_out/r5b/pytest_smb2.txt 2>&1
(timeout 600 python bench.py --workload S1 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --steady-warmup 45) > gpurun_out/r5b/bench_S1_keep2.json 2> gpurun_out/r5b/bench_S1_keep2.err
(timeout 600 python bench.py --workload S1 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --steady-warmup 45 --tuning no_inc=1) > gpurun_out/r5b/bench_S1_noinc2.json 2> gpurun_out/r5b/bench_S1_noinc2.err
(timeout 600 python bench.py --workload S1 --envs 65536 --steps 10 --warmup 3 --no-cpu-baseline --no-legs --steady-warmup 0) > gpurun_out/r5b/bench_S1_keep_64k.json 2> gpurun_out/r5b/bench_S1_keep_64k.err
tail -n 5 gpurun_out/r5b/pytest_round5.txt gpurun_out/r5b/pytest_smb2.txt; python - <<'PY'
import json
for f in ("keep2","noinc2","keep_64k"):
    try:
        d=json.loads(open("gpurun_out/r5b/bench_S1_%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("steady_state",{}).get("value"))
    except Exception as ex: print(f, ex)
PY
