set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 300 python tools/probe/step_multi_cost.py) > gpurun_out/r5b/step_multi_cost2.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_node.py -x -q) > gpurun_out/r5b/pytest_node.txt 2>&1
(timeout 600 python bench.py --workload S1 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --steady-warmup 45) > gpurun_out/r5b/bench_S1_keep.json 2> gpurun_out/r5b/bench_S1_keep.err
(timeout 600 python bench.py --workload S1 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --steady-warmup 45 --tuning no_inc=1) > gpurun_out/r5b/bench_S1_noinc.json 2> gpurun_out/r5b/bench_S1_noinc.err
tail -n 8 gpurun_out/r5b/step_multi_cost2.txt gpurun_out/r5b/pytest_node.txt; python - <<'PY'
import json
for f in ("keep","noinc"):
    d=json.loads(open("gpurun_out/r5b/bench_S1_%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("steady_state",{}).get("value"))
PY
