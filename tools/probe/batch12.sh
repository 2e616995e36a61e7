set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 600 python bench.py --workload S1 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --steady-warmup 45) > gpurun_out/r5b/bench_S1_prio.json 2> gpurun_out/r5b/bench_S1_prio.err
(timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -k "smb or rollout_vs_oracle or generic_search_path") > gpurun_out/r5b/pytest_huge.txt 2>&1
tail -n 6 gpurun_out/r5b/pytest_huge.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5b/bench_S1_prio.json").read().strip().splitlines()[-1])
print("S1 prio", d["value"], d["ms_per_step"], d.get("steady_state",{}).get("value"))
PY
