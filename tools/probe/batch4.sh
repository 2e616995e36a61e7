set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 300 python tools/probe/big_prof.py 50) > gpurun_out/r5b/big_prof2.txt 2>&1
(timeout 300 python tools/smb_prof.py 16384 10) > gpurun_out/r5b/smb_prof_cur.txt 2>&1
(PCGRL_SMB_PROF_FLAGS="-DPCGRL_SMB_NO_SEEN" timeout 300 python tools/smb_prof.py 16384 10) > gpurun_out/r5b/smb_prof_noseen.txt 2>&1
(timeout 600 python bench.py --workload B1 --steps 50 --warmup 10 --no-cpu-baseline --no-legs --steady-warmup 0) > gpurun_out/r5b/bench_B1.json 2> gpurun_out/r5b/bench_B1.err
(timeout 2400 python -m pytest tests -x -q -m gpu) > gpurun_out/r5b/pytest_full.txt 2>&1
tail -n 6 gpurun_out/r5b/big_prof2.txt gpurun_out/r5b/smb_prof_cur.txt gpurun_out/r5b/smb_prof_noseen.txt gpurun_out/r5b/pytest_full.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5b/bench_B1.json").read().strip().splitlines()[-1])
print("B1", d["value"], d["ms_per_step"])
PY
