set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 300 python tools/probe/big_prof.py 50) > gpurun_out/r5b/big_prof7.txt 2>&1
for t in 2 3 4 6; do
(timeout 600 python bench.py --workload B1 --steps 50 --warmup 10 --no-cpu-baseline --no-legs --steady-warmup 0 --tuning big_team=$t) > gpurun_out/r5b/bench_B1f$t.json 2> gpurun_out/r5b/bench_B1f$t.err
done
tail -n 3 gpurun_out/r5b/big_prof7.txt
python - <<'PY'
import json
for f in ("f2","f3","f4","f6"):
    try:
        d=json.loads(open("gpurun_out/r5b/bench_B1%s.json"%f).read().strip().splitlines()[-1])
        print("B1",f, d["value"], d["ms_per_step"])
    except Exception as ex: print(f, ex, open("gpurun_out/r5b/bench_B1%s.err"%f).read()[-800:])
PY
