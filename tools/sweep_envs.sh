for n in 1024 8192 32768 65536 262144; do
timeout 300 python bench.py --workload C2 --envs $n --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print($n, round(d['value']/1e6,1), round(d['ms_per_step']*1e3,1), {k:round(v,1) for k,v in list(r['phase_us_per_step_with_event_overhead'].items())[:2]})"
done
