#!/bin/bash
# lazy_draws 0 / 1 on the search problems: lockstep step, k_update's share of it (event pass) and the asynchronous ticks
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for w in $1; do for rep in 1 2; do for t in lazy_draws=0 lazy_draws=1; do
  python bench.py --workload $w --no-legs --no-cpu-baseline --no-rollout --tuning $t 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
ph = d['roofline']['phase_us_per_step_with_event_overhead']; ev = min(ph.values())
a = d.get('async', {})
print('%-4s %-14s step %.1f us  | update %.2f stats %.2f us (event pass) | async %.1f M actions/s, %.1f us a tick' % ('$w', '$t', d['roofline']['gpu_ms_per_step'] * 1e3, ph['update'] - ev, ph['stats'] - ev, a.get('value', 0) / 1e6, a.get('ms_per_tick', 0) * 1e3))"
done; done; done
