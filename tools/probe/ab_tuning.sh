#!/bin/bash
# A/B of developer switches of ONE library on one box:  tools/probe/ab_tuning.sh "C2w C3w" "obs_at_end=1 obs_at_end=0" [repeats]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for w in $1; do for rep in $(seq 1 ${3:-2}); do for t in $2; do
  python bench.py --workload $w --no-legs --no-cpu-baseline --no-rollout --tuning $t 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-4s %-24s first %.2f us  steady %.2f us' % ('$w', '$t', d['roofline']['gpu_ms_per_step'] * 1e3, d['steady_state']['gpu_ms_per_step'] * 1e3))"
done; done; done
