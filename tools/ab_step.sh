#!/bin/bash
# A/B of k_step's developer switches on the headline workloads (first window / steady state GPU us per step).
#   tools/ab_step.sh "C2 C3" "full_per_wave=4 full_per_wave=2 full_per_wave=1"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
WLS=${1:-"C2 C3"}
VARS=${2:-"full_per_wave=4 full_per_wave=2 full_per_wave=1"}
for w in $WLS; do for v in $VARS; do
  python bench.py --workload $w --no-legs --no-cpu-baseline --no-rollout --tuning "$v" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-4s %-34s first %.2f us  steady %.2f us' % ('$w', '$v', d['roofline']['gpu_ms_per_step'] * 1e3, d['steady_state']['gpu_ms_per_step'] * 1e3))"
done; done
