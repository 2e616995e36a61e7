#!/bin/bash
# A/B of prebuilt libraries on one box:  tools/ab_so.sh "C2 C3" "gym_pcgrl_amd/lib/libexp_old.so gym_pcgrl_amd/lib/libexp_new.so"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for w in $1; do for rep in 1 2; do for so in $2; do
  python tools/exp_build_bench.py so:$so $w 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-4s %-40s first %.2f us  steady %.2f us' % ('$w', '$so', d['roofline']['gpu_ms_per_step'] * 1e3, d['steady_state']['gpu_ms_per_step'] * 1e3))"
done; done; done
