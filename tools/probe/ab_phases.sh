#!/bin/bash
# A/B of prebuilt libraries on one box with the per-kernel phases:  tools/probe/ab_phases.sh "C5 C5b" "lib1.so lib2.so" [repeats]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for w in $1; do for rep in $(seq 1 ${3:-2}); do for so in $2; do
  python tools/exp_build_bench.py so:$so $w 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
ph = d['roofline']['phase_us_per_step_with_event_overhead']
ev = min(ph.values())
print('%-4s %-44s first %.2f us  steady %.2f us | update %.2f stats %.2f (event pass, pair overhead %.2f taken off)' % ('$w', '$so', d['roofline']['gpu_ms_per_step'] * 1e3, d['steady_state']['gpu_ms_per_step'] * 1e3, ph['update'] - ev, ph['stats'] - ev, ev))"
done; done; done
