#!/usr/bin/env python3
"""Randomised parity fuzz on the GPU box: random (problem, representation, map size, parameters, seed) combinations,
every step compared with the CPU oracle (done, reward, info; maps at the end); 40 % of the cases as one pcgrl_rollout
tape, 15 % as an odd-length rollout followed by single steps on the same handle.  A seeded 60-configuration slice of this
runs under `pytest -m gpu` (tests/test_gpu_parity.py::test_fuzz_slice); this script is for long sessions.
    python tools/fuzz_parity.py [cases] [seed] [problem|-] [big|goal]
(big: sizes beyond the tuned kernels; goal: episodes that end by the problem's goal all the time -- parity_harness.draw_config_extra)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_harness as ph

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ONLY = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
MODE = sys.argv[4] if len(sys.argv) > 4 else None
t0 = time.time()
for case in range(cases):
    desc, err = ph.fuzz_case(rs, ONLY, rollout_share=0.4, mixed_share=0.15, mode=MODE)
    if err:
        print(err, "(case %d)" % case)
        sys.exit(1)
    print("ok", case, desc, "%.0fs" % (time.time() - t0), flush=True)
print("fuzz passed:", cases, "cases")
