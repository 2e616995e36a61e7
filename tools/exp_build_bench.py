#!/usr/bin/env python3
"""Timing experiments on the GPU box: build a copy of the library with extra compiler flags (-DPCGRL_EXP_... switches that make
results WRONG but show what a piece of the step costs) and run bench.py's stepping loop on it.
    python tools/exp_build_bench.py "<flags>" <workload> [bench.py arguments]
    python tools/exp_build_bench.py so:<path of a prebuilt library, relative to the repo> <workload> [...]      (A/B of two builds on one box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
wl = sys.argv[2]
if sys.argv[1].startswith("so:"):          # a library built beforehand (in the container: eight parts side by side are much faster than one unit on the box)
    so = os.path.join(ROOT, sys.argv[1][3:])
else:
    so = "/tmp/libpcgrl_hip_exp.so"
    subprocess.check_call([os.environ.get("HIPCC", "hipcc")] + _lib.HIPCC_FLAGS + sys.argv[1].split() + _lib.SOURCES + ["-o", so], stderr=subprocess.DEVNULL)
_lib.SO = so
import bench
sys.argv = ["bench.py", "--workload", wl, "--no-legs", "--no-cpu-baseline", "--no-rollout"] + sys.argv[3:]
bench.main()
