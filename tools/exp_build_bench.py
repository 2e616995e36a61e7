#!/usr/bin/env python3
"""Timing experiments on the GPU box: build a copy of the library with extra compiler flags (-DPCGRL_EXP_... switches that make
results WRONG but show what a piece of the step costs) and run bench.py's stepping loop on it.
    python tools/exp_build_bench.py "<flags>" <workload> [bench.py arguments]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
flags, wl = sys.argv[1].split(), sys.argv[2]
so = "/tmp/libpcgrl_hip_exp.so"
subprocess.check_call([os.environ.get("HIPCC", "hipcc")] + _lib.HIPCC_FLAGS + flags + _lib.SOURCES + ["-o", so], stderr=subprocess.DEVNULL)
_lib.SO = so
import bench
sys.argv = ["bench.py", "--workload", wl, "--no-legs", "--no-cpu-baseline", "--no-rollout"] + sys.argv[3:]
bench.main()
