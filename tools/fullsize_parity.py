#!/usr/bin/env python3
"""Full-size parity spot check on the GPU box: the benchmark configurations at their real batch sizes (the paths that only
large batches take: paired certain resets, long work lists, every bucket in use), a sample of environments compared step by
step with the CPU oracle -- stepping, and the same tape as one pcgrl_rollout call.  The same checks run (with at most 60
steps) under `pytest -m gpu` (tests/test_gpu_parity.py::test_full_size_vs_oracle).   python tools/fullsize_parity.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_harness as ph

for name in ph.FULLSIZE_CASES:
    for use_rollout in (False, True):
        n = ph.fullsize_case(name, use_rollout)
        print("ok", "rollout" if use_rollout else "steps", name, ph.FULLSIZE_CASES[name][:2], "sampled envs", n, flush=True)
print("full-size parity passed")
