#!/usr/bin/env python3
"""GPU box: what a collector row could cost -- the 64-byte policy and the wrapped step alone, no bookkeeping copies."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gym_pcgrl_amd.utils import make_vec_envs
N = 65536
v = make_vec_envs("binary-narrow-v0", "narrow", log_dir=None, n_cpu=N, seed=0)
obs = v.reset()
w = v.env
wts = torch.arange(1, 65, device="cuda", dtype=torch.int32)
small = lambda o: (o.reshape(o.shape[0], -1)[:, 360:424].to(torch.int32) * wts).sum(1) % 3
small32 = lambda o: ((o.reshape(o.shape[0], -1)[:, 360:424].to(torch.int32) * wts).sum(1) % 3).to(torch.int32)
def run(fn, T=400):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(T): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / T * 1e6
state = {"o": obs}
def f1():
    state["o"], r, d, _ = w.step(small(state["o"]))
def f2():
    state["o"], r, d, _ = w.step(small32(state["o"]))
acts = torch.zeros(N, dtype=torch.int32, device="cuda")
def f3():
    state["o"], r, d, _ = w.step(acts)
def f4():
    a = small(state["o"])
print("policy + step: %.1f us a row; policy ending in int32: %.1f; step alone: %.1f; policy alone: %.1f" % (run(f1), run(f2), run(f3), run(f4)))
