#!/usr/bin/env python3
"""Store-stream microbenchmark of the observation kernel (GPU box): k_obs (pcgrl_observe) against plain device fills / copies
of the same number of bytes -- how close the image writer is to what the memory system takes.

    python tools/obs_bench.py [iters]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from gym_pcgrl_amd import _lib
import _tuning_env; _tuning_env.apply()      # PCGRL_* environment variables -> the binding's tuning overrides (developer tools only)
from gym_pcgrl_amd.envs import BatchedPcgrlEnv

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def timed(fn, n=iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us


CASES = [("binary", "narrow", (), 65536, 28, 28, 1, 0), ("zelda", "wide", (dict(width=11, height=16),), 65536, 16, 11, 0, 1),
         ("zelda", "narrow", (), 65536, 22, 22, 1, 1), ("sokoban", "narrow", (), 131072, 10, 10, 1, 1),
         ("binary", "turtle", (dict(width=64, height=64),), 8192, 28, 28, 1, 0)]
for prob, rep, calls, n, oh, ow, centered, onehot in CASES:
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, seed=0)
    for kw in calls:
        env.adjust_param(**kw)
    env.reset()
    depth = env.get_num_tiles() if onehot else 1
    out = torch.empty((n, oh, ow, depth), dtype=torch.uint8, device="cuda")
    src = torch.empty_like(out)
    pad = env.get_border_tile()
    L = env._lib
    t_obs = timed(lambda: _lib.check(L.pcgrl_observe(env._handle, C.c_void_p(out.data_ptr()), oh, ow, centered, pad, onehot, env._stream()), "observe"))
    t_fill = timed(lambda: out.fill_(1))
    t_copy = timed(lambda: out.copy_(src))
    mb = out.numel() / 1e6
    print("%-8s %-7s %3dx%-3d d%d  %7.1f MB  k_obs %6.1f us (%5.2f TB/s)   fill %6.1f us (%5.2f TB/s)   copy %6.1f us (%5.2f TB/s written)" % (
        prob, rep, oh, ow, depth, mb, t_obs, mb / t_obs, t_fill, mb / t_fill, t_copy, mb / t_copy), flush=True)
    env.close()
