#!/usr/bin/env python3
"""Where the cycles of a Sokoban A* pop go (GPU box): a copy of the library with -DPCGRL_SMB_PROF (sok_search_fast sums the cycles of
its phases, with a full wait at every mark, into a debug buffer), the C4 workload stepped, cycles per pop printed.
    python tools/sok_prof.py"""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
so = "/tmp/libpcgrl_hip_sokprof.so"
subprocess.check_call(["hipcc"] + _lib.HIPCC_FLAGS + ["-DPCGRL_SMB_PROF"] + _lib.SOURCES + ["-o", so], stderr=subprocess.DEVNULL)
_lib.SO = so
import torch, bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
n = 131072
env = BatchedPcgrlEnv(prob="sokoban", rep="narrow", num_envs=n, seed=0)
env.reset()
W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
acts = bench.make_actions(torch, "narrow", 64, n, W, H, nt, env.device, 1234)
for t in range(10):
    env.step(acts[t])
L = _lib.load(); L.pcgrl_debug_timeline.argtypes = [C.c_void_p]
buf = torch.zeros((64,), dtype=torch.int64, device=env.device)
_lib.check(L.pcgrl_debug_timeline(C.c_void_p(buf.data_ptr())), "tl")
torch.cuda.synchronize(); t0 = time.time()
steps = 20
for t in range(steps):
    env.step(acts[10 + t])
torch.cuda.synchronize(); dt = time.time() - t0
_lib.check(L.pcgrl_debug_timeline(None), "tl")
a = buf.cpu().numpy().astype(np.float64)
it = max(a[38], 1)
print("%.2f ms/step; A* searches %d per step, %.0f pops each" % (dt / steps * 1e3, a[39] / steps, it / max(a[39], 1)))
for i, nm in enumerate(["loop head + poll", "pop: top, node, repair, prefetch", "bitboard, win, visited probe", "best + four children", "pushes (pool, cache, heap)"]):
    print("  %-34s %7.0f cycles/pop %5.1f%%" % (nm, a[32 + i] / it, 100 * a[32 + i] / a[32:37].sum()))
print("  total %.0f cycles/pop" % (a[32:37].sum() / it))
sn = max(a[46], 1)
print("heap server, cycles per served pop (%d pops):" % a[46])
for i, nm in enumerate(["wait for (1)", "repair + look-ahead issue", "wait for (2)", "appends", "look-ahead node -> box", "between searches"]):
    print("  %-28s %7.0f" % (nm, a[40 + i] / sn))
