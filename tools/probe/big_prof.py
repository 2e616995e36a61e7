#!/usr/bin/env python3
"""Developer probe (GPU box): where the cycles of a full recomputation on a map beyond 64 x 64 go (k_big, bigmap.h big_regions_path).
Builds a copy of the library with -DPCGRL_BIG_PROF and steps the B1 workload.
    python tools/probe/big_prof.py [steps] [warm-up steps]"""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
so = os.path.join(ROOT, "tools", "probe", "libpcgrl_hip_bigprof.so")       # (built ahead with `big_prof.py build`: travels with the tree)
if not os.path.exists(so) or (len(sys.argv) > 1 and sys.argv[1] == "build"):
    subprocess.check_call([os.environ.get("HIPCC", "hipcc")] + _lib.HIPCC_FLAGS + ["-DPCGRL_BIG_PROF"] + _lib.SOURCES + ["-o", so])
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        sys.exit(0)
_lib.SO = so
import torch
import bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 20       # (7000: every step then holds a reset -- an episode of B1 is ~6 000 steps)
prob, rep, adj, n, _ = bench.WORKLOADS["B1"]
env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, seed=0)
for kw in adj:
    env.adjust_param(**kw)
env.reset()
acts = bench.make_actions(torch, rep, steps + 20, n, 100, 100, 2, env.device, 1234)
for t in range(warm):
    env.step(acts[t % 20])
L = _lib.load()
L.pcgrl_debug_timeline.argtypes = [C.c_void_p]
buf = torch.zeros((64,), dtype=torch.int64, device=env.device)
_lib.check(L.pcgrl_debug_timeline(C.c_void_p(buf.data_ptr())), "timeline")
torch.cuda.synchronize()
t0 = time.time()
for t in range(steps):
    env.step(acts[20 + t % steps])
torch.cuda.synchronize()
dt = time.time() - t0
_lib.check(L.pcgrl_debug_timeline(None), "timeline")
a = buf.cpu().numpy().astype(np.float64)
calls = max(a[8], 1)
print("%d envs, %d steps: %.3f ms/step; %.1f full recomputations a step" % (n, steps, dt / steps * 1e3, a[8] / steps))
print("per full recomputation (cycles): planes %.0f, tiny components %.0f, first + fill %.0f (%.1f components, %.0f cycles each), size + sweeps %.0f, whole %.0f"
      % (a[4] / calls, a[0] / calls, a[1] / calls, a[5] / calls, a[1] / max(a[5], 1), a[2] / calls, a[3] / calls))
print("regions from the closed forms %.1f; double sweeps started %.1f; the sweeps put off to the end: %.0f cycles (part of `whole`)"
      % (a[9] / calls, a[6] / calls, a[7] / calls))
if a[25]:
    t = a[25]
    print("team form: %.1f a step; cycles each: planes + tiny %.0f, phase A %.0f, phase B %.0f, sweeps %.0f, champion %.0f; resets %.1f a step, %.0f cycles each"
          % (t / steps, a[20] / t, a[21] / t, a[22] / t, a[23] / t, a[24] / t, a[27] / steps, a[26] / max(a[27], 1)))
    print("incremental items of blocks 0..3 (cycles per step, wavefront 0):", [int(a[28 + k] / steps) for k in range(4)])
