#!/usr/bin/env python3
"""Where the cycles of an mdungeon A* pop go (GPU box): builds a copy of the library with -DPCGRL_SMB_PROF (md_search_fast sums the
cycles of its phases, with a full wait at every mark, into a debug buffer), steps the M1 workload and prints cycles per pop.
    python tools/md_prof.py"""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT",".")
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
import _tuning_env; _tuning_env.apply()      # PCGRL_* environment variables -> the binding's tuning overrides (developer tools only)
so = "/tmp/libpcgrl_hip_mdprof.so"
subprocess.check_call(["hipcc"] + _lib.HIPCC_FLAGS + ["-DPCGRL_SMB_PROF"] + _lib.SOURCES + ["-o", so])
_lib.SO = so
import torch, bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
n=65536
env = BatchedPcgrlEnv(prob="mdungeon", rep="narrow", num_envs=n, seed=0)
env.reset()
W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
acts = bench.make_actions(torch, "narrow", 64, n, W, H, nt, env.device, 1234)
for t in range(20): env.step(acts[t])
L=_lib.load(); L.pcgrl_debug_timeline.argtypes=[C.c_void_p]
buf=torch.zeros((64,),dtype=torch.int64,device=env.device)
_lib.check(L.pcgrl_debug_timeline(C.c_void_p(buf.data_ptr())),"tl")
torch.cuda.synchronize(); t0=time.time()
steps=20
for t in range(steps): env.step(acts[20+t])
torch.cuda.synchronize(); dt=time.time()-t0
_lib.check(L.pcgrl_debug_timeline(None),"tl")
a=buf.cpu().numpy().astype(np.float64)
it=max(a[21],1)
print("%.2f ms/step; A* searches %d per step, %.0f pops each"%(dt/steps*1e3, a[22]/steps, it/max(a[22],1)))
for i,nm in enumerate(["pop: sift + node","win/visited/best","children","pushes","loop head"]):
    print("  %-18s %7.0f cycles/pop %5.1f%%"%(nm,a[16+i]/it,100*a[16+i]/a[16:21].sum()))
print("  total %.0f cycles/pop"%(a[16:21].sum()/it))
