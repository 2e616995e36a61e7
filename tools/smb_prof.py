#!/usr/bin/env python3
"""Where the cycles of an smb search go (GPU box): builds a copy of the library with -DPCGRL_SMB_PROF
(k_smb sums the cycles of its searches by phase into a debug buffer), steps the S1 workload and prints cycles per pop.
    python tools/smb_prof.py [envs] [steps]"""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
import _tuning_env; _tuning_env.apply()      # PCGRL_* environment variables -> the binding's tuning overrides (developer tools only)
# PCGRL_SMB_PROF_FLAGS: extra -D switches for an A/B build; a library built ahead (`smb_prof.py build`, on the CPU box) travels with the tree
extra = os.environ.get("PCGRL_SMB_PROF_FLAGS", "").split()
so = os.path.join(ROOT, "tools", "probe", "libpcgrl_hip_smbprof%s.so" % "".join(f.replace("-D", "_") for f in extra))
if not os.path.exists(so) or (len(sys.argv) > 1 and sys.argv[1] == "build"):
    subprocess.check_call([os.environ.get("HIPCC", "hipcc")] + _lib.HIPCC_FLAGS + ["-DPCGRL_SMB_PROF"] + extra + _lib.SOURCES + ["-o", so])
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        sys.exit(0)
_lib.SO = so
import torch
import bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
env = BatchedPcgrlEnv(prob="smb", rep="narrow", num_envs=n, seed=0)
env.reset()
W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
acts = bench.make_actions(torch, "narrow", 64, n, W, H, nt, env.device, 1234)
for t in range(3):
    env.step(acts[t])
L = _lib.load()
L.pcgrl_debug_timeline.argtypes = [C.c_void_p]
buf = torch.zeros((64,), dtype=torch.int64, device=env.device)
_lib.check(L.pcgrl_debug_timeline(C.c_void_p(buf.data_ptr())), "timeline")
torch.cuda.synchronize()
t0 = time.time()
for t in range(steps):
    env.step(acts[3 + t])
torch.cuda.synchronize()
dt = time.time() - t0
_lib.check(L.pcgrl_debug_timeline(None), "timeline")
a = buf.cpu().numpy().astype(np.float64)
print("%d envs, %d steps: %.1f ms/step" % (n, steps, dt / steps * 1e3))
if a[2]:
    print("wavefront search (balance 1): %d per step, %d of them outgrew the LDS heap; %.0f pops each, %.0f cycles/pop" % (a[2] / steps, a[4] / steps, a[3] / a[2], a[5] / max(a[3], 1)))
if a[2] and a[7]:
    print("  of which the walk back through the expansion log: %.0f cycles a search (%.1f %% of its time)" % (a[7] / a[2], 100 * a[7] / max(a[5], 1)))
if a[0]:
    print("general search (lanes 0..3): %d per step; %.0f pops each, %.0f cycles/pop" % (a[0] / steps, a[1] / a[0], a[6] / max(a[1], 1)))
