#!/usr/bin/env python3
"""Where the cycles of a Sokoban A* pop go (GPU box): a copy of the library with -DPCGRL_SMB_PROF (sok_search_fast sums the cycles of
its phases, with a full wait at every mark, into a debug buffer), the C4 workload stepped, cycles per pop of the two wavefronts printed (own work / waiting at the barrier).
    python tools/sok_prof.py [min_pops [sokoban|mdungeon]]"""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
import _tuning_env; _tuning_env.apply()      # PCGRL_* environment variables -> the binding's tuning overrides (developer tools only)
so = "/tmp/libpcgrl_hip_sokprof.so"
MIN_POPS = int(sys.argv[1]) if len(sys.argv) > 1 else 0       # only searches of at least that many pops
PROB = sys.argv[2] if len(sys.argv) > 2 else "sokoban"         # or mdungeon: own work / waiting of the two wavefronts only
if os.environ.get("PCGRL_PROF_SO"):      # built ahead on the CPU box: python tools/exp_build_local.py "-DPCGRL_SMB_PROF -DPCGRL_SKD_MIN_POPS=4000" sokprof
    so = os.path.join(ROOT, os.environ["PCGRL_PROF_SO"])
else:
    subprocess.check_call(["hipcc"] + _lib.HIPCC_FLAGS + ["-DPCGRL_SMB_PROF", "-DPCGRL_SKD_MIN_POPS=%d" % MIN_POPS] + _lib.SOURCES + ["-o", so], stderr=subprocess.DEVNULL)
_lib.SO = so
import torch, bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
n = 131072 if PROB == "sokoban" else 65536
env = BatchedPcgrlEnv(prob=PROB, rep="narrow", num_envs=n, seed=0)
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
if PROB != "sokoban":
    print("search wavefront: %.0f cycles of own work + %.0f waiting at the barrier, per pop" % (a[32] / it, a[33] / it))
else:
  print("search wavefront, cycles per pop:")
  for i, nm in [(2, "loop head, node"), (3, "bitboard, win test, visited probe"), (4, "best + four children"), (0, "pool / cache / box writes"), (1, "waiting at the barrier"), (5, "next top, look-ahead issue")]:
      print("  %-36s %6.0f" % (nm, a[32 + i] / it))
sn = max(a[46], 1)
print("heap server:      remove + repair %.0f, appends %.0f, waiting at the barrier %.0f cycles per pop" % (a[40] / sn, a[42] / sn, a[41] / sn))
