#!/usr/bin/env python3
"""GPU box: where the search wavefronts of k_search_async spend a tick (a copy of the library with -DPCGRL_ASYNC_PROF).
    python tools/probe/async_prof.py [workload=C4] [budget=128]"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
# (`async_prof.py build` on the CPU box: the library travels with the tree and the GPU call does not spend minutes compiling)
so = os.path.join(ROOT, "tools", "probe", "libpcgrl_hip_asyncprof.so")
if not os.path.exists(so) or (len(sys.argv) > 1 and sys.argv[1] == "build"):
    subprocess.check_call(["hipcc"] + _lib.HIPCC_FLAGS + ["-DPCGRL_ASYNC_PROF", "-DPCGRL_SMB_PROF"] + _lib.SOURCES + ["-o", so], stderr=subprocess.DEVNULL)
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        sys.exit(0)
_lib.SO = so
import torch, bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 128
prob, rep, adj, n, _ = bench.WORKLOADS[wl]
env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, seed=0)
for kw in adj: env.adjust_param(**kw)
env.reset(); env.enable_async(2048)
W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
T = 100
acts = bench.make_actions(torch, rep, T + 60, n, W, H, nt, env.device, 1234)
for t in range(60): env.tick(acts[t], pop_budget=budget)
L = _lib.load(); L.pcgrl_debug_timeline.argtypes = [C.c_void_p]
buf = torch.zeros((64,), dtype=torch.int64, device=env.device)
_lib.check(L.pcgrl_debug_timeline(C.c_void_p(buf.data_ptr())), "tl")
torch.cuda.synchronize(); t0 = time.time()
for t in range(60, 60 + T): env.tick(acts[t], pop_budget=budget)
torch.cuda.synchronize(); dt = time.time() - t0
_lib.check(L.pcgrl_debug_timeline(None), "tl")
a = buf.cpu().numpy().astype(float)
print("%s budget %d: %.3f ms/tick (instrumented); per tick: %.0f resumed jobs, %.0f fresh jobs" % (wl, budget, dt / T * 1e3, a[9] / T, a[10] / T))
names = ["kernel start, list prefix", "ticket", "map staging + level build", "table clear / restore", "agent (search)", "finalize / save", "exit"]
tot = a[:7].sum()
for i, nm in enumerate(names):
    print("  %-28s %7.1f us per block and tick  (%4.1f %%)" % (nm, a[i] / 100.0 / 256 / T, 100 * a[i] / tot))
