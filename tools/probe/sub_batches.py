#!/usr/bin/env python3
"""GPU box: a batch stepped as K sub-batches (own handle + stream each, ONE GPU, one pcgrl_step_multi call per step) -- the
double-buffered form of a rollout (the policy works on sub-batch A while B steps): kernels of different streams overlap, so the
tail of one sub-batch's k_step (single wavefronts finishing the longest tasks, three quarters of the SIMDs idle) runs under the
front of the other's.  Per environment nothing changes (shard invariance is tested bitwise: test_node_driver_shard_invariance).
    python tools/probe/sub_batches.py [workload=C2] [K ...]          wall clock per full step of all N environments"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from gym_pcgrl_amd.node import MultiGpuPcgrlEnv
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
Ks = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
prob_, rep_, adj_, n, desc = bench.WORKLOADS[wl]
T, WARM, STEADY = 200, 20, 800
print("workload", wl, desc)
for K in Ks:
    env = MultiGpuPcgrlEnv(prob=prob_, rep=rep_, num_envs=n, devices=["cuda:0"] * K, seed=0, sync_streams=False)
    for kw in adj_:
        env.adjust_param(**kw)
    env.reset()
    sh = env.shards[0]
    Wd, H, nt = sh._prob._width, sh._prob._height, sh.get_num_tiles()
    L = T + WARM + 64
    acts = bench.make_actions(torch, rep_, L, n, Wd, H, nt, sh.device, 1234)
    parts = [[acts[t][lo:hi].contiguous() for (lo, hi) in env.ranges] for t in range(L)]
    torch.cuda.synchronize()

    def run(t0, t1):
        for t in range(t0, t1):
            env.step(parts[t % L])

    def timed(t0):
        torch.cuda.synchronize(); a = time.perf_counter()
        run(t0, t0 + T)
        b = time.perf_counter()
        torch.cuda.synchronize(); c = time.perf_counter()
        return (c - a) / T * 1e6, (b - a) / T * 1e6
    run(0, WARM)
    first, h1 = timed(WARM)
    run(WARM + T, STEADY)
    steady, h2 = timed(STEADY)
    print("K=%2d sub-batches of %6d: first window %.2f us per full step (host issue %.2f), steady %.2f (host %.2f) -> %.3f G / %.3f G env-steps/s"
          % (K, n // K, first, h1, steady, h2, n / first / 1e3, n / steady / 1e3), flush=True)
    env.close(); del env
