#!/usr/bin/env python3
"""GPU box: what K independent slices of a search-problem batch (own handle + stream each, one GPU) give today.
    [GPU_MAX_HW_QUEUES=n] python tools/probe/async_slices.py [workload=C4] [K ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from gym_pcgrl_amd.node import MultiGpuPcgrlEnv
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
Ks = [int(a) for a in sys.argv[2:]] or [1, 8, 32]
W = bench.WORKLOADS[wl]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), "workload", wl, W)
prob_, rep_, adj_, n, _ = W
T = 40
for K in Ks:
    env = MultiGpuPcgrlEnv(prob=prob_, rep=rep_, num_envs=n, devices=["cuda:0"] * K, seed=0)
    env.reset()
    sh = env.shards[0]
    Wd, H, nt = sh._prob._width, sh._prob._height, sh.get_num_tiles()
    acts = bench.make_actions(torch, rep_, T + 10, n, Wd, H, nt, sh.device, 1234)
    parts = [[acts[t][lo:hi].contiguous() for (lo, hi) in env.ranges] for t in range(T + 10)]
    torch.cuda.synchronize()
    def run(t0, t1):
        for t in range(t0, t1):
            for g, s in enumerate(env.shards):
                with torch.cuda.stream(env.streams[g]):
                    s.step(parts[t][g])
    run(0, 10)
    torch.cuda.synchronize(); a = time.perf_counter()
    run(10, 10 + T)
    b = time.perf_counter()
    torch.cuda.synchronize(); c = time.perf_counter()
    print("K=%3d slices of %6d: %.3f ms per full step (host issue %.3f ms) -> %.1f M env-steps/s" % (K, n // K, (c - a) / T * 1e3, (b - a) / T * 1e3, n * T / (c - a) / 1e6))
    env.close(); del env
