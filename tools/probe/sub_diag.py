#!/usr/bin/env python3
"""GPU box: why do two sub-batches sometimes serialise?  For a sequence of environments in one process: do their two streams overlap
by the spin test (before / after stepping), and what does a step cost through pcgrl_step_multi and shard by shard."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench
import _tuning_env; print("tuning", _tuning_env.apply())
from gym_pcgrl_amd import node
from gym_pcgrl_amd.node import MultiGpuPcgrlEnv
dev = torch.device("cuda", 0)
for wl in sys.argv[1:]:
    prob, rep, calls, n, desc = bench.WORKLOADS[wl]
    env = MultiGpuPcgrlEnv(prob=prob, rep=rep, num_envs=n, devices=["cuda:0"] * 2, seed=0, sync_streams=False)
    for kw in calls:
        env.adjust_param(**kw)
    env.reset()
    sh = env.shards[0]
    W, H, nt = sh._prob._width, sh._prob._height, sh.get_num_tiles()
    acts = bench.make_actions(torch, rep, 64, n, W, H, nt, dev, 1234)
    parts = [[acts[t][lo:hi].contiguous() for (lo, hi) in env.ranges] for t in range(64)]
    ov0 = node._pair_overlaps(torch, env.streams[0], env.streams[1])
    def timed(fn, T=300):
        for t in range(20): fn(t)
        torch.cuda.synchronize(); a = time.perf_counter()
        for t in range(T): fn(t)
        torch.cuda.synchronize(); return (time.perf_counter() - a) / T * 1e6
    multi = timed(lambda t: env.step(parts[t % 64]))
    def by_shard(t):
        for g, s in enumerate(env.shards):
            with torch.cuda.stream(env.streams[g]):
                s.step(parts[t % 64][g])
    shard = timed(by_shard)
    ov1 = node._pair_overlaps(torch, env.streams[0], env.streams[1])
    print("%s: streams %x %x  spin test before %s after %s | step_multi %.2f us, shard by shard %.2f us" % (wl, env.streams[0].cuda_stream, env.streams[1].cuda_stream, ov0, ov1, multi, shard), flush=True)
    env.close()
