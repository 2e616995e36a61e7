#!/usr/bin/env python3
"""GPU box: throughput of asynchronous ticks (BatchedPcgrlEnv.tick) against the lockstep step.
    python tools/probe/async_bench.py [workload=C4] [budget ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
budgets = [int(a) for a in sys.argv[2:]] or [128, 256, 512]
prob, rep, adj, n, _ = bench.WORKLOADS[wl]
T = 300
for budget in budgets:
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, seed=0)
    for kw in adj: env.adjust_param(**kw)
    env.reset()
    assert env.enable_async(2048)
    W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
    acts = bench.make_actions(torch, rep, T + 60, n, W, H, nt, env.device, 1234)
    for t in range(60): env.tick(acts[t], pop_budget=budget)
    torch.cuda.synchronize(); c0 = env.async_counters(); a = time.perf_counter()
    for t in range(60, 60 + T): env.tick(acts[t], pop_budget=budget)
    torch.cuda.synchronize(); b = time.perf_counter(); c1 = env.async_counters()
    steps = c1["consumed"] - c0["consumed"]
    pend = int((env._async["pending"] != 0).sum())
    print("%s budget %4d: %.3f ms/tick, %.1f M env-steps/s (%.2f %% of the offered actions taken; pending now %d; suspended %d late %d overflow %d pops/tick %.0f)" % (
        wl, budget, (b - a) / T * 1e3, steps / (b - a) / 1e6, 100.0 * steps / (n * T), pend, c1["suspended"] - c0["suspended"], c1["late"] - c0["late"],
        c1["overflow"] - c0["overflow"], (c1["pops"] - c0["pops"]) / T), flush=True)
    env.close(); del env
