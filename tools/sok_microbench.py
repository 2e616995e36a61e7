#!/usr/bin/env python3
"""Latency of the Sokoban solver kernel on the capped levels of the golden fixtures (GPU only).
    python tools/sok_microbench.py [copies]
Each level whose BFS runs into the pop cap is the long pole of a whole Sokoban step; this times set_maps()
(planes + stats + solver) on a batch holding just those levels."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_pcgrl_amd as gp
import _tuning_env; _tuning_env.apply()      # PCGRL_* environment variables -> the binding's tuning overrides (developer tools only)

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "stats_sokoban_5x5.npz"))
a = d["agents"]
idx = np.nonzero(a[:, 0] >= 5000)[0]
maps = np.repeat(d["maps"][idx], copies, axis=0)
print("levels", len(idx), "iterations", a[idx, :4].tolist(), "batch", len(maps))
env = gp.make_batched("sokoban-narrow-v0", num_envs=len(maps), seed=0)
env.reset()
m = torch.as_tensor(maps, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.set_maps(m)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("set_maps %.3f ms" % ((t1 - t0) * 1e3))
exp = np.repeat(d["stats"][idx], copies, axis=0)
got = env.stats.cpu().numpy()
assert np.array_equal(got, exp), (got[:3], exp[:3])
print("stats ok")
