#!/usr/bin/env python3
"""What k_big (csrc/bigmap.h: maps beyond 64 x 64) costs per kind of map: set_maps() of a batch of identical maps, ms per launch.
    python tools/big_microbench.py [width] [height] [envs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
w = int(sys.argv[1]) if len(sys.argv) > 1 else 100
h = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
env = BatchedPcgrlEnv(prob="binary", rep="wide", num_envs=n)
env.adjust_param(width=w, height=h)
env.reset()
rs = np.random.RandomState(0)
kinds = {}
kinds["all empty"] = np.zeros((h, w), np.uint8)
kinds["all solid"] = np.ones((h, w), np.uint8)
kinds["checkerboard"] = (np.indices((h, w)).sum(0) % 2).astype(np.uint8)
m = np.zeros((h, w), np.uint8); m[1::2, :] = 1; kinds["stripes"] = m
m = np.zeros((h, w), np.uint8)
for y in range(1, h, 2):
    m[y, :] = 1; m[y, (w - 1) if (y // 2) % 2 == 0 else 0] = 0
kinds["serpentine"] = m
for p in (0.3, 0.5, 0.7):
    kinds["random %.1f solid" % p] = (rs.random_sample((h, w)) < p).astype(np.uint8)
for name, m in kinds.items():
    maps = torch.as_tensor(np.repeat(m[None], n, 0), device="cuda")
    env.set_maps(maps); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        env.set_maps(maps)
    torch.cuda.synchronize()
    print("%-18s %8.3f ms per launch of %d maps   stats %s" % (name, (time.perf_counter() - t0) / 3 * 1e3, n, env.stats[0].tolist()))
