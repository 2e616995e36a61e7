#!/usr/bin/env python3
"""Debug helper: the general searches (k_search_big) on the fixture maps, row by row against the fixture."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
for name in sys.argv[1:] or ["stats_sokoban_16x24_p2500", "stats_sokoban_20x20_p700", "stats_sokoban_8x8_p20000", "stats_mdungeon_20x20_p900", "stats_ddave_20x20_p900"]:
    d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    prob = name.split("_")[1]
    maps = d["maps"]; n, h, w = maps.shape
    env = BatchedPcgrlEnv(prob=prob, rep="wide", num_envs=n, seed=1)
    env.adjust_param(width=w, height=h); env.adjust_param(solver_power=int(d["solver_power"]))
    env.reset(); env.set_maps(maps); torch.cuda.synchronize()
    got = env.stats.cpu().numpy().astype(np.int64)
    bad = np.nonzero((got != d["stats"]).any(1))[0]
    print(name, "n", n, "bad", bad.tolist(), "status", env._lib and 0)
    for i in bad[:6]:
        print("   ", i, "got", got[i].tolist(), "exp", d["stats"][i].tolist(), "agents", d["agents"][i].tolist())
    env.close()
