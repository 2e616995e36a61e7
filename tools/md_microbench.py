#!/usr/bin/env python3
"""Time of one capped planner search per agent on the GPU (k_mdungeon, PCGRL_MD_ONLY_AGENT): an unsolvable level --
the exit behind a wall of ogres -- in every environment, statistics recomputed with set_maps.
    python tools/md_microbench.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    import _tuning_env; _tuning_env.apply()      # PCGRL_MD_ONLY_AGENT -> the binding's tuning overrides (developer tools only)
    n = 64
    m = np.zeros((11, 7), np.uint8)
    m[0, 0] = 2; m[10, 6] = 3
    m[7, :] = 7; m[8, :] = 7                       # two rows of ogres: 4 damage per crossing ... plus
    m[9, ::2] = 6                                  # goblins behind them
    m[2, 1] = 4; m[3, 4] = 4; m[1, 5] = 5; m[4, 2] = 5; m[5, 5] = 5
    env = BatchedPcgrlEnv(prob="mdungeon", rep="wide", num_envs=n)
    env.reset()
    maps = np.repeat(m[None], n, 0)
    env.set_maps(maps); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        env.set_maps(maps)
    torch.cuda.synchronize()
    print("agent", os.environ.get("PCGRL_MD_ONLY_AGENT", "all"), "ms per launch %.2f" % ((time.perf_counter() - t0) / 5 * 1e3),
          "stats", env.stats[0].tolist())
else:
    for a in ("", "0", "1", "2", "3"):
        envv = dict(os.environ)
        if a:
            envv["PCGRL_MD_ONLY_AGENT"] = a
        subprocess.call([sys.executable, __file__, "run"], env=envv)
