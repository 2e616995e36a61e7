#!/usr/bin/env python3
"""Full-size parity spot check on the GPU box: the benchmark configurations at their real batch sizes (the paths that only
large batches take: paired certain resets, long work lists, every bucket in use), a sample of environments compared step by
step with the CPU oracle.   python tools/fullsize_parity.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from gym_pcgrl_amd.envs import BatchedPcgrlEnv

CASES = [("binary", "narrow", (), 65536, 160), ("zelda", "wide", (dict(width=11, height=16),), 65536, 80),
         ("binary", "turtle", (dict(width=64, height=64),), 8192, 100), ("sokoban", "narrow", (), 131072, 40),
         ("mdungeon", "narrow", (), 65536, 40), ("ddave", "narrow", (), 65536, 40)]
# every case twice: stepping, and the same tape as one pcgrl_rollout call (the fused / persistent kernels at full size)
for prob, rep, calls, N, T in [c for c in CASES for _ in (0, 1)]:
    use_rollout = getattr(sys.modules[__name__], "_flip", False)
    _flip = not use_rollout
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=N, seed=0)
    for kw in calls:
        env.adjust_param(**kw)
    env.reset()
    sp = env.single_action_space
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    if hasattr(sp, "n"):
        acts = torch.randint(0, sp.n, (T, N, 1), device="cuda", dtype=torch.int32, generator=g)
    else:
        acts = torch.stack([torch.randint(0, int(k), (T, N), device="cuda", dtype=torch.int32, generator=g) for k in sp.nvec], -1).contiguous()
    idx = np.unique(np.concatenate([np.arange(0, 64), np.linspace(0, N - 1, 200).astype(int), np.arange(N - 32, N)]))
    a_host = acts[:, torch.as_tensor(idx, device="cuda")].cpu().numpy()
    exp = []
    for j, i in enumerate(idx):
        o = ol.OracleEnv(prob, rep)
        for kw in calls:
            o.adjust_param(**kw)
        o.seed(int(i)); o.reset()
        exp.append(o.rollout(a_host[:, j], want_heat=False))
    keys = list(env._prob.info_keys) + ["iterations", "changes"]
    ti = torch.as_tensor(idx, device="cuda")
    if use_rollout:
        rew_t, done_t, info_t = env.rollout(acts if acts.shape[2] > 1 else acts[:, :, 0])
        got = np.stack([info_t[k].view(T, N)[:, ti].cpu().numpy() for k in keys], 2).astype(np.int64)
        ok = np.array_equal(done_t[:, ti].cpu().numpy(), np.stack([x["done"] for x in exp], 1)) and \
            np.array_equal(rew_t[:, ti].cpu().numpy(), np.stack([x["reward"] for x in exp], 1)) and \
            np.array_equal(got, np.stack([x["info"] for x in exp], 1))
        assert ok, ("ROLLOUT MISMATCH", prob, rep)
        obs = env._obs()
    for t in range(0 if not use_rollout else T, T):
        obs, rew, done, info = env.step(acts[t] if acts.shape[2] > 1 else acts[t, :, 0])
        ok = np.array_equal(done[ti].cpu().numpy(), np.array([x["done"][t] for x in exp])) and \
            np.array_equal(rew[ti].cpu().numpy(), np.array([x["reward"][t] for x in exp])) and \
            np.array_equal(np.stack([info[k][ti].cpu().numpy() for k in keys], 1).astype(np.int64), np.stack([x["info"][t] for x in exp]))
        assert ok, ("MISMATCH", prob, rep, "step", t)
    assert np.array_equal(obs["map"][ti].cpu().numpy(), np.stack([x["maps"][-1] for x in exp]))
    env.check_status()
    print("ok", "rollout" if use_rollout else "steps", prob, rep, "N", N, "steps", T, "sampled envs", len(idx), flush=True)
    env.close()
print("full-size parity passed")
