#!/usr/bin/env python3
"""GPU box: RolloutCollector over one batch against DoubleBufferedCollector over its two halves, with a CHEAP policy (one [N,784] x [784,3]
matrix product + argmax: the environment and the collector's bookkeeping are then what is timed, not a network).
    python tools/probe/collector_double.py [env_id=binary-narrow-v0] [rep=narrow] [N=65536] [T=64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gym_pcgrl_amd.rollout import DoubleBufferedCollector, RolloutCollector
from gym_pcgrl_amd.utils import make_vec_envs
env_id = sys.argv[1] if len(sys.argv) > 1 else "binary-narrow-v0"
rep = sys.argv[2] if len(sys.argv) > 2 else "narrow"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
T = int(sys.argv[4]) if len(sys.argv) > 4 else 64
kw = dict(width=11, height=16) if env_id.startswith("zelda") and rep == "wide" else {}
one = make_vec_envs(env_id, rep, log_dir=None, n_cpu=N, seed=0, **kw)
n_act = one.action_space.n
feat = 1
for d in one.observation_space.shape:
    feat *= d
g = torch.Generator(device="cuda").manual_seed(3)
Wt = torch.randn(feat, n_act, generator=g, device="cuda", dtype=torch.bfloat16)
policy = lambda obs: torch.argmax(obs.reshape(obs.shape[0], -1).to(torch.bfloat16) @ Wt, 1)

def run(col, rounds=4):
    col.collect(policy)                    # reset + warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rounds):
        col.collect(policy)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (rounds * T) * 1e6

a = run(RolloutCollector(one, T))
one.close()
halves = [make_vec_envs(env_id, rep, log_dir=None, n_cpu=N // 2, seed=k * (N // 2), **kw) for k in range(2)]
b = run(DoubleBufferedCollector(halves, T))
print("%s %s x %d, cheap policy, %d-step rollouts: one batch %.1f us a row (%.2f G env-steps/s), two sub-batches on two streams %.1f us (%.2f G)"
      % (env_id, rep, N, T, a, N / a / 1e3, b, N / b / 1e3))
