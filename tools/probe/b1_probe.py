import sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
for tuning in ({}, {"no_inc": 1}):
    prob, rep, adj, n, _ = bench.WORKLOADS["B1"]
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, seed=0, tuning=tuning)
    for kw in adj: env.adjust_param(**kw)
    env.reset()
    acts = bench.make_actions(torch, rep, 120, n, 100, 100, 2, env.device, 1234)
    for t in range(20): env.step(acts[t])
    torch.cuda.synchronize(); a = time.perf_counter()
    for t in range(20, 120): env.step(acts[t])
    torch.cuda.synchronize(); b = time.perf_counter()
    st = env._bufs["stats"].cpu().numpy()
    print(tuning, "%.3f ms/step" % ((b - a) * 10), "champion flag set in %.1f %% of envs; mean regions %.0f path %.0f" % (100 * (st[:, 2] != 0).mean(), st[:, 0].mean(), st[:, 1].mean()))
    # list sizes of one step: count via the work-list counters is internal; estimate: changed cells touching champion
    env.close()
