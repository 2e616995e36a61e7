import sys, time, torch
sys.path.insert(0,'/root/repo')
import gym_pcgrl_amd as gp
def run(n, **kw):
    env = gp.make_batched("binary-narrow-v0", num_envs=n, seed=0)
    if kw: env.adjust_param(**kw)
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    acts = torch.randint(0, 3, (260, n), device="cuda", dtype=torch.int32, generator=g)
    for t in range(40): env.step(acts[t])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(40, 240): env.step(acts[t])
    e1.record(); torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / 200 * 1e3
    env.profile(True)
    for t in range(40, 90): env.step(acts[t])
    ph, st = env.profile_read()
    print(n, kw, "us/step %.1f" % tot, {k: round(1e3 * v / st, 1) for k, v in list(ph.items())[:2]})
run(1024)
run(1024, change_percentage=1.0)
run(65536)
run(65536, change_percentage=1.0)
