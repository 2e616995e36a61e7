import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
import gym_pcgrl_amd as gp
import _tuning_env; _tuning_env.apply()      # PCGRL_* environment variables -> the binding's tuning overrides (developer tools only)
def run(env_id, n, calls=()):
    env = gp.make_batched(env_id, num_envs=n, seed=0)
    for kw in calls: env.adjust_param(**kw)
    env.reset()
    sp = env.single_action_space
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    tot = np.zeros(7)
    T0, T1 = 100, 140
    for t in range(T1):
        if hasattr(sp, "n"): a = torch.randint(0, sp.n, (n,), device="cuda", dtype=torch.int32, generator=g)
        else: a = torch.stack([torch.randint(0, int(k), (n,), device="cuda", dtype=torch.int32, generator=g) for k in sp.nvec], -1).contiguous()
        env.step(a)
        if t >= T0:
            torch.cuda.synchronize()
            sc = env._bufs["scratch"][:2*7*64*16*4].view(torch.int32).cpu().numpy().reshape(2,7,64,16)[:,:,:,0]
            # the parity used by this step: the one with non-zero counters
            c = sc.sum(2)
            p = int(c.sum(1).argmax())
            tot += c[p]
            if t == T0: shard0 = sc[p,0,0]
    names = ["CHG","RST","SOL","SOL2","RST2","SOL3","INC"]
    print(env_id, {k: round(v/(T1-T0),1) for k,v in zip(names,tot)}, "CHG shard0 (lone)", shard0)
run("binary-narrow-v0", 65536)
run("zelda-wide-v0", 65536, (dict(width=11,height=16),))
run("binary-turtle-v0", 8192, (dict(width=64,height=64),))
