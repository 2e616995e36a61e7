import sys, time, torch
sys.path.insert(0,'/root/repo')
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
import _tuning_env; _tuning_env.apply()      # PCGRL_* environment variables -> the binding's tuning overrides (developer tools only)
for n in (1024, 65536):
    env = BatchedPcgrlEnv("binary","narrow",num_envs=n,seed=0); env.reset()
    acts = torch.randint(0,3,(300,n),device="cuda",dtype=torch.int32)
    for t in range(20): env.step(acts[t])
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for t in range(20,220): env.step(acts[t])
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(n, "enqueue us/step", (t1-t0)/200*1e6, "total us/step", (t2-t0)/200*1e6)
