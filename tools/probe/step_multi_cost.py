#!/usr/bin/env python3
"""Developer probe (round 5, GPU box): where do the host microseconds of MultiGpuPcgrlEnv.step go?  Times (a) env.step(parts) as
bench.py's node_driver leg does, (b) the bare pcgrl_step_multi call on the arrays step() prepared, (c) one bare pcgrl_step.
    python tools/probe/step_multi_cost.py [handles] [envs per handle]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gym_pcgrl_amd.node import MultiGpuPcgrlEnv
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
env = MultiGpuPcgrlEnv(prob="binary", rep="narrow", num_envs=G * n_per, devices=[dev] * G, seed=0, sync_streams=False)
env.reset()
parts = [torch.zeros(n_per, dtype=torch.int32, device=dev) for _ in range(G)]
for _ in range(50):
    env.step(parts)
torch.cuda.synchronize()
def timed(f, calls=2000):
    t0 = time.perf_counter()
    for _ in range(calls):
        f()
    dt = (time.perf_counter() - t0) / calls * 1e6
    torch.cuda.synchronize()
    return dt
M = env._multi
L = M["lib"]
print("env.step(parts)                : %6.1f us" % timed(lambda: env.step(parts)))
print("bare pcgrl_step_multi (ctypes) : %6.1f us" % timed(lambda: L.pcgrl_step_multi(M["handles"], M["actions"], M["streams"], M["n"])))
sh = env.shards[0]
h, a, s = sh._handle, C.c_void_p(parts[0].data_ptr()), C.c_void_p(env.streams[0].cuda_stream)
print("bare pcgrl_step, one handle    : %6.1f us" % timed(lambda: L.pcgrl_step(h, a, s)))
print("empty python call              : %6.1f us" % timed(lambda: None))
bufs = env.action_buffers()
env.step(bufs)
print("env.step(action_buffers())     : %6.1f us" % timed(lambda: env.step(bufs)))
