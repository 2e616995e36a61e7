#!/usr/bin/env python3
"""Benchmark of the batched PCGRL hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C3d|C4|C5|C5b|M1|D1|S1|C2w|C3w] [--envs E]

With --gpus N > 1 and no RANK in the environment the script launches itself under torch.distributed.run with one rank per
GPU (127.0.0.1 rendezvous, RCCL barrier, max-over-ranks time) and prints the one JSON line of rank 0 with n_gpus = N; it
exits non-zero rather than print a line for fewer GPUs than were asked for.  Started by torch.distributed.run itself (the
driver's way) it behaves the same from the second step on.

A "step" is one BatchedPcgrlEnv.step() over the whole batch (E environments per GPU, weak scaling):
Representation.update + Problem.get_stats/get_reward + in-kernel auto-reset, random actions that
are generated on the device *before* the timed region.  The metric is
env-steps/s over all GPUs.  One JSON line is printed by rank 0.

Extra objects on the JSON line:
  roofline      HBM roofline of the step pipeline: achieved = E * B_alg / (GPU time of the step's
                kernels, from HIP events recorded around every launch on the launch stream, over
                the same timed steps), B_alg = 2*H*W + 64 bytes per env-step.
  configs       (default workload, one GPU) short driver-timed legs of the other BASELINE.json configs (C3, C4, C5), of smb
                (S1) and of the trainer-shaped wrapped steps (C2w, C3w): first window + steady state, roofline fraction,
                dominant kernel -- so that every config has a number under the same clock as `value`.
  cpu_baseline  the CPU oracle (oracle/pcgrl_oracle.c, a bit-exact port of the reference) timed on
                this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (prob, rep, adjust_param calls, envs per GPU, description)
    "C2": ("binary", "narrow", (), 65536, "binary-narrow-v0 14x14, 65536 envs/GPU"),
    "C3": ("zelda", "wide", (dict(width=11, height=16),), 65536, "zelda-wide-v0 11x16, 65536 envs/GPU"),
    "C4": ("sokoban", "narrow", (), 131072, "sokoban-narrow-v0 5x5, 131072 envs/GPU"),
    "C5": ("binary", "turtle", (dict(width=64, height=64),), 8192, "binary-turtle-v0 64x64 (adjust_param), 8192 envs/GPU"),
    # SURVEY 8d secondary variants: zelda at its default size; C5 after a second adjust_param call (Q9: only then does
    # max_changes follow the 64x64 map: 819 changes per episode instead of 39, so full recomputations dominate)
    "C3d": ("zelda", "wide", (), 65536, "zelda-wide-v0 11x7 (default size), 65536 envs/GPU"),
    "C5b": ("binary", "turtle", (dict(width=64, height=64), dict(change_percentage=0.2)), 8192,
            "binary-turtle-v0 64x64, adjust_param(width,height) then adjust_param(change_percentage=0.2): max_changes 819, 8192 envs/GPU"),
    # not a BASELINE.json config: the mdungeon problem, reported for completeness
    "M1": ("mdungeon", "narrow", (), 65536, "mdungeon-narrow-v0 7x11, 65536 envs/GPU"),
    "D1": ("ddave", "narrow", (), 65536, "ddave-narrow-v0 11x7, 65536 envs/GPU"),
    "S1": ("smb", "narrow", (), 16384, "smb-narrow-v0 114x14, 16384 envs/GPU"),
    # round 4, not BASELINE configs: the general paths beyond the tuned kernels' sizes (csrc/bigmap.h, csrc/search_big.h)
    "B1": ("binary", "narrow", (dict(width=100, height=100), dict(change_percentage=0.2)), 4096, "binary-narrow-v0 100x100 (adjust_param; general path k_big), 4096 envs/GPU"),
    "K1": ("sokoban", "narrow", (dict(width=20, height=20),), 16384, "sokoban-narrow-v0 20x20 (adjust_param; general searches k_search_big), 16384 envs/GPU"),
    # the trainer-shaped step (utils.make_vec_envs :60-71): the same batches behind the reference's composite
    # wrappers -- the step also leaves the policy's image (crop 28 centred on the cursor / one-hot map) in a device tensor
    "C2w": ("binary", "narrow", (), 65536, "binary-narrow-v0 14x14 behind CroppedImagePCGRLWrapper(crop 28): step + [N,28,28,1] image, 65536 envs/GPU"),
    "C3w": ("zelda", "wide", (dict(width=11, height=16),), 65536, "zelda-wide-v0 11x16 behind ActionMapImagePCGRLWrapper: flat actions, step + one-hot [N,16,11,8] image, 65536 envs/GPU"),
}
WRAPPED = {"C2w": ("cropped", 28), "C3w": ("actionmap", None)}
# legs of the default run besides the headline workload: every BASELINE.json config (and smb, and the wrapped steps) gets a
# driver-timed figure.  (steps, warmup, steady warm-up) are sized so that the whole default run stays well under a minute of GPU time.
LEGS = {"C3": (20, 5, 800), "C4": (10, 3, 40), "C5": (20, 5, 800), "S1": (5, 2, 45), "C2w": (20, 5, 800), "C3w": (20, 5, 800),
        "B1": (50, 10, 7000)}       # (B1: an episode is ~6 000 steps long -- max_changes 2 000, a third of the random actions change a tile)
# asynchronous ticks of the search problems (pcgrl_step_async): (workload, ticks, warm-up ticks, pop budget per search and tick)
ASYNC_LEGS = {"C4_async": ("C4", 300, 60, 64), "M1_async": ("M1", 300, 60, 64), "D1_async": ("D1", 300, 60, 64)}
# the batch as K sub-batches on K streams of the one GPU (sub_batch_leg): (workload, K)
SUB_BATCH_LEGS = {"C2_sub2": ("C2", 2), "C3_sub2": ("C3", 2)}
GPU_CLOCK_HZ = 2.4e9     # MI355X engine clock (MI355X_MICROARCH.md), for the cycles-per-pop figures
DOMINANT = {"B1": "k_big", "K1": "k_search_big", "C2": "k_step", "C3": "k_step", "C3d": "k_step", "C4": "k_sokoban", "C5": "k_stats_wide", "C5b": "k_stats_wide", "M1": "k_mdungeon",
            "D1": "k_ddave", "S1": "k_smb", "C2w": "k_step (writes the image)", "C3w": "k_step (writes the image)"}


def make_stepper(torch, workload, n, device, seed):
    """-> (env, step(actions), reset(), action maker): the bare batched environment, or the same behind the reference's wrapper."""
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    prob, rep, calls, _, _ = WORKLOADS[workload]
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, device=device, seed=seed)
    for kw in calls:
        env.adjust_param(**kw)
    if workload not in WRAPPED:
        return env, env.step, env.reset, None
    from gym_pcgrl_amd import wrappers
    kind, size = WRAPPED[workload]
    w = wrappers.CroppedImagePCGRLWrapper(env, size) if kind == "cropped" else wrappers.ActionMapImagePCGRLWrapper(env)
    return env, w.step, w.reset, w


def timed_steps(torch, device, step, acts, t0, k):
    """k steps from tape row t0 (cyclic) between two HIP events on the launch stream -> (wall s, GPU ms per step)."""
    L = acts.shape[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()          # (the first record of an event creates it: ~60 us of host time that are not part of a step)
    torch.cuda.synchronize(device)
    w0 = time.perf_counter()
    e0.record()
    for t in range(t0, t0 + k):
        step(acts[t % L])
    e1.record()
    torch.cuda.synchronize(device)
    return time.perf_counter() - w0, e0.elapsed_time(e1) / k


def n1_facade_leg(budget_s=2.0):
    """The reference-shaped single environment (`PcgrlEnv`, the N = 1 view: one launch and one device -> host round trip per
    step, numpy observations, info dict): C1's GPU-side counterpart.  The reference's own Python env does 2 278 steps/s on one
    Xeon core; this is what "drops in unchanged" costs when it is used one environment at a time."""
    import numpy as np

    import gym_pcgrl_amd
    env = gym_pcgrl_amd.make("binary-narrow-v0")
    env.seed(0)
    env.reset()
    rs = np.random.RandomState(0)
    acts = rs.randint(0, 3, size=100000)
    for t in range(50):
        _, _, d, _ = env.step(int(acts[t]))
        if d:
            env.reset()
    n, resets = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        _, _, d, _ = env.step(int(acts[50 + n]))
        n += 1
        if d:
            env.reset()
            resets += 1
    dt = time.perf_counter() - t0
    env.close()
    return {"workload": "binary-narrow-v0 14x14, ONE environment through gym_pcgrl_amd.make(): step() -> numpy obs, reward, done, info dict; reset() on done",
            "value": n / dt, "unit": "env-steps/s", "us_per_step": dt / n * 1e6, "steps": n, "episodes": resets,
            "reference_python_env_steps_per_s": 2278, "reference_hardware": "1 vCPU Xeon 2.1 GHz (BASELINE.md section 2; measured in the build container)"}


def collector_leg(torch, device, n=65536, n_steps=8, warm_steps=2):
    """`RolloutCollector` over `make_vec_envs("binary-narrow-v0", "narrow", n_cpu=65536)` with a stand-in policy in the loop: three
    3x3 convolutions (32, 64, 64 filters) + a 512-unit layer + a 3-way head on the [N, 28, 28, 1] image, the shape of the reference's
    `Cnn1` (model.py:9-15; random weights, bf16, no learner) -- env-steps/s of the loop a trainer runs, and the environment's share."""
    import torch.nn.functional as F

    from gym_pcgrl_amd.rollout import RolloutCollector
    from gym_pcgrl_amd.utils import make_vec_envs
    venv = make_vec_envs("binary-narrow-v0", "narrow", n_cpu=n, seed=0, device=str(device))
    g = torch.Generator(device=device).manual_seed(7)
    def mk(*shape):                                # He-initialised random weights
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return (torch.randn(*shape, generator=g, device=device) * (2.0 / fan_in) ** 0.5).to(torch.bfloat16)

    w1, w2, w3 = mk(32, 1, 3, 3), mk(64, 32, 3, 3), mk(64, 64, 3, 3)
    fc, head = mk(512, 64 * 22 * 22), mk(3, 512)
    chunk = 8192                                   # activations of a chunk: 8192 x 64 x 24 x 24 bf16 = 0.6 GB
    t_env = [0.0]
    marks = []

    def policy(obs):
        outs = []
        with torch.no_grad():
            for lo in range(0, obs.shape[0], chunk):
                x = obs[lo:lo + chunk].permute(0, 3, 1, 2).to(torch.bfloat16)
                x = F.relu(F.conv2d(x, w1)); x = F.relu(F.conv2d(x, w2)); x = F.relu(F.conv2d(x, w3))
                x = F.relu(x.flatten(1) @ fc.t())
                outs.append(torch.argmax((x @ head.t()).float() + torch.rand((x.shape[0], 3), generator=g, device=device), 1))
        return torch.cat(outs)

    col = RolloutCollector(venv, n_steps)
    w = venv.env
    step0 = w.step

    def timed_step(actions):                       # GPU time of the environment's share: events around every step on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = step0(actions)
        e1.record()
        marks.append((e0, e1))
        return out

    w.step = timed_step
    for _ in range(max(1, warm_steps // n_steps + 1)):       # first rollout: reset, MIOpen's kernel selection, allocator warm-up
        col.collect(policy)
    torch.cuda.synchronize(device)
    marks.clear()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a0.record()
    col.collect(policy)
    a1.record()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    gpu_ms = a0.elapsed_time(a1)
    env_ms = sum(e0.elapsed_time(e1) for e0, e1 in marks)
    # the same loop with a policy that is next to nothing (64 bytes of the image -> an action: five small kernels): what the collector's
    # own row costs -- the wrapped step plus ~ 15 dependent small kernels (copies of actions / rewards / dones / episode starts) at ~ 3.5 us
    # each; GPU-side: the rollout captured into a HIP graph replays no faster (profiles/r6_round6/NOTES.md)
    w.step = step0
    wts = torch.arange(1, 65, device=device, dtype=torch.int32)
    small = lambda obs: (obs.reshape(obs.shape[0], -1)[:, 360:424].to(torch.int32) * wts).sum(1) % 3
    col.collect(small)
    torch.cuda.synchronize(device)
    s0 = time.perf_counter()
    for _ in range(4):
        col.collect(small)
    torch.cuda.synchronize(device)
    small_us = (time.perf_counter() - s0) / (4 * n_steps) * 1e6
    venv.close()
    return {"small_policy": {"what": "the same collector with a 64-byte policy: the loop's own cost per row", "us_per_row": small_us, "value": n / small_us * 1e6, "unit": "env-steps/s"},
            "workload": "RolloutCollector over make_vec_envs('binary-narrow-v0', 'narrow', n_cpu=%d): stand-in Cnn1-shaped policy (3 conv + fc512, bf16, random weights) "
                        "-> actions -> wrapped step writing the [N,28,28,1] image into the rollout buffer" % n,
            "envs": n, "steps": n_steps, "value": n * n_steps / dt, "unit": "env-steps/s", "ms_per_step": dt / n_steps * 1e3,
            "gpu_ms_per_step": gpu_ms / n_steps, "env_gpu_ms_per_step": env_ms / n_steps, "env_share_of_gpu_time": env_ms / gpu_ms,
            "direct_rows": bool(col.direct)}


def async_leg(torch, device, workload, ticks, warm, budget, nslots=2048, seed=0):
    """Asynchronous stepping of a search problem (BatchedPcgrlEnv.tick = pcgrl_step_async; csrc/kernels_search_async.h): `ticks`
    ticks of random actions on the workload's full batch.  value = actions TAKEN per second = environment steps completed per second
    in the long run (an environment whose search is suspended sits ticks out; the library counts the taken actions).  Every tick
    leaves observation / reward / done / info of every non-pending environment in the live tensors, as a step does."""
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    prob, rep, calls, n, desc = WORKLOADS[workload]
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, device=device, seed=seed)
    for kw in calls:
        env.adjust_param(**kw)
    env.reset()
    if not env.enable_async(nslots):
        env.close()
        return {"error": "no asynchronous form for " + workload}
    W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
    acts = make_actions(torch, rep, ticks + warm, n, W, H, nt, device, 1234)
    for t in range(warm):
        env.tick(acts[t], pop_budget=budget)
    torch.cuda.synchronize(device)
    c0 = env.async_counters()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    torch.cuda.synchronize(device)
    w0 = time.perf_counter()
    e0.record()
    for t in range(warm, warm + ticks):
        env.tick(acts[t], pop_budget=budget)
    e1.record()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - w0
    gms = e0.elapsed_time(e1) / ticks
    c1 = env.async_counters()
    pend = int((env._async["pending"] != 0).sum())
    taken = c1["consumed"] - c0["consumed"]
    pops = c1["pops"] - c0["pops"]
    b_alg = 2 * H * W + 64
    leg = {"workload": desc + " -- asynchronous ticks (pcgrl_step_async), pop budget %d per search and tick, %d slots" % (budget, nslots),
           "envs": n, "ticks": ticks, "warmup_ticks": warm, "pop_budget": budget,
           "value": taken / wall, "unit": "env-steps/s", "ms_per_tick": wall / ticks * 1e3, "gpu_ms_per_tick": gms,
           "actions_taken_of_offered": taken / float(n * ticks), "pending_environments_at_the_end": pend,
           "searches_suspended_per_tick": (c1["suspended"] - c0["suspended"]) / ticks, "slot_overflows": c1["overflow"] - c0["overflow"],
           "search_pops_per_tick": pops / ticks, "search_pops_per_s": pops / wall,
           "algorithmic_bytes_per_env_step": b_alg, "roofline_frac": taken / wall * b_alg / 1e9 / HBM_PEAK_GBPS,
           "dominant_kernel": "k_search_async"}
    env.close()
    return leg


def search_chain(torch, env, step, acts, t0, k=8):
    """Lockstep step of a search problem: the step waits for its longest search.  GPU time of the solver phase per step (HIP events
    of pcgrl_profile around the search kernel) and what that is per pop if the longest search ran into the cap of solver_power
    pops -- at the benchmark's batch sizes nearly every step holds one that does."""
    env.profile(True)
    for t in range(t0, t0 + k):
        step(acts[t % acts.shape[0]])
    ph, n = env.profile_read()
    env.profile(False)
    n = max(n, 1)
    ev = min(ph.values()) / n * 1e3
    us = max(ph.get("solver_or_reset", 0.0) / n * 1e3 - ev, 0.0)
    power = int(getattr(env._prob, "_solver_power", 5000))
    return {"solver_phase_us": us, "solver_power": power, "us_per_pop_if_capped": us / power, "cycles_per_pop_if_capped": us * 1e-6 * GPU_CLOCK_HZ / power,
            "pops_per_s_of_the_longest_search": power / (us * 1e-6) if us > 0 else None}


def node_driver_leg(torch, device, G=8, n_per=256, calls=300):
    """Host cost of stepping a whole node from ONE process (node.MultiGpuPcgrlEnv; SURVEY 8e: at 28 us per C2 step per GPU the host
    must issue the step of all eight GPUs in less than one kernel's time or it is the bottleneck).  G handles with a stream each --
    all on this one GPU: the driver's box has one -- and tiny batches, so that what is timed is the host: microseconds per step()
    call of all G handles, nothing waited for, (a) through pcgrl_step_multi (gather="list": one call of the library per step) with
    per-shard action tensors, (b) shard by shard (G calls of pcgrl_step inside torch stream contexts: the round-4 path), (c) through
    pcgrl_step_multi with the driver's own action buffers written in place (nothing but the library call is left on the host)."""
    from gym_pcgrl_amd.node import MultiGpuPcgrlEnv
    out = {"handles": G, "envs_per_handle": n_per, "calls": calls, "workload": "binary-narrow-v0 14x14"}
    for name, sync_streams in (("host_us_per_call", False), ("host_us_per_call_with_stream_ordering", True)):
        env = MultiGpuPcgrlEnv(prob="binary", rep="narrow", num_envs=G * n_per, devices=[str(device)] * G, seed=0, sync_streams=sync_streams)
        env.reset()
        parts = [torch.zeros(n_per, dtype=torch.int32, device=device) for _ in range(G)]
        for _ in range(20):
            env.step(parts)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(calls):
            env.step(parts)
        out[name] = (time.perf_counter() - t0) / calls * 1e6
        torch.cuda.synchronize(device)
        if not sync_streams:
            # (c) the driver's own action buffers, written in place by the policy (MultiGpuPcgrlEnv.action_buffers): step() then is
            # the one library call and nothing else
            bufs = env.action_buffers()
            for _ in range(20):
                env.step(bufs)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(calls):
                env.step(bufs)
            out["host_us_per_call_action_buffers"] = (time.perf_counter() - t0) / calls * 1e6
            torch.cuda.synchronize(device)
        if sync_streams:
            t0 = time.perf_counter()
            for _ in range(calls):
                env._each(lambda g, sh: sh.step(parts[g]))
            out["host_us_per_call_shard_by_shard"] = (time.perf_counter() - t0) / calls * 1e6
            torch.cuda.synchronize(device)
        env.close()
    return out


def sub_batch_leg(torch, device, workload, K=2, steps=200, warmup=20, steady_warmup=800):
    """The same batch stepped as K sub-batches -- K handles with a stream each on this ONE GPU, one pcgrl_step_multi call per step
    (node.MultiGpuPcgrlEnv with the same device K times): the double-buffered form of a rollout, in which the policy works on one
    sub-batch while the other steps.  Nothing orders sub-batch A's step k + 1 behind sub-batch B's step k, so the tail of one k_step launch
    (single wavefronts finishing the longest tasks, most SIMDs idle) runs under the front of the other's.  Per environment the results
    are those of the one-batch form (contiguous slices, global-index seeds: test_node_driver_shard_invariance, bitwise).  Wall clock
    per step of ALL n environments between two device synchronisations; NOT the headline `value`, which is one batch on one stream."""
    from gym_pcgrl_amd.node import MultiGpuPcgrlEnv
    prob, rep, calls, n, desc = WORKLOADS[workload]
    env = MultiGpuPcgrlEnv(prob=prob, rep=rep, num_envs=n, devices=[str(device)] * K, seed=0, sync_streams=False)
    for kw in calls:
        env.adjust_param(**kw)
    env.reset()
    sh = env.shards[0]
    W, H, nt = sh._prob._width, sh._prob._height, sh.get_num_tiles()
    L = steps + warmup + 64
    acts = make_actions(torch, rep, L, n, W, H, nt, device, 1234)
    parts = [[acts[t][lo:hi].contiguous() for (lo, hi) in env.ranges] for t in range(L)]

    def timed(t0):
        torch.cuda.synchronize(device)
        w0 = time.perf_counter()
        for t in range(t0, t0 + steps):
            env.step(parts[t % L])
        h = time.perf_counter() - w0
        torch.cuda.synchronize(device)
        return (time.perf_counter() - w0) / steps, h / steps
    for t in range(warmup):
        env.step(parts[t])
    # HIP maps streams to hardware queues as it goes (node.side_by_side_streams): when the two sit on one queue the sub-batches run one
    # after the other (2 x 23 us instead of 25 us a C3 step).  The leg says how the streams stood, and asks for new ones up to three times.
    repicks = 0
    while not env.streams_overlap() and repicks < 3:
        env.repick_streams()
        repicks += 1
        for t in range(warmup):
            env.step(parts[t])
    first, host1 = timed(warmup)
    for t in range(warmup + steps, steady_warmup):
        env.step(parts[t % L])
    steady, host2 = timed(max(warmup + steps, steady_warmup))
    overlap_after = env.streams_overlap()
    env.close()
    b_alg = 2 * H * W + 64
    return {"workload": desc, "sub_batches": K, "streams_overlap_after": bool(overlap_after), "stream_repicks": repicks, "envs_per_sub_batch": n // K, "envs": n, "steps": steps, "warmup": warmup,
            "value": n / first, "unit": "env-steps/s", "ms_per_step": first * 1e3, "host_issue_ms_per_step": host1 * 1e3,
            "roofline_frac": n * b_alg / first / 1e9 / HBM_PEAK_GBPS,
            "steady_state": {"value": n / steady, "ms_per_step": steady * 1e3, "host_issue_ms_per_step": host2 * 1e3, "after_steps": max(warmup + steps, steady_warmup),
                             "roofline_frac": n * b_alg / steady / 1e9 / HBM_PEAK_GBPS},
            "what": "K handles + streams on one GPU, one pcgrl_step_multi call per step of all of them; wall clock per step of all envs (no per-kernel figure: "
                    "the sub-batches' launches overlap)"}


def run_leg(torch, device, workload, steps, warmup, steady_warmup, seed=0):
    """One short driver-timed measurement of another workload (rank 0, one GPU): first window after a reset + steady state."""
    prob, rep, calls, n, desc = WORKLOADS[workload]
    env, step, reset, wrapper = make_stepper(torch, workload, n, device, seed)
    reset()
    W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
    acts = make_actions(torch, rep, steps + warmup + 64, n, W, H, nt, device, 1234, flat=(workload == "C3w"))
    for t in range(warmup):
        step(acts[t])
    wall, gms = timed_steps(torch, device, step, acts, warmup, steps)
    b_alg = 2 * H * W + 64
    obs_bytes = 0
    if wrapper is not None:
        o = wrapper._obs
        obs_bytes = int(o.numel() // n)
    leg = {"workload": desc, "envs": n, "steps": steps, "warmup": warmup, "value": n * steps / wall, "unit": "env-steps/s",
           "ms_per_step": wall / steps * 1e3, "gpu_ms_per_step": gms, "dominant_kernel": DOMINANT[workload],
           "algorithmic_bytes_per_env_step": b_alg + obs_bytes,
           "roofline_frac": n * (b_alg + obs_bytes) / (gms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    if steady_warmup > 0:
        t_now = warmup + steps
        for t in range(t_now, steady_warmup):
            step(acts[t % acts.shape[0]])
        t_now = max(t_now, steady_warmup)
        swall, sgms = timed_steps(torch, device, step, acts, t_now, steps)
        leg["steady_state"] = {"after_steps": t_now, "value": n * steps / swall, "ms_per_step": swall / steps * 1e3, "gpu_ms_per_step": sgms,
                               "roofline_frac": n * (b_alg + obs_bytes) / (sgms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    if obs_bytes:
        leg["image_bytes_per_env_step"] = obs_bytes
    if prob in ("sokoban", "mdungeon", "ddave"):
        leg["search"] = search_chain(torch, env, step, acts, warmup + steps)
    env.close()
    add_traffic(leg, workload, n * (b_alg + obs_bytes))
    return leg


def add_traffic(leg, workload, alg_bytes_per_step):
    """The counter-measured HBM line traffic of the workload's step (measured_traffic: the last committed rocprofv3 PMC passes) next to
    the leg's NOMINAL roofline fraction -- which prices the algorithmic bytes 2*H*W + 64 per env-step whether or not the step touched
    them (incremental statistics leave most maps alone: C5's counter traffic is a tenth of its algorithmic bytes) -- and the HBM
    bandwidth the step really drew: traffic / GPU time of a step."""
    traffic, src, head = measured_traffic(workload)
    if traffic is None:
        return
    gms = leg.get("gpu_ms_per_step")
    leg["traffic"] = {"bytes_per_step": traffic, "source": src, "stale": head != tree_hash(), "vs_algorithmic": traffic / float(alg_bytes_per_step),
                      "hbm_gbps_drawn": (traffic / (gms * 1e-3) / 1e9) if gms else None,
                      "hbm_frac_drawn": (traffic / (gms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if gms else None}
    st = leg.get("steady_state")
    if st and st.get("gpu_ms_per_step"):
        st["hbm_gbps_drawn"] = traffic / (st["gpu_ms_per_step"] * 1e-3) / 1e9

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
VALU_ISSUE_PEAK = 1.0e12  # wave64 VALU instructions per second, whole chip: measured (profiles/r3a_round3/valu_calibration.md); 1 024 SIMDs x 2.4 GHz / 2.45


def make_actions(torch, rep, steps, n, W, H, nt, device, seed, flat=False):
    g = torch.Generator(device=device).manual_seed(seed)
    if flat:        # ActionMap (wrappers.py:139-154): one index into (H, W, tiles)
        return torch.randint(0, W * H * nt, (steps, n), generator=g, device=device, dtype=torch.int32)
    if rep == "narrow":
        return torch.randint(0, nt + 1, (steps, n), generator=g, device=device, dtype=torch.int32)
    if rep == "turtle":
        return torch.randint(0, nt + 4, (steps, n), generator=g, device=device, dtype=torch.int32)
    return torch.stack([torch.randint(0, W, (steps, n), generator=g, device=device, dtype=torch.int32),
                        torch.randint(0, H, (steps, n), generator=g, device=device, dtype=torch.int32),
                        torch.randint(0, nt, (steps, n), generator=g, device=device, dtype=torch.int32)], -1).contiguous()


def measured_traffic(workload):
    """-> (bytes per step, file, hash of the kernel sources the passes ran on).  HBM bytes per step from the last committed rocprofv3 PMC passes (profiles/*/<W>_traffic.json: FETCH_SIZE and
    WRITE_SIZE summed over the step's kernels), corrected as calibrated on this GPU with tools/traffic_calib.hip
    (profiles/*/traffic_calibration.md): FETCH_SIZE reports half of the bytes of the 128-byte lines that are read, for
    wide coalesced and for narrow scattered reads alike, so fetched bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact for
    coalesced writes and counts a 32-byte sector per scattered narrow write, so it is taken as it is.
    None when no profile is committed for the workload."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", workload + "_traffic.json")))
    if not files:
        return None, None, None
    d = json.load(open(files[-1]))
    return 2 * d["fetch_bytes_per_step"] + d["write_bytes_per_step"], os.path.relpath(files[-1], ROOT), d.get("csrc_hash")


def tree_hash():
    from gym_pcgrl_amd import _lib
    return _lib.source_hash()


def measured_valu(workload, kernel="k_stats"):
    """Wave-level VALU instructions per dispatch of `kernel` from the last committed rocprofv3 SQ pass
    (profiles/*/<W>_pmc.json), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", workload + "_pmc.json")))
    if not files:
        return None, None, None
    j = json.load(open(files[-1]))
    per = j["per_dispatch"]
    d = per.get(kernel + "_wide") or per.get(kernel)      # tall binary maps run k_stats_wide
    if d is None and kernel == "k_step":
        return None, None, None
    return (d.get("SQ_INSTS_VALU") if d else None), os.path.relpath(files[-1], ROOT), j.get("csrc_hash")


def cpu_baseline(prob, rep, calls, budget_s=12.0):
    """Oracle (kind 'port') on every host core: one environment per thread, random actions."""
    import concurrent.futures as cf

    import numpy as np

    import oracle_lib as ol
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def make(i):
        e = ol.OracleEnv(prob, rep)
        for kw in calls:
            e.adjust_param(**kw)
        e.seed(i)
        e.reset()
        return e

    def actions(e, T, seed):
        rs = np.random.RandomState(seed)
        nt = e.num_tiles
        a = np.zeros((T, 3), np.int32)
        if rep == "narrow":
            a[:, 0] = rs.randint(0, nt + 1, T)
        elif rep == "turtle":
            a[:, 0] = rs.randint(0, nt + 4, T)
        else:
            a[:, 0], a[:, 1], a[:, 2] = rs.randint(0, e.width, T), rs.randint(0, e.height, T), rs.randint(0, nt, T)
        return a

    def run_all(envs, acts):
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(len(envs)) as ex:   # ctypes releases the GIL inside orc_rollout
            list(ex.map(lambda ea: ea[0].rollout(ea[1], want_maps=False, want_heat=False), zip(envs, acts)))
        return time.perf_counter() - t0

    probe = make(0)
    t0 = time.perf_counter()
    probe.rollout(actions(probe, 2000, 1), want_maps=False, want_heat=False)
    rate1 = 2000 / (time.perf_counter() - t0)
    envs = [make(i) for i in range(cores)]
    Tp = 400                                            # short all-core probe to size the bounded sample
    dtp = run_all(envs, [actions(e, Tp, 50 + i) for i, e in enumerate(envs)])
    T = int(max(Tp, min(Tp * budget_s / dtp, 50 * Tp * budget_s)))
    acts = [actions(e, T, 100 + i) for i, e in enumerate(envs)]
    dt = run_all(envs, acts)
    return {"value": cores * T / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d random-action steps of the same workload, one env per host thread" % (cores, T),
            "single_core": rate1}


TALL = "C5"        # BASELINE.json config 5 (north_star's 8-GPU one): binary-turtle 64x64, 8 192 environments per GPU


def tall_maps_leg(torch, dist, device, rank, world, use_dist, same_gpu, steps=20, warmup=5, steady_warmup=800):
    """Under N > 1 every rank also times BASELINE config 5 on its own GPU (8 192 environments per rank, seeded with their global
    indices: rank r owns [r * 8192, (r + 1) * 8192)), first window and steady state, between barriers; the times are the MAX over
    ranks and the values whole-job env-steps/s, like the headline.  No collective on the step path."""
    prob, rep, calls, n, desc = WORKLOADS[TALL]
    env, step, reset, _ = make_stepper(torch, TALL, n, device, rank * n)
    reset()
    W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
    acts = make_actions(torch, rep, steps + warmup + 64, n, W, H, nt, device, 4321 + rank)

    def window(t0):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)
        wall, gms = timed_steps(torch, device, step, acts, t0, steps)
        if use_dist:
            dist.barrier()
            tt = torch.tensor([wall, gms], device="cpu" if same_gpu else device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall, gms = float(tt[0].item()), float(tt[1].item())
        return wall, gms

    for t in range(warmup):
        step(acts[t])
    wall, gms = window(warmup)
    b_alg = 2 * H * W + 64
    leg = {"workload": desc, "envs_per_gpu": n, "n_gpus": world, "steps": steps, "warmup": warmup, "value": float(n) * world * steps / wall,
           "unit": "env-steps/s (whole job, max-over-ranks time)", "ms_per_step": wall / steps * 1e3, "gpu_ms_per_step": gms,
           "dominant_kernel": DOMINANT[TALL], "algorithmic_bytes_per_env_step": b_alg, "roofline_frac": n * b_alg / (gms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    t_now = warmup + steps
    for t in range(t_now, steady_warmup):
        step(acts[t % acts.shape[0]])
    t_now = max(t_now, steady_warmup)
    swall, sgms = window(t_now)
    leg["steady_state"] = {"after_steps": t_now, "value": float(n) * world * steps / swall, "ms_per_step": swall / steps * 1e3, "gpu_ms_per_step": sgms,
                           "roofline_frac": n * b_alg / (sgms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    env.close()
    if rank == 0:
        add_traffic(leg, TALL, n * b_alg)
    return leg


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(a):
    """python bench.py --gpus N (N > 1) without a launcher: run N ranks of this script under torch.distributed.run."""
    import subprocess
    same_gpu = os.environ.get("PCGRL_BENCH_SAME_GPU") == "1"
    if not same_gpu and not a.dry_run:
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            sys.stderr.write("bench.py: --gpus %d asked for but %d visible; refusing to print a line for fewer GPUs\n" % (a.gpus, have))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or len(lines) != 1:
        sys.stderr.write("bench.py: the %d-rank run failed (rc %d, %d result lines)\n%s\n" % (a.gpus, out.returncode, len(lines), out.stdout[-2000:]))
        return out.returncode or 3
    if json.loads(lines[0]).get("n_gpus") != a.gpus:
        sys.stderr.write("bench.py: the result line does not carry n_gpus = %d\n" % a.gpus)
        return 4
    print(lines[0])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the secondary pcgrl_rollout measurement")
    ap.add_argument("--dry-run", action="store_true", help="process plumbing only (launcher, rendezvous, barrier, max-over-ranks reduction over gloo); "
                                                          "no GPU, no environment: the line carries value null and dry_run true")
    ap.add_argument("--no-legs", action="store_true", help="skip the short legs of the other configs (the `configs` object of the default line)")
    ap.add_argument("--legs", default="C3,C4,C4_async,M1_async,C5,S1,B1,C2w,C3w,C2_sub2,C3_sub2,n1_facade,collector,node_driver", help="which legs the default line carries")
    ap.add_argument("--tuning", default="", help="developer switches of the library for A/B runs: field=value[,field=value...] "
                                                 "(include/pcgrl_hip.h pcgrl_tuning, e.g. no_fused=1,step_epb=128)")
    ap.add_argument("--steady-warmup", type=int, default=800, help="steps before the steady_state measurement (0: skip it)")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be at least 1")
    if "RANK" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))

    import torch
    import torch.distributed as dist

    if a.tuning:
        from gym_pcgrl_amd import _lib as _pl
        for kv in a.tuning.split(","):
            k, v = kv.split("=")
            _pl.TUNING_OVERRIDES[k.strip()] = int(v)
        _pl.make_tuning()          # (unknown names fail here)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also with one rank)
    if world != a.gpus:                      # never report a line for another number of GPUs than was asked for
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d; launch with torch.distributed.run --nproc-per-node %d "
                         "(or without a launcher: the script starts its own ranks)\n" % (a.gpus, world, a.gpus))
        sys.exit(2)
    if a.dry_run:
        # the multi-process skeleton of the measurement without a GPU (CPU tests): same launcher, same barrier / max-over-ranks
        # reduction, over gloo; nothing is measured and the line says so
        if use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        if use_dist:
            dist.barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if use_dist:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if rank == 0:
            line = {"metric": "env-steps/sec (whole node)", "value": None, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps,
                    "warmup": a.warmup, "dry_run": True, "max_over_ranks_s": float(tt[0]), "scaling": "weak"}
            if world > 1:      # the shape of the N > 1 line: BASELINE config 5 is timed on every rank as well (tall_maps_leg)
                line["configs"] = {TALL: {"workload": WORKLOADS[TALL][4], "envs_per_gpu": WORKLOADS[TALL][3], "n_gpus": world, "value": None,
                                          "unit": "env-steps/s (whole job, max-over-ranks time)", "dry_run": True}}
            print(json.dumps(line))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    # PCGRL_BENCH_SAME_GPU=1 (tests on a one-GPU box): every rank uses cuda:0 and the ranks meet over gloo
    same_gpu = os.environ.get("PCGRL_BENCH_SAME_GPU") == "1"
    if same_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)   # RCCL; only used for the barrier and the max-over-ranks time

    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    prob, rep, calls, n_default, desc = WORKLOADS[a.workload]
    n = a.envs or n_default
    wrapped = a.workload in WRAPPED
    # environment axis sharded contiguously: rank r owns global envs [r*n, (r+1)*n), seed = global index
    env, step, reset, wrapper = make_stepper(torch, a.workload, n, device, rank * n)
    reset()
    W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
    acts = make_actions(torch, rep, a.steps + a.warmup, n, W, H, nt, device, 1234 + rank, flat=(a.workload == "C3w"))
    for t in range(a.warmup):
        step(acts[t])

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); ev1.record()   # (the first record of an event creates it: ~60 us of host time that are not part of a step)
    barrier()
    t0 = time.perf_counter()
    ev0.record()                # torch's current stream IS the stream the kernels are launched on
    for t in range(a.warmup, a.warmup + a.steps):
        step(acts[t])
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    gpu_ms_per_step = ev0.elapsed_time(ev1) / a.steps
    if use_dist:
        tt = torch.tensor([dt, gpu_ms_per_step], device="cpu" if same_gpu else device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, gpu_ms_per_step = float(tt[0].item()), float(tt[1].item())
    # N > 1: BASELINE config 5 (north_star: 8 192 tall-map environments per GPU, the 8-GPU configuration) on every rank as well
    tall = None
    if world > 1 and a.workload == "C2" and n == n_default and not a.no_legs:
        tall = tall_maps_leg(torch, dist, device, rank, world, use_dist, same_gpu)
    # informational per-kernel breakdown: a second pass with HIP events around every launch
    # (each event record costs a few us on the stream, so it is kept out of the timed region)
    phase_ms, prof_steps = {}, 0
    if rank == 0:
        env.profile(True)
        for t in range(a.warmup, a.warmup + min(a.steps, 50)):
            step(acts[t])
        phase_ms, prof_steps = env.profile_read()
        env.profile(False)

    # steady state: the driver's short window right after a reset sees almost no episode end and every map at 50 % density.
    # The same K steps again once the batch has run for >= --steady-warmup steps (tape rows reused cyclically).
    steady = None
    if rank == 0 and a.steady_warmup > 0:
        L = acts.shape[0]
        t_now = a.warmup + a.steps + prof_steps
        for t in range(t_now, max(t_now, a.steady_warmup)):
            step(acts[t % L])
        t_now = max(t_now, a.steady_warmup)
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(); s1.record()
        torch.cuda.synchronize(device)
        w0 = time.perf_counter()
        s0.record()
        for t in range(t_now, t_now + a.steps):
            step(acts[t % L])
        s1.record()
        torch.cuda.synchronize(device)
        w1 = time.perf_counter()
        sms = s0.elapsed_time(s1) / a.steps
        steady = {"value": float(n) / ((w1 - w0) / a.steps), "unit": "env-steps/s (this rank)", "ms_per_step": (w1 - w0) / a.steps * 1e3,
                  "gpu_ms_per_step": sms, "after_steps": t_now, "steps": a.steps,
                  "what": "the same stepping loop once the batch has run for after_steps steps (episode ends, resets and the whole range of map densities in the mix)"}

    # secondary figure: the same K steps as ONE pcgrl_rollout call on the action tape (a single launch where the fused
    # step kernel applies; per-step reward / done / info still written for every step).  Not the headline `value`.
    rollout = None
    if rank == 0 and not a.no_rollout and not wrapped:
        # a second batch brought to the same state as the first one had when its timed loop started: same seeds, same
        # warm-up actions -- the tape below is then exactly the work of the timed loop above
        env2 = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, device=device, seed=rank * n)
        for kw in calls:
            env2.adjust_param(**kw)
        env2.reset()
        if a.warmup > 0:
            env2.rollout(acts[:a.warmup], want_info=False)
        tape = acts[a.warmup:a.warmup + a.steps].contiguous()
        outs = (torch.empty((a.steps, n), dtype=torch.float64, device=device), torch.empty((a.steps, n), dtype=torch.uint8, device=device),
                torch.empty((a.steps, n, 10), dtype=torch.int32, device=device))     # allocated outside the timed region
        torch.cuda.synchronize(device)
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        env2.rollout(tape, out=outs)
        r1.record()
        torch.cuda.synchronize(device)
        rms = r0.elapsed_time(r1) / a.steps
        env2.close()
        rollout = {"value": float(n) / (rms * 1e-3), "unit": "env-steps/s (this rank)", "ms_per_step": rms, "steps": a.steps,
                   "what": "pcgrl_rollout on the same action tape: K steps in one call, per-step reward/done/info kept"}

    if rank == 0:
        # dominant kernel (k_stats; for sokoban the solver): its share of the step from the event pass -- the idle
        # intervals of that pass measure the cost of an event pair, which is subtracted -- and, from the committed
        # SQ counters, how close it runs to the VALU issue limit (measured: one wave64 VALU instruction per SIMD per 2.45 cycles)
        ph = {k: 1e3 * v / max(prof_steps, 1) for k, v in phase_ms.items()}
        ev_us = min(ph.values()) if ph else 0.0
        # binary maps of <= 16 rows run the whole step as ONE launch (k_step): the "stats" interval is then empty and
        # the kernel's duration is the step time of the timed region itself
        solver = prob in ("sokoban", "mdungeon", "ddave", "smb")     # problems with a search kernel after k_stats
        fused = not solver and ph.get("update", 0.0) > 4 * max(ph.get("stats", 0.0), 1e-3)
        dom_name = "k_sokoban" if prob == "sokoban" else "k_mdungeon" if prob == "mdungeon" else "k_ddave" if prob == "ddave" else "k_smb" if prob == "smb" else "k_step" if fused else (
            "k_stats_wide" if (prob == "binary" and H > 16) else "k_stats")
        dom_us = max((ph.get("solver_or_reset", 0.0) if solver else ph.get("stats", 0.0)) - ev_us, 0.0)
        if fused:
            dom_us = gpu_ms_per_step * 1e3
        elif not solver:          # the event pass perturbs short steps: never more than the step minus the other kernel
            dom_us = min(dom_us, max(gpu_ms_per_step * 1e3 - max(ph.get("update", 0.0) - ev_us, 0.0), 0.0))
        head = tree_hash()
        valu, valu_src, valu_head = measured_valu(a.workload, "k_step" if fused else "k_stats") if n == n_default else (None, None, None)
        dominant = {"name": dom_name, "avg_us": dom_us, "event_pair_overhead_us": ev_us}
        if valu and dom_us > 0 and not solver:
            peak = VALU_ISSUE_PEAK              # measured on this GPU: profiles/r3a_round3/valu_calibration.md (tools/valu_calib.hip)
            dominant.update({"valu_wave_instr_per_launch": valu, "valu_source": valu_src, "valu_head": valu_head, "valu_stale": valu_head != head,
                             "valu_issue_rate": valu / (dom_us * 1e-6), "valu_issue_peak": peak,
                             "valu_issue_frac": valu / (dom_us * 1e-6) / peak})
        total_steps = float(n) * world * a.steps
        value = total_steps / dt
        b_alg = 2 * H * W + 64
        if wrapped:        # the trainer-shaped step also writes the policy's image: algorithmic bytes = the step's + the image
            b_alg += int(wrapper._obs.numel() // n)
        achieved = n * b_alg / (gpu_ms_per_step * 1e-3) / 1e9
        # counter figures come from the committed PMC passes (a --pmc run cannot share a process with the timed loop); they are stamped
        # with the hash of the kernel sources they were measured on, and the line says so when that is not this tree
        traffic, traffic_src, traffic_head = measured_traffic(a.workload) if n == n_default else (None, None, None)
        out = {
            "metric": "env-steps/sec (whole node), binary-narrow 14x14 @ 65536 envs" if (a.workload == "C2" and n == n_default) else "env-steps/sec (whole node)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc, "envs_per_gpu": n, **({"tuning": a.tuning} if a.tuning else {}), "width": W, "height": H, "max_changes": env._max_changes,
                       "max_iterations": env._max_iterations, "actions": "uniform random, device-generated before the timed region",
                       "parallelism": "env-axis shard x%d, no collective on the step path" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_head": traffic_head, "tree_head": head, "traffic_stale": (traffic is not None and traffic_head != head),
                         "algorithmic_bytes_per_launch": n * b_alg,
                         "kernel": ("one step = one launch of k_step (state of 64 / 128 / 256 environments per block staged in LDS: update + stats + resets)" if fused else
                                    "one step = k_update + k_stats (resets inside k_stats); the search problems add k_reset + their search kernel"),
                         "dominant_kernel": dominant,
                         "algorithmic_bytes_per_env_step": b_alg, "gpu_ms_per_step": gpu_ms_per_step,
                         "phase_us_per_step_with_event_overhead": ph},
        }
        if steady is not None:
            steady["roofline_frac"] = n * b_alg / (steady["gpu_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            out["steady_state"] = steady
        if rollout is not None:
            out["rollout"] = rollout
        if solver and prob != "smb" and not wrapped:
            out["search"] = search_chain(torch, env, step, acts, 0)
            if n == n_default and (a.workload + "_async") in ASYNC_LEGS:
                env.close()
                out["async"] = async_leg(torch, device, *ASYNC_LEGS[a.workload + "_async"])
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, rep, calls)
        if world == 1 and not a.no_legs and a.workload == "C2" and n == n_default:
            # every other BASELINE.json config (and smb, and the wrapped steps) under the same clock: short legs, rank 0
            env.close()
            legs = {}
            for name in a.legs.split(","):
                if name in LEGS:
                    legs[name] = run_leg(torch, device, name, *LEGS[name])
                if name in ASYNC_LEGS:
                    legs[name] = async_leg(torch, device, *ASYNC_LEGS[name])
                if name in SUB_BATCH_LEGS:
                    legs[name] = sub_batch_leg(torch, device, *SUB_BATCH_LEGS[name])
            # what a user of the reference's surfaces gets: the single environment (C1's counterpart) and the trainer's loop
            if "n1_facade" in a.legs.split(","):
                legs["n1_facade"] = n1_facade_leg()
            if "node_driver" in a.legs.split(","):
                legs["node_driver"] = node_driver_leg(torch, device)
            if "collector" in a.legs.split(","):
                try:
                    legs["collector"] = collector_leg(torch, device)
                except Exception as ex:        # (a stand-in policy must not take the line down: e.g. no convolution backend on the box)
                    legs["collector"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
            for wname, bare in (("C2w", None), ("C3w", "C3")):       # wrapped step against the bare step of the same batch
                if wname in legs:
                    base = gpu_ms_per_step if bare is None else legs.get(bare, {}).get("gpu_ms_per_step")
                    if base:
                        legs[wname]["gpu_time_vs_bare_step"] = legs[wname]["gpu_ms_per_step"] / base
            out["configs"] = legs
        if tall is not None:
            out.setdefault("configs", {})[TALL] = tall
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
