#!/usr/bin/env python3
"""Joint sweep of the developer switches (pcgrl_tuning) on the final code, one process, GPU box:
    python tools/knob_sweep.py [C2 C3 C5 ...]
For every workload the library's defaults and a list of single- and two-switch variations: GPU microseconds per step in the
bench's first window (20 warm-up + 200 timed steps from a reset) and in the steady state (200 steps after 800), each the best of
`REPEATS` passes of the same environment.  Says whether the defaults are still the best setting now that the kernels around them
have changed since each switch was measured on its own (profiles/*/NOTES.md)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv

REPEATS = 2
VARIANTS = {
    "C2": [{}, {"step_epb": 128}, {"step_prio": 0}, {"step_prio": 15 | (3 << 6)}, {"step_prio": 15 | (4 << 8)}, {"step_prio": 11}, {"step_prio": 7},
           {"full_per_wave": 2}, {"inc_per_wave": 2}, {"full_per_wave": 2, "inc_per_wave": 2}, {"no_touch": 1}, {"touch_tight": 0}, {"touch_tight": 3},
           {"step_pair": 2}, {"step_epb": 128, "full_per_wave": 2}],
    "C3": [{}, {"step_epb": 128}, {"step_pair": 4}, {"step_pair": 8}, {"step_pair": 100}, {"step_prio": 3}, {"step_prio": 15}, {"full_per_wave": 2},
           {"inc_per_wave": 2}, {"step_pair": 4, "full_per_wave": 2}, {"step_epb": 128, "step_pair": 4}],
    "C5": [{}, {"wide_grid": 1024}, {"wide_grid": 4096}, {"wide_pairs": 0}, {"wide_few": 2}, {"wide_few": 8}, {"wide_few": 16}, {"wide_waves": 4},
           {"wide_grid": 1024, "wide_few": 8}, {"wide_spin": 100}, {"pair_min": 64}],
    "C5few": [{}, {"wide_few": 16}, {"wide_few": 24}, {"wide_few": 32}, {"wide_few": 48}, {"wide_few": 64}, {"wide_few": 128}, {"wide_few": 100000},
              {"wide_few": 32, "wide_grid": 1024}, {"wide_few": 32, "wide_grid": 4096}],
    "C5bfew": [{}, {"wide_few": 16}, {"wide_few": 32}, {"wide_few": 64}],
}


def measure(workload, tuning):
    prob, rep, calls, n, _ = bench.WORKLOADS[{"C5few": "C5", "C5bfew": "C5b"}.get(workload, workload)]
    if workload in bench.WRAPPED:
        raise SystemExit("bare workloads only")
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, device="cuda:0", seed=0, tuning=tuning)
    for kw in calls:
        env.adjust_param(**kw)
    W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
    acts = bench.make_actions(torch, rep, 284, n, W, H, nt, env.device, 1234)
    first, steady = [], []
    for r in range(REPEATS):
        env.reset()
        for t in range(20):
            env.step(acts[t])
        first.append(bench.timed_steps(torch, env.device, env.step, acts, 20, 200)[1] * 1e3)
        for t in range(220, 800):
            env.step(acts[t % acts.shape[0]])
        steady.append(bench.timed_steps(torch, env.device, env.step, acts, 800, 200)[1] * 1e3)
    env.close()
    return min(first), min(steady)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--defaults":      # A/B of two builds: the defaults only, three times each workload
        for w in sys.argv[2:]:
            for i in range(3):
                f, s = measure(w, {})
                print("%-6s defaults  %7.2f / %7.2f" % (w, f, s), flush=True)
        return
    todo = sys.argv[1:] or list(VARIANTS)
    for w in todo:
        print("## %s  (us per step, GPU events: first window / steady; best of %d)" % (w, REPEATS), flush=True)
        base = None
        for tun in VARIANTS[w]:
            t0 = time.time()
            try:
                f, s = measure(w, tun)
            except Exception as ex:      # a switch combination the library refuses
                print("%-44s  refused: %s" % (tun or "defaults", ex), flush=True)
                continue
            if base is None:
                base = (f, s)
            print("%-44s  %7.2f / %7.2f   (%+5.2f / %+5.2f)   [%.1f s]" % (str(tun) if tun else "defaults", f, s, f - base[0], s - base[1], time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
