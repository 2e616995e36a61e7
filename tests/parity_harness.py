"""Step-by-step GPU-vs-oracle comparisons shared by the `-m gpu` tests and the command-line tools (tools/fuzz_parity.py,
tools/fullsize_parity.py).  Test infrastructure: the oracle is the checker here, never the thing measured.

fuzz_case(rs, ...)      one random (problem, representation, map size, parameters, seed) configuration; every step of the
                        HIP path -- stepping, or the whole tape as one pcgrl_rollout call -- against the CPU oracle
fullsize_case(...)      a benchmark configuration at its real batch size (the paths only large batches take: paired certain
                        resets, every bucket in use, 512 environments per persistent block); a sample of environments
                        compared with the oracle at every step
"""
import numpy as np

import oracle_lib as ol

REPS = ["narrow", "wide", "turtle", "narrowcast", "narrowmulti", "turtlecast"]
PROB_MIX = ["binary", "binary", "zelda", "zelda", "sokoban", "mdungeon", "mdungeon", "ddave", "ddave", "smb"]


def draw_config_extra(rs, mode, only=None):
    """The fuzz modes of round 4 (tools/fuzz_parity.py ... big|goal; their own draw stream, so that the seeded slice of draw_config
    that test_fuzz_slice runs stays what it was):
      big   sizes beyond the tuned kernels: maps with a side of 65..130 (csrc/bigmap.h), search levels of 257..1 200 bordered cells
            and now and then a solver_power beyond 16 383 (csrc/search_big.h);
      goal  binary / zelda on the fused step kernel with goals that are met all the time (target_path 1..3, a near enemy allowed):
            episodes end where nobody saw it coming, which is what the draw-cache hand-over between the update wavefronts and the
            in-kernel reset is for (StepLocal::pend)."""
    if mode == "goal":
        prob = only or ("binary", "zelda")[rs.randint(2)]
        rep = ("narrow", "narrow", "turtle", "wide")[rs.randint(4)]
        w, h = int(rs.randint(3, 33)), int(rs.randint(3, 17))
        calls = [dict(width=w, height=h), dict(change_percentage=float(rs.choice([0.2, 0.5, 1.0])), target_path=int(rs.randint(1, 4)))]
        if prob == "zelda":
            calls.append(dict(target_enemy_dist=int(rs.randint(1, 3)), probs={"empty": 0.8, "solid": 0.1, "player": 0.02, "key": 0.02, "door": 0.02,
                                                                              "bat": 0.01, "scorpion": 0.01, "spider": 0.01}))
        return prob, rep, (w, h), calls, int(rs.choice([96, 300, 1000])), 150, int(rs.randint(1, 10 ** 6))
    prob = only or ("binary", "zelda", "sokoban", "mdungeon", "ddave")[rs.randint(5)]
    rep = REPS[rs.randint(6)]
    if prob in ("binary", "zelda"):
        w, h = int(rs.randint(65, 131)), int(rs.randint(1, 131))
        if rs.rand() < 0.5:
            w, h = h, w
        calls = [dict(width=w, height=h), dict(change_percentage=float(rs.choice([0.001, 0.003, 0.01])))]
        E, T = int(rs.choice([5, 16, 40])), 60
    else:
        while True:
            w, h = int(rs.randint(3, 41)), int(rs.randint(3, 41))
            if 256 < (w + 2) * (h + 2) <= 1200:
                break
        power = int(rs.choice([60, 300, 1000])) if rs.rand() < 0.85 else 17000
        calls = [dict(width=w, height=h), dict(change_percentage=float(rs.choice([0.01, 0.03, 0.1])), solver_power=power)]
        if rs.rand() < 0.7:     # open maps with few of the things a level must have exactly one of: the planner runs
            few = float(rs.choice([0.002, 0.004]))
            if prob == "sokoban":
                calls.append(dict(probs={"empty": 0.93, "solid": 0.04, "player": few, "crate": few, "target": few}))
            elif prob == "mdungeon":
                calls.append(dict(probs={"empty": 0.9, "solid": 0.05, "player": few, "exit": few, "potion": 0.01, "treasure": 0.01, "goblin": 0.01, "ogre": 0.01}))
            else:
                calls.append(dict(probs={"empty": 0.8, "solid": 0.17, "player": few, "exit": few, "diamond": 0.004, "key": few, "spike": 0.004}))
        E, T = int(rs.choice([16, 48])), 80
    if rep in ("turtle", "turtlecast") and rs.rand() < 0.5:
        calls.append(dict(warp=True))
    return prob, rep, (w, h), calls, E, T, int(rs.randint(1, 10 ** 6))


def draw_config(rs, only=None):
    """-> (prob, rep, (w, h), calls, E, T, seed0, use_rollout): everything random about one fuzz configuration."""
    prob = only or PROB_MIX[rs.randint(len(PROB_MIX))]
    rep = REPS[rs.randint(6)]
    if prob == "sokoban":
        w, h = int(rs.randint(2, 8)), int(rs.randint(2, 8))
    elif prob in ("mdungeon", "ddave"):
        w, h = int(rs.randint(1, 13)), int(rs.randint(1, 13))
    elif prob == "smb":
        w, h = int(rs.randint(1, 61)), int(rs.randint(3, 17))
    else:
        w, h = int(rs.randint(1, 41)), int(rs.randint(1, 41))
        if rs.rand() < 0.4:
            h = int(rs.randint(1, 17)); w = int(rs.randint(1, 33))
    calls = [dict(width=w, height=h), dict(change_percentage=float(rs.choice([0.05, 0.2, 0.5, 1.0])))]
    if prob == "sokoban":
        calls.append(dict(solver_power=int(rs.choice([50, 300, 1000]))))
    if prob == "mdungeon":
        calls.append(dict(solver_power=int(rs.choice([50, 300, 1000, 5000]))))
        if rs.rand() < 0.7:     # open maps with few players / exits: the planner runs in a good share of the steps
            mon = float(rs.choice([0.0, 0.03, 0.15]))
            calls.append(dict(probs={"empty": 0.75, "solid": float(rs.choice([0.02, 0.1])), "player": 0.03, "exit": 0.03,
                                     "goblin": mon, "ogre": mon}))
        if rs.rand() < 0.5:
            calls.append(dict(target_solution=int(rs.randint(1, 8)), target_col_enemies=float(rs.choice([0.0, 0.3, 0.5])),
                              max_enemies=int(rs.randint(1, 5)), max_potions=int(rs.randint(0, 3)), max_treasures=int(rs.randint(0, 3)),
                              rewards={"dist-win": float(rs.choice([0.1, 0.3, 1.0])), "sol-length": float(rs.choice([1, 0.7]))}))
    if prob == "ddave":
        calls.append(dict(solver_power=int(rs.choice([50, 300, 1000, 5000]))))
        if rs.rand() < 0.7:     # open maps with few players / exits / keys: the planner runs in a good share of the steps
            calls.append(dict(probs={"empty": 0.7, "solid": float(rs.choice([0.05, 0.15])), "player": 0.04, "exit": 0.04, "key": 0.04,
                                     "spike": float(rs.choice([0.0, 0.03]))}))
        if rs.rand() < 0.5:
            calls.append(dict(target_solution=int(rs.randint(1, 8)), target_jumps=int(rs.randint(0, 3)), max_diamonds=int(rs.randint(0, 4)),
                              min_spikes=int(rs.randint(0, 6)), rewards={"dist-win": float(rs.choice([0.1, 0.3, 1.0])), "dist-floor": float(rs.choice([2, 0.5]))}))
    if prob == "smb" and rs.rand() < 0.7:      # mostly blocked levels: episodes do not end at once (smb_prob.py:191-192)
        calls.append(dict(probs={"empty": 0.5, "solid": float(rs.choice([0.3, 0.45])), "tube": float(rs.choice([0.02, 0.1]))}))
    if rep in ("narrow", "narrowcast", "narrowmulti") and rs.rand() < 0.3:
        calls.append(dict(random_tile=False))
    if rep in ("turtle", "turtlecast") and rs.rand() < 0.5:
        calls.append(dict(warp=True))
    if rs.rand() < 0.2:
        calls.append(dict(random_start=False))
    E = int(rs.choice([8, 33, 96])) if w * h > 400 else int(rs.choice([33, 96, 200]))
    T = 120 if w * h > 400 else 200
    seed0 = int(rs.randint(1, 10 ** 6))
    return prob, rep, (w, h), calls, E, T, seed0


MAP_EVERY = 5      # stepping: the cursor and the heat map are compared at every step, the whole map every MAP_EVERY steps and at the end


def _obs_mismatch(obs, exp, t, has_pos, sel=None):
    """Observation of step t (device tensors, optionally the rows `sel`) against the oracle's: cursor and heat map always,
    the map every MAP_EVERY steps.  -> None or the name of what differs."""
    pick = (lambda x: x) if sel is None else (lambda x: x[sel])
    if has_pos and not np.array_equal(pick(obs["pos"]).cpu().numpy().astype(np.int64), np.stack([x["pos"][t] for x in exp])):
        return "pos"
    if not np.array_equal(pick(obs["heatmap"]).cpu().numpy().astype(np.int64), np.stack([x["heatmap"][t] for x in exp]).astype(np.int64)):
        return "heatmap"
    if t % MAP_EVERY == 0 and not np.array_equal(pick(obs["map"]).cpu().numpy(), np.stack([x["maps"][t] for x in exp])):
        return "map"
    return None


def _oracle_rollouts(prob, rep, calls, seeds, acts_by_env):
    out = []
    for seed, a in zip(seeds, acts_by_env):
        o = ol.OracleEnv(prob, rep)
        for kw in calls:
            o.adjust_param(**kw)
        o.seed(int(seed))
        o.reset()
        out.append(o.rollout(a))
    return out


def run_config(prob, rep, calls, E, T, seed0, rs, use_rollout, mixed=False, steps_scale=1.0):
    """One configuration on the GPU against the oracle.  `mixed`: the first part of the tape as one rollout of ODD length,
    the rest as single steps on the same handle (switching between the fused and the work-list pipelines).  Returns None
    or a string describing the first mismatch."""
    import torch
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    T = max(4, int(T * steps_scale))
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=E, seed=seed0)
    try:
        for kw in calls:
            env.adjust_param(**kw)
        env.reset()
        sp = env.single_action_space
        if hasattr(sp, "n"):
            acts = rs.randint(0, sp.n, size=(T, E, 1)).astype(np.int32)
        else:
            acts = np.stack([rs.randint(0, int(k), size=(T, E)) for k in sp.nvec], -1).astype(np.int32)
        exp = _oracle_rollouts(prob, rep, calls, [seed0 + i for i in range(E)], [acts[:, i] for i in range(E)])
        keys = list(env._prob.info_keys) + ["iterations", "changes"]
        t_roll = 0
        obs = env._obs()
        if use_rollout or mixed:
            t_roll = T if not mixed else (T // 3) | 1
            tape = torch.as_tensor(acts[:t_roll] if acts.shape[2] > 1 else acts[:t_roll, :, 0], device="cuda")
            rew_t, done_t, info_t = env.rollout(tape)
            got_info = np.stack([info_t[k].cpu().numpy() for k in keys], 1).astype(np.int64).reshape(t_roll, E, len(keys))
            ok = np.array_equal(done_t.cpu().numpy(), np.stack([x["done"][:t_roll] for x in exp], 1)) and \
                np.array_equal(rew_t.cpu().numpy(), np.stack([x["reward"][:t_roll] for x in exp], 1)) and \
                np.array_equal(got_info, np.stack([x["info"][:t_roll] for x in exp], 1))
            if not ok:
                return "ROLLOUT MISMATCH %s %s %s E %d seed %d" % (prob, rep, calls, E, seed0)
        for t in range(t_roll, T):
            obs, rew, done, info = env.step(acts[t] if acts.shape[2] > 1 else acts[t, :, 0])
            ok = np.array_equal(done.cpu().numpy(), np.array([x["done"][t] for x in exp])) and \
                np.array_equal(rew.cpu().numpy(), np.array([x["reward"][t] for x in exp])) and \
                np.array_equal(np.stack([info[k].cpu().numpy() for k in keys], 1).astype(np.int64), np.stack([x["info"][t] for x in exp]))
            if not ok:
                return "MISMATCH %s %s %s E %d seed %d step %d" % (prob, rep, calls, E, seed0, t)
            bad = _obs_mismatch(obs, exp, t, env._rep.has_pos)
            if bad:
                return "OBS MISMATCH (%s) %s %s %s E %d seed %d step %d" % (bad, prob, rep, calls, E, seed0, t)
        # the state the tape / the steps end in: map, cursor, heat map
        if not np.array_equal(obs["map"].cpu().numpy(), np.stack([x["maps"][-1] for x in exp])):
            return "MAP MISMATCH %s %s %s E %d seed %d" % (prob, rep, calls, E, seed0)
        if env._rep.has_pos and not np.array_equal(obs["pos"].cpu().numpy().astype(np.int64), np.stack([x["pos"][-1] for x in exp])):
            return "POS MISMATCH %s %s %s E %d seed %d" % (prob, rep, calls, E, seed0)
        if not np.array_equal(obs["heatmap"].cpu().numpy().astype(np.int64), np.stack([x["heatmap"][-1] for x in exp]).astype(np.int64)):
            return "HEATMAP MISMATCH %s %s %s E %d seed %d" % (prob, rep, calls, E, seed0)
        env.check_status()
    finally:
        env.close()
    return None


def fuzz_case(rs, only=None, rollout_share=0.4, mixed_share=0.0, steps_scale=1.0, mode=None):
    """Draw one configuration from `rs` and run it.  -> (description, error or None)."""
    prob, rep, wh, calls, E, T, seed0 = draw_config(rs, only) if mode is None else draw_config_extra(rs, mode, only)
    u = rs.rand()
    use_rollout = u < rollout_share
    mixed = (not use_rollout) and u < rollout_share + mixed_share
    err = run_config(prob, rep, calls, E, T, seed0, rs, use_rollout, mixed, steps_scale)
    how = "rollout" if use_rollout else ("rollout+steps" if mixed else "steps")
    return "%s %s %s %s %s E %d" % (how, prob, rep, wh, [list(c.items())[0] for c in calls[1:]], E), err


FULLSIZE_CASES = {
    "C2": ("binary", "narrow", (), 65536, 160),
    "C3": ("zelda", "wide", (dict(width=11, height=16),), 65536, 80),
    "C5": ("binary", "turtle", (dict(width=64, height=64),), 8192, 100),
    "C4": ("sokoban", "narrow", (), 131072, 40),
    "M1": ("mdungeon", "narrow", (), 65536, 40),
    "D1": ("ddave", "narrow", (), 65536, 40),
    "S1": ("smb", "narrow", (), 16384, 30),
}


def fullsize_case(name, use_rollout, max_steps=None):
    """A benchmark configuration at its real batch size; 294 sampled environments (the first 64, the last 32, 200 spread
    over the batch) compared with the oracle at every step.  Raises AssertionError on a mismatch; returns the number of
    sampled environments."""
    import torch
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    prob, rep, calls, N, T = FULLSIZE_CASES[name]
    if max_steps:
        T = min(T, max_steps)
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=N, seed=0)
    try:
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
        exp = _oracle_rollouts(prob, rep, calls, idx, [a_host[:, j] for j in range(len(idx))])
        keys = list(env._prob.info_keys) + ["iterations", "changes"]
        ti = torch.as_tensor(idx, device="cuda")
        obs = env._obs()
        if use_rollout:
            rew_t, done_t, info_t = env.rollout(acts if acts.shape[2] > 1 else acts[:, :, 0])
            got = np.stack([info_t[k].view(T, N)[:, ti].cpu().numpy() for k in keys], 2).astype(np.int64)
            assert np.array_equal(done_t[:, ti].cpu().numpy(), np.stack([x["done"] for x in exp], 1)), ("rollout done", name)
            assert np.array_equal(rew_t[:, ti].cpu().numpy(), np.stack([x["reward"] for x in exp], 1)), ("rollout reward", name)
            assert np.array_equal(got, np.stack([x["info"] for x in exp], 1)), ("rollout info", name)
        else:
            for t in range(T):
                obs, rew, done, info = env.step(acts[t] if acts.shape[2] > 1 else acts[t, :, 0])
                assert np.array_equal(done[ti].cpu().numpy(), np.array([x["done"][t] for x in exp])), ("done", name, t)
                assert np.array_equal(rew[ti].cpu().numpy(), np.array([x["reward"][t] for x in exp])), ("reward", name, t)
                assert np.array_equal(np.stack([info[k][ti].cpu().numpy() for k in keys], 1).astype(np.int64),
                                      np.stack([x["info"][t] for x in exp])), ("info", name, t)
                bad = _obs_mismatch(obs, exp, t, env._rep.has_pos, ti)
                assert bad is None, (bad, name, t)
        assert np.array_equal(obs["map"][ti].cpu().numpy(), np.stack([x["maps"][-1] for x in exp])), ("map", name)
        if env._rep.has_pos:
            assert np.array_equal(obs["pos"][ti].cpu().numpy().astype(np.int64), np.stack([x["pos"][-1] for x in exp])), ("pos", name)
        assert np.array_equal(obs["heatmap"][ti].cpu().numpy().astype(np.int64), np.stack([x["heatmap"][-1] for x in exp]).astype(np.int64)), ("heatmap", name)
        env.check_status()
    finally:
        env.close()
    return len(idx)


def async_case(prob, rep, calls, E, ticks, seed0, rs, pop_budget, nslots, sample=None, flush_every=0, tuning=None):
    """Asynchronous stepping (BatchedPcgrlEnv.tick, pcgrl_step_async) against the oracle: `ticks` ticks of random actions; per
    environment the actions it *took* (the ticks it was not pending at) and the outputs of every step it completed are recorded,
    and afterwards the oracle is stepped through exactly the taken actions -- reward, done, info, cursor, heat map and map of
    every completed step must be equal, bit for bit.  `sample`: indices of the environments compared (default: all).
    flush_every > 0: every that many ticks the pending steps are finished by flush() and a lockstep step() is taken in between
    (the two kinds of stepping on one handle).  Returns a dict of counters; raises AssertionError on a mismatch."""
    import torch
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=E, seed=seed0, tuning=tuning)
    try:
        for kw in calls:
            env.adjust_param(**kw)
        env.reset()
        assert env.enable_async(nslots), "no asynchronous form for this configuration"
        sp = env.single_action_space
        if hasattr(sp, "n"):
            acts = rs.randint(0, sp.n, size=(ticks, E, 1)).astype(np.int32)
        else:
            acts = np.stack([rs.randint(0, int(k), size=(ticks, E)) for k in sp.nvec], -1).astype(np.int32)
        idx = np.arange(E) if sample is None else np.asarray(sample)
        ti = torch.as_tensor(idx, device="cuda")
        keys = list(env._prob.info_keys) + ["iterations", "changes"]
        taken = [[] for _ in idx]
        got = [[] for _ in idx]
        pending = np.zeros(len(idx), bool)
        in_flight = [None] * len(idx)
        n_pending_ticks = 0
        for t in range(ticks):
            lock = flush_every and t % flush_every == flush_every - 1
            a = acts[t] if acts.shape[2] > 1 else acts[t, :, 0]
            if lock:
                # lockstep step on the same handle: finishes what is pending first (those steps complete with their own actions),
                # then every environment takes a[t]
                env.flush()
                assert not env._async["pending"][ti].any()
                rows_f = _async_rows(env, ti, keys)
                for j in np.nonzero(pending)[0]:
                    got[j].append(tuple(x[j] for x in rows_f))
                pending[:] = False
                obs, rew, done, info = env.step(a)
                after = np.zeros(len(idx), bool)
            else:
                obs, rew, done, info, pend = env.tick(a, pop_budget=pop_budget)
                after = pend[ti].cpu().numpy() != 0
            rows = None
            for j in range(len(idx)):
                if not pending[j]:
                    taken[j].append(acts[t, idx[j]])
                if not after[j]:
                    if rows is None:
                        rows = _async_rows(env, ti, keys)
                    got[j].append(tuple(x[j] for x in rows))
            pending = after
            n_pending_ticks += int(after.sum())
        env.flush()
        rows = _async_rows(env, ti, keys)
        for j in np.nonzero(pending)[0]:
            got[j].append(tuple(x[j] for x in rows))
        cnt = env.async_counters()
        exp = _oracle_rollouts(prob, rep, calls, [seed0 + int(i) for i in idx], [np.asarray(tk) for tk in taken])
        for j, x in enumerate(exp):
            assert len(got[j]) == len(taken[j]), ("steps completed vs actions taken", prob, idx[j], len(got[j]), len(taken[j]))
            for k, (rew_k, done_k, info_k, pos_k, heat_k, map_k) in enumerate(got[j]):
                where = (prob, rep, "env", int(idx[j]), "its step", k)
                assert rew_k == x["reward"][k] and bool(done_k) == bool(x["done"][k]), ("reward/done",) + where + (rew_k, x["reward"][k], done_k, x["done"][k])
                assert np.array_equal(info_k, x["info"][k]), ("info",) + where + (info_k, x["info"][k])
                if env._rep.has_pos:
                    assert np.array_equal(pos_k, x["pos"][k]), ("pos",) + where
                assert np.array_equal(heat_k, x["heatmap"][k].astype(np.int64)), ("heatmap",) + where
                assert np.array_equal(map_k, x["maps"][k]), ("map",) + where
        env.check_status()
        cnt["pending_env_ticks"] = n_pending_ticks
        cnt["steps"] = sum(len(tk) for tk in taken)
        return cnt
    finally:
        env.close()


def _async_rows(env, ti, keys):
    b = env._bufs
    info = env._prob.decode_rows(b["info"]) if env._prob.packed_rows else None
    tab = b["info"][ti].cpu().numpy().astype(np.int64)
    if info is not None:
        from gym_pcgrl_amd.envs.batched_env import InfoBatch
        ib = InfoBatch(env._prob.info_keys, b["info"], env._max_iterations, env._max_changes, env._prob.decode_rows)
        inf = np.stack([ib[k][ti].cpu().numpy() for k in keys], 1).astype(np.int64)
    else:
        nk = len(env._prob.info_keys)
        inf = np.concatenate([tab[:, :nk], tab[:, 8:10]], 1)
    return (b["reward"][ti].cpu().numpy(), b["done"][ti].cpu().numpy(), inf, b["pos"][ti].cpu().numpy().astype(np.int64),
            b["heatmap"][ti].cpu().numpy().astype(np.int64), b["map"][ti].cpu().numpy())



def expected_image(m, pos, oh, ow, centered, pad, depth):
    """wrappers.py restated with numpy: Cropped.transform :197-206 (np.pad with the border tile, window at the cursor),
    OneHotEncoding.transform :101-104 (np.eye(dim)[map]), ToImage.transform :53-60.  m [N,H,W], pos [N,2] = (x, y)."""
    n, H, W = m.shape
    out = np.full((n, oh, ow), pad, dtype=np.int64)
    for i in range(n):
        if centered:
            ph_, pw_ = oh // 2 + oh, ow // 2 + ow
            padded = np.pad(m[i].astype(np.int64), ((ph_, ph_), (pw_, pw_)), constant_values=pad)
            x, y = int(pos[i, 0]), int(pos[i, 1])
            out[i] = padded[y + ph_ - oh // 2: y + ph_ - oh // 2 + oh, x + pw_ - ow // 2: x + pw_ - ow // 2 + ow]
        else:
            hh, ww = min(oh, H), min(ow, W)
            out[i, :hh, :ww] = m[i, :hh, :ww]
    if depth == 1:
        return out[..., None].astype(np.uint8)
    return np.eye(depth, dtype=np.uint8)[out]


def fullsize_wrapped_case(name, max_steps=40):
    """The trainer-shaped configurations of bench.py at their real batch size against the ORACLE (not against another call of the
    library): C2w = binary-narrow behind CroppedImagePCGRLWrapper(28) -- the fused step writes the [N, 28, 28, 1] image --, C3w =
    zelda-wide 11 x 16 behind ActionMapImagePCGRLWrapper -- flat ActionMap indices decoded inside the step (pcgrl_step_flat), one-hot
    [N, 16, 11, 8] image.  294 sampled environments: the oracle is stepped with the decoded actions (wrappers.py:139-154: index into
    (H, W, tiles)); reward, done, info and the IMAGE (the wrappers' transform of the oracle's map and cursor) of every step."""
    import torch
    from gym_pcgrl_amd import wrappers
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    N = 65536
    if name == "C2w":
        prob, rep, calls = "binary", "narrow", []
    else:
        prob, rep, calls = "zelda", "wide", [dict(width=11, height=16)]
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=N, seed=0)
    for kw in calls:
        env.adjust_param(**kw)
    w = wrappers.CroppedImagePCGRLWrapper(env, 28) if name == "C2w" else wrappers.ActionMapImagePCGRLWrapper(env)
    try:
        img = w.reset()
        W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
        T = max_steps
        g = torch.Generator(device="cuda"); g.manual_seed(6)
        if name == "C2w":
            acts = torch.randint(0, nt + 1, (T, N), device="cuda", dtype=torch.int32, generator=g)
        else:
            acts = torch.randint(0, W * H * nt, (T, N), device="cuda", dtype=torch.int32, generator=g)
        idx = np.unique(np.concatenate([np.arange(0, 64), np.linspace(0, N - 1, 200).astype(int), np.arange(N - 32, N)]))
        ti = torch.as_tensor(idx, device="cuda")
        a_host = acts[:, ti].cpu().numpy()
        if name == "C2w":
            dec = a_host[:, :, None]
        else:       # ActionMap: flat -> (x, y, tile)
            dec = np.stack([(a_host // nt) % W, a_host // (nt * W), a_host % nt], -1)
        exp = _oracle_rollouts(prob, rep, calls, idx, [dec[:, j] for j in range(len(idx))])
        keys = list(env._prob.info_keys) + ["iterations", "changes"]
        oh, ow, centered, pad = w._window()
        depth = nt if w.one_hot else 1
        for t in range(T):
            img, rew, done, info = w.step(acts[t])
            assert np.array_equal(done[ti].cpu().numpy(), np.array([x["done"][t] for x in exp])), ("done", name, t)
            assert np.array_equal(rew[ti].cpu().numpy(), np.array([x["reward"][t] for x in exp])), ("reward", name, t)
            assert np.array_equal(np.stack([info[k][ti].cpu().numpy() for k in keys], 1).astype(np.int64), np.stack([x["info"][t] for x in exp])), ("info", name, t)
            e_img = expected_image(np.stack([x["maps"][t] for x in exp]), np.stack([x["pos"][t] for x in exp]), oh, ow, centered, pad, depth)
            got = img[ti].cpu().numpy()
            bad = np.nonzero((got != e_img).reshape(len(idx), -1).any(1))[0]
            assert bad.size == 0, ("image", name, t, idx[bad[:5]])
        env.check_status()
    finally:
        env.close()
    return len(idx)
