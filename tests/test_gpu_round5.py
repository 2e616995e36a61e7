"""Round-5 GPU tests that belong to no larger suite: the one-call node step (pcgrl_step_multi), what adjust_param reports and
when, the tape of a search problem on a one-row map wider than 64 cells."""
import numpy as np
import pytest

import oracle_lib as ol
import parity_harness as ph

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prob,rep,calls,n", [("binary", "narrow", (), 1500), ("zelda", "wide", (dict(width=11, height=16),), 900), ("sokoban", "turtle", (), 700)])
@pytest.mark.parametrize("G", [2, 8])
def test_one_call_node_step_equals_the_single_batch(prob, rep, calls, n, G):
    """node.MultiGpuPcgrlEnv.step with gather="list" goes through ONE pcgrl_step_multi call for all G handles: every step's outputs
    must be those of one BatchedPcgrlEnv over the same environments (the handles share this box's one GPU)."""
    import torch
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    from gym_pcgrl_amd.node import MultiGpuPcgrlEnv
    one = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, seed=31)
    node = MultiGpuPcgrlEnv(prob=prob, rep=rep, num_envs=n, devices=["cuda:0"] * G, seed=31)
    for kw in calls:
        one.adjust_param(**kw); node.adjust_param(**kw)
    one.reset(); node.reset()
    sp = one.single_action_space
    rs = np.random.RandomState(3)
    for t in range(40):
        a = rs.randint(0, sp.n, size=n).astype(np.int32) if hasattr(sp, "n") else np.stack([rs.randint(0, int(k), size=n) for k in sp.nvec], -1).astype(np.int32)
        o1, r1, d1, i1 = one.step(a)
        o2, r2, d2, i2 = node.step(torch.as_tensor(a, device="cuda"))
        assert node._multi is not None                     # (the one-call path was taken)
        assert torch.equal(r1, r2.to("cuda:0")) and torch.equal(d1, d2.to("cuda:0")), t
        assert torch.equal(o1["map"], o2["map"].to("cuda:0")) and torch.equal(o1["heatmap"], o2["heatmap"].to("cuda:0")), t
        assert torch.equal(i1.table, torch.cat([x.table for x in i2], 0)), t
    one.close(); node.close()


def test_adjust_param_reports_an_impossible_value_at_the_call():
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    env = BatchedPcgrlEnv(prob="sokoban", rep="narrow", num_envs=8, seed=1)
    env.reset()
    with pytest.raises(ValueError, match="solver_power"):
        env.adjust_param(solver_power=0)
    env.adjust_param(solver_power=5000)                 # (the handle is unharmed)
    env.step(np.zeros(8, np.int32))
    env.adjust_param(solver_power=20000)                # valid, but needs the other search arena: deferred to reset(), and step() says why
    with pytest.raises(RuntimeError, match="search arena"):
        env.step(np.zeros(8, np.int32))
    env.reset()
    env.step(np.zeros(8, np.int32))
    env.close()


@pytest.mark.parametrize("prob", ["sokoban", "mdungeon"])
def test_search_problem_tape_on_a_wide_one_row_map(prob):
    """70 x 1: the map takes the general map path (wider than 64 cells: no bit planes) while its level of 72 x 3 bordered cells still
    takes the compact searches -- pcgrl_rollout must not hand such a batch to the persistent-block kernel, which works on the
    planes."""
    rs = np.random.RandomState(8)
    err = ph.run_config(prob, "narrow", [dict(width=70, height=1), dict(change_percentage=0.3)], 48, 60, 515, rs, use_rollout=True)
    assert err is None, err


def test_heat_map_counts_beyond_int16():
    """One cell of a 182 x 182 binary-wide map rewritten 32 900 times (max_changes 33 124): the reference's float64 heat map passes
    32 767 (pcgrl_env.py:35,137; fixture heat_boundary.npz generated from it); the build's 16-bit counters must hold the same counts
    and the observation must show them unsigned."""
    import os
    import torch
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "heat_boundary.npz"))
    W, H, max_changes, max_iter, seed, T = [int(v) for v in d["cfg"]]
    x, y, _ = [int(v) for v in d["cell"]]
    env = BatchedPcgrlEnv(prob="binary", rep="wide", num_envs=2, seed=seed)
    env.adjust_param(width=W, height=H, probs={"empty": 0.0, "solid": 1.0})
    env.adjust_param(change_percentage=1.0)
    assert (env._max_changes, env._max_iterations) == (max_changes, max_iter)
    obs = env.reset()
    assert obs["heatmap"].dtype == torch.uint16
    acts = np.zeros((T, 2, 3), np.int32)
    acts[:, :, 0], acts[:, :, 1], acts[:, :, 2] = x, y, (np.arange(T) % 2)[:, None]
    rew, done, info = env.rollout(torch.as_tensor(acts, device="cuda"))
    steps = d["steps"]
    keys = list(env._prob.info_keys) + ["iterations", "changes"]
    got = np.stack([info[k].view(T, 2)[:, 0].cpu().numpy() for k in keys], 1).astype(np.int64)
    assert np.array_equal(rew[:, 0].cpu().numpy()[steps], d["reward"]) and np.array_equal(done[:, 0].cpu().numpy()[steps], d["done"])
    assert np.array_equal(got[steps], d["info"])
    heat = env._obs()["heatmap"][0].cpu().numpy().astype(np.int64)
    cells = np.argwhere(heat != 0)
    assert np.array_equal(cells, d["heat_cells"]) and np.array_equal(heat[cells[:, 0], cells[:, 1]], d["heat_counts"])
    assert np.array_equal(np.argwhere(env._bufs["map"][0].cpu().numpy() == 0), d["empty_cells"])
    env.close()


def test_sizes_the_library_takes_and_refuses():
    """The wide representation beyond 255 per side (no cursor to report: wide_rep.py:42-45); a cursor representation stays at 255;
    max_changes beyond the heat map's 16 bits is refused at the call."""
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    env = BatchedPcgrlEnv(prob="binary", rep="narrow", num_envs=2, seed=1)
    env.reset()
    with pytest.raises(ValueError):
        env.adjust_param(width=300, height=40)
    env.close()
    env = BatchedPcgrlEnv(prob="binary", rep="wide", num_envs=2, seed=1)
    env.reset()
    env.adjust_param(width=1000, height=80)
    with pytest.raises(ValueError):
        env.adjust_param(change_percentage=0.9)           # max_changes 72 000 > 65 535
    env.close()


@pytest.mark.parametrize("rep,calls", [("narrow", ()), ("turtle", ()), ("wide", (dict(width=40, height=12), dict(probs={"empty": 0.5, "solid": 0.35, "brick": 0.08})))])
def test_smb_kept_play_throughs_change_nothing(rep, calls):
    """smb: a change that keeps the cell blocked / free, or a cell the last play-through never read, keeps jumps / jumps-dist /
    dist-win without a search (k_update flags the job, k_smb takes the three from the previous statistics).  Against the same batch
    with the shortcut off (tuning no_inc: every change is played through), and the first environments against the oracle."""
    import torch
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    n, T, seed = 1536, 50, 90
    a_env = BatchedPcgrlEnv(prob="smb", rep=rep, num_envs=n, seed=seed)
    b_env = BatchedPcgrlEnv(prob="smb", rep=rep, num_envs=n, seed=seed, tuning={"no_inc": 1})
    for kw in calls:
        a_env.adjust_param(**kw); b_env.adjust_param(**kw)
    oa = a_env.reset(); ob = b_env.reset()
    assert torch.equal(oa["map"], ob["map"])
    sp = a_env.single_action_space
    rs = np.random.RandomState(4)
    n_or = 6
    orc = []
    for i in range(n_or):
        o = ol.OracleEnv("smb", rep)
        for kw in calls:
            o.adjust_param(**kw)
        o.seed(seed + i); o.reset()
        orc.append(o)
    keys = a_env._prob.info_keys
    for t in range(T):
        a = rs.randint(0, sp.n, size=n).astype(np.int32) if hasattr(sp, "n") else np.stack([rs.randint(0, int(k), size=n) for k in sp.nvec], -1).astype(np.int32)
        o1, r1, d1, i1 = a_env.step(a)
        o2, r2, d2, i2 = b_env.step(a)
        assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(i1.table, i2.table) and torch.equal(o1["map"], o2["map"]), t
        rr, dd, tab = r1.cpu().numpy(), d1.cpu().numpy(), i1.table.cpu().numpy()
        for i, o in enumerate(orc):
            _, er, ed, einf = o.step(a[i])
            assert er == rr[i] and ed == bool(dd[i]) and [einf[k] for k in keys] == [int(v) for v in tab[i, :len(keys)]], (t, i)
            if ed:
                o.reset()
    a_env.close(); b_env.close()


@pytest.mark.parametrize("team", [0, 2, 3, 4, 8])
def test_big_binary_maps_by_whole_blocks_vs_oracle(team, monkeypatch):
    """k_big, binary maps beyond 64 x 64: the few full recomputations of a step are made by all wavefronts of a block (bigmap_team.h:
    a band of rows per wavefront, crossing components by wavefront 0, sweeps claimed largest first).  Every step against the oracle,
    with 2 / 3 / 4 / 8 wavefronts a block and with the one-wavefront form (0); short episodes, so that resets -- and the first
    statistics of fresh maps -- come through the same path; 100 x 100 (two words a row) and 130 x 70 (three)."""
    from gym_pcgrl_amd import _lib
    monkeypatch.setitem(_lib.TUNING_OVERRIDES, "big_team", team)
    rs = np.random.RandomState(40 + team)
    err = ph.run_config("binary", "narrow", [dict(width=100, height=100), dict(change_percentage=0.002)], 6, 150, 8100, rs, use_rollout=False)
    assert err is None, err
    err = ph.run_config("binary", "turtle", [dict(width=130, height=70), dict(change_percentage=0.001)], 5, 120, 8200, rs, use_rollout=False)
    assert err is None, err
