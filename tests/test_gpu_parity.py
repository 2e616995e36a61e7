"""GPU parity tests (run with `-m gpu` on the MI355X box).  Everything goes through the C ABI
(gym_pcgrl_amd/lib/libpcgrl_hip.so) via BatchedPcgrlEnv / PcgrlEnv and is compared bit-exactly with
(a) the committed golden fixtures produced by the reference and (b) the CPU oracle on seeded inputs.
Nothing here reads /root/reference."""
import ast
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUPPORTED = ("binary", "zelda", "sokoban", "mdungeon", "ddave", "smb")


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _make(prob, rep, n, calls=(), seed=1000, auto_reset=True):
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, seed=seed, auto_reset=auto_reset)
    for kw in calls:
        env.adjust_param(**kw)
    return env


def _tune(monkeypatch, field, value):
    """A developer switch of the library (include/pcgrl_hip.h pcgrl_tuning) for the environments made in this test."""
    from gym_pcgrl_amd import _lib
    monkeypatch.setitem(_lib.TUNING_OVERRIDES, field, int(value))


def _supported(prob):
    return prob in SUPPORTED


# ------------------------------------------------------------------ map -> stats known answers
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "stats_*.npz"))), ids=os.path.basename)
def test_stats_kat(path):
    _torch()
    d = np.load(path)
    prob = os.path.basename(path).split("_")[1]
    if not _supported(prob):
        pytest.skip("%s not built yet" % prob)
    maps = d["maps"]
    n, h, w = maps.shape
    calls = [dict(width=w, height=h)]
    if "solver_power" in d.files:
        calls.append(dict(solver_power=int(d["solver_power"])))
    env = _make(prob, "wide", n, calls)
    if prob == "smb":       # smb_prob.py:40-53: solver_power is an attribute there, not an adjust_param key
        env._prob._solver_power = int(d["solver_power"])
        env.adjust_param()
    env.reset()
    env.set_maps(maps)
    got = env.stats.cpu().numpy().astype(np.int64)
    assert env.check_status() == 0
    bad = np.nonzero((got != d["stats"]).any(1))[0]
    assert bad.size == 0, (bad[:5], got[bad[:5]], d["stats"][bad[:5]])
    assert np.array_equal(env._bufs["map"].cpu().numpy(), maps)


# ------------------------------------------------------------------ golden trajectories
def _compare_traj(env, d, T, check_every=1):
    torch = _torch()
    rep = str(d["rep"])
    keys = [str(k) for k in d["info_keys"]]
    acts = d["actions"]
    for t in range(T):
        a = acts[t] if acts.shape[2] > 1 else acts[t, :, 0]
        obs, rew, done, info = env.step(a)
        if t % check_every:
            continue
        torch.cuda.synchronize()
        assert np.array_equal(done.cpu().numpy(), d["done"][t]), ("done", t)
        assert np.array_equal(rew.cpu().numpy(), d["reward"][t]), ("reward", t, rew.cpu().numpy(), d["reward"][t])
        got_info = np.stack([info[k].cpu().numpy() for k in keys], 1).astype(np.int64)
        assert np.array_equal(got_info, d["info"][t]), ("info", t, got_info, d["info"][t])
        assert np.array_equal(obs["map"].cpu().numpy(), d["maps"][t]), ("map", t)
        if rep != "wide":
            assert np.array_equal(obs["pos"].cpu().numpy(), d["pos"][t]), ("pos", t)
        assert np.array_equal(obs["heatmap"].cpu().numpy().astype(np.uint16), d["heatmap"][t]), ("heatmap", t)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "traj_*.npz"))), ids=os.path.basename)
def test_golden_trajectory(path):
    _torch()
    d = np.load(path)
    prob, rep = str(d["prob"]), str(d["rep"])
    if not _supported(prob):
        pytest.skip("%s not built yet" % prob)
    calls = ast.literal_eval(str(d["calls"]))
    W, H, max_changes, max_iter, seed0, _ = [int(v) for v in d["cfg"]]
    T, E = d["actions"].shape[:2]
    env = _make(prob, rep, E, calls, seed=seed0)
    assert (env._max_changes, env._max_iterations) == (max_changes, max_iter)
    obs = env.reset()
    assert np.array_equal(obs["map"].cpu().numpy(), d["map0"])
    if rep != "wide":
        assert np.array_equal(obs["pos"].cpu().numpy(), d["pos0"])
    assert int(obs["heatmap"].abs().sum().item()) == 0
    _compare_traj(env, d, T)


_SWITCH_TRAJ = [("no_fused", p) for p in sorted(glob.glob(os.path.join(G, "traj_binary_*.npz")))
                if "64" not in os.path.basename(p) and "40x33" not in os.path.basename(p) and "cast" not in os.path.basename(p)
                and "multi" not in os.path.basename(p)] + \
               [("fused_zelda", p) for p in sorted(glob.glob(os.path.join(G, "traj_zelda_*.npz")))
                if "cast" not in os.path.basename(p)]


@pytest.mark.gpu
@pytest.mark.parametrize("switch,path", _SWITCH_TRAJ, ids=lambda v: os.path.basename(v) if v.endswith(".npz") else v)
def test_golden_trajectory_other_step_pipeline(switch, path, monkeypatch):
    """Binary and zelda maps of at most 16 rows take the fused one-launch step (k_step) by default; the library switches
    (pcgrl_tuning: no_fused = 1, fused_zelda = 0) select the two-launch pipeline (k_update + k_stats), which must reproduce the
    reference's trajectories just the same."""
    _tune(monkeypatch, switch, 0 if switch == "fused_zelda" else 1)
    test_golden_trajectory(path)


# ------------------------------------------------------------------ seeded rollouts against the oracle
ORACLE_CASES = [
    ("binary", "narrow", (), 192, 160),
    ("binary", "turtle", (dict(change_percentage=0.4),), 96, 200),
    ("binary", "wide", (dict(width=33, height=20),), 64, 80),
    ("binary", "narrow", (dict(width=40, height=12), dict(change_percentage=0.2)), 64, 150),
    ("zelda", "wide", (dict(width=11, height=16),), 192, 120),
    ("zelda", "narrow", (), 128, 200),
    ("zelda", "turtle", (dict(width=20, height=18), dict(change_percentage=0.5)), 48, 150),
    ("sokoban", "narrow", (), 256, 120),
    ("sokoban", "wide", (dict(width=6, height=6), dict(change_percentage=0.5)), 96, 100),
    ("mdungeon", "narrow", (), 128, 120),
    ("mdungeon", "wide", (dict(width=6, height=5), dict(change_percentage=0.7, probs={"empty": 0.7, "solid": 0.08},
                                                         target_solution=4, target_col_enemies=0.25)), 192, 150),
    ("ddave", "narrow", (), 128, 120),
    ("ddave", "wide", (dict(width=6, height=5), dict(change_percentage=0.7, probs={"empty": 0.7, "solid": 0.1, "spike": 0.02},
                                                      target_solution=3, target_jumps=0)), 192, 150),
    ("ddave", "turtle", (dict(width=9, height=7), dict(change_percentage=0.5, solver_power=150,
                                                        probs={"empty": 0.6, "solid": 0.15, "player": 0.03, "exit": 0.03, "key": 0.03})), 96, 150),
    ("mdungeon", "turtle", (dict(width=8, height=8), dict(change_percentage=0.5, solver_power=200,
                                                           probs={"empty": 0.5, "solid": 0.03, "ogre": 0.2, "goblin": 0.1})), 96, 150),
    ("smb", "narrow", (), 24, 60),
    ("smb", "wide", (dict(width=40, height=12), dict(change_percentage=0.1, probs={"empty": 0.5, "solid": 0.35, "brick": 0.08})), 64, 100),
    ("smb", "turtle", (dict(width=150, height=9), dict(change_percentage=0.05, probs={"empty": 0.55, "solid": 0.3}, min_empty=500, min_jumps=3,
                                                       rewards={"noise": 1.5, "jumps-dist": 0.5})), 32, 80),
    ("smb", "narrow", (dict(width=22, height=7), dict(change_percentage=0.5, probs={"empty": 0.5, "solid": 0.45}, random_tile=False)), 48, 100),
    # round 4: maps beyond 64 x 64 (bigmap.h: a wavefront per map on multi-word row masks; episodes end often: tiny change budgets)
    ("binary", "narrow", (dict(width=90, height=70), dict(change_percentage=0.002)), 40, 80),
    ("binary", "turtle", (dict(width=130, height=12), dict(change_percentage=0.004)), 33, 80),
    ("binary", "wide", (dict(width=7, height=200), dict(change_percentage=0.004)), 20, 60),
    ("zelda", "wide", (dict(width=65, height=30), dict(change_percentage=0.003)), 40, 80),
    ("zelda", "narrow", (dict(width=20, height=66), dict(change_percentage=0.004, probs={"empty": 0.93, "solid": 0.03, "player": 0.002, "key": 0.002,
                                                                                      "door": 0.002, "bat": 0.01, "scorpion": 0.01, "spider": 0.01})), 40, 120),
    # ... and the search problems beyond the compact searches (search_big.h): levels of more than 256 bordered cells, solver_power > 16 383
    ("sokoban", "narrow", (dict(width=20, height=20), dict(change_percentage=0.02, solver_power=300,
                                                           probs={"empty": 0.93, "solid": 0.04, "player": 0.003, "crate": 0.003, "target": 0.003})), 48, 100),
    ("mdungeon", "wide", (dict(width=22, height=18), dict(change_percentage=0.02, solver_power=250,
                                                          probs={"empty": 0.9, "solid": 0.05, "player": 0.003, "exit": 0.003, "potion": 0.01, "treasure": 0.01,
                                                                 "goblin": 0.01, "ogre": 0.01})), 48, 100),
    ("ddave", "turtle", (dict(width=30, height=12), dict(change_percentage=0.02, solver_power=250,
                                                         probs={"empty": 0.8, "solid": 0.18, "player": 0.004, "exit": 0.004, "diamond": 0.004, "key": 0.004,
                                                                "spike": 0.004})), 48, 100),
    ("sokoban", "wide", (dict(width=70, height=40), dict(change_percentage=0.001, solver_power=100)), 10, 40),
    # round 5: levels of more than 4 096 bordered cells (up to 16 384)
    ("sokoban", "narrow", (dict(width=70, height=70), dict(change_percentage=0.0008, solver_power=120,
                                                           probs={"empty": 0.96, "solid": 0.03, "player": 0.0003, "crate": 0.0003, "target": 0.0003})), 8, 30),
    ("mdungeon", "wide", (dict(width=126, height=126), dict(change_percentage=0.0003, solver_power=80)), 4, 16),
    ("sokoban", "wide", (dict(solver_power=17000, change_percentage=0.9, probs={"empty": 0.8, "solid": 0.05, "player": 0.05, "crate": 0.05, "target": 0.05}),), 64, 60),
]


@pytest.mark.parametrize("prob,rep,calls,E,T", ORACLE_CASES, ids=lambda v: str(v) if isinstance(v, (str, int)) else "cfg")
def test_rollout_vs_oracle(prob, rep, calls, E, T):
    torch = _torch()
    if not _supported(prob):
        pytest.skip("%s not built yet" % prob)
    seed0 = 777
    env = _make(prob, rep, E, calls, seed=seed0)
    obs = env.reset()
    W, H = env._prob._width, env._prob._height
    nt = env.get_num_tiles()
    rs = np.random.RandomState(11)
    if rep == "narrow":
        acts = rs.randint(0, nt + 1, size=(T, E, 1))
    elif rep == "turtle":
        acts = rs.randint(0, nt + 4, size=(T, E, 1))
    else:
        acts = np.stack([rs.randint(0, W, size=(T, E)), rs.randint(0, H, size=(T, E)), rs.randint(0, nt, size=(T, E))], -1)
    acts = acts.astype(np.int32)
    # oracle side
    exp = []
    map0 = np.zeros((E, H, W), np.uint8)
    for i in range(E):
        o = ol.OracleEnv(prob, rep)
        for kw in calls:
            o.adjust_param(**kw)
        o.seed(seed0 + i)
        map0[i] = o.reset()["map"]
        a3 = np.zeros((T, 3), np.int32)
        a3[:, :acts.shape[2]] = acts[:, i]
        exp.append(o.rollout(a3))
    assert np.array_equal(obs["map"].cpu().numpy(), map0)
    ni = len(env._prob.info_keys)
    keys = list(env._prob.info_keys) + ["iterations", "changes"]
    for t in range(T):
        a = acts[t] if rep == "wide" else acts[t, :, 0]
        obs, rew, done, info = env.step(a)
        torch.cuda.synchronize()
        e_done = np.array([x["done"][t] for x in exp])
        e_rew = np.array([x["reward"][t] for x in exp])
        e_info = np.stack([x["info"][t] for x in exp])
        e_map = np.stack([x["maps"][t] for x in exp])
        e_heat = np.stack([x["heatmap"][t] for x in exp])
        assert np.array_equal(done.cpu().numpy(), e_done), ("done", t)
        assert np.array_equal(rew.cpu().numpy(), e_rew), ("reward", t)
        got_info = np.stack([info[k].cpu().numpy() for k in keys], 1).astype(np.int64)
        assert np.array_equal(got_info, e_info), ("info", t, np.nonzero((got_info != e_info).any(1))[0][:4])
        assert np.array_equal(obs["map"].cpu().numpy(), e_map), ("map", t)
        assert np.array_equal(obs["heatmap"].cpu().numpy().astype(np.uint16), e_heat), ("heat", t)
        if rep != "wide":
            e_pos = np.stack([x["pos"][t] for x in exp]).astype(np.uint8)
            assert np.array_equal(obs["pos"].cpu().numpy(), e_pos), ("pos", t)


# ------------------------------------------------------------------ single-env facade (reference surface)
def test_facade_matches_golden():
    _torch()
    import gym_pcgrl_amd
    d = np.load(os.path.join(G, "traj_binary_narrow.npz"))
    env = gym_pcgrl_amd.make("binary-narrow-v0")
    assert "PcgrlEnv" in str(type(env))
    assert env.seed(1000) == [1000]
    o = env.reset()
    assert list(o.keys()) == ["pos", "map", "heatmap"]
    assert o["map"].dtype == np.uint8 and o["pos"].dtype == np.uint8 and o["heatmap"].dtype == np.float64
    assert np.array_equal(o["map"], d["map0"][0]) and np.array_equal(o["pos"], d["pos0"][0])
    keys = [str(k) for k in d["info_keys"]]
    for t in range(150):
        o, r, dn, info = env.step(int(d["actions"][t, 0, 0]))
        assert r == d["reward"][t, 0] and dn == bool(d["done"][t, 0])
        assert [info[k] for k in keys] == list(d["info"][t, 0])
        assert info["max_changes"] == 39 and info["max_iterations"] == 7644
        if dn:
            o = env.reset()
        assert np.array_equal(o["map"], d["maps"][t, 0]) and np.array_equal(o["pos"], d["pos"][t, 0])
        assert np.array_equal(o["heatmap"], d["heatmap"][t, 0].astype(np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("name,env_id", [("traj_mdungeon_narrow", "mdungeon-narrow-v0"), ("traj_ddave_narrow", "ddave-narrow-v0")])
def test_facade_packed_problems(name, env_id):
    """The single-environment facade on the two problems whose device rows pack several statistics into a slot: the info dict
    has the reference's keys and values (ddave: get_debug_info's own key order, without dist-floor)."""
    _torch()
    import gym_pcgrl_amd
    d = np.load(os.path.join(G, name + ".npz"))
    env = gym_pcgrl_amd.make(env_id)
    env.seed(int(d["cfg"][4]))
    o = env.reset()
    assert np.array_equal(o["map"], d["map0"][0])
    keys = [str(k) for k in d["info_keys"]]
    for t in range(80):
        o, r, dn, info = env.step(int(d["actions"][t, 0, 0]))
        assert r == d["reward"][t, 0] and dn == bool(d["done"][t, 0])
        assert [info[k] for k in keys] == list(d["info"][t, 0])
        assert set(info) == set(keys) | {"max_iterations", "max_changes"}
        if dn:
            o = env.reset()
        assert np.array_equal(o["map"], d["maps"][t, 0])


# ------------------------------------------------------------------ shard invariance / determinism
def test_shard_invariance_and_determinism():
    torch = _torch()
    N, T = 1024, 40
    rs = np.random.RandomState(5)
    acts = rs.randint(0, 3, size=(T, N)).astype(np.int32)

    def run(lo, hi):
        env = _make("binary", "narrow", hi - lo, seed=lo)
        env.reset()
        outs = []
        for t in range(T):
            obs, rew, done, info = env.step(acts[t, lo:hi])
            outs.append((obs["map"].clone(), obs["pos"].clone(), rew.clone(), done.clone(), info.table.clone()))
        torch.cuda.synchronize()
        return outs

    full = run(0, N)
    again = run(0, N)
    a, b = run(0, N // 2), run(N // 2, N)
    for t in range(T):
        for k in range(5):
            assert torch.equal(full[t][k], again[t][k]), ("determinism", t, k)
            assert torch.equal(full[t][k], torch.cat([a[t][k], b[t][k]])), ("shard", t, k)


# ------------------------------------------------------------------ full-size properties (BASELINE configs)
@pytest.mark.parametrize("prob,rep,calls,N", [
    ("binary", "narrow", (), 65536),
    ("zelda", "wide", (dict(width=11, height=16),), 65536),
    ("binary", "turtle", (dict(width=64, height=64),), 8192),
    ("sokoban", "narrow", (), 131072),
    ("mdungeon", "narrow", (), 65536),
    ("ddave", "narrow", (), 65536),
], ids=["C2", "C3", "C5", "C4", "M1", "D1"])
def test_full_size_properties(prob, rep, calls, N):
    torch = _torch()
    if not _supported(prob):
        pytest.skip("%s not built yet" % prob)
    env = _make(prob, rep, N, calls, seed=0)
    env.reset()
    W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
    g = torch.Generator(device="cuda").manual_seed(1234)
    total_done = 0
    for t in range(30):
        if rep == "wide":
            a = torch.stack([torch.randint(0, W, (N,), generator=g, device="cuda"), torch.randint(0, H, (N,), generator=g, device="cuda"),
                             torch.randint(0, nt, (N,), generator=g, device="cuda")], 1).int()
        else:
            a = torch.randint(0, nt + (1 if rep == "narrow" else 4), (N,), generator=g, device="cuda").int()
        obs, rew, done, info = env.step(a)
        total_done += int(done.sum().item())
        # environments that were done have been reset: counters are zero, heatmap is clear
        cnt = env._bufs["counters"]
        assert int(cnt[done].abs().sum().item()) == 0
        assert int(obs["heatmap"][done].abs().sum().item()) == 0
        # info mirrors the counters for the others
        assert torch.equal(info["iterations"][~done], cnt[~done][:, 0])
    # planes are the bit planes of the byte map
    m = obs["map"].long()
    planes = env._bufs["planes"].long()
    xs = torch.arange(W, device="cuda")
    for b in range(planes.shape[2]):
        rows = (((m >> b) & 1) << xs).sum(-1)
        got = planes[:, :H, b]
        if env._bufs["planes"].dtype == torch.int32:
            got = got & 0xFFFFFFFF
        assert torch.equal(rows, got), ("plane", b)
    # current stats of a sample equal the oracle's stats of the current maps
    idx = np.linspace(0, N - 1, 200).astype(int)
    maps = obs["map"][idx].cpu().numpy()
    st = env.stats[idx].cpu().numpy().astype(np.int64)
    power = getattr(env._prob, "_solver_power", 5000)
    for k, i in enumerate(idx):
        assert np.array_equal(st[k], ol.get_stats(prob, maps[k], solver_power=power)), (i, st[k])
    assert total_done > 0 or prob == "binary"


# ------------------------------------------------------------------ batched wrappers
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "wrap_*.npz"))), ids=os.path.basename)
def test_wrapper_observations_match_reference(path):
    torch = _torch()
    from gym_pcgrl_amd import wrappers
    d = np.load(path)
    game, kind, size = str(d["game"]), str(d["kind"]), int(d["size"])
    T, E = d["actions"].shape
    if kind == "cropped":
        w = wrappers.CroppedImagePCGRLWrapper(game, size, num_envs=E, seed=int(d["seed0"]))
    else:
        w = wrappers.ActionMapImagePCGRLWrapper(game, num_envs=E, seed=int(d["seed0"]))
    obs = w.reset()
    assert obs.dtype == torch.uint8 and tuple(obs.shape) == tuple(d["obs0"].shape)
    assert np.array_equal(obs.cpu().numpy(), d["obs0"])
    for t in range(T):
        obs, rew, done, info = w.step(d["actions"][t])
        assert np.array_equal(done.cpu().numpy(), d["done"][t]), t
        assert np.array_equal(rew.cpu().numpy(), d["reward"][t]), t
        assert np.array_equal(obs.cpu().numpy(), d["obs"][t]), t


def test_composable_wrappers_match_the_composites_and_the_reference_rule():
    """The single-purpose wrappers (wrappers.py:18-206) as separately composable classes: the chain the reference's composite
    builds gives the composite's image; ToImage stacks several entries; ActionMap on a representation with a cursor follows the
    reference's rule (step with the tile where the cursor stands on the chosen cell, else with the tile under the cursor)."""
    torch = _torch()
    import gym_pcgrl_amd as gp
    from gym_pcgrl_amd import wrappers as wr
    N, T = 70, 30
    for game, size in (("zelda-narrow-v0", 22), ("binary-turtle-v0", 28)):
        comp = wr.CroppedImagePCGRLWrapper(game, size, num_envs=N, seed=31)
        env = gp.make_batched(game, num_envs=N, seed=31)
        chain = wr.Cropped(env, size, env.get_border_tile(), "map")
        if "binary" not in game:
            chain = wr.OneHotEncoding(chain, "map")
        chain = wr.ToImage(chain, ["map"])
        assert tuple(chain.observation_space.shape) == tuple(comp._bind().shape[1:])
        a, b = comp.reset(), chain.reset()
        assert torch.equal(a, b)
        rs = np.random.RandomState(4)
        n_act = env.action_space.n
        for t in range(T):
            act = torch.as_tensor(rs.randint(0, n_act, size=N).astype(np.int32), device="cuda")
            (a, ra, da, _), (b, rb, db, _) = comp.step(act), chain.step(act)
            assert torch.equal(a, b) and torch.equal(ra, rb) and torch.equal(da, db), (game, t)
        comp.close(); chain.close()
    # several entries in one image
    env = gp.make_batched("binary-narrow-v0", num_envs=9, seed=3)
    img = wr.ToImage(env, ["map", "heatmap"])
    o = img.reset()
    o, _, _, _ = img.step(torch.ones(9, dtype=torch.int32, device="cuda"))
    assert tuple(o.shape) == (9, 14, 14, 2)
    assert torch.equal(o[..., 0].to(torch.uint8), env._bufs["map"]) and torch.equal(o[..., 1].to(torch.int16), env._bufs["heatmap"])
    img.close()
    # ActionMap with a cursor against a twin stepped with the inner actions worked out by hand
    am = wr.ActionMap(gp.make_batched("zelda-narrow-v0", num_envs=N, seed=8))
    twin = gp.make_batched("zelda-narrow-v0", num_envs=N, seed=8)
    am.reset(); obs = twin.reset()
    rs = np.random.RandomState(6)
    for t in range(T):
        flat = rs.randint(0, am.action_space.n, size=N)
        y, x, v = np.unravel_index(flat, (am.h, am.w, am.dim))
        pos = obs["pos"].cpu().numpy().astype(int)
        m = obs["map"].cpu().numpy()
        inner = np.where((pos[:, 0] == x) & (pos[:, 1] == y), v, m[np.arange(N), pos[:, 1], pos[:, 0]]).astype(np.int32)
        o1, r1, d1, _ = am.step(torch.as_tensor(flat, device="cuda"))
        obs, r2, d2, _ = twin.step(torch.as_tensor(inner, device="cuda"))
        assert torch.equal(o1["map"], obs["map"]) and torch.equal(o1["pos"], obs["pos"]) and torch.equal(r1, r2) and torch.equal(d1, d2), t
    am.close(); twin.close()


def _expected_image(m, pos, oh, ow, centered, pad, depth):
    """wrappers.py restated with numpy: Cropped.transform :197-206 (np.pad with the border tile, window at the cursor),
    OneHotEncoding.transform :101-104 (np.eye(dim)[map]), ToImage.transform :53-60.  m [N,H,W], pos [N,2] = (x, y)."""
    n, H, W = m.shape
    out = np.full((n, oh, ow), pad, dtype=np.int64)
    for i in range(n):
        if centered:
            ph, pw = oh // 2 + oh, ow // 2 + ow          # generous padding: windows of any size fit
            padded = np.pad(m[i].astype(np.int64), ((ph, ph), (pw, pw)), constant_values=pad)
            x, y = int(pos[i, 0]), int(pos[i, 1])
            out[i] = padded[y + ph - oh // 2: y + ph - oh // 2 + oh, x + pw - ow // 2: x + pw - ow // 2 + ow]
        else:
            hh, ww = min(oh, H), min(ow, W)
            out[i, :hh, :ww] = m[i, :hh, :ww]
    if depth == 1:
        return out[..., None].astype(np.uint8)
    return np.eye(depth, dtype=np.uint8)[out]


@pytest.mark.parametrize("prob,rep,calls,n,oh,ow,centered,onehot", [
    # the fused step kernel writes the image (binary / zelda, at most 16 rows, single-cell representations)
    ("binary", "narrow", (), 200, 28, 28, 1, 0), ("binary", "turtle", (), 65, 5, 7, 1, 0), ("binary", "narrow", (), 64, 31, 17, 1, 1),
    ("binary", "wide", (), 130, 14, 14, 0, 0), ("binary", "narrow", (dict(width=30, height=9),), 70, 40, 66, 1, 0),
    ("binary", "narrow", (dict(width=40, height=12),), 50, 12, 40, 0, 0),
    ("zelda", "narrow", (), 100, 22, 22, 1, 1), ("zelda", "wide", (dict(width=11, height=16),), 129, 16, 11, 0, 1),
    ("zelda", "turtle", (), 64, 9, 9, 1, 0), ("zelda", "wide", (), 77, 7, 11, 0, 0),
    # one more kernel after the step (k_obs): search problems, tall maps, block representations, smb's byte maps
    ("sokoban", "narrow", (), 150, 10, 10, 1, 1), ("sokoban", "wide", (), 33, 5, 5, 0, 1), ("mdungeon", "turtle", (), 70, 14, 14, 1, 1),
    ("ddave", "narrow", (), 41, 22, 22, 1, 1), ("binary", "turtle", (dict(width=64, height=64),), 20, 28, 28, 1, 0),
    ("binary", "narrow", (dict(width=20, height=30),), 66, 28, 28, 1, 1), ("zelda", "narrowcast", (), 48, 22, 22, 1, 1),
    ("binary", "narrowmulti", (), 35, 28, 28, 1, 0), ("smb", "narrow", (dict(width=30, height=8),), 9, 12, 20, 1, 1),
    # windows of 65 536 bytes and more per environment (k_obs_huge: 64-bit offsets; the reference's Cropped takes any crop_size):
    # a 228 x 228 x 7 crop of an smb level, a 128 x 128 one-hot crop, a huge id window on a batch that ends mid-piece
    ("smb", "narrow", (), 3, 228, 228, 1, 1), ("zelda", "narrow", (), 5, 128, 128, 1, 1), ("binary", "turtle", (dict(width=20, height=30),), 7, 257, 255, 1, 0),
], ids=lambda v: str(v) if not isinstance(v, tuple) else "adj%d" % len(v))
def test_bound_observation_every_step(prob, rep, calls, n, oh, ow, centered, onehot):
    """pcgrl_bind_observation: after reset, after every step (auto-resets included) and after a rollout the bound tensor
    holds the wrappers' image of the returned state -- fused into k_step where that kernel runs, k_obs elsewhere; batch
    sizes that leave a partial last block, windows larger than the map, images that are not a multiple of 16 bytes."""
    torch = _torch()
    env = _make(prob, rep, n, list(calls) + [dict(change_percentage=0.3, solver_power=200)], seed=77)
    if prob == "smb":
        env._prob._solver_power = 300
    pad = env.get_border_tile()
    depth = env.get_num_tiles() if onehot else 1
    img = env.bind_observation(oh, ow, centered, pad, onehot)
    obs = env.reset()
    has_pos = env._rep.has_pos

    def check(tag):
        m = env._bufs["map"].cpu().numpy()
        pos = env._bufs["pos"].cpu().numpy() if has_pos else np.zeros((n, 2), np.uint8)
        exp = _expected_image(m, pos, oh, ow, centered, pad, depth)
        got = img.cpu().numpy()
        bad = np.nonzero((got != exp).reshape(n, -1).any(1))[0]
        assert bad.size == 0, (tag, bad[:5])

    check("reset")
    sp = env.single_action_space
    rs = np.random.RandomState(5)
    T = (12 if prob == "smb" else 40) if oh * ow * depth < 65536 else 6
    draw = (lambda: rs.randint(0, sp.n, size=(n,))) if hasattr(sp, "n") else (lambda: np.stack([rs.randint(0, int(k), size=(n,)) for k in sp.nvec], -1))
    ndone = 0
    for t in range(T):
        _, _, done, _ = env.step(torch.as_tensor(draw().astype(np.int32), device="cuda"))
        ndone += int(done.sum().item())
        check(("step", t))
    tape = np.stack([draw() for _ in range(7)]).astype(np.int32)
    env.rollout(torch.as_tensor(tape, device="cuda"))
    check("rollout")
    # redirecting the target, the explicit call, and switching the feature off
    other = torch.zeros_like(img)
    env.set_observation_target(other)
    before = img.clone()
    env.step(torch.as_tensor(draw().astype(np.int32), device="cuda"))
    assert torch.equal(img, before)
    img = other
    check("retarget")
    env.unbind_observation()
    env.step(torch.as_tensor(draw().astype(np.int32), device="cuda"))
    assert torch.equal(other, img)
    assert env.check_status() == 0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("at_end", [0, 1])
@pytest.mark.parametrize("epb", [64, 128, 256])
@pytest.mark.parametrize("prob,rep,calls,oh,ow,centered,onehot", [
    ("binary", "narrow", (), 28, 28, 1, 0), ("zelda", "wide", (dict(width=11, height=16),), 16, 11, 0, 1), ("zelda", "narrow", (), 22, 22, 1, 1),
    ("binary", "turtle", (), 28, 28, 1, 0)], ids=lambda v: str(v) if not isinstance(v, tuple) else "adj%d" % len(v))
def test_images_written_while_the_step_runs(prob, rep, calls, oh, ow, centered, onehot, epb, at_end, monkeypatch):
    """k_step with a bound observation writes the images while it runs (round 6): a reset's wavefront its new map's, the wide
    representation's lane the piece its change touches, observation tasks between the statistics tasks the rest, and the images of
    episode ends nobody saw coming again at the end -- for every block size, against the wrappers' rule after every step, with
    short episodes (many certain resets, and unannounced ones: a tenth of the steps' resets), next to the form that writes
    everything at the end (pcgrl_tuning obs_at_end)."""
    torch = _torch()
    _tune(monkeypatch, "step_epb", str(epb))
    _tune(monkeypatch, "obs_at_end", str(at_end))
    n = 700
    env = _make(prob, rep, n, list(calls) + [dict(change_percentage=0.08)], seed=4100 + epb)
    pad = env.get_border_tile()
    depth = env.get_num_tiles() if onehot else 1
    img = env.bind_observation(oh, ow, centered, pad, onehot)
    env.reset()
    has_pos = env._rep.has_pos
    sp = env.single_action_space
    rs = np.random.RandomState(9)
    draw = (lambda: rs.randint(0, sp.n, size=(n,))) if hasattr(sp, "n") else (lambda: np.stack([rs.randint(0, int(k), size=(n,)) for k in sp.nvec], -1))
    ndone = 0
    for t in range(60):
        _, _, done, _ = env.step(torch.as_tensor(draw().astype(np.int32), device="cuda"))
        ndone += int(done.sum().item())
        m = env._bufs["map"].cpu().numpy()
        pos = env._bufs["pos"].cpu().numpy() if has_pos else np.zeros((n, 2), np.uint8)
        exp = _expected_image(m, pos, oh, ow, centered, pad, depth)
        bad = np.nonzero((img.cpu().numpy() != exp).reshape(n, -1).any(1))[0]
        assert bad.size == 0, (t, bad[:5])
    assert ndone > 20 and env.check_status() == 0
    env.close()


# ------------------------------------------------------------------ edge shapes (layout corners of the kernels)
@pytest.mark.parametrize("prob,rep,w,h", [
    ("binary", "narrow", 1, 1), ("binary", "turtle", 1, 9), ("binary", "wide", 9, 1), ("binary", "narrow", 32, 16),
    ("binary", "narrow", 33, 16), ("binary", "turtle", 64, 17), ("binary", "narrowmulti", 16, 17), ("binary", "wide", 64, 64),
    ("zelda", "narrow", 2, 2), ("zelda", "turtlecast", 33, 3), ("zelda", "wide", 64, 5), ("zelda", "narrowcast", 5, 40),
    ("sokoban", "narrow", 1, 3), ("sokoban", "turtle", 14, 14), ("sokoban", "wide", 3, 8),
    ("mdungeon", "narrow", 1, 2), ("mdungeon", "turtlecast", 14, 14), ("mdungeon", "wide", 3, 9), ("mdungeon", "narrowmulti", 18, 10),
    ("ddave", "narrow", 1, 1), ("ddave", "turtle", 14, 14), ("ddave", "wide", 9, 3), ("ddave", "narrowcast", 12, 16),
], ids=lambda v: str(v))
def test_edge_shapes_vs_oracle(prob, rep, w, h):
    torch = _torch()
    E, T, seed0 = 5, 60, 31
    calls = (dict(width=w, height=h), dict(change_percentage=0.5, solver_power=300))
    env = _make(prob, rep, E, calls, seed=seed0)
    obs = env.reset()
    nt = env.get_num_tiles()
    rs = np.random.RandomState(3)
    aw = env._rep.action_width()
    if rep == "narrow":
        acts = rs.randint(0, nt + 1, size=(T, E, 1))
    elif rep == "turtle":
        acts = rs.randint(0, nt + 4, size=(T, E, 1))
    elif rep == "wide":
        acts = np.stack([rs.randint(0, w, size=(T, E)), rs.randint(0, h, size=(T, E)), rs.randint(0, nt, size=(T, E))], -1)
    elif rep == "narrowcast":
        acts = np.stack([rs.randint(0, 3, size=(T, E)), rs.randint(0, nt, size=(T, E))], -1)
    elif rep == "turtlecast":
        acts = np.stack([rs.randint(0, 6, size=(T, E)), rs.randint(0, nt, size=(T, E))], -1)
    else:
        acts = rs.randint(0, nt + 1, size=(T, E, 9))
    acts = acts.astype(np.int32)
    exp = []
    for i in range(E):
        o = ol.OracleEnv(prob, rep)
        for kw in calls:
            o.adjust_param(**kw)
        o.seed(seed0 + i)
        m0 = o.reset()["map"]
        assert np.array_equal(obs["map"][i].cpu().numpy(), m0)
        exp.append(o.rollout(acts[:, i]))
    keys = list(env._prob.info_keys) + ["iterations", "changes"]
    for t in range(T):
        obs, rew, done, info = env.step(acts[t] if aw > 1 else acts[t, :, 0])
        assert np.array_equal(obs["map"].cpu().numpy(), np.stack([x["maps"][t] for x in exp])), ("map", t)
        assert np.array_equal(rew.cpu().numpy(), np.array([x["reward"][t] for x in exp])), ("reward", t)
        assert np.array_equal(done.cpu().numpy(), np.array([x["done"][t] for x in exp])), ("done", t)
        got_info = np.stack([info[k].cpu().numpy() for k in keys], 1).astype(np.int64)
        assert np.array_equal(got_info, np.stack([x["info"][t] for x in exp])), ("info", t)
        assert np.array_equal(obs["heatmap"].cpu().numpy().astype(np.uint16), np.stack([x["heatmap"][t] for x in exp])), ("heat", t)
        if rep != "wide":
            assert np.array_equal(obs["pos"].cpu().numpy(), np.stack([x["pos"][t] for x in exp]).astype(np.uint8)), ("pos", t)
    assert env.check_status() == 0


def test_make_vec_envs_surface():
    torch = _torch()
    from gym_pcgrl_amd.utils import make_vec_envs
    v = make_vec_envs("zelda-narrow-v0", "narrow", None, 64, seed=5, cropped_size=22)
    assert v.num_envs == 64 and v.observation_space.shape == (22, 22, 8) and v.action_space.n == 9
    obs = v.reset()
    assert tuple(obs.shape) == (64, 22, 22, 8) and int(obs.sum().item()) == 64 * 22 * 22
    obs, r, d, info = v.step(torch.randint(0, 9, (64,), device="cuda"))
    assert tuple(r.shape) == (64,) and d.dtype == torch.bool
    v.close()
    v = make_vec_envs("binary-wide-v0", "wide", None, 32, seed=5)
    assert v.observation_space.shape == (14, 14, 1) and v.action_space.n == 14 * 14 * 2
    v.reset()
    v.step_async(torch.randint(0, 392, (32,), device="cuda"))
    obs, r, d, info = v.step_wait()
    assert tuple(obs.shape) == (32, 14, 14, 1)
    v.close()


# ------------------------------------------------------------------ episode statistics + rollout collection
@pytest.mark.gpu
@pytest.mark.parametrize("env_id", ["binary-narrow-v0", "zelda-wide-v0", "sokoban-turtle-v0", "mdungeon-narrow-v0", "ddave-wide-v0"])
def test_episode_stats_match_oracle_sums(env_id):
    """The in-kernel Monitor (pcgrl_bind_episode_stats): return and length latched when an episode ends must equal
    the sums over the oracle's rewards of that episode, for every environment and every episode."""
    import gym_pcgrl_amd as gp
    torch = _torch()
    prob, rep = env_id.split("-")[0], env_id.split("-")[1]
    N, T = 48, 160
    env = gp.make_batched(env_id, num_envs=N, seed=700)
    env.enable_episode_stats()
    env.reset()
    rs = np.random.RandomState(11)
    sp = env.single_action_space
    if hasattr(sp, "n"):
        acts = rs.randint(0, sp.n, size=(T, N)).astype(np.int32)
    else:
        acts = np.stack([rs.randint(0, int(k), size=(T, N)) for k in sp.nvec], -1).astype(np.int32)
    got_r, got_l, got_d = [], [], []
    for t in range(T):
        _, _, done, _ = env.step(torch.as_tensor(acts[t], device="cuda"))
        st = env.episode_stats()
        got_r.append(st["last_return"].cpu().numpy().copy()); got_l.append(st["last_length"].cpu().numpy().copy())
        got_d.append(done.cpu().numpy().astype(bool).copy())
    checked = 0
    for i in range(N):
        e = ol.OracleEnv(prob, rep)
        e.seed(700 + i)
        e.reset()
        out = e.rollout(acts[:, i])
        ret, length = 0.0, 0
        for t in range(T):
            ret += float(out["reward"][t]); length += 1
            assert bool(out["done"][t]) == bool(got_d[t][i])
            if out["done"][t]:
                assert got_r[t][i] == ret and got_l[t][i] == length, (i, t, got_r[t][i], ret, got_l[t][i], length)
                ret, length = 0.0, 0
                checked += 1
    assert checked >= N


@pytest.mark.gpu
def test_vec_env_monitor_and_rollout_collector():
    from gym_pcgrl_amd.rollout import RolloutCollector
    from gym_pcgrl_amd.utils import make_vec_envs
    torch = _torch()
    venv = make_vec_envs("binary-narrow-v0", "narrow", log_dir="unused", n_cpu=64, seed=5)
    assert venv.monitor
    n_act = venv.action_space.n
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    policy = lambda obs: torch.randint(0, n_act, (obs.shape[0],), device=obs.device, generator=g)
    col = RolloutCollector(venv, n_steps=120)
    b = col.collect(policy)
    assert b.obs.shape[:2] == (120, 64) and b.obs.dtype == torch.uint8
    assert bool(b.episode_starts[0].all()) and torch.equal(b.episode_starts[1:], b.dones[:-1])
    # the Monitor sums equal the sums of the collected rewards per episode
    rew, done = b.rewards.cpu().numpy(), b.dones.cpu().numpy()
    ep_r = torch.stack(col.episode_returns).cpu().numpy()
    ep_l = torch.stack(col.episode_lengths).cpu().numpy()
    n_ep = 0
    for i in range(64):
        acc, ln = 0.0, 0
        for t in range(120):
            acc += rew[t, i]; ln += 1
            if done[t, i]:
                assert ep_r[t, i] == acc and ep_l[t, i] == ln
                acc, ln = 0.0, 0
                n_ep += 1
    assert n_ep > 0
    # second rollout continues the same trajectories; the Monitor layer of the VecEnv reports `episode` infos
    b2 = col.collect(policy)
    assert torch.equal(b2.episode_starts[0], torch.as_tensor(done[-1], device="cuda"))
    obs, r, d, infos = venv.step(policy(b2.last_obs))
    lst = infos.to_list()
    for i in np.nonzero(d.cpu().numpy())[0]:
        assert set(lst[i]["episode"]) == {"r", "l"}


@pytest.mark.gpu
@pytest.mark.parametrize("env_name,rep", [("binary-narrow-v0", "narrow"), ("zelda-wide-v0", "wide")])
def test_double_buffered_collector_matches_one_batch(env_name, rep):
    """Two sub-batches of one GPU on two streams (rollout.DoubleBufferedCollector: policy and step of one sub-batch next to those of the
    other) hold the transitions of the one batch with the same seeds -- observations, actions, rewards, dones, episode starts, row by
    row -- under a policy that looks at nothing but an environment's own observation."""
    from gym_pcgrl_amd.rollout import DoubleBufferedCollector, RolloutCollector
    from gym_pcgrl_amd.utils import make_vec_envs
    torch = _torch()
    N, T, seed = 192, 90, 11
    one = make_vec_envs(env_name, rep, log_dir=None, n_cpu=N, seed=seed)
    n_act = one.action_space.n
    policy = lambda obs: (obs.reshape(obs.shape[0], -1).to(torch.int64) * torch.arange(1, obs[0].numel() + 1, device=obs.device)).sum(1) % n_act
    ref = RolloutCollector(one, n_steps=T)
    halves = [make_vec_envs(env_name, rep, log_dir=None, n_cpu=N // 2, seed=seed + k * (N // 2)) for k in range(2)]
    col = DoubleBufferedCollector(halves, n_steps=T)
    assert col.streams[0].cuda_stream != col.streams[1].cuda_stream
    for rollout in range(2):            # (the second one goes on where the first stopped)
        b = ref.collect(policy)
        parts = col.collect(policy)
        torch.cuda.synchronize()
        assert rep != "narrow" or int(b.dones.sum()) > 0          # (episode ends and in-kernel resets are part of what is compared)
        for name in ("obs", "actions", "rewards", "dones", "episode_starts", "last_obs"):
            whole = getattr(b, name)
            got = torch.cat([getattr(q, name) for q in parts], dim=0 if name == "last_obs" else 1)
            assert torch.equal(whole, got), (rollout, name)


# ------------------------------------------------------------------ both Sokoban search implementations on the device
@pytest.mark.gpu
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "stats_sokoban_*.npz"))), ids=os.path.basename)
def test_sokoban_generic_search_path(path, monkeypatch):
    """k_sokoban picks the register-resident search for levels with <= 7 crates, which is every fixture level; the
    generic search (more crates, LDS workspace) must give the same answers: the tuning switch sok_generic routes every level
    through it."""
    _torch()
    _tune(monkeypatch, "sok_generic", "1")
    d = np.load(path)
    maps = d["maps"]
    n, h, w = maps.shape
    env = _make("sokoban", "wide", n, [dict(width=w, height=h), dict(solver_power=int(d["solver_power"]))])
    env.reset()
    env.set_maps(maps)
    got = env.stats.cpu().numpy().astype(np.int64)
    assert env.check_status() == 0
    assert np.array_equal(got, d["stats"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "stats_mdungeon_*.npz"))), ids=os.path.basename)
def test_mdungeon_generic_search_path(path, monkeypatch):
    """k_mdungeon picks the compact search (mdungeon_fast.h) for levels with <= 48 things, which is nearly every
    fixture level; the generic search must give the same answers: the tuning switch sok_generic routes every level through it."""
    _torch()
    _tune(monkeypatch, "sok_generic", "1")
    d = np.load(path)
    maps = d["maps"]
    n, h, w = maps.shape
    env = _make("mdungeon", "wide", n, [dict(width=w, height=h), dict(solver_power=int(d["solver_power"]))])
    env.reset()
    env.set_maps(maps)
    got = env.stats.cpu().numpy().astype(np.int64)
    assert env.check_status() == 0
    assert np.array_equal(got, d["stats"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "stats_ddave_*.npz"))), ids=os.path.basename)
def test_ddave_generic_search_path(path, monkeypatch):
    """k_ddave picks the compact search (ddave_fast.h) for levels with <= 48 diamonds, which is every fixture level; the
    generic search must give the same answers: the tuning switch sok_generic routes every level through it."""
    _torch()
    _tune(monkeypatch, "sok_generic", "1")
    d = np.load(path)
    maps = d["maps"]
    n, h, w = maps.shape
    env = _make("ddave", "wide", n, [dict(width=w, height=h), dict(solver_power=int(d["solver_power"]))])
    env.reset()
    env.set_maps(maps)
    got = env.stats.cpu().numpy().astype(np.int64)
    assert env.check_status() == 0
    assert np.array_equal(got, d["stats"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "stats_smb_*.npz"))), ids=os.path.basename)
def test_smb_search_fallback_path(path, monkeypatch):
    """k_smb runs the balance-1 play-through on a two-label heap of at most 4 095 slots in LDS and repeats a search whose
    queue outgrows that with the general search (lanes 0..3, heap continued in global memory) -- which no level of the
    default size ever needs.  The tuning switch smb_lds_heap = 1024 makes most full-size levels outgrow it: same answers."""
    _torch()
    _tune(monkeypatch, "smb_lds_heap", "1024")
    d = np.load(path)
    maps = d["maps"]
    n, h, w = maps.shape
    env = _make("smb", "wide", n, [dict(width=w, height=h)])
    env._prob._solver_power = int(d["solver_power"])
    env.adjust_param()
    env.reset()
    env.set_maps(maps)
    got = env.stats.cpu().numpy().astype(np.int64)
    assert env.check_status() == 0
    assert np.array_equal(got, d["stats"])


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(130, 14), (250, 8), (60, 32), (122, 20), (123, 5), (7, 3)], ids=lambda s: "%dx%d" % s)
def test_smb_sizes_vs_oracle(size):
    """smb beyond the registered 114 x 14: levels wider than 122 (the engine's grid no longer fits two column registers per
    lane: the general search does the balance-1 play-through too), the widest and the tallest the kernel takes, the
    boundary between the two searches, and the smallest -- 6 environments, 16 steps against the oracle, blocked and open maps."""
    _torch()
    import parity_harness as ph
    w, h = size
    for k, calls in enumerate(([dict(width=w, height=h)],
                               [dict(width=w, height=h), dict(probs={"empty": 0.5, "solid": 0.4, "tube": 0.05})])):
        err = ph.run_config("smb", "narrow" if k == 0 else "wide", calls, 6, 16, 300 + k, np.random.RandomState(11 + k), False)
        assert err is None, err


@pytest.mark.gpu
def test_smb_search_fallback_rollout_vs_oracle(monkeypatch):
    """The same switch on stepping environments (resets inside k_smb, both searches, the overflow path): 24 environments,
    40 steps against the oracle."""
    _torch()
    import parity_harness as ph
    _tune(monkeypatch, "smb_lds_heap", "1024")
    err = ph.run_config("smb", "narrow", [], 24, 40, 77, np.random.RandomState(5), False)
    assert err is None, err


@pytest.mark.gpu
def test_ddave_large_solver_power_vs_oracle():
    """ddave levels with a solver_power beyond the LDS heap (heap and visited table in the global arena) against the
    oracle: open levels with ledges, many diamonds."""
    _torch()
    rs = np.random.RandomState(29)
    h, w, power = 8, 12, 6000
    maps = []
    while len(maps) < 32:
        m = np.zeros((h, w), np.uint8)
        for _k in range(rs.randint(1, 5)):
            y = rs.randint(1, h); x0 = rs.randint(0, w); x1 = rs.randint(x0, w) + 1
            m[y, x0:x1] = 1
        cells = rs.permutation(h * w)
        m.flat[cells[0]] = 2; m.flat[cells[1]] = 3; m.flat[cells[2]] = 5
        k = rs.randint(0, 8)
        m.flat[cells[3:3 + k]] = 4
        m.flat[cells[3 + k:3 + k + rs.randint(0, 4)]] = 6
        maps.append(m)
    maps = np.array(maps)
    env = _make("ddave", "wide", len(maps), [dict(width=w, height=h), dict(solver_power=power)])
    env.reset()
    env.set_maps(maps)
    got = env.stats.cpu().numpy().astype(np.int64)
    exp = np.array([ol.get_stats("ddave", m, solver_power=power) for m in maps])
    assert np.array_equal(got, exp), np.nonzero((got != exp).any(1))[0]
    assert (exp[:, 10] > 0).any() and (exp[:, 10] == 0).any()


@pytest.mark.gpu
def test_mdungeon_crowded_levels_vs_oracle():
    """Levels with more than 48 things (beyond the compact search) and a solver_power beyond the LDS heap (global
    arena) against the oracle."""
    _torch()
    rs = np.random.RandomState(23)
    for (h, w, power, fill) in [(9, 9, 400, 0.8), (8, 10, 6000, 0.25), (7, 11, 6000, 0.75)]:
        maps = []
        while len(maps) < 24:
            m = np.zeros((h, w), np.uint8)
            cells = rs.permutation(h * w)
            m.flat[cells[0]] = 2
            m.flat[cells[1]] = 3
            k = int(fill * h * w)
            m.flat[cells[2:2 + k]] = rs.choice([4, 5, 6, 7], size=k, p=[0.3, 0.3, 0.25, 0.15])
            maps.append(m)
        maps = np.array(maps)
        env = _make("mdungeon", "wide", len(maps), [dict(width=w, height=h), dict(solver_power=power)])
        env.reset()
        env.set_maps(maps)
        got = env.stats.cpu().numpy().astype(np.int64)
        exp = np.array([ol.get_stats("mdungeon", m, solver_power=power) for m in maps])
        assert np.array_equal(got, exp), np.nonzero((got != exp).any(1))[0]
        assert (exp[:, 10] > 0).any() and (exp[:, 2:5].sum(1) > (48 if fill > 0.5 else 0)).all()


@pytest.mark.gpu
def test_sokoban_many_crates_vs_oracle():
    """Levels with 8..12 crates (beyond the register-resident search) against the oracle: engineered 8x8 levels,
    crates next to their targets in open space, small solver_power so that every agent runs into the cap or wins."""
    _torch()
    rs = np.random.RandomState(21)
    maps = []
    while len(maps) < 24:
        m = np.zeros((8, 8), np.uint8)
        k = rs.randint(8, 13)
        cells = rs.permutation(64)
        m.flat[cells[0]] = 2
        m.flat[cells[1:1 + k]] = 3
        m.flat[cells[1 + k:1 + 2 * k]] = 4
        m.flat[cells[1 + 2 * k:1 + 2 * k + rs.randint(0, 6)]] = 1
        maps.append(m)
    maps = np.stack(maps)
    power = 400
    exp = np.stack([ol.get_stats("sokoban", m, solver_power=power) for m in maps])
    ran = sum(1 for m in maps if ol.get_stats("sokoban", m, solver_power=power, with_iters=True)[1][0] > 0)
    assert ran >= 12          # the solver precondition (one region) holds for most of them
    env = _make("sokoban", "wide", len(maps), [dict(width=8, height=8), dict(solver_power=power)])
    env.reset()
    env.set_maps(maps)
    got = env.stats.cpu().numpy().astype(np.int64)
    assert env.check_status() == 0
    assert np.array_equal(got, exp), (got, exp)


@pytest.mark.gpu
def test_render_layout():
    """render(): reference layout (16-px tiles, one-tile border, red cursor frame) with the fallback grey tiles."""
    import gym_pcgrl_amd as gp
    _torch()
    env = gp.make("zelda-narrow-v0")
    env.seed(3)
    obs = env.reset()
    img = np.asarray(env.render("rgb_array"))
    h, w = obs["map"].shape
    assert img.shape == ((h + 2) * 16, (w + 2) * 16, 3) and img.dtype == np.uint8
    grey = [int(i * 255 / 8) for i in range(8)]
    x, y = [int(v) for v in obs["pos"]]
    for yy in range(h):
        for xx in range(w):
            px = img[(yy + 1) * 16 + 8, (xx + 1) * 16 + 8]
            assert tuple(px) == (grey[obs["map"][yy, xx]],) * 3
    assert tuple(img[8, 8]) == (grey[1],) * 3                               # border = solid
    assert tuple(img[(y + 1) * 16, (x + 1) * 16]) == (255, 0, 0)            # cursor frame
    wide = gp.make("binary-wide-v0")
    wide.reset()
    assert (np.asarray(wide.render()) != np.array([255, 0, 0])).any(-1).all()   # no cursor for wide
    # tile pictures instead of the grey fallback: the package's own drawings, or a caller's dict (the reference's `_graphics`)
    from gym_pcgrl_amd.envs import tile_art
    env.set_graphics("drawn")
    img2 = np.asarray(env.render("rgb_array"))
    art = tile_art.make_graphics(env._prob.tiles, 16)
    tiles = env._prob.tiles
    for yy in range(h):
        for xx in range(w):
            if (xx, yy) != (x, y):
                assert np.array_equal(img2[(yy + 1) * 16:(yy + 2) * 16, (xx + 1) * 16:(xx + 2) * 16], art[tiles[obs["map"][yy, xx]]])
    assert np.array_equal(img2[:16, :16], art["solid"]) and tuple(img2[(y + 1) * 16, (x + 1) * 16]) == (255, 0, 0)
    env.set_graphics({t: np.full((16, 16, 3), 10 * i, np.uint8) for i, t in enumerate(tiles)})
    assert np.asarray(env.render())[8, 8, 0] == 10 * tiles.index("solid")
    with pytest.raises(KeyError):
        env.set_graphics({"empty": np.zeros((16, 16, 3), np.uint8)})
    env.set_graphics(None)
    assert np.array_equal(np.asarray(env.render("rgb_array")), img)


@pytest.mark.gpu
@pytest.mark.parametrize("cap", ["0", "1"])
def test_sokoban_hard_list_overflow(cap, monkeypatch):
    """k_sokoban publishes long levels so that their A* agents run on other blocks; when the list is full the block
    that ran the BFS runs them itself.  With the list shrunk to 0 / 1 entries the capped fixture levels take both
    routes at once; the answers must not change."""
    _torch()
    _tune(monkeypatch, "sok_hard_cap", cap)
    d = np.load(os.path.join(G, "stats_sokoban_5x5.npz"))
    maps, n = d["maps"], len(d["maps"])
    env = _make("sokoban", "wide", n, [dict(width=5, height=5), dict(solver_power=int(d["solver_power"]))])
    env.reset()
    for _ in range(2):
        env.set_maps(maps)
        assert np.array_equal(env.stats.cpu().numpy().astype(np.int64), d["stats"])


# ------------------------------------------------------------------ long runs through the incremental statistics routes
SOAK_CASES = [
    ("binary", "narrow", (dict(change_percentage=1.0),), 256, 400),                      # long episodes: long chains of incremental updates
    ("binary", "wide", (dict(width=32, height=16), dict(change_percentage=0.6)), 64, 300),   # widest map of the 32-bit route
    ("binary", "narrow", (dict(width=5, height=5), dict(change_percentage=1.0)), 256, 300),  # tiny components: often no champion
    ("binary", "turtle", (dict(width=33, height=16), dict(change_percentage=0.5)), 48, 300),  # 64-bit rows: full route only
    ("binary", "turtle", (dict(width=64, height=40), dict(change_percentage=0.3), dict(warp=True)), 12, 400),  # tall map: k_stats_wide + incremental
    ("binary", "wide", (dict(width=20, height=30), dict(change_percentage=0.5)), 24, 300),   # tall, 32-bit rows
    ("zelda", "narrow", (dict(width=32, height=16), dict(change_percentage=0.5)), 64, 300),
    ("zelda", "wide", (dict(width=11, height=16), dict(change_percentage=1.0)), 128, 400),
    ("zelda", "turtle", (dict(width=5, height=4), dict(change_percentage=1.0)), 128, 300),
]


@pytest.mark.gpu
@pytest.mark.parametrize("prob,rep,calls,E,T", SOAK_CASES, ids=lambda v: str(v) if isinstance(v, (str, int)) else "cfg")
def test_incremental_routes_soak(prob, rep, calls, E, T):
    """Hundreds of consecutive steps per environment against the oracle: the incremental statistics (binary champion
    cache on 16-row and on tall maps, zelda region count) carry state from step to step, so errors would accumulate."""
    torch = _torch()
    seed0 = 4242
    env = _make(prob, rep, E, calls, seed=seed0)
    obs = env.reset()
    W, H = env._prob._width, env._prob._height
    nt = env.get_num_tiles()
    rs = np.random.RandomState(5)
    if rep == "narrow":
        acts = rs.randint(0, nt + 1, size=(T, E, 1))
    elif rep == "turtle":
        acts = rs.randint(0, nt + 4, size=(T, E, 1))
    else:
        acts = np.stack([rs.randint(0, W, size=(T, E)), rs.randint(0, H, size=(T, E)), rs.randint(0, nt, size=(T, E))], -1)
    acts = acts.astype(np.int32)
    exp = []
    for i in range(E):
        o = ol.OracleEnv(prob, rep)
        for kw in calls:
            o.adjust_param(**kw)
        o.seed(seed0 + i)
        o.reset()
        a3 = np.zeros((T, 3), np.int32)
        a3[:, :acts.shape[2]] = acts[:, i]
        exp.append(o.rollout(a3, want_heat=False))
    keys = list(env._prob.info_keys) + ["iterations", "changes"]
    for t in range(T):
        obs, rew, done, info = env.step(acts[t] if rep == "wide" else acts[t, :, 0])
        assert np.array_equal(done.cpu().numpy(), np.array([x["done"][t] for x in exp])), ("done", t)
        assert np.array_equal(rew.cpu().numpy(), np.array([x["reward"][t] for x in exp])), ("reward", t)
        got_info = np.stack([info[k].cpu().numpy() for k in keys], 1).astype(np.int64)
        e_info = np.stack([x["info"][t] for x in exp])
        assert np.array_equal(got_info, e_info), ("info", t, np.nonzero((got_info != e_info).any(1))[0][:4])
        if t % 50 == 49 or t == T - 1:
            assert np.array_equal(obs["map"].cpu().numpy(), np.stack([x["maps"][t] for x in exp])), ("map", t)
    assert env.check_status() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("prob,rep,calls,E,T", [
    ("zelda", "wide", (dict(width=11, height=16),), 192, 150),
    ("binary", "narrow", (), 192, 200),
    ("binary", "turtle", (dict(change_percentage=0.1),), 128, 200),
], ids=lambda v: str(v) if isinstance(v, (str, int)) else "cfg")
def test_paired_certain_resets(prob, rep, calls, E, T, monkeypatch):
    """With thousands of certain resets per launch a wavefront of k_stats takes two of them (four statistics side by
    side).  pair_min = 1 forces that mode on small batches; the rollout must still equal the oracle's."""
    _tune(monkeypatch, "pair_min", "1")
    _tune(monkeypatch, "no_fused", "1")      # k_stats is the kernel that pairs (binary would take k_step otherwise)
    test_rollout_vs_oracle(prob, rep, calls, E, T)


@pytest.mark.gpu
@pytest.mark.parametrize("grid,E,few,spin", [("2", 96, None, None), ("6", 97, "1000", None), ("64", 160, "0", None), ("2048", 160, None, None),
                                             ("2048", 130, "1000", None), ("2048", 160, None, 1), ("6", 97, None, 1)])
def test_tall_map_resets_split_over_two_blocks(grid, E, few, spin, monkeypatch):
    """Tall binary maps (k_stats_wide): a certain reset is two work items for two blocks -- the statistics of the map the step
    ended on, and the reset with the statistics of the regenerated map -- that talk through DevBufs::wide_sync.  With
    change_percentage = 0.01 on a 20 x 24 map (max_changes 4) episodes end every few steps, for many environments in the same
    step; tiny grids make every block walk through several rounds of halves (the grid is made even: an odd block only ever waits
    for its left neighbour).  The full items of maps with few regions go two to a block (the tuning switch wide_few moves the line)."""
    _tune(monkeypatch, "wide_grid", grid)
    if spin is not None:       # the block with the reset gives up waiting at once and takes the old map's statistics over whenever its partner
        _tune(monkeypatch, "wide_spin", spin)      # has not read the planes yet (the hand-over must not depend on dispatch order)
    if few is not None:        # which full items go two to a block (maps with at most that many regions): all of them / none
        _tune(monkeypatch, "wide_few", few)
    test_rollout_vs_oracle("binary", "turtle", (dict(width=20, height=24), dict(change_percentage=0.01)), E, 60)
    test_rollout_vs_oracle("binary", "wide", (dict(width=40, height=17), dict(change_percentage=0.004)), E, 40)


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [0, 1, 2])
def test_soak_two_launch_binary_pipeline(idx, monkeypatch):
    """The binary soak cases on maps of at most 16 rows through k_update + k_stats (PCGRL_NO_FUSED=1) instead of the fused
    k_step -- with two certain resets per wavefront forced as well (PCGRL_PAIR_MIN=1), which only that pipeline has."""
    _tune(monkeypatch, "no_fused", "1")
    _tune(monkeypatch, "pair_min", "1")
    test_incremental_routes_soak(*SOAK_CASES[idx])


@pytest.mark.gpu
@pytest.mark.parametrize("prob,calls,n", [("zelda", (dict(width=11, height=16),), 300),           # fused: decoded inside k_step
                                          ("binary", (), 130),                                    # fused, binary
                                          ("binary", (dict(width=20, height=30),), 70),           # tall map: k_action_map + the pipeline
                                          ("sokoban", (), 64)], ids=lambda v: v if isinstance(v, str) else "")
def test_step_flat_equals_action_map_plus_step(prob, calls, n):
    """pcgrl_step_flat (ActionMap.step + PcgrlEnv.step in one call; the fused step kernel decodes the flat indices itself) against
    pcgrl_action_map followed by pcgrl_step on a twin batch: every step's reward / done / info and the maps, out-of-range indices
    (clamped and reported) included."""
    torch = _torch()
    import ctypes as C
    from gym_pcgrl_amd import _lib
    a_env, b_env = _make(prob, "wide", n, list(calls), seed=4321), _make(prob, "wide", n, list(calls), seed=4321)
    a_env.reset(); b_env.reset()
    W, H, nt = a_env._prob._width, a_env._prob._height, a_env.get_num_tiles()
    rs = np.random.RandomState(99)
    xa = torch.empty((n, 3), dtype=torch.int32, device=a_env.device)
    xb = torch.empty((n, 3), dtype=torch.int32, device=a_env.device)
    for t in range(60):
        flat = rs.randint(0, W * H * nt, size=n).astype(np.int32)
        if t % 7 == 3:
            flat[rs.randint(0, n, size=3)] = [-5, W * H * nt, W * H * nt + 1000]      # out of range: clamped, PCGRL_STATUS_BAD_ACTION
        f = torch.as_tensor(flat, device=a_env.device)
        _, ra, da, ia = a_env.step_flat(f, xa)
        _lib.check(b_env._lib.pcgrl_action_map(b_env._handle, C.c_void_p(f.data_ptr()), C.c_void_p(xb.data_ptr()), b_env._stream()), "pcgrl_action_map")
        _, rb, db, ib = b_env.step(xb)
        assert torch.equal(ra, rb) and torch.equal(da, db), (prob, t)
        assert torch.equal(a_env._bufs["info"], b_env._bufs["info"]), (prob, t)
        assert torch.equal(a_env._bufs["map"], b_env._bufs["map"]), (prob, t)
    for env in (a_env, b_env):
        with pytest.raises(IndexError):
            env.check_status()


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [0, 2, 7, 8])
@pytest.mark.parametrize("pair", [0, 1])
def test_soak_fused_step_reset_pairs(idx, pair, monkeypatch):
    """k_step with two certain resets per wavefront task from the first one on (step_pair = 1; the default pairs from six a block)
    and never (0): binary and zelda soak cases, short episodes included (5 x 5 / 5 x 4 maps: resets all the time)."""
    _tune(monkeypatch, "step_pair", pair)
    test_incremental_routes_soak(*SOAK_CASES[idx])


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [0, 1, 2])
@pytest.mark.parametrize("switches", [dict(no_touch=1), dict(touch_tight=0), dict(touch_tight=3), dict(step_prio=0), dict(step_prio=0xE4 | (3 << 8)),
                                      dict(no_touch=1, no_inc=1)], ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_soak_fused_step_alternatives(idx, switches, monkeypatch):
    """The binary soak cases through k_step with the round-4 switches set the other way: no binary_touch (every change in or next to
    the champion recomputes), the bound on the other components loose / tight everywhere, no wavefront priorities / other
    priorities -- every alternative must reproduce the oracle step for step."""
    for k, v in switches.items():
        _tune(monkeypatch, k, v)
    test_incremental_routes_soak(*SOAK_CASES[idx])


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,calls,N,T", [
    ("binary-narrow-v0", (), 1000, 150),                              # one launch for the whole tape (k_step); 1000: a partial last block
    ("binary-turtle-v0", (dict(change_percentage=0.1),), 300, 120),
    ("binary-wide-v0", (dict(width=21, height=9),), 257, 100),
    ("zelda-wide-v0", (), 200, 60),                                    # rollout: one launch of k_step; the twin steps through k_update + k_stats
    ("sokoban-narrow-v0", (), 128, 40),
    ("mdungeon-turtle-v0", (dict(width=6, height=6),), 96, 40),
    ("sokoban-wide-v0", (dict(change_percentage=0.6),), 1000, 60),     # persistent search-problem kernel, 16 blocks of 64
    ("sokoban-turtle-v0", (dict(solver_power=300),), 20000, 30),          # 128 environments per block
    ("ddave-narrow-v0", (dict(width=6, height=5), dict(change_percentage=0.8, probs={"empty": 0.7, "solid": 0.1, "player": 0.05,
                                                                                      "exit": 0.05, "key": 0.05})), 300, 60),
    ("mdungeon-wide-v0", (dict(width=6, height=6), dict(change_percentage=0.8, probs={"empty": 0.7, "solid": 0.05, "ogre": 0.08})), 300, 60),
    ("sokoban-narrow-v0", (dict(solver_power=200),), 131072 + 64, 6),    # more than 256 x 512 environments: the sequence-of-steps fallback
    ("binary-narrowcast-v0", (), 200, 50),                                # a representation without the fused kernels
], ids=lambda v: v if isinstance(v, str) else "")
def test_rollout_equals_steps(env_id, calls, N, T):
    """pcgrl_rollout (a tape of actions, one launch where the fused step kernel applies) against the same tape fed to
    pcgrl_step one row at a time on a twin batch: per-step reward / done / info and the complete final state."""
    torch = _torch()
    import gym_pcgrl_amd as gp
    prob, rep = env_id.split("-")[:2]
    a_env = _make(prob, rep, N, list(calls), seed=77)
    b_env = _make(prob, rep, N, list(calls), seed=77)
    a_env.enable_episode_stats(); b_env.enable_episode_stats()
    a_env.reset(); b_env.reset()
    sp = a_env.single_action_space
    g = torch.Generator(device="cuda").manual_seed(5)
    if hasattr(sp, "n"):
        tape = torch.randint(0, int(sp.n), (T, N), generator=g, device="cuda", dtype=torch.int32)
    else:
        tape = torch.stack([torch.randint(0, int(k), (T, N), generator=g, device="cuda", dtype=torch.int32) for k in sp.nvec], -1)
    # first half as one rollout, second half as another (state carries over), against T single steps
    h = T // 2
    r1, d1, i1 = a_env.rollout(tape[:h])
    r2, d2, i2 = a_env.rollout(tape[h:])
    rew = torch.cat([r1, r2]); done = torch.cat([d1, d2]); info = torch.cat([i1.table.view(h, N, 10), i2.table.view(T - h, N, 10)])
    for t in range(T):
        obs, r, d, inf = b_env.step(tape[t])
        assert torch.equal(rew[t], r), ("reward", t)
        assert torch.equal(done[t], d), ("done", t)
        assert torch.equal(info[t], inf.table), ("info", t)
    assert int(done.sum().item()) > 0
    sa, sb = a_env.state_dict(), b_env.state_dict()
    for k in sa:
        if sa[k] is not None:
            assert torch.equal(sa[k], sb[k]), k
    # decoded info of the tape
    key = a_env._prob.info_keys[0]
    assert torch.equal(i2[key], a_env._prob.decode_rows(i2.table, [key])[:, 0] if a_env._prob.packed_rows else i2.table[:, 0])


@pytest.mark.gpu
def test_rollout_many_long_searches_per_block():
    """The persistent search-problem kernel when most environments of a block need a long search in the same step: every
    environment starts from an open Sokoban level whose searches run beyond the small tier (the deferred-job list of a
    block then holds most of its environments), against single steps on a twin batch."""
    torch = _torch()
    N, T = 400, 6
    m = np.zeros((6, 6), np.uint8)
    m[0, 0] = 2; m[2, 2] = 3; m[3, 3] = 3; m[2, 4] = 3; m[5, 5] = 4; m[0, 5] = 4; m[5, 0] = 4
    maps = np.repeat(m[None], N, 0)
    envs = []
    for _ in range(2):
        e = _make("sokoban", "wide", N, [dict(width=6, height=6), dict(solver_power=1500, change_percentage=1.0)], seed=5)
        e.reset()
        e.set_maps(maps)
        envs.append(e)
    g = torch.Generator(device="cuda").manual_seed(3)
    # writes of empty tiles into the empty area: the level changes little, the searches stay long
    tape = torch.stack([torch.randint(3, 6, (T, N), generator=g, device="cuda", dtype=torch.int32),
                        torch.randint(4, 6, (T, N), generator=g, device="cuda", dtype=torch.int32),
                        torch.randint(0, 2, (T, N), generator=g, device="cuda", dtype=torch.int32)], -1)
    rew, done, info = envs[0].rollout(tape)
    for t in range(T):
        obs, r, d, inf = envs[1].step(tape[t])
        assert torch.equal(rew[t], r) and torch.equal(done[t], d) and torch.equal(info.table.view(T, N, 10)[t], inf.table), t
    sa, sb = envs[0].state_dict(), envs[1].state_dict()
    for k in sa:
        if sa[k] is not None:
            assert torch.equal(sa[k], sb[k]), k


class _HeapWord:
    """An entry of the searches' queues: compared by priority only, like Node.__lt__ (sokoban/engine.py:49-50)."""
    __slots__ = ("w",)

    def __init__(self, w):
        self.w = w

    def __lt__(self, other):
        return (self.w >> 16) < (other.w >> 16)


@pytest.mark.gpu
def test_range_reward_table_on_the_device():
    """get_range_reward (helper.py:366-376) is what every reward term of every problem goes through; the kernels evaluate it in
    integers (range_reward_i, csrc/pcgrl_algos.h).  The reference's exhaustive table (tests/golden/range_reward.npz: every band the
    problems use, +-inf bounds, values on both sides of and inside the band) through pcgrl_selftest_range_reward ON THE DEVICE --
    until round 6 the table was held against the oracle and the host build of the header only, the device form through trajectories."""
    import ctypes as C
    torch = _torch()
    from gym_pcgrl_amd import _lib
    L = _lib.load()
    tab = np.load(os.path.join(os.path.dirname(__file__), "golden", "range_reward.npz"))["table"]
    enc = lambda b: 2147483647 if b == np.inf else (-2147483648 if b == -np.inf else int(b))
    rows = np.array([[int(nv), int(ov), enc(lo), enc(hi)] for lo, hi, nv, ov, _ in tab], np.int32)
    assert len(rows) == 3240 and np.isinf(tab[:, :2]).any()
    t_rows = torch.from_numpy(rows).to("cuda:0")
    t_out = torch.full((len(rows),), 12345, dtype=torch.int32, device="cuda:0")
    _lib.check(L.pcgrl_selftest_range_reward(C.c_void_p(t_rows.data_ptr()), len(rows), C.c_void_p(t_out.data_ptr()), None), "pcgrl_selftest_range_reward")
    torch.cuda.synchronize()
    got = t_out.cpu().numpy()
    assert np.array_equal(got.astype(np.float64), tab[:, 4]), np.nonzero(got != tab[:, 4])[0][:10]



@pytest.mark.gpu
@pytest.mark.parametrize("seed,spread,n_ops,p_push", [(0, 1, 30000, 0.75), (1, 2, 30000, 0.7), (2, 3, 60000, 0.62), (3, 40, 60000, 0.6),
                                                      (4, 400, 40000, 0.8), (5, 2, 3000, 0.5)])
def test_heap_server_primitives_match_heapq(seed, spread, n_ops, p_push):
    """The heap server of the two-wavefront searches (sok_duo_append: heappush on lanes; sok_duo_repair: heappop's repair six
    levels a round) against CPython's heapq on a tape of pushes and pops: same pops, same array afterwards, slot for slot.
    Few distinct priorities (ties decide the order), a drifting floor like an A* queue, heaps up to fourteen levels deep, a
    drain at the end."""
    import ctypes as C
    import heapq
    torch = _torch()
    from gym_pcgrl_amd import _lib
    L = _lib.load()
    rs = np.random.RandomState(seed)
    ops, h, pops = [], [], []
    idx = 0
    for i in range(n_ops):
        drain = i > 0.8 * n_ops
        if rs.rand() < (0.3 if drain else p_push) and len(h) < 16384:
            w = (min(0xFFF0, i // 400 + int(rs.randint(spread))) << 16) | (idx & 0xFFFF)
            idx += 1
            heapq.heappush(h, _HeapWord(w))
            ops.append(w)
        else:
            ops.append(0xFFFFFFFF)
            pops.append(heapq.heappop(h).w if h else 0xFFFFFFFF)
    dev = "cuda:0"
    t_ops = torch.from_numpy(np.array(ops, np.uint32).view(np.int32)).to(dev)
    t_pops = torch.zeros((max(len(pops), 1),), dtype=torch.int32, device=dev)
    t_heap = torch.zeros((16384,), dtype=torch.int32, device=dev)
    t_n = torch.zeros((1,), dtype=torch.int32, device=dev)
    _lib.check(L.pcgrl_selftest_heap(C.c_void_p(t_ops.data_ptr()), len(ops), C.c_void_p(t_pops.data_ptr()), C.c_void_p(t_heap.data_ptr()),
                                     C.c_void_p(t_n.data_ptr()), None), "pcgrl_selftest_heap")
    torch.cuda.synchronize()
    n = int(t_n.item())
    assert n == len(h)
    got_pops = t_pops.cpu().numpy().view(np.uint32)[:len(pops)]
    assert np.array_equal(got_pops, np.array(pops, np.uint32)), "first difference at pop %d" % int(np.argmax(got_pops != np.array(pops, np.uint32)))
    assert np.array_equal(t_heap.cpu().numpy().view(np.uint32)[:n], np.array([x.w for x in h], np.uint32))
    assert max(len(pops), 1) > 100


@pytest.mark.gpu
def test_device_seeding_matches_numpy():
    """pcgrl_seed_words: MT19937 init_by_array on the device against numpy's RandomState.seed(list) -- keys of two words
    (what gym's hash_seed gives) and of one word."""
    import ctypes as C
    torch = _torch()
    from gym_pcgrl_amd import _lib, seeding
    N = 300
    env = _make("binary", "narrow", N, seed=123456)
    env.reset()                                           # allocates, seeds through pcgrl_seed_words
    exp = seeding.mt_states_for_seeds([123456 + i for i in range(N)])
    # the reset consumed draws in place, so re-seed and look at the untouched rings
    env.seed(123456)
    torch.cuda.synchronize()
    got = env._bufs["rng_rep"].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, exp)
    assert np.array_equal(env._bufs["rng_prob"].cpu().numpy().view(np.uint32), exp)
    assert int(env._bufs["rng_cursor"].abs().sum().item()) == 0
    words = np.zeros((N, 3), np.uint32)
    words[:, 0] = ((np.arange(N, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(2 ** 32)).astype(np.uint32)
    words[:, 2] = 1
    words[0, 0] = 0                                       # the key [0]
    _lib.check(env._lib.pcgrl_seed_words(env._handle, words.ctypes.data_as(C.c_void_p), 0, N, env._stream()), "pcgrl_seed_words")
    got = env._bufs["rng_rep"].cpu().numpy().view(np.uint32)
    for i in (0, 1, 17, N - 1):
        rs = np.random.RandomState()
        rs.seed([int(words[i, 0])])
        assert np.array_equal(got[i], rs.get_state()[1])


@pytest.mark.gpu
@pytest.mark.parametrize("env_id", ["binary-narrow-v0", "zelda-turtle-v0", "sokoban-wide-v0", "mdungeon-turtle-v0", "ddave-narrow-v0"])
def test_state_dict_round_trip(env_id):
    """Checkpoint / resume of the environment state: a second batch that loads the state_dict of the
    first continues exactly like it."""
    import gym_pcgrl_amd as gp
    torch = _torch()
    N, T = 96, 60
    a = gp.make_batched(env_id, num_envs=N, seed=11)
    b = gp.make_batched(env_id, num_envs=N, seed=999)
    a.reset(); b.reset()
    sp = a.single_action_space
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    def act():
        if hasattr(sp, "n"):
            return torch.randint(0, sp.n, (N,), device="cuda", dtype=torch.int32, generator=g)
        return torch.stack([torch.randint(0, int(k), (N,), device="cuda", dtype=torch.int32, generator=g) for k in sp.nvec], -1).contiguous()
    for _ in range(T):
        a.step(act())
    b.load_state_dict(a.state_dict())
    for _ in range(T):
        x = act()
        oa, ra, da, ia = a.step(x)
        ob, rb, db, ib = b.step(x)
        assert torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ia.table, ib.table)
        for k in oa:
            assert torch.equal(oa[k], ob[k]), k


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """bench.py's multi-process path (rank-sharded environments, barrier, max-over-ranks time, one JSON line from rank 0),
    with both ranks on cuda:0 and gloo instead of RCCL (PCGRL_BENCH_SAME_GPU=1) so that it runs on a one-GPU box."""
    import json, subprocess, sys
    _torch()
    root = os.path.dirname(HERE) if "HERE" in globals() else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCGRL_BENCH_SAME_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                          "--envs", "4096"], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 4096 * 20 / (d["ms_per_step"] * 1e-3 * 20)) / d["value"] < 1e-6


# ------------------------------------------------------------------ switching between the fused and the work-list pipelines
@pytest.mark.gpu
@pytest.mark.parametrize("env_id,calls,T1", [
    ("zelda-wide-v0", (), 1), ("zelda-wide-v0", (), 3), ("zelda-narrow-v0", (dict(change_percentage=0.5),), 7),
    ("binary-narrow-v0", (), 5), ("zelda-turtle-v0", (), 2),
], ids=lambda v: str(v) if isinstance(v, (str, int)) else "")
def test_rollout_then_step_same_handle(env_id, calls, T1):
    """reset(); rollout(T1 steps, T1 odd: the fused kernel for zelda) ; step() x 20 (zelda: k_update -> work lists -> k_stats);
    set_maps(); rollout(); step() ... on ONE handle, against a twin that only ever calls step().  A fused launch must leave
    the work-list parity alone: the lists of the parity a later step uses have to be the cleared ones."""
    torch = _torch()
    prob, rep = env_id.split("-")[:2]
    N = 700
    a_env = _make(prob, rep, N, list(calls), seed=31)
    b_env = _make(prob, rep, N, list(calls), seed=31)
    a_env.enable_episode_stats(); b_env.enable_episode_stats()
    a_env.reset(); b_env.reset()
    sp = a_env.single_action_space
    g = torch.Generator(device="cuda").manual_seed(8)

    def tape(T):
        if hasattr(sp, "n"):
            return torch.randint(0, int(sp.n), (T, N), generator=g, device="cuda", dtype=torch.int32)
        return torch.stack([torch.randint(0, int(k), (T, N), generator=g, device="cuda", dtype=torch.int32) for k in sp.nvec], -1)

    def same_state(tag):
        sa, sb = a_env.state_dict(), b_env.state_dict()
        for k in sa:
            if sa[k] is not None:
                assert torch.equal(sa[k], sb[k]), (tag, k)

    for phase, T in enumerate((T1, 1, T1 + 2)):
        tp = tape(T)
        rew, done, info = a_env.rollout(tp)
        for t in range(T):
            _, r, d, inf = b_env.step(tp[t])
            assert torch.equal(rew[t], r) and torch.equal(done[t], d) and torch.equal(info.table.view(T, N, 10)[t], inf.table), (phase, t)
        same_state("after rollout %d" % phase)
        tp = tape(20)
        for t in range(20):
            _, ra, da, ia = a_env.step(tp[t])
            ra, da, ia = ra.clone(), da.clone(), ia.table.clone()
            _, rb, db, ib = b_env.step(tp[t])
            assert torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ia, ib.table), (phase, "step", t)
        same_state("after steps %d" % phase)
        if phase == 1:
            m = b_env._bufs["map"].clone()
            a_env.set_maps(m); b_env.set_maps(m)
    a_env.check_status()


def test_out_of_range_actions_are_reported():
    """Actions outside the action space are clamped into it and flagged in the sticky status word (the reference raises
    IndexError or writes the bad value: narrow_rep.py:101-103, wide_rep.py:68-69)."""
    torch = _torch()
    env = _make("binary", "narrow", 64)
    env.reset()
    env.step(torch.full((64,), 2, dtype=torch.int32, device="cuda"))
    assert env.check_status() == 0
    a = torch.zeros(64, dtype=torch.int32, device="cuda"); a[5] = 3
    env.step(a)
    with pytest.raises(IndexError):
        env.check_status()
    env.close()
    env = _make("zelda", "wide", 64)
    env.reset()
    a = torch.zeros((64, 3), dtype=torch.int32, device="cuda"); a[7, 0] = 11
    env.step(a)
    with pytest.raises(IndexError):
        env.check_status()


# ------------------------------------------------------------------ random configurations and full batch sizes vs the oracle
@pytest.mark.gpu
@pytest.mark.parametrize("chunk", range(6))
def test_fuzz_slice(chunk):
    """A seeded 60-configuration slice of tools/fuzz_parity.py (six chunks of ten): random problem (all five),
    representation (all six), map size, parameters, seeds; 40 % of the cases as one pcgrl_rollout tape, 15 % as an
    odd-length rollout followed by single steps; every step against the CPU oracle."""
    _torch()
    import parity_harness as ph
    rs = np.random.RandomState(9000 + chunk)
    for _ in range(10):
        desc, err = ph.fuzz_case(rs, None, rollout_share=0.4, mixed_share=0.15, steps_scale=0.6)
        assert err is None, err


@pytest.mark.gpu
@pytest.mark.parametrize("mode,seed,n", [("big", 9100, 8), ("big", 9101, 8), ("goal", 9200, 10), ("goal", 9201, 10)])
def test_fuzz_slice_round4_modes(mode, seed, n):
    """Seeded slices of the two fuzz modes of round 4 (parity_harness.draw_config_extra): sizes beyond the tuned kernels (maps with
    a side of 65..130, search levels of 257..1 200 bordered cells, solver_power 17 000), and goals that are met all the time
    (episodes ending where the update did not see it coming: the draw-cache hand-over of the fused step kernel)."""
    _torch()
    import parity_harness as ph
    rs = np.random.RandomState(seed)
    for _ in range(n):
        desc, err = ph.fuzz_case(rs, None, rollout_share=0.4, mixed_share=0.15, steps_scale=0.6, mode=mode)
        assert err is None, (desc, err)


@pytest.mark.gpu
def test_hypothesis_configurations():
    """property-style search over the configuration space with hypothesis -- (problem, representation,
    width, height, change_percentage, flags, batch size, seed) drawn by its strategies (and shrunk to a minimal failing
    configuration when one fails); every step of the HIP path against the CPU oracle: reward, done, info, cursor, heat map,
    and the map every few steps.  Derandomised: the same examples on every run."""
    _torch()
    import parity_harness as ph
    from hypothesis import HealthCheck, given, settings, strategies as st

    sizes = {"binary": (40, 40), "zelda": (40, 40), "sokoban": (7, 7), "mdungeon": (12, 12), "ddave": (12, 12), "smb": (40, 14)}

    @st.composite
    def configs(draw):
        prob = draw(st.sampled_from(sorted(sizes)))
        rep = draw(st.sampled_from(ph.REPS))
        wmax, hmax = sizes[prob]
        w = draw(st.integers(1, wmax))
        h = draw(st.integers(3 if prob == "smb" else 1, hmax))
        calls = [dict(width=w, height=h), dict(change_percentage=draw(st.sampled_from([0.05, 0.2, 0.5, 1.0])))]
        if prob in ("sokoban", "mdungeon", "ddave"):
            calls.append(dict(solver_power=draw(st.sampled_from([30, 200, 1000]))))
        if rep.startswith("narrow") and draw(st.booleans()):
            calls.append(dict(random_tile=False))
        if rep.startswith("turtle") and draw(st.booleans()):
            calls.append(dict(warp=True))
        if draw(st.integers(0, 4)) == 0:
            calls.append(dict(random_start=False))
        E = draw(st.sampled_from([1, 7, 64, 65, 130]))
        mode = draw(st.sampled_from(["steps", "steps", "rollout", "mixed"]))
        return prob, rep, calls, E, draw(st.integers(1, 10 ** 6)), mode

    @settings(max_examples=30, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
    @given(configs())
    def check(cfg):
        prob, rep, calls, E, seed0, mode = cfg
        T = 24 if prob == "smb" else 60
        err = ph.run_config(prob, rep, calls, E, T, seed0, np.random.RandomState(seed0 % 1000), mode == "rollout", mode == "mixed")
        assert err is None, err

    check()


@pytest.mark.gpu
@pytest.mark.parametrize("use_rollout", [False, True], ids=["steps", "rollout"])
@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5", "M1", "D1", "S1"])
def test_full_size_vs_oracle(name, use_rollout):
    """tools/fullsize_parity.py under the driver: the benchmark configurations at their real batch sizes (paired certain
    resets, every difficulty bucket in use, 512 environments per persistent block), 294 sampled environments compared with
    the oracle at every step (at most 60), stepping and as one pcgrl_rollout tape."""
    _torch()
    import parity_harness as ph
    assert ph.fullsize_case(name, use_rollout, max_steps=60) >= 290


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["C2w", "C3w"])
def test_full_size_wrapped_vs_oracle(name):
    """bench.py's trainer-shaped configurations at 65 536 environments against the oracle: the image the fused step writes and --
    C3w -- pcgrl_step_flat's decode of the ActionMap indices, held against the oracle stepped with the decoded actions and the
    wrappers' transform of its maps (not against pcgrl_action_map + pcgrl_step of the library itself)."""
    _torch()
    import parity_harness as ph
    assert ph.fullsize_wrapped_case(name) >= 290


@pytest.mark.gpu
def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts its own two ranks (torch.distributed.run inside) and
    prints ONE line with n_gpus == 2 -- both ranks on cuda:0 over gloo here (PCGRL_BENCH_SAME_GPU=1: a one-GPU box)."""
    import json, subprocess, sys
    _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCGRL_BENCH_SAME_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--envs", "4096",
                          "--steady-warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 4096 * 20 / (d["ms_per_step"] * 1e-3 * 20)) / d["value"] < 1e-6
    # one rank, same workload: the per-GPU rate of the two-rank line is in the same ballpark (they share one GPU here)
    out1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3", "--envs", "4096",
                           "--steady-warmup", "0", "--no-cpu-baseline", "--no-rollout"], env=env, capture_output=True, text=True, timeout=600)
    d1 = json.loads([l for l in out1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["config"]["envs_per_gpu"] == 4096


@pytest.mark.gpu
def test_gym_make_and_vector_env_under_a_gym_module():
    """With a gym module present (the test shim; gym itself is not on the image): `gym.make('zelda-narrow-v0')` gives a
    gym.Env that a gym.Wrapper built the way the reference's wrappers are (wrappers.py:11,21-24: gym.make(game), find the
    PcgrlEnv by class name, adjust_param) drives through the reference trajectory; PcgrlVectorEnv steps like the batch."""
    import subprocess, sys
    _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gym_shim
gym = gym_shim.install(reference_root=os.path.join(%r, "_no_reference_here"))
import gym_pcgrl_amd                         # in place of `import gym_pcgrl`: registers the ids
gym_pcgrl_amd.register_with_gym()
get_pcgrl_env = lambda env: env if "PcgrlEnv" in str(type(env)) else get_pcgrl_env(env.env)
class Plain(gym.Wrapper):
    def __init__(self, game, **kwargs):
        self.env = gym.make(game)
        get_pcgrl_env(self.env).adjust_param(**kwargs)
        gym.Wrapper.__init__(self, self.env)
d = np.load(os.path.join(%r, "traj_zelda_narrow.npz"))
env = Plain("zelda-narrow-v0")
assert isinstance(env.unwrapped, gym.Env)
get_pcgrl_env(env).seed(int(d["cfg"][4]))
o = env.reset()
assert np.array_equal(o["map"], d["map0"][0])
keys = [str(k) for k in d["info_keys"]]
for t in range(120):
    o, r, dn, info = env.step(int(d["actions"][t, 0, 0]))
    assert r == d["reward"][t, 0] and dn == bool(d["done"][t, 0]), t
    assert [info[k] for k in keys] == list(d["info"][t, 0]), t
    if dn:
        o = env.reset()
    assert np.array_equal(o["map"], d["maps"][t, 0]), t
import torch
from gym_pcgrl_amd.vector import PcgrlVectorEnv
N = 64
v = PcgrlVectorEnv("binary-narrow-v0", num_envs=N, seed=5)
b = gym_pcgrl_amd.make_batched("binary-narrow-v0", num_envs=N, seed=5)
assert v.num_envs == N and v.single_action_space.n == 3 and v.observation_space["map"].shape == (N, 14, 14)
o = v.reset(); ob = b.reset()
assert o["heatmap"].dtype == np.float64 and np.array_equal(o["map"], ob["map"].cpu().numpy())
rs = np.random.RandomState(0)
for t in range(60):
    a = rs.randint(0, 3, N)
    v.step_async(a)
    o, r, dn, infos = v.step_wait()
    ob, rb, db, ib = b.step(a)
    assert np.array_equal(o["map"], ob["map"].cpu().numpy()) and np.array_equal(r, rb.cpu().numpy()) and np.array_equal(dn, db.cpu().numpy())
    assert len(infos) == N and infos[3] == ib.to_list()[3]
v.close()
print("ok")
''' % (root, G, G, G)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-3000:]


@pytest.mark.gpu
def test_integration_md_stub_runs_and_matches_the_package():
    """The ctypes stub of INTEGRATION.md, executed as it stands in the document (library path, N, the seeds' MT19937 states
    and the actions filled in): after create / bind / seed / reset / step its map, reward and done buffers equal those of a
    BatchedPcgrlEnv with the same seeds and actions."""
    torch = _torch()
    from gym_pcgrl_amd import _lib, seeding
    from gym_pcgrl_amd.envs import BatchedPcgrlEnv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    code = md[md.index("import ctypes as C, torch"):]
    code = code[:code.index("```")]
    code = code.replace('C.CDLL("libpcgrl_hip.so")', "C.CDLL(%r)" % _lib.build())
    N, seed0 = 192, 4321
    keys = np.ascontiguousarray(seeding.mt_states_for_seeds([seed0 + i for i in range(N)]), dtype=np.uint32)
    assert keys.shape == (N, 624)
    actions = torch.randint(0, 3, (N,), dtype=torch.int32, device="cuda")
    ns = {"N": N, "keys": keys, "actions": actions}
    exec(code, ns)
    torch.cuda.synchronize()
    order = ns["BUFS"]
    buf = dict(zip(order, ns["keep"]))
    lay = ns["lay"]
    env = BatchedPcgrlEnv(prob="binary", rep="narrow", num_envs=N, seed=seed0)
    env.reset()
    obs, rew, done, info = env.step(actions)
    got_map = buf["map"][:N * 14 * 14].view(N, 14, 14)
    assert torch.equal(got_map, obs["map"])
    assert torch.equal(buf["reward"][:N * 8].view(torch.float64), rew)
    assert torch.equal(buf["done"][:N], env._bufs["done"])
    assert torch.equal(buf["pos"][:N * 2].view(N, 2), obs["pos"])
    assert ns["L"].pcgrl_destroy(ns["h"]) == 0
    env.close()


@pytest.mark.gpu
def test_integration_md_package_surface_runs():
    """The calls INTEGRATION.md section 1 shows (make / make_batched / adjust_param / step / info access / rollout, the two
    composite wrappers, the vector-env adapter) run as written, with the shapes and types the document states."""
    torch = _torch()
    import gym_pcgrl_amd
    env = gym_pcgrl_amd.make("binary-narrow-v0")
    env.seed(42)
    obs = env.reset()
    obs, reward, done, info = env.step(env.action_space.sample())
    assert obs["map"].shape == (14, 14) and isinstance(info, dict) and "path-length" in info
    N = 256
    venv = gym_pcgrl_amd.make_batched("binary-narrow-v0", num_envs=N, seed=0)
    venv.adjust_param(change_percentage=0.2)
    obs = venv.reset()
    assert obs["map"].shape == (N, 14, 14) and obs["map"].dtype == torch.uint8 and obs["pos"].shape == (N, 2) and obs["heatmap"].dtype == torch.int16
    actions = torch.randint(0, 3, (N,), dtype=torch.int32, device="cuda")
    obs, reward, done, info = venv.step(actions)
    assert reward.dtype == torch.float64 and done.dtype == torch.bool and info["path-length"].shape == (N,)
    assert isinstance(info.to_list()[0], dict)
    tape = torch.randint(0, 3, (7, N), dtype=torch.int32, device="cuda")
    reward, done, info = venv.rollout(tape)
    assert reward.shape == (7, N) and done.shape == (7, N)
    from gym_pcgrl_amd.wrappers import CroppedImagePCGRLWrapper, ActionMapImagePCGRLWrapper
    w = CroppedImagePCGRLWrapper("zelda-narrow-v0", 22, num_envs=32)
    img = w.reset()
    assert img.dtype == torch.uint8 and img.shape[0] == 32 and img.dim() == 4
    w2 = ActionMapImagePCGRLWrapper("zelda-wide-v0", num_envs=32)
    img2 = w2.reset()
    assert img2.dtype == torch.uint8 and img2.shape[0] == 32
    from gym_pcgrl_amd.vector import PcgrlVectorEnv
    v = PcgrlVectorEnv("binary-narrow-v0", 16)
    o = v.reset()
    v.step_async(np.zeros(16, np.int64))
    out = v.step_wait()
    assert len(out) == 4 and np.asarray(out[1]).shape == (16,)
    for x in (env, venv, w, w2, v):
        x.close()


# ------------------------------------------------------------------ the node driver: one process, a handle and a stream per device
_NODE_CASES = {
    "binary-narrow": ("binary", "narrow", (), 1001, 40),
    "zelda-wide-11x16": ("zelda", "wide", (dict(width=11, height=16),), 1001, 30),
    "sokoban-narrow": ("sokoban", "narrow", (), 1001, 25),
    "binary-turtle-64x64": ("binary", "turtle", (dict(width=64, height=64),), 67, 30),
}
_NODE_REF = {}


def _node_actions(env, rs, T, N):
    sp = env.single_action_space
    if hasattr(sp, "n"):
        return rs.randint(0, sp.n, size=(T, N)).astype(np.int32)
    return np.stack([rs.randint(0, int(k), size=(T, N)) for k in sp.nvec], -1).astype(np.int32)


def _node_reference(name):
    """The whole batch on one handle: what every sharding has to reproduce bit for bit."""
    torch = _torch()
    if name not in _NODE_REF:
        prob, rep, calls, N, T = _NODE_CASES[name]
        env = _make(prob, rep, N, calls, seed=300)
        o0 = {k: v.clone() for k, v in env.reset().items()}
        acts = _node_actions(env, np.random.RandomState(77), T, N)
        outs = []
        for t in range(T):
            obs, rew, done, info = env.step(acts[t])
            outs.append(({k: v.clone() for k, v in obs.items()}, rew.clone(), done.clone(), info.table.clone()))
        torch.cuda.synchronize()
        env.close()
        _NODE_REF[name] = (acts, o0, outs)
    return _NODE_REF[name]


@pytest.mark.gpu
@pytest.mark.parametrize("G", [2, 4, 8])
@pytest.mark.parametrize("name", sorted(_NODE_CASES))
def test_node_driver_shard_invariance(name, G):
    """MultiGpuPcgrlEnv with G handles on G streams (all on cuda:0 here: a one-GPU box) against one handle that owns the whole
    batch: observation, reward, done and the info table of every step are bitwise the same for G in {2, 4, 8}, uneven shards
    included -- for the four shapes of BASELINE.json's configs."""
    torch = _torch()
    from gym_pcgrl_amd.node import MultiGpuPcgrlEnv
    prob, rep, calls, N, T = _NODE_CASES[name]
    acts, o0, outs = _node_reference(name)
    env = MultiGpuPcgrlEnv(prob=prob, rep=rep, num_envs=N, devices=["cuda:0"] * G, seed=300)
    for kw in calls:
        env.adjust_param(**kw)
    assert [hi - lo for lo, hi in env.ranges] == [N // G + (1 if g < N % G else 0) for g in range(G)] and len(set(env.streams)) == G
    obs = env.reset()
    for k in o0:
        assert torch.equal(obs[k].to("cuda:0"), o0[k]), ("reset", k)
    for t in range(T):
        obs, rew, done, infos = env.step(acts[t])
        ref = outs[t]
        assert torch.equal(rew.to("cuda:0"), ref[1]) and torch.equal(done.to("cuda:0"), ref[2]), ("reward/done", t)
        assert torch.equal(torch.cat([i.table for i in infos]), ref[3]), ("info", t)
        for k in ref[0]:
            assert torch.equal(obs[k].to("cuda:0"), ref[0][k]), (k, t)
    assert env.check_status() == [0] * G
    env.close()


@pytest.mark.gpu
def test_sub_batch_streams_run_side_by_side():
    """Sub-batches of one GPU (MultiGpuPcgrlEnv with the same device twice: the double-buffered rollout) get streams whose kernels
    really overlap -- two streams of one device can share a hardware queue, and then the sub-batches would run one after the other."""
    torch = _torch()
    from gym_pcgrl_amd import node
    for _ in range(3):          # (streams come from torch's pool: a few rounds see different ones)
        ss = node.side_by_side_streams(torch, "cuda:0", 2)
        assert len(ss) == 2 and ss[0].cuda_stream != ss[1].cuda_stream
    env = node.MultiGpuPcgrlEnv(prob="binary", rep="narrow", num_envs=256, devices=["cuda:0"] * 2, seed=1, sync_streams=False)
    # (HIP moves streams between hardware queues as it goes: what is asserted is that asking again gets there)
    ok = False
    for _ in range(6):
        ok = bool(env.streams_overlap())
        if ok:
            break
        env.repick_streams()
    if not ok:          # (HIP's to decide, not the library's: the sub-batches are then stepped one after the other, correctly -- below)
        import warnings
        warnings.warn("no pair of streams on different hardware queues in six picks: sub-batches of this GPU serialise")
    env.reset()
    env.step([torch.zeros(128, dtype=torch.int32, device="cuda") for _ in range(2)])
    env.close()


@pytest.mark.gpu
def test_node_driver_host_gather_rollout_and_presplit_actions():
    """The other forms of the node driver: outputs gathered into one pinned host tensor, actions handed over already split
    per device, and a whole tape as one pcgrl_rollout per handle -- all equal to the single-handle batch."""
    torch = _torch()
    from gym_pcgrl_amd.node import MultiGpuPcgrlEnv
    name = "zelda-wide-11x16"
    prob, rep, calls, N, T = _NODE_CASES[name]
    acts, o0, outs = _node_reference(name)
    env = MultiGpuPcgrlEnv(prob=prob, rep=rep, num_envs=N, devices=["cuda:0"] * 3, seed=300, gather="host")
    for kw in calls:
        env.adjust_param(**kw)
    obs = env.reset()
    assert obs["map"].device.type == "cpu" and obs["map"].is_pinned() and torch.equal(obs["map"], o0["map"].cpu())
    for t in range(6):
        parts = [torch.as_tensor(acts[t, lo:hi], device="cuda:0") for lo, hi in env.ranges]
        obs, rew, done, infos = env.step(parts)
        assert torch.equal(rew, outs[t][1].cpu()) and torch.equal(done, outs[t][2].cpu()) and torch.equal(obs["map"], outs[t][0]["map"].cpu())
    rew, done, infos = env.rollout(acts[6:T])
    assert tuple(rew.shape) == (T - 6, N) and torch.equal(rew, torch.stack([o[1] for o in outs[6:]]).cpu())
    assert torch.equal(done, torch.stack([o[2] for o in outs[6:]]).cpu())
    env.synchronize()
    last = torch.cat([sh._bufs["map"] for sh in env.shards])
    assert torch.equal(last, outs[-1][0]["map"])
    env.close()


@pytest.mark.gpu
def test_bench_eight_ranks_tall_maps_on_one_gpu():
    """`bench.py --gpus 8 --workload C5` (BASELINE.json's sharded config: binary-turtle 64x64 on eight GPUs) with its own launcher;
    all eight ranks on cuda:0 over gloo here (PCGRL_BENCH_SAME_GPU=1), a small batch per rank."""
    import json, subprocess, sys
    _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCGRL_BENCH_SAME_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "C5", "--steps", "5", "--warmup", "2", "--envs", "256",
                          "--steady-warmup", "0", "--no-rollout"], env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["width"] == 64 and d["config"]["height"] == 64 and d["config"]["max_changes"] == 39
    assert abs(d["value"] - 8 * 256 * 5 / (d["ms_per_step"] * 1e-3 * 5)) / d["value"] < 1e-6


@pytest.mark.gpu
def test_bench_two_ranks_carry_the_tall_map_config_on_the_headline_line():
    """under N > 1 the default line (C2 headline) also carries `configs.C5` -- BASELINE config 5, 8 192 tall-map
    environments per rank, first window and steady state, max-over-ranks -- so that the day an 8-GPU node runs it the one line tells
    both stories.  Two ranks on cuda:0 over gloo here."""
    import json, subprocess, sys
    _torch()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCGRL_BENCH_SAME_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--steady-warmup", "0", "--no-rollout",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["envs_per_gpu"] == 65536
    c5 = d["configs"]["C5"]
    assert c5["n_gpus"] == 2 and c5["envs_per_gpu"] == 8192 and c5["dominant_kernel"] == "k_stats_wide"
    assert abs(c5["value"] - 2 * 8192 * c5["steps"] / (c5["ms_per_step"] * 1e-3 * c5["steps"])) / c5["value"] < 1e-6
    assert c5["steady_state"]["after_steps"] >= 800 and c5["steady_state"]["gpu_ms_per_step"] > 0


@pytest.mark.gpu
def test_observe_into_the_bound_tensor_with_another_window_drops_the_in_place_state():
    """pcgrl_observe() into the bound tensor with a different geometry must not mark it as the image of the current
    state -- the next step of the wide representation would patch one 16-byte piece of an image laid out differently."""
    torch = _torch()
    import ctypes as C
    from gym_pcgrl_amd import _lib
    n = 96
    env = _make("zelda", "wide", n, [dict(width=11, height=16)], seed=5)
    img = env.bind_observation(16, 11, 0, 0, 1)                  # the map itself, one-hot: updated in place by k_step
    env.reset()
    rs = np.random.RandomState(3)
    draw = lambda: torch.as_tensor(np.stack([rs.randint(0, 11, n), rs.randint(0, 16, n), rs.randint(0, 8, n)], -1).astype(np.int32), device="cuda")
    env.step(draw())
    # the caller scribbles another window (8 x 22 cells, same byte count) into the same tensor through pcgrl_observe
    _lib.check(env._lib.pcgrl_observe(env._handle, C.c_void_p(img.data_ptr()), 8, 22, 0, 0, 1, env._stream()), "pcgrl_observe")
    env.step(draw())
    torch.cuda.synchronize()
    exp = _expected_image(env._bufs["map"].cpu().numpy(), np.zeros((n, 2), np.uint8), 16, 11, 0, 0, 8)
    assert np.array_equal(img.cpu().numpy(), exp)
    env.close()


# ------------------------------------------------------------------ round 4: the reference's behaviours the build used to refuse
@pytest.mark.gpu
def test_adjust_param_size_without_reset_steps_the_old_maps():
    """pcgrl_env.py:106-115 + representation.py:40-45: adjust_param(width, height) changes the problem's size at once but only
    reset() makes maps of the new size -- the reference goes on stepping the old maps, with the new size in the problem's
    formulas (zelda's nearest-enemy default W * H, zelda_prob.py:99) and in max_iterations (Q9).  The oracle models exactly that."""
    torch = _torch()
    E, T = 32, 30
    env = _make("zelda", "wide", E, [dict(change_percentage=0.9)], seed=50)
    orc = []
    for i in range(E):
        o = ol.OracleEnv("zelda", "wide")
        o.adjust_param(change_percentage=0.9)
        o.seed(50 + i)
        o.reset()
        orc.append(o)
    env.reset()
    rs = np.random.RandomState(3)

    live = np.ones(E, bool)

    def steps(n, W, H, stale=False):
        for _ in range(n):
            a = np.stack([rs.randint(0, W, E), rs.randint(0, H, E), rs.randint(0, 8, E)], -1).astype(np.int32)
            obs, rew, done, info = env.step(a)
            torch.cuda.synchronize()
            for i, o in enumerate(orc):
                if not live[i]:
                    continue
                eo, er, ed, einf = o.step(a[i])
                assert er == rew[i].item() and ed == bool(done[i].item()), (i, er, rew[i].item())
                assert einf["nearest-enemy"] == info["nearest-enemy"][i].item() and einf["path-length"] == info["path-length"][i].item()
                if ed and stale:
                    # the reference's reset of this one environment would make a map of the NEW size; a batch has one shape, so the
                    # in-kernel reset regenerates at the old size until reset() switches the whole batch: not compared any further
                    live[i] = False
                    continue
                if ed:
                    eo = o.reset()
                assert np.array_equal(eo["map"], obs["map"][i].cpu().numpy()), i

    steps(T, 11, 7)
    for o in orc:
        o.adjust_param(width=9, height=12)
    env.adjust_param(width=9, height=12)            # no reset(): the 11 x 7 maps go on
    assert env.single_observation_space["map"].shape == (12, 9) and tuple(env._bufs["map"].shape) == (E, 7, 11)
    steps(T, 9, 7, stale=True)                       # (actions inside both the old map and the new action space)
    assert live.sum() >= E // 2
    obs = env.reset()                                # now the maps are 9 x 12
    assert tuple(obs["map"].shape) == (E, 12, 9)
    for i, o in enumerate(orc):
        if live[i]:
            assert np.array_equal(o.reset()["map"], obs["map"][i].cpu().numpy())
    steps(10, 9, 12)
    env.close()


@pytest.mark.gpu
def test_strict_actions_raise_at_the_offending_call():
    """wide_rep.py:68-69: the reference raises IndexError on an x / y outside the map.  strict_actions=True does the same at the
    call (one synchronisation per step); the default clamps and reports through check_status()."""
    _torch()
    env = _make("binary", "wide", 16, seed=1)
    env.strict_actions = True
    env.reset()
    ok = np.zeros((16, 3), np.int32)
    env.step(ok)
    bad = ok.copy(); bad[5, 0] = 14
    with pytest.raises(IndexError):
        env.step(bad)
    env.step(ok)                                     # the next call is judged on its own actions
    with pytest.raises(IndexError):
        env.rollout(np.stack([ok, bad]))
    env.step(ok)
    env.close()
    lax = _make("binary", "wide", 16, seed=1)
    lax.reset()
    lax.step(bad); lax.step(ok)
    with pytest.raises(IndexError):
        lax.check_status()
    lax.close()


@pytest.mark.gpu
def test_solver_power_beyond_the_allocated_arena_reallocates():
    """sokoban_prob.py:60-73: solver_power is an ordinary adjust_param key.  One that the handle's arena cannot take (more than it
    was allocated for, or beyond 16 383: the general searches) makes the next reset() re-allocate instead of failing."""
    torch = _torch()
    d = np.load(os.path.join(G, "stats_sokoban_8x8_p20000.npz"))
    maps = d["maps"]
    env = _make("sokoban", "wide", len(maps), [dict(width=8, height=8)], seed=3)
    env.reset()
    env.adjust_param(solver_power=20000)
    with pytest.raises(RuntimeError):
        env.step(np.zeros((len(maps), 3), np.int32))
    env.reset()
    env.set_maps(maps)
    assert np.array_equal(env.stats.cpu().numpy().astype(np.int64), d["stats"]) and env.check_status() == 0
    env.adjust_param(solver_power=300)               # shrinking stays in place
    env.set_maps(maps)
    exp = np.stack([ol.get_stats("sokoban", m, solver_power=300) for m in maps])
    assert np.array_equal(env.stats.cpu().numpy().astype(np.int64), exp)
    env.close()
