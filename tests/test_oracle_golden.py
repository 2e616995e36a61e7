"""Pin the CPU oracle (oracle/pcgrl_oracle.c) to the reference: every fixture under tests/golden/
was produced by the unmodified reference (tests/golden/make_golden.py); the oracle must reproduce
all of them exactly.  CPU only."""
import ast
import glob
import os

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import OracleEnv, _p, lib

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


# ------------------------------------------------------------------ RNG
def _key_for(seed_u64):
    return ol.seed_key(int(seed_u64))


def test_rng_key_and_draws():
    d = load("rng")
    L = lib()
    for si, s in enumerate(d["seeds"]):
        key = _key_for(s)
        out = np.zeros(624, np.uint32)
        L.orc_rng_key(_p(key), len(key), _p(out))
        assert np.array_equal(out, d["mt_key"][si])
        for bi, n in enumerate(d["bounds"]):
            r = np.zeros(64, np.int64)
            L.orc_rng_randint(_p(key), len(key), int(n), 64, _p(r))
            assert np.array_equal(r, d["randint"][si, bi]), (s, n)
        f = np.zeros(700, np.float64)
        L.orc_rng_random(_p(key), len(key), 700, _p(f))
        assert np.array_equal(f, d["random"][si])
        m = np.zeros((14, 14), np.uint8)
        p2 = np.ascontiguousarray(d["choice2_p"])
        L.orc_rng_choice(_p(key), len(key), _p(p2), 2, 14, 14, _p(m))
        assert np.array_equal(m, d["choice2"][si])
        m = np.zeros((16, 11), np.uint8)
        p8 = np.ascontiguousarray(d["choice8_p"])
        L.orc_rng_choice(_p(key), len(key), _p(p8), 8, 11, 16, _p(m))
        assert np.array_equal(m, d["choice8"][si])


def test_rng_interleaved_stream():
    d = load("rng")
    L = lib()
    for si, s in enumerate(d["seeds"]):
        key = _key_for(s)
        m = np.zeros((5, 5), np.uint8)
        # the choice map consumes 50 draws; replay it through the mixed interface as 25 random()
        ops = [(1, 0)] * 25 + [(0, 14), (0, 14), (1, 0), (0, 5), (0, 5), (0, 5)]
        ops = np.asarray(ops, np.int32)
        ints = np.zeros(8, np.int64)
        fl = np.zeros(32, np.float64)
        L.orc_rng_mixed(_p(key), len(key), _p(ops), len(ops), _p(ints), _p(fl))
        tiles = (fl[:25] >= 0.5).astype(np.int64)
        assert np.array_equal(tiles, d["mixed_ints"][si][:25])
        assert np.array_equal(ints[:2], d["mixed_ints"][si][25:27])
        assert fl[25] == d["mixed_float"][si]
        assert np.array_equal(ints[2:5], d["mixed_ints"][si][27:30])
        if "nodraw_ints" in d.files:
            ops = np.asarray([(0, 1), (0, 14), (0, 1), (0, 5), (0, 1), (0, 64)], np.int32)
            ints = np.zeros(8, np.int64)
            L.orc_rng_mixed(_p(key), len(key), _p(ops), len(ops), _p(ints), _p(fl))
            assert np.array_equal(ints[:6], d["nodraw_ints"][si])


def test_cdf_matches_numpy():
    rs = np.random.RandomState(3)
    for n in (2, 5, 8):
        for _ in range(50):
            p = rs.random_sample(n) + 1e-3
            total = 0.0
            for v in p:
                total += v
            q = np.array([v / total for v in p])
            cdf = q.cumsum()
            cdf /= cdf[-1]
            out = np.zeros(n)
            lib().orc_build_cdf_export(_p(np.ascontiguousarray(p)), n, _p(out))
            assert np.array_equal(out, cdf)


# ------------------------------------------------------------------ range reward
def test_range_reward_table():
    t = load("range_reward")["table"]
    for lo, hi, nv, ov, r in t:
        assert lib().orc_range_reward(nv, ov, lo, hi) == r


# ------------------------------------------------------------------ stats KATs
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "stats_*.npz"))), ids=os.path.basename)
def test_stats_kat(path):
    d = np.load(path)
    prob = os.path.basename(path).split("_")[1]
    power = int(d["solver_power"]) if "solver_power" in d.files else 5000
    for i, m in enumerate(d["maps"]):
        got, it = ol.get_stats(prob, m, solver_power=power, with_iters=True)
        assert np.array_equal(got, d["stats"][i]), (i, got, d["stats"][i], m)
        if prob in ("sokoban", "mdungeon", "ddave") and d["agents"][i, 4] > -2:
            assert np.array_equal(it, d["agents"][i, :4]), (i, it, d["agents"][i])


# ------------------------------------------------------------------ adjust_param
def test_adjust_param_table():
    d = load("adjust_param")
    for case, row in zip(d["cases"], d["rows"]):
        prob, rep, calls = ast.literal_eval(str(case))
        e = OracleEnv(prob, rep)
        for kw in calls:
            e.adjust_param(**kw)
        assert [e.width, e.height, e.max_changes, e.max_iterations, e.num_tiles] == list(row[:5]), case


# ------------------------------------------------------------------ trajectories
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "traj_*.npz"))), ids=os.path.basename)
def test_trajectory(path):
    d = np.load(path)
    prob, rep = str(d["prob"]), str(d["rep"])
    calls = ast.literal_eval(str(d["calls"]))
    W, H, max_changes, max_iter, seed0, _ = [int(v) for v in d["cfg"]]
    acts = d["actions"]
    T, E = acts.shape[:2]
    for i in range(E):
        e = OracleEnv(prob, rep)
        for kw in calls:
            e.adjust_param(**kw)
        e.seed(seed0 + i)
        assert (e.max_changes, e.max_iterations) == (max_changes, max_iter)
        o = e.reset()
        assert np.array_equal(o["map"], d["map0"][i])
        if rep != "wide":
            assert np.array_equal(o["pos"], d["pos0"][i])
        out = e.rollout(acts[:, i])
        bad = np.nonzero((out["maps"] != d["maps"][:, i]).reshape(T, -1).any(1))[0]
        assert bad.size == 0, ("map mismatch first at step", bad[:3])
        if rep != "wide":
            assert np.array_equal(out["pos"].astype(np.uint8), d["pos"][:, i])
        assert np.array_equal(out["heatmap"], d["heatmap"][:, i])
        assert np.array_equal(out["done"], d["done"][:, i])
        assert np.array_equal(out["info"], d["info"][:, i]), np.nonzero((out["info"] != d["info"][:, i]).any(1))[0][:3]
        assert np.array_equal(out["reward"], d["reward"][:, i])


def test_oracle_under_sanitizers():
    """the oracle built with AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile `sanitize`) replays
    one statistics file and one trajectory of every problem -- same answers, no sanitizer report."""
    import shutil
    import subprocess
    import sys
    if os.environ.get("PCGRL_ORACLE_SO"):
        pytest.skip("already running on a substituted oracle build")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip() if shutil.which("gcc") else ""
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan on this host")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "sanitize"])
    picks = []
    for prob in ("binary", "zelda", "sokoban", "mdungeon", "ddave", "smb"):
        picks.append(sorted(glob.glob(os.path.join(G, "stats_%s_*.npz" % prob)))[0])
        picks.append(sorted(glob.glob(os.path.join(G, "traj_%s_*.npz" % prob)))[0])
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_oracle_golden as t\n"
            "for p in %r:\n"
            "    (t.test_stats_kat if '/stats_' in p else t.test_trajectory)(p)\n"
            "print('replayed', %d)\n") % (os.path.join(root, "tests"), root, picks, len(picks))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", PCGRL_ORACLE_SO=os.path.join(root, "oracle", "_san", "libpcgrl_oracle.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "replayed %d" % len(picks) in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_heat_map_beyond_int16_matches_reference():
    """heat_boundary.npz (tests/golden/make_golden.py gen_heat_boundary): the reference's float64 heat map (pcgrl_env.py:35,137) counts
    one cell of a 182 x 182 binary-wide map past 32 767; the oracle's uint16 must hold the same counts."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "heat_boundary.npz"))
    W, H, max_changes, max_iter, seed, T = [int(v) for v in d["cfg"]]
    x, y, _ = [int(v) for v in d["cell"]]
    o = ol.OracleEnv("binary", "wide")
    o.adjust_param(width=W, height=H, probs={"empty": 0.0, "solid": 1.0})
    o.adjust_param(change_percentage=1.0)
    o.seed(seed)
    o.reset()
    assert (o.max_changes, o.max_iterations) == (max_changes, max_iter)
    acts = np.zeros((T, 3), np.int32)
    acts[:, 0], acts[:, 1], acts[:, 2] = x, y, np.arange(T) % 2
    r = o.rollout(acts, want_maps=False, want_heat=False)
    steps = d["steps"]
    assert np.array_equal(r["reward"][steps], d["reward"]) and np.array_equal(r["done"][steps], d["done"]) and np.array_equal(r["info"][steps], d["info"])
    heat = o.obs()["heatmap"]
    got = np.argwhere(heat != 0)
    assert np.array_equal(got, d["heat_cells"]) and np.array_equal(heat[got[:, 0], got[:, 1]].astype(np.int64), d["heat_counts"])
    assert int(d["heat_counts"].max()) == T > 32767
    assert np.array_equal(np.argwhere(o.obs()["map"] == 0), d["empty_cells"])
