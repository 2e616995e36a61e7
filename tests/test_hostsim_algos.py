"""The bitboard templates of gym_pcgrl_amd/csrc (pcgrl_algos.h, mt19937.h) instantiated on a CPU
lane-group simulator (tests/hostsim/sim_algos.cpp) and checked against the golden fixtures and the
oracle.  This covers the algorithm logic on CPU; the DPP/ballot backend itself is covered by the
-m gpu tests.  CPU only."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import _p

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
SRC = os.path.join(HERE, "hostsim", "sim_algos.cpp")
SO = os.path.join(HERE, "hostsim", "libsim_algos.so")
CSRC = os.path.join(os.path.dirname(HERE), "gym_pcgrl_amd", "csrc")


@pytest.fixture(scope="module")
def sim():
    deps = [SRC] + glob.glob(os.path.join(CSRC, "*.h"))
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", SO, SRC])
    L = C.CDLL(SO)
    L.sim_stats.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.sim_range_reward.argtypes = [C.c_double] * 4
    L.sim_range_reward.restype = C.c_double
    L.sim_mt_randint.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sim_mt_random.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.sim_mt_mapgen.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def sim_stats(L, prob, m, variant=0):
    m = np.ascontiguousarray(m, np.uint8)
    h, w = m.shape
    out = np.zeros(8, np.int32)
    need = C.c_int()
    L.sim_stats(ol.PROBS[prob], _p(m), h, w, w, h, variant, _p(out), C.byref(need))
    return out, need.value


def _dims(path):
    """(h, w, solver_power) of a stats fixture, from its name: stats_<prob>_<h>x<w>[_p<power>].npz"""
    parts = os.path.basename(path)[:-4].split("_")
    h, w = (int(v) for v in parts[2].split("x"))
    return h, w, (int(parts[3][1:]) if len(parts) > 3 else 5000)


def _row_bitboards(path):          # what the lane-group kernels take: maps of at most 64 x 64 (larger ones: csrc/bigmap.h, GPU tests)
    h, w, _ = _dims(path)
    return h <= 64 and w <= 64


def _compact_search(path):         # what the compact searches take (larger levels / solver_power: csrc/search_big.h, GPU tests)
    h, w, power = _dims(path)
    return (h + 2) * (w + 2) <= 256 and power <= 16383


# (smb has no row-bitboard program: its statistics are plain loops over the byte map in kernels_smb.h, covered by the GPU tests)
@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(G, "stats_*.npz")) if "stats_smb_" not in p and _row_bitboards(p)), ids=os.path.basename)
def test_bitboard_stats_vs_golden(sim, path):
    d = np.load(path)
    prob = os.path.basename(path).split("_")[1]
    ns = ol.NSTATS[prob]
    for i, m in enumerate(d["maps"]):
        for variant in ((0,) if i % 7 else (0, 1, 2, 3)):
            out, need = sim_stats(sim, prob, m, variant)
            exp = d["stats"][i]
            if prob == "ddave":
                ran = d["agents"][i, 4] > -2
                assert bool(need) == bool(ran), i
                # packed row: player | exit << 8 | key << 16, dist-floor, diamonds, spikes, regions
                assert list(out[:5]) == [exp[0] | exp[2] << 8 | exp[4] << 16, exp[1], exp[3], exp[5], exp[6]], (i, out, exp, m)
                if not ran:
                    assert (out[5], out[6], out[7]) == (0, exp[9], 0) and exp[7] == exp[8] == exp[10] == 0, (i, out, exp)
            elif prob == "mdungeon":
                ran = d["agents"][i, 4] > -2
                assert bool(need) == bool(ran), i
                assert np.array_equal(out[:6], exp[:6]), (i, out, exp)
                if not ran:
                    assert (out[6], out[7]) == (exp[9], 0) and not exp[6:9].any() and exp[10] == 0, (i, out, exp)
            elif prob == "sokoban":
                ran = d["agents"][i, 4] > -2
                assert bool(need) == bool(ran), i
                assert np.array_equal(out[:4], exp[:4]), (i, out, exp)
                if not ran:
                    assert np.array_equal(out[:ns], exp), (i, out, exp)
            else:
                assert np.array_equal(out[:ns], exp), (i, variant, out, exp, m)


@pytest.mark.parametrize("fast", [0, 1], ids=["generic", "register-resident"])
def test_device_sokoban_solver_vs_golden(sim, fast):
    """gym_pcgrl_amd/csrc/sokoban_solver.h and sokoban_fast.h (the code k_sokoban runs) compiled for the host.
    Without the exhausted-BFS shortcut the per-agent iteration counts must equal the reference's; with it (what
    the GPU runs) dist-win and sol-length must still be equal, and the A* agents are skipped exactly when BFS
    exhausted the state space without a win."""
    sim.sim_sokoban_solve2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    n = skipped = 0
    for path in sorted(p for p in glob.glob(os.path.join(G, "stats_sokoban_*.npz")) if _compact_search(p)):
        d = np.load(path)
        power = int(d["solver_power"])
        for i, m in enumerate(d["maps"]):
            if d["agents"][i, 4] == -2:
                continue
            m = np.ascontiguousarray(m)
            dist, sol = C.c_int(), C.c_int()
            it = np.zeros(4, np.int32)
            assert sim.sim_sokoban_solve2(_p(m), m.shape[0], m.shape[1], power, 0, fast, C.byref(dist), C.byref(sol), _p(it)) == 0
            assert (dist.value, sol.value) == (d["stats"][i, 4], d["stats"][i, 5]), (path, i)
            assert np.array_equal(it, d["agents"][i, :4]), (path, i, it, d["agents"][i])
            it2 = np.zeros(4, np.int32)
            assert sim.sim_sokoban_solve2(_p(m), m.shape[0], m.shape[1], power, 1, fast, C.byref(dist), C.byref(sol), _p(it2)) == 0
            assert (dist.value, sol.value) == (d["stats"][i, 4], d["stats"][i, 5]), ("shortcut", path, i)
            if it2[1] == 0 and it[1] > 0:
                skipped += 1
                # the argument behind the shortcut, checked on the reference's own counts: every agent popped
                # exactly as many entries as BFS and none of them won
                assert d["agents"][i, 4] == -1 and (d["agents"][i, :4] == d["agents"][i, 0]).all(), (path, i, d["agents"][i])
            n += 1
    assert n > 300 and skipped > 100


@pytest.mark.parametrize("fast", [0, 1], ids=["generic", "compact"])
def test_device_mdungeon_solver_vs_golden(sim, fast):
    """gym_pcgrl_amd/csrc/mdungeon_solver.h (the code k_mdungeon runs) compiled for the host, against the reference's
    planner results.  Without the exhausted-search shortcut the per-agent iteration counts must equal the reference's;
    with it the five results must still be equal, and agents are skipped only after an agent exhausted the state
    space -- in which case the reference's own counts show that every agent popped the same number of entries."""
    sim.sim_mdungeon_solve2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    n = skipped = capped = took_fast = 0
    for path in sorted(p for p in glob.glob(os.path.join(G, "stats_mdungeon_*.npz")) if _compact_search(p)):
        d = np.load(path)
        power = int(d["solver_power"])
        for i, m in enumerate(d["maps"]):
            if d["agents"][i, 4] == -2:
                continue
            m = np.ascontiguousarray(m)
            exp = [d["stats"][i, 9], d["stats"][i, 10], d["stats"][i, 6], d["stats"][i, 7], d["stats"][i, 8]]
            out, it = np.zeros(5, np.int32), np.zeros(4, np.int32)
            rc = sim.sim_mdungeon_solve2(_p(m), m.shape[0], m.shape[1], power, 0, fast, _p(out), _p(it))
            assert rc in (0, 1)
            took_fast += rc
            assert list(out) == exp, (path, i, out, exp)
            assert np.array_equal(it, d["agents"][i, :4]), (path, i, it, d["agents"][i])
            out2, it2 = np.zeros(5, np.int32), np.zeros(4, np.int32)
            assert sim.sim_mdungeon_solve2(_p(m), m.shape[0], m.shape[1], power, 1, fast, _p(out2), _p(it2)) in (0, 1)
            assert list(out2) == exp, ("shortcut", path, i, out2, exp)
            if not np.array_equal(it, it2):
                skipped += 1
                a = d["agents"][i]
                assert a[4] == -1 and a[0] < power and (a[:4] == a[0]).all(), (path, i, a)
            capped += int((it >= power).any())
            n += 1
    assert n > 400 and skipped > 5 and capped > 20
    assert (took_fast > 0.9 * n) if fast else took_fast == 0


@pytest.mark.parametrize("fast", [0, 1], ids=["generic", "compact"])
def test_device_ddave_solver_vs_golden(sim, fast):
    """gym_pcgrl_amd/csrc/ddave_solver.h (the code k_ddave runs) compiled for the host, against the reference's planner
    results and per-agent iteration counts."""
    sim.sim_ddave_solve2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    n = capped = took_fast = 0
    for path in sorted(p for p in glob.glob(os.path.join(G, "stats_ddave_*.npz")) if _compact_search(p)):
        d = np.load(path)
        power = int(d["solver_power"])
        for i, m in enumerate(d["maps"]):
            if d["agents"][i, 4] == -2:
                continue
            m = np.ascontiguousarray(m)
            exp = [d["stats"][i, 9], d["stats"][i, 10], d["stats"][i, 7], d["stats"][i, 8]]
            out, it = np.zeros(4, np.int32), np.zeros(4, np.int32)
            rc = sim.sim_ddave_solve2(_p(m), m.shape[0], m.shape[1], power, fast, _p(out), _p(it))
            assert rc in (0, 1)
            took_fast += rc
            assert list(out) == exp, (path, i, out, exp)
            assert np.array_equal(it, d["agents"][i, :4]), (path, i, it, d["agents"][i])
            capped += int((it >= power).any())
            n += 1
    assert n > 400 and capped > 30
    assert (took_fast > 0.9 * n) if fast else took_fast == 0


@pytest.mark.parametrize("chunk", [1, 37, 1000])
def test_compact_searches_suspend_and_resume(sim, chunk):
    """SokResume (csrc/sokoban_fast.h; round 5, pcgrl_step_async): the compact searches of the three search problems run in
    pieces of `chunk` pops -- suspended in front of a pop, continued by the next call from the saved scalars with pool, heap
    and visited table left as they were -- must give the reference's results AND its per-agent iteration counts, exactly like
    the search in one piece (the one-wavefront loops, which is what the host simulator can run; the two-wavefront form of the
    GPU is held against the oracle by tests/test_gpu_async.py)."""
    sim.sim_sokoban_solve2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    sim.sim_mdungeon_solve2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    sim.sim_ddave_solve2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    sim.sim_pieces_reset.restype = C.c_long
    sim.sim_set_chunk(chunk)
    sim.sim_pieces_reset()
    stride = 1 if chunk > 1 else 7          # (one pop a piece is slow: every seventh map)
    n = agents = 0
    try:
        for prob in ("sokoban", "mdungeon", "ddave"):
            for path in sorted(p for p in glob.glob(os.path.join(G, "stats_%s_*.npz" % prob)) if _compact_search(p)):
                d = np.load(path)
                power = int(d["solver_power"])
                for i in range(0, len(d["maps"]), stride):
                    if d["agents"][i, 4] == -2:
                        continue
                    m = np.ascontiguousarray(d["maps"][i])
                    it = np.zeros(4, np.int32)
                    if prob == "sokoban":
                        dist, sol = C.c_int(), C.c_int()
                        assert sim.sim_sokoban_solve2(_p(m), m.shape[0], m.shape[1], power, 0, 1, C.byref(dist), C.byref(sol), _p(it)) == 0
                        got, exp = [dist.value, sol.value], [d["stats"][i, 4], d["stats"][i, 5]]
                    elif prob == "mdungeon":
                        out = np.zeros(5, np.int32)
                        assert sim.sim_mdungeon_solve2(_p(m), m.shape[0], m.shape[1], power, 0, 1, _p(out), _p(it)) in (0, 1)
                        got, exp = list(out), [d["stats"][i, 9], d["stats"][i, 10], d["stats"][i, 6], d["stats"][i, 7], d["stats"][i, 8]]
                    else:
                        out = np.zeros(4, np.int32)
                        assert sim.sim_ddave_solve2(_p(m), m.shape[0], m.shape[1], power, 1, _p(out), _p(it)) in (0, 1)
                        got, exp = list(out), [d["stats"][i, 9], d["stats"][i, 10], d["stats"][i, 7], d["stats"][i, 8]]
                    assert got == exp, (path, i, got, exp)
                    assert np.array_equal(it, d["agents"][i, :4]), (path, i, it, d["agents"][i])
                    n += 1
                    agents += int((it > 0).sum())
    finally:
        sim.sim_set_chunk(0)
    pieces = sim.sim_pieces_reset()
    assert n > (150 if chunk == 1 else 1000)
    assert pieces > agents          # searches really were cut into more than one piece


def test_bitboard_stats_vs_oracle_random(sim):
    rs = np.random.RandomState(99)
    for prob, nt in (("binary", 2), ("zelda", 8)):
        for _ in range(400):
            h, w = rs.randint(1, 24), rs.randint(1, 40)
            if prob == "binary":
                m = (rs.random_sample((h, w)) < rs.random_sample()).astype(np.uint8)
            else:
                p = np.array([0.5, rs.uniform(0, 0.5), 0.03, 0.03, 0.03, 0.03, 0.03, 0.03])
                m = rs.choice(8, size=(h, w), p=p / p.sum()).astype(np.uint8)
            out, _ = sim_stats(sim, prob, m)
            exp = ol.get_stats(prob, m)
            assert np.array_equal(out[:len(exp)], exp), (prob, m, out, exp)


def test_extra_wave_rounds_are_harmless(sim):
    """On the GPU four maps share a wavefront and loops exit on a wave-wide test, so a group may run
    extra rounds after it converged.  The simulator injects such spurious rounds."""
    sim.sim_set_spurious.argtypes = [C.c_int]
    rs = np.random.RandomState(5)
    try:
        for prob in ("binary", "zelda", "sokoban"):
            for _ in range(300):
                h, w = rs.randint(2, 20), rs.randint(2, 36)
                if prob == "binary":
                    m = (rs.random_sample((h, w)) < rs.random_sample()).astype(np.uint8)
                else:
                    nt = 8 if prob == "zelda" else 5
                    p = np.array([0.5, rs.uniform(0, 0.5)] + [0.04] * (nt - 2))
                    m = rs.choice(nt, size=(h, w), p=p / p.sum()).astype(np.uint8)
                sim.sim_set_spurious(int(rs.randint(1, 40)))
                out, _ = sim_stats(sim, prob, m)
                exp = ol.get_stats(prob, m, solver_power=1)
                k = 4 if prob == "sokoban" else len(exp)
                assert np.array_equal(out[:k], exp[:k]), (prob, m, out, exp)
    finally:
        sim.sim_set_spurious(0)


def test_binary_incremental_update(sim):
    """binary_incremental (pcgrl_algos.h): regions and path after single-cell changes from the previous answer and
    the cached champion component, against the oracle on the full map after every change -- long random walks
    at several shapes and densities, with spurious wave rounds injected."""
    sim.sim_binary_incremental2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    sim.sim_set_spurious.argtypes = [C.c_int]
    rs = np.random.RandomState(17)
    total = ninc = 0
    counts = np.zeros(4, np.int32)
    tot = np.zeros(4, np.int64)
    for (h, w) in ((14, 14), (16, 11), (5, 5), (9, 32), (3, 7), (16, 40), (1, 9)):
        for dens in (0.3, 0.5, 0.65):
            for rep in range(3):
                m = (rs.random_sample((h, w)) < dens).astype(np.uint8)
                T = 120
                flips = rs.randint(0, h * w, size=T).astype(np.int32)
                out = np.zeros((T + 1, 2), np.int32)
                sim.sim_set_spurious(int(rs.randint(0, 30)))
                ninc += sim.sim_binary_incremental2(_p(m), h, w, _p(flips), T, _p(out), _p(counts))
                tot += counts
                sim.sim_set_spurious(0)
                cur = m.copy()
                assert np.array_equal(out[0], ol.get_stats("binary", cur))
                for t in range(T):
                    cur.flat[flips[t]] ^= 1
                    assert np.array_equal(out[t + 1], ol.get_stats("binary", cur)), ((h, w), dens, t, out[t + 1], ol.get_stats("binary", cur))
                total += T
    assert ninc > total // 3          # the incremental route is the common one
    # changes in or next to the champion (binary_touch): the rest of the changes; given up (recomputed in full) only now and then
    assert tot[1] > total // 4 and tot[2] * 5 < tot[1], tot
    print("incremental / touch / given up / full:", tot)


def test_zelda_incremental_regions(sim):
    """zelda_stats with the incremental region count (regions_incremental) along random tile writes, every step
    against the oracle on the full map."""
    sim.sim_zelda_incremental.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rs = np.random.RandomState(23)
    p = np.array([0.58, 0.3, 0.02, 0.02, 0.02, 0.02, 0.02, 0.02])
    for (h, w) in ((16, 11), (7, 11), (5, 5), (16, 32), (3, 40), (1, 6)):
        for psolid in (0.3, 0.1, 0.6):
            q = p.copy(); q[1] = psolid; q[0] = 1 - q[1:].sum()
            m = rs.choice(8, size=(h, w), p=q).astype(np.uint8)
            T = 150
            writes = np.stack([rs.randint(0, h * w, size=T), rs.choice(8, size=T, p=q)], 1).astype(np.int32)
            out = np.zeros((T + 1, 7), np.int32)
            sim.sim_zelda_incremental(_p(m), h, w, _p(writes), T, _p(out))
            cur = m.copy()
            assert np.array_equal(out[0], ol.get_stats("zelda", cur))
            for t in range(T):
                cur.flat[writes[t, 0]] = writes[t, 1]
                assert np.array_equal(out[t + 1], ol.get_stats("zelda", cur)), ((h, w), psolid, t, out[t + 1], ol.get_stats("zelda", cur))


def test_shared_rest_binary_stats(sim):
    """The cooperative form of regions + longest path (k_stats_wide: four wavefronts per tall map sharing the set
    of unretired cells) against the oracle, under the interleaving with the most duplicate extractions."""
    sim.sim_stats_shared.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    sim.sim_stats_shared.restype = C.c_long
    rs = np.random.RandomState(5)
    cases = [m for m in np.load(os.path.join(G, "stats_binary_64x64.npz"))["maps"]]
    for shape in ((64, 64), (40, 33), (17, 9), (33, 64), (64, 20)):
        for dens in (0.3, 0.5, 0.62, 0.8, 1.0):
            cases.append((rs.random_sample(shape) >= dens).astype(np.uint8))
    dups = 0
    for m in cases:
        m = np.ascontiguousarray(m, np.uint8)
        exp = ol.get_stats("binary", m)
        for ng in (4, 3, 1):
            out = np.zeros(8, np.int32)
            dups += sim.sim_stats_shared(_p(m), m.shape[0], m.shape[1], ng, _p(out))
            assert np.array_equal(out[:2], exp), (m.shape, ng, out[:2], exp)
    assert dups > 0      # the retire rule was exercised


def test_range_reward_table(sim):
    sim.sim_range_reward_i.argtypes = [C.c_int] * 4
    enc = lambda b: 2147483647 if b == np.inf else (-2147483648 if b == -np.inf else int(b))
    for lo, hi, nv, ov, r in np.load(os.path.join(G, "range_reward.npz"))["table"]:
        assert sim.sim_range_reward(nv, ov, lo, hi) == r
        assert sim.sim_range_reward_i(int(nv), int(ov), enc(lo), enc(hi)) == r   # the integer form the kernels use


def test_lazy_ring_mt_matches_numpy(sim):
    d = np.load(os.path.join(G, "rng.npz"))
    for si in range(len(d["seeds"])):
        key = np.ascontiguousarray(d["mt_key"][si])
        for bi, n in enumerate(d["bounds"]):
            out = np.zeros(64, np.int64)
            sim.sim_mt_randint(_p(key), int(n), 64, _p(out))
            assert np.array_equal(out, d["randint"][si, bi])
        f = np.zeros(700, np.float64)
        sim.sim_mt_random(_p(key), 700, _p(f))
        assert np.array_equal(f, d["random"][si])


def test_parallel_mapgen_matches_numpy(sim):
    d = np.load(os.path.join(G, "rng.npz"))
    for si in range(len(d["seeds"])):
        key = np.ascontiguousarray(d["mt_key"][si])
        for (p, nt, w, h, name) in ((d["choice2_p"], 2, 14, 14, "choice2"), (d["choice8_p"], 8, 11, 16, "choice8")):
            tiles = np.zeros((h, w), np.uint8)
            xy = np.zeros(2, np.int32)
            ring = np.zeros(624, np.uint32)
            cur = C.c_int()
            sim.sim_mt_mapgen(_p(key), _p(np.ascontiguousarray(p)), nt, w, h, _p(tiles), _p(xy), _p(ring), C.byref(cur))
            assert np.array_equal(tiles, d[name][si])
            rs = np.random.RandomState()
            rs.set_state(("MT19937", key, 624))
            rs.random_sample((h, w))
            assert xy[0] == rs.randint(w) and xy[1] == rs.randint(h)
    # big map: many rounds, ring wraps several times
    key = np.ascontiguousarray(d["mt_key"][0])
    tiles = np.zeros((64, 64), np.uint8)
    xy = np.zeros(2, np.int32)
    ring = np.zeros(624, np.uint32)
    cur = C.c_int()
    p = np.array([0.37, 0.63])
    sim.sim_mt_mapgen(_p(key), _p(p), 2, 64, 64, _p(tiles), _p(xy), _p(ring), C.byref(cur))
    rs = np.random.RandomState()
    rs.set_state(("MT19937", key, 624))
    exp = rs.choice([0, 1], size=(64, 64), p=[0.37, 0.63]).astype(np.uint8)
    assert np.array_equal(tiles, exp)
    assert xy[0] == rs.randint(64) and xy[1] == rs.randint(64)


def test_general_searches_vs_golden(sim):
    """gym_pcgrl_amd/csrc/search_big.h (what k_search_big runs: levels of more than 256 bordered cells, solver_power beyond 16 383)
    compiled for the host.  Without the shortcuts the per-agent iteration counts equal the reference's; with them (what the GPU
    runs) the results still do.  Every search fixture goes through it -- the general searches take the small levels as well."""
    for f in ("sim_big_sokoban", "sim_big_mdungeon"):
        getattr(sim, f).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * (3 if f == "sim_big_sokoban" else 2)
    sim.sim_big_ddave.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    ran = big = 0
    for path in sorted(glob.glob(os.path.join(G, "stats_sokoban_*.npz")) + glob.glob(os.path.join(G, "stats_mdungeon_*.npz")) + glob.glob(os.path.join(G, "stats_ddave_*.npz"))):
        d = np.load(path)
        prob = os.path.basename(path).split("_")[1]
        power = int(d["solver_power"])
        h, w, _ = _dims(path)
        large = not _compact_search(path)
        for i, m in enumerate(d["maps"]):
            if d["agents"][i, 4] == -2 or (not large and i % 5):      # (every large level, a fifth of the small ones)
                continue
            m = np.ascontiguousarray(m)
            exp = d["stats"][i]
            for shortcut in (0, 1):
                it = np.zeros(4, np.int32)
                if prob == "sokoban":
                    dist, sol = C.c_int(), C.c_int()
                    assert sim.sim_big_sokoban(_p(m), h, w, power, shortcut, C.byref(dist), C.byref(sol), _p(it)) == 0
                    assert (dist.value, sol.value) == (exp[4], exp[5]), (path, i, dist.value, sol.value, exp)
                elif prob == "mdungeon":
                    out5 = np.zeros(5, np.int32)
                    assert sim.sim_big_mdungeon(_p(m), h, w, power, shortcut, _p(out5), _p(it)) == 0
                    assert list(out5) == [exp[9], exp[10], exp[6], exp[7], exp[8]], (path, i, out5, exp)
                else:
                    if shortcut:
                        continue
                    out4 = np.zeros(4, np.int32)
                    assert sim.sim_big_ddave(_p(m), h, w, power, _p(out4), _p(it)) == 0
                    assert list(out4) == [exp[9], exp[10], exp[7], exp[8]], (path, i, out4, exp)
                if not shortcut:
                    assert np.array_equal(it, d["agents"][i, :4]), (path, i, it, d["agents"][i])
            ran += 1
            big += int(large)
    assert ran > 150 and big >= 40, (ran, big)
