#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the UNMODIFIED reference.

Run only where /root/reference exists (this container):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only NAME]

The reference ships no tests/golden vectors of its own, so every parity
artefact is produced here: inputs (seeds, maps, actions) and the reference's outputs.  The
fixtures are data only; no reference source travels.  `gym` is provided by gym_shim.py.

Fixture families:
  rng.npz          numpy-legacy draw rules under gym<=0.21 hashed seeding
  stats_*.npz      map -> get_stats() known-answer tests per problem (random + adversarial maps)
  sokoban_solver.npz  engineered solvable/unsolvable levels -> dist-win, sol-length, agent iterations
  range_reward.npz exhaustive small table of helper.get_range_reward
  adjust_param.npz  the adjust_param ordering quirk (Q9) and spaces
  traj_*.npz       full env trajectories (obs map/pos/heatmap, reward, done, info) with auto-reset
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import gym_shim  # noqa: E402

gym = gym_shim.install()
import gym_pcgrl  # noqa: E402,F401  (registers ids)
from gym_pcgrl.envs.helper import get_range_reward, get_string_map  # noqa: E402
from gym_pcgrl.envs.probs import PROBLEMS  # noqa: E402
from gym_pcgrl.envs.probs.sokoban.engine import AStarAgent, BFSAgent, State  # noqa: E402

INFO_KEYS = {
    "binary": ["regions", "path-length", "path-imp"],
    "zelda": ["player", "key", "door", "enemies", "regions", "nearest-enemy", "path-length"],
    "sokoban": ["player", "crate", "target", "regions", "dist-win", "sol-length"],
    "mdungeon": ["player", "exit", "potions", "treasures", "enemies", "regions", "col-potions", "col-treasures", "col-enemies",
                 "dist-win", "sol-length"],
    "ddave": ["player", "exit", "diamonds", "key", "spikes", "regions", "col-diamonds", "num-jumps", "dist-win", "sol-length"],
    "smb": ["dist-floor", "disjoint-tubes", "enemies", "empty", "noise", "jumps", "jumps-dist", "dist-win"],
}
STAT_KEYS = {
    "binary": ["regions", "path-length"],
    "zelda": ["player", "key", "door", "enemies", "regions", "nearest-enemy", "path-length"],
    "sokoban": ["player", "crate", "target", "regions", "dist-win", "sol-length"],
    "mdungeon": ["player", "exit", "potions", "treasures", "enemies", "regions", "col-potions", "col-treasures", "col-enemies",
                 "dist-win", "sol-length"],
    "ddave": ["player", "dist-floor", "exit", "diamonds", "key", "spikes", "regions", "num-jumps", "col-diamonds", "dist-win", "sol-length"],
    "smb": ["dist-floor", "disjoint-tubes", "enemies", "empty", "noise", "jumps", "jumps-dist", "dist-win"],
}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("  wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


# ----------------------------------------------------------------------------- rng
def gen_rng():
    from gym.utils import seeding
    seeds = [0, 1, 42, 2 ** 32 + 7, 1000, 2 ** 64 + 5]
    bounds = [14, 5, 11, 16, 64, 7, 1, 2, 3]
    out = {"seeds": np.array([s % 2 ** 64 for s in seeds], dtype=np.uint64), "bounds": np.array(bounds)}
    keys, randints, randoms, choices2, choices8, mixed, nodraw = [], [], [], [], [], [], []
    for s in seeds:
        rng, used = seeding.np_random(s)
        st = rng.get_state()
        assert st[2] == 624
        keys.append(np.asarray(st[1], dtype=np.uint32))
        rows = []
        for n in bounds:
            rng, _ = seeding.np_random(s)
            rows.append([rng.randint(n) for _ in range(64)])
        randints.append(rows)
        rng, _ = seeding.np_random(s)
        randoms.append([rng.random() for _ in range(700)])  # crosses the 624-word regeneration
        rng, _ = seeding.np_random(s)
        choices2.append(rng.choice([0, 1], size=(14, 14), p=[0.3, 0.7]).astype(np.uint8))
        rng, _ = seeding.np_random(s)
        p8 = np.array([0.58, 0.3, 0.02, 0.02, 0.02, 0.02, 0.02, 0.02])
        choices8.append(rng.choice(list(range(8)), size=(16, 11), p=list(p8 / p8.sum())).astype(np.uint8))
        # interleaved stream: choice map, randint(14) x2, random(), randint(5) x3
        rng, _ = seeding.np_random(s)
        m = rng.choice([0, 1], size=(5, 5), p=[0.5, 0.5]).astype(np.int64).ravel().tolist()
        m += [rng.randint(14), rng.randint(14)]
        r = rng.random()
        m += [rng.randint(5) for _ in range(3)]
        mixed.append((m, r))
        rng, _ = seeding.np_random(s)      # randint(1) must consume no draw
        nodraw.append([rng.randint(1), rng.randint(14), rng.randint(1), rng.randint(5), rng.randint(1), rng.randint(64)])
    out["mt_key"] = np.array(keys)
    out["randint"] = np.array(randints, dtype=np.int64)          # [seed, bound, 64]
    out["random"] = np.array(randoms, dtype=np.float64)          # [seed, 700]
    out["choice2_p"] = np.array([0.3, 0.7])
    out["choice2"] = np.array(choices2)
    out["choice8_p"] = p8 / p8.sum()
    out["choice8"] = np.array(choices8)
    out["mixed_ints"] = np.array([m for m, _ in mixed], dtype=np.int64)
    out["nodraw_ints"] = np.array(nodraw, dtype=np.int64)
    out["mixed_float"] = np.array([r for _, r in mixed], dtype=np.float64)
    save("rng", **out)


# ----------------------------------------------------------------------------- maps for KATs
def adversarial_binary(h, w):
    maps = []
    z = np.zeros((h, w), np.uint8)
    o = np.ones((h, w), np.uint8)
    maps += [z.copy(), o.copy()]
    m = o.copy(); m[h // 2, w // 2] = 0; maps.append(m)                       # single cell
    m = o.copy(); m[0, 0] = 0; m[h - 1, w - 1] = 0; maps.append(m)              # two corners
    m = z.copy(); m[:, 1::2] = 1; maps.append(m)                                # vertical comb, disconnected
    m = z.copy(); m[:-1, 1::2] = 1; maps.append(m)                              # comb joined at bottom
    m = z.copy(); m[1:, 1::2] = 1; maps.append(m)                               # comb joined at top
    m = z.copy(); m[1::2, :] = 1; maps.append(m)                                # horizontal stripes
    m = z.copy()                                                                # serpentine
    for y in range(1, h, 2):
        m[y, :] = 1
        m[y, (w - 1) if (y // 2) % 2 == 0 else 0] = 0
    maps.append(m)
    m = np.indices((h, w)).sum(0) % 2; maps.append(m.astype(np.uint8))          # checkerboard
    maps.append((1 - m).astype(np.uint8))
    m = o.copy()                                                                # spiral
    y0, x0, y1, x1 = 0, 0, h - 1, w - 1
    m[:] = 1
    yy, xx, k = 0, 0, 0
    sp = np.ones((h, w), np.uint8)
    top, left, bot, right = 0, 0, h - 1, w - 1
    while top <= bot and left <= right:
        sp[top, left:right + 1] = 0
        sp[top:bot + 1, right] = 0
        if top + 2 <= bot:
            sp[bot, left + 2 if left + 2 <= right else right:right + 1] = 0
        top += 2; left += 2; bot -= 2; right -= 2
    maps.append(sp)
    m = z.copy(); m[h // 2, :] = 1; maps.append(m)                              # two halves
    m = z.copy(); m[h // 2, :] = 1; m[:, w // 2] = 1; maps.append(m)            # four quadrants
    m = o.copy(); m[h // 2, :] = 0; maps.append(m)                              # a corridor (argmax ties at the ends)
    m = o.copy(); m[:, w // 2] = 0; maps.append(m)
    m = o.copy(); m[h // 2, :] = 0; m[:, w // 2] = 0; maps.append(m)            # plus sign: 4-way argmax tie
    m = z.copy(); m[1:-1, 1:-1] = 1; maps.append(m)                             # ring
    m = z.copy(); m[0, :] = 1; maps.append(m)
    return [np.ascontiguousarray(x, dtype=np.uint8) for x in maps]


def random_maps(rs, n, h, w, ntiles, probs=None):
    maps = []
    for k in range(n):
        if ntiles == 2:
            p = rs.random_sample()
            maps.append((rs.random_sample((h, w)) < p).astype(np.uint8))
        else:
            pr = np.asarray(probs, dtype=np.float64)
            solid = rs.uniform(0.0, 0.6)
            pr = pr.copy(); pr[1] = solid; pr[0] = max(1e-3, 1 - solid - pr[2:].sum())
            pr /= pr.sum()
            maps.append(rs.choice(ntiles, size=(h, w), p=pr).astype(np.uint8))
    return maps


def stats_of(prob, m):
    st = prob.get_stats(get_string_map(m, prob.get_tile_types()))
    return st


def gen_stats_binary():
    rs = np.random.RandomState(7)
    prob = PROBLEMS["binary"]()
    for (h, w, nrand) in [(14, 14, 260), (5, 5, 80), (16, 11, 60), (9, 32, 40), (64, 64, 10), (3, 7, 30), (1, 1, 0), (1, 9, 6), (8, 1, 6)]:
        prob._width, prob._height = w, h
        maps = adversarial_binary(h, w) + random_maps(rs, nrand, h, w, 2)
        if (h, w) == (64, 64):
            maps = maps[:12] + maps[-nrand:]
        res = np.array([[int(stats_of(prob, m)[k]) for k in STAT_KEYS["binary"]] for m in maps], dtype=np.int64)
        save("stats_binary_%dx%d" % (h, w), maps=np.array(maps), stats=res, keys=np.array(STAT_KEYS["binary"]))


def engineer_zelda(rs, h, w, n):
    """Maps that satisfy player==1 (often regions==1, key==1, door==1) so the BFS branches run."""
    maps = []
    for k in range(n):
        solid = rs.uniform(0.0, 0.45)
        m = (rs.random_sample((h, w)) < solid).astype(np.uint8)  # 0 empty / 1 solid
        cells = rs.permutation(h * w)
        c = 0
        def put(tile, cnt):
            nonlocal c
            for _ in range(cnt):
                m.flat[cells[c]] = tile
                c += 1
        put(2, 1 if rs.random_sample() < 0.9 else rs.randint(0, 3))
        put(3, 1 if rs.random_sample() < 0.85 else rs.randint(0, 3))
        put(4, 1 if rs.random_sample() < 0.85 else rs.randint(0, 3))
        for t in (5, 6, 7):
            put(t, rs.randint(0, 3))
        if rs.random_sample() < 0.5:
            # connect: carve solids away with decreasing density so regions==1 is likely
            m[m == 1] = (rs.random_sample((m == 1).sum()) < 0.3).astype(np.uint8)
        if rs.random_sample() < 0.15:
            # wall the door off (the d2 == -1 branch)
            ys, xs = np.where(m == 4)
            for y, x in zip(ys, xs):
                for dy, dx in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    yy, xx = y + dy, x + dx
                    if 0 <= yy < h and 0 <= xx < w and m[yy, xx] not in (2, 3):
                        m[yy, xx] = 1
        maps.append(m)
    return maps


def gen_stats_zelda():
    rs = np.random.RandomState(11)
    prob = PROBLEMS["zelda"]()
    pr = [prob._prob[t] for t in prob.get_tile_types()]
    for (h, w, nrand, neng) in [(7, 11, 120, 300), (16, 11, 80, 300), (11, 16, 30, 80), (5, 5, 40, 100)]:
        prob._width, prob._height = w, h
        maps = [np.zeros((h, w), np.uint8), np.ones((h, w), np.uint8)]
        for t in range(2, 8):
            maps.append(np.full((h, w), t, np.uint8))
        maps += random_maps(rs, nrand, h, w, 8, pr) + engineer_zelda(rs, h, w, neng)
        res = np.array([[int(stats_of(prob, m)[k]) for k in STAT_KEYS["zelda"]] for m in maps], dtype=np.int64)
        hit = int(((res[:, 0] == 1) & (res[:, 4] == 1)).sum())
        print("  zelda %dx%d: %d maps, precondition hit %d, key/door branch %d, d2=-1-ish %d" % (
            h, w, len(maps), hit, int(((res[:, 0] == 1) & (res[:, 4] == 1) & (res[:, 1] == 1) & (res[:, 2] == 1)).sum()),
            int((res[:, 6] < 0).sum() + 0)))
        save("stats_zelda_%dx%d" % (h, w), maps=np.array(maps), stats=res, keys=np.array(STAT_KEYS["zelda"]))


def gen_stats_big():
    """Maps beyond 64 x 64 (round 4): the reference takes any width / height (pcgrl_env.py:106-115, probs/problem.py:66-72)."""
    rs = np.random.RandomState(101)
    prob = PROBLEMS["binary"]()
    for (h, w, nrand) in [(100, 100, 6), (65, 130, 6), (70, 200, 3), (255, 3, 4), (2, 255, 4)]:
        prob._width, prob._height = w, h
        t0 = time.time()
        maps = adversarial_binary(h, w) + random_maps(rs, nrand, h, w, 2)
        res = np.array([[int(stats_of(prob, m)[k]) for k in STAT_KEYS["binary"]] for m in maps], dtype=np.int64)
        print("  binary %dx%d: %d maps, %.1fs" % (h, w, len(maps), time.time() - t0))
        save("stats_binary_%dx%d" % (h, w), maps=np.array(maps), stats=res, keys=np.array(STAT_KEYS["binary"]))
    prob = PROBLEMS["zelda"]()
    pr = [prob._prob[t] for t in prob.get_tile_types()]
    for (h, w, nrand, neng) in [(80, 70, 6, 40), (20, 130, 4, 40)]:
        prob._width, prob._height = w, h
        t0 = time.time()
        maps = random_maps(rs, nrand, h, w, 8, pr) + engineer_zelda(rs, h, w, neng)
        res = np.array([[int(stats_of(prob, m)[k]) for k in STAT_KEYS["zelda"]] for m in maps], dtype=np.int64)
        print("  zelda %dx%d: %d maps, precondition hit %d, key/door branch %d, %.1fs" % (
            h, w, len(maps), int(((res[:, 0] == 1) & (res[:, 4] == 1)).sum()),
            int(((res[:, 0] == 1) & (res[:, 4] == 1) & (res[:, 1] == 1) & (res[:, 2] == 1)).sum()), time.time() - t0))
        save("stats_zelda_%dx%d" % (h, w), maps=np.array(maps), stats=res, keys=np.array(STAT_KEYS["zelda"]))


def gen_stats_big_search():
    """The search problems beyond the compact searches (round 4): bordered levels of more than 256 cells, solver_power beyond
    16 383 (sokoban_prob.py:60-73, mdungeon_prob.py:68-84, ddave_prob.py:67-82 take any)."""
    rs = np.random.RandomState(211)
    prob = PROBLEMS["sokoban"]()
    pr = [prob._prob[t] for t in prob.get_tile_types()]
    for (h, w, nrand, neng, max_k, power, smax) in [(20, 20, 4, 14, 3, 700, 0.3), (16, 24, 0, 8, 2, 2500, 0.12), (8, 8, 0, 8, 4, 20000, 0.03)]:
        prob._width, prob._height, prob._solver_power = w, h, power
        maps = random_maps(rs, nrand, h, w, 5, pr) + engineer_sokoban(rs, h, w, neng, max_k, smax)
        t0 = time.time()
        res, agents = [], []
        for m in maps:
            st = stats_of(prob, m)
            row = [int(st[k]) if k != "sol-length" else len(st["solution"]) for k in STAT_KEYS["sokoban"]]
            res.append(row)
            if st["player"] == 1 and st["crate"] == st["target"] and st["crate"] > 0 and st["regions"] == 1:
                iters, win, dist, sl = run_agents(prob, m)
                assert dist == row[4] and sl == row[5], (dist, sl, row)
                agents.append(iters + [win])
            else:
                agents.append([0, 0, 0, 0, -2])
        res = np.array(res, dtype=np.int64); agents = np.array(agents, dtype=np.int64)
        print("  sokoban %dx%d power %d: %d maps, solver ran %d, wins by agent %s, cap hits %d, %.1fs" % (
            h, w, power, len(maps), int((agents[:, 4] > -2).sum()), [int((agents[:, 4] == k).sum()) for k in (-1, 0, 1, 2, 3)],
            int((agents[:, :4] >= power).any(1).sum()), time.time() - t0))
        save("stats_sokoban_%dx%d_p%d" % (h, w, power), maps=np.array(maps), stats=res, agents=agents,
             solver_power=np.array(power), keys=np.array(STAT_KEYS["sokoban"]))
    prob = PROBLEMS["mdungeon"]()
    pr = [prob._prob[t] for t in prob.get_tile_types()]
    for (h, w, nrand, neng, power, smax, guard) in [(20, 20, 4, 14, 900, 0.3, 0.3), (30, 12, 0, 8, 2500, 0.2, 0.5), (12, 12, 0, 6, 20000, 0.03, 0.95)]:
        prob._width, prob._height, prob._solver_power = w, h, power
        maps = random_maps(rs, nrand, h, w, 8, pr) + engineer_mdungeon(rs, h, w, neng, smax, guard)
        t0 = time.time()
        res, agents = [], []
        for m in maps:
            st = stats_of(prob, m)
            row = [int(st[k]) for k in STAT_KEYS["mdungeon"]]
            res.append(row)
            if st["player"] == 1 and st["exit"] == 1 and st["regions"] == 1:
                iters, win, dist, sl, gs = run_agents_mdungeon(prob, m)
                assert dist == row[9] and sl == row[10], (dist, sl, row)
                agents.append(iters + [win])
            else:
                agents.append([0, 0, 0, 0, -2])
        res = np.array(res, dtype=np.int64); agents = np.array(agents, dtype=np.int64)
        print("  mdungeon %dx%d power %d: %d maps, solver ran %d, wins by agent %s, cap hits %d, %.1fs" % (
            h, w, power, len(maps), int((agents[:, 4] > -2).sum()), [int((agents[:, 4] == k).sum()) for k in (-1, 0, 1, 2, 3)],
            int((agents[:, :4] >= power).any(1).sum()), time.time() - t0))
        save("stats_mdungeon_%dx%d_p%d" % (h, w, power), maps=np.array(maps), stats=res, agents=agents,
             solver_power=np.array(power), keys=np.array(STAT_KEYS["mdungeon"]))
    prob = PROBLEMS["ddave"]()
    pr = [prob._prob[t] for t in prob.get_tile_types()]
    for (h, w, nrand, neng, power, smax, spk) in [(20, 20, 4, 14, 900, 0.3, 0.1), (12, 30, 0, 8, 2500, 0.15, 0.05), (12, 12, 0, 6, 20000, 0.08, 0.0)]:
        prob._width, prob._height, prob._solver_power = w, h, power
        probs7 = np.array(pr + [0.0])
        maps = [np.minimum(x, 6) for x in random_maps(rs, nrand, h, w, 8, list(probs7))] + engineer_ddave(rs, h, w, neng, smax, spk)
        t0 = time.time()
        res, agents = [], []
        for m in maps:
            st = stats_of(prob, m)
            row = [int(st[k]) for k in STAT_KEYS["ddave"]]
            res.append(row)
            if st["player"] == 1 and st["exit"] == 1 and st["key"] == 1 and st["regions"] == 1:
                iters, win, dist, sl, gs = run_agents_ddave(prob, m)
                assert dist == row[9] and sl == row[10], (dist, sl, row)
                agents.append(iters + [win])
            else:
                agents.append([0, 0, 0, 0, -2])
        res = np.array(res, dtype=np.int64); agents = np.array(agents, dtype=np.int64)
        print("  ddave %dx%d power %d: %d maps, solver ran %d, wins by agent %s, cap hits %d, %.1fs" % (
            h, w, power, len(maps), int((agents[:, 4] > -2).sum()), [int((agents[:, 4] == k).sum()) for k in (-1, 0, 1, 2, 3)],
            int((agents[:, :4] >= power).any(1).sum()), time.time() - t0))
        save("stats_ddave_%dx%d_p%d" % (h, w, power), maps=np.array(maps), stats=res, agents=agents,
             solver_power=np.array(power), keys=np.array(STAT_KEYS["ddave"]))


def gen_stats_huge_search():
    """Search levels of more than 4 096 bordered cells (round 5): sokoban 70 x 70 and mdungeon 66 x 80 -- the
    reference takes any size (sokoban_prob.py:60-73, mdungeon_prob.py:68-84)."""
    rs = np.random.RandomState(977)
    prob = PROBLEMS["sokoban"]()
    h, w, power = 70, 70, 150
    prob._width, prob._height, prob._solver_power = w, h, power
    maps = engineer_sokoban(rs, h, w, 6, 2, 0.05)
    t0 = time.time()
    res, agents = [], []
    for m in maps:
        st = stats_of(prob, m)
        row = [int(st[k]) if k != "sol-length" else len(st["solution"]) for k in STAT_KEYS["sokoban"]]
        res.append(row)
        if st["player"] == 1 and st["crate"] == st["target"] and st["crate"] > 0 and st["regions"] == 1:
            iters, win, dist, sl = run_agents(prob, m)
            assert dist == row[4] and sl == row[5], (dist, sl, row)
            agents.append(iters + [win])
        else:
            agents.append([0, 0, 0, 0, -2])
    res = np.array(res, dtype=np.int64); agents = np.array(agents, dtype=np.int64)
    print("  sokoban %dx%d power %d: %d maps, solver ran %d, wins by agent %s, %.1fs" % (
        h, w, power, len(maps), int((agents[:, 4] > -2).sum()), [int((agents[:, 4] == k).sum()) for k in (-1, 0, 1, 2, 3)], time.time() - t0))
    save("stats_sokoban_%dx%d_p%d" % (h, w, power), maps=np.array(maps), stats=res, agents=agents,
         solver_power=np.array(power), keys=np.array(STAT_KEYS["sokoban"]))
    prob = PROBLEMS["mdungeon"]()
    h, w, power = 66, 80, 150
    prob._width, prob._height, prob._solver_power = w, h, power
    maps = engineer_mdungeon(rs, h, w, 5, 0.04, 0.3)
    t0 = time.time()
    res, agents = [], []
    for m in maps:
        st = stats_of(prob, m)
        row = [int(st[k]) for k in STAT_KEYS["mdungeon"]]
        res.append(row)
        if st["player"] == 1 and st["exit"] == 1 and st["regions"] == 1:
            iters, win, dist, sl, gs = run_agents_mdungeon(prob, m)
            assert dist == row[9] and sl == row[10], (dist, sl, row)
            agents.append(iters + [win])
        else:
            agents.append([0, 0, 0, 0, -2])
    res = np.array(res, dtype=np.int64); agents = np.array(agents, dtype=np.int64)
    print("  mdungeon %dx%d power %d: %d maps, solver ran %d, wins by agent %s, %.1fs" % (
        h, w, power, len(maps), int((agents[:, 4] > -2).sum()), [int((agents[:, 4] == k).sum()) for k in (-1, 0, 1, 2, 3)], time.time() - t0))
    save("stats_mdungeon_%dx%d_p%d" % (h, w, power), maps=np.array(maps), stats=res, agents=agents,
         solver_power=np.array(power), keys=np.array(STAT_KEYS["mdungeon"]))


def engineer_sokoban(rs, h, w, n, max_k=3, solid_max=0.35):
    maps = []
    for _ in range(n):
        solid = rs.uniform(0.0, solid_max)
        m = (rs.random_sample((h, w)) < solid).astype(np.uint8)
        cells = rs.permutation(h * w)
        k = rs.randint(1, max_k + 1)
        m.flat[cells[0]] = 2
        for i in range(k):
            m.flat[cells[1 + i]] = 3
        nt = k if rs.random_sample() < 0.9 else rs.randint(0, max_k + 1)
        for i in range(nt):
            m.flat[cells[1 + k + i]] = 4
        maps.append(m)
    return maps


def run_agents(prob, m):
    """Re-run what SokobanProblem._run_game does (sokoban_prob.py:85-122), keeping iteration counts."""
    smap = get_string_map(m, prob.get_tile_types())
    chars = " #@$."
    s2c = dict((s, chars[i]) for i, s in enumerate(prob.get_tile_types()))
    W = prob._width
    lvl = "#" * (W + 2) + "\n"
    for row in smap:
        lvl += "#" + "".join(s2c[c] for c in row) + "#\n"
    lvl += "#" * (W + 2) + "\n"
    state = State()
    state.stringInitialize(lvl.split("\n"))
    iters = []
    win = -1
    sol, st, it = BFSAgent().getSolution(state, prob._solver_power)
    iters.append(it)
    if st.checkWin():
        win = 0
    else:
        for k, bal in enumerate((1, 0.5, 0)):
            sol, st, it = AStarAgent().getSolution(state, bal, prob._solver_power)
            iters.append(it)
            if st.checkWin():
                win = k + 1
                break
    while len(iters) < 4:
        iters.append(0)
    return iters, win, (0 if win >= 0 else st.getHeuristic()), (len(sol) if win >= 0 else 0)


def gen_stats_sokoban():
    rs = np.random.RandomState(13)
    prob = PROBLEMS["sokoban"]()
    pr = [prob._prob[t] for t in prob.get_tile_types()]
    for (h, w, nrand, neng, max_k, power, smax) in [(5, 5, 150, 260, 3, 5000, 0.35), (6, 6, 20, 60, 3, 5000, 0.35),
                                                     (7, 7, 10, 40, 4, 5000, 0.35), (4, 6, 20, 60, 2, 300, 0.35),
                                                     (5, 6, 0, 120, 3, 5000, 0.12), (7, 8, 0, 40, 4, 5000, 0.08),
                                                     (8, 8, 0, 24, 3, 2000, 0.05)]:
        prob._width, prob._height, prob._solver_power = w, h, power
        maps = [np.zeros((h, w), np.uint8), np.ones((h, w), np.uint8)]
        maps += random_maps(rs, nrand, h, w, 5, pr) + engineer_sokoban(rs, h, w, neng, max_k, smax)
        t0 = time.time()
        res, agents = [], []
        for m in maps:
            st = stats_of(prob, m)
            row = [int(st[k]) if k != "sol-length" else len(st["solution"]) for k in STAT_KEYS["sokoban"]]
            res.append(row)
            if st["player"] == 1 and st["crate"] == st["target"] and st["crate"] > 0 and st["regions"] == 1:
                iters, win, dist, sl = run_agents(prob, m)
                assert dist == row[4] and sl == row[5], (dist, sl, row)
                agents.append(iters + [win])
            else:
                agents.append([0, 0, 0, 0, -2])
        res = np.array(res, dtype=np.int64)
        agents = np.array(agents, dtype=np.int64)
        print("  sokoban %dx%d: %d maps, solver ran %d, wins by agent %s, cap hits %d, %.1fs" % (
            h, w, len(maps), int((agents[:, 4] > -2).sum()),
            [int((agents[:, 4] == k).sum()) for k in (-1, 0, 1, 2, 3)],
            int((agents[:, :4] >= power).any(1).sum()), time.time() - t0))
        save("stats_sokoban_%dx%d" % (h, w), maps=np.array(maps), stats=res, agents=agents,
             solver_power=np.array(power), keys=np.array(STAT_KEYS["sokoban"]))


# ----------------------------------------------------------------------------- mdungeon
def engineer_mdungeon(rs, h, w, n, solid_max=0.35, guard=0.25):
    """Maps with one player and one exit (often connected), a sprinkle of potions / treasures / goblins / ogres, and
    now and then an exit walled in by ogres so that the planner cannot win and runs into its iteration cap."""
    maps = []
    for _ in range(n):
        solid = rs.uniform(0.0, solid_max)
        m = (rs.random_sample((h, w)) < solid).astype(np.uint8)
        cells = rs.permutation(h * w)
        c = 0
        def put(tile, cnt):
            nonlocal c
            for _ in range(cnt):
                if c < len(cells):
                    m.flat[cells[c]] = tile
                    c += 1
        put(2, 1 if rs.random_sample() < 0.92 else rs.randint(0, 3))
        put(3, 1 if rs.random_sample() < 0.92 else rs.randint(0, 3))
        dens = rs.uniform(0.0, 1.0)
        for t in (4, 5, 6, 7):
            put(t, rs.randint(0, 1 + int(dens * 4)))
        if rs.random_sample() < guard:
            ys, xs = np.where(m == 3)
            R = rs.randint(1, 4)           # rings of monsters around the exit: 2-3 of them cost more than 5 health
            strong = rs.uniform(0.6, 1.0)
            for y, x in zip(ys, xs):
                for dy in range(-R, R + 1):
                    for dx in range(-R, R + 1):
                        yy, xx = y + dy, x + dx
                        if (dy or dx) and 0 <= yy < h and 0 <= xx < w and m[yy, xx] not in (2, 3):
                            m[yy, xx] = 7 if rs.random_sample() < strong else 6
        maps.append(m)
    return maps


def run_agents_mdungeon(prob, m):
    """What MDungeonProblem._run_game does (mdungeon_prob.py:91-126), keeping the agents' iteration counts."""
    from gym_pcgrl.envs.probs.mdungeon.engine import AStarAgent as MA, BFSAgent as MB, State as MS
    smap = get_string_map(m, prob.get_tile_types())
    chars = " #@H*$go"
    s2c = dict((s, chars[i]) for i, s in enumerate(prob.get_tile_types()))
    W = prob._width
    lvl = "#" * (W + 2) + "\n"
    for row in smap:
        lvl += "#" + "".join(s2c[c] for c in row) + "#\n"
    lvl += "#" * (W + 2) + "\n"
    state = MS()
    state.stringInitialize(lvl.split("\n"))
    iters, win = [], -1
    for k, bal in enumerate((1, 0.5, 0)):
        sol, st, it = MA().getSolution(state, bal, prob._solver_power)
        iters.append(it)
        if st.checkWin():
            win = k
            break
    if win < 0:
        sol, st, it = MB().getSolution(state, prob._solver_power)
        iters.append(it)
        if st.checkWin():
            win = 3
    while len(iters) < 4:
        iters.append(0)
    gs = st.getGameStatus()
    return iters, win, (0 if win >= 0 else st.getHeuristic()), (len(sol) if win >= 0 else 0), gs


def gen_stats_mdungeon():
    rs = np.random.RandomState(17)
    prob = PROBLEMS["mdungeon"]()
    pr = [prob._prob[t] for t in prob.get_tile_types()]
    for (h, w, nrand, neng, power, smax, guard) in [(11, 7, 120, 260, 5000, 0.35, 0.25), (7, 11, 20, 80, 5000, 0.3, 0.3),
                                                     (5, 5, 40, 120, 5000, 0.3, 0.3), (6, 9, 10, 80, 300, 0.25, 0.5),
                                                     (11, 7, 0, 60, 1200, 0.1, 0.9), (14, 14, 0, 24, 5000, 0.2, 0.5),
                                                     (1, 6, 0, 30, 5000, 0.1, 0.2)]:
        prob._width, prob._height, prob._solver_power = w, h, power
        maps = [np.zeros((h, w), np.uint8), np.ones((h, w), np.uint8)]
        maps += random_maps(rs, nrand, h, w, 8, pr) + engineer_mdungeon(rs, h, w, neng, smax, guard)
        t0 = time.time()
        res, agents = [], []
        for m in maps:
            st = stats_of(prob, m)
            row = [int(st[k]) for k in STAT_KEYS["mdungeon"]]
            res.append(row)
            if st["player"] == 1 and st["exit"] == 1 and st["regions"] == 1:
                iters, win, dist, sl, gs = run_agents_mdungeon(prob, m)
                assert dist == row[9] and sl == row[10], (dist, sl, row)
                assert [gs["col_potions"], gs["col_treasures"], gs["col_enemies"]] == row[6:9]
                agents.append(iters + [win])
            else:
                agents.append([0, 0, 0, 0, -2])
        res = np.array(res, dtype=np.int64)
        agents = np.array(agents, dtype=np.int64)
        print("  mdungeon %dx%d power %d: %d maps, solver ran %d, wins by agent %s, cap hits %d, %.1fs" % (
            h, w, power, len(maps), int((agents[:, 4] > -2).sum()),
            [int((agents[:, 4] == k).sum()) for k in (-1, 0, 1, 2, 3)],
            int((agents[:, :4] >= power).any(1).sum()), time.time() - t0))
        save("stats_mdungeon_%dx%d_p%d" % (h, w, power), maps=np.array(maps), stats=res, agents=agents,
             solver_power=np.array(power), keys=np.array(STAT_KEYS["mdungeon"]))


# ----------------------------------------------------------------------------- ddave
def engineer_ddave(rs, h, w, n, solid_max=0.35, spike_max=0.15):
    """Maps with one player, exit and key (often connected), platforms, diamonds and spikes."""
    maps = []
    for _ in range(n):
        solid = rs.uniform(0.0, solid_max)
        m = (rs.random_sample((h, w)) < solid).astype(np.uint8)
        if rs.random_sample() < 0.5 and h > 2:
            m[:] = 0                                  # platform style: a few horizontal ledges
            for _k in range(rs.randint(0, 4)):
                y = rs.randint(1, h); x0 = rs.randint(0, w); x1 = rs.randint(x0, w) + 1
                m[y, x0:x1] = 1
        cells = rs.permutation(h * w)
        c = 0
        def put(tile, cnt):
            nonlocal c
            for _ in range(cnt):
                if c < len(cells):
                    m.flat[cells[c]] = tile
                    c += 1
        put(2, 1 if rs.random_sample() < 0.92 else rs.randint(0, 3))
        put(3, 1 if rs.random_sample() < 0.92 else rs.randint(0, 3))
        put(5, 1 if rs.random_sample() < 0.92 else rs.randint(0, 3))
        put(4, rs.randint(0, 5))
        put(6, rs.randint(0, 1 + int(spike_max * h * w)))
        maps.append(m)
    return maps


def run_agents_ddave(prob, m):
    """What DDaveProblem._run_game does (ddave_prob.py:92-127), keeping the agents' iteration counts."""
    from gym_pcgrl.envs.probs.ddave.engine import AStarAgent as DA, BFSAgent as DB, State as DS
    smap = get_string_map(m, prob.get_tile_types())
    chars = " #@H$V*"
    s2c = dict((s, chars[i]) for i, s in enumerate(prob.get_tile_types()))
    W = prob._width
    lvl = "#" * (W + 2) + "\n"
    for row in smap:
        lvl += "#" + "".join(s2c[c] for c in row) + "#\n"
    lvl += "#" * (W + 2) + "\n"
    state = DS()
    state.stringInitialize(lvl.split("\n"))
    iters, win = [], -1
    for k, bal in enumerate((1, 0.5, 0)):
        sol, st, it = DA().getSolution(state, bal, prob._solver_power)
        iters.append(it)
        if st.checkWin():
            win = k
            break
    if win < 0:
        sol, st, it = DB().getSolution(state, prob._solver_power)
        iters.append(it)
        if st.checkWin():
            win = 3
    while len(iters) < 4:
        iters.append(0)
    return iters, win, (0 if win >= 0 else st.getHeuristic()), (len(sol) if win >= 0 else 0), st.getGameStatus()


def gen_stats_ddave():
    rs = np.random.RandomState(19)
    prob = PROBLEMS["ddave"]()
    pr = [prob._prob[t] for t in prob.get_tile_types()]
    for (h, w, nrand, neng, power, smax, spk) in [(7, 11, 120, 300, 5000, 0.35, 0.15), (11, 7, 20, 80, 5000, 0.3, 0.1),
                                                  (5, 5, 40, 120, 5000, 0.3, 0.15), (6, 9, 10, 100, 150, 0.25, 0.1),
                                                  (7, 11, 0, 80, 600, 0.1, 0.3), (14, 14, 0, 24, 5000, 0.2, 0.1),
                                                  (2, 6, 0, 40, 5000, 0.1, 0.1), (1, 5, 0, 12, 5000, 0.0, 0.0)]:
        prob._width, prob._height, prob._solver_power = w, h, power
        maps = [np.zeros((h, w), np.uint8), np.ones((h, w), np.uint8)]
        probs7 = np.array(pr + [0.0])
        maps += [np.minimum(x, 6) for x in random_maps(rs, nrand, h, w, 8, list(probs7))] + engineer_ddave(rs, h, w, neng, smax, spk)
        t0 = time.time()
        res, agents = [], []
        for m in maps:
            st = stats_of(prob, m)
            row = [int(st[k]) for k in STAT_KEYS["ddave"]]
            res.append(row)
            if st["player"] == 1 and st["exit"] == 1 and st["key"] == 1 and st["regions"] == 1:
                iters, win, dist, sl, gs = run_agents_ddave(prob, m)
                assert dist == row[9] and sl == row[10], (dist, sl, row)
                assert [gs["num_jumps"], gs["col_diamonds"]] == row[7:9]
                agents.append(iters + [win])
            else:
                agents.append([0, 0, 0, 0, -2])
        res = np.array(res, dtype=np.int64)
        agents = np.array(agents, dtype=np.int64)
        print("  ddave %dx%d power %d: %d maps, solver ran %d, wins by agent %s, cap hits %d, %.1fs" % (
            h, w, power, len(maps), int((agents[:, 4] > -2).sum()),
            [int((agents[:, 4] == k).sum()) for k in (-1, 0, 1, 2, 3)],
            int((agents[:, :4] >= power).any(1).sum()), time.time() - t0))
        save("stats_ddave_%dx%d_p%d" % (h, w, power), maps=np.array(maps), stats=res, agents=agents,
             solver_power=np.array(power), keys=np.array(STAT_KEYS["ddave"]))



# ----------------------------------------------------------------------------- smb
def engineer_smb(rs, h, w, n):
    """Platformer levels: a floor with gaps, walls of different heights, stairs, ledges, tubes, enemies and coins."""
    maps = []
    for k in range(n):
        m = np.zeros((h, w), np.uint8)
        style = k % 7
        if style != 5:
            m[h - 2:, :] = 1                                        # floor
            for _ in range(rs.randint(0, 4)):                       # gaps (1-5 wide)
                x0 = rs.randint(0, w); m[h - 2:, x0:x0 + rs.randint(1, 6)] = 0
        if style in (1, 2, 4):
            for _ in range(rs.randint(1, 5)):                       # walls / stairs
                x0 = rs.randint(0, w); hh = rs.randint(1, min(7, h - 2))
                if style == 2:
                    for i in range(hh):
                        if x0 + i < w:
                            m[h - 2 - (i + 1):h - 2, x0 + i] = 3
                else:
                    m[h - 2 - hh:h - 2, x0] = rs.choice([1, 3, 4, 6])
        if style in (3, 4):
            for _ in range(rs.randint(1, 6)):                       # ledges
                y = rs.randint(1, h - 2); x0 = rs.randint(0, w); m[y, x0:x0 + rs.randint(1, 8)] = rs.choice([3, 4])
        if style == 5:
            m = (rs.random_sample((h, w)) < rs.uniform(0.05, 0.3)).astype(np.uint8)
        if style == 6:                                              # a wall nobody can jump: the searches run long
            m[:h - 2, rs.randint(w // 2, w)] = 1
        for _ in range(rs.randint(0, 4)):                           # tubes: pairs and singles
            x0 = rs.randint(0, w - 1); y0 = rs.randint(max(1, h - 6), h - 1)
            m[y0:h - 2, x0] = 6
            if rs.random_sample() < 0.7:
                m[y0:h - 2, x0 + 1] = 6
        for _ in range(rs.randint(0, 8)):
            m[rs.randint(0, h), rs.randint(0, w)] = 2               # enemies
        for _ in range(rs.randint(0, 8)):
            m[rs.randint(0, h), rs.randint(0, w)] = 5               # coins
        maps.append(m)
    return maps


def run_agents_smb(prob, m):
    """What SMBProblem._run_game does (smb_prob.py:106-145), keeping the agents' iteration counts."""
    from gym_pcgrl.envs.probs.smb.engine import AStarAgent as SA, State as SS
    smap = get_string_map(m, prob.get_tile_types())
    chars = " # ## #"
    s2c = dict((s, chars[i]) for i, s in enumerate(prob.get_tile_types()))
    H = prob._height
    lvl = ""
    for i, row in enumerate(smap):
        lvl += "   " if i < H - 3 else (" @ " if i == H - 3 else "###")
        lvl += "".join(s2c[c] for c in row)
        lvl += " | " if i < H - 3 else (" # " if i == H - 3 else "###")
        lvl += "\n"
    state = SS()
    state.stringInitialize(lvl.split("\n"))
    iters = [0, 0, 0, 0]
    sol, st, it = SA().getSolution(state, 1, prob._solver_power)
    iters[0] = it
    win = 0 if st.checkWin() else -1
    if win < 0:
        sol, st, it = SA().getSolution(state, 0, prob._solver_power)
        iters[1] = it
        win = 1 if st.checkWin() else -1
    return iters, win, (0 if win >= 0 else st.getHeuristic()), st.getGameStatus()


def gen_stats_smb():
    rs = np.random.RandomState(23)
    prob = PROBLEMS["smb"]()
    pr = [prob._prob[t] for t in prob.get_tile_types()]
    for (h, w, nrand, neng, power) in [(14, 114, 3, 9, 10000), (14, 114, 4, 20, 1500), (10, 30, 10, 60, 10000), (8, 24, 10, 60, 400),
                                       (6, 12, 10, 40, 10000), (4, 9, 6, 20, 200), (14, 40, 4, 24, 2500)]:
        prob._width, prob._height, prob._solver_power = w, h, power
        maps = [np.zeros((h, w), np.uint8), np.ones((h, w), np.uint8)]
        flat = np.zeros((h, w), np.uint8); flat[h - 2:, :] = 1
        maps.append(flat)
        maps += [np.minimum(x, 6) for x in random_maps(rs, nrand, h, w, 7, pr)] + engineer_smb(rs, h, w, neng)
        t0 = time.time()
        res, agents = [], []
        for m in maps:
            st = stats_of(prob, m)
            row = [int(st[k]) for k in STAT_KEYS["smb"]]
            res.append(row)
            iters, win, dist, gs = run_agents_smb(prob, m)
            assert dist == row[7] and gs["jumps"] == row[5], (dist, gs["jumps"], row)
            agents.append(iters + [win])
        res = np.array(res, dtype=np.int64)
        agents = np.array(agents, dtype=np.int64)
        print("  smb %dx%d power %d: %d maps, wins by agent %s, cap hits %d, %.1fs" % (
            h, w, power, len(maps), [int((agents[:, 4] == k).sum()) for k in (-1, 0, 1)],
            int((agents[:, :2] >= power).any(1).sum()), time.time() - t0))
        save("stats_smb_%dx%d_p%d" % (h, w, power), maps=np.array(maps), stats=res, agents=agents,
             solver_power=np.array(power), keys=np.array(STAT_KEYS["smb"]))


# ----------------------------------------------------------------------------- range reward
def gen_range_reward():
    bands = [(1, 1), (np.inf, np.inf), (-np.inf, -np.inf), (2, 5), (4, np.inf), (1, 3), (0, 0), (3, 3), (2, 2), (1, 5)]
    vals = list(range(-2, 12)) + [25, 176, 196, 250]
    rows = []
    for lo, hi in bands:
        for nv in vals:
            for ov in vals:
                r = get_range_reward(nv, ov, lo, hi)
                assert r is not None
                rows.append((lo, hi, nv, ov, float(r)))
    save("range_reward", table=np.array(rows, dtype=np.float64))


# ----------------------------------------------------------------------------- adjust_param / spaces
def space_desc(sp):
    from gym import spaces
    if isinstance(sp, spaces.Discrete):
        return [0, sp.n, 0, 0]
    if isinstance(sp, spaces.MultiDiscrete):
        return [1] + [int(v) for v in sp.nvec]
    raise TypeError(sp)


def gen_adjust_param():
    rows = []
    cases = [
        ("binary", "narrow", []),
        ("binary", "narrow", [dict(width=64, height=64)]),
        ("binary", "turtle", [dict(width=64, height=64), dict(change_percentage=0.2)]),
        ("binary", "turtle", [dict(width=64, height=64, change_percentage=0.2)]),
        ("binary", "wide", [dict(change_percentage=0.5)]),
        ("binary", "narrow", [dict(change_percentage=2.0)]),
        ("binary", "narrow", [dict(change_percentage=-1.0)]),
        ("binary", "narrow", [dict(change_percentage=0.001)]),
        ("zelda", "wide", []),
        ("zelda", "wide", [dict(width=11, height=16)]),
        ("zelda", "wide", [dict(width=11, height=16), dict(change_percentage=0.2)]),
        ("zelda", "narrow", [dict(width=16, height=11, change_percentage=0.1)]),
        ("sokoban", "narrow", []),
        ("sokoban", "narrow", [dict(width=7, height=6), dict(change_percentage=0.4)]),
        ("sokoban", "turtle", [dict(change_percentage=1.0)]),
        ("mdungeon", "narrow", []),
        ("mdungeon", "wide", [dict(width=5, height=6), dict(change_percentage=0.9)]),
        ("mdungeon", "turtle", [dict(width=14, height=14, change_percentage=0.3)]),
        ("ddave", "narrow", []),
        ("ddave", "wide", [dict(width=6, height=5), dict(change_percentage=0.9)]),
    ]
    for prob, rep, calls in cases:
        env = gym.make("%s-%s-v0" % (prob, rep))
        for kw in calls:
            env.adjust_param(**kw)
        osp = env.observation_space.spaces
        rows.append([env._prob._width, env._prob._height, env._max_changes, env._max_iterations,
                     env.get_num_tiles(), env.get_border_tile()] + space_desc(env.action_space)
                    + [int(osp["map"].shape[0]), int(osp["map"].shape[1]), int(osp["map"].high.max()),
                       int(osp["heatmap"].high.max()), int("pos" in osp)])
    save("adjust_param", cases=np.array([repr(c) for c in cases]), rows=np.array(rows, dtype=np.int64))


# ----------------------------------------------------------------------------- trajectories
def sample_actions(rs, rep, T, E, W, H, ntiles):
    if rep == "narrow":
        return rs.randint(0, ntiles + 1, size=(T, E, 1)).astype(np.int32)
    if rep == "turtle":
        return rs.randint(0, ntiles + 4, size=(T, E, 1)).astype(np.int32)
    if rep == "narrowcast":
        return np.stack([rs.randint(0, 3, size=(T, E)), rs.randint(0, ntiles, size=(T, E))], -1).astype(np.int32)
    if rep == "turtlecast":
        return np.stack([rs.randint(0, 6, size=(T, E)), rs.randint(0, ntiles, size=(T, E))], -1).astype(np.int32)
    if rep == "narrowmulti":
        return rs.randint(0, ntiles + 1, size=(T, E, 9)).astype(np.int32)
    a = np.stack([rs.randint(0, W, size=(T, E)), rs.randint(0, H, size=(T, E)), rs.randint(0, ntiles, size=(T, E))], -1)
    return a.astype(np.int32)


def gen_traj(name, prob, rep, E, T, calls=(), seed0=1000, action_seed=5):
    t0 = time.time()
    envs = []
    for i in range(E):
        env = gym.make("%s-%s-v0" % (prob, rep))
        for kw in calls:
            env.adjust_param(**kw)
        env.seed(seed0 + i)
        envs.append(env)
    W, H = envs[0]._prob._width, envs[0]._prob._height
    nt = envs[0].get_num_tiles()
    keys = INFO_KEYS[prob]
    acts = sample_actions(np.random.RandomState(action_seed), rep, T, E, W, H, nt)
    has_pos = rep != "wide"
    map0 = np.zeros((E, H, W), np.uint8); pos0 = np.zeros((E, 2), np.uint8)
    maps = np.zeros((T, E, H, W), np.uint8); poss = np.zeros((T, E, 2), np.uint8)
    heat = np.zeros((T, E, H, W), np.uint16)
    rew = np.zeros((T, E), np.float64); done = np.zeros((T, E), np.bool_)
    info = np.zeros((T, E, len(keys) + 2), np.int64)
    for i, env in enumerate(envs):
        o = env.reset()
        map0[i] = o["map"]
        if has_pos:
            pos0[i] = o["pos"]
        assert o["heatmap"].sum() == 0
    for t in range(T):
        for i, env in enumerate(envs):
            a = acts[t, i]
            o, r, d, inf = env.step(int(a[0]) if len(a) == 1 else [int(v) for v in a])
            rew[t, i] = r
            done[t, i] = d
            info[t, i] = [int(inf[k]) for k in keys] + [inf["iterations"], inf["changes"]]
            if d:
                o = env.reset()   # vector-env auto-reset semantics: terminal r/d/info, post-reset obs
            maps[t, i] = o["map"]
            if has_pos:
                poss[t, i] = o["pos"]
            assert o["heatmap"].max() < 65536 and (o["heatmap"] == np.floor(o["heatmap"])).all()
            heat[t, i] = o["heatmap"].astype(np.uint16)
    env = envs[0]
    cfg = np.array([W, H, env._max_changes, env._max_iterations, seed0, action_seed], dtype=np.int64)
    save("traj_" + name, prob=np.array(prob), rep=np.array(rep), calls=np.array(repr(list(calls))), cfg=cfg,
         actions=acts, map0=map0, pos0=pos0, maps=maps, pos=poss, heatmap=heat, reward=rew, done=done,
         info=info, info_keys=np.array(keys + ["iterations", "changes"]))
    print("    %s: %d env-steps, %d episodes ended, %.1fs" % (name, T * E, int(done.sum()), time.time() - t0))


TRAJS = [
    ("binary_narrow", "binary", "narrow", 32, 400, ()),
    ("binary_turtle", "binary", "turtle", 16, 300, ()),
    ("binary_wide", "binary", "wide", 16, 300, ()),
    ("binary_narrow_seq", "binary", "narrow", 8, 300, (dict(random_tile=False),)),
    ("binary_narrow_fixedstart", "binary", "narrow", 8, 300, (dict(random_start=False, random_probs=False),)),
    ("binary_turtle_warp", "binary", "turtle", 8, 300, (dict(warp=True, target_path=5),)),
    ("binary_narrow_9x21", "binary", "narrow", 8, 300, (dict(width=21, height=9), dict(change_percentage=0.3))),
    ("binary_turtle_64", "binary", "turtle", 4, 300, (dict(width=64, height=64),)),
    ("binary_turtle_64_cp", "binary", "turtle", 3, 240, (dict(width=64, height=64), dict(change_percentage=0.2))),
    ("binary_wide_40x33", "binary", "wide", 4, 120, (dict(width=40, height=33),)),
    ("zelda_wide_11x16", "zelda", "wide", 32, 300, (dict(width=11, height=16),)),
    ("zelda_wide", "zelda", "wide", 16, 300, ()),
    ("zelda_narrow", "zelda", "narrow", 16, 300, ()),
    ("zelda_turtle", "zelda", "turtle", 16, 300, (dict(change_percentage=0.6),)),
    ("zelda_wide_params", "zelda", "wide", 8, 200, (dict(max_enemies=3, target_enemy_dist=2, target_path=6,
                                                          rewards={"regions": 2.5, "path-length": 0.25}),)),
    ("sokoban_narrow", "sokoban", "narrow", 64, 400, ()),
    ("sokoban_wide", "sokoban", "wide", 16, 300, ()),
    ("sokoban_turtle", "sokoban", "turtle", 16, 300, (dict(change_percentage=0.8),)),
    ("sokoban_narrow_7x6", "sokoban", "narrow", 8, 300, (dict(width=7, height=6), dict(change_percentage=0.4, solver_power=400, max_crates=2))),
    # the 3x3 "cast" / "multi" representations
    ("binary_narrowcast", "binary", "narrowcast", 16, 300, ()),
    ("zelda_narrowcast_seq", "zelda", "narrowcast", 8, 300, (dict(random_tile=False, change_percentage=0.5),)),
    ("binary_narrowmulti", "binary", "narrowmulti", 16, 300, ()),
    ("sokoban_narrowmulti", "sokoban", "narrowmulti", 16, 300, (dict(change_percentage=0.8),)),
    ("binary_turtlecast", "binary", "turtlecast", 16, 300, ()),
    ("zelda_turtlecast_warp", "zelda", "turtlecast", 8, 300, (dict(warp=True, width=13, height=9), dict(change_percentage=0.5))),
    ("binary_turtlecast_64", "binary", "turtlecast", 3, 200, (dict(width=64, height=64),)),
    # the mdungeon problem (the open / small variants make the planner run in a large share of the steps)
    ("mdungeon_narrow", "mdungeon", "narrow", 32, 300, ()),
    ("mdungeon_wide_open", "mdungeon", "wide", 24, 300, (dict(probs={"empty": 0.82, "solid": 0.04, "player": 0.02, "exit": 0.02,
                                                                      "potion": 0.02, "treasure": 0.03, "goblin": 0.03, "ogre": 0.02},
                                                               change_percentage=0.6),)),
    ("mdungeon_turtle_5x6", "mdungeon", "turtle", 24, 400, (dict(width=5, height=6), dict(
        change_percentage=0.9, probs={"empty": 0.7, "solid": 0.1, "player": 0.04, "exit": 0.04}, target_solution=4,
        target_col_enemies=0.3, max_enemies=3, max_potions=1, max_treasures=1, solver_power=600,
        rewards={"dist-win": 0.3, "col-enemies": 1.5, "regions": 2}),)),
    ("ddave_narrow", "ddave", "narrow", 32, 300, ()),
    ("ddave_wide_open", "ddave", "wide", 24, 300, (dict(probs={"empty": 0.8, "solid": 0.1, "player": 0.025, "exit": 0.025, "diamond": 0.02,
                                                                "key": 0.025, "spike": 0.005}, change_percentage=0.6),)),
    ("ddave_turtle_6x5", "ddave", "turtle", 24, 400, (dict(width=6, height=5), dict(
        change_percentage=0.9, probs={"empty": 0.7, "solid": 0.12, "player": 0.05, "exit": 0.05, "key": 0.05, "spike": 0.01},
        target_solution=3, target_jumps=0, max_diamonds=1, min_spikes=2, solver_power=400,
        rewards={"dist-win": 0.3, "num-jumps": 1.5, "dist-floor": 0.5}),)),
    # smb (the planner runs on every changed map: small levels keep the reference affordable here)
    ("smb_narrow_24x8", "smb", "narrow", 8, 200, (dict(width=24, height=8), dict(change_percentage=0.3, probs={"empty": 0.5, "solid": 0.42}))),
    ("smb_wide_30x10", "smb", "wide", 6, 160, (dict(width=30, height=10), dict(change_percentage=0.2, min_empty=200, min_enemies=1,
                                                                               max_enemies=4, min_jumps=2, probs={"empty": 0.5, "solid": 0.3, "brick": 0.12},
                                                                               rewards={"noise": 1, "dist-win": 2.5}))),
    ("smb_turtle_20x7", "smb", "turtle", 4, 160, (dict(width=20, height=7), dict(change_percentage=0.4, probs={"empty": 0.6, "solid": 0.25}))),
    ("smb_narrow", "smb", "narrow", 3, 60, ()),
    ("smb_narrowcast_16x6", "smb", "narrowcast", 4, 120, (dict(width=16, height=6), dict(change_percentage=0.5, probs={"empty": 0.45, "solid": 0.5}))),
    # round 4: maps beyond 64 x 64 (bigmap.h) -- max_changes follows Q9 (the second adjust_param call sets it)
    ("binary_narrow_100x100", "binary", "narrow", 3, 90, (dict(width=100, height=100), dict(change_percentage=0.002))),
    ("binary_wide_70x65", "binary", "wide", 3, 60, (dict(width=70, height=65), dict(change_percentage=0.003))),
    ("zelda_turtle_66x20", "zelda", "turtle", 3, 120, (dict(width=66, height=20), dict(change_percentage=0.01))),
    ("binary_turtlecast_30x90", "binary", "turtlecast", 3, 80, (dict(width=30, height=90), dict(change_percentage=0.01))),
    # round 4: the search problems on levels beyond 256 cells, and a solver_power beyond 16 383 (search_big.h)
    ("sokoban_narrow_20x20", "sokoban", "narrow", 3, 100, (dict(width=20, height=20), dict(change_percentage=0.02, solver_power=600,
        probs={"empty": 0.9, "solid": 0.06, "player": 0.004, "crate": 0.004, "target": 0.004}))),
    ("mdungeon_turtle_18x22", "mdungeon", "turtle", 3, 100, (dict(width=18, height=22), dict(change_percentage=0.02, solver_power=500,
        probs={"empty": 0.86, "solid": 0.1, "player": 0.004, "exit": 0.004, "potion": 0.01, "treasure": 0.01, "goblin": 0.006, "ogre": 0.006}))),
    ("ddave_wide_24x16", "ddave", "wide", 3, 100, (dict(width=24, height=16), dict(change_percentage=0.02, solver_power=500,
        probs={"empty": 0.8, "solid": 0.18, "player": 0.004, "exit": 0.004, "diamond": 0.004, "key": 0.004, "spike": 0.004}))),
    ("sokoban_wide_p20000", "sokoban", "wide", 4, 60, (dict(solver_power=20000, change_percentage=0.9,
        probs={"empty": 0.8, "solid": 0.05, "player": 0.05, "crate": 0.05, "target": 0.05}),)),
    # round 5: the wide representation has no cursor, so no uint8 `pos` caps its map: sides beyond 255 (wide_rep.py:42-45, 67-70)
    ("binary_wide_300x40", "binary", "wide", 2, 36, (dict(width=300, height=40), dict(change_percentage=0.001))),
    ("zelda_wide_12x270", "zelda", "wide", 2, 36, (dict(width=12, height=270), dict(change_percentage=0.003))),
    ("mdungeon_narrow_monsters", "mdungeon", "narrow", 16, 300, (dict(width=6, height=6), dict(
        change_percentage=0.8, solver_power=250, target_solution=3, target_col_enemies=0.2,
        probs={"empty": 0.45, "solid": 0.03, "player": 0.03, "exit": 0.03, "potion": 0.06, "treasure": 0.05, "goblin": 0.1, "ogre": 0.25}),)),
]


# ----------------------------------------------------------------------------- wrappers
def gen_wrappers():
    """Observations of the two composite wrappers train.py uses (wrappers.py:215-248), per step, with the
    vector-env auto-reset; the flat ActionMap action index is recorded as the action."""
    from gym_pcgrl import wrappers
    cases = [
        ("cropped_binary_narrow", "binary-narrow-v0", "cropped", 28, 8, 200),
        ("cropped_zelda_narrow", "zelda-narrow-v0", "cropped", 22, 8, 200),
        ("cropped_sokoban_turtle", "sokoban-turtle-v0", "cropped", 10, 8, 200),
        ("cropped_binary_turtle_odd", "binary-turtle-v0", "cropped", 9, 4, 120),
        ("actionmap_zelda_wide", "zelda-wide-v0", "actionmap", 0, 8, 150),
        ("actionmap_binary_wide", "binary-wide-v0", "actionmap", 0, 8, 150),
        ("actionmap_sokoban_wide", "sokoban-wide-v0", "actionmap", 0, 8, 150),
        ("cropped_mdungeon_turtle", "mdungeon-turtle-v0", "cropped", 22, 8, 150),
        ("cropped_ddave_narrow", "ddave-narrow-v0", "cropped", 22, 8, 150),
        ("actionmap_ddave_wide", "ddave-wide-v0", "actionmap", 0, 8, 150),
    ]
    only = os.environ.get("PCGRL_GOLDEN_WRAP_ONLY")
    if only:
        cases = [c for c in cases if c[0] in only.split(",")]
    for name, game, kind, size, E, T in cases:
        envs = []
        for i in range(E):
            w = wrappers.CroppedImagePCGRLWrapper(game, size) if kind == "cropped" else wrappers.ActionMapImagePCGRLWrapper(game)
            wrappers.get_pcgrl_env(w).seed(2000 + i)
            envs.append(w)
        base = wrappers.get_pcgrl_env(envs[0])
        nt = base.get_num_tiles()
        H, W = base._prob._height, base._prob._width
        rs = np.random.RandomState(9)
        rep = game.split("-")[1]
        if kind == "actionmap":
            acts = rs.randint(0, H * W * nt, size=(T, E)).astype(np.int64)
        elif rep == "narrow":
            acts = rs.randint(0, nt + 1, size=(T, E)).astype(np.int64)
        else:
            acts = rs.randint(0, nt + 4, size=(T, E)).astype(np.int64)
        obs0 = np.stack([np.asarray(e.reset()) for e in envs])
        obs = np.zeros((T,) + obs0.shape, dtype=np.uint8)
        rew = np.zeros((T, E)); done = np.zeros((T, E), np.bool_)
        for t in range(T):
            for i, e in enumerate(envs):
                o, r, d, _ = e.step(int(acts[t, i]))
                rew[t, i] = r; done[t, i] = d
                if d:
                    o = e.reset()
                o = np.asarray(o)
                assert (o == o.astype(np.uint8)).all()
                obs[t, i] = o
        save("wrap_" + name, game=np.array(game), kind=np.array(kind), size=np.array(size), actions=acts,
             obs0=obs0.astype(np.uint8), obs=obs, reward=rew, done=done, seed0=np.array(2000))
        print("    %s obs %s" % (name, obs.shape))


def gen_heat_boundary():
    """The heat map beyond 32 767 (pcgrl_env.py:35,137: a float64 count per cell; the build keeps 16 bits).  binary-wide on a
    182 x 182 map that starts all solid, change_percentage 1.0 -> max_changes 33 124; ONE cell is rewritten with alternating tiles
    32 900 times, every write a change: its count passes 2^15.  Stored: the per-step reward / done / info of the first and last 200
    steps and of every 100th one, the final map's empty cells and the final heat map as (cell, count) pairs."""
    t0 = time.time()
    W = H = 182
    T = 32900
    env = gym.make("binary-wide-v0")
    env.seed(7)
    env.adjust_param(width=W, height=H, probs={"empty": 0.0, "solid": 1.0})
    env.adjust_param(change_percentage=1.0)
    obs = env.reset()
    assert int(obs["map"].sum()) == W * H
    keep = sorted(set(list(range(200)) + list(range(0, T, 100)) + list(range(T - 200, T))))
    rew, done, info = [], [], []
    for t in range(T):
        obs, r, d, inf = env.step([5, 7, t % 2])
        assert not d
        if t in keep:
            rew.append(r); done.append(d); info.append([inf[k] for k in INFO_KEYS["binary"]] + [inf["iterations"], inf["changes"]])
    heat = np.asarray(obs["heatmap"])
    cells = np.argwhere(heat != 0)
    save("heat_boundary", cfg=np.array([W, H, env._max_changes, env._max_iterations, 7, T], np.int64), cell=np.array([5, 7, 0], np.int64),
         steps=np.array(keep, np.int64), reward=np.array(rew, np.float64), done=np.array(done, np.bool_), info=np.array(info, np.int64),
         heat_cells=cells.astype(np.int64), heat_counts=np.array([heat[y, x] for y, x in cells], np.int64),
         empty_cells=np.argwhere(np.asarray(obs["map"]) == 0).astype(np.int64))
    print("heat_boundary", heat.max(), "%.0f s" % (time.time() - t0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    jobs = {
        "rng": gen_rng, "stats_binary": gen_stats_binary, "stats_zelda": gen_stats_zelda,
        "stats_sokoban": gen_stats_sokoban, "stats_mdungeon": gen_stats_mdungeon, "stats_ddave": gen_stats_ddave, "stats_smb": gen_stats_smb, "range_reward": gen_range_reward, "adjust_param": gen_adjust_param,
        "wrappers": gen_wrappers, "stats_big": gen_stats_big, "stats_big_search": gen_stats_big_search, "stats_huge_search": gen_stats_huge_search, "heat_boundary": gen_heat_boundary,
    }
    for k, fn in jobs.items():
        if a.only in (None, k):
            print(k)
            fn()
    if a.only in (None, "traj") or (a.only or "").startswith("traj_"):
        print("traj")
        for name, prob, rep, E, T, calls in TRAJS:
            if a.only in (None, "traj", "traj_" + name):
                gen_traj(name, prob, rep, E, T, calls)


if __name__ == "__main__":
    main()
