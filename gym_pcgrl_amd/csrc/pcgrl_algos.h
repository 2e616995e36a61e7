// Level statistics as row-bitboard programs over a "lane group".
//
// A map row is one bit mask per tile class; lane r of a group holds row r.  One 4-neighbour
// flood step is  f' = (f | f<<1 | f>>1 | up(f) | down(f)) & passable, where up/down move a value
// one lane.  BFS distance to a cell = index of the step at which its bit turns on, so none of the
// reference's per-cell distance arrays are needed:
//
//   helper.py:197-207 calc_num_regions   -> count_regions()
//   helper.py:222-237 run_dikjstra       -> bfs_levels() / bfs_dist()
//   helper.py:250-264 calc_longest_path  -> longest_path() (per component: sweep from the first
//                                           cell in row-major order, np.argmax == first bit of the
//                                           last frontier, second sweep, max over components)
//   helper.py:16-23,272-273 tile histograms -> popcounts of class masks
//   binary_prob.py:81-86, zelda_prob.py:80-112, sokoban_prob.py:133-145 -> *_stats()
//   helper.py:366-376, *_prob.get_reward/get_episode_over -> range_reward(), compute_reward(), episode_over()
//
// The code is a template over a backend `B` that supplies the cross-lane primitives:
//   B::mask_t            per-lane row mask (device: uint32_t / uint64_t)
//   up(m), down(m)       m of the row above / below (0 outside the map)
//   any(m)               group-uniform: is any lane's m non-zero
//   any_ne(a, b)         group-uniform: a != b in any lane
//   first_bit(m)         m with only its first set bit in row-major order kept (whole group)
//   popcount_sum(m)      group-uniform total popcount
// On the GPU the backend is DevGroup (lanegroup_dev.h: DPP row shifts + ballot); tests instantiate
// the same templates with a CPU lane-group simulator.
#pragma once
#include "pcgrl_common.h"

struct PcgrlParams {
    int32_t prob, rep, num_envs, width, height, ntiles, nplanes, group, mask_bytes;
    int32_t max_changes, max_iterations;
    int32_t random_start, random_tile, warp, random_probs, auto_reset;
    int32_t target_path, max_enemies, target_enemy_dist, max_crates, target_solution, solver_power;
    int32_t prob_width, prob_height;   // the Problem's own width/height (zelda_prob.py:99, sokoban_prob.py:140)
    int32_t pad_;
    double rewards[PCGRL_MAX_STATS];
    double cdf[PCGRL_MAX_TILES];
};

// ---------------------------------------------------------------- scalar reward logic
// helper.py:366-376 (bounds may be +-inf; the five cases are exhaustive for finite values)
PCGRL_HD double range_reward(double nv, double ov, double lo, double hi) {
    if (nv >= lo && nv <= hi && ov >= lo && ov <= hi) return 0.0;
    if (ov <= hi && nv <= hi) return (nv < lo ? nv : lo) - (ov < lo ? ov : lo);
    if (ov >= lo && nv >= lo) return (ov > hi ? ov : hi) - (nv > hi ? nv : hi);
    if (nv > hi && ov < lo) return hi - nv + ov - lo;
    return hi - ov + nv - lo;
}
#if defined(__HIPCC__)
#define PCGRL_INF (__builtin_huge_val())
#else
#define PCGRL_INF (__builtin_huge_val())
#endif

// binary_prob.py:98-106 | zelda_prob.py:124-142 | sokoban_prob.py:157-175 (same summation order)
PCGRL_HD double compute_reward(const PcgrlParams& P, const int32_t* n, const int32_t* o) {
    const double* w = P.rewards;
    if (P.prob == PCGRL_PROB_BINARY) {
        return range_reward(n[0], o[0], 1, 1) * w[0] + range_reward(n[1], o[1], PCGRL_INF, PCGRL_INF) * w[1];
    } else if (P.prob == PCGRL_PROB_ZELDA) {
        double r = range_reward(n[0], o[0], 1, 1) * w[0];
        r = r + range_reward(n[1], o[1], 1, 1) * w[1];
        r = r + range_reward(n[2], o[2], 1, 1) * w[2];
        r = r + range_reward(n[3], o[3], 2, P.max_enemies) * w[4];
        r = r + range_reward(n[4], o[4], 1, 1) * w[3];
        r = r + range_reward(n[5], o[5], P.target_enemy_dist, PCGRL_INF) * w[5];
        r = r + range_reward(n[6], o[6], PCGRL_INF, PCGRL_INF) * w[6];
        return r;
    } else {
        int nr = n[1] - n[2], orr = o[1] - o[2];
        nr = nr < 0 ? -nr : nr; orr = orr < 0 ? -orr : orr;
        double r = range_reward(n[0], o[0], 1, 1) * w[0];
        r = r + range_reward(n[1], o[1], 1, P.max_crates) * w[1];
        r = r + range_reward(n[2], o[2], 1, P.max_crates) * w[2];
        r = r + range_reward(n[3], o[3], 1, 1) * w[3];
        r = r + range_reward(nr, orr, -PCGRL_INF, -PCGRL_INF) * w[4];
        r = r + range_reward(n[4], o[4], -PCGRL_INF, -PCGRL_INF) * w[5];
        r = r + range_reward(n[5], o[5], PCGRL_INF, PCGRL_INF) * w[6];
        return r;
    }
}
// binary_prob.py:119-120 | zelda_prob.py:155-156 | sokoban_prob.py:188-189
PCGRL_HD bool episode_over(const PcgrlParams& P, const int32_t* n, const int32_t* start) {
    if (P.prob == PCGRL_PROB_BINARY) return n[0] == 1 && n[1] - start[1] >= P.target_path;
    if (P.prob == PCGRL_PROB_ZELDA) return n[5] >= P.target_enemy_dist && n[6] >= P.target_path;
    return n[5] >= P.target_solution;
}
PCGRL_HD int num_stats(int prob) { return prob == PCGRL_PROB_BINARY ? 2 : (prob == PCGRL_PROB_ZELDA ? 7 : 6); }

// ---------------------------------------------------------------- bitboard programs
template <class B>
PCGRL_D typename B::mask_t pcg_expand(B& g, typename B::mask_t f) {
    return f | (f << 1) | (f >> 1) | g.up(f) | g.down(f);
}

// BFS from `src` through `pass` until nothing new is reached.  Returns the number of levels
// (eccentricity of src); `reached` = component, `last` = cells at maximum distance.
template <class B>
PCGRL_D int bfs_levels(B& g, typename B::mask_t src, typename B::mask_t pass,
                       typename B::mask_t& reached, typename B::mask_t& last) {
    typedef typename B::mask_t M;
    M f = src, prev = src ^ src;
    int t = 0;
    for (;;) {
        M n = pcg_expand(g, f) & pass;
        if (!g.any_ne(n, f)) break;
        prev = f;
        f = n;
        ++t;
    }
    reached = f;
    last = f & ~prev;
    return t;
}

// Distance from `src` to the nearest cell of `dst` (dst not containing src) through `pass`;
// -1 when no cell of dst is reachable (run_dikjstra leaves -1 there).
template <class B>
PCGRL_D int bfs_dist(B& g, typename B::mask_t src, typename B::mask_t dst, typename B::mask_t pass) {
    typedef typename B::mask_t M;
    M f = src;
    int t = 0;
    for (;;) {
        M n = pcg_expand(g, f) & pass;
        M fresh = n & ~f;
        if (!g.any(fresh)) return -1;
        ++t;
        if (g.any(fresh & dst)) return t;
        f = n;
    }
}

// helper.py:197-207: number of 4-connected components of `pass`.
template <class B>
PCGRL_D int count_regions(B& g, typename B::mask_t pass) {
    typedef typename B::mask_t M;
    // single-cell components need no flood
    M nb = (pass << 1) | (pass >> 1) | g.up(pass) | g.down(pass);
    M iso = pass & ~nb;
    int regions = g.popcount_sum(iso);
    M unvis = pass & ~iso;
    while (g.any(unvis)) {
        M f = g.first_bit(unvis);
        for (;;) {
            M n = pcg_expand(g, f) & unvis;
            if (!g.any_ne(n, f)) break;
            f = n;
        }
        unvis = unvis & ~f;
        ++regions;
    }
    return regions;
}

// binary_prob.py:81-86: regions + helper.py:250-264 double-sweep longest path, one pass over
// the components in row-major order of their first cell.
template <class B>
PCGRL_D void regions_and_longest_path(B& g, typename B::mask_t pass, int& regions, int& path) {
    typedef typename B::mask_t M;
    M nb = (pass << 1) | (pass >> 1) | g.up(pass) | g.down(pass);
    M iso = pass & ~nb;                 // isolated cells: one region each, path 0
    regions = g.popcount_sum(iso);
    path = 0;
    M unvis = pass & ~iso;
    while (g.any(unvis)) {
        M src = g.first_bit(unvis);
        M comp, last, tmp0, tmp1;
        bfs_levels(g, src, unvis, comp, last);
        M far = g.first_bit(last);      // np.argmax: first maximum in row-major order
        int ecc = bfs_levels(g, far, comp, tmp0, tmp1);
        path = ecc > path ? ecc : path;
        unvis = unvis & ~comp;
        ++regions;
    }
}

// Row masks of each tile class from the bit planes of the tile id (plane b = bit b of the id).
template <class M>
struct ZeldaMasks {
    M empty, solid, player, key, door, enemy;
};
template <class M>
PCGRL_D ZeldaMasks<M> zelda_masks(M b0, M b1, M b2, M valid) {
    // ids: 0 empty 1 solid 2 player 3 key 4 door 5 bat 6 scorpion 7 spider (zelda_prob.py:45-46)
    ZeldaMasks<M> z;
    z.empty = ~b2 & ~b1 & ~b0 & valid;
    z.solid = ~b2 & ~b1 & b0 & valid;
    z.player = ~b2 & b1 & ~b0 & valid;
    z.key = ~b2 & b1 & b0 & valid;
    z.door = b2 & ~b1 & ~b0 & valid;
    z.enemy = b2 & (b1 | b0) & valid;
    return z;
}

// zelda_prob.py:80-112.  out: player,key,door,enemies,regions,nearest-enemy,path-length
template <class B>
PCGRL_D void zelda_stats(B& g, const PcgrlParams& P, typename B::mask_t b0, typename B::mask_t b1,
                         typename B::mask_t b2, typename B::mask_t valid, int32_t* out) {
    typedef typename B::mask_t M;
    ZeldaMasks<M> z = zelda_masks(b0, b1, b2, valid);
    int player = g.popcount_sum(z.player), key = g.popcount_sum(z.key), door = g.popcount_sum(z.door);
    int enemies = g.popcount_sum(z.enemy);
    M walk = z.empty | z.player | z.key | z.enemy;          // regions / player->key passable set
    int regions = count_regions(g, walk);
    int nearest = 0, path = 0;
    if (player == 1 && regions == 1) {
        if (enemies > 0) {
            int d = bfs_dist(g, z.player, z.enemy, z.empty | z.player | z.enemy);   // key is NOT passable here
            nearest = d > 0 ? d : P.prob_width * P.prob_height;
        }
        if (key == 1 && door == 1) {
            path = bfs_dist(g, z.player, z.key, walk);
            path += bfs_dist(g, z.key, z.door, walk | z.door);  // -1 when the door is walled off
        }
    }
    out[0] = player; out[1] = key; out[2] = door; out[3] = enemies; out[4] = regions; out[5] = nearest; out[6] = path;
}

// sokoban_prob.py:133-145 without the solver.  out: player,crate,target,regions,dist-win(default),sol-length(0).
// Returns true when the solver precondition (sokoban_prob.py:143) holds.
template <class B>
PCGRL_D bool sokoban_stats(B& g, const PcgrlParams& P, typename B::mask_t b0, typename B::mask_t b1,
                           typename B::mask_t b2, typename B::mask_t valid, int32_t* out) {
    typedef typename B::mask_t M;
    // ids: 0 empty 1 solid 2 player 3 crate 4 target (sokoban_prob.py:44-45)
    M solid = ~b2 & ~b1 & b0 & valid;
    M player = ~b2 & b1 & ~b0 & valid;
    M crate = ~b2 & b1 & b0 & valid;
    M target = b2 & ~b1 & ~b0 & valid;
    int np_ = g.popcount_sum(player), nc = g.popcount_sum(crate), nt = g.popcount_sum(target);
    int regions = count_regions(g, valid & ~solid);
    out[0] = np_; out[1] = nc; out[2] = nt; out[3] = regions;
    out[4] = P.prob_width * P.prob_height * (P.prob_width + P.prob_height);
    out[5] = 0;
    return np_ == 1 && nc == nt && nc > 0 && regions == 1;
}
