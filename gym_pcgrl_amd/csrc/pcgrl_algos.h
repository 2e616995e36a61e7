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
//   B::mask_t / ivec_t   per-lane row mask (device: uint32_t / uint64_t) / per-lane int
//   up(m), down(m)       m of the row above / below (0 outside the map)
//   any(m)               group-uniform: is any lane's m non-zero
//   wave_any(m)          true if m != 0 in any lane of the wavefront (>= any(m); only used where an
//                        extra loop round is harmless)
//   first_bit(m)         m with only its first set bit in row-major order kept (whole group)
//   popcount_sum(m)      group-uniform total popcount;  popc_lanes(m) per-lane popcount;  popcount_sum2 / 3: several counts at once
//   imax(v)              group-uniform max of a per-lane int
//   isel_ne / msel_ne    per lane: a != b ? x : y;   keep_where_eq(v, x, m): v == x ? m : 0
//   bitrev(m)            reverse the bits of the mask word
//   rows_down(m, k) / rows_up(m, k)   m of the row 2^k above / below, k < kLog2Group; may be confined
//                        to aligned blocks of 16 rows (0 across a block edge)
// On the GPU the backend is DevGroup (lanegroup_dev.h: DPP row shifts + ballot); tests instantiate
// the same templates with a CPU lane-group simulator.
#pragma once
#include "pcgrl_common.h"

// Cost-model hook for the CPU lane-group simulator (tests/hostsim); compiles to nothing in the product.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define PCGRL_NO_IFCVT() asm volatile("" ::: "memory")
#else
#define PCGRL_NO_IFCVT() do {} while (0)
#endif
#ifndef PCGRL_TRACE
#define PCGRL_TRACE(g, site)
#endif

struct PcgrlParams {
    int32_t prob, rep, num_envs, width, height, ntiles, nplanes, group, mask_bytes;
    int32_t max_changes, max_iterations;
    int32_t random_start, random_tile, warp, random_probs, auto_reset;
    int32_t target_path, max_enemies, target_enemy_dist, max_crates, target_solution, solver_power;
    int32_t prob_width, prob_height;   // the Problem's own width/height (zelda_prob.py:99, sokoban_prob.py:140)
    int32_t max_potions, max_treasures;   // mdungeon_prob.py:25-26
    int32_t max_diamonds, min_spikes, target_jumps;   // ddave_prob.py:23-27
    int32_t min_empty, min_enemies, min_jumps;   // smb_prob.py:21-24
    int32_t big, big_search;   // the map is beyond the row-bitboard kernels (bigmap.h) / the level or solver_power beyond the compact searches (search_big.h)
    double target_col_enemies;         // mdungeon_prob.py:28
    double rewards[PCGRL_MAX_REWARDS];
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
#define PCGRL_INF (__builtin_huge_val())

// The stats are small integers and every band bound is an integer or +-inf, so get_range_reward is
// evaluated in integer arithmetic (INT_MAX / INT_MIN stand for +-inf: the comparisons and min/max then
// behave exactly like the float version, and the two "crossing" cases cannot be reached with an infinite
// bound).  Checked against the reference's table in tests (range_reward.npz).
#define PCGRL_IPOS 2147483647
#define PCGRL_INEG (-2147483647 - 1)
PCGRL_HD int range_reward_i(int nv, int ov, int lo, int hi) {
    if (nv >= lo && nv <= hi && ov >= lo && ov <= hi) return 0;
    if (ov <= hi && nv <= hi) return (nv < lo ? nv : lo) - (ov < lo ? ov : lo);
    if (ov >= lo && nv >= lo) return (ov > hi ? ov : hi) - (nv > hi ? nv : hi);
    if (nv > hi && ov < lo) return hi - nv + ov - lo;
    return hi - ov + nv - lo;
}

// The mdungeon statistics row has eleven values in the reference (mdungeon_prob.py:139-157); the per-environment rows of
// this library have eight 32-bit slots, so the planner's five results share two of them:
//   s[0..5] = player, exit, potions, treasures, enemies, regions
//   s[6]    = sol-length if the planner won, else dist-win
//   s[7]    = col-potions | col-treasures << 8 | col-enemies << 16 | won << 24    (a solvable level has < 256 things)
// (dist-win is 0 when the planner won and sol-length is 0 when it did not, mdungeon_prob.py:112-126.)
PCGRL_HD int md_won(const int32_t* s) { return (s[7] >> 24) & 1; }
PCGRL_HD int md_dist_win(const int32_t* s) { return md_won(s) ? 0 : s[6]; }
PCGRL_HD int md_sol_length(const int32_t* s) { return md_won(s) ? s[6] : 0; }
PCGRL_HD int md_col_potions(const int32_t* s) { return s[7] & 255; }
PCGRL_HD int md_col_treasures(const int32_t* s) { return (s[7] >> 8) & 255; }
PCGRL_HD int md_col_enemies(const int32_t* s) { return (s[7] >> 16) & 255; }
PCGRL_HD void md_pack(int32_t* s, const int* out5) {   // out5 = dist-win, sol-length, col-potions, col-treasures, col-enemies
    const int won = out5[1] > 0 ? 1 : 0;              // the exit is never under the player at depth 0
    s[6] = won ? out5[1] : out5[0];
    s[7] = (out5[2] & 255) | ((out5[3] & 255) << 8) | ((out5[4] & 255) << 16) | (won << 24);
}

// The ddave row (ddave_prob.py:141-161 has eleven values too; a level the planner accepts has fewer than 256 cells):
//   s[0] = player | exit << 8 | key << 16, s[1] = dist-floor, s[2] = diamonds, s[3] = spikes, s[4] = regions,
//   s[5] = num-jumps, s[6] = sol-length if the planner won else dist-win, s[7] = col-diamonds | won << 24
PCGRL_HD int dd_player(const int32_t* s) { return s[0] & 255; }
PCGRL_HD int dd_exit(const int32_t* s) { return (s[0] >> 8) & 255; }
PCGRL_HD int dd_key(const int32_t* s) { return (s[0] >> 16) & 255; }
PCGRL_HD void dd_pack(int32_t* s, const int* out4) {   // out4 = dist-win, sol-length, num-jumps, col-diamonds
    const int won = out4[1] > 0 ? 1 : 0;              // the player never starts on the exit holding the key
    s[5] = out4[2];
    s[6] = won ? out4[1] : out4[0];
    s[7] = (out4[3] & 255) | (won << 24);
}

// binary_prob.py:98-106 | zelda_prob.py:124-142 | sokoban_prob.py:157-175 (same summation order; the
// products and sums are done in fp64 exactly as Python does them with int * weight)
PCGRL_HD double compute_reward(const PcgrlParams& P, const int32_t* n, const int32_t* o, int prob) {
#ifdef PCGRL_EXP_NOREWARD     /* timing experiment (tools/exp_build_bench.py): what the reward arithmetic costs; results are wrong */
    return (double)(n[0] - o[0]);
#endif
    const double* w = P.rewards;
    if (prob == PCGRL_PROB_BINARY) {
        // (a band at +inf -- "the longer the better" -- is the second case of get_range_reward for every finite value: new - old)
        return (double)range_reward_i(n[0], o[0], 1, 1) * w[0] + (double)(n[1] - o[1]) * w[1];
    } else if (prob == PCGRL_PROB_ZELDA) {
        double r = (double)range_reward_i(n[0], o[0], 1, 1) * w[0];
        r = r + (double)range_reward_i(n[1], o[1], 1, 1) * w[1];
        r = r + (double)range_reward_i(n[2], o[2], 1, 1) * w[2];
        r = r + (double)range_reward_i(n[3], o[3], 2, P.max_enemies) * w[4];
        r = r + (double)range_reward_i(n[4], o[4], 1, 1) * w[3];
        r = r + (double)range_reward_i(n[5], o[5], P.target_enemy_dist, PCGRL_IPOS) * w[5];
        r = r + (double)range_reward_i(n[6], o[6], PCGRL_IPOS, PCGRL_IPOS) * w[6];
        return r;
    } else if (prob == PCGRL_PROB_DDAVE) {
        // weights in the order of DDaveProblem._rewards, summed in the order of get_reward (ddave_prob.py:187-209);
        // dist-win / sol-length / won share slots the way mdungeon's do
        double r = (double)range_reward_i(dd_player(n), dd_player(o), 1, 1) * w[0];
        r = r + (double)range_reward_i(n[1], o[1], 0, 0) * w[1];
        r = r + (double)range_reward_i(dd_exit(n), dd_exit(o), 1, 1) * w[2];
        r = r + (double)range_reward_i(n[3], o[3], P.min_spikes, PCGRL_IPOS) * w[5];
        r = r + (double)range_reward_i(n[2], o[2], PCGRL_INEG, P.max_diamonds) * w[3];
        r = r + (double)range_reward_i(dd_key(n), dd_key(o), 1, 1) * w[4];
        r = r + (double)range_reward_i(n[4], o[4], 1, 1) * w[6];
        r = r + (double)range_reward_i(n[5], o[5], PCGRL_IPOS, PCGRL_IPOS) * w[7];
        r = r + (double)range_reward_i(md_dist_win(n), md_dist_win(o), PCGRL_INEG, PCGRL_INEG) * w[8];
        r = r + (double)range_reward_i(md_sol_length(n), md_sol_length(o), PCGRL_IPOS, PCGRL_IPOS) * w[9];
        return r;
    } else if (prob == PCGRL_PROB_SMB) {
        // weights and summation in the order of SMBProblem._rewards / get_reward (smb_prob.py:26-35, 169-189)
        double r = (double)range_reward_i(n[0], o[0], 0, 0) * w[0];
        r = r + (double)range_reward_i(n[1], o[1], 0, 0) * w[1];
        r = r + (double)range_reward_i(n[2], o[2], P.min_enemies, P.max_enemies) * w[2];
        r = r + (double)range_reward_i(n[3], o[3], P.min_empty, PCGRL_IPOS) * w[3];
        r = r + (double)range_reward_i(n[4], o[4], 0, 0) * w[4];
        r = r + (double)range_reward_i(n[5], o[5], P.min_jumps, PCGRL_IPOS) * w[5];
        r = r + (double)range_reward_i(n[6], o[6], 0, 0) * w[6];
        r = r + (double)range_reward_i(n[7], o[7], 0, 0) * w[7];
        return r;
    } else if (prob == PCGRL_PROB_MDUNGEON) {
        // weights in the order of MDungeonProblem._rewards, summed in the order of get_reward (mdungeon_prob.py:183-206)
        double r = (double)range_reward_i(n[0], o[0], 1, 1) * w[0];
        r = r + (double)range_reward_i(n[1], o[1], 1, 1) * w[1];
        r = r + (double)range_reward_i(n[4], o[4], 1, P.max_enemies) * w[4];
        r = r + (double)range_reward_i(n[3], o[3], PCGRL_INEG, P.max_treasures) * w[3];
        r = r + (double)range_reward_i(n[2], o[2], PCGRL_INEG, P.max_potions) * w[2];
        r = r + (double)range_reward_i(n[5], o[5], 1, 1) * w[5];
        r = r + (double)range_reward_i(md_col_enemies(n), md_col_enemies(o), PCGRL_IPOS, PCGRL_IPOS) * w[6];
        r = r + (double)range_reward_i(md_dist_win(n), md_dist_win(o), PCGRL_INEG, PCGRL_INEG) * w[7];
        r = r + (double)range_reward_i(md_sol_length(n), md_sol_length(o), PCGRL_IPOS, PCGRL_IPOS) * w[8];
        return r;
    } else {
        int nr = n[1] - n[2], orr = o[1] - o[2];
        nr = nr < 0 ? -nr : nr; orr = orr < 0 ? -orr : orr;
        double r = (double)range_reward_i(n[0], o[0], 1, 1) * w[0];
        r = r + (double)range_reward_i(n[1], o[1], 1, P.max_crates) * w[1];
        r = r + (double)range_reward_i(n[2], o[2], 1, P.max_crates) * w[2];
        r = r + (double)range_reward_i(n[3], o[3], 1, 1) * w[3];
        r = r + (double)range_reward_i(nr, orr, PCGRL_INEG, PCGRL_INEG) * w[4];
        r = r + (double)range_reward_i(n[4], o[4], PCGRL_INEG, PCGRL_INEG) * w[5];
        r = r + (double)range_reward_i(n[5], o[5], PCGRL_IPOS, PCGRL_IPOS) * w[6];
        return r;
    }
}
PCGRL_HD double compute_reward(const PcgrlParams& P, const int32_t* n, const int32_t* o) { return compute_reward(P, n, o, P.prob); }
// binary_prob.py:119-120 | zelda_prob.py:155-156 | sokoban_prob.py:188-189
PCGRL_HD bool episode_over(const PcgrlParams& P, const int32_t* n, const int32_t* start, int prob) {
    if (prob == PCGRL_PROB_BINARY) return n[0] == 1 && n[1] - start[1] >= P.target_path;
    if (prob == PCGRL_PROB_ZELDA) return n[5] >= P.target_enemy_dist && n[6] >= P.target_path;
    if (prob == PCGRL_PROB_DDAVE) return md_sol_length(n) >= P.target_solution && n[5] > P.target_jumps;   // ddave_prob.py:218-220
    if (prob == PCGRL_PROB_SMB) return n[7] <= 0;                                                            // smb_prob.py:191-192
    if (prob == PCGRL_PROB_MDUNGEON) {   // mdungeon_prob.py:219-222 (true division, compared in fp64)
        const int en = n[4] > 1 ? n[4] : 1;
        return md_sol_length(n) >= P.target_solution && n[4] > 0 && (double)md_col_enemies(n) / (double)en > P.target_col_enemies;
    }
    return n[5] >= P.target_solution;
}
PCGRL_HD bool episode_over(const PcgrlParams& P, const int32_t* n, const int32_t* start) { return episode_over(P, n, start, P.prob); }
PCGRL_HD int num_stats(int prob) { return prob == PCGRL_PROB_BINARY ? 2 : (prob == PCGRL_PROB_ZELDA ? 7 : (prob == PCGRL_PROB_SOKOBAN ? 6 : 8)); }   // mdungeon, ddave: packed rows

// ---------------------------------------------------------------- bitboard programs
//
// Work-saving structure (results are order-independent, so none of this changes an answer):
//   * components of 1, 2 or 3 cells are recognised in closed form from per-cell neighbour counts
//     (their region count and path length 0/1/2 need no flood at all);
//   * a component is extracted with a *run fill*: a whole horizontal run is filled in O(1) with the
//     carry chain of an integer add ((p + s) ^ p) & p, and its bit-reversed twin for the other
//     direction), so extraction takes a few vertical hops instead of one step per BFS level;
//   * the exact double BFS sweep only runs for a component that can still raise the maximum
//     (a k-cell component cannot contain a shortest path longer than k-1);
//   * the first component visited is the one through the fullest row, which is almost always the
//     giant one, so the four maps that share a wavefront do their long sweeps at the same time;
//   * BFS levels are kept per lane (iteration of the lane's last change) and reduced once per
//     sweep, so the sweep loop needs a single wave-uniform exit test and no per-group ballot.
template <class B>
PCGRL_D typename B::mask_t pcg_expand(B& g, typename B::mask_t f) {
    return f | (f << 1) | (f >> 1) | g.up(f) | g.down(f);
}
template <class B>
PCGRL_D typename B::mask_t pcg_neighbours(B& g, typename B::mask_t f) {
    return (f << 1) | (f >> 1) | g.up(f) | g.down(f);
}

// Fill every horizontal run of `p` that contains a seed bit of `s` (s subset of p).  rp = bitrev(p).
template <class B>
PCGRL_D typename B::mask_t pcg_fill_rows(B& g, typename B::mask_t s, typename B::mask_t p, typename B::mask_t rp) {
    typedef typename B::mask_t M;
    M hi = (((p + s) ^ p) & p) | s;              // carry chain: seed .. top of its run
    M rs = g.bitrev(s);
    M lo = (((rp + rs) ^ rp) & rp) | rs;         // same in the mirrored word: seed .. bottom of its run
    return hi | g.bitrev(lo);
}

// Per-map constants of the fills: mirrored passable mask and the Kogge-Stone "propagate" masks of
// the vertical direction (pro_k[r] = passable in rows r, r-1, .., r-(2^k - 1), and the same upwards).
template <class B>
struct PcgFillCtx {
    typename B::mask_t pass, rpass;
    typename B::mask_t dn[B::kLog2Group], up[B::kLog2Group];
};
template <class B>
PCGRL_D PcgFillCtx<B> pcg_fill_ctx(B& g, typename B::mask_t pass) {
    PcgFillCtx<B> c;
    c.pass = pass;
    c.rpass = g.bitrev(pass);
    c.dn[0] = pass; c.up[0] = pass;
#pragma unroll
    for (int k = 1; k < B::kLog2Group; k++) {
        c.dn[k] = c.dn[k - 1] & g.rows_down(c.dn[k - 1], k - 1);
        c.up[k] = c.up[k - 1] & g.rows_up(c.up[k - 1], k - 1);
    }
    return c;
}
// Fill every vertical run of passable cells that contains a bit of `f` (log2(rows) doubling steps).
template <class B>
PCGRL_D typename B::mask_t pcg_fill_cols(B& g, typename B::mask_t f, const PcgFillCtx<B>& c) {
    typename B::mask_t d = f, u = f;
#pragma unroll
    for (int k = 0; k < B::kLog2Group; k++) {
        d = d | (c.dn[k] & g.rows_down(d, k));    // rows_down(x, k): row r receives row r - 2^k
        u = u | (c.up[k] & g.rows_up(u, k));
    }
    // With more than 16 rows the doubling steps are confined to blocks of 16 rows (one DPP row); one plain
    // row hop then lets the next round carry the fill across a block boundary.
    typename B::mask_t r = d | u;
    if (B::kGroup > 16) r = r | (c.pass & (g.up(r) | g.down(r)));
    return r;
}

// The run / column fills of pcg_component from a connected start set (also on its own: bigmap.h takes the plain flood steps first and
// makes the fill constants only for a component that is still growing after them).
template <class B>
PCGRL_D typename B::mask_t pcg_component_fills(B& g, typename B::mask_t seed, const PcgFillCtx<B>& c) {
    typedef typename B::mask_t M;
    M f = pcg_fill_rows(g, seed, c.pass, c.rpass);
    for (;;) {
        M n = pcg_fill_cols(g, f, c);
        f = pcg_fill_rows(g, n, c.pass, c.rpass);
        // done when no passable cell borders the set (it grew from one seed, so it is then exactly the seed's component): a
        // ten-instruction test instead of one more round of fills that finds nothing to add.  (An extra round is harmless: a
        // group that is done idles while another one of the wavefront still grows.)
        if (!g.wave_any(pcg_neighbours(g, f) & c.pass & ~f)) break;
    }
    return f;
}

// The 4-connected component containing `seed`: alternate full-column and full-row fills until stable
// (one round per "turn" of the most winding path instead of one step per cell).
template <class B>
PCGRL_D typename B::mask_t pcg_component(B& g, typename B::mask_t seed, const PcgFillCtx<B>& c) {
    typedef typename B::mask_t M;
    PCGRL_TRACE(g, 1);
    if (B::kGroup != 16) {
        // Whole-wavefront groups (tall maps): a fill round costs ~100 instructions there (64-bit masks, the extra
        // block hop) and most components of a large random map are small, so plain flood steps go first -- two per
        // exit test, eight at most -- and the run/column fills only take over for what is still growing.
        M f = seed;
        for (int i = 0; i < 4; i++) {
            M n = pcg_expand(g, f) & c.pass;
            n = pcg_expand(g, n) & c.pass;
            if (!g.wave_any(n ^ f)) return f;
            f = n;
        }
        seed = f;
    }
    return pcg_component_fills(g, seed, c);
}

// Components with at most 3 cells, from neighbour counts.  Returns their union; adds their number to
// `regions` and raises `path` to their longest path (1 for a 2-cell, 2 for a 3-cell component).
template <class B>
PCGRL_D typename B::mask_t pcg_tiny_components(B& g, typename B::mask_t p, int& regions, int& path) {
    typedef typename B::mask_t M;
    const M a = (p << 1) & p, b = (p >> 1) & p, c = g.up(p) & p, d = g.down(p) & p;   // has left/right/up/down neighbour
    const M s0 = a ^ b, c0 = a & b, s1 = c ^ d, c1 = c & d;
    const M n0 = s0 ^ s1, k = s0 & s1;            // count = n0 + 2*(c0 + c1 + k), at most two of c0,c1,k set
    const M two_or_more = c0 | c1 | k;
    const M iso = p & ~(a | b | c | d);
    const M deg1 = n0 & ~two_or_more;
    const M deg2 = ~n0 & (c0 ^ c1 ^ k) & ~(c0 & c1);
    // 2-cell components: a degree-1 cell next to a degree-1 cell
    const M dom = deg1 & pcg_neighbours(g, deg1);
    // 3-cell components: a degree-2 centre whose two neighbours both have degree 1, plus those two
    const M e1 = deg1 << 1, e2 = deg1 >> 1, e3 = g.up(deg1), e4 = g.down(deg1);
    const M centre = deg2 & ((e1 & e2) | (e3 & e4) | ((e1 | e2) & (e3 | e4)));
    const M ends = deg1 & pcg_neighbours(g, centre);
    int n_iso, n_dom, n_tri;
    g.popcount_sum3(iso, dom, centre, n_iso, n_dom, n_tri);          // (one reduction for the three counts)
    regions += n_iso + (n_dom >> 1) + n_tri;
    const int tiny_path = n_tri > 0 ? 2 : (n_dom > 0 ? 1 : 0);
    path = tiny_path > path ? tiny_path : path;
    return iso | dom | centre | ends;
}

// BFS from `src` through `pass` until nothing new is reached.  Returns the number of levels
// (eccentricity of src); `last` = cells at maximum distance.  Levels are tracked per lane.
//
// Backends with kHistBfs (16-row groups of 32-bit masks on the device; the simulator) keep the per-lane bookkeeping of a level
// in ONE instruction: the "changed" bit of the level's compare is shifted into a per-lane history word through the carry
// (hist = hist + hist + carry), and the level of a lane's last change is read out of the word (count of trailing zeros) every 32
// levels.  WANT_LAST = false (the second sweep of a double sweep) also drops the copy of the set before a lane's last change.
template <bool WANT_LAST = true, class B>
PCGRL_D int bfs_levels(B& g, typename B::mask_t src, typename B::mask_t pass, typename B::mask_t& last) {
    typedef typename B::mask_t M;
    typedef typename B::ivec_t I;
    if constexpr (B::kHistBfs != 0) {
        M n = src, prev = src ^ src;
        I hist = g.izero(), last_it = g.izero();
        int it = 0;
        PCGRL_TRACE(g, 2);
        // bfs_run: pairs of levels (a level after the last one changes nothing) until nothing changes any more or `it` reaches a
        // multiple of 32 -- a history word holds 32 levels
        while (g.template bfs_run<WANT_LAST>(n, pass, hist, prev, it)) { last_it = g.hist_fold(hist, it, last_it); hist = g.izero(); }
        last_it = g.hist_fold(hist, it, last_it);          // hist != 0 ? it - ctz(hist) : last_it
        const int ecc = g.imax(last_it);
        if (WANT_LAST) last = g.keep_where_eq(last_it, ecc, n & ~prev);
        return ecc;
    }
    M f = src, prev = src ^ src;
    I last_it = g.izero();
    int it = 0;
    constexpr int kLevels = B::kGroup == 16 ? 2 : 4;
    PCGRL_TRACE(g, 2);
    for (;;) {
        // two (four on tall maps, whose sweeps run to hundreds of levels) levels per exit test: a level after the last one changes
        // nothing (the bookkeeping selects on "changed"), so testing every few levels costs a few idle levels at the end and saves a
        // compare + scalar branch per level
        M n = f;
        bool more = false;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int uu = 0; uu < kLevels; uu++) {
            ++it;
            const M nn = pcg_expand(g, n) & pass;
            last_it = g.isel_ne(nn, n, it, last_it);
            prev = g.msel_ne(nn, n, n, prev);
            if (uu == kLevels - 1) more = g.wave_any(nn ^ n);    // wave-uniform exit; a converged group just idles
            n = nn;
        }
        f = n;
        if (!more) break;
    }
    const int ecc = g.imax(last_it);
    // rows that changed at the final level hold the last frontier; with ecc == 0 it is src itself
    last = g.keep_where_eq(last_it, ecc, f & ~prev);
    return ecc;
}

// Distance from `src` to the nearest cell of `dst` (dst not containing src) through `pass`;
// -1 when no cell of dst is reachable (run_dikjstra leaves -1 there).
template <class B>
PCGRL_D int bfs_dist(B& g, typename B::mask_t src, typename B::mask_t dst, typename B::mask_t pass) {
    typedef typename B::mask_t M;
    M f = src;
    int t = 0;
    for (;;) {
        // two levels per pair of tests (bfs_levels: a level after the last one adds nothing)
        const M n1 = pcg_expand(g, f) & pass, fresh1 = n1 & ~f;
        const M n2 = pcg_expand(g, n1) & pass, fresh2 = n2 & ~n1;
        if (g.any((fresh1 | fresh2) & dst)) return g.any(fresh1 & dst) ? t + 1 : t + 2;
        if (!g.any(fresh2)) return -1;
        t += 2;
        f = n2;
    }
}

// helper.py:197-207: number of 4-connected components of `pass`.
template <class B>
PCGRL_D int count_regions(B& g, typename B::mask_t pass) {
    typedef typename B::mask_t M;
    int regions = 0, dummy = 0;
    M rest = pass & ~pcg_tiny_components(g, pass, regions, dummy);
    if (!g.any(rest)) return regions;
    const PcgFillCtx<B> ctx = pcg_fill_ctx(g, pass);    // components never touch, so the full mask is safe
    while (g.any(rest)) {
        M comp = pcg_component(g, g.first_bit(rest), ctx);
        rest = rest & ~comp;
        ++regions;
    }
    return regions;
}

// helper.py:250-264 for one component: sweep from its first cell in row-major order, np.argmax == first
// bit of the last frontier, second sweep; returns the second eccentricity (or 0 when it provably cannot exceed `best`).
// `bound`: what is then known of the component's value -- the value itself, or 2 * e1 (<= best) when the second sweep was not needed.
template <class B>
PCGRL_D int pcg_double_sweep(B& g, typename B::mask_t comp, int best, int& bound) {
#ifdef PCGRL_EXP_NOSWEEP      /* timing experiment (tools/timeline.py with PCGRL_TL_FLAGS): what a task costs without its sweeps; results are wrong */
    return bound = g.popcount_sum(comp) / 4 + 1;
#endif
    typename B::mask_t last, unused;
    const int e1 = bfs_levels(g, g.first_bit(comp), comp, last);
    // the second sweep measures an eccentricity, which cannot exceed the diameter <= 2 * e1
    if (2 * e1 <= best) { bound = 2 * e1; return 0; }
    return bound = bfs_levels<false>(g, g.first_bit(last), comp, unused);
}
template <class B>
PCGRL_D int pcg_double_sweep(B& g, typename B::mask_t comp, int best) {
#ifdef PCGRL_EXP_NOSWEEP
    return g.popcount_sum(comp) / 4 + 1;
#endif
    typename B::mask_t last, unused;
    const int e1 = bfs_levels(g, g.first_bit(comp), comp, last);
    if (2 * e1 <= best) return 0;
    return bfs_levels<false>(g, g.first_bit(last), comp, unused);
}

// binary_prob.py:81-86: regions + helper.py:250-264 double-sweep longest path.
//
// 16-lane groups (four maps per wavefront): phase 1 extracts every component (run/column fills only) and
// remembers the two largest; phase 2 sweeps the largest -- at the same time in all four maps, which
// keeps the lockstep cost near max-over-maps instead of sum-over-maps; the second largest and, rarely,
// the rest are swept only if their size says they could still raise the maximum (a k-cell component
// cannot hold a shortest path longer than k-1).  Whole-wave groups sweep as they go.
// `champ` receives a component whose sweep produced the returned path (the "champion"), or an empty mask when the
// path comes from the closed-form tiny components: binary_incremental below builds on it.
// `ub2` (16-lane groups; -1 = not known): an upper bound of the double-sweep value of every component OTHER than the champion --
// the exact value where a component was swept, else min(size - 1, path).  binary_touch below needs it: when a change to the champion
// leaves its new value at or above that bound, no other component has to be looked at.
template <class B>
PCGRL_D void regions_and_longest_path(B& g, typename B::mask_t pass, int& regions, int& path, typename B::mask_t& champ, int& ub2, bool tight = true) {
    typedef typename B::mask_t M;
    regions = 0;
    path = 0;
    ub2 = -1;
    champ = pass ^ pass;
    const M nontiny = pass & ~pcg_tiny_components(g, pass, regions, path);
    if (!g.any(nontiny)) return;
    const PcgFillCtx<B> ctx = pcg_fill_ctx(g, pass);    // components never touch, so the full mask is safe
    if (B::kGroup != 16) {
        // One map per wavefront: no lockstep partner to align with, so sweep as we go.  Start in the
        // fullest row (almost always the giant component) so that the size test prunes the most.
        typedef typename B::ivec_t I;
        M rest = nontiny;
        const I cnt = g.popc_lanes(rest);
        M seed = g.first_bit(g.keep_where_eq(cnt, g.imax(cnt), rest));
        while (g.any(rest)) {
            const M comp = pcg_component(g, seed, ctx);
            rest = rest & ~comp;
            ++regions;
            if (g.popcount_sum(comp) - 1 > path) { const int e = pcg_double_sweep(g, comp, path); if (e > path) { path = e; champ = comp; } }
            seed = g.first_bit(rest);
        }
        return;
    }
    M rest = nontiny, big1 = nontiny ^ nontiny, big2 = big1;
    int size1 = 0, size2 = 0, size3 = 0;
    while (g.any(rest)) {
        const M comp = pcg_component(g, g.first_bit(rest), ctx);
        rest = rest & ~comp;
        ++regions;
        const int size = g.popcount_sum(comp);
        if (size > size1) { size3 = size2; size2 = size1; big2 = big1; size1 = size; big1 = comp; }
        else if (size > size2) { size3 = size2; size2 = size; big2 = comp; }
        else if (size > size3) size3 = size;
    }
    const int tiny_path = path;
    int bd1 = size1 - 1, bd2 = size2 - 1, who = 0;      // what is known of the two largest; which one is the champion (3: another one)
    if (size1 - 1 > path) { const int e = pcg_double_sweep(g, big1, path, bd1); if (e > path) { path = e; champ = big1; who = 1; } }
    // (the second largest is also swept when its size alone would set ub2: a tight bound keeps binary_touch from giving up)
    {
        const int rest_b = size3 - 1 > tiny_path ? size3 - 1 : tiny_path;
        if (size2 - 1 > path) { const int e = pcg_double_sweep(g, big2, path, bd2); if (e > path) { path = e; champ = big2; who = 2; } }
        else if (tight && size2 - 1 > rest_b) pcg_double_sweep(g, big2, rest_b, bd2);
    }
    if (size3 - 1 > path) {   // rare: a third component is still large enough to matter
        rest = nontiny & ~big1 & ~big2;
        while (g.any(rest)) {
            const M comp = pcg_component(g, g.first_bit(rest), ctx);
            rest = rest & ~comp;
            if (g.popcount_sum(comp) - 1 > path) { const int e = pcg_double_sweep(g, comp, path); if (e > path) { path = e; champ = comp; who = 3; } }
        }
    }
    // every component but the champion: the tiny ones, the two largest by what is known of them, the rest by their size
    const int b1 = who == 1 ? 0 : bd1, b2 = who == 2 ? 0 : bd2, b3 = size3 - 1;
    ub2 = tiny_path;
    ub2 = b1 > ub2 ? b1 : ub2; ub2 = b2 > ub2 ? b2 : ub2; ub2 = b3 > ub2 ? b3 : ub2;
    ub2 = ub2 > path ? path : ub2;                   // (no component has a value above the maximum)
}
template <class B>
PCGRL_D void regions_and_longest_path(B& g, typename B::mask_t pass, int& regions, int& path, typename B::mask_t& champ) {
    int ub2;
    regions_and_longest_path(g, pass, regions, path, champ, ub2);
}
template <class B>
PCGRL_D void regions_and_longest_path(B& g, typename B::mask_t pass, int& regions, int& path) {
    typename B::mask_t champ;
    regions_and_longest_path(g, pass, regions, path, champ);
}

// The same two statistics after ONE cell `c` changed, from the previous answer -- exact, and a fraction of the work.
// Preconditions (the caller routes everything else to the full computation): the previous map had a champion
// (a component `champ_old` whose double sweep gave path_old) and the changed cell is neither in it nor 4-adjacent
// to it.  Then the champion is still a component of the new map, every component that does not touch c is
// unchanged and cannot beat it, and only the components around c need looking at:
//   * count the distinct components k of (new map minus c) among the up to four neighbours of c;
//   * c became passable (`added`): they merge with c into one component -- regions + 1 - k, and that union is
//     swept if its size says it could beat the champion;
//   * c became impassable: its old component fell into those k pieces -- regions + k - 1, each piece swept if its
//     size allows (removing a cell can lengthen the shortest paths around it).
// `cbit` has the bit of c in the lane of its row and is zero elsewhere; pass_new is the new passable mask.
// ub2_old -> ub2: the bound on the other components (regions_and_longest_path) carried along: it only grows here -- by the bound of
// every component this update makes, and by the old path when the champion is replaced (a negative ub2_old, "not known", stays).
template <class B>
PCGRL_D void binary_incremental(B& g, typename B::mask_t pass_new, typename B::mask_t cbit, bool added, int regions_old, int path_old,
                                typename B::mask_t champ_old, int ub2_old, int& regions, int& path, typename B::mask_t& champ, int& ub2) {
    typedef typename B::mask_t M;
    const M base = pass_new & ~cbit;                 // the new map with c impassable
    M rest = pcg_neighbours(g, cbit) & base;
    path = path_old;
    champ = champ_old;
    // (the bound is only kept where binary_touch runs: 16-row groups.  Whole-wavefront groups compile to the plain form -- three more
    //  registers took k_stats_wide from two blocks per CU to one: C5 steady 49.5 -> 58 us)
    constexpr bool kTrack = B::kGroup == 16;
    const bool track = kTrack && ub2_old >= 0;
    int k = 0, ub = track ? ub2_old : 0;
    M uni = cbit;
    if (g.any(rest)) {
        const PcgFillCtx<B> ctx = pcg_fill_ctx(g, base);
        while (g.any(rest)) {
            const M comp = pcg_component(g, g.first_bit(rest), ctx);
            rest = rest & ~comp;
            ++k;
            if (added) uni = uni | comp;
            else {
                // swept when it could beat the champion -- or, with a bound to keep, when its size alone would raise it
                int bd = g.popcount_sum(comp) - 1, e = 0;
                const int thr = track ? ub : path;
                if (bd > thr) e = pcg_double_sweep(g, comp, thr, bd);
                if (e > path) { if (kTrack) ub = path > ub ? path : ub; path = e; champ = comp; }
                else if (kTrack) ub = bd > ub ? bd : ub;
            }
        }
    }
    if (added) {
        regions = regions_old + 1 - k;
        int bd = g.popcount_sum(uni) - 1, e = 0;
        const int thr = track ? ub : path;
        if (bd > thr) e = pcg_double_sweep(g, uni, thr, bd);
        if (e > path) { if (kTrack) ub = path > ub ? path : ub; path = e; champ = uni; }
        else if (kTrack) ub = bd > ub ? bd : ub;
    } else {
        regions = regions_old + k - 1;
    }
    ub2 = track ? ub : -1;
}
template <class B>
PCGRL_D void binary_incremental(B& g, typename B::mask_t pass_new, typename B::mask_t cbit, bool added, int regions_old, int path_old,
                                typename B::mask_t champ_old, int& regions, int& path, typename B::mask_t& champ) {
    int ub2;
    binary_incremental(g, pass_new, cbit, added, regions_old, path_old, champ_old, -1, regions, path, champ, ub2);
}

// The same two statistics after ONE cell `c` changed that IS in the champion (it became impassable) or next to it (it became
// passable) -- exact whenever it returns true.  Every component that does not touch c is unchanged, and none of them has a double-sweep
// value above ub2 (regions_and_longest_path), so:
//   * c became passable: the champion, c and the other components around c (k of them, looked for outside the champion) are one
//     component now -- regions - k; its double sweep is the new path if it is at least ub2;
//   * c became impassable: the champion fell into k pieces around c (fills inside the old champion only) -- regions + k - 1; the
//     largest is swept, a second one if its size allows; the maximum is the new path if it is at least ub2.  The other pieces join
//     "the others": ub2 grows by their bounds.
// false (the new value is below the bound, the champion is gone, three pieces that matter): the caller computes from scratch.
// One double sweep serves both cases, so the lane groups of a wavefront stay in lockstep whatever their changes were.
template <class B>
PCGRL_D bool binary_touch(B& g, typename B::mask_t pass_new, typename B::mask_t cbit, bool added, int regions_old,
                          typename B::mask_t champ_old, int ub2_old, int& regions, int& path, typename B::mask_t& champ, int& ub2) {
    typedef typename B::mask_t M;
    const M base = pass_new & ~cbit;
    const M dom = added ? (base & ~champ_old) : (champ_old & ~cbit);      // where the fills may go
    M rest = pcg_neighbours(g, cbit) & dom;
    int k = 0, size1 = 0, size2 = 0, size3 = 0;
    M uni = cbit ^ cbit, big1 = uni, big2 = uni;
    if (g.any(rest)) {
        const PcgFillCtx<B> ctx = pcg_fill_ctx(g, dom);
        while (g.any(rest)) {
            const M comp = pcg_component(g, g.first_bit(rest), ctx);
            rest = rest & ~comp;
            ++k;
            uni = uni | comp;
            const int size = g.popcount_sum(comp);
            if (size > size1) { size3 = size2; size2 = size1; big2 = big1; size1 = size; big1 = comp; }
            else if (size > size2) { size3 = size2; size2 = size; big2 = comp; }
            else if (size > size3) size3 = size;
        }
    }
    const M target = added ? (champ_old | cbit | uni) : big1;
    // (added: a sweep that cannot reach ub2 may give up -- the answer is then "false" anyway)
    int e = pcg_double_sweep(g, target, added ? ub2_old - 1 : -1);
    bool ok = true;
    int ub = ub2_old;
    champ = target;
    if (added) {
        regions = regions_old - k;
    } else {
        regions = regions_old + k - 1;
        ok = k > 0;
        if (size2 > 0) {          // the second piece: swept when it could be the new champion or when its size alone would raise the bound
            int bd = size2 - 1, e2 = 0;
            const int thr = e < ub ? e : ub;
            if (bd > thr) e2 = pcg_double_sweep(g, big2, thr, bd);
            if (e2 > e) { ub = e > ub ? e : ub; e = e2; champ = big2; }
            else ub = bd > ub ? bd : ub;
        }
        if (size3 > 0) {
            if (size3 - 1 > e) ok = false;
            else ub = size3 - 1 > ub ? size3 - 1 : ub;
        }
    }
    path = e;
    ub2 = ub;
    return ok && e >= ub2_old;
}

// ---------------------------------------------------------------- one map, several cooperating lane groups
// The same result computed by several groups (wavefronts on the device) that all hold the whole map in their
// registers and share, through `sh`, the set of cells whose component has not been retired yet plus the running
// maximum.  A group picks a seed -- from its own band of rows first, anywhere once that is used up -- extracts
// the component and retires it; two groups that started in the same component both extract it, and the one
// whose atomic test-and-clear of the component's first cell finds the bit still set owns it (counts it, sweeps
// it if its size says it can still raise the maximum).  Timing changes who does what, never the result.
//   Shared: M load_rest(); bool retire(B&, M comp) [uniform]; int best(); void raise(int)
template <class B>
PCGRL_D typename B::mask_t rlp_choose_seed(B& g, typename B::mask_t rest, int row_lo, int row_hi) {
    const typename B::mask_t mine = g.rows_between(rest, row_lo, row_hi);
    return g.first_bit(g.any(mine) ? mine : rest);
}
// my_best / my_champ: the longest sweep result of THIS group so far and its component (the group whose my_best equals
// the final shared maximum holds a champion component for binary_incremental).
#if defined(PCGRL_TIMELINE) && defined(__HIPCC__)
__device__ __forceinline__ void tl_mark(int tag);       // worklist.h (developer builds: tools/timeline_wide.py)
#define PCGRL_TLA(tag) tl_mark(tag)
#else
#define PCGRL_TLA(tag) do {} while (0)
#endif
// (device, k_stats_wide: a step is as long as its longest double sweep -- one wavefront's chain of 500-700 levels, which shares its SIMD
//  with wavefronts of the compute unit's other block that are extracting components: a level costs 88 cycles alone, 150 next to one
//  other busy wavefront, 270 next to three -- tools/probe/bfs_level_cost.hip.  The sweeping wavefront is served first: C5 42.5 -> 41.0 us
//  first window, C5b 39.2 / 38.4 -> 37.5 / 36.8, same box, alternating libraries: profiles/r5_round5/probe/ab_sweep_prio.txt.)
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__) && !defined(PCGRL_EXP_NO_SWEEP_PRIO)
#define PCGRL_SWEEP_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define PCGRL_SWEEP_PRIO(p) do {} while (0)
#endif
template <class B, class Shared>
PCGRL_D void rlp_process_seed(B& g, typename B::mask_t seed, const PcgFillCtx<B>& ctx, Shared& sh, int& regions, int& my_best,
                              typename B::mask_t& my_champ) {
    typedef typename B::mask_t M;
    const M comp = pcg_component(g, seed, ctx);
    PCGRL_TLA(30);
    if (!sh.retire(g, comp)) return;
    ++regions;
    const int best = sh.best();
    if (g.popcount_sum(comp) - 1 > best) {
        PCGRL_TLA(31);
        PCGRL_SWEEP_PRIO(3);
        const int e = pcg_double_sweep(g, comp, best);
        PCGRL_SWEEP_PRIO(0);
        if (e > my_best) { my_best = e; my_champ = comp; }
        sh.raise(e);
        PCGRL_TLA(32);
    }
}
template <class B, class Shared>
PCGRL_D void rlp_process_seed(B& g, typename B::mask_t seed, const PcgFillCtx<B>& ctx, Shared& sh, int& regions) {
    int my_best = 0;
    typename B::mask_t my_champ = seed ^ seed;
    rlp_process_seed(g, seed, ctx, sh, regions, my_best, my_champ);
}
// What every group does before the shared loop; returns the non-tiny cells (identical in every group).
template <class B>
PCGRL_D typename B::mask_t rlp_prepare(B& g, typename B::mask_t pass, int& tiny_regions, int& tiny_path) {
    tiny_regions = 0; tiny_path = 0;
    return pass & ~pcg_tiny_components(g, pass, tiny_regions, tiny_path);
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

// The number of 4-connected components after ONE cell changed its passability, from the previous count (the same local
// argument as binary_incremental): with k = the number of distinct components of (new map minus the cell) among the
// cell's up to four neighbours, a cell that became passable merges them into one (regions + 1 - k), a cell that became
// impassable leaves its component in k pieces (regions - 1 + k).  `cbit` has the cell's bit in the lane of its row.
template <class B>
PCGRL_D int regions_incremental(B& g, typename B::mask_t pass_new, typename B::mask_t cbit, bool added, int regions_old) {
    typedef typename B::mask_t M;
    const M base = pass_new & ~cbit;
    M rest = pcg_neighbours(g, cbit) & base;
    int k = 0;
    if (g.any(rest)) {
        const PcgFillCtx<B> ctx = pcg_fill_ctx(g, base);
        while (g.any(rest)) {
            rest = rest & ~pcg_component(g, g.first_bit(rest), ctx);
            ++k;
        }
    }
    return added ? regions_old + 1 - k : regions_old - 1 + k;
}

// zelda_prob.py:80-112.  out: player,key,door,enemies,regions,nearest-enemy,path-length.
// pass_change < 0: count the regions from scratch; otherwise one cell (`cbit`) changed since the map had `regions_old`
// regions: 0 = its passability (zelda_prob.py:93: everything but solid and door) did not change, 1 = it became
// passable, 2 = it became impassable.
template <class B>
PCGRL_D void zelda_stats(B& g, const PcgrlParams& P, typename B::mask_t b0, typename B::mask_t b1,
                         typename B::mask_t b2, typename B::mask_t valid, int32_t* out,
                         int pass_change = -1, typename B::mask_t cbit = typename B::mask_t(), int regions_old = 0) {
    typedef typename B::mask_t M;
    ZeldaMasks<M> z = zelda_masks(b0, b1, b2, valid);
    int player, key, door;
    g.popcount_sum3(z.player, z.key, z.door, player, key, door);
    int enemies = g.popcount_sum(z.enemy);
    M walk = z.empty | z.player | z.key | z.enemy;          // regions / player->key passable set
    int regions;
#ifdef PCGRL_EXP_NOREGIONS    /* timing experiment (tools/exp_build_local.py): zelda without its region counts; results are wrong */
    regions = regions_old + g.popcount_sum(cbit);
#else
    if (pass_change < 0) regions = count_regions(g, walk);
    else if (pass_change == 0) regions = regions_old;
    else regions = regions_incremental(g, walk, cbit, pass_change == 1, regions_old);
#endif
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

// mdungeon_prob.py:139-157 without the planner.  out: the packed row described at md_pack (dist-win default W*H, nothing
// collected).  Returns true when the planner precondition (mdungeon_prob.py:152) holds.
template <class B>
PCGRL_D bool mdungeon_stats(B& g, const PcgrlParams& P, typename B::mask_t b0, typename B::mask_t b1,
                            typename B::mask_t b2, typename B::mask_t valid, int32_t* out) {
    typedef typename B::mask_t M;
    // ids: 0 empty 1 solid 2 player 3 exit 4 potion 5 treasure 6 goblin 7 ogre (mdungeon_prob.py:49-50)
    M solid = ~b2 & ~b1 & b0 & valid;
    M player = ~b2 & b1 & ~b0 & valid;
    M exitm = ~b2 & b1 & b0 & valid;
    M potion = b2 & ~b1 & ~b0 & valid;
    M treasure = b2 & ~b1 & b0 & valid;
    M enemy = b2 & b1 & valid;
    const int np_ = g.popcount_sum(player), nx = g.popcount_sum(exitm);
    out[0] = np_; out[1] = nx;
    out[2] = g.popcount_sum(potion); out[3] = g.popcount_sum(treasure); out[4] = g.popcount_sum(enemy);
    const int regions = count_regions(g, valid & ~solid);
    out[5] = regions;
    out[6] = P.prob_width * P.prob_height;
    out[7] = 0;
    return np_ == 1 && nx == 1 && regions == 1;
}

// helper.py:37-43, 56-62 get_floor_dist(map, ["player"], ["solid"]): for every player tile the number of cells between
// it and the first solid tile below it in its column, or H - 1 if there is none.  All player bits fall together, one
// row per round; `valid` has the rows of the map.
template <class B>
PCGRL_D int floor_dist(B& g, typename B::mask_t player, typename B::mask_t solid, typename B::mask_t valid, int H) {
    typedef typename B::mask_t M;
    M act = player;
    int n_act = g.popcount_sum(act), result = 0, lost = 0;
    for (int dy = 1; dy < H && n_act > 0; dy++) {
        const M moved = g.up(act) & valid;               // row r receives row r - 1
        const int n_moved = g.popcount_sum(moved);
        lost += n_act - n_moved;                         // left the map through its last row
        const M hit = moved & solid;
        const int n_hit = g.popcount_sum(hit);
        result += n_hit * (dy - 1);
        act = moved & ~solid;
        n_act = n_moved - n_hit;
    }
    return result + (lost + n_act) * (H - 1);
}

// ddave_prob.py:141-161 without the planner.  out: the packed row described at dd_pack (dist-win default W*H, no jumps,
// nothing collected).  Returns true when the planner precondition (ddave_prob.py:156-157) holds.
template <class B>
PCGRL_D bool ddave_stats(B& g, const PcgrlParams& P, typename B::mask_t b0, typename B::mask_t b1,
                         typename B::mask_t b2, typename B::mask_t valid, int32_t* out) {
    typedef typename B::mask_t M;
    // ids: 0 empty 1 solid 2 player 3 exit 4 diamond 5 key 6 spike (ddave_prob.py:48-49)
    M solid = ~b2 & ~b1 & b0 & valid;
    M player = ~b2 & b1 & ~b0 & valid;
    M exitm = ~b2 & b1 & b0 & valid;
    M diamond = b2 & ~b1 & ~b0 & valid;
    M key = b2 & ~b1 & b0 & valid;
    M spike = b2 & b1 & ~b0 & valid;
    const int np_ = g.popcount_sum(player), nx = g.popcount_sum(exitm), nk = g.popcount_sum(key);
    out[0] = (np_ & 255) | ((nx & 255) << 8) | ((nk & 255) << 16);
    out[1] = floor_dist(g, player, solid, valid, P.height);
    out[2] = g.popcount_sum(diamond);
    out[3] = g.popcount_sum(spike);
    const int regions = count_regions(g, valid & ~solid & ~spike);
    out[4] = regions;
    out[5] = 0;
    out[6] = P.prob_width * P.prob_height;
    out[7] = 0;
    return np_ == 1 && nx == 1 && nk == 1 && regions == 1;
}
