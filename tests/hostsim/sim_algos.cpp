// CPU lane-group simulator for gym_pcgrl_amd/csrc/pcgrl_algos.h and mt19937.h.  TEST ONLY.
//
// Instantiates the same bitboard templates the GPU kernels use with a backend whose "lane
// group" is a plain array of G row masks, so the algorithm logic (not the DPP plumbing) can be
// checked against the oracle and the golden fixtures without a GPU.  Never loaded by the product.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// trace of loop entries: site 1 = component fill, site 2 = BFS sweep; the counts of wave_any() calls that
// follow each entry give the rounds of that loop (used by the lockstep cost model in tools/)
static int g_trace_site[4096], g_trace_rounds[4096], g_trace_n = 0;
#define PCGRL_TRACE(g, site) do { if (g_trace_n < 4096) { g_trace_site[g_trace_n] = (site); g_trace_rounds[g_trace_n] = 0; g_trace_n++; } } while (0)
#include "../../gym_pcgrl_amd/csrc/mt19937.h"
#include "../../gym_pcgrl_amd/csrc/pcgrl_algos.h"
#include "../../gym_pcgrl_amd/csrc/sokoban_solver.h"
#include "../../gym_pcgrl_amd/csrc/sokoban_fast.h"
#include "../../gym_pcgrl_amd/csrc/mdungeon_solver.h"
#include "../../gym_pcgrl_amd/csrc/mdungeon_fast.h"
#include "../../gym_pcgrl_amd/csrc/ddave_solver.h"
#include "../../gym_pcgrl_amd/csrc/ddave_fast.h"
#include "../../gym_pcgrl_amd/csrc/search_big.h"
#include <vector>

template <class T, int G>
struct SimVec {
    T v[G];
    SimVec() { for (int i = 0; i < G; i++) v[i] = 0; }
    SimVec(T x) { for (int i = 0; i < G; i++) v[i] = x; }
};
#define SV_BIN(op) \
    template <class T, int G> SimVec<T, G> operator op(const SimVec<T, G>& a, const SimVec<T, G>& b) { \
        SimVec<T, G> r; for (int i = 0; i < G; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
SV_BIN(&) SV_BIN(|) SV_BIN(^) SV_BIN(+)
template <class T, int G> SimVec<T, G> operator~(const SimVec<T, G>& a) { SimVec<T, G> r; for (int i = 0; i < G; i++) r.v[i] = ~a.v[i]; return r; }
template <class T, int G> SimVec<T, G> operator<<(const SimVec<T, G>& a, int s) { SimVec<T, G> r; for (int i = 0; i < G; i++) r.v[i] = a.v[i] << s; return r; }
template <class T, int G> SimVec<T, G> operator>>(const SimVec<T, G>& a, int s) { SimVec<T, G> r; for (int i = 0; i < G; i++) r.v[i] = a.v[i] >> s; return r; }

static long g_sim_iters = 0;
static int g_spurious = 0;          // wave_any() may report "another group is still busy" this many times
static unsigned g_rng = 12345u;
template <class T> static T sim_brev(T v) { T r = 0; for (unsigned i = 0; i < 8 * sizeof(T); i++) if ((v >> i) & 1) r |= (T)1 << (8 * sizeof(T) - 1 - i); return r; }
template <int G, class T>
struct SimGroup {
    typedef SimVec<T, G> mask_t;
    typedef SimVec<int, G> ivec_t;
    enum { kGroup = G, kLog2Group = 4, kHistBfs = 1 };
    // two BFS levels with the carry-history bookkeeping (lanegroup_dev.h has the device form)
    template <bool WANT_LAST>
    bool bfs_run(mask_t& n, const mask_t& pass, ivec_t& hist, mask_t& prev, int& it) const {
        for (;;) {
            const bool more = bfs_pair<WANT_LAST>(n, pass, hist, prev);
            it += 2;
            if (!more) return false;
            if ((it & 31) == 0) return true;
        }
    }
    template <bool WANT_LAST>
    bool bfs_pair(mask_t& n, const mask_t& pass, ivec_t& hist, mask_t& prev) const {
        bool more = false;
        for (int lv = 0; lv < 2; lv++) {
            mask_t nn;
            for (int i = 0; i < G; i++) {
                T e = n.v[i] | (T)(n.v[i] << 1) | (T)(n.v[i] >> 1);
                if (i > 0) e |= n.v[i - 1];
                if (i + 1 < G) e |= n.v[i + 1];
                nn.v[i] = e & pass.v[i];
            }
            mask_t diff;
            for (int i = 0; i < G; i++) {
                const bool ch = nn.v[i] != n.v[i];
                diff.v[i] = nn.v[i] ^ n.v[i];
                if (WANT_LAST && ch) prev.v[i] = n.v[i];
                hist.v[i] = (int)(((unsigned)hist.v[i] << 1) | (ch ? 1u : 0u));
            }
            if (lv == 1) more = wave_any(diff);
            n = nn;
        }
        return more;
    }
    ivec_t hist_fold(const ivec_t& hist, int it, const ivec_t& last_it) const {
        ivec_t r; for (int i = 0; i < G; i++) r.v[i] = hist.v[i] != 0 ? it - __builtin_ctz((unsigned)hist.v[i]) : last_it.v[i]; return r;
    }
    // like the device: the doubling moves never cross an aligned block of 16 rows (one DPP row)
    mask_t rows_down(const mask_t& m, int k) const { mask_t r; int n = 1 << k; for (int i = 0; i < G; i++) if ((i & 15) >= n) r.v[i] = m.v[i - n]; return r; }
    mask_t rows_up(const mask_t& m, int k) const { mask_t r; int n = 1 << k; for (int i = 0; i < G; i++) if ((i & 15) + n < 16) r.v[i] = m.v[i + n]; return r; }
    ivec_t izero() const { return ivec_t(); }
    bool wave_any(const mask_t& m) const {
        g_sim_iters++;
        if (g_trace_n > 0) g_trace_rounds[g_trace_n - 1]++;
        if (any(m)) return true;
        g_rng = g_rng * 1664525u + 1013904223u;
        if (g_spurious > 0 && (g_rng >> 16) % 3 != 0) { g_spurious--; return true; }   // another group of the wave is not done yet
        return false;
    }
    void popcount_sum2(const mask_t& a, const mask_t& b, int& x, int& y) const { x = popcount_sum(a); y = popcount_sum(b); }
    void popcount_sum3(const mask_t& a, const mask_t& b, const mask_t& c, int& x, int& y, int& z) const { x = popcount_sum(a); y = popcount_sum(b); z = popcount_sum(c); }
    ivec_t popc_lanes(const mask_t& m) const { ivec_t r; for (int i = 0; i < G; i++) r.v[i] = __builtin_popcountll((unsigned long long)m.v[i]); return r; }
    int imax(const ivec_t& v) const { int m = v.v[0]; for (int i = 1; i < G; i++) m = v.v[i] > m ? v.v[i] : m; return m; }
    ivec_t isel_ne(const mask_t& a, const mask_t& b, int x, const ivec_t& y) const { ivec_t r; for (int i = 0; i < G; i++) r.v[i] = a.v[i] != b.v[i] ? x : y.v[i]; return r; }
    mask_t msel_ne(const mask_t& a, const mask_t& b, const mask_t& x, const mask_t& y) const { mask_t r; for (int i = 0; i < G; i++) r.v[i] = a.v[i] != b.v[i] ? x.v[i] : y.v[i]; return r; }
    mask_t keep_where_eq(const ivec_t& v, int x, const mask_t& m) const { mask_t r; for (int i = 0; i < G; i++) r.v[i] = v.v[i] == x ? m.v[i] : (T)0; return r; }
    mask_t bitrev(const mask_t& m) const { mask_t r; for (int i = 0; i < G; i++) r.v[i] = sim_brev(m.v[i]); return r; }
    mask_t up(const mask_t& m) const { mask_t r; for (int i = 1; i < G; i++) r.v[i] = m.v[i - 1]; return r; }
    mask_t down(const mask_t& m) const { mask_t r; for (int i = 0; i + 1 < G; i++) r.v[i] = m.v[i + 1]; return r; }
    bool any(const mask_t& m) const { for (int i = 0; i < G; i++) if (m.v[i]) return true; return false; }
    mask_t rows_between(const mask_t& m, int lo, int hi) const { mask_t r; for (int i = 0; i < G; i++) if (i >= lo && i < hi) r.v[i] = m.v[i]; return r; }
    int first_row(const mask_t& m) const { for (int i = 0; i < G; i++) if (m.v[i]) return i; return -1; }
    bool any_ne(const mask_t& a, const mask_t& b) const { g_sim_iters++; for (int i = 0; i < G; i++) if (a.v[i] != b.v[i]) return true; return false; }
    mask_t first_bit(const mask_t& m) const {
        mask_t r;
        for (int i = 0; i < G; i++) if (m.v[i]) { r.v[i] = m.v[i] & (T)(0 - m.v[i]); break; }
        return r;
    }
    int popcount_sum(const mask_t& m) const { int n = 0; for (int i = 0; i < G; i++) n += __builtin_popcountll((unsigned long long)m.v[i]); return n; }
};

template <int G, class T>
static void run(int prob, const uint8_t* map, int h, int w, int pw, int ph, int32_t* out, int* need_solver) {
    typedef SimGroup<G, T> Gp;
    typedef typename Gp::mask_t M;
    Gp g;
    M b0, b1, b2, valid;
    for (int y = 0; y < h; y++) {
        valid.v[y] = (w >= (int)(8 * sizeof(T))) ? ~(T)0 : (((T)1 << w) - 1);
        for (int x = 0; x < w; x++) {
            T t = map[y * w + x];
            b0.v[y] |= (t & 1) << x; b1.v[y] |= ((t >> 1) & 1) << x; b2.v[y] |= ((t >> 2) & 1) << x;
        }
    }
    PcgrlParams P; memset(&P, 0, sizeof(P));
    P.prob = prob; P.width = w; P.height = h; P.prob_width = pw; P.prob_height = ph;
    for (int k = 0; k < 8; k++) out[k] = 0;
    *need_solver = 0;
    if (prob == PCGRL_PROB_BINARY) {
        int regions, path;
        regions_and_longest_path(g, ~b0 & valid, regions, path);
        out[0] = regions; out[1] = path;
    } else if (prob == PCGRL_PROB_ZELDA) {
        zelda_stats(g, P, b0, b1, b2, valid, out);
    } else if (prob == PCGRL_PROB_DDAVE) {
        *need_solver = ddave_stats(g, P, b0, b1, b2, valid, out) ? 1 : 0;
    } else if (prob == PCGRL_PROB_MDUNGEON) {
        *need_solver = mdungeon_stats(g, P, b0, b1, b2, valid, out) ? 1 : 0;
    } else {
        *need_solver = sokoban_stats(g, P, b0, b1, b2, valid, out) ? 1 : 0;
    }
}

// binary stats the way k_stats_wide computes them: `ngroups` cooperating groups sharing the rest set.  The groups
// take turns; in every turn ALL of them choose their seed from the same snapshot before any of them retires its
// component (the worst interleaving: as many duplicate extractions as possible).
template <int G, class T>
struct SimShared {
    typedef SimGroup<G, T> Gp;
    typedef typename Gp::mask_t M;
    M rest; int bestv; long dup;
    M load_rest() const { return rest; }
    bool retire(Gp& g, const M& comp) {
        const M fb = g.first_bit(comp);
        bool won = false;
        for (int i = 0; i < G; i++) if (fb.v[i] && (rest.v[i] & fb.v[i])) won = true;
        for (int i = 0; i < G; i++) rest.v[i] &= ~comp.v[i];
        if (!won) dup++;
        return won;
    }
    int best() const { return bestv; }
    void raise(int v) { if (v > bestv) bestv = v; }
};
template <int G, class T>
static long run_shared(const uint8_t* map, int h, int w, int ngroups, int32_t* out) {
    typedef SimGroup<G, T> Gp;
    typedef typename Gp::mask_t M;
    Gp g;
    M b0, valid;
    for (int y = 0; y < h; y++) {
        valid.v[y] = (w >= (int)(8 * sizeof(T))) ? ~(T)0 : (((T)1 << w) - 1);
        for (int x = 0; x < w; x++) b0.v[y] |= (T)(map[y * w + x] & 1) << x;
    }
    const M pass = ~b0 & valid;
    SimShared<G, T> sh;
    int tr, tp, regions = 0;
    sh.rest = rlp_prepare(g, pass, tr, tp);
    sh.bestv = tp; sh.dup = 0;
    regions = tr;
    const PcgFillCtx<Gp> ctx = pcg_fill_ctx(g, pass);
    const int bh = (h + ngroups - 1) / ngroups;
    while (g.any(sh.rest)) {
        std::vector<M> seeds;
        for (int k = 0; k < ngroups; k++) {
            const int lo = k * bh, hi = (lo + bh < h) ? lo + bh : h;
            seeds.push_back(rlp_choose_seed(g, sh.rest, lo, hi));
        }
        for (int k = 0; k < ngroups; k++) rlp_process_seed(g, seeds[k], ctx, sh, regions);
    }
    out[0] = regions; out[1] = sh.bestv;
    return sh.dup;
}

// Binary statistics along a sequence of single-cell changes the way the step kernels compute them: full
// computation when there is no champion or the change touches it, binary_incremental otherwise.
// flips: cell indices (y * w + x) toggled one after the other; out: [nflips + 1][2] (regions, path); returns the
// number of incremental updates.
// counts (optional): [0] incremental updates, [1] updates through the champion (binary_touch), [2] of those given up and computed in
// full, [3] full computations for want of a champion.
template <int G, class T>
static int run_incremental(const uint8_t* map0, int h, int w, const int* flips, int nflips, int32_t* out, int* counts = nullptr) {
    typedef SimGroup<G, T> Gp;
    typedef typename Gp::mask_t M;
    Gp g;
    M pass, valid;
    for (int y = 0; y < h; y++) {
        valid.v[y] = (w >= (int)(8 * sizeof(T))) ? ~(T)0 : (((T)1 << w) - 1);
        for (int x = 0; x < w; x++) if (!(map0[y * w + x] & 1)) pass.v[y] |= (T)1 << x;
    }
    int regions, path, ninc = 0, ub2 = -1, ntouch = 0, ngiveup = 0, nfull = 0;
    M champ;
    regions_and_longest_path(g, pass, regions, path, champ, ub2);
    out[0] = regions; out[1] = path;
    for (int f = 0; f < nflips; f++) {
        const int y = flips[f] / w, x = flips[f] % w;
        M cbit; cbit.v[y] = (T)1 << x;
        const bool added = !(pass.v[y] >> x & 1);
        pass.v[y] ^= (T)1 << x;
        M touch = (pcg_expand(g, cbit)) & champ;
        if (!g.any(champ) || ub2 < 0) {
            regions_and_longest_path(g, pass, regions, path, champ, ub2);
            nfull++;
        } else if (g.any(touch)) {
            int r2, p2, u2; M c2;
            ntouch++;
            if (binary_touch(g, pass, cbit, added, regions, champ, ub2, r2, p2, c2, u2)) { regions = r2; path = p2; champ = c2; ub2 = u2; }
            else {
                regions_and_longest_path(g, pass, regions, path, champ, ub2); ngiveup++;
            }
        } else {
            int r2, p2, u2; M c2;
            binary_incremental(g, pass, cbit, added, regions, path, champ, ub2, r2, p2, c2, u2);
            regions = r2; path = p2; champ = c2; ub2 = u2;
            ninc++;
        }
        out[2 * (f + 1)] = regions; out[2 * (f + 1) + 1] = path;
    }
    if (counts) { counts[0] = ninc; counts[1] = ntouch; counts[2] = ngiveup; counts[3] = nfull; }
    return ninc;
}

// Zelda statistics along a sequence of single-tile writes: the first map from scratch, every following one with the
// incremental region count.  writes: [n][2] = (cell, tile); out: [n + 1][7]
template <int G, class T>
static void run_zelda_incremental(const uint8_t* map0, int h, int w, const int* writes, int n, int32_t* out) {
    typedef SimGroup<G, T> Gp;
    typedef typename Gp::mask_t M;
    Gp g;
    std::vector<uint8_t> map(map0, map0 + h * w);
    PcgrlParams P; memset(&P, 0, sizeof(P));
    P.prob = PCGRL_PROB_ZELDA; P.width = w; P.height = h; P.prob_width = w; P.prob_height = h;
    auto planes = [&](M& b0, M& b1, M& b2, M& valid) {
        b0 = M(); b1 = M(); b2 = M(); valid = M();
        for (int y = 0; y < h; y++) {
            valid.v[y] = (w >= (int)(8 * sizeof(T))) ? ~(T)0 : (((T)1 << w) - 1);
            for (int x = 0; x < w; x++) {
                T t = map[y * w + x];
                b0.v[y] |= (t & 1) << x; b1.v[y] |= ((t >> 1) & 1) << x; b2.v[y] |= ((t >> 2) & 1) << x;
            }
        }
    };
    M b0, b1, b2, valid;
    planes(b0, b1, b2, valid);
    zelda_stats(g, P, b0, b1, b2, valid, out);
    for (int i = 0; i < n; i++) {
        const int cell = writes[2 * i], tile = writes[2 * i + 1];
        const int old = map[cell];
        map[cell] = (uint8_t)tile;
        planes(b0, b1, b2, valid);
        const bool po = old != 1 && old != 4, pn = tile != 1 && tile != 4;
        M cbit; cbit.v[cell / w] = (T)1 << (cell % w);
        zelda_stats(g, P, b0, b1, b2, valid, out + 7 * (i + 1), po == pn ? 0 : (pn ? 1 : 2), cbit, out[7 * i + 4]);
    }
}

extern "C" {
void sim_zelda_incremental(const uint8_t* map0, int h, int w, const int* writes, int n, int32_t* out) {
    if (w > 32) run_zelda_incremental<16, uint64_t>(map0, h, w, writes, n, out);
    else run_zelda_incremental<16, uint32_t>(map0, h, w, writes, n, out);
}
int sim_binary_incremental(const uint8_t* map0, int h, int w, const int* flips, int nflips, int32_t* out) {
    if (w > 32) return run_incremental<16, uint64_t>(map0, h, w, flips, nflips, out);
    return run_incremental<16, uint32_t>(map0, h, w, flips, nflips, out);
}
int sim_binary_incremental2(const uint8_t* map0, int h, int w, const int* flips, int nflips, int32_t* out, int* counts) {
    if (w > 32) return run_incremental<16, uint64_t>(map0, h, w, flips, nflips, out, counts);
    return run_incremental<16, uint32_t>(map0, h, w, flips, nflips, out, counts);
}
long sim_stats_shared(const uint8_t* map, int h, int w, int ngroups, int32_t* out) {
    if (w > 32) return run_shared<64, uint64_t>(map, h, w, ngroups, out);
    return run_shared<64, uint32_t>(map, h, w, ngroups, out);
}
// the device solver (sokoban_solver.h / sokoban_fast.h) run on the host: same pool/heap/table layout as k_sokoban.
// fast = 1 takes the register-resident search for levels with at most SOKF_MAXC crates (what the kernel does),
// fast = 0 forces the generic one.
static int g_chunk = 0;       // > 0: the compact searches run in pieces of that many pops (sim_set_chunk)
static long g_pieces = 0;
void sim_set_chunk(int n) { g_chunk = n; }
long sim_pieces_reset() { long v = g_pieces; g_pieces = 0; return v; }
int sim_sokoban_solve2(const uint8_t* map, int h, int w, int power, int shortcut, int fast, int* dist, int* sol, int* iters) {
    SokLevel L; SokNode root;
    int ncr = sok_build_level(map, w, h, L, root);
    if (ncr > SOK_MAXC) return -1;
    sok_init_deadlocks(L);
    root.h = (uint16_t)sok_heuristic(L, root.crate);
    std::vector<SokNode> pool(4 * (size_t)power + 4);
    const int heap_cap = 4 * power + 4;
    std::vector<uint32_t> heap(heap_cap);
    int tsize = 1024; while (tsize < 2 * power) tsize <<= 1;
    if (power <= SOK_LDS_POWER) tsize = SOK_LDS_TABLE;
    if (!(fast && L.nc <= SOKF_MAXC)) {
        std::vector<uint32_t> table(tsize);
        SokNode work;
        uint32_t* tp = table.data();
        sok_run_game(L, pool.data(), heap.data(), tp, tsize, work, root, power, shortcut != 0,
                     [tp](int n) { for (int i = 0; i < n; i++) tp[i] = 0; }, *dist, *sol, iters);
        return 0;
    }
    std::vector<uint64_t> table(tsize);
    const int KS[4] = {-1, 2, 1, 0};
    bool win = false;
    int hh = 0, dd = 0;
    for (int a = 0; a < 4; a++) iters[a] = 0;
    for (int a = 0; a < 4 && !win; a++) {
        for (int i = 0; i < tsize; i++) table[i] = 0;
        bool exhausted = false;
        SokFastNode* fp = reinterpret_cast<SokFastNode*>(pool.data());
        SokFastNode cache[4];
        if (g_chunk > 0) {       // the same search in pieces of g_chunk pops (SokResume: what pcgrl_step_async does across launches)
            SokResume st; memset(&st, 0, sizeof(st));
            do {
                const SokResumeArg ra = {&st, st.iterations + g_chunk};
                if (L.cells <= 64)
                    win = sok_search_fast<1>(L, fp, heap.data(), table.data(), tsize - 1, cache, root, KS[a], power, hh, dd, iters[a], exhausted, SokNoHook(), SokKidsSerial(), nullptr, ra);
                else
                    win = sok_search_fast<4>(L, fp, heap.data(), table.data(), tsize - 1, cache, root, KS[a], power, hh, dd, iters[a], exhausted, SokNoHook(), SokKidsSerial(), nullptr, ra);
                g_pieces++;
            } while (st.suspended);
        } else if (L.cells <= 64)
            win = sok_search_fast<1>(L, fp, heap.data(), table.data(), tsize - 1, cache, root, KS[a], power, hh, dd, iters[a], exhausted, SokNoHook(), SokKidsSerial());
        else
            win = sok_search_fast<4>(L, fp, heap.data(), table.data(), tsize - 1, cache, root, KS[a], power, hh, dd, iters[a], exhausted, SokNoHook(), SokKidsSerial());
        if (a == 0 && !win && exhausted && shortcut) break;
    }
    *dist = win ? 0 : hh;
    *sol = win ? dd : 0;
    return 0;
}
int sim_sokoban_solve(const uint8_t* map, int h, int w, int power, int shortcut, int* dist, int* sol, int* iters) {
    return sim_sokoban_solve2(map, h, w, power, shortcut, 0, dist, sol, iters);
}
// the device planner of the mdungeon problem (mdungeon_solver.h) run on the host: same pool/heap/table layout as k_mdungeon.
// out5 = dist-win, sol-length, col-potions, col-treasures, col-enemies
int sim_mdungeon_solve(const uint8_t* map, int h, int w, int power, int shortcut, int* out5, int* iters) {
    if ((w + 2) * (h + 2) > 256) return -1;
    MdLevel L; MdNode root, work;
    md_build_level(map, w, h, L, root);
    std::vector<MdNode> pool(4 * (size_t)power + 4);
    std::vector<uint32_t> heap(4 * (size_t)power + 4);
    int tsize = 1024; while (tsize < 2 * power) tsize <<= 1;
    if (power <= SOK_LDS_POWER) tsize = SOK_LDS_TABLE;
    std::vector<uint32_t> table(tsize);
    uint32_t* tp = table.data();
    md_run_game(L, pool.data(), heap.data(), tp, tsize, work, root, power, shortcut != 0,
                [tp](int n) { for (int i = 0; i < n; i++) tp[i] = 0; }, out5, iters);
    return 0;
}
// fast = 1: the compact search (mdungeon_fast.h) when the level qualifies, as k_mdungeon does
int sim_mdungeon_solve2(const uint8_t* map, int h, int w, int power, int shortcut, int fast, int* out5, int* iters) {
    if ((w + 2) * (h + 2) > 256) return -1;
    MdLevel L; MdNode root;
    md_build_level(map, w, h, L, root);
    MdFastLevel F;
    const int nitems = mdf_level(L, root, F);
    if (!(fast && nitems <= MDF_MAXI && power <= SOK_LDS_POWER)) return sim_mdungeon_solve(map, h, w, power, shortcut, out5, iters);
    std::vector<MdFastNode> pool(4 * (size_t)power + 4);
    std::vector<uint32_t> heap(4 * (size_t)power + 4);
    const int tsize = SOK_LDS_TABLE;
    std::vector<uint64_t> table(tsize);
    const int KS[4] = {2, 1, 0, -1};
    bool win = false;
    uint64_t key = 0; int hh = 0, dd = 0;
    for (int a = 0; a < 4; a++) iters[a] = 0;
    for (int a = 0; a < 4 && !win; a++) {
        for (int i = 0; i < tsize; i++) table[i] = 0;
        bool exhausted = false;
        MdFastNode cache[4];
        if (g_chunk > 0) {
            SokResume st; memset(&st, 0, sizeof(st));
            do {
                const SokResumeArg ra = {&st, st.iterations + g_chunk};
                win = md_search_fast(L, F, pool.data(), heap.data(), table.data(), tsize - 1, cache, root, KS[a], power, key, hh, dd, iters[a], exhausted, SokNoHook(), MdKidsSerial(), nullptr, ra);
                g_pieces++;
            } while (st.suspended);
        } else
        win = md_search_fast(L, F, pool.data(), heap.data(), table.data(), tsize - 1, cache, root, KS[a], power, key, hh, dd, iters[a], exhausted, SokNoHook(), MdKidsSerial());
        if (a < 3 && !win && exhausted && shortcut) a = 2;
    }
    mdf_result(F, key, hh, dd, win, out5);
    return 1;
}
// the device planner of the ddave problem (ddave_solver.h) run on the host.  out4 = dist-win, sol-length, num-jumps, col-diamonds
int sim_ddave_solve(const uint8_t* map, int h, int w, int power, int* out4, int* iters) {
    if ((w + 2) * (h + 2) > 256) return -1;
    DdLevel L; DdNode root, work;
    dd_build_level(map, w, h, L, root);
    std::vector<DdNode> pool(4 * (size_t)power + 4);
    std::vector<uint32_t> heap(4 * (size_t)power + 4);
    int tsize = 1024; while (tsize < 2 * power) tsize <<= 1;
    if (power <= SOK_LDS_POWER) tsize = SOK_LDS_TABLE;
    std::vector<uint32_t> table(tsize);
    uint32_t* tp = table.data();
    dd_run_game(L, pool.data(), heap.data(), tp, tsize, work, root, power, [tp](int n) { for (int i = 0; i < n; i++) tp[i] = 0; }, out4, iters);
    return 0;
}
// fast = 1: the compact search (ddave_fast.h) when the level qualifies, as k_ddave does
int sim_ddave_solve2(const uint8_t* map, int h, int w, int power, int fast, int* out4, int* iters) {
    if ((w + 2) * (h + 2) > 256) return -1;
    DdLevel L; DdNode root;
    dd_build_level(map, w, h, L, root);
    DdFastLevel F;
    const int nd = ddf_level(L, F);
    if (!(fast && nd <= DDF_MAXD && power <= SOK_LDS_POWER)) return sim_ddave_solve(map, h, w, power, out4, iters);
    std::vector<DdFastNode> pool(4 * (size_t)power + 4);
    std::vector<uint32_t> heap(4 * (size_t)power + 4);
    const int tsize = SOK_LDS_TABLE;
    std::vector<uint64_t> table(tsize);
    const int KS[4] = {2, 1, 0, -1};
    bool win = false;
    uint64_t key = 0; int hh = 0, dd = 0, jj = 0;
    for (int a = 0; a < 4; a++) iters[a] = 0;
    for (int a = 0; a < 4 && !win; a++) {
        for (int i = 0; i < tsize; i++) table[i] = 0;
        bool exhausted = false;
        DdFastNode cache[4];
        if (g_chunk > 0) {
            SokResume st; memset(&st, 0, sizeof(st));
            do {
                const SokResumeArg ra = {&st, st.iterations + g_chunk};
                win = dd_search_fast(L, F, pool.data(), heap.data(), table.data(), tsize - 1, cache, root, KS[a], power, key, hh, dd, jj, iters[a], exhausted,
                                     SokNoHook(), DdKidsSerial(), nullptr, ra);
                g_pieces++;
            } while (st.suspended);
        } else
        win = dd_search_fast(L, F, pool.data(), heap.data(), table.data(), tsize - 1, cache, root, KS[a], power, key, hh, dd, jj, iters[a], exhausted,
                             SokNoHook(), DdKidsSerial());
    }
    ddf_result(F, key, hh, dd, jj, win, out4);
    return 1;
}
long sim_iters_reset() { long v = g_sim_iters; g_sim_iters = 0; return v; }
void sim_set_spurious(int n) { g_spurious = n; }
int sim_trace_get(int* sites, int* rounds) { int n = g_trace_n; for (int i = 0; i < n; i++) { sites[i] = g_trace_site[i]; rounds[i] = g_trace_rounds[i]; } g_trace_n = 0; return n; }
// variant: 0 = smallest fitting (G16 if h<=16 else G64; u32 if w<=32 else u64), 1 = force G64, 2 = force u64, 3 = both
int sim_stats(int prob, const uint8_t* map, int h, int w, int pw, int ph, int variant, int32_t* out, int* need_solver) {
    bool g64 = h > 16 || (variant & 1), m64 = w > 32 || (variant & 2);
    if (!g64 && !m64) run<16, uint32_t>(prob, map, h, w, pw, ph, out, need_solver);
    else if (!g64) run<16, uint64_t>(prob, map, h, w, pw, ph, out, need_solver);
    else if (!m64) run<64, uint32_t>(prob, map, h, w, pw, ph, out, need_solver);
    else run<64, uint64_t>(prob, map, h, w, pw, ph, out, need_solver);
    return 0;
}
double sim_range_reward(double n, double o, double lo, double hi) { return range_reward(n, o, lo, hi); }
int sim_range_reward_i(int n, int o, int lo, int hi) { return range_reward_i(n, o, lo, hi); }
double sim_reward(const PcgrlParams* P, const int32_t* n, const int32_t* o) { return compute_reward(*P, n, o); }
int sim_params_size() { return (int)sizeof(PcgrlParams); }

// lazy-ring MT19937 against numpy's stream
void sim_mt_randint(const uint32_t* key624, int n, int count, int64_t* out) {
    uint32_t ring[624]; memcpy(ring, key624, sizeof(ring)); int cur = 0;
    for (int i = 0; i < count; i++) out[i] = mt_randint(ring, cur, n);
}
void sim_mt_random(const uint32_t* key624, int count, double* out) {
    uint32_t ring[624]; memcpy(ring, key624, sizeof(ring)); int cur = 0;
    for (int i = 0; i < count; i++) out[i] = mt_random(ring, cur);
}
// the reset kernel's parallel generation (reset_env.h wave_reset_env), emulated lane by lane: a round makes the words of 113 cells --
// lane l those of cell c0 + l, lanes 0..48 also those of cell c0 + 64 + l -- with every read before every write
void sim_mt_mapgen(const uint32_t* key624, const double* prob, int ntiles, int w, int h, uint8_t* tiles, int* xy, uint32_t* ring_out, int* cur_out) {
    uint32_t mt[624]; memcpy(mt, key624, sizeof(mt)); int cur = 0;
    double cdf[8]; pcgrl_build_cdf(prob, ntiles, cdf);
    int cells = w * h;
    for (int c0 = 0; c0 < cells; c0 += 113) {
        const int rem = cells - c0;
        const int nA = rem < 64 ? rem : 64, nB = rem <= 64 ? 0 : (rem - 64 < 49 ? rem - 64 : 49);
        uint32_t ya[2][64], yb[2][64]; int ss[2][64];
        for (int lane = 0; lane < 64; lane++)
            for (int k = 0; k < 2; k++) {
                int s = mt_wrap(cur + 2 * lane >= 624 ? cur + 2 * lane - 624 : cur + 2 * lane);
                if (k) s = mt_wrap(s + 128);
                ss[k][lane] = s;
                uint32_t x0 = mt[s], x1 = mt[mt_wrap(s + 1)], x2 = mt[mt_wrap(s + 2)];
                uint32_t xm0 = mt[mt_wrap(s + 397)], xm1 = mt[mt_wrap(s + 398)];
                ya[k][lane] = mt_twist(x0, x1, xm0); yb[k][lane] = mt_twist(x1, x2, xm1);
            }
        for (int k = 0; k < 2; k++)
            for (int lane = 0; lane < (k ? nB : nA); lane++) {
                int c = c0 + 64 * k + lane;
                mt[ss[k][lane]] = ya[k][lane]; mt[mt_wrap(ss[k][lane] + 1)] = yb[k][lane];
                tiles[c] = (uint8_t)pcgrl_pick_tile(cdf, ntiles, mt_to_double(mt_temper(ya[k][lane]), mt_temper(yb[k][lane])));
            }
        cur += 2 * (nA + nB); cur = cur >= 624 ? cur - 624 : cur;
    }
    xy[0] = mt_randint(mt, cur, w); xy[1] = mt_randint(mt, cur, h);
    memcpy(ring_out, mt, sizeof(mt)); *cur_out = cur;
}

// ---- the general searches (search_big.h: levels beyond 256 bordered cells, solver_power beyond 16 383) run on the host, with the
// arena k_search_big gives them.  shortcut = 0: every agent runs as in the reference (iteration counts comparable).
struct BigHost {
    std::vector<uint8_t> pool; std::vector<uint64_t> heap; std::vector<uint32_t> table; std::vector<uint16_t> cx, cy;
    BigSearchCtx C;
    BigHost(int w, int h, int power, int stride) {
        C.w = w + 2; C.h = h + 2; C.cells = C.w * C.h; C.nwb = (C.cells + 63) >> 6;
        C.nodes_cap = 4 * power + 4; C.power = power;
        int tsize = 1024; while (tsize < 2 * power) tsize <<= 1;
        C.table_mask = tsize - 1;
        pool.resize((size_t)C.nodes_cap * stride); heap.resize(C.nodes_cap); table.resize(tsize); cx.resize(C.cells); cy.resize(C.cells);
        C.pool = pool.data(); C.heap = heap.data(); C.table = table.data(); C.cx = cx.data(); C.cy = cy.data();
    }
    void clear() { for (auto& t : table) t = 0; }
};
int sim_big_sokoban(const uint8_t* map, int h, int w, int power, int shortcut, int* dist, int* sol, int* iters) {
    if ((w + 2) * (h + 2) > BIG_MAX_WORDS * 64) return -1;
    BigHost H(w, h, power, sokb_stride(SOKB_MAXC));
    static SokbLevel L; static SokbNode root, work;
    uint64_t used[SOKB_MAXC / 64];
    const int ncr = sokb_build_level(H.C, map, w, L, root);
    if (ncr > SOKB_MAXC) return -2;
    std::vector<uint16_t> corners(H.C.cells);
    sokb_init_deadlocks(H.C, L, corners.data());
    root.h = (uint16_t)sokb_heuristic(H.C, L, root.crate, used);
    const int KS[4] = {-1, 2, 1, 0};
    bool win = false;
    int hh = 0, dd = 0;
    for (int a = 0; a < 4; a++) iters[a] = 0;
    for (int a = 0; a < 4 && !win; a++) {
        H.clear();
        bool exhausted = false;
        win = sokb_search(H.C, L, work, root, KS[a], used, hh, dd, iters[a], exhausted);
        if (a == 0 && !win && exhausted && shortcut) break;
    }
    *dist = win ? 0 : hh; *sol = win ? dd : 0;
    return 0;
}
int sim_big_mdungeon(const uint8_t* map, int h, int w, int power, int shortcut, int* out5, int* iters) {
    if ((w + 2) * (h + 2) > BIG_MAX_WORDS * 64) return -1;
    BigHost H(w, h, power, mdb_stride(BIG_MAX_WORDS));
    static MdbLevel L;
    uint64_t alive[BIG_MAX_WORDS];
    MdbWork work; work.alive = alive;
    mdb_build_level(H.C, map, w, L, work);
    mdb_store(H.C, 0, work);
    const int KS[4] = {2, 1, 0, -1};
    bool win = false;
    for (int a = 0; a < 4; a++) iters[a] = 0;
    for (int a = 0; a < 4 && !win; a++) {
        H.clear();
        bool exhausted = false;
        win = mdb_search(H.C, L, work, KS[a], iters[a], exhausted);
        if (a < 3 && !win && exhausted && shortcut) a = 2;
    }
    const uint64_t* root_alive = reinterpret_cast<const uint64_t*>(H.C.pool);
    int pot = 0, ene = 0;
    for (int i = 0; i < H.C.nwb; i++) {
        const uint64_t gone = root_alive[i] & ~work.alive[i];
        pot += md_popcount(gone & L.potion[i]);
        ene += md_popcount(gone & (L.goblin[i] | L.ogre[i]));
    }
    out5[0] = win ? 0 : (int)work.t.h; out5[1] = win ? (int)work.t.depth : 0; out5[2] = pot; out5[3] = work.t.treasures; out5[4] = ene;
    return 0;
}
int sim_big_ddave(const uint8_t* map, int h, int w, int power, int* out4, int* iters) {
    if ((w + 2) * (h + 2) > BIG_MAX_WORDS * 64) return -1;
    BigHost H(w, h, power, mdb_stride(BIG_MAX_WORDS));
    static DdbLevel L;
    uint64_t alive[BIG_MAX_WORDS];
    MdbWork work; work.alive = alive;
    ddb_build_level(H.C, map, w, L, work);
    mdb_store(H.C, 0, work);
    const int KS[4] = {2, 1, 0, -1};
    bool win = false;
    for (int a = 0; a < 4; a++) iters[a] = 0;
    for (int a = 0; a < 4 && !win; a++) {
        H.clear();
        bool exhausted = false;
        win = ddb_search(H.C, L, work, KS[a], iters[a], exhausted);
    }
    out4[0] = win ? 0 : (int)work.t.h; out4[1] = win ? (int)work.t.depth : 0;
    out4[2] = (int)work.t.jumps_lo | ((int)work.t.jumps_hi << 8); out4[3] = ddb_diamonds(H.C, L, work.alive);
    return 0;
}
}
