// CPU lane-group simulator for gym_pcgrl_amd/csrc/pcgrl_algos.h and mt19937.h.  TEST ONLY.
//
// Instantiates the same bitboard templates the GPU kernels use with a backend whose "lane
// group" is a plain array of G row masks, so the algorithm logic (not the DPP plumbing) can be
// checked against the oracle and the golden fixtures without a GPU.  Never loaded by the product.
#include <stdint.h>
#include <string.h>

#include "../../gym_pcgrl_amd/csrc/mt19937.h"
#include "../../gym_pcgrl_amd/csrc/pcgrl_algos.h"

template <class T, int G>
struct SimVec {
    T v[G];
    SimVec() { for (int i = 0; i < G; i++) v[i] = 0; }
    SimVec(T x) { for (int i = 0; i < G; i++) v[i] = x; }
};
#define SV_BIN(op) \
    template <class T, int G> SimVec<T, G> operator op(const SimVec<T, G>& a, const SimVec<T, G>& b) { \
        SimVec<T, G> r; for (int i = 0; i < G; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
SV_BIN(&) SV_BIN(|) SV_BIN(^)
template <class T, int G> SimVec<T, G> operator~(const SimVec<T, G>& a) { SimVec<T, G> r; for (int i = 0; i < G; i++) r.v[i] = ~a.v[i]; return r; }
template <class T, int G> SimVec<T, G> operator<<(const SimVec<T, G>& a, int s) { SimVec<T, G> r; for (int i = 0; i < G; i++) r.v[i] = a.v[i] << s; return r; }
template <class T, int G> SimVec<T, G> operator>>(const SimVec<T, G>& a, int s) { SimVec<T, G> r; for (int i = 0; i < G; i++) r.v[i] = a.v[i] >> s; return r; }

static long g_sim_iters = 0;
template <int G, class T>
struct SimGroup {
    typedef SimVec<T, G> mask_t;
    mask_t up(const mask_t& m) const { mask_t r; for (int i = 1; i < G; i++) r.v[i] = m.v[i - 1]; return r; }
    mask_t down(const mask_t& m) const { mask_t r; for (int i = 0; i + 1 < G; i++) r.v[i] = m.v[i + 1]; return r; }
    bool any(const mask_t& m) const { for (int i = 0; i < G; i++) if (m.v[i]) return true; return false; }
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
    } else {
        *need_solver = sokoban_stats(g, P, b0, b1, b2, valid, out) ? 1 : 0;
    }
}

extern "C" {
long sim_iters_reset() { long v = g_sim_iters; g_sim_iters = 0; return v; }
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
// the reset kernel's 64-cells-per-round parallel generation, emulated lane by lane (reads before writes)
void sim_mt_mapgen(const uint32_t* key624, const double* prob, int ntiles, int w, int h, uint8_t* tiles, int* xy, uint32_t* ring_out, int* cur_out) {
    uint32_t mt[624]; memcpy(mt, key624, sizeof(mt)); int cur = 0;
    double cdf[8]; pcgrl_build_cdf(prob, ntiles, cdf);
    int cells = w * h;
    for (int c0 = 0; c0 < cells; c0 += 64) {
        uint32_t ya[64], yb[64]; int ss[64];
        for (int lane = 0; lane < 64; lane++) {
            int s = cur + 2 * lane; s = s >= 624 ? s - 624 : s; ss[lane] = s;
            uint32_t x0 = mt[s], x1 = mt[mt_wrap(s + 1)], x2 = mt[mt_wrap(s + 2)];
            uint32_t xm0 = mt[mt_wrap(s + 397)], xm1 = mt[mt_wrap(s + 398)];
            ya[lane] = mt_twist(x0, x1, xm0); yb[lane] = mt_twist(x1, x2, xm1);
        }
        for (int lane = 0; lane < 64; lane++) {
            int c = c0 + lane;
            if (c < cells) {
                mt[ss[lane]] = ya[lane]; mt[mt_wrap(ss[lane] + 1)] = yb[lane];
                tiles[c] = (uint8_t)pcgrl_pick_tile(cdf, ntiles, mt_to_double(mt_temper(ya[lane]), mt_temper(yb[lane])));
            }
        }
        int adv = 2 * ((cells - c0) < 64 ? (cells - c0) : 64);
        cur += adv; cur = cur >= 624 ? cur - 624 : cur;
    }
    xy[0] = mt_randint(mt, cur, w); xy[1] = mt_randint(mt, cur, h);
    memcpy(ring_out, mt, sizeof(mt)); *cur_out = cur;
}
}
