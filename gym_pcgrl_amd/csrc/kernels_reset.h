// k_reset: PcgrlEnv.reset (map generation + start stats), one wavefront per environment.
// Part of the single translation unit pcgrl_abi.hip (see its header comment for the overall picture).
#pragma once
// ------------------------------------------------------------------------------------------
// Row bit planes from a tile byte map staged in LDS; lanes [0,G) of the wave each take one row.
template <class MaskT>
__device__ __forceinline__ void planes_from_tiles(const PcgrlParams& P, const uint8_t* tiles, MaskT* planes_e, int lane,
                                                  MaskT& m0, MaskT& m1, MaskT& m2) {
    const int W = P.width, H = P.height, G = P.group, NPL = P.nplanes;
    m0 = 0; m1 = 0; m2 = 0;
    if (lane < G) {
        if (lane < H) {
            const uint8_t* row = tiles + lane * W;
            for (int x = 0; x < W; x++) {
                const MaskT t = row[x];
                m0 |= (t & 1) << x;
                m1 |= ((t >> 1) & 1) << x;
                m2 |= ((t >> 2) & 1) << x;
            }
        }
        planes_e[lane] = m0;
        if (NPL > 1) { planes_e[G + lane] = m1; planes_e[2 * G + lane] = m2; }
    }
}

// k_reset: wavefront per environment to reset -- PcgrlEnv.reset (pcgrl_env.py:66-76) including the start stats
template <int PROB, int G, class MaskT>
__global__ __launch_bounds__(PCGRL_BLOCK) void k_reset(PcgrlParams P, DevBufs B, int parity, int gen_map, int clear_parity) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int W = P.width, H = P.height, cells = W * H;
    const int tiles_bytes = (cells + 15) & ~15;
    uint32_t* mt = reinterpret_cast<uint32_t*>(smem + (size_t)wv * (PCGRL_MT_N * 4 + tiles_bytes));
    uint8_t* tiles = reinterpret_cast<uint8_t*>(mt + PCGRL_MT_N);
    __shared__ int s_pref[WL_NSHARD + 1];
    const int n = wl_load_prefix(B, parity, WL_RST, s_pref);
    for (int item = blockIdx.x * 4 + wv; item < n; item += gridDim.x * 4) {
        const int e = wl_get(B, WL_RST, s_pref, item);
        uint32_t* ring_g = B.rng_rep + (size_t)e * PCGRL_MT_N;
        uint8_t* map_g = B.map + (size_t)e * cells;
        uint8_t* old_g = B.old_map + (size_t)e * cells;
        const int2 curs = reinterpret_cast<const int2*>(B.rng_cur)[e];
        int cur = curs.x;
        for (int i = lane; i < PCGRL_MT_N; i += 64) mt[i] = ring_g[i];
        // BinaryProblem.reset (binary_prob.py:68-72) draws one double = two words from the *problem* stream
        // after the map is made.  Its five operand words are fetched now, by five lanes, off the critical path.
        const bool prob_draw = PROB == PCGRL_PROB_BINARY && P.random_probs;
        uint32_t pw = 0;
        if (prob_draw && lane < 5) {
            const int off = lane < 3 ? lane : PCGRL_MT_M + (lane - 3);
            int sl = curs.y + off; sl = sl >= PCGRL_MT_N ? sl - PCGRL_MT_N : sl;
            pw = B.rng_prob[(size_t)e * PCGRL_MT_N + sl];
        }
        __builtin_amdgcn_wave_barrier();
        if (gen_map) {
            // helper.py:310-312 gen_random_map == RandomState.choice(keys, (H,W), p), Representation.reset
            double cdf[PCGRL_MAX_TILES];
            if (P.prob == PCGRL_PROB_BINARY) {
                double p[2] = {B.tile_p[2 * e], B.tile_p[2 * e + 1]};
                pcgrl_build_cdf(p, 2, cdf);
            } else {
                for (int i = 0; i < P.ntiles; i++) cdf[i] = P.cdf[i];
            }
            for (int c0 = 0; c0 < cells; c0 += 64) {
                // cell c draws ring words 2c, 2c+1 of this episode: 128 new words per round, every
                // operand is an *old* word (distance 397 > 128), so all reads come before all writes
                const int c = c0 + lane;
                int s = cur + 2 * lane; s = s >= PCGRL_MT_N ? s - PCGRL_MT_N : s;
                const uint32_t x0 = mt[s], x1 = mt[mt_wrap(s + 1)], x2 = mt[mt_wrap(s + 2)];
                const uint32_t xm0 = mt[mt_wrap(s + PCGRL_MT_M)], xm1 = mt[mt_wrap(s + PCGRL_MT_M + 1)];
                const uint32_t ya = mt_twist(x0, x1, xm0), yb = mt_twist(x1, x2, xm1);
                __builtin_amdgcn_wave_barrier();
                if (c < cells) {
                    mt[s] = ya;
                    mt[mt_wrap(s + 1)] = yb;
                    const double u = mt_to_double(mt_temper(ya), mt_temper(yb));
                    const uint8_t t = (uint8_t)pcgrl_pick_tile(cdf, P.ntiles, u);
                    tiles[c] = t;
                    map_g[c] = t;
                    old_g[c] = t;
                }
                __builtin_amdgcn_wave_barrier();
                const int adv = 2 * ((cells - c0) < 64 ? (cells - c0) : 64);
                cur += adv; cur = cur >= PCGRL_MT_N ? cur - PCGRL_MT_N : cur;
            }
        } else {
            // representation.py:44-45: restore the first map of this environment
            for (int c = lane; c < cells; c += 64) { const uint8_t t = old_g[c]; tiles[c] = t; map_g[c] = t; }
        }
        __builtin_amdgcn_wave_barrier();
        if (P.rep != PCGRL_REP_WIDE) {   // narrow_rep.py:28-31, turtle_rep.py:30-33
            int x = 0, y = 0;
            if (lane == 0) {
                x = mt_randint(mt, cur, W);
                y = mt_randint(mt, cur, H);
                reinterpret_cast<uchar2*>(B.pos)[e] = make_uchar2((unsigned char)x, (unsigned char)y);
            }
            cur = __shfl(cur, 0, 64);
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < PCGRL_MT_N; i += 64) ring_g[i] = mt[i];
        MaskT b0, b1, b2;
        planes_from_tiles<MaskT>(P, tiles, reinterpret_cast<MaskT*>(B.planes) + (size_t)e * P.nplanes * P.group, lane, b0, b1, b2);
        uint16_t* heat_g = B.heat + (size_t)e * cells;
        for (int c = lane; c < cells; c += 64) heat_g[c] = 0;        // pcgrl_env.py:72
        // start stats (pcgrl_env.py:70-71, problem.py:45-46): the rows are already in registers.  With
        // 16-lane groups only the first DPP row holds the map; the other rows see an empty map and idle.
        DevGroup<G, MaskT> g;
        const MaskT valid = (lane < G) ? row_valid<MaskT>(g.lane, W, H) : (MaskT)0;
        int32_t st[PCGRL_MAX_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool need_solver = compute_item_stats<PROB>(g, P, b0, b1, b2, valid, st);
        if (lane == 0) {
            B.rng_cur[2 * e] = cur;
            reinterpret_cast<int2*>(B.counters)[e] = make_int2(0, 0);   // pcgrl_env.py:67-68
            finish_or_park(P, B, e, st, need_solver, MODE_START, parity, item & (WL_NSHARD - 1));
        }
        if (prob_draw) {
            // two consecutive lazy-ring draws at cursor c: word c uses (c, c+1, c+397), word c+1 uses
            // (c+1, c+2, c+398); none of those operands is the slot the first draw rewrites
            const uint32_t x0 = __shfl(pw, 0, 64), x1 = __shfl(pw, 1, 64), x2 = __shfl(pw, 2, 64);
            const uint32_t xm0 = __shfl(pw, 3, 64), xm1 = __shfl(pw, 4, 64);
            if (lane == 0) {
                const uint32_t ya = mt_twist(x0, x1, xm0), yb = mt_twist(x1, x2, xm1);
                uint32_t* ring_p = B.rng_prob + (size_t)e * PCGRL_MT_N;
                ring_p[curs.y] = ya;
                ring_p[mt_wrap(curs.y + 1)] = yb;
                B.rng_cur[2 * e + 1] = mt_wrap(mt_wrap(curs.y + 1) + 1);
                const double pe = mt_to_double(mt_temper(ya), mt_temper(yb));
                B.tile_p[2 * e] = pe;
                B.tile_p[2 * e + 1] = 1 - pe;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// set_maps support: byte maps -> planes (wavefront per environment)
template <class MaskT>
__global__ __launch_bounds__(PCGRL_BLOCK) void k_planes_from_map(PcgrlParams P, DevBufs B, const uint8_t* __restrict__ src) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cells = P.width * P.height;
    uint8_t* tiles = smem + (size_t)wv * ((cells + 15) & ~15);
    for (int e = blockIdx.x * 4 + wv; e < P.num_envs; e += gridDim.x * 4) {
        for (int c = lane; c < cells; c += 64) {
            uint8_t t = src[(size_t)e * cells + c];
            t = t < P.ntiles ? t : (uint8_t)(P.ntiles - 1);
            tiles[c] = t;
            B.map[(size_t)e * cells + c] = t;
        }
        __builtin_amdgcn_wave_barrier();
        MaskT b0, b1, b2;
        planes_from_tiles<MaskT>(P, tiles, reinterpret_cast<MaskT*>(B.planes) + (size_t)e * P.nplanes * P.group, lane, b0, b1, b2);
        __builtin_amdgcn_wave_barrier();
    }
}
