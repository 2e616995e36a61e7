// Device functions of PcgrlEnv.reset shared by k_reset (kernels_reset.h) and the in-kernel reset of k_stats.
// Part of the single translation unit pcgrl_abi.hip.
#pragma once
// ------------------------------------------------------------------------------------------
// Row bit planes from a tile byte map staged in LDS; lanes [0,G) of the wave each take one row.
// `row` is the map row this lane owns (0 .. G-1) or -1 for a lane that does not take part.
template <class MaskT>
__device__ __forceinline__ void planes_from_tiles(const PcgrlParams& P, const uint8_t* tiles, MaskT* planes_e, int row,
                                                  MaskT& m0, MaskT& m1, MaskT& m2, bool store = true) {
    const int W = P.width, H = P.height, G = P.group, NPL = P.nplanes;
    m0 = 0; m1 = 0; m2 = 0;
    if (NPL == 0) return;                 // smb keeps no bit planes
    if (row >= 0 && row < G) {
        const int lane = row;
        if (lane < H) {
            const uint8_t* row = tiles + lane * W;
            for (int x = 0; x < W; x++) {
                const MaskT t = row[x];
                m0 |= (t & 1) << x;
                m1 |= ((t >> 1) & 1) << x;
                m2 |= ((t >> 2) & 1) << x;
            }
        }
        if (store) {
            planes_e[lane * NPL] = m0;
            if (NPL > 1) { planes_e[lane * NPL + 1] = m1; planes_e[lane * NPL + 2] = m2; }
        }
    }
}

// The rows wave_reset_env cut out of its ballots, stored as the environment's plane rows (lanes with row in [0, G)) and handed
// back in the mask type of the caller.
template <class MaskT>
__device__ __forceinline__ void reset_rows_to_planes(const PcgrlParams& P, MaskT* planes_e, int row, const uint64_t r0, const uint64_t r1, const uint64_t r2,
                                                     MaskT& m0, MaskT& m1, MaskT& m2) {
    const int G = P.group, NPL = P.nplanes;
    m0 = (MaskT)r0; m1 = (MaskT)r1; m2 = (MaskT)r2;
    if (NPL == 0) { m0 = 0; m1 = 0; m2 = 0; return; }
    if (row >= 0 && row < G) {
        planes_e[row * NPL] = m0;
        if (NPL > 1) { planes_e[row * NPL + 1] = m1; planes_e[row * NPL + 2] = m2; }
    }
}

// numpy's randint(W) then randint(H) (masked rejection) on the next words of the ring staged in `mt`, by a whole wavefront: eight
// lanes make the next eight words at once (every operand is an old word), two ballots pick the first accepted x and the first
// accepted y after it; only if eight words do not hold both (probability < 1e-4 for any W, H) lane 0 goes on one word at a time.
// The consumed words are written to the staged ring; returns the cursor after them, x / y in every lane.
// `ring_g` (optional; wave_reset_env with only part of the ring staged): before lane 0 goes on one word at a time -- it may then read
// any word -- the rest of the ring is staged: every word outside the `dirty` ones from dirty0 on, which are newer in `mt` than in memory.
template <bool RESTAGE = false>
__device__ __forceinline__ int wave_draw_xy(uint32_t* mt, int cur, int W, int H, int lane, int& xv, int& yv, const uint32_t* ring_g = nullptr,
                                            int dirty0 = 0, int dirty = 0) {
    const uint32_t rx = (uint32_t)(W - 1), ry = (uint32_t)(H - 1);
    uint32_t mx = rx, my = ry;
    mx |= mx >> 1; mx |= mx >> 2; mx |= mx >> 4; mx |= mx >> 8; mx |= mx >> 16;
    my |= my >> 1; my |= my >> 2; my |= my >> 4; my |= my >> 8; my |= my >> 16;
    const int sl = mt_wrap(cur + (lane & 7));
    const uint32_t yw = mt_twist(mt[sl], mt[mt_wrap(sl + 1)], mt[mt_wrap(sl + PCGRL_MT_M)]);
    const uint32_t v = mt_temper(yw);
    const uint32_t okx = (uint32_t)__ballot(lane < 8 && (v & mx) <= rx) & 0xFFu;
    const uint32_t oky = (uint32_t)__ballot(lane < 8 && (v & my) <= ry) & 0xFFu;
    // index of the word that gives x (-1: randint(1) draws nothing), then of the word that gives y
    const int ix = rx == 0 ? -1 : (okx ? __ffs((int)okx) - 1 : 8);
    const uint32_t oky_after = ix >= 7 ? 0u : (ix < 0 ? oky : (oky & ~((2u << ix) - 1u)));       // words after ix (all of them when ix = -1)
    const int iy = ry == 0 ? ix : (ix >= 8 ? 8 : (oky_after ? __ffs((int)oky_after) - 1 : 8));
    __builtin_amdgcn_wave_barrier();
    if (iy < 8) {
        const int used = iy + 1;                                   // words consumed (0 when neither axis draws)
        if (lane < used) mt[sl] = yw;
        xv = rx == 0 ? 0 : (int)(__shfl(v, ix < 0 ? 0 : ix, 64) & mx);
        yv = ry == 0 ? 0 : (int)(__shfl(v, iy < 0 ? 0 : iy, 64) & my);
        cur = mt_wrap(cur + used);
    } else {
        if (RESTAGE && ring_g) {
#pragma clang loop unroll(disable)
            for (int i = lane; i < PCGRL_MT_N; i += 64) {
                int d = i - dirty0; d = d < 0 ? d + PCGRL_MT_N : d;
                if (d >= dirty) mt[i] = ring_g[i];
            }
            __builtin_amdgcn_wave_barrier();
        }
        int x = 0, y = 0;
        if (lane == 0) {
            x = mt_randint(mt, cur, W);
            y = mt_randint(mt, cur, H);
        }
        cur = __shfl(cur, 0, 64); xv = __shfl(x, 0, 64); yv = __shfl(y, 0, 64);
    }
    __builtin_amdgcn_wave_barrier();
    return cur;
}

// The whole-wavefront part of PcgrlEnv.reset (pcgrl_env.py:66-76) for environment e: new map (or the saved
// first map), cursor, both MT19937 rings, heatmap, counters, BinaryProblem.reset.  All 64 lanes call it;
// `mt` (624 words) and `tiles` (H*W bytes) are this wave's LDS scratch; the tile bytes of the new map are
// left in `tiles` for the caller to turn into row planes.
// `step_draws` (fused step kernel, narrow representation, an environment that is certain to be reset): the cursor move of the step
// that ended the episode (narrow_rep.py:104-113: randint(W), randint(H)) has not been drawn yet -- the block's update leaves it to
// the reset -- so its words are consumed here, from the staged ring, before the map is made (the position itself is not needed: the
// reset draws a new one).  On return the draw cache of the environment is rebuilt for the new cursor.
// What a row of the new map gets from 64 consecutive cells [base, base + 64) of the map's cell string, whose bit plane is the
// ballot q: row bits [0, W) are the cells [o, o + W), o = row * W.  (The caller masks the result to W bits at the end.)
__device__ __forceinline__ uint64_t reset_row_bits(uint64_t q, int base, int o, int W) {
    const int sh = o - base;
    if (sh >= 64 || sh + W <= 0) return 0ull;
    return sh >= 0 ? (q >> sh) : (q << -sh);
}

struct ResetRows { uint64_t m0, m1, m2; };      // bit planes of the tile ids of one row of the regenerated map

// `row` / `rows` (optional): the lanes that pass a row index 0 .. H-1 get that row of the new map as bit planes of the tile ids
// (bit x of m_b = bit b of the tile of cell (x, row)) -- cut out of the wave ballots of the tiles as they are made, so that no lane
// has to walk over the bytes of its row afterwards (planes_from_tiles: fourteen dependent LDS reads per row on a 14-column map);
// lanes with row < 0 or >= H get zeros.  `tiles` may be null when nobody needs the tile bytes in LDS.
// PARTIAL (compile time): stage only the stretches of the ring a small map's reset reads (below).  Off in the fused step kernel: its
// BASELINE shapes make 350-400 words -- all of the ring is read -- and the second path cost the kernel registers (C2 + 0.4 us a step).
template <int PROB, bool PARTIAL = true>
__device__ __forceinline__ void wave_reset_env(const PcgrlParams& P, const DevBufs& B, int e, int gen_map, uint32_t* mt,
                                               uint8_t* tiles, int lane, int step_draws = 0, int row = -1, ResetRows* rows = nullptr, int pend = 0) {
    const int W = P.width, H = P.height, cells = W * H;
    const bool want_rows = rows != nullptr;
    const int row_o = (row >= 0 && row < H) ? row * W : -(1 << 20);       // (a start far outside every batch: contributes nothing)
    uint64_t acc0 = 0ull, acc1 = 0ull, acc2 = 0ull;
    constexpr bool kThreePlanes = PROB != PCGRL_PROB_BINARY;
    uint32_t* ring_g = B.rng_rep + (size_t)e * PCGRL_MT_N;
    uint8_t* map_g = B.map + (size_t)e * cells;
    uint8_t* old_g = B.old_map + (size_t)e * cells;
    const int2 curs = reinterpret_cast<const int2*>(B.rng_cur)[e];
    // Small maps (round 6): a reset that makes fewer than 227 words -- the reach of MT19937's recurrence -- reads only old words, and only
    // those at the offsets 0 .. n + 1 and 397 .. 397 + n from where it starts: two short stretches of the ring are staged instead of
    // all 624 words (a 5 x 5 Sokoban level: 2 x 76 words; an 11 x 7 zelda map: 2 x 180).  The count includes eight words for each of
    // the (at most two) cursor draws; the one reset in a few hundred whose rejection sampling needs more stages the rest of the ring then
    // (wave_draw_xy).
    // (... and, with a draw cache to rebuild at the end, the nine words after the last draw)
    const int need = (step_draws ? 8 : 0) + (gen_map ? 2 * cells : 0) + 8 + (B.fifo ? PCGRL_FIFO_N + 1 : 0);
    const bool partial = PARTIAL && pend + need + 2 <= 226;
    int cur = curs.x;
    if (partial) {
        int base = cur - pend; base = base < 0 ? base + PCGRL_MT_N : base;
        const int len = pend + need + 2;
        uint32_t ra[4], rb[4];                                 // (every load first, then the LDS stores: one round trip; len <= 226)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = j * 64 + lane;
            ra[j] = k < len ? ring_g[mt_wrap(base + k)] : 0u;
            rb[j] = k < len ? ring_g[mt_wrap(mt_wrap(base + PCGRL_MT_M) + k)] : 0u;
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = j * 64 + lane;
            if (k < len) { mt[mt_wrap(base + k)] = ra[j]; mt[mt_wrap(mt_wrap(base + PCGRL_MT_M) + k)] = rb[j]; }
        }
    } else {   // the ring: every load first, then the LDS stores (one round trip; 624 = 9 x 64 + 48)
        uint32_t rw[10];
#pragma unroll
        for (int j = 0; j < 10; j++) rw[j] = (j < 9 || lane < PCGRL_MT_N - 9 * 64) ? ring_g[j * 64 + lane] : 0u;
        asm volatile("" ::: "memory");      // (keeps the compiler from pairing each load with its store)
#pragma unroll
        for (int j = 0; j < 10; j++) if (j < 9 || lane < PCGRL_MT_N - 9 * 64) mt[j * 64 + lane] = rw[j];
    }
    // BinaryProblem.reset (binary_prob.py:68-72) draws one double = two words from the *problem* stream
    // after the map is made.  Its five operand words are fetched now, by five lanes, off the critical path.
    const bool prob_draw = PROB == PCGRL_PROB_BINARY && P.random_probs;
    uint32_t pw = 0;
    if (prob_draw && lane < 5) {
        const int off = lane < 3 ? lane : PCGRL_MT_M + (lane - 3);
        int sl = curs.y + off; sl = sl >= PCGRL_MT_N ? sl - PCGRL_MT_N : sl;
        pw = B.rng_prob[(size_t)e * PCGRL_MT_N + sl];
    }
    if (pend > 0) {
        // (fused step kernel, an episode end nobody saw coming: the environment's cursor already counts `pend` draws of this step whose
        //  words are still in its draw cache and not in its ring -- they are patched into the staged ring)
        __builtin_amdgcn_wave_barrier();
        if (lane < pend) {
            int sl = cur - pend + lane; sl = sl < 0 ? sl + PCGRL_MT_N : sl;
            mt[sl] = B.fifo[(size_t)e * PCGRL_FIFO_N + lane];
        }
    }
    // the words this reset puts into the ring are the `dirty` ones from dirty0 on (the lazy ring makes one word per draw, in place): only
    // those go back to memory at the end -- 352 of 624 for an 11 x 16 map, 400 for 14 x 14 -- instead of the whole ring
    int dirty0 = cur - pend; dirty0 = dirty0 < 0 ? dirty0 + PCGRL_MT_N : dirty0;
    int dirty = pend;
    __builtin_amdgcn_wave_barrier();
    TL(13);
    if (step_draws) {
        int sx, sy;
        const int c0 = cur;
        cur = wave_draw_xy<PARTIAL>(mt, cur, W, H, lane, sx, sy, partial ? ring_g : (const uint32_t*)nullptr, dirty0, dirty);
        dirty += cur >= c0 ? cur - c0 : cur - c0 + PCGRL_MT_N;
    }
    if (gen_map) {
        // helper.py:310-312 gen_random_map == RandomState.choice(keys, (H,W), p), Representation.reset
        // the number of tiles is a property of the problem: constant indices only -- a run-time index into the by-value
        // parameter block makes the compiler keep a copy of the whole block in scratch memory
        constexpr int NT = PROB == PCGRL_PROB_BINARY ? 2 : PROB == PCGRL_PROB_SOKOBAN ? 5 : (PROB == PCGRL_PROB_DDAVE || PROB == PCGRL_PROB_SMB) ? 7 : 8;
        double cdf[NT];
        if (PROB == PCGRL_PROB_BINARY) {
            double p[2] = {B.tile_p[2 * e], B.tile_p[2 * e + 1]};
            pcgrl_build_cdf(p, 2, cdf);
        } else {
#pragma unroll
            for (int i = 0; i < NT; i++) cdf[i] = P.cdf[i];
        }
        // Cell c draws ring words 2c, 2c + 1 of this episode.  MT19937's recurrence x[k + 624] = f(x[k], x[k + 1], x[k + 397])
        // reaches back 227 words, so up to 226 consecutive words can be made from old words alone: a round makes the words of 113
        // cells -- every lane those of cell c0 + lane, lanes 0..48 also those of cell c0 + 64 + lane -- with all reads before all
        // writes.  (Two rounds for a 14 x 14 map where one cell per lane took four: the rounds are a chain of LDS round trips.)
        for (int c0 = 0; c0 < cells; c0 += 113) {
            const int rem = cells - c0;
            const int nA = rem < 64 ? rem : 64, nB = rem <= 64 ? 0 : (rem - 64 < 49 ? rem - 64 : 49);
            int sA = cur + 2 * lane; sA = sA >= PCGRL_MT_N ? sA - PCGRL_MT_N : sA;
            const int sB = mt_wrap(sA + 128);
            const uint32_t a0 = mt[sA], a1 = mt[mt_wrap(sA + 1)], a2 = mt[mt_wrap(sA + 2)];
            const uint32_t am0 = mt[mt_wrap(sA + PCGRL_MT_M)], am1 = mt[mt_wrap(sA + PCGRL_MT_M + 1)];
            uint32_t b0 = 0, b1 = 0, b2 = 0, bm0 = 0, bm1 = 0;
            if (nB > 0) {      // wave-uniform
                b0 = mt[sB]; b1 = mt[mt_wrap(sB + 1)]; b2 = mt[mt_wrap(sB + 2)];
                bm0 = mt[mt_wrap(sB + PCGRL_MT_M)]; bm1 = mt[mt_wrap(sB + PCGRL_MT_M + 1)];
            }
            const uint32_t yaA = mt_twist(a0, a1, am0), ybA = mt_twist(a1, a2, am1);
            const uint32_t yaB = mt_twist(b0, b1, bm0), ybB = mt_twist(b1, b2, bm1);
            __builtin_amdgcn_wave_barrier();
            int tA = 0, tB = 0;
            if (lane < nA) {
                mt[sA] = yaA;
                mt[mt_wrap(sA + 1)] = ybA;
                tA = pcgrl_pick_tile_c<NT>(cdf, mt_to_double(mt_temper(yaA), mt_temper(ybA)));
                const int c = c0 + lane;
                if (tiles) tiles[c] = (uint8_t)tA;          // (k_big passes no staging area: its maps are read back from memory)
                map_g[c] = (uint8_t)tA;
                old_g[c] = (uint8_t)tA;
            }
            if (lane < nB) {
                mt[sB] = yaB;
                mt[mt_wrap(sB + 1)] = ybB;
                tB = pcgrl_pick_tile_c<NT>(cdf, mt_to_double(mt_temper(yaB), mt_temper(ybB)));
                const int c = c0 + 64 + lane;
                if (tiles) tiles[c] = (uint8_t)tB;
                map_g[c] = (uint8_t)tB;
                old_g[c] = (uint8_t)tB;
            }
            if (want_rows) {
                acc0 |= reset_row_bits(__ballot(tA & 1), c0, row_o, W);
                if (kThreePlanes) { acc1 |= reset_row_bits(__ballot(tA & 2), c0, row_o, W); acc2 |= reset_row_bits(__ballot(tA & 4), c0, row_o, W); }
                if (nB > 0) {
                    acc0 |= reset_row_bits(__ballot(tB & 1), c0 + 64, row_o, W);
                    if (kThreePlanes) { acc1 |= reset_row_bits(__ballot(tB & 2), c0 + 64, row_o, W); acc2 |= reset_row_bits(__ballot(tB & 4), c0 + 64, row_o, W); }
                }
            }
            __builtin_amdgcn_wave_barrier();
            cur += 2 * (nA + nB); cur = cur >= PCGRL_MT_N ? cur - PCGRL_MT_N : cur;
            dirty += 2 * (nA + nB);
        }
    } else {
        // representation.py:44-45: restore the first map of this environment
        for (int c0 = 0; c0 < cells; c0 += 64) {
            const int c = c0 + lane;
            int t = 0;
            if (c < cells) { t = old_g[c]; if (tiles) tiles[c] = (uint8_t)t; map_g[c] = (uint8_t)t; }
            if (want_rows) {
                acc0 |= reset_row_bits(__ballot(t & 1), c0, row_o, W);
                if (kThreePlanes) { acc1 |= reset_row_bits(__ballot(t & 2), c0, row_o, W); acc2 |= reset_row_bits(__ballot(t & 4), c0, row_o, W); }
            }
        }
    }
    if (want_rows) {
        const uint64_t wm = W >= 64 ? ~0ull : ((1ull << W) - 1ull);
        rows->m0 = acc0 & wm; rows->m1 = acc1 & wm; rows->m2 = acc2 & wm;
    }
    __builtin_amdgcn_wave_barrier();
    TL(14);
    if (P.rep != PCGRL_REP_WIDE) {   // narrow_rep.py:28-31, turtle_rep.py:30-33: x = randint(W), y = randint(H)
        int xv, yv;
        const int c0 = cur;
        cur = wave_draw_xy<PARTIAL>(mt, cur, W, H, lane, xv, yv, partial ? ring_g : (const uint32_t*)nullptr, dirty0, dirty);
        dirty += cur >= c0 ? cur - c0 : cur - c0 + PCGRL_MT_N;
        if (lane == 0) reinterpret_cast<uchar2*>(B.pos)[e] = make_uchar2((unsigned char)xv, (unsigned char)yv);
    }
    __builtin_amdgcn_wave_barrier();
    TL(15);
    if (B.fifo && lane < PCGRL_FIFO_N) {     // the words of the next draws, computed ahead (the ring itself stays lazy)
        const int sl = mt_wrap(cur + lane);
        B.fifo[(size_t)e * PCGRL_FIFO_N + lane] = mt_twist(mt[sl], mt[mt_wrap(sl + 1)], mt[mt_wrap(sl + PCGRL_MT_M)]);
        if (lane == 0) B.fifo_tag[e] = cur;
    }
    {   // (one loop: a whole ring is the stretch of 624 words from dirty0 on)
        const int nd = dirty < PCGRL_MT_N ? dirty : PCGRL_MT_N;
#pragma clang loop unroll(disable)
        for (int k = lane; k < nd; k += 64) {
            int i = dirty0 + k; i = i >= PCGRL_MT_N ? i - PCGRL_MT_N : i;
            ring_g[i] = mt[i];
        }
    }
    uint16_t* heat_g = B.heat + (size_t)e * cells;
    for (int c = lane; c < cells; c += 64) heat_g[c] = 0;        // pcgrl_env.py:72
    if (lane == 0) {
        B.rng_cur[2 * e] = cur;
        reinterpret_cast<int2*>(B.counters)[e] = make_int2(0, 0);   // pcgrl_env.py:67-68
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
    TL(16);
}



// The same reset by a whole block (k_stats_wide: binary maps of up to 64 x 64 cells, 8 192 MT19937 words per map).  One
// wavefront makes 128 words per round -- 64 rounds for such a map, ~38 us, with the other wavefronts of the block waiting for
// it: the longest chain of a C5 step.  Here (round 3; the in-kernel timeline showed 14 us for the map and 7 us for turning its
// bytes into row masks):
//   1. the raw words.  MT19937's recurrence x[k+624] = f(x[k], x[k+1], x[k+397]) reaches back 227 words; a thread that has just
//      made word k also holds the one operand of word k+227 that is new (x[k+624]), so 227 threads make TWO words each per round
//      -- 454 words between two barriers, 19 rounds for 8 192 words -- and nothing else happens in a round: the words go to a
//      buffer in LDS (`raw`, 2 * cells words: the per-wavefront scratch sets the block reset does not use);
//   2. the tiles, a thread per cell: tempering, numpy's 53-bit double, the cdf comparison (RandomState.choice, helper.py:310-312),
//      byte stores to map / first map -- all of it parallel over the cells; a wavefront's 64 cells give 64 bits of the map's
//      bit string with one ballot (`cellbits`);
//   3. the row masks: lane r cuts its W bits out of that bit string (no loop over the bytes of a row).
// Same draws, same order, same results as wave_reset_env; `s_cur`: one shared word.  Every thread of the block calls this; it
// ends with a block barrier.  Returns the mask of row `lane` (bit x = tile bit 0 of cell (x, lane)) in every wavefront;
// `tiles` still receives the tile bytes.
template <int PROB, int NTHREADS, class MaskT>
__device__ __forceinline__ MaskT block_reset_env(const PcgrlParams& P, const DevBufs& B, int e, int gen_map, uint32_t* mt, uint8_t* tiles, uint32_t* raw,
                                                 uint64_t* cellbits, int* s_cur) {
    static_assert(NTHREADS >= 256, "227 generating threads");
    static_assert(PROB == PCGRL_PROB_BINARY, "one plane: the tall-map path of the binary problem");
    constexpr int RW = 227;                                  // words of one reach of the recurrence; a round makes 2 * RW
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int W = P.width, H = P.height, cells = W * H;
    uint32_t* ring_g = B.rng_rep + (size_t)e * PCGRL_MT_N;
    uint8_t* map_g = B.map + (size_t)e * cells;
    uint8_t* old_g = B.old_map + (size_t)e * cells;
    const int2 curs = reinterpret_cast<const int2*>(B.rng_cur)[e];
    int cur = curs.x;
    for (int i = tid; i < PCGRL_MT_N; i += NTHREADS) mt[i] = ring_g[i];
    const bool prob_draw = PROB == PCGRL_PROB_BINARY && P.random_probs;
    uint32_t pw = 0;
    if (prob_draw && tid < 5) {                               // BinaryProblem.reset's five operand words (see wave_reset_env)
        const int off = tid < 3 ? tid : PCGRL_MT_M + (tid - 3);
        int sl = curs.y + off; sl = sl >= PCGRL_MT_N ? sl - PCGRL_MT_N : sl;
        pw = B.rng_prob[(size_t)e * PCGRL_MT_N + sl];
    }
    const int nchunks = (cells + 63) >> 6;
    if (tid <= nchunks) cellbits[tid] = 0ull;                 // (one word beyond the last: step 3 reads two)
    __syncthreads();
    if (gen_map) {
        const int nwords = 2 * cells;
        for (int w0 = 0; w0 < nwords; w0 += 2 * RW) {
            const bool on1 = tid < RW && w0 + tid < nwords, on2 = tid < RW && w0 + RW + tid < nwords;
            const int s1 = mt_wrap(cur + (tid < RW ? tid : 0)), s2 = mt_wrap(s1 + RW);
            uint32_t y1 = 0, y2 = 0;
            if (on1) {
                const uint32_t a0 = mt[s1], a1 = mt[mt_wrap(s1 + 1)], am = mt[mt_wrap(s1 + PCGRL_MT_M)];
                const uint32_t b0 = mt[s2], b1 = mt[mt_wrap(s2 + 1)];
                y1 = mt_twist(a0, a1, am);
                y2 = mt_twist(b0, b1, y1);                    // x[(k + 227) + 397] = x[k + 624]: the word just made
            }
            __syncthreads();                                  // every operand was read before any slot is rewritten
            if (on1) { mt[s1] = y1; raw[w0 + tid] = y1; }
            if (on2) { mt[s2] = y2; raw[w0 + RW + tid] = y2; }
            const int adv = (nwords - w0) < 2 * RW ? (nwords - w0) : 2 * RW;
            cur = mt_wrap(mt_wrap(cur + (adv > RW ? RW : adv)) + (adv > RW ? adv - RW : 0));
            __syncthreads();
        }
        double cdf[2];
        {
            double p[2] = {B.tile_p[2 * e], B.tile_p[2 * e + 1]};
            pcgrl_build_cdf(p, 2, cdf);
        }
        for (int ch = wv; ch < nchunks; ch += NTHREADS / 64) {
            const int c = ch * 64 + lane;                     // cell c draws words 2c, 2c + 1 (helper.py:310-312, RandomState.choice)
            uint8_t t = 0;
            if (c < cells) {
                const double u = mt_to_double(mt_temper(raw[2 * c]), mt_temper(raw[2 * c + 1]));
                t = (uint8_t)pcgrl_pick_tile_c<2>(cdf, u);
                tiles[c] = t; map_g[c] = t; old_g[c] = t;
            }
            const uint64_t bits = __ballot(t & 1);
            if (lane == 0) cellbits[ch] = bits;
        }
    } else {
        for (int ch = wv; ch < nchunks; ch += NTHREADS / 64) {
            const int c = ch * 64 + lane;
            uint8_t t = 0;
            if (c < cells) { t = old_g[c]; tiles[c] = t; map_g[c] = t; }
            const uint64_t bits = __ballot(t & 1);
            if (lane == 0) cellbits[ch] = bits;
        }
    }
    __syncthreads();
    MaskT row = 0;
    if (lane < H) {
        const int o = lane * W, wd = o >> 6, sh = o & 63;
        const uint64_t lo = cellbits[wd] >> sh, hi = sh ? cellbits[wd + 1] << (64 - sh) : 0ull;
        row = (MaskT)((lo | hi) & (W >= 64 ? ~0ull : ((1ull << W) - 1ull)));
    }
    if (P.rep != PCGRL_REP_WIDE) {   // narrow_rep.py:28-31, turtle_rep.py:30-33
        if (tid == 0) {
            const int x = mt_randint(mt, cur, W);
            const int y = mt_randint(mt, cur, H);
            reinterpret_cast<uchar2*>(B.pos)[e] = make_uchar2((unsigned char)x, (unsigned char)y);
            *s_cur = cur;
        }
        __syncthreads();
        cur = *s_cur;
    }
    if (B.fifo && tid < PCGRL_FIFO_N) {
        const int sl = mt_wrap(cur + tid);
        B.fifo[(size_t)e * PCGRL_FIFO_N + tid] = mt_twist(mt[sl], mt[mt_wrap(sl + 1)], mt[mt_wrap(sl + PCGRL_MT_M)]);
        if (tid == 0) B.fifo_tag[e] = cur;
    }
    for (int i = tid; i < PCGRL_MT_N; i += NTHREADS) ring_g[i] = mt[i];
    uint16_t* heat_g = B.heat + (size_t)e * cells;
    for (int c = tid; c < cells; c += NTHREADS) heat_g[c] = 0;        // pcgrl_env.py:72
    if (tid == 0) {
        B.rng_cur[2 * e] = cur;
        reinterpret_cast<int2*>(B.counters)[e] = make_int2(0, 0);   // pcgrl_env.py:67-68
    }
    if (prob_draw && tid < 64) {
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
    __syncthreads();
    return row;
}
