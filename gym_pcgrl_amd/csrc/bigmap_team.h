// Binary maps beyond 64 x 64, the full recomputation by ALL wavefronts of a block (k_big; bigmap.h has the one-wavefront form).
// Part of the single translation unit pcgrl_abi.hip.
//
// A step of 4 096 environments on 100 x 100 maps has ~55 full recomputations (a change in or next to the champion component, a
// reset) among ~1 300 cheap incremental updates, and the step ends with the slowest of them: one wavefront going through ~230
// components and a dozen double sweeps, ~0.5 ms, while most of the GPU idles.  Here the wavefronts of the block share the map
// (helper.py:197-207 calc_num_regions + :250-264 calc_longest_path: a count and a maximum over the components, so the order and the
// worker do not matter):
//   planes, tiny components   the words of the masks are dealt out to the threads, block barriers between the passes
//   phase A                   the rows are cut into one band per wavefront; a wavefront takes the components that lie INSIDE its band
//                             (fill confined to the band, in a 64 x 64 register window: bigmap.h) -- it alone reads and writes the
//                             band's rows of `rest` and `cross`, so there is nothing to lock; a piece that touches the band's border
//                             with the map going on behind it, leaves the window, or meets an earlier such piece is moved to `cross`
//   phase B                   wavefront 0 goes through `cross` -- whole components again -- the way the one-wavefront form does
//   sweeps                    every component whose size calls for a sweep was put on its finder's list; now every wavefront takes
//                             the largest one still on any list (compare-and-swap on the entry), sweeps it, raises the shared
//                             maximum -- until the largest one left cannot beat it
//   champion                  the best window component (64-bit maximum of sweep << 32 | seed) against the best component of the
//                             word-array path (kept in `champ` as it was found); which of several components with the same sweep
//                             value becomes the champion may differ from run to run -- big_incremental needs A champion, any one
#pragma once

struct BigTeamShared {
    int n_iso, n_dom, n_tri;         // tiny components (closed forms), summed over the threads
    int regions;                     // components counted in phases A and B
    int path;                        // running maximum of the sweeps
    int lds_path, lds_has;           // best component of the word-array path (phase B): its sweep, and whether `champ` holds one
    int ncand[8];                    // entries on each wavefront's list
    unsigned long long win;          // best window component: sweep << 32 | word index << 8 | bit
};
#define BIG_TEAM_MAX_WAVES 8

__device__ __forceinline__ uint64_t big_window_row(const uint64_t* a, const BigGeom& G, const BigWindow& Wd, int row) {
    const int i = row * G.KW + Wd.kx;
    uint64_t v = a[i] >> Wd.sh;
    if (Wd.sh != 0 && Wd.kx + 1 < G.KW) v |= a[i + 1] << (64 - Wd.sh);
    return v;
}

// One window sweep for the team: the component of seed (i0, b0) -- it fitted a window when it was counted -- swept against the shared
// maximum; a result raises it and competes for the champion.
__device__ __forceinline__ void big_team_sweep(const uint64_t* pass, const BigGeom& G, BigTeamShared& T, int i0, int b0, int lane) {
    const int r0 = big_row(G, i0), c0 = 64 * (i0 - r0 * G.KW) + b0;
    BigWindow Wd = big_window_at(G, r0, c0);
    uint64_t cw;
    big_window_component(pass, G, Wd, c0, lane, cw);
    DevGroup<64, uint64_t> g;
    const int snap = __hip_atomic_load(&T.path, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int e2 = pcg_double_sweep(g, cw, snap);                  // 0: cannot beat the maximum as it was when the sweep began
    if (e2 > snap && lane == 0) {
        atomicMax(&T.path, e2);
        atomicMax(&T.win, ((unsigned long long)(unsigned)e2 << 32) | ((unsigned long long)(unsigned)i0 << 8) | (unsigned long long)(unsigned)b0);
    }
}

// Every wavefront of the block calls this for the same map `m`.  ar0: wavefront 0's masks (shared: pass = ar0, rest = ar0 + NW, two
// scratch masks behind them, champ = ar0 + 6 NW), ar1: wavefront 1's (its first mask is `cross`), mine: this wavefront's own (its
// comp / X / Y / Z for the word-array path), lists: wavefront 0's MT19937 area (the lists of put-off sweeps: `stride` words from one
// wavefront's to the next, `cap` entries each).  Returns regions / path / has-champion in every thread; `champ` = ar0 + 6 NW holds the
// champion's rows.
__device__ __forceinline__ void big_team_binary(const uint8_t* __restrict__ m, const BigGeom& G, uint64_t* ar0, uint64_t* ar1, uint64_t* mine, uint64_t* lists,
                                                size_t stride, int cap, BigTeamShared& T, int wv, int nwv, int lane, int& regions, int& path, int& has) {
    const int NW = G.NW, tid = threadIdx.x, nth = blockDim.x;
    uint64_t *pass = ar0, *rest = ar0 + NW, *d1 = ar0 + 2 * NW, *d2 = ar0 + 3 * NW, *champ = ar0 + 6 * NW, *cross = ar1;
    uint64_t *comp = mine + 2 * NW, *X = mine + 3 * NW, *Y = mine + 4 * NW, *Z = mine + 5 * NW;
    uint64_t* my_list = lists + (size_t)wv * stride;
    const unsigned long long bp_t0 = BP_NOW();
    if (tid == 0) {
        T.n_iso = 0; T.n_dom = 0; T.n_tri = 0; T.regions = 0; T.path = 0; T.lds_path = 0; T.lds_has = 0; T.win = 0ull;
        for (int w = 0; w < BIG_TEAM_MAX_WAVES; w++) T.ncand[w] = 0;
    }
    // ---- the passable set (binary_prob.py:82-86: "empty" = tile 0): the map's cells as one bit string (64 cells a ballot, the chunks
    // dealt out to the wavefronts), a row word cut out of it
    {
        uint64_t* str = d2;                                        // (cells / 64 + 2 words: at most NW + 2 -- d2 and what follows it are free)
        const int cells = G.W * G.H, nch = (cells + 63) >> 6;
        constexpr int U = 16;
        for (int c0 = wv * U; c0 < nch; c0 += nwv * U) {
            uint8_t t[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int c = (c0 + u) * 64 + lane;
                const uint8_t v = m[c < cells ? c : cells - 1];
                t[u] = c < cells ? v : (uint8_t)1;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (c0 + u >= nch) break;          // wave-uniform
                const uint64_t q0 = __ballot((t[u] & 1) == 0);
                if (lane == 0) str[c0 + u] = q0;
            }
        }
        if (tid < 2) str[nch + tid] = 0ull;
        __syncthreads();
        for (int i = tid; i < NW; i += nth) {
            const int r = big_row(G, i), k = i - r * G.KW;
            const int o = r * G.W + 64 * k, wd = o >> 6, sh = o & 63;
            const uint64_t valid = k == G.KW - 1 ? G.last : ~0ull;
            pass[i] = ((str[wd] >> sh) | (sh ? str[wd + 1] << (64 - sh) : 0ull)) & valid;
            cross[i] = 0ull; champ[i] = 0ull;
        }
        for (int i = lane; i < NW; i += 64) comp[i] = 0ull;       // (every wavefront its own; wavefront 0's is scratch of the next passes)
        __syncthreads();
    }
    // ---- components of one, two and three cells in closed form (big_regions_path: the same three passes, the words dealt out)
    {
        int n_iso = 0, n_dom = 0, n_tri = 0;
        for (int i = tid; i < NW; i += nth) {
            const int r = big_row(G, i), k = i - r * G.KW;
            const uint64_t p = pass[i];
            uint64_t lf = p << 1, rt = p >> 1;
            if (k > 0) lf |= pass[i - 1] >> 63;
            if (k < G.KW - 1) rt |= pass[i + 1] << 63;
            const uint64_t a = lf & p, b = rt & p, c = (r > 0 ? pass[i - G.KW] : 0ull) & p, d = (r < G.H - 1 ? pass[i + G.KW] : 0ull) & p;
            const uint64_t s0 = a ^ b, c0 = a & b, s1 = c ^ d, c1 = c & d, n0 = s0 ^ s1, kk = s0 & s1, two = c0 | c1 | kk;
            const uint64_t iso = p & ~(a | b | c | d);
            d1[i] = n0 & ~two;                                     // degree 1
            d2[i] = ~n0 & (c0 ^ c1 ^ kk) & ~(c0 & c1);             // degree 2
            n_iso += __popcll(iso);
            rest[i] = p & ~iso;
        }
        __syncthreads();
        for (int i = tid; i < NW; i += nth) {
            const int r = big_row(G, i), k = i - r * G.KW;
            const uint64_t q = d1[i];
            uint64_t e1 = q << 1, e2 = q >> 1;
            if (k > 0) e1 |= d1[i - 1] >> 63;
            if (k < G.KW - 1) e2 |= d1[i + 1] << 63;
            const uint64_t e3 = r > 0 ? d1[i - G.KW] : 0ull, e4 = r < G.H - 1 ? d1[i + G.KW] : 0ull;
            const uint64_t dom = q & (e1 | e2 | e3 | e4);
            const uint64_t centre = d2[i] & ((e1 & e2) | (e3 & e4) | ((e1 | e2) & (e3 | e4)));
            n_dom += __popcll(dom); n_tri += __popcll(centre);
            d2[i] = dom | centre;                                  // (d2 is read at this thread's own words only)
        }
        __syncthreads();
        for (int i = tid; i < NW; i += nth) {
            const int r = big_row(G, i), k = i - r * G.KW;
            // the ends of the 3-cell components: degree-1 cells next to a centre (next to a cell of a 2-cell component there is only its partner)
            const uint64_t t = d2[i] | (d1[i] & big_neighbours_word(d2, i, r, k, G));
            rest[i] &= ~t;
        }
        n_iso = big_wave_sum(n_iso); n_dom = big_wave_sum(n_dom); n_tri = big_wave_sum(n_tri);
        if (lane == 0) { atomicAdd(&T.n_iso, n_iso); atomicAdd(&T.n_dom, n_dom); atomicAdd(&T.n_tri, n_tri); }
        __syncthreads();
        // wavefront 0's own comp = d1 and X = d2 were the scratch of these passes: comp has to be zero again
        if (wv == 0) { for (int i = lane; i < NW; i += 64) d1[i] = 0ull; }
        __syncthreads();
    }
    const int tiny_regions = T.n_iso + (T.n_dom >> 1) + T.n_tri;
    const int tiny_path = T.n_tri > 0 ? 2 : (T.n_dom > 0 ? 1 : 0);
    if (tid == 0) T.path = tiny_path;
    __syncthreads();
    const unsigned long long bp_t1 = BP_NOW();
    if (wv == 0) { BP_ADD(20, bp_t1 - bp_t0); BP_ADD(25, 1); }
    // ---- phase A: the components inside this wavefront's band of rows
    int my_regions = 0, ncand = 0;
    {
        const int BH = (G.H + nwv - 1) / nwv, lo_r = wv * BH, hi_r = lo_r + BH < G.H ? lo_r + BH : G.H;
        int from = lo_r * G.KW;
        const int end = hi_r * G.KW;
        while (from < end) {
            int b0 = 0;
            const int i0 = big_first(rest, nullptr, from, end, lane, b0);
            if (i0 < 0) break;
            from = i0;
            const int r0 = big_row(G, i0), c0 = 64 * (i0 - r0 * G.KW) + b0;
            BigWindow Wd = big_window_at(G, r0, c0);
            uint64_t cw;
            const bool fits = big_window_component(pass, G, Wd, c0, lane, cw, hi_r);
            // not this band's alone: a cell of the top row with the map open above it (the band's border, or the rest of a piece that
            // left an earlier window), a cell of the band's last row with the map open below it, a cell already on `cross`
            bool other = false;
            if (lane == 0 && r0 > 0) other = (cw & big_window_row(pass, G, Wd, r0 - 1)) != 0ull;
            if (hi_r < G.H && lane == hi_r - 1 - r0) other = other || (cw & big_window_row(pass, G, Wd, hi_r)) != 0ull;
            if (r0 + lane < hi_r) other = other || (cw & big_window_row(cross, G, Wd, r0 + lane)) != 0ull;
            const bool alone = fits && __ballot(other) == 0ull;
            big_window_store<true>(rest, G, Wd, lane, cw);
            if (!alone) big_window_store<false>(cross, G, Wd, lane, cw);
            else {
                DevGroup<64, uint64_t> g;
                ++my_regions;
                const int size = g.popcount_sum(cw);
                if (size - 1 > tiny_path) {
                    if (ncand < cap) {
                        if (lane == 0) my_list[ncand] = ((uint64_t)(uint32_t)size << 32) | ((uint64_t)(uint32_t)i0 << 8) | (uint64_t)(uint32_t)b0;
                        ++ncand;
                    } else {
                        big_team_sweep(pass, G, T, i0, b0, lane);
                    }
                }
            }
            big_sync();
        }
    }
    __syncthreads();
    const unsigned long long bp_t2 = BP_NOW();
    if (wv == 0) BP_ADD(21, bp_t2 - bp_t1);
    // ---- phase B: what crosses the bands, by wavefront 0 (whole components: every piece of one was moved)
    if (wv == 0) {
        int c_lo = 0, c_hi = 0, from = 0;
        for (;;) {
            int b0 = 0;
            const int i0 = big_first(cross, nullptr, from, NW, lane, b0);
            if (i0 < 0) break;
            from = i0;
            int r0 = big_row(G, i0), r1 = r0;
            const int c0 = 64 * (i0 - r0 * G.KW) + b0;
            BigWindow Wd = big_window_at(G, r0, c0);
            uint64_t cw;
            if (big_window_component(pass, G, Wd, c0, lane, cw)) {
                DevGroup<64, uint64_t> g;
                ++my_regions;
                const int size = g.popcount_sum(cw);
                if (size - 1 > __hip_atomic_load(&T.path, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    if (ncand < cap) {
                        if (lane == 0) my_list[ncand] = ((uint64_t)(uint32_t)size << 32) | ((uint64_t)(uint32_t)i0 << 8) | (uint64_t)(uint32_t)b0;
                        ++ncand;
                    } else {
                        big_team_sweep(pass, G, T, i0, b0, lane);
                    }
                }
                big_window_store<true>(cross, G, Wd, lane, cw);
                big_sync();
                continue;
            }
            if (big128_fits(G)) {      // too large for a window, the map small enough for the registers (bigmap.h big128_*)
                Big128 cv;
                int size;
                const int snap = __hip_atomic_load(&T.path, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int e2 = big128_component_sweep(pass, G, r0, c0, lane, true, snap, cv, size);
                ++my_regions;
                if (e2 > snap) {
                    if (lane == 0) { atomicMax(&T.path, e2); T.lds_path = e2; T.lds_has = 1; }
                    big128_store<false>(champ, G, lane, cv);
                    c_lo = 0; c_hi = NW;
                }
                big128_store<true>(cross, G, lane, cv);
                big_sync();
                continue;
            }
            // the word-array path (bigmap.h), swept at once when its size calls for it
            if (lane == 0) comp[i0] = 1ull << b0;
            big_sync();
            big_fill(comp, pass, G, lane, r0, r1);
            const int lo = r0 * G.KW, hi = (r1 + 1) * G.KW;
            ++my_regions;
            const int size = big_popcount(comp, lo, hi, lane);
            const int snap = __hip_atomic_load(&T.path, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (size - 1 > snap) {
                const int e2 = big_double_sweep(comp, G, r0, r1, X, Y, Z, lane, snap);
                if (e2 > snap) {
                    if (lane == 0) { atomicMax(&T.path, e2); T.lds_path = e2; T.lds_has = 1; }
                    for (int i = c_lo + lane; i < c_hi; i += 64) champ[i] = 0ull;
                    big_sync();
                    for (int i = lo + lane; i < hi; i += 64) champ[i] = comp[i];
                    c_lo = lo; c_hi = hi;
                }
            }
            for (int i = lo + lane; i < hi; i += 64) { cross[i] &= ~comp[i]; comp[i] = 0ull; }
            big_sync();
        }
    }
    if (lane == 0) { T.ncand[wv] = ncand; if (my_regions) atomicAdd(&T.regions, my_regions); }
    __syncthreads();
    const unsigned long long bp_t3 = BP_NOW();
    if (wv == 0) BP_ADD(22, bp_t3 - bp_t2);
    // ---- the sweeps that were put off: every wavefront takes the largest component still on any list
    for (;;) {
        uint64_t best = 0ull;
        int at = -1;
        for (int w = 0; w < nwv; w++) {
            const uint64_t* L = lists + (size_t)w * stride;
            const int n = T.ncand[w];
            for (int q = lane; q < n; q += 64) {
                const uint64_t v = __hip_atomic_load(&L[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v > best) { best = v; at = w * cap + q; }
            }
        }
        const int msize = -big_wave_min(-(int)(best >> 32));
        if (msize - 1 <= __hip_atomic_load(&T.path, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        const uint32_t low = (int)(best >> 32) == msize ? (uint32_t)best : 0u;
        const int mlow = -big_wave_min(-(int)low);
        const uint64_t who = __ballot((int)(best >> 32) == msize && (int)(uint32_t)best == mlow);
        const int owner = __ffsll((unsigned long long)who) - 1;
        const int slot = __builtin_amdgcn_readlane(at, owner);
        const uint64_t val = ((uint64_t)(uint32_t)msize << 32) | (uint32_t)mlow;
        int got = 0;
        if (lane == 0) {
            unsigned long long* p = reinterpret_cast<unsigned long long*>(lists + (size_t)(slot / cap) * stride + (slot % cap));
            got = atomicCAS(p, (unsigned long long)val, 0ull) == (unsigned long long)val ? 1 : 0;
        }
        got = __builtin_amdgcn_readfirstlane(got);
        if (!got) continue;                                        // another wavefront took it: look again
        big_team_sweep(pass, G, T, mlow >> 8, mlow & 255, lane);
    }
    __syncthreads();
    const unsigned long long bp_t4 = BP_NOW();
    if (wv == 0) BP_ADD(23, bp_t4 - bp_t3);
    // ---- the champion
    const unsigned long long win = T.win;
    const int we2 = (int)(win >> 32);
    path = T.path;
    regions = tiny_regions + T.regions;
    has = (T.lds_has || we2 > 0) ? 1 : 0;
    if (we2 > T.lds_path && we2 > 0) {                             // a window component beats the word-array path's: `champ` gets it
        for (int i = tid; i < NW; i += nth) champ[i] = 0ull;
        __syncthreads();
        if (wv == 0) {
            const int i0 = (int)((uint32_t)win >> 8), b0 = (int)(win & 255ull), r0 = big_row(G, i0), c0 = 64 * (i0 - r0 * G.KW) + b0;
            BigWindow Wd = big_window_at(G, r0, c0);
            uint64_t cw;
            big_window_component(pass, G, Wd, c0, lane, cw);
            big_window_store<false>(champ, G, Wd, lane, cw);
        }
    }
    __syncthreads();
    if (wv == 0) BP_ADD(24, BP_NOW() - bp_t4);
}
