// Maps beyond the row-bitboard kernels: more than 64 rows or more than 64 columns (up to 255 x 255, the largest map whose cursor
// fits the observation's uint8 `pos`, narrow_rep.py:60-64).  The reference takes any width / height (pcgrl_env.py:106-115,
// probs/problem.py:66-72); the kernels of kernels_stats.h hold a map as one 32/64-bit row mask per lane, which ends at 64 x 64.
// Part of the single translation unit pcgrl_abi.hip.
//
// Here ONE WAVEFRONT owns a map and keeps its rows as multi-word bit masks in LDS (KW = ceil(W / 64) 64-bit words per row, H rows:
// 8 KB per mask for 255 x 255), built from the byte map -- such configurations keep no bit planes (pcgrl_layout.nplanes = 0, like smb),
// Representation.update reads and writes the byte map alone.  The statistics are the same set programs as pcgrl_algos.h, restated
// on word arrays (lane l takes words l, l + 64, ...):
//   big_fill          the 4-connected component of a seed: in-place sweeps (monotone, so any order of the words is right) with
//                     an O(1) run fill inside a word, confined to the rows the component has reached so far
//   big_bfs_levels    helper.py:222-237 run_dikjstra as level-synchronous set expansion inside one component (visited set + two
//                     frontier buffers; a level only touches the rows next to the frontier); eccentricity = number of levels
//   big_regions_path  helper.py:197-207 + :250-264: components in row-major order of their first cell, double sweep from that
//                     cell, np.argmax = first cell of the last frontier; a component of k cells is only swept while k - 1 can
//                     still raise the maximum (and the second sweep only while 2 * e1 can)
//   big_bfs_dist      distance from a set to the nearest cell of another set (zelda_prob.py:98-110)
// No incremental route, no champion cache: every change recomputes.  This is the general path, not the tuned one -- the
// configurations BASELINE.json names all fit the row-bitboard kernels.
#pragma once

// developer build only (tools/probe/big_prof.py: -DPCGRL_BIG_PROF): cycles of a full recomputation by phase, summed into g_tl_buf
#ifdef PCGRL_BIG_PROF
#define BP_NOW() clock64()
#define BP_ADD(i, v) do { if (lane == 0 && g_tl_buf) atomicAdd(&g_tl_buf[i], (unsigned long long)(v)); } while (0)
#else
#define BP_NOW() 0ull
#define BP_ADD(i, v) do {} while (0)
#endif

struct BigGeom {
    int W, H, KW, NW;          // words per row, words per mask
    uint32_t magic;            // floor(i / KW) = (i * magic) >> 16 for i * KW < 65536 (validate_config keeps NW * KW below that)
    uint64_t last;             // valid bits of the last word of a row
};
__device__ __forceinline__ BigGeom big_geom(int W, int H) {
    BigGeom G;
    G.W = W; G.H = H; G.KW = (W + 63) >> 6; G.NW = H * G.KW;
    G.magic = 65536u / (uint32_t)G.KW + 1u;
    const int tail = W - 64 * (G.KW - 1);
    G.last = tail >= 64 ? ~0ull : ((1ull << tail) - 1ull);
    return G;
}
__host__ __device__ inline int big_words(int W, int H) { return H * ((W + 63) >> 6); }
#define BIG_NARRAYS 7          /* masks a wavefront keeps in LDS (k_big) */
__host__ __device__ inline size_t big_wave_lds(int W, int H) {
    // the MT19937 ring of the in-kernel reset + BIG_NARRAYS masks; the bit strings of the plane build (3 x (cells / 64 + 2) words)
    // are laid over masks 3.. (3 * (cells/64 + 2) <= 3 * NW + 6 words)
    return (size_t)PCGRL_MT_N * 4 + ((size_t)BIG_NARRAYS * big_words(W, H) + 8) * 8;
}
__device__ __forceinline__ int big_row(const BigGeom& G, int i) { return (int)(((uint32_t)i * G.magic) >> 16); }

__device__ __forceinline__ int big_wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ bool big_any(bool p) { return __ballot(p) != 0; }
// (the masks live in LDS and are read and written by all lanes of ONE wavefront: its ds instructions execute in order, so a
//  compiler barrier between the phases is all the synchronisation there is)
__device__ __forceinline__ void big_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// s | its 4-neighbours, at word i (row r, word k of the row)
__device__ __forceinline__ uint64_t big_expand_word(const uint64_t* s, int i, int r, int k, const BigGeom& G) {
    const uint64_t c = s[i];
    uint64_t n = c | (c << 1) | (c >> 1);
    if (k > 0) n |= s[i - 1] >> 63;
    if (k < G.KW - 1) n |= s[i + 1] << 63;
    if (r > 0) n |= s[i - G.KW];
    if (r < G.H - 1) n |= s[i + G.KW];
    return n;
}
// every horizontal run of p (inside one word) that holds a bit of s (s a subset of p): the carry chain of an add, both ways
__device__ __forceinline__ uint64_t big_runfill(uint64_t s, uint64_t p) {
    const uint64_t hi = (((p + s) ^ p) & p) | s;
    const uint64_t rp = __brevll(p), rs = __brevll(s);
    const uint64_t lo = (((rp + rs) ^ rp) & rp) | rs;
    return hi | __brevll(lo);
}
__device__ __forceinline__ void big_zero(uint64_t* a, int lo, int hi, int lane) { for (int i = lo + lane; i < hi; i += 64) a[i] = 0ull; }
__device__ __forceinline__ int big_popcount(const uint64_t* a, int lo, int hi, int lane) {
    int n = 0;
    for (int i = lo + lane; i < hi; i += 64) n += __popcll(a[i]);
    return big_wave_sum(n);
}
// First set cell in row-major order of `a` (or of a & ~b when b is given) among the words [lo, hi): word index (-1: none) and bit.
__device__ __forceinline__ int big_first(const uint64_t* a, const uint64_t* b, int lo, int hi, int lane, int& bit) {
    for (int base = lo; base < hi; base += 64) {
        const int i = base + lane;
        uint64_t v = 0ull;
        if (i < hi) { v = a[i]; if (b) v &= ~b[i]; }
        const uint64_t nz = __ballot(v != 0ull);
        if (nz) {
            const int l = __ffsll((unsigned long long)nz) - 1;
            const uint32_t vlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), vhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
            bit = vlo ? __ffs((int)vlo) - 1 : 32 + __ffs((int)vhi) - 1;
            return base + l;
        }
    }
    bit = 0;
    return -1;
}

// The component of `pass` that holds the bits of f (f: zero outside rows [r0, r1], a subset of pass), in place.  On return
// [r0, r1] is the component's row range.
__device__ __forceinline__ void big_fill(uint64_t* f, const uint64_t* pass, const BigGeom& G, int lane, int& r0, int& r1) {
    for (;;) {
        const int a = r0 > 0 ? r0 - 1 : 0, b = r1 < G.H - 1 ? r1 + 1 : G.H - 1;
        bool changed = false, top = false, bot = false;
        for (int i = a * G.KW + lane; i < (b + 1) * G.KW; i += 64) {
            const int r = big_row(G, i), k = i - r * G.KW;
            const uint64_t p = pass[i], old = f[i];
            uint64_t n = big_expand_word(f, i, r, k, G) & p;
            if (n) n = big_runfill(n, p);
            if (n != old) { f[i] = n; changed = true; top = top || r < r0; bot = bot || r > r1; }
        }
        big_sync();
        if (!big_any(changed)) return;
        if (big_any(top)) r0 -= 1;
        if (big_any(bot)) r1 += 1;
    }
}

__device__ __forceinline__ int big_wave_min(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// the four neighbours of the cells of s (not s itself), at word i
__device__ __forceinline__ uint64_t big_neighbours_word(const uint64_t* s, int i, int r, int k, const BigGeom& G) {
    const uint64_t c = s[i];
    uint64_t n = (c << 1) | (c >> 1);
    if (k > 0) n |= s[i - 1] >> 63;
    if (k < G.KW - 1) n |= s[i + 1] << 63;
    if (r > 0) n |= s[i - G.KW];
    if (r < G.H - 1) n |= s[i + G.KW];
    return n;
}

// BFS inside `comp` (rows [r0, r1]) from the single cell (word i0, bit b0): helper.py:222-237 as level-synchronous set expansion.
// V: the visited set, F / Fn: this level's and the next level's frontier (scratch masks; F and Fn are swapped, the caller gets the
// final roles back).  A level only looks at the rows next to the current frontier, [fa - 1, fb + 1] -- a corridor's frontier is a
// cell or two, and a level then costs a handful of words instead of the whole component's rows.  (Frontier bits of earlier levels
// may be left in the two buffers outside that window: harmless, every neighbour of an earlier frontier cell is visited already.)
// Returns the eccentricity; the last frontier is F within the rows [fa, fb].
__device__ __forceinline__ int big_bfs_levels(const uint64_t* comp, int i0, int b0, const BigGeom& G, int r0, int r1, uint64_t* V, uint64_t*& F,
                                              uint64_t*& Fn, int lane, int& fa, int& fb) {
    const int lo = r0 * G.KW, hi = (r1 + 1) * G.KW;
    big_zero(V, lo, hi, lane); big_zero(F, lo, hi, lane); big_zero(Fn, lo, hi, lane);
    big_sync();
    if (lane == 0) { F[i0] = 1ull << b0; V[i0] = 1ull << b0; }
    big_sync();
    fa = fb = big_row(G, i0);
    int ecc = 0;
    for (;;) {
        const int a = fa - 1 > r0 ? fa - 1 : r0, b = fb + 1 < r1 ? fb + 1 : r1;
        int mn = 1 << 20, mx = 1 << 20;                       // (mx holds -row: one kind of reduction)
        for (int i = a * G.KW + lane; i < (b + 1) * G.KW; i += 64) {
            const int r = big_row(G, i), k = i - r * G.KW;
            const uint64_t c = F[i];
            uint64_t n = (c << 1) | (c >> 1);
            if (k > 0) n |= F[i - 1] >> 63;
            if (k < G.KW - 1) n |= F[i + 1] << 63;
            if (r > r0) n |= F[i - G.KW];
            if (r < r1) n |= F[i + G.KW];
            const uint64_t v = V[i], nw = n & comp[i] & ~v;
            Fn[i] = nw;
            if (nw) { V[i] = v | nw; mn = r < mn ? r : mn; mx = -r < mx ? -r : mx; }
        }
        big_sync();
        mn = big_wave_min(mn); mx = big_wave_min(mx);
        if (mn == (1 << 20)) return ecc;                      // nothing new: F holds the last frontier
        ++ecc;
        uint64_t* t = F; F = Fn; Fn = t;
        fa = mn; fb = -mx;
    }
}

// A component through a 64 x 64 window, in registers.  Most components of a large map are a few cells to a few hundred: the fills and
// sweeps of the word arrays cost a loop over LDS words, a wavefront reduction or two and a compiler barrier per round (~9 000 cycles a
// component, ~1 900 a BFS level: tools/probe/big_prof.py), while the row-bitboard algorithms of the maps up to 64 x 64 (pcgrl_algos.h:
// pcg_component's run / column fills, pcg_double_sweep with per-lane level bookkeeping) do the same on one 64-bit row mask per lane.
// So: cut the window [r0, r0 + 64) x [cw0, cw0 + 64) around the seed out of `pass` (lane l = row r0 + l, a funnel shift of two words),
// extract the component there, and if it does not touch a window edge that is not also the map's, it IS the component -- counted,
// measured and swept in registers (the first cell in row-major order and the first cell of the last frontier are the same cells in
// the window's coordinates).  One that reaches an edge is left to the word-array path.
struct BigWindow { int r0, cw0, kx, sh; };
__device__ __forceinline__ BigWindow big_window_at(const BigGeom& G, int r0, int c) {
    BigWindow Wd;
    int cw0 = c - 32;
    const int cmax = G.W > 64 ? G.W - 64 : 0;
    cw0 = cw0 < 0 ? 0 : (cw0 > cmax ? cmax : cw0);
    Wd.r0 = r0; Wd.cw0 = cw0; Wd.kx = cw0 >> 6; Wd.sh = cw0 & 63;
    return Wd;
}
__device__ __forceinline__ uint64_t big_window_load(const uint64_t* a, const BigGeom& G, const BigWindow& Wd, int lane) {
    const int row = Wd.r0 + lane;
    if (row >= G.H) return 0ull;
    const int i = row * G.KW + Wd.kx;
    uint64_t v = a[i] >> Wd.sh;
    if (Wd.sh != 0 && Wd.kx + 1 < G.KW) v |= a[i + 1] << (64 - Wd.sh);
    return v;
}
// a[.] &= ~m (CLEAR) or a[.] = m over the window's rows (the other words of those rows are the caller's business)
template <bool CLEAR>
__device__ __forceinline__ void big_window_store(uint64_t* a, const BigGeom& G, const BigWindow& Wd, int lane, uint64_t m) {
    const int row = Wd.r0 + lane;
    if (row >= G.H) return;
    const int i = row * G.KW + Wd.kx;
    if (m == 0ull) return;            // (rows the component does not reach are not touched: they may be another wavefront's, bigmap_team.h)
    if (CLEAR) {
        a[i] &= ~(m << Wd.sh);
        if (Wd.sh != 0 && Wd.kx + 1 < G.KW) a[i + 1] &= ~(m >> (64 - Wd.sh));
    } else {
        a[i] |= m << Wd.sh;
        if (Wd.sh != 0 && Wd.kx + 1 < G.KW) a[i + 1] |= m >> (64 - Wd.sh);
    }
}
// The component of the seed (row r0 = the component's top row, column c) if it fits the window: true, comp = its rows in window
// coordinates.  false: it reaches a window edge beyond which the map goes on.
// row_hi: the fill is confined to the rows below it (the row bands of the wavefronts that share a map: bigmap_team.h).
// A component that leaves the window on one side only gets a second try with the window moved the other way as far as what was
// filled allows (the seed is the component's top-left cell, not its middle: `Wd` is then the moved window).
__device__ __forceinline__ bool big_window_fill(const uint64_t* pass, const BigGeom& G, const BigWindow& Wd, int c, int lane, uint64_t& comp, int row_hi,
                                                bool& out_l, bool& out_r) {
    DevGroup<64, uint64_t> g;
    const uint64_t pw = Wd.r0 + lane < row_hi ? big_window_load(pass, G, Wd, lane) : 0ull;
    uint64_t f = lane == 0 ? 1ull << (c - Wd.cw0) : 0ull;
    // pcg_component for whole-wavefront groups: plain flood steps first (most components are small) -- and the constants of the run /
    // column fills only for one that is still growing after eight of them
    bool settled = false;
    for (int i = 0; i < 4 && !settled; i++) {
        uint64_t n = pcg_expand(g, f) & pw;
        n = pcg_expand(g, n) & pw;
        settled = !g.wave_any(n ^ f);
        f = n;
    }
    if (!settled) {
        const PcgFillCtx<DevGroup<64, uint64_t>> ctx = pcg_fill_ctx(g, pw);
        f = pcg_component_fills(g, f, ctx);
    }
    comp = f;
    out_l = __ballot(Wd.cw0 > 0 && (comp & 1ull) != 0ull) != 0ull;
    out_r = __ballot(Wd.cw0 + 64 < G.W && (comp >> 63) != 0ull) != 0ull;
    const bool out_b = lane == 63 && Wd.r0 + 64 < (row_hi < G.H ? row_hi : G.H) && comp != 0ull;
    return __ballot(out_b) == 0ull && !out_l && !out_r;
}
__device__ __forceinline__ bool big_window_component(const uint64_t* pass, const BigGeom& G, BigWindow& Wd, int c, int lane, uint64_t& comp, int row_hi = 1 << 30) {
    bool out_l, out_r;
    if (big_window_fill(pass, G, Wd, c, lane, comp, row_hi, out_l, out_r)) return true;
    if (out_l == out_r) return false;                          // both sides (or the bottom alone): too large for a window
    int shift;                                                 // columns to move the window to the left (negative: to the right)
    if (out_l) {
        const int hi = -big_wave_min(comp ? (int)__builtin_clzll(comp) - 63 : 1);      // last column reached
        shift = 62 - hi < Wd.cw0 ? 62 - hi : Wd.cw0;
        if (shift <= 0) return false;
    } else {
        const int lo = big_wave_min(comp ? (int)__builtin_ctzll(comp) : 64);          // first column reached
        const int cmax = G.W > 64 ? G.W - 64 : 0;
        shift = -(lo - 1 < cmax - Wd.cw0 ? lo - 1 : cmax - Wd.cw0);
        if (shift >= 0) return false;
    }
    const int cw0 = Wd.cw0 - shift;
    Wd.cw0 = cw0; Wd.kx = cw0 >> 6; Wd.sh = cw0 & 63;
    return big_window_fill(pass, G, Wd, c, lane, comp, row_hi, out_l, out_r);
}

// A component too large for a window, on maps of at most 128 x 128: the whole map in registers -- lane l holds rows l and 64 + l, two
// words each -- and helper.py:222-237's level-synchronous BFS on that: ~70 instructions a level and no LDS, where the word-array form
// (big_bfs_levels) pays a loop over LDS words, a compiler barrier and two wavefront reductions (~2 400 cycles a level on a component
// that spans the map).  It is the case that matters in use: a generator trained on the binary problem makes ONE large region with a
// long path (binary_prob.py:100-118), not the hundreds of small ones of a random map.  The BFS from a seed over the passable set
// yields the component itself (the visited set), its size, the first eccentricity and the last frontier in one go -- no separate fill.
struct Big128 { uint64_t a0, a1, b0, b1; };       // rows l (a) and 64 + l (b), words 0 and 1
__device__ __forceinline__ bool big128_fits(const BigGeom& G) { return G.H <= 128 && G.KW <= 2; }
__device__ __forceinline__ Big128 big128_load(const uint64_t* m, const BigGeom& G, int lane) {
    Big128 v = {0ull, 0ull, 0ull, 0ull};
    if (lane < G.H) { v.a0 = m[lane * G.KW]; if (G.KW > 1) v.a1 = m[lane * G.KW + 1]; }
    if (lane + 64 < G.H) { v.b0 = m[(lane + 64) * G.KW]; if (G.KW > 1) v.b1 = m[(lane + 64) * G.KW + 1]; }
    return v;
}
template <bool CLEAR>       // m &= ~v, or m = v (every row of the mask)
__device__ __forceinline__ void big128_store(uint64_t* m, const BigGeom& G, int lane, const Big128& v) {
    if (lane < G.H) {
        if (CLEAR) { m[lane * G.KW] &= ~v.a0; if (G.KW > 1) m[lane * G.KW + 1] &= ~v.a1; }
        else { m[lane * G.KW] = v.a0; if (G.KW > 1) m[lane * G.KW + 1] = v.a1; }
    }
    if (lane + 64 < G.H) {
        if (CLEAR) { m[(lane + 64) * G.KW] &= ~v.b0; if (G.KW > 1) m[(lane + 64) * G.KW + 1] &= ~v.b1; }
        else { m[(lane + 64) * G.KW] = v.b0; if (G.KW > 1) m[(lane + 64) * G.KW + 1] = v.b1; }
    }
}
__device__ __forceinline__ uint64_t big128_readlane(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ Big128 big128_bit(int row, int col, int lane) {
    Big128 v = {0ull, 0ull, 0ull, 0ull};
    const uint64_t bit = 1ull << (col & 63);
    if (lane == (row & 63)) {
        if (row < 64) { if (col < 64) v.a0 = bit; else v.a1 = bit; }
        else { if (col < 64) v.b0 = bit; else v.b1 = bit; }
    }
    return v;
}
__device__ __forceinline__ bool big128_any(const Big128& v) { return __ballot((v.a0 | v.a1 | v.b0 | v.b1) != 0ull) != 0ull; }
// the four neighbours of the cells of f (rows beyond the map and columns beyond a row's end are cut off by the caller's mask)
__device__ __forceinline__ Big128 big128_neighbours(const Big128& f, int lane) {
    Big128 n;
    n.a0 = (f.a0 << 1) | (f.a0 >> 1) | (f.a1 << 63);
    n.a1 = (f.a1 << 1) | (f.a1 >> 1) | (f.a0 >> 63);
    n.b0 = (f.b0 << 1) | (f.b0 >> 1) | (f.b1 << 63);
    n.b1 = (f.b1 << 1) | (f.b1 >> 1) | (f.b0 >> 63);
    // the row above a row: lane l - 1 of the same set, and for row 64 (lane 0 of b) row 63 (lane 63 of a); below: the other way round
    const uint64_t a0_63 = big128_readlane(f.a0, 63), a1_63 = big128_readlane(f.a1, 63), b0_0 = big128_readlane(f.b0, 0), b1_0 = big128_readlane(f.b1, 0);
    uint64_t ub0 = dpp_mov0<0x138>(f.b0), ub1 = dpp_mov0<0x138>(f.b1), da0 = dpp_mov0<0x130>(f.a0), da1 = dpp_mov0<0x130>(f.a1);
    if (lane == 0) { ub0 = a0_63; ub1 = a1_63; }
    if (lane == 63) { da0 = b0_0; da1 = b1_0; }
    n.a0 |= dpp_mov0<0x138>(f.a0) | da0;
    n.a1 |= dpp_mov0<0x138>(f.a1) | da1;
    n.b0 |= ub0 | dpp_mov0<0x130>(f.b0);
    n.b1 |= ub1 | dpp_mov0<0x130>(f.b1);
    return n;
}
// BFS inside `in` from the cells of src (a subset of it): returns the number of levels (the eccentricity for a single cell), V = every
// cell reached, last = the last frontier (src itself when nothing is reached)
__device__ __forceinline__ int big128_bfs(const Big128& in, const Big128& src, Big128& V, Big128& last, int lane) {
    V = src;
    Big128 f = src;
    int ecc = 0;
    for (;;) {
        Big128 n = big128_neighbours(f, lane);
        n.a0 &= in.a0 & ~V.a0; n.a1 &= in.a1 & ~V.a1; n.b0 &= in.b0 & ~V.b0; n.b1 &= in.b1 & ~V.b1;
        if (!big128_any(n)) break;
        V.a0 |= n.a0; V.a1 |= n.a1; V.b0 |= n.b0; V.b1 |= n.b1;
        f = n;
        ++ecc;
    }
    last = f;
    return ecc;
}
// first cell of s in row-major order (s not empty)
__device__ __forceinline__ void big128_first(const Big128& s, int& row, int& col) {
    const uint64_t za = __ballot((s.a0 | s.a1) != 0ull);
    uint64_t w0, w1;
    if (za) { const int l = __ffsll((unsigned long long)za) - 1; row = l; w0 = big128_readlane(s.a0, l); w1 = big128_readlane(s.a1, l); }
    else {
        const uint64_t zb = __ballot((s.b0 | s.b1) != 0ull);
        const int l = __ffsll((unsigned long long)zb) - 1;
        row = 64 + l; w0 = big128_readlane(s.b0, l); w1 = big128_readlane(s.b1, l);
    }
    col = w0 ? __ffsll((unsigned long long)w0) - 1 : 64 + __ffsll((unsigned long long)w1) - 1;
}
__device__ __forceinline__ int big128_popcount(const Big128& v) { return big_wave_sum(__popcll(v.a0) + __popcll(v.a1) + __popcll(v.b0) + __popcll(v.b1)); }
// The component of the cell (row, col) of `pass` and -- if its size and its first sweep allow it to beat `path` -- its double sweep
// (helper.py:250-264: BFS from the component's first cell in row-major order, which (row, col) must be; np.argmax = the first cell of the
// last frontier; BFS from there).  Returns the second eccentricity or 0; comp = the component, size = its cells.
__device__ __forceinline__ int big128_component_sweep(const uint64_t* pass, const BigGeom& G, int row, int col, int lane, bool want_path, int path,
                                                      Big128& comp, int& size) {
    const Big128 P = big128_load(pass, G, lane);
    Big128 last;
    const int e1 = big128_bfs(P, big128_bit(row, col, lane), comp, last, lane);
    size = want_path ? big128_popcount(comp) : 0;
    if (!want_path || size - 1 <= path || 2 * e1 <= path) return 0;
    int r2, c2;
    big128_first(last, r2, c2);
    Big128 V2, l2;
    return big128_bfs(comp, big128_bit(r2, c2, lane), V2, l2, lane);
}

// helper.py:197-207 calc_num_regions + :250-264 calc_longest_path over `pass`.  rest, comp, X, Y, Z: scratch masks (comp must be
// all zero on entry and is on return; Y, Z only for want_path).  want_path = false: regions only (zelda and the search problems).
// champ (may be null; NW words): receives the rows of a champion component -- one whose double sweep gave the returned path -- or
// stays all zero when the path comes from the closed-form tiny components (has_champ says which): big_incremental builds on it.
// cand (may be null; cand_cap words of scratch): the sweeps of the components that fit a window are put off until every component
// has been counted and then made in order of size, largest first -- the answer is a maximum over the components, a component of k
// cells cannot beat k - 1, and in row-major order a step's full recomputation swept eighteen components of a random 100 x 100 map
// before the running maximum had grown (tools/probe/big_prof.py); after the largest one or two most of the others need no sweep.
__device__ __forceinline__ void big_regions_path(const uint64_t* pass, uint64_t* rest, uint64_t* comp, uint64_t* X, uint64_t* Y, uint64_t* Z,
                                                 const BigGeom& G, int lane, bool want_path, int& regions, int& path, uint64_t* champ = nullptr,
                                                 int* has_champ = nullptr, uint64_t* cand = nullptr, int cand_cap = 0) {
    regions = 0; path = 0;
    const unsigned long long bp_t0 = BP_NOW();
    int c_lo = 0, c_hi = 0;                                    // words of the current champion in `champ`
    if (champ) big_zero(champ, 0, G.NW, lane);
    if (has_champ) *has_champ = 0;
    // Components of one, two and three cells in closed form, from bit-sliced neighbour counts (pcg_tiny_components in pcgrl_algos.h:
    // most components of a random map): an isolated cell; two cells of degree 1 next to each other; a cell of degree 2 whose two
    // neighbours both have degree 1.  deg1 goes to `comp`, deg2 -- then the cells of the 2- and 3-cell components found from
    // them -- to X.
    int n_iso = 0, n_dom = 0, n_tri = 0;
    for (int i = lane; i < G.NW; i += 64) {
        const int r = big_row(G, i), k = i - r * G.KW;
        const uint64_t p = pass[i];
        uint64_t lf = p << 1, rt = p >> 1;
        if (k > 0) lf |= pass[i - 1] >> 63;
        if (k < G.KW - 1) rt |= pass[i + 1] << 63;
        const uint64_t a = lf & p, b = rt & p, c = (r > 0 ? pass[i - G.KW] : 0ull) & p, d = (r < G.H - 1 ? pass[i + G.KW] : 0ull) & p;
        const uint64_t s0 = a ^ b, c0 = a & b, s1 = c ^ d, c1 = c & d, n0 = s0 ^ s1, kk = s0 & s1, two = c0 | c1 | kk;
        const uint64_t iso = p & ~(a | b | c | d);
        comp[i] = n0 & ~two;                                   // degree 1
        X[i] = ~n0 & (c0 ^ c1 ^ kk) & ~(c0 & c1);              // degree 2
        n_iso += __popcll(iso);
        rest[i] = p & ~iso;
    }
    big_sync();
    for (int i = lane; i < G.NW; i += 64) {
        const int r = big_row(G, i), k = i - r * G.KW;
        const uint64_t d1 = comp[i];
        uint64_t e1 = d1 << 1, e2 = d1 >> 1;
        if (k > 0) e1 |= comp[i - 1] >> 63;
        if (k < G.KW - 1) e2 |= comp[i + 1] << 63;
        const uint64_t e3 = r > 0 ? comp[i - G.KW] : 0ull, e4 = r < G.H - 1 ? comp[i + G.KW] : 0ull;
        const uint64_t dom = d1 & (e1 | e2 | e3 | e4);
        const uint64_t centre = X[i] & ((e1 & e2) | (e3 & e4) | ((e1 | e2) & (e3 | e4)));
        n_dom += __popcll(dom); n_tri += __popcll(centre);
        X[i] = dom | centre;
    }
    big_sync();
    for (int i = lane; i < G.NW; i += 64) {
        const int r = big_row(G, i), k = i - r * G.KW;
        // the ends of the 3-cell components: degree-1 cells next to a centre (next to a cell of a 2-cell component there is only its partner)
        const uint64_t t = X[i] | (comp[i] & big_neighbours_word(X, i, r, k, G));
        rest[i] &= ~t;
        comp[i] = 0ull;
    }
    n_iso = big_wave_sum(n_iso); n_dom = big_wave_sum(n_dom); n_tri = big_wave_sum(n_tri);
    regions = n_iso + (n_dom >> 1) + n_tri;
    path = n_tri > 0 ? 2 : (n_dom > 0 ? 1 : 0);
    big_sync();
    BP_ADD(0, BP_NOW() - bp_t0); BP_ADD(8, 1); BP_ADD(9, regions);
    int from = 0, ncand = 0;
    for (;;) {
        const unsigned long long bp_t1 = BP_NOW();
        int b0 = 0;
        const int i0 = big_first(rest, nullptr, from, G.NW, lane, b0);
        if (i0 < 0) break;
        from = i0;
        int r0 = big_row(G, i0), r1 = r0;
        {   // the component in a 64 x 64 window, in registers (the seed is the first cell of `rest` in row-major order: its top row)
            const int c0 = 64 * (i0 - r0 * G.KW) + b0;
            BigWindow Wd = big_window_at(G, r0, c0);
            uint64_t cw;
            if (big_window_component(pass, G, Wd, c0, lane, cw)) {
                DevGroup<64, uint64_t> g;
                ++regions;
                BP_ADD(1, BP_NOW() - bp_t1); BP_ADD(5, 1);
                const unsigned long long bp_t2 = BP_NOW();
                if (want_path) {
                    const int size = g.popcount_sum(cw);
                    if (size - 1 > path && ncand < cand_cap) {
                        if (lane == 0) cand[ncand] = ((uint64_t)(uint32_t)size << 32) | ((uint64_t)(uint32_t)i0 << 8) | (uint64_t)(uint32_t)b0;
                        ++ncand;
                    } else if (size - 1 > path) {
                        BP_ADD(6, 1);
                        const int e2 = pcg_double_sweep(g, cw, path);           // 0: the first sweep says it cannot beat `path`
                        if (e2 > path) {
                            path = e2;
                            if (champ) {
                                for (int i = c_lo + lane; i < c_hi; i += 64) champ[i] = 0ull;
                                big_sync();
                                big_window_store<false>(champ, G, Wd, lane, cw);
                                c_lo = r0 * G.KW; c_hi = (r0 + 64 < G.H ? r0 + 64 : G.H) * G.KW;
                                if (has_champ) *has_champ = 1;
                            }
                        }
                    }
                }
                BP_ADD(2, BP_NOW() - bp_t2);
                big_window_store<true>(rest, G, Wd, lane, cw);
                big_sync();
                continue;
            }
        }
        if (big128_fits(G)) {          // too large for a window, the map small enough for the registers: the whole component at once
            Big128 cv;
            int size;
            const int e2 = big128_component_sweep(pass, G, r0, 64 * (i0 - r0 * G.KW) + b0, lane, want_path, path, cv, size);
            ++regions;
            BP_ADD(1, BP_NOW() - bp_t1); BP_ADD(5, 1);
            if (e2 > path) {
                path = e2;
                if (champ) {
                    big128_store<false>(champ, G, lane, cv);
                    c_lo = 0; c_hi = G.NW;
                    if (has_champ) *has_champ = 1;
                }
            }
            big128_store<true>(rest, G, lane, cv);
            big_sync();
            continue;
        }
        if (lane == 0) comp[i0] = 1ull << b0;
        big_sync();
        big_fill(comp, pass, G, lane, r0, r1);
        const int lo = r0 * G.KW, hi = (r1 + 1) * G.KW;
        ++regions;
        BP_ADD(1, BP_NOW() - bp_t1); BP_ADD(5, 1);
        const unsigned long long bp_t2 = BP_NOW();
        if (want_path) {
            const int size = big_popcount(comp, lo, hi, lane);
            if (size - 1 > path) {
                BP_ADD(6, 1);
                // the first cell of the component in row-major order is the seed itself (rest only ever loses whole components)
                int fa, fb;
                const int e1 = big_bfs_levels(comp, i0, b0, G, r0, r1, X, Y, Z, lane, fa, fb);
                if (2 * e1 > path) {
                    int b1 = 0;
                    const int i1 = big_first(Y, nullptr, fa * G.KW, (fb + 1) * G.KW, lane, b1);       // np.argmax: first cell of the last frontier
                    const int e2 = big_bfs_levels(comp, i1, b1, G, r0, r1, X, Y, Z, lane, fa, fb);
                    BP_ADD(10, e1 + e2);
                    if (e2 > path) {
                        path = e2;
                        if (champ) {
                            for (int i = c_lo + lane; i < c_hi; i += 64) champ[i] = 0ull;
                            big_sync();
                            for (int i = lo + lane; i < hi; i += 64) champ[i] = comp[i];
                            c_lo = lo; c_hi = hi;
                            if (has_champ) *has_champ = 1;
                        }
                    }
                }
            }
        }
        BP_ADD(2, BP_NOW() - bp_t2);
        for (int i = lo + lane; i < hi; i += 64) { rest[i] &= ~comp[i]; comp[i] = 0ull; }
        big_sync();
    }
    // the sweeps that were put off, largest component first, until the largest one left cannot beat the maximum
    const unsigned long long bp_t3 = BP_NOW();
    while (ncand > 0) {
        big_sync();
        uint64_t best = 0ull;
        int at = -1;
        for (int q = lane; q < ncand; q += 64) { const uint64_t v = cand[q]; if (v > best) { best = v; at = q; } }
        // (largest size; among equals the entry with the larger word index -- any order is right, this one is fixed)
        const int msize = -big_wave_min(-(int)(best >> 32));
        if (msize - 1 <= path) break;
        const uint32_t low = (int)(best >> 32) == msize ? (uint32_t)best : 0u;
        const int mlow = -big_wave_min(-(int)low);                      // (word index << 8 | bit: below 2^31)
        const uint64_t mine = __ballot((int)(best >> 32) == msize && (int)(uint32_t)best == mlow);
        const int owner = __ffsll((unsigned long long)mine) - 1;
        const int slot = __builtin_amdgcn_readlane(at, owner);
        if (lane == 0) cand[slot] = 0ull;
        const int i0 = mlow >> 8, b0 = mlow & 255, r0 = big_row(G, i0), c0 = 64 * (i0 - r0 * G.KW) + b0;
        BigWindow Wd = big_window_at(G, r0, c0);
        uint64_t cw;
        big_window_component(pass, G, Wd, c0, lane, cw);                // (it fitted when it was counted)
        DevGroup<64, uint64_t> g;
        BP_ADD(6, 1);
        const int e2 = pcg_double_sweep(g, cw, path);
        if (e2 > path) {
            path = e2;
            if (champ) {
                for (int i = c_lo + lane; i < c_hi; i += 64) champ[i] = 0ull;
                big_sync();
                big_window_store<false>(champ, G, Wd, lane, cw);
                c_lo = r0 * G.KW; c_hi = (r0 + 64 < G.H ? r0 + 64 : G.H) * G.KW;
                if (has_champ) *has_champ = 1;
            }
        }
    }
    BP_ADD(7, BP_NOW() - bp_t3);
    BP_ADD(3, BP_NOW() - bp_t0);
}

// helper.py:250-264 for ONE component `comp` (rows [r0, r1]): BFS from its first cell in row-major order, np.argmax = the first cell
// of the last frontier, BFS from there.  Returns the second eccentricity, or 0 when the first one already says that the result
// cannot exceed `path` (e2 <= 2 e1).  X, Y, Z: scratch masks.
__device__ __forceinline__ int big_double_sweep(const uint64_t* comp, const BigGeom& G, int r0, int r1, uint64_t* X, uint64_t* Y, uint64_t* Z, int lane, int path) {
    int b0 = 0, fa, fb;
    const int i0 = big_first(comp, nullptr, r0 * G.KW, (r1 + 1) * G.KW, lane, b0);
    const int e1 = big_bfs_levels(comp, i0, b0, G, r0, r1, X, Y, Z, lane, fa, fb);
    if (2 * e1 <= path) return 0;
    int b1 = 0;
    const int i1 = big_first(Y, nullptr, fa * G.KW, (fb + 1) * G.KW, lane, b1);
    return big_bfs_levels(comp, i1, b1, G, r0, r1, X, Y, Z, lane, fa, fb);
}
__device__ __forceinline__ bool big_bit_at(const uint64_t* a, const BigGeom& G, int x, int y) { return (a[y * G.KW + (x >> 6)] >> (x & 63)) & 1ull; }

// regions and longest path after ONE cell (cx, cy) changed, from the previous answer (binary_incremental in pcgrl_algos.h, restated on
// word arrays).  Preconditions (k_update routes everything else to the full computation): the previous map had a champion -- a
// component whose double sweep gave path_old, rows in `champ` -- and the cell is neither in it nor 4-adjacent to it.  The champion is
// then still a component, nothing that does not touch the cell can beat it, and only the components around the cell are looked at:
// the cell became passable (`added`): the k components around it and the cell are one now -- regions + 1 - k, swept if its size
// allows; it became impassable: its component fell into k pieces -- regions + k - 1, each swept if its size allows.
// pass: the NEW passable set (temporarily modified, restored).  comp, U: scratch masks, all zero on entry and on return.  champ is
// replaced when a sweep beats the old path.  Returns whether champ changed.
__device__ __forceinline__ bool big_incremental(uint64_t* pass, uint64_t* comp, uint64_t* U, uint64_t* X, uint64_t* Y, uint64_t* Z, uint64_t* champ,
                                                const BigGeom& G, int lane, int cx, int cy, bool added, int regions_old, int path_old, int& regions, int& path) {
    const int ci = cy * G.KW + (cx >> 6);
    const uint64_t cbit = 1ull << (cx & 63);
    if (added && lane == 0) pass[ci] &= ~cbit;                  // base = the new map with the cell impassable
    big_sync();
    path = path_old;
    int k = 0, u0 = cy, u1 = cy;                               // row range of the union
    bool new_champ = false;
    int ch_r0 = 0, ch_r1 = -1;                                 // rows of the piece that became the champion (added == false)
    const int nx[4] = {cx - 1, cx + 1, cx, cx}, ny[4] = {cy, cy, cy - 1, cy + 1};
    for (int d = 0; d < 4; d++) {
        const int x = nx[d], y = ny[d];
        if (x < 0 || x >= G.W || y < 0 || y >= G.H) continue;
        if (!big_bit_at(pass, G, x, y) || big_bit_at(U, G, x, y)) continue;      // (wave-uniform: LDS words every lane reads alike)
        int r0 = y, r1 = y;
        if (lane == 0) comp[y * G.KW + (x >> 6)] = 1ull << (x & 63);
        big_sync();
        big_fill(comp, pass, G, lane, r0, r1);
        ++k;
        const int lo = r0 * G.KW, hi = (r1 + 1) * G.KW;
        if (!added) {
            const int size = big_popcount(comp, lo, hi, lane);
            if (size - 1 > path) {
                const int e = big_double_sweep(comp, G, r0, r1, X, Y, Z, lane, path);
                if (e > path) {
                    path = e;
                    big_zero(champ, 0, G.NW, lane);
                    big_sync();
                    for (int i = lo + lane; i < hi; i += 64) champ[i] = comp[i];
                    new_champ = true; ch_r0 = r0; ch_r1 = r1;
                }
            }
        }
        for (int i = lo + lane; i < hi; i += 64) { U[i] |= comp[i]; comp[i] = 0ull; }
        u0 = r0 < u0 ? r0 : u0; u1 = r1 > u1 ? r1 : u1;
        big_sync();
    }
    (void)ch_r0; (void)ch_r1;
    if (added) {
        regions = regions_old + 1 - k;
        if (lane == 0) { pass[ci] |= cbit; U[ci] |= cbit; }
        big_sync();
        const int lo = u0 * G.KW, hi = (u1 + 1) * G.KW;
        const int size = big_popcount(U, lo, hi, lane);
        if (size - 1 > path) {
            const int e = big_double_sweep(U, G, u0, u1, X, Y, Z, lane, path);
            if (e > path) {
                path = e;
                big_zero(champ, 0, G.NW, lane);
                big_sync();
                for (int i = lo + lane; i < hi; i += 64) champ[i] = U[i];
                new_champ = true;
            }
        }
    } else {
        regions = regions_old + k - 1;
    }
    for (int i = u0 * G.KW + lane; i < (u1 + 1) * G.KW; i += 64) U[i] = 0ull;
    big_sync();
    return new_champ;
}

// Distance from the cells of C (the source set; overwritten) to the nearest cell of dst (dst not containing the source) through pass;
// -1: none reachable (run_dikjstra leaves -1 there).  N: scratch mask.
__device__ __forceinline__ int big_bfs_dist(uint64_t* C, const uint64_t* dst, const uint64_t* pass, uint64_t* N, const BigGeom& G, int lane) {
    int t = 0;
    for (;;) {
        bool hit = false, grew = false;
        for (int i = lane; i < G.NW; i += 64) {
            const int r = big_row(G, i), k = i - r * G.KW;
            const uint64_t n = big_expand_word(C, i, r, k, G) & pass[i], fresh = n & ~C[i];
            N[i] = n;
            hit = hit || (fresh & dst[i]) != 0ull;
            grew = grew || fresh != 0ull;
        }
        big_sync();
        ++t;
        if (big_any(hit)) return t;
        if (!big_any(grew)) return -1;
        uint64_t* x = C; C = N; N = x;
    }
}

// The bit planes of the tile ids of map m (global, u8 [H][W]) as row masks pl[b * NW + i] (NPL planes).  The map is one string
// of cells: 64 consecutive cells give 64 bits of each plane's string with one ballot (eight loads in flight per round), and a
// row word is cut out of the string.  `str`: scratch for the NPL strings of cells / 64 + 2 words each.
template <int NPL>
__device__ __forceinline__ void big_planes(const uint8_t* __restrict__ m, const BigGeom& G, uint64_t* pl, uint64_t* str, int lane) {
    const int cells = G.W * G.H, nch = (cells + 63) >> 6, sw = nch + 2;
    // (thirty-two loads in flight per round: a round costs one trip to memory, and the map of a reset or of another wavefront's
    //  change is not in any cache -- with eight, a 100 x 100 map took twenty trips, 40 us of a full recomputation's 500)
    constexpr int U = 32;
    for (int c0 = 0; c0 < nch; c0 += U) {
        uint8_t t[U];
#pragma unroll
        for (int u = 0; u < U; u++) {      // (an unconditional load of a clamped index: a guarded one is a branch and a wait per load)
            const int c = (c0 + u) * 64 + lane;
            const uint8_t v = m[c < cells ? c : cells - 1];
            t[u] = c < cells ? v : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (c0 + u >= nch) break;          // wave-uniform
            const uint64_t q0 = __ballot(t[u] & 1), q1 = NPL > 1 ? __ballot(t[u] & 2) : 0ull, q2 = NPL > 1 ? __ballot(t[u] & 4) : 0ull;
            if (lane == 0) { str[c0 + u] = q0; if (NPL > 1) { str[sw + c0 + u] = q1; str[2 * sw + c0 + u] = q2; } }
        }
    }
    if (lane < 2) { str[nch + lane] = 0ull; if (NPL > 1) { str[sw + nch + lane] = 0ull; str[2 * sw + nch + lane] = 0ull; } }
    big_sync();
    for (int i = lane; i < G.NW; i += 64) {
        const int r = big_row(G, i), k = i - r * G.KW;
        const int o = r * G.W + 64 * k, wd = o >> 6, sh = o & 63;
        const uint64_t valid = k == G.KW - 1 ? G.last : ~0ull;
#pragma unroll
        for (int b = 0; b < NPL; b++) {
            const uint64_t lo = str[b * sw + wd] >> sh, hi = sh ? str[b * sw + wd + 1] << (64 - sh) : 0ull;
            pl[b * G.NW + i] = (lo | hi) & valid;
        }
    }
    big_sync();
}

// helper.py:37-43, 56-62 get_floor_dist(map, ["player"], ["solid"]) on word arrays: all player bits fall together, a row per
// round (floor_dist in pcgrl_algos.h).  A, Bm: scratch masks.
__device__ __forceinline__ int big_floor_dist(const uint64_t* player, const uint64_t* solid, uint64_t* A, uint64_t* Bm, const BigGeom& G, int lane) {
    for (int i = lane; i < G.NW; i += 64) A[i] = player[i];
    big_sync();
    int n_act = big_popcount(A, 0, G.NW, lane), result = 0, lost = 0;
    for (int dy = 1; dy < G.H && n_act > 0; dy++) {
        int moved = 0, hit = 0;
        for (int i = lane; i < G.NW; i += 64) {
            const uint64_t mv = i >= G.KW ? A[i - G.KW] : 0ull;          // row r receives row r - 1
            const uint64_t h = mv & solid[i];
            moved += __popcll(mv); hit += __popcll(h);
            Bm[i] = mv & ~solid[i];
        }
        big_sync();
        const int n_moved = big_wave_sum(moved), n_hit = big_wave_sum(hit);
        lost += n_act - n_moved;
        result += n_hit * (dy - 1);
        n_act = n_moved - n_hit;
        uint64_t* x = A; A = Bm; Bm = x;
    }
    return result + (lost + n_act) * (G.H - 1);
}

// Problem.get_stats of the map `m` by one wavefront (the search problems: without the planner).  `ar`: BIG_NARRAYS masks.
// Returns true when a search kernel has to finish the job.  The tile classes are those of pcgrl_algos.h (zelda_masks,
// sokoban_stats, mdungeon_stats, ddave_stats).
template <int PROB>
__device__ __forceinline__ bool big_item_stats(const PcgrlParams& P, const DevBufs& B, const uint8_t* __restrict__ m, const BigGeom& G, uint64_t* ar, int lane,
                                               int32_t* s) {
    const int NW = G.NW;
    uint64_t *a0 = ar, *a1 = ar + NW, *a2 = ar + 2 * NW, *a3 = ar + 3 * NW, *a4 = ar + 4 * NW, *a5 = ar + 5 * NW, *a6 = ar + 6 * NW;
    for (int k = 0; k < PCGRL_MAX_STATS; k++) s[k] = 0;
    if (PROB == PCGRL_PROB_BINARY) {
        const unsigned long long bp_tp = BP_NOW();
        big_planes<1>(m, G, a0, a3, lane);                                  // a0 = solid
        BP_ADD(4, BP_NOW() - bp_tp);
        for (int i = lane; i < NW; i += 64) {
            const int r = big_row(G, i), k = i - r * G.KW;
            a0[i] = ~a0[i] & (k == G.KW - 1 ? G.last : ~0ull);             // a0 = empty (binary_prob.py:82-86: passable = ["empty"])
            a2[i] = 0ull;
        }
        big_sync();
        int regions, path, has = 0;
        // a6: the champion component (big_incremental); the list of put-off sweeps in the wavefront's MT19937 area, which only a
        // reset uses (k_big: it sits right below the masks)
        big_regions_path(a0, a1, a2, a3, a4, a5, G, lane, true, regions, path, a6, &has, ar - PCGRL_MT_N / 2, PCGRL_MT_N / 2);
        s[0] = regions; s[1] = path; s[2] = has;                             // s[2]: "there is a champion"
        return false;
    }
    big_planes<3>(m, G, a0, a3, lane);                                      // a0, a1, a2 = bits 0, 1, 2 of the tile id
    int cnt[5] = {0, 0, 0, 0, 0};
    // one pass: the class counts, and the passable set of the region count in a3
    for (int i = lane; i < NW; i += 64) {
        const int r = big_row(G, i), k = i - r * G.KW;
        const uint64_t v = k == G.KW - 1 ? G.last : ~0ull, b0 = a0[i], b1 = a1[i], b2 = a2[i];
        const uint64_t c1 = ~b2 & ~b1 & b0 & v, c2 = ~b2 & b1 & ~b0 & v, c3 = ~b2 & b1 & b0 & v, c4 = b2 & ~b1 & ~b0 & v, c5 = b2 & ~b1 & b0 & v,
                       c6 = b2 & b1 & ~b0 & v, c7 = b2 & b1 & b0 & v;
        uint64_t walk;
        if (PROB == PCGRL_PROB_ZELDA) {            // zelda_prob.py:45-46: 1 solid 2 player 3 key 4 door 5 bat 6 scorpion 7 spider
            cnt[0] += __popcll(c2); cnt[1] += __popcll(c3); cnt[2] += __popcll(c4); cnt[3] += __popcll(c5 | c6 | c7);
            walk = v & ~c1 & ~c4;
        } else if (PROB == PCGRL_PROB_SOKOBAN) {   // sokoban_prob.py:44-45: 1 solid 2 player 3 crate 4 target
            cnt[0] += __popcll(c2); cnt[1] += __popcll(c3); cnt[2] += __popcll(c4);
            walk = v & ~c1;
        } else if (PROB == PCGRL_PROB_MDUNGEON) {  // mdungeon_prob.py:49-50: 1 solid 2 player 3 exit 4 potion 5 treasure 6 goblin 7 ogre
            cnt[0] += __popcll(c2); cnt[1] += __popcll(c3); cnt[2] += __popcll(c4); cnt[3] += __popcll(c5); cnt[4] += __popcll(c6 | c7);
            walk = v & ~c1;
        } else {                                   // ddave_prob.py:48-49: 1 solid 2 player 3 exit 4 diamond 5 key 6 spike
            cnt[0] += __popcll(c2); cnt[1] += __popcll(c3); cnt[2] += __popcll(c5); cnt[3] += __popcll(c4); cnt[4] += __popcll(c6);
            walk = v & ~c1 & ~c6;
        }
        a3[i] = walk;
        a5[i] = 0ull;
    }
    big_sync();
#pragma unroll
    for (int q = 0; q < 5; q++) cnt[q] = big_wave_sum(cnt[q]);
    int regions, unused;
    big_regions_path(a3, a4, a5, a6, nullptr, nullptr, G, lane, false, regions, unused);
    if (PROB == PCGRL_PROB_ZELDA) {                // zelda_prob.py:80-112
        const int player = cnt[0], key = cnt[1], door = cnt[2], enemies = cnt[3];
        int nearest = 0, path = 0;
        if (player == 1 && regions == 1) {
            // a3 = passable set of the sweep, a4 = source (then the growing set), a5 = target, a6 = scratch
            if (enemies > 0) {                     // through empty | player | enemies: the key is NOT passable here (:98)
                for (int i = lane; i < NW; i += 64) {
                    const int r = big_row(G, i), k = i - r * G.KW;
                    const uint64_t v = k == G.KW - 1 ? G.last : ~0ull, b0 = a0[i], b1 = a1[i], b2 = a2[i];
                    a4[i] = ~b2 & b1 & ~b0 & v;                                          // player
                    a5[i] = b2 & (b1 | b0) & v;                                          // enemies
                    a3[i] = v & ~(~b2 & b0) & ~(b2 & ~b1 & ~b0);                         // not solid (1), not key (3), not door (4)
                }
                big_sync();
                const int d = big_bfs_dist(a4, a5, a3, a6, G, lane);
                nearest = d > 0 ? d : P.prob_width * P.prob_height;
            }
            if (key == 1 && door == 1) {           // player -> key through everything but solid and door, key -> door with the door (:104-110)
                for (int i = lane; i < NW; i += 64) {
                    const int r = big_row(G, i), k = i - r * G.KW;
                    const uint64_t v = k == G.KW - 1 ? G.last : ~0ull, b0 = a0[i], b1 = a1[i], b2 = a2[i];
                    a4[i] = ~b2 & b1 & ~b0 & v;                                          // player
                    a5[i] = ~b2 & b1 & b0 & v;                                           // key
                    a3[i] = v & ~(~b2 & ~b1 & b0) & ~(b2 & ~b1 & ~b0);                   // not solid, not door
                }
                big_sync();
                path = big_bfs_dist(a4, a5, a3, a6, G, lane);
                for (int i = lane; i < NW; i += 64) {
                    const int r = big_row(G, i), k = i - r * G.KW;
                    const uint64_t v = k == G.KW - 1 ? G.last : ~0ull, b0 = a0[i], b1 = a1[i], b2 = a2[i];
                    a4[i] = ~b2 & b1 & b0 & v;                                           // key
                    a5[i] = b2 & ~b1 & ~b0 & v;                                          // door
                    a3[i] = v & ~(~b2 & ~b1 & b0);                                       // not solid
                }
                big_sync();
                path += big_bfs_dist(a4, a5, a3, a6, G, lane);                            // -1 when the door is walled off
            }
        }
        s[0] = player; s[1] = key; s[2] = door; s[3] = enemies; s[4] = regions; s[5] = nearest; s[6] = path;
        return false;
    }
    if (PROB == PCGRL_PROB_SOKOBAN) {              // sokoban_prob.py:133-145 without the solver
        s[0] = cnt[0]; s[1] = cnt[1]; s[2] = cnt[2]; s[3] = regions;
        s[4] = P.prob_width * P.prob_height * (P.prob_width + P.prob_height); s[5] = 0;
        return cnt[0] == 1 && cnt[1] == cnt[2] && cnt[1] > 0 && regions == 1;
    }
    if (PROB == PCGRL_PROB_MDUNGEON) {             // mdungeon_prob.py:139-157 without the planner (row layout: md_pack)
        s[0] = cnt[0]; s[1] = cnt[1]; s[2] = cnt[2]; s[3] = cnt[3]; s[4] = cnt[4]; s[5] = regions;
        s[6] = P.prob_width * P.prob_height; s[7] = 0;
        return cnt[0] == 1 && cnt[1] == 1 && regions == 1;
    }
    // ddave_prob.py:141-161 without the planner (row layout: dd_pack).  The three counts that share slot 0 have eight bits each:
    // a map with more than 255 player, exit or key tiles is reported through the status word instead of being packed wrongly.
    if (cnt[0] > 255 || cnt[1] > 255 || cnt[2] > 255) { if (lane == 0) atomicOr(B.status, PCGRL_STATUS_TOO_MANY_CRATES); }
    for (int i = lane; i < NW; i += 64) {
        const int r = big_row(G, i), k = i - r * G.KW;
        const uint64_t v = k == G.KW - 1 ? G.last : ~0ull, b0 = a0[i], b1 = a1[i], b2 = a2[i];
        a4[i] = ~b2 & b1 & ~b0 & v;                                                      // player
        a5[i] = ~b2 & ~b1 & b0 & v;                                                      // solid
    }
    big_sync();
    s[0] = (cnt[0] & 255) | ((cnt[1] & 255) << 8) | ((cnt[2] & 255) << 16);
    s[1] = big_floor_dist(a4, a5, a3, a6, G, lane);
    s[2] = cnt[3]; s[3] = cnt[4]; s[4] = regions; s[5] = 0;
    s[6] = P.prob_width * P.prob_height; s[7] = 0;
    return cnt[0] == 1 && cnt[1] == 1 && cnt[2] == 1 && regions == 1;
}
