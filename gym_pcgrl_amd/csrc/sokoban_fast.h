// Register-resident Sokoban search: the same agents as sok_search (sokoban_solver.h), for levels with at
// most SOKF_MAXC crates -- every level the benchmark ever sees; the rest take sok_search.
//
// sok_search keeps the node under expansion and the level in LDS because their arrays are indexed
// dynamically; on one lane that is a chain of ~100 dependent LDS reads per pop (measured on MI355X:
// 3 900 cycles per BFS pop, 8 000 per A* pop).  Here a state is two registers -- the ordered crate list as
// the bytes of a 64-bit word and the player cell -- plus a crate bitboard, and the level is a handful of
// 64-bit masks, so a pop is register arithmetic around three memory structures:
//   * pool   16-byte nodes in global memory (one dwordx4 load/store).  A pop never waits for it: the next
//            node to be popped is either the heap top left by the repair (fetched ahead, in flight while
//            the children are made) or one of the up to four children just pushed (kept in `cache`),
//   * table  visited set in LDS, open addressing on the *exact* 64-bit key (player | crates << 8): a hit
//            needs no look at the pool,
//   * heap   CPython heapq on packed (priority << 16 | node) words in LDS; the sift loops walk two
//            levels per round with no bounds tests while the grandchildren exist.
// Order of exploration, visited-on-pop, iteration counting and best-node rules are those of sok_search.
#pragma once
#include "sokoban_solver.h"

#define SOKF_MAXC 7
#define SOK_POLL_MASK 31           /* an A* agent looks at its stop word every 32 pops (the hooks in kernels_sokoban.h / _mdungeon.h / _ddave.h) */

struct alignas(16) SokFastNode { uint64_t cr; uint32_t ph; uint32_t depth; };   // ph = player | h << 16

template <int NW>
struct SokFastLevel {
    uint64_t solid[NW], dead[NW], tmask[NW];
    uint64_t tx, ty;        // x / y of target i in byte i
    uint32_t inv_w;         // ceil(2^16 / w): p / w == (p * inv_w) >> 16 for p < 256
    int w, h, nc;
};

template <int NW>
PCGRL_D bool sokf_bit(const uint64_t* m, int p) {
    if (NW == 1) return (m[0] >> p) & 1ull;
    const uint64_t lo = (p & 64) ? m[1] : m[0], hi = (p & 64) ? m[3] : m[2];
    return (((p & 128) ? hi : lo) >> (p & 63)) & 1ull;
}
template <int NW>
PCGRL_D void sokf_flip(uint64_t* m, int p) {
    if (NW == 1) { m[0] ^= 1ull << p; return; }
    const uint64_t b = 1ull << (p & 63);
    const int wd = p >> 6;
    m[0] ^= wd == 0 ? b : 0; m[1] ^= wd == 1 ? b : 0; m[2] ^= wd == 2 ? b : 0; m[3] ^= wd == 3 ? b : 0;
}
template <int NW>
PCGRL_D bool sokf_any_and(const uint64_t* a, const uint64_t* b) {
    uint64_t r = a[0] & b[0];
    for (int i = 1; i < NW; i++) r |= a[i] & b[i];
    return r != 0;
}
template <int NW>
PCGRL_D bool sokf_covers(const uint64_t* a, const uint64_t* t) {    // every bit of t set in a
    uint64_t r = ~a[0] & t[0];
    for (int i = 1; i < NW; i++) r |= ~a[i] & t[i];
    return r == 0;
}

template <int NW>
PCGRL_D void sokf_level(const SokLevel& L, SokFastLevel<NW>& F) {
    for (int i = 0; i < NW; i++) { F.solid[i] = L.solid[i]; F.dead[i] = L.dead[i]; F.tmask[i] = L.targetmask[i]; }
    F.w = L.w; F.h = L.h; F.nc = L.nc;
    F.inv_w = (65536u + (uint32_t)L.w - 1u) / (uint32_t)L.w;
    F.tx = 0; F.ty = 0;
    for (int i = 0; i < L.nc; i++) {
        F.tx |= (uint64_t)L.cx[L.target[i]] << (8 * i);
        F.ty |= (uint64_t)L.cy[L.target[i]] << (8 * i);
    }
}
PCGRL_D int sokf_absdiff(int a, int b) { return a > b ? a - b : b - a; }
// engine.py:282-296 (same greedy matching as sok_heuristic) on the packed crate list
template <int NW>
PCGRL_D int sokf_heuristic(const SokFastLevel<NW>& F, uint64_t cr) {
    uint32_t used = 0;
    int distance = 0;
    for (int c = 0; c < F.nc; c++) {
        const int p = (int)((cr >> (8 * c)) & 0xFF);
        const int cy = (int)(((uint32_t)p * F.inv_w) >> 16), cx = p - cy * F.w;
        int best = F.w + F.h, match = -1, firstfree = -1, matchd = 0, firstd = 0;
        for (int i = 0; i < F.nc; i++) {
            if ((used >> i) & 1u) continue;
            const int d = sokf_absdiff(cx, (int)((F.tx >> (8 * i)) & 0xFF)) + sokf_absdiff(cy, (int)((F.ty >> (8 * i)) & 0xFF));
            if (firstfree < 0) { firstfree = i; firstd = d; }
            if (best > d) { match = i; best = d; matchd = d; }
        }
        if (match < 0) { match = firstfree; matchd = firstd; }
        distance += matchd;
        used |= 1u << match;
    }
    return distance;
}
// index of the crate standing on cell p (p != 0; unused bytes of cr are 0), known to exist
PCGRL_D int sokf_crate_index(uint64_t cr, int p) {
    const uint64_t x = cr ^ (0x0101010101010101ull * (uint64_t)p);
    const uint64_t t = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;   // lowest marked byte is exact
#if defined(__HIPCC__)
    return (__ffsll((unsigned long long)t) - 1) >> 3;
#else
    return __builtin_ctzll(t) >> 3;
#endif
}

// One child of Node.getChildren (State.update engine.py:298-327) for direction d = 0..3 (L, R, U, D): dropped if the
// player did not move, if the pushed crate is blocked, or if after the push any crate stands on a deadlock cell.
struct SokChild { uint64_t cr; int np, h, ok; };
PCGRL_D int sokf_dir(int d, int w) { return (d & 2) ? ((d & 1) ? w : -w) : ((d & 1) ? 1 : -1); }      // L, R, U, D
template <int NW>
PCGRL_D SokChild sokf_child_dir(const SokFastLevel<NW>& F, uint64_t cr, const uint64_t* cb, int player, int h, int dir) {
    SokChild c;
    const int np = player + dir;
    c.cr = cr; c.np = np; c.h = h; c.ok = 0;
    if (sokf_bit<NW>(F.solid, np)) return c;                     // player did not move
    if (sokf_bit<NW>(cb, np)) {
        const int cp = np + dir;
        if (sokf_bit<NW>(F.solid, cp) || sokf_bit<NW>(cb, cp)) return c;   // blocked crate: no move
        const int i = sokf_crate_index(cr, np);
        c.cr = cr ^ ((uint64_t)(np ^ cp) << (8 * i));
        uint64_t nb[NW];
        for (int j = 0; j < NW; j++) nb[j] = cb[j];
        sokf_flip<NW>(nb, np); sokf_flip<NW>(nb, cp);
        if (sokf_any_and<NW>(nb, F.dead)) return c;              // checkDeadlock looks at every crate
        c.h = sokf_heuristic(F, c.cr);
    }
    c.ok = 1;
    return c;
}
template <int NW>
PCGRL_D SokChild sokf_child(const SokFastLevel<NW>& F, uint64_t cr, const uint64_t* cb, int player, int h, int d) {
    return sokf_child_dir<NW>(F, cr, cb, player, h, sokf_dir(d, F.w));
}
// How the four children of a pop get made: one after the other (host, generic), or -- on the device -- by four
// lanes at once (SokKidsLanes in kernels_sokoban.h), the rest of the search being uniform across those lanes.
struct SokKidsSerial {
    template <int NW>
    PCGRL_D void operator()(const SokFastLevel<NW>& F, uint64_t cr, const uint64_t* cb, int player, int h, SokChild* out) const {
        for (int d = 0; d < 4; d++) out[d] = sokf_child<NW>(F, cr, cb, player, h, d);
    }
};

// heapq on packed words.  A lone lane executes about one instruction every 5-6 cycles, so the sift loops are
// written for instruction count: while both children and all four grandchildren exist, two levels are
// walked per round without any bounds test (paired LDS reads); the last level or two use plain heapq.
template <class HP>
PCGRL_D void sokf_siftdown(HP heap, int pos) {
    const uint32_t newitem = heap[pos];
    while (pos > 0) {
        const int p1 = (pos - 1) >> 1;
        const int p2 = p1 > 0 ? (p1 - 1) >> 1 : 0;
        const uint32_t v1 = heap[p1], v2 = heap[p2];
        if (!sok_lt(newitem, v1)) break;
        heap[pos] = v1; pos = p1;
        if (pos == 0 || !sok_lt(newitem, v2)) break;
        heap[pos] = v2; pos = p2;
    }
    heap[pos] = newitem;
}
template <class HP>
PCGRL_D void sokf_siftup_root(HP heap, int endpos) {
    int pos = 0;
    const uint32_t newitem = heap[0];
    // three levels a round while all eight great-grandchildren exist: one LDS round trip (fourteen words) and three
    // compare-and-select steps instead of three round trips
    while (8 * pos + 14 < endpos) {
        const int c1 = 2 * pos + 1, g = 4 * pos + 3, t = 8 * pos + 7;
        const uint32_t a0 = heap[c1], a1 = heap[c1 + 1];
        const uint32_t g0 = heap[g], g1 = heap[g + 1], g2 = heap[g + 2], g3 = heap[g + 3];
        const uint32_t t0 = heap[t], t1 = heap[t + 1], t2 = heap[t + 2], t3 = heap[t + 3];
        const uint32_t t4 = heap[t + 4], t5 = heap[t + 5], t6 = heap[t + 6], t7 = heap[t + 7];
        const bool r1 = !sok_lt(a0, a1);
        const uint32_t u0 = r1 ? t4 : t0, u1 = r1 ? t5 : t1, u2 = r1 ? t6 : t2, u3 = r1 ? t7 : t3;
        const uint32_t b0 = r1 ? g2 : g0, b1 = r1 ? g3 : g1;
        heap[pos] = r1 ? a1 : a0;
        const int p1 = c1 + (r1 ? 1 : 0);
        const bool r2 = !sok_lt(b0, b1);
        const uint32_t e0 = r2 ? u2 : u0, e1 = r2 ? u3 : u1;
        heap[p1] = r2 ? b1 : b0;
        const int p2 = 2 * p1 + 1 + (r2 ? 1 : 0);
        const bool r3 = !sok_lt(e0, e1);
        heap[p2] = r3 ? e1 : e0;
        pos = 2 * p2 + 1 + (r3 ? 1 : 0);
    }
    while (4 * pos + 6 < endpos) {
        const int c1 = 2 * pos + 1, g = 4 * pos + 3;
        const uint32_t a0 = heap[c1], a1 = heap[c1 + 1];
        const uint32_t g0 = heap[g], g1 = heap[g + 1], g2 = heap[g + 2], g3 = heap[g + 3];
        const bool r1 = !sok_lt(a0, a1);
        heap[pos] = r1 ? a1 : a0;
        const int p1 = c1 + (r1 ? 1 : 0);
        const uint32_t b0 = r1 ? g2 : g0, b1 = r1 ? g3 : g1;
        const bool r2 = !sok_lt(b0, b1);
        heap[p1] = r2 ? b1 : b0;
        pos = 2 * p1 + 1 + (r2 ? 1 : 0);
    }
    int childpos = 2 * pos + 1;
    while (childpos < endpos) {
        const int rightpos = childpos + 1;
        uint32_t c = heap[childpos];
        if (rightpos < endpos) {
            const uint32_t r = heap[rightpos];
            if (!sok_lt(c, r)) { childpos = rightpos; c = r; }
        }
        heap[pos] = c;
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    heap[pos] = newitem;
    sokf_siftdown(heap, pos);
}

// ---- Two wavefronts per A* search (k_sokoban; round 3).  A wavefront that is alone on its SIMD gets one instruction per 5-9
// cycles (profiles/r3a_round3/valu_calibration.md), and a second wavefront on the same compute unit runs at full speed beside it:
// the only way to shorten a pop is to split its work.  Of a pop's ~3 900 cycles (tools/sok_prof.py) the heap operations -- the
// repair after the removal of the top, the appends of the children -- are 2 400 and everything else (loop head, node, crate
// bitboard, win test, visited probe, four children with their heuristics) 1 500, and the heap only ever needs the children's
// packed words.  So a second wavefront, the *heap server*, owns the heap: while the search wavefront expands the node of pop i
// the server removes that top and repairs (CPython heappop); at barrier (A) the two trade -- the children's words one way, the
// word of the top the repair left the other -- and the server appends the children (heappush, in their order) while the search
// wavefront is already on pop i + 1: the next top is the one the repair left unless a child has a strictly smaller priority, and
// then it is the first child with the smallest priority (a child climbs past everything that is not smaller than the root, or
// stays below the root), so the search wavefront works it out from the words it holds.  The removal is speculative (the search
// may end at this pop: cap, win, abandoned); a heap that is thrown away does not care.  ONE block barrier per pop; `SokDuoBox`
// in LDS carries the trade.  The array operations are those of the one-wavefront form in the same order, so the pop order,
// iteration counts and results are the same (tests: the fixtures' iteration counts; the heap primitives alone against Python's
// heapq through pcgrl_selftest_heap).
struct SokDuoBox {
    int session;             // outer handshake: 1 = a search starts, 0 = leave the kernel
    // two sets, used alternately (pop parity): with one barrier per pop the search wavefront is already filling in the children
    // of the next pop while the server still reads these
    int npush[2];            // (A) search -> server: children of this pop (0..4), or -1 = the search is over
    uint32_t push[2][4];
    uint32_t ahead_word[2];  // (A) server -> search: the packed word of the top the repair left (SOK_DUO_NONE: the heap is empty)
    // session == 2: a suspended search goes on (SokResume).  Its heap has resume_n entries, the pending pop's entry is out of it
    // already and resume_aw is the top that removal left: the server's first turn skips its removal.
    int resume_n; uint32_t resume_aw;
};
#define SOK_DUO_NONE 0xFFFFFFFFu
#if defined(__HIPCC__)
// (LDS traffic only has to have landed: the box and the heap live there; global loads may stay in flight across it)
__device__ __forceinline__ void sok_duo_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); }
#endif

// The lanes of a search run in lockstep on the same values, but the compiler cannot know: every `if` on a loaded value
// becomes a divergent branch (save exec, mask, restore, merge the loop masks: ~40 scalar instructions a pop that do nothing).
// SOK_UNI(c) states the uniformity -- a ballot that is compared with zero is a scalar condition -- and SOK_SCALAR(x) moves a
// uniform value to the scalar file.
#if defined(__HIP_DEVICE_COMPILE__)
#define SOK_UNI(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#define SOK_SCALAR(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define SOK_UNI(c) (c)
#define SOK_SCALAR(x) ((uint32_t)(x))
#endif

#if defined(__HIPCC__)
// The first child with the smallest priority among the (up to) four children the lanes 0..3 of a search hold, as a scalar word
// (SOK_DUO_NONE: no child): the minimum of (priority << 2 | lane) over the quad with two DPP steps, then one readlane -- instead of
// four readlanes and a chain of scalar compares.  `have`: this lane has a child; `word`: its packed word.
__device__ __forceinline__ uint32_t sok_duo_first_smallest(bool have, uint32_t word, int lane) {
    const uint32_t key = have ? (((word >> 16) << 2) | (uint32_t)(lane & 3)) : 0xFFFFFFFFu;
    uint32_t k1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)key, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    k1 = k1 < key ? k1 : key;
    uint32_t k2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)k1, 0x4E, 0xF, 0xF, true);        // quad_perm [2,3,0,1]
    k2 = k2 < k1 ? k2 : k1;
    const uint32_t kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)k2);
    if (kmin == 0xFFFFFFFFu) return SOK_DUO_NONE;
    return (uint32_t)__builtin_amdgcn_readlane((int)word, (int)(kmin & 3u));
}
#endif

#if defined(PCGRL_SMB_PROF) && defined(__HIPCC__)
extern __device__ unsigned long long* g_tl_buf;      // worklist.h (developer builds: tools/sok_prof.py)
#define SKP_DECL unsigned long long skp_t = clock64(), skp_a[6] = {0, 0, 0, 0, 0, 0}
#define SKP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = clock64(); skp_a[i] += n_ - skp_t; skp_t = n_; } while (0)
#define SKP_FLUSH(it) do { if (g_tl_buf && k >= 0) { for (int i_ = 0; i_ < 6; i_++) atomicAdd(&g_tl_buf[32 + i_], skp_a[i_]); atomicAdd(&g_tl_buf[38], (unsigned long long)(it)); atomicAdd(&g_tl_buf[39], 1ull); } } while (0)
// (the two-wavefront loops; only searches of at least PCGRL_SKD_MIN_POPS pops are counted: -DPCGRL_SKD_MIN_POPS=4000 looks at
//  the capped searches a lockstep step waits for)
#ifndef PCGRL_SKD_MIN_POPS
#define PCGRL_SKD_MIN_POPS 0
#endif
#define SKD_DECL unsigned long long skd_t = clock64(), skd_a[6] = {0, 0, 0, 0, 0, 0}
#define SKD_MARK(i) do { const unsigned long long n_ = clock64(); skd_a[i] += n_ - skd_t; skd_t = n_; } while (0)
#define SKD_MARKW(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SKD_MARK(i); } while (0)
#define SKD_FLUSH(base, it) do { if (g_tl_buf && (it) >= PCGRL_SKD_MIN_POPS) { for (int i_ = 0; i_ < 6; i_++) atomicAdd(&g_tl_buf[(base) + i_], skd_a[i_]); atomicAdd(&g_tl_buf[(base) + 6], (unsigned long long)(it)); atomicAdd(&g_tl_buf[(base) + 7], 1ull); } } while (0)
#else
#define SKD_DECL do {} while (0)
#define SKD_MARK(i) do {} while (0)
#define SKD_MARKW(i) do {} while (0)
#define SKD_FLUSH(base, it) do {} while (0)
#define SKP_DECL do {} while (0)
#define SKP(i) do {} while (0)
#define SKP_FLUSH(it) do {} while (0)
#endif
// ---- Searches that can be suspended and resumed (round 5: pcgrl_step_async, kernels_search_async.h).  A search is handed
// `SokResumeArg{state, limit}`: it stops *before* the pop that would take `iterations` past `limit`, leaves everything the next
// call needs in `state` (the scalars below; the caller keeps the pool, the visited table and the heap array as they are) and
// sets state->suspended.  A call with state->iterations > 0 continues that search instead of starting one: same pops in the
// same order as the uninterrupted search, so same result and same iteration count (tests/test_hostsim_algos.py holds the
// one-wavefront loops against themselves in one piece; the two-wavefront form is held against the oracle on the GPU).
// SokNoResume (the default) compiles every trace of this away: the lockstep kernels are the code they were.
struct SokResume {
    int32_t iterations, npool, head, heapn;       // heapn: the heap's entries (two-wavefront A*: after the removal for the pending pop)
    uint32_t cur_word, aw;                        // two-wavefront A*: the pending pop's word and the top its removal left (SokDuoBox)
    int32_t best_h, best_depth, have_best, suspended;
    uint64_t best_key;                            // (mdungeon, ddave: the key of the best node)
    int32_t best_aux, pad;                        // (ddave: the best node's air time / jumps word)
};
struct SokResumeArg { SokResume* st; int limit; };
struct SokNoResume {};
template <class RSP> struct SokRs { static constexpr bool on = false; };
template <> struct SokRs<SokResumeArg> { static constexpr bool on = true; };
PCGRL_D SokResume* sok_rs_state(const SokResumeArg& a) { return a.st; }
PCGRL_D SokResume* sok_rs_state(const SokNoResume&) { return nullptr; }
PCGRL_D int sok_rs_limit(const SokResumeArg& a) { return a.limit; }
PCGRL_D int sok_rs_limit(const SokNoResume&) { return 0x7FFFFFFF; }

// One search.  `table` must be all zeros; `cache`
// is room for four nodes (LDS on the device).  Same contract as sok_search otherwise.
template <int NW, class HP, class TP, class Hook, class Kids, class RSP = SokNoResume>
PCGRL_D bool sok_search_fast(const SokLevel& L, SokFastNode* pool, HP heap, TP table, int table_mask,
                             SokFastNode* cache, const SokNode& root, int k, int power, int& out_h, int& out_depth, int& out_iters,
                             bool& out_exhausted, Hook hook, Kids kids, SokDuoBox* duo = nullptr, RSP rsp = RSP()) {
    constexpr bool RS = SokRs<RSP>::on;
    SokResume* const rst = sok_rs_state(rsp);
    const int rs_limit = sok_rs_limit(rsp);
    const bool resumed = RS && rst->iterations > 0;
    bool suspended = false;
    SokFastLevel<NW> F;
    sokf_level(L, F);
    const int nc = F.nc;
    int npool = 0, head = 0, heapn = 0, iterations = 0, best_h = 0, best_depth = 0;
    bool have_best = false, aborted = false, win = false;
    SokFastNode n0;
    n0.cr = 0;
    for (int i = 0; i < nc; i++) n0.cr |= (uint64_t)root.crate[i] << (8 * i);
    n0.ph = (uint32_t)root.player | ((uint32_t)root.h << 16);
    n0.depth = 0;
    SokFastNode ahead = n0;
    int ahead_idx = 0, cache_base = 0, cache_n = 0;   // cache[j] = pool[cache_base + j], j < cache_n
    if (!resumed) {
        pool[0] = n0;
        npool = 1;
        if (k >= 0) { heap[0] = ((uint32_t)(2 * root.h + k * root.depth) << 16) | 0u; heapn = 1; }
    } else {      // pool, table and heap are the ones the suspended search left
        npool = rst->npool; head = rst->head; heapn = rst->heapn; iterations = rst->iterations;
        best_h = rst->best_h; best_depth = rst->best_depth; have_best = rst->have_best != 0;
        ahead_idx = -1;
    }
    int result_h = root.h, result_depth = 0;
#if defined(__HIPCC__)
    if (duo && k >= 0) {
        // the search wavefront of a two-wavefront A* search: see SokDuoBox.  (heap[0] = the root's word is in place.)
        uint32_t cur_word = (uint32_t)(2 * root.h + k * root.depth) << 16;     // the root's word: pool index 0
        int hn = 0;                                      // (RS) the heap's entries after the server's removal for the pending pop
        if (resumed) { cur_word = rst->cur_word; hn = heapn; duo->resume_n = heapn; duo->resume_aw = rst->aw; }
        duo->session = resumed ? 2 : 1;
        sok_duo_sync();                                  // (0) wakes the heap server of this block
        SKD_DECL;
        bool empty = false;
        int turn = 1;                                    // the pop the coming barrier (A) belongs to: its parity selects the set
        for (;;) {
            if (SOK_UNI(cur_word == SOK_DUO_NONE)) { empty = true; break; }
            if (iterations >= power) break;
            if (RS && iterations >= rs_limit) { suspended = true; break; }
            iterations++;
            if ((iterations & SOK_POLL_MASK) == 0 && SOK_UNI(hook(iterations))) { aborted = true; break; }     // (the hooks of the A* agents poll at that rate)
            const int cur = (int)(cur_word & 0xFFFFu);
            SokFastNode nd = ahead;
            if (SOK_UNI(cur != ahead_idx)) {
                if (SOK_UNI((unsigned)(cur - cache_base) < (unsigned)cache_n)) nd = cache[cur - cache_base];
                else nd = pool[cur];
            }
            const uint64_t cr = nd.cr;
            SKD_MARKW(2);
            const int node_player = (int)(nd.ph & 0xFFu), node_h = (int)(nd.ph >> 16), node_depth = (int)nd.depth;
            const uint64_t key = (cr << 8) | (uint64_t)node_player;
            const uint64_t hs = key * 0x9E3779B97F4A7C15ull;
            uint32_t slot = (uint32_t)(hs >> 40) & (uint32_t)table_mask;
            uint64_t v = table[slot];                    // (the visited probe is in flight while the crate bitboard is made)
            uint64_t cb[NW];
            for (int i = 0; i < NW; i++) cb[i] = 0;
            for (int i = 0; i < nc; i++) sokf_flip<NW>(cb, (int)((cr >> (8 * i)) & 0xFF));
            if (SOK_UNI(sokf_covers<NW>(cb, F.tmask))) { win = true; result_h = node_h; result_depth = node_depth; break; }   // engine.py:272-280
            bool seen = false;
            for (;;) {
                if (SOK_UNI(v == 0)) break;
                if (SOK_UNI(v == key)) { seen = true; break; }
                slot = (slot + 1) & (uint32_t)table_mask;
                v = table[slot];
            }
            SKD_MARKW(3);
            int npush = 0;
            uint32_t cmin = SOK_DUO_NONE;              // the first child with the smallest priority
            if (!seen) {
                table[slot] = key;
                cache_base = npool; cache_n = 0;
                const bool better = !have_best || node_h < best_h || (node_h == best_h && node_depth < best_depth);
                best_h = better ? node_h : best_h; best_depth = better ? node_depth : best_depth; have_best = true;
                // Node.getChildren: L, R, U, D -- lane d makes child d and files it itself (pool, cache, the server's box) at its
                // rank among the children that exist: one masked store each instead of four rounds of scalar copies
                const SokChild mine = kids.mine(F, cr, cb, node_player, node_h);
                SKD_MARKW(4);
                const uint32_t okm = (uint32_t)__builtin_amdgcn_ballot_w64(mine.ok != 0) & 15u;
                const int rank = __builtin_popcount(okm & ((1u << (kids.lane & 3)) - 1u));
                const uint32_t word = ((uint32_t)(2 * mine.h + k * (node_depth + 1)) << 16) | (uint32_t)(npool + rank);
                if (mine.ok) {
                    SokFastNode ch;
                    ch.cr = mine.cr; ch.ph = (uint32_t)mine.np | ((uint32_t)mine.h << 16); ch.depth = (uint32_t)(node_depth + 1);
                    pool[npool + rank] = ch;
                    cache[rank] = ch;
                    duo->push[turn & 1][rank] = word;
                }
                npush = __builtin_popcount(okm);
                cache_n = npush; npool += npush;
                cmin = sok_duo_first_smallest(mine.ok != 0, word, kids.lane);
            }
            duo->npush[turn & 1] = npush;
            if (RS) { hn += npush; hn -= hn > 0 ? 1 : 0; }         // the server appends the children, then removes the top for the next pop
            SKD_MARK(0);
            sok_duo_sync();                              // (A) children one way, the top the repair left the other
            SKD_MARK(1);
            const uint32_t aw = SOK_SCALAR(duo->ahead_word[turn & 1]);
            turn++;
            uint32_t nxt = aw;                           // the next pop: that top, unless a child is strictly smaller (then the first smallest)
            if (cmin != SOK_DUO_NONE && (nxt == SOK_DUO_NONE || sok_lt(cmin, nxt))) nxt = cmin;
            cur_word = nxt;
            ahead_idx = -1;
            if (SOK_UNI(aw != SOK_DUO_NONE && nxt == aw)) { ahead_idx = (int)(aw & 0xFFFFu); ahead = pool[ahead_idx]; }
            SKD_MARK(5);
        }
        duo->npush[turn & 1] = -1;                       // cap, empty heap, win or abandoned: the server leaves the search (only the
                                                         // set of the coming barrier is written: the server may still be reading the other)
        sok_duo_sync();                                  // (A)
        SKD_FLUSH(32, iterations);
        if (RS) {
            // suspended in front of the pop of `cur_word`: the server has taken that entry out of the heap already and repaired;
            // the top it left is in this turn's set.  The heap array stays as it is.
            rst->suspended = suspended ? 1 : 0;
            rst->iterations = iterations; rst->npool = npool; rst->head = 0; rst->heapn = hn;
            rst->cur_word = cur_word; rst->aw = SOK_SCALAR(duo->ahead_word[turn & 1]);
            rst->best_h = best_h; rst->best_depth = best_depth; rst->have_best = have_best ? 1 : 0;
        }
        if (!win) { result_h = best_h; result_depth = best_depth; }
        out_h = result_h; out_depth = result_depth; out_iters = iterations;
        out_exhausted = !win && !aborted && empty;
        return win;
    }
#endif
    SKP_DECL;
    while (iterations < power && (k >= 0 ? heapn > 0 : head < npool)) {
        if (RS && iterations >= rs_limit) { suspended = true; break; }
        iterations++;
        if (hook(iterations)) { aborted = true; break; }
        SKP(0);
        int cur;
        SokFastNode nd = ahead;
        if (k >= 0) {
            const uint32_t top = heap[0];
            const uint32_t last = heap[--heapn];
            cur = (int)(top & 0xFFFFu);
            if (cur != ahead_idx) {
                if ((unsigned)(cur - cache_base) < (unsigned)cache_n) nd = cache[cur - cache_base];
                else nd = pool[cur];
            }
            if (heapn > 0) { heap[0] = last; sokf_siftup_root(heap, heapn); }
            ahead_idx = -1;
            if (heapn > 0) { ahead_idx = (int)(heap[0] & 0xFFFFu); ahead = pool[ahead_idx]; }
        } else {
            cur = head++;
            if (cur != ahead_idx) nd = pool[cur];
            ahead_idx = -1;
            if (head < npool) { ahead_idx = head; ahead = pool[head]; }
        }
        SKP(1);
        const uint64_t cr = nd.cr;
        const int node_player = (int)(nd.ph & 0xFFu), node_h = (int)(nd.ph >> 16), node_depth = (int)nd.depth;
        uint64_t cb[NW];
        for (int i = 0; i < NW; i++) cb[i] = 0;
        for (int i = 0; i < nc; i++) sokf_flip<NW>(cb, (int)((cr >> (8 * i)) & 0xFF));
        if (sokf_covers<NW>(cb, F.tmask)) { win = true; result_h = node_h; result_depth = node_depth; break; }   // engine.py:272-280
        // visited test-and-add on the exact key
        const uint64_t key = (cr << 8) | (uint64_t)node_player;
        uint64_t hs = key * 0x9E3779B97F4A7C15ull;
        uint32_t slot = (uint32_t)(hs >> 40) & (uint32_t)table_mask;
        bool seen = false;
        for (;;) {
            const uint64_t v = table[slot];
            if (v == 0) break;
            if (v == key) { seen = true; break; }
            slot = (slot + 1) & (uint32_t)table_mask;
        }
        SKP(2);
        if (seen) continue;
        table[slot] = key;
        cache_base = npool; cache_n = 0;
        if (!have_best || node_h < best_h || (node_h == best_h && node_depth < best_depth)) { have_best = true; best_h = node_h; best_depth = node_depth; }
        SokChild kid[4];                        // Node.getChildren: L, R, U, D
        kids(F, cr, cb, node_player, node_h, kid);
        SKP(3);
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int d = 0; d < 4; d++) {
            if (!kid[d].ok) continue;
            SokFastNode ch;
            ch.cr = kid[d].cr; ch.ph = (uint32_t)kid[d].np | ((uint32_t)kid[d].h << 16); ch.depth = (uint32_t)(node_depth + 1);
            pool[npool] = ch;
            if (k >= 0) {
                cache[cache_n++] = ch;
                heap[heapn] = ((uint32_t)(2 * kid[d].h + k * (node_depth + 1)) << 16) | (uint32_t)npool;
                heapn++;
                sokf_siftdown(heap, heapn - 1);
            }
            npool++;
        }
        SKP(4);
    }
    SKP_FLUSH(iterations);
    if (RS) {
        rst->suspended = suspended ? 1 : 0;
        rst->iterations = iterations; rst->npool = npool; rst->head = head; rst->heapn = heapn;
        rst->best_h = best_h; rst->best_depth = best_depth; rst->have_best = have_best ? 1 : 0;
    }
    if (!win) { result_h = best_h; result_depth = best_depth; }
    out_h = result_h; out_depth = result_depth; out_iters = iterations;
    out_exhausted = !win && !aborted && !suspended && !(k >= 0 ? heapn > 0 : head < npool);
    return win;
}

#if defined(__HIPCC__)
// heappop's repair, six levels of the heap a round, on the server's 63 lanes.  CPython's _siftup walks the smaller-child path to a
// leaf moving every child up, puts the displaced last entry there and lets it climb back (_siftdown) while it is strictly smaller
// than its parent.  The priorities along that path never decrease, so the climb ends exactly where a top-down walk that stops at
// the first node whose smaller child is strictly greater than the entry (or that has no child) would have put it, and nothing
// below that node changes: same array, one pass.  A round covers the 63 nodes of the six levels below `pos`: lane j owns node j
// of that subtree (breadth-first), reads its two children with one ds_read2, and one compare each gives the "right child is not
// greater" bit and the "stop here" bit of all 63 nodes at once (two ballots).  A node is on the path when the direction bits of
// its ancestors lead to it and none of them stops: per lane two constant 64-bit masks (A: its ancestors, D: those it hangs under
// as a right child) and `(R & A) == D && !(S & A)` -- no walk.  The nodes on the path take their smaller child, the one that
// stops takes the entry; otherwise the deepest one's choice is the next round's `pos`.  ~45 instructions and one LDS round trip
// for six levels (the scalar walk: ~20 instructions and a third of a round trip per level).
struct SokDuoLanes { int lj, cj1; uint64_t A, D; };     // lane j: level and offset in the subtree, ancestor masks
__device__ __forceinline__ SokDuoLanes sok_duo_lanes(int lane) {
    SokDuoLanes T;
    const int j1 = lane + 1;
    T.lj = 31 - __builtin_clz(j1);
    T.cj1 = j1 - (1 << T.lj) - 1;
    T.A = 0; T.D = 0;
    for (int k = 1; k <= T.lj; k++) {
        const int a = (j1 >> k) - 1;
        T.A |= 1ull << a;
        if ((j1 >> (k - 1)) & 1) T.D |= 1ull << a;
    }
    return T;
}
// heap[0..n) with the root vacant: `item` (the old last entry) goes in; returns the new root's word.
__device__ __forceinline__ uint32_t sok_duo_repair(uint32_t* heap, int n, uint32_t item, int lane, const SokDuoLanes& T) {
    int pos = 0;
    uint32_t top = item;
    for (;;) {
        const int q = ((pos + 1) << T.lj) + T.cj1;
        const int lc = 2 * q + 1;
        const bool has = lane < 63 && lc < n;
        uint32_t a = 0, b = 0;
        if (has) { a = heap[lc]; b = heap[lc + 1]; }             // (index n still belongs to the heap's room)
        const bool r = has && lc + 1 < n && !sok_lt(a, b);
        const uint32_t m = r ? b : a;
        const bool stop = !has || sok_lt(item, m);
        const uint64_t R = __builtin_amdgcn_ballot_w64(r), S = __builtin_amdgcn_ballot_w64(stop);
        const bool on = lane < 63 && (R & T.A) == T.D && (S & T.A) == 0;
        const uint64_t OP = __builtin_amdgcn_ballot_w64(on);
        if (on && !stop) heap[q] = m;
        if (pos == 0 && !(S & 1ull)) top = (uint32_t)__builtin_amdgcn_readlane((int)m, 0);
        const uint64_t ST = OP & S;
        if (ST != 0) {                                            // exactly one node: the entry's place
            const int t = __builtin_ctzll(ST);
            const int qt = __builtin_amdgcn_readlane(q, t);
            heap[qt] = item;
            break;
        }
        const int d = 63 - __builtin_clzll(OP);                  // a node of the sixth level: the path goes on under it
        const int qd = __builtin_amdgcn_readlane(q, d);
        pos = 2 * qd + 1 + (int)((R >> d) & 1ull);
    }
    return top;
}

// heappush on the server's lanes: `item` goes to the new leaf `p` and climbs while it is strictly smaller than its parent
// (CPython _siftdown).  Lane k reads ancestor k of p (at most 16 levels), one ballot says which of them the item beats, the run of
// ones from the parent up is the climb: those ancestors move down one place each and the item lands above them.  One LDS round
// trip and a dozen instructions whatever the climb (the scalar loop: a round trip and ~12 instructions per two levels).
__device__ __forceinline__ void sok_duo_append(uint32_t* heap, int p, uint32_t item, int lane) {
    const int up = lane < 16 ? (p + 1) >> (lane + 1) : 0;            // ancestor `lane` of p is heap[up - 1]; 0: above the root
    const bool valid = up != 0;
    uint32_t v = 0;
    if (valid) v = heap[up - 1];
    const uint32_t beats = (uint32_t)__builtin_amdgcn_ballot_w64(valid && sok_lt(item, v));
    const int c = __builtin_ctz(~beats);                             // (bit 16 is never set)
    if (lane < c) heap[((p + 1) >> (lane & 15)) - 1] = v;            // ancestor k moves to where ancestor k - 1 (k = 0: the leaf) was
    if (lane == 0) heap[((p + 1) >> c) - 1] = item;
}

// The heap server: the second wavefront of a k_sokoban / k_mdungeon / k_ddave block (see SokDuoBox).  Waits for searches (barrier 0),
// owns their heap -- removes the top the search wavefront is expanding and repairs (sok_duo_repair), hands over the top that left,
// appends the children (sok_duo_append) -- and leaves when the block does.  All 64 lanes take part; the barriers are the wavefront's.
template <bool RS = false>
__device__ __forceinline__ void sok_duo_server(uint32_t* heap, SokDuoBox* box, int lane) {
    // (the appends: every lane runs the same chain on the same addresses -- the values are wave-uniform, so the compiler keeps
    //  the index arithmetic and the comparisons on the scalar unit; the repair after a removal: sok_duo_repair)
    const SokDuoLanes T = sok_duo_lanes(lane);
    for (;;) {
        sok_duo_sync();                                 // (0) a search starts, or the block is done
        if (box->session == 0) return;
        int n = 1;                                      // the root's word is in heap[0]
        bool skip = false;
        uint32_t aw0 = SOK_DUO_NONE;
        if (RS && box->session == 2) { n = box->resume_n; aw0 = box->resume_aw; skip = true; }
        SKD_DECL;
        int pop = 1;
        for (;; pop++) {                     // (the search wavefront's `iterations`: the parity selects the set)
            uint32_t aw = SOK_DUO_NONE;
            if (RS && skip) { aw = aw0; skip = false; }
            else if (n > 0) {                           // heappop of the entry the search wavefront is expanding: the last entry goes
                const uint32_t last = heap[--n];        // to the root and sinks (CPython _siftup)
                if (n > 0) aw = sok_duo_repair(heap, n, last, lane, T);
            }
            box->ahead_word[pop & 1] = aw;
            SKD_MARK(0);
            sok_duo_sync();                             // (A)
            SKD_MARK(1);
            const int m = box->npush[pop & 1];
            if (m < 0) break;
            const uint32_t mine = box->push[pop & 1][lane & 3];   // (one read for the four words)
            for (int j = 0; j < m; j++)                           // heappush, in the children's order
                sok_duo_append(heap, n + j, (uint32_t)__builtin_amdgcn_readlane((int)mine, j), lane);
            n += m;
            SKD_MARKW(2);
        }
        if (lane == 0) SKD_FLUSH(40, pop);
    }
}
#endif
