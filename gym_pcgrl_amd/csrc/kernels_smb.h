// k_smb: SMBProblem.get_stats (probs/smb_prob.py:145-167) for the environments on a work list -- the five map statistics
// from the byte map and the play-through by two A* agents (probs/smb/engine.py), one wavefront per environment.
// Part of the single translation unit pcgrl_abi.hip.
//
// smb levels are 114 tiles wide: beyond the 64-bit row masks of the statistics kernels, and none of its statistics needs a
// flood fill, so this problem keeps no bit planes (pcgrl_layout.nplanes = 0) and works on the byte map:
//   dist-floor      helper.get_floor_dist(map, ["enemy"], solid/brick/question)   (helper.py:37-62)
//   disjoint-tubes  helper.get_type_grouping(map, ["tube"], [(-1,0),(1,0)], 1, 1)  (helper.py:100-108)
//   enemies, empty  tile counts;  noise = helper.get_changes horizontally + vertically (helper.py:120-138)
//   jumps, jumps-dist, dist-win   SMBProblem._run_game (smb_prob.py:106-145, 155-166): AStarAgent with balance 1, then --
//                   if that one did not win -- balance 0, solver_power pops each; the winner's (else the second agent's best
//                   node's) jump count, widest gap between successive jumps, and exit distance.
// The engine (engine.py:128-296): a grid of (W + 6) x H cells -- three padding columns either side, solid in the last two
// rows, the player at (1, H-3), a block under the pole at (W+4, H-3), exit column W+4 -- and a player state (x, y, airTime);
// four moves per state (stay / right / jump / right+jump), always four children; visited on pop by (x, y, airTime); the
// queue is CPython's heapq on h + balance * depth with h = exit - x.  jump_locs only ever feeds "widest gap between
// successive jump columns".
//
// The level lives in registers: lane l holds columns l, 64 + l, 128 + l, 192 + l as bit masks over the rows (bit y + 8 of a
// column: blocked; rows above the screen free, rows below it blocked), so a move is a 2 x 3 window around the player read with
// four readlanes and no memory access.  Two searches:
//  * smb_search_two_label -- balance 1, the whole wavefront, labels and level in registers, items in LDS (the deepest heap
//    level in the arena), see there: 24 of 25 searches;
//  * smb_search -- any balance, lanes 0..3 (the four children of a pop side by side), 8-byte nodes in the wavefront's global
//    arena and a heap of packed (priority << 16 | node) words whose first levels are in LDS: balance 0 (short searches on the
//    levels balance 1 did not win) and whatever does not fit the first.
// LDS per search: 2 048 heap words (deeper slots: the wavefront's arena) + the visited bitmap over (x, y, airTime): ~10 KB,
// sixteen searches per compute unit.
#pragma once

// developer build only (tools/smb_prof.py: -DPCGRL_SMB_PROF): cycles and pops of the searches, summed into g_tl_buf
#ifdef PCGRL_SMB_PROF
#define SP_ADD(i, v) do { if (lane == 0 && g_tl_buf) atomicAdd(&g_tl_buf[i], (unsigned long long)(v)); } while (0)
#define SP_NOW() clock64()
#else
#define SP_ADD(i, v) do {} while (0)
#define SP_NOW() 0ull
#endif

#define SMB_LDS_HEAP 2048      /* heap words a search keeps in LDS; deeper levels (slots) live in its arena */
#define SMB_MAX_WAVES 16       /* searches a block runs side by side, a wavefront each (the launch takes as many as its LDS allows) */
#define SMB_MAX_H 32
#define SMB_YOFF 8             /* y ranges over [-5, H): a jump from the top row rises four cells above the screen */
#define SMB_ROOT_PAR 0x3FFFu

// bytes of one search's arena: smb_search's node pool + its heap's overflow beyond the LDS part; smb_search_two_label keeps its
// expansion log (4 bytes per expansion) in the first and the deepest level of its heap in the second (host: pcgrl_abi.hip sizes the block's share)
__host__ __device__ __forceinline__ size_t smb_wave_arena_bytes(int power) {
    const size_t nodes = 4 * (size_t)power + 4;
    return (nodes * 8 + nodes * 4 + 255) & ~(size_t)255;
}

struct SmbCols { uint64_t c0, c1, c2, c3; };
struct SmbState { int x, y, air, depth, jumps, prev_jump_x, max_gap; };
struct SmbResult { int won, jumps, prev_jump_x, max_gap, x, iters; };

__device__ __forceinline__ uint64_t smb_readlane64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
// the 2 x 3 window around (x, y) -- x, y wavefront-uniform: wx / wx1 = columns x / x + 1, bit 0: row y - 1, bit 1: row y, bit 2: row y + 1
// (every column register is read with readlane and the choice made on the scalar side: a vector select would only be
// executed by the active lanes, and smb_search runs with lanes 0..3 active while it reads the columns of all 64)
template <int NW>
__device__ __forceinline__ uint64_t smb_column(const SmbCols& C, int x) {
    const int l = x & 63;
    const uint64_t v0 = smb_readlane64(C.c0, l), v1 = smb_readlane64(C.c1, l);
    if (NW == 2) return x < 64 ? v0 : v1;
    const uint64_t v2 = smb_readlane64(C.c2, l), v3 = smb_readlane64(C.c3, l);
    return x < 64 ? v0 : (x < 128 ? v1 : (x < 192 ? v2 : v3));
}
template <int NW>
__device__ __forceinline__ void smb_window(const SmbCols& C, int x, int y, uint32_t& wx, uint32_t& wx1) {
    wx = (uint32_t)(smb_column<NW>(C, x) >> (y + 7)) & 7u;
    wx1 = (uint32_t)(smb_column<NW>(C, x + 1) >> (y + 7)) & 7u;
}
// State.update (engine.py:212-250) for direction d = 0..3: (0,0), (1,0), (0,-1), (1,-1), on the window of the state (which
// is neither won nor lost: the searches test that before they expand a node)
__device__ __forceinline__ SmbState smb_child_win(SmbState s, int d, uint32_t wx, uint32_t wx1, int h) {
    s.depth += 1;
    const int dx = d & 1, jump = d >> 1;
    const bool ground = s.y < h - 1 && s.y >= -1 && ((wx >> 2) & 1u);
    int nx = s.x, ny = s.y;
    uint32_t wn = wx;
    if (dx && !((wx1 >> 1) & 1u)) { nx += 1; wn = wx1; }
    const bool up_free = !(wn & 1u), down_free = !((wn >> 2) & 1u);
    if (jump) {
        if (ground && up_free) {
            s.air = 5; s.jumps += 1;
            const int gap = s.x - s.prev_jump_x;                     // jump_locs.append((x, y)): the column before the move
            s.max_gap = gap > s.max_gap ? gap : s.max_gap;
            s.prev_jump_x = s.x;
        }
    } else if (s.air > 0) {
        s.air = 1;
    }
    if (s.air > 1) {
        s.air -= 1;
        if (up_free) ny -= 1; else s.air = 1;
    } else if (s.air == 1) {
        s.air = 0;
    } else if (down_free) {
        ny += 1;
    }
    s.x = nx; s.y = ny;
    return s;
}

// ---------------------------------------------------------------------------------------------------------------------
// The general search (any balance): lanes 0..3 of a wavefront
// ---------------------------------------------------------------------------------------------------------------------
struct SmbHeap {
    uint32_t* lds; uint32_t* glob; int lds_n;
    __device__ __forceinline__ uint32_t get(int i) const { return i < lds_n ? lds[i] : glob[i - lds_n]; }
    __device__ __forceinline__ void set(int i, uint32_t v) const { if (i < lds_n) lds[i] = v; else glob[i - lds_n] = v; }
};
__device__ __forceinline__ bool smb_lt(uint32_t a, uint32_t b) { return (a >> 16) < (b >> 16); }
// heapq._siftdown(heap, 0, pos) / heapq._siftup(heap, 0) on the packed words
__device__ __forceinline__ void smb_siftdown(const SmbHeap& H, int pos) {
    const uint32_t item = H.get(pos);
    while (pos > 0) {
        const int parent = (pos - 1) >> 1;
        const uint32_t pv = H.get(parent);
        if (!smb_lt(item, pv)) break;
        H.set(pos, pv);
        pos = parent;
    }
    H.set(pos, item);
}
__device__ __forceinline__ void smb_siftup_root(const SmbHeap& H, int endpos) {
    int pos = 0, child = 1;
    const uint32_t item = H.get(0);
    while (child < endpos) {
        uint32_t c = H.get(child);
        if (child + 1 < endpos) {
            const uint32_t r = H.get(child + 1);
            if (!smb_lt(c, r)) { child++; c = r; }
        }
        H.set(pos, c);
        pos = child;
        child = 2 * pos + 1;
    }
    H.set(pos, item);
    smb_siftdown(H, pos);
}
__device__ __forceinline__ uint2 smb_pack(const SmbState& s) {
    uint2 n;
    n.x = (uint32_t)s.x | ((uint32_t)(s.y + SMB_YOFF) << 8) | ((uint32_t)s.air << 14) | ((uint32_t)s.depth << 17);
    n.y = (uint32_t)s.jumps | ((uint32_t)s.prev_jump_x << 14) | ((uint32_t)s.max_gap << 22);
    return n;
}
__device__ __forceinline__ SmbState smb_unpack(uint2 n) {
    SmbState s;
    s.x = (int)(n.x & 255u); s.y = (int)((n.x >> 8) & 63u) - SMB_YOFF; s.air = (int)((n.x >> 14) & 7u); s.depth = (int)(n.x >> 17);
    s.jumps = (int)(n.y & 0x3FFFu); s.prev_jump_x = (int)((n.y >> 14) & 255u); s.max_gap = (int)((n.y >> 22) & 255u);
    return s;
}
// AStarAgent.getSolution (engine.py:101-126) by lanes 0..3 of a wavefront: everything is uniform across them (they keep the
// same heap, pool and visited set, writing the same values) except the four children of a pop, which they make side by side.
// `visited` must be all zeros.  out = the winning node's state, else the best node's (smallest h, then smallest depth, first seen).
__device__ __forceinline__ void smb_search(const SmbCols& C, int h, int exit_x, const SmbState& root, int balance, int power, uint2* pool, const SmbHeap& H,
                                           uint32_t* visited, SmbResult& out, int lane) {
    const int ky = h + SMB_YOFF + 1;
    int npool = 1, heapn = 1, iterations = 0;
    pool[0] = smb_pack(root);
    H.set(0, ((uint32_t)(exit_x - root.x) << 16) | 0u);
    bool have_best = false, win = false;
    SmbState best = root;
    uint2 ahead = pool[0];
    int ahead_idx = 0;
    // the four children of the last expansion (pool[kid_base .. kid_base + 3]): the next pop is usually one of them, and a
    // store to the arena followed by a load of the same node would cost a memory round trip per pop
    uint2 kid0 = ahead, kid1 = ahead, kid2 = ahead, kid3 = ahead;
    int kid_base = -8;
    while (iterations < power && heapn > 0) {
        iterations++;
        const uint32_t top = H.get(0), last = H.get(--heapn);
        const int cur = (int)(top & 0xFFFFu);
        uint2 raw;
        const unsigned kd = (unsigned)(cur - kid_base);
        if (cur == ahead_idx) raw = ahead;
        else if (kd < 4u) raw = kd == 0 ? kid0 : (kd == 1 ? kid1 : (kd == 2 ? kid2 : kid3));
        else raw = pool[cur];
        ahead_idx = -1;
        if (heapn > 0) {
            H.set(0, last);
            smb_siftup_root(H, heapn);
            ahead_idx = (int)(H.get(0) & 0xFFFFu);
            const unsigned ka = (unsigned)(ahead_idx - kid_base);
            ahead = ka < 4u ? (ka == 0 ? kid0 : (ka == 1 ? kid1 : (ka == 2 ? kid2 : kid3))) : pool[ahead_idx];
        }
        raw.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.x);
        raw.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.y);
        const SmbState s = smb_unpack(raw);
        if (s.y >= h) continue;                                       // checkLose
        if (s.x >= exit_x) { win = true; best = s; break; }           // checkWin
        const int key = (s.x * ky + (s.y + SMB_YOFF)) * 5 + s.air;          // airTime of a stored state is 0..4
        const uint32_t bit = 1u << (key & 31);
        const uint32_t word = visited[key >> 5];
        if (word & bit) continue;
        visited[key >> 5] = word | bit;
        const int hh = exit_x - s.x, bh = exit_x - best.x;
        if (!have_best || hh < bh || (hh == bh && s.depth < best.depth)) { have_best = true; best = s; }
        kid_base = npool;
        uint32_t wx, wx1;
        smb_window<4>(C, s.x, s.y, wx, wx1);
        const uint2 mine = smb_pack(smb_child_win(s, lane & 3, wx, wx1, h));    // Node.getChildren: (0,0), (1,0), (0,-1), (1,-1), one per lane
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint2 pk;
            pk.x = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, d);
            pk.y = (uint32_t)__builtin_amdgcn_readlane((int)mine.y, d);
            if (d == 0) kid0 = pk; else if (d == 1) kid1 = pk; else if (d == 2) kid2 = pk; else kid3 = pk;
            pool[npool] = pk;
            const int cx = (int)(pk.x & 255u), cdepth = (int)(pk.x >> 17);
            H.set(heapn, ((uint32_t)((exit_x - cx) + balance * cdepth) << 16) | (uint32_t)npool);
            heapn++;
            smb_siftdown(H, heapn - 1);
            npool++;
        }
    }
    out.won = win ? 1 : 0;
    out.jumps = best.jumps; out.prev_jump_x = best.prev_jump_x; out.max_gap = best.max_gap; out.x = best.x; out.iters = iterations;
    SP_ADD(0, 1); SP_ADD(1, iterations);
}

// ---------------------------------------------------------------------------------------------------------------------
// AStarAgent with balance 1, by a whole wavefront
// ---------------------------------------------------------------------------------------------------------------------
// With balance 1 the priority is f = (exit - x) + depth; a child is one move deeper and at most one column further, so its f
// is its parent's (it moved right) or one more, pops come in non-decreasing f, and at any time the queue holds only
// f = fmin (label 0) and f = fmin + 1 (label 1).  heapq on such a queue needs no priority comparisons, only the labels:
//   heappush: a label-1 item stays where it is appended; a label-0 item climbs past its label-1 ancestors (they move down one
//             level each) and stops under the first label-0 ancestor;
//   heappop:  the vacated root is filled along a path that takes the right child unless (left, right) = (0, 1), down to a
//             leaf -- inside the label-1 part that is "right while there is one, then left", a closed form; the last item goes
//             to the leaf and, if its label is 0, climbs back to just below the zeros of that path.  Net effect: the first m
//             path entries move up one level and the last item lands on path position m (m = the leaf's depth for a label-1
//             last item, the number of zeros on the path below the root otherwise), and at most one label flips.  The
//             label-0 part of that path is the same from one pop to the next except at its end: it is kept, not searched for.
// The labels of heap slots 1..4095 (1-based heap index) are bits spread over the wavefront's registers (word w in lane w & 63
// of a / b), read with readlane: finding a path costs no memory access, and the moves along it are one LDS read and one LDS
// write with a lane per level.  An item is 32 bits: x | (y + 8) << 8 | airTime << 14 | parent << 17, `parent` = the expansion
// number of the node it is a child of (the root: SMB_ROOT_PAR).  Depth follows from the label (depth = f - (exit - x)); what
// the reference carries along in jump_locs is recovered at the end by walking the result's ancestors through the expansion
// log (log[e] = item of the e-th expanded node, in the wavefront's arena): a node was made by a jump start exactly when its
// airTime is 4 -- the jump sets 5, the same update takes it to 4, and nothing else produces a 4 -- and the jump's column is its
// parent's x.  Returns 1 won, 0 not won, 2 the queue outgrew `cap` slots (the caller repeats the search with smb_search).
struct SmbLab { uint32_t a, b; };
__device__ __forceinline__ uint32_t smb_lab_word(const SmbLab& Lb, int w) {
    return (uint32_t)__builtin_amdgcn_readlane((int)(w < 64 ? Lb.a : Lb.b), w & 63);
}
__device__ __forceinline__ int smb_lab_get(const SmbLab& Lb, int q) { return (int)((smb_lab_word(Lb, q >> 5) >> (q & 31)) & 1u); }
// (branch-free: the owner lane's mask is the bit, everybody else's is 0 -- vector selects instead of exec-mask juggling, which is
// scalar work, and the scalar pipe is the busier one in this kernel)
__device__ __forceinline__ void smb_lab_set(SmbLab& Lb, int q, int v, int lane) {
    const int w = q >> 5;
    const uint32_t m = lane == (w & 63) ? 1u << (q & 31) : 0u;
    const uint32_t ma = w < 64 ? m : 0u, mb = w < 64 ? 0u : m;
    if (v) { Lb.a |= ma; Lb.b |= mb; } else { Lb.a &= ~ma; Lb.b &= ~mb; }
}
// "right while there is one, then left" from slot q of a heap of n slots: the leaf the fill path of a pop ends on once it is
// inside the label-1 part (and the whole path when every label is 0)
__device__ __forceinline__ int smb_rightmost_leaf(int q, int n) {
    const int a = q + 1, bb = n + 1;
    int t = (31 - __builtin_clz(bb)) - (31 - __builtin_clz(a));
    if ((a << t) > bb) t--;
    q = (a << t) - 1;
    return 2 * q <= n ? 2 * q : q;
}
// The chain "root, then the right child if its label is 0, else the left child if its label is 0" is the part of a pop's fill
// path that lies among the label-0 items; `u` is its last slot (0: there is no label-0 item).  It is kept up to date across
// pushes and pops (a push adds one label-0 slot, a pop takes away at most the chain's end and the last slot), so a pop does not
// walk down from the root: only when the end of the chain goes away and its left sibling carries a label 0 does the chain
// continue downwards from there (1.2 steps a pop on average instead of 6.5).
__device__ __forceinline__ int smb_chain_descend(const SmbLab& lab, int q, int n) {
    for (;;) {
        const int c = 2 * q;
        if (c > n) return q;
        const uint32_t w = smb_lab_word(lab, c >> 5) >> (c & 31);          // bit 0: label(left), bit 1: label(right)
        if (c < n && !(w & 2u)) q = c + 1;
        else if (!(w & 1u)) q = c;
        else return q;
    }
}
__device__ __forceinline__ int smb_chain_remove_end(const SmbLab& lab, int e, int n) {      // slot e, the chain's end, is no label-0 slot any more
    if (e == 1) return 0;
    if ((e & 1) && smb_lab_get(lab, e - 1) == 0) return smb_chain_descend(lab, e - 1, n);   // (e - 1 < e <= n + 1)
    return e >> 1;
}
__device__ __forceinline__ int smb_chain_add(int u, int g) {                                 // slot g (its parent has label 0, or g = 1) got label 0
    if (g == 1) return 1;
    if (u == 0) return u;
    const int p = g >> 1;
    const int du = 31 - __builtin_clz(u), dp = 31 - __builtin_clz(p);
    if (du >= dp && (u >> (du - dp)) == p && (p == u || (g & 1))) return g;
    return u;
}
// The items of the two-label heap: slots below `lds_n` in LDS, the rest (the deepest level of a large heap: only the searches
// whose queue grows beyond lds_n slots ever touch it) in the wavefront's arena.  Half the LDS per search buys twice the
// searches per compute unit; a slot in the arena costs the wavefront that needs it a memory round trip, which the other
// wavefronts of the SIMD fill.
struct SmbItems {
    uint32_t* lds; uint32_t* glob; int lds_n;
    __device__ __forceinline__ uint32_t get(int q) const {
        return q < lds_n ? lds[q] : __hip_atomic_load(&glob[q - lds_n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void set(int q, uint32_t v) const { if (q < lds_n) lds[q] = v; else glob[q - lds_n] = v; }
};
// a label-0 item just appended at slot q (its label bit already 0) climbs to its place; returns the slot it ends on
__device__ __forceinline__ int smb_climb(SmbLab& lab, const SmbItems& ent, int q, uint32_t item, int lane) {
    // b = how many label-1 ancestors it passes: lane j looks at the j-th ancestor (the label words come over with bpermute),
    // one ballot (labels along a root path are zeros, then ones)
    const int sh = lane + 1 < 31 ? lane + 1 : 31;
    const int anc = q >> sh, w = anc >> 5;
    uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute((w & 63) << 2, (int)lab.a);
    if (q >= 2048) {                                                            // (wavefront-uniform) slots 2048.. have their labels in b
        const uint32_t wb = (uint32_t)__builtin_amdgcn_ds_bpermute((w & 63) << 2, (int)lab.b);
        word = w < 64 ? word : wb;
    }
    const bool one = anc >= 1 && ((word >> (anc & 31)) & 1u);
    const int b = __builtin_ctzll(~__ballot(one));
    if (b > 0) {
        uint32_t mv = item;
        if (lane < b) mv = ent.lds[q >> (lane + 1)];                              // (ancestors: always in LDS)
        __builtin_amdgcn_wave_barrier();
        if (lane < b) ent.set(q >> lane, mv);
        if (lane == 0) ent.lds[q >> b] = item;
        smb_lab_set(lab, q >> b, 0, lane);
        smb_lab_set(lab, q, 1, lane);
    }
    return q >> b;
}
__device__ __forceinline__ int smb_search_two_label(const SmbCols& C, int h, int exit_x, int root_x, int root_y, int power, const SmbItems& ent, int cap,
                                                    uint32_t* visited, uint32_t* log, SmbResult& out, int lane) {
    const int ky = h + SMB_YOFF + 1;
    SmbLab lab = {0u, 0u};
    int n = 1, u = 1, iterations = 0, nexp = 0, fmin = exit_x - root_x;
    const uint32_t root = (uint32_t)root_x | ((uint32_t)(root_y + SMB_YOFF) << 8) | (SMB_ROOT_PAR << 17);
    if (lane == 0) ent.lds[1] = root;
    bool have_best = false;
    int status = 0, best_x = root_x, best_depth = 0;
    uint32_t res = root;
    while (iterations < power && n > 0) {
        iterations++;
        uint32_t rp = ent.lds[1];
        const uint32_t lastp = ent.get(n);
        const bool relabel = (smb_lab_word(lab, 0) & 2u) != 0;                  // no label-0 item left: the ones become the zeros
        if (relabel) { lab.a = 0u; lab.b = 0u; fmin++; }
        const int ll = smb_lab_get(lab, n);
        const int nold = n;
        n--;
        if (n > 0) {
            if (relabel) u = smb_rightmost_leaf(1, n);
            else if (u == nold) u = smb_chain_remove_end(lab, nold, n);
            // the fill path: the chain, and for a label-1 last item on through the ones to a leaf; every entry on it moves up
            // one level and the last item takes the end of it
            const int leaf = ll ? smb_rightmost_leaf(u, n) : u, k = 31 - __builtin_clz(leaf);
            uint32_t mv = lastp;
            const bool act = lane < k;
            if (act) mv = ent.get(leaf >> (k - lane - 1));
            __builtin_amdgcn_wave_barrier();
            if (act) ent.lds[leaf >> (k - lane)] = mv;                          // (a parent: in LDS)
            if (lane == 0) ent.set(leaf, lastp);
            if (ll) {                                                           // the labels on the path move up with the entries: one flip
                smb_lab_set(lab, u, 1, lane);
                u = smb_chain_remove_end(lab, u, n);
            }
        } else {
            u = 0;
        }
        rp = (uint32_t)__builtin_amdgcn_readfirstlane((int)rp);
        const int x = (int)(rp & 255u), y = (int)((rp >> 8) & 63u) - SMB_YOFF, air = (int)((rp >> 14) & 7u);
        if (y >= h) continue;                                                   // checkLose
        const int depth = fmin - (exit_x - x);                                  // the popped item's label is 0
        if (x >= exit_x) { status = 1; res = rp; break; }                       // checkWin
        const int key = (x * ky + (y + SMB_YOFF)) * 5 + air;                // airTime of a stored state is 0..4
        const uint32_t bit = 1u << (key & 31);
        const uint32_t word = (uint32_t)__builtin_amdgcn_readfirstlane((int)visited[key >> 5]);
        if (word & bit) continue;
        if (lane == 0) visited[key >> 5] = word | bit;
        const int hh = exit_x - x, bh = exit_x - best_x;
        if (!have_best || hh < bh || (hh == bh && depth < best_depth)) { have_best = true; best_x = x; best_depth = depth; res = rp; }
        if (n + 4 > cap) { status = 2; break; }
        if (lane == 0) log[nexp] = rp;
        uint32_t wx, wx1;
        smb_window<2>(C, x, y, wx, wx1);
        SmbState s = {x, y, air, 0, 0, 0, 0};
        s = smb_child_win(s, lane & 3, wx, wx1, h);                             // Node.getChildren: (0,0), (1,0), (0,-1), (1,-1), one per lane
        const uint32_t mine = (uint32_t)s.x | ((uint32_t)(s.y + SMB_YOFF) << 8) | ((uint32_t)s.air << 14) | ((uint32_t)nexp << 17);
        nexp++;
        if (lane < 4) ent.set(n + 1 + lane, mine);
        // labels of the four: 1 for a child that stayed in its column, 0 for one that moved right (children 1 and 3, both or neither)
        const bool canr = !((wx1 >> 1) & 1u);
        {
            const int q0 = n + 1, sh = q0 & 31, w0 = q0 >> 5;
            const uint64_t msk = 0xFull << sh, pat = (canr ? 0x5ull : 0xFull) << sh;
            const uint32_t m_lo = (uint32_t)msk, m_hi = (uint32_t)(msk >> 32), p_lo = (uint32_t)pat, p_hi = (uint32_t)(pat >> 32);
            const bool own0 = lane == (w0 & 63), own1 = lane == ((w0 + 1) & 63);
            const uint32_t c0 = own0 ? m_lo : 0u, s0 = own0 ? p_lo : 0u, c1 = own1 ? m_hi : 0u, s1 = own1 ? p_hi : 0u;
            const bool a0 = w0 < 64, a1 = w0 + 1 < 64;
            lab.a = (lab.a & ~((a0 ? c0 : 0u) | (a1 ? c1 : 0u))) | (a0 ? s0 : 0u) | (a1 ? s1 : 0u);
            lab.b = (lab.b & ~((a0 ? 0u : c0) | (a1 ? 0u : c1))) | (a0 ? 0u : s0) | (a1 ? 0u : s1);
        }
        n += 4;
        if (canr) {
            u = smb_chain_add(u, smb_climb(lab, ent, n - 2, (uint32_t)__builtin_amdgcn_readlane((int)mine, 1), lane));
            u = smb_chain_add(u, smb_climb(lab, ent, n, (uint32_t)__builtin_amdgcn_readlane((int)mine, 3), lane));
        }
    }
    // jump_locs of the result, from its ancestors (newest first): jumps, the last jump's column, the widest gap between
    // successive jump columns counted from column 0 (smb_prob.py:155-166)
    int jumps = 0, later = -1, max_gap = 0, prev_jump_x = 0;
    const unsigned long long t_wb = SP_NOW();
    if (status != 2) {
        __threadfence_block();
        uint32_t node = res;
        for (;;) {
            const uint32_t par = node >> 17;
            if (par == SMB_ROOT_PAR) break;
            const uint32_t pe = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&log[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (((node >> 14) & 7u) == 4u) {
                const int jx = (int)(pe & 255u);
                jumps++;
                if (later < 0) prev_jump_x = jx;
                else max_gap = later - jx > max_gap ? later - jx : max_gap;
                later = jx;
            }
            node = pe;
        }
        if (later > max_gap) max_gap = later;
    }
    out.won = status == 1; out.jumps = jumps; out.prev_jump_x = prev_jump_x; out.max_gap = max_gap; out.x = (int)(res & 255u); out.iters = iterations;
    SP_ADD(2, 1); SP_ADD(3, iterations); SP_ADD(4, status == 2); SP_ADD(7, SP_NOW() - t_wb);
    return status;
}

// Which cells of the level did a search read?  Only expansions read it (smb_window), an expanded state (x, y, airTime) is exactly a
// bit of the visited bitmap, and what its window reads depends on (x, y) alone: column x in the rows y - 1 .. y + 1, of column x + 1
// the row y and -- when that cell is free, so that the player can move there -- the rows y - 1 and y + 1 (smb_child_win reads
// nothing else).  So nothing is recorded while a search runs: afterwards lane l collects, for its columns l + 64 k, the rows with an
// expanded state from the bitmap and widens them.  A level that differs from this one only in cells outside that set is searched
// step for step the same way.  seenw: uint32 [W] of the environment (DevBufs::champ), bit y of word c = row y of map column c
// (engine column c + 3) was read; `first`: the play-through's first search stores, the later ones add -- the set lives in memory, not
// in registers across the searches (k_smb sits at its register limit).
__device__ __noinline__ void smb_seen_accumulate(uint64_t c0, uint64_t c1, uint64_t c2, uint64_t c3, const uint32_t* visited, int h, int ew, int W, int lane,
                                                 uint32_t* seenw, bool first) {
    // (a real call, and one set of columns at a time: k_smb sits at its register limit and this runs once per search)
    const int ky = h + SMB_YOFF + 1;
    uint64_t prev63 = 0ull;                                             // the expanded rows of column 64 k - 1
#pragma clang loop unroll(disable)
    for (int k = 0; k < 4 && 64 * k < ew + 1; k++) {
        const int x = 64 * k + lane;
        uint64_t own = 0ull;
        if (x < ew) {
            for (int yy = 0; yy < ky; yy++) {                           // yy = y + SMB_YOFF
                const int off = (x * ky + yy) * 5, w = off >> 5, sh = off & 31;
                uint32_t v = visited[w] >> sh;
                if (sh > 27) v |= visited[w + 1] << (32 - sh);
                own |= (uint64_t)((v & 31u) != 0u) << yy;
            }
        }
        // the expanded rows of the column to the left: lane l - 1, for lane 0 lane 63 of the previous set
        uint64_t left = (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)own, 1, 64) | ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(own >> 32), 1, 64) << 32);
        if (lane == 0) left = prev63;
        prev63 = smb_readlane64(own, 63);
        const uint64_t col = k == 0 ? c0 : (k == 1 ? c1 : (k == 2 ? c2 : c3));
        const uint64_t wide = own | (left & ~col), stayed = left & col;
        const uint32_t rows = (uint32_t)((wide | (wide << 1) | (wide >> 1) | stayed) >> SMB_YOFF);
        const int c = x - 3;
        if (c >= 0 && c < W) seenw[c] = first ? rows : (seenw[c] | rows);
    }
}

// SMBProblem.get_stats of one map (`m`: its tile bytes, in global memory or -- right after an in-kernel reset -- in LDS) by one
// wavefront, and the end of the step / reset it belongs to (finalize_item).  Returns whether the episode ended (auto_reset).
struct SmbWave {
    uint32_t* heap; uint32_t* visited; uint8_t* arena; int lds_heap_n, vis_words;
};
// keep_play (MODE_STEP, flagged by k_update): the change left every cell the last play-through read as it was -- its three results
// are taken from the previous statistics and nothing is searched.
__device__ __forceinline__ bool smb_job(const PcgrlParams& P, const DevBufs& B, const SmbWave& S, int e, int mode, const uint8_t* m, int parity, int rst_list,
                                        bool push_reset, int lane, bool keep_play = false) {
    const int W = P.width, Hh = P.height, cells = W * Hh;
    const int ew = W + 6;
    const int exit_x = Hh > 3 ? W + 4 : -1;
    // ---- the five statistics of the byte map (tiles: 0 empty 1 solid 2 enemy 3 brick 4 question 5 coin 6 tube)
    int c_floor = 0, c_tubes = 0, c_enemy = 0, c_empty = 0, c_noise = 0;
    for (int c = lane; c < cells; c += 64) {
        const int y = c / W, x = c - y * W;
        const int tl = m[c];
        c_empty += tl == 0;
        c_enemy += tl == 2;
        if (x > 0) c_noise += tl != m[c - 1];
        if (y > 0) c_noise += tl != m[c - W];
        if (tl == 6) {
            const int v = (x > 0 && m[c - 1] == 6) + (x < W - 1 && m[c + 1] == 6);
            c_tubes += v == 1;
        }
        if (tl == 2) {                                            // _calc_dist_floor: the first floor tile at or below the cell
            int r = Hh - 1;
            for (int dy = 0; y + dy < Hh; dy++) {
                const int tb = m[c + dy * W];
                if (tb == 1 || tb == 3 || tb == 4) { r = dy - 1; break; }
            }
            c_floor += r;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        c_floor += __shfl_xor(c_floor, o, 64); c_tubes += __shfl_xor(c_tubes, o, 64); c_enemy += __shfl_xor(c_enemy, o, 64);
        c_empty += __shfl_xor(c_empty, o, 64); c_noise += __shfl_xor(c_noise, o, 64);
    }
    if (keep_play) {
        int done = 0;
        if (lane == 0) {
            const int32_t* prev = B.stats + (size_t)e * 8;
            int32_t s[PCGRL_MAX_STATS] = {c_floor, c_tubes, c_enemy, c_empty, c_noise, prev[5], prev[6], prev[7]};
            done = finalize_item<PCGRL_PROB_SMB>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), push_reset, rst_list) ? 1 : 0;
        }
        __threadfence_block();
        return __shfl(done, 0, 64) != 0;
    }
    // ---- the engine's grid (" # ## #": solid, brick, question and tube block) as column masks: lane l holds columns l + 64 k
    SmbCols C;
    {
        uint64_t col[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ex = 64 * k + lane;
            uint64_t c = ~0ull << (Hh + SMB_YOFF);
            if (ex < ew) {
                for (int y = 0; y < Hh; y++) {
                    bool sol;
                    if (ex < 3) sol = y > Hh - 3;
                    else if (ex >= W + 3) sol = y > Hh - 3 || (y == Hh - 3 && ex == W + 4);
                    else { const int tl = m[y * W + ex - 3]; sol = tl == 1 || tl == 3 || tl == 4 || tl == 6; }
                    c |= (uint64_t)sol << (y + SMB_YOFF);
                }
            } else {
                c = ~0ull << SMB_YOFF;
            }
            col[k] = c;
        }
        C.c0 = col[0]; C.c1 = col[1]; C.c2 = col[2]; C.c3 = col[3];
    }
    __builtin_amdgcn_wave_barrier();                                  // (m may be in the LDS the searches are about to use)
    // ---- SMBProblem._run_game: AStarAgent with balance 1, then -- if it did not win -- balance 0 (smb_prob.py:133-141)
    const bool two_label = ew <= 128 && P.solver_power < (int)SMB_ROOT_PAR;
    uint2* pool = reinterpret_cast<uint2*>(S.arena);
    const SmbHeap HP = {S.heap, reinterpret_cast<uint32_t*>(S.arena + (4 * (size_t)P.solver_power + 4) * 8), S.lds_heap_n};
    const SmbItems items = {S.heap, HP.glob, S.lds_heap_n};           // (the arena part of smb_search's heap: free while this search runs)
    // the label bits cover slots 1..4095; the parents of all of them (slots < 2048) must be in LDS for the arena part to be used
    const int cap = S.lds_heap_n >= 2048 ? 4095 : S.lds_heap_n - 1;
    SmbResult res = {0, 0, 0, 0, 1, 0};
    int first_iters = 0;
    // the cells the play-through reads (smb_seen_accumulate after every search; k_update: a change elsewhere keeps the play-through)
#ifdef PCGRL_SMB_NO_SEEN     /* developer build (tools/smb_prof.py): the searches without the pass over the visited bitmap, for A/B timing */
    uint32_t* const seenw = nullptr;
#else
    uint32_t* const seenw = B.champ != nullptr ? reinterpret_cast<uint32_t*>(B.champ) + (size_t)e * W : nullptr;
#endif
    bool seen_first = true;
#pragma clang loop unroll(disable)
    for (int agent = 0; agent < 2 && !res.won; agent++) {
        for (int i = lane; i < S.vis_words; i += 64) S.visited[i] = 0;
        __threadfence_block();
        int status = 2;
        if (agent == 0 && two_label) {
            const unsigned long long t0 = SP_NOW();
            status = smb_search_two_label(C, Hh, exit_x, 1, Hh - 3, P.solver_power, items, cap, S.visited, reinterpret_cast<uint32_t*>(S.arena), res, lane);
            SP_ADD(5, SP_NOW() - t0);
            __threadfence_block();
            if (status == 2) {
                for (int i = lane; i < S.vis_words; i += 64) S.visited[i] = 0;
                __threadfence_block();
            } else if (seenw) {
                smb_seen_accumulate(C.c0, C.c1, C.c2, C.c3, S.visited, Hh, ew, W, lane, seenw, seen_first);
                seen_first = false;
            }
        }
        if (status == 2) {
            const unsigned long long t0 = SP_NOW();
            if (lane < 4) {
                const SmbState root = {1, Hh - 3, 0, 0, 0, 0, 0};
                smb_search(C, Hh, exit_x, root, agent == 0 ? 1 : 0, P.solver_power, pool, HP, S.visited, res, lane);
            }
            res.won = __shfl(res.won, 0, 64); res.jumps = __shfl(res.jumps, 0, 64); res.prev_jump_x = __shfl(res.prev_jump_x, 0, 64);
            res.max_gap = __shfl(res.max_gap, 0, 64); res.x = __shfl(res.x, 0, 64); res.iters = __shfl(res.iters, 0, 64);
            SP_ADD(6, SP_NOW() - t0);
            __threadfence_block();
            if (seenw) { smb_seen_accumulate(C.c0, C.c1, C.c2, C.c3, S.visited, Hh, ew, W, lane, seenw, seen_first); seen_first = false; }
        }
        if (agent == 0) first_iters = res.iters;
    }
    int done = 0;
    if (lane == 0) {
        B.sok_cnt[e] = first_iters;      // how long this level's play-through was: the next step's map differs by a tile (k_update: SMB_LONG_POPS)
        const int dist_win = res.won ? 0 : exit_x - res.x;
        const int tail = P.prob_width - res.prev_jump_x;              // smb_prob.py:166: max(value, self._width - prev_jump)
        const int jumps_dist = res.max_gap > tail ? res.max_gap : tail;
        int32_t s[PCGRL_MAX_STATS] = {c_floor, c_tubes, c_enemy, c_empty, c_noise, res.jumps, jumps_dist, dist_win};
        done = finalize_item<PCGRL_PROB_SMB>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), push_reset, rst_list) ? 1 : 0;
    }
    __threadfence_block();
    return __shfl(done, 0, 64) != 0;
}

// Jobs = list_a (mode_a) followed by list_b (mode_b); list_b < 0: none.  `sync[0]` (zeroed by the host) hands the jobs out, a
// wavefront at a time.  An environment that finishes its episode here (a level that can be won ends it: smb_prob.py:191-192) is
// reset by the same wavefront right away (`inline_reset`: PcgrlEnv.reset, reset_env.h; the new map's statistics and play-through
// follow as MODE_START) -- one launch and one tail of long searches per step instead of two; without `inline_reset` it goes to
// `rst_list`.
// (a template only so that every part of the library can include this header: instantiated where it is launched)
template <int PART_TAG>
__global__ __launch_bounds__(SMB_MAX_WAVES * 64) void k_smb(PcgrlParams P, DevBufs B, int list_pre, int list_a, int mode_a, int list_b, int mode_b, int parity,
                                                            int rst_list, int32_t* sync, int clear_parity, int lds_heap_n, int inline_reset, int gen_map) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smb_lds[];       // per wavefront: heap (lds_heap_n words), then the visited bitmap
    __shared__ int s_pref_p[WL_NSHARD + 1], s_pref_a[WL_NSHARD + 1], s_pref_b[WL_NSHARD + 1];
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // list_pre (mode_a): the levels k_update expects to take long (their last play-through did) -- they go first, so that the
    // launch does not end on a long search that was started late
    const int n_p = list_pre >= 0 ? wl_load_prefix(B, parity, list_pre, s_pref_p) : 0;
    const int n_a = n_p + wl_load_prefix(B, parity, list_a, s_pref_a);
    const int n_b = list_b >= 0 ? wl_load_prefix(B, parity, list_b, s_pref_b) : 0;
    const int n = n_a + n_b;
    const int cells = P.width * P.height;
    const int vis_words = ((P.width + 6) * (P.height + SMB_YOFF + 1) * 5 + 31) / 32;
    uint32_t* my_lds = smb_lds + (size_t)wv * (lds_heap_n + ((vis_words + 3) & ~3));
    const SmbWave S = {my_lds, my_lds + lds_heap_n,
                       reinterpret_cast<uint8_t*>(B.sok_pool) + (size_t)blockIdx.x * B.sok_pool_stride * sizeof(SokNode) + (size_t)wv * smb_wave_arena_bytes(P.solver_power),
                       lds_heap_n, vis_words};
    uint32_t* mt = my_lds;                                                    // in-kernel reset: MT19937 ring + tile bytes, where the heap is between searches
    uint8_t* tiles = reinterpret_cast<uint8_t*>(my_lds + PCGRL_MT_N);
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(sync, 1);
        t = __shfl(t, 0, 64);
        if (t >= n) break;
        int e, mode;
        if (t < n_p) { e = wl_get(B, list_pre, s_pref_p, t); mode = mode_a; }
        else if (t < n_a) { e = wl_get(B, list_a, s_pref_a, t - n_p); mode = mode_a; }
        else { e = wl_get(B, list_b, s_pref_b, t - n_a); mode = mode_b; }
        const bool keep_play = (e & SMB_KEEP_PLAY) != 0;       // (k_update: the play-through of the map before the change still holds)
        e &= ~SMB_KEEP_PLAY;
        const bool ended = smb_job(P, B, S, e, mode, B.map + (size_t)e * cells, parity, rst_list, !inline_reset, lane, keep_play);
        if (ended && inline_reset) {
            wave_reset_env<PCGRL_PROB_SMB>(P, B, e, gen_map, mt, tiles, lane);
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            smb_job(P, B, S, e, MODE_START, tiles, parity, rst_list, false, lane);
        }
    }
}
