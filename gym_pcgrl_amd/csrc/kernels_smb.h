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
// successive jump columns", which is carried in the node (last jump column, widest gap so far).
// Search state: 8-byte nodes in the wavefront's global arena -- the node to be popped next is either fetched ahead or one of
// the four children just made, which stay in registers -- a heap of packed (priority << 16 | node) words whose first 8 192
// entries (levels 0..12; the median search is ~500 pops, nine in ten stay below 2 047) are in LDS and whose deeper levels
// continue in the arena, and the visited set as a bitmap over (x, y, airTime) in LDS.  Four searches per block (a wavefront
// each): ~35 KB of LDS per search.
#pragma once

#define SMB_LDS_HEAP 8192      /* heap words a search keeps in LDS (levels 0..12: enough for 2 047 pops); deeper levels live in its arena */
#define SMB_WAVES 4            /* searches a block runs side by side, a wavefront each */
#define SMB_MAX_H 32
#define SMB_YOFF 8            /* y ranges over [-5, H): a jump from the top row rises four cells above the screen */

// bytes of one search's arena: node pool + the heap's overflow beyond its LDS part (host: pcgrl_abi.hip sizes the block's share)
__host__ __device__ __forceinline__ size_t smb_wave_arena_bytes(int power) {
    const size_t nodes = 4 * (size_t)power + 4;
    const size_t ovf = nodes > SMB_LDS_HEAP ? nodes - SMB_LDS_HEAP : 0;
    return (nodes * 8 + ovf * 4 + 255) & ~(size_t)255;
}

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

struct SmbLevel { const uint64_t (*rows)[4]; int w, h, exit_x; };
struct SmbState { int x, y, air, depth, jumps, prev_jump_x, max_gap; };
__device__ __forceinline__ bool smb_solid(const SmbLevel& L, int x, int y) { return (L.rows[y][x >> 6] >> (x & 63)) & 1ull; }
__device__ __forceinline__ bool smb_movable(const SmbLevel& L, int x, int y) {          // engine.py:207-210
    if (y < 0) return true;
    if (x < 0 || x >= L.w || y >= L.h) return false;
    return !smb_solid(L, x, y);
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
// State.update (engine.py:212-250) for direction d = 0..3: (0,0), (1,0), (0,-1), (1,-1)
__device__ __forceinline__ SmbState smb_child(const SmbLevel& L, SmbState s, int d) {
    s.depth += 1;
    if (s.x >= L.exit_x || s.y >= L.h) return s;                      // checkOver: a finished state does not move
    const int dx = d & 1, jump = d >> 1;
    bool ground = false;
    if (s.y < L.h - 1 && s.y >= -1) ground = smb_solid(L, s.x, s.y + 1);
    int nx = s.x, ny = s.y;
    if (dx && smb_movable(L, nx + 1, ny)) nx += 1;
    if (jump) {
        if (ground && smb_movable(L, nx, ny - 1)) {
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
        if (smb_movable(L, nx, ny - 1)) ny -= 1; else s.air = 1;
    } else if (s.air == 1) {
        s.air = 0;
    } else if (smb_movable(L, nx, ny + 1)) {
        ny += 1;
    }
    s.x = nx; s.y = ny;
    return s;
}
// AStarAgent.getSolution (engine.py:101-126) by lanes 0..3 of a wavefront: everything is uniform across them (they keep the
// same heap, pool and visited set, writing the same values) except the four children of a pop, which they make side by side
// -- each child is a handful of dependent LDS lookups in the row masks.  `visited` must be all zeros.  Returns whether it won;
// `out` = the winning node's state, else the best node's (smallest h, then smallest depth, first seen).
__device__ __forceinline__ bool smb_search(const SmbLevel& L, const SmbState& root, int balance, int power, uint2* pool, const SmbHeap& H,
                                           uint32_t* visited, SmbState& out, int& out_iters, int lane) {
    const int ky = L.h + SMB_YOFF + 1;
    int npool = 1, heapn = 1, iterations = 0;
    pool[0] = smb_pack(root);
    H.set(0, ((uint32_t)(L.exit_x - root.x) << 16) | 0u);
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
        const SmbState s = smb_unpack(raw);
        if (s.y >= L.h) continue;                                     // checkLose
        if (s.x >= L.exit_x) { win = true; best = s; break; }         // checkWin
        const int key = (s.x * ky + (s.y + SMB_YOFF)) * 8 + s.air;
        const uint32_t bit = 1u << (key & 31);
        const uint32_t word = visited[key >> 5];
        if (word & bit) continue;
        visited[key >> 5] = word | bit;
        const int h = L.exit_x - s.x, bh = L.exit_x - best.x;
        if (!have_best || h < bh || (h == bh && s.depth < best.depth)) { have_best = true; best = s; }
        kid_base = npool;
        const uint2 mine = smb_pack(smb_child(L, s, lane & 3));       // Node.getChildren: (0,0), (1,0), (0,-1), (1,-1), one per lane
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint2 pk;
            pk.x = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, d);
            pk.y = (uint32_t)__builtin_amdgcn_readlane((int)mine.y, d);
            if (d == 0) kid0 = pk; else if (d == 1) kid1 = pk; else if (d == 2) kid2 = pk; else kid3 = pk;
            pool[npool] = pk;
            const int cx = (int)(pk.x & 255u), cdepth = (int)(pk.x >> 17);
            H.set(heapn, ((uint32_t)((L.exit_x - cx) + balance * cdepth) << 16) | (uint32_t)npool);
            heapn++;
            smb_siftdown(H, heapn - 1);
            npool++;
        }
    }
    out = best;
    out_iters = iterations;
    return win;
}

// Jobs = list_a (mode_a) followed by list_b (mode_b); list_b < 0: none.  `sync[0]` (zeroed by the host) hands the jobs out, a
// wavefront at a time.  Environments that finish their episode here go to `rst_list` (auto_reset).
__global__ __launch_bounds__(SMB_WAVES * 64) void k_smb(PcgrlParams P, DevBufs B, int list_a, int mode_a, int list_b, int mode_b, int parity, int rst_list,
                                                        int32_t* sync, int clear_parity, int lds_heap_n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smb_lds[];       // per wavefront: heap (lds_heap_n words), then the visited bitmap
    __shared__ uint64_t s_rows[SMB_WAVES][SMB_MAX_H][4];
    __shared__ int s_pref_a[WL_NSHARD + 1], s_pref_b[WL_NSHARD + 1];
    __shared__ int s_red[SMB_WAVES][8];
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_a = wl_load_prefix(B, parity, list_a, s_pref_a);
    const int n_b = list_b >= 0 ? wl_load_prefix(B, parity, list_b, s_pref_b) : 0;
    const int n = n_a + n_b;
    const int W = P.width, Hh = P.height, cells = W * Hh;
    const int ew = W + 6, ky = Hh + SMB_YOFF + 1;
    const int vis_words = (ew * ky * 8 + 31) / 32;
    uint32_t* my_lds = smb_lds + (size_t)wv * (lds_heap_n + ((vis_words + 3) & ~3));
    uint32_t* visited = my_lds + lds_heap_n;
    uint8_t* arena = reinterpret_cast<uint8_t*>(B.sok_pool) + (size_t)blockIdx.x * B.sok_pool_stride * sizeof(SokNode) + (size_t)wv * smb_wave_arena_bytes(P.solver_power);
    uint2* pool = reinterpret_cast<uint2*>(arena);
    SmbHeap HP = {my_lds, reinterpret_cast<uint32_t*>(arena + (4 * (size_t)P.solver_power + 4) * 8), lds_heap_n};
    uint64_t (*rows)[4] = s_rows[wv];
    int* red = s_red[wv];
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(sync, 1);
        t = __shfl(t, 0, 64);
        if (t >= n) break;
        int e, mode;
        if (t < n_a) { e = wl_get(B, list_a, s_pref_a, t); mode = mode_a; }
        else { e = wl_get(B, list_b, s_pref_b, t - n_a); mode = mode_b; }
        const uint8_t* m = B.map + (size_t)e * cells;
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
        // ---- the engine's grid as row bit masks (" # ## #": solid, brick, question and tube block)
        for (int y = 0; y < Hh; y++) {
            for (int k = 0; k < 4; k++) {
                const int ex = 64 * k + lane;
                bool sol = false;
                if (ex < ew) {
                    if (ex < 3) sol = y > Hh - 3;
                    else if (ex >= W + 3) sol = y > Hh - 3 || (y == Hh - 3 && ex == W + 4);
                    else { const int tl = m[y * W + ex - 3]; sol = tl == 1 || tl == 3 || tl == 4 || tl == 6; }
                }
                const uint64_t bal = __ballot(sol);
                if (lane == 0) rows[y][k] = bal;
            }
        }
        // ---- SMBProblem._run_game by one lane: AStarAgent with balance 1, then -- if it did not win -- balance 0 (smb_prob.py:133-141)
        int won = 0;
#pragma clang loop unroll(disable)
        for (int agent = 0; agent < 2 && !won; agent++) {
            for (int i = lane; i < vis_words; i += 64) visited[i] = 0;
            __threadfence_block();
            if (lane < 4) {
                SmbLevel L = {rows, ew, Hh, Hh > 3 ? W + 4 : -1};
                SmbState root = {1, Hh - 3, 0, 0, 0, 0, 0}, res = root;
                int it = 0;
                const bool win = smb_search(L, root, agent == 0 ? 1 : 0, P.solver_power, pool, HP, visited, res, it, lane);
                if (lane == 0) {
                    red[0] = win ? 1 : 0;
                    red[1] = res.jumps; red[2] = res.prev_jump_x; red[3] = res.max_gap; red[4] = res.x;
                }
            }
            __threadfence_block();
            won = red[0];
        }
        if (lane == 0) {
            const int exit_x = Hh > 3 ? W + 4 : -1;
            const int dist_win = red[0] ? 0 : exit_x - red[4];
            const int tail = P.prob_width - red[2];                   // smb_prob.py:166: max(value, self._width - prev_jump)
            const int jumps_dist = red[3] > tail ? red[3] : tail;
            int32_t s[PCGRL_MAX_STATS] = {c_floor, c_tubes, c_enemy, c_empty, c_noise, red[1], jumps_dist, dist_win};
            finalize_item<PCGRL_PROB_SMB>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), true, rst_list);
        }
        __threadfence_block();
    }
}
