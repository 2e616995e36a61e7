// HIP kernels (gfx950 / CDNA4) and C ABI of the batched PCGRL environment.
//
// One `pcgrl_step` is three launches on the caller's stream (five for Sokoban):
//
//   k_update   thread per environment.  Representation.update (narrow_rep.py:99-114, wide_rep.py:67-70,
//              turtle_rep.py:101-129), counters + heatmap (pcgrl_env.py:130-137).  Unchanged
//              environments are finished here (reward 0, done, info); changed ones are compacted into
//              a sharded work list, bucketed by expected difficulty (LDS histogram, one atomic per bucket
//              per 256-thread block).
//   k_stats    one lane group (16 lanes = one DPP row, or a full wavefront for maps taller than 16) per
//              changed environment: Problem.get_stats as row-bitboard programs (pcgrl_algos.h), get_reward,
//              get_episode_over, get_debug_info (pcgrl_env.py:138-148).  Done environments go to the
//              reset list.
//   k_sokoban  (Sokoban only) one wavefront per solver job parked by k_stats / k_reset.
//   k_reset    one wavefront per environment to reset: PcgrlEnv.reset (pcgrl_env.py:66-76): the MT19937
//              ring is staged in LDS and the wave produces 128 words per round, tiles are drawn with
//              numpy's choice() rule, written coalesced as uint8 and transposed through LDS into the
//              row bit planes; cursor draw; BinaryProblem.reset (binary_prob.py:68-72); start stats
//              (problem.py:45-46) on the rows that are already in registers.
//
// State is structure-of-arrays over the environment axis, all in HBM, owned by the caller
// (include/pcgrl_hip.h).  The uint8 map is the observation; the kernels compute on `planes`
// (row bitboards of the tile-id bits, [N][nplanes][group]) which k_update keeps in sync, so the
// statistics never re-read or transpose the byte map.  No MFMA: integer/bit work only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#include "../../include/pcgrl_hip.h"
#include "lanegroup_dev.h"
#include "mt19937.h"
#include "pcgrl_algos.h"
#include "sokoban_solver.h"

#define PCGRL_BLOCK 256
enum { MODE_STEP = 0, MODE_START = 1, MODE_SETMAP = 2 };

// Work lists (changed environments, environments to reset, sokoban solver jobs).  A list is 64 shards,
// each with its own counter on its own 64-byte line and its own segment of the item array, so appends
// never pile up on one address (thousands of same-address atomics per step cost ~12 ns each);
// consumers rebuild the dense index with a 64-entry prefix sum in LDS.  Counters are double-buffered
// by step parity: the last kernel of a step zeroes the other parity's counters.
#define WL_NSHARD 64
#define WL_CSTRIDE 16
enum { WL_CHG = 0, WL_RST = 1, WL_SOL = 2, WL_SOL2 = 3, WL_NLIST = 4 };   // SOL: solver jobs of the step, SOL2: of the resets

struct DevBufs {
    uint8_t* map; uint8_t* old_map; uint16_t* heat; uint8_t* pos; void* planes;
    int32_t* counters; int32_t* stats; int32_t* start_stats; int32_t* info;
    double* reward; uint8_t* done; double* tile_p;
    uint32_t* rng_rep; uint32_t* rng_prob; int32_t* rng_cur;
    int32_t* wl_cnt;                 // [2 parities][WL_NLIST][WL_NSHARD * WL_CSTRIDE]
    int32_t* wl_items[WL_NLIST];     // [WL_NSHARD][wl_cap[list]]
    int32_t wl_cap[WL_NLIST];
    // sokoban solver arena (per resident solver block) and sticky status word
    SokNode* sok_pool; uint32_t* sok_heap; uint32_t* sok_table; int32_t* status;
    int32_t sok_pool_stride, sok_heap_stride, sok_table_size, sok_use_lds;
};

__device__ __forceinline__ int32_t* wl_counters(const DevBufs& B, int parity, int list) {
    return B.wl_cnt + (size_t)(parity * WL_NLIST + list) * WL_NSHARD * WL_CSTRIDE;
}
__device__ __forceinline__ void wl_push(const DevBufs& B, int parity, int list, int shard, int value) {
    const int i = atomicAdd(wl_counters(B, parity, list) + shard * WL_CSTRIDE, 1);
    B.wl_items[list][(size_t)shard * B.wl_cap[list] + i] = value;
}
// Every thread of the block calls this once; s_pref has WL_NSHARD + 1 entries.  Returns the list length.
__device__ __forceinline__ int wl_load_prefix(const DevBufs& B, int parity, int list, int* s_pref) {
    if (threadIdx.x < WL_NSHARD) {
        int v = wl_counters(B, parity, list)[threadIdx.x * WL_CSTRIDE];
        for (int o = 1; o < WL_NSHARD; o <<= 1) {
            const int t = __shfl_up(v, o, 64);
            if ((int)threadIdx.x >= o) v += t;
        }
        s_pref[threadIdx.x + 1] = v;
        if (threadIdx.x == 0) s_pref[0] = 0;
    }
    __syncthreads();
    return s_pref[WL_NSHARD];
}
__device__ __forceinline__ int wl_get(const DevBufs& B, int list, const int* s_pref, int i) {
    int lo = 0;
#pragma unroll
    for (int step = WL_NSHARD / 2; step > 0; step >>= 1)
        if (s_pref[lo + step] <= i) lo += step;
    return B.wl_items[list][(size_t)lo * B.wl_cap[list] + (i - s_pref[lo])];
}
__device__ __forceinline__ void wl_clear(const DevBufs& B, int parity) {   // one block, any size
    for (int i = threadIdx.x; i < WL_NLIST * WL_NSHARD; i += blockDim.x) wl_counters(B, parity, 0)[i * WL_CSTRIDE] = 0;
}

// ------------------------------------------------------------------------------------------
// Block-wide stream compaction into a work list: every thread of the block must call this.
__device__ __forceinline__ void block_append(bool flag, int value, const DevBufs& B, int parity, int list,
                                             int* s_cnt, int* s_base) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int shard = blockIdx.x & (WL_NSHARD - 1);
    const uint64_t m = __ballot(flag);
    if (lane == 0) s_cnt[w] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        *s_base = tot ? atomicAdd(wl_counters(B, parity, list) + shard * WL_CSTRIDE, tot) : 0;
    }
    __syncthreads();
    if (flag) {
        int off = *s_base;
        for (int i = 0; i < w; i++) off += s_cnt[i];
        off += __popcll(m & ((1ull << lane) - 1ull));
        B.wl_items[list][(size_t)shard * B.wl_cap[list] + off] = value;
    }
}

// Changed environments are bucketed by how hard their statistics are expected to be (the previous
// stats are a good predictor: one tile changed), one bucket per shard, so that the four maps sharing a
// wavefront in k_stats have similar trip counts.  Every thread of the block must call this.
__device__ __forceinline__ void block_append_bucketed(bool flag, int bucket, int value, const DevBufs& B, int parity, int list,
                                                      int* s_hist, int* s_gbase) {
    if (threadIdx.x < WL_NSHARD) s_hist[threadIdx.x] = 0;
    __syncthreads();
    int rank = 0;
    if (flag) rank = atomicAdd(&s_hist[bucket], 1);
    __syncthreads();
    if (threadIdx.x < WL_NSHARD) {
        const int c = s_hist[threadIdx.x];
        if (c > 0) s_gbase[threadIdx.x] = atomicAdd(wl_counters(B, parity, list) + threadIdx.x * WL_CSTRIDE, c);
    }
    __syncthreads();
    if (flag) B.wl_items[list][(size_t)bucket * B.wl_cap[list] + s_gbase[bucket] + rank] = value;
}
__device__ __forceinline__ int difficulty_bucket(const PcgrlParams& P, const int4& s0, const int4& s1) {
    if (P.prob == PCGRL_PROB_BINARY) {   // (path-length / 6, regions / 3), 8 x 8
        const int a = min(max(s0.y, 0) / 6, 7), b = min(max(s0.x, 0) / 3, 7);
        return a * 8 + b;
    }
    const int regions = P.prob == PCGRL_PROB_ZELDA ? s1.x : s0.w;
    return min(max(regions, 0), WL_NSHARD - 1);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------------------------------------
// k_update: thread per environment
// The kernel is a chain of dependent scattered loads per thread at one wavefront per SIMD, so it is
// written to keep that chain at three round trips: (1) action, counters, cursor, old stats; (2) the map
// cell, its plane words and the MT19937 ring words of up to PCGRL_SPEC_DRAWS speculative draws (every
// operand of draw i is an *old* word: distance 397); (3) the heatmap cell of the new cursor.
#define PCGRL_SPEC_DRAWS 6
template <int REP, class MaskT>
__global__ __launch_bounds__(PCGRL_BLOCK) void k_update(PcgrlParams P, DevBufs B, const int32_t* __restrict__ actions, int parity) {
    __shared__ int s_cnt[2][4];
    __shared__ int s_base[2];
    __shared__ int s_hist[WL_NSHARD], s_gbase[WL_NSHARD];
    const int e = blockIdx.x * PCGRL_BLOCK + threadIdx.x;
    const bool act = e < P.num_envs;
    bool chg = false, rst = false;
    int bucket = 0;
    if (act) {
        const int W = P.width, H = P.height, G = P.group, NPL = P.nplanes;
        // ---- round trip 1
        const int2 c = reinterpret_cast<const int2*>(B.counters)[e];
        int a0_ = 0, a1_ = 0, a2_ = 0;
        if (REP == PCGRL_REP_WIDE) { a0_ = actions[3 * e + 0]; a1_ = actions[3 * e + 1]; a2_ = actions[3 * e + 2]; }
        else a0_ = actions[e];
        uchar2 p0 = make_uchar2(0, 0);
        if (REP != PCGRL_REP_WIDE) p0 = reinterpret_cast<const uchar2*>(B.pos)[e];
        const bool draws = REP == PCGRL_REP_NARROW && P.random_tile;
        int cur = 0;
        if (draws) cur = B.rng_cur[2 * e];
        const int4* sp = reinterpret_cast<const int4*>(B.stats + (size_t)e * 8);
        const int4* tp = reinterpret_cast<const int4*>(B.start_stats + (size_t)e * 8);
        const int4 s0 = sp[0], s1 = sp[1], t0 = tp[0], t1 = tp[1];

        bucket = difficulty_bucket(P, s0, s1);
        const int iter = c.x + 1;
        int changes = c.y;
        int x = p0.x, y = p0.y;
        int tile = -1, wx = 0, wy = 0, hx = 0, hy = 0;
        if (REP == PCGRL_REP_NARROW) {
            const int a = clampi(a0_, 0, P.ntiles);
            if (a > 0) tile = a - 1;
            wx = x; wy = y;
        } else if (REP == PCGRL_REP_WIDE) {
            wx = clampi(a0_, 0, W - 1);
            wy = clampi(a1_, 0, H - 1);
            tile = clampi(a2_, 0, P.ntiles - 1);
            hx = wx; hy = wy;
        } else {
            const int a = clampi(a0_, 0, P.ntiles + 3);
            if (a < 4) {   // turtle_rep.py:18,103-125: L,R,U,D with clamp or warp on both axes
                const int dx = (a == 0) ? -1 : (a == 1 ? 1 : 0), dy = (a == 2) ? -1 : (a == 3 ? 1 : 0);
                x += dx;
                if (x < 0) x = P.warp ? x + W : 0;
                if (x >= W) x = P.warp ? x - W : W - 1;
                y += dy;
                if (y < 0) y = P.warp ? y + H : 0;
                if (y >= H) y = P.warp ? y - H : H - 1;
            } else {
                tile = a - 4;
            }
            wx = x; wy = y; hx = x; hy = y;
        }
        // ---- round trip 2: everything addressed by the cursor cell and by the ring cursor
        uint8_t* cell = B.map + ((size_t)e * H + wy) * W + wx;
        MaskT* pl = reinterpret_cast<MaskT*>(B.planes) + (size_t)e * NPL * G + wy;
        const int old = *cell;
        MaskT m0 = pl[0], m1 = 0, m2 = 0;
        if (NPL > 1) { m1 = pl[G]; m2 = pl[2 * G]; }
        uint32_t* ring = B.rng_rep + (size_t)e * PCGRL_MT_N;
        uint32_t xa[PCGRL_SPEC_DRAWS + 1], xb[PCGRL_SPEC_DRAWS];
        if (draws) {
#pragma unroll
            for (int i = 0; i <= PCGRL_SPEC_DRAWS; i++) xa[i] = ring[mt_wrap(cur + i)];
#pragma unroll
            for (int i = 0; i < PCGRL_SPEC_DRAWS; i++) xb[i] = ring[mt_wrap(mt_wrap(cur + PCGRL_MT_M) + i)];
        }
        if (tile >= 0 && old != tile) {
            chg = true;
            *cell = (uint8_t)tile;
            const MaskT bit = (MaskT)1 << wx;
            pl[0] = (tile & 1) ? (m0 | bit) : (m0 & ~bit);
            if (NPL > 1) {
                pl[G] = (tile & 2) ? (m1 | bit) : (m1 & ~bit);
                pl[2 * G] = (tile & 4) ? (m2 | bit) : (m2 & ~bit);
            }
        }
        if (REP == PCGRL_REP_NARROW) {   // the cursor moves on every step (narrow_rep.py:104-113)
            if (draws) {
                // numpy randint(W) then randint(H): masked rejection, consumed in order from the speculative words
                const uint32_t rx = (uint32_t)(W - 1), ry = (uint32_t)(H - 1);
                uint32_t mx = rx, my = ry;
                mx |= mx >> 1; mx |= mx >> 2; mx |= mx >> 4; mx |= mx >> 8; mx |= mx >> 16;
                my |= my >> 1; my |= my >> 2; my |= my >> 4; my |= my >> 8; my |= my >> 16;
                int stage = 0, used = 0;             // stage 0: drawing x, 1: drawing y, 2: done
                if (rx == 0) { x = 0; stage = 1; }   // randint(1) draws nothing
                if (stage == 1 && ry == 0) { y = 0; stage = 2; }
#pragma unroll
                for (int i = 0; i < PCGRL_SPEC_DRAWS; i++) {
                    if (stage < 2) {
                        const uint32_t yv = mt_twist(xa[i], xa[i + 1], xb[i]);
                        ring[mt_wrap(cur + i)] = yv;
                        used = i + 1;
                        const uint32_t v = mt_temper(yv);
                        if (stage == 0) {
                            if ((v & mx) <= rx) { x = (int)(v & mx); stage = 1; if (ry == 0) { y = 0; stage = 2; } }
                        } else {
                            if ((v & my) <= ry) { y = (int)(v & my); stage = 2; }
                        }
                    }
                }
                cur = mt_wrap(cur + used);
                if (stage < 2) {                      // (1/8)^k tail: finish with ordinary draws
                    if (stage == 0) { x = mt_randint(ring, cur, W); y = mt_randint(ring, cur, H); }
                    else y = mt_randint(ring, cur, H);
                }
                B.rng_cur[2 * e] = cur;
            } else {
                x += 1;
                if (x >= W) { x = 0; y += 1; if (y >= H) y = 0; }
            }
            hx = x; hy = y;   // pcgrl_env.py:137 marks the *new* cursor cell
        }
        // ---- round trip 3
        if (chg) {
            changes += 1;
            B.heat[((size_t)e * H + hy) * W + hx] += 1;
        }
        reinterpret_cast<int2*>(B.counters)[e] = make_int2(iter, changes);
        if (REP != PCGRL_REP_WIDE) reinterpret_cast<uchar2*>(B.pos)[e] = make_uchar2((unsigned char)x, (unsigned char)y);
        if (!chg) {
            // new_stats is old_stats (pcgrl_env.py:132-142): reward 0, done/info from the current stats
            int32_t s[PCGRL_MAX_STATS], st[PCGRL_MAX_STATS];
            s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
            st[0] = t0.x; st[1] = t0.y; st[2] = t0.z; st[3] = t0.w; st[4] = t1.x; st[5] = t1.y; st[6] = t1.z; st[7] = t1.w;
            const bool d = episode_over(P, s, st) || changes >= P.max_changes || iter >= P.max_iterations;
            B.reward[e] = 0.0;
            B.done[e] = d ? 1 : 0;
            int32_t* inf = B.info + (size_t)e * 10;
            inf[0] = s0.x; inf[1] = s0.y; inf[2] = s0.z; inf[3] = s0.w;
            inf[4] = s1.x; inf[5] = s1.y; inf[6] = s1.z; inf[7] = s1.w;
            if (P.prob == PCGRL_PROB_BINARY) inf[2] = s0.y - t0.y;   // path-imp (binary_prob.py:137)
            inf[8] = iter; inf[9] = changes;
            rst = d && P.auto_reset;
        }
    }
    // bucketing pays where four maps share a wavefront and their cost varies a lot (binary); elsewhere the
    // plain per-block append is cheaper (kernel-uniform branch)
    if (P.prob == PCGRL_PROB_BINARY && P.group == 16) block_append_bucketed(chg, bucket, e, B, parity, WL_CHG, s_hist, s_gbase);
    else block_append(chg, e, B, parity, WL_CHG, s_cnt[0], &s_base[0]);
    block_append(rst, e, B, parity, WL_RST, s_cnt[1], &s_base[1]);
}

// ------------------------------------------------------------------------------------------
// k_update_block: the 3x3 "cast" / "multi" representations (narrow_cast_rep.py:36-59, narrow_multi_rep.py:39-59,
// turtle_cast_rep.py:38-76).  Same contract as k_update; up to nine tiles change per step and `change`
// counts them (pcgrl_env.py:136 adds it to _changes; the heatmap still gets +1).
template <int REP, class MaskT>
__global__ __launch_bounds__(PCGRL_BLOCK) void k_update_block(PcgrlParams P, DevBufs B, const int32_t* __restrict__ actions, int parity) {
    __shared__ int s_cnt[2][4];
    __shared__ int s_base[2];
    const int e = blockIdx.x * PCGRL_BLOCK + threadIdx.x;
    const bool act = e < P.num_envs;
    bool chg = false, rst = false;
    if (act) {
        const int W = P.width, H = P.height, G = P.group, NPL = P.nplanes, NT = P.ntiles;
        const int2 c = reinterpret_cast<const int2*>(B.counters)[e];
        const uchar2 p0 = reinterpret_cast<const uchar2*>(B.pos)[e];
        const int iter = c.x + 1;
        int changes = c.y, x = p0.x, y = p0.y;
        int vals[9];
#pragma unroll
        for (int i = 0; i < 9; i++) vals[i] = -1;
        if (REP == PCGRL_REP_NARROW_MULTI) {
#pragma unroll
            for (int i = 0; i < 9; i++) { const int a = clampi(actions[9 * e + i], 0, NT); vals[i] = a - 1; }
        } else {
            const int type = actions[2 * e], value = clampi(actions[2 * e + 1], 0, NT - 1);
            if (REP == PCGRL_REP_NARROW_CAST) {
                const int t = clampi(type, 0, 2);
                if (t == 1) vals[4] = value;
                if (t == 2) { for (int i = 0; i < 9; i++) vals[i] = value; }
            } else {
                const int t = clampi(type, 0, 5);
                if (t < 4) {   // turtle move (turtle_rep.py:103-125 semantics)
                    const int dx = (t == 0) ? -1 : (t == 1 ? 1 : 0), dy = (t == 2) ? -1 : (t == 3 ? 1 : 0);
                    x += dx;
                    if (x < 0) x = P.warp ? x + W : 0;
                    if (x >= W) x = P.warp ? x - W : W - 1;
                    y += dy;
                    if (y < 0) y = P.warp ? y + H : 0;
                    if (y >= H) y = P.warp ? y - H : H - 1;
                }
                if (t == 4) vals[4] = value;
                if (t == 5) { for (int i = 0; i < 9; i++) vals[i] = value; }
            }
        }
        int change = 0;
        uint8_t* map_e = B.map + (size_t)e * H * W;
        MaskT* pl_e = reinterpret_cast<MaskT*>(B.planes) + (size_t)e * NPL * G;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            if (vals[(dy + 1) * 3] < 0 && vals[(dy + 1) * 3 + 1] < 0 && vals[(dy + 1) * 3 + 2] < 0) continue;
            MaskT m0 = pl_e[yy], m1 = NPL > 1 ? pl_e[G + yy] : (MaskT)0, m2 = NPL > 1 ? pl_e[2 * G + yy] : (MaskT)0;
            bool touched = false;
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const int xx = x + dx, v = vals[(dy + 1) * 3 + dx + 1];
                if (xx < 0 || xx >= W || v < 0) continue;
                uint8_t* cell = map_e + yy * W + xx;
                if (*cell != v) {
                    change++;
                    touched = true;
                    *cell = (uint8_t)v;
                    const MaskT bit = (MaskT)1 << xx;
                    m0 = (v & 1) ? (m0 | bit) : (m0 & ~bit);
                    m1 = (v & 2) ? (m1 | bit) : (m1 & ~bit);
                    m2 = (v & 4) ? (m2 | bit) : (m2 & ~bit);
                }
            }
            if (touched) { pl_e[yy] = m0; if (NPL > 1) { pl_e[G + yy] = m1; pl_e[2 * G + yy] = m2; } }
        }
        if (REP != PCGRL_REP_TURTLE_CAST) {   // narrow cursor move (narrow_rep.py:104-113), after the write
            if (P.random_tile) {
                uint32_t* ring = B.rng_rep + (size_t)e * PCGRL_MT_N;
                int cur = B.rng_cur[2 * e];
                x = mt_randint(ring, cur, W);
                y = mt_randint(ring, cur, H);
                B.rng_cur[2 * e] = cur;
            } else {
                x += 1;
                if (x >= W) { x = 0; y += 1; if (y >= H) y = 0; }
            }
        }
        if (change > 0) {
            chg = true;
            changes += change;
            B.heat[((size_t)e * H + y) * W + x] += 1;
        }
        reinterpret_cast<int2*>(B.counters)[e] = make_int2(iter, changes);
        reinterpret_cast<uchar2*>(B.pos)[e] = make_uchar2((unsigned char)x, (unsigned char)y);
        if (!chg) {
            int32_t s[PCGRL_MAX_STATS], st[PCGRL_MAX_STATS];
            int32_t* inf = B.info + (size_t)e * 10;
            for (int k = 0; k < 8; k++) { s[k] = B.stats[(size_t)e * 8 + k]; st[k] = B.start_stats[(size_t)e * 8 + k]; inf[k] = s[k]; }
            const bool d = episode_over(P, s, st) || changes >= P.max_changes || iter >= P.max_iterations;
            B.reward[e] = 0.0;
            B.done[e] = d ? 1 : 0;
            if (P.prob == PCGRL_PROB_BINARY) inf[2] = s[1] - st[1];
            inf[8] = iter; inf[9] = changes;
            rst = d && P.auto_reset;
        }
    }
    block_append(chg, e, B, parity, WL_CHG, s_cnt[0], &s_base[0]);
    block_append(rst, e, B, parity, WL_RST, s_cnt[1], &s_base[1]);
}

// ------------------------------------------------------------------------------------------
// k_stats: lane group per work item
template <class MaskT>
__device__ __forceinline__ MaskT row_valid(int lane, int W, int H) {
    MaskT full = (W >= (int)(8 * sizeof(MaskT))) ? ~(MaskT)0 : (((MaskT)1 << W) - 1);
    return lane < H ? full : (MaskT)0;
}

__device__ __forceinline__ void finalize_item(const PcgrlParams& P, const DevBufs& B, int e, const int32_t* s,
                                              int mode, int parity, int shard) {
    int32_t* st = B.stats + (size_t)e * 8;
    int32_t* start = B.start_stats + (size_t)e * 8;
    if (mode == MODE_STEP) {
        int32_t old[PCGRL_MAX_STATS], sv[PCGRL_MAX_STATS];
        for (int k = 0; k < 8; k++) { old[k] = st[k]; sv[k] = start[k]; }
        const int2 c = reinterpret_cast<const int2*>(B.counters)[e];
        const double r = compute_reward(P, s, old);
        const bool d = episode_over(P, s, sv) || c.y >= P.max_changes || c.x >= P.max_iterations;
        B.reward[e] = r;
        B.done[e] = d ? 1 : 0;
        int32_t* inf = B.info + (size_t)e * 10;
        for (int k = 0; k < 8; k++) { st[k] = s[k]; inf[k] = s[k]; }
        if (P.prob == PCGRL_PROB_BINARY) inf[2] = s[1] - sv[1];      // path-imp (binary_prob.py:137)
        inf[8] = c.x; inf[9] = c.y;
        if (d && P.auto_reset) wl_push(B, parity, WL_RST, shard, e);
    } else {
        for (int k = 0; k < 8; k++) st[k] = s[k];
        if (mode == MODE_START)
            for (int k = 0; k < 8; k++) start[k] = s[k];
    }
}

// Problem.get_stats on the row masks of one map (b0..b2 = bit planes of the tile id).  Returns true
// when the Sokoban solver has to finish the job.
template <int PROB, class G, class MaskT>
__device__ __forceinline__ bool compute_item_stats(G& g, const PcgrlParams& P, MaskT b0, MaskT b1, MaskT b2, MaskT valid, int32_t* s) {
    if (PROB == PCGRL_PROB_BINARY) {
        int regions, path;
        regions_and_longest_path(g, (MaskT)(~b0 & valid), regions, path);
        s[0] = regions; s[1] = path;
        return false;
    }
    if (PROB == PCGRL_PROB_ZELDA) { zelda_stats(g, P, b0, b1, b2, valid, s); return false; }
    return sokoban_stats(g, P, b0, b1, b2, valid, s);
}

// Lane 0 of the group: hand the item to the solver or finish it.
__device__ __forceinline__ void finish_or_park(const PcgrlParams& P, const DevBufs& B, int e, const int32_t* s, bool need_solver,
                                               int mode, int parity, int shard) {
    if (need_solver) {
        // park the partial stats and hand the environment to the solver kernel.  STEP: in the info
        // row (the old stats are still needed for the reward); otherwise in the stats row itself
        // (the info row keeps the terminal info of an environment that is being reset).
        int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
        for (int k = 0; k < 8; k++) park[k] = s[k];
        wl_push(B, parity, mode == MODE_STEP ? WL_SOL : WL_SOL2, shard, e);
    } else {
        finalize_item(P, B, e, s, mode, parity, shard);
    }
}

template <int PROB, int G, class MaskT>
__global__ __launch_bounds__(PCGRL_BLOCK) void k_stats(PcgrlParams P, DevBufs B, int list, int parity, int mode, int clear_parity) {
    __shared__ int s_pref[WL_NSHARD + 1];
    // the last kernel of a step zeroes the *other* parity's work-list counters for the next step
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    DevGroup<G, MaskT> g;
    constexpr int GPB = PCGRL_BLOCK / G;
    const int n = wl_load_prefix(B, parity, list, s_pref);
    const int gi = threadIdx.x / G;
    const int NPL = (PROB == PCGRL_PROB_BINARY) ? 1 : 3;
    for (int item = blockIdx.x * GPB + gi; item < n; item += gridDim.x * GPB) {
        const int e = wl_get(B, list, s_pref, item);
        const int shard = (item >> 4) & (WL_NSHARD - 1);
        const MaskT* pl = reinterpret_cast<const MaskT*>(B.planes) + (size_t)e * NPL * G + g.lane;
        const MaskT valid = row_valid<MaskT>(g.lane, P.width, P.height);
        int32_t s[PCGRL_MAX_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
        const MaskT b0 = pl[0], b1 = NPL > 1 ? pl[G] : (MaskT)0, b2 = NPL > 1 ? pl[2 * G] : (MaskT)0;
        const bool need_solver = compute_item_stats<PROB>(g, P, b0, b1, b2, valid, s);
        if (g.lane == 0) finish_or_park(P, B, e, s, need_solver, mode, parity, shard);
    }
}

// ------------------------------------------------------------------------------------------
// k_sokoban: one wavefront per solver job (sokoban_solver.h).  Finishes what k_stats parked.
__global__ __launch_bounds__(64) void k_sokoban(PcgrlParams P, DevBufs B, int list, int parity, int mode, int clear_parity) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sok_lds[];
    __shared__ int s_pref[WL_NSHARD + 1];
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x;
    const int n = wl_load_prefix(B, parity, list, s_pref);
    __shared__ SokLevel s_L;             // level + node workspace in LDS: they are indexed dynamically
    __shared__ SokNode s_root, s_work;
    SokNode* pool = B.sok_pool + (size_t)blockIdx.x * B.sok_pool_stride;
    uint32_t* g_heap = B.sok_use_lds ? nullptr : B.sok_heap + (size_t)blockIdx.x * B.sok_heap_stride;
    uint32_t* g_table = B.sok_use_lds ? nullptr : B.sok_table + (size_t)blockIdx.x * B.sok_table_size;
    const int tsize = B.sok_use_lds ? SOK_LDS_TABLE : B.sok_table_size;
    for (int item = blockIdx.x; item < n; item += gridDim.x) {
        const int e = wl_get(B, list, s_pref, item);
        const int W = P.width, H = P.height;
        if (lane == 0) {
            const int ncr = sok_build_level(B.map + (size_t)e * W * H, W, H, s_L, s_root);
            if (ncr > SOK_MAXC) atomicOr(B.status, 1);
            sok_init_deadlocks(s_L);
            s_root.h = (uint16_t)sok_heuristic(s_L, s_root.crate);
        }
        int dist = 0, sol = 0;
        // The four agents of _run_game, with the exact exhausted-BFS shortcut (sokoban_solver.h).  The search is
        // driven by lane 0; every lane helps to clear the visited table between agents.
        const int KS[4] = {-1, 2, 1, 0};
        int go = 1;
        for (int a = 0; a < 4 && go; a++) {
            if (B.sok_use_lds) { for (int i = lane; i < tsize; i += 64) sok_lds[SOK_LDS_HEAP + i] = 0; }
            else { for (int i = lane; i < tsize; i += 64) g_table[i] = 0; }
            __threadfence_block();
            if (lane == 0) {
                int hh, dd, it;
                bool exhausted = false, win;
                if (B.sok_use_lds)   // two instantiations: LDS pointers compile to ds_* instructions
                    win = sok_search(s_L, pool, sok_lds, sok_lds + SOK_LDS_HEAP, tsize - 1, s_work, s_root, KS[a], P.solver_power, hh, dd, it, exhausted);
                else
                    win = sok_search(s_L, pool, g_heap, g_table, tsize - 1, s_work, s_root, KS[a], P.solver_power, hh, dd, it, exhausted);
                dist = win ? 0 : hh;
                sol = win ? dd : 0;
                go = !(win || (a == 0 && exhausted));
            }
            go = __shfl(go, 0, 64);
            __threadfence_block();
        }
        if (lane == 0) {
            int32_t s[PCGRL_MAX_STATS];
            const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
            for (int k = 0; k < 8; k++) s[k] = park[k];
            s[4] = dist; s[5] = sol;
            finalize_item(P, B, e, s, mode, parity, item & (WL_NSHARD - 1));
        }
    }
}

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

// ------------------------------------------------------------------------------------------
// Observation formatting of the reference's composite wrappers (wrappers.py): Cropped.transform :197-206
// (pad with the border tile, window of `size` centred on the cursor), OneHotEncoding.transform :101-104,
// ToImage.transform :53-60 ([h, w, depth] image).  One thread per output cell; depth 1 = raw tile ids,
// depth T = one-hot.  Writes are coalesced along (cell, depth).
__global__ __launch_bounds__(PCGRL_BLOCK) void k_obs_window(PcgrlParams P, DevBufs B, uint8_t* __restrict__ out, int oh, int ow,
                                                             int centered, int pad_value, int depth) {
    const size_t per_env = (size_t)oh * ow;
    const size_t total = (size_t)P.num_envs * per_env;
    for (size_t i = (size_t)blockIdx.x * PCGRL_BLOCK + threadIdx.x; i < total; i += (size_t)gridDim.x * PCGRL_BLOCK) {
        const int e = (int)(i / per_env);
        const int rc = (int)(i - (size_t)e * per_env);
        const int r = rc / ow, c = rc - r * ow;
        int y = r, x = c;
        if (centered) {
            const uchar2 p = reinterpret_cast<const uchar2*>(B.pos)[e];
            y = (int)p.y + r - oh / 2;     // np.pad(map, size // 2) then padded[y : y + size, x : x + size]
            x = (int)p.x + c - ow / 2;
        }
        int t = pad_value;
        if (x >= 0 && y >= 0 && x < P.width && y < P.height) t = B.map[((size_t)e * P.height + y) * P.width + x];
        uint8_t* o = out + i * depth;
        if (depth == 1) o[0] = (uint8_t)t;
        else if (depth == 8) {
            *reinterpret_cast<uint64_t*>(o) = 1ull << (8 * t);
        } else {
            for (int d = 0; d < depth; d++) o[d] = (uint8_t)(d == t);
        }
    }
}
// ActionMap.step for the wide representation (wrappers.py:139-154): flat index into (h, w, tiles) -> (x, y, tile)
__global__ void k_action_map(const int32_t* __restrict__ flat, int32_t* __restrict__ xyv, int n, int w, int h, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int a = flat[i];
        a = a < 0 ? 0 : (a >= w * h * dim ? w * h * dim - 1 : a);
        const int v = a % dim, x = (a / dim) % w, y = a / (dim * w);
        xyv[3 * i] = x; xyv[3 * i + 1] = y; xyv[3 * i + 2] = v;
    }
}

__global__ void k_fill_all(DevBufs B, int n, int parity, int list) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) B.wl_items[list][(size_t)(i & (WL_NSHARD - 1)) * B.wl_cap[list] + (i >> 6)] = i;
    if (i < WL_NSHARD) wl_counters(B, parity, list)[i * WL_CSTRIDE] = (n - i + WL_NSHARD - 1) / WL_NSHARD;
}
__global__ void k_bcast_tile_p(double* tile_p, int n, double p0, double p1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { tile_p[2 * i] = p0; tile_p[2 * i + 1] = p1; }
}
__global__ void k_zero_cursors(int32_t* cur, int first, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { cur[2 * (first + i)] = 0; cur[2 * (first + i) + 1] = 0; }
}

// ------------------------------------------------------------------------------------------
// Host side of the ABI
#define PCGRL_MAX_SUB 4
struct pcgrl_env {
    pcgrl_config cfg;
    PcgrlParams P;
    pcgrl_layout L;
    DevBufs B;
    int bound;
    int has_old;      // at least one random reset happened (representation.py:41)
    int was_reset;
    int parity;
    int device;
    // optional per-phase timing with HIP events on the caller's stream (pcgrl_profile)
    int alloc_solver_power;
    // Sub-batches: the environment axis is cut into nsub contiguous slices whose kernel chains run on
    // separate HIP streams (slice 0 on the caller's stream), so that one slice's latency-bound kernels
    // (k_update, k_reset) overlap another slice's throughput-bound k_stats.  Fork/join with events.
    int nsub;
    PcgrlParams subP[PCGRL_MAX_SUB];
    DevBufs subB[PCGRL_MAX_SUB];
    hipStream_t sub_stream[PCGRL_MAX_SUB];
    hipEvent_t ev_fork, ev_join[PCGRL_MAX_SUB];
    int profiling;
    std::vector<hipEvent_t> events;
    size_t ev_used;
    int prof_steps;
};
#define PCGRL_NPHASE 6   /* update, stats(step), solver(step), mapgen, stats(start), solver(start) */
static int prof_mark(pcgrl_env* h, hipStream_t st) {
    if (!h->profiling) return PCGRL_OK;
    if (h->ev_used == h->events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return PCGRL_EHIP;
        h->events.push_back(e);
    }
    if (hipEventRecord(h->events[h->ev_used++], st) != hipSuccess) return PCGRL_EHIP;
    return PCGRL_OK;
}

static thread_local int g_last_hip = 0;
#define HIPCHK(expr) do { hipError_t err_ = (expr); if (err_ != hipSuccess) { g_last_hip = (int)err_; return PCGRL_EHIP; } } while (0)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int validate_config(const pcgrl_config* c) {
    if (!c) return PCGRL_EINVAL;
    if (c->prob < 0 || c->prob > 2 || c->rep < 0 || c->rep > 5) return PCGRL_EINVAL;
    if (c->num_envs < 1) return PCGRL_EINVAL;
    if (c->width < 1 || c->width > 64 || c->height < 1 || c->height > 64) return PCGRL_EINVAL;
    if (c->max_changes < 1 || c->max_iterations < 1) return PCGRL_EINVAL;
    if (c->prob == PCGRL_SOKOBAN) {   // limits of the solver kernel (sokoban_solver.h)
        if ((c->width + 2) * (c->height + 2) > 256) return PCGRL_EINVAL;
        if (c->solver_power < 1 || c->solver_power > 16383) return PCGRL_EINVAL;
    }
    return PCGRL_OK;
}
static int ntiles_of(int prob) { return prob == PCGRL_BINARY ? 2 : (prob == PCGRL_ZELDA ? 8 : 5); }

static void fill_params(const pcgrl_config* c, PcgrlParams* P) {
    memset(P, 0, sizeof(*P));
    P->prob = c->prob; P->rep = c->rep; P->num_envs = c->num_envs;
    P->width = c->width; P->height = c->height;
    P->prob_width = c->width; P->prob_height = c->height;
    P->ntiles = ntiles_of(c->prob);
    P->nplanes = c->prob == PCGRL_BINARY ? 1 : 3;
    P->group = c->height <= 16 ? 16 : 64;
    P->mask_bytes = c->width <= 32 ? 4 : 8;
    P->max_changes = c->max_changes; P->max_iterations = c->max_iterations;
    P->random_start = c->random_start; P->random_tile = c->random_tile; P->warp = c->warp;
    P->random_probs = c->random_probs; P->auto_reset = c->auto_reset;
    P->target_path = c->target_path; P->max_enemies = c->max_enemies; P->target_enemy_dist = c->target_enemy_dist;
    P->max_crates = c->max_crates; P->target_solution = c->target_solution; P->solver_power = c->solver_power;
    for (int i = 0; i < 8; i++) P->rewards[i] = c->rewards[i];
    pcgrl_build_cdf(c->tile_probs, P->ntiles, P->cdf);
}

static const size_t WL_CNT_BYTES = 2 * WL_NLIST * WL_NSHARD * WL_CSTRIDE * sizeof(int32_t);
// shard capacity: the changed list is bucketed by difficulty, so one bucket may receive every environment
static int wl_capacity(int num_envs, int list) {
    return list == WL_CHG ? num_envs : 2 * ((num_envs + WL_NSHARD - 1) / WL_NSHARD + 256);
}
static size_t wl_list_bytes(int num_envs, int list) { return align_up((size_t)WL_NSHARD * wl_capacity(num_envs, list) * 4, 256); }
#define SOK_BLOCKS 256   /* resident solver blocks (one per CU: heap + table fill most of its LDS) */
static size_t wl_bytes(const pcgrl_config* c) {
    size_t b = WL_CNT_BYTES + 256;
    for (int k = 0; k < WL_NLIST; k++) b += wl_list_bytes(c->num_envs, k);
    return b;
}
static int sok_table_size(int power) { int t = 1024; while (t < 2 * power) t <<= 1; return t; }
static int num_subbatches(const pcgrl_config* c) {
    const char* e = getenv("PCGRL_SUBBATCHES");
    // Measured on MI355X (C2/C3, 65 536 envs): 1 slice 61 us/step, 2 slices 74, 4 slices 113 -- the
    // cross-stream event fork/join costs more than the overlap wins, so slicing is opt-in only.
    int n = e ? atoi(e) : 1;
    if (c->prob == PCGRL_SOKOBAN) n = 1;          // the solver arena is per handle
    if (n < 1) n = 1;
    if (n > PCGRL_MAX_SUB) n = PCGRL_MAX_SUB;
    while (n > 1 && c->num_envs / n < 1024) n--;
    return n;
}
static void sub_range(int num_envs, int nsub, int k, int* lo, int* hi) {
    const int base = num_envs / nsub, extra = num_envs % nsub;
    *lo = k * base + (k < extra ? k : extra);
    *hi = *lo + base + (k < extra ? 1 : 0);
}
static size_t scratch_bytes(const pcgrl_config* c) {
    const int nsub = num_subbatches(c);
    pcgrl_config cc = *c;
    cc.num_envs = (c->num_envs + nsub - 1) / nsub;
    size_t b = nsub * wl_bytes(&cc);
    if (c->prob == PCGRL_SOKOBAN) {
        const size_t nodes = 4 * (size_t)c->solver_power + 4;
        b += SOK_BLOCKS * align_up(nodes * sizeof(SokNode), 256);
        if (c->solver_power > SOK_LDS_POWER) b += SOK_BLOCKS * (align_up(nodes * 4, 256) + (size_t)sok_table_size(c->solver_power) * 4);
    }
    return b;
}

extern "C" {

int pcgrl_abi_version(void) { return PCGRL_ABI_VERSION; }
int pcgrl_last_hip_error(void) { return g_last_hip; }
const char* pcgrl_error_string(int code) {
    switch (code) {
        case PCGRL_OK: return "ok";
        case PCGRL_EINVAL: return "invalid argument or unsupported configuration";
        case PCGRL_EHIP: return "HIP runtime error";
        case PCGRL_ESTATE: return "call order violated (bind, seed and reset before step)";
        default: return "unknown error";
    }
}

int pcgrl_query_layout(const pcgrl_config* c, pcgrl_layout* L) {
    int rc = validate_config(c);
    if (rc) return rc;
    if (!L) return PCGRL_EINVAL;
    PcgrlParams P;
    fill_params(c, &P);
    const size_t n = (size_t)c->num_envs, cells = (size_t)c->width * c->height;
    memset(L, 0, sizeof(*L));
    L->group = P.group; L->mask_bytes = P.mask_bytes; L->nplanes = P.nplanes; L->nstats = num_stats(c->prob);
    L->map = n * cells; L->old_map = n * cells; L->heatmap = n * cells * 2; L->pos = n * 2;
    L->planes = n * P.nplanes * P.group * P.mask_bytes;
    L->counters = n * 8; L->stats = n * 32; L->start_stats = n * 32; L->info = n * 40;
    L->reward = n * 8; L->done = n; L->tile_p = n * 16;
    L->rng_rep = n * PCGRL_MT_N * 4; L->rng_prob = c->prob == PCGRL_BINARY ? n * PCGRL_MT_N * 4 : 0;
    L->rng_cursor = n * 8;
    L->scratch = scratch_bytes(c);
    return PCGRL_OK;
}

int pcgrl_create(const pcgrl_config* c, pcgrl_env** out) {
    int rc = validate_config(c);
    if (rc) return rc;
    if (!out) return PCGRL_EINVAL;
    pcgrl_env* h = new pcgrl_env();
    h->bound = h->has_old = h->was_reset = h->parity = h->device = 0;
    h->profiling = 0; h->ev_used = 0; h->prof_steps = 0;
    h->nsub = 1; h->ev_fork = nullptr;
    for (int k = 0; k < PCGRL_MAX_SUB; k++) { h->sub_stream[k] = nullptr; h->ev_join[k] = nullptr; }
    memset(&h->B, 0, sizeof(h->B));
    h->cfg = *c;
    fill_params(c, &h->P);
    pcgrl_query_layout(c, &h->L);
    *out = h;
    return PCGRL_OK;
}

int pcgrl_destroy(pcgrl_env* h) {
    if (h) {
        for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
        for (int k = 0; k < PCGRL_MAX_SUB; k++) {
            if (h->sub_stream[k]) { (void)hipStreamSynchronize(h->sub_stream[k]); (void)hipStreamDestroy(h->sub_stream[k]); }
            if (h->ev_join[k]) (void)hipEventDestroy(h->ev_join[k]);
        }
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    }
    delete h;
    return PCGRL_OK;
}

int pcgrl_bind(pcgrl_env* h, const pcgrl_buffers* b, void* stream) {
    if (!h || !b) return PCGRL_EINVAL;
    if (!b->map || !b->old_map || !b->heatmap || !b->pos || !b->planes || !b->counters || !b->stats ||
        !b->start_stats || !b->info || !b->reward || !b->done || !b->tile_p || !b->rng_rep || !b->rng_cursor || !b->scratch)
        return PCGRL_EINVAL;
    if (h->cfg.prob == PCGRL_BINARY && !b->rng_prob) return PCGRL_EINVAL;
    DevBufs& B = h->B;
    B.map = (uint8_t*)b->map; B.old_map = (uint8_t*)b->old_map; B.heat = (uint16_t*)b->heatmap; B.pos = (uint8_t*)b->pos;
    B.planes = b->planes; B.counters = (int32_t*)b->counters; B.stats = (int32_t*)b->stats;
    B.start_stats = (int32_t*)b->start_stats; B.info = (int32_t*)b->info; B.reward = (double*)b->reward;
    B.done = (uint8_t*)b->done; B.tile_p = (double*)b->tile_p; B.rng_rep = (uint32_t*)b->rng_rep;
    B.rng_prob = (uint32_t*)b->rng_prob; B.rng_cur = (int32_t*)b->rng_cursor;
    uint8_t* s = (uint8_t*)b->scratch;
    B.wl_cnt = (int32_t*)s;
    B.status = (int32_t*)(s + WL_CNT_BYTES);
    {
        uint8_t* q = s + WL_CNT_BYTES + 256;
        for (int k = 0; k < WL_NLIST; k++) {
            B.wl_cap[k] = wl_capacity(h->cfg.num_envs, k);
            B.wl_items[k] = (int32_t*)q;
            q += wl_list_bytes(h->cfg.num_envs, k);
        }
    }
    HIPCHK(hipMemsetAsync(B.wl_cnt, 0, WL_CNT_BYTES + 256, (hipStream_t)stream));
    B.sok_pool = nullptr; B.sok_heap = nullptr; B.sok_table = nullptr;
    if (h->cfg.prob == PCGRL_SOKOBAN) {
        // the arena is sized for the solver_power the buffers were allocated with
        const int power = h->alloc_solver_power = h->cfg.solver_power;
        const size_t nodes = 4 * (size_t)power + 4;
        uint8_t* a = s + wl_bytes(&h->cfg);   // (nsub == 1 for sokoban)
        B.sok_pool = (SokNode*)a;
        B.sok_pool_stride = (int32_t)(align_up(nodes * sizeof(SokNode), 256) / sizeof(SokNode));
        a += SOK_BLOCKS * align_up(nodes * sizeof(SokNode), 256);
        B.sok_use_lds = power <= SOK_LDS_POWER;
        B.sok_table_size = sok_table_size(power);
        B.sok_heap_stride = (int32_t)(align_up(nodes * 4, 256) / 4);
        if (!B.sok_use_lds) {
            B.sok_heap = (uint32_t*)a;
            a += SOK_BLOCKS * align_up(nodes * 4, 256);
            B.sok_table = (uint32_t*)a;
        }
    }
    // sub-batch views: every per-environment pointer offset to the slice, private work lists
    h->nsub = num_subbatches(&h->cfg);
    if (h->nsub > 1) {
        const PcgrlParams& P0 = h->P;
        const size_t cells = (size_t)P0.width * P0.height;
        pcgrl_config cc = h->cfg;
        cc.num_envs = (h->cfg.num_envs + h->nsub - 1) / h->nsub;
        const size_t wlb = wl_bytes(&cc);
        for (int k = 0; k < h->nsub; k++) {
            int lo, hi;
            sub_range(h->cfg.num_envs, h->nsub, k, &lo, &hi);
            PcgrlParams& P = h->subP[k]; DevBufs& S = h->subB[k];
            P = P0; P.num_envs = hi - lo;
            S = B;
            S.map = B.map + lo * cells; S.old_map = B.old_map + lo * cells; S.heat = B.heat + lo * cells; S.pos = B.pos + 2 * (size_t)lo;
            S.planes = (uint8_t*)B.planes + (size_t)lo * P0.nplanes * P0.group * P0.mask_bytes;
            S.counters = B.counters + 2 * (size_t)lo; S.stats = B.stats + 8 * (size_t)lo; S.start_stats = B.start_stats + 8 * (size_t)lo;
            S.info = B.info + 10 * (size_t)lo; S.reward = B.reward + lo; S.done = B.done + lo; S.tile_p = B.tile_p + 2 * (size_t)lo;
            S.rng_rep = B.rng_rep + (size_t)lo * PCGRL_MT_N; S.rng_prob = B.rng_prob ? B.rng_prob + (size_t)lo * PCGRL_MT_N : nullptr;
            S.rng_cur = B.rng_cur + 2 * (size_t)lo;
            uint8_t* q = s + (size_t)k * wlb;
            S.wl_cnt = (int32_t*)q;
            q += WL_CNT_BYTES + 256;
            for (int l = 0; l < WL_NLIST; l++) { S.wl_cap[l] = wl_capacity(cc.num_envs, l); S.wl_items[l] = (int32_t*)q; q += wl_list_bytes(cc.num_envs, l); }
            if (k > 0 && !h->sub_stream[k]) {
                HIPCHK(hipStreamCreateWithFlags(&h->sub_stream[k], hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
            }
        }
        if (!h->ev_fork) HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCHK(hipMemsetAsync(s, 0, h->nsub * wlb, (hipStream_t)stream));
    }
    h->bound = 1; h->has_old = 0; h->was_reset = 0; h->parity = 0;
    return PCGRL_OK;   // tile_p is caller state: call pcgrl_set_tile_probs once after the first bind
}

int pcgrl_configure(pcgrl_env* h, const pcgrl_config* c) {
    if (!h) return PCGRL_EINVAL;
    int rc = validate_config(c);
    if (rc) return rc;
    if (c->prob != h->cfg.prob || c->rep != h->cfg.rep || c->num_envs != h->cfg.num_envs ||
        c->width != h->cfg.width || c->height != h->cfg.height)
        return PCGRL_EINVAL;
    if (h->bound && c->prob == PCGRL_SOKOBAN && c->solver_power > h->alloc_solver_power) return PCGRL_EINVAL;   // arena too small: re-create
    h->cfg = *c;
    fill_params(c, &h->P);
    for (int k = 0; k < h->nsub && h->nsub > 1; k++) { const int n = h->subP[k].num_envs; h->subP[k] = h->P; h->subP[k].num_envs = n; }
    return PCGRL_OK;
}

int pcgrl_set_tile_probs(pcgrl_env* h, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    const int n = h->cfg.num_envs;
    hipLaunchKernelGGL(k_bcast_tile_p, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->B.tile_p, n,
                       h->cfg.tile_probs[0], h->cfg.tile_probs[1]);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}

int pcgrl_seed(pcgrl_env* h, const uint32_t* keys, int32_t first, int32_t count, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    if (!keys || first < 0 || count < 1 || first + count > h->cfg.num_envs) return PCGRL_EINVAL;
    const size_t bytes = (size_t)count * PCGRL_MT_N * 4, off = (size_t)first * PCGRL_MT_N;
    HIPCHK(hipMemcpyAsync(h->B.rng_rep + off, keys, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    if (h->B.rng_prob) HIPCHK(hipMemcpyAsync(h->B.rng_prob + off, keys, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    hipLaunchKernelGGL(k_zero_cursors, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->B.rng_cur, first, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));   // `keys` may be pageable host memory
    return PCGRL_OK;
}

}  // extern "C"

// ---- launch helpers ------------------------------------------------------------------------
static int grid_for(int items, int per_block, int cap) {
    int g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return g < cap ? g : cap;
}

template <int PROB>
static int launch_stats_p(pcgrl_env* h, int list, int parity, int mode, int clr, hipStream_t st) {
    const PcgrlParams& P = h->P;
    const int gpb = PCGRL_BLOCK / P.group;
    // per-step reset lists are short: a small grid-stride grid avoids dispatching thousands of empty blocks
    const int grid = grid_for(P.num_envs, gpb, (mode == MODE_START && h->was_reset) ? 512 : 8192);
    if (P.group == 16 && P.mask_bytes == 4)
        hipLaunchKernelGGL((k_stats<PROB, 16, uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, list, parity, mode, clr);
    else if (P.group == 16)
        hipLaunchKernelGGL((k_stats<PROB, 16, uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, list, parity, mode, clr);
    else if (P.mask_bytes == 4)
        hipLaunchKernelGGL((k_stats<PROB, 64, uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, list, parity, mode, clr);
    else
        hipLaunchKernelGGL((k_stats<PROB, 64, uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, list, parity, mode, clr);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
static int launch_stats(pcgrl_env* h, int list, int parity, int mode, int clr, hipStream_t st) {
    switch (h->P.prob) {
        case PCGRL_PROB_BINARY: return launch_stats_p<PCGRL_PROB_BINARY>(h, list, parity, mode, clr, st);
        case PCGRL_PROB_ZELDA: return launch_stats_p<PCGRL_PROB_ZELDA>(h, list, parity, mode, clr, st);
        default: return launch_stats_p<PCGRL_PROB_SOKOBAN>(h, list, parity, mode, clr, st);
    }
}

template <class MaskT>
static int launch_update_m(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st) {
    const PcgrlParams& P = h->P;
    const int grid = (P.num_envs + PCGRL_BLOCK - 1) / PCGRL_BLOCK;
    switch (P.rep) {
        case PCGRL_REP_NARROW:
            hipLaunchKernelGGL((k_update<PCGRL_REP_NARROW, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        case PCGRL_REP_WIDE:
            hipLaunchKernelGGL((k_update<PCGRL_REP_WIDE, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        case PCGRL_REP_TURTLE:
            hipLaunchKernelGGL((k_update<PCGRL_REP_TURTLE, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        case PCGRL_REP_NARROW_CAST:
            hipLaunchKernelGGL((k_update_block<PCGRL_REP_NARROW_CAST, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        case PCGRL_REP_NARROW_MULTI:
            hipLaunchKernelGGL((k_update_block<PCGRL_REP_NARROW_MULTI, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        default:
            hipLaunchKernelGGL((k_update_block<PCGRL_REP_TURTLE_CAST, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
    }
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}

static int launch_solver(pcgrl_env* h, int list, int parity, int mode, int clr, hipStream_t st) {
    const size_t lds = h->B.sok_use_lds ? (size_t)(SOK_LDS_HEAP + SOK_LDS_TABLE) * 4 : 0;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sokoban), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((SOK_LDS_HEAP + SOK_LDS_TABLE) * 4)));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_sokoban, dim3(SOK_BLOCKS), dim3(64), lds, st, h->P, h->B, list, parity, mode, clr);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}

template <int PROB>
static int launch_reset_p(pcgrl_env* h, int parity, int clr, hipStream_t st) {
    const PcgrlParams& P = h->P;
    const int cells = P.width * P.height;
    const size_t lds = 4 * (size_t)(PCGRL_MT_N * 4 + ((cells + 15) & ~15));
    const int grid = grid_for(P.num_envs, 4, h->was_reset ? 512 : 4096);
    const int gen = (P.random_start || !h->has_old) ? 1 : 0;
    if (P.group == 16 && P.mask_bytes == 4)
        hipLaunchKernelGGL((k_reset<PROB, 16, uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, parity, gen, clr);
    else if (P.group == 16)
        hipLaunchKernelGGL((k_reset<PROB, 16, uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, parity, gen, clr);
    else if (P.mask_bytes == 4)
        hipLaunchKernelGGL((k_reset<PROB, 64, uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, parity, gen, clr);
    else
        hipLaunchKernelGGL((k_reset<PROB, 64, uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, parity, gen, clr);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
// map generation + start stats of every environment on the reset list
static int launch_reset(pcgrl_env* h, int parity, int clr, hipStream_t st) {
    switch (h->P.prob) {
        case PCGRL_PROB_BINARY: return launch_reset_p<PCGRL_PROB_BINARY>(h, parity, clr, st);
        case PCGRL_PROB_ZELDA: return launch_reset_p<PCGRL_PROB_ZELDA>(h, parity, clr, st);
        default: return launch_reset_p<PCGRL_PROB_SOKOBAN>(h, parity, clr, st);
    }
}

extern "C" {

static int reset_one(pcgrl_env* h, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int n = h->P.num_envs, par = h->parity;   // (h->P is the sub-batch view here)
    hipLaunchKernelGGL(k_fill_all, dim3((n + 255) / 256), dim3(256), 0, st, h->B, n, par, (int)WL_RST);
    HIPCHK(hipGetLastError());
    const bool sok = h->P.prob == PCGRL_PROB_SOKOBAN;
    int rc = launch_reset(h, par, sok ? -1 : (par ^ 1), st);
    if (rc) return rc;
    if (sok && (rc = launch_solver(h, WL_SOL2, par, MODE_START, par ^ 1, st))) return rc;
    return PCGRL_OK;
}

static int step_one(pcgrl_env* h, const int32_t* actions, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int par = h->parity;
    int rc;
    if ((rc = prof_mark(h, st))) return rc;
    rc = (h->P.mask_bytes == 4) ? launch_update_m<uint32_t>(h, actions, par, st) : launch_update_m<uint64_t>(h, actions, par, st);
    if (rc) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    // the last kernel of the step clears the other parity's work-list counters
    const bool sok = h->P.prob == PCGRL_PROB_SOKOBAN, ar = h->P.auto_reset != 0;
    rc = launch_stats(h, WL_CHG, par, MODE_STEP, (ar || sok) ? -1 : (par ^ 1), st);
    if (rc) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    if (sok && (rc = launch_solver(h, WL_SOL, par, MODE_STEP, ar ? -1 : (par ^ 1), st))) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    if (ar) {
        rc = launch_reset(h, par, sok ? -1 : (par ^ 1), st);
        if (rc) return rc;
    }
    if ((rc = prof_mark(h, st))) return rc;
    if ((rc = prof_mark(h, st))) return rc;   // (the start stats are part of k_reset)
    if (ar && sok && (rc = launch_solver(h, WL_SOL2, par, MODE_START, par ^ 1, st))) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    return PCGRL_OK;
}

// Run `fn` once per sub-batch: slice 0 on the caller's stream, the others on their own streams between a
// fork event and join events; h->P / h->B are swapped to the slice's view for the duration of the call.
static int for_each_sub(pcgrl_env* h, hipStream_t caller, const std::function<int(int, hipStream_t)>& fn) {
    if (h->nsub <= 1) return fn(0, caller);
    const PcgrlParams P0 = h->P;
    const DevBufs B0 = h->B;
    const int prof = h->profiling;
    int rc = PCGRL_OK;
    if (hipEventRecord(h->ev_fork, caller) != hipSuccess) return PCGRL_EHIP;
    for (int k = h->nsub - 1; k >= 0 && rc == PCGRL_OK; k--) {   // side streams first, the caller's slice last
        hipStream_t st = k == 0 ? caller : h->sub_stream[k];
        if (k > 0 && hipStreamWaitEvent(st, h->ev_fork, 0) != hipSuccess) { rc = PCGRL_EHIP; break; }
        h->P = h->subP[k]; h->B = h->subB[k];
        h->profiling = (k == 0) ? prof : 0;          // phase timing follows slice 0
        rc = fn(k, st);
        if (rc == PCGRL_OK && k > 0 && hipEventRecord(h->ev_join[k], st) != hipSuccess) rc = PCGRL_EHIP;
    }
    h->P = P0; h->B = B0; h->profiling = prof;
    for (int k = 1; k < h->nsub && rc == PCGRL_OK; k++)
        if (hipStreamWaitEvent(caller, h->ev_join[k], 0) != hipSuccess) rc = PCGRL_EHIP;
    return rc;
}

int pcgrl_reset(pcgrl_env* h, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    int rc = for_each_sub(h, (hipStream_t)stream, [&](int, hipStream_t st) { return reset_one(h, st); });
    if (rc) return rc;
    h->parity ^= 1;
    h->has_old = 1;
    h->was_reset = 1;
    return PCGRL_OK;
}

int pcgrl_step(pcgrl_env* h, const int32_t* actions, void* stream) {
    if (!h || !h->bound || !h->was_reset) return PCGRL_ESTATE;
    if (!actions) return PCGRL_EINVAL;
    static const int kActionWidth[6] = {1, 3, 1, 2, 9, 2};
    const int aw = kActionWidth[h->P.rep];
    int rc = for_each_sub(h, (hipStream_t)stream, [&](int k, hipStream_t st) {
        int lo = 0, hi = 0;
        sub_range(h->cfg.num_envs, h->nsub, k, &lo, &hi);
        return step_one(h, actions + (size_t)lo * aw, st);
    });
    if (rc) return rc;
    h->parity ^= 1;
    if (h->profiling) h->prof_steps++;
    return PCGRL_OK;
}

int pcgrl_profile(pcgrl_env* h, int enable) {
    if (!h) return PCGRL_EINVAL;
    h->profiling = enable ? 1 : 0;
    h->ev_used = 0;
    h->prof_steps = 0;
    return PCGRL_OK;
}

// Sums the per-phase GPU time (ms) of every step issued since pcgrl_profile(h, 1); synchronises.
int pcgrl_profile_read(pcgrl_env* h, double* phase_ms, int32_t* steps) {
    if (!h || !phase_ms || !steps) return PCGRL_EINVAL;
    for (int k = 0; k < PCGRL_NPHASE; k++) phase_ms[k] = 0.0;
    *steps = h->prof_steps;
    if (h->ev_used == 0) return PCGRL_OK;
    HIPCHK(hipEventSynchronize(h->events[h->ev_used - 1]));
    const size_t per = PCGRL_NPHASE + 1;
    for (size_t s0 = 0; s0 + per <= h->ev_used; s0 += per)
        for (int k = 0; k < PCGRL_NPHASE; k++) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h->events[s0 + k], h->events[s0 + k + 1]));
            phase_ms[k] += ms;
        }
    return PCGRL_OK;
}

int pcgrl_observe(pcgrl_env* h, uint8_t* out, int32_t out_h, int32_t out_w, int32_t centered, int32_t pad_value, int32_t onehot, void* stream) {
    if (!h || !h->bound || !h->was_reset) return PCGRL_ESTATE;
    if (!out || out_h < 1 || out_w < 1) return PCGRL_EINVAL;
    if (centered && h->P.rep == PCGRL_REP_WIDE) return PCGRL_EINVAL;   // Cropped needs a cursor (wrappers.py:170)
    const int depth = onehot ? h->P.ntiles : 1;
    const size_t total = (size_t)h->P.num_envs * out_h * out_w;
    const int grid = (int)((total + PCGRL_BLOCK - 1) / PCGRL_BLOCK < 16384 ? (total + PCGRL_BLOCK - 1) / PCGRL_BLOCK : 16384);
    hipLaunchKernelGGL(k_obs_window, dim3(grid), dim3(PCGRL_BLOCK), 0, (hipStream_t)stream, h->P, h->B, out, out_h, out_w, centered, pad_value, depth);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}

int pcgrl_action_map(pcgrl_env* h, const int32_t* flat, int32_t* xyv, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    if (!flat || !xyv) return PCGRL_EINVAL;
    const int n = h->P.num_envs;
    hipLaunchKernelGGL(k_action_map, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, flat, xyv, n, h->P.width, h->P.height, h->P.ntiles);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}

int pcgrl_status(pcgrl_env* h, void* stream, int32_t* status) {
    if (!h || !h->bound || !status) return PCGRL_ESTATE;
    HIPCHK(hipMemcpyAsync(status, h->B.status, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return PCGRL_OK;
}

static int set_maps_one(pcgrl_env* h, const uint8_t* maps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const PcgrlParams& P = h->P;
    const int n = P.num_envs, par = h->parity, cells = P.width * P.height;
    const size_t lds = 4 * (size_t)((cells + 15) & ~15);
    const int grid = grid_for(n, 4, 4096);
    if (P.mask_bytes == 4)
        hipLaunchKernelGGL((k_planes_from_map<uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, maps);
    else
        hipLaunchKernelGGL((k_planes_from_map<uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, maps);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_fill_all, dim3((n + 255) / 256), dim3(256), 0, st, h->B, n, par, (int)WL_CHG);
    HIPCHK(hipGetLastError());
    const bool sok = P.prob == PCGRL_PROB_SOKOBAN;
    int rc = launch_stats(h, WL_CHG, par, MODE_SETMAP, sok ? -1 : (par ^ 1), st);
    if (rc) return rc;
    if (sok && (rc = launch_solver(h, WL_SOL2, par, MODE_SETMAP, par ^ 1, st))) return rc;
    return PCGRL_OK;
}

int pcgrl_set_maps(pcgrl_env* h, const uint8_t* maps, void* stream) {
    if (!h || !h->bound || !h->was_reset) return PCGRL_ESTATE;
    if (!maps) return PCGRL_EINVAL;
    const size_t cells = (size_t)h->P.width * h->P.height;
    int rc = for_each_sub(h, (hipStream_t)stream, [&](int k, hipStream_t st) {
        int lo = 0, hi = 0;
        sub_range(h->cfg.num_envs, h->nsub, k, &lo, &hi);
        return set_maps_one(h, maps + (size_t)lo * cells, st);
    });
    if (rc) return rc;
    h->parity ^= 1;
    return PCGRL_OK;
}

}  // extern "C"
