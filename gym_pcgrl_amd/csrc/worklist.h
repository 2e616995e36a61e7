// Device buffers, sharded work lists and block-wide compaction shared by all kernels.
// Part of the single translation unit pcgrl_abi.hip (see its header comment for the overall picture).
#pragma once
#define PCGRL_BLOCK 256
// bits of the sticky status word (pcgrl_status; include/pcgrl_hip.h)
#define PCGRL_STATUS_TOO_MANY_CRATES 1
#define PCGRL_STATUS_BAD_ACTION 2
#define PCGRL_STATUS_BAD_TILE 4
enum { MODE_STEP = 0, MODE_START = 1, MODE_SETMAP = 2 };

// Work lists (changed environments, environments to reset, sokoban solver jobs).  A list is 64 shards,
// each with its own counter on its own 64-byte line and its own segment of the item array, so appends
// never pile up on one address (thousands of same-address atomics per step cost ~12 ns each);
// consumers rebuild the dense index with a 64-entry prefix sum in LDS.  Counters are double-buffered
// by step parity: the last kernel of a step zeroes the other parity's counters.
#define WL_NSHARD 64
#define WL_WIDE_FEW_REGIONS 32     /* tall binary maps: at most that many regions = "one sweep and little else" (difficulty_bucket) */
#define WL_CSTRIDE 16
// CHG: changed environments; RST: to reset; SOL: solver jobs of the step; SOL2: of the resets.  Sokoban only: RST2 =
// environments whose episode the solver kernel ended, SOL3 = solver jobs of *their* resets.
// INC (binary, 16-row maps): changed environments whose statistics can be updated incrementally (binary_incremental).
enum { WL_CHG = 0, WL_RST = 1, WL_SOL = 2, WL_SOL2 = 3, WL_RST2 = 4, WL_SOL3 = 5, WL_INC = 6, WL_NLIST = 7 };
// smb: a level whose last play-through took at least this many pops goes on WL_INC, the list k_smb starts with (kernels_smb.h)
#ifndef SMB_LONG_POPS
#define SMB_LONG_POPS 2000
#endif
// An INC item (and, for zelda, a CHG item too): environment in bits 0..20, changed cell (row * 32 + column) in bits 21..29,
// bits 30..31 = what happened to the cell's passability (binary: 1 = became passable, 0 = impassable; zelda: 0 = unchanged,
// 1 = became passable, 2 = impassable).
#define WL_INC_ENV_MASK 0x1FFFFF
// 64-lane groups (maps taller than 16 rows, binary only): environment in bits 0..18, cell (row * 64 + column) in bits 19..30,
// bit 31 = the cell became passable.
#define WL_INC64_ENV_MASK 0x7FFFF
__device__ __forceinline__ int wl_inc_pack(int group, int e, int row, int col, unsigned code) {
    return group == 16 ? (int)((unsigned)e | ((unsigned)(row * 32 + col) << 21) | (code << 30))
                       : (int)((unsigned)e | ((unsigned)(row * 64 + col) << 19) | ((code & 1u) << 31));
}
// maps beyond 64 x 64 (k_big, binary): environment in bits 0..14, column in 15..22, row in 23..30, bit 31 = the cell became passable
#define WL_INCBIG_ENV_MASK 0x7FFF
__device__ __forceinline__ int wl_incbig_pack(int e, int x, int y, bool added) { return (int)((unsigned)e | ((unsigned)x << 15) | ((unsigned)y << 23) | (added ? 0x80000000u : 0u)); }
template <int G> __device__ __forceinline__ int wl_inc_env(int raw) { return raw & (G == 16 ? WL_INC_ENV_MASK : WL_INC64_ENV_MASK); }
template <int G> __device__ __forceinline__ int wl_inc_row(int raw) { return G == 16 ? ((raw >> 26) & 15) : ((raw >> 25) & 63); }
template <int G> __device__ __forceinline__ int wl_inc_col(int raw) { return G == 16 ? ((raw >> 21) & 31) : ((raw >> 19) & 63); }
template <int G> __device__ __forceinline__ unsigned wl_inc_code(int raw) { return G == 16 ? (((unsigned)raw >> 30) & 3u) : ((unsigned)raw >> 31); }
// An item of the changed list with this bit set is an unchanged environment whose episode ended (iteration cap):
// k_stats resets it without recomputing anything.
#define WL_RESET_ONLY (1 << 30)
#define SMB_KEEP_PLAY (1 << 29)    /* smb, changed list: the change cannot alter the play-through (k_update), k_smb keeps the previous one */

// Optional in-kernel timeline (tools/timeline.py builds a copy of the library with -DPCGRL_TIMELINE; the product is
// compiled without it and TL() is nothing): wavefront-private slots, 100 MHz wall clock << 8 | tag.
#if defined(PCGRL_TIMELINE) || defined(PCGRL_SMB_PROF) || defined(PCGRL_BIG_PROF) || defined(PCGRL_WIDE_TL)
__device__ unsigned long long* g_tl_buf;
#endif
// developer build (tools/probe/wide_blocks.py, -DPCGRL_WIDE_TL): k_stats_wide, per block eight words -- entry, lists read, item kind, item
// done, end (100 MHz wall clock) -- written by thread 0 only, so that the kernel keeps its registers and its two blocks per compute unit
#ifdef PCGRL_WIDE_TL
#define WTL(i, v) do { if (threadIdx.x == 0 && g_tl_buf) g_tl_buf[(size_t)blockIdx.x * 8 + (i)] = (unsigned long long)(v); } while (0)
#else
#define WTL(i, v) do {} while (0)
#endif
#ifdef PCGRL_TIMELINE
#define TL_SLOTS 48
__shared__ int s_tl_idx[16];
__device__ __forceinline__ void tl_mark(int tag) {
    if ((threadIdx.x & 63) == 0 && g_tl_buf) {
        const int wv = threadIdx.x >> 6;
        const int i = s_tl_idx[wv];
        if (i < TL_SLOTS) { g_tl_buf[((size_t)blockIdx.x * (blockDim.x >> 6) + wv) * TL_SLOTS + i] = ((unsigned long long)wall_clock64() << 8) | (unsigned)tag; s_tl_idx[wv] = i + 1; }
    }
}
#define TL_INIT() do { if ((threadIdx.x & 63) == 0) s_tl_idx[threadIdx.x >> 6] = 0; } while (0)
#define TL(tag) tl_mark(tag)
#else
#define TL_INIT() do {} while (0)
#define TL(tag) do {} while (0)
#endif

struct LocalLists;
// The wrapped observation a handle keeps up to date (pcgrl_bind_observation; kernels_obs.h): uint8 [N][oh][ow][depth], written
// by the last kernel of every step / reset.  out == nullptr: off.
struct ObsSpec { uint8_t* out; int32_t oh, ow, depth, centered, pad;
                 int32_t fused,        // k_step writes the image itself (a shape with a lean routine, kernels_obs.h); else k_obs follows the step
                         delta; };     // ... and may update it in place: `out` holds the image of the state the step starts from
struct DevBufs {
    uint8_t* map; uint8_t* old_map; uint16_t* heat; uint8_t* pos; void* planes;
    void* champ;                     // mask [N][16]: rows of the champion component (binary, 16-row maps); stats[e][2] = it is valid
                                     // smb: uint32 [N][W], per map column the rows its last play-through read (kernels_smb.h)
    int32_t* wide_sync;              // tall binary maps (k_stats_wide): i32 [N][4] = {epoch: planes read, epoch: result there, regions | path << 16, claim} --
    int32_t wide_spin;               //   sleeps the odd block of a certain reset waits for the even one before it takes the old half over (pcgrl_tuning wide_spin)
    int32_t wide_few;                //   (regions up to which a tall map counts as "few regions": WL_WIDE_FEW_REGIONS, PCGRL_WIDE_FEW for experiments)
    int32_t wide_epoch;              //   how the two blocks of a certain reset (old map / new map) talk; the launch's epoch (host counter)
    const uint16_t* heat_end;        // end of the caller's heatmap buffer
    // optional episode statistics (pcgrl_bind_episode_stats): running return/length, latched at the end of an episode
    double* ep_return; int32_t* ep_length; double* last_return; int32_t* last_length;
    int32_t* counters; int32_t* stats; int32_t* start_stats; int32_t* info;
    double* reward; uint8_t* done; double* tile_p;
    uint32_t* rng_rep; uint32_t* rng_prob; int32_t* rng_cur;
    int32_t* wl_cnt;                 // [2 parities][WL_NLIST][WL_NSHARD * WL_CSTRIDE]
    int32_t* wl_items[WL_NLIST];     // [WL_NSHARD][wl_cap[list]]
    int32_t wl_cap[WL_NLIST];
    // sokoban solver arena (per resident solver block) and sticky status word
    SokNode* sok_pool; uint32_t* sok_heap; uint32_t* sok_table; int32_t* status;
    int32_t* sok_res;                // [num_envs][4 agents][win, h, depth, exhausted]
    int32_t* sok_cnt; int32_t* sok_stop;   // [num_envs] agents reported / stop level (kernels_sokoban.h)
    struct LocalLists* local;        // non-null: wl_push goes to these block-local LDS lists (k_step_solver sets it in its own copy)
    int32_t md_only_agent;           // >= 0: k_mdungeon runs only this agent (PCGRL_MD_ONLY_AGENT, timing experiments; results are then wrong)
    int32_t* sok_sync;               // [2 launches per step][SOK_SY_WORDS + SOK_HARD_CAP] scheduling words
    int32_t sok_pool_stride, sok_heap_stride, sok_table_size, sok_use_lds;
    int32_t sok_spawn_iters;         // a Sokoban BFS still running after that many pops publishes its level (PCGRL_SOK_SPAWN: experiments)
    int32_t sok_hard_cap;            // levels k_sokoban may publish per launch (SOK_HARD_CAP; PCGRL_SOK_HARD_CAP lowers it for tests)
    int32_t sok_fast_maxc;           // most crates the register-resident search takes (SOKF_MAXC; -1: PCGRL_SOK_GENERIC=1 forces the generic one)
    int32_t inline_reset;   // k_stats resets finished environments itself (every problem but Sokoban)
    int32_t pair_min;       // from this many certain resets per launch on, a wavefront of k_stats takes two of them (PCGRL_PAIR_MIN)
    int32_t zelda_inc;      // zelda, single-cell representations, maps of at most 16 x 32: changed/incremental items carry (cell, passability change)
    // Narrow representation: the next PCGRL_FIFO_N words of the representation stream, computed ahead (untempered) -- a derived
    // cache of (ring, cursor), valid while fifo_tag[e] equals the cursor.  The fused step kernel draws the cursor moves from it
    // (no ring access on the step's critical path) and refills it behind the barrier; every reset rebuilds it; the other
    // pipelines draw from the ring and mark it invalid (-1).
    uint32_t* fifo; int32_t* fifo_tag;
    ObsSpec obs;
    int32_t step_fpw, step_ipw;      // k_step: full / incremental items per wavefront task (1, 2 or 4)
    int32_t step_prio;               // k_step: s_setprio levels by kind of work (pcgrl_tuning step_prio)
    int32_t step_tight;              //   bit 0: a full recomputation of the step sweeps the second largest component for a tight bound, bit 1: a reset does
    const int32_t* flat;             // k_step, wide representation: this step's actions as the ActionMap wrapper's flat indices (pcgrl_step_flat), else null
    int32_t step_pair;               // k_step: certain resets per block and step from which a wavefront takes two of them (0: never; pcgrl_tuning step_pair)
    int32_t big_team;                // k_big, binary: the full recomputations of a step by whole blocks (bigmap_team.h; pcgrl_tuning big_team)
    int32_t step_touch;              // k_step, binary: changes in or next to the champion try binary_touch before a full recomputation (pcgrl_tuning no_touch)
    uint8_t* big_arena;              // search_big.h: per-block node pool + heap + visited table (levels / solver_power beyond the compact searches)
    // pcgrl_step_async (kernels_search_async.h): non-null only inside a tick -- pending[e] != 0: environment e's step is in flight
    // (a suspended search), it takes no action; async_stats[0] counts the actions that were taken
    uint8_t* pending; unsigned long long* async_stats;
    // ... and the slots of the suspended searches, for the update kernel's side job: the threads of its first blocks list the
    // runnable slots (async_run[0 .. *async_run_n)) for the search kernel that follows
    const uint8_t* async_slots; size_t async_slot_bytes; int32_t async_nslots; int32_t* async_run; int32_t* async_run_n;
};
#define PCGRL_FIFO_N 8

// Block-local state of the fused step kernel (k_step) that the shared device functions have to know about: the block keeps
// the per-environment state of its 64 environments in LDS (DevBufs pointers rebased into it) for the whole launch.
// zelda's reward by lanes (zelda_reward_lanes, kernels_stats.h): weight and band of term k, in the order get_reward sums them
struct ZeldaRewardTab { double w[8]; int lo[8]; int hi[8]; };
struct ObsView {          // one block's view of the observation target: environments [0, ne) relative to the block's first one (kernels_obs.h)
    uint8_t* out;         // the block's stretch of the output: image of its first environment (16-byte aligned)
    int oh, ow, depth, centered, pad, W, H;
};
struct StepLocal {
    int e0;
    int par, need;              // this step's slot of refill_done; how many update wavefronts have to report
    int refill_done[2];         // update wavefronts that are through with the cursor moves of the step (their byte-map / heat-map writes have landed)
    // Draw-cache words an environment consumed in this step that are not in its ring yet.  The update wavefronts write them to the
    // rings and top the caches up only after the block's tasks (ring traffic nobody waits for); an episode end nobody saw coming
    // needs the ring earlier, so the two sides claim the words with an atomic exchange: k > 0 = that many pending, 0 = none,
    // -1 = the update wavefront has taken them (late_done[its index] says when they have landed).
    int pend[256];
    int late_done[4];
    uint8_t dirty[256];         // planes / champion / start statistics changed: write them back
    // the wrapped observation written by k_step while it runs (round 6; kernels_step.h "observation tasks"):
    uint8_t obs_skip[256];      //   certain to be reset in this step: the image is written by the reset's wavefront, not by an observation task
    uint8_t late[256];          //   reset by an episode end nobody saw coming: the image is written again at the end of the launch
    int n_late;
    ObsView obs_v;              //   the block's view, written once by thread 0: a task that writes an image reads it from here (kept in scalar
                                //   registers across the kernel it cost zelda's instantiation, which is at its limit of 100, 80 bytes of scratch)
    ZeldaRewardTab zr;          // (zelda; written once by thread 0 at the start of k_step: constant indices into the parameter block only)
};

// thread gi of a launch looks at slot gi of the suspended searches (header word 0 = state, 1 = runnable: kernels_search_async.h)
__device__ __forceinline__ void async_list_runnable(const DevBufs& B, int gi) {
    if (gi < B.async_nslots && __hip_atomic_load(reinterpret_cast<const int32_t*>(B.async_slots + (size_t)gi * B.async_slot_bytes), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1)
        B.async_run[atomicAdd(B.async_run_n, 1)] = gi;
}
__device__ __forceinline__ int32_t* wl_counters(const DevBufs& B, int parity, int list) {
    return B.wl_cnt + (size_t)(parity * WL_NLIST + list) * WL_NSHARD * WL_CSTRIDE;
}
// Block-local work lists in LDS (k_step_solver: a block that owns its environments for a whole tape of steps keeps its
// lists to itself).  Entries are environment indices relative to the block's first environment.
#define WL_LOCAL_CAP 512
struct LocalLists { int n[WL_NLIST]; int e0; uint16_t items[WL_NLIST][WL_LOCAL_CAP]; };
__device__ __forceinline__ void wl_push(const DevBufs& B, int parity, int list, int shard, int value) {
    if (B.local) {
        const int i = atomicAdd(&B.local->n[list], 1);
        B.local->items[list][i] = (uint16_t)(value - B.local->e0);
        return;
    }
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
// Up to three lists at once (one barrier, the counter loads of all in flight together): wavefront k of the block takes
// list k (a negative id = no list: length 0).  Block of at least 192 threads; returns the length of list_a.
__device__ __forceinline__ int wl_load_prefix3(const DevBufs& B, int parity, int list_a, int list_b, int list_c, int* s_pref_a,
                                               int* s_pref_b, int* s_pref_c, int* n_b, int* n_c) {
    if (threadIdx.x < 3 * WL_NSHARD) {
        const int t = threadIdx.x & (WL_NSHARD - 1), k = threadIdx.x >> 6;
        const int list = k == 0 ? list_a : (k == 1 ? list_b : list_c);
        int v = list >= 0 ? wl_counters(B, parity, list)[t * WL_CSTRIDE] : 0;   // (also list_a may be absent)
        for (int o = 1; o < WL_NSHARD; o <<= 1) {
            const int u = __shfl_up(v, o, 64);
            if (t >= o) v += u;
        }
        int* sp = k == 0 ? s_pref_a : (k == 1 ? s_pref_b : s_pref_c);
        sp[t + 1] = v;
        if (t == 0) sp[0] = 0;
    }
    __syncthreads();
    *n_b = s_pref_b[WL_NSHARD];
    *n_c = s_pref_c[WL_NSHARD];
    return s_pref_a[WL_NSHARD];
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

// Two lists in one pass: the two global atomics share a round trip.  Every thread of the block must call this.
__device__ __forceinline__ void block_append2(bool flag_a, int value_a, int list_a, bool flag_b, int value_b, int list_b,
                                              const DevBufs& B, int parity, int (*s_cnt)[4], int* s_base) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int shard = blockIdx.x & (WL_NSHARD - 1);
    const uint64_t ma = __ballot(flag_a), mb = __ballot(flag_b);
    if (lane == 0) { s_cnt[0][w] = __popcll(ma); s_cnt[1][w] = __popcll(mb); }
    __syncthreads();
    if (threadIdx.x == 0 || threadIdx.x == 64) {
        const int k = threadIdx.x >> 6;
        const int tot = s_cnt[k][0] + s_cnt[k][1] + s_cnt[k][2] + s_cnt[k][3];
        s_base[k] = tot ? atomicAdd(wl_counters(B, parity, k ? list_b : list_a) + shard * WL_CSTRIDE, tot) : 0;
    }
    __syncthreads();
    if (flag_a || flag_b) {
        const int k = flag_a ? 0 : 1;
        const int list = k ? list_b : list_a;
        int off = s_base[k];
        for (int i = 0; i < w; i++) off += s_cnt[k][i];
        off += __popcll((k ? mb : ma) & ((1ull << lane) - 1ull));
        B.wl_items[list][(size_t)shard * B.wl_cap[list] + off] = k ? value_b : value_a;
    }
}

// Three lists (changed, incremental, certain resets) in one pass.  dest: -1 none, 0 = WL_CHG, 1 = WL_INC, 2 = WL_RST.
__device__ __forceinline__ void block_append3(int dest, int value, const DevBufs& B, int parity, int (*s_cnt)[4], int* s_base) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int shard = blockIdx.x & (WL_NSHARD - 1);
    const uint64_t m0 = __ballot(dest == 0), m1 = __ballot(dest == 1), m2 = __ballot(dest == 2);
    if (lane == 0) { s_cnt[0][w] = __popcll(m0); s_cnt[1][w] = __popcll(m1); s_cnt[2][w] = __popcll(m2); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 192) {
        const int k = threadIdx.x >> 6;
        const int list = k == 0 ? WL_CHG : (k == 1 ? WL_INC : WL_RST);
        const int tot = s_cnt[k][0] + s_cnt[k][1] + s_cnt[k][2] + s_cnt[k][3];
        s_base[k] = tot ? atomicAdd(wl_counters(B, parity, list) + shard * WL_CSTRIDE, tot) : 0;
    }
    __syncthreads();
    if (dest >= 0) {
        const int list = dest == 0 ? WL_CHG : (dest == 1 ? WL_INC : WL_RST);
        const uint64_t m = dest == 0 ? m0 : (dest == 1 ? m1 : m2);
        int off = s_base[dest];
        for (int i = 0; i < w; i++) off += s_cnt[dest][i];
        off += __popcll(m & ((1ull << lane) - 1ull));
        B.wl_items[list][(size_t)shard * B.wl_cap[list] + off] = value;
    }
}

// Changed environments are bucketed by how hard their statistics are expected to be (the previous
// stats are a good predictor: one tile changed), one bucket per shard, so that the four maps sharing a
// wavefront in k_stats have similar trip counts.  Every thread of the block must call this.
__device__ __forceinline__ void block_append_bucketed(bool flag, int bucket, int value, const DevBufs& B, int parity, int list,
                                                      int* s_hist, int* s_gbase, bool flag2 = false, int value2 = 0, int list2 = -1) {
    // s_hist / s_gbase have WL_NSHARD + 1 entries: the last one serves the optional second list (plain, one shard per
    // block like block_append), whose global atomic shares the round trip of the bucket atomics.
    if (threadIdx.x <= WL_NSHARD) s_hist[threadIdx.x] = 0;
    __syncthreads();
    int rank = 0;
    if (flag) rank = atomicAdd(&s_hist[bucket], 1);
    else if (flag2) rank = atomicAdd(&s_hist[WL_NSHARD], 1);
    __syncthreads();
    const int shard2 = blockIdx.x & (WL_NSHARD - 1);
    if (threadIdx.x < WL_NSHARD) {
        const int c = s_hist[threadIdx.x];
        if (c > 0) s_gbase[threadIdx.x] = atomicAdd(wl_counters(B, parity, list) + threadIdx.x * WL_CSTRIDE, c);
    } else if (threadIdx.x == WL_NSHARD && list2 >= 0) {
        const int c = s_hist[WL_NSHARD];
        if (c > 0) s_gbase[WL_NSHARD] = atomicAdd(wl_counters(B, parity, list2) + shard2 * WL_CSTRIDE, c);
    }
    __syncthreads();
    if (flag) B.wl_items[list][(size_t)bucket * B.wl_cap[list] + s_gbase[bucket] + rank] = value;
    else if (flag2) B.wl_items[list2][(size_t)shard2 * B.wl_cap[list2] + s_gbase[WL_NSHARD] + rank] = value2;
}
__device__ __forceinline__ int difficulty_bucket(const PcgrlParams& P, const int4& s0, const int4& s1, int B_few = WL_WIDE_FEW_REGIONS) {
    if (P.prob == PCGRL_PROB_BINARY && P.group == 64) {
        // tall maps (k_stats_wide, a block per item, the blocks start in list order and only ~512 are resident): the dearest
        // first.  A full recomputation costs ~0.8 us per region spread over the block's eight wavefronts plus ~0.1 us per step of the
        // longest path on one of them (tools/timeline_wide.py): the two count alike; shard 0 = dearest
        // Two classes, the upper half of the shards for maps with few regions (one long double sweep on one wavefront and little
        // else: those go two to a block in k_stats_wide), dearest first within a class.
        const int cost = (max(s0.x, 0) + max(s0.y, 0)) / 12;
        const int half = WL_NSHARD / 2;
        return (s0.x <= B_few ? half : 0) + half - 1 - min(cost, half - 1);
    }
    if (P.prob == PCGRL_PROB_BINARY) {   // (path-length / 6, regions / 3), 8 x 8
        const int a = min(max(s0.y, 0) / 6, 7), b = min(max(s0.x, 0) / 3, 7);
        return a * 8 + b;
    }
    const int regions = P.prob == PCGRL_PROB_ZELDA ? s1.x : s0.w;
    return min(max(regions, 0), WL_NSHARD - 1);
}

// What stable-baselines' Monitor does around the reference env (utils.py:13-29, 60-71): sum the rewards of the running
// episode in step order and count its steps; when the episode ends, latch both and start over.
__device__ __forceinline__ void episode_account(const DevBufs& B, int e, double r, bool d) {
    if (!B.ep_return) return;
    double R = B.ep_return[e] + r;
    int L = B.ep_length[e] + 1;
    if (d) { B.last_return[e] = R; B.last_length[e] = L; R = 0.0; L = 0; }
    B.ep_return[e] = R; B.ep_length[e] = L;
}

// heatmap[cell] += 1 (pcgrl_env.py:137) without waiting for the old value: a no-return 32-bit atomic add on the
// word that holds the 16-bit counter (counts stay below 2^16, so a half never carries into the other).  Only the
// very last counter of a buffer with an odd number of cells has no complete word; it takes the plain path.
__device__ __forceinline__ void heat_increment(const DevBufs& B, uint16_t* cell) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(cell);
    uint32_t* word = reinterpret_cast<uint32_t*>(a & ~(uintptr_t)3);
    if (reinterpret_cast<uintptr_t>(word) + 4 <= reinterpret_cast<uintptr_t>(B.heat_end)) atomicAdd(word, 1u << ((a & 2) * 8));
    else *cell += 1;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
