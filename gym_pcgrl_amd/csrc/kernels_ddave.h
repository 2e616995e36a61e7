// k_ddave: the planner jobs that k_stats / k_reset parked for the ddave problem (ddave_solver.h), one wavefront per
// search.  Part of the single translation unit pcgrl_abi.hip.
//
// The same scheme as k_mdungeon: the four agents of a level (A*(1), A*(0.5), A*(0), BFS; ddave_prob.py:111-127) are
// four tickets and run concurrently in different workgroups, every agent records (win, h, depth, jumps, diamonds) and
// the last of the four to finish selects what the sequential loop would have returned.  An agent stops early only when
// an earlier agent has won: this engine's visited key ignores the air time, so the states an agent gets to see depend
// on its own order of exploration and an exhausted agent says nothing about the others (ddave_solver.h).
#pragma once

struct DdPollHook {
    const int32_t* stop; int a;
    __device__ __forceinline__ bool operator()(int it) const { return (it & SOK_POLL_MASK) == 0 && (sok_ld(stop) & 255) >= 4 - a; }
};

// The four children of a pop, one per lane (lanes 0..3 run the compact search in lockstep).
struct DdKidsLanes {
    int lane;
    // this lane's child only (two-wavefront searches: each lane files its own child)
    template <class TP>
    __device__ __forceinline__ DdChild mine(const DdLevel& L, const DdFastLevel& F, TP table, int table_mask, uint64_t key, int aj, bool ground,
                                            bool ceiling) const {
        return ddf_child(L, F, table, table_mask, key, aj, ground, ceiling, lane & 3);
    }
    template <class TP>
    __device__ __forceinline__ void operator()(const DdLevel& L, const DdFastLevel& F, TP table, int table_mask, uint64_t key, int aj, bool ground,
                                               bool ceiling, DdChild* out) const {
        const DdChild mine = ddf_child(L, F, table, table_mask, key, aj, ground, ceiling, lane & 3);
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine.key, d);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine.key >> 32), d);
            out[d].key = ((uint64_t)hi << 32) | lo;
            out[d].h = __builtin_amdgcn_readlane(mine.h, d);
            out[d].aj = __builtin_amdgcn_readlane(mine.aj, d);
            out[d].drop = __builtin_amdgcn_readlane(mine.drop, d);
        }
    }
};

// Agent a of environment e is done.  The fourth report selects the result and finishes the item.
__device__ __forceinline__ void dd_report(const PcgrlParams& P, const DevBufs& B, int e, int a, bool win, const int* out4, int mode, int parity,
                                          int rst_list) {
    int32_t* r = B.sok_res + ((size_t)e * 4 + a) * 4;
    __hip_atomic_store(r + 0, win ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r + 1, out4[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r + 2, out4[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r + 3, (out4[2] & 0xFFFF) | ((out4[3] & 255) << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (win) atomicMax(B.sok_stop + e, 3 - a);
    __threadfence();
    if (atomicAdd(B.sok_cnt + e, 1) != 3) return;
    __threadfence();
    int chosen = 3;
    for (int k = 2; k >= 0; k--) if (sok_ld(B.sok_res + ((size_t)e * 4 + k) * 4)) chosen = k;
    const int32_t* q = B.sok_res + ((size_t)e * 4 + chosen) * 4;
    const int jd = sok_ld(q + 3);
    const int res4[4] = {sok_ld(q + 1), sok_ld(q + 2), jd & 0xFFFF, (jd >> 16) & 255};
    B.sok_cnt[e] = 0;      // ready for the next job of this environment (a later launch)
    B.sok_stop[e] = 0;
    int32_t s[PCGRL_MAX_STATS];
    const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
    for (int k = 0; k < 8; k++) s[k] = park[k];
    dd_pack(s, res4);
    finalize_item<PCGRL_PROB_DDAVE>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), true, rst_list);
}

// Jobs = list_a (mode_a) followed by list_b (mode_b); list_b < 0: none.  Environments that finish their episode here
// go to `rst_list`.
// (a template only so that every part of the library can include this header: instantiated where it is launched)
template <int PART_TAG>
// Two wavefronts per block: the search wavefront (everything below) and the heap server of its A* searches (sokoban_fast.h).
__global__ __launch_bounds__(128) void k_ddave(PcgrlParams P, DevBufs B, int list_a, int mode_a, int list_b, int mode_b, int parity,
                                              int rst_list, int32_t* sync, int clear_parity) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dd_lds[];
    __shared__ int s_pref_a[WL_NSHARD + 1], s_pref_b[WL_NSHARD + 1];
    __shared__ DdLevel s_L;              // level + node workspace in LDS: they are indexed dynamically
    __shared__ DdNode s_root, s_work;
    __shared__ DdFastLevel s_F;
    __shared__ DdFastNode s_cache[4];
    __shared__ int s_fast;
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x & 63;
    const int n_a = wl_load_prefix(B, parity, list_a, s_pref_a);
    const int n_b = list_b >= 0 ? wl_load_prefix(B, parity, list_b, s_pref_b) : 0;
    __shared__ SokDuoBox s_box;
    if (threadIdx.x >= 64) { sok_duo_server(dd_lds, &s_box, lane); return; }
    SokDuoBox* const duo = B.sok_use_lds ? &s_box : nullptr;       // (the heap has to be the LDS one)
    const int n = n_a + n_b;
    DdNode* pool = reinterpret_cast<DdNode*>(B.sok_pool + (size_t)blockIdx.x * B.sok_pool_stride);
    uint32_t* g_heap = B.sok_use_lds ? nullptr : B.sok_heap + (size_t)blockIdx.x * B.sok_heap_stride;
    uint32_t* g_table = B.sok_use_lds ? nullptr : B.sok_table + (size_t)blockIdx.x * B.sok_table_size;
    const int tsize = B.sok_use_lds ? SOK_LDS_TABLE : B.sok_table_size;
    const int W = P.width, H = P.height;
    const int KS[4] = {2, 1, 0, -1};
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(sync + SOK_SY_TICKET_A, 1);
        t = __shfl(t, 0, 64);
        if (t >= 4 * n) break;
        const int job = t >> 2, a = t & 3;
        int e, mode;
        if (job < n_a) { e = wl_get(B, list_a, s_pref_a, job); mode = mode_a; }
        else { e = wl_get(B, list_b, s_pref_b, job - n_a); mode = mode_b; }
        int skip = 0;
        if (lane == 0) {
            const DdPollHook hook = {B.sok_stop + e, a};
            skip = hook(0) ? 1 : 0;                                  // already decided before this agent started
        }
        skip = __shfl(skip, 0, 64);
        if (!skip) {
            // the level by all 64 lanes (level_build_wave.h); the compact search (ddave_fast.h) for levels with few diamonds;
            // PCGRL_SOK_GENERIC=1 switches it off (tests)
            const int nd = dd_build_level_wave(B.map + (size_t)e * W * H, W, H, s_L, s_root, s_F, lane);
            if (lane == 0) s_fast = (nd <= DDF_MAXD && B.sok_use_lds && B.sok_fast_maxc >= 0) ? 1 : 0;
        }
        __threadfence_block();
        const int fast = skip ? 0 : s_fast;
        if (!skip) {
            if (fast) { for (int i = lane; i < 2 * tsize; i += 64) dd_lds[SOK_LDS_HEAP + i] = 0; }   // 64-bit keys
            else if (B.sok_use_lds) { for (int i = lane; i < tsize; i += 64) dd_lds[SOK_LDS_HEAP + i] = 0; }
            else { for (int i = lane; i < tsize; i += 64) g_table[i] = 0; }
        }
        __threadfence_block();
        if (lane < (fast ? 4 : 1)) {
            int it = 0, out4[4] = {0, 0, 0, 0};
            bool exhausted = false, win = false;
            if (!skip) {
                const DdPollHook hook = {B.sok_stop + e, a};
                if (fast) {
                    uint64_t key = 0;
                    int hh = 0, dd = 0, jj = 0;
                    const DdKidsLanes kids = {lane};
                    win = dd_search_fast(s_L, s_F, reinterpret_cast<DdFastNode*>(pool), dd_lds, reinterpret_cast<uint64_t*>(dd_lds + SOK_LDS_HEAP),
                                         tsize - 1, s_cache, s_root, KS[a], P.solver_power, key, hh, dd, jj, it, exhausted, hook, kids, duo);
                    ddf_result(s_F, key, hh, dd, jj, win, out4);
                } else if (B.sok_use_lds)   // two instantiations: LDS pointers compile to ds_* instructions
                    win = dd_search(s_L, pool, dd_lds, dd_lds + SOK_LDS_HEAP, tsize - 1, s_work, s_root, KS[a], P.solver_power, it, exhausted, hook);
                else
                    win = dd_search(s_L, pool, g_heap, g_table, tsize - 1, s_work, s_root, KS[a], P.solver_power, it, exhausted, hook);
                if (!fast) dd_result(s_L, s_work, win, out4);
            }
            if (lane == 0) dd_report(P, B, e, a, win, out4, mode, parity, rst_list);
        }
        __threadfence_block();
    }
    s_box.session = 0;          // the heap server leaves with us
    sok_duo_sync();
}
