// k_sokoban: the solver jobs that k_stats / k_reset parked (sokoban_solver.h), one wavefront per search.
// Part of the single translation unit pcgrl_abi.hip (see its header comment for the overall picture).
//
// SokobanProblem._run_game runs BFS, A*(1), A*(0.5), A*(0) one after the other and stops at the first
// winner (sokoban_prob.py:104-122).  The agents are independent searches from the same root, so only the
// *selection* is sequential.  A step of 131 072 environments almost always holds a few levels whose BFS
// runs into the 5 000-pop cap, and those four back-to-back capped searches were the long pole of the whole
// step.  Here a BFS that is still running after SOK_SPAWN_ITERS pops publishes its level on the "hard"
// list; idle solver blocks pick up the three A* agents of that level and run them concurrently with the
// BFS.  Every agent records (win, h, depth, exhausted); the last of the four to finish selects exactly
// what the sequential loop would have returned.  An agent whose result cannot matter any more (an earlier
// agent won, or BFS exhausted the state space: see the shortcut in sokoban_solver.h) is abandoned at its
// next poll.  Short jobs (the median BFS ends after ~30 pops) never spawn anything.
//
// Scheduling is by tickets on words the host zeroes before the launch: BFS jobs are handed out with an
// atomic counter; A* tickets are only ever taken for published levels (compare-and-swap), so no block
// waits on work that may never come.  A block leaves when every BFS has finished (the hard list is then
// final) and every A* ticket has been taken.  Blocks only ever wait for blocks that hold a ticket, i.e.
// that are resident and running: no forward-progress assumption between unscheduled blocks.
#pragma once

enum { SOK_SY_TICKET_A = 0, SOK_SY_TICKET_B = 1, SOK_SY_HARD = 2, SOK_SY_BFS_DONE = 3, SOK_SY_WORDS = 16 };
#define SOK_HARD_CAP 4096          /* published levels per launch; beyond it a BFS block runs its A* agents itself */
#define SOK_SPAWN_ITERS 128

__device__ __forceinline__ int sok_ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct SokSpawnHook {     // BFS: publish the level once the search has proven to be a long one
    int32_t* sync; int32_t* hard; int spawn_at; int tag; int* spawned; int lane; int cap;
    __device__ __forceinline__ bool operator()(int it) const {
        if (it == spawn_at && lane == 0) {
            const int idx = atomicAdd(sync + SOK_SY_HARD, 1);
            if (idx < cap) { __hip_atomic_store(hard + idx, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *spawned = 1; }
        }
        return false;
    }
};
struct SokPollHook {      // A*: stop when the result cannot be selected any more
    const int32_t* stop; int need;
    __device__ __forceinline__ bool operator()(int it) const { return (it & SOK_POLL_MASK) == 0 && sok_ld(stop) >= need; }
};

// The four children of a pop, one per lane (lanes 0..3 run the search in lockstep; everything else in it is
// uniform across them).  The results come back through v_readlane, i.e. as scalars.
struct SokKidsLanes {
    int lane, dir;      // dir: this lane's move as a cell offset (sokf_dir(lane & 3, level width)), made once per search
    // this lane's child only (two-wavefront searches: each lane files its own child)
    template <int NW>
    __device__ __forceinline__ SokChild mine(const SokFastLevel<NW>& F, uint64_t cr, const uint64_t* cb, int player, int h) const {
        return sokf_child_dir<NW>(F, cr, cb, player, h, dir);
    }
    template <int NW>
    __device__ __forceinline__ void operator()(const SokFastLevel<NW>& F, uint64_t cr, const uint64_t* cb, int player, int h, SokChild* out) const {
        const SokChild mine = sokf_child_dir<NW>(F, cr, cb, player, h, dir);
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine.cr, d);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine.cr >> 32), d);
            out[d].cr = ((uint64_t)hi << 32) | lo;
            out[d].np = __builtin_amdgcn_readlane(mine.np, d);
            out[d].h = __builtin_amdgcn_readlane(mine.h, d);
            out[d].ok = __builtin_amdgcn_readlane(mine.ok, d);
        }
    }
};

// One agent of environment e is done.  The fourth report selects the result and finishes the item.
__device__ __forceinline__ void sok_report(const PcgrlParams& P, const DevBufs& B, int e, int a, bool win, int hh, int dd, bool exhausted,
                                           int mode, int parity, int rst_list) {
    int32_t* r = B.sok_res + ((size_t)e * 4 + a) * 4;
    __hip_atomic_store(r + 0, win ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r + 1, hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r + 2, dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(r + 3, exhausted ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // agents after `a` are not needed once a wins (or BFS has expanded every reachable state): stop level 3 - a
    if (win || (a == 0 && exhausted)) atomicMax(B.sok_stop + e, 3 - a);
    __threadfence();
    if (atomicAdd(B.sok_cnt + e, 1) != 3) return;
    __threadfence();
    int dist = 0, sol = 0;
    bool chosen = false;
    for (int k = 0; k < 4 && !chosen; k++) {
        const int32_t* q = B.sok_res + ((size_t)e * 4 + k) * 4;
        if (sok_ld(q + 0)) { dist = 0; sol = sok_ld(q + 2); chosen = true; }
    }
    if (!chosen) {
        const int32_t* q0 = B.sok_res + (size_t)e * 16;
        dist = sok_ld(q0 + 3) ? sok_ld(q0 + 1) : sok_ld(q0 + 12 + 1);   // exhausted BFS, else the last agent's best node
    }
    B.sok_cnt[e] = 0;      // ready for the next job of this environment (a later launch)
    B.sok_stop[e] = 0;
    int32_t s[PCGRL_MAX_STATS];
    const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
    for (int k = 0; k < 8; k++) s[k] = park[k];
    s[4] = dist; s[5] = sol;
    finalize_item<PCGRL_PROB_SOKOBAN>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), true, rst_list);
}

// One agent on the level in L: the register-resident search (sokoban_fast.h; LDS heap + 64-bit-key table) when the
// level qualifies, else the generic one (LDS or global-arena heap/table).  Called by lanes 0..3 for the former (the
// four children of a pop are made side by side), by lane 0 for the latter.
template <class Hook>
__device__ __forceinline__ bool sok_run_agent(const DevBufs& B, int power, const SokLevel& L, SokNode& work, const SokNode& root, SokNode* pool,
                                              uint32_t* lds, SokFastNode* cache, uint32_t* g_heap, uint32_t* g_table, int tsize, int fast, int k,
                                              int& hh, int& dd, int& it, bool& exhausted, Hook hook, int lane, int table_off = SOK_LDS_HEAP,
                                              SokDuoBox* duo = nullptr) {
    if (fast) {
        const SokKidsLanes kids = {lane, sokf_dir(lane & 3, L.w)};
        uint64_t* tab = reinterpret_cast<uint64_t*>(lds + table_off);
        SokFastNode* fp = reinterpret_cast<SokFastNode*>(pool);
        if (L.cells <= 64) return sok_search_fast<1>(L, fp, lds, tab, tsize - 1, cache, root, k, power, hh, dd, it, exhausted, hook, kids, duo);
        return sok_search_fast<4>(L, fp, lds, tab, tsize - 1, cache, root, k, power, hh, dd, it, exhausted, hook, kids, duo);
    }
    if (lane != 0) return false;
    if (B.sok_use_lds)   // two instantiations: LDS pointers compile to ds_* instructions
        return sok_search(L, pool, lds, lds + table_off, tsize - 1, work, root, k, power, hh, dd, it, exhausted, hook);
    return sok_search(L, pool, g_heap, g_table, tsize - 1, work, root, k, power, hh, dd, it, exhausted, hook);
}

// pcgrl_selftest_heap: the heap server's two primitives (sok_duo_append / sok_duo_repair) driven by a tape of operations, so that
// a test can hold them against CPython's heapq slot for slot.  ops[i] = a packed word to push, or 0xFFFFFFFF = pop (the popped
// word goes to pops[], 0xFFFFFFFF for an empty heap).  One wavefront; the heap (at most `cap` words) in dynamic LDS.
template <int PART_TAG>
__global__ __launch_bounds__(64) void k_selftest_heap(const uint32_t* ops, int n_ops, int cap, uint32_t* pops, uint32_t* heap_out, int32_t* n_out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t st_heap[];
    const int lane = threadIdx.x;
    const SokDuoLanes T = sok_duo_lanes(lane);
    int n = 0, np = 0;
    for (int i = 0; i < n_ops; i++) {
        const uint32_t w = ops[i];
        if (w != 0xFFFFFFFFu) { if (n < cap) { sok_duo_append(st_heap, n, w, lane); n++; } continue; }
        uint32_t top = 0xFFFFFFFFu;
        if (n > 0) {
            top = st_heap[0];
            const uint32_t last = st_heap[--n];
            if (n > 0) {
                const uint32_t root = sok_duo_repair(st_heap, n, last, lane, T);
                if (root != st_heap[0]) top = 0xFFFFFFFEu;       // (the word the server hands the search wavefront is the new root)
            }
        }
        if (lane == 0) pops[np] = top;
        np++;
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) heap_out[i] = st_heap[i];
    if (lane == 0) *n_out = n;
}

// Jobs = list_a (mode_a) followed by list_b (mode_b); list_b < 0: none.  `sync`/`hard` are this launch's
// zeroed scheduling words.  Environments that finish their episode here go to `rst_list`.
// (a template only so that every part of the library can include this header: instantiated where it is launched)
// Two wavefronts per block: the search wavefront (everything below) and the heap server of its A* searches (sokoban_fast.h).
template <int PART_TAG>
__global__ __launch_bounds__(128) void k_sokoban(PcgrlParams P, DevBufs B, int list_a, int mode_a, int list_b, int mode_b, int parity,
                                                int rst_list, int32_t* sync, int32_t* hard, int clear_parity) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sok_lds[];
    __shared__ int s_pref_a[WL_NSHARD + 1], s_pref_b[WL_NSHARD + 1];
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x & 63;
    const int n_a = wl_load_prefix(B, parity, list_a, s_pref_a);
    const int n_b = list_b >= 0 ? wl_load_prefix(B, parity, list_b, s_pref_b) : 0;
    const int n = n_a + n_b;
    __shared__ SokDuoBox s_box;
    if (threadIdx.x >= 64) { sok_duo_server(sok_lds, &s_box, lane); return; }
    SokDuoBox* const duo = B.sok_use_lds ? &s_box : nullptr;       // (the heap has to be the LDS one)
    __shared__ SokLevel s_L;             // level + node workspace in LDS: they are indexed dynamically
    __shared__ SokNode s_root, s_work;
    __shared__ int s_spawned, s_fast;
    __shared__ uint8_t s_scr[64];
    __shared__ SokFastNode s_cache[4];
    SokNode* pool = B.sok_pool + (size_t)blockIdx.x * B.sok_pool_stride;
    uint32_t* g_heap = B.sok_use_lds ? nullptr : B.sok_heap + (size_t)blockIdx.x * B.sok_heap_stride;
    uint32_t* g_table = B.sok_use_lds ? nullptr : B.sok_table + (size_t)blockIdx.x * B.sok_table_size;
    const int tsize = B.sok_use_lds ? SOK_LDS_TABLE : B.sok_table_size;
    const int W = P.width, H = P.height;
    const int KS[4] = {-1, 2, 1, 0};
    for (;;) {
        int kind = 0, t = 0;   // 0: nothing to do right now, 1: BFS job t, 2: A* ticket t, 3: leave
        if (lane == 0) {
            int hc = sok_ld(sync + SOK_SY_HARD);
            hc = hc < B.sok_hard_cap ? hc : B.sok_hard_cap;
            int tb = sok_ld(sync + SOK_SY_TICKET_B);
            while (tb < 3 * hc) {
                const int seen = atomicCAS(sync + SOK_SY_TICKET_B, tb, tb + 1);
                if (seen == tb) { kind = 2; t = tb; break; }
                tb = seen;
            }
            if (kind == 0 && sok_ld(sync + SOK_SY_TICKET_A) < n) {
                const int ta = atomicAdd(sync + SOK_SY_TICKET_A, 1);
                if (ta < n) { kind = 1; t = ta; }
            }
            if (kind == 0 && sok_ld(sync + SOK_SY_BFS_DONE) >= n) {
                // every BFS has finished, so the hard list is final (the counter was read after the data it depends on)
                int hf = sok_ld(sync + SOK_SY_HARD);
                hf = hf < B.sok_hard_cap ? hf : B.sok_hard_cap;
                if (sok_ld(sync + SOK_SY_TICKET_B) >= 3 * hf) kind = 3;
            }
        }
        kind = __shfl(kind, 0, 64);
        t = __shfl(t, 0, 64);
        if (kind == 3) break;
        if (kind == 0) { __builtin_amdgcn_s_sleep(127); continue; }

        int e = 0, mode = 0, first = 0, last = 3;
        if (kind == 1) {
            if (t < n_a) { e = wl_get(B, list_a, s_pref_a, t); mode = mode_a; }
            else { e = wl_get(B, list_b, s_pref_b, t - n_a); mode = mode_b; }
        } else {
            int tag = 0;
            if (lane == 0) { while ((tag = sok_ld(hard + t / 3)) == 0) __builtin_amdgcn_s_sleep(8); }
            tag = __shfl(tag, 0, 64);
            e = (tag & 0x0FFFFFFF) - 1;
            mode = (tag >> 28) & 3;
            first = last = 1 + t % 3;
        }
        {   // the level by all 64 lanes (level_build_wave.h: ~40 us on one lane, a few on the wavefront)
            const int ncr = sok_build_level_wave(B.map + (size_t)e * W * H, W, H, s_L, s_root, lane);
            sok_init_deadlocks_wave(s_L, s_scr, lane);
            if (lane == 0) {
                if (ncr > SOK_MAXC) atomicOr(B.status, 1);
                s_root.h = (uint16_t)sok_heuristic(s_L, s_root.crate);
                s_spawned = 0;
                s_fast = (B.sok_use_lds && s_L.nc <= B.sok_fast_maxc) ? 1 : 0;
            }
        }
        __threadfence_block();
        const int fast = s_fast;
        // BFS job: agent 0 and -- only if the level could not be published -- the other agents after it, with the
        // exact exhausted-BFS shortcut.  A* ticket: that one agent.  The search runs on lanes 0..3 (register-resident
        // path: uniform except for the four children of a pop) or on lane 0 (generic path); every lane helps to clear
        // the visited table.
        int dist = 0, sol = 0, go = 1, reported = 0;
        for (int a = first; a <= last && go; a++) {
            if (fast) { for (int i = lane; i < 2 * tsize; i += 64) sok_lds[SOK_LDS_HEAP + i] = 0; }   // 64-bit keys
            else if (B.sok_use_lds) { for (int i = lane; i < tsize; i += 64) sok_lds[SOK_LDS_HEAP + i] = 0; }
            else { for (int i = lane; i < tsize; i += 64) g_table[i] = 0; }
            __threadfence_block();
            if (lane < (fast ? 4 : 1)) {
                int hh = 0, dd = 0, it = 0;
                bool exhausted = false, win = false;
                if (kind == 1 && a == 0) {
                    int sp = P.solver_power < B.sok_spawn_iters ? P.solver_power : B.sok_spawn_iters;
                    SokSpawnHook hook = {sync, hard, sp, (e + 1) | (mode << 28), &s_spawned, lane, B.sok_hard_cap};
                    win = sok_run_agent(B, P.solver_power, s_L, s_work, s_root, pool, sok_lds, s_cache, g_heap, g_table, tsize, fast, -1, hh, dd, it, exhausted, hook, lane);
                    __threadfence_block();
                    if (s_spawned) { if (lane == 0) sok_report(P, B, e, 0, win, hh, dd, exhausted, mode, parity, rst_list); reported = 1; go = 0; }
                    else go = !(win || exhausted);
                } else if (kind == 1) {
                    win = sok_run_agent(B, P.solver_power, s_L, s_work, s_root, pool, sok_lds, s_cache, g_heap, g_table, tsize, fast, KS[a], hh, dd, it, exhausted, SokNoHook(), lane, SOK_LDS_HEAP, duo);
                    go = !win;
                } else {
                    SokPollHook hook = {B.sok_stop + e, 4 - a};
                    if (sok_ld(B.sok_stop + e) < 4 - a) {
                        win = sok_run_agent(B, P.solver_power, s_L, s_work, s_root, pool, sok_lds, s_cache, g_heap, g_table, tsize, fast, KS[a], hh, dd, it, exhausted, hook, lane, SOK_LDS_HEAP, duo);
                    }
                    if (lane == 0) sok_report(P, B, e, a, win, hh, dd, false, mode, parity, rst_list);
                    reported = 1;
                }
                dist = win ? 0 : hh;
                sol = win ? dd : 0;
            }
            go = __shfl(go, 0, 64);
            __threadfence_block();
        }
        if (lane == 0 && kind == 1) {
            if (!reported) {
                int32_t s[PCGRL_MAX_STATS];
                const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
                for (int k = 0; k < 8; k++) s[k] = park[k];
                s[4] = dist; s[5] = sol;
                finalize_item<PCGRL_PROB_SOKOBAN>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), true, rst_list);
            }
            __threadfence();
            atomicAdd(sync + SOK_SY_BFS_DONE, 1);
        }
    }
    s_box.session = 0;          // the heap server leaves with us
    sok_duo_sync();
}
