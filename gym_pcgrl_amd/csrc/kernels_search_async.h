// k_search_async: the search kernel of pcgrl_step_async (round 5) -- Sokoban / MiniDungeons / Dave steps in which NO environment
// waits for another environment's search.  Part of the single translation unit pcgrl_abi.hip.
//
// A lockstep step of 131 072 Sokoban environments is one capped A* search long (5 000 pops, 4 ms) although 99.98 % of the
// environments are done after 0.2 ms: per step about 1 300 levels need the solver at all, their median search is 31 pops, about 28
// take more than 256 and one or two run into the cap (oracle statistics, profiles/r5_round5/NOTES.md).  The environments are
// independent, so the batch does not have to advance as one: pcgrl_step_async is a *tick* --
//   * an environment whose step is complete takes its action from this tick's action array and steps;
//   * every search gets at most `pop_budget` pops per tick.  A search that is not finished by then is SUSPENDED (SokResume,
//     sokoban_fast.h): its scalars, its heap and its visited table go to a slot of a global arena, its environment is marked
//     pending and the tick ends without it;
//   * a pending environment takes no action (the caller's action for it is ignored) until a later tick has finished its step --
//     the suspended searches are the first thing every tick continues, for another `pop_budget` pops each.
// Per environment the sequence (action taken -> observation, reward, done, info) is bitwise the lockstep one; only *when* an
// environment steps differs.  The tick's length is bounded by pop_budget, the stall of a hard level is its own.
//
// The agents of a level run one after the other as in the reference (sokoban_prob.py:104-122, mdungeon_prob.py:110-126,
// ddave_prob.py): nothing is searched speculatively -- in lockstep idle compute units are free, here every pop is throughput.
// A block is the two wavefronts of k_sokoban (search wavefront + heap server); jobs are handed out by tickets (first the
// suspended slots, then the fresh jobs of the work lists) and no block ever waits for another.
#pragma once

#define ASYNC_SLOT_HDR 256                   /* bytes in front of a slot's pool */
struct AsyncSlotHdr {
    int32_t state;                           // 0 free, 1 suspended (runnable), 2 taken (being run or written)
    int32_t env, mode, agent;                // the job: environment, MODE_*, index of the agent that goes on
    int32_t stamp;                           // the tick that suspended it last (a slot is not continued in the launch that wrote it)
    int32_t tsaved;                          // slots of the saved visited table (a power of two, sized by the search's pops so far: async_table_need)
    int32_t pad[2];
    SokResume rs;                            // iterations == 0: agent `agent` starts from the root
};
static_assert(sizeof(AsyncSlotHdr) <= ASYNC_SLOT_HDR, "slot header");
// counters the kernels keep (unsigned 64-bit, pcgrl_async_counters)
#define ASYNC_NSHARD 16                      /* the count of taken actions: sixteen u64 words, 64 bytes apart, behind the eight counters */
#define ASYNC_PEND_SEARCH 1                  /* pending[e]: a search of the environment's step is suspended */
#define ASYNC_PEND_RESET 2                   /*             its search ended the episode: the next tick resets it (and searches the new map) */
enum { ASYNC_ST_CONSUMED = 0,                // actions taken = environment steps started (k_update; kept in the shards, summed by the reader)
       ASYNC_ST_SUSPENDED = 1,               // searches (pieces) that ran out of budget and went to a slot
       ASYNC_ST_LATE = 2,                    // jobs finished from a slot (in a later tick than they started in)
       ASYNC_ST_OVERFLOW = 3,                // no free slot: the search ran to its end in place (the tick was that much longer)
       ASYNC_ST_POPS = 4,                    // pops of the resumable searches
       ASYNC_ST_WORDS = 8 };
struct AsyncCtl {
    uint8_t* pending;                        // [num_envs] 1 = the environment's step is in flight
    unsigned long long* stats;               // [ASYNC_ST_WORDS]
    uint8_t* slots;                          // [nslots][slot_bytes]: header | pool | heap | visited table
    size_t slot_bytes;
    int32_t nslots, nodes_cap, tick, pad;
    int32_t* runlist;                        // [nslots] the runnable slots of this tick (written by k_update / k_async_collect; count in tickets[2])
    // the tick's two launches: the FRESH jobs of the step's lists run in blocks with a small heap and table (a piece is at most
    // ASYNC_SMALL_POPS pops: a few KB of LDS, several blocks per compute unit), the suspended ones in blocks with the full ones
    uint8_t* small_pool;                     // [ASYNC_SMALL_BLOCKS][ASYNC_SMALL_NODES * 16] node pools of the small launch's blocks
    int32_t* overflow;                       // [num_envs] jobs the small launch hands to the full one (count in tickets[3]): env | mode << 28 | reason << 30
};
#define ASYNC_SMALL_POPS 128
#define ASYNC_SMALL_NODES (4 * ASYNC_SMALL_POPS + 8)
#define ASYNC_SMALL_HEAP (4 * ASYNC_SMALL_POPS + 8)          /* words; the table follows */
#define ASYNC_SMALL_TABLE 512                                /* slots (64-bit keys) */
#define ASYNC_SMALL_BLOCKS 1024
__device__ __forceinline__ AsyncSlotHdr* async_hdr(const AsyncCtl& A, int s) { return reinterpret_cast<AsyncSlotHdr*>(A.slots + (size_t)s * A.slot_bytes); }
__device__ __forceinline__ uint8_t* async_pool(const AsyncCtl& A, int s) { return A.slots + (size_t)s * A.slot_bytes + ASYNC_SLOT_HDR; }
__device__ __forceinline__ uint32_t* async_heap(const AsyncCtl& A, int s) { return reinterpret_cast<uint32_t*>(async_pool(A, s) + (size_t)A.nodes_cap * 16); }
__device__ __forceinline__ uint32_t* async_table(const AsyncCtl& A, int s) { return async_heap(A, s) + SOK_LDS_HEAP; }

// The visited table of a piece of a search is as large as the piece needs, not as large as a 5 000-pop search needs: four slots per
// entry it can hold at the end of the piece (n = pops so far + the piece's budget), a power of two between 512 and `tmax`.  A piece of 64
// pops of a young search clears / restores / saves 4-16 KB instead of the 64 KB of the full table; when a search outgrows its table
// the keys are re-hashed into one of twice the size as it is continued (which slot a key sits in depends on the order of insertion,
// what the table answers does not).
__device__ __forceinline__ int async_table_need(int n, int tmax) {
    int t = 512;
    while (t < tmax && t < 4 * n) t <<= 1;
    return t < tmax ? t : tmax;
}
// 16-byte pieces by the 64 lanes of a wavefront (both sides 16-byte aligned, `words` rounded up to four)
__device__ __forceinline__ void async_copy(uint32_t* dst, const uint32_t* src, int words, int lane) {
    const int n4 = (words + 3) >> 2;
    for (int i = lane; i < n4; i += 64) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
}

// One agent of a problem's _run_game, or a piece of it (lanes 0..3 of the search wavefront; the compact searches only).
// res[]: what get_stats takes from the agent (kept from the last agent that ran to its end).
template <int PROB>
struct AsyncGame;
template <>
struct AsyncGame<PCGRL_PROB_SOKOBAN> {
    typedef SolverGame<PCGRL_PROB_SOKOBAN>::Shared Shared;
    // the 64 lanes of the search wavefront (level_build_wave.h); m: the map; scratch: 64 bytes of LDS
    static __device__ __forceinline__ void build(const PcgrlParams& P, const DevBufs& B, const uint8_t* m, Shared& S, uint8_t* scratch, int lane) {
        const int ncr = sok_build_level_wave(m, P.width, P.height, S.L, S.root, lane);
        sok_init_deadlocks_wave(S.L, S.scratch, lane);
        if (lane == 0) {
            if (ncr > SOK_MAXC) atomicOr(B.status, 1);
            S.root.h = (uint16_t)sok_heuristic(S.L, S.root.crate);
            S.fast = (S.L.nc <= B.sok_fast_maxc) ? 1 : 0;
        }
    }
    static __device__ __forceinline__ bool agent(Shared& S, int a, void* pool, uint32_t* lds, int toff, int tsize, SokDuoBox* duo, int power, const SokResumeArg& ra,
                                                 int lane, int* res, bool& exhausted) {
        const int KS[4] = {-1, 2, 1, 0};                                            // BFS, A*(1), A*(0.5), A*(0): sokoban_prob.py:104-122
        const SokKidsLanes kids = {lane, sokf_dir(lane & 3, S.L.w)};
        uint64_t* tab = reinterpret_cast<uint64_t*>(lds + toff);
        int hh = 0, dd = 0, it = 0;
        bool w;
        if (S.L.cells <= 64) w = sok_search_fast<1>(S.L, reinterpret_cast<SokFastNode*>(pool), lds, tab, tsize - 1, S.cache, S.root, KS[a], power, hh, dd, it, exhausted, SokNoHook(), kids, duo, ra);
        else w = sok_search_fast<4>(S.L, reinterpret_cast<SokFastNode*>(pool), lds, tab, tsize - 1, S.cache, S.root, KS[a], power, hh, dd, it, exhausted, SokNoHook(), kids, duo, ra);
        res[0] = w ? 0 : hh; res[1] = w ? dd : 0;
        return w;
    }
    static __device__ __forceinline__ int next(int a, bool win, bool exhausted) { return (win || (a == 0 && exhausted) || a == 3) ? 4 : a + 1; }   // (the exact exhausted-BFS shortcut)
    static __device__ __forceinline__ void pack(int32_t* s, const int* res) { s[4] = res[0]; s[5] = res[1]; }
};
template <>
struct AsyncGame<PCGRL_PROB_MDUNGEON> {
    typedef SolverGame<PCGRL_PROB_MDUNGEON>::Shared Shared;
    static __device__ __forceinline__ void build(const PcgrlParams& P, const DevBufs& B, const uint8_t* m, Shared& S, uint8_t*, int lane) {
        const int n = md_build_level_wave(m, P.width, P.height, S.L, S.root, S.F, lane);
        if (lane == 0) S.fast = (n <= MDF_MAXI && B.sok_fast_maxc >= 0) ? 1 : 0;
    }
    static __device__ __forceinline__ bool agent(Shared& S, int a, void* pool, uint32_t* lds, int toff, int tsize, SokDuoBox* duo, int power, const SokResumeArg& ra,
                                                 int lane, int* res, bool& exhausted) {
        const int KS[4] = {2, 1, 0, -1};                                            // mdungeon_prob.py:110-126
        const MdKidsLanes kids = {lane};
        uint64_t key = 0; int hh = 0, dd = 0, it = 0;
        const bool w = md_search_fast(S.L, S.F, reinterpret_cast<MdFastNode*>(pool), lds, reinterpret_cast<uint64_t*>(lds + toff), tsize - 1, S.cache,
                                      S.root, KS[a], power, key, hh, dd, it, exhausted, SokNoHook(), kids, duo, ra);
        mdf_result(S.F, key, hh, dd, w, res);
        return w;
    }
    static __device__ __forceinline__ int next(int a, bool win, bool exhausted) { return (win || a == 3) ? 4 : ((a < 3 && exhausted) ? 3 : a + 1); }   // md_run_game: straight to BFS
    static __device__ __forceinline__ void pack(int32_t* s, const int* res) { md_pack(s, res); }
};
template <>
struct AsyncGame<PCGRL_PROB_DDAVE> {
    typedef SolverGame<PCGRL_PROB_DDAVE>::Shared Shared;
    static __device__ __forceinline__ void build(const PcgrlParams& P, const DevBufs& B, const uint8_t* m, Shared& S, uint8_t*, int lane) {
        const int n = dd_build_level_wave(m, P.width, P.height, S.L, S.root, S.F, lane);
        if (lane == 0) S.fast = (n <= DDF_MAXD && B.sok_fast_maxc >= 0) ? 1 : 0;
    }
    static __device__ __forceinline__ bool agent(Shared& S, int a, void* pool, uint32_t* lds, int toff, int tsize, SokDuoBox* duo, int power, const SokResumeArg& ra,
                                                 int lane, int* res, bool& exhausted) {
        const int KS[4] = {2, 1, 0, -1};
        const DdKidsLanes kids = {lane};
        uint64_t key = 0; int hh = 0, dd = 0, jj = 0, it = 0;
        const bool w = dd_search_fast(S.L, S.F, reinterpret_cast<DdFastNode*>(pool), lds, reinterpret_cast<uint64_t*>(lds + toff), tsize - 1, S.cache,
                                      S.root, KS[a], power, key, hh, dd, jj, it, exhausted, SokNoHook(), kids, duo, ra);
        ddf_result(S.F, key, hh, dd, jj, w, res);
        return w;
    }
    static __device__ __forceinline__ int next(int a, bool win, bool) { return (win || a == 3) ? 4 : a + 1; }
    static __device__ __forceinline__ void pack(int32_t* s, const int* res) { dd_pack(s, res); }
};

// developer build (tools/probe/async_prof.py, -DPCGRL_ASYNC_PROF): where the search wavefront's time goes, 10 ns ticks summed over all blocks
#if defined(PCGRL_ASYNC_PROF)
#define AP_DECL unsigned long long ap_t = wall_clock64()
#define AP(i) do { const unsigned long long n_ = wall_clock64(); if (lane == 0 && g_tl_buf) atomicAdd(&g_tl_buf[i], n_ - ap_t); ap_t = n_; } while (0)
#define AP_COUNT(i) do { if (lane == 0 && g_tl_buf) atomicAdd(&g_tl_buf[i], 1ull); } while (0)
#else
#define AP_DECL do {} while (0)
#define AP(i) do {} while (0)
#define AP_COUNT(i) do {} while (0)
#endif
// Jobs: the suspended slots (resume != 0), the jobs the small launch handed over (resume != 0), then list_a (mode_a) and list_b (mode_b;
// < 0: none).  `tickets`: [0], [1], [4] counters the host zeroed, [2] the number of runnable slots in A.runlist (k_update), [3] the number
// of handed-over jobs in A.overflow.  budget: pops per job and launch.  toff / tsize: where the visited table starts in the block's
// LDS (words) and its slots -- the full search region of k_sokoban, or (small != 0) the small one of the launch that takes the fresh
// jobs: what does not fit there (a level for the generic search, a suspension that finds no free slot) goes to A.overflow and is run
// to its end by the full launch behind it.  An environment whose episode a finished job ends goes to rst_list (pcgrl_async_flush:
// reset and searched again behind this launch) or, rst_list < 0 (a tick), is marked ASYNC_PEND_RESET: the next tick's k_update
// puts it on its reset list instead of giving it an action -- one launch sequence per tick, not two.
template <int PROB>
__global__ __launch_bounds__(128) void k_search_async(PcgrlParams P, DevBufs B, AsyncCtl A, int list_a, int mode_a, int list_b, int mode_b, int parity,
                                                     int rst_list, int32_t* tickets, int clear_parity, int budget, int resume, int toff, int tsize, int small) {
    typedef AsyncGame<PROB> Game;
    extern __shared__ __attribute__((aligned(16))) uint32_t as_lds[];       // heap | visited table, as in k_sokoban
    __shared__ int s_pref_a[WL_NSHARD + 1], s_pref_b[WL_NSHARD + 1];
    __shared__ SokDuoBox s_box;
    __shared__ typename Game::Shared s_game;
    __shared__ SokResume s_rs;
    __shared__ int s_slot;
    __shared__ uint8_t s_scratch[64];
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x & 63;
    const int n_a = list_a >= 0 ? wl_load_prefix(B, parity, list_a, s_pref_a) : 0;
    const int n_b = list_b >= 0 ? wl_load_prefix(B, parity, list_b, s_pref_b) : 0;
    const int n = n_a + n_b;
    if (threadIdx.x >= 64) { sok_duo_server<true>(as_lds, &s_box, lane); return; }
    if (small && budget > ASYNC_SMALL_POPS) budget = ASYNC_SMALL_POPS;
    uint8_t* const block_pool = small ? A.small_pool + (size_t)blockIdx.x * ASYNC_SMALL_NODES * 16
                                      : reinterpret_cast<uint8_t*>(B.sok_pool + (size_t)blockIdx.x * B.sok_pool_stride);
    AP_DECL;
    AP(0);
    for (;;) {
        // ---- a ticket: a suspended slot, a handed-over job, a fresh job -- or leave
        int slot = -1, e = 0, mode = 0, a0 = 0, kind = 0, unbounded = 0;
        // suspended slots: the update kernel listed the runnable ones (async_list_runnable, worklist.h); a ticket is an entry of that list
        if (lane == 0 && resume) {
            const int nrun = sok_ld(tickets + 2);
            if (sok_ld(tickets + 1) < nrun) {
                const int t = atomicAdd(tickets + 1, 1);
                if (t < nrun) { slot = A.runlist[t]; kind = 1; async_hdr(A, slot)->state = 2; }
            }
            if (kind == 0) {
                const int novf = sok_ld(tickets + 3);
                if (sok_ld(tickets + 4) < novf) {
                    const int t = atomicAdd(tickets + 4, 1);
                    if (t < novf) { kind = 3; e = __hip_atomic_load(A.overflow + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                }
            }
        }
        slot = __shfl(slot, 0, 64);
        if (lane == 0 && kind == 0 && sok_ld(tickets) < n) {
            const int t = atomicAdd(tickets, 1);
            if (t < n) { kind = 2; e = t; }
        }
        kind = __shfl(kind, 0, 64);
        AP(1);
        if (kind == 0) break;
        AP_COUNT(8 + (kind == 3 ? 2 : kind));
        e = __shfl(e, 0, 64);
        int tsm = 0;
        if (kind == 2) {
            const int t = e;
            if (t < n_a) { e = wl_get(B, list_a, s_pref_a, t); mode = mode_a; }
            else { e = wl_get(B, list_b, s_pref_b, t - n_a); mode = mode_b; }
            if (lane == 0) { s_rs = SokResume{}; }
        } else if (kind == 3) {                 // from the small launch: from the root, to the end (the budget was spent there)
            mode = (e >> 28) & 3;
            if (lane == 0) { s_rs = SokResume{}; if (e & (1 << 30)) atomicAdd(A.stats + ASYNC_ST_OVERFLOW, 1ull); }
            e &= 0x0FFFFFFF;
            unbounded = 1;
        } else {
            const AsyncSlotHdr* hd = async_hdr(A, slot);
            e = hd->env; mode = hd->mode; a0 = hd->agent; tsm = hd->tsaved;
            if (lane == 0) s_rs = hd->rs;
        }
        Game::build(P, B, B.map + (size_t)e * P.width * P.height, s_game, s_scratch, lane);
        __threadfence_block();
        AP(2);
        if (!s_game.fast) {
            // a level the compact searches do not take (more than SOKF_MAXC crates ...): the generic search, in one piece (rare) -- in
            // the full search region
            if (small) {
                if (lane == 0) { A.overflow[atomicAdd(tickets + 3, 1)] = e | (mode << 28); A.pending[e] = ASYNC_PEND_SEARCH; }
                __threadfence_block();
                continue;
            }
            int32_t s[PCGRL_MAX_STATS];
            const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
            if (lane == 0) for (int k = 0; k < 8; k++) s[k] = park[k];
            SolverGame<PROB>::run(P, B, e, s_game, as_lds, toff, tsize, P.solver_power, reinterpret_cast<SokNode*>(block_pool), lane, s);
            if (lane == 0) {
                const bool ended = finalize_item<PROB>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), rst_list >= 0, rst_list);
                A.pending[e] = (ended && rst_list < 0) ? ASYNC_PEND_RESET : 0;
                if (slot >= 0) { async_hdr(A, slot)->state = 0; atomicAdd(A.stats + ASYNC_ST_LATE, 1ull); }
            }
            __threadfence_block();
            continue;
        }
        // ---- the agents one after the other, at most `budget` pops in this launch
        void* pool = slot >= 0 ? (void*)async_pool(A, slot) : (void*)block_pool;
        int a = a0, remaining = unbounded ? 0x3FFFFFFF : budget, done = 0, handed = 0;
        int how = (slot >= 0 && s_rs.iterations > 0) ? 1 : 0;          // 0: the agent starts (clear the table), 1: restore from the slot, 2: go on in place
        int res[5] = {0, 0, 0, 0, 0};
        int tcur = tsize;                                              // slots of the table this piece works on (async_table_need)
        for (;;) {
            for (;;) {
                if (how == 1) {
                    async_copy(as_lds, async_heap(A, slot), s_rs.heapn, lane);
                    tcur = async_table_need(s_rs.iterations + (remaining < tsize ? remaining : tsize), tsize);
                    if (tcur < tsm) tcur = tsm < tsize ? tsm : tsize;      // (never smaller than what was saved)
                    if (tsm == tcur) {
                        async_copy(as_lds + toff, async_table(A, slot), 2 * tcur, lane);
                    } else {
                        // the search has outgrown the table it was saved with (or comes from the small launch): its keys go into a
                        // cleared larger one
                        uint4* t4 = reinterpret_cast<uint4*>(as_lds + toff);
                        for (int i = lane; i < tcur / 2; i += 64) t4[i] = make_uint4(0, 0, 0, 0);
                        __threadfence_block();
                        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(async_table(A, slot));
                        unsigned long long* tab = reinterpret_cast<unsigned long long*>(as_lds + toff);
                        for (int i = lane; i < tsm; i += 64) {
                            const unsigned long long key = src[i];
                            if (key == 0ull) continue;
                            uint32_t sl = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (uint32_t)(tcur - 1);
                            while (atomicCAS(tab + sl, 0ull, key) != 0ull) sl = (sl + 1) & (uint32_t)(tcur - 1);
                        }
                    }
                } else if (how == 0) {
                    tcur = async_table_need(remaining < tsize ? remaining : tsize, tsize);
                    uint4* t4 = reinterpret_cast<uint4*>(as_lds + toff);
                    for (int i = lane; i < tcur / 2; i += 64) t4[i] = make_uint4(0, 0, 0, 0);
                }
                __threadfence_block();
                AP(3);
                const int before = s_rs.iterations;
                int win = 0, exh = 0;
                if (lane < 4) {
                    bool ex = false;
                    const SokResumeArg ra = {&s_rs, before + remaining};
                    win = Game::agent(s_game, a, pool, as_lds, toff, tcur, &s_box, P.solver_power, ra, lane, res, ex) ? 1 : 0;
                    exh = ex ? 1 : 0;
                }
                __threadfence_block();
                win = __shfl(win, 0, 64); exh = __shfl(exh, 0, 64);
                AP(4);
                const int used = s_rs.iterations - before;
                remaining -= used;
                if (lane == 0) atomicAdd(A.stats + ASYNC_ST_POPS, (unsigned long long)used);
                if (s_rs.suspended) break;                               // in front of a pop of agent a
                const int nx = Game::next(a, win != 0, exh != 0);
                if (nx >= 4) { done = 1; break; }
                a = nx;
                __threadfence_block();
                if (lane == 0) s_rs = SokResume{};                       // the next agent starts from the root
                __threadfence_block();
                how = 0;
                if (remaining <= 0) break;                               // ... in a later tick
            }
            if (done) break;
            // ---- out of budget: the job goes to (stays in) a slot
            if (slot < 0) {
                if (lane == 0) {
                    int got = -1;
                    const unsigned start = ((unsigned)e * 2654435761u) % (unsigned)A.nslots;
                    for (int k = 0; k < A.nslots && got < 0; k++) {
                        const int sidx = (int)((start + (unsigned)k) % (unsigned)A.nslots);
                        if (sok_ld(&async_hdr(A, sidx)->state) == 0 && atomicCAS(&async_hdr(A, sidx)->state, 0, 2) == 0) got = sidx;
                    }
                    s_slot = got;
                }
                __threadfence_block();
                slot = s_slot;
                if (slot < 0) {
                    // every slot is taken: the search runs to its end inside this tick (the tick is that much longer) -- here, or, from
                    // the small launch, once more from the root in the full one (ASYNC_ST_OVERFLOW is counted there)
                    if (small) {
                        if (lane == 0) { A.overflow[atomicAdd(tickets + 3, 1)] = e | (mode << 28) | (1 << 30); A.pending[e] = ASYNC_PEND_SEARCH; }
                        handed = 1;
                        break;
                    }
                    // (from the root once more when the piece's table is not the full one: at most a budget's pops are redone)
                    if (lane == 0) atomicAdd(A.stats + ASYNC_ST_OVERFLOW, 1ull);
                    remaining = 0x3FFFFFFF;
                    if (tcur != tsize && s_rs.iterations > 0) {
                        __threadfence_block();
                        if (lane == 0) s_rs = SokResume{};
                        __threadfence_block();
                    }
                    how = s_rs.iterations > 0 ? 2 : 0;
                    continue;
                }
                if (s_rs.iterations > 0) async_copy(reinterpret_cast<uint32_t*>(async_pool(A, slot)), reinterpret_cast<const uint32_t*>(block_pool), 4 * s_rs.npool, lane);
            }
            break;
        }
        if (handed) {
            __threadfence_block();
        } else if (done) {
            if (lane == 0) {
                int32_t s[PCGRL_MAX_STATS];
                const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
                for (int k = 0; k < 8; k++) s[k] = park[k];
                Game::pack(s, res);
                // an episode this result ends: with a reset list the caller resets it in this tick (flush), else the next tick does
                const bool ended = finalize_item<PROB>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), rst_list >= 0, rst_list);
                A.pending[e] = (ended && rst_list < 0) ? ASYNC_PEND_RESET : 0;
                if (slot >= 0) { __threadfence(); async_hdr(A, slot)->state = 0; atomicAdd(A.stats + ASYNC_ST_LATE, 1ull); }
            }
        } else {
            if (s_rs.iterations > 0) {
                async_copy(async_heap(A, slot), as_lds, s_rs.heapn, lane);
                async_copy(async_table(A, slot), as_lds + toff, 2 * tcur, lane);
            }
            __threadfence();
            if (lane == 0) {
                AsyncSlotHdr* hd = async_hdr(A, slot);
                hd->env = e; hd->mode = mode; hd->agent = a; hd->stamp = A.tick; hd->rs = s_rs;
                hd->tsaved = s_rs.iterations > 0 ? tcur : 0;
                A.pending[e] = ASYNC_PEND_SEARCH;
                __threadfence();
                __hip_atomic_store(&hd->state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(A.stats + ASYNC_ST_SUSPENDED, 1ull);
            }
        }
        __threadfence_block();
        AP(5);
    }
    AP(6);
    s_box.session = 0;          // the heap server leaves with us
    sok_duo_sync();
}
