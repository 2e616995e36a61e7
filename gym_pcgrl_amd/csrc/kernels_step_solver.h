// k_step_solver: pcgrl_rollout for the problems whose statistics need a search (Sokoban, mdungeon, ddave) -- a whole
// tape of steps by persistent blocks, one block per compute unit.  Part of the single translation unit pcgrl_abi.hip.
//
// Stepping these problems one launch sequence at a time makes every step wait for its slowest search: with 131 072
// Sokoban environments three or four levels per step run into the 5 000-pop cap (6.5 ms) while the other 131 000
// environments are done after 0.2 ms.  The environments do not depend on each other, so for a tape of actions there is no
// reason to wait: here a block owns up to 512 environments for the whole tape and goes through the phases of a step
// (update, statistics, resets, searches, the resets and searches those caused) with block barriers only.  A block that
// meets a capped search is late by that search; nobody else is.  The work lists live in LDS (LocalLists behind wl_push);
// the device functions are the ones the step kernels use (update_env, stats_wave_task, wave_reset_env, the searches of
// sokoban_fast.h / sokoban_solver.h, mdungeon_fast.h / mdungeon_solver.h, ddave_solver.h), so the results are the same by
// construction -- and by test (tests/test_gpu_parity.py::test_rollout_equals_steps).
// One full search region (heap + visited table, 142 KB of LDS) per block; the eight wavefronts first try the jobs in small
// private regions carved out of it (the median search is a few dozen pops), the rest is redone by wavefront 0 in the full one.
#pragma once

// list ids reuse the global ones: WL_CHG changed, WL_RST reset before the searches, WL_SOL / WL_SOL2 search jobs of changed /
// regenerated maps, WL_RST2 reset after the searches, WL_SOL3 their jobs
//
// SolverGame<PROB>::run: _run_game of one level by one wavefront with the agents in sequence, inside a given search region
// (`heap`, `table` of `tsize` slots at `heap + table_off` words) and with at most `power` pops per agent.  Returns (on every
// lane) whether the result is final: with power < solver_power an agent that is stopped by the limit makes the whole job
// "not final" -- it is then run again with the full region and the full power.  A search that ends by winning or by running
// out of states before the limit gives what the full search gives (the table size only changes the probe sequences).
#define SS_SMALL_POPS 256
#define SS_SMALL_TABLE 1024
#define SS_SMALL_HEAP 1028                                /* 1 + 4 * SS_SMALL_POPS entries, padded */
#define SS_SMALL_WORDS (SS_SMALL_HEAP + 2 * SS_SMALL_TABLE) /* per wavefront, 32-bit words */
#define SS_SMALL_NODES 1040
#define SS_THREADS 512                                   /* one block per compute unit: eight wavefronts share the phases of a step */
#define SS_SEARCH_WAVES 8                                 /* wavefronts that take search jobs (a small region each) */

template <int PROB>
struct SolverGame;

template <>
struct SolverGame<PCGRL_PROB_SOKOBAN> {
    struct Shared { SokLevel L; SokNode root, work; SokFastNode cache[4]; int fast; uint8_t scratch[64]; };
    static __device__ __forceinline__ bool run(const PcgrlParams& P, const DevBufs& B, int e, Shared& S, uint32_t* heap, int table_off, int tsize, int power,
                                               SokNode* pool, int lane, int32_t* s) {
        {   // the level by all 64 lanes (level_build_wave.h)
            const int ncr = sok_build_level_wave(B.map + (size_t)e * P.width * P.height, P.width, P.height, S.L, S.root, lane);
            sok_init_deadlocks_wave(S.L, S.scratch, lane);
            if (lane == 0) {
                if (ncr > SOK_MAXC) atomicOr(B.status, 1);
                S.root.h = (uint16_t)sok_heuristic(S.L, S.root.crate);
                S.fast = (S.L.nc <= B.sok_fast_maxc) ? 1 : 0;
            }
        }
        __threadfence_block();
        const int fast = S.fast;
        const int KS[4] = {-1, 2, 1, 0};
        int hh = 0, dd = 0, win = 0, final = 1;
        for (int a = 0; a < 4; a++) {
            for (int i = lane; i < (fast ? 2 : 1) * tsize; i += 64) heap[table_off + i] = 0;      // 64-bit keys on the fast path
            __threadfence_block();
            int stop = 0;
            if (lane < (fast ? 4 : 1)) {
                int it = 0; bool exhausted = false;
                const bool w = sok_run_agent(B, power, S.L, S.work, S.root, pool, heap, S.cache, (uint32_t*)nullptr, (uint32_t*)nullptr, tsize, fast,
                                             KS[a], hh, dd, it, exhausted, SokNoHook(), lane, table_off);
                win = w ? 1 : 0;
                if (!w && !exhausted && power < P.solver_power) { final = 0; stop = 1; }        // stopped by the reduced limit
                else stop = (w || (a == 0 && exhausted)) ? 1 : 0;   // sok_run_game: first winner, or the exact exhausted-BFS shortcut
            }
            stop = __shfl(stop, 0, 64);
            __threadfence_block();
            if (stop) break;
        }
        if (lane == 0) { s[4] = win ? 0 : hh; s[5] = win ? dd : 0; }
        return __shfl(final, 0, 64) != 0;
    }
};
template <>
struct SolverGame<PCGRL_PROB_MDUNGEON> {
    struct Shared { MdLevel L; MdNode root, work; MdFastLevel F; MdFastNode cache[4]; int fast; };
    static __device__ __forceinline__ bool run(const PcgrlParams& P, const DevBufs& B, int e, Shared& S, uint32_t* heap, int table_off, int tsize, int power,
                                               SokNode* pool, int lane, int32_t* s) {
        {
            const int nthings = md_build_level_wave(B.map + (size_t)e * P.width * P.height, P.width, P.height, S.L, S.root, S.F, lane);
            if (lane == 0) S.fast = (nthings <= MDF_MAXI && B.sok_fast_maxc >= 0) ? 1 : 0;
        }
        __threadfence_block();
        const int fast = S.fast;
        const int KS[4] = {2, 1, 0, -1};
        int out5[5] = {0, 0, 0, 0, 0}, final = 1;
        for (int a = 0; a < 4; a++) {
            for (int i = lane; i < (fast ? 2 : 1) * tsize; i += 64) heap[table_off + i] = 0;
            __threadfence_block();
            int next = a + 1;
            if (lane < (fast ? 4 : 1)) {
                int it = 0; bool exhausted = false, w;
                if (fast) {
                    uint64_t key = 0; int hh = 0, dd = 0;
                    const MdKidsLanes kids = {lane};
                    w = md_search_fast(S.L, S.F, reinterpret_cast<MdFastNode*>(pool), heap, reinterpret_cast<uint64_t*>(heap + table_off), tsize - 1,
                                       S.cache, S.root, KS[a], power, key, hh, dd, it, exhausted, SokNoHook(), kids);
                    mdf_result(S.F, key, hh, dd, w, out5);
                } else {
                    w = md_search(S.L, reinterpret_cast<MdNode*>(pool), heap, heap + table_off, tsize - 1, S.work, S.root, KS[a], power, it, exhausted);
                    md_result(S.L, S.root, S.work, w, out5);
                }
                if (!w && !exhausted && power < P.solver_power) { final = 0; next = 4; }
                else if (w) next = 4;
                else if (a < 3 && exhausted) next = 3;            // md_run_game: straight to BFS
            }
            next = __shfl(next, 0, 64);
            __threadfence_block();
            a = next - 1;
        }
        if (lane == 0) md_pack(s, out5);
        return __shfl(final, 0, 64) != 0;
    }
};
template <>
struct SolverGame<PCGRL_PROB_DDAVE> {
    struct Shared { DdLevel L; DdNode root, work; DdFastLevel F; DdFastNode cache[4]; int fast; };
    static __device__ __forceinline__ bool run(const PcgrlParams& P, const DevBufs& B, int e, Shared& S, uint32_t* heap, int table_off, int tsize, int power,
                                               SokNode* pool, int lane, int32_t* s) {
        {
            const int nd = dd_build_level_wave(B.map + (size_t)e * P.width * P.height, P.width, P.height, S.L, S.root, S.F, lane);
            if (lane == 0) S.fast = (nd <= DDF_MAXD && B.sok_fast_maxc >= 0) ? 1 : 0;
        }
        __threadfence_block();
        const int fast = S.fast;
        const int KS[4] = {2, 1, 0, -1};
        int out4[4] = {0, 0, 0, 0}, final = 1;
        for (int a = 0; a < 4; a++) {
            for (int i = lane; i < (fast ? 2 : 1) * tsize; i += 64) heap[table_off + i] = 0;
            __threadfence_block();
            int stop = 0;
            if (lane < (fast ? 4 : 1)) {
                int it = 0; bool exhausted = false, w;
                if (fast) {
                    uint64_t key = 0; int hh = 0, dd = 0, jj = 0;
                    const DdKidsLanes kids = {lane};
                    w = dd_search_fast(S.L, S.F, reinterpret_cast<DdFastNode*>(pool), heap, reinterpret_cast<uint64_t*>(heap + table_off), tsize - 1, S.cache,
                                       S.root, KS[a], power, key, hh, dd, jj, it, exhausted, SokNoHook(), kids);
                    ddf_result(S.F, key, hh, dd, jj, w, out4);
                } else {
                    w = dd_search(S.L, reinterpret_cast<DdNode*>(pool), heap, heap + table_off, tsize - 1, S.work, S.root, KS[a], power, it, exhausted,
                                  SokNoHook());
                    dd_result(S.L, S.work, w, out4);
                }
                if (!w && !exhausted && power < P.solver_power) { final = 0; stop = 1; }
                else stop = w ? 1 : 0;
            }
            stop = __shfl(stop, 0, 64);
            __threadfence_block();
            if (stop) break;
        }
        if (lane == 0) dd_pack(s, out4);
        return __shfl(final, 0, 64) != 0;
    }
};

template <int PROB, int REP, class MaskT>
__global__ __launch_bounds__(SS_THREADS) void k_step_solver(PcgrlParams P, DevBufs Bg, const int32_t* __restrict__ actions, int gen_map, int steps,
                                                             size_t action_stride, int envs_per_block, double* reward_out, uint8_t* done_out,
                                                             int32_t* info_out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t ss_lds[];     // heap + visited table of the block's one search at a time
    __shared__ LocalLists s_lists;
    __shared__ typename SolverGame<PROB>::Shared s_game0;      // the full-region search; the small ones keep theirs in the region itself
    __shared__ uint16_t s_big[2 * WL_LOCAL_CAP];    // search jobs that need the full region (at most every WL_SOL + park-list entry)
    __shared__ int s_nbig;
    // the resets (MT ring + tile bytes per wavefront) borrow the search region: resets and searches are separate phases
    constexpr int kMtBytes = PCGRL_MT_N * 4 + 272;
    constexpr int G = 16, GPW = 4;
    DevBufs B = Bg;
    B.local = &s_lists;
    const int W = P.width, H = P.height;
    const int e0 = blockIdx.x * envs_per_block;
    const int ne = (P.num_envs - e0) < envs_per_block ? (P.num_envs - e0) : envs_per_block;
    SokNode* pool = B.sok_pool + (size_t)blockIdx.x * B.sok_pool_stride;
#pragma clang loop unroll(disable)
    for (int t = 0; t < steps; t++) {
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));                 // see k_step: keeps per-lane values from being hoisted across the steps
        const int lane64 = tid & 63, wv = tid >> 6, gw = lane64 / G;
        DevGroup<G, MaskT> g(lane64);
        const MaskT rowmask = row_valid<MaskT>(g.lane, W, H);
        const int32_t* actions_t = actions + (size_t)t * action_stride;
        if (tid < WL_NLIST) s_lists.n[tid] = 0;
        if (tid == 0) s_lists.e0 = e0;
        __syncthreads();
        // ---- Representation.update + bookkeeping; unchanged environments are finished here
        for (int sub = wv * 64; sub < ne; sub += SS_THREADS) {
            const int e = e0 + sub + lane64;
            UpdateOut u = {};
            if (sub + lane64 < ne) u = update_env<REP, MaskT>(P, B, actions_t, e);
            const uint64_t mc = __ballot(u.chg), mr = __ballot(u.rst);
            const uint64_t below = (1ull << lane64) - 1ull;
            int bc = 0, br = 0;
            if (lane64 == 0) { bc = atomicAdd(&s_lists.n[WL_CHG], __popcll(mc)); br = atomicAdd(&s_lists.n[WL_RST], __popcll(mr)); }
            bc = __shfl(bc, 0, 64); br = __shfl(br, 0, 64);
            if (u.chg) s_lists.items[WL_CHG][bc + __popcll(mc & below)] = (uint16_t)(sub + lane64);
            if (u.rst) s_lists.items[WL_RST][br + __popcll(mr & below)] = (uint16_t)(sub + lane64);
        }
        __syncthreads();
        // ---- statistics of the changed maps: finished, or parked for the search (WL_SOL); finished episodes to WL_RST
        {
            const int n = s_lists.n[WL_CHG];
            for (int w0 = wv; w0 * GPW < n; w0 += SS_THREADS / 64) {
                const int item = w0 * GPW + gw;
                const bool have = item < n;
                const int raw = have ? e0 + (int)s_lists.items[WL_CHG][item] : 0;
                stats_wave_task<PROB, G, MaskT>(P, B, g, lane64, gw, false, false, false, false, have, raw, lane64, MODE_STEP, 0, 0, gen_map,
                                                ss_lds, reinterpret_cast<uint8_t*>(ss_lds), rowmask);      // (no in-kernel resets on this path: unused)
            }
        }
        __syncthreads();
        // ---- two rounds of (resets, searches): the episodes that ended before the searches, then the ones a search ended
        for (int round = 0; round < 2; round++) {
            const int rst_list = round == 0 ? WL_RST : WL_RST2, park_list = round == 0 ? WL_SOL2 : WL_SOL3;
            const int nr = s_lists.n[rst_list];
            {
                uint32_t* mt = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(ss_lds) + wv * kMtBytes);
                uint8_t* tiles = reinterpret_cast<uint8_t*>(mt) + PCGRL_MT_N * 4;
                for (int i = wv; i < nr; i += SS_THREADS / 64) {
                    const int e = e0 + (int)s_lists.items[rst_list][i];
                    ResetRows rr;
                    wave_reset_env<PROB>(P, B, e, gen_map, mt, (uint8_t*)nullptr, lane64, 0, lane64 < G ? lane64 : -1, &rr);
                    MaskT b0, b1, b2;
                    reset_rows_to_planes<MaskT>(P, reinterpret_cast<MaskT*>(B.planes) + (size_t)e * P.nplanes * G, lane64 < G ? lane64 : -1, rr.m0, rr.m1, rr.m2, b0, b1, b2);
                    const MaskT valid = (lane64 < G) ? rowmask : (MaskT)0;
                    int32_t st[PCGRL_MAX_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
                    MaskT champ;
                    const bool need_solver = compute_item_stats<PROB>(g, P, b0, b1, b2, valid, st, champ);
                    if (lane64 == 0) finish_or_park<PROB>(P, B, e, st, need_solver, MODE_START, 0, 0, true, park_list);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
            // searches: every wavefront takes jobs and tries them inside a small private region with a small pop limit -- the
            // median search is a few dozen pops -- and the few that do not finish there are redone by wavefront 0 with the
            // full region and the full solver_power
            const int na = round == 0 ? s_lists.n[WL_SOL] : 0, nb = s_lists.n[park_list];
            if (tid == 0) s_nbig = 0;
            __syncthreads();
            const int small_power = P.solver_power < SS_SMALL_POPS ? P.solver_power : SS_SMALL_POPS;
            typedef typename SolverGame<PROB>::Shared GameShared;
            GameShared* small_games = reinterpret_cast<GameShared*>(ss_lds + SS_SEARCH_WAVES * SS_SMALL_WORDS);
            for (int j = wv; j < na + nb && wv < SS_SEARCH_WAVES; j += SS_SEARCH_WAVES) {
                const int mode = j < na ? MODE_STEP : MODE_START;
                const int e = e0 + (int)(j < na ? s_lists.items[WL_SOL][j] : s_lists.items[park_list][j - na]);
                int32_t s[PCGRL_MAX_STATS];
                const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
                if (lane64 == 0) for (int k = 0; k < 8; k++) s[k] = park[k];
                const bool final = SolverGame<PROB>::run(P, B, e, small_games[wv], ss_lds + wv * SS_SMALL_WORDS, SS_SMALL_HEAP, SS_SMALL_TABLE, small_power,
                                                         pool + (size_t)wv * SS_SMALL_NODES, lane64, s);
                if (lane64 == 0) {
                    if (final) finalize_item<PROB>(P, B, e, s, mode, 0, 0, true, WL_RST2);
                    else s_big[atomicAdd(&s_nbig, 1)] = (uint16_t)j;
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            if (wv == 0) {
                const int nbig = s_nbig;
                for (int q = 0; q < nbig; q++) {
                    const int j = s_big[q];
                    const int mode = j < na ? MODE_STEP : MODE_START;
                    const int e = e0 + (int)(j < na ? s_lists.items[WL_SOL][j] : s_lists.items[park_list][j - na]);
                    int32_t s[PCGRL_MAX_STATS];
                    const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
                    if (lane64 == 0) for (int k = 0; k < 8; k++) s[k] = park[k];
                    SolverGame<PROB>::run(P, B, e, s_game0, ss_lds, SOK_LDS_HEAP, SOK_LDS_TABLE, P.solver_power, pool, lane64, s);
                    if (lane64 == 0) finalize_item<PROB>(P, B, e, s, mode, 0, 0, true, WL_RST2);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
        }
        if (reward_out || done_out || info_out) {   // kernel-uniform: the per-step outputs of the block's environments, row t
            const size_t row = (size_t)t * P.num_envs + e0;
            for (int i = tid; i < ne; i += SS_THREADS) {
                if (reward_out) reward_out[row + i] = B.reward[e0 + i];
                if (done_out) done_out[row + i] = B.done[e0 + i];
            }
            if (info_out) for (int i = tid; i < ne * 10; i += SS_THREADS) info_out[row * 10 + i] = B.info[(size_t)e0 * 10 + i];
        }
    }
}
