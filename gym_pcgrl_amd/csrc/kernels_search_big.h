// k_search_big: the planner jobs that k_stats / k_big / k_reset parked, for levels or solver_power beyond the compact searches
// (search_big.h).  One wavefront per block, a job per wavefront at a time (ticket on the launch's scheduling words); the four
// agents of a level run one after the other, with the exact shortcuts of sokoban_solver.h / mdungeon_solver.h.  Same job lists,
// same park / finalize protocol as k_sokoban / k_mdungeon / k_ddave.  Part of the single translation unit pcgrl_abi.hip.
#pragma once

struct BigSearchArena {          // per-block slices of DevBufs::big_arena (host: big_arena_of)
    uint8_t* base; size_t block_bytes, heap_off, table_off;
    int nodes_cap, tsize;
};

template <int PROB>
__global__ __launch_bounds__(64) void k_search_big(PcgrlParams P, DevBufs B, BigSearchArena A, int list_a, int mode_a, int list_b, int mode_b, int parity,
                                                  int rst_list, int32_t* sync, int clear_parity) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sb_lds[];      // cx, cy: u16 [cells] each; the node workspace's words; scratch
    __shared__ int s_pref_a[WL_NSHARD + 1], s_pref_b[WL_NSHARD + 1];
    __shared__ SokbLevel s_sok;          // (only the problem's own level is used; the others cost 6 KB of LDS per block)
    __shared__ MdbLevel s_md;
    __shared__ DdbLevel s_dd;
    __shared__ SokbNode s_root, s_work;
    __shared__ int s_job[2];
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x;
    const int n_a = wl_load_prefix(B, parity, list_a, s_pref_a);
    const int n_b = list_b >= 0 ? wl_load_prefix(B, parity, list_b, s_pref_b) : 0;
    const int n = n_a + n_b;
    const int W = P.width, H = P.height;
    BigSearchCtx C;
    C.w = W + 2; C.h = H + 2; C.cells = C.w * C.h; C.nwb = (C.cells + 63) >> 6;
    C.cx = reinterpret_cast<uint16_t*>(sb_lds);
    C.cy = C.cx + C.cells;
    uint64_t* lds_words = reinterpret_cast<uint64_t*>(sb_lds + (((size_t)C.cells * 4 + 15) & ~(size_t)15));
    uint64_t* alive = lds_words;                 // nwb words: the node being expanded (mdungeon, ddave)
    uint64_t* used = lds_words + BIG_MAX_WORDS;  // SOKB_MAXC / 64 words (sokoban heuristic)
    uint8_t* blk = A.base + (size_t)blockIdx.x * A.block_bytes;
    C.pool = blk;
    C.heap = reinterpret_cast<uint64_t*>(blk + A.heap_off);
    C.table = reinterpret_cast<uint32_t*>(blk + A.table_off);
    C.nodes_cap = A.nodes_cap; C.table_mask = A.tsize - 1; C.power = P.solver_power;
    for (;;) {
        if (lane == 0) { const int t = atomicAdd(sync + SOK_SY_TICKET_A, 1); s_job[0] = t; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int t = s_job[0];
        if (t >= n) break;
        int e, mode;
        if (t < n_a) { e = wl_get(B, list_a, s_pref_a, t); mode = mode_a; }
        else { e = wl_get(B, list_b, s_pref_b, t - n_a); mode = mode_b; }
        const uint8_t* m = B.map + (size_t)e * W * H;
        MdbWork work; work.alive = alive;
        int nagents = 4;
        if (lane == 0) {
            if (PROB == PCGRL_PROB_SOKOBAN) {
                const int ncr = sokb_build_level(C, m, W, s_sok, s_root);
                if (ncr > SOKB_MAXC) atomicOr(B.status, PCGRL_STATUS_TOO_MANY_CRATES);
                sokb_init_deadlocks(C, s_sok, reinterpret_cast<uint16_t*>(C.heap));       // (the heap is not in use yet: scratch for the corner list)
                s_root.h = (uint16_t)sokb_heuristic(C, s_sok, s_root.crate, used);
#ifdef PCGRL_DBG_BIG
                printf("job e=%d nc=%d ncr=%d player=%d h=%d crate0=%d target0=%d w=%d h=%d nwb=%d cap=%d tmask=%d power=%d\n", e, s_sok.nc, ncr, (int)s_root.player, (int)s_root.h,
                       (int)s_root.crate[0], (int)s_sok.target[0], C.w, C.h, C.nwb, C.nodes_cap, C.table_mask, C.power);
#endif
            } else if (PROB == PCGRL_PROB_MDUNGEON) {
                mdb_build_level(C, m, W, s_md, work);
                mdb_store(C, 0, work);
            } else {
                ddb_build_level(C, m, W, s_dd, work);
                mdb_store(C, 0, work);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the agents in the order of _run_game: sokoban BFS, A*(1), A*(0.5), A*(0) (sokoban_prob.py:104-122); mdungeon / ddave
        // A*(1), A*(0.5), A*(0), BFS (mdungeon_prob.py:110-126, ddave_prob.py:111-127)
        bool win = false;
        int hh = 0, dd = 0, a = 0;
        while (a < nagents && !win) {
            for (int i = lane; i < A.tsize; i += 64) C.table[i] = 0u;
            __threadfence();                         // (global stores of 64 lanes, read by lane 0)
            __builtin_amdgcn_wave_barrier();
            int go_bfs = 0, stop = 0;
            if (lane == 0) {
                int it = 0;
                bool exhausted = false;
                if (PROB == PCGRL_PROB_SOKOBAN) {
                    const int KS[4] = {-1, 2, 1, 0};
                    win = sokb_search(C, s_sok, s_work, s_root, KS[a], used, hh, dd, it, exhausted);
#ifdef PCGRL_DBG_BIG
                    printf("  e=%d agent %d win %d hh %d dd %d it %d exh %d\n", e, a, (int)win, hh, dd, it, (int)exhausted);
#endif
                    // exact shortcut: an exhausted BFS has expanded every reachable state; the A* agents would expand the same states,
                    // find no win and end on a state of minimum heuristic -- which BFS already has (sokoban_solver.h)
                    if (a == 0 && !win && exhausted) stop = 1;
                } else if (PROB == PCGRL_PROB_MDUNGEON) {
                    const int KS[4] = {2, 1, 0, -1};
                    win = mdb_search(C, s_md, work, KS[a], it, exhausted);
                    // exact shortcut: an exhausted A* agent means no agent can win or reach the cap; only the BFS agent's best node matters
                    if (a < 3 && !win && exhausted) go_bfs = 1;
                } else {
                    const int KS[4] = {2, 1, 0, -1};
                    win = ddb_search(C, s_dd, work, KS[a], it, exhausted);
                }
            }
            win = __shfl((int)win, 0, 64) != 0;
            go_bfs = __shfl(go_bfs, 0, 64); stop = __shfl(stop, 0, 64);
            if (stop) break;
            a = go_bfs ? 3 : a + 1;
        }
        if (lane == 0) {
            int32_t s[PCGRL_MAX_STATS];
            const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
            for (int k = 0; k < 8; k++) s[k] = park[k];
            if (PROB == PCGRL_PROB_SOKOBAN) {
                s[4] = win ? 0 : hh; s[5] = win ? dd : 0;
            } else if (PROB == PCGRL_PROB_MDUNGEON) {
                // md_result: what the returned node collected = what lay on the floor at the root and no longer does
                const uint64_t* root_alive = reinterpret_cast<const uint64_t*>(C.pool);
                int pot = 0, ene = 0;
                for (int i = 0; i < C.nwb; i++) {
                    const uint64_t gone = root_alive[i] & ~work.alive[i];
                    pot += md_popcount(gone & s_md.potion[i]);
                    ene += md_popcount(gone & (s_md.goblin[i] | s_md.ogre[i]));
                }
                const int out5[5] = {win ? 0 : (int)work.t.h, win ? (int)work.t.depth : 0, pot, (int)work.t.treasures, ene};
                // (the row keeps the three "collected" counts in a byte each: md_pack)
                if (pot > 255 || out5[3] > 255 || ene > 255) atomicOr(B.status, PCGRL_STATUS_TOO_MANY_CRATES);
                md_pack(s, out5);
            } else {
                const int dia = ddb_diamonds(C, s_dd, work.alive);
                const int out4[4] = {win ? 0 : (int)work.t.h, win ? (int)work.t.depth : 0, (int)work.t.jumps_lo | ((int)work.t.jumps_hi << 8), dia};
                if (dia > 255) atomicOr(B.status, PCGRL_STATUS_TOO_MANY_CRATES);
                dd_pack(s, out4);
            }
            finalize_item<PROB>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), true, rst_list);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
