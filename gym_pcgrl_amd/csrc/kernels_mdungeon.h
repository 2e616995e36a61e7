// k_mdungeon: the planner jobs that k_stats / k_reset parked for the mdungeon problem (mdungeon_solver.h), one
// wavefront per level.  Part of the single translation unit pcgrl_abi.hip.
//
// MDungeonProblem._run_game (mdungeon_prob.py:110-126) runs A*(1), A*(0.5), A*(0) and BFS one after the other and stops
// at the first winner.  A level reaches the planner only when it has one player, one exit and is connected
// (mdungeon_prob.py:152), so the exit is reachable and A*(1) -- manhattan distance to the exit -- wins within a few
// dozen pops unless monsters that cost more health than the player has stand in the way.  The agents therefore run in
// sequence inside one wavefront (lane 0 drives the search; every lane helps to clear the visited table), with the exact
// exhausted-search shortcut of md_run_game.  Jobs are handed out with an atomic ticket on a word the host zeroes
// before the launch; the node pool, and for a large solver_power the heap and table, are the arena the Sokoban
// solver uses.
#pragma once

// Jobs = list_a (mode_a) followed by list_b (mode_b); list_b < 0: none.  Environments that finish their episode here
// go to `rst_list`.
__global__ __launch_bounds__(64) void k_mdungeon(PcgrlParams P, DevBufs B, int list_a, int mode_a, int list_b, int mode_b, int parity,
                                                 int rst_list, int32_t* sync, int clear_parity) {
    extern __shared__ __attribute__((aligned(16))) uint32_t md_lds[];
    __shared__ int s_pref_a[WL_NSHARD + 1], s_pref_b[WL_NSHARD + 1];
    __shared__ MdLevel s_L;              // level + node workspace in LDS: they are indexed dynamically
    __shared__ MdNode s_root, s_work;
    __shared__ int s_go, s_win;
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x;
    const int n_a = wl_load_prefix(B, parity, list_a, s_pref_a);
    const int n_b = list_b >= 0 ? wl_load_prefix(B, parity, list_b, s_pref_b) : 0;
    const int n = n_a + n_b;
    MdNode* pool = reinterpret_cast<MdNode*>(B.sok_pool + (size_t)blockIdx.x * B.sok_pool_stride);
    uint32_t* g_heap = B.sok_use_lds ? nullptr : B.sok_heap + (size_t)blockIdx.x * B.sok_heap_stride;
    uint32_t* g_table = B.sok_use_lds ? nullptr : B.sok_table + (size_t)blockIdx.x * B.sok_table_size;
    const int tsize = B.sok_use_lds ? SOK_LDS_TABLE : B.sok_table_size;
    const int W = P.width, H = P.height;
    const int KS[4] = {2, 1, 0, -1};
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(sync + SOK_SY_TICKET_A, 1);
        t = __shfl(t, 0, 64);
        if (t >= n) break;
        int e, mode;
        if (t < n_a) { e = wl_get(B, list_a, s_pref_a, t); mode = mode_a; }
        else { e = wl_get(B, list_b, s_pref_b, t - n_a); mode = mode_b; }
        if (lane == 0) { md_build_level(B.map + (size_t)e * W * H, W, H, s_L, s_root); s_go = 1; s_win = 0; }
        __threadfence_block();
        for (int a = 0; a < 4; a++) {
            if (!s_go) break;                                     // wave-uniform (LDS word written by lane 0 before the fence)
            if (B.sok_use_lds) { for (int i = lane; i < tsize; i += 64) md_lds[SOK_LDS_HEAP + i] = 0; }
            else { for (int i = lane; i < tsize; i += 64) g_table[i] = 0; }
            __threadfence_block();
            if (lane == 0) {
                int it = 0;
                bool exhausted = false, win;
                if (B.sok_use_lds)   // two instantiations: LDS pointers compile to ds_* instructions
                    win = md_search(s_L, pool, md_lds, md_lds + SOK_LDS_HEAP, tsize - 1, s_work, s_root, KS[a], P.solver_power, it, exhausted);
                else
                    win = md_search(s_L, pool, g_heap, g_table, tsize - 1, s_work, s_root, KS[a], P.solver_power, it, exhausted);
                if (win) { s_win = 1; s_go = 0; }
                else if (a < 3 && exhausted) a = 2;               // exact shortcut (md_run_game): straight to BFS
            }
            a = __shfl(a, 0, 64);
            __threadfence_block();
        }
        if (lane == 0) {
            int out5[5];
            md_result(s_L, s_root, s_work, s_win != 0, out5);
            int32_t s[PCGRL_MAX_STATS];
            const int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
            for (int k = 0; k < 8; k++) s[k] = park[k];
            md_pack(s, out5);
            finalize_item<PCGRL_PROB_MDUNGEON>(P, B, e, s, mode, parity, e & (WL_NSHARD - 1), true, rst_list);
        }
        __threadfence_block();
    }
}
