// k_reset: PcgrlEnv.reset (map generation + start stats), one wavefront per environment.
// Part of the single translation unit pcgrl_abi.hip (see its header comment for the overall picture).
#pragma once
// k_reset: wavefront per environment on the reset list -- PcgrlEnv.reset including the start stats.  Used by
// pcgrl_reset (every environment) and by the Sokoban step (list = WL_RST before the solver kernel, WL_RST2 after
// it; maps that need the solver are parked on park_list); in a step the other problems reset inside k_stats.
template <int PROB, int G, class MaskT>
__global__ __launch_bounds__(PCGRL_BLOCK) void k_reset(PcgrlParams P, DevBufs B, int list, int park_list, int parity, int gen_map, int clear_parity) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int W = P.width, H = P.height, cells = W * H;
    const int tiles_bytes = (cells + 15) & ~15;
    uint32_t* mt = reinterpret_cast<uint32_t*>(smem + (size_t)wv * (PCGRL_MT_N * 4 + tiles_bytes));
    uint8_t* tiles = reinterpret_cast<uint8_t*>(mt + PCGRL_MT_N);
    __shared__ int s_pref[WL_NSHARD + 1];
    const int n = wl_load_prefix(B, parity, list, s_pref);
    for (int item = blockIdx.x * 4 + wv; item < n; item += gridDim.x * 4) {
        const int e = wl_get(B, list, s_pref, item);
        ResetRows rr;
        wave_reset_env<PROB>(P, B, e, gen_map, mt, tiles, lane, 0, lane < G ? lane : -1, &rr);
        MaskT b0, b1, b2;
        reset_rows_to_planes<MaskT>(P, reinterpret_cast<MaskT*>(B.planes) + (size_t)e * P.nplanes * P.group, lane < G ? lane : -1, rr.m0, rr.m1, rr.m2, b0, b1, b2);
        // start stats (pcgrl_env.py:70-71, problem.py:45-46): the rows are already in registers.  With
        // 16-lane groups only the first DPP row holds the map; the other rows see an empty map and idle.
        DevGroup<G, MaskT> g;
        const MaskT valid = (lane < G) ? row_valid<MaskT>(g.lane, W, H) : (MaskT)0;
        int32_t st[PCGRL_MAX_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
        MaskT champ;
        const bool need_solver = compute_item_stats<PROB>(g, P, b0, b1, b2, valid, st, champ);
        if (PROB == PCGRL_PROB_BINARY && B.champ && lane < G) reinterpret_cast<MaskT*>(B.champ)[(size_t)e * G + lane] = champ;
        if (lane == 0) finish_or_park<PROB>(P, B, e, st, need_solver, MODE_START, parity, item & (WL_NSHARD - 1), true, park_list);
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
            if (t >= P.ntiles) { atomicOr(B.status, PCGRL_STATUS_BAD_TILE); t = (uint8_t)(P.ntiles - 1); }   // clamped and reported
            tiles[c] = t;
            B.map[(size_t)e * cells + c] = t;
        }
        __builtin_amdgcn_wave_barrier();
        MaskT b0, b1, b2;
        planes_from_tiles<MaskT>(P, tiles, reinterpret_cast<MaskT*>(B.planes) + (size_t)e * P.nplanes * P.group, lane < P.group ? lane : -1, b0, b1, b2);
        __builtin_amdgcn_wave_barrier();
    }
}
