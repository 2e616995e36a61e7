// k_big: Problem.get_stats + reward / done / info (and the resets) of the changed environments on maps beyond the row-bitboard
// kernels -- one wavefront per work item on multi-word row masks in LDS (bigmap.h).  Takes the place of k_stats, k_stats_wide and
// k_reset for such configurations; same work lists, same finalize / park protocol, same in-kernel reset for the problems without a
// search (binary, zelda).  Part of the single translation unit pcgrl_abi.hip.
#pragma once

// mode MODE_STEP:   items of WL_RST first (with the in-kernel reset: the environments that are certain to be reset -- the statistics
//                   of the map the step ended on, the step's reward / done / info with the counters read before the reset, the
//                   reset, the start statistics), then the items of `list` (statistics; an episode that ends is reset right here
//                   when inline_reset, else pushed on the reset list by finalize_item)
//      MODE_START:  PcgrlEnv.reset of every item of `list` (pcgrl_env.py:66-76), then its start statistics
//      MODE_SETMAP: statistics of every item of `list`
// park_list: where maps that need the planner go (the search problems; finish_or_park).
template <int PROB>
__global__ __launch_bounds__(512) void k_big(PcgrlParams P, DevBufs B, int list, int parity, int mode, int clear_parity, int inline_reset, int gen_map,
                                            int park_list) {
    extern __shared__ __attribute__((aligned(16))) uint8_t big_lds[];
    __shared__ int s_pref[WL_NSHARD + 1], s_pref_rst[WL_NSHARD + 1], s_pref_inc[WL_NSHARD + 1];
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const bool with_rst = mode == MODE_STEP && inline_reset;
    const int n_chg = wl_load_prefix(B, parity, list, s_pref);
    const int n_rst = with_rst ? wl_load_prefix(B, parity, WL_RST, s_pref_rst) : 0;
    // binary: the changes that neither touch nor neighbour the champion component come on WL_INC (k_update) and are answered from
    // the previous statistics (big_incremental) -- after the full items, which are the long ones
    const bool with_inc = PROB == PCGRL_PROB_BINARY && mode == MODE_STEP && B.champ != nullptr;
    const int n_inc = with_inc ? wl_load_prefix(B, parity, WL_INC, s_pref_inc) : 0;
    const int W = P.width, H = P.height;
    const size_t cells = (size_t)W * H;
    const BigGeom G = big_geom(W, H);
    uint8_t* base = big_lds + (size_t)wv * big_wave_lds(W, H);
    uint32_t* mt = reinterpret_cast<uint32_t*>(base);
    uint64_t* ar = reinterpret_cast<uint64_t*>(base + PCGRL_MT_N * 4);
    uint64_t* const champ_l = ar + 6 * G.NW;                      // the champion component in LDS (big_item_stats leaves it there: binary)
    // binary: when a launch has no more full recomputations than blocks (a step: a few dozen among a thousand incremental updates, and
    // it ends with the slowest of them) each is made by a whole block (bigmap_team.h), then the wavefronts go through the
    // incremental items on their own.  A launch full of them (a reset of the batch, a representation without the incremental route)
    // keeps a wavefront per map.
    int first_item = blockIdx.x * nwv + wv, item_stride = gridDim.x * nwv;
    if (PROB == PCGRL_PROB_BINARY && B.big_team && nwv >= 2 && nwv <= BIG_TEAM_MAX_WAVES && n_rst + n_chg <= (int)gridDim.x) {
        __shared__ BigTeamShared s_team;
        __shared__ int s_want;
        const int n_full = n_rst + n_chg;
        uint64_t* const ar0 = reinterpret_cast<uint64_t*>(big_lds + PCGRL_MT_N * 4);
        uint64_t* const ar1 = reinterpret_cast<uint64_t*>(big_lds + big_wave_lds(W, H) + PCGRL_MT_N * 4);
        uint64_t* const lists = reinterpret_cast<uint64_t*>(big_lds);
        uint64_t* const champ0 = ar0 + 6 * G.NW;
        uint32_t* const mt0 = reinterpret_cast<uint32_t*>(big_lds);
        for (int item = blockIdx.x; item < n_full; item += gridDim.x) {
            const bool lone = item < n_rst;
            const int raw = lone ? wl_get(B, WL_RST, s_pref_rst, item) : wl_get(B, list, s_pref, item - n_rst);
            const bool reset_only = (raw & WL_RESET_ONLY) != 0;
            const int e = raw & ~WL_RESET_ONLY;
            const int shard = (item >> 4) & (WL_NSHARD - 1);
            const uint8_t* m = B.map + (size_t)e * cells;
            uint64_t* const champ_g = B.champ ? reinterpret_cast<uint64_t*>(B.champ) + (size_t)e * G.NW : nullptr;
            int32_t s[PCGRL_MAX_STATS];
            for (int k = 0; k < PCGRL_MAX_STATS; k++) s[k] = 0;
            // one pass of the statistics by the whole block; the champion's rows go to memory
            auto team_stats = [&]() {
                int regions, path, has;
                big_team_binary(m, G, ar0, ar1, ar, lists, big_wave_lds(W, H) / 8, PCGRL_MT_N / 2, s_team, wv, nwv, lane, regions, path, has);
                s[0] = regions; s[1] = path; s[2] = has;
                if (champ_g && has) { for (int i = threadIdx.x; i < G.NW; i += blockDim.x) champ_g[i] = champ0[i]; }
            };
            int want = 0;
            if (mode != MODE_STEP) {
                if (mode == MODE_START) {
                    if (wv == 0) { wave_reset_env<PROB>(P, B, e, gen_map, mt0, (uint8_t*)nullptr, lane); __threadfence(); }
                    __syncthreads();
                }
                team_stats();
                if (threadIdx.x == 0) finish_or_park<PROB>(P, B, e, s, false, mode, parity, shard, true, park_list);
                __syncthreads();
                continue;
            }
            if (lone) {
                if (!reset_only) {
                    int2 pre = make_int2(0, 0);
                    if (threadIdx.x == 0) pre = reinterpret_cast<const int2*>(B.counters)[e];
                    team_stats();
                    if (threadIdx.x == 0) finalize_item<PROB>(P, B, e, s, MODE_STEP, parity, shard, false, WL_RST, &pre);
                }
                want = 1;
            } else if (reset_only) {
                want = 1;
            } else {
                team_stats();
                if (threadIdx.x == 0) s_want = finish_or_park<PROB>(P, B, e, s, false, MODE_STEP, parity, shard, !inline_reset, park_list) ? 1 : 0;
                __syncthreads();
                want = s_want;
            }
            if (inline_reset && want) {
                __syncthreads();                       // (the step is finished: its counters have been read)
                if (wv == 0) {
                    const unsigned long long bp_r = BP_NOW();
                    wave_reset_env<PROB>(P, B, e, gen_map, mt0, (uint8_t*)nullptr, lane); __threadfence();
                    BP_ADD(26, BP_NOW() - bp_r); BP_ADD(27, 1);
                }
                __syncthreads();
                team_stats();
                if (threadIdx.x == 0) finish_or_park<PROB>(P, B, e, s, false, MODE_START, parity, shard, true, park_list);
            }
            __syncthreads();
        }
        first_item = n_full + blockIdx.x * nwv + wv;
    }
    const unsigned long long bp_k0 = BP_NOW();
    for (int item = first_item; item < n_rst + n_chg + n_inc; item += item_stride) {
        const bool lone = item < n_rst, inc = item >= n_rst + n_chg;
        const int raw = lone ? wl_get(B, WL_RST, s_pref_rst, item) : (inc ? wl_get(B, WL_INC, s_pref_inc, item - n_rst - n_chg) : wl_get(B, list, s_pref, item - n_rst));
        const bool reset_only = !inc && (raw & WL_RESET_ONLY) != 0;
        const int e = inc ? (raw & WL_INCBIG_ENV_MASK) : (raw & ~WL_RESET_ONLY);
        const int shard = (item >> 4) & (WL_NSHARD - 1);
        const uint8_t* m = B.map + (size_t)e * cells;
        uint64_t* const champ_g = B.champ ? reinterpret_cast<uint64_t*>(B.champ) + (size_t)e * G.NW : nullptr;
        int32_t s[PCGRL_MAX_STATS];
        if (PROB == PCGRL_PROB_BINARY && inc) {
            // one changed cell away from the champion: the new passable set from the byte map, the champion from memory, the
            // components around the cell
            uint64_t *a0 = ar, *a1 = ar + G.NW, *a2 = ar + 2 * G.NW, *a3 = ar + 3 * G.NW, *a4 = ar + 4 * G.NW, *a5 = ar + 5 * G.NW;
            big_planes<1>(m, G, a0, a3, lane);
            for (int i = lane; i < G.NW; i += 64) {
                const int r = big_row(G, i), k = i - r * G.KW;
                a0[i] = ~a0[i] & (k == G.KW - 1 ? G.last : ~0ull);
                a1[i] = 0ull; a2[i] = 0ull;
                champ_l[i] = champ_g[i];
            }
            big_sync();
            const int2 old = *reinterpret_cast<const int2*>(B.stats + (size_t)e * 8);
            int regions, path;
            const bool nc = big_incremental(a0, a1, a2, a3, a4, a5, champ_l, G, lane, (raw >> 15) & 255, (raw >> 23) & 255, raw < 0, old.x, old.y, regions, path);
            if (nc) { for (int i = lane; i < G.NW; i += 64) champ_g[i] = champ_l[i]; }
            for (int k = 0; k < PCGRL_MAX_STATS; k++) s[k] = 0;
            s[0] = regions; s[1] = path; s[2] = 1;
            int want = 0;
            if (lane == 0) want = finalize_item<PROB>(P, B, e, s, MODE_STEP, parity, shard, !inline_reset) ? 1 : 0;
            want = __builtin_amdgcn_readfirstlane(want);
            if (inline_reset && want) {
                __builtin_amdgcn_wave_barrier();
                wave_reset_env<PROB>(P, B, e, gen_map, mt, (uint8_t*)nullptr, lane);
                __threadfence();
                big_item_stats<PROB>(P, B, m, G, ar, lane, s);
                if (s[2]) { for (int i = lane; i < G.NW; i += 64) champ_g[i] = champ_l[i]; }
                if (lane == 0) finalize_item<PROB>(P, B, e, s, MODE_START, parity, shard);
            }
            continue;
        }
        if (mode != MODE_STEP) {
            if (mode == MODE_START) {
                wave_reset_env<PROB>(P, B, e, gen_map, mt, (uint8_t*)nullptr, lane);
                __threadfence();               // the new map is read back from memory below
            }
            const bool ns = big_item_stats<PROB>(P, B, m, G, ar, lane, s);
            if (PROB == PCGRL_PROB_BINARY && champ_g && s[2]) { for (int i = lane; i < G.NW; i += 64) champ_g[i] = champ_l[i]; }
            if (lane == 0) finish_or_park<PROB>(P, B, e, s, ns, mode, parity, shard, true, park_list);
            continue;
        }
        int want = 0;
        if (lone) {
            // certain reset (or an unchanged environment whose episode ended): the step is finished with the counters read
            // before the reset zeroes them
            int2 pre = make_int2(0, 0);
            if (!reset_only) {
                if (lane == 0) pre = reinterpret_cast<const int2*>(B.counters)[e];
                big_item_stats<PROB>(P, B, m, G, ar, lane, s);
                if (lane == 0) finalize_item<PROB>(P, B, e, s, MODE_STEP, parity, shard, false, WL_RST, &pre);
            }
            want = 1;
        } else if (reset_only) {
            want = 1;                  // (an unchanged environment whose episode ended: k_update finished its step)
        } else {
            const bool ns = big_item_stats<PROB>(P, B, m, G, ar, lane, s);
            if (PROB == PCGRL_PROB_BINARY && champ_g && s[2]) { for (int i = lane; i < G.NW; i += 64) champ_g[i] = champ_l[i]; }
            if (lane == 0) want = finish_or_park<PROB>(P, B, e, s, ns, MODE_STEP, parity, shard, !inline_reset, park_list) ? 1 : 0;
            want = __builtin_amdgcn_readfirstlane(want);
        }
        if (inline_reset && want) {
            __builtin_amdgcn_wave_barrier();
            wave_reset_env<PROB>(P, B, e, gen_map, mt, (uint8_t*)nullptr, lane);
            __threadfence();
            const bool ns = big_item_stats<PROB>(P, B, m, G, ar, lane, s);
            if (PROB == PCGRL_PROB_BINARY && champ_g && s[2]) { for (int i = lane; i < G.NW; i += 64) champ_g[i] = champ_l[i]; }
            if (lane == 0) finish_or_park<PROB>(P, B, e, s, ns, MODE_START, parity, shard, true, park_list);
        }
    }
    if (wv == 0 && blockIdx.x < 4) BP_ADD(28 + blockIdx.x, BP_NOW() - bp_k0);
}

// pcgrl_set_maps on such maps: the byte maps are the whole state (no planes to rebuild).  Tile ids beyond the problem's are
// clamped and reported, as k_planes_from_map does.
template <int>
__global__ __launch_bounds__(256) void k_copy_map(PcgrlParams P, DevBufs B, const uint8_t* __restrict__ src) {
    const size_t total = (size_t)P.num_envs * P.width * P.height;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        uint8_t t = src[i];
        if (t >= P.ntiles) { atomicOr(B.status, PCGRL_STATUS_BAD_TILE); t = (uint8_t)(P.ntiles - 1); }
        B.map[i] = t;
    }
}
