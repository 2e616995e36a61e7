// k_stats / k_sokoban: Problem.get_stats + reward/done/info for the changed environments.
// Part of the single translation unit pcgrl_abi.hip (see its header comment for the overall picture).
#pragma once
// ------------------------------------------------------------------------------------------
// k_stats: lane group per work item
template <class MaskT>
__device__ __forceinline__ MaskT row_valid(int lane, int W, int H) {
    MaskT full = (W >= (int)(8 * sizeof(MaskT))) ? ~(MaskT)0 : (((MaskT)1 << W) - 1);
    return lane < H ? full : (MaskT)0;
}

// Returns true when the environment finished its episode and has to be reset; with push_reset the
// environment is also appended to the reset list (consumed by k_reset), otherwise the caller resets it.
// `pre` (STEP): iteration/changes read before the environment's counters were reset (k_stats resets an environment that
// is certain to end its episode *before* it finishes the step).
// PROB: the problem when the caller is compiled for one (reward and episode-end code of the others drops out), else -1.
template <int PROB = -1>
__device__ __forceinline__ bool finalize_item(const PcgrlParams& P, const DevBufs& B, int e, const int32_t* s,
                                              int mode, int parity, int shard, bool push_reset = true, int rst_list = WL_RST,
                                              const int2* pre = nullptr, const double* reward_pre = nullptr) {
    const int prob = PROB >= 0 ? PROB : P.prob;
    int32_t* st = B.stats + (size_t)e * 8;
    int32_t* start = B.start_stats + (size_t)e * 8;
    if (mode == MODE_STEP) {
        int32_t old[PCGRL_MAX_STATS], sv[PCGRL_MAX_STATS];
        for (int k = 0; k < 8; k++) { old[k] = st[k]; sv[k] = start[k]; }
        const int2 c = pre ? *pre : reinterpret_cast<const int2*>(B.counters)[e];
        const double r = reward_pre ? *reward_pre : compute_reward(P, s, old, prob);      // (reward_pre: zelda_reward_lanes below)
        const bool d = episode_over(P, s, sv, prob) || c.y >= P.max_changes || c.x >= P.max_iterations;
        B.reward[e] = r;
        B.done[e] = d ? 1 : 0;
        episode_account(B, e, r, d);
        int32_t* inf = B.info + (size_t)e * 10;
        // (binary: slots 2 and 3 of the stats row are the library's own -- champion flag, bound on the other components; the info row has
        //  path-imp (binary_prob.py:137) and zero there.  One store per slot: an extra one cost k_stats_wide three registers and with
        //  them its second block per CU)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            st[k] = s[k];
            inf[k] = (prob == PCGRL_PROB_BINARY && k == 2) ? s[1] - sv[1] : ((prob == PCGRL_PROB_BINARY && k == 3) ? 0 : s[k]);
        }
        inf[8] = c.x; inf[9] = c.y;
        if (d && P.auto_reset && push_reset) wl_push(B, parity, rst_list, shard, e);
        return d && P.auto_reset;
    } else {
        for (int k = 0; k < 8; k++) st[k] = s[k];
        if (mode == MODE_START)
            for (int k = 0; k < 8; k++) start[k] = s[k];
    }
    return false;
}

// Problem.get_stats on the row masks of one map (b0..b2 = bit planes of the tile id).  Returns true
// when the Sokoban solver has to finish the job.
// Binary: `champ` receives the rows of the champion component (pcgrl_algos.h) and s[2] says whether there is one --
// a slot of the stats row the binary problem does not use otherwise; k_update reads it to route the next change.
template <int PROB, class G, class MaskT>
__device__ __forceinline__ bool compute_item_stats(G& g, const PcgrlParams& P, MaskT b0, MaskT b1, MaskT b2, MaskT valid, int32_t* s,
                                                   MaskT& champ, bool tight = true) {
    champ = 0;
    if (PROB == PCGRL_PROB_BINARY) {
        int regions, path;
        if (G::kGroup == 16) {
            int ub2;
            regions_and_longest_path(g, (MaskT)(~b0 & valid), regions, path, champ, ub2, tight);
            s[3] = ub2 + 1;      // bound on the components other than the champion, + 1 (0 = not known): binary_touch
        } else {
            regions_and_longest_path(g, (MaskT)(~b0 & valid), regions, path, champ);
        }
        s[0] = regions; s[1] = path; s[2] = g.any(champ) ? 1 : 0;
        return false;
    }
    if (PROB == PCGRL_PROB_ZELDA) { zelda_stats(g, P, b0, b1, b2, valid, s); return false; }
    if (PROB == PCGRL_PROB_MDUNGEON) return mdungeon_stats(g, P, b0, b1, b2, valid, s);
    if (PROB == PCGRL_PROB_DDAVE) return ddave_stats(g, P, b0, b1, b2, valid, s);
    if (PROB == PCGRL_PROB_SMB) return true;      // no bit planes: everything is computed by k_smb (kernels_smb.h) on the byte map
    return sokoban_stats(g, P, b0, b1, b2, valid, s);
}
template <int PROB, class G, class MaskT>
__device__ __forceinline__ bool compute_item_stats(G& g, const PcgrlParams& P, MaskT b0, MaskT b1, MaskT b2, MaskT valid, int32_t* s) {
    MaskT champ;
    return compute_item_stats<PROB>(g, P, b0, b1, b2, valid, s, champ);
}

// Lane 0 of the group: hand the item to the solver or finish it.
template <int PROB = -1>
__device__ __forceinline__ bool finish_or_park(const PcgrlParams& P, const DevBufs& B, int e, const int32_t* s, bool need_solver,
                                               int mode, int parity, int shard, bool push_reset = true, int park_list = -1,
                                               const double* reward_pre = nullptr) {
    if (need_solver) {
        // park the partial stats and hand the environment to the solver kernel.  STEP: in the info
        // row (the old stats are still needed for the reward); otherwise in the stats row itself
        // (the info row keeps the terminal info of an environment that is being reset).
        int32_t* park = (mode == MODE_STEP) ? B.info + (size_t)e * 10 : B.stats + (size_t)e * 8;
        for (int k = 0; k < 8; k++) park[k] = s[k];
        wl_push(B, parity, park_list >= 0 ? park_list : (mode == MODE_STEP ? WL_SOL : WL_SOL2), shard, e);
        return false;
    }
    return finalize_item<PROB>(P, B, e, s, mode, parity, shard, push_reset, WL_RST, nullptr, reward_pre);
}

// The zelda reward (zelda_prob.py:124-142: seven range rewards times their weights, summed left to right in fp64) by the sixteen lanes
// of a group instead of its lane 0: lane k evaluates term k -- one range_reward_i and one product for all seven at once where lane 0
// alone went through them one after the other (a third of k_step's vector instructions on C3) -- and the products are added up in the
// reference's order as they are moved down to lane 0.  Every lane of the group calls it; lane 0 has the result.  `old`: the previous
// statistics row (B.stats + e * 8: read by lanes 0..6); s: the new one (the same in every lane).
template <int J>
__device__ __forceinline__ double dpp_down(double v) {      // lane i receives lane i + J of its row (row_shl:J)
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x100 + J, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x100 + J, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// (the weights and bands come from a table in LDS, zelda_reward_tab: selecting them per lane from the by-value parameter block would be a
//  run-time index into it, which puts a copy of it into scratch memory)
__device__ __forceinline__ void zelda_reward_tab(const PcgrlParams& P, ZeldaRewardTab* T) {      // one thread
    // term k of get_reward: player, key, door, enemies (weight 4), regions (weight 3), nearest-enemy, path-length
    T->w[0] = P.rewards[0]; T->w[1] = P.rewards[1]; T->w[2] = P.rewards[2]; T->w[3] = P.rewards[4]; T->w[4] = P.rewards[3];
    T->w[5] = P.rewards[5]; T->w[6] = P.rewards[6]; T->w[7] = 0.0;
    T->lo[0] = 1; T->lo[1] = 1; T->lo[2] = 1; T->lo[3] = 2; T->lo[4] = 1; T->lo[5] = P.target_enemy_dist; T->lo[6] = PCGRL_IPOS; T->lo[7] = 0;
    T->hi[0] = 1; T->hi[1] = 1; T->hi[2] = 1; T->hi[3] = P.max_enemies; T->hi[4] = 1; T->hi[5] = PCGRL_IPOS; T->hi[6] = PCGRL_IPOS; T->hi[7] = 0;
}
__device__ __forceinline__ double zelda_reward_lanes(const ZeldaRewardTab* T, const int32_t* old, const int32_t* s, int k) {
    const int s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3], s4 = s[4], s5 = s[5], s6 = s[6];
    int n = s6;
    n = k == 5 ? s5 : n; n = k == 4 ? s4 : n; n = k == 3 ? s3 : n; n = k == 2 ? s2 : n; n = k == 1 ? s1 : n; n = k == 0 ? s0 : n;
    const int kk = k & 7;
    const int o = old[kk];
    const double t = k < 7 ? (double)range_reward_i(n, o, T->lo[kk], T->hi[kk]) * T->w[kk] : 0.0;
    double r = t;
    r = r + dpp_down<1>(t); r = r + dpp_down<2>(t); r = r + dpp_down<3>(t);
    r = r + dpp_down<4>(t); r = r + dpp_down<5>(t); r = r + dpp_down<6>(t);
    return r;
}

// What one wavefront does with its share of the work of a launch (k_stats, and the fused step kernel k_step): `lone` -- a
// certain reset (two when `pair`), an even lane group for the map the step ended on and the odd one next to it for the
// regenerated map; `inc` -- incremental items; else full recomputations.  `have` / `raw`: this lane group's item.
// FUSED (compile time): the caller is the fused step kernel (SL is given).
template <int PROB, int G, class MaskT, bool FUSED = false>
__device__ __forceinline__ void stats_wave_task(const PcgrlParams& P, const DevBufs& B, DevGroup<G, MaskT>& g, int lane64, int gw, bool lone,
                                                bool inc, bool pair, bool zinc, bool have, int raw, int shard, int mode, int parity,
                                                int inline_reset, int gen_map, uint32_t* mt, uint8_t* tiles, MaskT rowmask,
                                                StepLocal* SL = nullptr, bool step_obs = false) {
    // (the problems with a search kernel never reset inside the statistics kernel -- their resets wait for the search -- so the reset
    //  code is not compiled into their instantiations: it cost k_stats<sokoban> its registers, 68 bytes of scratch per lane)
    constexpr bool kCanReset = PROB == PCGRL_PROB_BINARY || PROB == PCGRL_PROB_ZELDA;
    if (!kCanReset) inline_reset = 0;
    constexpr int GPW = 64 / G;
    constexpr bool kInc = PROB == PCGRL_PROB_BINARY;
    constexpr bool kZinc = PROB == PCGRL_PROB_ZELDA && G == 16 && sizeof(MaskT) == 4;
    constexpr int NPL = (PROB == PCGRL_PROB_BINARY) ? 1 : 3;
    MaskT* champ_base = reinterpret_cast<MaskT*>(B.champ);
    // (binary, full list: a negative item is a packed one too -- a change in or next to the champion, k_step)
    const bool tch = kInc && G == 16 && !inc && !lone && have && raw < 0;
    const bool packed = inc || (kZinc && zinc && !lone) || tch;        // (environment, cell, passability change) in one word
    const bool reset_only = have && !packed && (raw & WL_RESET_ONLY) != 0;
    const bool compute = have && !reset_only && !(lone && (gw & 1));
    const int e = packed ? wl_inc_env<G>(raw) : (raw & ~WL_RESET_ONLY);
    MaskT* planes_e = reinterpret_cast<MaskT*>(B.planes) + (size_t)e * NPL * G;
    MaskT b0 = 0, b1 = 0, b2 = 0;
    if (compute) {
        b0 = planes_e[g.lane * NPL];
        if (NPL > 1) { b1 = planes_e[g.lane * NPL + 1]; b2 = planes_e[g.lane * NPL + 2]; }
    }
    if (GPW >= 2 && lone && inline_reset) {
        // Certain resets.  The new map does not depend on the statistics of the old one, so the environment is reset
        // first and then both statistics -- of the map the step ended on (rows already in registers, even group) and
        // of the regenerated one (odd group) -- are computed side by side: the chain is one statistics computation
        // long instead of two.  The step is finished with the counters read before the reset zeroed them.
        const int role = gw & 1;
        int2 pre = make_int2(0, 0);
        if (g.lane == 0 && role == 0 && have) pre = reinterpret_cast<const int2*>(B.counters)[e];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // old planes and counters are in registers
#pragma unroll
        for (int k = 0; k < GPW / 2; k++) {
            if (k > 0 && !pair) break;                        // wave-uniform
            if (!__builtin_amdgcn_readlane((int)have, 2 * k * G)) continue;
            const int ek = __builtin_amdgcn_readlane(e, 2 * k * G);
            // k_step, narrow representation: the cursor move of this step has not been drawn (the block's update leaves it to the reset)
            const int step_draws = (SL && P.rep == PCGRL_REP_NARROW && P.random_tile) ? 1 : 0;
            if (SL && lane64 == 0) SL->dirty[ek - SL->e0] = 1;
            ResetRows rr;
            wave_reset_env<PROB, !FUSED>(P, B, ek, gen_map, mt, (uint8_t*)nullptr, lane64, step_draws, gw == 2 * k + 1 ? g.lane : -1, &rr);
            MaskT t0, t1, t2;
            reset_rows_to_planes<MaskT>(P, reinterpret_cast<MaskT*>(B.planes) + (size_t)ek * NPL * G, gw == 2 * k + 1 ? g.lane : -1, rr.m0, rr.m1, rr.m2, t0, t1, t2);
            if (gw == 2 * k + 1) { b0 = t0; b1 = t1; b2 = t2; }
            __builtin_amdgcn_wave_barrier();
        }
        TL(9);
        const bool act = have && (role == 1 || !reset_only);
        int32_t sl[PCGRL_MAX_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
        MaskT champ_l = 0;
        bool ns = false;
        if (act) ns = compute_item_stats<PROB>(g, P, b0, b1, b2, rowmask, sl, champ_l, (B.step_tight & 2) != 0);
        TL(10);
        constexpr bool kLaneReward = PROB == PCGRL_PROB_ZELDA && G == 16;
        double rpre = 0.0;
        if (kLaneReward && SL && role == 0 && act) rpre = zelda_reward_lanes(&SL->zr, B.stats + (size_t)e * 8, sl, g.lane);
        if (g.lane == 0 && role == 0 && act) finalize_item<PROB>(P, B, e, sl, MODE_STEP, parity, shard, false, WL_RST, &pre, (kLaneReward && SL) ? &rpre : nullptr);
        __builtin_amdgcn_wave_barrier();
        if (role == 1 && have) {
            if (kInc && champ_base) champ_base[(size_t)e * G + g.lane] = champ_l;
            if (g.lane == 0) finish_or_park<PROB>(P, B, e, sl, ns, MODE_START, parity, shard);
        }
        if (FUSED && step_obs) {
            // k_step with a bound observation: the image of the regenerated map goes out now, from this wavefront (the rows and the
            // cursor of the new episode are in the block's LDS copy) -- a store stream under the tasks that are still running,
            // instead of at the end of the launch behind everything
#pragma unroll
            for (int k = 0; k < GPW / 2; k++) {
                if (k > 0 && !pair) break;                        // wave-uniform
                if (!__builtin_amdgcn_readlane((int)have, 2 * k * G)) continue;
                const ObsPlanes<MaskT, NPL> src = {reinterpret_cast<const MaskT*>(B.planes), G};        // (k_step: B's per-environment arrays are the block's LDS copy)
                obs_write_env_lean(src, obs_view_from_lds(&SL->obs_v), B.pos, __builtin_amdgcn_readlane(e, 2 * k * G), lane64);
            }
        }
        return;
    }
    int32_t s[PCGRL_MAX_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool need_solver = false;
    MaskT champ = 0;
    if (kInc && inc) {
        if (compute) {      // one cell changed away from the champion: update the previous answer
            const MaskT cbit = (g.lane == wl_inc_row<G>(raw)) ? (MaskT)1 << wl_inc_col<G>(raw) : (MaskT)0;
            const MaskT champ_old = champ_base[(size_t)e * G + g.lane];
            const int4 old = *reinterpret_cast<const int4*>(B.stats + (size_t)e * 8);
            int regions, path, ub2;
            binary_incremental(g, (MaskT)(~b0 & rowmask), cbit, wl_inc_code<G>(raw) != 0, old.x, old.y, champ_old, old.w - 1, regions, path, champ, ub2);
            s[0] = regions; s[1] = path; s[2] = 1; s[3] = ub2 + 1;
        }
    } else if (kZinc && packed) {
        if (compute) {      // zelda: one cell was written; keep or update the region count
            const MaskT cbit = (g.lane == wl_inc_row<G>(raw)) ? (MaskT)1 << wl_inc_col<G>(raw) : (MaskT)0;
            zelda_stats(g, P, b0, b1, b2, rowmask, s, (int)wl_inc_code<G>(raw), cbit, B.stats[(size_t)e * 8 + 4]);
        }
    } else {
        bool full = compute;
        if (kInc && G == 16) {
            const bool t = compute && tch;
            if (t) {       // the change is in the champion or next to it: its pieces / its union with the cell, one double sweep
                const MaskT cbit = (g.lane == wl_inc_row<G>(raw)) ? (MaskT)1 << wl_inc_col<G>(raw) : (MaskT)0;
                const MaskT champ_old = champ_base[(size_t)e * G + g.lane];
                const int4 old = *reinterpret_cast<const int4*>(B.stats + (size_t)e * 8);
                int regions, path, ub2;
                if (binary_touch(g, (MaskT)(~b0 & rowmask), cbit, (wl_inc_code<G>(raw) & 1u) != 0, old.x, champ_old, old.w - 1, regions, path, champ, ub2)) {
                    s[0] = regions; s[1] = path; s[2] = 1; s[3] = ub2 + 1;
                    full = false;
                }
            }
        }
        if (full) need_solver = compute_item_stats<PROB>(g, P, b0, b1, b2, rowmask, s, champ, (B.step_tight & 1) != 0);      // (also what binary_touch gave up on)
    }
    if (kInc && compute && champ_base) champ_base[(size_t)e * G + g.lane] = champ;
    TL(10);
    int want_reset = 0;
    constexpr bool kLaneReward = PROB == PCGRL_PROB_ZELDA && G == 16;
    double rpre = 0.0;
    const bool lane_reward = kLaneReward && SL != nullptr && mode == MODE_STEP;      // (k_step: the table is in its StepLocal)
    if (lane_reward && have && !reset_only) rpre = zelda_reward_lanes(&SL->zr, B.stats + (size_t)e * 8, s, g.lane);
    if (g.lane == 0 && have) {
        if (reset_only) want_reset = 1;
        else want_reset = finish_or_park<PROB>(P, B, e, s, need_solver, mode, parity, shard, !inline_reset, -1, lane_reward ? &rpre : nullptr) ? 1 : 0;
    }
    if (inline_reset) {
        const uint64_t want = __ballot(want_reset != 0);     // one bit per group, at its lane 0
        TL(11);
        if (want) {
            TL(12);
            // k_step: wavefront 0 writes the draws of this step into the rings (and its byte-map / heatmap writes land) behind
            // the barrier; a reset that nobody saw coming waits for that (it has long happened by now)
            if (SL) { while (__hip_atomic_load(&SL->refill_done[SL->par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < SL->need) __builtin_amdgcn_s_sleep(2); }
            bool mine = false;
#pragma unroll
            for (int k = 0; k < GPW; k++) {
                if ((want >> (k * G)) & 1ull) {               // wave-uniform
                    const int ek = __builtin_amdgcn_readlane(e, k * G);
                    if (SL && lane64 == 0) { SL->dirty[ek - SL->e0] = 1; SL->late[ek - SL->e0] = 1; SL->n_late = 1; }
                    int pend = 0;
                    if (SL) {
                        // k_step: the draws of this step may still be in the environment's draw cache, not in its ring -- take them
                        // (the staged ring is patched), or wait for the update wavefront that already has (StepLocal::pend)
                        if (lane64 == 0) {
                            pend = atomicExch(&SL->pend[ek - SL->e0], 0);
                            if (pend < 0) {
                                while (__hip_atomic_load(&SL->late_done[(ek - SL->e0) >> 6], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(2);
                                pend = 0;
                            }
                        }
                        pend = __builtin_amdgcn_readfirstlane(pend);
                    }
                    ResetRows rr;
                    wave_reset_env<PROB, !FUSED>(P, B, ek, gen_map, mt, (uint8_t*)nullptr, lane64, 0, gw == k ? g.lane : -1, &rr, pend);
                    MaskT t0, t1, t2;
                    reset_rows_to_planes<MaskT>(P, reinterpret_cast<MaskT*>(B.planes) + (size_t)ek * NPL * G, gw == k ? g.lane : -1, rr.m0, rr.m1, rr.m2, t0, t1, t2);
                    if (gw == k) { b0 = t0; b1 = t1; b2 = t2; mine = true; }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // start stats of the regenerated maps (pcgrl_env.py:70-71, problem.py:45-46)
            if (mine) {
                int32_t st[PCGRL_MAX_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
                const bool ns = compute_item_stats<PROB>(g, P, b0, b1, b2, rowmask, st, champ, (B.step_tight & 2) != 0);
                if (kInc && champ_base) champ_base[(size_t)e * G + g.lane] = champ;
                if (g.lane == 0) finish_or_park<PROB>(P, B, e, st, ns, MODE_START, parity, shard);
            }
        }
    }
}

// inline_reset (kernel-uniform, STEP mode, every problem but Sokoban): an environment whose episode ended is
// reset right here by the wavefront that found out -- wave_reset_env with all 64 lanes, then the start
// stats on the regenerated rows -- instead of going through a reset list and another latency-bound launch.
// The item loop is therefore wave-uniform: all groups of a wavefront iterate together, a group without
// an item computes on an empty map.
template <int PROB, int G, class MaskT>
__global__ __launch_bounds__(PCGRL_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_stats(PcgrlParams P, DevBufs B, int list, int parity, int mode, int clear_parity,
                                                        int inline_reset, int gen_map, int lone0) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // inline_reset: per wave MT ring + tile bytes
    __shared__ int s_pref[WL_NSHARD + 1], s_pref_inc[WL_NSHARD + 1], s_pref_rst[WL_NSHARD + 1];
    // the last kernel of a step zeroes the *other* parity's work-list counters for the next step
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    DevGroup<G, MaskT> g;
    constexpr int GPB = PCGRL_BLOCK / G, GPW = 64 / G;
    // Binary in a step: after the items of the changed list (padded to whole wavefronts) come the items of the
    // incremental list -- a wavefront works on one kind only.  (Maps taller than 16 rows: `list` < 0, the full
    // recomputations run in k_stats_wide and this launch only has the incremental items.)
    constexpr bool kInc = PROB == PCGRL_PROB_BINARY;
    // Zelda (same restriction on the map size): every changed item of a step carries the cell and what happened to its
    // passability; the region count is updated (items of `list`) or simply kept (items of WL_INC).
    constexpr bool kZinc = PROB == PCGRL_PROB_ZELDA && G == 16 && sizeof(MaskT) == 4;
    const bool zinc = kZinc && mode == MODE_STEP && B.zelda_inc;
    const bool with_inc = (kInc && mode == MODE_STEP && B.champ != nullptr) || zinc;
    // lone0 = 1: the certain resets are shard 0 of `list` (bucketed list of the binary problem); 2: they are the list WL_RST
    int n_inc = 0, n_rst = 0;
    const int n_full = (with_inc || lone0 == 2 || list < 0) ? wl_load_prefix3(B, parity, list, with_inc ? WL_INC : -1, lone0 == 2 ? WL_RST : -1,
                                                                              s_pref, s_pref_inc, s_pref_rst, &n_inc, &n_rst)
                                                             : wl_load_prefix(B, parity, list, s_pref);
    // lone0: shard 0 of the list holds the environments that are certain to be reset in this launch (k_update puts them
    // there).  Stats + reset + start stats is the longest chain of dependent steps in the kernel, so those items come
    // first and get a wavefront each: the chains start at once and none waits behind another reset of its wavefront.
    const int n0 = lone0 == 1 ? s_pref[1] : (lone0 == 2 ? n_rst : 0);
    const int f0 = lone0 == 1 ? n0 : 0;                           // items of `list` that the lone wavefronts take
    const int w_full = (n_full - f0 + GPW - 1) / GPW;             // wavefronts of the remaining full items
    // With many certain resets (zelda: thousands per step) the launch is throughput-bound and a wavefront takes two of
    // them: four statistics side by side instead of two.  With few, one each keeps the chain short.
    const bool pair = GPW >= 4 && n0 >= B.pair_min;
    const int w_lone = pair ? (n0 + 1) >> 1 : n0;
    const int w_total = w_lone + w_full + (n_inc + GPW - 1) / GPW;
    MaskT* champ_base = reinterpret_cast<MaskT*>(B.champ);
    const int lane64 = threadIdx.x & 63, wv = threadIdx.x >> 6, gw = lane64 / G;
    const int NPL = (PROB == PCGRL_PROB_BINARY) ? 1 : 3;
    const int W = P.width, H = P.height;
    const int tiles_bytes = (W * H + 15) & ~15;
    uint32_t* mt = reinterpret_cast<uint32_t*>(smem + (size_t)wv * (PCGRL_MT_N * 4 + tiles_bytes));
    uint8_t* tiles = reinterpret_cast<uint8_t*>(mt + PCGRL_MT_N);
    const MaskT rowmask = row_valid<MaskT>(g.lane, W, H);
    for (int wid = blockIdx.x * (PCGRL_BLOCK / 64) + wv; wid < w_total; wid += gridDim.x * (PCGRL_BLOCK / 64)) {
        const bool lone = wid < w_lone;                            // wave-uniform, like inc
        const bool inc = wid >= w_lone + w_full;
        // a certain reset occupies two lane groups: an even one for the map the step ended on, the odd one next to it for
        // the regenerated map
        const int item = lone ? (pair ? 2 * wid + (gw >> 1) : wid) : (inc ? (wid - w_lone - w_full) * GPW + gw : f0 + (wid - w_lone) * GPW + gw);
        const bool have = lone ? (item < n0 && (pair || gw < 2)) : item < (inc ? n_inc : n_full);
        const bool from_rst = lone && lone0 == 2;
        const int raw = !have ? 0 : (inc ? wl_get(B, WL_INC, s_pref_inc, item) : (from_rst ? wl_get(B, WL_RST, s_pref_rst, item) : wl_get(B, list, s_pref, item)));
        const int shard = (item >> 4) & (WL_NSHARD - 1);
        stats_wave_task<PROB, G, MaskT>(P, B, g, lane64, gw, lone, inc, pair, zinc, have, raw, shard, mode, parity, inline_reset, gen_map, mt, tiles, rowmask);
    }
}

// ------------------------------------------------------------------------------------------
// k_stats_wide: binary problem on maps of more than 16 rows -- one block of four wavefronts per work item.
// A 64x64 map costs one wavefront ~30k dependent instructions and a step only has ~1 400 such items for 1 024 SIMDs:
// with a wavefront per item the launch lasts as long as the unluckiest SIMD (two heavy items).  Here the four
// wavefronts of a block all hold the map and work through its components together (pcgrl_algos.h, "one map, several
// cooperating lane groups"): the set of unretired cells and the running path maximum live in LDS.  Same work list,
// same in-kernel reset as k_stats.
template <class MaskT>
struct LdsShared {
    MaskT* rest; int* bestp; int lane;
    __device__ __forceinline__ MaskT load_rest() const { return *reinterpret_cast<volatile MaskT*>(rest + lane); }
    template <class G>
    __device__ __forceinline__ bool retire(G& g, MaskT comp) const {
        const MaskT fb = g.first_bit(comp);                     // non-zero in one lane only: the component's first cell
        MaskT old = 0;
        if (fb) old = atomicAnd(rest + lane, ~fb);
        const bool won = __ballot((old & fb) != 0) != 0;
        if (comp) atomicAnd(rest + lane, ~comp);
        return won;
    }
    __device__ __forceinline__ int best() const { return *reinterpret_cast<volatile int*>(bestp); }
    __device__ __forceinline__ void raise(int v) const { if (lane == 0 && v > 0) atomicMax(bestp, v); }
};
// regions + longest path of the map whose rows are in `pass`, by the four wavefronts of the block.  Results in
// s_regions / s_best after the trailing barrier.  Every thread of the block calls this.
// The wavefronts of a block work as one team of NWAVES or -- for an environment that is certain to be reset -- as two teams
// of NWAVES / 2 on two maps at once (the one the step ended on and the regenerated one).  `tw` / `ts`: index in the team and
// team size; the s_* pointers are the team's own shared words.  `champ_e` (may be null): receives the rows of a champion
// component -- one whose sweep gave the final maximum -- or zeros when the path comes from the closed-form tiny
// components; *s_owner tells which wavefront of the team wrote it (-1: none).  Every thread of the block calls this (the
// barriers are block-wide: both teams pass them together).
template <class MaskT>
__device__ __forceinline__ void block_regions_and_path(DevGroup<64, MaskT>& g, MaskT pass, int tw, int ts, int lane, int H,
                                                       MaskT* s_rest, int* s_regions, int* s_best, int* s_owner, MaskT* champ_e) {
    const int bh = (H + ts - 1) / ts;
    const int row_lo = tw * bh, row_hi = (row_lo + bh < H) ? row_lo + bh : H;
    int tiny_regions, tiny_path;
    TL(20);
    const MaskT nontiny = rlp_prepare(g, pass, tiny_regions, tiny_path);
    TL(21);
    if (tw == 0) {
        s_rest[lane] = nontiny;
        if (lane == 0) { *s_regions = tiny_regions; *s_best = tiny_path; *s_owner = -1; }
    }
    __syncthreads();
    LdsShared<MaskT> sh = {s_rest, s_best, lane};
    int regions = 0, my_best = 0;
    MaskT my_champ = 0;
    MaskT rest = sh.load_rest();
    if (g.any(rest)) {
        const PcgFillCtx<DevGroup<64, MaskT>> ctx = pcg_fill_ctx(g, pass);
        do {
            rlp_process_seed(g, rlp_choose_seed(g, rest, row_lo, row_hi), ctx, sh, regions, my_best, my_champ);
            rest = sh.load_rest();
        } while (g.any(rest));
    }
    TL(22);
    if (lane == 0 && regions) atomicAdd(s_regions, regions);
    __syncthreads();
    TL(23);
    // one of the wavefronts whose own best sweep equals the final maximum writes its component
    if (lane == 0 && my_best > tiny_path && my_best == *s_best) atomicCAS(s_owner, -1, tw);
    __syncthreads();
    if (champ_e) {
        const int owner = *s_owner;
        if (tw == (owner < 0 ? 0 : owner)) champ_e[lane] = owner < 0 ? (MaskT)0 : my_champ;
    }
}

// One incremental item by one wavefront (binary_incremental), with the wavefront-level reset for the rare episode that
// ends on it (episodes that are certain to end are not routed here).
template <class MaskT>
__device__ __forceinline__ void wave_incremental_item(const PcgrlParams& P, const DevBufs& B, DevGroup<64, MaskT>& g, int raw, int parity,
                                                      int inline_reset, int gen_map, uint32_t* mt, uint8_t* tiles, int lane, MaskT rowmask) {
    const int e = wl_inc_env<64>(raw);
    MaskT* planes_e = reinterpret_cast<MaskT*>(B.planes) + (size_t)e * 64;
    MaskT* champ_e = reinterpret_cast<MaskT*>(B.champ) + (size_t)e * 64;
    const MaskT b0 = planes_e[lane];
    const MaskT champ_old = champ_e[lane];
    const int2 old = *reinterpret_cast<const int2*>(B.stats + (size_t)e * 8);
    const MaskT cbit = (lane == wl_inc_row<64>(raw)) ? (MaskT)1 << wl_inc_col<64>(raw) : (MaskT)0;
    int regions, path;
    MaskT champ;
    binary_incremental(g, (MaskT)(~b0 & rowmask), cbit, wl_inc_code<64>(raw) != 0, old.x, old.y, champ_old, regions, path, champ);
    champ_e[lane] = champ;
    int want = 0;
    if (lane == 0) {
        int32_t s[PCGRL_MAX_STATS] = {regions, path, 1, 0, 0, 0, 0, 0};
        want = finalize_item<PCGRL_PROB_BINARY>(P, B, e, s, MODE_STEP, parity, e & (WL_NSHARD - 1), !inline_reset) ? 1 : 0;
    }
    want = __builtin_amdgcn_readfirstlane(want);
    if (inline_reset && want) {
        ResetRows rr;
        wave_reset_env<PCGRL_PROB_BINARY>(P, B, e, gen_map, mt, (uint8_t*)nullptr, lane, 0, lane, &rr);
        MaskT n0, n1, n2;
        reset_rows_to_planes<MaskT>(P, planes_e, lane, rr.m0, rr.m1, rr.m2, n0, n1, n2);
        int32_t st[PCGRL_MAX_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
        compute_item_stats<PCGRL_PROB_BINARY>(g, P, n0, n1, n2, rowmask, st, champ);
        champ_e[lane] = champ;
        if (lane == 0) finalize_item<PCGRL_PROB_BINARY>(P, B, e, st, MODE_START, parity, e & (WL_NSHARD - 1));
    }
}

// The two blocks of a certain reset (k_stats_wide below) agree on who computes the old map's statistics through one word of
// DevBufs::wide_sync: the first to swing it to (epoch << 1 | who) has the half.  who = 0: the even block whose item it is, when it
// gets there; who = 1: the odd block, after waiting DevBufs::wide_spin sleeps (default WIDE_SPIN_LIMIT) for a sign of the even one.  Returns 1 to the winner.
#define WIDE_SPIN_LIMIT 400          /* x s_sleep(4) + an atomic load each: ~50 us */
__device__ __forceinline__ int wide_claim(int32_t* word, int epoch, int who) {
    int cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if ((cur >> 1) == epoch) return 0;                   // claimed in this launch already
        const int seen = atomicCAS(word, cur, (epoch << 1) | who);
        if (seen == cur) return 1;
        cur = seen;
    }
}

template <class MaskT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_stats_wide(PcgrlParams P, DevBufs B, int list, int parity, int mode, int clear_parity,
                                                             int inline_reset, int gen_map, int pair_few) {
    // inline_reset: MT ring + tile bytes, one set per wavefront (the block-wide reset uses the first)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_pref[WL_NSHARD + 1], s_pref_rst[WL_NSHARD + 1], s_pref_inc[WL_NSHARD + 1];
    __shared__ MaskT s_rest[2][64];
    __shared__ int s_regions[2], s_best[2], s_owner[2], s_flag, s_flag2[2], s_cur, s_claim;
    __shared__ int2 s_pre;
    if (clear_parity >= 0 && blockIdx.x == 0) wl_clear(B, clear_parity);
    TL_INIT(); TL(1);
    WTL(0, wall_clock64());
    DevGroup<64, MaskT> g;
    // with the in-kernel reset, k_update puts the environments that are certain to be reset on WL_RST: those come first;
    // the incremental items of a step (WL_INC) are taken, a wavefront each, by the blocks that have no full item
    const bool with_inc = mode == MODE_STEP && B.champ != nullptr;
    int n_rst = 0, n_inc = 0;
    const int n_chg = wl_load_prefix3(B, parity, list, with_inc ? WL_INC : -1, inline_reset ? WL_RST : -1, s_pref, s_pref_inc, s_pref_rst, &n_inc, &n_rst);
    const int n = n_rst + n_chg;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int W = P.width, H = P.height;
    const int wave_lds = PCGRL_MT_N * 4 + ((W * H + 15) & ~15);
    uint32_t* mt = reinterpret_cast<uint32_t*>(smem);
    uint8_t* tiles = reinterpret_cast<uint8_t*>(mt + PCGRL_MT_N);
    // the block-wide reset's buffers (raw MT19937 words, the map's bit string): in the scratch sets of wavefronts 1 .. NWAVES-1,
    // which only the incremental items at the end of the kernel use (the host sizes the allocation for both: launch_stats_p)
    uint32_t* rst_raw = reinterpret_cast<uint32_t*>(smem + wave_lds);
    uint64_t* rst_bits = reinterpret_cast<uint64_t*>(smem + wave_lds + (size_t)8 * W * H);
    const MaskT rowmask = row_valid<MaskT>(lane, W, H);
    // A certain reset is TWO items, for two blocks: the statistics of the map the step ended on (even items below 2 n_rst: the
    // "old-map" half, which only reads the planes and hands regions / path over through B.wide_sync) and the reset proper with the
    // statistics of the regenerated map (the odd item after it), which also finishes the step -- every wavefront of a block on one
    // map instead of half of them on each.  The second half waits for "planes read" before it overwrites them and for the result
    // before it finishes the step.  It cannot wait for ever: the grid is even (launch_stats_p), so an even block only ever gets
    // even items -- old-map halves and full items, which wait for nobody -- and the block an odd one waits for is its left
    // neighbour in the same round, which was dispatched before it.
    // pair_few (a step; k_update ranks the list: difficulty_bucket): the full items of the list's second class -- maps with few regions,
    // one long double sweep on which seven of eight wavefronts would only wait -- go two to a block, half of the wavefronts each, so
    // that (nearly) every item of a step starts at once instead of a quarter of them behind the first to finish.
    const int n_many = pair_few ? s_pref[WL_NSHARD / 2] : n_chg;
    const int n_items = 2 * n_rst + n_many + (n_chg - n_many + 1) / 2;
    const int epoch = B.wide_epoch;
    WTL(1, wall_clock64());
    WTL(5, (unsigned long long)n_items | ((unsigned long long)n_rst << 16) | ((unsigned long long)n_many << 32) | ((unsigned long long)n_inc << 48));
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const bool lone = item < 2 * n_rst;                  // certain reset (block-uniform, like everything below that is not per lane)
        const bool old_half = lone && (item & 1) == 0;
        WTL(2, (old_half ? 1 : (lone ? 2 : (item >= 2 * n_rst + n_many ? 4 : 3))) | (item << 8));
        if (item >= 2 * n_rst + n_many) {
            constexpr int TS = NWAVES / 2;
            const int team = wv / TS, tw = wv % TS;
            const int idx0 = n_many + 2 * (item - 2 * n_rst - n_many);
            const int e0 = wl_get(B, list, s_pref, idx0) & ~WL_RESET_ONLY;
            const int e1 = idx0 + 1 < n_chg ? (wl_get(B, list, s_pref, idx0 + 1) & ~WL_RESET_ONLY) : -1;
            const int et = team ? e1 : e0;
            const int shard = (item >> 4) & (WL_NSHARD - 1);
            MaskT pass = 0;
            if (et >= 0) pass = (MaskT)(~(reinterpret_cast<const MaskT*>(B.planes) + (size_t)et * 64)[lane] & rowmask);
            if (threadIdx.x < 2) s_flag2[threadIdx.x] = 0;
            block_regions_and_path(g, pass, tw, TS, lane, H, s_rest[team], &s_regions[team], &s_best[team], &s_owner[team],
                                   (et >= 0 && B.champ) ? reinterpret_cast<MaskT*>(B.champ) + (size_t)et * 64 : (MaskT*)nullptr);
            if (tw == 0 && lane == 0 && et >= 0) {
                int32_t s[PCGRL_MAX_STATS] = {s_regions[team], s_best[team], s_owner[team] >= 0 ? 1 : 0, 0, 0, 0, 0, 0};
                const bool want = finalize_item<PCGRL_PROB_BINARY>(P, B, et, s, mode, parity, shard, !inline_reset);
                s_flag2[team] = (want && inline_reset) ? 1 : 0;
            }
            __syncthreads();
            for (int t = 0; t < 2; t++) {
                if (!(inline_reset && s_flag2[t])) continue;     // block-uniform: PcgrlEnv.reset of that environment by the whole block, then its start stats
                const int er = t ? e1 : e0;
                MaskT* planes_r = reinterpret_cast<MaskT*>(B.planes) + (size_t)er * 64;
                const MaskT b0 = block_reset_env<PCGRL_PROB_BINARY, NWAVES * 64, MaskT>(P, B, er, gen_map, mt, tiles, rst_raw, rst_bits, &s_cur);
                if (wv == 0) planes_r[lane] = b0;
                block_regions_and_path(g, (MaskT)(~b0 & rowmask), wv, NWAVES, lane, H, s_rest[0], &s_regions[0], &s_best[0], &s_owner[0],
                                       B.champ ? reinterpret_cast<MaskT*>(B.champ) + (size_t)er * 64 : (MaskT*)nullptr);
                if (threadIdx.x == 0) {
                    int32_t s[PCGRL_MAX_STATS] = {s_regions[0], s_best[0], s_owner[0] >= 0 ? 1 : 0, 0, 0, 0, 0, 0};
                    finalize_item<PCGRL_PROB_BINARY>(P, B, er, s, MODE_START, parity, shard);
                }
                __syncthreads();
            }
            continue;
        }
        const int raw = lone ? wl_get(B, WL_RST, s_pref_rst, item >> 1) : wl_get(B, list, s_pref, item - 2 * n_rst);
        const bool reset_only = (raw & WL_RESET_ONLY) != 0;
        const int e = raw & ~WL_RESET_ONLY;
        const int shard = (item >> 4) & (WL_NSHARD - 1);
        MaskT* planes_e = reinterpret_cast<MaskT*>(B.planes) + (size_t)e * 64;
        MaskT* champ_e = B.champ ? reinterpret_cast<MaskT*>(B.champ) + (size_t)e * 64 : nullptr;
        int32_t* sync_e = B.wide_sync + (size_t)e * 4;
        if (old_half) {
            if (reset_only) continue;                        // (an unchanged map: the step needs no statistics)
            // claim the half (the other block takes it over when this one has not shown up after a long wait: wide_claim)
            if (threadIdx.x == 0) s_claim = wide_claim(sync_e + 3, epoch, 0);
            __syncthreads();
            if (!s_claim) { __syncthreads(); continue; }
            const MaskT b_old = planes_e[lane];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(sync_e + 0, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // the planes may go
            block_regions_and_path(g, (MaskT)(~b_old & rowmask), wv, NWAVES, lane, H, s_rest[0], &s_regions[0], &s_best[0], &s_owner[0], (MaskT*)nullptr);
            if (threadIdx.x == 0) {
                sync_e[2] = s_regions[0] | (s_best[0] << 16);     // (a 64 x 64 map has at most 2 048 regions and no path beyond 4 095)
                __hip_atomic_store(sync_e + 1, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            continue;
        }
        if (lone) {
            // the reset, then the statistics of the regenerated map; the step is finished with the counters read before the reset
            // (generation + statistics of a fresh map is one of the step's longest items: its wavefronts are served before those of the
            //  compute unit's other block until its first sweep ends -- the sweeps themselves run at level 3, rlp_process_seed.  C5 steady
            //  49.05 -> 48.75 us on one box, profiles/r5_round5/probe/ab_lone_prio.txt; the same for the paired items measured nothing)
            __builtin_amdgcn_s_setprio(1);
            if (threadIdx.x == 0) s_pre = reinterpret_cast<const int2*>(B.counters)[e];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            TL(24);
            const MaskT n0 = block_reset_env<PCGRL_PROB_BINARY, NWAVES * 64, MaskT>(P, B, e, gen_map, mt, tiles, rst_raw, rst_bits, &s_cur);     // (every wavefront helps: reset_env.h)
            int own_old = 0;         // this block computed the old map's statistics itself (block-uniform)
            if (!reset_only) {
                // wait for "planes read" from the block that has the other half -- for a bounded time: the hand-over assumes that block
                // was dispatched (it is this block's left neighbour of the same round, and workgroups start in index order), which the
                // programming model does not promise.  After ~50 us without a sign of it this block claims the half for itself.
                if (threadIdx.x == 0) {
                    int spins = 0, mine = 0;
                    while (__hip_atomic_load(sync_e + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins == B.wide_spin && wide_claim(sync_e + 3, epoch, 1)) { mine = 1; break; }
                    }
                    s_claim = mine;
                }
                __syncthreads();
                own_old = s_claim;
                if (own_old) {
                    const MaskT b_old = planes_e[lane];
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    block_regions_and_path(g, (MaskT)(~b_old & rowmask), wv, NWAVES, lane, H, s_rest[1], &s_regions[1], &s_best[1], &s_owner[1], (MaskT*)nullptr);
                }
                __syncthreads();
            }
            if (wv == 0) planes_e[lane] = n0;
            TL(26);
            block_regions_and_path(g, (MaskT)(~n0 & rowmask), wv, NWAVES, lane, H, s_rest[0], &s_regions[0], &s_best[0], &s_owner[0], champ_e);
            if (threadIdx.x == 0) {
                if (!reset_only) {
                    int32_t s[PCGRL_MAX_STATS] = {s_regions[1], s_best[1], 0, 0, 0, 0, 0, 0};
                    if (!own_old) {
                        while (__hip_atomic_load(sync_e + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(4);
                        const int packed = __hip_atomic_load(sync_e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s[0] = packed & 0xFFFF; s[1] = (packed >> 16) & 0xFFFF;
                    }
                    finalize_item<PCGRL_PROB_BINARY>(P, B, e, s, MODE_STEP, parity, shard, false, WL_RST, &s_pre);
                }
                int32_t st[PCGRL_MAX_STATS] = {s_regions[0], s_best[0], s_owner[0] >= 0 ? 1 : 0, 0, 0, 0, 0, 0};
                finalize_item<PCGRL_PROB_BINARY>(P, B, e, st, MODE_START, parity, shard);
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            continue;
        }
        if (threadIdx.x == 0) s_flag = reset_only ? 1 : 0;
        if (!reset_only) {
            const MaskT b0 = planes_e[lane];
            block_regions_and_path(g, (MaskT)(~b0 & rowmask), wv, NWAVES, lane, H, s_rest[0], &s_regions[0], &s_best[0], &s_owner[0], champ_e);
            if (threadIdx.x == 0) {
                int32_t s[PCGRL_MAX_STATS] = {s_regions[0], s_best[0], s_owner[0] >= 0 ? 1 : 0, 0, 0, 0, 0, 0};
                const bool want = finalize_item<PCGRL_PROB_BINARY>(P, B, e, s, mode, parity, shard, !inline_reset);
                s_flag = (want && inline_reset) ? 1 : 0;
            }
        }
        __syncthreads();
        if (inline_reset && s_flag) {   // block-uniform: PcgrlEnv.reset of this environment, then its start stats
            const MaskT b0 = block_reset_env<PCGRL_PROB_BINARY, NWAVES * 64, MaskT>(P, B, e, gen_map, mt, tiles, rst_raw, rst_bits, &s_cur);
            if (wv == 0) planes_e[lane] = b0;
            block_regions_and_path(g, (MaskT)(~b0 & rowmask), wv, NWAVES, lane, H, s_rest[0], &s_regions[0], &s_best[0], &s_owner[0], champ_e);
            if (threadIdx.x == 0) {
                int32_t s[PCGRL_MAX_STATS] = {s_regions[0], s_best[0], s_owner[0] >= 0 ? 1 : 0, 0, 0, 0, 0, 0};
                finalize_item<PCGRL_PROB_BINARY>(P, B, e, s, MODE_START, parity, shard);
            }
        }
        __syncthreads();
    }
    TL(27);
    WTL(3, wall_clock64());
    if (n_inc > 0) {
        // incremental items: by the blocks beyond the full items if there are any, else by all
        const int first = ((int)gridDim.x > n_items) ? n_items : 0;
        const int nblk = (int)gridDim.x - first;
        if ((int)blockIdx.x >= first) {
            uint32_t* mtw = reinterpret_cast<uint32_t*>(smem + (size_t)wv * wave_lds);
            uint8_t* tilesw = reinterpret_cast<uint8_t*>(mtw + PCGRL_MT_N);
            for (int j = ((int)blockIdx.x - first) * NWAVES + wv; j < n_inc; j += nblk * NWAVES)
                wave_incremental_item<MaskT>(P, B, g, wl_get(B, WL_INC, s_pref_inc, j), parity, inline_reset, gen_map, mtw, tilesw, lane, rowmask);
        }
    }
    TL(28);
    WTL(4, wall_clock64());
}
