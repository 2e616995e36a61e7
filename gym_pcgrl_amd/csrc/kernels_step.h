// k_step: the whole PcgrlEnv.step of 64 environments by one block of four wavefronts -- one launch per step, no work lists.
// Part of the single translation unit pcgrl_abi.hip (see its header comment for the overall picture).
//
// The two-launch pipeline (k_update -> global work lists -> k_stats) spends most of a 65 536-environment step waiting:
// launch latencies, the append atomics, the list prefix, the item and plane round trips, and k_update runs at one wavefront
// per SIMD with the rest of the chip idle.  Here wavefront 0 of a block does Representation.update for the block's 64
// environments (update_env, the body of k_update), the changed environments are compacted into LDS by kind -- certain resets,
// full recomputations ordered by difficulty bucket, incremental updates -- and after one barrier the four wavefronts work
// through those tasks (stats_wave_task, the body of k_stats: same statistics, same in-kernel resets).  A block sees ~20
// changed environments of 64, i.e. five or six wavefront tasks: one or two rounds.
// pcgrl_rollout runs a whole tape of actions in ONE launch of this kernel (the loop over `steps`).
// For the binary and zelda problems on maps of at most 16 rows with the single-cell representations and auto-reset; every
// other configuration takes the two-launch pipeline.
#pragma once

// MULTI: the pcgrl_rollout form (loop over the tape); the single-step form is compiled without the loop so that it pays
// nothing for it.
template <int PROB, int REP, class MaskT, bool MULTI>
__global__ __launch_bounds__(PCGRL_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_step(PcgrlParams P, DevBufs B, const int32_t* __restrict__ actions, int parity, int gen_map,
                                                                                                           int steps, size_t action_stride, double* reward_out, uint8_t* done_out, int32_t* info_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // per wave MT ring + tile bytes (in-kernel resets)
    __shared__ int s_items[3][64];      // 0: certain resets, 1: full recomputations (by bucket), 2: incremental updates
    __shared__ int s_n[3];
    __shared__ int s_hist[64];
    constexpr int G = 16, GPW = 4;
    const int W = P.width, H = P.height;
    // steps > 1 (pcgrl_rollout): the environments of a block do not depend on any other block, so the block simply goes on
    // with the next row of the action tape -- no launch, no grid-wide barrier between steps, blocks run ahead of each other
#pragma clang loop unroll(disable)
  for (int t = 0; t < (MULTI ? steps : 1); t++) {
    // the thread index goes through an opaque move so that nothing per-lane is hoisted out of the loop and kept in
    // registers across steps (the single-step kernel needs 97 VGPRs; with hoisting the loop form spilled)
    int tid = (int)threadIdx.x;
    if (MULTI) asm volatile("" : "+v"(tid));
    const int lane64 = tid & 63, wv = tid >> 6, gw = lane64 / G;
    DevGroup<G, MaskT> g(lane64);
    const int32_t* actions_t = actions + (size_t)t * action_stride;
    if (wv == 1) s_hist[lane64] = 0;
    __syncthreads();                        // also: everything the previous step wrote is visible to the whole block
    if (wv == 0) {
        const int e = blockIdx.x * 64 + lane64;
        UpdateOut u = {false, false, false, false, 0, 0};
        if (e < P.num_envs) u = update_env<REP, MaskT>(P, B, actions_t, e);
        const bool first = u.rst || u.sure_done;               // reset-only, or certain to end: k_stats' "lone" items
        const bool packed_full = PROB == PCGRL_PROB_ZELDA && B.zelda_inc;
        int dest = -1, v = e;
        if (first) { dest = 0; v = u.rst ? (e | WL_RESET_ONLY) : e; }
        else if (u.chg) { dest = u.cheap ? 2 : 1; v = (u.cheap || packed_full) ? u.inc_item : e; }
        const uint64_t m0 = __ballot(dest == 0), m1 = __ballot(dest == 1), m2 = __ballot(dest == 2);
        const uint64_t below = (1ull << lane64) - 1ull;
        if (dest == 0) s_items[0][__popcll(m0 & below)] = v;
        if (dest == 2) s_items[2][__popcll(m2 & below)] = v;
        // full recomputations in bucket order (the four maps that share a wavefront should cost about the same)
        int rank = 0;
        const int bucket = u.bucket & 63;
        if (dest == 1) rank = atomicAdd(&s_hist[bucket], 1);
        __builtin_amdgcn_wave_barrier();
        int incl = s_hist[lane64];
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane64 >= o) incl += up;
        }
        const int excl = incl - s_hist[lane64];
        __builtin_amdgcn_wave_barrier();
        s_hist[lane64] = excl;
        __builtin_amdgcn_wave_barrier();
        if (dest == 1) s_items[1][s_hist[bucket] + rank] = v;
        if (lane64 == 0) { s_n[0] = __popcll(m0); s_n[1] = __popcll(m1); s_n[2] = __popcll(m2); }
    }
    __syncthreads();
    const int n0 = s_n[0], n1 = s_n[1], n2 = s_n[2];
    const int w_full = (n1 + GPW - 1) / GPW;
    const int w_total = n0 + w_full + (n2 + GPW - 1) / GPW;
    const int tiles_bytes = (W * H + 15) & ~15;
    uint32_t* mt = reinterpret_cast<uint32_t*>(smem + (size_t)wv * (PCGRL_MT_N * 4 + tiles_bytes));
    uint8_t* tiles = reinterpret_cast<uint8_t*>(mt + PCGRL_MT_N);
    const MaskT rowmask = row_valid<MaskT>(g.lane, W, H);
    const bool zinc = PROB == PCGRL_PROB_ZELDA && sizeof(MaskT) == 4 && B.zelda_inc;
    for (int wid = wv; wid < w_total; wid += PCGRL_BLOCK / 64) {
        const bool lone = wid < n0, inc = wid >= n0 + w_full;
        const int item = lone ? wid : (inc ? (wid - n0 - w_full) * GPW + gw : (wid - n0) * GPW + gw);
        const bool have = lone ? gw < 2 : item < (inc ? n2 : n1);
        const int raw = have ? s_items[lone ? 0 : (inc ? 2 : 1)][item] : 0;
        stats_wave_task<PROB, G, MaskT>(P, B, g, lane64, gw, lone, inc, false, zinc, have, raw, lane64, MODE_STEP, parity, 1, gen_map, mt, tiles, rowmask);
    }
    if (MULTI && (reward_out || done_out || info_out)) {   // kernel-uniform: the per-step outputs of the block's environments, row t
        __syncthreads();
        const int e0 = blockIdx.x * 64, ne = (P.num_envs - e0) < 64 ? (P.num_envs - e0) : 64;
        const size_t row = (size_t)t * P.num_envs + e0;
        if (reward_out && tid < ne) reward_out[row + tid] = B.reward[e0 + tid];
        if (done_out && tid >= 64 && tid - 64 < ne) done_out[row + tid - 64] = B.done[e0 + tid - 64];
        if (info_out) for (int i = tid; i < ne * 10; i += PCGRL_BLOCK) info_out[row * 10 + i] = B.info[(size_t)e0 * 10 + i];
    }
  }
}
