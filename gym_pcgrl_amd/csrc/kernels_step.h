// k_step: the whole PcgrlEnv.step of 64 environments by one block of four wavefronts -- one launch per step, no work lists,
// the block's environment state staged in LDS.  Part of the single translation unit pcgrl_abi.hip.
//
// A step is a chain of dependent memory round trips per environment (counters / cursor / statistics -> the cell's plane row,
// champion rows, MT19937 words -> the plane rows again for the statistics -> previous statistics for the reward ...), and a
// thread-per-environment access to per-environment records is one 128-byte line per lane: under the load of 65 536
// environments every such round trip costs 2-3 us (tools/timeline.py).  Here a block first copies everything that is
// addressed by the environment index alone -- row planes, champion rows, current and start statistics, counters, cursors,
// the draw cache -- for its 64 consecutive environments into LDS: a handful of fully coalesced wide loads by all four
// wavefronts, ONE round trip.  The shared device functions then run on that copy (a DevBufs whose per-environment pointers
// are rebased into the LDS block; the byte map, heatmap and MT19937 rings stay where they are):
//   * wavefront 0 does Representation.update for the 64 environments (update_env, the body of k_update); the cursor draws of
//     the narrow representation come out of the per-environment draw cache (the next eight words of the stream, computed
//     ahead), so the update touches no ring;
//   * the changed environments are compacted into LDS task lists by kind -- certain resets, full recomputations ordered by
//     difficulty bucket, incremental updates -- and after one (LDS-only) barrier the four wavefronts take those tasks as they
//     become free (stats_wave_task, the body of k_stats: same statistics, same in-kernel resets);
//   * behind that barrier wavefront 0 first writes the consumed draws into the rings and refills the draw caches
//     (fifo_refill): ring traffic that nothing in the step waits for;
//   * reward / done / info rows are produced in LDS; at the end the block writes its state back, coalesced -- plane rows,
//     champion rows and start statistics only for the environments that changed them.
// pcgrl_rollout runs a whole tape of actions in ONE launch of this kernel (the loop over `steps`): the state stays in LDS
// from the first step to the last.
// For the binary and zelda problems on maps of at most 16 rows with the single-cell representations and auto-reset; every
// other configuration takes the two-launch pipeline.
#pragma once

struct StepLds { int planes, champ, stats, start, info, rew, cnt, cur, fifo, tag, pos, done, total; };
__host__ __device__ __forceinline__ StepLds step_lds_layout(int plane_row_bytes, int champ_row_bytes, bool fifo) {
    StepLds L;
    int o = 0;
    L.planes = o; o += 64 * plane_row_bytes;
    L.champ = o; o += 64 * champ_row_bytes;
    L.stats = o; o += 64 * 32;
    L.start = o; o += 64 * 32;
    L.info = o; o += 64 * 40;
    L.rew = o; o += 64 * 8;
    L.cnt = o; o += 64 * 8;
    L.cur = o; o += 64 * 8;
    L.fifo = o; o += fifo ? 64 * PCGRL_FIFO_N * 4 : 0;
    L.tag = o; o += fifo ? 64 * 4 : 0;
    L.pos = o; o += 64 * 2;
    L.done = o; o += 64;
    L.total = (o + 15) & ~15;
    return L;
}

// Block-wide copy of `bytes` bytes, both sides 16-byte aligned (global <-> LDS; every thread of the block calls it).
__device__ __forceinline__ void blk_copy(uint8_t* dst, const uint8_t* src, int bytes) {
    const int nv = bytes >> 4;
    for (int i = threadIdx.x; i < nv; i += PCGRL_BLOCK) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (int i = (nv << 4) + threadIdx.x; i < bytes; i += PCGRL_BLOCK) dst[i] = src[i];
}
// The same for per-environment rows of `row` bytes (a multiple of 16), only the rows whose flag is set.
__device__ __forceinline__ void blk_copy_rows(uint8_t* dst, const uint8_t* src, int rows, int row, const uint8_t* flag) {
    const int per = row >> 4;
    for (int i = threadIdx.x; i < rows * per; i += PCGRL_BLOCK)
        if (flag[i / per]) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
}

// Wavefront 0, one lane per environment, behind the task barrier: the k words the step drew from the draw cache go into the
// ring (the canonical lazy-ring state: exactly what k draws from the ring would have left there), the cache is shifted and
// topped up with the k words after it.  B: the block's view (rings in global memory, fifo / fifo_tag in its LDS copy).
__device__ __forceinline__ void fifo_refill(const DevBufs& B, int e, int k, int cur0) {
    uint32_t* ring = B.rng_rep + (size_t)e * PCGRL_MT_N;
    uint32_t* fl = B.fifo + (size_t)e * PCGRL_FIFO_N;
    const int base = mt_wrap(cur0 + PCGRL_FIFO_N), basem = mt_wrap(base + PCGRL_MT_M);
    uint32_t a[PCGRL_FIFO_N + 1], m[PCGRL_FIFO_N];
#pragma unroll
    for (int i = 0; i <= PCGRL_FIFO_N; i++) a[i] = (i <= k) ? ring[mt_wrap(base + i)] : 0u;       // operands of the new words: none of
#pragma unroll
    for (int i = 0; i < PCGRL_FIFO_N; i++) m[i] = (i < k) ? ring[mt_wrap(basem + i)] : 0u;        // them is a slot written below
#pragma unroll
    for (int i = 0; i < PCGRL_FIFO_N; i++) if (i < k) ring[mt_wrap(cur0 + i)] = fl[i];
#pragma unroll
    for (int j = 0; j < PCGRL_FIFO_N; j++) if (j + k < PCGRL_FIFO_N) fl[j] = fl[j + k];          // ascending: reads stay ahead of writes
#pragma unroll
    for (int i = 0; i < PCGRL_FIFO_N; i++) if (i < k) fl[PCGRL_FIFO_N - k + i] = mt_twist(a[i], a[i + 1], m[i]);
    B.fifo_tag[e] = mt_wrap(cur0 + k);
}

// MULTI: the pcgrl_rollout form (loop over the tape); the single-step form is compiled without the loop so that it pays
// nothing for it.
template <int PROB, int REP, class MaskT, bool MULTI>
__global__ __launch_bounds__(PCGRL_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_step(PcgrlParams P, DevBufs Bg, const int32_t* __restrict__ actions, int parity, int gen_map,
                                                                                                           int steps, size_t action_stride, double* reward_out, uint8_t* done_out, int32_t* info_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // the block's state copy, then per wave MT ring + tile bytes (in-kernel resets)
    __shared__ int s_items[3][64];      // 0: certain resets, 1: full recomputations (by bucket), 2: incremental updates
    __shared__ int s_n[3];
    __shared__ int s_hist[64];
    __shared__ int s_next;              // next wavefront task of the step (the wavefronts take them as they become free)
    __shared__ StepLocal s_loc;
    constexpr int G = 16, GPW = 4;
    constexpr int NPL = (PROB == PCGRL_PROB_BINARY) ? 1 : 3;
    constexpr int kPlaneRow = G * NPL * (int)sizeof(MaskT);
    const int W = P.width, H = P.height;
    TL_INIT(); TL(1);
    const int e0 = blockIdx.x * 64;
    const int ne = (P.num_envs - e0) < 64 ? (P.num_envs - e0) : 64;
    const bool has_champ = PROB == PCGRL_PROB_BINARY && Bg.champ != nullptr;
    const bool has_fifo = REP == PCGRL_REP_NARROW && Bg.fifo != nullptr;
    const int champ_row = has_champ ? G * (int)sizeof(MaskT) : 0;
    const StepLds L = step_lds_layout(kPlaneRow, champ_row, has_fifo);
    // ---- the block's copy of the per-environment state, and a DevBufs for the shared device functions in which index 0 is
    // the block's first environment: the staged arrays point into the LDS copy, the arrays that stay in global memory (byte
    // maps, heatmap, MT19937 rings, tile probabilities, episode statistics) are moved forward to the block's slice.  Inside the
    // block an environment is known by its index 0..63 only.
    DevBufs B = Bg;
    const size_t cells_ = (size_t)W * H;
    B.map = Bg.map + (size_t)e0 * cells_;
    B.old_map = Bg.old_map + (size_t)e0 * cells_;
    B.heat = Bg.heat + (size_t)e0 * cells_;
    B.tile_p = Bg.tile_p + (size_t)e0 * 2;
    B.rng_rep = Bg.rng_rep + (size_t)e0 * PCGRL_MT_N;
    if (Bg.rng_prob) B.rng_prob = Bg.rng_prob + (size_t)e0 * PCGRL_MT_N;
    if (Bg.ep_return) { B.ep_return = Bg.ep_return + e0; B.ep_length = Bg.ep_length + e0; B.last_return = Bg.last_return + e0; B.last_length = Bg.last_length + e0; }
    B.planes = smem + L.planes;
    if (has_champ) B.champ = smem + L.champ;
    B.stats = reinterpret_cast<int32_t*>(smem + L.stats);
    B.start_stats = reinterpret_cast<int32_t*>(smem + L.start);
    B.info = reinterpret_cast<int32_t*>(smem + L.info);
    B.reward = reinterpret_cast<double*>(smem + L.rew);
    B.counters = reinterpret_cast<int32_t*>(smem + L.cnt);
    B.rng_cur = reinterpret_cast<int32_t*>(smem + L.cur);
    if (has_fifo) {
        B.fifo = reinterpret_cast<uint32_t*>(smem + L.fifo);
        B.fifo_tag = reinterpret_cast<int32_t*>(smem + L.tag);
    } else {
        B.fifo = nullptr; B.fifo_tag = nullptr;
    }
    B.pos = smem + L.pos;
    B.done = smem + L.done;
    blk_copy(smem + L.planes, reinterpret_cast<const uint8_t*>(Bg.planes) + (size_t)e0 * kPlaneRow, ne * kPlaneRow);
    if (has_champ) blk_copy(smem + L.champ, reinterpret_cast<const uint8_t*>(Bg.champ) + (size_t)e0 * champ_row, ne * champ_row);
    blk_copy(smem + L.stats, reinterpret_cast<const uint8_t*>(Bg.stats + (size_t)e0 * 8), ne * 32);
    blk_copy(smem + L.start, reinterpret_cast<const uint8_t*>(Bg.start_stats + (size_t)e0 * 8), ne * 32);
    blk_copy(smem + L.cnt, reinterpret_cast<const uint8_t*>(Bg.counters + (size_t)e0 * 2), ne * 8);
    blk_copy(smem + L.cur, reinterpret_cast<const uint8_t*>(Bg.rng_cur + (size_t)e0 * 2), ne * 8);
    blk_copy(smem + L.pos, Bg.pos + (size_t)e0 * 2, ne * 2);
    if (has_fifo) {
        blk_copy(smem + L.fifo, reinterpret_cast<const uint8_t*>(Bg.fifo + (size_t)e0 * PCGRL_FIFO_N), ne * PCGRL_FIFO_N * 4);
        blk_copy(smem + L.tag, reinterpret_cast<const uint8_t*>(Bg.fifo_tag + e0), ne * 4);
    }
    if (threadIdx.x < 64) s_loc.dirty[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_loc.e0 = 0;
    uint8_t* reset_scratch = smem + L.total;
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
    const int32_t* actions_t = actions + (size_t)t * action_stride + (size_t)e0 * (REP == PCGRL_REP_WIDE ? 3 : 1);
    if (wv == 1) s_hist[lane64] = 0;
    __syncthreads();                        // the state copy is complete; everything the previous step wrote is visible to the whole block
    TL(17);
    if (wv == 0) {
        const int e = lane64;                                  // block-local index (see B above)
        UpdateOut u = {};
        if (lane64 < ne) u = update_env<REP, MaskT, true>(P, B, actions_t, e);
        TL(2);
        const bool first = u.rst || u.sure_done;               // reset-only, or certain to end: k_stats' "lone" items
        const bool packed_full = PROB == PCGRL_PROB_ZELDA && B.zelda_inc;
        int dest = -1, v = e;
        if (first) { dest = 0; v = u.rst ? (e | WL_RESET_ONLY) : e; }
        else if (u.chg) { dest = u.cheap ? 2 : 1; v = (u.cheap || packed_full) ? u.inc_item : e; }
        s_loc.k[lane64] = (uint8_t)u.k;
        if (u.chg) s_loc.dirty[lane64] = 1;
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
        if (lane64 == 0) { s_n[0] = __popcll(m0); s_n[1] = __popcll(m1); s_n[2] = __popcll(m2); s_next = 0; s_loc.refill_done = 0; }
        // LDS-only barrier: what wavefront 0 has in flight to global memory (byte-map cells, heatmap increments) concerns no
        // task that starts now -- environments that are certain to be reset got no such write, the other tasks work on the
        // LDS copy -- and is waited for below, before refill_done is published.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (has_fifo && lane64 < ne && u.k > 0 && !first) fifo_refill(B, e, u.k, u.cur0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // ring words, byte-map cells and heatmap increments have landed
        if (lane64 == 0) __hip_atomic_store(&s_loc.refill_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    TL(3);
    const int n0 = s_n[0], n1 = s_n[1], n2 = s_n[2];
    const int w_full = (n1 + GPW - 1) / GPW;
    const int w_total = n0 + w_full + (n2 + GPW - 1) / GPW;
    const int tiles_bytes = (W * H + 15) & ~15;
    uint32_t* mt = reinterpret_cast<uint32_t*>(reset_scratch + (size_t)wv * (PCGRL_MT_N * 4 + tiles_bytes));
    uint8_t* tiles = reinterpret_cast<uint8_t*>(mt + PCGRL_MT_N);
    const MaskT rowmask = row_valid<MaskT>(g.lane, W, H);
    const bool zinc = PROB == PCGRL_PROB_ZELDA && sizeof(MaskT) == 4 && B.zelda_inc;
    // Tasks in the order certain resets (the longest chains), full recomputations, incremental updates; a wavefront takes
    // the next one whenever it is free, so the block ends when the work is done, not when its unluckiest quarter is
    // (a static split left the last wavefront of a block ~10 us behind the others).
    for (;;) {
        int wid = 0;
        if (lane64 == 0) wid = atomicAdd(&s_next, 1);
        wid = __builtin_amdgcn_readfirstlane(wid);
        if (wid >= w_total) break;
        const bool lone = wid < n0, inc = wid >= n0 + w_full;
        const int item = lone ? wid : (inc ? (wid - n0 - w_full) * GPW + gw : (wid - n0) * GPW + gw);
        const bool have = lone ? gw < 2 : item < (inc ? n2 : n1);
        const int raw = have ? s_items[lone ? 0 : (inc ? 2 : 1)][item] : 0;
        TL(lone ? 4 : (inc ? 6 : 5));
        stats_wave_task<PROB, G, MaskT>(P, B, g, lane64, gw, lone, inc, false, zinc, have, raw, lane64, MODE_STEP, parity, 1, gen_map, mt, tiles, rowmask, &s_loc);
        TL(7);
    }
    TL(8);
    if (MULTI && (reward_out || done_out || info_out)) {   // kernel-uniform: the per-step outputs of the block's environments, row t
        __syncthreads();
        const size_t row = (size_t)t * P.num_envs + e0;
        if (reward_out && tid < ne) reward_out[row + tid] = reinterpret_cast<const double*>(smem + L.rew)[tid];
        if (done_out && tid >= 64 && tid - 64 < ne) done_out[row + tid - 64] = smem[L.done + tid - 64];
        if (info_out) for (int i = tid; i < ne * 10; i += PCGRL_BLOCK) info_out[row * 10 + i] = reinterpret_cast<const int32_t*>(smem + L.info)[i];
    }
  }
    // ---- the block's state goes back, coalesced; plane rows, champion rows and start statistics only where they changed
    __syncthreads();
    blk_copy(reinterpret_cast<uint8_t*>(Bg.stats + (size_t)e0 * 8), smem + L.stats, ne * 32);
    blk_copy(reinterpret_cast<uint8_t*>(Bg.info + (size_t)e0 * 10), smem + L.info, ne * 40);
    blk_copy(reinterpret_cast<uint8_t*>(Bg.reward + e0), smem + L.rew, ne * 8);
    blk_copy(reinterpret_cast<uint8_t*>(Bg.counters + (size_t)e0 * 2), smem + L.cnt, ne * 8);
    blk_copy(reinterpret_cast<uint8_t*>(Bg.rng_cur + (size_t)e0 * 2), smem + L.cur, ne * 8);
    blk_copy(Bg.pos + (size_t)e0 * 2, smem + L.pos, ne * 2);
    blk_copy(Bg.done + e0, smem + L.done, ne);
    if (has_fifo) {
        blk_copy(reinterpret_cast<uint8_t*>(Bg.fifo + (size_t)e0 * PCGRL_FIFO_N), smem + L.fifo, ne * PCGRL_FIFO_N * 4);
        blk_copy(reinterpret_cast<uint8_t*>(Bg.fifo_tag + e0), smem + L.tag, ne * 4);
    }
    blk_copy_rows(reinterpret_cast<uint8_t*>(Bg.planes) + (size_t)e0 * kPlaneRow, smem + L.planes, ne, kPlaneRow, s_loc.dirty);
    if (has_champ) blk_copy_rows(reinterpret_cast<uint8_t*>(Bg.champ) + (size_t)e0 * champ_row, smem + L.champ, ne, champ_row, s_loc.dirty);
    blk_copy_rows(reinterpret_cast<uint8_t*>(Bg.start_stats + (size_t)e0 * 8), smem + L.start, ne, 32, s_loc.dirty);
}
