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

// LDS copy of a block: first the segments that are loaded at the start, contiguous and in this order (each a multiple of 16
// bytes for a full block of 64 environments), then the ones that are only produced.  Sizes depend on the kernel's template
// parameters only (the champion rows of the binary problem and the draw cache of the narrow representation keep their room
// even when the feature is off), so that the prefetch can be laid out at compile time.
struct StepLds { int planes, champ, stats, start, cnt, cur, fifo, tag, pos, act, flat, in_total, info, rew, done, total; };
__host__ __device__ constexpr StepLds step_lds_layout(int plane_row_bytes, int champ_row_bytes, bool fifo, int action_width, int epb) {
    StepLds L = {};
    int o = 0;
    L.planes = o; o += epb * plane_row_bytes;
    L.champ = o; o += epb * champ_row_bytes;
    L.stats = o; o += epb * 32;
    L.start = o; o += epb * 32;
    L.cnt = o; o += epb * 8;
    L.cur = o; o += epb * 8;
    L.fifo = o; o += fifo ? epb * PCGRL_FIFO_N * 4 : 0;
    L.tag = o; o += fifo ? epb * 4 : 0;
    L.pos = o; o += epb * 2;
    L.act = o; o += epb * 4 * action_width;
    L.flat = o; o += action_width == 3 ? epb * 4 : 0;       // wide representation: the ActionMap wrapper's flat indices (pcgrl_step_flat), decoded into `act` by the update
    L.in_total = o;
    L.info = o; o += epb * 40;
    L.rew = o; o += epb * 8;
    L.done = o; o += epb;
    L.total = (o + 15) & ~15;
    return L;
}

// Block-wide copy of `bytes` bytes, both sides 16-byte aligned (global <-> LDS; every thread of the block calls it).
template <int TPB>
__device__ __forceinline__ void blk_copy(uint8_t* dst, const uint8_t* src, int bytes) {
    const int nv = bytes >> 4;
    for (int i = threadIdx.x; i < nv; i += TPB) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (int i = (nv << 4) + threadIdx.x; i < bytes; i += TPB) dst[i] = src[i];
}
// The same for per-environment rows of `row` bytes (a multiple of 16), only the rows whose flag is set.
template <int TPB>
__device__ __forceinline__ void blk_copy_rows(uint8_t* dst, const uint8_t* src, int rows, int row, const uint8_t* flag) {
    const int per = row >> 4;
    for (int i = threadIdx.x; i < rows * per; i += TPB)
        if (flag[i / per]) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
}

// One segment of the batched prefetch of a FULL block: the 16-byte slots [SLOT0, SLOT0 + NV) of the LDS copy come from `g`.
// Thread tid owns the slots tid + 256 c; everything here is resolved at compile time except the predicate and the address, so
// all loads of all segments are issued back to back and waited for once.
template <int TPB, int C, int SLOT0, int NV>
__device__ __forceinline__ void seg_load_c(uint4& rc, const void* g, bool on, int tid) {
    if (C * TPB + TPB - 1 < SLOT0 || C * TPB >= SLOT0 + NV) return;     // compile time: no overlap
    const int slot = C * TPB + tid;
    if (on && slot >= SLOT0 && slot < SLOT0 + NV) rc = reinterpret_cast<const uint4*>(g)[slot - SLOT0];
}
// (named registers, not an array: an array indexed through a reference ends up in scratch memory)
#define PCGRL_SEG_LOAD(SLOT0, NV, G, ON) do { const void* g_ = (G); const bool on_ = (ON); \
    seg_load_c<TPB, 0, SLOT0, NV>(r0, g_, on_, tid0); seg_load_c<TPB, 1, SLOT0, NV>(r1, g_, on_, tid0); seg_load_c<TPB, 2, SLOT0, NV>(r2, g_, on_, tid0); \
    seg_load_c<TPB, 3, SLOT0, NV>(r3, g_, on_, tid0); seg_load_c<TPB, 4, SLOT0, NV>(r4, g_, on_, tid0); seg_load_c<TPB, 5, SLOT0, NV>(r5, g_, on_, tid0); \
    seg_load_c<TPB, 6, SLOT0, NV>(r6, g_, on_, tid0); seg_load_c<TPB, 7, SLOT0, NV>(r7, g_, on_, tid0); seg_load_c<TPB, 8, SLOT0, NV>(r8, g_, on_, tid0); } while (0)
template <int TPB, int C, int TOTAL>
__device__ __forceinline__ void seg_store_c(uint8_t* smem, const uint4& rc, int tid) {
    if (C * TPB >= TOTAL) return;
    const int slot = C * TPB + tid;
    if (slot < TOTAL) reinterpret_cast<uint4*>(smem)[slot] = rc;
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

// s_setprio with a wave-uniform level (the instruction takes an immediate).  A wavefront that works through a long dependent
// chain -- a certain reset, the dearest full tasks -- issues an instruction every ~9 cycles when it has its SIMD to itself and a
// fraction of that next to three others; with a higher level the arbiter serves it first and the short tasks fill the gaps.
__device__ __forceinline__ void step_set_prio(int p) {
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

// MULTI: the pcgrl_rollout form (loop over the tape); the single-step form is compiled without the loop so that it pays
// nothing for it.  EPB: environments per block, 64 (four wavefronts) or 128 (eight): a block of 128 pools the tasks of twice
// as many environments over twice as many wavefronts, which evens out the spread between blocks -- a block that happens to
// hold four certain resets used to end 10 us after the median block, and the launch ends with the last block.
// OBS (single step, 32-bit row masks, an observation bound whose shape has a lean routine): the images are written while the step
// runs -- a kernel of its own so that the steps without an image carry none of it (the observation view is fourteen scalar registers:
// in zelda's kernel, which is at its limit of 100, they were 80 bytes of scratch and 2.7 us of the bare C3 step).
template <int PROB, int REP, class MaskT, bool MULTI, int EPB, bool OBS = false>
__global__ __launch_bounds__(EPB * 4) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_step(PcgrlParams P, DevBufs Bg, const int32_t* __restrict__ actions, int parity, int gen_map,
                                                                                                     int steps, size_t action_stride, double* reward_out, uint8_t* done_out, int32_t* info_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // the block's state copy, then per wave MT ring + tile bytes (in-kernel resets)
    constexpr int TPB = EPB * 4, NUPD = EPB / 64;                     // threads; wavefronts that do Representation.update
    __shared__ int s_items[3][EPB];     // 0: certain resets, 1: full recomputations (by cost level), 2: incremental updates
    __shared__ int s_n[2][4];           // list lengths and the task ticket of the step, double-buffered by step parity: the
                                        // set of the NEXT step is zeroed while this one runs (no extra barrier)
    __shared__ StepLocal s_loc;
    __shared__ int s_obs_ticket;        // (OBS) the observation tasks' own ticket
    static_assert(!(OBS && MULTI), "the tape form writes its images at the end");
    constexpr int G = 16, GPW = 4;
    constexpr int NPL = (PROB == PCGRL_PROB_BINARY) ? 1 : 3;
    constexpr int kPlaneRow = G * NPL * (int)sizeof(MaskT);
    const int W = P.width, H = P.height;
    TL_INIT(); TL(1);
    const int e0 = blockIdx.x * EPB;
    const int ne = (P.num_envs - e0) < EPB ? (P.num_envs - e0) : EPB;
    const bool has_champ = PROB == PCGRL_PROB_BINARY && Bg.champ != nullptr;
    const bool has_fifo = REP == PCGRL_REP_NARROW && Bg.fifo != nullptr;
    constexpr int kChampRow = PROB == PCGRL_PROB_BINARY ? G * (int)sizeof(MaskT) : 0;
    constexpr int AW = REP == PCGRL_REP_WIDE ? 3 : 1;                 // int32 values of an action
    constexpr StepLds L = step_lds_layout(kPlaneRow, kChampRow, REP == PCGRL_REP_NARROW, AW, EPB);
    const int champ_row = kChampRow;
    // ---- the block's copy of the per-environment state, and a DevBufs for the shared device functions in which index 0 is
    // the block's first environment: the staged arrays point into the LDS copy, the arrays that stay in global memory (byte
    // maps, heatmap, MT19937 rings, tile probabilities, episode statistics) are moved forward to the block's slice.  Inside the
    // block an environment is known by its index 0..EPB-1 only.
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
    const int32_t* act_lds = reinterpret_cast<const int32_t*>(smem + L.act);
    const uint8_t* g_planes = reinterpret_cast<const uint8_t*>(Bg.planes) + (size_t)e0 * kPlaneRow;
    const uint8_t* g_champ = has_champ ? reinterpret_cast<const uint8_t*>(Bg.champ) + (size_t)e0 * champ_row : nullptr;
    if (ne == EPB) {
        // a full block: every load of every segment first, then one wait, then the LDS stores -- one round trip
        static_assert(L.in_total / 16 <= 9 * TPB, "nine 16-byte slots per thread");
        uint4 r0 = {}, r1 = {}, r2 = {}, r3 = {}, r4 = {}, r5 = {}, r6 = {}, r7 = {}, r8 = {};
        const int tid0 = (int)threadIdx.x;
        PCGRL_SEG_LOAD(L.planes / 16, EPB * kPlaneRow / 16, g_planes, true);
        if (kChampRow) PCGRL_SEG_LOAD(L.champ / 16, (kChampRow ? EPB * kChampRow / 16 : 1), g_champ, has_champ);
        PCGRL_SEG_LOAD(L.stats / 16, EPB * 2, Bg.stats + (size_t)e0 * 8, true);
        PCGRL_SEG_LOAD(L.start / 16, EPB * 2, Bg.start_stats + (size_t)e0 * 8, true);
        PCGRL_SEG_LOAD(L.cnt / 16, EPB / 2, Bg.counters + (size_t)e0 * 2, true);
        PCGRL_SEG_LOAD(L.cur / 16, EPB / 2, Bg.rng_cur + (size_t)e0 * 2, true);
        if (REP == PCGRL_REP_NARROW) {
            PCGRL_SEG_LOAD(L.fifo / 16, EPB * PCGRL_FIFO_N * 4 / 16, has_fifo ? Bg.fifo + (size_t)e0 * PCGRL_FIFO_N : nullptr, has_fifo);
            PCGRL_SEG_LOAD(L.tag / 16, EPB / 4, has_fifo ? Bg.fifo_tag + e0 : nullptr, has_fifo);
        }
        PCGRL_SEG_LOAD(L.pos / 16, EPB / 8, Bg.pos + (size_t)e0 * 2, true);
        PCGRL_SEG_LOAD(L.act / 16, EPB / 4 * AW, actions + (size_t)e0 * AW, !(AW == 3 && Bg.flat));
        if (AW == 3) PCGRL_SEG_LOAD(L.flat / 16, (AW == 3 ? EPB / 4 : 1), Bg.flat ? Bg.flat + e0 : nullptr, Bg.flat != nullptr);
        asm volatile("" ::: "memory");      // every load above is issued before the first LDS store below waits for its data
        constexpr int TOT = L.in_total / 16;
        seg_store_c<TPB, 0, TOT>(smem, r0, tid0); seg_store_c<TPB, 1, TOT>(smem, r1, tid0); seg_store_c<TPB, 2, TOT>(smem, r2, tid0);
        seg_store_c<TPB, 3, TOT>(smem, r3, tid0); seg_store_c<TPB, 4, TOT>(smem, r4, tid0); seg_store_c<TPB, 5, TOT>(smem, r5, tid0);
        seg_store_c<TPB, 6, TOT>(smem, r6, tid0); seg_store_c<TPB, 7, TOT>(smem, r7, tid0); seg_store_c<TPB, 8, TOT>(smem, r8, tid0);
    } else {
        blk_copy<TPB>(smem + L.planes, g_planes, ne * kPlaneRow);
        if (has_champ) blk_copy<TPB>(smem + L.champ, g_champ, ne * champ_row);
        blk_copy<TPB>(smem + L.stats, reinterpret_cast<const uint8_t*>(Bg.stats + (size_t)e0 * 8), ne * 32);
        blk_copy<TPB>(smem + L.start, reinterpret_cast<const uint8_t*>(Bg.start_stats + (size_t)e0 * 8), ne * 32);
        blk_copy<TPB>(smem + L.cnt, reinterpret_cast<const uint8_t*>(Bg.counters + (size_t)e0 * 2), ne * 8);
        blk_copy<TPB>(smem + L.cur, reinterpret_cast<const uint8_t*>(Bg.rng_cur + (size_t)e0 * 2), ne * 8);
        blk_copy<TPB>(smem + L.pos, Bg.pos + (size_t)e0 * 2, ne * 2);
        if (has_fifo) {
            blk_copy<TPB>(smem + L.fifo, reinterpret_cast<const uint8_t*>(Bg.fifo + (size_t)e0 * PCGRL_FIFO_N), ne * PCGRL_FIFO_N * 4);
            blk_copy<TPB>(smem + L.tag, reinterpret_cast<const uint8_t*>(Bg.fifo_tag + e0), ne * 4);
        }
        if (AW == 3 && Bg.flat) blk_copy<TPB>(smem + L.flat, reinterpret_cast<const uint8_t*>(Bg.flat + e0), ne * 4);
        else blk_copy<TPB>(smem + L.act, reinterpret_cast<const uint8_t*>(actions + (size_t)e0 * AW), ne * 4 * AW);
    }
    if (threadIdx.x < EPB) { s_loc.dirty[threadIdx.x] = 0; if (OBS) { s_loc.obs_skip[threadIdx.x] = 0; s_loc.late[threadIdx.x] = 0; } }
    if (threadIdx.x < 8) s_n[threadIdx.x >> 2][threadIdx.x & 3] = 0;
    if (threadIdx.x == 0) { s_loc.e0 = 0; s_loc.need = NUPD; s_loc.refill_done[0] = 0; s_loc.refill_done[1] = 0; s_loc.n_late = 0; s_obs_ticket = 0; }
    // ---- the wrapped observation of the block's environments (pcgrl_bind_observation), straight from the LDS copy: the row planes are
    // the map, the cursors are there too -- no byte map is read.  (Only the shapes with a lean routine -- kernels_obs.h: binary tile
    // ids, one-hot over eight tiles -- are written here, the host sends the others to k_obs: Bg.obs.fused.)  Single step: the
    // images are written WHILE the step runs (round 6; until then all of them at the end of the launch: every block ends at about
    // the same time, so the whole image traffic -- 51 MB for 65 536 crops of 28 x 28 -- came after the compute instead of under it):
    //   * an environment that is certain to be reset: by its reset's wavefront, as soon as the new map is there (stats_wave_task);
    //   * the wide representation's map image whose target still holds the previous state (Bg.obs.delta): the one piece a change
    //     touches, by the update wavefront's lane right behind the task barrier;
    //   * every other image: "observation tasks" of OBS_PER_TASK environments each, with a ticket of their own: a wavefront takes one
    //     after every statistics task, so the stores leave from the first tasks' end on instead of all behind the compute;
    //   * an episode end nobody saw coming rewrites planes and cursor under such a task: those few images are written again at the
    //     end, behind a barrier before which every wavefront has waited for its own stores.
    constexpr int OBS_PER_TASK = 8;
    // (the host launches the OBS instantiation only with Bg.obs.out && Bg.obs.fused)
    const bool obs_delta = OBS && REP == PCGRL_REP_WIDE && NPL == 3 && Bg.obs.delta;
    const int n_obs = (OBS && !obs_delta) ? (ne + OBS_PER_TASK - 1) / OBS_PER_TASK : 0;
    const ObsPlanes<MaskT, NPL> obs_src = {reinterpret_cast<const MaskT*>(smem + L.planes), G};
    if (OBS && threadIdx.x == 0) s_loc.obs_v = obs_view(P, Bg.obs, e0);       // (visible behind the barrier at the top of the step)
    if (PROB == PCGRL_PROB_ZELDA && threadIdx.x == 64) zelda_reward_tab(P, &s_loc.zr);
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
    const int sp = t & 1;                   // this step's set of list counters
    DevGroup<G, MaskT> g(lane64);
    if (MULTI && t > 0) {   // this step's row of the action tape (the first one came with the state).  A row starts at a multiple
                            // of num_envs * AW ints -- 4-byte aligned only (65 environments, narrow) -- so it is copied word by word.
        const int32_t* row = actions + (size_t)t * action_stride + (size_t)e0 * AW;
        for (int i = tid; i < ne * AW; i += TPB) reinterpret_cast<int32_t*>(smem + L.act)[i] = row[i];
    }
    __syncthreads();                        // the state copy is complete; everything the previous step wrote is visible to the whole block
    TL(17);
    const int prio = B.step_prio, prio_nfull = (prio >> 8) & 15;
    int prio_now = 0;                       // the level this wavefront is at: s_setprio only when it changes (the instruction is not free)
    if (wv < NUPD) {
        if ((prio >> 6) & 3) { prio_now = (prio >> 6) & 3; step_set_prio(prio_now); }
        const int e = wv * 64 + lane64;                        // block-local index (see B above)
        UpdateOut u = {};
        UpdateMid mid = {};
        if (REP == PCGRL_REP_WIDE && !MULTI && Bg.flat && e < ne) {
            // ActionMap.step (wrappers.py:139-154) folded in: flat index into (H, W, tiles) -> (x, y, tile), clamped and reported like k_action_map
            const int dim = P.ntiles, total = W * H * dim;
            int a = reinterpret_cast<const int32_t*>(smem + L.flat)[e];
            if (a < 0 || a >= total) atomicOr(B.status, PCGRL_STATUS_BAD_ACTION);
            a = a < 0 ? 0 : (a >= total ? total - 1 : a);
            int32_t* t3 = reinterpret_cast<int32_t*>(smem + L.act) + 3 * e;
            const int q = a / dim;
            t3[0] = q % W; t3[1] = q / W; t3[2] = a - q * dim;
        }
        if (e < ne) u = update_env<REP, MaskT, true, true>(P, B, act_lds, e, &mid);      // the decision part: everything the task lists need
        TL(2);
        const bool first = u.rst || u.sure_done;               // reset-only, or certain to end: k_stats' "lone" items
        const bool packed_full = PROB == PCGRL_PROB_ZELDA && B.zelda_inc;
        int dest = -1, v = e;
        if (first) { dest = 0; v = u.rst ? (e | WL_RESET_ONLY) : e; }
        // (binary: a change in or next to the champion goes to the full list as a packed item with bit 31 set: binary_touch first)
        else if (u.chg) { dest = u.cheap ? 2 : 1; v = (u.cheap || packed_full) ? u.inc_item : ((u.touch && B.step_touch) ? (u.inc_item | (int)0x80000000) : e); }
        if (u.chg) s_loc.dirty[e] = 1;
        if (OBS && first && e < EPB) s_loc.obs_skip[e] = 1;
        const uint64_t m0 = __ballot(dest == 0), m1 = __ballot(dest == 1), m2 = __ballot(dest == 2);
        const uint64_t below = (1ull << lane64) - 1ull;
        // this wavefront's stretch of each list (one LDS atomic per list)
        int b0 = 0, b1 = 0, b2 = 0;
        if (lane64 == 0) {
            b0 = m0 ? atomicAdd(&s_n[sp][0], __popcll(m0)) : 0;
            b1 = m1 ? atomicAdd(&s_n[sp][1], __popcll(m1)) : 0;
            b2 = m2 ? atomicAdd(&s_n[sp][2], __popcll(m2)) : 0;
            if (wv == 0) { s_n[sp ^ 1][0] = 0; s_n[sp ^ 1][1] = 0; s_n[sp ^ 1][2] = 0; s_n[sp ^ 1][3] = 0; s_loc.refill_done[sp ^ 1] = 0; s_loc.par = sp; }
        }
        b0 = __builtin_amdgcn_readfirstlane(b0); b1 = __builtin_amdgcn_readfirstlane(b1); b2 = __builtin_amdgcn_readfirstlane(b2);
        if (dest == 0) s_items[0][b0 + __popcll(m0 & below)] = v;
        if (dest == 2) s_items[2][b2 + __popcll(m2 & below)] = v;
        // full recomputations ordered by expected cost (the four maps that share a wavefront should cost about the same):
        // eight levels of the previous path length, ranked with eight ballots -- no LDS round trips
        {
            const int lvl = (u.bucket >> 3) & 7;
            int pos1 = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const uint64_t mb = __ballot(dest == 1 && lvl == b);
                pos1 += b > lvl ? __popcll(mb) : (b == lvl ? __popcll(mb & below) : 0);      // dearest first: the cheap ones fill the gaps at the end
            }
            if (dest == 1) s_items[1][b1 + pos1] = v;
        }
        // LDS-only barrier: what the update wavefronts have in flight to global memory (byte-map cells, heatmap increments)
        // concerns no task that starts now -- environments that are certain to be reset got no such write, the other tasks work
        // on the LDS copy -- and is waited for below, before refill_done is published.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // behind the barrier, while the other wavefronts compute: the cursor move of the step (narrow: the draws, the new cursor, the
        // heat-map cell it marks), then the consumed draws go into the rings and the draw caches are topped up.  Environments that
        // are certain to be reset are left alone: their reset consumes the step's draws and rewrites everything.
        int k_used = 0, cur0 = 0;
        if (lane64 == 0) s_loc.late_done[wv] = 0;
        if (e < ne && !first) update_env_cursor<REP, MaskT>(P, B, e, mid, k_used, cur0);
        if (OBS && REP == PCGRL_REP_WIDE && NPL == 3 && obs_delta && e < ne && !first && u.chg) obs_write_delta_piece(obs_src, obs_view_from_lds(&s_loc.obs_v), e, act_lds[3 * e], act_lds[3 * e + 1]);
        // the consumed draws stay in the draw cache for now (StepLocal::pend): writing them to the rings and topping the caches up
        // is ring traffic nothing waits for, and this wavefront is a quarter of the block's capacity for the tasks -- it does that
        // after the task loop (below)
        if (e < EPB) s_loc.pend[e] = (has_fifo && e < ne && !first) ? k_used : 0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // byte-map cells and heatmap increments have landed (an episode end nobody saw coming rewrites both)
        if (lane64 == 0) __hip_atomic_fetch_add(&s_loc.refill_done[sp], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // (the other wavefronts take that barrier inside the first round of the task loop below: what the compiler hoists out of the
    //  loop -- scalar loads of the parameter block, address arithmetic of every task kind: ~2.4 us on the timeline -- then runs
    //  while they would only be waiting for the lists)
    int n0 = 0, n1 = 0, n2 = 0, w_full = 0, w_total = 0, w_lone = 0;
    bool stats_left = true, obs_left = OBS;
    bool pair = false;      // two certain resets per wavefront task (four statistics side by side) from DevBufs::step_pair of them on
    // maps per wavefront task (of the GPW = 4 lane groups): the four maps of a task run their component and sweep loops in lockstep,
    // so a task lasts as long as its slowest map in every phase, and a block has only about one and a half tasks per wavefront --
    // fewer maps per task give shorter chains and more units to balance (pcgrl_tuning full_per_wave / inc_per_wave)
    const int fpw = B.step_fpw, ipw = B.step_ipw;
    const int tiles_bytes = (W * H + 15) & ~15;
    uint32_t* mt = reinterpret_cast<uint32_t*>(reset_scratch + (size_t)wv * (PCGRL_MT_N * 4 + tiles_bytes));
    uint8_t* tiles = reinterpret_cast<uint8_t*>(mt + PCGRL_MT_N);
    const MaskT rowmask = row_valid<MaskT>(g.lane, W, H);
    const bool zinc = PROB == PCGRL_PROB_ZELDA && sizeof(MaskT) == 4 && B.zelda_inc;
    // Tasks in the order certain resets (the longest chains), full recomputations, incremental updates; a wavefront takes
    // the next one whenever it is free, so the block ends when the work is done, not when its unluckiest wavefront is
    // (a static split left the last wavefront of a block ~10 us behind the others).
    for (bool first_round = true;; first_round = false) {
        if (first_round) {
            if (wv >= NUPD) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            TL(3);
            n0 = s_n[sp][0]; n1 = s_n[sp][1]; n2 = s_n[sp][2];
            w_full = (n1 + fpw - 1) / fpw;
            // (zelda only, at compile time: a step of the binary problem has two or three certain resets a block, and a second inlined
            //  copy of the reset cost its kernel 1 400 instructions and half again as many scalar-register spills: + 4 % vector instructions)
            pair = PROB == PCGRL_PROB_ZELDA && B.step_pair > 0 && n0 >= B.step_pair;
            w_lone = pair ? (n0 + 1) >> 1 : n0;
            w_total = w_lone + w_full + (n2 + ipw - 1) / ipw;
        }
        if (OBS && !first_round && obs_left) {
            // an observation task between two statistics tasks: the images of OBS_PER_TASK environments.  The cursors are final once the
            // update wavefronts are through with the cursor moves of the step (refill_done: long the case after a first task); rows and
            // cursor of an environment that is being reset under this task give a torn image, which the end of the launch writes
            // again (StepLocal::late).
            int oid = 0;
            if (lane64 == 0) oid = atomicAdd(&s_obs_ticket, 1);
            oid = __builtin_amdgcn_readfirstlane(oid);
            if (oid >= n_obs) obs_left = false;
            else {
                if (prio_now) { prio_now = 0; step_set_prio(0); }
                while (__hip_atomic_load(&s_loc.refill_done[sp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < NUPD) __builtin_amdgcn_s_sleep(2);
                const int o0 = oid * OBS_PER_TASK;
                const ObsView V = obs_view_from_lds(&s_loc.obs_v);
#pragma clang loop unroll(disable)
                for (int j = 0; j < OBS_PER_TASK; j++) {
                    const int eo = o0 + j;                    // wave-uniform
                    if (eo < ne && !s_loc.obs_skip[eo]) obs_write_env_lean(obs_src, V, smem + L.pos, eo, lane64);
                }
            }
        }
        if (!stats_left) { if (!OBS || !obs_left) break; continue; }
        int wid = 0;
        if (lane64 == 0) wid = atomicAdd(&s_n[sp][3], 1);
        wid = __builtin_amdgcn_readfirstlane(wid);
        TL(18);
        if (wid >= w_total) { stats_left = false; if (!OBS || !obs_left) break; continue; }
        const bool lone = wid < w_lone, inc = wid >= w_lone + w_full;
        const int item = lone ? (pair ? 2 * wid + (gw >> 1) : wid) : (inc ? (wid - w_lone - w_full) * ipw + gw : (wid - w_lone) * fpw + gw);
        const bool have = lone ? (item < n0 && (pair || gw < 2)) : (gw < (inc ? ipw : fpw) && item < (inc ? n2 : n1));
        const int raw = have ? s_items[lone ? 0 : (inc ? 2 : 1)][item] : 0;
        TL(lone ? 4 : (inc ? 6 : 5));
        if (prio) {
            const int want = lone ? (prio & 3) : (inc ? ((prio >> 4) & 3) : ((prio_nfull == 0 || wid - w_lone < prio_nfull) ? ((prio >> 2) & 3) : 0));
            if (want != prio_now) { prio_now = want; step_set_prio(want); }
        }
        stats_wave_task<PROB, G, MaskT, true>(P, B, g, lane64, gw, lone, inc, PROB == PCGRL_PROB_ZELDA && pair, zinc, have, raw, lane64, MODE_STEP, parity, 1, gen_map, mt, tiles, rowmask, &s_loc, OBS);
        TL(7);
    }
    if (prio_now) { prio_now = 0; step_set_prio(0); }
    if (has_fifo && wv < NUPD) {
        // the draws of the step go into the rings and the draw caches are topped up, for the environments whose words no reset
        // has taken in the meantime (StepLocal::pend)
        const int e = wv * 64 + lane64;
        int k = 0;
        if (e < ne) k = atomicExch(&s_loc.pend[e], -1);
        if (k > 0) {
            int c0 = B.rng_cur[2 * e] - k;                      // the cursor before the step's draws
            c0 = c0 < 0 ? c0 + PCGRL_MT_N : c0;
            fifo_refill(B, e, k, c0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane64 == 0) __hip_atomic_store(&s_loc.late_done[wv], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    TL(8);
    if (MULTI && (reward_out || done_out || info_out)) {   // kernel-uniform: the per-step outputs of the block's environments, row t
        __syncthreads();
        const size_t row = (size_t)t * P.num_envs + e0;
        if (reward_out && tid < ne) reward_out[row + tid] = reinterpret_cast<const double*>(smem + L.rew)[tid];
        if (done_out && tid >= TPB / 2 && tid - TPB / 2 < ne) done_out[row + tid - TPB / 2] = smem[L.done + tid - TPB / 2];
        if (info_out) for (int i = tid; i < ne * 10; i += TPB) info_out[row * 10 + i] = reinterpret_cast<const int32_t*>(smem + L.info)[i];
    }
  }
    // ---- the block's state goes back, coalesced; plane rows, champion rows and start statistics only where they changed
    __syncthreads();
    blk_copy<TPB>(reinterpret_cast<uint8_t*>(Bg.stats + (size_t)e0 * 8), smem + L.stats, ne * 32);
    blk_copy<TPB>(reinterpret_cast<uint8_t*>(Bg.info + (size_t)e0 * 10), smem + L.info, ne * 40);
    blk_copy<TPB>(reinterpret_cast<uint8_t*>(Bg.reward + e0), smem + L.rew, ne * 8);
    blk_copy<TPB>(reinterpret_cast<uint8_t*>(Bg.counters + (size_t)e0 * 2), smem + L.cnt, ne * 8);
    blk_copy<TPB>(reinterpret_cast<uint8_t*>(Bg.rng_cur + (size_t)e0 * 2), smem + L.cur, ne * 8);
    blk_copy<TPB>(Bg.pos + (size_t)e0 * 2, smem + L.pos, ne * 2);
    blk_copy<TPB>(Bg.done + e0, smem + L.done, ne);
    if (has_fifo) {
        blk_copy<TPB>(reinterpret_cast<uint8_t*>(Bg.fifo + (size_t)e0 * PCGRL_FIFO_N), smem + L.fifo, ne * PCGRL_FIFO_N * 4);
        blk_copy<TPB>(reinterpret_cast<uint8_t*>(Bg.fifo_tag + e0), smem + L.tag, ne * 4);
    }
    blk_copy_rows<TPB>(reinterpret_cast<uint8_t*>(Bg.planes) + (size_t)e0 * kPlaneRow, smem + L.planes, ne, kPlaneRow, s_loc.dirty);
    if (has_champ) blk_copy_rows<TPB>(reinterpret_cast<uint8_t*>(Bg.champ) + (size_t)e0 * champ_row, smem + L.champ, ne, champ_row, s_loc.dirty);
    blk_copy_rows<TPB>(reinterpret_cast<uint8_t*>(Bg.start_stats + (size_t)e0 * 8), smem + L.start, ne, 32, s_loc.dirty);
    // ---- the wrapped observation of the block's environments (pcgrl_bind_observation), straight from the LDS copy: the row
    // planes are the map, the cursors are there too -- no byte map is read.  A store stream that overlaps with the blocks that
    // are still computing (the step is latency-bound, the memory system nearly idle).
    // (only the shapes with a lean routine -- kernels_obs.h: binary tile ids, one-hot over eight tiles -- are written here, the
    // host sends the others to k_obs: Bg.obs.fused.  Wide representation, single step, the target still holding the previous
    // image: only what changed, Bg.obs.delta.)
    if (!OBS && Bg.obs.out && Bg.obs.fused) {        // (the tape form, 64-bit row masks, pcgrl_tuning obs_at_end: all images here)
        const ObsPlanes<MaskT, NPL> src = {reinterpret_cast<const MaskT*>(smem + L.planes), G};
        obs_write_block_lean(src, obs_view(P, Bg.obs, e0), smem + L.pos, ne, REP == PCGRL_REP_WIDE && !MULTI && Bg.obs.delta, act_lds, s_loc.dirty,
                             smem + L.done, (int)threadIdx.x, TPB);
    }
    if (OBS && s_loc.n_late) {                    // block-uniform (read behind the barrier above)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the stores of this wavefront's observation tasks have landed ...
        __syncthreads();                                          // ... everybody's have
        const ObsView V = obs_view_from_lds(&s_loc.obs_v);
        for (int eo = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6); eo < ne; eo += TPB >> 6)
            if (s_loc.late[eo]) obs_write_env_lean(obs_src, V, smem + L.pos, eo, (int)threadIdx.x & 63);
    }
}
