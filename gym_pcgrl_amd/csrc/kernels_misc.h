// Action-map decode (wrappers) and small utility kernels; the observation image is kernels_obs.h.
// Part of the single translation unit pcgrl_abi.hip (see its header comment for the overall picture).
#pragma once
// ActionMap.step for the wide representation (wrappers.py:139-154): flat index into (h, w, tiles) -> (x, y, tile)
__global__ void k_action_map(const int32_t* __restrict__ flat, int32_t* __restrict__ xyv, int n, int w, int h, int dim, int32_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int a = flat[i];
        if (a < 0 || a >= w * h * dim) atomicOr(status, PCGRL_STATUS_BAD_ACTION);     // clamped and reported
        a = a < 0 ? 0 : (a >= w * h * dim ? w * h * dim - 1 : a);
        const int v = a % dim, x = (a / dim) % w, y = a / (dim * w);
        xyv[3 * i] = x; xyv[3 * i + 1] = y; xyv[3 * i + 2] = v;
    }
}

// pcgrl_selftest_range_reward: helper.py:366-376 in the integer form every reward of the step kernels goes through (range_reward_i,
// pcgrl_algos.h), on the device, one thread per row of a table (new value, old value, low, high).
__global__ void k_selftest_range_reward(const int32_t* __restrict__ rows, int n, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = range_reward_i(rows[4 * i], rows[4 * i + 1], rows[4 * i + 2], rows[4 * i + 3]);
}

__global__ void k_fill_all(DevBufs B, int n, int parity, int list) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) B.wl_items[list][(size_t)(i & (WL_NSHARD - 1)) * B.wl_cap[list] + (i >> 6)] = i;
    if (i < WL_NSHARD) wl_counters(B, parity, list)[i * WL_CSTRIDE] = (n - i + WL_NSHARD - 1) / WL_NSHARD;
}
// pcgrl_async_flush: the environments whose search ended their episode in the last tick (kernels_search_async.h ASYNC_PEND_RESET)
// go on `list` -- a tick's k_update would have done it
__global__ void k_async_collect(DevBufs B, uint8_t* pending, int n, int parity, int list) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && pending[e] == 2) { pending[e] = 0; wl_push(B, parity, list, e & (WL_NSHARD - 1), e); }
    async_list_runnable(B, e);
}
__global__ void k_bcast_tile_p(double* tile_p, int n, double p0, double p1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { tile_p[2 * i] = p0; tile_p[2 * i + 1] = p1; }
}
__global__ void k_zero_cursors(int32_t* cur, int first, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { cur[2 * (first + i)] = 0; cur[2 * (first + i) + 1] = 0; }
}

// MT19937 init_by_array (Matsumoto & Nishimura 2002; numpy RandomState.seed(list) as used by gym's seeding.np_random,
// reps/representation.py:28-30, probs/problem.py:34-36) for `count` environments, thread per environment: the key words
// (1 or 2) and their number were uploaded into words 0..2 of the environment's ring.  1 247 dependent steps per
// environment on its own 2.5 KB ring in global memory; environments run side by side.
__device__ uint32_t g_mt_genrand[PCGRL_MT_N];      // init_genrand(19650218), the same for every key: filled by the host once
__global__ void k_init_by_array(uint32_t* __restrict__ rng_rep, uint32_t* __restrict__ rng_prob, int32_t* __restrict__ cur, int first, int count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const size_t e = (size_t)first + t;
    uint32_t* mt = rng_rep + e * PCGRL_MT_N;
    const uint32_t key[2] = {mt[0], mt[1]};
    const int klen = (int)mt[2] == 1 ? 1 : 2;
    for (int i = 0; i < PCGRL_MT_N; i++) mt[i] = g_mt_genrand[i];
    uint32_t prev = g_mt_genrand[0];
    int i = 1, j = 0;
    for (int k = 0; k < PCGRL_MT_N; k++) {
        prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        mt[i] = prev;
        i++; j++;
        if (i >= PCGRL_MT_N) { mt[0] = prev; i = 1; }
        if (j >= klen) j = 0;
    }
    for (int k = 0; k < PCGRL_MT_N - 1; k++) {
        prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
        mt[i] = prev;
        i++;
        if (i >= PCGRL_MT_N) { mt[0] = prev; i = 1; }
    }
    mt[0] = 0x80000000u;
    if (rng_prob) {
        uint32_t* mp = rng_prob + e * PCGRL_MT_N;
        for (int q = 0; q < PCGRL_MT_N; q++) mp[q] = mt[q];
    }
    cur[2 * e] = 0; cur[2 * e + 1] = 0;
}


