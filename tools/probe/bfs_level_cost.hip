// Developer probe (round 5, last session): what does ONE BFS level of a 64-row map cost a wavefront -- alone on its SIMD, and next to
// one / three other sweeping wavefronts?  k_stats_wide (C5) is as long as the longest double sweep of a step (paths of up to ~180 cells
// on 64 x 64 maps: ~350 levels), so cycles per level is the number that sets C5.  The product's loops, from the product's headers:
//   mode 0  bfs_levels<true>  DevGroup<64, uint64_t>   the compiler's loop (four levels per exit test)
//   mode 1  bfs_levels<false> DevGroup<64, uint64_t>   the same without the "last frontier" bookkeeping (second sweep of a double sweep)
//   mode 2  bfs_levels<true>  the written-out loop of bfs_asm.h on 64-bit masks (kHistBfs = 1; not used by the product)
//   mode 3  bfs_levels<false> the same, second-sweep form
//   mode 4/5 the 32-bit-mask forms (maps of at most 32 columns): compiler / written out
// (block of 4 / 8 / 16 wavefronts = 1 / 2 / 4 per SIMD, all sweeping; the slowest wavefront's time is reported)
// Map: a serpentine -- even rows open, odd rows open in one end cell, alternating ends -- so a sweep from the first cell runs ~2 000 levels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gym_pcgrl_amd/csrc tools/probe/bfs_level_cost.hip -o tools/probe/bfs_level_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "lanegroup_dev.h"
#include "pcgrl_algos.h"

template <class M>
struct GroupHist : DevGroup<64, M> {
    enum { kGroup = 64, kLog2Group = 4, kHistBfs = 1 };
};
template <class M>
__device__ M serpentine(int lane) {
    const int bits = (int)sizeof(M) * 8;
    if ((lane & 1) == 0) return ~(M)0;
    return (lane & 3) == 1 ? (M)1 << (bits - 1) : (M)1;
}
template <int MODE>
__global__ void k(int* ecc_out, long long* cyc, int slot) {
    const int lane = threadIdx.x & 63;
    long long t0 = 0, t1 = 0;
    int ecc = 0;
    if (MODE < 4) {
        typedef uint64_t M;
        const M pass = serpentine<M>(lane);
        const M src = lane == 0 ? (M)1 : (M)0;
        M last = 0;
        if (MODE < 2) {
            DevGroup<64, M> g;
            t0 = clock64();
            ecc = MODE == 0 ? bfs_levels<true>(g, src, pass, last) : bfs_levels<false>(g, src, pass, last);
            t1 = clock64();
        } else {
            GroupHist<M> g;
            t0 = clock64();
            ecc = MODE == 2 ? bfs_levels<true>(g, src, pass, last) : bfs_levels<false>(g, src, pass, last);
            t1 = clock64();
        }
        ecc += (int)(last & 1);
    } else {
        typedef uint32_t M;
        const M pass = serpentine<M>(lane);
        const M src = lane == 0 ? (M)1 : (M)0;
        M last = 0;
        if (MODE == 4) {
            DevGroup<64, M> g;
            t0 = clock64();
            ecc = bfs_levels<true>(g, src, pass, last);
            t1 = clock64();
        } else {
            GroupHist<M> g;
            t0 = clock64();
            ecc = bfs_levels<true>(g, src, pass, last);
            t1 = clock64();
        }
        ecc += (int)(last & 1);
    }
    // the SLOWEST wavefront of the block counts (the issue arbiter serves the oldest first: wavefront 0 alone would look unshared)
    if (lane == 0 && blockIdx.x == 0) { atomicMax((unsigned long long*)&cyc[slot], (unsigned long long)(t1 - t0)); ecc_out[slot] = ecc; }
}
template <int MODE>
void run(const char* name, int* ecc, long long* cyc, int slot0) {
    const int nw[4] = {1, 4, 8, 16};
    long long hc[4]; int he[4];
    for (int i = 0; i < 4; i++) {
        k<MODE><<<1, 64 * nw[i]>>>(ecc, cyc, slot0 + i);     // warm (code fetch)
        hipDeviceSynchronize();
        hipMemset(cyc + slot0 + i, 0, sizeof(long long));
        k<MODE><<<1, 64 * nw[i]>>>(ecc, cyc, slot0 + i);
    }
    hipDeviceSynchronize();
    hipMemcpy(hc, cyc + slot0, sizeof(hc), hipMemcpyDeviceToHost);
    hipMemcpy(he, ecc + slot0, sizeof(he), hipMemcpyDeviceToHost);
    printf("%-58s levels %5d   cycles/level: alone %6.1f | 1 per SIMD %6.1f | 2 per SIMD %6.1f | 4 per SIMD %6.1f\n", name, he[0],
           (double)hc[0] / he[0], (double)hc[1] / he[1], (double)hc[2] / he[2], (double)hc[3] / he[3]);
}
int main() {
    int* ecc; long long* cyc;
    hipMalloc(&ecc, 64 * 4); hipMalloc(&cyc, 64 * 8);
    hipMemset(cyc, 0, 64 * 8);
    run<0>("64-bit rows, compiler loop, first sweep (keeps last)", ecc, cyc, 0);
    run<1>("64-bit rows, compiler loop, second sweep", ecc, cyc, 4);
    run<2>("64-bit rows, written-out loop (bfs_asm.h), first sweep", ecc, cyc, 8);
    run<3>("64-bit rows, written-out loop (bfs_asm.h), second sweep", ecc, cyc, 12);
    run<4>("32-bit rows, compiler loop, first sweep", ecc, cyc, 16);
    run<5>("32-bit rows, written-out loop, first sweep", ecc, cyc, 20);
    return 0;
}
