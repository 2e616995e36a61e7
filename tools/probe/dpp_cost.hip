// Developer probe (round 5): what does one DEPENDENT cross-lane move cost a lone wavefront on gfx950?  The BFS level of a 64-row map
// (lanegroup_dev.h DevGroup<64>) moves its frontier one lane up and one lane down with DPP wave_shr:1 / wave_shl:1 on both halves of a
// 64-bit mask; a level measures ~240 cycles for 18 instructions (profiles/r4_round4/NOTES.md).  Which of them are the expensive ones?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
template <int MODE>
__global__ void k(unsigned* out, long long* cyc) {
    unsigned v = threadIdx.x * 2654435761u + 1u;
    const long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; i++) {
        unsigned w;
        if (MODE == 0) w = v;                                                                         // plain dependent xor-add chain
        else if (MODE == 1) w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
        else if (MODE == 2) w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);   // wave_shr:1
        else if (MODE == 3) w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);   // wave_shl:1
        else if (MODE == 4) w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xF, 0xF, true);   // row_bcast:15
        else if (MODE == 5) w = (unsigned)__builtin_amdgcn_ds_bpermute(((threadIdx.x + 63) & 63) << 2, (int)v);
        else if (MODE == 6) w = (unsigned)__shfl_up((int)v, 1, 64);
        else w = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x101, 0xF, 0xF, true);                  // row_shl:1
        v = (w ^ (v + 0x9E3779B9u));
    }
    const long long t1 = clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 256); hipMalloc(&cyc, 64);
    hipMemset(cyc, 0, 64);
    k<0><<<1, 64>>>(out, cyc); k<1><<<1, 64>>>(out, cyc); k<2><<<1, 64>>>(out, cyc); k<3><<<1, 64>>>(out, cyc);
    k<4><<<1, 64>>>(out, cyc); k<5><<<1, 64>>>(out, cyc); k<6><<<1, 64>>>(out, cyc); k<7><<<1, 64>>>(out, cyc);
    long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char* nm[8] = {"no cross-lane move (xor + add)", "row_shr:1", "wave_shr:1", "wave_shl:1", "row_bcast:15", "ds_bpermute", "__shfl_up", "row_shl:1"};
    for (int i = 0; i < 8; i++) printf("%-32s %6.1f cycles per dependent step (one lone wavefront)\n", nm[i], (double)h[i] / N);
    return 0;
}
