// Issue-rate and latency calibration for the step kernels' roofline (MI355X): what a wavefront pays per instruction, alone and
// in company -- the numbers behind `valu_issue_peak` in bench.py and behind the per-pop budgets of the search kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_calib.hip -o /tmp/valu_calib && /tmp/valu_calib
// For W = 1, 2, 4, 8 wavefronts per SIMD (grid = 256 CUs x W blocks of 256 threads) each kernel runs a chain of N
// instructions per wavefront; reported: chip-wide wave-instructions per second (HIP events) and cycles per instruction as one
// wavefront sees them (s_memtime around the chain, 100 MHz constant clock converted with the measured kernel time).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define N_ITER 2048
#define UNROLL 16

// dependent chain of full-rate VALU instructions (v_add_u32 on one register)
__global__ void k_valu_dep(uint32_t* out, uint32_t y) {
    uint32_t x = threadIdx.x;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
        for (int k = 0; k < UNROLL; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
    if (x == 0x12345u) out[0] = x;
}
// eight independent chains: nothing but issue limits a wavefront
__global__ void k_valu_indep(uint32_t* out, uint32_t y) {
    uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
        for (int k = 0; k < UNROLL / 8; k++)
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(y));
    }
    a ^= b ^ c ^ d ^ e ^ f ^ g ^ h;
    if (a == 0x12345u) out[0] = a;
}
// dependent chain of 32-bit integer multiplies (v_mul_lo_u32: not full rate)
__global__ void k_mul_dep(uint32_t* out, uint32_t y) {
    uint32_t x = threadIdx.x | 1;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
        for (int k = 0; k < UNROLL; k++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
    if (x == 0x12345u) out[0] = x;
}
// dependent chain of scalar instructions (s_add_u32)
__global__ void k_salu_dep(uint32_t* out, uint32_t y) {
    uint32_t x = blockIdx.x;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
        for (int k = 0; k < UNROLL; k++) asm volatile("s_add_u32 %0, %0, %1" : "+s"(x) : "s"(y) : "scc");
    }
    if (x == 0x12345u) out[0] = x;
}
// dependent chain of LDS reads (pointer chase through a 64-word table per wavefront, all lanes the same address: a broadcast read)
__global__ void k_lds_dep(uint32_t* out, uint32_t y) {
    __shared__ uint32_t tab[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    tab[wv][lane] = (uint32_t)((lane * 5 + 1 + y) & 63) * 4u;       // byte offsets of the next hop
    __syncthreads();
    uint32_t p = (uint32_t)(uintptr_t)(&tab[wv][0]) & 0xFFFFu, o = 0;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            uint32_t a = p + o;
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(o) : "v"(a) : "memory");
        }
    }
    if (o == 0x12345u) out[0] = o;
}
typedef void (*kern_t)(uint32_t*, uint32_t);
int main() {
    uint32_t* out;
    (void)hipMalloc(&out, 256);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    fflush(stdout);
    struct { const char* name; kern_t k; double per_iter; } K[] = {
        {"VALU v_add_u32, dependent chain", k_valu_dep, UNROLL}, {"VALU v_add_u32, 8 independent chains", k_valu_indep, UNROLL},
        {"VALU v_mul_lo_u32, dependent chain", k_mul_dep, UNROLL}, {"SALU s_add_u32, dependent chain", k_salu_dep, UNROLL},
        {"LDS ds_read_b32 + wait, dependent chain", k_lds_dep, UNROLL}};
    printf("| chain | wavefronts / SIMD | chip wave-instr/s | cycles / instr seen by one wavefront (2.4 GHz) |\n|---|---|---|---|\n");
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (auto& kk : K)
        for (int w = 1; w <= 8; w *= 2) {
            const dim3 grid(cus * w), block(256);
            kk.k<<<grid, block>>>(out, 3);       // warm-up
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int r = 0; r < 5; r++) kk.k<<<grid, block>>>(out, 3);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double t = ms * 1e-3 / 5, instr = (double)N_ITER * kk.per_iter;
            const double waves = (double)cus * w * 4;
            printf("| %s | %d | %.3e | %.2f |\n", kk.name, w, waves * instr / t, t * 2.4e9 / instr);
            fflush(stdout);
        }
    return 0;
}
