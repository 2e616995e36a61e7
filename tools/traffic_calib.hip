// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access patterns of the PCGRL kernels (MI355X_MICROARCH.md,
// "HBM": only wide coalesced streaming reads are calibrated there).  Each kernel touches a 1 GiB buffer (beyond the
// 256 MiB Infinity Cache) in a known pattern; tools/calibrate_traffic.sh runs this under `rocprofv3 --pmc FETCH_SIZE` /
// `--pmc WRITE_SIZE` and prints reported / expected bytes per kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/traffic_calib.hip -o /tmp/traffic_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// one 4-byte word per `stride` bytes, consecutive lanes `stride` apart (stride 64/128: one word per line; 4: coalesced)
__global__ void calib_read_u32(const uint32_t* __restrict__ buf, size_t nwords_touched, size_t stride_words, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < nwords_touched; i += (size_t)gridDim.x * blockDim.x) acc += buf[i * stride_words];
    if (acc == 0x12345678u) out[0] = acc;     // never true for the zeroed buffer; keeps the loads alive
}
// the same with a pseudo-random line order (what per-environment records look like to the memory system)
__global__ void calib_read_u32_scattered(const uint32_t* __restrict__ buf, size_t nlines, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) {
        const size_t line = (i * 2654435761ull) % nlines;      // odd multiplier: a permutation when nlines is a power of two
        acc += buf[line * 32];                                   // 128-byte lines
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_read_u128(const uint4* __restrict__ buf, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = buf[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_write_u32(uint32_t* __restrict__ buf, size_t nwords_touched, size_t stride_words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nwords_touched; i += (size_t)gridDim.x * blockDim.x) buf[i * stride_words] = (uint32_t)i;
}
__global__ void calib_write_u8_scattered(uint8_t* __restrict__ buf, size_t nlines) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) buf[((i * 2654435761ull) % nlines) * 128] = (uint8_t)i;
}

int main() {
    const size_t bytes = 1ull << 30;
    uint8_t* buf; uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes); hipMemset(out, 0, 256);
    hipDeviceSynchronize();
    const dim3 grid(4096), block(256);
    // name, expected bytes (algorithmic words) -- the shell script pairs them with the counters by kernel name and order
    for (int rep = 0; rep < 3; rep++) {
        calib_read_u128<<<grid, block>>>((const uint4*)buf, bytes / 16, out);                       // 1 GiB wide coalesced
        calib_read_u32<<<grid, block>>>((const uint32_t*)buf, bytes / 4, 1, out);                   // 1 GiB, 4 B per lane coalesced
        calib_read_u32<<<grid, block>>>((const uint32_t*)buf, bytes / 64, 16, out);                 // one word per 64 B
        calib_read_u32<<<grid, block>>>((const uint32_t*)buf, bytes / 128, 32, out);                // one word per 128 B
        calib_read_u32_scattered<<<grid, block>>>((const uint32_t*)buf, bytes / 128, out);          // one word per 128 B line, random order
        calib_write_u32<<<grid, block>>>((uint32_t*)buf, bytes / 4, 1);                             // 1 GiB coalesced
        calib_write_u32<<<grid, block>>>((uint32_t*)buf, bytes / 128, 32);                          // one word per 128 B
        calib_write_u8_scattered<<<grid, block>>>(buf, bytes / 128);                                // one byte per 128 B line, random order
        hipDeviceSynchronize();
    }
    printf("calibration kernels done\n");
    return 0;
}
