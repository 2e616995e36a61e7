// Developer probe: how many kernels from different HIP streams run at the same time on this GPU / runtime configuration?
// spin_launch(stream, cycles, blocks): a kernel whose blocks busy-wait `cycles` ticks of the 100 MHz wall clock.
#include <hip/hip_runtime.h>
extern "C" __global__ void k_spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (sink && threadIdx.x == 9999) *sink = 1;
}
extern "C" int spin_launch(void* stream, long long ticks, int blocks, int threads, int lds) {
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, ticks, (int*)nullptr);
    return (int)hipGetLastError();
}
