// lds_poison.hip — overwrite the LDS of every CU with a NaN pattern (tools/stress_tile.py).  LDS is not cleared between workgroups, so a
// kernel that reads a fragment before its LDS-DMA has landed normally sees the PREVIOUS launch's bytes — in a stress loop that repeats
// one problem those are the correct bytes and the race is invisible.  After this kernel they are NaNs.
// 160 KiB per workgroup = one workgroup per CU; many more workgroups than CUs, each holding its CU for a few microseconds, so that the
// dispatcher has to place one on every CU.
#include <hip/hip_runtime.h>

extern "C" __global__ __launch_bounds__(512) void lds_poison_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds[];
    constexpr int WORDS = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < WORDS; i += 512) lds[i] = pattern;
    __syncthreads();
    unsigned acc = 0;
    for (int r = 0; r < 4; ++r)                                   // a few microseconds of residence
        for (int i = threadIdx.x; i < WORDS; i += 512) acc += lds[i] ^ (unsigned)r;
    if (acc == 0x12345u && sink) sink[0] = acc;                   // never true for the patterns used; keeps the reads alive
}

extern "C" int lds_poison(void* stream, int workgroups, unsigned pattern) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
        attr = true;
    }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(workgroups), dim3(512), 160 * 1024, reinterpret_cast<hipStream_t>(stream), pattern, (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
