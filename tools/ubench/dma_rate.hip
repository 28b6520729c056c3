// Micro-benchmark: peak rate of LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) per CU.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_rate.hip -o tools/ubench/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lptr_t;

// Each wave issues `inflight` pieces, waits for all, repeats.  span = bytes of source each workgroup cycles through.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const char* src, int iters, int inflight, unsigned span, unsigned wg_stride, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * wg_stride), 0, (int)0x7fffffff, 0x00020000);
    unsigned off = wave * 1024 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < inflight; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(smem + ((wave * 8 + (i & 7)) << 10)), 16, (int)off, 0, 0, 0);
            off += WAVES * 1024;
            if (off >= span) off -= span;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = *(float*)(smem + 4 * (iters & 63));
}

template <int WAVES>
void run(const char* name, const char* src, unsigned span, unsigned wg_stride, int inflight, float* out) {
    const int blocks = 256, iters = 2048 / inflight * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)&k<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES * 8 * 1024);
    k<WAVES><<<blocks, WAVES * 64, WAVES * 8 * 1024>>>(src, 4, inflight, span, wg_stride, out);
    (void)hipEventRecord(e0);
    k<WAVES><<<blocks, WAVES * 64, WAVES * 8 * 1024>>>(src, iters, inflight, span, wg_stride, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * WAVES * iters * inflight * 1024.0;
    printf("%-44s waves/CU=%2d inflight=%d: %.3f ms  %.2f TB/s  %.1f B/clk/CU at 2.4 GHz\n", name, WAVES, inflight, ms,
           bytes / (ms * 1e9), bytes / (ms * 1e6) / 256 / 2.4);
}

int main() {
    char* src; (void)hipMalloc(&src, 1u << 30); (void)hipMemset(src, 1, 1u << 30);
    float* out; (void)hipMalloc(&out, 4096);
    for (int inflight : {2, 4, 8}) {
        run<4>("shared 64 KiB source (L2-hot)", src, 65536, 0, inflight, out);
        run<8>("shared 64 KiB source (L2-hot)", src, 65536, 0, inflight, out);
        run<16>("shared 64 KiB source (L2-hot)", src, 65536, 0, inflight, out);
    }
    run<8>("per-CU 2 MiB stream (L2 misses, 512 MiB total)", src, 2u << 20, 2u << 20, 4, out);
    run<16>("per-CU 2 MiB stream (L2 misses, 512 MiB total)", src, 2u << 20, 2u << 20, 4, out);
    run<8>("per-CU 128 KiB (32 MiB total, L2-resident)", src, 128u << 10, 128u << 10, 4, out);
    run<16>("per-CU 128 KiB (32 MiB total, L2-resident)", src, 128u << 10, 128u << 10, 4, out);
    return 0;
}
