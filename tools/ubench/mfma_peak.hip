// Achievable MFMA rate of this chip under load: v_mfma_f32_32x32x16_bf16 with register-resident operands, 4 independent
// accumulators per wave, 1-8 waves per SIMD, zero-filled vs random operands (the chip clocks to its power budget).
// Result on MI355X: 32x32x16 1.57-1.82 PFLOP/s on random operands (2.0-2.43 on zeros); 16x16x32 0.84-1.27 (0.89-1.43).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak && tools/ubench/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

typedef __attribute__((ext_vector_type(4))) float f4v;
__global__ __launch_bounds__(256) void k16(const bf8* __restrict__ src, float* __restrict__ out, int iters) {
    const bf8 a0 = src[threadIdx.x], a1 = src[256 + threadIdx.x], b0 = src[512 + threadIdx.x], b1 = src[768 + threadIdx.x];
    f4v c[8];
    for (int j = 0; j < 8; ++j) c[j] = (f4v){0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((j & 1) ? a1 : a0, (j & 2) ? b1 : b0, c[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k(const bf8* __restrict__ src, float* __restrict__ out, int iters) {
    const bf8 a0 = src[threadIdx.x], a1 = src[256 + threadIdx.x], b0 = src[512 + threadIdx.x], b1 = src[768 + threadIdx.x];
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int n = 1024 * 8;
    __bf16* h = (__bf16*)malloc(n * 2);
    bf8* d; float* out;
    (void)hipMalloc(&d, n * 2); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int fill = 0; fill < 2; ++fill) {
        srand(7);
        for (int i = 0; i < n; ++i) h[i] = (__bf16)(fill ? (rand() % 2001 - 1000) * 1e-3f : 0.f);
        (void)hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
        for (int wps : {1, 2, 4, 8}) {                    // waves per SIMD = blocks of 4 waves per CU
            const int blocks = 256 * wps, iters = 1 << 16;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            k<<<blocks, 256>>>(d, out, 256);
            (void)hipEventRecord(e0);
            k<<<blocks, 256>>>(d, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 4 * iters * 4 * 32768.0;
            printf("32x32x16 %s operands, %d wave(s)/SIMD: %.1f TFLOP/s  (%.0f %% of 2.5 PFLOP/s; effective clock %.2f GHz)\n",
                   fill ? "random" : "zero  ", wps, fl / (ms * 1e-3) / 1e12, 100.0 * fl / (ms * 1e-3) / 2.5e15,
                   fl / (ms * 1e-3) / (256.0 * 4 * 1024.0) / 1e9);
            k16<<<blocks, 256>>>(d, out, 256);
            (void)hipEventRecord(e0);
            k16<<<blocks, 256>>>(d, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double fl16 = (double)blocks * 4 * iters * 8 * 16384.0;
            printf("16x16x32 %s operands, %d wave(s)/SIMD: %.1f TFLOP/s\n", fill ? "random" : "zero  ", wps, fl16 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
