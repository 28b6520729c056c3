// Micro-benchmark: issue rate of v_exp_f32 / v_fma_f32 / v_max3_f32 / v_pk_fma_f32 per CU (wave64 instructions per cycle).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
            else if (OP == 1) a[i] = fmaf(a[i], 1.0001f, 0.5f);
            else if (OP == 2) a[i] = fmaxf(fmaxf(a[i], a[(i + 1) & 7]), seed);
            else if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i + 2) & 6]));
            else if (OP == 4) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            else if (OP == 5) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            else if (OP == 6) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            else if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            else if (OP == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, float* out) {
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 16, 0.5f);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 8;          // wave-instructions
    printf("%-14s %.3f ms  %.2f wave-instr/ns chip  = %.3f wave-instr/clk/CU at 2.4 GHz (%.1f lanes/clk/SIMD)\n", name, ms,
           winstr / (ms * 1e6), winstr / (ms * 1e6) / 256 / 2.4, winstr / (ms * 1e6) / 256 / 2.4 / 4 * 64);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_exp_f32", out); run<1>("v_fma_f32", out); run<2>("v_max3_f32", out); run<3>("v_pk_fma_f32", out);
    run<4>("v_dot2_f32_bf16", out); run<5>("v_dot2c_f32_bf16", out); run<6>("v_dot2_f32_f16", out); run<7>("v_rcp_f32", out);
    run<8>("v_cvt_pk_bf16_f32", out);
    return 0;
}
