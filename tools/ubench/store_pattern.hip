// Experiment: what a row-per-lane ("straight from the MFMA accumulators") store / load pattern costs against fully coalesced
// rows.  A wave owns a 64-row x 64-channel (128 B, 16-bit) piece of a [M][N] matrix, as the 128 x 128 GEMM tile's waves do.
//   mode 0: coalesced — lane l writes 16 B at row (i * 8 + l / 8), byte (l % 8) * 16: 8 rows x 128 B per instruction
//   mode 1: direct    — lane (l31, hi) writes 16 B at row l31 (+ 32 mi), byte 32 k + 16 hi: 32 rows x 32 B per instruction
//   mode 2: direct, 8-byte pieces (16 instructions per 32 rows)
// and the same three as loads (mode + 4).  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(char* __restrict__ Y, int M, int N, unsigned* sink) {
    // tile = 128 rows x 128 channels (256 B); 4 waves 2 x 2
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1;
    const int ntn = N / 128;
    const int ntiles = (M / 128) * ntn;
    unsigned acc = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tm = tile / ntn, tn = tile - tm * ntn;
        char* base = Y + ((size_t)(tm * 128 + wm * 64) * N + tn * 128 + wn * 64) * 2;
        const size_t ld = (size_t)N * 2;
        if (MODE == 0 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                char* p = base + (size_t)(i * 8 + (lane >> 3)) * ld + (lane & 7) * 16;
                if (MODE == 0) { u32x4 v = {(unsigned)tile, (unsigned)i, (unsigned)lane, 1u}; *reinterpret_cast<u32x4*>(p) = v; }
                else { u32x4 v = *reinterpret_cast<const u32x4*>(p); acc += v[0] + v[3]; }
            }
        } else if (MODE == 1 || MODE == 5) {
            const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    char* p = base + (size_t)(mi * 32 + l31) * ld + kk * 32 + hi * 16;
                    if (MODE == 1) { u32x4 v = {(unsigned)tile, (unsigned)kk, (unsigned)lane, 1u}; *reinterpret_cast<u32x4*>(p) = v; }
                    else { u32x4 v = *reinterpret_cast<const u32x4*>(p); acc += v[0] + v[3]; }
                }
        } else {
            const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    char* p = base + (size_t)(mi * 32 + l31) * ld + kk * 16 + hi * 8;
                    if (MODE == 2) { u32x2 v = {(unsigned)tile, (unsigned)lane}; *reinterpret_cast<u32x2*>(p) = v; }
                    else { u32x2 v = *reinterpret_cast<const u32x2*>(p); acc += v[0] + v[1]; }
                }
        }
    }
    if (MODE >= 4 && acc == 0x12345678u) *sink = acc;
}

template <int MODE>
void run(char* Y, int M, int N, unsigned* sink, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, Y, M, N, sink);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, Y, M, N, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)M * N * 2;
    printf("M=%d N=%d grid=%d mode %d: %.1f us  %.2f TB/s\n", M, N, grid, MODE, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
    unsigned* sink; hipMalloc(&sink, 4);
    const int M = 294912;
    for (int N : {320 * 0 + 384, 1280, 2560}) {
        char* Y; hipMalloc(&Y, (size_t)M * N * 2); hipMemset(Y, 1, (size_t)M * N * 2);
        for (int grid : {1024, 4096}) {
            run<0>(Y, M, N, sink, grid); run<1>(Y, M, N, sink, grid); run<2>(Y, M, N, sink, grid);
            run<4>(Y, M, N, sink, grid); run<5>(Y, M, N, sink, grid); run<6>(Y, M, N, sink, grid);
        }
        hipFree(Y);
    }
    return 0;
}
