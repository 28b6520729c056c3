// Experiment: the 256x256x64-tile, EIGHT-wave "phase" schedule of cdna_hip_programming.md (8 waves as 2 (M) x 4 (N), a wave
// owns 128 x 64 of the tile as 8 x 4 v_mfma_f32_16x16x32_bf16 fragments = 128 accumulators; 128 KiB of LDS = two K-tile
// buffers x {A rows 0-127, A rows 128-255, B cols 0-127, B cols 128-255}; operand half-tiles staged by
// global_load_lds_dwordx4 into 1-KiB 16x32 subtiles with the st_16x32 XOR swizzle; waves of M-half 1 run one barrier behind
// those of M-half 0, so that on every SIMD one wave multiplies while the other reads fragments and issues DMA).
//
// Per K-tile t (LDS buffer t & 1), four phases, each = LOAD section | barrier | lgkmcnt(0), 16 MFMAs (one 64 x 32 quadrant
// x K = 64) | barrier:
//   phase   ds_read (fragments)                   MFMA quadrant (m half, n half)   DMA issued
//   0       B n-half 0 (4), A m-half 0 (8)        (0, 0)                           own A half of tile t + 1 -> buffer (t + 1) & 1
//   1       B n-half 1 (4)                        (0, 1)                           -
//   2       A m-half 1 (8)                        (1, 1)                           - ; vmcnt(4): B(t + 1) has landed
//   3       -                                     (1, 0)                           B halves of tile t + 2 -> buffer t & 1 ; vmcnt(4): A(t + 1)
// WAR: the A halves of a buffer are last read in phase 2 of its tile and re-staged in phase 0 of the next tile; the B halves
// are last read in phase 1 and re-staged in phase 3 (two phases later: the lagging wave group's reads have retired).  RAW: an
// A half is staged, retired and read by ONE wave group (lock-stepped: retire, barrier, read); the B halves, read by both
// groups, are retired a phase earlier, so that the lagging group's retire still precedes the leading group's first read by a
// barrier.  Every DMA has three phases to land.
//
// Result on MI355X (random operands in [-1, 1)): 1184 TFLOP/s at 4096^3 (one tile per CU: prologue / epilogue exposed; 1029 with
// s_setprio 1 around the MFMA sections), 1277 at 8192^3, 1328 with s_setprio — against 1031 / 1076 of the 128x128 kernels in
// mudg_amd/csrc and 1476 / 1545 of the vendor library's assembly kernels (tools/exp_blas.py); results equal a reference
// GEMM to 4e-7 over repeated runs at 256^3 ... 2048^3.  On the path's own plain-GEMM shapes (bf16 result, 8-byte stores
// straight from the accumulators) the margin over the 128x128 kernels shrinks to +2...9 %: 18432 x 10240 x 1280 988-1037
// (947), x 1280 x 5120 1046 (994), x 3840 x 1280 958 (936), 73728 x 5120 x 640 850 (818), 294912 x 2560 x 320 607 (630) — at
// K = 320 ... 1280 a tile is 5-20 K-tiles and its prologue / epilogue weigh as much as in the small-tile kernels; the
// schedule pays where K is long (the 3x3 convs, K = 2880 ... 23040).  Not wired into mudg_gemm yet: it needs the epilogue family (GEGLU,
// residual / output storage kinds, GroupNorm partials) and the implicit-GEMM loaders of gemm.hip underneath it.
// Plain bf16 GEMM C[M][N] (fp32) = A[M][K] B[N][K]^T, M, N % 256 == 0, K % 64 == 0.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_8p.hip -o tools/ubench/gemm_8p && tools/ubench/gemm_8p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __bf16 h16;
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

constexpr int HALF = 16384;              // one operand half-tile: 128 rows x 64 k x 2 B = 16 subtiles of 1 KiB
constexpr int BUF = 4 * HALF;            // A0 A1 B0 B1
constexpr int SMEM = 2 * BUF;            // 128 KiB

#define RAW_BARRIER()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

template <int PRIO, bool OUT16 = false>
__global__ __launch_bounds__(512, 2) void gemm_8p(const h16* __restrict__ A, const h16* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;          // waves wc and wc + 4 share a SIMD: the two M halves

    const int ntn = N >> 8, ntm = M >> 8;
    int tile;
    {
        const int total = gridDim.x, q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    int tm, tn;
    {
        const int per = 8 * ntn, g = tile / per, first = g * 8;
        const int gsz = (ntm - first) < 8 ? (ntm - first) : 8;
        const int r = tile - g * per;
        tn = r / gsz;
        tm = first + (r - tn * gsz);
    }
    const int m0 = tm << 8, n0 = tn << 8;

    // DMA: wave w stages the 16-row block w of every half-tile (two 1-KiB subtiles: k 0-31, k 32-63).  An instruction writes
    // its 64 x 16 B lane-linear; lane l therefore fetches the element whose SWIZZLED position is byte l * 16 of the subtile.
    const int pos = lane * 16;
    const int byte = pos ^ (((pos >> 9) & 1) << 5);
    const int srow = byte >> 6, schunk = (byte >> 4) & 3;
    // A half h is read only by the waves with wr == h: they stage it themselves (wave wc: 16-row blocks 2 wc and 2 wc + 1), so
    // that its DMA -> read ordering stays inside one lock-stepped wave group.  The B halves are read by both groups: every
    // wave stages 16-row block `wave` of both, and they are retired a phase earlier (see the loop).
    const h16* gA[2];
    const h16* gB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        gA[h] = A + (size_t)(m0 + wr * 128 + (wc * 2 + h) * 16 + srow) * K + schunk * 8;
        gB[h] = B + (size_t)(n0 + h * 128 + wave * 16 + srow) * K + schunk * 8;
    }
    auto stage_a = [&](int kt, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                __builtin_amdgcn_global_load_lds((gptr_t)(gA[h] + kt * 64 + kb * 32),
                                                 (lptr_t)(smem + buf * BUF + wr * HALF + ((wc * 2 + h) * 2 + kb) * 1024), 16, 0, 0);
    };
    auto stage_b = [&](int kt, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                __builtin_amdgcn_global_load_lds((gptr_t)(gB[h] + kt * 64 + kb * 32),
                                                 (lptr_t)(smem + buf * BUF + (2 + h) * HALF + (wave * 2 + kb) * 1024), 16, 0, 0);
    };

    // fragment reads: a 16 x 32 fragment is one subtile; lane l holds row l % 16, k chunk l / 16
    const int fbyte0 = (lane & 15) * 64 + (lane >> 4) * 16;
    const int fbyte = fbyte0 ^ (((fbyte0 >> 9) & 1) << 5);
    const char* a_base = smem + wr * HALF + fbyte;                              // + buf * BUF + (mf * 2 + ks) * 1024
    const char* b_base = smem + (2 + (wc >> 1)) * HALF + ((wc & 1) * 4) * 2048 + fbyte;      // + buf * BUF + (j * 2 + ks) * 1024

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 af[4][2], bf[2][2][2];         // [m frag of the current m half][ks], [n half][j][ks]

    auto read_a = [&](int buf, int mh) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                af[mi][ks] = *reinterpret_cast<const h16x8*>(a_base + buf * BUF + ((mh * 4 + mi) * 2 + ks) * 1024);
    };
    auto read_b = [&](int buf, int nh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bf[nh][j][ks] = *reinterpret_cast<const h16x8*>(b_base + buf * BUF + ((nh * 2 + j) * 2 + ks) * 1024);
    };
    auto quadrant = [&](int mh, int nh) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j)          // operands swapped: a lane ends up with 4 consecutive n of one row m
                    acc[mh * 4 + mi][nh * 2 + j] = MFMA16(bf[nh][j][ks], af[mi][ks], acc[mh * 4 + mi][nh * 2 + j]);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };

    const int nk = K >> 6;
    stage_a(0, 0);
    stage_b(0, 0);
    if (nk > 1) stage_b(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER();
    if (wr == 1) RAW_BARRIER();                      // the stagger: M-half 1 runs one barrier behind M-half 0

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        // phase 0
        read_b(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf, 0);
        if (t + 1 < nk) stage_a(t + 1, buf ^ 1);
        RAW_BARRIER();
        quadrant(0, 0);
        RAW_BARRIER();
        // phase 1
        read_b(buf, 1);
        RAW_BARRIER();
        quadrant(0, 1);
        RAW_BARRIER();
        // phase 2
        read_a(buf, 1);
        RAW_BARRIER();
        quadrant(1, 1);
        // B(t + 1) (staged in phase 3 of tile t - 1) has landed; only the A(t + 1) requests of phase 0 may still be in flight
        if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RAW_BARRIER();
        // phase 3
        if (t + 2 < nk) stage_b(t + 2, buf);
        RAW_BARRIER();
        quadrant(1, 0);
        // A(t + 1) has landed; only the B(t + 2) requests just issued may still be in flight
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RAW_BARRIER();
    }
    if (wr == 0) RAW_BARRIER();                      // evens out the stagger barrier

    // C: lane l holds, per fragment, row m = l % 16 and the 4 consecutive columns n = 4 (l / 16) .. + 3
    if constexpr (OUT16) {           // bf16 result (C reinterpreted): 8 bytes per lane, 32-byte runs per row and fragment
        h16* cw = reinterpret_cast<h16*>(C) + (size_t)(m0 + wr * 128 + (lane & 15)) * N + n0 + wc * 64 + (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef __attribute__((ext_vector_type(4))) h16 h16x4;
                h16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (h16)acc[i][j][e];
                *reinterpret_cast<h16x4*>(cw + (size_t)(i * 16) * N + j * 16) = v;
            }
    } else {
        float* cw = C + (size_t)(m0 + wr * 128 + (lane & 15)) * N + n0 + wc * 64 + (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(cw + (size_t)(i * 16) * N + j * 16) = acc[i][j];
    }
}

__global__ void ref_kernel(const h16* A, const h16* B, float* R, int M, int N, int K, int rows) {
    const int j = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const int m = j * (M / rows);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)B[(size_t)c * K + k];
    R[(size_t)j * N + c] = s;
}

template <int PRIO>
static void run(int m, int n, int k, const h16* A, const h16* B, float* C) {
    (void)hipFuncSetAttribute((const void*)&gemm_8p<PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const int grid = (m / 256) * (n / 256);
    for (int i = 0; i < 3; ++i) gemm_8p<PRIO><<<grid, 512, SMEM>>>(A, B, C, m, n, k);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) gemm_8p<PRIO><<<grid, 512, SMEM>>>(A, B, C, m, n, k);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("gemm_8p<prio %d> %d x %d x %d: %.1f us  %.1f TFLOP/s (%s)\n", PRIO, m, n, k, ms / it * 1e3,
           2.0 * m * n * (double)k / (ms / it * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int nmax = 8192;
    std::vector<h16> ha((size_t)nmax * nmax), hb((size_t)nmax * nmax);
    srand(1);
    for (size_t i = 0; i < ha.size(); ++i) { ha[i] = (h16)((rand() % 2001 - 1000) * 1e-3f); hb[i] = (h16)((rand() % 2001 - 1000) * 1e-3f); }
    h16 *A, *B; float *C, *R;
    (void)hipMalloc(&A, ha.size() * 2); (void)hipMalloc(&B, hb.size() * 2); (void)hipMalloc(&C, (size_t)nmax * nmax * 4); (void)hipMalloc(&R, 64 * nmax * 4);
    (void)hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    for (int n : {256, 1024, 2048}) {        // correctness (the first n x n elements as dense matrices), repeated: a race shows as a flaky error
        double worst = 0;
        for (int rep = 0; rep < (n == 256 ? 1 : 5); ++rep) {
            (void)hipFuncSetAttribute((const void*)&gemm_8p<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
            (void)hipMemset(C, 0xff, (size_t)n * n * 4);
            gemm_8p<1><<<(n / 256) * (n / 256), 512, SMEM>>>(A, B, C, n, n, n);
            ref_kernel<<<dim3(n / 256, 64), 256>>>(A, B, R, n, n, n, 64);
            std::vector<float> c((size_t)n * n), r(64 * n);
            (void)hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(r.data(), R, r.size() * 4, hipMemcpyDeviceToHost);
            double num = 0, den = 0;
            for (int j = 0; j < 64; ++j)
                for (int col = 0; col < n; ++col) {
                    const double d = (double)c[(size_t)(j * (n / 64)) * n + col] - r[(size_t)j * n + col];
                    num += d * d; den += (double)r[(size_t)j * n + col] * r[(size_t)j * n + col];
                }
            const double e = sqrt(num / den);
            worst = e > worst || !(e == e) ? e : worst;
        }
        printf("check %d^3: worst rel-L2 %.3e (%s)\n", n, worst, hipGetErrorString(hipGetLastError()));
    }
    run<0>(4096, 4096, 4096, A, B, C);
    run<1>(4096, 4096, 4096, A, B, C);
    run<0>(8192, 8192, 8192, A, B, C);
    run<1>(8192, 8192, 8192, A, B, C);
    // the path's plain-GEMM shapes whose M and N are multiples of 256, bf16 result as in the product
    {
        const size_t amax = (size_t)294912 * 320 > (size_t)18432 * 5120 ? (size_t)294912 * 320 : (size_t)18432 * 5120;
        h16 *A2, *B2; float* C2;
        (void)hipMalloc(&A2, amax * 2); (void)hipMalloc(&B2, (size_t)10240 * 1280 * 2); (void)hipMalloc(&C2, (size_t)294912 * 2560 * 2);
        for (size_t off = 0; off < amax; off += ha.size())
            (void)hipMemcpy(A2 + off, ha.data(), (amax - off < ha.size() ? amax - off : ha.size()) * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(B2, hb.data(), (size_t)10240 * 1280 * 2, hipMemcpyHostToDevice);
        const int shapes[][3] = {{18432, 10240, 1280}, {18432, 1280, 5120}, {18432, 3840, 1280}, {18432, 1280, 1280}, {73728, 5120, 640},
                                 {73728, 1280, 640}, {294912, 2560, 320}};
        for (auto& sh : shapes) {
            (void)hipFuncSetAttribute((const void*)&gemm_8p<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
            (void)hipFuncSetAttribute((const void*)&gemm_8p<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
            const int grid = (sh[0] / 256) * (sh[1] / 256);
            for (int prio = 0; prio < 2; ++prio) {
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                for (int i = 0; i < 23; ++i) {
                    if (i == 3) (void)hipEventRecord(e0);
                    if (prio) gemm_8p<1, true><<<grid, 512, SMEM>>>(A2, B2, C2, sh[0], sh[1], sh[2]);
                    else gemm_8p<0, true><<<grid, 512, SMEM>>>(A2, B2, C2, sh[0], sh[1], sh[2]);
                }
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                printf("gemm_8p<prio %d, bf16 out> %d x %d x %d: %.1f us  %.1f TFLOP/s (%s)\n", prio, sh[0], sh[1], sh[2], ms / 20 * 1e3,
                       2.0 * sh[0] * sh[1] * (double)sh[2] / (ms / 20 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
            }
        }
    }
    return 0;
}
