// Experiment: 256x256x64 tile, FOUR waves (one per SIMD), each owning 128x128 of the tile (16 accumulators of 32x32 =
// 256 registers), software-pipelined inside the wave: fragments of k-step s+1 are read while k-step s multiplies, the
// DMA of K-tile t+1 / t+2 is spread over k-steps 3 and 0, one barrier per K-tile between k-steps 2 and 3.
// Result on MI355X: 1005-1011 TFLOP/s at 4096^3, 1106-1125 at 8192^3 with LDS-DMA staging (VAR 0), 887 / 879 with
// register staging + ds_write_b128 (VAR 1) — no better than the kernels in mudg_amd/csrc (the vendor library's assembly
// kernels reach 1439 / 1530 on the same shapes); kept as the record of the experiment.
// Plain bf16 GEMM Y[M][N] = X[M][K] W[N][K]^T, M, N % 256 == 0, K % 64 == 0.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_w4.hip -o tools/ubench/gemm_w4 && tools/ubench/gemm_w4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __bf16 h16;
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) h16 h16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lptr_t;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

constexpr int KB_BYTES = 65536;          // one K-tile: X 32 KiB + W 32 KiB
constexpr int W_OFF = 32768;

template <int VAR>
__global__ __launch_bounds__(256, 1) void gemm_w4(const h16* __restrict__ X, const h16* __restrict__ W, h16* __restrict__ Y,
                                                  int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, hi = lane >> 5;

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

    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (size_t)m0 * K), 0, (int)0x80000000u, 0x00020000);
    __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * K), 0, (int)0x80000000u, 0x00020000);
    const int rsub = lane >> 3, slot = lane & 7;
    unsigned vo[8];                       // same row pattern for both operands (ldx == ldw == K here)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 64 * wave + 8 * i + rsub;
        vo[i] = (unsigned)row * (unsigned)K * 2u + (unsigned)(slot ^ ((row >> 1) & 7)) * 16u;
    }
    auto dma = [&](int kt, int kb, int i0, int i1) {     // pieces [i0, i1) of both operands of K-tile kt into buffer kb
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < i0 || i >= i1) continue;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lptr_t)(smem + kb * KB_BYTES + (64 * wave + 8 * i) * 128), 16, (int)vo[i], kt * 128, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(smem + kb * KB_BYTES + W_OFF + (64 * wave + 8 * i) * 128), 16, (int)vo[i], kt * 128, 0, 0);
        }
    };

    const int swz = (l31 >> 1) & 7;
    int ax[4], aw[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int so = ((ks * 2 + hi) ^ swz) << 4;
        ax[ks] = (wm * 128 + l31) * 128 + so;
        aw[ks] = W_OFF + (wn * 128 + l31) * 128 + so;
    }

    f32x16 acc[4][4];                     // [nb][mb]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    h16x8 fx[2][4], fw[2][4];             // [parity][block]
    auto read_frags = [&](int par, int kb, int ks) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            fx[par][b] = *reinterpret_cast<const h16x8*>(smem + kb * KB_BYTES + ax[ks] + b * 4096);
            fw[par][b] = *reinterpret_cast<const h16x8*>(smem + kb * KB_BYTES + aw[ks] + b * 4096);
        }
    };
    auto mma = [&](int par) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                acc[nb][mb] = MFMA(fw[par][nb], fx[par][mb], acc[nb][mb]);
    };

    // VAR 1: register staging instead of LDS-DMA: 16 global loads per K-tile per lane, written with ds_write_b128
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    u32x4 sx[8], sw[8];
    unsigned wo[8];                      // LDS byte offset of this lane's 16-byte slot inside an operand tile
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 64 * wave + 8 * i + rsub;
        wo[i] = (unsigned)row * 128u + (unsigned)(slot ^ ((row >> 1) & 7)) * 16u;
    }
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 64 * wave + 8 * i + rsub;
            sx[i] = *reinterpret_cast<const u32x4*>(X + (size_t)(m0 + row) * K + kt * 64 + slot * 8);
            sw[i] = *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + row) * K + kt * 64 + slot * 8);
        }
    };
    auto lstore = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<u32x4*>(smem + kb * KB_BYTES + wo[i]) = sx[i];
            *reinterpret_cast<u32x4*>(smem + kb * KB_BYTES + W_OFF + wo[i]) = sw[i];
        }
    };

    const int nk = K >> 6;
    if constexpr (VAR == 1) {
        gload(0);
        lstore(0);
        __syncthreads();
        read_frags(0, 0, 0);
        auto ktile1 = [&](int kt, auto KBc) {
            constexpr int KB = decltype(KBc)::value;
            read_frags(1, KB, 1);
            if (kt + 1 < nk) gload(kt + 1);
            mma(0);
            read_frags(0, KB, 2);
            mma(1);
            read_frags(1, KB, 3);
            mma(0);
            if (kt + 1 < nk) lstore(KB ^ 1);          // buffer kb^1 was released by the previous K-tile's barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) read_frags(0, KB ^ 1, 0);
            mma(1);
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            ktile1(kt, std::integral_constant<int, 0>{});
            ktile1(kt + 1, std::integral_constant<int, 1>{});
        }
        if (kt < nk) ktile1(kt, std::integral_constant<int, 0>{});
    } else {
    dma(0, 0, 0, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0);
    if (nk > 1) dma(1, 1, 0, 4);

    auto ktile = [&](int kt, auto KBc) {
        constexpr int KB = decltype(KBc)::value;
        // k-step 0: multiply frags(0) | read frags(1) | second half of DMA(kt+1)
        read_frags(1, KB, 1);
        if (kt + 1 < nk) dma(kt + 1, KB ^ 1, 4, 8);
        mma(0);
        // k-step 1
        read_frags(0, KB, 2);
        mma(1);
        // k-step 2
        read_frags(1, KB, 3);
        mma(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // k-step 3: multiply frags(3) | read frags(kt+1, 0) | first half of DMA(kt+2) into the buffer just released
        if (kt + 1 < nk) read_frags(0, KB ^ 1, 0);
        if (kt + 2 < nk) dma(kt + 2, KB, 0, 4);
        mma(1);
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        ktile(kt, std::integral_constant<int, 0>{});
        ktile(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < nk) ktile(kt, std::integral_constant<int, 0>{});
    }

    // epilogue: lane holds rows (mb*32 + l31), 4 consecutive channels per (nb, g): 8-byte stores
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const size_t m = (size_t)m0 + wm * 128 + mb * 32 + l31;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 128 + nb * 32 + 8 * g + 4 * hi;
                h16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (h16)acc[nb][mb][4 * g + j];
                *reinterpret_cast<h16x4*>(Y + m * N + n) = o;
            }
    }
}

__global__ void ref_kernel(const h16* X, const h16* W, float* R, int M, int N, int K, int rows) {
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y * (M / rows);
    if (n >= N) return;
    float a = 0.f;
    for (int k = 0; k < K; ++k) a += (float)X[(size_t)m * K + k] * (float)W[(size_t)n * K + k];
    R[(size_t)blockIdx.y * N + n] = a;
}

template <int VAR>
void run(int n, const h16* X, const h16* W, h16* Y) {
    const int tiles = (n / 256) * (n / 256);
    (void)hipFuncSetAttribute((const void*)&gemm_w4<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KB_BYTES);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) gemm_w4<VAR><<<tiles, 256, 2 * KB_BYTES>>>(X, W, Y, n, n, n);
    (void)hipEventRecord(e0);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) gemm_w4<VAR><<<tiles, 256, 2 * KB_BYTES>>>(X, W, Y, n, n, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    printf("gemm_w4<%d> %d^3: %.1f us  %.1f TFLOP/s  (%s)\n", VAR, n, ms * 1e3 / iters, 2.0 * n * n * (double)n / (ms / iters * 1e-3) / 1e12,
           hipGetErrorString(err));
}

int main() {
    const int nmax = 8192;
    std::vector<h16> hx((size_t)nmax * nmax), hw((size_t)nmax * nmax);
    srand(1);
    for (size_t i = 0; i < hx.size(); ++i) { hx[i] = (h16)((rand() % 2001 - 1000) * 1e-3f); hw[i] = (h16)((rand() % 2001 - 1000) * 1e-3f); }
    h16 *X, *W, *Y; float* R;
    (void)hipMalloc(&X, hx.size() * 2); (void)hipMalloc(&W, hw.size() * 2); (void)hipMalloc(&Y, hx.size() * 2); (void)hipMalloc(&R, 64 * nmax * 4);
    (void)hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    // correctness at 1024^3 (uses the first 1024x1024 elements as dense matrices)
    {
        const int n = 1024;
        (void)hipFuncSetAttribute((const void*)&gemm_w4<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KB_BYTES);
        gemm_w4<0><<<(n / 256) * (n / 256), 256, 2 * KB_BYTES>>>(X, W, Y, n, n, n);
        ref_kernel<<<dim3(n / 256, 64), 256>>>(X, W, R, n, n, n, 64);
        std::vector<h16> y((size_t)n * n); std::vector<float> r(64 * n);
        (void)hipMemcpy(y.data(), Y, y.size() * 2, hipMemcpyDeviceToHost);
        (void)hipMemcpy(r.data(), R, r.size() * 4, hipMemcpyDeviceToHost);
        double num = 0, den = 0;
        for (int j = 0; j < 64; ++j)
            for (int c = 0; c < n; ++c) {
                const double d = (double)(float)y[(size_t)(j * (n / 64)) * n + c] - r[(size_t)j * n + c];
                num += d * d; den += (double)r[(size_t)j * n + c] * r[(size_t)j * n + c];
            }
        printf("check 1024^3: rel-L2 %.3e (%s)\n", sqrt(num / den), hipGetErrorString(hipGetLastError()));
    }
    run<0>(4096, X, W, Y);
    run<0>(8192, X, W, Y);
    run<1>(4096, X, W, Y);
    run<1>(8192, X, W, Y);
    return 0;
}
