// Experiment (round 5): a 288 x 320 x 64 tile on EIGHT waves, v_mfma_f32_16x16x32_bf16, six phases per K-tile.
//
// Why 288 rows: one workgroup per CU means the number of tiles should be a multiple of 256, and every MDM1024 level is — a frame
// is 9216 / 2304 / 576 pixels = 32 / 8 / 2 x 288, so the guidance batch (2 x 16 frames) gives 1024 x 1, 256 x 2, 64 x 4 tiles of
// 288 x 320 at levels 0 / 1 / 2: 4.0 / 2.0 / 1.0 rounds, where 256-row tiles give 4.5 / 2.25 / 1.125 (= 5 / 3 / 2 rounds of cost).
// 288 = 2 x 9 x 16 and 320 = 4 x 5 x 16: waves as 2 (M) x 4 (N), a wave owns 144 x 80 = 9 x 5 fragments of 16 x 16 = 180
// accumulators; 151 FLOP per staged byte (64 for the 128 x 128 tile: the L2 -> LDS path stops binding).
//
// LDS: two K-tile buffers x two k halves (32 deep) x {A: 18 subtiles, B: 20 subtiles} of 1 KiB (16 rows x 32 k, st_16x32 swizzle) =
// 152 KiB.  The ring is managed per k HALF: half ks of buffer b is free once phase 3 ks + 2 of its tile is over, and is re-staged
// for the tile two ahead — about 1.5 K-tiles (114 KiB) of DMA are in flight at any time, each piece has 13 barrier slots to land.
//
// Per K-tile t (buffer t & 1), phases p = 0..5 = (ks = p / 3, third = p % 3); each phase = LOAD section | barrier | 15 MFMAs | barrier:
//   p   ds_read                               MFMA                         DMA issued in the LOAD section       wait at the end of the MFMA section
//   0   B ks 0 (5), A ks 0 rows 0-2 (3)       rows 0-2 x 5 x ks 0          -
//   1   A ks 0 rows 3-5 (3)                   rows 3-5 ...                 ks 1 of tile t + 1 -> buffer (t+1)&1   vmcnt(2 n): ks 1 of tile t has landed
//   2   A ks 0 rows 6-8 (3)                   rows 6-8 ...                 -
//   3   B ks 1 (5), A ks 1 rows 0-2           rows 0-2 x 5 x ks 1          -
//   4   A ks 1 rows 3-5                       ...                          ks 0 of tile t + 2 -> buffer t & 1     vmcnt(2 n): ks 0 of tile t + 1 has landed
//   5   A ks 1 rows 6-8                       ...                          -
// The waves of M-half 1 run one barrier behind those of M-half 0: in every barrier slot one wave of a SIMD multiplies while the other
// reads fragments and issues DMA.  RAW: a wait sits a whole phase before the first read of what it retires (the lagging group's
// wait still precedes the leading group's read by a barrier).  WAR: a k half is re-staged two phases after its last fragment read.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wgemm_288.hip -o tools/ubench/wgemm_288 && tools/ubench/wgemm_288
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __bf16 h16;
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) h16 h16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lptr_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

constexpr int BM = 288, BN = 320;
constexpr int NA = 18, NB = 20;                  // 16-row subtiles of the A / B operand tile
constexpr int KS_BYTES = (NA + NB) * 1024;       // one k half of a buffer
constexpr int BUF = 2 * KS_BYTES;
constexpr int SMEM = 2 * BUF;                    // 155648

#define RAW_BARRIER()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const h16* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(base), 0, (int)0x80000000u, 0x00020000);
}

// NPH: phases per k half (3: row thirds, 15 MFMAs per phase; 2: rows 0-4 / 5-8, 25 / 20 MFMAs).  SPREAD (NPH = 3): the DMA of a k half is
// issued over two LOAD sections instead of one.  ABL: ablations (wrong results): 1 = no DMA inside the loop, 2 = no fragment reads inside the loop.
template <int NPH, int SPREAD, int ABL>
__global__ __launch_bounds__(512, 2) void wgemm_288(const h16* __restrict__ A, const h16* __restrict__ B, h16* __restrict__ C,
                                                    int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;          // waves wc and wc + 4 share a SIMD: the two M halves

    const int ntn = N / BN, ntm = M / BM;
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
    const int m0 = tm * BM, n0 = tn * BN;

    // DMA lane geometry (see gemm_8p.hip): lane l writes byte l * 16 of a subtile and fetches the element whose swizzled position that is
    const int pos = lane * 16;
    const int byte = pos ^ (((pos >> 9) & 1) << 5);
    const int srow = byte >> 6, schunk = (byte >> 4) & 3;
    const unsigned va = (unsigned)(srow * K + schunk * 8) * 2u;      // lane part of the byte offset (same for A and B: ld = K)
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(A + (size_t)m0 * K), rB = make_rsrc(B + (size_t)n0 * K);
    // pieces per k half: A half wr: 9 subtiles over its 4 waves as 3 2 2 2; B: 20 subtiles over 8 waves as 2 3 3 2 | 2 3 3 2
    const int a_first = wc == 0 ? 0 : 1 + 2 * wc, a_cnt = wc == 0 ? 3 : 2;
    const int b_cnt = (wc == 0 || wc == 3) ? 2 : 3;
    const int b_first = wr * 10 + (wc == 0 ? 0 : (wc == 1 ? 2 : (wc == 2 ? 5 : 8)));
    const bool five = a_cnt + b_cnt == 5;             // pieces per k half of this wave: 5 (wc 0-2) or 4 (wc 3)
    // part: 0 = all pieces, 1 = the A pieces, 2 = the B pieces
    auto stage = [&](int kt, int ks, int buf, int part) {       // this wave's pieces of k half ks of K-tile kt
        char* base = smem + buf * BUF + ks * KS_BYTES;
        const int koff = (kt * 64 + ks * 32) * 2;
        if (part != 2)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < a_cnt) {
                const int st = wr * 9 + a_first + q;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(base + st * 1024), 16, (int)va, st * 16 * K * 2 + koff, 0, 0);
            }
        if (part != 1)
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < b_cnt) {
                const int st = b_first + q;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lptr_t)(base + (NA + st) * 1024), 16, (int)va, st * 16 * K * 2 + koff, 0, 0);
            }
    };

    auto stage_piece = [&](int kt, int ks, int buf, int q) {       // piece q (A pieces first) of this wave's share of a k half
        char* base = smem + buf * BUF + ks * KS_BYTES;
        const int koff = (kt * 64 + ks * 32) * 2;
        if (q < a_cnt) {
            const int st = wr * 9 + a_first + q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(base + st * 1024), 16, (int)va, st * 16 * K * 2 + koff, 0, 0);
        } else if (q < a_cnt + b_cnt) {
            const int st = b_first + q - a_cnt;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lptr_t)(base + (NA + st) * 1024), 16, (int)va, st * 16 * K * 2 + koff, 0, 0);
        }
    };
    // fragment reads: a 16 x 32 fragment is one subtile; lane l holds row l % 16, k chunk l / 16
    const int fbyte0 = (lane & 15) * 64 + (lane >> 4) * 16;
    const int fbyte = fbyte0 ^ (((fbyte0 >> 9) & 1) << 5);
    const char* a_base = smem + (wr * 9) * 1024 + fbyte;
    const char* b_base = smem + (NA + wc * 5) * 1024 + fbyte;

    f32x4 acc[9][5];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int AF = NPH == 3 ? 3 : 5;
    h16x8 af[AF], bf[5];
#pragma unroll
    for (int i = 0; i < AF; ++i) af[i] = *reinterpret_cast<const h16x8*>(a_base + i * 1024);
#pragma unroll
    for (int j = 0; j < 5; ++j) bf[j] = *reinterpret_cast<const h16x8*>(b_base + j * 1024);

    auto read_a = [&](int buf, int ks, auto r0_tag, auto cnt_tag) {
        constexpr int r0 = decltype(r0_tag)::value, cnt = decltype(cnt_tag)::value;
        if (ABL & 2) return;
#pragma unroll
        for (int i = 0; i < cnt; ++i) af[i] = *reinterpret_cast<const h16x8*>(a_base + buf * BUF + ks * KS_BYTES + (r0 + i) * 1024);
    };
    auto read_b = [&](int buf, int ks) {
        if (ABL & 2) return;
#pragma unroll
        for (int j = 0; j < 5; ++j) bf[j] = *reinterpret_cast<const h16x8*>(b_base + buf * BUF + ks * KS_BYTES + j * 1024);
    };
    auto mma = [&](auto r0_tag, auto cnt_tag) {
        constexpr int r0 = decltype(r0_tag)::value, cnt = decltype(cnt_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < cnt; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j)       // operands swapped: a lane ends up with 4 consecutive n of one row m
                acc[r0 + i][j] = MFMA16(bf[j], af[i], acc[r0 + i][j]);
    };
    // SPREAD = 2: the DMA pieces are issued INSIDE the MFMA sections (two per section, each behind five MFMAs), the LOAD sections only read
    auto mma_dma = [&](auto r0_tag, bool on, int kt, int ks, int buf, int q0) {
        constexpr int r0 = decltype(r0_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[r0 + i][j] = MFMA16(bf[j], af[i], acc[r0 + i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (i < 2 && on) stage_piece(kt, ks, buf, q0 + i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
    using I0 = std::integral_constant<int, 0>; using I3 = std::integral_constant<int, 3>; using I6 = std::integral_constant<int, 6>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;

    const int nk = K >> 6;
    stage(0, 0, 0, 0);
    stage(0, 1, 0, 0);
    if (nk > 1) stage(1, 0, 1, 0);
    if (nk <= 1) VMCNT(0); else if (five) VMCNT(10); else VMCNT(8);       // ks 0 of tile 0 has landed
    RAW_BARRIER();
    if (wr == 1) RAW_BARRIER();                      // the stagger: M-half 1 runs one barrier behind M-half 0

    if constexpr (NPH == 3 && SPREAD == 2) {
        for (int t = 0; t < nk; ++t) {
            const int buf = t & 1;
            const bool n1 = (t + 1 < nk) && !(ABL & 1), n2 = (t + 2 < nk) && !(ABL & 1);
            read_b(buf, 0);
            read_a(buf, 0, I0{}, I3{});
            RAW_BARRIER();
            mma_dma(I0{}, n1, t + 1, 1, buf ^ 1, 0);
            RAW_BARRIER();
            read_a(buf, 0, I3{}, I3{});
            RAW_BARRIER();
            mma_dma(I3{}, n1, t + 1, 1, buf ^ 1, 2);
            // ks 1 of tile t has landed; behind it: ks 0 of tile t + 1 (n pieces) and the four pieces of ks 1 of tile t + 1 issued so far
            if (ABL & 1) {} else if (!n1) VMCNT(0); else if (five) VMCNT(9); else VMCNT(8);
            RAW_BARRIER();
            read_a(buf, 0, I6{}, I3{});
            RAW_BARRIER();
            mma_dma(I6{}, n1, t + 1, 1, buf ^ 1, 4);
            RAW_BARRIER();
            read_b(buf, 1);
            read_a(buf, 1, I0{}, I3{});
            RAW_BARRIER();
            mma_dma(I0{}, n2, t + 2, 0, buf, 0);
            RAW_BARRIER();
            read_a(buf, 1, I3{}, I3{});
            RAW_BARRIER();
            mma_dma(I3{}, n2, t + 2, 0, buf, 2);
            if (ABL & 1) {} else if (!n2) VMCNT(0); else if (five) VMCNT(9); else VMCNT(8);
            RAW_BARRIER();
            read_a(buf, 1, I6{}, I3{});
            RAW_BARRIER();
            mma_dma(I6{}, n2, t + 2, 0, buf, 4);
            RAW_BARRIER();
        }
    } else
    if constexpr (NPH == 3) {
        for (int t = 0; t < nk; ++t) {
            const int buf = t & 1;
            const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
            // phase 0
            read_b(buf, 0);
            read_a(buf, 0, I0{}, I3{});
            RAW_BARRIER();
            mma(I0{}, I3{});
            RAW_BARRIER();
            // phase 1
            read_a(buf, 0, I3{}, I3{});
            if (!(ABL & 1) && n1) stage(t + 1, 1, buf ^ 1, SPREAD ? 2 : 0);
            RAW_BARRIER();
            mma(I3{}, I3{});
            // ks 1 of tile t (issued a tile ago) has landed; behind it: ks 0 of tile t + 1 and what this phase just issued
            if (ABL & 1) {}
            else if (!n1) VMCNT(0);
            else if (!SPREAD) { if (five) VMCNT(10); else VMCNT(8); }
            else { if (five) { if (b_cnt == 3) VMCNT(8); else VMCNT(7); } else VMCNT(6); }
            RAW_BARRIER();
            // phase 2
            read_a(buf, 0, I6{}, I3{});
            if (!(ABL & 1) && SPREAD && n1) stage(t + 1, 1, buf ^ 1, 1);
            RAW_BARRIER();
            mma(I6{}, I3{});
            RAW_BARRIER();
            // phase 3
            read_b(buf, 1);
            read_a(buf, 1, I0{}, I3{});
            RAW_BARRIER();
            mma(I0{}, I3{});
            RAW_BARRIER();
            // phase 4
            read_a(buf, 1, I3{}, I3{});
            if (!(ABL & 1) && n2) stage(t + 2, 0, buf, SPREAD ? 2 : 0);
            RAW_BARRIER();
            mma(I3{}, I3{});
            // ks 0 of tile t + 1 has landed; behind it: ks 1 of tile t + 1 and what this phase just issued
            if (ABL & 1) {}
            else if (!n2) VMCNT(0);
            else if (!SPREAD) { if (five) VMCNT(10); else VMCNT(8); }
            else { if (five) { if (b_cnt == 3) VMCNT(8); else VMCNT(7); } else VMCNT(6); }
            RAW_BARRIER();
            // phase 5
            read_a(buf, 1, I6{}, I3{});
            if (!(ABL & 1) && SPREAD && n2) stage(t + 2, 0, buf, 1);
            RAW_BARRIER();
            mma(I6{}, I3{});
            RAW_BARRIER();
        }
    } else {
        // four phases per K-tile: (ks, rows 0-4 | rows 5-8).  W_B (ks 1 of tile t) at the end of phase 0, W_A (ks 0 of tile t + 1) at the end of phase 2.
        for (int t = 0; t < nk; ++t) {
            const int buf = t & 1;
            const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
            // phase 0
            read_b(buf, 0);
            read_a(buf, 0, I0{}, I5{});
            RAW_BARRIER();
            mma(I0{}, I5{});
            if (ABL & 1) {} else if (!n1) VMCNT(0); else if (five) VMCNT(5); else VMCNT(4);      // behind ks 1 of tile t: ks 0 of tile t + 1
            RAW_BARRIER();
            // phase 1
            read_a(buf, 0, I5{}, I4{});
            if (!(ABL & 1) && n1) stage(t + 1, 1, buf ^ 1, 0);
            RAW_BARRIER();
            mma(I5{}, I4{});
            RAW_BARRIER();
            // phase 2
            read_b(buf, 1);
            read_a(buf, 1, I0{}, I5{});
            RAW_BARRIER();
            mma(I0{}, I5{});
            if (ABL & 1) {} else if (!n1) VMCNT(0); else if (five) VMCNT(5); else VMCNT(4);      // behind ks 0 of tile t + 1: ks 1 of tile t + 1
            RAW_BARRIER();
            // phase 3
            read_a(buf, 1, I5{}, I4{});
            if (!(ABL & 1) && n2) stage(t + 2, 0, buf, 0);
            RAW_BARRIER();
            mma(I5{}, I4{});
            RAW_BARRIER();
        }
    }
    if (wr == 0) RAW_BARRIER();                      // evens out the stagger barrier

    // C (bf16): lane l holds, per fragment, row m = l % 16 and the 4 consecutive columns n = 4 (l / 16) .. + 3
    h16* cw = C + (size_t)(m0 + wr * 144 + (lane & 15)) * N + n0 + wc * 80 + (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            h16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (h16)acc[i][j][e];
            *reinterpret_cast<h16x4*>(cw + (size_t)(i * 16) * N + j * 16) = v;
        }
}

__global__ void ref_kernel(const h16* A, const h16* B, float* R, int M, int N, int K, int rows) {
    const int j = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const int m = (int)((long long)j * M / rows) + (j * 37) % 288;
    if (m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)B[(size_t)c * K + k];
    R[(size_t)j * N + c] = s;
}

template <int NPH, int SPREAD, int ABL>
static float timeit(const h16* A, const h16* B, h16* C, int m, int n, int k) {
    (void)hipFuncSetAttribute((const void*)&wgemm_288<NPH, SPREAD, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const int grid = (m / BM) * (n / BN);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 13; ++i) {
        if (i == 3) (void)hipEventRecord(e0);
        wgemm_288<NPH, SPREAD, ABL><<<grid, 512, SMEM>>>(A, B, C, m, n, k);
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10 * 1e3f;
}

template <int NPH, int SPREAD>
static void check(const h16* A, const h16* B, h16* C, float* R) {
    (void)hipFuncSetAttribute((const void*)&wgemm_288<NPH, SPREAD, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const int checks[][3] = {{288, 320, 64}, {288, 320, 128}, {288, 320, 192}, {576, 640, 320}, {2880, 1280, 1280}, {73728, 640, 2880}};
    for (auto& sh : checks) {
        const int m = sh[0], n = sh[1], k = sh[2];
        double worst = 0;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipMemset(C, 0xff, (size_t)m * n * 2);
            wgemm_288<NPH, SPREAD, 0><<<(m / BM) * (n / BN), 512, SMEM>>>(A, B, C, m, n, k);
            ref_kernel<<<dim3((n + 255) / 256, 64), 256>>>(A, B, R, m, n, k, 64);
            std::vector<h16> c((size_t)m * n); std::vector<float> r(64 * n);
            (void)hipMemcpy(c.data(), C, c.size() * 2, hipMemcpyDeviceToHost);
            (void)hipMemcpy(r.data(), R, r.size() * 4, hipMemcpyDeviceToHost);
            double num = 0, den = 0;
            for (int j = 0; j < 64; ++j) {
                const int row = (int)((long long)j * m / 64) + (j * 37) % 288;
                if (row >= m) continue;
                for (int col = 0; col < n; ++col) {
                    const double d = (double)(float)c[(size_t)row * n + col] - (double)(float)(h16)r[(size_t)j * n + col];
                    num += d * d; den += (double)r[(size_t)j * n + col] * r[(size_t)j * n + col];
                }
            }
            const double e = sqrt(num / den);
            worst = e > worst || !(e == e) ? e : worst;
        }
        printf("check<%d,%d> %d x %d x %d: worst rel-L2 vs the bf16-rounded reference %.3e (%s)\n", NPH, SPREAD, m, n, k, worst, hipGetErrorString(hipGetLastError()));
    }
}

int main() {
    const size_t amax = (size_t)294912 * 8640, bmax = (size_t)1280 * 23040, cmax = (size_t)294912 * 1280;
    std::vector<h16> ha((size_t)1 << 24), hb((size_t)1 << 24);
    srand(1);
    for (size_t i = 0; i < ha.size(); ++i) { ha[i] = (h16)((rand() % 2001 - 1000) * 1e-3f); hb[i] = (h16)((rand() % 2001 - 1000) * 1e-3f); }
    h16 *A, *B, *C; float* R;
    (void)hipMalloc(&A, amax * 2); (void)hipMalloc(&B, bmax * 2); (void)hipMalloc(&C, cmax * 2); (void)hipMalloc(&R, 64 * 1280 * 4);
    for (size_t off = 0; off < amax; off += ha.size())
        (void)hipMemcpy(A + off, ha.data(), (amax - off < ha.size() ? amax - off : ha.size()) * 2, hipMemcpyHostToDevice);
    for (size_t off = 0; off < bmax; off += hb.size())
        (void)hipMemcpy(B + off, hb.data(), (bmax - off < hb.size() ? bmax - off : hb.size()) * 2, hipMemcpyHostToDevice);
    check<3, 0>(A, B, C, R);
    check<3, 1>(A, B, C, R);
    check<2, 0>(A, B, C, R);
    check<3, 2>(A, B, C, R);
    const int shapes[][3] = {{294912, 320, 2880}, {294912, 320, 960}, {294912, 320, 1280}, {294912, 320, 320},
                             {73728, 640, 5760}, {73728, 640, 17280}, {73728, 640, 1920}, {73728, 640, 2560}, {73728, 640, 640}, {73728, 1920, 640},
                             {18432, 1280, 11520}, {18432, 1280, 23040}, {18432, 1280, 3840}, {18432, 1280, 5120}, {18432, 1280, 1280}, {18432, 3840, 1280}};
    printf("%-22s %9s %9s %9s | %9s %9s %9s %9s  (us; TFLOP/s of the first three)\n", "shape", "3ph-inmma", "3ph-spread", "2ph", "3s-noDMA", "3s-noRD", "inmma #2", "spread #2");
    for (auto& sh : shapes) {
        const int m = sh[0], n = sh[1], k = sh[2];
        const float a = timeit<3, 2, 0>(A, B, C, m, n, k), b = timeit<3, 1, 0>(A, B, C, m, n, k), c = timeit<2, 0, 0>(A, B, C, m, n, k);
        const float d = timeit<3, 1, 1>(A, B, C, m, n, k), e = timeit<3, 1, 2>(A, B, C, m, n, k);
        const float f = timeit<3, 2, 0>(A, B, C, m, n, k), g = timeit<3, 1, 0>(A, B, C, m, n, k);
        const double fl = 2.0 * m * n * (double)k * 1e-6;
        printf("%6d x %4d x %5d %9.1f %9.1f %9.1f | %9.1f %9.1f %9.1f %9.1f   %6.0f %6.0f %6.0f (%s)\n", m, n, k, a, b, c, d, e, f, g, fl / a, fl / b, fl / c,
               hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
