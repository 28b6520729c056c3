// wgemm.hip — the 288 x 320 tile of the contraction family (round 5): eight waves, v_mfma_f32_16x16x32, one workgroup per CU,
// LDS-DMA ring managed per k half, six barrier-delimited phases per K-tile with the two wave groups of a SIMD in anti-phase.
// Plain GEMM, same-size 3x3 conv and temporal 3-tap conv (descriptor loader, 16-bit builds).  MudgGemmDesc semantics: mudg_hip.h.
//
// Why this tile.  (1) One workgroup per CU wants the tile count to be a multiple of the 256 CUs, and 288 rows make it one at
// every level of the benchmarked resolution: a frame is 9216 / 2304 / 576 pixels = 32 / 8 / 2 x 288, so a guidance batch of
// 2 x 16 frames is 1024 x 1 / 256 x 2 / 64 x 4 tiles of 288 x 320 at levels 0 / 1 / 2 = 4 / 2 / 1 full rounds (256-row tiles:
// 4.5 / 2.25 / 1.125 rounds, i.e. 5 / 3 / 2 rounds of time).  (2) 151 FLOP per staged byte against the 64 of the 128 x 128 tile, whose
// main loop is bound by the L2 -> LDS path: here the loop runs at 1300-1400 TFLOP/s on the long-K shapes where the
// 128 x 128 kernels reach 1000.  (3) 288 = 2 x 9 x 16, 320 = 4 x 5 x 16: waves as 2 (M) x 4 (N), a wave owns 144 x 80 = 9 x 5
// fragments of 16 x 16 (180 accumulators, 219 registers, two waves per SIMD).
//
// LDS (152 KiB): two K-tile buffers x two k halves (32 deep) x {A: 18, B: 20 subtiles of 1 KiB = 16 rows x 32 k}; a subtile is one
// DMA piece and one MFMA fragment; XOR swizzle st_16x32 (16-byte chunk c of rows 8-15 sits in slot c ^ 2) applied at the SOURCE of
// the DMA and on the fragment read: conflict-free ds_read_b128.  The ring is managed per k HALF: half ks of buffer t & 1 is
// re-staged for K-tile t + 2 two phases after its last fragment read — about 1.5 K-tiles (114 KiB) are in flight, a piece has
// nine barrier slots to land.
//
// Per K-tile t, phases p = 0..5 = (ks = p / 3, row third = p % 3); a phase = LOAD section | barrier | 15 MFMAs | barrier:
//   p   fragment reads                      DMA issued (this wave's share)                  wait at the end of the MFMA section
//   0   W ks 0 (5), X ks 0 rows 0-2 (3)     -
//   1   X ks 0 rows 3-5 (3)                 W pieces of ks 1 of tile t + 1                  vmcnt: ks 1 of tile t has landed
//   2   X ks 0 rows 6-8 (3)                 X pieces of ks 1 of tile t + 1
//   3   W ks 1 (5), X ks 1 rows 0-2         -
//   4   X ks 1 rows 3-5                     W pieces of ks 0 of tile t + 2                  vmcnt: ks 0 of tile t + 1 has landed
//   5   X ks 1 rows 6-8                     X pieces of ks 0 of tile t + 2
// The waves of M-half 1 run one barrier behind those of M-half 0: in every barrier slot one wave of a SIMD multiplies while the
// other reads fragments and issues DMA.  RAW: a wait sits a whole phase before the first read of what it retires (so the lagging
// group's wait still precedes the leading group's read by a barrier); the counts are exact per wave (4 or 5 pieces per k half).
// WAR: a k half is re-staged two phases after its last read.  Measured alternatives (tools/ubench/wgemm_288.hip): all DMA of a
// k half in one LOAD section - 7 %; DMA issued between the MFMAs - 13 %; four larger phases per K-tile - 3 %; without any DMA the
// loop would run at 1780 TFLOP/s (the real-data MFMA rate), without fragment reads + 2 %.
//
// Epilogue: accumulators (+ bias + group bias) -> fp32 LDS staging in three passes of 128 columns -> coalesced 16-byte rows
// (+ residual, any storage kind; GroupNorm partials of what was stored, per 288-ROW BLOCK = per tile: mudg_gemm_stats_rows).
// Summation order over K is the K-tile order of the other kernels, but v_mfma_f32_16x16x32 adds 32 products per instruction where
// v_mfma_f32_32x32x16 adds 16: results differ from the 128 x 128 kernels' in the last bits.  The selection rule (wgemm_ok) therefore
// never looks at M — a clip's result must not depend on the batch it travels in — only at the problem's per-frame geometry.
#include "gemm_shared.h"
#include <type_traits>

#if MUDG_PLANES == 1
namespace {

constexpr int WBM = 288, WBN = 320;
constexpr int WNA = WBM / 16, WNB = WBN / 16;            // 16-row subtiles of the X / W operand tile
constexpr int W_KS = (WNA + WNB) * 1024;                 // one k half of a buffer
constexpr int W_BUF = 2 * W_KS;
constexpr int W_LOOP = 2 * W_BUF;                        // 155648
constexpr int W_STG = 132;                               // fp32 per staging row of a 128-column pass (+ 4: conflict-free row-per-lane writes)
constexpr int W_SMEM = W_LOOP;
static_assert(WBM * W_STG * 4 + WBN * 4 <= W_LOOP, "the staging rows and the column constants reuse the ring");

#define W_BARRIER()                            \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
#define W_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ f32x4 mfma16(h16x8 a, h16x8 b, f32x4 c) {
#ifdef MUDG_OPERAND_FP16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

struct KPos { int kt, tap, c; };                         // a K-tile: its index along W's K axis, its tap and first input channel

template <int MODE>
__global__ __launch_bounds__(512, 2) void wgemm_kernel(const MudgGemmDesc p, const int vflags) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;             // waves wc and wc + 4 share a SIMD: the two M halves

    // XCD-aware tile numbering (gemm.hip): every XCD a contiguous tile range, walked in 8-row groups column by column
    const int ntn = p.N / WBN, ntm = (p.M + WBM - 1) / WBM;
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
    const int m0 = tm * WBM, n0 = tn * WBN;
    constexpr int ntaps = MODE == 0 ? 1 : (MODE == 1 ? 9 : 3);

    // DMA lane geometry: lane l writes byte 16 l of a subtile (lane-linear) and therefore fetches the element whose swizzled position
    // that is: row srow, 16-byte chunk schunk of the 16 x 32 subtile
    const int pos = lane * 16;
    const int sbyte = pos ^ (((pos >> 9) & 1) << 5);
    const int srow = sbyte >> 6, schunk = (sbyte >> 4) & 3;
    const h16* X = reinterpret_cast<const h16*>(p.X);
    const h16* X2 = p.X2 ? reinterpret_cast<const h16*>(p.X2) : nullptr;
    const h16* W = reinterpret_cast<const h16*>(p.W);
    // descriptors based at the tile's first source row (minus the reach of the first tap): lane offsets are tile-row relative
    const int64_t shift = MODE == 1 ? -(int64_t)(p.Win + 1) : (MODE == 2 ? -(int64_t)p.HW : 0);
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(X + ((int64_t)m0 + shift) * p.ldx);
    const __amdgpu_buffer_rsrc_t rX2 = X2 ? make_rsrc(X2 + ((int64_t)m0 + shift) * p.ldx2) : rX;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(W + (int64_t)n0 * p.ldw);
    const int ldx2e = X2 ? p.ldx2 : p.ldx;
    const unsigned va1 = (unsigned)(srow * p.ldx) * 2u + (unsigned)schunk * 16u;
    const unsigned va2 = (unsigned)(srow * ldx2e) * 2u + (unsigned)schunk * 16u;
    const unsigned vw = (unsigned)(srow * p.ldw) * 2u + (unsigned)schunk * 16u;
    // this wave's pieces per k half: the X half wr (9 subtiles) over its four waves as 3 2 2 2; W (20 subtiles) over the eight
    // waves as 2 3 3 2 | 2 3 3 2: five pieces per k half for wc 0-2, four for wc 3
    const int a_first = wc == 0 ? 0 : 1 + 2 * wc, a_cnt = wc == 0 ? 3 : 2;
    const int b_cnt = (wc == 0 || wc == 3) ? 2 : 3;
    const int b_first = wr * 10 + (wc == 0 ? 0 : (wc == 1 ? 2 : (wc == 2 ? 5 : 8)));
    // validity of the lane's source row per tap (bit t): rows beyond M, taps that leave the image / the clip -> zero-filled by the DMA
    unsigned amask[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int m = m0 + (wr * 9 + a_first + q) * 16 + srow;
        unsigned mask = 0;
        if (q < a_cnt && m < p.M) {
            if (MODE == 0) mask = 1;
            else if (MODE == 1) {
                const int hw = p.Hout * p.Wout;
                const int f = m / hw, r = m - f * hw;
                const int oy = r / p.Wout, ox = r - oy * p.Wout;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = oy - 1 + t / 3, ix = ox - 1 + t % 3;
                    if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mask |= 1u << t;
                }
            } else {
                const int fr = (m / p.HW) % p.T;
#pragma unroll
                for (int t = 0; t < 3; ++t) if (fr + t - 1 >= 0 && fr + t - 1 < p.T) mask |= 1u << t;
            }
        }
        amask[q] = mask;
    }
    auto advance = [&](KPos& k) {
        k.kt += 1;
        if (MODE == 0) { k.c += BK; return; }
        const int t1 = k.tap + 1, c1 = k.c + BK;
        const bool slab = p.korder != 0;
        const bool wrap = slab ? (t1 == ntaps) : (c1 == p.Cin);
        k.tap = slab ? (wrap ? 0 : t1) : (wrap ? t1 : k.tap);
        k.c = slab ? (wrap ? c1 : k.c) : (wrap ? 0 : c1);
    };
    // part: 0 = all of this wave's pieces of k half ks of K-tile k, 1 = its X pieces, 2 = its W pieces
    auto stage = [&](const KPos& k, int ks, int buf, int part) {
        char* base = smem + buf * W_BUF + ks * W_KS;
        if (part != 2) {
            const bool s2 = k.c >= p.csplit;
            const int cc = s2 ? k.c - p.csplit : k.c;
            const int ld = s2 ? ldx2e : p.ldx;
            int soff = (cc + ks * 32) * 2;
            if (MODE == 1) { const int dy = k.tap / 3, dx = k.tap - 3 * dy; soff += (dy * p.Win + dx) * ld * 2; }
            if (MODE == 2) soff += k.tap * p.HW * ld * 2;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < a_cnt) {
                    const int st = wr * 9 + a_first + q;
                    const unsigned v = ((amask[q] >> k.tap) & 1u) ? (s2 ? va2 : va1) : OOB;
                    if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, (lptr_t)(base + st * 1024), 16, (int)v, soff + st * 16 * ld * 2, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lptr_t)(base + st * 1024), 16, (int)v, soff + st * 16 * ld * 2, 0, 0);
                }
        }
        if (part != 1) {
            const int soffw = (k.kt * BK + ks * 32) * 2;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < b_cnt) {
                    const int st = b_first + q;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(base + (WNA + st) * 1024), 16, (int)vw, soffw + st * 16 * p.ldw * 2, 0, 0);
                }
        }
    };

    // fragment reads: a 16 x 32 fragment is one subtile; lane l holds row l % 16, 16-byte k chunk l / 16
    const int fbyte0 = (lane & 15) * 64 + (lane >> 4) * 16;
    const int fbyte = fbyte0 ^ (((fbyte0 >> 9) & 1) << 5);
    const char* a_base = smem + (wr * 9) * 1024 + fbyte;
    const char* b_base = smem + (WNA + wc * 5) * 1024 + fbyte;

    f32x4 acc[9][5];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 af[3], bf[5];
    auto read_a = [&](int buf, int ks, int third) {
#pragma unroll
        for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const h16x8*>(a_base + buf * W_BUF + ks * W_KS + (third * 3 + i) * 1024);
    };
    auto read_b = [&](int buf, int ks) {
#pragma unroll
        for (int j = 0; j < 5; ++j) bf[j] = *reinterpret_cast<const h16x8*>(b_base + buf * W_BUF + ks * W_KS + j * 1024);
    };
    auto mma = [&](auto third_tag) {
        constexpr int third = decltype(third_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);              // (register-only MFMAs may otherwise be hoisted above the wait)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j)                 // operands swapped: a lane ends up with 4 consecutive channels of one pixel
                acc[third * 3 + i][j] = mfma16(bf[j], af[i], acc[third * 3 + i][j]);
    };
    // "everything but the pieces issued after the k half that is about to be read": that half's successor (n pieces) plus the W
    // pieces of the one after (issued in this phase)
    auto wait_half = [&](bool more) {
        if (!more) W_VMCNT(0);
        else if (a_cnt + b_cnt == 5) { if (b_cnt == 3) W_VMCNT(8); else W_VMCNT(7); }
        else W_VMCNT(6);
    };
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;

    const int nk = p.K / BK;
    KPos kA{0, 0, 0};
    stage(kA, 0, 0, 0);
    stage(kA, 1, 0, 0);
    advance(kA);                                         // kA = K-tile t + 1, kB = K-tile t + 2 at the top of iteration t
    if (nk > 1) stage(kA, 0, 1, 0);
    KPos kB = kA;
    advance(kB);
    if (nk <= 1) W_VMCNT(0); else if (a_cnt + b_cnt == 5) W_VMCNT(10); else W_VMCNT(8);      // ks 0 of tile 0 has landed
    W_BARRIER();
    if (wr == 1) W_BARRIER();                            // the stagger: M-half 1 runs one barrier behind M-half 0

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
        // phase 0
        read_b(buf, 0);
        read_a(buf, 0, 0);
        W_BARRIER();
        mma(T0{});
        W_BARRIER();
        // phase 1
        read_a(buf, 0, 1);
        if (n1) stage(kA, 1, buf ^ 1, 2);
        W_BARRIER();
        mma(T1{});
        wait_half(n1);                                   // ks 1 of tile t
        W_BARRIER();
        // phase 2
        read_a(buf, 0, 2);
        if (n1) stage(kA, 1, buf ^ 1, 1);
        W_BARRIER();
        mma(T2{});
        W_BARRIER();
        // phase 3
        read_b(buf, 1);
        read_a(buf, 1, 0);
        W_BARRIER();
        mma(T0{});
        W_BARRIER();
        // phase 4
        read_a(buf, 1, 1);
        if (n2) stage(kB, 0, buf, 2);
        W_BARRIER();
        mma(T1{});
        wait_half(n2);                                   // ks 0 of tile t + 1
        W_BARRIER();
        // phase 5
        read_a(buf, 1, 2);
        if (n2) stage(kB, 0, buf, 1);
        W_BARRIER();
        mma(T2{});
        W_BARRIER();
        kA = kB;
        advance(kB);
    }
    if (wr == 0) W_BARRIER();                            // evens out the stagger: every fragment read has retired, every DMA has landed

    // ------------------------------------------------------------------ epilogue
    float* stg = reinterpret_cast<float*>(smem);
    float* sbias = stg + WBM * W_STG;
    if (tid < WBN) {
        float b = p.bias ? p.bias[n0 + tid] : 0.f;
        if (p.gbias) b += p.gbias[(int64_t)(m0 / p.rows_per_group) * p.N + n0 + tid];      // host-checked: one group per tile
        sbias[tid] = b;
    }
    __syncthreads();
    const float alpha = p.alpha;
    const int cc = tid & 15, rr = tid >> 4;              // store loop: 16 eight-channel chunks x 32 row classes (rows rr + 32 k, k = 0..8)
    const int RK = p.R ? p.res_fp32 : 3, OK = p.out_fp32;
    const char* Rb = reinterpret_cast<const char*>(p.R);
    const int rsz = RK == KIND_F32 ? 4 : 2;
    auto run_pass = [&](auto qtag) __attribute__((always_inline)) {
        constexpr int q = decltype(qtag)::value;
        // ---- accumulators -> staging (fp32, + bias): the fragments whose 16 columns lie in [128 q, 128 q + 128)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int c0 = 80 * wc + 16 * j;
            if ((c0 >> 7) == q) {
                const f32x4 sb = *reinterpret_cast<const f32x4*>(&sbias[c0 + 4 * (lane >> 4)]);
                float* dst = stg + (wr * 144 + (lane & 15)) * W_STG + (c0 & 127) + 4 * (lane >> 4);
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = alpha * acc[i][j][e] + sb[e];
                    *reinterpret_cast<f32x4*>(dst + i * 16 * W_STG) = v;
                }
            }
        }
        __syncthreads();
        // ---- staging -> HBM
        constexpr int ncols = q < 2 ? 128 : 64;
        const bool live = cc * 8 < ncols;
        const int n = n0 + q * 128 + cc * 8;
        float gs[8], gq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < 9; kb += 3) {
            float v[3][8];
            u32x4 ra[3], rb[3];
            bool ok[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {                // every residual request of the batch before the first use
                const int64_t m = (int64_t)m0 + rr + 32 * (kb + u);
                ok[u] = live && m < p.M;
                ra[u] = zero16(); rb[u] = zero16();
                if (RK != 3 && ok[u]) {
                    const char* rp = Rb + (m * p.ldr + n) * rsz;
                    ra[u] = ld16(rp);
                    if (RK == KIND_F32) rb[u] = ld16(rp + 16);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int row = rr + 32 * (kb + u);
                const f32x4 a = *reinterpret_cast<const f32x4*>(&stg[row * W_STG + cc * 8]);
                const f32x4 b = *reinterpret_cast<const f32x4*>(&stg[row * W_STG + cc * 8 + 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[u][j] = a[j]; v[u][4 + j] = b[j]; }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (!ok[u]) continue;
                const int64_t m = (int64_t)m0 + rr + 32 * (kb + u);
                if (RK == KIND_F32) {
                    union { u32x4 w; f32x4 f; } ta, tb; ta.w = ra[u]; tb.w = rb[u];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[u][j] += ta.f[j]; v[u][4 + j] += tb.f[j]; }
                } else if (RK == KIND_F16) {
                    union { u32x4 w; f16x8 h; } t; t.w = ra[u];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[u][j] += (float)t.h[j];
                } else if (RK == KIND_OPERAND) {
                    const h16x8 t = as_h16x8(ra[u]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[u][j] += (float)t[j];
                }
                if (p.stats) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = OK == KIND_F32 ? v[u][j] : (OK == KIND_F16 ? (float)f16_sat(v[u][j]) : (float)(h16)v[u][j]);
                        gs[j] += t; gq[j] = fmaf(t, t, gq[j]);
                    }
                }
                const int64_t yoff = m * p.ldy + n;
                if (OK == KIND_F16) store8_f16(reinterpret_cast<_Float16*>(p.Y) + yoff, v[u]);
                else if (OK == KIND_F32) {
                    float* yp = reinterpret_cast<float*>(p.Y) + yoff;
                    f32x4 a, b;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[j] = v[u][j]; b[j] = v[u][4 + j]; }
                    *reinterpret_cast<f32x4*>(yp) = a;
                    *reinterpret_cast<f32x4*>(yp + 4) = b;
                } else store8_operand(reinterpret_cast<h16*>(p.Y) + yoff, p.ldy, v[u]);
            }
        }
        __syncthreads();                                 // the staging rows are free again
        if (p.stats) {
            // per 288-row block (= this tile) and channel: the 32 row classes of a chunk folded in a fixed order
            float* red = stg;
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[tid * 17 + j] = gs[j]; red[tid * 17 + 8 + j] = gq[j]; }
            __syncthreads();
            if (tid < 256) {
                const int c2 = tid >> 4, j = tid & 15;
                float t = 0.f;
                for (int k = 0; k < 32; ++k) t += red[(k * 16 + c2) * 17 + j];
                const int n2 = n0 + q * 128 + c2 * 8 + (j & 7);
                if (c2 * 8 < ncols) p.stats[((int64_t)tm * p.N + n2) * 2 + (j >> 3)] = t;
            }
            __syncthreads();
        }
    };
    run_pass(T0{});
    run_pass(T1{});
    run_pass(T2{});
}

// Variant switch GEMM_W288 (debug-variants build; read at every call so that one process can compare kernels): 0 = never, 1 = the rule
// below, 2 = every eligible problem.
int variant() { return mudg_variant("GEMM_W288", 1); }

}  // namespace

// What the kernel can run at all.
static bool wgemm_eligible(const MudgGemmDesc& d, int vflags) {
    if (d.batch != 1 || d.geglu || d.act || d.Y8 || d.subpixel || (d.mode == 1 && d.upsample)) return false;
    if (d.N % WBN != 0 || !(vflags & VF_Y) || (d.R && !(vflags & VF_R))) return false;
    const int cin = d.mode == 0 ? d.K : d.Cin;
    if ((d.K & 63) || (cin & 63) || (d.csplit & 63)) return false;
    if (d.mode == 1 && (d.stride != 1 || d.pad != 1 || d.Hin != d.Hout || d.Win != d.Wout || d.K != 9 * d.Cin)) return false;
    if (d.mode == 2 && (d.korder || d.K != 3 * d.Cin)) return false;       // (korder 1 means tiles of 8 pixels x 16 frames to the callers: gemm.hip)
    if (d.gbias && (d.rows_per_group % WBM != 0)) return false;            // one group per tile: the group bias rides in the column constants
    if (d.out_fp32 == KIND_OPERAND && (d.ldy & 7)) return false;
    // 32-bit reach of the descriptor offsets
    const int64_t ld = d.X2 && d.ldx2 > d.ldx ? d.ldx2 : d.ldx;
    int64_t rows = WBM + 16;
    if (d.mode == 1) rows += 2 * (int64_t)d.Win + 2;
    if (d.mode == 2) rows += 2 * (int64_t)d.HW;
    const int64_t lim = (int64_t)1 << 31;
    return rows * ld * 2 + (int64_t)cin * 2 + 256 < lim && (int64_t)(WBN + 16) * d.ldw * 2 + (int64_t)d.K * 2 + 256 < lim;
}

// Where it is used.  The rule never looks at M (see the header): `S`, the rows of one frame (mode 0: the caller's hint in d.HW), must be
// whole tiles — then every frame batch of the benchmarked resolution fills whole rounds of the 256 CUs.
bool mudg_wgemm_ok(const MudgGemmDesc& d, int vflags) {
    const int mode = variant();
    if (!mode || !wgemm_eligible(d, vflags)) return false;
    if (mode == 2) return true;
    const int S = d.mode == 1 ? d.Hout * d.Wout : d.HW;
    if (S <= 0 || S % WBM != 0) return false;
    // Measured per shape against the 128 x 128 kernels (tools/exp_w288.py, profiles/r5/w288_*.txt): 3x3 convs + 17 ... + 36 %;
    // temporal convs + 1 / + 6 / + 18 % at N = 320 / 640 / 1280; plain GEMMs only where K is long enough for the main loop to outweigh
    // the (un-overlapped) epilogue of a tile that is alone on its CU.
    if (d.mode == 1) return true;
    if (d.mode == 2) return d.N >= 640;
    return d.K >= 2560 || (d.K >= 1280 && d.N == 320);
}

int mudg_wgemm_launch(const MudgGemmDesc& d, int vflags, hipStream_t s) {
    static bool attr_done[MAX_DEVICES][3] = {};
    const int dev = mudg_current_device();
    if (dev < 0) MUDG_FAIL(MUDG_ELAUNCH, "gemm: no current device");
    const void* fn = d.mode == 0 ? reinterpret_cast<const void*>(&wgemm_kernel<0>)
                   : (d.mode == 1 ? reinterpret_cast<const void*>(&wgemm_kernel<1>) : reinterpret_cast<const void*>(&wgemm_kernel<2>));
    if (!attr_done[dev][d.mode]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, W_SMEM);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev][d.mode] = true;
    }
    const int tiles = ((d.M + WBM - 1) / WBM) * (d.N / WBN);
    if (d.mode == 0) hipLaunchKernelGGL((wgemm_kernel<0>), dim3(tiles), dim3(512), W_SMEM, s, d, vflags);
    else if (d.mode == 1) hipLaunchKernelGGL((wgemm_kernel<1>), dim3(tiles), dim3(512), W_SMEM, s, d, vflags);
    else hipLaunchKernelGGL((wgemm_kernel<2>), dim3(tiles), dim3(512), W_SMEM, s, d, vflags);
    return mudg_check_launch("mudg_gemm");
}
#else
bool mudg_wgemm_ok(const MudgGemmDesc&, int) { return false; }
int mudg_wgemm_launch(const MudgGemmDesc&, int, hipStream_t) { MUDG_FAIL(MUDG_EINVAL, "gemm: no 288 x 320 kernel in this build"); }
#endif
