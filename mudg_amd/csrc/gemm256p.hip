// gemm256p.hip — "ping-pong" variant of the 256x256x64 contraction kernel (same MudgGemmDesc semantics; FAST
// problems only, see gemm.hip) for long-K shapes.
//
// Eight waves (512 threads, one workgroup per CU, two waves per SIMD) as 2 (M) x 4 (N); a wave owns 128 x 64 of the
// tile (4 x 2 v_mfma_f32_32x32x16 blocks, 128 accumulators).  Waves 0-3 land on the four SIMDs, waves 4-7 on the same
// four again, so every SIMD hosts one wave of group A (rows 0-127) and one of group B (rows 128-255).  The K loop is
// cut into phases — per K-tile four, each = a LOAD section (ds_read of the fragments the phase multiplies, plus the
// DMA of one operand half-tile of a later K-tile) and an MFMA section (8 MFMAs on 4 independent accumulators) — with
// a workgroup barrier after every section, and group B runs one barrier behind group A: while one wave of a SIMD
// issues MFMAs the other one issues LDS reads and DMA, and they swap at the next barrier.
//
//   phase (K-tile kt, buffer kb = kt & 1)   ds_read                     MFMA (acc[ni][mi], k-steps)      DMA issued
//   P0                                      W[n 0,1][k 0,1] X[m 0,1][k 0,1]   mi 0,1 x ni 0,1 x k 0,1     X rows   0-127 of kt+1 -> kb^1
//   P1                                      W[n 0,1][k 2,3] X[m 0,1][k 2,3]   mi 0,1 x ni 0,1 x k 2,3     X rows 128-255 of kt+1 -> kb^1
//   P2                                      X[m 2,3][k 0,1]                   mi 2,3 x ni 0,1 x k 0,1     W rows   0-127 of kt+2 -> kb
//   P3                                      X[m 2,3][k 2,3]                   mi 2,3 x ni 0,1 x k 2,3     W rows 128-255 of kt+2 -> kb
//
// Ordering rules this schedule obeys (MI355X: LDS-DMA is ordered for a reader only by the issuer's vmcnt wait
// followed by a barrier the reader passes; a region may be re-staged only after a barrier that follows the
// lgkmcnt(0) of its last readers):
//   * every LOAD section ends with s_waitcnt lgkmcnt(0) BEFORE its barrier, so a passed barrier retires those reads;
//   * W of K-tile kt is last read in P1, X in P3 — the DMA that overwrites W(kt) is first issued in P2 of kt by
//     group A, one full slot after group B's P1 reads; the DMA that overwrites X(kt) is issued in P0/P1 of kt+1;
//   * K-tile kt+1 is complete once X rows 128-255 (issued in P1 of kt) landed: P3 waits vmcnt(4) — the four pieces
//     issued after it (W halves of kt+2) may stay in flight — and both groups pass >= 1 barrier before P0 of kt+1.
// Two schedules: the four-phase one tabulated above and (default) a two-phase one with 16 MFMAs per section (see
// phase2 below; +1...5 %).  Rejected before this one: an 8-wave kernel with a single barrier per K-tile (MFMA, ds_read and DMA streams did not
// overlap: 660 TFLOP/s) — see gemm256.hip for the 16-wave kernel that is used when this one does not apply.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BK = 64;
constexpr int KB_BYTES = 256 * 64 * 2;          // one operand tile of one K-tile: 32 KiB
constexpr int W_BASE = 2 * KB_BYTES;            // X[0] X[1] W[0] W[1]
constexpr int SMEM_MAIN = 4 * KB_BYTES;         // 128 KiB
constexpr int STGLD = 132;
constexpr int VF_Y = 1, VF_R = 2;
constexpr unsigned OOB = 0x80000000u;

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const h16* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(base), 0, (int)0x80000000u, 0x00020000);
}

__device__ __forceinline__ float gelu_fast3(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(e, x));
}

#define SECTION_BARRIER()                      \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

template <int MODE, bool PH2>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(const MudgGemmDesc p, const int vflags, const int ablate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    const int ntn = (p.N + 255) >> 8;
    int tile;
    {
        const int total = gridDim.x, q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    // Tile order inside the XCD's range: groups of 8 tile rows, column by column, so that the ~64 tiles an XCD runs at
    // once form an 8 x 8 patch (16 operand panels through its L2) instead of one 1 x 64 strip (65 panels) when N is wide.
    int tm, tn;
    {
        const int ntm = (p.M + 256 - 1) / 256;
        const int per = 8 * ntn, g = tile / per, first = g * 8;
        const int gsz = (ntm - first) < 8 ? (ntm - first) : 8;
        const int r = tile - g * per;
        tn = r / gsz;
        tm = first + (r - tn * gsz);
    }
    const int m0 = tm * 256, n0 = tn * 256;
    const int64_t bz = blockIdx.z;
    const h16* X = reinterpret_cast<const h16*>(p.X) + bz * p.sX;
    const h16* X2 = p.X2 ? reinterpret_cast<const h16*>(p.X2) + bz * p.sX : nullptr;
    const h16* W = reinterpret_cast<const h16*>(p.W) + bz * p.sW;

    // ---- DMA geometry: half-tile h (rows 128h .. 128h+127) is 16 one-KiB pieces; wave w issues pieces 2w, 2w+1, i.e.
    // rows 128h + 16w + 8i + (lane >> 3), and fetches chunk (lane & 7) ^ ((row >> 1) & 7) of that row (gemm.hip swizzle).
    const int rsub = lane >> 3, slot = lane & 7;
    unsigned vx[2][2], vx2[2][2], vw[2][2], vmask[2][2];
    __amdgpu_buffer_rsrc_t rX, rX2, rW;
    {
        int64_t pix0 = m0;
        if (MODE == 1) {
            const int hw = p.Hout * p.Wout;
            const int f = m0 / hw, r = m0 - f * hw;
            const int oy = r / p.Wout, ox = r - oy * p.Wout;
            pix0 = ((int64_t)f * p.Hin + oy * p.stride) * p.Win + ox * p.stride;
        }
        const int64_t shift = MODE == 1 ? -(int64_t)(p.pad * p.Win + p.pad) : (MODE == 2 ? -(int64_t)p.HW : 0);
        rX = make_rsrc(X + (pix0 + shift) * p.ldx);
        rX2 = X2 ? make_rsrc(X2 + (pix0 + shift) * p.ldx2) : rX;
        rW = make_rsrc(W + (int64_t)n0 * p.ldw);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rl = 128 * h + 16 * wave + 8 * i + rsub;
                const unsigned cb = (unsigned)(slot ^ ((rl >> 1) & 7)) * 16u;
                const int m = m0 + rl;
                const bool rv = m < p.M;
                int rel = rl;
                unsigned mask = rv ? 1u : 0u;
                if (MODE == 1) {
                    const int hw = p.Hout * p.Wout;
                    const int f = m / hw, r = m - f * hw;
                    const int oy = r / p.Wout, ox = r - oy * p.Wout;
                    rel = (int)((((int64_t)f * p.Hin + oy * p.stride) * p.Win + ox * p.stride) - pix0);
                    const int by = oy * p.stride - p.pad, bx = ox * p.stride - p.pad;
                    mask = 0;
                    if (rv) {
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const int iy = by + t / 3, ix = bx + t % 3;
                            if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mask |= 1u << t;
                        }
                    }
                } else if (MODE == 2) {
                    const int tt = (m / p.HW) % p.T;
                    mask = 0;
                    if (rv) {
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            const int it = tt + t - 1;
                            if (it >= 0 && it < p.T) mask |= 1u << t;
                        }
                    }
                }
                vmask[h][i] = mask;
                vx[h][i] = (MODE == 0 && !rv) ? OOB : (unsigned)rel * (unsigned)p.ldx * 2u + cb;
                vx2[h][i] = (MODE == 0 && !rv) ? OOB : (unsigned)rel * (unsigned)(X2 ? p.ldx2 : p.ldx) * 2u + cb;
                vw[h][i] = (n0 + rl < p.N) ? (unsigned)rl * (unsigned)p.ldw * 2u + cb : OOB;
            }
    }

    const int nk = p.K / BK;                   // FAST: K % 64 == 0
    int tap_s = 0, c_s = 0;                    // tap / channel of the next K-tile whose X rows 0-127 get issued
    int x_soff = 0, x_tap = 0;                 // ... and the values saved for its rows 128-255 one phase later
    bool x_s2 = false;

    auto issue_x = [&](int h, int kb) {
        if (h == 0) {
            x_s2 = c_s >= p.csplit;
            const int cc = x_s2 ? c_s - p.csplit : c_s;
            const int ld = x_s2 ? p.ldx2 : p.ldx;
            if (MODE == 0) x_soff = cc * 2;
            else if (MODE == 1) { const int dy = tap_s / 3, dx = tap_s - 3 * dy; x_soff = ((dy * p.Win + dx) * ld + cc) * 2; }
            else x_soff = (tap_s * p.HW * ld + cc) * 2;
            x_tap = tap_s;
            if (MODE == 0) {
                c_s += BK;
            } else {
                const int t1 = tap_s + 1, c1 = c_s + BK;
                const bool slab = MODE == 1 && p.korder;
                const bool wrap = slab ? (t1 == 9) : (c1 == p.Cin);
                tap_s = slab ? (wrap ? 0 : t1) : (wrap ? t1 : tap_s);
                c_s = slab ? (wrap ? c1 : c_s) : (wrap ? 0 : c1);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned v = x_s2 ? vx2[h][i] : vx[h][i];
            if (MODE != 0) v = ((vmask[h][i] >> x_tap) & 1u) ? v : OOB;
            lptr_t dst = (lptr_t)(smem + kb * KB_BYTES + h * (KB_BYTES / 2) + (16 * wave + 8 * i) * 128);
            if (x_s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, dst, 16, (int)v, x_soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, dst, 16, (int)v, x_soff, 0, 0);
        }
    };
    auto issue_w = [&](int h, int kb, int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lptr_t dst = (lptr_t)(smem + W_BASE + kb * KB_BYTES + h * (KB_BYTES / 2) + (16 * wave + 8 * i) * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, dst, 16, (int)vw[h][i], kt * (BK * 2), 0, 0);
        }
    };

    // ---- fragment read addresses (bytes): row (wr*128 + l31 [+ 32 mi]) of X, row (wc*64 + l31 [+ 32 ni]) of W,
    // k-step ks -> chunk 2 ks + hi stored in slot chunk ^ ((l31 >> 1) & 7); the K-buffer / block offsets are immediates.
    const int swz = (l31 >> 1) & 7;
    int ax[4], aw[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int so = (((ks * 2 + hi) ^ swz) << 4);
        ax[ks] = (wr * 128 + l31) * 128 + so;
        aw[ks] = W_BASE + (wc * 64 + l31) * 128 + so;
    }

    f32x16 acc[2][4];       // [ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    h16x8 wf[2][4];         // [ni][ks], live across the four phases of a K-tile
    h16x8 xf[2][PH2 ? 4 : 2];   // [mi local][ks local], reloaded every phase

    // ---- prologue: K-tile 0 whole, W of K-tile 1 (what P2 / P3 of a K-tile "-1" would have issued)
    issue_x(0, 0);
    issue_x(1, 0);
    issue_w(0, 0, 0);
    issue_w(1, 0, 0);
    if (nk > 1) {
        issue_w(0, 1, 1);
        issue_w(1, 1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SECTION_BARRIER();
    if (wr == 1) SECTION_BARRIER();            // group B runs one barrier behind group A

    auto phase = [&](auto Pc, auto KBc, int kt) {
        constexpr int P = decltype(Pc)::value, KB = decltype(KBc)::value;
        constexpr int K0 = (P & 1) * 2;        // first k-step of this phase
        constexpr int M0 = (P >> 1) * 2;       // first m-block of this phase
        // ---------------- LOAD section
        if (P < 2) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    wf[ni][K0 + k] = *reinterpret_cast<const h16x8*>(smem + aw[K0 + k] + KB * KB_BYTES + ni * 4096);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int k = 0; k < 2; ++k)
                xf[mi][k] = *reinterpret_cast<const h16x8*>(smem + ax[K0 + k] + KB * KB_BYTES + (M0 + mi) * 4096);
        const bool dma = !(ablate & 1);
        if (P == 0) { if (kt + 1 < nk && dma) issue_x(0, KB ^ 1); }
        else if (P == 1) { if (kt + 1 < nk && dma) issue_x(1, KB ^ 1); }
        else if (P == 2) { if (kt + 2 < nk && dma) issue_w(0, KB, kt + 2); }
        else {
            // K-tile kt+1 is complete when X rows 128-255 (P1) have landed; younger: the W halves of kt+2 (P2, P3)
            if (kt + 2 < nk && dma) {
                issue_w(1, KB, kt + 2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SECTION_BARRIER();
        // ---------------- MFMA section
        __builtin_amdgcn_s_setprio(1);
        if (!(ablate & 4)) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[ni][M0 + mi] = MFMA_32x32x16(wf[ni][K0 + k], xf[mi][k], acc[ni][M0 + mi]);
        }
        __builtin_amdgcn_s_setprio(0);
        SECTION_BARRIER();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // Two-phase variant (PH2): a phase = one M half x all four k-steps = 16 MFMAs per section, half as many barriers.
    //   P0: ds_read W[n 0,1][k 0-3] X[m 0,1][k 0-3]; DMA X rows 0-255 of kt+1 -> kb^1 (X(kt-1) was last read in P1 of kt-1)
    //   P1: ds_read X[m 2,3][k 0-3];                 DMA W rows 0-255 of kt+2 -> kb   (W(kt) was last read in P0 of kt);
    //       then vmcnt(4): everything older than those four pieces — X(kt+1), W(kt+1) — has landed.
    auto phase2 = [&](auto Pc, auto KBc, int kt) {
        constexpr int P = decltype(Pc)::value, KB = decltype(KBc)::value;
        constexpr int M0 = P * 2;
        if (P == 0) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    wf[ni][k] = *reinterpret_cast<const h16x8*>(smem + aw[k] + KB * KB_BYTES + ni * 4096);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                xf[mi][k] = *reinterpret_cast<const h16x8*>(smem + ax[k] + KB * KB_BYTES + (M0 + mi) * 4096);
        const bool dma = !(ablate & 1);
        if (P == 0) {
            if (kt + 1 < nk && dma) { issue_x(0, KB ^ 1); issue_x(1, KB ^ 1); }
        } else {
            if (kt + 2 < nk && dma) {
                issue_w(0, KB, kt + 2); issue_w(1, KB, kt + 2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SECTION_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        if (!(ablate & 4)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[ni][M0 + mi] = MFMA_32x32x16(wf[ni][k], xf[mi][k], acc[ni][M0 + mi]);
        }
        __builtin_amdgcn_s_setprio(0);
        SECTION_BARRIER();
    };

    int kt = 0;
    if constexpr (PH2) {
        for (; kt + 1 < nk; kt += 2) {
            phase2(I0{}, I0{}, kt); phase2(I1{}, I0{}, kt);
            phase2(I0{}, I1{}, kt + 1); phase2(I1{}, I1{}, kt + 1);
        }
        if (kt < nk) { phase2(I0{}, I0{}, kt); phase2(I1{}, I0{}, kt); }
    } else {
    for (; kt + 1 < nk; kt += 2) {
        phase(I0{}, I0{}, kt); phase(I1{}, I0{}, kt); phase(I2{}, I0{}, kt); phase(I3{}, I0{}, kt);
        phase(I0{}, I1{}, kt + 1); phase(I1{}, I1{}, kt + 1); phase(I2{}, I1{}, kt + 1); phase(I3{}, I1{}, kt + 1);
    }
    if (kt < nk) { phase(I0{}, I0{}, kt); phase(I1{}, I0{}, kt); phase(I2{}, I0{}, kt); phase(I3{}, I0{}, kt); }
    }
    if (wr == 0) SECTION_BARRIER();            // group A waits for group B's last section
    __syncthreads();

    // ------------------------------------------------------------------ epilogue: one 128x128 quadrant at a time
    float* stg = reinterpret_cast<float*>(smem);
    float* sbias = stg + 128 * STGLD;
    const float alpha = p.alpha;
    const int Nout = p.geglu ? p.N / 2 : p.N;
    const h16* R = (p.R && !p.res_fp32) ? reinterpret_cast<const h16*>(p.R) + bz * p.sR : nullptr;
    const float* Rf = (p.R && p.res_fp32) ? reinterpret_cast<const float*>(p.R) + bz * p.sR : nullptr;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int mq = m0 + h * 128, nq = n0 + g * 128;
            if (tid < 128) sbias[tid] = (p.bias && nq + tid < p.N) ? p.bias[nq + tid] : 0.f;
            __syncthreads();
            if (wr == h && (wc >> 1) == g) {                 // the two waves that own this quadrant stage it
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int ml = mi * 32 + l31;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int nl = (wc & 1) * 64 + ni * 32 + 8 * q + 4 * hi;
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = alpha * acc[ni][mi][4 * q + j] + sbias[nl + j];
                            if (p.act && !p.geglu) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = gelu_fast3(v[j]);
                            }
                            *reinterpret_cast<f32x4*>(&stg[ml * STGLD + nl]) = v;
                        }
                }
            }
            __syncthreads();
            const int NT = p.geglu ? 64 : 128;
            const int nout0 = p.geglu ? nq / 2 : nq;
            const int cpr = NT / 8;
            float gs[8], gq[8];        // GroupNorm partials of this thread's 8 output channels (p.stats)
#pragma unroll
            for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
            for (int c = tid; c < 128 * cpr; c += 512) {
                const int row = c / cpr, cc = c - row * cpr;
                const int m = mq + row, n = nout0 + cc * 8;
                if (m >= p.M || n >= Nout) continue;
                const int nvalid = (Nout - n) < 8 ? (Nout - n) : 8;
                float v[8];
                if (!p.geglu) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&stg[row * STGLD + cc * 8]);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(&stg[row * STGLD + cc * 8 + 4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
                } else {
                    // staged columns: [64j', 64j'+32) values, [64j'+32, 64j'+64) their gates; 8 outputs never straddle
                    const int o0 = cc * 8, blk = o0 >> 5, in = o0 & 31;
                    const float* sv = &stg[row * STGLD + blk * 64 + in];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = sv[j] * gelu_fast3(sv[32 + j]);
                }
                if (p.gbias) {
                    const float* gb = p.gbias + (int64_t)(m / p.rows_per_group) * Nout + n;
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += gb[j];
                }
                if (R) {
                    const h16* rp = R + (int64_t)m * p.ldr + n;
                    if (nvalid == 8 && (vflags & VF_R)) {
                        const h16x8 rr = as_h16x8(ld16(rp));
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += (float)rr[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += (float)rp[j];
                    }
                }
                if (Rf) {
                    const float* rp = Rf + (int64_t)m * p.ldr + n;
                    if (nvalid == 8 && (vflags & VF_R)) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(rp), b = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[j] += a[j]; v[4 + j] += b[j]; }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += rp[j];
                    }
                }
                if (p.stats) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = (j < nvalid) ? (p.out_fp32 ? v[j] : (float)(h16)v[j]) : 0.f;
                        gs[j] += t; gq[j] = fmaf(t, t, gq[j]);
                    }
                }
                if (p.out_fp32) {
                    float* yp = reinterpret_cast<float*>(p.Y) + bz * p.sY + (int64_t)m * p.ldy + n;
                    if (nvalid == 8 && (vflags & VF_Y)) {
                        f32x4 a, b;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
                        *reinterpret_cast<f32x4*>(yp) = a;
                        *reinterpret_cast<f32x4*>(yp + 4) = b;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (j < nvalid) yp[j] = v[j];
                    }
                } else {
                    h16* yp = reinterpret_cast<h16*>(p.Y) + bz * p.sY + (int64_t)m * p.ldy + n;
                    if (nvalid == 8 && (vflags & VF_Y)) {
                        h16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (h16)v[j];
                        st16(yp, as_u32x4(o));
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (j < nvalid) yp[j] = (h16)v[j];
                    }
                }
            }
            __syncthreads();
            if (p.stats) {         // fold the 512 / cpr threads of each channel chunk in a fixed order (staging tile is free)
                float* red = stg;
#pragma unroll
                for (int j = 0; j < 8; ++j) { red[tid * 17 + j] = gs[j]; red[tid * 17 + 8 + j] = gq[j]; }
                __syncthreads();
                if (tid < cpr * 16) {
                    const int cc = tid >> 4, j = tid & 15;
                    float t = 0.f;
                    for (int k = cc; k < 512; k += cpr) t += red[k * 17 + j];
                    const int n = nout0 + cc * 8 + (j & 7);
                    if (n < Nout && mq < p.M) p.stats[((int64_t)(mq / 128) * Nout + n) * 2 + (j >> 3)] = t;
                }
                __syncthreads();
            }
        }
}

template <int MODE, bool PH2>
int launch256p(const MudgGemmDesc& d, int vflags, int ablate, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<MODE, PH2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_MAIN);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm256p: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ((d.M + 255) / 256) * ((d.N + 255) / 256);
    hipLaunchKernelGGL((gemm256p_kernel<MODE, PH2>), dim3(tiles, 1, d.batch), dim3(512), SMEM_MAIN, s, d, vflags, ablate);
    return mudg_check_launch("mudg_gemm[256p]");
}

}  // namespace

// Called by mudg_gemm256_dispatch (gemm256.hip) for FAST problems.
int mudg_gemm256p_dispatch(const MudgGemmDesc& d, int vflags, hipStream_t s) {
    static int ablate = -1;                  // MUDG_ABLATE bit 0: no DMA in the K loop, bit 2: no MFMA (timing experiments only)
    if (ablate < 0) { const char* e = getenv("MUDG_ABLATE"); ablate = e ? atoi(e) : 0; }
    static int ph2 = -1;                     // MUDG_PP_PHASES=4 selects the four-phase schedule (A/B measurements)
    if (ph2 < 0) { const char* e = getenv("MUDG_PP_PHASES"); ph2 = (e && atoi(e) == 4) ? 0 : 1; }
    if (ph2) {
        if (d.mode == 0) return launch256p<0, true>(d, vflags, ablate, s);
        if (d.mode == 1) return launch256p<1, true>(d, vflags, ablate, s);
        return launch256p<2, true>(d, vflags, ablate, s);
    }
    if (d.mode == 0) return launch256p<0, false>(d, vflags, ablate, s);
    if (d.mode == 1) return launch256p<1, false>(d, vflags, ablate, s);
    return launch256p<2, false>(d, vflags, ablate, s);
}
