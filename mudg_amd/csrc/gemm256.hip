// gemm256.hip — the 16-wave large-tile variant of the contraction kernel (same MudgGemmDesc semantics as gemm.hip) for
// shapes with enough 256x256 tiles to fill the chip; in the default selection it serves the nearest-2x upsample convs
// (mudg_gemm, use_gemm256), everything else having moved back to the 128x128 kernels once those were tuned.
//
// Why it exists: a 256x256x64 block tile stages half the bytes per FLOP of the 128x128 tile (whose first version was
// bound by the L2 -> LDS DMA stream at ~600-750 TFLOP/s).  Sixteen waves (1024 threads, one workgroup per CU, 4 waves per SIMD) form a 4 x 4
// grid and each keeps the 64x64 wave tile of gemm.hip (2x2 v_mfma_f32_32x32x16, 64 accumulators, ~125 VGPRs), so the
// per-wave instruction stream is unchanged while each K-tile's 64 KiB arrive with four 1-KiB global_load_lds
// instructions per wave.  Two K-tile buffers (2 x 64 KiB, XOR-swizzled [256][64] tiles as in gemm.hip), one barrier
// per K-tile: the DMA of tile k+1 flies under 16 MFMAs x 4 waves per SIMD.
//
// (An 8-wave variant with 128x64 wave tiles, half-tile pieces and counted vmcnt was tried first: correct, but its two
// waves per SIMD ran in lock-step and the MFMA, ds_read and DMA streams did not overlap — MFMA-only 424 us, DMA-only
// 406-498 us, ds_read-only 274 us, all together 823 us on conv 18x32 2560->1280 — and staggering the two wave groups
// with a barrier per sub-phase made it slower still.  Four independent waves per SIMD give the scheduler that overlap
// for free.)
//
// Epilogue: four passes (one 128x128 quadrant each) through the same fp32 LDS tile as gemm.hip; GEGLU is applied in
// the coalesced pass (value / gate columns of the usual [32 value | 32 gate] packing sit 32 apart in the staged tile).
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BK = 64;
constexpr int TROWS = 256;                     // rows per operand tile
constexpr int OTILE = TROWS * 64;              // elements per operand tile
constexpr int KBUF = 2 * OTILE;                // elements per K-tile buffer: X, W
constexpr int SMEM_MAIN = 2 * KBUF * 2;        // bytes (131072)
constexpr int STGLD = 132;
constexpr int SMEM_STG = 128 * STGLD * 4 + 128 * 4;
constexpr int SMEM_BYTES = SMEM_MAIN > SMEM_STG ? SMEM_MAIN : SMEM_STG;
constexpr int VF_Y = 1, VF_R = 2;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float gelu_fast2(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(e, x));
}


constexpr unsigned OOB = 0x80000000u;          // see gemm.hip: out-of-range voffset -> the buffer load zero-fills

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const h16* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(base), 0, (int)0x80000000u, 0x00020000);
}

template <int MODE, bool FAST>
__global__ __launch_bounds__(1024, 4) void gemm256_kernel(const MudgGemmDesc p, const int vflags, const h16* __restrict__ zpage, const int ablate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* L = reinterpret_cast<h16*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..15
    const int wm = wave & 3, wn = wave >> 2;
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

    // DMA geometry: wave w stages rows [16w, 16w+16) of both operand tiles with two 1-KiB instructions each; in
    // instruction i lane l lands in row 16w + 8i + (l >> 3), slot l & 7 and fetches chunk slot ^ ((row >> 1) & 7).
    const int rsub = lane >> 3, slot = lane & 7;
    int ch[2], rl[2], rm[2], ra[2], rb[2], rc[2];
    bool rv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rl[i] = 16 * wave + 8 * i + rsub;
        ch[i] = slot ^ ((rl[i] >> 1) & 7);
        const int m = m0 + rl[i];
        rm[i] = m;
        rv[i] = m < p.M;
        ra[i] = rb[i] = rc[i] = 0;
        if (MODE == 1) {
            const int hw = p.Hout * p.Wout;
            const int f = m / hw, r = m - f * hw;
            const int oy = r / p.Wout, ox = r - oy * p.Wout;
            ra[i] = f * p.Hin * p.Win;
            rb[i] = oy * p.stride - p.pad;
            rc[i] = ox * p.stride - p.pad;
        } else if (MODE == 2) {
            rb[i] = (m / p.HW) % p.T;
        }
    }
    const bool tap_uniform = MODE != 0 && (p.Cin & 63) == 0;

    // FAST path (same scheme as gemm.hip): block-relative buffer descriptors, invariant lane offsets, tap bits.
    __amdgpu_buffer_rsrc_t rX, rX2, rW;
    unsigned vx[2], vx2[2], vw[2], vmask[2];
    int tap_s = 0, c_s = 0;
    if constexpr (FAST) {
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
        for (int i = 0; i < 2; ++i) {
            int rel = rl[i];
            unsigned mask = rv[i] ? 1u : 0u;
            if (MODE == 1) {
                rel = (int)((int64_t)(ra[i] + (rb[i] + p.pad) * p.Win + rc[i] + p.pad) - pix0);
                mask = 0;
                if (rv[i]) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int iy = rb[i] + t / 3, ix = rc[i] + t % 3;
                        if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mask |= 1u << t;
                    }
                }
            } else if (MODE == 2) {
                mask = 0;
                if (rv[i]) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int it = rb[i] + t - 1;
                        if (it >= 0 && it < p.T) mask |= 1u << t;
                    }
                }
            }
            vmask[i] = mask;
            const unsigned cb = (unsigned)ch[i] * 16u;
            vx[i] = (MODE == 0 && !rv[i]) ? OOB : (unsigned)rel * (unsigned)p.ldx * 2u + cb;
            vx2[i] = (MODE == 0 && !rv[i]) ? OOB : (unsigned)rel * (unsigned)(X2 ? p.ldx2 : p.ldx) * 2u + cb;
            vw[i] = (n0 + rl[i] < p.N) ? (unsigned)rl[i] * (unsigned)p.ldw * 2u + cb : OOB;
        }
    }

    auto issue_fast = [&](int kt) {
        const bool s2 = c_s >= p.csplit;
        const int cc = s2 ? c_s - p.csplit : c_s;
        const int ld = s2 ? p.ldx2 : p.ldx;
        int soff;
        if (MODE == 0) soff = cc * 2;
        else if (MODE == 1) { const int dy = tap_s / 3, dx = tap_s - 3 * dy; soff = ((dy * p.Win + dx) * ld + cc) * 2; }
        else soff = (tap_s * p.HW * ld + cc) * 2;
        const int soffw = kt * (BK * 2);
        h16* xdst = L + (kt & 1) * KBUF;
        h16* wdst = xdst + OTILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned v = s2 ? vx2[i] : vx[i];
            if (MODE != 0) v = ((vmask[i] >> tap_s) & 1u) ? v : OOB;
            lptr_t lx = (lptr_t)(xdst + (16 * wave + 8 * i) * 64);
            if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, lx, 16, (int)v, soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, lx, 16, (int)v, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(wdst + (16 * wave + 8 * i) * 64), 16, (int)vw[i], soffw, 0, 0);
        }
        if (MODE == 0) {
            c_s += BK;
        } else {                                   // select form: the branchy update sent tap_s / c_s to scratch memory
            const int t1 = tap_s + 1, c1 = c_s + BK;
            const bool slab = MODE == 1 && p.korder;
            const bool wrap = slab ? (t1 == 9) : (c1 == p.Cin);
            tap_s = slab ? (wrap ? 0 : t1) : (wrap ? t1 : tap_s);
            c_s = slab ? (wrap ? c1 : c_s) : (wrap ? 0 : c1);
        }
    };

    auto issue_tiles = [&](int kt) {
        if constexpr (FAST) { issue_fast(kt); return; }
        const int k0 = kt * BK;
        int tap_u = 0, c_u = 0;
        if (MODE != 0 && tap_uniform) {
            if (MODE == 1 && p.korder) { const int slab = kt / 9; tap_u = kt - slab * 9; c_u = slab * 64; }
            else { tap_u = k0 / p.Cin; c_u = k0 - tap_u * p.Cin; }
        }
        h16* xdst = L + (kt & 1) * KBUF;
        h16* wdst = xdst + OTILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + ch[i] * 8;
            const bool kv = k < p.K;
            const h16* src = zpage;
            if (MODE == 0) {
                if (rv[i] && kv)
                    src = (k < p.csplit) ? X + (int64_t)rm[i] * p.ldx + k : X2 + (int64_t)rm[i] * p.ldx2 + (k - p.csplit);
            } else {
                int tap, c;
                if (tap_uniform) { tap = tap_u; c = c_u + ch[i] * 8; }
                else { tap = k / p.Cin; c = k - tap * p.Cin; }
                const h16* base = X; int cc = c, ld = p.ldx;
                if (c >= p.csplit) { base = X2; cc = c - p.csplit; ld = p.ldx2; }
                if (MODE == 1) {
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const int hlim = p.upsample ? 2 * p.Hin : p.Hin;
                    const int wlim = p.upsample ? 2 * p.Win : p.Win;
                    int iy = rb[i] + dy, ix = rc[i] + dx;
                    const bool ok = rv[i] && kv && iy >= 0 && iy < hlim && ix >= 0 && ix < wlim;
                    if (p.upsample) { iy >>= 1; ix >>= 1; }
                    if (ok) src = base + (int64_t)(ra[i] + iy * p.Win + ix) * ld + cc;
                } else {
                    const int it = rb[i] + tap - 1;
                    if (rv[i] && kv && it >= 0 && it < p.T)
                        src = base + ((int64_t)rm[i] + (int64_t)(tap - 1) * p.HW) * ld + cc;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xdst + (16 * wave + 8 * i) * 64), 16, 0, 0);
            const int n = n0 + rl[i];
            const h16* wsrc = (n < p.N && kv) ? W + (int64_t)n * p.ldw + k : zpage;
            __builtin_amdgcn_global_load_lds((gptr_t)wsrc, (lptr_t)(wdst + (16 * wave + 8 * i) * 64), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];     // [ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    const int sw = (l31 >> 1) & 7;
    issue_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                       // vmcnt(0) + barrier: tile kt has landed, tile kt-1's buffer is free
        if (kt + 1 < nk && !(ablate & 1)) issue_tiles(kt + 1);
        const h16* xs = L + (kt & 1) * KBUF + (wm * 64 + l31) * 64;
        const h16* ws = L + (kt & 1) * KBUF + OTILE + (wn * 64 + l31) * 64;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int off = ((ks * 2 + hi) ^ sw) << 3;
            h16x8 wf[2], xf[2];
            wf[0] = *reinterpret_cast<const h16x8*>(ws + off);
            wf[1] = *reinterpret_cast<const h16x8*>(ws + 32 * 64 + off);
            xf[0] = *reinterpret_cast<const h16x8*>(xs + off);
            xf[1] = *reinterpret_cast<const h16x8*>(xs + 32 * 64 + off);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[ni][mi] = MFMA_32x32x16(wf[ni], xf[mi], acc[ni][mi]);
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ epilogue: one 128x128 quadrant at a time
    float* stg = reinterpret_cast<float*>(smem);
    float* sbias = stg + 128 * STGLD;
    const float alpha = p.alpha;
    const int Nout = p.geglu ? p.N / 2 : p.N;
    const h16* R = (p.R && p.res_fp32 == KIND_OPERAND) ? reinterpret_cast<const h16*>(p.R) + bz * p.sR : nullptr;
    const float* Rf = (p.R && p.res_fp32 == KIND_F32) ? reinterpret_cast<const float*>(p.R) + bz * p.sR : nullptr;
    const _Float16* Rh = (p.R && p.res_fp32 == KIND_F16) ? reinterpret_cast<const _Float16*>(p.R) + bz * p.sR : nullptr;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int mq = m0 + h * 128, nq = n0 + g * 128;
            if (tid < 128) sbias[tid] = (p.bias && nq + tid < p.N) ? p.bias[nq + tid] : 0.f;
            __syncthreads();
            if ((wm >> 1) == h && (wn >> 1) == g) {          // the four waves that own this quadrant stage it
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int ml = (wm & 1) * 64 + mi * 32 + l31;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int nl = (wn & 1) * 64 + ni * 32 + 8 * q + 4 * hi;
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = alpha * acc[ni][mi][4 * q + j] + sbias[nl + j];
                            if (p.act && !p.geglu) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = gelu_fast2(v[j]);
                            }
                            *reinterpret_cast<f32x4*>(&stg[ml * STGLD + nl]) = v;
                        }
                }
            }
            __syncthreads();
            const int NT = p.geglu ? 64 : 128;
            const int nout0 = p.geglu ? nq / 2 : nq;
            const int cpr = NT / 8;
            for (int c = tid; c < 128 * cpr; c += 1024) {
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
                    for (int j = 0; j < 8; ++j) v[j] = sv[j] * gelu_fast2(sv[32 + j]);
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
                if (Rh) {
                    const _Float16* rp = Rh + (int64_t)m * p.ldr + n;
                    if (nvalid == 8 && (vflags & VF_R)) {
                        float rr[8];
                        load8_f16(rp, rr);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += rr[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += (float)rp[j];
                    }
                }
                if (p.out_fp32 == KIND_F16) {
                    _Float16* yp = reinterpret_cast<_Float16*>(p.Y) + bz * p.sY + (int64_t)m * p.ldy + n;
                    if (nvalid == 8 && (vflags & VF_Y)) {
                        store8_f16(yp, v);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (j < nvalid) yp[j] = (_Float16)v[j];
                    }
                } else if (p.out_fp32) {
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
        }
}

template <int MODE, bool FAST>
int launch256(const MudgGemmDesc& d, int vflags, const h16* zp, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<MODE, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm256: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ((d.M + 255) / 256) * ((d.N + 255) / 256);
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("MUDG_ABLATE"); ablate = e ? atoi(e) : 0; }
    hipLaunchKernelGGL((gemm256_kernel<MODE, FAST>), dim3(tiles, 1, d.batch), dim3(1024), SMEM_BYTES, s, d, vflags, zp, ablate);
    return mudg_check_launch("mudg_gemm[256]");
}

}  // namespace

// Called by mudg_gemm (gemm.hip) once the descriptor is validated and the large-tile path is selected.
bool mudg_gemm_fast_ok(const MudgGemmDesc& d);      // gemm.hip

int mudg_gemm256_dispatch(const MudgGemmDesc& d, int vflags, const h16* zpage, hipStream_t s) {
    if (d.stats) return 1;                   // declined: the 16-wave kernel does not write GroupNorm partials (see mudg_gemm)
    if (mudg_gemm_fast_ok(d)) {
        if (d.mode == 0) return launch256<0, true>(d, vflags, zpage, s);
        if (d.mode == 1) return launch256<1, true>(d, vflags, zpage, s);
        return launch256<2, true>(d, vflags, zpage, s);
    }
    if (d.mode == 0) return launch256<0, false>(d, vflags, zpage, s);
    if (d.mode == 1) return launch256<1, false>(d, vflags, zpage, s);
    return launch256<2, false>(d, vflags, zpage, s);
}
