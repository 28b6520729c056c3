// attention.hip — softmax(scale * Q K^T) V for head dim 64 on gfx950.
//
// mudg_attention: flash-style (online softmax, score matrix never leaves registers).
//   Workgroup = 4 waves = 128 query rows; each wave owns 32 query rows for the whole key loop.
//   Scores are produced TRANSPOSED: S^T = K Q^T via v_mfma_f32_32x32x16_bf16 with A = K tile, B = Q fragments, so
//   lane (q = lane&31, half = lane>>5) holds 16 of the 32 keys of a sub-tile for ITS query row: row max / row sum
//   are in-lane reductions plus one lane<->lane^32 exchange.  O^T = V^T P^T is accumulated the same way
//   (A = V^T tile rows = head-dim, B = P of the lane's own keys), which needs no cross-lane movement of P at all:
//   the contraction index of the second MFMA is simply enumerated in the key order the first MFMA left in the
//   registers (keys {0-3, 8-11} / {4-7, 12-15} per half), and the V^T fragment is gathered with two 8-byte LDS
//   reads in that same order.  V arrives already transposed (the projection GEMM writes V^T), so both LDS tiles
//   are filled with plain 16-byte row copies.
//   K/V tiles (64 keys) are staged global -> registers -> LDS with two LDS buffers and one barrier per tile.
//   (Tried: running the K stream one tile ahead so the score MFMAs of tile t+1 could overlap the softmax of tile t —
//   763 -> 634 TFLOP/s at N = 9216: +26 VGPRs and 32 extra register moves per tile, no interleave by hipcc.
//   Also measured, same box, N = 9216, all within +-0.5 % of the baseline or worse: row sums through the matrix pipe
//   (ones x P, -3.5 %), a constant zero accumulator instead of the per-tile zero fill, v_max3_f32 for the row max, and an
//   8-byte stagger of V^T rows 16-31 that removes the 2-way bank conflict of the ds_read_b64 fragment reads: the kernel
//   is bound by the S -> softmax -> PV dependency chain of each wave, not by VALU or LDS issue.  Splitting a staged tile
//   into two 32-key online-softmax steps (so the second half's score MFMAs could run under the first half's softmax)
//   measured -3 %: hipcc keeps the MFMA and VALU groups apart.)
//   Workgroups are numbered so that the query tiles of one (frame, head) run on one XCD and share its L2.
//
// mudg_temporal_attention: T <= 32 keys per pixel — a bandwidth problem.  One wave per (pixel, head), fp32 VALU
//   dot products with K/V of that pixel in LDS; no MFMA.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int QB = 128;     // query rows per workgroup
constexpr int KB = 64;      // keys per tile
constexpr int ALD = 72;     // LDS row stride in h16 (64 + 8 pad -> 144 B)
constexpr int ATILE = 64 * ALD;

// TWO: a second key / value set (p.K2 / p.Vt2: the image tokens) with its own softmax follows the first; the two
// normalised results are summed in registers and stored once.
template <bool TWO>
__global__ __launch_bounds__(256, 2) void attn_kernel(const MudgAttnDesc p, const int nqt, const int total) {
    __shared__ __attribute__((aligned(16))) h16 Ks[2 * ATILE];
    __shared__ __attribute__((aligned(16))) h16 Vs[2 * ATILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;

    // XCD-aware numbering: hardware places block b on XCD b % 8; give each XCD a contiguous range of work items.
    int w;
    {
        const int q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pair = w / nqt, qt = w - pair * nqt;
    const int f = pair / p.heads, h = pair - f * p.heads;
    const h16* Qp = reinterpret_cast<const h16*>(p.Q) + (int64_t)f * p.Nq * p.ldq + h * 64;
    h16* Op = reinterpret_cast<h16*>(p.O) + (int64_t)f * p.Nq * p.ldo + h * 64;
    // the key / value set being walked (set 0, then set 1 when TWO)
    const h16* Kp = reinterpret_cast<const h16*>(p.K) + (int64_t)(f / p.kv_div) * p.Nk * p.ldk + h * 64;
    const h16* Vp = reinterpret_cast<const h16*>(p.Vt) + (int64_t)(f / p.kv_div) * p.svt + (int64_t)(h * 64) * p.ldvt;
    int Nk = p.Nk, ldk = p.ldk, ldvt = p.ldvt;

    const int q = qt * QB + wave * 32 + l31;
    const bool qok = q < p.Nq;

    h16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = as_h16x8(qok ? ld16(Qp + (int64_t)q * p.ldq + ks * 16 + hi * 8) : zero16());

    const int lrow = tid >> 3, kc = tid & 7;   // staging: rows lrow, lrow+32; 16-byte chunk kc
    u32x4 kr[2], vr[2];
    auto load_tiles = [&](int kt) {
        const int j0 = kt * KB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = lrow + 32 * i;
            const int j = j0 + row;                       // key index for the K tile row
            kr[i] = (j < Nk) ? ld16(Kp + (int64_t)j * ldk + kc * 8) : zero16();
            const int jc = j0 + kc * 8;                   // first key of this V^T chunk (row = head-dim index)
            u32x4 v = zero16();
            if (jc < Nk) {
                v = ld16(Vp + (int64_t)row * ldvt + jc);
                if (jc + 8 > Nk) {                       // ragged tail: keys >= Nk must contribute exactly 0
                    h16x8 hv = as_h16x8(v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (jc + e >= Nk) hv[e] = (h16)0.f;
                    v = as_u32x4(hv);
                }
            }
            vr[i] = v;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            st16(&Ks[buf * ATILE + (lrow + 32 * i) * ALD + kc * 8], kr[i]);
            st16(&Vs[buf * ATILE + (lrow + 32 * i) * ALD + kc * 8], vr[i]);
        }
    };

    f32x16 o[2], res[TWO ? 2 : 1];
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;   // scores are exponentiated in base 2
    float inv = 0.f;
#pragma unroll
    for (int set = 0; set < (TWO ? 2 : 1); ++set) {
    if (TWO && set == 1) {
        Kp = reinterpret_cast<const h16*>(p.K2) + (int64_t)(f / p.kv_div2) * p.Nk2 * p.ldk2 + h * 64;
        Vp = reinterpret_cast<const h16*>(p.Vt2) + (int64_t)(f / p.kv_div2) * p.svt2 + (int64_t)(h * 64) * p.ldvt2;
        Nk = p.Nk2; ldk = p.ldk2; ldvt = p.ldvt2;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (Nk + KB - 1) / KB;
    load_tiles(0);
    stage(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) load_tiles(kt + 1);

        // ---- S^T = K Q^T for the two 32-key sub-tiles
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
            const h16* kp = Ks + cur * ATILE + (sub * 32 + l31) * ALD + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(kp + ks * 16);
                s[sub] = MFMA_32x32x16(kf, qf[ks], s[sub]);
            }
        }
        if (kt * KB + KB > Nk) {   // ragged last tile
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = kt * KB + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (j >= Nk) s[sub][r] = -INFINITY;
                }
        }

        // ---- online softmax (per query row = per lane; halves exchange once)
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // The running maximum only moves during the first few tiles; when no row of this wave raised it, alpha is
        // exactly 1 and the 32-register rescale of O (and its exp) is skipped — bit-identical, not a threshold trick.
        const bool grew = !__all(mx <= m_run);
        const float m_new = grew ? fmaxf(m_run, mx) : m_run;
        const float alpha = grew ? __builtin_amdgcn_exp2f((m_run - m_new) * c) : 1.0f;
        const float mc = m_new * c;
        m_run = m_new;
        float ps = 0.f;
        h16x8 pk[2][2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[sub][r], c, -mc));   // explicit: contraction is off globally
                ps += e;
                pk[sub][r >> 3][r & 7] = (h16)e;
            }
        l_run = l_run * alpha + ps;
        if (grew) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }

        // ---- O^T += V^T P^T ; contraction slots follow the key order the score MFMA left in registers
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const h16* vp = Vs + cur * ATILE + (dt * 32 + l31) * ALD + 4 * hi;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int kk = sub * 32 + jj * 16;
                    const h16x4 lo = *reinterpret_cast<const h16x4*>(vp + kk);
                    const h16x4 up = *reinterpret_cast<const h16x4*>(vp + kk + 8);
                    h16x8 vf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = up[e]; }
                    o[dt] = MFMA_32x32x16(vf, pk[sub][jj], o[dt]);
                }
        }

        if (more) stage(cur ^ 1);
        __syncthreads();
    }

    // ---- normalise this set's result
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    inv = 1.f / l_tot;
    if (!TWO && p.Lse && qok && hi == 0)               // log2-sum-exp of the scaled scores: P = 2^(c s - L) (mudg_attention_bwd)
        p.Lse[((int64_t)f * p.Nq + q) * p.heads + h] = m_run * c + __log2f(l_tot);
    if (TWO) {
        if (set == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { res[0][r] = o[0][r] * inv; res[1][r] = o[1][r] * inv; }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] = fmaf(o[0][r], inv, res[0][r]); o[1][r] = fmaf(o[1][r], inv, res[1][r]); }
            inv = 1.f;
        }
    }
    }   // sets
    // ---- store: lane holds, for its query row, head-dim columns dt*32 + 8g + 4*hi + {0..3}
    if (qok) {
        h16* orow = Op + (int64_t)q * p.ldo;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16* dst = orow + dt * 32 + 8 * g + 4 * hi;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[dt][4 * g + j] * inv;
                if (p.accumulate) {
                    Pack8 old; old.u = *reinterpret_cast<const u32x2*>(dst);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)old.h[j];
                }
                Pack8 nw;
#pragma unroll
                for (int j = 0; j < 4; ++j) nw.h[j] = (h16)v[j];
                *reinterpret_cast<u32x2*>(dst) = nw.u;
            }
    }
}

// Short key sets — the text (77) + image (16) tokens of the cross-attention — with MANY query tiles: the whole K / V^T of a (frame, head)
// is three 64-key tiles, so a workgroup stages them ONCE and then walks `xq` query tiles of that (frame, head) with no barrier at all
// (round 5).  attn_kernel stages the same three tiles for every 128 queries — 24 KB of K / V^T per 32 KB of Q + O — and its workgroup
// lives for four dependent global round trips: 227 us for the level-0 launch (377 MB: 1.7 TB/s).  Per query row the arithmetic is
// attn_kernel's, operation for operation: the same bits.
#if MUDG_PLANES == 1
constexpr int XK_TILES = 3;                      // key tiles held: two of the first set (Nk <= 128), one of the second (Nk2 <= 64)
template <bool TWO>
__global__ __launch_bounds__(256, 2) void xattn_kernel(const MudgAttnDesc p, const int nqt, const int xq, const int nqc, const int total) {
    extern __shared__ __attribute__((aligned(16))) h16 xlds[];
    h16* Ks = xlds;
    h16* Vs = xlds + XK_TILES * ATILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int w;
    {
        const int q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pair = w / nqc, qc = w - pair * nqc;
    const int f = pair / p.heads, h = pair - f * p.heads;
    const h16* Qp = reinterpret_cast<const h16*>(p.Q) + (int64_t)f * p.Nq * p.ldq + h * 64;
    h16* Op = reinterpret_cast<h16*>(p.O) + (int64_t)f * p.Nq * p.ldo + h * 64;
    const int nt0 = (p.Nk + KB - 1) / KB;

    // ---- stage every key tile of both sets (rows lrow, lrow + 32 of a tile; 16-byte chunk kc), as attn_kernel's load_tiles / stage
    {
        const int lrow = tid >> 3, kc = tid & 7;
#pragma unroll
        for (int set = 0; set < (TWO ? 2 : 1); ++set) {
            const h16* Kp = set == 0 ? reinterpret_cast<const h16*>(p.K) + (int64_t)(f / p.kv_div) * p.Nk * p.ldk + h * 64
                                     : reinterpret_cast<const h16*>(p.K2) + (int64_t)(f / p.kv_div2) * p.Nk2 * p.ldk2 + h * 64;
            const h16* Vp = set == 0 ? reinterpret_cast<const h16*>(p.Vt) + (int64_t)(f / p.kv_div) * p.svt + (int64_t)(h * 64) * p.ldvt
                                     : reinterpret_cast<const h16*>(p.Vt2) + (int64_t)(f / p.kv_div2) * p.svt2 + (int64_t)(h * 64) * p.ldvt2;
            const int Nk = set == 0 ? p.Nk : p.Nk2, ldk = set == 0 ? p.ldk : p.ldk2, ldvt = set == 0 ? p.ldvt : p.ldvt2;
            const int nkt = (Nk + KB - 1) / KB;
            for (int kt = 0; kt < nkt; ++kt) {
                const int ti = set == 0 ? kt : nt0 + kt, j0 = kt * KB;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = lrow + 32 * i, j = j0 + row;
                    const u32x4 kr = (j < Nk) ? ld16(Kp + (int64_t)j * ldk + kc * 8) : zero16();
                    const int jc = j0 + kc * 8;
                    u32x4 v = zero16();
                    if (jc < Nk) {
                        v = ld16(Vp + (int64_t)row * ldvt + jc);
                        if (jc + 8 > Nk) {
                            h16x8 hv = as_h16x8(v);
#pragma unroll
                            for (int e = 0; e < 8; ++e) if (jc + e >= Nk) hv[e] = (h16)0.f;
                            v = as_u32x4(hv);
                        }
                    }
                    st16(&Ks[ti * ATILE + row * ALD + kc * 8], kr);
                    st16(&Vs[ti * ATILE + row * ALD + kc * 8], v);
                }
            }
        }
    }
    __syncthreads();

    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;
    for (int qi = 0; qi < xq; ++qi) {
        const int qt = qc * xq + qi;
        if (qt >= nqt) break;
        const int q = qt * QB + wave * 32 + l31;
        const bool qok = q < p.Nq;
        h16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = as_h16x8(qok ? ld16(Qp + (int64_t)q * p.ldq + ks * 16 + hi * 8) : zero16());
        f32x16 o[2], res[TWO ? 2 : 1];
        float inv = 0.f;
#pragma unroll
        for (int set = 0; set < (TWO ? 2 : 1); ++set) {
            const int Nk = set == 0 ? p.Nk : p.Nk2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
            float m_run = -INFINITY, l_run = 0.f;
            const int nkt = (Nk + KB - 1) / KB;
            for (int kt = 0; kt < nkt; ++kt) {
                const int ti = set == 0 ? kt : nt0 + kt;
                f32x16 sc[2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[sub][r] = 0.f;
                    const h16* kp = Ks + ti * ATILE + (sub * 32 + l31) * ALD + hi * 8;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const h16x8 kf = *reinterpret_cast<const h16x8*>(kp + ks * 16);
                        sc[sub] = MFMA_32x32x16(kf, qf[ks], sc[sub]);
                    }
                }
                if (kt * KB + KB > Nk) {
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int j = kt * KB + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            if (j >= Nk) sc[sub][r] = -INFINITY;
                        }
                }
                float mx = sc[0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const bool grew = !__all(mx <= m_run);
                const float m_new = grew ? fmaxf(m_run, mx) : m_run;
                const float alpha = grew ? __builtin_amdgcn_exp2f((m_run - m_new) * c) : 1.0f;
                const float mc = m_new * c;
                m_run = m_new;
                float ps = 0.f;
                h16x8 pk[2][2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(sc[sub][r], c, -mc));
                        ps += e;
                        pk[sub][r >> 3][r & 7] = (h16)e;
                    }
                l_run = l_run * alpha + ps;
                if (grew) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const h16* vp = Vs + ti * ATILE + (dt * 32 + l31) * ALD + 4 * hi;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int kk = sub * 32 + jj * 16;
                            const h16x4 lo = *reinterpret_cast<const h16x4*>(vp + kk);
                            const h16x4 up = *reinterpret_cast<const h16x4*>(vp + kk + 8);
                            h16x8 vf;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = up[e]; }
                            o[dt] = MFMA_32x32x16(vf, pk[sub][jj], o[dt]);
                        }
                }
            }
            const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
            inv = 1.f / l_tot;
            if (!TWO && p.Lse && qok && hi == 0)
                p.Lse[((int64_t)f * p.Nq + q) * p.heads + h] = m_run * c + __log2f(l_tot);
            if (TWO) {
                if (set == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { res[0][r] = o[0][r] * inv; res[1][r] = o[1][r] * inv; }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[0][r] = fmaf(o[0][r], inv, res[0][r]); o[1][r] = fmaf(o[1][r], inv, res[1][r]); }
                    inv = 1.f;
                }
            }
        }
        if (qok) {
            h16* orow = Op + (int64_t)q * p.ldo;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h16* dst = orow + dt * 32 + 8 * g + 4 * hi;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = o[dt][4 * g + j] * inv;
                    if (p.accumulate) {
                        Pack8 old; old.u = *reinterpret_cast<const u32x2*>(dst);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float)old.h[j];
                    }
                    Pack8 nw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) nw.h[j] = (h16)v[j];
                    *reinterpret_cast<u32x2*>(dst) = nw.u;
                }
        }
    }
}
#endif

// Variant with 64 query rows per wave (two 32-row blocks): every K / V^T fragment read from LDS feeds two MFMAs instead
// of one — half the LDS traffic per FLOP — and the two blocks' softmax chains are independent.  Workgroup = 256 queries.
__global__ __launch_bounds__(256, 2) void attn64q_kernel(const MudgAttnDesc p, const int nqt, const int total) {
    __shared__ __attribute__((aligned(16))) h16 Ks[2 * ATILE];
    __shared__ __attribute__((aligned(16))) h16 Vs[2 * ATILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int w;
    {
        const int q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pair = w / nqt, qt = w - pair * nqt;
    const int f = pair / p.heads, h = pair - f * p.heads;
    const int kvb = f / p.kv_div;

    const h16* Qp = reinterpret_cast<const h16*>(p.Q) + (int64_t)f * p.Nq * p.ldq + h * 64;
    const h16* Kp = reinterpret_cast<const h16*>(p.K) + (int64_t)kvb * p.Nk * p.ldk + h * 64;
    const h16* Vp = reinterpret_cast<const h16*>(p.Vt) + (int64_t)kvb * p.svt + (int64_t)(h * 64) * p.ldvt;
    h16* Op = reinterpret_cast<h16*>(p.O) + (int64_t)f * p.Nq * p.ldo + h * 64;

    int qrow[2];
    bool qok[2];
    h16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = qt * 256 + wave * 64 + qb * 32 + l31;
        qok[qb] = qrow[qb] < p.Nq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qb][ks] = as_h16x8(qok[qb] ? ld16(Qp + (int64_t)qrow[qb] * p.ldq + ks * 16 + hi * 8) : zero16());
    }

    const int lrow = tid >> 3, kc = tid & 7;
    u32x4 kr[2], vr[2];
    auto load_tiles = [&](int kt) {
        const int j0 = kt * KB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = lrow + 32 * i;
            const int j = j0 + row;
            kr[i] = (j < p.Nk) ? ld16(Kp + (int64_t)j * p.ldk + kc * 8) : zero16();
            const int jc = j0 + kc * 8;
            u32x4 v = zero16();
            if (jc < p.Nk) {
                v = ld16(Vp + (int64_t)row * p.ldvt + jc);
                if (jc + 8 > p.Nk) {
                    h16x8 hv = as_h16x8(v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (jc + e >= p.Nk) hv[e] = (h16)0.f;
                    v = as_u32x4(hv);
                }
            }
            vr[i] = v;
        }
    };
    // V^T columns are stored permuted inside each group of 16 keys — [0-3, 8-11, 4-7, 12-15] — which is the order the score
    // MFMA leaves a lane's keys in: a lane's 8 contraction slots of a PV MFMA are then one 16-byte LDS read, not two of 8.
    const int vlo = (kc >> 1) * 16 + ((kc & 1) ? 4 : 0), vhi = (kc >> 1) * 16 + ((kc & 1) ? 12 : 8);
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            st16(&Ks[buf * ATILE + (lrow + 32 * i) * ALD + kc * 8], kr[i]);
            h16* vrow = &Vs[buf * ATILE + (lrow + 32 * i) * ALD];
            *reinterpret_cast<u32x2*>(vrow + vlo) = u32x2{vr[i][0], vr[i][1]};
            *reinterpret_cast<u32x2*>(vrow + vhi) = u32x2{vr[i][2], vr[i][3]};
        }
    };

    f32x16 o[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[qb][0][r] = 0.f; o[qb][1][r] = 0.f; }
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    const int nkt = (p.Nk + KB - 1) / KB;
    load_tiles(0);
    stage(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) load_tiles(kt + 1);

        f32x16 s[2][2];        // [qb][sub]
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][sub][r] = 0.f;
            const h16* kp = Ks + cur * ATILE + (sub * 32 + l31) * ALD + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(kp + ks * 16);
                s[0][sub] = MFMA_32x32x16(kf, qf[0][ks], s[0][sub]);
                s[1][sub] = MFMA_32x32x16(kf, qf[1][ks], s[1][sub]);
            }
        }
        if (kt * KB + KB > p.Nk) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = kt * KB + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (j >= p.Nk) { s[0][sub][r] = -INFINITY; s[1][sub][r] = -INFINITY; }
                }
        }

        h16x8 pk[2][2][2];     // [qb][sub][jj]
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = s[qb][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][1][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const bool grew = !__all(mx <= m_run[qb]);
            const float m_new = grew ? fmaxf(m_run[qb], mx) : m_run[qb];
            const float alpha = grew ? __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c) : 1.0f;
            const float mc = m_new * c;
            m_run[qb] = m_new;
            float ps = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[qb][sub][r], c, -mc));
                    ps += e;
                    pk[qb][sub][r >> 3][r & 7] = (h16)e;
                }
            l_run[qb] = l_run[qb] * alpha + ps;
            if (grew) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[qb][0][r] *= alpha; o[qb][1][r] *= alpha; }
            }
        }

#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const h16* vp = Vs + cur * ATILE + (dt * 32 + l31) * ALD + 8 * hi;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const h16x8 vf = *reinterpret_cast<const h16x8*>(vp + sub * 32 + jj * 16);
                    o[0][dt] = MFMA_32x32x16(vf, pk[0][sub][jj], o[0][dt]);
                    o[1][dt] = MFMA_32x32x16(vf, pk[1][sub][jj], o[1][dt]);
                }
        }

        if (more) stage(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.f / l_tot;
        if (p.Lse && qok[qb] && hi == 0) p.Lse[((int64_t)f * p.Nq + qrow[qb]) * p.heads + h] = m_run[qb] * c + __log2f(l_tot);
        if (qok[qb]) {
            h16* orow = Op + (int64_t)qrow[qb] * p.ldo;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h16* dst = orow + dt * 32 + 8 * g + 4 * hi;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = o[qb][dt][4 * g + j] * inv;
                    if (p.accumulate) {
                        Pack8 old; old.u = *reinterpret_cast<const u32x2*>(dst);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float)old.h[j];
                    }
                    Pack8 nw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) nw.h[j] = (h16)v[j];
                    *reinterpret_cast<u32x2*>(dst) = nw.u;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ LDS-DMA variant
// attn64q_kernel with the K / V^T tiles staged by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write
// pass, 16 fewer live registers) and, on top of that, the lean softmax (template LEAN; in the register-staged kernel it
// spilled).  Needs Nk % 64 == 0.
// LEAN (p.q_prescaled: Q carries scale * log2 e, so S = K Q^T is the base-2 exponent): the softmax chain — at head width
// 64 the kernel is bound by VALU issue, not by MFMA — is cut from {max, fma, exp, add, cvt} to {exp, add, cvt} per score:
// a reference maximum m_ref per query row (the row maximum over the first key tile, from one extra set of score MFMAs per
// block) enters as the INITIAL VALUE of the score accumulators, so S - m_ref comes straight out of the MFMA; the per-tile
// maximum is not computed and O / l are never rescaled.  Should a row's tile sum exceed LEAN_LIMIT (2^40 — a later score 40
// binary orders above the first tile's maximum — with bf16 operands, 2^15 with fp16 ones) the workgroup redoes its block with the classic online softmax — exact, merely
// slower, and not observed on real data.
//  * A DMA instruction writes 1 KiB lane-linear (8 rows x 128 B), so the tiles are unpadded [64][64] h16 with the XOR
//    swizzle of the GEMM kernel applied on the SOURCE side (which 16-byte chunk of the row a lane fetches) and mirrored on
//    the fragment reads: chunk c of row r lives in slot c ^ ((r >> 1) & 7).
//  * The score MFMA leaves a lane's keys in the order {0-3, 8-11 | 4-7, 12-15} per 16; instead of permuting V^T's columns
//    (8-byte granules — impossible for a 16-byte DMA) the K tile's ROWS are permuted at the source by the involution that
//    swaps the two middle 4-blocks of every 16: softmax does not care in which order keys arrive, and a lane's 8
//    contraction slots of a PV MFMA are then 8 consecutive keys = one natural 16-byte chunk of V^T.
//  * Tile t + 1 is requested at the top of iteration t into the other buffer; the barrier that closes the iteration
//    (vmcnt(0) + s_barrier) is the only synchronisation.
constexpr int DTILE = 64 * 64;          // h16 per unpadded tile
// Largest per-lane tile sum of 2^(s - m_ref) the lean softmax accepts before the workgroup falls back to the classic loop.
// P is stored as h16: a bf16 P has fp32's exponent range (2^40 leaves room for the row sum), an IEEE-half P overflows to
// inf beyond 65504 — there the limit is 2^15, so that no single exponential can reach the h16 maximum undetected.
#ifdef MUDG_OPERAND_FP16
constexpr float LEAN_LIMIT = 32768.f;
#else
constexpr float LEAN_LIMIT = 1099511627776.f;      // 2^40
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t attn_rsrc(const h16* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(base), 0, (int)0x80000000u, 0x00020000);
}
typedef __attribute__((address_space(3))) void* attn_lptr_t;

// F8 (needs LEAN): the scores come from MX-fp8 copies of Q and K (MudgAttnDesc.Q8 ...): one
// v_mfma_scale_f32_32x32x64_f8f6f4 per (32 keys x 32 queries) block covers the whole head width at twice the bf16 MFMA
// rate; the K tile is 64 keys x 64 bytes, staged by DMA with a 4-slot XOR swizzle, its E8M0 scales come straight from L2
// one tile ahead.  Softmax and P V are the bf16 kernel's.  (Operand / scale layout of that MFMA: measured with
// tools/ubench/mxfp8_layout.hip.)
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <bool LEAN, bool F8>
__global__ __launch_bounds__(256, 2) void attn64d_kernel(const MudgAttnDesc p, const int nqt, const int total) {
    static_assert(LEAN || !F8, "the fp8 score path builds on the lean softmax");
    __shared__ __attribute__((aligned(1024))) h16 Ks[F8 ? DTILE : 2 * DTILE];      // F8: 2 x (64 keys x 64 B) = one bf16 tile's worth
    __shared__ __attribute__((aligned(1024))) h16 Vs[2 * DTILE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int w;
    {
        const int q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pair = w / nqt, qt = w - pair * nqt;
    const int f = pair / p.heads, h = pair - f * p.heads;
    const int kvb = f / p.kv_div;

    const h16* Qp = reinterpret_cast<const h16*>(p.Q) + (int64_t)f * p.Nq * p.ldq + h * 64;
    const h16* Kp = reinterpret_cast<const h16*>(p.K) + (int64_t)kvb * p.Nk * p.ldk + h * 64;
    const h16* Vp = reinterpret_cast<const h16*>(p.Vt) + (int64_t)kvb * p.svt + (int64_t)(h * 64) * p.ldvt;
    h16* Op = reinterpret_cast<h16*>(p.O) + (int64_t)f * p.Nq * p.ldo + h * 64;

    int qrow[2];
    bool qok[2];
    h16x8 qf[2][F8 ? 1 : 4];
    i32x8 qf8[2];                 // F8: the lane's 32 fp8 dims [32 hi, 32 hi + 32) of its query row, and their scale
    int qsc[2] = {127, 127};
    const unsigned char* K8p = nullptr;
    const unsigned char* Ksp = nullptr;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = qt * 256 + wave * 64 + qb * 32 + l31;
        qok[qb] = qrow[qb] < p.Nq;
        if constexpr (F8) {
            // operand layout of the f8f6f4 MFMA (tools/ubench/mxfp8_layout.hip): a lane's first 16 bytes are K elements
            // 16 half + [0, 16) — scaled by the scale lanes 0-31 supply — its second 16 bytes 32 + 16 half + [0, 16), scaled by
            // lanes 32-63's: lane half `hi` therefore fetches dims [16 hi, +16) and [32 + 16 hi, +16), and carries the scale
            // of dims [32 hi, +32)
            const unsigned char* q8 = reinterpret_cast<const unsigned char*>(p.Q8) + ((int64_t)f * p.Nq + qrow[qb]) * p.ldq8 + h * 64 + hi * 16;
            const u32x4 a = qok[qb] ? ld16(q8) : zero16(), b = qok[qb] ? ld16(q8 + 32) : zero16();
            qf8[qb] = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
            if (qok[qb]) qsc[qb] = reinterpret_cast<const unsigned char*>(p.Qs)[((int64_t)f * p.Nq + qrow[qb]) * p.ldqs + h * 2 + hi];
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[qb][ks] = as_h16x8(qok[qb] ? ld16(Qp + (int64_t)qrow[qb] * p.ldq + ks * 16 + hi * 8) : zero16());
        }
    }
    if constexpr (F8) {
        K8p = reinterpret_cast<const unsigned char*>(p.K8) + (int64_t)kvb * p.Nk * p.ldk8 + h * 64;
        Ksp = reinterpret_cast<const unsigned char*>(p.Ks) + (int64_t)kvb * p.Nk * p.ldks + h * 2;
    }

    // DMA geometry: wave w stages rows [16w, 16w + 16) of both tiles, two 1-KiB instructions each; in instruction i lane l
    // lands in row 16w + 8i + (l >> 3), slot l & 7.  (F8 K tile: ONE instruction per wave, row 16w + (l >> 2), slot l & 3.)
    const __amdgpu_buffer_rsrc_t rK = F8 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(K8p), 0, (int)0x80000000u, 0x00020000)
                                         : attn_rsrc(Kp);
    const __amdgpu_buffer_rsrc_t rV = attn_rsrc(Vp);
    unsigned vk[2], vv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * wave + 8 * i + (lane >> 3), slot = lane & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int i16 = row & 15;
        const int key = (row & ~15) | (i16 & 3) | ((i16 & 8) >> 1) | ((i16 & 4) << 1);     // swap the middle 4-blocks
        vk[i] = (unsigned)key * (unsigned)p.ldk * 2u + (unsigned)chunk * 16u;
        vv[i] = (unsigned)row * (unsigned)p.ldvt * 2u + (unsigned)chunk * 16u;
    }
    int kscale_off[2] = {0, 0};      // F8: byte offsets (inside a tile) of this lane's K scales for sub-tiles 0 / 1
    if constexpr (F8) {
        const int row = 16 * wave + (lane >> 2), slot = lane & 3;
        const int chunk = slot ^ ((row >> 2) & 3);
        const int i16 = row & 15;
        const int key = (row & ~15) | (i16 & 3) | ((i16 & 8) >> 1) | ((i16 & 4) << 1);
        vk[0] = (unsigned)key * (unsigned)p.ldk8 + (unsigned)chunk * 16u;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int r = sub * 32 + l31, j16 = r & 15;
            const int k = (r & ~15) | (j16 & 3) | ((j16 & 8) >> 1) | ((j16 & 4) << 1);      // the key LDS row r holds
            kscale_off[sub] = k * p.ldks + hi;
        }
    }
    auto request = [&](int kt, int buf) {
        const int sv = kt * 128;
        if constexpr (F8) {
            unsigned char* ks8 = reinterpret_cast<unsigned char*>(Ks);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (attn_lptr_t)(ks8 + buf * 4096 + (16 * wave) * 64), 16, (int)vk[0], kt * 64 * p.ldk8, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (!F8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (attn_lptr_t)(Ks + buf * DTILE + (16 * wave + 8 * i) * 64), 16, (int)vk[i], kt * 64 * p.ldk * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (attn_lptr_t)(Vs + buf * DTILE + (16 * wave + 8 * i) * 64), 16, (int)vv[i], sv, 0, 0);
        }
    };
    const int sw = (l31 >> 1) & 7;          // read-side swizzle (row bases are multiples of 32)
    const int sw8 = (l31 >> 2) & 3;         // F8 K tile: 64-byte rows, 4 slots
    // one (32 keys x 32 queries) score block for both query blocks from K tile `buf`, sub-tile `sub`
    auto score_block = [&](int buf, int sub, f32x16& s0, f32x16& s1, int kscale) {
        if constexpr (F8) {
            const unsigned char* kp = reinterpret_cast<const unsigned char*>(Ks) + buf * 4096 + (sub * 32 + l31) * 64;
            const u32x4 a = ld16(kp + ((hi ^ sw8) << 4)), b = ld16(kp + (((2 + hi) ^ sw8) << 4));
            const i32x8 kf = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
            s0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf8[0], s0, 0, 0, 0, kscale, 0, qsc[0]);
            s1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf8[1], s1, 0, 0, 0, kscale, 0, qsc[1]);
        } else {
            const h16* kp = Ks + buf * DTILE + (sub * 32 + l31) * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(kp + (((ks * 2 + hi) ^ sw) << 3));
                s0 = MFMA_32x32x16(kf, qf[0][ks], s0);
                s1 = MFMA_32x32x16(kf, qf[1][ks], s1);
            }
        }
    };

    f32x16 o[2][2];
    float m_run[2], l_run[2];
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;
    const int nkt = p.Nk / KB;
    bool overflow = false;

    auto key_loop = [&](auto lean_tag) {
        constexpr bool LN = decltype(lean_tag)::value;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[qb][0][r] = 0.f; o[qb][1][r] = 0.f; }
            m_run[qb] = LN ? 0.f : -INFINITY;
            l_run[qb] = 0.f;
        }
        request(0, 0);
        __syncthreads();
        if constexpr (LN) {   // reference maxima = the first tile's row maxima (one extra set of score MFMAs per block)
            f32x16 s0[2][2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s0[qb][sub][r] = 0.f;
                score_block(0, sub, s0[0][sub], s0[1][sub], F8 ? (int)Ksp[kscale_off[sub]] : 127);
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float mx = s0[qb][0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[qb][0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s0[qb][1][r]);
                m_run[qb] = fmaxf(mx, __shfl_xor(mx, 32, 64));
            }
        }

        int ksc[2] = {127, 127};          // F8: this tile's K scales (requested one tile ahead)
        if constexpr (F8) { ksc[0] = Ksp[kscale_off[0]]; ksc[1] = Ksp[kscale_off[1]]; }
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) request(kt + 1, cur ^ 1);
            int ksn[2] = {127, 127};
            if constexpr (F8) {
                if (kt + 1 < nkt) {
                    ksn[0] = Ksp[(int64_t)(kt + 1) * 64 * p.ldks + kscale_off[0]];
                    ksn[1] = Ksp[(int64_t)(kt + 1) * 64 * p.ldks + kscale_off[1]];
                }
            }

            f32x16 s[2][2];        // [qb][sub]
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const float init = LN ? -m_run[qb] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[qb][sub][r] = init;
                }
                score_block(cur, sub, s[0][sub], s[1][sub], ksc[sub]);
            }
            if constexpr (F8) { ksc[0] = ksn[0]; ksc[1] = ksn[1]; }

            h16x8 pk[2][2][2];     // [qb][sub][jj]
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if constexpr (LN) {
                    float ps = 0.f;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float e = __builtin_amdgcn_exp2f(s[qb][sub][r]);
                            ps += e;
                            pk[qb][sub][r >> 3][r & 7] = (h16)e;
                        }
                    l_run[qb] += ps;
                    overflow = overflow || !(ps <= LEAN_LIMIT);             // true for inf / nan as well
                } else {
                    float mx = s[qb][0][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][0][r]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][1][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    const bool grew = !__all(mx <= m_run[qb]);
                    const float m_new = grew ? fmaxf(m_run[qb], mx) : m_run[qb];
                    const float alpha = grew ? __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c) : 1.0f;
                    const float mc = m_new * c;
                    m_run[qb] = m_new;
                    float ps = 0.f;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(s[qb][sub][r], c, -mc));
                            ps += e;
                            pk[qb][sub][r >> 3][r & 7] = (h16)e;
                        }
                    l_run[qb] = l_run[qb] * alpha + ps;
                    if (grew) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) { o[qb][0][r] *= alpha; o[qb][1][r] *= alpha; }
                    }
                }
            }

            // the K rows were permuted at the source: register group (sub, jj) of half hi holds keys 32 sub + 16 jj + 8 hi ..+7
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const h16* vp = Vs + cur * DTILE + (dt * 32 + l31) * 64;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const h16x8 vf = *reinterpret_cast<const h16x8*>(vp + (((4 * sub + 2 * jj + hi) ^ sw) << 3));
                        o[0][dt] = MFMA_32x32x16(vf, pk[0][sub][jj], o[0][dt]);
                        o[1][dt] = MFMA_32x32x16(vf, pk[1][sub][jj], o[1][dt]);
                    }
            }
            __syncthreads();         // tile kt + 1 has landed (vmcnt(0)) and everyone is done reading tile kt
        }
    };

    if constexpr (LEAN) {
        key_loop(std::true_type{});
        if (__syncthreads_or(overflow ? 1 : 0)) key_loop(std::false_type{});      // cold: exact, merely slower
    } else {
        key_loop(std::false_type{});
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.f / l_tot;
        if (p.Lse && qok[qb] && hi == 0) p.Lse[((int64_t)f * p.Nq + qrow[qb]) * p.heads + h] = m_run[qb] * c + __log2f(l_tot);
        if (qok[qb]) {
            h16* orow = Op + (int64_t)qrow[qb] * p.ldo;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h16* dst = orow + dt * 32 + 8 * g + 4 * hi;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = o[qb][dt][4 * g + j] * inv;
                    if (p.accumulate) {
                        Pack8 old; old.u = *reinterpret_cast<const u32x2*>(dst);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float)old.h[j];
                    }
                    Pack8 nw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) nw.h[j] = (h16)v[j];
                    *reinterpret_cast<u32x2*>(dst) = nw.u;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ temporal
// TP = padded sequence length (16 or 32); DP = 64 / TP lanes share one query row, each owning DW = 64 / DP dims.
// A wave walks TATTN_ITEMS consecutive (pixel, head) items and fetches the next item's q / k / v rows before it computes
// the current one: with one item per wave the ~3 loads per lane were issued, waited for and only then followed by ~800
// VALU instructions, so most resident waves were computing and too few bytes were in flight (3.1 TB/s).
// K and V of an item live in a per-wave LDS region: LDS operations of one wave execute in order, so no workgroup barrier.
constexpr int TATTN_ITEMS = 4;

template <int TP>
__global__ __launch_bounds__(256, TP == 16 ? 3 : 1) void tattn_kernel(const h16* __restrict__ QKV, h16* __restrict__ O,
                                                     int B, int T, int HW, int heads, int ldqkv, int ldo,
                                                     float scale, int total) {
    constexpr int DP = 64 / TP, DW = 64 / DP;
    __shared__ __attribute__((aligned(16))) h16 Ks[4][TP * 64];
    __shared__ __attribute__((aligned(16))) h16 Vs[4][TP * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tq = lane / DP, dp = lane % DP;
    const int C = heads * 64;
    const int w0 = (blockIdx.x * 4 + wave) * TATTN_ITEMS;

    u32x4 qn[DW / 8], kn[DW / 8], vn[DW / 8];        // the item being fetched
    int64_t rown = 0; int hn = 0; bool okn = false;
    auto fetch = [&](int w) {
        okn = w < total && tq < T;
        const int bp = w < total ? w / heads : 0;
        hn = w < total ? w - bp * heads : 0;
        const int b = bp / HW, px = bp - b * HW;
        rown = ((int64_t)(b * T + tq) * HW + px);
        const h16* src = QKV + rown * ldqkv + hn * 64 + dp * DW;
#pragma unroll
        for (int i = 0; i < DW / 8; ++i) {
            qn[i] = okn ? ld16(src + i * 8) : zero16();
            kn[i] = okn ? ld16(src + C + i * 8) : zero16();
            vn[i] = okn ? ld16(src + 2 * C + i * 8) : zero16();
        }
    };
    fetch(w0);

    for (int it = 0; it < TATTN_ITEMS; ++it) {
        if (w0 + it >= total) break;                 // wave-uniform
        // q stays packed: the score dot products run on v_dot2c_f32_{bf16,f16} (two exact products + fp32 accumulate
        // per instruction, no widening converts)
        h16x8 qv[DW / 8];
        const int64_t row = rown; const int h = hn; const bool rok = okn;
        __builtin_amdgcn_wave_barrier();             // the previous item's LDS reads are issued before these writes
#pragma unroll
        for (int i = 0; i < DW / 8; ++i) {
            qv[i] = as_h16x8(qn[i]);
            st16(&Ks[wave][tq * 64 + dp * DW + i * 8], kn[i]);
            st16(&Vs[wave][tq * 64 + dp * DW + i * 8], vn[i]);
        }
        __builtin_amdgcn_wave_barrier();
        if (it + 1 < TATTN_ITEMS) fetch(w0 + it + 1);

        float sc[TP];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < DW / 8; ++i) {
                const h16x8 kk = *reinterpret_cast<const h16x8*>(&Ks[wave][j * 64 + dp * DW + i * 8]);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const h16x2 qp = {qv[i][e], qv[i][e + 1]}, kp = {kk[e], kk[e + 1]};
                    a = DOT2_H16(qp, kp, a);
                }
            }
#pragma unroll
            for (int o = 1; o < DP; o <<= 1) a += __shfl_xor(a, o, 64);
            a = (j < T) ? a * scale : -INFINITY;
            sc[j] = a;
            mx = fmaxf(mx, a);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < TP; ++j) { sc[j] = __expf(sc[j] - mx); sum += sc[j]; }
        const float inv = 1.f / sum;

        float ov[DW];
#pragma unroll
        for (int d = 0; d < DW; ++d) ov[d] = 0.f;
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const float pj = sc[j] * inv;
#pragma unroll
            for (int i = 0; i < DW / 8; ++i) {
                const h16x8 vv = *reinterpret_cast<const h16x8*>(&Vs[wave][j * 64 + dp * DW + i * 8]);
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[i * 8 + e] = fmaf(pj, (float)vv[e], ov[i * 8 + e]);
            }
        }
        if (rok) {
            h16* dst = O + row * ldo + h * 64 + dp * DW;
#pragma unroll
            for (int i = 0; i < DW / 8; ++i) {
                h16x8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (h16)ov[i * 8 + e];
                st16(dst + i * 8, as_u32x4(t));
            }
        }
    }
}


#if MUDG_PLANES == 1
// The same problem on the matrix cores (T <= 16).  The VALU kernel above spends ~800 instructions per (pixel, head) item — at the
// UNet's level 0 that is ~120 us of pure VALU time for a pass whose 755 MB take ~135 us at the rate the norms reach — so it was
// compute-bound on the wrong unit.  Here an item is 2 + 4 MFMAs:
//   S^T = K Q^T   two v_mfma_f32_16x16x32: lane (c = l % 16, g = l / 16) feeds row c of K as A and row c of Q as B, dims 8g + 32s ..
//                 + 7; the result lands as S^T[tk = 4g + i][tq = c], i = 0..3, so the softmax over tk is 4 in-lane values and two
//                 cross-lane steps (xor 16, xor 32).  The fragments could come straight from global memory (one 16-byte load per lane),
//                 but then adjacent lanes ask for different frames' rows, 17 MB apart: q, k, v are fetched eight lanes per row — a
//                 request is whole 128-byte head slices — and pass through a per-wave LDS tile (chunk-swizzled, 16-byte reads);
//   O^T = V^T P^T 2 x four v_mfma_f32_16x16x16 (k = tk): P^T is already the B fragment (lane (c, g) holds tk = 4g + i of column tq = c);
//                 V^T[d][tk = 4g + j] is the one transposed read — V goes through a per-wave 2-KiB LDS tile (8-byte writes, 2-byte
//                 reads, the 8-byte units of a row XOR-swizzled by row / 4 so the 64 lanes of a read meet 32 distinct banks).  Row
//                 r of output block m stands for dim 32 (m / 2) + 8 (r / 4) + 4 (m % 2) + r % 4, so lane (c, g) ends up with dims
//                 8g .. 8g + 7 and 32 + 8g .. 32 + 8g + 7 of query c: two 16-byte stores, each instruction 64 contiguous bytes per row.
// P enters the second contraction as two 16-bit pieces (eight MFMAs instead of four): fp32-class probabilities, as in the VALU kernel.
#ifdef MUDG_OPERAND_FP16
typedef __attribute__((ext_vector_type(4))) _Float16 mf4;
#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0)
#else
typedef __attribute__((ext_vector_type(4))) short mf4;
#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)
#endif

__global__ __launch_bounds__(256) void tattn_mfma_kernel(const h16* __restrict__ QKV, h16* __restrict__ O,
                                                          int B, int T, int HW, int heads, int ldqkv, int ldo,
                                                          float scale, int total) {
    // per wave: Q, K, V tiles of one item, 16 rows x 128 B each
    __shared__ __attribute__((aligned(16))) h16 Ts[4][3][16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;          // MFMA coordinates: row / column c, k-group g
    const int lr = lane >> 3, lj = lane & 7;         // fetch coordinates: rows lr and lr + 8, 16-byte chunk lj — a request covers
    const int C = heads * 64;                        //   whole 128-byte head slices, eight lanes per row
    const int w0 = (blockIdx.x * 4 + wave) * TATTN_ITEMS;
    char* qs = reinterpret_cast<char*>(&Ts[wave][0][0]);
    char* ks = reinterpret_cast<char*>(&Ts[wave][1][0]);
    char* vs = reinterpret_cast<char*>(&Ts[wave][2][0]);
    auto fz = [](int rg) { return (rg & 1) | ((rg >> 1) << 3); };     // V's swizzle of row group rg: bits 0 and 3 of the 8-byte unit index

    u32x4 qn[2], kn[2], vn[2];                       // the item being fetched: [row half]
    int64_t pixn = 0; int hn = 0; bool itemn = false;
    auto fetch = [&](int w) {
        itemn = w < total;
        const int bp = itemn ? w / heads : 0;
        hn = itemn ? w - bp * heads : 0;
        const int b = bp / HW, px = bp - b * HW;
        pixn = (int64_t)b * T * HW + px;             // row of frame t: pixn + t * HW
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int t = lr + 8 * s;
            const bool ok = itemn && t < T;
            const h16* src = QKV + (pixn + (int64_t)(ok ? t : 0) * HW) * ldqkv + hn * 64 + lj * 8;
            qn[s] = ok ? ld16(src) : zero16();
            kn[s] = ok ? ld16(src + C) : zero16();
            vn[s] = ok ? ld16(src + 2 * C) : zero16();
        }
    };
    fetch(w0);

    for (int it = 0; it < TATTN_ITEMS; ++it) {
        if (w0 + it >= total) break;                 // wave-uniform
        const int64_t pix = pixn; const int h = hn;
        __builtin_amdgcn_wave_barrier();             // the previous item's LDS reads are issued before these writes
        // Q / K: chunk j of row r at chunk j ^ (r & 7) (fragment reads of one chunk column by 16 rows: two rows per bank group);
        // V: the two 8-byte units of chunk j at units (2j, 2j + 1) ^ fz(r / 4) (2-byte transposed reads, see the header)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int r = lr + 8 * s;
            st16(qs + r * 128 + ((lj ^ (r & 7)) << 4), qn[s]);
            st16(ks + r * 128 + ((lj ^ (r & 7)) << 4), kn[s]);
            const int z = fz(r >> 2);
            u32x2 lo = {vn[s][0], vn[s][1]}, hi = {vn[s][2], vn[s][3]};
            *reinterpret_cast<u32x2*>(vs + r * 128 + (((2 * lj) ^ z) << 3)) = lo;
            *reinterpret_cast<u32x2*>(vs + r * 128 + (((2 * lj + 1) ^ z) << 3)) = hi;
        }
        __builtin_amdgcn_wave_barrier();
        if (it + 1 < TATTN_ITEMS) fetch(w0 + it + 1);

        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {                // dims 8g + 32s .. + 7 = chunk g + 4s of row c
            const int off = c * 128 + (((g + 4 * s) ^ (c & 7)) << 4);
            st = MFMA_16x16x32(as_h16x8(ld16(ks + off)), as_h16x8(ld16(qs + off)), st);
        }

        // softmax over tk = 4g + i: in-lane over i, then across the four lane groups
        float sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[i] = (4 * g + i < T) ? st[i] * scale : -INFINITY; mx = fmaxf(mx, sc[i]); }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // the probabilities enter the second contraction as TWO 16-bit pieces (p = p0 + p1, four more MFMAs on a kernel that waits
        // for memory): the fp32-probability arithmetic of the VALU kernel it replaces, not one more 16-bit rounding per layer
        float sum = 0.f;
        union { mf4 m; h16 e[4]; } pt, pr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pv = __expf(sc[i] - mx);
            sum += pv;
            pt.e[i] = (h16)pv;
            pr.e[i] = (h16)(pv - (float)pt.e[i]);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;

        float ov[16];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            union { mf4 m4; unsigned short e[4]; } vt;
            // row c of block m stands for dim 32 (m / 2) + 8 (c / 4) + 4 (m % 2) + c % 4 = unit 8 (m / 2) + 2 (c / 4) + m % 2, element c % 4
            const int u = (8 * (m >> 1) + 2 * (c >> 2) + (m & 1)) ^ fz(g);    // row 4g + j has r / 4 = g
#pragma unroll
            for (int j = 0; j < 4; ++j)
                vt.e[j] = *reinterpret_cast<const unsigned short*>(vs + (4 * g + j) * 128 + (u << 3) + 2 * (c & 3));
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = MFMA_16x16x16(vt.m4, pr.m, acc);
            acc = MFMA_16x16x16(vt.m4, pt.m, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) ov[4 * m + i] = acc[i] * inv;
        }
        if (c < T) {
            // ov[4m + i] is dim 32 (m / 2) + 8g + 4 (m % 2) + i: blocks 0, 1 are dims 8g .. 8g + 7, blocks 2, 3 the same + 32 — a store
            // instruction covers 64 contiguous bytes per query row
            h16* dst = O + (pix + (int64_t)c * HW) * ldo + h * 64 + g * 8;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                h16x8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (h16)ov[half * 8 + e];
                st16(dst + half * 32, as_u32x4(t));
            }
        }
    }
}
#endif

#if MUDG_PLANES > 1
// ------------------------------------------------------------------------------------------------ split-operand builds
// attn_kernel with every operand carried as PLANES bf16 pieces (common.h): Q / K / V^T pieces come from planes of the
// input matrices, the probabilities are split in registers, and both contractions accumulate the kept (piece, piece)
// partial products — NSEG MFMAs where the 16-bit kernel issues one.  Exponentials and the output normalisation are the
// same fp32 arithmetic.  One K / V^T tile per piece in LDS; precision first: 32 queries per wave, no 64-query variant.
template <bool TWO>
__global__ __launch_bounds__(256, PLANES == 2 ? 2 : 1) void attn_split_kernel(const MudgAttnDesc p, const int nqt, const int total) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    h16* Ks = reinterpret_cast<h16*>(smem_raw);                   // [2 buffers][PLANES][ATILE]
    h16* Vs = Ks + 2 * PLANES * ATILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int w;
    {
        const int q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pair = w / nqt, qt = w - pair * nqt;
    const int f = pair / p.heads, h = pair - f * p.heads;
    const int psq = p.ldq / PLANES, pso = p.ldo / PLANES;
    int psk = p.ldk / PLANES, psv = p.ldvt / PLANES;

    const h16* Qp = reinterpret_cast<const h16*>(p.Q) + (int64_t)f * p.Nq * p.ldq + h * 64;
    h16* Op = reinterpret_cast<h16*>(p.O) + (int64_t)f * p.Nq * p.ldo + h * 64;
    const h16* Kp = reinterpret_cast<const h16*>(p.K) + (int64_t)(f / p.kv_div) * p.Nk * p.ldk + h * 64;
    const h16* Vp = reinterpret_cast<const h16*>(p.Vt) + (int64_t)(f / p.kv_div) * p.svt + (int64_t)(h * 64) * p.ldvt;
    int Nk = p.Nk, ldk = p.ldk, ldvt = p.ldvt;

    const int q = qt * QB + wave * 32 + l31;
    const bool qok = q < p.Nq;
    h16x8 qf[PLANES][4];
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[pl][ks] = as_h16x8(qok ? ld16(Qp + (int64_t)q * p.ldq + pl * psq + ks * 16 + hi * 8) : zero16());

    const int lrow = tid >> 3, kc = tid & 7;
    u32x4 kr[PLANES][2], vr[PLANES][2];
    auto load_tiles = [&](int kt) {
        const int j0 = kt * KB;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = lrow + 32 * i;
                const int j = j0 + row;
                kr[pl][i] = (j < Nk) ? ld16(Kp + (int64_t)j * ldk + pl * psk + kc * 8) : zero16();
                const int jc = j0 + kc * 8;
                u32x4 v = zero16();
                if (jc < Nk) {
                    v = ld16(Vp + (int64_t)row * ldvt + pl * psv + jc);
                    if (jc + 8 > Nk) {
                        h16x8 hv = as_h16x8(v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (jc + e >= Nk) hv[e] = (h16)0.f;
                        v = as_u32x4(hv);
                    }
                }
                vr[pl][i] = v;
            }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                st16(&Ks[(buf * PLANES + pl) * ATILE + (lrow + 32 * i) * ALD + kc * 8], kr[pl][i]);
                st16(&Vs[(buf * PLANES + pl) * ATILE + (lrow + 32 * i) * ALD + kc * 8], vr[pl][i]);
            }
    };

    f32x16 o[2], res[TWO ? 2 : 1];
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;
#pragma unroll
    for (int set = 0; set < (TWO ? 2 : 1); ++set) {
    if (TWO && set == 1) {
        Kp = reinterpret_cast<const h16*>(p.K2) + (int64_t)(f / p.kv_div2) * p.Nk2 * p.ldk2 + h * 64;
        Vp = reinterpret_cast<const h16*>(p.Vt2) + (int64_t)(f / p.kv_div2) * p.svt2 + (int64_t)(h * 64) * p.ldvt2;
        Nk = p.Nk2; ldk = p.ldk2; ldvt = p.ldvt2; psk = ldk / PLANES; psv = ldvt / PLANES;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (Nk + KB - 1) / KB;
    load_tiles(0);
    stage(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) load_tiles(kt + 1);

        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                const h16* kp = Ks + (cur * PLANES + seg_xp(sg)) * ATILE + (sub * 32 + l31) * ALD + hi * 8;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const h16x8 kf = *reinterpret_cast<const h16x8*>(kp + ks * 16);
                    s[sub] = MFMA_32x32x16(kf, qf[seg_wp(sg)][ks], s[sub]);
                }
            }
        }
        if (kt * KB + KB > Nk) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = kt * KB + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (j >= Nk) s[sub][r] = -INFINITY;
                }
        }

        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // The softmax of this kernel was its bottleneck (rocprofv3: twice as many VALU issue cycles as MFMA cycles per key tile): the
        // exponentials run on v_exp_f32 directly (1 ulp — fp32-class, as the split products around them) instead of the library
        // exp2f with its range handling, and O is rescaled only when some row's maximum actually grew (wave-uniform test).
        const bool grew = !__all(mx <= m_run);
        const float m_new = grew ? fmaxf(m_run, mx) : m_run;
        const float alpha = grew ? __builtin_amdgcn_exp2f((m_run - m_new) * c) : 1.0f;
        const float mc = m_new * c;
        m_run = m_new;
        float ps = 0.f;
        h16x8 pk[PLANES][2][2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[sub][r], c, -mc));
                ps += e;
                h16 piece[PLANES];
                split_operand(e, piece);
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) pk[pl][sub][r >> 3][r & 7] = piece[pl];
            }
        l_run = l_run * alpha + ps;
        if (grew) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }

#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                const h16* vp = Vs + (cur * PLANES + seg_xp(sg)) * ATILE + (dt * 32 + l31) * ALD + 4 * hi;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int kk = sub * 32 + jj * 16;
                        const h16x4 lo = *reinterpret_cast<const h16x4*>(vp + kk);
                        const h16x4 up = *reinterpret_cast<const h16x4*>(vp + kk + 8);
                        h16x8 vf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = up[e]; }
                        o[dt] = MFMA_32x32x16(vf, pk[seg_wp(sg)][sub][jj], o[dt]);
                    }
            }

        if (more) stage(cur ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] /= l_tot; o[1][r] /= l_tot; }
    if (TWO) {
        if (set == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { res[0][r] = o[0][r]; res[1][r] = o[1][r]; }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] += res[0][r]; o[1][r] += res[1][r]; }
        }
    }
    }   // sets
    if (qok) {
        h16* orow = Op + (int64_t)q * p.ldo;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16* dst = orow + dt * 32 + 8 * g + 4 * hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = o[dt][4 * g + j];
                    if (p.accumulate) v += load1_operand(dst + j, pso);
                    store1_operand(dst + j, pso, v);
                }
            }
    }
}

#if MUDG_PLANES == 2
// The long self-attention of the bf16x3 build on the staging of attn64d_kernel (round 4; attn_split_kernel above was the
// round-1 register-staged design: global loads -> registers -> ds_write -> 48 eight-byte V reads per key tile, 38 % MFMA-busy):
//  * K and V^T tiles of BOTH pieces go HBM/L2 -> LDS by LDS-DMA (no register round trip, no ds_write pass), unpadded [64][64]
//    tiles with the source-side XOR swizzle, K rows permuted at the source by the involution that swaps the middle 4-blocks of
//    every 16, so that a P V fragment is one natural 16-byte chunk of V^T (see attn64d_kernel);
//  * every fragment of piece 0 is read once and feeds two of the three kept products (k0 q1 + k0 q0, v0 p1 + v0 p0);
//  * the lean softmax (q prescaled: the reference maximum of the first key tile is the score accumulators' initial value, one
//    v_exp + one add + the split per score, no rescale of O), with the classic online softmax as the overflow fallback;
//  * 32 queries per wave (q pieces 32 + scores 32 + P pieces 32 + O 32 registers): the 64-query form of the 16-bit kernel
//    would need 256 registers for those alone.
// Same arithmetic as attn_split_kernel per (query, key tile): x1 w0 + x0 w1 + x0 w0 for both contractions, fp32 softmax.
constexpr int SDT = 64 * 64;             // h16 per unpadded tile
constexpr int SPLIT_DMA_SMEM = 2 * 2 * 2 * SDT * (int)sizeof(h16);        // [buffer][K | V][piece] = 64 KiB
template <bool LEAN>
__global__ __launch_bounds__(256, 2) void attn_split_dma_kernel(const MudgAttnDesc p, const int nqt, const int total) {
    extern __shared__ __attribute__((aligned(1024))) char smem_dma[];
    h16* Ks = reinterpret_cast<h16*>(smem_dma);                   // [2 buffers][2 pieces][SDT]
    h16* Vs = Ks + 4 * SDT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int w;
    {
        const int q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pair = w / nqt, qt = w - pair * nqt;
    const int f = pair / p.heads, h = pair - f * p.heads;
    const int kvb = f / p.kv_div;
    const int psq = p.ldq / 2, psk = p.ldk / 2, psv = p.ldvt / 2, pso = p.ldo / 2;

    const h16* Qp = reinterpret_cast<const h16*>(p.Q) + (int64_t)f * p.Nq * p.ldq + h * 64;
    const h16* Kp = reinterpret_cast<const h16*>(p.K) + (int64_t)kvb * p.Nk * p.ldk + h * 64;
    const h16* Vp = reinterpret_cast<const h16*>(p.Vt) + (int64_t)kvb * p.svt + (int64_t)(h * 64) * p.ldvt;
    h16* Op = reinterpret_cast<h16*>(p.O) + (int64_t)f * p.Nq * p.ldo + h * 64;

    const int q = qt * QB + wave * 32 + l31;
    const bool qok = q < p.Nq;
    h16x8 qf[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[pl][ks] = as_h16x8(qok ? ld16(Qp + (int64_t)q * p.ldq + pl * psq + ks * 16 + hi * 8) : zero16());

    // DMA geometry: wave w stages rows [16w, 16w + 16) of the four tiles, two 1-KiB instructions each
    const __amdgpu_buffer_rsrc_t rK = attn_rsrc(Kp), rV = attn_rsrc(Vp);
    unsigned vk[2], vv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * wave + 8 * i + (lane >> 3), slot = lane & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int i16 = row & 15;
        const int key = (row & ~15) | (i16 & 3) | ((i16 & 8) >> 1) | ((i16 & 4) << 1);     // swap the middle 4-blocks
        vk[i] = (unsigned)key * (unsigned)p.ldk * 2u + (unsigned)chunk * 16u;
        vv[i] = (unsigned)row * (unsigned)p.ldvt * 2u + (unsigned)chunk * 16u;
    }
    auto request = [&](int kt, int buf) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (attn_lptr_t)(Ks + (buf * 2 + pl) * SDT + (16 * wave + 8 * i) * 64), 16, (int)vk[i],
                                                         (kt * 64 * p.ldk + pl * psk) * 2, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (attn_lptr_t)(Vs + (buf * 2 + pl) * SDT + (16 * wave + 8 * i) * 64), 16, (int)vv[i],
                                                         (kt * 64 + pl * psv) * 2, 0, 0);
            }
    };
    const int sw = (l31 >> 1) & 7;
    // scores of 32 keys (sub-tile `sub` of K tile `buf`) x this wave's 32 queries: k1 q0 + k0 q1 + k0 q0, small terms first
    auto score_block = [&](int buf, int sub, f32x16& sc) {
        const h16* k0 = Ks + (buf * 2) * SDT + (sub * 32 + l31) * 64;
        const h16* k1 = k0 + SDT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = ((ks * 2 + hi) ^ sw) << 3;
            const h16x8 f1 = *reinterpret_cast<const h16x8*>(k1 + off), f0 = *reinterpret_cast<const h16x8*>(k0 + off);
            sc = MFMA_32x32x16(f1, qf[0][ks], sc);
            sc = MFMA_32x32x16(f0, qf[1][ks], sc);
            sc = MFMA_32x32x16(f0, qf[0][ks], sc);
        }
    };

    f32x16 o[2];
    float m_run, l_run;
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;
    const int nkt = p.Nk / KB;
    bool overflow = false;

    auto key_loop = [&](auto lean_tag) {
        constexpr bool LN = decltype(lean_tag)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
        m_run = LN ? 0.f : -INFINITY;
        l_run = 0.f;
        request(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (LN) {      // reference maximum = the first tile's row maximum (one extra set of score MFMAs)
            f32x16 s0[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s0[sub][r] = 0.f;
                score_block(0, sub, s0[sub]);
            }
            float mx = s0[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s0[1][r]);
            m_run = fmaxf(mx, __shfl_xor(mx, 32, 64));
        }
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) request(kt + 1, cur ^ 1);
            f32x16 sc[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const float init = LN ? -m_run : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[sub][r] = init;
                score_block(cur, sub, sc[sub]);
            }
            h16x8 pk[2][2][2];        // [piece][sub][jj]
            float ps = 0.f;
            if constexpr (LN) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __builtin_amdgcn_exp2f(sc[sub][r]);
                        ps += e;
                        const h16 p0 = (h16)e;
                        pk[0][sub][r >> 3][r & 7] = p0;
                        pk[1][sub][r >> 3][r & 7] = (h16)(e - (float)p0);
                    }
                l_run += ps;
                overflow = overflow || !(ps <= LEAN_LIMIT);
            } else {
                float mx = sc[0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const bool grew = !__all(mx <= m_run);
                const float m_new = grew ? fmaxf(m_run, mx) : m_run;
                const float alpha = grew ? __builtin_amdgcn_exp2f((m_run - m_new) * c) : 1.0f;
                const float mc = m_new * c;
                m_run = m_new;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(sc[sub][r], c, -mc));
                        ps += e;
                        const h16 p0 = (h16)e;
                        pk[0][sub][r >> 3][r & 7] = p0;
                        pk[1][sub][r >> 3][r & 7] = (h16)(e - (float)p0);
                    }
                l_run = l_run * alpha + ps;
                if (grew) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                }
            }
            // P V: v1 p0 + v0 p1 + v0 p0; register group (sub, jj) of half hi holds keys 32 sub + 16 jj + 8 hi .. + 7
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const h16* v0 = Vs + (cur * 2) * SDT + (dt * 32 + l31) * 64;
                const h16* v1 = v0 + SDT;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int off = ((4 * sub + 2 * jj + hi) ^ sw) << 3;
                        const h16x8 f1 = *reinterpret_cast<const h16x8*>(v1 + off), f0 = *reinterpret_cast<const h16x8*>(v0 + off);
                        o[dt] = MFMA_32x32x16(f1, pk[0][sub][jj], o[dt]);
                        o[dt] = MFMA_32x32x16(f0, pk[1][sub][jj], o[dt]);
                        o[dt] = MFMA_32x32x16(f0, pk[0][sub][jj], o[dt]);
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt + 1
            __syncthreads();                                      // every wave's; everyone is done reading tile kt
        }
    };

    if constexpr (LEAN) {
        key_loop(std::true_type{});
        if (__syncthreads_or(overflow ? 1 : 0)) key_loop(std::false_type{});      // cold: exact, merely slower
    } else {
        key_loop(std::false_type{});
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qok) {
        h16* orow = Op + (int64_t)q * p.ldo;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h16* dst = orow + dt * 32 + 8 * g + 4 * hi;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[dt][4 * g + j] * inv;
                if (p.accumulate) {
                    Pack8 a, b; a.u = *reinterpret_cast<const u32x2*>(dst); b.u = *reinterpret_cast<const u32x2*>(dst + pso);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)a.h[j] + (float)b.h[j];
                }
                Pack8 n0, n1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { n0.h[j] = (h16)v[j]; n1.h[j] = (h16)(v[j] - (float)n0.h[j]); }
                *reinterpret_cast<u32x2*>(dst) = n0.u;
                *reinterpret_cast<u32x2*>(dst + pso) = n1.u;
            }
    }
}
#endif

// Temporal attention of the split builds: q / k / v pieces are summed to fp32 on load and the T x T problem runs on
// fp32 FMAs (this kernel is bandwidth-bound in every build).  One wave per (pixel, head), K / V of the item in LDS.
template <int TP>
__global__ __launch_bounds__(256, 1) void tattn_split_kernel(const h16* __restrict__ QKV, h16* __restrict__ O, int B, int T, int HW,
                                                              int heads, int ldqkv, int ldo, float scale, int total) {
    constexpr int DP = 64 / TP, DW = 64 / DP;
    __shared__ float Ks[4][TP * 64];
    __shared__ float Vs[4][TP * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tq = lane / DP, dp = lane % DP;
    const int C = heads * 64;
    const int w = blockIdx.x * 4 + wave;
    if (w >= total) return;
    const int bp = w / heads, h = w - bp * heads;
    const int b = bp / HW, px = bp - b * HW;
    const bool ok = tq < T;
    const int64_t row = ((int64_t)(b * T + tq) * HW + px);
    const h16* src = QKV + row * ldqkv + h * 64 + dp * DW;
    const int64_t ps = ldqkv / PLANES;
    float qv[DW];
#pragma unroll
    for (int i = 0; i < DW / 8; ++i) {
        float q8[8], k8[8], v8[8];
        if (ok) {
            load8_operand(src + i * 8, ps, q8);
            load8_operand(src + C + i * 8, ps, k8);
            load8_operand(src + 2 * C + i * 8, ps, v8);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { q8[e] = 0.f; k8[e] = 0.f; v8[e] = 0.f; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qv[i * 8 + e] = q8[e];
            Ks[wave][tq * 64 + dp * DW + i * 8 + e] = k8[e];
            Vs[wave][tq * 64 + dp * DW + i * 8 + e] = v8[e];
        }
    }
    __builtin_amdgcn_wave_barrier();
    float sc[TP];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < DW; ++d) a = fmaf(qv[d], Ks[wave][j * 64 + dp * DW + d], a);
#pragma unroll
        for (int o = 1; o < DP; o <<= 1) a += __shfl_xor(a, o, 64);
        a = (j < T) ? a * scale : -INFINITY;
        sc[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < TP; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
    float ov[DW];
#pragma unroll
    for (int d = 0; d < DW; ++d) ov[d] = 0.f;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const float pj = sc[j] / sum;
#pragma unroll
        for (int d = 0; d < DW; ++d) ov[d] = fmaf(pj, Vs[wave][j * 64 + dp * DW + d], ov[d]);
    }
    if (ok) {
        h16* dst = O + row * ldo + h * 64 + dp * DW;
#pragma unroll
        for (int i = 0; i < DW / 8; ++i) {
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = ov[i * 8 + e];
            store8_operand(dst + i * 8, ldo / PLANES, o8);
        }
    }
}

#if MUDG_PLANES == 2
// tattn_mfma_kernel (above, 16-bit builds) for the bf16x3 build: q, k, v arrive as two bf16 pieces each, every contraction keeps
// the three partial products of the fused-piece GEMMs — S^T = k0 q1 + k1 q0 + k0 q0 (six v_mfma_f32_16x16x32), the probabilities are
// split in registers (p = p0 + p1), O^T = v0 p1 + v1 p0 + v0 p0 (twelve v_mfma_f32_16x16x16) — small terms first; exponentials and the
// normalisation are full-precision fp32 as everywhere in this build.  One wave per item at a time, TATTN_SPLIT_ITEMS items per wave
// with the next item's twelve requests in flight; six 2-KiB LDS tiles per wave.
constexpr int TATTN_SPLIT_ITEMS = 4;
typedef __attribute__((ext_vector_type(4))) short mf4s;
__global__ __launch_bounds__(256) void tattn_split_mfma_kernel(const h16* __restrict__ QKV, h16* __restrict__ O,
                                                                int B, int T, int HW, int heads, int ldqkv, int ldo,
                                                                float scale, int total) {
    __shared__ __attribute__((aligned(16))) h16 Ts[4][3][2][16 * 64];          // [wave][q | k | v][piece][16 rows x 64 dims]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;          // MFMA coordinates
    const int lr = lane >> 3, lj = lane & 7;         // fetch coordinates: rows lr and lr + 8, 16-byte chunk lj
    const int C = heads * 64;
    const int64_t ps = ldqkv / PLANES;
    const int w0 = (blockIdx.x * 4 + wave) * TATTN_SPLIT_ITEMS;
    auto tile = [&](int which, int piece) { return reinterpret_cast<char*>(&Ts[wave][which][piece][0]); };
    auto fz = [](int rg) { return (rg & 1) | ((rg >> 1) << 3); };

    u32x4 nx[3][2][2];                               // the item being fetched: [q | k | v][piece][row half]
    int64_t pixn = 0; int hn = 0;
    auto fetch = [&](int w) {
        const bool item = w < total;
        const int bp = item ? w / heads : 0;
        hn = item ? w - bp * heads : 0;
        const int b = bp / HW, px = bp - b * HW;
        pixn = (int64_t)b * T * HW + px;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int t = lr + 8 * s;
            const bool ok = item && t < T;
            const h16* src = QKV + (pixn + (int64_t)(ok ? t : 0) * HW) * ldqkv + hn * 64 + lj * 8;
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) nx[x][pc][s] = ok ? ld16(src + x * C + pc * ps) : zero16();
        }
    };
    fetch(w0);

    for (int it = 0; it < TATTN_SPLIT_ITEMS; ++it) {
        if (w0 + it >= total) break;                 // wave-uniform
        const int64_t pix = pixn; const int h = hn;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int r = lr + 8 * s, z = fz(r >> 2);
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                st16(tile(0, pc) + r * 128 + ((lj ^ (r & 7)) << 4), nx[0][pc][s]);
                st16(tile(1, pc) + r * 128 + ((lj ^ (r & 7)) << 4), nx[1][pc][s]);
                u32x2 lo = {nx[2][pc][s][0], nx[2][pc][s][1]}, hi = {nx[2][pc][s][2], nx[2][pc][s][3]};
                *reinterpret_cast<u32x2*>(tile(2, pc) + r * 128 + (((2 * lj) ^ z) << 3)) = lo;
                *reinterpret_cast<u32x2*>(tile(2, pc) + r * 128 + (((2 * lj + 1) ^ z) << 3)) = hi;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (it + 1 < TATTN_SPLIT_ITEMS) fetch(w0 + it + 1);

        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int term = 0; term < 3; ++term) {       // (k piece, q piece): (0,1) (1,0) (0,0)
            const int kp = term == 1 ? 1 : 0, qp = term == 0 ? 1 : 0;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = c * 128 + (((g + 4 * s) ^ (c & 7)) << 4);
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_h16x8(ld16(tile(1, kp) + off)), as_h16x8(ld16(tile(0, qp) + off)), st, 0, 0, 0);
            }
        }
        float sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[i] = (4 * g + i < T) ? st[i] * scale : -INFINITY; mx = fmaxf(mx, sc[i]); }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        union { mf4s m; h16 e[4]; } p0, p1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pv = expf(sc[i] - mx);
            sum += pv;
            p0.e[i] = (h16)pv;
            p1.e[i] = (h16)(pv - (float)p0.e[i]);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);

        float ov[16];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            union { mf4s m4; unsigned short e[4]; } v0, v1;
            const int u = (8 * (m >> 1) + 2 * (c >> 2) + (m & 1)) ^ fz(g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int off = (4 * g + j) * 128 + (u << 3) + 2 * (c & 3);
                v0.e[j] = *reinterpret_cast<const unsigned short*>(tile(2, 0) + off);
                v1.e[j] = *reinterpret_cast<const unsigned short*>(tile(2, 1) + off);
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(v0.m4, p1.m, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(v1.m4, p0.m, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(v0.m4, p0.m, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) ov[4 * m + i] = acc[i] / sum;
        }
        if (c < T) {
            h16* dst = O + (pix + (int64_t)c * HW) * ldo + h * 64 + g * 8;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float o8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = ov[half * 8 + e];
                store8_operand(dst + half * 32, ldo / PLANES, o8);
            }
        }
    }
}
#endif
#endif  // MUDG_PLANES > 1

}  // namespace

extern "C" int mudg_attention(const MudgAttnDesc* dp, void* stream) {
    MUDG_REQUIRE(dp, "mudg_attention: null descriptor");
    const MudgAttnDesc d = *dp;
    MUDG_REQUIRE(d.Q && d.K && d.Vt && d.O, "mudg_attention: null pointer");
    MUDG_REQUIRE(d.F > 0 && d.heads > 0 && d.Nq > 0 && d.Nk > 0, "mudg_attention: empty problem");
    MUDG_REQUIRE(d.kv_div >= 1 && d.F % d.kv_div == 0, "mudg_attention: kv_div=%d F=%d", d.kv_div, d.F);
    MUDG_REQUIRE(d.ldq % (8 * PLANES) == 0 && d.ldk % (8 * PLANES) == 0 && d.ldvt % (8 * PLANES) == 0 && d.ldo % (4 * PLANES) == 0 &&
                 (d.svt & 7) == 0, "mudg_attention: row strides must be multiples of %d (ldo of %d)", 8 * PLANES, 4 * PLANES);
    MUDG_REQUIRE(d.ldvt / PLANES >= d.Nk, "mudg_attention: ldvt=%d < %d x Nk=%d", d.ldvt, PLANES, d.Nk);
    MUDG_REQUIRE(d.ldq / PLANES >= d.heads * 64 && d.ldk / PLANES >= d.heads * 64 && d.ldo / PLANES >= d.heads * 64,
                 "mudg_attention: row strides too small for %d heads", d.heads);
    MUDG_REQUIRE(aligned16(d.Q) && aligned16(d.K) && aligned16(d.Vt) && aligned16(d.O), "mudg_attention: alignment");
    if (d.K2) {
        MUDG_REQUIRE(d.Vt2 && d.Nk2 > 0 && d.kv_div2 >= 1 && d.F % d.kv_div2 == 0, "mudg_attention: second key/value set");
        MUDG_REQUIRE(d.ldk2 % (8 * PLANES) == 0 && d.ldvt2 % (8 * PLANES) == 0 && (d.svt2 & 7) == 0 && d.ldvt2 / PLANES >= d.Nk2 &&
                     d.ldk2 / PLANES >= d.heads * 64 && aligned16(d.K2) && aligned16(d.Vt2), "mudg_attention: second key/value set strides");
        MUDG_REQUIRE(!d.accumulate, "mudg_attention: accumulate and a second key/value set are exclusive");
    }
    MUDG_REQUIRE(!d.Lse || (PLANES == 1 && !d.K2 && !d.q_prescaled && !d.Q8 && !d.accumulate),
                 "mudg_attention: Lse (the softmax statistics for mudg_attention_bwd) comes from the plain single-set kernels of the 16-bit builds");
    const int nqt = (d.Nq + QB - 1) / QB;
    const int64_t total = (int64_t)nqt * d.F * d.heads;
    MUDG_REQUIRE(total < (1ll << 31), "mudg_attention: grid too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int slot = mudg_prof_begin(MUDG_FAM_ATTN, s);
    // Long self-attention runs 64 queries per wave (attn64q_kernel: +4.7 % at N = 9216); short key sequences (the text /
    // image cross-attention) and small query counts keep the 32-query kernel.  MUDG_ATTN_Q=32 / 64 forces one of them.
    static int var = -1;
    if (var < 0) var = mudg_variant("ATTN_Q", 0);
    const bool wide = !d.K2 && (var == 64 ? d.Nq >= 256 : (var == 32 ? false : (d.Nq >= 512 && d.Nk >= 256)));
#if MUDG_PLANES > 1
    {
        constexpr int smem = 4 * PLANES * ATILE * (int)sizeof(h16);
        static bool attr_done[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) MUDG_FAIL(MUDG_ELAUNCH, "mudg_attention: hipGetDevice");
        if (!attr_done[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_split_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_split_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "mudg_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_done[dev] = true;
        }
        bool dma_ok = false;
#if MUDG_PLANES == 2
        // the long self-attention on LDS-DMA staged tiles (whole key tiles inside the 2-GiB window of a buffer descriptor)
        dma_ok = wide && d.Nk % 64 == 0 && (int64_t)d.Nk * d.ldk * 2 < (1ll << 31) && (int64_t)64 * d.ldvt * 2 + (int64_t)d.Nk * 2 + d.ldvt < (1ll << 31);
        if (dma_ok) {
            static bool attr2[64] = {};
            if (!attr2[dev]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_split_dma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SPLIT_DMA_SMEM);
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_split_dma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SPLIT_DMA_SMEM);
                if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "mudg_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
                attr2[dev] = true;
            }
            if (d.q_prescaled) hipLaunchKernelGGL(attn_split_dma_kernel<true>, dim3((unsigned)total), dim3(256), SPLIT_DMA_SMEM, s, d, nqt, (int)total);
            else hipLaunchKernelGGL(attn_split_dma_kernel<false>, dim3((unsigned)total), dim3(256), SPLIT_DMA_SMEM, s, d, nqt, (int)total);
        }
#endif
        (void)wide;
        if (dma_ok) {}
        else if (d.K2) hipLaunchKernelGGL(attn_split_kernel<true>, dim3((unsigned)total), dim3(256), smem, s, d, nqt, (int)total);
        else hipLaunchKernelGGL(attn_split_kernel<false>, dim3((unsigned)total), dim3(256), smem, s, d, nqt, (int)total);
    }
#else
    MUDG_REQUIRE(!d.Q8 || wide, "mudg_attention: the MX-fp8 score path serves the long self-attention kernel only (Nq >= 512, Nk >= 256)");
    if (wide) {
        const int nqt2 = (d.Nq + 255) / 256;
        const int64_t total2 = (int64_t)nqt2 * d.F * d.heads;
        // LDS-DMA staging (+ the lean softmax when Q is prescaled) for whole key tiles whose tiles stay inside the 2-GiB
        // window of a buffer descriptor; the register-staged kernel otherwise.  MUDG_ATTN_DMA=0 / MUDG_ATTN_LEAN=0: A/B.
        static int dma = -1, lean = -1;
        if (dma < 0) dma = mudg_variant("ATTN_DMA", 1);
        if (lean < 0) lean = mudg_variant("ATTN_LEAN", 1);
        const bool dma_ok = dma && d.Nk % 64 == 0 && (int64_t)d.Nk * d.ldk * 2 < (1ll << 31) && (int64_t)64 * d.ldvt * 2 + (int64_t)d.Nk * 2 < (1ll << 31);
        if (d.Q8) {
            MUDG_REQUIRE(dma_ok && d.q_prescaled && d.K8 && d.Qs && d.Ks && !d.accumulate, "mudg_attention: the MX-fp8 score path needs "
                         "q_prescaled, Nk %% 64 == 0 and all four of Q8 / K8 / Qs / Ks");
            MUDG_REQUIRE((d.ldq8 & 15) == 0 && (d.ldk8 & 15) == 0 && aligned16(d.Q8) && aligned16(d.K8) && d.ldq8 >= d.heads * 64 &&
                         d.ldk8 >= d.heads * 64 && d.ldqs >= d.heads * 2 && d.ldks >= d.heads * 2 && (int64_t)d.Nk * d.ldk8 < (1ll << 31),
                         "mudg_attention: fp8 strides");
            hipLaunchKernelGGL((attn64d_kernel<true, true>), dim3((unsigned)total2), dim3(256), 0, s, d, nqt2, (int)total2);
        }
        else if (dma_ok && d.q_prescaled && lean) hipLaunchKernelGGL((attn64d_kernel<true, false>), dim3((unsigned)total2), dim3(256), 0, s, d, nqt2, (int)total2);
        else if (dma_ok) hipLaunchKernelGGL((attn64d_kernel<false, false>), dim3((unsigned)total2), dim3(256), 0, s, d, nqt2, (int)total2);
        else hipLaunchKernelGGL(attn64q_kernel, dim3((unsigned)total2), dim3(256), 0, s, d, nqt2, (int)total2);
    } else {
        // Short key sets with many query tiles (the cross-attention of the fine levels): the key tiles resident, xq query tiles per
        // workgroup (xattn_kernel: the same bits).  xq from the tile count only: >= 4 workgroups per CU stay.  MUDG_ATTN_X=0: off (A/B).
        static int xv = -1;
        if (xv < 0) xv = mudg_variant("ATTN_X", 1);
        int xq = (int)(total / 1024);
        xq = xq > 8 ? 8 : xq;
        if (xv && xq >= 2 && d.Nk <= 2 * KB && (!d.K2 || d.Nk2 <= KB)) {
            constexpr int smem = 2 * XK_TILES * ATILE * (int)sizeof(h16);
            static bool attr_done[64] = {};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) MUDG_FAIL(MUDG_ELAUNCH, "mudg_attention: hipGetDevice");
            if (!attr_done[dev]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
                if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "mudg_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
                attr_done[dev] = true;
            }
            const int nqc = (nqt + xq - 1) / xq;
            const int64_t totx = (int64_t)nqc * d.F * d.heads;
            if (d.K2) hipLaunchKernelGGL(xattn_kernel<true>, dim3((unsigned)totx), dim3(256), smem, s, d, nqt, xq, nqc, (int)totx);
            else hipLaunchKernelGGL(xattn_kernel<false>, dim3((unsigned)totx), dim3(256), smem, s, d, nqt, xq, nqc, (int)totx);
        }
        else if (d.K2) hipLaunchKernelGGL(attn_kernel<true>, dim3((unsigned)total), dim3(256), 0, s, d, nqt, (int)total);
        else hipLaunchKernelGGL(attn_kernel<false>, dim3((unsigned)total), dim3(256), 0, s, d, nqt, (int)total);
    }
#endif
    const int rc = mudg_check_launch("mudg_attention");
    const double bh = (double)d.F * d.heads;
    mudg_prof_end(slot, s, 4.0 * bh * d.Nq * (double)(d.Nk + (d.K2 ? d.Nk2 : 0)) * 64.0,
                  bh * (2.0 * d.Nq + 2.0 * d.Nk / d.kv_div + (d.K2 ? 2.0 * d.Nk2 / d.kv_div2 : 0.0)) * 64.0 * 2.0);
    return rc;
}

#if MUDG_PLANES == 1
namespace {
// One thread per (row, 32-column block): OCP MX e4m3 with an E8M0 block scale.
__global__ __launch_bounds__(256) void quantize_mxfp8_kernel(const h16* __restrict__ X, int ldx, int64_t rows, int nblk,
                                                              unsigned char* __restrict__ Y, int ldy, unsigned char* __restrict__ S, int lds) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * nblk) return;
    const int64_t r = i / nblk;
    const int b = (int)(i - r * nblk);
    const h16* src = X + r * ldx + b * 32;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const h16x8 t = as_h16x8(ld16(src + q * 8));
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[q * 8 + e] = (float)t[e]; amax = fmaxf(amax, fabsf(v[q * 8 + e])); }
    }
    const int E = mx_block_exponent(amax);       // shared exponent: floor(log2 amax) - emax(e4m3) = ... - 8
    const float inv = __uint_as_float((unsigned)(127 - E) << 23);           // 2^-E
    u32x4 lo, hi4;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned w = mx_pack4_e4m3(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3], inv);
        if (q < 4) lo[q] = w; else hi4[q - 4] = w;
    }
    unsigned char* dst = Y + r * ldy + b * 32;
    st16(dst, lo);
    st16(dst + 16, hi4);
    S[r * lds + b] = (unsigned char)(E + 127);
}
}  // namespace
#endif

extern "C" int mudg_quantize_mxfp8(const void* X, int ldx, int64_t rows, int cols, void* Y8, int ldy, void* S, int lds, void* stream) {
#if MUDG_PLANES == 1
    MUDG_REQUIRE(X && Y8 && S && rows > 0 && cols > 0 && cols % 32 == 0, "mudg_quantize_mxfp8: bad arguments");
    MUDG_REQUIRE((ldx & 7) == 0 && (ldy & 15) == 0 && ldx >= cols && ldy >= cols && lds >= cols / 32 && aligned16(X) && aligned16(Y8),
                 "mudg_quantize_mxfp8: strides / alignment");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nblk = cols / 32;
    const int64_t n = rows * nblk;
    const int slot = mudg_prof_begin(MUDG_FAM_MISC, s);
    hipLaunchKernelGGL(quantize_mxfp8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const h16*)X, ldx, rows, nblk,
                       (unsigned char*)Y8, ldy, (unsigned char*)S, lds);
    const int rc = mudg_check_launch("mudg_quantize_mxfp8");
    mudg_prof_end(slot, s, 0.0, (double)rows * cols * 3.0);
    return rc;
#else
    (void)X; (void)ldx; (void)rows; (void)cols; (void)Y8; (void)ldy; (void)S; (void)lds; (void)stream;
    MUDG_FAIL(MUDG_EUNSUPPORTED, "mudg_quantize_mxfp8: fp8 scores belong to the 16-bit operand builds");
#endif
}

extern "C" int mudg_temporal_attention(const void* QKV, void* O, int B, int T, int HW, int heads,
                                       int ldqkv, int ldo, float scale, void* stream) {
    MUDG_REQUIRE(QKV && O, "mudg_temporal_attention: null pointer");
    MUDG_REQUIRE(B > 0 && HW > 0 && heads > 0, "mudg_temporal_attention: empty problem");
    MUDG_REQUIRE(T >= 1 && T <= 32, "mudg_temporal_attention: T=%d outside [1,32]", T);
    MUDG_REQUIRE(ldqkv % (8 * PLANES) == 0 && ldo % (8 * PLANES) == 0 && aligned16(QKV) && aligned16(O), "mudg_temporal_attention: alignment");
    MUDG_REQUIRE(ldqkv / PLANES >= 3 * heads * 64 && ldo / PLANES >= heads * 64, "mudg_temporal_attention: row strides too small");
    const int64_t total = (int64_t)B * HW * heads;
    MUDG_REQUIRE(total < (1ll << 31), "mudg_temporal_attention: grid too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int slot = mudg_prof_begin(MUDG_FAM_TATTN, s);
#if MUDG_PLANES > 1
    const unsigned grid = (unsigned)((total + 3) / 4);
#if MUDG_PLANES == 2
    static int use_mfma = -1;               // MUDG_TATTN_MFMA=0: the fp32 FMA kernel for every length (A/B, tests)
    if (use_mfma < 0) use_mfma = mudg_variant("TATTN_MFMA", 1);
    if (T <= 16 && use_mfma)
        hipLaunchKernelGGL(tattn_split_mfma_kernel, dim3((unsigned)((total + 4 * TATTN_SPLIT_ITEMS - 1) / (4 * TATTN_SPLIT_ITEMS))), dim3(256), 0, s,
                           (const h16*)QKV, (h16*)O, B, T, HW, heads, ldqkv, ldo, scale, (int)total);
    else
#endif
    if (T <= 16)
        hipLaunchKernelGGL(tattn_split_kernel<16>, dim3(grid), dim3(256), 0, s, (const h16*)QKV, (h16*)O, B, T, HW, heads,
                           ldqkv, ldo, scale, (int)total);
    else
        hipLaunchKernelGGL(tattn_split_kernel<32>, dim3(grid), dim3(256), 0, s, (const h16*)QKV, (h16*)O, B, T, HW, heads,
                           ldqkv, ldo, scale, (int)total);
#else
    const unsigned grid = (unsigned)((total + 4 * TATTN_ITEMS - 1) / (4 * TATTN_ITEMS));
    static int use_mfma = -1;               // MUDG_TATTN_MFMA=0: the VALU kernel for every length (A/B, tests)
    if (use_mfma < 0) use_mfma = mudg_variant("TATTN_MFMA", 1);
    if (T <= 16 && use_mfma)
        hipLaunchKernelGGL(tattn_mfma_kernel, dim3(grid), dim3(256), 0, s, (const h16*)QKV, (h16*)O, B, T, HW, heads,
                           ldqkv, ldo, scale, (int)total);
    else if (T <= 16)
        hipLaunchKernelGGL(tattn_kernel<16>, dim3(grid), dim3(256), 0, s, (const h16*)QKV, (h16*)O, B, T, HW, heads,
                           ldqkv, ldo, scale, (int)total);
    else
        hipLaunchKernelGGL(tattn_kernel<32>, dim3(grid), dim3(256), 0, s, (const h16*)QKV, (h16*)O, B, T, HW, heads,
                           ldqkv, ldo, scale, (int)total);
#endif
    const int rc = mudg_check_launch("mudg_temporal_attention");
    mudg_prof_end(slot, s, 4.0 * total * (double)T * T * 64.0, (double)total * T * 64.0 * 2.0 * 4.0);
    return rc;
}
