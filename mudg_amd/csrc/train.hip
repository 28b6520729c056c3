// train.hip — the kernels only the TRAINING step needs (reference: lvdm/models/ddpm3d.py:741-802 p_losses,
// :1267-1300 configure_optimizers -> AdamW): backward passes of the normalisations, of GEGLU, of the softmax inside attention
// and of the temporal attention, the transposed / gathered operand copies the weight-gradient GEMMs contract over, the
// column sums behind bias gradients, the weighted MSE and its gradient, nearest-2x resampling both ways, and AdamW.
// The contractions of the backward pass themselves (dX = dY W, dW = dY^T X, conv / temporal-conv input gradients) run on
// the forward GEMM kernels (gemm.hip) — see mudg_amd/train/functions.py for how each one is expressed.
// Gradients and the training stream are fp32 rows matrices; operands for the MFMA kernels are made by mudg_cast_rows /
// mudg_transpose_gather.  Every reduction has a fixed order: no atomics, bit-reproducible gradients.
#include "common.h"

namespace {

__device__ __forceinline__ float gelu_grad(float x) {        // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}
__device__ __forceinline__ float silu_grad(float z) {        // d/dz [z sigma(z)] = sigma (1 + z (1 - sigma))
    const float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

// ---------------------------------------------------------------------------------------------- transpose / gather
// dst[c][p] = src[srcrow(p)][c] (or 0 where the source pixel of output position p does not exist), p < P; columns P .. Ppad
// are zero.  dst is an MFMA operand matrix [C][ldd]; src fp32 rows [*][lds].  One 64 x 64 tile per 256-thread workgroup.
//   mode 0: srcrow = p
//   mode 1: p = (f, oy, ox) of an Hout x Wout grid, tap (dy, dx): source pixel (oy stride - pad + dy, ox stride - pad + dx)
//   mode 2: p = ((b T + t) HW + s), tap dt: source row p + (dt - 1) HW while 0 <= t + dt - 1 < T
struct GatherGeo { int mode, Hin, Win, Hout, Wout, stride, pad, dy, dx, T, HW, dt; };

__device__ __forceinline__ int64_t gather_row(const GatherGeo& g, int64_t p) {
    if (g.mode == 0) return p;
    if (g.mode == 1) {
        const int hw = g.Hout * g.Wout;
        const int64_t f = p / hw;
        const int r = (int)(p - f * hw);
        const int oy = r / g.Wout, ox = r - oy * g.Wout;
        const int iy = oy * g.stride - g.pad + g.dy, ix = ox * g.stride - g.pad + g.dx;
        if (iy < 0 || iy >= g.Hin || ix < 0 || ix >= g.Win) return -1;
        return (f * g.Hin + iy) * g.Win + ix;
    }
    const int t = (int)((p / g.HW) % g.T) + g.dt - 1;
    if (t < 0 || t >= g.T) return -1;
    return p + (int64_t)(g.dt - 1) * g.HW;
}

__global__ __launch_bounds__(256) void transpose_gather_kernel(const float* __restrict__ src, int64_t lds, h16* __restrict__ dst, int64_t ldd,
                                                                int64_t P, int64_t Ppad, int C, GatherGeo g, int64_t sbs, int64_t dbs) {
    __shared__ float tile[64][65];
    src += (int64_t)blockIdx.z * sbs;               // batch entry z: its own source rows and destination block
    dst += (int64_t)blockIdx.z * dbs;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4: a wave reads 256 B of a source row, writes 128 B of a destination row
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int64_t p = p0 + ty + 4 * i;
        const int c = c0 + tx;
        float v = 0.f;
        if (p < P && c < C) {
            const int64_t r = gather_row(g, p);
            if (r >= 0) v = src[r * lds + c];
        }
        tile[ty + 4 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i;
        const int64_t p = p0 + tx;
        if (c < C && p < Ppad) store1_operand(dst + (int64_t)c * ldd + p, ldd / PLANES, tile[tx][ty + 4 * i]);
    }
}

// The gradient entering a linear / conv layer is needed three ways: transposed (left operand of the weight-gradient GEMM), as
// operand rows (left operand of the input-gradient GEMM) and summed over the rows (bias gradient).  One pass over the fp32 rows
// writes all three: a 64 x 64 tile per workgroup, 16-byte reads, 8-byte operand stores, and per tile the 64-row column sums
// (fixed order) into part[tile][C] — mudg_group_colsum folds the tiles.  dst / rows / part may each be null.  C % 4 == 0.
__global__ __launch_bounds__(256) void xpose_cast_sum_kernel(const float* __restrict__ src, int64_t lds, h16* __restrict__ dst, int64_t ldd,
                                                              h16* __restrict__ rows, int64_t ldr, float* __restrict__ part, int64_t P,
                                                              int64_t Ppad, int C) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    {
        const int r = t >> 4, c = c0 + (t & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t p = p0 + r + 16 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < P && c < C) {
                v = *reinterpret_cast<const f32x4*>(src + p * lds + c);
                if (rows) {
                    float w[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
                    for (int pl = 0; pl < PLANES; ++pl) {
                        h16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[e] = (h16)w[e]; w[e] -= (float)o[e]; }
                        *reinterpret_cast<h16x4*>(rows + p * ldr + pl * (ldr / PLANES) + c) = o;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r + 16 * i][(t & 15) * 4 + e] = v[e];
        }
    }
    __syncthreads();
    if (dst) {
        const int cl = t >> 4, p4 = (t & 15) * 4;
        if (p0 + p4 < Ppad)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + cl + 16 * i;
                if (c >= C) continue;
                float w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = tile[p4 + e][cl + 16 * i];
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    h16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (h16)w[e]; w[e] -= (float)o[e]; }
                    *reinterpret_cast<h16x4*>(dst + (int64_t)c * ldd + pl * (ldd / PLANES) + p0 + p4) = o;
                }
            }
    }
    if (part && t < 64 && c0 + t < C) {
        float sum = 0.f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) sum += tile[r][t];
        part[(int64_t)blockIdx.x * C + c0 + t] = sum;
    }
}

// ---------------------------------------------------------------------------------------------- column sums
// out[g][c] = sum over the rows r of group g (rows_per_group consecutive rows) of a[r][c] * (b ? b[r][c] : 1).
// One workgroup per (64 columns, group): four waves walk the rows 4 apart, their partials are folded in a fixed order.
__global__ __launch_bounds__(256) void group_colsum_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm, int64_t ldb,
                                                            int64_t rows_per_group, int cols, float* __restrict__ out) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_group;
    double s = 0.0;
    if (c < cols)
        for (int64_t r = r0 + wave; r < r0 + rows_per_group; r += 4) {
            const float a = A[r * lda + c];
            s += Bm ? (double)(a * Bm[r * ldb + c]) : (double)a;
        }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < cols) out[(int64_t)blockIdx.y * cols + c] = (float)(((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]);
}

// ---------------------------------------------------------------------------------------------- GroupNorm backward
// y = act(z), z = xhat gamma + beta, xhat = (x - mean) rstd per (sample, group); act = SiLU or identity.
//   dz = dy act'(z);  A[s][c] = sum_rows dz;  B[s][c] = sum_rows dz xhat
//   dgamma[c] = sum_s B,  dbeta[c] = sum_s A
//   dx = rstd (dz gamma - m1 - xhat m2),  m1 = sum_{c in g} gamma_c A_sc / n,  m2 = sum_{c in g} gamma_c B_sc / n,  n = cpg rows
// Pass 1 (one workgroup per (64 channels, sample, row chunk)) writes chunk partials of A and B; pass 2 folds the chunks in
// fp64 in a fixed order into AB[s][c][2] and the group terms m1 / m2; pass 3 writes dx.
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ dY, int64_t ldy,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ stat, int rows, int C, int groups, int nchunks, int silu,
                                                              float* __restrict__ part) {
    __shared__ float pa[4][64], pb[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, smp = blockIdx.y, chunk = blockIdx.z;
    const int rpc = (rows + nchunks - 1) / nchunks;
    const int r0 = chunk * rpc, r1 = (r0 + rpc < rows) ? r0 + rpc : rows;
    float a = 0.f, b = 0.f;
    if (c < C) {
        const int g = c / (C / groups);
        const float mean = stat[(smp * groups + g) * 2], rstd = stat[(smp * groups + g) * 2 + 1];
        const float ga = gamma[c], be = beta[c];
        for (int r = r0 + wave; r < r1; r += 4) {
            const int64_t row = (int64_t)smp * rows + r;
            const float xh = (X[row * ldx + c] - mean) * rstd;
            float dz = dY[row * ldy + c];
            if (silu) dz *= silu_grad(fmaf(xh, ga, be));
            a += dz; b = fmaf(dz, xh, b);
        }
    }
    pa[wave][lane] = a; pb[wave][lane] = b;
    __syncthreads();
    if (wave == 0 && c < C) {
        float* o = part + (((int64_t)smp * nchunks + chunk) * C + c) * 2;
        o[0] = ((pa[0][lane] + pa[1][lane]) + pa[2][lane]) + pa[3][lane];
        o[1] = ((pb[0][lane] + pb[1][lane]) + pb[2][lane]) + pb[3][lane];
    }
}

__global__ __launch_bounds__(256) void gn_bwd_fold_kernel(const float* __restrict__ part, const float* __restrict__ gamma, int C, int groups,
                                                           int nchunks, double count, float* __restrict__ AB, float* __restrict__ m12) {
    // one workgroup per (sample, group): thread c of the group folds its channel's chunks, then thread 0 folds the channels
    __shared__ double sa[256], sb[256];
    const int smp = blockIdx.x / groups, g = blockIdx.x - smp * groups;
    const int cpg = C / groups, t = threadIdx.x;
    double a = 0.0, b = 0.0;
    if (t < cpg) {
        const int c = g * cpg + t;
        for (int ch = 0; ch < nchunks; ++ch) {
            const float* p = part + (((int64_t)smp * nchunks + ch) * C + c) * 2;
            a += (double)p[0]; b += (double)p[1];
        }
        AB[((int64_t)smp * C + c) * 2] = (float)a;
        AB[((int64_t)smp * C + c) * 2 + 1] = (float)b;
        a *= (double)gamma[c]; b *= (double)gamma[c];
    }
    sa[t] = a; sb[t] = b;
    __syncthreads();
    if (t == 0) {
        double m1 = 0.0, m2 = 0.0;
        for (int i = 0; i < cpg; ++i) { m1 += sa[i]; m2 += sb[i]; }
        m12[(smp * groups + g) * 2] = (float)(m1 / count);
        m12[(smp * groups + g) * 2 + 1] = (float)(m2 / count);
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ dY, int64_t ldy,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ stat, const float* __restrict__ m12, int rows, int C,
                                                            int groups, int silu, float* __restrict__ dX, int64_t lddx, int64_t total,
                                                            const float* __restrict__ dRes, int64_t lddres) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t row = i / C;
    const int c = (int)(i - row * C);
    const int smp = (int)(row / rows), g = c / (C / groups);
    const float mean = stat[(smp * groups + g) * 2], rstd = stat[(smp * groups + g) * 2 + 1];
    const float res = dRes ? dRes[row * lddres + c] : 0.f;       // the gradient of the residual branch that bypassed the norm
    const float xh = (X[row * ldx + c] - mean) * rstd;
    float dz = dY[row * ldy + c];
    if (silu) dz *= silu_grad(fmaf(xh, gamma[c], beta[c]));
    dX[row * lddx + c] = rstd * (dz * gamma[c] - m12[(smp * groups + g) * 2] - xh * m12[(smp * groups + g) * 2 + 1]) + res;
}

// ---------------------------------------------------------------------------------------------- LayerNorm backward
// A workgroup owns LN_CHUNK consecutive rows, a wave every fourth of them: per row it recomputes mean / rstd and writes dx; the
// parameter-gradient terms dy xhat (dgamma) and dy (dbeta) are summed per lane over the wave's rows, then over the four waves
// in a fixed order, into part[chunk][2][C] — mudg_group_colsum folds the chunks.  Lane l owns columns l, l + 64, ... (C <= 64 LN_MAXC).
constexpr int LN_CHUNK = 64, LN_MAXC = 20;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ dY, int64_t ldy,
                                                      const float* __restrict__ gamma, float* __restrict__ dX, int64_t lddx,
                                                      float* __restrict__ part, int64_t rows, int C, float eps,
                                                      const float* __restrict__ dRes, int64_t lddres) {
    __shared__ float red[3][2][LN_MAXC * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = (C + 63) / 64;
    float ag[LN_MAXC], ab[LN_MAXC], ga[LN_MAXC];
#pragma unroll
    for (int j = 0; j < LN_MAXC; ++j) { ag[j] = 0.f; ab[j] = 0.f; ga[j] = (j < nc && lane + 64 * j < C) ? gamma[lane + 64 * j] : 0.f; }
    const int64_t r0 = (int64_t)blockIdx.x * LN_CHUNK;
    for (int64_t row = r0 + wave; row < r0 + LN_CHUNK && row < rows; row += 4) {
        const float* x = X + row * ldx;
        const float* dy = dY + row * ldy;
        float xv[LN_MAXC], dv[LN_MAXC], rv[LN_MAXC];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) {
            const bool ok = j < nc && lane + 64 * j < C;
            xv[j] = ok ? x[lane + 64 * j] : 0.f;
            dv[j] = ok ? dy[lane + 64 * j] : 0.f;
            rv[j] = (ok && dRes) ? dRes[row * lddres + lane + 64 * j] : 0.f;      // fetched with the rest: its latency hides under the reductions
            s += xv[j];
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) if (j < nc && lane + 64 * j < C) { const float d = xv[j] - mean; q = fmaf(d, d, q); }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) if (j < nc && lane + 64 * j < C) {
            const float dh = dv[j] * ga[j], xh = (xv[j] - mean) * rstd;
            m1 += dh; m2 = fmaf(dh, xh, m2);
        }
        m1 = wave_sum(m1) / (float)C; m2 = wave_sum(m2) / (float)C;
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) if (j < nc && lane + 64 * j < C) {
            const float xh = (xv[j] - mean) * rstd;
            dX[row * lddx + lane + 64 * j] = rstd * (dv[j] * ga[j] - m1 - xh * m2) + rv[j];     // + the gradient of the branch that bypassed the norm
            ag[j] = fmaf(dv[j], xh, ag[j]);
            ab[j] += dv[j];
        }
    }
    if (wave > 0)
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) if (j < nc) { red[wave - 1][0][j * 64 + lane] = ag[j]; red[wave - 1][1][j * 64 + lane] = ab[j]; }
    __syncthreads();
    if (wave == 0)
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) if (j < nc && lane + 64 * j < C) {
            const int i = j * 64 + lane;
            part[((int64_t)blockIdx.x * 2 + 0) * C + lane + 64 * j] = ((ag[j] + red[0][0][i]) + red[1][0][i]) + red[2][0][i];
            part[((int64_t)blockIdx.x * 2 + 1) * C + lane + 64 * j] = ((ab[j] + red[0][1][i]) + red[1][1][i]) + red[2][1][i];
        }
}

// ---------------------------------------------------------------------------------------------- GEGLU (unfused, training)
// H rows [M][2 N] = [value | gate] as the reference's chunk(2) lays them out (attention.py:579-586).
__global__ void geglu_fwd_kernel(const float* __restrict__ H, int64_t ldh, float* __restrict__ Y, int64_t ldy, int64_t M, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t m = i / N;
    const int n = (int)(i - m * N);
    Y[m * ldy + n] = H[m * ldh + n] * gelu_erf_f(H[m * ldh + N + n]);
}
__global__ void geglu_bwd_kernel(const float* __restrict__ H, int64_t ldh, const float* __restrict__ dY, int64_t ldy, float* __restrict__ dH,
                                 int64_t lddh, int64_t M, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t m = i / N;
    const int n = (int)(i - m * N);
    const float v = H[m * ldh + n], g = H[m * ldh + N + n], dy = dY[m * ldy + n];
    dH[m * lddh + n] = dy * gelu_erf_f(g);
    dH[m * lddh + N + n] = dy * v * gelu_grad(g);
}

// GEGLU followed by the feed-forward's Dropout (attention.py:579-606: GEGLU -> Dropout -> Linear), four output columns per thread:
// forward  Y = keep(value * gelu(gate)) / (1 - p), written as fp32 rows and (optionally) as the operand rows the next GEMM reads;
// backward dH = [dy' gelu(gate) | dy' value gelu'(gate)] with dy' = keep(dY) / (1 - p).  The keep mask is the counter-based one of
// dropout_kernel on the output element index (regenerated, never stored); p = 0: plain GEGLU.
__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t i, float p, float inv) {
    uint64_t z = seed + (uint64_t)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f) >= p ? inv : 0.f;
}
__global__ __launch_bounds__(256) void geglu_drop_fwd_kernel(const float* __restrict__ H, int64_t ldh, float* __restrict__ Y, int64_t ldy,
                                                              h16* __restrict__ Y16, int64_t ldy16, int64_t M, int N, float p, uint64_t seed) {
    const int nv = N >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * nv) return;
    const int64_t m = i / nv;
    const int n = (int)(i - m * nv) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(H + m * ldh + n), g = *reinterpret_cast<const f32x4*>(H + m * ldh + N + n);
    const float inv = 1.0f / (1.0f - p);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        y[e] = v[e] * gelu_erf_f(g[e]);
        if (p > 0.f) y[e] *= keep_scale(seed, m * N + n + e, p, inv);
    }
    *reinterpret_cast<f32x4*>(Y + m * ldy + n) = y;
    if (Y16) {
        float w[4] = {y[0], y[1], y[2], y[3]};
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            h16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (h16)w[e]; w[e] -= (float)o[e]; }
            *reinterpret_cast<h16x4*>(Y16 + m * ldy16 + pl * (ldy16 / PLANES) + n) = o;
        }
    }
}
__global__ __launch_bounds__(256) void geglu_drop_bwd_kernel(const float* __restrict__ H, int64_t ldh, const float* __restrict__ dY, int64_t lddy,
                                                              float* __restrict__ dH, int64_t lddh, int64_t M, int N, float p, uint64_t seed) {
    const int nv = N >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * nv) return;
    const int64_t m = i / nv;
    const int n = (int)(i - m * nv) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(H + m * ldh + n), g = *reinterpret_cast<const f32x4*>(H + m * ldh + N + n);
    f32x4 dy = *reinterpret_cast<const f32x4*>(dY + m * lddy + n);
    const float inv = 1.0f / (1.0f - p);
    f32x4 dv, dg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (p > 0.f) dy[e] *= keep_scale(seed, m * N + n + e, p, inv);
        dv[e] = dy[e] * gelu_erf_f(g[e]);
        dg[e] = dy[e] * v[e] * gelu_grad(g[e]);
    }
    *reinterpret_cast<f32x4*>(dH + m * lddh + n) = dv;
    *reinterpret_cast<f32x4*>(dH + m * lddh + N + n) = dg;
}

// ---------------------------------------------------------------------------------------------- softmax (attention recompute)
// P = softmax(S) row-wise, fp32 in place-capable; dS = scale * P (dP - sum_j dP_j P_j).  One workgroup per row.
__global__ __launch_bounds__(256) void softmax_f32_kernel(const float* __restrict__ S, int64_t lds, float* __restrict__ P, int64_t ldp, int cols) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = S + (int64_t)blockIdx.x * lds;
    float* p = P + (int64_t)blockIdx.x * ldp;
    float mx = -INFINITY;
    for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, s[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = tid; c < cols; c += 256) sum += expf(s[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.f / (((red[0] + red[1]) + red[2]) + red[3]);
    for (int c = tid; c < cols; c += 256) p[c] = expf(s[c] - mx) * inv;
}
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, int64_t ldp, const float* __restrict__ dP, int64_t lddp,
                                                           float* __restrict__ dS, int64_t ldds, int cols, float scale) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = P + (int64_t)blockIdx.x * ldp;
    const float* dp = dP + (int64_t)blockIdx.x * lddp;
    float* ds = dS + (int64_t)blockIdx.x * ldds;
    float dot = 0.f;
    for (int c = tid; c < cols; c += 256) dot = fmaf(p[c], dp[c], dot);
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    dot = ((red[0] + red[1]) + red[2]) + red[3];
    for (int c = tid; c < cols; c += 256) ds[c] = scale * p[c] * (dp[c] - dot);
}

// ---------------------------------------------------------------------------------------------- temporal attention backward
// Per (pixel, head): T <= 32 tokens of width 64, rows ((b T + t) HW + s).  fp32 in, fp32 out; everything of one item lives in
// LDS.  dQ = scale dS K, dK = scale dS^T Q, dV = P^T dO with P = softmax(scale Q K^T), dS = P (dP - rowsum(dP P)), dP = dO V^T.
template <int TMAX>
__global__ __launch_bounds__(64) void tattn_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                        const float* __restrict__ dO, int64_t ldq, int64_t ldo, float* __restrict__ dQ,
                                                        float* __restrict__ dK, float* __restrict__ dV, int64_t ldg, int T, int HW, int heads,
                                                        float scale) {
    __shared__ float q[TMAX][65], k[TMAX][65], v[TMAX][65], go[TMAX][65], p[TMAX][TMAX + 1], ds[TMAX][TMAX + 1];
    const int lane = threadIdx.x;
    const int64_t item = blockIdx.x;                       // (b, s, h)
    const int h = (int)(item % heads);
    const int64_t bs = item / heads;
    const int s = (int)(bs % HW);
    const int64_t b = bs / HW;
    auto row = [&](int t) { return (b * T + t) * HW + s; };
    for (int t = 0; t < T; ++t) {
        q[t][lane] = Q[row(t) * ldq + h * 64 + lane];
        k[t][lane] = K[row(t) * ldq + h * 64 + lane];
        v[t][lane] = V[row(t) * ldq + h * 64 + lane];
        go[t][lane] = dO[row(t) * ldo + h * 64 + lane];
    }
    __syncthreads();
    for (int e = lane; e < T * T; e += 64) {                // scores and dP
        const int i = e / T, j = e - i * T;
        float a = 0.f, d = 0.f;
        for (int c = 0; c < 64; ++c) { a = fmaf(q[i][c], k[j][c], a); d = fmaf(go[i][c], v[j][c], d); }
        p[i][j] = a * scale; ds[i][j] = d;
    }
    __syncthreads();
    if (lane < T) {                                          // softmax and dS, one row per lane
        const int i = lane;
        float mx = -INFINITY;
        for (int j = 0; j < T; ++j) mx = fmaxf(mx, p[i][j]);
        float sum = 0.f;
        for (int j = 0; j < T; ++j) { p[i][j] = expf(p[i][j] - mx); sum += p[i][j]; }
        float dot = 0.f;
        for (int j = 0; j < T; ++j) { p[i][j] /= sum; dot = fmaf(p[i][j], ds[i][j], dot); }
        for (int j = 0; j < T; ++j) ds[i][j] = scale * p[i][j] * (ds[i][j] - dot);
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {                            // lane = head dim
        float gq = 0.f, gk = 0.f, gv = 0.f;
        for (int j = 0; j < T; ++j) {
            gq = fmaf(ds[t][j], k[j][lane], gq);
            gk = fmaf(ds[j][t], q[j][lane], gk);
            gv = fmaf(p[j][t], go[j][lane], gv);
        }
        dQ[row(t) * ldg + h * 64 + lane] = gq;
        dK[row(t) * ldg + h * 64 + lane] = gk;
        dV[row(t) * ldg + h * 64 + lane] = gv;
    }
}

// ---------------------------------------------------------------------------------------------- loss, resampling, optimiser
// Weighted MSE of ddpm3d.py:766-787: per-sample mean of (pred - target)^2 (deterministic two-stage sum) and the gradient
// w[b] 2 (pred - target) / n of sum_b w[b] mse_b.
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n,
                                                           int nblk, double* __restrict__ part) {
    __shared__ double red[4];
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const float* pp = pred + (int64_t)b * n;
    const float* tt = target + (int64_t)b * n;
    double s = 0.0;
    for (int64_t i = (int64_t)blk * 256 + tid; i < n; i += (int64_t)nblk * 256) { const float d = pp[i] - tt[i]; s += (double)(d * d); }
    s = wave_sum_d(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) part[(int64_t)b * nblk + blk] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ void mse_finish_kernel(const double* __restrict__ part, int nblk, int64_t n, float* __restrict__ loss, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += part[(int64_t)b * nblk + i];
    loss[b] = (float)(s / (double)n);
}
__global__ void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ w, int64_t n,
                                float* __restrict__ grad, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    grad[i] = w[i / n] * 2.0f * (pred[i] - target[i]) / (float)n;
}

// nearest-2x of channels-last rows (F, h, w, C) -> (F, 2h, 2w, C) and its adjoint (sum of the 2 x 2 block).
__global__ void upsample2x_kernel(const float* __restrict__ src, float* __restrict__ dst, int F, int h, int w, int C, int adjoint) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)F * h * w * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int64_t f = r / h;
    const int64_t hi00 = ((f * 2 * h + 2 * y) * 2 * w + 2 * x) * C + c;
    const int64_t dxs = C, dys = (int64_t)2 * w * C;
    if (adjoint) dst[i] = (src[hi00] + src[hi00 + dxs]) + (src[hi00 + dys] + src[hi00 + dys + dxs]);
    else { const float v = src[i]; dst[hi00] = v; dst[hi00 + dxs] = v; dst[hi00 + dys] = v; dst[hi00 + dys + dxs] = v; }
}
// Zero insertion: (F, ho, wo, C) gradient of a stride-2 conv's output laid onto the (F, hi, wi, C) input grid at (2 oy, 2 ox).
__global__ void dilate2x_kernel(const float* __restrict__ src, float* __restrict__ dst, int F, int ho, int wo, int hi, int wi, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)F * hi * wi * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int x = (int)(r % wi); r /= wi;
    const int y = (int)(r % hi);
    const int64_t f = r / hi;
    float v = 0.f;
    if (!(x & 1) && !(y & 1) && (y >> 1) < ho && (x >> 1) < wo) v = src[((f * ho + (y >> 1)) * wo + (x >> 1)) * C + c];
    dst[i] = v;
}

// AdamW (torch.optim.AdamW semantics, decoupled weight decay): fp32 parameters, gradients and moments, in place.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

// The same update over MANY parameter tensors in one launch (the UNet has 1520: one launch each was 2.4 % of a training step).
// `table` lists chunks of at most CLIP_CHUNK values as rows (p, g, m, v, count); one workgroup per chunk.
__global__ __launch_bounds__(256) void adamw_multi_kernel(const int64_t* __restrict__ table, float lr, float b1, float b2, float eps, float wd,
                                                           float bc1, float bc2) {
    const int64_t* row = table + 5 * (int64_t)blockIdx.x;
    float* p = reinterpret_cast<float*>(row[0]);
    const float* g = reinterpret_cast<const float*>(row[1]);
    float* m = reinterpret_cast<float*>(row[2]);
    float* v = reinterpret_cast<float*>(row[3]);
    const int n = (int)row[4];
    const float step = lr / bc1, decay = 1.0f - lr * wd;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = g[i];
        const float pi = p[i] * decay;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = pi - step * (mi / denom);
    }
}

// Global gradient norm and clipping over many tensors without a host round trip (torch.nn.utils.clip_grad_norm_ semantics: the
// reference's trainer clips to norm 0.5).  `table` lists chunks of at most CLIP_CHUNK fp32 values as (address, count) pairs; one
// workgroup sums the squares of a chunk in fp64 (fixed order), one workgroup then folds the chunk sums in order and writes
// out[0] = the norm, out[1] = min(1, max_norm / (norm + 1e-6)); the scale pass multiplies every chunk by out[1].
constexpr int CLIP_CHUNK = 16384;
__global__ __launch_bounds__(256) void chunk_sumsq_kernel(const int64_t* __restrict__ table, double* __restrict__ partial) {
    __shared__ double red[256];
    const float* x = reinterpret_cast<const float*>(table[2 * blockIdx.x]);
    const int n = (int)table[2 * blockIdx.x + 1];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const double v = (double)x[i]; s += v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void norm_fold_kernel(const double* __restrict__ partial, int nchunks, float max_norm, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nchunks; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        out[0] = norm;
        const float coef = max_norm / (norm + 1e-6f);
        out[1] = coef < 1.0f ? coef : 1.0f;
    }
}
__global__ __launch_bounds__(256) void chunk_scale_kernel(const int64_t* __restrict__ table, const float* __restrict__ out) {
    const float coef = out[1];
    if (coef == 1.0f) return;
    float* x = reinterpret_cast<float*>(table[2 * blockIdx.x]);
    const int n = (int)table[2 * blockIdx.x + 1];
    for (int i = threadIdx.x; i < n; i += 256) x[i] *= coef;
}

// Exact (erf) GELU of the Perceiver feed-forward (resampler.py:27-34) and its derivative Phi(x) + x phi(x).
__global__ void gelu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float xi = x[i];
    const float cdf = 0.5f * (1.0f + erff(xi * 0.70710678118654752440f));
    out[i] = dy ? dy[i] * (cdf + xi * 0.39894228040143267794f * expf(-0.5f * xi * xi)) : xi * cdf;
}

__global__ void silu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = dy ? dy[i] * silu_grad(x[i]) : x[i] / (1.0f + expf(-x[i]));
}

// (mean, rstd) per (sample, group) of fp32 rows — the statistics the GroupNorm forward used, for the backward pass.
__global__ __launch_bounds__(256) void gn_stat_f32_kernel(const float* __restrict__ X, int64_t ldx, int rows, int C, int groups, float eps,
                                                           float* __restrict__ stat) {
    __shared__ double ra[4], rb[4];
    const int smp = blockIdx.x / groups, g = blockIdx.x - smp * groups, cpg = C / groups, tid = threadIdx.x;
    double a = 0.0, b = 0.0;
    const int64_t n = (int64_t)rows * cpg;
    for (int64_t i = tid; i < n; i += 256) {
        const int64_t r = i / cpg;
        const float x = X[((int64_t)smp * rows + r) * ldx + g * cpg + (int)(i - r * cpg)];
        a += (double)x; b += (double)x * (double)x;
    }
    a = wave_sum_d(a); b = wave_sum_d(b);
    if ((tid & 63) == 0) { ra[tid >> 6] = a; rb[tid >> 6] = b; }
    __syncthreads();
    if (tid == 0) {
        a = ((ra[0] + ra[1]) + ra[2]) + ra[3]; b = ((rb[0] + rb[1]) + rb[2]) + rb[3];
        const double mean = a / (double)n;
        double var = b / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[blockIdx.x * 2] = (float)mean;
        stat[blockIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
// Inverted dropout with a counter-based mask: keep(i) depends on (seed, i) only, so the backward pass regenerates the mask
// instead of storing it.  splitmix64 finaliser; 24 random bits against the keep probability.
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float p, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = x[i] * keep_scale(seed, i, p, 1.0f / (1.0f - p));
}
// The same on rows [M][C] (C % 4 == 0), four columns per thread, with the operand rows of the result written alongside (the conv /
// projection that follows reads those): element index = m C + c, as in dropout_kernel on the flattened contiguous rows.
__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                            h16* __restrict__ Y16, int64_t ldy16, int64_t M, int C, float p, uint64_t seed) {
    const int cv = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * cv) return;
    const int64_t m = i / cv;
    const int c = (int)(i - m * cv) * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(X + m * ldx + c);
    const float inv = 1.0f / (1.0f - p);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = x[e] * keep_scale(seed, m * C + c + e, p, inv);
    *reinterpret_cast<f32x4*>(Y + m * ldy + c) = y;
    if (Y16) {
        float w[4] = {y[0], y[1], y[2], y[3]};
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) {
            h16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (h16)w[e]; w[e] -= (float)o[e]; }
            *reinterpret_cast<h16x4*>(Y16 + m * ldy16 + pl * (ldy16 / PLANES) + c) = o;
        }
    }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int mudg_transpose_gather(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t P, int C, int mode, int Hin, int Win,
                          int Hout, int Wout, int stride, int pad, int dy, int dx, int T, int HW, int dt, int batch, int64_t src_batch_stride,
                          int64_t dst_batch_stride, void* stream) {
    MUDG_REQUIRE(src && dst && P > 0 && C > 0 && batch >= 1 && batch <= 65535, "mudg_transpose_gather: bad arguments");
    MUDG_REQUIRE(mode >= 0 && mode <= 2, "mudg_transpose_gather: mode %d", mode);
    const int64_t Ppad = (P + 7) / 8 * 8;
    MUDG_REQUIRE(ldd % PLANES == 0 && ldd / PLANES >= Ppad, "mudg_transpose_gather: ldd=%lld too small for %lld columns", (long long)ldd, (long long)Ppad);
    if (mode == 1) MUDG_REQUIRE(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && stride > 0 && P % ((int64_t)Hout * Wout) == 0, "mudg_transpose_gather: conv geometry");
    if (mode == 2) MUDG_REQUIRE(T > 0 && HW > 0 && P % ((int64_t)T * HW) == 0, "mudg_transpose_gather: temporal geometry");
    GatherGeo g{mode, Hin, Win, Hout, Wout, stride, pad, dy, dx, T, HW, dt};
    hipLaunchKernelGGL(transpose_gather_kernel, dim3((unsigned)((Ppad + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)batch), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), src, lds, (h16*)dst, ldd, P, Ppad, C, g, src_batch_stride, dst_batch_stride);
    return mudg_check_launch("mudg_transpose_gather");
}

int mudg_group_colsum(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t rows, int cols, int64_t rows_per_group, float* out,
                      void* stream) {
    MUDG_REQUIRE(A && out && rows > 0 && cols > 0 && rows_per_group > 0 && rows % rows_per_group == 0, "mudg_group_colsum: bad arguments");
    hipLaunchKernelGGL(group_colsum_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)(rows / rows_per_group)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), A, lda, B, ldb, rows_per_group, cols, out);
    return mudg_check_launch("mudg_group_colsum");
}

int64_t mudg_groupnorm_bwd_ws_floats(int samples, int rows, int C, int groups) {
    int nch = rows / 256; if (nch < 1) nch = 1; if (nch > 256) nch = 256;
    return (int64_t)samples * nch * C * 2 + (int64_t)samples * groups * 2;
}

/* stat: (mean, rstd) per (sample, group) as the forward pass left them; AB: fp32 [samples][C][2] out (dbeta / dgamma before the
 * sum over samples). */
int mudg_groupnorm_bwd(const float* X, int64_t ldx, const float* dY, int64_t ldy, const float* gamma, const float* beta, const float* stat,
                       int samples, int rows, int C, int groups, int silu, float* dX, int64_t lddx, float* AB, float* ws, const float* dres,
                       int64_t lddres, void* stream) {
    MUDG_REQUIRE(X && dY && gamma && beta && stat && dX && AB && ws, "mudg_groupnorm_bwd: null pointer");
    MUDG_REQUIRE(samples > 0 && rows > 0 && C > 0 && groups > 0 && C % groups == 0 && C / groups <= 256 && samples <= 65535, "mudg_groupnorm_bwd: shape");
    int nch = rows / 256; if (nch < 1) nch = 1; if (nch > 256) nch = 256;
    float* part = ws;
    float* m12 = ws + (int64_t)samples * nch * C * 2;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)samples, (unsigned)nch), dim3(256), 0, s, X, ldx, dY, ldy,
                       gamma, beta, stat, rows, C, groups, nch, silu, part);
    hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3((unsigned)(samples * groups)), dim3(256), 0, s, part, gamma, C, groups, nch,
                       (double)rows * (C / groups), AB, m12);
    const int64_t total = (int64_t)samples * rows * C;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(blocks_for(total)), dim3(256), 0, s, X, ldx, dY, ldy, gamma, beta, stat, m12, rows, C, groups, silu,
                       dX, lddx, total, dres, lddres);
    return mudg_check_launch("mudg_groupnorm_bwd");
}

int mudg_groupnorm_stats(const float* X, int64_t ldx, int samples, int rows, int C, int groups, float eps, float* stat, void* stream) {
    MUDG_REQUIRE(X && stat && samples > 0 && rows > 0 && C > 0 && groups > 0 && C % groups == 0, "mudg_groupnorm_stats: bad arguments");
    hipLaunchKernelGGL(gn_stat_f32_kernel, dim3((unsigned)(samples * groups)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, ldx, rows, C,
                       groups, eps, stat);
    return mudg_check_launch("mudg_groupnorm_stats");
}

int64_t mudg_layernorm_bwd_chunks(int64_t rows) { return (rows + LN_CHUNK - 1) / LN_CHUNK; }

int mudg_layernorm_bwd(const float* X, int64_t ldx, const float* dY, int64_t ldy, const float* gamma, float* dX, int64_t lddx, float* part,
                       int64_t rows, int C, float eps, const float* dres, int64_t lddres, void* stream) {
    MUDG_REQUIRE(X && dY && gamma && dX && part && rows > 0 && C > 0, "mudg_layernorm_bwd: bad arguments");
    MUDG_REQUIRE(C <= 64 * LN_MAXC, "mudg_layernorm_bwd: C=%d above %d", C, 64 * LN_MAXC);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)mudg_layernorm_bwd_chunks(rows)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, ldx,
                       dY, ldy, gamma, dX, lddx, part, rows, C, eps, dres, lddres);
    return mudg_check_launch("mudg_layernorm_bwd");
}

int mudg_transpose_cast_sum(const float* src, int64_t lds, void* dst, int64_t ldd, void* rows, int64_t ldr, float* part, int64_t P, int C,
                            void* stream) {
    MUDG_REQUIRE(src && (dst || rows || part) && P > 0 && C > 0 && (C & 3) == 0 && (lds & 3) == 0, "mudg_transpose_cast_sum: bad arguments (C and the row stride must be multiples of 4)");
    const int64_t Ppad = (P + 7) / 8 * 8;
    MUDG_REQUIRE(!dst || (ldd % PLANES == 0 && ldd / PLANES >= Ppad && ((ldd / PLANES) & 3) == 0), "mudg_transpose_cast_sum: ldd=%lld too small for %lld columns", (long long)ldd, (long long)Ppad);
    MUDG_REQUIRE(!rows || (ldr % PLANES == 0 && ldr / PLANES >= C && ((ldr / PLANES) & 3) == 0), "mudg_transpose_cast_sum: ldr=%lld", (long long)ldr);
    MUDG_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15u) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0 && (reinterpret_cast<uintptr_t>(rows) & 7u) == 0,
                 "mudg_transpose_cast_sum: alignment");
    hipLaunchKernelGGL(xpose_cast_sum_kernel, dim3((unsigned)((Ppad + 63) / 64), (unsigned)((C + 63) / 64)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), src, lds, (h16*)dst, ldd, (h16*)rows, ldr, part, P, Ppad, C);
    return mudg_check_launch("mudg_transpose_cast_sum");
}

int mudg_geglu(const float* H, int64_t ldh, const float* dY, int64_t lddy, float* out, int64_t ldo, int64_t M, int N, void* stream) {
    MUDG_REQUIRE(H && out && M > 0 && N > 0, "mudg_geglu: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dY) hipLaunchKernelGGL(geglu_bwd_kernel, dim3(blocks_for(M * N)), dim3(256), 0, s, H, ldh, dY, lddy, out, ldo, M, N);
    else hipLaunchKernelGGL(geglu_fwd_kernel, dim3(blocks_for(M * N)), dim3(256), 0, s, H, ldh, out, ldo, M, N);
    return mudg_check_launch("mudg_geglu");
}

int mudg_geglu_dropout(const float* H, int64_t ldh, const float* dY, int64_t lddy, float* out, int64_t ldo, void* out16, int64_t ldo16,
                       int64_t M, int N, float p, uint64_t seed, void* stream) {
    MUDG_REQUIRE(H && out && M > 0 && N > 0 && (N & 3) == 0 && p >= 0.f && p < 1.f, "mudg_geglu_dropout: bad arguments (N must be a multiple of 4)");
    MUDG_REQUIRE((ldh & 3) == 0 && (ldo & 3) == 0 && (!dY || (lddy & 3) == 0) && aligned16(H) && aligned16(out) && aligned16(dY),
                 "mudg_geglu_dropout: row strides must be multiples of 4 floats and the bases 16-byte aligned");
    MUDG_REQUIRE(!out16 || (!dY && ldo16 % PLANES == 0 && ldo16 / PLANES >= N && ((ldo16 / PLANES) & 3) == 0 && (reinterpret_cast<uintptr_t>(out16) & 7u) == 0),
                 "mudg_geglu_dropout: operand output");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned blocks = blocks_for(M * (N / 4));
    if (dY) hipLaunchKernelGGL(geglu_drop_bwd_kernel, dim3(blocks), dim3(256), 0, s, H, ldh, dY, lddy, out, ldo, M, N, p, seed);
    else hipLaunchKernelGGL(geglu_drop_fwd_kernel, dim3(blocks), dim3(256), 0, s, H, ldh, out, ldo, (h16*)out16, ldo16, M, N, p, seed);
    return mudg_check_launch("mudg_geglu_dropout");
}

int mudg_softmax_f32(const float* S, int64_t lds, float* P, int64_t ldp, int64_t rows, int cols, void* stream) {
    MUDG_REQUIRE(S && P && rows > 0 && cols > 0, "mudg_softmax_f32: bad arguments");
    hipLaunchKernelGGL(softmax_f32_kernel, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S, lds, P, ldp, cols);
    return mudg_check_launch("mudg_softmax_f32");
}
int mudg_softmax_bwd(const float* P, int64_t ldp, const float* dP, int64_t lddp, float* dS, int64_t ldds, int64_t rows, int cols, float scale,
                     void* stream) {
    MUDG_REQUIRE(P && dP && dS && rows > 0 && cols > 0, "mudg_softmax_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P, ldp, dP, lddp, dS, ldds, cols,
                       scale);
    return mudg_check_launch("mudg_softmax_bwd");
}

int mudg_temporal_attention_bwd(const float* Q, const float* K, const float* V, const float* dO, int64_t ldqkv, int64_t ldo, float* dQ, float* dK,
                                float* dV, int64_t ldg, int B, int T, int HW, int heads, float scale, void* stream) {
    MUDG_REQUIRE(Q && K && V && dO && dQ && dK && dV && B > 0 && T > 0 && T <= 32 && HW > 0 && heads > 0, "mudg_temporal_attention_bwd: bad arguments");
    const int64_t items = (int64_t)B * HW * heads;
    if (T <= 16) hipLaunchKernelGGL(tattn_bwd_kernel<16>, dim3((unsigned)items), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), Q, K, V, dO, ldqkv,
                                    ldo, dQ, dK, dV, ldg, T, HW, heads, scale);
    else hipLaunchKernelGGL(tattn_bwd_kernel<32>, dim3((unsigned)items), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), Q, K, V, dO, ldqkv, ldo, dQ,
                            dK, dV, ldg, T, HW, heads, scale);
    return mudg_check_launch("mudg_temporal_attention_bwd");
}

int64_t mudg_mse_ws_doubles(int B) { return (int64_t)B * 256; }
int mudg_mse(const float* pred, const float* target, const float* w, int B, int64_t n, float* loss, float* grad, double* ws, void* stream) {
    MUDG_REQUIRE(pred && target && loss && ws && B > 0 && n > 0, "mudg_mse: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nblk = 256;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(nblk, (unsigned)B), dim3(256), 0, s, pred, target, n, nblk, ws);
    hipLaunchKernelGGL(mse_finish_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, ws, nblk, n, loss, B);
    if (grad) {
        MUDG_REQUIRE(w, "mudg_mse: the gradient needs the per-sample weights");
        hipLaunchKernelGGL(mse_grad_kernel, dim3(blocks_for((int64_t)B * n)), dim3(256), 0, s, pred, target, w, n, grad, (int64_t)B * n);
    }
    return mudg_check_launch("mudg_mse");
}

int mudg_upsample2x(const float* src, float* dst, int F, int h, int w, int C, int adjoint, void* stream) {
    MUDG_REQUIRE(src && dst && F > 0 && h > 0 && w > 0 && C > 0, "mudg_upsample2x: bad arguments");
    hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks_for((int64_t)F * h * w * C)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, F, h, w,
                       C, adjoint);
    return mudg_check_launch("mudg_upsample2x");
}
int mudg_dilate2x(const float* src, float* dst, int F, int ho, int wo, int hi, int wi, int C, void* stream) {
    MUDG_REQUIRE(src && dst && F > 0 && ho > 0 && wo > 0 && hi > 0 && wi > 0 && C > 0, "mudg_dilate2x: bad arguments");
    hipLaunchKernelGGL(dilate2x_kernel, dim3(blocks_for((int64_t)F * hi * wi * C)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, F, ho,
                       wo, hi, wi, C);
    return mudg_check_launch("mudg_dilate2x");
}

int mudg_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
               void* stream) {
    MUDG_REQUIRE(p && g && m && v && n > 0 && step > 0, "mudg_adamw: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2);
    return mudg_check_launch("mudg_adamw");
}

int mudg_adamw_multi(const int64_t* table, int nchunks, float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
    MUDG_REQUIRE(table && nchunks > 0 && step > 0, "mudg_adamw_multi: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2);
    return mudg_check_launch("mudg_adamw_multi");
}

int mudg_clip_chunk(void) { return CLIP_CHUNK; }

int mudg_clip_grad_norm(const int64_t* table, int nchunks, double* partial, float max_norm, float* out, void* stream) {
    MUDG_REQUIRE(table && partial && out && nchunks > 0 && max_norm > 0.f, "mudg_clip_grad_norm: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(chunk_sumsq_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, table, partial);
    hipLaunchKernelGGL(norm_fold_kernel, dim3(1), dim3(256), 0, s, partial, nchunks, max_norm, out);
    hipLaunchKernelGGL(chunk_scale_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, table, out);
    return mudg_check_launch("mudg_clip_grad_norm");
}

int mudg_dropout(const float* x, float* out, int64_t n, float p, uint64_t seed, void* stream) {
    MUDG_REQUIRE(x && out && n > 0 && p >= 0.f && p < 1.f, "mudg_dropout: bad arguments");
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out, n, p, seed);
    return mudg_check_launch("mudg_dropout");
}

int mudg_dropout_rows(const float* X, int64_t ldx, float* Y, int64_t ldy, void* Y16, int64_t ldy16, int64_t M, int C, float p, uint64_t seed,
                      void* stream) {
    MUDG_REQUIRE(X && Y && M > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && p >= 0.f && p < 1.f && aligned16(X) && aligned16(Y),
                 "mudg_dropout_rows: bad arguments (C and the row strides must be multiples of 4)");
    MUDG_REQUIRE(!Y16 || (ldy16 % PLANES == 0 && ldy16 / PLANES >= C && ((ldy16 / PLANES) & 3) == 0 && (reinterpret_cast<uintptr_t>(Y16) & 7u) == 0),
                 "mudg_dropout_rows: operand output");
    hipLaunchKernelGGL(dropout_rows_kernel, dim3(blocks_for(M * (C / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, ldx, Y, ldy,
                       (h16*)Y16, ldy16, M, C, p, seed);
    return mudg_check_launch("mudg_dropout_rows");
}

int mudg_gelu(const float* x, const float* dy, float* out, int64_t n, void* stream) {
    MUDG_REQUIRE(x && out && n > 0, "mudg_gelu: bad arguments");
    hipLaunchKernelGGL(gelu_kernel, dim3(blocks_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, dy, out, n);
    return mudg_check_launch("mudg_gelu");
}

int mudg_silu(const float* x, const float* dy, float* out, int64_t n, void* stream) {
    MUDG_REQUIRE(x && out && n > 0, "mudg_silu: bad arguments");
    hipLaunchKernelGGL(silu_kernel, dim3(blocks_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, dy, out, n);
    return mudg_check_launch("mudg_silu");
}

}  // extern "C"
