// misc.hip — the small kernels around the UNet: sinusoidal embeddings, the tiny fp32 MLPs on them, the
// (b c t h w) <-> channels-last conversions at the API boundary and the fused DDIM update.
#include "common.h"

namespace {

__global__ void temb_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out, int n,
                            int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= n * half) return;
    const int r = i / half, j = i - r * half;
    // freqs[j] = exp(-log(max_period) * j / half) is a host-made table (utils_diffusion.py:19-22 builds it on the CPU
    // too); the product with the timestep and the cos/sin happen here, as they do on the device in the reference.
    const float a = (float)t[r] * freqs[j];
    out[(int64_t)r * dim + j] = cosf(a);
    out[(int64_t)r * dim + half + j] = sinf(a);
    if ((dim & 1) && j == 0) out[(int64_t)r * dim + dim - 1] = 0.f;
}

template <typename WT>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, const WT* __restrict__ W,
                                                            const float* __restrict__ b, float* __restrict__ y, int M, int N,
                                                            int K, int act_in, int act_out, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m = blockIdx.y;
    if (n >= N) return;
    const float* xr = x + (int64_t)m * K;
    const WT* wr = W + (int64_t)n * K;
    float a = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv = xr[k];
        if (act_in) xv = silu_f(xv);
        a = fmaf(xv, (float)wr[k], a);
    }
    a = wave_sum(a);
    if (lane == 0) {
        if (b) a += b[n];
        if (act_out) a = silu_f(a);
        float* o = y + (int64_t)m * N + n;
        *o = accumulate ? *o + a : a;
    }
}

template <typename ST>
__global__ void ncthw_to_rows_kernel(const ST* __restrict__ src, h16* __restrict__ dst, int B, int C, int T, int HW,
                                     int ld, int coff, int Ttot, int t0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over (b, t, p)
    const int64_t n = (int64_t)B * T * HW;
    if (i >= n) return;
    const int p = (int)(i % HW);
    const int64_t bt = i / HW;
    const int t = (int)(bt % T), b = (int)(bt / T);
    for (int c = 0; c < C; ++c)
        store1_operand(dst + i * ld + coff + c, ld / PLANES, (float)src[(((int64_t)b * C + c) * Ttot + t0 + t) * HW + p]);
}

// A rows-matrix element as fp32: operand storage (PLANES pieces) or plain fp32.
__device__ __forceinline__ float row_value(const h16* p, int ld) { return load1_operand(p, ld / PLANES); }
__device__ __forceinline__ float row_value(const float* p, int) { return *p; }
struct StreamH { _Float16 v; };       // fp16 residual-stream storage (KIND_F16); a type of its own since h16 may be _Float16 too
__device__ __forceinline__ float row_value(const StreamH* p, int) { return (float)p->v; }

template <typename ST, typename DT>
__global__ void rows_to_ncthw_kernel(const ST* __restrict__ src, int ld, int coff, DT* __restrict__ dst, int B, int C,
                                     int T, int HW, float scale, int Ttot, int t0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)B * T * HW;
    if (i >= n) return;
    const int p = (int)(i % HW);
    const int64_t bt = i / HW;
    const int t = (int)(bt % T), b = (int)(bt / T);
    for (int c = 0; c < C; ++c)
        dst[(((int64_t)b * C + c) * Ttot + t0 + t) * HW + p] = (DT)(row_value(src + i * ld + coff + c, ld) * scale);
}

__global__ void zero_channels_kernel(h16* __restrict__ dst, int64_t rows, int ld, int c0, int c1) {
    const int w = c1 - c0;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
#pragma unroll
    for (int p = 0; p < PLANES; ++p) dst[(i / w) * ld + p * (ld / PLANES) + c0 + (int)(i % w)] = (h16)0.f;
}

__global__ void copy_rows_kernel(const h16* __restrict__ src, int64_t lds, h16* __restrict__ dst, int64_t ldd, int64_t rows,
                                 int64_t cols, int vec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t cv = cols >> 3;
        if (i >= rows * cv) return;
        const int64_t r = i / cv, c = (i - r * cv) << 3;
#pragma unroll
        for (int p = 0; p < PLANES; ++p) st16(dst + r * ldd + p * (ldd / PLANES) + c, ld16(src + r * lds + p * (lds / PLANES) + c));
    } else {
        if (i >= rows * cols) return;
        const int64_t r = i / cols, c = i - r * cols;
#pragma unroll
        for (int p = 0; p < PLANES; ++p) dst[r * ldd + p * (ldd / PLANES) + c] = src[r * lds + p * (lds / PLANES) + c];
    }
}

// dst[r][c] = src[r][c] between rows matrices of any storage kind, any direction.  VEC: 8 consecutive columns per thread
// with 16-byte accesses (cols, both row strides and both bases 8-element aligned).
template <typename T> struct Row8;
template <> struct Row8<h16> {
    static __device__ __forceinline__ void get(const h16* p, int64_t ld, float (&v)[8]) { load8_operand(p, ld / PLANES, v); }
    static __device__ __forceinline__ void put(h16* p, int64_t ld, const float (&v)[8]) { store8_operand(p, ld / PLANES, v); }
};
template <> struct Row8<float> {
    static __device__ __forceinline__ void get(const float* p, int64_t, float (&v)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    }
    static __device__ __forceinline__ void put(float* p, int64_t, const float (&v)[8]) {
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; }
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
};
template <> struct Row8<StreamH> {
    static __device__ __forceinline__ void get(const StreamH* p, int64_t, float (&v)[8]) { load8_f16(reinterpret_cast<const _Float16*>(p), v); }
    static __device__ __forceinline__ void put(StreamH* p, int64_t, const float (&v)[8]) { store8_f16(reinterpret_cast<_Float16*>(p), v); }
};

template <typename ST, typename DT, bool VEC>
__global__ void cast_rows_kernel(const ST* __restrict__ src, int64_t lds, DT* __restrict__ dst, int64_t ldd, int64_t rows,
                                 int64_t cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const int64_t cv = cols >> 3;
        if (i >= rows * cv) return;
        const int64_t r = i / cv, c = (i - r * cv) << 3;
        float v[8];
        Row8<ST>::get(src + r * lds + c, lds, v);
        Row8<DT>::put(dst + r * ldd + c, ldd, v);
    } else {
        if (i >= rows * cols) return;
        const int64_t r = i / cols, c = i - r * cols;
        const float v = row_value(src + r * lds + c, (int)lds);
        if constexpr (sizeof(DT) == 4) dst[r * ldd + c] = v;
        else if constexpr (__is_same(DT, StreamH)) dst[r * ldd + c].v = f16_sat(v);
        else store1_operand(dst + r * ldd + c, ldd / PLANES, v);
    }
}

__global__ void cast_kernel(const float* __restrict__ src, h16* __restrict__ dst, int64_t n8, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8), b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (h16)a[e]; o[4 + e] = (h16)b[e]; }
        st16(dst + i * 8, as_u32x4(o));
    } else if (i == n8) {
        for (int64_t j = n8 * 8; j < n; ++j) dst[j] = (h16)src[j];
    }
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += alpha * x[i];
}

// out[b][i] = ca[b] * x[b][i] + cb[b] * y[b][i]  (per-sample schedule coefficients gathered on the host side)
__global__ void lincomb_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ y,
                               const float* __restrict__ ca, const float* __restrict__ cb, int64_t n) {
    const int b = blockIdx.y;
    const float a = ca[b], c = cb[b];
    const int64_t off = (int64_t)b * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[off + i] = a * x[off + i] + c * y[off + i];
}

__global__ void gaussian_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise, float* __restrict__ out,
                                       int C, int HW, float scale, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over (n, c, p)
    if (i >= total) return;
    const int64_t chw = (int64_t)C * HW;
    const int64_t n = i / chw, r = i - n * chw;
    const float mean = mom[n * 2 * chw + r];
    float lv = mom[n * 2 * chw + chw + r];
    lv = fminf(fmaxf(lv, -30.f), 20.f);
    float z = mean;
    if (noise) z = mean + expf(0.5f * lv) * noise[i];
    out[i] = scale * z;
}

// ---- DDIM -------------------------------------------------------------------------------------------------
constexpr int DDIM_BLK = 64;   // partial-sum blocks per sample

__global__ __launch_bounds__(256) void ddim_stats_kernel(const float* __restrict__ ec, const float* __restrict__ eu,
                                                          const float* __restrict__ em, int64_t n, float cfg, float cfg_img,
                                                          double* __restrict__ ws) {
    __shared__ double red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const float* c = ec + (int64_t)b * n;
    const float* u = eu ? eu + (int64_t)b * n : nullptr;
    const float* mid = em ? em + (int64_t)b * n : nullptr;
    const int64_t per = (n + DDIM_BLK - 1) / DDIM_BLK;
    const int64_t i0 = blockIdx.x * per, i1 = (i0 + per < n) ? i0 + per : n;
    double sc = 0, qc = 0, sv = 0, qv = 0;
    for (int64_t i = i0 + tid; i < i1; i += 256) {
        const float a = c[i];
        float v = a;
        if (u) v = mid ? u[i] + cfg_img * (mid[i] - u[i]) + cfg * (a - mid[i]) : u[i] + cfg * (a - u[i]);
        sc += a; qc += (double)a * a; sv += v; qv += (double)v * v;
    }
    sc = wave_sum_d(sc); qc = wave_sum_d(qc); sv = wave_sum_d(sv); qv = wave_sum_d(qv);
    if (lane == 0) { red[wave][0] = sc; red[wave][1] = qc; red[wave][2] = sv; red[wave][3] = qv; }
    __syncthreads();
    if (tid < 4)
        ws[((int64_t)b * DDIM_BLK + blockIdx.x) * 4 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

struct DdimCoef { float cfg, phi, sqrt_ac, sqrt_1mac, rescale, sqrt_a_prev, dir_coef, sigma, cfg_img, eps_form; };

__global__ __launch_bounds__(256) void ddim_update_kernel(const float* __restrict__ x, const float* __restrict__ ec,
                                                           const float* __restrict__ eu, const float* __restrict__ em,
                                                           const float* __restrict__ noise,
                                                           float* __restrict__ x_prev, float* __restrict__ pred_x0, int64_t n,
                                                           DdimCoef k, const double* __restrict__ ws) {
    __shared__ float s_ratio;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        float ratio = 1.f;
        if (k.phi > 0.f) {
            double sc = 0, qc = 0, sv = 0, qv = 0;
            for (int i = 0; i < DDIM_BLK; ++i) {
                const double* p = ws + ((int64_t)b * DDIM_BLK + i) * 4;
                sc += p[0]; qc += p[1]; sv += p[2]; qv += p[3];
            }
            const double dn = (double)n;
            const double var_c = (qc - sc * sc / dn) / (dn - 1.0);   // unbiased, as torch.std
            const double var_v = (qv - sv * sv / dn) / (dn - 1.0);
            const float std_c = (float)sqrt(var_c > 0 ? var_c : 0.0), std_v = (float)sqrt(var_v > 0 ? var_v : 0.0);
            ratio = std_c / std_v;
        }
        s_ratio = ratio;
    }
    __syncthreads();
    const float ratio = s_ratio;
    const int64_t off = (int64_t)b * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float xv = x[off + i], a = ec[off + i];
        float v = a;
        if (eu) {
            const float u = eu[off + i];
            v = em ? u + k.cfg_img * (em[off + i] - u) + k.cfg * (a - em[off + i]) : u + k.cfg * (a - u);
        }
        if (k.phi > 0.f) {
            const float resc = v * ratio;
            v = k.phi * resc + (1.f - k.phi) * v;
        }
        float e, x0;
        if (k.eps_form != 0.f) {          // the model predicts eps: sqrt_ac = sqrt(a_t), sqrt_1mac = sqrt(1 - a_t)
            e = v;
            x0 = (xv - k.sqrt_1mac * e) / k.sqrt_ac;
        } else {                          // v-prediction
            e = k.sqrt_ac * v + k.sqrt_1mac * xv;
            x0 = k.sqrt_ac * xv - k.sqrt_1mac * v;
        }
        x0 *= k.rescale;
        const float dir = k.dir_coef * e;
        const float nz = noise ? k.sigma * noise[off + i] : 0.f;
        pred_x0[off + i] = x0;
        x_prev[off + i] = k.sqrt_a_prev * x0 + dir + nz;
    }
}

}  // namespace

extern "C" int mudg_timestep_embedding(const int64_t* t, const float* freqs, float* out, int n, int dim, void* stream) {
    MUDG_REQUIRE(t && freqs && out && n > 0 && dim >= 2, "mudg_timestep_embedding: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int tot = n * (dim / 2);
    hipLaunchKernelGGL(temb_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, t, freqs, out, n, dim);
    return mudg_check_launch("mudg_timestep_embedding");
}

extern "C" int mudg_small_linear(const float* x, const void* W, int w_is_bf16, const float* b, float* y, int M, int N, int K,
                                 int act_in, int act_out, int accumulate, void* stream) {
    MUDG_REQUIRE(x && W && y && M > 0 && N > 0 && K > 0, "mudg_small_linear: bad arguments");
    MUDG_REQUIRE(M <= 65535, "mudg_small_linear: M too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((N + 3) / 4, M);
    if (w_is_bf16)
        hipLaunchKernelGGL(small_linear_kernel<h16>, grid, dim3(256), 0, s, x, (const h16*)W, b, y, M, N, K, act_in, act_out, accumulate);
    else
        hipLaunchKernelGGL(small_linear_kernel<float>, grid, dim3(256), 0, s, x, (const float*)W, b, y, M, N, K, act_in, act_out, accumulate);
    return mudg_check_launch("mudg_small_linear");
}

extern "C" int mudg_ncthw_to_rows(const void* src, int src_is_fp32, void* dst, int B, int C, int T, int HW, int ld, int coff,
                                  int Ttot, int t0, void* stream) {
    MUDG_REQUIRE(src && dst && B > 0 && C > 0 && T > 0 && HW > 0 && coff >= 0 && ld % PLANES == 0 && coff + C <= ld / PLANES, "mudg_ncthw_to_rows: bad arguments");
    if (Ttot <= 0) { Ttot = T; t0 = 0; }
    MUDG_REQUIRE(t0 >= 0 && t0 + T <= Ttot, "mudg_ncthw_to_rows: frame window [%d, %d) outside %d", t0, t0 + T, Ttot);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)B * T * HW;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (src_is_fp32) hipLaunchKernelGGL(ncthw_to_rows_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (h16*)dst, B, C, T, HW, ld, coff, Ttot, t0);
    else hipLaunchKernelGGL(ncthw_to_rows_kernel<h16>, grid, dim3(256), 0, s, (const h16*)src, (h16*)dst, B, C, T, HW, ld, coff, Ttot, t0);
    return mudg_check_launch("mudg_ncthw_to_rows");
}

extern "C" int mudg_rows_to_ncthw(const void* src, int src_is_fp32, int ld, int coff, void* dst, int dst_is_fp32, int B, int C,
                                  int T, int HW, float scale, int Ttot, int t0, void* stream) {
    MUDG_REQUIRE(src && dst && B > 0 && C > 0 && T > 0 && HW > 0 && coff >= 0 && (src_is_fp32 ? coff + C <= ld : (ld % PLANES == 0 && coff + C <= ld / PLANES)),
                 "mudg_rows_to_ncthw: bad arguments");
    if (Ttot <= 0) { Ttot = T; t0 = 0; }
    MUDG_REQUIRE(t0 >= 0 && t0 + T <= Ttot, "mudg_rows_to_ncthw: frame window [%d, %d) outside %d", t0, t0 + T, Ttot);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)B * T * HW;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (src_is_fp32 == KIND_F16) {
        if (dst_is_fp32) hipLaunchKernelGGL((rows_to_ncthw_kernel<StreamH, float>), grid, dim3(256), 0, s, (const StreamH*)src, ld, coff, (float*)dst, B, C, T, HW, scale, Ttot, t0);
        else hipLaunchKernelGGL((rows_to_ncthw_kernel<StreamH, h16>), grid, dim3(256), 0, s, (const StreamH*)src, ld, coff, (h16*)dst, B, C, T, HW, scale, Ttot, t0);
    } else if (src_is_fp32) {
        if (dst_is_fp32) hipLaunchKernelGGL((rows_to_ncthw_kernel<float, float>), grid, dim3(256), 0, s, (const float*)src, ld, coff, (float*)dst, B, C, T, HW, scale, Ttot, t0);
        else hipLaunchKernelGGL((rows_to_ncthw_kernel<float, h16>), grid, dim3(256), 0, s, (const float*)src, ld, coff, (h16*)dst, B, C, T, HW, scale, Ttot, t0);
    } else {
        if (dst_is_fp32) hipLaunchKernelGGL((rows_to_ncthw_kernel<h16, float>), grid, dim3(256), 0, s, (const h16*)src, ld, coff, (float*)dst, B, C, T, HW, scale, Ttot, t0);
        else hipLaunchKernelGGL((rows_to_ncthw_kernel<h16, h16>), grid, dim3(256), 0, s, (const h16*)src, ld, coff, (h16*)dst, B, C, T, HW, scale, Ttot, t0);
    }
    return mudg_check_launch("mudg_rows_to_ncthw");
}

extern "C" int mudg_zero_channels(void* dst, int rows, int ld, int c0, int c1, void* stream) {
    MUDG_REQUIRE(dst && rows > 0 && c0 >= 0 && c1 > c0 && ld % PLANES == 0 && c1 <= ld / PLANES, "mudg_zero_channels: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)rows * (c1 - c0);
    hipLaunchKernelGGL(zero_channels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (h16*)dst, (int64_t)rows, ld, c0, c1);
    return mudg_check_launch("mudg_zero_channels");
}

template <typename ST, typename DT>
static void launch_cast_rows2(const ST* src, int64_t lds, DT* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t s) {
    const int sg = __is_same(ST, h16) ? 8 * PLANES : 8, dg = __is_same(DT, h16) ? 8 * PLANES : 8;      // row-stride granules
    const bool vec = (cols & 7) == 0 && lds % sg == 0 && ldd % dg == 0 && aligned16(src) && aligned16(dst);
    if (vec) {
        const dim3 grid((unsigned)((rows * (cols >> 3) + 255) / 256));
        hipLaunchKernelGGL((cast_rows_kernel<ST, DT, true>), grid, dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
    } else {
        const dim3 grid((unsigned)((rows * cols + 255) / 256));
        hipLaunchKernelGGL((cast_rows_kernel<ST, DT, false>), grid, dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
    }
}

template <typename ST>
static void launch_cast_rows(const ST* src, int64_t lds, void* dst, int dst_kind, int64_t ldd, int64_t rows, int64_t cols, hipStream_t s) {
    if (dst_kind == KIND_F32) launch_cast_rows2(src, lds, (float*)dst, ldd, rows, cols, s);
    else if (dst_kind == KIND_F16) launch_cast_rows2(src, lds, (StreamH*)dst, ldd, rows, cols, s);
    else launch_cast_rows2(src, lds, (h16*)dst, ldd, rows, cols, s);
}

extern "C" int mudg_cast_rows(const void* src, int src_fp32, int64_t lds, void* dst, int dst_fp32, int64_t ldd, int64_t rows,
                              int64_t cols, void* stream) {
    MUDG_REQUIRE(src && dst && rows > 0 && cols > 0, "mudg_cast_rows: bad arguments");
    MUDG_REQUIRE(src_fp32 >= 0 && src_fp32 <= 2 && dst_fp32 >= 0 && dst_fp32 <= 2, "mudg_cast_rows: kinds are 0 (operand), 1 (fp32), 2 (fp16)");
    MUDG_REQUIRE(src_fp32 ? lds >= cols : (lds % PLANES == 0 && lds / PLANES >= cols), "mudg_cast_rows: lds=%lld", (long long)lds);
    MUDG_REQUIRE(dst_fp32 ? ldd >= cols : (ldd % PLANES == 0 && ldd / PLANES >= cols), "mudg_cast_rows: ldd=%lld", (long long)ldd);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (src_fp32 == KIND_F32) launch_cast_rows((const float*)src, lds, dst, dst_fp32, ldd, rows, cols, s);
    else if (src_fp32 == KIND_F16) launch_cast_rows((const StreamH*)src, lds, dst, dst_fp32, ldd, rows, cols, s);
    else launch_cast_rows((const h16*)src, lds, dst, dst_fp32, ldd, rows, cols, s);
    return mudg_check_launch("mudg_cast_rows");
}

extern "C" int mudg_copy_rows(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, void* stream) {
    MUDG_REQUIRE(src && dst && rows > 0 && cols > 0 && lds % PLANES == 0 && ldd % PLANES == 0 && lds / PLANES >= cols &&
                 ldd / PLANES >= cols, "mudg_copy_rows: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int vec = aligned16(src) && aligned16(dst) && lds % (8 * PLANES) == 0 && ldd % (8 * PLANES) == 0 && !(cols & 7);
    const int64_t n = vec ? rows * (cols >> 3) : rows * cols;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const h16*)src, lds, (h16*)dst,
                       ldd, rows, cols, vec);
    return mudg_check_launch("mudg_copy_rows");
}

extern "C" int mudg_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
    MUDG_REQUIRE(src && dst && n > 0 && aligned16(src) && aligned16(dst), "mudg_cast_f32_bf16: bad arguments");
    if (PLANES > 1) MUDG_FAIL(MUDG_EUNSUPPORTED, "mudg_cast_f32_bf16: a flat cast has no plane layout in the split-operand builds; use mudg_cast_rows");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t n8 = n >> 3;
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)((n8 + 1 + 255) / 256)), dim3(256), 0, s, src, (h16*)dst, n8, n);
    return mudg_check_launch("mudg_cast_f32_bf16");
}

extern "C" int mudg_axpy_f32(float* y, const float* x, int64_t n, float alpha, void* stream) {
    MUDG_REQUIRE(y && x && n > 0, "mudg_axpy_f32: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, x, n, alpha);
    return mudg_check_launch("mudg_axpy_f32");
}

extern "C" int mudg_lincomb(float* out, const float* x, const float* y, const float* ca, const float* cb, int B, int64_t n,
                            void* stream) {
    MUDG_REQUIRE(out && x && y && ca && cb && B > 0 && B <= 65535 && n > 0, "mudg_lincomb: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int gx = (int)((n + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(lincomb_kernel, dim3(gx, B), dim3(256), 0, s, out, x, y, ca, cb, n);
    return mudg_check_launch("mudg_lincomb");
}

extern "C" int mudg_gaussian_sample(const float* moments, const float* noise, float* out, int N, int C, int HW, float scale,
                                    void* stream) {
    MUDG_REQUIRE(moments && out && N > 0 && C > 0 && HW > 0, "mudg_gaussian_sample: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)N * C * HW;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, moments, noise, out, C, HW,
                       scale, total);
    return mudg_check_launch("mudg_gaussian_sample");
}

extern "C" int64_t mudg_ddim_ws_doubles(int B) { return B > 0 ? (int64_t)B * DDIM_BLK * 4 : 0; }

extern "C" int mudg_ddim_step(const float* x, const float* e_c, const float* e_u, const float* e_m, const float* noise,
                              float* x_prev, float* pred_x0, int B, int64_t n, const float* host_coef, double* ws,
                              void* stream) {
    MUDG_REQUIRE(x && e_c && x_prev && pred_x0 && host_coef && ws, "mudg_ddim_step: null pointer");
    MUDG_REQUIRE(B > 0 && B <= 65535 && n > 1, "mudg_ddim_step: bad sizes");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DdimCoef k;
    k.cfg = host_coef[0]; k.phi = host_coef[1]; k.sqrt_ac = host_coef[2]; k.sqrt_1mac = host_coef[3];
    k.rescale = host_coef[4]; k.sqrt_a_prev = host_coef[5]; k.dir_coef = host_coef[6]; k.sigma = host_coef[7];
    k.cfg_img = host_coef[8]; k.eps_form = host_coef[9];
    MUDG_REQUIRE(!e_m || e_u, "mudg_ddim_step: e_m needs e_u");
    const int slot = mudg_prof_begin(MUDG_FAM_MISC, s);
    if (k.phi > 0.f)
        hipLaunchKernelGGL(ddim_stats_kernel, dim3(DDIM_BLK, B), dim3(256), 0, s, e_c, e_u, e_m, n, k.cfg, k.cfg_img, ws);
    int gx = (int)((n + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(ddim_update_kernel, dim3(gx, B), dim3(256), 0, s, x, e_c, e_u, e_m, noise, x_prev, pred_x0, n, k, ws);
    const int rc = mudg_check_launch("mudg_ddim_step");
    mudg_prof_end(slot, s, 0.0, (double)B * n * 4.0 * 8.0);
    return rc;
}
