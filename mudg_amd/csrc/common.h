// common.h — shared device helpers for the gfx950 kernels (wave64, MFMA, h16 storage / fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mudg_hip.h"

// h16 = the 16-bit MFMA operand type of this build: bfloat16 by default (libmudg_hip.so), IEEE half when compiled
// with -DMUDG_OPERAND_FP16 (libmudg_hip_fp16.so).  Same MFMA rate, same bytes; fp16 has three more mantissa bits
// (operand rounding 2^-12 instead of 2^-9) and is what the reference itself computes in under torch.autocast.
//
// Split-operand precision modes (-DMUDG_PLANES=2 -> libmudg_hip_x3.so, =3 -> libmudg_hip_x6.so): an operand value x is
// carried as PLANES bf16 numbers x = x0 + x1 (+ x2) with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)
// (16 / 24 significand bits), and a product sum_k x_k w_k is accumulated from the bf16 x bf16 partial products whose
// magnitude is above the representation error: x0 w0 + x0 w1 + x1 w0 (3 MFMAs per tile, PLANES = 2) or additionally
// x0 w2 + x2 w0 + x1 w1 (6 MFMAs, PLANES = 3: fp32-class).  In memory, an operand matrix [rows][C] with row stride ld
// holds plane p of element (r, c) at r * ld + p * (ld / PLANES) + c: allocations are PLANES times as wide and any column
// slice of one keeps the same plane distance.  The contraction kernels run the ordinary K loop once per kept (x plane,
// w plane) pair with shifted column offsets — the MFMA code itself is the same as in the 16-bit builds.
#ifndef MUDG_PLANES
#define MUDG_PLANES 1
#endif
#ifdef MUDG_OPERAND_FP16
#if MUDG_PLANES != 1
#error "split operands are bf16"
#endif
typedef _Float16 h16;
#define MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define DOT2_H16(a, b, c) __builtin_amdgcn_fdot2(a, b, c, false)               /* v_dot2c_f32_f16: c + a.x b.x + a.y b.y */
#define MUDG_OPERAND_CODE 1
#else
typedef __bf16 h16;
#define MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define DOT2_H16(a, b, c) __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false)      /* v_dot2c_f32_bf16 */
#define MUDG_OPERAND_CODE (MUDG_PLANES == 1 ? 0 : MUDG_PLANES)
#endif
constexpr int PLANES = MUDG_PLANES;
// (x plane, w plane) pairs of the kept partial products, smallest terms first so they are not absorbed by the large one.
constexpr int NSEG = PLANES == 1 ? 1 : (PLANES == 2 ? 3 : 6);
__host__ __device__ constexpr int seg_xp(int s) { return PLANES == 1 ? 0 : (PLANES == 2 ? (s == 0 ? 1 : 0) : (s == 0 ? 2 : (s == 1 ? 0 : (s == 2 ? 1 : (s == 3 ? 1 : 0))))); }
__host__ __device__ constexpr int seg_wp(int s) { return PLANES == 1 ? 0 : (PLANES == 2 ? (s == 1 ? 1 : 0) : (s == 0 ? 0 : (s == 1 ? 2 : (s == 2 ? 1 : (s == 4 ? 1 : 0))))); }
// PLANES = 2: (1,0) (0,1) (0,0);  PLANES = 3: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) h16 h16x4;
typedef __attribute__((ext_vector_type(2))) h16 h16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x4 zero16() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }

union Pack16 { u32x4 u; h16x8 h; };
union Pack8 { u32x2 u; h16x4 h; };

__device__ __forceinline__ h16x8 as_h16x8(u32x4 v) { Pack16 p; p.u = v; return p.h; }
__device__ __forceinline__ u32x4 as_u32x4(h16x8 v) { Pack16 p; p.h = v; return p.u; }

// ---- residual-stream / plain buffers: storage kind codes used by out_fp32 / res_fp32 / x_fp32 style arguments ----
// 0 = MFMA operand (h16, PLANES pieces), 1 = fp32, 2 = IEEE fp16 "stream" storage: the tensors later layers add onto
// (block outputs, the transformers' token stream, skips) are kept at 2 bytes per value in the 16-bit builds — the
// reference's own stream is fp16 under torch.autocast — and at fp32 in the split-operand precision builds.
enum { KIND_OPERAND = 0, KIND_F32 = 1, KIND_F16 = 2 };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ void load8_f16(const _Float16* p, float (&v)[8]) {
    union { u32x4 u; f16x8 h; } t; t.u = ld16(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t.h[e];
}
// Stream stores saturate: a residual sum beyond the IEEE-half range becomes +-65504 (one v_med3 per value), never inf —
// an inf in the stream turns the next GroupNorm / LayerNorm into NaN for the whole sample.  MUDG_STREAM=fp32 (ops.STREAM) is
// the escape hatch for checkpoints whose stream really leaves the half range.
__device__ __forceinline__ _Float16 f16_sat(float v) { return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
__device__ __forceinline__ void store8_f16(_Float16* p, const float (&v)[8]) {
    union { u32x4 u; f16x8 h; } t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t.h[e] = f16_sat(v[e]);
    st16(p, t.u);
}

// ---- operand element I/O: one logical value = PLANES h16 numbers `ps` elements apart (ps = row stride / PLANES) ----
// Eight consecutive channels: 16-byte accesses per plane.
__device__ __forceinline__ void store8_operand(h16* dst, int64_t ps, const float (&v)[8]) {
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = v[e];
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = (h16)r[e]; r[e] -= (float)o[e]; }
        st16(dst + p * ps, as_u32x4(o));
    }
}
__device__ __forceinline__ void load8_operand(const h16* src, int64_t ps, float (&v)[8]) {
    const h16x8 t0 = as_h16x8(ld16(src));
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t0[e];
#pragma unroll
    for (int p = 1; p < PLANES; ++p) {
        const h16x8 t = as_h16x8(ld16(src + p * ps));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)t[e];
    }
}
__device__ __forceinline__ void store1_operand(h16* dst, int64_t ps, float v) {
#pragma unroll
    for (int p = 0; p < PLANES; ++p) { const h16 o = (h16)v; dst[p * ps] = o; v -= (float)o; }
}
__device__ __forceinline__ float load1_operand(const h16* src, int64_t ps) {
    float v = (float)src[0];
#pragma unroll
    for (int p = 1; p < PLANES; ++p) v += (float)src[p * ps];
    return v;
}
// One value -> its PLANES pieces in registers (P of the attention kernels).
__device__ __forceinline__ void split_operand(float v, h16 (&o)[PLANES]) {
#pragma unroll
    for (int p = 0; p < PLANES; ++p) { o[p] = (h16)v; v -= (float)o[p]; }
}
// The value an operand store will represent (GroupNorm partial sums are taken over what was stored).
__device__ __forceinline__ float operand_round(float v) {
    float r = v, acc = 0.f;
#pragma unroll
    for (int p = 0; p < PLANES; ++p) { const h16 o = (h16)r; acc += (float)o; r -= (float)o; }
    return acc;
}

// OCP MX-fp8 (e4m3 + E8M0 scale per 32 values): the shared exponent of a block with maximum magnitude amax and the e4m3 bytes
// of four values scaled by 2^-E (saturating, round to nearest even in hardware).  Used by mudg_quantize_mxfp8 and by the GEMM
// epilogue's fused fp8 copy: one definition, bit-equal results.
#if MUDG_PLANES == 1
__device__ __forceinline__ int mx_block_exponent(float amax) {
    int E = 0;
    if (amax > 0.f) {
        E = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127 - 8;
        E = E < -127 ? -127 : (E > 127 ? 127 : E);
    }
    return E;
}
__device__ __forceinline__ unsigned mx_pack4_e4m3(float a0, float a1, float a2, float a3, float inv) {
    a0 = fminf(fmaxf(a0 * inv, -448.f), 448.f); a1 = fminf(fmaxf(a1 * inv, -448.f), 448.f);
    a2 = fminf(fmaxf(a2 * inv, -448.f), 448.f); a3 = fminf(fmaxf(a3 * inv, -448.f), 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, w, true);
    return (unsigned)w;
}
#endif

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- host side ---------------------------------------------------------------------------------
// Kernel-variant switches (A/B measurements, tests that force a particular kernel): compiled in only with
// -DMUDG_DEBUG_VARIANTS (libmudg_hip_dbg.so, used by tests/test_gemm_variants_gpu.py and tools/); the shipped libraries
// take the default — no environment variable changes which kernel they run.
#ifdef MUDG_DEBUG_VARIANTS
int mudg_variant(const char* name, int dflt);         // getenv("MUDG_" name), read at every call (call sites cache it)
#else
static inline int mudg_variant(const char*, int dflt) { return dflt; }
#endif
void mudg_set_error(const char* fmt, ...);
#define MUDG_FAIL(code, ...) do { mudg_set_error(__VA_ARGS__); return (code); } while (0)
#define MUDG_REQUIRE(cond, ...) do { if (!(cond)) MUDG_FAIL(MUDG_EINVAL, __VA_ARGS__); } while (0)

// Event profiler hooks (capi.hip).  begin returns a slot (or -1 when the family is not traced).
int  mudg_prof_begin(int fam, hipStream_t s);
void mudg_prof_end(int slot, hipStream_t s, double flops, double bytes);

static inline int mudg_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mudg_set_error("%s: %s", what, hipGetErrorString(e)); return MUDG_ELAUNCH; }
    return MUDG_OK;
}
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
