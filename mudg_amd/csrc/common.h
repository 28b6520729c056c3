// common.h — shared device helpers for the gfx950 kernels (wave64, MFMA, h16 storage / fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mudg_hip.h"

// h16 = the 16-bit MFMA operand type of this build: bfloat16 by default (libmudg_hip.so), IEEE half when compiled
// with -DMUDG_OPERAND_FP16 (libmudg_hip_fp16.so).  Same MFMA rate, same bytes; fp16 has three more mantissa bits
// (operand rounding 2^-12 instead of 2^-9) and is what the reference itself computes in under torch.autocast.
#ifdef MUDG_OPERAND_FP16
typedef _Float16 h16;
#define MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define DOT2_H16(a, b, c) __builtin_amdgcn_fdot2(a, b, c, false)               /* v_dot2c_f32_f16: c + a.x b.x + a.y b.y */
#define MUDG_OPERAND_CODE 1
#else
typedef __bf16 h16;
#define MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define DOT2_H16(a, b, c) __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false)      /* v_dot2c_f32_bf16 */
#define MUDG_OPERAND_CODE 0
#endif
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
