// gemm_shared.h — what the contraction kernels of gemm.hip and pgemm.hip have in common: K-tile geometry, the buffer-descriptor
// helpers of the LDS-DMA loader, the GELU forms of the epilogues, and the host-side state both launchers use.
#pragma once
#include "common.h"

constexpr int BK = 64;
constexpr int LDSLD = 64;                       // h16 elements per LDS row: unpadded, XOR-swizzled (see gemm.hip)
constexpr int VF_Y = 1, VF_R = 2;               // 16-byte access allowed on Y / R
constexpr int VF_TM = 8;                        // temporal conv: a tile is 8 pixels x 16 frames (gemm.hip, TMAP / TSHARE)
constexpr int VF_XS = 4;                        // 3x3 conv: the three dx taps of a dy share one staged activation tile (gemm.hip, XSHARE)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// bf16x3 build, descriptor loader (FUSED): ONE K-tile stage holds both pieces of both operands (see gemm.hip).
constexpr bool fused_planes(bool fast) { return PLANES == 2 && fast; }

// Used by the 16-bit builds where the table is off and by the bf16x3 build (whose operands carry 16 significand bits: the
// polynomial's 1.5e-7 is two orders below their representation error); bf16x6 evaluates erff exactly.
__device__ __forceinline__ float gelu_fast(float x) {
    // 0.5 x (1 + erf(x / sqrt 2)) with Abramowitz-Stegun 7.1.26 for erf (|abs err| < 1.5e-7): the exact erff
    // costs about as much as the whole K loop of a K = 320 tile; the result is rounded to h16 anyway.
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-z * z);          // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

// voffset of a lane whose source row / tap does not exist: at num_records, so the buffer load returns 0 into the LDS.
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const h16* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(base), 0, (int)0x80000000u, 0x00020000);
}

// GEGLU's gate: gelu(x) = x Phi(x) with Phi linearly interpolated from a 1025-entry table over [-8, 8] held in LDS
// (|error| <= h^2/8 max|Phi''| = 7.4e-6 at h = 1/64 — below the h16 rounding of the result by two orders).
constexpr int PHI_N = 1024;
constexpr int PHI_BYTES = (PHI_N + 4) * 4;
__device__ __forceinline__ float gelu_lut(float x, const float* __restrict__ T) {
    float u = fmaf(x, 64.0f, 512.0f);
    u = __builtin_amdgcn_fmed3f(u, 0.0f, 1023.99f);
    const int i = (int)u;
    const float f = u - (float)i;
    const float a = T[i], b = T[i + 1];
    return x * fmaf(f, b - a, a);
}


// Host side (gemm.hip): lazily created per-device state and the kernel-selection rule shared by both launchers.
constexpr int MAX_DEVICES = 64;
int mudg_current_device();
const float* mudg_phi_table(bool split_ok = false);      // device Phi table, or nullptr (split-operand builds unless split_ok in bf16x3 / variant switch)
// pgemm.hip: the persistent 128 x 128 kernel; wgs = workgroups per CU (4 | 3: one K-tile stage, 2: two)
int mudg_pgemm_launch(const MudgGemmDesc& d, int vflags, int wgs, hipStream_t s);
// wgemm.hip: the 288 x 320 eight-wave tile (16-bit builds); _ok = eligible AND selected by its M-independent rule
bool mudg_wgemm_ok(const MudgGemmDesc& d, int vflags);
int mudg_wgemm_rows(const MudgGemmDesc& d, int vflags);      // 288 | 160 (w160_kernel, 16-bit builds) | 0
int mudg_wgemm_launch(const MudgGemmDesc& d, int vflags, hipStream_t s);
