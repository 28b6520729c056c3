// wgemm.hip — the 288 x 320 tile of the contraction family (round 5): eight waves, v_mfma_f32_16x16x32, one workgroup per CU,
// LDS-DMA ring managed per k half, six barrier-delimited phases per K-tile with the two wave groups of a SIMD in anti-phase.
// Plain GEMM, same-size 3x3 conv and temporal 3-tap conv (descriptor loader, 16-bit builds).  MudgGemmDesc semantics: mudg_hip.h.
// Also in this file: the persistent form of the tile (wgemm_pkernel), the two-workgroup 144 x 256 GEGLU kernel (hgeglu_kernel, round 6: a
// measured negative, rule "never") and the 160 x 320 tile for frames of whole 160-row tiles (wq_kernel, round 6) — each under its own header.
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
// Epilogue: straight from the accumulators, no LDS staging (w_epilogue): a lane owns 8 consecutive channels of each of its 9 rows per
// fragment pair — one 16-byte piece per row — so value = alpha acc + bias (+ group bias) is rounded and stored from registers; the
// GroupNorm partials of what was stored go out per 288-ROW BLOCK = per tile (mudg_gemm_stats_rows).  A residual is not fetched by the
// epilogue at all: it SEEDS the accumulators when the tile starts (w_seed).
// Bits.  Summation order over K is the K-tile order of the 128 x 128 kernels; without a residual the results are BIT-IDENTICAL to the
// 128 x 128 kernels' on every shape and epilogue the suite runs (tests/test_gemm_variants_gpu.py::
// test_wide288_is_bit_identical_to_the_one_tile_kernels — an observation about v_mfma_f32_16x16x32 vs two v_mfma_f32_32x32x16 on gfx950,
// asserted by that test, not a documented property of the instructions).  With a residual they are not — the sum is ((r + x w) + bias) here
// and ((x w + bias) + r) there — and that alone is why the selection rule (mudg_wgemm_ok) never looks at M: a clip's result must not
// depend on the batch it travels in, so whether a residual problem runs on this tile is decided by its per-frame geometry only.
//
// bf16x3 build (MUDG_PLANES = 2), same tile, same epilogue: a k half holds both bf16 pieces of both operands (x0, x1, w0, w1: 76 KiB),
// so the ring is two k halves and a piece is re-staged two phases after its last read.  Per k half h, three phases of 45 MFMAs:
//   p   fragment reads                          MFMAs                     DMA issued (this wave's share)          wait at the end of the MFMA section
//   0   w0, w1 (10), x1 / x0 rows 0-2 (6)       w0 x1, w0 x0, w1 x0       x1 / x0 rows 3-5 of k half h + 1        x rows 6-8 of k half h
//   1   x1 / x0 rows 3-5                        (rows 3-5)                x1 / x0 rows 6-8 of k half h + 1        w0, w1, x rows 0-2 of k half h + 1
//   2   x1 / x0 rows 6-8                        (rows 6-8)                w0, w1, x1 / x0 rows 0-2 of h + 2       x rows 3-5 of k half h + 1
// x1 w0 + x0 w0 + x0 w1 per k half (x1 w1, 2^-18 relative, is dropped as in the 128 x 128 fused-piece kernel); 135 MFMAs per 28 fragment
// reads and 6 barriers.  Accumulators + fragments = 244 of the 256 registers: the MFMAs are inline assembly with the accumulator tied to
// its result (the register allocator otherwise splits accumulator live ranges: copies at every phase, then spills), the per-lane DMA
// state is two packed registers, and nothing of the K loop spills (a scratch reload between two DMA issues would drain the ring).
// Measured on the way (same box, MI355X, tools/exp_w288.py under MUDG_OPERAND=bf16x3; profiles/r5/w288_x3_phases.txt): nine phases of 15
// MFMAs (the 16-bit loop's shape, w0 x1 | w0 x0 | w1 x0 per row third) 398-444 TFLOP/s on the 3x3 convs, six phases (30 30 30 15 15 15)
// 422-468, these three 468-521 (= 1400-1560 TFLOP/s of MFMA work; the 128 x 128 fused-piece kernels: 345-415): what a barrier slot
// costs beside its MFMAs (~140 cycles) is amortised over three times as many of them.
#include "gemm_shared.h"
#include <type_traits>

#if MUDG_PLANES <= 2
namespace {

constexpr int WBM = 288;
constexpr int WNA = WBM / 16;                            // 16-row subtiles of the X operand tile
constexpr int W_TAIL = 2 * 320 * 2 * 4;                  // epilogue hand-off of the GroupNorm partials between the two M halves (320-wide tile)
constexpr int W_TAIL_GEGLU = (PHI_N + 4) * 8;            // the GEGLU tile (256 wide): the Phi table as (value, step to the next entry) pairs
template <int NREP> struct WGeo {
    static constexpr int BN = 64 * NREP;                 // 4 wave columns x NREP fragments of 16
    static constexpr int NB = BN / 16;                   // 16-row subtiles of the W operand tile
    static constexpr int PL = (WNA + NB) * 1024;         // one operand piece (plane) of a k half: the X subtiles, then the W subtiles
    static constexpr int KS = PLANES * PL;               // one k half
    static constexpr int BUF = 2 * KS;                   // (16-bit builds) one K-tile buffer
    static constexpr int LOOP = (4 / PLANES) * KS;       // 16-bit: two K-tile buffers; bf16x3: two k halves.  NREP 5: 155648, NREP 4: 139264
    static constexpr int SMEM = LOOP + (NREP == 4 ? W_TAIL_GEGLU : W_TAIL);
};

#define W_BARRIER()                            \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
#define W_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// Debug-variants build only: s_memtime stamps of wave 0 of every workgroup at the phase boundaries of the GEGLU kernels (tools/exp_stamps.py).
#ifdef MUDG_DEBUG_VARIANTS
[[maybe_unused]] __device__ unsigned long long* g_stamps = nullptr;         // [workgroup][64]
#define W_STAMP(slot) do { if (tid == 0 && g_stamps && (slot) < 64) g_stamps[(size_t)blockIdx.x * 64 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W_STAMP(slot) do { } while (0)
#endif
#define H_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)        /* s_waitcnt lgkmcnt(0) (vmcnt, expcnt untouched) as the compiler-visible builtin */

__device__ __forceinline__ f32x4 mfma16(h16x8 a, h16x8 b, f32x4 c) {
#ifdef MUDG_OPERAND_FP16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
// The same instruction with the accumulator tied to its result register: in the bf16x3 loop (224 live accumulator and fragment
// registers under branches) the register allocator otherwise splits accumulator live ranges — copies at every phase, then spills.
__device__ __forceinline__ void mfma16_inplace(h16x8 a, h16x8 b, f32x4& c) {
#ifdef MUDG_OPERAND_FP16
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#endif
}
__device__ __forceinline__ unsigned opaque(unsigned v) { asm volatile("" : "+v"(v)); return v; }
template <int CTRL>
__device__ __forceinline__ float dpp_add16(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// Sum over the 16 lanes of a DPP row (the 16 pixels of a fragment column), butterfly in a fixed order: every lane gets the total.
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add16<0xB1>(v);          // quad_perm [1,0,3,2]
    v = dpp_add16<0x4E>(v);          // quad_perm [2,3,0,1]
    v = dpp_add16<0x141>(v);         // row_half_mirror
    v = dpp_add16<0x140>(v);         // row_mirror
    return v;
}

// Which columns of a tile a wave column owns.  A wave's NREP fragments are NREP / 2 PAIRS (32 channels: a lane's 2 x 4 accumulator
// registers are 8 consecutive ones, a 16-byte piece of a 16-bit result row) and, on the 320-wide tile, one unpaired fragment (16
// channels, 8-byte pieces).  The pairs of all four wave columns come first, 32 channels each — so every 16-byte-per-lane store (four
// lanes = 64 contiguous bytes) starts on a 64-byte boundary of the row and the two pairs of a wave fill one 128-byte line; the four
// unpaired fragments follow at channel 256.  (A wave column as 80 CONSECUTIVE channels put the pieces of the odd wave columns across
// sector boundaries: the epilogue ran at half the store / residual-fetch rate.)  Free: which W rows a wave's fragments fetch.
// GEGLU's gate through the Phi table of gemm_shared.h held as PAIRS (Phi_i, Phi_{i+1} - Phi_i): one 8-byte LDS read per value instead of two
// 4-byte ones, v_fract instead of a conversion back and a subtraction — 8 instead of 11 vector instructions, the same bits as gelu_lut
// (the step is the same fp32 difference, taken once when the workgroup copies the table).
__device__ __forceinline__ float gelu_lut2(float x, const f32x2* __restrict__ T) {
    float u = fmaf(x, 64.0f, 512.0f);
    u = __builtin_amdgcn_fmed3f(u, 0.0f, 1023.99f);
    const f32x2 t = T[(int)u];
    return x * fmaf(__builtin_amdgcn_fractf(u), t[1], t[0]);
}
// bf16x3: the same table with the SLOPE as second entry ((Phi_i, pdf(x_i) / 64): the kernel fills it in when it copies the table) and a
// cubic Hermite piece between two nodes — error h^4 / 384 max|d4 Phi| = 9e-11 beside the table's own fp32 rounding (6e-8), the class of
// gelu_fast's 1.5e-7, for 14 vector instructions and one LDS read of two adjacent nodes instead of a polynomial around v_exp and v_rcp.
__device__ __forceinline__ float gelu_hermite(float x, const f32x2* __restrict__ T) {
    float u = fmaf(x, 64.0f, 512.0f);
    u = __builtin_amdgcn_fmed3f(u, 0.0f, 1023.99f);
    const int i = (int)u;
    const float t = __builtin_amdgcn_fractf(u);
    const f32x2 n0 = T[i], n1 = T[i + 1];
    const float dl = n1[0] - n0[0];
    const float c3 = fmaf(-2.0f, dl, n0[1] + n1[1]);
    const float c2 = dl - n0[1] - c3;
    float r = fmaf(t, c3, c2);
    r = fmaf(t, r, n0[1]);
    r = fmaf(t, r, n0[0]);
    return x * r;
}
template <int NREP> __device__ __forceinline__ int wave_pair_col(int wc, int pair) { return 32 * ((NREP / 2) * wc + pair); }
template <int NREP> __device__ __forceinline__ int wave_single_col(int wc) { return 32 * (NREP / 2) * 4 + 16 * wc; }

// A residual as the accumulators' INITIAL VALUE (alpha = 1, host-checked): its 27 pieces per lane are requested when the tile starts — no
// accumulator is live yet, so all of them are in flight at once, next to the first K-tiles' DMA — instead of three dependent rounds of nine
// in an epilogue that nothing overlaps on a CU holding one workgroup (the 294912 x 320 x 320 out-projection 178 -> 158 us, 73728 x 640 x 640
// 102 -> 83, 18432 x 2560 x 1280 135 -> 121).  The
// sum is ((r + x w) + bias) instead of ((x w + bias) + r): not the bits of the 128 x 128 one-tile kernels, so whether a problem with a
// residual runs here never depends on M (mudg_wgemm_ok).
template <int NREP, int NI>
__device__ __forceinline__ void w_seed(const MudgGemmDesc& p, f32x4 (&acc)[NI][NREP], const int m0, const int n0, const int wr, const int wc, const int lane) {
    constexpr int NPAIR = NREP / 2;
    const int px = lane & 15, q4 = lane >> 4;
    const int RK = p.res_fp32;
    const int64_t mrow = (int64_t)m0 + wr * (16 * NI) + px;
    const int np = n0 + 8 * q4, ns = n0 + wave_single_col<NREP>(wc) + 4 * q4;
    if (RK == KIND_F32) {
        const float* R = reinterpret_cast<const float*>(p.R);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t m = mrow + 16 * i;
            const bool live = m < p.M;
#pragma unroll
            for (int P = 0; P < NPAIR; ++P) {
                const float* rp = R + m * p.ldr + np + wave_pair_col<NREP>(wc, P);
                acc[i][2 * P] = live ? *reinterpret_cast<const f32x4*>(rp) : f32x4{0.f, 0.f, 0.f, 0.f};
                acc[i][2 * P + 1] = live ? *reinterpret_cast<const f32x4*>(rp + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (NREP & 1) acc[i][2 * NPAIR] = live ? *reinterpret_cast<const f32x4*>(R + m * p.ldr + ns) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else if (RK == KIND_F16 || PLANES == 1) {           // 16-bit storage: the fp16 stream or a one-piece operand matrix
        const char* R = reinterpret_cast<const char*>(p.R);
        u32x4 ra[NI][NPAIR];
        u32x2 rs[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t m = mrow + 16 * i;
            const bool live = m < p.M;
#pragma unroll
            for (int P = 0; P < NPAIR; ++P) ra[i][P] = live ? ld16(R + (m * p.ldr + np + wave_pair_col<NREP>(wc, P)) * 2) : zero16();
            rs[i] = (live && (NREP & 1)) ? *reinterpret_cast<const u32x2*>(R + (m * p.ldr + ns) * 2) : u32x2{0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int P = 0; P < NPAIR; ++P) {
                if (RK == KIND_F16) {
                    union { u32x4 w; f16x8 h; } t; t.w = ra[i][P];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[i][2 * P + (e >> 2)][e & 3] = (float)t.h[e];
                } else {
                    const h16x8 t = as_h16x8(ra[i][P]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[i][2 * P + (e >> 2)][e & 3] = (float)t[e];
                }
            }
            if constexpr (NREP & 1) {
                if (RK == KIND_F16) {
                    union { u32x2 w; _Float16 h[4]; } t; t.w = rs[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][2 * NPAIR][e] = (float)t.h[e];
                } else {
                    Pack8 t; t.u = rs[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][2 * NPAIR][e] = (float)t.h[e];
                }
            }
        }
    } else {                                              // an operand matrix of the bf16x3 build: the sum of its pieces
        const h16* R = reinterpret_cast<const h16*>(p.R);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t m = mrow + 16 * i;
            const bool live = m < p.M;
#pragma unroll
            for (int P = 0; P < NPAIR; ++P) {
                float rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (live) load8_operand(R + m * p.ldr + np + wave_pair_col<NREP>(wc, P), p.ldr / PLANES, rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][2 * P + (e >> 2)][e & 3] = rr[e];
            }
            if constexpr (NREP & 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][2 * NPAIR][e] = live ? load1_operand(R + m * p.ldr + ns + e, p.ldr / PLANES) : 0.f;
            }
        }
    }
}

// Epilogue of a tile, straight from the accumulators (shared by the one-tile and the persistent kernel).
// Lane (pixel px = lane % 16, q = lane / 16) holds, for each of its 9 rows m = m0 + 144 wr + 16 i + px and each fragment pair p,
// the 8 consecutive output channels wave_pair_col(wc, p) + 8 q .. + 7: one 16-byte piece per row (operand / fp16 result; two for
// fp32), four lanes = 64 contiguous bytes of the row.  A piece is value = alpha acc + (bias + group bias), GEGLU'd, rounded, summed into the
// GroupNorm partials, stored.  A residual is already in the accumulators (w_seed): the epilogue fetches nothing.
// LUT: how the Phi table lies in `tail` — 2 = (value, step) pairs (gelu_lut2, 8 KiB), 1 = plain values (gelu_lut, 4 KiB: the two-workgroup
// kernel below has no room for the pairs); the same bits either way (the step is the same fp32 difference, taken once or per value).
// UASPEC = false: no alpha == 1 copy of the plain epilogue (the persistent plain kernel has no registers for two).
// DR (wq_kernel's deferred residual): the residual of the lane's pieces is in registers, RAW (sraw: the two 16-byte pieces of a row's
// fragment pairs, srs: the 8-byte piece of its unpaired fragment; fp16 stream or 16-bit operand storage), and is added HERE, after the bias —
// ((x w + bias) + r), the order of the 128 x 128 kernels.
template <int NREP, bool GEGLU, int LUT = 2, bool UASPEC = true, int NI = 9, bool DR = false>
__device__ __forceinline__ void w_epilogue(const MudgGemmDesc& p, f32x4 (&acc)[NI][NREP], const int m0, const int n0, const int tm, const int wr, const int wc,
                                           const int lane, const int tid, float* tail, const float* __restrict__ phi,
                                           const u32x4 (*sraw)[2] = nullptr, const u32x2* srs = nullptr) {
    constexpr int WBN = 64 * NREP, NPAIR = NREP / 2;
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
    const int px = lane & 15, q4 = lane >> 4;
    const float alpha = p.alpha;
    const int OK = p.out_fp32;
    const int64_t mrow = (int64_t)m0 + wr * (16 * NI) + px;    // row of i = 0
    float* sred = tail;                                  // [M half][tile column][2]: GroupNorm partials meet here
    const f32x2* phis = reinterpret_cast<const f32x2*>(tail);      // GEGLU: the Phi table as pairs
    const int64_t gb0 = p.gbias ? (int64_t)(m0 / p.rows_per_group) * p.N : 0;          // host-checked: one group per tile

    // uatag: alpha == 1 (the product is not scaled: one v_add instead of v_mul + v_add per value, the same bits); tabtag: GEGLU's Phi from the
    // table.  Both are decided ONCE per tile, outside the loops: with `phi ? table : polynomial` inside the per-value expression the
    // compiler kept a branch per value and every table read was followed by its own s_waitcnt — 72 serialised LDS round trips per lane,
    // the longest part of a K = 320 tile (round 6: DESIGN §3.2).  Here a row's eight reads are in flight together.
    auto piece8 = [&](auto ptag, auto uatag, auto tabtag, auto rtag) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;         // fragment pair: accumulator fragments 2 P (channels + 0..3) and 2 P + 1 (+ 4..7)
        constexpr bool UA = decltype(uatag)::value != 0, TAB = decltype(tabtag)::value != 0;
        const int cw = n0 + wave_pair_col<NREP>(wc, GEGLU ? 0 : P) + 8 * q4;          // first W row (bias index) of the value
        const int n = GEGLU ? (n0 >> 1) + wc * 32 + 8 * q4 : cw;                        // first output channel
        float bv[8], bg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bv[e] = p.bias ? p.bias[cw + e] : 0.f;
            if (p.gbias) bv[e] += p.gbias[gb0 + cw + e];
            bg[e] = (GEGLU && p.bias) ? p.bias[cw + 32 + e] : 0.f;
        }
        float gs[8], gq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t m = mrow + 16 * i;
            const bool live = m < p.M;
            float v[8];
            if constexpr (GEGLU) {
                float val[8], gate[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = acc[i][e >> 2][e & 3], b = acc[i][2 + (e >> 2)][e & 3];
                    val[e] = UA ? a + bv[e] : alpha * a + bv[e];
                    gate[e] = UA ? b + bg[e] : alpha * b + bg[e];
                }
                if constexpr (!TAB) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = val[e] * gelu_fast(gate[e]);
                } else if constexpr (PLANES == 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = val[e] * gelu_hermite(gate[e], phis);
                } else if constexpr (LUT == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = val[e] * gelu_lut(gate[e], tail);
                } else {                                  // gelu_lut2 in stages: eight indices, eight reads, eight interpolations
                    float u[8];
                    f32x2 t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] = __builtin_amdgcn_fmed3f(fmaf(gate[e], 64.0f, 512.0f), 0.0f, 1023.99f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = phis[(int)u[e]];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = val[e] * (gate[e] * fmaf(__builtin_amdgcn_fractf(u[e]), t[e][1], t[e][0]));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = acc[i][2 * P + (e >> 2)][e & 3];
                    v[e] = UA ? a + bv[e] : alpha * a + bv[e];
                }
                if constexpr (DR) {
                    constexpr int RT = decltype(rtag)::value;             // 1: fp16 stream, 2: operand storage
                    if constexpr (RT == 1) {
                        union { u32x4 w; f16x8 hh; } t; t.w = sraw[i][P];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)t.hh[e];
                    } else {
                        const h16x8 t = as_h16x8(sraw[i][P]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)t[e];
                    }
                }
            }
            if (!GEGLU && p.stats) {
                // what the store will hold, per storage kind — the kind decided once per row, not inside the per-value expression (there
                // it was a chain of scalar compares and branches per VALUE: round 6)
                float t[8];
                if (OK == KIND_F32) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = v[e];
                } else if (OK == KIND_F16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (float)f16_sat(v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = operand_round(v[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float tt = live ? t[e] : 0.f;
                    gs[e] += tt; gq[e] = fmaf(tt, tt, gq[e]);
                }
            }
            if (live) {
                const int64_t yoff = m * p.ldy + n;
                if (OK == KIND_F16) store8_f16(reinterpret_cast<_Float16*>(p.Y) + yoff, v);
                else if (OK == KIND_F32) {
                    float* yp = reinterpret_cast<float*>(p.Y) + yoff;
                    f32x4 a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; }
                    *reinterpret_cast<f32x4*>(yp) = a;
                    *reinterpret_cast<f32x4*>(yp + 4) = b;
                } else store8_operand(reinterpret_cast<h16*>(p.Y) + yoff, p.ldy / PLANES, v);
            }
        }
        if (!GEGLU && p.stats) {
            // the 16 pixels of the fragment column (a DPP row) folded in a fixed order; lane px = 0 of each q hands its M half's sums over
#pragma unroll
            for (int e = 0; e < 8; ++e) { gs[e] = row16_sum(gs[e]); gq[e] = row16_sum(gq[e]); }
            if (px == 0) {
                float* d = sred + ((wr * WBN) + wave_pair_col<NREP>(wc, P) + 8 * q4) * 2;
#pragma unroll
                for (int e = 0; e < 8; ++e) { d[2 * e] = gs[e]; d[2 * e + 1] = gq[e]; }
            }
        }
    };
    // the unpaired fifth fragment of the 320-wide tile: 4 consecutive channels per lane (8-byte operand / fp16 pieces)
    auto piece4 = [&](auto uatag, auto rtag) __attribute__((always_inline)) {
        constexpr int J = 2 * NPAIR;
        constexpr bool UA = decltype(uatag)::value != 0;
        const int n = n0 + wave_single_col<NREP>(wc) + 4 * q4;
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bv[e] = p.bias ? p.bias[n + e] : 0.f;
            if (p.gbias) bv[e] += p.gbias[gb0 + n + e];
        }
        float gs[4], gq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int64_t m = mrow + 16 * i;
            const bool live = m < p.M;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float a = acc[i][J < NREP ? J : 0][e]; v[e] = UA ? a + bv[e] : alpha * a + bv[e]; }
            if constexpr (DR) {
                if constexpr (decltype(rtag)::value == 1) {
                    union { u32x2 w; _Float16 hh[4]; } t; t.w = srs[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)t.hh[e];
                } else {
                    Pack8 t; t.u = srs[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)t.h[e];
                }
            }
            if (p.stats) {
                float t[4];
                if (OK == KIND_F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = v[e];
                } else if (OK == KIND_F16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = (float)f16_sat(v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = operand_round(v[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float tt = live ? t[e] : 0.f;
                    gs[e] += tt; gq[e] = fmaf(tt, tt, gq[e]);
                }
            }
            if (live) {
                const int64_t yoff = m * p.ldy + n;
                if (OK == KIND_F16) {
                    union { u32x2 w; _Float16 h[4]; } t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t.h[e] = f16_sat(v[e]);
                    *reinterpret_cast<u32x2*>(reinterpret_cast<_Float16*>(p.Y) + yoff) = t.w;
                } else if (OK == KIND_F32) {
                    f32x4 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = v[e];
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.Y) + yoff) = a;
                } else {
#pragma unroll
                    for (int pl = 0; pl < PLANES; ++pl) {               // the pieces of store8_operand, four channels wide
                        Pack8 t;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { t.h[e] = (h16)v[e]; v[e] -= (float)t.h[e]; }
                        *reinterpret_cast<u32x2*>(reinterpret_cast<h16*>(p.Y) + yoff + pl * (p.ldy / PLANES)) = t.u;
                    }
                }
            }
        }
        if (p.stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { gs[e] = row16_sum(gs[e]); gq[e] = row16_sum(gq[e]); }
            if (px == 0) {
                float* d = sred + ((wr * WBN) + wave_single_col<NREP>(wc) + 4 * q4) * 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) { d[2 * e] = gs[e]; d[2 * e + 1] = gq[e]; }
            }
        }
    };
    const bool ua = UASPEC && alpha == 1.f;
    using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>;
    if constexpr (GEGLU) {
        if (phi) { if (ua) piece8(T0{}, T1{}, T1{}, R0{}); else piece8(T0{}, T0{}, T1{}, R0{}); }
        else { if (ua) piece8(T0{}, T1{}, T0{}, R0{}); else piece8(T0{}, T0{}, T0{}, R0{}); }
    } else {
        auto pieces = [&](auto uatag, auto rtag) __attribute__((always_inline)) {
            piece8(T0{}, uatag, T0{}, rtag);
            if constexpr (NPAIR > 1) piece8(T1{}, uatag, T0{}, rtag);
            if constexpr (NREP & 1) piece4(uatag, rtag);
        };
        if constexpr (DR) {               // (alpha == 1: host-checked for every problem with a residual)
            if (p.res_fp32 == KIND_F16) pieces(T1{}, R1{}); else pieces(T1{}, R2{});
        } else if (ua) pieces(T1{}, R0{});
        else pieces(T0{}, R0{});
        if (p.stats) {
            // per 288-row block (= this tile) and channel: M half 0 + M half 1
            __syncthreads();
            if (tid < WBN) {
                const float s0 = sred[tid * 2] + sred[(WBN + tid) * 2], s1 = sred[tid * 2 + 1] + sred[(WBN + tid) * 2 + 1];
                *reinterpret_cast<f32x2*>(&p.stats[((int64_t)tm * p.N + n0 + tid) * 2]) = f32x2{s0, s1};
            }
        }
    }
}

struct KPos { int kt, tap, c; };                         // a K-tile: its index along W's K axis, its tap and first input channel

// NREP = 5: 288 x 320 (every channel count of the UNet is a multiple of 320); NREP = 4 + GEGLU: 288 x 256, a wave's four fragments are
// one [32 value | 32 gate] block of the packed GEGLU weights.
template <int MODE, int NREP, bool GEGLU>
__global__ __launch_bounds__(512, 2) void wgemm_kernel(const MudgGemmDesc p, const int vflags, const float* __restrict__ phi) {
    using G = WGeo<NREP>;
    constexpr int WBN = G::BN, W_KS = G::KS, W_BUF = G::BUF;
    // The 16-bit builds run six phases of 15 MFMAs per K-tile (table above), bf16x3 three phases of 45 per k half.  The three-phase loop also
    // runs the 16-bit operands (-DMUDG_W1_THREE: a unit = a K-tile, 30 MFMAs per phase, bit-identical results) and was measured: - 3 ... - 8 %
    // on every shape (profiles/r5/w288_16bit_phases.txt) — there a phase's loads (8 DMA pieces + 16 fragment reads) outlast its MFMAs.
#ifdef MUDG_W1_THREE
    constexpr bool SIX = false;
#else
    constexpr bool SIX = PLANES == 1;
#endif
    constexpr int NPAIR = NREP / 2;                      // fragment pairs whose 2 x 4 accumulator registers are 8 consecutive channels
    static_assert(!GEGLU || (MODE == 0 && NREP == 4), "GEGLU: plain GEMM on the 256-wide tile");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;             // waves wc and wc + 4 share a SIMD: the two M halves
    float* tail = reinterpret_cast<float*>(smem + G::LOOP);
    if (GEGLU && phi) {                                  // visible after the K loop's barriers
        for (int t = tid; t <= PHI_N; t += 512) {         // entry PHI_N exists (x = 8); its step is never used (u < 1024)
            const float a = phi[t], b = phi[t < PHI_N ? t + 1 : t];
            if constexpr (PLANES == 2) {                 // (value, slope x node distance): gelu_hermite
                const float xt = -8.0f + (float)t * (1.0f / 64.0f);
                *reinterpret_cast<f32x2*>(&tail[2 * t]) = f32x2{a, expf(-0.5f * xt * xt) * (0.3989422804014327f / 64.0f)};
            } else *reinterpret_cast<f32x2*>(&tail[2 * t]) = f32x2{a, b - a};
        }
    }

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
    // W rows are fetched PERMUTED, so that the epilogue can leave the accumulators in 16-byte pieces: subtile j of a wave column
    // (j = 2 p + odd) holds in its row r = 4 q + e the channel 32 p + 8 q + 4 odd + e of the wave's 16 NREP columns — lane (pixel, q) of
    // the MFMA result then owns, over the fragment pair p, the 8 CONSECUTIVE channels 32 p + 8 q .. + 7.  The fifth fragment of the
    // 320-wide tile has no partner and keeps its rows (4 consecutive channels per lane).  Free: a row permutation of the source.
    const unsigned vw_pair = (unsigned)((8 * (srow >> 2) + (srow & 3)) * p.ldw) * 2u + (unsigned)schunk * 16u;
    const unsigned vw_single = (unsigned)(srow * p.ldw) * 2u + (unsigned)schunk * 16u;
    // this wave's pieces per k half.  W (4 NREP subtiles per piece) over the eight waves as 2 3 3 2 | 2 3 3 2 (NREP 5) or two each
    // (NREP 4).  X: a row third (3 subtiles) per staging point over the waves wc = 0, 1, 2 (slot q of amask = third q); the six-phase
    // loop (SIX, kept for A/B measurements) — the X half wr (9 subtiles) over its four waves as 3 2 2 2
    const int a_first = SIX ? (wc == 0 ? 0 : 1 + 2 * wc) : wc, a_cnt = SIX ? (wc == 0 ? 3 : 2) : (wc < 3 ? 3 : 0);
    constexpr int a_step = SIX ? 1 : 3;
    const int b_cnt = NREP == 5 ? ((wc == 0 || wc == 3) ? 2 : 3) : 2;
    const int b_first = NREP == 5 ? wr * 10 + (wc == 0 ? 0 : (wc == 1 ? 2 : (wc == 2 ? 5 : 8))) : wave * 2;
    // validity of the lane's source row per tap (bit t): rows beyond M, taps that leave the image / the clip -> zero-filled by the DMA
    unsigned amask[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int m = m0 + (wr * 9 + a_first + q * a_step) * 16 + srow;
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
    W_STAMP(0);
    f32x4 acc[9][NREP];
    if (!GEGLU && p.R) w_seed<NREP, 9>(p, acc, m0, n0, wr, wc, lane);
    else {
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // fragment reads: a 16 x 32 fragment is one subtile; lane l holds row l % 16, 16-byte k chunk l / 16
    const int fbyte0 = (lane & 15) * 64 + (lane >> 4) * 16;
    const int fbyte = fbyte0 ^ (((fbyte0 >> 9) & 1) << 5);
    const char* a_base = smem + (wr * 9) * 1024 + fbyte;
    const char* b_base = smem + (WNA + wc * NREP) * 1024 + fbyte;

    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
    const int nk = p.K / BK;
    if constexpr (SIX) {
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
                        const int st = b_first + q, wcol = st / NREP, j = st - wcol * NREP;
                        const bool single = j >= 2 * NPAIR;
                        const int row0 = single ? wave_single_col<NREP>(wcol) : wave_pair_col<NREP>(wcol, j >> 1) + 4 * (j & 1);       // first channel of the piece
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(base + (WNA + st) * 1024), 16, (int)(single ? vw_single : vw_pair),
                                                                 soffw + row0 * p.ldw * 2, 0, 0);
                    }
            }
        };

        h16x8 af[3], bf[NREP];
        auto read_a = [&](int buf, int ks, int third) {
#pragma unroll
            for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const h16x8*>(a_base + buf * W_BUF + ks * W_KS + (third * 3 + i) * 1024);
        };
        auto read_b = [&](int buf, int ks) {
#pragma unroll
            for (int j = 0; j < NREP; ++j) bf[j] = *reinterpret_cast<const h16x8*>(b_base + buf * W_BUF + ks * W_KS + j * 1024);
        };
        auto mma = [&](auto third_tag) {
            constexpr int third = decltype(third_tag)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);              // (register-only MFMAs may otherwise be hoisted above the wait)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j)              // operands swapped: a lane ends up with 4 consecutive (permuted) channels of one pixel
                    acc[third * 3 + i][j] = mfma16(bf[j], af[i], acc[third * 3 + i][j]);
        };
        // "everything but the pieces issued after the k half that is about to be read": that half's successor (a_cnt + b_cnt pieces) plus
        // the W pieces of the one after (b_cnt, issued in this phase): 7 | 8 | 6 (NREP 5: wc 0 | 1, 2 | 3), 7 | 6 (NREP 4: wc 0 | others)
        auto wait_half = [&](bool more) {
            if (!more) W_VMCNT(0);
            else if (a_cnt + 2 * b_cnt == 8) W_VMCNT(8);
            else if (a_cnt + 2 * b_cnt == 7) W_VMCNT(7);
            else W_VMCNT(6);
        };

        KPos kA{0, 0, 0};
        stage(kA, 0, 0, 0);
        stage(kA, 1, 0, 0);
        advance(kA);                                         // kA = K-tile t + 1, kB = K-tile t + 2 at the top of iteration t
        if (nk > 1) stage(kA, 0, 1, 0);
        KPos kB = kA;
        advance(kB);
        if (nk <= 1) W_VMCNT(0); else if (a_cnt + b_cnt == 5) W_VMCNT(10); else W_VMCNT(8);      // ks 0 of tile 0 has landed
        W_BARRIER();
        W_STAMP(1);
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
    } else {
        // ---------------------------------------------------------------- the three-phase loop (header): a UNIT is what one ring slot holds — a
        // K-tile (16-bit builds: its two k halves are the slot's two pieces) or a k half (bf16x3: the two bf16 pieces of both operands)
        constexpr int W_PL = G::PL, W_SLOT = 2 * G::PL;
        const int wclass = wc == 3 ? 2 : ((NREP == 5 && wc != 0) ? 1 : 0);      // (W pieces, X pieces) per staging point and piece: (2,1) (3,1) (2,0)
        // Per-lane DMA state of the K loop in TWO registers (the loop keeps 244 accumulator and fragment registers): lanepk = source
        // row | permuted W row << 8 | 16-byte chunk offset << 16, maskpk = the three row thirds' tap masks, 9 bits each; the byte offsets
        // are rebuilt per piece (two extractions + one multiply-add).  opaque(): without it every staging point's offset is hoisted out
        // of the loop as one more live register, they spill, and a reload between two DMA issues drains the ring.
        const unsigned lanepk = (unsigned)srow | ((unsigned)(8 * (srow >> 2) + (srow & 3)) << 8) | ((unsigned)(schunk * 16) << 16);
        const unsigned maskpk = amask[0] | (amask[1] << 9) | (amask[2] << 18);
        auto lane_off = [&](int rowshift, int ld2) -> unsigned {           // rowshift 0: the X / unpaired-W row, 8: the permuted W row
            const unsigned lp = opaque(lanepk);
            return ((lp >> rowshift) & 0xffu) * (unsigned)ld2 + (lp >> 16);
        };
        // X subtile of row third `third` of this wave's M half, piece pc of the unit (K-tile k, k half kh: bf16x3 only — in the 16-bit builds
        // the piece IS the k half) -> ring slot `slot` (waves wc < 3)
        auto stage_x = [&](const KPos& k, int kh, int slot, int pc, auto third_tag) {
            constexpr int third = decltype(third_tag)::value;
            if (wc >= 3) return;
            const int ks = PLANES == 2 ? kh : pc, plane = PLANES == 2 ? pc : 0;
            char* base = smem + slot * W_SLOT + pc * W_PL;
            const bool s2 = k.c >= p.csplit;
            const int cc = s2 ? k.c - p.csplit : k.c;
            const int ld = s2 ? ldx2e : p.ldx;
            int soff = (cc + ks * 32) * 2 + plane * ld;          // bf16 piece 1 of a row: ld / 2 elements further
            if (MODE == 1) { const int dy = k.tap / 3, dx = k.tap - 3 * dy; soff += (dy * p.Win + dx) * ld * 2; }
            if (MODE == 2) soff += k.tap * p.HW * ld * 2;
            const int st = wr * 9 + 3 * third + wc;
            const unsigned v = ((opaque(maskpk) >> (9 * third + k.tap)) & 1u) ? lane_off(0, ld * 2) : OOB;
            if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, (lptr_t)(base + st * 1024), 16, (int)v, soff + st * 16 * ld * 2, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lptr_t)(base + st * 1024), 16, (int)v, soff + st * 16 * ld * 2, 0, 0);
        };
        auto stage_w = [&](const KPos& k, int kh, int slot, int pc) {
            const int ks = PLANES == 2 ? kh : pc, plane = PLANES == 2 ? pc : 0;
            char* base = smem + slot * W_SLOT + pc * W_PL;
            const int soffw = (k.kt * BK + ks * 32) * 2 + plane * p.ldw;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < b_cnt) {
                    const int st = b_first + q, wcol = st / NREP, j = st - wcol * NREP;
                    const bool single = j >= 2 * NPAIR;
                    const int row0 = single ? wave_single_col<NREP>(wcol) : wave_pair_col<NREP>(wcol, j >> 1) + 4 * (j & 1);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(base + (WNA + st) * 1024), 16, (int)lane_off(single ? 0 : 8, p.ldw * 2),
                                                             soffw + row0 * p.ldw * 2, 0, 0);
                }
        };
        // both pieces of a staging point (piece 1 first: in bf16x3 it is read first)
        auto stage_x2 = [&](const KPos& k, int kh, int slot, auto third_tag) { stage_x(k, kh, slot, 1, third_tag); stage_x(k, kh, slot, 0, third_tag); };
        auto stage_w2 = [&](const KPos& k, int kh, int slot) { stage_w(k, kh, slot, 0); stage_w(k, kh, slot, 1); };
        // Counted waits.  With b W pieces and x X pieces per (piece, staging point), what a wave has issued AFTER the pieces the phase two
        // ahead will read (header table, issue order of the loop below): kind 0 = 2 b + 4 x, kind 1 = 4 x.  Near the end of K staging
        // points are skipped: then everything is awaited.
        auto wait3 = [&](int kind, bool counted) {
            if (!counted) { W_VMCNT(0); return; }
            if (wclass == 0) { if (kind == 0) W_VMCNT(8); else W_VMCNT(4); }
            else if (wclass == 1) { if (kind == 0) W_VMCNT(10); else W_VMCNT(4); }
            else { if (kind == 0) W_VMCNT(4); else W_VMCNT(0); }
        };
        h16x8 bw0[NREP], bw1[NREP], a0[3], a1[3];
        const int a_off = (int)(a_base - smem), b_off = (int)(b_base - smem);
        auto read_w = [&](h16x8 (&dst)[NREP], const char* sb, int pc) {
#pragma unroll
            for (int j = 0; j < NREP; ++j) dst[j] = *reinterpret_cast<const h16x8*>(sb + b_off + pc * W_PL + j * 1024);
        };
        auto read_x = [&](h16x8 (&dst)[3], const char* sb, int pc, int third) {
#pragma unroll
            for (int i = 0; i < 3; ++i) dst[i] = *reinterpret_cast<const h16x8*>(sb + a_off + pc * W_PL + (third * 3 + i) * 1024);
        };
        auto mma3 = [&](auto third_tag, const h16x8 (&b)[NREP], const h16x8 (&a)[3]) {
            constexpr int third = decltype(third_tag)::value;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) mfma16_inplace(b[j], a[i], acc[third * 3 + i][j]);
        };
        // a phase: fragments of row third `third` (and, first, the unit's W fragments) | barrier | 30 (16-bit: k half 0, then k half 1 — the
        // K order of the 128 x 128 kernels) or 45 MFMAs (bf16x3: x1 w0, x0 w0, x0 w1)
        auto reads = [&](const char* sb, int third, bool with_w) {
            if (with_w) read_w(bw0, sb, 0);
            read_x(a1, sb, 1, third);
            read_x(a0, sb, 0, third);
            if (with_w) read_w(bw1, sb, 1);
        };
        auto multiply = [&](auto third_tag) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PLANES == 2) { mma3(third_tag, bw0, a1); mma3(third_tag, bw0, a0); mma3(third_tag, bw1, a0); }
            else { mma3(third_tag, bw0, a0); mma3(third_tag, bw1, a1); }
        };

        const int NU = PLANES * nk;                          // units
        KPos kC{0, 0, 0}, kN{0, 0, 0};                       // bf16x3: the K-tile of unit u and the next one; 16-bit: the K-tiles of units u + 1 and u + 2
        if constexpr (PLANES == 2) { advance(kN); } else { advance(kC); kN = kC; advance(kN); }
        {   // units 0 and 1, in the order the loop would have issued them (the counted waits rely on it)
            const KPos k0{0, 0, 0};
            stage_w2(k0, 0, 0); stage_x2(k0, 0, 0, T0{});
            stage_x2(k0, 0, 0, T1{});
            stage_x2(k0, 0, 0, T2{});
            if (NU > 1) { const KPos& k1 = PLANES == 2 ? k0 : kC; stage_w2(k1, 1, 1); stage_x2(k1, 1, 1, T0{}); }
        }
        wait3(0, NU > 1);                                    // W and the X rows 0-5 of unit 0 have landed
        W_BARRIER();
        if (wr == 1) W_BARRIER();                            // the stagger: M-half 1 runs one barrier behind M-half 0
        for (int u = 0; u < NU; ++u) {
            const int slot = u & 1;
            const bool n1 = u + 1 < NU, n2 = u + 2 < NU;
            const char* sb = smem + slot * W_SLOT;
            const KPos k1 = PLANES == 2 ? (slot ? kN : kC) : kC;     // unit u + 1 (bf16x3: its k half is slot ^ 1); unit u + 2 is (kN, slot)
            reads(sb, 0, true);
            if (n1) stage_x2(k1, slot ^ 1, slot ^ 1, T1{});
            W_BARRIER(); multiply(T0{}); wait3(0, n1); W_BARRIER();
            reads(sb, 1, false);
            if (n1) stage_x2(k1, slot ^ 1, slot ^ 1, T2{});
            W_BARRIER(); multiply(T1{}); wait3(1, n1); W_BARRIER();
            reads(sb, 2, false);
            if (n2) { stage_w2(kN, slot, slot); stage_x2(kN, slot, slot, T0{}); }
            W_BARRIER(); multiply(T2{}); wait3(0, n2); W_BARRIER();
            if (PLANES == 1 || slot) { kC = kN; advance(kN); }
        }
        // the MFMAs above are inline assembly: the compiler does not know that the epilogue's first reads depend on matrix results
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
    if (wr == 0) W_BARRIER();                            // evens out the stagger: every fragment read has retired, every DMA has landed
    W_STAMP(2);

    w_epilogue<NREP, GEGLU>(p, acc, m0, n0, tm, wr, wc, lane, tid, tail, phi);
    W_STAMP(3);
}

#if MUDG_PLANES == 1
// ---------------------------------------------------------------------------------------------------------------------------------------
// The persistent form for plain GEMMs / GEGLU (MODE 0, whole 288-row tiles, K >= 128): one workgroup per CU walks every gridDim / 8-th
// tile of its XCD's range, and the K-tiles of its tiles form ONE stream through the LDS ring — while the last K-tiles of a tile are
// multiplied the first 1.5 K-tiles of the next tile are already being staged, the epilogue (accumulators -> HBM, no LDS) runs with the
// ring full, and its stores drain under the next tile's first phases.  What a one-tile workgroup pays per tile and this form does not:
// the launch of a workgroup, the first-fetch latency, the drain of the stores before the CU is handed on — at K = 320 more than the
// K loop itself — and the lockstep of 256 CUs that all fetch, then all multiply, then all store.  Per tile the arithmetic is the
// one-tile kernel's, instruction for instruction: the same bits.
// Counted waits across an epilogue: the two waits of a tile's first K-tile retire pieces issued BEFORE the epilogue, so the epilogue's
// stores (27 | 45 per lane: 16-bit | fp32 results; 9 | 18 GEGLU) are younger than what they wait for and are added to the count.
#define W_VMCNT_CASE(n) case n: W_VMCNT(n); break;
__device__ __forceinline__ void w_vmcnt_runtime(int n) {          // n: wave-uniform
    switch (n) {
        W_VMCNT_CASE(15) W_VMCNT_CASE(16) W_VMCNT_CASE(24) W_VMCNT_CASE(25)
        W_VMCNT_CASE(33) W_VMCNT_CASE(34) W_VMCNT_CASE(35) W_VMCNT_CASE(51) W_VMCNT_CASE(52) W_VMCNT_CASE(53)
        default: W_VMCNT(0); break;
    }
}

template <int NREP, bool GEGLU>
__global__ __launch_bounds__(512, 2) void wgemm_pkernel(const MudgGemmDesc p, const int vflags, const float* __restrict__ phi, const int ntiles) {
    using G = WGeo<NREP>;
    constexpr int WBN = G::BN, W_KS = G::KS, W_BUF = G::BUF, NPAIR = NREP / 2;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    float* tail = reinterpret_cast<float*>(smem + G::LOOP);
    if (GEGLU && phi) {
        for (int t = tid; t <= PHI_N; t += 512) {         // entry PHI_N exists (x = 8); its step is never used (u < 1024)
            const float a = phi[t], b = phi[t < PHI_N ? t + 1 : t];
            if constexpr (PLANES == 2) {                 // (value, slope x node distance): gelu_hermite
                const float xt = -8.0f + (float)t * (1.0f / 64.0f);
                *reinterpret_cast<f32x2*>(&tail[2 * t]) = f32x2{a, expf(-0.5f * xt * xt) * (0.3989422804014327f / 64.0f)};
            } else *reinterpret_cast<f32x2*>(&tail[2 * t]) = f32x2{a, b - a};
        }
    }
    // this workgroup's tiles: the XCD's contiguous range of the one-tile kernel's numbering, every (gridDim / 8)-th tile of it
    const int ntn = p.N / WBN, ntm = p.M / WBM;
    const int xcd = blockIdx.x & 7, stride = gridDim.x >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, count = q8 + (xcd < r8 ? 1 : 0);
    int li = blockIdx.x >> 3;                              // (host: ntiles >= gridDim, so every workgroup has a tile)
    const h16* X = reinterpret_cast<const h16*>(p.X);
    const h16* X2 = p.X2 ? reinterpret_cast<const h16*>(p.X2) : nullptr;
    const h16* W = reinterpret_cast<const h16*>(p.W);
    const int ldx2e = X2 ? p.ldx2 : p.ldx;
    // ONE descriptor per operand for the whole problem: a tile's first row rides in the scalar offset of its pieces (host: the 32-bit reach)
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(X), rX2 = X2 ? make_rsrc(X2) : rX, rW = make_rsrc(W);
    struct Tile { int m0, n0, tm; };
    auto locate = [&](int local) {
        const int tile = first + local;
        const int per = 8 * ntn, g = tile / per, f8 = g * 8;
        const int gsz = (ntm - f8) < 8 ? (ntm - f8) : 8;
        const int r = tile - g * per;
        const int tn = r / gsz, tm = f8 + (r - tn * gsz);
        return Tile{tm * WBM, tn * WBN, tm};
    };
    Tile cur = locate(li), nxt = cur;
    if (li + stride < count) nxt = locate(li + stride);

    const int pos = lane * 16;
    const int sbyte = pos ^ (((pos >> 9) & 1) << 5);
    const int srow = sbyte >> 6, schunk = (sbyte >> 4) & 3;
    const unsigned va1 = (unsigned)(srow * p.ldx) * 2u + (unsigned)schunk * 16u;
    const unsigned va2 = (unsigned)(srow * ldx2e) * 2u + (unsigned)schunk * 16u;
    const unsigned vw_pair = (unsigned)((8 * (srow >> 2) + (srow & 3)) * p.ldw) * 2u + (unsigned)schunk * 16u;
    const unsigned vw_single = (unsigned)(srow * p.ldw) * 2u + (unsigned)schunk * 16u;
    const int a_first = wc == 0 ? 0 : 1 + 2 * wc, a_cnt = wc == 0 ? 3 : 2;
    const int b_cnt = NREP == 5 ? ((wc == 0 || wc == 3) ? 2 : 3) : 2;
    const int b_first = NREP == 5 ? wr * 10 + (wc == 0 ? 0 : (wc == 1 ? 2 : (wc == 2 ? 5 : 8))) : wave * 2;
    // pieces of k half ks of K-tile kt of the current (sel 0) or the next (sel 1) tile -> ring buffer buf; part as in the one-tile kernel
    auto stage = [&](int sel, int kt, int ks, int buf, int part) {
        char* base = smem + buf * W_BUF + ks * W_KS;
        if (part != 2) {
            const int c = kt * BK;
            const bool s2 = c >= p.csplit;
            const int cc = s2 ? c - p.csplit : c;
            const int ld = s2 ? ldx2e : p.ldx;
            const int soff = (cc + ks * 32) * 2 + (sel ? nxt.m0 : cur.m0) * ld * 2;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < a_cnt) {
                    const int st = wr * 9 + a_first + q;
                    if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, (lptr_t)(base + st * 1024), 16, (int)va2, soff + st * 16 * ld * 2, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lptr_t)(base + st * 1024), 16, (int)va1, soff + st * 16 * ld * 2, 0, 0);
                }
        }
        if (part != 1) {
            const int soffw = (kt * BK + ks * 32) * 2 + (sel ? nxt.n0 : cur.n0) * p.ldw * 2;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < b_cnt) {
                    const int st = b_first + q, wcol = st / NREP, j = st - wcol * NREP;
                    const bool single = j >= 2 * NPAIR;
                    const int row0 = single ? wave_single_col<NREP>(wcol) : wave_pair_col<NREP>(wcol, j >> 1) + 4 * (j & 1);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(base + (WNA + st) * 1024), 16, (int)(single ? vw_single : vw_pair),
                                                             soffw + row0 * p.ldw * 2, 0, 0);
                }
        }
    };
    const int fbyte0 = (lane & 15) * 64 + (lane >> 4) * 16;
    const int fbyte = fbyte0 ^ (((fbyte0 >> 9) & 1) << 5);
    const char* a_base = smem + (wr * 9) * 1024 + fbyte;
    const char* b_base = smem + (WNA + wc * NREP) * 1024 + fbyte;
    f32x4 acc[9][NREP];
    h16x8 af[3], bf[NREP];
    auto read_a = [&](int buf, int ks, int third) {
#pragma unroll
        for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const h16x8*>(a_base + buf * W_BUF + ks * W_KS + (third * 3 + i) * 1024);
    };
    auto read_b = [&](int buf, int ks) {
#pragma unroll
        for (int j = 0; j < NREP; ++j) bf[j] = *reinterpret_cast<const h16x8*>(b_base + buf * W_BUF + ks * W_KS + j * 1024);
    };
    auto mma = [&](auto third_tag) {
        constexpr int third = decltype(third_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc[third * 3 + i][j] = mfma16(bf[j], af[i], acc[third * 3 + i][j]);
    };
    const int base_cnt = a_cnt + 2 * b_cnt;                // the one-tile kernel's wait_half count: 7 | 8 | 6
    const int nst = GEGLU ? (p.out_fp32 == KIND_F32 ? 18 : 9) : (p.out_fp32 == KIND_F32 ? 45 : 27);       // epilogue stores per lane
    auto wait_half = [&](bool more, bool fresh) {
        if (!more) W_VMCNT(0);
        else if (fresh) w_vmcnt_runtime(base_cnt + nst);
        else if (base_cnt == 8) W_VMCNT(8);
        else if (base_cnt == 7) W_VMCNT(7);
        else W_VMCNT(6);
    };
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
    const int nk = p.K / BK;                               // >= 2 (host)

    W_STAMP(0);
    [[maybe_unused]] int stamp_tile = 0;
    stage(0, 0, 0, 0, 0);
    stage(0, 0, 1, 0, 0);
    stage(0, 1, 0, 1, 0);
    if (a_cnt + b_cnt == 5) W_VMCNT(10); else W_VMCNT(8);  // ks 0 of the first K-tile has landed
    W_BARRIER();
    if (wr == 1) W_BARRIER();                              // the stagger: M-half 1 runs one barrier behind M-half 0
    int buf = 0;
    bool fresh = false;                                    // the K-tile that follows an epilogue
    for (;;) {
        const bool has_next = li + stride < count;
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int t = 0; t < nk; ++t, buf ^= 1) {
            const bool in1 = t + 1 < nk, in2 = t + 2 < nk;
            const bool n1 = in1 || has_next, n2 = in2 || has_next;
            const int sel1 = in1 ? 0 : 1, kt1 = in1 ? t + 1 : 0;
            const int sel2 = in2 ? 0 : 1, kt2 = in2 ? t + 2 : t + 2 - nk;
            const bool fr = fresh && t == 0;
            read_b(buf, 0);
            read_a(buf, 0, 0);
            W_BARRIER();
            mma(T0{});
            W_BARRIER();
            read_a(buf, 0, 1);
            if (n1) stage(sel1, kt1, 1, buf ^ 1, 2);
            W_BARRIER();
            mma(T1{});
            wait_half(n1, fr);
            W_BARRIER();
            read_a(buf, 0, 2);
            if (n1) stage(sel1, kt1, 1, buf ^ 1, 1);
            W_BARRIER();
            mma(T2{});
            W_BARRIER();
            read_b(buf, 1);
            read_a(buf, 1, 0);
            W_BARRIER();
            mma(T0{});
            W_BARRIER();
            read_a(buf, 1, 1);
            if (n2) stage(sel2, kt2, 0, buf, 2);
            W_BARRIER();
            mma(T1{});
            wait_half(n2, fr);
            W_BARRIER();
            read_a(buf, 1, 2);
            if (n2) stage(sel2, kt2, 0, buf, 1);
            W_BARRIER();
            mma(T2{});
            W_BARRIER();
        }
        if (wr == 0) W_BARRIER();                          // evens out the stagger
        __builtin_amdgcn_sched_barrier(0);
        W_STAMP(1 + 2 * stamp_tile);
        w_epilogue<NREP, GEGLU, 2, GEGLU>(p, acc, cur.m0, cur.n0, cur.tm, wr, wc, lane, tid, tail, phi);
        __builtin_amdgcn_sched_barrier(0);
        W_STAMP(2 + 2 * stamp_tile);
        ++stamp_tile;
        if (!has_next) break;
        cur = nxt;
        li += stride;
        if (li + stride < count) nxt = locate(li + stride);
        fresh = true;
        if (wr == 1) W_BARRIER();                          // the stagger again
    }
}
#endif

#if MUDG_PLANES == 1
// ---------------------------------------------------------------------------------------------------------------------------------------
// The half-height GEGLU tile (round 6): 144 x 256, FOUR waves, TWO workgroups per CU.
// Why.  The eight-wave tile is alone on its CU: while it runs its epilogue — GEGLU's is the longest, a table lookup and ~12 vector
// instructions per output value, as long as the whole K loop at K = 320 — and while it waits for its first K-tile, the matrix pipes idle
// (profiles/r5/pmc_mfma.md: 39 % MFMA-busy in the persistent GEGLU form).  Here a CU holds two INDEPENDENT workgroups of half the height:
// each SIMD has one wave of either, the per-wave work is the eight-wave kernel's (9 x 4 fragments of 16 x 16, 144 accumulator
// registers), and nothing couples the two — one multiplies while the other stores, fetches or sits in its barrier.
// LDS per workgroup (79.0 KiB; 2 x 80 KiB is the CU): a RING OF THREE k halves (32 deep) x {X: 9, W: 16 subtiles of 1 KiB, st_16x32
// swizzled at the DMA source as above} = 75 KiB + the Phi table as plain values (4 KiB: the pairs of the eight-wave kernel do not fit).
// Per k half h (slot h % 3), ONE barrier:
//     s_waitcnt vmcnt(pieces of h + 1)   this wave's pieces of h have landed
//     s_barrier                          so have everybody's; and everybody has finished reading k half h - 1 ...
//     DMA of k half h + 2                ... whose slot these pieces go to (two k halves = 72 MFMAs of latency budget)
//     fragments W (4), X rows 0-2 (3), X rows 3-5 (3) | 12 MFMAs | X rows 6-8 | 12 MFMAs | 12 MFMAs
// (a wave's own fragment reads of the next row third are in flight under its MFMAs; the compiler's counted lgkmcnt waits order them).
// Bits: the K order of every other kernel and gelu_lut on the same table — identical to the eight-wave tile and the 128 x 128 kernels,
// so the selection rule may look at M.
constexpr int H_NA = 9, H_NB = 16;
constexpr int H_KS = (H_NA + H_NB) * 1024;               // one k half: 25 KiB
constexpr int H_RING = 3 * H_KS;
constexpr int H_SMEM = H_RING + ((PHI_N + 1) * 4 + 15) / 16 * 16;

template <bool PF>
__global__ __launch_bounds__(256, 2) void hgeglu_kernel(const MudgGemmDesc p, const int vflags, const float* __restrict__ phi, const int first_round, const int delay) {
    constexpr int HBM_ = 16 * H_NA, HBN = 256;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // ANTI-PHASE START.  All workgroups of a launch take the same time, and the two that share a CU start together: left alone they run
    // in lockstep for the whole launch — both fetching, both multiplying, both in the epilogue at the same moments — and a CU with two
    // workgroups behaves like one with a single twice as large.  The workgroup that got the SECOND wave slot of its SIMDs in the first
    // round of the launch (HW_ID.wave_id odd) therefore starts `delay` x 8128 cycles late; every later workgroup inherits the phase of the
    // one whose slot it takes over.
    if ((delay & 0xffff) > 0 && (int)blockIdx.x < first_round) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID (4), offset 0, width 4: wave_id
        if (hw & 1u)
            for (int i = 0; i < (delay & 0xffff); ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave = wave column: 64 of the tile's 256 W rows
    float* tail = reinterpret_cast<float*>(smem + H_RING);
    if (phi) for (int t = tid; t <= PHI_N; t += 256) tail[t] = phi[t];        // visible after the K loop's barriers

    // XCD-aware tile numbering (as wgemm_kernel): every XCD a contiguous tile range, walked in 8-row groups column by column
    const int ntn = p.N / HBN, ntm = (p.M + HBM_ - 1) / HBM_;
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
    const int m0 = tm * HBM_, n0 = tn * HBN;

    const int pos = lane * 16;
    const int sbyte = pos ^ (((pos >> 9) & 1) << 5);
    const int srow = sbyte >> 6, schunk = (sbyte >> 4) & 3;
    const h16* X = reinterpret_cast<const h16*>(p.X);
    const h16* X2 = p.X2 ? reinterpret_cast<const h16*>(p.X2) : nullptr;
    const h16* W = reinterpret_cast<const h16*>(p.W);
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(X + (int64_t)m0 * p.ldx);
    const __amdgpu_buffer_rsrc_t rX2 = X2 ? make_rsrc(X2 + (int64_t)m0 * p.ldx2) : rX;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(W + (int64_t)n0 * p.ldw);
    const int ldx2e = X2 ? p.ldx2 : p.ldx;
    const unsigned va1 = (unsigned)(srow * p.ldx) * 2u + (unsigned)schunk * 16u;
    const unsigned va2 = (unsigned)(srow * ldx2e) * 2u + (unsigned)schunk * 16u;
    const unsigned vw_pair = (unsigned)((8 * (srow >> 2) + (srow & 3)) * p.ldw) * 2u + (unsigned)schunk * 16u;     // (permuted W rows: wgemm_kernel)
    // this wave's pieces per k half: its own 4 W subtiles; of the 9 X subtiles 3 | 2 | 2 | 2
    const int a_first = wc == 0 ? 0 : 1 + 2 * wc, a_cnt = wc == 0 ? 3 : 2;
    unsigned alive = 0;                                   // bit q: the lane's source row of X piece q exists
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < a_cnt && m0 + (a_first + q) * 16 + srow < p.M) alive |= 1u << q;
    auto stage = [&](int h, int slot) {
        char* base = smem + slot * H_KS;
        const int c = (h >> 1) * BK;
        const bool s2 = c >= p.csplit;
        const int cc = s2 ? c - p.csplit : c;
        const int ld = s2 ? ldx2e : p.ldx;
        const int soff = (cc + (h & 1) * 32) * 2;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < a_cnt) {
                const int st = a_first + q;
                const unsigned v = ((alive >> q) & 1u) ? (s2 ? va2 : va1) : OOB;
                if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, (lptr_t)(base + st * 1024), 16, (int)v, soff + st * 16 * ld * 2, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lptr_t)(base + st * 1024), 16, (int)v, soff + st * 16 * ld * 2, 0, 0);
            }
        const int soffw = h * 64;                          // k half h of W's K axis: 32 elements each
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row0 = wave_pair_col<4>(wc, j >> 1) + 4 * (j & 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(base + (H_NA + wc * 4 + j) * 1024), 16, (int)vw_pair, soffw + row0 * p.ldw * 2, 0, 0);
        }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fbyte0 = (lane & 15) * 64 + (lane >> 4) * 16;
    const int fbyte = fbyte0 ^ (((fbyte0 >> 9) & 1) << 5);
    const char* a_base = smem + fbyte;
    const char* b_base = smem + (H_NA + wc * 4) * 1024 + fbyte;

#ifdef MUDG_DEBUG_VARIANTS
    const int abl = (delay >> 16) & 3;                    // measurements only (GEMM_H144ABL): 1 = no K loop, 2 = no epilogue, 3 = neither
    const int NH = (abl & 1) ? 0 : 2 * (p.K / BK);
#else
    const int NH = 2 * (p.K / BK);                        // k halves (>= 2, even)
#endif
    W_STAMP(0);
    if (NH > 0) {
        stage(0, 0);
        stage(1, 1);
    }
    auto wait_landed = [&](bool more) {                   // this wave's pieces of the k half about to be read: all but the a_cnt + 4 of the next one
        if (more) { if (a_cnt == 3) W_VMCNT(7); else W_VMCNT(6); }
        else W_VMCNT(0);
    };
    auto mma3 = [&](auto third_tag, const h16x8 (&b)[4], const h16x8 (&a)[3]) {
        constexpr int third = decltype(third_tag)::value;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[third * 3 + i][j] = mfma16(b[j], a[i], acc[third * 3 + i][j]);
    };
    auto read_b = [&](h16x8 (&dst)[4], int sl) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const h16x8*>(b_base + sl * H_KS + j * 1024);
    };
    auto read_a = [&](h16x8 (&dst)[3], int sl, int third) {
#pragma unroll
        for (int i = 0; i < 3; ++i) dst[i] = *reinterpret_cast<const h16x8*>(a_base + sl * H_KS + (third * 3 + i) * 1024);
    };
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
    // The K loop runs at raised priority: when both workgroups of a CU have an instruction ready, the one that feeds the matrix pipe goes
    // first and the other one's epilogue arithmetic fills the slots in between (GEMM_H144PRIO = 0 in the variant build: off).
    if ((delay >> 20) == 0) __builtin_amdgcn_s_setprio(3);
    if constexpr (!PF) {
        int slot = 0;
        h16x8 bf[4], a0[3], a1[3];
#pragma unroll 1
        for (int h = 0; h < NH; ++h) {
            wait_landed(h + 1 < NH);
            W_BARRIER();
            if (h + 2 < NH) stage(h + 2, slot == 0 ? 2 : slot - 1);
            read_b(bf, slot); read_a(a0, slot, 0); read_a(a1, slot, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma3(T0{}, bf, a0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(a0, slot, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma3(T1{}, bf, a1);
            __builtin_amdgcn_sched_barrier(0);
            mma3(T2{}, bf, a0);
            __builtin_amdgcn_sched_barrier(0);
            slot = slot == 2 ? 0 : slot + 1;
        }
    } else {
        // The same ring with the NEXT k half's first fragments (W, X rows 0-2) requested before the last row third of this one is
        // multiplied: the wait for the next k half, the barrier and the DMA issue sit between the second and the third row third, and a
        // wave never starts a k half by waiting for its own ds_reads.  W fragments alternate between two register sets (k halves come in
        // pairs), X fragments rotate through three.
        h16x8 bA[4], bB[4], ax[3], ay[3], az[3];
        if (NH > 0) {
            wait_landed(true);
            W_BARRIER();
            W_STAMP(1);
            if (NH > 2) stage(2, 2);
            read_b(bA, 0); read_a(ax, 0, 0);
        }
        int slot = 0;                                      // slot of k half h
        // Every wait is a FULL lgkmcnt(0) for fragments requested one MFMA group (12 MFMAs, ~200 cycles) earlier, placed BEFORE the next
        // group's reads are issued — the builtin (not inline asm), so that the compiler's own wait insertion knows the fragments have
        // arrived and adds nothing after the reads (its counted waits merge conservatively across the `last k half` branch).
        auto half = [&](int h, const h16x8 (&bc)[4], h16x8 (&bn)[4]) {
            const int nslot = slot == 2 ? 0 : slot + 1;
            H_LGKM0();                                     // W fragments and X rows 0-2 of this k half
            read_a(ay, slot, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma3(T0{}, bc, ax);
            __builtin_amdgcn_sched_barrier(0);
            H_LGKM0();                                     // X rows 3-5
            read_a(az, slot, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma3(T1{}, bc, ay);
            __builtin_amdgcn_sched_barrier(0);
            H_LGKM0();                                     // X rows 6-8: every read of this k half has returned, its slot may be re-staged
            if (h + 1 < NH) {
                wait_landed(h + 2 < NH);
                W_BARRIER();
                if (h + 3 < NH) stage(h + 3, slot);
                read_b(bn, nslot); read_a(ax, nslot, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma3(T2{}, bc, az);
            __builtin_amdgcn_sched_barrier(0);
            slot = nslot;
        };
#pragma unroll 1
        for (int h = 0; h < NH; h += 2) {
            half(h, bA, bB);
            half(h + 1, bB, bA);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    W_STAMP(2);
#ifdef MUDG_DEBUG_VARIANTS
    if (abl & 2) {                                         // keep the accumulators alive without an epilogue
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 12345.678f) reinterpret_cast<float*>(p.Y)[0] = t;
        return;
    }
    if (NH == 0) __syncthreads();                          // (the table copy is otherwise made visible by the K loop's barriers)
#endif
    w_epilogue<4, true, 1>(p, acc, m0, n0, tm, 0, wc, lane, tid, tail, phi);
    W_STAMP(3);
}
#endif

#if MUDG_PLANES == 1
// ---------------------------------------------------------------------------------------------------------------------------------------
// The 160-row tile (round 6): 160 x 320 (GEGLU: 160 x 256), eight waves as 2 (M) x 4 (N), one workgroup per CU — for the resolutions whose
// frames are no multiple of 288 rows but of 160: MDM512 (BASELINE configs[1]) has 2560 / 640 / 160 pixels per frame = 16 / 4 / 1 tiles, a
// guidance batch of 2 x 16 frames 512 / 128 / 32 tiles per 320 columns — whole rounds of the 256 CUs at level 0 where the 288-row tile
// would leave 1.11 rounds and the 128 x 128 kernels run 1920 tiles of which every third column is half empty (N = 320 = 2.5 x 128).
// A wave owns 80 x 80 = 5 x 5 fragments (100 accumulators); 107 FLOP per staged byte (288 x 320: 151, 128 x 128: 64).
// LDS: a RING OF FIVE k halves (32 deep) x {X: 10, W: 4 NREP subtiles of 1 KiB, st_16x32 swizzled at the DMA source as in wgemm_kernel}
// = 150 KiB (GEGLU: 130 + the Phi pairs).  The 30 | 26 pieces of a k half go over the eight waves as pieces wave, wave + 8, wave + 16,
// wave + 24: four per wave (pieces beyond the 30 | 26 zero-fill a spare KiB).  Per k half h, ONE barrier (the skeleton of hgeglu_kernel, with the fragments double-buffered):
//     s_waitcnt vmcnt(4 c)       this wave's pieces of k half h + 1 have landed (c = k halves issued after it: 2, at the end of K 1, 0)
//     s_barrier                  so have everybody's; and everybody's fragment reads of k half h - 1 have returned ...
//     DMA of k half h + 4        ... whose slot these pieces go to: three k halves (75 MFMAs per wave, ~ 5000 cycles) to land
//     s_waitcnt lgkmcnt(0)       the fragments of k half h (requested under the MFMAs of k half h - 1)
//     fragments of k half h + 1  (5 W + 5 X reads into the other register set) | 25 MFMAs of k half h
// Bits: the K order and the per-row arithmetic of every other contraction kernel; without a residual identical to the 128 x 128 kernels,
// and with a 16-bit one too (it is added by the epilogue, after the bias: RS = 2 below); an fp32 residual seeds the accumulators as on the
// 288-row tile — ((r + x w) + bias) — so the rule (mudg_wgemm_rows) looks at the frame geometry, never at M.
// The same skeleton carries TWO tile heights (NI = 16-row fragments per wave and M half): NI = 5 — 160 rows, above — and, in the variant
// builds only (GEMM_W288Q, a measurement: mudg_wgemm_launch), NI = 9 — 288 rows, the tile of wgemm_kernel on this loop: what took the 160-row
// loop from 1050 to 1290 TFLOP/s (fragment reads between the MFMAs, no branch in the steady state, one barrier per k half) applied to the
// tile with 151 FLOP per staged byte — same bits, same speed as the six-phase loop (profiles/r6/w288q_shapes.txt).  At 180
// accumulators there is no room for a second fragment set, so the fragments are refreshed IN PLACE: a row's X fragment is requested for
// the next k half as soon as its five MFMAs are issued, a W fragment after its last use in the last row — each has at least five MFMAs
// (plus the barrier and the DMA issue) to arrive.  Ring of FOUR k halves of 38 | 34 KiB (five pieces per wave).
constexpr int QBM = 160;
template <int NREP, int NI> struct QGeo {
    static constexpr int BM = 32 * NI;                   // 160 | 288
    static constexpr int NA = 2 * NI;                    // X subtiles of a k half
    static constexpr int BN = 64 * NREP;
    static constexpr int NB = BN / 16;
    static constexpr int NP = NA + NB;                   // pieces of a k half: 30 | 26 (160 rows), 38 | 34 (288 rows)
    static constexpr int PW = (NP + 7) / 8;              // ... per wave: 4 | 5
    static constexpr int KS = NP * 1024;
    static constexpr int R = NI == 5 ? 5 : 4;            // ring slots
    static constexpr int LOOP = R * KS;                  // 153600 | 133120; 155648 | 139264
    static constexpr int SMEM = LOOP + (NREP == 4 ? W_TAIL_GEGLU : W_TAIL) + 1024;         // + the dummy pieces' KiB
    static constexpr bool DB = NI == 5;                  // two fragment sets (else refreshed in place)
};
template <int N> __device__ __forceinline__ void w_vmcnt_pieces(int c) {       // at most c k halves' worth of a wave's N pieces still in flight (c wave-uniform)
    if (c <= 0) W_VMCNT(0);
    else if (c == 1) { if constexpr (N == 4) W_VMCNT(4); else W_VMCNT(5); }
    else if (c == 2) { if constexpr (N == 4) W_VMCNT(8); else W_VMCNT(10); }
    else if (c == 3) { if constexpr (N == 4) W_VMCNT(12); else W_VMCNT(15); }
    else { if constexpr (N == 4) W_VMCNT(16); else W_VMCNT(20); }
}

// The same with EXTRA younger operations of the wave known to be in flight (the deferred residual pieces of wq_kernel).
template <int N, int EXTRA> __device__ __forceinline__ void w_vmcnt_pieces_plus(int c) {
    static_assert(N == 4 && EXTRA == 15, "literal counts below");
    if (c <= 0) W_VMCNT(15);
    else if (c == 1) W_VMCNT(19);
    else if (c == 2) W_VMCNT(23);
    else W_VMCNT(27);
}
// One residual piece requested WITHOUT the compiler's knowledge (inline assembly: its own bookkeeping would put a vmcnt(0) — the whole
// DMA ring — in front of the first use); the caller guarantees arrival by a counted wait and pins the use behind it (w_pin).
__device__ __forceinline__ u32x4 w_load16_async(const void* base, unsigned off) {          // base: wave-uniform (SGPR pair), off: the lane's byte offset
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
__device__ __forceinline__ u32x2 w_load8_async(const void* base, unsigned off) {
    u32x2 r;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
__device__ __forceinline__ void w_pin(u32x4& r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ void w_pin(u32x2& r) { asm volatile("" : "+v"(r)); }

template <int MODE, int NREP, bool GEGLU, int RS, int NI>          // RS: 0 = no residual, 1 = it seeds the accumulators, 2 = deferred (below)
__global__ __launch_bounds__(512, 2) void wq_kernel(const MudgGemmDesc p, const int vflags, const float* __restrict__ phi) {
    using G = QGeo<NREP, NI>;
    constexpr int WBN = G::BN, KS = G::KS, R = G::R, NPAIR = NREP / 2, BMq = G::BM, NA = G::NA, PW = G::PW;
    static_assert(!GEGLU || (MODE == 0 && NREP == 4), "GEGLU: plain GEMM on the 256-wide tile");
    static_assert(PW == 4 || PW == 5, "pieces per wave");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int wr = wave >> 2, wc = wave & 3;             // waves wc and wc + 4 share a SIMD: the two M halves
    float* tail = reinterpret_cast<float*>(smem + G::LOOP);
    if (GEGLU && phi) {                                  // (value, step) pairs: gelu_lut2; visible after the K loop's barriers
        for (int t = tid; t <= PHI_N; t += 512) {
            const float a = phi[t], b = phi[t < PHI_N ? t + 1 : t];
            *reinterpret_cast<f32x2*>(&tail[2 * t]) = f32x2{a, b - a};
        }
    }
    // XCD-aware tile numbering (as wgemm_kernel)
    const int ntn = p.N / WBN, ntm = (p.M + BMq - 1) / BMq;
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
    const int m0 = tm * BMq, n0 = tn * WBN;
    constexpr int ntaps = MODE == 0 ? 1 : (MODE == 1 ? 9 : 3);

    const int pos = lane * 16;
    const int sbyte = pos ^ (((pos >> 9) & 1) << 5);
    const int srow = sbyte >> 6, schunk = (sbyte >> 4) & 3;
    const h16* X = reinterpret_cast<const h16*>(p.X);
    const h16* X2 = p.X2 ? reinterpret_cast<const h16*>(p.X2) : nullptr;
    const h16* W = reinterpret_cast<const h16*>(p.W);
    const int64_t shift = MODE == 1 ? -(int64_t)(p.Win + 1) : (MODE == 2 ? -(int64_t)p.HW : 0);
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(X + ((int64_t)m0 + shift) * p.ldx);
    const __amdgpu_buffer_rsrc_t rX2 = X2 ? make_rsrc(X2 + ((int64_t)m0 + shift) * p.ldx2) : rX;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(W + (int64_t)n0 * p.ldw);
    const int ldx2e = X2 ? p.ldx2 : p.ldx;
    const unsigned va1 = (unsigned)(srow * p.ldx) * 2u + (unsigned)schunk * 16u;
    const unsigned va2 = (unsigned)(srow * ldx2e) * 2u + (unsigned)schunk * 16u;
    const unsigned vw_pair = (unsigned)((8 * (srow >> 2) + (srow & 3)) * p.ldw) * 2u + (unsigned)schunk * 16u;     // (permuted W rows: wgemm_kernel)
    const unsigned vw_single = (unsigned)(srow * p.ldw) * 2u + (unsigned)schunk * 16u;
    // This wave's pieces of a k half: piece indices wave + 8 q (q = 0 .. PW - 1); a piece below NA is an X subtile, below NP the W subtile
    // piece - NA, and beyond NP (the last pieces of some waves) a DUMMY: the same instruction with every lane out of range, zero-filling a
    // spare KiB behind the tail — every wave issues exactly PW operations per k half, so the counted waits are the same for all.
    // What piece q is — X for every wave (XQ leading pieces), W for every wave, or wave-dependent — is loop-invariant.
    constexpr int XQ = NA / 8;                            // pieces 0 .. XQ - 1 are X pieces of every wave (1 | 2)
    constexpr int XMAX = (NA + 7) / 8;                    // a wave has at most XMAX X pieces (2 | 3)
    const bool xmix = wave + 8 * XQ < NA;                 // piece XQ: X for the first NA - 8 XQ waves, W for the others
    int wso[5];                                           // per W piece: the scalar offset of its first W row  (fixed bounds: with [PW] — a
                                                          // constexpr local of the template — hipcc 7.2 silently drops the kernel's HOST stub)
    unsigned wv[5];                                       // ... and the lanes' offsets (permuted rows for paired fragments; OOB: dummy)
#pragma unroll
    for (int q = XQ; q < PW; ++q) {
        const int pc = wave + 8 * q;
        const bool live = pc >= NA && pc < G::NP;
        const int st = live ? pc - NA : 0, wcol = st / NREP, j = st - wcol * NREP;
        const bool single = j >= 2 * NPAIR;
        const int row0 = single ? wave_single_col<NREP>(wcol) : wave_pair_col<NREP>(wcol, j >> 1) + 4 * (j & 1);       // first channel of the piece
        wso[q] = row0 * p.ldw * 2;
        wv[q] = live ? (single ? vw_single : vw_pair) : OOB;
    }
    const bool dummy_last = !(wave + 8 * (PW - 1) < G::NP);
    unsigned amask[3];                                    // tap validity of the lane's source row in this wave's X subtiles
#pragma unroll
    for (int q = 0; q < XMAX; ++q) {
        const int st = wave + 8 * q;
        const int m = m0 + st * 16 + srow;
        unsigned mask = 0;
        if (st < NA && m < p.M) {
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
    const int NH = 2 * (p.K / BK);                        // k halves (>= 2, even)
    auto advance = [&](KPos& k) {
        k.kt += 1;
        if (MODE == 0) { k.c += BK; return; }
        const int t1 = k.tap + 1, c1 = k.c + BK;
        const bool slab = p.korder != 0;
        const bool wrap = slab ? (t1 == ntaps) : (c1 == p.Cin);
        k.tap = slab ? (wrap ? 0 : t1) : (wrap ? t1 : k.tap);
        k.c = slab ? (wrap ? c1 : k.c) : (wrap ? 0 : c1);
    };
    auto stage = [&](const KPos& k, int ks, int slot) {
        char* base = smem + slot * KS;
        const bool s2 = k.c >= p.csplit;
        const int cc = s2 ? k.c - p.csplit : k.c;
        const int ld = s2 ? ldx2e : p.ldx;
        int soff = (cc + ks * 32) * 2;
        if (MODE == 1) { const int dy = k.tap / 3, dx = k.tap - 3 * dy; soff += (dy * p.Win + dx) * ld * 2; }
        if (MODE == 2) soff += k.tap * p.HW * ld * 2;
        const int soffw = (k.kt * BK + ks * 32) * 2;
        // No branch in here (a taken scalar branch costs the wave its instruction buffer, and both waves of a SIMD run this right after the
        // same barrier with the matrix pipe idle): the second source and "piece XQ is an X piece" are selects of descriptor and offsets.
        const __amdgpu_buffer_rsrc_t rA = s2 ? rX2 : rX;
        const unsigned vxa = s2 ? va2 : va1;
        static_assert(PW == XQ + 3, "pieces of a wave: XQ X pieces, one X-or-W piece, one W piece, one W-or-dummy piece");
        const unsigned v0 = ((amask[0] >> k.tap) & 1u) ? vxa : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(base + wave * 1024), 16, (int)v0, soff + wave * 16 * ld * 2, 0, 0);
        if constexpr (XQ == 2) {
            const unsigned v1 = ((amask[1] >> k.tap) & 1u) ? vxa : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(base + (wave + 8) * 1024), 16, (int)v1, soff + (wave + 8) * 16 * ld * 2, 0, 0);
        }
        {
            const int pc = wave + 8 * XQ;
            const unsigned vm = ((amask[XQ] >> k.tap) & 1u) ? vxa : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xmix ? rA : rW, (lptr_t)(base + pc * 1024), 16, (int)(xmix ? vm : wv[XQ]),
                                                     xmix ? soff + pc * 16 * ld * 2 : soffw + wso[XQ], 0, 0);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(base + (wave + 8 * (XQ + 1)) * 1024), 16, (int)wv[XQ + 1], soffw + wso[XQ + 1], 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(dummy_last ? smem + G::SMEM - 1024 : base + (wave + 8 * (XQ + 2)) * 1024), 16, (int)wv[XQ + 2],
                                                 soffw + wso[XQ + 2], 0, 0);
    };

    W_STAMP(0);
    KPos kS{0, 0, 0};
    int hs = 0, sslot = 0;                                // the next k half to stage and its slot
    auto stage_next = [&]() {
        stage(kS, hs & 1, sslot);
        ++hs;
        sslot = sslot == R - 1 ? 0 : sslot + 1;
        if (!(hs & 1)) advance(kS);
    };
    // RS (a residual seeds the accumulators) is a template parameter: as a run-time branch its merge point carried a vmcnt(0) — every tile
    // waited for all the prefetched k halves before its first MFMA, residual or not.
    // DEFERRED RESIDUAL (RS = 2: the 160-row tile, 16-bit residual kinds).  The first k halves are requested FIRST, then the 15 residual pieces
    // of the lane — raw, by loads the compiler does not track (w_load16_async), into 50 registers this loop leaves free — and the MFMAs
    // start on zeroed accumulators; the waits for k halves 1 .. 3 (older than the pieces) allow 15 more operations in flight, the wait for
    // k half 4 (younger) retires them, and the EPILOGUE adds them: ((x w + bias) + r), the order of the 128 x 128 one-tile kernels — with a
    // 16-bit residual the tile's results are BIT-IDENTICAL to theirs (tools/exp_w160.py parity).  Built to take the residual off the
    // path to the first MFMA (vector-memory results return in issue order, and profiles/r6/stamps.txt has a K = 320 workgroup spend 43 % of
    // its life in front of it); measured: x 0.97 ... 1.03 of the seeded form on every shape (w160_shapes.txt) — the residual's bytes cost
    // their bandwidth wherever they are requested.  Kept for the bits.  tests/test_isa_rules.py checks on the generated code that nothing
    // touches the 50 pending registers before the counted wait in front of the epilogue.
    constexpr bool DEFER = RS == 2;
    static_assert(!DEFER || (G::DB && NREP == 5 && PLANES == 1), "deferred residual: the 160 x 320 tile of the 16-bit builds");
    constexpr int NSEED = DEFER ? NI : 1;
    u32x4 sraw[NSEED][2];
    u32x2 srs[NSEED];
    f32x4 acc[NI][NREP];
    if constexpr (RS == 1) w_seed<NREP, NI>(p, acc, m0, n0, wr, wc, lane);
    else {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // R - 1 k halves are staged ahead: in iteration h the fragments of k half h are in registers, those of h + 1 are read from their slot,
    // and k half h + R - 1 goes to the slot of k half h - 1.  (Filling the whole ring — k half h + R into k half h's own slot, which needs
    // an lgkmcnt(0) in front of the barrier — was measured: - 3 ... - 4 % on the convs at either height; profiles/r6/w288q_shapes_d4.txt.)
#pragma unroll
    for (int i = 0; i < R - 1; ++i)
        if (i < NH) stage_next();
    constexpr int SEED_OPS = 3 * NI;                      // per lane: two 16-byte pieces and one 8-byte piece per row fragment
    if constexpr (DEFER) {
        {                     // every lane of every wave issues all of them (rows beyond M read the last row: never stored): the counts below rely on it
            // addresses as wave-uniform bases (the tile's first row and the wave's columns) + one 32-bit offset per lane and row: 64-bit
            // lane addresses for 15 requests at once spilled (and a spilled destination of an untracked load is garbage)
            const int px = lane & 15, q4 = lane >> 4;
            const char* Rt = reinterpret_cast<const char*>(p.R) + ((int64_t)m0 * p.ldr + n0) * 2;
            const char* b0 = Rt + wave_pair_col<NREP>(wc, 0) * 2;
            const char* b1 = Rt + wave_pair_col<NREP>(wc, 1) * 2;
            const char* bs = Rt + wave_single_col<NREP>(wc) * 2;
            const int rlast = (int)(p.M - 1 - m0);         // rows beyond M read the last row (never stored)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int r = wr * (16 * NI) + 16 * i + px;
                const unsigned off = (unsigned)((r < rlast ? r : rlast) * p.ldr) * 2u;
                sraw[i][0] = w_load16_async(b0, off + 16u * q4);
                sraw[i][1] = w_load16_async(b1, off + 16u * q4);
                srs[i] = w_load8_async(bs, off + 8u * q4);
            }
        }
    }
    const int fbyte0 = (lane & 15) * 64 + (lane >> 4) * 16;
    const int fbyte = fbyte0 ^ (((fbyte0 >> 9) & 1) << 5);
    const char* a_base = smem + (wr * NI) * 1024 + fbyte;
    const char* b_base = smem + (NA + wc * NREP) * 1024 + fbyte;
    auto frag_a = [&](int slot, int i) { return *reinterpret_cast<const h16x8*>(a_base + slot * KS + i * 1024); };
    auto frag_b = [&](int slot, int j) { return *reinterpret_cast<const h16x8*>(b_base + slot * KS + j * 1024); };
    constexpr int NSET = G::DB ? 2 : 1;
    h16x8 fb[NSET][NREP], fa[NSET][NI];
    if constexpr (DEFER) w_vmcnt_pieces_plus<PW, SEED_OPS>((NH < R - 1 ? NH : R - 1) - 1);
    else w_vmcnt_pieces<PW>((NH < R - 1 ? NH : R - 1) - 1);     // k half 0 has landed
    W_BARRIER();
    W_STAMP(1);
#pragma unroll
    for (int j = 0; j < NREP; ++j) fb[0][j] = frag_b(0, j);
#pragma unroll
    for (int i = 0; i < NI; ++i) fa[0][i] = frag_a(0, i);
    int slot = 0;                                         // slot of k half h
    // The MFMAs of the k half whose fragments are in set `cur`, with the NEXT k half's fragment reads (slot nslot) between them: a wave
    // issues in order, and all the reads in front of the MFMAs — while the other seven waves' reads queue at the same LDS — kept the matrix
    // pipe idle until the last was issued (first version of the 160-row loop: 2650 cycles per k half where the MFMAs need 1600).  After the
    // last k half the reads fetch a stale slot: unused.
    auto multiply = [&](auto cur_tag, int nslot) {
        constexpr int cur = decltype(cur_tag)::value;
        if constexpr (G::DB) {                            // into the other set, one read per two MFMAs
            constexpr int nxt = cur ^ 1;
            H_LGKM0();                                    // this k half's fragments (requested under the previous one's MFMAs)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NREP; ++j) fb[nxt][j] = frag_b(nslot, j);
#pragma unroll
            for (int i = 0; i < NI; ++i) fa[nxt][i] = frag_a(nslot, i);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) acc[i][j] = mfma16(fb[cur][j], fa[cur][i], acc[i][j]);
#pragma unroll
            for (int g = 0; g < NREP + NI; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);          // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // 1 DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NI * NREP - 2 * (NREP + NI), 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {                                          // in place: X fragment i after row i, W fragment j after its MFMA of the last row
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < NREP; ++j) {
                    acc[i][j] = mfma16(fb[0][j], fa[0][i], acc[i][j]);
                    if (i == NI - 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        fb[0][j] = frag_b(nslot, j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                fa[0][i] = frag_a(nslot, i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, G::DB ? 1 : 0>;
    // Steady state — while a k half is staged in every iteration (h + R - 1 <= NH - 1) and R - 3 younger ones than h + 1 are in flight: no
    // condition inside.  `ks`: which half of its K-tile the staged k half h + R - 1 is.
    auto steady = [&](int ks, auto cur_tag) {
        const int nslot = slot == R - 1 ? 0 : slot + 1;
        w_vmcnt_pieces<PW>(R - 3);
        W_BARRIER();
        stage(kS, ks, sslot);
        sslot = sslot == R - 1 ? 0 : sslot + 1;
        if (ks) advance(kS);
        multiply(cur_tag, nslot);
        slot = nslot;
    };
    auto half = [&](int h, auto cur_tag, auto plus_tag) {
        const int nslot = slot == R - 1 ? 0 : slot + 1;
        if (h + 1 < NH) {
            const int last = NH - 1 < h + R - 2 ? NH - 1 : h + R - 2;          // the youngest k half issued so far
            if constexpr (decltype(plus_tag)::value != 0) w_vmcnt_pieces_plus<PW, SEED_OPS>(last - (h + 1));
            else w_vmcnt_pieces<PW>(last - (h + 1));
        }
        W_BARRIER();
        if (hs < NH) stage_next();
        multiply(cur_tag, nslot);
        slot = nslot;
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, DEFER ? 1 : 0>;
    int h = 0;
    if constexpr (DEFER) {
        // k halves 1 .. 3 were requested before the residual pieces: the waits of iterations 0 .. 2 leave the pieces in flight; k half 4
        // after them: its wait (iteration 3) retires them (a shorter K: the vmcnt(0) in front of the epilogue does).
        half(0, C0{}, P1{});
        half(1, C1{}, P1{});
        h = 2;
        if (NH > 2) {
            half(2, C0{}, P1{});
            half(3, C1{}, P0{});
            h = 4;
        }
    }
#pragma unroll 1
    for (; h + R + 1 <= NH; h += 2) {                    // both k halves of the pair are steady: h + 1 <= NH - R
        steady((R - 1) & 1, C0{});
        steady(R & 1, C1{});
    }
    hs = h + R - 1 < NH ? h + R - 1 : NH;                 // (what the steady iterations staged; sslot followed them)
#pragma unroll 1
    for (; h < NH; h += 2) {
        half(h, C0{}, P0{});
        half(h + 1, C1{}, P0{});
    }
    W_STAMP(2);
    if constexpr (DEFER) {
        W_VMCNT(0);
#pragma unroll
        for (int i = 0; i < NI; ++i) { w_pin(sraw[i][0]); w_pin(sraw[i][1]); w_pin(srs[i]); }
        w_epilogue<NREP, GEGLU, 2, true, NI, true>(p, acc, m0, n0, tm, wr, wc, lane, tid, tail, phi, sraw, srs);
    } else w_epilogue<NREP, GEGLU, 2, true, NI>(p, acc, m0, n0, tm, wr, wc, lane, tid, tail, phi);
    W_STAMP(3);
}
#endif

// Variant switch GEMM_W288 (debug-variants build; read at every call so that one process can compare kernels): 0 = never, 1 = the rule
// below, 2 = every eligible problem.
int variant() { return mudg_variant("GEMM_W288", 1); }

}  // namespace

#if MUDG_PLANES == 1
static int persistent_grid(const MudgGemmDesc& d);
// The half-height GEGLU kernel (hgeglu_kernel).  Variant switch GEMM_H144: 0 = never, 1 = the rule, 2 = every GEGLU problem the 288 x 256
// tile is eligible for (the callers have checked wgemm_eligible: mode 0, N % 256 == 0, K % 64 == 0, 16-byte Y pieces, no residual / group
// bias / partials).  Same bits as the kernels it replaces, so the rule may look at M.
static bool half_height_ok(const MudgGemmDesc& d) {
    const int hv = mudg_variant("GEMM_H144", 1);
    if (!hv || !d.geglu) return false;
    if (hv == 2) return true;
    // Measured (MI355X; profiles/r6/h144*.txt, bench_h144_off / _rule.json): per launch x 0.90 ... 1.08 of the persistent 288 x 256 form in
    // isolated timings (+ 3 ... 7 % at K = 320 on the level-0 rows and where the eight-wave tile has less than a round of tiles, - 5 ...
    // - 10 % from K = 512: 92 instead of 136 FLOP per staged byte), and INSIDE the step — the same library, the same box, the rule "where
    // it won in isolation" against never — 122.39 against 121.53 ms: the rocprofv3 trace of that step has its level-0 launches at 710 us
    // where the persistent form's were 681.  The rule is therefore: never.  The kernel stays, tested (tests/test_gemm_variants_gpu.py
    // runs the parity suites with it forced, test_half_height_geglu_kernel_is_bit_identical...), as the measured answer to "two
    // workgroups per CU" (DESIGN §3.2).
    return false;
}
#endif
// What the kernel can run at all.
static bool wgemm_eligible(const MudgGemmDesc& d, int vflags, int bm = WBM) {
    if (d.batch != 1 || d.act || d.Y8 || d.subpixel || (d.mode == 1 && d.upsample)) return false;
    if (d.geglu ? (d.mode != 0 || d.N % 256 != 0 || d.R || d.gbias || d.stats) : d.N % 320 != 0) return false;
    if (!(vflags & VF_Y) || (d.R && !(vflags & VF_R))) return false;
    const int cin = d.mode == 0 ? d.K : d.Cin;
    if ((d.K & 63) || (cin & 63) || (d.csplit & 63)) return false;
    if (d.mode == 1 && (d.stride != 1 || d.pad != 1 || d.Hin != d.Hout || d.Win != d.Wout || d.K != 9 * d.Cin)) return false;
    if (d.mode == 2 && (d.korder || d.K != 3 * d.Cin)) return false;       // (korder 1 means tiles of 8 pixels x 16 frames to the callers: gemm.hip)
    if (d.gbias && (d.rows_per_group % bm != 0)) return false;             // one group per tile: the group bias rides in the column constants
    if (d.R && d.alpha != 1.f) return false;                               // the residual seeds the accumulators (w_seed)
    if ((d.ldy & 7) || (d.R && (d.ldr & 7))) return false;                 // 8-byte pieces of the unpaired fragment
    // 32-bit reach of the descriptor offsets
    const int64_t ld = d.X2 && d.ldx2 > d.ldx ? d.ldx2 : d.ldx;
    int64_t rows = bm + 16 + (PLANES > 1);                                 // (the second piece of a row: ld / 2 elements further)
    if (d.mode == 1) rows += 2 * (int64_t)d.Win + 2;
    if (d.mode == 2) rows += 2 * (int64_t)d.HW;
    const int64_t lim = (int64_t)1 << 31;
    return rows * ld * 2 + (int64_t)cin * 2 + 256 < lim && (int64_t)(320 + 16 + (PLANES > 1)) * d.ldw * 2 + (int64_t)d.K * 2 + 256 < lim;
}

// Where it is used.  The rule never looks at M (see the header): `S`, the rows of one frame (mode 0: the caller's hint in d.HW), must be
// whole tiles — then every frame batch of the benchmarked resolution fills whole rounds of the 256 CUs.
static bool wgemm288_ok(const MudgGemmDesc& d, int vflags) {
    const int mode = variant();
    if (!mode || !wgemm_eligible(d, vflags)) return false;
    if (mode == 2) return true;
#if MUDG_PLANES == 1
    if (d.geglu && half_height_ok(d)) return true;         // (same bits as every other GEGLU kernel: no frame geometry needed)
#endif
    const int S = d.mode == 1 ? d.Hout * d.Wout : d.HW;
    if (S <= 0 || S % WBM != 0) return false;
    // Measured per shape against the 128 x 128 kernels (tools/exp_w288.py, profiles/r5/w288_shapes.txt; MI355X, frames of whole tiles):
    // 3x3 convs + 20 ... + 40 %, temporal convs + 19 ... + 28 %; plain GEMMs + 16 ... + 37 % from K = 1280, + 1 ... + 26 % at K = 320 / 640
    // (N <= K: every projection of the UNet; - 1 ... - 3 % for N = 2 ... 3 K with a residual, which the UNet does not have); GEGLU below.
    // bf16x3 (same tool with MUDG_OPERAND=bf16x3, profiles/r5/w288_x3_shapes.txt; the 128 x 128 side is the fused-piece kernel, one-tile or
    // persistent as gemm.hip selects): 3x3 convs + 27 ... + 44 %, temporal convs + 22 ... + 27 %, GEGLU + 11 ... + 13 %, plain GEMMs + 11 ...
    // + 42 % down to K = 320: three times the MFMAs per staged byte and per epilogue — every problem whose frames are whole tiles.
    if (d.mode != 0 || PLANES == 2) return true;
#if MUDG_PLANES == 1
    // GEGLU (bit-identical to the persistent 128 x 128 kernel it replaces, so M may decide): + 7 ... + 16 % in the persistent form (more
    // tiles than CUs) at every K; the one-tile form + 11 % at K = 1280, - 1 % at K = 640, - 5 ... - 12 % at K = 320.  Since round 6 the
    // two-workgroup half-height kernel (hgeglu_kernel) runs GEGLU wherever its rule says so (half_height_ok).
    if (d.geglu && d.K < 640) return persistent_grid(d) > 0;
#endif
    return true;
}

#if MUDG_PLANES == 1
// The 160-row tile (w160_kernel): frames of whole 160-row tiles that are not whole 288-row tiles.  Variant switch GEMM_W160: 0 = never,
// 1 = the rule, 2 = every eligible problem (as GEMM_W288 = 2; the 288-row tile's own rule is asked first).
static bool w160_ok(const MudgGemmDesc& d, int vflags) {
    const int mode = mudg_variant("GEMM_W160", 1);
    if (!mode || !wgemm_eligible(d, vflags, QBM)) return false;
    if (mode == 2) return true;
    const int S = d.mode == 1 ? d.Hout * d.Wout : d.HW;
    if (S <= 0 || S % QBM != 0) return false;
    // Measured per shape against the 128 x 128 kernels, twice.  In isolation (tools/exp_w160.py, profiles/r6/w160_shapes.txt; MI355X, MDM512's
    // frame batches, every operand hot in the 256-MiB Infinity Cache after the first repeat): 3x3 convs + 28 ... + 45 %, temporal convs + 14 ...
    // + 27 % at 2560- and 640-pixel frames; plain GEMMs + 13 ... + 35 % from K = 640 with N <= K, + 0 ... + 9 % for N = 2 ... 3 K, at K = 320
    // + 7 ... + 14 % without and - 2 ... - 3 % with a residual; GEGLU - 9 ... - 22 % against the persistent 128 x 128 kernel.  And INSIDE
    // the step (tools/shape_profile.py 512 with GEMM_W160 = 0 / 1 / 2, profiles/r6/shapes_m512_w160_*.md), where a kernel's operands come
    // from HBM or from its producer: convs + 12 ... + 27 %, temporal convs + 5 ... + 14 %, plain GEMMs + 11 ... + 16 % at K >= 1280 — and
    // - 3 ... - 13 % at K = 320 / 640, where the one workgroup of a CU waits for its first k half alone while the 128 x 128 kernels have
    // four workgroups per CU to cover for each other (the isolated timing hides it: its operands never leave the cache).  The rule follows
    // the step.  One workgroup per CU wants a frame batch to bring enough tiles: the 160-pixel level (32 frames = 32 tile rows x 4 ... 8
    // columns: half the CUs) stays on the 128 x 128 kernels (- 20 ... - 45 % there).  Never M: S, K, the mode.
    if (S < 640 || d.geglu) return false;
    if (d.mode != 0) return true;
    return d.K >= 1280;
}
#endif
// Height of the tile that will run the problem: 288, 160 or 0 (none of the kernels of this file).
int mudg_wgemm_rows(const MudgGemmDesc& d, int vflags) {
    if (wgemm288_ok(d, vflags)) return WBM;
#if MUDG_PLANES == 1
    if (w160_ok(d, vflags)) return QBM;
#endif
    return 0;
}
bool mudg_wgemm_ok(const MudgGemmDesc& d, int vflags) { return mudg_wgemm_rows(d, vflags) != 0; }

template <int MODE, int NREP, bool GEGLU>
static int wgemm_launch_one(const MudgGemmDesc& d, int vflags, hipStream_t s, int slot) {
    static bool attr_done[MAX_DEVICES][4] = {};
    const int dev = mudg_current_device();
    if (dev < 0) MUDG_FAIL(MUDG_ELAUNCH, "gemm: no current device");
    using G = WGeo<NREP>;
    if (!attr_done[dev][slot]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgemm_kernel<MODE, NREP, GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev][slot] = true;
    }
    const int tiles = ((d.M + WBM - 1) / WBM) * (d.N / G::BN);
    const float* phi = GEGLU ? mudg_phi_table(PLANES == 2) : nullptr;
    hipLaunchKernelGGL((wgemm_kernel<MODE, NREP, GEGLU>), dim3(tiles), dim3(512), G::SMEM, s, d, vflags, phi);
    return mudg_check_launch("mudg_gemm");
}

#if MUDG_PLANES == 1
// The persistent form (wgemm_pkernel): whole tiles, at least two K-tiles, more tiles than CUs.  Same bits as the one-tile form, so M may
// decide.  Measured (same box, tools/exp_w288.py with MUDG_GEMM_W288P = 0 / 2, profiles/r5/w288_persistent.txt): GEGLU + 2 ... + 9 % (its
// epilogue is the longest and fetches nothing); plain GEMMs - 7 ... + 6 % with no pattern worth a rule — a residual's fetches queue
// behind the next tile's staged pieces, and what persistence saves per tile (launch, first-fetch latency) is small beside what bounds
// the short-K problems (the epilogue's own traffic).  Variant switch GEMM_W288P: 0 = never, 1 = GEGLU only (the rule), 2 = every
// problem the kernel can run.
template <int NREP, bool GEGLU>
static int wgemm_launch_persistent(const MudgGemmDesc& d, int vflags, hipStream_t s, int slot, int grid) {
    static bool attr_done[MAX_DEVICES][2] = {};
    const int dev = mudg_current_device();
    using G = WGeo<NREP>;
    if (!attr_done[dev][slot]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgemm_pkernel<NREP, GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev][slot] = true;
    }
    const int tiles = (d.M / WBM) * (d.N / G::BN);
    const float* phi = GEGLU ? mudg_phi_table(PLANES == 2) : nullptr;
    hipLaunchKernelGGL((wgemm_pkernel<NREP, GEGLU>), dim3(grid), dim3(512), G::SMEM, s, d, vflags, phi, tiles);
    return mudg_check_launch("mudg_gemm");
}
static int persistent_grid(const MudgGemmDesc& d) {
    static int cus[MAX_DEVICES] = {};
    const int pv = mudg_variant("GEMM_W288P", 1);
    if (!pv || (pv == 1 && !d.geglu) || d.mode != 0 || d.R || d.M % WBM != 0 || d.K < 2 * BK) return 0;
    {   // the whole problem behind one descriptor per operand: rows ride in 32-bit scalar offsets
        const int64_t ld = d.X2 && d.ldx2 > d.ldx ? d.ldx2 : d.ldx, lim = (int64_t)1 << 31;
        if (((int64_t)d.M + 16) * ld * 2 + (int64_t)d.K * 2 + 256 >= lim || ((int64_t)d.N + 16) * d.ldw * 2 + (int64_t)d.K * 2 + 256 >= lim) return 0;
    }
    const int dev = mudg_current_device();
    if (dev < 0) return 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 8;
        cus[dev] = n & ~7;                                   // whole XCDs' worth of workgroups
    }
    const int tiles = (d.M / WBM) * (d.N / (d.geglu ? 256 : 320));
    return tiles > cus[dev] ? cus[dev] : 0;
}
#endif

#if MUDG_PLANES == 1
template <bool PF>
static int hgeglu_launch_one(const MudgGemmDesc& d, int vflags, hipStream_t s) {
    static bool attr_done[MAX_DEVICES] = {};
    const int dev = mudg_current_device();
    if (dev < 0) MUDG_FAIL(MUDG_ELAUNCH, "gemm: no current device");
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hgeglu_kernel<PF>), hipFuncAttributeMaxDynamicSharedMemorySize, H_SMEM);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev] = true;
    }
    const int tiles = ((d.M + 16 * H_NA - 1) / (16 * H_NA)) * (d.N / 256);
    static int cus[MAX_DEVICES] = {};
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cus[dev] = n;
    }
    const int delay = mudg_variant("GEMM_H144DELAY", 0) | (mudg_variant("GEMM_H144ABL", 0) << 16) | (mudg_variant("GEMM_H144PRIO", 1) ? 0 : 1 << 20);
    hipLaunchKernelGGL(hgeglu_kernel<PF>, dim3(tiles), dim3(256), H_SMEM, s, d, vflags, mudg_phi_table(false), 2 * cus[dev], delay);
    return mudg_check_launch("mudg_gemm");
}
// Variant switch GEMM_H144PF (measurements): 0 = the plain loop (every k half starts with its own fragment reads), 1 = the prefetching loop.
static int hgeglu_launch(const MudgGemmDesc& d, int vflags, hipStream_t s) {
    return mudg_variant("GEMM_H144PF", 1) ? hgeglu_launch_one<true>(d, vflags, s) : hgeglu_launch_one<false>(d, vflags, s);
}
#endif

#if defined(MUDG_DEBUG_VARIANTS) && MUDG_PLANES == 1
// tools/exp_stamps.py (debug-variants build; not part of the ABI of include/mudg_hip.h)
extern "C" int mudg_debug_set_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

#if MUDG_PLANES == 1
template <int MODE, int NREP, bool GEGLU, int RS, int NI>
static int wq_launch_one(const MudgGemmDesc& d, int vflags, hipStream_t s, int slot) {
    static bool attr_done[MAX_DEVICES][20] = {};
    const int dev = mudg_current_device();
    if (dev < 0) MUDG_FAIL(MUDG_ELAUNCH, "gemm: no current device");
    using G = QGeo<NREP, NI>;
    if (!attr_done[dev][slot]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wq_kernel<MODE, NREP, GEGLU, RS, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev][slot] = true;
    }
    const int tiles = ((d.M + G::BM - 1) / G::BM) * (d.N / G::BN);
    const float* phi = GEGLU ? mudg_phi_table(false) : nullptr;
    hipLaunchKernelGGL((wq_kernel<MODE, NREP, GEGLU, RS, NI>), dim3(tiles), dim3(512), G::SMEM, s, d, vflags, phi);
    return mudg_check_launch("mudg_gemm");
}
#endif

int mudg_wgemm_launch(const MudgGemmDesc& d, int vflags, hipStream_t s) {
#if MUDG_PLANES == 1
    const int rows = mudg_wgemm_rows(d, vflags);
    if (rows == QBM) {
        if (d.geglu) return wq_launch_one<0, 4, true, 0, 5>(d, vflags, s, 6);
        // a residual of 16-bit storage (the fp16 stream, an operand matrix) is deferred to the epilogue, an fp32 one seeds the accumulators.
        // Variant switch GEMM_W160DEFER = 0: every residual seeds.
        const int rs = !d.R ? 0 : ((d.res_fp32 != KIND_F32 && mudg_variant("GEMM_W160DEFER", 1)) ? 2 : 1);
        if (d.mode == 0) return rs == 0 ? wq_launch_one<0, 5, false, 0, 5>(d, vflags, s, 0) : (rs == 1 ? wq_launch_one<0, 5, false, 1, 5>(d, vflags, s, 1) : wq_launch_one<0, 5, false, 2, 5>(d, vflags, s, 14));
        if (d.mode == 1) return rs == 0 ? wq_launch_one<1, 5, false, 0, 5>(d, vflags, s, 2) : (rs == 1 ? wq_launch_one<1, 5, false, 1, 5>(d, vflags, s, 3) : wq_launch_one<1, 5, false, 2, 5>(d, vflags, s, 15));
        return rs == 0 ? wq_launch_one<2, 5, false, 0, 5>(d, vflags, s, 4) : (rs == 1 ? wq_launch_one<2, 5, false, 1, 5>(d, vflags, s, 5) : wq_launch_one<2, 5, false, 2, 5>(d, vflags, s, 16));
    }
#ifdef MUDG_DEBUG_VARIANTS
    // The 288-row tile on the loop of the 160-row one (wq_kernel<..., 9>), variant builds only.  Variant switch GEMM_W288Q: 0 = never (the
    // rule: wgemm_kernel's six-phase loop), 2 = every one-tile problem of the 288-row tile (GEGLU included).  Measured per shape against the
    // six-phase loop (tools/exp_w288.py q, profiles/r6/w288q_shapes.txt): 3x3 convs x 0.96 ... 1.04, temporal convs x 1.00 ... 1.02, plain
    // GEMMs x 0.85 ... 1.07, GEGLU (one-tile against the persistent six-phase form) x 0.90 ... 0.98 — the same bits and no gain: two
    // different schedules of the same 45 MFMAs, 14 fragment reads and 38 DMA pieces per k half end at the same 1300 - 1430 TFLOP/s.
    if (mudg_variant("GEMM_W288P", 1) != 2 && mudg_variant("GEMM_W288Q", 0) == 2) {
        if (d.geglu) return wq_launch_one<0, 4, true, 0, 9>(d, vflags, s, 13);
        if (d.mode == 0) return d.R ? wq_launch_one<0, 5, false, 1, 9>(d, vflags, s, 7) : wq_launch_one<0, 5, false, 0, 9>(d, vflags, s, 8);
        if (d.mode == 1) return d.R ? wq_launch_one<1, 5, false, 1, 9>(d, vflags, s, 9) : wq_launch_one<1, 5, false, 0, 9>(d, vflags, s, 10);
        return d.R ? wq_launch_one<2, 5, false, 1, 9>(d, vflags, s, 11) : wq_launch_one<2, 5, false, 0, 9>(d, vflags, s, 12);
    }
#endif
    if (d.geglu && half_height_ok(d)) return hgeglu_launch(d, vflags, s);
    if (const int grid = persistent_grid(d))
        return d.geglu ? wgemm_launch_persistent<4, true>(d, vflags, s, 1, grid) : wgemm_launch_persistent<5, false>(d, vflags, s, 0, grid);
#endif
    if (d.geglu) return wgemm_launch_one<0, 4, true>(d, vflags, s, 3);
    if (d.mode == 0) return wgemm_launch_one<0, 5, false>(d, vflags, s, 0);
    if (d.mode == 1) return wgemm_launch_one<1, 5, false>(d, vflags, s, 1);
    return wgemm_launch_one<2, 5, false>(d, vflags, s, 2);
}
#else
bool mudg_wgemm_ok(const MudgGemmDesc&, int) { return false; }
int mudg_wgemm_rows(const MudgGemmDesc&, int) { return 0; }
int mudg_wgemm_launch(const MudgGemmDesc&, int, hipStream_t) { MUDG_FAIL(MUDG_EINVAL, "gemm: no 288 x 320 kernel in this build"); }
#endif
