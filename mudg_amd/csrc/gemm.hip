// gemm.hip — the workhorse MFMA contraction kernel of the path (plain GEMM, 3x3 conv and temporal 3-tap conv as
// implicit GEMMs on channels-last activations).  See include/mudg_hip.h (MudgGemmDesc) for semantics.
//
// Tiling (gfx950, wave64): 128x128 output tile per 256-thread workgroup, BK = 64, four waves as 2(M) x 2(N),
// each wave owns 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles (64 fp32 accumulators per lane).
// The MFMA is issued "transposed" (A operand = weight rows, B operand = activation rows) so that each lane ends
// up with 4 consecutive output channels of one pixel — the epilogue then moves 16-byte pieces.
// Staging: HBM/L2 -> LDS directly by LDS-DMA (no VGPR round trip, no ds_write pass: the first version staged through
// registers and was LDS-write-bound at ~600 TFLOP/s).  The DMA writes lane-linear 1-KiB pieces (8 rows x 128 B), so the
// LDS tile is unpadded [128][64] h16 and bank conflicts are removed by an XOR swizzle applied on the SOURCE side (which
// 16-byte chunk of the row a lane fetches) and mirrored on the fragment reads: chunk c of row r lives in slot
// c ^ ((r >> 1) & 7); with that, every 16-lane ds_read_b128 group touches 16 distinct 16-byte slots of the 256-byte bank row.
// Two address paths (template FAST): buffer_load ... lds through block-relative buffer descriptors with loop-invariant lane
// offsets and hardware zero-fill for padding (every K / Cin % 64 == 0 problem without upsampling), or global_load_lds with
// per-lane 64-bit addresses and a zero page (everything else).
// Two occupancy variants (template SB): one K-tile buffer + two-pass epilogue = 4 workgroups per CU that overlap each
// other's fetch / multiply / store phases, or two K-tile buffers (the DMA of tile k+1 flies under the MFMAs of tile k) at 2
// workgroups per CU; use_single_buffer() picks by tile count.
// Epilogue: accumulators (+bias, GEGLU through an LDS Phi table) -> fp32 LDS tile -> coalesced 16-B rows (+group bias,
// +residual, optional GroupNorm partial sums of what was stored) -> HBM.
// Workgroups are numbered so that each XCD gets a contiguous run of tiles, walked in 8-row groups (8 x 8 tile patches).
#include "common.h"
#include <cstdlib>
#include <cmath>

bool mudg_gemm_fast_ok(const MudgGemmDesc& d);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDSLD = 64;                       // h16 elements per LDS row: unpadded, XOR-swizzled (see header)
constexpr int STGLD = 132;                      // fp32 per staging row (128 + 4 pad)
constexpr int TILE = BM * LDSLD;                // elements per operand per buffer
constexpr int SMEM_MAIN = 4 * TILE * 2;         // X[2] + W[2], bytes
constexpr int SMEM_STG = BM * STGLD * 4 + BN * 4;
constexpr int SMEM_BYTES = SMEM_MAIN > SMEM_STG ? SMEM_MAIN : SMEM_STG;
// Single-buffer variant (short K): one K-tile buffer, the epilogue staged in two 64-row passes -> 4 workgroups per CU.
constexpr int SMEM_STG_SB = (BM / 2) * STGLD * 4 + BN * 4;
constexpr int SMEM_BYTES_SB = 2 * TILE * 2 > SMEM_STG_SB ? 2 * TILE * 2 : SMEM_STG_SB;
// bf16x3 build, descriptor loader (FUSED): ONE K-tile stage holds both pieces of both operands — x0, x1, w0, w1, 4 x 16 KiB —
// and every fragment read feeds the three kept products x1 w0 + x0 w1 + x0 w0: each piece is fetched once per K-tile (4 tile
// fetches instead of the 6 of three whole passes over K) and 12 MFMAs follow 8 fragment reads instead of 4 following 4.
// Single K-buffer + two-pass epilogue = 64 KiB -> 2 workgroups per CU.
constexpr bool fused_planes(bool fast) { return PLANES == 2 && fast; }
constexpr int SMEM_BYTES_FUSED = 2 * PLANES * TILE * 2;
constexpr int smem_main(bool fast, bool sb) { return fused_planes(fast) ? SMEM_BYTES_FUSED : (sb ? SMEM_BYTES_SB : SMEM_BYTES); }

constexpr int VF_Y = 1, VF_R = 2;               // 16-byte access allowed on Y / R

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

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

// FAST (host-checked: K, Cin, csplit multiples of 64, no upsample, block-relative offsets < 2 GiB): operand tiles are
// fetched with buffer_load_dwordx4 ... lds through block-relative buffer descriptors — the per-lane byte offset
// (row * ld + swizzled chunk) is loop-invariant, the K / tap advance is one SGPR offset, and a missing row or tap is
// an out-of-range offset that the hardware zero-fills.  The generic path (any K % 8 == 0, upsample, straddling
// sources) computes 64-bit addresses per lane per K-tile and fetches padding from a zero page.
// SB (short-K plain GEMMs, K <= 640): with 5-10 K-tiles per output tile the fixed cost of a tile (first fetch, epilogue)
// is as long as its K loop, and only other resident workgroups can hide it.  One K-tile buffer instead of two and a
// two-pass epilogue bring the LDS footprint to 34 KiB, so four workgroups share a CU (16 waves, 4 per SIMD) and overlap
// each other's fetch / multiply / store phases; inside a workgroup fetch and multiply then alternate.
// GEGLU's gate: gelu(x) = x Phi(x) with Phi linearly interpolated from a 1025-entry table over [-8, 8] held in LDS
// (|error| <= h^2/8 max|Phi''| = 7.4e-6 at h = 1/64 — below the h16 rounding of the result by two orders): ten VALU
// instructions and one ds_read2 instead of the ~21 issue slots of the erf polynomial + v_exp + v_rcp, which made the
// K = 320 GEGLU tiles VALU-bound (GELU was 21 % of their time).
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

template <int MODE, bool FAST, bool SB>
__global__ __launch_bounds__(256, (SB && !fused_planes(FAST)) ? 4 : 2) void gemm_kernel(const MudgGemmDesc p, const int vflags, const h16* __restrict__ zpage,
                                                                const float* __restrict__ phi) {
    constexpr bool FUSED = fused_planes(FAST);          // all pieces of a K-tile staged at once (see SMEM_BYTES_FUSED)
    constexpr bool ONEBUF = SB || FUSED;                // one K-tile stage, two-pass epilogue
    constexpr int XT = FUSED ? PLANES : 1;              // tiles per operand per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* Xs = reinterpret_cast<h16*>(smem);
    h16* Ws = Xs + (ONEBUF ? 1 : 2) * XT * TILE;
    float* phis = reinterpret_cast<float*>(smem + smem_main(FAST, SB));     // beyond every other LDS use
    if (p.geglu && phi) {            // visible after the K loop's barriers
        const int t4 = threadIdx.x * 4;
        *reinterpret_cast<f32x4*>(&phis[t4]) = *reinterpret_cast<const f32x4*>(&phi[t4]);
        if (threadIdx.x == 0) phis[PHI_N] = phi[PHI_N];
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, hi = lane >> 5;

    // XCD-aware tile numbering: hardware puts workgroup b on XCD b % 8; give every XCD a contiguous tile range.
    const int ntn = (p.N + BN - 1) / BN;
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
        const int ntm = (p.M + BM - 1) / BM;
        const int per = 8 * ntn, g = tile / per, first = g * 8;
        const int gsz = (ntm - first) < 8 ? (ntm - first) : 8;
        const int r = tile - g * per;
        tn = r / gsz;
        tm = first + (r - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int64_t bz = blockIdx.z;
    // Sub-pixel form of "nearest-2x upsample, then 3x3 conv" (MudgGemmDesc.subpixel): batch entry z = 2 py + px computes the
    // output pixels (2 oy + py, 2 ox + px) from the 2x2 low-resolution neighbourhood that their nine taps collapse onto.
    const bool sub = MODE == 1 && p.subpixel;
    const int ntaps = sub ? 4 : 9;
    const int dy0 = sub ? (int)(blockIdx.z >> 1) : 0, dx0 = sub ? (int)(blockIdx.z & 1) : 0;
    const h16* X = reinterpret_cast<const h16*>(p.X) + bz * p.sX;
    const h16* X2 = p.X2 ? reinterpret_cast<const h16*>(p.X2) + bz * p.sX : nullptr;
    const h16* W = reinterpret_cast<const h16*>(p.W) + bz * p.sW;

    // DMA geometry: wave w stages rows [32w, 32w+32) of both operand tiles with four 1-KiB instructions each;
    // in instruction i, lane l lands in row 32w + 8i + (l >> 3), slot l & 7, and therefore fetches the logical
    // chunk (l & 7) ^ ((row >> 1) & 7) of that row.
    const int rsub = lane >> 3, slot = lane & 7;
    int rm[4], ra[4], rb[4], rc[4], ch[4];
    bool rv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = 32 * wave + 8 * i + rsub;
        ch[i] = slot ^ ((rl >> 1) & 7);
        const int m = m0 + rl;
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
    const bool tap_uniform = MODE != 0 && (p.Cin & 63) == 0;     // a 64-wide K tile never straddles two taps

    // FAST path state: descriptors based at the block's first source row, invariant lane offsets, tap validity bits.
    __amdgpu_buffer_rsrc_t rX, rX2, rW;
    unsigned vx[4], vx2[4], vw[4], vmask[4];
    int tap_s = 0, c_s = 0;                                      // tap / channel of the next K-tile to issue
    int seg_s = 0, kt_s = 0;                                     // split operands: (x plane, w plane) pass and K-tile inside it
    const int nk = (p.K + BK - 1) / BK;
    if constexpr (FAST) {
        int64_t pix0 = m0;                                       // source pixel (row of X) of output row m0, before the tap shift
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
        for (int i = 0; i < 4; ++i) {
            const int rl = 32 * wave + 8 * i + rsub;
            int rel = rl;
            unsigned mask = rv[i] ? 1u : 0u;
            if (MODE == 1) {
                rel = (int)((int64_t)(ra[i] + (rb[i] + p.pad) * p.Win + rc[i] + p.pad) - pix0);
                mask = 0;
                if (rv[i]) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int iy = rb[i] + (sub ? (t >> 1) + dy0 : t / 3), ix = rc[i] + (sub ? (t & 1) + dx0 : t % 3);
                        if (t < ntaps && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mask |= 1u << t;
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
            vw[i] = (n0 + rl < p.N) ? (unsigned)rl * (unsigned)p.ldw * 2u + cb : OOB;
        }
    }

    auto issue_fast = [&](int kt, int buf) {
        const bool s2 = c_s >= p.csplit;
        const int cc = s2 ? c_s - p.csplit : c_s;
        const int ld = s2 ? p.ldx2 : p.ldx;
        int soff;
        if (MODE == 0) soff = cc * 2;
        else if (MODE == 1) {
            int dy = tap_s / 3, dx = tap_s - 3 * dy;
            if (sub) { dy = (tap_s >> 1) + dy0; dx = (tap_s & 1) + dx0; }
            soff = ((dy * p.Win + dx) * ld + cc) * 2;
        }
        else soff = (tap_s * p.HW * ld + cc) * 2;
        int soffw = ((PLANES > 1 && !FUSED) ? kt_s : kt) * (BK * 2);
        if constexpr (PLANES > 1 && !FUSED) {        // this pass's operand planes: column offsets of ld / PLANES elements
            soff += seg_xp(seg_s) * (ld / PLANES) * 2;
            soffw += seg_wp(seg_s) * (p.ldw / PLANES) * 2;
        }
#pragma unroll
        for (int pl = 0; pl < XT; ++pl) {            // FUSED: piece pl of both operands into its own tile of the stage
            const int so = soff + pl * (ld / PLANES) * 2, sow = soffw + pl * (p.ldw / PLANES) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned v = s2 ? vx2[i] : vx[i];
                if (MODE != 0) v = ((vmask[i] >> tap_s) & 1u) ? v : OOB;
                lptr_t lx = (lptr_t)(Xs + (buf * XT + pl) * TILE + (32 * wave + 8 * i) * LDSLD);
                if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, lx, 16, (int)v, so, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, lx, 16, (int)v, so, 0, 0);
                // W rows beyond N are never multiplied (their waves are idle, see wave_live): skip the zero-fill pieces
                if (n0 + 32 * wave + 8 * i < p.N || (32 * wave + 8 * i) < 64)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(Ws + (buf * XT + pl) * TILE + (32 * wave + 8 * i) * LDSLD), 16,
                                                             (int)vw[i], sow, 0, 0);
            }
        }
        if (MODE == 0) {
            c_s += BK;
        } else {                                   // select form: the branchy update sent tap_s / c_s to scratch memory
            const int t1 = tap_s + 1, c1 = c_s + BK;
            const bool slab = MODE == 1 && p.korder;
            const bool wrap = slab ? (t1 == ntaps) : (c1 == p.Cin);
            tap_s = slab ? (wrap ? 0 : t1) : (wrap ? t1 : tap_s);
            c_s = slab ? (wrap ? c1 : c_s) : (wrap ? 0 : c1);
        }
        if constexpr (PLANES > 1 && !FUSED) {        // end of a pass over K: next plane pair, K walk restarts (select form)
            const bool last = kt_s + 1 == nk;
            kt_s = last ? 0 : kt_s + 1;
            seg_s = last ? seg_s + 1 : seg_s;
            tap_s = last ? 0 : tap_s;
            c_s = last ? 0 : c_s;
        }
    };

    auto issue_tiles = [&](int kt, int buf) {
        if constexpr (FAST) { issue_fast(kt, buf); return; }
        const int seg = PLANES > 1 ? kt / nk : 0;
        kt -= seg * nk;
        const int xo1 = seg_xp(seg) * (p.ldx / PLANES), xo2 = seg_xp(seg) * (p.ldx2 / PLANES), wo = seg_wp(seg) * (p.ldw / PLANES);
        const int k0 = kt * BK;
        int tap_u = 0, c_u = 0;
        if (MODE != 0 && tap_uniform) {
            if (MODE == 1 && p.korder) { const int slab = kt / 9; tap_u = kt - slab * 9; c_u = slab * 64; }
            else { tap_u = k0 / p.Cin; c_u = k0 - tap_u * p.Cin; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ch[i] * 8;
            const bool kv = k < p.K;
            const h16* src = zpage;
            if (MODE == 0) {
                if (rv[i] && kv)
                    src = (k < p.csplit) ? X + (int64_t)rm[i] * p.ldx + k + xo1 : X2 + (int64_t)rm[i] * p.ldx2 + (k - p.csplit) + xo2;
            } else {
                int tap, c;
                if (tap_uniform) { tap = tap_u; c = c_u + ch[i] * 8; }
                else { tap = k / p.Cin; c = k - tap * p.Cin; }
                const h16* base = X; int cc = c + xo1, ld = p.ldx;
                if (c >= p.csplit) { base = X2; cc = c - p.csplit + xo2; ld = p.ldx2; }
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
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xs + buf * TILE + (32 * wave + 8 * i) * LDSLD), 16, 0, 0);
            const int n = n0 + 32 * wave + 8 * i + rsub;
            const h16* wsrc = (n < p.N && kv) ? W + (int64_t)n * p.ldw + k + wo : zpage;
            __builtin_amdgcn_global_load_lds((gptr_t)wsrc, (lptr_t)(Ws + buf * TILE + (32 * wave + 8 * i) * LDSLD), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nkt = FUSED ? nk : nk * NSEG;      // K-tiles over all (x plane, w plane) passes; FUSED: all pieces per K-tile
    const int sw = (l31 >> 1) & 7;       // read-side swizzle: the row bases are multiples of 32, so only lane bits count
    // A wave whose 64 output columns (or rows) all lie beyond N (M) has nothing to multiply: it still stages its share
    // of the operand tiles but leaves its SIMD's MFMA pipe to the other resident workgroups (N = 320: 1/6 of the waves).
    const bool wave_live = (n0 + wn * 64 < p.N) && (m0 + wm * 64 < p.M);
    auto multiply = [&](int cur) {
        if (!wave_live) return;
        const h16* xs = Xs + cur * XT * TILE + (wm * 64 + l31) * LDSLD;
        const h16* ws = Ws + cur * XT * TILE + (wn * 64 + l31) * LDSLD;
        if constexpr (FUSED) {
            // x = x0 + x1, w = w0 + w1 (bf16 pieces): x1 w0 + x0 w1 + x0 w0 per fragment pair — the bf16 x bf16 products are
            // exact in the fp32 accumulator; what is dropped (x1 w1) is 2^-18 relative.  Small terms first within a k-step.
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int off = ((ks * 2 + hi) ^ sw) << 3;
                h16x8 wf[2][2], xf[2][2];          // [piece][32-row block]
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    wf[pl][0] = *reinterpret_cast<const h16x8*>(ws + pl * TILE + off);
                    wf[pl][1] = *reinterpret_cast<const h16x8*>(ws + pl * TILE + 32 * LDSLD + off);
                    xf[pl][0] = *reinterpret_cast<const h16x8*>(xs + pl * TILE + off);
                    xf[pl][1] = *reinterpret_cast<const h16x8*>(xs + pl * TILE + 32 * LDSLD + off);
                }
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    const int wp = term == 1 ? 1 : 0, xp = term == 0 ? 1 : 0;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
                            acc[ni][mi] = MFMA_32x32x16(wf[wp][ni], xf[xp][mi], acc[ni][mi]);
                }
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int off = ((ks * 2 + hi) ^ sw) << 3;
            h16x8 wf[2], xf[2];
            wf[0] = *reinterpret_cast<const h16x8*>(ws + off);
            wf[1] = *reinterpret_cast<const h16x8*>(ws + 32 * LDSLD + off);
            xf[0] = *reinterpret_cast<const h16x8*>(xs + off);
            xf[1] = *reinterpret_cast<const h16x8*>(xs + 32 * LDSLD + off);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[ni][mi] = MFMA_32x32x16(wf[ni], xf[mi], acc[ni][mi]);
        }
    };
    if constexpr (ONEBUF) {
        for (int kt = 0; kt < nkt; ++kt) {
            issue_tiles(kt, 0);
            __syncthreads();                 // vmcnt(0) + barrier: the tile has landed
            multiply(0);
            __syncthreads();                 // every wave is done reading before the next fetch overwrites the buffer
        }
    } else {
        issue_tiles(0, 0);
        __syncthreads();                     // drains the DMA (vmcnt(0)) before anyone reads the tile
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) issue_tiles(kt + 1, cur ^ 1);
            multiply(cur);
            __syncthreads();
        }
    }

    // ------------------------------------------------------------------ epilogue
    float* stg = reinterpret_cast<float*>(smem);
    constexpr int NPASS = ONEBUF ? 2 : 1, PROWS = BM / NPASS;  // SB / FUSED stage the tile in two 64-row passes (wave rows wm = pass)
    float* sbias = stg + PROWS * STGLD;
    // Per-group bias (a ResBlock's embedding term): when all rows of the tile belong to one group — always, for the
    // UNet's shapes — it is one more per-column constant and rides in the staged bias; a tile that straddles groups adds
    // it row by row in the store loop.
    bool gbias_rows = p.gbias != nullptr;
    if (p.gbias && !p.geglu) {
        const int mlast = (m0 + BM <= p.M ? m0 + BM : p.M) - 1;
        const int g0 = m0 / p.rows_per_group;
        if (g0 == mlast / p.rows_per_group) {
            gbias_rows = false;
            if (tid < BN) {
                float b = (p.bias && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
                if (n0 + tid < p.N) b += p.gbias[(int64_t)g0 * p.N + n0 + tid];
                sbias[tid] = b;
            }
        }
    }
    if (gbias_rows || !p.gbias || p.geglu) {
        if (tid < BN) sbias[tid] = (p.bias && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
    }
    __syncthreads();

    const float alpha = p.alpha;
    const int NT = p.geglu ? BN / 2 : BN;
    const int Nout = p.geglu ? p.N / 2 : p.N;
    const int nout0 = p.geglu ? n0 / 2 : n0;
    const int cpr = NT / 8;
    const h16* R = (p.R && p.res_fp32 == KIND_OPERAND) ? reinterpret_cast<const h16*>(p.R) + bz * p.sR : nullptr;
    const float* Rf = (p.R && p.res_fp32 == KIND_F32) ? reinterpret_cast<const float*>(p.R) + bz * p.sR : nullptr;
    const _Float16* Rh = (p.R && p.res_fp32 == KIND_F16) ? reinterpret_cast<const _Float16*>(p.R) + bz * p.sR : nullptr;
    float gs[8], gq[8];            // GroupNorm partials of this thread's 8 output channels (p.stats)
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
    if (NPASS == 1 || wm == pass) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int ml = (NPASS == 1 ? wm * 64 : 0) + mi * 32 + l31;
        if (!p.geglu) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wn * 64 + ni * 32 + 8 * g + 4 * hi;
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = alpha * acc[ni][mi][4 * g + j] + sbias[nl + j];
                    if (p.act) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = PLANES > 1 ? gelu_erf_f(v[j]) : gelu_fast(v[j]);
                    }
                    *reinterpret_cast<f32x4*>(&stg[ml * STGLD + nl]) = v;
                }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 64 + 8 * g + 4 * hi;      // value columns; gates sit 32 further
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float val = alpha * acc[0][mi][4 * g + j] + sbias[nl + j];
                    const float gate = alpha * acc[1][mi][4 * g + j] + sbias[nl + 32 + j];
                    v[j] = val * (PLANES > 1 ? gelu_erf_f(gate) : (phi ? gelu_lut(gate, phis) : gelu_fast(gate)));
                }
                *reinterpret_cast<f32x4*>(&stg[ml * STGLD + wn * 32 + 8 * g + 4 * hi]) = v;
            }
        }
    }
    }
    __syncthreads();

    // cpr is 16 (or 8 with GEGLU): a thread keeps ONE 8-channel chunk (tid & (cpr - 1)) and walks rows 256 / cpr apart — no
    // per-chunk division / modulo (the short-K GEMMs spent 8-12 VALU instructions per MFMA, most of them here)
    const int cshift = p.geglu ? 3 : 4;
    const int cc = tid & (cpr - 1);
    const int n = nout0 + cc * 8;
    const int nvalid = n >= Nout ? 0 : ((Nout - n) < 8 ? (Nout - n) : 8);
    if (sub) {
        // sub-pixel conv: bias only (host-checked), rows scattered to this parity class's pixels of the full-resolution image
        for (int row = tid >> 4; row < PROWS; row += 16) {
            const int m = m0 + pass * PROWS + row;
            if (m >= p.M || nvalid == 0) break;
            const int hw = p.Hout * p.Wout;
            const int f = m / hw, r = m - f * hw;
            const int oy = r / p.Wout, ox = r - oy * p.Wout;
            const int64_t yoff = (((int64_t)(f * p.Hout + oy) * 2 + dy0) * (2 * p.Wout) + 2 * ox + dx0) * p.ldy + n;
            float v[8];
            const f32x4 a = *reinterpret_cast<const f32x4*>(&stg[row * STGLD + cc * 8]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&stg[row * STGLD + cc * 8 + 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
            const bool wide = nvalid == 8 && (vflags & VF_Y);
            if (p.out_fp32 == KIND_F16) {
                _Float16* yp = reinterpret_cast<_Float16*>(p.Y) + yoff;
                if (wide) store8_f16(yp, v);
                else for (int j = 0; j < nvalid; ++j) yp[j] = f16_sat(v[j]);
            } else if (p.out_fp32) {
                float* yp = reinterpret_cast<float*>(p.Y) + yoff;
                for (int j = 0; j < nvalid; ++j) yp[j] = v[j];
            } else {
                h16* yp = reinterpret_cast<h16*>(p.Y) + yoff;
                if (wide) store8_operand(yp, p.ldy / PLANES, v);
                else for (int j = 0; j < nvalid; ++j) store1_operand(yp + j, p.ldy / PLANES, v[j]);
            }
        }
    } else
    for (int row = tid >> cshift; row < PROWS; row += (256 >> cshift)) {
        const int m = m0 + pass * PROWS + row;
        if (m >= p.M || nvalid == 0) break;
        float v[8];
        {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&stg[row * STGLD + cc * 8]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&stg[row * STGLD + cc * 8 + 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
        }
        if (gbias_rows) {
            const float* gb = p.gbias + (int64_t)(m / p.rows_per_group) * Nout + n;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += gb[j];
        }
        if (R) {
            const h16* rp = R + (int64_t)m * p.ldr + n;
            if (nvalid == 8 && (vflags & VF_R)) {
                float rr[8];
                load8_operand(rp, p.ldr / PLANES, rr);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rr[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += load1_operand(rp + j, p.ldr / PLANES);
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
        if (p.stats) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = (j < nvalid) ? (p.out_fp32 == KIND_F32 ? v[j] : (p.out_fp32 == KIND_F16 ? (float)f16_sat(v[j]) : operand_round(v[j]))) : 0.f;
                gs[j] += t; gq[j] = fmaf(t, t, gq[j]);
            }
        }
        const int64_t yoff = bz * p.sY + (int64_t)m * p.ldy + n;
        if (p.out_fp32 == KIND_F16) {
            _Float16* yp = reinterpret_cast<_Float16*>(p.Y) + yoff;
            if (nvalid == 8 && (vflags & VF_Y)) {
                store8_f16(yp, v);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < nvalid) yp[j] = f16_sat(v[j]);
            }
        } else if (p.out_fp32) {
            float* yp = reinterpret_cast<float*>(p.Y) + yoff;
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
            h16* yp = reinterpret_cast<h16*>(p.Y) + yoff;
            if (nvalid == 8 && (vflags & VF_Y)) {
                store8_operand(yp, p.ldy / PLANES, v);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < nvalid) store1_operand(yp + j, p.ldy / PLANES, v[j]);
            }
        }
    }
    if (NPASS > 1) __syncthreads();
    }
    if (p.stats) {
        // thread t always handled channel chunk t % cpr: fold the 256 / cpr threads of a chunk in a fixed order
        if (NPASS == 1) __syncthreads();
        float* red = stg;
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid * 17 + j] = gs[j]; red[tid * 17 + 8 + j] = gq[j]; }
        __syncthreads();
        if (tid < cpr * 16) {
            const int cc = tid >> 4, j = tid & 15;
            float t = 0.f;
            for (int k = cc; k < 256; k += cpr) t += red[k * 17 + j];
            const int n = nout0 + cc * 8 + (j & 7);
            if (n < Nout) p.stats[((int64_t)(m0 / BM) * Nout + n) * 2 + (j >> 3)] = t;
        }
    }
}

// Lazily created per-device state (a zero page, the Phi table, the kernels' LDS opt-in): keyed by the current device so
// that one process may drive several GPUs.
constexpr int MAX_DEVICES = 64;
int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return -1;
    return dev;
}

const h16* zero_page() {
    static h16* page[MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0) return nullptr;
    if (!page[dev]) {
        void* ptr = nullptr;
        if (hipMalloc(&ptr, 256) != hipSuccess || hipMemset(ptr, 0, 256) != hipSuccess) return nullptr;
        page[dev] = static_cast<h16*>(ptr);
    }
    return page[dev];
}

// Phi(x) = 0.5 erfc(-x / sqrt 2) at x = -8 + i / 64, i = 0..1024, built once on the host in double precision.
// Variant switch GELU_LUT=0 keeps the erf polynomial (A/B measurements).
const float* phi_table() {
    static float* tabs[MAX_DEVICES] = {};
    static int mode = -1;
    if (mode < 0) mode = mudg_variant("GELU_LUT", 1);
    if (!mode || PLANES > 1) return nullptr;          // the split-operand builds evaluate erf exactly
    const int dev = current_device();
    if (dev < 0) return nullptr;
    float*& tab = tabs[dev];
    if (!tab) {
        float host[PHI_N + 4];
        for (int i = 0; i < PHI_N + 4; ++i) {
            const double x = -8.0 + (double)(i < PHI_N ? i : PHI_N) / 64.0;
            host[i] = (float)(0.5 * erfc(-x * 0.70710678118654752440));
        }
        void* ptr = nullptr;
        if (hipMalloc(&ptr, sizeof(host)) != hipSuccess || hipMemcpy(ptr, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess)
            return nullptr;
        tab = static_cast<float*>(ptr);
    }
    return tab;
}

template <int MODE, bool FAST, bool SB = false>
int launch(const MudgGemmDesc& d, int vflags, hipStream_t s) {
    static bool attr_done[MAX_DEVICES] = {};
    const h16* zp = zero_page();
    if (!zp) MUDG_FAIL(MUDG_ELAUNCH, "gemm: could not allocate the zero page");
    bool& attr_set = attr_done[current_device()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<MODE, FAST, SB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem_main(FAST, SB) + PHI_BYTES);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    dim3 grid(tiles, 1, d.batch);
    const float* phi = d.geglu ? phi_table() : nullptr;
    hipLaunchKernelGGL((gemm_kernel<MODE, FAST, SB>), grid, dim3(256), smem_main(FAST, SB) + (phi ? PHI_BYTES : 0), s, d, vflags, zp, phi);
    return mudg_check_launch("mudg_gemm");
}

// Problems with at least three tiles per CU (eight for the 3x3 convs) go to the single-buffer / 4-workgroups-per-CU
// variant (see gemm_kernel): measured faster at every K (MDM1024 shapes: +3...+20 %); with fewer tiles the
// double-buffered 2-per-CU kernel wins.
// Variant switch GEMM_SB=0 disables it, =2 forces it for every FAST problem.  (bf16x3 build: the fused-piece kernel is the
// only FAST kernel, see fused_planes.)
bool use_single_buffer(const MudgGemmDesc& d) {
    if (fused_planes(true)) return true;
    static int mode = -1;
    if (mode < 0) mode = mudg_variant("GEMM_SB", 1);
    if (mode == 0) return false;
    if (mode == 2) return true;
    const int64_t tiles = (int64_t)((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN) * d.batch;
    return tiles >= (d.mode == 1 ? 2048 : 768);
}

}  // namespace

// Whether the buffer-descriptor (FAST) kernels can run this problem.  Variant switch GEMM_FAST=0 forces the generic
// address path (for A/B measurements and tests of both paths).
bool mudg_gemm_fast_ok(const MudgGemmDesc& d) {
    static int en = -1;
    if (en < 0) en = mudg_variant("GEMM_FAST", 1);
    if (!en) return false;
    const int cin = d.mode == 0 ? d.K : d.Cin;
    if ((d.K & 63) || (cin & 63) || (d.csplit & 63)) return false;
    if (d.mode == 1 && d.upsample) return false;
    const int64_t ld = d.X2 && d.ldx2 > d.ldx ? d.ldx2 : d.ldx;
    int64_t rel = 255, soff = (int64_t)cin * 2 + (PLANES > 1 ? ld * 2 : 0);
    if (d.mode == 1) {
        rel = (int64_t)(255 / (d.Hout * d.Wout) + 2) * d.Hin * d.Win;
        soff += (int64_t)(2 * d.Win + 2) * ld * 2;
    } else if (d.mode == 2) {
        soff += (int64_t)2 * d.HW * ld * 2;
    }
    const int64_t lim = (int64_t)1 << 31;
    return rel * ld * 2 + 128 + soff + 16 < lim && (int64_t)(255 + (PLANES > 1)) * d.ldw * 2 + (int64_t)d.K * 2 + 144 < lim;
}

extern "C" int mudg_conv_subpixel_ok(const MudgGemmDesc* dp) {
    if (!dp) return 0;
    MudgGemmDesc d = *dp;
    if (d.mode != 1 || !d.subpixel || d.batch != 4 || d.stride != 1 || d.pad != 1 || d.upsample || !d.korder) return 0;
    if (d.X2 || d.R || d.gbias || d.stats || d.geglu || d.sX != 0 || d.sY != 0) return 0;
    if (d.Hout != d.Hin || d.Wout != d.Win || d.K != 4 * d.Cin || (d.Cin & 63)) return 0;
    d.csplit = d.Cin;
    return mudg_gemm_fast_ok(d) ? 1 : 0;
}

extern "C" int mudg_gemm(const MudgGemmDesc* dp, void* stream) {
    MUDG_REQUIRE(dp, "mudg_gemm: null descriptor");
    MudgGemmDesc d = *dp;
    MUDG_REQUIRE(d.X && d.W && d.Y, "mudg_gemm: null X/W/Y");
    MUDG_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "mudg_gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
    MUDG_REQUIRE(d.mode >= 0 && d.mode <= 2, "mudg_gemm: mode %d", d.mode);
    MUDG_REQUIRE((d.K & 7) == 0, "mudg_gemm: K=%d must be a multiple of 8", d.K);
    MUDG_REQUIRE(d.ldx % (8 * PLANES) == 0 && d.ldw % (8 * PLANES) == 0, "mudg_gemm: ldx=%d ldw=%d must be multiples of %d", d.ldx, d.ldw, 8 * PLANES);
    MUDG_REQUIRE(aligned16(d.X) && aligned16(d.W), "mudg_gemm: X/W must be 16-byte aligned");
    MUDG_REQUIRE((d.sX & 7) == 0 && (d.sW & 7) == 0, "mudg_gemm: batch strides must be multiples of 8");
    if (d.batch < 1) d.batch = 1;
    const int cin = d.mode == 0 ? d.K : d.Cin;
    if (!d.X2) d.csplit = cin;
    else {
        MUDG_REQUIRE(aligned16(d.X2) && d.ldx2 % (8 * PLANES) == 0, "mudg_gemm: X2 alignment");
        MUDG_REQUIRE(d.csplit > 0 && d.csplit < cin && (d.csplit & 7) == 0, "mudg_gemm: csplit=%d", d.csplit);
    }
    if (PLANES > 1) {      // plane p of an operand row sits p * (ld / PLANES) elements further: the planes must not overlap
        MUDG_REQUIRE(d.ldx / PLANES >= d.csplit && d.ldw / PLANES >= d.K && (!d.X2 || d.ldx2 / PLANES >= cin - d.csplit),
                     "mudg_gemm: split operands need ld / %d >= the logical width (ldx=%d ldx2=%d ldw=%d)", PLANES, d.ldx, d.ldx2, d.ldw);
        MUDG_REQUIRE(d.out_fp32 || d.ldy % PLANES == 0, "mudg_gemm: ldy=%d must be a multiple of %d", d.ldy, PLANES);
        MUDG_REQUIRE(!d.R || d.res_fp32 || d.ldr % PLANES == 0, "mudg_gemm: ldr=%d must be a multiple of %d", d.ldr, PLANES);
    }
    if (d.mode == 1) {
        MUDG_REQUIRE(d.Cin > 0 && (d.Cin & 7) == 0 && d.K == (d.subpixel ? 4 : 9) * d.Cin, "mudg_gemm: conv K=%d Cin=%d", d.K, d.Cin);
        if (d.subpixel)
            MUDG_REQUIRE(mudg_conv_subpixel_ok(&d), "mudg_gemm: subpixel needs batch 4, stride 1, pad 1, korder 1, Hout x Wout == Hin x Win, "
                         "no upsample / X2 / R / gbias / stats, and a problem the buffer-descriptor kernels accept");
        MUDG_REQUIRE(d.stride == 1 || d.stride == 2, "mudg_gemm: stride %d", d.stride);
        MUDG_REQUIRE(!(d.upsample && d.stride != 1), "mudg_gemm: upsample needs stride 1");
        MUDG_REQUIRE(d.Hin > 0 && d.Win > 0 && d.Hout > 0 && d.Wout > 0, "mudg_gemm: conv geometry");
        MUDG_REQUIRE(d.pad == 0 || d.pad == 1, "mudg_gemm: pad %d", d.pad);
        MUDG_REQUIRE(!d.korder || (d.Cin & 63) == 0, "mudg_gemm: korder=1 needs Cin %% 64 == 0");
        MUDG_REQUIRE(d.M % (d.Hout * d.Wout) == 0, "mudg_gemm: M not a whole number of frames");
    } else if (d.mode == 2) {
        MUDG_REQUIRE(d.Cin > 0 && (d.Cin & 7) == 0 && d.K == 3 * d.Cin, "mudg_gemm: tconv K=%d Cin=%d", d.K, d.Cin);
        MUDG_REQUIRE(d.T > 0 && d.HW > 0 && d.M % (d.T * d.HW) == 0, "mudg_gemm: tconv geometry");
    }
    if (d.geglu) MUDG_REQUIRE((d.N & 63) == 0, "mudg_gemm: geglu needs N %% 64 == 0");
    if (d.gbias) MUDG_REQUIRE(d.rows_per_group > 0 && d.batch == 1, "mudg_gemm: gbias needs rows_per_group");
    if (d.stats) MUDG_REQUIRE(d.batch == 1 && !d.geglu, "mudg_gemm: stats needs batch == 1 and no GEGLU");
    if (d.alpha == 0.f) d.alpha = 1.f;
    int vflags = 0;
    MUDG_REQUIRE(d.out_fp32 >= 0 && d.out_fp32 <= 2 && d.res_fp32 >= 0 && d.res_fp32 <= 2, "mudg_gemm: out_fp32 / res_fp32 are 0 (operand), 1 (fp32) or 2 (fp16)");
    const int ybytes = d.out_fp32 == KIND_F32 ? 4 : 2;
    if (aligned16(d.Y) && ((int64_t)d.ldy * ybytes) % (d.out_fp32 ? 16 : 16 * PLANES) == 0 && ((int64_t)d.sY * ybytes) % 16 == 0) vflags |= VF_Y;
    const int rbytes = d.res_fp32 == KIND_F32 ? 4 : 2;
    if (d.R && aligned16(d.R) && ((int64_t)d.ldr * rbytes) % (d.res_fp32 ? 16 : 16 * PLANES) == 0 && ((int64_t)d.sR * rbytes) % 16 == 0) vflags |= VF_R;

    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int fam = d.mode == 0 ? MUDG_FAM_GEMM : (d.mode == 1 ? MUDG_FAM_CONV : MUDG_FAM_TCONV);
    const int slot = mudg_prof_begin(fam, s);
    int rc;
    if (mudg_gemm_fast_ok(d)) {
        bool sb = use_single_buffer(d);
        if constexpr (fused_planes(true)) sb = true;
        if (sb)
            rc = d.mode == 0 ? launch<0, true, true>(d, vflags, s) : (d.mode == 1 ? launch<1, true, true>(d, vflags, s) : launch<2, true, true>(d, vflags, s));
        else if constexpr (!fused_planes(true))
            rc = d.mode == 0 ? launch<0, true, false>(d, vflags, s) : (d.mode == 1 ? launch<1, true, false>(d, vflags, s) : launch<2, true, false>(d, vflags, s));
        else
            rc = MUDG_EINVAL;
    } else {
        rc = d.mode == 0 ? launch<0, false>(d, vflags, s) : (d.mode == 1 ? launch<1, false>(d, vflags, s) : launch<2, false>(d, vflags, s));
    }
    const double flops = 2.0 * d.M * (double)d.N * d.K * d.batch;          // algorithmic (the split builds issue NSEG times as many)
    // algorithmic bytes at 16-bit storage: activations in (taps are re-reads of the same rows), weights, the result, and the
    // residual the epilogue adds when there is one
    const double nout = d.geglu ? d.N / 2 : d.N;
    double bytes = ((double)d.M * cin + (double)d.N * d.K + (double)d.M * nout * (d.R ? 2.0 : 1.0)) * 2.0 * d.batch;
    if (d.mode == 1 && d.subpixel) bytes = ((double)d.M * cin + 4.0 * d.N * d.K + 4.0 * d.M * nout) * 2.0;      // the low-res input once
    mudg_prof_end(slot, s, flops, bytes);
    return rc;
}
