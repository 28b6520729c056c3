// gemm256.hip — the large-tile variant of the contraction kernel (same MudgGemmDesc semantics as gemm.hip) for
// shapes with enough 256x256 tiles to fill the chip: 8 waves, 256x256x64 block tile, operands streamed into LDS by
// global_load_lds in HALF-TILE pieces that stay in flight across barriers (counted s_waitcnt vmcnt, raw s_barrier).
//
// Geometry.  Block rows (activations, "X") and block columns (weights, "W") are each split in two 128-row halves.
// LDS holds two K-tile buffers of four half-tiles each ([128][64] h16 = 16 KiB, XOR-swizzled exactly as in gemm.hip):
// 2 x 64 KiB = 128 KiB, one workgroup per CU, 2 waves per SIMD.  Waves form a 2 (M) x 4 (N) grid; wave (wm, wn) owns,
// in EACH X half, rows wm*64 + [0,64) and, in EACH W half, rows wn*32 + [0,32): eight 32x32 MFMA tiles = 128 fp32
// accumulators per lane.  A K-tile is computed in four quadrant phases (X half, W half) = (0,0) (0,1) (1,1) (1,0),
// 8 MFMAs each, re-using the fragments of the half that does not change.
//
// Pipeline.  While tile t is computed, the four half-tiles of tile t+1 are issued one per phase in the order the
// phases will need them (X0, W0, W1, X1), each wave contributing two 1-KiB DMA instructions per piece.  Nothing is
// drained to zero in the steady state: before a phase that needs a new piece the wave waits with s_waitcnt vmcnt(4)
// (its four most recent DMA instructions may stay in flight), then a raw s_barrier publishes the piece to the other
// waves.  Three barriers per 32 MFMAs instead of one full drain per 16.
//
// STATUS (round 1): bit-correct, but not yet faster than the 128x128 kernel, so it is opt-in (MUDG_GEMM256=1|2).
// Ablation on MI355X, conv 18x32 2560->1280 (K = 23040, 180 tiles): MFMA + barriers only 424 us, DMA + barriers only
// 406-498 us (8.5-10.5 TB/s L2->LDS), ds_read + barriers only 274 us, everything 823 us: the MFMA stream and the DMA
// stream do not overlap because all eight waves issue their DMA at the same barrier.  Tried and rejected this round:
// (a) putting the whole next tile in flight at the tile boundary (DMA-only 406 us, full kernel unchanged);
// (b) running the two wave groups one sub-phase apart with a barrier after every load / MFMA sub-phase (8 barriers
// per K-tile): correct, but slower (conv 530 vs 660 TFLOP/s) — the load sub-phase (tap decode + 64-bit address
// arithmetic + bounds tests per DMA instruction) is longer than the 8-MFMA sub-phase it should hide under.
// Next: make the load sub-phase cheap (incremental per-lane offsets into a buffer descriptor instead of recomputed
// 64-bit pointers) before staggering again.
//
// Epilogue: four passes (one per quadrant) through the same fp32 LDS tile as gemm.hip; GEGLU is applied in the
// coalesced pass (value / gate columns of the usual [32 value | 32 gate] packing sit 32 apart in the staged tile).
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BK = 64;
constexpr int HROWS = 128;                     // rows per half-tile
constexpr int HTILE = HROWS * 64;              // elements per half-tile
constexpr int KBUF = 4 * HTILE;                // elements per K-tile buffer: X0, X1, W0, W1
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

#define WAIT_VM4() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// raw barrier (no vmcnt drain, unlike __syncthreads) fenced against compiler motion of LDS accesses on both sides
#define BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const MudgGemmDesc p, const int vflags, const h16* __restrict__ zpage, const int ablate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* L = reinterpret_cast<h16*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int ntn = (p.N + 255) >> 8;
    int tile;
    {
        const int total = gridDim.x, q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int tm = tile / ntn, tn = tile - tm * ntn;
    const int m0 = tm * 256, n0 = tn * 256;
    const int64_t bz = blockIdx.z;
    const h16* X = reinterpret_cast<const h16*>(p.X) + bz * p.sX;
    const h16* X2 = p.X2 ? reinterpret_cast<const h16*>(p.X2) + bz * p.sX : nullptr;
    const h16* W = reinterpret_cast<const h16*>(p.W) + bz * p.sW;

    // DMA geometry: a half-tile is staged by all 8 waves, wave w rows [16w, 16w+16) with two instructions;
    // in instruction i lane l lands in row 16w + 8i + (l >> 3), slot l & 7 and fetches chunk slot ^ ((row >> 1) & 7).
    const int rsub = lane >> 3, slot = lane & 7;
    int ch[2], rl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rl[i] = 16 * wave + 8 * i + rsub;
        ch[i] = slot ^ ((rl[i] >> 1) & 7);
    }
    // per (half h, instr i) activation row state
    int rm[2][2], ra[2][2], rb[2][2], rc[2][2];
    bool rv[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + h * HROWS + rl[i];
            rm[h][i] = m;
            rv[h][i] = m < p.M;
            ra[h][i] = rb[h][i] = rc[h][i] = 0;
            if (MODE == 1) {
                const int hw = p.Hout * p.Wout;
                const int f = m / hw, r = m - f * hw;
                const int oy = r / p.Wout, ox = r - oy * p.Wout;
                ra[h][i] = f * p.Hin * p.Win;
                rb[h][i] = oy * p.stride - p.pad;
                rc[h][i] = ox * p.stride - p.pad;
            } else if (MODE == 2) {
                rb[h][i] = (m / p.HW) % p.T;
            }
        }
    const bool tap_uniform = MODE != 0 && (p.Cin & 63) == 0;

    auto issue_x = [&](int kt, int h) {
        const int k0 = kt * BK;
        int tap_u = 0, c_u = 0;
        if (MODE != 0 && tap_uniform) {
            if (MODE == 1 && p.korder) { const int slab = kt / 9; tap_u = kt - slab * 9; c_u = slab * 64; }
            else { tap_u = k0 / p.Cin; c_u = k0 - tap_u * p.Cin; }
        }
        h16* dst = L + (kt & 1) * KBUF + h * HTILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + ch[i] * 8;
            const bool kv = k < p.K;
            const h16* src = zpage;
            if (MODE == 0) {
                if (rv[h][i] && kv)
                    src = (k < p.csplit) ? X + (int64_t)rm[h][i] * p.ldx + k : X2 + (int64_t)rm[h][i] * p.ldx2 + (k - p.csplit);
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
                    int iy = rb[h][i] + dy, ix = rc[h][i] + dx;
                    const bool ok = rv[h][i] && kv && iy >= 0 && iy < hlim && ix >= 0 && ix < wlim;
                    if (p.upsample) { iy >>= 1; ix >>= 1; }
                    if (ok) src = base + (int64_t)(ra[h][i] + iy * p.Win + ix) * ld + cc;
                } else {
                    const int it = rb[h][i] + tap - 1;
                    if (rv[h][i] && kv && it >= 0 && it < p.T)
                        src = base + ((int64_t)rm[h][i] + (int64_t)(tap - 1) * p.HW) * ld + cc;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + (16 * wave + 8 * i) * 64), 16, 0, 0);
        }
    };
    auto issue_w = [&](int kt, int g) {
        const int k0 = kt * BK;
        h16* dst = L + (kt & 1) * KBUF + (2 + g) * HTILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + ch[i] * 8;
            const int n = n0 + g * HROWS + rl[i];
            const h16* src = (n < p.N && k < p.K) ? W + (int64_t)n * p.ldw + k : zpage;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + (16 * wave + 8 * i) * 64), 16, 0, 0);
        }
    };

    f32x16 acc[2][2][2];     // [g][h][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    const int sw = (l31 >> 1) & 7;
    const int xrow = (wm * 64 + l31) * 64, wrow = (wn * 32 + l31) * 64;

    h16x8 xf[2][4], wf[4];
    auto load_x = [&](const h16* buf, int h) {
        const h16* base = buf + h * HTILE + xrow;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                xf[mi][ks] = *reinterpret_cast<const h16x8*>(base + mi * 32 * 64 + (((ks * 2 + hi) ^ sw) << 3));
    };
    auto load_w = [&](const h16* buf, int g) {
        const h16* base = buf + (2 + g) * HTILE + wrow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            wf[ks] = *reinterpret_cast<const h16x8*>(base + (((ks * 2 + hi) ^ sw) << 3));
    };
    auto mma = [&](int g, int h) {
        WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                acc[g][h][mi] = MFMA_32x32x16(wf[ks], xf[mi][ks], acc[g][h][mi]);
        __builtin_amdgcn_s_setprio(0);
    };

    // prologue: all four pieces of tile 0, in the order the phases consume them
    issue_x(0, 0); issue_w(0, 0); issue_w(0, 1); issue_x(0, 1);

    for (int kt = 0; kt < nk; ++kt) {
        const h16* buf = L + (kt & 1) * KBUF;
        const bool more = (kt + 1 < nk) && !(ablate & 1);
        // ---- tile boundary: X0, W0 of this tile must have landed (W1, X1 may still fly); the barrier also retires
        // the other buffer (last read during tile kt-1), into which tile kt+1 is streamed one piece per phase.
        WAIT_VM4();
        BARRIER();
        // phase (X0, W0)
        if (more) issue_x(kt + 1, 0);
        if (!(ablate & 2) || kt == 0) { load_x(buf, 0); load_w(buf, 0); }
        if (!(ablate & 4)) mma(0, 0);
        // phase (X0, W1): needs W1
        if (more) { WAIT_VM4(); } else { WAIT_VM0(); }
        BARRIER();
        if (more) issue_w(kt + 1, 0);
        if (!(ablate & 2)) load_w(buf, 1);
        if (!(ablate & 4)) mma(1, 0);
        // phase (X1, W1): needs X1
        if (more) { WAIT_VM4(); } else { WAIT_VM0(); }
        BARRIER();
        if (more) issue_w(kt + 1, 1);
        if (!(ablate & 2)) load_x(buf, 1);
        if (!(ablate & 4)) mma(1, 1);
        // phase (X1, W0)
        if (more) issue_x(kt + 1, 1);
        if (!(ablate & 2)) load_w(buf, 0);
        if (!(ablate & 4)) mma(0, 1);
        if (ablate & 4) { asm volatile("" :: "v"(xf[0][0]), "v"(xf[1][3]), "v"(wf[0]), "v"(wf[3])); }
    }
    WAIT_VM0();
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
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int ml = wm * 64 + mi * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = wn * 32 + 8 * q + 4 * hi;
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = alpha * acc[g][h][mi][4 * q + j] + sbias[nl + j];
                    if (p.act && !p.geglu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = gelu_fast2(v[j]);
                    }
                    *reinterpret_cast<f32x4*>(&stg[ml * STGLD + nl]) = v;
                }
            }
            __syncthreads();
            const int NT = p.geglu ? 64 : 128;
            const int nout0 = p.geglu ? nq / 2 : nq;
            const int cpr = NT / 8;
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
        }
}

template <int MODE>
int launch256(const MudgGemmDesc& d, int vflags, const h16* zp, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm256: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ((d.M + 255) / 256) * ((d.N + 255) / 256);
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("MUDG_ABLATE"); ablate = e ? atoi(e) : 0; }
    hipLaunchKernelGGL(gemm256_kernel<MODE>, dim3(tiles, 1, d.batch), dim3(512), SMEM_BYTES, s, d, vflags, zp, ablate);
    return mudg_check_launch("mudg_gemm[256]");
}

}  // namespace

// Called by mudg_gemm (gemm.hip) once the descriptor is validated and the large-tile path is selected.
int mudg_gemm256_dispatch(const MudgGemmDesc& d, int vflags, const h16* zpage, hipStream_t s) {
    if (d.mode == 0) return launch256<0>(d, vflags, zpage, s);
    if (d.mode == 1) return launch256<1>(d, vflags, zpage, s);
    return launch256<2>(d, vflags, zpage, s);
}
