// pgemm.hip — the persistent form of the 128 x 128 MFMA contraction kernel (plain GEMM, 3x3 conv, temporal 3-tap conv as
// implicit GEMMs; descriptor loader; 16-bit builds and the fused-piece bf16x3 build).  Tiling, LDS image, swizzle and the
// DMA addressing scheme are gemm.hip's; what is different is everything around the K loop.
//
// A one-tile workgroup runs  setup -> first fetch (a full DMA latency) -> K loop -> two staged epilogue passes (four barriers,
// one wave row idle during each)  strictly in sequence.  Here:
//   * a workgroup is PERSISTENT: WGS (4 / 3 / 2) per CU, each walking its XCD's contiguous tile range with a stride of the
//     workgroups on that XCD — the same tiles in flight per XCD, in the same 8 x 8 patch order, as the hardware dispatcher gives
//     the one-tile-per-workgroup kernel;
//   * the NEXT tile's first K-tile is DMA'd into the (free) stage before the current tile's epilogue starts, together with its
//     column constants: the fetch latency and the epilogue overlap inside the workgroup;
//   * the epilogue goes STRAIGHT from the accumulators to HBM: no fp32 LDS staging tile, no passes, no barriers.  A lane of a
//     32 x 32 MFMA result holds output channels 8 g + 4 hi + {0..3} (g = 0..3) of one pixel — 8-byte pieces, which measured
//     3.3 TB/s as stores (tools/ubench/store_pattern.hip).  The weight-fragment ROWS are therefore read permuted (bits 2 and 3
//     of the row index swapped: free, the 16-lane ds_read_b128 groups stay conflict-free because the permutation is a bijection
//     mod 16): accumulator registers 8 q .. 8 q + 7 of lane (l31, hi) are then the 8 CONSECUTIVE channels 16 q + 8 hi .. + 7 of
//     pixel l31, one 16-byte store each, 32 rows x 32 contiguous bytes per store instruction — 5.1-6.2 TB/s against 5.4-6.6
//     of fully coalesced rows;
//   * a residual SEEDS the accumulators (alpha = 1): its 16-byte row-per-lane loads are issued at the top of the tile and land
//     under the wait for the first K-tile.  (Loaded in the epilogue they would queue behind the previous piece's stores — vmcnt
//     counts a wave's loads and stores in issue order on gfx9 — one store round trip per piece.)
//   * GroupNorm partials: a lane adds its two rows, five DPP adds fold the 32 pixels of a half-wave (fixed order), the two wave
//     rows meet in 4 KiB of LDS — the only barrier of the epilogue, and only for problems that ask for partials;
//   * registers: the DMA lane offsets of the plain and temporal problems are tile-invariant (row clamping is the buffer
//     descriptor's num_records; the tile lives in the descriptor base), result rows are 32-bit offsets from a scalar tile base:
//     without seeds / partials the 4-per-CU variant fits 128 VGPRs with no spill (template RS);
//   * the waits for a stage are explicit (stage_barrier): hipcc's alias-based vmcnt bookkeeping of LDS-DMA waited for only some
//     of a stage's pieces in this loop shape (wrong results, caught by tests/test_kernels_gpu.py);
//   * LDS: one (two) K-tile stage(s) + 5 KiB (column constants in two parities, Phi table | partial hand-off) = 37 (69) KiB.
// Where it wins and where it does not: gemm.hip, persistent_ok().  Summation order over K is the one-tile kernels' (one tile =
// one workgroup = K-tiles in order; a residual enters the sum first instead of last), so results do not depend on M or on the
// batch a clip travels in.
#include "gemm_shared.h"
#include <type_traits>

#if MUDG_PLANES <= 2
namespace {

constexpr int TILE = 128 * LDSLD;                        // elements of one operand tile (128 rows x 64 k)
constexpr bool FUSEDP = fused_planes(true);              // bf16x3 build: both pieces of both operands in one stage
constexpr int XT = FUSEDP ? PLANES : 1;
constexpr int STAGE_BYTES = XT * 2 * TILE * 2;
constexpr int PK_SBIAS_BYTES = 2 * 128 * 4;
constexpr int PK_SRED_BYTES = 2 * 2 * 128 * 2 * 4;       // [parity][wave row][channel][sum | sum of squares]
constexpr int PK_TAIL_BYTES = PK_SBIAS_BYTES + (PHI_BYTES > PK_SRED_BYTES ? PHI_BYTES : PK_SRED_BYTES);
constexpr int pk_stages(int wgs) { return (wgs > 2 || FUSEDP) ? 1 : 2; }
constexpr int pk_smem(int wgs) { return pk_stages(wgs) * STAGE_BYTES + PK_TAIL_BYTES; }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
// Sum over the 32 lanes of each half-wave in a fixed order; lanes 16-31 / 48-63 end up holding their half's total.
__device__ __forceinline__ float half_wave_sum(float v) {
    v = dpp_add<0xB1, 0xf>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);      // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);      // row_mirror
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    return v;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_n(const h16* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(base), 0, (int)bytes, 0x00020000);
}

// hipcc's own vmcnt bookkeeping for LDS-DMA is alias-based and was seen to wait for only some of the pieces of a stage in this
// loop shape: every wave waits for ITS pieces explicitly, the barrier then publishes the whole stage.
__device__ __forceinline__ void stage_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// RS: the problem may carry a residual (accumulator seed) or ask for GroupNorm partials.  Without them the 4-per-CU variant fits
// its 128 registers; with them it does not (seed addressing + the partial sums of a piece), so those problems run 2 per CU.
template <int MODE, bool GEGLU, int WGS, bool RS>
__global__ __launch_bounds__(256, FUSEDP ? 2 : WGS) void pgemm_kernel(const MudgGemmDesc p, const int vflags, const float* __restrict__ phi,
                                                                     const int ntm, const int ntn) {
    constexpr int NI = 2, MI = 2, NSTAGE = pk_stages(WGS);
    constexpr bool ONEBUF = NSTAGE == 1;
    constexpr bool Y8OK = PLANES == 1 && WGS == 2 && !GEGLU;      // the fused MX-fp8 copy needs the registers of the 2-per-CU variant
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* Xs = reinterpret_cast<h16*>(smem);
    h16* Ws = Xs + NSTAGE * XT * TILE;
    float* sbias = reinterpret_cast<float*>(smem + NSTAGE * STAGE_BYTES);
    float* phis = sbias + 256;                       // GEGLU problems; the partial hand-off of the others lives in the same bytes
    float* sred = sbias + 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, hi = lane >> 5;
    if (GEGLU && phi) {
        for (int t4 = tid * 4; t4 < PHI_N; t4 += 256 * 4)
            *reinterpret_cast<f32x4*>(&phis[t4]) = *reinterpret_cast<const f32x4*>(&phi[t4]);
        if (tid == 0) phis[PHI_N] = phi[PHI_N];
    }

    // ---- this workgroup's tiles: XCD x = blockIdx % 8 owns the contiguous range [lo, lo + cnt) of the flattened (batch, tile)
    // index space; its workgroups (blockIdx / 8 = 0 .. nwg - 1) take every nwg-th tile of it.
    const int per_batch = ntm * ntn;
    int t_next, t_end, t_step;
    {
        const int total = per_batch * p.batch;
        const int q8 = total >> 3, r8 = total & 7;
        const int xcd = blockIdx.x & 7;
        const int lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        t_step = ((int)gridDim.x - xcd + 7) >> 3;
        t_next = lo + (int)(blockIdx.x >> 3);
        t_end = lo + q8 + (xcd < r8 ? 1 : 0);
    }
    if (t_next >= t_end) return;

    const bool subp = MODE == 1 && p.subpixel;
    const int ntaps = subp ? 4 : 9;
    const int nk = p.K / BK;

    // ---- DMA geometry.  Wave w stages rows [32 w, 32 w + 32) of both operand tiles, one 1-KiB instruction per 8 rows: in
    // instruction i, lane l lands in row 32 w + 8 i + (l >> 3), slot l & 7, and fetches the chunk (l & 7) ^ ((row >> 1) & 7) of
    // that row — chunk offsets cb0 (i even) and cb0 ^ 64 (i odd).  Tile-invariant; the tile is in the descriptor bases.
    const int rl0 = 32 * wave + (lane >> 3);
    const unsigned cb0 = (unsigned)(((lane & 7) ^ ((lane >> 4) & 3)) * 16);
    const unsigned wlane = (unsigned)rl0 * (unsigned)p.ldw * 2u + cb0;          // W: row rl0 (+ 8 i rows at issue time)
    // ---- per-tile state (set by setup(), consumed by issue())
    int m0 = 0, n0 = 0, zb = 0, dy0 = 0, dx0 = 0;
    __amdgpu_buffer_rsrc_t rX, rX2, rW;
    int rel[MODE == 1 ? 4 : 1];                     // 3x3 convs: source pixel of the lane's four rows, relative to the tile's first
    unsigned vmask0 = 0, vmask1 = 0;                // tap validity bits: 9 (3) per row, rows 0-1 (0-3) in vmask0, rows 2-3 in vmask1
    int tap_s = 0, c_s = 0;

    auto setup = [&](int t) {
        const int z = t / per_batch;
        const int tile = t - z * per_batch;
        int tm, tn;
        {
            const int per = 8 * ntn, g = tile / per, first = g * 8;
            const int gsz = (ntm - first) < 8 ? (ntm - first) : 8;
            const int r = tile - g * per;
            tn = r / gsz;
            tm = first + (r - tn * gsz);
        }
        m0 = tm * 128; n0 = tn * 128; zb = z;
        dy0 = subp ? (z >> 1) : 0; dx0 = subp ? (z & 1) : 0;
        const h16* X = reinterpret_cast<const h16*>(p.X) + (int64_t)z * p.sX;
        const h16* X2 = p.X2 ? reinterpret_cast<const h16*>(p.X2) + (int64_t)z * p.sX : nullptr;
        const h16* W = reinterpret_cast<const h16*>(p.W) + (int64_t)z * p.sW;
        const int wrows = (p.N - n0) < 128 ? (p.N - n0) : 128;
        rW = make_rsrc_n(W + (int64_t)n0 * p.ldw, (unsigned)wrows * (unsigned)p.ldw * 2u);       // rows beyond N read as zero
        if (MODE == 0) {
            const int xrows = (p.M - m0) < 128 ? (p.M - m0) : 128;                              // rows beyond M read as zero
            rX = make_rsrc_n(X + (int64_t)m0 * p.ldx, (unsigned)xrows * (unsigned)p.ldx * 2u);
            rX2 = X2 ? make_rsrc_n(X2 + (int64_t)m0 * p.ldx2, (unsigned)xrows * (unsigned)p.ldx2 * 2u) : rX;
        } else if (MODE == 2) {
            rX = make_rsrc(X + ((int64_t)m0 - p.HW) * p.ldx);
            rX2 = X2 ? make_rsrc(X2 + ((int64_t)m0 - p.HW) * p.ldx2) : rX;
            unsigned mask = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + rl0 + 8 * i;
                const int tt = (m / p.HW) % p.T;
                if (m < p.M) {
#pragma unroll
                    for (int tp = 0; tp < 3; ++tp) {
                        const int it = tt + tp - 1;
                        if (it >= 0 && it < p.T) mask |= 1u << (3 * i + tp);
                    }
                }
            }
            vmask0 = mask;
        } else {
            const int hw = p.Hout * p.Wout;
            int64_t pix0;
            {
                const int f = m0 / hw, r = m0 - f * hw;
                const int oy = r / p.Wout, ox = r - oy * p.Wout;
                pix0 = ((int64_t)f * p.Hin + oy * p.stride) * p.Win + ox * p.stride;
            }
            const int64_t shift = -(int64_t)(p.pad * p.Win + p.pad);
            rX = make_rsrc(X + (pix0 + shift) * p.ldx);
            rX2 = X2 ? make_rsrc(X2 + (pix0 + shift) * p.ldx2) : rX;
            // (frame, oy, ox) of the lane's first row by division, of the next three by stepping 8 pixels
            int f, oy, ox;
            {
                const int m = m0 + rl0;
                f = m / hw;
                const int r = m - f * hw;
                oy = r / p.Wout; ox = r - oy * p.Wout;
            }
            unsigned mk[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool rv = m0 + rl0 + 8 * i < p.M;
                const int rb = oy * p.stride - p.pad, rc = ox * p.stride - p.pad;
                rel[MODE == 1 ? i : 0] = (int)(((int64_t)f * p.Hin * p.Win + (int64_t)(rb + p.pad) * p.Win + rc + p.pad) - pix0);
                unsigned mask = 0;
                if (rv) {
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const int iy = rb + (subp ? (tp >> 1) + dy0 : tp / 3), ix = rc + (subp ? (tp & 1) + dx0 : tp % 3);
                        if (tp < ntaps && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) mask |= 1u << tp;
                    }
                }
                mk[i] = mask;
                ox += 8;
                while (ox >= p.Wout) { ox -= p.Wout; ++oy; }
                while (oy >= p.Hout) { oy -= p.Hout; ++f; }
            }
            vmask0 = mk[0] | (mk[1] << 9);
            vmask1 = mk[2] | (mk[3] << 9);
        }
        tap_s = 0; c_s = 0;
    };

    auto issue = [&](int kt, int buf) {
        const bool s2 = c_s >= p.csplit;
        const int cc = s2 ? c_s - p.csplit : c_s;
        const int ld = s2 ? p.ldx2 : p.ldx;
        int soff;
        if (MODE == 0) soff = cc * 2;
        else if (MODE == 1) {
            int dy = tap_s / 3, dx = tap_s - 3 * dy;
            if (subp) { dy = (tap_s >> 1) + dy0; dx = (tap_s & 1) + dx0; }
            soff = ((dy * p.Win + dx) * ld + cc) * 2;
        }
        else soff = (tap_s * p.HW * ld + cc) * 2;
        const int soffw = kt * (BK * 2);
        // lane offsets of the four X pieces
        unsigned vx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned cb = (i & 1) ? (cb0 ^ 64u) : cb0;
            if (MODE == 1) {
                vx[i] = __umul24((unsigned)rel[MODE == 1 ? i : 0], (unsigned)ld * 2u) + cb;
                const unsigned bits = (i < 2 ? vmask0 : vmask1) >> (9 * (i & 1) + tap_s);
                vx[i] = (bits & 1u) ? vx[i] : OOB;
            } else {
                vx[i] = __umul24((unsigned)(rl0 + 8 * i), (unsigned)ld * 2u) + cb;
                if (MODE == 2) vx[i] = ((vmask0 >> (3 * i + tap_s)) & 1u) ? vx[i] : OOB;
            }
        }
#pragma unroll
        for (int pl = 0; pl < XT; ++pl) {
            const int so = soff + pl * (ld / PLANES) * 2, sow = soffw + pl * (p.ldw / PLANES) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lptr_t lx = (lptr_t)(Xs + (buf * XT + pl) * TILE + (32 * wave + 8 * i) * LDSLD);
                if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, lx, 16, (int)vx[i], so, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, lx, 16, (int)vx[i], so, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // W rows beyond N are never multiplied (their waves are idle, see wave_live): skip the zero-fill pieces
                if (n0 + 32 * wave + 8 * i < p.N || (32 * wave + 8 * i) < 64) {
                    const unsigned vw = wlane + (unsigned)(8 * i) * (unsigned)p.ldw * 2u + ((i & 1) ? ((cb0 ^ 64u) - cb0) : 0u);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(Ws + (buf * XT + pl) * TILE + (32 * wave + 8 * i) * LDSLD), 16,
                                                             (int)vw, sow, 0, 0);
                }
            }
        }
        if (MODE == 0) {
            c_s += BK;
        } else {                                   // select form: a branchy update sends tap_s / c_s to scratch memory
            const int t1 = tap_s + 1, c1 = c_s + BK;
            const bool slab = MODE == 1 && p.korder;
            const bool wrap = slab ? (t1 == ntaps) : (c1 == p.Cin);
            tap_s = slab ? (wrap ? 0 : t1) : (wrap ? t1 : tap_s);
            c_s = slab ? (wrap ? c1 : c_s) : (wrap ? 0 : c1);
        }
    };

    // The tile's 128 column constants (bias + the group bias: host-checked to be constant over a tile's rows): loaded into a
    // register beside the first DMA of the tile, parked in LDS at the top of its K loop.
    auto column_constant = [&]() -> float {
        float b = 0.f;
        if (tid < 128 && n0 + tid < p.N) {
            if (!GEGLU && p.gbias) b = p.gbias[(int64_t)(m0 / p.rows_per_group) * p.N + n0 + tid];
            if (p.bias) b += p.bias[n0 + tid];
        }
        return b;
    };

    f32x16 acc[NI][MI];
    // fragment rows: X rows as they are, W rows with bits 2 and 3 of the row index swapped (see the header)
    const int srow = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    bool wave_live = false;
    auto multiply = [&](int cur) {
        if (!wave_live) return;
        const int swx = (l31 >> 1) & 7, sww = (srow >> 1) & 7;
        const h16* xs = Xs + cur * XT * TILE + (wm * 64 + l31) * LDSLD;
        const h16* ws = Ws + cur * XT * TILE + (wn * 64 + srow) * LDSLD;
        if constexpr (FUSEDP) {
            // x = x0 + x1, w = w0 + w1 (bf16 pieces): x1 w0 + x0 w1 + x0 w0 per fragment pair, small terms first
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int offx = ((ks * 2 + hi) ^ swx) << 3, offw = ((ks * 2 + hi) ^ sww) << 3;
                h16x8 wf[2][NI], xf[2][MI];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) wf[pl][ni] = *reinterpret_cast<const h16x8*>(ws + pl * TILE + ni * 32 * LDSLD + offw);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) xf[pl][mi] = *reinterpret_cast<const h16x8*>(xs + pl * TILE + mi * 32 * LDSLD + offx);
                }
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    const int wp = term == 1 ? 1 : 0, xp = term == 0 ? 1 : 0;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            acc[ni][mi] = MFMA_32x32x16(wf[wp][ni], xf[xp][mi], acc[ni][mi]);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int offx = ((ks * 2 + hi) ^ swx) << 3, offw = ((ks * 2 + hi) ^ sww) << 3;
                h16x8 wf[NI], xf[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const h16x8*>(ws + ni * 32 * LDSLD + offw);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const h16x8*>(xs + mi * 32 * LDSLD + offx);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = MFMA_32x32x16(wf[ni], xf[mi], acc[ni][mi]);
            }
        }
    };

    // The accumulators start from the residual (host-checked: alpha == 1, no activation, no GEGLU): its loads are issued at the top
    // of the tile and land under the first K-tile's wait, in the row-per-lane pattern of the stores — the epilogue itself then
    // never waits for memory (a load issued after a piece's stores would queue behind their acknowledgements: vmcnt counts
    // loads and stores in order on gfx9).  16-byte accesses only: the host sends ragged / unaligned problems elsewhere.
    auto seed = [&]() {
        const int RK = (RS && !GEGLU && p.R && wave_live) ? p.res_fp32 : 3;
        if (RK == 3) {
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            return;
        }
        const int row0 = wm * 64 + l31;
        const int mrem = p.M - m0 - row0;
        const int rsz = RK == KIND_F32 ? 4 : 2;
        const char* Rb = reinterpret_cast<const char*>(p.R) + ((int64_t)zb * p.sR + (int64_t)m0 * p.ldr) * rsz;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int n = n0 + wn * 64 + ni * 32 + q * 16 + hi * 8;
                    const size_t re = (size_t)((unsigned)(row0 + 32 * mi) * (unsigned)p.ldr) + n;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = 0.f;
                    if (32 * mi < mrem && n < p.N) {
                        if (RK == KIND_F32) {
                            const f32x4 a = *reinterpret_cast<const f32x4*>(Rb + re * 4), b = *reinterpret_cast<const f32x4*>(Rb + re * 4 + 16);
#pragma unroll
                            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
                        } else if (RK == KIND_F16) {
                            load8_f16(reinterpret_cast<const _Float16*>(Rb) + re, v);
                        } else {
                            load8_operand(reinterpret_cast<const h16*>(Rb) + re, p.ldr / PLANES, v);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[ni][mi][8 * q + j] = v[j];
                }
    };

    // ---------------------------------------------------------------- epilogue: accumulators (+ bias, GEGLU) -> HBM
    // Host-checked: Nout % 8 == 0 and 16-byte-aligned rows (every piece is whole and one or two 16-byte stores), a group bias that
    // is constant over a tile (it rides in the column constants), no plain activation.
    const int Nout = GEGLU ? p.N >> 1 : p.N;
    auto epilogue = [&](const int em0, const int en0, const int ez, const int edy, const int edx, const int par) {
        const float alpha = p.alpha;
        const int OK = p.out_fp32;
        const int osz = OK == KIND_F32 ? 4 : 2;
        const float* sb = sbias + par * 128;
        const int nout0 = GEGLU ? en0 >> 1 : en0;
        // rows: 32-bit element offsets from the tile's (scalar) first row
        int64_t ybase;
        unsigned yo[MI];
        const int row0 = wm * 64 + l31;                       // + 32 mi
        if (subp) {
            const int hw = p.Hout * p.Wout;
            auto off = [&](int m) -> int64_t {
                const int f = m / hw, r = m - f * hw;
                const int oy = r / p.Wout, ox = r - oy * p.Wout;
                return (((int64_t)(f * p.Hout + oy) * 2 + edy) * (2 * p.Wout) + 2 * ox + edx) * p.ldy;
            };
            ybase = off(em0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) yo[mi] = (unsigned)(off(em0 + row0 + 32 * mi) - ybase);
        } else {
            ybase = (int64_t)ez * p.sY + (int64_t)em0 * p.ldy;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) yo[mi] = (unsigned)(row0 + 32 * mi) * (unsigned)p.ldy;
        }
        char* Yb = reinterpret_cast<char*>(p.Y) + ybase * osz;
        const int mrem = p.M - em0 - row0;                    // row mi is inside the matrix iff 32 mi < mrem
        float* red = sred + par * 512;                        // [wave row][channel][2]
        float y8v[Y8OK ? MI : 1][2][8], y8a[Y8OK ? MI : 1];    // fused MX-fp8 copy: the rounded values of a 32-column block's two pieces

        // one piece = this lane's 8 consecutive output channels of both of its rows; NIX = the 32-column block, Q = the half of it
        auto piece = [&](auto nitag, auto qtag) __attribute__((always_inline)) {
            constexpr int NIX = decltype(nitag)::value, Q = decltype(qtag)::value;
            const int cw = wn * 64 + NIX * 32 + Q * 16 + hi * 8;                       // column of the W tile (bias index) of the value
            const int co = GEGLU ? wn * 32 + Q * 16 + hi * 8 : cw;                      // column of the output tile
            const int n = nout0 + co;
            const bool cols = n < Nout;
            float bv[8], bg[8];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(&sb[cw]), b1 = *reinterpret_cast<const f32x4*>(&sb[cw + 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { bv[j] = b0[j]; bv[4 + j] = b1[j]; }
                if constexpr (GEGLU) {
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(&sb[cw + 32]), g1 = *reinterpret_cast<const f32x4*>(&sb[cw + 36]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bg[j] = g0[j]; bg[4 + j] = g1[j]; }
                }
            }
            float t0[8], gs[8], gq[8];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const bool live = 32 * mi < mrem && cols;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = alpha * acc[GEGLU ? 0 : NIX][mi][8 * Q + j] + bv[j];
                if constexpr (GEGLU) {
                    float gate[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) gate[j] = alpha * acc[1][mi][8 * Q + j] + bg[j];
                    if (phi) {                            // (decided per row, not per value: wgemm.hip, w_epilogue)
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= gelu_lut(gate[j], phis);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= gelu_fast(gate[j]);
                    }
                } else if (RS && p.stats) {
                    // partial sums over what is stored: the lane's two rows here, the 32 pixels of the half-wave below
                    float tk[8];                        // the storage kind decided once per row (wgemm.hip, w_epilogue)
                    if (OK == KIND_F32) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) tk[j] = v[j];
                    } else if (OK == KIND_F16) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) tk[j] = (float)f16_sat(v[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) tk[j] = operand_round(v[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = live ? tk[j] : 0.f;
                        if (mi == 0) t0[j] = t;
                        else { gs[j] = t0[j] + t; gq[j] = fmaf(t, t, t0[j] * t0[j]); }
                    }
                }
                if (live) {
                    const size_t ye = (size_t)yo[mi] + n;                                     // element offset from Yb
                    if (OK == KIND_F32) {
                        float* yp = reinterpret_cast<float*>(Yb) + ye;
                        f32x4 a, b;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
                        *reinterpret_cast<f32x4*>(yp) = a;
                        *reinterpret_cast<f32x4*>(yp + 4) = b;
                    } else if (OK == KIND_F16) {
                        store8_f16(reinterpret_cast<_Float16*>(Yb) + ye, v);
                    } else {
                        store8_operand(reinterpret_cast<h16*>(Yb) + ye, p.ldy / PLANES, v);
                        if constexpr (Y8OK) if (p.Y8) {
                            float amax = 0.f;
#pragma unroll
                            for (int j = 0; j < 8; ++j) { y8v[mi][Q][j] = (float)(h16)v[j]; amax = fmaxf(amax, fabsf(y8v[mi][Q][j])); }
                            y8a[mi] = Q == 0 ? amax : fmaxf(y8a[mi], amax);
                        }
                    }
                }
            }
#if MUDG_PLANES == 1
            if constexpr (Y8OK && Q == 1) if (p.Y8 && cols) {
                // the 32-column block of a row = this lane's pieces Q = 0, 1 and those of lane ^ 32: one E8M0 scale, 2 x 8 e4m3 bytes per lane
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const bool live = 32 * mi < mrem;
                    const float mine = live ? y8a[mi] : 0.f;
                    const float amax = fmaxf(mine, __shfl_xor(mine, 32, 64));
                    if (!live) continue;
                    const int E = mx_block_exponent(amax);
                    const float inv = __uint_as_float((unsigned)(127 - E) << 23);
                    const int64_t m = em0 + row0 + 32 * mi;
                    unsigned char* y8 = reinterpret_cast<unsigned char*>(p.Y8) + m * p.ldy8 + n - 16;
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        u32x2 w8;
                        w8[0] = mx_pack4_e4m3(y8v[mi][qq][0], y8v[mi][qq][1], y8v[mi][qq][2], y8v[mi][qq][3], inv);
                        w8[1] = mx_pack4_e4m3(y8v[mi][qq][4], y8v[mi][qq][5], y8v[mi][qq][6], y8v[mi][qq][7], inv);
                        *reinterpret_cast<u32x2*>(y8 + 16 * qq) = w8;
                    }
                    if (hi == 0) reinterpret_cast<unsigned char*>(p.S8)[m * p.lds8 + (n >> 5)] = (unsigned char)(E + 127);
                }
            }
#endif
            if constexpr (!GEGLU && RS) if (p.stats) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { gs[j] = half_wave_sum(gs[j]); gq[j] = half_wave_sum(gq[j]); }
                if (l31 == 31) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        red[(wm * 128 + co + j) * 2] = gs[j];
                        red[(wm * 128 + co + j) * 2 + 1] = gq[j];
                    }
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        piece(I0{}, I0{});
        piece(I0{}, I1{});
        if constexpr (!GEGLU) {
            piece(I1{}, I0{});
            piece(I1{}, I1{});
        }
    };

    // ---------------------------------------------------------------- the tile loop
    int par = 0;
    setup(t_next);
    issue(0, 0);
    float bnext = column_constant();
    for (;;) {
        const int em0 = m0, en0 = n0, ez = zb, edy = dy0, edx = dx0;
        wave_live = (n0 + wn * 64 < p.N) && (m0 + wm * 64 < p.M);
        seed();
        if (tid < 128) sbias[par * 128 + tid] = bnext;
        if constexpr (ONEBUF) {
            for (int kt = 0; kt < nk; ++kt) {
                stage_barrier();                 // K-tile kt has landed (and the column constants are visible)
                multiply(0);
                __syncthreads();                 // every wave is done reading the stage
                if (kt + 1 < nk) issue(kt + 1, 0);
            }
        } else {
            stage_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
                multiply(cur);
                stage_barrier();                 // K-tile kt + 1 has landed; every wave is done with K-tile kt
            }
        }
        // every stage is free: the next tile's first K-tile and column constants travel under this tile's epilogue
        t_next += t_step;
        const bool more = t_next < t_end;
        if (more) {
            setup(t_next);
            issue(0, 0);
            bnext = column_constant();
        }
        epilogue(em0, en0, ez, edy, edx, par);
        if constexpr (!GEGLU && RS) if (p.stats) {
            __syncthreads();
            const float* red = sred + par * 512;
            const int ch = tid & 127, which = tid >> 7;
            const float t = red[ch * 2 + which] + red[(128 + ch) * 2 + which];
            if (en0 + ch < p.N) p.stats[((int64_t)(em0 >> 7) * Nout + en0 + ch) * 2 + which] = t;
        }
        if (!more) break;
        par ^= 1;
    }
}

int cu_count() {
    static int cus[MAX_DEVICES] = {};
    const int dev = mudg_current_device();
    if (dev < 0) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

// One workgroup per residency slot (WGS per CU), or one per tile when there are fewer tiles than slots.
template <int MODE, bool GEGLU, int WGS, bool RS>
int launch_p(const MudgGemmDesc& d, int vflags, hipStream_t s) {
    static bool attr_done[MAX_DEVICES] = {};
    const int dev = mudg_current_device();
    if (dev < 0) MUDG_FAIL(MUDG_ELAUNCH, "gemm: no current device");
    constexpr int smem = pk_smem(WGS);
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pgemm_kernel<MODE, GEGLU, WGS, RS>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev] = true;
    }
    const int ntm = (d.M + 127) / 128, ntn = (d.N + 127) / 128;
    const int64_t total = (int64_t)ntm * ntn * d.batch;
    if (total >= ((int64_t)1 << 31)) MUDG_FAIL(MUDG_EINVAL, "gemm: %lld tiles", (long long)total);
    const int64_t slots = (int64_t)cu_count() * (FUSEDP ? 2 : WGS);
    const int grid = (int)(total < slots ? total : slots);
    const float* phi = GEGLU ? mudg_phi_table() : nullptr;
    hipLaunchKernelGGL((pgemm_kernel<MODE, GEGLU, WGS, RS>), dim3(grid), dim3(256), smem, s, d, vflags, phi, ntm, ntn);
    return mudg_check_launch("mudg_gemm");
}

template <int WGS, bool RS>
int by_problem(const MudgGemmDesc& d, int vflags, hipStream_t s) {
    if (d.mode == 0) return d.geglu ? launch_p<0, true, WGS, false>(d, vflags, s) : launch_p<0, false, WGS, RS>(d, vflags, s);
    return d.mode == 1 ? launch_p<1, false, WGS, RS>(d, vflags, s) : launch_p<2, false, WGS, RS>(d, vflags, s);
}

}  // namespace

int mudg_pgemm_launch(const MudgGemmDesc& d, int vflags, int wgs, hipStream_t s) {
    if (d.geglu && (d.mode != 0 || d.R || d.gbias || d.stats || d.act)) MUDG_FAIL(MUDG_EINVAL, "gemm: the persistent GEGLU kernel is bias-only");
    if (d.R && (d.alpha != 1.f || d.act)) MUDG_FAIL(MUDG_EINVAL, "gemm: the persistent kernel seeds the accumulators with the residual (alpha 1, no activation)");
    const bool rs = d.R || d.stats;
    if (FUSEDP || d.Y8 || rs) wgs = wgs > 2 ? (rs && !FUSEDP && !d.Y8 ? 3 : 2) : 2;       // 64-KiB stage / the fused fp8 copy / seeds and partials
    if (wgs >= 4) return by_problem<4, false>(d, vflags, s);
    if (wgs == 3) return by_problem<3, true>(d, vflags, s);
    return rs ? by_problem<2, true>(d, vflags, s) : by_problem<2, false>(d, vflags, s);
}
#else
int mudg_pgemm_launch(const MudgGemmDesc&, int, int, hipStream_t) { MUDG_FAIL(MUDG_EINVAL, "gemm: no persistent kernel in this build"); }
#endif
