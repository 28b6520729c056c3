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
#include "gemm_shared.h"
#include <cstdlib>
#include <cmath>
#include <type_traits>

bool mudg_gemm_fast_ok(const MudgGemmDesc& d);

namespace {


// Tile geometry.  Every variant is waves laid out WM (rows) x 2 (columns), a wave owning 64 rows x 32 NI columns as
// 2 x NI v_mfma_f32_32x32x16 tiles:
//   G128  (4 waves, 2 x 2, NI = 2): 128 x 128 — the high-occupancy tile (2 or 4 workgroups per CU); serves everything.
//   G320  (8 waves, 4 x 2, NI = 5): 256 x 320 — every channel count of the UNet is a multiple of 320, so no column of the tile
//         is wasted; 144 KiB of LDS (two K-tile stages), one workgroup per CU, two waves per SIMD.  Per FLOP it pulls 2.2x
//         fewer bytes through L2 -> LDS than the 128 x 128 tile (142 against 64 FLOP per staged byte: at 1 PFLOP/s the small
//         tile asks the L2s for 15.6 TB/s, close to half of what they deliver and more than the LDS-DMA path of a CU sustains)
//         and a wave issues 10 MFMAs per 7 fragment reads instead of 4 per 4.
//   G256  (8 waves, 4 x 2, NI = 4): 256 x 256 — the same for GEGLU problems, whose [32 value | 32 gate] row blocks must pair up
//         inside one wave (NI even).
template <int NWAVES_, int NI_>
struct Geo {
    static constexpr int NWAVES = NWAVES_, NI = NI_, MI = 2, WN = 2, WM = NWAVES_ / 2;
    static constexpr int NTH = NWAVES * 64;
    static constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    static constexpr int XI = BM / NWAVES / 8, WI = BN / NWAVES / 8;       // 1-KiB DMA instructions per wave per operand tile
    static constexpr int TILE_X = BM * LDSLD, TILE_W = BN * LDSLD;         // elements per operand per stage
    static constexpr int STGLD = BN + 4;                                   // fp32 per staging row (the 128-wide kernels)
    static constexpr bool WIDE = NWAVES > 4;
    // Which 32-column blocks of the tile a wave column wn owns, as rows of the W tile: block(ni, wn) = WSTRIDE wn + nioff(ni).
    // The 128 x 128 tile gives a wave NI consecutive blocks.  The wide tiles interleave them, so that the blocks 2 q and 2 q + 1
    // of both wave columns together are the 128 consecutive output columns [128 q, 128 q + 128) — what one epilogue pass stages:
    // NI = 5: block = 2 ni + wn;  NI = 4: block = 4 (ni / 2) + 2 wn + (ni & 1) (a GEGLU [value | gate] pair stays in one wave).
    static constexpr int WSTRIDE = !WIDE ? NI : (NI & 1 ? 1 : 2);
    static constexpr int nioff(int ni) { return !WIDE ? ni : (NI & 1 ? 2 * ni : 4 * (ni >> 1) + (ni & 1)); }
};
using G128 = Geo<4, 2>;
using G320 = Geo<8, 5>;
using G256 = Geo<8, 4>;

// bf16x3 build, descriptor loader (FUSED): ONE K-tile stage holds both pieces of both operands — x0, x1, w0, w1, 4 x 16 KiB —
// and every fragment read feeds the three kept products x1 w0 + x0 w1 + x0 w0: each piece is fetched once per K-tile (4 tile
// fetches instead of the 6 of three whole passes over K) and 12 MFMAs follow 8 fragment reads instead of 4 following 4.
// Single K-buffer + two-pass epilogue = 64 KiB -> 2 workgroups per CU.
// Stages: the single-buffer variant (SB, short K: one K-tile stage, the epilogue in 64-row passes -> 34 KiB, 4 workgroups per
// CU) and the fused-piece variant hold one stage, everything else two.
template <typename G> constexpr int stage_elems(bool fast) { return (fused_planes(fast) ? PLANES : 1) * (G::TILE_X + G::TILE_W); }
template <typename G> constexpr int epi_rows(bool fast, bool sb) { return (G::WIDE || sb || fused_planes(fast)) ? 64 : G::BM; }
constexpr int WSTG = 132;                        // wide tiles: fp32 per staging row of a 128-column pass (128 + 4 pad)
template <typename G> constexpr int smem_main(bool fast, bool sb) {
    const bool one = sb || fused_planes(fast);
    const int loop = (one ? 1 : 2) * stage_elems<G>(fast) * 2 + ((one && !G::WIDE) ? 1024 * (fused_planes(fast) ? PLANES : 1) : 0);   // + XSHARE's halo pieces
    const int stg = G::WIDE ? G::BM * WSTG * 4 + G::BN * 4 : epi_rows<G>(fast, sb) * G::STGLD * 4 + G::BN * 4;
    return loop > stg ? loop : stg;
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
template <typename G, int MODE, bool FAST, bool SB>
__global__ __launch_bounds__(G::NTH, G::WIDE ? 2 : ((SB && !fused_planes(FAST)) ? 4 : 2)) void gemm_kernel(const MudgGemmDesc p, const int vflags,
                                                                const h16* __restrict__ zpage, const float* __restrict__ phi) {
    static_assert(!G::WIDE || (FAST && !SB && PLANES == 1), "the wide tiles exist for the descriptor loader of the 16-bit builds");
    constexpr int BM = G::BM, BN = G::BN, NI = G::NI, MI = G::MI, NTH = G::NTH, STGLD = G::STGLD;
    constexpr int XI = G::XI, WI = G::WI, TILE_X = G::TILE_X, TILE_W = G::TILE_W;
    constexpr int XROWS = 8 * XI, WROWS = 8 * WI;       // operand-tile rows a wave stages
    constexpr bool FUSED = fused_planes(FAST);          // all pieces of a K-tile staged at once
    constexpr bool ONEBUF = SB || FUSED;                // one K-tile stage
    constexpr int XT = FUSED ? PLANES : 1;              // tiles per operand per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* Xs = reinterpret_cast<h16*>(smem);
    h16* Ws = Xs + (ONEBUF ? 1 : 2) * XT * TILE_X;
    float* phis = reinterpret_cast<float*>(smem + smem_main<G>(FAST, SB));     // beyond every other LDS use
    if (p.geglu && phi) {            // visible after the K loop's barriers
        for (int t4 = threadIdx.x * 4; t4 < PHI_N; t4 += NTH * 4)
            *reinterpret_cast<f32x4*>(&phis[t4]) = *reinterpret_cast<const f32x4*>(&phi[t4]);
        if (threadIdx.x == 0) phis[PHI_N] = phi[PHI_N];
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % G::WM, wn = wave / G::WM;
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
    // TMAP (temporal convs of 16-frame clips, slab-major K; host flag VF_TM): tile tm is NOT 128 consecutive rows but 8 pixels x 16
    // frames of one clip — logical row r of the tile is frame r / 8, pixel r % 8 — so that the three temporal taps of an output row
    // are rows of the SAME tile, 8 up / down.  The one-stage kernels then stage the activation tile once per 64-channel slab for all
    // three taps (TSHARE: 48 -> 16 activation pieces per slab; frames -1 and 16 are the zero rows of a spare piece); every kernel
    // maps rows this way when the flag is set (loader, residual, result, GroupNorm partial blocks), so that the partial sums a
    // consumer folds do not depend on which kernel ran.
    constexpr bool TMAP_OK = MODE == 2 && FAST && !G::WIDE && PLANES <= 2 && BM == 128;
    const bool tmap = TMAP_OK && (vflags & VF_TM);
    auto phys = [&](int64_t m) -> int64_t {            // logical row -> row of X / Y / R
        if constexpr (!TMAP_OK) return m;
        if (!tmap) return m;
        const int64_t t_ = m >> 7; const int r = (int)(m & 127);
        const int per = p.HW >> 3;                     // tiles per clip
        const int64_t b = t_ / per; const int pb = (int)(t_ - b * per);
        return (b * p.T + (r >> 3)) * p.HW + pb * 8 + (r & 7);
    };
    const int64_t bz = blockIdx.z;
    // Sub-pixel form of "nearest-2x upsample, then 3x3 conv" (MudgGemmDesc.subpixel): batch entry z = 2 py + px computes the
    // output pixels (2 oy + py, 2 ox + px) from the 2x2 low-resolution neighbourhood that their nine taps collapse onto.
    const bool sub = MODE == 1 && p.subpixel;
    const int ntaps = MODE == 2 ? 3 : (sub ? 4 : 9);
    const int dy0 = sub ? (int)(blockIdx.z >> 1) : 0, dx0 = sub ? (int)(blockIdx.z & 1) : 0;
    const h16* X = reinterpret_cast<const h16*>(p.X) + bz * p.sX;
    const h16* X2 = p.X2 ? reinterpret_cast<const h16*>(p.X2) + bz * p.sX : nullptr;
    const h16* W = reinterpret_cast<const h16*>(p.W) + bz * p.sW;

    // DMA geometry: wave w stages rows [XROWS w, XROWS (w + 1)) of the X tile and [WROWS w, WROWS (w + 1)) of the W tile, one
    // 1-KiB instruction per 8 rows; in instruction i, lane l lands in row 8 i + (l >> 3) of the wave's slice, slot l & 7, and
    // therefore fetches the logical chunk (l & 7) ^ ((row >> 1) & 7) of that row.
    const int rsub = lane >> 3, slot = lane & 7;
    int rm[XI], ra[XI], rb[XI], rc[XI], ch[XI], chw[WI];
    bool rv[XI];
#pragma unroll
    for (int i = 0; i < WI; ++i) chw[i] = slot ^ (((WROWS * wave + 8 * i + rsub) >> 1) & 7);
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int rl = XROWS * wave + 8 * i + rsub;
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
            rb[i] = tmap ? (rl >> 3) : (m / p.HW) % p.T;
        }
    }
    const bool tap_uniform = MODE != 0 && (p.Cin & 63) == 0;     // a 64-wide K tile never straddles two taps

    // FAST path state: descriptors based at the block's first source row, invariant lane offsets, tap validity bits.
    __amdgpu_buffer_rsrc_t rX, rX2, rW;
    unsigned vx[XI], vx2[XI], vw[WI], vmask[XI];
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
        if constexpr (TMAP_OK) if (tmap) pix0 = phys(m0);
        const int64_t shift = MODE == 1 ? -(int64_t)(p.pad * p.Win + p.pad) : (MODE == 2 ? -(int64_t)p.HW : 0);
        rX = make_rsrc(X + (pix0 + shift) * p.ldx);
        rX2 = X2 ? make_rsrc(X2 + (pix0 + shift) * p.ldx2) : rX;
        rW = make_rsrc(W + (int64_t)n0 * p.ldw);
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int rl = WROWS * wave + 8 * i + rsub;
            vw[i] = (n0 + rl < p.N) ? (unsigned)rl * (unsigned)p.ldw * 2u + (unsigned)chw[i] * 16u : OOB;
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int rl = XROWS * wave + 8 * i + rsub;
            int rel = rl;
            if constexpr (TMAP_OK) if (tmap) rel = (int)(phys(m0 + rl) - pix0);
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
        }
    }
    // XSHARE (same-size stride-1 3x3 convs, slab-major K, the one-stage kernels): the taps (dy, 0), (dy, 1), (dy, 2) read the same
    // 128 source pixels shifted by one, so the activation tile is staged ONCE per dy — the centre tap's tile plus one halo pixel on
    // either side (a 1-KiB piece behind the W tile: row 7 = the pixel before the tile, row 0 = the pixel after it, rows 1 - 6 zero) — and the dx = 0 /
    // dx = 2 K-steps fetch their weights only and read the fragments one row up / down.  144 -> 51 activation pieces per 64-channel
    // slab; the stage stays 33 KiB (four workgroups per CU).  Rows whose tap leaves the image vertically are zero-filled by the DMA as
    // before; a horizontal neighbour that belongs to the next image row is zeroed when the fragment is read (x == 0 / x == W - 1).
    constexpr bool TSHARE_OK = TMAP_OK && ONEBUF;                  // the one-stage kernels share the staged tile between the three temporal taps
    const bool tshare = TSHARE_OK && tmap;
    constexpr bool XSHARE_OK = MODE == 1 && FAST && ONEBUF && !G::WIDE && PLANES <= 2;      // the one-stage kernels: SB (16-bit builds), fused pieces (bf16x3)
    const bool xshare = XSHARE_OK && (vflags & VF_XS);
    unsigned xedge = 0;      // bits 0-3: x == 0 / x == W - 1 of the lane's two fragment rows; bits 8-10 (halo lanes): dy validity of their pixel
    if constexpr (XSHARE_OK) if (xshare) {
        if (wave == 0 && (rsub == 0 || rsub == 7)) {       // piece row 7: the pixel before the tile; piece row 0: the pixel after it
            const int rl = rsub == 7 ? 0 : BM - 1, dxh = rsub == 7 ? 0 : 2;
            const int m = m0 + rl;
            if (m < p.M) {
                const int hw = p.Hout * p.Wout;
                const int f = m / hw, r = m - f * hw;
                const int oy = r / p.Wout, ox = r - oy * p.Wout;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int iy = oy - 1 + dy, ix = ox - 1 + dxh;
                    if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) xedge |= 256u << dy;
                }
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wm * 64 + mi * 32 + l31;
            const int ox = m % p.Wout;
            if (ox == 0) xedge |= 1u << (2 * mi);
            if (ox == p.Wout - 1) xedge |= 2u << (2 * mi);
        }
    }
    int dx_m = 0;                                                 // XSHARE: dx of the K-step being multiplied

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
            bool xstage = true;
            int xtap = tap_s, sox = so;
            if constexpr (TSHARE_OK) if (tshare) {
                xstage = tap_s == 0;                         // the slab's first K-step stages the centre frame rows for all three taps
                xtap = 1;
                sox = so + p.HW * ld * 2;                    // tap 1 = the rows themselves (the descriptor is based one frame back)
                if (xstage && wave == 0)                     // a piece of zero rows behind the W tile(s): frames -1 and 16
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lptr_t)(Ws + XT * TILE_W + pl * 512), 16, (int)OOB, 0, 0, 0);
            }
            if constexpr (XSHARE_OK) if (xshare) {
                const int dy = tap_s / 3, dxs = tap_s - 3 * dy;
                xstage = dxs == 0;                           // the dy group's first K-step stages the centre tile (+ halo) for all three
                xtap = tap_s + 1;
                sox = so + ld * 2;                           // centre tap: one pixel right of (dy, 0)
                if (xstage && wave == 0) {                   // the halo piece: lanes rsub 0 / 1 fetch the pixel before / after the tile
                    // lanes rsub 7: the pixel before the tile (row 0 at tap (dy, 0)), stored like a row -1 (odd row, swizzle key 7); rsub 0: the
                    // pixel after it (row BM - 1 at tap (dy, 2)), like a row BM (even, key 0); rows 1 - 6 of the piece are zero-filled
                    const unsigned v = ((rsub == 0 || rsub == 7) && ((xedge >> (8 + dy)) & 1u))
                                       ? (unsigned)(rsub == 0 ? BM + 1 : 0) * (unsigned)ld * 2u + (unsigned)(slot ^ (rsub == 7 ? 7 : 0)) * 16u : OOB;
                    lptr_t lh = (lptr_t)(Ws + XT * TILE_W + pl * 512);
                    if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, lh, 16, (int)v, so, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, lh, 16, (int)v, so, 0, 0);
                }
            }
#ifdef GEMM_EXP_NOX      // ablation (wrong results, profiles/r4/ablation_fetch_first_ktile_only.txt): activation tiles fetched for K-tile 0 only
            xstage = xstage && kt == 0;
#endif
            if (xstage)
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                unsigned v = s2 ? vx2[i] : vx[i];
                if (MODE != 0) v = ((vmask[i] >> xtap) & 1u) ? v : OOB;
                lptr_t lx = (lptr_t)(Xs + (buf * XT + pl) * TILE_X + (XROWS * wave + 8 * i) * LDSLD);
                if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, lx, 16, (int)v, sox, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, lx, 16, (int)v, sox, 0, 0);
            }
#ifdef GEMM_EXP_NOW      // ablation (wrong results): weight tiles fetched for K-tile 0 only
            if (kt == 0)
#endif
#pragma unroll
            for (int i = 0; i < WI; ++i) {
                // W rows beyond N are never multiplied (their waves are idle, see wave_live): skip the zero-fill pieces
                if (G::WIDE || n0 + WROWS * wave + 8 * i < p.N || (WROWS * wave + 8 * i) < 64)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(Ws + (buf * XT + pl) * TILE_W + (WROWS * wave + 8 * i) * LDSLD), 16,
                                                             (int)vw[i], sow, 0, 0);
            }
        }
        if (MODE == 0) {
            c_s += BK;
        } else {                                   // select form: the branchy update sent tap_s / c_s to scratch memory
            const int t1 = tap_s + 1, c1 = c_s + BK;
            const bool slab = (MODE == 1 || MODE == 2) && p.korder;
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
            if ((MODE == 1 || MODE == 2) && p.korder) { const int slab = kt / ntaps; tap_u = kt - slab * ntaps; c_u = slab * 64; }
            else { tap_u = k0 / p.Cin; c_u = k0 - tap_u * p.Cin; }
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
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
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xs + buf * TILE_X + (XROWS * wave + 8 * i) * LDSLD), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int k = k0 + chw[i] * 8;
            const int n = n0 + WROWS * wave + 8 * i + rsub;
            const h16* wsrc = (n < p.N && k < p.K) ? W + (int64_t)n * p.ldw + k + wo : zpage;
            __builtin_amdgcn_global_load_lds((gptr_t)wsrc, (lptr_t)(Ws + buf * TILE_W + (WROWS * wave + 8 * i) * LDSLD), 16, 0, 0);
        }
    };

    f32x16 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nkt = FUSED ? nk : nk * NSEG;      // K-tiles over all (x plane, w plane) passes; FUSED: all pieces per K-tile
    const int sw = (l31 >> 1) & 7;       // read-side swizzle: the row bases are multiples of 32, so only lane bits count
    // A wave whose output columns (or rows) all lie beyond N (M) has nothing to multiply: it still stages its share
    // of the operand tiles but leaves its SIMD's MFMA pipe to the other resident workgroups (N = 320: 1/6 of the waves).
    const bool wave_live = (G::WIDE || n0 + wn * (32 * NI) < p.N) && (m0 + wm * 64 < p.M);
    auto multiply = [&](int cur) {
        if (!wave_live) return;
        const h16* xs = Xs + cur * XT * TILE_X + (wm * 64 + l31) * LDSLD;
        const h16* ws = Ws + cur * XT * TILE_W + (wn * (32 * G::WSTRIDE) + l31) * LDSLD;
        if constexpr (XSHARE_OK || TSHARE_OK) {
            // TSHARE: the fragment rows of temporal tap dt are the staged rows 8 (dt - 1) further; frames -1 / 16 are zero rows.
            // XSHARE: fragment rows one up / down from the staged centre tile (d = dx - 1); the tile's first / last row reaches into
            // the halo piece (rows 7 / 0 behind the W tiles, one piece per plane), a row whose neighbour belongs to another image row
            // reads one of that piece's zero rows.  Without XSHARE d = 0 and no edges: the plain fragment rows.
            const int d = MODE == 2 ? (tshare ? 8 * (dx_m - 1) : 0) : (xshare ? dx_m - 1 : 0);
            int xo[MI];                                  // element offset inside a plane's tile (or from the plane's halo piece, bit 19) | swizzle key << 20
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int rr = wm * 64 + mi * 32 + l31 + d;
                const bool edge = MODE == 2 ? (rr < 0 || rr >= BM)
                                            : ((d < 0 && ((xedge >> (2 * mi)) & 1u)) || (d > 0 && ((xedge >> (2 * mi)) & 2u)));
                // every substitute row keeps the bank pattern of the row it stands for (row parity and swizzle key): conflict-free reads
                const int key = ((rr >> 1) & 7) << 20;
                xo[mi] = edge ? ((1 << 19) + (2 + (rr & 1)) * LDSLD) | key
                              : (rr < 0 ? ((1 << 19) + 7 * LDSLD) | key : (rr >= BM ? (1 << 19) | key : (rr * LDSLD) | key));
            }
            auto xfrag = [&](int pl, int mi, int ks) {
                const int o = xo[mi] & 0x7ffff;
                const h16* base = (xo[mi] & (1 << 19)) ? Ws + XT * TILE_W + pl * 512 : Xs + (cur * XT + pl) * TILE_X;
                return *reinterpret_cast<const h16x8*>(base + o + (((ks * 2 + hi) ^ (xo[mi] >> 20)) << 3));
            };
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int off = ((ks * 2 + hi) ^ sw) << 3;
                if constexpr (FUSED) {
                    h16x8 wf[2][NI], xf[2][MI];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) wf[pl][ni] = *reinterpret_cast<const h16x8*>(ws + pl * TILE_W + G::nioff(ni) * 32 * LDSLD + off);
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) xf[pl][mi] = xfrag(pl, mi, ks);
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
                } else {
                    h16x8 wf[NI], xf[MI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const h16x8*>(ws + G::nioff(ni) * 32 * LDSLD + off);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) xf[mi] = xfrag(0, mi, ks);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            acc[ni][mi] = MFMA_32x32x16(wf[ni], xf[mi], acc[ni][mi]);
                }
            }
            dx_m = dx_m == 2 ? 0 : dx_m + 1;
            return;
        }
        if constexpr (FUSED) {
            // x = x0 + x1, w = w0 + w1 (bf16 pieces): x1 w0 + x0 w1 + x0 w0 per fragment pair — the bf16 x bf16 products are
            // exact in the fp32 accumulator; what is dropped (x1 w1) is 2^-18 relative.  Small terms first within a k-step.
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int off = ((ks * 2 + hi) ^ sw) << 3;
                h16x8 wf[2][NI], xf[2][MI];          // [piece][32-row block]
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) wf[pl][ni] = *reinterpret_cast<const h16x8*>(ws + pl * TILE_W + G::nioff(ni) * 32 * LDSLD + off);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) xf[pl][mi] = *reinterpret_cast<const h16x8*>(xs + pl * TILE_X + mi * 32 * LDSLD + off);
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
            return;
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int off = ((ks * 2 + hi) ^ sw) << 3;
            h16x8 wf[NI], xf[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const h16x8*>(ws + G::nioff(ni) * 32 * LDSLD + off);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const h16x8*>(xs + mi * 32 * LDSLD + off);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
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

    // ------------------------------------------------------------------ epilogue of the wide tiles
    // One workgroup per CU: nothing else hides this tile's store latency, so the epilogue is built for memory-level parallelism.
    // A pass stages 128 output columns of ALL 256 rows (every wave writes the two blocks 2 q, 2 q + 1 it owns: see Geo::nioff) as
    // fp32 in LDS; then thread t keeps the 8-channel chunk t & 15 and the rows (t >> 8) 128 + ((t >> 4) & 15) + 16 k, k = 0..7 —
    // for GroupNorm partials exactly the rows, in exactly the order, that a thread of the 128 x 128 kernels sums, so both kernels
    // write bit-identical partials — in two batches of four rows whose residual loads are all in flight before the first is used.
    if constexpr (G::WIDE) {
        float* stg = reinterpret_cast<float*>(smem);
        float* sbias = stg + BM * WSTG;
        bool gbias_rows = p.gbias != nullptr;
        if (p.gbias && !p.geglu) {
            const int mlast = (m0 + BM <= p.M ? m0 + BM : p.M) - 1;
            const int g0 = m0 / p.rows_per_group;
            if (g0 == mlast / p.rows_per_group) {
                gbias_rows = false;
                for (int t = tid; t < BN; t += NTH) {
                    float b = (p.bias && n0 + t < p.N) ? p.bias[n0 + t] : 0.f;
                    if (n0 + t < p.N) b += p.gbias[(int64_t)g0 * p.N + n0 + t];
                    sbias[t] = b;
                }
            }
        }
        if (gbias_rows || !p.gbias || p.geglu) {     // straddling tile: bias + group bias go row by row as one pre-summed constant (see below)
            for (int t = tid; t < BN; t += NTH) sbias[t] = (!(gbias_rows && !p.geglu) && p.bias && n0 + t < p.N) ? p.bias[n0 + t] : 0.f;
        }
        __syncthreads();

        const float alpha = p.alpha;
        const int Nout = p.geglu ? p.N / 2 : p.N;
        const int nout0 = p.geglu ? n0 / 2 : n0;
        const int cc = tid & 15, rr = (tid >> 4) & 15, half = tid >> 8;
        const int hw_o = MODE == 1 ? p.Hout * p.Wout : 1;
        constexpr int NPASSW = (NI + 1) / 2;
        const int RK = p.R ? p.res_fp32 : 3, OK = p.out_fp32;          // storage kinds of the residual (3 = none) and of Y
        const char* Rb = reinterpret_cast<const char*>(p.R) + bz * p.sR * (RK == KIND_F32 ? 4 : 2);
        const int rsz = RK == KIND_F32 ? 4 : 2;
        auto run_pass = [&](auto qtag) __attribute__((always_inline)) {
            constexpr int q = decltype(qtag)::value;
            // ---- accumulators -> staging (fp32, + bias, GEGLU)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int ml = wm * 64 + mi * 32 + l31;
                if (!p.geglu) {
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const int ni = 2 * q + d;
                        if (ni < NI) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int bl = (NI & 1) ? 2 * d + wn : 2 * wn + d;          // block inside the pass's 128 columns
                                const int nl = bl * 32 + 8 * g + 4 * hi;
                                const int nb = q * 128 + nl;                               // column inside the tile (bias index)
                                f32x4 v;
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] = alpha * acc[ni < NI ? ni : 0][mi][4 * g + j] + sbias[nb + j];
                                if (p.act) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[j] = gelu_fast(v[j]);
                                }
                                *reinterpret_cast<f32x4*>(&stg[ml * WSTG + nl]) = v;
                            }
                        }
                    }
                } else if constexpr (NI % 2 == 0) {
                    // the pass's pair of this wave: value block 4 q + 2 wn, gate block 4 q + 2 wn + 1 -> output columns 64 q + 32 wn ..
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nb = q * 128 + wn * 64 + 8 * g + 4 * hi;                 // value columns inside the tile
                        f32x4 v;
                        float gate[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = alpha * acc[2 * q][mi][4 * g + j] + sbias[nb + j];
                            gate[j] = alpha * acc[2 * q + 1][mi][4 * g + j] + sbias[nb + 32 + j];
                        }
                        // (table or polynomial: decided per group, not inside the per-value expression — there the compiler kept a branch
                        // and a serialised LDS round trip per value; wgemm.hip, w_epilogue)
                        if (phi) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] *= gelu_lut(gate[j], phis);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] *= gelu_fast(gate[j]);
                        }
                        *reinterpret_cast<f32x4*>(&stg[ml * WSTG + wn * 32 + 8 * g + 4 * hi]) = v;
                    }
                }
            }
            __syncthreads();
            // ---- staging -> HBM
            const int ncols = p.geglu ? 64 : 128;                                         // output columns of this pass
            const int n = nout0 + q * ncols + cc * 8;
            const int nvalid = (cc * 8 >= ncols || (!p.geglu && q * 128 + cc * 8 >= BN) || n >= Nout) ? 0 : ((Nout - n) < 8 ? (Nout - n) : 8);
            float gs[8], gq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
            const bool wideY = nvalid == 8 && (vflags & VF_Y), wideR = nvalid == 8 && (vflags & VF_R);
#pragma unroll
            for (int kb = 0; kb < 8; kb += 4) {
                float v[4][8];
                u32x4 ra[4], rb[4];
                int64_t mrow[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = half * 128 + rr + 16 * (kb + u);
                    const int m = m0 + row;
                    mrow[u] = m;
                    ok[u] = m < p.M && nvalid > 0;
                    ra[u] = zero16(); rb[u] = zero16();
                    if (RK != 3 && ok[u] && wideR) {             // 16 bytes whatever the kind (fp32: the second half follows)
                        const char* rp = Rb + ((int64_t)m * p.ldr + n) * rsz;
                        ra[u] = ld16(rp);
                        if (RK == KIND_F32) rb[u] = ld16(rp + 16);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = half * 128 + rr + 16 * (kb + u);
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&stg[row * WSTG + cc * 8]);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(&stg[row * WSTG + cc * 8 + 4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[u][j] = a[j]; v[u][4 + j] = b[j]; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!ok[u]) continue;
                    const int64_t m = mrow[u];
                    if (gbias_rows) {
                        const float* gb = p.gbias + (int64_t)(m / p.rows_per_group) * Nout + n;
                        const bool withb = p.bias && !p.geglu;
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (j < nvalid) v[u][j] += withb ? p.bias[n + j] + gb[j] : gb[j];
                    }
                    if (RK != 3) {
                        if (wideR) {
                            if (RK == KIND_F32) {
                                union { u32x4 w; f32x4 f; } ta, tb; ta.w = ra[u]; tb.w = rb[u];
#pragma unroll
                                for (int j = 0; j < 4; ++j) { v[u][j] += ta.f[j]; v[u][4 + j] += tb.f[j]; }
                            } else if (RK == KIND_F16) {
                                union { u32x4 w; f16x8 h; } t; t.w = ra[u];
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[u][j] += (float)t.h[j];
                            } else {
                                const h16x8 t = as_h16x8(ra[u]);
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[u][j] += (float)t[j];
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) if (j < nvalid) {
                                const int64_t ro = bz * p.sR + m * p.ldr + n + j;
                                v[u][j] += RK == KIND_F32 ? reinterpret_cast<const float*>(p.R)[ro]
                                         : (RK == KIND_F16 ? (float)reinterpret_cast<const _Float16*>(p.R)[ro] : (float)reinterpret_cast<const h16*>(p.R)[ro]);
                            }
                        }
                    }
                    if (p.stats) {
                        float t[8];                     // what the store will hold — the storage kind decided once per row (wgemm.hip, w_epilogue)
                        if (OK == KIND_F32) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) t[j] = v[u][j];
                        } else if (OK == KIND_F16) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) t[j] = (float)f16_sat(v[u][j]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) t[j] = (float)(h16)v[u][j];
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float tt = (j < nvalid) ? t[j] : 0.f;
                            gs[j] += tt; gq[j] = fmaf(tt, tt, gq[j]);
                        }
                    }
                    int64_t yoff;
                    if (sub) {
                        const int f = (int)(m / hw_o), r = (int)(m - (int64_t)f * hw_o);
                        const int oy = r / p.Wout, ox = r - oy * p.Wout;
                        yoff = (((int64_t)(f * p.Hout + oy) * 2 + dy0) * (2 * p.Wout) + 2 * ox + dx0) * p.ldy + n;
                    } else {
                        yoff = bz * p.sY + m * p.ldy + n;
                    }
                    if (OK == KIND_F16) {
                        _Float16* yp = reinterpret_cast<_Float16*>(p.Y) + yoff;
                        if (wideY) store8_f16(yp, v[u]);
                        else for (int j = 0; j < nvalid; ++j) yp[j] = f16_sat(v[u][j]);
                    } else if (OK == KIND_F32) {
                        float* yp = reinterpret_cast<float*>(p.Y) + yoff;
                        if (wideY) {
                            f32x4 a, b;
#pragma unroll
                            for (int j = 0; j < 4; ++j) { a[j] = v[u][j]; b[j] = v[u][4 + j]; }
                            *reinterpret_cast<f32x4*>(yp) = a;
                            *reinterpret_cast<f32x4*>(yp + 4) = b;
                        } else for (int j = 0; j < nvalid; ++j) yp[j] = v[u][j];
                    } else {
                        h16* yp = reinterpret_cast<h16*>(p.Y) + yoff;
                        if (wideY) store8_operand(yp, p.ldy, v[u]);
                        else for (int j = 0; j < nvalid; ++j) yp[j] = (h16)v[u][j];
                    }
                }
            }
            __syncthreads();                      // staging is free again
            if (p.stats) {
                // fold the sixteen row classes of each (half, chunk) in a fixed order — the 128 x 128 kernels' fold
                float* red = stg;
#pragma unroll
                for (int j = 0; j < 8; ++j) { red[tid * 17 + j] = gs[j]; red[tid * 17 + 8 + j] = gq[j]; }
                __syncthreads();
                {
                    const int hf = tid >> 8, c2 = (tid >> 4) & 15, j = tid & 15;
                    float t = 0.f;
                    for (int k = 0; k < 16; ++k) t += red[((hf * 16 + k) * 16 + c2) * 17 + j];
                    const int n2 = nout0 + q * 128 + c2 * 8 + (j & 7);
                    const int64_t blk = (int64_t)(m0 / 128) + hf;
                    if (q * 128 + c2 * 8 < BN && n2 < Nout && blk * 128 < p.M) p.stats[(blk * Nout + n2) * 2 + (j >> 3)] = t;
                }
                __syncthreads();
            }
        };
        run_pass(std::integral_constant<int, 0>{});
        if constexpr (NPASSW > 1) run_pass(std::integral_constant<int, 1>{});
        if constexpr (NPASSW > 2) run_pass(std::integral_constant<int, 2>{});
        return;
    }

    // ------------------------------------------------------------------ epilogue
    // The tile goes through an fp32 LDS staging area in passes of PROWS rows (a pass = the wave row(s) wm whose 64 rows it holds),
    // then leaves in coalesced 16-byte row pieces: thread t keeps ONE 8-channel chunk (t % cpr) and walks the rows.
    float* stg = reinterpret_cast<float*>(smem);
    constexpr int PROWS = epi_rows<G>(FAST, SB), NPASS = BM / PROWS;
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
            for (int t = tid; t < BN; t += NTH) {
                float b = (p.bias && n0 + t < p.N) ? p.bias[n0 + t] : 0.f;
                if (n0 + t < p.N) b += p.gbias[(int64_t)g0 * p.N + n0 + t];
                sbias[t] = b;
            }
        }
    }
    // (a straddling tile adds bias + group bias row by row, as ONE pre-summed constant like the staged one: a row's bits must not
    // depend on which of the two paths its tile took — that changes with M, i.e. with how many clips share the launch)
    if (gbias_rows || !p.gbias || p.geglu) {
        for (int t = tid; t < BN; t += NTH) sbias[t] = (!(gbias_rows && !p.geglu) && p.bias && n0 + t < p.N) ? p.bias[n0 + t] : 0.f;
    }
    __syncthreads();

    const float alpha = p.alpha;
    const int NT = p.geglu ? BN / 2 : BN;
    const int Nout = p.geglu ? p.N / 2 : p.N;
    const int nout0 = p.geglu ? n0 / 2 : n0;
    const int cpr = NT / 8;                        // 8-channel chunks per staged row
    // thread -> (chunk, first row, row step): no per-chunk division in the store loop (the short-K GEMMs spent 8-12 VALU
    // instructions per MFMA, most of them there).  NTH / cpr rows per sweep; the threads beyond rstep * cpr (BN = 320 only) idle.
    int cc, r0, rstep;
    if constexpr ((BN & (BN - 1)) == 0) {          // power-of-two tile widths: shifts
        const int cshift = __builtin_ctz(cpr);
        cc = tid & (cpr - 1); r0 = tid >> cshift; rstep = NTH >> cshift;
    } else {
        rstep = NTH / cpr; r0 = tid / cpr; cc = tid - r0 * cpr;
        if (r0 >= rstep) r0 = PROWS;               // idle
    }
    const h16* R = (p.R && p.res_fp32 == KIND_OPERAND) ? reinterpret_cast<const h16*>(p.R) + bz * p.sR : nullptr;
    const float* Rf = (p.R && p.res_fp32 == KIND_F32) ? reinterpret_cast<const float*>(p.R) + bz * p.sR : nullptr;
    const _Float16* Rh = (p.R && p.res_fp32 == KIND_F16) ? reinterpret_cast<const _Float16*>(p.R) + bz * p.sR : nullptr;
    float gs[8], gq[8];            // GroupNorm partials of this thread's 8 output channels (p.stats)
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
    constexpr int PASS_UNROLL = NPASS > 2 ? 1 : NPASS;      // four passes stay a loop (4x the code of the store branches otherwise)
#pragma unroll PASS_UNROLL
    for (int pass = 0; pass < NPASS; ++pass) {
    if (NPASS == 1 || wm == pass) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int ml = (NPASS == 1 ? wm * 64 : 0) + mi * 32 + l31;
        if (!p.geglu) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wn * (32 * NI) + ni * 32 + 8 * g + 4 * hi;
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = alpha * acc[ni][mi][4 * g + j] + sbias[nl + j];
                    if (p.act) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = PLANES > 2 ? gelu_erf_f(v[j]) : gelu_fast(v[j]);
                    }
                    *reinterpret_cast<f32x4*>(&stg[ml * STGLD + nl]) = v;
                }
        } else if constexpr (NI % 2 == 0) {
#pragma unroll
            for (int q = 0; q < NI / 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * (32 * NI) + q * 64 + 8 * g + 4 * hi;      // value columns of pair q; its gates sit 32 further
                f32x4 v;
                float gate[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = alpha * acc[2 * q][mi][4 * g + j] + sbias[nl + j];
                    gate[j] = alpha * acc[2 * q + 1][mi][4 * g + j] + sbias[nl + 32 + j];
                }
                if (PLANES > 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_f(gate[j]);
                } else if (phi) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= gelu_lut(gate[j], phis);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= gelu_fast(gate[j]);
                }
                *reinterpret_cast<f32x4*>(&stg[ml * STGLD + wn * (16 * NI) + q * 32 + 8 * g + 4 * hi]) = v;
            }
        }
    }
    }
    __syncthreads();

    const int n = nout0 + cc * 8;
    const int nvalid = n >= Nout ? 0 : ((Nout - n) < 8 ? (Nout - n) : 8);
    if (sub) {
        // sub-pixel conv: bias only (host-checked), rows scattered to this parity class's pixels of the full-resolution image
        for (int row = r0; row < PROWS; row += rstep) {
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
    for (int row = r0; row < PROWS; row += rstep) {
        const int m = m0 + pass * PROWS + row;
        if (m >= p.M || nvalid == 0) break;
        const int64_t mp = phys(m);                 // TMAP: where logical row m lives in R / Y
        float v[8];
        {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&stg[row * STGLD + cc * 8]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&stg[row * STGLD + cc * 8 + 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
        }
        if (gbias_rows) {
            const float* gb = p.gbias + (int64_t)(m / p.rows_per_group) * Nout + n;
            const bool withb = p.bias && !p.geglu;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < nvalid) v[j] += withb ? p.bias[n + j] + gb[j] : gb[j];
        }
        if (R) {
            const h16* rp = R + mp * p.ldr + n;
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
            const float* rp = Rf + mp * p.ldr + n;
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
            const _Float16* rp = Rh + mp * p.ldr + n;
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
            float t[8];
            if (p.out_fp32 == KIND_F32) {
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = v[j];
            } else if (p.out_fp32 == KIND_F16) {
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = (float)f16_sat(v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = operand_round(v[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float tt = (j < nvalid) ? t[j] : 0.f;
                gs[j] += tt; gq[j] = fmaf(tt, tt, gq[j]);
            }
        }
        const int64_t yoff = bz * p.sY + mp * p.ldy + n;
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
#if MUDG_PLANES == 1
            if constexpr (!SB) if (p.Y8) {      // (the 4-per-CU variant has no registers to spare: Y8 problems run on this one) MX-fp8 copy of what was just stored: a 32-column block = the four adjacent lanes cc & ~3 .. + 3 of this row
                float r8[8], amax = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { r8[j] = (float)(h16)v[j]; amax = fmaxf(amax, fabsf(r8[j])); }
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                const int E = mx_block_exponent(amax);
                const float inv = __uint_as_float((unsigned)(127 - E) << 23);
                u32x2 w8;
                w8[0] = mx_pack4_e4m3(r8[0], r8[1], r8[2], r8[3], inv);
                w8[1] = mx_pack4_e4m3(r8[4], r8[5], r8[6], r8[7], inv);
                *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(p.Y8) + mp * p.ldy8 + n) = w8;
                if ((cc & 3) == 0) reinterpret_cast<unsigned char*>(p.S8)[(int64_t)m * p.lds8 + (n >> 5)] = (unsigned char)(E + 127);
            }
#endif
        }
    }
    // GroupNorm partials are per 128-row block of Y (MudgGemmDesc.stats): flush whenever the passes done so far end one.
    const bool flush = p.stats && (((pass + 1) * PROWS) % 128 == 0 || pass == NPASS - 1);
    if (NPASS > 1 || flush) __syncthreads();
    if (flush) {
        // thread t always handled channel chunk t % cpr: fold the threads of a chunk in a fixed order
        float* red = stg;
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid * 17 + j] = gs[j]; red[tid * 17 + 8 + j] = gq[j]; gs[j] = 0.f; gq[j] = 0.f; }
        __syncthreads();
        const int blk = (m0 + (pass + 1) * PROWS - 1) / 128;           // the 128-row block these rows belong to
        for (int u = tid; u < cpr * 16; u += NTH) {
            const int c2 = u >> 4, j = u & 15;
            float t = 0.f;
            for (int k = c2; k < rstep * cpr; k += cpr) t += red[k * 17 + j];
            const int n2 = nout0 + c2 * 8 + (j & 7);
            if (n2 < Nout && (int64_t)blk * 128 < p.M) p.stats[((int64_t)blk * Nout + n2) * 2 + (j >> 3)] = t;
        }
        if (pass + 1 < NPASS) __syncthreads();
    }
    }
}

// Lazily created per-device state (a zero page, the Phi table, the kernels' LDS opt-in): keyed by the current device so
// that one process may drive several GPUs.
}  // namespace
int mudg_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return -1;
    return dev;
}
namespace {
int current_device() { return mudg_current_device(); }

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
}  // namespace
const float* mudg_phi_table(bool split_ok) {
    static float* tabs[MAX_DEVICES] = {};
    static int mode = -1;
    if (mode < 0) mode = mudg_variant("GELU_LUT", 1);
    // the split-operand builds evaluate erf (bf16x3: to 1.5e-7); bf16x3's 288 x 256 GEGLU tile asks for the table and interpolates it
    // to the same accuracy with a cubic (split_ok, wgemm.hip)
    if (!mode || (PLANES > 1 && !(split_ok && PLANES == 2))) return nullptr;
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
namespace {
const float* phi_table() { return mudg_phi_table(); }

template <typename G, int MODE, bool FAST, bool SB = false>
int launch(const MudgGemmDesc& d, int vflags, hipStream_t s) {
    static bool attr_done[MAX_DEVICES] = {};
    const h16* zp = zero_page();
    if (!zp) MUDG_FAIL(MUDG_ELAUNCH, "gemm: could not allocate the zero page");
    bool& attr_set = attr_done[current_device()];
    constexpr int smem = smem_main<G>(FAST, SB);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<G, MODE, FAST, SB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem + PHI_BYTES);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = ((d.M + G::BM - 1) / G::BM) * ((d.N + G::BN - 1) / G::BN);
    dim3 grid(tiles, 1, d.batch);
    const float* phi = d.geglu ? phi_table() : nullptr;
    hipLaunchKernelGGL((gemm_kernel<G, MODE, FAST, SB>), grid, dim3(G::NTH), smem + (phi ? PHI_BYTES : 0), s, d, vflags, zp, phi);
    return mudg_check_launch("mudg_gemm");
}

// Problems with at least three tiles per CU (eight for the 3x3 convs) go to the single-buffer / 4-workgroups-per-CU
// variant (see gemm_kernel): measured faster at every K (MDM1024 shapes: +3...+20 %); with fewer tiles the
// double-buffered 2-per-CU kernel wins.
// Variant switch GEMM_SB=0 disables it, =2 forces it for every FAST problem.  (bf16x3 build: the fused-piece kernel is the
// only FAST kernel, see fused_planes.)
bool use_single_buffer(const MudgGemmDesc& d) {
    if (fused_planes(true)) return true;
    if (d.Y8) return false;                   // the fused fp8 copy is compiled into the two-per-CU variant only
    static int mode = -1;
    if (mode < 0) mode = mudg_variant("GEMM_SB", 1);
    if (mode == 0) return false;
    if (mode == 2) return true;
    const int64_t tiles = (int64_t)((d.M + G128::BM - 1) / G128::BM) * ((d.N + G128::BN - 1) / G128::BN) * d.batch;
    // (3x3 convs whose dx taps share a staged tile — XSHARE, one-stage kernel only — switch at half the tile count: 18432 x 1280 x
    //  11520 551 -> 530 us, x 23040 1072 -> 1040; 4608 rows stay on the two-stage kernel, 170 against 224 us.  Both kernels add the
    //  same products in the same order: the choice changes no bits.)
    const bool xs = PLANES == 1 && d.mode == 1 && d.korder && !d.subpixel && !d.upsample && d.stride == 1 && d.pad == 1 && d.Hin == d.Hout &&
                    d.Win == d.Wout;
    return tiles >= (d.mode == 1 ? (xs ? 1024 : 2048) : 768);
}

// Variant switch GEMM_PERSIST=0: the non-persistent 128 x 128 kernels (A/B measurements); 2 / 3 / 4: that many persistent
// workgroups per CU for every problem.
int persist_mode() {
    static int mode = -1;
    if (mode < 0) mode = mudg_variant("GEMM_PERSIST", 1);
    return mode;
}
bool use_persistent() { return persist_mode() != 0; }
// Workgroups per CU of the persistent kernel: one K-tile stage at 4 per CU where the one-tile-per-workgroup rule chose the
// single-buffer variant, two stages at 2 per CU otherwise.
// What the persistent kernel (pgemm.hip) takes, and where it is used.
// Takes: whole 16-byte pieces everywhere (Nout % 8 == 0, aligned Y / R rows), a residual that may seed the accumulators (alpha 1),
// a group bias constant over a 128-row tile, bias-only GEGLU, no plain activation; ragged widths, the Perceiver's GELU and odd
// row groups stay on the one-tile-per-workgroup kernels.
// Used (measured per shape on MI355X, tools/exp_tiles.py with MUDG_GEMM_PERSIST = 0 / 1 / 4 / 3 / 2, profiles/r4/tiles_*.txt):
//   * every GEGLU problem: + 13 % / + 10 % / + 5 % at levels 0 / 1 / 2 (its epilogue is the longest of all and has no residual:
//     the direct epilogue + the prefetch across it pay, and the kernel fits 128 registers);
//   * the plain GEMMs of the 1280-wide levels (at most ~ 6 tiles per CU at one clip): + 4 ... + 16 % — no tail of one-tile
//     workgroups, no first-fetch latency per tile;
//   * NOT the long plain / conv problems: there four one-tile workgroups per CU with the staged epilogue are as fast or faster
//     (- 5 ... - 13 % persistent).  Residual seeds and GroupNorm partials cost the persistent kernel its fourth workgroup
//     (150-170 registers), and with its stores disabled the same kernel runs the level-0 shapes 1.5-2 x faster: what bounds
//     these problems is the result stream slowing every fetch of a K loop that has one stage in flight per workgroup
//     (DESIGN §6), which neither persistence nor the direct epilogue changes.
// Variant switch GEMM_PERSIST: 0 = never, 1 = this rule, 2 / 3 / 4 = every eligible problem at that many workgroups per CU.
bool persistent_ok(const MudgGemmDesc& d, int vflags) {
    if (!use_persistent() || PLANES > 2) return false;
    if (d.mode == 2 && d.korder) return false;            // slab-major temporal convs (TMAP): the one-tile kernels' row mapping and K walk
    if (d.act || !(vflags & VF_Y) || (d.R && !(vflags & VF_R))) return false;
    if ((d.geglu ? d.N / 2 : d.N) % 8 != 0) return false;
    if (d.geglu && (d.mode != 0 || d.R || d.gbias || d.stats)) return false;
    if (d.R && d.alpha != 1.f) return false;
    if (d.gbias && d.rows_per_group % 128 != 0) return false;
    if (persist_mode() != 1 || d.geglu) return true;
    // The rule must not look at M: a residual enters the persistent kernel's sum first and the one-tile kernels' last, and a clip's
    // result may not depend on the batch it travels in (tests/test_fullsize_gpu.py).  N, K >= 1280 = the plain GEMMs of the
    // 1280-wide levels, whatever the number of clips.
    return d.mode == 0 && d.N >= 1280 && d.K >= 1280;
}
bool use_single_buffer(const MudgGemmDesc& d);
int persistent_wgs(const MudgGemmDesc& d) {
    const int m = persist_mode();
    if (m >= 2 && m <= 4) return m;
    return use_single_buffer(d) ? 4 : 2;
}

#if MUDG_PLANES == 1
// The wide tiles (G320 / G256, 16-bit builds, descriptor loader): 0 = the 128 x 128 kernels, 5 / 4 = NI of the wide tile.
// Measured per shape on MI355X (tools/exp_tiles.py, profiles/r3/tiles_*.txt): with all 256 CUs holding one wide tile each the
// main loop runs at 1280-1370 TFLOP/s against 980-1070 of the 128 x 128 kernels (long K), but a wide tile is alone on its CU —
// nothing overlaps its epilogue, and a last round that is partly empty costs a whole tile time — so over a whole problem it
// only wins where K is long AND the rounds are full enough: the 3x3 convs with K >= 8000 at N = 320 (+ 6.7 %), parity at
// K = 5760, and it loses on every plain / GEGLU GEMM of the UNet (K <= 5120: - 15 ... - 55 %).  The rule below is that measurement.
// Variant switch GEMM_WIDE=0 disables them, =1 forces them wherever the columns fit (tests run every epilogue that way).
int use_wide(const MudgGemmDesc& d) {
    static int mode = -1;
    if (mode < 0) mode = mudg_variant("GEMM_WIDE", 2);
    if (mode == 0) return 0;
    if (d.Y8) return 0;                       // the fused fp8 copy lives in the 128 x 128 kernels' epilogue
    int ni = 0;
    if (d.geglu) ni = (d.N % 256 == 0) ? 4 : 0;
    else if (d.N % 320 == 0) ni = 5;
    else if (d.N % 256 == 0) ni = 4;
    if (!ni) return 0;
    if (mode == 1) return ni;
    const int64_t tiles = (int64_t)((d.M + 255) / 256) * (d.N / (64 * ni)) * d.batch;
    const int64_t rounds = (tiles + 255) / 256;
    if (d.mode != 1 || d.geglu || d.K < 8000) return 0;
    if (tiles < 4 * 256 || tiles * 10 < rounds * 256 * 9) return 0;        // >= 4 rounds, >= 90 % of the slots of the rounds used
    return ni;
}
#endif

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
        if (d.korder) rel = (int64_t)(d.T - 1) * d.HW + 8;      // TMAP: a tile's rows span all frames of a clip
    }
    const int64_t lim = (int64_t)1 << 31;
    return rel * ld * 2 + 128 + soff + 16 < lim && (int64_t)(255 + (PLANES > 1)) * d.ldw * 2 + (int64_t)d.K * 2 + 144 < lim;
}

// Whether Y / R can be accessed in whole 16-byte pieces (VF_Y / VF_R).
static int access_flags(const MudgGemmDesc& d) {
    int vflags = 0;
    const int ybytes = d.out_fp32 == KIND_F32 ? 4 : 2;
    if (aligned16(d.Y) && ((int64_t)d.ldy * ybytes) % (d.out_fp32 ? 16 : 16 * PLANES) == 0 && ((int64_t)d.sY * ybytes) % 16 == 0) vflags |= VF_Y;
    const int rbytes = d.res_fp32 == KIND_F32 ? 4 : 2;
    if (d.R && aligned16(d.R) && ((int64_t)d.ldr * rbytes) % (d.res_fp32 ? 16 : 16 * PLANES) == 0 && ((int64_t)d.sR * rbytes) % 16 == 0) vflags |= VF_R;
    return vflags;
}

// The defaults mudg_gemm gives a caller's descriptor before any kernel-selection rule reads it: batch < 1 -> 1, no second source ->
// csplit = the whole channel axis, alpha == 0 (a zero-initialised C struct) -> 1.  ONE place, so that every query about "which kernel
// will run this" (mudg_gemm_stats_rows) sees exactly the descriptor mudg_gemm dispatches on.
static void normalise_desc(MudgGemmDesc& d) {
    if (d.batch < 1) d.batch = 1;
    if (!d.X2) d.csplit = d.mode == 0 ? d.K : d.Cin;
    if (d.alpha == 0.f) d.alpha = 1.f;
}

// Height of the row blocks `stats` will be written in for this problem (mudg_hip.h): 288 / 160 where a tile kernel of wgemm.hip runs it, else 128.
extern "C" int mudg_gemm_stats_rows(const MudgGemmDesc* dp) {
    if (!dp) return 128;
    MudgGemmDesc d = *dp;
    if (d.out_fp32 < 0 || d.out_fp32 > 2 || d.res_fp32 < 0 || d.res_fp32 > 2 || d.mode < 0 || d.mode > 2) return 128;
    normalise_desc(d);
    const int rows = mudg_wgemm_rows(d, access_flags(d));
    return rows ? rows : 128;
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
    normalise_desc(d);
    const int cin = d.mode == 0 ? d.K : d.Cin;
    if (d.X2) {
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
    // the group bias is a second bias (added with it, before nothing): an activation or a GEGLU gate between them is not defined
    MUDG_REQUIRE(!(d.gbias && (d.act || d.geglu)), "mudg_gemm: gbias combines with neither act nor geglu");
    if (d.stats) MUDG_REQUIRE(d.batch == 1 && !d.geglu, "mudg_gemm: stats needs batch == 1 and no GEGLU");
    if (d.Y8) {
        MUDG_REQUIRE(PLANES == 1, "mudg_gemm: the fused fp8 copy belongs to the 16-bit builds");
        MUDG_REQUIRE(d.S8 && d.out_fp32 == KIND_OPERAND && !d.geglu && !d.subpixel && d.batch == 1 && (d.N & 31) == 0 && (d.ldy8 & 7) == 0 &&
                     d.ldy8 >= d.N && d.lds8 >= d.N / 32 && (reinterpret_cast<uintptr_t>(d.Y8) & 7u) == 0,
                     "mudg_gemm: Y8 needs S8, an operand-kind Y, N %% 32 == 0, no GEGLU / sub-pixel / batch, ldy8 %% 8 == 0");
    }
    MUDG_REQUIRE(d.out_fp32 >= 0 && d.out_fp32 <= 2 && d.res_fp32 >= 0 && d.res_fp32 <= 2, "mudg_gemm: out_fp32 / res_fp32 are 0 (operand), 1 (fp32) or 2 (fp16)");
    int vflags = access_flags(d);

    if (d.mode == 2 && d.korder) {      // TMAP / TSHARE (see gemm_kernel): slab-major temporal convs are defined for 16-frame clips on the descriptor loader
        MUDG_REQUIRE(PLANES <= 2 && d.T == 16 && (d.HW & 7) == 0 && (d.Cin & 63) == 0 && d.K == 3 * d.Cin && !d.X2 && d.M == (d.M / (d.T * d.HW)) * d.T * d.HW,
                     "mudg_gemm: temporal conv with korder = 1 needs T = 16, HW %% 8 == 0, Cin %% 64 == 0, K = 3 Cin, one source, whole clips");
        // (the group-bias index of the epilogue is the LOGICAL row / rows_per_group: with the tile mapping that is the clip, nothing finer)
        MUDG_REQUIRE(!d.gbias || d.rows_per_group % (d.T * d.HW) == 0, "mudg_gemm: temporal conv with korder = 1 takes a group bias per clip(s) only");
        if (mudg_gemm_fast_ok(d)) vflags |= VF_TM;      // else: the generic loader walks the slab-major K axis over plain 128-row tiles
    }
    {   // XSHARE (see gemm_kernel): same-size stride-1 3x3 convs with the slab-major K order.  Variant switch CONV_XSHARE=0: off.
        static int xs = -1;
        if (xs < 0) xs = mudg_variant("CONV_XSHARE", 1);
        if (xs && d.mode == 1 && d.korder && !d.subpixel && !d.upsample && d.stride == 1 && d.pad == 1 && d.Hin == d.Hout && d.Win == d.Wout &&
            d.K == 9 * d.Cin && (d.Cin & 63) == 0)
            vflags |= VF_XS;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int fam = d.mode == 0 ? MUDG_FAM_GEMM : (d.mode == 1 ? MUDG_FAM_CONV : MUDG_FAM_TCONV);
    const int slot = mudg_prof_begin(fam, s);
    int rc = MUDG_EINVAL;
    auto by_mode = [&](auto g, auto fast, auto sb) {
        using G = decltype(g);
        constexpr bool F = decltype(fast)::value, S = decltype(sb)::value;
        return d.mode == 0 ? launch<G, 0, F, S>(d, vflags, s) : (d.mode == 1 ? launch<G, 1, F, S>(d, vflags, s) : launch<G, 2, F, S>(d, vflags, s));
    };
    if (mudg_wgemm_ok(d, vflags)) {
        rc = mudg_wgemm_launch(d, vflags, s);
    } else if (mudg_gemm_fast_ok(d)) {
#if MUDG_PLANES == 1
        const int wide = use_wide(d);
        if (wide) rc = wide == 5 ? by_mode(G320{}, std::true_type{}, std::false_type{}) : by_mode(G256{}, std::true_type{}, std::false_type{});
        else
#endif
#if MUDG_PLANES <= 2
        if (persistent_ok(d, vflags)) {
            rc = mudg_pgemm_launch(d, vflags, persistent_wgs(d), s);
        } else
#endif
        if (use_single_buffer(d)) rc = by_mode(G128{}, std::true_type{}, std::true_type{});
#if MUDG_PLANES != 2
        else rc = by_mode(G128{}, std::true_type{}, std::false_type{});
#endif
    } else {
        rc = by_mode(G128{}, std::false_type{}, std::false_type{});
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
