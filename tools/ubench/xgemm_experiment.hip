// EXPERIMENT (round 4; not part of the library — it compiled as mudg_amd/csrc/xgemm.hip behind mudg_gemm while it was measured).
// The activation-stationary GEMM for K = 320 (level-0 Linears: GEGLU FF1 2560 x 320, q|k|v 960 x 320, the 320 x 320 projections).
//   * the four waves of a workgroup split a tile's 128 rows; a wave keeps the MFMA B-operand fragments of its 32 rows for the WHOLE
//     K in registers (20 k-steps x 4 = 80 VGPRs + 64 accumulators), loaded once per row tile straight from HBM, and walks all the
//     column tiles of its unit with them: the activations never pass through LDS;
//   * only W streams through LDS: a four-slot ring of 16-KiB K-tiles, three tiles in flight ahead of the multiply, issued by inline
//     asm (so that hipcc's alias-based LDS-DMA bookkeeping does not drain the ring with a vmcnt(0) before every fragment read) and
//     awaited by COUNTED s_waitcnt vmcnt(N), N = the operations of the wave younger than the tile (later tiles' pieces + the stores
//     of the epilogue since) — every wave issues every store, dead lanes aimed past the descriptor, so that N is known;
//   * the direct epilogue of pgemm.hip.
// Results on MI355X (us; one-tile kernels | pgemm | this): FF1 294912 x 2560 x 320 GEGLU 939 | 808 | 806-815; 294912 x 960 x 320
// 342 | 367 | 320; 294912 x 320 x 320 133 | 130 | 164.  Ablations of this kernel on FF1 (tools: MUDG_XDBG bits): everything 806,
// no stores 707, no MFMA 513, no W DMA 747, no fragment loads 629, none of the four 357 (the loop skeleton + GEGLU epilogue alone;
// 126 without the epilogue) — the phases ADD instead of overlapping: with 180-190 registers only two waves share a SIMD, a wave
// runs fetch-wait, 80 MFMAs and a latency-bound VALU epilogue one after the other, and de-phasing the two workgroups of a CU
// changed nothing.  The ring / counted-wait machinery works (bit-identical results) but does not pay without more waves per
// SIMD; kept as the record of the experiment.
#include "../../mudg_amd/csrc/gemm_shared.h"
#include <type_traits>

#if MUDG_PLANES == 1
namespace {

constexpr int WTILE = 128 * LDSLD;                       // elements of one W K-tile (128 rows x 64 k)
constexpr int XK = 320;                                  // the K this kernel is built for: 20 k-steps of 16 = 80 fragment registers
constexpr int XRING = 4;                                 // W K-tile buffers: three tiles in flight ahead of the one being multiplied
constexpr int XCHUNK_MAX = 10;                           // column tiles per unit: their bias lives in LDS (two unit parities)
constexpr int XWGS = 2;                                  // persistent workgroups per CU
constexpr int X_SMEM = XRING * WTILE * 2 + 2 * XCHUNK_MAX * 128 * 4 + PHI_BYTES;

typedef int i32x4 __attribute__((ext_vector_type(4)));

// A buffer descriptor in four SGPRs (raw buffer: stride 0, num_records = bytes, the data-format word of make_buffer_rsrc).
__device__ __forceinline__ i32x4 x_desc(const void* base, unsigned bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// One 1-KiB LDS-DMA piece, issued where hipcc cannot see it: hipcc orders every later ds_read that might alias a pending
// LDS-DMA behind a vmcnt(0) of its own, which would drain a multi-tile ring at every K-step.  The waits for these pieces are the
// counted ones of the K loop (x_wait).  M0 = LDS byte address of the piece; nothing else in this kernel uses M0.  The leading
// s_nop covers a descriptor word that a v_readfirstlane has only just written (VALU -> SGPR -> VMEM: 5 wait states).
__device__ __forceinline__ void x_dma16(unsigned lds_byte, unsigned voff, i32x4 rsrc, int soff) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// s_waitcnt vmcnt(N) for a run-time (wave-uniform) N: the largest multiple of four that does not exceed it.  vmcnt counts loads
// and stores of a wave in issue order on gfx9: "at most N younger operations are still pending" means the awaited one has landed.
__device__ __forceinline__ void x_wait(int n) {
    if (n >= 16) {
        if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else if (n >= 8) {
        if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool GEGLU>
__global__ __launch_bounds__(256, XWGS) void xgemm_kernel(const MudgGemmDesc p, const float* __restrict__ phi, const int ntm, const int ntn, const int chunks, const int dbg_arg) {
#ifdef MUDG_DEBUG_VARIANTS
    const int dbg = dbg_arg;          // ablation switches of tools/ (1: no stores, 2: no MFMA, 4: no W DMA, 8: no fragment loads)
#else
    constexpr int dbg = 0;
#endif
    constexpr int NI = 4, NKT = XK / BK;              // a wave: 32 rows x 128 columns = 4 MFMA tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* Ws = reinterpret_cast<h16*>(smem);
    float* sbias = reinterpret_cast<float*>(smem + XRING * WTILE * 2);
    float* phis = sbias + 2 * XCHUNK_MAX * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    if (GEGLU && phi) {
        for (int t4 = tid * 4; t4 < PHI_N; t4 += 256 * 4)
            *reinterpret_cast<f32x4*>(&phis[t4]) = *reinterpret_cast<const f32x4*>(&phi[t4]);
        if (tid == 0) phis[PHI_N] = phi[PHI_N];
    }
    constexpr int nk = NKT;
    const int units = ntm * chunks;
    const int Nout = GEGLU ? p.N >> 1 : p.N;
    const int OK = p.out_fp32;
    const int osz = OK == KIND_F32 ? 4 : 2;
    const int nstores = (GEGLU ? 4 : 8) * (OK == KIND_F32 ? 2 : 1);          // store instructions of one epilogue, every wave alike

    // W DMA geometry (pgemm.hip): wave w stages rows [32 w, 32 w + 32) of the tile, lane l of piece i lands in row
    // 32 w + 8 i + (l >> 3), slot l & 7, and fetches chunk (l & 7) ^ ((row >> 1) & 7) of that row.
    const int rl0 = 32 * wave + (lane >> 3);
    const unsigned cb0 = (unsigned)(((lane & 7) ^ ((lane >> 4) & 3)) * 16);
    const unsigned wlane = (unsigned)rl0 * (unsigned)p.ldw * 2u + cb0;
    const h16* Wg = reinterpret_cast<const h16*>(p.W);
    const unsigned ws_base = (unsigned)(size_t)(lptr_t)Ws;
    // always four pieces per wave and tile (rows beyond N read as zero through the descriptor): the counted waits rely on it
    auto issue_w = [&](int j, int kt, int slot) {
        const int n0 = j * 128;
        const int wrows = (p.N - n0) < 128 ? (p.N - n0) : 128;
        const i32x4 rW = x_desc(Wg + (int64_t)n0 * p.ldw, (unsigned)wrows * (unsigned)p.ldw * 2u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vw = wlane + (unsigned)(8 * i) * (unsigned)p.ldw * 2u + ((i & 1) ? ((cb0 ^ 64u) - cb0) : 0u);
            x_dma16(ws_base + (unsigned)((slot * WTILE + (32 * wave + 8 * i) * LDSLD) * 2), vw, rW, kt * (BK * 2));
        }
    };

    // fragment rows of W with bits 2 and 3 of the row index swapped: accumulator registers 8 q .. 8 q + 7 of a lane are then 8
    // consecutive output channels (pgemm.hip)
    const int srow = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int sww = (srow >> 1) & 7;
    const int row0 = wave * 32 + l31;                    // the lane's row inside the tile

    // ---- the workgroup's W-tile stream: (unit, column tile, K-tile) in the order they are multiplied.  The issue pointer runs
    // XRING - 1 tiles ahead of the multiply pointer, across column tiles and across units.
    int iu = blockIdx.x, ij = 0, ij1 = 0, ikt = 0;       // issue pointer
    bool imore = iu < units;
    auto unit_range = [&](int u, int& j0, int& j1) {
        const int c = u % chunks;
        j0 = (c * ntn) / chunks; j1 = ((c + 1) * ntn) / chunks;
    };
    if (imore) unit_range(iu, ij, ij1);
    int gi = 0;                                           // tiles issued so far
    auto issue_next = [&]() {
        if (!imore) return;
        const int slot = gi & 3;
        if (!(dbg & 4) || gi < 4) issue_w(ij, ikt, slot); else { asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" ::: "memory"); }
        ++gi;
        if (++ikt == nk) {
            ikt = 0;
            if (++ij == ij1) {
                iu += gridDim.x;
                imore = iu < units;
                if (imore) unit_range(iu, ij, ij1);
            }
        }
    };
    issue_next(); issue_next(); issue_next();
    // The two workgroups of a CU (blockIdx b and b + gridDim / 2 under the round-robin placement) would run their K loops and their
    // epilogues in lock step, both waiting for the matrix pipe and then both for the VALU: start the second half a tile later.
    if ((dbg & 64) == 0 && blockIdx.x >= (gridDim.x >> 1)) __builtin_amdgcn_s_sleep(48);

    int gc = 0, upar = 0;                                 // tiles multiplied so far; unit parity (bias halves)
    bool stored = false;                                  // an epilogue has been issued
    for (int u = blockIdx.x; u < units; u += gridDim.x, upar ^= 1) {
        int j0, j1;
        unit_range(u, j0, j1);
        const int m0 = (u / chunks) * 128;
        const bool row_ok = m0 + row0 < p.M;
        const int xrows = (p.M - m0) < 128 ? (p.M - m0) : 128;

        // ---- the unit's activation fragments: B operand of v_mfma_f32_32x32x16 = row l31, k-chunk hi of every k-step
        u32x4 xf[NKT * 4];
        {
            const __amdgpu_buffer_rsrc_t rX = x_rsrc(reinterpret_cast<const h16*>(p.X) + (int64_t)m0 * p.ldx, (unsigned)xrows * (unsigned)p.ldx * 2u);
            const unsigned vrow = (unsigned)row0 * (unsigned)p.ldx * 2u + (unsigned)hi * 16u;
#pragma unroll
            for (int ks = 0; ks < NKT * 4; ++ks) {
                xf[ks] = zero16();
                if (!(dbg & 8)) xf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rX, (int)vrow, ks * 32, 0);
            }
        }
        // the unit's column constants, in this unit's half of the bias area: the other half may still be read by a wave that is
        // finishing the previous unit; the one before that is behind at least one barrier of the previous unit
        float* sb_unit = sbias + upar * (XCHUNK_MAX * 128);
        for (int t = tid; t < (j1 - j0) * 128; t += 256) {
            const int n = j0 * 128 + t;
            sb_unit[t] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
        }
        const __amdgpu_buffer_rsrc_t rY = x_rsrc(reinterpret_cast<char*>(p.Y) + (int64_t)m0 * p.ldy * osz, (unsigned)xrows * (unsigned)p.ldy * (unsigned)osz);

        for (int j = j0; j < j1; ++j) {
            const int n0 = j * 128;
            f32x16 acc[NI];
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                {
                    const int slot = gc & 3;
                    // Operations of this wave younger than the pieces of tile gc: the tiles issued after it (four pieces each) and —
                    // the tile having been issued three K-steps ago, BEFORE the epilogue that followed that K-step — the stores of the
                    // one epilogue since, when the tile is among the first three of its column tile.
                    x_wait(4 * (gi - gc - 1) + ((kt < XRING - 1 && stored) ? nstores : 0));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (!(dbg & 32)) __builtin_amdgcn_s_barrier();              // every wave's pieces; every wave is done with the previous tile's slot
                    asm volatile("" ::: "memory");              // (no fragment read may move above the barrier)
                    issue_next();                              // ... which the tile three ahead now takes
                    if (!(dbg & 2)) {
                        const h16* ws = Ws + slot * WTILE + srow * LDSLD;
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) {
                            const int offw = ((k4 * 2 + hi) ^ sww) << 3;
                            h16x8 wf[NI];
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const h16x8*>(ws + ni * 32 * LDSLD + offw);
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) acc[ni] = MFMA_32x32x16(wf[ni], as_h16x8(xf[kt * 4 + k4]), acc[ni]);
                        }
                    }
                    ++gc;
                }
            }

            // ---- epilogue: accumulators (+ bias, GEGLU) -> HBM in 16-byte pieces (host-checked: Nout % 8 == 0, aligned rows).
            // EVERY wave issues EVERY store — lanes without a row or column aim beyond the descriptor and are dropped — so that
            // the counted waits above know how many operations are younger than a tile.
            if (!(dbg & 16)) {
                const float alpha = p.alpha;
                const float* sb = sb_unit + (j - j0) * 128;
                const int nout0 = GEGLU ? n0 >> 1 : n0;
                // one piece = this lane's 8 consecutive output channels: NIX = the 32-column block, Q = the half of it.  GEGLU: the
                // blocks alternate [value | gate], the pair (2 P, 2 P + 1) makes output columns 32 P .. 32 P + 31 of the tile's 64.
                auto piece = [&](auto nitag, auto qtag) __attribute__((always_inline)) {
                    constexpr int NIX = decltype(nitag)::value, Q = decltype(qtag)::value;
                    const int cw = NIX * 32 + Q * 16 + hi * 8;
                    const int co = GEGLU ? (NIX >> 1) * 32 + Q * 16 + hi * 8 : cw;
                    const int n = nout0 + co;
                    float v[8];
                    {
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(&sb[cw]), b1 = *reinterpret_cast<const f32x4*>(&sb[cw + 4]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) { v[t] = alpha * acc[NIX][8 * Q + t] + b0[t]; v[4 + t] = alpha * acc[NIX][8 * Q + 4 + t] + b1[t]; }
                    }
                    if constexpr (GEGLU) {
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&sb[cw + 32]), g1 = *reinterpret_cast<const f32x4*>(&sb[cw + 36]);
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const float gate = alpha * acc[NIX + 1][8 * Q + t] + (t < 4 ? g0[t & 3] : g1[t & 3]);
                            v[t] *= phi ? gelu_lut(gate, phis) : gelu_fast(gate);
                        }
                    }
                    const unsigned yb = (row_ok && n < Nout && !(dbg & 1)) ? ((unsigned)row0 * (unsigned)p.ldy + (unsigned)n) * (unsigned)osz : OOB;
                    if (OK == KIND_F32) {
                        union { u32x4 w; f32x4 f; } a, b;
#pragma unroll
                        for (int t = 0; t < 4; ++t) { a.f[t] = v[t]; b.f[t] = v[4 + t]; }
                        __builtin_amdgcn_raw_buffer_store_b128(a.w, rY, (int)yb, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(b.w, rY, (int)yb, 16, 0);
                    } else if (OK == KIND_F16) {
                        union { u32x4 w; f16x8 h; } o;
#pragma unroll
                        for (int t = 0; t < 8; ++t) o.h[t] = f16_sat(v[t]);
                        __builtin_amdgcn_raw_buffer_store_b128(o.w, rY, (int)yb, 0, 0);
                    } else {
                        h16x8 o;
#pragma unroll
                        for (int t = 0; t < 8; ++t) o[t] = (h16)v[t];
                        __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(o), rY, (int)yb, 0, 0);
                    }
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>;
                using I3 = std::integral_constant<int, 3>;
                piece(I0{}, I0{});
                piece(I0{}, I1{});
                if constexpr (!GEGLU) { piece(I1{}, I0{}); piece(I1{}, I1{}); }
                piece(I2{}, I0{});
                piece(I2{}, I1{});
                if constexpr (!GEGLU) { piece(I3{}, I0{}); piece(I3{}, I1{}); }
                stored = true;
            }
        }
    }
}

int x_cu_count() {
    static int cus[MAX_DEVICES] = {};
    const int dev = mudg_current_device();
    if (dev < 0) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

template <bool GEGLU>
int launch_x(const MudgGemmDesc& d, hipStream_t s) {
    static bool attr_done[MAX_DEVICES] = {};
    const int dev = mudg_current_device();
    if (dev < 0) MUDG_FAIL(MUDG_ELAUNCH, "gemm: no current device");
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xgemm_kernel<GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, X_SMEM);
        if (e != hipSuccess) MUDG_FAIL(MUDG_ELAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done[dev] = true;
    }
    const int ntm = (d.M + 127) / 128, ntn = (d.N + 127) / 128;
    const int slots = XWGS * x_cu_count();
    // Units = row tiles x chunks of column tiles.  One chunk reads the activations once; more chunks even out the last round of
    // units over the persistent workgroups.  Take the fewest chunks whose last round wastes <= 4 % (or one column tile per unit).
    int chunks = 1;
    while ((ntn + chunks - 1) / chunks > XCHUNK_MAX) ++chunks;
    for (; chunks < ntn; ++chunks) {
        const int64_t units = (int64_t)ntm * chunks;
        const int64_t rounds = (units + slots - 1) / slots;
        if (units >= slots && rounds * slots * 100 <= units * 104) break;
    }
    const int64_t units = (int64_t)ntm * chunks;
    const int grid = (int)(units < slots ? units : slots);
    const float* phi = GEGLU ? mudg_phi_table() : nullptr;
    hipLaunchKernelGGL((xgemm_kernel<GEGLU>), dim3(grid), dim3(256), X_SMEM, s, d, phi, ntm, ntn, chunks, mudg_variant("XDBG", 0));
    return mudg_check_launch("mudg_gemm");
}

}  // namespace

bool mudg_xgemm_ok(const MudgGemmDesc& d, int vflags) {
    if (d.mode != 0 || d.batch != 1 || d.X2 || d.R || d.gbias || d.stats || d.act || d.Y8) return false;
    if (d.K != XK || !(vflags & VF_Y)) return false;
    if ((d.geglu ? d.N / 2 : d.N) % 8 != 0) return false;
    if ((int64_t)128 * d.ldx * 2 >= ((int64_t)1 << 31) || (int64_t)128 * d.ldw * 2 + d.K * 2 >= ((int64_t)1 << 31)) return false;
    if ((int64_t)128 * d.ldy * 4 >= ((int64_t)1 << 31)) return false;
    return (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128) >= 2048;        // short problems: the one-tile kernels
}

int mudg_xgemm_launch(const MudgGemmDesc& d, hipStream_t s) {
    return d.geglu ? launch_x<true>(d, s) : launch_x<false>(d, s);
}
#else
bool mudg_xgemm_ok(const MudgGemmDesc&, int) { return false; }
int mudg_xgemm_launch(const MudgGemmDesc&, hipStream_t) { MUDG_FAIL(MUDG_EINVAL, "gemm: no activation-stationary kernel in this build"); }
#endif
