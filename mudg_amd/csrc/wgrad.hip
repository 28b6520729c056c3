// wgrad.hip — weight gradients of linear / 3x3 / temporal conv layers as a contraction over the ROWS of two row-major operand
// matrices, with no transposed copies (training step; 16-bit operand builds):
//
//     out[m][tap * C + c] = sum_p  A[p][m] * B[src(p, tap)][c]          A = dY rows [P][M],  B = layer-input rows [*][C]
//
// src(p, tap) is the input pixel tap `tap` of output position p read (mudg_transpose_gather's modes: 0 identity, 1 the 3x3 tap
// (dy, dx) of a conv with stride / padding, 2 the temporal tap dt) or "none" (zero).  Both operands are consumed in the layout the
// forward pass already has them in: a K-step stages 64 positions x 128 channels of each by LDS-DMA (buffer_load ... lds, 1 KiB per
// instruction, two stages: the next step's fetch in flight under the current step's MFMAs; a missing tap or a row past the slice
// is an out-of-range offset the hardware zero-fills), and the MFMA fragments — 8 consecutive POSITIONS of one channel per lane —
// come out of the position-major images through gfx950's transposing LDS read ds_read_b64_tr_b16 (a 16-lane group reads a
// 4-position x 16-channel block and each lane receives one channel's 4 positions).  The DMA lane order keeps the global reads
// coalesced and swizzles the image so that every transposing read is bank-conflict free (see tr_fragment).
// The contraction over all positions is cut into `slices` ranges that run as blockIdx.y; each writes its own fp32 slab
// out[slice][M][taps C] and mudg_group_colsum adds the slabs in a fixed order (bit-reproducible, no atomics).
// A 128-column tile of the output is two 64-column halves, each inside one tap (C % 64 == 0), so a tile may straddle taps.
#include "common.h"

#if MUDG_PLANES == 1
namespace {

constexpr int WK = 64;       // positions per K-step
constexpr int WT = 128;      // tile: 128 (M) x 128 (taps C)

// Where output position `pos` sits — (frame base, oy, ox) of the conv grid, or (pixel, t) of the clip — kept per staged row and
// advanced by the 64 positions of a K-step with carries instead of divisions.
struct Walker {
    int a, b;            // mode 1: ox, oy;  mode 2: pixel inside the frame, t
    int64_t fbase;       // mode 1: first input row of the frame
    __device__ __forceinline__ void init(const MudgWgradDesc& g, int64_t pos) {
        a = b = 0; fbase = 0;
        if (g.mode == 1) {
            const int hw = g.Hout * g.Wout;
            const int64_t f = pos / hw;
            const int r = (int)(pos - f * hw);
            b = r / g.Wout; a = r - b * g.Wout;
            fbase = f * g.Hin * g.Win;
        } else if (g.mode == 2) {
            const int64_t q = pos / g.HW;
            a = (int)(pos - q * g.HW);
            b = (int)(q % g.T);
        }
    }
    __device__ __forceinline__ void advance(const MudgWgradDesc& g) {
        if (g.mode == 1) {
            a += 64;
            while (a >= g.Wout) { a -= g.Wout; if (++b == g.Hout) { b = 0; fbase += (int64_t)g.Hin * g.Win; } }
        } else if (g.mode == 2) {
            a += 64;
            while (a >= g.HW) { a -= g.HW; if (++b == g.T) b = 0; }
        }
    }
    // the input row tap `tap` of this position reads, or -1
    __device__ __forceinline__ int64_t src(const MudgWgradDesc& g, int64_t pos, int dy, int dx, int tap) const {
        if (g.mode == 0) return pos;
        if (g.mode == 1) {
            const int iy = b * g.stride - g.pad + dy, ix = a * g.stride - g.pad + dx;
            if (iy < 0 || iy >= g.Hin || ix < 0 || ix >= g.Win) return -1;
            return fbase + (int64_t)iy * g.Win + ix;
        }
        const int t = b + tap - 1;
        if (t < 0 || t >= g.T) return -1;
        return pos + (int64_t)(tap - 1) * g.HW;
    }
};

typedef __attribute__((address_space(3))) void* lptr_t;
// The transposing reads are issued as inline assembly: behind the builtin the compiler cannot tell them from the LDS-DMA writes
// of the OTHER stage and drains vmcnt before every group of reads, which serialises fetch and multiply.  In assembly the only
// waits are the ones written here: lgkmcnt before a fragment is used (tr_wait ties the fragment registers to the wait), vmcnt(0)
// + barrier once per K-step (__syncthreads).
struct Frag { u32x2 lo, up; };
__device__ __forceinline__ void tr_issue(Frag& f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(f.lo), "=&v"(f.up) : "v"(addr) : "memory");
}
template <int N>
__device__ __forceinline__ void tr_wait(Frag& a0, Frag& a1, Frag& b0, Frag& b1) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a0.lo), "+v"(a0.up), "+v"(a1.lo), "+v"(a1.up), "+v"(b0.lo), "+v"(b0.up), "+v"(b1.lo), "+v"(b1.up) : "n"(N) : "memory");
}
__device__ __forceinline__ h16x8 frag_value(const Frag& f) {
    const u32x4 v = {f.lo[0], f.lo[1], f.up[0], f.up[1]};
    return as_h16x8(v);
}
// LDS image of a K-step: 16 pieces of 1 KiB, piece t = positions 4 t .. 4 t + 3 of all 128 channels, written by ONE LDS-DMA
// instruction (lane i delivers 16 bytes to piece + 16 i).  Lane i = 16 r + s fetches position 4 t + r and the 8-channel chunk
// c(r, s) = 2 ((s >> 1) ^ 2 r) + (s & 1): sixteen consecutive lanes read one 256-byte row segment (each lane quad 64 contiguous
// bytes — what the memory pipeline coalesces), in an order that XORs the 32-byte chunk-pair index with 2 r.  A transposing read
// of a 4-position x 16-channel block (one chunk pair, rows r = 0..3) then finds its four 32-byte row segments at four different
// pair slots, and the two blocks a 32-lane half reads together cover all eight: every bank exactly once.
// Byte offset, inside an operand image, of the first of the two reads that give lane `lane` the 8-position fragment of channel
// cbase + (lane & 31) at positions kbase + 8 (lane >> 5) + 0..7 (the second read is the next piece: + 1024 bytes):
__device__ __forceinline__ unsigned tr_offset(int kbase, int cbase, int lane) {
    const int g = lane >> 4, q = lane & 15;
    const int piece = (kbase >> 2) + 2 * (g >> 1);
    const int pair = (cbase >> 4) + (g & 1), r = q >> 2;
    return (unsigned)(2 * (piece * 512 + r * 128 + (((pair ^ (2 * r)) * 2 + ((q & 3) >> 1)) * 8) + (q & 1) * 4));
}

constexpr unsigned OOB = 0x80000000u;            // a voffset at num_records: the buffer load returns 0 into the LDS
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const h16* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(base), 0, (int)0x80000000u, 0x00020000);
}

constexpr int IMG = WK * WT;                     // h16 per operand image (16 KiB)

// GM: how the B rows of a K-step are found.
//   0  src(p) = p (linear layers): lane offsets are loop-invariant, a K-step advances one scalar offset.
//   1  3x3 taps of a same-size stride-1 conv whose grid keeps a K-step inside whole image rows of one frame ((H W) % 64 == 0 and
//      W % 64 == 0 or 64 % W == 0): src = p + (dy - 1) W + (dx - 1), again affine; what varies is which lanes' taps fall off the
//      image — two compares per lane on the K-step's uniform (row, column) origin.
//   2  temporal taps with HW % 64 == 0: src = p + (dt - 1) HW, a K-step is inside one frame, the tap exists or not for all of it.
//   3  anything else (stride 2, odd grids): per-lane walkers (Walker), correct for every geometry mudg_transpose_gather accepts.
template <int GM>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const MudgWgradDesc p, const int ntn) {
    __shared__ __attribute__((aligned(1024))) h16 smem[4 * IMG];           // two stages x (A image, B image)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int slice = blockIdx.y;
    const int64_t pbeg = (int64_t)slice * p.chunk, pend = pbeg + p.chunk < p.P ? pbeg + p.chunk : p.P;
    // DMA lane roles (see tr_fragment): position lane >> 4 of a piece, channel chunk swizzled; a wave stages pieces 4 wave + i, i < 4
    const int prow = lane >> 4, chunk = ((((lane & 15) >> 1) ^ (2 * prow)) << 1) | (lane & 1);
    const int am = tm * WT + chunk * 8;
    const bool aok = am < p.M;
    const int halves_per_tap = p.C >> 6;
    const int half = tn * 2 + (chunk >> 3);
    const int tap = half / halves_per_tap;
    const int bc = (half - tap * halves_per_tap) * 64 + (chunk & 7) * 8;
    const bool bok = tap < p.taps;
    const int tdy = tap / 3, tdx = tap - 3 * tdy;
    // buffer descriptors at the slice's first row (B: moved back by the most negative tap shift), so that offsets stay < 2 GiB
    const int64_t back = GM == 1 ? p.Wout + 1 : (GM == 2 ? p.HW : 0);
    const int64_t shift = GM == 1 ? (int64_t)(tdy - 1) * p.Wout + (tdx - 1) : (GM == 2 ? (int64_t)(tap - 1) * p.HW : 0);
    const h16* A = reinterpret_cast<const h16*>(p.A) + pbeg * p.lda;
    const h16* B = reinterpret_cast<const h16*>(p.B) + (GM == 3 ? 0 : (pbeg - back) * p.ldb);
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(A), rB = make_rsrc(B);
    unsigned va[4], vb[4];
    int rx[4], ry[4];
    Walker wk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * wave + 4 * i + prow;                         // position inside the K-step
        va[i] = aok ? (unsigned)(((int64_t)row * p.lda + am) * 2) : OOB;
        vb[i] = bok ? (unsigned)(((row + back + shift) * p.ldb + bc) * 2) : OOB;
        rx[i] = ry[i] = 0;
        if (GM == 1) { ry[i] = row / p.Wout; rx[i] = row - ry[i] * p.Wout + tdx - 1; ry[i] += tdy - 1; }
        if (GM == 3) wk[i].init(p, pbeg + row);
    }
    int sa = 0, sb = 0;                                                   // scalar byte offsets of the K-step inside the slice
    // uniform origin of the K-step: GM 1 (column, row) of its first position; GM 2 (pixel inside the frame, frame index t)
    int ux = 0, uy = 0;
    if (GM == 1) { const int64_t q = pbeg / p.Wout; ux = (int)(pbeg - q * p.Wout); uy = (int)(q % p.Hout); }
    if (GM == 2) { const int64_t q = pbeg / p.HW; ux = (int)(pbeg - q * p.HW); uy = (int)(q % p.T); }
    auto issue = [&](int64_t p0, int buf) {                               // called with p0 = pbeg, pbeg + 64, ...
        h16* As = smem + buf * 2 * IMG;
        h16* Bs = As + IMG;
        const int left = (int)(pend - p0 < WK ? pend - p0 : WK);          // positions of this K-step inside the slice
        const bool tap_live = GM != 2 || (unsigned)(uy + tap - 1) < (unsigned)p.T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * wave + 4 * i + prow;
            const bool in = row < left;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(As + (4 * wave + i) * 512), 16, (int)(in ? va[i] : OOB), sa, 0, 0);
            if (GM == 3) {
                unsigned v = OOB;
                if (in && bok) {
                    const int64_t s = wk[i].src(p, p0 + row, tdy, tdx, tap);
                    if (s >= 0) v = (unsigned)((s * p.ldb + bc) * 2);
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lptr_t)(Bs + (4 * wave + i) * 512), 16, (int)v, 0, 0, 0);
                wk[i].advance(p);
            } else {
                bool ok = in && tap_live;
                if (GM == 1) ok = ok && (unsigned)(ux + rx[i]) < (unsigned)p.Wout && (unsigned)(uy + ry[i]) < (unsigned)p.Hout;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lptr_t)(Bs + (4 * wave + i) * 512), 16, (int)(ok ? vb[i] : OOB), sb, 0, 0);
            }
        }
        sa += (int)(WK * p.lda * 2);
        sb += (int)(WK * p.ldb * 2);
        if (GM == 1) {                                                    // 64 positions on: whole rows (W <= 64) or part of one (W % 64 == 0)
            ux += WK;
            if (ux >= p.Wout) { uy += ux / p.Wout; ux = ux % p.Wout; }
            if (uy >= p.Hout) uy -= p.Hout;
        }
        if (GM == 2) {
            ux += WK;
            if (ux >= p.HW) { ux = 0; if (++uy == p.T) uy = 0; }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][0][i] = 0.f; acc[0][1][i] = 0.f; acc[1][0][i] = 0.f; acc[1][1][i] = 0.f; }
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned offa[2] = {tr_offset(0, wm * 64, lane), tr_offset(0, wm * 64 + 32, lane)};
    const unsigned offb[2] = {tr_offset(0, wn * 64, lane), tr_offset(0, wn * 64 + 32, lane)};
    if (pbeg < pend) issue(pbeg, 0);
    int buf = 0;
    for (int64_t p0 = pbeg; p0 < pend; p0 += WK, buf ^= 1) {
        __syncthreads();                                   // vmcnt(0) + barrier: this step's images have landed, the other stage is free
        if (p0 + WK < pend) issue(p0 + WK, buf ^ 1);       // in flight under the MFMAs below
        // fragment reads of k-step ks + 1 are in flight while the MFMAs of ks run (two register sets)
        const unsigned abase = lds0 + (unsigned)(buf * 2 * IMG * 2), bbase = abase + IMG * 2;
        Frag fa[2][2], fb[2][2];
        auto reads = [&](int set, int ks) {
            tr_issue(fa[set][0], abase + offa[0] + ks * 4096);           // 16 positions = 4 pieces = 4 KiB on
            tr_issue(fa[set][1], abase + offa[1] + ks * 4096);
            tr_issue(fb[set][0], bbase + offb[0] + ks * 4096);
            tr_issue(fb[set][1], bbase + offb[1] + ks * 4096);
        };
        reads(0, 0);
#pragma unroll
        for (int ks = 0; ks < WK / 16; ++ks) {
            const int set = ks & 1;
            if (ks + 1 < WK / 16) {
                reads(set ^ 1, ks + 1);
                tr_wait<8>(fa[set][0], fa[set][1], fb[set][0], fb[set][1]);
            } else {
                tr_wait<0>(fa[set][0], fa[set][1], fb[set][0], fb[set][1]);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = MFMA_32x32x16(frag_value(fa[set][mb]), frag_value(fb[set][nb]), acc[mb][nb]);
        }
    }
    // acc[mb][nb][i]: row m = (i & 3) + 8 (i >> 2) + 4 hi of the block, column n = lane & 31: 128-byte runs per register
    const int N = p.taps * p.C;
    const int l31 = lane & 31, hi = lane >> 5;
    float* out = p.out + (int64_t)slice * p.M * N;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = tn * WT + wn * 64 + nb * 32 + l31;
        if (n >= N) continue;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = tm * WT + wm * 64 + mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
                if (m < p.M) out[(int64_t)m * N + n] = acc[mb][nb][i];
            }
    }
}

}  // namespace
#endif

extern "C" int mudg_wgrad(const MudgWgradDesc* dp, void* stream) {
#if MUDG_PLANES == 1
    MUDG_REQUIRE(dp, "mudg_wgrad: null descriptor");
    const MudgWgradDesc d = *dp;
    MUDG_REQUIRE(d.A && d.B && d.out && d.P > 0 && d.M > 0 && d.C > 0 && d.taps > 0, "mudg_wgrad: bad arguments");
    MUDG_REQUIRE((d.M & 7) == 0 && (d.C & 63) == 0, "mudg_wgrad: M=%d must be a multiple of 8 and C=%d of 64", d.M, d.C);
    MUDG_REQUIRE(d.lda >= d.M && d.ldb >= d.C && (d.lda & 7) == 0 && (d.ldb & 7) == 0 && aligned16(d.A) && aligned16(d.B), "mudg_wgrad: strides / alignment");
    MUDG_REQUIRE(d.mode >= 0 && d.mode <= 2, "mudg_wgrad: mode %d", d.mode);
    if (d.mode == 0) MUDG_REQUIRE(d.taps == 1, "mudg_wgrad: mode 0 has one tap");
    if (d.mode == 1) MUDG_REQUIRE(d.taps == 9 && d.Hin > 0 && d.Win > 0 && d.Hout > 0 && d.Wout > 0 && d.stride > 0 && d.P % ((int64_t)d.Hout * d.Wout) == 0,
                                  "mudg_wgrad: conv geometry");
    if (d.mode == 2) MUDG_REQUIRE(d.taps == 3 && d.T > 0 && d.HW > 0 && d.P % ((int64_t)d.T * d.HW) == 0, "mudg_wgrad: temporal geometry");
    MUDG_REQUIRE(d.slices >= 1 && d.slices <= 65535 && d.chunk > 0 && d.chunk % WK == 0 && (int64_t)d.slices * d.chunk >= d.P, "mudg_wgrad: slices=%d chunk=%lld",
                 d.slices, (long long)d.chunk);
    const int ntm = (d.M + WT - 1) / WT, ntn = (d.taps * d.C + WT - 1) / WT;
    int gm = 3;
    if (d.mode == 0) gm = 0;
    else if (d.mode == 1 && d.stride == 1 && d.pad == 1 && d.Hin == d.Hout && d.Win == d.Wout && ((int64_t)d.Hout * d.Wout) % WK == 0 &&
             (d.Wout % WK == 0 || WK % d.Wout == 0)) gm = 1;
    else if (d.mode == 2 && d.HW % WK == 0) gm = 2;
    // 32-bit offsets from the descriptors' bases: A and (geometries 0-2) B are based at the slice's first row — B moved back by the
    // most negative tap shift — so a slice must fit the window; the generic geometry addresses B from the operand's start.
    {
        const int64_t back = gm == 1 ? d.Wout + 1 : (gm == 2 ? d.HW : 0);
        const int64_t brows = gm == 3 ? (d.mode == 1 ? (d.P / ((int64_t)d.Hout * d.Wout)) * d.Hin * d.Win : d.P) : d.chunk + WK + 2 * back;
        MUDG_REQUIRE((d.chunk + WK) * d.lda * 2 < (1ll << 31) && brows * d.ldb * 2 < (1ll << 31),
                     "mudg_wgrad: a slice exceeds the 2 GiB window of its buffer descriptor (more slices, or smaller operands)");
    }
    const dim3 grid((unsigned)(ntm * ntn), (unsigned)d.slices);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (gm == 0) hipLaunchKernelGGL(wgrad_kernel<0>, grid, dim3(256), 0, s, d, ntn);
    else if (gm == 1) hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(256), 0, s, d, ntn);
    else if (gm == 2) hipLaunchKernelGGL(wgrad_kernel<2>, grid, dim3(256), 0, s, d, ntn);
    else hipLaunchKernelGGL(wgrad_kernel<3>, grid, dim3(256), 0, s, d, ntn);
    return mudg_check_launch("mudg_wgrad");
#else
    (void)dp; (void)stream;
    MUDG_FAIL(MUDG_EUNSUPPORTED, "mudg_wgrad: the row-contracting weight-gradient kernel belongs to the 16-bit operand builds (the split builds use transposed copies + mudg_gemm)");
#endif
}
