// norm.hip — GroupNorm(+SiLU), LayerNorm and row softmax on channels-last h16 data.  All HBM-bound:
// every pass moves 16 bytes per lane, statistics are fp32 partials combined in fp64 in a fixed order
// (no atomics: results are bit-reproducible run to run).
#include "common.h"

namespace {

constexpr int GN_CMAX = 4096;

// Eight consecutive channels of one pixel as fp32, from operand (h16, PLANES pieces `ld / PLANES` apart) or fp32 storage.
// `raw` only issues the 16-byte requests and `decode` turns them into values, so a kernel can put several vectors' requests in
// flight before it touches any of them.
template <typename T> struct Load8;
template <> struct Load8<h16> {
    static constexpr int NR = PLANES;
    static __device__ __forceinline__ void get(const h16* p, int ld, float (&x)[8]) { load8_operand(p, ld / PLANES, x); }
    static __device__ __forceinline__ void raw(const h16* p, int ld, u32x4 (&r)[NR]) {
#pragma unroll
        for (int k = 0; k < PLANES; ++k) r[k] = ld16(p + (int64_t)k * (ld / PLANES));
    }
    static __device__ __forceinline__ void decode(const u32x4 (&r)[NR], float (&x)[8]) {
#pragma unroll
        for (int k = 0; k < PLANES; ++k) {
            const h16x8 t = as_h16x8(r[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = k ? x[e] + (float)t[e] : (float)t[e];
        }
    }
};
struct StreamH { _Float16 v; };       // fp16 residual-stream storage (KIND_F16): a type of its own, h16 may be _Float16 too
template <> struct Load8<StreamH> {
    static constexpr int NR = 1;
    static __device__ __forceinline__ void get(const StreamH* p, int, float (&x)[8]) { load8_f16(reinterpret_cast<const _Float16*>(p), x); }
    static __device__ __forceinline__ void raw(const StreamH* p, int, u32x4 (&r)[NR]) { r[0] = ld16(p); }
    static __device__ __forceinline__ void decode(const u32x4 (&r)[NR], float (&x)[8]) {
        union { u32x4 u; f16x8 h; } t; t.u = r[0];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)t.h[e];
    }
};
template <> struct Load8<float> {
    static constexpr int NR = 2;
    static __device__ __forceinline__ void get(const float* p, int, float (&x)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = a[e]; x[4 + e] = b[e]; }
    }
    static __device__ __forceinline__ void raw(const float* p, int, u32x4 (&r)[NR]) { r[0] = ld16(p); r[1] = ld16(p + 4); }
    static __device__ __forceinline__ void decode(const u32x4 (&r)[NR], float (&x)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = __uint_as_float(r[0][e]); x[4 + e] = __uint_as_float(r[1][e]); }
    }
};

// Row chunks per sample of the statistics pass.  A function of `rows` only: the partition decides the order of the
// fp32 partial sums, and a sample's result must not depend on how many other samples share the launch (clip-level data
// parallelism relies on bit-identical per-clip results, tests/test_fullsize_gpu.py).
__host__ __device__ inline int gn_chunks(int samples, int rows) {
    (void)samples;
    int want = rows / (rows >= 2048 ? 64 : 16);      // short samples (the 576- and 144-pixel levels): 16-row chunks, or 64 .. 288 workgroups
    if (want > 1024) want = 1024;                    //   walk 47 .. 94 MB (1.6 TB/s measured)
    return want < 1 ? 1 : want;
}

// Vectors per sweep of the statistics pass: a divisor of nvec, <= 256; the largest one whose row classes (256 / nvp of them) keep
// at least 240 lanes busy, else the one that keeps most busy.
inline int gn_stats_sweep(int nvec) {
    int best = 1, used = 0;
    for (int d = nvec < 256 ? nvec : 256; d >= 1; --d) {
        if (nvec % d) continue;
        const int u = (256 / d) * d;
        if (u >= 240) return d;
        if (u > used) { used = u; best = d; }
    }
    return best;
}

// Pass 1: per (sample, row-chunk) per-group partial sum / sum of squares.  A workgroup takes the chunk's rows nvp channel vectors at a
// time (gn_stats_sweep: the largest divisor of C / 8 that keeps >= 240 of the 256 lanes busy): thread (rsub, v) owns vector v of rows rsub, rsub + 256 / nvp, ... (240 of 256 lanes busy at 40 / 80 / 120
// vectors per row; the first kernel gave a wave one row and 64 vectors, 40 of 64 lanes at C = 320), four requests in flight per lane,
// the row classes of a vector folded through LDS in a fixed order, then channels -> groups.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ X, const T* __restrict__ X2,
                                                        int csplit, int ldx, int ldx2, int rows, int C, int groups,
                                                        int nchunks, int nvp, float* __restrict__ part_out) {
    extern __shared__ float gn_lds[];
    float* part = gn_lds;                               // [256][16]: a thread's eight sums and eight sums of squares
    float* chsum = gn_lds + 256 * 16;                   // [C][2]
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, smp = blockIdx.y;
    const int rpc = (rows + nchunks - 1) / nchunks;
    const int r0 = chunk * rpc, r1 = (r0 + rpc < rows) ? r0 + rpc : rows;
    const int nvec = C >> 3;
    const int RP = 256 / nvp;                           // row classes (nvp: vectors per sweep, a divisor of nvec chosen by the host)
    const int rsub = tid / nvp, vl = tid - rsub * nvp;
    const int64_t srow = (int64_t)smp * rows;
    constexpr int NR = Load8<T>::NR;

    for (int vbase = 0; vbase < nvec; vbase += nvp) {
        const int v = vbase + vl;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        if (rsub < RP && v < nvec) {
            const int c0 = v * 8;
            const T* base = X; int cc = c0, ld = ldx;
            if (c0 >= csplit) { base = X2; cc = c0 - csplit; ld = ldx2; }
            for (int r = r0 + rsub; r < r1; r += 4 * RP) {
                u32x4 raw[4][NR];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = (r + u * RP < r1) ? r + u * RP : r;
                    Load8<T>::raw(base + (srow + rr) * ld + cc, ld, raw[u]);
                }
                if constexpr (NR == 1) asm volatile("" : "+v"(raw[0][0]), "+v"(raw[1][0]), "+v"(raw[2][0]), "+v"(raw[3][0]));
                else if constexpr (NR == 2) asm volatile("" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]), "+v"(raw[2][0]),
                                                         "+v"(raw[2][1]), "+v"(raw[3][0]), "+v"(raw[3][1]));
                else asm volatile("" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[0][2]), "+v"(raw[1][0]), "+v"(raw[1][1]), "+v"(raw[1][2]),
                                  "+v"(raw[2][0]), "+v"(raw[2][1]), "+v"(raw[2][2]), "+v"(raw[3][0]), "+v"(raw[3][1]), "+v"(raw[3][2]));
                static_assert(NR <= 3, "the pin above lists its operands");
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (r + u * RP >= r1) break;
                    float xv[8];
                    Load8<T>::decode(raw[u], xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s[e] += xv[e]; q[e] = fmaf(xv[e], xv[e], q[e]); }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { part[tid * 16 + e] = s[e]; part[tid * 16 + 8 + e] = q[e]; }
        __syncthreads();
        for (int i = tid; i < nvp * 16; i += 256) {
            const int ln = i >> 4, j = i & 15;
            if (vbase + ln < nvec) {
                float t = 0.f;
                for (int k = 0; k < RP; ++k) t += part[(k * nvp + ln) * 16 + j];
                chsum[((vbase + ln) * 8 + (j & 7)) * 2 + (j >> 3)] = t;
            }
        }
        __syncthreads();
    }
    if (tid < groups) {
        const int cpg = C / groups;
        float a = 0.f, b = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += chsum[c * 2]; b += chsum[c * 2 + 1]; }
        float* o = part_out + (((int64_t)smp * nchunks + chunk) * groups + tid) * 2;
        o[0] = a; o[1] = b;
    }
}

// Pass 2: fold the chunk partials into mean / rstd per (sample, group): one wave per pair, lanes stride over the
// chunks in fp64 and combine with a fixed butterfly, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stat,
                                                           int samples, int groups, int nchunks, double count, float eps) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= samples * groups) return;
    const int smp = i / groups, g = i - smp * groups;
    double a = 0.0, b = 0.0;
    for (int ch = lane; ch < nchunks; ch += 64) {
        const float* p = part + (((int64_t)smp * nchunks + ch) * groups + g) * 2;
        a += (double)p[0]; b += (double)p[1];
    }
    a = wave_sum_d(a); b = wave_sum_d(b);
    if (lane == 0) {
        const double mean = a / count;
        double var = b / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[i * 2] = (float)mean;
        stat[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// Pass 2': the same fold over the per-(128-row block, channel) partials a producing GEMM / conv epilogue wrote
// (MudgGemmDesc.stats): one 256-thread workgroup per (sample, group); thread t takes row blocks t, t + 256, ... and walks
// the group's channels (8-byte loads; the channels may straddle the two sources); fp64 partials are combined by a fixed
// wave butterfly and a fixed 4-term sum, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void gn_finalize_ch_kernel(const float* __restrict__ P1, const float* __restrict__ P2,
                                                              int csplit, int C, float* __restrict__ stat, int samples,
                                                              int groups, int blocks_per_sample, int blocks_per_sample2, double count, float eps) {
    __shared__ double red[4][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x;
    const int smp = i / groups, g = i - smp * groups;
    const int cpg = C / groups, c0 = g * cpg;
    double a = 0.0, b = 0.0;
    if (blocks_per_sample2 != blocks_per_sample) {
        // the two sources' producers wrote blocks of different heights (128 | 160 | 288): each source's blocks on their own, source 1 first
        for (int src = 0; src < 2; ++src) {
            const int bps = src ? blocks_per_sample2 : blocks_per_sample;
            const int lo = src ? (c0 > csplit ? c0 : csplit) : c0, hi = src ? c0 + cpg : (c0 + cpg < csplit ? c0 + cpg : csplit);
            for (int rb = tid; rb < bps && lo < hi; rb += 256) {
                const int64_t blk = (int64_t)smp * bps + rb;
                float sa = 0.f, sb = 0.f;
                for (int c = lo; c < hi; ++c) {
                    const f32x2 q = *reinterpret_cast<const f32x2*>(src ? P2 + (blk * (C - csplit) + (c - csplit)) * 2 : P1 + (blk * csplit + c) * 2);
                    sa += q[0]; sb += q[1];
                }
                a += (double)sa; b += (double)sb;
            }
        }
    } else
    for (int rb = tid; rb < blocks_per_sample; rb += 256) {
        const int64_t blk = (int64_t)smp * blocks_per_sample + rb;
        float sa = 0.f, sb = 0.f;                 // <= 80 channels of one block: fp32 is exact enough before the fp64 fold
#pragma unroll 4
        for (int c = c0; c < c0 + cpg; ++c) {
            const f32x2 q = *reinterpret_cast<const f32x2*>((c < csplit) ? P1 + (blk * csplit + c) * 2
                                                                         : P2 + (blk * (C - csplit) + (c - csplit)) * 2);
            sa += q[0]; sb += q[1];
        }
        a += (double)sa; b += (double)sb;
    }
    a = wave_sum_d(a); b = wave_sum_d(b);
    if (lane == 0) { red[wave][0] = a; red[wave][1] = b; }
    __syncthreads();
    if (tid == 0) {
        a = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        b = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
        const double mean = a / count;
        double var = b / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[i * 2] = (float)mean;
        stat[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// Pass 3: y = (x - mean) * rstd * gamma + beta, optional SiLU.  HBM-bound (2 B in + 2 B out per element on the 16-bit builds).
// One-shot workgroups, as ln_rows_kernel: a workgroup owns RB rows x one slab of CS channels (<= 640, so its scale / shift table
// is a few KiB of dynamic LDS and costs fewer requests than the payload), every lane puts its GN_UNROLL 16-byte requests in
// flight first, builds the table while they travel, applies, stores and ENDS.  A looping workgroup (rounds 1-3: 2048 resident
// workgroups walking 144 rows each) issues its next loads behind its own stores — vmcnt retires in order, so every iteration pays
// a load AND a store round trip — and measured 3.6-4.3 TB/s whatever the unroll; short-lived workgroups let the dispatcher
// overlap one workgroup's stores with the next one's loads.
constexpr int GN_UNROLL = 4;
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ X, const T* __restrict__ X2,
                                                        int csplit, int ldx, int ldx2, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, h16* __restrict__ Y, int ldy,
                                                        int rows, int C, int groups, int CS, int RB, int silu,
                                                        const float* __restrict__ stat) {
    extern __shared__ float gn_tab[];
    float* sc = gn_tab;
    float* sh = gn_tab + CS;
    const int tid = threadIdx.x;
    const int smp = blockIdx.y;
    const int cbase = blockIdx.z * CS;
    const int nvec = CS >> 3;
    const int r0 = blockIdx.x * RB, r1 = (r0 + RB < rows) ? r0 + RB : rows;
    const int64_t srow = (int64_t)smp * rows;
    const int n = (r1 - r0) * nvec;                        // <= 256 * GN_UNROLL (the host sizes RB)
    const int dr = 256 / nvec, dv = 256 - dr * nvec;       // entry i + 256 is dr rows and dv vectors further (with carry)
    int r = r0 + tid / nvec, v = tid - (tid / nvec) * nvec;
    constexpr int NR = Load8<T>::NR;
    u32x4 raw[GN_UNROLL][NR];
    int64_t yoff[GN_UNROLL];
    int c0s[GN_UNROLL];
#pragma unroll
    for (int u = 0; u < GN_UNROLL; ++u) {
        const bool live = tid + 256 * u < n;               // past the end: re-read the workgroup's first vector, store nothing
        const int cl = live ? v * 8 : 0;                   // (unconditional loads: the GN_UNROLL requests leave back to back)
        const int rr = live ? r : r0;
        const int c0 = cbase + cl;
        c0s[u] = cl;
        yoff[u] = (srow + rr) * ldy + c0;
        const T* base = X; int cc = c0, ld = ldx;
        if (c0 >= csplit) { base = X2; cc = c0 - csplit; ld = ldx2; }
        Load8<T>::raw(base + (srow + rr) * ld + cc, ld, raw[u]);
        r += dr; v += dv;
        if (v >= nvec) { v -= nvec; ++r; }
    }
    const int cpg = C / groups;
    for (int cl = tid; cl < CS; cl += 256) {
        const int c = cbase + cl;
        const int g = c / cpg;
        const float mean = stat[(smp * groups + g) * 2], rstd = stat[(smp * groups + g) * 2 + 1];
        const float a = rstd * gamma[c];
        sc[cl] = a;
        sh[cl] = beta[c] - mean * a;
    }
    __syncthreads();
    // every request is pinned here, in straight-line code, before any value is decoded: without it the compiler sinks each
    // load into the conditional block that stores its result, and the requests go out one at a time
    if constexpr (NR == 1) asm volatile("" : "+v"(raw[0][0]), "+v"(raw[1][0]), "+v"(raw[2][0]), "+v"(raw[3][0]));
    else if constexpr (NR == 2) asm volatile("" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]), "+v"(raw[2][0]),
                                             "+v"(raw[2][1]), "+v"(raw[3][0]), "+v"(raw[3][1]));
    else asm volatile("" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[0][2]), "+v"(raw[1][0]), "+v"(raw[1][1]), "+v"(raw[1][2]),
                      "+v"(raw[2][0]), "+v"(raw[2][1]), "+v"(raw[2][2]), "+v"(raw[3][0]), "+v"(raw[3][1]), "+v"(raw[3][2]));
    static_assert(GN_UNROLL == 4 && NR <= 3, "the pin above lists its operands");
#pragma unroll
    for (int u = 0; u < GN_UNROLL; ++u) {
        if (tid + 256 * u >= n) break;
        float xv[8];
        Load8<T>::decode(raw[u], xv);
        const int cl = c0s[u];
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = fmaf(xv[e], sc[cl + e], sh[cl + e]);
            if (silu) y = PLANES > 1 ? y / (1.0f + expf(-y))                   // split builds: IEEE division, full-precision exp
                                     : y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));      // 1-ulp reciprocal: the result is rounded to h16
            o[e] = y;
        }
        store8_operand(Y + yoff[u], ldy / PLANES, o);
    }
}

// The same pass with the scale / shift of a lane's eight channels in REGISTERS: a lane keeps one channel vector and takes it from
// GN_UNROLL rows (256 / nvec rows per pass, so 240 of 256 lanes work at nvec = 40 / 80), gamma / beta / the statistics come straight
// from global memory (L1 / L2 hits) while the payload requests travel — no LDS table, no barrier, no bank conflicts (the table's
// 16-byte reads at a 32-byte lane stride were two-way conflicted).  Needs nvec <= 256.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_reg_kernel(const T* __restrict__ X, const T* __restrict__ X2,
                                                            int csplit, int ldx, int ldx2, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, h16* __restrict__ Y, int ldy,
                                                            int rows, int C, int groups, int CS, int silu,
                                                            const float* __restrict__ stat) {
    const int tid = threadIdx.x;
    const int nvec = CS >> 3;
    const int RP = 256 / nvec;                             // rows per pass
    const int rsub = tid / nvec, v = tid - rsub * nvec;
    const int r0 = blockIdx.x * (RP * GN_UNROLL) + rsub;
    if (rsub >= RP || r0 >= rows) return;
    const int smp = blockIdx.y;
    const int c0 = blockIdx.z * CS + v * 8;
    const int64_t srow = (int64_t)smp * rows;
    const T* base = X; int cc = c0, ld = ldx;
    if (c0 >= csplit) { base = X2; cc = c0 - csplit; ld = ldx2; }
    constexpr int NR = Load8<T>::NR;
    u32x4 raw[GN_UNROLL][NR];
#pragma unroll
    for (int u = 0; u < GN_UNROLL; ++u) {
        const int r = r0 + RP * u;
        Load8<T>::raw(base + (srow + (r < rows ? r : rows - 1)) * ld + cc, ld, raw[u]);
    }
    const int cpg = C / groups;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c0), b1 = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
    const f32x2* st = reinterpret_cast<const f32x2*>(stat) + smp * groups;
    f32x2 mr[8];
    const int gq = c0 / cpg, rem = c0 - gq * cpg;
    if (cpg >= 8) {                                        // eight channels meet at most two groups: two requests, one division
        const f32x2 ma = st[gq], mb = st[gq + 1 < groups ? gq + 1 : gq];
#pragma unroll
        for (int e = 0; e < 8; ++e) mr[e] = (rem + e >= cpg) ? mb : ma;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) mr[e] = st[(c0 + e) / cpg];
    }
    // Everything above is requests only; nothing below may be scheduled above this line (left alone, the scheduler spreads the
    // requests between the uses to save registers: three round trips).  The pin keeps the payload loads out of the conditional
    // blocks that consume them (the IR-level sinking the scheduling barrier does not see).
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NR == 1) asm volatile("" : "+v"(raw[0][0]), "+v"(raw[1][0]), "+v"(raw[2][0]), "+v"(raw[3][0]));
    else if constexpr (NR == 2) asm volatile("" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]), "+v"(raw[2][0]),
                                             "+v"(raw[2][1]), "+v"(raw[3][0]), "+v"(raw[3][1]));
    else asm volatile("" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[0][2]), "+v"(raw[1][0]), "+v"(raw[1][1]), "+v"(raw[1][2]),
                      "+v"(raw[2][0]), "+v"(raw[2][1]), "+v"(raw[2][2]), "+v"(raw[3][0]), "+v"(raw[3][1]), "+v"(raw[3][2]));
    static_assert(GN_UNROLL == 4 && NR <= 3, "the pin above lists its operands");
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = mr[e][1] * (e < 4 ? g0[e & 3] : g1[e & 3]);
        sc[e] = a;
        sh[e] = (e < 4 ? b0[e & 3] : b1[e & 3]) - mr[e][0] * a;
    }
#pragma unroll
    for (int u = 0; u < GN_UNROLL; ++u) {
        const int r = r0 + RP * u;
        if (u > 0 && r >= rows) break;
        float xv[8];
        Load8<T>::decode(raw[u], xv);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = fmaf(xv[e], sc[e], sh[e]);
            if (silu) y = PLANES > 1 ? y / (1.0f + expf(-y))
                                     : y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
            o[e] = y;
        }
        store8_operand(Y + (srow + r) * ldy + c0, ldy / PLANES, o);
    }
}

// The apply pass's geometry: the channel slab CS = the largest divisor of C that is <= 640 and a multiple of 64 (of 8 if there is
// none; C itself when C <= 640 or when it has no such divisor >= 128); a workgroup takes GN_UNROLL passes of 256 / nvec rows.
void launch_gn_apply(int x_fp32, const void* X, const void* X2, int csplit, int ldx, int ldx2, const float* gamma, const float* beta,
                     void* Y, int ldy, int samples, int rows, int C, int groups, int silu, const float* stat, hipStream_t s) {
    int CS = C;
    if (C > 640) {
        int best = 0;
        for (int d = 640; d >= 128 && !best; d -= 64)      // whole 128-byte lines per row segment first (960 -> 3 x 320, not 2 x 480)
            if (C % d == 0) best = d;
        for (int d = 640; d >= 128 && !best; d -= 8)
            if (C % d == 0) best = d;
        if (best) CS = best;
    }
    static int reg_table = -1;              // MUDG_GN_REG=0: the LDS-table kernel for every width (A/B, tests)
    if (reg_table < 0) reg_table = mudg_variant("GN_REG", 1);
    const int nvec = CS >> 3;
    if (reg_table && nvec <= 256 && aligned16(gamma) && aligned16(beta)) {
        const int RB = (256 / nvec) * GN_UNROLL;
        const dim3 grid((rows + RB - 1) / RB, samples, C / CS);
        if (x_fp32 == KIND_F16)
            hipLaunchKernelGGL(gn_apply_reg_kernel<StreamH>, grid, dim3(256), 0, s, (const StreamH*)X, (const StreamH*)X2,
                               csplit, ldx, ldx2, gamma, beta, (h16*)Y, ldy, rows, C, groups, CS, silu, stat);
        else if (x_fp32)
            hipLaunchKernelGGL(gn_apply_reg_kernel<float>, grid, dim3(256), 0, s, (const float*)X, (const float*)X2,
                               csplit, ldx, ldx2, gamma, beta, (h16*)Y, ldy, rows, C, groups, CS, silu, stat);
        else
            hipLaunchKernelGGL(gn_apply_reg_kernel<h16>, grid, dim3(256), 0, s, (const h16*)X, (const h16*)X2,
                               csplit, ldx, ldx2, gamma, beta, (h16*)Y, ldy, rows, C, groups, CS, silu, stat);
        return;
    }
    int RB = (256 * GN_UNROLL) / nvec;
    if (RB < 1) RB = 1;
    const dim3 grid((rows + RB - 1) / RB, samples, C / CS);
    const size_t lds = 2 * (size_t)CS * sizeof(float);
    if (x_fp32 == KIND_F16)
        hipLaunchKernelGGL(gn_apply_kernel<StreamH>, grid, dim3(256), lds, s, (const StreamH*)X, (const StreamH*)X2,
                           csplit, ldx, ldx2, gamma, beta, (h16*)Y, ldy, rows, C, groups, CS, RB, silu, stat);
    else if (x_fp32)
        hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), lds, s, (const float*)X, (const float*)X2,
                           csplit, ldx, ldx2, gamma, beta, (h16*)Y, ldy, rows, C, groups, CS, RB, silu, stat);
    else
        hipLaunchKernelGGL(gn_apply_kernel<h16>, grid, dim3(256), lds, s, (const h16*)X, (const h16*)X2,
                           csplit, ldx, ldx2, gamma, beta, (h16*)Y, ldy, rows, C, groups, CS, RB, silu, stat);
}

// LayerNorm: one wave per row, up to VMAX 16-byte vectors per lane kept in registers (C <= 512 * VMAX).
template <int VMAX, typename T>
__global__ __launch_bounds__(256) void ln_kernel(const T* __restrict__ X, int ldx, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, h16* __restrict__ Y, int ldy,
                                                  int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = C >> 3;
    float x[VMAX][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int v = lane + 64 * i;
        if (v < nvec) {
            Load8<T>::get(X + row * ldx + v * 8, ldx, x[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[i][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[i][e] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int v = lane + 64 * i;
        if (v < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = x[i][e] - mean; q = fmaf(d, d, q); }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int v = lane + 64 * i;
        if (v < nvec) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + v * 8);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + v * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + v * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(beta + v * 8 + 4);
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = fmaf((x[i][e] - mean) * rstd, g0[e], b0[e]);
                o[4 + e] = fmaf((x[i][4 + e] - mean) * rstd, g1[e], b1[e]);
            }
            store8_operand(Y + row * ldy + v * 8, ldy / PLANES, o);
        }
    }
}

// LayerNorm for the UNet's widths (C = 8 * LPR * NV): LPR lanes per row, 64 / LPR rows per wave, NV 16-byte vectors per lane.
// Every request of a lane — its NV payload vectors AND the gamma / beta of its 8 NV channels — is issued before anything waits
// (rounds 1-3 fetched gamma / beta inside the store loop: the compiler serialised them into seven L2 round trips per wave, and the
// butterflies ran on ds_bpermute); the payload stays packed in registers and is decoded once per pass (sum, squared deviations,
// apply), the two reductions are DPP steps inside a 16-lane row (+ one cross-row exchange at LPR = 32).  NV <= 5 keeps gamma /
// beta at <= 80 registers: 320 = 8 x 5, 640 = 16 x 5, 1280 = 32 x 5, 512 = 16 x 4, 1024 = 32 x 4.
template <int CTRL>
__device__ __forceinline__ float ln_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int LPR>
__device__ __forceinline__ float ln_group_sum(float v) {       // every lane of an LPR-lane group ends with the group's total
    v = ln_dpp_add<0xB1>(v);                                   // quad_perm [1,0,3,2]
    v = ln_dpp_add<0x4E>(v);                                   // quad_perm [2,3,0,1]
    v = ln_dpp_add<0x141>(v);                                  // row_half_mirror
    if constexpr (LPR >= 16) v = ln_dpp_add<0x140>(v);         // row_mirror
    if constexpr (LPR >= 32) v += __shfl_xor(v, 16, 64);
    return v;
}

template <int LPR, int NV, typename T>
__global__ __launch_bounds__(256) void ln_rows_kernel(const T* __restrict__ X, int ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, h16* __restrict__ Y, int ldy,
                                                       int rows, float eps) {
    constexpr int RPW = 64 / LPR;                       // rows per wave
    constexpr int C = LPR * NV * 8;
    constexpr int NR = Load8<T>::NR;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1);
    int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < rows;
    if (!live) row = rows - 1;                          // keeps the reductions uniform; nothing is stored for it
    u32x4 raw[NV][NR];
    f32x4 g[NV][2], b[NV][2];
#pragma unroll
    for (int i = 0; i < NV; ++i) Load8<T>::raw(X + row * ldx + (sub + LPR * i) * 8, ldx, raw[i]);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (sub + LPR * i) * 8;
        g[i][0] = *reinterpret_cast<const f32x4*>(gamma + c); g[i][1] = *reinterpret_cast<const f32x4*>(gamma + c + 4);
        b[i][0] = *reinterpret_cast<const f32x4*>(beta + c); b[i][1] = *reinterpret_cast<const f32x4*>(beta + c + 4);
    }
    __builtin_amdgcn_sched_barrier(0);                  // requests above, arithmetic below
    // (and pinned in this unconditional block: left alone, the gamma / beta loads sink into the `live` branch that consumes them)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        asm volatile("" : "+v"(g[i][0]), "+v"(g[i][1]), "+v"(b[i][0]), "+v"(b[i][1]));
#pragma unroll
        for (int k = 0; k < NR; ++k) asm volatile("" : "+v"(raw[i][k]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float x[8];
        Load8<T>::decode(raw[i], x);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[e];
    }
    const float mean = ln_group_sum<LPR>(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float x[8];
        Load8<T>::decode(raw[i], x);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = x[e] - mean; q = fmaf(d, d, q); }
    }
    const float rstd = rsqrtf(ln_group_sum<LPR>(q) / (float)C + eps);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float x[8], o[8];
        Load8<T>::decode(raw[i], x);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = fmaf((x[e] - mean) * rstd, g[i][0][e], b[i][0][e]);
            o[4 + e] = fmaf((x[4 + e] - mean) * rstd, g[i][1][e], b[i][1][e]);
        }
        store8_operand(Y + row * ldy + (sub + LPR * i) * 8, ldy / PLANES, o);
    }
}

template <int LPR, int NV>
void launch_ln_rows(int x_fp32, const void* X, int ldx, const float* gamma, const float* beta, void* Y, int ldy, int rows,
                    float eps, hipStream_t s) {
    const dim3 grid((rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)));
    if (x_fp32 == KIND_F16) hipLaunchKernelGGL((ln_rows_kernel<LPR, NV, StreamH>), grid, dim3(256), 0, s, (const StreamH*)X, ldx, gamma, beta, (h16*)Y, ldy, rows, eps);
    else if (x_fp32) hipLaunchKernelGGL((ln_rows_kernel<LPR, NV, float>), grid, dim3(256), 0, s, (const float*)X, ldx, gamma, beta, (h16*)Y, ldy, rows, eps);
    else hipLaunchKernelGGL((ln_rows_kernel<LPR, NV, h16>), grid, dim3(256), 0, s, (const h16*)X, ldx, gamma, beta, (h16*)Y, ldy, rows, eps);
}

// Row softmax fp32 -> h16, one workgroup per row, three passes over the row (second and third hit L2).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int lds, h16* __restrict__ P,
                                                            int ldp, int cols) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = S + (int64_t)blockIdx.x * lds;
    h16* p = P + (int64_t)blockIdx.x * ldp;
    float mx = -INFINITY;
    for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, s[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = tid; c < cols; c += 256) sum += PLANES > 1 ? expf(s[c] - mx) : __expf(s[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = ((red[0] + red[1]) + red[2]) + red[3];
    const float inv = 1.f / sum;
    for (int c = tid; c < cols; c += 256) store1_operand(p + c, ldp / PLANES, (PLANES > 1 ? expf(s[c] - mx) : __expf(s[c] - mx)) * inv);
}

}  // namespace

extern "C" int64_t mudg_groupnorm_ws_floats(int samples, int groups, int rows) {
    if (samples <= 0 || groups <= 0 || rows <= 0) return 0;
    return (int64_t)samples * groups * 2 * (gn_chunks(samples, rows) + 1);
}

extern "C" int mudg_groupnorm(const void* X, const void* X2, int csplit, int ldx, int ldx2, int x_fp32, const float* gamma,
                              const float* beta, void* Y, int ldy, int samples, int rows, int C, int groups, float eps,
                              int silu, float* ws, void* stream) {
    MUDG_REQUIRE(X && Y && gamma && beta && ws, "mudg_groupnorm: null pointer");
    MUDG_REQUIRE(samples > 0 && rows > 0 && C > 0 && groups > 0, "mudg_groupnorm: empty problem");
    MUDG_REQUIRE(C % groups == 0 && (C & 7) == 0 && C <= GN_CMAX, "mudg_groupnorm: C=%d groups=%d unsupported", C, groups);
    MUDG_REQUIRE(groups <= 256, "mudg_groupnorm: groups=%d > 256", groups);
    MUDG_REQUIRE(x_fp32 >= 0 && x_fp32 <= 2, "mudg_groupnorm: x_fp32 is 0 (operand), 1 (fp32) or 2 (fp16)");
    const int xq = x_fp32 == KIND_F32 ? 4 : (x_fp32 == KIND_F16 ? 8 : 8 * PLANES);      // row stride granule so that every 8-channel vector (of every plane) is 16-byte aligned
    MUDG_REQUIRE(ldx % xq == 0 && ldy % (8 * PLANES) == 0 && aligned16(X) && aligned16(Y), "mudg_groupnorm: alignment");
    MUDG_REQUIRE(ldy / PLANES >= C, "mudg_groupnorm: ldy=%d too small for %d plane(s) of %d channels", ldy, PLANES, C);
    MUDG_REQUIRE(samples <= 65535, "mudg_groupnorm: too many samples");
    if (!X2) { csplit = C; ldx2 = ldx; }
    else MUDG_REQUIRE(csplit > 0 && csplit < C && (csplit & 7) == 0 && ldx2 % xq == 0 && aligned16(X2), "mudg_groupnorm: X2/csplit");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nchunks = gn_chunks(samples, rows);
    float* part = ws;
    float* stat = ws + (int64_t)samples * nchunks * groups * 2;
    const int slot = mudg_prof_begin(MUDG_FAM_GNORM, s);
    const size_t stats_lds = (256 * 16 + 2 * (size_t)C) * sizeof(float);
    const int nvp = gn_stats_sweep(C >> 3);
    if (x_fp32 == KIND_F16)
        hipLaunchKernelGGL(gn_stats_kernel<StreamH>, dim3(nchunks, samples), dim3(256), stats_lds, s, (const StreamH*)X, (const StreamH*)X2,
                           csplit, ldx, ldx2, rows, C, groups, nchunks, nvp, part);
    else if (x_fp32)
        hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(nchunks, samples), dim3(256), stats_lds, s, (const float*)X, (const float*)X2,
                           csplit, ldx, ldx2, rows, C, groups, nchunks, nvp, part);
    else
        hipLaunchKernelGGL(gn_stats_kernel<h16>, dim3(nchunks, samples), dim3(256), stats_lds, s, (const h16*)X, (const h16*)X2,
                           csplit, ldx, ldx2, rows, C, groups, nchunks, nvp, part);
    const int ng = samples * groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((ng + 3) / 4), dim3(256), 0, s, part, stat, samples, groups, nchunks,
                       (double)rows * (C / groups), eps);
    launch_gn_apply(x_fp32, X, X2, csplit, ldx, ldx2, gamma, beta, Y, ldy, samples, rows, C, groups, silu, stat, s);
    const int rc = mudg_check_launch("mudg_groupnorm");
    mudg_prof_end(slot, s, 0.0, (double)samples * rows * C * (x_fp32 == KIND_F32 ? 10.0 : 6.0));
    return rc;
}

extern "C" int mudg_groupnorm_fused_rows(const void* X, const void* X2, int csplit, int ldx, int ldx2, int x_fp32,
                                         const float* gamma, const float* beta, void* Y, int ldy, int samples, int rows, int C,
                                         int groups, float eps, int silu, const float* P1, int p1_rows, const float* P2, int p2_rows,
                                         float* ws, void* stream);
extern "C" int mudg_groupnorm_fused(const void* X, const void* X2, int csplit, int ldx, int ldx2, int x_fp32,
                                    const float* gamma, const float* beta, void* Y, int ldy, int samples, int rows, int C,
                                    int groups, float eps, int silu, const float* P1, const float* P2, float* ws, void* stream) {
    return mudg_groupnorm_fused_rows(X, X2, csplit, ldx, ldx2, x_fp32, gamma, beta, Y, ldy, samples, rows, C, groups, eps, silu, P1, 128, P2, 128,
                                     ws, stream);
}
extern "C" int mudg_groupnorm_fused_rows(const void* X, const void* X2, int csplit, int ldx, int ldx2, int x_fp32,
                                         const float* gamma, const float* beta, void* Y, int ldy, int samples, int rows, int C,
                                         int groups, float eps, int silu, const float* P1, int p1_rows, const float* P2, int p2_rows,
                                         float* ws, void* stream) {
    MUDG_REQUIRE(X && Y && gamma && beta && ws && P1, "mudg_groupnorm_fused: null pointer");
    MUDG_REQUIRE((p1_rows == 128 || p1_rows == 160 || p1_rows == 288) && (!X2 || p2_rows == 128 || p2_rows == 160 || p2_rows == 288),
                 "mudg_groupnorm_fused: partial blocks are 128, 160 or 288 rows high");
    if (!X2) p2_rows = p1_rows;
    MUDG_REQUIRE(rows % p1_rows == 0 && rows % p2_rows == 0, "mudg_groupnorm_fused: rows=%d per sample must be whole partial blocks (%d / %d rows)", rows, p1_rows, p2_rows);
    MUDG_REQUIRE(samples > 0 && rows > 0 && C > 0 && groups > 0, "mudg_groupnorm_fused: empty problem");
    MUDG_REQUIRE(C % groups == 0 && (C & 7) == 0 && C <= GN_CMAX && groups <= 256, "mudg_groupnorm_fused: C=%d groups=%d unsupported", C, groups);
    MUDG_REQUIRE(x_fp32 >= 0 && x_fp32 <= 2, "mudg_groupnorm_fused: x_fp32 is 0 (operand), 1 (fp32) or 2 (fp16)");
    const int xq = x_fp32 == KIND_F32 ? 4 : (x_fp32 == KIND_F16 ? 8 : 8 * PLANES);
    MUDG_REQUIRE(ldx % xq == 0 && ldy % (8 * PLANES) == 0 && aligned16(X) && aligned16(Y), "mudg_groupnorm_fused: alignment");
    MUDG_REQUIRE(ldy / PLANES >= C, "mudg_groupnorm_fused: ldy=%d too small for %d plane(s) of %d channels", ldy, PLANES, C);
    MUDG_REQUIRE(samples <= 65535, "mudg_groupnorm_fused: too many samples");
    if (!X2) { csplit = C; ldx2 = ldx; P2 = P1; }
    else MUDG_REQUIRE(P2 && csplit > 0 && csplit < C && (csplit & 7) == 0 && ldx2 % xq == 0 && aligned16(X2), "mudg_groupnorm_fused: X2/csplit/P2");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* stat = ws;
    const int slot = mudg_prof_begin(MUDG_FAM_GNORM, s);
    const int ng = samples * groups;
    hipLaunchKernelGGL(gn_finalize_ch_kernel, dim3(ng), dim3(256), 0, s, P1, P2, csplit, C, stat, samples, groups,
                       rows / p1_rows, rows / p2_rows, (double)rows * (C / groups), eps);
    launch_gn_apply(x_fp32, X, X2, csplit, ldx, ldx2, gamma, beta, Y, ldy, samples, rows, C, groups, silu, stat, s);
    const int rc = mudg_check_launch("mudg_groupnorm_fused");
    mudg_prof_end(slot, s, 0.0, (double)samples * rows * C * (x_fp32 == KIND_F32 ? 6.0 : 4.0));
    return rc;
}

extern "C" int mudg_layernorm(const void* X, int ldx, int x_fp32, const float* gamma, const float* beta, void* Y, int ldy,
                              int rows, int C, float eps, void* stream) {
    MUDG_REQUIRE(X && Y && gamma && beta, "mudg_layernorm: null pointer");
    MUDG_REQUIRE(rows > 0 && C > 0 && (C & 7) == 0 && C <= 4096, "mudg_layernorm: rows=%d C=%d unsupported", rows, C);
    MUDG_REQUIRE(x_fp32 >= 0 && x_fp32 <= 2, "mudg_layernorm: x_fp32 is 0 (operand), 1 (fp32) or 2 (fp16)");
    MUDG_REQUIRE(ldx % (x_fp32 == KIND_F32 ? 4 : (x_fp32 == KIND_F16 ? 8 : 8 * PLANES)) == 0 && ldy % (8 * PLANES) == 0 && aligned16(X) && aligned16(Y) && aligned16(gamma) &&
                 aligned16(beta), "mudg_layernorm: alignment");
    MUDG_REQUIRE(ldy / PLANES >= C, "mudg_layernorm: ldy=%d too small for %d plane(s) of %d channels", ldy, PLANES, C);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int slot = mudg_prof_begin(MUDG_FAM_LNORM, s);
    const dim3 grid((rows + 3) / 4);
    const int nvec = C >> 3;
    static int rows_kernel = -1;            // MUDG_LN_ROWS=0: the one-wave-per-row kernel for every width (A/B, tests)
    if (rows_kernel < 0) rows_kernel = mudg_variant("LN_ROWS", 1);
    if (rows_kernel && C == 320) launch_ln_rows<8, 5>(x_fp32, X, ldx, gamma, beta, Y, ldy, rows, eps, s);
    else if (rows_kernel && C == 512) launch_ln_rows<16, 4>(x_fp32, X, ldx, gamma, beta, Y, ldy, rows, eps, s);
    else if (rows_kernel && C == 640) launch_ln_rows<16, 5>(x_fp32, X, ldx, gamma, beta, Y, ldy, rows, eps, s);
    else if (rows_kernel && C == 1024) launch_ln_rows<32, 4>(x_fp32, X, ldx, gamma, beta, Y, ldy, rows, eps, s);
    else if (rows_kernel && C == 1280) launch_ln_rows<32, 5>(x_fp32, X, ldx, gamma, beta, Y, ldy, rows, eps, s);
    else if (nvec <= 64 * 3) {
        if (x_fp32 == KIND_F16) hipLaunchKernelGGL((ln_kernel<3, StreamH>), grid, dim3(256), 0, s, (const StreamH*)X, ldx, gamma, beta, (h16*)Y, ldy, rows, C, eps);
        else if (x_fp32) hipLaunchKernelGGL((ln_kernel<3, float>), grid, dim3(256), 0, s, (const float*)X, ldx, gamma, beta, (h16*)Y, ldy, rows, C, eps);
        else hipLaunchKernelGGL((ln_kernel<3, h16>), grid, dim3(256), 0, s, (const h16*)X, ldx, gamma, beta, (h16*)Y, ldy, rows, C, eps);
    } else {
        if (x_fp32 == KIND_F16) hipLaunchKernelGGL((ln_kernel<8, StreamH>), grid, dim3(256), 0, s, (const StreamH*)X, ldx, gamma, beta, (h16*)Y, ldy, rows, C, eps);
        else if (x_fp32) hipLaunchKernelGGL((ln_kernel<8, float>), grid, dim3(256), 0, s, (const float*)X, ldx, gamma, beta, (h16*)Y, ldy, rows, C, eps);
        else hipLaunchKernelGGL((ln_kernel<8, h16>), grid, dim3(256), 0, s, (const h16*)X, ldx, gamma, beta, (h16*)Y, ldy, rows, C, eps);
    }
    const int rc = mudg_check_launch("mudg_layernorm");
    mudg_prof_end(slot, s, 0.0, (double)rows * C * (x_fp32 == KIND_F32 ? 6.0 : 4.0));
    return rc;
}

extern "C" int mudg_softmax_rows(const float* S, int lds, void* P, int ldp, int rows, int cols, void* stream) {
    MUDG_REQUIRE(S && P && rows > 0 && cols > 0 && ldp % PLANES == 0 && ldp / PLANES >= cols, "mudg_softmax_rows: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int slot = mudg_prof_begin(MUDG_FAM_MISC, s);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, s, S, lds, (h16*)P, ldp, cols);
    const int rc = mudg_check_launch("mudg_softmax_rows");
    mudg_prof_end(slot, s, 0.0, (double)rows * cols * 6.0);
    return rc;
}
