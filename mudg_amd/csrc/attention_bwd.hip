// attention_bwd.hip — backward of softmax(scale Q K^T) V for head width 64 without materialising the scores (training step,
// SURVEY §8 f4; forward: attention.hip, reference attention.py:81-144).  16-bit operand builds.
//
// With P = softmax(S), S = scale Q K^T, O = P V and D_q = sum_k P[q][k] dP[q][k] (= sum_d dO[q][d] O[q][d]):
//     dV = P^T dO,   dP = dO V^T,   dS = P (dP - D) scale,   dQ = dS K,   dK = dS^T Q.
// Three passes, all in the forward kernel's idiom (32 rows per wave as MFMA B-operand fragments in registers, the other side
// streamed through LDS in 64-row tiles, scores left in registers in the order the next MFMA contracts over):
//   stats   (skipped when the forward pass saved L and O: then D = sum_d dO O)  per query: L = log2 sum_k 2^(c s_k) (so that P = 2^(c s - L), c = scale log2 e) and D = sum_k P_k dP_k (equal to
//           sum_d dO O of THIS key / value set — taken from P so that the two-set cross-attention needs no per-set output);
//   Q side  a wave owns 32 QUERIES (fragments of q and dO): per key tile  S^T = K q^T, dP^T = V dO^T (lane = query, registers =
//           keys), P, dS in registers, then dQ^T += K^T dS^T exactly as the forward accumulates O^T += V^T P^T;
//   K side  the mirror image — a wave owns 32 KEYS (fragments of k and v): per query tile  S = Q k^T, dP = dO v^T (lane = key,
//           registers = queries), L and D of the tile's queries from LDS, then dV^T += dO^T P and dK^T += Q^T dS.
// The transposed operands (K^T for the Q side; dO^T, Q^T for the K side) are never made: the row-major tiles already in LDS for the
// score products are read a second time through ds_read_b64_tr_b16 (accumulate_tr).  Each pass keeps the next tile's global loads in
// flight in registers under the current tile's MFMAs.  Key / value batches shared by kv_div frames (the text tokens of the
// cross-attention) are handled by the K side walking all kv_div * Nq query rows of its batch.  No atomics: every output
// element is written by exactly one wave.
#include "common.h"

#if MUDG_PLANES == 1
namespace {

constexpr int BTQ = 128;    // rows (queries or keys) per workgroup: 4 waves x 32
constexpr int BT = 64;      // streamed rows per tile
constexpr int BLD = 72;     // LDS row stride in h16 (64 + 8 pad)
constexpr int BTILE = 64 * BLD;

// 64 x 64 h16 tile of a row-major matrix (rows r0 .. r0 + 63, 64 columns from `base`), rows >= nrows zero: two 16-byte pieces per thread
__device__ __forceinline__ void load_rows(const h16* base, int64_t ld, int64_t r0, int64_t nrows, int tid, u32x4 (&r)[2]) {
    const int lrow = tid >> 3, kc = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t row = r0 + lrow + 32 * i;
        r[i] = row < nrows ? ld16(base + row * ld + kc * 8) : zero16();
    }
}
__device__ __forceinline__ void stage(h16* tile, int tid, const u32x4 (&r)[2]) {
    const int lrow = tid >> 3, kc = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) st16(&tile[(lrow + 32 * i) * BLD + kc * 8], r[i]);
}
// scores of one 32-row sub-tile of a staged row-major tile against this lane's row fragments: acc[streamed row][own row]
__device__ __forceinline__ void scores(const h16* tile, int sub, int l31, int hi, const h16x8 (&own)[4], f32x16& acc) {
    const h16* p = tile + (sub * 32 + l31) * BLD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = MFMA_32x32x16(*reinterpret_cast<const h16x8*>(p + ks * 16), own[ks], acc);
}
// out^T[d][own row] += T^T[d][streamed row] * packed[streamed row][own row] for both 32-wide halves of d, with T^T taken from
// the ROW-major tile T[streamed row][d] through gfx950's transposing LDS read: a 16-lane
// group reads a 4-row x 16-column block and each lane receives one column's 4 rows — here the four streamed rows
// kk + 4 hi + 0..3 (then + 8) of head channel dt 32 + (lane & 31), exactly the operand order `packed` was built in.  No transposed
// copy of the streamed matrix exists anywhere.  (With the 144-byte row stride the two 16-lane groups of a half wave collide on
// half of their banks: 2 LDS cycles more per read, against 32 MFMA cycles per read.)
#ifdef MUDG_OPERAND_FP16
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 tr_vec_t;
#define TR_READ(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4f16(ptr)
#else
typedef h16x4 tr_vec_t;
#define TR_READ(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4bf16(ptr)
#endif
typedef __attribute__((address_space(3))) tr_vec_t* lds_tr_ptr;
__device__ __forceinline__ h16x4 tr_read(const h16* p) {
    const tr_vec_t v = TR_READ((lds_tr_ptr)p);
    h16x4 o;
    __builtin_memcpy(&o, &v, sizeof(o));
    return o;
}
__device__ __forceinline__ void accumulate_tr(const h16* tile, int lane, const h16x8 (&pk)[2][2], f32x16 (&out)[2]) {
    const int g = lane >> 4, q = lane & 15;
    const h16* base = tile + (4 * (g >> 1) + (q >> 2)) * BLD + 16 * (g & 1) + 4 * (q & 3);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const h16* p = base + (sub * 32 + jj * 16) * BLD + dt * 32;
                const h16x4 lo = tr_read(p), up = tr_read(p + 8 * BLD);
                h16x8 f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = up[e]; }
                out[dt] = MFMA_32x32x16(f, pk[sub][jj], out[dt]);
            }
}
// fp32 store of an accumulator pair: lane holds, for its row, columns dt*32 + 8g + 4 hi + {0..3}
__device__ __forceinline__ void store_rows(float* row, int hi, const f32x16 (&o)[2]) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = o[dt][4 * g + j];
            *reinterpret_cast<f32x4*>(row + dt * 32 + 8 * g + 4 * hi) = v;
        }
}
__device__ __forceinline__ void own_fragments(const h16* row, bool ok, int hi, h16x8 (&f)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = as_h16x8(ok ? ld16(row + ks * 16 + hi * 8) : zero16());
}

struct Geo {
    int w, f, h, g, t;       // work item, frame, head, key / value batch, tile index inside
};
__device__ __forceinline__ int xcd_item(int total) {
    const int q8 = total >> 3, r8 = total & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
}

// ---------------------------------------------------------------------------------------------- stats
__global__ __launch_bounds__(256, 2) void attn_bwd_stats_kernel(const MudgAttnBwdDesc p, const int nqt, const int total) {
    __shared__ __attribute__((aligned(16))) h16 Ks[BTILE], Vs[BTILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int w = xcd_item(total);
    const int pair = w / nqt, qt = w - pair * nqt;
    const int f = pair / p.heads, h = pair - f * p.heads;
    const int q = qt * BTQ + wave * 32 + l31;
    const bool qok = q < p.Nq;
    const int64_t qrow = (int64_t)f * p.Nq + q;
    const h16* Kp = reinterpret_cast<const h16*>(p.K) + (int64_t)(f / p.kv_div) * p.Nk * p.ldk + h * 64;
    const h16* Vp = reinterpret_cast<const h16*>(p.V) + (int64_t)(f / p.kv_div) * p.Nk * p.ldv + h * 64;
    h16x8 qf[4], dof[4];
    own_fragments(reinterpret_cast<const h16*>(p.Q) + qrow * p.ldq + h * 64, qok, hi, qf);
    own_fragments(reinterpret_cast<const h16*>(p.dO) + qrow * p.lddo + h * 64, qok, hi, dof);
    const float c = p.scale * 1.4426950408889634f;
    // online over the key tiles, per lane over the keys its registers hold: l = sum 2^(c (s - m)), d = sum 2^(c (s - m)) dP
    float m_run = -INFINITY, l_run = 0.f, d_run = 0.f;
    const int nkt = (p.Nk + BT - 1) / BT;
    for (int kt = 0; kt < nkt; ++kt) {
        u32x4 rk[2], rv[2];
        load_rows(Kp, p.ldk, (int64_t)kt * BT, p.Nk, tid, rk);
        load_rows(Vp, p.ldv, (int64_t)kt * BT, p.Nk, tid, rv);
        __syncthreads();
        stage(Ks, tid, rk); stage(Vs, tid, rv);
        __syncthreads();
        f32x16 s[2], dp[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[sub][i] = 0.f; dp[sub][i] = 0.f; }
            scores(Ks, sub, l31, hi, qf, s[sub]);
            scores(Vs, sub, l31, hi, dof, dp[sub]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = kt * BT + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
                if (j >= p.Nk) s[sub][i] = -INFINITY;
                mx = fmaxf(mx, s[sub][i]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        float ps = 0.f, pd = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f((s[sub][i] - m_new) * c);
                ps += e;
                pd = fmaf(e, dp[sub][i], pd);
            }
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        l_run = l_run * alpha + ps;
        d_run = d_run * alpha + pd;
        m_run = m_new;
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float d_tot = d_run + __shfl_xor(d_run, 32, 64);
    if (qok && hi == 0) {
        p.L[qrow * p.heads + h] = m_run * c + __log2f(l_tot);
        p.D[qrow * p.heads + h] = d_tot / l_tot;
    }
}

// ---------------------------------------------------------------------------------------------- Q side: dQ
__global__ __launch_bounds__(256, 2) void attn_bwd_q_kernel(const MudgAttnBwdDesc p, const int nqt, const int total) {
    __shared__ __attribute__((aligned(16))) h16 Ks[BTILE], Vs[BTILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int w = xcd_item(total);
    const int pair = w / nqt, qt = w - pair * nqt;
    const int f = pair / p.heads, h = pair - f * p.heads;
    const int g = f / p.kv_div;
    const int q = qt * BTQ + wave * 32 + l31;
    const bool qok = q < p.Nq;
    const int64_t qrow = (int64_t)f * p.Nq + q;
    const h16* Kp = reinterpret_cast<const h16*>(p.K) + (int64_t)g * p.Nk * p.ldk + h * 64;
    const h16* Vp = reinterpret_cast<const h16*>(p.V) + (int64_t)g * p.Nk * p.ldv + h * 64;
    h16x8 qf[4], dof[4];
    own_fragments(reinterpret_cast<const h16*>(p.Q) + qrow * p.ldq + h * 64, qok, hi, qf);
    own_fragments(reinterpret_cast<const h16*>(p.dO) + qrow * p.lddo + h * 64, qok, hi, dof);
    const float Lq = qok ? p.L[qrow * p.heads + h] : 0.f;
    float Dq;
    if (p.O) {                                     // the forward pass left L and O: D = sum_d dO O, kept for the key side
        h16x8 of[4];
        own_fragments(reinterpret_cast<const h16*>(p.O) + qrow * p.ldo + h * 64, qok, hi, of);
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum = fmaf((float)dof[ks][e], (float)of[ks][e], dsum);
        Dq = dsum + __shfl_xor(dsum, 32, 64);
        if (qok && hi == 0) p.D[qrow * p.heads + h] = Dq;
    } else {
        Dq = qok ? p.D[qrow * p.heads + h] : 0.f;
    }
    const float c = p.scale * 1.4426950408889634f;
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
    const int nkt = (p.Nk + BT - 1) / BT;
    u32x4 rk[2], rv[2];
    load_rows(Kp, p.ldk, 0, p.Nk, tid, rk);
    load_rows(Vp, p.ldv, 0, p.Nk, tid, rv);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                           // the previous tile's reads are done
        stage(Ks, tid, rk); stage(Vs, tid, rv);
        __syncthreads();
        if (kt + 1 < nkt) {                        // the next tile's rows travel while this one is multiplied
            load_rows(Kp, p.ldk, (int64_t)(kt + 1) * BT, p.Nk, tid, rk);
            load_rows(Vp, p.ldv, (int64_t)(kt + 1) * BT, p.Nk, tid, rv);
        }
        h16x8 pk[2][2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
            scores(Ks, sub, l31, hi, qf, s);       // S^T[key][query]
            scores(Vs, sub, l31, hi, dof, dp);     // dP^T[key][query]
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = kt * BT + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
                const float pr = j < p.Nk ? __builtin_amdgcn_exp2f(fmaf(s[i], c, -Lq)) : 0.f;
                pk[sub][i >> 3][i & 7] = (h16)(pr * (dp[i] - Dq) * p.scale);
            }
        }
        accumulate_tr(Ks, lane, pk, dq);           // dQ^T[d][query] += K^T[d][key] dS^T[key][query], K^T read out of the K rows
    }
    if (qok) store_rows(p.dQ + qrow * p.ldgq + h * 64, hi, dq);
}

// ---------------------------------------------------------------------------------------------- K side: dK, dV
__global__ __launch_bounds__(256, 2) void attn_bwd_k_kernel(const MudgAttnBwdDesc p, const int nktile, const int total) {
    __shared__ __attribute__((aligned(16))) h16 Qs[BTILE], dOs[BTILE];
    __shared__ __attribute__((aligned(16))) float Ls[BT], Ds[BT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int w = xcd_item(total);
    const int pair = w / nktile, ktile = w - pair * nktile;
    const int g = pair / p.heads, h = pair - g * p.heads;
    const int k = ktile * BTQ + wave * 32 + l31;
    const bool kok = k < p.Nk;
    const int64_t krow = (int64_t)g * p.Nk + k;
    const int64_t NQ = (int64_t)p.kv_div * p.Nq;                 // query rows served by this key / value batch
    const int64_t q0 = (int64_t)g * NQ;
    const h16* Qp = reinterpret_cast<const h16*>(p.Q) + q0 * p.ldq + h * 64;
    const h16* dOp = reinterpret_cast<const h16*>(p.dO) + q0 * p.lddo + h * 64;
    h16x8 kf[4], vf[4];
    own_fragments(reinterpret_cast<const h16*>(p.K) + krow * p.ldk + h * 64, kok, hi, kf);
    own_fragments(reinterpret_cast<const h16*>(p.V) + krow * p.ldv + h * 64, kok, hi, vf);
    const float c = p.scale * 1.4426950408889634f;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
    const int64_t nqt = (NQ + BT - 1) / BT;
    u32x4 rq[2], rd[2];
    float lq = INFINITY, dq_ = 0.f;                                // a query row that does not exist: P = 2^(-inf) = 0
    auto fetch = [&](int64_t qt) {
        load_rows(Qp, p.ldq, qt * BT, NQ, tid, rq);
        load_rows(dOp, p.lddo, qt * BT, NQ, tid, rd);
        lq = INFINITY; dq_ = 0.f;
        if (tid < BT && qt * BT + tid < NQ) {
            lq = p.L[(q0 + qt * BT + tid) * p.heads + h];
            dq_ = p.D[(q0 + qt * BT + tid) * p.heads + h];
        }
    };
    fetch(0);
    for (int64_t qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        stage(Qs, tid, rq); stage(dOs, tid, rd);
        if (tid < BT) { Ls[tid] = lq; Ds[tid] = dq_; }
        __syncthreads();
        if (qt + 1 < nqt) fetch(qt + 1);                           // the next tile travels while this one is multiplied
        h16x8 pp[2][2], pds[2][2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
            scores(Qs, sub, l31, hi, kf, s);       // S[query][key]: lane = key, registers = queries 32 sub + 8 g + 4 hi + j
            scores(dOs, sub, l31, hi, vf, dp);     // dP[query][key]
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[sub * 32 + 8 * gq + 4 * hi]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&Ds[sub * 32 + 8 * gq + 4 * hi]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * gq + j;
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[i], c, -l4[j]));
                    pp[sub][i >> 3][i & 7] = (h16)pr;
                    pds[sub][i >> 3][i & 7] = (h16)(pr * (dp[i] - d4[j]) * p.scale);
                }
            }
        }
        accumulate_tr(dOs, lane, pp, dv);          // dV^T[d][key] += dO^T[d][query] P[query][key], dO^T read out of the dO rows
        accumulate_tr(Qs, lane, pds, dk);          // dK^T[d][key] += Q^T[d][query] dS[query][key]
    }
    if (kok) {
        store_rows(p.dK + krow * p.ldgk + h * 64, hi, dk);
        store_rows(p.dV + krow * p.ldgk + h * 64, hi, dv);
    }
}

}  // namespace
#endif

extern "C" int mudg_attention_bwd(const MudgAttnBwdDesc* dp, void* stream) {
#if MUDG_PLANES == 1
    MUDG_REQUIRE(dp, "mudg_attention_bwd: null descriptor");
    const MudgAttnBwdDesc d = *dp;
    MUDG_REQUIRE(d.Q && d.K && d.V && d.dO && d.L && d.D && d.dQ && d.dK && d.dV, "mudg_attention_bwd: null pointer");
    MUDG_REQUIRE(d.F > 0 && d.heads > 0 && d.Nq > 0 && d.Nk > 0 && d.kv_div > 0 && d.F % d.kv_div == 0, "mudg_attention_bwd: geometry");
    const int C = d.heads * 64;
    MUDG_REQUIRE(d.ldq >= C && d.ldk >= C && d.ldv >= C && d.lddo >= C && d.ldgq >= C && d.ldgk >= C, "mudg_attention_bwd: row strides");
    MUDG_REQUIRE((d.ldq & 7) == 0 && (d.ldk & 7) == 0 && (d.ldv & 7) == 0 && (d.lddo & 7) == 0 &&
                 (d.ldgq & 3) == 0 && (d.ldgk & 3) == 0, "mudg_attention_bwd: strides must keep 16-byte accesses aligned");
    MUDG_REQUIRE(aligned16(d.Q) && aligned16(d.K) && aligned16(d.V) && aligned16(d.dO) &&
                 aligned16(d.dQ) && aligned16(d.dK) && aligned16(d.dV), "mudg_attention_bwd: alignment");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nqt = (d.Nq + BTQ - 1) / BTQ;
    const int64_t tq = (int64_t)nqt * d.F * d.heads;
    const int nkt = (d.Nk + BTQ - 1) / BTQ;
    const int64_t tk = (int64_t)nkt * (d.F / d.kv_div) * d.heads;
    MUDG_REQUIRE(tq < (1ll << 31) && tk < (1ll << 31), "mudg_attention_bwd: grid too large");
    if (d.O) MUDG_REQUIRE(d.ldo >= C && (d.ldo & 7) == 0 && aligned16(d.O), "mudg_attention_bwd: O stride / alignment");
    else hipLaunchKernelGGL(attn_bwd_stats_kernel, dim3((unsigned)tq), dim3(256), 0, s, d, nqt, (int)tq);
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3((unsigned)tq), dim3(256), 0, s, d, nqt, (int)tq);
    hipLaunchKernelGGL(attn_bwd_k_kernel, dim3((unsigned)tk), dim3(256), 0, s, d, nkt, (int)tk);
    return mudg_check_launch("mudg_attention_bwd");
#else
    (void)dp; (void)stream;
    MUDG_FAIL(MUDG_EUNSUPPORTED, "mudg_attention_bwd: the fused backward belongs to the 16-bit operand builds (the split builds recompute P explicitly)");
#endif
}
