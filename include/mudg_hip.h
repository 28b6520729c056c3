/*
 * mudg_hip.h — C-ABI of libmudg_hip.so: the MI355X (gfx950) kernels under MuDG's
 * video-diffusion denoising path (3D-UNet + DDIM update + AutoencoderKL decode).
 *
 * The reference (heiheishuang/MuDG) has no FFI layer: its hot path is PyTorch
 * module code whose arithmetic is dispatched implicitly to ATen/cuDNN/cuBLAS.
 * Each entry point below replaces one such implicit kernel class; the reference
 * call sites it stands in for are cited as file:line (relative to the reference
 * tree).  SURVEY.md §2.1 numbers the classes K1..K16.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     named host_*; the caller owns and allocates all buffers (outputs too);
 *   - activations are channels-last: a tensor (F, H, W, C) is a row-major
 *     matrix of F*H*W rows ("pixels"/"tokens") by C channels, bf16;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and returns immediately;
 *   - return value: 0 on success, a negative MUDG_E* code on error; nothing is
 *     thrown across the ABI and no call synchronises the device;
 *   - no hidden global state apart from the optional event profiler.
 */
#ifndef MUDG_HIP_H
#define MUDG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUDG_OK          0
#define MUDG_EINVAL     -1   /* bad argument (shape / alignment / null pointer) */
#define MUDG_ELAUNCH    -2   /* the HIP runtime rejected a launch */
#define MUDG_EUNSUPPORTED -3 /* shape outside what the kernels implement */

/* ABI version; bumped whenever a struct below changes. */
int mudg_version(void);
/* MFMA operand type of this build: 0 = bfloat16 (libmudg_hip.so), 1 = IEEE fp16 (libmudg_hip_fp16.so),
 * 2 = split bf16 x 2 planes (libmudg_hip_x3.so: 16 significand bits, 3 MFMAs per product tile),
 * 3 = split bf16 x 3 planes (libmudg_hip_x6.so: 24 significand bits, 6 MFMAs per product tile: fp32-class).
 * Wherever this header says "bf16" for an operand buffer, the fp16 build expects fp16 in its place, and the split builds
 * expect PLANES bf16 pieces per value: piece p of element (r, c) of a rows matrix with row stride ld lives at
 * r * ld + p * (ld / PLANES) + c, value = sum of the pieces, piece 0 = bf16(value), piece 1 = bf16(value - piece 0), ...
 * (so an operand matrix of width C needs ld >= PLANES * C, and ld a multiple of 8 * PLANES).  fp32 buffers (residual
 * stream, biases, statistics) are the same in every build. */
int mudg_operand_dtype(void);
/* Text for the most recent non-zero return on this thread. */
const char* mudg_last_error(void);

/* ------------------------------------------------------------------ GEMM / conv
 * Y[m][n] = epilogue( alpha * sum_k X[m][k] * W[n][k] )          (bf16 in, fp32 accumulate)
 *
 * One MFMA kernel serves every dense contraction on the path:
 *   mode 0  plain GEMM           nn.Linear / 1x1 Conv / Conv1d k=1
 *                                (attention.py:53-57,424,447,493,519,582,602; openaimodel3d.py:168-174,187;
 *                                 ae_modules.py:31-50 q/k/v/proj_out, :181 nin_shortcut; autoencoder.py:35)
 *   mode 1  3x3 conv, pad 1      implicit GEMM over 9 taps on NHWC input, stride 1|2, optional
 *                                nearest-2x upsample fused into the read
 *                                (openaimodel3d.py:66-70,96,103,154,179,401,564; ae_modules.py:117-127,162-175,499,535)
 *   mode 2  temporal (3,1,1) conv implicit GEMM over 3 taps along T on (B,T,H,W,C)
 *                                (openaimodel3d.py:255-266)
 * W is always [N][K] row-major with K = taps*Cin ordered tap-major (the host packs it once).
 * X may be split over two sources along channels (skip-concat, openaimodel3d.py:621):
 * channels [0,csplit) come from X, [csplit,Cin) from X2.
 */
typedef struct MudgGemmDesc {
    const void* X;        /* bf16 activations */
    const void* X2;       /* second channel source or NULL */
    const void* W;        /* bf16 packed weights [N][K] */
    void*       Y;        /* bf16 (or fp32 if out_fp32) [M][ldy] */
    const float* bias;    /* fp32 [N] or NULL (packed like W's rows) */
    const float* gbias;   /* fp32 [M/rows_per_group][Nout] or NULL: per-row-group bias
                             (timestep-embedding add, openaimodel3d.py:219-228) */
    const void* R;        /* residual [M][ldr] (bf16, or fp32 if res_fp32) or NULL, added last */
    int M, N, K;
    int ldx, ldx2, ldw, ldy, ldr;   /* row strides in elements */
    int csplit;           /* channels served by X (== Cin when X2 is NULL) */
    int batch;            /* blockIdx.z count (>=1) */
    int64_t sX, sW, sY, sR;         /* per-batch strides in elements */
    int rows_per_group;   /* for gbias; 0 = unused */
    int out_fp32;         /* storage of Y: 0 = MFMA operand (bf16 ...), 1 = fp32, 2 = IEEE fp16 */
    int res_fp32;         /* storage of R, same codes.  The residual stream (block outputs, the transformers' token stream,
                             encoder skips: what later layers add onto) is kept as fp16 by the 16-bit builds' callers — the
                             reference's own stream is fp16 under torch.autocast — and as fp32 by the split-operand builds' */
    int geglu;            /* 1: W rows are packed [32 value | 32 gate] blocks and
                             Y[m][j] = v_j * gelu_erf(g_j), Nout = N/2 (attention.py:579-586) */
    int act;              /* 1: Y = gelu_erf(.) applied after bias (Perceiver FeedForward, resampler.py:27-34); 0: none */
    float alpha;
    int mode;             /* 0 | 1 | 2 */
    /* mode 1 */
    int Hin, Win, Hout, Wout, Cin, stride, upsample;
    int pad;              /* mode 1: leading zero padding rows/cols (1 = the usual pad 1; 0 = AutoencoderKL's stride-2
                             downsample, which pads (0,1,0,1): only bottom/right, ae_modules.py:104-106) */
    int korder;           /* mode 1: 0 = W's K axis is [tap][Cin]; 1 = [Cin/64][tap][64] (needs Cin % 64 == 0): the nine
                             taps of a 64-channel slab are consecutive K tiles, so the shifted re-reads of the input
                             hit L2 instead of coming back after a whole sweep over Cin.
                             mode 2: 1 = the same slab-major order [Cin/64][tap][64]; needs T = 16, HW % 8 == 0, Cin % 64 == 0, one
                             source, the 16-bit or bf16x3 builds.  A 128-row tile is then 8 pixels x 16 frames of one clip (the
                             three temporal taps of a row are rows of the same tile), and `stats` blocks are those TILES, in
                             clip order: only a clip-level GroupNorm may fold them */
    /* mode 2 */
    int T, HW;            /* mode 0: HW may carry a HINT — the rows of one frame of the matrix (0 = none): it lets the library choose a tile
                             height that divides a frame (mudg_gemm_stats_rows).  Never M decides, so a row's result does not depend on
                             the batch it travels in; with a residual R the 288-row tile (and the 160-row tile for an fp32 R) adds R first instead of last (last-bit
                             differences against the un-hinted call), without one the bits are the same */
    float* stats;         /* NULL, or fp32 [ceil(M/rows)][Nout][2], rows = mudg_gemm_stats_rows(d) (128 | 160 | 288): the epilogue also writes, per row block and output
                             channel, the sum and the sum of squares of the values it stored (as stored: after rounding to
                             bf16 when Y is bf16) — the first pass of the GroupNorm that consumes Y
                             (mudg_groupnorm_fused).  Needs batch == 1 and no GEGLU. */
    int subpixel;         /* mode 1: 1 = the sub-pixel form of "nearest-2x upsample, then 3x3 / pad 1 conv" (openaimodel3d.py:92-106,
                             ae_modules.py:77-92).  An output pixel (2 oy + py, 2 ox + px) of the upsampled image only ever sees a
                             2 x 2 neighbourhood of the LOW-resolution input, its nine taps collapsing onto four summed weights:
                             4/9 of the multiply-adds.  X is the low-resolution image (Hout x Wout == Hin x Win is the grid
                             walked, M = frames * Hin * Win); batch = 4 with entry z = 2 py + px; W holds the four [N][4 Cin]
                             matrices sW apart, K ordered [Cin/64][tap 0..3][64] with tap = 2 a + b reading input pixel
                             (oy - 1 + py + a, ox - 1 + px + b); Y is the (2 Hin x 2 Win) image, row ((f 2Hin) + 2 oy + py) 2Win
                             + 2 ox + px.  Needs what mudg_conv_subpixel_ok checks. */
    void* Y8; void* S8;   /* 16-bit builds, optional: also write the OCP MX-fp8 copy of Y (an operand-kind result of a plain or conv
                             problem, N % 32 == 0, no GEGLU) — e4m3 bytes Y8[m][ldy8] and one E8M0 scale per 32 columns S8[m][lds8],
                             bit-equal to mudg_quantize_mxfp8 of Y: the q | k projection of the long self-attention quantises its own
                             output for the fp8 score path (BASELINE config 5) instead of two more passes over it */
    int ldy8, lds8;
} MudgGemmDesc;
int mudg_gemm(const MudgGemmDesc* d, void* stream);
/* Height of the row blocks in which mudg_gemm will write `stats` for this problem: 128, or 288 where the 288 x 320-tile kernel
 * runs it (same-size 3x3 convs, temporal convs with korder 0 and plain GEMMs with N % 320 == 0 whose frames are
 * whole 288-row tiles: Hout * Wout, HW — for mode 0 the caller's hint in HW — a multiple of 288), or 160 where the 160 x 320-tile
 * kernel of the 16-bit builds does (round 6: the same problems when a frame is whole 160-row tiles but not whole 288-row ones and
 * has at least 640 rows — MDM512's 2560- and 640-pixel frames; plain GEMMs there from K = 640, or without a residual).  `stats` then holds
 * fp32 [ceil(M / rows)][Nout][2], and the consumer (mudg_groupnorm_fused) is told the same height.  The answer depends on the
 * descriptor's geometry, strides and pointer alignment, never on M: a clip's results do not depend on the batch it travels in. */
int mudg_gemm_stats_rows(const MudgGemmDesc* d);
/* 1 when `d` (mode 1, subpixel = 1) can run: batch 4, stride 1, pad 1, korder 1, Cin % 64 == 0, no X2 / R / gbias / stats /
 * upsample, and offsets within reach of the buffer-descriptor loader; else the caller uses upsample = 1 with 3x3 weights. */
int mudg_conv_subpixel_ok(const MudgGemmDesc* d);

/* ------------------------------------------------------------------ attention
 * Flash-style softmax(scale * Q K^T) V for head dim 64, never materialising the score matrix.
 *   spatial self-attention  attention.py:81-144 (attn1, 393)     Nq = Nk = H*W
 *   text / image cross-attn attention.py:89-94,128-142 (attn2)   Nk = 77 / 16; `accumulate` adds
 *                                                                 the second softmax's output
 * Q : rows (f*Nq + i), head h at columns [h*64, h*64+64), row stride ldq
 * K : rows ((f / kv_div)*Nk + j), same column convention, row stride ldk
 * Vt: V transposed per (kv batch, head): row (h*64 + d), column j, row stride ldvt,
 *     batch stride svt elements (the producing GEMM writes it in this layout)
 * O : like Q with row stride ldo
 */
typedef struct MudgAttnDesc {
    const void* Q; const void* K; const void* Vt; void* O;
    int F, heads, Nq, Nk;
    int ldq, ldk, ldvt, ldo;
    int64_t svt;
    int kv_div;          /* frames sharing one K/V batch entry (T for text context, 1 otherwise) */
    float scale;         /* dim_head**-0.5 */
    int accumulate;      /* 1: O += result */
    /* Optional second key / value set with its OWN softmax, outputs summed — the text + image cross-attention of
     * attention.py:128-142 (out = softmax(q k_text^T) v_text + softmax(q k_ip^T) v_ip) in one launch: Q is read once and
     * O written once.  K2 == NULL: single set.  Same layout conventions as K / Vt. */
    const void* K2; const void* Vt2;
    int Nk2, ldk2, ldvt2, kv_div2;
    int64_t svt2;
    /* MX-fp8 scores (BASELINE config 5; needs q_prescaled, Nk % 64 == 0, Nq >= 512): when Q8 is not NULL, Q K^T runs on
     * v_mfma_scale_f32_32x32x64_f8f6f4 — twice the bf16 MFMA rate — from OCP e4m3 copies of Q and K with one E8M0 power-of-two
     * scale per 32 consecutive head dims (mudg_quantize_mxfp8 makes both); softmax and P V stay as in the bf16 kernel
     * (P and V^T are bf16).  Q8 / K8: fp8 rows with the row -> (frame, token) and column -> (head, dim) conventions of
     * Q / K, row strides ldq8 / ldk8 BYTES; Qs / Ks: scale bytes, column 2 * head + (dim / 32), row strides ldqs / ldks. */
    const void* Q8; const void* K8; const void* Qs; const void* Ks;
    int ldq8, ldk8, ldqs, ldks;
    int q_prescaled;     /* 1: Q already carries scale * log2(e) (folded into the q-projection weights when they are packed),
                            so Q K^T is directly the base-2 exponent and `scale` is ignored.  The long self-attention kernel
                            then runs its lean softmax: the running reference maximum enters as the score accumulator's
                            initial value, one v_exp + one add per score, and the rescale of O only happens when a row sum
                            outgrows 2^40 (never on real data after the first tile). */
    float* Lse;          /* optional, fp32 [F Nq][heads]: the log2-sum-exp of the scaled scores of every (query, head) — what
                            mudg_attention_bwd needs to rebuild P without a statistics pass (single key / value set, not
                            prescaled, 16-bit builds) */
} MudgAttnDesc;
int mudg_attention(const MudgAttnDesc* d, void* stream);
/* OCP microscaling quantisation of an operand matrix (16-bit builds): every 32 consecutive columns of a row share one
 * E8M0 scale 2^(floor(log2 amax) - 8) and are stored as e4m3 (saturating): Y8[r][c] (row stride ldy bytes),
 * S[r][c / 32] (row stride lds bytes).  cols % 32 == 0. */
int mudg_quantize_mxfp8(const void* X, int ldx, int64_t rows, int cols, void* Y8, int ldy, void* S, int lds, void* stream);

/* Temporal self-attention over T <= 32 frames per pixel (attention.py:529-576 via 81-144):
 * QKV rows are ((b*T + t)*HW + p); q at columns [h*64..], k at C + h*64, v at 2C + h*64. */
int mudg_temporal_attention(const void* QKV, void* O, int B, int T, int HW, int heads,
                            int ldqkv, int ldo, float scale, void* stream);

/* ------------------------------------------------------------------ normalisation
 * GroupNorm(32 groups) on channels-last data with optional fused SiLU / swish
 * (basics.py:76-87 GroupNormSpecific eps 1e-5; openaimodel3d.py:256-265; attention.py:420,488 eps 1e-6;
 *  ae_modules.py:10-16 eps 1e-6).  Statistics are per (sample, group) over `rows` pixels, where a
 * sample is one frame (2D norms) or one clip of T frames (the 5-D norms).  Two sources as in GEMM.
 * `ws` is fp32 scratch of at least mudg_groupnorm_ws_floats(samples, groups, rows).
 */
int64_t mudg_groupnorm_ws_floats(int samples, int groups, int rows);
int mudg_groupnorm(const void* X, const void* X2, int csplit, int ldx, int ldx2, int x_fp32,
                   const float* gamma, const float* beta, void* Y, int ldy,
                   int samples, int rows, int C, int groups, float eps, int silu,
                   float* ws, void* stream);
/* x_fp32: storage of X (and X2): 0 = operand (bf16 ...), 1 = fp32, 2 = IEEE fp16 (the residual stream of the 16-bit
 * builds); Y is always an MFMA operand (it feeds a GEMM).  Same codes in mudg_layernorm, mudg_cast_rows (src_fp32 /
 * dst_fp32) and mudg_rows_to_ncthw (src_is_fp32). */
/* Same normalisation with the statistics pass replaced by the per-(128-row block, channel) partial sums the producing
 * GEMM / conv wrote (MudgGemmDesc.stats): P1 covers X's csplit channels, P2 (NULL without X2) X2's C - csplit.
 * Needs rows % 128 == 0 (a row block never straddles two samples).  ws: fp32 scratch of 2 * samples * groups floats. */
int mudg_groupnorm_fused(const void* X, const void* X2, int csplit, int ldx, int ldx2, int x_fp32,
                         const float* gamma, const float* beta, void* Y, int ldy,
                         int samples, int rows, int C, int groups, float eps, int silu,
                         const float* P1, const float* P2, float* ws, void* stream);

/* The same with the height of the partial blocks stated per source (mudg_gemm_stats_rows of the producer: 128 | 160 | 288; `rows` must be
 * a multiple of both; mudg_groupnorm_fused = 128, 128). */
int mudg_groupnorm_fused_rows(const void* X, const void* X2, int csplit, int ldx, int ldx2, int x_fp32,
                              const float* gamma, const float* beta, void* Y, int ldy,
                              int samples, int rows, int C, int groups, float eps, int silu,
                              const float* P1, int p1_rows, const float* P2, int p2_rows, float* ws, void* stream);

/* LayerNorm over the last dim (attention.py:363-365, eps 1e-5). */
int mudg_layernorm(const void* X, int ldx, int x_fp32, const float* gamma, const float* beta,
                   void* Y, int ldy, int rows, int C, float eps, void* stream);

/* Row softmax of fp32 scores (already scaled) to bf16 probabilities
 * (ae_modules.py:64-68: the VAE's single-head d=512 attention). */
int mudg_softmax_rows(const float* S, int lds, void* P, int ldp, int rows, int cols, void* stream);

/* ------------------------------------------------------------------ embeddings / small linears
 * Sinusoidal embedding (utils_diffusion.py:8-28): out[i][:] = [cos(t_i f_j), sin(t_i f_j)], fp32.
 * freqs: device fp32 [dim/2] table exp(-ln(max_period) j / (dim/2)), built on the host as the reference does. */
int mudg_timestep_embedding(const int64_t* t, const float* freqs, float* out, int n, int dim, void* stream);
/* y[m][n] = act_out( sum_k act_in(x[m][k]) * W[n][k] + b[n] ), fp32 I/O, bf16 or fp32 weights, m <= 64
 * (time/class/fps MLPs openaimodel3d.py:377-397,569-602; ResBlock emb_layers 168-174). act: 0 none, 1 SiLU. */
int mudg_small_linear(const float* x, const void* W, int w_is_bf16, const float* b, float* y,
                      int M, int N, int K, int act_in, int act_out, int accumulate, void* stream);

/* ------------------------------------------------------------------ layout changes at the API boundary
 * (b c t h w) fp32/bf16 -> channels-last bf16 rows ((b t) h w) with channel offset/stride
 * (openaimodel3d.py:591; ddpm3d.py:1317-1319 channel concat of x and c_concat), and back (openaimodel3d.py:627). */
int mudg_ncthw_to_rows(const void* src, int src_is_fp32, void* dst, int B, int C, int T, int HW,
                       int ld, int coff, int Ttot, int t0, void* stream);
int mudg_rows_to_ncthw(const void* src, int src_is_fp32, int ld, int coff, void* dst, int dst_is_fp32,
                       int B, int C, int T, int HW, float scale, int Ttot, int t0, void* stream);
/* Ttot/t0: the (b c t h w) tensor holds Ttot frames and frames [t0, t0+T) are converted (Ttot <= 0: Ttot = T). */
/* Strided 2-D copy of bf16 rows (context token split, openaimodel3d.py:582-585) and y += alpha * x on fp32
 * vectors (summing the time / class / fps embeddings, openaimodel3d.py:576,602). */
int mudg_copy_rows(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, void* stream);
int mudg_axpy_f32(float* y, const float* x, int64_t n, float alpha, void* stream);
/* out[b][:] = ca[b] x[b][:] + cb[b] y[b][:] on fp32 (B, n): q_sample / predict_start_from_z_and_v /
 * predict_eps_from_z_and_v (ddpm3d.py:239-251) with ca, cb gathered per sample (device fp32 [B]). */
int mudg_lincomb(float* out, const float* x, const float* y, const float* ca, const float* cb, int B, int64_t n,
                 void* stream);
/* fp32 -> bf16 cast of n contiguous elements (a fp32 stream tensor entering an MFMA GEMM as an operand).
 * 16-bit-operand builds only (a flat cast has no plane layout); mudg_cast_rows serves every build. */
int mudg_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst[r][c] = src[r][c] for rows x cols elements between rows matrices of any kind, any direction: operand
 * (*_fp32 = 0: h16, or PLANES pieces per value in the split builds), fp32 (1) or fp16 (2).  Row strides in elements.
 * This is how fp32 tensors (context tokens ddpm3d.py:1320-1322, skip-conv inputs) become MFMA operands and how an
 * operand matrix is read back as fp32. */
int mudg_cast_rows(const void* src, int src_fp32, int64_t lds, void* dst, int dst_fp32, int64_t ldd, int64_t rows,
                   int64_t cols, void* stream);
/* Zero the channel range [c0, c1) of a rows buffer (padding lanes of the stem input). */
int mudg_zero_channels(void* dst, int rows, int ld, int c0, int c1, void* stream);

/* ------------------------------------------------------------------ DDIM update (ddim.py:205-279)
 * One call per step on fp32 latents, n elements per sample:
 *   v  = e_u + cfg*(e_c - e_u)                      two-way guidance, or with e_m (image-only conditioning) the
 *        e_u + cfg_img*(e_m - e_u) + cfg*(e_c - e_m) three-way form of ddim_multiplecond.py:226-233;
 *   v = phi*v*std(e_c)/std(v) + (1-phi)*v            (utils_diffusion.py:147-157)
 *   e  = sqrt_ac*v + sqrt_1mac*x ;  x0 = sqrt_ac*x - sqrt_1mac*v          (v-prediction, ddpm3d.py:239-251; eps_form = 0)
 *   e  = v ;                        x0 = (x - sqrt_1mac*e) / sqrt_ac      (eps-prediction, ddim.py:227-258; eps_form = 1,
 *                                                                          sqrt_ac = sqrt(a_t), sqrt_1mac = sqrt(1 - a_t))
 *   x0 *= rescale ;  x_prev = sqrt(a_prev)*x0 + dir_coef*e + sigma*noise
 * host_coef = {cfg, phi, sqrt_ac, sqrt_1mac, rescale, sqrt_a_prev, dir_coef, sigma, cfg_img, eps_form}
 *             (HOST pointer, 10 floats).
 * ws: fp64 device scratch of mudg_ddim_ws_doubles(B) doubles. e_u may be NULL (no guidance: v = e_c);
 * noise may be NULL (eta = 0). */
int64_t mudg_ddim_ws_doubles(int B);
int mudg_ddim_step(const float* x, const float* e_c, const float* e_u, const float* e_m, const float* noise,
                   float* x_prev, float* pred_x0, int B, int64_t n, const float* host_coef,
                   double* ws, void* stream);

/* ------------------------------------------------------------------ VAE posterior (distributions.py:24-40)
 * moments (N, 2C, HW) fp32 planar: mean = [:, :C], logvar = clamp([:, C:], -30, 20);
 * out (N, C, HW) = scale * (mean + exp(0.5 logvar) * noise); noise NULL = posterior mode. */
int mudg_gaussian_sample(const float* moments, const float* noise, float* out, int N, int C, int HW, float scale,
                         void* stream);

/* ------------------------------------------------------------------ post-processing of decoded frames
 * (virtual_render/eval_tools.py: byte / integer work, results bit-equal to the reference's)
 * frames_to_u8:     out[b][t][p][c] = (uint8) trunc((clamp(v[b][c][t][p], -1, 1) + 1) / 2 * 255)      eval_tools.py:22-27
 * depth_from_u8:    depth[i] = ((r + g + b) / 3) / 255 over `pixels` (.., 3)-interleaved uint8 pixels   eval_tools.py:71
 * semantic_nearest: planar (3, hw) uint8 image -> label of the nearest of the 19 palette colours (Euclidean distance,
 *                   first minimum wins) and the image recoloured with the palette                       eval_tools.py:309-347 */
int mudg_frames_to_u8(const float* video, uint8_t* out, int B, int C, int T, int64_t HW, void* stream);
int mudg_depth_from_u8(const uint8_t* frames, float* depth, int64_t pixels, void* stream);
int mudg_semantic_nearest(const uint8_t* img, uint8_t* vis, int64_t* labels, int64_t hw, void* stream);

/* ------------------------------------------------------------------ training step
 * Reference: lvdm/models/ddpm3d.py:741-802 (p_losses), :1267-1300 (configure_optimizers -> torch.optim.AdamW),
 * main/utils_train.py:126-137 (data-parallel strategy).  The contractions of the backward pass (dX = dY W, dW = dY^T X,
 * conv / temporal-conv input gradients) run on mudg_gemm; these entries are everything else the backward pass needs.
 * Gradients and the training stream are fp32 rows matrices (row strides in elements).  Fixed-order reductions throughout. */
/* dst[c][p] = src[srcrow(p)][c] as an MFMA operand matrix [C][ldd] (ldd / PLANES >= P rounded up to 8; the tail columns are
 * zeroed): the transposed (and, for convolutions, tap-shifted) operand copies that weight-gradient GEMMs contract over.
 * mode 0: srcrow = p.  mode 1 (3x3 conv tap (dy, dx)): p = (f, oy, ox) on the Hout x Wout grid reads input pixel
 * (oy stride - pad + dy, ox stride - pad + dx) of the Hin x Win image, zero outside.  mode 2 (temporal tap dt): p = ((b T + t) HW + s)
 * reads row p + (dt - 1) HW while 0 <= t + dt - 1 < T. */
int mudg_transpose_gather(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t P, int C, int mode, int Hin, int Win,
                          int Hout, int Wout, int stride, int pad, int dy, int dx, int T, int HW, int dt, int batch,
                          int64_t src_batch_stride, int64_t dst_batch_stride, void* stream);
/* batch > 1: entry z reads src + z * src_batch_stride and writes dst + z * dst_batch_stride (elements): the per-(frame, head)
 * transposes of the attention backward pass in one launch. */
/* out[g][c] = sum over the rows of group g (rows_per_group consecutive rows) of A[r][c] * (B ? B[r][c] : 1): bias gradients
 * (one group), the timestep-embedding gradient of a ResBlock (one group per clip). */
int mudg_group_colsum(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t rows, int cols, int64_t rows_per_group,
                      float* out, void* stream);
/* GroupNorm (+SiLU) backward.  stat = (mean, rstd) per (sample, group) (mudg_groupnorm_stats of the same X); writes dX and
 * AB[sample][c] = (sum dz, sum dz xhat): dbeta / dgamma are its sums over the samples.  ws: mudg_groupnorm_bwd_ws_floats. */
int64_t mudg_groupnorm_bwd_ws_floats(int samples, int rows, int C, int groups);
int mudg_groupnorm_stats(const float* X, int64_t ldx, int samples, int rows, int C, int groups, float eps, float* stat, void* stream);
int mudg_groupnorm_bwd(const float* X, int64_t ldx, const float* dY, int64_t ldy, const float* gamma, const float* beta,
                       const float* stat, int samples, int rows, int C, int groups, int silu, float* dX, int64_t lddx, float* AB,
                       float* ws, const float* dres, int64_t lddres, void* stream);
/* LayerNorm backward: dX, and part[chunks][2][C] (chunks = mudg_layernorm_bwd_chunks(rows)): per chunk of rows the sums of
 * dY * xhat (row 0) and dY (row 1) — dgamma / dbeta are their sums over the chunks (mudg_group_colsum).  C <= 1280.  dres (optional,
 * rows [rows][lddres]): added to dX — the gradient of the residual branch that bypassed the norm (x + f(LN(x))). */
int64_t mudg_layernorm_bwd_chunks(int64_t rows);
int mudg_layernorm_bwd(const float* X, int64_t ldx, const float* dY, int64_t ldy, const float* gamma, float* dX, int64_t lddx,
                       float* part, int64_t rows, int C, float eps, const float* dres, int64_t lddres, void* stream);
/* Weight gradient of a linear / 3x3 conv / temporal conv layer straight from the row-major operands of the forward pass (16-bit
 * operand builds): out[slice][m][tap * C + c] = sum over the positions p of the slice of A[p][m] * B[src(p, tap)][c], with A = the
 * output gradient as operand rows [P][lda], B = the layer input as operand rows [*][ldb], src as in mudg_transpose_gather (mode 0:
 * identity, taps = 1; mode 1: 3x3 taps, tap = 3 dy + dx, taps = 9; mode 2: temporal taps, taps = 3).  M % 8 == 0, C % 64 == 0.
 * The positions are cut into `slices` ranges of `chunk` (a multiple of 64) each; the caller adds the slabs (mudg_group_colsum). */
typedef struct MudgWgradDesc {
    const void* A; const void* B; float* out;
    int64_t lda, ldb, P;
    int M, C, taps, mode;
    int Hin, Win, Hout, Wout, stride, pad, T, HW;
    int slices; int64_t chunk;
} MudgWgradDesc;
int mudg_wgrad(const MudgWgradDesc* d, void* stream);
/* One pass over fp32 rows src[P][C] (C % 4 == 0) for the three forms a layer's output gradient is needed in: dst[c][p] = src[p][c]
 * as an operand matrix [C][ldd] (columns P .. P rounded up to 8 zero; optional); rows (optional): the operand-row copy [P][ldr];
 * part (optional): fp32 [ceil(P / 64)][C], the column sums of each 64-row tile (the bias gradient is their sum). */
int mudg_transpose_cast_sum(const float* src, int64_t lds, void* dst, int64_t ldd, void* rows, int64_t ldr, float* part, int64_t P, int C,
                            void* stream);
/* GEGLU on H = [value | gate] rows [M][2 N] (attention.py:579-586): dY NULL -> out[M][N] = value * gelu(gate);
 * else out = dH [M][2 N]. */
int mudg_geglu(const float* H, int64_t ldh, const float* dY, int64_t lddy, float* out, int64_t ldo, int64_t M, int N, void* stream);
/* GEGLU followed by the feed-forward's Dropout(p) in one pass (attention.py:579-606; N % 4 == 0).  dY NULL: out [M][N] = keep(value *
 * gelu(gate)) / (1 - p) as fp32 rows and, when out16 is given, as operand rows [M][ldo16] for the next GEMM.  dY given: out = dH
 * [M][2 N].  The keep mask depends on (seed, output element index) only and is regenerated in the backward pass; p = 0: no dropout. */
int mudg_geglu_dropout(const float* H, int64_t ldh, const float* dY, int64_t lddy, float* out, int64_t ldo, void* out16, int64_t ldo16,
                       int64_t M, int N, float p, uint64_t seed, void* stream);
/* Row softmax in fp32 and its backward dS = scale * P (dP - sum_j dP_j P_j): the recomputed probabilities of the attention
 * backward pass. */
int mudg_softmax_f32(const float* S, int64_t lds, float* P, int64_t ldp, int64_t rows, int cols, void* stream);
int mudg_softmax_bwd(const float* P, int64_t ldp, const float* dP, int64_t lddp, float* dS, int64_t ldds, int64_t rows, int cols,
                     float scale, void* stream);
/* Backward of mudg_temporal_attention on fp32 Q / K / V / dO rows ((b T + t) HW + s), head h at columns [64 h, 64 h + 64). */
int mudg_temporal_attention_bwd(const float* Q, const float* K, const float* V, const float* dO, int64_t ldqkv, int64_t ldo,
                                float* dQ, float* dK, float* dV, int64_t ldg, int B, int T, int HW, int heads, float scale,
                                void* stream);
/* Backward of softmax(scale Q K^T) V, head width 64, without materialising the scores (16-bit operand builds; the forward is
 * mudg_attention, reference attention.py:81-144).  Q, dO: operand rows [F Nq][ld*]; K, V: operand rows [(F / kv_div) Nk][ld*]
 * (kv_div frames share a key / value batch: the text tokens of a clip); head h at columns [64 h, 64 h + 64).  No transposed
 * copies are needed (the kernels read K^T, Q^T, dO^T out of the row tiles with ds_read_b64_tr_b16).  L, D: fp32 [F Nq][heads]:
 * the log2-sum-exp of the scaled scores and sum_k P dP — filled by the call, or, with O given, L read (MudgAttnDesc.Lse of the
 * forward pass) and D taken from dO . O.  dQ [F Nq][ldgq], dK, dV [(F / kv_div) Nk][ldgk]: fp32, every element of the head
 * columns written once. */
typedef struct MudgAttnBwdDesc {
    const void* Q; const void* K; const void* V; const void* dO;
    const void* O;       /* optional: the forward output (operand rows, row stride ldo).  Not NULL: L holds what the forward pass
                            wrote (MudgAttnDesc.Lse) and D = sum_d dO O is taken from O — the statistics pass is skipped */
    float* L; float* D;
    float* dQ; float* dK; float* dV;
    int F, heads, Nq, Nk, kv_div;
    int ldq, ldk, ldv, lddo, ldo;
    int64_t ldgq, ldgk;
    float scale;
} MudgAttnBwdDesc;
int mudg_attention_bwd(const MudgAttnBwdDesc* d, void* stream);
/* loss[b] = mean over the n elements of sample b of (pred - target)^2; grad (optional) = w[b] * 2 (pred - target) / n — the
 * gradient of sum_b w[b] loss[b] (ddpm3d.py:766-787 folds logvar, l_simple_weight and the vlb term into w). */
int64_t mudg_mse_ws_doubles(int B);
int mudg_mse(const float* pred, const float* target, const float* w, int B, int64_t n, float* loss, float* grad, double* ws,
             void* stream);
/* Nearest-2x of channels-last fp32 rows (F, h, w, C) -> (F, 2h, 2w, C) (adjoint = 1: the sum over each 2 x 2 block, its
 * backward), and zero insertion of a stride-2 conv's output gradient onto the input grid (its input gradient is then a
 * stride-1 conv with the flipped filter). */
int mudg_upsample2x(const float* src, float* dst, int F, int h, int w, int C, int adjoint, void* stream);
int mudg_dilate2x(const float* src, float* dst, int F, int ho, int wo, int hi, int wi, int C, void* stream);
/* torch.optim.AdamW step (decoupled weight decay, bias correction with `step` >= 1), fp32, in place. */
int mudg_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, void* stream);
/* The same step over many parameter tensors in ONE launch (the UNet has 1520): table = device int64 [nchunks][5] rows
 * (p, g, m, v addresses, count <= mudg_clip_chunk()) covering every parameter; bit-identical to mudg_adamw per element. */
int mudg_adamw_multi(const int64_t* table, int nchunks, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                     void* stream);
/* torch.nn.utils.clip_grad_norm_ over many fp32 tensors with no host round trip (the reference's trainer: gradient_clip_val 0.5,
 * norm).  table: device int64 [nchunks][2] = (address, count <= mudg_clip_chunk()) covering every gradient; partial: fp64 [nchunks]
 * scratch; out: float [2] = (total 2-norm, clip coefficient min(1, max_norm / (norm + 1e-6))); the gradients are scaled in place. */
int mudg_clip_chunk(void);
int mudg_clip_grad_norm(const int64_t* table, int nchunks, double* partial, float max_norm, float* out, void* stream);
/* Inverted dropout out[i] = keep(seed, i) ? x[i] / (1 - p) : 0 with a counter-based mask (the backward pass applies the same
 * call to the gradient: nothing is stored). */
int mudg_dropout(const float* x, float* out, int64_t n, float p, uint64_t seed, void* stream);
/* The same mask (element index = m C + c) on fp32 rows [M][C] (C % 4 == 0) with any row strides, optionally also writing the
 * operand rows of the result (what the conv / projection after the Dropout reads). */
int mudg_dropout_rows(const float* X, int64_t ldx, float* Y, int64_t ldy, void* Y16, int64_t ldy16, int64_t M, int C, float p, uint64_t seed,
                      void* stream);
/* out = gelu_erf(x) (dy NULL) or dy * gelu_erf'(x): the Perceiver Resampler's feed-forward activation (resampler.py:27-34). */
int mudg_gelu(const float* x, const float* dy, float* out, int64_t n, void* stream);
/* out = silu(x) (dy NULL) or dy * silu'(x). */
int mudg_silu(const float* x, const float* dy, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------ event profiler (bench.py roofline)
 * When enabled, every launch of kernel family `fam` is bracketed by hipEvents on its own stream. */
enum { MUDG_FAM_GEMM = 0, MUDG_FAM_CONV = 1, MUDG_FAM_TCONV = 2, MUDG_FAM_ATTN = 3, MUDG_FAM_TATTN = 4,
       MUDG_FAM_GNORM = 5, MUDG_FAM_LNORM = 6, MUDG_FAM_MISC = 7, MUDG_FAM_COUNT = 8 };
int mudg_prof_enable(int family_mask);       /* 0 disables and drops pending events */
/* Synchronises the recorded events and returns totals since the last reset. */
int mudg_prof_collect(int fam, double* total_ms, int64_t* launches, double* flops, double* bytes);
int mudg_prof_reset(void);

#ifdef __cplusplus
}
#endif
#endif
