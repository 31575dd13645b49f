/*
 * vda.h — C ABI of libvda_hip.so, the MI355X (gfx950) kernel library behind
 * VideoDepthAnything.forward / infer_video_depth.
 *
 * The reference has no FFI boundary: it is pure Python/PyTorch and every entry
 * point below replaces an ATen / xformers op site on its hot path
 * (SURVEY.md §2.2, K1-K20). The reference call site each one stands in for is
 * cited per function, paths relative to the reference's video_depth_anything/.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    the name says host; no torch types, no ownership transfer;
 *  - every function enqueues on `stream` (a hipStream_t passed as void*) and
 *    returns immediately: 0 = ok, non-zero = refused (bad shape/alignment; the
 *    text is available from vda_last_error()); nothing is launched on refusal;
 *  - activations are token-major / NHWC; the *_f16 entry points take fp16 ("h") activations and fp16 MFMA operands with
 *    fp32 accumulation (the reference's autocast path, video_depth.py:203-205 with fp32=False); the *_f32 twins take
 *    fp32 activations and fp32 MFMA operands (exact fp32 products: the reference's fp32=True path, run.py:31);
 *  - the per-kernel entry points are not thread-safe per stream and do no allocation or synchronisation inside
 *    (graph-capturable); the handle API at the end of this file owns memory and says where it allocates.
 */
#ifndef VDA_H
#define VDA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vda_stream_t;

const char* vda_last_error(void);
int vda_abi_version(void);

/* ---------------------------------------------------------------- GEMM / conv
 * out = epilogue( A[M,K] * W[N,K]^T ), fp16 operands, fp32 accumulate (MFMA).
 * Replaces nn.Linear / 1x1 Conv2d / 3x3 Conv2d / ConvTranspose2d(k==stride):
 *   dinov2_layers/attention.py:51,60   dinov2_layers/mlp.py:36-40
 *   dinov2_layers/patch_embed.py:76    dpt.py:60-90,117-121   util/blocks.py:20-32,79-84,160
 *   motion_module/motion_module.py:113,120,242-250,290   motion_module/attention.py:333,383
 */
enum vda_a_mode {
    VDA_A_DENSE = 0,      /* A row m at A + m*lda (fp16) */
    VDA_A_CONV3X3 = 1     /* implicit GEMM: A row m = output pixel (b,oy,ox), K = (ky,kx,ci) over an NHWC fp16 input, pad 1 */
};

enum vda_epilogue {
    VDA_EPI_BIAS_F16 = 0,        /* out_h[m,n] = acc + bias[n]                                   */
    VDA_EPI_BIAS_GELU_F16 = 1,   /* out_h = gelu_erf(acc + bias)                 mlp.py:36-37    */
    VDA_EPI_BIAS_RELU_F16 = 2,   /* out_h = relu(acc + bias)                     blocks.py:79-83 */
    VDA_EPI_SCALE_RES_F32 = 3,   /* out_f[m,n] = res_f[m,n] + gamma[n]*(acc + bias[n])  block.py:105-106, layer_scale.py:28 (gamma NULL = 1) */
    VDA_EPI_RES_F16 = 4,         /* out_h = acc + bias + res_h[m,n] (+ res2_h[m,n])  blocks.py:91,145; motion_module.py:123 */
    VDA_EPI_GEGLU_F16 = 5,       /* W rows interleaved [16 value | 16 gate]: out_h[m, n/2] = (acc_v+b_v) * gelu(acc_g+b_g)  attention.py:383-384 */
    VDA_EPI_PATCH_F32 = 6,       /* out_f[(m/P)*(P+1) + 1 + m%P, n] = acc + bias[n] + pos[(1 + m%P)*N + n]   dinov2.py:218-219 */
    VDA_EPI_CONVT_F16 = 7,       /* W rows ordered (ky,kx,co): pixel-shuffle scatter of a k==stride ConvTranspose2d  dpt.py:71-82 */
    VDA_EPI_BIAS_F32 = 8,        /* out_f[m,n] = acc + bias[n] */
    VDA_EPI_SCALE_RES_F32_H = 9, /* out_h[m,n] = (fp16)(res_f[m,n] + gamma[n]*(acc + bias[n])): leaves an fp32 residual stream */
    /* ---- LayerNorm folded into the GEMMs either side of it (block.py:105-106 + :56/:68 of the next sub-block), fp16 path only.
     * The fp32 residual stream x is kept as TWO fp16 planes, x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22+ significant
     * bits): the hi plane IS the A operand of the GEMM that consumes LayerNorm(x), so no LayerNorm pass touches memory. */
    VDA_EPI_SCALE_RES_SPLIT = 10,/* x' = (res_h + res2_h) - c[m] + gamma[n]*(acc + bias[n]); out_h = fp16(x'), out2_h = fp16(x' - out_h) (in
                                    place over res / res2 is allowed); stats[n/64, m, :] = (sum, centred sum of squares) of x' over the 64
                                    columns n/64*64.. (fp32; N % 64 == 0): vda_ln_stats_finalize turns them into (mean, rstd) rows.
                                    c[m] = pos[2m] when pos != NULL (the [M, 2] (mean, rstd) rows of the LayerNorm in front of this
                                    branch: the stream is RE-CENTRED, i.e. kept relative to each token's own mean - every reader of it
                                    is a LayerNorm, invariant to a per-row shift - so the fp16 operand plane rounds relative to the
                                    token's spread, not its offset), 0 when pos == NULL (zero_page must then be set) */
    VDA_EPI_LN_BIAS_F16 = 11,    /* A = hi plane, W = W*diag(ln_w) (vda_fold_ln_weight): out_h = rstd[m]*(acc - mean[m]*gamma[n]) + bias[n]
                                    with (mean, rstd) = stats[m, 0:2], gamma = c1 = row sums of the folded W, bias = c2 = b + W.ln_b */
    VDA_EPI_LN_GELU_F16 = 12     /* the same followed by gelu_erf (mlp.fc1) */
};

typedef struct vda_gemm_args {
    const void* A;          /* fp16: dense [M,lda] or NHWC input [B,H,W,Cin] */
    const void* W;          /* fp16 [N,K], K contiguous */
    const float* bias;      /* [N] or NULL */
    void* out;
    const void* res;        /* residual (type per epilogue) or NULL */
    const void* res2;       /* second fp16 residual for VDA_EPI_RES_F16 or NULL */
    const float* gamma;     /* [N] LayerScale or NULL */
    const float* pos;       /* VDA_EPI_PATCH_F32: pos-embed [(P+1), N]; VDA_EPI_SCALE_RES_SPLIT: NULL or re-centring rows [M, 2] */
    const void* zero_page;  /* >= 256 B of zeros (conv padding source) */
    int32_t M, N, K;
    int32_t lda, ldc;       /* in elements */
    int32_t a_mode, epilogue;
    int32_t relu_in;        /* apply relu to A elements on load (blocks.py:78) */
    /* VDA_A_CONV3X3 */
    int32_t cB, cH, cW, cCin, cHo, cWo, cStride;
    /* VDA_EPI_PATCH_F32: P patches per frame. VDA_EPI_CONVT_F16: k, input h, w, Cout */
    int32_t P, tK, tH, tW, tCout;
    void* out2;             /* VDA_EPI_SCALE_RES_SPLIT: lo plane of the result */
    float* stats;           /* VDA_EPI_SCALE_RES_SPLIT: out, [N/64, M, 2] partial row statistics;
                               VDA_EPI_LN_*: in, [M, 2] (mean, rstd) */
    int32_t* sched;         /* NULL, or eight int32 counters ZEROED before the launch (one per XCD): the 8-phase kernel then draws its
                               tiles dynamically - a launch that shares the GPU with another kernel degrades by the CUs it lost,
                               not by a whole shift of tiles. Other kernels ignore it. Results do not depend on it. */
    int32_t stats_ld;       /* VDA_EPI_SCALE_RES_SPLIT: rows per column block of the stats array (0 = M): lets a caller run a GEMM as
                               several row ranges (pointers advanced, M = rows of the range) into one [N/64, stats_ld, 2] array */
    int32_t tile_rows;      /* 0 = the dispatcher's choice; 192 = 192-row tiles where the kernel family has them (the remainder
                               launch of a row split, vda_gemm_plan_split). Results do not depend on it. */
} vda_gemm_args;

int vda_gemm_f16(const vda_gemm_args* args, vda_stream_t stream);
/* fp32-operand twin (v_mfma_f32_32x32x2_f32, exact fp32): A, W, res, res2 and out are fp32; K % 16 == 0, conv Cin % 16 == 0.
 * The epilogue ids keep their meaning (what is fused); every "_F16" output / residual is fp32 here. */
int vda_gemm_f32(const vda_gemm_args* args, vda_stream_t stream);
/* Tuning / A-B hook (process-wide): -1 (default) picks the kernel per shape; 0 = 128-row tiles; 1 / 2 = 256x256 / 256x128 tiles on
 * 32x32x16 MFMA; 3 / 4 = the same tiles on 16x16x32 MFMA, one barrier per K tile; 5 = 256x256 8-phase two-group schedule (what -1
 * uses for wide N), 5 + 16*flags = the same with debug switches (gemm8p_kernel.h: 1 early next-tile prefetch, 2 clock stamps,
 * 4 / 8 prefetch placement, 16 no start stagger), 5 + 32*s = K-loop schedule s; 7 = patch-in-LDS direct 3x3 conv where it
 * applies; 8 = 192x128 tiles on six waves; 9 = 256x128 8-phase; 10 = 192x384 tiles on twelve waves where N is a multiple of 384
 * (an A/B form: not what -1 picks, see gemm.hip); 5 + 16*64 = 8-phase on 192x256 tiles where built. Environment: VDA_CONV_LDS, VDA_GEMM_8P128, VDA_GEMM_BIG_MIN_N,
 * VDA_GEMM_STAGGER, VDA_GEMM_BM192 (defaults 1, 0, 192, 0, 1). */
int vda_gemm_set_variant(int v);
/* Diagnostic switches of the 8-phase kernel (0 = none; never set by the model). Bit 0: clock stamps - thread 0 of every workgroup
 * writes (s_memtime, s_memrealtime) at tile start / K-loop start / K-loop end / tile end of its first 16 tiles into args.pos as
 * int64 [workgroups][16][4][2] (an epilogue that does not use pos; the in-kernel clock is d(memtime) / d(memrealtime) x 100 MHz:
 * tools/gemm_clock.py). Bit 1: L2-blocked tile order (an XCD keeps four column panels for the whole launch). Bit 2 (with bit 0): the
 * shader clock at the top of the first 32 K tiles of each workgroup's second tile, int64 [workgroups][32] behind the tile stamps. */
int vda_gemm_set_debug(int flags);
/* Cap on the workgroups the persistent GEMM / conv kernels launch (0 = one per CU, the default; a multiple of 8): lets two
 * independent launch sequences on two streams take half the chip each instead of queueing behind each other's full grids. */
int vda_set_max_wgs(int n);
/* Row split of a large dense GEMM on the current device: returns M1 <= M. Rows [0, M1) fill whole rounds of 256 x 256 tiles on the
 * device's CUs; rows [M1, M) run as one more call with tile_rows = 192 (a 192-row round costs ~0.8 of a 256-row one). M1 == M: no
 * split pays (or the epilogue / shape has no 192-row kernel). vda_gemm_f16 applies the same plan by itself when sched == NULL and
 * tile_rows == 0; a caller that wants the two launches bracketed separately (or has per-launch sched counters) splits by hand:
 * advance A / out / res / res2 / out2 by M1 rows, stats (VDA_EPI_LN_*) and pos (VDA_EPI_SCALE_RES_SPLIT) by M1 rows of 2 floats,
 * stats (VDA_EPI_SCALE_RES_SPLIT) by M1 rows of 2 floats with stats_ld = M. A row's result is bit-identical either way. */
int vda_gemm_plan_split(int M, int N, int K, int epilogue, int a_mode);
/* *out = the arguments of rows [r0, r0 + rows) of *args by exactly those rules. lda / ldc of 0 are read as K / N HERE (a row range
 * is of a real matrix); vda_gemm_f16 itself reads lda == 0 as "every row is A's row 0" (broadcast) and never row-splits such a
 * call on its own. Any kernel family honours stats_ld (the 8-phase, one-barrier and 128-row kernels alike). */
int vda_gemm_row_range(const vda_gemm_args* args, int r0, int rows, vda_gemm_args* out);
/* Name of the kernel family the last vda_gemm_f16 call on this thread dispatched to (for profiling reports). */
const char* vda_gemm_last_kernel(void);

/* ---------------------------------------------------------------- norms
 * LayerNorm over the last dim, fp32 statistics (dinov2.py:95,310; block.py:56,68;
 * motion_module.py:155,161,166). in: fp32 [rows, D]; out: fp16.
 *  group/skip: when group > 0, row r with r % group < skip is dropped and the
 *  output is compacted (cls token removal for the taps, dpt_temporal.py:61).
 *  pe/pe_rows_per_step: when pe != NULL, out += pe[(r / pe_rows_per_step) % pe_steps, :]
 *  (temporal sinusoidal PE added to the normed input, motion_module.py:235).
 */
int vda_layernorm_f32_f16(const float* in, void* out, const float* w, const float* b, float eps,
                          int rows, int D, int group, int skip,
                          const float* pe, int pe_rows_per_step, int pe_steps, vda_stream_t stream);

/* Residual add + LayerNorm in one pass (dinov2_layers/block.py:105-106 followed by :56/:68 of the next sub-block):
 *   x[r,:] += gamma[:] * y[r,:]   (x fp32 in place; y fp16 = the bias-added projection output; gamma = LayerScale, NULL = 1)
 *   out[r',:] = LayerNorm(x[r,:]) (fp16; group/skip as above)
 * The rounding of y to fp16 before the LayerScale multiply is the reference's own under autocast (the Linear returns fp16,
 * layer_scale.py:28 promotes to fp32). */
int vda_layernorm_residual_f32_f16(float* x, const void* y, const float* gamma, void* out, const float* w, const float* b,
                                   float eps, int rows, int D, int group, int skip, vda_stream_t stream);
/* fp32-operand path: same, fp32 out. */
int vda_layernorm_f32_f32(const float* in, float* out, const float* w, const float* b, float eps,
                          int rows, int D, int group, int skip,
                          const float* pe, int pe_rows_per_step, int pe_steps, vda_stream_t stream);

/* ---- LayerNorm folded into the neighbouring GEMMs (VDA_EPI_SCALE_RES_SPLIT / VDA_EPI_LN_*; block.py:56,68,105-106)
 * vda_split_stats_f32: fp32 rows x [rows, D] -> the two fp16 planes (hi = fp16(x), lo = fp16(x - hi)) and the row statistics
 *   stat[r] = (mean, rstd = 1/sqrt(var + eps)) (two-pass, fp32): the entry into the split stream after the patch embedding.
 * vda_split_center_stats_f32: the same with the row's mean taken out first: hi + lo = x - mean(x), stat[r] = (0, rstd) - the entry
 *   the model uses (the stream then stays relative to each token's mean: VDA_EPI_SCALE_RES_SPLIT's pos).
 * vda_ln_stats_finalize: partial[np, r, 2] (sum, centred sum of squares per 64 columns, as VDA_EPI_SCALE_RES_SPLIT writes them)
 *   -> stat[r] = (mean, rstd), combined in column order (Chan et al.), D = 64*np. overflow (device int32, may be NULL) is set to 1
 *   when a row's statistics are not finite: the fp16 planes hold a token only up to 65 504 from its own mean (the reference's
 *   stream is fp32, block.py:105-106, and has no such limit); a saturated plane shows as an infinite / NaN sum here.
 * vda_layernorm_split_f16: LayerNorm of x = hi + lo (fp32 statistics), fp16 out, group/skip as vda_layernorm_f32_f16 (the taps).
 * vda_fold_ln_weight: pack-time fold of LayerNorm's affine into the Linear that follows it: Wf[n,k] = fp16(W[n,k]*ln_w[k]),
 *   c1[n] = sum_k Wf[n,k] (of the ROUNDED values, fp32), c2[n] = b[n] + sum_k W[n,k]*ln_b[k] (fp32; b may be NULL). */
int vda_split_stats_f32(const float* x, void* hi, void* lo, float* stat, float eps, int rows, int D, vda_stream_t stream);
int vda_split_center_stats_f32(const float* x, void* hi, void* lo, float* stat, float eps, int rows, int D, vda_stream_t stream);
int vda_ln_stats_finalize(const float* partial, float* stat, float eps, int rows, int np, int32_t* overflow, vda_stream_t stream);
int vda_layernorm_split_f16(const void* hi, const void* lo, void* out, const float* w, const float* b, float eps,
                            int rows, int D, int group, int skip, vda_stream_t stream);
int vda_fold_ln_weight(const float* W, const float* bias, const float* ln_w, const float* ln_b, void* Wf, float* c1, float* c2,
                       int N, int K, vda_stream_t stream);

/* ---- One block's MLP branch on the split residual stream in ONE kernel (dinov2_layers/mlp.py:35-41, block.py:105-106,
 * layer_scale.py:27-28; fp16 path with the LayerNorm fold): hi + lo += gamma * (fc2(GELU(fc1(LayerNorm(hi + lo)))) + b2), hid never
 * leaves the registers. Built for the widths vda_mlp_fused_supported(D, hidden) reports (D = 384, hidden = 1536: ViT-S).
 *   hi_in  fp16 [M, D]   the A operand (the hi plane; may be `hi` itself: a workgroup reads and writes its own rows only)
 *   stats  fp32 [M, 2]   (mean, rstd) of the LayerNorm in front of fc1 (vda_ln_stats_finalize); mean also re-centres the stream
 *   w1, c1, c2           fc1 with the LayerNorm folded in: vda_fold_ln_weight's Wf [hidden, D], c1 [hidden], c2 [hidden]
 *   w2p    fp16 [D, hidden]  fc2's weight with its hidden columns in the order vda_mlp_permute_w2_f16 gives them
 *   b2, gamma fp32 [D]   fc2's bias, LayerScale
 *   hi, lo fp16 [M, D]   the planes, updated in place;   part fp32 [D/64, stats_ld, 2]: partial row statistics (as VDA_EPI_SCALE_RES_SPLIT)
 * Deterministic; a row's result does not depend on its position. */
int vda_mlp_fused_supported(int D, int hidden);
int vda_mlp_permute_w2_f16(const void* w2, void* w2p, int D, int hidden, vda_stream_t stream);
int vda_mlp_fused_f16(const void* hi_in, const float* stats, const void* w1, const float* c1, const float* c2, const void* w2p, const float* b2,
                      const float* gamma, void* hi, void* lo, float* part, int M, int D, int hidden, int stats_ld, vda_stream_t stream);

/* pe = 'rope' (motion_module.py:254-257, motion_module/attention.py:403-429): channel pairs (2i, 2i+1) of the q and k thirds of the
 * fused projection qkv [T*hw, 3*C] (frame-major rows) rotated in place by  frame * 10000^(-2i/C)  (fp32 arithmetic, over the FULL
 * width C, before the head split); v is untouched. */
int vda_rope_qk_f16(void* qkv, int T, int hw, int C, vda_stream_t stream);
int vda_rope_qk_f32(float* qkv, int T, int hw, int C, vda_stream_t stream);

/* GroupNorm(32 groups) per frame on NHWC fp16 [frames, hw, C] -> fp16 [frames*hw, C]
 * (motion_module.py:84,110). `partial` is workspace of frames*chunks*groups*2 floats. */
int vda_groupnorm_nhwc_f16(const void* in, void* out, const float* w, const float* b, float eps,
                           int frames, int hw, int C, int groups, float* partial, int chunks, vda_stream_t stream);
int vda_groupnorm_nhwc_f32(const float* in, float* out, const float* w, const float* b, float eps,
                           int frames, int hw, int C, int groups, float* partial, int chunks, vda_stream_t stream);

/* ---------------------------------------------------------------- attention
 * Spatial self-attention of the ViT (dinov2_layers/attention.py:51-59 or the xformers
 * call at :76): qkv fp16 [B, N, 3, heads, 64] -> out fp16 [B, N, heads*64],
 * softmax(q k^T / 8) v with fp32 softmax, never materialising the N x N scores. */
int vda_attention_f16(const void* qkv, void* out, int B, int N, int heads, vda_stream_t stream);
/* fp32-operand twin: qkv fp32 [B, N, 3, heads, 64] -> out fp32 [B, N, heads*64], every product on fp32 MFMA. */
int vda_attention_f32(const float* qkv, float* out, int B, int N, int heads, vda_stream_t stream);
/* A-B hook: -1 = the default kernel (10); 8 / 9 / 10 = the softmax reference point enters through the score MFMAs' C operand (Q
 * pre-scaled by log2(e)/8), 9 with the lazy rescale (reference point moved only when a score exceeds it by more than 2^6), 10 deciding
 * that from the row sums instead of a running maximum and staging K / V with scalar-offset buffer loads; 1 = the round-2
 * kernel (V^T fragments via ds_read_b64_tr_b16, scalar softmax math); 2 = packed fp32 softmax math; 0 = scalar LDS
 * reads of V (cross-check); 3 = row sums through the matrix pipe; 4 / 5 = running max through the matrix pipe (without / with 3);
 * 7 = software-pipelined form (next tile's score MFMAs inside the softmax; three waves per SIMD);
 * 11 = 10 with LDS counters in place of the per-tile workgroup barrier (A/B); 20 + k = timing ablations (WRONG results: no exp / max / row sums / PV MFMAs / 1 of 4 QK k-steps), tools/attn_one.py. */
int vda_attention_set_variant(int v);

/* Temporal attention across T frames per pixel (motion_module.py:232-295,
 * motion_module/attention.py:182-211): qkv fp16 [T*hw, 3*C] frame-major rows,
 * 8 heads of d = C/8 -> out fp16 [T*hw, C]. Batch b>1 is expressed by calling per clip. */
int vda_temporal_attention_f16(const void* qkv, void* out, int T, int hw, int C, int heads, vda_stream_t stream);
int vda_temporal_attention_f32(const float* qkv, float* out, int T, int hw, int C, int heads, vda_stream_t stream);
/* 1 (default): MFMA kernel for head dims 32 / 64 / 128, VALU kernel otherwise; 0: VALU kernel everywhere (cross-check). */
int vda_temporal_attention_set_variant(int v);

/* ---------------------------------------------------------------- resampling / layout
 * Bilinear, align_corners=True (util/blocks.py:156-158, dpt_temporal.py:94-96,
 * video_depth.py:162,208). NHWC fp16 -> NHWC fp16, optional elementwise add of `add`
 * (same shape as out). */
int vda_bilinear_nhwc_f16(const void* in, void* out, const void* add, int B, int h, int w, int H, int W, int C,
                          vda_stream_t stream);
int vda_bilinear_nhwc_f32(const float* in, float* out, const float* add, int B, int h, int w, int H, int W, int C,
                          vda_stream_t stream);
/* fp32 planes [B,h,w] -> [B,H,W], optional relu (video_depth.py:162-163,208). */
int vda_bilinear_plane_f32(const float* in, float* out, int B, int h, int w, int H, int W, int relu, vda_stream_t stream);

/* Patch gather for the 14x14/s14 patch-embed conv (patch_embed.py:76): fp32 NCHW
 * [B,3,H,W] -> fp16 [B*(H/14)*(W/14), Kpad], column = c*196 + ky*14 + kx, zero padded to Kpad. */
int vda_patchify_f32_f16(const float* x, void* out, int B, int H, int W, int Kpad, vda_stream_t stream);
int vda_patchify_f32_f32(const float* x, float* out, int B, int H, int W, int Kpad, vda_stream_t stream);

/* Positional-embedding grid for a (ph x pw)-patch input (dinov2.py:185-210): pe fp32 [1 + g*g, D] -> out fp32 [1 + ph*pw, D];
 * row 0 (cls) is copied, the g x g grid is resampled bicubic (a = -0.75, half-pixel centres, clamped taps) with the
 * reference's explicit scale factors ((ph + 0.1)/g, (pw + 0.1)/g). Not needed when ph*pw == g*g and the input is square. */
int vda_pos_embed_resample_f32(const float* pe, float* out, int g, int ph, int pw, int D, vda_stream_t stream);

/* cls rows of the token matrix: tok[b*(P+1), :] = cls + pos[0] (dinov2.py:218-219). */
int vda_cls_rows_f32(float* tok, const float* cls, const float* pos, int B, int P, int D, vda_stream_t stream);

/* Input of the head's readout projection when use_clstoken=True (dpt_temporal.py:56-59, dpt.py:129-132): final-norm'd tokens
 * [frames*(P+1), D] (cls at row 0 of every frame) -> [frames*P, 2D] = cat(patch token, the frame's cls token). The projection itself
 * (Linear 2D -> D + GELU) is vda_gemm_* with VDA_EPI_BIAS_GELU_F16. */
int vda_readout_concat_f16(const void* tok, void* out, int frames, int P, int D, vda_stream_t stream);
int vda_readout_concat_f32(const float* tok, float* out, int frames, int P, int D, vda_stream_t stream);

/* Final 1x1 conv 32->1 + ReLU on NHWC fp16 [rows, Cpad] (first 32 channels used)
 * -> fp32 [rows] (dpt.py:121-122). */
int vda_head_out_f16_f32(const void* in, const float* w, float bias, float* out, int rows, int Cpad, vda_stream_t stream);
int vda_head_out_f32_f32(const float* in, const float* w, float bias, float* out, long long rows, int Cpad, vda_stream_t stream);

/* Fused depth tail (dpt.py:118-122, dpt_temporal.py:93-100, video_depth.py:162-163): NHWC fp16 [B,h,w,C] ->
 * [bilinear align_corners resize to H x W when (h,w) != (H,W)] -> 3x3 conv C->32 (+b2, ReLU) -> 1x1 conv 32->1 (+b3, ReLU)
 * -> fp32 [B,H,W]. w2: fp16 [32, 9*C] with K ordered (ky,kx,ci); C a multiple of 32; zero_page: >= 256 B of zeros. */
int vda_depth_tail_f16(const void* in, const void* w2, const float* b2, const float* w3, float b3, float* out,
                       const void* zero_page, int B, int h, int w, int H, int W, int C, vda_stream_t stream);
/* A/B switch of the resizing form: 0 (default) = the persistent kernel with the weights resident in LDS (C <= 128), 1 = the round-1
 * kernel (one 8 x 32 tile per workgroup). Same arithmetic, bit-identical results. */
int vda_depth_tail_set_variant(int v);

/* output_conv1 applied to the 2x-upsampled output of refinenet1 (dpt.py:117 over util/blocks.py:156-160's
 * F.interpolate(scale_factor=2, mode="bilinear", align_corners=True)) in one pass: NHWC fp16 in [B,h,w,C] ->
 * out [B,2h,2w,ldc] = conv3x3(bilinear2x(in)) + bias, channels 0..N-1 written. w: fp16 [N, 9*C] with K ordered (ky,kx,ci)
 * (the layout vda_gemm_f16's VDA_A_CONV3X3 takes); C a multiple of 16; N <= 128, N and ldc multiples of 4; bias fp32 [N] or NULL.
 * The interpolated pixels are rounded to fp16 once, exactly as vda_bilinear_nhwc_f16 would have stored them. */
int vda_conv3x3_up2_f16(const void* in, const void* w, const float* bias, void* out, int B, int h, int wd, int C, int N, int ldc,
                        vda_stream_t stream);
/* Timing experiments of the fused kernel at N > 64 (results invalid): 1 = no interpolation, 2 = no MFMA groups; 0 = the kernel. */
int vda_conv3x3_up2_set_variant(int v);
/* The LDS-patch 3x3 convolution behind vda_gemm_f16 (N <= 64, stride 1): 0 (default) = the persistent C = 64 kernel where it applies,
 * 1 = the per-pass kernel for every shape. Bit-identical results; A/B only. */
int vda_conv_lds_set_variant(int v);

/* uint8 RGB frames [n,H,W,3] (already at network size) -> normalised fp32 NCHW
 * [n,3,H,W]: (x/255 - mean)/std  (video_depth.py:198, util/transform.py:134,147). */
int vda_normalize_u8_f32(const uint8_t* frames, float* out, int n, int H, int W, vda_stream_t stream);
/* Same, gathering frame idx[i] (device int32[n], each < n_video) of a uint8 video [n_video,H,W,3] resident in HBM:
 * the window gather + key-frame refill of video_depth.py:197-201 without a host round trip. */
int vda_gather_normalize_u8_f32(const uint8_t* video, const int32_t* idx, float* out, int n, int n_video, int H, int W,
                                vda_stream_t stream);

/* Window gather + resize to the network size + normalise, for source frames that are not already H x W (the usual case for
 * real video, util/transform.py:109-147): out[i] = normalise(bicubic(video[idx[i]] / 255)), uint8 [n_video,H0,W0,3] ->
 * fp32 NCHW [n,3,H,W]. Bicubic as cv2.INTER_CUBIC defines it: a = -0.75, half-pixel centres, taps clamped to the border,
 * no antialiasing. (cv2 is absent offline: parity of this leg against cv2 itself is unpinned; it is tested against the
 * same definition evaluated by torch on the CPU.) */
int vda_gather_resize_normalize_u8_f32(const uint8_t* video, const int32_t* idx, float* out, int n, int n_video, int H0, int W0,
                                       int H, int W, vda_stream_t stream);

/* ---- window stitcher on the device (video_depth.py:216-254) ------------------------------------------------
 * Least-squares scale / shift of `pred` against `target` over all n pixels (utils/util.py:40-62 with the all-ones
 * mask of video_depth.py:232): scale_shift[0..1] (device) = closed form on fp64 sums, (1, 0) when det == 0.
 * workspace: >= 4*nblk doubles (device); deterministic (fixed reduction order, no atomics). */
int vda_lsq_scale_shift_f32(const float* pred, const float* target, long long n, double* workspace, int nblk, float* scale_shift,
                            vda_stream_t stream);
/* One window k > 0, frames of px pixels, win = fp32 [32, px] (video_depth.py:235-250, utils/util.py:65-74), aff(d) = max(d*scale+shift, 0):
 *   chunk[0..7]  = tail[j]*wts[j] + aff(win[2+j])*wts[8+j]   (cross-fade, wts = (1-w_j | w_j), w = 0,1/7,..,1)
 *   chunk[8..21] = aff(win[10..23]);  tail[0..7] = aff(win[24..31]);  ref1 = aff(win[12]).
 * chunk: fp32 [22, px]; tail: fp32 [8, px] updated in place; scale_shift, wts: device fp32 [2], [16]. */
int vda_stitch_window_f32(const float* win, const float* scale_shift, float* chunk, float* tail, float* ref1, long long px,
                          const float* wts, vda_stream_t stream);
/* out[i] = aff(in[i]) for n elements, the same arithmetic as the stitch (video_depth.py:238,243,249): aligns key frames another rank
 * computed (the key-frame exchange of the multi-rank schedule). in == out is allowed. */
int vda_affine_clamp_f32(const float* in, const float* scale_shift, float* out, long long n, vda_stream_t stream);

/* ================================================================ handle API: the model behind one pointer
 * What a C / C++ host binds in place of the reference's Python class (the seam of SURVEY.md section 8b):
 *   VideoDepthAnything(**model_configs[enc])          run.py:45, video_depth.py:38-63      vda_create
 *   .load_state_dict(torch.load(ckpt), strict=True)   run.py:46                            vda_load_weight per tensor, then
 *                                                                                          vda_finalize_weights
 *   .to('cuda')                                       run.py:47                            (the handle lives on the device
 *                                                                                           that is current at vda_create)
 *   model.forward(x)                                  video_depth.py:89-93,161-164         vda_forward
 * Ownership: the caller owns `in`, `out` and (optionally) the workspace block; the handle owns its weights and frees them
 * in vda_destroy. Errors: non-zero return, text from vda_last_error() - for state-dict problems in torch's own wording
 * ("Missing key(s) in state_dict: ...", "Unexpected key(s) ...", "size mismatch for ..."). A handle is not thread-safe and
 * is bound to one device: every call must be made with that device current.
 * Allocation / synchronisation: vda_load_weight copies synchronously (host or device source); vda_finalize_weights packs
 * the fp16 layouts on the device and synchronises; the FIRST vda_forward of a new (shape, precision) - or vda_prepare -
 * may allocate (the fp32 weight pack, the positional embedding at that grid, the handle's own workspace when the caller
 * gave none). After that vda_forward only enqueues kernels on `stream`. */
typedef struct vda_model vda_model;

typedef struct vda_config {
    int32_t embed_dim;        /* 384 (vits) / 1024 (vitl); num_heads * 64                  dinov2.py:339-378 */
    int32_t depth;            /* 12 / 24 transformer blocks */
    int32_t num_heads;        /* 6 / 16 */
    int32_t taps[4];          /* blocks whose output feeds the head: 2,5,8,11 / 4,11,17,23   video_depth.py:53-56 */
    int32_t features;         /* 64 / 256                                                    run.py:41-42 */
    int32_t out_channels[4];  /* 48,96,192,384 / 256,512,1024,1024 */
    int32_t num_frames;       /* 32: temporal window (pos_encoder.pe rows) */
    int32_t use_clstoken;     /* 0 (every released config) / 1: head.readout_projects fold the cls token in, dpt.py:92-98 */
    int32_t use_bn;           /* 0 (every released config) / 1: BatchNorm2d after each conv of the fusion blocks' ResidualConvUnits
                                 (util/blocks.py:60-62,80-86; inference = an affine per channel, folded into the conv at pack time) */
    int32_t pe_rope;          /* 0 = pe 'ape' (every released config): sinusoidal pos_encoder.pe added before q/k/v; 1 = pe 'rope':
                                 no pe buffer in the checkpoint, q and k rotated per frame (motion_module.py:221-224,254-257) */
} vda_config;

enum vda_precision {
    VDA_PREC_F16 = 0,         /* fp16 MFMA operands, fp32 accumulate / residual streams / norm + softmax statistics */
    VDA_PREC_F32 = 1          /* fp32 operands everywhere (exact-fp32 MFMA): the reference's fp32=True */
};
enum vda_dtype { VDA_DTYPE_F32 = 0 };

int vda_create(const vda_config* cfg, vda_model** out);
int vda_destroy(vda_model* h);
/* Number of tensors the state dict must hold (351 for vits, 519 for vitl). */
int vda_num_weights(const vda_model* h);
/* One tensor of the flat fp32 state dict, under the reference's key (e.g. "pretrained.blocks.0.attn.qkv.weight"); `ptr` is
 * host or device memory and is copied. Unknown keys and shape mismatches are refused. */
int vda_load_weight(vda_model* h, const char* name, const void* ptr, const int64_t* dims, int ndim, int dtype);
/* strict=True check (every key present) + repack to the kernel layouts. */
int vda_finalize_weights(vda_model* h);
/* Bytes of workspace a forward of x[B,T,3,H,W] needs at this precision (-1 on error). */
int64_t vda_workspace_bytes(vda_model* h, int B, int T, int H, int W, int precision);
/* Optional: run in a caller-owned, 256-byte aligned block (e.g. a torch tensor) instead of one the handle allocates. */
int vda_set_workspace(vda_model* h, void* ptr, int64_t bytes);
/* Optional: do now whatever the first forward of this shape / precision would allocate. */
int vda_prepare(vda_model* h, int B, int T, int H, int W, int precision);
/* in: fp32 device [B,T,3,H,W] (normalised frames; H, W multiples of 14; T <= num_frames) -> out: fp32 device [B,T,H,W]. */
int vda_forward(vda_model* h, const float* in, float* out, int B, int T, int H, int W, int precision, vda_stream_t stream);
/* Deferred status of the forwards enqueued so far (vda_forward itself only enqueues). 0 = every forward whose stream work has
 * COMPLETED was valid; 4 = one of them left fp16's range in the split residual stream of the default fp16 path ("ln_fold": a token
 * further than 65 504 from its own mean; the reference's stream is fp32, block.py:105-106): its depth was overwritten with NaN and
 * vda_last_error() names the way out (vda_set_option "ln_fold" 0, or the fp32 path). Call it after synchronising the stream(s);
 * a report nobody collected fails the NEXT vda_forward instead. Reports are cleared once returned. */
int vda_forward_status(vda_model* h);
/* Parity hook: copy `bytes` of a named intermediate of the last forward ("tap0".."tap3", "l1", "l2", "l3t", "l4t", "p4t",
 * "p3t", "p2", "p1"; activation dtype of that forward's precision, channels padded to multiples of 64) to device `dst`. */
int vda_debug_copy(vda_model* h, const char* name, void* dst, int64_t bytes, vda_stream_t stream);
/* Test utility: occupy `wgs` workgroups (256 threads, lds_bytes of LDS each) for about `cycles` shader clocks on `stream` - a
 * stand-in for a communication kernel running beside the forward (tools/contention.py). Bounded: every wave leaves by itself. */
int vda_debug_occupy(int wgs, int lds_bytes, long long cycles, vda_stream_t stream);
/* Launch-sequence switches (A/B and cross-checks). "residual_in_ln" (default 0; fp16 path only): 1 = attn.proj / mlp.fc2 store
 * their output as fp16 and the residual add runs inside the following LayerNorm (vda_layernorm_residual_f32_f16); 0 = the add
 * is the GEMM's fp32 in-place epilogue (VDA_EPI_SCALE_RES_F32; measured 2 % faster end to end). Changes the workspace size.
 * "ln_fold" (default 1), "dyn_sched" (default 0): csrc/host.hip. "oc1_fused" (default 1; fp16 path): refinenet1's 2x upsample is
 * evaluated inside output_conv1 (vda_conv3x3_up2_f16) and path_1 never exists at full size; 0 = vda_bilinear_nhwc + the conv.
 * "mlp_fused" (default 0; fp16 path with ln_fold, widths vda_mlp_fused_supported reports): 1 = a block's fc1 + GELU + fc2 + residual
 * run as vda_mlp_fused_f16 instead of the two GEMM launches (measured slower on the MI355X: kept as a tested option).
 * "head_overlap" (default 0): 1 = the part of the head that needs taps 0..2 only (dpt_temporal.py:55-69 for i < 3, :75, :78-80) runs
 * on a side stream of the handle as soon as tap 2 exists, under the remaining encoder blocks; bit-identical results. */
int vda_set_option(vda_model* h, const char* name, int value);
/* Measurement hook (bench.py): from vda_profile_start until vda_profile_stop every `every`-th GEMM / conv launch of each
 * (shape, epilogue) inside vda_forward is bracketed by two events on the launch stream. vda_profile_stop waits for them and
 * writes a JSON object {kernel: {"launches", "flops", "timed", "timed_ms", "timed_flops"}} into json[cap]. */
int vda_profile_start(vda_model* h, int every);
int vda_profile_stop(vda_model* h, char* json, int cap);

#ifdef __cplusplus
}
#endif
#endif
