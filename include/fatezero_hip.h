/* fatezero_hip.h -- C ABI of libfatezero_hip.so, the MI355X (gfx950) kernels behind the FateZero
 * DDIM-inversion -> attention-fusion denoise loop.
 *
 * The reference (ChenyangQiQi/FateZero) has no native code and no FFI layer: its boundary is the Python
 * `controller(attn, is_cross, place_in_unet)` hook called between softmax and P.V inside the patched
 * attention forward (video_diffusion/prompt_attention/attention_register.py:23-59) plus plain torch modules.
 * Each entry point below replaces the torch ops named in its comment; the Python host
 * (fatezero_amd/) binds them with ctypes -- see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - every tensor is caller-allocated device memory, fp16 unless noted, 16-byte aligned;
 *   - activations are token-major: x[n][token][channel], n = b*F + f (batch-major frame index),
 *     channels = heads*head_dim with the head as the slow sub-index (diffusers reshape_heads_to_batch_dim);
 *   - `stream` is a hipStream_t; calls are asynchronous, never allocate, never synchronise, keep no state;
 *   - return 0 on success, <0 on error (FZ_ERR_*); nothing is thrown.
 */
#ifndef FATEZERO_HIP_H
#define FATEZERO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FZ_ATTN_FLASH 0   /* O = softmax(scale QK^T) V, no map materialised                       */
#define FZ_ATTN_CAPTURE 1 /* same, and the fp16 probability map is written to `p` (exact softmax) */
#define FZ_ATTN_INJECT 2  /* rows take P from `p` (inversion-time map) instead of the live softmax */

#define FZ_MAX_KV_SLOTS 4

/* Sparse-causal spatio-temporal self-attention: `spatial_temporal_forward` + `_attention`
 * (attention_register.py:131-218, :23-59; SparseCausalAttention.forward attention.py:340-422), with the
 * controller's capture (AttentionStore.forward attention_store.py:81-93) and self-attention replacement /
 * blend-mask fusion (AttentionControlEdit.replace_self_attention attention_util.py:80-92) fused in.
 * K/V of frame f are the concatenation over kv slots j of frame kv_src(j,f):
 *   kv_abs[j] ? kv_val[j] : clamp(f + kv_val[j], 0, F-1)          (attention_register.py:162-188)
 * -- the concat is never materialised. V is passed TRANSPOSED per frame: vt[n][channel][token], token
 * stride 1, each row zero-padded to a multiple of 64 tokens. */
typedef struct FzAttnSelfDesc {
    int32_t n_frames;   /* frames covered by this launch                                     */
    int32_t frame0;     /* global index n of the first one (n = b*F + f)                     */
    int32_t clip_len;   /* F                                                                 */
    int32_t heads;
    int32_t head_dim;   /* 16, 32, 40, 64, 80, 128 or 160                                    */
    int32_t lq;         /* query tokens per frame                                            */
    int32_t lkf;        /* key tokens per kv slot (= tokens per source frame)                */
    int32_t n_kv;       /* 1..FZ_MAX_KV_SLOTS                                                */
    int32_t kv_abs[FZ_MAX_KV_SLOTS];
    int32_t kv_val[FZ_MAX_KV_SLOTS];
    float scale;        /* head_dim^-0.5                                                     */
    int32_t mode;       /* FZ_ATTN_*                                                         */
    int64_t q_frame_stride, q_row_stride;   /* elements */
    int64_t k_frame_stride, k_row_stride;
    int64_t vt_frame_stride, vt_chan_stride;
    int64_t o_frame_stride, o_row_stride;
    /* probability map p[frame][head][q][col], col = j*lkf + key  (reference layout [F, heads, Lq, Lk]) */
    int64_t p_frame_stride, p_head_stride, p_row_stride;
    int32_t p_frame_off;    /* p frame index of this launch's first frame                    */
    int32_t mask_frame_off; /* row_mask frame index of this launch's first frame             */
    int64_t k_head_stride;  /* elements between heads of K: 0 = head_dim (heads interleaved in a row);
                               lkf*head_dim with k_row_stride = head_dim for a head-major K [n][head][key][d] */
    int32_t q_log2_scaled;  /* 1: the producer of q already folded scale*log2(e) into it (e.g. into the rows of
                               Wq), so q.k IS the log2-domain logit and `scale` is not applied again.  For
                               head_dim % 16 != 0 this lets the running max ride in a free contraction slot of
                               the QK^T MFMA (csrc/attn_flash.hip).  0: q as the reference's to_q produces it. */
    /* Frame-sharded clips (one clip's frames split over GPUs): k / vt then hold, per batch element, kv_clip_len
     * frames [left halo | the rank's own frames | right halo | anchor frames] gathered from their owners, while q / o
     * keep clip_len (= local) frames.  A relative slot reads frame clamp(f + kv_frame_off + kv_val, 0, kv_clip_len-1),
     * an absolute slot reads frame kv_val of that extended axis.  kv_clip_len == 0: k / vt have clip_len frames and
     * kv_frame_off is ignored (the single-GPU layout). */
    int32_t kv_clip_len;
    int32_t kv_frame_off;
    int32_t reserved0;
} FzAttnSelfDesc;

/* row_mask (INJECT only, may be NULL): float [frames][lq]; 1 -> the row keeps the live attention,
 * 0 -> the row takes the stored map (blend mask of spatial_blend.py:58-124 reshaped [F,1,Lq,1]).
 * NULL -> every row takes the stored map and QK^T is skipped. */
int fz_attn_self(const FzAttnSelfDesc* desc, const void* q, const void* k, const void* vt, void* o,
                 void* p, const float* row_mask, void* stream);

#define FZ_CROSS_MAX_KEYS 96
#define FZ_CROSS_P_STRIDE 80 /* arena row stride of a 77-token cross map (16-byte aligned rows) */

/* Cross-attention `forward` + `_attention` (attention_register.py:71-128, :23-59) with capture and the
 * fused prompt-to-prompt edit of AttentionControlEdit.forward (attention_util.py:129-132):
 *   new = (base @ M) * A + cur * B          per key n: A[n], B[n]
 * which covers Replace (attention_util.py:213-223, M = mapper), Refine (:243-253, M = one-hot gather of
 * mapper, alphas folded into A/B), Reweight (:282-286, equalizer folded) and the per-step word gate
 * cross_replace_alpha (ptp_utils.py:179-199).  K/V are per batch element: k[b][token][channel],
 * vt[b][channel][token] (rows zero-padded to 96 tokens). */
typedef struct FzAttnCrossDesc {
    int32_t n_frames, frame0, clip_len, heads, head_dim, lq, lk;
    float scale;
    int32_t mode; /* FZ_ATTN_FLASH (plain), FZ_ATTN_CAPTURE, FZ_ATTN_INJECT */
    int64_t q_frame_stride, q_row_stride;
    int64_t k_batch_stride, k_row_stride;
    int64_t vt_batch_stride, vt_chan_stride;
    int64_t o_frame_stride, o_row_stride;
    int64_t p_frame_stride, p_head_stride, p_row_stride; /* fp16 map (capture dst / inject base), row stride >= 80 */
    int32_t p_frame_off;
    int32_t store_cur; /* INJECT: also write the un-edited live map to `cur_out` (same strides as p) */
} FzAttnCrossDesc;

/* mapper_t: fp16 [96][96], mapper_t[n][w] = M[w][n] (zero padded); coef: float [2][96] = {A, B};
 * cur_out: fp16 map buffer or NULL. */
int fz_attn_cross(const FzAttnCrossDesc* desc, const void* q, const void* k, const void* vt, void* o,
                  void* p, const void* mapper_t, const float* coef, void* cur_out, void* stream);

/* Temporal attention over frames (un-patched CrossAttention.forward, attention.py:327-337):
 * q,k,v,o: [B*F][tokens][channels] (row stride given); each (b, token, head) attends over its F frames. */
int fz_attn_temporal(const void* q, const void* k, const void* v, void* o, int batch, int clip_len,
                     int tokens, int heads, int head_dim, int64_t qkv_row_stride, int64_t o_row_stride,
                     float scale, void* stream);

/* Same with q/o holding q_frames frames per batch element and k/v holding kv_frames (a frame-sharded clip: the rank's
 * own query frames against the all-gathered keys/values); q/o rows are [B*q_frames][tokens], k/v rows [B*kv_frames][tokens]. */
int fz_attn_temporal_ex(const void* q, const void* k, const void* v, void* o, int batch, int q_frames, int kv_frames,
                        int tokens, int heads, int head_dim, int64_t q_row_stride, int64_t kv_row_stride,
                        int64_t o_row_stride, float scale, void* stream);

/* Blend mask (SpatialBlender.get_mask, spatial_blend.py:24-56): maps: n_maps pointers to fp16 cross maps
 * [P][F][heads][r*r][p_row_stride] (P = n_prompts, prompt stride given), alpha: float [P][80] word weights;
 * out: float [P][F][h][w] of 0/1 after 3x3 max-pool, nearest resize, per-(P,F) max normalisation, > th.
 * If or_with_first != 0 (prompt_choose == 'both'): out[p] |= out[0]. */
int fz_blend_mask(const void* const* maps, int n_maps, int n_prompts, int64_t prompt_stride, int frames,
                  int heads, int res, int64_t p_row_stride, const float* alpha, float th, int out_h, int out_w,
                  int or_with_first, float* out, float* scratch, void* stream);

/* GroupNorm (+SiLU) on token-major activations x[n][tokens][C] (resnet.py:338-339,369,384; attention.py:110;
 * unet_3d_condition.py:439-440).  span = number of consecutive frames sharing statistics: F for the 5-D
 * ResNet norms (stats over C/G x F x H x W), 1 for the per-frame transformer norm.
 * partial: float scratch, >= n_frames * fz_groupnorm_chunks(tokens, C) * G * 3 + (n_frames / span) * G * 2.
 * Small launches (a (stat set, group) of <= 20 channel pairs per thread of one 1024-thread workgroup, few workgroups) run as ONE kernel with
 * exact two-sweep statistics; the others as statistics partials + merge + normalise.  The two forms agree to the rounding of the fp32
 * statistics (an fp16 ulp of y here and there): fz_groupnorm_stats + fz_groupnorm_apply reproduce the three-kernel form bit for bit,
 * fz_groupnorm as a whole to that rounding. */
int fz_groupnorm_chunks(int tokens, int channels);
int fz_groupnorm(const void* x, void* y, const void* gamma, const void* beta, int n_frames, int span,
                 int tokens, int channels, int groups, float eps, int silu, float* partial, void* stream);

/* The same on torch.cat([x1, x2], channel) WITHOUT the concatenated copy: the skip connections of the up blocks
 * (unet_3d_blocks.py:384-395 `hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)` feeding
 * ResnetBlockPseudo3D.norm1, resnet.py:338).  x1: [n][tokens][channels1], x2: [n][tokens][channels2], y: [n][tokens][channels1 +
 * channels2] = the normalised concatenation; gamma / beta over channels1 + channels2; channels1, channels2 % 8 == 0. */
int fz_groupnorm_cat(const void* x1, int channels1, const void* x2, int channels2, void* y, const void* gamma, const void* beta,
                     int n_frames, int span, int tokens, int groups, float eps, int silu, float* partial, void* stream);

/* The two halves of fz_groupnorm for statistics that span frames living on several GPUs:
 *   fz_groupnorm_stats  writes this rank's Welford partials  partial[n_frames][G][chunks][3] = (count, mean, M2);
 *   (the caller all-gathers them over the ranks and orders them [stat_sets][frames_per_set][G][chunks][3])
 *   fz_groupnorm_apply  Chan-merges partial_all per (stat set, group) in a fixed order -- bitwise identical on every
 *                       rank -- into stats[stat_sets][G][2] (scratch) and normalises the local frames:
 *                       frame n uses stat set n / span, so n_frames / span must equal stat_sets. */
int fz_groupnorm_stats(const void* x, int n_frames, int tokens, int channels, int groups, float* partial, void* stream);
int fz_groupnorm_apply(const void* x, void* y, const void* gamma, const void* beta, int n_frames, int span, int tokens,
                       int channels, int groups, float eps, int silu, const float* partial_all, int stat_sets,
                       int frames_per_set, float* stats, void* stream);

/* Linear projection GEMM with fused epilogues: nn.Linear / 1x1 nn.Conv2d of the transformer blocks and ResNet
 * shortcuts -- attention.py:64-66,91-93 (proj_in / proj_out), :199,216 (CrossAttention to_q/k/v/out [3P diffusers 0.11.1]),
 * :232 (FeedForward GEGLU), resnet.py:290 (time_emb_proj), the 1x1 conv_shortcut (resnet.py:302-305):
 *   y[b][r][o] = sum_i x[b][r][i] * w[o][i] + bias[o] (+ res[b][r][o]) (+ res2[b][r][o])
 * FZ_GEMM_GEGLU: w / bias rows are the GEGLU projection (2*inner rows) PACKED so that every group of 64 rows holds 32 `h`
 *   rows followed by the 32 matching `gate` rows (fatezero_amd.kernels.pack_geglu); y[r][c] = h * gelu_erf(gate), inner
 *   columns -- the 2*inner wide intermediate of GEGLU.forward is never written.
 * transpose_out: y[b][o][r] = sum_i x[b][r][i] * w[o][i] (no bias / residual) -- V^T, the value operand of fz_attn_self /
 *   fz_attn_cross, straight out of the projection; columns [rows, rows_store) of every output row are written as zeros.
 * in_features % 8 == 0; ldx, ldw, ldy, ldres % 8 == 0 for the vector paths; fp16 in/out, fp32 accumulation.
 * workspace (optional): fp32 scratch of workspace_floats elements enabling split-K for shapes with too few output tiles
 *   to fill the chip; fz_gemm_workspace_floats(rows, out_features, batch) is always enough.
 * tile_cfg / split_k: 0 = chosen by the library (the normal use); non-zero values pin the tile shape / K step / ring depth
 *   (e.g. 254214 = 2x4 waves of 5x2 MFMA tiles, K step 32, 4-deep LDS ring; the list is in csrc/igemm.hip) and the K split,
 *   for benchmarking. */
#define FZ_GEMM_PLAIN 0
#define FZ_GEMM_GEGLU 1
typedef struct FzGemmDesc {
    int64_t rows;          /* rows of x (tokens) per batch element                     */
    int64_t rows_store;    /* transpose_out: zero-padded row length (>= rows), else 0  */
    int32_t in_features;
    int32_t out_features;  /* rows of w (GEGLU: 2 * inner)                             */
    int64_t ldx, ldw, ldy, ldres; /* row strides in elements (ldres 0: = ldy)          */
    int32_t batch;         /* >= 1; w and bias are shared by the batch                 */
    int32_t epilogue;      /* FZ_GEMM_*                                                */
    int64_t x_batch_stride, y_batch_stride, res_batch_stride;
    int32_t transpose_out;
    int32_t tile_cfg;
    int32_t split_k;
    int32_t reserved0;
    int64_t workspace_floats;
    int64_t w_batch_stride; /* elements between the w matrices of consecutive batch elements; 0: w (and bias) shared by the batch.
                             * Per-batch w = the two batched products of a single-head attention block: scores = q k^T and P v^T^T
                             * (diffusers AttentionBlock of the VAE mid block [3P], reached from stable_diffusion.py:297-319). */
} FzGemmDesc;
int64_t fz_gemm_workspace_floats(int64_t rows, int out_features, int batch);
int fz_gemm(const FzGemmDesc* desc, const void* x, const void* w, const void* bias, const void* res, const void* res2,
            void* y, void* workspace, void* stream);

/* The q | k | V^T projection of a self-attention in ONE launch (SparseCausalAttention.forward, attention.py:340-372: to_q, to_k and
 * to_v all read the same LayerNorm output): w holds the rows [Wq ; Wk ; Wv] (out_features = split_col + C_v), x is read ONCE;
 *   y [row][o]                 = sum_i x[row][i] * w[o][i]                 for o <  split_col   (row stride desc->ldy >= split_col)
 *   yt[row / rows_per_frame][o - split_col][row % rows_per_frame] = the same for o >= split_col  -- V^T, the value operand of
 *                                fz_attn_self, frames yt_frame_stride elements apart, channel rows ldyt elements apart.
 * desc: rows (all frames), in_features, out_features, ldx, ldw, ldy, tile_cfg; batch <= 1, plain epilogue, no bias / residual.
 * split_col % 64 == 0 (every UNet width is), rows_per_frame % 8 == 0 and ldyt >= rows_per_frame: a caller whose attention kernel
 * wants V^T rows zero-padded beyond the frame's tokens (token counts that are not multiples of 64) uses fz_gemm twice instead. */
int fz_gemm_qkvt(const FzGemmDesc* desc, const void* x, const void* w, void* y, void* yt, int split_col, int64_t rows_per_frame,
                 int64_t yt_frame_stride, int64_t ldyt, void* stream);

/* The FeedForward CHAIN of the 64x64-level transformer block in ONE launch (csrc/ff_chain.hip): `ff(norm3(x)) + x` and the LayerNorm that
 * consumes it (attention.py:312-321, 327-331; ff = diffusers FeedForward [GEGLU(dim, 4 dim), Linear(4 dim, dim)] [3P]):
 *   val | gate = xn W1^T + b1 ;  h = val * gelu_erf(gate) ;  y = h W2^T + b2 (+ res) ;  y_ln = LayerNorm(y; gamma, beta, eps)  (y_ln may be NULL)
 * xn, res, y, y_ln: [rows][channels] fp16, contiguous rows; channels == 320; inner % 32 == 0 (SD-1.x: 1280).  The rows x inner intermediate
 * never exists outside registers.  Weights come PRE-PACKED: fz_ff_chain_pack writes W1 [2 inner][channels] (rows [val ; gate], diffusers
 * GEGLU.proj), b1 [2 inner] (or NULL) and W2 [channels][inner] into `packed` (fz_ff_chain_pack_bytes bytes, 16-byte aligned) as the MFMA
 * operand fragments the kernel streams; pack once per weight set.  Results are bit-identical to fz_gemm(FZ_GEMM_GEGLU) + fz_gemm_lnout.
 * fz_ff_chain_ok: 1 where the launch exists; fz_ff_chain_preferred: 1 where it is also the faster form on MI355X. */
int fz_ff_chain_ok(int64_t rows, int channels, int inner);
int fz_ff_chain_preferred(int64_t rows, int channels, int inner);
int64_t fz_ff_chain_pack_bytes(int channels, int inner);
int fz_ff_chain_pack(const void* w1, const void* b1, const void* w2, void* packed, int channels, int inner, void* stream);
int fz_ff_chain(const void* xn, const void* packed, const void* b2, const void* res, void* y, const void* ln_gamma, const void* ln_beta,
                float ln_eps, void* y_ln, int64_t rows, int channels, int inner, void* stream);

/* The CROSS-ATTENTION CHAIN of the 64x64-level transformer block in ONE launch (csrc/xattn_chain.hip): `attn2(norm2(x), context) + x` and the
 * LayerNorm that consumes it (attention.py:303-311 through the patched forward attention_register.py:71-128; diffusers CrossAttention [3P]):
 *   q = xn Wq^T ;  P = softmax(q_h K_h^T * scale) over the lk text keys ;  o_h = P V_h ;  y = o Wo^T + bias_out + res ;  y_ln = LayerNorm(y)
 * and with `front` != 0 the step in front of it as well (attention.py:295-301): x is then attn1's attention output and
 *   y1 = x Wo1^T + bias_out1 + res ;  xn = LayerNorm(y1; ln1_gamma, ln1_beta, ln1_eps) ;  ... ;  y = o Wo^T + bias_out + y1
 * (bias_out1, ln1_gamma, ln1_beta travel in the pack).
 * No controller reads or edits maps of more than 32 x 32 queries (attention_store.py:83, attention_util.py:104): this is the plain branch.
 * x, res, y, y_ln, y1: [rows][channels] fp16, contiguous rows; channels == 320, heads == 8; rows_per_frame % 128 == 0; frame n uses text context
 * n / frames_per_batch.  Operands come PRE-PACKED as the MFMA fragments the kernel streams: fz_xattn_chain_pack (Wq, Wo [320][320] and, for
 * `front`, Wo1 with its bias (or NULL) and the LayerNorm's gamma / beta -- wo1 == NULL packs the form without it;
 * fz_xattn_chain_pack_bytes(with_front) bytes) once per weight set; fz_xattn_chain_kv_pack
 * (K [batch][lk][320] and V^T [batch][320][>= 96] as fz_attn_cross takes them; fz_xattn_chain_kv_pack_bytes(batch) bytes) once per context.
 * Results are bit-identical to fz_gemm + fz_attn_cross(FZ_ATTN_FLASH) + fz_gemm_lnout (and fz_gemm_lnout in front).
 * fz_xattn_chain_ok: 1 where the launch exists; fz_xattn_chain_preferred: 1 where it is also the faster form on MI355X. */
typedef struct FzXattnChain {
    const void* x;          /* front == 0: LayerNorm'ed hidden states;  front != 0: attn1's attention output             */
    const void* res;        /* residual (front == 0: of attn2, may be NULL;  front != 0: of attn1, required)             */
    const void* packed;     /* fz_xattn_chain_pack                                                                        */
    const void* kv_packed;  /* fz_xattn_chain_kv_pack                                                                     */
    const void* bias_out;   /* [channels] or NULL                                                                         */
    void* y;
    void* y_ln;             /* or NULL                                                                                    */
    const void* ln_gamma;
    const void* ln_beta;
    void* y1;               /* front: attn1's result                                                                      */
    int64_t rows, rows_per_frame;
    int32_t frames_per_batch, channels, heads, lk;
    float scale;            /* softmax scale (head_dim ** -0.5)                                                           */
    float ln_eps, ln1_eps;
    int32_t front;
} FzXattnChain;
int fz_xattn_chain_ok(int64_t rows, int64_t rows_per_frame, int channels, int heads, int lk);
int fz_xattn_chain_preferred(int64_t rows, int64_t rows_per_frame, int channels, int heads, int lk);
int64_t fz_xattn_chain_pack_bytes(int with_front);
int64_t fz_xattn_chain_kv_pack_bytes(int batch);
int fz_xattn_chain_pack(const void* wq, const void* wo, const void* wo1, const void* bias_out1, const void* ln1_gamma, const void* ln1_beta,
                        void* packed, void* stream);
int fz_xattn_chain_kv_pack(const void* k, int64_t k_batch_stride, int64_t k_row_stride, const void* vt, int64_t vt_batch_stride,
                           int64_t vt_chan_stride, int batch, int lk, void* packed, void* stream);
int fz_xattn_chain(const FzXattnChain* desc, void* stream);

/* LayerNorm fused around fz_gemm (the `norm2 / norm3 / norm_temporal` + Linear pairs of SpatioTemporalTransformerBlock,
 * attention.py:295-337: `attn(norm(x)) + x`).  Two independent halves:
 *   stats_out  the GEMM that PRODUCES a LayerNorm input (out-projection + residual) also writes, per output row, the sum and the
 *              sum of squares of every 64-column block of what it stores: float [batch * rows][out_features / 64][2].
 *              Plain epilogue, out_features % 64 == 0, 16-byte aligned rows.  When the library splits K for this shape the
 *              statistics are NOT written and the call returns FZ_GEMM_NO_STATS (> 0; y is complete): run fz_layernorm.
 *   stats_in   the GEMM that CONSUMES a LayerNorm output reads the RAW input x instead, with w = gamma (.) W (fp16),
 *              c1[o] = sum_k w[o][k], c0[o] = sum_k beta[k] W[o][k] + bias[o] (float; the `bias` argument is ignored):
 *              y = rstd (x . w - mean c1) + c0, mean / rstd from stats_in [batch * rows][in_features / 64][2] summed in a
 *              fixed order.  Plain or GEGLU epilogue (c1 / c0 in the packed row order), in_features and out_features % 64 == 0. */
#define FZ_GEMM_NO_STATS 1
typedef struct FzGemmLn {
    const float* stats_in;
    const float* c1;
    const float* c0;
    float eps;
    int32_t reserved0;
    float* stats_out;
} FzGemmLn;
int fz_gemm_ln(const FzGemmDesc* desc, const FzGemmLn* ln, const void* x, const void* w, const void* bias, const void* res,
               const void* res2, void* y, void* workspace, void* stream);

/* 3x3 convolution (pad 1) of PseudoConv3d's spatial part (resnet.py:57-64) on token-major activations, as an MFMA
 * implicit GEMM with the elementwise tail fused: y = conv(x) + bias (+ temb[n / frames_per_batch]) (+ res).
 * x: [n][hi][wi][cin]; wt: weights packed [cout][3*3][cin]; y / res: [n][ho][wo][cout]; temb: n/frames_per_batch rows of
 * cout values, temb_stride elements apart.
 * stride 1 or 2; upsample != 0 reads x through a nearest-2x upsampling (UpsamplePseudo3D, resnet.py:145) without
 * materialising it.  cin % 8 == 0 (MFMA path, any cout) or cin < 8 with cout % 8 == 0 (conv_in: direct convolution).
 * workspace / workspace_floats / tile_cfg / split_k: as for fz_gemm (split-K is what fills the chip at the 16x16 and 8x8
 * levels); workspace may be NULL.  With tile_cfg = 0 the library picks the data path as well: the implicit GEMM, or -- stride 1, no
 * upsampling, whole 256-pixel tiles (a frame of whole tiles, or whole frames per tile), cin % 64 == 0, cout % 160 == 0 -- the kernel
 * that keeps the pixel rows + halo resident in LDS across the nine taps (csrc/conv_halo.hip), whole or in K slices with the split-K
 * tail; tile_cfg 154299 pins that kernel (split_k = its K slices), 154264 its narrow form (64 output channels per workgroup,
 * cout % 64 == 0); both return FZ_ERR_UNSUPPORTED for shapes they do not carry. */
int fz_conv3x3(const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride, const void* res,
               void* y, int n, int hi, int wi, int cin, int cout, int stride, int upsample, int frames_per_batch,
               void* workspace, int64_t workspace_floats, int tile_cfg, int split_k, void* stream);

/* Nearest-2x upsampling + 3x3 convolution (UpsamplePseudo3D, /root/reference/video_diffusion/models/resnet.py:145 + :57-64) as FOUR 2x2
 * convolutions of the low-resolution input, one per output parity: U[y][x] = X[y >> 1][x >> 1] makes the three taps of a kernel row hit only
 * two input rows, so 4 instead of 9 multiply-adds per output and input channel -- on weights summed ONCE:
 *   wt_up[z = 2 py + px][co][t = 2 ty + tx][ci] = sum of wt[co][3 ky + kx][ci] over ky in R(py, ty), kx in R(px, tx),
 *   R(0, 0) = {0}, R(0, 1) = {1, 2}, R(1, 0) = {0, 1}, R(1, 1) = {2}      (fp32 sums in (ky, kx) order, one rounding to fp16)
 * and y[n][2 r + py][2 c + px][co] = bias[co] + sum_{t, ci} wt_up[z][co][t][ci] x[n][r + ty + py - 1][c + tx + px - 1][ci] (zero outside).
 * fz_conv3x3_up2_pack builds wt_up (fz_conv3x3_up2_pack_halves(cin, cout) halves) from fz_conv3x3's packed weights [cout][9][cin].
 * x: [n][h][w][cin]; y: [n][2 h][2 w][cout].  fz_conv3x3_up2_ok: the shapes the kernel carries (8 <= w <= 128 and 256 % w == 0, frames of whole
 * 256-pixel tiles or whole frames per tile, cin % 64 == 0, cout % 160 == 0); elsewhere fz_conv3x3(..., upsample = 1) is the path.  Against that path the
 * result differs by the fp16 rounding of the summed weights (both are checked against the same fp32 reference). */
int fz_conv3x3_up2_ok(int n, int h, int w, int cin, int cout);
int fz_conv3x3_up2_preferred(int n, int h, int w, int cin, int cout);   /* ... and where it is measured faster than fz_conv3x3(upsample = 1) */
int64_t fz_conv3x3_up2_pack_halves(int cin, int cout);
int fz_conv3x3_up2_pack(const void* wt, void* wt_up, int cin, int cout, void* stream);
int fz_conv3x3_up2(const void* x, const void* wt_up, const void* bias, void* y, int n, int h, int w, int cin, int cout, void* stream);

/* Temporal k=3 convolution over the frame axis (the two Conv1d of LoRALinearLayer, lora.py:31-54, applied on
 * '(b h w) c f'): y[n][tok][co] = sum_{t=0..2, ci} x[n - f + (f+t-1)][tok][ci] * wt[co][t][ci] (+ res), zero padded at the
 * clip ends (f = n % clip_len).  x: [n][tokens][cin]; wt: [cout][3][cin]; y/res/res2: [n][tokens][cout];
 * temb (optional): [n/clip_len] rows of cout values, temb_stride elements apart (ResnetBlock's time embedding add,
 * resnet.py:366-376, fused behind the temporal conv).  Also the plain nn.Conv1d temporal convolution of configs without a
 * `lora` key (resnet.py:42-55; its bias rides in `temb`).  cin % 8 == 0 and cout >= 8: MFMA implicit GEMM; cin, cout <= 8
 * (conv_out: 4 -> 2 -> 4 and 4 -> 4 channels): a direct VALU kernel.
 * workspace / workspace_floats: optional split-K scratch as for fz_gemm (the rank-160 projection at the small levels has too
 * few output tiles to fill the chip otherwise). */
int fz_temporal_conv3(const void* x, const void* wt, const void* res, const void* res2, const void* temb,
                      int64_t temb_stride, void* y, int n, int tokens, int cin, int cout, int clip_len, void* workspace,
                      int64_t workspace_floats, void* stream);

/* The temporal LoRA pair of a PseudoConv3d in ONE launch (lora.py:31-54 LoRALinearLayer.forward: `up(down(x)) + x`, called from
 * resnet.py:57-80): y = conv1d_3(conv1d_3(x, w_down), w_up) + x (+ temb per clip) (+ res2), both convolutions zero padded over the
 * frame axis of each clip; the rank-`rank` intermediate is rounded to fp16 once and never leaves the workgroup's LDS.  Bit-identical
 * to fz_temporal_conv3 called twice (without split-K).  x, res2, y: [n][tokens][channels]; w_down [rank][3][channels];
 * w_up [channels][3][rank]; temb as for fz_temporal_conv3.  fz_lora_pair_ok: 1 where the launch exists (rank == 160,
 * channels % 320 == 0, channels <= 1280, clip_len divides 128, tokens % (128 / clip_len) == 0); fz_lora_pair returns
 * FZ_ERR_UNSUPPORTED elsewhere -- call fz_temporal_conv3 twice. */
int fz_lora_pair_ok(int n, int tokens, int channels, int rank, int clip_len);
/* 1 where the one launch is also the FASTER form on MI355X (it needs >= 256 workgroups of 128 rows: n * tokens >= 32768) */
int fz_lora_pair_preferred(int n, int tokens, int channels, int rank, int clip_len);
int fz_lora_pair(const void* x, const void* w_down, const void* w_up, const void* temb, int64_t temb_stride, const void* res2, void* y,
                 int n, int tokens, int channels, int rank, int clip_len, void* stream);
/* fz_lora_pair that ALSO writes the Welford partials (count, mean, M2) of y's GroupNorm statistics (resnet.py:338,369: the norm that follows
 * the convolution), for fz_groupnorm_from_partials: gn_partial[n][gn_groups][chunks][3] floats with chunks = fz_lora_pair_gn_chunks(...)
 * records per (frame, group) -- one per min(64, 128 / clip_len)-row piece of a workgroup's tile; 0 where the form does not exist (group
 * width other than 10 / 20 channels, clip_len > 32), and fz_lora_pair_gn then returns FZ_ERR_UNSUPPORTED.  y is bit-identical to fz_lora_pair's. */
int fz_lora_pair_gn_chunks(int n, int tokens, int channels, int rank, int clip_len, int gn_groups);
int fz_lora_pair_gn(const void* x, const void* w_down, const void* w_up, const void* temb, int64_t temb_stride, const void* res2, void* y,
                    int n, int tokens, int channels, int rank, int clip_len, float* gn_partial, int gn_groups, void* stream);

/* GroupNorm statistics out of the PRODUCING launch's epilogue (resnet.py:338,369 norm1 / norm2, attention.py:110 `self.norm`,
 * unet_3d_condition.py:439 conv_norm_out: each normalises what a projection or a temporal convolution just stored): fz_gemm_gn /
 * fz_temporal_conv3_gn are fz_gemm (plain epilogue, no batch) / fz_temporal_conv3 that ALSO write the Welford partials (count, mean, M2)
 * of the stored tensor per (frame, group, 128-row chunk): gn_partial[frames][gn_groups][rows_per_frame / 128][3] floats (fz_temporal_conv3_gn:
 * rows_per_frame = tokens); fz_groupnorm_from_partials then normalises without a statistics pass over the tensor.  Deterministic (fixed
 * summation order).  Returns 0 when the partials were written, FZ_GEMM_NO_STATS (> 0) when the launch the library picks for this shape
 * cannot produce them (tile narrower than 320 columns, split-K, rows_per_frame % 128 != 0, out_features % 320 != 0, odd group width ...):
 * y is complete, gn_partial untouched -- run fz_groupnorm. */
int fz_gn_epilogue_chunks(int64_t rows_per_frame); /* rows_per_frame / 128, or 0 when the epilogue form does not apply */
int fz_gemm_gn(const FzGemmDesc* desc, const void* x, const void* w, const void* bias, const void* res, const void* res2, void* y,
               float* gn_partial, int gn_groups, int64_t rows_per_frame, void* stream);
int fz_temporal_conv3_gn(const void* x, const void* wt, const void* res, const void* res2, const void* temb, int64_t temb_stride, void* y,
                         int n, int tokens, int cin, int cout, int clip_len, void* workspace, int64_t workspace_floats,
                         float* gn_partial, int gn_groups, void* stream);
/* GroupNorm (+SiLU) from partials with `partial_chunks` records per (frame, group) -- fz_groupnorm_apply with an explicit record count
 * (the partials of fz_groupnorm_stats have fz_groupnorm_chunks(tokens, channels) of them): partial [n_frames][groups][partial_chunks][3],
 * frame n uses the statistics of frames [n / span * span, + span); stats: float scratch [n_frames / span][groups][2]. */
int fz_groupnorm_from_partials(const void* x, void* y, const void* gamma, const void* beta, int n_frames, int span, int tokens,
                               int channels, int groups, float eps, int silu, const float* partial, int partial_chunks, float* stats,
                               void* stream);

/* fz_gemm that ALSO writes the LayerNorm of the rows it stores: every `x = f(norm(x)) + x` step of SpatioTemporalTransformerBlock.forward
 * (attention.py:295-337) ends in a Linear + residual whose output is the NEXT LayerNorm's input (attn1.to_out -> norm2, attn2.to_out -> norm3,
 * ff.net[2] -> norm_temporal; proj_in -> norm1).  Where the launch the library picks for the shape is a 320-wide tile that holds WHOLE rows
 * (out_features == 320: the 64x64 level), the epilogue computes exact two-sweep row statistics on the fp16 values it stored and writes
 *   y_ln[row][o] = (y[row][o] - mean_row) * rstd_row * gamma[o] + beta[o]        (row stride ld_ln)
 * beside y: the LayerNorm launch and its read of y are gone.  Returns 0 when y_ln was written, FZ_GEMM_NO_STATS (> 0) when the launch cannot
 * produce it (other widths, split-K, narrower tiles): y is complete, y_ln untouched -- run fz_layernorm.  workspace as for fz_gemm. */
int fz_gemm_lnout(const FzGemmDesc* desc, const void* x, const void* w, const void* bias, const void* res, const void* res2, void* y,
                  const void* gamma, const void* beta, float eps, void* y_ln, int64_t ld_ln, void* workspace, void* stream);

/* LayerNorm over channels, rows = tokens (attention.py:193-233). gamma/beta fp16. */
int fz_layernorm(const void* x, void* y, const void* gamma, const void* beta, int64_t rows, int channels,
                 float eps, void* stream);

/* GEGLU gate (diffusers FeedForward): y[r][c] = x[r][c] * gelu_erf(x[r][inner + c]). */
int fz_geglu(const void* x, void* y, int64_t rows, int inner, void* stream);

/* Row softmax of an fp16 score matrix in fp32 arithmetic: y[r][:cols] = softmax(scale * x[r][:cols]); cols, ldx, ldy multiples
 * of 8.  Replaces the fp32 softmax of diffusers' AttentionBlock [3P, diffusers 0.11.1 models/attention.py] inside the VAE that
 * p2p_ddim_spatial_temporal.py:94-96 (encode) and stable_diffusion.py:297-319 (decode) call. */
int fz_softmax_rows(const void* x, void* y, int64_t rows, int cols, int64_t ldx, int64_t ldy, float scale, void* stream);

/* out[n][c][lp] = in[n][l][c] transposed, zero padded l -> lp (V^T operand of the attention kernels). */
int fz_transpose_pad(const void* in, void* out, int n, int l, int c, int64_t in_frame_stride, int64_t in_row_stride,
                     int lp, void* stream);

/* Fused latent update (p2p_ddim_spatial_temporal.py:150-161 inverse step; :400-407 CFG + DDIMScheduler.step):
 *   eps = eps_u + g (eps_c - eps_u)   (eps_c only when eps_u == NULL)
 *   z   = cz * z + ce * eps           [+ latent blend z = inv + mask (z - inv), spatial_blend.py:121]
 * z: float [4][F][hw] (b c f h w, b = 1) updated in place; eps: fp16 token-major [F][hw][4];
 * next_in: fp16 token-major [F][hw][4] = the next UNet input (may be NULL). */
int fz_latent_update(float* z, const void* eps_u, const void* eps_c, float guidance, float cz, float ce,
                     const float* inv, const float* mask, void* next_in, int frames, int hw, void* stream);

/* fp16 running-sum helper for the edit controller's accumulated cross maps (attention_store.py:95-101):
 * acc (float) += x (fp16), n elements. */
int fz_accumulate(float* acc, const void* x, int64_t n, void* stream);

/* One-sided exchanges between the GPUs that share ONE frame-sharded clip (csrc/peer.hip): every rank owns a symmetric
 * heap its peers have mapped (hipIpc over xGMI).  The couplings they carry: 5-D GroupNorm statistics (resnet.py:338,369), the halo
 * frames of the k=3 temporal convolutions (lora.py:31-54), K / V^T of neighbour and anchor frames (attention.py:374-388), K | V of the
 * temporal attention (attention.py:327-337).
 *   fz_peer_put   copy `bytes` (a multiple of 16; src and every dst 16-byte aligned) from src to dst[0..n_dst) -- addresses inside the
 *                 peers' (or the own) heap --, make them visible system-wide, then store `epoch` to flags[0..n_dst): the flag word each
 *                 receiver reserves for THIS sender.  done_counter: a zero-initialised uint32 in local memory, one per put in flight.
 *   fz_peer_wait  one workgroup polling flags[r] for every bit r of sender_mask until (int32)(flags[r] - epoch) >= 0; kernels queued
 *                 behind it on `stream` may read what those senders put.  timeout_us > 0 bounds the spin: on expiry err[0] = 1.
 * Epochs only grow (wrap-safe comparison); flags are never reset. */
int fz_peer_put(const void* src, int64_t bytes, void* const* dst, uint32_t* const* flags, int n_dst, uint32_t epoch,
                uint32_t* done_counter, void* stream);
int fz_peer_wait(const uint32_t* flags, uint64_t sender_mask, uint32_t epoch, uint32_t* err, int64_t timeout_us, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------------
 * Native issue plans.  The reference issues a UNet forward by walking its module tree in Python at every DDIM step
 * (`self.unet(latent_model_input, t, encoder_hidden_states=...)`, p2p_ddim_spatial_temporal.py:286 / stable_diffusion.py:360-372); the launch
 * list that walk produces is a pure function of (clip geometry, controller plan).  A plan is that list recorded once: between
 * fz_plan_begin and fz_plan_end every launch the library makes on behalf of ANY fz_* entry point is issued as usual and appended to the plan
 * (kernel, grid, block, LDS bytes, a byte copy of the kernel arguments); fz_plan_replay re-issues records [first, first + count) on
 * `stream` with one call.  What changes between steps is data behind pointers: fz_plan_relocate rewrites, inside the argument bytes of
 * records [first, first + count), every pointer-sized, pointer-aligned word that points into [old_base, old_base + nbytes) to the same
 * offset from new_base, and returns the number of words rewritten (< 0: bad arguments).  The match is by VALUE: an argument word that is not
 * a pointer but happens to hold a number inside the range would move too -- the ranges are device (or pinned host) allocations, and no size, stride,
 * count or scale argument of this library takes values up there (2^46 and beyond).  fz_plan_pause(p, 1) ... (p, 0) brackets host work
 * whose launches must NOT enter the plan (what the host repeats live at every replay: the attention controller's own kernels).
 * The recorder is PER THREAD (thread-local; the library keeps no process-wide state): one recording at a time per thread, it sees the launches
 * that thread makes, other threads' launches are issued unrecorded.  A plan never allocates device memory, never synchronises, and owns
 * nothing but host memory (the buffers its records point into are the host layer's to keep alive: fatezero_amd/issue.py).
 * Returns FZ_OK or a negative FZ_ERR_* code like every other entry point. */
typedef struct FzPlan FzPlan;
int fz_plan_begin(FzPlan** out);
int fz_plan_pause(FzPlan* plan, int paused);
int fz_plan_end(FzPlan* plan);
int64_t fz_plan_launches(const FzPlan* plan);
int64_t fz_plan_relocate(FzPlan* plan, int64_t first, int64_t count, const void* old_base, int64_t nbytes, const void* new_base);
int fz_plan_replay(const FzPlan* plan, int64_t first, int64_t count, void* stream);
/* All records with ONE runtime call: the plan as an executable hipGraph (a chain of kernel nodes in record order, built and instantiated at the
 * first call); records whose arguments fz_plan_relocate changed since the previous launch are refreshed first
 * (hipGraphExecKernelNodeSetParams).  For a forward whose controller events can all be taken before the first launch. */
int fz_plan_graph_launch(FzPlan* plan, void* stream);
void fz_plan_destroy(FzPlan* plan);

const char* fz_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FATEZERO_HIP_H */
