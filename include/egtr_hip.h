/*
 * egtr_hip.h -- C ABI of libegtr_hip.so: the MI355X (gfx950) hot path of EGTR scene-graph generation.
 *
 * Plain C: raw device pointers, sizes and a HIP stream handle; no torch / ATen types.  Every entry point
 * enqueues on the given stream and returns immediately (no host sync, no allocation, re-entrant, no global
 * state).  Return value: 0 on success, a negative EGTR_E_* code otherwise (egtr_status_string() names it;
 * launch errors are returned, not printf'd as the reference does at ms_deform_im2col_cuda.cuh:948-952).
 *
 * Reference interfaces replaced (all paths relative to naver-ai/egtr):
 *   egtr_msda_forward_*   <- ms_deformable_im2col_cuda()   model/custom_kernel/cuda/ms_deform_im2col_cuda.cuh:924-955
 *                            as driven by ms_deform_attn_cuda_forward(), cuda/ms_deform_attn_cuda.cu:23-83,
 *                            exported to Python as ms_deform_attn_forward (vision.cpp:13, ms_deform_attn.h:20-39)
 *   egtr_msda_backward_*  <- ms_deformable_col2im_cuda()   ms_deform_im2col_cuda.cuh:957-1327
 *                            as driven by ms_deform_attn_cuda_backward(), cuda/ms_deform_attn_cuda.cu:86-156,
 *                            exported as ms_deform_attn_backward (vision.cpp:14, ms_deform_attn.h:41-61)
 *   egtr_self_attn_*      <- the bmm / softmax / bmm core of DeformableDetrMultiheadAttention.forward,
 *                            model/deformable_detr.py:1170-1253 (plain PyTorch in the reference), including the
 *                            retained per-layer scaled-Q / K maps of :1179-1185
 *   egtr_rel_head_*       <- the pairwise gate + gated sum + two 3-layer MLPs of
 *                            DetrForSceneGraphGeneration.forward, model/egtr.py:366-416 (plain PyTorch)
 *
 * Layout conventions (row-major, innermost last), identical to the reference's tensors:
 *   value          [B, S, M, D]        spatial_shapes [L, 2] int64 (H, W), DEVICE memory
 *   sampling_loc   [B, Lq, M, L, P, 2] level_start_index [L] int64, DEVICE memory
 *   attn_weight    [B, Lq, M, L, P]    out [B, Lq, M*D]
 * The reference chunks the batch by im2col_step (cu:53-64); these kernels take the whole batch in one launch,
 * so the binding accepts and ignores im2col_step.
 */
#ifndef EGTR_HIP_H
#define EGTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* egtr_stream_t; /* a hipStream_t; NULL = the default stream */

enum {
  EGTR_OK = 0,
  EGTR_E_ARG = -1,      /* null pointer / non-positive size / unsupported shape */
  EGTR_E_LAUNCH = -2,   /* hipGetLastError() reported a launch failure */
  EGTR_E_UNSUPPORTED = -3
};

/* Bumped whenever an entry point is added / removed or the meaning of an argument changes; egtr_amd/_lib.py refuses a
 * library whose number differs from the one it was written against. */
#define EGTR_ABI_VERSION 5
int egtr_abi_version(void);
const char* egtr_status_string(int status);
/* last HIP error string seen by this thread (for EGTR_E_LAUNCH) */
const char* egtr_last_hip_error(void);

/* ---- multi-scale deformable attention: sample + weighted sum over L levels x P points ------------------- */
/* out must hold B*Lq*M*D elements; it is fully overwritten (no zero-init needed, unlike cu:57). */
int egtr_msda_forward_f32(egtr_stream_t stream, const float* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const float* sampling_loc, const float* attn_weight,
                          int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                          int num_point, float* out);

/* Forward with the prologue of DeformableDetrMultiscaleDeformableAttention.forward fused in
 * (model/deformable_detr.py:1055-1073, 2-d reference points): sampling_offsets [B,Lq,M,L,P,2] and attn_logits
 * [B,Lq,M,L*P] are the raw outputs of the two Linear layers, reference_points is [B,Lq,L,2]; the kernel forms
 * loc = ref + offset / (W_l, H_l) and softmax(logits) itself.  attn_weight_out (optional, [B,Lq,M,L*P]) receives the
 * softmaxed weights.  ld_offsets / ld_logits: floats between consecutive queries (256 / 128 when dense; larger when
 * the two are column blocks of one wider Linear output).  keep_mask (optional, [B,S] bytes, non-zero = valid token)
 * or keep_bits (optional, [B, ceil(S/32)] words, bit s%32 of word s/32 -- as written by egtr_level_geometry_f32;
 * takes precedence, kept in LDS): padded tokens are skipped in the sum, which equals zeroing their value rows
 * (deformable_detr.py:1050-1052).
 * Only M = 8, D = 32, L*P = 16, P even; EGTR_E_UNSUPPORTED otherwise (compose the prologue on the host and call
 * egtr_msda_forward_f32).
 * value_bias (optional, [M*D]): the value_proj bias applied inside the kernel -- `value` is then the bias-free projection
 * W x of the (unmasked) encoder states; out = sum_s w_s v_s + value_bias * sum_s w_s over the in-range, unpadded corner
 * weights, identical to sampling (W x + b) with padded rows zeroed (deformable_detr.py:1048-1052), without the bias / mask
 * pass over the [S, 256] value tensor. */
int egtr_msda_forward_fused_vbias_f32(egtr_stream_t stream, const float* value, const int64_t* spatial_shapes,
                                      const int64_t* level_start_index, const float* sampling_offsets,
                                      const float* attn_logits, const float* reference_points, int batch,
                                      int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                      int num_point, float* out, float* attn_weight_out, int ld_offsets, int ld_logits,
                                      const unsigned char* keep_mask, const unsigned* keep_bits,
                                      const float* value_bias);

/* The same with 4-d reference BOXES [B, Lq, L, 4] = (cx, cy, w, h) scaled by the valid ratios: sampling location =
 * box.xy + offset / num_point * box.wh * 0.5 (model/deformable_detr.py:1074-1081) -- the form the decoder's
 * cross-attention takes from its second layer on under iterative box refinement (dd:1903-1918, egtr.py:148-154). */
int egtr_msda_forward_fused_box_f32(egtr_stream_t stream, const float* value, const int64_t* spatial_shapes,
                                    const int64_t* level_start_index, const float* sampling_offsets,
                                    const float* attn_logits, const float* reference_boxes, int batch,
                                    int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                    int num_point, float* out, float* attn_weight_out, int ld_offsets, int ld_logits,
                                    const unsigned char* keep_mask, const unsigned* keep_bits,
                                    const float* value_bias);

/* grad_value [B,S,M,D] MUST be zero-initialised by the caller (accumulated with atomics, as cu:124 relies on);
 * grad_sampling_loc [B,Lq,M,L,P,2] and grad_attn_weight [B,Lq,M,L,P] are fully overwritten. */
int egtr_msda_backward_f32(egtr_stream_t stream, const float* grad_out, const float* value,
                           const int64_t* spatial_shapes, const int64_t* level_start_index,
                           const float* sampling_loc, const float* attn_weight, int batch, int spatial_size,
                           int num_heads, int channels, int num_levels, int num_query, int num_point,
                           float* grad_value, float* grad_sampling_loc, float* grad_attn_weight);

/* The same backward with ALL three gradients written by the call: grad_value need not be initialised -- what the reference's
 * op does one level up, where `ms_deform_attn_cuda_backward` allocates `at::zeros_like(value)` itself
 * (model/custom_kernel/cuda/ms_deform_attn_cuda.cu:124).  Encoder-shaped calls (num_query == spatial_size) clear grad_value
 * inside the first kernel of the pair instead of in a 4 * B * S * 256-byte fill launch; other shapes run a memset on the stream. */
int egtr_msda_backward_out_f32(egtr_stream_t stream, const float* grad_out, const float* value,
                               const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* sampling_loc, const float* attn_weight, int batch, int spatial_size,
                               int num_heads, int channels, int num_levels, int num_query, int num_point,
                               float* grad_value, float* grad_sampling_loc, float* grad_attn_weight);

/* float64 forward / backward: the reference extension dispatches AT_DISPATCH_FLOATING_TYPES
 * (model/custom_kernel/cuda/ms_deform_attn_cuda.cu:67, 137), so double callers (gradcheck) are served too -- by the
 * generic one-thread-per-element kernels, any (M, D, L, P).  Same argument meaning as the f32 entries; grad_value must be
 * zero-initialised. */
int egtr_msda_forward_f64(egtr_stream_t stream, const double* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const double* sampling_loc, const double* attn_weight,
                          int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                          int num_point, double* out);
int egtr_msda_backward_f64(egtr_stream_t stream, const double* grad_out, const double* value,
                           const int64_t* spatial_shapes, const int64_t* level_start_index, const double* sampling_loc,
                           const double* attn_weight, int batch, int spatial_size, int num_heads, int channels,
                           int num_levels, int num_query, int num_point, double* grad_value, double* grad_sampling_loc,
                           double* grad_attn_weight);

/* bf16 backward (we add bf16; the reference has no half kernel): grad_out [B,Lq,M*D] and value [B,S,M,D] are raw
 * bfloat16, sampling_loc / attn_weight and all three gradients fp32 (grad_value zero-initialised by the caller).
 * For M = 8, D = 32, L * P = 16 the kernels read the bf16 operands directly (widened on load: the same results as widening
 * first) and workspace may be NULL; other shapes widen them into workspace.  egtr_msda_backward_bf16_workspace_floats
 * returns the floats a call needs: 0, or B*S*M*D + B*Lq*M*D. */
long long egtr_msda_backward_bf16_workspace_floats(int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                                   int num_query, int num_point);
int egtr_msda_backward_bf16(egtr_stream_t stream, const uint16_t* grad_out, const uint16_t* value,
                            const int64_t* spatial_shapes, const int64_t* level_start_index, const float* sampling_loc,
                            const float* attn_weight, int batch, int spatial_size, int num_heads, int channels,
                            int num_levels, int num_query, int num_point, float* grad_value, float* grad_sampling_loc,
                            float* grad_attn_weight, float* workspace);

/* bf16 storage (uint16_t = raw bfloat16 bits), fp32 accumulation.  The reference dispatches float/double only
 * (cu:67,137); this is the added path for the bf16 stress configuration.  loc / attn stay fp32; the per-corner sample weight
 * (bilinear x attention, formed in fp32) is rounded to bf16 as the second operand of v_dot2_f32_bf16. */
int egtr_msda_forward_bf16(egtr_stream_t stream, const uint16_t* value, const int64_t* spatial_shapes,
                           const int64_t* level_start_index, const float* sampling_loc, const float* attn_weight,
                           int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                           int num_point, uint16_t* out);

/* bf16 forward with the prologue fused in (see egtr_msda_forward_fused_vbias_f32): sampling_offsets / attn_logits /
 * reference_points are raw bf16 tensors ([B,Lq,M,L,P,2] / [B,Lq,M,L*P] with row strides ld_offsets / ld_logits elements,
 * [B,Lq,L,2]); softmax and sampling locations are formed in fp32 inside the kernel; padded tokens are skipped: keep_mask
 * (optional, [B,S] bytes, 0 = padded) or -- preferred, it wins when both are given -- keep_bits ([B, ceil(S / 32)] words, one
 * bit per token, staged in LDS by the kernel for S <= 32768). */
int egtr_msda_forward_fused_bf16(egtr_stream_t stream, const uint16_t* value, const int64_t* spatial_shapes,
                                 const int64_t* level_start_index, const uint16_t* sampling_offsets,
                                 const uint16_t* attn_logits, const uint16_t* reference_points, int batch,
                                 int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                 int num_point, uint16_t* out, int ld_offsets, int ld_logits,
                                 const unsigned char* keep_mask, const unsigned* keep_bits);

/* ---- decoder multi-head self-attention core ------------------------------------------------------------- */
/* q (already scaled by D^-1/2, dd:1166), k, v: [B, N, M*D] as produced by the q/k/v projections.
 * out [B, N, M*D] = softmax(q k^T) v per head (dd:1190-1253, no mask, attention_dropout = 0).
 * q_heads / k_heads (optional, may be NULL): the retained maps [B, M, N, D] of dd:1179-1185.
 * lse (optional): [B, M, N] log-sum-exp of each score row, saved for the backward. */
int egtr_self_attn_forward_f32(egtr_stream_t stream, const float* q, const float* k, const float* v, int batch,
                               int num_query, int num_heads, int head_dim, float* out, float* q_heads,
                               float* k_heads, float* lse);

/* The same on raw bfloat16 tensors (the bf16 model's decoder at inference): fp32 arithmetic, out rounded to nearest even, the
 * retained maps bit copies of q / k.  16-byte aligned operands. */
int egtr_self_attn_forward_bf16(egtr_stream_t stream, const uint16_t* q, const uint16_t* k, const uint16_t* v, int batch,
                                int num_query, int num_heads, int head_dim, uint16_t* out, uint16_t* q_heads,
                                uint16_t* k_heads);

/* grads wrt q, k, v ([B, N, M*D] each, fully overwritten) from grad_out [B,N,M*D], the forward's out and lse. */
int egtr_self_attn_backward_f32(egtr_stream_t stream, const float* q, const float* k, const float* v,
                                const float* out, const float* lse, const float* grad_out, int batch,
                                int num_query, int num_heads, int head_dim, float* grad_q, float* grad_k,
                                float* grad_v);

/* The same with gradients that reach q / k by another route added in the epilogue (optional, [B, N, M*D] each): the retained
 * scaled-q / k maps of dd:1179-1185 feed the relation head, whose gradient meets the attention's at q and k (training node
 * of a decoder layer, egtr_amd.ops.DecoderLayerTrainFunction). */
int egtr_self_attn_backward_acc_f32(egtr_stream_t stream, const float* q, const float* k, const float* v,
                                    const float* out, const float* lse, const float* grad_out, int batch,
                                    int num_query, int num_heads, int head_dim, float* grad_q, float* grad_k,
                                    float* grad_v, const float* grad_q_add, const float* grad_k_add);

/* ---- skinny linear layer (object-query rows) --------------------------------------------------------------- */
/* y[M,N] = act((x[M,K] . w[N,K]^T + bias[N]) * alpha); bias may be NULL; relu != 0 applies max(.,0) last.
 * Replaces the nn.Linear calls of the decoder layers (model/deformable_detr.py:1132-1135, 990-995, 1386-1387),
 * the detection heads (model/egtr.py:128-134) and the relation-head projections (model/egtr.py:196-209) for the
 * M = num_queries regime, where the vendor GEMM is latency-bound on one workgroup.  Requires K % 64 == 0
 * (EGTR_E_UNSUPPORTED otherwise); exact-f32 MFMA. */
int egtr_linear_f32(egtr_stream_t stream, const float* x, const float* w, const float* bias, float* y, int M, int K,
                    int N, float alpha, int relu);

/* Backward of egtr_linear_f32 in ONE launch (training; autograd of the same nn.Linear call sites): with
 * g' = alpha * grad_y * [relu_output > 0] (relu_output: the layer's post-ReLU output, NULL without ReLU) writes
 * grad_x [M, K] = g' . w, grad_w [N, K] = g'^T . x and grad_bias [N] = column sums of g' (each output may be NULL).
 * Exact-f32 MFMA.  K % 64 == 0 and N % 64 == 0, 16-byte aligned contiguous operands, else EGTR_E_UNSUPPORTED. */
int egtr_linear_backward_f32(egtr_stream_t stream, const float* grad_y, const float* relu_output, const float* x,
                             const float* w, float alpha, float* grad_x, float* grad_w, float* grad_bias, int M, int K,
                             int N);

/* The same with up to two [M, K] tensors added to grad_x in the epilogue (NULL = none; may alias grad_x): where the gradient
 * branches of a layer meet (x feeds q / k / v projections AND the residual), the sum is formed by the product that
 * arrives last instead of by add launches (training node of a decoder layer). */
int egtr_linear_backward_acc_f32(egtr_stream_t stream, const float* grad_y, const float* relu_output, const float* x,
                                 const float* w, float alpha, float* grad_x, float* grad_w, float* grad_bias, int M,
                                 int K, int N, const float* grad_x_add1, const float* grad_x_add2);

/* Up to 16 independent skinny linears in ONE launch (every launch costs ~5 us in a graph-replayed forward):
 *   y_g[M_g, N_g] (row stride ldy_g floats) = act((alpha_x_g * X_g W_g^T + b_g) * alpha_g),  X_g [M_g, K], W_g [N_g, K].
 * All arrays are HOST arrays of num_groups entries (pointers are device pointers; bias[g] may be NULL); K % 64 == 0. */
int egtr_linear_grouped_f32(egtr_stream_t stream, int num_groups, const float* const* x, const float* const* w,
                            const float* const* bias, float* const* y, const int* M, const int* N, const int* ldy,
                            const float* alpha_x, const float* alpha, const int* relu, int K);

/* egtr_linear_grouped_f32 with a LayerNorm PROLOGUE per group (K == 256): where ln_gamma[g] != NULL the layer's input is
 *   LayerNorm(x_g + ln_residual_g) * ln_gamma_g + ln_beta_g  [+ pos_g[row % pos_rows_g]]       (rows of 256 channels)
 * -- the decoder's "residual add + LayerNorm" (model/deformable_detr.py:1437-1438, 1456-1457, 1466-1468) evaluated by its
 * consumer instead of in a launch of its own -- and ln_out[g] (may be NULL) receives the LayerNorm result (without pos),
 * [M_g, 256] contiguous, written by the workgroups of output-column tile 0.  Groups with ln_gamma[g] == NULL behave as in
 * egtr_linear_grouped_f32; ln_gamma == NULL: no group has a prologue (the other ln_* / pos arrays are then ignored). */
int egtr_linear_grouped_ln_f32(egtr_stream_t stream, int num_groups, const float* const* x, const float* const* w,
                               const float* const* bias, float* const* y, const int* M, const int* N, const int* ldy,
                               const float* alpha_x, const float* alpha, const int* relu, int K,
                               const float* const* ln_residual, const float* const* ln_gamma, const float* const* ln_beta,
                               const float* ln_eps, const float* const* pos, const int* pos_rows, float* const* ln_out);

/* ---- one decoder layer in ONE launch (inference, fp32) ---------------------------------------------------------------
 * model/deformable_detr.py:1390-1489 (DeformableDetrDecoderLayer.forward: self-attention with position rows on q / k
 * :1107-1262, residual + LayerNorm, multi-scale deformable cross-attention :1026-1104, residual + LayerNorm, fc1 / ReLU /
 * fc2, residual + LayerNorm; dropout is the identity at inference) followed by the NEXT layer's scaled-q / k / v
 * projections (:1161-1168), for d_model 256, 8 heads, 4 levels x 4 points, 1024 hidden units, 2-d reference points.
 * The layer's q (already scaled by D^-1/2, the retained query map of :1179-1185), k and v come in; the next layer's go out.
 * Weight matrices are PRE-PACKED for the kernel: pack(W [N, K]) = W.view(N / 64, 64, K / 4, 4).permute(0, 2, 1, 3), i.e.
 * [N / 64 tiles][K / 4][64 output columns][4 consecutive k]; per-head tiles are zero-padded to 64 columns:
 *   w_attn_out   pack(self_attn.out_proj.weight)            [4][64][64][4]      b_attn_out [256]
 *   w_off_logit  per head h: pack([sampling_offsets.weight[32h:32h+32]; attention_weights.weight[16h:16h+16]; 0 x 16])
 *                                                           [8][64][64][4]      b_off_logit [8][64] laid out alike
 *   w_cross_out  pack(encoder_attn.output_proj.weight)      [4][64][64][4]      b_cross_out [256]
 *   w_fc1        pack(fc1.weight [1024, 256])               [16][64][64][4]     b_fc1 [1024]
 *   w_fc2        pack(fc2.weight [256, 1024])               [4][256][64][4]     b_fc2 [256]
 *   w_qkv_next   per head h two tiles: pack([q_proj.weight[32h:32h+32]; k_proj.weight[32h:32h+32]]),
 *                pack([v_proj.weight[32h:32h+32]; 0 x 32])  [8][2][64][64][4]   b_qkv_next [8][128] = (q, k, v, 0) x 32
 * value is the BIAS-FREE value projection [B, S, 8, 32] of this layer; value_bias [256] (may be NULL) is applied by the
 * kernel times the sum of the in-range, unpadded corner weights; keep_bits (may be NULL): one bit per token, 0 = padded
 * (== the zeroed value rows of :1050-1052).  reference_points: either [B * N, 4, 2] with the valid ratios applied
 * (:1874-1880), or the plain [ref_rows, 2] points together with valid_ratios [B, 4, 2] -- the kernel multiplies.
 * q_next == NULL (last layer): no projections.  Workspace (egtr_decoder_layer_workspace): `partials` floats, `barriers`
 * 32-bit words (ZEROED ONCE when allocated, never again), `xcc_ids` ints, `status` one word zeroed by the caller: after
 * the launch bit 0 = a cluster barrier timed out, bit 1 = the workgroups of a cluster did not share one XCD -- the
 * results are void in both cases and the caller must use the per-operation entries instead.
 * num_query <= 320, else EGTR_E_UNSUPPORTED. */
typedef struct EgtrDecoderLayer {
  const float* x_in;              /* [x_rows, 256] layer input, row index taken modulo x_rows */
  const float* pos;               /* [pos_rows, 256] query position rows, row index taken modulo pos_rows */
  const float* q;                 /* [qkv_rows, 256] this layer's projections, row index modulo qkv_rows */
  const float* k;
  const float* v;
  const float* reference_points;  /* valid_ratios == NULL: [B * N, 4, 2], the per-level points; else [ref_rows, 2] */
  const float* valid_ratios;      /* [B, 4, 2] (w, h) or NULL: point of level l = reference_points[row % ref_rows] * valid_ratios[b][l] (:1865-1867) */
  const float* value;             /* [B, S, 8, 32] */
  const float* value_bias;        /* [256] or NULL */
  const unsigned* keep_bits;      /* [B, ceil(S / 32)] or NULL */
  const int64_t* spatial_shapes;  /* [4, 2] (H, W), device */
  const int64_t* level_start_index;
  float* x_out;                   /* [B * N, 256] */
  float* q_next;                  /* [B * N, 256] or NULL */
  float* k_next;
  float* v_next;
  const float* w_attn_out;
  const float* b_attn_out;
  const float* ln1_gamma;
  const float* ln1_beta;
  const float* w_off_logit;
  const float* b_off_logit;
  const float* w_cross_out;
  const float* b_cross_out;
  const float* ln2_gamma;
  const float* ln2_beta;
  const float* w_fc1;
  const float* b_fc1;
  const float* w_fc2;
  const float* b_fc2;
  const float* ln3_gamma;
  const float* ln3_beta;
  const float* w_qkv_next;
  const float* b_qkv_next;
  float* partials;
  unsigned* barriers;
  unsigned* status;
  int* xcc_ids;
  float q_scale;                  /* D^-1/2 */
  float ln_eps;
  /* x_rows / pos_rows / qkv_rows: B * N, or N when the rows are the same for every image (layer 0: query table) */
  int batch, num_query, spatial_size, x_rows, pos_rows, qkv_rows, num_clusters; /* num_clusters = batch * ceil(num_query / 8) */
  /* must be 0 (the cluster's workgroups meet at L2 barriers).  Until ABI version 3 the values 1..3 selected a barrier-free
   * tagged-data hand-over; it was slower and is gone -- the field keeps the struct layout. */
  int generation;
  int ref_rows;                   /* rows of reference_points when valid_ratios != NULL (N: the same points for every image) */
} EgtrDecoderLayer;
int egtr_decoder_layer_f32(egtr_stream_t stream, const EgtrDecoderLayer* layer);
int egtr_decoder_layer_workspace(int batch, int num_query, long long* partial_floats, int* barrier_words, int* id_words);

/* ---- MSDA prologue under autograd (training) -------------------------------------------------------------------- */
/* sampling_locations [rows, M, L, P, 2] and attention_weights [rows, M, L, P] (softmax over the L * P samples of a head)
 * from the two nn.Linear outputs sampling_offsets [rows, M * L * P * 2] / attention_logits [rows, M * L * P] (row strides
 * ld_*, floats) and reference_points [rows, L, ref_dim]: ref_dim 2 -> ref + offset / (W_l, H_l); ref_dim 4 ->
 * ref_xy + offset / P * ref_wh * 0.5 -- model/deformable_detr.py:1055-1073 in ONE pass (the reference: softmax, division,
 * multiplications, addition as separate kernels).  rows = batch * queries; spatial_shapes int64 [L, 2] (H, W) on the
 * device.  M = 8, L = P = 4 (one wavefront per row), 16-byte aligned inputs with ld % 4 == 0, else EGTR_E_UNSUPPORTED. */
int egtr_msda_geometry_forward_f32(egtr_stream_t stream, const float* sampling_offsets, long long ld_offsets,
                                   const float* attention_logits, long long ld_logits, const float* reference_points,
                                   int ref_dim, const int64_t* spatial_shapes, float* sampling_locations,
                                   float* attention_weights, long long rows, int num_heads, int num_levels,
                                   int num_points);
/* Its backward: from d loss / d sampling_locations, d loss / d attention_weights and the saved attention_weights writes
 * grad_offsets [rows, M * L * P * 2], grad_logits [rows, M * L * P] (row strides ld_grad_*: two column blocks of one
 * buffer when one nn.Linear produced both inputs) and, when grad_reference != NULL,
 * d loss / d reference_points [rows, L, ref_dim] (sum over heads and points; the decoder's reference points are a learned
 * function of the queries). */
int egtr_msda_geometry_backward_f32(egtr_stream_t stream, const float* grad_locations, const float* grad_weights,
                                    const float* attention_weights, const float* sampling_offsets, long long ld_offsets,
                                    const float* reference_points, int ref_dim, const int64_t* spatial_shapes,
                                    float* grad_offsets, long long ld_grad_offsets, float* grad_logits,
                                    long long ld_grad_logits, float* grad_reference, long long rows, int num_heads,
                                    int num_levels, int num_points);

/* ---- fused memory-bound epilogues --------------------------------------------------------------------------- */
/* y = act(x + bias[c] (+ residual)) on an NCHW fp32 activation [N, C, HW]; residual may be NULL; y may alias x.
 * (Folded frozen-BN shift + bottleneck residual + ReLU of the ResNet-50 backbone in one pass.) */
int egtr_bias_act_nchw_f32(egtr_stream_t stream, const float* x, const float* bias, const float* residual, float* y,
                           int N, int C, int HW, int relu);
/* y = LayerNorm(x (+ residual)) * gamma + beta over rows of `dim` (= 256) channels, biased variance, eps inside the
 * sqrt: the residual + LayerNorm of every encoder / decoder sub-layer (model/deformable_detr.py:1329-1330 etc.). */
int egtr_add_layernorm_f32(egtr_stream_t stream, const float* x, const float* residual, const float* gamma,
                           const float* beta, float* y, int rows, int dim, float eps);
/* Backward of egtr_add_layernorm_f32 (training; autograd of model/deformable_detr.py:1329-1330 etc.): from x, residual
 * (may be NULL), gamma and grad_y [rows, 256] writes grad_sum [rows, 256] = d loss / d (x + residual) (the gradient of both
 * inputs) and grad_gamma_beta [512] = (d gamma [256], d beta [256]).  The row statistics are recomputed (nothing is kept
 * from the forward).  workspace: egtr_add_layernorm_backward_workspace_floats(rows) floats.  Fixed summation order. */
long long egtr_add_layernorm_backward_workspace_floats(int rows);
int egtr_add_layernorm_backward_f32(egtr_stream_t stream, const float* x, const float* residual, const float* gamma,
                                    const float* grad_y, float* grad_sum, float* workspace, float* grad_gamma_beta,
                                    int rows, int dim, float eps);

/* Training form of the encoder layer's "dropout + residual + LayerNorm" (model/deformable_detr.py:1326-1330, 1341-1351 in
 * train mode) as ONE pass: y = LayerNorm(residual + keep * keep_scale * x) * gamma + beta.  keep [rows, 256] bytes (non-zero =
 * kept; NULL = no dropout), keep_scale = 1 / (1 - p).  nonfinite_flag (optional, device int, OR-ed with 1 when an output
 * element is inf / nan): the reference's "clamp the states iff any element is inf / nan" decision (dd:1346-1351) without an
 * extra pass -- follow with egtr_clamp_if_flag_f32, which returns at once while the flag is clear. */
int egtr_dropout_add_layernorm_f32(egtr_stream_t stream, const float* x, const float* residual, const unsigned char* keep,
                                   float keep_scale, const float* gamma, const float* beta, float* y, int rows, int dim,
                                   float eps, int* nonfinite_flag);
/* Its backward, one pass + a fixed-order reduction of per-workgroup partials: grad_sum [rows, 256] = d loss / d (residual +
 * dropped x) (the residual's gradient), grad_x = keep * keep_scale * grad_sum (the gradient of the Linear that produced x;
 * NULL exactly when keep is NULL: then it equals grad_sum), grad_gamma_beta_bias [768] = (d gamma, d beta, column sums of
 * grad_x = that Linear's bias gradient).  clamp_flag / y_out / clamp_value (optional): where the forward clamp was active
 * (*clamp_flag != 0 and |y_out| >= clamp_value, or NaN) the incoming gradient counts as zero (torch.clamp's backward).
 * workspace: egtr_dropout_add_layernorm_backward_workspace_floats(rows) floats. */
long long egtr_dropout_add_layernorm_backward_workspace_floats(int rows);
int egtr_dropout_add_layernorm_backward_f32(egtr_stream_t stream, const float* x, const float* residual,
                                            const unsigned char* keep, float keep_scale, const float* gamma,
                                            const float* grad_y, const int* clamp_flag, const float* y_out,
                                            float clamp_value, float* grad_sum, float* grad_x, float* workspace,
                                            float* grad_gamma_beta_bias, int rows, int dim, float eps);

/* out_t[r, :] = w_t[r, :] * scale_t[r] for n <= 64 row-major fp32 tensors in ONE launch (rows[t] x cols[t], cols % 4 == 0,
 * 16-byte aligned; out_t may alias w_t): the frozen-BN scale riding on every trainable convolution weight of the backbone in
 * training, and the same product on the weight gradients.  The pointer / size arrays are HOST arrays. */
int egtr_scale_rows_multi_f32(egtr_stream_t stream, int n, const float* const* w, const float* const* scale,
                              float* const* out, const int* rows, const int* cols);

/* bf16 storage (raw bfloat16 bits), fp32 arithmetic -- the same two epilogues for the bf16 stress configuration:
 * y = act(x + bias[c] (+ residual)) on an NCHW activation (bias fp32), and y = LayerNorm(x + residual) over 256 channels
 * (gamma / beta bf16; the residual sum is rounded to bf16 before the statistics, as the PyTorch composition does). */
int egtr_bias_act_nchw_bf16(egtr_stream_t stream, const uint16_t* x, const float* bias, const uint16_t* residual,
                            uint16_t* y, int N, int C, int HW, int relu);
int egtr_add_layernorm_bf16(egtr_stream_t stream, const uint16_t* x, const uint16_t* residual, const uint16_t* gamma,
                            const uint16_t* beta, uint16_t* y, int rows, int dim, float eps);
/* The bias / residual / ReLU epilogue on a channels-last (NHWC) bf16 activation = a [rows, C] matrix (rows = N*H*W; C % 8 == 0,
 * 16-byte aligned; y may alias x): the layout in which the bf16 backbone runs (MIOpen's NHWC convolutions, 1x1 convolutions as
 * plain GEMMs). */
int egtr_bias_act_nhwc_bf16(egtr_stream_t stream, const uint16_t* x, const float* bias, const uint16_t* residual, uint16_t* y,
                            long long rows, int C, int relu);
int egtr_bias_act_nhwc_f32(egtr_stream_t stream, const float* x, const float* bias, const float* residual, float* y,
                           long long rows, int C, int relu);   /* fp32 twin (C % 4 == 0) */
/* ... and y_plus_pos = bf16(y + pos[row % pos_rows]) from the SAME launch: the next encoder layer's `hidden + pos`
 * (deformable_detr.py:1041), rounded like the reference's bf16 add of the already rounded y.  rows % pos_rows == 0. */
int egtr_add_layernorm_pos_bf16(egtr_stream_t stream, const uint16_t* x, const uint16_t* residual, const uint16_t* gamma,
                                const uint16_t* beta, uint16_t* y, int rows, int dim, float eps, const uint16_t* pos,
                                int pos_rows, uint16_t* y_plus_pos);

/* Same, and additionally y_plus_pos = y + pos[row % pos_rows] (the "with_pos_embed" input of the next sub-layer,
 * deformable_detr.py:1023-1024 / 1148-1149), saving one elementwise launch per sub-layer. */
int egtr_add_layernorm_pos_f32(egtr_stream_t stream, const float* x, const float* residual, const float* gamma,
                               const float* beta, float* y, int rows, int dim, float eps, const float* pos,
                               int pos_rows, float* y_plus_pos);

/* In place: y[g, r, :] = keep[r] ? y[g, r, :] + bias[g, :] : 0  (bias add + padding-mask select of
 * deformable_detr.py:1050-1052 for the value projections of all decoder layers at once; keep may be NULL). */
int egtr_bias_mask_rows_f32(egtr_stream_t stream, float* y, const float* bias, const unsigned char* keep, int groups,
                            int rows, int cols);

/* ResNet stem epilogue in one pass: y = relu(maxpool(x, 3x3, stride 2, padding 1) + shift[c]), which equals
 * maxpool(relu(x + shift[c])) bit for bit; x [N, C, H, W] is the bias-free stem convolution output with the frozen BN
 * scale folded into its weights, y [N, C, (H-1)/2+1, (W-1)/2+1]. */
int egtr_bias_relu_maxpool3x3s2_f32(egtr_stream_t stream, const float* x, const float* bias, float* y, int N, int C,
                                    int H, int W);

/* Box decoding of the detection head for all decoder levels at once (model/egtr.py:286-305 with the shared bbox_embed,
 * with_box_refine = False): boxes[b, l, n, :] = sigmoid(delta[b, l, n, :] + [inverse_sigmoid(reference_l[b, n, :]), 0..]),
 * reference_0 = init_reference [B, N, ref_dim], reference_l = inter_references[:, l-1] ([B, Ld, N, ref_dim]) for l >= 1;
 * inverse_sigmoid as model/deformable_detr.py:658-662 with its eps.  ref_dim 2 or 4; delta / boxes [B, Ld, N, 4].
 * Two fusions of the inference forward: inter_references == NULL means "every level uses init_reference"
 * (no iterative box refinement: the decoder passes the same reference points to every layer), and with logits_all
 * ([batch, num_levels, num_query, num_classes], may be NULL) node_cls [batch, num_query] int64 receives
 * argmax_c logits_all[b, num_levels - 1, n, c] -- the class lookup of the relation head's frequency bias
 * (model/egtr.py:405-413), first maximum on ties and NaN as maximum like torch.argmax. */
int egtr_box_decode_argmax_f32(egtr_stream_t stream, const float* delta, const float* init_reference,
                               const float* inter_references, int batch, int num_levels, int num_query, int ref_dim,
                               float eps, float* boxes, const float* logits_all, int num_classes, int64_t* node_cls);

/* Hungarian matcher on the device: DeformableDetrHungarianMatcher.forward (model/deformable_detr.py:2925-3015) -- the
 * focal / L1 / GIoU cost matrix (:2949-2982), the adaptive-smoothing offset (:2989-2999) and
 * scipy.optimize.linear_sum_assignment per image (:3001-3005) -- in one launch, one workgroup per image, without the
 * reference's copy of the cost matrix to the host.
 *   logits [batch, num_query, num_logits], boxes [batch, num_query, 4] (cx, cy, w, h);
 *   tgt_ids int64 / tgt_boxes [sum T_b, 4]: the targets of all images concatenated, tgt_offsets int32 [batch + 1]
 *   (device) their per-image ranges; out_offsets int32 [batch + 1] (device): image b owns output entries
 *   [out_offsets[b], out_offsets[b] + min(num_query, T_b)); max_targets = max_b T_b (sizes the LDS request).
 *   cost = bbox_cost * L1 + class_cost * focal + giou_cost * (-GIoU); with smoothing != 0 additionally
 *   (cost - cost_min) + inverse_sigmoid_smoothing, both fp32 scalars evaluated by the caller as at :2992-2998.
 * Outputs: pred_idx / tgt_idx int64 and match_cost fp32 (the cost of each matched pair), sorted by query index within an
 * image, exactly the index pairs scipy returns for that cost matrix (float64 solve, same tie rule).  Optional:
 * cost_out fp32 [num_query * sum T_b] (image b's [num_query, T_b] block at num_query * tgt_offsets[b]); cost_in (same
 * layout): solve GIVEN matrices instead of evaluating the cost (logits / boxes / tgt_* may then be null); status int32
 * [batch]: 0 ok, 1 NaN / -inf in the matrix (scipy raises ValueError), 2 infeasible -- indices are -1 in those cases.
 * The float64 matrix of an image (min(N, T_b) x max(N, T_b)) is held in LDS when it fits (~150 KB: T <= 90 at N = 200);
 * otherwise in `scratch` (device, egtr_hungarian_match_scratch_doubles(...) doubles, 0 = not needed).
 * Limit: max(num_query, T_b) <= 1024. */
int egtr_hungarian_match_f32(egtr_stream_t stream, const float* logits, const float* boxes, const int64_t* tgt_ids,
                             const float* tgt_boxes, const int* tgt_offsets, const int* out_offsets, int batch,
                             int num_query, int num_logits, int max_targets, float class_cost, float bbox_cost,
                             float giou_cost, int smoothing, float cost_min, float inverse_sigmoid_smoothing,
                             int64_t* pred_idx, int64_t* tgt_idx, float* match_cost, float* cost_out,
                             const float* cost_in, int* status, double* scratch);
long long egtr_hungarian_match_scratch_doubles(int num_query, int max_targets, long long total_targets);

/* Relation + connectivity losses of SceneGraphGenerationLoss in training mode (model/egtr.py:754-923 with
 * rel_sample_negatives / rel_sample_nonmatching set and *_largest = True), value AND gradient, without the reference's
 * per-image nonzero() index lists, host-side counts and gathers:
 *   loss_out[0] = loss_rel = mean over {true relations of the matched block} U {k1 highest-logit false candidates of the
 *                 block} U {k2 highest-logit elements with an unmatched subject or object} of
 *                 BCEWithLogits(pred_rel, target * w_a w_b),  w = 1 - sigmoid(matching cost), k = min(sample * n_true, #)
 *   loss_out[1] = loss_connectivity = mean BCEWithLogits(pred_conn, any_r target != 0)  over [batch, N, N]
 *   grad_rel [batch, N, N, R], grad_conn [batch, N, N] = d loss / d logits (dense, zeros where unselected).
 * pred_idx / tgt_idx / match_cost / out_offsets: the packed outputs of egtr_hungarian_match_f32 (out_offsets int32
 * [batch + 1], device); target_rel: DEVICE array of `batch` device pointers to the images' dense [N, N, R] targets;
 * nonmatching_cost: the criterion's constant (egtr:603-608).  workspace: egtr_relation_loss_workspace_bytes() bytes.
 * Elements tied with the k-th largest logit are taken in arrival order (the reference's topk leaves that unspecified). */
int egtr_relation_loss_f32(egtr_stream_t stream, const float* pred_rel, const float* pred_conn,
                           const float* const* target_rel, const int64_t* pred_idx, const int64_t* tgt_idx,
                           const float* match_cost, const int* out_offsets, int batch, int num_query, int num_rel,
                           float nonmatching_cost, int sample_negatives, int sample_nonmatching, float* loss_out,
                           float* grad_rel, float* grad_conn, void* workspace);
long long egtr_relation_loss_workspace_bytes(int batch, int num_query);

/* IoU matrix of the reference's native evaluator routine, lib/fpn/box_intersections_cpu/bbox.pyx: mode 0 =
 * bbox_overlaps (:21-61), mode 1 = bbox_intersections (:64-108).  boxes [num_boxes, 4], query_boxes [num_query, 4]
 * (x0, y0, x1, y1), float64 like the reference (DTYPE = np.float), "+1 pixel" convention; out [num_boxes, num_query],
 * zero where the boxes do not overlap.  Bit-identical to the Cython loops (same operation order, no contraction).
 * Either count may be 0 (nothing is launched). */
int egtr_bbox_overlaps_f64(egtr_stream_t stream, const double* boxes, const double* query_boxes, int num_boxes,
                           int num_query, int mode, double* out);

/* Sine position embedding of DeformableDetrSinePositionEmbedding(normalize=True) (model/deformable_detr.py:850-876)
 * from y_embed / x_embed = cumsum of the mask along H / W ([B,H,W] fp32) and dim_t [E] (the reference's
 * temperature ** (2*(i//2)/E) table); out [B, 2E, H, W]. */
int egtr_sine_pos_embed_f32(egtr_stream_t stream, const float* y_embed, const float* x_embed, const float* dim_t,
                            float* out, int B, int H, int W, int E, float scale, float eps);

/* Everything DeformableDetrModel.forward derives from pixel_mask alone (model/deformable_detr.py:2195-2278, 1616-1648,
 * 850-876), for up to 4 feature levels level_hw = {H_0, W_0, H_1, W_1, ...} (HOST array): nearest-resized masks
 * flattened over the levels (mask_flat [B,S] bytes, 1 = valid), normalised sine position embeddings + level_embed
 * (pos_flat [B,S,2*embed_dim]; dim_t [embed_dim] = temperature^(2*(i/2)/embed_dim), level_embed [L,2*embed_dim]),
 * valid_ratios [B,L,2] and the encoder reference points [B,S,L,2]; mask_bits (optional, [B, ceil(S/32)] words,
 * fully overwritten) receives the same mask one bit per token.  pixel_mask is [B,height,width] of int64
 * (mask_elem_size 8) or bytes (1), non-zero = valid. */
int egtr_level_geometry_f32(egtr_stream_t stream, const void* pixel_mask, int mask_elem_size, const float* dim_t,
                            const float* level_embed, const int* level_hw, int num_levels, int batch, int height,
                            int width, int embed_dim, float scale, float eps, unsigned char* mask_flat,
                            float* pos_flat, float* valid_ratios, float* ref_points, unsigned* mask_bits);
/* The same for a bf16 model: pos_flat holds bfloat16 bits, bf16(bf16(sine) + level_embed) -- the two roundings of the
 * reference's `position_embedding(..).to(dtype)` and its bf16 `+ level_embed[level]` (deformable_detr.py:2224, 2259);
 * level_embed is passed widened to fp32; the other outputs are as above. */
int egtr_level_geometry_bf16(egtr_stream_t stream, const void* pixel_mask, int mask_elem_size, const float* dim_t,
                             const float* level_embed, const int* level_hw, int num_levels, int batch, int height,
                             int width, int embed_dim, float scale, float eps, unsigned char* mask_flat,
                             uint16_t* pos_flat, float* valid_ratios, float* ref_points, unsigned* mask_bits);

/* Epilogue of the per-level input projections (model/deformable_detr.py:2209-2262): conv bias + GroupNorm(num_groups)
 * + flatten(2).transpose(1, 2) + concatenation over the levels, in two launches for all levels.  x[l] is the BIAS-FREE
 * convolution output [B, 256, H_l, W_l] (NCHW); conv_bias / gamma / beta are per level [256]; level_hw = {H_0, W_0, ...};
 * the pointer arrays and level_hw are HOST arrays.  stats: device scratch of num_levels*B*num_groups*2 floats;
 * out: [B, S, 256]. */
int egtr_input_proj_groupnorm_flatten_f32(egtr_stream_t stream, int num_levels, const float* const* x,
                                          const float* const* conv_bias, const float* const* gamma,
                                          const float* const* beta, const int* level_hw, int batch, int channels,
                                          int num_groups, float eps, float* stats, float* out);
/* bf16 activations (raw bits) in and out, fp32 parameters and statistics: x[l] is the bf16 bias-free convolution output. */
int egtr_input_proj_groupnorm_flatten_bf16(egtr_stream_t stream, int num_levels, const uint16_t* const* x,
                                           const float* const* conv_bias, const float* const* gamma,
                                           const float* const* beta, const int* level_hw, int batch, int channels,
                                           int num_groups, float eps, float* stats, uint16_t* out);

/* The same epilogue for TOKEN-MAJOR bf16 projections (the channels-last backbone): x[l] is [B, level_tokens[l], 256], the
 * bias-free output of the level's 1x1 convolution run as a plain GEMM on the channels-last feature map; 32 groups of 8 channels
 * (8 consecutive channels of a token = one group).  Three small launches (partial sums per 256 tokens, their ordered reduction,
 * the normalisation).  stats: device scratch of egtr_input_proj_groupnorm_tokens_workspace_floats(...) floats, 16-byte aligned
 * (level_tokens is a HOST array); out: [B, S, 256]. */
long long egtr_input_proj_groupnorm_tokens_workspace_floats(int num_levels, const int* level_tokens, int batch);
int egtr_input_proj_groupnorm_tokens_bf16(egtr_stream_t stream, int num_levels, const uint16_t* const* x,
                                          const float* const* conv_bias, const float* const* gamma,
                                          const float* const* beta, const int* level_tokens, int batch, int channels,
                                          int num_groups, float eps, float* stats, uint16_t* out);
int egtr_input_proj_groupnorm_tokens_f32(egtr_stream_t stream, int num_levels, const float* const* x,
                                         const float* const* conv_bias, const float* const* gamma, const float* const* beta,
                                         const int* level_tokens, int batch, int channels, int num_groups, float eps,
                                         float* stats, float* out);   /* fp32 twin */

/* ---- EGTR relation head ---------------------------------------------------------------------------------- */
/* Inputs are the separable pieces of egtr.py:366-401 (see DESIGN.md "relation head algebra"):
 *   gate_q [B, N, T], gate_k [B, N, T]   : w_g[:d].q^[i,t]  and  w_g[d:].k^[j,t] + b_g      (T = Ld + 1 slots)
 *   uq [B, N, T, 2*Hd], uk [B, N, T, 2*Hd]: first-layer pre-activations W1[:, :d] q^ and W1[:, d:] k^ of the
 *                                           relation MLP (cols 0..Hd-1) and connectivity MLP (cols Hd..2Hd-1)
 *   b1 [2*Hd]; w2r [Hd, Hd], b2r [Hd], w3r [R, Hd], b3r [R]   (rel_predictor.layers.1/.2)
 *   w2c [Hd, Hd], b2c [Hd], w3c [1, Hd], b3c [1]              (connectivity_layer.layers.1/.2)
 *   rel_bias (optional) [B, N, N, R] is NOT taken: the frequency bias is gathered inside from
 *   triplet_dist [C+1, C+1, R] with node_cls [B, N] int64 (argmax of the class logits, egtr.py:405-413);
 *   pass triplet_dist = NULL to disable (use_freq_bias = False).
 * Outputs: rel_logits [B, N, N, R] (incl. frequency bias), conn_logits [B, N, N] (pre-sigmoid, what the loss
 * consumes, egtr.py:450-454); gate_mean (optional) [T] = mean over (b,i,j) of the gate (egtr.py:496-505);
 * it is accumulated with atomics, zero it first.
 * Training: h1_save / h2_save [2 (0 = relation, 1 = connectivity)][B*N*N][hidden] (each may be NULL) receive the post-ReLU
 * hidden activations of layer 1 and layer 2 of both MLPs for the backward. */
int egtr_rel_head_forward_save_f32(egtr_stream_t stream, const float* gate_q, const float* gate_k, const float* uq,
                                   const float* uk, const float* b1, const float* w2r, const float* b2r,
                                   const float* w3r, const float* b3r, const float* w2c, const float* b2c,
                                   const float* w3c, const float* b3c, const float* triplet_dist,
                                   const int64_t* node_cls, int batch, int num_query, int num_slots, int hidden,
                                   int num_rel, int num_cls_plus1, float* rel_logits, float* conn_logits,
                                   float* gate_mean, float* h1_save, float* h2_save);

/* Forward with layers 2 and 3 on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): w2r / w3r / w2c / w3c are the
 * model's bf16 parameters (raw bfloat16 bits, nn.Linear layout), the hidden activations are rounded to bf16 as matrix
 * operands, accumulation, layer 1, biases, frequency bias and outputs stay fp32.  For bf16 models (stress configuration). */
int egtr_rel_head_forward_bf16w(egtr_stream_t stream, const float* gate_q, const float* gate_k, const float* uq,
                                const float* uk, const float* b1, const uint16_t* w2r, const float* b2r,
                                const uint16_t* w3r, const float* b3r, const uint16_t* w2c, const float* b2c,
                                const uint16_t* w3c, const float* b3c, const float* triplet_dist,
                                const int64_t* node_cls, int batch, int num_query, int num_slots, int hidden,
                                int num_rel, int num_cls_plus1, float* rel_logits, float* conn_logits,
                                float* gate_mean);

/* The same forward with ALL three layers on the bf16 matrix cores (csrc/rel_head_bf16.hip): the gated sum over the slots
 * (egtr.py:386-401) is a matrix product too -- per 32 pairs (4 subjects x 8 objects) twelve K = 16 steps, one per
 * per-query row set, with the gates as the other operand -- W2 of a workgroup's MLP stays in LDS, and the relation tile is
 * stored straight from the accumulators.  uq_packed / uk_packed: the per-query tables rounded to bf16 and packed in operand
 * order by egtr_rel_head_pack_tables_bf16 -- (batch * num_query) rows of [mlp 2][channel tile 8][half 2][channel 32][8 slots]
 * bfloat16 (16 KiB per row; slot = 8 half + e, zero for slots >= num_slots).  num_slots <= 10, hidden == 256, num_rel <= 64. */
int egtr_rel_head_pack_tables_bf16(egtr_stream_t stream, const void* u /* [rows, num_slots, 512] fp32, or bf16 bits */,
                                   int u_is_bf16, int rows, int num_slots, uint16_t* packed);
int egtr_rel_head_forward_bf16p(egtr_stream_t stream, const float* gate_q, const float* gate_k,
                                const uint16_t* uq_packed, const uint16_t* uk_packed, const float* b1,
                                const uint16_t* w2r, const float* b2r, const uint16_t* w3r, const float* b3r,
                                const uint16_t* w2c, const float* b2c, const uint16_t* w3c, const float* b3c,
                                const float* triplet_dist, const int64_t* node_cls, int batch, int num_query,
                                int num_slots, int hidden, int num_rel, int num_cls_plus1, float* rel_logits,
                                float* conn_logits, float* gate_mean);

/* Object-query-sized nn.Linear of a bf16 model (csrc/linear_bf16.hip): y[M, N] (row stride ldy) = act(alpha (x[M, K] (row
 * stride ldx) . W[N, K]^T + bias)), raw bfloat16 bits in and out, fp32 accumulation on v_mfma_f32_32x32x16_bf16, the result
 * rounded to bf16 once (after bias / scale / ReLU; alpha = the attention scaling of q_proj, deformable_detr.py:1166).  bias
 * may be NULL.  K % 16 == 0, N % 32 == 0, ldx % 8 == 0, x / weight 16-byte aligned. */
int egtr_linear_bf16(egtr_stream_t stream, const uint16_t* x, int ldx, const uint16_t* weight, const uint16_t* bias,
                     uint16_t* y, int ldy, int M, int N, int K, int relu, float alpha);

/* Token-sized nn.Linear, y[M, N] (row stride ldy) = act(x[M, K] (row stride ldx) . W[N, K]^T + bias): fp32 in, fp32 out,
 * fp32-level accuracy, evaluated on the bf16 matrix cores from exact three-way bf16 splits of both operands (six cross
 * terms, fp32 accumulation; csrc/gemm_split.hip).  Replaces the vendor fp32 GEMM behind the reference's encoder
 * nn.Linear layers (model/deformable_detr.py:1049, 1053-1058, 1102, 1337-1343) in inference.  w_tiled: W pre-split and
 * pre-tiled, raw bfloat16 bits, [N/128][K/32][3 pieces hi, mid, lo][128 rows][32]
 * (egtr_amd/ops.py::gemm_split_weights).  K % 32 == 0, N % 128 == 0, x 16-byte aligned with ldx % 4 == 0, else
 * EGTR_E_UNSUPPORTED.  bias may be NULL; relu != 0 applies max(., 0). */
int egtr_linear_split_bf16_f32(egtr_stream_t stream, const float* x, int ldx, const uint16_t* w_tiled,
                               const float* bias, float* y, int ldy, int M, int K, int N, int relu);

/* The w_tiled stream of the two entries around this one from W [N, K] fp32 (row stride ldw), or, with transposed != 0,
 * from the [K, N] array w holding W^T (element (n, k) at w[k * ldw + n]: the data-gradient product g . W of the training
 * backward is "linear" with weight W^T).  Pieces rounded to nearest even, bit-identical to ops.gemm_split_weights.
 * One launch -- the training step re-tiles every weight after each optimizer step. */
int egtr_gemm_split_tile_weights_f32(egtr_stream_t stream, const float* w, int ldw, int transposed, int N, int K,
                                     uint16_t* w_tiled);
/* Both streams of W [N, K] in one launch: w_tiled_pair = [tiling of W (3 N K) | tiling of W^T (3 N K)] (N, K % 128 == 0). */
int egtr_gemm_split_tile_weights_pair_f32(egtr_stream_t stream, const float* w, int ldw, int N, int K,
                                          uint16_t* w_tiled_pair);
/* The pair tilings of up to 8 weights in ONE launch (a training step re-tiles every encoder weight after each optimizer
 * step).  Weight i is [N[i], K[i]] (both multiples of 128): its first split_rows[i] rows are read from w[i] (row stride
 * ldw[i]) and the remaining rows from w2[i] (row stride ldw2[i]) -- two nn.Linear weights applied to the same input, e.g.
 * sampling_offsets | attention_weights (model/deformable_detr.py:1053-1058), tiled as ONE weight without a materialised
 * concatenation; w2 == NULL or w2[i] == NULL: a single source.  w_tiled_pair[i]: 6 N K bf16, as the pair entry writes them.
 * All arrays are HOST arrays. */
int egtr_gemm_split_tile_weights_multi_f32(egtr_stream_t stream, int num_weights, const float* const* w, const int* ldw,
                                           const float* const* w2, const int* ldw2, const int* split_rows, const int* N,
                                           const int* K, uint16_t* const* w_tiled_pair);

/* Weight gradient of such a layer in training: grad_weight [N, K] (contiguous) = g[M, N]^T . x[M, K] (row strides ldg,
 * ldx), same split arithmetic; the M rows are cut into chunks whose 128 x 128 partial products go through `workspace`
 * (egtr_linear_split_bf16_wgrad_workspace_floats(M, N, K) floats) and are summed in a fixed order (bit-reproducible).
 * Replaces the vendor GEMM autograd calls for model/deformable_detr.py's encoder nn.Linear layers (K = rows of the token
 * sequence: 66 TFLOP/s in the vendor library for N = K = 256).  N, K % 128 == 0, else EGTR_E_UNSUPPORTED. */
long long egtr_linear_split_bf16_wgrad_workspace_floats(int M, int N, int K);
int egtr_linear_split_bf16_wgrad_f32(egtr_stream_t stream, const float* g, int ldg, const float* x, int ldx,
                                     float* grad_weight, float* workspace, int M, int N, int K);

/* Up to 8 such products with the same M and K in ONE launch (their tiles share the grid): the value projection and the
 * offsets / attention-weights projection of an encoder layer, the six value projections of the decoder.  All arrays are
 * HOST arrays of num_problems entries (pointers inside are device pointers; bias[i] may be NULL).
 * A problem may add a [pos_rows[g], K] table to the rows of x_g on the way in (row % pos_rows): `hidden_states + position_embeddings` (model/deformable_detr.py:1041) as the input of the sampling-offset /
 * attention-weight projection without a materialised sum.  pos == NULL or pos[g] == NULL: none. */
int egtr_linear_split_bf16_grouped_pos_f32(egtr_stream_t stream, int num_problems, const float* const* x, const int* ldx,
                                           const uint16_t* const* w_tiled, const float* const* bias, float* const* y,
                                           const int* ldy, const int* N, const int* relu, int M, int K,
                                           const float* const* pos, const int* pos_rows);

/* The same with the epilogue options of the TRAINING step (every array may be NULL = none, every entry NULL = none), applied
 * in this order to v = act(x_g W_g^T + bias_g):
 *   row_keep[g] [M] bytes      rows with 0 yield zeros (value rows of padded tokens, model/deformable_detr.py:1052; in the
 *                              backward: their gradients);
 *   relu_ref[g] [M, ldref[g]]  v = relu_ref > 0 ? v : 0 -- the ReLU backward of the layer whose output relu_ref is, fused into
 *                              this data-gradient product (replaces threshold_backward over the [M, 1024] FFN activation);
 *   add1[g], add2[g] [M, ldadd[g]]  v += add1 + add2: the gradients of the other branches that meet at the same tensor
 *                              (autograd's AccumulateGrad adds); either may alias y[g];
 *   colpart[g] [ceil(M / 32), N[g]]  column sums of the stored values over each block of 32 rows -- the bias gradient of the
 *                              layer the gradient flows into, finished in a fixed order by egtr_column_sum_f32 over the blocks. */
int egtr_linear_split_bf16_ex_f32(egtr_stream_t stream, int num_problems, const float* const* x, const int* ldx,
                                  const uint16_t* const* w_tiled, const float* const* bias, float* const* y,
                                  const int* ldy, const int* N, const int* relu, int M, int K, const float* const* pos,
                                  const int* pos_rows, const unsigned char* const* row_keep, const float* const* relu_ref,
                                  const int* ldref, const float* const* add1, const float* const* add2, const int* ldadd,
                                  float* const* colpart);
/* egtr_linear_split_bf16_wgrad_f32 for a layer whose input was `x + pos` formed on load (x_pos [pos_rows, K], row m uses
 * x_pos[m % pos_rows]; NULL = none) and / or whose gradient rows are masked (row_keep [M] bytes, 0 = the row counts as zero;
 * NULL = none). */
int egtr_linear_split_bf16_wgrad_ex_f32(egtr_stream_t stream, const float* g, int ldg, const float* x, int ldx,
                                        float* grad_weight, float* workspace, int M, int N, int K, const float* x_pos,
                                        int pos_rows, const unsigned char* row_keep);

/* ---- the "XS" operand format of the row-panel kernels (csrc/xs_format.h, csrc/xs_split.hip) --------------------------
 * XS(X) of a logical fp32 matrix X[rows][K] (K % 16 == 0): the exact three-way bf16 split x = hi + mid + lo, stored as
 * 1 KiB fragments of 32 rows x 16 k of ONE piece; fragment (rb = row / 32, ks = k / 16, piece p) at byte
 * ((rb * (K / 16) + ks) * 3 + p) * 1024, element (r = row % 32, kk = k % 16) inside it at (kk / 8) * 512 + r * 16 +
 * (kk % 8) * 2 (csrc/xs_format.h).  A fragment is what one global_load_lds_dwordx4 wave-instruction moves and what one
 * ds_read_b128 wave-instruction hands to v_mfma_f32_32x32x16_bf16.  egtr_xs_bytes: size of XS(X) in bytes (rows rounded
 * up to 32; 0 for K % 16 != 0). */
long long egtr_xs_bytes(int rows, int K);
/* fp32 row-major x [rows, K] (row stride ldx) -> XS(x) in xs_out and / or XS(x + pos) in xs_pos_out (pos [pos_rows, K]
 * contiguous, row r uses pos[r % pos_rows]; either output may be NULL).  round_to_nearest != 0: pieces rounded to nearest
 * even (weights, prepared once); 0: truncation split (activations).  Non-finite elements: hi carries the inf / quiet NaN,
 * mid = lo = 0. */
int egtr_xs_split_f32(egtr_stream_t stream, const float* x, int ldx, const float* pos, int pos_rows, int rows, int K,
                      void* xs_out, void* xs_pos_out, int round_to_nearest);

/* The tail of a ResNet bottleneck on channels-last fp32 data as one launch (inference, frozen batch norm folded):
 *     y = act_out( act_in(a + a_shift) . W^T + bias + shortcut )
 * a [M, lda] = the raw output of the 3x3 convolution (M = B H W pixel rows, K = planes), a_shift [K] its folded batch-norm
 * shift (NULL: none), relu_in != 0: ReLU behind it; W [N, K] the 1x1 convolution as XS(W) (egtr_xs_split_f32 with
 * round_to_nearest = 1); bias [N] (NULL: none); shortcut [M, ld_shortcut] (NULL: none); relu_out != 0: closing ReLU.
 * Replaces, per block: an in-place shift + ReLU pass, a vendor fp32 GEMM and a shift + shortcut + ReLU pass
 * (egtr_amd/backbone.py::Bottleneck.forward_folded_nhwc; reference: model/deformable_detr.py:735-760, the timm ResNet-50
 * backbone with frozen batch norm -- conv3 -> bn3 -> += shortcut -> relu).  Six-term split-bf16 arithmetic: the error of an
 * fp32 GEMM.  K in {64, 128, 256, 512}, N % 64 == 0, 16-byte aligned pointers, row strides % 4 == 0 (EGTR_E_UNSUPPORTED
 * otherwise).  tile_rows (0 | 32 | 64) / tile_cols (0 | 128 | 256): 0 = the library's choice; other values pin the workgroup
 * tile (tools/conv3_fused_ab.py sweeps them). */
int egtr_conv1x1_tail_x6_f32(egtr_stream_t stream, const float* a, int lda, const float* a_shift, int relu_in,
                             const void* w_xs, const float* bias, const float* shortcut, int ld_shortcut, int relu_out,
                             float* y, int ldy, int M, int K, int N, int tile_rows, int tile_cols);

/* The ResNet stem in inference as one launch: 7x7 convolution (stride 2, padding 3, 3 -> 64 channels) + folded batch-norm shift
 * + ReLU + 3x3 max-pool (stride 2, padding 1); x [B, 3, H, W] NCHW fp32 -> y [B, Hp, Wp, 64] channels-last fp32 with
 * Hc = (H - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1 (model/deformable_detr.py:735-760: timm ResNet-50 conv1 -> bn1 -> act1 -> maxpool).
 * Six-term split-bf16 products (error of an fp32 convolution).  w_xs = XS(Wm [64, 224]) (egtr_xs_split_f32, round_to_nearest = 1)
 * with Wm[n][ky * 32 + kx * 4 + c] = W[n][c][ky][kx] and zeros at kx == 7 / c == 3 (egtr_amd.ops.stem_weights). */
int egtr_stem_conv7x7_pool_x6_f32(egtr_stream_t stream, const float* x, const void* w_xs, const float* bias, float* y, int B,
                                  int H, int W);

/* The bf16 twin of egtr_stem_conv7x7_pool_x6_f32 (the bf16 model of the stress configuration): x [B, 3, H, W] NCHW bf16 ->
 * y [B, Hp, Wp, 64] channels-last bf16, fp32 accumulation, shift fp32; rounding points of the composition it replaces (bf16
 * convolution output, shift + ReLU in fp32, one more rounding; the pool commutes).  w_packed = egtr_conv1x1_tail_pack_weights_bf16
 * of Wm [64, 224] with Wm[n][ky * 32 + kx * 4 + c] = W[n][c][ky][kx], zeros at kx == 7 / c == 3 (egtr_amd.ops.stem_weights_bf16). */
int egtr_stem_conv7x7_pool_bf16(egtr_stream_t stream, const uint16_t* x, const uint16_t* w_packed, const float* bias, uint16_t* y,
                                int B, int H, int W);

/* 3x3 convolution, stride 1 or 2, padding 1, no bias, channels-last fp32 (x [B, H, W, C] -> y [B, Ho, Wo, N], Ho = (H - 1) /
 * stride + 1) with the six-term split-bf16 arithmetic (error of an fp32 convolution): the middle convolution of a ResNet
 * bottleneck in inference (model/deformable_detr.py:735-760, timm ResNet-50; the folded batch norm's shift + ReLU is applied by
 * the consumer, egtr_conv1x1_tail_x6_f32).  w_xs = XS(Wm [N, 9 C]) (egtr_xs_split_f32, round_to_nearest = 1) with the
 * channels in phases of CP = egtr_conv3x3_phase_channels(C, N, stride, variant):
 *     Wm[n][((ph * 3 + dy) * 3 + dx) * CP + c'] = W[n][ph * CP + c'][dy][dx]        (CP == C: Wm[n][(dy * 3 + dx) * C + c])
 * Served: C == N in {64, 128, 256, 512} at stride 1, {128, 256, 512} at stride 2 (phase_channels returns 0 and the
 * convolution EGTR_E_UNSUPPORTED otherwise).  variant: 0 = the library's tile, other values pin a tile (tools/conv3x3_ab.py). */
int egtr_conv3x3_phase_channels(int C, int N, int stride, int variant);
int egtr_conv3x3_x6_f32(egtr_stream_t stream, const float* x, const void* w_xs, float* y, int B, int H, int W, int C, int N,
                        int stride, int variant);

/* 1x1 convolution with stride 1 or 2 (no padding, no bias) on channels-last fp32 data: x [B, H, W, C] -> y [B, Ho, Wo, N],
 * Ho = (H - 1) / stride + 1 -- the shortcut projection of a bottleneck that changes resolution (model/deformable_detr.py:735-760:
 * timm ResNet-50 `downsample.0`; its folded batch-norm shift rides on the tail's shift3).  The machinery of egtr_conv3x3_x6_f32
 * with one tap: the 4 x 8 input pixels a tile reads are gathered with the stride, channels in phases of 256; w_xs = XS(W [N, C])
 * (egtr_xs_split_f32, round_to_nearest = 1).  C in {256, 512, 1024}, N % 128 == 0 (EGTR_E_UNSUPPORTED otherwise). */
int egtr_conv1x1_strided_x6_f32(egtr_stream_t stream, const float* x, const void* w_xs, float* y, int B, int H, int W, int C, int N,
                                int stride);

/* The bf16 twin (the bf16 model of the stress configuration): a, shortcut, y bf16 (raw bits), shifts fp32, fp32 accumulation;
 *     y = act_out( bf16( act_in(a + a_shift) . W^T ) + bias + shortcut )
 * with the rounding points of the composition it replaces (the shifted + rectified input rounded to bf16, the product rounded
 * to bf16, the sum rounded once more): egtr_bias_act_nhwc_bf16, a vendor bf16 GEMM, egtr_bias_act_nhwc_bf16.  w_packed: W [N, K]
 * in MFMA operand order from egtr_conv1x1_tail_pack_weights_bf16 (N * K elements; a derived constant of the weights).
 * K in {64, 128, 256, 512}, N % 256 == 0, 16-byte aligned pointers, row strides % 8 == 0 (EGTR_E_UNSUPPORTED otherwise). */
int egtr_conv1x1_tail_pack_weights_bf16(egtr_stream_t stream, const uint16_t* w, int ldw, int N, int K, uint16_t* w_packed);
int egtr_conv1x1_tail_bf16(egtr_stream_t stream, const uint16_t* a, int lda, const float* a_shift, int relu_in,
                           const uint16_t* w_packed, const float* bias, const uint16_t* shortcut, int ld_shortcut, int relu_out,
                           uint16_t* y, int ldy, int M, int K, int N);

/* The encoder layer's feed-forward block in ONE launch (csrc/ffn_x6.hip; reference: two nn.Linear + ReLU + dropout(eval) +
 * residual + LayerNorm, model/deformable_detr.py:1335-1345): out = fc2(relu(fc1(x))), or with ln_gamma / ln_beta
 * out = LayerNorm(x + fc2(relu(fc1(x)))) and optionally out_pos = out + pos[row % pos_rows].  x [M, ldx] fp32;
 * w1_xs = XS(W1 [ffn_dim, d_model]), w2_xs = XS(W2 [d_model, ffn_dim]) (egtr_xs_split_f32, round_to_nearest = 1); fp32
 * biases; fp32-level accuracy (six bf16 cross terms per product, fp32 accumulation).  The [M, ffn_dim] hidden activation
 * never leaves the compute units.  d_model == 256 and ffn_dim % 64 == 0, else EGTR_E_UNSUPPORTED.  Inference only. */
int egtr_ffn_x6_f32(egtr_stream_t stream, const float* x, int ldx, const void* w1_xs, const float* b1, const void* w2_xs,
                    const float* b2, const float* ln_gamma, const float* ln_beta, float eps, const float* pos,
                    int pos_rows, float* out, float* out_pos, int M, int d_model, int ffn_dim);

/* The feed-forward block of a bf16 model in one launch (csrc/ffn_bf16.hip): y = LayerNorm(x + fc2(relu(fc1(x)))) and
 * optionally y_pos = y + pos[row % pos_rows], raw bfloat16 bits everywhere, fp32 accumulation on v_mfma_f32_32x32x16_bf16,
 * intermediate roundings where the reference's bf16 tensors have them (fc1 output, fc2 output, residual sum, y before + pos).
 * The [M, ffn_dim] activation never leaves the registers.  w_packed: both weight matrices in operand order, written once per
 * weight version by egtr_ffn_pack_weights_bf16 from the nn.Linear layouts w1 [ffn_dim, 256], w2 [256, ffn_dim]
 * (egtr_ffn_packed_weights_bytes(ffn_dim) bytes = 2 * 256 * ffn_dim * 2).  d_model == 256, ffn_dim % 32 == 0 and <= 1024,
 * 16-byte aligned tensors; pos and y_pos both NULL or both set.  Inference only. */
long long egtr_ffn_packed_weights_bytes(int ffn_dim);
int egtr_ffn_pack_weights_bf16(egtr_stream_t stream, const uint16_t* w1, const uint16_t* w2, int d_model, int ffn_dim,
                               uint16_t* packed);
int egtr_ffn_layernorm_bf16(egtr_stream_t stream, const uint16_t* x, const uint16_t* w_packed, const uint16_t* b1,
                            const uint16_t* b2, const uint16_t* gamma, const uint16_t* beta, float eps, const uint16_t* pos,
                            int pos_rows, uint16_t* y, uint16_t* y_pos, int M, int d_model, int ffn_dim);

/* The whole TAIL of an encoder layer in one launch (csrc/ffn_x6.hip, ffn_x6_kernel<true>):
 *   y1  = LayerNorm1(hidden + context . Wp^T + bp)                 (output_proj + self_attn_layer_norm, dd:1102, 1326-1330)
 *   out = LayerNorm2(y1 + fc2(relu(fc1(y1))))   [out_pos = out + pos[row % pos_rows]]            (dd:1335-1345)
 * context = the multi-scale deformable attention's output before its output projection [M, ldc], hidden = the layer's input
 * [M, ldh]; wp_xs / w1_xs / w2_xs in the XS format (egtr_xs_split_f32, round_to_nearest = 1).  y1 never goes to memory.
 * d_model == 256, ffn_dim % 64 == 0, 16-byte aligned operands (EGTR_E_UNSUPPORTED otherwise).  Inference only. */
int egtr_encoder_tail_x6_f32(egtr_stream_t stream, const float* context, int ldc, const float* hidden, int ldh,
                             const void* wp_xs, const float* bp, const float* ln1_gamma, const float* ln1_beta, float eps1,
                             const void* w1_xs, const float* b1, const void* w2_xs, const float* b2, const float* ln2_gamma,
                             const float* ln2_beta, float eps2, const float* pos, int pos_rows, float* out, float* out_pos,
                             int M, int d_model, int ffn_dim);

/* y = x . W^T + bias for a 256 -> 256 linear layer, or with ln_gamma / ln_beta y = LayerNorm(residual + x . W^T + bias) and
 * optionally out_pos = y + pos[row % pos_rows], in ONE launch (csrc/ffn_x6.hip, proj_x6_kernel): the attention block's
 * output projection with its residual add and LayerNorm (model/deformable_detr.py:1102, 1326-1330).  w_xs = XS(W [256, 256])
 * (egtr_xs_split_f32, round_to_nearest = 1); x [M, ldx], residual [M, ldr] fp32; fp32-level accuracy.  d_model == 256,
 * else EGTR_E_UNSUPPORTED.  Inference only. */
int egtr_proj_ln_x6_f32(egtr_stream_t stream, const float* x, int ldx, const void* w_xs, const float* bias,
                        const float* residual, int ldr, const float* ln_gamma, const float* ln_beta, float eps,
                        const float* pos, int pos_rows, float* out, float* out_pos, int M, int d_model);

/* out[w] = x . W_w^T + bias_w for num_weights stacked 256 -> 256 linear layers applied to the SAME rows x [M, ldx], one
 * launch: the input panel of a workgroup is split once and re-used by every weight (the decoder's six cross-attention value
 * projections of the encoder output, model/deformable_detr.py:1048-1049).  w_xs = XS of the stacked weights
 * [num_weights * 256, 256]; bias [num_weights * 256] or NULL; out [num_weights, M, 256] contiguous. */
int egtr_proj_multi_x6_f32(egtr_stream_t stream, const float* x, int ldx, const void* w_xs, const float* bias, float* out,
                           int M, int d_model, int num_weights);

/* fp32 forward (fp32 operands in, fp32 out, fp32-level accuracy) with layers 2 and 3 on the bf16 matrix cores from
 * THREE-way bf16 splits of both operands, x = hi + mid + lo, keeping the six leading cross terms (the dropped ones are
 * <= 2^-24 of the product) and accumulating in fp32: on gfx950 the fp32 matrix rate equals the fp32 vector rate, the
 * bf16 matrix rate is 16x that, so this is 2.67x less matrix time than egtr_rel_head_forward_save_f32 for the same result to
 * fp32 rounding.  Inference only.  The weights arrive pre-split and in operand order (raw bfloat16 bits):
 *   w2x_*  [8 nt][16 t][3 piece][64 lane][8]:  piece(W2)[32 nt + (lane & 31)][16 t + 8 (lane >> 5) + e]
 *   w3x_rel [8 nt][2 kb][OT][3 piece][64 lane][8], OT = 1 if num_rel <= 32 else 2:
 *           piece(W3)[32 ot + (lane & 31)][32 nt + 16 kb + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)], rows >= num_rel zero
 * (egtr_amd/ops.py::rel_head_split_weights builds them); w3c / all biases / tables are fp32 as in the f32 entry.
 * num_slots <= 9 (EGTR_E_UNSUPPORTED above: the f32 entry serves 10).  apply_sigmoid != 0: the outputs are
 * sigmoid(logit), i.e. the model's pred_rel / pred_connectivity (model/egtr.py:450-454), not the logits. */
int egtr_rel_head_forward_bf16x6_f32(egtr_stream_t stream, const float* gate_q, const float* gate_k, const float* uq,
                                     const float* uk, const float* b1, const uint16_t* w2x_rel, const float* b2r,
                                     const uint16_t* w3x_rel, const float* b3r, const uint16_t* w2x_conn,
                                     const float* b2c, const float* w3c, const float* b3c, const float* triplet_dist,
                                     const int64_t* node_cls, int batch, int num_query, int num_slots, int hidden,
                                     int num_rel, int num_cls_plus1, float* rel_logits, float* conn_logits,
                                     float* gate_mean, int apply_sigmoid);

/* The same kernel as the TRAINING forward: logits (never the sigmoid) and, for the backward, the post-ReLU activations of both
 * layers h1_save / h2_save [2 (mlp)][B*N*N][hidden] exactly as egtr_rel_head_forward_save_f32 stores them.  num_slots 4 or 7
 * (three / six decoder layers), else EGTR_E_UNSUPPORTED: the caller stays on egtr_rel_head_forward_save_f32. */
int egtr_rel_head_forward_bf16x6_save_f32(egtr_stream_t stream, const float* gate_q, const float* gate_k, const float* uq,
                                          const float* uk, const float* b1, const uint16_t* w2x_rel, const float* b2r,
                                          const uint16_t* w3x_rel, const float* b3r, const uint16_t* w2x_conn,
                                          const float* b2c, const float* w3c, const float* b3c,
                                          const float* triplet_dist, const int64_t* node_cls, int batch, int num_query,
                                          int num_slots, int hidden, int num_rel, int num_cls_plus1, float* rel_logits,
                                          float* conn_logits, float* gate_mean, float* h1_save, float* h2_save);

/* The three operand streams above from the fp32 weights W2_rel / W2_conn [hidden, hidden] and W3_rel [num_rel, hidden] in one
 * launch (pieces rounded to nearest): what a training step rebuilds after every optimizer step.  hidden = 256, num_rel <= 64. */
int egtr_rel_head_streams_f32(egtr_stream_t stream, const float* w2_rel, const float* w2_conn, const float* w3_rel,
                              int hidden, int num_rel, uint16_t* w2x_rel, uint16_t* w2x_conn, uint16_t* w3x_rel);


/* Pairwise part of the relation-head backward (everything that is not a plain GEMM).  dh1 [2][B*N*N][hidden] is the
 * gradient wrt the pre-ReLU layer-1 output (relation half, connectivity half), produced by rocBLAS GEMMs from the
 * saved activations.  Outputs (fully overwritten): grad_uq / grad_uk [B,N,T,2*hidden], grad_gate_q / grad_gate_k
 * [B,N,T].  dz_workspace: B*N*N*T floats of scratch (receives d loss / d gate logit per pair and slot). */
int egtr_rel_head_backward_pairs_f32(egtr_stream_t stream, const float* dh1, const float* gate_q, const float* gate_k,
                                     const float* uq, const float* uk, int batch, int num_query, int num_slots,
                                     int hidden, float* grad_uq, float* grad_uk, float* grad_gate_q,
                                     float* grad_gate_k, float* dz_workspace);

/* Detection losses of ONE output set in one launch (model/egtr.py:611-659 loss_labels, 661-670 loss_cardinality,
 * 692-712 loss_boxes): sigmoid focal loss (gamma = 2) over logits [B, N, C] against the matched target classes, the
 * argmax-based cardinality count, L1 and generalised-IoU loss of the matched boxes -- values AND gradients.
 * pred_idx / tgt_idx: the matcher's packed indices (egtr_hungarian_match_f32), entries match_offsets[b] ..
 * match_offsets[b + 1] belong to image b, tgt_idx is relative to the image's own targets; target_labels / target_boxes:
 * all targets concatenated, image b's at target_offsets[b] ...  out [B, 4] = per image {focal, L1, GIoU} sums already
 * divided by num_boxes, and the cardinality count; grad_logits [B, N, C], grad_boxes_l1 / grad_boxes_giou [B, N, 4]:
 * gradients of the three (summed) losses w.r.t. logits / pred_boxes.  num_query <= 2048. */
int egtr_detection_loss_f32(egtr_stream_t stream, const float* logits, const float* pred_boxes, const int64_t* pred_idx,
                            const int64_t* tgt_idx, const int* match_offsets, const int64_t* target_labels,
                            const float* target_boxes, const int* target_offsets, int batch, int num_query,
                            int num_classes, float focal_alpha, float num_boxes, float* grad_logits,
                            float* grad_boxes_l1, float* grad_boxes_giou, float* out);

/* "Clamp the encoder states iff any element is inf / nan" (model/deformable_detr.py:1346-1351) without the reference's
 * host branch: egtr_any_nonfinite_f32 raises *flag (int, zero before the call) if x [n] holds a non-finite value;
 * egtr_clamp_if_flag_f32 then clamps t [n] to +-clamp_value in place (mask_gradient = 0; NaN stays NaN), or zeroes the
 * gradient t [n] where |x| >= clamp_value or x is NaN (mask_gradient = 1, x = the clamped states) -- both return at once
 * while *flag == 0. */
int egtr_any_nonfinite_f32(egtr_stream_t stream, const float* x, long long n, int* flag);
int egtr_clamp_if_flag_f32(egtr_stream_t stream, float* t, const float* x, long long n, const int* flag,
                           float clamp_value, int mask_gradient);

/* Bias gradient of a token-sized nn.Linear in training: out [N] = column sums of g [M, N].  With relu_output (the layer's
 * post-ReLU output y [M, N]) the ReLU backward is applied on the way: g_masked = g * [y > 0] is written and summed.
 * g_masked may alias g.  workspace: egtr_column_sum_workspace_floats(M, N) floats.  Fixed summation order
 * (bit-reproducible). */
long long egtr_column_sum_workspace_floats(int M, int N);
int egtr_column_sum_f32(egtr_stream_t stream, const float* g, const float* relu_output, float* g_masked,
                        float* workspace, float* out, int M, int N);
/* out [N] = sum_r row_weight[r] * g[r][:]  ([1, M] x [M, N]: the weight gradient of a one-output linear layer -- the
 * relation head's connectivity output, model/egtr.py:414-416 under autograd); workspace as above; N % 4 == 0. */
int egtr_weighted_column_sum_f32(egtr_stream_t stream, const float* g, const float* row_weight, float* workspace,
                                 float* out, int M, int N);

/* pad_and_create_pixel_mask on the device (the feature extractor's batching step; reference preprocessing around
 * model/deformable_detr.py:270-385): `images` is a DEVICE array of `batch` device pointers to fp32 [channels, h_b, w_b]
 * images, heights_widths a DEVICE array [batch][2] = (h_b, w_b); writes the zero-padded, top-left aligned batch
 * pixel_values [batch, channels, H, W] and pixel_mask [batch, H, W] int64 (1 = real pixel) in one launch. */
int egtr_pad_batch_f32(egtr_stream_t stream, const float* const* images, const int* heights_widths, int batch,
                       int channels, int H, int W, float* pixel_values, int64_t* pixel_mask);

#ifdef __cplusplus
}
#endif
#endif /* EGTR_HIP_H */
