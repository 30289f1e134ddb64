/* libiggt_hip.so -- C ABI of the MI355X (gfx950) kernels behind the IGGT forward path.
 *
 * Boundary contract: the reference's drop-in surface is its Python
 * nn.Module API (iggt.models.vggt.IGGT etc.); the reference has no FFI of its own.  This header is
 * the layer *below* that API: plain device pointers, sizes and a hipStream_t (as void*), no torch
 * types.  The Python host (iggt_official_amd/_C.py, ctypes) binds exactly these symbols; each
 * entry cites the reference operation (file:line under /root/reference) it replaces.
 *
 * Conventions: all pointers are device pointers unless stated; `ld*`/strides are in ELEMENTS;
 * 16-bit tensors are passed as void*; every function returns 0 on success, a negative value for an
 * argument-contract violation, or a positive hipError_t.  Functions only enqueue work on
 * `stream` (no host sync, no allocation) and are hipGraph-capturable.
 *
 * 16-bit operand format of the trunk kernels: the `_bf16` entry points take/produce bfloat16 (the
 * reference's autocast GPU mode, demo.py:190-193), their `_f16` twins IEEE half -- same kernels, same MFMA
 * rate (v_mfma_f32_32x32x16_{bf16,f16}), 11 instead of 8 significant bits per operand, stores saturating at
 * +-65504.  fp16 is what the host model uses by default: it is what brings the outputs within 1e-3 of the
 * fp32 CPU reference (oracle/precision_sim.py).
 */
#ifndef IGGT_HIP_H
#define IGGT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* Library/ABI version: bumped on any signature change. */
int iggt_hip_abi_version(void);

/* C = A[M,K] . W[N,K]^T with fp32 accumulate on bf16 MFMA, fused epilogue:
 *   val = act(acc + bias[n]) * gamma[n] (+ add_table[m % rows_in][n]);
 *   out[row(m)][n] (= | +=) val,  row(m) = rows_in ? (m / rows_in) * rows_out + row_off + m % rows_in : m
 * act: 0 none, 1 exact (erf) GELU, 2 ReLU.  out is fp32 (out_is_f32) or bf16.  K % 64 == 0.
 * Replaces nn.Linear qkv/proj (iggt/layers/attention.py:40,45,52,75), Mlp fc1+GELU/fc2
 * (iggt/layers/mlp.py:34-39), LayerScale + residual (iggt/layers/layer_scale.py:26,
 * iggt/layers/block.py:105-106) and the patch-embed conv + pos-embed add
 * (iggt/layers/patch_embed.py:75-77, iggt/layers/vision_transformer.py:223). */
int iggt_gemm_bf16(const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                   const float* bias, const float* gamma, const float* add_table,
                   void* out, long ldo, int out_is_f32, int accumulate, int act,
                   int rows_in, int rows_out, int row_off, void* stream);
int iggt_gemm_f16(const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                  const float* bias, const float* gamma, const float* add_table,
                  void* out, long ldo, int out_is_f32, int accumulate, int act,
                  int rows_in, int rows_out, int row_off, void* stream);

/* (ABI v25's iggt_gemm_*_ws / iggt_gemm_ws_bytes -- a two-slice split-K behind a caller-owned workspace -- measured slower at
 * every shape it applied to and were removed in ABI v26, round 6.) */

/* softmax(scale * Q K^T) V, head dim 64, bf16 in/out, fp32 softmax; element (b,h,n,d) at
 * ptr + b*bs + n*rs + h*64 + d.  q_rows_per_wg: 0 (auto: tile chosen by shape) or an explicit code
 * 5128 / 5256 / 6128 / 6256 (128 / 256 query rows per workgroup, 64 / 128-key macro tiles).
 * Replaces F.scaled_dot_product_attention (iggt/layers/attention.py:60-66). */
int iggt_flash_attn_bf16_d64(const void* q, const void* k, const void* v, void* o, int B, int H,
                             int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                             long v_bs, long v_rs, long o_bs, long o_rs, float scale,
                             int q_rows_per_wg, void* stream);
int iggt_flash_attn_f16_d64(const void* q, const void* k, const void* v, void* o, int B, int H,
                            int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                            long v_bs, long v_rs, long o_bs, long o_rs, float scale,
                            int q_rows_per_wg, void* stream);

/* The same attention with a STATIC softmax bound (csrc/attention_v3.hip): q must already carry scale * log2(e)
 * (iggt_qknorm_rope_* with q_scale) and qkmax[16 + h] >= max_j |k_j| (Euclidean norms of the 16-bit head vectors; entries
 * 0..15, the q maxima, are not read any more: the kernel takes every query row's own norm from its operand fragments).  By
 * Cauchy-Schwarz every score of query row i is <= |q_i| * qkmax[16 + h], so the numerators are 2^(s - bound_i) from the
 * first tile on: no running maximum, no rescale.  Rows whose numerators would sink into the 16-bit subnormal range (row sum
 * below a fixed threshold) get their query tile flagged in `flags` (int[flags_len], scratch, >= B * H * ceil(Nq / 128)
 * entries, zeroed here) and are recomputed by the online-max kernel in the same call, so the result meets the tolerance of
 * iggt_flash_attn_* for ANY input.  part_ws (NULL or part_ws_len >= iggt_flash_attn_static_ws_bytes(..) bytes of scratch)
 * lets a grid too small for the chip split the keys into ranges whose partial results are folded by a second kernel.
 * guard (NULL or int[8], persistent per call site, initialised to {-1, 0, ...}) makes the launch adaptive: the gated
 * online-max pass counts the flagged tiles and, when more than 1/8 were flagged, lets the next 16 calls skip the static
 * kernel (flag every tile at once) before it is tried again; a call site that has never been measured (guard[0] < 0)
 * inherits the verdict of guard_prev (NULL or the guard of the same kind of launch one layer earlier).  guard[1..3] =
 * flagged tiles (-1: skipped) / tiles / calls of the last launch.  Worst case of a launch: ~1.05x the online-max kernel
 * averaged over calls instead of static + online-max.  Same reference operation.
 * ABI 23 (round 4): guard is int[8], initialised to {-1, 0, 0, 0, 0, 0, 0, 0}; guard[4] = mode of the static kernel (0: norm
 * bound, 1: estimated shift), guard[5] = rows handed to the online-max pass one by one (-1: skipped).
 * est_ws (NULL: the behaviour above; else est_ws_len >= iggt_flash_attn_static_est_ws_bytes(..) bytes of 16-byte aligned
 * scratch) turns on, for one-pass launches, the ESTIMATED shift (csrc/attention_est.hip): as soon as ANY tile is flagged under
 * the norm bound (mode 0: trained-like q/k-norm affines, sink keys, register tokens of outlying norm) the guard moves the call
 * site to mode 1, in which a pre-pass takes every row's exact maximum over a key sample -- the first key_nspecial keys of every
 * key_period keys (the special tokens of each view; 0: none), ~Nk / 64 strided keys, and the keys whose norm exceeds half the
 * head's maximum -- and the static kernel shifts row i by min(norm bound, sampled maximum + headroom).  Rows are then handed
 * over ONE BY ONE (a row whose true maximum lies beyond the headroom ends with a non-finite accumulator and is marked like a
 * row whose sum is below the threshold): a small kernel compacts them into ascending lists per (batch, head), lists of up to
 * Nq / 8 rows get an exact second static pass (exact row maxima per 1/16 of the keys, the 128-row static kernel per key range
 * under that maximum, a fold), longer ones are recomputed by the online-max pass.  More than 1/8 of the work redone in mode 1 ->
 * online-max only for 16 calls, then mode 1 again.
 * est_mode: the mode when guard is NULL (0 / 1); ignored otherwise. */
int iggt_flash_attn_static_bf16_d64(const void* q, const void* k, const void* v, void* o, int B, int H,
                                    int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                                    long v_bs, long v_rs, long o_bs, long o_rs, const float* qkmax,
                                    int* flags, int flags_len, void* part_ws, long part_ws_len, int q_rows_per_wg,
                                    int* guard, const int* guard_prev, void* est_ws, long est_ws_len, int key_period,
                                    int key_nspecial, int est_mode, void* stream);
int iggt_flash_attn_static_f16_d64(const void* q, const void* k, const void* v, void* o, int B, int H,
                                   int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                                   long v_bs, long v_rs, long o_bs, long o_rs, const float* qkmax,
                                   int* flags, int flags_len, void* part_ws, long part_ws_len, int q_rows_per_wg,
                                   int* guard, const int* guard_prev, void* est_ws, long est_ws_len, int key_period,
                                   int key_nspecial, int est_mode, void* stream);
long iggt_flash_attn_static_ws_bytes(int B, int H, int Nq, int Nk);
long iggt_flash_attn_static_est_ws_bytes(int B, int H, int Nq, int Nk);
/* number of key ranges iggt_flash_attn_static_* would cut this shape into when given a workspace (1: one pass) */
int iggt_flash_attn_static_ksplit(int B, int H, int Nq, int Nk);

/* The two halves of the split form, for callers that own the key segments themselves (multi-GPU: a rank's own keys while the
 * all-gather of the others is in flight, iggt_official_amd/dist.py):
 *   _partial_: static-bound pass of ALL queries over ONE key segment (k, v, Nk), cut into `ksplit` ranges -> slots
 *              [slot0, slot0 + ksplit) of o_part [slots][B][Nq][H*64] (16-bit, each row normalised by its own row sum),
 *              l_part [slots][B][H][Nq] (fp32 row sums) and c_part (same shape: the shift each row was computed under).
 *              Segments may use DIFFERENT qkmax (own keys: this rank's measured maximum; gathered keys: the maximum over
 *              the gathered rows, iggt_k_rownorm_max_*): the combine step re-weights by 2^(shift_s - max_s shift_s).
 *              seg_len > 0 (segment mode): the ksplit = ceil(Nk / seg_len) ranges are the key segments [s * seg_len,
 *              (s + 1) * seg_len) -- one rank's rows of the gathered buffer each; segment skip_seg (-1: none) is left out
 *              and the later ones move up one slot: every rank of a view-sharded run launches the same (world - 1)-range
 *              grid over the gathered buffer whatever its position.  seg_len = 0: equal shares of the key tiles.
 *              seg_kmax (segment mode; NULL: qkmax for every segment): float [ksplit][32], the qkmax heads of the ranks as
 *              iggt_qknorm_rope_* left them, all-gathered beside the K/V rows -- segment s is bounded by seg_kmax[s][16 + h],
 *              its own rank's key maximum (qkmax may then be NULL).
 *   _combine_: o = sum_s w_s O_s / sum_s w_s over nslots slots, then the flag / online-max fallback pass over the full key
 *              set (k, v, Nk).  q_rows_per_wg: 0 (= 6256) or the code both calls were given.  guard / guard_prev as above
 *              (the partial launches only read them). */
int iggt_flash_attn_static_partial_bf16_d64(const void* q, const void* k, const void* v, int B, int H, int Nq, int Nk,
                                            long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                            const float* qkmax, void* o_part, float* l_part, float* c_part, int slot0,
                                            int ksplit, int seg_len, int skip_seg, const float* seg_kmax, int q_rows_per_wg,
                                            const int* guard, const int* guard_prev, void* stream);
int iggt_flash_attn_static_partial_f16_d64(const void* q, const void* k, const void* v, int B, int H, int Nq, int Nk,
                                           long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                           const float* qkmax, void* o_part, float* l_part, float* c_part, int slot0,
                                           int ksplit, int seg_len, int skip_seg, const float* seg_kmax, int q_rows_per_wg,
                                           const int* guard, const int* guard_prev, void* stream);
int iggt_flash_attn_static_combine_bf16_d64(const void* o_part, const float* l_part, const float* c_part, int nslots,
                                            const void* q, const void* k, const void* v, void* o, int B, int H, int Nq,
                                            int Nk, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                            long o_bs, long o_rs, int* flags, int flags_len, int q_rows_per_wg, int* guard,
                                            const int* guard_prev, void* stream);
int iggt_flash_attn_static_combine_f16_d64(const void* o_part, const float* l_part, const float* c_part, int nslots,
                                           const void* q, const void* k, const void* v, void* o, int B, int H, int Nq,
                                           int Nk, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                           long o_bs, long o_rs, int* flags, int flags_len, int q_rows_per_wg, int* guard,
                                           const int* guard_prev, void* stream);

/* qkmax[16 + h] = largest Euclidean norm of the head-h vectors k[r][h * 64 .. h * 64 + 63] over r < rows (k: 16-bit
 * [rows][ldk], 16 heads; qkmax: float[32 + 32 * 4096] as for iggt_qknorm_rope_*, entries 0..15 untouched): the key half of
 * the static softmax bound for keys that were not produced by this rank's iggt_qknorm_rope_* call -- the gathered K rows of a
 * view-sharded run (no reference counterpart: the reference has no inference parallelism). */
int iggt_k_rownorm_max_bf16(const void* k, long ldk, int rows, float* qkmax, void* stream);
int iggt_k_rownorm_max_f16(const void* k, long ldk, int rows, float* qkmax, void* stream);

/* Writes the name of the kernel instantiation the attention dispatcher picks for a shape into buf (host only, no launch;
 * buf_len >= 96): reports must name the kernel that actually ran. */
int iggt_flash_attn_d64_kernel_name(int B, int H, int Nq, int Nk, int f16, int static_bound, int with_part_ws,
                                    int q_rows_per_wg, char* buf, int buf_len);

/* LayerNorm over C in {128 (no concat / remap), 256, 512, 1024, 2048}; fp32 in ((x0|x1) concatenation when x1 != NULL); out_type 0 = bf16,
 * 1 = fp32, 2 = fp16, 3 (ABI v24; C >= 256, ldo >= 3 C) = fp16 [hi | lo | hi] with hi = fp16(y), lo = fp16(y - hi), the segments C
 * elements apart: the A' operand of a three-pass GEMM of the x3 precision rung (see iggt_flash_attn_x3_f16_d64); optional input-row remap in_row = (r / rows_in) * rows_stride + row_off + r % rows_in and
 * output-row remap out_row = (r / rows_in) * orows_stride + orow_off + r % rows_in (orows_stride > 0).
 * Replaces nn.LayerNorm at iggt/layers/block.py:84,87, iggt/layers/vision_transformer.py:274,
 * iggt/heads/dpt_head.py:232 and the norms of iggt/heads/window_sa.py (HAB / OCAB / wrapper norms). */
int iggt_layernorm_f32(const float* x0, long ld0, const float* x1, long ld1, const float* w,
                       const float* b, void* out, long ldo, int out_type, int rows, int C,
                       float eps, int rows_in, int rows_stride, int row_off, int orows_stride, int orow_off,
                       void* stream);

/* Per-head LayerNorm(64) on q,k + 2-D RoPE (+ optional v copy) on a 16-bit [T][3*1024] qkv matrix.
 * cos_t/sin_t: fp32 [max_pos+1][16], 16-byte aligned.  heads_per_group in {1,2,4,8} writes k / v in head-group layout -- head h of
 * token t at out + (h / hpg) * group_stride + t * ld + (h % hpg) * 64 -- so that the multi-GPU K/V all-gather can be
 * pipelined over head groups (iggt_official_amd/dist.py); 0 or 16: flat rows.  q_scale (> 0; 1 = none) is folded into the q
 * output (the softmax scale * log2 e for iggt_flash_attn_static_*); qkmax (NULL, or float[32 + 32 * 4096]: 32 results + scratch
 * for the per-block maxima): entries 0..15 receive the largest norm of the written q head vectors per head, 16..31 of k.  Replaces iggt/layers/attention.py:54-58 and
 * iggt/layers/rope.py:119-188 (positions: iggt/models/aggregator.py:236-245). */
int iggt_qknorm_rope_bf16(const void* qkv, long ld_in, void* q_out, long ldq, void* k_out, long ldk,
                          void* v_out, long ldv, const float* qw, const float* qb, const float* kw,
                          const float* kb, const float* cos_t, const float* sin_t, int T, int P,
                          int gw, int patch_start, float eps, int heads_per_group, long k_group_stride,
                          long v_group_stride, float q_scale, float* qkmax, void* stream);
int iggt_qknorm_rope_f16(const void* qkv, long ld_in, void* q_out, long ldq, void* k_out, long ldk,
                         void* v_out, long ldv, const float* qw, const float* qb, const float* kw,
                         const float* kb, const float* cos_t, const float* sin_t, int T, int P,
                         int gw, int patch_start, float eps, int heads_per_group, long k_group_stride,
                         long v_group_stride, float q_scale, float* qkmax, void* stream);

/* ImageNet-normalise + im2row of 14x14 patches: img fp32 [S][3][H][W] -> bf16 (out_f16 = 0) or fp16 (1)
 * [S*gh*gw][Kpad]; out_f16 = 2 (ABI v24): fp16 [S*gh*gw][3 Kpad] = [hi | lo | hi] (x3 precision rung).  Replaces iggt/models/aggregator.py:206 and the unfold half of iggt/layers/patch_embed.py:75. */
int iggt_im2row_patch14(const float* img, void* out, int out_f16, int S, int H, int W, int Kpad, void* stream);

/* ---- x3 precision rung of the trunk blocks (ABI v24, round 5; csrc/x3.hip) ------------------------------------------------
 * Every MFMA operand of an escalated block travels as an fp16 PAIR x = hi + lo (hi = fp16(x), lo = fp16(x - hi): 22 significant
 * bits) and every product takes three fp16 MFMA passes, a.b ~= a_hi.b_hi + a_lo.b_hi + a_hi.b_lo.  The GEMMs are ordinary
 * iggt_gemm_f16 calls over a concatenated K axis, A' = [A_hi | A_lo | A_hi] against W' = [W_hi | W_hi | W_lo]; the entry points
 * below produce the split operands and run the attention on them.
 *
 * iggt_qkv_split_f16: qkv fp32 [T][ld_in] (3 x 1024 columns: q | k | v, the fp32 output of the qkv GEMM) -> optional per-head
 * LayerNorm(64) on q and k (qw / qb / kw / kb, all NULL: none) -> optional 2-D RoPE (cos_t / sin_t fp32 [max_pos + 1][16], both
 * NULL: none; token t sits at position t % P of its view, the first patch_start positions are un-rotated, the rest a row-major
 * grid gw wide) -> q multiplied by q_scale (> 0: softmax scale * log2 e) -> fp16 pairs: the hi part of token t at q_out + t * ldq
 * (1024 columns), the lo part q_lo ELEMENTS behind it; k and v likewise.  Replaces iggt/layers/attention.py:50-58 and
 * iggt/layers/rope.py:119-188 for escalated blocks. */
int iggt_qkv_split_f16(const float* qkv, long ld_in, void* q_out, long ldq, long q_lo, void* k_out, long ldk, long k_lo,
                       void* v_out, long ldv, long v_lo, const float* qw, const float* qb, const float* kw, const float* kb,
                       const float* cos_t, const float* sin_t, int T, int P, int gw, int patch_start, float eps, float q_scale,
                       void* stream);
/* x fp32 [rows][ldx] (N columns, N % 4 == 0) -> act (0 none, 1 exact erf GELU: nn.GELU of iggt/layers/mlp.py:34) -> fp16
 * out [rows][ldo >= 3 N] = [hi | lo | hi]. */
int iggt_split3_f16(const float* x, long ldx, void* out, long ldo, int rows, int N, int act, void* stream);
/* softmax(q k^T) v, head dim 64, token-major like iggt_flash_attn_f16_d64, on fp16 PAIRS: q (already carrying scale * log2 e), k and
 * v each as a hi and a lo matrix of identical strides.  S in three passes, online max / row sums in fp32, the numerators split
 * as well, O in three passes; o receives the result as fp16 hi at +0 and, when o_seg > 0, lo at +o_seg and hi again at +2 o_seg
 * (elements): the A' operand of the proj GEMM.  Replaces F.scaled_dot_product_attention at iggt/layers/attention.py:60-66 for
 * escalated blocks (48 instead of 16 MFMAs per 32 x 64 score block). */
int iggt_flash_attn_x3_f16_d64(const void* q, const void* q_lo, const void* k, const void* k_lo, const void* v, const void* v_lo,
                               void* o, long o_seg, int B, int H, int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                               long v_bs, long v_rs, long o_bs, long o_rs, void* stream);

/* Tail of the DPT heads on an NHWC fp32 map x [npix][ldx] (32 input channels): 1x1 convolution to Cout (2..8)
 * channels + activate_head: pts [npix][Cout-1] = act(first Cout-1 channels), conf [npix] = conf_act(last channel).
 * act: 0 linear, 1 exp, 2 relu, 3 inv_log, 4 sigmoid, 5 norm;  conf_act: 0 expp1, 1 expp0, 2 sigmoid.
 * Replaces scratch.output_conv2[2] (iggt/heads/dpt_head.py:121-128) and iggt/heads/head_act.py:61-125. */
int iggt_head_tail_f32(const float* x, long ldx, const float* w, const float* b, float* pts, float* conf,
                       long npix, int Cout, int act, int conf_act, void* stream);

/* Window attention of the part head, head dim 32 or 64, fp32: 8x8 query windows, ow x ow key/value windows at stride 8 with
 * `pad` zero padding (ow = 8, pad = 0: HAB self-attention; ow = 12, pad = 2: OCAB), optional additive bias
 * [heads][ow*ow][64].  q: NHWC map (q_mode 0) or window-major [nW][64][q_ld] (q_mode 1); k, v, o: NHWC maps
 * [b][h][w][ld]; channel offsets are folded into the pointers; head hd uses channels [head_dim * hd, head_dim * (hd + 1)).
 * Replaces window_partition + attention core + window_reverse (iggt/heads/window_sa.py:71-81,163-227) and
 * Unfold + biased softmax attention of OCAB (window_sa.py:229-319). */
int iggt_window_attn_f32(const float* q, long q_ld, int q_mode, const float* k, long k_ld, const float* v,
                         long v_ld, float* o, long o_ld, const float* bias, int b, int h, int w, int heads,
                         int head_dim, int ow, int pad, float scale, void* stream);

/* Fused tail of the DPT heads: x [N][Hi][Wi][128] fp32 NHWC -> bilinear upsample (align_corners) to (Ho, Wo) +
 * separable position map (xpart [Wo][64] on channels 0..63, ypart [Ho][64] on 64..127; both NULL: none) +
 * conv3x3 128 -> 32 (weights as packed for iggt_conv2d_nhwc_f32: bf16 hi / lo [32][9*128] tap-major, bias b1) + ReLU +
 * conv1x1 32 -> Cout (w2 [Cout][32], b2) + activate_head: pts [N][Ho][Wo][Cout-1], conf [N][Ho][Wo].
 * act: 0 linear, 1 exp, 2 relu, 3 inv_log, 4 sigmoid;  conf_act: 0 expp1, 1 expp0, 2 sigmoid.
 * Replaces custom_interpolate + _apply_pos_embed + scratch.output_conv2 (iggt/heads/dpt_head.py:251-256,274-284,
 * 121-128,484-509) + activate_head (iggt/heads/head_act.py:61-125) in one pass.
 * out_nchw != 0 (ABI v26): the part head's tail (iggt/heads/part_head.py:228-243: same stages, NO activation): all Cout channels
 * un-activated as planes pts [N][Cout][Ho][Wo], conf unused (may be NULL), act / conf_act ignored. */
int iggt_dpt_tail_f32(const float* x, int N, int Hi, int Wi, int Ho, int Wo, const float* xpart, const float* ypart,
                      const void* w_hi, const void* w_lo, const float* b1, const float* w2, const float* b2,
                      float* pts, float* conf, int Cout, int act, int conf_act, int out_nchw, void* stream);

/* Mean-input compensation of the 16-bit weight rounding (no counterpart in the reference, which is fp32 on the
 * CPU path this repository is checked against; see iggt_official_amd/precision.py):
 *   iggt_colmean_h16:      mu[k] = mean over rows 0, row_step, 2*row_step, ... of the 16-bit matrix x [rows][K]
 *   iggt_bias_correct_h16: out[n] = (bias ? bias[n] : 0) + sum_k dw[n][k] * mu[k],  dw = W - round16(W) as 16-bit
 * The corrected bias replaces the Linear's bias in the following iggt_gemm_* call (f16: 0 = bf16, 1 = fp16). */
int iggt_colmean_h16(const void* x, long ld, int rows, int K, int row_step, int f16, float* mu, void* stream);
int iggt_bias_correct_h16(const void* dw, long ldw, int N, int K, const float* mu, const float* bias, float* out,
                          int f16, void* stream);
/* Debug telemetry of the fp16 operand format: adds to *counter (device unsigned long long) the number of entries of the 16-bit
 * matrix x [rows][cols] (row stride ld) that are saturated (|x| = 65504, i.e. a clamped store) or not finite (f16 = 1), or
 * not finite (f16 = 0, bf16).  No counterpart in the reference (its GPU mode is bf16 autocast, demo.py:193-195). */
int iggt_count_saturated_h16(const void* x, long ld, int rows, int cols, int f16, void* counter, void* stream);

/* dst[s][row_off + r][:] = (s == 0 && first_view_is_zero ? src0 : src1)[r][:]  (fp32).
 * Replaces iggt/layers/vision_transformer.py:222-234 and iggt/models/aggregator.py:230-234,338-361. */
int iggt_write_special_tokens(float* dst, long view_stride, long ldd, const float* src0,
                              const float* src1, int S, int nrows, int row_off, int C,
                              int first_view_is_zero, void* stream);

/* Implicit-GEMM convolution on MFMA, NHWC fp32 in/out, bf16 (prec 1) or split-bf16 hi+lo (prec 3, fp32-grade)
 * operands with fp32 accumulate.  w_hi/w_lo: bf16 [Cout][KH*KW*Cin] tap-major.  GEMM rows = (img, oy, ox) over an
 * Ho x Wo placement grid; input pixel = (oy*stride - pad_y + ky, ox*stride - pad_x + kx); output pixel =
 * (oy*osy + ooy + py, ox*osx + oox + px) in an Hout x Wout map, where (py, px) = phase of n / cout_phys when
 * ps > 1 (pixel shuffle: ConvTranspose2d with kernel == stride).  relu_in: ReLU on loaded inputs; res: residual
 * added after the activation (relu_res: add max(res,0)); res2: optional second residual (plain add); act: 0 none, 1 ReLU, 2 LeakyReLU(0.01), 3 GELU(erf).
 * prec 2 (needs the _ws entry point): w_hi = fp16(W) [Cout][KH*KW*Cin], w_lo = bf16(W - fp16(W)); the MFMAs multiply
 * fp16 hi + fp16 lo activations (exact to 2^-22) by w_hi -- two passes instead of three -- and the epilogue adds, in place of
 * the bias, bias + mean_input * w_lo summed over the taps that fall inside the image for the placement's border class
 * (first / middle / last row x column: nine vectors, computed per call by two small kernels from a strided sample of the
 * input; csrc/conv_meancomp.hip).  Returns -6 for a geometry without such a description (pad > stride) or without workspace.
 * Replaces the nn.Conv2d / nn.ConvTranspose2d of iggt/heads/dpt_head.py:72-128,345-411,441-479,
 * iggt/heads/adaptor.py:9-35,152-175 and iggt/heads/window_sa.py:40-47,383-391. */
int iggt_conv2d_nhwc_f32(const float* x, int ldx, const void* w_hi, const void* w_lo, const float* bias,
                         const float* res, const float* res2, int ldr, float* y, int ldy, int Nimg, int Hi, int Wi,
                         int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_y,
                         int pad_x, int Hout, int Wout, int osy, int osx, int ooy, int oox,
                         int cout_phys, int ps, int relu_in, int relu_res, int act, int prec,
                         void* stream);
/* The same with a scratch buffer (16-byte aligned, used by one call at a time on a stream) that lets plain convolutions with
 * few output tiles and a long K (1024 channels x 9 taps on the 19^2 / 37^2 maps of 3-4 views) split the K loop over
 * several workgroups per tile; the partial tiles are added in a fixed order by a finalize pass.  ws == NULL: as above.
 * prec 2 keeps its channel sums and correction vectors in the first (64 Cin + 9 Cout) * 4 + 256 bytes of ws. */
int iggt_conv2d_nhwc_f32_ws(const float* x, int ldx, const void* w_hi, const void* w_lo, const float* bias,
                         const float* res, const float* res2, int ldr, float* y, int ldy, int Nimg, int Hi, int Wi,
                         int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_y,
                         int pad_x, int Hout, int Wout, int osy, int osx, int ooy, int oox,
                         int cout_phys, int ps, int relu_in, int relu_res, int act, int prec,
                         void* ws, long ws_bytes, void* stream);

/* Bilinear resize with align_corners=True, NHWC fp32, optional separable additive position map
 * (xpart [Wo][C/2] for channels [0,C/2), ypart [Ho][C/2] for [C/2,C)).
 * Replaces custom_interpolate (iggt/heads/dpt_head.py:484-509) and _apply_pos_embed (dpt_head.py:274-284). */
int iggt_bilinear_ac_nhwc_f32(const float* x, int ldx, float* y, int ldy, int N, int Hi, int Wi, int Ho,
                              int Wo, int C, const float* xpart, const float* ypart, void* stream);

/* ---- fp32 small operators of the heads (csrc/smallops.hip) ------------------------------------------------------- */

/* Exact-fp32 nn.Linear for skinny problems:  out[m][n] = act(sum_k x[m][k] w[n][k] + bias[n]) * gamma[n] (+ res[m][n]),
 * fp32 MFMA (v_mfma_f32_32x32x2_f32).  act: 0 none, 1 exact GELU, 2 ReLU, 3 SiLU, 4 sigmoid.  res may alias out.
 * Replaces the Linears of iggt/heads/camera_head.py:83-154 (trunk blocks through iggt/layers/block.py:81-107,
 * embed_pose, poseLN_modulation, pose_branch) and ChannelAttention's pooled 1x1 convs (iggt/heads/window_sa.py:26-37). */
int iggt_linear_f32(const float* x, long ldx, const float* w, long ldw, const float* bias, const float* gamma,
                    const float* res, long ldr, float* out, long ldo, int M, int N, int K, int act, void* stream);
/* The same with a scratch buffer that lets the kernel split K over several workgroups per output tile (the op streams the
 * fp32 weights once and needs >= ~1000 workgroups to keep enough loads in flight; a 2048 -> 2048 Linear has 64 tiles):
 * ws = iggt_linear_f32_ws_bytes() bytes, 16-byte aligned, used by one call at a time (calls on one stream).  The partial
 * sums are added in a fixed order by a second small kernel: results do not depend on scheduling.  ws == NULL: one
 * workgroup per tile, as iggt_linear_f32. */
int iggt_linear_f32_ws(const float* x, long ldx, const float* w, long ldw, const float* bias, const float* gamma,
                       const float* res, long ldr, float* out, long ldo, int M, int N, int K, int act, void* ws,
                       long ws_bytes, void* stream);
long iggt_linear_f32_ws_bytes(void);

/* softmax(scale * q k^T) v in fp32, head_dim 32 / 64 / 128; element (b, h, n, d) at ptr + b*bs + n*rs + h*head_dim + d.
 * Replaces the attention core of the camera trunk (iggt/layers/attention.py:60-66 with 16 heads x 128 over the S views)
 * and CrossAttention / Attention of the part head (iggt/heads/block.py:120-150, 212-242: 8 heads x 32). */
int iggt_attn_f32(const float* q, const float* k, const float* v, float* o, int B, int H, int Nq, int Nk,
                  int head_dim, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs, long o_rs,
                  float scale, void* stream);

/* out = gate * (LayerNorm_noaffine(x; eps) * (1 + scale) + shift) + x, rows x C fp32 (shift / scale / gate share row
 * stride ldm).  Replaces iggt/heads/camera_head.py:130-134,157-162. */
int iggt_adaln_modulate_f32(const float* x, long ldx, const float* shift, const float* scale, const float* gate,
                            long ldm, float* out, long ldo, int rows, int C, float eps, void* stream);

/* pred = first ? delta : pred + delta;  out = activate_pose(pred) for absT_quaR_FoV with trans / quat "linear", FoV "relu";
 * n rows of 9.  Replaces iggt/heads/camera_head.py:139-151 and iggt/heads/head_act.py:9-57. */
int iggt_pose_update_f32(const float* delta, float* pred, float* out, int n, int first, void* stream);

/* 1x1 convolution of an NHWC fp32 map with 32 channels (pixel stride ldx) to Cout <= 8 NCHW planes
 * y[img][c][hw]; npix = images * hw.  Replaces scratch.output_conv2[2] of the part head
 * (iggt/heads/part_head.py:128,240-243; no activation). */
int iggt_conv1x1_c32_nchw_f32(const float* x, long ldx, const float* w, const float* b, float* y, long hw, long npix,
                              int Cout, void* stream);

/* pose encoding [n][9] (T, quaternion xyzw, fov_h, fov_w) -> extrinsics [n][3][4] = [R | T] and, when intri != NULL,
 * intrinsics [n][3][3] for H x W images.  Replaces pose_encoding_to_extri_intri (iggt/utils/pose_enc.py:65-130) and
 * quat_to_mat (iggt/utils/rotation.py:14-44). */
int iggt_pose_to_extri_intri_f32(const float* pose, float* extri, float* intri, int n, int H, int W, void* stream);

/* depth [S][H][W] + extrinsics (camera from world) + intrinsics -> world points [S][H][W][3] (fp64 arithmetic inside,
 * as the reference's numpy path).  Replaces unproject_depth_map_to_point_map / depth_to_world_coords_points /
 * depth_to_cam_coords_points / closed_form_inverse_se3 (iggt/utils/geometry.py:151-181,184-236,239-268,271-330). */
int iggt_unproject_depth_f32(const float* depth, const float* extri, const float* intri, float* out, int S, int H, int W,
                             void* stream);

/* ---- image preprocessing in front of the forward path (csrc/preprocess.hip) ---------------------------------------- */

/* Pillow-exact 8-bit bicubic resize of an RGB image, uint8 HWC [Hi][Wi][3] -> [Ho][Wo][3]: horizontal pass into tmp
 * [Hi][Wo][3], vertical pass into out; bounds = int[n][2] (first source index, tap count), kk = int[n][ksize] coefficients
 * scaled by 2^22 (host-built exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc).  Bit-identical to
 * PIL.Image.resize(size, BICUBIC), i.e. to iggt/utils/load_fn.py:85. */
int iggt_resize_bicubic_u8(const void* in, int Hi, int Wi, const int* hbounds, const int* hkk, int hksize,
                           const int* vbounds, const int* vkk, int vksize, void* tmp, void* out, int Ho, int Wo,
                           void* stream);

/* ToTensor + crop + constant pad: dst fp32 [3][Hd][Wd]; the h x w window of src (uint8 HWC [Hs][Ws][3]) at (crop_y, crop_x)
 * lands at (pad_y, pad_x) divided by 255, everything else is pad_value.  Replaces torchvision ToTensor, the centre crop
 * and the white padding of iggt/utils/load_fn.py:86-123. */
int iggt_u8hwc_to_f32chw(const void* src, int Hs, int Ws, float* dst, int Hd, int Wd, int crop_y, int crop_x, int pad_y,
                         int pad_x, int h, int w, float pad_value, void* stream);

/* ---- post-processing behind the forward path (csrc/postprocess.hip; reference iggt/utils/misc.py, demo.py:365-400) ------ */

/* Exact k-nearest neighbours (k <= 32, the point itself excluded) among M points fp32 [M][3], all in one batch, as
 * torch_cluster.knn_graph(points, k, batch=0, loop=False) in knn_avg_features_pyg (misc.py:61-65).
 * 1. iggt_knn_morton_codes: 30-bit Morton code of every point on a 1024^3 grid around (cx,cy,cz), cell = 1/inv_cell
 *    (the grid only orders the points; any centre / cell gives the exact result).  The caller sorts the codes.
 * 2. iggt_knn_search: order = argsort of the codes (int64 [M]); sorted_ws = 16 * 256 * ceil(M/256) bytes, boxes =
 *    6 * ceil(M/256) floats of scratch.  idx_out int32 [M][k]: row i = the neighbours of point i, ascending (distance,
 *    index), -1 where fewer than k other points exist; d2_out (may be NULL) fp32 [M][k] squared distances. */
int iggt_knn_morton_codes(const float* points, long M, float cx, float cy, float cz, float inv_cell, int* codes,
                          void* stream);
int iggt_knn_search(const float* points, const long* order, long M, int k, void* sorted_ws, float* boxes, int* idx_out,
                    float* d2_out, void* stream);

/* out[i][:] = mean over the valid neighbours j of feat[idx[i][j]][:] (0 when there is none): scatter_mean over the kNN
 * edges, misc.py:68-71.  feat / out fp32 [M][F]. */
int iggt_knn_mean_features_f32(const float* feat, const int* idx, long M, int k, int F, float* out, void* stream);

/* First and second moments for the PCA of apply_pca_colormap (misc.py:272-331), C <= 16: every block b writes
 * iggt_moments_width(C) floats to partials[b]: sum(x - shift) per channel (padded to CT = 4 / 8 / 16), then the upper
 * triangle of sum((x - shift)(x - shift)^T) row by row over CT channels.  The host adds the partials in fp64. */
int iggt_moments_f32(const float* x, long M, int C, const float* shift, float* partials, int nblocks, void* stream);
int iggt_moments_width(int C);

/* out[M][3] = x[M][C] @ v[C][3] (misc.py:299), and the in-place percentile stretch of misc.py:309-326:
 * channel j -> clamp((v - lohi[j]) / (lohi[3+j] - lohi[j]), 0, 1), or 0.5 where lohi[3+j] <= lohi[j]. */
int iggt_project3_f32(const float* x, long M, int C, const float* v, float* out, void* stream);
int iggt_stretch3_f32(float* img, long M, const float* lohi, void* stream);

/* out[i] = ref_labels[argmin_j |query[i] - ref[j]|^2] (first minimum), C <= 16: the nearest-labelled-pixel fill of the
 * clustering step (NearestNeighbors(n_neighbors=1), misc.py:130-144). */
int iggt_nn1_label_f32(const float* query, long Mq, const float* ref, long Mr, int C, const int* ref_labels, int* out,
                       void* stream);
/* The search behind it with the samples split over nsplit workgroups per tile of 256 queries (ABI v22): part s scans a contiguous
 * range of ceil(Mr / nsplit) rows (rounded up to whole tiles of 256) and writes best_d2 [nsplit][Mq] (inf: empty range) and
 * best_idx [nsplit][Mq] (row of ref, -1: none) -- the first minimum inside its range.  The caller takes the smallest distance, then
 * the smallest index, over the planes: few queries against many samples no longer run as a handful of long workgroups. */
int iggt_nn1_search_split_f32(const float* query, long Mq, const float* ref, long Mr, int C, int nsplit, float* best_d2,
                              int* best_idx, void* stream);
/* The same result from a local search (ABI v22).  query [Mq][C] and ref [Mr][C] are given ordered along ONE space-filling curve
 * (any order is exact; a spatial one lets far tiles be skipped), with the per-channel bounding box of every tile of 256
 * consecutive rows: qbox_lo / qbox_hi [ceil(Mq / 256)][C], rbox_lo / rbox_hi [ceil(Mr / 256)][C].  ref_idx [Mr]: the position
 * each ref row had in the caller's original order -- among equal distances the smallest ref_idx wins, which reproduces the
 * "first minimum" of iggt_nn1_label_f32 on the unsorted arrays.  ref_labels [Mr] in the given (sorted) order; out [Mq] likewise. */
int iggt_nn1_label_tiled_f32(const float* query, long Mq, const float* qbox_lo, const float* qbox_hi, const float* ref, long Mr,
                             const float* rbox_lo, const float* rbox_hi, int C, const int* ref_idx, const int* ref_labels,
                             int* out, void* stream);

/* ---- track head: the query_points path of IGGT.forward (csrc/track.hip; reference iggt/models/vggt.py:220-227,
 *      iggt/heads/track_head.py:75-109, iggt/heads/track_modules/) ------------------------------------------------------ */

/* out[r][:] = LayerNorm(x[r][:] (+ x2[r][:] when x2 != NULL); w, b, eps) over rows of ANY width C (fp32, unaligned rows
 * allowed; the optional addend is the `tokens + init_tokens` of blocks.py:139-142).  Replaces the
 * nn.LayerNorm(388 / 384) layers of EfficientUpdateFormer / AttnBlock / CrossAttnBlock (track_modules/blocks.py:44,48,
 * modules.py:168-169,203-205) and GroupNorm(1, 128) on a [rows][128] matrix (base_track_predictor.py:74,183). */
int iggt_layernorm_rows_f32(const float* x, long ldx, const float* x2, long ldx2, const float* w, const float* b,
                            float* out, long ldo, int rows, int C, float eps, void* stream);

/* y[n][H/2][W/2][C] = 2 x 2 average pooling, stride 2 (floor sizes), NHWC fp32, C % 4 == 0.  Replaces F.avg_pool2d of
 * CorrBlock's pyramid (track_modules/blocks.py:170-180). */
int iggt_avgpool2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);

/* out[n][0:C] = bilinear sample of feat [H][W][C] at pixel (xy[n][0], xy[n][1]), align_corners=True, border padding.
 * Replaces sample_features4d / bilinear_sampler (track_modules/utils.py:124-226) for the query features. */
int iggt_sample_points_nhwc_f32(const float* feat, int H, int W, int C, const float* xy, long ldxy, float* out,
                                long ldo, int N, void* stream);

/* CorrBlock.corr_sample (track_modules/blocks.py:189-241, compute_corr_level 244-249) without the correlation volume.
 * fmaps / Hs / Ws: HOST arrays of `levels` device pointers / sizes, level l = [S][Hs[l]][Ws[l]][C] fp32 (NHWC pyramid);
 * feats [N][S][C], coords [N][S][2] (level-0 pixels, x then y), both track-major.  out[(n*S + s)][l*(2r+1)^2 + ix*(2r+1) + iy]
 * = bilinear sample (zero padding, align_corners) of <feats[n][s], fmap_l[s]> / sqrt(C) at coords / 2^l + (ix - r, iy - r);
 * columns [levels*(2r+1)^2, ldo) are set to 0 (row padding for the Linear that follows).  C == 128, radius == 4. */
int iggt_track_corr_f32(const float* const* fmaps, const int* Hs, const int* Ws, int levels, int S, int C,
                        const float* feats, const float* coords, int N, int radius, float* out, long ldo, void* stream);

/* out[n][0:2Ch] = get_2d_sincos_pos_embed(2Ch, (H, W)) sampled like sample_features4d at xy[n] (track_modules/utils.py:17-87,
 * base_track_predictor.py:152-154).  tabx [W][Ch] / taby [H][Ch]: the 1-D tables [sin | cos](pos * omega) the 2-D one is
 * built from (its first Ch channels depend on x only, the others on y only). */
int iggt_track_posemb_f32(const float* tabx, const float* taby, int H, int W, int Ch, const float* xy, long ldxy,
                          float* out, long ldo, int N, void* stream);

/* Transformer input of one refinement iteration (base_track_predictor.py:139-163): row (n, s) =
 * [get_2d_embedding(flow, E) (2E) | flow / max_scale (2) | flow / max_scale (2) | corr[n][s][0:Cc] | feats[n][s][0:Cf]]
 * + pos[n][:] + ref[s > 0][:], flow = coords[n][s] - coords[n][0]; row width D = 2E + 4 + Cc + Cf. */
int iggt_track_tokens_f32(const float* coords, const float* corr, long ldc, int Cc, const float* feats, long ldf, int Cf,
                          const float* pos, long ldp, const float* ref, float* out, long ldo, int N, int S, int E,
                          float max_scale, void* stream);

/* coords[n][s] += delta[n][s][0:2] for s > 0 (frame 0 stays the query point); pred[s][n] = coords[n][s] * stride
 * (base_track_predictor.py:168-195). */
int iggt_track_update_f32(float* coords, const float* delta, long ldd, float* pred, int N, int S, float stride,
                          void* stream);

/* ---- HDBSCAN behind cluster_features_to_masks_mv (iggt/utils/misc.py:81-170; the library call at misc.py:123-129) ----------------
 * core[i] = Euclidean distance from x[i] to its k-th nearest row of x [M][C] (fp32, C in {3, 8, 16}), the row itself counted
 * (k = min_samples <= 128): `NearestNeighbors(n_neighbors=min_samples).kneighbors(X)[0][:, -1]` of the reference's clusterer.
 * box_lo / box_hi [ceil(M / 256)][C]: per-channel minimum / maximum of every tile of 256 consecutive rows -- any row order gives
 * the exact result; an order that keeps near rows together (a space-filling curve) lets far tiles be skipped. */
int iggt_hdbscan_core_dist_f32(const float* x, long M, int C, int k, const float* box_lo, const float* box_hi, float* core,
                               void* stream);
/* One Boruvka round of the minimum spanning tree of the mutual-reachability graph max(core_i, core_j, |x_i - x_j|).  All arrays
 * are ordered by component id: x [M][C], core2 [M] (= core^2), comp [M], idx [M] (original index of the point at each position:
 * ties are broken on (min, max) of the ORIGINAL indices), tile_lo / tile_hi [ceil(M / 256)] (smallest / largest component id inside
 * each 256-position tile), box_lo / box_hi [ceil(M / 256)][C] (bounding boxes of the tiles, as above).  Writes, per position, the
 * squared weight best_w2 (inf: no other component) and the POSITION best_p of the cheapest partner outside its own component
 * (-1: none).
 * comp_bound (ABI v22; nullable): [max component id + 1] words the caller fills with 0x7f800000 (+inf) before the launch.  The
 * round only needs the cheapest outgoing edge PER COMPONENT: workgroups whose points share one component publish their best
 * squared weight here (atomic minimum on the float bits) and skip candidates strictly worse than the published value, so
 * points in the interior of a large component stop early.  With it, best_w2 / best_p of a point may read (inf, -1) although a
 * foreign point exists; the minimum over each component under the order (weight, min index, max index) is unchanged.
 * nsplit (ABI v22; 1 .. 64): every block of 512 positions is searched by nsplit workgroups, each over every nsplit-th tile of the
 * walk; best_w2 / best_p are then [nsplit][M] planes (plane s = what workgroup s of each block found) and the caller takes,
 * per position, the minimum over the planes under the same total order.  1 = one plane, the result itself. */
int iggt_hdbscan_nearest_foreign_f32(const float* x, const float* core2, const int* comp, const int* idx, const int* tile_lo,
                                     const int* tile_hi, const float* box_lo, const float* box_hi, long M, int C,
                                     float* best_w2, int* best_p, unsigned* comp_bound, int nsplit, void* stream);
/* HOST function (no GPU needed): spanning tree edges (eu[e], ev[e], ew[e]), e < n_points - 1, of the mutual-reachability graph
 * -> flat HDBSCAN labels [n_points] (-1 = noise; clusters numbered like scikit-learn's): single-linkage dendrogram, condensed
 * tree for min_cluster_size, excess-of-mass selection, cluster_selection_epsilon, allow_single_cluster.  Returns 0, or a
 * negative code for malformed input (-3: the edges do not form a tree). */
int iggt_hdbscan_labels_from_mst(const int* eu, const int* ev, const float* ew, long n_points, int min_cluster_size,
                                 double cluster_selection_epsilon, int allow_single_cluster, int* labels);

#ifdef __cplusplus
}
#endif
#endif /* IGGT_HIP_H */
