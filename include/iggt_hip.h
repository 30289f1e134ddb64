/* libiggt_hip.so -- C ABI of the MI355X (gfx950) kernels behind the IGGT forward path.
 *
 * Boundary contract: the reference's drop-in surface is its Python
 * nn.Module API (iggt.models.vggt.IGGT etc.); the reference has no FFI of its own.  This header is
 * the layer *below* that API: plain device pointers, sizes and a hipStream_t (as void*), no torch
 * types.  The Python host (iggt_official_amd/_C.py, ctypes) binds exactly these symbols; each
 * entry cites the reference operation (file:line under /root/reference) it replaces.
 *
 * Conventions: all pointers are device pointers unless stated; `ld*`/strides are in ELEMENTS;
 * bf16 tensors are passed as void*; every function returns 0 on success, a negative value for an
 * argument-contract violation, or a positive hipError_t.  Functions only enqueue work on
 * `stream` (no host sync, no allocation) and are hipGraph-capturable.
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

/* softmax(scale * Q K^T) V, head dim 64, bf16 in/out, fp32 softmax; element (b,h,n,d) at
 * ptr + b*bs + n*rs + h*64 + d.  q_rows_per_wg: 0 (auto), 128 or 256.
 * Replaces F.scaled_dot_product_attention (iggt/layers/attention.py:60-66). */
int iggt_flash_attn_bf16_d64(const void* q, const void* k, const void* v, void* o, int B, int H,
                             int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                             long v_bs, long v_rs, long o_bs, long o_rs, float scale,
                             int q_rows_per_wg, void* stream);

/* LayerNorm over C in {256,512,1024,2048}; fp32 in ((x0|x1) concatenation when x1 != NULL), bf16 or
 * fp32 out; optional input-row remap in_row = (r / rows_in) * rows_stride + row_off + r % rows_in and
 * output-row remap out_row = (r / rows_in) * orows_stride + orow_off + r % rows_in (orows_stride > 0).
 * Replaces nn.LayerNorm at iggt/layers/block.py:84,87, iggt/layers/vision_transformer.py:274 and
 * iggt/heads/dpt_head.py:232. */
int iggt_layernorm_f32(const float* x0, long ld0, const float* x1, long ld1, const float* w,
                       const float* b, void* out, long ldo, int out_is_f32, int rows, int C,
                       float eps, int rows_in, int rows_stride, int row_off, int orows_stride, int orow_off,
                       void* stream);

/* Per-head LayerNorm(64) on q,k + 2-D RoPE (+ optional v copy) on a bf16 [T][3*1024] qkv matrix.
 * cos_t/sin_t: fp32 [max_pos+1][16].  Replaces iggt/layers/attention.py:54-58 and
 * iggt/layers/rope.py:119-188 (positions: iggt/models/aggregator.py:236-245). */
int iggt_qknorm_rope_bf16(const void* qkv, long ld_in, void* q_out, long ldq, void* k_out, long ldk,
                          void* v_out, long ldv, const float* qw, const float* qb, const float* kw,
                          const float* kb, const float* cos_t, const float* sin_t, int T, int P,
                          int gw, int patch_start, float eps, void* stream);

/* ImageNet-normalise + im2row of 14x14 patches: img fp32 [S][3][H][W] -> bf16 [S*gh*gw][Kpad].
 * Replaces iggt/models/aggregator.py:206 and the unfold half of iggt/layers/patch_embed.py:75. */
int iggt_im2row_patch14(const float* img, void* out, int S, int H, int W, int Kpad, void* stream);

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
 * Replaces the nn.Conv2d / nn.ConvTranspose2d of iggt/heads/dpt_head.py:72-128,345-411,441-479,
 * iggt/heads/adaptor.py:9-35,152-175 and iggt/heads/window_sa.py:40-47,383-391. */
int iggt_conv2d_nhwc_f32(const float* x, int ldx, const void* w_hi, const void* w_lo, const float* bias,
                         const float* res, const float* res2, int ldr, float* y, int ldy, int Nimg, int Hi, int Wi,
                         int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_y,
                         int pad_x, int Hout, int Wout, int osy, int osx, int ooy, int oox,
                         int cout_phys, int ps, int relu_in, int relu_res, int act, int prec,
                         void* stream);

/* Bilinear resize with align_corners=True, NHWC fp32, optional separable additive position map
 * (xpart [Wo][C/2] for channels [0,C/2), ypart [Ho][C/2] for [C/2,C)).
 * Replaces custom_interpolate (iggt/heads/dpt_head.py:484-509) and _apply_pos_embed (dpt_head.py:274-284). */
int iggt_bilinear_ac_nhwc_f32(const float* x, int ldx, float* y, int ldy, int N, int Hi, int Wi, int Ho,
                              int Wo, int C, const float* xpart, const float* ypart, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IGGT_HIP_H */
