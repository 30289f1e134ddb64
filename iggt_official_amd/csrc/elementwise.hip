// HBM-bound fused elementwise / normalisation kernels of the IGGT aggregator.
// All loads/stores are 8- or 16-byte vectors, one wave (or 8-lane group) per reduction row, no LDS.
#include "common.h"
#include "../../include/iggt_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim, fp32 in -> bf16 out (the A operand of the following GEMM).
// Reference: nn.LayerNorm norm1/norm2 in iggt/layers/block.py:41,50,67,84,87 (eps 1e-5 in the
// aggregator blocks, 1e-6 in the DINOv2 blocks, vision_transformer.py:94), final DINOv2 norm
// (vision_transformer.py:274), DPT token norm on the concatenated [frame|global] halves
// (iggt/heads/dpt_head.py:232).  Under the reference's autocast, LayerNorm runs in fp32 and the
// consumer Linear rounds its input to bf16: same rounding point as here.
//
// One wave per row; NV float4 per lane; two-pass (mean, then centred variance) in registers.
// Input may be the concatenation of two row-major matrices (x0 | x1), each C/2 wide, and input rows
// may be remapped (skip the 5 special tokens of each view).
struct LnParams {
    const float* x0;
    const float* x1;  // null: single source of width C
    long ld0, ld1;
    const float* w;
    const float* b;
    bf16_t* out;
    long ldo;
    float* out_f32;  // optional fp32 output instead of the 16-bit one
    int f16;         // 16-bit output format: 0 bf16, 1 fp16
    long split_seg;  // > 0 (fp16 only; round 5, the x3 precision rung of layers/blocks.py): the row is written as fp16 hi at +0, lo =
                     // fp16(y - hi) at +split_seg and hi again at +2 split_seg -- the A' = [A_hi | A_lo | A_hi] operand of a three-pass GEMM
    int rows;
    float eps;
    int rows_in, rows_stride, row_off;  // in_row = (r / rows_in) * rows_stride + row_off + r % rows_in
    int orows_stride, orow_off;         // out_row likewise (orows_stride == 0: out_row = r)
};

template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnParams p) {
    constexpr int C = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    long irow = row, orow = row;
    if (p.rows_in > 0) {
        const int g = row / p.rows_in;
        irow = (long)g * p.rows_stride + p.row_off + (row - g * p.rows_in);
        if (p.orows_stride > 0) orow = (long)g * p.orows_stride + p.orow_off + (row - g * p.rows_in);
    }
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int col = j * 256 + lane * 4;
        const float* src;
        if (p.x1 != nullptr && j >= NV / 2) src = p.x1 + irow * p.ld1 + (col - C / 2);
        else src = p.x0 + irow * p.ld0 + col;
        v[j] = *reinterpret_cast<const f32x4*>(src);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[j][e] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + p.eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int col = j * 256 + lane * 4;
        const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + col);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b + col);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[j][e] - mean) * rstd * w[e] + bb[e];
        if (p.out_f32) {
            *reinterpret_cast<f32x4*>(p.out_f32 + orow * p.ldo + col) = y;
        } else {
            u32x2 o;
            o[0] = p.f16 ? pack_h2<FMT_F16>(y[0], y[1]) : pack_h2<FMT_BF16>(y[0], y[1]);
            o[1] = p.f16 ? pack_h2<FMT_F16>(y[2], y[3]) : pack_h2<FMT_BF16>(y[2], y[3]);
            *reinterpret_cast<u32x2*>(p.out + orow * p.ldo + col) = o;
            if (p.split_seg > 0) {
                u32x2 lo;
                lo[0] = pack_h2<FMT_F16>(y[0] - h2_lo<FMT_F16>(o[0]), y[1] - h2_hi<FMT_F16>(o[0]));
                lo[1] = pack_h2<FMT_F16>(y[2] - h2_lo<FMT_F16>(o[1]), y[3] - h2_hi<FMT_F16>(o[1]));
                *reinterpret_cast<u32x2*>(p.out + orow * p.ldo + p.split_seg + col) = lo;
                *reinterpret_cast<u32x2*>(p.out + orow * p.ldo + 2 * p.split_seg + col) = o;
            }
        }
    }
}

// LayerNorm over C = 128 (window-attention stages of the part head, iggt/heads/window_sa.py:176-181,330-333):
// 32 lanes x one float4 per row, two rows per wave, same two-pass arithmetic; no concat / row remap.
__global__ __launch_bounds__(256) void layernorm128_kernel(const LnParams p) {
    const int sub = threadIdx.x & 31;
    const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= p.rows) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.x0 + row * p.ld0 + sub * 4);
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.0f / 128);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float d = v[e] - mean;
        q += d * d;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q * (1.0f / 128) + p.eps);
    const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + sub * 4);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b + sub * 4);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (v[e] - mean) * rstd * w[e] + bb[e];
    if (p.out_f32) {
        *reinterpret_cast<f32x4*>(p.out_f32 + row * p.ldo + sub * 4) = y;
    } else {
        u32x2 o;
        o[0] = p.f16 ? pack_h2<FMT_F16>(y[0], y[1]) : pack_h2<FMT_BF16>(y[0], y[1]);
        o[1] = p.f16 ? pack_h2<FMT_F16>(y[2], y[3]) : pack_h2<FMT_BF16>(y[2], y[3]);
        *reinterpret_cast<u32x2*>(p.out + row * p.ldo + sub * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Per-head LayerNorm(64) on q and k followed by 2-D RoPE; optional copy of v.
// Reference: iggt/layers/attention.py:54-58 (q_norm/k_norm then rope) and
// iggt/layers/rope.py:119-188.  Head dim 64 = [y-half 32 | x-half 32]; within a half the
// rotation pairs element i with i+16 and angle index i%16:
//     out[i] = t[i]*cos(th_{i%16}) + (i<16 ? -t[i+16] : t[i-16]) * sin(th_{i%16}),  th_f = pos * 100^(-f/16)
// cos/sin come from a host-built fp32 table [max_pos+1][16] (same values torch computes in the
// reference's frequency cache, rope.py:100-117).  Token p of a view: p < patch_start -> pos (0,0)
// (identity), else (y,x) = ((p-ps)/gw + 1, (p-ps)%gw + 1)  (aggregator.py:236-245).
//
// One block (256 threads) per token: thread -> (which = q|k, head, 8-element slice j); the 8 lanes
// of a head reduce with xor-shuffles 1,2,4; the RoPE partner slice is lane^2.
struct QkParams {
    const bf16_t* qkv;  // [T][3*C]
    long ld_in;
    bf16_t* q_out; long ldq;
    bf16_t* k_out; long ldk;
    bf16_t* v_out; long ldv;  // optional
    const float* qw; const float* qb; const float* kw; const float* kb;  // [64] each
    const float* cos_t; const float* sin_t;  // [npos][16]
    int T, P, gw, patch_start;
    float eps;
    int C;  // 1024
    // head-group layout of the k / v outputs (multi-GPU K/V gather pipelined over head groups, dist.py): head h
    // goes to  out + (h / hg) * group_stride + t * ld + (h % hg) * 64.   hg = 16, stride 0: flat [T][C] rows.
    int hg;
    long kgs, vgs;
    // static-bound softmax (attention_v3.hip): q is written pre-multiplied by q_scale (softmax scale * log2 e; 1 = off) and
    // the largest Euclidean norms of the ROUNDED q and k head vectors end up in qkmax[h] / qkmax[16 + h] (nullptr = off):
    // every block writes the maxima of its tokens to qkmax[32 + 32 * block ...], qkmax_reduce_kernel folds them.
    // (A first version used one atomic max per block and (q|k, head): 131k atomics on 32 addresses cost 70 us per call.)
    float q_scale;
    float* qkmax;
};
constexpr int QK_MAX_BLOCKS = 4096;

template <int FMT>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(const QkParams p) {
    const int tid = threadIdx.x;
    const int which = tid >> 7, head = (tid >> 3) & 15, j = tid & 7;
    float nmax2 = 0.f;   // largest squared norm this thread's (q|k, head) has produced (static softmax bound)
    // The affine parameters of this thread's 8 elements live in registers for the whole token loop, and the NEXT token's q / k
    // (and v) pieces are requested before the current token's arithmetic: one 16-byte load in flight per thread left the kernel
    // latency-bound (32 KB in flight per CU: 3.4 TB/s of read + write at 44 k tokens).
    float wreg[8], breg[8];
    {
        const float* w = which ? p.kw : p.qw;
        const float* bb = which ? p.kb : p.qb;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wreg[e] = w[j * 8 + e];
            breg[e] = bb[j * 8 + e];
        }
    }
    const long qk_off = (long)which * p.C + head * 64 + j * 8;
    const bool copy_v = p.v_out != nullptr && tid < 128;
    u32x4 raw_next = {0u, 0u, 0u, 0u}, v_next = {0u, 0u, 0u, 0u};
    if ((int)blockIdx.x < p.T) {
        raw_next = *reinterpret_cast<const u32x4*>(p.qkv + (long)blockIdx.x * p.ld_in + qk_off);
        if (copy_v) v_next = *reinterpret_cast<const u32x4*>(p.qkv + (long)blockIdx.x * p.ld_in + 2 * p.C + tid * 8);
    }
    for (int t = blockIdx.x; t < p.T; t += gridDim.x) {   // grid-stride over tokens: one norm atomic per block, not per token
    const u32x4 raw = raw_next, vv = v_next;
    {
        const int tn = t + (int)gridDim.x;
        if (tn < p.T) {
            raw_next = *reinterpret_cast<const u32x4*>(p.qkv + (long)tn * p.ld_in + qk_off);
            if (copy_v) v_next = *reinterpret_cast<const u32x4*>(p.qkv + (long)tn * p.ld_in + 2 * p.C + tid * 8);
        }
    }
    float x[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[2 * e] = h2_lo<FMT>(raw[e]);
        x[2 * e + 1] = h2_hi<FMT>(raw[e]);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += x[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const float mean = s * (1.0f / 64);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = x[e] - mean; q += d * d; }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
    const float rstd = rsqrtf(q * (1.0f / 64) + p.eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (x[e] - mean) * rstd * wreg[e] + breg[e];

    // RoPE
    const int pt = t % p.P;
    int py = 0, px = 0;
    if (pt >= p.patch_start) {
        const int idx = pt - p.patch_start;
        py = idx / p.gw + 1;
        px = idx - (py - 1) * p.gw + 1;
    }
    const int pos = (j < 4) ? py : px;
    const int f0 = (j & 1) * 8;
    float y[8];
    // the 8 cosines / sines of this slice are contiguous in the table: two 16-byte loads each (as 16 scalar loads the table
    // reads were 16 of the kernel's 18 vector-memory instructions per token)
    const f32x4* ct = reinterpret_cast<const f32x4*>(p.cos_t + pos * 16 + f0);
    const f32x4* st = reinterpret_cast<const f32x4*>(p.sin_t + pos * 16 + f0);
    const f32x4 c0 = ct[0], c1 = ct[1], s0 = st[0], s1 = st[1];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float partner = __shfl_xor(x[e], 2, 64);
        const float rot = (j & 2) ? partner : -partner;
        const float cs = e < 4 ? c0[e & 3] : c1[e & 3], sn = e < 4 ? s0[e & 3] : s1[e & 3];
        y[e] = x[e] * cs + rot * sn;
    }
    if (which == 0 && p.q_scale != 1.0f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] *= p.q_scale;
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_h2<FMT>(y[2 * e], y[2 * e + 1]);
    if (p.qkmax != nullptr) {   // |rounded vector|^2, this lane's 8 elements (reduced over the head's 8 lanes after the loop)
        float n2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = h2_lo<FMT>(o[e]), b2 = h2_hi<FMT>(o[e]);
            n2 += a * a + b2 * b2;
        }
        n2 += __shfl_xor(n2, 1, 64); n2 += __shfl_xor(n2, 2, 64); n2 += __shfl_xor(n2, 4, 64);
        nmax2 = fmaxf(nmax2, n2);
    }
    const int hgrp = head / p.hg, hin = head - hgrp * p.hg;
    bf16_t* dst = which ? (p.k_out + hgrp * p.kgs + (long)t * p.ldk + hin * 64) : (p.q_out + (long)t * p.ldq + head * 64);
    *reinterpret_cast<u32x4*>(dst + j * 8) = o;

    if (copy_v) {  // copy v: 1024 x 16 bit = 128 x 16 B; thread -> (head tid/8, slice tid%8)
        const int vh = tid >> 3, vg = vh / p.hg;
        *reinterpret_cast<u32x4*>(p.v_out + vg * p.vgs + (long)t * p.ldv + (vh - vg * p.hg) * 64 + (tid & 7) * 8) = vv;
    }
    }   // token loop
    if (p.qkmax != nullptr && j == 0) p.qkmax[32 + (long)(which * 16 + head) * QK_MAX_BLOCKS + blockIdx.x] = sqrtf(nmax2);
}

// qkmax[s] = max over blocks of the per-block maxima written by qknorm_rope_kernel (s = q|k * 16 + head); one workgroup
// per slot, the slot's partial maxima are contiguous.
__global__ __launch_bounds__(256) void qkmax_reduce_kernel(float* qkmax, int nblocks, int slot0) {
    __shared__ float sm[4];
    const int s = slot0 + blockIdx.x;
    float m = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) m = fmaxf(m, qkmax[32 + (long)s * QK_MAX_BLOCKS + b]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) qkmax[s] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// Largest Euclidean norm per head over the rows of a 16-bit [rows][16 * 64] key matrix -> qkmax[16 + h] (the k half of the
// static softmax bound; entries 0..15 are left alone).  For keys that did not come out of this rank's qknorm_rope_kernel:
// the gathered K rows of a view-sharded run (layers/blocks.py) -- the bound has to cover every rank's keys, and measuring
// it (one pass over K at the HBM rate, ~20 us for 44 k rows) is far tighter than any data-independent bound.
// Two rows per iteration and block; thread -> (row parity, head, 8-element slice), reduced like qknorm_rope_kernel.
template <int FMT>
__global__ __launch_bounds__(256) void krownorm_kernel(const bf16_t* k, long ld, int rows, float* qkmax) {
    const int tid = threadIdx.x;
    const int sub = tid >> 7, head = (tid >> 3) & 15, j = tid & 7;
    float nmax2 = 0.f;
    for (long r = (long)blockIdx.x * 2 + sub; r < rows; r += (long)gridDim.x * 2) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(k + r * ld + head * 64 + j * 8);
        float n2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = h2_lo<FMT>(raw[e]), b2 = h2_hi<FMT>(raw[e]);
            n2 += a * a + b2 * b2;
        }
        n2 += __shfl_xor(n2, 1, 64); n2 += __shfl_xor(n2, 2, 64); n2 += __shfl_xor(n2, 4, 64);
        nmax2 = fmaxf(nmax2, n2);
    }
    __shared__ float sm[2][16];
    if (j == 0) sm[sub][head] = nmax2;
    __syncthreads();
    if (tid < 16) qkmax[32 + (long)(16 + tid) * QK_MAX_BLOCKS + blockIdx.x] = sqrtf(fmaxf(sm[0][tid], sm[1][tid]));
}

// ---------------------------------------------------------------------------------------------
// im2row for the 14x14/14 patch-embed conv, fused with the ImageNet normalisation.
// Reference: (images - mean)/std at iggt/models/aggregator.py:206; Conv2d(3,1024,14,14) at
// iggt/layers/patch_embed.py:62,75-77.  Row (s, gy, gx) of the output holds the 588 = 3*14*14 taps in
// the conv-weight order (c, ky, kx), zero-padded to Kpad (multiple of 64) -> bf16 A operand.
struct Im2rowParams {
    const float* img;  // [S][3][H][W] in [0,1]
    bf16_t* out;       // [S*gh*gw][Kpad]
    int S, H, W, gh, gw, Kpad;
    int split;
};

template <int FMT>
__global__ __launch_bounds__(256) void im2row_patch14_kernel(const Im2rowParams p) {
    const long total = (long)p.S * p.gh * p.gw * (p.Kpad / 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kp = (int)(i % (p.Kpad / 2));
        const long row = i / (p.Kpad / 2);
        const int gx = (int)(row % p.gw);
        const int gy = (int)((row / p.gw) % p.gh);
        const int s = (int)(row / ((long)p.gw * p.gh));
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = kp * 2 + e;
            if (k < 588) {
                const int c = k / 196, r = k - c * 196, ky = r / 14, kx = r - ky * 14;
                const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
                const float stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
                const float px = p.img[(((long)s * 3 + c) * p.H + gy * 14 + ky) * p.W + gx * 14 + kx];
                v[e] = (px - mean) / stdv;
            } else {
                v[e] = 0.f;
            }
        }
        if (p.split) {   // fp16 [hi | lo | hi], row stride 3 Kpad (the x3 precision rung)
            const uint32_t hi = pack_h2<FMT>(v[0], v[1]);
            bf16_t* d = p.out + row * (3L * p.Kpad) + kp * 2;
            *reinterpret_cast<uint32_t*>(d) = hi;
            *reinterpret_cast<uint32_t*>(d + p.Kpad) = pack_h2<FMT>(v[0] - h2_lo<FMT>(hi), v[1] - h2_hi<FMT>(hi));
            *reinterpret_cast<uint32_t*>(d + 2 * p.Kpad) = hi;
        } else {
            *reinterpret_cast<uint32_t*>(p.out + row * p.Kpad + kp * 2) = pack_h2<FMT>(v[0], v[1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Broadcast "special" token rows into a [S][P][C] fp32 token matrix:
//   dst[s][row_off + r][:] = (s == 0 ? src0 : src1)[r][:]   for r < nrows
// Reference: cls/register insertion (vision_transformer.py:222-234, src0 == src1) and
// camera/register tokens with the view-0 / other-views split (aggregator.py:230-234,338-361).
struct SpecialParams {
    float* dst; long view_stride; long ldd;
    const float* src0; const float* src1;
    int S, nrows, row_off, C, first_view_is_zero;
};

__global__ __launch_bounds__(256) void write_special_tokens_kernel(const SpecialParams p) {
    const int per_view = p.nrows * (p.C / 4);
    const long total = (long)p.S * per_view;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int s = (int)(i / per_view);
        const int rem = (int)(i - (long)s * per_view);
        const int r = rem / (p.C / 4), c4 = rem - r * (p.C / 4);
        const float* src = (s == 0 && p.first_view_is_zero) ? p.src0 : p.src1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)r * p.C + c4 * 4);
        *reinterpret_cast<f32x4*>(p.dst + (long)s * p.view_stride + (long)(p.row_off + r) * p.ldd + c4 * 4) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Mean-input compensation of the 16-bit weight rounding (iggt_official_amd/precision.py).
//   y = x W^T + b  with  W = Wh + dW (Wh = round16(W)):   x W^T = x Wh^T + mu dW^T + (x - mu) dW^T
// The last term averages out over tokens like any operand rounding; the middle one is the same vector for every
// token and is what makes weight rounding the dominant error of the 16-bit trunk (oracle/precision_sim.py: token
// error 8.6e-4 -> 4.3e-4 when it is restored).  It costs a column mean of the GEMM input (over a row sample) and one
// small matrix-vector product, folded into the GEMM's bias:  b' = b + dW mu.
//
// colmean: mu[k] = mean over rows r = 0, step, 2*step, ... of x[r][k].  One block per 64 columns; thread -> (8-column
// slot, one of 128 row lanes); 16-byte loads, 4 in flight; fixed-order LDS reduction over the row lanes.
struct ColMeanParams {
    const bf16_t* x;
    long ld;
    int rows, K, step, nsamp;
    float* mu;
};

template <int FMT>
__global__ __launch_bounds__(1024) void colmean_kernel(const ColMeanParams p) {
    constexpr int RL = 128;  // row lanes: 1024 threads = 8 column slots x 128 row lanes
    __shared__ float red[RL][65];
    const int slot = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col = blockIdx.x * 64 + slot * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < p.K) {
        const bf16_t* src = p.x + col;
        const long rstride = (long)p.step * p.ld;
#pragma unroll 4
        for (int i = rl; i < p.nsamp; i += RL) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(src + i * rstride);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += h2_lo<FMT>(raw[e]);
                acc[2 * e + 1] += h2_hi<FMT>(raw[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][slot * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // fixed summation order: deterministic
#pragma unroll 8
        for (int r = 0; r < RL; r += 4) {
            s0 += red[r][threadIdx.x];
            s1 += red[r + 1][threadIdx.x];
            s2 += red[r + 2][threadIdx.x];
            s3 += red[r + 3][threadIdx.x];
        }
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < p.K) p.mu[c] = ((s0 + s1) + (s2 + s3)) / (float)p.nsamp;
    }
}

// bias_correct: out[n] = (bias ? bias[n] : 0) + sum_k dW[n][k] * mu[k];  one wave per output row.
struct BiasCorrParams {
    const bf16_t* dw;
    long ldw;
    int N, K;
    const float* mu;
    const float* bias;
    float* out;
};

template <int FMT>
__global__ __launch_bounds__(256) void bias_correct_kernel(const BiasCorrParams p) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.N) return;
    float acc = 0.f;
    for (int k = lane * 8; k < p.K; k += 512) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(p.dw + (long)n * p.ldw + k);
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(p.mu + k);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(p.mu + k + 4);
        acc += h2_lo<FMT>(raw[0]) * m0[0] + h2_hi<FMT>(raw[0]) * m0[1] + h2_lo<FMT>(raw[1]) * m0[2] +
               h2_hi<FMT>(raw[1]) * m0[3] + h2_lo<FMT>(raw[2]) * m1[0] + h2_hi<FMT>(raw[2]) * m1[1] +
               h2_lo<FMT>(raw[3]) * m1[2] + h2_hi<FMT>(raw[3]) * m1[3];
    }
    acc = wave_sum(acc);
    if (lane == 0) p.out[n] = acc + (p.bias ? p.bias[n] : 0.f);
}

// (Round 6 built both steps as ONE launch -- a workgroup per 64 columns of K, per-slice sums, agent-scope release, a ticket, the
// last arriver folds the slices -- and measured it SLOWER: 18 - 34 us per launch against 7 + 6, per-rank forward of an 8-GPU run
// 61.0 -> 67.4 ms (profiles/r06_comp_bias_ab.txt).  The agent-scope release every workgroup needs before its ticket writes back
// the XCD's L2, which holds the GEMM output that was just produced; on this chip a hand-over between workgroups of one launch
// costs more than the launch boundary it saves.  Removed.)

// ---------------------------------------------------------------------------------------------
// Tail of the DPT heads: 1x1 convolution 32 -> Cout (2..8) on an NHWC fp32 map fused with activate_head.
// Reference: scratch.output_conv2[2] (iggt/heads/dpt_head.py:121-128) + activate_head (iggt/heads/head_act.py:61-125):
// the first Cout-1 channels go through `act` (0 linear, 1 exp, 2 relu, 3 inv_log = sign(x) expm1(|x|), 4 sigmoid,
// 5 x / ||x||), the last one through `conf_act` (0 1 + exp, 1 exp, 2 sigmoid).
// HBM-bound (128 B in, <= 16 B out per pixel): 8 lanes per pixel, one float4 of the 32 channels each, 3-step xor
// reduction, lane o of the group finishes output o.  fp32 throughout (the reference runs this stage in fp32).
struct TailParams {
    const float* x;
    long ldx;
    const float* w;   // [Cout][32]
    const float* b;   // [Cout]
    float* pts;       // [npix][Cout-1]
    float* conf;      // [npix]
    long npix;
    int Cout, act, conf_act;
};

__global__ __launch_bounds__(256) void head_tail_kernel(const TailParams p) {
    const int sub = threadIdx.x & 7;
    float wreg[8][4], breg[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const bool on = o < p.Cout;
        const f32x4 wv = on ? *reinterpret_cast<const f32x4*>(p.w + o * 32 + sub * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) wreg[o][e] = wv[e];
        breg[o] = on ? p.b[o] : 0.f;
    }
    const long groups = (long)gridDim.x * 32;
    for (long pix = (long)blockIdx.x * 32 + (threadIdx.x >> 3); pix < p.npix; pix += groups) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + pix * p.ldx + sub * 4);
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float t = v[0] * wreg[o][0] + v[1] * wreg[o][1] + v[2] * wreg[o][2] + v[3] * wreg[o][3];
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 4, 64);
            acc[o] = t + breg[o];
        }
        float mine = 0.f, nrm2 = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            if (o == sub) mine = acc[o];
            if (o < p.Cout - 1) nrm2 += acc[o] * acc[o];
        }
        if (sub < p.Cout - 1) {
            float r = mine;
            if (p.act == 1) r = expf(mine);
            else if (p.act == 2) r = fmaxf(mine, 0.f);
            else if (p.act == 3) r = copysignf(expm1f(fabsf(mine)), mine);
            else if (p.act == 4) r = 1.0f / (1.0f + expf(-mine));
            else if (p.act == 5) r = mine / sqrtf(nrm2);
            p.pts[pix * (p.Cout - 1) + sub] = r;
        } else if (sub == p.Cout - 1) {
            float r;
            if (p.conf_act == 0) r = 1.0f + expf(mine);
            else if (p.conf_act == 1) r = expf(mine);
            else r = 1.0f / (1.0f + expf(-mine));
            p.conf[pix] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Window attention of the part head, head dim 32 (HAB) or 64 (OCAB), fp32 (the reference runs these stages in fp32):
//   HAB  (iggt/heads/window_sa.py:163-227 + heads/block.py:120-150): 8x8 windows, queries = keys = the window;
//   OCAB (window_sa.py:229-319): 8x8 query windows, 12x12 overlapping key/value windows (stride 8, zero padding 2:
//        out-of-image keys are ZERO VECTORS that still take part in the softmax with score = bias, as nn.Unfold
//        pads with zeros), learned relative-position bias.
// One wave per (window, head).  q comes either from an NHWC map (q_mode 0) or from a window-major [nW][64][ld] tensor (q_mode 1:
// OCAB's scrambled query windows); k, v are NHWC maps read in place (no window_partition / Unfold copies), the output goes straight
// into an NHWC map.
struct WinAttnParams {
    const float* q; long q_ld; int q_mode;
    const float* k; long k_ld;
    const float* v; long v_ld;
    float* o; long o_ld;
    const float* bias;   // [heads][ow*ow][64] or null
    int b, h, w, heads, ow, pad;
    float scale;
};

// Round 6: on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation, the arithmetic of the fma
// chains it replaces), same dataflow as attn_f32_kernel (smallops.hip).  The first version ran one wave per (window, head) with
// lane = query on the vector ALU, key / value rows as scalar loads: 18.5 TFLOP/s, 5.2 + 5.8 ms per 32-view forward at 532^2
// (profiles/r06_bench_n1_b.json) -- two dependent scalar-load round trips and a max / vote / branch per key.  Here a wave still
// TWO waves own one (window, head), 32 queries each: per tile of 32 keys the pair stages K (padded rows) and V in the head's LDS
// region (zero rows for the padded positions of OCAB's 12 x 12 windows and past the last key), S^T[key][query] = K . Q^T is
// D / 2 MFMAs per half with Q held in registers, the bias is added in the accumulator layout, the online softmax is lane-local
// plus one exchange with lane ^ 32, and the exponentiated accumulators are the B operand of O^T = V^T . P^T.  The output rows go
// through LDS so that every lane stores 16 contiguous bytes.
template <int D>
__global__ __launch_bounds__(512) void window_attn_kernel(const WinAttnParams p) {
    constexpr int KP = D + 1;                              // padded K row: the A-operand read (lane -> key row) is conflict-free
    constexpr int OP = D + 4;                              // output staging row
    constexpr int HEAD_FLOATS = 2 * 32 * OP;               // >= 32 * KP + 32 * D: K tile + V tile, later the two halves' outputs
    extern __shared__ __attribute__((aligned(16))) float smw[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int head = wv >> 1, qh = wv & 1;                 // two waves per (window, head): 32 queries each, one K / V tile
    float* Ks = smw + head * HEAD_FLOATS;
    float* Vs = Ks + 32 * KP;
    const int r = lane & 31, hh = lane >> 5;
    const int nwx = p.w >> 3, nwy = p.h >> 3;
    const int win = blockIdx.x;
    const int bi = win / (nwy * nwx), wrem = win - bi * (nwy * nwx);
    const int wy = wrem / nwx, wx = wrem - wy * nwx;
    const float sl2 = p.scale * 1.44269504088896340736f;
    // B operand of MFMA j: Q[query qh * 32 + r][2 j + hh], pre-scaled by scale * log2 e
    float qreg[D / 2];
    {
        const int qi = qh * 32 + r;
        const long pix = ((long)bi * p.h + wy * 8 + (qi >> 3)) * p.w + wx * 8 + (qi & 7);
        const float* qp = p.q_mode ? p.q + ((long)win * 64 + qi) * p.q_ld + head * D : p.q + pix * p.q_ld + head * D;
#pragma unroll
        for (int c = 0; c < D / 4; ++c) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 4 * c);
            qreg[2 * c] = (hh ? t[1] : t[0]) * sl2;
            qreg[2 * c + 1] = (hh ? t[3] : t[2]) * sl2;
        }
    }
    f32x16 o[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int nk = p.ow * p.ow;
    const int lp = qh * 64 + lane;                         // lane of the head's wave pair
    for (int t0 = 0; t0 < nk; t0 += 32) {
        __syncthreads();                                   // (uniform: every wave of the workgroup walks the same key tiles)
#pragma unroll
        for (int it = 0; it < D / 16; ++it) {              // 32 keys x D / 4 float4 pieces over the pair's 128 lanes
            const int i = it * 128 + lp;
            const int kr = i / (D / 4), c4 = i - kr * (D / 4);
            const int j = t0 + kr;
            const int ky = j / p.ow, kx = j - ky * p.ow;
            const int py = wy * 8 - p.pad + ky, px = wx * 8 - p.pad + kx;
            const bool inb = j < nk && py >= 0 && py < p.h && px >= 0 && px < p.w;
            f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (inb) {
                const long kpix = ((long)bi * p.h + py) * p.w + px;
                kk = *reinterpret_cast<const f32x4*>(p.k + kpix * p.k_ld + head * D + 4 * c4);
                vv = *reinterpret_cast<const f32x4*>(p.v + kpix * p.v_ld + head * D + 4 * c4);
            }
            Ks[kr * KP + 4 * c4] = kk[0]; Ks[kr * KP + 4 * c4 + 1] = kk[1];
            Ks[kr * KP + 4 * c4 + 2] = kk[2]; Ks[kr * KP + 4 * c4 + 3] = kk[3];
            *reinterpret_cast<f32x4*>(Vs + kr * D + 4 * c4) = vv;
        }
        __syncthreads();
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int j = 0; j < D / 2; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[r * KP + 2 * j + hh], qreg[j], s, 0, 0, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = t0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            float v = s[e];
            if (p.bias && key < nk) v += p.bias[((long)head * nk + key) * 64 + qh * 32 + r] * 1.44269504088896340736f;
            v = key < nk ? v : -INFINITY;
            s[e] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);     // first tile: exp2(-inf) = 0
        m = mn;
        float ls = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = __builtin_amdgcn_exp2f(s[e] - mn);
            ls += s[e];
        }
        l = l * alpha + ls;
#pragma unroll
        for (int i = 0; i < D / 32; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = (e & 3) + 8 * (e >> 2) + 4 * hh;
                o[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * D + i * 32 + r], s[e], o[i], 0, 0, 0);
            }
        }
    }
    // output: O^T fragments -> this wave's half of the head's LDS region as [32 queries][D + 4] -> 16-byte row pieces to the map
    __syncthreads();
    float* Os = Ks + qh * (32 * OP);
    {
        const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) Os[r * OP + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh] = o[i][e] * inv;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < D / 8; ++it) {                   // 32 queries x D / 4 float4 pieces over 64 lanes
        const int i = it * 64 + lane;
        const int qr = i / (D / 4), c4 = i - qr * (D / 4);
        const int qi = qh * 32 + qr;
        const long pix = ((long)bi * p.h + wy * 8 + (qi >> 3)) * p.w + wx * 8 + (qi & 7);
        *reinterpret_cast<f32x4*>(p.o + pix * p.o_ld + head * D + 4 * c4) = *reinterpret_cast<const f32x4*>(Os + qr * OP + 4 * c4);
    }
}

}  // namespace

extern "C" int iggt_layernorm_f32(const float* x0, long ld0, const float* x1, long ld1, const float* w,
                                  const float* b, void* out, long ldo, int out_type, int rows, int C,
                                  float eps, int rows_in, int rows_stride, int row_off, int orows_stride, int orow_off,
                                  void* stream) {
    if (rows <= 0 || out_type < 0 || out_type > 3) return -1;
    if ((ld0 % 4) || (x1 && (ld1 % 4)) || (ldo % 4)) return -2;
    if (out_type == 3 && (ldo < 3L * C || C < 256)) return -2;   // [hi | lo | hi], segments C apart
    const int out_is_f32 = out_type == 1;
    LnParams p;
    p.f16 = out_type >= 2;
    p.split_seg = out_type == 3 ? C : 0;
    p.x0 = x0; p.x1 = x1; p.ld0 = ld0; p.ld1 = ld1; p.w = w; p.b = b;
    p.out = out_is_f32 ? nullptr : (bf16_t*)out;
    p.out_f32 = out_is_f32 ? (float*)out : nullptr;
    p.ldo = ldo; p.rows = rows; p.eps = eps;
    p.rows_in = rows_in; p.rows_stride = rows_stride; p.row_off = row_off;
    p.orows_stride = orows_stride; p.orow_off = orow_off;
    const dim3 grid((rows + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (C == 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, st, p);
    else if (C == 2048) hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, st, p);
    else if (C == 256 && !x1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, p);
    else if (C == 512 && !x1) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, p);
    else if (C == 128 && !x1 && rows_in == 0)
        hipLaunchKernelGGL(layernorm128_kernel, dim3((rows + 7) / 8), block, 0, st, p);
    else return -3;
    IGGT_CHECK_LAUNCH();
    return 0;
}

static int qknorm_rope_h16(int fmt, const void* qkv, long ld_in, void* q_out, long ldq, void* k_out, long ldk,
                           void* v_out, long ldv, const float* qw, const float* qb, const float* kw, const float* kb,
                           const float* cos_t, const float* sin_t, int T, int P, int gw, int patch_start, float eps,
                           int heads_per_group, long k_group_stride, long v_group_stride, float q_scale, float* qkmax,
                           void* stream) {
    if (T <= 0 || P <= 0) return -1;
    if (!(q_scale > 0.f)) return -4;
    if ((ld_in % 8) || (ldq % 8) || (ldk % 8) || (v_out && (ldv % 8))) return -2;
    if (((uintptr_t)cos_t | (uintptr_t)sin_t) % 16) return -2;   // the tables are read as 16-byte vectors
    QkParams p;
    p.qkv = (const bf16_t*)qkv; p.ld_in = ld_in;
    p.q_out = (bf16_t*)q_out; p.ldq = ldq; p.k_out = (bf16_t*)k_out; p.ldk = ldk;
    p.v_out = (bf16_t*)v_out; p.ldv = ldv;
    p.qw = qw; p.qb = qb; p.kw = kw; p.kb = kb; p.cos_t = cos_t; p.sin_t = sin_t;
    p.T = T; p.P = P; p.gw = gw; p.patch_start = patch_start; p.eps = eps; p.C = 1024;
    if (heads_per_group <= 0 || heads_per_group >= 16) {
        p.hg = 16; p.kgs = 0; p.vgs = 0;
    } else {
        if ((16 % heads_per_group) || (k_group_stride % 8) || (v_group_stride % 8)) return -3;
        p.hg = heads_per_group; p.kgs = k_group_stride; p.vgs = v_group_stride;
    }
    p.q_scale = q_scale; p.qkmax = qkmax;
    const int grid = T < QK_MAX_BLOCKS ? T : QK_MAX_BLOCKS;   // each block strides over T / grid tokens
    if (fmt == FMT_F16) hipLaunchKernelGGL(qknorm_rope_kernel<FMT_F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(qknorm_rope_kernel<FMT_BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    if (qkmax != nullptr) {
        hipLaunchKernelGGL(qkmax_reduce_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, qkmax, grid, 0);
        IGGT_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int iggt_qknorm_rope_bf16(const void* qkv, long ld_in, void* q_out, long ldq, void* k_out, long ldk,
                                     void* v_out, long ldv, const float* qw, const float* qb, const float* kw,
                                     const float* kb, const float* cos_t, const float* sin_t, int T, int P,
                                     int gw, int patch_start, float eps, int heads_per_group, long k_group_stride,
                                     long v_group_stride, float q_scale, float* qkmax, void* stream) {
    return qknorm_rope_h16(FMT_BF16, qkv, ld_in, q_out, ldq, k_out, ldk, v_out, ldv, qw, qb, kw, kb, cos_t, sin_t, T, P,
                           gw, patch_start, eps, heads_per_group, k_group_stride, v_group_stride, q_scale, qkmax, stream);
}

extern "C" int iggt_qknorm_rope_f16(const void* qkv, long ld_in, void* q_out, long ldq, void* k_out, long ldk,
                                    void* v_out, long ldv, const float* qw, const float* qb, const float* kw,
                                    const float* kb, const float* cos_t, const float* sin_t, int T, int P,
                                    int gw, int patch_start, float eps, int heads_per_group, long k_group_stride,
                                    long v_group_stride, float q_scale, float* qkmax, void* stream) {
    return qknorm_rope_h16(FMT_F16, qkv, ld_in, q_out, ldq, k_out, ldk, v_out, ldv, qw, qb, kw, kb, cos_t, sin_t, T, P,
                           gw, patch_start, eps, heads_per_group, k_group_stride, v_group_stride, q_scale, qkmax, stream);
}

static int k_rownorm_max(int fmt, const void* k, long ldk, int rows, float* qkmax, void* stream) {
    if (k == nullptr || qkmax == nullptr || rows <= 0 || (ldk % 8) || ((uintptr_t)k % 16)) return -1;
    const int grid = rows / 2 + 1 < 2048 ? rows / 2 + 1 : 2048;
    if (fmt == FMT_F16) hipLaunchKernelGGL(krownorm_kernel<FMT_F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)k, ldk, rows, qkmax);
    else hipLaunchKernelGGL(krownorm_kernel<FMT_BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)k, ldk, rows, qkmax);
    IGGT_CHECK_LAUNCH();
    hipLaunchKernelGGL(qkmax_reduce_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, qkmax, grid, 16);
    IGGT_CHECK_LAUNCH();
    return 0;
}
extern "C" int iggt_k_rownorm_max_bf16(const void* k, long ldk, int rows, float* qkmax, void* stream) {
    return k_rownorm_max(FMT_BF16, k, ldk, rows, qkmax, stream);
}
extern "C" int iggt_k_rownorm_max_f16(const void* k, long ldk, int rows, float* qkmax, void* stream) {
    return k_rownorm_max(FMT_F16, k, ldk, rows, qkmax, stream);
}

extern "C" int iggt_im2row_patch14(const float* img, void* out, int out_f16, int S, int H, int W, int Kpad,
                                   void* stream) {
    if (S <= 0 || (H % 14) || (W % 14) || Kpad < 588 || (Kpad % 64) || out_f16 < 0 || out_f16 > 2) return -1;
    Im2rowParams p;
    p.img = img; p.out = (bf16_t*)out; p.S = S; p.H = H; p.W = W; p.gh = H / 14; p.gw = W / 14; p.Kpad = Kpad;
    p.split = out_f16 == 2;   // fp16 [hi | lo | hi] rows of 3 Kpad (round 5)
    const long total = (long)S * p.gh * p.gw * (Kpad / 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (out_f16) hipLaunchKernelGGL(im2row_patch14_kernel<FMT_F16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(im2row_patch14_kernel<FMT_BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_write_special_tokens(float* dst, long view_stride, long ldd, const float* src0,
                                         const float* src1, int S, int nrows, int row_off, int C,
                                         int first_view_is_zero, void* stream) {
    if (S <= 0 || nrows <= 0 || (C % 4) || (ldd % 4) || (view_stride % 4)) return -1;
    SpecialParams p;
    p.dst = dst; p.view_stride = view_stride; p.ldd = ldd; p.src0 = src0; p.src1 = src1;
    p.S = S; p.nrows = nrows; p.row_off = row_off; p.C = C; p.first_view_is_zero = first_view_is_zero;
    const long total = (long)S * nrows * (C / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(write_special_tokens_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

// Debug telemetry for the fp16 operand format: counts the entries of a 16-bit matrix that sit at the saturation value
// (|x| = 65504: a clamped store, common.h pack_h2) or are not finite.  Run by the host model after every 16-bit-producing
// kernel when IGGT_DEBUG_SATURATION=1 (iggt_official_amd/precision.py); no cost otherwise.
__global__ __launch_bounds__(256) void count_saturated_kernel(const uint16_t* x, long ld, int rows, int cols, int f16,
                                                              unsigned long long* counter) {
    unsigned int local = 0;
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cols;
        const uint16_t v = x[r * ld + (i - r * cols)] & 0x7fffu;
        local += f16 ? (v >= 0x7bffu) : (v >= 0x7f80u);   // fp16: max finite 0x7bff; bf16: inf / nan only (no clamp needed)
    }
    local = (unsigned int)wave_sum((float)local);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(counter, (unsigned long long)local);
}

extern "C" int iggt_count_saturated_h16(const void* x, long ld, int rows, int cols, int f16, void* counter, void* stream) {
    if (rows <= 0 || cols <= 0 || counter == nullptr) return -1;
    const long total = (long)rows * cols;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(count_saturated_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, ld, rows, cols,
                       f16, (unsigned long long*)counter);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_colmean_h16(const void* x, long ld, int rows, int K, int row_step, int f16, float* mu,
                                void* stream) {
    if (rows <= 0 || K <= 0 || (K % 8) || (ld % 8) || row_step <= 0) return -1;
    ColMeanParams p;
    p.x = (const bf16_t*)x; p.ld = ld; p.rows = rows; p.K = K; p.step = row_step;
    p.nsamp = (rows + row_step - 1) / row_step;
    p.mu = mu;
    const dim3 grid((K + 63) / 64), block(1024);
    if (f16) hipLaunchKernelGGL(colmean_kernel<FMT_F16>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(colmean_kernel<FMT_BF16>, grid, block, 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_bias_correct_h16(const void* dw, long ldw, int N, int K, const float* mu, const float* bias,
                                     float* out, int f16, void* stream) {
    if (N <= 0 || K <= 0 || (K % 8) || (ldw % 8)) return -1;
    BiasCorrParams p;
    p.dw = (const bf16_t*)dw; p.ldw = ldw; p.N = N; p.K = K; p.mu = mu; p.bias = bias; p.out = out;
    const dim3 grid((N + 3) / 4), block(256);
    if (f16) hipLaunchKernelGGL(bias_correct_kernel<FMT_F16>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(bias_correct_kernel<FMT_BF16>, grid, block, 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_head_tail_f32(const float* x, long ldx, const float* w, const float* b, float* pts, float* conf,
                                  long npix, int Cout, int act, int conf_act, void* stream) {
    if (npix <= 0 || Cout < 2 || Cout > 8 || (ldx % 4) != 0 || ldx < 32) return -1;
    if (act < 0 || act > 5 || conf_act < 0 || conf_act > 2) return -2;
    TailParams p;
    p.x = x; p.ldx = ldx; p.w = w; p.b = b; p.pts = pts; p.conf = conf; p.npix = npix;
    p.Cout = Cout; p.act = act; p.conf_act = conf_act;
    long blocks = (npix + 31) / 32;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(head_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_window_attn_f32(const float* q, long q_ld, int q_mode, const float* k, long k_ld, const float* v,
                                    long v_ld, float* o, long o_ld, const float* bias, int b, int h, int w, int heads,
                                    int head_dim, int ow, int pad, float scale, void* stream) {
    if (b <= 0 || h <= 0 || w <= 0 || (h % 8) || (w % 8) || heads < 1 || heads > 4 || ow < 1 || pad < 0) return -1;
    if (head_dim != 32 && head_dim != 64) return -3;
    if ((q_ld % 4) || (k_ld % 4) || (v_ld % 4) || (o_ld % 4)) return -2;
    WinAttnParams p;
    p.q = q; p.q_ld = q_ld; p.q_mode = q_mode; p.k = k; p.k_ld = k_ld; p.v = v; p.v_ld = v_ld; p.o = o; p.o_ld = o_ld;
    p.bias = bias; p.b = b; p.h = h; p.w = w; p.heads = heads; p.ow = ow; p.pad = pad; p.scale = scale;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16) != 0) return -2;     // 16-byte row pieces
    const long nwin = (long)b * (h / 8) * (w / 8);
    const int lds = heads * 2 * 32 * (head_dim + 4) * 4;       // per head: K tile (padded rows) + V tile, later the output rows
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)window_attn_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           4 * 2 * 32 * 68 * 4);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (head_dim == 32)
        hipLaunchKernelGGL(window_attn_kernel<32>, dim3((unsigned)nwin), dim3(128 * heads), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(window_attn_kernel<64>, dim3((unsigned)nwin), dim3(128 * heads), lds, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}
