// "x3" precision rung of the trunk blocks (round 5): fp16 hi + lo operand PAIRS, three MFMA passes per product.
//
// Why: on heavy-tailed but well-conditioned checkpoints (log-normal LayerNorm / q-k-norm scales of sigma 0.75-1,
// iggt_official_amd/synthetic.py "trained_like") the reference's fp32 arithmetic sits 1e-5 from an fp64 evaluation while
// single fp16 operands land 3e-3 .. 2e-2 away -- and no single rounding site carries that: leave-one-in ablation
// (probes/precision_groups.py, profiles/r05_precision_groups.txt) shows weights, LayerNorm outputs, the qkv GEMM output, the
// post-norm q / k, the attention output and the MLP hidden activation EACH contribute 2e-3 .. 1e-2, in quadrature; even the
// softmax numerators alone keep the total at 6e-4 .. 7.5e-4.  So an escalated block splits EVERY MFMA operand:
//     x = hi + lo,  hi = fp16(x),  lo = fp16(x - hi)            (22 significant bits; lo may be subnormal: absolute 6e-8)
//     a . b  ~=  a_hi . b_hi + a_lo . b_hi + a_hi . b_lo        (the dropped lo . lo term is 2^-22 relative)
// at 3x the fp16 MFMA count -- 5x cheaper than the fp32 matrix pipe.  GEMMs need no new kernel: the three passes are ONE
// fp16 GEMM over the concatenated K axis, A' = [A_hi | A_lo | A_hi] against W' = [W_hi | W_hi | W_lo]
// (layers/blocks.py Block._forward_x3); this file holds what produces the split operands and the attention that consumes them:
//   * iggt_qkv_split_f16       qkv fp32 (the GEMM's fp32 output: no 16-bit rounding before the q/k LayerNorm) -> optional per-head
//                              LayerNorm(64) + RoPE (reference attention.py:54-58, rope.py:119-188), q pre-multiplied by
//                              scale * log2 e, then q, k, v as hi / lo pairs
//   * iggt_split3_f16          fp32 matrix -> optional exact-erf GELU (mlp.py:34) -> [hi | lo | hi] (the A' of the next GEMM)
//   * iggt_flash_attn_x3_f16_d64   softmax(q k^T) v with S = 3 passes, online max in fp32, P split as well, O = 3 passes; writes the
//                              output as [hi | lo | hi] (the A' of the proj GEMM).  Reference: attention.py:60-66.
// (LayerNorm and the patch-embed im2row write [hi | lo | hi] through a new output kind of their existing entry points.)
// Same LDS images, swapped-QK^T / in-register-P scheme and DMA staging as attention_v3.hip (128-row workgroups, 64-key tiles,
// online max): this is the precision path, scheduled by the compiler; per 32 x 64 score block 48 MFMAs instead of 16.
#include <stdlib.h>

#include "attention_common.h"
#include "../../include/iggt_hip.h"

using namespace iggt_attn;

namespace {

struct X3Params {
    const bf16_t* q;  const bf16_t* ql;
    const bf16_t* k;  const bf16_t* kl;
    const bf16_t* v;  const bf16_t* vl;
    bf16_t* o;
    long o_seg;    // > 0: output written as hi at +0, lo at +o_seg, hi again at +2 o_seg; 0: one fp16 value
    int B, H, Nq, Nk;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
    int qtiles;
};

constexpr int X3_TILE_BYTES = 4 * K_BYTES;   // K_hi, K_lo, V_hi, V_lo images of one 64-key tile: 32 KiB
constexpr float X3_P_SHIFT = 8.0f;           // numerators 2^(s - m + 8): small ones stay in fp16's normal range
constexpr float X3_DEFER = 4.0f;

// SAT: clamp to fp16's finite range first (activations; the softmax numerators are <= 2^12 by construction)
struct HiLo {
    uint32_t hi, lo;
};
template <bool SAT = true>
IGGT_DEVINL HiLo split_pack(float a, float b) {
    HiLo r;
    r.hi = pack_h2<FMT_F16, SAT>(a, b);
    r.lo = pack_h2<FMT_F16, SAT>(a - h2_lo<FMT_F16>(r.hi), b - h2_hi<FMT_F16>(r.hi));
    return r;
}

// PSPLIT = false (IGGT_X3_P_SINGLE=1, measurement only): the softmax numerators stay single fp16 -- O in two passes, 40 instead of
// 48 MFMAs per score block.  The numerators are the smallest of the rounding sites (profiles/r05_precision_groups.txt: 5.8e-4 alone
// in the CPU simulation); DESIGN.md section 2 has what the dose fixtures measure with it.  The shipped rung splits them.
template <bool PSPLIT>
__global__ __launch_bounds__(256, 2) void flash_attn_x3_kernel(const X3Params p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * X3_TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = work % p.qtiles, bh = work / p.qtiles;
    const int h = bh % p.H, b = bh / p.H;
    const long qoff = (long)b * p.q_bs + h * 64, koff0 = (long)b * p.k_bs + h * 64, voff0 = (long)b * p.v_bs + h * 64;
    const int q_base = qt * 128 + wave * 32;

    bf16x8 qh[4], ql[4];
    {
        int qr = q_base + frow;
        qr = qr < p.Nq ? qr : p.Nq - 1;
        const long o_ = qoff + (long)qr * p.q_rs + 8 * fhalf;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            qh[kc] = *reinterpret_cast<const bf16x8*>(p.q + o_ + 16 * kc);
            ql[kc] = *reinterpret_cast<const bf16x8*>(p.ql + o_ + 16 * kc);
        }
    }

    // LDS-DMA staging as in attention_v3.hip: a 64-key image is 8 chunks of 1 KiB (8 rows x 128 B); wave w moves chunks 2w, 2w + 1
    // of each of the four images; the row-image swizzles are applied to the SOURCE piece.
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int c_row = lane >> 3, c_pos = lane & 7;
    long ksrc[2], vsrc[2];   // element offsets of this lane's pieces in tile 0
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 8 + c_row;
        ksrc[i] = koff0 + (long)r * p.k_rs + (c_pos ^ ((r >> 1) & 7)) * 8;
        vsrc[i] = voff0 + (long)r * p.v_rs + ((((c_pos >> 1) ^ (r & 2)) << 1) | (c_pos & 1)) * 8;
    }
    auto dma = [&](int t, int buf) {
        char* base = smem + buf * X3_TILE_BYTES;
        const int kv0 = t * KV_TILE;
        const bool ragged = kv0 + KV_TILE > p.Nk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int chunk = (2 * wave + i) * 1024;
            long ko = ksrc[i] + (long)kv0 * p.k_rs, vo = vsrc[i] + (long)kv0 * p.v_rs;
            if (ragged) {   // rows past the end are clamped to the last valid row (their scores are masked to -inf)
                const int r = (2 * wave + i) * 8 + c_row;
                const int over = kv0 + r - (p.Nk - 1);
                if (over > 0) {
                    ko -= (long)over * p.k_rs;
                    vo -= (long)over * p.v_rs;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t*)(p.k + ko), (lptr_t*)(base + chunk), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)(p.kl + ko), (lptr_t*)(base + K_BYTES + chunk), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)(p.v + vo), (lptr_t*)(base + 2 * K_BYTES + chunk), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)(p.vl + vo), (lptr_t*)(base + 3 * K_BYTES + chunk), 16, 0, 0);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dh][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int tr_i = lane & 15, tr_g = (lane >> 4) & 1;
    int koff[4], voff[2];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) koff[kc] = frow * 128 + ((((2 * kc + fhalf) ^ (frow >> 1)) & 7) << 4);
    {
        const int vr = 4 * fhalf + (tr_i >> 2);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) voff[dh] = vr * 128 + ((((dh * 2 + tr_g) ^ (vr & 2))) << 5) + 8 * (tr_i & 3);
    }
    const int NT = (p.Nk + KV_TILE - 1) / KV_TILE;

    dma(0, 0);
    __syncthreads();   // vmcnt(0) + barrier: tile 0 resident
    for (int t = 0; t < NT; ++t) {
        if (t + 1 < NT) dma(t + 1, (t + 1) & 1);
        const char* sK = smem + (t & 1) * X3_TILE_BYTES;
        const char* sKl = sK + K_BYTES;
        const char* sV = sK + 2 * K_BYTES;
        const char* sVl = sK + 3 * K_BYTES;

        // ---- S = K Q^T (swapped: a lane owns one query column), three passes, the small terms first
        f32x16 s[2];
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh) {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 kl_ = *reinterpret_cast<const bf16x8*>(sKl + kvh * 4096 + koff[kc]);
                acc = mfma32h<FMT_F16>(kl_, qh[kc], acc);
            }
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 kh_ = *reinterpret_cast<const bf16x8*>(sK + kvh * 4096 + koff[kc]);
                acc = mfma32h<FMT_F16>(kh_, ql[kc], acc);
                acc = mfma32h<FMT_F16>(kh_, qh[kc], acc);
            }
            s[kvh] = acc;
        }
        if ((t + 1) * KV_TILE > p.Nk) {
            const int kv0 = t * KV_TILE + 4 * fhalf;
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + kvh * 32 + (r & 3) + 8 * (r >> 2);
                    if (kv >= p.Nk) s[kvh][r] = -INFINITY;
                }
        }
        // ---- online max (fp32; deferred: m only advances when some row's tile maximum exceeds it by 2^X3_DEFER)
        {
            float mx = s[0][0];
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kvh][r]);
            const uint32_t bits = __builtin_bit_cast(uint32_t, mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
            if (__any(mx > m_run + X3_DEFER)) {
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dh][r] *= alpha;
            }
        }
        // ---- numerators, fp32 row sum, hi / lo fragments
        bf16x8 ph[2][2], pl[2][2];
        {
            const float m = m_run - X3_P_SHIFT;
            float ls = 0.f;
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[kvh][r] = __builtin_amdgcn_exp2f(s[kvh][r] - m);
                    ls += s[kvh][r];
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    u32x4 wh, wl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const HiLo w = split_pack<false>(s[kvh][8 * cc + 2 * j], s[kvh][8 * cc + 2 * j + 1]);
                        wh[j] = w.hi;
                        wl[j] = w.lo;
                    }
                    ph[kvh][cc] = __builtin_bit_cast(bf16x8, wh);
                    pl[kvh][cc] = __builtin_bit_cast(bf16x8, wl);
                }
            }
            l_run += ls;
        }
        // ---- O += V^T P, three passes
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    typedef __attribute__((address_space(3))) short4v lds_s4;
                    typedef short short8v __attribute__((ext_vector_type(8)));
                    const int off = (kvh * 32 + 16 * cc) * 128 + voff[dh];
                    const short4v l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(sVl + off));
                    const short4v l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(sVl + off + 8 * 128));
                    const short4v h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(sV + off));
                    const short4v h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(sV + off + 8 * 128));
                    const bf16x8 vl_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                    const bf16x8 vh_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                    o[dh] = mfma32h<FMT_F16>(vl_, ph[kvh][cc], o[dh]);
                    if constexpr (PSPLIT) o[dh] = mfma32h<FMT_F16>(vh_, pl[kvh][cc], o[dh]);
                    o[dh] = mfma32h<FMT_F16>(vh_, ph[kvh][cc], o[dh]);
                }
        __syncthreads();   // everyone done with buffer t & 1; the DMA of tile t + 1 has landed
    }

    const int qr = q_base + frow;
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l;
    if (qr < p.Nq) {
        bf16_t* dst = p.o + (long)b * p.o_bs + (long)qr * p.o_rs + h * 64 + 4 * fhalf;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 wh, wl;
                const HiLo w0 = split_pack(o[dh][4 * g + 0] * inv, o[dh][4 * g + 1] * inv);
                const HiLo w1 = split_pack(o[dh][4 * g + 2] * inv, o[dh][4 * g + 3] * inv);
                wh[0] = w0.hi; wl[0] = w0.lo; wh[1] = w1.hi; wl[1] = w1.lo;
                bf16_t* d = dst + dh * 32 + 8 * g;
                *reinterpret_cast<u32x2*>(d) = wh;
                if (p.o_seg > 0) {
                    *reinterpret_cast<u32x2*>(d + p.o_seg) = wl;
                    *reinterpret_cast<u32x2*>(d + 2 * p.o_seg) = wh;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// qkv fp32 [T][3 C] -> q, k, v as fp16 hi / lo pairs; optional per-head LayerNorm(64) on q and k + 2-D RoPE (same arithmetic and
// thread mapping as qknorm_rope_kernel in elementwise.hip: one block per token, thread -> (q | k, head, 8-element slice)).
struct QkvSplitParams {
    const float* qkv; long ld_in;
    bf16_t* q_out; long ldq, q_lo;      // lo part at q_out + q_lo (elements)
    bf16_t* k_out; long ldk, k_lo;
    bf16_t* v_out; long ldv, v_lo;
    const float* qw; const float* qb; const float* kw; const float* kb;   // [64] each; qw == nullptr: no q/k LayerNorm
    const float* cos_t; const float* sin_t;                               // [npos][16]; nullptr: no RoPE
    int T, P, gw, patch_start;
    float eps, q_scale;
    int C;
};

IGGT_DEVINL void store_split8(bf16_t* hi_dst, long lo_off, const float (&y)[8]) {
    u32x4 wh, wl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const HiLo w = split_pack(y[2 * e], y[2 * e + 1]);
        wh[e] = w.hi;
        wl[e] = w.lo;
    }
    *reinterpret_cast<u32x4*>(hi_dst) = wh;
    *reinterpret_cast<u32x4*>(hi_dst + lo_off) = wl;
}

__global__ __launch_bounds__(256) void qkv_split_kernel(const QkvSplitParams p) {
    const int tid = threadIdx.x;
    const int which = tid >> 7, head = (tid >> 3) & 15, j = tid & 7;
    float wreg[8], breg[8];
    const bool norm = p.qw != nullptr;
    if (norm) {
        const float* w = which ? p.kw : p.qw;
        const float* bb = which ? p.kb : p.qb;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wreg[e] = w[j * 8 + e];
            breg[e] = bb[j * 8 + e];
        }
    }
    const long qk_off = (long)which * p.C + head * 64 + j * 8;
    for (int t = blockIdx.x; t < p.T; t += gridDim.x) {
        const float* src = p.qkv + (long)t * p.ld_in;
        float x[8];
        {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + qk_off), c = *reinterpret_cast<const f32x4*>(src + qk_off + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = a[e];
                x[4 + e] = c[e];
            }
        }
        if (norm) {
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
        }
        float y[8];
        if (p.cos_t != nullptr) {
            const int pt = t % p.P;
            int py = 0, px = 0;
            if (pt >= p.patch_start) {
                const int idx = pt - p.patch_start;
                py = idx / p.gw + 1;
                px = idx - (py - 1) * p.gw + 1;
            }
            const int pos = (j < 4) ? py : px;
            const int f0 = (j & 1) * 8;
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
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = x[e];
        }
        if (which == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] *= p.q_scale;
            store_split8(p.q_out + (long)t * p.ldq + head * 64 + j * 8, p.q_lo, y);
        } else {
            store_split8(p.k_out + (long)t * p.ldk + head * 64 + j * 8, p.k_lo, y);
        }
        if (tid < 128) {   // v: 1024 values = 128 threads x 8
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + 2 * p.C + tid * 8);
            const f32x4 c = *reinterpret_cast<const f32x4*>(src + 2 * p.C + tid * 8 + 4);
            const float vv[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            store_split8(p.v_out + (long)t * p.ldv + tid * 8, p.v_lo, vv);
        }
    }
}

// fp32 [rows][N] -> optional exact GELU -> fp16 [rows][3 N] = [hi | lo | hi]
__global__ __launch_bounds__(256) void split3_kernel(const float* x, long ldx, bf16_t* out, long ldo, int rows, int N, int act) {
    const int n4 = N / 4;
    const long total = (long)rows * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / n4;
        const int c = (int)(i - r * n4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
        }
        u32x2 wh, wl;
        const HiLo w0 = split_pack(v[0], v[1]), w1 = split_pack(v[2], v[3]);
        wh[0] = w0.hi; wl[0] = w0.lo; wh[1] = w1.hi; wl[1] = w1.lo;
        bf16_t* d = out + r * ldo + c;
        *reinterpret_cast<u32x2*>(d) = wh;
        *reinterpret_cast<u32x2*>(d + N) = wl;
        *reinterpret_cast<u32x2*>(d + 2 * N) = wh;
    }
}

}  // namespace

extern "C" int iggt_flash_attn_x3_f16_d64(const void* q, const void* q_lo, const void* k, const void* k_lo, const void* v,
                                          const void* v_lo, void* o, long o_seg, int B, int H, int Nq, int Nk, long q_bs,
                                          long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs, long o_rs,
                                          void* stream) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return -1;
    if (!q || !q_lo || !k || !k_lo || !v || !v_lo || !o) return -5;
    if ((q_rs % 8) || (k_rs % 8) || (v_rs % 8) || (o_rs % 4) || (q_bs % 8) || (k_bs % 8) || (v_bs % 8) || (o_bs % 4)) return -2;
    if (o_seg < 0 || (o_seg % 4)) return -2;
    const uintptr_t align = (uintptr_t)q | (uintptr_t)q_lo | (uintptr_t)k | (uintptr_t)k_lo | (uintptr_t)v | (uintptr_t)v_lo;
    if ((align % 16) || ((uintptr_t)o % 8)) return -2;
    X3Params p;
    p.q = (const bf16_t*)q; p.ql = (const bf16_t*)q_lo; p.k = (const bf16_t*)k; p.kl = (const bf16_t*)k_lo;
    p.v = (const bf16_t*)v; p.vl = (const bf16_t*)v_lo; p.o = (bf16_t*)o; p.o_seg = o_seg;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.qtiles = (Nq + 127) / 128;
    const long nwg = (long)B * H * p.qtiles;
    if (nwg > 0x7fffffffL) return -1;
    static int p_single = -1;
    if (p_single < 0) {
        const char* e = getenv("IGGT_X3_P_SINGLE");
        p_single = (e && e[0] == '1') ? 1 : 0;
    }
    if (p_single) hipLaunchKernelGGL(flash_attn_x3_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(flash_attn_x3_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_qkv_split_f16(const float* qkv, long ld_in, void* q_out, long ldq, long q_lo, void* k_out, long ldk, long k_lo,
                                  void* v_out, long ldv, long v_lo, const float* qw, const float* qb, const float* kw,
                                  const float* kb, const float* cos_t, const float* sin_t, int T, int P, int gw,
                                  int patch_start, float eps, float q_scale, void* stream) {
    if (T <= 0 || !qkv || !q_out || !k_out || !v_out) return -1;
    if (!(q_scale > 0.f)) return -4;
    if ((ld_in % 4) || (ldq % 8) || (ldk % 8) || (ldv % 8) || (q_lo % 8) || (k_lo % 8) || (v_lo % 8)) return -2;
    if ((qw == nullptr) != (kw == nullptr) || (qw != nullptr && (qb == nullptr || kb == nullptr))) return -3;
    if ((cos_t == nullptr) != (sin_t == nullptr) || (cos_t != nullptr && (P <= 0 || gw <= 0))) return -3;
    if (((uintptr_t)cos_t | (uintptr_t)sin_t) % 16) return -2;   // the tables are read as 16-byte vectors
    // the kernel reads qkv and writes q / k / v (and their lo halves) as 16-byte vectors over a fixed [3 x 16 heads x 64] row
    if (((uintptr_t)qkv | (uintptr_t)q_out | (uintptr_t)k_out | (uintptr_t)v_out) % 16) return -2;
    if (ld_in < 3072 || ldq < 1024 || ldk < 1024 || ldv < 1024) return -2;
    QkvSplitParams p;
    p.qkv = qkv; p.ld_in = ld_in;
    p.q_out = (bf16_t*)q_out; p.ldq = ldq; p.q_lo = q_lo;
    p.k_out = (bf16_t*)k_out; p.ldk = ldk; p.k_lo = k_lo;
    p.v_out = (bf16_t*)v_out; p.ldv = ldv; p.v_lo = v_lo;
    p.qw = qw; p.qb = qb; p.kw = kw; p.kb = kb; p.cos_t = cos_t; p.sin_t = sin_t;
    p.T = T; p.P = P > 0 ? P : 1; p.gw = gw > 0 ? gw : 1; p.patch_start = patch_start; p.eps = eps; p.q_scale = q_scale; p.C = 1024;
    const int grid = T < 8192 ? T : 8192;
    hipLaunchKernelGGL(qkv_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_split3_f16(const float* x, long ldx, void* out, long ldo, int rows, int N, int act, void* stream) {
    if (rows <= 0 || N <= 0 || (N % 4) || (ldx % 4) || (ldo % 4) || ldo < 3L * N || act < 0 || act > 1) return -1;
    if (((uintptr_t)x % 16) || ((uintptr_t)out % 8)) return -2;
    const long total = (long)rows * (N / 4);
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, (bf16_t*)out, ldo, rows, N, act);
    IGGT_CHECK_LAUNCH();
    return 0;
}
