// fp32 "small" operators of the heads: everything the reference computes in fp32 on a handful of tokens or as a cheap
// per-pixel map, where a 16-bit MFMA GEMM would cost accuracy and a tensor-library call costs a launch each.
//
//   iggt_linear_f32          exact-fp32 nn.Linear for skinny problems (M = S camera tokens, squeeze-excite vectors, ...):
//                            v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: bit-for-bit an fmaf chain), one 32 x 32 output
//                            tile per workgroup, K split over its 8 waves, weights streamed once -- the op is bound by the
//                            weight bytes (4 B per parameter), not by the matrix pipe.
//                            reference: camera_head.py:83-154 (trunk Linears, embed_pose, poseLN_modulation, pose_branch),
//                            window_sa.py:26-37 (ChannelAttention 1x1 convs on a pooled vector)
//   iggt_attn_f32            softmax(scale q k^T) v in fp32, head dim 32 / 64 / 128, token-major strided operands:
//                            camera trunk attention over the S views (16 heads x 128, camera_head.py:124 -> layers/block.py),
//                            CrossAttention of the part head (8 heads x 32, heads/block.py:212-242)
//   iggt_adaln_modulate_f32  gate * (LayerNorm_noaffine(x) * (1 + scale) + shift) + x           (camera_head.py:130-134)
//   iggt_pose_update_f32     pred (+)= delta; activate_pose (T, quat linear; FoV relu)           (camera_head.py:144-151, head_act.py:9-37)
//   iggt_conv1x1_c32_nchw_f32  last 1x1 conv of the part head, NHWC in -> NCHW out               (part_head.py:128,240-243)
//   iggt_pose_to_extri_intri_f32 / iggt_unproject_depth_f32   pose encoding -> [R|t], K; depth -> world points
//                            (utils/pose_enc.py:65-130, utils/rotation.py:14-44, utils/geometry.py:151-268)
#include "common.h"
#include "gemm_common.h"
#include "../../include/iggt_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Linear, fp32 MFMA.  A = x rows (lane -> row m0 + lane % 32, k pair member lane / 32), B = W rows (= columns of W^T).
struct LinParams {
    const float* x; long ldx;
    const float* w; long ldw;
    const float* bias; const float* gamma;
    const float* res; long ldr;
    float* out; long ldo;
    int M, N, K, act;
    // split-K over blockIdx.z (ks > 1): the ks workgroups of a tile write their partial 32 x 32 sums to part[tile][z][1024];
    // a second, tiny kernel adds them in z order (fixed summation order) and applies the epilogue.  (One kernel with a
    // "last workgroup to arrive reduces" counter was 10x SLOWER than no split at all: the device-scope release it needs
    // writes back the L2 of the XCD, and the 8 XCDs of this chip do not share an L2.)
    int ks, kchunks;          // K chunks of 16 per z slice
    float* part;
};

IGGT_DEVINL float lin_act(float v, int act) {
    switch (act) {
        case 1: return gelu_erf(v);
        case 2: return fmaxf(v, 0.f);
        case 3: return v / (1.f + __expf(-v));          // SiLU
        case 4: return 1.f / (1.f + __expf(-v));        // sigmoid
        default: return v;
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(512) void linear_f32_kernel(const LinParams p) {
    __shared__ float red[8][32][33];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int r = lane & 31, h = lane >> 5;
    int mr = m0 + r, nr = n0 + r;
    mr = mr < p.M ? mr : p.M - 1;
    nr = nr < p.N ? nr : p.N - 1;
    const float* xrow = p.x + (long)mr * p.ldx;
    const float* wrow = p.w + (long)nr * p.ldw;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // k chunks of 16 interleaved over the 8 waves: lane half h owns k = 16 c + 8 h + [0, 8)
    const int nchunks_all = (p.K + 15) >> 4;
    const int cbeg = blockIdx.z * p.kchunks;
    const int nchunks = (cbeg + p.kchunks < nchunks_all) ? cbeg + p.kchunks : nchunks_all;
#pragma unroll 4
    for (int c = cbeg + wave; c < nchunks; c += 8) {
        const int k0 = c * 16 + 8 * h;
        float xa[8], wb[8];
        if (ALIGNED && c * 16 + 16 <= p.K) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xrow + k0), x1 = *reinterpret_cast<const f32x4*>(xrow + k0 + 4);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow + k0), w1 = *reinterpret_cast<const f32x4*>(wrow + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xa[e] = x0[e]; xa[4 + e] = x1[e]; wb[e] = w0[e]; wb[4 + e] = w1[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = k0 + e < p.K;
                xa[e] = ok ? xrow[k0 + e] : 0.f;
                wb[e] = ok ? wrow[k0 + e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[e], wb[e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][mfma32_row(i, lane)][r] = acc[i];
    __syncthreads();
    float vsum[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = e * 512 + tid, row = idx >> 5, col = idx & 31;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[w][row][col];
        vsum[e] = v;
    }
    if (p.ks > 1) {   // split K: this workgroup's partial sums; linear_finalize_kernel adds the slices and applies the epilogue
        const int tile = blockIdx.y * gridDim.x + blockIdx.x;
        float* mine = p.part + ((long)tile * p.ks + blockIdx.z) * 1024;
        mine[tid] = vsum[0];
        mine[512 + tid] = vsum[1];
        return;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = e * 512 + tid, row = idx >> 5, col = idx & 31;
        float v = vsum[e];
        const int m = m0 + row, n = n0 + col;
        if (m < p.M && n < p.N) {
            if (p.bias) v += p.bias[n];
            v = lin_act(v, p.act);
            if (p.gamma) v *= p.gamma[n];
            if (p.res) v += p.res[(long)m * p.ldr + n];
            p.out[(long)m * p.ldo + n] = v;
        }
    }
}

__global__ __launch_bounds__(256) void linear_finalize_kernel(const LinParams p, int tiles_n) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // element (m, n) of the padded [tiles_m * 32][tiles_n * 32] grid
    const int ncols = tiles_n * 32;
    const int m = idx / ncols, n = idx - m * ncols;
    if (m >= p.M || n >= p.N) return;
    const int tile = (m >> 5) * tiles_n + (n >> 5);
    const float* src = p.part + (long)tile * p.ks * 1024 + (m & 31) * 32 + (n & 31);
    float v = 0.f;
    for (int z = 0; z < p.ks; ++z) v += src[z * 1024];
    if (p.bias) v += p.bias[n];
    v = lin_act(v, p.act);
    if (p.gamma) v *= p.gamma[n];
    if (p.res) v += p.res[(long)m * p.ldr + n];
    p.out[(long)m * p.ldo + n] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// Attention, fp32, on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate, 157 TFLOP/s).
// Same dataflow as the 16-bit flash kernel, one precision class up: a wave owns 32 query rows, S^T[key][q] = K . Q^T
// (A = K rows from LDS, B = Q held in registers) so a lane owns one query column and the online softmax is lane-local plus
// one exchange with lane ^ 32; the exponentiated accumulator registers feed O^T = V^T . P^T directly as the B operand --
// register j of lane half h is key (j & 3) + 8 (j >> 2) + 4 h, so the A operand of MFMA j reads exactly those two V rows.
// (A first version did the dot products on the VALU with 4 threads per row and two LDS shuffles per key: 7.0 ms for the
// part head's cross attention at 8 x 504^2 -- 8 x 8 heads x 1296^2 x 32 -- against ~0.15 ms of matrix-pipe time.)
struct AttnF32Params {
    const float* q; const float* k; const float* v; float* o;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;   // elements; head h at column offset h * D
    int B, H, Nq, Nk;
    float scale_log2;
};

template <int D>
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnF32Params p) {
    constexpr int KP = D + 1;                  // padded K row: the A-operand read (lane -> key row) is conflict-free
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float* Ks = smf;                           // [32][D + 1]
    float* Vs = smf + 32 * KP;                 // [32][D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    int qr = blockIdx.x * 128 + wave * 32 + r;
    const bool valid = qr < p.Nq;
    qr = valid ? qr : p.Nq - 1;
    const float* qp = p.q + (long)b * p.q_bs + (long)qr * p.q_rs + h * D;
    float qreg[D / 2];                         // B operand of MFMA j: Q[q][2 j + hh], pre-scaled by scale * log2 e
#pragma unroll
    for (int j = 0; j < D / 2; ++j) qreg[j] = qp[2 * j + hh] * p.scale_log2;
    f32x16 o[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float* kb = p.k + (long)b * p.k_bs + h * D;
    const float* vb = p.v + (long)b * p.v_bs + h * D;
    for (int t0 = 0; t0 < p.Nk; t0 += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * (D / 4); i += 256) {
            const int kr = i / (D / 4), c4 = i - kr * (D / 4);
            int kg = t0 + kr;
            kg = kg < p.Nk ? kg : p.Nk - 1;
            const f32x4 kk = *reinterpret_cast<const f32x4*>(kb + (long)kg * p.k_rs + 4 * c4);
            Ks[kr * KP + 4 * c4] = kk[0]; Ks[kr * KP + 4 * c4 + 1] = kk[1];
            Ks[kr * KP + 4 * c4 + 2] = kk[2]; Ks[kr * KP + 4 * c4 + 3] = kk[3];
            *reinterpret_cast<f32x4*>(Vs + kr * D + 4 * c4) = *reinterpret_cast<const f32x4*>(vb + (long)kg * p.v_rs + 4 * c4);
        }
        __syncthreads();
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int j = 0; j < D / 2; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[r * KP + 2 * j + hh], qreg[j], s, 0, 0, 0);
        const int nk = p.Nk - t0;              // valid keys in this tile (>= 1)
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = (e & 3) + 8 * (e >> 2) + 4 * hh;
            s[e] = key < nk ? s[e] : -INFINITY;
            mx = fmaxf(mx, s[e]);
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
    l += __shfl_xor(l, 32, 64);
    if (valid) {
        const float inv = 1.f / l;
        float* op = p.o + (long)b * p.o_bs + (long)qr * p.o_rs + h * D;
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) op[i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh] = o[i][e] * inv;
    }
}

// Many short sequences (the tracker's attention along time: thousands of tracks x 8 heads, S <= 16 frames each): the
// MFMA kernel above gives a 4-wave workgroup to every (sequence, head) and fills 8 of its 128 query rows (107 us for
// 1 088 x 8 sequences of 8 tokens).  Here a THREAD owns one query row: q and the output row live in registers, the L keys /
// values of its sequence are read straight from global memory (the L threads of a sequence read the same rows: broadcast
// loads), online softmax in the loop.  D = 64.
__global__ __launch_bounds__(256) void attn_short_f32_kernel(const AttnF32Params p) {
    constexpr int D = 64;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.B * p.H * p.Nq;
    if (idx >= total) return;
    const int i = (int)(idx % p.Nq);
    const long bh = idx / p.Nq;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const f32x4* qp = reinterpret_cast<const f32x4*>(p.q + (long)b * p.q_bs + (long)i * p.q_rs + h * D);
    f32x4 q[D / 4], o[D / 4];
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        q[c] = qp[c] * p.scale_log2;
        o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float m = -INFINITY, l = 0.f;
    const float* kb = p.k + (long)b * p.k_bs + h * D;
    const float* vb = p.v + (long)b * p.v_bs + h * D;
    for (int j = 0; j < p.Nk; ++j) {
        const f32x4* kp = reinterpret_cast<const f32x4*>(kb + (long)j * p.k_rs);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < D / 4; ++c) {
            const f32x4 kk = kp[c];
            s += q[c][0] * kk[0] + q[c][1] * kk[1] + q[c][2] * kk[2] + q[c][3] * kk[3];
        }
        const float mn = fmaxf(m, s);
        const float alpha = __builtin_amdgcn_exp2f(m - mn), pj = __builtin_amdgcn_exp2f(s - mn);
        m = mn;
        l = l * alpha + pj;
        const f32x4* vp = reinterpret_cast<const f32x4*>(vb + (long)j * p.v_rs);
#pragma unroll
        for (int c = 0; c < D / 4; ++c) o[c] = o[c] * alpha + vp[c] * pj;
    }
    const float inv = 1.f / l;
    f32x4* op = reinterpret_cast<f32x4*>(p.o + (long)b * p.o_bs + (long)i * p.o_rs + h * D);
#pragma unroll
    for (int c = 0; c < D / 4; ++c) op[c] = o[c] * inv;
}

// ---------------------------------------------------------------------------------------------------------------
// adaLN modulation: out = gate * (LN(x) * (1 + scale) + shift) + x, LN without affine, one block per row.
struct AdaLnParams {
    const float* x; long ldx;
    const float* shift; const float* scale; const float* gate; long ldm;
    float* out; long ldo;
    int rows, C; float eps;
};

__global__ __launch_bounds__(256) void adaln_modulate_kernel(const AdaLnParams p) {
    __shared__ float sh[2][4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = p.x + (long)row * p.ldx;
    float s = 0.f, s2 = 0.f;
    for (int c = tid; c < p.C; c += 256) { const float v = x[c]; s += v; }
    s = wave_sum(s);
    if ((tid & 63) == 0) sh[0][tid >> 6] = s;
    __syncthreads();
    const float mean = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) / p.C;
    for (int c = tid; c < p.C; c += 256) { const float d = x[c] - mean; s2 += d * d; }
    s2 = wave_sum(s2);
    if ((tid & 63) == 0) sh[1][tid >> 6] = s2;
    __syncthreads();
    const float rstd = rsqrtf((sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) / p.C + p.eps);
    const long mo = (long)row * p.ldm;
    for (int c = tid; c < p.C; c += 256) {
        const float ln = (x[c] - mean) * rstd;
        p.out[(long)row * p.ldo + c] = p.gate[mo + c] * (ln * (1.f + p.scale[mo + c]) + p.shift[mo + c]) + x[c];
    }
}

// pred = first ? delta : pred + delta;  out = activate_pose(pred): T (0..2) and quaternion (3..6) linear, FoV (7, 8) relu.
__global__ void pose_update_kernel(const float* delta, float* pred, float* out, int n, int first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 9) return;
    const float v = first ? delta[i] : pred[i] + delta[i];
    pred[i] = v;
    out[i] = (i % 9) >= 7 ? fmaxf(v, 0.f) : v;
}

// 1x1 convolution from 32 NHWC channels to Cout <= 8 NCHW planes.
__global__ __launch_bounds__(256) void conv1x1_c32_nchw_kernel(const float* x, long ldx, const float* w, const float* b,
                                                                float* y, long hw, long npix, int Cout) {
    __shared__ float ws[8 * 32 + 8];
    for (int i = threadIdx.x; i < Cout * 32; i += 256) ws[i] = w[i];
    if (threadIdx.x < Cout) ws[8 * 32 + threadIdx.x] = b ? b[threadIdx.x] : 0.f;
    __syncthreads();
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(x + pix * ldx + 4 * i);
    const long n = pix / hw, r = pix - n * hw;
    for (int c = 0; c < Cout; ++c) {
        float a = ws[8 * 32 + c];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) a += v[i][e] * ws[c * 32 + 4 * i + e];
        y[(n * Cout + c) * hw + r] = a;
    }
}

// pose encoding [n][9] = (T, quat xyzw, fov_h, fov_w) -> extrinsics [n][3][4] = [R | T], intrinsics [n][3][3]
__global__ void pose_to_extri_intri_kernel(const float* pose, float* extri, float* intri, int n, float H, float W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* pe = pose + 9 * i;
    const float qi = pe[3], qj = pe[4], qk = pe[5], qr = pe[6];
    const float two_s = 2.0f / (qi * qi + qj * qj + qk * qk + qr * qr);
    float* e = extri + 12 * i;
    e[0] = 1 - two_s * (qj * qj + qk * qk); e[1] = two_s * (qi * qj - qk * qr);     e[2] = two_s * (qi * qk + qj * qr);     e[3] = pe[0];
    e[4] = two_s * (qi * qj + qk * qr);     e[5] = 1 - two_s * (qi * qi + qk * qk); e[6] = two_s * (qj * qk - qi * qr);     e[7] = pe[1];
    e[8] = two_s * (qi * qk - qj * qr);     e[9] = two_s * (qj * qk + qi * qr);     e[10] = 1 - two_s * (qi * qi + qj * qj); e[11] = pe[2];
    if (intri != nullptr) {
        float* k = intri + 9 * i;
        const float fy = (H / 2.0f) / tanf(pe[7] / 2.0f), fx = (W / 2.0f) / tanf(pe[8] / 2.0f);
        k[0] = fx; k[1] = 0.f; k[2] = W / 2; k[3] = 0.f; k[4] = fy; k[5] = H / 2; k[6] = 0.f; k[7] = 0.f; k[8] = 1.f;
    }
}

// world = R^T (cam - t),  cam = ((u - cu) d / fu, (v - cv) d / fv, d); fp64 inside like the reference's numpy path.
__global__ __launch_bounds__(256) void unproject_depth_kernel(const float* depth, const float* extri, const float* intri,
                                                               float* out, int S, int H, int W) {
    const long hw = (long)H * W;
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= S * hw) return;
    const int s = (int)(pix / hw);
    const long r = pix - s * hw;
    const int vy = (int)(r / W), ux = (int)(r - (long)vy * W);
    const float* e = extri + 12 * s;
    const float* k = intri + 9 * s;
    const double d = depth[pix];
    const double xc = ((double)ux - (double)k[2]) * d / (double)k[0];
    const double yc = ((double)vy - (double)k[5]) * d / (double)k[4];
    // the reference stacks the camera coordinates as float32 before the rigid transform (geometry.py:266)
    const double cx = (double)(float)xc, cy = (double)(float)yc, cz = (double)(float)d;
    const double tx = -((double)e[0] * e[3] + (double)e[4] * e[7] + (double)e[8] * e[11]);
    const double ty = -((double)e[1] * e[3] + (double)e[5] * e[7] + (double)e[9] * e[11]);
    const double tz = -((double)e[2] * e[3] + (double)e[6] * e[7] + (double)e[10] * e[11]);
    out[3 * pix + 0] = (float)(cx * e[0] + cy * e[4] + cz * e[8] + tx);
    out[3 * pix + 1] = (float)(cx * e[1] + cy * e[5] + cz * e[9] + ty);
    out[3 * pix + 2] = (float)(cx * e[2] + cy * e[6] + cz * e[10] + tz);
}

}  // namespace

static const long LIN_WS_COUNTERS = 4096;   // ints in front of the partial sums

extern "C" int iggt_linear_f32_ws(const float* x, long ldx, const float* w, long ldw, const float* bias, const float* gamma,
                                  const float* res, long ldr, float* out, long ldo, int M, int N, int K, int act,
                                  void* ws, long ws_bytes, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 4) return -1;
    if (ldx < K || ldw < K || ldo < N || (res && ldr < N)) return -2;
    LinParams p{x, ldx, w, ldw, bias, gamma, res, ldr, out, ldo, M, N, K, act, 1, (K + 15) >> 4, nullptr};
    const int tiles = ((N + 31) / 32) * ((M + 31) / 32), nchunks = (K + 15) >> 4;
    // weight streaming needs many loads in flight: aim at >= 1024 workgroups (4 per CU), at least 8 chunks (one per wave) each
    // (measured at M = 32, K = 2048: 64 tiles 21.6 -> 15.3 us, 32 tiles 21.5 -> 11.9, 64 tiles with K = 8192 74 -> 30;
    //  192 / 256 tiles 22.7 -> 24.4 / 24.6 -> 27.5: the second launch costs more than the split gains there)
    if (ws != nullptr && ((uintptr_t)ws % 16) == 0 && tiles < 128 && nchunks >= 16) {
        int ks = (1024 + tiles - 1) / tiles;
        if (ks > nchunks / 8) ks = nchunks / 8;
        if (ks > 32) ks = 32;
        const long need = LIN_WS_COUNTERS * 4 + (long)tiles * ks * 1024 * 4;
        if (ks > 1 && need <= ws_bytes) {
            p.ks = ks;
            p.kchunks = (nchunks + ks - 1) / ks;
            p.ks = (nchunks + p.kchunks - 1) / p.kchunks;   // no empty slice
            p.part = (float*)((char*)ws + LIN_WS_COUNTERS * 4);
        }
    }
    const dim3 grid((N + 31) / 32, (M + 31) / 32, p.ks), block(512);
    const bool aligned = (ldx % 4 == 0) && (ldw % 4 == 0) && (((uintptr_t)x | (uintptr_t)w) % 16 == 0);
    if (aligned) hipLaunchKernelGGL(linear_f32_kernel<true>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(linear_f32_kernel<false>, grid, block, 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    if (p.ks > 1) {
        const int tiles_n = (N + 31) / 32, tiles_m = (M + 31) / 32;
        hipLaunchKernelGGL(linear_finalize_kernel, dim3((unsigned)(tiles_n * tiles_m * 4)), dim3(256), 0, (hipStream_t)stream, p,
                           tiles_n);
        IGGT_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int iggt_linear_f32(const float* x, long ldx, const float* w, long ldw, const float* bias, const float* gamma,
                               const float* res, long ldr, float* out, long ldo, int M, int N, int K, int act,
                               void* stream) {
    return iggt_linear_f32_ws(x, ldx, w, ldw, bias, gamma, res, ldr, out, ldo, M, N, K, act, nullptr, 0, stream);
}

extern "C" long iggt_linear_f32_ws_bytes(void) { return LIN_WS_COUNTERS * 4 + 2048L * 1024 * 4; }

extern "C" int iggt_attn_f32(const float* q, const float* k, const float* v, float* o, int B, int H, int Nq, int Nk,
                             int head_dim, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs,
                             long o_rs, float scale, void* stream) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return -1;
    if ((q_rs | k_rs | v_rs | o_rs | q_bs | k_bs | v_bs | o_bs) % 4) return -2;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16) return -2;
    AttnF32Params p{q, k, v, o, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, B, H, Nq, Nk, scale * 1.4426950408889634f};
    if (head_dim == 64 && Nq <= 16 && Nk <= 16 && (long)B * H >= 512) {      // many short sequences: one thread per query row
        // (measured on the tracker's time attention, 1 088 x 8 sequences: 8 frames 107 -> 48 us; at 32 frames the MFMA kernel wins)
        const long rows = (long)B * H * Nq;
        hipLaunchKernelGGL(attn_short_f32_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
        IGGT_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((Nq + 127) / 128, H, B), block(256);
    const size_t lds = 32 * (size_t)(2 * head_dim + 1) * sizeof(float);
    if (head_dim == 32) hipLaunchKernelGGL(attn_f32_kernel<32>, grid, block, lds, (hipStream_t)stream, p);
    else if (head_dim == 64) hipLaunchKernelGGL(attn_f32_kernel<64>, grid, block, lds, (hipStream_t)stream, p);
    else if (head_dim == 128) hipLaunchKernelGGL(attn_f32_kernel<128>, grid, block, lds, (hipStream_t)stream, p);
    else return -3;
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_adaln_modulate_f32(const float* x, long ldx, const float* shift, const float* scale, const float* gate,
                                       long ldm, float* out, long ldo, int rows, int C, float eps, void* stream) {
    if (rows <= 0 || C <= 0) return -1;
    AdaLnParams p{x, ldx, shift, scale, gate, ldm, out, ldo, rows, C, eps};
    hipLaunchKernelGGL(adaln_modulate_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_pose_update_f32(const float* delta, float* pred, float* out, int n, int first, void* stream) {
    if (n <= 0) return -1;
    hipLaunchKernelGGL(pose_update_kernel, dim3((n * 9 + 255) / 256), dim3(256), 0, (hipStream_t)stream, delta, pred, out, n,
                       first);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_conv1x1_c32_nchw_f32(const float* x, long ldx, const float* w, const float* b, float* y, long hw,
                                         long npix, int Cout, void* stream) {
    if (npix <= 0 || hw <= 0 || Cout <= 0 || Cout > 8 || ldx < 32 || (ldx % 4) || (npix % hw)) return -1;
    hipLaunchKernelGGL(conv1x1_c32_nchw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       w, b, y, hw, npix, Cout);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_pose_to_extri_intri_f32(const float* pose, float* extri, float* intri, int n, int H, int W,
                                            void* stream) {
    if (n <= 0 || H <= 0 || W <= 0) return -1;
    hipLaunchKernelGGL(pose_to_extri_intri_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, extri, intri, n,
                       (float)H, (float)W);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_unproject_depth_f32(const float* depth, const float* extri, const float* intri, float* out, int S,
                                        int H, int W, void* stream) {
    if (S <= 0 || H <= 0 || W <= 0) return -1;
    const long npix = (long)S * H * W;
    hipLaunchKernelGGL(unproject_depth_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, depth,
                       extri, intri, out, S, H, W);
    IGGT_CHECK_LAUNCH();
    return 0;
}
