// Mean-input compensation for the two-pass convolutions (PREC = 2 of conv_igemm.hip / conv3x3_halo.hip).
//
// PREC = 2 multiplies exact activations (fp16 hi + fp16 lo) by weights rounded ONCE to fp16: two MFMA passes instead of the
// three of the split-bf16 scheme.  What is lost is  x * dW  with dW = W - fp16(W).  As in the 16-bit trunk
// (csrc/elementwise.hip colmean / bias_correct) the part of it that does not average out is the response to the MEAN input:
//     x * dW = mu * dW + (x - mu) * dW,     mu[c] = mean of input channel c over all pixels (after the fused ReLU-on-load)
// and mu * dW is a per-output-channel constant -- except along the image border, where the taps that fall on the zero padding
// contribute nothing.  So the constant is computed per BORDER CLASS (first row / middle / last row) x (first column / middle /
// last column): nine vectors of Cout floats, bias included, that the convolution epilogue adds instead of the bias.
// Measured on photographs (profiles/r03_conv_precision.txt): head output error 1.7e-4 with the compensation on every layer,
// 5.4e-4 ... 1.7e-3 without it (post-ReLU feature maps have large means).
//
//   conv_chanmean_kernel   partial[rb][c] = sum over a strided sample of the pixels (<= ~16 K rows: the mean only has to be
//                          close, any vector gives an exact identity above) -- grid (C / 64, R), fixed summation order
//   conv_corr_kernel       mu = sum of the partials / nsamp;  T[tap][n] = sum_c mu[c] dW[n][tap][c]  (dW bf16);
//                          corr[class][n] = bias[n] + sum of T[tap][n] over the taps that are inside the image for that class
// Both are launched by iggt_conv2d_nhwc_f32_ws in front of the convolution, on its stream, into its workspace.
#include "common.h"

namespace {

struct ChanMeanParams {
    const float* x;
    long ldx, step;
    int C, nsamp, R, relu;
    float* partial;   // [R][C]
};

__global__ __launch_bounds__(256) void conv_chanmean_kernel(const ChanMeanParams p) {
    __shared__ float red[16][65];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cg * 4;
    const int rb = blockIdx.y;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < p.C) {
        const float floor_v = p.relu ? 0.f : -INFINITY;
        const float* src = p.x + c;
        const long rstride = p.step * p.ldx;
#pragma unroll 4
        for (long i = rb * 16 + rl; i < p.nsamp; i += (long)p.R * 16) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * rstride);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += fmaxf(v[e], floor_v);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cg * 4 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;   // fixed order: deterministic
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
        const int cc = blockIdx.x * 64 + threadIdx.x;
        if (cc < p.C) p.partial[(long)rb * p.C + cc] = s;
    }
}

struct CorrParams {
    const float* partial;
    int R, nsamp;
    const bf16_t* dw;     // [Cout][KH * KW * Cin], tap-major like the weights
    int Cin, Cout, ntaps, KW;
    const float* bias;    // or null
    int rowmask[3], colmask[3];   // bit ky / kx set: the tap is inside the image for that class
    float* corr;          // [9][Cout], class = 3 * cy + cx
};

__global__ __launch_bounds__(256) void conv_corr_kernel(const CorrParams p) {
    extern __shared__ float mean[];
    const float inv = 1.0f / (float)p.nsamp;
    for (int c = threadIdx.x; c < p.Cin; c += 256) {
        // eight loads in flight (one dependent L2 load after the other cost 34 us for R = 64); fixed order: deterministic
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = 0;
        for (; r + 8 <= p.R; r += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += p.partial[(long)(r + u) * p.Cin + c];
        }
        for (; r < p.R; ++r) s[0] += p.partial[(long)r * p.Cin + c];
        mean[c] = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) * inv;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.Cout) return;
    float T[16];
    const bf16_t* row = p.dw + (long)n * p.ntaps * p.Cin;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float acc = 0.f;
        if (t < p.ntaps) {
            for (int k = lane * 8; k < p.Cin; k += 512) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(row + (long)t * p.Cin + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc += bf16_lo(raw[e]) * mean[k + 2 * e] + bf16_hi(raw[e]) * mean[k + 2 * e + 1];
            }
            acc = wave_sum(acc);
        }
        T[t] = acc;
    }
    if (lane < 9) {
        const int cy = lane / 3, cx = lane - cy * 3;
        float s = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int ky = t / p.KW, kx = t - ky * p.KW;
            if (t < p.ntaps && ((p.rowmask[cy] >> ky) & 1) && ((p.colmask[cx] >> kx) & 1)) s += T[t];
        }
        p.corr[(long)lane * p.Cout + n] = s;
    }
}

}  // namespace

// Bytes of workspace the two kernels need in front of the convolution: R x Cin partial sums + 9 x Cout corrections.
long iggt_conv_meancomp_ws_bytes(int Cin, int Cout) { return (64L * Cin + 9L * Cout) * 4 + 256; }

// Tap validity of the first / a middle / the last kernel placement along one axis (bit k: tap k reads inside the map).
// Returns false when a middle placement touches the padding too (then nine classes are not enough: pad > stride).
static bool axis_masks(int n_out, int n_in, int K, int stride, int pad, int mask[3]) {
    auto valid = [&](int o) {
        int m = 0;
        for (int k = 0; k < K; ++k) {
            const int i = o * stride - pad + k;
            if (i >= 0 && i < n_in) m |= 1 << k;
        }
        return m;
    };
    const int full = (1 << K) - 1;
    mask[0] = valid(0);
    mask[2] = valid(n_out - 1);
    mask[1] = full;
    for (int o = 1; o < n_out - 1; ++o) {   // every middle placement must see the whole kernel (cheap: only near the ends can fail)
        if (o > 2 && o < n_out - 3) continue;
        if (valid(o) != full) return false;
    }
    return true;
}

// Launches the two kernels.  corr_out ([9][Cout] floats) and the partial sums live in `ws`.  *uniform = 1 when all nine
// classes are equal (no tap ever falls outside: 1 x 1 convolutions) -- the caller then passes corr_out + 4 * Cout as a bias.
// Returns 0, or -100 when the geometry has no nine-class description (the caller falls back to PREC = 3).
int iggt_launch_conv_meancomp(const float* x, int ldx, int Nimg, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int KH,
                              int KW, int stride, int pad_y, int pad_x, int relu_in, const void* dw, const float* bias,
                              void* ws, long ws_bytes, float** corr_out, int* uniform, hipStream_t st) {
    if (KH * KW > 16 || Cin > 8192 || (Cin % 8) != 0 || (ldx % 4) != 0) return -100;
    if (ws == nullptr || ws_bytes < iggt_conv_meancomp_ws_bytes(Cin, Cout) || ((uintptr_t)ws % 16) != 0) return -100;
    CorrParams c;
    if (!axis_masks(Ho, Hi, KH, stride, pad_y, c.rowmask) || !axis_masks(Wo, Wi, KW, stride, pad_x, c.colmask)) return -100;
    const long rows = (long)Nimg * Hi * Wi;
    long step = rows / 16384;
    if (step < 1) step = 1;
    if (step > 1) {
        step |= 1;                                  // odd and not a divisor of the row length: no column aliasing
        while (Wi % step == 0 && step > 1) step += 2;
    }
    ChanMeanParams m;
    m.x = x; m.ldx = ldx; m.step = step; m.C = Cin; m.relu = relu_in;
    m.nsamp = (int)((rows + step - 1) / step);
    m.R = m.nsamp >= 64 * 16 ? 64 : (m.nsamp + 15) / 16;
    m.partial = (float*)ws;
    hipLaunchKernelGGL(conv_chanmean_kernel, dim3((unsigned)((Cin + 63) / 64), (unsigned)m.R), dim3(256), 0, st, m);
    IGGT_CHECK_LAUNCH();
    float* corr = (float*)ws + 64L * Cin;
    c.partial = m.partial; c.R = m.R; c.nsamp = m.nsamp; c.dw = (const bf16_t*)dw; c.Cin = Cin; c.Cout = Cout;
    c.ntaps = KH * KW; c.KW = KW; c.bias = bias; c.corr = corr;
    hipLaunchKernelGGL(conv_corr_kernel, dim3((unsigned)((Cout + 3) / 4)), dim3(256), Cin * sizeof(float), st, c);
    IGGT_CHECK_LAUNCH();
    *corr_out = corr;
    const int fy = (1 << KH) - 1, fx = (1 << KW) - 1;
    *uniform = (c.rowmask[0] == fy && c.rowmask[2] == fy && c.colmask[0] == fx && c.colmask[2] == fx) ? 1 : 0;
    return 0;
}
