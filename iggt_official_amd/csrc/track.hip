// Track head (the `query_points` path of IGGT.forward / VGGT.forward, reference iggt/models/vggt.py:220-227): the
// operators of BaseTrackerPredictor that are not a Linear / LayerNorm(128) / attention (those run on iggt_linear_f32,
// iggt_layernorm_f32 and iggt_attn_f32).  Everything fp32, like the reference (autocast is off for the heads).
//
//   iggt_layernorm_rows_f32      nn.LayerNorm over rows of ANY width (388, 384, 130-strided ...), one wave per row
//                                (track_modules/blocks.py:44,48; modules.py:168-169,203-205; GroupNorm(1, C) on a [M, C]
//                                matrix is the same function: base_track_predictor.py:74,183)
//   iggt_avgpool2_nhwc_f32       F.avg_pool2d(2, 2) of the correlation pyramid (blocks.py:170-180)
//   iggt_sample_points_nhwc_f32  sample_features4d: bilinear, align_corners, border padding (utils.py:192-226)
//   iggt_track_corr_f32          CorrBlock.corr_sample (blocks.py:189-241) WITHOUT the correlation volume: bilinear
//                                sampling is linear, so the (2r+1)^2 samples of <target, fmap>/sqrt(C) around a point are
//                                mixes of the (2r+2)^2 dot products of the target with the feature vectors of the patch
//                                under the window -- 100 x 512 B per (track, frame, level) instead of a H x W x N volume
//                                (at 518^2, 1024 tracks, 8 frames: 2.2 GB of volume per level and iteration)
//   iggt_track_tokens_f32        transformer input: flow embedding (utils.py:90-121) | flows / max_scale x 2 |
//                                corr_mlp output | track features, + sampled position embedding + query / reference
//                                token (base_track_predictor.py:139-163)
//   iggt_track_posemb_f32        get_2d_sincos_pos_embed sampled at the query points (utils.py:17-87, 192-226)
//   iggt_track_update_f32        coords += delta, frame 0 pinned to the query, prediction in image pixels
//                                (base_track_predictor.py:168-195)
#include "common.h"
#include "../../include/iggt_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// one wave per row; x, w, b, out may be unaligned (row offsets like delta[:, 2:130]); x2 != NULL: LayerNorm(x + x2)
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* x, long ldx, const float* x2, long ldx2,
                                                              const float* w, const float* b, float* out, long ldo,
                                                              int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * ldx;
    const float* yr = x2 ? x2 + (long)row * ldx2 : nullptr;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += yr ? xr[c] + yr[c] : xr[c];
    const float mean = wave_sum(s) / C;
    float s2 = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = (yr ? xr[c] + yr[c] : xr[c]) - mean; s2 += d * d; }
    const float rstd = rsqrtf(wave_sum(s2) / C + eps);
    float* o = out + (long)row * ldo;
    for (int c = lane; c < C; c += 64) o[c] = ((yr ? xr[c] + yr[c] : xr[c]) - mean) * rstd * w[c] + b[c];
}

// out[n][y][x][c] = ((a + b) + c + d) / 4 over the 2 x 2 block, floor sizes; one thread per 4 channels
__global__ __launch_bounds__(256) void avgpool2_nhwc_kernel(const float* x, float* y, int N, int H, int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const f32x4* p = reinterpret_cast<const f32x4*>(x) + (((long)n * H + 2 * oy) * W + 2 * ox) * C4 + c;
    const f32x4 a = p[0], b2 = p[C4], c2 = p[(long)W * C4], d = p[(long)W * C4 + C4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (a[e] + b2[e] + c2[e] + d[e]) * 0.25f;
    reinterpret_cast<f32x4*>(y)[i] = o;
}

// bilinear corner set with border clamping (grid_sample(padding_mode="border", align_corners=True))
struct Corners { int x0, y0, x1, y1; float fx, fy; };
IGGT_DEVINL Corners border_corners(float x, float y, int W, int H) {
    x = fminf(fmaxf(x, 0.f), (float)(W - 1));
    y = fminf(fmaxf(y, 0.f), (float)(H - 1));
    Corners c;
    const float xf = floorf(x), yf = floorf(y);
    c.x0 = (int)xf; c.y0 = (int)yf;
    c.fx = x - xf; c.fy = y - yf;
    c.x1 = c.x0 + 1 < W ? c.x0 + 1 : c.x0;      // weight of an out-of-range corner is 0 after the clamp
    c.y1 = c.y0 + 1 < H ? c.y0 + 1 : c.y0;
    return c;
}

// out[n][c] = bilinear(feat[H][W][C], xy[n]); one thread per (point, 4 channels)
__global__ __launch_bounds__(256) void sample_points_kernel(const float* feat, int H, int W, int C4, const float* xy,
                                                             long ldxy, float* out, long ldo, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * C4) return;
    const int n = (int)(i / C4), c = (int)(i - (long)n * C4);
    const Corners k = border_corners(xy[n * ldxy], xy[n * ldxy + 1], W, H);
    const f32x4* f = reinterpret_cast<const f32x4*>(feat);
    const f32x4 a = f[((long)k.y0 * W + k.x0) * C4 + c], b = f[((long)k.y0 * W + k.x1) * C4 + c];
    const f32x4 d = f[((long)k.y1 * W + k.x0) * C4 + c], e = f[((long)k.y1 * W + k.x1) * C4 + c];
    const float w00 = (1.f - k.fx) * (1.f - k.fy), w01 = k.fx * (1.f - k.fy), w10 = (1.f - k.fx) * k.fy, w11 = k.fx * k.fy;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = a[j] * w00 + b[j] * w01 + d[j] * w10 + e[j] * w11;
    *reinterpret_cast<f32x4*>(out + (long)n * ldo + 4 * c) = o;
}

// ---------------------------------------------------------------------------------------------------------------
// Correlation pyramid sampling.  One wave per (track n, frame s, level l); C = 128.
// 16 lanes own one patch pixel (8 channels each, two 16-byte loads = the pixel's 512 contiguous bytes), 4 pixels per step,
// 25 steps for the 10 x 10 patch; the 81 window samples are then mixed from the 100 dot products in LDS.
constexpr int TRK_MAX_LEVELS = 8;
struct CorrParams {
    const float* fmap[TRK_MAX_LEVELS];     // level l: [S][H_l][W_l][128]
    int H[TRK_MAX_LEVELS], W[TRK_MAX_LEVELS];
    const float* feats;                    // [N][S][128] track features (track-major)
    const float* coords;                   // [N][S][2] level-0 pixel coordinates (x, y)
    float* out; long ldo;                  // [N*S][ldo], columns l * 81 + ix * 9 + iy; columns >= levels * 81 zeroed
    int N, S, levels;
    float inv_sqrt_c;
};

__global__ __launch_bounds__(256) void track_corr_kernel(const CorrParams p) {
    __shared__ float cs[4][104];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long total = (long)p.N * p.S * p.levels;
    long wid = (long)blockIdx.x * 4 + wave;
    const bool live = wid < total;
    wid = live ? wid : total - 1;
    const int l = (int)(wid % p.levels);
    const long ns = wid / p.levels;                 // n * S + s
    const int s = (int)(ns % p.S);
    const int H = p.H[l], W = p.W[l];
    const float* fm = p.fmap[l] + (long)s * H * W * 128;
    const float inv = 1.f / (float)(1 << l);
    const float cx = p.coords[2 * ns] * inv, cy = p.coords[2 * ns + 1] * inv;
    // A map side of ONE pixel (the coarsest levels of small inputs) is degenerate in the reference: bilinear_sampler
    // scales by 2 / max(size - 1, 1) and grid_sample(align_corners=True) maps back with (size - 1) / 2 = 0, so every
    // sample position along that axis lands on pixel 0 whatever the coordinate (utils.py:176-189).  Reproduced.
    const bool flat_x = W == 1, flat_y = H == 1;
    const float xf = flat_x ? 0.f : floorf(cx), yf = flat_y ? 0.f : floorf(cy);
    const float fx = flat_x ? 0.f : cx - xf, fy = flat_y ? 0.f : cy - yf;
    // window origin; clamped far outside the map so that the int conversion is defined for any coordinate
    const int x0 = flat_x ? 0 : (int)fminf(fmaxf(xf, -1.0e6f), 1.0e6f) - 4;
    const int y0 = flat_y ? 0 : (int)fminf(fmaxf(yf, -1.0e6f), 1.0e6f) - 4;
    const int l16 = lane & 15, grp = lane >> 4;
    const f32x4* tp = reinterpret_cast<const f32x4*>(p.feats + ns * 128 + 8 * l16);
    const f32x4 t0 = tp[0], t1 = tp[1];
#pragma unroll 5
    for (int it = 0; it < 25; ++it) {
        const int pi = it * 4 + grp, j = pi / 10, i = pi - 10 * j;
        const int y = y0 + j, x = x0 + i;
        float d = 0.f;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const f32x4* fp = reinterpret_cast<const f32x4*>(fm + ((long)y * W + x) * 128 + 8 * l16);
            const f32x4 a = fp[0], b = fp[1];
            d = a[0] * t0[0] + a[1] * t0[1] + a[2] * t0[2] + a[3] * t0[3] + b[0] * t1[0] + b[1] * t1[1] + b[2] * t1[2] +
                b[3] * t1[3];
        }
        d += __shfl_xor(d, 8, 64);
        d += __shfl_xor(d, 4, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 1, 64);
        if (l16 == 0) cs[wave][pi] = d * p.inv_sqrt_c;
    }
    __syncthreads();
    if (!live) return;
    float* o = p.out + ns * p.ldo + l * 81;
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
    for (int e = lane; e < 81; e += 64) {
        const int ix = e / 9, iy = e - 9 * ix;      // reference order: the x offset is the slow index (blocks.py:184-186)
        const float* c = &cs[wave][(flat_y ? 0 : iy) * 10 + (flat_x ? 0 : ix)];
        o[e] = c[0] * w00 + c[1] * w01 + c[10] * w10 + c[11] * w11;
    }
    if (l == 0) for (int e = p.levels * 81 + lane; e < p.ldo; e += 64) p.out[ns * p.ldo + e] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// position embedding at the query points: tabx [W][Ch] / taby [H][Ch] hold [sin | cos](pos * omega) (fp64 -> fp32 on the
// host, exactly as the reference builds its table); channels [0, Ch) depend on x only, [Ch, 2 Ch) on y only, so the
// bilinear sample of the 2-D table is a linear interpolation of the 1-D ones.
__global__ __launch_bounds__(256) void track_posemb_kernel(const float* tabx, const float* taby, int H, int W, int Ch,
                                                            const float* xy, long ldxy, float* out, long ldo, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * 2 * Ch) return;
    const int n = (int)(i / (2 * Ch)), c = (int)(i - (long)n * 2 * Ch);
    const Corners k = border_corners(xy[n * ldxy], xy[n * ldxy + 1], W, H);
    float v;
    if (c < Ch) v = tabx[(long)k.x0 * Ch + c] * (1.f - k.fx) + tabx[(long)k.x1 * Ch + c] * k.fx;
    else v = taby[(long)k.y0 * Ch + c - Ch] * (1.f - k.fy) + taby[(long)k.y1 * Ch + c - Ch] * k.fy;
    out[(long)n * ldo + c] = v;
}

// transformer input row (n, s): [flow embedding 2 E | fx/ms fy/ms fx/ms fy/ms | corr[n][s][Cc] | feats[n][s][Cf]]
//                               + pos[n] + ref[s > 0]
struct TokParams {
    const float* coords;      // [N][S][2]
    const float* corr; long ldc; int Cc;
    const float* feats; long ldf; int Cf;
    const float* pos; long ldp;           // [N][D]
    const float* ref;                     // [2][D]
    float* out; long ldo;
    int N, S, E, D;
    float inv_max_scale;
};

__global__ __launch_bounds__(128) void track_tokens_kernel(const TokParams p) {
    const long ns = blockIdx.x;
    const int n = (int)(ns / p.S), s = (int)(ns - (long)n * p.S);
    const float fx = p.coords[2 * ns] - p.coords[2 * (long)n * p.S];
    const float fy = p.coords[2 * ns + 1] - p.coords[2 * (long)n * p.S + 1];
    const float* ref = p.ref + (s > 0 ? p.D : 0);
    const float step = 1000.0f / (float)p.E;
    for (int c = threadIdx.x; c < p.D; c += 128) {
        float v;
        if (c < 2 * p.E) {
            const int cc = c < p.E ? c : c - p.E;
            const float a = (c < p.E ? fx : fy) * ((float)(cc & ~1) * step);     // div_term = arange(0, E, 2) * (1000 / E)
            v = (cc & 1) ? cosf(a) : sinf(a);
        } else if (c < 2 * p.E + 4) {
            v = (((c - 2 * p.E) & 1) ? fy : fx) * p.inv_max_scale;
        } else if (c < 2 * p.E + 4 + p.Cc) {
            v = p.corr[ns * p.ldc + c - (2 * p.E + 4)];
        } else {
            v = p.feats[ns * p.ldf + c - (2 * p.E + 4 + p.Cc)];
        }
        p.out[ns * p.ldo + c] = v + p.pos[(long)n * p.ldp + c] + ref[c];
    }
}

// coords[n][s] += delta[n][s][0:2] for s > 0 (frame 0 stays the query); pred[s][n] = coords * stride
__global__ __launch_bounds__(256) void track_update_kernel(float* coords, const float* delta, long ldd, float* pred, int N,
                                                            int S, float stride) {
    const long ns = (long)blockIdx.x * 256 + threadIdx.x;
    if (ns >= (long)N * S) return;
    const int n = (int)(ns / S), s = (int)(ns - (long)n * S);
    float x = coords[2 * ns], y = coords[2 * ns + 1];
    if (s > 0) {
        x += delta[ns * ldd];
        y += delta[ns * ldd + 1];
        coords[2 * ns] = x;
        coords[2 * ns + 1] = y;
    }
    pred[2 * ((long)s * N + n)] = x * stride;
    pred[2 * ((long)s * N + n) + 1] = y * stride;
}

}  // namespace

extern "C" int iggt_layernorm_rows_f32(const float* x, long ldx, const float* x2, long ldx2, const float* w, const float* b,
                                       float* out, long ldo, int rows, int C, float eps, void* stream) {
    if (rows <= 0 || C <= 0 || ldx < C || ldo < C || (x2 && ldx2 < C)) return -1;
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, x2, ldx2, w, b,
                       out, ldo, rows, C, eps);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_avgpool2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if (N <= 0 || H < 2 || W < 2 || C <= 0 || (C % 4)) return -1;
    if (((uintptr_t)x | (uintptr_t)y) % 16) return -2;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(avgpool2_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, N,
                       H, W, C / 4);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_sample_points_nhwc_f32(const float* feat, int H, int W, int C, const float* xy, long ldxy, float* out,
                                           long ldo, int N, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || ldxy < 2 || ldo < C || (ldo % 4)) return -1;
    if (((uintptr_t)feat | (uintptr_t)out) % 16) return -2;
    const long total = (long)N * (C / 4);
    hipLaunchKernelGGL(sample_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat, H,
                       W, C / 4, xy, ldxy, out, ldo, N);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_track_corr_f32(const float* const* fmaps, const int* Hs, const int* Ws, int levels, int S, int C,
                                   const float* feats, const float* coords, int N, int radius, float* out, long ldo,
                                   void* stream) {
    if (levels <= 0 || levels > TRK_MAX_LEVELS || S <= 0 || N <= 0) return -1;
    if (C != 128 || radius != 4) return -3;          // the one configuration IGGT builds (track_head.py:18-28)
    if (ldo < (long)levels * 81) return -2;
    CorrParams p;
    for (int l = 0; l < levels; ++l) {
        if (Hs[l] <= 0 || Ws[l] <= 0 || ((uintptr_t)fmaps[l] % 16)) return -2;
        p.fmap[l] = fmaps[l]; p.H[l] = Hs[l]; p.W[l] = Ws[l];
    }
    if ((uintptr_t)feats % 16) return -2;
    p.feats = feats; p.coords = coords; p.out = out; p.ldo = ldo; p.N = N; p.S = S; p.levels = levels;
    p.inv_sqrt_c = 1.0f / sqrtf((float)C);
    const long waves = (long)N * S * levels;
    hipLaunchKernelGGL(track_corr_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_track_posemb_f32(const float* tabx, const float* taby, int H, int W, int Ch, const float* xy, long ldxy,
                                     float* out, long ldo, int N, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Ch <= 0 || ldxy < 2 || ldo < 2 * Ch) return -1;
    const long total = (long)N * 2 * Ch;
    hipLaunchKernelGGL(track_posemb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tabx, taby,
                       H, W, Ch, xy, ldxy, out, ldo, N);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_track_tokens_f32(const float* coords, const float* corr, long ldc, int Cc, const float* feats, long ldf,
                                     int Cf, const float* pos, long ldp, const float* ref, float* out, long ldo, int N, int S,
                                     int E, float max_scale, void* stream) {
    if (N <= 0 || S <= 0 || E <= 0 || (E & 1) || Cc <= 0 || Cf <= 0 || max_scale <= 0.f) return -1;
    const int D = 2 * E + 4 + Cc + Cf;
    if (ldc < Cc || ldf < Cf || ldp < D || ldo < D) return -2;
    TokParams p{coords, corr, ldc, Cc, feats, ldf, Cf, pos, ldp, ref, out, ldo, N, S, E, D, 1.0f / max_scale};
    hipLaunchKernelGGL(track_tokens_kernel, dim3((unsigned)((long)N * S)), dim3(128), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_track_update_f32(float* coords, const float* delta, long ldd, float* pred, int N, int S, float stride,
                                     void* stream) {
    if (N <= 0 || S <= 0 || ldd < 2) return -1;
    const long total = (long)N * S;
    hipLaunchKernelGGL(track_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coords,
                       delta, ldd, pred, N, S, stride);
    IGGT_CHECK_LAUNCH();
    return 0;
}
