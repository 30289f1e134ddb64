// Fused tail of the DPT heads: bilinear upsample (align_corners) + position map + conv3x3(128 -> 32) + ReLU +
// conv1x1(32 -> Cout) + activate_head, from the 128-channel map at half resolution to the per-pixel outputs.
//
// Reference (file:line): custom_interpolate to (H, W) iggt/heads/dpt_head.py:251-256,484-509; _apply_pos_embed
// 274-284; scratch.output_conv2 = [Conv2d(128, 32, 3, 1, 1), ReLU, Conv2d(32, Cout, 1)] 121-128; activate_head
// iggt/heads/head_act.py:61-125.  Unfused this stage costs three passes over full-resolution maps (write + read of
// the 128-channel 518^2 map: 8.8 GB per 32 views, then a 32-channel one) and a 3x3 convolution whose loader re-reads
// and re-splits every activation once per tap while feeding only 32 output channels: 7.3 ms per 32-view head pass
// (resize 2.07 + conv 4.9 + tail 0.3) where the MFMAs need 0.9 ms.
//
// Here one workgroup (4 waves) owns a 4 x 32 tile of OUTPUT pixels.  Per 32-channel slice of the input it builds the
// 6 x 34 halo of the *upsampled* map directly in LDS -- bilinear sample of the half-resolution map + position row/col
// value, split into bf16 hi + lo (fp32-grade products, 3 MFMAs per term as in conv_igemm.hip), zero outside the image
// (the convolution's padding) -- together with that slice's weights for all 9 taps, then runs the 9 taps out of LDS
// (a tap is a row/column shift of the halo: wave w reads LDS rows (w + ky) * 36 + kx + 0..31); the weights of a slice
// are staged in two tap groups so that 48 KB of LDS (three workgroups per CU) suffice.  Each sample is interpolated and split once instead of nine times, the full-resolution 128- and
// 32-channel maps never exist in HBM, and the epilogue applies bias + ReLU, the 1x1 convolution and the activations
// and writes 8-16 bytes per pixel.
#include "common.h"
#include "../../include/iggt_hip.h"

namespace {

constexpr int TH = 4, TW = 32;           // output tile
constexpr int HH = TH + 2, HWD = TW + 2;   // halo 6 x 34
constexpr int HP = 36;                   // LDS pitch of a halo row (pixels)
constexpr int A_PLANE = HH * HP * 64;    // 13 824 B: [halo pixel][32 ch] bf16
constexpr int W_TAPS = 5;                // taps staged at a time (5 + 4): 48 KB of LDS -> three workgroups per CU
constexpr int W_PLANE = W_TAPS * 32 * 64;  // 10 240 B: [(tap - first) * 32 + cout][32 ch] bf16
constexpr int SMEM_BYTES = 2 * A_PLANE + 2 * W_PLANE;  // 48 128 B
constexpr int ST_PITCH = 33;             // epilogue tile [128 px][32 ch] fp32, padded

struct TailParams {
    const float* x;      // [N][Hi][Wi][128]
    const float* xpart;  // [Wo][64] or null: added to channels 0..63
    const float* ypart;  // [Ho][64] or null: added to channels 64..127
    const bf16_t* w_hi;  // [32][9 * 128] tap-major
    const bf16_t* w_lo;
    const float* b1;     // [32]
    const float* w2;     // [Cout][32]
    const float* b2;     // [Cout]
    float* pts;          // [N][Ho][Wo][Cout - 1]
    float* conf;         // [N][Ho][Wo]
    int N, Hi, Wi, Ho, Wo, Cout, act, conf_act;
    float sy, sx;
    int tiles_x, tiles_y;
    int nchw;            // round 6: all Cout channels, un-activated last one included, as planes pts[N][Cout][Ho][Wo] (part_feat)
};

IGGT_DEVINL int swz(int row, int slot) { return row * 64 + (((slot ^ (row >> 2) ^ ((row >> 1) & 1)) & 3) << 4); }

__global__ __launch_bounds__(256, 2) void dpt_tail_kernel(const TailParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sAh = smem;
    char* sAl = smem + A_PLANE;
    char* sWh = smem + 2 * A_PLANE;
    char* sWl = sWh + W_PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;

    int t = blockIdx.x;
    const int tx0 = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty0 = t % p.tiles_y;
    const int img = t / p.tiles_y;
    const int oy0 = ty0 * TH, ox0 = tx0 * TW;
    const float* xin = p.x + (long)img * p.Hi * p.Wi * 128;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
    }

#pragma unroll 1
    for (int cc = 0; cc < 4; ++cc) {
        if (cc > 0) __syncthreads();  // every wave is done with the previous slice's halo and weights
        // ---- halo of the upsampled map, channels [32 cc, 32 cc + 32): thread -> (halo pixel, 16-channel half) --------
        // (Variants that issue all loads of the slice up front, or map 8 lanes to one pixel's full 128-byte lines, need
        //  240-256 VGPRs and measured the same end-to-end time; this one needs 122.)
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int ph = pass * 256 + tid;
            if (ph < 2 * HH * HWD) {
                const int px = ph >> 1, half = ph & 1;
                const int hy = px / HWD, hx = px - hy * HWD;
                const int oy = oy0 - 1 + hy, ox = ox0 - 1 + hx;
                const bool ok = oy >= 0 && oy < p.Ho && ox >= 0 && ox < p.Wo;
                u32x4 h[2], l[2];
                if (ok) {
                    // PyTorch upsample_bilinear2d (align_corners): src = dst * (in-1)/(out-1), lambda from the floor
                    const float fy = oy * p.sy, fx = ox * p.sx;
                    const int y0 = (int)fy, x0 = (int)fx;
                    const int y1 = y0 + (y0 < p.Hi - 1 ? 1 : 0), x1 = x0 + (x0 < p.Wi - 1 ? 1 : 0);
                    const float ly = fy - y0, lx = fx - x0;
                    const float hyw = 1.f - ly, hxw = 1.f - lx;
                    const int c = cc * 32 + half * 16;
                    const float* b00 = xin + ((long)y0 * p.Wi + x0) * 128 + c;
                    const float* b01 = xin + ((long)y0 * p.Wi + x1) * 128 + c;
                    const float* b10 = xin + ((long)y1 * p.Wi + x0) * 128 + c;
                    const float* b11 = xin + ((long)y1 * p.Wi + x1) * 128 + c;
                    const float* pe = nullptr;
                    if (p.xpart) pe = (c < 64) ? (p.xpart + (long)ox * 64 + c) : (p.ypart + (long)oy * 64 + (c - 64));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 v00 = *reinterpret_cast<const f32x4*>(b00 + 4 * i);
                        const f32x4 v01 = *reinterpret_cast<const f32x4*>(b01 + 4 * i);
                        const f32x4 v10 = *reinterpret_cast<const f32x4*>(b10 + 4 * i);
                        const f32x4 v11 = *reinterpret_cast<const f32x4*>(b11 + 4 * i);
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[e] = hyw * (hxw * v00[e] + lx * v01[e]) + ly * (hxw * v10[e] + lx * v11[e]);
                        if (pe) {
                            const f32x4 a = *reinterpret_cast<const f32x4*>(pe + 4 * i);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += a[e];
                        }
                        const uint32_t h0 = pack_bf16x2(o[0], o[1]), h1 = pack_bf16x2(o[2], o[3]);
                        h[i >> 1][2 * (i & 1)] = h0;
                        h[i >> 1][2 * (i & 1) + 1] = h1;
                        l[i >> 1][2 * (i & 1)] = pack_bf16x2(o[0] - bf16_lo(h0), o[1] - bf16_hi(h0));
                        l[i >> 1][2 * (i & 1) + 1] = pack_bf16x2(o[2] - bf16_lo(h1), o[3] - bf16_hi(h1));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        h[i] = u32x4{0u, 0u, 0u, 0u};
                        l[i] = u32x4{0u, 0u, 0u, 0u};
                    }
                }
                const int row = hy * HP + hx;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int off = swz(row, half * 2 + i);
                    *reinterpret_cast<u32x4*>(sAh + off) = h[i];
                    *reinterpret_cast<u32x4*>(sAl + off) = l[i];
                }
            }
        }
        // ---- weights + taps in two groups (taps 0-4, then 5-8): LDS row = (tap - first) * 32 + cout --------------------
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            const int first = grp * W_TAPS, ntap = grp == 0 ? W_TAPS : 9 - W_TAPS;
            if (grp == 1) __syncthreads();   // taps 0-4 are done with the weight buffer
#pragma unroll 1
            for (int pass = 0; pass < 3; ++pass) {
                const int pi = pass * 256 + tid;
                if (pi < ntap * 32 * 4) {
                    const int row = pi >> 2, piece = pi & 3;
                    const int co = row & 31, tap = first + (row >> 5);
                    const long src = (long)co * 1152 + tap * 128 + cc * 32 + piece * 8;
                    const int off = swz(row, piece);
                    *reinterpret_cast<u32x4*>(sWh + off) = *reinterpret_cast<const u32x4*>(p.w_hi + src);
                    *reinterpret_cast<u32x4*>(sWl + off) = *reinterpret_cast<const u32x4*>(p.w_lo + src);
                }
            }
            __syncthreads();
            // wave = tile row; two accumulators alternate to halve the dependent-MFMA chain
#pragma unroll
            for (int tt = 0; tt < W_TAPS; ++tt) {
                if (tt >= ntap) break;
                const int tap = first + tt;
                const int ky = tap / 3, kx = tap - ky * 3;
                const int arow = (wave + ky) * HP + kx + frow;
                const int wrow = tt * 32 + frow;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int aoff = swz(arow, 2 * kc + fhalf), woff = swz(wrow, 2 * kc + fhalf);
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sAh + aoff);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(sAl + aoff);
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sWh + woff);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(sWl + woff);
                    if (tap & 1) {
                        acc1 = mfma32(al, bh, acc1);
                        acc1 = mfma32(ah, bl, acc1);
                        acc1 = mfma32(ah, bh, acc1);
                    } else {
                        acc0 = mfma32(al, bh, acc0);
                        acc0 = mfma32(ah, bl, acc0);
                        acc0 = mfma32(ah, bh, acc0);
                    }
                }
            }
        }
    }

    // ---- epilogue: bias + ReLU -> LDS [pixel][32 ch] -> 1x1 conv + activations, <= 16 B per pixel to HBM --------------
    __syncthreads();
    float* stile = reinterpret_cast<float*>(smem);
    {
        const float b1 = p.b1[lane & 31];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            stile[(wave * 32 + mfma32_row(r, lane)) * ST_PITCH + (lane & 31)] = fmaxf(acc0[r] + acc1[r] + b1, 0.f);
    }
    __syncthreads();
    const int pxl = tid >> 1, oh = tid & 1;           // tile pixel, output-channel parity
    const int ty = pxl >> 5, txx = pxl & 31;
    const int oy = oy0 + ty, ox = ox0 + txx;
    if (oy < p.Ho && ox < p.Wo) {
        const float* row = stile + pxl * ST_PITCH;
        const long pix = ((long)img * p.Ho + oy) * p.Wo + ox;
        for (int o = oh; o < p.Cout; o += 2) {
            const float* w = p.w2 + o * 32;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int c = 0; c < 32; c += 4) {
                s0 = fmaf(row[c], w[c], s0);
                s1 = fmaf(row[c + 1], w[c + 1], s1);
                s2 = fmaf(row[c + 2], w[c + 2], s2);
                s3 = fmaf(row[c + 3], w[c + 3], s3);
            }
            const float v = (s0 + s1) + (s2 + s3) + p.b2[o];
            if (p.nchw) {      // the part head's tail (reference part_head.py:240-243: no activation), NCHW like the reference's output
                p.pts[(((long)img * p.Cout + o) * p.Ho + oy) * p.Wo + ox] = v;
            } else if (o < p.Cout - 1) {
                float r = v;
                if (p.act == 1) r = expf(v);
                else if (p.act == 2) r = fmaxf(v, 0.f);
                else if (p.act == 3) r = copysignf(expm1f(fabsf(v)), v);
                else if (p.act == 4) r = 1.0f / (1.0f + expf(-v));
                p.pts[pix * (p.Cout - 1) + o] = r;
            } else {
                float r;
                if (p.conf_act == 0) r = 1.0f + expf(v);
                else if (p.conf_act == 1) r = expf(v);
                else r = 1.0f / (1.0f + expf(-v));
                p.conf[pix] = r;
            }
        }
    }
}

}  // namespace

extern "C" int iggt_dpt_tail_f32(const float* x, int N, int Hi, int Wi, int Ho, int Wo, const float* xpart,
                                 const float* ypart, const void* w_hi, const void* w_lo, const float* b1,
                                 const float* w2, const float* b2, float* pts, float* conf, int Cout, int act,
                                 int conf_act, int out_nchw, void* stream) {
    if (N <= 0 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || Cout < 2 || Cout > 8) return -1;
    if (act < 0 || act > 4 || conf_act < 0 || conf_act > 2) return -2;   // "norm" needs all channels: not fused
    if (!pts || (!out_nchw && !conf)) return -1;
    if ((xpart == nullptr) != (ypart == nullptr)) return -3;
    TailParams p;
    p.x = x; p.xpart = xpart; p.ypart = ypart; p.w_hi = (const bf16_t*)w_hi; p.w_lo = (const bf16_t*)w_lo;
    p.b1 = b1; p.w2 = w2; p.b2 = b2; p.pts = pts; p.conf = conf;
    p.N = N; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.act = act; p.conf_act = conf_act; p.nchw = out_nchw;
    p.sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    p.sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    p.tiles_x = (Wo + TW - 1) / TW;
    p.tiles_y = (Ho + TH - 1) / TH;
    const long blocks = (long)N * p.tiles_y * p.tiles_x;
    if (blocks >= (1L << 31)) return -4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)dpt_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(dpt_tail_kernel, dim3((unsigned)blocks), dim3(256), SMEM_BYTES, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}
