// Image preprocessing on the GPU: the step in front of the forward path (reference iggt/utils/load_fn.py:12-128).
//
// The reference resizes every decoded image with PIL (Image.resize(..., BICUBIC), load_fn.py:85) and converts it with
// torchvision's ToTensor (uint8 HWC -> float CHW / 255), one image after the other on one host core.  Pillow's 8-bit
// resampler is pure integer arithmetic: a horizontal pass then a vertical pass (src/libImaging/Resample.c), each output
// byte = clip8((2^21 + sum_x in[x] * kk[x]) >> 22) with int32 coefficients kk = round(k * 2^22) of the normalised bicubic
// (a = -0.5) kernel whose support is stretched by the down-scaling factor, and the intermediate image is rounded to
// uint8 again.  These kernels reproduce that arithmetic BIT FOR BIT (tests/test_preprocess_gpu.py compares with PIL);
// the coefficient tables (a few KB) are built on the host in double precision exactly as precompute_coeffs() does
// (iggt_official_amd/utils/load_fn.py).
#include "common.h"
#include "../../include/iggt_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

IGGT_DEVINL uint8_t clip8(int v) {
    v >>= PRECISION_BITS;   // arithmetic shift, as Pillow's lookup index
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out[y][xx][c] = clip8(half + sum_{x < n} in[y][x0 + x][c] * kk[xx][x]),  (x0, n) = bounds[xx]
__global__ __launch_bounds__(256) void resample_h_u8_kernel(const uint8_t* in, int Hi, int Wi, const int* bounds,
                                                             const int* kk, int ksize, uint8_t* out, int Wo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Hi * Wo) return;
    const int y = (int)(idx / Wo), xx = (int)(idx - (long)y * Wo);
    const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (long)xx * ksize;
    const uint8_t* row = in + ((long)y * Wi + x0) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int kv = k[x];
        s0 += row[3 * x] * kv;
        s1 += row[3 * x + 1] * kv;
        s2 += row[3 * x + 2] * kv;
    }
    uint8_t* o = out + idx * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// out[yy][x][c] = clip8(half + sum_{y < n} in[y0 + y][x][c] * kk[yy][y])
__global__ __launch_bounds__(256) void resample_v_u8_kernel(const uint8_t* in, int W, const int* bounds, const int* kk,
                                                             int ksize, uint8_t* out, int Ho) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Ho * W) return;
    const int yy = (int)(idx / W), x = (int)(idx - (long)yy * W);
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = kk + (long)yy * ksize;
    const uint8_t* col = in + ((long)y0 * W + x) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
        const int kv = k[y];
        const uint8_t* px = col + (long)y * W * 3;
        s0 += px[0] * kv;
        s1 += px[1] * kv;
        s2 += px[2] * kv;
    }
    uint8_t* o = out + idx * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// ToTensor + crop + constant padding: dst[c][y][x] = inside ? src[y - py + cy][x - px + cx][c] / 255 : pad
__global__ __launch_bounds__(256) void u8hwc_to_f32chw_kernel(const uint8_t* src, int Hs, int Ws, float* dst, int Hd, int Wd,
                                                               int cy, int cx, int py, int px, int h, int w, float pad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Hd * Wd) return;
    const int y = (int)(idx / Wd), x = (int)(idx - (long)y * Wd);
    const int sy = y - py, sx = x - px;
    const bool inside = sy >= 0 && sy < h && sx >= 0 && sx < w;
    const long hw = (long)Hd * Wd;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = pad;
        if (inside) v = (float)src[((long)(sy + cy) * Ws + (sx + cx)) * 3 + c] / 255.0f;   // IEEE division, as torch's .div(255)
        dst[c * hw + idx] = v;
    }
}

}  // namespace

extern "C" int iggt_resize_bicubic_u8(const void* in, int Hi, int Wi, const int* hbounds, const int* hkk, int hksize,
                                      const int* vbounds, const int* vkk, int vksize, void* tmp, void* out, int Ho, int Wo,
                                      void* stream) {
    if (Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || hksize <= 0 || vksize <= 0) return -1;
    const long n1 = (long)Hi * Wo, n2 = (long)Ho * Wo;
    hipLaunchKernelGGL(resample_h_u8_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)in, Hi, Wi, hbounds, hkk, hksize, (uint8_t*)tmp, Wo);
    IGGT_CHECK_LAUNCH();
    hipLaunchKernelGGL(resample_v_u8_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)tmp, Wo, vbounds, vkk, vksize, (uint8_t*)out, Ho);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_u8hwc_to_f32chw(const void* src, int Hs, int Ws, float* dst, int Hd, int Wd, int crop_y, int crop_x,
                                    int pad_y, int pad_x, int h, int w, float pad_value, void* stream) {
    if (Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || h < 0 || w < 0) return -1;
    if (crop_y < 0 || crop_x < 0 || crop_y + h > Hs || crop_x + w > Ws) return -2;
    const long n = (long)Hd * Wd;
    hipLaunchKernelGGL(u8hwc_to_f32chw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)src, Hs, Ws, dst, Hd, Wd, crop_y, crop_x, pad_y, pad_x, h, w, pad_value);
    IGGT_CHECK_LAUNCH();
    return 0;
}
