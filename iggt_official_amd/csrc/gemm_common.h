// Shared parameter block + fused epilogue of the bf16 MFMA GEMM kernels (gemm_bf16.hip, gemm_bf16_t256.hip).
#pragma once
#include "common.h"

struct GemmParams {
    const bf16_t* A;
    const bf16_t* W;
    int M, N, K;
    long lda, ldw;
    int tiles_n;
    int tiles_m, group_m;  // big-tile kernel: tile order grouped over group_m row-tiles (0: n-fastest)
    // epilogue
    const float* bias;       // [N] or null
    const float* gamma;      // [N] or null
    const float* add_table;  // [rows_in][N] fp32 or null, indexed by (m % rows_in)
    float* out_f32;          // exactly one of out_f32 / out_bf16
    bf16_t* out_bf16;        // 16-bit output (bf16 or fp16 container, see the kernels' FMT)
    long ldo;
    int accumulate;  // out_f32 += value
    int act;         // 0 none, 1 exact GELU, 2 ReLU
    int rows_in, rows_out, row_off;  // output row remap (rows_in == 0: identity)
};

// Exact (erf) GELU, nn.GELU() of mlp.py:34.  erfc(|x|/sqrt2) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below
// the bf16 rounding of the fc1 output): 1 v_rcp + 1 v_exp + 9 FMA-class ops per element instead of libm erff's two
// divergent branches -- the erff epilogue was 27 % of the fc1 GEMM (0.62 -> 0.45 ms with the activation removed).
// The complementary form keeps the negative tail free of cancellation: gelu = x >= 0 ? x (1 - E/2) : x E/2.
IGGT_DEVINL float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float half_e = 0.5f * poly * t * __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
    return x >= 0.f ? fmaf(-x, half_e, x) : x * half_e;
}

// GELU from an LDS table (round 4, gemm_bf16_t256.hip; round 6 also the 192-row duo kernel): 2 048 intervals of
// Phi(x) = (1 + erf(x / sqrt 2)) / 2 over [-8, 8) as (value, forward difference) pairs, built per workgroup through the exact erff
// while its first DMA stages are in flight; gelu(x) = x (f_i + frac d_i): fma, med3, floor, sub, cvt, ds_read_b64, fma, mul.
// |error| <= 2.5e-6 absolute (tests/test_kernels_f16_gpu.py test_gemm_gelu_table_against_erf).
constexpr int GELU_LUT_N = 2048;
constexpr float GELU_LUT_SCALE = 128.f, GELU_LUT_OFF = 1024.f;
constexpr int GELU_LUT_BYTES = (GELU_LUT_N + 1) * 8;

IGGT_DEVINL float gelu_lut(float x, const float2* lut) {
    float t = fmaf(x, GELU_LUT_SCALE, GELU_LUT_OFF);
    t = __builtin_amdgcn_fmed3f(t, 0.f, 2047.996f);   // x <= -8: Phi = 6e-16; x >= 8: Phi = 1 to fp32
    const float fi = floorf(t);
    const float2 e = lut[(int)fi];
    return x * fmaf(t - fi, e.y, e.x);
}

// every thread of an `nthreads`-thread workgroup fills its GELU_LUT_N / nthreads entries (+ the guard entry); the caller's next
// barrier publishes the table
IGGT_DEVINL void gelu_lut_build(float2* lut, int tid, int nthreads) {
    const int per = GELU_LUT_N / nthreads;
    float prev = 0.5f + 0.5f * erff((float)(tid * per - (int)GELU_LUT_OFF) * (1.0f / GELU_LUT_SCALE) * 0.70710678118654752440f);
    for (int e = 0; e < per; ++e) {
        const float x1 = (float)(tid * per + e + 1 - (int)GELU_LUT_OFF) * (1.0f / GELU_LUT_SCALE);
        const float next = 0.5f + 0.5f * erff(x1 * 0.70710678118654752440f);
        lut[tid * per + e] = make_float2(prev, next - prev);
        prev = next;
    }
    if (tid == 0) lut[GELU_LUT_N] = make_float2(1.0f, 0.0f);
}

// Epilogue of one 32x32 accumulator fragment whose top-left element is (m_base, n - (lane & 31)):
//   val = act(acc + bias[n]) * gamma[n] (+ add_table[m % rows_in][n]);  out[row(m)][n] (= | +=) val
// MODE 0: every feature (runtime flags).  Specialised modes keep the inlined code (and with it the register
// allocation of the big-tile kernel) small:  1 = bf16 out, bias + act;  2 = fp32 accumulate, bias + gamma;
// 3 = fp32 store, bias (+ row remap / additive table).
template <int MODE, int FMT>
IGGT_DEVINL void gemm_epilogue_tile(const GemmParams& p, const f32x16& acc, int m_base, int n, int lane, float bias,
                                    float gamma) {
    // bias / gamma of column n come from the caller, loaded once per column BEFORE the first store of the tile: a load placed
    // between the fragments' stores makes the compiler wait for every store issued so far (vmcnt counts both)
    if (n >= p.N) return;
    if (MODE == 0 && p.out_f32 && p.accumulate && p.rows_in == 0 && p.act == 0) {
        // residual accumulate: request the 16 old values before the first store -- as load / add / store per element every
        // load has to wait behind the previous (possibly aliasing) store and the epilogue runs at one access in flight
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m_base + mfma32_row(r, lane);
            m = m < p.M ? m : p.M - 1;
            old[r] = p.out_f32[(long)m * p.ldo + n];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_base + mfma32_row(r, lane);
            if (m < p.M) p.out_f32[(long)m * p.ldo + n] = fmaf(acc[r] + bias, gamma, old[r]);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m_base + mfma32_row(r, lane);
        if (m >= p.M) continue;
        float val = acc[r] + bias;
        if (MODE == 0 || MODE == 1) {
            if (p.act == 1) val = gelu_erf(val);
            else if (p.act == 2) val = fmaxf(val, 0.f);
        }
        if (MODE == 0 || MODE == 2) val *= gamma;
        long orow = m;
        if (MODE == 0 || MODE == 3) {
            if (p.rows_in > 0) {
                const int g = m / p.rows_in, w = m - g * p.rows_in;
                orow = (long)g * p.rows_out + p.row_off + w;
                if (p.add_table) val += p.add_table[(long)w * p.N + n];
            }
        }
        if (MODE == 1) {
            reinterpret_cast<uint16_t*>(p.out_bf16)[orow * p.ldo + n] = pack_h1<FMT>(val);
        } else if (MODE == 2) {
            float* dst = p.out_f32 + orow * p.ldo + n;
            *dst = *dst + val;
        } else if (MODE == 3) {
            p.out_f32[orow * p.ldo + n] = val;
        } else if (p.out_f32) {
            float* dst = p.out_f32 + orow * p.ldo + n;
            *dst = p.accumulate ? (*dst + val) : val;
        } else {
            reinterpret_cast<uint16_t*>(p.out_bf16)[orow * p.ldo + n] = pack_h1<FMT>(val);
        }
    }
}

// MODE 3 (fp32 store, optional row remap + additive table) with the bias already added by the caller
template <int MODE, int FMT>
IGGT_DEVINL void gemm_epilogue_row4_nobias(const GemmParams& p, f32x4 v, int m, int n) {
    long orow = m;
    if (p.rows_in > 0) {
        const int g = m / p.rows_in, w = m - g * p.rows_in;
        orow = (long)g * p.rows_out + p.row_off + w;
        if (p.add_table) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p.add_table + (long)w * p.N + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += t[e];
        }
    }
    *reinterpret_cast<f32x4*>(p.out_f32 + orow * p.ldo + n) = v;
}

// returns -100 when the parameter combination has no specialised big-tile kernel (caller falls back)
int iggt_launch_gemm_t256(const GemmParams& p, int fmt, hipStream_t stream);
// 256 x 128 tile, two workgroups per CU (gemm_bf16_duo.hip); same return convention
int iggt_launch_gemm_duo(const GemmParams& p, int fmt, int rows, hipStream_t stream);
