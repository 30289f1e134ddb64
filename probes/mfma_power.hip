// Sustained rate of v_mfma_f32_32x32x16_{f16,bf16} from registers only (no LDS, no memory in the loop) as a function of the
// OPERAND DATA: zeros, a constant, or random values.  The arithmetic is identical; what changes is the switching activity
// of the multipliers, i.e. power -- and with it the clock the chip sustains.  Shows what "dense peak" means for real data.
//   hipcc --offload-arch=gfx950 -O3 -o probes/mfma_power probes/mfma_power.hip && probes/mfma_power
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int BF16>
__global__ __launch_bounds__(256) void mfma_loop(const u32x4* operands, float* sink, int iters) {
    // 4 A and 4 B fragments per lane, 8 independent accumulators: the matrix pipe is never waiting on a dependency
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = operands[(threadIdx.x * 8 + i) % 4096];
        b[i] = operands[(threadIdx.x * 8 + 4 + i) % 4096];
    }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (BF16)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 3]),
                                                                 __builtin_bit_cast(bf16x8, b[(i + (i >> 2)) & 3]), acc[i], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i & 3]),
                                                                __builtin_bit_cast(f16x8, b[(i + (i >> 2)) & 3]), acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 1.2345e-30f) sink[0] = s;
}

static uint16_t f2h(float f, int bf16) {
    if (bf16) {
        uint32_t u;
        memcpy(&u, &f, 4);
        return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
    _Float16 h = (_Float16)f;
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
}

int main() {
    const int iters = 20000, wgs = 256 * 8;
    u32x4* d_op;
    float* d_sink;
    hipMalloc(&d_op, 4096 * sizeof(u32x4));
    hipMalloc(&d_sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"zeros", "constant 1/64", "random N(0,1)/64", "random bit patterns"};
    for (int bf16 = 0; bf16 < 2; ++bf16)
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<uint16_t> h(4096 * 8);
            srand(1);
            for (auto& x : h) {
                float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
                float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
                if (mode == 0) x = 0;
                else if (mode == 1) x = f2h(1.f / 64, bf16);
                else if (mode == 2) x = f2h(g / 64, bf16);
                else x = f2h(ldexpf(g, (rand() % 9) - 8), bf16);   // random mantissas AND exponents (finite, |x| < 8)
            }
            hipMemcpy(d_op, h.data(), h.size() * 2, hipMemcpyHostToDevice);
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (bf16) hipLaunchKernelGGL(mfma_loop<1>, dim3(wgs), dim3(256), 0, 0, d_op, d_sink, iters);
                else hipLaunchKernelGGL(mfma_loop<0>, dim3(wgs), dim3(256), 0, 0, d_op, d_sink, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;   // first repetition warms up
            }
            const double flop = (double)wgs * 4 * iters * 8 * 32768.0;
            const double tf = flop / best / 1e9;
            // 1024 SIMDs x 1024 FLOP / clk at full rate
            printf("%-5s operands = %-22s %8.1f TFLOP/s   (= %.2f GHz x 1024 SIMDs x 1024 FLOP/clk)  %.1f ms\n", bf16 ? "bf16" : "fp16",
                   names[mode], tf, tf * 1e12 / (1024.0 * 1024.0) / 1e9, best);
        }
    return 0;
}
