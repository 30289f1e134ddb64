// Standalone hardware probe (gfx950): verifies the MFMA fragment layouts and the
// ds_read_b64_tr_b16 semantics that csrc/attention.hip and csrc/gemm_bf16.hip assume.
//   hipcc --offload-arch=gfx950 -O2 probes/mfma_probe.hip -o probes/mfma_probe.out && ./probes/mfma_probe.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mfma_probe(const float* A, const float* B, float* C) {  // A[32][16], B[16][32], C[32][32]
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = (__bf16)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

__global__ void tr_probe(float* out) {  // out[64][4]
    __shared__ __attribute__((aligned(16))) __bf16 lds[256];
    const int l = threadIdx.x;
    for (int e = 0; e < 4; ++e) lds[4 * l + e] = (__bf16)(float)(4 * l + e);
    __syncthreads();
    typedef __attribute__((address_space(3))) short4v lds_s4;
    short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + 4 * l));
    for (int e = 0; e < 4; ++e) {
        unsigned u = ((unsigned)(unsigned short)t[e]) << 16;
        out[l * 4 + e] = __builtin_bit_cast(float, u);
    }
}

int main() {
    std::vector<float> A(512), B(512), C(1024), R(1024);
    srand(1);
    for (auto& x : A) x = (float)(rand() % 7 - 3);
    for (auto& x : B) x = (float)(rand() % 5 - 2);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j];
            R[i * 32 + j] = s;
        }
    float *dA, *dB, *dC, *dT;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096); hipMalloc(&dT, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += (C[i] != R[i]);
    printf("MFMA_32x32x16_bf16 layout assumption: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    if (bad) {
        printf("C (device, assumed layout) row 0..3:\n");
        for (int i = 0; i < 4; ++i) { for (int j = 0; j < 32; ++j) printf("%g ", C[i * 32 + j]); printf("\n"); }
        printf("R (host) row 0..3:\n");
        for (int i = 0; i < 4; ++i) { for (int j = 0; j < 32; ++j) printf("%g ", R[i * 32 + j]); printf("\n"); }
    }
    std::vector<float> T(256);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, dT);
    hipMemcpy(T.data(), dT, 1024, hipMemcpyDeviceToHost);
    // assumed: within each 16-lane group g, lane i element j = lds[64 g + 16 j + i]
    int tbad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) tbad += (T[l * 4 + j] != (float)(64 * (l >> 4) + 16 * j + (l & 15)));
    printf("ds_read_b64_tr_b16 assumption: %s (%d mismatches)\n", tbad ? "WRONG" : "OK", tbad);
    printf("tr16 table (lane: e0 e1 e2 e3), LDS[e]=e, lane address = element 4*lane:\n");
    for (int l = 0; l < 64; ++l) printf("%2d: %3g %3g %3g %3g\n", l, T[l * 4], T[l * 4 + 1], T[l * 4 + 2], T[l * 4 + 3]);
    return (bad || tbad) ? 1 : 0;
}
