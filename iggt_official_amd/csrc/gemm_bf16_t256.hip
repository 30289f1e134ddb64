// 256x256x64 bf16 MFMA GEMM with LDS-DMA staging (global_load_lds) -- the large-M path of iggt_gemm_bf16.
//
// Why a second GEMM kernel: PMC/arithmetic on the 128^2 register-staged kernel (gemm_bf16.hip) shows the
// LDS pipe, not the matrix pipe, is the limiter -- per K-step a 128^2 tile writes 32 KB through
// ds_write_b128 (~79 B/clk/CU) and reads 64 KB for only 64 MFMAs.  This kernel
//   * quadruples the tile (256x256, 8 waves as 2(M) x 4(N), 128x64 per wave = 32 MFMA 32x32x16 per K-step
//     per wave): LDS bytes moved per MFMA drop from 1.5 KB to 0.625 KB;
//   * stages operands with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip, no ds_write);
//     the LDS destination of that instruction is wave-uniform base + lane*16, so the XOR swizzle of the
//     row image (common.h swz_off) is applied to the per-lane *source* address instead (guide rule 21):
//     lane l of 1-KiB chunk j fills row 8j + l/8, slot l%8 with global piece (l%8) ^ key(row);
//   * double-buffers the 64-KiB stage (128 KiB LDS, one workgroup per CU, 2 waves per SIMD); the next
//     stage's DMA is issued before the MFMA phase and drained by the single barrier per K-step.
// Same epilogue contract as gemm_bf16.hip.
#include "common.h"
#include "gemm_common.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int OP_BYTES = TM * TK * 2;       // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OP_BYTES;   // A + W

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t256_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, each 128 x 64
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;

    // ---- DMA map: wave w moves chunks j = 4w..4w+3 (8 rows each) of A and of W per stage -----------
    const int c_row = lane >> 3, c_pos = lane & 7;
    int a_off[4], w_off[4];  // element offsets (M*lda, N*ldw < 2^31 for every IGGT shape; checked by the launcher)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (4 * wave + i) * 8 + c_row;           // tile row 0..255
        const int piece = c_pos ^ ((r >> 1) & 7);           // source piece that belongs at LDS slot c_pos
        int ra = m0 + r;
        ra = ra < p.M ? ra : p.M - 1;
        int rw = n0 + r;
        rw = rw < p.N ? rw : p.N - 1;
        a_off[i] = ra * (int)p.lda + piece * 8;
        w_off[i] = rw * (int)p.ldw + piece * 8;
    }
    auto dma = [&](int kt, int buf) {
        char* sA = smem + buf * STAGE_BYTES;
        char* sW = sA + OP_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int chunk = (4 * wave + i) * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t*)(p.A + a_off[i] + kt * TK), (lptr_t*)(sA + chunk), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)(p.W + w_off[i] + kt * TK), (lptr_t*)(sW + chunk), 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const int KT = p.K / TK;
    // LDS operand offsets: row = (multiple of 32) + frow, so the swizzle key ((row >> 1) & 7) depends on the
    // lane only: 4 lane-dependent offsets (one per 16-wide k chunk), everything else is wave-uniform.
    int lane_off[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) lane_off[kc] = frow * 128 + ((((2 * kc + fhalf) ^ (frow >> 1)) & 7) << 4);
    const int a_base = wm * 128 * 128, w_base = OP_BYTES + wn * 64 * 128;
    dma(0, 0);
    __syncthreads();  // drains the DMA (vmcnt(0)) and publishes stage 0
#pragma unroll 1
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) dma(kt + 1, (kt + 1) & 1);
        const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 32 * 128 + lane_off[kc]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(st + w_base + j * 32 * 128 + lane_off[kc]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        }
        __syncthreads();  // all waves done reading stage kt; DMA of stage kt+1 landed (vmcnt(0) + barrier)
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i) gemm_epilogue_tile<MODE>(p, acc[i][j], m0 + wm * 128 + i * 32, n, lane);
    }
}

}  // namespace

int iggt_launch_gemm_t256(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    if ((long)p.M * p.lda >= (1L << 31) || (long)p.N * p.ldw >= (1L << 31)) return -6;
    p.tiles_n = (p.N + TN - 1) / TN;
    const int tiles_m = (p.M + TM - 1) / TM;
    const int lds = 2 * STAGE_BYTES;  // 128 KiB
    int mode;
    if (p.out_bf16 && !p.gamma && p.rows_in == 0) mode = 1;
    else if (p.out_f32 && p.accumulate && p.rows_in == 0 && p.act == 0) mode = 2;
    else if (p.out_f32 && !p.accumulate && p.act == 0 && !p.gamma) mode = 3;
    else return -100;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_t256_kernel<1>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16_t256_kernel<2>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16_t256_kernel<3>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const dim3 grid(tiles_m * p.tiles_n), block(512);
    if (mode == 1) hipLaunchKernelGGL(gemm_bf16_t256_kernel<1>, grid, block, lds, stream, p);
    else if (mode == 2) hipLaunchKernelGGL(gemm_bf16_t256_kernel<2>, grid, block, lds, stream, p);
    else hipLaunchKernelGGL(gemm_bf16_t256_kernel<3>, grid, block, lds, stream, p);
    return 0;
}
