// 256x256 bf16 MFMA GEMM with a 4-stage LDS-DMA pipeline -- the large-M path of iggt_gemm_bf16.
//
// Why a second GEMM kernel: arithmetic on the 128^2 register-staged kernel (gemm_bf16.hip) shows the LDS and
// the per-CU vector-memory path, not the matrix pipe, are the limiters -- per K-step a 128^2 tile writes 32 KB
// through ds_write_b128 (~79 B/clk/CU) and pulls only 64 FLOP per byte from L2.  This kernel
//   * quadruples the tile (256x256, 8 waves as 2(M) x 4(N), 128x64 per wave): 128 FLOP per staged byte, LDS bytes
//     moved per MFMA drop from 1.5 KB to 0.625 KB;
//   * stages operands with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip, no ds_write).  The LDS
//     destination of that instruction is wave-uniform base + lane*16, so the XOR swizzle of the 64-byte row image
//     is applied to the per-lane *source* address instead (guide rule 21): lane l of 1-KiB chunk j fills row
//     16j + l/4, 16-B slot l%4 with source piece (l%4) ^ ((row >> 2) & 3) -- conflict-free ds_read_b128;
//   * runs a 4-deep ring of 32-wide K stages (4 x 32 KiB = 128 KiB LDS, one workgroup per CU, 2 waves/SIMD) with
//     three stages of DMA in flight: counted `s_waitcnt vmcnt(N)` + raw `s_barrier` (a __syncthreads() would
//     drain the DMA queue every step -- guide section 5 "Pipelining across barriers").  One barrier per stage:
//         wait(stage kt landed) ; barrier ; issue DMA(stage kt+3 -> buffer of stage kt-1) ; 16 MFMA on stage kt
// Same epilogue contract as gemm_bf16.hip (specialised per mode to keep register allocation clean).
#include "common.h"
#include "gemm_common.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 32, NSTAGE = 4;
constexpr int OP_BYTES = TM * TK * 2;       // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OP_BYTES;   // A + W = 32 KiB

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int N>
IGGT_DEVINL void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t256_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, each 128 x 64
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;

    // ---- DMA map: an operand stage is 16 chunks of 1 KiB (16 rows x 64 B); wave w moves chunks 2w, 2w+1 -------
    const int c_row = lane >> 2, c_pos = lane & 3;
    int a_off[2], w_off[2];  // element offsets (< 2^31, checked by the launcher)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 16 + c_row;          // tile row 0..255
        const int piece = c_pos ^ ((r >> 2) & 3);           // source 16-B piece that belongs at LDS slot c_pos
        int ra = m0 + r;
        ra = ra < p.M ? ra : p.M - 1;
        int rw = n0 + r;
        rw = rw < p.N ? rw : p.N - 1;
        a_off[i] = ra * (int)p.lda + piece * 8;
        w_off[i] = rw * (int)p.ldw + piece * 8;
    }
    auto dma = [&](int kt) {
        char* sA = smem + (kt & (NSTAGE - 1)) * STAGE_BYTES;
        char* sW = sA + OP_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int chunk = (2 * wave + i) * 1024;
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
    // LDS operand offsets: row = (multiple of 32) + frow -> swizzle key ((row >> 2) & 3) depends on the lane only
    int lane_off[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) lane_off[kc] = frow * 64 + ((((2 * kc + fhalf) ^ (frow >> 2)) & 3) << 4);
    const int a_base = wm * 128 * 64, w_base = OP_BYTES + wn * 64 * 64;

    // prologue: three stages in flight (4 DMA instructions per wave per stage)
    dma(0);
    if (KT > 1) dma(1);
    if (KT > 2) dma(2);
#pragma unroll 1
    for (int kt = 0; kt < KT; ++kt) {
        // stage kt must have landed: allow the DMAs of the (up to two) younger stages to stay in flight
        const int younger = (KT - 1 - kt) < 2 ? (KT - 1 - kt) : 2;
        if (younger == 2) wait_vmcnt<8>();
        else if (younger == 1) wait_vmcnt<4>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // publishes stage kt to all waves; everyone is done reading stage kt-1
        if (kt + 3 < KT) dma(kt + 3);   // refill the buffer stage kt-1 just vacated
        const char* st = smem + (kt & (NSTAGE - 1)) * STAGE_BYTES;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 32 * 64 + lane_off[kc]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(st + w_base + j * 32 * 64 + lane_off[kc]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        }
    }

    // ---- epilogue through LDS: the MFMA C layout gives every lane ONE column, i.e. 2/4-byte stores at a row
    //      stride -- 128 store instructions per lane, measured at >50 % of the tile time at K = 1024 (store-issue
    //      bound).  Instead each half of the tile (128 rows x 256 fp32 = 128 KiB, the whole ring) is transposed
    //      through LDS and leaves as fully coalesced 16-byte accesses: one wave instruction = one 1-KiB output row.
    wait_vmcnt<0>();
    __syncthreads();
    float* stile = reinterpret_cast<float*>(smem);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stile[(i * 32 + mfma32_row(r, lane)) * TN + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll 4
        for (int pass = 0; pass < 16; ++pass) {
            const int idx = pass * 512 + tid;
            const int row = idx >> 6, c4 = idx & 63;
            const int m = m0 + half * 128 + row;
            if (m < p.M) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(stile + row * TN + c4 * 4);
                gemm_epilogue_row4<MODE>(p, v, m, n0 + c4 * 4);
            }
        }
        __syncthreads();
    }
}

}  // namespace

int iggt_launch_gemm_t256(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    if ((long)p.M * p.lda >= (1L << 31) || (long)p.N * p.ldw >= (1L << 31)) return -100;
    if ((p.ldo % 4) != 0) return -100;  // 16-byte epilogue accesses
    p.tiles_n = (p.N + TN - 1) / TN;
    const int tiles_m = (p.M + TM - 1) / TM;
    const int lds = NSTAGE * STAGE_BYTES;  // 128 KiB
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
