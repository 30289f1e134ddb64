// 256x256 16-bit-operand (bf16 / fp16) MFMA GEMM with a 4-stage LDS-DMA ring and a ping-pong wave schedule -- the
// large-M path of iggt_gemm_bf16 / iggt_gemm_f16.
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
//     up to three stages of DMA in flight behind counted `s_waitcnt vmcnt(N)` + raw `s_barrier` (a __syncthreads()
//     would drain the DMA queue every step -- guide section 5 "Pipelining across barriers").
// History (profiles/r01_gemm_pmc.txt): a first version ran all 8 waves in lockstep, one barrier per stage (fc2 shape
// 745 TF/s); the ping-pong schedule below replaced it (799 TF/s).
// Same epilogue contract as gemm_bf16.hip (specialised per mode to keep register allocation clean).
#include <stdlib.h>

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

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong schedule.
//
// With all 8 waves in lockstep (one barrier per stage) the two waves that share a SIMD read LDS at the same time
// and then compete for the matrix pipe, which idles during every read phase.  Here the K loop is
// cut into phases of one 16-wide k-slice = 6 ds_read_b128 + 8 MFMA (256 matrix-pipe cycles) with TWO barriers per
// phase, and the wm = 1 waves run one barrier behind the wm = 0 waves:
//        barrier interval   2j            2j+1          2j+2
//        wm = 0 waves       read(j)       MFMA(j)       read(j+1)
//        wm = 1 waves       MFMA(j-1)     read(j)       MFMA(j)
// so on every SIMD one wave feeds the matrix pipe (at raised priority) while the other fetches its next fragments
// and issues its share of the DMA.  Stage s+3 is requested in two halves, in phases (s,1) and (s+1,0) -- two phases
// after the last read of the slot it overwrites -- and stage s+1 is retired by a counted vmcnt(6) in phase (s,1),
// one phase before its first read (guide: "read a staged buffer one phase AFTER the wait that retires it").
// DBG: ablation / placement switches (bit 0: no DMA in the loop, bit 1: no fragment reads, bit 2: L2-hot DMA source,
// bits 3-4: DMA placement).  Production = 16: the A piece of each half is issued in the read interval, the W piece
// between the MFMAs.  Measured at M=43968, N=1024, K=4096 (TF/s): lockstep kernel 745; ping-pong 789 / 799 / 799 for
// placement 0 / 1 / 2; L2-hot source 962; no DMA 1258; no fragment reads 808 -> the LDS-DMA path (issue + LDS write
// ~24 %, L2-miss latency ~18 %) is what separates this kernel from the matrix pipe, not the LDS reads.
//
// Round 4 -- GELU from an LDS table (template parameter LUT; fc1 only).  Counters on the shipping fc1 kernel
// (profiles/r03_gemm_pmc.txt) showed 7.0 vector instructions per MFMA, 5 of them the erf-GELU of the epilogue (1 v_rcp +
// 1 v_exp + ~12 FMA-class operations per element, the two transcendentals at quarter rate), ~19 us of a 46 us tile round with
// the matrix pipe idle.  The LDS pipe is idle in that phase (1 LDS instruction per MFMA over the kernel), and 32 KiB of the
// CU's 160 KiB are unused beside the 128 KiB ring: every workgroup builds, before its K loop, a 2 048-interval table of
// Phi(x) = (1 + erf(x / sqrt 2)) / 2 over [-8, 8) as (value, forward difference) pairs -- 4 entries per thread through the
// exact erff, no memory traffic -- and the epilogue evaluates gelu(x) = x (f_i + frac * d_i): fma, med3, floor, sub, cvt,
// ds_read_b64, fma, mul per element.  |error| <= 2.5e-6 absolute (1e-4 of the value in the negative tail, where fp16's own
// rounding is 4.9e-4), tests/test_kernels_f16_gpu.py gates it against torch's erf GELU at the kernel tolerance.
constexpr int LUT_BYTES = GELU_LUT_BYTES;   // table + helpers: gemm_common.h (shared with the 192-row duo kernel)

template <int MODE, int DBG, int FMT, bool LUT = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t256pp_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* lut = reinterpret_cast<float2*>(smem + NSTAGE * STAGE_BYTES);   // behind the ring (LUT builds only)
    const int wm = wave >> 2, wn = wave & 3;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    // Tile order inside an XCD's contiguous chunk: groups of group_m row-tiles, m fastest inside a group, so the ~32
    // tiles an XCD runs concurrently form a (group_m x 32/group_m) block instead of a (32/tiles_n x tiles_n) strip --
    // fewer distinct A / W slices per K step have to come from beyond the L2.
    int tm, tn;
    if (p.group_m > 1) {
        const int per_group = p.group_m * p.tiles_n;
        const int grp = v / per_group, in = v - grp * per_group;
        const int first = grp * p.group_m;
        const int gsz = (p.tiles_m - first) < p.group_m ? (p.tiles_m - first) : p.group_m;
        tn = in / gsz;
        tm = first + (in - tn * gsz);
    } else {
        tm = v / p.tiles_n;
        tn = v - tm * p.tiles_n;
    }
    const int m0 = tm * TM, n0 = tn * TN;

    const int c_row = lane >> 2, c_pos = lane & 3;
    int a_off[2], w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 16 + c_row;
        const int piece = c_pos ^ ((r >> 2) & 3);
        int ra = m0 + r;
        ra = ra < p.M ? ra : p.M - 1;
        int rw = n0 + r;
        rw = rw < p.N ? rw : p.N - 1;
        a_off[i] = ra * (int)p.lda + piece * 8;
        w_off[i] = rw * (int)p.ldw + piece * 8;
    }
    // half h of this wave's share of stage kt: one 1-KiB chunk of A and one of W
    auto dma_a = [&](int kt, int h) {
        const int ks = (DBG & 4) ? (kt & 1) : kt;  // DBG 4: same DMA instruction stream, L2-hot source
        __builtin_amdgcn_global_load_lds((gptr_t*)(p.A + a_off[h] + ks * TK),
                                         (lptr_t*)(smem + (kt & (NSTAGE - 1)) * STAGE_BYTES + (2 * wave + h) * 1024),
                                         16, 0, 0);
    };
    auto dma_w = [&](int kt, int h) {
        const int ks = (DBG & 4) ? (kt & 1) : kt;
        __builtin_amdgcn_global_load_lds(
            (gptr_t*)(p.W + w_off[h] + ks * TK),
            (lptr_t*)(smem + (kt & (NSTAGE - 1)) * STAGE_BYTES + OP_BYTES + (2 * wave + h) * 1024), 16, 0, 0);
    };
    auto dma_half = [&](int kt, int h) {
        dma_a(kt, h);
        dma_w(kt, h);
    };
    constexpr int PLACE = (DBG >> 3) & 3;  // 0: both DMAs in the read interval, 1: both among the MFMAs, 2: one each

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const int KT = p.K / TK;  // >= 4 (launcher)
    int lane_off[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) lane_off[kc] = frow * 64 + ((((2 * kc + fhalf) ^ (frow >> 2)) & 3) << 4);
    const int a_base = wm * 128 * 64, w_base = OP_BYTES + wn * 64 * 64;

    bf16x8 a[4], b[2];
    auto read6 = [&](const char* st, int kc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 32 * 64 + lane_off[kc]);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(st + w_base + j * 32 * 64 + lane_off[kc]);
    };
    auto mfma8 = [&](int kt_issue, int h, bool issue) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = q >> 1, j = q & 1;
            acc[i][j] = mfma32h<FMT>(a[i], b[j], acc[i][j]);
            if (PLACE == 1 && q == 1 && issue) dma_a(kt_issue, h);
            if (PLACE != 0 && q == 4 && issue) dma_w(kt_issue, h);
            if (PLACE != 0 && (q == 1 || q == 4)) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: stages 0, 1 and the first half of stage 2 (10 DMA instructions per wave); stage 0 must land
    dma_half(0, 0);
    dma_half(0, 1);
    dma_half(1, 0);
    dma_half(1, 1);
    dma_half(2, 0);
    if constexpr (LUT) gelu_lut_build(lut, tid, 512);   // while the first stages are in flight: 4 table entries per thread
    wait_vmcnt<6>();
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // the wm = 1 waves run one barrier behind

#pragma unroll 1
    for (int s = 0; s < KT; ++s) {
        const char* st = smem + (s & (NSTAGE - 1)) * STAGE_BYTES;
        const bool i2 = !(DBG & 1) && s + 2 < KT, i3 = !(DBG & 1) && s + 3 < KT;
        // phase (s, 0): second half of stage s+2
        if (!(DBG & 2) || s == 0) read6(st, 0);
        if (i2) {
            if (PLACE == 0) dma_half(s + 2, 1);
            if (PLACE == 2) dma_a(s + 2, 1);
        }
        mfma8(s + 2, 1, i2);
        // phase (s, 1): first half of stage s+3, then retire stage s+1
        if (!(DBG & 2)) read6(st, 1);
        if (i3) {
            if (PLACE == 0) dma_half(s + 3, 0);
            if (PLACE == 2) dma_a(s + 3, 0);
        }
        // allowed in flight: stage s+2 (4) + what this phase has issued of stage s+3 so far (2 / 0 / 1)
        if (s + 3 < KT) wait_vmcnt<(PLACE == 0 ? 6 : PLACE == 1 ? 4 : 5)>();
        else if (s + 2 < KT) wait_vmcnt<4>();
        else wait_vmcnt<0>();
        mfma8(s + 3, 0, i3);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // even out the barrier count

    wait_vmcnt<0>();
    __syncthreads();
    if (DBG & 32) {   // ablation: no epilogue (keeps the accumulators alive through a never-true store)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 1.2345e-30f) p.A[0] == p.W[0] ? (void)0 : (void)(*reinterpret_cast<volatile float*>(smem) = sum);
        return;
    }
    // (Measured and dropped for the 16-bit output: the whole tile through LDS as 16-bit in ONE phase -- bias / activation in
    //  the accumulator layout, ds_write_b16, 16-byte stores: plain 46.3 / GELU 50.2 us per single-round tile vs 44.2 / 50.9 with
    //  the two fp32 half-tile phases below, and no difference at the full shapes: qkv 746 vs 747, fc1 677 vs 672 TF/s.)
    float* stile = reinterpret_cast<float*>(smem);
    const int c4 = tid & 63, r0 = tid >> 6;
    const int n = n0 + c4 * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, gamma4 = {1.f, 1.f, 1.f, 1.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    if (MODE == 2 && p.gamma) gamma4 = *reinterpret_cast<const f32x4*>(p.gamma + n);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        // fp32 accumulate (x += gamma * (acc + bias)): a thread's 16 row segments of this half (16 B each, row = 8 * pass +
        // tid / 64, same 4 columns) are read-modify-write.  ALL 16 old values are requested before the accumulators go through
        // LDS, so they are in flight during the transposition: as load -> add -> store per pass every load sat behind the
        // previous, possibly aliasing store (one access in flight per thread: proj 500 TF/s); in two groups of 8 behind the
        // barrier the second group still waited for the first group's stores (578).
        f32x4 old[MODE == 2 ? 16 : 1];
        if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + half * 128 + i * 8 + r0;
                const int mc = m < p.M ? m : p.M - 1;
                old[i] = *reinterpret_cast<const f32x4*>(p.out_f32 + (long)mc * p.ldo + n);
            }
        }
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
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) {
            const int row = pass * 8 + r0;
            const int m = m0 + half * 128 + row;
            f32x4 v4 = *reinterpret_cast<const f32x4*>(stile + row * TN + c4 * 4);
            if constexpr (MODE == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = fmaf(v4[e] + bias4[e], gamma4[e], old[pass][e]);
                if (m < p.M) *reinterpret_cast<f32x4*>(p.out_f32 + (long)m * p.ldo + n) = v4;
            } else if constexpr (MODE == 1) {
                // the bias is loaded ONCE per thread (its 32 row segments share their 4 columns).  Loaded inside the pass loop
                // the compiler put `s_waitcnt vmcnt(0)` behind it -- which also waits for every store issued so far: each of
                // the 32 passes of a tile then cost a memory round trip, 5 of the 12 us this epilogue took (qkv 662 -> 736 TF/s)
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] += bias4[e];
                if constexpr (LUT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = gelu_lut(v4[e], lut);
                } else if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = gelu_erf(v4[e]);
                } else if (p.act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = fmaxf(v4[e], 0.f);
                }
                u32x2 o;
                o[0] = pack_h2<FMT>(v4[0], v4[1]);
                o[1] = pack_h2<FMT>(v4[2], v4[3]);
                if (m < p.M) *reinterpret_cast<u32x2*>(p.out_bf16 + (long)m * p.ldo + n) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] += bias4[e];
                if (m < p.M) gemm_epilogue_row4_nobias<MODE, FMT>(p, v4, m, n);
            }
        }
        __syncthreads();
    }
}

}  // namespace

template <int FMT>
static int launch_t256(const GemmParams& p, int mode, int tiles_m, hipStream_t stream) {
    const int lds = NSTAGE * STAGE_BYTES;  // 128 KiB
    static int gelu_lut_on = -1;           // IGGT_GELU_LUT=0: the polynomial erfc epilogue (gemm_common.h gelu_erf)
    if (gelu_lut_on < 0) {
        const char* e = getenv("IGGT_GELU_LUT");
        gelu_lut_on = (e && atoi(e) == 0) ? 0 : 1;
    }
    // Production: DMA placement 2 (DBG = 16).  IGGT_GEMM_PPDBG=<bits> (mode-2 GEMMs only) selects an ablation:
    // 1 no DMA in the loop, 4 L2-hot DMA source, 0 / 8 DMA placement 0 / 1 (profiles/r01_gemm_pmc.txt).
    static int ppdbg = -2;
    if (ppdbg == -2) {
        const char* e = getenv("IGGT_GEMM_PPDBG");
        ppdbg = e ? atoi(e) : -1;
    }
    static bool attr_set = false;
    if (!attr_set) {
        const void* kernels[] = {
            (const void*)gemm_bf16_t256pp_kernel<1, 16, FMT>, (const void*)gemm_bf16_t256pp_kernel<2, 16, FMT>,
            (const void*)gemm_bf16_t256pp_kernel<3, 16, FMT>, (const void*)gemm_bf16_t256pp_kernel<2, 0, FMT>,
            (const void*)gemm_bf16_t256pp_kernel<2, 8, FMT>,  (const void*)gemm_bf16_t256pp_kernel<2, 17, FMT>,
            (const void*)gemm_bf16_t256pp_kernel<2, 20, FMT>, (const void*)gemm_bf16_t256pp_kernel<1, 48, FMT>,
            (const void*)gemm_bf16_t256pp_kernel<2, 48, FMT>};
        for (const void* k : kernels) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
        }
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_t256pp_kernel<1, 16, FMT, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds + LUT_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const dim3 grid(tiles_m * p.tiles_n), block(512);
    if (ppdbg == 32 && mode != 3) {
        if (mode == 1) hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<1, 48, FMT>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<2, 48, FMT>), grid, block, lds, stream, p);
        return 0;
    }
    if (ppdbg >= 0 && mode == 2) {
        if (ppdbg == 1) hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<2, 17, FMT>), grid, block, lds, stream, p);
        else if (ppdbg == 4) hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<2, 20, FMT>), grid, block, lds, stream, p);
        else if (ppdbg == 8) hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<2, 8, FMT>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<2, 0, FMT>), grid, block, lds, stream, p);
        return 0;
    }
    if (mode == 1 && p.act == 1 && gelu_lut_on)
        hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<1, 16, FMT, true>), grid, block, lds + LUT_BYTES, stream, p);
    else if (mode == 1) hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<1, 16, FMT>), grid, block, lds, stream, p);
    else if (mode == 2) hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<2, 16, FMT>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((gemm_bf16_t256pp_kernel<3, 16, FMT>), grid, block, lds, stream, p);
    return 0;
}

int iggt_launch_gemm_t256(const GemmParams& p_in, int fmt, hipStream_t stream) {
    GemmParams p = p_in;
    if ((long)p.M * p.lda >= (1L << 31) || (long)p.N * p.ldw >= (1L << 31)) return -100;
    if ((p.ldo % 8) != 0 || ((uintptr_t)p.out_f32 % 16) || ((uintptr_t)p.out_bf16 % 16)) return -100;  // 16-byte epilogue accesses
    if (p.K / TK < 4) return -100;      // the pipeline prologue needs 3 stages
    p.tiles_n = (p.N + TN - 1) / TN;
    const int tiles_m = (p.M + TM - 1) / TM;
    p.tiles_m = tiles_m;
    static int gm = -1;
    if (gm < 0) {
        const char* e = getenv("IGGT_GEMM_GROUP_M");
        gm = e ? atoi(e) : 4;   // measured at M=43968: qkv 695-717 -> 734-742 TF/s, fc1 679-683 -> 692-696 (GM 2 / 8: no gain)
    }
    p.group_m = (p.tiles_n > 4) ? gm : 0;
    int mode;
    if (p.out_bf16 && !p.gamma && p.rows_in == 0) mode = 1;
    else if (p.out_f32 && p.accumulate && p.rows_in == 0 && p.act == 0) mode = 2;
    else if (p.out_f32 && !p.accumulate && p.act == 0 && !p.gamma) mode = 3;
    else return -100;
    return fmt == FMT_F16 ? launch_t256<FMT_F16>(p, mode, tiles_m, stream)
                          : launch_t256<FMT_BF16>(p, mode, tiles_m, stream);
}
