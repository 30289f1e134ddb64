// 256 x 128 16-bit-operand MFMA GEMM, TWO workgroups per CU ("duo") -- the epilogue-overlap experiment VERDICT r2 item 4 asks
// for, kept behind IGGT_GEMM_DUO until it wins on a shape.
//
// Why: rocprofv3 counters on the production 256 x 256 kernel (profiles/r03_gemm_pmc.txt) put the matrix pipe at 37 % busy for
// fc1 + GELU (51 % for fc2): a tile costs ~46 us of which ~27 us are the K loop and ~19 us the epilogue -- 7 vector
// instructions per MFMA (erf-GELU, bias, convert, the LDS transposition) that run with the matrix pipe idle, because the one
// resident workgroup of a CU (128 KiB ring) has nothing else to issue.  Half the tile width halves the ring: a 3-stage ring of
// 24 KiB stages (A 256 x 32, W 128 x 32) is 72 KiB, two workgroups fit the 160 KiB LDS, and while one runs its epilogue the
// other runs its K loop.  Same wave tile as the big kernel (128 x 64, 8 MFMAs per 6 fragment reads), 4 waves per workgroup as
// 2 (M) x 2 (N), one wave of each workgroup per SIMD.  Costs: 1.5 x the LDS-DMA bytes per FLOP (the 128-wide W panel is shared
// by half as many MFMAs), and the ping-pong pairing of the big kernel (the two waves of a SIMD one barrier apart) becomes
// whatever the two independent workgroups fall into.
//
// Stage ring: at the top of iteration s stage s is resident, stage s + 1 in flight; the iteration requests stage s + 2 into the
// slot stage s - 1 occupied (every wave finished reading it before the barrier that ended iteration s - 1), runs the 16 MFMAs of
// stage s, waits until only the 6 newest DMA instructions of this thread are outstanding (= stage s + 1 has landed) and meets
// the others at one barrier.  LDS images and swizzles are those of gemm_bf16_t256.hip (64-byte rows, 16-byte slot index XOR
// ((row >> 2) & 3), applied to the DMA source address).
//
// Round 4 -- a 192-row variant (template parameter TM: wave tile 96 x 64 = 3 x 2 MFMA tiles, 20-KiB stages).  At the per-rank
// shapes of an 8-GPU run (M = 5 496) the 256-row tile quantises badly: fc2 / proj (N = 1 024) are 22 x 8 = 176 workgroups,
// one per CU on 176 of the 256 CUs, and fc1 (N = 4 096) is 704 workgroups = two rounds of the 512 slots with the second 37 %
// full.  192 rows: 29 x 8 = 232 workgroups of 3/4 the length (one round either way), 928 of 3/4 the length for fc1.  The
// dispatcher (gemm_bf16.hip) picks the row count with the smaller rounds x rows.
#include <stdlib.h>

#include "common.h"
#include "gemm_common.h"

namespace {

constexpr int TN = 128, TK = 32, NSTAGE = 3;
constexpr int W_BYTES = TN * TK * 2;            // 8 KiB

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int N>
IGGT_DEVINL void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Round 5 built a two-slice split-K of this kernel (ticket + fp32 slab hand-over); it measured slower at every per-rank shape (qkv
// 61 -> 93 us: pipeline fill and epilogue do not halve with K) and was removed in round 6 (profiles/r05_gemm_splitk_ab.txt, git history).
// Round 6 gave the 192-row tiles the GELU table of gemm_bf16_t256.hip (2 x (60 + 16.4) KiB of LDS fit) for the per-rank fc1:
// 583 vs 584 TF/s at M = 5 496 -- nothing; removed again (profiles/r06_bench_emu8_*.json).
template <int MODE, int FMT, int TM>
__global__ __launch_bounds__(256, 2) void gemm_h16_duo_kernel(const GemmParams p) {
    static_assert(TM == 256 || TM == 192, "row tile");
    constexpr int A_BYTES = TM * TK * 2;            // 16 | 12 KiB
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;  // 24 | 20 KiB
    constexpr int HALF = TM / 2;                    // rows of one wave row (wm) = rows of one epilogue phase
    constexpr int MI = HALF / 32;                   // 32-row MFMA tiles per wave: 4 | 3
    constexpr int AC = TM / 64;                     // 16-row A chunks moved per wave and stage: 4 | 3
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (p.group_m > 1) {   // groups of group_m row tiles, m fastest inside a group (see gemm_bf16_t256.hip)
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

    // DMA map: a 1-KiB chunk = 16 rows x 64 B; lane l fills row l / 4, slot l % 4 with source piece (l % 4) ^ ((row >> 2) & 3).
    // Wave w moves A chunks AC w .. AC w + AC - 1 (rows 16 AC w ..) and W chunks 2w, 2w+1 (rows 32w .. 32w+31).
    const int c_row = lane >> 2, c_pos = lane & 3;
    int a_off[AC], w_off[2];
#pragma unroll
    for (int i = 0; i < AC; ++i) {
        const int r = (AC * wave + i) * 16 + c_row;
        int ra = m0 + r;
        ra = ra < p.M ? ra : p.M - 1;
        a_off[i] = ra * (int)p.lda + (c_pos ^ ((r >> 2) & 3)) * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 16 + c_row;
        int rw = n0 + r;
        rw = rw < p.N ? rw : p.N - 1;
        w_off[i] = rw * (int)p.ldw + (c_pos ^ ((r >> 2) & 3)) * 8;
    }
    const int KT = p.K / TK;                          // >= 3 (launcher)
    auto dma_a = [&](int kt, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t*)(p.A + a_off[i] + kt * TK),
                                         (lptr_t*)(smem + (kt % NSTAGE) * STAGE_BYTES + (AC * wave + i) * 1024), 16, 0, 0);
    };
    auto dma_w = [&](int kt, int i) {
        __builtin_amdgcn_global_load_lds((gptr_t*)(p.W + w_off[i] + kt * TK),
                                         (lptr_t*)(smem + (kt % NSTAGE) * STAGE_BYTES + A_BYTES + (2 * wave + i) * 1024), 16, 0, 0);
    };
    auto dma_stage = [&](int kt) {
#pragma unroll
        for (int i = 0; i < AC; ++i) dma_a(kt, i);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma_w(kt, i);
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    int lane_off[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) lane_off[kc] = frow * 64 + ((((2 * kc + fhalf) ^ (frow >> 2)) & 3) << 4);
    const int a_base = wm * HALF * 64, w_base = A_BYTES + wn * 64 * 64;

    dma_stage(0);
    dma_stage(1);
    wait_vm<AC + 2>();
    __builtin_amdgcn_s_barrier();

#pragma unroll 1
    for (int s = 0; s < KT; ++s) {
        const char* st = smem + (s % NSTAGE) * STAGE_BYTES;
        const bool issue = s + 2 < KT;
        bf16x8 a0[MI], b0[2], a1[MI], b1[2];
#pragma unroll
        for (int i = 0; i < MI; ++i) a0[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 32 * 64 + lane_off[0]);
#pragma unroll
        for (int j = 0; j < 2; ++j) b0[j] = *reinterpret_cast<const bf16x8*>(st + w_base + j * 32 * 64 + lane_off[0]);
        if (issue) {   // the A pieces of stage s + 2 go out beside the first fragment reads, the W pieces among the MFMAs
#pragma unroll
            for (int i = 0; i < AC; ++i) dma_a(s + 2, i);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) a1[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 32 * 64 + lane_off[1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) b1[j] = *reinterpret_cast<const bf16x8*>(st + w_base + j * 32 * 64 + lane_off[1]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 2 * MI; ++q) acc[q >> 1][q & 1] = mfma32h<FMT>(a0[q >> 1], b0[q & 1], acc[q >> 1][q & 1]);
        if (issue) {
            dma_w(s + 2, 0);
            dma_w(s + 2, 1);
        }
#pragma unroll
        for (int q = 0; q < 2 * MI; ++q) acc[q >> 1][q & 1] = mfma32h<FMT>(a1[q >> 1], b1[q & 1], acc[q >> 1][q & 1]);
        __builtin_amdgcn_s_setprio(0);
        // stage s + 1 must have landed (everything but the AC + 2 newest requests of this thread) before anyone reads it
        if (issue) wait_vm<AC + 2>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: two HALF-row phases through LDS as fp32 [HALF][128] (64 of the 72 | 48 of the 60 KiB), 16-byte row segments ----
    float* stile = reinterpret_cast<float*>(smem);
    const int c4 = tid & 31, r0 = tid >> 5;          // 32 threads x 4 columns, 8 rows per pass
    const int n = n0 + c4 * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, gamma4 = {1.f, 1.f, 1.f, 1.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    if (MODE == 2 && p.gamma) gamma4 = *reinterpret_cast<const f32x4*>(p.gamma + n);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        constexpr int PASSES = HALF / 8;   // 16 | 12
        f32x4 old[MODE == 2 ? PASSES : 1];
        if constexpr (MODE == 2) {   // read-modify-write: all old row segments requested before the LDS transposition
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const int m = m0 + half * HALF + i * 8 + r0;
                const int mc = m < p.M ? m : p.M - 1;
                old[i] = *reinterpret_cast<const f32x4*>(p.out_f32 + (long)mc * p.ldo + n);
            }
        }
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stile[(i * 32 + mfma32_row(r, lane)) * TN + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            const int row = pass * 8 + r0;
            const int m = m0 + half * HALF + row;
            f32x4 v4 = *reinterpret_cast<const f32x4*>(stile + row * TN + c4 * 4);
            if constexpr (MODE == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = fmaf(v4[e] + bias4[e], gamma4[e], old[pass][e]);
                if (m < p.M) *reinterpret_cast<f32x4*>(p.out_f32 + (long)m * p.ldo + n) = v4;
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] += bias4[e];
                if (p.act == 1) {
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

template <int FMT, int TM>
int launch_duo(const GemmParams& p, int mode, hipStream_t stream) {
    const int lds = NSTAGE * (TM * TK * 2 + W_BYTES);  // 72 | 60 KiB: two workgroups per CU
    static bool attr_set = false;
    if (!attr_set) {
        const void* kernels[] = {(const void*)gemm_h16_duo_kernel<1, FMT, TM>, (const void*)gemm_h16_duo_kernel<2, FMT, TM>,
                                 (const void*)gemm_h16_duo_kernel<3, FMT, TM>};
        for (const void* k : kernels) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    const dim3 grid(p.tiles_m * p.tiles_n), block(256);
    if (mode == 1) hipLaunchKernelGGL((gemm_h16_duo_kernel<1, FMT, TM>), grid, block, lds, stream, p);
    else if (mode == 2) hipLaunchKernelGGL((gemm_h16_duo_kernel<2, FMT, TM>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((gemm_h16_duo_kernel<3, FMT, TM>), grid, block, lds, stream, p);
    return 0;
}

}  // namespace

// rows: 256 | 192 rows per tile.  Returns -100 when the parameter combination is not covered (caller falls through to the
// other kernels)
int iggt_launch_gemm_duo(const GemmParams& p_in, int fmt, int rows, hipStream_t stream) {
    GemmParams p = p_in;
    if (rows != 256 && rows != 192) return -100;
    if ((long)p.M * p.lda >= (1L << 31) || (long)p.N * p.ldw >= (1L << 31)) return -100;
    if ((p.ldo % 8) != 0 || ((uintptr_t)p.out_f32 % 16) || ((uintptr_t)p.out_bf16 % 16)) return -100;
    if ((p.N % TN) != 0 || p.K / TK < 3) return -100;
    p.tiles_n = p.N / TN;
    p.tiles_m = (p.M + rows - 1) / rows;
    static int gm = -1;
    if (gm < 0) {
        const char* e = getenv("IGGT_GEMM_GROUP_M");
        gm = e ? atoi(e) : 4;
    }
    p.group_m = (p.tiles_n > 8) ? gm : 0;
    int mode;
    if (p.out_bf16 && !p.gamma && p.rows_in == 0) mode = 1;
    else if (p.out_f32 && p.accumulate && p.rows_in == 0 && p.act == 0) mode = 2;
    else if (p.out_f32 && !p.accumulate && p.act == 0 && !p.gamma) mode = 3;
    else return -100;
    if (rows == 192) return fmt == FMT_F16 ? launch_duo<FMT_F16, 192>(p, mode, stream) : launch_duo<FMT_BF16, 192>(p, mode, stream);
    return fmt == FMT_F16 ? launch_duo<FMT_F16, 256>(p, mode, stream) : launch_duo<FMT_BF16, 256>(p, mode, stream);
}
