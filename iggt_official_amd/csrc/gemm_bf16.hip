// 16-bit-operand (bf16 / fp16, template FMT) MFMA GEMM with fused epilogues for the IGGT linear layers.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        A: [M,K] bf16 row-major (lda), W: [N,K] bf16 (ldw)
//
// W is in torch.nn.Linear layout ([out,in]), so both MFMA operands are read along K ("B^T
// input"), each lane taking 8 contiguous bf16 (16 B) of one row.  fp32 accumulation.
//
// Replaces on the hot path (reference file:line):
//   attn.qkv / attn.proj            iggt/layers/attention.py:40,45,52,75
//   mlp.fc1 + GELU / mlp.fc2        iggt/layers/mlp.py:34-39
//   LayerScale + residual add       iggt/layers/layer_scale.py:26, iggt/layers/block.py:105-106
//   patch-embed conv (as im2row GEMM) + pos-embed add   iggt/layers/patch_embed.py:75-77,
//                                                       iggt/layers/vision_transformer.py:223
//
// Tile: 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16.
// Global->register->LDS staging, issued one K-tile ahead and written after the MFMA phase
// (split issue/write), double-buffered LDS (64 KiB -> 2 workgroups/CU), XOR-swizzled rows
// (common.h swz_off) so the ds_read_b128 operand reads are bank-conflict free.
// Grid: 1-D over (m-tile, n-tile), n fastest, XCD-chunked so one XCD's L2 keeps the A row-panel.
#include <stdlib.h>

#include "common.h"
#include "gemm_common.h"
#include "../../include/iggt_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per buffer

template <int FMT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging map: thread -> (row = tid/8 + 32*i, 16-B piece = tid%8) -------------------------
    const int ld_row = tid >> 3, ld_piece = tid & 7;
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ld_row + 32 * i;
        int ra = m0 + r;
        ra = ra < p.M ? ra : p.M - 1;  // tail rows: duplicate the last row, masked at the store
        int rw = n0 + r;
        rw = rw < p.N ? rw : p.N - 1;
        a_src[i] = p.A + (long)ra * p.lda + ld_piece * 8;
        w_src[i] = p.W + (long)rw * p.ldw + ld_piece * 8;
        lds_off[i] = swz_off(r, ld_piece);
    }
    u32x4 sa[4], sw[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa[i] = *reinterpret_cast<const u32x4*>(a_src[i] + kt * BK);
            sw[i] = *reinterpret_cast<const u32x4*>(w_src[i] + kt * BK);
        }
    };
    auto swrite = [&](int buf) {
        char* sA = smem + buf * (2 * TILE_BYTES);
        char* sW = sA + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(sA + lds_off[i]) = sa[i];
            *reinterpret_cast<u32x4*>(sW + lds_off[i]) = sw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // operand read offsets: row = base + (lane&31); slot = 2*kc + (lane>>5); key = (row>>1)&7
    const int frow = lane & 31, fhalf = lane >> 5;
    const int KT = p.K / BK;

    gload(0);
    swrite(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) gload(kt + 1);
        const char* sA = smem + (kt & 1) * (2 * TILE_BYTES);
        const char* sW = sA + TILE_BYTES;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(sA + swz_off(wm * 64 + i * 32 + frow, 2 * kc + fhalf));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(sW + swz_off(wn * 64 + j * 32 + frow, 2 * kc + fhalf));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32h<FMT>(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < KT) swrite((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------
    float bias[2], gamma[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        bias[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
        gamma[j] = (p.gamma && n < p.N) ? p.gamma[n] : 1.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            gemm_epilogue_tile<0, FMT>(p, acc[i][j], m0 + wm * 64 + i * 32, n, lane, bias[j], gamma[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Same 128 x 128 tile, operands staged by LDS-DMA through a 4-deep ring of 32-wide K stages (4 x 16 KiB = 64 KiB, still two
// workgroups per CU) instead of global -> register -> ds_write one K tile ahead.  The register-staged loop above keeps ONE
// 64-wide K tile in flight per workgroup: a K step (16 MFMAs per wave, ~0.3 us of matrix pipe) cannot cover a global-load
// latency of 1-2 us, and the kernel ran latency-bound at 245-476 TF/s on the shapes it gets in a sharded run (per-rank
// proj / fc2: M = 5 496, N = 1 024).  Here up to three stages are in flight behind counted vmcnt waits, one barrier per
// stage (all four waves in lockstep -- the second workgroup of the CU fills the read phases).  LDS image and source-side
// swizzle as in gemm_bf16_t256.hip: 64-byte rows, chunk j of 1 KiB = rows 16j .. 16j + 15, lane l -> row 16j + l / 4,
// slot l % 4 filled with source piece (l % 4) ^ ((row >> 2) & 3).
constexpr int DK = 32, DSTAGES = 4;
constexpr int D_OP_BYTES = BM * DK * 2;        // 8 KiB per operand per stage
constexpr int D_STAGE_BYTES = 2 * D_OP_BYTES;  // A + W

template <int N>
IGGT_DEVINL void wait_vmcnt_n() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EPI_LDS: the accumulators leave through LDS (the idle ring: 128 x 128 fp32 = 64 KiB) as 16-byte row segments -- the direct
// epilogue issues 64 scalar loads + 64 scalar 4-byte (or 2-byte) stores per lane, which was most of the 41 us a K = 1 024
// tile took.  Needs N % 4 == 0, ldo % 4 == 0 and a 16-byte aligned output (launcher).
template <int FMT, bool EPI_LDS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_dma_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // wave w moves chunks 2w, 2w + 1 of A and of W (rows 32w .. 32w + 31 of each operand)
    const int c_row = lane >> 2, c_pos = lane & 3;
    int a_off[2], w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 16 + c_row;
        const int piece = c_pos ^ ((r >> 2) & 3);
        int ra = m0 + r;
        ra = ra < p.M ? ra : p.M - 1;   // tail rows: duplicate the last row, masked at the store
        int rw = n0 + r;
        rw = rw < p.N ? rw : p.N - 1;
        a_off[i] = ra * (int)p.lda + piece * 8;
        w_off[i] = rw * (int)p.ldw + piece * 8;
    }
    auto dma = [&](int kt) {
        char* st = smem + (kt & (DSTAGES - 1)) * D_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t*)(p.A + a_off[i] + kt * DK), (lptr_t*)(st + (2 * wave + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)(p.W + w_off[i] + kt * DK),
                                             (lptr_t*)(st + D_OP_BYTES + (2 * wave + i) * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    int lane_off[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) lane_off[kc] = frow * 64 + ((((2 * kc + fhalf) ^ (frow >> 2)) & 3) << 4);
    const int a_base = wm * 64 * 64, w_base = D_OP_BYTES + wn * 64 * 64;
    const int KT = p.K / DK;

    dma(0);
    if (KT > 1) dma(1);
    if (KT > 2) dma(2);
#pragma unroll 1
    for (int s = 0; s < KT; ++s) {
        // stage s has landed: at most the stages s + 1, s + 2 (4 DMA instructions each) may still be in flight
        if (s + 2 < KT) wait_vmcnt_n<8>();
        else if (s + 1 < KT) wait_vmcnt_n<4>();
        else wait_vmcnt_n<0>();
        __builtin_amdgcn_s_barrier();          // ... in every wave, and every wave is done with stage s - 1
        __builtin_amdgcn_sched_barrier(0);
        if (s + 3 < KT) dma(s + 3);            // into the slot of stage s - 1
        const char* st = smem + (s & (DSTAGES - 1)) * D_STAGE_BYTES;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 32 * 64 + lane_off[kc]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(st + w_base + j * 32 * 64 + lane_off[kc]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32h<FMT>(a[i], b[j], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of stage s are complete
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (EPI_LDS) {
        __syncthreads();   // every wave is done with the ring
        float* stile = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stile[(wm * 64 + i * 32 + mfma32_row(r, lane)) * BN + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
        // a thread's 16 row segments (row = 8 * pass + tid / 32) share their 4 columns: bias / gamma once, and in the
        // accumulate mode all 16 old values are requested before the first store (a load behind a store waits for it)
        const int c4 = tid & 31, r0 = tid >> 5;
        const int n = n0 + c4 * 4;
        const bool ncol = n < p.N;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, gamma4 = {1.f, 1.f, 1.f, 1.f};
        if (ncol && p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
        if (ncol && p.gamma) gamma4 = *reinterpret_cast<const f32x4*>(p.gamma + n);
        const bool rmw = p.out_f32 && p.accumulate && p.rows_in == 0;
        f32x4 old[16];
        if (rmw) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + i * 8 + r0;
                const int mc = m < p.M ? m : p.M - 1;
                old[i] = ncol ? *reinterpret_cast<const f32x4*>(p.out_f32 + (long)mc * p.ldo + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) {
            const int row = pass * 8 + r0;
            const int m = m0 + row;
            if (m >= p.M || !ncol) continue;
            f32x4 v4 = *reinterpret_cast<const f32x4*>(stile + row * BN + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] += bias4[e];
            if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = gelu_erf(v4[e]);
            } else if (p.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = fmaxf(v4[e], 0.f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] *= gamma4[e];
            long orow = m;
            if (p.rows_in > 0) {
                const int g = m / p.rows_in, w = m - g * p.rows_in;
                orow = (long)g * p.rows_out + p.row_off + w;
                if (p.add_table) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(p.add_table + (long)w * p.N + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] += t[e];
                }
            }
            if (p.out_f32) {
                float* dst = p.out_f32 + orow * p.ldo + n;
                if (rmw) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] += old[pass][e];
                } else if (p.accumulate) {
                    const f32x4 o4 = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] += o4[e];
                }
                *reinterpret_cast<f32x4*>(dst) = v4;
            } else {
                u32x2 o;
                o[0] = pack_h2<FMT>(v4[0], v4[1]);
                o[1] = pack_h2<FMT>(v4[2], v4[3]);
                *reinterpret_cast<u32x2*>(p.out_bf16 + orow * p.ldo + n) = o;
            }
        }
        return;
    }

    float bias[2], gamma[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        bias[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
        gamma[j] = (p.gamma && n < p.N) ? p.gamma[n] : 1.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            gemm_epilogue_tile<0, FMT>(p, acc[i][j], m0 + wm * 64 + i * 32, n, lane, bias[j], gamma[j]);
    }
}

}  // namespace

static int force_small_tile() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("IGGT_GEMM_TILE128");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v;
}

static int gemm_h16(int fmt, const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias,
                    const float* gamma, const float* add_table, void* out, long ldo, int out_is_f32, int accumulate,
                    int act, int rows_in, int rows_out, int row_off, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0) return -1;
    if ((lda % 8) != 0 || (ldw % 8) != 0) return -2;  // 16-B aligned operand rows
    if (accumulate && !out_is_f32) return -3;
    if (add_table && rows_in <= 0) return -4;
    GemmParams p;
    p.A = (const bf16_t*)A;
    p.W = (const bf16_t*)W;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldw = ldw;
    p.tiles_n = (N + BN - 1) / BN;
    p.bias = bias; p.gamma = gamma; p.add_table = add_table;
    p.out_f32 = out_is_f32 ? (float*)out : nullptr;
    p.out_bf16 = out_is_f32 ? nullptr : (bf16_t*)out;
    p.ldo = ldo;
    p.accumulate = accumulate; p.act = act;
    p.rows_in = rows_in; p.rows_out = rows_out; p.row_off = row_off;
    // large problems: 256x256 LDS-DMA kernel (gemm_bf16_t256.hip); small / ragged-N ones: this 128x128 kernel
    // Exception: grids of < 128 big tiles (e.g. N = 1024 with a few thousand rows: the per-rank proj / fc2 GEMMs of an
    // 8-GPU run) leave most CUs idle -- the 128^2 kernel's 4x finer grid wins there (measured 108 vs 131 us).
    const long big_tiles = (long)((M + 255) / 256) * (N / 256);
    // 256 x 128 two-workgroups-per-CU kernel (gemm_bf16_duo.hip): slower than the 256^2 tile on full grids (32 views: qkv 653
    // vs 721, fc2 753 vs 901 TF/s -- 1.5x the LDS-DMA bytes per FLOP), faster where the 256^2 grid is one or two ragged rounds
    // (M = 5 496, the per-rank shapes of an 8-GPU run: qkv 437 vs 417, fc1 516 vs 490, fc2 519 vs 500 TF/s; whole per-rank
    // forward 67.4 -> 64.5 ms).  IGGT_GEMM_DUO: 0 never, 1 wherever it applies (A/B runs), default 2 = small grids only.
    static int duo = -1;
    if (duo < 0) {
        const char* e = getenv("IGGT_GEMM_DUO");
        duo = e ? atoi(e) : 2;
    }
    const long duo_tiles = (long)((M + 255) / 256) * (N / 128);
    // (at 8 / 16 views -- M = 10 992 / 21 984 -- the 256^2 tile wins on every shape again: fc2 740 vs 593, fc1 636 vs 576 TF/s)
    const bool duo_auto = duo == 2 && M >= 1024 && M < 8192 && duo_tiles <= 1024 && (big_tiles >= 128 || K >= 2048);
    // Round 4 -- 192-row tiles where they need fewer rounds x rows of the 512 workgroup slots than 256-row ones
    // (gemm_bf16_duo.hip).  M = 5 496, one rank of an 8-GPU run (probes/gemm_rank_ab.py, profiles/r04_gemm_rank_ab.txt):
    // fc2 176 -> 232 workgroups of 3/4 the length, 76.2 -> 67.2 us; qkv 528 (two rounds, the second 3 % full) -> 696, 66.0 ->
    // 58.4 us; fc1 704 -> 928, 81.6 -> 81.4 us (no gain: its rounds are the GELU epilogue's).  IGGT_GEMM_DUO192=0: 256 rows always.
    static int duo192 = -1;
    if (duo192 < 0) {
        const char* e = getenv("IGGT_GEMM_DUO192");
        duo192 = (e && e[0] == '0') ? 0 : 1;
    }
    const long slots = 512, t192 = (long)((M + 191) / 192) * (N / 128);
    const long cost256 = (duo_tiles + slots - 1) / slots * 256, cost192 = (t192 + slots - 1) / slots * 192;
    const int rows = (duo192 && cost192 < cost256) ? 192 : 256;
    // (An earlier remedy for the qkv shape -- 21 x 24 full 256-row tiles here + the last 120 rows on the 128^2 kernel, 61.7 us --
    // is gone: wherever it saved a round, 192-row tiles save it too and win.)
    // (Round 5's two-slice split-K of the duo kernel measured slower at every per-rank shape and was removed in round 6.)
    if ((duo == 1 || duo_auto) && M >= 512 && (N % 128) == 0 && force_small_tile() == 0) {
        const int rc = iggt_launch_gemm_duo(p, fmt, rows, (hipStream_t)stream);
        if (rc == 0) {
            IGGT_CHECK_LAUNCH();
            return 0;
        }
        if (rc != -100) return rc;
    }
    if (M >= 1024 && (N % 256) == 0 && big_tiles >= 128 && force_small_tile() == 0) {
        const int rc = iggt_launch_gemm_t256(p, fmt, (hipStream_t)stream);
        if (rc == 0) {
            IGGT_CHECK_LAUNCH();
            return 0;
        }
        if (rc != -100) return rc;
    }
    const int tiles_m = (M + BM - 1) / BM;
    const int lds = 2 * 2 * TILE_BYTES;  // 64 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<FMT_BF16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<FMT_F16>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const dim3 grid(tiles_m * p.tiles_n), block(256);
    // LDS-DMA variant: needs 32-bit operand offsets and 16-byte aligned, 8-element-strided operand rows (checked above);
    // IGGT_GEMM128_DMA=0 selects the register-staged loop (A/B runs)
    static int dma_on = -1;
    if (dma_on < 0) {
        const char* e = getenv("IGGT_GEMM128_DMA");
        dma_on = (e && e[0] == '0') ? 0 : 1;
    }
    const bool small_offsets = (long)M * lda < (1L << 31) && (long)N * ldw < (1L << 31);
    const bool aligned = (((uintptr_t)A | (uintptr_t)W) % 16) == 0;
    if (dma_on && small_offsets && aligned) {
        constexpr int DLDS = DSTAGES * D_STAGE_BYTES;
        static bool attr_dma = false;
        if (!attr_dma) {
            const void* ks[] = {(const void*)gemm_bf16_dma_kernel<FMT_BF16, false>, (const void*)gemm_bf16_dma_kernel<FMT_F16, false>,
                                (const void*)gemm_bf16_dma_kernel<FMT_BF16, true>, (const void*)gemm_bf16_dma_kernel<FMT_F16, true>};
            for (const void* k : ks) {
                const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, DLDS);
                if (e != hipSuccess) return (int)e;
            }
            attr_dma = true;
        }
        const bool epi_lds = (N % 4) == 0 && (ldo % 4) == 0 && ((uintptr_t)out % 16) == 0 &&
                             (!bias || ((uintptr_t)bias % 16) == 0) && (!gamma || ((uintptr_t)gamma % 16) == 0) &&
                             (!add_table || ((uintptr_t)add_table % 16) == 0);
        hipStream_t st = (hipStream_t)stream;
        if (epi_lds) {
            if (fmt == FMT_F16) hipLaunchKernelGGL((gemm_bf16_dma_kernel<FMT_F16, true>), grid, block, DLDS, st, p);
            else hipLaunchKernelGGL((gemm_bf16_dma_kernel<FMT_BF16, true>), grid, block, DLDS, st, p);
        } else {
            if (fmt == FMT_F16) hipLaunchKernelGGL((gemm_bf16_dma_kernel<FMT_F16, false>), grid, block, DLDS, st, p);
            else hipLaunchKernelGGL((gemm_bf16_dma_kernel<FMT_BF16, false>), grid, block, DLDS, st, p);
        }
        IGGT_CHECK_LAUNCH();
        return 0;
    }
    if (fmt == FMT_F16) hipLaunchKernelGGL(gemm_bf16_kernel<FMT_F16>, grid, block, lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_bf16_kernel<FMT_BF16>, grid, block, lds, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_gemm_bf16(const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                              const float* bias, const float* gamma, const float* add_table,
                              void* out, long ldo, int out_is_f32, int accumulate, int act,
                              int rows_in, int rows_out, int row_off, void* stream) {
    return gemm_h16(FMT_BF16, A, lda, W, ldw, M, N, K, bias, gamma, add_table, out, ldo, out_is_f32, accumulate, act,
                    rows_in, rows_out, row_off, stream);
}

extern "C" int iggt_gemm_f16(const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                             const float* bias, const float* gamma, const float* add_table,
                             void* out, long ldo, int out_is_f32, int accumulate, int act,
                             int rows_in, int rows_out, int row_off, void* stream) {
    return gemm_h16(FMT_F16, A, lda, W, ldw, M, N, K, bias, gamma, add_table, out, ldo, out_is_f32, accumulate, act,
                    rows_in, rows_out, row_off, stream);
}
