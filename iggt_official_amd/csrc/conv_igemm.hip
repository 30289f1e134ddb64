// Implicit-GEMM 2-D convolution on MFMA for the dense heads (DPT depth/point heads, SamProjector,
// part head): NHWC fp32 activations in HBM, bf16 MFMA with fp32 accumulation, optional split-bf16
// ("x3") operands for fp32-grade results.
//
// Replaces (reference file:line): every nn.Conv2d / nn.ConvTranspose2d of iggt/heads/dpt_head.py
// (resize_layers 72-88, scratch.layerN_rn 345-357, ResidualConvUnit 369-411, FeatureFusionBlock.out_conv
// 441-443,479, output_conv1/2 117-128), iggt/heads/adaptor.py Projects 9-35 + resize stacks 152-175 and
// the 3x3 convs of iggt/heads/window_sa.py (CAB 40-47, conv_after_body/conv_before_upsample/conv_last
// 383-391), which the reference runs in fp32 through cuDNN under autocast(enabled=False) (vggt.py:189).
//
// GEMM view:  Y[m][n] = sum_k A[m][k] W[n][k],  m = (img, oy, ox) output pixel, n = output channel,
// k = (tap, c) with the tap major, so a 32-wide K chunk stays inside one tap and is a contiguous
// 128-byte run of the NHWC input (zero-filled when the tap falls outside the image).
//
// Precision: the reference heads are fp32.  PREC=3 splits every operand into bf16 hi + bf16 lo
// (x = hi + lo + O(2^-17 x)) and issues three MFMAs per product (hi*hi + hi*lo + lo*hi), i.e.
// ~2^-16-relative products accumulated in fp32 -- fp32-grade results at 1/3 of the bf16 MFMA rate
// (~5x the fp32-MFMA rate).  PREC=1 uses plain bf16 operands.  Activations are split on the fly in
// the loader (fp32 -> hi/lo while staging to LDS); weights are pre-split once on the host side.
// PREC=2: fp16 operands, activations hi + lo (exact to 2^-22), weights rounded ONCE to fp16 -- two
// MFMAs per product; the mean response to the weight rounding is restored per border class by
// conv_meancomp.hip (`corr`, added by the epilogue in place of the bias).  Head output error 1.7e-4 on
// photographs with every layer at PREC=2 (profiles/r03_conv_precision.txt), against 1.2e-5 for PREC=3.
//
// Fused into the loader: ReLU on the input (ResidualConvUnit pre-activation).  Fused into the epilogue:
// bias, ReLU / LeakyReLU(0.01) / exact GELU, residual add (optionally of the rectified residual -- the
// reference's in-place-ReLU skip), and a generic output scatter
//     out pixel = (oy * osy + ooy, ox * osx + oox), channel = n % cout_phys  with phase n / cout_phys
// which implements ConvTranspose2d with kernel == stride as a 1x1 GEMM + pixel shuffle and
// ConvTranspose2d(k4,s2,p1) as four 2x2 parity convolutions.
//
// Tiling: 128 x (32*WN*WAVES_N) output tile, BK = 32, 256 threads; LDS images [rows][32] bf16 with
// the 16-B slot XOR ((row >> 2) & 3) (conflict-free ds_read_b128 for 64-byte rows), double-buffered,
// register-staged two K-steps ahead (two register sets).
#include <stdlib.h>

#include "common.h"
#include "../../include/iggt_hip.h"

namespace {

struct ConvParams {
    const float* x;        // [Nimg][Hi][Wi][ldx] fp32, channels [0, Cin) used
    const bf16_t* w_hi;    // [Cout][K] bf16, K = KH*KW*Cin (tap-major)
    const bf16_t* w_lo;    // PREC == 3 only
    const float* bias;     // [Cout] or null
    const float* corr;     // PREC == 2 with padding: [9][Cout] bias + mean-input correction per border class, else null
    const float* res;      // residual, same indexing as y (ldr channels stride), or null
    const float* res2;     // second residual (never rectified), same layout as res, or null
    float* y;              // [Nimg][Hout][Wout][ldy]
    int Nimg, Hi, Wi, Cin, ldx;
    int Ho, Wo;            // GEMM-row grid (number of kernel placements per image)
    int Cout;              // GEMM N
    int KH, KW, stride, pad_y, pad_x;
    int Hout, Wout, ldy, ldr;
    int osy, osx, ooy, oox;  // output scatter
    int cout_phys;           // channels per output pixel (pixel-shuffle when Cout > cout_phys)
    int ps;                  // pixel-shuffle factor (phase -> (phase / ps, phase % ps) added to the pixel)
    int relu_in, relu_res, act;  // act: 0 none, 1 relu, 2 leaky 0.01, 3 gelu(erf)
    int tiles_n;
    long M;
    // split K (SPLITK instantiation only): grid.y slices of kchunks K-chunks each write their raw partial tile to
    // part[slice][M][Cout]; conv_splitk_finalize_kernel adds the slices in order and applies the epilogue
    int ksplit, kchunks;
    float* part;
};

constexpr int BK = 32;

// key = (row>>2)&3 keeps ds_read_b128's 16-lane groups conflict-free (rows with equal row%4 get distinct slots); the
// extra ^ ((row>>1)&1) gives rows r and r+2 -- same 128-byte window of the ds_write_b128 bank map, written by one
// 8-lane group of the A loader -- different slots (PMC: 2-way write conflicts on 20 % of the LDS cycles before).
IGGT_DEVINL int slot_swz(int row, int slot) {
    return row * 64 + (((slot ^ (row >> 2) ^ ((row >> 1) & 1)) & 3) << 4);
}

// Tile geometry of one instantiation (shared by the kernel and its launcher).
template <int PREC, int WM, int WN, int WAVES_M, int WAVES_N>
struct ConvTile {
    static constexpr int THREADS = 64 * WAVES_M * WAVES_N;
    static constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    static constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64;
    static constexpr int STAGE = PREC == 3 ? 2 * (A_BYTES + W_BYTES) : PREC == 2 ? 2 * A_BYTES + W_BYTES : A_BYTES + W_BYTES;
    // the epilogue transposes the tile through the (then idle) stage ring, at most 128 KiB (= 128 rows of 256) a pass
    static constexpr int EPI_ROWS = (BM * BN * 4 > 131072) ? BM / 2 : BM;
    static constexpr int SMEM = (2 * STAGE > EPI_ROWS * BN * 4) ? 2 * STAGE : EPI_ROWS * BN * 4;
};

template <int PREC, int WM, int WN, int WAVES_M, int WAVES_N, bool SPLITK = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void conv_igemm_kernel(const ConvParams pin) {
    using T = ConvTile<PREC, WM, WN, WAVES_M, WAVES_N>;
    constexpr int FMT = PREC == 2 ? FMT_F16 : FMT_BF16;
    // SPLITK: this workgroup owns K chunks [kt_beg, kt_beg + kchunks) and writes its raw partial sums (no bias / activation
    // / residual, plain [M][Cout] layout) to its slice of the scratch buffer -- the epilogue code below runs unchanged on a
    // parameter block whose output side has been redirected.
    ConvParams ploc = pin;
    int kt_beg = 0;
    if constexpr (SPLITK) {
        kt_beg = blockIdx.y * pin.kchunks;
        ploc.y = pin.part + (long)blockIdx.y * pin.M * pin.Cout;
        ploc.ldy = pin.Cout;
        ploc.bias = nullptr; ploc.corr = nullptr; ploc.res = nullptr; ploc.res2 = nullptr;
        ploc.act = 0; ploc.relu_res = 0;
    }
    const ConvParams& p = ploc;
    constexpr int THREADS = T::THREADS, BM = T::BM, BN = T::BN;
    static_assert(BM == THREADS / 2, "A loader: two threads per tile row");
    static_assert(T::EPI_ROWS == BM || WAVES_M == 2, "two-pass epilogue: one wave row per pass");
    constexpr int A_BYTES = T::A_BYTES, W_BYTES = T::W_BYTES, STAGE = T::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;

    // ---- A loader: thread -> (row = tid/2, 16 channels = 64 B at half = tid&1) ---------------------
    const int a_row = tid >> 1, a_half = tid & 1;
    long am = m0 + a_row;
    const bool a_ok = am < p.M;
    if (!a_ok) am = p.M - 1;
    const int hw = p.Ho * p.Wo;
    const int a_img = (int)(am / hw);
    const int a_rem = (int)(am - (long)a_img * hw);
    const int a_oy = a_rem / p.Wo, a_ox = a_rem - a_oy * p.Wo;
    const int iy0 = a_oy * p.stride - p.pad_y, ix0 = a_ox * p.stride - p.pad_x;
    const float* a_base = p.x + ((long)a_img * p.Hi * p.Wi) * p.ldx + a_half * 16;
    // ---- W loader: rows of 64 B (32 bf16); THREADS x 16 B = THREADS/4 rows per pass ------------------
    constexpr int RPP = THREADS / 4;
    constexpr int W_PASSES = (BN + RPP - 1) / RPP;
    const int w_row = tid >> 2, w_piece = tid & 3;  // 4 x 16 B per row
    const long Ktot = (long)p.KH * p.KW * p.Cin;
    const int chunks_per_tap = p.Cin / BK;
    const int KT_all = p.KH * p.KW * chunks_per_tap;
    const int KT = SPLITK ? ((KT_all - kt_beg) < p.kchunks ? (KT_all - kt_beg) : p.kchunks) : KT_all;

    // ---- loader state: chunks are requested strictly in order, so (tap, channel offset) is tracked incrementally
    //      (no per-chunk integer division) and the tap's pixel pointer / in-bounds flag only change with the tap.
    int ld_c0 = 0, ld_ky = 0, ld_kx = 0;
    if constexpr (SPLITK) {
        const int tap0 = kt_beg / chunks_per_tap;
        ld_c0 = (kt_beg - tap0 * chunks_per_tap) * BK;
        ld_ky = tap0 / p.KW;
        ld_kx = tap0 - ld_ky * p.KW;
    }
    bool ld_ok = false;
    const float* ld_src = a_base;
    auto set_tap = [&]() {
        const int iy = iy0 + ld_ky, ix = ix0 + ld_kx;
        ld_ok = a_ok && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        ld_src = a_base + ((long)(ld_ok ? iy : 0) * p.Wi + (ld_ok ? ix : 0)) * p.ldx;
    };
    set_tap();
    const bf16_t* w_ptr_hi[W_PASSES];
    const bf16_t* w_ptr_lo[W_PASSES];
#pragma unroll
    for (int q = 0; q < W_PASSES; ++q) {
        int r = n0 + w_row + RPP * q;
        if (w_row + RPP * q >= BN) r = n0;  // BN < rows per pass: the surplus threads duplicate row 0
        r = r < p.Cout ? r : p.Cout - 1;
        w_ptr_hi[q] = p.w_hi + (long)r * Ktot + w_piece * 8;
        w_ptr_lo[q] = (PREC == 3) ? p.w_lo + (long)r * Ktot + w_piece * 8 : nullptr;
    }
    int w_k = kt_beg * BK;  // running K offset (elements) of the weight loads
    const float relu_floor = p.relu_in ? 0.f : -INFINITY;  // fused ReLU-on-load as one v_max (x = max(x, floor))

    // Two register sets: the loads of chunk kt+2 are in flight while chunk kt is multiplied and chunk kt+1 is written
    // to LDS (one set was measured to leave the waves parked on vmcnt 45 % of the time, profiles/r01_conv_pmc.txt).
    struct Stage {
        f32x4 ra[4];
        u32x4 rwh[W_PASSES], rwl[W_PASSES];
    };
    Stage R0, R1;
    auto gload = [&](Stage& R) {
        const float* src = ld_src + ld_c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * i);
            if (!ld_ok) t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = fmaxf(t[e], relu_floor);
            R.ra[i] = t;
        }
#pragma unroll
        for (int q = 0; q < W_PASSES; ++q) {
            R.rwh[q] = *reinterpret_cast<const u32x4*>(w_ptr_hi[q] + w_k);
            if (PREC == 3) R.rwl[q] = *reinterpret_cast<const u32x4*>(w_ptr_lo[q] + w_k);
        }
        w_k += BK;
        ld_c0 += BK;
        if (ld_c0 == p.Cin) {  // wave-uniform: next tap
            ld_c0 = 0;
            if (++ld_kx == p.KW) {
                ld_kx = 0;
                ++ld_ky;
            }
            set_tap();
        }
    };
    auto swrite = [&](int buf, const Stage& R) {
        char* sAh = smem + buf * STAGE;
        char* sWh = sAh + A_BYTES;
        char* sAl = sWh + W_BYTES;  // PREC >= 2 only
        char* sWl = sAl + A_BYTES;  // PREC == 3 only
        u32x4 h[2], l[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = R.ra[2 * i + (e >> 1)][2 * (e & 1)], x1 = R.ra[2 * i + (e >> 1)][2 * (e & 1) + 1];
                const uint32_t hp = pack_h2<FMT>(x0, x1);
                h[i][e] = hp;
                if (PREC >= 2) l[i][e] = pack_h2<FMT>(x0 - h2_lo<FMT>(hp), x1 - h2_hi<FMT>(hp));
            }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = slot_swz(a_row, a_half * 2 + i);
            *reinterpret_cast<u32x4*>(sAh + off) = h[i];
            if (PREC >= 2) *reinterpret_cast<u32x4*>(sAl + off) = l[i];
        }
#pragma unroll
        for (int q = 0; q < W_PASSES; ++q) {
            const int r = w_row + RPP * q;
            if (r < BN) {
                const int off = slot_swz(r, w_piece);
                *reinterpret_cast<u32x4*>(sWh + off) = R.rwh[q];
                if (PREC == 3) *reinterpret_cast<u32x4*>(sWl + off) = R.rwl[q];
            }
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    auto compute = [&](int buf) {
        const char* sAh = smem + buf * STAGE;
        const char* sWh = sAh + A_BYTES;
        const char* sAl = sWh + W_BYTES;
        const char* sWl = sAl + A_BYTES;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            if constexpr (WM >= 4) {   // big tile: fragments of one A row-block at a time (register budget)
                bf16x8 bh[WN], bl[WN];
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int off = slot_swz((wn * WN + j) * 32 + frow, 2 * kc + fhalf);
                    bh[j] = *reinterpret_cast<const bf16x8*>(sWh + off);
                    if (PREC == 3) bl[j] = *reinterpret_cast<const bf16x8*>(sWl + off);
                }
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const int off = slot_swz((wm * WM + i) * 32 + frow, 2 * kc + fhalf);
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sAh + off);
                    bf16x8 al = ah;
                    if (PREC >= 2) al = *reinterpret_cast<const bf16x8*>(sAl + off);
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        if (PREC >= 2) acc[i][j] = mfma32h<FMT>(al, bh[j], acc[i][j]);   // small terms first
                        if (PREC == 3) acc[i][j] = mfma32h<FMT>(ah, bl[j], acc[i][j]);
                        acc[i][j] = mfma32h<FMT>(ah, bh[j], acc[i][j]);
                    }
                }
                continue;
            }
            bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int off = slot_swz((wm * WM + i) * 32 + frow, 2 * kc + fhalf);
                ah[i] = *reinterpret_cast<const bf16x8*>(sAh + off);
                if (PREC >= 2) al[i] = *reinterpret_cast<const bf16x8*>(sAl + off);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int off = slot_swz((wn * WN + j) * 32 + frow, 2 * kc + fhalf);
                bh[j] = *reinterpret_cast<const bf16x8*>(sWh + off);
                if (PREC == 3) bl[j] = *reinterpret_cast<const bf16x8*>(sWl + off);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    if (PREC >= 2) acc[i][j] = mfma32h<FMT>(al[i], bh[j], acc[i][j]);   // small terms first
                    if (PREC == 3) acc[i][j] = mfma32h<FMT>(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = mfma32h<FMT>(ah[i], bh[j], acc[i][j]);
                }
        }
    };
    // 256-thread tiles prefetch two chunks ahead (two register sets); the 512-thread 256x256 tile has no registers to
    // spare (128 accumulators + 48 fragment registers) and prefetches one chunk ahead.
    constexpr bool TWO_AHEAD = THREADS == 256;
    if constexpr (TWO_AHEAD) {
        gload(R0);            // chunk 0
        if (KT > 1) gload(R1);  // chunk 1
        swrite(0, R0);
        __syncthreads();
        for (int kt = 0; kt < KT; kt += 2) {
            // even step: chunk kt in LDS buffer 0, chunk kt+1 in R1, request chunk kt+2 into R0
            if (kt + 2 < KT) gload(R0);
            compute(0);
            if (kt + 1 < KT) swrite(1, R1);
            __syncthreads();
            if (kt + 1 >= KT) break;
            // odd step: chunk kt+1 in buffer 1, chunk kt+2 in R0, request chunk kt+3 into R1
            if (kt + 3 < KT) gload(R1);
            compute(1);
            if (kt + 2 < KT) swrite(0, R0);
            __syncthreads();
        }

    } else {
        gload(R0);
        swrite(0, R0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) gload(R0);
            compute(kt & 1);
            if (kt + 1 < KT) swrite((kt + 1) & 1, R0);
            __syncthreads();
        }
    }

    // ---- epilogue through LDS ------------------------------------------------------------------
    // The MFMA C layout gives a lane one column: 4-byte accesses at a row stride, 16*WM*WN store (+ residual
    // load) instructions per lane -- store-issue bound (same finding as gemm_bf16_t256.hip).  The tile is
    // transposed through the (now idle) stage buffers and leaves as 16-byte accesses, a wave covering whole
    // contiguous output rows; bias / activation / residuals are applied on float4s.
    float* stile = reinterpret_cast<float*>(smem);
    constexpr int EPI_ROWS = T::EPI_ROWS, NPASS = BM / EPI_ROWS;
    constexpr int C4 = BN / 4;
    // a thread's output segments all have the same 4 channels (THREADS % C4 == 0): the bias is loaded once.  Loaded per
    // segment it put an `s_waitcnt vmcnt(0)` -- which also waits for every earlier STORE -- in front of every segment of
    // the layers without a residual (same finding as gemm_bf16_t256.hip).
    static_assert(THREADS % C4 == 0, "a thread must keep its channel group over the epilogue passes");
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    {
        const float* bsrc = p.corr ? p.corr + 4L * p.Cout : p.bias;   // class 4 = interior
        if (bsrc) {
            const int nb = n0 + (tid % C4) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (nb + e < p.Cout) bias4[e] = bsrc[nb + e];
        }
    }
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
    if (NPASS == 1 || wm == pass) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stile[((wm * WM + i) * 32 + mfma32_row(r, lane) - pass * EPI_ROWS) * BN + (wn * WN + j) * 32 +
                          (lane & 31)] = acc[i][j][r];
    }
    __syncthreads();
    // Batches of EB output segments per thread: addresses and the residual loads of the whole batch first, stores after --
    // with load / add / store per segment every residual load waits behind the previous (possibly aliasing) store and the
    // epilogue of the residual layers ran with one access in flight (same finding as gemm_bf16_t256.hip).
    constexpr int EB = NPASS > 1 ? 2 : 4;   // accumulators of the later passes are still live in the two-pass (256-row) tile
#pragma unroll 1
    for (int idx0 = tid; idx0 < EPI_ROWS * C4; idx0 += EB * THREADS) {
        long pixs[EB];
        int ns[EB], cos[EB];
        bool ok[EB];
        f32x4 r1[EB], r2[EB], b4[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            const int idx = idx0 + u * THREADS;
            const int row = idx / C4, c4 = idx - row * C4;
            const long m = m0 + pass * EPI_ROWS + row;
            const int n = n0 + c4 * 4;
            ok[u] = idx < EPI_ROWS * C4 && m < p.M && n < p.Cout;
            const long mm = ok[u] ? m : m0;
            const int img = (int)(mm / hw);
            const int rem = (int)(mm - (long)img * hw);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            b4[u] = bias4;
            if (PREC == 2 && p.corr && ok[u]) {   // border placements: the correction of their class (conv_meancomp.hip)
                const int cls = (oy == 0 ? 0 : (oy == p.Ho - 1 ? 2 : 1)) * 3 + (ox == 0 ? 0 : (ox == p.Wo - 1 ? 2 : 1));
                if (cls != 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.Cout) b4[u][e] = p.corr[(long)cls * p.Cout + n + e];
                }
            }
            int co = n, py = 0, px = 0;
            if (p.ps > 1) {
                const int phase = n / p.cout_phys;
                co = n - phase * p.cout_phys;
                py = phase / p.ps;
                px = phase - py * p.ps;
            }
            pixs[u] = ((long)img * p.Hout + oy * p.osy + p.ooy + py) * p.Wout + ox * p.osx + p.oox + px;
            ns[u] = n;
            cos[u] = co;
            const bool full = ok[u] && (n + 3 < p.Cout);
#pragma unroll
            for (int e = 0; e < 4; ++e) { r1[u][e] = 0.f; r2[u][e] = 0.f; }
            if (full && p.res) {
                r1[u] = *reinterpret_cast<const f32x4*>(p.res + pixs[u] * p.ldr + co);
                if (p.res2) r2[u] = *reinterpret_cast<const f32x4*>(p.res2 + pixs[u] * p.ldr + co);
            }
        }
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            if (!ok[u]) continue;
            const int idx = idx0 + u * THREADS;
            const int row = idx / C4, c4 = idx - row * C4;
            const int n = ns[u], co = cos[u];
            const long pix = pixs[u];
            f32x4 v = *reinterpret_cast<const f32x4*>(stile + row * BN + c4 * 4);
            const bool full = (n + 3 < p.Cout);  // Cout % 4 != 0 only for the padded 42-channel bottleneck
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b4[u][e];
            if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
            } else if (p.act == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
            }
            if (full) {
                if (p.res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (p.relu_res ? fmaxf(r1[u][e], 0.f) : r1[u][e]) + r2[u][e];
                }
                *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + co) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.Cout) break;
                    float val = v[e];
                    if (p.res) {
                        const float rv = p.res[pix * p.ldr + co + e];
                        val += p.relu_res ? fmaxf(rv, 0.f) : rv;
                        if (p.res2) val += p.res2[pix * p.ldr + co + e];
                    }
                    p.y[pix * p.ldy + co + e] = val;
                }
            }
        }
    }
    if (NPASS > 1) __syncthreads();
    }
}

// y[m][n] = act(sum_z part[z][m][n] + bias[n]) + (relu_res ? max(res, 0) : res) + res2, slices added in z order
__global__ __launch_bounds__(256) void conv_splitk_finalize_kernel(const ConvParams p) {
    const int C4 = p.Cout >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.M * C4) return;
    const long m = idx / C4;
    const int n = (int)(idx - m * C4) * 4;
    const long slice = p.M * p.Cout;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.part + m * p.Cout + n);
    for (int z = 1; z < p.ksplit; ++z) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.part + z * slice + m * p.Cout + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += t[e];
    }
    if (p.corr) {   // PREC == 2: bias + mean-input correction of the placement's border class
        const int hw = p.Ho * p.Wo;
        const int rem = (int)(m % hw);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int cls = (oy == 0 ? 0 : (oy == p.Ho - 1 ? 2 : 1)) * 3 + (ox == 0 ? 0 : (ox == p.Wo - 1 ? 2 : 1));
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.corr + (long)cls * p.Cout + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += b[e];
    } else if (p.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += b[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (p.act == 1) v[e] = fmaxf(v[e], 0.f);
        else if (p.act == 2) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
        else if (p.act == 3) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
    }
    if (p.res) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.res + m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p.relu_res ? fmaxf(r[e], 0.f) : r[e];
        if (p.res2) {
            const f32x4 r2 = *reinterpret_cast<const f32x4*>(p.res2 + m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r2[e];
        }
    }
    *reinterpret_cast<f32x4*>(p.y + m * p.ldy + n) = v;
}

// split-K launch of the 128 x 128 tile: ks slices of the K loop per output tile + the finalize pass
template <int PREC>
int launch_splitk(const ConvParams& p, int ks, int kchunks, float* part, hipStream_t st) {
    using T = ConvTile<PREC, 2, 2, 2, 2>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_kernel<PREC, 2, 2, 2, 2, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T::SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    ConvParams q = p;
    q.tiles_n = (p.Cout + T::BN - 1) / T::BN;
    q.ksplit = ks; q.kchunks = kchunks; q.part = part;
    const long tiles_m = (p.M + T::BM - 1) / T::BM;
    hipLaunchKernelGGL((conv_igemm_kernel<PREC, 2, 2, 2, 2, true>), dim3((unsigned)(tiles_m * q.tiles_n), (unsigned)ks),
                       dim3(T::THREADS), T::SMEM, st, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const long n4 = p.M * (p.Cout >> 2);
    hipLaunchKernelGGL(conv_splitk_finalize_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, q);
    return 0;
}

template <int PREC, int WM, int WN, int WAVES_M, int WAVES_N>
int launch(const ConvParams& p, hipStream_t st) {
    using T = ConvTile<PREC, WM, WN, WAVES_M, WAVES_N>;
    static bool attr_set = false;
    if (!attr_set) {   // > 64 KiB of dynamic LDS needs the opt-in (one-time, like the GEMM kernels)
        hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_kernel<PREC, WM, WN, WAVES_M, WAVES_N>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T::SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    ConvParams q = p;
    q.tiles_n = (p.Cout + T::BN - 1) / T::BN;
    const long tiles_m = (p.M + T::BM - 1) / T::BM;
    hipLaunchKernelGGL((conv_igemm_kernel<PREC, WM, WN, WAVES_M, WAVES_N>), dim3((unsigned)(tiles_m * q.tiles_n)),
                       dim3(T::THREADS), T::SMEM, st, q);
    return 0;
}

}  // namespace

// 3x3 / stride 1 / pad 1 fast path with a spatial halo tile (conv3x3_halo.hip); -100: not applicable
int iggt_launch_conv3x3_halo(const float* x, int ldx, const void* w_hi, const void* w_lo, const float* bias,
                             const float* corr, const float* res, const float* res2, int ldr, float* y, int ldy, int Nimg,
                             int H, int W, int Cin, int Cout, int relu_in, int relu_res, int act, int prec, hipStream_t st);
// mean-input compensation of PREC = 2 (conv_meancomp.hip): channel means + the nine border-class correction vectors
long iggt_conv_meancomp_ws_bytes(int Cin, int Cout);
int iggt_launch_conv_meancomp(const float* x, int ldx, int Nimg, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int KH,
                              int KW, int stride, int pad_y, int pad_x, int relu_in, const void* dw, const float* bias,
                              void* ws, long ws_bytes, float** corr_out, int* uniform, hipStream_t st);

extern "C" int iggt_conv2d_nhwc_f32_ws(const float* x, int ldx, const void* w_hi, const void* w_lo, const float* bias,
                                       const float* res, const float* res2, int ldr, float* y, int ldy, int Nimg, int Hi,
                                       int Wi, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_y,
                                       int pad_x, int Hout, int Wout, int osy, int osx, int ooy, int oox,
                                       int cout_phys, int ps, int relu_in, int relu_res, int act, int prec,
                                       void* ws, long ws_bytes, void* stream) {
    if (Nimg <= 0 || Cin <= 0 || (Cin % BK) != 0 || Cout <= 0 || KH <= 0 || KW <= 0) return -1;
    if ((ldx % 4) != 0 || ldx < Cin) return -2;
    if ((ldy % 4) != 0 || (res && (ldr % 4) != 0) || (cout_phys % 4) != 0 && ps > 1) return -2;
    if (prec != 1 && prec != 2 && prec != 3) return -3;
    if (prec >= 2 && w_lo == nullptr) return -3;   // prec 3: bf16 lo plane; prec 2: bf16 residual W - fp16(W) (never an MFMA operand)
    if (res2 && !res) return -5;
    if (ps < 1 || cout_phys <= 0 || (ps > 1 && Cout != cout_phys * ps * ps)) return -4;
    ConvParams p;
    p.x = x; p.w_hi = (const bf16_t*)w_hi; p.w_lo = (const bf16_t*)w_lo; p.bias = bias; p.corr = nullptr; p.res = res; p.res2 = res2;
    p.y = y;
    p.Nimg = Nimg; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.ldx = ldx; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.stride = stride; p.pad_y = pad_y; p.pad_x = pad_x;
    p.Hout = Hout; p.Wout = Wout; p.ldy = ldy; p.ldr = ldr;
    p.osy = osy; p.osx = osx; p.ooy = ooy; p.oox = oox; p.cout_phys = cout_phys; p.ps = ps;
    p.relu_in = relu_in; p.relu_res = relu_res; p.act = act;
    p.M = (long)Nimg * Ho * Wo;
    p.tiles_n = 0;
    p.ksplit = 1; p.kchunks = 0; p.part = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (prec == 2) {
        // channel means of the input + correction vectors, into the head of the workspace (the split-K scratch follows)
        float* corr = nullptr;
        int uniform = 0;
        const int mrc = iggt_launch_conv_meancomp(x, ldx, Nimg, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad_y, pad_x, relu_in,
                                                  w_lo, bias, ws, ws_bytes, &corr, &uniform, st);
        if (mrc == -100) return -6;   // no nine-class description of this geometry / no workspace: the caller asks for prec 3
        if (mrc != 0) return mrc;
        const long used = (iggt_conv_meancomp_ws_bytes(Cin, Cout) + 255) & ~255L;
        ws = (char*)ws + used;
        ws_bytes -= used;
        if (uniform) p.bias = corr + 4L * Cout;   // 1 x 1 convolutions: one vector, the plain bias path
        else p.corr = corr;
    }
    if (prec >= 2 && KH == 3 && KW == 3 && stride == 1 && pad_y == 1 && pad_x == 1 && ps == 1 && osy == 1 && osx == 1 &&
        ooy == 0 && oox == 0 && Ho == Hi && Wo == Wi && Hout == Hi && Wout == Wi && cout_phys == Cout) {
        const int hrc = iggt_launch_conv3x3_halo(x, ldx, w_hi, w_lo, p.bias, p.corr, res, res2, ldr, y, ldy, Nimg, Hi, Wi, Cin,
                                                 Cout, relu_in, relu_res, act, prec, st);
        if (hrc != -100) {
            if (hrc) return hrc;
            IGGT_CHECK_LAUNCH();
            return 0;
        }
    }
    // 256x256 tile (8 waves, one workgroup per CU): every split activation is used for 256 output channels instead of
    // 128 -- half the loader work (loads, hi/lo split, LDS writes) per MFMA.  Only when the grid still fills the chip.
    static int big = -1, cus = 256;
    if (big < 0) {
        const char* e = getenv("IGGT_CONV_TILE256");
        big = (e && e[0] == '0') ? 0 : 1;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
    }
    const bool use_big = big && prec >= 2 && (Cout % 256) == 0 && ((p.M + 255) / 256) * (Cout / 256) >= 2L * cus;
    int rc;
    // Few output pixels AND a long K (1024 channels x 9 taps = 288 chunks on a 19^2 / 37^2 map): every workgroup of the few
    // tiles walks the whole K alone (~1 us per chunk: 300 us whatever the map size).  Split K over grid.y so that the
    // (tile, slice) pairs fill the chip; the partial tiles go through a scratch buffer and a finalize pass (fixed summation
    // order).  Plain convolutions only (no pixel shuffle / scatter).  IGGT_CONV_SPLITK=0 disables it.
    static int splitk = -1;
    if (splitk < 0) {
        const char* e = getenv("IGGT_CONV_SPLITK");
        splitk = (e && e[0] == '0') ? 0 : 1;
    }
    if (splitk && prec >= 2 && !use_big && ws != nullptr && ((uintptr_t)ws % 16) == 0 && ps == 1 && osy == 1 && osx == 1 &&
        ooy == 0 && oox == 0 && Hout == Ho && Wout == Wo && cout_phys == Cout && Cout >= 128 && (Cout % 4) == 0 &&
        (ldy % 4) == 0 && (!res || (ldr % 4) == 0)) {
        const long tiles = ((p.M + 127) / 128) * ((Cout + 127) / 128);
        const int KT = KH * KW * (Cin / BK);
        if (tiles * 2 <= cus && KT >= 48) {
            long ks = cus / tiles;
            if (ks > KT / 16) ks = KT / 16;
            if (ks > 16) ks = 16;
            while (ks > 1 && ks * p.M * Cout * 4 > ws_bytes) --ks;
            if (ks > 1) {
                const int kchunks = (int)((KT + ks - 1) / ks);
                ks = (KT + kchunks - 1) / kchunks;       // no empty slice
                const int rc2 = prec == 3 ? launch_splitk<3>(p, (int)ks, kchunks, (float*)ws, st)
                                          : launch_splitk<2>(p, (int)ks, kchunks, (float*)ws, st);
                if (rc2 != 0) return rc2;
                IGGT_CHECK_LAUNCH();
                return 0;
            }
        }
    }
    // Few output pixels (the 19^2 / 37^2 maps of a handful of views: the demo's 3-4 images, one rank of an 8-GPU run): the
    // 128 x 128 tiling leaves most CUs idle while every workgroup walks a long K (1024 channels x 9 taps = 288 chunks).
    // Narrower column tiles (128 x 64, 128 x 32) multiply the workgroups; the activation split per MFMA they add back is
    // cheaper than idle CUs.  IGGT_CONV_NARROW=0 restores the fixed choice.
    static int narrow = -1;
    if (narrow < 0) {
        const char* e = getenv("IGGT_CONV_NARROW");
        narrow = (e && e[0] == '0') ? 0 : 1;
    }
    int bn = Cout > 64 ? 128 : Cout > 32 ? 64 : 32;
    if (narrow && prec >= 2 && !use_big) {
        const long tiles_m = (p.M + 127) / 128;
        while (bn > 32 && tiles_m * ((Cout + bn - 1) / bn) * 4 < 3L * cus) bn >>= 1;   // fewer than 3/4 of the CUs busy
    }
    if (prec == 3) {
        if (use_big) rc = launch<3, 4, 2, 2, 4>(p, st);
        else if (bn == 128) rc = launch<3, 2, 2, 2, 2>(p, st);
        else if (bn == 64) rc = launch<3, 1, 2, 4, 1>(p, st);
        else rc = launch<3, 1, 1, 4, 1>(p, st);
    } else if (prec == 2) {
        if (use_big) rc = launch<2, 4, 2, 2, 4>(p, st);
        else if (bn == 128) rc = launch<2, 2, 2, 2, 2>(p, st);
        else if (bn == 64) rc = launch<2, 1, 2, 4, 1>(p, st);
        else rc = launch<2, 1, 1, 4, 1>(p, st);
    } else {
        if (Cout > 64) rc = launch<1, 2, 2, 2, 2>(p, st);
        else if (Cout > 32) rc = launch<1, 1, 2, 4, 1>(p, st);
        else rc = launch<1, 1, 1, 4, 1>(p, st);
    }
    if (rc != 0) return rc;
    IGGT_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Bilinear resize, align_corners=True, NHWC fp32 (reference: custom_interpolate, dpt_head.py:484-509,
// used by FeatureFusionBlock 471-478 and the head tails 251-256 / part_head.py 228-238).
// Optional separable additive position map (xpart [Wo][C/2] | ypart [Ho][C/2]) fused in.
namespace {
struct ResizeParams {
    const float* x; float* y;
    int N, Hi, Wi, Ho, Wo, C, ldx, ldy;
    float sy, sx;
    const float* xpart; const float* ypart;  // optional
};

// One output ROW (n, oy) per blockIdx.y / .z, threads over (ox, 4-channel group) with 32-bit indices: the first version walked a
// flat 64-bit index and paid three 64-bit divisions per 16 bytes written -- ALU-bound at 2 TB/s on the 148^2 -> 296^2 upsampling of
// a 32-view head pass (2.9 GB written).  Row coordinates and weights are uniform per workgroup (scalar).
__global__ __launch_bounds__(256) void bilinear_ac_nhwc_kernel(const ResizeParams p) {
    const unsigned c4n = (unsigned)p.C / 4u;
    const unsigned row_items = (unsigned)p.Wo * c4n;
    const int oy = blockIdx.y, n = blockIdx.z;
    // source coordinate = index * scale in fp32; the FRACTION is taken from the exact product (one fused multiply-subtract,
    // stated explicitly so that it does not depend on what the compiler contracts): closest to the real-number coordinate (1.3e-6
    // of a pixel at 296 -> 518; rounding the product to fp32 first: 1.6e-5).  tests/test_conv_gpu.py compares against fp64.
    const float fy = __fmul_rn((float)oy, p.sy);
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < p.Hi - 1 ? 1 : 0);
    const float ly = __fmaf_rn((float)oy, p.sy, -(float)y0), hy = 1.f - ly;
    const float* img = p.x + (long)n * p.Hi * p.Wi * p.ldx;
    const float* row0 = img + (long)y0 * p.Wi * p.ldx;
    const float* row1 = img + (long)y1 * p.Wi * p.ldx;
    float* out_row = p.y + ((long)n * p.Ho + oy) * (long)p.Wo * p.ldy;
    const int half = p.C / 2;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < row_items; i += gridDim.x * 256u) {
        const unsigned ox = i / c4n;
        const int c = (int)(i - ox * c4n) * 4;
        // PyTorch upsample_bilinear2d (align_corners): src = dst * (in-1)/(out-1), lambda from the floor
        const float fx = __fmul_rn((float)(int)ox, p.sx);
        const int x0 = (int)fx;
        const int x1 = x0 + (x0 < p.Wi - 1 ? 1 : 0);
        const float lx = __fmaf_rn((float)(int)ox, p.sx, -(float)x0), hx = 1.f - lx;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(row0 + (long)x0 * p.ldx + c);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(row0 + (long)x1 * p.ldx + c);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(row1 + (long)x0 * p.ldx + c);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(row1 + (long)x1 * p.ldx + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
        if (p.xpart) {
            const float* t = (c < half) ? (p.xpart + (long)ox * half + c) : (p.ypart + (long)oy * half + (c - half));
            const f32x4 a = *reinterpret_cast<const f32x4*>(t);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += a[e];
        }
        *reinterpret_cast<f32x4*>(out_row + (long)ox * p.ldy + c) = o;
    }
}
}  // namespace

extern "C" int iggt_bilinear_ac_nhwc_f32(const float* x, int ldx, float* y, int ldy, int N, int Hi, int Wi, int Ho,
                                         int Wo, int C, const float* xpart, const float* ypart, void* stream) {
    if (N <= 0 || (C % 8) != 0 || (ldx % 4) || (ldy % 4)) return -1;
    ResizeParams p;
    p.x = x; p.y = y; p.N = N; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo; p.C = C; p.ldx = ldx; p.ldy = ldy;
    p.sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    p.sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    p.xpart = xpart; p.ypart = ypart;
    if (Ho > 65535 || N > 65535 || (long)Wo * (C / 4) >= (1L << 31)) return -1;
    long bx = ((long)Wo * (C / 4) + 255) / 256;
    if (bx > 64) bx = 64;                       // a thread then takes several (ox, channel group) items of its row
    hipLaunchKernelGGL(bilinear_ac_nhwc_kernel, dim3((unsigned)bx, (unsigned)Ho, (unsigned)N), dim3(256), 0, (hipStream_t)stream, p);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_conv2d_nhwc_f32(const float* x, int ldx, const void* w_hi, const void* w_lo, const float* bias,
                                    const float* res, const float* res2, int ldr, float* y, int ldy, int Nimg, int Hi, int Wi,
                                    int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_y,
                                    int pad_x, int Hout, int Wout, int osy, int osx, int ooy, int oox,
                                    int cout_phys, int ps, int relu_in, int relu_res, int act, int prec,
                                    void* stream) {
    return iggt_conv2d_nhwc_f32_ws(x, ldx, w_hi, w_lo, bias, res, res2, ldr, y, ldy, Nimg, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW,
                                   stride, pad_y, pad_x, Hout, Wout, osy, osx, ooy, oox, cout_phys, ps, relu_in, relu_res,
                                   act, prec, nullptr, 0, stream);
}
