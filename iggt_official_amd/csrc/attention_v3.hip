// Flash attention d=64, production kernel ("v3").
//
// Same data layout, LDS images, swapped-QK^T / in-register-P scheme as attention.hip (the v1 baseline, see there
// for the MFMA/LDS layout derivations).  What changed, each step A/B-measured on the 32-view global attention
// (N = 43 968, 16 heads; profiles/r01_attn_variants.txt, profiles/r01_attn_pmc.txt):
//   v1  805 TF/s  PMC: matrix pipe ~35 % busy, VALU issue ~60 % busy, and the two nearly serialised because a
//                 wave's MFMAs (QK^T, PV) and its softmax VALU work sit in separate dependent phases.
//   +   staggered query blocks: the two 32-row blocks of a wave are offset so each MFMA group has independent
//       VALU work of the other block in the same basic block
//           [QK^T(q0)] [max(q0)] | [QK^T(q1) || exp/pack(q0)] [max(q1)] | [PV(q0) || exp/pack(q1)] | [PV(q1)]
//       and deferred-max rescale: m only advances (O, l rescaled) when a row's tile max exceeds it by > 2^THR
//       (wave vote, rare after the first tiles; guide T13); P <= 2^THR is harmless in bf16/fp32      -> 860 TF/s
//   +   LDS-DMA staging (global_load_lds_dwordx4, swizzles applied to the source address, no staging VGPRs,
//       no ds_write) and 128-key macro tiles (one barrier + one DMA drain per 128 keys)               -> 910 TF/s
//   +   row-max exchange with lane^32 via v_permlane32_swap (VALU) instead of ds_bpermute (LDS pipe)  -> 940 TF/s
// Tried and rejected (measured slower): 8-wave ping-pong specialisation (attention.hip, 725), s_setprio around
// the MFMA groups (855), row sums on the matrix pipe with a ones fragment (880), side-stream tail balancing,
// "optimistic" exponentiation against the stale running max with a post-hoc sum check and a rare redo path instead
// of the per-tile max tree (-15 % VALU work, but the extra branch splits the block in which QK^T(q1) and exp(q0)
// interleave: 809 vs 935 TF/s, fp16); a mixed-tile launch that hands the last, 37 %-full round of 256-row tiles to
// 128-row tiles (correct, but 1.5 % slower: a CU left with ONE resident workgroup already runs it ~1.7x faster, so
// the partial round costs ~0.6 of a full one, not 1.0).
// Round 2, static-bound kernel: row sums on the matrix pipe again, this time almost free -- a packed P fragment read as the
// B operand of a 16x16x32 MFMA against a 0/1 selector A (row 0 = ones on k-groups {0, 2}, row 1 = on {1, 3}) adds its 16 keys
// to D[0][n] = l(query n), D[1][n] = l(query n + 16): 4 four-pass MFMAs per 64-key tile instead of 36 v_add, a third of the
// VALU instructions gone, results identical -- and 7.02 vs 7.09 ms (fp16), 6.75 vs 6.68 ms (bf16): the loop is not bound by VALU
// throughput (VALU pipe 74 % busy, matrix pipe 60 %) but by the in-order issue of each wave.  128-row tiles at three
// waves per SIMD instead of the staggered pair: 1 000 vs 1 100 TF/s (probes/attn_tile_codes.py).
// Also measured (probes/legacy/attention_r2_v5_tile_pipelined.hip): the four MFMA groups re-ordered across tiles so that EVERY
// group has half an exp/pack unit beside it (QK^T(q0, t+1) before PV(q1, t); 238 VGPRs, no spill): 7.68 vs 7.28 ms; both query
// blocks sharing every K / V fragment (half the LDS reads, no intra-wave overlap): 7.70 ms.  Three instruction orders within
// 5 %, bf16 operands 7 % faster than fp16 on the identical instruction stream, effective clock 1.77 GHz: the kernel runs at
// the chip's power limit for 16-bit MFMA on real data.
// Ablation: the same kernel without any softmax VALU work reaches 1 210 TF/s -- the d = 64 softmax (64 exp +
// ~140 other VALU ops per 32 MFMA) is what separates this kernel from the matrix-pipe limit.
//
// Round 2: STATIC-BOUND softmax (template parameter STATIC, "v4").  q and k of the aggregator blocks leave the fused
// q/k-LayerNorm + RoPE kernel with known norms, so by Cauchy-Schwarz  s_ij = q^_i . k^_j <= max|q^| max|k^| =: c_h  holds
// for every score of head h (q^ already carries softmax scale * log2 e; the two maxima are by-products of that kernel,
// csrc/elementwise.hip).  The numerators are then  P_ij = 2^(s_ij - c_h + SHIFT)  <= 2^SHIFT  with a shift that is known
// BEFORE the first tile: no row-max tree, no lane exchange, no wave vote, no rescale of O and l, no branch in the tile
// loop -- and the subtraction itself is free, because -(c_h - SHIFT) enters as the C operand of the first QK^T MFMA
// of every score block (one wave-uniform 16-register vector).  Per 32 x 32 score block that leaves 16 v_exp + 16 v_add
// (row sum) + 8 v_cvt_pk of VALU work beside 8 MFMAs (dynamic kernel: + 16 v_fma + 16 v_max + exchange + vote).
// What the bound cannot give is a LOWER limit of a row's true maximum: a row whose scores all sit far below c_h ends
// with numerators in fp16's subnormal range.  Such rows are detected after the loop by their row sum
// (l_i < Nk * 2^-13, attention_common.h) and their 256-row tile is flagged; the dynamic kernel (same file, gated on the flag
// array) then recomputes exactly the flagged tiles.  With LayerNorm-ed q, k the slack is ~5 bits of the 29 available.
#include <type_traits>

#include "attention_common.h"
#include "../../include/iggt_hip.h"

using namespace iggt_attn;

namespace {

constexpr float DEFER_THR = 4.0f;  // log2 units: P <= 16

// A/B build of the estimated-shift instantiation (probes/build_alt.py est_nodelta): IGGT_EST_NODELTA gives a lane's two rows
// one shared shift (the larger), as under the norm bound.  (A second variant -- zero accumulator input, both blocks' shifts by
// packed adds -- measured 8.12-8.15 ms against 7.87 and was removed: profiles/r04_attn_est_ab.txt)
#ifdef IGGT_EST_NODELTA
constexpr bool EST_NODELTA = true;
#else
constexpr bool EST_NODELTA = false;
#endif
// Largest difference (bits) between the shifts of a lane's two rows that a wave leaves uncorrected.  The rows are dealt in
// ascending shift order, so the second row's own shift is the LARGER one: computed under the first row's it keeps
// 2^(3 + delta) for its sampled maximum (the acceptance test only gets easier) and loses delta of its 13 bits of overflow
// headroom -- an overflow is detected and the row handed over, never wrong.  Measured (profiles/r04_attn_est_ab.txt, est forced,
// N = 43 968): 0.5 -> 8 bits takes 0.2-0.3 ms off the three adversarial regimes for 1-6 % more rows handed over; 12 is no better.
#ifndef IGGT_EST_DELTA_MAX
#define IGGT_EST_DELTA_MAX 8.0f
#endif

#ifdef IGGT_ATTN_NO_PIN   // A/B builds only (probes/build_alt.py)
constexpr bool PIN_DEFAULT = false;
#else
constexpr bool PIN_DEFAULT = true;
#endif

// Adaptive switch, evaluated by workgroup 0 of the gated online-max pass (it runs right behind the static-bound kernel or
// the combine kernel on the same stream, so the flags are final and nobody reads the guard word any more in this call):
// count the flagged tiles and decide what the NEXT call of this block does (attention_common.h AttnParams::guard).
IGGT_DEVINL void guard_update(const AttnParams& p, int nwork, int rows_per_item, char* smem) {
    int n = 0, nrows = 0;
    for (int i = threadIdx.x; i < nwork; i += 256) n += p.flags[i] != 0;
    if (p.est_ws != nullptr) {   // rows handed over one by one (estimated-shift / row-granular launches): in work items
        const int* rowcount = est_view(p).rowcount;
        for (int i = threadIdx.x; i < p.B * p.H; i += 256) {
            const int c = rowcount[i];
            nrows += c;
            n += (c + rows_per_item - 1) / rows_per_item;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n += __shfl_xor(n, o, 64);
        nrows += __shfl_xor(nrows, o, 64);
    }
    int* red = reinterpret_cast<int*>(smem);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = n;
        red[4 + (threadIdx.x >> 6)] = nrows;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        n = red[0] + red[1] + red[2] + red[3];
        nrows = red[4] + red[5] + red[6] + red[7];
        const int g0 = p.guard[0];
        const bool skipped = guard_skips(p.guard, p.guard_prev);
        // a launch WITHOUT the estimated-shift workspace (partial / combine path, key-range split) runs the norm bound whatever
        // the word says: it must not reset the mode the call site's one-pass launches have settled in
        int mode = p.est_ws != nullptr ? guard_mode(p) : p.guard[4];
        int g;
        if (skipped) {
            g = (g0 < 0 ? p.guard_retry : g0) - 1;                    // this call ran the online-max kernel only: count down
        } else if (mode == 0 && p.est_ws != nullptr && n > 0) {
            // the norm bound is loose here: estimate the shift from now on.  ANY flagged tile decides: under the norm bound a
            // flagged tile is a full-length online-max workgroup, and a handful of those already cost a whole extra round
            // (~1.5 ms at N = 43 968: bf16 "affine", 48 of 2 752 tiles flagged, 8.56 ms against 7.50 ms in estimated mode),
            // more than the estimated-shift machinery costs when it hands nothing over (~0.5 ms)
            mode = 1;
            g = 0;
        } else if ((long)n * 8 > (long)nwork) {                       // > 1/8 of the work redone: static + redo loses
            g = p.guard_retry;                                        // nothing static helps: online-max only for a while
        } else {
            g = 0;
        }
        p.guard[0] = g;
        p.guard[1] = skipped ? -1 : n;
        p.guard[2] = nwork;
        p.guard[3] = p.guard[3] + 1;
        p.guard[4] = mode;
        p.guard[5] = skipped ? -1 : nrows;
    }
    __syncthreads();   // smem is reused by the K / V staging below
}

// PART (static bound only): the workgroup covers ONE of p.ksplit equal ranges of the key macro tiles and writes a partial
// result -- O_s / l_s as a 16-bit row and l_s in fp32 -- that attn_combine_kernel folds: with the static bound every range
// uses the same shift, so partial sums simply add (no running-max bookkeeping between ranges).  Used (a) to balance small
// grids: the per-rank global attention of an 8-GPU run has 352 256-row tiles for 512 workgroup slots; four key ranges make
// 1 408 quarter-length workgroups (0.75 instead of 1.0 tile-times), and (b) to start on a rank's own keys while the K/V
// all-gather of the other ranks is still in flight (iggt_official_amd/dist.py).
// EST (static bound, one pass only; round 4): the shift of every query row comes from the pre-pass table (attention_est.hip) instead
// of the norms, rows are handed to the online-max pass one by one (rowflag) instead of as 256-row tiles, and a non-finite
// accumulator (an fp16 numerator above the range) marks a row like a row sum below the threshold does.  A separate
// instantiation on purpose: the norm-bound kernel sits at the 256-register limit, and the same code with both paths behind a
// run-time switch put a 16-byte spill reload into its tile loop (8-11 spilled registers instead of 2 outside the loop).  Both
// instantiations are launched; each returns at once unless the adaptive-switch word names its mode.
// LIST (static bound, PART, 128-row tiles only; second chance of attention_est.hip): the workgroup's rows are positions
// [qt * 128, ...) of the (batch, head)'s list of handed-over rows, its shift the exact row maximum (+ 1) over all keys, its
// keys range ks of EST_KS2, its results go to list-position slots.
template <int QB, int KVM, int FMT, bool STATIC, bool PIN = PIN_DEFAULT, bool PART = false, bool EST = false, bool LIST = false>
__global__ __launch_bounds__(256, 2) void flash_attn_d64_v3_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * KVM * BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;

    // LIST launches are sparse -- most workgroups find nothing to do -- and the rows that are there concentrate on few heads
    // (measured: 3 575 of 5 888 listed rows in ONE head): under the XCD-chunked order all of that head's work items sit in one
    // XCD's contiguous share of the list and run on its 32 CUs alone (7 rounds, 369 us); in launch order they spread over the chip
    const int work = LIST ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    // online-max pass in LIST mode (round 4): the work item recomputes rows [qt * 128 QB, ...) of the (batch, head)'s list of
    // flagged rows (attention_est.hip attn_rowlist_kernel) instead of a contiguous query tile
    int nlist = -1;
    const int* rlist = nullptr;
    if constexpr (!STATIC) {
        // fallback pass behind the static-bound kernel: only the flagged query tiles / rows are recomputed
        if (p.flags != nullptr) {
            if (p.guard != nullptr && blockIdx.x == 0) guard_update(p, (int)gridDim.x, 128 * QB, smem);
            if (p.flags[work] == 0) {
                if (p.est_ws == nullptr) return;
                const EstView ev = est_view(p);
                const int bh_ = work / p.qtiles;
                nlist = ev.rowcount[bh_];
                // short lists were served by the second chance (attention_est.hip); long ones are recomputed here
                if (nlist <= ev.NqL || (work % p.qtiles) * (128 * QB) >= nlist) return;
                rlist = ev.rowlist + (long)bh_ * p.Nq;
            }
        }
    } else {
        // adaptive switch: a block whose tiles kept failing the acceptance test goes straight to the online-max kernel
        if (guard_skips(p.guard, p.guard_prev)) {
            if constexpr (!PART && !EST) {
                if (tid == 0) p.flags[work] = 1;
            }
            return;
        }
        if constexpr (!PART) {
            if ((guard_mode(p) == 1) != EST) return;   // the other instantiation's turn
        }
    }
    const int qt = work % p.qtiles;
    int bh = work / p.qtiles, ks = 0;
    int nk = p.Nk;          // keys this workgroup sees (segment mode: its segment, addressed from 0)
    long kseg0 = 0;         // first key row of the segment
    if constexpr (PART) {   // work = ((b, h), key range, q tile): the q tiles of one key range stay adjacent (same K/V in L2)
        ks = bh % p.ksplit;
        bh /= p.ksplit;
        if (p.seg_len > 0) {
            if (ks == p.skip_seg) return;
            kseg0 = (long)ks * p.seg_len;
            nk = p.Nk - (int)kseg0 < p.seg_len ? p.Nk - (int)kseg0 : p.seg_len;
        }
    }
    const int h = bh % p.H, b = bh / p.H;
    if constexpr (LIST) {
        const EstView ev = est_view(p);
        nlist = ev.rowcount[bh];
        if (nlist == 0 || nlist > ev.NqL || qt * (128 * QB) >= nlist) return;
        rlist = ev.rowlist + (long)bh * p.Nq;
    }
    const bf16_t* qb_ptr = p.q + (long)b * p.q_bs + h * 64;
    const bf16_t* kb_ptr = p.k + (long)b * p.k_bs + kseg0 * p.k_rs + h * 64;
    const bf16_t* vb_ptr = p.v + (long)b * p.v_bs + kseg0 * p.v_rs + h * 64;
    bf16_t* ob_ptr = p.o + (long)b * p.o_bs + h * 64;

    const int q_base = qt * (128 * QB) + wave * (32 * QB);
    bf16x8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qr = q_base + qb * 32 + frow;
        if constexpr (EST && QB == 2) {   // the tile's rows are dealt to the lanes in the order of their shift (attention_est.hip)
            const EstView ev = est_view(p);
            qr = ev.slotrow[(long)bh * ev.NqS + qr];
        }
        qr = (unsigned)qr < (unsigned)p.Nq ? qr : p.Nq - 1;   // (unsigned: a slot table nobody wrote must not index backwards)
        if constexpr (!STATIC || LIST) {
            if (rlist != nullptr) qr = rlist[q_base + qb * 32 + frow < nlist ? q_base + qb * 32 + frow : nlist - 1];
        }
        const bf16_t* src = qb_ptr + (long)qr * p.q_rs + 8 * fhalf;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) qf[qb][kc] = *reinterpret_cast<const bf16x8*>(src + 16 * kc);
    }

    // ---- LDS-DMA staging: a 64-key K (or V) tile is 8 chunks of 1 KiB (8 rows x 128 B); wave w moves chunks
    //      2w, 2w+1 of K and of V.  Lane l fills row 8j + l/8, 16-B slot l%8; the row-image swizzles are applied
    //      to the SOURCE piece (K: slot ^ ((row>>1)&7); V: 32-B chunk ^ (row&2)).  Rows past Nk are clamped to the
    //      last valid row (finite data; their scores are masked to -inf, so they contribute exactly 0).
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int c_row = lane >> 3, c_pos = lane & 7;
    // per-lane source pointers of macro tile 0 (row r of each 64-key tile, swizzled piece); later tiles add a
    // wave-uniform offset, only the last (ragged) macro tile re-derives clamped rows.
    const bf16_t* ksrc[2];
    const bf16_t* vsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 8 + c_row;
        ksrc[i] = kb_ptr + (long)r * p.k_rs + (c_pos ^ ((r >> 1) & 7)) * 8;
        vsrc[i] = vb_ptr + (long)r * p.v_rs + ((((c_pos >> 1) ^ (r & 2)) << 1) | (c_pos & 1)) * 8;
    }
    auto dma = [&](int mt, int buf) {   // macro tile mt -> buffer buf
        char* base = smem + buf * (KVM * BUF_BYTES);
        const bool ragged = (mt + 1) * (KVM * KV_TILE) > nk;
#pragma unroll
        for (int sub = 0; sub < KVM; ++sub) {
            char* sK = base + sub * BUF_BYTES;
            char* sV = sK + K_BYTES;
            const int kv0 = (mt * KVM + sub) * KV_TILE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int chunk = (2 * wave + i) * 1024;
                const bf16_t* ks = ksrc[i] + (long)kv0 * p.k_rs;
                const bf16_t* vs = vsrc[i] + (long)kv0 * p.v_rs;
                if (ragged) {   // clamp rows past the end to the last valid row (scores are masked to -inf)
                    const int r = (2 * wave + i) * 8 + c_row;
                    const int over = kv0 + r - (nk - 1);
                    if (over > 0) {
                        ks -= (long)over * p.k_rs;
                        vs -= (long)over * p.v_rs;
                    }
                }
                __builtin_amdgcn_global_load_lds((gptr_t*)ks, (lptr_t*)(sK + chunk), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t*)vs, (lptr_t*)(sV + chunk), 16, 0, 0);
            }
        }
    };

    f32x16 o[QB][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -1e30f;
        l_run[qb] = 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][dh][r] = 0.f;
    }
    const int tr_i = lane & 15, tr_g = (lane >> 4) & 1;
    const float c = p.scale_log2;
    // static bound: the (negated) shift enters through the accumulator input of the first QK^T MFMA of a score block
    f32x16 cinit;
    float est_delta = 0.f;   // EST, QB = 2: shift of the lane's second row minus that of its first
    if constexpr (STATIC) {
        // PER-ROW bound: a lane owns one query column of the swapped score block, so the shift may depend on the lane's
        // query: s_ij <= |q^_i| max_j |k^_j|.  The norm is taken from the very fragments the MFMAs consume (this lane holds
        // half of the row, lane ^ 32 the other half).  The two query blocks of a lane (rows r and r + 32) share one C-operand
        // vector -- a second one would cost 16 more VGPRs in a kernel that sits at the 256-register limit of two waves per
        // SIMD -- so the lane uses the larger of its two norms.  Rows with a small |q^| (most rows, when a few outlier tokens
        // dominate max_i |q^_i|) no longer inherit the outliers' shift.
        float shift;
        if constexpr (LIST) {   // exact row maximum over all keys (the largest of the key ranges' maxima), + 1: numerators <= 2^14
            const EstView ev = est_view(p);
            const int lp = q_base + frow < nlist ? q_base + frow : nlist - 1;
            shift = -INFINITY;
            for (int s2 = 0; s2 < EST_KS2; ++s2) shift = fmaxf(shift, ev.pmax[((long)s2 * p.B * p.H + bh) * ev.NqL + lp]);
            shift += 1.0f - (FMT == FMT_F16 ? STATIC_SHIFT_F16 : 0.f);
        } else if constexpr (EST) {   // min(norm bound, sampled row maximum + headroom) per row, from the pre-pass
            const float* rs = reinterpret_cast<const float*>(p.est_ws) + (long)bh * p.Nq;
            if constexpr (QB == 2) {
                // each of the lane's two rows keeps ITS shift: the accumulator-input vector carries the first row's, the
                // difference is subtracted from the second block's scores before the exponential (16 packed adds per 64-key
                // tile; tile_loop below).  A shared shift -- the larger of the two, as under the norm bound -- flushes the other
                // row's numerators whenever the two differ by more than a few bits: 38 % of the rows of the "sinks" regime were
                // handed over for that
                const EstView ev = est_view(p);
                const int* sr = ev.slotrow + (long)bh * ev.NqS + q_base + frow;
                const int r0 = sr[0], r1 = sr[32];
                shift = rs[(unsigned)r0 < (unsigned)p.Nq ? r0 : p.Nq - 1] - EST_BIAS;
                const float sh1 = rs[(unsigned)r1 < (unsigned)p.Nq ? r1 : p.Nq - 1] - EST_BIAS;
                if constexpr (EST_NODELTA) shift = fmaxf(shift, sh1);
                else est_delta = sh1 - shift;
            } else {
                const int r0 = q_base + frow;
                shift = rs[r0 < p.Nq ? r0 : p.Nq - 1] - EST_BIAS;
            }
            shift -= (FMT == FMT_F16 ? STATIC_SHIFT_F16 : 0.f);
        } else {
            float n2 = 0.f;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float a2 = 0.f;
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const u32x4 w = __builtin_bit_cast(u32x4, qf[qb][kc]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = h2_lo<FMT>(w[e]), hi = h2_hi<FMT>(w[e]);
                        a2 += lo * lo + hi * hi;
                    }
                }
                a2 += __shfl_xor(a2, 32, 64);
                n2 = fmaxf(n2, a2);
            }
            const float kmax = (PART && p.seg_kmax != nullptr) ? p.seg_kmax[ks * 32 + 16 + h] : p.qkmax[16 + h];
            shift = sqrtf(n2) * kmax * 1.00002f + 1e-3f - (FMT == FMT_F16 ? STATIC_SHIFT_F16 : 0.f);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[r] = -shift;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
    }
    const int NT = (nk + KV_TILE - 1) / KV_TILE;

    // ---- building blocks ---------------------------------------------------------------------
    // lane-dependent LDS offsets, hoisted: row = (multiple of 32) + frow, so the K swizzle key ((row >> 1) & 7) and the
    // V chunk flip (row & 2) depend on the lane only; everything else is an immediate.
    int koff[4], voff[2];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) koff[kc] = frow * 128 + ((((2 * kc + fhalf) ^ (frow >> 1)) & 7) << 4);
    {
        const int vr = 4 * fhalf + (tr_i >> 2);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) voff[dh] = vr * 128 + ((((dh * 2 + tr_g) ^ (vr & 2))) << 5) + 8 * (tr_i & 3);
    }
    auto qk = [&](const char* sK, int qb, f32x16 (&s)[2]) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kvh * 4096 + koff[kc]);
                s[kvh] = mfma32h<FMT>(kf, qf[qb][kc], kc == 0 ? (STATIC ? cinit : zero) : s[kvh]);
            }
        }
    };
    auto mask_tail = [&](int t, f32x16 (&s)[2]) {
        const int kv0 = t * KV_TILE + 4 * fhalf;
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + kvh * 32 + (r & 3) + 8 * (r >> 2);
                if (kv >= nk) s[kvh][r] = -INFINITY;
            }
    };
    // row max of the tile; advance m (rescaling O and l) only if some row needs it
    auto update_max = [&](int qb, const f32x16 (&s)[2]) {
        float mx = s[0][0];
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kvh][r]);
        {   // exchange with lane^32 on the VALU (v_permlane32_swap) instead of an LDS-pipe ds_bpermute
            const uint32_t bits = __builtin_bit_cast(uint32_t, mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1])) * c;
        }
        if (__any(mx > m_run[qb] + DEFER_THR)) {
            const float m_new = fmaxf(m_run[qb], mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
            m_run[qb] = m_new;
            l_run[qb] *= alpha;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][dh][r] *= alpha;
        }
    };
    auto exp_pack = [&](int qb, f32x16 (&s)[2], bf16x8 (&pf)[2][2], auto use_delta) {
        const float m = m_run[qb] - (FMT == FMT_F16 ? P_SHIFT_F16 : 0.f);
        float ls0 = 0.f, ls1 = 0.f, ls2 = 0.f, ls3 = 0.f;
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh) {
            if (decltype(use_delta)::value && qb == 1) {
                // the accumulator holds s - c + SHIFT with c of the FIRST block's row: move to this row's shift, two scores per
                // v_pk_add_f32 (written as scalar subtractions the compiler emitted 32 v_sub_f32 per tile)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 d2 = {est_delta, est_delta};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 t = {s[kvh][r], s[kvh][r + 1]};
                    t = t - d2;
                    s[kvh][r] = t[0];
                    s[kvh][r + 1] = t[1];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (STATIC) {
                    s[kvh][r] = __builtin_amdgcn_exp2f(s[kvh][r]);   // the accumulator already holds s - c + SHIFT
                } else {
                    const float a = __builtin_fmaf(s[kvh][r], c, -m);
                    s[kvh][r] = __builtin_amdgcn_exp2f(a);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                ls0 += s[kvh][r];
                ls1 += s[kvh][r + 1];
                ls2 += s[kvh][r + 2];
                ls3 += s[kvh][r + 3];
            }
            pf[kvh][0] = pack8h<FMT>(s[kvh], 0);
            pf[kvh][1] = pack8h<FMT>(s[kvh], 8);
        }
        l_run[qb] += (ls0 + ls1) + (ls2 + ls3);
    };
    auto pv = [&](const char* sV, int qb, const bf16x8 (&pf)[2][2]) {
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    // keys of this lane half: kvb + {0..3} (elements 0-3) and kvb + 8 + {0..3} (elements 4-7)
                    typedef __attribute__((address_space(3))) short4v lds_s4;
                    const char* base = sV + (kvh * 32 + 16 * cc) * 128 + voff[dh];
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(base));
                    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(base + 8 * 128));
                    typedef short short8v __attribute__((ext_vector_type(8)));
                    const short8v v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[qb][dh] = mfma32h<FMT>(__builtin_bit_cast(bf16x8, v8), pf[kvh][cc], o[qb][dh]);
                }
    };

    // keep the packed numerators of q-block 0 "used" right behind exp/pack: without this hipcc sinks the whole exp/pack(q0)
    // group below the run-time tail branch that follows QK^T(q1), i.e. out of the basic block in which it is meant to
    // run beside those MFMAs (round-1 code: QK^T(q1) ran bare and PV(q0), PV(q1) shared one block with ALL the VALU work)
    auto pin = [&](bf16x8 (&pf)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x4 w = __builtin_bit_cast(u32x4, pf[i][j]);
                asm volatile("" : "+v"(w));
                pf[i][j] = __builtin_bit_cast(bf16x8, w);
            }
    };
    const int NMT_all = (NT + KVM - 1) / KVM;
    int mt0 = 0, NMT = NMT_all;
    if constexpr (PART) {
        if (p.seg_len <= 0) {
            mt0 = (int)((long)ks * NMT_all / p.ksplit);
            NMT = (int)((long)(ks + 1) * NMT_all / p.ksplit);
        }
    }
    // the tile loop, in two copies for the estimated-shift kernel: with and without the second block's shift correction (chosen
    // per wave, below)
    auto tile_loop = [&](auto use_delta) {
        dma(mt0, mt0 & 1);
        wg_barrier_counted();  // vmcnt(0) + barrier: first macro tile resident
        for (int mt = mt0; mt < NMT; ++mt) {
            if (mt + 1 < NMT) dma(mt + 1, (mt + 1) & 1);
    #pragma unroll
            for (int sub = 0; sub < KVM; ++sub) {
                const int t = mt * KVM + sub;
                if (t < NT) {
                    const char* sK = smem + (mt & 1) * (KVM * BUF_BYTES) + sub * BUF_BYTES;
                    const char* sV = sK + K_BYTES;
                    const bool tail = (t + 1) * KV_TILE > nk;
                    f32x16 s0[2], s1[2];
                    bf16x8 pf0[2][2], pf1[2][2];
                    qk(sK, 0, s0);
                    if (tail) mask_tail(t, s0);
                    if constexpr (!STATIC) update_max(0, s0);
                    if constexpr (QB == 2) {
                        qk(sK, 1, s1);
                        exp_pack(0, s0, pf0, use_delta);
                        if constexpr (PIN) pin(pf0);
                        if (tail) mask_tail(t, s1);
                        if constexpr (!STATIC) update_max(1, s1);
                        pv(sV, 0, pf0);
                        exp_pack(1, s1, pf1, use_delta);
                        pv(sV, 1, pf1);
                    } else {
                        exp_pack(0, s0, pf0, use_delta);
                        pv(sV, 0, pf0);
                    }
                }
            }
            wg_barrier_counted();  // everyone done with buffer mt&1; DMA of macro tile mt+1 landed
        }
    };
    if constexpr (EST && QB == 2 && !EST_NODELTA) {
        // per WAVE: the correction only if some lane's two rows differ by more than IGGT_EST_DELTA_MAX bits (the pre-pass dealt
        // the tile's 256 rows in shift order, so neighbours rarely do).  Waves of one workgroup may therefore run DIFFERENT
        // copies of the tile loop and meet at different barrier INSTRUCTIONS.  That is outside what HIP documents for
        // __syncthreads(), so the loops do not use it: their barrier is the ISA-level wg_barrier_counted()
        // (attention_common.h) -- s_barrier makes a wave wait until every wave of its workgroup has executed AN s_barrier; the
        // hardware counts waves, not program counters, and both copies execute the same number per macro tile.  The two-
        // workgroups-per-CU GEMM's ping-pong schedule (gemm_bf16_t256.hip) relies on the same property.  Measured alternatives
        // (round 6, profiles/r06_attn_est_barrier_ab.txt): the choice made workgroup-uniform through __syncthreads_or costs
        // +6 ... +8 % in the three adversarial regimes (any in-pair jump of a 256-row tile puts all four waves on the corrected
        // loop); ONE loop around two barrier-free bodies allocates 256 VGPRs + 84 spilled.
        if (__any(fabsf(est_delta) > IGGT_EST_DELTA_MAX)) tile_loop(std::true_type{});
        else tile_loop(std::false_type{});
    } else {
        tile_loop(std::false_type{});
    }

    bool weak = false;   // static bound only: some row's numerators sank towards the subnormal range
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qr = q_base + qb * 32 + frow;
        if constexpr (EST && QB == 2) {
            const EstView ev = est_view(p);
            qr = ev.slotrow[(long)bh * ev.NqS + qr];    // re-read: nothing extra stays alive across the tile loop
            qr = (unsigned)qr < (unsigned)p.Nq ? qr : p.Nq;   // slots past the end (and anything out of range): not stored
        }
        const float l = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = (PART && !(l > 0.f)) ? 0.f : 1.0f / l;
        if constexpr (STATIC && !PART) {
            bool bad = !(l >= p.static_min_l);
            if constexpr (EST) {   // an fp16 numerator above the range rounds to +inf: the accumulator is then inf or NaN
                float chk = 0.f;
#pragma unroll
                for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) chk = __builtin_fmaf(o[qb][dh][r], 0.f, chk);
                chk += __shfl_xor(chk, 32, 64);
                bad = bad || !(chk == 0.f);
            }
            if constexpr (EST) {   // row-granular hand-over
                const EstView ev = est_view(p);
                if (fhalf == 0 && qr < p.Nq) ev.rowflag[(long)bh * ev.NqP + qr] = bad ? 1 : 0;
            } else {
                weak = weak || (qr < p.Nq && bad);
            }
        }
        bool live = qr < p.Nq;
        if constexpr (!STATIC) {
            if (rlist != nullptr) {
                live = qr < nlist;
                qr = rlist[live ? qr : nlist - 1];
            }
        }
        if constexpr (LIST) live = qr < nlist;
        if (live) {
            bf16_t* dst = ob_ptr + (long)qr * p.o_rs + 4 * fhalf;
            if constexpr (LIST) {   // list-position slot of key range ks: [ks][bh][position][64], row sum beside it
                const EstView ev = est_view(p);
                const long li = ((long)ks * p.B * p.H + bh) * ev.NqL + qr;
                dst = ev.o2 + li * 64 + 4 * fhalf;
                if (fhalf == 0) ev.l2[li] = l;
            } else if constexpr (PART) {   // partial slot (slot0 + ks): dense [slot][B][Nq][H * 64] rows, l as [slot][B][H][Nq]
                const long slot = p.slot0 + ks - ((p.seg_len > 0 && p.skip_seg >= 0 && ks > p.skip_seg) ? 1 : 0);
                dst = p.o_part + ((slot * p.B + b) * p.Nq + qr) * (long)(p.H * 64) + h * 64 + 4 * fhalf;
                if (fhalf == 0) {
                    const long li = ((slot * p.B + b) * p.H + h) * (long)p.Nq + qr;
                    p.l_part[li] = l;
                    p.c_part[li] = -cinit[0];   // the shift this row was computed under
                }
            }
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w;
                    w[0] = pack_h2<FMT>(o[qb][dh][4 * g + 0] * inv, o[qb][dh][4 * g + 1] * inv);
                    w[1] = pack_h2<FMT>(o[qb][dh][4 * g + 2] * inv, o[qb][dh][4 * g + 3] * inv);
                    *reinterpret_cast<u32x2*>(dst + dh * 32 + 8 * g) = w;
                }
        }
    }
    if constexpr (STATIC && !PART) {
        if (weak) p.flags[work] = 1;   // benign race: every writer stores 1; the gated dynamic pass recomputes the tile
    }
}

// out[b][row][h*64 + d] = sum_s l_s O_s / sum_s l_s over the partial slots; rows whose total row sum is below the acceptance
// threshold flag their query tile for the gated online-max pass (same rule as the one-pass static kernel).
template <int FMT>
__global__ __launch_bounds__(256) void attn_combine_kernel(const AttnParams p, int nslots, int tile_rows) {
    const long row = blockIdx.x;                  // (b, query row)
    const int b = (int)(row / p.Nq), qr = (int)(row - (long)b * p.Nq);
    const int C = p.H * 64;
    if (guard_skips(p.guard, p.guard_prev)) {     // the partial launches returned at once: hand every tile to the online-max pass
        if (threadIdx.x < p.H) p.flags[((long)b * p.H + threadIdx.x) * p.qtiles + qr / tile_rows] = 1;
        return;
    }
    for (int c = threadIdx.x * 4; c < C; c += 256 * 4) {
        const int h = c >> 6;
        float L = 0.f, wsum = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
        // segments may have been computed under different shifts (different key bounds): weights l_s 2^(shift_s - shift_max)
        float cmax = -INFINITY;
        for (int s = 0; s < nslots; ++s) cmax = fmaxf(cmax, p.c_part[(((long)s * p.B + b) * p.H + h) * (long)p.Nq + qr]);
        for (int s = 0; s < nslots; ++s) {
            const long li = (((long)s * p.B + b) * p.H + h) * (long)p.Nq + qr;
            const float wgt = __builtin_amdgcn_exp2f(p.c_part[li] - cmax);
            const float l = p.l_part[li] * wgt;
            wsum += wgt;
            const u32x2 w = *reinterpret_cast<const u32x2*>(p.o_part + (((long)s * p.B + b) * p.Nq + qr) * (long)C + c);
            L += l;
            acc[0] += l * h2_lo<FMT>(w[0]); acc[1] += l * h2_hi<FMT>(w[0]);
            acc[2] += l * h2_lo<FMT>(w[1]); acc[3] += l * h2_hi<FMT>(w[1]);
        }
        const float inv = 1.0f / L;
        u32x2 o;
        o[0] = pack_h2<FMT>(acc[0] * inv, acc[1] * inv);
        o[1] = pack_h2<FMT>(acc[2] * inv, acc[3] * inv);
        *reinterpret_cast<u32x2*>(p.o + (long)b * p.o_bs + (long)qr * p.o_rs + c) = o;
        // acceptance (attention_common.h): the mass a segment can have lost to fp16 flushing is bounded under ITS OWN shift --
        // n_s 2^-24 there, n_s 2^-24 2^(shift_s - shift_max) in the common units of L -- so the threshold is the per-key one
        // times sum_s n_s w_s (segments of equal length: Nk / nslots each), not times Nk
        if ((c & 63) == 0 && !(L >= p.static_min_l * (wsum / (float)nslots)))
            p.flags[((long)b * p.H + h) * p.qtiles + qr / tile_rows] = 1;
    }
}

}  // namespace

// Launched from the dispatcher in attention.hip: q_rows = 256 | 128 query rows per workgroup, kvm = 64-key tiles per macro tile.
template <int FMT, bool STATIC, bool PART, bool EST = false>
static void launch_v3(const AttnParams& p_in, int q_rows, int kvm, hipStream_t stream) {
    AttnParams p = p_in;
    const int mult = PART ? p.ksplit : 1;
    if (q_rows == 256) {
        p.qtiles = (p.Nq + 255) / 256;
        const dim3 grid(p.B * p.H * p.qtiles * mult), block(256);
        // (the estimated-shift kernel exists with 128-key macro tiles only at 256 rows: its two copies of the tile loop leave the
        // 64-key variant 9-14 VGPRs short, spilled inside the loop)
        if (kvm == 2 || EST) hipLaunchKernelGGL((flash_attn_d64_v3_kernel<2, 2, FMT, STATIC, PIN_DEFAULT, PART, EST>), grid, block, 0, stream, p);
        else if constexpr (!EST) hipLaunchKernelGGL((flash_attn_d64_v3_kernel<2, 1, FMT, STATIC, PIN_DEFAULT, PART, EST>), grid, block, 0, stream, p);
    } else {
        p.qtiles = (p.Nq + 127) / 128;
        const dim3 grid(p.B * p.H * p.qtiles * mult), block(256);
        if (kvm == 2) hipLaunchKernelGGL((flash_attn_d64_v3_kernel<1, 2, FMT, STATIC, PIN_DEFAULT, PART, EST>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((flash_attn_d64_v3_kernel<1, 1, FMT, STATIC, PIN_DEFAULT, PART, EST>), grid, block, 0, stream, p);
    }
}

#ifdef IGGT_ATTN_EST_TU
// attention_v3_est.hip: the estimated-shift instantiations only, in their own translation unit so that they can be built
// with -mllvm -amdgpu-sched-strategy=max-ilp (iggt_official_amd/build_ext.py): under the default strategy the f16 256-row
// instantiation allocates 255 VGPRs + 7 spilled with a 16-byte reload inside the tile loop, under max-ilp 251 and none; the
// norm-bound and online-max kernels keep the flags their measurements were taken with.
int iggt_launch_flash_attn_v3_est(const AttnParams& p, int q_rows, int kvm, int fmt, hipStream_t stream) {
    if (fmt == FMT_F16) launch_v3<FMT_F16, true, false, true>(p, q_rows, kvm, stream);
    else launch_v3<FMT_BF16, true, false, true>(p, q_rows, kvm, stream);
    return 0;
}
#else
int iggt_launch_flash_attn_v3_est(const AttnParams& p, int q_rows, int kvm, int fmt, hipStream_t stream);

int iggt_launch_flash_attn_v3_list(const AttnParams& p_in, int fmt, hipStream_t stream) {
    AttnParams p = p_in;
    const EstView ev = est_view(p);
    p.ksplit = EST_KS2; p.slot0 = 0; p.seg_len = 0; p.skip_seg = -1; p.seg_kmax = nullptr;
    p.qtiles = ev.NqL / 128;
    const dim3 grid((unsigned)((long)p.B * p.H * p.qtiles * EST_KS2)), block(256);
    if (fmt == FMT_F16) hipLaunchKernelGGL((flash_attn_d64_v3_kernel<1, 2, FMT_F16, true, PIN_DEFAULT, true, false, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((flash_attn_d64_v3_kernel<1, 2, FMT_BF16, true, PIN_DEFAULT, true, false, true>), grid, block, 0, stream, p);
    return 0;
}

int iggt_launch_flash_attn_v3(const AttnParams& p, int q_rows, int kvm, int fmt, bool static_bound, hipStream_t stream) {
    if (static_bound && p.ksplit > 0) {
        if (fmt == FMT_F16) launch_v3<FMT_F16, true, true>(p, q_rows, kvm, stream);
        else launch_v3<FMT_BF16, true, true>(p, q_rows, kvm, stream);
    } else if (static_bound) {
        if (fmt == FMT_F16) launch_v3<FMT_F16, true, false>(p, q_rows, kvm, stream);
        else launch_v3<FMT_BF16, true, false>(p, q_rows, kvm, stream);
        // the estimated-shift instantiation: runs when the adaptive switch names mode 1
        if (p.est_ws != nullptr) iggt_launch_flash_attn_v3_est(p, q_rows, kvm, fmt, stream);
    } else {
        if (fmt == FMT_F16) launch_v3<FMT_F16, false, false>(p, q_rows, kvm, stream);
        else launch_v3<FMT_BF16, false, false>(p, q_rows, kvm, stream);
    }
    return 0;
}

int iggt_launch_attn_combine(const AttnParams& p_in, int nslots, int q_rows, int fmt, hipStream_t stream) {
    AttnParams p = p_in;
    p.qtiles = (p.Nq + q_rows - 1) / q_rows;
    const dim3 grid((unsigned)((long)p.B * p.Nq)), block(256);
    if (fmt == FMT_F16) hipLaunchKernelGGL(attn_combine_kernel<FMT_F16>, grid, block, 0, stream, p, nslots, q_rows);
    else hipLaunchKernelGGL(attn_combine_kernel<FMT_BF16>, grid, block, 0, stream, p, nslots, q_rows);
    return 0;
}
#endif  // IGGT_ATTN_EST_TU
