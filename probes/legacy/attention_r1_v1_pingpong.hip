// NOT BUILT.  Round-1 baseline kernels (v1 register-staged kernel and the 8-wave ping-pong variant) that used to ship in
// libiggt_hip.so for A/B tests; the production kernel is iggt_official_amd/csrc/attention_v3.hip.  Kept for reference only.
// Flash-attention forward, head dim 64, bf16 operands on MFMA, fp32 online softmax.
//
//   O[b,h,i,:] = softmax_j(scale * Q[b,h,i,:].K[b,h,j,:]) V[b,h,j,:]      (no mask, no bias)
//
// Replaces F.scaled_dot_product_attention at reference iggt/layers/attention.py:60-66 for the
// DINOv2 blocks (B=S, N=P), the frame blocks (B=S, N=P) and the global blocks (B=1, N=S*P)
// (iggt/models/aggregator.py:277-336).
//
// Layout: token-major.  element (b,h,n,d) of q/k/v/o lives at ptr + b*bs + n*rs + h*64 + d, so the
// kernel reads q,k,v straight out of the [T, 3*C] qkv GEMM output and writes o as the [T, C] input
// of the proj GEMM -- no head transposes anywhere, and K/V of different views are contiguous in the
// token dimension (what the multi-GPU all-gather wants).
//
// CDNA4 mapping (wave64, v_mfma_f32_32x32x16_bf16):
//  * workgroup = 4 waves; each wave owns QB 32-row query blocks (QB=2 -> 256 query rows / WG);
//    K/V tiles of 64 keys are staged global->reg->LDS (issue one tile ahead, write after the MFMA
//    phase), double-buffered, one barrier per tile.
//  * "swapped" QK^T:  S^T[kv][q] = K_tile . Q^T  (A = K rows from LDS, B = Q rows held in VGPRs), so
//    every lane owns ONE query column: row max / row sum are in-lane reductions plus a single
//    exchange with lane^32; alpha rescales are lane-local.
//  * P never leaves registers: the S^T accumulator registers 8c..8c+7 of a lane are, by the C/D
//    layout, exactly 8 keys {16c + 8(j>>2) + 4(lane>>5) + (j&3)}; since a dot product is invariant
//    under a permutation of its K index, they are fed as the B fragment of O^T = V^T . P^T directly,
//    and the matching A fragment (V^T) is gathered with two ds_read_b64_tr_b16 transpose reads.
//  * K image in LDS: 128-B rows, 16-B slot XOR ((row>>1)&7)  -> conflict-free ds_read_b128.
//    V image in LDS: 128-B rows, 32-B chunk XOR (row&2)       -> conflict-free ds_read_b64_tr_b16.
//  * exp2 domain: scores are scaled by scale*log2(e) inside one v_fma, v_exp_f32 directly.
//  * grid is 1-D, XCD-chunked: all q-tiles of one (batch, head) are adjacent, so the workgroups that
//    are co-resident on an XCD stream the same head's K/V through that XCD's L2.
#include "attention_common.h"
#include "../../include/iggt_hip.h"

using namespace iggt_attn;

namespace {

template <int QB>
__global__ __launch_bounds__(256, 2) void flash_attn_d64_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fhalf = lane >> 5;

    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = work % p.qtiles;
    const int bh = work / p.qtiles;
    const int h = bh % p.H, b = bh / p.H;

    const bf16_t* qb_ptr = p.q + (long)b * p.q_bs + h * 64;
    const bf16_t* kb_ptr = p.k + (long)b * p.k_bs + h * 64;
    const bf16_t* vb_ptr = p.v + (long)b * p.v_bs + h * 64;
    bf16_t* ob_ptr = p.o + (long)b * p.o_bs + h * 64;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane -> q row frow, d = 16*kc + 8*fhalf .. +7 ---
    const int q_base = qt * (128 * QB) + wave * (32 * QB);
    bf16x8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qr = q_base + qb * 32 + frow;
        qr = qr < p.Nq ? qr : p.Nq - 1;
        const bf16_t* src = qb_ptr + (long)qr * p.q_rs + 8 * fhalf;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) qf[qb][kc] = *reinterpret_cast<const bf16x8*>(src + 16 * kc);
    }

    // ---- K/V staging map ---------------------------------------------------------------------
    const int ld_row = tid >> 3, ld_piece = tid & 7;
    u32x4 sk[2], sv[2];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kv = t * KV_TILE + ld_row + 32 * i;
            const bool ok = kv < p.Nk;
            const int kvc = ok ? kv : p.Nk - 1;
            u32x4 a = *reinterpret_cast<const u32x4*>(kb_ptr + (long)kvc * p.k_rs + ld_piece * 8);
            u32x4 c = *reinterpret_cast<const u32x4*>(vb_ptr + (long)kvc * p.v_rs + ld_piece * 8);
            if (!ok) {  // V rows past the end must be exact zeros (0 * garbage could be NaN)
                a = u32x4{0, 0, 0, 0};
                c = u32x4{0, 0, 0, 0};
            }
            sk[i] = a;
            sv[i] = c;
        }
    };
    auto swrite = [&](int buf) {
        char* sK = smem + buf * BUF_BYTES;
        char* sV = sK + K_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = ld_row + 32 * i;
            *reinterpret_cast<u32x4*>(sK + swz_off(r, ld_piece)) = sk[i];
            *reinterpret_cast<u32x4*>(sV + v_lds_off(r, ld_piece >> 1) + ((ld_piece & 1) << 4)) = sv[i];
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

    // transpose-read lane constants: 16-lane group g16 -> d chunk, lane i -> (row i>>2, 8-B piece i&3)
    const int tr_i = lane & 15, tr_g = (lane >> 4) & 1;
    const float c = p.scale_log2;
    const int NT = (p.Nk + KV_TILE - 1) / KV_TILE;

    gload(0);
    swrite(0);
    __syncthreads();
    for (int t = 0; t < NT; ++t) {
        if (t + 1 < NT) gload(t + 1);
        const char* sK = smem + (t & 1) * BUF_BYTES;
        const char* sV = sK + K_BYTES;

        // ---- S^T = K . Q^T ------------------------------------------------------------------
        f32x16 s[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][kvh][r] = 0.f;
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 kf =
                    *reinterpret_cast<const bf16x8*>(sK + swz_off(kvh * 32 + frow, 2 * kc + fhalf));
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[qb][kvh] = mfma32(kf, qf[qb][kc], s[qb][kvh]);
            }
        }
        // ---- tail mask (wave-uniform branch, last tile only) ----------------------------------
        if ((t + 1) * KV_TILE > p.Nk) {
            const int kv0 = t * KV_TILE + 4 * fhalf;
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + kvh * 32 + (r & 3) + 8 * (r >> 2);
                    if (kv >= p.Nk) {
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) s[qb][kvh][r] = -INFINITY;
                    }
                }
        }
        // ---- online softmax (per lane = per query column); P is packed to bf16 B-fragments at once
        //      so the fp32 score registers die before the PV phase ------------------------------
        bf16x8 pf[QB][2][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = s[qb][0][0];
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kvh][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qb], mx * c);
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
            m_run[qb] = m_new;
            float lsum = 0.f;
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kvh][r], c, -m_new));
                    s[qb][kvh][r] = pv;
                    lsum += pv;
                }
                pf[qb][kvh][0] = pack8(s[qb][kvh], 0);
                pf[qb][kvh][1] = pack8(s[qb][kvh], 8);
            }
            l_run[qb] = l_run[qb] * alpha + lsum;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][dh][r] *= alpha;
        }
        // ---- O^T += V^T . P^T ------------------------------------------------------------------
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    // keys of this lane half: kvb + {0..3} (elements 0-3) and kvb + 8 + {0..3} (4-7)
                    const int kvb = kvh * 32 + 16 * cc + 4 * fhalf;
                    const int row0 = kvb + (tr_i >> 2);
                    const int chunk = dh * 2 + tr_g;
                    typedef __attribute__((address_space(3))) short4v lds_s4;
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s4*)(sV + v_lds_off(row0, chunk) + 8 * (tr_i & 3)));
                    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s4*)(sV + v_lds_off(row0 + 8, chunk) + 8 * (tr_i & 3)));
                    typedef short short8v __attribute__((ext_vector_type(8)));
                    const short8v v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, v8);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) o[qb][dh] = mfma32(vf, pf[qb][kvh][cc], o[qb][dh]);
                }
            }
        }
        if (t + 1 < NT) swrite((t + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: O / l, bf16, 8-B stores (4 consecutive d per register quad) ---------------------
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qr = q_base + qb * 32 + frow;
        const float l = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l;
        if (qr < p.Nq) {
            bf16_t* dst = ob_ptr + (long)qr * p.o_rs + 4 * fhalf;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w;
                    w[0] = pack_bf16x2(o[qb][dh][4 * g + 0] * inv, o[qb][dh][4 * g + 1] * inv);
                    w[1] = pack_bf16x2(o[qb][dh][4 * g + 2] * inv, o[qb][dh][4 * g + 3] * inv);
                    *reinterpret_cast<u32x2*>(dst + dh * 32 + 8 * g) = w;
                }
        }
    }
}

// =================================================================================================
// Ping-pong variant for long sequences (global attention): 8 waves / workgroup, 512 query rows.
//
// Two wave groups (A = waves 0-3, B = waves 4-7; wave w and w+4 share a SIMD) run the same
// per-tile pipeline  QK^T(t) -> softmax(t) -> PV(t)  half a tile apart, in barrier-delimited segments:
//
//     segment 2u   : A: MFMA  [PV(u-1), QK^T(u)]      B: VALU  [softmax(u-1)]
//     segment 2u+1 : A: VALU  [softmax(u)]            B: MFMA  [PV(u-1), QK^T(u)]
//
// so on every SIMD one wave feeds the matrix pipe while its partner does the exp/max/sum VALU work
// (CDNA4 issues MFMA and VALU from different waves concurrently; two waves in the same phase just
// queue behind each other -- MI355X_MICROARCH "Two waves per SIMD").  LDS "stage" X = {K tile X,
// V tile X-1} is what both groups read during segments 2X and 2X+1; stages are double-buffered and
// written one stage ahead by the group that is in its VALU segment (registers are loaded with the
// next stage at the start of each MFMA segment).  K/V tiles are shared by all 8 waves, halving the
// L2->LDS traffic per query row relative to the 4-wave kernel.
template <int QB>
__global__ __launch_bounds__(512, 2) void flash_attn_d64_pp_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0 = A, 1 = B
    const int frow = lane & 31, fhalf = lane >> 5;

    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = work % p.qtiles;
    const int bh = work / p.qtiles;
    const int h = bh % p.H, b = bh / p.H;
    const bf16_t* qb_ptr = p.q + (long)b * p.q_bs + h * 64;
    const bf16_t* kb_ptr = p.k + (long)b * p.k_bs + h * 64;
    const bf16_t* vb_ptr = p.v + (long)b * p.v_bs + h * 64;
    bf16_t* ob_ptr = p.o + (long)b * p.o_bs + h * 64;

    const int q_base = qt * (256 * QB) + wave * (32 * QB);
    bf16x8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qr = q_base + qb * 32 + frow;
        qr = qr < p.Nq ? qr : p.Nq - 1;
        const bf16_t* src = qb_ptr + (long)qr * p.q_rs + 8 * fhalf;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) qf[qb][kc] = *reinterpret_cast<const bf16x8*>(src + 16 * kc);
    }

    const int NT = (p.Nk + KV_TILE - 1) / KV_TILE;
    // staging: 512 threads x 16 B = one 64x64 bf16 tile; thread -> (row = tid/8, piece = tid%8)
    const int ld_row = tid >> 3, ld_piece = tid & 7;
    u32x4 sk, sv;
    auto gload = [&](int X) {  // stage X = {K tile X, V tile X-1}
        sk = u32x4{0, 0, 0, 0};
        sv = u32x4{0, 0, 0, 0};
        const int kr = X * KV_TILE + ld_row;
        if (X < NT && kr < p.Nk) sk = *reinterpret_cast<const u32x4*>(kb_ptr + (long)kr * p.k_rs + ld_piece * 8);
        const int vr = (X - 1) * KV_TILE + ld_row;
        if (X >= 1 && vr < p.Nk) sv = *reinterpret_cast<const u32x4*>(vb_ptr + (long)vr * p.v_rs + ld_piece * 8);
    };
    auto swrite = [&](int X) {
        char* sK = smem + (X & 1) * BUF_BYTES;
        char* sV = sK + K_BYTES;
        *reinterpret_cast<u32x4*>(sK + swz_off(ld_row, ld_piece)) = sk;
        *reinterpret_cast<u32x4*>(sV + v_lds_off(ld_row, ld_piece >> 1) + ((ld_piece & 1) << 4)) = sv;
    };

    f32x16 o[QB][2];
    f32x16 s[QB][2];
    bf16x8 pf[QB][2][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -1e30f;
        l_run[qb] = 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][dh][r] = 0.f;
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[qb][kvh][r] = 0.f;
            pf[qb][kvh][0] = pf[qb][kvh][1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    const int tr_i = lane & 15, tr_g = (lane >> 4) & 1;
    const float c = p.scale_log2;

    // prologue: stage 0 resident; group B pre-loads stage 1 (it writes it in segment 0)
    gload(0);
    swrite(0);
    if (grp == 1) gload(1);
    __syncthreads();

    // ---- MFMA segment: PV(t-1) then QK^T(t) ----------------------------------------------------
    auto mfma_segment = [&](int t) {
        {
            const int X = grp ? t + 2 : t + 1;  // next stage this thread has to deliver
            if (X <= NT) gload(X);
        }
        const char* sK = smem + (t & 1) * BUF_BYTES;
        const char* sV = sK + K_BYTES;
        if (t >= 1) {
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh) {
                        const int kvb = kvh * 32 + 16 * cc + 4 * fhalf;
                        const int row0 = kvb + (tr_i >> 2);
                        const int chunk = dh * 2 + tr_g;
                        typedef __attribute__((address_space(3))) short4v lds_s4;
                        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds_s4*)(sV + v_lds_off(row0, chunk) + 8 * (tr_i & 3)));
                        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds_s4*)(sV + v_lds_off(row0 + 8, chunk) + 8 * (tr_i & 3)));
                        typedef short short8v __attribute__((ext_vector_type(8)));
                        const short8v v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        const bf16x8 vf = __builtin_bit_cast(bf16x8, v8);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) o[qb][dh] = mfma32(vf, pf[qb][kvh][cc], o[qb][dh]);
                    }
        }
        if (t < NT) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[qb][kvh][r] = 0.f;
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const bf16x8 kf =
                        *reinterpret_cast<const bf16x8*>(sK + swz_off(kvh * 32 + frow, 2 * kc + fhalf));
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) s[qb][kvh] = mfma32(kf, qf[qb][kc], s[qb][kvh]);
                }
        }
    };
    // ---- VALU segment: write stage X, then softmax(t) ---------------------------------------------
    auto valu_segment = [&](int X, int t) {
        if (X <= NT) swrite(X);
        if (t >= 0 && t < NT) {
            if ((t + 1) * KV_TILE > p.Nk) {
                const int kv0 = t * KV_TILE + 4 * fhalf;
#pragma unroll
                for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + kvh * 32 + (r & 3) + 8 * (r >> 2);
                        if (kv >= p.Nk) {
#pragma unroll
                            for (int qb = 0; qb < QB; ++qb) s[qb][kvh][r] = -INFINITY;
                        }
                    }
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float mx = s[qb][0][0];
#pragma unroll
                for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kvh][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[qb], mx * c);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
                m_run[qb] = m_new;
                float lsum = 0.f;
#pragma unroll
                for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kvh][r], c, -m_new));
                        s[qb][kvh][r] = pv;
                        lsum += pv;
                    }
                    pf[qb][kvh][0] = pack8(s[qb][kvh], 0);
                    pf[qb][kvh][1] = pack8(s[qb][kvh], 8);
                }
                l_run[qb] = l_run[qb] * alpha + lsum;
#pragma unroll
                for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][dh][r] *= alpha;
            }
        }
    };

    // Both groups execute exactly 2*(NT+1) barriers; s_barrier counts arrivals, not program counters.
    if (grp == 0) {
        for (int u = 0; u <= NT; ++u) {
            mfma_segment(u);          // segment 2u
            __syncthreads();
            valu_segment(u + 1, u);   // segment 2u+1
            __syncthreads();
        }
    } else {
        for (int u = 0; u <= NT; ++u) {
            valu_segment(u + 1, u - 1);  // segment 2u
            __syncthreads();
            mfma_segment(u);             // segment 2u+1
            __syncthreads();
        }
    }

#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qr = q_base + qb * 32 + frow;
        const float l = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l;
        if (qr < p.Nq) {
            bf16_t* dst = ob_ptr + (long)qr * p.o_rs + 4 * fhalf;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w;
                    w[0] = pack_bf16x2(o[qb][dh][4 * g + 0] * inv, o[qb][dh][4 * g + 1] * inv);
                    w[1] = pack_bf16x2(o[qb][dh][4 * g + 2] * inv, o[qb][dh][4 * g + 3] * inv);
                    *reinterpret_cast<u32x2*>(dst + dh * 32 + 8 * g) = w;
                }
        }
    }
}

}  // namespace

static int flash_attn_h16(int fmt, const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                          long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs, long o_rs,
                          float scale, int q_rows_per_wg, void* stream) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return -1;
    if ((q_rs % 8) || (k_rs % 8) || (v_rs % 8) || (o_rs % 4)) return -2;
    if ((q_bs % 8) || (k_bs % 8) || (v_bs % 8) || (o_bs % 4)) return -2;
    AttnParams p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs;
    p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.scale_log2 = scale * 1.4426950408889634f;
    if (q_rows_per_wg == 0) {
        // Production kernel = v3 (attention_v3.hip).  Tile choice fitted to measurements (probes/attn_tiles.py,
        // profiles/r01_microbench_kernels.txt): the 256-row tile (2 resident workgroups per CU) is ~10 % faster per
        // query row, unless (a) it cannot put >= 1.2 rounds of workgroups on the chip -- the per-rank global attention
        // of an 8-GPU run has 352 -- or (b) it pads the sequence > 5 % more than the 128-row tile (1374-token frames).
        static int cus = 0;
        if (cus == 0) {
            hipDeviceProp_t prop;
            int dev = 0;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                      ? prop.multiProcessorCount : 256;
        }
        const long w256 = (long)B * H * ((Nq + 255) / 256);
        const int pad256 = ((Nq + 255) / 256) * 256, pad128 = ((Nq + 127) / 128) * 128;
        const bool small_grid = w256 * 10 < (long)cus * 2 * 12;
        const bool pads_more = (long)pad256 * 100 > (long)pad128 * 105;
        q_rows_per_wg = (small_grid || pads_more) ? 5128 : 6256;
    }
    const bool v3 = q_rows_per_wg == 5128 || q_rows_per_wg == 5256 || q_rows_per_wg == 6128 || q_rows_per_wg == 6256;
    if (fmt != FMT_BF16 && !v3) return -4;  // the earlier kernel generations (128 / 256 / 512) are bf16-only
    if (q_rows_per_wg == 256) {
        p.qtiles = (Nq + 255) / 256;
        hipLaunchKernelGGL(flash_attn_d64_kernel<2>, dim3(B * H * p.qtiles), dim3(256), 0, (hipStream_t)stream, p);
    } else if (q_rows_per_wg == 128) {
        p.qtiles = (Nq + 127) / 128;
        hipLaunchKernelGGL(flash_attn_d64_kernel<1>, dim3(B * H * p.qtiles), dim3(256), 0, (hipStream_t)stream, p);
    } else if (v3) {
        iggt_launch_flash_attn_v3(p, q_rows_per_wg % 1000, q_rows_per_wg / 1000 - 4, fmt, (hipStream_t)stream);
    } else if (q_rows_per_wg == 512) {
        p.qtiles = (Nq + 511) / 512;
        hipLaunchKernelGGL(flash_attn_d64_pp_kernel<2>, dim3(B * H * p.qtiles), dim3(512), 0, (hipStream_t)stream, p);
    } else {
        return -3;
    }
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_flash_attn_bf16_d64(const void* q, const void* k, const void* v, void* o, int B, int H,
                                        int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                                        long v_bs, long v_rs, long o_bs, long o_rs, float scale,
                                        int q_rows_per_wg, void* stream) {
    return flash_attn_h16(FMT_BF16, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale,
                          q_rows_per_wg, stream);
}

extern "C" int iggt_flash_attn_f16_d64(const void* q, const void* k, const void* v, void* o, int B, int H,
                                       int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                                       long v_bs, long v_rs, long o_bs, long o_rs, float scale,
                                       int q_rows_per_wg, void* stream) {
    return flash_attn_h16(FMT_F16, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale,
                          q_rows_per_wg, stream);
}
