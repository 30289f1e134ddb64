// Flash attention d=64, variant 2: staggered query blocks + deferred-max rescale.
//
// Same data layout, LDS images, swapped-QK^T / in-register-P scheme as attention.hip (see there).
// PMC on the v1 kernel at 32 views (profiles/r01_attn_pmc.txt): matrix pipe busy 32-38 %, VALU issue
// busy ~60 %, and the two are almost serialised because a wave's MFMAs (QK^T, then PV) and its
// softmax VALU work sit in separate dependent phases.  v2 staggers the two 32-row query blocks of a
// wave so that every MFMA group has independent VALU work of the *other* block next to it in the
// same basic block:
//
//      [QK^T(q0)] [max(q0)] | [QK^T(q1)  ||  exp/sum/pack(q0)] [max(q1)] | [PV(q0) || exp/sum/pack(q1)] | [PV(q1)]
//
// and removes the per-tile O rescale: the running max m is only advanced (and O, l rescaled) when
// some row's tile maximum exceeds m by more than 2^THR (wave-uniform vote, rare after the first
// tiles); P = exp2(s*c - m) is then bounded by 2^THR, harmless in bf16/fp32 (guide T13).
// K and V fragments are re-read from LDS for the second block (LDS pipe is < 20 % busy).
#include "attention_common.h"
#include "../../include/iggt_hip.h"

using namespace iggt_attn;

namespace {

constexpr float DEFER_THR = 4.0f;  // log2 units: P <= 16

template <int QB, int ABL = 0>
__global__ __launch_bounds__(256, 2) void flash_attn_d64_v2_kernel(const AttnParams p) {
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
            if (!ok) {
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
    const int tr_i = lane & 15, tr_g = (lane >> 4) & 1;
    const float c = p.scale_log2;
    const int NT = (p.Nk + KV_TILE - 1) / KV_TILE;

    // ---- building blocks ---------------------------------------------------------------------
    auto qk = [&](const char* sK, int qb, f32x16 (&s)[2]) {
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kvh][r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 kf =
                    *reinterpret_cast<const bf16x8*>(sK + swz_off(kvh * 32 + frow, 2 * kc + fhalf));
                s[kvh] = mfma32(kf, qf[qb][kc], s[kvh]);
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
                if (kv >= p.Nk) s[kvh][r] = -INFINITY;
            }
    };
    // row max of the tile; advance m (rescaling O and l) only if some row needs it
    auto update_max = [&](int qb, const f32x16 (&s)[2]) {
        float mx = s[0][0];
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kvh][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
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
    auto exp_pack = [&](int qb, f32x16 (&s)[2], bf16x8 (&pf)[2][2]) {
        if constexpr (ABL == 2) {  // ablation: no softmax VALU at all (scores packed directly)
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh) {
                pf[kvh][0] = pack8(s[kvh], 0);
                pf[kvh][1] = pack8(s[kvh], 8);
            }
            l_run[qb] = 1.f;
            return;
        }
        const float m = m_run[qb];
        float ls0 = 0.f, ls1 = 0.f, ls2 = 0.f, ls3 = 0.f;
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = __builtin_fmaf(s[kvh][r], c, -m);
                s[kvh][r] = (ABL == 1) ? a : __builtin_amdgcn_exp2f(a);  // ABL 1: ablation without v_exp
            }
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                ls0 += s[kvh][r];
                ls1 += s[kvh][r + 1];
                ls2 += s[kvh][r + 2];
                ls3 += s[kvh][r + 3];
            }
            pf[kvh][0] = pack8(s[kvh], 0);
            pf[kvh][1] = pack8(s[kvh], 8);
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
                    o[qb][dh] = mfma32(__builtin_bit_cast(bf16x8, v8), pf[kvh][cc], o[qb][dh]);
                }
    };

    gload(0);
    swrite(0);
    __syncthreads();
    for (int t = 0; t < NT; ++t) {
        if (t + 1 < NT) gload(t + 1);
        const char* sK = smem + (t & 1) * BUF_BYTES;
        const char* sV = sK + K_BYTES;
        const bool tail = (t + 1) * KV_TILE > p.Nk;

        f32x16 s0[2], s1[2];
        bf16x8 pf0[2][2], pf1[2][2];
        qk(sK, 0, s0);
        if (tail) mask_tail(t, s0);
        if constexpr (ABL != 2) update_max(0, s0);
        if constexpr (QB == 2) {
            qk(sK, 1, s1);        // MFMA  ||  VALU of block 0 (same basic block)
            exp_pack(0, s0, pf0);
            if (tail) mask_tail(t, s1);
            if constexpr (ABL != 2) update_max(1, s1);
            pv(sV, 0, pf0);       // MFMA  ||  VALU of block 1
            exp_pack(1, s1, pf1);
            pv(sV, 1, pf1);
        } else {
            exp_pack(0, s0, pf0);
            pv(sV, 0, pf0);
        }
        if (t + 1 < NT) swrite((t + 1) & 1);
        __syncthreads();
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

// Launched from iggt_flash_attn_bf16_d64 (attention.hip); q_rows = 128 or 256.
int iggt_launch_flash_attn_v2(const AttnParams& p_in, int q_rows, hipStream_t stream) {
    AttnParams p = p_in;
    if (q_rows == 1256 || q_rows == 2256) {  // timing ablations only (wrong results by construction)
        p.qtiles = (p.Nq + 255) / 256;
        if (q_rows == 1256)
            hipLaunchKernelGGL((flash_attn_d64_v2_kernel<2, 1>), dim3(p.B * p.H * p.qtiles), dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL((flash_attn_d64_v2_kernel<2, 2>), dim3(p.B * p.H * p.qtiles), dim3(256), 0, stream, p);
    } else if (q_rows == 256) {
        p.qtiles = (p.Nq + 255) / 256;
        hipLaunchKernelGGL(flash_attn_d64_v2_kernel<2>, dim3(p.B * p.H * p.qtiles), dim3(256), 0, stream, p);
    } else {
        p.qtiles = (p.Nq + 127) / 128;
        hipLaunchKernelGGL(flash_attn_d64_v2_kernel<1>, dim3(p.B * p.H * p.qtiles), dim3(256), 0, stream, p);
    }
    return 0;
}
