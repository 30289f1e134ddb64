// NOT BUILT (kept for the record, round 2).  Measured on MI355X against the production order of attention_v3.hip, same launch,
// interleaved processes (probes/attn_pipe_ab.py at the time), 32 views, N = 43 968, 16 heads:
//     v3 order 7.28 ms (fp16) / 6.81 ms (bf16)   |   this file, tile-pipelined 7.68 / 7.18   |   this file, FUSED variant (both query
//     blocks share every K / V fragment: half the LDS reads, no intra-wave overlap at all) 7.70 / 7.35
// i.e. three different instruction orders -- vector work next to two of four MFMA groups, next to all four, next to none -- and
// LDS traffic halved or not all land within 5 %, while the SAME instruction stream with bf16 instead of fp16 operands is 7 %
// faster and the effective clock under this kernel is 1.77 GHz (profiles/r02_attn_v3_pmc.txt): the kernel sits at the power
// limit of the chip for fp16 MFMA on real data, not at an issue / LDS / VALU limit.  (All three variants produce checksums
// identical to 7 digits.)
// Flash attention d = 64, static-bound softmax, SOFTWARE-PIPELINED ACROSS KEY TILES ("v5"): the global-attention kernel
// for key counts that are a multiple of 64 (32 views: 43 968 = 687 x 64; every per-rank launch of a sharded run).
//
// Same data layout, LDS images, swapped-QK^T / in-register-P scheme, LDS-DMA macro tiles and static softmax bound as
// attention_v3.hip.  What changes is the order of the four MFMA groups of a 64-key tile inside a wave.  v3 runs, per tile,
//        [QK^T(q0)]   [QK^T(q1) || exp(q0)]   [PV(q0) || exp(q1)]   [PV(q1)]
// so two of the four groups (2 x 256 matrix-pipe cycles) have no vector work beside them and the other two carry all of it
// (2 x ~460 VALU cycles beside 256 of MFMA): a wave alone needs ~1 440 cycles per tile for 1 024 cycles of MFMA, and two waves
// per SIMD do not interleave well enough to hide it (PMC: matrix pipe 60 % busy, waves issue-stalled 41 % of their cycles;
// removing a third of the VALU instructions changed nothing -- attention_v3.hip header).  Here the score block of q0 for
// tile t+1 is computed BEFORE the second PV group of tile t, and every exp/pack unit is cut in two 32-key halves:
//        A: QK^T(q1, t)    || exp(q0, t)   keys 32-63
//        B: PV(q0, t)      || exp(q1, t)   keys  0-31
//        C: QK^T(q0, t+1)  || exp(q1, t)   keys 32-63
//        D: PV(q1, t)      || exp(q0, t+1) keys  0-31
// Every group now has 16 v_exp + 8 v_cvt_pk + 19 v_add (~230 VALU cycles) beside its 8 MFMAs (256 cycles), and every
// consumer is at least one full group behind its producer (no MFMA -> VALU or VALU -> MFMA result stall).  The register
// budget is the one of v3 (at most two score blocks and two half-packed P fragments live); the groups are kept apart with
// sched_barrier so that the compiler schedules each one on its own (one region over the whole tile spills 200+ VGPRs).
// Cost: group C of the last tile of a macro tile reads K of the NEXT macro tile, so the "next buffer landed" barrier moves
// in front of it and the "this buffer is free" barrier + the DMA of the macro tile after next follow group D: two barriers
// per 128 keys instead of one.
// No key masking anywhere: Nk % 64 == 0 is a launch condition (other shapes run attention_v3.hip).
#include <stdlib.h>
#include <type_traits>

#include "attention_common.h"
#include "../../include/iggt_hip.h"

using namespace iggt_attn;

namespace {

constexpr int KVM = 2;   // 64-key tiles per macro tile

template <int FMT, bool PART, bool FUSED = false>
__global__ __launch_bounds__(256, 2) void flash_attn_d64_v5_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * KVM * BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;

    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = work % p.qtiles;
    int bh = work / p.qtiles, ks = 0;
    if constexpr (PART) {
        ks = bh % p.ksplit;
        bh /= p.ksplit;
    }
    const int h = bh % p.H, b = bh / p.H;
    const bf16_t* qb_ptr = p.q + (long)b * p.q_bs + h * 64;
    const bf16_t* kb_ptr = p.k + (long)b * p.k_bs + h * 64;
    const bf16_t* vb_ptr = p.v + (long)b * p.v_bs + h * 64;
    bf16_t* ob_ptr = p.o + (long)b * p.o_bs + h * 64;

    const int q_base = qt * 256 + wave * 64;
    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q_base + qb * 32 + frow;
        qr = qr < p.Nq ? qr : p.Nq - 1;
        const bf16_t* src = qb_ptr + (long)qr * p.q_rs + 8 * fhalf;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) qf[qb][kc] = *reinterpret_cast<const bf16x8*>(src + 16 * kc);
    }

    // LDS-DMA staging exactly as attention_v3.hip (8 chunks of 1 KiB per 64-key K / V tile, swizzles on the source address)
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int c_row = lane >> 3, c_pos = lane & 7;
    const bf16_t* ksrc[2];
    const bf16_t* vsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 8 + c_row;
        ksrc[i] = kb_ptr + (long)r * p.k_rs + (c_pos ^ ((r >> 1) & 7)) * 8;
        vsrc[i] = vb_ptr + (long)r * p.v_rs + ((((c_pos >> 1) ^ (r & 2)) << 1) | (c_pos & 1)) * 8;
    }
    const int NT = p.Nk / KV_TILE;
    auto dma = [&](int mt, int buf) {
        char* base = smem + buf * (KVM * BUF_BYTES);
#pragma unroll
        for (int sub = 0; sub < KVM; ++sub) {
            if (mt * KVM + sub < NT) {
                char* sK = base + sub * BUF_BYTES;
                char* sV = sK + K_BYTES;
                const long kv0 = (long)(mt * KVM + sub) * KV_TILE;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int chunk = (2 * wave + i) * 1024;
                    __builtin_amdgcn_global_load_lds((gptr_t*)(ksrc[i] + kv0 * p.k_rs), (lptr_t*)(sK + chunk), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gptr_t*)(vsrc[i] + kv0 * p.v_rs), (lptr_t*)(sV + chunk), 16, 0, 0);
                }
            }
        }
    };

    f32x16 o[2][2];
    float l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][dh][r] = 0.f;
    const int tr_i = lane & 15, tr_g = (lane >> 4) & 1;
    f32x16 cinit;   // -(c_h - SHIFT): enters through the accumulator input of the first QK^T MFMA of a score block
    {
        const float shift = p.qkmax[h] * p.qkmax[16 + h] * 1.00002f + 1e-3f - (FMT == FMT_F16 ? STATIC_SHIFT_F16 : 0.f);
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[r] = -shift;
    }
    int koff[4], voff[2];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) koff[kc] = frow * 128 + ((((2 * kc + fhalf) ^ (frow >> 1)) & 7) << 4);
    {
        const int vr = 4 * fhalf + (tr_i >> 2);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) voff[dh] = vr * 128 + ((((dh * 2 + tr_g) ^ (vr & 2))) << 5) + 8 * (tr_i & 3);
    }

    // ---- building blocks (each one group of 8 MFMAs or half an exp/pack unit) ----------------------------------------------
    auto qk = [&](const char* sK, int qb, f32x16 (&s)[2]) {
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kvh * 4096 + koff[kc]);
                s[kvh] = mfma32h<FMT>(kf, qf[qb][kc], kc == 0 ? cinit : s[kvh]);
            }
    };
    auto exp_half = [&](int qb, f32x16& s, bf16x8 (&pf)[2]) {   // 32 keys x 32 queries: 16 v_exp, 8 v_cvt_pk, 19 v_add
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
        float ls0 = 0.f, ls1 = 0.f, ls2 = 0.f, ls3 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            ls0 += s[r];
            ls1 += s[r + 1];
            ls2 += s[r + 2];
            ls3 += s[r + 3];
        }
        pf[0] = pack8h<FMT>(s, 0);
        pf[1] = pack8h<FMT>(s, 8);
        l_run[qb] += (ls0 + ls1) + (ls2 + ls3);
        asm volatile("" : "+v"(l_run[qb]));   // the row sum is taken HERE (else all adds of a macro tile sink to the loop latch
                                              // and 128 exponentials stay live: 100+ spilled VGPRs)
    };
    auto pv = [&](const char* sV, int qb, const bf16x8 (&pf)[2][2]) {
#pragma unroll
        for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    typedef __attribute__((address_space(3))) short4v lds_s4;
                    const char* base = sV + (kvh * 32 + 16 * cc) * 128 + voff[dh];
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(base));
                    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(base + 8 * 128));
                    typedef short short8v __attribute__((ext_vector_type(8)));
                    const short8v v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[qb][dh] = mfma32h<FMT>(__builtin_bit_cast(bf16x8, v8), pf[kvh][cc], o[qb][dh]);
                }
    };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };   // keeps the four groups of a tile in separate regions

    // tile range of this workgroup: whole macro tiles [mt0, NMT), tiles [mt0 * KVM, t_end)
    const int NMT_all = (NT + KVM - 1) / KVM;
    int mt0 = 0, NMT = NMT_all;
    if constexpr (PART) {
        mt0 = (int)((long)ks * NMT_all / p.ksplit);
        NMT = (int)((long)(ks + 1) * NMT_all / p.ksplit);
    }
    const int t_end = NMT * KVM < NT ? NMT * KVM : NT;

    // the packed numerators are "used" right behind their exp/pack: without this the compiler sinks whole exp/pack units
    // to their consumers (IR-level code motion ignores sched_barrier) and the groups lose their vector work again
    auto pin = [&](bf16x8 (&pf)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 w = __builtin_bit_cast(u32x4, pf[i]);
            asm volatile("" : "+v"(w));
            pf[i] = __builtin_bit_cast(bf16x8, w);
        }
    };

    f32x16 s0[2], s1[2];
    bf16x8 pf0[2][2], pf1[2][2];
    // One tile = groups A-D.  STEADY (compile time): the tile has a successor -- no branch inside the groups.
    auto tile = [&](int mt, auto sub_c, auto steady_c) {
        constexpr int sub = decltype(sub_c)::value;
        constexpr bool STEADY = decltype(steady_c)::value;
        const int t = mt * KVM + sub;
        const bool has_next = STEADY ? true : (t + 1 < t_end);
        const char* sK = smem + (mt & 1) * (KVM * BUF_BYTES) + sub * BUF_BYTES;
        const char* sV = sK + K_BYTES;
        // A
        qk(sK, 1, s1);
        exp_half(0, s0[1], pf0[1]);
        pin(pf0[1]);
        fence();
        // B
        pv(sV, 0, pf0);
        exp_half(1, s1[0], pf1[0]);
        pin(pf1[0]);
        fence();
        if (sub == KVM - 1 && has_next) {
            // the next macro tile (requested one macro tile ago) has landed and is visible to every wave
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            fence();
        }
        // C
        if (has_next) {
            const char* sKn = (sub == KVM - 1) ? smem + ((mt + 1) & 1) * (KVM * BUF_BYTES) : sK + BUF_BYTES;
            qk(sKn, 0, s0);
        }
        exp_half(1, s1[1], pf1[1]);
        pin(pf1[1]);
        fence();
        // D
        pv(sV, 1, pf1);
        if (has_next) {
            exp_half(0, s0[0], pf0[0]);
            pin(pf0[0]);
        }
        fence();
    };
    auto release = [&](int mt) {   // every wave is done with buffer mt & 1: it takes the macro tile after next
        if (mt + 2 < NMT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            fence();
            dma(mt + 2, mt & 1);
        }
    };
    typedef std::integral_constant<int, 0> sub0_t;
    typedef std::integral_constant<int, 1> sub1_t;

    if constexpr (FUSED) {
        // experiment: both query blocks share every K / V fragment (half the LDS reads per MFMA), no intra-wave stagger
        dma(mt0, mt0 & 1);
        __syncthreads();
        for (int mt = mt0; mt < NMT; ++mt) {
            if (mt + 1 < NMT) dma(mt + 1, (mt + 1) & 1);
#pragma unroll
            for (int sub = 0; sub < KVM; ++sub) {
                if (mt * KVM + sub < t_end) {
                    const char* sK = smem + (mt & 1) * (KVM * BUF_BYTES) + sub * BUF_BYTES;
                    const char* sV = sK + K_BYTES;
#pragma unroll
                    for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc) {
                            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kvh * 4096 + koff[kc]);
                            s0[kvh] = mfma32h<FMT>(kf, qf[0][kc], kc == 0 ? cinit : s0[kvh]);
                            s1[kvh] = mfma32h<FMT>(kf, qf[1][kc], kc == 0 ? cinit : s1[kvh]);
                        }
                    fence();
                    exp_half(0, s0[0], pf0[0]);
                    exp_half(0, s0[1], pf0[1]);
                    exp_half(1, s1[0], pf1[0]);
                    exp_half(1, s1[1], pf1[1]);
                    pin(pf0[0]); pin(pf0[1]); pin(pf1[0]); pin(pf1[1]);
                    fence();
#pragma unroll
                    for (int kvh = 0; kvh < 2; ++kvh)
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                            for (int dh = 0; dh < 2; ++dh) {
                                typedef __attribute__((address_space(3))) short4v lds_s4;
                                const char* base = sV + (kvh * 32 + 16 * cc) * 128 + voff[dh];
                                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(base));
                                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(base + 8 * 128));
                                typedef short short8v __attribute__((ext_vector_type(8)));
                                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                                o[0][dh] = mfma32h<FMT>(vf, pf0[kvh][cc], o[0][dh]);
                                o[1][dh] = mfma32h<FMT>(vf, pf1[kvh][cc], o[1][dh]);
                            }
                    fence();
                }
            }
            __syncthreads();
        }
    } else {
    // prologue: first macro tile resident, second one in flight; score block q0 of the first tile and its first exp half
    dma(mt0, mt0 & 1);
    __syncthreads();
    if (mt0 + 1 < NMT) dma(mt0 + 1, (mt0 + 1) & 1);
    qk(smem + (mt0 & 1) * (KVM * BUF_BYTES), 0, s0);
    fence();
    exp_half(0, s0[0], pf0[0]);
    pin(pf0[0]);
    fence();

    // steady state: macro tiles whose tiles all have a successor; then the last one or two tiles
    const int NMT_steady = mt0 + (t_end - 1 - mt0 * KVM) / KVM;
    int mt = mt0;
    for (; mt < NMT_steady; ++mt) {
        tile(mt, sub0_t{}, std::true_type{});
        tile(mt, sub1_t{}, std::true_type{});
        release(mt);
    }
    for (; mt < NMT; ++mt) {
        tile(mt, sub0_t{}, std::false_type{});
        if (mt * KVM + 1 < t_end) tile(mt, sub1_t{}, std::false_type{});
        release(mt);
    }
    }

    bool weak = false;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qr = q_base + qb * 32 + frow;
        const float l = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = (PART && !(l > 0.f)) ? 0.f : 1.0f / l;
        if constexpr (!PART) weak = weak || (qr < p.Nq && !(l >= p.static_min_l));
        if (qr < p.Nq) {
            bf16_t* dst = ob_ptr + (long)qr * p.o_rs + 4 * fhalf;
            if constexpr (PART) {
                const long slot = p.slot0 + ks;
                dst = p.o_part + ((slot * p.B + b) * p.Nq + qr) * (long)(p.H * 64) + h * 64 + 4 * fhalf;
                if (fhalf == 0) p.l_part[((slot * p.B + b) * p.H + h) * (long)p.Nq + qr] = l;
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
    if constexpr (!PART) {
        if (weak) p.flags[work] = 1;
    }
}

template <int FMT, bool PART>
void launch_v5(const AttnParams& p_in, hipStream_t stream) {
    AttnParams p = p_in;
    p.qtiles = (p.Nq + 255) / 256;
    const dim3 grid(p.B * p.H * p.qtiles * (PART ? p.ksplit : 1)), block(256);
    static int fused = -1;
    if (fused < 0) {
        const char* e = getenv("IGGT_ATTN_FUSED");
        fused = (e && e[0] == '1') ? 1 : 0;
    }
    if (fused) hipLaunchKernelGGL((flash_attn_d64_v5_kernel<FMT, PART, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((flash_attn_d64_v5_kernel<FMT, PART>), grid, block, 0, stream, p);
}

}  // namespace

bool iggt_flash_attn_v5_applies(const AttnParams& p, int q_rows, int kvm) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("IGGT_ATTN_PIPELINE");
        enabled = (e && e[0] == '0') ? 0 : 1;
    }
    return enabled && q_rows == 256 && kvm == KVM && (p.Nk % KV_TILE) == 0 && p.Nk >= 2 * KVM * KV_TILE;
}

int iggt_launch_flash_attn_v5(const AttnParams& p, int fmt, hipStream_t stream) {
    if (p.ksplit > 0) {
        if (fmt == FMT_F16) launch_v5<FMT_F16, true>(p, stream);
        else launch_v5<FMT_BF16, true>(p, stream);
    } else {
        if (fmt == FMT_F16) launch_v5<FMT_F16, false>(p, stream);
        else launch_v5<FMT_BF16, false>(p, stream);
    }
    return 0;
}
