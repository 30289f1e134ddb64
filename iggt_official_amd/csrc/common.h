// Common device helpers for the IGGT gfx950 (CDNA4) kernels.
// wave = 64 lanes; MFMA = v_mfma_f32_32x32x16_bf16 (A 32x16, B 16x32, C/D 32x32 fp32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define IGGT_DEVINL __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// XCD-aware work-item remap.  Hardware places block b on XCD (b % 8) (speed only, never relied on
// for correctness).  We give every XCD one contiguous chunk of the logical work list so that the
// workgroups sharing an operand panel (same head's K/V, same A row-panel) hit the same 4 MiB L2.
// Bijective for any nwg (guide section 5 "XCD swizzle must be bijective").
IGGT_DEVINL int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// fp32 -> bf16 round-to-nearest-even (compiler emits v_cvt_pk_bf16_f32 on gfx950)
IGGT_DEVINL uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 v;
    v[0] = (bf16_t)lo;
    v[1] = (bf16_t)hi;
    return __builtin_bit_cast(uint32_t, v);
}
IGGT_DEVINL float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
IGGT_DEVINL float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

IGGT_DEVINL f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// 16-bit operand format of the trunk kernels (template parameter FMT): the same kernels, LDS images and MFMA rate
// serve both; only the conversion and the MFMA opcode differ.  Fragments travel as raw 16-bit containers (bf16x8).
//   FMT_BF16 (0): 8 significant bits -- the reference's autocast(bf16) GPU mode (demo.py:190-193)
//   FMT_F16  (1): 11 significant bits -- 8x smaller operand rounding; what the 1e-3 parity target against the fp32
//                 CPU reference needs (oracle/precision_sim.py: token error 6.5e-3 with bf16, 8.4e-4 with fp16).
//                 Stores saturate at +-65504 instead of producing inf.
constexpr int FMT_BF16 = 0, FMT_F16 = 1;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int FMT>
IGGT_DEVINL f32x16 mfma32h(bf16x8 a, bf16x8 b, f32x16 c) {
    if (FMT == FMT_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0,
                                                      0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// fp32 pair -> packed 16-bit pair, round-to-nearest-even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).  SAT clamps fp16 to
// the finite range first (activations of a real checkpoint are far below it; the clamp only keeps an outlier finite).
template <int FMT, bool SAT = true>
IGGT_DEVINL uint32_t pack_h2(float lo, float hi) {
    if (FMT == FMT_F16) {
        if (SAT) {
            lo = __builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f);
            hi = __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f);
        }
        f16x2 v;
        v[0] = (_Float16)lo;
        v[1] = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, v);
    }
    return pack_bf16x2(lo, hi);
}
template <int FMT>
IGGT_DEVINL uint16_t pack_h1(float x) {
    return (uint16_t)(pack_h2<FMT>(x, 0.f) & 0xffffu);
}
// packed 16-bit pair -> fp32
template <int FMT>
IGGT_DEVINL float h2_lo(uint32_t w) {
    if (FMT == FMT_F16) return (float)__builtin_bit_cast(f16x2, w)[0];
    return bf16_lo(w);
}
template <int FMT>
IGGT_DEVINL float h2_hi(uint32_t w) {
    if (FMT == FMT_F16) return (float)__builtin_bit_cast(f16x2, w)[1];
    return bf16_hi(w);
}

// C/D fragment of the 32x32 MFMA: lane l, register r holds element
//   (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31).
IGGT_DEVINL int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// LDS image of a [rows][64] bf16 tile (128-byte rows = 8 slots of 16 B): the 16-B slot index is
// XOR-ed with ((row >> 1) & 7).  An MFMA operand read (lane -> row = lane&31, fixed slot) is then
// conflict-free for ds_read_b128's 16-lane groups (bank row = 256 B = two tile rows).
IGGT_DEVINL int swz_off(int row, int slot) { return row * 128 + (((slot ^ (row >> 1)) & 7) << 4); }

IGGT_DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define IGGT_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t _e = hipGetLastError();                    \
        if (_e != hipSuccess) return (int)_e;                 \
    } while (0)
