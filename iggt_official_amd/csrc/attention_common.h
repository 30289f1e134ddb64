// Shared definitions of the head-dim-64 flash attention kernels (attention.hip, attention_v2.hip).
#pragma once
#include "common.h"

namespace iggt_attn {

struct AttnParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    bf16_t* o;
    int B, H, Nq, Nk;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;  // in elements
    float scale_log2;  // softmax scale * log2(e)
    int qtiles;
};

constexpr int KV_TILE = 64;
constexpr int K_BYTES = KV_TILE * 128;  // 8 KiB
constexpr int BUF_BYTES = 2 * K_BYTES;  // K + V

IGGT_DEVINL int v_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 2)) << 5); }

IGGT_DEVINL bf16x8 pack8(const f32x16& s, int base) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)s[base + j];
    return r;
}


}  // namespace iggt_attn

// experimental variants live in their own translation units
int iggt_launch_flash_attn_v3(const iggt_attn::AttnParams& p, int q_rows, int kvm, hipStream_t stream);
