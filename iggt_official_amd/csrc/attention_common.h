// Shared definitions of the head-dim-64 flash attention kernels (attention.hip, attention_v2.hip).
#pragma once
#include "common.h"

namespace iggt_attn {

// Workgroup barrier at the ISA level: exactly what __syncthreads() lowers to (release fence -- the compiler drains vmcnt / lgkmcnt
// --, s_barrier, acquire fence), spelled out because the estimated-shift kernel (attention_v3.hip) lets the waves of one workgroup
// choose between two copies of its tile loop: s_barrier releases a wave when EVERY wave of the workgroup has executed an s_barrier
// -- the hardware counts arrivals, it does not compare program counters -- so waves in different copies synchronise correctly as
// long as both copies execute the same number of barriers, which they do (one per macro tile + one in front of the loop).
__device__ __forceinline__ void wg_barrier_counted() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct AttnParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    bf16_t* o;
    int B, H, Nq, Nk;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;  // in elements
    float scale_log2;  // softmax scale * log2(e)
    int qtiles;
    // static-bound softmax (attention_v3.hip): qkmax[h] = max_i |q^_i|, qkmax[H + h] = max_j |k^_j| of head h (q^ carries
    // scale * log2 e), flags[work item] = 1 for query tiles the dynamic kernel has to redo, static_min_l = smallest
    // acceptable row sum.  flags alone (qkmax == nullptr) gates the dynamic kernel on the flagged tiles.
    const float* qkmax;
    int* flags;
    float static_min_l;
    // key-split partial results (static bound only; ksplit == 0: one pass): workgroup (.., key range ks, ..) writes slot
    // slot0 + ks of o_part [slots][B][Nq][H*64] (16-bit, normalised by its own row sum) and l_part [slots][B][H][Nq] (fp32)
    int ksplit, slot0;
    // segment mode (seg_len > 0): the `ksplit` ranges are the key SEGMENTS [s * seg_len, min((s + 1) * seg_len, Nk)) -- one
    // rank's rows of the gathered K/V buffer each -- instead of equal shares of the macro tiles; segment skip_seg (the
    // rank's own keys, already done from its local buffer while the gather was in flight; -1: none) is left out and the
    // segments after it move up one slot, so that every rank launches the same 7-of-8-ranges grid whatever its position
    int seg_len, skip_seg;
    // segment mode only: per-segment key bounds, float [ksplit][32] -- the 32 norm maxima every rank's q/k-norm kernel leaves
    // (entries 16..31 = keys), all-gathered next to the K/V rows; segment s is then shifted by its OWN rank's maximum (tighter
    // than one maximum over all ranks, and no pass over the gathered keys).  nullptr: qkmax[16 + h] for every segment.
    const float* seg_kmax;
    bf16_t* o_part;
    float* l_part;
    // per-row shift each partial was computed under, [slots][B][H][Nq] fp32 like l_part: segments launched with different key
    // bounds (a rank's own keys before the gather has landed, the gathered keys after) still combine exactly
    float* c_part;
    // adaptive switch (static bound only).  guard[0] > 0: the static kernel flags every tile and returns at once -- this call
    // runs the online-max kernel only; guard[0] == 0: static bound; guard[0] < 0: never measured -> inherit the verdict of
    // guard_prev (the same kind of block one layer earlier).  The gated online-max pass updates the word from the number of
    // flagged tiles (attention_v3.hip guard_update): more than 1/8 flagged -> the next guard_retry calls skip the static
    // kernel.  guard[1] / [2] / [3] = flagged tiles / tiles / calls of the last launch (reports).  nullptr: always static.
    //   guard[4] = mode of the static kernel (round 4): 0 = Cauchy-Schwarz bound from the norms; 1 = ESTIMATED shift (below);
    //   guard[5] = query rows the last launch handed to the online-max pass one by one (-1: static kernel skipped).
    int* guard;
    const int* guard_prev;
    int guard_retry;
    // Round 4 -- estimated-shift static softmax with a row-granular hand-over (attention_est.hip; one-pass launches only).
    // est_ws: ONE caller-owned scratch buffer; the kernels derive its sub-arrays from B, H, Nq, Nk (est_view below -- more
    // pointers in this argument block cost the static kernel scalar registers it does not have: the spill that followed put
    // a 16-byte reload into its tile loop).  nullptr: round-3 behaviour (norm bound, whole 256-row tiles flagged).
    unsigned char* est_ws;
    int est_force;    // guard == nullptr: 1 = estimated shift, 0 = norm bound
    // second-chance / list-mode launches of the static kernel (attention_est.hip): rows come from rowlist, shifts from the
    // exact row maxima, partial results go to list-position slots
    int list_mode;
};
constexpr int EST_HI_CAP = 1024;     // keys of outlying norm kept per (batch, head); more than that is not an outlier set
constexpr int EST_KS2 = 16;          // key ranges of the second-chance pass
constexpr float EST_BIAS = 1048576.f;   // rowshift is stored + 2^20 (resolution 1/8 bit): positive floats order like their bits
// Layout of est_ws (BH = B * H; nWG = ceil(Nk / 32) key-scan workgroups; NqL = second-chance row capacity per (batch, head)
// = Nq / 8 rounded up to 128; NqP = Nq rounded up to 16):
//   rowshift f32 [BH][Nq]   shift of every query row (+ EST_BIAS), written by the pre-pass
//   rowlist  i32 [BH][Nq]   ascending list of the rows the static kernel did not vouch for; rowcount i32 [BH]
//   hicount  i32 [BH], hilist i32 [BH][EST_HI_CAP]   keys of outlying norm (hicount > EST_HI_CAP: not an outlier set, ignored)
//   dense    i32 [BH]       set by a key-scan workgroup that saw > 4 such keys among its 32 rows
//   wgcnt    i32 [BH][nWG], wglist i32 [BH][nWG][4]    per-workgroup finds of the key scan (no atomics), compacted into hilist
//   slotrow  i32 [BH][NqS]  (NqS = Nq rounded up to 256) the rows of every 256-row query tile in the order the 256-row static
//            kernel deals them to its lanes: sorted by shift, so that the two rows a lane owns are neighbours in that order
//   pmax     f32 [EST_KS2][BH][NqL]                    second chance: exact row maxima per key range
//   l2       f32 [EST_KS2][BH][NqL], o2 16-bit [EST_KS2][BH][NqL][64]   second chance: partial row sums / outputs
//   rowflag  u8  [BH][NqP]  (16-byte aligned) 1 = row handed over
struct EstView {
    float* rowshift;
    int* rowlist;
    int* rowcount;
    int* hicount;
    int* dense;
    int* hilist;
    int* wgcnt;
    int* wglist;
    int* slotrow;
    float* pmax;
    float* l2;
    bf16_t* o2;
    unsigned char* rowflag;
    int NqP, NqL, nWG, NqS;
};
__host__ __device__ inline long est_nql(long Nq) { return ((Nq + 7) / 8 + 127) / 128 * 128; }
// byte offsets of the sub-arrays (in the order of the layout above) and the total size
struct EstOffsets {
    long rowlist, rowcount, hicount, dense, hilist, wgcnt, wglist, slotrow, pmax, l2, o2, rowflag, total;
    int NqP, NqL, nWG, NqS;
};
__host__ __device__ inline EstOffsets est_offsets(int B, int H, int Nq, int Nk) {
    const long BH = (long)B * H, nq = Nq;
    EstOffsets o;
    o.NqP = (int)((nq + 15) / 16 * 16);
    o.NqL = (int)est_nql(nq);
    o.nWG = (Nk + 31) / 32;
    o.NqS = (int)((nq + 255) / 256 * 256);
    long w = BH * nq * 4;                        // rowshift sits at offset 0
    o.rowlist = w;   w += BH * nq * 4;
    o.rowcount = w;  w += BH * 4;
    o.hicount = w;   w += BH * 4;
    o.dense = w;     w += BH * 4;
    o.hilist = w;    w += BH * EST_HI_CAP * 4;
    o.wgcnt = w;     w += BH * (long)o.nWG * 4;
    o.wglist = w;    w += BH * (long)o.nWG * 16;
    o.slotrow = w;   w += BH * (long)o.NqS * 4;
    o.pmax = w;      w += (long)EST_KS2 * BH * o.NqL * 4;
    o.l2 = w;        w += (long)EST_KS2 * BH * o.NqL * 4;
    w = (w + 15) / 16 * 16;
    o.o2 = w;        w += (long)EST_KS2 * BH * o.NqL * 128;
    o.rowflag = w;   w += BH * o.NqP;
    o.total = w;
    return o;
}
__host__ __device__ inline long est_ws_size(int B, int H, int Nq, int Nk) { return est_offsets(B, H, Nq, Nk).total; }
__host__ __device__ inline EstView est_view(const AttnParams& p) {
    const EstOffsets o = est_offsets(p.B, p.H, p.Nq, p.Nk);
    unsigned char* b = p.est_ws;
    EstView v;
    v.NqP = o.NqP; v.NqL = o.NqL; v.nWG = o.nWG; v.NqS = o.NqS;
    v.rowshift = reinterpret_cast<float*>(b);
    v.rowlist = reinterpret_cast<int*>(b + o.rowlist);
    v.rowcount = reinterpret_cast<int*>(b + o.rowcount);
    v.hicount = reinterpret_cast<int*>(b + o.hicount);
    v.dense = reinterpret_cast<int*>(b + o.dense);
    v.hilist = reinterpret_cast<int*>(b + o.hilist);
    v.wgcnt = reinterpret_cast<int*>(b + o.wgcnt);
    v.wglist = reinterpret_cast<int*>(b + o.wglist);
    v.slotrow = reinterpret_cast<int*>(b + o.slotrow);
    v.pmax = reinterpret_cast<float*>(b + o.pmax);
    v.l2 = reinterpret_cast<float*>(b + o.l2);
    v.o2 = reinterpret_cast<bf16_t*>(b + o.o2);
    v.rowflag = b + o.rowflag;
    return v;
}
constexpr int GUARD_RETRY_DEFAULT = 16;
constexpr int GUARD_WORDS = 8;
// resolve the guard word(s) to "skip the static-bound kernel in this call"
IGGT_DEVINL bool guard_skips(const int* guard, const int* guard_prev) {
    if (guard == nullptr) return false;
    int g = guard[0];
    if (g < 0) g = (guard_prev != nullptr && guard_prev[0] > 0) ? 1 : 0;
    return g > 0;
}
// ... and to the mode the static kernel runs in (0: norm bound, 1: estimated shift); a call site that has never been measured
// inherits the mode of the same kind of launch one layer earlier
IGGT_DEVINL int guard_mode(const AttnParams& p) {
    if (p.est_ws == nullptr) return 0;
    if (p.guard == nullptr) return p.est_force;
    if (p.guard[0] < 0) return p.guard_prev != nullptr ? p.guard_prev[4] : 0;
    return p.guard[4];
}

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

// softmax numerators -> packed operand fragment (no saturation: P <= 2^(DEFER_THR + P_SHIFT))
template <int FMT>
IGGT_DEVINL bf16x8 pack8h(const f32x16& s, int base) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = pack_h2<FMT, false>(s[base + 2 * j], s[base + 2 * j + 1]);
    return __builtin_bit_cast(bf16x8, r);
}
// fp16 operands: the numerators are computed as 2^(s - m + P_SHIFT) so that entries far below the row maximum stay
// in fp16's normal range (smallest normal 6.1e-5 * 2^-8 = 2.4e-7 of the maximum, subnormals to 2.3e-10); the
// shift cancels in O / l.  bf16 has fp32's exponent range and needs none.
constexpr float P_SHIFT_F16 = 8.0f;
// static-bound kernel: numerators 2^(s - c_h + 15) <= 2^15 (fp16 max 2^16).  A row is accepted when its row sum l is at
// least Nk * 2^-13: numerators below fp16's smallest subnormal (2^-24) flush to zero and those between 2^-24 and 2^-14 are
// rounded to a 2^-25 grid, so the mass a row can lose is < Nk * 2^-24 -- at most 2^-11 = 5e-4 of an accepted row's sum in
// the contrived worst case (almost all keys just under the flush threshold), orders of magnitude less for any score
// distribution that is not bimodal.  bf16 numerators have fp32's exponent range: no shift, and the acceptance threshold only
// guards against fp32 underflow of l.
constexpr float STATIC_SHIFT_F16 = 15.0f;
constexpr float STATIC_MIN_L_PER_KEY_F16 = 1.0f / 8192.0f;
constexpr float STATIC_MIN_L_BF16 = 1e-30f;

// estimated-shift mode: headroom (log2 units) between the sampled row maximum and the top of the fp16 range; the sampled
// maximum itself lands on 2^(STATIC_SHIFT_F16 - slack), which must stay above the acceptance threshold Nk * 2^-13 (host side)
inline float est_slack_for(int Nk) {
    // largest s with 2^(15 - s) >= 1.25 * Nk * 2^-13, i.e. s <= 28 - log2(Nk) - 0.33; at most 12
    int s = 12;
    while (s > 4 && (float)(1L << (15 - s)) < 1.25f * (float)Nk * STATIC_MIN_L_PER_KEY_F16) --s;
    return (float)s;
}

}  // namespace iggt_attn

// experimental variants live in their own translation units
int iggt_launch_flash_attn_v3(const iggt_attn::AttnParams& p, int q_rows, int kvm, int fmt, bool static_bound,
                              hipStream_t stream);
int iggt_launch_attn_combine(const iggt_attn::AttnParams& p, int nslots, int q_rows, int fmt, hipStream_t stream);
// attention_est.hip: key scan (outlying norms), row-shift pre-pass, flagged-row compaction
int iggt_launch_attn_est_prepass(const iggt_attn::AttnParams& p, int key_period, int key_nspecial, int fmt, int dbg,
                                 hipStream_t stream);
int iggt_launch_attn_rowlist(const iggt_attn::AttnParams& p, hipStream_t stream);
// second chance for the listed rows: exact row maxima per key range -> static kernel in list mode -> fold (attention_est.hip)
int iggt_launch_attn_second_chance(const iggt_attn::AttnParams& p, int fmt, hipStream_t stream);
// attention_v3.hip: the 128-row static kernel over the listed rows and one key range per workgroup
int iggt_launch_flash_attn_v3_list(const iggt_attn::AttnParams& p, int fmt, hipStream_t stream);
