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
// This file is the host-side dispatcher (tile selection, argument contract, the two-pass static-bound launch); the
// kernel itself -- swapped QK^T with in-register P, LDS-DMA staged K/V macro tiles, XCD-chunked 1-D grid -- is
// attention_v3.hip, whose header carries the CDNA4 mapping and the measured ladder.  The round-1 baseline kernels
// ("v1" register-staged and the 8-wave ping-pong variant) are kept un-built under probes/legacy/.
#include <stdio.h>
#include <stdlib.h>

#include "attention_common.h"
#include "../../include/iggt_hip.h"

using namespace iggt_attn;

namespace {

int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                  ? prop.multiProcessorCount : 256;
    }
    return cus;
}

// Tile choice fitted to measurements (probes/attn_tiles.py, profiles/r01_microbench_kernels.txt): the 256-row tile (2 resident
// workgroups per CU) is ~10 % faster per query row, unless (a) it cannot put >= 1.2 rounds of workgroups on the chip -- the
// per-rank global attention of an 8-GPU run has 352 -- or (b) it pads the sequence > 5 % more than the 128-row tile
// (1374-token frames).  Codes: 5128 / 5256 = 128 / 256 query rows with 64-key macro tiles, 6128 / 6256 with 128-key ones.
int pick_tile(int B, int H, int Nq, int code) {
    if (code != 0) return code;
    const long w256 = (long)B * H * ((Nq + 255) / 256);
    const int pad256 = ((Nq + 255) / 256) * 256, pad128 = ((Nq + 127) / 128) * 128;
    const bool small_grid = w256 * 10 < (long)device_cus() * 2 * 12;
    // (b) only while the grid is a few rounds: from four full rounds of 256-row tiles on, the big tile wins despite its padding
    // (frame attention of 32 views, 32 x 16 sequences of 1 374 tokens = 3 072 tiles: 326 vs 346 us static, 337 vs 350 us
    // online-max, probes/attn_frame_codes.py; 4 views: 66 vs 61 us, the small tile stays)
    const bool pads_more = (long)pad256 * 100 > (long)pad128 * 105 && w256 < (long)device_cus() * 2 * 4;
    return (small_grid || pads_more) ? 5128 : 6256;
}

bool valid_code(int c) { return c == 5128 || c == 5256 || c == 6128 || c == 6256; }

int fill_params(AttnParams& p, const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs, long o_rs) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return -1;
    if ((q_rs % 8) || (k_rs % 8) || (v_rs % 8) || (o_rs % 4)) return -2;
    if ((q_bs % 8) || (k_bs % 8) || (v_bs % 8) || (o_bs % 4)) return -2;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs;
    p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.scale_log2 = 1.f; p.qtiles = 0; p.qkmax = nullptr; p.flags = nullptr; p.static_min_l = 0.f;
    p.ksplit = 0; p.slot0 = 0; p.seg_len = 0; p.skip_seg = -1; p.seg_kmax = nullptr; p.o_part = nullptr; p.l_part = nullptr; p.c_part = nullptr;
    p.guard = nullptr; p.guard_prev = nullptr; p.guard_retry = GUARD_RETRY_DEFAULT;
    p.est_ws = nullptr; p.est_force = 0; p.list_mode = 0;
    return 0;
}

// Key ranges for a grid too small to fill the chip with 256-row tiles (cost in units of one full-length 256-row workgroup
// at 2 workgroups per CU: rounds / s; the unsplit alternative -- 128-row tiles -- measures ~1.0 at the per-rank shape of an
// 8-GPU run).  0: one pass.
int choose_ksplit(int B, int H, int Nq, int Nk) {
    const long w256 = (long)B * H * ((Nq + 255) / 256), slots = (long)device_cus() * 2;
    if (w256 * 10 >= slots * 12) return 0;
    const int nmt = (Nk + 127) / 128;
    int best = 0;
    double best_cost = 0.92;
    for (int s = 2; s <= 8; ++s) {
        if (nmt / s < 8) break;
        const double cost = (double)((w256 * s + slots - 1) / slots) / s + 0.02;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}

long part_ws_bytes(int slots, int B, int H, int Nq) {
    return (long)slots * B * Nq * ((long)H * 64 * 2 + (long)H * 4 * 2);   // O rows (16 bit) + row sums + row shifts (fp32)
}

}  // namespace

static int flash_attn_h16(int fmt, const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                          long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs, long o_rs,
                          float scale, int q_rows_per_wg, void* stream) {
    AttnParams p;
    const int rc = fill_params(p, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs);
    if (rc) return rc;
    p.scale_log2 = scale * 1.4426950408889634f;
    const int code = pick_tile(B, H, Nq, q_rows_per_wg);
    if (!valid_code(code)) return -3;
    iggt_launch_flash_attn_v3(p, code % 1000, code / 1000 - 4, fmt, false, (hipStream_t)stream);
    IGGT_CHECK_LAUNCH();
    return 0;
}

// One pass of the static-bound kernel over a key segment into partial slots [slot0, slot0 + ksplit).
static int flash_attn_static_partial_h16(int fmt, const void* q, const void* k, const void* v, int B, int H, int Nq, int Nk,
                                         long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                         const float* qkmax, void* o_part, float* l_part, float* c_part, int slot0,
                                         int ksplit, int seg_len, int skip_seg, const float* seg_kmax, int q_rows_per_wg,
                                         const int* guard, const int* guard_prev, void* stream) {
    AttnParams p;
    const int rc = fill_params(p, q, k, v, o_part, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, 0, 64);
    if (rc) return rc;
    if (o_part == nullptr || l_part == nullptr || c_part == nullptr || H > 16 || ksplit < 1 || slot0 < 0) return -5;
    if (qkmax == nullptr && !(seg_len > 0 && seg_kmax != nullptr)) return -5;   // key bounds: per segment, or one for all
    if (seg_kmax != nullptr && seg_len <= 0) return -5;
    p.guard = const_cast<int*>(guard); p.guard_prev = guard_prev; p.c_part = c_part;
    const int code = q_rows_per_wg ? q_rows_per_wg : 6256;
    if (!valid_code(code)) return -3;
    const int kvm = code / 1000 - 4;
    if (seg_len > 0) {   // one range per key segment of seg_len rows (the last may be shorter), optionally leaving one out
        if (ksplit != (Nk + seg_len - 1) / seg_len || skip_seg < -1 || skip_seg >= ksplit) return -7;
    } else if ((Nk + 64 * kvm - 1) / (64 * kvm) < ksplit || skip_seg != -1) {
        return -7;
    }
    p.seg_len = seg_len > 0 ? seg_len : 0; p.skip_seg = skip_seg; p.seg_kmax = seg_kmax;
    p.qkmax = qkmax; p.ksplit = ksplit; p.slot0 = slot0; p.o_part = (bf16_t*)o_part; p.l_part = l_part;
    iggt_launch_flash_attn_v3(p, code % 1000, kvm, fmt, true, (hipStream_t)stream);
    IGGT_CHECK_LAUNCH();
    return 0;
}

// Fold nslots partial results into o, flag the rows below the acceptance threshold and redo their tiles (online-max kernel
// over all Nk keys).
static int flash_attn_static_combine_h16(int fmt, const void* o_part, const float* l_part, const float* c_part, int nslots,
                                         const void* q, const void* k, const void* v, void* o, int B, int H, int Nq, int Nk,
                                         long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs, long o_rs,
                                         int* flags, int flags_len, int q_rows_per_wg, int* guard, const int* guard_prev,
                                         void* stream) {
    AttnParams p;
    const int rc = fill_params(p, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs);
    if (rc) return rc;
    if (o_part == nullptr || l_part == nullptr || c_part == nullptr || flags == nullptr || nslots < 1) return -5;
    p.guard = guard; p.guard_prev = guard_prev; p.c_part = const_cast<float*>(c_part);
    const int code = q_rows_per_wg ? q_rows_per_wg : 6256;
    if (!valid_code(code)) return -3;
    const int rows = code % 1000;
    const long nwork = (long)B * H * ((Nq + rows - 1) / rows);
    if (nwork > flags_len) return -6;
    const hipError_t e = hipMemsetAsync(flags, 0, nwork * sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    p.flags = flags; p.o_part = (bf16_t*)const_cast<void*>(o_part); p.l_part = const_cast<float*>(l_part);
    p.static_min_l = fmt == FMT_F16 ? STATIC_MIN_L_PER_KEY_F16 * (float)Nk : STATIC_MIN_L_BF16;
    iggt_launch_attn_combine(p, nslots, rows, fmt, (hipStream_t)stream);
    IGGT_CHECK_LAUNCH();
    p.o_part = nullptr; p.l_part = nullptr; p.c_part = nullptr;
    iggt_launch_flash_attn_v3(p, rows, code / 1000 - 4, fmt, false, (hipStream_t)stream);   // gated on the flags; updates the guard
    IGGT_CHECK_LAUNCH();
    return 0;
}

// Static-bound launch: the fast kernel over every query tile, then the dynamic kernel over the tiles it flagged.  With a
// partial workspace and a grid too small for 256-row tiles the keys are split into ranges (choose_ksplit).
static int flash_attn_static_h16(int fmt, const void* q, const void* k, const void* v, void* o, int B, int H, int Nq,
                                 int Nk, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs,
                                 long o_rs, const float* qkmax, int* flags, int flags_len, void* part_ws, long part_ws_len,
                                 int q_rows_per_wg, int* guard, const int* guard_prev, void* est_ws, long est_ws_len,
                                 int key_period, int key_nspecial, int est_mode, void* stream) {
    AttnParams p;
    const int rc = fill_params(p, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs);
    if (rc) return rc;
    if (qkmax == nullptr || flags == nullptr || H > 16) return -5;
    if (q_rows_per_wg == 0 && part_ws != nullptr) {
        const int ks = choose_ksplit(B, H, Nq, Nk);
        if (ks > 1 && part_ws_bytes(ks, B, H, Nq) <= part_ws_len) {
            char* ws = (char*)part_ws;
            float* l_part = (float*)(ws + (long)ks * B * Nq * H * 64 * 2);
            float* c_part = l_part + (long)ks * B * Nq * H;
            int r = flash_attn_static_partial_h16(fmt, q, k, v, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, qkmax, ws,
                                                  l_part, c_part, 0, ks, 0, -1, nullptr, 6256, guard, guard_prev, stream);
            if (r) return r;
            return flash_attn_static_combine_h16(fmt, ws, l_part, c_part, ks, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs,
                                                 v_bs, v_rs, o_bs, o_rs, flags, flags_len, 6256, guard, guard_prev, stream);
        }
    }
    const int code = pick_tile(B, H, Nq, q_rows_per_wg);
    if (!valid_code(code)) return -3;
    const int rows = code % 1000;
    const long nwork = (long)B * H * ((Nq + rows - 1) / rows);
    if (nwork > flags_len) return -6;
    const hipError_t e = hipMemsetAsync(flags, 0, nwork * sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    p.qkmax = qkmax; p.flags = flags; p.guard = guard; p.guard_prev = guard_prev;
    p.static_min_l = fmt == FMT_F16 ? STATIC_MIN_L_PER_KEY_F16 * (float)Nk : STATIC_MIN_L_BF16;
    const bool est = est_ws != nullptr;
    // IGGT_EST_DEBUG (developer bit mask, bisecting FAULTS -- results are not meaningful): 1 no memset of the dense marks, 2 no key scan, 4 no pre-pass, 8 no
    // estimated-shift instantiation, 16 no row list (the online-max pass then sees no workspace), 32 no second chance
    static int dbg = -1;
    if (dbg < 0) {
        const char* e = getenv("IGGT_EST_DEBUG");
        dbg = e ? atoi(e) : 0;
    }
    if (est) {   // row-granular hand-over + (adaptive, or est_mode = 1 without a guard) the estimated shift
        if (est_ws_len < est_ws_size(B, H, Nq, Nk) || ((uintptr_t)est_ws % 16)) return -8;
        p.est_ws = (unsigned char*)est_ws;
        p.est_force = est_mode ? 1 : 0;
        const int r = iggt_launch_attn_est_prepass(p, key_period, key_nspecial, fmt, dbg, (hipStream_t)stream);
        if (r) return r;
        IGGT_CHECK_LAUNCH();
    }
    if (dbg & 8) {
        AttnParams q2 = p;
        q2.est_ws = nullptr;
        iggt_launch_flash_attn_v3(q2, rows, code / 1000 - 4, fmt, true, (hipStream_t)stream);
    } else {
        iggt_launch_flash_attn_v3(p, rows, code / 1000 - 4, fmt, true, (hipStream_t)stream);
    }
    IGGT_CHECK_LAUNCH();
    if (est && !(dbg & 16)) {
        iggt_launch_attn_rowlist(p, (hipStream_t)stream);
        IGGT_CHECK_LAUNCH();
        if (!(dbg & 32)) {
            const int r = iggt_launch_attn_second_chance(p, fmt, (hipStream_t)stream);
            if (r) return r;
            IGGT_CHECK_LAUNCH();
        }
    }
    if (dbg & 16) p.est_ws = nullptr;
    p.qkmax = nullptr;   // gated dynamic pass: q already carries scale * log2 e
    iggt_launch_flash_attn_v3(p, rows, code / 1000 - 4, fmt, false, (hipStream_t)stream);
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

extern "C" int iggt_flash_attn_static_bf16_d64(const void* q, const void* k, const void* v, void* o, int B, int H,
                                   int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                                   long v_bs, long v_rs, long o_bs, long o_rs, const float* qkmax,
                                   int* flags, int flags_len, void* part_ws, long part_ws_bytes_len, int q_rows_per_wg,
                                   int* guard, const int* guard_prev, void* est_ws, long est_ws_len, int key_period,
                                   int key_nspecial, int est_mode, void* stream) {
    return flash_attn_static_h16(FMT_BF16, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                                 qkmax, flags, flags_len, part_ws, part_ws_bytes_len, q_rows_per_wg, guard, guard_prev,
                                 est_ws, est_ws_len, key_period, key_nspecial, est_mode, stream);
}

extern "C" int iggt_flash_attn_static_partial_bf16_d64(const void* q, const void* k, const void* v, int B, int H, int Nq,
                                           int Nk, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                           const float* qkmax, void* o_part, float* l_part, float* c_part, int slot0,
                                           int ksplit, int seg_len, int skip_seg, const float* seg_kmax, int q_rows_per_wg,
                                           const int* guard, const int* guard_prev, void* stream) {
    return flash_attn_static_partial_h16(FMT_BF16, q, k, v, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, qkmax, o_part,
                                         l_part, c_part, slot0, ksplit, seg_len, skip_seg, seg_kmax, q_rows_per_wg, guard,
                                         guard_prev, stream);
}

extern "C" int iggt_flash_attn_static_combine_bf16_d64(const void* o_part, const float* l_part, const float* c_part,
                                           int nslots, const void* q, const void* k, const void* v, void* o, int B, int H,
                                           int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                           long o_bs, long o_rs, int* flags, int flags_len, int q_rows_per_wg, int* guard,
                                           const int* guard_prev, void* stream) {
    return flash_attn_static_combine_h16(FMT_BF16, o_part, l_part, c_part, nslots, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs,
                                         k_rs, v_bs, v_rs, o_bs, o_rs, flags, flags_len, q_rows_per_wg, guard, guard_prev,
                                         stream);
}

extern "C" int iggt_flash_attn_static_f16_d64(const void* q, const void* k, const void* v, void* o, int B, int H,
                                   int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs,
                                   long v_bs, long v_rs, long o_bs, long o_rs, const float* qkmax,
                                   int* flags, int flags_len, void* part_ws, long part_ws_bytes_len, int q_rows_per_wg,
                                   int* guard, const int* guard_prev, void* est_ws, long est_ws_len, int key_period,
                                   int key_nspecial, int est_mode, void* stream) {
    return flash_attn_static_h16(FMT_F16, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                                 qkmax, flags, flags_len, part_ws, part_ws_bytes_len, q_rows_per_wg, guard, guard_prev,
                                 est_ws, est_ws_len, key_period, key_nspecial, est_mode, stream);
}

extern "C" int iggt_flash_attn_static_partial_f16_d64(const void* q, const void* k, const void* v, int B, int H, int Nq,
                                           int Nk, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                           const float* qkmax, void* o_part, float* l_part, float* c_part, int slot0,
                                           int ksplit, int seg_len, int skip_seg, const float* seg_kmax, int q_rows_per_wg,
                                           const int* guard, const int* guard_prev, void* stream) {
    return flash_attn_static_partial_h16(FMT_F16, q, k, v, B, H, Nq, Nk, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, qkmax, o_part,
                                         l_part, c_part, slot0, ksplit, seg_len, skip_seg, seg_kmax, q_rows_per_wg, guard,
                                         guard_prev, stream);
}

extern "C" int iggt_flash_attn_static_combine_f16_d64(const void* o_part, const float* l_part, const float* c_part,
                                           int nslots, const void* q, const void* k, const void* v, void* o, int B, int H,
                                           int Nq, int Nk, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                           long o_bs, long o_rs, int* flags, int flags_len, int q_rows_per_wg, int* guard,
                                           const int* guard_prev, void* stream) {
    return flash_attn_static_combine_h16(FMT_F16, o_part, l_part, c_part, nslots, q, k, v, o, B, H, Nq, Nk, q_bs, q_rs, k_bs,
                                         k_rs, v_bs, v_rs, o_bs, o_rs, flags, flags_len, q_rows_per_wg, guard, guard_prev,
                                         stream);
}

// Name of the kernel instantiation the dispatcher launches for a shape (reports / bench.py: the roofline entry must name
// the kernel that actually ran).  Host-only, no launch.
extern "C" int iggt_flash_attn_d64_kernel_name(int B, int H, int Nq, int Nk, int f16, int static_bound, int with_part_ws,
                                               int q_rows_per_wg, char* buf, int buf_len) {
    if (buf == nullptr || buf_len < 96) return -1;
    const int ks = (static_bound && with_part_ws && q_rows_per_wg == 0) ? choose_ksplit(B, H, Nq, Nk) : 0;
    const int code = ks > 1 ? 6256 : pick_tile(B, H, Nq, q_rows_per_wg);
    if (!valid_code(code)) return -3;
    if (ks > 1)
        snprintf(buf, buf_len, "flash_attn_d64_v3_kernel<QB=2,KVM=2,%s,static-bound,%d key ranges> + attn_combine_kernel",
                 f16 ? "f16" : "bf16", ks);
    else
        snprintf(buf, buf_len, "flash_attn_d64_v3_kernel<QB=%d,KVM=%d,%s,%s>", (code % 1000) / 128, code / 1000 - 4,
                 f16 ? "f16" : "bf16", static_bound ? "static-bound" : "online-max");
    return 0;
}

/* number of key ranges the static-bound dispatcher would cut (B, H, Nq, Nk) into (1: one pass) -- callers that drive the
 * partial / combine pair themselves (a rank's own keys in a view-sharded run) use it to fill a small grid */
extern "C" int iggt_flash_attn_static_ksplit(int B, int H, int Nq, int Nk) {
    const int ks = choose_ksplit(B, H, Nq, Nk);
    return ks > 1 ? ks : 1;
}

/* bytes of the estimated-shift / row-granular workspace of iggt_flash_attn_static_* for this shape */
extern "C" long iggt_flash_attn_static_est_ws_bytes(int B, int H, int Nq, int Nk) {
    return est_ws_size(B, H, Nq, Nk);
}

/* bytes of partial workspace iggt_flash_attn_static_* needs to be allowed to split the keys of this shape (0: never splits) */
extern "C" long iggt_flash_attn_static_ws_bytes(int B, int H, int Nq, int Nk) {
    const int ks = choose_ksplit(B, H, Nq, Nk);
    return ks > 1 ? part_ws_bytes(ks, B, H, Nq) : 0;
}
