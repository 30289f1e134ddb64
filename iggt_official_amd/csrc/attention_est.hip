// Estimated-shift static softmax: the helper kernels around flash_attn_d64_v3_kernel<STATIC> (round 4).
//
// Why.  The static-bound kernel (attention_v3.hip) needs, before the first key tile, a shift c_i with
//   (a) c_i >= max_j s_ij - headroom     (no numerator above the fp16 range), and
//   (b) c_i <= max_j s_ij + ~25 bits     (the row's largest numerators stay normal fp16 numbers).
// Cauchy-Schwarz, c_i = |q^_i| max_j |k^_j|, gives (a) unconditionally but (b) only while scores are "LayerNorm-of-noise"
// sized: with trained-like q/k-norm affines (heavy-tailed per-channel scales), a few sink keys of 10x the norm, or register /
// camera tokens with 30x the query norm the bound sits 40-100 bits above every real score, every tile is flagged and the
// call falls back to the online-max kernel (profiles/r03_attn_static_robustness.txt: 8.4-8.9 ms instead of 7.0-7.3 ms at
// N = 43 968, i.e. 0.36-0.38 of the MFMA peak instead of 0.44-0.46).
//
// What.  A LOWER bound of the row maximum is cheap: the exact maximum over a small key SAMPLE,
//   m_i = max_{j in S} s_ij <= max_j s_ij,   S = special tokens of every view + every stride-th key + keys of outlying norm,
// costs |S| / Nk of the QK^T work (2-3 % at |S| ~ 1 000).  The pre-pass below writes
//   shift_i = min( |q^_i| max_j |k^_j| ,  m_i + slack )
// i.e. the norm bound where it is tight anyway, else the sampled maximum plus `slack` bits of headroom.  With that shift
// (b) holds by construction (the sampled key itself contributes 2^(15 - slack) >= the acceptance threshold), and (a) can only
// fail when the true maximum lies more than slack + 1 bits (8.3 nats at slack = 11) above the sampled one: the numerator then
// rounds to +inf in fp16, the row's accumulator turns non-finite, and the row -- THE ROW, not its 256-row tile -- is handed to
// the online-max pass.  Why the sample contains what it contains:
//   * strided keys: for a row whose scores over the keys are roughly bell-shaped the maximum over 1 000 of 44 000 keys sits
//     ~1 standard deviation below the maximum over all of them (sqrt(2 ln n): 3.7 vs 4.6);
//   * special tokens (camera / register rows of every view, reference aggregator.py:338-361): where attention sinks live;
//   * keys of outlying norm (|k^_j| > max_j |k^_j| / 2, at most EST_HI_CAP per head; more than that is not an outlier set):
//     a sink ANYWHERE scores up to |q^||k^| -- the key scan finds them by norm, position-independent.
// Rows with an outlying QUERY norm (register tokens: 30x) have score spreads no sample can bracket; they overflow and are handed
// over row by row -- 160 rows x 16 heads at 32 views, instead of 512 flagged tiles.
//
// SECOND CHANCE.  What to do with the handed-over rows?  First measurement (profiles/r04_attn_static_robustness.txt, first
// table): giving them to the online-max kernel in list mode costs ~1 ms however FEW they are -- ONE workgroup streaming all
// 44 000 keys for its 256 rows takes as long as a full-length workgroup of the main launch, and nothing runs beside it.  So
// the listed rows get a second, EXACT static pass whose cost is proportional to their number: (1) their exact row maxima, one
// workgroup per (256 listed rows, 1/16 of the keys); (2) the 128-row static kernel in list mode, shift = exact maximum + 1 --
// nothing can overflow or underflow any more -- again one workgroup per key range, partial results to list-position slots; (3) a
// fold over the 16 ranges (same shift in every range: partial sums add).  Lists longer than Nq / 8 rows per head still go to
// the online-max pass (and make the adaptive switch give up on the static kernel for a while).
//
// Kernels (all read the adaptive-switch word and return at once when the call runs in another mode):
//   attn_keyscan_kernel     K rows -> per key-scan workgroup (32 rows) and head: the rows with |k^| > kmax / 2, at most 4
//                           (more: the head is marked "dense" -- not an outlier set); no atomics: with LayerNorm-of-noise
//                           keys EVERY key qualifies, and 44 000 same-address atomics per head cost 1 ms in the first version
//   attn_keycompact_kernel  per (batch, head): the workgroups' finds -> hilist / hicount, in key order (deterministic)
//   attn_rowshift_kernel    <0> the pre-pass: Q block x sampled keys on the matrix pipe, running maximum, shift_i;
//                           <1> exact maxima of the listed rows over one key range (second chance)
//   attn_rowlist_kernel     rowflag bytes -> ascending row list + count per (batch, head) (deterministic compaction)
//   attn_fold_list_kernel   second chance: sum of the key ranges' partial results -> output rows
#include "attention_common.h"

using namespace iggt_attn;

namespace {

IGGT_DEVINL bool est_active(const AttnParams& p) {
    return p.est_ws != nullptr && !guard_skips(p.guard, p.guard_prev) && guard_mode(p) == 1;
}

// grid (ceil(Nk / 32), B), 256 threads: 32 key rows x all heads; 8 lanes share one (row, head) vector of 128 bytes
template <int FMT>
__global__ __launch_bounds__(256) void attn_keyscan_kernel(const AttnParams p) {
    if (!est_active(p)) return;
    const EstView ev = est_view(p);
    __shared__ unsigned long long wmask[16][4];
    const int tid = threadIdx.x, piece = tid & 7, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x * 32 + (tid >> 3), b = blockIdx.y;
    const bool in = j < p.Nk;
    const bf16_t* row = p.k + (long)b * p.k_bs + (long)(in ? j : p.Nk - 1) * p.k_rs + piece * 8;
    for (int h = 0; h < p.H; ++h) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(row + h * 64);
        float a2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = h2_lo<FMT>(w[e]), hi = h2_hi<FMT>(w[e]);
            a2 += lo * lo + hi * hi;
        }
        a2 += __shfl_xor(a2, 1, 64);
        a2 += __shfl_xor(a2, 2, 64);
        a2 += __shfl_xor(a2, 4, 64);
        const float thr = 0.5f * p.qkmax[16 + h];
        const unsigned long long m = __ballot(in && piece == 0 && a2 > thr * thr);   // bit 8 r = row r of this wave
        if (lane == 0) wmask[h][wave] = m;
    }
    __syncthreads();
    if (tid < p.H) {
        const int h = tid;
        const long bh = (long)b * p.H + h;
        int n = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) n += __popcll(wmask[h][w]);
        int* cnt = ev.wgcnt + bh * ev.nWG + blockIdx.x;
        if (n > 4) {
            ev.dense[bh] = 1;   // benign race: every writer stores 1
            *cnt = 0;
        } else {
            *cnt = n;
            int* dst = ev.wglist + (bh * ev.nWG + blockIdx.x) * 4;
            int k = 0;
            for (int w = 0; w < 4; ++w) {
                unsigned long long m = wmask[h][w];
                while (m) {
                    const int bit = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    dst[k++] = blockIdx.x * 32 + w * 8 + (bit >> 3);
                }
            }
        }
    }
}

// grid B * H, 256 threads: hilist = the key-scan workgroups' finds in key order, hicount = their number (EST_HI_CAP + 1 when
// the head is dense or holds more than the cap); clears the dense mark for the next call
__global__ __launch_bounds__(256) void attn_keycompact_kernel(const AttnParams p) {
    if (!est_active(p)) return;
    const EstView ev = est_view(p);
    const int bh = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int wsum[4];
    const bool dense = ev.dense[bh] != 0;
    __syncthreads();
    if (tid == 0) ev.dense[bh] = 0;
    if (dense) {
        if (tid == 0) ev.hicount[bh] = EST_HI_CAP + 1;
        return;
    }
    const int* cnt = ev.wgcnt + (long)bh * ev.nWG;
    const int* lst = ev.wglist + (long)bh * ev.nWG * 4;
    int* out = ev.hilist + (long)bh * EST_HI_CAP;
    int base = 0;
    for (int w0 = 0; w0 < ev.nWG; w0 += 256) {
        const int w = w0 + tid;
        const int c = w < ev.nWG ? cnt[w] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w2 = 0; w2 < wave; ++w2) woff += wsum[w2];
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        int pos = base + woff + incl - c;
        for (int k = 0; k < c; ++k, ++pos)
            if (pos < EST_HI_CAP) out[pos] = lst[(long)w * 4 + k];
        base += total;
        __syncthreads();
    }
    if (tid == 0) ev.hicount[bh] = base;
}

// MODE 0 (pre-pass): grid B * H * ceil(Nq / 256) (XCD-chunked like the main kernel), 256 threads = 4 waves x 64 query rows;
//   keys = the sample.  MODE 1 (second chance): grid B * H * (NqL / 256) * EST_KS2; rows = the (batch, head)'s list, keys = key
//   range ks of EST_KS2.  Key tiles of 64 are gathered row by row (16-byte pieces, two per thread) into the main kernel's
//   swizzled K image, double-buffered, the next tile's pieces in flight during the MFMAs of the current one.
template <int FMT, int MODE>
__global__ __launch_bounds__(256) void attn_rowshift_kernel(const AttnParams p, int stride, int period, int nspecial,
                                                            float slack) {
    if (MODE == 0 ? !est_active(p) : (p.est_ws == nullptr || guard_skips(p.guard, p.guard_prev))) return;
    const EstView ev = est_view(p);
    __shared__ __attribute__((aligned(16))) char smem[2 * K_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    // (MODE 1 is a sparse launch whose rows concentrate on few heads: launch order, not the XCD-chunked one -- attention_v3.hip)
    int work = MODE == 1 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x), ks = 0;
    if (MODE == 1) {
        ks = work % EST_KS2;
        work /= EST_KS2;
    }
    const int qtiles = MODE == 0 ? (p.Nq + 255) / 256 : ev.NqL / 256 + (ev.NqL % 256 != 0);
    const int qt = work % qtiles, bh = work / qtiles;
    const int h = bh % p.H, b = bh / p.H;
    const int q_base = qt * 256 + wave * 64;
    int nlist = 0;
    const int* rlist = nullptr;
    if (MODE == 1) {
        nlist = ev.rowcount[bh];
        if (nlist == 0 || nlist > ev.NqL || qt * 256 >= nlist) return;
        rlist = ev.rowlist + (long)bh * p.Nq;
    }
    const bf16_t* qb_ptr = p.q + (long)b * p.q_bs + h * 64;
    const bf16_t* kb_ptr = p.k + (long)b * p.k_bs + h * 64;

    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q_base + qb * 32 + frow;
        if (MODE == 1) qr = rlist[qr < nlist ? qr : nlist - 1];
        else qr = qr < p.Nq ? qr : p.Nq - 1;
        const bf16_t* src = qb_ptr + (long)qr * p.q_rs + 8 * fhalf;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) qf[qb][kc] = *reinterpret_cast<const bf16x8*>(src + 16 * kc);
    }

    // MODE 0, the sample: [special tokens of every period][every stride-th key][keys of outlying norm]
    // MODE 1: the contiguous key range [k0, k0 + n_tot)
    int n_spec = 0, n_str = 0, n_tot, k0 = 0;
    const int* hl = nullptr;
    if (MODE == 0) {
        n_spec = nspecial > 0 ? ((p.Nk + period - 1) / period) * nspecial : 0;
        n_str = (p.Nk + stride - 1) / stride;
        const int cnt = ev.hicount[bh];
        n_tot = n_spec + n_str + (cnt <= EST_HI_CAP ? cnt : 0);
        hl = ev.hilist + (long)bh * EST_HI_CAP;
    } else {
        const int per = ((p.Nk + 63) / 64 + EST_KS2 - 1) / EST_KS2 * 64;
        k0 = ks * per;
        n_tot = p.Nk - k0 < per ? p.Nk - k0 : per;
        if (n_tot <= 0) {   // (an empty last range: nothing to contribute)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int lp = q_base + qb * 32 + frow;
                if (fhalf == 0 && lp < nlist) ev.pmax[((long)ks * p.B * p.H + bh) * ev.NqL + lp] = -INFINITY;
            }
            return;
        }
    }
    const int NT = (n_tot + 63) / 64;
    auto key_of = [&](int t) -> int {
        t = t < n_tot ? t : n_tot - 1;             // the last tile repeats the last key: a maximum does not mind
        int j;
        if (MODE == 1) j = k0 + t;
        else if (t < n_spec) j = (t / nspecial) * period + (t % nspecial);
        else if (t < n_spec + n_str) j = (t - n_spec) * stride;
        else j = hl[t - n_spec - n_str];
        return j < p.Nk ? j : p.Nk - 1;
    };
    // three tiles of key pieces in flight per thread: a gathered tile is two dependent global loads deep (the sample index, then
    // the key row), and with one tile of prefetch the loop ran at that latency, not at the matrix pipe's pace
    constexpr int PF = 3;
    u32x4 st[PF][2];
    auto gload = [&](int tile, u32x4 (&dst)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pid = tid + 256 * i, row = pid >> 3, slot = pid & 7;
            dst[i] = *reinterpret_cast<const u32x4*>(kb_ptr + (long)key_of(tile * 64 + row) * p.k_rs + slot * 8);
        }
    };
    auto lstore = [&](int buf, const u32x4 (&src)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pid = tid + 256 * i, row = pid >> 3, slot = pid & 7;
            *reinterpret_cast<u32x4*>(smem + buf * K_BYTES + swz_off(row, slot)) = src[i];
        }
    };
    int koff[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) koff[kc] = frow * 128 + ((((2 * kc + fhalf) ^ (frow >> 1)) & 7) << 4);

    float m[2] = {-INFINITY, -INFINITY};
    auto step = [&](int t, u32x4 (&reg)[2]) {
        lstore(t & 1, reg);
        __syncthreads();   // tile t visible; everyone is past the reads of tile t - 1 (the buffer tile t + 1 will overwrite)
        if (t + PF < NT) gload(t + PF, reg);
        const char* sK = smem + (t & 1) * K_BYTES;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int kvh = 0; kvh < 2; ++kvh) {
                f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kvh * 4096 + koff[kc]);
                    s = mfma32h<FMT>(kf, qf[qb][kc], s);
                }
                float mx = m[qb];
#pragma unroll
                for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(s[r], s[r + 1]), mx);   // v_max3_f32
                m[qb] = mx;
            }
        }
    };
#pragma unroll
    for (int d = 0; d < PF; ++d)
        if (d < NT) gload(d, st[d]);
    for (int t = 0; t < NT; t += PF) {
        step(t, st[0]);
        if (t + 1 < NT) step(t + 1, st[1]);
        if (t + 2 < NT) step(t + 2, st[2]);
    }
    float* skey = reinterpret_cast<float*>(smem);          // MODE 0: the staging buffers become the sort's key / value arrays
    int* sval = reinterpret_cast<int*>(smem + 1024);
    if (MODE == 0) __syncthreads();
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float mx = fmaxf(m[qb], __shfl_xor(m[qb], 32, 64));   // the two lane halves hold different keys of the same row
        const int qr = q_base + qb * 32 + frow;
        if (MODE == 1) {
            if (fhalf == 0 && qr < nlist) ev.pmax[((long)ks * p.B * p.H + bh) * ev.NqL + qr] = mx;
            continue;
        }
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
        const float cs = sqrtf(a2) * p.qkmax[16 + h] * 1.00002f + 1e-3f;   // the norm bound of attention_v3.hip, same rounding slack
        const float sh = fminf(cs, mx + slack);
        if (fhalf == 0) {
            if (qr < p.Nq) ev.rowshift[(long)bh * p.Nq + qr] = sh + EST_BIAS;
            const int li = wave * 64 + qb * 32 + frow;
            skey[li] = qr < p.Nq ? sh : INFINITY;                           // rows past the end sort last
            sval[li] = qr;
        }
    }
    if (MODE == 1) return;
    // The 256-row static kernel gives each lane TWO rows (query blocks 0 and 1) and applies the second row's own shift by 16
    // packed adds per key tile -- 0.3 ms per launch at N = 43 968.  Waves whose 32 row pairs all agree to within 8 bits skip
    // those adds (a second copy of the tile loop, chosen per wave), so the rows of the tile are dealt to the lanes in the order
    // of their shift: slot (wave w, block qb, lane row f) gets sorted entry 2 (32 w + f) + qb, the second block the larger
    // shift of the pair (attention_v3.hip IGGT_EST_DELTA_MAX for why that direction is the safe one).
    __syncthreads();
    for (int k = 2; k <= 256; k <<= 1) {        // bitonic sort of the tile's 256 (shift, row) pairs, ascending; ties by row
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            const int partner = tid ^ jj;
            if (partner > tid) {
                const float a = skey[tid], bk = skey[partner];
                const int av = sval[tid], bv = sval[partner];
                const bool up = (tid & k) == 0;
                const bool gt = a > bk || (a == bk && av > bv);
                if (gt == up) {
                    skey[tid] = bk; skey[partner] = a;
                    sval[tid] = bv; sval[partner] = av;
                }
            }
            __syncthreads();
        }
    }
    ev.slotrow[(long)bh * ev.NqS + qt * 256 + tid] = sval[2 * (32 * (tid >> 6) + (tid & 31)) + ((tid >> 5) & 1)];
}

// grid B * H, 256 threads: ascending list of the flagged rows of one (batch, head); 4 096 flag bytes per round
__global__ __launch_bounds__(256) void attn_rowlist_kernel(const AttnParams p) {
    const int bh = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const EstView ev = est_view(p);
    if (guard_skips(p.guard, p.guard_prev) || guard_mode(p) != 1) {
        // the static kernel did not run (every tile is flagged instead) or ran under the norm bound (whole tiles flagged)
        if (tid == 0) ev.rowcount[bh] = 0;
        return;
    }
    __shared__ int wsum[4];
    const unsigned char* f = ev.rowflag + (long)bh * ev.NqP;
    int* out = ev.rowlist + (long)bh * p.Nq;
    int base = 0;
    for (int c0 = 0; c0 < p.Nq; c0 += 4096) {
        const int i0 = c0 + tid * 16;
        u32x4 w = {0u, 0u, 0u, 0u};
        if (i0 < ev.NqP) w = *reinterpret_cast<const u32x4*>(f + i0);
        unsigned mask = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (((w[e >> 2] >> (8 * (e & 3))) & 0xffu) != 0u && i0 + e < p.Nq) mask |= 1u << e;
        const int cnt = __popc(mask);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w2 = 0; w2 < wave; ++w2) woff += wsum[w2];
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        int pos = base + woff + incl - cnt;
        while (mask) {
            const int e = __ffs(mask) - 1;
            mask &= mask - 1;
            out[pos++] = i0 + e;
        }
        base += total;
        __syncthreads();
    }
    if (tid == 0) ev.rowcount[bh] = base;
}

// second chance, step 3: grid B * H * (NqL / 16), 256 threads = 16 listed rows x 16 threads x 4 output elements.  Every key
// range ran under the same shift (the exact row maximum + 1), so o = sum_s l_s O_s / sum_s l_s.
template <int FMT>
__global__ __launch_bounds__(256) void attn_fold_list_kernel(const AttnParams p) {
    if (p.est_ws == nullptr || guard_skips(p.guard, p.guard_prev)) return;
    const EstView ev = est_view(p);
    const int per_bh = ev.NqL / 16;
    const int bh = blockIdx.x / per_bh, lp = (blockIdx.x % per_bh) * 16 + (threadIdx.x >> 4);
    const int n = ev.rowcount[bh];
    if (n == 0 || n > ev.NqL || lp >= n) return;
    const int h = bh % p.H, b = bh / p.H, d = (threadIdx.x & 15) * 4;
    const long BH = (long)p.B * p.H;
    float L = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < EST_KS2; ++s) {
        const long li = ((long)s * BH + bh) * ev.NqL + lp;
        const float l = ev.l2[li];
        const u32x2 w = *reinterpret_cast<const u32x2*>(ev.o2 + li * 64 + d);
        L += l;
        acc[0] += l * h2_lo<FMT>(w[0]); acc[1] += l * h2_hi<FMT>(w[0]);
        acc[2] += l * h2_lo<FMT>(w[1]); acc[3] += l * h2_hi<FMT>(w[1]);
    }
    const float inv = 1.0f / L;
    u32x2 o;
    o[0] = pack_h2<FMT>(acc[0] * inv, acc[1] * inv);
    o[1] = pack_h2<FMT>(acc[2] * inv, acc[3] * inv);
    const int row = ev.rowlist[(long)bh * p.Nq + lp];
    *reinterpret_cast<u32x2*>(p.o + (long)b * p.o_bs + (long)row * p.o_rs + h * 64 + d) = o;
}

}  // namespace

// dbg: developer bit mask (IGGT_EST_DEBUG, attention.hip) -- 1: no memset of the dense marks, 2: no key scan / compaction,
// 4: no pre-pass
int iggt_launch_attn_est_prepass(const AttnParams& p, int key_period, int key_nspecial, int fmt, int dbg, hipStream_t stream) {
    const EstView ev = est_view(p);
    if (!(dbg & 1)) {
        const hipError_t e = hipMemsetAsync(ev.dense, 0, (size_t)p.B * p.H * sizeof(int), stream);
        if (e != hipSuccess) return (int)e;
    }
    // ~Nk / 64 strided keys, between 128 (frame attention: 1 374 keys) and 512 (global attention): the expected maximum of n
    // bell-shaped scores grows like sqrt(2 ln n) -- 3.5 sigma at 512, 3.7 at 1 200, 4.6 at 44 000
    int target = p.Nk / 64;
    target = target < 128 ? 128 : (target > 512 ? 512 : target);
    int stride = p.Nk / target;
    if (stride < 1) stride = 1;
    const int period = key_period > 0 ? key_period : p.Nk;
    const int nspecial = (key_nspecial > 0 && key_nspecial < period) ? key_nspecial : 0;
    const float slack = est_slack_for(p.Nk);
    const dim3 g1((p.Nk + 31) / 32, p.B), block(256);
    const dim3 g2((unsigned)((long)p.B * p.H * ((p.Nq + 255) / 256)));
    if (!(dbg & 2)) {
        if (fmt == FMT_F16) hipLaunchKernelGGL(attn_keyscan_kernel<FMT_F16>, g1, block, 0, stream, p);
        else hipLaunchKernelGGL(attn_keyscan_kernel<FMT_BF16>, g1, block, 0, stream, p);
        hipLaunchKernelGGL(attn_keycompact_kernel, dim3(p.B * p.H), block, 0, stream, p);
    }
    if (!(dbg & 4)) {
        if (fmt == FMT_F16) hipLaunchKernelGGL((attn_rowshift_kernel<FMT_F16, 0>), g2, block, 0, stream, p, stride, period, nspecial, slack);
        else hipLaunchKernelGGL((attn_rowshift_kernel<FMT_BF16, 0>), g2, block, 0, stream, p, stride, period, nspecial, slack);
    }
    return 0;
}

int iggt_launch_attn_rowlist(const AttnParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(attn_rowlist_kernel, dim3(p.B * p.H), dim3(256), 0, stream, p);
    return 0;
}

int iggt_launch_attn_second_chance(const AttnParams& p, int fmt, hipStream_t stream) {
    const EstView ev = est_view(p);
    const dim3 block(256);
    const dim3 g1((unsigned)((long)p.B * p.H * ((ev.NqL + 255) / 256) * EST_KS2));
    if (fmt == FMT_F16) hipLaunchKernelGGL((attn_rowshift_kernel<FMT_F16, 1>), g1, block, 0, stream, p, 1, 1, 0, 0.f);
    else hipLaunchKernelGGL((attn_rowshift_kernel<FMT_BF16, 1>), g1, block, 0, stream, p, 1, 1, 0, 0.f);
    const int rc = iggt_launch_flash_attn_v3_list(p, fmt, stream);
    if (rc) return rc;
    const dim3 g3((unsigned)((long)p.B * p.H * (ev.NqL / 16)));
    if (fmt == FMT_F16) hipLaunchKernelGGL(attn_fold_list_kernel<FMT_F16>, g3, block, 0, stream, p);
    else hipLaunchKernelGGL(attn_fold_list_kernel<FMT_BF16>, g3, block, 0, stream, p);
    return 0;
}
