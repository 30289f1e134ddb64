// Post-processing of the part-feature maps on the GPU: the step behind the forward path in the reference's demo
// (demo.py:365-400 -> iggt/utils/misc.py): exact k-nearest-neighbour feature averaging over the predicted 3-D points
// (knn_avg_features_pyg, misc.py:24-78), the PCA colour map (apply_pca_colormap, misc.py:272-331) and the nearest-label fill
// of the clustering step (cluster_features_to_masks_mv, misc.py:130-144).
//
// kNN.  The reference calls torch_cluster's knn_graph on all S*H*W points as ONE batch (batch index all zero,
// misc.py:61-65): a brute-force O(M^2) search -- 1.8e12 point pairs for 8 views at 504 x 336.  Here the points are sorted
// along a Morton curve (codes from a robustly scaled 1024^3 grid; torch.sort does the radix sort) and cut into tiles of 256
// consecutive points with an axis-aligned box each.  One workgroup owns one tile of QUERIES (one query per thread, its k
// best candidates in registers) and visits candidate tiles nearest-first in curve order; a candidate tile is skipped when
// its box is farther from the query tile's box than the largest current k-th distance in the tile, and a wave skips a
// staged tile when none of its 64 queries can improve.  Boxes only prune -- every point that could be among the k nearest
// is tested with the exact squared distance, so the result is the exact kNN set (ties at the k-th distance aside, which
// torch_cluster does not order either) whatever the point distribution; a degenerate cloud just prunes less.  For surface-
// like point maps a query tests a few thousand candidates instead of M.  HBM traffic is the sorted points (16 B each, read
// from L2 by the neighbouring tiles) + k indices per point; the kernel is VALU-bound on the distance tests.
#include "common.h"
#include "../../include/iggt_hip.h"

#include <math.h>

namespace {

constexpr int TILE = 256;

// ---- Morton codes ---------------------------------------------------------------------------------------------------------
IGGT_DEVINL uint32_t spread10(uint32_t v) {   // 10 bits -> every third bit
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void morton_kernel(const float* pts, long M, float cx, float cy, float cz, float inv_cell,
                                                     int* codes) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float x = (pts[3 * i] - cx) * inv_cell + 512.f, y = (pts[3 * i + 1] - cy) * inv_cell + 512.f,
                z = (pts[3 * i + 2] - cz) * inv_cell + 512.f;
    // fmed3 maps NaN to a bound; a non-finite point only sorts badly, it never breaks the search
    const uint32_t ix = (uint32_t)__builtin_amdgcn_fmed3f(x, 0.f, 1023.f), iy = (uint32_t)__builtin_amdgcn_fmed3f(y, 0.f, 1023.f),
                   iz = (uint32_t)__builtin_amdgcn_fmed3f(z, 0.f, 1023.f);
    codes[i] = (int)(spread10(ix) | (spread10(iy) << 1) | (spread10(iz) << 2));   // 30 bits: non-negative as int32
}

// ---- sorted copy + tile boxes ------------------------------------------------------------------------------------------------
// sp[t*256 + j] = (x, y, z, bits(original index)) of the j-th point of tile t in curve order; slots past M hold +inf
// coordinates (their squared distance to anything is inf or NaN: never accepted).  box[t] = (min xyz, max xyz) of the
// finite points of the tile.
__global__ __launch_bounds__(256) void gather_tiles_kernel(const float* pts, const long* order, long M, float4* sp,
                                                           float* boxes) {
    __shared__ float red[6][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long i = (long)blockIdx.x * TILE + tid;
    float4 v = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(-1));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < M) {
        const long src = order[i];
        v = make_float4(pts[3 * src], pts[3 * src + 1], pts[3 * src + 2], __int_as_float((int)src));
        const float c[3] = {v.x, v.y, v.z};
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (fabsf(c[a]) <= 3.0e38f) lo[a] = hi[a] = c[a];
    }
    sp[i] = v;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
        if (lane == 0) {
            red[a][wave] = lo[a];
            red[3 + a][wave] = hi[a];
        }
    }
    __syncthreads();
    if (tid < 6) {
        float r = red[tid][0];
        for (int w = 1; w < 4; ++w) r = tid < 3 ? fminf(r, red[tid][w]) : fmaxf(r, red[tid][w]);
        boxes[(long)blockIdx.x * 6 + tid] = r;
    }
}

// ---- search -------------------------------------------------------------------------------------------------------------------
IGGT_DEVINL float gap(float qlo, float qhi, float tlo, float thi) { return fmaxf(0.f, fmaxf(tlo - qhi, qlo - thi)); }

template <int KMAX>
__global__ __launch_bounds__(256) void knn_search_kernel(const float4* __restrict__ sp, const float* __restrict__ boxes,
                                                         long M, int ntiles, int k, int* __restrict__ idx_out,
                                                         float* __restrict__ d2_out) {
    __shared__ float4 cand[TILE];
    __shared__ int list[TILE];
    __shared__ int wcnt[4];
    __shared__ float wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const long qi = (long)tile * TILE + tid;
    const float4 q = sp[qi];
    const int self = __float_as_int(q.w);
    const bool live = qi < M;

    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        bd[s] = (s < k && live) ? INFINITY : -1.f;
        bi[s] = -1;
    }
    float thr = live ? INFINITY : -1.f;   // the largest of the k kept squared distances: a candidate must beat it

    // the six numbers of this tile's box (queries), wave-uniform
    const float qb0 = boxes[(long)tile * 6 + 0], qb1 = boxes[(long)tile * 6 + 1], qb2 = boxes[(long)tile * 6 + 2],
                qb3 = boxes[(long)tile * 6 + 3], qb4 = boxes[(long)tile * 6 + 4], qb5 = boxes[(long)tile * 6 + 5];

    auto scan = [&](int t) {
        __syncthreads();                     // the previous tile's readers are done
        cand[tid] = sp[(long)t * TILE + tid];
        __syncthreads();
        // wave-level skip: squared distance from each query to the candidate tile's box
        const float* tb = boxes + (long)t * 6;
        const float gx = gap(q.x, q.x, tb[0], tb[3]), gy = gap(q.y, q.y, tb[1], tb[4]), gz = gap(q.z, q.z, tb[2], tb[5]);
        const float bd2 = (gx * gx + gy * gy + gz * gz) * 0.99999f;
        if (!__any(bd2 < thr || (thr == INFINITY))) return;
#pragma unroll 4
        for (int j = 0; j < TILE; ++j) {
            const float4 c = cand[j];
            const float dx = c.x - q.x, dy = c.y - q.y, dz = c.z - q.z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            const int ci = __float_as_int(c.w);
            if (d2 < thr && ci != self) {
                // replace the slot that holds the current worst, then find the new worst
                bool done = false;
                float m = -1.f;
#pragma unroll
                for (int s = 0; s < KMAX; ++s) {
                    const bool hit = !done && bd[s] == thr;
                    bd[s] = hit ? d2 : bd[s];
                    bi[s] = hit ? ci : bi[s];
                    done = done || hit;
                    m = fmaxf(m, bd[s]);
                }
                thr = m;
            }
        }
    };

    // phase A: the own tile and its neighbours along the curve give every query k candidates and a first bound
    for (int o = 0; o <= 2; ++o) {
        if (o == 0) {
            scan(tile);
        } else {
            if (tile + o < ntiles) scan(tile + o);
            if (tile - o >= 0) scan(tile - o);
        }
    }

    // phase B: every other tile whose box is nearer to this tile's box than the largest bound in the tile
    for (int base = 0; base < ntiles; base += TILE) {
        float m = thr;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        __syncthreads();
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        const float rmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        const int t = base + tid;
        bool need = false;
        if (t < ntiles && (t > tile + 2 || t < tile - 2)) {
            const float* tb = boxes + (long)t * 6;
            const float gx = gap(qb0, qb3, tb[0], tb[3]), gy = gap(qb1, qb4, tb[1], tb[4]), gz = gap(qb2, qb5, tb[2], tb[5]);
            need = (gx * gx + gy * gy + gz * gz) * 0.99999f < rmax;   // also true while rmax is still inf
        }
        const unsigned long long b = __ballot(need);
        if (lane == 0) wcnt[wave] = __popcll(b);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            off += w < wave ? wcnt[w] : 0;
            total += wcnt[w];
        }
        if (need) list[off + __popcll(b & ((1ull << lane) - 1ull))] = t;
        __syncthreads();
        for (int n = 0; n < total; ++n) scan(list[n]);
    }

    if (!live) return;
    // ascending (distance, index); unused slots (fewer than k other points) last with index -1
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
        if (bi[s] < 0) bd[s] = INFINITY;
#pragma unroll
    for (int i = 0; i < KMAX - 1; ++i) {
#pragma unroll
        for (int j = 0; j < KMAX - 1 - i; ++j) {
            const bool sw = bd[j] > bd[j + 1] || (bd[j] == bd[j + 1] && (unsigned)bi[j] > (unsigned)bi[j + 1]);
            const float td = sw ? bd[j + 1] : bd[j], tu = sw ? bd[j] : bd[j + 1];
            const int ti = sw ? bi[j + 1] : bi[j], tv = sw ? bi[j] : bi[j + 1];
            bd[j] = td; bd[j + 1] = tu; bi[j] = ti; bi[j + 1] = tv;
        }
    }
    int* io = idx_out + (long)self * k;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
        if (s < k) io[s] = bi[s];
    if (d2_out) {
        float* dd = d2_out + (long)self * k;
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
            if (s < k) dd[s] = bi[s] < 0 ? INFINITY : bd[s];
    }
}

// ---- neighbour mean (scatter_mean over the kNN edges, misc.py:69-71) -------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_mean_kernel(const float* __restrict__ feat, const int* __restrict__ idx, long M,
                                                       int k, int F, float* __restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * F) return;
    const long p = e / F;
    const int f = (int)(e - p * F);
    const int* row = idx + p * k;
    float s = 0.f;
    int n = 0;
    for (int j = 0; j < k; ++j) {
        const int nb = row[j];
        if (nb >= 0) {
            s += feat[(long)nb * F + f];
            ++n;
        }
    }
    out[e] = n ? s / (float)n : 0.f;
}

// ---- PCA colour map ----------------------------------------------------------------------------------------------------------------
// Per-block partial sums of (x - shift) and of its outer product (upper triangle), C <= CT.  part[b] = CT + CT*(CT+1)/2 floats.
template <int CT>
__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ x, long M, int C, const float* __restrict__ shift,
                                                      float* __restrict__ part) {
    constexpr int NT = CT * (CT + 1) / 2, NV = CT + NT;
    __shared__ float red[4][NV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float sh[CT], acc[NV];
#pragma unroll
    for (int c = 0; c < CT; ++c) sh[c] = c < C ? shift[c] : 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    for (long r = (long)blockIdx.x * 256 + tid; r < M; r += (long)gridDim.x * 256) {
        float v[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) v[c] = c < C ? x[r * C + c] - sh[c] : 0.f;
        int o = CT;
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            acc[a] += v[a];
#pragma unroll
            for (int b = a; b < CT; ++b) acc[o++] += v[a] * v[b];
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    for (int i = tid; i < NV; i += 256) part[(long)blockIdx.x * NV + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// out[r][j] = sum_c x[r][c] * v[c][j], j < 3 (the reference projects the UNCENTRED features, misc.py:299)
__global__ __launch_bounds__(256) void project3_kernel(const float* __restrict__ x, long M, int C, const float* __restrict__ v,
                                                       float* __restrict__ out) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const float xv = x[r * C + c];
        a0 += xv * v[3 * c];
        a1 += xv * v[3 * c + 1];
        a2 += xv * v[3 * c + 2];
    }
    out[3 * r] = a0;
    out[3 * r + 1] = a1;
    out[3 * r + 2] = a2;
}

// percentile stretch + clamp in place: channel j -> clamp((v - lo_j) / (hi_j - lo_j), 0, 1), or 0.5 when hi_j <= lo_j
__global__ __launch_bounds__(256) void stretch3_kernel(float* __restrict__ img, long M, const float* __restrict__ lohi) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= 3 * M) return;
    const int j = (int)(e % 3);
    const float lo = lohi[j], hi = lohi[3 + j];
    float v = 0.5f;
    if (hi > lo) v = (img[e] - lo) / (hi - lo);
    img[e] = fminf(fmaxf(v, 0.f), 1.f);
}

// ---- nearest labelled sample in feature space (NearestNeighbors(n_neighbors=1), misc.py:137-141) ----------------------------------
template <int CT>
__global__ __launch_bounds__(256) void nn1_label_kernel(const float* __restrict__ qf, long Mq, const float* __restrict__ rf,
                                                        long Mr, int C, const int* __restrict__ rlabel,
                                                        int* __restrict__ out) {
    __shared__ float cand[TILE][CT + 1];
    const int tid = threadIdx.x;
    const long qi = (long)blockIdx.x * TILE + tid;
    float q[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) q[c] = (qi < Mq && c < C) ? qf[qi * C + c] : 0.f;
    float best = INFINITY;
    long besti = -1;
    for (long base = 0; base < Mr; base += TILE) {
        __syncthreads();
        const long r = base + tid;
#pragma unroll
        for (int c = 0; c < CT; ++c) cand[tid][c] = (r < Mr && c < C) ? rf[r * C + c] : INFINITY;
        __syncthreads();
        const int n = (int)((Mr - base) < TILE ? (Mr - base) : TILE);
        for (int j = 0; j < n; ++j) {
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float d = (c < C) ? cand[j][c] - q[c] : 0.f;
                d2 += d * d;
            }
            if (d2 < best) {
                best = d2;
                besti = base + j;
            }
        }
    }
    if (qi < Mq) out[qi] = besti >= 0 ? rlabel[besti] : -1;
}

// The brute-force search with the samples split over gridDim.y workgroups per query tile: part y scans the contiguous sample range
// [y * chunk, (y + 1) * chunk) and writes its own (distance, index) plane; the caller folds the planes (smallest distance, then
// smallest index = the first minimum).  For FEW queries against MANY samples (the usual noise fill: a few thousand noise pixels
// against 1.35 M labelled ones is 9 query tiles -- 9 workgroups walking 5 292 sample tiles each took 0.49 s).
template <int CT>
__global__ __launch_bounds__(256) void nn1_search_split_kernel(const float* __restrict__ qf, long Mq, const float* __restrict__ rf,
                                                               long Mr, int C, long chunk, float* __restrict__ best_d2,
                                                               int* __restrict__ best_i) {
    __shared__ float cand[TILE][CT + 1];
    const int tid = threadIdx.x;
    const long qi = (long)blockIdx.x * TILE + tid;
    float q[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) q[c] = (qi < Mq && c < C) ? qf[qi * C + c] : 0.f;
    float best = INFINITY;
    long besti = -1;
    const long r_begin = (long)blockIdx.y * chunk;
    const long r_end = (r_begin + chunk) < Mr ? (r_begin + chunk) : Mr;
    for (long base = r_begin; base < r_end; base += TILE) {
        __syncthreads();
        const long r = base + tid;
#pragma unroll
        for (int c = 0; c < CT; ++c) cand[tid][c] = (r < r_end && c < C) ? rf[r * C + c] : INFINITY;
        __syncthreads();
        const int n = (int)((r_end - base) < TILE ? (r_end - base) : TILE);
        for (int j = 0; j < n; ++j) {
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float d = (c < C) ? cand[j][c] - q[c] : 0.f;
                d2 += d * d;
            }
            if (d2 < best) {
                best = d2;
                besti = base + j;
            }
        }
    }
    if (qi < Mq) {
        best_d2[(long)blockIdx.y * Mq + qi] = best;
        best_i[(long)blockIdx.y * Mq + qi] = (int)besti;
    }
}

// The same search made local (round 3): queries and labelled samples arrive sorted along one space-filling curve, every tile of
// 256 consecutive rows with its bounding box.  A workgroup (one query tile) starts at the sample tile whose box is nearest to its
// own, walks outward in curve order and skips -- after a block-wide vote -- every tile whose box is farther from the query box
// than each query's best distance so far (the box gap is a lower bound of every pair distance: exact).  "First minimum" of the
// brute-force kernel = smallest ORIGINAL sample index among equal distances: ridx carries that index, ties are broken on it.
// At the demo's size (1.35 M pixels, ~10 % noise) the brute-force pass is 1.6e11 pairs; a surface-like cloud leaves a few
// thousand per query.
template <int CT>
__global__ __launch_bounds__(256) void nn1_label_tiled_kernel(const float* __restrict__ qf, long Mq, const float* __restrict__ qblo,
                                                              const float* __restrict__ qbhi, const float* __restrict__ rf, long Mr,
                                                              const float* __restrict__ rblo, const float* __restrict__ rbhi, int C,
                                                              const int* __restrict__ ridx, const int* __restrict__ rlabel,
                                                              int* __restrict__ out) {
    __shared__ float cand[TILE][CT + 1];
    __shared__ int cidx[TILE];
    __shared__ float sgap[TILE];
    __shared__ int sarg[TILE];
    const int tid = threadIdx.x;
    const long qi = (long)blockIdx.x * TILE + tid;
    const int nrt = (int)((Mr + TILE - 1) / TILE);
    float q[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) q[c] = (qi < Mq && c < C) ? qf[qi * C + c] : 0.f;
    const float* mylo = qblo + (long)blockIdx.x * C;
    const float* myhi = qbhi + (long)blockIdx.x * C;
    auto gap2_of = [&](int tl) {
        float g2 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float g = fmaxf(fmaxf(mylo[c] - rbhi[(long)tl * C + c], rblo[(long)tl * C + c] - myhi[c]), 0.f);
            g2 = fmaf(g, g, g2);
        }
        return g2;
    };
    // seed: the sample tile nearest to the query box (smallest index among equals)
    {
        float g = INFINITY;
        int arg = 0x7fffffff;
        for (int tl = tid; tl < nrt; tl += TILE) {
            const float gg = gap2_of(tl);
            if (gg < g) {
                g = gg;
                arg = tl;
            }
        }
        sgap[tid] = g;
        sarg[tid] = arg;
        __syncthreads();
        for (int o = TILE / 2; o > 0; o >>= 1) {
            if (tid < o) {
                const float g2 = sgap[tid + o];
                const int a2 = sarg[tid + o];
                if (g2 < sgap[tid] || (g2 == sgap[tid] && a2 < sarg[tid])) {
                    sgap[tid] = g2;
                    sarg[tid] = a2;
                }
            }
            __syncthreads();
        }
    }
    int seed = sarg[0];
    seed = (seed < 0 || seed >= nrt) ? 0 : seed;
    float best = qi < Mq ? INFINITY : -1.f;   // dead lanes never ask for a tile
    int bestpos = -1, besti = 0x7fffffff;
    for (int step = 0; step < 2 * nrt; ++step) {
        const int m = (step + 1) >> 1;
        const int tl = (step & 1) ? seed + m : seed - m;
        if (tl < 0 || tl >= nrt) continue;
        const float gap2 = gap2_of(tl);
        if (!__syncthreads_or(gap2 <= best)) continue;      // (also the barrier that protects the previous tile's readers)
        const long r0 = (long)tl * TILE;
        {
            const long r = r0 + tid;
#pragma unroll
            for (int c = 0; c < CT; ++c) cand[tid][c] = c < C ? (r < Mr ? rf[r * C + c] : INFINITY) : 0.f;
            cidx[tid] = r < Mr ? ridx[r] : 0x7fffffff;
        }
        __syncthreads();
        const int n = (int)((Mr - r0) < TILE ? (Mr - r0) : TILE);
        for (int j = 0; j < n; ++j) {
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float d = cand[j][c] - q[c];
                d2 += d * d;
            }
            const int oi = cidx[j];
            if (d2 < best || (d2 == best && oi < besti)) {
                best = d2;
                bestpos = (int)(r0 + j);
                besti = oi;
            }
        }
    }
    if (qi < Mq) out[qi] = bestpos >= 0 ? rlabel[bestpos] : -1;
}

template <int KMAX>
int launch_search(const float4* sp, const float* boxes, long M, int ntiles, int k, int* idx, float* d2, hipStream_t st) {
    hipLaunchKernelGGL(knn_search_kernel<KMAX>, dim3((unsigned)ntiles), dim3(256), 0, st, sp, boxes, M, ntiles, k, idx, d2);
    IGGT_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int iggt_knn_morton_codes(const float* points, long M, float cx, float cy, float cz, float inv_cell, int* codes,
                                     void* stream) {
    if (M <= 0 || M > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(morton_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points, M,
                       cx, cy, cz, inv_cell, codes);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_knn_search(const float* points, const long* order, long M, int k, void* sorted_ws, float* boxes,
                               int* idx_out, float* d2_out, void* stream) {
    if (M <= 0 || M > 0x7fffffffL || k <= 0 || k > 32) return -1;
    const int ntiles = (int)((M + TILE - 1) / TILE);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gather_tiles_kernel, dim3((unsigned)ntiles), dim3(256), 0, st, points, order, M, (float4*)sorted_ws,
                       boxes);
    IGGT_CHECK_LAUNCH();
    const float4* sp = (const float4*)sorted_ws;
    if (k <= 8) return launch_search<8>(sp, boxes, M, ntiles, k, idx_out, d2_out, st);
    if (k <= 16) return launch_search<16>(sp, boxes, M, ntiles, k, idx_out, d2_out, st);
    if (k <= 24) return launch_search<24>(sp, boxes, M, ntiles, k, idx_out, d2_out, st);
    return launch_search<32>(sp, boxes, M, ntiles, k, idx_out, d2_out, st);
}

extern "C" int iggt_knn_mean_features_f32(const float* feat, const int* idx, long M, int k, int F, float* out, void* stream) {
    if (M <= 0 || k <= 0 || F <= 0) return -1;
    const long n = M * F;
    hipLaunchKernelGGL(knn_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat, idx, M, k,
                       F, out);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_moments_f32(const float* x, long M, int C, const float* shift, float* partials, int nblocks,
                                void* stream) {
    if (M <= 0 || C <= 0 || nblocks <= 0) return -1;
    if (C > 16) return -3;
    hipStream_t st = (hipStream_t)stream;
    if (C <= 4)
        hipLaunchKernelGGL(moments_kernel<4>, dim3(nblocks), dim3(256), 0, st, x, M, C, shift, partials);
    else if (C <= 8)
        hipLaunchKernelGGL(moments_kernel<8>, dim3(nblocks), dim3(256), 0, st, x, M, C, shift, partials);
    else
        hipLaunchKernelGGL(moments_kernel<16>, dim3(nblocks), dim3(256), 0, st, x, M, C, shift, partials);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_moments_width(int C) {
    const int ct = C <= 4 ? 4 : (C <= 8 ? 8 : 16);
    return C > 16 || C <= 0 ? -3 : ct + ct * (ct + 1) / 2;
}

extern "C" int iggt_project3_f32(const float* x, long M, int C, const float* v, float* out, void* stream) {
    if (M <= 0 || C <= 0) return -1;
    hipLaunchKernelGGL(project3_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, M, C, v, out);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_stretch3_f32(float* img, long M, const float* lohi, void* stream) {
    if (M <= 0) return -1;
    hipLaunchKernelGGL(stretch3_kernel, dim3((unsigned)((3 * M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, M, lohi);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_nn1_label_f32(const float* query, long Mq, const float* ref, long Mr, int C, const int* ref_labels,
                                  int* out, void* stream) {
    if (Mq <= 0 || Mr <= 0 || C <= 0) return -1;
    if (C > 16) return -3;
    hipStream_t st = (hipStream_t)stream;
    const unsigned g = (unsigned)((Mq + TILE - 1) / TILE);
    if (C <= 4)
        hipLaunchKernelGGL(nn1_label_kernel<4>, dim3(g), dim3(256), 0, st, query, Mq, ref, Mr, C, ref_labels, out);
    else if (C <= 8)
        hipLaunchKernelGGL(nn1_label_kernel<8>, dim3(g), dim3(256), 0, st, query, Mq, ref, Mr, C, ref_labels, out);
    else
        hipLaunchKernelGGL(nn1_label_kernel<16>, dim3(g), dim3(256), 0, st, query, Mq, ref, Mr, C, ref_labels, out);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_nn1_label_tiled_f32(const float* query, long Mq, const float* qbox_lo, const float* qbox_hi, const float* ref,
                                        long Mr, const float* rbox_lo, const float* rbox_hi, int C, const int* ref_idx,
                                        const int* ref_labels, int* out, void* stream) {
    if (query == nullptr || ref == nullptr || qbox_lo == nullptr || qbox_hi == nullptr || rbox_lo == nullptr || rbox_hi == nullptr ||
        ref_idx == nullptr || ref_labels == nullptr || out == nullptr)
        return -1;
    if (Mq <= 0 || Mr <= 0 || Mr >= (1L << 31) || C <= 0) return -1;
    if (C > 16) return -3;
    hipStream_t st = (hipStream_t)stream;
    const unsigned g = (unsigned)((Mq + TILE - 1) / TILE);
    if (C <= 4)
        hipLaunchKernelGGL(nn1_label_tiled_kernel<4>, dim3(g), dim3(256), 0, st, query, Mq, qbox_lo, qbox_hi, ref, Mr, rbox_lo, rbox_hi, C,
                           ref_idx, ref_labels, out);
    else if (C <= 8)
        hipLaunchKernelGGL(nn1_label_tiled_kernel<8>, dim3(g), dim3(256), 0, st, query, Mq, qbox_lo, qbox_hi, ref, Mr, rbox_lo, rbox_hi, C,
                           ref_idx, ref_labels, out);
    else
        hipLaunchKernelGGL(nn1_label_tiled_kernel<16>, dim3(g), dim3(256), 0, st, query, Mq, qbox_lo, qbox_hi, ref, Mr, rbox_lo, rbox_hi, C,
                           ref_idx, ref_labels, out);
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_nn1_search_split_f32(const float* query, long Mq, const float* ref, long Mr, int C, int nsplit, float* best_d2,
                                         int* best_idx, void* stream) {
    if (query == nullptr || ref == nullptr || best_d2 == nullptr || best_idx == nullptr) return -1;
    if (Mq <= 0 || Mr <= 0 || Mr >= (1L << 31) || C <= 0 || nsplit < 1 || nsplit > 65535) return -1;
    if (C > 16) return -3;
    hipStream_t st = (hipStream_t)stream;
    long chunk = (Mr + nsplit - 1) / nsplit;
    chunk = (chunk + TILE - 1) / TILE * TILE;          // whole tiles: a part's range starts on a tile boundary
    const dim3 g((unsigned)((Mq + TILE - 1) / TILE), (unsigned)nsplit);
    if (C <= 4) hipLaunchKernelGGL(nn1_search_split_kernel<4>, g, dim3(256), 0, st, query, Mq, ref, Mr, C, chunk, best_d2, best_idx);
    else if (C <= 8) hipLaunchKernelGGL(nn1_search_split_kernel<8>, g, dim3(256), 0, st, query, Mq, ref, Mr, C, chunk, best_d2, best_idx);
    else hipLaunchKernelGGL(nn1_search_split_kernel<16>, g, dim3(256), 0, st, query, Mq, ref, Mr, C, chunk, best_d2, best_idx);
    IGGT_CHECK_LAUNCH();
    return 0;
}
