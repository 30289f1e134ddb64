// HDBSCAN, device half (reference iggt/utils/misc.py:123-129 -- the clustering step of demo.py:385-394 over all N * H * W pixels'
// 8-channel part features): the two O(M^2) pieces of the algorithm as exact fp32 brute-force kernels.
//
//   core distances      core[i] = distance from point i to its k-th nearest point, itself counted (k = min_samples): what
//                       scikit-learn's / hdbscan's `kneighbors(X, min_samples)[:, -1]` returns.
//   nearest foreign     one Boruvka round of the minimum spanning tree of the MUTUAL-REACHABILITY graph
//                       mr(i, j) = max(core[i], core[j], |x_i - x_j|):  for every point the cheapest edge to a point of ANOTHER
//                       component, ties broken by the (smaller index, larger index) pair so that edges are totally ordered and the
//                       rounds cannot close a cycle.  The per-component minimum, the hooking and the pointer jumping are a few
//                       torch index operations per round (iggt_official_amd/utils/hdbscan.py); the M - 1 edges then go to the host
//                       walk in csrc/hdbscan_tree.hip.
// Both kernels give one thread a query point (coordinates in registers) and stream all points past it in tiles of 256 staged in
// LDS, channel-major, so that every lane reads the same candidate (a broadcast, no bank conflict).  Bound: vector ALU --
// 3 C + 6 operations per pair, no reuse to exploit beyond the LDS tile; M = 1.35 M points (8 views x 504 x 336) is 1.8e12 pairs.
//   * core distances keep the k smallest squared distances of a query in LDS, slot-major ([k][256]: lane t owns column t), with
//     the current maximum and its slot in registers; a candidate below the maximum replaces it and the column is rescanned
//     (k reads).  Replacements become rare quickly (~k ln(M / k) per query), the scan over candidates dominates.
//   * nearest foreign: points arrive SORTED BY COMPONENT; a tile whose 256 points all belong to the component of every query of
//     the workgroup is skipped, so that once a giant component has formed a round costs ~2 |giant| |rest| pairs instead of M^2.
#include <math.h>

#include "common.h"
#include "../../include/iggt_hip.h"

namespace {

constexpr int TILE = 256;
constexpr int KMAX = 128;

template <int C>
__global__ __launch_bounds__(256) void hdb_core_dist_kernel(const float* __restrict__ x, long M, int k, float* __restrict__ core) {
    extern __shared__ float smem[];
    float* tile = smem;                 // [C][TILE]
    float* best = smem + C * TILE;      // [k][TILE]
    const int t = threadIdx.x;
    const long i = (long)blockIdx.x * TILE + t;
    float q[C];
#pragma unroll
    for (int c = 0; c < C; ++c) q[c] = i < M ? x[i * C + c] : 0.f;
    for (int s = 0; s < k; ++s) best[s * TILE + t] = INFINITY;
    float cur_max = INFINITY;
    int cur_slot = 0;
    for (long j0 = 0; j0 < M; j0 += TILE) {
        __syncthreads();
        {
            const long j = j0 + t;
#pragma unroll
            for (int c = 0; c < C; ++c) tile[c * TILE + t] = j < M ? x[j * C + c] : INFINITY;   // padding: distance inf
        }
        __syncthreads();
        const int nj = (int)((M - j0) < TILE ? (M - j0) : TILE);
        for (int jj = 0; jj < nj; ++jj) {
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d = q[c] - tile[c * TILE + jj];
                d2 = fmaf(d, d, d2);
            }
            if (d2 < cur_max) {          // replace the current maximum, find the new one
                best[cur_slot * TILE + t] = d2;
                float m = -1.f;
                int ms = 0;
                for (int s = 0; s < k; ++s) {
                    const float v = best[s * TILE + t];
                    if (v > m) {
                        m = v;
                        ms = s;
                    }
                }
                cur_max = m;
                cur_slot = ms;
            }
        }
    }
    if (i < M) core[i] = sqrtf(cur_max);   // inf when M < k: fewer than k points exist
}

// One Boruvka round.  x, core2 (= core^2), comp, idx are given in an order sorted by component; idx[p] = original index of the
// point at position p (ties are broken on ORIGINAL indices).  tile_lo / tile_hi: smallest / largest component id inside each tile
// of 256 positions.  Output per position: best_w2 (squared mutual reachability, inf if none), best_p (position of the partner, -1).
template <int C>
__global__ __launch_bounds__(256) void hdb_nearest_foreign_kernel(const float* __restrict__ x, const float* __restrict__ core2,
                                                                  const int* __restrict__ comp, const int* __restrict__ idx,
                                                                  const int* __restrict__ tile_lo, const int* __restrict__ tile_hi,
                                                                  long M, float* __restrict__ best_w2, int* __restrict__ best_p) {
    __shared__ float tile[C][TILE];
    __shared__ float tcore[TILE];
    __shared__ int tcomp[TILE], tidx[TILE];
    const int t = threadIdx.x;
    const long i = (long)blockIdx.x * TILE + t;
    const bool live = i < M;
    float q[C];
#pragma unroll
    for (int c = 0; c < C; ++c) q[c] = live ? x[i * C + c] : 0.f;
    const float qc2 = live ? core2[i] : 0.f;
    const int qcomp = live ? comp[i] : -1, qidx = live ? idx[i] : 0;
    // all queries of this workgroup in one component?  (positions are sorted by component)
    const int wg_lo = tile_lo[blockIdx.x], wg_hi = tile_hi[blockIdx.x];
    float bw = INFINITY;
    int bp = -1, blo = 0x7fffffff, bhi = 0x7fffffff;
    const int ntiles = (int)((M + TILE - 1) / TILE);
    for (int tl = 0; tl < ntiles; ++tl) {
        // skip a tile that lies entirely inside the single component all queries here belong to (block-uniform test)
        if (wg_lo == wg_hi && tile_lo[tl] == wg_lo && tile_hi[tl] == wg_lo) continue;
        const long j0 = (long)tl * TILE;
        __syncthreads();
        {
            const long j = j0 + t;
            const bool ok = j < M;
#pragma unroll
            for (int c = 0; c < C; ++c) tile[c][t] = ok ? x[j * C + c] : 0.f;
            tcore[t] = ok ? core2[j] : INFINITY;
            tcomp[t] = ok ? comp[j] : -1;
            tidx[t] = ok ? idx[j] : 0;
        }
        __syncthreads();
        if (!live) continue;
        const int nj = (int)((M - j0) < TILE ? (M - j0) : TILE);
        for (int jj = 0; jj < nj; ++jj) {
            if (tcomp[jj] == qcomp) continue;
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d = q[c] - tile[c][jj];
                d2 = fmaf(d, d, d2);
            }
            const float w = fmaxf(fmaxf(qc2, tcore[jj]), d2);
            if (w > bw) continue;
            const int oj = tidx[jj];
            const int lo = qidx < oj ? qidx : oj, hi = qidx < oj ? oj : qidx;
            if (w < bw || lo < blo || (lo == blo && hi < bhi)) {
                bw = w;
                bp = (int)(j0 + jj);
                blo = lo;
                bhi = hi;
            }
        }
    }
    if (live) {
        best_w2[i] = bw;
        best_p[i] = bp;
    }
}

template <int C>
int launch_core(const float* x, long M, int k, float* core, hipStream_t stream) {
    const size_t lds = (size_t)(C + k) * TILE * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)hdb_core_dist_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)((C + KMAX) * TILE * sizeof(float)));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(hdb_core_dist_kernel<C>, dim3((unsigned)((M + TILE - 1) / TILE)), dim3(TILE), lds, stream, x, M, k, core);
    return 0;
}

}  // namespace

extern "C" int iggt_hdbscan_core_dist_f32(const float* x, long M, int C, int k, float* core, void* stream) {
    if (x == nullptr || core == nullptr || M <= 0 || k < 1 || k > KMAX) return -1;
    int rc;
    if (C == 8) rc = launch_core<8>(x, M, k, core, (hipStream_t)stream);
    else if (C == 3) rc = launch_core<3>(x, M, k, core, (hipStream_t)stream);
    else if (C == 16) rc = launch_core<16>(x, M, k, core, (hipStream_t)stream);
    else return -2;
    if (rc) return rc;
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_hdbscan_nearest_foreign_f32(const float* x, const float* core2, const int* comp, const int* idx,
                                                const int* tile_lo, const int* tile_hi, long M, int C, float* best_w2,
                                                int* best_p, void* stream) {
    if (x == nullptr || core2 == nullptr || comp == nullptr || idx == nullptr || tile_lo == nullptr || tile_hi == nullptr ||
        best_w2 == nullptr || best_p == nullptr || M <= 0 || M >= (1L << 31))
        return -1;
    const dim3 grid((unsigned)((M + TILE - 1) / TILE)), block(TILE);
    hipStream_t st = (hipStream_t)stream;
    if (C == 8) hipLaunchKernelGGL(hdb_nearest_foreign_kernel<8>, grid, block, 0, st, x, core2, comp, idx, tile_lo, tile_hi, M, best_w2, best_p);
    else if (C == 3) hipLaunchKernelGGL(hdb_nearest_foreign_kernel<3>, grid, block, 0, st, x, core2, comp, idx, tile_lo, tile_hi, M, best_w2, best_p);
    else if (C == 16) hipLaunchKernelGGL(hdb_nearest_foreign_kernel<16>, grid, block, 0, st, x, core2, comp, idx, tile_lo, tile_hi, M, best_w2, best_p);
    else return -2;
    IGGT_CHECK_LAUNCH();
    return 0;
}
