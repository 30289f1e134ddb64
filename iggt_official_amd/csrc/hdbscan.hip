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
// Both kernels give a thread its query point(s) (coordinates in registers) and stream all points past them in tiles of 256 staged
// in LDS, candidate-major, so that every lane reads the same candidate with one or two 16-byte broadcast reads (a first version
// read it channel by channel: 8 LDS instructions per pair beside 8 FMAs).  Bound: vector ALU -- 3 C + 6 operations per pair, no
// reuse to exploit beyond the LDS tile; M = 1.35 M points (8 views x 504 x 336) is 1.8e12 pairs per pass.
//   * core distances keep the k smallest squared distances of a query in LDS as a max-heap, node-major ([k][QT]: lane t owns
//     column t; QT = 256 queries per workgroup, 128 for k > 48); a candidate below the root replaces it and sifts down (log2 k
//     steps).  (First version: unsorted slots, rescanned on every replacement -- 5.9 s of the 6.5 s the pass took at k = 100 and
//     1.35 M points, against 0.6 s at k = 5.)
//   * nearest foreign: points arrive SORTED BY COMPONENT; a tile whose 256 points all belong to the component of every query of
//     the workgroup is skipped, so that once a giant component has formed a round costs ~2 |giant| |rest| pairs instead of M^2;
//     workgroups inside one component share the component's best weight so far (atomic minimum) and drop what is strictly worse;
//     every block of 512 queries is searched by up to 32 workgroups over interleaved tiles (load balance: one outlier query used to
//     drag its workgroup through nearly every tile while the chip idled).
//   * both: the points are ordered along a Morton curve over their first three principal axes and every 256-point tile carries its
//     bounding box; tiles are visited outward from the query tile and one whose box lies farther away than every query's current
//     bound is skipped after a block-wide vote -- exact (the box gap is a lower bound of every pair distance), and what turns
//     the M^2 passes into a neighbourhood search on data that is not uniformly spread in all C dimensions.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "../../include/iggt_hip.h"

namespace {

constexpr int TILE = 256;
constexpr int KMAX = 128;

// C floats of one staged candidate -> registers (all lanes read the same address: LDS broadcast)
template <int C>
IGGT_DEVINL void load_point(const float* src, float (&p)[C]) {
    if constexpr (C % 4 == 0) {
#pragma unroll
        for (int v = 0; v < C / 4; ++v) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(src + 4 * v);
            p[4 * v] = w[0]; p[4 * v + 1] = w[1]; p[4 * v + 2] = w[2]; p[4 * v + 3] = w[3];
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) p[c] = src[c];
    }
}

// squared distance between two axis-aligned boxes (lower bound of every point-pair distance across them); all arguments are
// workgroup-uniform, so this is scalar work
template <int C>
IGGT_DEVINL float box_gap2(const float* __restrict__ alo, const float* __restrict__ ahi, const float* __restrict__ blo,
                           const float* __restrict__ bhi) {
    float g2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float g = fmaxf(fmaxf(alo[c] - bhi[c], blo[c] - ahi[c]), 0.f);
        g2 = fmaf(g, g, g2);
    }
    return g2;
}

// tiles in the order own, own + 1, own - 1, own + 2, ...: points are sorted along a space-filling curve, so the nearest tiles
// come first and tighten the bounds that let the far ones be skipped.  step in [0, 2 * ntiles); -1: outside
IGGT_DEVINL int outward_tile(int own, int step, int ntiles) {
    const int m = (step + 1) >> 1;
    const int tl = (step & 1) ? own + m : own - m;
    return (tl < 0 || tl >= ntiles) ? -1 : tl;
}

// Points are given SORTED along a Morton curve over their first three principal axes (iggt_official_amd/utils/hdbscan.py), with the
// bounding box of every 256-point tile (box_lo / box_hi [ntiles][C]).  A candidate tile whose box is farther from the query tile's
// box than every query's current k-th distance cannot change any result and is skipped (block-wide vote, exact).
// QT = queries (threads) per workgroup: 256, or 128 for k > 48 -- the heaps of 256 queries at k = 100 are 100 KB of LDS, ONE
// workgroup per CU, and the sift-down of a replacement is a chain of dependent LDS reads that nothing else on the SIMD hides; half
// the queries per workgroup fit three workgroups per CU.  The candidate tiles stay 256 points; the walk starts at the 256-point tile
// that contains the queries and uses its box (a superset of theirs: a valid, slightly looser gap).
template <int C, int QT>
__global__ __launch_bounds__(QT) void hdb_core_dist_kernel(const float* __restrict__ x, long M, int k,
                                                           const float* __restrict__ box_lo, const float* __restrict__ box_hi,
                                                           float* __restrict__ core) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                 // [TILE][C]: a candidate's C coordinates are contiguous (one or two 16-byte broadcast reads)
    float* heap = smem + C * TILE;      // [k][QT]: per-query max-heap of the k smallest squared distances, node-major
    const int t = threadIdx.x;
    const int own = (int)(((long)blockIdx.x * QT) / TILE);
    const int ntiles = (int)((M + TILE - 1) / TILE);
    const long i = (long)blockIdx.x * QT + t;
    float q[C];
#pragma unroll
    for (int c = 0; c < C; ++c) q[c] = i < M ? x[i * C + c] : 0.f;
    for (int s = 0; s < k; ++s) heap[s * QT + t] = INFINITY;
    float cur_max = i < M ? INFINITY : -1.f;     // dead lanes never ask for a tile
    for (int step = 0; step < 2 * ntiles; ++step) {
        const int tl = outward_tile(own, step, ntiles);
        if (tl < 0) continue;
        const float gap2 = box_gap2<C>(box_lo + (long)own * C, box_hi + (long)own * C, box_lo + (long)tl * C, box_hi + (long)tl * C);
        if (!__syncthreads_or(gap2 < cur_max)) continue;      // (also the barrier that protects the previous tile's readers)
        const long j0 = (long)tl * TILE;
        for (int sidx = t; sidx < TILE; sidx += QT) {
            const long j = j0 + sidx;
#pragma unroll
            for (int c = 0; c < C; ++c) tile[sidx * C + c] = j < M ? x[j * C + c] : INFINITY;   // padding: distance inf
        }
        __syncthreads();
        const int nj = (int)((M - j0) < TILE ? (M - j0) : TILE);
        for (int jj = 0; jj < nj; ++jj) {
            float p[C];
            load_point<C>(tile + jj * C, p);
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d = q[c] - p[c];
                d2 = fmaf(d, d, d2);
            }
            if (d2 < cur_max) {          // replace the root of the max-heap and sift down
                int n = 0;
                for (;;) {
                    const int l = 2 * n + 1;
                    if (l >= k) break;
                    const float lv = heap[l * QT + t];
                    const float rv = (l + 1 < k) ? heap[(l + 1) * QT + t] : -1.f;
                    const int cn = rv > lv ? l + 1 : l;
                    const float cv = fmaxf(lv, rv);
                    if (cv <= d2) break;
                    heap[n * QT + t] = cv;
                    n = cn;
                }
                heap[n * QT + t] = d2;
                cur_max = heap[t];
            }
        }
    }
    if (i < M) core[i] = sqrtf(cur_max);   // inf when M < k: fewer than k points exist
}

// One Boruvka round.  x, core2 (= core^2), comp, idx are given in an order sorted by component (and spatially inside a component);
// idx[p] = index of the point at position p in the caller's numbering (ties are broken on those).  tile_lo / tile_hi: smallest /
// largest component id inside each tile of 256 positions; box_lo / box_hi [ntiles][C]: bounding boxes of the tiles.
// Output per position: best_w2 (squared mutual reachability, inf if none), best_p (position of the partner, -1).
// Every thread owns QPT = 2 queries (positions base + t and base + 256 + t of a 512-position block): a staged candidate is read
// from LDS once for two pairs.  Skipped: tiles inside the one component all queries of the workgroup belong to, and tiles whose
// box is farther from the queries' box than every query's best weight so far (mr >= distance >= box gap; strict, so that an
// equal-weight candidate can still win the index tie-break).
constexpr int QPT = 2;

#ifdef IGGT_HDB_STATS   // developer builds (probes/build_alt.py hdb_stats): where a round's time goes
__device__ unsigned long long g_hdb_stats[8];
__device__ unsigned long long g_hdb_blk[131072][4];   // per workgroup: start, end (100 MHz wall clock), steps, tiles processed
#define HDB_COUNT(i, n) do { if (threadIdx.x == 0) hdb_local[i] += (n); } while (0)
#define HDB_BEGIN() unsigned long long hdb_local[4] = {0, 0, 0, 0}; const unsigned long long hdb_t0 = wall_clock64()
#define HDB_END() do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 4; ++i_) atomicAdd(&g_hdb_stats[i_], hdb_local[i_]); \
        if (blockIdx.x < 131072) { g_hdb_blk[blockIdx.x][0] = hdb_t0; g_hdb_blk[blockIdx.x][1] = wall_clock64(); \
                                 g_hdb_blk[blockIdx.x][2] = hdb_local[0]; g_hdb_blk[blockIdx.x][3] = hdb_local[2] + hdb_local[3]; } } } while (0)
#else
#define HDB_COUNT(i, n) do { } while (0)
#define HDB_BEGIN() do { } while (0)
#define HDB_END() do { } while (0)
#endif

template <int C>
__global__ __launch_bounds__(256) void hdb_nearest_foreign_kernel(const float* __restrict__ x, const float* __restrict__ core2,
                                                                  const int* __restrict__ comp, const int* __restrict__ idx,
                                                                  const int* __restrict__ tile_lo, const int* __restrict__ tile_hi,
                                                                  const float* __restrict__ box_lo, const float* __restrict__ box_hi,
                                                                  long M, float* __restrict__ best_w2, int* __restrict__ best_p,
                                                                  unsigned* __restrict__ comp_bound, int nsplit) {
    __shared__ __attribute__((aligned(16))) float tile[TILE * C];
    __shared__ float tcore[TILE];
    __shared__ int tcomp[TILE], tidx[TILE];
    __shared__ float red[8];
    const int t = threadIdx.x;
    const int ntiles = (int)((M + TILE - 1) / TILE);
    // nsplit workgroups share one block of 512 queries: workgroup `part` takes every nsplit-th step of the outward walk (part 0
    // the own tile, part 1 the next one, ...) and writes its own (best_w2, best_p) plane; the caller folds the planes under the
    // same total order.  Why: the block-wide vote makes ONE query with a large bound (an outlier: its core distance is the
    // floor of its bound) drag all 512 through nearly every tile -- a handful of such workgroups ran 200 - 300 ms while the
    // mean was 4 - 20 ms, and the round waited for them (profiles/r03_postprocess_timing.txt).
    const int qblock = blockIdx.x / nsplit, part = blockIdx.x - qblock * nsplit;
    float q[QPT][C], qc2[QPT], bw[QPT];
    int qcomp[QPT], qidx[QPT], bp[QPT], blo[QPT], bhi[QPT];
    bool live[QPT];
    int wg_lo = 0x7fffffff, wg_hi = -1;                 // component range of all queries of this workgroup
    float qlo[C], qhi[C];                               // their bounding box (workgroup-uniform)
#pragma unroll
    for (int c = 0; c < C; ++c) {
        qlo[c] = INFINITY;
        qhi[c] = -INFINITY;
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const long i = ((long)qblock * QPT + u) * TILE + t;
        live[u] = i < M;
#pragma unroll
        for (int c = 0; c < C; ++c) q[u][c] = live[u] ? x[i * C + c] : 0.f;
        qc2[u] = live[u] ? core2[i] : 0.f;
        qcomp[u] = live[u] ? comp[i] : -1;
        qidx[u] = live[u] ? idx[i] : 0;
        bw[u] = live[u] ? INFINITY : -1.f;
        bp[u] = -1;
        blo[u] = bhi[u] = 0x7fffffff;
        const int qt = qblock * QPT + u;
        if (qt < ntiles) {
            wg_lo = tile_lo[qt] < wg_lo ? tile_lo[qt] : wg_lo;
            wg_hi = tile_hi[qt] > wg_hi ? tile_hi[qt] : wg_hi;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                qlo[c] = fminf(qlo[c], box_lo[(long)qt * C + c]);
                qhi[c] = fmaxf(qhi[c], box_hi[(long)qt * C + c]);
            }
        }
    }
    const int own = qblock * QPT;
    // COMPONENT bound (Boruvka only needs the cheapest outgoing edge per component, not per point): comp_bound[c] holds, as float
    // bits, the smallest squared weight any workgroup has found so far for an edge out of component c (+inf at launch).  A
    // workgroup whose queries all belong to ONE component publishes its own minimum there after every tile it processes
    // (one atomicMin, non-negative floats order like their bit patterns) and reads the value back: candidates and tiles that are
    // strictly worse than it cannot be the component's minimum and are skipped -- strictly, so that an equal-weight edge with a
    // smaller index pair still wins the tie-break; the endpoint of the true minimum edge never prunes that edge (its weight is
    // <= every published value), so the per-component minimum the host takes over best_w2 is unchanged while interior points of
    // a large component stop after the first foreign tile instead of searching for a far partner nobody needs.  A stale read
    // only prunes less.  Queries whose own core distance exceeds the bound are dead (mr >= core); a workgroup of dead queries
    // leaves the loop.  Pruned queries report (inf, -1).
    const bool single = comp_bound != nullptr && wg_lo == wg_hi;
    float cb = INFINITY;
    HDB_BEGIN();
    if (single) {
        if (t == 0) red[4] = __uint_as_float(__hip_atomic_load(comp_bound + wg_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        cb = red[4];
    }
    for (int step = part; step < 2 * ntiles; step += nsplit) {
        const int tl = outward_tile(own, step, ntiles);
        if (tl < 0) continue;
        HDB_COUNT(0, 1);
        // skip a tile that lies entirely inside the single component all queries here belong to (block-uniform test)
        if (wg_lo == wg_hi && tile_lo[tl] == wg_lo && tile_hi[tl] == wg_lo) continue;
        HDB_COUNT(1, 1);
        const float gap2 = box_gap2<C>(qlo, qhi, box_lo + (long)tl * C, box_hi + (long)tl * C);
        bool need = false;
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            const float wl = fminf(bw[u], cb);
            need = need || (live[u] && gap2 <= wl && qc2[u] <= wl);
        }
        if (!__syncthreads_or(need)) continue;
        HDB_COUNT(single ? 2 : 3, 1);
        const long j0 = (long)tl * TILE;
        {
            const long j = j0 + t;
            const bool ok = j < M;
#pragma unroll
            for (int c = 0; c < C; ++c) tile[t * C + c] = ok ? x[j * C + c] : 0.f;
            tcore[t] = ok ? core2[j] : INFINITY;
            tcomp[t] = ok ? comp[j] : -1;
            tidx[t] = ok ? idx[j] : 0;
        }
        __syncthreads();
        const int nj = (int)((M - j0) < TILE ? (M - j0) : TILE);
        for (int jj = 0; jj < nj; ++jj) {
            float p[C];
            load_point<C>(tile + jj * C, p);
            const float pc2 = tcore[jj];
            const int pcomp = tcomp[jj];
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                if (pcomp == qcomp[u] || !live[u]) continue;
                float d2 = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float d = q[u][c] - p[c];
                    d2 = fmaf(d, d, d2);
                }
                const float w = fmaxf(fmaxf(qc2[u], pc2), d2);
                if (w > fminf(bw[u], cb)) continue;
                const int oj = tidx[jj];
                const int lo = qidx[u] < oj ? qidx[u] : oj, hi = qidx[u] < oj ? oj : qidx[u];
                if (w < bw[u] || lo < blo[u] || (lo == blo[u] && hi < bhi[u])) {
                    bw[u] = w;
                    bp[u] = (int)(j0 + jj);
                    blo[u] = lo;
                    bhi[u] = hi;
                }
            }
        }
        if (single) {   // publish this workgroup's minimum, fetch the component's
            float mine = INFINITY;
#pragma unroll
            for (int u = 0; u < QPT; ++u) mine = live[u] ? fminf(mine, bw[u]) : mine;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mine = fminf(mine, __shfl_xor(mine, o, 64));
            if ((t & 63) == 0) red[t >> 6] = mine;
            __syncthreads();
            if (t == 0) {
                const float bm = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
                float gl = __uint_as_float(__hip_atomic_load(comp_bound + wg_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (bm < gl) {
                    atomicMin(comp_bound + wg_lo, __float_as_uint(bm));
                    gl = bm;
                }
                red[4] = gl;
            }
            __syncthreads();
            cb = red[4];
            bool alive = false;
#pragma unroll
            for (int u = 0; u < QPT; ++u) alive = alive || (live[u] && qc2[u] <= fminf(bw[u], cb));
            if (!__syncthreads_or(alive)) break;
        }
    }
    HDB_END();
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const long i = ((long)qblock * QPT + u) * TILE + t;
        if (live[u]) {
            best_w2[(long)part * M + i] = bw[u];
            best_p[(long)part * M + i] = bp[u];
        }
    }
}

template <int C, int QT>
int launch_core_qt(const float* x, long M, int k, const float* box_lo, const float* box_hi, float* core, hipStream_t stream) {
    const size_t lds = ((size_t)C * TILE + (size_t)k * QT) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)hdb_core_dist_kernel<C, QT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)(((size_t)C * TILE + (size_t)KMAX * QT) * sizeof(float)));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((hdb_core_dist_kernel<C, QT>), dim3((unsigned)((M + QT - 1) / QT)), dim3(QT), lds, stream, x, M, k, box_lo,
                       box_hi, core);
    return 0;
}

template <int C>
int launch_core(const float* x, long M, int k, const float* box_lo, const float* box_hi, float* core, hipStream_t stream) {
    static int qt_env = -1;
    if (qt_env < 0) {
        const char* e = getenv("IGGT_HDB_CORE_QT");   // A/B switch: 128 / 256 queries per workgroup (default: by k)
        qt_env = e ? atoi(e) : 0;
    }
    const bool small = qt_env == 128 || (qt_env != 256 && k > 48);
    return small ? launch_core_qt<C, 128>(x, M, k, box_lo, box_hi, core, stream)
                 : launch_core_qt<C, 256>(x, M, k, box_lo, box_hi, core, stream);
}

}  // namespace

#ifdef IGGT_HDB_STATS
extern "C" int iggt_hdb_block_stats(unsigned long long* out, int nblocks) {   // [nblocks][4]
    if (nblocks > 131072) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hdb_blk), (size_t)nblocks * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}

extern "C" int iggt_hdb_stats(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_hdb_stats), sizeof(g_hdb_stats)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_hdb_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" int iggt_hdbscan_core_dist_f32(const float* x, long M, int C, int k, const float* box_lo, const float* box_hi,
                                          float* core, void* stream) {
    if (x == nullptr || core == nullptr || box_lo == nullptr || box_hi == nullptr || M <= 0 || M >= (1L << 31) || k < 1 || k > KMAX)
        return -1;
    int rc;
    if (C == 8) rc = launch_core<8>(x, M, k, box_lo, box_hi, core, (hipStream_t)stream);
    else if (C == 3) rc = launch_core<3>(x, M, k, box_lo, box_hi, core, (hipStream_t)stream);
    else if (C == 16) rc = launch_core<16>(x, M, k, box_lo, box_hi, core, (hipStream_t)stream);
    else return -2;
    if (rc) return rc;
    IGGT_CHECK_LAUNCH();
    return 0;
}

extern "C" int iggt_hdbscan_nearest_foreign_f32(const float* x, const float* core2, const int* comp, const int* idx,
                                                const int* tile_lo, const int* tile_hi, const float* box_lo,
                                                const float* box_hi, long M, int C, float* best_w2, int* best_p,
                                                unsigned* comp_bound, int nsplit, void* stream) {
    if (x == nullptr || core2 == nullptr || comp == nullptr || idx == nullptr || tile_lo == nullptr || tile_hi == nullptr ||
        box_lo == nullptr || box_hi == nullptr || best_w2 == nullptr || best_p == nullptr || M <= 0 || M >= (1L << 31))
        return -1;
    if (nsplit < 1 || nsplit > 64) return -1;
    const dim3 grid((unsigned)((M + QPT * TILE - 1) / (QPT * TILE)) * (unsigned)nsplit), block(TILE);
    hipStream_t st = (hipStream_t)stream;
    if (C == 8) hipLaunchKernelGGL(hdb_nearest_foreign_kernel<8>, grid, block, 0, st, x, core2, comp, idx, tile_lo, tile_hi, box_lo, box_hi, M, best_w2, best_p, comp_bound, nsplit);
    else if (C == 3) hipLaunchKernelGGL(hdb_nearest_foreign_kernel<3>, grid, block, 0, st, x, core2, comp, idx, tile_lo, tile_hi, box_lo, box_hi, M, best_w2, best_p, comp_bound, nsplit);
    else if (C == 16) hipLaunchKernelGGL(hdb_nearest_foreign_kernel<16>, grid, block, 0, st, x, core2, comp, idx, tile_lo, tile_hi, box_lo, box_hi, M, best_w2, best_p, comp_bound, nsplit);
    else return -2;
    IGGT_CHECK_LAUNCH();
    return 0;
}
