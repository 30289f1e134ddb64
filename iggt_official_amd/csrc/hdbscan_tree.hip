// HDBSCAN, host half: minimum spanning tree of the mutual-reachability graph -> flat cluster labels.
//
// Replaces the library call inside cluster_features_to_masks_mv (reference iggt/utils/misc.py:123-129: `HDBSCAN(
// cluster_selection_epsilon, min_samples, min_cluster_size, allow_single_cluster=False).fit(all_pixels).labels_`, where HDBSCAN is
// cuml's, hdbscan's or scikit-learn's, misc.py:19-22).  The O(M^2) part -- core distances and the spanning tree -- runs on the GPU
// (csrc/hdbscan.hip); what is left is a walk over the M - 1 tree edges, done here on the host in one pass each:
//   1. single-linkage dendrogram: edges sorted by weight (stable), merged with a union-find (Campello, Moulavi, Sander 2013, section 3);
//   2. condensed tree for min_cluster_size (a split counts only if both sides keep >= min_cluster_size points, otherwise the points
//      "fall out" of the surviving cluster at lambda = 1 / distance);
//   3. stability S(C) = sum over points of (lambda_leave - lambda_birth), excess-of-mass selection bottom-up (the root is not a
//      candidate unless allow_single_cluster);
//   4. cluster_selection_epsilon (Malzer & Baum 2020): a selected cluster born at a distance below epsilon is replaced by its
//      first ancestor born above it;
//   5. labels: a point belongs to the selected cluster it falls out of, or below; everything else is noise (-1).
// Label numbering: selected clusters in increasing order of their condensed-tree id -- the convention of scikit-learn / hdbscan,
// which tests/test_hdbscan.py uses as the oracle (identical labels on its fixtures, not merely the same partition).
// Pure host code: callable without a GPU (the CPU test suite runs it).
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "../../include/iggt_hip.h"

namespace {

struct Cond {        // condensed-tree row
    int parent, child;
    double lambda;
    int size;
};

struct UF {
    std::vector<int> p;
    explicit UF(int n) : p(n) { std::iota(p.begin(), p.end(), 0); }
    int find(int x) {
        while (p[x] != x) {
            p[x] = p[p[x]];
            x = p[x];
        }
        return x;
    }
};

}  // namespace

extern "C" int iggt_hdbscan_labels_from_mst(const int* eu, const int* ev, const float* ew, long n_points, int min_cluster_size,
                                            double cluster_selection_epsilon, int allow_single_cluster, int* labels) {
    const int n = (int)n_points;
    if (n_points <= 0 || n_points > (1L << 30) || min_cluster_size < 2 || labels == nullptr) return -1;
    if (n == 1) {
        labels[0] = -1;
        return 0;
    }
    if (eu == nullptr || ev == nullptr || ew == nullptr) return -1;
    const int ne = n - 1;
    for (int e = 0; e < ne; ++e)
        if (eu[e] < 0 || eu[e] >= n || ev[e] < 0 || ev[e] >= n || !(ew[e] >= 0.f)) return -2;

    // ---- 1. dendrogram (scipy linkage convention: leaves 0 .. n-1, merge k creates node n + k) --------------------------------
    std::vector<int> order(ne);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ew[a] < ew[b]; });
    std::vector<int> left(ne), right(ne), size(ne);
    std::vector<double> dist(ne);
    {
        UF uf(n);
        std::vector<int> node_of(n);      // union-find root -> current dendrogram node
        std::vector<int> cnt(n, 1);
        std::iota(node_of.begin(), node_of.end(), 0);
        for (int k = 0; k < ne; ++k) {
            const int e = order[k];
            const int a = uf.find(eu[e]), b = uf.find(ev[e]);
            if (a == b) return -3;        // not a tree
            left[k] = node_of[a];
            right[k] = node_of[b];
            dist[k] = (double)ew[e];
            size[k] = cnt[a] + cnt[b];
            uf.p[b] = a;
            cnt[a] = size[k];
            node_of[a] = n + k;
        }
    }
    auto node_size = [&](int x) { return x < n ? 1 : size[x - n]; };

    // ---- 2. condensed tree -----------------------------------------------------------------------------------------------------
    const int root = 2 * ne;              // = n + ne - 1
    std::vector<Cond> ct;
    ct.reserve((size_t)n + 64);
    std::vector<int> relabel((size_t)root + 1, -1);
    relabel[root] = n;
    int next_label = n + 1;
    // breadth-first over the internal nodes that still carry a cluster label; a side that falls below min_cluster_size is
    // flattened into its points at once (they all leave the surviving cluster at this node's lambda)
    std::vector<int> queue{root}, stack;
    auto fall_out = [&](int sub, int parent_label, double lam) {
        stack.clear();
        stack.push_back(sub);
        while (!stack.empty()) {
            const int x = stack.back();
            stack.pop_back();
            if (x < n) {
                ct.push_back({parent_label, x, lam, 1});
            } else {
                stack.push_back(right[x - n]);
                stack.push_back(left[x - n]);
            }
        }
    };
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const int node = queue[qi];
        if (node < n) continue;
        const int l = left[node - n], r = right[node - n];
        const double d = dist[node - n];
        const double lam = d > 0.0 ? 1.0 / d : INFINITY;
        const int lc = node_size(l), rc = node_size(r);
        const int lab = relabel[node];
        if (lc >= min_cluster_size && rc >= min_cluster_size) {
            relabel[l] = next_label++;
            ct.push_back({lab, relabel[l], lam, lc});
            relabel[r] = next_label++;
            ct.push_back({lab, relabel[r], lam, rc});
            queue.push_back(l);
            queue.push_back(r);
        } else if (lc < min_cluster_size && rc < min_cluster_size) {
            fall_out(l, lab, lam);
            fall_out(r, lab, lam);
        } else if (lc < min_cluster_size) {
            relabel[r] = lab;
            fall_out(l, lab, lam);
            queue.push_back(r);
        } else {
            relabel[l] = lab;
            fall_out(r, lab, lam);
            queue.push_back(l);
        }
    }

    // ---- 3. stabilities and excess-of-mass selection ---------------------------------------------------------------------------
    const int nclus = next_label - n;     // cluster ids n .. next_label - 1, root = n
    std::vector<double> birth(nclus, 0.0), stab(nclus, 0.0);
    std::vector<int> cparent(nclus, -1);
    std::vector<std::vector<int>> kids(nclus);
    for (const Cond& c : ct)
        if (c.size > 1) {
            birth[c.child - n] = c.lambda;
            cparent[c.child - n] = c.parent - n;
            kids[c.parent - n].push_back(c.child - n);
        }
    for (const Cond& c : ct) {
        // a point that leaves at lambda = inf (duplicate points) contributes inf: kept as in the oracle
        stab[c.parent - n] += (c.lambda - birth[c.parent - n]) * c.size;
    }
    std::vector<char> is_cluster(nclus, 1);
    if (!allow_single_cluster) is_cluster[0] = 0;
    for (int c = nclus - 1; c >= (allow_single_cluster ? 0 : 1); --c) {   // children have larger ids than their parent
        double sub = 0.0;
        for (int k : kids[c]) sub += stab[k];
        if (sub > stab[c]) {
            is_cluster[c] = 0;
            stab[c] = sub;
        } else {   // this cluster wins: nothing below it is a cluster
            stack.assign(kids[c].begin(), kids[c].end());
            while (!stack.empty()) {
                const int x = stack.back();
                stack.pop_back();
                is_cluster[x] = 0;
                for (int k : kids[x]) stack.push_back(k);
            }
        }
    }

    // ---- 4. cluster_selection_epsilon ------------------------------------------------------------------------------------------
    const bool has_cluster_tree = nclus > 1;
    if (cluster_selection_epsilon != 0.0 && has_cluster_tree) {
        std::vector<int> eom;
        for (int c = 0; c < nclus; ++c)
            if (is_cluster[c]) eom.push_back(c);
        std::vector<char> selected(nclus, 0), processed(nclus, 0);
        if (eom.size() == 1 && eom[0] == 0) {
            if (allow_single_cluster) selected[0] = 1;
        } else {
            for (int leaf : eom) {
                const double eps_leaf = 1.0 / birth[leaf];
                if (eps_leaf < cluster_selection_epsilon) {
                    if (processed[leaf]) continue;
                    int cur = leaf;
                    for (;;) {   // first ancestor born at a distance above epsilon (or the child of the root / the root)
                        const int par = cparent[cur];
                        if (par == 0) {
                            if (allow_single_cluster) cur = 0;
                            break;
                        }
                        cur = par;
                        if (1.0 / birth[cur] > cluster_selection_epsilon) break;
                    }
                    selected[cur] = 1;
                    stack.assign(kids[cur].begin(), kids[cur].end());
                    while (!stack.empty()) {
                        const int x = stack.back();
                        stack.pop_back();
                        processed[x] = 1;
                        for (int k : kids[x]) stack.push_back(k);
                    }
                } else {
                    selected[leaf] = 1;
                }
            }
        }
        for (int c = 0; c < nclus; ++c) is_cluster[c] = selected[c];
    }

    // ---- 5. labels ----------------------------------------------------------------------------------------------------------------
    std::vector<int> label_of(nclus, -1);
    int nsel = 0;
    for (int c = 0; c < nclus; ++c)
        if (is_cluster[c]) label_of[c] = nsel++;
    // a point takes the label of the nearest selected ancestor-or-self of the cluster it falls out of
    std::vector<int> owner(nclus, -2);    // -2 unknown, -1 none
    auto resolve = [&](int c) {
        stack.clear();
        int x = c;
        while (x >= 0 && owner[x] == -2 && !is_cluster[x]) {
            stack.push_back(x);
            x = cparent[x];
        }
        const int res = x < 0 ? -1 : (is_cluster[x] ? x : owner[x]);
        for (int y : stack) owner[y] = res;
        return res;
    };
    double root_death = 0.0;              // allow_single_cluster with the root as the only cluster: oracle's extra rule
    if (allow_single_cluster && nsel == 1 && is_cluster[0]) {
        for (const Cond& c : ct)
            if (c.parent == n) root_death = std::max(root_death, c.lambda);
    }
    for (int i = 0; i < n; ++i) labels[i] = -1;
    for (const Cond& c : ct) {
        if (c.size != 1) continue;
        const int pc = c.parent - n;
        const int own = is_cluster[pc] ? pc : resolve(pc);
        if (own < 0) continue;
        if (own == 0) {   // only possible with allow_single_cluster: the root keeps the points that stay until its threshold
            if (!(allow_single_cluster && nsel == 1)) continue;
            const double thr = cluster_selection_epsilon != 0.0 ? 1.0 / cluster_selection_epsilon : root_death;
            if (!(c.lambda >= thr)) continue;
        }
        labels[c.child] = label_of[own];
    }
    return 0;
}
