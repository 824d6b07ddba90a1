// HDBSCAN* for the clustering sweep of llc_kbdm (reference llckbdm.py:264-321 calls `hdbscan.HDBSCAN(min_samples)`
// once per min_samples = 1 .. len(m_range) - 1 on the same pooled lines).  Written from the published algorithm
// (Campello, Moulavi, Sander 2013; McInnes, Healy 2017), Euclidean metric, alpha = 1, excess-of-mass selection,
// no single-cluster result, semantics of scikit-learn's `HDBSCAN(min_samples=k, min_cluster_size=c)`:
//
//   core(i)        = distance from i to its k-th nearest sample, i itself included (k = 1: 0)
//   d_mr(i, j)     = max(core(i), core(j), |x_i - x_j|)
//   MST            = Prim from sample 0 over the complete graph with weights d_mr
//   dendrogram     = single linkage of the MST edges sorted by weight (ties: edge order of the MST - deterministic)
//   condensed tree = splits whose two sides both hold >= c samples; smaller sides "fall out" as points
//   selection      = excess of mass on the cluster stabilities, root excluded; labels 0.. in order of cluster id
//
// What runs where: everything O(n^2) is on the GPU - ONE pass for the k-nearest-neighbour distances of EVERY k of
// the sweep (k_knn_dist), then one workgroup per k runs Prim (k_prim_mst), all k concurrently; the O(n log n) tree
// part below is host C++.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <vector>

namespace kb {

struct MstEdge {
    int a, b;
    double w;
};

// MST edges (n - 1, any order) -> labels (n), -1 = noise.  Returns the number of clusters.
inline int hdbscan_labels_from_mst(int n, const MstEdge* mst, int min_cluster_size, int32_t* labels) {
    for (int i = 0; i < n; ++i) labels[i] = -1;
    if (n < 2) return 0;
    const int ne = n - 1;
    // ---- single linkage: edges by ascending weight (stable), union-find with dendrogram node ids n, n+1, ...
    std::vector<int> order(ne);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return mst[x].w < mst[y].w; });
    const int nnode = 2 * n - 1;
    std::vector<int> uf(nnode), left(nnode, -1), right(nnode, -1), size(nnode, 1);
    std::vector<double> dist(nnode, 0.0);
    std::iota(uf.begin(), uf.end(), 0);
    auto find = [&](int x) {
        int r = x;
        while (uf[r] != r) r = uf[r];
        while (uf[x] != r) { const int nx = uf[x]; uf[x] = r; x = nx; }
        return r;
    };
    for (int e = 0; e < ne; ++e) {
        const MstEdge& ed = mst[order[e]];
        const int ra = find(ed.a), rb = find(ed.b);
        const int id = n + e;
        left[id] = ra; right[id] = rb; dist[id] = ed.w; size[id] = size[ra] + size[rb];
        uf[ra] = id; uf[rb] = id;
    }
    const int root = nnode - 1;
    // ---- condensed tree.  Rows: (parent cluster, child, lambda, child size); child < n: a sample, else a cluster id.
    struct Row { int parent, child; double lambda; int csize; };
    std::vector<Row> rows;
    rows.reserve(n + 64);
    std::vector<int> relabel(nnode, -1);
    std::vector<char> ignore(nnode, 0);
    std::vector<int> bfs;
    bfs.reserve(nnode);
    bfs.push_back(root);
    for (size_t h = 0; h < bfs.size(); ++h) {
        const int nd = bfs[h];
        if (nd >= n) { bfs.push_back(left[nd]); bfs.push_back(right[nd]); }
    }
    int next_label = n + 1;
    relabel[root] = n;
    std::vector<int> stack;
    auto fall_out = [&](int sub, int parent_label, double lambda) {     // every sample under `sub` leaves `parent_label`
        stack.clear();
        stack.push_back(sub);
        while (!stack.empty()) {
            const int x = stack.back();
            stack.pop_back();
            if (x < n) rows.push_back(Row{parent_label, x, lambda, 1});
            else { ignore[x] = 1; stack.push_back(left[x]); stack.push_back(right[x]); }
        }
    };
    for (int nd : bfs) {
        if (nd < n || ignore[nd]) continue;
        const int l = left[nd], r = right[nd];
        const double lambda = dist[nd] > 0.0 ? 1.0 / dist[nd] : std::numeric_limits<double>::infinity();
        const int lc = size[l], rc = size[r];
        const int pl = relabel[nd];
        if (lc >= min_cluster_size && rc >= min_cluster_size) {
            relabel[l] = next_label++;
            rows.push_back(Row{pl, relabel[l], lambda, lc});
            relabel[r] = next_label++;
            rows.push_back(Row{pl, relabel[r], lambda, rc});
        } else if (lc < min_cluster_size && rc < min_cluster_size) {
            fall_out(l, pl, lambda);
            fall_out(r, pl, lambda);
        } else if (lc < min_cluster_size) {
            relabel[r] = pl;
            fall_out(l, pl, lambda);
        } else {
            relabel[l] = pl;
            fall_out(r, pl, lambda);
        }
    }
    const int ncl = next_label - n;                        // cluster ids n .. next_label - 1
    if (ncl <= 1) return 0;                                // only the root: everything is noise
    // ---- stabilities
    std::vector<double> birth(ncl, 0.0), stab(ncl, 0.0);
    std::vector<int> parent_of(ncl, -1);
    for (const Row& rw : rows)
        if (rw.child >= n) { birth[rw.child - n] = rw.lambda; parent_of[rw.child - n] = rw.parent - n; }
    for (const Row& rw : rows) {
        const int c = rw.parent - n;
        // a sample at infinite lambda (duplicates) inside a cluster born at infinite lambda contributes nothing
        const double dl = rw.lambda - birth[c];
        if (std::isfinite(dl)) stab[c] += dl * rw.csize;
        else if (std::isinf(rw.lambda) && std::isfinite(birth[c])) stab[c] = std::numeric_limits<double>::infinity();
    }
    // ---- excess of mass, children before parents (ids grow downwards), root (0) excluded
    std::vector<double> child_sum(ncl, 0.0);
    std::vector<char> selected(ncl, 0);
    for (int c = ncl - 1; c >= 1; --c) {
        if (child_sum[c] > stab[c]) { selected[c] = 0; stab[c] = child_sum[c]; }
        else selected[c] = 1;
        child_sum[parent_of[c]] += stab[c];
    }
    // a selected cluster unselects all its descendants
    std::vector<char> under_selected(ncl, 0);
    for (int c = 1; c < ncl; ++c) {
        const int p = parent_of[c];
        under_selected[c] = (p >= 0) && (under_selected[p] || selected[p]);
        if (under_selected[c]) selected[c] = 0;
    }
    // ---- labels: cluster ids in ascending order -> 0, 1, ...; a sample belongs to the selected ancestor of the
    // cluster it fell out of (or is noise)
    std::vector<int> lab_of(ncl, -1);
    int k = 0;
    for (int c = 1; c < ncl; ++c)
        if (selected[c]) lab_of[c] = k++;
    std::vector<int> resolved(ncl, -2);
    auto resolve = [&](int c) {
        int x = c;
        while (x > 0 && !selected[x]) x = parent_of[x];
        return (x > 0) ? lab_of[x] : -1;
    };
    for (const Row& rw : rows)
        if (rw.child < n) {
            const int c = rw.parent - n;
            if (resolved[c] == -2) resolved[c] = resolve(c);
            labels[rw.child] = resolved[c];
        }
    return k;
}

}  // namespace kb
