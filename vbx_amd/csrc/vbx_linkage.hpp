// vbx_linkage.hpp -- average-linkage agglomerative clustering of one recording (vbhmm.py:140-141), host code.
//
// The reference calls fastcluster.linkage(condensed, method='average'); the algorithm for this method, there and in
// scipy.cluster.hierarchy.linkage (the stand-in where fastcluster is not installed; SciPy 1.15, `_hierarchy.pyx`:
// nn_chain + label), is the nearest-neighbour chain:
//
//   grow a chain x -> nn(x) -> nn(nn(x)) ... over the live clusters (ties: the previous element of the chain wins,
//   then the lowest index) until two clusters are each other's nearest neighbour; merge them (the merged cluster
//   keeps the larger index), update its distances d(i, x u y) = (n_x d(i,x) + n_y d(i,y)) / (n_x + n_y), pop both
//   from the chain; after n - 1 merges sort them by distance (stable) and relabel through a union-find so that
//   row k of Z = (smaller id, larger id, distance, members), new clusters numbered n, n + 1, ... in sorted order.
//
// This host version serves callers that bring their own condensed matrix (and is the model the device version is
// tested against bit for bit).  The driver uses the device version (vbx_ahc.hpp, nn_chain_kernel: one persistent
// workgroup walks the chain on the score matrix where it lies in HBM -- no kernel launch per step, and the T^2 / 2
// doubles of the condensed matrix never cross PCIe): 3 ms here vs the device at T = 1 000, but 0.6 s vs 0.1 s at
// T = 10 000 and 2.6 s vs 0.4 s at T = 20 000, where the README of the reference (README.md:24) puts the pain.
//
// Working storage is the full matrix (rows contiguous: every scan and update streams), n^2 doubles.  The arithmetic
// on live entries is exactly SciPy's (same operations in the same order), so Z is reproduced bit for bit.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <vector>

namespace vbx {

struct ChainMerge { int a, b; double d; };          // clusters a < b merged at distance d (b keeps the merged cluster)
inline void finish_linkage(int64_t n, const ChainMerge* merges, double* Z);

inline void average_linkage(int64_t n, const double* cond, double* Z) {
    if (n < 2) return;
    // Full matrix, rows contiguous.  A merge rewrites only ROW y of the merged cluster (one streaming pass); the
    // matching column is not touched (n stores a cache line apart each).  Instead every row remembers at which merge
    // it was last brought up to date, and before a row is scanned the entries of the clusters merged since then are
    // fetched from THEIR rows, which are newer.  Dead clusters are masked by a penalty vector (+inf), the diagonal
    // holds +inf: the scan itself is branch-free.
    const double inf = std::numeric_limits<double>::infinity();
    // The n^2 working matrix lives in a per-thread buffer that is kept between calls: a driver thread clusters one
    // recording after another, and returning 8+ MB to the system each time (munmap: a TLB shoot-down on every core
    // the process runs on) slowed the OTHER threads of the process by 5-10x.
    static thread_local std::unique_ptr<double[]> workspace;
    static thread_local size_t workspace_size = 0;
    if (workspace_size < (size_t)n * n) {
        workspace.reset();
        workspace.reset(new double[(size_t)n * n]);            // (not value-initialised: every entry is written below)
        workspace_size = (size_t)n * n;
    }
    double* const D = workspace.get();
    {
        const double* p = cond;
        for (int64_t i = 0; i < n; ++i) {                       // upper triangle: rows of the condensed vector
            double* row = &D[(size_t)i * n];
            row[i] = inf;
            for (int64_t j = i + 1; j < n; ++j) row[j] = *p++;
        }
        constexpr int64_t B = 32;                               // lower triangle: mirrored tile by tile
        for (int64_t i0 = 0; i0 < n; i0 += B)
            for (int64_t j0 = 0; j0 <= i0; j0 += B)
                for (int64_t i = i0; i < std::min(i0 + B, n); ++i)
                    for (int64_t j = j0; j < std::min(j0 + B, i); ++j) D[(size_t)i * n + j] = D[(size_t)j * n + i];
    }
    std::vector<int> size((size_t)n, 1);
    std::vector<double> dead((size_t)n, 0.0);                   // 0 for a live cluster, +inf for a dead one
    std::vector<int64_t> chain((size_t)n), row_version((size_t)n, 0);
    int64_t chain_length = 0, first_live = 0;
    std::vector<ChainMerge> merges((size_t)(n - 1));
    std::vector<int64_t> stamp((size_t)n, 0);                  // merge count at which a cluster's row was last rewritten
    std::vector<int64_t> todo((size_t)n);
    auto refresh = [&](int64_t r, int64_t k) {                  // entries of the live clusters rewritten since row r was
        double* row = &D[(size_t)r * n];                        // current: one compare per cluster, then strided fetches
        const int64_t since = row_version[(size_t)r];           // a few lines ahead of their use
        if (since < k) {
            int64_t cnt = 0;
            for (int64_t c = 0; c < n; ++c) {
                todo[(size_t)cnt] = c;
                cnt += (stamp[(size_t)c] > since) & (size[(size_t)c] > 0) & (c != r);
            }
            constexpr int64_t kAhead = 8;
            for (int64_t q = 0; q < std::min(kAhead, cnt); ++q) __builtin_prefetch(&D[(size_t)todo[(size_t)q] * n + r]);
            for (int64_t q = 0; q < cnt; ++q) {
                if (q + kAhead < cnt) __builtin_prefetch(&D[(size_t)todo[(size_t)(q + kAhead)] * n + r]);
                const int64_t c = todo[(size_t)q];
                row[c] = D[(size_t)c * n + r];
            }
        }
        row_version[(size_t)r] = k;
    };
    for (int64_t k = 0; k < n - 1; ++k) {
        int64_t x = 0, y = 0;
        double current_min = 0.0;
        if (chain_length == 0) {
            chain_length = 1;
            while (size[(size_t)first_live] == 0) ++first_live;   // the lowest live index starts a chain
            chain[0] = first_live;
        }
        while (true) {                                          // go down the chain
            x = chain[(size_t)(chain_length - 1)];
            double* row = &D[(size_t)x * n];
            refresh(x, k);
            if (chain_length > 1) {                             // the previous element wins ties
                y = chain[(size_t)(chain_length - 2)];
                current_min = row[y];
            } else {
                current_min = inf;
            }
            // first index of the minimum over the live clusters, if it beats the previous element: `dist <
            // current_min` of a scan in index order, as two streaming passes
            double m0 = inf, m1 = inf, m2 = inf, m3 = inf;      // four running minima: no dependence between the lanes
            int64_t i4 = 0;
            for (; i4 + 4 <= n; i4 += 4) {
                const double v0 = row[i4] + dead[(size_t)i4], v1 = row[i4 + 1] + dead[(size_t)i4 + 1];
                const double v2 = row[i4 + 2] + dead[(size_t)i4 + 2], v3 = row[i4 + 3] + dead[(size_t)i4 + 3];
                m0 = v0 < m0 ? v0 : m0;
                m1 = v1 < m1 ? v1 : m1;
                m2 = v2 < m2 ? v2 : m2;
                m3 = v3 < m3 ? v3 : m3;
            }
            for (; i4 < n; ++i4) {
                const double v = row[i4] + dead[(size_t)i4];
                m0 = v < m0 ? v : m0;
            }
            m0 = m1 < m0 ? m1 : m0;
            m2 = m3 < m2 ? m3 : m2;
            const double m = m2 < m0 ? m2 : m0;
            if (m < current_min) {
                int64_t i = 0;
                while (row[i] + dead[(size_t)i] != m) ++i;
                y = i;
                current_min = row[i];
            }
            if (chain_length > 1 && y == chain[(size_t)(chain_length - 2)]) break;
            chain[(size_t)chain_length++] = y;
        }
        chain_length -= 2;                                      // x and y are reciprocal nearest neighbours
        refresh(y, k);                                          // (y was scanned before the merges that shortened the chain)
        if (x > y) std::swap(x, y);
        const int nx = size[(size_t)x], ny = size[(size_t)y];
        merges[(size_t)k] = ChainMerge{(int)x, (int)y, current_min};
        size[(size_t)x] = 0;                                    // x is dropped, y becomes the merged cluster
        size[(size_t)y] = nx + ny;
        dead[(size_t)x] = inf;
        const double* rx = &D[(size_t)x * n];
        double* ry = &D[(size_t)y * n];
        const double fx = (double)nx, fy = (double)ny, fs = (double)(nx + ny);
        for (int64_t i = 0; i < n; ++i) ry[i] = (fx * rx[i] + fy * ry[i]) / fs;      // (inf on the diagonal stays inf)
        stamp[(size_t)y] = k + 1;
        row_version[(size_t)y] = k + 1;
    }
    finish_linkage(n, merges.data(), Z);
}

// The same clustering as fastcluster 1.2 computes it (fastcluster.cpp: NN_chain_core<METHOD_METR_AVERAGE> +
// generate_SciPy_dendrogram; D. Muellner, "fastcluster: Fast Hierarchical, Agglomerative Clustering Routines for R and
// Python", J. Stat. Softw. 53(9), 2013) -- the package vbhmm.py:140-141 actually calls, not installed here, restated from
// its published source.  Two things differ from SciPy's routine above:
//   * the update: d(i, x u y) = s d(i,x) + t d(i,y) with s = n_x / (n_x + n_y), t = n_y / (n_x + n_y) divided FIRST
//     (f_average), against (n_x d(i,x) + n_y d(i,y)) / (n_x + n_y) -- the same number up to its last bits;
//   * the chain: a new chain starts at the lowest live index with that cluster's nearest neighbour found by a scan in
//     index order (strict <: the lowest index wins ties); after a merge THREE elements leave the chain and the walk
//     resumes at the element before them, its predecessor being the candidate that wins ties.
// Without ties both give the same tree (reciprocal nearest neighbours are unique) and distances that agree to rounding;
// with ties either may pick another -- equally valid -- merge.  Same output convention (stable sort + union-find).
// Parity of this routine with the real package is UNPINNED (no fastcluster here, no fixture of it besides the RTTM of the
// example, which both routines reproduce: tests/test_driver.py).
inline void average_linkage_fastcluster(int64_t n, const double* cond, double* Z) {
    if (n < 2) return;
    std::vector<double> Dm((size_t)n * (size_t)(n - 1) / 2);
    std::memcpy(Dm.data(), cond, sizeof(double) * Dm.size());
    // condensed index of (r, c), r < c  (fastcluster's D_ macro)
    auto D = [&](int64_t r, int64_t c) -> double& { return Dm[(size_t)(((2 * n - 3 - r) * r) >> 1) + (size_t)c - 1]; };
    std::vector<int64_t> succ((size_t)n + 1), pred((size_t)n + 1);      // doubly linked list of the live clusters
    for (int64_t i = 0; i <= n; ++i) { succ[(size_t)i] = i + 1; pred[(size_t)i] = i - 1; }
    int64_t start = 0;
    auto remove = [&](int64_t idx) {
        if (idx == start) start = succ[(size_t)idx];
        else {
            succ[(size_t)pred[(size_t)idx]] = succ[(size_t)idx];
            pred[(size_t)succ[(size_t)idx]] = pred[(size_t)idx];
        }
        succ[(size_t)idx] = 0;                                          // (marks idx as dead, as fastcluster does)
    };
    std::vector<int64_t> chain((size_t)n);
    std::vector<double> members((size_t)n, 1.0);
    std::vector<ChainMerge> merges((size_t)(n - 1));
    int64_t tip = 0, idx1 = 0, idx2 = 0;
    double mn = 0.0;
    for (int64_t j = 0; j < n - 1; ++j) {
        if (tip <= 3) {
            chain[0] = idx1 = start;
            tip = 1;
            idx2 = succ[(size_t)idx1];
            mn = D(idx1, idx2);
            for (int64_t i = succ[(size_t)idx2]; i < n; i = succ[(size_t)i])
                if (D(idx1, i) < mn) { mn = D(idx1, i); idx2 = i; }
        } else {
            tip -= 3;
            idx1 = chain[(size_t)(tip - 1)];
            idx2 = chain[(size_t)tip];
            mn = idx1 < idx2 ? D(idx1, idx2) : D(idx2, idx1);
        }
        do {
            chain[(size_t)tip] = idx2;
            for (int64_t i = start; i < idx2; i = succ[(size_t)i])
                if (D(i, idx2) < mn) { mn = D(i, idx2); idx1 = i; }
            for (int64_t i = succ[(size_t)idx2]; i < n; i = succ[(size_t)i])
                if (D(idx2, i) < mn) { mn = D(idx2, i); idx1 = i; }
            idx2 = idx1;
            idx1 = chain[(size_t)tip++];
        } while (idx2 != chain[(size_t)(tip - 2)]);
        if (idx1 > idx2) std::swap(idx1, idx2);
        merges[(size_t)j] = ChainMerge{(int)idx1, (int)idx2, mn};
        const double size1 = members[(size_t)idx1], size2 = members[(size_t)idx2];
        members[(size_t)idx2] += members[(size_t)idx1];
        remove(idx1);
        const double s = size1 / (size1 + size2), t = size2 / (size1 + size2);
        int64_t i = start;
        for (; i < idx1; i = succ[(size_t)i]) D(i, idx2) = s * D(i, idx1) + t * D(i, idx2);
        for (; i < idx2; i = succ[(size_t)i]) D(i, idx2) = s * D(idx1, i) + t * D(i, idx2);
        for (i = succ[(size_t)idx2]; i < n; i = succ[(size_t)i]) D(idx2, i) = s * D(idx1, i) + t * D(idx2, i);
    }
    finish_linkage(n, merges.data(), Z);
}

// Second half of the algorithm, shared with the device chain (vbx_ahc.hpp nn_chain_kernel): the n - 1 merges in the
// order the chain found them -> Z.
inline void finish_linkage(int64_t n, const ChainMerge* merges_in, double* Z) {
    const ChainMerge* merges = merges_in;
    // stable sort by distance, then cluster ids through a union-find (labels n, n + 1, ... in sorted order)
    std::vector<int64_t> order((size_t)(n - 1));
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t p, int64_t q) { return merges[(size_t)p].d < merges[(size_t)q].d; });
    std::vector<int64_t> parent((size_t)(2 * n - 1));
    std::vector<int64_t> members((size_t)(2 * n - 1), 1);
    std::iota(parent.begin(), parent.end(), (int64_t)0);
    int64_t next_label = n;
    auto find = [&](int64_t v) {
        int64_t root = v;
        while (parent[(size_t)root] != root) root = parent[(size_t)root];
        while (parent[(size_t)v] != root) {                     // path compression
            const int64_t up = parent[(size_t)v];
            parent[(size_t)v] = root;
            v = up;
        }
        return root;
    };
    for (int64_t k = 0; k < n - 1; ++k) {
        const ChainMerge& m = merges[(size_t)order[(size_t)k]];
        const int64_t ra = find(m.a), rb = find(m.b);
        double* z = Z + 4 * k;
        z[0] = (double)std::min(ra, rb);
        z[1] = (double)std::max(ra, rb);
        z[2] = m.d;
        parent[(size_t)ra] = next_label;
        parent[(size_t)rb] = next_label;
        members[(size_t)next_label] = members[(size_t)ra] + members[(size_t)rb];
        z[3] = (double)members[(size_t)next_label];
        ++next_label;
    }
}

// Flat clusters of the dendrogram Z cut at cophenetic distance t (vbhmm.py:145-146 `fcluster(lin_mat, t,
// criterion='distance')`; SciPy 1.15 `_hierarchy.pyx`: get_max_dist_for_each_cluster + cluster_monocrit): a subtree
// whose largest merge distance is <= t is one cluster; clusters are numbered 1, 2, ... in the order a depth-first
// walk from the root (left child first) meets them.  labels: [n].
inline void fcluster_distance(int64_t n, const double* Z, double t, int32_t* labels) {
    if (n == 1) { labels[0] = 1; return; }
    auto lc = [&](int64_t node) { return (int64_t)Z[4 * node]; };
    auto rc = [&](int64_t node) { return (int64_t)Z[4 * node + 1]; };
    // largest merge distance inside every subtree (children are always created before their parent)
    std::vector<double> md((size_t)(n - 1));
    for (int64_t k = 0; k < n - 1; ++k) {
        double m = Z[4 * k + 2];
        if (lc(k) >= n) m = std::max(m, md[(size_t)(lc(k) - n)]);
        if (rc(k) >= n) m = std::max(m, md[(size_t)(rc(k) - n)]);
        md[(size_t)k] = m;
    }
    std::vector<int64_t> stack((size_t)n);
    std::vector<char> visited((size_t)(2 * n - 1), 0);
    int32_t n_cluster = 0;
    int64_t leader = -1, k = 0;
    stack[0] = 2 * n - 2;
    while (k >= 0) {
        const int64_t root = stack[(size_t)k] - n;
        const int64_t l = lc(root), r = rc(root);
        if (leader == -1 && md[(size_t)root] <= t) {
            leader = root;
            ++n_cluster;
        }
        if (l >= n && !visited[(size_t)l]) {
            visited[(size_t)l] = 1;
            stack[(size_t)++k] = l;
            continue;
        }
        if (r >= n && !visited[(size_t)r]) {
            visited[(size_t)r] = 1;
            stack[(size_t)++k] = r;
            continue;
        }
        if (l < n) {
            if (leader == -1) ++n_cluster;
            labels[l] = n_cluster;
        }
        if (r < n) {
            if (leader == -1) ++n_cluster;
            labels[r] = n_cluster;
        }
        if (leader == root) leader = -1;
        --k;
    }
}

// Index of a binary Kaldi vector archive held in memory (vbhmm.py:117 kaldi_io.read_vec_flt_ark): for every entry
// '<key> \0B' ('FV ' | 'DV ') '\4' <int32 dim> <dim values>, the offsets of key and data.  Returns the number of
// entries, -1 for a malformed or non-binary archive (text archives go through the Python reader), -2 when the
// output arrays are too short (call again with more room).
inline int64_t ark_index(const unsigned char* buf, int64_t len, int64_t cap, int64_t* key_off, int32_t* key_len,
                         int64_t* data_off, int32_t* dim, int32_t* elem_size) {
    int64_t pos = 0, count = 0;
    while (pos < len) {
        while (pos < len && (buf[pos] == '\n' || buf[pos] == '\r' || buf[pos] == '\t')) ++pos;
        const int64_t k0 = pos;
        while (pos < len && buf[pos] != ' ') ++pos;
        if (pos == k0) break;                                   // end of the archive
        if (pos + 11 > len) return -1;
        const int64_t klen = pos - k0;
        ++pos;                                                  // the space
        if (buf[pos] != 0 || buf[pos + 1] != 'B') return -1;
        pos += 2;
        int32_t es;
        if (buf[pos] == 'F' && buf[pos + 1] == 'V' && buf[pos + 2] == ' ') es = 4;
        else if (buf[pos] == 'D' && buf[pos + 1] == 'V' && buf[pos + 2] == ' ') es = 8;
        else return -1;
        pos += 3;
        if (buf[pos] != 4) return -1;
        int32_t d;
        std::memcpy(&d, buf + pos + 1, 4);
        pos += 5;
        if (d < 0 || pos + (int64_t)d * es > len) return -1;
        if (count >= cap) return -2;
        key_off[count] = k0;
        key_len[count] = (int32_t)klen;
        data_off[count] = pos;
        dim[count] = d;
        elem_size[count] = es;
        ++count;
        pos += (int64_t)d * es;
    }
    return count;
}

}  // namespace vbx
