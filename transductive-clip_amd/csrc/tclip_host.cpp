// Host half of the accuracy tail (include/tclip.h: tclip_match_clusters_host).
//
// The reference assigns predicted clusters to classes with scipy.optimize.linear_sum_assignment
// on cost = -prototype (src/utils.py:380-405).  scipy is a third-party dependency of the
// reference (unpinned there; 1.15.3 in this image); its solver is the shortest-augmenting-path
// algorithm of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES
// 52(4), 2016, restated here with the same scanning order and tie rules (unassigned column
// preferred among equal reduced costs, remaining-column list filled in reverse), because on
// probability prototypes exact ties (e.g. all-zero columns) do occur and decide the labels.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <sched.h>
#include <thread>
#include <vector>

#include "../../include/tclip.h"

namespace {

// rows <= cols.  Returns col_of_row.  cost is row-major rows x cols.
bool assign_rows(int rows, int cols, const double* cost, std::vector<int>& col_of_row) {
    std::vector<double> u(rows, 0.0), v(cols, 0.0), dist(cols);
    std::vector<int> pred(cols, -1), row_of_col(cols, -1), todo(cols);
    std::vector<char> row_seen(rows), col_seen(cols);
    col_of_row.assign(rows, -1);
    for (int cur = 0; cur < rows; cur++) {
        int n_todo = cols;
        for (int it = 0; it < cols; it++) todo[it] = cols - it - 1;
        std::fill(row_seen.begin(), row_seen.end(), 0);
        std::fill(col_seen.begin(), col_seen.end(), 0);
        std::fill(dist.begin(), dist.end(), INFINITY);
        double min_val = 0.0;
        int i = cur, sink = -1;
        while (sink < 0) {
            int pick = -1;
            double lowest = INFINITY;
            row_seen[i] = 1;
            for (int it = 0; it < n_todo; it++) {
                const int j = todo[it];
                const double r = min_val + cost[(size_t)i * cols + j] - u[i] - v[j];
                if (r < dist[j]) {
                    pred[j] = i;
                    dist[j] = r;
                }
                if (dist[j] < lowest || (dist[j] == lowest && row_of_col[j] == -1)) {
                    lowest = dist[j];
                    pick = it;
                }
            }
            min_val = lowest;
            if (min_val == INFINITY) return false;
            const int j = todo[pick];
            if (row_of_col[j] == -1) sink = j;
            else i = row_of_col[j];
            col_seen[j] = 1;
            todo[pick] = todo[--n_todo];
        }
        u[cur] += min_val;
        for (int r = 0; r < rows; r++)
            if (row_seen[r] && r != cur) u[r] += min_val - dist[col_of_row[r]];
        for (int j = 0; j < cols; j++)
            if (col_seen[j]) v[j] -= min_val - dist[j];
        int j = sink;
        while (true) {
            const int r = pred[j];
            row_of_col[j] = r;
            std::swap(col_of_row[r], j);
            if (r == cur) break;
        }
    }
    return true;
}

}  // namespace

extern "C" int tclip_match_clusters_host(int32_t T, int32_t Q, int32_t K, const int32_t* preds,
                                         const int32_t* n_clusters, const int32_t* cluster_ids,
                                         const float* prototypes, const int64_t* y_q, int32_t graph_matching,
                                         int32_t* new_preds, float* acc) {
    return tclip_match_clusters_host_strided(T, Q, K, preds, n_clusters, cluster_ids, prototypes, y_q, graph_matching,
                                             Q < K ? Q : K, new_preds, acc);
}

namespace {

// tasks t0 .. t1-1; returns TCLIP_OK or the first error
int match_range(int t0, int t1, int Q, int K, int Cmax, const int32_t* preds, const int32_t* n_clusters,
                const int32_t* cluster_ids, const float* prototypes, const int64_t* y_q, int graph_matching,
                int32_t* new_preds, float* acc) {
    std::vector<double> cost;
    std::vector<int> col_of_row, lut(K);
    for (int t = t0; t < t1; t++) {
        const int C = n_clusters[t];
        if (C < 1 || C > Cmax) return TCLIP_ERR_ARG;
        const int32_t* ids = cluster_ids + (size_t)t * Cmax;
        const float* pr = prototypes + (size_t)t * Cmax * K;
        for (int c = 0; c < C; c++)
            if (ids[c] < 0 || ids[c] >= K) return TCLIP_ERR_ARG;             // labels index the look-up table below
        for (int q = 0; q < Q; q++)
            if (preds[(size_t)t * Q + q] < 0 || preds[(size_t)t * Q + q] >= K) return TCLIP_ERR_ARG;
        std::fill(lut.begin(), lut.end(), 0);
        if (graph_matching) {
            cost.resize((size_t)C * K);
            for (int c = 0; c < C; c++)
                for (int d = 0; d < K; d++) cost[(size_t)c * K + d] = -(double)pr[(size_t)c * K + d];
            if (!assign_rows(C, K, cost.data(), col_of_row)) return TCLIP_ERR_ARG;
            for (int c = 0; c < C; c++) lut[ids[c]] = col_of_row[c];
        } else {
            // compute_basic_matching: class = argmax of the cluster's prototype (first maximum)
            for (int c = 0; c < C; c++) {
                int best = 0;
                for (int d = 1; d < K; d++)
                    if (pr[(size_t)c * K + d] > pr[(size_t)c * K + best]) best = d;
                lut[ids[c]] = best;
            }
        }
        int hit = 0;
        for (int q = 0; q < Q; q++) {
            const int np = lut[preds[(size_t)t * Q + q]];
            new_preds[(size_t)t * Q + q] = np;
            hit += (int64_t)np == y_q[(size_t)t * Q + q];
        }
        // torch: (new == y).float().mean(1): sum of 0/1 floats (exact) divided by Q in fp32
        acc[t] = (float)hit / (float)Q;
    }
    return TCLIP_OK;
}

}  // namespace

// Host threads for the matching: TCLIP_HOST_THREADS if set, else up to 16 but no more than this process's share of the
// cores it may run on - the affinity mask divided by LOCAL_WORLD_SIZE (set by torch.distributed.run: one process per
// GPU, eight of them on a node would otherwise start 8 x 16 threads at the same moment of every step).
static int host_thread_cap() {
    if (const char* e = getenv("TCLIP_HOST_THREADS")) return atoi(e) > 0 ? atoi(e) : 1;
    int cores = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cores = CPU_COUNT(&set);
    if (cores <= 0) cores = (int)std::thread::hardware_concurrency();
    if (cores <= 0) cores = 1;
    int ranks = 1;
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = atoi(e) > 0 ? atoi(e) : 1;
    int n = cores / ranks;
    if (n > 16) n = 16;
    return n < 1 ? 1 : n;
}
extern "C" int tclip_host_threads(void) { return host_thread_cap(); }

// Tasks are independent: large batches are matched by a few host threads (the K = 1000 bench step spent 150 ms
// here on one thread, 1.2 % of the step).
extern "C" int tclip_match_clusters_host_strided(int32_t T, int32_t Q, int32_t K, const int32_t* preds,
                                                 const int32_t* n_clusters, const int32_t* cluster_ids,
                                                 const float* prototypes, const int64_t* y_q, int32_t graph_matching,
                                                 int32_t c_stride, int32_t* new_preds, float* acc) {
    if (T < 1 || Q < 1 || K < 2 || !preds || !n_clusters || !cluster_ids || !prototypes || !y_q || !new_preds || !acc)
        return TCLIP_ERR_ARG;
    const int Cmax = c_stride;
    if (Cmax < 1 || Cmax > (Q < K ? Q : K)) return TCLIP_ERR_ARG;
    int n_threads = host_thread_cap();
    const long work = (long)T * K;                             // below ~8 k cost-matrix columns per thread a thread is not worth starting
    if (n_threads > work / 8192) n_threads = (int)(work / 8192);
    if (n_threads > T) n_threads = T;
    if (n_threads <= 1)
        return match_range(0, T, Q, K, Cmax, preds, n_clusters, cluster_ids, prototypes, y_q, graph_matching, new_preds, acc);
    std::vector<int> rc(n_threads, TCLIP_OK);
    std::vector<std::thread> pool;
    for (int i = 0; i < n_threads; i++) {
        const int t0 = (int)((long)T * i / n_threads), t1 = (int)((long)T * (i + 1) / n_threads);
        pool.emplace_back([=, &rc] {
            rc[i] = match_range(t0, t1, Q, K, Cmax, preds, n_clusters, cluster_ids, prototypes, y_q, graph_matching, new_preds, acc);
        });
    }
    for (auto& th : pool) th.join();
    for (int i = 0; i < n_threads; i++)
        if (rc[i] != TCLIP_OK) return rc[i];
    return TCLIP_OK;
}
