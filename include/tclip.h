/* tclip.h - C ABI of the MI355X (gfx950) EM-Dirichlet / Hard EM-Dirichlet engine.
 *
 * The reference (SegoleneMartin/transductive-CLIP) has no native layer: its hot path is a chain
 * of PyTorch-eager ops inside four Python method classes.  This header is therefore the FFI a
 * maintainer of the reference would bind (ctypes stub in INTEGRATION.md) to replace, per batch,
 *     src/methods/zero_shot/em_dirichlet.py:179-246        EM_DIRICHLET.run_method
 *     src/methods/zero_shot/hard_em_dirichlet.py:200-271   HARD_EM_DIRICHLET.run_method
 *     src/methods/few_shot/em_dirichlet.py:149-220         EM_DIRICHLET.run_method
 *     src/methods/few_shot/hard_em_dirichlet.py:172-251    HARD_EM_DIRICHLET.run_method
 * and the accuracy tail src/methods/zero_shot/em_dirichlet.py:61-92 + src/utils.py:380-417.
 *
 * Conventions: every pointer marked "device" is a HIP device pointer of a row-major contiguous
 * array; the library never allocates device memory (the caller passes a workspace sized by
 * tclip_workspace_bytes, so PyTorch's caching allocator stays the owner), never synchronises
 * the stream, and keeps no state between calls except a thread-local error string.  All entry
 * points return 0 on success and a non-zero code otherwise (see tclip_last_error()).
 */
#ifndef TCLIP_H
#define TCLIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: tclip_alpha_tim_run, tclip_laplacian_shot_run, tclip_match_clusters_host_strided, tclip_debug_set_mm_split added
 * 3: tclip_em_dirichlet_run_tasks (tclip_task_source) added
 * 4: tclip_check_task_indices, tclip_profile_last_split_sorts, tclip_debug_set_split_keep_placement added
 * 5: TCLIP_ERR_INDEX: tclip_check_task_indices reports an out-of-range VALUE with its own code (bad arguments stay TCLIP_ERR_ARG);
 *    tclip_debug_set_dead_head added
 * (every entry point of an earlier version keeps its signature) */
#define TCLIP_ABI_VERSION 5

enum {
    TCLIP_OK = 0,
    TCLIP_ERR_ARG = 1,        /* bad argument (null pointer, non-positive size, K out of range) */
    TCLIP_ERR_WORKSPACE = 2,  /* workspace too small or misaligned */
    TCLIP_ERR_HIP = 3,        /* a HIP call failed; text in tclip_last_error() */
    TCLIP_ERR_INDEX = 4       /* tclip_check_task_indices: a value of an index tensor lies outside its table (torch: IndexError) */
};

/* One call = n_batches independent reference batches of tasks_per_batch tasks each.  Tasks of
 * one batch are coupled by the majorize-minimize stop test, which the reference evaluates over
 * the whole (n_task, K, K) tensor every 50 inner iterations (em_dirichlet.py:169-175); batches
 * never interact.  Task t = b * tasks_per_batch + n. */
typedef struct tclip_problem {
    int32_t n_batches;        /* B >= 1                                                        */
    int32_t tasks_per_batch;  /* N >= 1   (the reference's batch_size)                         */
    int32_t n_query;          /* Q >= 1   (75 in every reference config)                       */
    int32_t n_class;          /* K, 2..1024: classes == feature dimension (softmax features)   */
    int32_t n_support;        /* S >= 0: support rows per task; 0 selects the zero-shot variant */
    int32_t iters;            /* outer iterations (args.iter: 20 soft / 10 hard)               */
    int32_t iter_mm;          /* inner MM iteration cap (args.iter_mm: 1000)                   */
    int32_t lambd;            /* the reference's integer lambd = int(K/5)*Q or int(K/k_eff)*Q  */
    int32_t hard;             /* 0 = EM_DIRICHLET, 1 = HARD_EM_DIRICHLET                       */
} tclip_problem;

int tclip_abi_version(void);
const char* tclip_last_error(void);

/* Bytes of device workspace tclip_em_dirichlet_run needs for this problem (0 on bad input). */
size_t tclip_workspace_bytes(const tclip_problem* p);

/* Runs the whole loop (A3-A12 of SURVEY.md section 8a) on `stream` and returns without waiting.
 *   x_q   device [B*N, Q, K] f32  probability features of the query set (rows on the simplex)
 *   x_s   device [B*N, S, K] f32  support features, NULL iff S == 0
 *   y_s   device [B*N, S]   i64   support labels in [0, K), NULL iff S == 0
 *   u     device [B*N, Q, K] f32  out: responsibilities after the last E-step (one-hot if hard)
 *   v     device [B*N, K]    f32  out: dual variable / log class proportions
 *   alpha device [B*N, K, K] f32  out: Dirichlet parameters, one K-vector per class
 *   preds device [B*N, Q]    i32  out: argmax_k u (first maximum, as torch.argmax)
 *   criterions device [B, iters] f32 out: per outer iteration, mean_n ||a_old-a||_F/||a_old||_F
 *   mm_iters   device [B, iters] i32 out: MM iterations executed in each outer iteration
 * Inputs are not modified (the few-shot reference logs x_s/x_q in place; this does not). */
int tclip_em_dirichlet_run(const tclip_problem* p, const float* x_q, const float* x_s, const int64_t* y_s,
                           float* u, float* v, float* alpha, int32_t* preds, float* criterions,
                           int32_t* mm_iters, void* workspace, size_t workspace_bytes, void* stream);

/* The same loop fed from the feature TABLES of the task-batch loop instead of per-task tensors: what
 *     src/eval_few_shot.py:233-241            all_features_{support,query}[indices, :]   (zero-shot: eval_zero_shot.py:160-163)
 *     src/task_generator_few_shot.py:41-52    data[:, unique_labels] with unique_labels = flip(unique(labels_support))
 * materialise as x_s (T,S,K) and x_q (T,Q,K) - 16 MB per task at K = 1000, 4 shots - is read in place: row r of task t is
 * table[idx[t,r]], its column d is table column cols[t,d].  Results are those of tclip_em_dirichlet_run on the materialised
 * tensors, bit for bit.
 *   table_q device [rows_q, K] f32, q_idx device [B*N, Q] i64 (rows of table_q)
 *   table_s device [rows_s, K] f32, s_idx device [B*N, S] i64 (rows of table_s); both NULL iff S == 0
 *   cols    device [B*N, K] i32 column permutation per task (applied to both tables), or NULL for the identity
 *   y_s     device [B*N, S] i64 support labels AFTER get_task's re-indexing (new label j = position of the old label in
 *           unique_labels), NULL iff S == 0
 * tclip_em_dirichlet_run_tasks itself does not know the tables' row counts and reads what the index tensors say: call
 * tclip_check_task_indices on them first (the Python binding does, for host and device tensors alike). */
typedef struct tclip_task_source {
    const float* table_q;
    const int64_t* q_idx;
    const float* table_s;
    const int64_t* s_idx;
    const int32_t* cols;
} tclip_task_source;
int tclip_em_dirichlet_run_tasks(const tclip_problem* p, const tclip_task_source* src, const int64_t* y_s,
                                 float* u, float* v, float* alpha, int32_t* preds, float* criterions,
                                 int32_t* mm_iters, void* workspace, size_t workspace_bytes, void* stream);

/* Range check of DEVICE-resident index tensors before tclip_em_dirichlet_run_tasks / tclip_gather_rows - what torch's own
 * `all_features_query[indices, :]` (src/eval_zero_shot.py:160-163, src/eval_few_shot.py:233-241) does by raising IndexError:
 *   idx  device [n_idx]  i64, every value must lie in [0, n_rows)         (NULL with n_idx == 0: nothing to check)
 *   cols device [n_cols] i32, every value must lie in [0, n_class)        (NULL with n_cols == 0)
 * One pass over each on `stream`, then the call WAITS for the stream (hipStreamSynchronize: a host sync per call) and returns
 * TCLIP_ERR_INDEX (text in tclip_last_error) when a value is out of range - negative values included: they are rejected, not
 * wrapped as torch wraps them -, TCLIP_ERR_ARG for a null pointer / negative count, TCLIP_OK otherwise.  Negligible next to the loop it protects (a few microseconds per
 * million indices); keeps one int32 of device memory per calling thread. */
int tclip_check_task_indices(const int64_t* idx, int64_t n_idx, int64_t n_rows, const int32_t* cols, int64_t n_cols,
                             int32_t n_class, void* stream);

/* Accuracy tail, device half: per task the clusters present in `preds` in first-appearance order
 * and the mean raw feature of each (compute_acc_clustering, em_dirichlet.py:61-71).
 *   x_q device [T,Q,K] f32, preds device [T,Q] i32
 *   n_clusters device [T] i32 out; cluster_ids device [T, Cmax] i32 out (first-appearance order,
 *   -1 padded); prototypes device [T, Cmax, K] f32 out, with Cmax = min(Q, K)
 *   workspace: tclip_prototype_workspace_bytes(T, Q, K) device bytes */
size_t tclip_prototype_workspace_bytes(int32_t n_task, int32_t n_query, int32_t n_class);
int tclip_cluster_prototypes(int32_t n_task, int32_t n_query, int32_t n_class, const float* x_q,
                             const int32_t* preds, int32_t* n_clusters, int32_t* cluster_ids,
                             float* prototypes, void* workspace, size_t workspace_bytes, void* stream);

/* Accuracy tail, host half (all pointers HOST memory): assigns clusters to classes by a
 * minimum-cost rectangular assignment on cost = -prototype (float64, as utils.py:380-405 feeds
 * scipy.optimize.linear_sum_assignment) when graph_matching != 0, else by argmax of the
 * prototype (utils.py:408-417), relabels preds and scores them against y_q.
 *   new_preds host [T,Q] i32 out, acc host [T] f32 out (mean_q new_pred == y_q) */
int tclip_match_clusters_host(int32_t n_task, int32_t n_query, int32_t n_class, const int32_t* preds,
                              const int32_t* n_clusters, const int32_t* cluster_ids,
                              const float* prototypes, const int64_t* y_q, int32_t graph_matching,
                              int32_t* new_preds, float* acc);
/* The same with an explicit row count per task of cluster_ids / prototypes ([T, c_stride] / [T, c_stride, K]
 * instead of Cmax): lets the caller copy to the host only as many prototype rows as the fullest task uses. */
int tclip_match_clusters_host_strided(int32_t n_task, int32_t n_query, int32_t n_class, const int32_t* preds,
                                      const int32_t* n_clusters, const int32_t* cluster_ids, const float* prototypes,
                                      const int64_t* y_q, int32_t graph_matching, int32_t c_stride, int32_t* new_preds,
                                      float* acc);

/* Host threads tclip_match_clusters_host* may start: TCLIP_HOST_THREADS if set, else min(16, cores this process may run
 * on / LOCAL_WORLD_SIZE) - one process per GPU must not oversubscribe the node's cores. */
int tclip_host_threads(void);

/* Task construction for the task-batch loop (eval_zero_shot.py:160-168): gathers rows of a
 * device-resident feature table.  table device [n_rows, K] f32, idx device [n_out] i64,
 * out device [n_out, K] f32. */
int tclip_gather_rows(const float* table, int64_t n_rows, int32_t n_class, const int64_t* idx,
                      int64_t n_out, float* out, void* stream);

/* SOFT_KMEANS on probability features (reference: src/methods/zero_shot/soft_kmeans.py:105-220;
 * BASELINE config 3's second method).  Uses p->n_batches * p->tasks_per_batch tasks, n_query,
 * n_class and iters; n_support must be 0; the other fields are ignored.  temperature = args.T.
 *   x_q device [T,Q,K] f32;  u device [T,Q,K] out;  w device [T,K,K] out (centroids);
 *   preds device [T,Q] i32 out.  The criterion the reference logs is identically 0. */
size_t tclip_soft_kmeans_workspace_bytes(const tclip_problem* p);
int tclip_soft_kmeans_run(const tclip_problem* p, const float* x_q, float temperature, float* u, float* w,
                          int32_t* preds, void* workspace, size_t workspace_bytes, void* stream);

/* EM_GAUSSIAN on probability features (reference: src/methods/zero_shot/em_gaussian.py:107-229):
 * SOFT_KMEANS with the class-proportion term, u = softmax_k(T * (-1/2 ||w_k - z_q||^2) + lambd v_k / Q),
 * v = log(mean_q u + eps) + 1 (p->lambd as for EM-Dirichlet: int(K/5) * n_query).  Same problem
 * fields and workspace (tclip_soft_kmeans_workspace_bytes) as SOFT_KMEANS; v device [T,K] out. */
int tclip_em_gaussian_run(const tclip_problem* p, const float* x_q, float temperature, float* u, float* v, float* w,
                          int32_t* preds, void* workspace, size_t workspace_bytes, void* stream);

/* EM_GAUSSIAN_COV on probability features (reference: src/methods/zero_shot/em_gaussian_cov.py:106-257):
 * EM_GAUSSIAN with a diagonal inverse covariance per cluster and no temperature,
 *   s = sum_q u / max(sum_q u (w - z_q)^2, eps)   (empty clusters keep w and s),
 *   u = softmax_k(-1/2 sum_d s (w - z)^2 + 1/2 sum_d log(s + eps) + lambd v_k / Q).
 * Same problem fields and workspace (tclip_soft_kmeans_workspace_bytes) as EM_GAUSSIAN;
 * s device [T,K,K] out. */
int tclip_em_gaussian_cov_run(const tclip_problem* p, const float* x_q, float* u, float* v, float* w, float* s,
                              int32_t* preds, void* workspace, size_t workspace_bytes, void* stream);

/* HARD_KMEANS on probability features (reference: src/methods/zero_shot/hard_kmeans.py:26-35,
 * 128-152, 186-204): centroids = means of the members, zero for empty clusters;
 * u = one_hot(argmin_k softmax_k ||w_k - z_q||^2) (first minimum).  Same problem fields as
 * SOFT_KMEANS (iters from p->iters, n_support = 0).
 *   x_q device [T,Q,K] f32;  u device [T,Q,K] out (one-hot);  w device [T,K,K] out;
 *   preds device [T,Q] i32 out;  criterions device [n_batches, iters] out: mean over the batch's
 *   tasks of ||u_old - u||_F (the reference logs each value twice). */
size_t tclip_hard_kmeans_workspace_bytes(const tclip_problem* p);
int tclip_hard_kmeans_run(const tclip_problem* p, const float* x_q, float* u, float* w, int32_t* preds,
                          float* criterions, void* workspace, size_t workspace_bytes, void* stream);

/* KL_KMEANS on probability features (reference: src/methods/zero_shot/kl_kmeans.py:123-189):
 * centroids w = (u^T z) / max(sum_q u, 1) with empty clusters at zero, every query assigned to the
 * centroid of smallest KL(z + eps || w + eps) (first minimum).  Problem fields, outputs and
 * workspace (tclip_hard_kmeans_workspace_bytes) as for HARD_KMEANS. */
int tclip_kl_kmeans_run(const tclip_problem* p, const float* x_q, float* u, float* w, int32_t* preds,
                        float* criterions, void* workspace, size_t workspace_bytes, void* stream);

/* BD-CSPN on probability features (reference: src/methods/few_shot/bdcspn.py:42-200; feature
 * dimension = n_class).  One pass: feature normalisation (norm_type 0 = UN, 1 = L2N, 2 = CL2N with
 * the support mean of the task), support class means, query shift eta = mean(support) - mean(query),
 * soft assignment u' = softmax_k(temp * -1/2 ||w_k/|w_k| - a/|a|||^2) of the support + shifted
 * queries a, rectified prototypes = u'-weighted means of the normalised a, prediction
 * u = softmax_k(temp * -1/2 ||p_k/|p_k| - z_q/|z_q|||^2) (args.temp of bdcspn.yaml).
 * Uses n_batches * tasks_per_batch tasks, n_query, n_class, n_support; iters is ignored.
 *   x_q device [T,Q,K] f32;  x_s device [T,S,K] f32;  y_s device [T,S] i64;
 *   prototypes device [T,K,K] out;  u device [T,Q,K] out;  preds device [T,Q] i32 out (argmax of u). */
size_t tclip_bdcspn_workspace_bytes(const tclip_problem* p);
int tclip_bdcspn_run(const tclip_problem* p, const float* x_q, const float* x_s, const int64_t* y_s, float temp,
                     int32_t norm_type, float* prototypes, float* u, int32_t* preds, void* workspace,
                     size_t workspace_bytes, void* stream);

/* PADDLE on probability features (reference: src/methods/few_shot/paddle.py:94-219; feature
 * dimension = n_class).  Prototypes start as the class means of the support set; each iteration:
 * u = softmax_k(-1/2 ||w_k - z_q||^2 + lambd v_k / Q), v = log(mean_q u + eps) + 1,
 * w = (sum_q u z + support sums) / (sum_q u + support counts).  Uses n_batches * tasks_per_batch
 * tasks, n_query, n_class, n_support and iters; lambd is the method's float (paddle.yaml).
 *   x_q device [T,Q,K] f32;  x_s device [T,S,K] f32;  y_s device [T,S] i64;
 *   u device [T,Q,K] out;  v device [T,K] out;  w device [T,K,K] out;  preds device [T,Q] i32 out
 *   (argmax of u).  The criterion the reference logs is identically 0. */
size_t tclip_paddle_workspace_bytes(const tclip_problem* p);
int tclip_paddle_run(const tclip_problem* p, const float* x_q, const float* x_s, const int64_t* y_s, float lambd,
                     float* u, float* v, float* w, int32_t* preds, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ALPHA_TIM on probability features (reference: src/methods/few_shot/tim.py:192-322; feature dimension =
 * n_class).  Weights start as the class means of the support set; each of `iters` iterations takes one
 * torch.optim.Adam step (betas 0.9/0.999, eps 1e-8) of size lr on
 *   loss_weights[0] * CE(support) - (loss_weights[1] * H(mean_q p_q) - loss_weights[2] * mean_q H(p_q)),
 * p = softmax_k(temp * (x.w_k - |w_k|^2/2 - |x|^2/2)), every entropy either Shannon or the alpha-entropy of
 * order alpha_value (tim.py:276-305), with the gradient in closed form instead of autograd.  The reference's
 * matmuls and backward pass have no fixed operation order, so this entry is pinned to the reference within a
 * float tolerance (tests/test_alpha_tim.py), not bit for bit.  Uses n_batches, tasks_per_batch, n_query,
 * n_class, n_support, iters (>= 1).
 *   x_q device [T,Q,K] f32;  x_s device [T,S,K] f32;  y_s device [T,S] i64;
 *   weights device [T,K,K] out (after the last step);  logits_q device [T,Q,K] out and preds device [T,Q] i32
 *   out: the query logits of the LAST iteration's forward pass and their argmax, which is what the reference
 *   scores (tim.py:321);  criterions device [n_batches, iters] out: mean_{task,class} ||w_old - w|| per step. */
#define TCLIP_TIM_SHANNON 0
#define TCLIP_TIM_ALPHA 1
typedef struct tclip_tim_params {
    double lr;               /* lr_alpha_tim */
    float temp;              /* args.temp */
    float alpha_value;       /* order of the alpha-entropies */
    float loss_weights[3];   /* [cross-entropy, marginal entropy, conditional entropy] */
    int32_t entropies[3];    /* TCLIP_TIM_SHANNON / TCLIP_TIM_ALPHA for the same three terms */
} tclip_tim_params;
size_t tclip_alpha_tim_workspace_bytes(const tclip_problem* p);
int tclip_alpha_tim_run(const tclip_problem* p, const tclip_tim_params* prm, const float* x_q, const float* x_s,
                        const int64_t* y_s, float* weights, float* logits_q, int32_t* preds, float* criterions,
                        void* workspace, size_t workspace_bytes, void* stream);

/* LAPLACIAN_SHOT on probability features (reference: src/methods/few_shot/laplacian_shot.py:66-249; feature
 * dimension = n_class).  Rows L2-normalised (norm_type 1) or left as they are (0), prototypes = support class
 * means, unary = squared distances query-prototype, a kNN graph over the queries of a task (the knn-1 nearest
 * OTHER queries of each query, as `knnind[:, 1:]`), then `iters` bound updates Y <- softmax_k(-unary + lmd W Y)
 * with the reference's energy and its freeze rule (:162-171).  Host numpy/scipy/sklearn code in the reference,
 * one workgroup per task here; pinned to reference-made fixtures within a tolerance (tests/test_laplacian_shot.py).
 * Uses n_batches * tasks_per_batch tasks, n_query (<= 1024), n_class, n_support, iters (>= 1).
 *   x_q device [T,Q,K] f32;  x_s device [T,S,K] f32;  y_s device [T,S] i64;
 *   unary device [T,Q,K] f32 out;  neighbours device [T,Q,knn-1] i32 out (nearest first);
 *   preds_iter device [T,iters,Q] i32 out (the assignment after every update; the last one is the prediction);
 *   energies device [T,iters] f64 out. */
size_t tclip_laplacian_shot_workspace_bytes(const tclip_problem* p);
int tclip_laplacian_shot_run(const tclip_problem* p, const float* x_q, const float* x_s, const int64_t* y_s, int32_t knn,
                             double lmd, int32_t norm_type, float* unary, int32_t* neighbours, int32_t* preds_iter,
                             double* energies, void* workspace, size_t workspace_bytes, void* stream);

/* Inductive zero-shot CLIP on probability features (reference: src/methods/zero_shot/inductive_clip.py:45-49,
 * 112-126): the prediction is the arg-max of each query's probability vector, no adaptation.
 *   x device [n_rows, n_class] f32;  labels device [n_rows] i32 out (first maximum, as torch.argmax). */
int tclip_argmax_rows(const float* x, int64_t n_rows, int32_t n_class, int32_t* labels, void* stream);

/* Probability features from visual embeddings (reference: extract_features_softmax,
 * src/utils.py:287-290): out[n,:] = softmax_k(T * (visual[n]/||visual[n]||) . text[k]).
 *   visual device [n_rows, dim] f32 (any norm), text device [n_class, dim] f32 (unit-norm rows, as
 *   clip_weights returns them, src/utils.py:363-377), out device [n_rows, n_class] f32.
 * This is the only reference-consistent way to feed visual features to EM-Dirichlet (which raises
 * ValueError on features outside the simplex, em_dirichlet.py:204-208). */
int tclip_probability_features(const float* visual, const float* text, int64_t n_rows, int32_t dim, int32_t n_class,
                               float temperature, float* out, void* stream);

/* Optional instrumentation used by bench.py (thread-local, off by default).  While enabled,
 * every launch of the live-row majorize-minimize kernel (k_mm_live, the dominant kernel) issued by
 * tclip_em_dirichlet_run on this thread is bracketed by HIP events on the stream it is launched
 * on (independent batches run on a few internal streams, so launches overlap) and the
 * element-updates it executes are counted on the device.  tclip_profile_collect synchronises the device and returns, since the last collection:
 * the time during which at least one such launch was running (union of the intervals), the sum
 * of the individual launch durations, the number of launches and the element-update count. */
int tclip_profile_enable(int on);
int tclip_profile_collect(double* mm_busy_ms, double* mm_launch_ms_sum, int64_t* mm_launches,
                          int64_t* element_updates);
/* The same four figures of the LAST tclip_profile_collect of this thread per kernel: index 0 = k_mm_live (the first outer
 * iteration and row lengths without a split instantiation), 1 = k_mm_split.  Every argument points to two values. */
int tclip_profile_last_kernels(double* busy_ms, double* launch_ms_sum, int64_t* launches, int64_t* element_updates);
/* k_mm_split keeps the placement of a wavefront's elements in its three class queues from one MM iteration to the next and
 * sorts anew only when an element has left its class: the wavefront-iterations the kernel ran in the window of the LAST
 * tclip_profile_collect of this thread, and how many of them ran the full placement (the first of every launch does). */
int tclip_profile_last_split_sorts(int64_t* wave_iterations, int64_t* sorts);
/* Test hook: 0 makes k_mm_split sort its class queues in every MM iteration (what rounds 3-4 did), non-zero restores the default
 * (the placement of an iteration is kept for the next one and checked by the dense passes).  Process-wide; results do not
 * depend on it (tests/test_gpu_round5.py::test_kept_placement_is_invisible). */
int tclip_debug_set_split_keep_placement(int32_t on);

/* Dead rows are spared the rest of their schedule once a limit-cycle probe (run after each of the
 * first `chunks` 50-iteration chunks) finds them on a cycle of the fp32 map - an exact shortcut.
 * For tests: 0 disables the probe (every dead row iterates its whole schedule once), negative
 * restores the default.  Process-wide; results do not depend on it. */
int tclip_debug_set_probe_chunks(int32_t chunks);

/* Rows that have just died run only their first `iterations` MM iterations (default 12) before an early probe looks for
 * the limit cycle from there on (further snapshots 8 and 32 iterations later find rows that were still approaching their cycle);
 * rows it cannot finish take the path above from the start.  For tests: 0 disables the early probe (the path above for
 * every dead row), negative restores the default, more than 18 is refused (TCLIP_ERR_ARG).  Also off while
 * tclip_debug_set_probe_chunks(0) is in force.  Process-wide; results do not depend on it. */
int tclip_debug_set_dead_head(int32_t iterations);

/* Rows of up to 256 elements are spread over 16 lanes in the MM kernels (4 rows per wavefront), longer ones
 * over 32.  For tests: 0 forces the 32-lane layout for every row length, negative restores the default rule.
 * Process-wide; results do not depend on it. */
int tclip_debug_set_rowset_min_rows(int32_t rows);

/* From the second outer iteration on the live rows run through the class-split MM kernel (k_mm_split: every element
 * executes only what its value class a+1 < 2.3 / [2.3, 10) / >= 10 needs).  For tests: 0 never uses it, 1 uses it from
 * the first iteration on, 100 + n from outer iteration n on, negative restores the default rule (n = 1).  Process-wide;
 * results do not depend on it. */
int tclip_debug_set_mm_split(int32_t mode);

/* The MM kernels are also compiled with the row length as a constant for the class counts of the reference's datasets that
 * BASELINE.json runs (1000, 397, 100): same layout and operations as the run-time-K kernel of the row length's bucket.  For
 * tests: 0 launches the run-time-K kernels for every row length, anything else restores the default.  Process-wide; results
 * do not depend on it. */
int tclip_debug_set_fixed_k_kernels(int32_t on);

/* The squared distances of the k-means family (SOFT/HARD_KMEANS, EM_GAUSSIAN, PADDLE, BD-CSPN) and KL_KMEANS's divergences run one
 * lane per class on a 64-centroid tile staged in LDS for rows of 32 .. 511 elements (k_kmeans_logits_tile, k_kl_divergences_tile),
 * 32 lanes per class otherwise; the centroid statistics of 75-query problems stage 64 feature columns per block (k_mstats_cols75).  For
 * tests: 0 uses the round-3 kernels for every shape, negative restores the default rule.  Process-wide; results do not
 * depend on it. */
int tclip_debug_set_kmeans_tile(int32_t mode);

/* Device self-test (used by tests/test_gpu_primitives.py).  out host [14]:
 *   [0] 1/x: fast exact reciprocal vs IEEE quotient, every float of a binade at 3 exponents
 *   [1] a/b: 2^29 operand pairs            [2] fused digamma(a+1), digamma of row sums vs generic
 *   [3] fused lgamma(a+1) vs generic        [4] whole MM update, branch-free vs generic (2^24 each)
 *   [5] 1/x with two correction steps (informational)
 *   [6..13] 64-bit checksums of the restated library routines (digamma, lgamma, sqrt, exp, log,
 *           fused psi, fused lgamma, digamma_pos) over the argument streams of
 *           csrc/tclip_selftest_inputs.h; oracle/mathcheck.cpp computes the host values.
 * [0]..[4] must be 0.  Allocates (and frees) 112 bytes of device memory for the counters. */
int tclip_selftest_primitives(uint64_t* mismatches);

#ifdef __cplusplus
}
#endif
#endif /* TCLIP_H */
