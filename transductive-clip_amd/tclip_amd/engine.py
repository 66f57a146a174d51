"""Host-side driver of the HIP engine: torch tensors in, torch tensors out.

PyTorch is plumbing here (device memory through its caching allocator, the current HIP stream);
every number is produced by libtclip.so.  One call handles n_batches independent reference
batches (the MM stop test couples the tasks of one batch, so the batch is the
unit of parity and of multi-GPU sharding)."""
import ctypes

import torch

from . import _capi


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the EM-Dirichlet engine has no CPU path")


class EMDirichletResult:
    __slots__ = ("u", "v", "alpha", "preds", "criterions", "mm_iters")

    def __init__(self, **kw):
        for k, val in kw.items():
            setattr(self, k, val)


class _Call:
    """One engine call: workspace from PyTorch's caching allocator (256-byte aligned view), output tensors on
    the inputs' device, the entry point launched on the current stream, the workspace kept alive until that
    stream has consumed it.  `args(ws_ptr, ws_bytes, stream)` builds the C argument tuple."""

    def __init__(self, dev, problem, ws_query):
        self.dev, self.p, self.lib = dev, problem, _capi.lib()
        self.ws_bytes = getattr(self.lib, ws_query)(ctypes.byref(problem))
        if self.ws_bytes == 0:
            raise RuntimeError(f"{ws_query} rejected the problem: " + self.lib.tclip_last_error().decode())

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    def launch(self, entry, args):
        with torch.cuda.device(self.dev):
            ws = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=self.dev)
            off = (-ws.data_ptr()) % 256
            rc = getattr(self.lib, entry)(ctypes.byref(self.p), *args(ctypes.c_void_p(ws.data_ptr() + off), self.ws_bytes, _stream()))
            _capi.check(rc, entry)
            ws.record_stream(torch.cuda.current_stream())


def _query(x_q, name="x_q"):
    _require_cuda(x_q, name)
    return x_q.contiguous().float()


def _support(x_q, x_s, y_s):
    _require_cuda(x_s, "x_s")
    _require_cuda(y_s, "y_s")
    x_s = x_s.contiguous().float()
    y_s = y_s.reshape(x_s.shape[0], -1).contiguous().long()
    T, _, K = x_q.shape
    if x_s.shape[0] != T or x_s.shape[2] != K or y_s.shape != x_s.shape[:2]:
        raise ValueError("x_s must be (T,S,K) and y_s (T,S) with the T and K of x_q")
    return x_s, y_s


def run_em_dirichlet(x_q, x_s=None, y_s=None, *, n_batches=1, iters, iter_mm=1000, lambd, hard=False):
    """x_q (T,Q,K) f32 cuda with T = n_batches * tasks_per_batch; x_s (T,S,K), y_s (T,S) for few-shot.

    Returns EMDirichletResult of cuda tensors; nothing is synchronised."""
    x_q = _query(x_q)
    T, Q, K = x_q.shape
    if T % n_batches:
        raise ValueError("number of tasks must be a multiple of n_batches")
    S = 0
    if x_s is not None:
        x_s, y_s = _support(x_q, x_s, y_s.to(x_q.device))
        S = x_s.shape[1]
    c = _Call(x_q.device, _capi.Problem(n_batches, T // n_batches, Q, K, S, iters, iter_mm, int(lambd), int(bool(hard))),
              "tclip_workspace_bytes")
    u, v, alpha, preds = c.empty(T, Q, K), c.empty(T, K), c.empty(T, K, K), c.empty(T, Q, dtype=torch.int32)
    crit = torch.zeros(n_batches, max(iters, 1), device=x_q.device)[:, :iters].contiguous()
    mm = torch.zeros(n_batches, max(iters, 1), dtype=torch.int32, device=x_q.device)[:, :iters].contiguous()
    c.launch("tclip_em_dirichlet_run", lambda ws, n, st: (_ptr(x_q), _ptr(x_s), _ptr(y_s), _ptr(u), _ptr(v), _ptr(alpha),
                                                          _ptr(preds), _ptr(crit), _ptr(mm), ws, n, st))
    return EMDirichletResult(u=u, v=v, alpha=alpha, preds=preds, criterions=crit, mm_iters=mm)


def _check_on_device(idx, n_rows, cols, n_class, name):
    """tclip_check_task_indices on device-resident tensors: IndexError where torch's own `table[idx]` would raise one
    (TCLIP_ERR_INDEX only; a bad argument is a RuntimeError like every other failed call).  Negative values are rejected,
    not wrapped as torch wraps them - on the host path (`_index_tensor`) too.  The call waits for the stream: one host
    synchronisation per checked tensor."""
    with torch.cuda.device(idx.device if idx is not None else cols.device):
        rc = _capi.lib().tclip_check_task_indices(_ptr(idx) if idx is not None else None, idx.numel() if idx is not None else 0, max(1, int(n_rows)),
                                                  _ptr(cols) if cols is not None else None, cols.numel() if cols is not None else 0, int(n_class), _stream())
    if rc == 4:                             # TCLIP_ERR_INDEX
        raise IndexError(f"{name}: {_capi.lib().tclip_last_error().decode()}")
    _capi.check(rc, "tclip_check_task_indices")


def _index_tensor(idx, n_rows, dev, name):
    """int64 (T,R) index tensor on the device, every value checked against the table's row count as torch's own
    `table[idx]` checks it (IndexError; negative values are rejected, not wrapped): on the host when that is where the tensor
    is (the samplers produce CPU tensors), by one pass on the device otherwise (tclip_check_task_indices, which synchronises
    the stream: a caller that keeps its index tensors on the device pays one host sync per call)"""
    idx = idx.long()
    if not idx.is_cuda:
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= n_rows):
            raise IndexError(f"{name}: index out of range for a table of {n_rows} rows")
        return idx.to(dev).contiguous()
    idx = idx.to(dev).contiguous()
    if idx.numel():
        _check_on_device(idx, n_rows, None, 1, name)
    return idx


def run_em_dirichlet_tasks(table_q, q_idx, table_s=None, s_idx=None, y_s=None, cols=None, *, n_batches=1, iters, iter_mm=1000,
                           lambd, hard=False):
    """The loop of run_em_dirichlet fed from the task-batch loop's feature tables (tclip_em_dirichlet_run_tasks):
    table_q (rows,K) f32 cuda, q_idx (T,Q) rows of it; few-shot: table_s, s_idx (T,S), y_s (T,S) the re-indexed support
    labels; cols (T,K) the per-task column permutation of Tasks_Generator_few_shot.get_task or None.  No (T,S,K) /
    (T,Q,K) tensor is built; the results are those of run_em_dirichlet on the materialised tensors, bit for bit."""
    _require_cuda(table_q, "table_q")
    table_q = table_q.contiguous().float()
    dev, K = table_q.device, table_q.shape[1]
    q_idx = _index_tensor(q_idx, table_q.shape[0], dev, "q_idx")
    T, Q = q_idx.shape
    if T % n_batches:
        raise ValueError("number of tasks must be a multiple of n_batches")
    S = 0
    if table_s is not None:
        _require_cuda(table_s, "table_s")
        table_s = table_s.contiguous().float()
        s_idx = _index_tensor(s_idx, table_s.shape[0], dev, "s_idx")
        S = s_idx.shape[1]
        y_s = y_s.reshape(T, -1).long().to(dev).contiguous()
        if table_s.shape[1] != K or s_idx.shape[0] != T or tuple(y_s.shape) != (T, S):
            raise ValueError("table_s must be (rows,K), s_idx and y_s (T,S) with the T of q_idx")
    if cols is not None:
        cols = cols.to(torch.int32)
        if tuple(cols.shape) != (T, K) or (not cols.is_cuda and (int(cols.min()) < 0 or int(cols.max()) >= K)):
            raise IndexError("cols must be (T,K) with values in [0, K)")
        on_device = cols.is_cuda
        cols = cols.to(dev).contiguous()
        if on_device:
            _check_on_device(None, 1, cols, K, "cols")
    c = _Call(dev, _capi.Problem(n_batches, T // n_batches, Q, K, S, iters, iter_mm, int(lambd), int(bool(hard))), "tclip_workspace_bytes")
    u, v, alpha, preds = c.empty(T, Q, K), c.empty(T, K), c.empty(T, K, K), c.empty(T, Q, dtype=torch.int32)
    crit = torch.zeros(n_batches, max(iters, 1), device=dev)[:, :iters].contiguous()
    mm = torch.zeros(n_batches, max(iters, 1), dtype=torch.int32, device=dev)[:, :iters].contiguous()
    ptr = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    src = _capi.TaskSource(ptr(table_q), ptr(q_idx), ptr(table_s), ptr(s_idx), ptr(cols))
    c.launch("tclip_em_dirichlet_run_tasks", lambda ws, n, st: (ctypes.byref(src), _ptr(y_s), _ptr(u), _ptr(v), _ptr(alpha),
                                                                _ptr(preds), _ptr(crit), _ptr(mm), ws, n, st))
    return EMDirichletResult(u=u, v=v, alpha=alpha, preds=preds, criterions=crit, mm_iters=mm)


def run_soft_kmeans(x_q, *, iters, temperature):
    """SOFT_KMEANS: x_q (T,Q,K) f32 cuda -> (u (T,Q,K), w (T,K,K), preds (T,Q) i32), cuda, not synchronised."""
    x_q = _query(x_q)
    T, Q, K = x_q.shape
    c = _Call(x_q.device, _capi.Problem(1, T, Q, K, 0, iters, 1, 0, 0), "tclip_soft_kmeans_workspace_bytes")
    u, w, preds = c.empty(T, Q, K), c.empty(T, K, K), c.empty(T, Q, dtype=torch.int32)
    c.launch("tclip_soft_kmeans_run", lambda ws, n, st: (_ptr(x_q), ctypes.c_float(float(temperature)), _ptr(u), _ptr(w),
                                                         _ptr(preds), ws, n, st))
    return u, w, preds


def run_em_gaussian(x_q, *, iters, temperature, lambd):
    """EM_GAUSSIAN: x_q (T,Q,K) f32 cuda -> (u (T,Q,K), v (T,K), w (T,K,K), preds (T,Q) i32), cuda,
    not synchronised."""
    x_q = _query(x_q)
    T, Q, K = x_q.shape
    c = _Call(x_q.device, _capi.Problem(1, T, Q, K, 0, iters, 1, int(lambd), 0), "tclip_soft_kmeans_workspace_bytes")
    u, v, w, preds = c.empty(T, Q, K), c.empty(T, K), c.empty(T, K, K), c.empty(T, Q, dtype=torch.int32)
    c.launch("tclip_em_gaussian_run", lambda ws, n, st: (_ptr(x_q), ctypes.c_float(float(temperature)), _ptr(u), _ptr(v),
                                                         _ptr(w), _ptr(preds), ws, n, st))
    return u, v, w, preds


def run_em_gaussian_cov(x_q, *, iters, lambd):
    """EM_GAUSSIAN_COV: x_q (T,Q,K) f32 cuda -> (u (T,Q,K), v (T,K), w (T,K,K), s (T,K,K), preds (T,Q) i32),
    cuda, not synchronised."""
    x_q = _query(x_q)
    T, Q, K = x_q.shape
    c = _Call(x_q.device, _capi.Problem(1, T, Q, K, 0, iters, 1, int(lambd), 0), "tclip_soft_kmeans_workspace_bytes")
    u, v, w, s, preds = c.empty(T, Q, K), c.empty(T, K), c.empty(T, K, K), c.empty(T, K, K), c.empty(T, Q, dtype=torch.int32)
    c.launch("tclip_em_gaussian_cov_run", lambda ws, n, st: (_ptr(x_q), _ptr(u), _ptr(v), _ptr(w), _ptr(s), _ptr(preds), ws, n, st))
    return u, v, w, s, preds


def run_kl_kmeans(x_q, *, iters, n_batches=1):
    """KL_KMEANS: same outputs as run_hard_kmeans."""
    return run_hard_kmeans(x_q, iters=iters, n_batches=n_batches, _entry="tclip_kl_kmeans_run")


def run_hard_kmeans(x_q, *, iters, n_batches=1, _entry="tclip_hard_kmeans_run"):
    """HARD_KMEANS: x_q (T,Q,K) f32 cuda -> (u one-hot (T,Q,K), w (T,K,K), preds (T,Q) i32,
    criterions (n_batches, iters)), cuda, not synchronised."""
    x_q = _query(x_q)
    T, Q, K = x_q.shape
    if T % n_batches:
        raise ValueError("the number of tasks must be a multiple of n_batches")
    c = _Call(x_q.device, _capi.Problem(n_batches, T // n_batches, Q, K, 0, iters, 1, 0, 0), "tclip_hard_kmeans_workspace_bytes")
    u, w, preds, crit = c.empty(T, Q, K), c.empty(T, K, K), c.empty(T, Q, dtype=torch.int32), c.empty(n_batches, iters)
    c.launch(_entry, lambda ws, n, st: (_ptr(x_q), _ptr(u), _ptr(w), _ptr(preds), _ptr(crit), ws, n, st))
    return u, w, preds, crit


def run_paddle(x_q, x_s, y_s, *, iters, lambd):
    """PADDLE: x_q (T,Q,K), x_s (T,S,K) f32 cuda, y_s (T,S) int64 cuda ->
    (u (T,Q,K), v (T,K), w (T,K,K), preds (T,Q) i32), cuda, not synchronised."""
    x_q = _query(x_q)
    x_s, y_s = _support(x_q, x_s, y_s)
    T, Q, K = x_q.shape
    c = _Call(x_q.device, _capi.Problem(1, T, Q, K, x_s.shape[1], iters, 1, 0, 0), "tclip_paddle_workspace_bytes")
    u, v, w, preds = c.empty(T, Q, K), c.empty(T, K), c.empty(T, K, K), c.empty(T, Q, dtype=torch.int32)
    c.launch("tclip_paddle_run", lambda ws, n, st: (_ptr(x_q), _ptr(x_s), _ptr(y_s), ctypes.c_float(float(lambd)), _ptr(u),
                                                    _ptr(v), _ptr(w), _ptr(preds), ws, n, st))
    return u, v, w, preds


ENTROPIES = {"Shannon": 0, "Alpha": 1}


def run_alpha_tim(x_q, x_s, y_s, *, iters, temp, lr, alpha_value, loss_weights=(1.0, 1.0, 1.0),
                  entropies=("Shannon", "Alpha", "Alpha"), n_batches=1):
    """ALPHA_TIM: x_q (T,Q,K), x_s (T,S,K) f32 cuda, y_s (T,S) int64 cuda -> (weights (T,K,K), logits_q (T,Q,K) of the
    last iteration's forward pass, preds (T,Q) i32 = their argmax, criterions (n_batches, iters)), cuda, not synchronised."""
    for e in entropies:
        if e not in ENTROPIES:
            raise ValueError("Entropies must be in ['Shannon', 'Alpha']")        # tim.py:286, 295, 305
    x_q = _query(x_q)
    x_s, y_s = _support(x_q, x_s, y_s)
    T, Q, K = x_q.shape
    if T % n_batches:
        raise ValueError("the number of tasks must be a multiple of n_batches")
    prm = _capi.TimParams(float(lr), float(temp), float(alpha_value), (ctypes.c_float * 3)(*[float(w) for w in loss_weights]),
                          (ctypes.c_int32 * 3)(*[ENTROPIES[e] for e in entropies]))
    c = _Call(x_q.device, _capi.Problem(n_batches, T // n_batches, Q, K, x_s.shape[1], iters, 1, 0, 0),
              "tclip_alpha_tim_workspace_bytes")
    weights, logits_q, preds, crit = c.empty(T, K, K), c.empty(T, Q, K), c.empty(T, Q, dtype=torch.int32), c.empty(n_batches, iters)
    c.launch("tclip_alpha_tim_run", lambda ws, n, st: (ctypes.byref(prm), _ptr(x_q), _ptr(x_s), _ptr(y_s), _ptr(weights),
                                                       _ptr(logits_q), _ptr(preds), _ptr(crit), ws, n, st))
    return weights, logits_q, preds, crit


def run_laplacian_shot(x_q, x_s, y_s, *, iters, knn, lmd, norm_type="L2N"):
    """LAPLACIAN_SHOT: x_q (T,Q,K), x_s (T,S,K) f32 cuda, y_s (T,S) int64 cuda -> (unary (T,Q,K), neighbours (T,Q,knn-1) i32,
    preds_iter (T,iters,Q) i32, energies (T,iters) f64), cuda, not synchronised."""
    if norm_type not in ("UN", "L2N"):
        raise ValueError("norm_type must be 'UN' or 'L2N' (the reference's CL2N needs a train mean it never passes)")
    x_q = _query(x_q)
    x_s, y_s = _support(x_q, x_s, y_s)
    T, Q, K = x_q.shape
    c = _Call(x_q.device, _capi.Problem(1, T, Q, K, x_s.shape[1], iters, 1, 0, 0), "tclip_laplacian_shot_workspace_bytes")
    unary, nbr = c.empty(T, Q, K), c.empty(T, Q, max(int(knn) - 1, 1), dtype=torch.int32)
    preds_iter, energies = c.empty(T, max(iters, 1), Q, dtype=torch.int32), c.empty(T, max(iters, 1), dtype=torch.float64)
    c.launch("tclip_laplacian_shot_run", lambda ws, n, st: (_ptr(x_q), _ptr(x_s), _ptr(y_s), ctypes.c_int32(int(knn)),
                                                            ctypes.c_double(float(lmd)), ctypes.c_int32(NORM_TYPES[norm_type]),
                                                            _ptr(unary), _ptr(nbr), _ptr(preds_iter), _ptr(energies), ws, n, st))
    return unary, nbr, preds_iter, energies


def argmax_rows(x):
    """x (..., K) f32 cuda -> int32 (...) indices of the first maximum of every row, cuda, not synchronised."""
    _require_cuda(x, "x")
    x = x.contiguous().float()
    K = x.shape[-1]
    rows = x.numel() // K
    with torch.cuda.device(x.device):
        labels = torch.empty(x.shape[:-1], dtype=torch.int32, device=x.device)
        rc = _capi.lib().tclip_argmax_rows(_ptr(x), ctypes.c_int64(rows), ctypes.c_int32(K), _ptr(labels), _stream())
        _capi.check(rc, "tclip_argmax_rows")
    return labels


NORM_TYPES = {"UN": 0, "L2N": 1, "CL2N": 2}


def run_bdcspn(x_q, x_s, y_s, *, temp, norm_type="L2N"):
    """BD-CSPN: x_q (T,Q,K), x_s (T,S,K) f32 cuda, y_s (T,S) int64 cuda ->
    (prototypes (T,K,K), u (T,Q,K), preds (T,Q) i32), cuda, not synchronised."""
    if norm_type not in NORM_TYPES:
        raise ValueError(f"norm_type must be one of {sorted(NORM_TYPES)}")
    x_q = _query(x_q)
    x_s, y_s = _support(x_q, x_s, y_s)
    T, Q, K = x_q.shape
    c = _Call(x_q.device, _capi.Problem(1, T, Q, K, x_s.shape[1], 1, 1, 0, 0), "tclip_bdcspn_workspace_bytes")
    prototypes, u, preds = c.empty(T, K, K), c.empty(T, Q, K), c.empty(T, Q, dtype=torch.int32)
    c.launch("tclip_bdcspn_run", lambda ws, n, st: (_ptr(x_q), _ptr(x_s), _ptr(y_s), ctypes.c_float(float(temp)),
                                                    ctypes.c_int32(NORM_TYPES[norm_type]), _ptr(prototypes), _ptr(u),
                                                    _ptr(preds), ws, n, st))
    return prototypes, u, preds


def clustering_accuracy(x_q, preds, y_q, graph_matching=True):
    """Zero-shot accuracy tail: device prototypes of the predicted clusters, host assignment.

    x_q (T,Q,K) cuda f32, preds (T,Q) cuda i32, y_q (T,Q) int64 (any device).
    Returns (acc (T,) f32 cpu, new_preds (T,Q) i32 cpu)."""
    _require_cuda(x_q, "x_q")
    x_q = x_q.contiguous().float()
    T, Q, K = x_q.shape
    dev = x_q.device
    lib = _capi.lib()
    cmax = min(Q, K)
    with torch.cuda.device(dev):
        ws_bytes = lib.tclip_prototype_workspace_bytes(T, Q, K)
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        n_clusters = torch.empty(T, dtype=torch.int32, device=dev)
        ids = torch.empty(T, cmax, dtype=torch.int32, device=dev)
        protos = torch.empty(T, cmax, K, device=dev)        # rows beyond a task's cluster count are never read
        preds = preds.to(dev).int().contiguous()
        rc = lib.tclip_cluster_prototypes(T, Q, K, _ptr(x_q), _ptr(preds), _ptr(n_clusters), _ptr(ids), _ptr(protos),
                                          ctypes.c_void_p(ws.data_ptr() + off), ws_bytes, _stream())
        _capi.check(rc, "tclip_cluster_prototypes")
        preds_h, nc_h = preds.cpu(), n_clusters.cpu()
        used = max(1, min(cmax, int(nc_h.max())))          # rows of the fullest task: only those travel to the host
        # page-locked staging buffers (torch caches them): the prototype block is the one sizeable device-to-host copy of a step
        ids_h = torch.empty((T, used), dtype=torch.int32, pin_memory=True)
        protos_h = torch.empty((T, used, K), dtype=torch.float32, pin_memory=True)
        ids_h.copy_(ids[:, :used], non_blocking=True)
        protos_h.copy_(protos[:, :used], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    y_h = y_q.reshape(T, Q).long().cpu().contiguous()
    new_preds = torch.empty(T, Q, dtype=torch.int32)
    acc = torch.empty(T, dtype=torch.float32)
    rc = lib.tclip_match_clusters_host_strided(T, Q, K, _ptr(preds_h), _ptr(nc_h), _ptr(ids_h), _ptr(protos_h), _ptr(y_h),
                                               int(bool(graph_matching)), used, _ptr(new_preds), _ptr(acc))
    _capi.check(rc, "tclip_match_clusters_host_strided")
    return acc, new_preds


def gather_rows(table, idx):
    """table (n,K) cuda f32, idx (m,) int64 -> (m,K) cuda f32 (device-side task construction)."""
    _require_cuda(table, "table")
    table = table.contiguous().float()
    idx = _index_tensor(idx.reshape(-1), table.shape[0], table.device, "idx")
    out = torch.empty(idx.numel(), table.shape[1], device=table.device)
    with torch.cuda.device(table.device):
        rc = _capi.lib().tclip_gather_rows(_ptr(table), table.shape[0], table.shape[1], _ptr(idx), idx.numel(), _ptr(out), _stream())
    _capi.check(rc, "tclip_gather_rows")
    return out


def debug_set_probe_chunks(chunks=-1):
    """Test hook: run the dead rows' limit-cycle probe after the first `chunks` chunks (0 = never,
    negative = default).  Results do not depend on it."""
    _capi.check(_capi.lib().tclip_debug_set_probe_chunks(int(chunks)), "tclip_debug_set_probe_chunks")


def debug_set_dead_head(iterations=-1):
    """Test hook: MM iterations a freshly dead row runs before the early limit-cycle probe (0 = no early probe: the whole first
    chunk, then the probe; negative = default).  Results do not depend on it."""
    _capi.check(_capi.lib().tclip_debug_set_dead_head(int(iterations)), "tclip_debug_set_dead_head")


def debug_set_rowset_min_rows(rows=-1):
    """Test hook: 0 forces the 32-lanes-per-row layout of the MM kernels for every row length (short rows
    normally use 8 or 16 lanes per row), negative restores the default rule.  Results do not depend on it."""
    _capi.check(_capi.lib().tclip_debug_set_rowset_min_rows(int(rows)), "tclip_debug_set_rowset_min_rows")


def profile_enable(on=True):
    _capi.check(_capi.lib().tclip_profile_enable(int(bool(on))), "tclip_profile_enable")


def profile_collect():
    """(mm_busy_ms, mm_launch_ms_sum, mm_launches, element_updates) since the last call;
    synchronises the device."""
    busy, total, n, upd = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _capi.check(_capi.lib().tclip_profile_collect(ctypes.byref(busy), ctypes.byref(total), ctypes.byref(n),
                                                  ctypes.byref(upd)), "tclip_profile_collect")
    return busy.value, total.value, n.value, upd.value


def profile_last_kernels():
    """{'k_mm_live': (busy_ms, launch_ms_sum, launches, element_updates), 'k_mm_split': (...)} of the last profile_collect()"""
    busy, total = (ctypes.c_double * 2)(), (ctypes.c_double * 2)()
    n, upd = (ctypes.c_int64 * 2)(), (ctypes.c_int64 * 2)()
    _capi.check(_capi.lib().tclip_profile_last_kernels(busy, total, n, upd), "tclip_profile_last_kernels")
    return {name: (busy[i], total[i], n[i], upd[i]) for i, name in enumerate(("k_mm_live", "k_mm_split"))}


def profile_last_split_sorts():
    """(wavefront-iterations k_mm_split ran, full placements among them) of the last profile_collect(): the kernel keeps
    the placement of its elements in the class queues across MM iterations and sorts anew only when one has left its class"""
    it, so = ctypes.c_int64(0), ctypes.c_int64(0)
    _capi.check(_capi.lib().tclip_profile_last_split_sorts(ctypes.byref(it), ctypes.byref(so)), "tclip_profile_last_split_sorts")
    return it.value, so.value
