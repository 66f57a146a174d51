"""ctypes binding of libtclip.so (C ABI declared in include/tclip.h).

There is deliberately no fallback: if the HIP library is missing or a call fails, this raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtclip.so")


class Problem(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "n_batches", "tasks_per_batch", "n_query", "n_class", "n_support", "iters", "iter_mm", "lambd", "hard")]


class TaskSource(ctypes.Structure):                 # struct tclip_task_source
    _fields_ = [(n, ctypes.c_void_p) for n in ("table_q", "q_idx", "table_s", "s_idx", "cols")]


class TimParams(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_double), ("temp", ctypes.c_float), ("alpha_value", ctypes.c_float),
                ("loss_weights", ctypes.c_float * 3), ("entropies", ctypes.c_int32 * 3)]


_P = ctypes.c_void_p
_SIGNATURES = {
    "tclip_abi_version": (ctypes.c_int, []),
    "tclip_last_error": (ctypes.c_char_p, []),
    "tclip_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Problem)]),
    "tclip_em_dirichlet_run": (ctypes.c_int, [ctypes.POINTER(Problem)] + [_P] * 10 + [ctypes.c_size_t, _P]),
    "tclip_em_dirichlet_run_tasks": (ctypes.c_int, [ctypes.POINTER(Problem), ctypes.POINTER(TaskSource)] + [_P] * 8 + [ctypes.c_size_t, _P]),
    "tclip_prototype_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32] * 3),
    "tclip_cluster_prototypes": (ctypes.c_int, [ctypes.c_int32] * 3 + [_P] * 6 + [ctypes.c_size_t, _P]),
    "tclip_match_clusters_host": (ctypes.c_int, [ctypes.c_int32] * 3 + [_P] * 5 + [ctypes.c_int32, _P, _P]),
    "tclip_match_clusters_host_strided": (ctypes.c_int, [ctypes.c_int32] * 3 + [_P] * 5 + [ctypes.c_int32, ctypes.c_int32, _P, _P]),
    "tclip_host_threads": (ctypes.c_int, []),
    "tclip_gather_rows": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int32, _P, ctypes.c_int64, _P, _P]),
    "tclip_soft_kmeans_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Problem)]),
    "tclip_soft_kmeans_run": (ctypes.c_int, [ctypes.POINTER(Problem), _P, ctypes.c_float, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "tclip_em_gaussian_run": (ctypes.c_int, [ctypes.POINTER(Problem), _P, ctypes.c_float, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "tclip_em_gaussian_cov_run": (ctypes.c_int, [ctypes.POINTER(Problem), _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "tclip_paddle_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Problem)]),
    "tclip_paddle_run": (ctypes.c_int, [ctypes.POINTER(Problem), _P, _P, _P, ctypes.c_float, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "tclip_alpha_tim_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Problem)]),
    "tclip_alpha_tim_run": (ctypes.c_int, [ctypes.POINTER(Problem), ctypes.POINTER(TimParams)] + [_P] * 8 + [ctypes.c_size_t, _P]),
    "tclip_laplacian_shot_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Problem)]),
    "tclip_laplacian_shot_run": (ctypes.c_int, [ctypes.POINTER(Problem), _P, _P, _P, ctypes.c_int32, ctypes.c_double, ctypes.c_int32,
                                                _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "tclip_bdcspn_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Problem)]),
    "tclip_bdcspn_run": (ctypes.c_int, [ctypes.POINTER(Problem), _P, _P, _P, ctypes.c_float, ctypes.c_int32, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "tclip_hard_kmeans_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Problem)]),
    "tclip_hard_kmeans_run": (ctypes.c_int, [ctypes.POINTER(Problem), _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "tclip_kl_kmeans_run": (ctypes.c_int, [ctypes.POINTER(Problem), _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "tclip_argmax_rows": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int32, _P, _P]),
    "tclip_probability_features": (ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _P, _P]),
    "tclip_selftest_primitives": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64)]),
    "tclip_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "tclip_debug_set_probe_chunks": (ctypes.c_int, [ctypes.c_int32]),
    "tclip_debug_set_dead_head": (ctypes.c_int, [ctypes.c_int32]),
    "tclip_debug_set_rowset_min_rows": (ctypes.c_int, [ctypes.c_int32]),
    "tclip_debug_set_mm_split": (ctypes.c_int, [ctypes.c_int32]),
    "tclip_debug_set_split_keep_placement": (ctypes.c_int, [ctypes.c_int32]),
    "tclip_debug_set_fixed_k_kernels": (ctypes.c_int, [ctypes.c_int32]),
    "tclip_debug_set_kmeans_tile": (ctypes.c_int, [ctypes.c_int32]),
    "tclip_profile_last_kernels": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                                  ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "tclip_check_task_indices": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int64, _P, ctypes.c_int64, ctypes.c_int32, _P]),
    "tclip_profile_last_split_sorts": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "tclip_profile_collect": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
}
EXPORTS = tuple(_SIGNATURES)
_lib = None


def lib():
    global _lib
    if _lib is None:
        # TCLIP_LIB: another build of the same library (tuning variants from scripts/build_variant.sh); never a fallback
        path = os.environ.get("TCLIP_LIB") or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python transductive-clip_amd/build.py` "
                "(there is no CPU or PyTorch fallback for the EM-Dirichlet path)")
        # torch must be loaded first: libtclip.so needs libamdhip64.so.7 and has to bind to the ONE
        # HIP runtime of the process, the copy bundled with torch (same SONAME).  Loaded the other
        # way round, /opt/rocm's copy and torch's copy would both be live and streams, events and
        # synchronisation would no longer be shared.
        import torch  # noqa: F401
        hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip_rt):
            ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
        l = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.tclip_abi_version() != 5:
            raise RuntimeError("libtclip.so ABI version mismatch")
        _lib = l
    return _lib


def _strip_comments(text):
    """C / C++ source without comments and without blank or all-space lines (string and character literals kept intact)"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":                       # literal: copy to its closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    return "\n".join(ln.rstrip() for ln in "".join(out).split("\n") if ln.strip())


def source_digest():
    """sha1 over the kernel sources (csrc/*.hip, *.h, *.inc, *.cpp, in name order) with comments and blank lines removed:
    what a committed profile was taken on.  profiles/pmc_current.json stores it and bench.py reports the PMC-derived fields
    only while it matches the tree; editing a comment does not invalidate a measurement, editing code does."""
    import hashlib
    csrc = os.path.join(os.path.dirname(_HERE), "csrc")
    h = hashlib.sha1()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".inc", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(csrc, name), "r", encoding="utf-8", errors="replace") as f:
                h.update(_strip_comments(f.read()).encode())
    return h.hexdigest()


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}: {lib().tclip_last_error().decode()}")
