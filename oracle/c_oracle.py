"""TEST INFRASTRUCTURE: ctypes wrapper of oracle/_tclip_oracle.so (the C++ CPU restatement)."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_P = ctypes.c_void_p
_lib = None


def lib():
    global _lib
    if _lib is None:
        from oracle import build as _b
        path = _b.build()[0]                       # rebuilt whenever the source or the math header changed
        l = ctypes.CDLL(path)
        for f in ("tclip_oracle_sum_inner", "tclip_oracle_sum_outer", "tclip_oracle_sum_reduce_all"):
            getattr(l, f).restype = ctypes.c_float
        _lib = l
    return _lib


def _ptr(a):
    return a.ctypes.data_as(_P) if a is not None else None


def run(x_q, x_s=None, y_s=None, *, iters, iter_mm=1000, lambd, hard=False):
    """numpy in, dict of numpy out; one reference batch."""
    z = np.ascontiguousarray(x_q, np.float32)
    N, Q, K = z.shape
    few = x_s is not None
    xs = np.ascontiguousarray(x_s, np.float32) if few else None
    ys = np.ascontiguousarray(np.asarray(y_s).reshape(N, -1), np.int64) if few else None
    S = xs.shape[1] if few else 0
    u = np.empty((N, Q, K), np.float32)
    v = np.empty((N, K), np.float32)
    alpha = np.empty((N, K, K), np.float32)
    crit = np.empty(iters, np.float32)
    mm = np.empty(iters, np.int32)
    am = np.empty((iters, N, Q), np.int16)
    lib().tclip_oracle_run(_ptr(z), _ptr(xs), _ptr(ys), N, Q, K, S, iters, iter_mm, int(lambd), int(bool(hard)),
                           _ptr(u), _ptr(v), _ptr(alpha), _ptr(crit), _ptr(mm), _ptr(am))
    return {"u": u, "v": v, "alpha": alpha, "criterions": crit, "mm_iters": mm, "argmax": am}


def run_soft_kmeans(x_q, *, iters, temperature):
    z = np.ascontiguousarray(x_q, np.float32)
    N, Q, K = z.shape
    u = np.empty((N, Q, K), np.float32)
    w = np.empty((N, K, K), np.float32)
    am = np.empty((iters, N, Q), np.int16)
    lib().tclip_oracle_soft_kmeans(_ptr(z), N, Q, K, iters, int(temperature), _ptr(u), _ptr(w), _ptr(am))
    return {"u": u, "w": w, "argmax": am}
