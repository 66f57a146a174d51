"""CPU: the C-ABI library loads, exports every symbol include/tclip.h declares, validates its
arguments without touching a GPU, and its host-side cluster matching equals scipy's."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

from conftest import ROOT
from tclip_amd import _capi


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "tclip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tclip_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _capi.lib()
    names = _declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tclip.h but not exported"
    assert set(_capi.EXPORTS) == set(names)
    assert lib.tclip_abi_version() == 5


def test_workspace_size_and_argument_checks():
    lib = _capi.lib()
    p = _capi.Problem(10, 100, 75, 100, 0, 20, 1000, 1500, 0)
    ws = lib.tclip_workspace_bytes(ctypes.byref(p))
    # logz + logit0 (2 x T*Q*K) + y, alpha_old, beta_dead (3 x T*K*K) fp32 dominate
    assert ws >= 4 * (2 * 1000 * 75 * 100 + 3 * 1000 * 100 * 100)
    bad = _capi.Problem(1, 1, 75, 2000, 0, 20, 1000, 1, 0)
    assert lib.tclip_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"n_class" in lib.tclip_last_error()
    # null pointers are rejected before any HIP call
    rc = lib.tclip_em_dirichlet_run(ctypes.byref(p), *([None] * 10), 0, None)
    assert rc == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "libtclip.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _capi.lib()


def _match(preds, protos_full, y, K, graph=True):
    lib = _capi.lib()
    T, Q = preds.shape
    cmax = min(Q, K)
    ncl = np.zeros(T, np.int32)
    ids = -np.ones((T, cmax), np.int32)
    pr = np.zeros((T, cmax, K), np.float32)
    for t in range(T):
        order = []
        for c in preds[t]:
            if c not in order:
                order.append(int(c))
        ncl[t] = len(order)
        ids[t, :len(order)] = order
        pr[t, :len(order)] = protos_full[t, order]
    newp = np.empty((T, Q), np.int32)
    acc = np.empty(T, np.float32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    preds32 = np.ascontiguousarray(preds, np.int32)
    y64 = np.ascontiguousarray(y, np.int64)
    rc = lib.tclip_match_clusters_host(T, Q, K, P(preds32), P(ncl), P(ids), P(pr), P(y64), int(graph), P(newp), P(acc))
    assert rc == 0
    return newp, acc, ids, ncl, pr


@pytest.mark.parametrize("K,Q,seed", [(10, 75, 0), (37, 75, 1), (100, 75, 2), (5, 75, 3), (200, 40, 4)])
def test_cluster_matching_equals_scipy(K, Q, seed):
    rng = np.random.default_rng(seed)
    T = 12
    preds = rng.integers(0, K, size=(T, Q))
    preds[0, :] = 3 % K                                  # a single cluster
    preds[1, :] = np.arange(Q) % K                       # as many clusters as possible
    protos = rng.random((T, K, K)).astype(np.float32)
    protos[2] = 0.0                                      # all-zero prototypes: pure tie-breaking
    protos[3, :, : K // 2] = protos[3, :, K // 2: 2 * (K // 2)]   # exact ties between columns
    y = rng.integers(0, K, size=(T, Q))
    newp, acc, ids, ncl, pr = _match(preds, protos, y, K, graph=True)
    for t in range(T):
        C = ncl[t]
        cost = -pr[t, :C].astype(np.float64)
        _, cols = linear_sum_assignment(cost, maximize=False)
        lut = {int(ids[t, i]): int(cols[i]) for i in range(C)}
        want = np.array([lut[int(c)] for c in preds[t]], np.int32)
        assert np.array_equal(newp[t], want), f"task {t}"
        assert acc[t] == np.float32((want == y[t]).sum()) / np.float32(Q)
    newp2, _, _, _, _ = _match(preds, protos, y, K, graph=False)
    for t in range(T):
        want = protos[t].argmax(-1)[preds[t]]
        assert np.array_equal(newp2[t], want.astype(np.int32))


def test_method_classes_keep_the_reference_contract():
    from src.methods.few_shot.em_dirichlet import EM_DIRICHLET as FS
    from src.methods.few_shot.hard_em_dirichlet import HARD_EM_DIRICHLET as FSH
    from src.methods.zero_shot.em_dirichlet import EM_DIRICHLET as ZS
    from src.methods.zero_shot.hard_em_dirichlet import HARD_EM_DIRICHLET as ZSH
    from src.utils import CfgNode
    a = CfgNode(iter=2, iter_mm=100, num_classes_test=20, n_class=20, n_query=75, k_eff=4, T=30,
                use_softmax_feature=True, graph_matching=True)
    for cls, lambd in ((ZS, 4 * 75), (ZSH, 4 * 75), (FS, 5 * 75), (FSH, 5 * 75)):
        m = cls(model=None, device=torch.device("cpu"), log_file=None, args=a)
        assert (m.lambd, m.iter, m.iter_mm, m.eps) == (lambd, 2, 100, 1e-15)
    m = ZS(model=None, device=torch.device("cpu"), log_file=None, args=a)
    task = {"x_q": torch.rand(2, 75, 20), "y_q": torch.zeros(2, 75, 1, dtype=torch.int64)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.run_task(dict(task))
    a.use_softmax_feature = False
    with pytest.raises(ValueError, match="unit simplex"):
        m.run_task(dict(task))


def test_widened_entry_points_check_their_arguments():
    """Every method entry rejects null pointers and a support size that does not fit the method
    before any HIP call (so this runs without a GPU), with a message in tclip_last_error()."""
    lib = _capi.lib()
    zs = _capi.Problem(1, 4, 75, 10, 0, 5, 1, 150, 0)
    fs = _capi.Problem(1, 4, 75, 10, 20, 5, 1, 150, 0)
    P = ctypes.c_void_p(4096)          # never dereferenced: the checks fail first
    none = None
    f = ctypes.c_float(30.0)
    assert lib.tclip_soft_kmeans_run(ctypes.byref(zs), none, f, none, none, none, none, 0, none) != 0
    assert lib.tclip_em_gaussian_run(ctypes.byref(zs), none, f, none, none, none, none, none, 0, none) != 0
    assert lib.tclip_em_gaussian_cov_run(ctypes.byref(zs), none, none, none, none, none, none, none, 0, none) != 0
    assert lib.tclip_hard_kmeans_run(ctypes.byref(zs), none, none, none, none, none, none, 0, none) != 0
    assert lib.tclip_kl_kmeans_run(ctypes.byref(zs), none, none, none, none, none, none, 0, none) != 0
    assert lib.tclip_paddle_run(ctypes.byref(fs), none, none, none, f, none, none, none, none, none, 0, none) != 0
    assert lib.tclip_bdcspn_run(ctypes.byref(fs), none, none, none, f, 1, none, none, none, none, 0, none) != 0
    assert lib.tclip_argmax_rows(none, 10, 5, none, none) != 0
    # zero-shot methods refuse a support set, few-shot methods need one
    assert lib.tclip_em_gaussian_cov_run(ctypes.byref(fs), P, P, P, P, P, P, P, 1 << 30, none) != 0
    assert b"zero-shot" in lib.tclip_last_error()
    assert lib.tclip_kl_kmeans_run(ctypes.byref(fs), P, P, P, P, P, P, 1 << 30, none) != 0
    assert b"zero-shot" in lib.tclip_last_error()
    assert lib.tclip_bdcspn_run(ctypes.byref(zs), P, P, P, f, 1, P, P, P, P, 1 << 30, none) != 0
    assert b"few-shot" in lib.tclip_last_error()
    assert lib.tclip_paddle_run(ctypes.byref(zs), P, P, P, f, P, P, P, P, P, 1 << 30, none) != 0
    assert b"few-shot" in lib.tclip_last_error()
    assert lib.tclip_bdcspn_run(ctypes.byref(fs), P, P, P, f, 7, P, P, P, P, 1 << 30, none) != 0
    assert b"norm_type" in lib.tclip_last_error()
    # too small a workspace is refused
    assert lib.tclip_bdcspn_workspace_bytes(ctypes.byref(fs)) > 0
    assert lib.tclip_bdcspn_run(ctypes.byref(fs), P, P, P, f, 1, P, P, P, P, 16, none) != 0
    assert b"workspace" in lib.tclip_last_error()


def test_widened_method_classes_have_no_cpu_path():
    from src.methods.few_shot.bdcspn import BDCSPN
    from src.methods.few_shot.paddle import PADDLE
    from src.methods.zero_shot.em_gaussian import EM_GAUSSIAN
    from src.methods.zero_shot.em_gaussian_cov import EM_GAUSSIAN_COV
    from src.methods.zero_shot.hard_kmeans import HARD_KMEANS
    from src.methods.zero_shot.inductive_clip import CLIP
    from src.methods.zero_shot.kl_kmeans import KL_KMEANS
    from src.methods.zero_shot.soft_kmeans import SOFT_KMEANS
    from src.utils import CfgNode
    zs_task = {"x_q": torch.rand(2, 75, 20).softmax(-1), "y_q": torch.zeros(2, 75, 1, dtype=torch.int64)}
    fs_task = dict(zs_task, x_s=torch.rand(2, 20, 20).softmax(-1), y_s=torch.arange(20).repeat(2, 1).unsqueeze(2))
    for cls in (SOFT_KMEANS, HARD_KMEANS, KL_KMEANS, EM_GAUSSIAN, EM_GAUSSIAN_COV, CLIP):
        a = CfgNode(iter=2, num_classes_test=20, n_class=20, n_query=75, k_eff=4, T=30, use_softmax_feature=True,
                    graph_matching=True)
        m = cls(model=None, device=torch.device("cpu"), log_file=None, args=a)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.run_task(dict(zs_task))
    for cls in (PADDLE, BDCSPN):
        a = CfgNode(iter=2, num_classes_test=20, n_class=20, n_query=75, k_eff=4, T=30, use_softmax_feature=True,
                    graph_matching=True, lambd=1.0, norm_type="L2N", temp=30.0)
        m = cls(model=None, device=torch.device("cpu"), log_file=None, args=a)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.run_task(dict(fs_task), shot=1)


def test_source_digest_ignores_comments_not_code(tmp_path, monkeypatch):
    """profiles/pmc_current.json is tied to the kernel sources by _capi.source_digest(): a comment or blank line may change,
    a token may not; and the committed PMC entries describe the tree as it is (bench.py reports their traffic / clock)."""
    import json
    import shutil
    assert _capi._strip_comments('int a = 1; // c\n/* b\n c */ const char* s = "// no /* no */";\n\n   \n') == \
        'int a = 1;\n const char* s = "// no /* no */";'
    csrc = os.path.join(os.path.dirname(_capi._HERE), "csrc")
    fake = tmp_path / "pkg"
    shutil.copytree(csrc, fake / "csrc")
    (fake / "tclip_amd").mkdir()
    monkeypatch.setattr(_capi, "_HERE", str(fake / "tclip_amd"))
    base = _capi.source_digest()
    f = fake / "csrc" / "tclip_pk.h"
    text = f.read_text()
    f.write_text("// a new remark\n\n" + text.replace("// RN(1/x), see rcp_rn_f32", "// RN(1/x)  (reworded)"))
    assert _capi.source_digest() == base
    f.write_text(text.replace("pk_fma(e, r, r)", "pk_fma(e, r, e)"))
    assert _capi.source_digest() != base
    monkeypatch.undo()
    # bench.py reports the committed PMC figures only for the kernel sources they were taken on
    import sys
    sys.path.insert(0, ROOT)
    import bench
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_current.json")))
    assert set(pmc) == {"k1000", "k100", "k397_hard", "fs_k1000"}
    monkeypatch.setattr(bench._capi, "source_digest", lambda: pmc["k1000"]["csrc_sha1"])
    fresh = bench.roofline_of((10.0, 10.0, 4, 10 ** 9), 1, 1000, "k1000")
    assert fresh["traffic"] == pmc["k1000"]["traffic_bytes_per_launch"] and "pmc_stale" not in fresh and fresh["measured_clock_ghz"] > 1.0
    monkeypatch.setattr(bench._capi, "source_digest", lambda: "0" * 40)
    stale = bench.roofline_of((10.0, 10.0, 4, 10 ** 9), 1, 1000, "k1000")
    assert stale["traffic"] is None and "measured_clock_ghz" not in stale and stale["pmc_stale"]["file"] == pmc["k1000"]["file"]


def test_bench_stdout_line_stays_small():
    """The driver reads bench.py's ONE stdout line through a bounded buffer: round 5's 20 kB line (three secondary workloads
    with their derivations) came back as `parsed: null`.  The line is now a digest of the full record (which goes to
    gpurun_out/bench_full.json and stderr): every field the contract names, roofline with traffic and the per-launch figures,
    cpu_baseline with both samples - in a few kB whatever the secondaries carry."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))          # a complete record of the old format
    text = bench.short_line(full, "gpurun_out/bench_full.json")
    assert len(text) < 4096 and "\n" not in text
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert len(d["config"]["workload"]) > 20 and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert set(d["secondary"]) >= {"k100", "k397_hard", "fs_k1000"} and d["config"]["mm_iters_batch0"] == "501,1000x19"
    # a record that would still be too long sheds its optional parts instead of growing the line
    full["config"]["path"] = "x" * 8000
    assert "secondary" not in json.loads(bench.short_line(full, "f"))
    assert bench._rle([201, 251, 51, 51, 51]) == "201,251,51x3"
