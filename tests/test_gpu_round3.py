"""GPU (round 3): the stop-test path the headline bench runs.

The reference's MM stop test is ONE norm over the whole (N,K,K) batch tensor (em_dirichlet.py:169-175).  For batches of
more than 16 384 (task, class) rows the engine sums the rows' fp64 terms in two stages: 64 slices per batch by
`k_mm_decide_partial`, then the slices in order by `k_mm_decide` (csrc/tclip_kernels.hip, `two_stage`).  Every batch
of the K = 1000 bench (125 tasks x 1000 classes = 125 000 rows) takes that path; the tests here are the ones that reach it:

* `bigbatch_zs_soft_K100_N170` - a fixture made by RUNNING THE REFERENCE on a 170-task batch (17 000 rows, full
  20 x 1000 schedule, tests/golden/make_golden.py), inputs regenerated from integer draws, outputs as digests + samples;
* K = 397 x 42 tasks (16 674 rows, 32 lanes per row) and K = 1000 x 17 tasks (17 000 rows, one wavefront per row)
  against the C++ oracle, alone and as three batches in one call (three stream groups at K = 1000).
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TWO_STAGE_MIN_ROWS = 16384          # csrc/tclip_kernels.hip: `two_stage = has_check && N * K > 16384`


def _sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_two_stage_stop_test_matches_reference_bigbatch():
    """k_mm_decide_partial + k_mm_decide against the reference itself: every `crit < 1e-11` decision of a 17 000-row batch
    (the MM iteration counts), and through them every bit of alpha, u and v."""
    from helpers import intsynth
    from tclip_amd import engine
    g = np.load(os.path.join(GOLDEN, "bigbatch_zs_soft_K100_N170.npz"))
    K, N, iters = int(g["K"]), int(g["N"]), int(g["iters"])
    assert N * K > TWO_STAGE_MIN_ROWS, "the fixture must reach the two-stage stop test"
    x_q, y_q = intsynth.make_tasks(int(g["seed"]), N, K, 75, boost=int(g["boost"]))
    assert _sha(x_q) == str(g["x_q_sha1"]), "input generator is not reproducible on this host"
    assert np.array_equal(y_q, g["y_q"].reshape(N, 75))
    # the reference's recorded decisions are consistent with its MM counts and none sits where fp64 and fp32 sums could disagree
    st = g["stop_test"].astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        crit = (st[:, :, 0] ** 2) / (st[:, :, 1] ** 2)
    seen = ~np.isnan(crit)
    margin = np.abs(crit[seen].astype(np.float64) / 1e-11 - 1.0)
    assert margin.min() > 1e-4, f"a recorded stop test sits within {margin.min():.1e} of the threshold"
    for i, n_mm in enumerate(g["mm_iters"].tolist()):
        k = int(seen[i].sum())
        stopped = k > 0 and crit[i, k - 1] < np.float32(1e-11)
        assert n_mm == (50 * k + 1 if stopped else int(g["iter_mm"])), (i, n_mm, k)
    assert (g["mm_iters"] < int(g["iter_mm"])).any(), "the fixture should contain an early stop decided by the two-stage sum"

    x = torch.from_numpy(x_q).to(DEV)
    res = engine.run_em_dirichlet(x, n_batches=1, iters=iters, iter_mm=int(g["iter_mm"]), lambd=int(K / 5) * 75, hard=False)
    torch.cuda.synchronize()
    assert np.array_equal(res.mm_iters.cpu().numpy()[0], g["mm_iters"]), "MM iteration counts differ from the reference's"
    assert np.array_equal(res.preds.cpu().numpy(), g["argmax"][-1].astype(np.int32))
    alpha = res.alpha.cpu().numpy()
    rows = g["alpha_rows_idx"]
    sampled = np.stack([alpha[n, rows[n]] for n in range(N)])
    assert np.array_equal(sampled, g["alpha_rows"]), "sampled alpha rows differ"
    a64 = alpha.astype(np.float64)
    np.testing.assert_allclose(a64.sum(-1), g["alpha_rowsum"], rtol=1e-12)
    np.testing.assert_allclose((a64 * a64).sum(-1), g["alpha_rowsumsq"], rtol=1e-12)
    assert _sha(alpha) == str(g["alpha_sha1"]), "alpha differs from the reference's"
    assert _sha(res.u.cpu().numpy()) == str(g["u_sha1"]), "responsibilities differ from the reference's"
    assert np.array_equal(res.v.cpu().numpy(), g["v"])
    assert np.array_equal(res.criterions.cpu().numpy()[0], g["criterions"])
    acc_t, _ = engine.clustering_accuracy(x, res.preds, torch.from_numpy(y_q))
    assert np.array_equal(acc_t.numpy().reshape(-1, 1), g["acc"])


@pytest.mark.parametrize("K,N", [(397, 42), (1000, 17)])
def test_two_stage_stop_test_equals_oracle_and_is_grouping_free(K, N):
    """Batches just over 16 384 rows (iters 2, iter_mm 101: two checkpoints per outer iteration) against the C++ oracle,
    then three such batches in ONE call (K = 1000: 51 000 rows -> three stream groups; K = 397: two) against each batch
    alone: k_mm_decide_partial's slices and the stream split are invisible in the results."""
    from oracle import c_oracle
    from tclip_amd import engine, synth
    assert N * K > TWO_STAGE_MIN_ROWS
    B, kw = 3, dict(iters=2, iter_mm=101, lambd=int(K / 5) * 75, hard=False)
    x_q, _ = synth.make_query_tasks(B * N, K, seed=7000 + K)
    x = x_q.to(DEV)
    full = engine.run_em_dirichlet(x, n_batches=B, **kw)
    torch.cuda.synchronize()
    c_oracle.lib().tclip_oracle_min_stop_margin.restype = __import__("ctypes").c_double
    c_oracle.lib().tclip_oracle_min_stop_margin(1)
    for b in range(B):
        sl = slice(b * N, (b + 1) * N)
        single = engine.run_em_dirichlet(x[sl], n_batches=1, **kw)
        for name in ("alpha", "u", "v", "preds"):
            assert torch.equal(getattr(full, name)[sl], getattr(single, name)), (b, name)
        assert torch.equal(full.mm_iters[b], single.mm_iters[0]) and torch.equal(full.criterions[b], single.criterions[0])
        if b == 0:                                   # the oracle costs N * K^2 * 202 element-updates on the host: one batch
            ref = c_oracle.run(x_q[sl].numpy(), iters=kw["iters"], iter_mm=kw["iter_mm"], lambd=kw["lambd"])
            assert np.array_equal(single.mm_iters[0].cpu().numpy(), ref["mm_iters"])
            assert np.array_equal(single.alpha.cpu().numpy(), ref["alpha"]), "alpha differs from the oracle's"
            assert np.array_equal(single.u.cpu().numpy(), ref["u"]) and np.array_equal(single.v.cpu().numpy(), ref["v"])
            assert np.array_equal(single.criterions[0].cpu().numpy(), ref["criterions"])
            assert np.array_equal(single.preds.cpu().numpy(), ref["argmax"][-1].astype(np.int32))
            # (c) no stop test of this run sits where the engine's two-stage fp64 sum and the oracle's serial one could decide differently
            assert c_oracle.lib().tclip_oracle_min_stop_margin(0) > 1e-6


def _set_split(mode):
    from tclip_amd import _capi
    _capi.check(_capi.lib().tclip_debug_set_mm_split(mode), "tclip_debug_set_mm_split")


@pytest.mark.parametrize("K,N,few,hard", [(5, 6, False, False), (10, 8, False, False), (37, 6, False, True), (100, 12, False, False),
                                           (100, 3, True, False), (196, 4, False, False), (256, 3, False, False), (300, 3, False, False),
                                           (397, 4, False, True), (512, 2, False, False), (620, 2, False, False), (750, 2, False, False),
                                           (880, 2, False, False), (1000, 5, False, False), (1024, 2, True, False)])
def test_class_split_kernel_is_invisible(K, N, few, hard):
    """k_mm_split (every element runs only what its value class a+1 < 2.3 / [2.3, 10) / >= 10 needs, through three dense
    LDS queues) against k_mm_live on the same problems: never / from the first outer iteration on / the default rule
    (from the second on) give identical bits - alpha, u, v, MM counts, criterions - in every lane layout (16, 32 and
    64 lanes per row) and with two batches per call.  (Rows shorter than 65 or of 769 .. 896 elements have no split
    instantiation - too little work per dense pass, too much LDS - and take k_mm_live in every mode.)"""
    from tclip_amd import engine, synth
    B = 2
    iters, iter_mm = (3, 151) if K >= 397 else (4, 230)
    x_q, _ = synth.make_query_tasks(B * N, K, seed=8100 + K, k_eff=(min(4, K) if few else None))
    x_s = y_s = None
    if few:
        x_s, y_s = synth.make_support(B * N, K, 1, seed=8200 + K)
        x_s, y_s = x_s.to(DEV), y_s.squeeze(2).to(DEV)
    kw = dict(n_batches=B, iters=iters, iter_mm=iter_mm, lambd=max(1, int(K / 5)) * 75, hard=hard)
    res = {}
    try:
        for mode in (0, 1, -1):
            _set_split(mode)
            res[mode] = engine.run_em_dirichlet(x_q.to(DEV), x_s, y_s, **kw)
            torch.cuda.synchronize()
    finally:
        _set_split(-1)
    for mode in (1, -1):
        for name in ("alpha", "u", "v", "preds", "mm_iters", "criterions"):
            assert torch.equal(getattr(res[mode], name), getattr(res[0], name)), (mode, name)
    assert torch.isfinite(res[0].alpha).all()


def _mm_launch_names(fn):
    """kernel names of the launches `fn` makes, from torch's profiler (roctracer): which MM kernel a test really ran"""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return {e.name for e in prof.events() if "k_mm_" in e.name}


@pytest.mark.parametrize("K,N,iter_mm", [(100, 4, 1000), (397, 2, 400), (1000, 2, 400)])
def test_class_split_kernel_with_nan_and_stopped_batches(K, N, iter_mm):
    """A NaN feature (the per-wavefront generic path inside k_mm_split: mm_iterate_wave_split's `!__all(in_domain)` branch)
    and a batch that stops early while the other keeps iterating (rows of a stopped batch are skipped by the same
    blocks): identical to k_mm_live.  Row lengths WITH a split instantiation in each lane layout: K = 100 (16 lanes x 7
    registers), 397 (32 x 13), 1000 (64 x 16) - round 3 ran this at K = 40, three registers per lane, below
    TCLIP_SPLIT_MIN_E, where both modes are k_mm_live and the test passed trivially.  The profiler's kernel names
    confirm that mode 1 launched k_mm_split and mode 0 did not."""
    from tclip_amd import engine, synth
    B = 2
    x_q, _ = synth.make_query_tasks(B * N, K, seed=8300 + K)
    # batch 1: nearly flat features - its MM loop never converges within iter_mm, batch 0's first one stops early
    x_q[N:] = torch.softmax(torch.log(x_q[N:]) * 0.02, -1)
    x_bad = x_q.clone()
    x_bad[1, 3, 5] = float("nan")
    kw = dict(n_batches=B, iters=3, iter_mm=iter_mm, lambd=int(K / 5) * 75, hard=False)
    out, names = {}, {}
    try:
        for mode in (0, 1):
            _set_split(mode)
            def go():
                out[mode] = (engine.run_em_dirichlet(x_q.to(DEV), **kw), engine.run_em_dirichlet(x_bad.to(DEV), **kw))
            names[mode] = _mm_launch_names(go)
    finally:
        _set_split(-1)
    assert any("k_mm_split" in n for n in names[1]), names[1]
    assert not any("k_mm_split" in n for n in names[0]), names[0]
    clean = out[0][0].mm_iters.cpu().numpy()
    # some outer iteration in which one batch stops at a checkpoint while the other runs on (measured: batch 0 stops at MM
    # iteration 151 / 201 of the first outer iteration, the flat batch 1 never stops) - in mode 1 that iteration runs k_mm_split
    assert ((clean[0] < iter_mm) & (clean[1] == iter_mm)).any(), f"one batch should stop while the other runs on: {clean.tolist()}"
    assert torch.isnan(out[0][1].alpha[1]).any(), "the NaN feature must reach alpha (the generic path ran)"
    for a, b in zip(out[0], out[1]):
        for name in ("alpha", "u", "v", "preds", "mm_iters"):
            x, y = getattr(a, name), getattr(b, name)
            assert torch.equal(torch.nan_to_num(x.float(), nan=-7.0), torch.nan_to_num(y.float(), nan=-7.0)), name


# dispatch_E's register counts for the 32-lane E-step kernels: k_logits<E, R> drops the bounds tests on registers below the next
# smaller count (K > 32 x that count for every K the instantiation sees), so the first and the last K of every bucket are the cases
@pytest.mark.parametrize("K", [32, 33, 64, 65, 96, 97, 128, 129, 192, 193, 256, 257, 320, 321, 416, 417, 512, 513, 640, 641,
                               768, 769, 896, 897])
def test_e_step_at_the_edges_of_every_row_width_bucket(K):
    """k_row_consts / k_logits at the first and last row length of each instantiation, three tasks (the row list's segments of
    2 or 4 rows straddle task boundaries), against the C++ oracle bit for bit."""
    from oracle import c_oracle
    from tclip_amd import engine, synth
    N = 3
    kw = dict(iters=3, iter_mm=60, lambd=max(1, int(K / 5)) * 75)
    x_q, _ = synth.make_query_tasks(N, K, seed=9100 + K)
    res = engine.run_em_dirichlet(x_q.to(DEV), n_batches=1, hard=False, **kw)
    torch.cuda.synchronize()
    ref = c_oracle.run(x_q.numpy(), **kw)
    assert np.array_equal(res.mm_iters[0].cpu().numpy(), ref["mm_iters"])
    assert np.array_equal(res.alpha.cpu().numpy(), ref["alpha"]), "alpha differs from the oracle's"
    assert np.array_equal(res.u.cpu().numpy(), ref["u"]), "responsibilities differ from the oracle's"
    assert np.array_equal(res.v.cpu().numpy(), ref["v"])
    assert np.array_equal(res.preds.cpu().numpy(), ref["argmax"][-1].astype(np.int32))


def test_integration_md_level2_binding_is_live():
    """The ctypes binding INTEGRATION.md shows a maintainer of the reference (Level 2) is executed AS PRINTED - the code block is
    cut out of the document, only the library path is filled in - bound to a stand-in for the reference's method object, and run on
    a reference fixture: alpha, u, v and the criterions it leaves on the object are the reference's, bit for bit.  Nothing of this
    repository's Python (tclip_amd) is involved: the C ABI alone carries the path."""
    import re
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    section = text[text.index("## Level 2"):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    lib_path = os.path.join(root, "transductive-clip_amd", "tclip_amd", "libtclip.so")
    assert 'ctypes.CDLL("libtclip.so")' in code
    ns = {}
    exec(compile(code.replace('ctypes.CDLL("libtclip.so")', f"ctypes.CDLL({lib_path!r})"), "INTEGRATION.md:level2", "exec"), ns)
    g = np.load(os.path.join(GOLDEN, "zs_soft_K47_N3.npz"))
    K = int(g["K"])
    method = types.SimpleNamespace(iter=int(g["iters"]), iter_mm=int(g["iter_mm"]), lambd=int(K / 5) * 75, device=DEV)
    query = torch.from_numpy(g["x_q"]).to(DEV)
    ns["run_method"](method, query, torch.from_numpy(g["y_q"]).to(DEV))
    assert np.array_equal(method.alpha.cpu().numpy(), g["alpha"])
    assert np.array_equal(method.u.cpu().numpy(), g["u"])
    assert np.array_equal(method.v.cpu().numpy(), g["v"])
    assert np.array_equal(np.array([float(c) for c in method.criterions], np.float32), g["criterions"])
