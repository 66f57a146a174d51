"""GPU (round 5): the benchmarked shapes pinned to the reference itself, end to end.

* `bigbatch_zs_soft_K1000_N17` - made by RUNNING THE REFERENCE on a 17-task batch at K = 1000 (17 000 rows, full 20 x 1000
  schedule; tests/golden/make_golden.py): the engine runs it on k_mm_live<16,4,false,1,64,1000> (first outer iteration),
  k_mm_split<16,64,1000> (the rest) and the two-stage stop test k_mm_decide_partial - the exact kernel combination every
  batch of the K = 1000 headline runs.  Until round 4 that combination was compared with the C++ oracle only (which shares
  csrc/tclip_math.h with the product); the reference's own K = 1000 fixtures had 1 and 3 tasks (single-stage stop test).
* `lean_fs_soft_K1000_N1_s4` - ONE few-shot task at K = 1000 with 4 shots, S = 4000 support rows: configs[4]'s support size, where
  the reference's (1,S,K,K) temporary is 16 GB.  K = 1000 few-shot had been pinned to the reference at 1 shot only.
(`eval_zs_soft_K100`, the evaluator fixture at a BASELINE class count, is a case of
tests/test_gpu_engine_properties.py::test_task_batch_loop_matches_reference.)
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from test_gpu_round4 import _check_bigbatch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _kernels_of(fn):
    """names of the kernels `fn` launched (torch profiler)"""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return {e.key for e in prof.key_averages()}


def test_headline_kernel_combination_matches_reference_k1000_n17():
    """17 tasks x K = 1000 = 17 000 rows, 20 x 1000, against the reference's own alpha (every bit: SHA-1), u, v, criterions,
    MM counts, sampled rows and accuracies; every recorded stop decision is checked for its margin (_check_bigbatch); and the
    kernels that ran are the headline's."""
    from helpers import intsynth
    from tclip_amd import engine
    g = _check_bigbatch("bigbatch_zs_soft_K1000_N17", hard=False, few=False)
    assert int(g["K"]) == 1000 and int(g["N"]) == 17 and int(g["iters"]) == 20
    assert g["mm_iters"].tolist() == [101] + [1000] * 19           # an early stop decided by the two-stage sum, then 19 x 19 decisions not to
    x_q, _ = intsynth.make_tasks(int(g["seed"]), 17, 1000, 75, boost=int(g["boost"]))
    x = torch.from_numpy(x_q).to(DEV)
    names = _kernels_of(lambda: engine.run_em_dirichlet(x, n_batches=1, iters=2, iter_mm=120, lambd=200 * 75))
    joined = " ".join(names)
    assert "k_mm_live<16, 4, false, 1, 64, 1000>" in joined and "k_mm_split<16, 64, 1000>" in joined and "k_mm_decide_partial" in joined, names


def test_few_shot_k1000_four_shots_matches_reference():
    """one task, K = 1000, S = 4000 support rows (configs[4]'s support size) against the reference's own run"""
    g = _check_bigbatch("lean_fs_soft_K1000_N1_s4", hard=False, few=True, two_stage=False)
    assert int(g["K"]) == 1000 and int(g["shots"]) == 4
    assert g["mm_iters"].tolist() == [151, 151] + [51] * 18                 # an early stop in EVERY outer iteration
    assert np.array_equal(np.asarray(g["u"]).shape, (1, 75, 1000))


@pytest.mark.parametrize("K,B,N,hard,shots", [(100, 2, 30, False, 0), (397, 1, 20, True, 0), (1000, 1, 6, False, 0), (1000, 1, 3, False, 1),
                                             (80, 2, 25, False, 2), (300, 1, 12, True, 0), (640, 1, 8, False, 0)])
def test_kept_placement_is_invisible(K, B, N, hard, shots):
    """k_mm_split keeps the placement of its elements in the three class queues from one MM iteration to the next and sorts anew
    only when a dense pass meets an entry that has left its class (round 5).  With the hook that makes it sort in EVERY iteration
    (what rounds 3-4 did) the results must be the same bits - 16-, 32- and 64-lane layouts, fixed-K and run-time-K kernels, soft /
    hard, zero- / few-shot - and the counters must show the mechanism at work: every iteration sorts with the hook, a small
    fraction without (the first iteration of every launch always does)."""
    from tclip_amd import _capi, engine, synth
    T = B * N
    x_q, _ = synth.make_query_tasks(T, K, seed=500 + K, k_eff=(5 if shots else None))
    x = x_q.to(DEV)
    xs = ys = None
    if shots:
        x_s, y_s = synth.make_support(T, K, shots, seed=501 + K)
        xs, ys = x_s.to(DEV), y_s.squeeze(2).to(DEV)
    kw = dict(n_batches=B, iters=4, iter_mm=230, lambd=max(1, int(K / 5)) * 75, hard=hard)
    out, counts = {}, {}
    try:
        for keep in (1, 0):
            _capi.check(_capi.lib().tclip_debug_set_split_keep_placement(keep), "tclip_debug_set_split_keep_placement")
            engine.profile_enable(True)
            engine.profile_collect()
            out[keep] = engine.run_em_dirichlet(x, xs, ys, **kw)
            engine.profile_collect()
            counts[keep] = engine.profile_last_split_sorts()
            engine.profile_enable(False)
    finally:
        _capi.lib().tclip_debug_set_split_keep_placement(1)
        engine.profile_enable(False)
    for name in ("alpha", "u", "v", "preds", "mm_iters", "criterions"):
        assert torch.equal(getattr(out[1], name), getattr(out[0], name)), name
    (it1, so1), (it0, so0) = counts[1], counts[0]
    assert it1 == it0 > 0, "k_mm_split did not run: the test would prove nothing"
    assert so0 == it0, "with the hook every wavefront-iteration sorts"
    assert 0 < so1 < 0.5 * it1, f"the kept placement is not kept: {so1} sorts in {it1} wavefront-iterations"
