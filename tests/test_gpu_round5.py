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
