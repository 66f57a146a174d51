"""GPU (round 6): the early limit-cycle probe of freshly dead rows (k_mm_probe_head) is an exact shortcut.

A row that has just died used to run the whole first chunk (51 iterations) before k_mm_probe looked for its limit cycle; it now
runs TCLIP_DEAD_HEAD = 12 iterations and the probe searches from there (further snapshots 8 and 32 iterations on), filling every checkpoint of the
row from the cycle it finds.  The cached stop-test terms decide when a BATCH stops (em_dirichlet.py:169-175), so a wrong entry
shows up as a different MM count - in the outer iteration in which the row dies or in any later one - and from there in alpha."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(x, B, K, iters, iter_mm, hard):
    from tclip_amd import engine
    r = engine.run_em_dirichlet(x, n_batches=B, iters=iters, iter_mm=iter_mm, lambd=max(1, int(K / 5)) * 75, hard=hard)
    torch.cuda.synchronize()
    return r


@pytest.mark.parametrize("K,N,B,iters,iter_mm,hard", [(100, 25, 2, 6, 1000, False), (397, 12, 2, 5, 1000, True), (1000, 6, 2, 5, 1000, False),
                                                      (10, 10, 3, 6, 1000, False), (37, 9, 2, 6, 300, False), (64, 8, 2, 5, 1000, True),
                                                      (129, 5, 2, 5, 120, False), (257, 4, 1, 4, 1000, False), (897, 3, 1, 4, 500, False),
                                                      (100, 7, 1, 4, 60, False), (100, 7, 1, 4, 101, False)])
def test_early_dead_row_probe_is_invisible(K, N, B, iters, iter_mm, hard):
    """default (12 head iterations), other head lengths (4 and 1: most rows are still approaching their cycle when the probe starts
    and are found through its second snapshot or handed back to the old path; 18: the longest allowed), the old path (0) and
    no probe at all: same bits"""
    from tclip_amd import engine, synth
    x_q, _ = synth.make_query_tasks(B * N, K, seed=600 + K)
    x = x_q.cuda()
    runs = {}
    try:
        for head in (-1, 0, 4, 18, 1, 16):
            engine.debug_set_dead_head(head)
            runs[head] = _run(x, B, K, iters, iter_mm, hard)
        engine.debug_set_dead_head(-1)
        engine.debug_set_probe_chunks(0)                     # every dead row iterates its whole schedule
        runs["brute"] = _run(x, B, K, iters, iter_mm, hard)
    finally:
        engine.debug_set_dead_head(-1)
        engine.debug_set_probe_chunks(-1)
    ref = runs["brute"]
    assert int((ref.u.sum(1) <= 1e-15).sum()) > 0 or K <= 10, "the case has no dead rows"
    for name, r in runs.items():
        assert torch.equal(r.mm_iters, ref.mm_iters), (name, r.mm_iters.tolist(), ref.mm_iters.tolist())
        assert torch.equal(r.alpha, ref.alpha) and torch.equal(r.u, ref.u) and torch.equal(r.v, ref.v), name
        assert torch.equal(r.criterions, ref.criterions), name
    with pytest.raises(RuntimeError):
        engine.debug_set_dead_head(19)                       # the last snapshot must lie before the first checkpoint


def test_early_probe_takes_the_dead_rows_off_the_first_chunk():
    """what ran: with the early probe the dead-row kernel's share of a K = 1000 call shrinks (12 + period iterations per death
    instead of 51 + period) - the per-kernel times of the torch profiler, same problem, both paths"""
    from torch.profiler import ProfilerActivity, profile
    from tclip_amd import engine, synth
    K, N = 1000, 20
    x_q, _ = synth.make_query_tasks(N, K, seed=77)
    x = x_q.cuda()

    def dead_us(head):
        engine.debug_set_dead_head(head)
        _run(x, 1, K, 3, 1000, False)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            r = _run(x, 1, K, 3, 1000, False)
        tot = {}
        for e in prof.key_averages():
            for tag in ("k_mm_probe_head", "k_mm_probe<", "true"):
                if tag in e.key and "k_mm" in e.key:
                    tot[tag] = tot.get(tag, 0.0) + e.device_time_total
        return tot, r
    try:
        old, r0 = dead_us(0)
        new, r1 = dead_us(-1)
    finally:
        engine.debug_set_dead_head(-1)
    assert torch.equal(r0.alpha, r1.alpha) and torch.equal(r0.mm_iters, r1.mm_iters)
    assert "k_mm_probe_head" in new and "k_mm_probe_head" not in old
    t_old, t_new = sum(old.values()), sum(new.values())
    print(f"dead-row kernels: {t_old:.0f} us (whole first chunk + probe) -> {t_new:.0f} us (12 iterations + early probe): {old} -> {new}")
    assert t_new < 0.8 * t_old


def test_headline_batch_with_and_without_the_round6_shortcuts():
    """bench.py's headline batch (125 tasks x K = 1000 from the bench's own table, iter 20 x iter_mm 1000 - 125 000 rows, the two-stage stop
    test, ~119 000 deaths after the first outer iteration): the round's shortcuts - the early probe of freshly dead rows, the M-step
    statistics over compacted lists of the live classes - switched off one by one give the same bits, MM counts and criterions."""
    from src.eval_zero_shot import Evaluator_zero_shot
    from src.utils import CfgNode
    from tclip_amd import _capi, engine, synth
    import random
    K, N = 1000, 125
    feats, labels = synth.make_feature_table(K, 50, seed=2020)
    cfg = CfgNode(iter=20, iter_mm=1000, num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30, use_softmax_feature=True,
                  graph_matching=True, shots=0, number_tasks=N, batch_size=N, name_method="EM_DIRICHLET", used_test_set="test")
    ev = Evaluator_zero_shot(device=torch.device(DEV), log_file=None, args=cfg)
    random.seed(2020); np.random.seed(2020); torch.manual_seed(2020)
    idx = ev.sample_indices(labels.numpy())
    x = engine.gather_rows(feats.to(DEV), idx.reshape(-1)).view(N, 75, K)
    runs = {}
    try:
        runs["default"] = _run(x, 1, K, 20, 1000, False)
        engine.debug_set_dead_head(0)
        runs["whole first chunk + k_mm_probe"] = _run(x, 1, K, 20, 1000, False)
        engine.debug_set_dead_head(-1)
        _capi.check(_capi.lib().tclip_debug_set_kmeans_tile(0), "tclip_debug_set_kmeans_tile")
        runs["statistics by k_mstats_rows"] = _run(x, 1, K, 20, 1000, False)
    finally:
        engine.debug_set_dead_head(-1)
        _capi.check(_capi.lib().tclip_debug_set_kmeans_tile(-1), "tclip_debug_set_kmeans_tile")
    ref = runs["default"]
    assert ref.mm_iters[0, 0] < 1000 and int(ref.mm_iters[0, 1:].min()) == 1000, ref.mm_iters.tolist()
    dead = int((ref.u.sum(1) <= 1e-15).sum())
    assert dead > 100000, dead                                # the shortcut's rows are the bulk of the batch
    for name, r in runs.items():
        for f in ("alpha", "u", "v", "preds", "mm_iters", "criterions"):
            assert torch.equal(getattr(r, f), getattr(ref, f)), (name, f)
