"""GPU (round 4): every BASELINE shape at FULL batch size and full schedule in the gated suite, and the two-stage stop test
in the two modes that had only been pinned for soft zero-shot.

* configs[3] / the headline: one 125-task K = 1000 batch, 20 x 1000 (125 000 rows: k_mm_live<16,64> in the first outer
  iteration, k_mm_split<16,64> afterwards, k_mm_decide_partial at every checkpoint) through Evaluator_zero_shot and through
  the engine directly;
* configs[2]: one 100-task Hard EM-Dirichlet batch at K = 397, 10 x 1000 (39 700 rows, 32 lanes per row, k_mm_split<13,32>),
  and SOFT_KMEANS on the same tasks;
* configs[4]: 25 few-shot tasks at K = 1000 with S = 4000 support rows each (4-shot), probability front-end included, through
  Evaluator_few_shot - and the evaluator's index-driven path (table rows read through the index tensors, label flip and column
  permutation inside the kernels) against the engine fed with materialised, relabelled (T,S,K) tensors: identical bits;
* the two-stage stop test: the reference fixtures `bigbatch_zs_hard_K397_N42` (HARD, 32-lane layout, 16 674 rows) and
  `bigbatch_fs_soft_K100_N170_s1` (FEW-SHOT, 17 000 rows), made by running the reference (tests/golden/make_golden.py);
  hard K = 397 x 42 and few-shot K = 1000 x 17 against the C++ oracle, alone and three batches per call.
"""
import hashlib
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TWO_STAGE_MIN_ROWS = 16384


def _sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def _seed(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def _mm_pattern_ok(mm, iter_mm):
    """an MM loop either runs out (iter_mm) or breaks at a checkpoint l = 50 k having executed l + 1 iterations"""
    mm = np.asarray(mm)
    return bool((((mm == iter_mm) | ((mm - 1) % 50 == 0)) & (mm >= min(51, iter_mm)) & (mm <= iter_mm)).all())


def _cfg(**kw):
    from src.utils import CfgNode
    base = dict(iter=20, iter_mm=1000, n_query=75, k_eff=5, T=30, use_softmax_feature=True, graph_matching=True, shots=0,
                used_test_set="test", tunable=False)
    base.update(kw)
    base.setdefault("n_class", base["num_classes_test"])
    return CfgNode(base)


# ---------------------------------------------------------------------------------------------------------------------
def test_headline_batch_k1000_full_schedule():
    """bench.py's headline batch: 125 tasks x K = 1000, iter 20 x iter_mm 1000, drawn from the bench's own table."""
    from src.eval_zero_shot import Evaluator_zero_shot
    from tclip_amd import engine, synth
    K, N = 1000, 125
    feats, labels = synth.make_feature_table(K, 50, seed=2020)
    ev = Evaluator_zero_shot(device=torch.device(DEV), log_file=None,
                             args=_cfg(num_classes_test=K, number_tasks=N, batch_size=N, name_method="EM_DIRICHLET"))
    _seed(2020)
    idx = ev.sample_indices(labels.numpy())
    assert tuple(idx.shape) == (1, N, 75) and N * K > TWO_STAGE_MIN_ROWS
    table = feats.to(DEV)
    acc_mean, _ = ev.evaluate_tasks(None, table, labels, indices=idx)
    x = engine.gather_rows(table, idx.reshape(-1)).view(N, 75, K)
    kw = dict(n_batches=1, iters=20, iter_mm=1000, lambd=int(K / 5) * 75, hard=False)
    res = engine.run_em_dirichlet(x, **kw)
    again = engine.run_em_dirichlet(x, **kw)
    torch.cuda.synchronize()
    for name in ("alpha", "u", "v", "preds", "mm_iters", "criterions"):                 # run-to-run identical bits
        assert torch.equal(getattr(res, name), getattr(again, name)), name
    m = ev.last_method
    assert torch.equal(m.alpha, res.alpha) and torch.equal(m.u, res.u) and torch.equal(m.v, res.v)
    mm = res.mm_iters[0].cpu().numpy()
    assert np.array_equal(ev.last_batch_mm_iters[0], mm) and _mm_pattern_ok(mm, 1000), mm.tolist()
    assert np.array_equal(ev.last_batch_criterions[0], res.criterions[0].cpu().numpy())
    assert torch.isfinite(res.alpha).all() and (res.alpha > 0).all()
    assert (res.u >= 0).all() and (res.u.sum(-1) - 1).abs().max() <= 1e-5
    assert torch.equal(res.preds.long(), res.u.argmax(-1))
    assert torch.isfinite(res.criterions).all() and (res.criterions >= 0).all()
    # v is the M-step's function of u (em_dirichlet.py:151): log(mean_q u + eps) + 1, recomputed here in fp64
    v64 = torch.log(res.u.double().mean(1) + 1e-15) + 1
    assert (res.v.double() - v64).abs().max() < 1e-4
    y = labels[idx.reshape(-1)].view(N, 75)
    acc, matched = engine.clustering_accuracy(x, res.preds, y)
    assert np.array_equal(acc.numpy().reshape(N), ev.last_task_accuracies[0])
    assert np.array_equal(matched.cpu().numpy().reshape(N, 75), ev.last_task_predictions[0])
    assert abs(float(acc.mean()) - float(acc_mean)) < 1e-6 and 0.2 < float(acc_mean) <= 1.0     # chance level: 0.001 (the bench's table gives 0.44)


def test_configs2_batch_hard_k397_and_soft_kmeans():
    """BASELINE configs[2]: a 100-task Hard EM-Dirichlet batch at K = 397 (iter 10 x 1000) and SOFT_KMEANS (iter 20, T = 30)
    on the same tasks, through the evaluator and through the engine."""
    from src.eval_zero_shot import Evaluator_zero_shot
    from src.utils import CfgNode
    from tclip_amd import engine, synth
    K, N = 397, 100
    feats, labels = synth.make_feature_table(K, 40, seed=2020)
    cfg = _cfg(num_classes_test=K, number_tasks=N, batch_size=N, name_method="HARD_EM_DIRICHLET", iter=10)
    ev = Evaluator_zero_shot(device=torch.device(DEV), log_file=None, args=cfg)
    _seed(2020)
    idx = ev.sample_indices(labels.numpy())
    table = feats.to(DEV)
    acc_hard, _ = ev.evaluate_tasks(None, table, labels, indices=idx)
    assert N * K > TWO_STAGE_MIN_ROWS
    x = engine.gather_rows(table, idx.reshape(-1)).view(N, 75, K)
    kw = dict(n_batches=1, iters=10, iter_mm=1000, lambd=int(K / 5) * 75, hard=True)
    res = engine.run_em_dirichlet(x, **kw)
    again = engine.run_em_dirichlet(x, **kw)
    torch.cuda.synchronize()
    for name in ("alpha", "u", "v", "preds", "mm_iters", "criterions"):
        assert torch.equal(getattr(res, name), getattr(again, name)), name
    assert torch.equal(ev.last_method.alpha, res.alpha) and torch.equal(ev.last_method.u, res.u)
    mm = res.mm_iters[0].cpu().numpy()
    assert np.array_equal(ev.last_batch_mm_iters[0], mm) and _mm_pattern_ok(mm, 1000), mm.tolist()
    assert ((res.u == 0) | (res.u == 1)).all() and (res.u.sum(-1) == 1).all()           # hard assignment (hard_em_dirichlet.py:255-258)
    assert torch.equal(res.preds.long(), res.u.argmax(-1))
    assert torch.isfinite(res.alpha).all() and (res.alpha > 0).all()
    # with one-hot u the cluster sizes are counts: v = log(count/75 + eps) + 1 exactly as fp32 evaluates it for integers
    counts = res.u.sum(1)
    assert torch.equal(counts, counts.round()) and (counts.sum(-1) == 75).all()
    assert 0.2 < float(acc_hard) <= 1.0                    # chance level: 1/397
    # SOFT_KMEANS on the same tasks
    ev2 = Evaluator_zero_shot(device=torch.device(DEV), log_file=None, args=CfgNode(dict(cfg, name_method="SOFT_KMEANS", iter=20)))
    acc_skm, _ = ev2.evaluate_tasks(None, table, labels, indices=idx)
    u, w, preds = engine.run_soft_kmeans(x, iters=20, temperature=30)
    u2, w2, preds2 = engine.run_soft_kmeans(x, iters=20, temperature=30)
    torch.cuda.synchronize()
    assert torch.equal(u, u2) and torch.equal(w, w2) and torch.equal(preds, preds2)
    assert torch.equal(ev2.last_method.u, u) and torch.equal(ev2.last_method.w, w)
    assert (u >= 0).all() and (u.sum(-1) - 1).abs().max() <= 1e-5 and torch.isfinite(w).all()
    assert torch.equal(preds.long(), u.argmax(-1))
    assert 0.1 < float(acc_skm) <= 1.0                     # chance level: 1/397 (the bench's table gives 0.22)
    # first task against the C++ oracle's SOFT_KMEANS (397 x 397 x 75 x 20 on the host is seconds)
    from oracle import c_oracle
    ref = c_oracle.run_soft_kmeans(x[:1].cpu().numpy(), iters=20, temperature=30)
    one_u, one_w, _ = engine.run_soft_kmeans(x[:1], iters=20, temperature=30)
    assert np.array_equal(one_u.cpu().numpy(), ref["u"]) and np.array_equal(one_w.cpu().numpy(), ref["w"])
    assert torch.equal(one_u[0], u[0]) and torch.equal(one_w[0], w[0])                  # tasks do not interact


def _config5_inputs(n_tasks, K=1000, D=512, shots=4, seed=2024):
    """bench.py's configs[4] recipe: unit text embeddings, embeddings = 6 x class direction + unit noise, 5 support rows
    and 20 query rows per class; softmax(30 cos) front-end on the device."""
    from src.eval_few_shot import Evaluator_few_shot
    from tclip_amd import features
    gen = torch.Generator().manual_seed(seed)
    text = torch.randn(K, D, generator=gen)
    text /= text.norm(dim=-1, keepdim=True)
    lab_s = torch.arange(K).repeat_interleave(5)
    lab_q = torch.arange(K).repeat_interleave(20)
    vis_s = (text[lab_s] * 6.0 + torch.randn(len(lab_s), D, generator=gen)).to(DEV)
    vis_q = (text[lab_q] * 6.0 + torch.randn(len(lab_q), D, generator=gen)).to(DEV)
    cfg = _cfg(num_classes_test=K, number_tasks=n_tasks, batch_size=n_tasks, name_method="EM_DIRICHLET", shots=shots)
    ev = Evaluator_few_shot(device=torch.device(DEV), log_file=None, args=cfg)
    _seed(2020)
    idx = ev.sample_indices(lab_s.numpy(), lab_q.numpy())
    tab_s = features.probability_features(vis_s, text.to(DEV), 30.0)
    tab_q = features.probability_features(vis_q, text.to(DEV), 30.0)
    return ev, tab_s, lab_s, tab_q, lab_q, idx


def test_configs4_batch_few_shot_k1000_s4000():
    """BASELINE configs[4] at its batch size: 25 tasks, K = 1000, S = 4000 (4-shot), 20 x 1000, through Evaluator_few_shot
    (whose engine call reads the tables through the index tensors) and through the engine on materialised tensors that were
    gathered, relabelled and column-permuted with torch exactly as Tasks_Generator_few_shot.get_task does
    (task_generator_few_shot.py:41-52): the same bits."""
    from src.eval_few_shot import relabel_batch
    from tclip_amd import engine
    N, K, S = 25, 1000, 4000
    ev, tab_s, lab_s, tab_q, lab_q, (s_idx, q_idx) = _config5_inputs(N)
    assert tuple(s_idx.shape) == (1, N, S) and tuple(q_idx.shape) == (1, N, 75)
    acc_mean, _ = ev.evaluate_tasks(None, tab_s, lab_s, tab_q, lab_q, indices=(s_idx, q_idx))
    m = ev.last_method
    si, qi = s_idx.reshape(-1), q_idx.reshape(-1)
    x_s = tab_s[si.to(DEV)].view(N, S, K)
    x_q = tab_q[qi.to(DEV)].view(N, 75, K)
    x_s, x_q, y_s, y_q = relabel_batch(x_s, x_q, lab_s[si].view(N, S), lab_q[qi].view(N, 75), True)
    kw = dict(n_batches=1, iters=20, iter_mm=1000, lambd=int(K / 5) * 75, hard=False)
    res = engine.run_em_dirichlet(x_q, x_s, y_s.to(DEV), **kw)
    torch.cuda.synchronize()
    for name in ("alpha", "u", "v", "preds"):
        assert torch.equal(getattr(m, name), getattr(res, name)), name
    mm = res.mm_iters[0].cpu().numpy()
    assert np.array_equal(ev.last_batch_mm_iters[0], mm) and _mm_pattern_ok(mm, 1000), mm.tolist()
    assert np.array_equal(ev.last_batch_criterions[0], res.criterions[0].cpu().numpy())
    assert torch.isfinite(res.alpha).all() and (res.alpha > 0).all()
    assert (res.u >= 0).all() and (res.u.sum(-1) - 1).abs().max() <= 1e-5
    acc = (res.preds.cpu().long() == y_q).float().mean(1)
    assert np.array_equal(acc.numpy(), ev.last_task_accuracies[0]) and abs(float(acc.mean()) - float(acc_mean)) < 1e-6
    assert float(acc_mean) > 0.2
    del x_s
    again, _ = ev.evaluate_tasks(None, tab_s, lab_s, tab_q, lab_q, indices=(s_idx, q_idx))
    assert again == acc_mean and torch.equal(ev.last_method.alpha, res.alpha)           # run-to-run


# ---------------------------------------------------------------------------------------------------------------------
def _check_bigbatch(name, hard, few, two_stage=True):
    """a lean reference fixture (digests + samples; inputs regenerated from integer draws) against the engine"""
    from helpers import intsynth
    from tclip_amd import engine
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K, N, iters, shots = int(g["K"]), int(g["N"]), int(g["iters"]), int(g["shots"])
    assert (N * K > TWO_STAGE_MIN_ROWS) == two_stage, "the fixture must reach the two-stage stop test"
    if few:
        x_q, y_q, x_s, y_s = intsynth.make_tasks(int(g["seed"]), N, K, 75, shots=shots, boost=int(g["boost"]))
        assert _sha(x_s) == str(g["x_s_sha1"]) and np.array_equal(y_s, g["y_s"].reshape(N, -1))
    else:
        x_q, y_q = intsynth.make_tasks(int(g["seed"]), N, K, 75, boost=int(g["boost"]))
    assert _sha(x_q) == str(g["x_q_sha1"]), "input generator is not reproducible on this host"
    assert np.array_equal(y_q, g["y_q"].reshape(N, 75))
    # the reference's recorded decisions: consistent with its MM counts, none where fp64 and fp32 sums could disagree
    st = g["stop_test"].astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        crit = (st[:, :, 0] ** 2) / (st[:, :, 1] ** 2)
    seen = ~np.isnan(st[:, :, 0])
    margin = np.abs(crit[seen].astype(np.float64) / 1e-11 - 1.0)
    assert margin.min() > 1e-4, f"a recorded stop test sits within {margin.min():.1e} of the threshold"
    for i, n_mm in enumerate(g["mm_iters"].tolist()):
        k = int(seen[i].sum())
        stopped = k > 0 and crit[i, k - 1] < np.float32(1e-11)
        assert n_mm == (50 * k + 1 if stopped else int(g["iter_mm"])), (i, n_mm, k)
    x = torch.from_numpy(x_q).to(DEV)
    xs = torch.from_numpy(x_s).to(DEV) if few else None
    ys = torch.from_numpy(y_s).to(DEV) if few else None
    res = engine.run_em_dirichlet(x, xs, ys, n_batches=1, iters=iters, iter_mm=int(g["iter_mm"]), lambd=int(K / 5) * 75, hard=hard)
    torch.cuda.synchronize()
    assert np.array_equal(res.mm_iters.cpu().numpy()[0], g["mm_iters"]), "MM iteration counts differ from the reference's"
    assert np.array_equal(res.preds.cpu().numpy(), g["argmax"][-1].astype(np.int32))
    alpha = res.alpha.cpu().numpy()
    rows = g["alpha_rows_idx"]
    assert np.array_equal(np.stack([alpha[n, rows[n]] for n in range(N)]), g["alpha_rows"]), "sampled alpha rows differ"
    assert _sha(alpha) == str(g["alpha_sha1"]), "alpha differs from the reference's"
    assert _sha(res.u.cpu().numpy()) == str(g["u_sha1"]), "responsibilities differ from the reference's"
    assert np.array_equal(res.v.cpu().numpy(), g["v"])
    assert np.array_equal(res.criterions.cpu().numpy()[0], g["criterions"])
    if few:
        acc = (res.preds.cpu().long() == torch.from_numpy(y_q)).float().mean(1, keepdim=True).numpy()
    else:
        acc_t, _ = engine.clustering_accuracy(x, res.preds, torch.from_numpy(y_q))
        acc = acc_t.numpy().reshape(-1, 1)
    assert np.array_equal(acc, g["acc"])
    return g


def test_two_stage_stop_test_matches_reference_hard_k397():
    """configs[2]'s path (Hard EM-Dirichlet, 32 lanes per row, k_mm_split<13,32> + k_mm_decide_partial) against the
    reference itself: 42 tasks x K = 397 = 16 674 rows, 10 x 1000."""
    _check_bigbatch("bigbatch_zs_hard_K397_N42", hard=True, few=False)


def test_two_stage_stop_test_matches_reference_few_shot():
    """the few-shot mode of the two-stage stop test (no dead rows, support statistics in the M-step) against the reference
    itself: 170 tasks x K = 100, 1 shot = 17 000 rows, 20 x 1000."""
    _check_bigbatch("bigbatch_fs_soft_K100_N170_s1", hard=False, few=True)


@pytest.mark.parametrize("K,N,hard,few", [(397, 42, True, False), (1000, 17, False, True)])
def test_two_stage_stop_test_other_modes_equal_oracle_and_are_grouping_free(K, N, hard, few):
    """round 3's oracle cases ran soft zero-shot only; here HARD at (397, 42) and FEW-SHOT (1 shot) at (1000, 17): a batch
    just over 16 384 rows against the C++ oracle, then three such batches in one call against each batch alone."""
    import ctypes
    from oracle import c_oracle
    from tclip_amd import engine, synth
    assert N * K > TWO_STAGE_MIN_ROWS
    B, kw = 3, dict(iters=2, iter_mm=101, lambd=int(K / 5) * 75, hard=hard)
    x_q, _ = synth.make_query_tasks(B * N, K, seed=7100 + K, k_eff=(5 if few else None))
    x = x_q.to(DEV)
    xs = ys = None
    if few:
        x_s, y_s = synth.make_support(B * N, K, 1, seed=7200 + K)
        xs, ys = x_s.to(DEV), y_s.squeeze(2).to(DEV)
    full = engine.run_em_dirichlet(x, xs, ys, n_batches=B, **kw)
    torch.cuda.synchronize()
    c_oracle.lib().tclip_oracle_min_stop_margin.restype = ctypes.c_double
    c_oracle.lib().tclip_oracle_min_stop_margin(1)
    for b in range(B):
        sl = slice(b * N, (b + 1) * N)
        single = engine.run_em_dirichlet(x[sl], xs[sl] if few else None, ys[sl] if few else None, n_batches=1, **kw)
        for name in ("alpha", "u", "v", "preds"):
            assert torch.equal(getattr(full, name)[sl], getattr(single, name)), (b, name)
        assert torch.equal(full.mm_iters[b], single.mm_iters[0]) and torch.equal(full.criterions[b], single.criterions[0])
        if b == 0:
            ref = c_oracle.run(x_q[sl].numpy(), x_s[sl].numpy() if few else None, y_s[sl].numpy() if few else None, **kw)
            assert np.array_equal(single.mm_iters[0].cpu().numpy(), ref["mm_iters"])
            assert np.array_equal(single.alpha.cpu().numpy(), ref["alpha"]), "alpha differs from the oracle's"
            assert np.array_equal(single.u.cpu().numpy(), ref["u"]) and np.array_equal(single.v.cpu().numpy(), ref["v"])
            assert np.array_equal(single.criterions[0].cpu().numpy(), ref["criterions"])
            assert np.array_equal(single.preds.cpu().numpy(), ref["argmax"][-1].astype(np.int32))
            assert c_oracle.lib().tclip_oracle_min_stop_margin(0) > 1e-6


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,N,B,shots,perm,hard", [(10, 4, 1, 2, True, False), (37, 5, 3, 1, True, True), (100, 6, 2, 4, False, False),
                                                   (37, 6, 2, 0, True, False), (300, 30, 2, 1, True, False), (300, 32, 2, 0, False, False)])
def test_table_fed_entry_equals_materialised_tensors(K, N, B, shots, perm, hard):
    """tclip_em_dirichlet_run_tasks (rows of the feature tables through index tensors, an ARBITRARY per-task column
    permutation) against tclip_em_dirichlet_run on the tensors torch materialises from the same indices: identical bits, in
    few-shot and zero-shot mode, with several batches per call (K = 300: two stream groups, whose index rows are offset)."""
    from tclip_amd import engine, synth
    T = N * B
    gen = torch.Generator().manual_seed(4000 + K + shots)
    tab_q, _ = synth.make_feature_table(K, 12, seed=41)
    q_idx = torch.randint(0, tab_q.shape[0], (T, 75), generator=gen)
    cols = torch.stack([torch.randperm(K, generator=gen) for _ in range(T)]) if perm else None
    tq = tab_q.to(DEV)
    x_q = tq[q_idx.to(DEV)]
    if perm:
        x_q = torch.stack([x_q[t][:, cols[t].to(DEV)] for t in range(T)])
    tab_s = s_idx = y_s = x_s = ts = None
    if shots:
        tab_s, lab_s = synth.make_feature_table(K, 6, seed=42)
        ts = tab_s.to(DEV)
        # `shots` rows of every class, in a shuffled order per task
        s_idx = torch.stack([torch.cat([torch.randperm(6, generator=gen)[:shots] + 6 * k for k in range(K)])[torch.randperm(K * shots, generator=gen)]
                             for _ in range(T)])
        old = lab_s[s_idx]                                             # (T,S) table labels
        if perm:                                                       # new label j stands for table column cols[t, j]
            inv = torch.empty_like(cols)
            inv.scatter_(1, cols, torch.arange(K).repeat(T, 1))
            y_s = torch.gather(inv, 1, old)
        else:
            y_s = old
        x_s = ts[s_idx.to(DEV)]
        if perm:
            x_s = torch.stack([x_s[t][:, cols[t].to(DEV)] for t in range(T)])
    kw = dict(n_batches=B, iters=3, iter_mm=120, lambd=max(1, int(K / 5)) * 75, hard=hard)
    ref = engine.run_em_dirichlet(x_q.contiguous(), x_s.contiguous() if shots else None, y_s.to(DEV) if shots else None, **kw)
    got = engine.run_em_dirichlet_tasks(tq, q_idx, ts, s_idx, y_s, cols, **kw)
    torch.cuda.synchronize()
    for name in ("alpha", "u", "v", "preds", "mm_iters", "criterions"):
        assert torch.equal(getattr(ref, name), getattr(got, name)), name
    with pytest.raises(IndexError):
        engine.run_em_dirichlet_tasks(tq, q_idx + tab_q.shape[0], ts, s_idx, y_s, cols, **kw)
    # (round 5) DEVICE-resident index tensors are range-checked too (tclip_check_task_indices): one bad entry raises, as
    # torch's own `table[idx]` does - until round 4 it was an out-of-bounds read in k_gather_log_features / k_support_stats
    q_dev = q_idx.to(DEV)
    got_dev = engine.run_em_dirichlet_tasks(tq, q_dev, ts, s_idx.to(DEV) if shots else None, y_s, cols.to(DEV) if cols is not None else None, **kw)
    assert torch.equal(got_dev.alpha, got.alpha) and torch.equal(got_dev.u, got.u)
    bad = q_dev.clone()
    bad[T - 1, 3] = tab_q.shape[0]
    with pytest.raises(IndexError):
        engine.run_em_dirichlet_tasks(tq, bad, ts, s_idx, y_s, cols, **kw)
    bad[T - 1, 3] = -1
    with pytest.raises(IndexError):
        engine.run_em_dirichlet_tasks(tq, bad, ts, s_idx, y_s, cols, **kw)
    if shots:
        bad_s = s_idx.to(DEV).clone()
        bad_s[0, 0] = tab_s.shape[0] + 5
        with pytest.raises(IndexError):
            engine.run_em_dirichlet_tasks(tq, q_dev, ts, bad_s, y_s, cols, **kw)
    if cols is not None:
        bad_c = cols.to(DEV).to(torch.int32).clone()
        bad_c[0, 1] = K
        with pytest.raises(IndexError):
            engine.run_em_dirichlet_tasks(tq, q_dev, ts, s_idx, y_s, bad_c, **kw)
    with pytest.raises(IndexError):
        engine.gather_rows(tq, torch.tensor([0, tab_q.shape[0]], device=DEV))
    # (round 6) only an out-of-range VALUE is an IndexError (TCLIP_ERR_INDEX = 4); a bad argument to the check itself stays what
    # it is for every other entry point: TCLIP_ERR_ARG = 1
    from tclip_amd import _capi
    st = torch.cuda.current_stream().cuda_stream
    assert _capi.lib().tclip_check_task_indices(bad.data_ptr(), bad.numel(), tab_q.shape[0], None, 0, 1, st) == 4
    assert _capi.lib().tclip_check_task_indices(q_dev.data_ptr(), q_dev.numel(), tab_q.shape[0], None, 0, 1, st) == 0
    assert _capi.lib().tclip_check_task_indices(None, 5, tab_q.shape[0], None, 0, 1, st) == 1
    assert _capi.lib().tclip_check_task_indices(q_dev.data_ptr(), -1, tab_q.shape[0], None, 0, 1, st) == 1


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [32, 33, 40, 47, 63, 64, 65, 96, 100, 127, 128, 129, 196, 255, 256, 257, 397, 448, 449, 511])
def test_kmeans_tile_kernel_is_invisible(K):
    """k_kmeans_logits_tile (one lane per class on a 64-centroid LDS tile, rows of 32 .. 511 elements) against
    k_kmeans_logits_rows (32 lanes per class) through SOFT_KMEANS, HARD_KMEANS and PADDLE, and k_kl_divergences_tile against
    k_kl_divergences through KL_KMEANS: identical u, centroids and predictions - row lengths with every K mod 32 structure (no / one / three leftover vectors, tails of 0 .. 7, the tile
    edges 64 k and 64 k + 1); the smaller ones also against the C++ oracle."""
    from oracle import c_oracle
    from tclip_amd import _capi, engine, synth
    N = 3
    x_q, _ = synth.make_query_tasks(N, K, seed=6000 + K)
    x_s, y_s = synth.make_support(N, K, 1, seed=6100 + K)
    x, xs, ys = x_q.to(DEV), x_s.to(DEV), y_s.squeeze(2).to(DEV)
    out = {}
    try:
        for mode in (0, -1):
            _capi.check(_capi.lib().tclip_debug_set_kmeans_tile(mode), "tclip_debug_set_kmeans_tile")
            out[mode] = (engine.run_soft_kmeans(x, iters=6, temperature=30), engine.run_hard_kmeans(x, iters=4),
                         engine.run_paddle(x, xs, ys, iters=5, lambd=2.5), engine.run_kl_kmeans(x, iters=4))
            torch.cuda.synchronize()
    finally:
        _capi.lib().tclip_debug_set_kmeans_tile(-1)
    for a, b in zip(out[0], out[-1]):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)
    if K <= 128:
        ref = c_oracle.run_soft_kmeans(x_q.numpy(), iters=6, temperature=30)
        u, w, _ = out[-1][0]
        assert np.array_equal(u.cpu().numpy(), ref["u"]) and np.array_equal(w.cpu().numpy(), ref["w"])


def test_bench_prints_one_json_line_under_rccl():
    """bench.py under the driver's launcher with the nccl (= RCCL) backend, one rank: librccl prints a version banner on the
    process's stdout when its communicator comes up; the script keeps fd 1 on stderr meanwhile, so that stdout carries the
    ONE JSON line of the contract and nothing else - with `rank_devices` (one GPU per rank) on it."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
           "--workload", "k100", "--no-secondary", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[:2000]
    d = json.loads(lines[0])
    assert d["backend"] == "nccl" and d["ranks_seen"] == 1 and d["rank_devices"] == [0]
    assert set(d["roofline"]["per_kernel"]) == {"k_mm_live", "k_mm_split"}
    # a launcher whose world size differs from --gpus is refused
    bad = subprocess.run(cmd[:cmd.index("--gpus") + 1] + ["2"] + cmd[cmd.index("--gpus") + 2:], capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "--gpus 2" in (bad.stderr + bad.stdout)


def _set_fixed_k(on):
    from tclip_amd import _capi
    _capi.check(_capi.lib().tclip_debug_set_fixed_k_kernels(on), "tclip_debug_set_fixed_k_kernels")


@pytest.mark.parametrize("K,N,shots,hard,nan", [(1000, 5, 0, False, False), (1000, 3, 1, False, False), (1000, 3, 0, True, True),
                                                (397, 6, 0, True, False), (397, 4, 2, False, False), (397, 4, 0, False, True),
                                                (100, 12, 0, False, False), (100, 6, 2, True, False), (100, 9, 0, False, True)])
def test_fixed_row_length_kernels_are_invisible(K, N, shots, hard, nan):
    """The MM kernels compiled with the row length as a constant (K = 1000 / 397 / 100, launch_mm) against the run-time-K
    kernels of the same bucket on the same problems: identical bits - alpha, u, v, predictions, MM counts, criterions - zero-
    and few-shot, soft and hard, two batches per call (one of which may stop early), and with a NaN feature (the generic
    per-wavefront path of both kernels).  The profiler's kernel names confirm which kernels ran."""
    from tclip_amd import engine, synth
    from test_gpu_round3 import _mm_launch_names
    B = 2
    x_q, _ = synth.make_query_tasks(B * N, K, seed=9100 + K + shots, k_eff=(5 if shots else None))
    if nan:
        x_q[1, 3, 2] = float("nan")
    x_s = y_s = None
    if shots:
        x_s, y_s = synth.make_support(B * N, K, shots, seed=9200 + K)
        x_s, y_s = x_s.to(DEV), y_s.squeeze(2).to(DEV)
    kw = dict(n_batches=B, iters=3, iter_mm=151, lambd=max(1, int(K / 5)) * 75, hard=hard)
    res, names = {}, {}
    try:
        for on in (0, 1):
            _set_fixed_k(on)

            def run(on=on):
                res[on] = engine.run_em_dirichlet(x_q.to(DEV), x_s, y_s, **kw)
            names[on] = _mm_launch_names(run)
    finally:
        _set_fixed_k(1)
    tag = f", {K}>"
    assert any(tag in n for n in names[1]), names[1]
    assert not any(tag in n for n in names[0]), names[0]
    for name in ("alpha", "u", "v", "preds", "mm_iters", "criterions"):
        a, b = getattr(res[1], name), getattr(res[0], name)
        assert torch.equal(a, b) or (nan and np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)), name
