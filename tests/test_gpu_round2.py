"""GPU (round 2): few-shot through the method classes, the config-5 pipeline at its real shape, world-size
independence of the task-batch loop, a bounded random sweep against the C++ oracle, and digests of the
torch-eager restatement of the reference (made on the fixture host) for the special functions."""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(K, **kw):
    from src.utils import CfgNode
    base = dict(iter=20, iter_mm=1000, num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30,
                use_softmax_feature=True, graph_matching=True, shots=0)
    base.update(kw)
    return CfgNode(**base)


@pytest.mark.parametrize("name", ["fs_soft_K10_N4_s4", "fs_hard_K10_N4_s4", "fs_soft_K100_N4_s4", "fs_hard_K100_N3_s4"])
def test_method_class_drop_in_few_shot(name):
    """The reference-shaped call sequence of eval_few_shot.py:250-257 on the GPU: a new instance per batch,
    run_task(task_dic, shot) with CPU tensors of the reference's shapes (few_shot/em_dirichlet.py:66-91)."""
    from src.methods.few_shot.em_dirichlet import EM_DIRICHLET
    from src.methods.few_shot.hard_em_dirichlet import HARD_EM_DIRICHLET
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    hard, K, shots = str(g["kind"]).endswith("hard"), int(g["K"]), int(g["shots"])
    cls = HARD_EM_DIRICHLET if hard else EM_DIRICHLET
    m = cls(model=None, device=torch.device(DEV), log_file=None, args=_cfg(K, iter=int(g["iters"]), shots=shots))
    task = {"x_s": torch.from_numpy(g["x_s"]), "y_s": torch.from_numpy(g["y_s"]),
            "x_q": torch.from_numpy(g["x_q"]), "y_q": torch.from_numpy(g["y_q"])}
    logs = m.run_task(task_dic=task, shot=shots)
    assert set(logs) == {"timestamps", "criterions", "acc"}
    assert logs["acc"].shape == (int(g["N"]), 1) and logs["criterions"].shape == (int(g["iters"]),)
    assert np.array_equal(logs["acc"], g["acc"])
    assert np.array_equal(m.alpha.cpu().numpy(), g["alpha"])
    assert np.array_equal(m.u.cpu().numpy(), g["u"]) and np.array_equal(m.v.cpu().numpy(), g["v"])
    assert np.array_equal(m.mm_iters[0], g["mm_iters"])
    if hard:
        assert (logs["criterions"] == 0).all()        # few_shot/hard_em_dirichlet.py:233-244 logs zeros
    else:
        assert np.array_equal(logs["criterions"], g["criterions"])
    assert m.lambd == int(K / 5) * 75                  # int(K / k_eff) * n_query, few_shot/em_dirichlet.py:14


def test_config5_pipeline_k1000_4shot():
    """BASELINE configs[4] at its real shape: visual embeddings -> probability features on the device
    (src/utils.py:287-290) -> few-shot tasks drawn by the reference's samplers (K = 1000, 4 shots: S = 4000
    support rows per task, relabelled and column-permuted as task_generator_few_shot.py does) -> few-shot
    EM-Dirichlet.  Short schedule; alpha, u, v and the MM counts against the C++ oracle on the same tensors."""
    from oracle import c_oracle
    from src.eval_few_shot import Evaluator_few_shot, relabel_batch
    from tclip_amd import engine, features
    K, D, shots, N = 1000, 512, 4, 2
    gen = torch.Generator().manual_seed(55)
    text = torch.randn(K, D, generator=gen)
    text /= text.norm(dim=-1, keepdim=True)
    lab_s = torch.arange(K).repeat_interleave(5)
    lab_q = torch.arange(K).repeat_interleave(20)
    vis_s = text[lab_s] * 2.0 + torch.randn(len(lab_s), D, generator=gen)
    vis_q = text[lab_q] * 2.0 + torch.randn(len(lab_q), D, generator=gen)
    tab_s = features.probability_features(vis_s.to(DEV), text.to(DEV), 30.0)
    tab_q = features.probability_features(vis_q.to(DEV), text.to(DEV), 30.0)
    assert (tab_q.sum(-1) - 1).abs().max() < 1e-5
    a = _cfg(K, iter=2, iter_mm=60, shots=shots, number_tasks=N, batch_size=N, name_method="EM_DIRICHLET")
    ev = Evaluator_few_shot(torch.device(DEV), a, None)
    random.seed(7); torch.manual_seed(7); np.random.seed(7)
    s_idx, q_idx = ev.sample_indices(lab_s.numpy(), lab_q.numpy())
    assert s_idx.shape == (1, N, K * shots) and q_idx.shape == (1, N, 75)
    x_s = engine.gather_rows(tab_s, s_idx.reshape(-1)).view(N, K * shots, K)
    x_q = engine.gather_rows(tab_q, q_idx.reshape(-1)).view(N, 75, K)
    x_s, x_q, y_s, y_q = relabel_batch(x_s, x_q, lab_s[s_idx.reshape(-1)].view(N, -1), lab_q[q_idx.reshape(-1)].view(N, -1), True)
    m = ev.get_method_builder(None, torch.device(DEV), a, None)
    m.run_method(support=x_s, query=x_q, y_s=y_s.to(DEV), y_q=y_q.to(DEV))
    ref = c_oracle.run(x_q.cpu().numpy(), x_s.cpu().numpy(), y_s.numpy(), iters=2, iter_mm=60, lambd=int(K / 5) * 75)
    assert np.array_equal(m.mm_iters[0], ref["mm_iters"])
    assert np.array_equal(m.alpha.cpu().numpy(), ref["alpha"])
    assert np.array_equal(m.u.cpu().numpy(), ref["u"]) and np.array_equal(m.v.cpu().numpy(), ref["v"])
    assert np.array_equal(m.preds.cpu().numpy(), ref["argmax"][-1].astype(np.int32))
    # and the same tasks through the evaluator's own loop give the same accuracies
    random.seed(7); torch.manual_seed(7); np.random.seed(7)
    mean_acc, _ = ev.evaluate_tasks(None, tab_s, lab_s, tab_q, lab_q)
    acc = (m.preds.cpu().long() == y_q).float().mean(1).numpy()
    assert abs(float(mean_acc) - float(acc.mean())) < 1e-7


# ---- world-size independence of the task-batch loop (two ranks share cuda:0, gloo for the one gather)
def _rank_main(rank, world, port, name, out):
    import torch.distributed as dist
    from src.eval_few_shot import Evaluator_few_shot
    from src.eval_zero_shot import Evaluator_zero_shot
    from tclip_amd import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"])
    a = _cfg(K, shots=int(g["shots"]), number_tasks=int(g["number_tasks"]), batch_size=int(g["batch_size"]),
             name_method="EM_DIRICHLET", used_test_set="test")
    feats, labels = synth.make_feature_table(K, int(g["rows_per_class"]), seed=int(g["seed"]))
    seed = int(g["seed"])
    random.seed(seed); torch.manual_seed(seed); np.random.seed(seed)
    if str(g["kind"]) == "zs":
        ev = Evaluator_zero_shot(torch.device(DEV), a, None)
        acc, _ = ev.evaluate_tasks(None, feats, labels)
    else:
        fs, ls = synth.make_feature_table(K, int(g["support_rows_per_class"]), seed=seed + 1)
        ev = Evaluator_few_shot(torch.device(DEV), a, None)
        acc, _ = ev.evaluate_tasks(None, fs, ls, feats, labels)
    if rank == 0:
        # everything the one collective carries: accuracies, per-task predictions, per-batch criterions and MM counts
        np.savez(out, acc=ev.last_task_accuracies, preds=ev.last_task_predictions, criterions=ev.last_batch_criterions,
                 mm_iters=ev.last_batch_mm_iters)
        assert abs(float(acc) - float(g["mean_accuracy"])) < 1e-7
    else:
        assert acc is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("eval_zs_soft_K10", 1), ("eval_zs_soft_K10", 2), ("eval_zs_soft_K10", 3),
                                        ("eval_fs_soft_K10", 2)])
def test_task_batch_loop_is_world_size_independent(name, world, tmp_path):
    """evaluate_tasks under 1, 2 and 3 ranks (more ranks than batches included): rank 0 ends up with the
    same per-task accuracies, per-task predictions, per-batch criterions and MM counts, bit for bit, and the
    reference's mean accuracy."""
    import torch.multiprocessing as mp
    outs = []
    for w in (1, world) if world > 1 else (1,):
        out = str(tmp_path / f"gathered_{w}.npz")
        mp.spawn(_rank_main, args=(w, 29600 + (os.getpid() + 7 * w) % 2000, name, out), nprocs=w, join=True)
        outs.append(dict(np.load(out)))
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_batches, N = int(g["number_tasks"]) // int(g["batch_size"]), int(g["batch_size"])
    assert outs[0]["preds"].shape == (n_batches, N, 75) and outs[0]["preds"].dtype == np.int32
    assert outs[0]["criterions"].shape == (n_batches, 20) and outs[0]["mm_iters"].shape == (n_batches, 20)
    assert (outs[0]["mm_iters"] >= 51).all() and (outs[0]["criterions"] > 0).any()
    for o in outs[1:]:
        for k in ("acc", "preds", "criterions", "mm_iters"):
            assert np.array_equal(o[k], outs[0][k]), k


# ---- bounded random sweep (the large campaigns live in scripts/gpu_fuzz.py)
def test_random_problems_equal_the_cpu_oracle():
    """~50 random short-schedule problems (K = 2 .. 300, 1 .. 130 queries, few-shot and hard included,
    1 .. 3 batches per call) against the C++ oracle, bit for bit."""
    from oracle import c_oracle
    from tclip_amd import engine, synth
    rng = random.Random(20261002)
    for case in range(50):
        K = rng.choice([2, 3, 5, 8, 9, 17, 31, 32, 33, 40, 64, 65, 96, 100, 101, 129, 160, 200, 257, 300])
        Q = rng.choice([1, 2, 5, 17, 20, 64, 75, 130])
        few, hard, B = rng.random() < 0.3, rng.random() < 0.4, rng.randint(1, 3)
        iter_mm, iters = rng.choice([30, 51, 60, 101, 120, 151]), rng.randint(2, 3)
        N = max(1, min(4, int(4e7 / (K * K) / (iter_mm * iters * B))))
        lambd = max(1, int(K / 5)) * Q
        x_q, _ = synth.make_query_tasks(B * N, K, seed=3000 + case, n_query=Q, k_eff=(min(3, K) if few else None))
        x_s = y_s = None
        if few:
            x_s, y_s = synth.make_support(B * N, K, rng.randint(1, 3), seed=4000 + case)
        r = engine.run_em_dirichlet(x_q.to(DEV), x_s.to(DEV) if few else None, y_s.squeeze(2).to(DEV) if few else None,
                                    n_batches=B, iters=iters, iter_mm=iter_mm, lambd=lambd, hard=hard)
        for b in range(B):
            sl = slice(b * N, (b + 1) * N)
            ref = c_oracle.run(x_q[sl].numpy(), x_s[sl].numpy() if few else None, y_s[sl].numpy() if few else None,
                               iters=iters, iter_mm=iter_mm, lambd=lambd, hard=hard)
            what = f"case {case}: K={K} Q={Q} N={N} B={B} few={few} hard={hard} iters={iters} iter_mm={iter_mm} batch {b}"
            assert np.array_equal(r.mm_iters[b].cpu().numpy(), ref["mm_iters"]), what
            assert np.array_equal(r.alpha[sl].cpu().numpy(), ref["alpha"]), what
            assert np.array_equal(r.u[sl].cpu().numpy(), ref["u"]), what
            assert np.array_equal(r.v[sl].cpu().numpy(), ref["v"]), what


# ---- digests of the torch-eager restatement of the reference (fixture host) on platform-independent inputs
def _sha(t):
    return hashlib.sha1(np.ascontiguousarray(t).tobytes()).hexdigest()


def _digest_cases():
    p = os.path.join(GOLDEN, "digests_em_dirichlet.json")
    return json.load(open(p))["cases"] if os.path.exists(p) else []


@pytest.mark.parametrize("c", _digest_cases(), ids=lambda c: f"seed{c['seed']}_K{c['K']}")
def test_engine_matches_torch_digests(c):
    """tests/golden/make_digests.py ran oracle/ref_torch.py (torch's own digamma / lgamma / log / sqrt / sums)
    on the fixture host; the engine must produce the same bits."""
    from helpers import intsynth
    from tclip_amd import engine
    t = intsynth.make_tasks(c["seed"], c["N"], c["K"], c["Q"], c["shots"])
    assert _sha(t[0]) + (_sha(t[2]) if c["shots"] else "") == c["inputs"], "input generator is not reproducible on this host"
    r = engine.run_em_dirichlet(torch.from_numpy(t[0]).to(DEV),
                                torch.from_numpy(t[2]).to(DEV) if c["shots"] else None,
                                torch.from_numpy(t[3]).to(DEV) if c["shots"] else None,
                                n_batches=1, iters=c["iters"], iter_mm=c["iter_mm"], lambd=c["lambd"], hard=c["hard"])
    assert r.mm_iters[0].cpu().tolist() == c["mm_iters"]
    assert _sha(r.alpha.cpu().numpy()) == c["alpha"], "alpha differs from torch's"
    assert _sha(r.u.cpu().numpy()) == c["u"] and _sha(r.v.cpu().numpy()) == c["v"]


def test_bench_multi_rank_path_on_one_gpu():
    """bench.py under the driver's launcher with TWO ranks (gloo for the collective, both ranks on cuda:0 - there is
    one GPU here): the N > 1 code path end to end - same index stream on both ranks, round-robin batches, one
    gather, max-over-ranks time, one JSON line from rank 0 with the work of both ranks."""
    import subprocess
    import sys
    from conftest import ROOT
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--workload", "k100", "--backend", "gloo", "--single-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["tasks_total"] == 2000
    assert d["value"] > 0 and abs(d["value"] - 2000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert 0.5 < d["config"]["mean_accuracy"] < 1.0 and "roofline" in d and "cpu_baseline" not in d
    # what the first hardware scaling curve will show: ranks seen, backend, per-rank step times, and what the one gather carried
    assert d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert len(d["rank_step_ms"]["all"]) == 2 and d["rank_step_ms"]["max"] == pytest.approx(d["ms_per_step"])
    assert d["config"]["gathered"] == {"predictions": [20, 100, 75], "criterions": [20, 20], "mm_iters": [20, 20]}


def test_bench_eight_rank_launch_on_one_gpu():
    """The command the driver will use for the N = 8 point of the scaling curve, with the eight ranks sharing the one GPU
    of this box (gloo, --single-device, K = 100 workload, one step): rendezvous, eight processes loading the library,
    round-robin batches (10 per rank), the packed gather of 80 batches, one JSON line."""
    import subprocess
    import sys
    from conftest import ROOT
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0",
           "--workload", "k100", "--backend", "gloo", "--single-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["config"]["tasks_total"] == 8000
    assert d["config"]["gathered"]["predictions"] == [80, 100, 75] and len(d["rank_step_ms"]["all"]) == 8
    assert 0.5 < d["config"]["mean_accuracy"] < 1.0
