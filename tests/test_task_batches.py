"""CPU: task sampling reproduces the reference's random-number consumption (index tensors
captured from the reference's own samplers, tests/golden/make_golden_eval.py), the few-shot
relabelling matches, and whole-batch sharding works across 2 processes (gloo)."""
import os
import random

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN
from src.utils import CfgNode
from tclip_amd import sharding, synth


def _seed(s):
    random.seed(s)
    torch.manual_seed(s)
    np.random.seed(s)


def _args(g, hard=False):
    return CfgNode(iter=20, iter_mm=1000, num_classes_test=int(g["K"]), n_class=int(g["K"]), n_query=75, k_eff=5,
                   T=30, use_softmax_feature=True, graph_matching=True, shots=int(g["shots"]),
                   number_tasks=int(g["number_tasks"]), batch_size=int(g["batch_size"]),
                   name_method="HARD_EM_DIRICHLET" if hard else "EM_DIRICHLET")


def test_zero_shot_sampler_reproduces_reference_indices():
    from src.eval_zero_shot import Evaluator_zero_shot
    g = np.load(os.path.join(GOLDEN, "eval_zs_soft_K10.npz"))
    _, labels = synth.make_feature_table(int(g["K"]), int(g["rows_per_class"]), seed=int(g["seed"]))
    _seed(int(g["seed"]))
    idx = Evaluator_zero_shot(torch.device("cpu"), _args(g), None).sample_indices(labels)
    assert np.array_equal(idx.numpy(), g["query_idx"])


def test_few_shot_sampler_reproduces_reference_indices():
    from src.eval_few_shot import Evaluator_few_shot
    g = np.load(os.path.join(GOLDEN, "eval_fs_soft_K10.npz"))
    _, labels = synth.make_feature_table(int(g["K"]), int(g["rows_per_class"]), seed=int(g["seed"]))
    _, labels_s = synth.make_feature_table(int(g["K"]), int(g["support_rows_per_class"]), seed=int(g["seed"]) + 1)
    _seed(int(g["seed"]))
    s_idx, q_idx = Evaluator_few_shot(torch.device("cpu"), _args(g), None).sample_indices(labels_s, labels)
    assert np.array_equal(q_idx.numpy(), g["query_idx"])
    assert np.array_equal(s_idx.numpy(), g["support_idx"])


def test_few_shot_relabelling():
    from src.task_generator_few_shot import relabel
    K = 6
    ys = torch.arange(K).repeat_interleave(2)
    xs, xq = torch.rand(12, K), torch.rand(5, K)
    yq = torch.tensor([0, 5, 2, 2, 4])
    a, b, c, d = relabel(xs, xq, ys, yq, True)
    assert torch.equal(c, K - 1 - ys) and torch.equal(d, K - 1 - yq)
    assert torch.equal(a, xs.flip(1)) and torch.equal(b, xq.flip(1))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_batches = 5
    mine = sharding.my_batches(n_batches)
    local = torch.tensor([[10.0 * b + j for j in range(3)] for b in mine]).reshape(len(mine), 3)
    got = sharding.gather_batch_results(local, n_batches)
    if rank == 0:
        torch.save(got, out)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_batch_sharding_two_ranks_gloo(tmp_path):
    assert sharding.my_batches(5, 0, 2) == [0, 2, 4] and sharding.my_batches(5, 1, 2) == [1, 3]
    out = str(tmp_path / "gathered.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    want = torch.tensor([[10.0 * b + j for j in range(3)] for b in range(5)])
    assert torch.equal(got, want)


def _worker_packed(rank, world, port, out):
    """the one collective of a step: predictions (int32), accuracies (f32), criterions (f32), MM counts (int32) of every
    batch of this rank in ONE all_gather block (sharding.gather_packed); 3 ranks, 4 batches: rank 2 holds one batch"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_batches, N, Q, iters = 4, 3, 5, 6
    calls = []
    real = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    mine = sharding.my_batches(n_batches)
    parts = {"preds": torch.stack([_packed_want(n_batches, N, Q, iters)["preds"][b] for b in mine]),
             "acc": torch.stack([_packed_want(n_batches, N, Q, iters)["acc"][b] for b in mine]),
             "criterions": torch.stack([_packed_want(n_batches, N, Q, iters)["criterions"][b] for b in mine]),
             "mm_iters": torch.stack([_packed_want(n_batches, N, Q, iters)["mm_iters"][b] for b in mine])}
    got = sharding.gather_packed(parts, n_batches)
    assert len(calls) == 1, "exactly one collective"
    if rank == 0:
        torch.save(got, out)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def _packed_want(n_batches, N, Q, iters):
    g = torch.Generator().manual_seed(5)
    return {"preds": torch.randint(0, 1000, (n_batches, N * Q), generator=g, dtype=torch.int32),
            "acc": torch.rand(n_batches, N, generator=g),
            "criterions": torch.rand(n_batches, iters, generator=g) * 1e-3,
            "mm_iters": torch.randint(51, 1001, (n_batches, iters), generator=g, dtype=torch.int32)}


def test_packed_gather_three_ranks_gloo(tmp_path):
    out = str(tmp_path / "packed.pt")
    mp.spawn(_worker_packed, args=(3, 29500 + ((os.getpid() + 313) % 2000), out), nprocs=3, join=True)
    got, want = torch.load(out), _packed_want(4, 3, 5, 6)
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k       # bit-cast through int32 and back
    # one rank, no process group: the same dict comes back in batch order
    alone = sharding.gather_packed(want, 4, 0, 1)
    assert all(torch.equal(alone[k], want[k]) for k in want)
    with pytest.raises(TypeError):
        sharding.gather_packed({"x": torch.zeros(1, 2, dtype=torch.float64)}, 1, 0, 1)


def test_method_parts_of_an_idle_rank_have_the_widths_of_a_busy_one():
    from types import SimpleNamespace as NS
    a = NS(name_method="EM_DIRICHLET", iter=20)
    idle = sharding.method_parts(a, None, None, 0, 4, 75, torch.device("cpu"))
    assert {k: tuple(v.shape) for k, v in idle.items()} == {"preds": (0, 300), "acc": (0, 4), "criterions": (0, 20), "mm_iters": (0, 20)}
    m = NS(matched_preds=torch.arange(2 * 4 * 75).view(8, 75), preds=None, criterions_per_batch=np.ones((2, 20), np.float32),
           mm_iters=np.full((2, 20), 51, np.int32))
    busy = sharding.method_parts(a, m, {"acc": np.full((8, 1), 0.5, np.float32)}, 2, 4, 75, torch.device("cpu"))
    assert {k: tuple(v.shape[1:]) for k, v in busy.items()} == {k: tuple(v.shape[1:]) for k, v in idle.items()}
    assert busy["preds"].dtype == torch.int32 and busy["mm_iters"].dtype == torch.int32
    b = NS(name_method="SOFT_KMEANS", iter=20)
    assert sorted(sharding.method_parts(b, None, None, 0, 4, 75, torch.device("cpu"))) == ["acc", "preds"]


def _worker_idle_rank(rank, world, port, out):
    """more ranks than batches: the idle rank hands an empty block to the one gather"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_batches = 1
    mine = sharding.my_batches(n_batches)
    assert mine == ([0] if rank == 0 else [])
    local = torch.full((len(mine), 4), 7.0)
    got = sharding.gather_batch_results(local, n_batches)
    if rank == 0:
        torch.save(got, out)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_more_ranks_than_batches_gloo(tmp_path):
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker_idle_rank, args=(2, 29500 + ((os.getpid() + 991) % 2000), out), nprocs=2, join=True)
    assert torch.equal(torch.load(out), torch.full((1, 4), 7.0))


def test_relabel_indices_equals_get_task_relabelling():
    """Evaluator_few_shot's index-only relabelling (column permutation + re-indexed labels, no feature tensor touched) against
    relabel(), the restatement of Tasks_Generator_few_shot.get_task (task_generator_few_shot.py:41-52), task by task; a support
    set that misses a class is reported as None (the evaluator then materialises the task tensors)."""
    from src.eval_few_shot import relabel_indices
    from src.task_generator_few_shot import relabel
    gen = torch.Generator().manual_seed(5)
    K, T, S, Q = 7, 5, 14, 9
    y_s = torch.stack([torch.arange(K).repeat_interleave(2)[torch.randperm(S, generator=gen)] for _ in range(T)])
    y_q = torch.randint(0, K, (T, Q), generator=gen)
    x_s, x_q = torch.rand(T, S, K, generator=gen), torch.rand(T, Q, K, generator=gen)
    cols, ys2, yq2 = relabel_indices(y_s, y_q, K)
    assert cols.dtype == torch.int32 and tuple(cols.shape) == (T, K)
    for t in range(T):
        xs_ref, xq_ref, ys_ref, yq_ref = relabel(x_s[t], x_q[t], y_s[t], y_q[t], True)
        assert torch.equal(x_s[t][:, cols[t].long()], xs_ref) and torch.equal(x_q[t][:, cols[t].long()], xq_ref)
        assert torch.equal(ys2[t], ys_ref) and torch.equal(yq2[t], yq_ref)
    y_missing = y_s.clone()
    y_missing[2][y_missing[2] == 3] = 4
    assert relabel_indices(y_missing, y_q, K) is None
    # a support set with MORE label values than n_class columns' worth takes the per-task path and is reported too
    assert relabel_indices(y_s, y_q, K + 1) is None


def _worker_eight(rank, world, port, out):
    """configs[3] as the 8-GPU bench runs it, minus the GPUs: 80 batches of 125 tasks dealt to 8 ranks, every rank hands its
    10 batches' records to the ONE packed all_gather; the ranks' step times and device indices travel as the bench moves
    them (sharding.gather_rank_values / check_one_device_per_rank)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_batches, N, Q, iters = 80, 125, 75, 20
    mine = sharding.my_batches(n_batches)
    assert len(mine) == 10 and mine[0] == rank
    calls = []
    real = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    parts = {"preds": torch.stack([torch.full((N * Q,), b, dtype=torch.int32) for b in mine]),
             "acc": torch.stack([torch.full((N,), b / 100.0) for b in mine]),
             "criterions": torch.stack([torch.full((iters,), b * 1e-3) for b in mine]),
             "mm_iters": torch.stack([torch.full((iters,), 51 + b, dtype=torch.int32) for b in mine])}
    got = sharding.gather_packed(parts, n_batches)
    assert len(calls) == 1, "exactly one collective per step"
    dist.all_gather = real
    secs = sharding.gather_rank_values(9.0 + 0.01 * rank)              # every rank sees every rank's time
    assert len(secs) == world and max(secs) == pytest.approx(9.07) and min(secs) == pytest.approx(9.0)
    assert sharding.check_one_device_per_rank(rank) == list(range(world))
    try:
        sharding.check_one_device_per_rank(0)                          # every rank on cuda:0: must be refused
        shared_refused = False
    except RuntimeError:
        shared_refused = True
    assert shared_refused
    # two nodes of four GPUs each: device indices 0..3 repeat across the nodes and must be accepted (until round 5 the check
    # compared bare indices over the whole world and refused every correct multi-node job)
    assert sharding.check_one_device_per_rank(rank % 4, node_key=1000 + rank // 4) == [r % 4 for r in range(world)]
    # a launcher that hands every rank ONE GPU through HIP_VISIBLE_DEVICES: every rank sees index 0, on eight different devices
    # (told apart by uuid / PCI address: `device_key` stands in for it here)
    assert sharding.check_one_device_per_rank(0, device_key=7000 + rank) == [0] * world
    try:
        sharding.check_one_device_per_rank(rank % 2, node_key=1000 + rank // 4)      # ... but not two ranks of ONE node on a device
        shared_refused = False
    except RuntimeError:
        shared_refused = True
    assert shared_refused
    if rank == 0:
        torch.save({"got": got, "secs": secs}, out)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_gather_configs3_records_gloo(tmp_path):
    """world size 8 (one process per GPU of a node), the shapes of the bench line's `config.gathered` and `rank_step_ms`"""
    out = str(tmp_path / "eight.pt")
    mp.spawn(_worker_eight, args=(8, 29500 + ((os.getpid() + 1543) % 2000), out), nprocs=8, join=True)
    r = torch.load(out)
    got = r["got"]
    assert tuple(got["preds"].view(80, 125, 75).shape) == (80, 125, 75) and got["preds"].dtype == torch.int32
    assert tuple(got["criterions"].shape) == (80, 20) and tuple(got["mm_iters"].shape) == (80, 20) and tuple(got["acc"].shape) == (80, 125)
    for b in range(80):                                                # batch order restored from the round-robin deal
        assert int(got["preds"][b, 0]) == b and int(got["mm_iters"][b, 0]) == 51 + b
        assert float(got["acc"][b, 0]) == np.float32(b / 100.0) and float(got["criterions"][b, 0]) == np.float32(b * 1e-3)
    secs = r["secs"]
    skew = (max(secs) - min(secs)) / max(secs)
    assert len(secs) == 8 and 0 < skew < 0.01


def test_node_key_comes_from_the_launcher():
    """the node of a rank: the launcher's GROUP_RANK, else RANK // LOCAL_WORLD_SIZE, else (only then) a host-name hash - nodes or
    containers that share a host name are still told apart under torch.distributed.run"""
    assert sharding._node_key({"GROUP_RANK": "3", "RANK": "25", "LOCAL_WORLD_SIZE": "8"}) == 3.0
    assert sharding._node_key({"RANK": "25", "LOCAL_WORLD_SIZE": "8"}) == 3.0
    assert sharding._node_key({"RANK": "7", "LOCAL_WORLD_SIZE": "8"}) == 0.0
    h = sharding._node_key({})
    assert h == sharding._node_key({"RANK": "5"}) and h > 2 ** 20 and float(int(h)) == h       # host-name hash, exact in float64
    assert sharding._device_key(3) == 3.0                                   # no GPU in this process: the index itself


def test_host_thread_cap_respects_local_world_size(monkeypatch):
    """tclip_host_threads: min(16, cores / LOCAL_WORLD_SIZE), TCLIP_HOST_THREADS wins"""
    import subprocess
    import sys
    from conftest import PKG
    code = "import sys; sys.path.insert(0, %r); from tclip_amd import _capi; print(_capi.lib().tclip_host_threads())" % PKG
    cores = len(os.sched_getaffinity(0))

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "TCLIP_HOST_THREADS")}
        e.update(env)
        return int(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.strip())
    assert run() == min(16, cores)
    assert run(LOCAL_WORLD_SIZE="8") == max(1, min(16, cores // 8))
    assert run(LOCAL_WORLD_SIZE="2") == max(1, min(16, cores // 2))
    assert run(LOCAL_WORLD_SIZE="8", TCLIP_HOST_THREADS="5") == 5
