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
