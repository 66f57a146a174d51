#!/usr/bin/env python3
"""Golden vectors for the task-batch loop, produced by RUNNING the
reference's Evaluator_*.evaluate_tasks on a synthetic feature table in the build container
(/root/reference is imported, never copied; clip/torchvision are stubbed as in make_golden.py).

Seeds are set the way main.py:42-46 sets them, the sampler iterators are wrapped to record the
index tensors they yield, and the returned (mean accuracy, mean time) is stored.

    python tests/golden/make_golden_eval.py
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# The product package also has a top-level `src/` (it mirrors the reference's module names), so it
# must NOT be importable while the reference is: load the synthetic-data module by file path.
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location(
    "tclip_synth", os.path.join(ROOT, "transductive-clip_amd", "tclip_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)
sys.path[:] = [p for p in sys.path if "transductive-clip_amd" not in p]

for _m in ("clip", "torchvision", "torchvision.transforms"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
REF = "/root/reference"
try:                                    # laplacian_shot.py imports matplotlib only to select a backend
    import matplotlib  # noqa: F401
except ImportError:
    _mpl = types.ModuleType("matplotlib")
    _mpl.use = lambda *a, **k: None
    sys.modules["matplotlib"] = _mpl
if not hasattr(np, "float"):            # laplacian_shot.py:100 uses the alias numpy 1.24 removed; see make_golden_lshot.py
    np.float = float


class Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def seed_all(seed):
    random.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)


def record_iter(cls, store):
    orig = cls.__iter__

    def wrapped(self):
        for item in orig(self):
            store.append(item.clone())
            yield item
    cls.__iter__ = wrapped
    return orig


def main():
    sys.path.insert(0, REF)
    import src.eval_zero_shot as ez
    import src.eval_few_shot as ef
    sys.path.pop(0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    seed = 2020
    only = set(sys.argv[1:])
    cases = [("zs", False, None), ("zs", True, None), ("fs", False, None), ("fs", True, None),
             ("zs", False, "HARD_KMEANS"), ("zs", False, "EM_GAUSSIAN"), ("zs", False, "EM_GAUSSIAN_COV"), ("zs", False, "SOFT_KMEANS"), ("zs", False, "KL_KMEANS"), ("zs", False, "CLIP"), ("fs", False, "PADDLE"), ("fs", False, "BDCSPN"), ("fs", False, "ALPHA_TIM"), ("fs", False, "LAPLACIAN_SHOT")]
    # (round 5) one evaluator fixture at a BASELINE class count: configs[1]'s shape, K = 100 with batch_size = 100 (two
    # batches; the reference's (N,Q,K,K) temporary is 300 MB); selected by name only: `make_golden_eval.py K100`
    big = {"K": 100, "number_tasks": 200, "batch_size": 100}
    if "K100" in only:
        only.discard("K100")
        cases = [("zs", False, None, big)]
    if "K100more" in only:          # few-shot soft and zero-shot hard at the same class count and batch size
        only.discard("K100more")
        cases = [("fs", False, None, big), ("zs", True, None, big)]
    if "K100r6" in only:            # (round 6) few-shot hard, and SOFT_KMEANS (configs[2]'s second method: the register-tiled statistics
        only.discard("K100r6")      # kernel), at the same class count and batch size
        cases = [("fs", True, None, big), ("zs", False, "SOFT_KMEANS", big)]
    for case in cases:
        kind, hard, other = case[:3]
        shape = case[3] if len(case) > 3 else {"K": 10, "number_tasks": 20, "batch_size": 10}
        K, n_tasks, bs = shape["K"], shape["number_tasks"], shape["batch_size"]
        method = other or ("HARD_EM_DIRICHLET" if hard else "EM_DIRICHLET")
        if only and method not in only and f"{kind}:{method}" not in only:
            continue
        if only and method in only and any(":" in o for o in only) and f"{kind}:{method}" not in only:
            continue
        args = Args(iter=10 if (hard or other in ("HARD_KMEANS", "KL_KMEANS")) else 20, iter_mm=1000, num_classes_test=K, n_class=K,
                    n_query=75, k_eff=5, T=30, use_softmax_feature=True, graph_matching=True, shots=2, number_tasks=n_tasks,
                    batch_size=bs, name_method=method, used_test_set="test", tunable=False, lambd=5.0,
                    method=method.lower(), dataset="synthetic", norm_type="L2N", temp=30.0, num_NN=1)
        model = None
        if other == "ALPHA_TIM":          # alpha_tim.yaml; the class toggles model.eval()/train(), so it needs an object
            args.update(iter=1000, temp=15, loss_weights=[1.0, 1.0, 1.0], lr_alpha_tim=1e-4, entropies=["Shannon", "Alpha", "Alpha"],
                        alpha_value=7.0)
            model = types.SimpleNamespace(eval=lambda: None, train=lambda: None)
        if other == "LAPLACIAN_SHOT":     # laplacian_shot.yaml
            args.update(iter=20, knn=3, lmd=0.7, norm_type="L2N", temp=30)
        feats, labels = synth.make_feature_table(K, 40, seed=seed)
        out = {"kind": kind, "hard": hard, "K": K, "seed": seed, "rows_per_class": 40, "method": method,
               "iters": args.iter, "lambd": args.lambd,
               "number_tasks": n_tasks, "batch_size": bs, "shots": 2}
        seed_all(seed)
        if kind == "zs":
            q = []
            o = record_iter(ez.SamplerQuery_zero_shot, q)
            ev = ez.Evaluator_zero_shot(device=torch.device("cpu"), args=args, log_file="/tmp/golden_eval.log")
            per_task = []                 # every task's accuracy, as handed to compute_confidence_interval (eval_zero_shot.py:176)
            real_ci = ez.compute_confidence_interval

            def recording_ci(data, *a, **k):
                per_task.append(np.asarray(data, np.float32).copy())
                return real_ci(data, *a, **k)
            ez.compute_confidence_interval = recording_ci
            try:
                acc, t = ev.evaluate_tasks(None, feats, labels)
            finally:
                ez.compute_confidence_interval = real_ci
            if K > 10:
                out["task_accuracy"] = np.stack(per_task)        # (batches, batch_size)
            ez.SamplerQuery_zero_shot.__iter__ = o
            out["query_idx"] = torch.stack(q).numpy().reshape(n_tasks // bs, bs, 75)
        else:
            feats_s, labels_s = synth.make_feature_table(K, 16, seed=seed + 1)
            q, s = [], []
            oq = record_iter(ef.SamplerQuery_few_shot, q)
            os_ = record_iter(ef.SamplerSupport_few_shot, s)
            ev = ef.Evaluator_few_shot(device=torch.device("cpu"), args=args, log_file="/tmp/golden_eval.log")
            per_task = []
            real_ci = ef.compute_confidence_interval

            def recording_ci(data, *a, **k):
                per_task.append(np.asarray(data, np.float32).copy())
                return real_ci(data, *a, **k)
            ef.compute_confidence_interval = recording_ci
            try:
                acc, t = ev.evaluate_tasks(model, feats_s, labels_s, feats, labels)
            finally:
                ef.compute_confidence_interval = real_ci
            if K > 10:
                out["task_accuracy"] = np.stack(per_task)        # (batches, batch_size)
            ef.SamplerQuery_few_shot.__iter__ = oq
            ef.SamplerSupport_few_shot.__iter__ = os_
            out["query_idx"] = torch.stack(q).numpy().reshape(n_tasks // bs, bs, 75)
            out["support_idx"] = torch.stack(s).numpy().reshape(n_tasks // bs, bs, -1)
            out["support_rows_per_class"] = 16
        out["mean_accuracy"] = np.float64(acc)
        name = f"eval_{kind}_{'hard' if hard else 'soft'}_K{K}" if other is None else f"eval_{kind}_{other.lower()}_K{K}"
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "acc", acc, "time", t, {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
