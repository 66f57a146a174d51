#!/usr/bin/env python3
"""Runs the reference's evaluation on saved feature pickles with the MI355X engine.

The reference's main.py needs CLIP and the image datasets to extract features; users who already
have the `.plk` files it writes (data/<dataset>/saved_features/<split>_softmax_<backbone>_T<T>.plk,
src/utils.py:266-267) can run the task loop from them:

    python main_features.py --query test_softmax_RN50_T30.plk [--support train_softmax_RN50_T30.plk] \\
        --opts method em_dirichlet dataset caltech101 number_tasks 1000 batch_size 100 shots 0

Without `--query` the files are looked up where the reference keeps them, as `main.py` does once the features exist:
<results-root>/data/<dataset>/saved_features/<used_test_set>_softmax_<backbone>_T<T>.plk (and train_... for the
support set of a few-shot run):

    python main_features.py --opts dataset food101 method paddle shots 4 number_tasks 1000 batch_size 100

Configuration follows main.py:19-35: with `--config-root DIR` (or a `config/` directory in the working
directory) the reference-format YAML files are read - DIR/main_config.yaml, then `--opts`, then
DIR/datasets_config/config_<dataset>.yaml and DIR/methods_config/<method>.yaml, then `--opts` again, so
the command line wins over all three files (values are literal-eval'ed, an override must keep the type of
the value it replaces).  Without a config directory the built-in copies of the reference's defaults below are
used the same way.  Test-split runs of a tunable few-shot method (PADDLE, BDCSPN) need the validation sweep
file under <results-root>/results_few_shot/val/, exactly as the reference does; runs with `used_test_set val` append to it.
Under `python -m torch.distributed.run --nproc-per-node N` batches are sharded over N GPUs.
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # see tclip_amd/__init__.py: the engine's streams need their own hardware queues

HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (HERE, os.path.join(HERE, "drop_in")):          # tclip_amd/ and the reference-named src/ package
    if _p not in sys.path:
        sys.path.insert(0, _p)

from src.utils import CfgNode, Logger, load_merged_config, merge_cfg_from_list  # noqa: E402
from tclip_amd import features, reporting  # noqa: E402

MAIN_DEFAULTS = dict(dataset="synthetic", method="em_dirichlet", number_tasks=5, batch_size=5, k_eff=5, n_query=75,
                     shots=0, log_path=".log/", save_results=True, used_test_set="test", device=0, T=30,
                     backbone="RN50", use_softmax_feature=True, seed=2020, cuda=True)
METHOD_DEFAULTS = {
    "em_dirichlet": dict(name_method="EM_DIRICHLET", iter=20, iter_mm=1000, graph_matching=True, tunable=False),
    "hard_em_dirichlet": dict(name_method="HARD_EM_DIRICHLET", iter=10, iter_mm=1000, graph_matching=True, tunable=False),
    "soft_kmeans": dict(name_method="SOFT_KMEANS", iter=20, graph_matching=True, tunable=False),
    "hard_kmeans": dict(name_method="HARD_KMEANS", iter=10, graph_matching=True, tunable=False),
    "kl_kmeans": dict(name_method="KL_KMEANS", iter=10, graph_matching=True, tunable=False),
    "em_gaussian": dict(name_method="EM_GAUSSIAN", iter=20, graph_matching=True, tunable=False),
    "inductive_clip": dict(name_method="CLIP", iter=1, graph_matching=False, tunable=False),
    "em_gaussian_cov": dict(name_method="EM_GAUSSIAN_COV", iter=20, graph_matching=True, tunable=False),
    "paddle": dict(name_method="PADDLE", iter=20, lambd=0.0, tunable=True),
    "bdcspn": dict(name_method="BDCSPN", num_NN=1, norm_type="L2N", temp=30.0, tunable=True),
    "laplacian_shot": dict(name_method="LAPLACIAN_SHOT", knn=3, lmd=0.7, norm_type="L2N", iter=20, temp=30, tunable=True),
    "alpha_tim": dict(name_method="ALPHA_TIM", temp=15, loss_weights=[1.0, 1.0, 1.0], lr_alpha_tim=1e-4, iter=1000,
                      entropies=["Shannon", "Alpha", "Alpha"], alpha_value=7.0, acc_clustering=False, tunable=True),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--query", default=None, help="feature pickle of the query/test split (default: the reference's "
                                                  "data/<dataset>/saved_features/ layout under --results-root)")
    ap.add_argument("--support", default=None, help="feature pickle of the support/train split (few-shot)")
    ap.add_argument("--results-root", default=".")
    ap.add_argument("--config-root", default=None, help="directory with main_config.yaml, datasets_config/, methods_config/ "
                                                        "(default: ./config if it exists, else the built-in defaults)")
    ap.add_argument("--opts", default=None, nargs=argparse.REMAINDER)
    ns = ap.parse_args(argv)
    opts = ns.opts or []
    if len(opts) % 2:
        ap.error("--opts takes key value pairs")
    root = ns.config_root or ("config" if os.path.isfile(os.path.join("config", "main_config.yaml")) else None)
    if root is not None:
        cfg = load_merged_config(root, opts)
        cfg.setdefault("seed", 2020)
    else:
        cfg = merge_cfg_from_list(CfgNode(MAIN_DEFAULTS), opts)
        if cfg.method not in METHOD_DEFAULTS:
            ap.error(f"method must be one of {sorted(METHOD_DEFAULTS)}")
        cfg.update(METHOD_DEFAULTS[cfg.method])
        cfg = merge_cfg_from_list(cfg, opts)           # command line wins, as in main.py:32-33
    cfg.results_root = ns.results_root
    return ns, cfg


def main(argv=None, keep_process_group=False):
    """keep_process_group: a caller that runs several evaluations in one process destroys the group itself"""
    ns, args = parse_args(argv)
    dist_on = "RANK" in os.environ
    local_rank = int(os.environ.get("LOCAL_RANK", args.device))
    if args.seed is not None:                  # main.py:42-46
        random.seed(args.seed)
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        if not dist.is_initialized():          # a caller may run main() repeatedly in one process: the group is created once
            dist.init_process_group("nccl", device_id=device)
    if ns.support is not None and ns.query is None:
        raise SystemExit("--support needs --query (or give neither and use the data/<dataset>/saved_features/ layout)")
    query_path = ns.query or reporting.saved_feature_path(args, args.used_test_set, ns.results_root)
    if not os.path.exists(query_path):
        raise SystemExit(f"{query_path} not found: extract the features with the reference first (CLIP is out of scope here)")
    feats_q, labels_q = features.load_features(query_path)
    args.num_classes_test = int(feats_q.shape[1]) if args.use_softmax_feature else int(labels_q.max()) + 1
    args.n_class = args.num_classes_test
    logger = Logger(__name__, None)
    if int(args.shots) > 0:
        from src.eval_few_shot import Evaluator_few_shot
        support_path = ns.support or (reporting.saved_feature_path(args, "train", ns.results_root) if ns.query is None else None)
        if support_path is None:
            raise SystemExit("few-shot evaluation needs --support")
        if not os.path.exists(support_path):
            raise SystemExit(f"{support_path} not found: extract the features with the reference first")
        feats_s, labels_s = features.load_features(support_path)
        ev = Evaluator_few_shot(device=device, args=args, log_file=None)
        acc, t = ev.evaluate_tasks(None, feats_s, labels_s, feats_q, labels_q)
    else:
        from src.eval_zero_shot import Evaluator_zero_shot
        ev = Evaluator_zero_shot(device=device, args=args, log_file=None)
        acc, t = ev.evaluate_tasks(None, feats_q, labels_q)
    path = None
    if acc is not None:                        # rank 0
        path = reporting.report_results(args, acc, t, logger, root=ns.results_root)
        print(f"mean accuracy {100 * float(acc):.2f} %  mean time/task statistic {t:.3e} s" + (f"  -> {path}" if path else ""))
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        if keep_process_group is False:
            dist.destroy_process_group()
    return acc, t, path


if __name__ == "__main__":
    main()
