#!/usr/bin/env python3
"""Method / dataset sweeps over saved features, in ONE process (the engine library and the feature tables are loaded
once).  Covers what the reference drives with shell loops around main.py (scripts/test_zero_shot.sh,
scripts/test_few_shot.sh, scripts/opt_parameters.sh): every zero-shot method, every few-shot method, and the
validation sweeps that tune the four tunable few-shot baselines.

    python sweep.py zero_shot --datasets food101 [--opts number_tasks 1000 batch_size 100]
    python sweep.py few_shot  --datasets food101 --shots 4
    python sweep.py tune      --datasets food101 dtd --shots 1 2 4 8 16

Feature files are read from <results-root>/data/<dataset>/saved_features/ (the reference's layout), configuration from
`--config-root` or the built-in defaults, exactly as main_features.py does; `--opts` apply to every run.  A run whose
feature files are missing is reported and skipped."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import main_features  # noqa: E402

ZERO_SHOT = ["hard_em_dirichlet", "em_dirichlet", "soft_kmeans", "em_gaussian_cov", "kl_kmeans", "em_gaussian", "hard_kmeans",
             "inductive_clip"]
FEW_SHOT = ["hard_em_dirichlet", "em_dirichlet", "paddle", "alpha_tim", "laplacian_shot", "bdcspn"]
# the grids of the reference's validation sweeps (scripts/opt_parameters.sh)
GRIDS = {
    "alpha_tim": ("alpha_value", [1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 5.5, 6.0, 6.5, 7.0]),
    "bdcspn": ("temp", [1.0, 3.0, 5.0, 10.0, 20.0, 30.0, 40.0, 50.0, 60.0]),
    "paddle": ("lambd", [0.0, 1.0, 2.0, 5.0, 10.0, 20.0, 35.0, 50.0, 100.0]),
    "laplacian_shot": ("lmd", [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0]),
}


def runs(ns):
    for dataset in ns.datasets:
        if ns.mode == "zero_shot":
            for method in ns.methods or ZERO_SHOT:
                yield ["dataset", dataset, "method", method, "shots", "0", "used_test_set", "test"]
        elif ns.mode == "few_shot":
            for shots in ns.shots:
                for method in ns.methods or FEW_SHOT:
                    yield ["dataset", dataset, "method", method, "shots", str(shots), "used_test_set", "test"]
        else:
            for shots in ns.shots:
                for method in ns.methods or list(GRIDS):
                    name, grid = GRIDS[method]
                    for value in grid:
                        yield ["dataset", dataset, "method", method, "shots", str(shots), "used_test_set", "val", name, str(value)]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("mode", choices=["zero_shot", "few_shot", "tune"])
    ap.add_argument("--datasets", nargs="+", required=True)
    ap.add_argument("--shots", nargs="+", type=int, default=[4])
    ap.add_argument("--methods", nargs="+", default=None, help="subset of the mode's methods")
    ap.add_argument("--results-root", default=".")
    ap.add_argument("--config-root", default=None)
    ap.add_argument("--opts", default=[], nargs=argparse.REMAINDER)
    ns = ap.parse_args(argv)
    if ns.mode == "tune" and ns.methods:
        for m in ns.methods:
            if m not in GRIDS:
                ap.error(f"{m} has no tuned parameter (tunable: {sorted(GRIDS)})")
    table = []
    for opts in runs(ns):
        args = ["--results-root", ns.results_root] + (["--config-root", ns.config_root] if ns.config_root else []) + \
               ["--opts"] + opts + list(ns.opts)
        label = " ".join(opts)
        try:
            acc, t, path = main_features.main(args, keep_process_group=True)
        except SystemExit as e:                      # missing feature files, bad options: report and go on (every rank takes the same exit)
            print(f"[skipped] {label}: {e}", flush=True)
            table.append((label, None))
            continue
        if acc is None:                              # under torch.distributed.run only rank 0 holds the gathered result
            continue
        table.append((label, float(acc)))
        print(f"[done] {label}: {100 * float(acc):.2f} %", flush=True)
    if "RANK" in os.environ:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    return table


if __name__ == "__main__":
    main()
