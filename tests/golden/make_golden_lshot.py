#!/usr/bin/env python3
"""Golden vectors for LAPLACIAN_SHOT from the REFERENCE's own class
(/root/reference/src/methods/few_shot/laplacian_shot.py), CPU (numpy / scipy.sparse / sklearn kNN), on seeded synthetic
probability features.  Run in the build container only; the .npz files are committed.

    python tests/golden/make_golden_lshot.py

The reference's create_affinity passes `dtype=np.float` (laplacian_shot.py:100), an alias of the builtin `float` that
numpy removed in 1.24; with numpy 2.2 the class cannot run as it stands.  This script restores the alias
(`np.float = float`) for its own process - a one-line environment fix, not a change of the algorithm - and stubs
matplotlib (imported for `matplotlib.use('Agg')` only) when it is absent.

Each file: inputs x_s, y_s, x_q, y_q; per task the kNN index lists, the unary term, the final assignment, the
per-iteration accuracies (N, iter) and bound energies (N, iter), and the method parameters."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("tclip_synth", os.path.join(ROOT, "transductive-clip_amd", "tclip_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)
sys.path[:] = [p for p in sys.path if "transductive-clip_amd" not in p]
for _m in ("clip", "torchvision", "torchvision.transforms"):      # absent from this image, unused on this path
    sys.modules.setdefault(_m, types.ModuleType(_m))
try:
    import matplotlib  # noqa: F401
except ImportError:
    _mpl = types.ModuleType("matplotlib")
    _mpl.use = lambda *a, **k: None
    sys.modules["matplotlib"] = _mpl
if not hasattr(np, "float"):
    np.float = float


class Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


# name: (K, N, shots, seed, knn, lmd, norm_type, iters)
CASES = {
    "fs_lshot_K5_N3_s2": (5, 3, 2, 4060, 3, 0.7, "L2N", 20),
    "fs_lshot_K10_N4_s4": (10, 4, 4, 4020, 3, 0.7, "L2N", 20),
    "fs_lshot_K10_N3_s1_un": (10, 3, 1, 4061, 5, 1.5, "UN", 20),
    "fs_lshot_K37_N3_s2": (37, 3, 2, 4021, 3, 0.7, "L2N", 20),
    "fs_lshot_K37_N2_s3_k7": (37, 2, 3, 4062, 7, 0.3, "L2N", 12),
    "fs_lshot_K100_N3_s2": (100, 3, 2, 4022, 3, 0.7, "L2N", 20),
    "fs_lshot_K397_N1_s1": (397, 1, 1, 4023, 3, 0.7, "L2N", 20),
}


def main():
    sys.path.insert(0, REF)
    from src.methods.few_shot.laplacian_shot import LAPLACIAN_SHOT
    sys.path.pop(0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    for name in (sys.argv[1:] or list(CASES)):
        K, N, shots, seed, knn, lmd, norm_type, iters = CASES[name]
        x_q, y_q = synth.make_query_tasks(N, K, seed=seed, k_eff=5)
        x_s, y_s = synth.make_support(N, K, shots, seed=seed)
        args = Args(knn=knn, norm_type=norm_type, iter=iters, batch_size=N, shots=shots, lmd=lmd, temp=30, num_classes_test=K,
                    n_class=K)
        m = LAPLACIAN_SHOT(model=None, device=torch.device("cpu"), log_file="/tmp/golden.log", args=args)
        seen = {"knn": [], "unary": [], "preds": []}
        real_aff, real_bound = m.create_affinity, m.bound_update

        def aff(X):
            W = real_aff(X)
            dense = W.toarray()
            seen["knn"].append(np.stack([np.sort(np.nonzero(dense[i])[0]) for i in range(dense.shape[0])]))
            return W

        def bound(**kw):
            seen["unary"].append(np.asarray(kw["unary"]).copy())
            out = real_bound(**kw)
            seen["preds"].append(np.asarray(out[0]).copy())
            return out
        m.create_affinity, m.bound_update = aff, bound
        logs = m.run_task(task_dic={"x_s": x_s.clone(), "y_s": y_s.clone(), "x_q": x_q.clone(), "y_q": y_q.clone()}, shot=shots)
        out = {"K": K, "N": N, "shots": shots, "seed": seed, "knn": knn, "lmd": lmd, "norm_type": norm_type, "iters": iters,
               "x_s": x_s.numpy(), "y_s": y_s.numpy(), "x_q": x_q.numpy(), "y_q": y_q.numpy(),
               "neighbours": np.stack(seen["knn"]).astype(np.int32), "unary": np.stack(seen["unary"]).astype(np.float32),
               "preds": np.stack(seen["preds"]).astype(np.int32), "acc": np.asarray(logs["acc"], np.float32),
               "ent_energy": np.asarray(logs["ent_energy"], np.float64), "torch_version": torch.__version__,
               "numpy_version": np.__version__}
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: acc first/last={out['acc'][:, 0].round(3).tolist()} / {out['acc'][:, -1].round(3).tolist()} "
              f"E[-1]={out['ent_energy'][:, -1].round(4).tolist()} -> {os.path.getsize(path) / 1e3:.0f} kB", flush=True)


if __name__ == "__main__":
    main()
