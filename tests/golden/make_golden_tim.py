#!/usr/bin/env python3
"""Golden vectors for ALPHA_TIM from the REFERENCE's own class
(/root/reference/src/methods/few_shot/tim.py:192-322), CPU autograd + torch.optim.Adam, on seeded synthetic
probability features.  Run in the build container only; the .npz files are committed.

    python tests/golden/make_golden_tim.py

Each file: inputs x_s, y_s, x_q, y_q; the weights after the last Adam step, the query logits of the last iteration
(what compute_acc sees), the logged criterions (iter,) and accuracies, and the method parameters."""
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


class Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


DEFAULT = ("Shannon", "Alpha", "Alpha")
# name: (K, N, shots, seed, iters, alpha_value, entropies, loss_weights, temp, lr)
CASES = {
    "fs_tim_K5_N3_s2": (5, 3, 2, 3060, 200, 7.0, DEFAULT, (1.0, 1.0, 1.0), 15, 1e-4),
    "fs_tim_K10_N4_s4": (10, 4, 4, 3020, 1000, 7.0, DEFAULT, (1.0, 1.0, 1.0), 15, 1e-4),
    "fs_tim_K10_N3_s1_shannon": (10, 3, 1, 3061, 300, 2.0, ("Shannon", "Shannon", "Shannon"), (1.0, 1.0, 0.5), 15, 1e-3),
    "fs_tim_K37_N3_s2": (37, 3, 2, 3021, 1000, 7.0, DEFAULT, (1.0, 1.0, 1.0), 15, 1e-4),
    "fs_tim_K37_N2_s3_a2": (37, 2, 3, 3062, 400, 2.0, ("Shannon", "Alpha", "Shannon"), (0.5, 1.0, 1.0), 10, 1e-3),
    "fs_tim_K100_N3_s2": (100, 3, 2, 3022, 1000, 7.0, DEFAULT, (1.0, 1.0, 1.0), 15, 1e-4),
    "fs_tim_K397_N1_s1": (397, 1, 1, 3023, 300, 7.0, DEFAULT, (1.0, 1.0, 1.0), 15, 1e-4),
}


def main():
    sys.path.insert(0, REF)
    from src.methods.few_shot.tim import ALPHA_TIM
    sys.path.pop(0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    model = types.SimpleNamespace(eval=lambda: None, train=lambda: None)       # the class only toggles its mode
    for name in (sys.argv[1:] or list(CASES)):
        K, N, shots, seed, iters, alpha, ent, lw, temp, lr = CASES[name]
        x_q, y_q = synth.make_query_tasks(N, K, seed=seed, k_eff=5)
        x_s, y_s = synth.make_support(N, K, shots, seed=seed)
        args = Args(iter=iters, loss_weights=list(lw), temp=temp, lr_alpha_tim=lr, entropies=list(ent), alpha_value=alpha,
                    num_classes_test=K, n_class=K, T=30)
        m = ALPHA_TIM(model=model, device=torch.device("cpu"), log_file="/tmp/golden.log", args=args)
        seen = {}
        real_acc = m.compute_acc

        def acc(y_q, logits_q):
            seen["logits_q"] = logits_q.detach().clone()
            return real_acc(y_q=y_q, logits_q=logits_q)
        m.compute_acc = acc
        logs = m.run_task(task_dic={"x_s": x_s.clone(), "y_s": y_s.clone(), "x_q": x_q.clone(), "y_q": y_q.clone()}, shot=shots)
        out = {"K": K, "N": N, "shots": shots, "seed": seed, "iters": iters, "alpha_value": alpha, "entropies": np.array(ent),
               "loss_weights": np.asarray(lw, np.float64), "temp": temp, "lr": lr,
               "x_s": x_s.numpy(), "y_s": y_s.numpy(), "x_q": x_q.numpy(), "y_q": y_q.numpy(),
               "weights": m.weights.detach().numpy(), "logits_q": seen["logits_q"].numpy(),
               "acc": np.asarray(logs["acc"], np.float32), "criterions": np.asarray(logs["criterions"], np.float32),
               "torch_version": torch.__version__}
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: acc={out['acc'].ravel().round(3).tolist()} crit[0,-1]={out['criterions'][[0, -1]].tolist()} "
              f"-> {os.path.getsize(path) / 1e3:.0f} kB", flush=True)


if __name__ == "__main__":
    main()
