#!/usr/bin/env python3
"""Golden vectors for BD-CSPN from the REFERENCE's own class
(/root/reference/src/methods/few_shot/bdcspn.py), CPU, on seeded synthetic probability features.
Run in the build container only; the .npz files are committed.

    python tests/golden/make_golden_bdcspn.py

Each file: inputs x_s, y_s, x_q, y_q; the rectified prototypes (return value of
proto_rectification), the last get_logits result (-1/2 squared cosine distances, (N,Q,K)), the
logged accuracies, norm_type, temp."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
# the product package mirrors the reference's `src` module names: load its synthetic-data module by file path
import importlib.util  # noqa: E402
import types  # noqa: E402
_spec = importlib.util.spec_from_file_location("tclip_synth", os.path.join(ROOT, "transductive-clip_amd", "tclip_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)
sys.path[:] = [p for p in sys.path if "transductive-clip_amd" not in p]
for _m in ("clip", "torchvision", "torchvision.transforms"):      # absent from this image, unused on this path
    sys.modules.setdefault(_m, types.ModuleType(_m))


class Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


# name: (K, N, shots, seed, norm_type, temp)
CASES = {
    "fs_bdcspn_K5_N3_s2": (5, 3, 2, 2060, "L2N", 30.0),
    "fs_bdcspn_K10_N4_s4": (10, 4, 4, 2020, "L2N", 30.0),
    "fs_bdcspn_K10_N4_s1_cl2n": (10, 4, 1, 2061, "CL2N", 30.0),
    "fs_bdcspn_K37_N3_s2": (37, 3, 2, 2021, "L2N", 30.0),
    "fs_bdcspn_K37_N3_s3_un": (37, 3, 3, 2062, "UN", 10.0),
    "fs_bdcspn_K100_N3_s1": (100, 3, 1, 2022, "L2N", 30.0),
    "fs_bdcspn_K100_N2_s2_cl2n": (100, 2, 2, 2063, "CL2N", 30.0),
    "fs_bdcspn_K397_N1_s1": (397, 1, 1, 2023, "L2N", 30.0),
}


def main():
    sys.path.insert(0, REF)
    from src.methods.few_shot.bdcspn import BDCSPN
    sys.path.pop(0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    for name in (sys.argv[1:] or list(CASES)):
        K, N, shots, seed, norm_type, temp = CASES[name]
        x_q, y_q = synth.make_query_tasks(N, K, seed=seed, k_eff=5)
        x_s, y_s = synth.make_support(N, K, shots, seed=seed)
        args = Args(norm_type=norm_type, temp=temp, n_class=K)
        m = BDCSPN(model=None, device=torch.device("cpu"), log_file="/tmp/golden.log", args=args)
        seen = {}
        real_rect, real_logits = m.proto_rectification, m.get_logits

        def rect(**kw):
            seen["prototypes"] = real_rect(**kw)
            return seen["prototypes"]

        def logits(w, samples):
            seen["logits"] = real_logits(w, samples)
            return seen["logits"]
        m.proto_rectification, m.get_logits = rect, logits
        logs = m.run_task(task_dic={"x_s": x_s.clone(), "y_s": y_s.clone(), "x_q": x_q.clone(), "y_q": y_q.clone()}, shot=shots)
        out = {"K": K, "N": N, "shots": shots, "seed": seed, "norm_type": norm_type, "temp": temp,
               "x_s": x_s.numpy(), "y_s": y_s.numpy(), "x_q": x_q.numpy(), "y_q": y_q.numpy(),
               "prototypes": seen["prototypes"].numpy(), "logits": seen["logits"].numpy(),
               "acc": np.asarray(logs["acc"], np.float32), "criterions": np.asarray(logs["criterions"], np.float32),
               "torch_version": torch.__version__}
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: acc={out['acc'].ravel().round(3).tolist()} -> {os.path.getsize(path) / 1e3:.0f} kB", flush=True)


if __name__ == "__main__":
    main()
