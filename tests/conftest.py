import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "transductive-clip_amd")        # holds tclip_amd (the engine) - the only entry INTEGRATION.md puts on PYTHONPATH
DROP_IN = os.path.join(PKG, "drop_in")                   # holds src/ (reference-named classes): NOT on the path of a Level-1 user
for p in (ROOT, PKG, DROP_IN):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(prefix=""):
    """method-level fixtures (the eval_* files pin the task-batch loop and have another schema)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and f.startswith(prefix) and not f.startswith("eval_")
                  and (prefix or not f.startswith(("bigbatch_", "lean_")))        # lean schema: digests + samples, inputs regenerated
                  and (prefix or not any(m in f for m in ("_skm_", "_hkm_", "_paddle_", "_emg_", "_klk_", "_emgc_", "_bdcspn_", "_tim_", "_lshot_"))))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# torch's CPU reductions split their work by thread count, and from 16 threads on the reference's
# own results change (measured: identical for 1, 2, 3, 5 and 8 threads, different for 16+).  The
# fixtures were made with 8 threads; wherever a test runs torch on the host as the comparison
# target it must do so under the same condition.
torch.set_num_threads(min(8, torch.get_num_threads()))
