"""BD-CSPN: the torch-eager oracle against the golden vectors produced by the
reference (CPU), torch's norm order restated, and the HIP path against the same vectors (GPU).
Everything is bit-exact: rectified prototypes, responsibilities, predictions, accuracies; the
fixtures cover the three normalisations (UN, L2N, CL2N) and K = 5 ... 397."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names
from oracle import ref_torch

NAMES = golden_names("fs_bdcspn_")


def test_fixtures_present():
    assert len(NAMES) >= 8
    kinds = {str(np.load(os.path.join(GOLDEN, n + ".npz"))["norm_type"]) for n in NAMES}
    assert kinds == {"UN", "L2N", "CL2N"}


def _reference_u(g):
    return (float(g["temp"]) * torch.from_numpy(g["logits"])).softmax(-1)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if str(g["torch_version"]) != torch.__version__:
        pytest.skip("fixtures were made with another torch build")
    K = int(g["K"])
    t = ref_torch.run_bdcspn(torch.from_numpy(g["x_q"]), torch.from_numpy(g["x_s"]), torch.from_numpy(g["y_s"]), n_class=K,
                             temp=float(g["temp"]), norm_type=str(g["norm_type"]))
    assert np.array_equal(t["prototypes"].numpy(), g["prototypes"])
    assert torch.equal(t["u"], _reference_u(g))
    acc = (t["preds"] == torch.from_numpy(g["y_q"]).squeeze(2)).float().mean(1, keepdim=True).numpy()
    assert np.array_equal(acc, g["acc"])


def _row_norm_restated(row):
    """The order csrc's row_norm_torch implements (probed on this torch build): 8 fused accumulators,
    added 0..7, then the K mod 8 tail: the first four as product + add, the rest fused; sqrt."""
    f32 = np.float32

    def fma(a, b, c):       # exact: a 24x24-bit product fits a double, one rounding to fp32... through one to fp64
        return f32(np.float64(a) * np.float64(b) + np.float64(c))
    n = len(row)
    nv = n - n % 8
    acc = np.zeros(8, f32)
    for d in range(nv):
        acc[d % 8] = fma(row[d], row[d], acc[d % 8])
    b = acc[0]
    for j in range(1, 8):
        b = f32(b + acc[j])
    d = nv
    if n - d >= 4:
        for _ in range(4):
            b = f32(b + f32(row[d] * row[d]))
            d += 1
    while d < n:
        b = fma(row[d], row[d], b)
        d += 1
    return np.sqrt(b)


@pytest.mark.skipif(torch.backends.cpu.get_cpu_capability() != "AVX512", reason="order pinned for the AVX-512 ATen kernels")
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 15, 16, 23, 37, 100, 101, 397])
def test_norm_order_restated(n):
    g = torch.Generator().manual_seed(n)
    x = torch.rand((40, n), generator=g) - 0.3
    want = x.norm(p=2, dim=-1).numpy()
    got = np.array([_row_norm_restated(r) for r in x.numpy()], np.float32)
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_matches_reference(name):
    from src.methods.few_shot.bdcspn import BDCSPN
    from src.utils import CfgNode
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"])
    a = CfgNode(norm_type=str(g["norm_type"]), temp=float(g["temp"]), n_class=K)
    m = BDCSPN(model=None, device=torch.device("cuda:0"), log_file=None, args=a)
    logs = m.run_task(task_dic={"x_s": torch.from_numpy(g["x_s"]), "y_s": torch.from_numpy(g["y_s"]),
                                "x_q": torch.from_numpy(g["x_q"]), "y_q": torch.from_numpy(g["y_q"])}, shot=int(g["shots"]))
    assert np.array_equal(m.prototypes.cpu().numpy(), g["prototypes"]), "rectified prototypes differ"
    # the responsibilities are compared through the reference's own softmax of its logged logits, computed by
    # this host's torch: softmax has no logarithm, so that holds on any AVX-512 host with <= 8 threads
    assert torch.equal(m.u.cpu(), _reference_u(g)), "responsibilities differ"
    assert np.array_equal(logs["acc"], g["acc"])
    assert logs["criterions"].shape == g["criterions"].shape and (logs["criterions"] == 0).all()


@pytest.mark.gpu
def test_run_method_takes_normalised_features():
    """Reference split (:160-170): run_task normalises, run_method works on the result."""
    from src.methods.few_shot.bdcspn import BDCSPN
    from src.utils import CfgNode
    g = np.load(os.path.join(GOLDEN, "fs_bdcspn_K10_N4_s4.npz"))
    a = CfgNode(norm_type="L2N", temp=float(g["temp"]), n_class=10)
    x_s, x_q = torch.from_numpy(g["x_s"]), torch.from_numpy(g["x_q"])
    x_s = x_s / x_s.norm(p=2, dim=2, keepdim=True)
    x_q = x_q / x_q.norm(p=2, dim=2, keepdim=True)
    m = BDCSPN(model=None, device=torch.device("cuda:0"), log_file=None, args=a)
    m.run_method(support=x_s.cuda(), query=x_q.cuda(), y_s=torch.from_numpy(g["y_s"]).squeeze(2).cuda(),
                 y_q=torch.from_numpy(g["y_q"]).squeeze(2).cuda(), shot=4)
    assert np.array_equal(m.prototypes.cpu().numpy(), g["prototypes"])
