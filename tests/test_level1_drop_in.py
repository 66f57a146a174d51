"""INTEGRATION.md's Level 1 ("copy five files into the reference checkout, put tclip_amd on PYTHONPATH"),
executed AS PRINTED.

The reference's `src/` is a namespace package (no __init__.py); a regular package named `src` anywhere on the path
takes every `src.*` import of the reference's main.py away from it (round 3's procedure did exactly that and died at
main.py:10).  Two tests keep the documented procedure alive:

* build container (needs /root/reference, nothing of it travels): a scratch copy of the reference, the document's
  bash block run in it, then - in a fresh interpreter started the way `python main.py` starts one - main.py:10-12's
  import lines, and the four classes built by the REFERENCE's Evaluator_{zero,few}_shot.get_method_builder
  (eval_zero_shot.py:113-138, eval_few_shot.py:189-198);
* GPU box (no reference there): the same block run into a scratch tree that holds a stand-in `src/utils.py`
  (a Logger written here), the copied classes imported under the name `src.methods...` with this repo's own
  `drop_in/` NOT on the path, and run against reference fixtures."""
import os
import shutil
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import GOLDEN, PKG, ROOT

REF = "/root/reference"
OVERLAID = ["src/methods/_em_dirichlet_base.py", "src/methods/zero_shot/em_dirichlet.py",
            "src/methods/zero_shot/hard_em_dirichlet.py", "src/methods/few_shot/em_dirichlet.py",
            "src/methods/few_shot/hard_em_dirichlet.py"]


def level1_block():
    """the first ```bash block of INTEGRATION.md's Level-1 section, line by line"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## Level 1"):doc.index("## Level 2")]
    return sec.split("```bash\n", 1)[1].split("```", 1)[0].splitlines()


def run_block(cwd, probe, run_build):
    """Runs the document's lines in ONE bash, with the placeholder path filled in, `python main.py ...` replaced by
    `python <probe>` (main.py itself needs CLIP weights and image datasets) and, on request, without the build line
    (the GPU box runs the library the snapshot carries)."""
    lines = []
    seen = set()
    for ln in level1_block():
        if ln.startswith("TCLIP="):
            ln = f"TCLIP={PKG}"
            seen.add("TCLIP")
        elif ln.startswith("python $TCLIP/build.py"):
            seen.add("build")
            if not run_build:
                continue
        elif ln.startswith("python main.py"):
            ln = f"{sys.executable} {probe}"
            seen.add("main")
        elif ln.startswith("export PYTHONPATH="):
            seen.add("path")
        lines.append(ln)
    assert seen == {"TCLIP", "build", "main", "path"}, f"INTEGRATION.md's Level-1 block changed shape: {seen}"
    assert sum(ln.startswith("cp ") for ln in lines) == len(OVERLAID)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}     # the document's export is the only entry
    return subprocess.run(["bash", "-ec", "\n".join(lines)], cwd=cwd, env=env, capture_output=True, text=True, timeout=900)


_STUBS = r"""
import os, sys, types
for _m in ("clip", "torchvision", "torchvision.transforms"):       # absent from the image, unused on this path
    sys.modules.setdefault(_m, types.ModuleType(_m))
try:
    import matplotlib  # noqa: F401
except ImportError:
    _mpl = types.ModuleType("matplotlib"); _mpl.use = lambda *a, **k: None; sys.modules["matplotlib"] = _mpl
"""

_PROBE_REFERENCE = _STUBS + r"""
import filecmp, torch
HERE = os.path.dirname(os.path.abspath(__file__))
assert sys.path[0] == HERE                              # started like `python main.py`
TCLIP = {pkg!r}
assert not any(os.path.isdir(os.path.join(p, "src")) for p in sys.path if os.path.abspath(p or ".") != HERE), \
    "a second `src` is importable: the reference's namespace package would be shadowed"
exec("\n".join(open(os.path.join(HERE, "main.py")).read().splitlines()[9:12]))      # main.py:10-12, as written there
import src, src.utils, src.eval_zero_shot, src.eval_few_shot, tclip_amd
assert getattr(src, "__file__", None) is None, "src must stay the reference's namespace package"
assert src.utils.__file__ == os.path.join(HERE, "src", "utils.py") and hasattr(src.utils, "get_log_file")
assert src.eval_zero_shot.__file__ == os.path.join(HERE, "src", "eval_zero_shot.py")
assert os.path.dirname(tclip_amd.__file__) == os.path.join(TCLIP, "tclip_amd")
for rel in {overlaid!r}:
    assert filecmp.cmp(os.path.join(HERE, rel), os.path.join(TCLIP, "drop_in", rel), shallow=False), rel


class Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


built = []
for Ev, pkg in ((Evaluator_zero_shot, "zero_shot"), (Evaluator_few_shot, "few_shot")):
    for name in ("EM_DIRICHLET", "HARD_EM_DIRICHLET"):
        args = Args(iter=3, iter_mm=50, num_classes_test=10, n_class=10, n_query=75, k_eff=5, T=30, shots=2,
                    use_softmax_feature=True, graph_matching=True, name_method=name, used_test_set="test",
                    tunable=False, lambd=0.0, number_tasks=2, batch_size=2, dataset="synthetic")
        log = os.path.join(HERE, "probe.log")
        ev = Ev(device=torch.device("cpu"), args=args, log_file=log)
        m = ev.get_method_builder(model=None, device=torch.device("cpu"), args=args, log_file=log)
        mod = sys.modules[type(m).__module__]
        assert type(m).__name__ == name and type(m).__module__ == "src.methods.%s.%s" % (pkg, name.lower())
        assert mod.__file__ == os.path.join(HERE, "src", "methods", pkg, name.lower() + ".py")
        assert any(c.__name__ == "EMDirichletBase" and c.__module__ == "src.methods._em_dirichlet_base" for c in type(m).__mro__)
        assert type(m.logger).__module__ == "src.utils"                   # the reference's own Logger
        # the engine call is reached: this container has no GPU, and the product has no CPU path
        x_q = torch.softmax(torch.randn(2, 75, 10), -1); y_q = torch.zeros(2, 75, 1, dtype=torch.long)
        task = dict(x_q=x_q, y_q=y_q, x_s=x_q[:, :20], y_s=(torch.arange(20) % 10).view(1, 20, 1).repeat(2, 1, 1))
        try:
            m.run_task(task_dic=task) if pkg == "zero_shot" else m.run_task(task_dic=task, shot=2)
        except RuntimeError as e:
            assert "no CPU fallback" in str(e), e
        else:
            assert torch.cuda.is_available(), "a CPU run must not succeed"
        built.append("%s.%s" % (pkg, name))
print("LEVEL1-OK", ",".join(built))
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists in the build container only")
def test_level1_as_documented_on_a_copy_of_the_reference(tmp_path):
    ref = tmp_path / "reference"
    shutil.copytree(REF, ref, ignore=shutil.ignore_patterns("__pycache__", "figures", "results_few_shot", ".git"))
    assert not (ref / "src" / "__init__.py").exists(), "the reference's src/ is a namespace package"
    probe = ref / "probe_main.py"
    probe.write_text(_PROBE_REFERENCE.format(pkg=PKG, overlaid=OVERLAID))
    r = run_block(str(ref), "probe_main.py", run_build=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "LEVEL1-OK zero_shot.EM_DIRICHLET,zero_shot.HARD_EM_DIRICHLET,few_shot.EM_DIRICHLET,few_shot.HARD_EM_DIRICHLET" in r.stdout


def test_drop_in_directory_is_not_importable_from_the_documented_path():
    """what the document puts on PYTHONPATH must not offer a package named `src`, with or without __init__.py"""
    assert not os.path.exists(os.path.join(PKG, "src"))
    assert os.path.isdir(os.path.join(PKG, "tclip_amd")) and os.path.isdir(os.path.join(PKG, "drop_in", "src"))
    code = "import importlib.util as u, sys; sys.exit(0 if u.find_spec('src') is None and u.find_spec('tclip_amd') else 1)"
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = PKG
    assert subprocess.run([sys.executable, "-c", code], env=env, cwd="/").returncode == 0


# A Logger of this test's own, standing in for the reference's src/utils.py:171-215 on the GPU box (same constructor and
# the three methods the classes use).
_STAND_IN_UTILS = '''
import logging


class Logger:
    def __init__(self, module_name, filename):
        self.logger = logging.getLogger(module_name)
        self.filename = filename
        self.lines = []

    def info(self, msg):
        self.lines.append(msg)

    def del_logger(self):
        self.lines = None
'''

_PROBE_GPU = r"""
import json, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
TCLIP, GOLDEN = {pkg!r}, {golden!r}
assert not any("drop_in" in p for p in sys.path), sys.path
import src.utils
assert src.utils.__file__ == os.path.join(HERE, "src", "utils.py")
from src.methods.zero_shot.em_dirichlet import EM_DIRICHLET as ZS
from src.methods.zero_shot.hard_em_dirichlet import HARD_EM_DIRICHLET as ZSH
from src.methods.few_shot.em_dirichlet import EM_DIRICHLET as FS
from src.methods.few_shot.hard_em_dirichlet import HARD_EM_DIRICHLET as FSH
import src.methods._em_dirichlet_base as base
assert base.__file__ == os.path.join(HERE, "src", "methods", "_em_dirichlet_base.py")
assert base.Logger is src.utils.Logger


class Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


done = []
for cls, name in ((ZS, "zs_soft_K10_N4"), (ZSH, "zs_hard_K37_N6"), (FS, "fs_soft_K37_N3_s2"), (FSH, "fs_hard_K10_N4_s4")):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"]); few = name.startswith("fs")
    args = Args(iter=int(g["iters"]), iter_mm=int(g["iter_mm"]), num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30,
                use_softmax_feature=True, graph_matching=True, shots=int(g["shots"]) if few else 0)
    m = cls(model=None, device=torch.device("cuda:0"), log_file=os.path.join(HERE, "probe.log"), args=args)
    task = dict(x_q=torch.from_numpy(g["x_q"]), y_q=torch.from_numpy(g["y_q"]))
    if few:
        task.update(x_s=torch.from_numpy(g["x_s"]), y_s=torch.from_numpy(g["y_s"]))
        logs = m.run_task(task_dic=task, shot=args.shots)
    else:
        logs = m.run_task(task_dic=task)
    assert np.array_equal(m.alpha.cpu().numpy(), g["alpha"]), name
    assert np.array_equal(m.u.cpu().numpy(), g["u"]) and np.array_equal(m.v.cpu().numpy(), g["v"]), name
    assert np.array_equal(logs["acc"], g["acc"]), name
    assert np.array_equal(logs["criterions"], g["criterions"]), name
    assert m.logger.lines and "Executing" in m.logger.lines[0]
    done.append(name)
print("LEVEL1-GPU-OK", ",".join(done))
"""


@pytest.mark.gpu
def test_level1_copied_classes_run_without_this_repos_src(tmp_path):
    tree = tmp_path / "checkout"
    for d in ("src/methods/zero_shot", "src/methods/few_shot"):
        (tree / d).mkdir(parents=True)
    (tree / "src" / "utils.py").write_text(textwrap.dedent(_STAND_IN_UTILS))
    (tree / "probe_main.py").write_text(_PROBE_GPU.format(pkg=PKG, golden=GOLDEN))
    r = run_block(str(tree), "probe_main.py", run_build=False)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "LEVEL1-GPU-OK zs_soft_K10_N4,zs_hard_K37_N6,fs_soft_K37_N3_s2,fs_hard_K10_N4_s4" in r.stdout


_PROBE_ALL_METHODS = _STUBS + r"""
import glob, shutil, torch
HERE = os.path.dirname(os.path.abspath(__file__))
TCLIP = {pkg!r}
# every method module this repo provides, copied over the reference's (the document's `cp` pattern, all of them)
copied = []
for src_file in sorted(glob.glob(os.path.join(TCLIP, "drop_in", "src", "methods", "*", "*.py")) +
                       [os.path.join(TCLIP, "drop_in", "src", "methods", "_em_dirichlet_base.py")]):
    rel = os.path.relpath(src_file, os.path.join(TCLIP, "drop_in"))
    if os.path.basename(rel) == "__init__.py":
        continue
    shutil.copyfile(src_file, os.path.join(HERE, rel))
    copied.append(rel)
exec("\n".join(open(os.path.join(HERE, "main.py")).read().splitlines()[9:12]))
import src.utils


class Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


built = []
zs = ["KL_KMEANS", "EM_DIRICHLET", "HARD_EM_DIRICHLET", "EM_GAUSSIAN", "EM_GAUSSIAN_COV", "SOFT_KMEANS", "HARD_KMEANS", "CLIP"]
fs = ["EM_DIRICHLET", "HARD_EM_DIRICHLET", "PADDLE", "BDCSPN", "ALPHA_TIM", "LAPLACIAN_SHOT"]
for Ev, names in ((Evaluator_zero_shot, zs), (Evaluator_few_shot, fs)):
    for name in names:
        args = Args(iter=3, iter_mm=50, num_classes_test=10, n_class=10, n_query=75, k_eff=5, T=30, shots=2, use_softmax_feature=True,
                    graph_matching=True, name_method=name, used_test_set="test", tunable=False, lambd=0.0, number_tasks=2, batch_size=2,
                    dataset="synthetic", norm_type="L2N", temp=15.0, num_NN=1, knn=3, lmd=0.7, loss_weights=[1.0, 1.0, 1.0],
                    lr_alpha_tim=1e-4, entropies=["Shannon", "Alpha", "Alpha"], alpha_value=7.0)
        log = os.path.join(HERE, "probe.log")
        ev = Ev(device=torch.device("cpu"), args=args, log_file=log)
        m = ev.get_method_builder(model=None, device=torch.device("cpu"), args=args, log_file=log)
        mod = sys.modules[type(m).__module__]
        assert type(m).__name__ == name, (name, type(m))
        assert mod.__file__.startswith(os.path.join(HERE, "src", "methods")), mod.__file__
        assert "tclip_amd" in open(mod.__file__).read() or "_em_dirichlet_base" in open(mod.__file__).read(), mod.__file__
        built.append(name)
assert src.utils.__file__ == os.path.join(HERE, "src", "utils.py")
print("LEVEL1-ALL-OK", len(copied), ",".join(built))
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists in the build container only")
def test_every_method_module_overlays_the_reference(tmp_path):
    """the widened methods (k-means family, EM_GAUSSIAN(_COV), CLIP, PADDLE, BDCSPN, ALPHA_TIM, LAPLACIAN_SHOT) drop in the same
    way: all of drop_in/src/methods/ copied over a copy of the reference, main.py:10-12's imports, and every method both of the
    reference's evaluators can build comes from the copied modules."""
    ref = tmp_path / "reference"
    shutil.copytree(REF, ref, ignore=shutil.ignore_patterns("__pycache__", "figures", "results_few_shot", ".git"))
    (ref / "probe_all.py").write_text(_PROBE_ALL_METHODS.format(pkg=PKG))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = PKG
    r = subprocess.run([sys.executable, "probe_all.py"], cwd=str(ref), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "LEVEL1-ALL-OK 15 " in r.stdout and r.stdout.strip().endswith("LAPLACIAN_SHOT"), r.stdout[-500:]
