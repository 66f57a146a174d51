"""CPU: the reference's configuration handling (main.py:19-35, src/utils.py:90-168) on own fixture YAMLs:
three files merged in order, --opts applied before (to pick the files) and after (to win), type rule."""
import os

import numpy as np
import pytest

from src.utils import CfgNode, load_cfg_from_cfg_file, load_merged_config, merge_cfg_from_list

CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "config")


def test_yaml_sections_are_flattened():
    cfg = load_cfg_from_cfg_file(os.path.join(CONFIG, "datasets_config", "config_toyset.yaml"))
    assert isinstance(cfg, CfgNode) and cfg.dataset == "toyset" and cfg.seed == 2020 and cfg.cuda is True
    with pytest.raises(AssertionError):
        load_cfg_from_cfg_file(os.path.join(CONFIG, "missing.yaml"))


def test_three_file_merge_order():
    cfg = load_merged_config(CONFIG)
    assert cfg.method == "em_dirichlet" and cfg.name_method == "EM_DIRICHLET"
    assert cfg.iter == 20                       # method file over main file
    assert cfg.k_eff == 4                       # dataset file over main file
    assert cfg.n_class == cfg.num_classes_test == 10
    assert cfg.number_tasks == 8 and cfg.save_results is False


def test_opts_pick_the_files_and_win_over_them():
    cfg = load_merged_config(CONFIG, ["method", "hard_em_dirichlet", "dataset", "otherset", "k_eff", "9", "iter", "2",
                                      "SOME.nested.new_key", "[1, 2]", "note", "free text"])
    assert cfg.name_method == "HARD_EM_DIRICHLET"        # the method file named on the command line was read
    assert cfg.num_classes_test == 37 and cfg.seed == 11  # and the dataset file
    assert cfg.k_eff == 9                                # set before the merge (5 -> 9), overwritten by the dataset/method files (6), set again
    assert cfg.iter == 2
    assert cfg.new_key == [1, 2] and cfg.note == "free text"      # unknown keys are added, only the last component counts
    # without the override the method file wins over dataset and main
    assert load_merged_config(CONFIG, ["method", "hard_em_dirichlet"]).k_eff == 6


def test_override_must_keep_the_type():
    cfg = load_cfg_from_cfg_file(os.path.join(CONFIG, "main_config.yaml"))
    with pytest.raises(ValueError):
        merge_cfg_from_list(cfg, ["number_tasks", "many"])          # str for int
    with pytest.raises(ValueError):
        merge_cfg_from_list(cfg, ["T", "30.5"])                     # float for int
    out = merge_cfg_from_list(cfg, ["probe_list", "(3, 4)", "use_softmax_feature", "False"])
    assert out.probe_list == [3, 4] and out.use_softmax_feature is False and cfg.use_softmax_feature is True
    with pytest.raises(AssertionError):
        merge_cfg_from_list(cfg, ["number_tasks"])


def test_main_features_reads_the_yaml_tree(tmp_path):
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    import main_features
    ns, cfg = main_features.parse_args(["--query", "x.plk", "--config-root", CONFIG, "--results-root", str(tmp_path),
                                        "--opts", "method", "paddle", "shots", "2", "lambd", "3.5"])
    assert cfg.name_method == "PADDLE" and cfg.tunable is True and cfg.lambd == 3.5 and cfg.shots == 2
    assert cfg.results_root == str(tmp_path)
    # built-in defaults when no config directory is given: the same override rules
    ns, cfg = main_features.parse_args(["--query", "x.plk", "--opts", "method", "hard_em_dirichlet", "number_tasks", "20"])
    assert cfg.name_method == "HARD_EM_DIRICHLET" and cfg.iter == 10 and cfg.number_tasks == 20


def test_tuned_parameter_comes_from_the_validation_sweep(tmp_path):
    """eval_few_shot.py:152-187: the last best row of results_few_shot/val/<dataset>/<METHOD>_softmax_s<shots>.txt;
    imagenet reads caltech101's file; no file is an error."""
    import torch
    from src.eval_few_shot import Evaluator_few_shot
    a = CfgNode(dataset="imagenet", name_method="PADDLE", use_softmax_feature=True, shots=4, lambd=0.0,
                results_root=str(tmp_path), tunable=True, used_test_set="test")
    ev = Evaluator_few_shot(torch.device("cpu"), a, None)
    with pytest.raises(ValueError):
        ev.set_method_opt_param()
    d = tmp_path / "results_few_shot" / "val" / "caltech101"
    d.mkdir(parents=True)
    (d / "PADDLE_softmax_s4.txt").write_text("val_param\tacc\n\t\n0.0\t71.2\t\n5.0\t80.1\t\n10.0\t80.1\t\n20.0\t79.0\t\n")
    assert ev.set_method_opt_param() == 10.0 and a.lambd == 10.0
    a.name_method, a.temp = "BDCSPN", 30.0
    (d / "BDCSPN_softmax_s4.txt").write_text("val_param\tacc\n\t\n15.0\t60.0\t\n")
    assert ev.set_method_opt_param() == 15.0 and a.temp == 15.0


def test_validation_sweep_file_round_trip(tmp_path):
    """eval_few_shot.py:282-302 writes one header line and a `param<TAB>acc<TAB>` row per validation run (whatever
    save_results says); the reader (:168-176) skips TWO lines, so the first row of a sweep never competes - kept."""
    import torch
    from src.eval_few_shot import Evaluator_few_shot
    a = CfgNode(dataset="dtd", name_method="PADDLE", use_softmax_feature=True, shots=2, lambd=0.0, n_query=75, k_eff=5,
                number_tasks=100, results_root=str(tmp_path), tunable=True, used_test_set="val", save_results=False)
    ev = Evaluator_few_shot(torch.device("cpu"), a, None)
    for lambd, acc in ((0.0, 0.9), (5.0, 0.71234), (10.0, 0.8), (20.0, 0.8), (50.0, 0.3)):
        a.lambd = lambd
        p = ev.report_results(acc, 0.01)
    assert p.endswith("results_few_shot/val/dtd/PADDLE_softmax_s2.txt")
    assert open(p).read() == "val_param\tacc\n0.0\t90.0\t\n5.0\t71.23\t\n10.0\t80.0\t\n20.0\t80.0\t\n50.0\t30.0\t\n"
    a.used_test_set = "test"
    assert ev.set_method_opt_param() == 20.0 and a.lambd == 20.0      # 0.0 scored best but sits on a skipped line
    assert ev.report_results(0.5, 0.01) is None                      # test mode, save_results off: log only
    a.save_results = True
    assert ev.report_results(0.5, 0.01).endswith("results_few_shot/test/dtd/PADDLE_softmax_s2.txt")
    a.used_test_set, a.name_method = "val", "EM_DIRICHLET"           # not tunable: the reference fails on self.val_param
    with pytest.raises(AttributeError):
        ev.report_results(0.5, 0.01)


def test_full_evaluation_needs_the_saved_feature_files(tmp_path):
    import torch
    from src.eval_few_shot import Evaluator_few_shot
    from src.eval_zero_shot import Evaluator_zero_shot
    a = CfgNode(dataset="dtd", name_method="EM_DIRICHLET", use_softmax_feature=True, shots=0, backbone="RN50", T=30,
                used_test_set="test", results_root=str(tmp_path))
    with pytest.raises(FileNotFoundError, match="data/dtd/saved_features/test_softmax_RN50_T30.plk"):
        Evaluator_zero_shot(torch.device("cpu"), a, None).run_full_evaluation(None, None)
    a.shots, a.use_softmax_feature = 4, False
    with pytest.raises(FileNotFoundError, match="data/dtd/saved_features/train_visual_RN50.plk"):
        Evaluator_few_shot(torch.device("cpu"), a, None).run_full_evaluation(None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("method,param,values", [("alpha_tim", "alpha_value", ("2.0", "5.0", "7.0")),
                                                 ("laplacian_shot", "lmd", ("0.1", "0.7", "1.5"))])
def test_cli_tuning_workflow_from_yaml(tmp_path, method, param, values):
    """main_features.py with the reference-format config tree for the two tunable baselines: validation runs sweep the
    parameter into results_few_shot/val/, the test run reads the best competing value back and writes its row."""
    import sys
    import torch
    from conftest import PKG
    from tclip_amd import features, synth
    sys.path.insert(0, PKG)
    import main_features
    K = 10
    plk = {}
    for split, seed in (("train", 11), ("val", 12), ("test", 13)):
        feats, labels = synth.make_feature_table(K, 40, seed=seed)
        plk[split] = str(tmp_path / f"{split}.plk")
        features.save_features(plk[split], feats, labels)
    common = ["--support", plk["train"], "--config-root", CONFIG, "--results-root", str(tmp_path), "--opts", "method", method,
              "shots", "2", "number_tasks", "8", "batch_size", "4", "save_results", "True"]
    accs = {}
    for v in values:
        acc, _, path = main_features.main(["--query", plk["val"]] + common + ["used_test_set", "val", param, v])
        accs[float(v)] = round(100 * float(acc), 2)
    rows = open(path).read().splitlines()
    assert rows[0] == "val_param\tacc" and [float(r.split("\t")[0]) for r in rows[1:]] == [float(v) for v in values]
    competing = [float(v) for v in values[1:]]
    best = [v for v in competing if accs[v] == max(accs[c] for c in competing)][-1]
    acc, _, path = main_features.main(["--query", plk["test"]] + common)
    assert path.endswith(f"results_few_shot/test/toyset/{method.upper()}_softmax_s2.txt")
    assert open(path).read().splitlines()[-1].split("\t")[:4] == ["2", "75", "4", str(round(100 * float(acc), 1))]
    # The reference builds batch 0's method BEFORE it reads the sweep file (eval_few_shot.py:250-254) and the methods copy
    # their parameter in __init__: batch 0 runs with the YAML default, batch 1 with the tuned value.  Reproduced by default;
    # per-batch accuracies of hand-fixed runs tell the two halves apart.
    from src import eval_few_shot
    per_batch = {}
    real = eval_few_shot.Evaluator_few_shot.evaluate_tasks

    def spy(self, *a, **k):
        r = real(self, *a, **k)
        per_batch["last"] = self.last_task_accuracies.copy()
        return r
    eval_few_shot.Evaluator_few_shot.evaluate_tasks = spy
    try:
        main_features.main(["--query", plk["test"]] + common)
        quirk = per_batch["last"]
        main_features.main(["--query", plk["test"]] + common + ["tunable", "False"])                       # YAML default everywhere
        default = per_batch["last"]
        acc2, _, _ = main_features.main(["--query", plk["test"]] + common + [param, str(best), "tunable", "False"])    # best everywhere
        fixed = per_batch["last"]
        acc3, _, _ = main_features.main(["--query", plk["test"]] + common + ["tuned_param_for_every_batch", "True"])
        every = per_batch["last"]
    finally:
        eval_few_shot.Evaluator_few_shot.evaluate_tasks = real
    assert quirk.shape == (2, 4)
    assert np.array_equal(quirk[0], default[0]) and np.array_equal(quirk[1], fixed[1])
    # tuned_param_for_every_batch: the tuned value from batch 0 on = the value fixed by hand with tuning off
    assert np.array_equal(every, fixed) and float(acc3) == float(acc2)
    torch.cuda.synchronize()
