"""Feature files (reference pickle schema) and the probability-feature front-end (F2)."""
import pickle

import numpy as np
import pytest
import torch


def test_pickle_schema_round_trip(tmp_path):
    from tclip_amd import features
    feats, labels = torch.rand(12, 5), torch.arange(12) % 5
    p = str(tmp_path / "test_softmax_RN50_T30.plk")
    features.save_features(p, feats, labels)
    with open(p, "rb") as f:
        d = pickle.load(f)
    assert set(d) == {"concat_features", "concat_labels"}            # src/utils.py:300-306
    f2, l2 = features.load_features(p)
    assert torch.equal(f2, feats) and torch.equal(l2, labels)


@pytest.mark.gpu
@pytest.mark.parametrize("n,D,K,T", [(7, 512, 10, 30), (65, 1024, 100, 30), (33, 512, 1000, 10), (5, 30, 397, 50)])
def test_probability_features_match_formula(n, D, K, T):
    """softmax(T * normalize(f) @ text.T) (src/utils.py:287-290) against an fp64 evaluation of the
    same formula and torch's fp32 one: the reference runs this on its own GPU through cuBLAS, so
    there are no reference bits to match; 2e-6 absolute on probabilities is fp32 GEMM noise."""
    from tclip_amd import features
    g = torch.Generator().manual_seed(n + K)
    f = torch.randn(n, D, generator=g) * 3
    text = torch.randn(K, D, generator=g)
    text /= text.norm(dim=-1, keepdim=True)
    z = features.probability_features(f.cuda(), text.cuda(), T).cpu()
    ref64 = (T * (f.double() / f.double().norm(dim=-1, keepdim=True)) @ text.double().T).softmax(-1)
    fn = f / f.norm(dim=-1, keepdim=True)
    ref32 = (T * fn @ text.T).softmax(-1)
    assert (z.sum(-1) - 1).abs().max() < 1e-5
    assert (z.double() - ref64).abs().max() < 2e-6
    assert (z - ref32).abs().max() < 2e-6
    assert torch.equal(z.argmax(-1), ref64.argmax(-1))


def test_report_results_file_format(tmp_path):
    """Row format of the reference's result files (eval_zero_shot.py:200-224, eval_few_shot.py:306-329)."""
    from src.utils import CfgNode
    from tclip_amd import reporting
    a = CfgNode(shots=0, n_query=75, number_tasks=1000, k_eff=5, use_softmax_feature=True, save_results=True,
                used_test_set="test", dataset="caltech101", name_method="EM_DIRICHLET")
    p = reporting.report_results(a, 0.88412, 0.01, root=str(tmp_path))
    assert p.endswith("results_zero_shot/test/caltech101/EM_DIRICHLET_softmax_0shot.txt")
    reporting.report_results(a, 0.5, 0.01, root=str(tmp_path))
    assert open(p).read() == "shots\tn_query\tn_task\tacc\n\t\n0\t75\t1000\t88.4\t\n0\t75\t1000\t50.0\t\n"
    a.shots = 4
    p = reporting.report_results(a, 0.7361, 0.01, root=str(tmp_path))
    assert p.endswith("results_few_shot/test/caltech101/EM_DIRICHLET_softmax_s4.txt")
    assert open(p).read() == "shots\tn_query\tk_eff\tacc\n\t\n4\t75\t5\t73.6\t\n"
    a.save_results = False
    assert reporting.report_results(a, 0.5, 0.01, root=str(tmp_path)) is None


@pytest.mark.gpu
def test_main_features_cli_matches_reference_accuracy(tmp_path):
    """The saved-features runner on a pickle of the seeded synthetic table: the reference's mean
    accuracy (fixture eval_zs_hard_K10) and a result file in its format."""
    import os
    import sys
    from conftest import GOLDEN, PKG
    from tclip_amd import features, synth
    sys.path.insert(0, PKG)
    import main_features
    g = np.load(os.path.join(GOLDEN, "eval_zs_hard_K10.npz"))
    feats, labels = synth.make_feature_table(int(g["K"]), int(g["rows_per_class"]), seed=int(g["seed"]))
    plk = str(tmp_path / "test_softmax_RN50_T30.plk")
    features.save_features(plk, feats, labels)
    acc, t, path = main_features.main(["--query", plk, "--results-root", str(tmp_path), "--opts", "method", "hard_em_dirichlet",
                                       "number_tasks", "20", "batch_size", "10", "dataset", "synthetic", "seed", str(int(g["seed"]))])
    assert abs(float(acc) - float(g["mean_accuracy"])) < 1e-7
    assert open(path).read().splitlines()[-1].split("\t")[:4] == ["0", "75", "20", str(round(100 * float(g["mean_accuracy"]), 1))]


@pytest.mark.gpu
def test_run_full_evaluation_from_saved_features(tmp_path):
    """Evaluator_zero_shot.run_full_evaluation on the reference's file layout (data/<dataset>/saved_features/...plk,
    utils.py:266-267): the reference's mean accuracy on the seeded table (fixture eval_zs_hard_K10) and its result row."""
    import os
    import random
    from conftest import GOLDEN
    from src.eval_zero_shot import Evaluator_zero_shot
    from src.utils import CfgNode
    from tclip_amd import features, synth
    g = np.load(os.path.join(GOLDEN, "eval_zs_hard_K10.npz"))
    K, seed = int(g["K"]), int(g["seed"])
    feats, labels = synth.make_feature_table(K, int(g["rows_per_class"]), seed=seed)
    d = tmp_path / "data" / "synthetic" / "saved_features"
    d.mkdir(parents=True)
    features.save_features(str(d / "test_softmax_RN50_T30.plk"), feats, labels)
    a = CfgNode(dataset="synthetic", name_method="HARD_EM_DIRICHLET", iter=10, iter_mm=1000, graph_matching=True,
                number_tasks=20, batch_size=10, k_eff=5, n_query=75, shots=0, used_test_set="test", T=30, backbone="RN50",
                use_softmax_feature=True, save_results=True, num_classes_test=K, n_class=K, results_root=str(tmp_path))
    random.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
    acc, _ = Evaluator_zero_shot(torch.device("cuda", 0), a, None).run_full_evaluation(None, None)
    assert abs(float(acc) - float(g["mean_accuracy"])) < 1e-7
    row = open(tmp_path / "results_zero_shot" / "test" / "synthetic" / "HARD_EM_DIRICHLET_softmax_0shot.txt").read().splitlines()[-1]
    assert row.split("\t")[:4] == ["0", "75", "20", str(round(100 * float(g["mean_accuracy"]), 1))]


@pytest.mark.gpu
def test_few_shot_validation_sweep_then_test_run(tmp_path):
    """The reference's tuning workflow for a tunable method on saved features: validation runs append
    `val_param<TAB>acc` rows (eval_few_shot.py:282-302), the test run reads the best one (:152-187) and
    evaluates with it."""
    import random
    from src.eval_few_shot import Evaluator_few_shot
    from src.utils import CfgNode
    from tclip_amd import features, synth
    K = 10
    d = tmp_path / "data" / "synthetic" / "saved_features"
    d.mkdir(parents=True)
    for split, seed in (("train", 1), ("val", 2), ("test", 3)):
        feats, labels = synth.make_feature_table(K, 40, seed=seed)
        features.save_features(str(d / f"{split}_softmax_RN50_T30.plk"), feats, labels)
    a = CfgNode(dataset="synthetic", name_method="PADDLE", iter=20, lambd=0.0, tunable=True, number_tasks=10, batch_size=5,
                k_eff=5, n_query=75, shots=2, used_test_set="val", T=30, backbone="RN50", use_softmax_feature=True,
                save_results=True, num_classes_test=K, n_class=K, results_root=str(tmp_path))

    def run():
        random.seed(7)
        torch.manual_seed(7)
        np.random.seed(7)
        return Evaluator_few_shot(torch.device("cuda", 0), a, None).run_full_evaluation(None, None)[0]

    accs = {}
    for lambd in (0.0, 1.0, 5.0, 25.0):
        a.lambd = lambd
        accs[lambd] = round(100 * float(run()), 2)
    sweep = open(tmp_path / "results_few_shot" / "val" / "synthetic" / "PADDLE_softmax_s2.txt").read().splitlines()
    assert sweep[0] == "val_param\tacc" and [r.split("\t")[0] for r in sweep[1:]] == ["0.0", "1.0", "5.0", "25.0"]
    assert [float(r.split("\t")[1]) for r in sweep[1:]] == [accs[v] for v in (0.0, 1.0, 5.0, 25.0)]
    competing = [v for v in (1.0, 5.0, 25.0)]                      # the reader skips two lines: 0.0 never competes
    best = [v for v in competing if accs[v] == max(accs[c] for c in competing)][-1]
    a.used_test_set, a.lambd = "test", 0.0
    acc_test = run()
    assert a.lambd == best
    row = open(tmp_path / "results_few_shot" / "test" / "synthetic" / "PADDLE_softmax_s2.txt").read().splitlines()[-1]
    assert row.split("\t")[:4] == ["2", "75", "5", str(round(100 * float(acc_test), 1))]
