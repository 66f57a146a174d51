"""Result files in the reference's format (src/eval_zero_shot.py:189-232, src/eval_few_shot.py:272-338):
the test-mode row per run, and for few-shot runs on the validation split the `val_param<TAB>acc` sweep
file that test runs of a tunable method (PADDLE, BDCSPN) read their parameter from
(eval_few_shot.py:152-187 -> Evaluator_few_shot.set_method_opt_param)."""
import os

# The parameter a validation sweep varies, per method (eval_few_shot.py:130-139).
VAL_PARAM = {"LAPLACIAN_SHOT": "lmd", "ALPHA_TIM": "alpha_value", "PADDLE": "lambd", "BDCSPN": "temp"}


def saved_feature_path(args, split, root="."):
    """data/<dataset>/saved_features/<split>_softmax_<backbone>_T<T>.plk, or <split>_visual_<backbone>.plk
    (src/utils.py:266-267, 324-325)."""
    name = ("{}_softmax_{}_T{}.plk".format(split, args.backbone, args.T) if args.use_softmax_feature
            else "{}_visual_{}.plk".format(split, args.backbone))
    return os.path.join(root, "data", str(args.dataset), "saved_features", name)


def report_results(args, mean_accuracies, mean_times, logger=None, root="."):
    """Appends one row to results_{zero,few}_shot/<used_test_set>/<dataset>/<METHOD>_<softmax|visual>_...txt
    exactly as the reference's Evaluator_*.report_results does when `save_results` is set; returns
    the file path (or None when nothing is written)."""
    few = int(getattr(args, "shots", 0)) > 0
    word = "_softmax" if args.use_softmax_feature else "_visual"
    info = logger.info if logger is not None else (lambda *_: None)
    info("----- Final results -----")
    info("{}-shot mean test accuracy over {} tasks: {}".format(args.shots, args.number_tasks, mean_accuracies))
    kind = "results_few_shot" if few else "results_zero_shot"
    path = os.path.join(root, "{}/{}/{}".format(kind, args.used_test_set, args.dataset))
    if few and args.used_test_set == "val":                 # eval_few_shot.py:282-302: written whatever save_results says
        if args.name_method not in VAL_PARAM:
            raise AttributeError("method {} has no validation parameter".format(args.name_method))  # reference: self.val_param unset
        val_param = args[VAL_PARAM[args.name_method]]
        name_file = os.path.join(path, "{}_s{}.txt".format(args.name_method + word, args.shots))
        os.makedirs(path, exist_ok=True)
        new = not os.path.isfile(name_file)
        with open(name_file, "w" if new else "a") as f:
            if new:
                f.write("val_param\tacc\n")
            f.write(str(val_param) + "\t")
            f.write(str(round(100 * float(mean_accuracies), 2)) + "\t")
            f.write("\n")
        return name_file
    info("{}-shot mean time over {} tasks: {}".format(args.shots, args.number_tasks, mean_times))
    if not getattr(args, "save_results", False) or (few and args.used_test_set != "test"):
        return None
    if few:
        var = "{}\t{}\t{}".format(args.shots, args.n_query, args.k_eff)
        names = "shots\tn_query\tk_eff\tacc\n"
        name_file = os.path.join(path, "{}_s{}.txt".format(args.name_method + word, args.shots))
    else:
        var = "{}\t{}\t{}".format(args.shots, args.n_query, args.number_tasks)
        names = "shots\tn_query\tn_task\tacc\n"
        name_file = os.path.join(path, "{}_{}shot.txt".format(args.name_method + word, args.shots))
    os.makedirs(path, exist_ok=True)
    new = not os.path.isfile(name_file)
    with open(name_file, "w" if new else "a") as f:
        if new:
            f.write(names + "\t" + "\n")
        f.write(var + "\t")
        f.write(str(round(100 * float(mean_accuracies), 1)) + "\t")
        f.write("\n")
    return name_file
