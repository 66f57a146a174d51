"""Result files in the reference's format (src/eval_zero_shot.py:189-232, src/eval_few_shot.py:272-338,
test mode; the validation-sweep branch serves only the tunable baselines and is not needed for
EM-Dirichlet, config/methods_config/em_dirichlet.yaml:9 `tunable: False`)."""
import os


def report_results(args, mean_accuracies, mean_times, logger=None, root="."):
    """Appends one row to results_{zero,few}_shot/<used_test_set>/<dataset>/<METHOD>_<softmax|visual>_...txt
    exactly as the reference's Evaluator_*.report_results does when `save_results` is set; returns
    the file path (or None when nothing is written)."""
    few = int(getattr(args, "shots", 0)) > 0
    word = "_softmax" if args.use_softmax_feature else "_visual"
    info = logger.info if logger is not None else (lambda *_: None)
    info("----- Final results -----")
    info("{}-shot mean test accuracy over {} tasks: {}".format(args.shots, args.number_tasks, mean_accuracies))
    info("{}-shot mean time over {} tasks: {}".format(args.shots, args.number_tasks, mean_times))
    if not getattr(args, "save_results", False):
        return None
    kind = "results_few_shot" if few else "results_zero_shot"
    path = os.path.join(root, "{}/{}/{}".format(kind, args.used_test_set, args.dataset))
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
