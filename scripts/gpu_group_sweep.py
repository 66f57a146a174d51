#!/usr/bin/env python3
"""The headline step (bench.HEADLINE through Evaluator_zero_shot.evaluate_tasks) under different stream-group layouts
(TCLIP_GROUP_SIZES, one child process each), with the MM instrumentation on and off:

    python scripts/gpu_group_sweep.py [k1000|k100] "4,3,3" "5,3,2" ...        ("default" = the library's own rule)
"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(which, steps):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
    import random
    import numpy as np, torch
    import bench
    from src.eval_zero_shot import Evaluator_zero_shot
    from src.utils import CfgNode
    from tclip_amd import engine, synth
    w = {"k100": bench.SECONDARY, "k1000": bench.HEADLINE}[which]
    K = w["K"]; dev = torch.device("cuda:0")
    feats, labels = synth.make_feature_table(K, w["rows_per_class"], seed=2020)
    n_tasks = w["tasks_per_batch"] * w["batches_per_gpu"]
    cfg = CfgNode(iter=20, iter_mm=1000, num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30, use_softmax_feature=True,
                  graph_matching=True, shots=0, number_tasks=n_tasks, batch_size=w["tasks_per_batch"], name_method="EM_DIRICHLET")
    ev = Evaluator_zero_shot(device=dev, args=cfg, log_file=None)
    random.seed(2020); np.random.seed(2020); torch.manual_seed(2020)
    idx = ev.sample_indices(labels.numpy())
    table, lab = feats.to(dev), labels.to(dev)
    out = {}
    ev.evaluate_tasks(None, table, lab, indices=idx)              # warm-up
    for prof in (False, True, False, True):
        torch.cuda.synchronize()
        engine.profile_enable(prof)
        t0 = time.perf_counter()
        for _ in range(steps):
            acc, _ = ev.evaluate_tasks(None, table, lab, indices=idx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        busy = engine.profile_collect()[0] / steps if prof else None
        engine.profile_enable(False)
        out.setdefault("profile_on" if prof else "profile_off", []).append(dt)
        if prof:
            out.setdefault("mm_busy_s", []).append(busy / 1e3)
    out["acc"] = float(acc)
    print("RESULT " + json.dumps(out), flush=True)


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "--child":
        return child(argv[1], int(argv[2]))
    which = argv[0] if argv and argv[0] in ("k100", "k1000") else "k1000"
    layouts = [a for a in argv if a not in ("k100", "k1000")] or ["default"]
    steps = int(os.environ.get("SWEEP_STEPS", "2"))
    for lay in layouts:
        env = dict(os.environ)
        env.pop("TCLIP_GROUP_SIZES", None); env.pop("TCLIP_STREAM_GROUPS", None)
        if lay.startswith("groups="):
            env["TCLIP_STREAM_GROUPS"] = lay[7:]
        elif lay != "default":
            env["TCLIP_GROUP_SIZES"] = lay
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which, str(steps)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(f"{lay}: FAILED\n{r.stderr[-2000:]}", flush=True)
            continue
        d = json.loads(line[0][7:])
        f = lambda v: "/".join(f"{x:.3f}" for x in v)
        print(f"{which} {lay:12s} s/step profile off {f(d['profile_off'])}  on {f(d['profile_on'])}  MM busy {f(d['mm_busy_s'])}  acc {d['acc']:.6f}", flush=True)


if __name__ == "__main__":
    main()
