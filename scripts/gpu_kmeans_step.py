#!/usr/bin/env python3
"""SOFT_KMEANS evaluator step at K = 397, 1000 tasks: engine alone, accuracy tail alone (and its host half), the method's run_method -
where the step's time goes.  (Round 5 also ran the tasks in 2 / 4 / 8 chunks with the host half of a chunk's tail under the next
chunk's engine call: 57.6 / 59.3 / 76.6 ms against 56.2 ms in one piece - the tail is 4 ms, smaller launches cost more; not kept.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from src.methods.zero_shot.soft_kmeans import SOFT_KMEANS
from src.utils import CfgNode
from tclip_amd import engine, synth
K, T = 397, 1000
x_q, y_q = synth.make_query_tasks(T, K, seed=3)
x, y = x_q.cuda(), y_q.squeeze(2).cuda()
cfg = CfgNode(iter=20, iter_mm=0, num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30, use_softmax_feature=True, graph_matching=True, name_method="SOFT_KMEANS")


def best(fn, reps=5):
    fn(); torch.cuda.synchronize()
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t)
    return b * 1e3


res = engine.run_soft_kmeans(x, iters=20, temperature=30)
print(f"engine alone {best(lambda: engine.run_soft_kmeans(x, iters=20, temperature=30)):.2f} ms")
print(f"accuracy tail alone {best(lambda: engine.clustering_accuracy(x, res[-1], y)):.2f} ms")
h = engine.clustering_accuracy_begin(x, res[-1], y)
print(f"  of which host half {best(lambda: engine.clustering_accuracy_finish(h)):.2f} ms")
def step():
    m = SOFT_KMEANS(model=None, device=torch.device("cuda:0"), log_file=None, args=cfg)
    m.run_method(query=x, y_q=y)
    return m.get_logs()


print(f"run_method (engine + tail + logs): {best(step):.2f} ms", flush=True)
