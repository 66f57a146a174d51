"""Where a bench step spends its time outside the engine: python3 scripts/gpu_step_breakdown.py [k100|k1000]"""
import os, sys, time, random
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import numpy as np, torch
import bench
from src.eval_zero_shot import Evaluator_zero_shot
from src.utils import CfgNode
from tclip_amd import engine, synth
w = bench.SECONDARY if (len(sys.argv) > 1 and sys.argv[1] == "k100") else bench.HEADLINE
K = w["K"]; dev = torch.device("cuda:0")
feats, labels = synth.make_feature_table(K, w["rows_per_class"], seed=2020)
n_tasks = w["tasks_per_batch"] * w["batches_per_gpu"]
cfg = CfgNode(iter=20, iter_mm=1000, num_classes_test=K, n_class=K, n_query=75, k_eff=5, T=30, use_softmax_feature=True,
              graph_matching=True, shots=0, number_tasks=n_tasks, batch_size=w["tasks_per_batch"], name_method="EM_DIRICHLET")
ev = Evaluator_zero_shot(device=dev, args=cfg, log_file=None)
random.seed(2020); np.random.seed(2020); torch.manual_seed(2020)
idx = ev.sample_indices(labels.numpy())
table, lab = feats.to(dev), labels.to(dev)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = sync()
    my = idx.reshape(-1)
    x_q = engine.gather_rows(table, my).view(n_tasks, 75, K); y_q = lab[my.to(dev)].view(n_tasks, 75)
    t1 = sync()
    res = engine.run_em_dirichlet(x_q, n_batches=w["batches_per_gpu"], iters=20, iter_mm=1000, lambd=int(K / 5) * 75)
    t2 = sync()
    mm = res.mm_iters.cpu(); cr = res.criterions.cpu()
    t3 = sync()
    acc, newp = engine.clustering_accuracy(x_q, res.preds, y_q)
    t4 = sync()
    whole0 = sync(); ev.evaluate_tasks(None, table, lab, indices=idx); whole1 = sync()
    print(f"{w['name']}: gather {1e3*(t1-t0):.1f} ms  engine {1e3*(t2-t1):.1f}  logs {1e3*(t3-t2):.1f}  accuracy tail {1e3*(t4-t3):.1f}   evaluate_tasks {1e3*(whole1-whole0):.1f}")
