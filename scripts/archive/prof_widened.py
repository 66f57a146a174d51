"""The widened methods once each, for a kernel-time breakdown: python3 scripts/prof_widened.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from tclip_amd import engine, synth
T, K = 1000, 397
x, _ = synth.make_query_tasks(T, K, seed=6); x = x.cuda()
engine.run_em_gaussian_cov(x, iters=20, lambd=79 * 75)
engine.run_kl_kmeans(x, iters=10, n_batches=10)
T, K = 1000, 100
x, _ = synth.make_query_tasks(T, K, seed=6, k_eff=5); xs, ys = synth.make_support(T, K, 4, seed=6)
engine.run_bdcspn(x.cuda(), xs.cuda(), ys.squeeze(2).cuda(), temp=30.0, norm_type="CL2N")
torch.cuda.synchronize()
