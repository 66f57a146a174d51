"""Wall time of the engine on the other BASELINE shapes (run on the GPU box): python3 scripts/gpu_shapes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from tclip_amd import engine, synth


def timed(label, fn):
    fn(); torch.cuda.synchronize()
    t = time.time(); res = fn(); torch.cuda.synchronize(); dt = time.time() - t
    print(f"{label}: {dt:.3f} s  mm_iters[0]={res.mm_iters[0].tolist()[:6]}...", flush=True)


x, _ = synth.make_query_tasks(250, 1000, seed=5); x = x.cuda()
timed("K=1000 zero-shot soft, 2 batches x 125 tasks, 20x1000",
      lambda: engine.run_em_dirichlet(x, n_batches=2, iters=20, iter_mm=1000, lambd=200 * 75, hard=False))
x, _ = synth.make_query_tasks(200, 397, seed=5); x = x.cuda()
timed("K=397 zero-shot hard, 2 batches x 100 tasks, 10x1000",
      lambda: engine.run_em_dirichlet(x, n_batches=2, iters=10, iter_mm=1000, lambd=79 * 75, hard=True))
x, _ = synth.make_query_tasks(1000, 397, seed=5); x = x.cuda()
timed("K=397 zero-shot hard, 10 batches x 100 tasks, 10x1000 (configs[2], first method)",
      lambda: engine.run_em_dirichlet(x, n_batches=10, iters=10, iter_mm=1000, lambd=79 * 75, hard=True))
del x
x, _ = synth.make_query_tasks(100, 1000, seed=5, k_eff=5); xs, ys = synth.make_support(100, 1000, 4, seed=5)
x, xs, ys = x.cuda(), xs.cuda(), ys.squeeze(2).cuda()
timed("K=1000 few-shot 4-shot soft, 4 batches x 25 tasks, 20x1000 (configs[4] shape, 100 tasks)",
      lambda: engine.run_em_dirichlet(x, xs, ys, n_batches=4, iters=20, iter_mm=1000, lambd=200 * 75, hard=False))
del x, xs, ys
x, _ = synth.make_query_tasks(100, 100, seed=5, k_eff=5); xs, ys = synth.make_support(100, 100, 4, seed=5)
x, xs, ys = x.cuda(), xs.cuda(), ys.squeeze(2).cuda()
timed("K=100 few-shot 4-shot soft, 1 batch x 100 tasks, 20x1000",
      lambda: engine.run_em_dirichlet(x, xs, ys, n_batches=1, iters=20, iter_mm=1000, lambd=20 * 75, hard=False))
x, _ = synth.make_query_tasks(100, 10, seed=5); x = x.cuda()
timed("K=10 zero-shot soft, 10 batches x 10 tasks, 20x1000",
      lambda: engine.run_em_dirichlet(x, n_batches=10, iters=20, iter_mm=1000, lambd=2 * 75, hard=False))
x, _ = synth.make_query_tasks(8, 1000, seed=5, k_eff=5); xs, ys = synth.make_support(8, 1000, 4, seed=5)
x, xs, ys = x.cuda(), xs.cuda(), ys.squeeze(2).cuda()
timed("K=1000 few-shot 4-shot soft, 1 batch x 8 tasks, 20x1000 (configs[4] shape)",
      lambda: engine.run_em_dirichlet(x, xs, ys, n_batches=1, iters=20, iter_mm=1000, lambd=200 * 75, hard=False))


def timed_fn(label, fn, note):
    fn(); torch.cuda.synchronize()
    t = time.time(); fn(); torch.cuda.synchronize(); dt = time.time() - t
    print(f"{label}: {dt * 1e3:.1f} ms  {note(dt)}", flush=True)


T, K, Q = 1000, 397, 75
x, _ = synth.make_query_tasks(T, K, seed=6); x = x.cuda()
flop = lambda iters: iters * T * K * Q * K * 5.0          # distance (3 flop per term) + statistics (2 flop per term)
timed_fn("SOFT_KMEANS K=397, 1000 tasks, 20 iterations (configs[2] second method)",
         lambda: engine.run_soft_kmeans(x, iters=20, temperature=30),
         lambda dt: f"{T / dt:.0f} tasks/s, {flop(20) / dt / 1e12:.2f} TFLOP/s fp32 of the two contractions")
timed_fn("HARD_KMEANS K=397, 1000 tasks, 10 iterations",
         lambda: engine.run_hard_kmeans(x, iters=10, n_batches=10),
         lambda dt: f"{T / dt:.0f} tasks/s, {flop(10) / dt / 1e12:.2f} TFLOP/s")
T, K = 1000, 100
x, _ = synth.make_query_tasks(T, K, seed=6, k_eff=5); xs, ys = synth.make_support(T, K, 4, seed=6)
x, xs, ys = x.cuda(), xs.cuda(), ys.squeeze(2).cuda()
timed_fn("PADDLE K=100 4-shot, 1000 tasks, 20 iterations",
         lambda: engine.run_paddle(x, xs, ys, iters=20, lambd=5.0),
         lambda dt: f"{T / dt:.0f} tasks/s")
T, K = 1000, 397
x, _ = synth.make_query_tasks(T, K, seed=6); x = x.cuda()
timed_fn("EM_GAUSSIAN K=397, 1000 tasks, 20 iterations",
         lambda: engine.run_em_gaussian(x, iters=20, temperature=30, lambd=79 * 75), lambda dt: f"{T / dt:.0f} tasks/s")
timed_fn("EM_GAUSSIAN_COV K=397, 1000 tasks, 20 iterations",
         lambda: engine.run_em_gaussian_cov(x, iters=20, lambd=79 * 75), lambda dt: f"{T / dt:.0f} tasks/s")
timed_fn("KL_KMEANS K=397, 1000 tasks, 10 iterations",
         lambda: engine.run_kl_kmeans(x, iters=10, n_batches=10), lambda dt: f"{T / dt:.0f} tasks/s")
T, K = 1000, 100
x, _ = synth.make_query_tasks(T, K, seed=6, k_eff=5); xs, ys = synth.make_support(T, K, 4, seed=6)
x, xs, ys = x.cuda(), xs.cuda(), ys.squeeze(2).cuda()
timed_fn("BDCSPN K=100 4-shot, 1000 tasks (L2N)",
         lambda: engine.run_bdcspn(x, xs, ys, temp=30.0, norm_type="L2N"), lambda dt: f"{T / dt:.0f} tasks/s")
