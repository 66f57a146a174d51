"""Randomised parity sweep on the GPU box: python3 scripts/gpu_fuzz.py [n_cases] [seed] [full|large|mid]
("large": long rows, K = 300 ... 1024, short schedules - the register-heavy instantiations)
("full": the reference's whole 20 x 1000 / 10 x 1000 schedule on small problems, which exercises the
stop test at every checkpoint, dead rows, their cache and the limit-cycle shortcut)
("mid": the whole schedule at K = 47 ... 128, one or two tasks per batch: dead rows by the hundred per task, whose limit cycles
have other periods than at K <= 40 - minutes of CPU oracle per case)
Random (K, Q, tasks, batches, hard, few-shot, schedule) against the C++ oracle, bit for bit, each
case run twice (run-to-run determinism)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import numpy as np
import torch
from oracle import c_oracle
from tclip_amd import engine, synth

if "TCLIP_FUZZ_ROWSET_MIN_ROWS" in os.environ:      # 0: force the 32-lanes-per-row layout of the MM kernels
    engine.debug_set_rowset_min_rows(int(os.environ["TCLIP_FUZZ_ROWSET_MIN_ROWS"]))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
full = len(sys.argv) > 3 and sys.argv[3] == "full"
large = len(sys.argv) > 3 and sys.argv[3] == "large"
mid = len(sys.argv) > 3 and sys.argv[3] == "mid"
full = full or mid
bad = 0
t0 = time.time()
for case in range(n_cases):
    K = rng.choice([2, 3, 5, 8, 9, 17, 31, 32, 33, 40, 64, 65, 96, 100, 101, 129, 160, 200, 257, 300])
    Q = rng.choice([1, 2, 5, 17, 20, 64, 75, 130])
    few = rng.random() < 0.3
    hard = rng.random() < 0.4
    B = rng.randint(1, 4)
    if full:
        K = rng.choice([2, 3, 5, 7, 8, 9, 10, 12, 16, 20, 33, 40])
    if mid:
        K, B, Q, few = rng.choice([47, 64, 65, 100, 101, 128]), rng.randint(1, 2), rng.choice([20, 75]), False
    if large:
        K, B = rng.choice([300, 397, 450, 512, 600, 640, 777, 900, 1000, 1024]), rng.randint(1, 2)
    budget = (4e8 if (full and not mid) else (8e8 if large else 1.5e8)) / (K * K)                # element-updates the CPU oracle can afford
    iter_mm = 1000 if full else rng.choice([30, 51, 60, 101, 120, 151, 230])
    iters = (10 if hard else 20) if full else rng.randint(2, 4)
    N = max(1, min(6, int(budget / (iter_mm * iters * B))))
    if mid:
        N = rng.randint(1, 2)
    lambd = max(1, int(K / 5)) * Q
    x_q, _ = synth.make_query_tasks(B * N, K, seed=1000 + case, n_query=Q, k_eff=(min(3, K) if few else None))
    x_s = y_s = None
    if few:
        x_s, y_s = synth.make_support(B * N, K, rng.randint(1, 3), seed=2000 + case)
    args = dict(n_batches=B, iters=iters, iter_mm=iter_mm, lambd=lambd, hard=hard)
    runs = [engine.run_em_dirichlet(x_q.cuda(), x_s.cuda() if few else None, y_s.squeeze(2).cuda() if few else None, **args)
            for _ in range(2)]
    torch.cuda.synchronize()
    ok = torch.equal(runs[0].alpha, runs[1].alpha) and torch.equal(runs[0].u, runs[1].u)
    for b in range(B):
        sl = slice(b * N, (b + 1) * N)
        ref = c_oracle.run(x_q[sl].numpy(), x_s[sl].numpy() if few else None, y_s[sl].numpy() if few else None,
                           iters=iters, iter_mm=iter_mm, lambd=lambd, hard=hard)
        r = runs[0]
        ok = ok and np.array_equal(r.mm_iters[b].cpu().numpy(), ref["mm_iters"]) \
            and np.array_equal(r.alpha[sl].cpu().numpy(), ref["alpha"]) and np.array_equal(r.u[sl].cpu().numpy(), ref["u"]) \
            and np.array_equal(r.v[sl].cpu().numpy(), ref["v"])
    bad += not ok
    print(f"case {case}: K={K} Q={Q} N={N} B={B} few={few} hard={hard} iters={iters} iter_mm={iter_mm} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{n_cases - bad}/{n_cases} cases identical to the oracle and repeatable, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
