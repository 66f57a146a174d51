"""Randomised parity sweep of the k-means family on the GPU box: python3 scripts/gpu_fuzz_kmeans.py [n_cases] [seed]
SOFT_KMEANS and HARD_KMEANS have no logarithm in their loop and must equal the torch-eager oracle
bit for bit when the host's torch runs the AVX-512 kernels with at most 8 threads.  EM_GAUSSIAN(_COV), KL_KMEANS and
PADDLE go through torch.log = MKL vsLn on the host side, whose kernel choice differs between hosts
(the GPU pool's host does not reproduce the fixture host's logs), so they are compared to 1e-4 here;
their bit-exactness is pinned by the fixtures the reference produced on the fixture host and by
tests/test_gpu_digests_kmeans.py (digests of the torch-eager restatement made on the fixture host, tests/golden/digests_kmeans.json,
on platform-independent inputs: exact).  EM_GAUSSIAN_COV multiplies by inverse variances of up to 2e15, so a 1-ulp
difference of the host's log can exceed the tolerance: its column is printed but not counted."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))
import torch
from oracle import ref_torch
from tclip_amd import engine, synth

torch.set_num_threads(min(8, torch.get_num_threads()))      # the reference changes with 16+ threads (scripts/host_threads_check.py)
# torch on this host is the comparison target: its reduction orders are the AVX-512 ones only there
exact_host = torch.backends.cpu.get_cpu_capability() == "AVX512"
print("host torch capability:", torch.backends.cpu.get_cpu_capability(), "-> exact comparison" if exact_host else "-> 1e-5 comparison")
close = lambda a, b: torch.allclose(a, b, rtol=1e-4, atol=1e-7, equal_nan=True)
same = torch.equal if exact_host else close
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    K = rng.choice([2, 3, 7, 8, 9, 16, 21, 31, 32, 33, 47, 64, 65, 100, 101, 130, 200])
    N, iters = rng.randint(1, 5), rng.randint(1, 6)
    x_q, _ = synth.make_query_tasks(N, K, seed=3000 + case, k_eff=min(4, K))
    xc = x_q.cuda()
    u, w, p = engine.run_soft_kmeans(xc, iters=iters, temperature=30)
    t = ref_torch.run_soft_kmeans(x_q, n_class=K, iters=iters, temperature=30)
    ok_skm = same(u.cpu(), t["u"]) and same(w.cpu(), t["w"])
    u, w, p, c = engine.run_hard_kmeans(xc, iters=iters)
    t = ref_torch.run_hard_kmeans(x_q, n_class=K, iters=iters)
    ok_hkm = same(u.cpu(), t["u"]) and same(w.cpu(), t["w"]) and torch.equal(p.cpu().long(), t["labels"][-1])
    u, v, w, p = engine.run_em_gaussian(xc, iters=iters, temperature=30, lambd=int(K / 5) * 75)
    t = ref_torch.run_em_gaussian(x_q, n_class=K, iters=iters, temperature=30, lambd=int(K / 5) * 75)
    ok_emg = close(u.cpu(), t["u"]) and close(w.cpu(), t["w"])
    u, v, w, sc, p = engine.run_em_gaussian_cov(xc, iters=iters, lambd=int(K / 5) * 75)
    t = ref_torch.run_em_gaussian_cov(x_q, n_class=K, iters=iters, lambd=int(K / 5) * 75)
    ok_cov = close(u.cpu(), t["u"]) and close(w.cpu(), t["w"]) and close(sc.cpu(), t["s"])
    u, w, p, c = engine.run_kl_kmeans(xc, iters=iters)
    t = ref_torch.run_kl_kmeans(x_q, n_class=K, iters=iters)
    ok_klk = close(u.cpu(), t["u"]) and close(w.cpu(), t["w"])
    shots = rng.randint(1, 3)
    x_s, y_s = synth.make_support(N, K, shots, seed=4000 + case)
    lam = rng.choice([0.0, 3.0, 12.5])
    u, v, w, p = engine.run_paddle(xc, x_s.cuda(), y_s.squeeze(2).cuda(), iters=iters, lambd=lam)
    t = ref_torch.run_paddle(x_q, x_s, y_s, n_class=K, iters=iters, lambd=lam)
    ok_pad = close(u.cpu(), t["u"]) and close(w.cpu(), t["w"])
    ok = ok_skm and ok_hkm and ok_emg and ok_pad and ok_klk      # cov: informational (see the docstring)
    bad += not ok
    print(f"case {case}: K={K} N={N} iters={iters} shots={shots} lambd={lam} skm={ok_skm} hkm={ok_hkm} emg={ok_emg} cov={ok_cov} klk={ok_klk} paddle={ok_pad} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{n_cases - bad}/{n_cases} cases passed")
sys.exit(1 if bad else 0)
