#!/usr/bin/env python3
"""A/B of several builds of libtclip.so on the k-means family (SOFT_KMEANS, HARD_KMEANS, EM_GAUSSIAN, EM_GAUSSIAN_COV, KL_KMEANS at
K = 397 and K = 100, 1000 tasks): milliseconds per engine call and a digest of the results, one child process per library
(TCLIP_LIB), as scripts/gpu_ab_libs.py does for EM-Dirichlet.

    python scripts/gpu_ab_kmeans_libs.py orig gpurun_variants/x.so ...
"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
    import torch
    from tclip_amd import engine, synth
    out = {}
    for K, T in ((397, 1000), (100, 1000)):
        x, _ = synth.make_query_tasks(T, K, seed=6); x = x.cuda()
        lambd = int(K / 5) * 75
        calls = {"SOFT_KMEANS": lambda: engine.run_soft_kmeans(x, iters=20, temperature=30),
                 "HARD_KMEANS": lambda: engine.run_hard_kmeans(x, iters=10),
                 "EM_GAUSSIAN": lambda: engine.run_em_gaussian(x, iters=20, temperature=30, lambd=lambd),
                 "EM_GAUSSIAN_COV": lambda: engine.run_em_gaussian_cov(x, iters=20, lambd=lambd),
                 "KL_KMEANS": lambda: engine.run_kl_kmeans(x, iters=10)}
        for name, fn in calls.items():
            best = 1e9
            for rep in range(4):
                torch.cuda.synchronize(); t = time.time()
                res = fn()
                torch.cuda.synchronize(); best = min(best, time.time() - t)
            h = hashlib.sha1()
            for r in res:
                h.update(r.cpu().numpy().tobytes())
            out[f"{name} K={K} T={T}"] = (best * 1e3, h.hexdigest()[:12])
    print("RESULT " + json.dumps(out), flush=True)


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "--child":
        return child()
    libs = argv or ["orig"]
    results = {}
    for lib in libs:
        env = dict(os.environ)
        env["TCLIP_LIB"] = "" if lib == "orig" else os.path.realpath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(f"== {lib}: FAILED\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}", flush=True)
            continue
        results[lib] = json.loads(line[0][7:])
    first = libs[0]
    for shape in results.get(first, {}):
        row = []
        for lib in libs:
            if lib not in results:
                continue
            t, h = results[lib][shape]
            t0, h0 = results[first][shape]
            row.append(f"{os.path.basename(lib)}: {t:.2f} ms ({t / t0:.3f}x) {'same' if h == h0 else 'DIFFERENT ' + h}")
        print(f"{shape:28s} " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
