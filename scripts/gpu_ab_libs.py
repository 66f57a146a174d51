#!/usr/bin/env python3
"""A/B of several builds of libtclip.so on the bench shapes: seconds per engine call and a digest of the results, one child
process per library (TCLIP_LIB), so that a kernel change is timed against its predecessor on the SAME box and checked to
return the same bits in the same call.

    python scripts/gpu_ab_libs.py orig gpurun_variants/base.so ... [-- K B N iters [hard [shots]] ...]
"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = [(1000, 3, 125, 20, 0, 0), (100, 10, 100, 20, 0, 0), (397, 4, 100, 10, 1, 0), (1000, 2, 12, 6, 0, 1)]


def child(shapes):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
    import torch
    from tclip_amd import engine, synth
    out = {}
    for K, B, N, iters, hard, shots in shapes:
        x_q, _ = synth.make_query_tasks(B * N, K, seed=3, k_eff=(5 if shots else None))
        x_q = x_q.cuda()
        x_s = y_s = None
        if shots:
            x_s, y_s = synth.make_support(B * N, K, shots, seed=3)
            x_s, y_s = x_s.cuda(), y_s.squeeze(2).cuda()
        best = 1e9
        for rep in range(int(os.environ.get("AB_REPS", "3"))):
            torch.cuda.synchronize(); t = time.time()
            res = engine.run_em_dirichlet(x_q, x_s, y_s, n_batches=B, iters=iters, iter_mm=1000, lambd=int(K / 5) * 75, hard=bool(hard))
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        h = hashlib.sha1()
        for n in ("alpha", "u", "v", "mm_iters", "criterions"):
            h.update(getattr(res, n).cpu().numpy().tobytes())
        out[f"K={K} B={B} N={N} it={iters} hard={hard} shots={shots}"] = (best, h.hexdigest()[:12])
    print("RESULT " + json.dumps(out), flush=True)


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "--child":
        return child(json.loads(argv[1]))
    shapes = DEFAULT
    if "--" in argv:
        i = argv.index("--")
        nums = [int(v) for v in argv[i + 1:]]
        argv = argv[:i]
        shapes = [tuple((nums[j:j + 6] + [0, 0])[:6]) for j in range(0, len(nums), 6)]
    libs = argv or ["orig"]
    results = {}
    for lib in libs:
        env = dict(os.environ)
        env["TCLIP_LIB"] = "" if lib == "orig" else os.path.realpath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", json.dumps(shapes)], env=env, capture_output=True, text=True)
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
            row.append(f"{os.path.basename(lib)}: {t:.3f}s ({t / t0:.3f}x) {'same' if h == h0 else 'DIFFERENT ' + h}")
        print(f"{shape}  " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
