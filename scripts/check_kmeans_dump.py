"""Exact fuzz of the log-based k-means methods, part 2 (fixture host, CPU): python3 scripts/check_kmeans_dump.py gpurun_out/kmeans_dump_<seed>.json
Recomputes every problem of the dump with the torch-eager oracle (oracle/ref_torch.py, the
reference's op sequence; this host's torch is the one the reference fixtures were made with) and
compares SHA-1 digests of the outputs with those of the engine: bit-exact or it counts as a mismatch."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from oracle import ref_torch
from tclip_amd import synth
from fuzz_kmeans_cases import cases

torch.set_num_threads(min(8, torch.get_num_threads()))


def digest(t):
    return hashlib.sha1(t.detach().contiguous().numpy().tobytes()).hexdigest()


d = json.load(open(sys.argv[1]))
bad = 0
for c in cases(d["n_cases"], d["seed"]):
    got = d["digests"][str(c["case"])]
    x_q, _ = synth.make_query_tasks(c["N"], c["K"], seed=c["data_seed"], k_eff=min(4, c["K"]))
    x_s, y_s = synth.make_support(c["N"], c["K"], c["shots"], seed=c["data_seed"] + 1000)
    K, lam = c["K"], int(c["K"] / 5) * 75
    inputs = got.pop("inputs")
    if digest(x_q) != inputs["x_q"] or digest(x_s) != inputs["x_s"]:
        print(f"case {c['case']}: the two hosts generated different inputs -> not comparable", flush=True)
        bad += 1
        continue
    t = {"emg": ref_torch.run_em_gaussian(x_q, n_class=K, iters=c["iters"], temperature=30, lambd=lam),
         "cov": ref_torch.run_em_gaussian_cov(x_q, n_class=K, iters=c["iters"], lambd=lam),
         "klk": ref_torch.run_kl_kmeans(x_q, n_class=K, iters=c["iters"]),
         "paddle": ref_torch.run_paddle(x_q, x_s, y_s, n_class=K, iters=c["iters"], lambd=c["paddle_lambd"])}
    if "bdcspn" in got:
        t["bdcspn"] = ref_torch.run_bdcspn(x_q, x_s, y_s, n_class=K, temp=30.0, norm_type=("UN", "L2N", "CL2N")[c["case"] % 3])
    res = {m: all(digest(t[m][a]) == h for a, h in got[m].items()) for m in got}
    ok = all(res.values())
    bad += not ok
    print(f"case {c['case']}: K={K} N={c['N']} iters={c['iters']} shots={c['shots']} {res} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{d['n_cases'] - bad}/{d['n_cases']} cases identical")
sys.exit(1 if bad else 0)
