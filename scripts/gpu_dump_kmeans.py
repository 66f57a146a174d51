"""Exact fuzz of the log-based k-means methods, part 1 (GPU box): python3 scripts/gpu_dump_kmeans.py [n_cases] [seed]
The GPU pool's host torch does not run the MKL vsLn kernel the reference fixtures were made with, so
the engine's outputs cannot be compared bit for bit with torch there.  This script only RUNS the
engine on seeded random problems and writes SHA-1 digests of every output array to
gpurun_out/kmeans_dump_<seed>.json; scripts/check_kmeans_dump.py recomputes the same problems with
the torch-eager oracle on the fixture host and compares the digests."""
import hashlib, json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))
import torch
from tclip_amd import engine, synth
from fuzz_kmeans_cases import cases


def digest(t):
    return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = {}
for c in cases(n_cases, seed):
    x_q, _ = synth.make_query_tasks(c["N"], c["K"], seed=c["data_seed"], k_eff=min(4, c["K"]))
    x_s, y_s = synth.make_support(c["N"], c["K"], c["shots"], seed=c["data_seed"] + 1000)
    xc = x_q.cuda()
    lam = int(c["K"] / 5) * 75
    r = {"inputs": {"x_q": digest(x_q), "x_s": digest(x_s)}}
    u, v, w, p = engine.run_em_gaussian(xc, iters=c["iters"], temperature=30, lambd=lam)
    r["emg"] = {"u": digest(u), "v": digest(v), "w": digest(w)}
    u, v, w, s, p = engine.run_em_gaussian_cov(xc, iters=c["iters"], lambd=lam)
    r["cov"] = {"u": digest(u), "v": digest(v), "w": digest(w), "s": digest(s)}
    u, w, p, cr = engine.run_kl_kmeans(xc, iters=c["iters"])
    r["klk"] = {"u": digest(u), "w": digest(w)}
    u, v, w, p = engine.run_paddle(xc, x_s.cuda(), y_s.squeeze(2).cuda(), iters=c["iters"], lambd=c["paddle_lambd"])
    r["paddle"] = {"u": digest(u), "v": digest(v), "w": digest(w)}
    nt = ("UN", "L2N", "CL2N")[c["case"] % 3]
    pr, u, p = engine.run_bdcspn(xc, x_s.cuda(), y_s.squeeze(2).cuda(), temp=30.0, norm_type=nt)
    r["bdcspn"] = {"prototypes": digest(pr), "u": digest(u)}
    out[str(c["case"])] = r
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
path = os.path.join(ROOT, "gpurun_out", f"kmeans_dump_{seed}.json")
json.dump({"n_cases": n_cases, "seed": seed, "digests": out}, open(path, "w"))
print("wrote", path, len(out), "cases")
