#!/usr/bin/env python3
"""Headline benchmark: transductive tasks/sec of the EM-Dirichlet hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline workload = the north-star shape, BASELINE.json configs[3] as it lands on one GPU:
zero-shot EM-Dirichlet, K=1000 classes (ImageNet-sized), 75 queries, reference batches of 125 tasks,
iter=20, iter_mm=1000, fp32.  configs[3] is 10 000 tasks = 80 batches over 8 GPUs, i.e. 10 batches
(1 250 tasks) per GPU; `--gpus N` runs 10*N batches (1 250*N tasks) dealt round-robin to the N ranks,
so N=8 IS configs[3] and N=1 is its per-GPU share ("scaling": "weak": the work per GPU is fixed;
10 000 tasks on one GPU would take two minutes per step, beyond what the driver's 25 steps allow).

One "step" = one pass of the task-batch loop (reference src/eval_zero_shot.py:140-187) through
`Evaluator_zero_shot.evaluate_tasks` of this package: device-side gather of the task rows from the
feature table resident in HBM (indices drawn beforehand, identically on every rank, with the
reference's sampler), one engine call for the rank's batches, the accuracy tail (device prototypes,
host assignment), and for N>1 the single RCCL all_gather of the per-task accuracies onto rank 0.

Extra objects on the JSON line:
  roofline     the dominant kernel k_mm_live, timed live with HIP events around each of its launches
               on the streams they run on (independent batches use a few internal streams, so
               launches overlap: `achieved` divides by the time during which at least one launch
               was running, `avg_launch_ms` is the plain mean launch duration).  The path is fp32
               vector-ALU bound, so the bound is "valu": achieved = 48
               flop-equivalents x element-updates executed / kernel time against 157.3 TFLOP/s.
               `traffic` = HBM-side bytes per launch of that kernel from the PMC passes recorded in
               profiles/pmc_current.json (FETCH_SIZE + WRITE_SIZE, calibrated on the dword-per-lane
               copy kernel of the same run), null when that file does not describe this build.
  secondary    the K=100 workload of round 1 (BASELINE configs[1]), a few steps, same roofline fields.
  cpu_baseline the torch-eager CPU restatement of the reference loop (oracle/ref_torch.py, kind
               "port") on this host: K=1000 is ~6 minutes per task on 8 cores, so, as SURVEY.md 8d
               prescribes, MM iterations of the first and of a later outer iteration plus one M/E-step are timed
               on a 2-task batch and extrapolated over the MM schedule the GPU run recorded.
"""
import argparse
import json
import os
import sys
import time

# Independent batches run on up to three HIP streams; HIP gives a process four hardware queues by default and RCCL's
# own streams take some of them under torch.distributed, after which the engine's streams share a queue and
# serialise (measured with one rank under torch.distributed.run: K=100 1 641 tasks/s against 2 119 with 8 queues,
# which is also the rate without a process group).  Read when the HIP runtime initialises, so set before torch.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))

import torch  # noqa: E402

N_QUERY = 75
ITERS = 20
ITER_MM = 1000
HEADLINE = dict(name="k1000", K=1000, tasks_per_batch=125, batches_per_gpu=10, rows_per_class=50,
                text="EM-Dirichlet zero-shot, K=1000 (imagenet-sized), 75-query, 1250 tasks per GPU as 10 batches of 125, "
                     "iter=20, iter_mm=1000 (BASELINE.json configs[3]: 10 000 tasks on 8 GPUs = this per GPU)")
SECONDARY = dict(name="k100", K=100, tasks_per_batch=100, batches_per_gpu=10, rows_per_class=40,
                 text="EM-Dirichlet zero-shot, K=100 (caltech101-sized), 75-query, 1000 tasks per GPU as 10 batches of 100, "
                      "iter=20, iter_mm=1000 (BASELINE.json configs[1])")
FLOP_EQ_PER_UPDATE = 48.0
PEAK_VALU_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_current.json")


_CPU_SNIPPET = r"""
import json, os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "transductive-clip_amd"))
import torch
from oracle import ref_torch
from tclip_amd import synth
torch.set_num_threads({threads})
x_q, _ = synth.make_query_tasks({n_tasks}, {K}, seed=0)
out = {{}}
if {warm}:
    ref_torch.run(x_q, n_class={K}, iters=1, iter_mm=2, lambd={lambd}, hard=False)      # thread pool, allocator
for iters, mm in {mm_list}:
    r = ref_torch.run(x_q, n_class={K}, iters=iters, iter_mm=mm, lambd={lambd}, hard=False)
    out[str(iters) + "x" + str(mm)] = r["seconds"]
print(json.dumps({{"seconds": out, "threads": torch.get_num_threads(), "torch": torch.__version__}}))
"""


def cpu_baseline(w, mm_schedule, budget_s=300):
    """Reference loop on the host CPU (kind "port": oracle/ref_torch.py issues the reference's own
    torch op sequence).  mm_schedule: MM iterations per outer iteration the GPU run recorded for
    batch 0.  Child process under a time budget: an oversubscribed host cannot stall the bench."""
    import subprocess
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    threads = max(1, min(usable, 16))
    K = w["K"]
    if K >= 397:          # extrapolated from one and two outer iterations with 101 and with 301 MM iterations each
        n_tasks, mm_list = 2, [(1, 101), (1, 301), (2, 101), (2, 301)]
    else:                 # affordable in full: one whole batch, whole schedule
        n_tasks, mm_list = w["tasks_per_batch"], [(ITERS, ITER_MM)]
    code = _CPU_SNIPPET.format(root=ROOT, threads=threads, n_tasks=n_tasks, K=K, mm_list=mm_list,
                               lambd=int(K / 5) * N_QUERY, warm=len(mm_list) > 1)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s)
        info = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # timeout or failure: report it, never fake a number
        return {"value": None, "unit": "tasks/s", "cores": threads, "kind": "port",
                "sample": f"not measured: {type(e).__name__} within {budget_s}s budget"}
    secs = info["seconds"]
    if len(mm_list) == 1:
        total = secs[f"{ITERS}x{ITER_MM}"]
        return {"value": n_tasks / total, "unit": "tasks/s", "cores": info["threads"], "kind": "port",
                "sample": f"{n_tasks} tasks (one batch) of the same K={K}, 75-query workload, full {ITERS}x{ITER_MM} "
                          f"schedule, torch {info['torch']} CPU eager, {total:.1f}s"}
    # an MM iteration costs differently in the first outer iteration (every class alive, alpha near 1) and in the
    # later ones (few live classes, large alpha), so both regimes are timed: all for the n_tasks batch
    a1, b1, a2, b2 = secs["1x101"], secs["1x301"], secs["2x101"], secs["2x301"]
    mm_first = (b1 - a1) / 200.0                                   # one MM iteration, first outer iteration
    mm_later = max((b2 - a2) / 200.0 - mm_first, 0.0)              # one MM iteration, second outer iteration
    per_me = max(a1 - 101 * mm_first, 0.0)                         # M-step statistics + E-step + criterion, once per outer iteration
    total = len(mm_schedule) * per_me + mm_schedule[0] * mm_first + sum(mm_schedule[1:]) * mm_later
    return {"value": n_tasks / total, "unit": "tasks/s", "cores": info["threads"], "kind": "port",
            "sample": f"{n_tasks}-task batch of the same K={K}, 75-query workload, timed: 1 outer iteration with 101 ({a1:.1f}s) and "
                      f"301 ({b1:.1f}s) MM iterations, 2 outer iterations with 101 ({a2:.1f}s) and 301 ({b2:.1f}s) -> "
                      f"{1e3 * mm_first:.1f} / {1e3 * mm_later:.1f} ms per MM iteration in the first / a later outer iteration, "
                      f"{per_me:.2f}s per M/E-step; extrapolated over the recorded schedule ({mm_schedule[0]} + "
                      f"{int(sum(mm_schedule[1:]))} MM iterations in {len(mm_schedule)} outer iterations) = {total:.0f}s "
                      f"(SURVEY.md 8d); torch {info['torch']} CPU eager",
            "extrapolated": True}


def load_pmc():
    try:
        with open(PMC_FILE) as f:
            return json.load(f)
    except Exception:
        return None


def roofline_of(prof, steps, K, pmc_key):
    """prof = engine.profile_collect() of `steps` timed steps."""
    mm_ms, mm_launch_sum, mm_launches, updates = prof
    achieved = FLOP_EQ_PER_UPDATE * updates / (mm_ms * 1e-3) / 1e12 if mm_ms > 0 else 0.0
    # algorithmic HBM bytes of the MM launches: every listed row is read (alpha, y) and written (alpha)
    # once per launch = 12 bytes per element per <=51-iteration launch
    rows_bytes = 12.0 * updates / 50.0
    hbm_gbs = rows_bytes / (mm_ms * 1e-3) / 1e9 if mm_ms > 0 else 0.0
    out = {"bound": "valu", "kernel": "k_mm_live", "achieved": achieved, "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
           "frac": achieved / PEAK_VALU_TFLOPS, "traffic": None,
           "flop_eq_per_element_update": FLOP_EQ_PER_UPDATE,
           "element_updates_executed_per_step": updates / steps,
           "element_updates_per_s": updates / (mm_ms * 1e-3) if mm_ms > 0 else 0.0,
           "element_updates_per_launch": updates / max(mm_launches, 1),
           "kernel_busy_ms_per_step": mm_ms / steps, "launches_per_step": mm_launches / steps,
           "avg_launch_ms": mm_launch_sum / max(mm_launches, 1),
           "launch_overlap": mm_launch_sum / mm_ms if mm_ms > 0 else 0.0,
           "algorithmic_bytes_per_launch": rows_bytes / max(mm_launches, 1),
           "hbm": {"bound": "hbm", "achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                   "frac": hbm_gbs / PEAK_HBM_GBS, "algorithmic_bytes_per_step": rows_bytes / steps}}
    pmc = load_pmc()
    if pmc and pmc_key in pmc:
        e = pmc[pmc_key]
        out["traffic"] = e.get("traffic_bytes_per_launch")
        out["traffic_source"] = {k: e.get(k) for k in ("file", "workload", "fetch_bytes_per_launch", "write_bytes_per_launch",
                                                       "algorithmic_bytes_per_launch", "calibration", "lane_instr_per_update",
                                                       "wait_frac", "scratch_bytes_per_lane", "commit")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--workload", choices=["k1000", "k100"], default="k1000", help="headline workload (k100: round-1 shape)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): 10 batches per GPU, N = 8 is configs[3]; strong: the whole configs[3] job (80 batches of 125 "
                         "K = 1000 tasks) on N GPUs - 90 s per step on one GPU")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to test the N>1 path on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="testing: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = "RANK" in os.environ          # launched by torch.distributed.run (any world size)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the EM-Dirichlet engine has no CPU path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import random

    import numpy as np
    from src.eval_zero_shot import Evaluator_zero_shot
    from src.utils import CfgNode
    from tclip_amd import engine, synth

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def prepare(w, n_ranks):
        """Feature table in HBM + the index stream of every batch (same on every rank)."""
        K = w["K"]
        feats, labels = synth.make_feature_table(K, w["rows_per_class"], seed=2020)
        n_tasks = w["tasks_per_batch"] * w["batches_per_gpu"] * (8 if args.scaling == "strong" else n_ranks)
        cfg = CfgNode(iter=ITERS, iter_mm=ITER_MM, num_classes_test=K, n_class=K, n_query=N_QUERY, k_eff=5, T=30,
                      use_softmax_feature=True, graph_matching=True, shots=0, number_tasks=n_tasks,
                      batch_size=w["tasks_per_batch"], name_method="EM_DIRICHLET", used_test_set="test")
        ev = Evaluator_zero_shot(device=dev, args=cfg, log_file=None)
        random.seed(2020); np.random.seed(2020); torch.manual_seed(2020)     # main.py:42-46
        idx = ev.sample_indices(labels.numpy())
        return ev, feats.to(dev), labels.to(dev), idx

    def run_steps(ev, table, labels, idx, warmup, steps):
        for _ in range(warmup):
            ev.evaluate_tasks(None, table, labels, indices=idx)
        fence()
        engine.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            acc_mean, _ = ev.evaluate_tasks(None, table, labels, indices=idx)
        fence()
        elapsed = time.perf_counter() - t0
        prof = engine.profile_collect()
        engine.profile_enable(False)
        if dist_on:
            t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, prof, acc_mean

    head = HEADLINE if args.workload == "k1000" else SECONDARY
    ev, table, labels, idx = prepare(head, world)
    elapsed, prof, acc_mean = run_steps(ev, table, labels, idx, args.warmup, args.steps)
    steps = max(args.steps, 1)
    mm_iters = ev.last_method.mm_iters                  # (local batches, iters)
    line = None
    if rank == 0:
        job = (8 if args.scaling == "strong" else world) * head["tasks_per_batch"] * head["batches_per_gpu"]
        tasks = job * steps
        K = head["K"]
        roof = roofline_of(prof, steps, K, head["name"])
        roof["element_updates_reference_semantics_per_step"] = float(K) ** 2 * head["tasks_per_batch"] * float(mm_iters.sum())
        line = {
            "metric": "transductive tasks/sec (75-query EM-Dirichlet)",
            "value": tasks / elapsed, "unit": "tasks/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": head["text"], "n_class": K, "n_query": N_QUERY, "tasks_per_batch": head["tasks_per_batch"],
                       "batches_per_gpu": head["batches_per_gpu"] * (8 // world if args.scaling == "strong" else 1), "tasks_total": job,
                       "parallelism": f"batch-sharded x{world}, one all_gather of per-task accuracies",
                       "path": "Evaluator_zero_shot.evaluate_tasks (device gather from a 50-rows-per-class synthetic table, "
                               "engine, accuracy tail, gather)",
                       "mean_accuracy": float(acc_mean), "mm_iters_batch0": mm_iters[0].tolist()},
            "roofline": roof,
        }
    del ev, table, labels, idx
    torch.cuda.empty_cache()

    if world == 1 and not args.no_secondary and args.workload == "k1000":
        ev2, table2, labels2, idx2 = prepare(SECONDARY, 1)
        e2, prof2, acc2 = run_steps(ev2, table2, labels2, idx2, 1, 3)
        roof2 = roofline_of(prof2, 3, SECONDARY["K"], SECONDARY["name"])
        line["secondary"] = {"workload": SECONDARY["text"], "value": 3 * 1000 / e2, "unit": "tasks/s", "steps": 3, "warmup": 1,
                             "ms_per_step": 1e3 * e2 / 3, "mean_accuracy": float(acc2),
                             "mm_iters_batch0": ev2.last_method.mm_iters[0].tolist(), "roofline": roof2}
        del ev2, table2, labels2, idx2
        torch.cuda.empty_cache()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(head, mm_iters[0].tolist())
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
