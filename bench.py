#!/usr/bin/env python3
"""Headline benchmark: transductive tasks/sec of the EM-Dirichlet hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one rank's workload: BASELINE.json configs[1]
(zero-shot EM-Dirichlet, K=100 classes, 75 queries, 1000 tasks as 10 reference batches of 100,
iter=20, iter_mm=1000, fp32) on synthetic peaked softmax features already resident in HBM,
from the engine call to the per-task accuracies on the host (accuracy tail included), followed
for N>1 by the single RCCL gather of the per-task predictions.  Weak scaling: every rank runs its
own 1000 tasks (independent batches, no data-path collective).

Extra objects on the JSON line:
  roofline     the dominant kernel k_mm_live (+ its dead-row twin k_mm_chunk), timed live with HIP events around each of its
               launches on the streams they run on (independent batches use a few internal
               streams, so launches overlap: `achieved` divides by the time during which at least
               one launch was running, `avg_launch_ms` is the plain mean launch duration).  The path is fp32 vector-ALU bound, so the bound is
               "valu": achieved = 48 flop-equivalents x element-updates executed / kernel time,
               peak = 157.3 TFLOP/s (fp32 vector, MI355X_MICROARCH.md); the compulsory HBM bytes
               of the same launches are reported beside it as hbm_* against 8 TB/s.
  cpu_baseline the torch-eager CPU restatement of the reference loop (oracle/ref_torch.py, same
               op sequence as the reference; kind "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd"))

import torch  # noqa: E402

K_CLASSES = 100
N_QUERY = 75
TASKS_PER_BATCH = 100
N_BATCHES = 10
ITERS = 20
ITER_MM = 1000
FLOP_EQ_PER_UPDATE = 48.0
PEAK_VALU_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
LANE_INSTR_PER_UPDATE = 327.0     # measured: SQ_INSTS_VALU 1.0347e11 x 64 lanes / 2.026e10 element-updates (profiles/r01_pmc_small_workload.txt)
PEAK_LANE_INSTR_T = 39.3          # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz, in 1e12 lane-instructions/s


_CPU_SNIPPET = r"""
import json, os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "transductive-clip_amd"))
import torch
from oracle import ref_torch
from tclip_amd import synth
torch.set_num_threads({threads})
x_q, _ = synth.make_query_tasks({n_tasks}, {K}, seed=0)
out = ref_torch.run(x_q, n_class={K}, iters={iters}, iter_mm={iter_mm}, lambd={lambd}, hard=False)
print(json.dumps({{"seconds": out["seconds"], "threads": torch.get_num_threads(), "torch": torch.__version__}}))
"""


def cpu_baseline(n_tasks=4, budget_s=240):
    """Reference loop on the host CPU (kind "port": oracle/ref_torch.py issues the reference's own
    torch op sequence), bounded sample: n_tasks tasks of the same workload in one batch, full
    20 x 1000 schedule.  Runs in a child process under a time budget so that an oversubscribed
    or throttled host cannot stall the benchmark."""
    import subprocess
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    threads = max(1, min(usable, 16))
    code = _CPU_SNIPPET.format(root=ROOT, threads=threads, n_tasks=n_tasks, K=K_CLASSES, iters=ITERS,
                               iter_mm=ITER_MM, lambd=int(K_CLASSES / 5) * N_QUERY)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s)
        info = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # timeout or failure: report it, never fake a number
        return {"value": None, "unit": "tasks/s", "cores": threads, "kind": "port",
                "sample": f"not measured: {type(e).__name__} within {budget_s}s budget"}
    secs = info["seconds"]
    return {"value": n_tasks / secs, "unit": "tasks/s", "cores": info["threads"], "kind": "port",
            "sample": f"{n_tasks} tasks (one batch) of the same K={K_CLASSES}, 75-query workload, full "
                      f"{ITERS}x{ITER_MM} schedule, torch {info['torch']} CPU eager, {secs:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = "RANK" in os.environ          # launched by torch.distributed.run (any world size)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the EM-Dirichlet engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from tclip_amd import engine, synth
    T = N_BATCHES * TASKS_PER_BATCH
    x_q, y_q = synth.make_query_tasks(T, K_CLASSES, seed=1000 + rank)
    x_q, y_q = x_q.to(dev), y_q.squeeze(2)
    lambd = int(K_CLASSES / 5) * N_QUERY

    def step():
        res = engine.run_em_dirichlet(x_q, n_batches=N_BATCHES, iters=ITERS, iter_mm=ITER_MM, lambd=lambd, hard=False)
        acc, _ = engine.clustering_accuracy(x_q, res.preds, y_q, graph_matching=True)
        if dist_on:
            blocks = [torch.empty_like(res.preds) for _ in range(world)]
            dist.all_gather(blocks, res.preds)     # the one exchange: per-task predictions
        return res, acc

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    engine.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, acc = step()
    fence()
    elapsed = time.perf_counter() - t0
    mm_ms, mm_launch_sum, mm_launches, updates = engine.profile_collect()
    engine.profile_enable(False)
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        steps = max(args.steps, 1)
        tasks = world * T * steps
        mm_iters = res.mm_iters.cpu().numpy()
        # algorithmic figures: element-updates under reference semantics vs executed
        ref_updates = float(K_CLASSES) ** 2 * TASKS_PER_BATCH * float(mm_iters.sum())
        achieved = FLOP_EQ_PER_UPDATE * updates / (mm_ms * 1e-3) / 1e12 if mm_ms > 0 else 0.0
        # compulsory HBM bytes of the MM launches: each processed row is read and written once per
        # launch (alpha + y in, alpha out) = 12 bytes per element per launch
        rows_bytes = 12.0 * updates / 50.0
        line = {
            "metric": "transductive tasks/sec (75-query EM-Dirichlet)",
            "value": tasks / elapsed, "unit": "tasks/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "EM-Dirichlet zero-shot, K=100 (caltech101-sized), 75-query, 1000 tasks "
                                   "per GPU as 10 batches of 100, iter=20, iter_mm=1000 (BASELINE.json configs[1])",
                       "n_class": K_CLASSES, "n_query": N_QUERY, "tasks_per_batch": TASKS_PER_BATCH,
                       "batches_per_gpu": N_BATCHES, "parallelism": f"batch-sharded x{world}",
                       "mean_accuracy": float(acc.mean()), "mm_iters_batch0": mm_iters[0].tolist()},
            "roofline": {"bound": "valu", "kernel": "k_mm_live", "achieved": achieved, "peak": PEAK_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_VALU_TFLOPS, "traffic": None,
                         "flop_eq_per_element_update": FLOP_EQ_PER_UPDATE,
                         "element_updates_executed_per_step": updates / steps,
                         "element_updates_reference_semantics_per_step": ref_updates,
                         "kernel_busy_ms_per_step": mm_ms / steps, "launches_per_step": mm_launches / steps,
                         "avg_launch_ms": mm_launch_sum / max(mm_launches, 1),
                         "launch_overlap": mm_launch_sum / mm_ms if mm_ms > 0 else 0.0,
                         # the same launches against the VALU ISSUE rate: lane-instructions per update is a
                         # PMC measurement (SQ_INSTS_VALU x 64 / updates, profiles/r01_pmc_small_workload.txt),
                         # the peak is one wave64 VALU instruction per SIMD every 4 cycles at 2.4 GHz
                         "valu_issue": {"achieved": LANE_INSTR_PER_UPDATE * updates / (mm_ms * 1e-3) / 1e12 if mm_ms > 0 else 0.0,
                                        "peak": PEAK_LANE_INSTR_T, "unit": "T lane-instr/s",
                                        "frac": (LANE_INSTR_PER_UPDATE * updates / (mm_ms * 1e-3) / 1e12) / PEAK_LANE_INSTR_T if mm_ms > 0 else 0.0,
                                        "lane_instructions_per_update": LANE_INSTR_PER_UPDATE},
                         # the same launches against the HBM roofline (north_star asks for it; the
                         # kernel keeps rows in registers for 50 iterations, so this is tiny by design)
                         "hbm": {"bound": "hbm", "achieved": rows_bytes / (mm_ms * 1e-3) / 1e9 if mm_ms > 0 else 0.0,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": (rows_bytes / (mm_ms * 1e-3) / 1e9) / PEAK_HBM_GBS if mm_ms > 0 else 0.0,
                                 "traffic": None, "algorithmic_bytes_per_step": rows_bytes / steps}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
