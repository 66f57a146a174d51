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

stdout carries ONE compact JSON line (the contract's fields, `roofline`, `cpu_baseline`, one figure per secondary
workload; a few kB - round 5's 20 kB line was more than the driver's reader takes); the full record described below
goes to gpurun_out/bench_full.json and to stderr (`full_record` on the line names the file).

Objects of the full record (the line carries their essential fields):
  roofline     the dominant kernels - k_mm_live (first outer iteration) and k_mm_split (the later ones), the MM loop of
               the live rows (`per_kernel`: each of the two alone) - timed live with HIP events around each of their launches on the streams they run on
               (independent batches use a few internal streams, so launches overlap: `achieved` divides by the time
               during which at least one launch was running, `avg_launch_ms` is the plain mean launch duration).
               The path is fp32 vector-ALU bound, so the bound is "valu": achieved = 48
               flop-equivalents x element-updates executed / kernel time against 157.3 TFLOP/s (the 2.4 GHz sheet
               value); `measured_clock_ghz` / `frac_at_measured_clock` rescale the peak to the clock the chip held in
               the committed PMC pass (GRBM_GUI_ACTIVE / 8 / kernel time).  `traffic` = HBM-side bytes per launch from
               the PMC passes recorded in profiles/pmc_current.json (FETCH_SIZE + WRITE_SIZE, calibrated on the
               dword-per-lane copy kernel of the same run) next to `traffic_algorithmic_bytes_per_launch` of THAT
               workload; null when the file does not describe this build.
  secondary    the other BASELINE.json configs on one GPU, a few steps each, every one with its own roofline and an
               extrapolated cpu_baseline: `k100` = configs[1] (K=100), `k397_hard` = configs[2] (Hard EM-Dirichlet at
               K=397 plus SOFT_KMEANS on the same tasks), `fs_k1000` = configs[4] (visual embeddings -> probability
               front-end -> 4-shot few-shot EM-Dirichlet at K=1000).
  cpu_baseline the torch-eager CPU restatement of the reference loop (oracle/ref_torch.py, kind "port") on this host:
               K=1000 is ~6 minutes per task on 8 cores, so, as SURVEY.md 8d prescribes, MM iterations of the first and
               of a later outer iteration plus one M/E-step are timed on a 2-task batch and extrapolated over the MM
               schedule the GPU run recorded.
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
sys.path.insert(0, os.path.join(ROOT, "transductive-clip_amd", "drop_in"))

import torch  # noqa: E402
from tclip_amd import _capi  # noqa: E402

N_QUERY = 75
ITERS = 20
ITER_MM = 1000
HEADLINE = dict(name="k1000", K=1000, tasks_per_batch=125, batches_per_gpu=10, rows_per_class=50,
                # the driver's record keeps the first 120 characters of this string: the schedule comes first
                text="EM-Dirichlet zero-shot K=1000, 75-query, iter=20 x iter_mm=1000, fp32, 1250 tasks/GPU as 10 batches of 125 "
                     "(BASELINE.json configs[3], imagenet-sized: 10 000 tasks on 8 GPUs = this per GPU)")
SECONDARY = dict(name="k100", K=100, tasks_per_batch=100, batches_per_gpu=10, rows_per_class=40,
                 text="EM-Dirichlet zero-shot, K=100 (caltech101-sized), 75-query, 1000 tasks per GPU as 10 batches of 100, "
                      "iter=20, iter_mm=1000 (BASELINE.json configs[1])")
K397_HARD = dict(name="k397_hard", K=397, tasks_per_batch=100, batches_per_gpu=10, rows_per_class=40, method="HARD_EM_DIRICHLET",
                 iters=10,
                 text="Hard EM-Dirichlet zero-shot, K=397 (sun397-sized), 75-query, 1000 tasks as 10 batches of 100, iter=10, "
                      "iter_mm=1000, plus SOFT_KMEANS (iter=20, T=30) on the same tasks (BASELINE.json configs[2])")
FS_K1000 = dict(name="fs_k1000", K=1000, tasks_per_batch=25, batches_per_gpu=4, shots=4, dim=512, support_rows_per_class=5,
                query_rows_per_class=20, signal=6.0,      # embedding = 6 x class direction + unit noise: cos 0.26 to the own class text,
                                                          # probability ~0.5 on it at T = 30 (CLIP-like peakedness)
                text="visual embeddings (512-d) -> probability features softmax(30 cos) on the device -> 4-shot few-shot "
                     "EM-Dirichlet, K=1000, S=4000 support rows per task, 75-query, 100 tasks as 4 batches of 25, iter=20, "
                     "iter_mm=1000 (BASELINE.json configs[4]; synthetic unit-norm text embeddings, SURVEY.md config-5 note)")
FLOP_EQ_PER_UPDATE = 48.0
PEAK_VALU_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_current.json")


_CPU_SNIPPET = r"""
import json, os, resource, sys, time
resource.setrlimit(resource.RLIMIT_AS, ({mem_gb} << 30, {mem_gb} << 30))      # fail with MemoryError rather than take the host down
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "transductive-clip_amd"))
import torch
from oracle import ref_torch
from tclip_amd import synth
torch.set_num_threads({threads})
x_q, _ = synth.make_query_tasks({n_tasks}, {K}, seed=0, k_eff={k_eff})
x_s = y_s = None
if {shots}:
    x_s, y_s = synth.make_support({n_tasks}, {K}, {shots}, seed=0)
    y_s = y_s.squeeze(2)
out = {{}}
if {warm}:
    ref_torch.run(x_q, x_s, y_s, n_class={K}, iters=1, iter_mm=52, lambd={lambd}, hard={hard})      # thread pool, allocator, page faults
r = ref_torch.run(x_q, x_s, y_s, n_class={K}, iters={iters}, iter_mm={iter_mm}, lambd={lambd}, hard={hard})
out = {{"mm": r["seconds_mm"], "iter": r["seconds_iter"], "ran": [int(v) for v in r["mm_iters"]]}}
print(json.dumps({{"seconds": out, "threads": torch.get_num_threads(), "torch": torch.__version__}}))
"""


def cpu_baseline(w, mm_schedule, budget_s=300, iters_total=ITERS):
    """Reference loop on the host CPU (kind "port": oracle/ref_torch.py issues the reference's own
    torch op sequence).  mm_schedule: MM iterations per outer iteration the GPU run recorded for
    batch 0.  Child process under a time budget and an address-space limit: an oversubscribed or small host
    cannot stall the bench.  Always a bounded sample: MM iterations of the first and of a later outer iteration
    and one M/E-step are timed on a 2-task batch and extrapolated over the recorded schedule.
    Few-shot at K=1000: the reference's (N,S,K,K) support temporary is 16 GB per task at 4 shots, so the sample
    runs ONE task at ONE shot (4 GB) and the support part of the M-step, linear in S, is scaled to 4 shots."""
    import subprocess
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    threads = max(1, min(usable, 16))
    K, shots, hard = w["K"], w.get("shots", 0), w.get("method", "") == "HARD_EM_DIRICHLET"
    # ONE run of 2 outer iterations x 151 MM iterations (501 for the headline); oracle/ref_torch.py reports the host time of each MM loop and of
    # each whole outer iteration, so nothing is a difference of separately timed runs
    n_tasks, sample_shots, iters, iter_mm = 2, shots, 2, 151
    if K >= 1000 and not shots:
        iter_mm = 501                                               # the headline's sample: ~10 s of host time on 16 threads
    if K < 397:                                                     # the reference batch itself fits ((N,Q,K,K) = 300 MB at K = 100)
        n_tasks = w["tasks_per_batch"]
    elif K < 1000:
        n_tasks = 8                                                 # (N,Q,K,K) = 380 MB at K = 397; the 100-task batch would need 4.7 GB per temporary
    if shots and K >= 397:
        n_tasks, sample_shots = 1, 1
    lambd = int(K / 5) * N_QUERY

    def child(shots_, k_eff):
        code = _CPU_SNIPPET.format(root=ROOT, threads=threads, n_tasks=n_tasks, K=K, iters=iters, iter_mm=iter_mm, lambd=lambd,
                                   warm=True, shots=shots_, hard=hard, k_eff=k_eff, mem_gb=40)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s)
        return json.loads(out.stdout.strip().splitlines()[-1])
    host = host_description(usable)
    per_me_zs = None
    if sample_shots != shots:                                       # support statistics are linear in S = K * shots: their share of the
        try:                                                        # M/E-step is what a zero-shot run of the same sample does not spend
            zs = child(0, 5)["seconds"]
            per_me_zs = 0.5 * ((zs["iter"][0] - zs["mm"][0]) + (zs["iter"][1] - zs["mm"][1]))
        except Exception:
            per_me_zs = 0.0
    # TWO samples of the same workload, one after the other: the figure is their mean and the line carries both, so that a
    # host whose other tenants came and went during the run shows up as spread instead of as a different number per box
    values, texts, info = [], [], None
    for rep in range(2):
        try:
            info = child(sample_shots, 5 if shots else None)
        except Exception as e:  # timeout or failure: report it, never fake a number
            if values:
                texts.append(f"second sample not measured: {type(e).__name__} within {budget_s}s budget")
                break
            return {"value": None, "unit": "tasks/s", "cores": threads, "kind": "port", **host,
                    "sample": f"not measured: {type(e).__name__} within {budget_s}s budget"}
        sec = info["seconds"]
        ran, t_mm, t_iter = sec["ran"], sec["mm"], sec["iter"]
        mm_first = t_mm[0] / max(ran[0], 1)                         # one MM iteration, first outer iteration (every class alive, alpha near 1)
        mm_later = t_mm[1] / max(ran[1], 1)                         # one MM iteration, second outer iteration
        per_me = 0.5 * ((t_iter[0] - t_mm[0]) + (t_iter[1] - t_mm[1]))  # M-step statistics + E-step + criterion, once per outer iteration
        note = ""
        if per_me_zs is not None:
            support_part = max(per_me - per_me_zs, 0.0)
            note = (f"; sampled at {sample_shots} shot (M/E-step {per_me:.2f}s, of which support statistics {support_part:.2f}s, "
                    f"linear in S) and scaled to {shots} shots")
            per_me = per_me_zs + support_part * shots / sample_shots
        total = len(mm_schedule) * per_me + mm_schedule[0] * mm_first + sum(mm_schedule[1:]) * mm_later
        values.append(n_tasks / total)
        texts.append(f"{sum(t_iter):.1f}s host time, MM loops {t_mm[0]:.2f}s / {t_mm[1]:.2f}s -> {1e3 * mm_first:.2f} / {1e3 * mm_later:.2f} ms per MM "
                     f"iteration in the first / a later outer iteration, {per_me:.2f}s per M/E-step{note} -> {total:.0f}s per {n_tasks} tasks")
    mean = sum(values) / len(values)
    spread = (max(values) - min(values)) / mean if len(values) > 1 else None
    return {"value": mean, "unit": "tasks/s", "cores": info["threads"], "kind": "port",
            "samples": values, "rel_spread": spread, **host,
            "sample_short": f"{n_tasks}-task batch of the same K={K} workload, {len(values)} runs of 2 outer iterations x {ran} MM iterations, per-iteration "
                            f"times extrapolated over the recorded schedule ({mm_schedule[0]} + {int(sum(mm_schedule[1:]))} MM iterations, SURVEY.md 8d); "
                            f"{info['threads']} threads, torch {info['torch']} CPU eager",
            "sample": f"{n_tasks}-task batch of the same K={K}, 75-query workload, {len(values)} runs of 2 outer iterations with {ran} MM iterations each, "
                      f"extrapolated over the recorded schedule ({mm_schedule[0]} + {int(sum(mm_schedule[1:]))} MM iterations in "
                      f"{len(mm_schedule)} outer iterations, SURVEY.md 8d): " + " | ".join(f"run {i + 1}: {t}" for i, t in enumerate(texts)) +
                      f"; mean {mean:.4g} tasks/s" + (f", relative spread {spread:.1%}" if spread is not None else "") +
                      f"; {info['threads']} threads on {host['cpu_model']} ({host['cpus_usable']} of {host['cpus_online']} CPUs usable, "
                      f"load average {host['loadavg_1min']:.1f} before the sample); torch {info['torch']} CPU eager",
            "extrapolated": True}


def host_description(usable):
    """What the cpu_baseline figure was measured on: the model string of the host's CPUs, how many of them this process may use
    (sched_getaffinity) and how busy the host was with other work when the sample started."""
    model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        load = os.getloadavg()[0]
    except OSError:
        load = float("nan")
    return {"cpu_model": model, "cpus_online": os.cpu_count() or 1, "cpus_usable": usable, "loadavg_1min": load}


def load_pmc():
    try:
        with open(PMC_FILE) as f:
            return json.load(f)
    except Exception:
        return None


LINE_BUDGET = 6000          # bytes: the driver keeps a bounded tail of stdout; round 5's 20 kB line came back unparsed


def _round(o, digits=8):
    """floats to `digits` significant digits, recursively (the line is a record, not a checkpoint)"""
    if isinstance(o, float):
        return float(f"{o:.{digits}g}") if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _round(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round(v, digits) for v in o]
    return o


def _rle(seq):
    """[501, 1000, 1000, ...] -> "501,1000x19" (the MM schedule of a batch)"""
    out, i = [], 0
    while i < len(seq):
        j = i
        while j < len(seq) and seq[j] == seq[i]:
            j += 1
        out.append(f"{seq[i]}x{j - i}" if j - i > 1 else f"{seq[i]}")
        i = j
    return ",".join(out)


def _short_roofline(r):
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_algorithmic_bytes_per_launch",
            "traffic_over_algorithmic", "avg_launch_ms", "launches_per_step", "kernel_busy_ms_per_step",
            "element_updates_per_launch", "algorithmic_bytes_per_launch", "flop_eq_per_element_update", "measured_clock_ghz",
            "frac_at_measured_clock", "pmc_stale")
    out = {k: r[k] for k in keep if k in r}
    if "per_kernel" in r:
        out["per_kernel"] = {n: {k: v[k] for k in ("frac", "avg_launch_ms", "launches_per_step")} for n, v in r["per_kernel"].items()}
    if "hbm" in r:
        out["hbm"] = {k: r["hbm"][k] for k in ("achieved", "peak", "unit", "frac")}
    src = r.get("traffic_source")
    if src:
        out["pmc"] = {k: src.get(k) for k in ("file", "lane_instr_per_update", "valu_busy_frac", "scratch_bytes_per_lane")}
    return out


def _short_cpu(c):
    out = {k: c[k] for k in ("value", "unit", "cores", "kind", "samples", "rel_spread", "cpu_model", "cpus_usable", "loadavg_1min",
                             "extrapolated") if k in c}
    out["sample"] = c.get("sample_short") or c.get("sample", "")[:300]
    return out


def short_line(full, record_path):
    """The ONE stdout line: the contract's fields, `roofline` and `cpu_baseline` in full meaning but without the
    derivations' intermediate figures, one figure per secondary workload.  Everything else is in the full record
    (`record_path`, also printed on stderr)."""
    line = {k: v for k, v in full.items() if k not in ("roofline", "cpu_baseline", "secondary", "config")}
    cfg = dict(full["config"])
    if isinstance(cfg.get("mm_iters_batch0"), list):
        cfg["mm_iters_batch0"] = _rle(cfg["mm_iters_batch0"])
    line["config"] = cfg
    line["roofline"] = _short_roofline(full["roofline"])
    if "cpu_baseline" in full:
        line["cpu_baseline"] = _short_cpu(full["cpu_baseline"])
    if "secondary" in full:
        sec = {}
        for name, o in full["secondary"].items():
            if not (isinstance(o, dict) and "roofline" in o):   # records up to round 5 repeated k100's fields beside the three objects
                continue
            sec[name] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"],
                         "roofline_frac": o["roofline"]["frac"]}
            if o.get("cpu_baseline"):
                sec[name]["cpu_baseline"] = o["cpu_baseline"].get("value")
            if "soft_kmeans" in o:
                sec[name]["soft_kmeans"] = o["soft_kmeans"]["value"]
        line["secondary"] = sec
    line["full_record"] = record_path
    line = _round(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_BUDGET:                          # never again a line the driver cannot take: shed the optional parts
        for k in ("secondary", "rank_devices", "rank_step_ms"):
            line.pop(k, None)
        line["roofline"].pop("per_kernel", None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def write_record(full):
    """Full record: gpurun_out/bench_full.json (merged back by gpurun; the committed copies live under profiles/) and stderr."""
    rel = os.path.join("gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(full, f)
            f.write("\n")
    except OSError:
        rel = "stderr"
    print("bench full record: " + json.dumps(full), file=sys.stderr, flush=True)
    return rel


def roofline_of(prof, steps, K, pmc_key):
    """prof = engine.profile_collect() of `steps` timed steps (+ engine.profile_last_kernels() as a fifth entry)."""
    mm_ms, mm_launch_sum, mm_launches, updates = prof[:4]
    kernels = prof[4] if len(prof) > 4 else {}
    achieved = FLOP_EQ_PER_UPDATE * updates / (mm_ms * 1e-3) / 1e12 if mm_ms > 0 else 0.0
    # algorithmic HBM bytes of the MM launches: every listed row is read (alpha, y) and written (alpha)
    # once per launch = 12 bytes per element per <=51-iteration launch
    rows_bytes = 12.0 * updates / 50.0
    hbm_gbs = rows_bytes / (mm_ms * 1e-3) / 1e9 if mm_ms > 0 else 0.0
    out = {"bound": "valu", "kernel": "k_mm_live + k_mm_split (MM loop of the live rows)", "achieved": achieved, "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
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
    # the two kernels of the combined figure, each against the same peak: busy = union of ITS launches' intervals (launches of
    # different stream groups overlap, so the two busy times add up to more than the combined one)
    out["per_kernel"] = {}
    for name, (busy, lsum, n, upd) in kernels.items():
        if n:
            tf = FLOP_EQ_PER_UPDATE * upd / (busy * 1e-3) / 1e12 if busy > 0 else 0.0
            out["per_kernel"][name] = {"achieved": tf, "frac": tf / PEAK_VALU_TFLOPS, "kernel_busy_ms_per_step": busy / steps,
                                       "launches_per_step": n / steps, "avg_launch_ms": lsum / n,
                                       "element_updates_per_step": upd / steps,
                                       "element_updates_per_s": upd / (busy * 1e-3) if busy > 0 else 0.0}
    pmc = load_pmc()
    if pmc and pmc_key in pmc and pmc[pmc_key].get("csrc_sha1") != _capi.source_digest():
        # the committed counters were taken on other kernel sources: traffic, clock and the fractions derived from them
        # would describe a build that no longer exists
        out["pmc_stale"] = {"file": pmc[pmc_key].get("file"), "commit": pmc[pmc_key].get("commit"),
                            "note": "csrc changed since this PMC pass: traffic / measured_clock_ghz not reported"}
        pmc = None
    if pmc and pmc_key in pmc:
        e = pmc[pmc_key]
        # traffic and the algorithmic bytes it is to be compared with come from the SAME (smaller) PMC workload
        out["traffic"] = e.get("traffic_bytes_per_launch")
        out["traffic_algorithmic_bytes_per_launch"] = e.get("algorithmic_bytes_per_launch")
        if out["traffic"] and e.get("algorithmic_bytes_per_launch"):
            out["traffic_over_algorithmic"] = out["traffic"] / e["algorithmic_bytes_per_launch"]
        if e.get("clock_ghz"):                       # the clock the chip held under this kernel: GRBM_GUI_ACTIVE / 8 / kernel time
            out["measured_clock_ghz"] = e["clock_ghz"]
            out["peak_at_measured_clock"] = PEAK_VALU_TFLOPS * e["clock_ghz"] / 2.4
            out["frac_at_measured_clock"] = achieved / out["peak_at_measured_clock"]
        out["traffic_source"] = {k: e.get(k) for k in ("file", "workload", "fetch_bytes_per_launch", "write_bytes_per_launch",
                                                       "algorithmic_bytes_per_launch", "calibration", "lane_instr_per_update",
                                                       "valu_busy_frac", "wait_frac", "scratch_bytes_per_lane", "clock_ghz", "commit")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--workload", choices=["k1000", "k100", "k397_hard", "fs_k1000"], default="k1000",
                    help="k1000: the headline (default); k100: the round-1 shape as the headline; k397_hard / fs_k1000: only that "
                         "secondary workload (BASELINE configs[2] / configs[4]), for profiling")
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
        import contextlib
        import torch.distributed as dist

        @contextlib.contextmanager
        def stdout_to_stderr():
            """librccl prints a version banner on the process's stdout (fd 1) when its first communicator comes up; the
            contract of this script is ONE JSON line on stdout, so fd 1 points at stderr while RCCL initialises"""
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            try:
                yield
            finally:
                sys.stdout.flush()
                os.dup2(keep, 1)
                os.close(keep)

        with stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(args.backend)
            dist.barrier()                               # the communicator (and its banner) comes up here at the latest
            torch.cuda.synchronize()
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the launcher started {dist.get_world_size()} ranks")

    import random

    import numpy as np
    if dist_on:
        # N ranks prepare the same synthetic table at the same time (200 MB of randn + softmax at K = 1000): each keeps to its
        # share of the host's cores instead of N x all of them
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        torch.set_num_threads(max(1, usable // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
    from src.eval_zero_shot import Evaluator_zero_shot
    from src.utils import CfgNode
    from tclip_amd import engine, sharding, synth

    rank_devices = None
    if dist_on and not args.single_device:               # one GPU per rank, or the per-N values would not be a scaling curve
        rank_devices = sharding.check_one_device_per_rank(torch.cuda.current_device(), dev if args.backend == "nccl" else None)

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def prepare(w, n_ranks, method="EM_DIRICHLET", iters=ITERS):
        """Feature table in HBM + the index stream of every batch (same on every rank)."""
        K = w["K"]
        feats, labels = synth.make_feature_table(K, w["rows_per_class"], seed=2020)
        n_tasks = w["tasks_per_batch"] * w["batches_per_gpu"] * (8 if args.scaling == "strong" else n_ranks)
        cfg = CfgNode(iter=iters, iter_mm=ITER_MM, num_classes_test=K, n_class=K, n_query=N_QUERY, k_eff=5, T=30,
                      use_softmax_feature=True, graph_matching=True, shots=0, number_tasks=n_tasks,
                      batch_size=w["tasks_per_batch"], name_method=method, used_test_set="test")
        ev = Evaluator_zero_shot(device=dev, args=cfg, log_file=None)
        random.seed(2020); np.random.seed(2020); torch.manual_seed(2020)     # main.py:42-46
        idx = ev.sample_indices(labels.numpy())
        return ev, feats.to(dev), labels.to(dev), idx

    def timed(step, warmup, steps):
        """`warmup` untimed and `steps` timed calls of step(); returns (seconds, profile, last result, per-rank seconds)."""
        for _ in range(warmup):
            step()
        fence()
        engine.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        fence()
        elapsed = time.perf_counter() - t0
        prof = engine.profile_collect() + (engine.profile_last_kernels(),)
        engine.profile_enable(False)
        per_rank = sharding.gather_rank_values(elapsed, dev if args.backend == "nccl" else None) if dist_on else [elapsed]
        elapsed = max(per_rank)                          # the job is done when its slowest rank is
        return elapsed, prof, res, per_rank

    def run_steps(ev, table, labels, idx, warmup, steps):
        e, prof, res, per_rank = timed(lambda: ev.evaluate_tasks(None, table, labels, indices=idx), warmup, steps)
        return e, prof, res[0], per_rank

    def zero_shot_secondary(w, method="EM_DIRICHLET", iters=ITERS, warmup=1, n_steps=3):
        ev2, table2, labels2, idx2 = prepare(w, 1, method, iters)
        e2, prof2, acc2, _ = run_steps(ev2, table2, labels2, idx2, warmup, n_steps)
        n_tasks = w["tasks_per_batch"] * w["batches_per_gpu"]
        sched = ev2.last_method.mm_iters[0].tolist()
        obj = {"workload": w["text"], "value": n_steps * n_tasks / e2, "unit": "tasks/s", "steps": n_steps, "warmup": warmup,
               "ms_per_step": 1e3 * e2 / n_steps, "mean_accuracy": float(acc2), "mm_iters_batch0": sched,
               "roofline": roofline_of(prof2, n_steps, w["K"], w["name"])}
        return obj, (ev2, table2, labels2, idx2), sched

    def run_k100():                                      # configs[1]: K = 100
        obj, keep, sched = zero_shot_secondary(SECONDARY)
        if not args.no_cpu_baseline:
            obj["cpu_baseline"] = cpu_baseline(SECONDARY, sched, budget_s=60)
        del keep
        torch.cuda.empty_cache()
        return obj

    def run_k397_hard():                                 # configs[2]: Hard EM-Dirichlet at K = 397, then SOFT_KMEANS on the same tasks
        obj, (ev2, table2, labels2, idx2), sched = zero_shot_secondary(K397_HARD, "HARD_EM_DIRICHLET", K397_HARD["iters"], 1, 2)
        ev3 = Evaluator_zero_shot(device=dev, args=CfgNode(dict(ev2.args, name_method="SOFT_KMEANS", iter=20)), log_file=None)
        e3, _, acc3, _ = run_steps(ev3, table2, labels2, idx2, 1, 2)
        T, K3 = K397_HARD["tasks_per_batch"] * K397_HARD["batches_per_gpu"], K397_HARD["K"]
        # algorithmic HBM bytes of one SOFT_KMEANS iteration: statistics (u, z in; w out), distances (w, z in; logits out),
        # softmax (logits in; u out) = 4 B x (6 T Q K + 2 T K K)
        skm_bytes = 20 * 4.0 * (6.0 * T * N_QUERY * K3 + 2.0 * T * K3 * K3)
        skm_flop = 20 * 5.0 * T * N_QUERY * K3 * K3                  # distances 3 flop per (q, k, d), statistics 2
        obj["soft_kmeans"] = {"value": 2 * T / e3, "unit": "tasks/s", "steps": 2, "warmup": 1, "ms_per_step": 1e3 * e3 / 2,
                              "mean_accuracy": float(acc3),
                              "roofline": {"bound": "hbm", "kernel": "whole SOFT_KMEANS step (k_mstats_rows, k_kmeans_logits_rows, k_softmax)",
                                           "achieved": skm_bytes / (e3 / 2) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                           "frac": skm_bytes / (e3 / 2) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                                           "algorithmic_bytes_per_step": skm_bytes,
                                           "fp32_tflops_of_the_two_contractions": skm_flop / (e3 / 2) / 1e12,
                                           "note": "sums rebuilt in torch's association order (no MFMA): L2-resident re-reads of the "
                                                   "task's feature block, not HBM, bound this step"}}
        if not args.no_cpu_baseline:
            obj["cpu_baseline"] = cpu_baseline(K397_HARD, sched, budget_s=60)
        del ev2, ev3, table2, labels2, idx2
        torch.cuda.empty_cache()
        return obj

    def run_fs_k1000():                                  # configs[4]: visual embeddings -> probability features -> 4-shot few-shot EM-Dirichlet at K = 1000
        from src.eval_few_shot import Evaluator_few_shot
        from tclip_amd import features
        w = FS_K1000
        K4, D = w["K"], w["dim"]
        gen = torch.Generator().manual_seed(2024)
        text = torch.randn(K4, D, generator=gen)
        text /= text.norm(dim=-1, keepdim=True)
        lab_s = torch.arange(K4).repeat_interleave(w["support_rows_per_class"])
        lab_q = torch.arange(K4).repeat_interleave(w["query_rows_per_class"])
        vis_s = (text[lab_s] * FS_K1000["signal"] + torch.randn(len(lab_s), D, generator=gen)).to(dev)
        vis_q = (text[lab_q] * FS_K1000["signal"] + torch.randn(len(lab_q), D, generator=gen)).to(dev)
        text_d = text.to(dev)
        n_tasks = w["tasks_per_batch"] * w["batches_per_gpu"]
        cfg = CfgNode(iter=ITERS, iter_mm=ITER_MM, num_classes_test=K4, n_class=K4, n_query=N_QUERY, k_eff=5, T=30,
                      use_softmax_feature=True, graph_matching=True, shots=w["shots"], number_tasks=n_tasks,
                      batch_size=w["tasks_per_batch"], name_method="EM_DIRICHLET", used_test_set="test", tunable=False)
        ev4 = Evaluator_few_shot(device=dev, args=cfg, log_file=None)
        random.seed(2020); np.random.seed(2020); torch.manual_seed(2020)
        idx4 = ev4.sample_indices(lab_s.numpy(), lab_q.numpy())

        def fs_step():                                   # the front-end is part of the step
            tab_s = features.probability_features(vis_s, text_d, 30.0)
            tab_q = features.probability_features(vis_q, text_d, 30.0)
            return ev4.evaluate_tasks(None, tab_s, lab_s, tab_q, lab_q, indices=idx4)

        e4, prof4, res4, _ = timed(fs_step, 1, 2)
        sched = ev4.last_method.mm_iters[0].tolist()
        obj = {"workload": w["text"], "value": 2 * n_tasks / e4, "unit": "tasks/s", "steps": 2, "warmup": 1, "ms_per_step": 1e3 * e4 / 2,
               "mean_accuracy": float(res4[0]), "mm_iters_batch0": sched, "roofline": roofline_of(prof4, 2, K4, w["name"]),
               "support_rows_per_task": K4 * w["shots"],
               "note": "few-shot has no dead rows: all K rows of every task iterate in every outer iteration "
                       "(roofline.element_updates_executed_per_step counts them)"}
        if not args.no_cpu_baseline:
            obj["cpu_baseline"] = cpu_baseline(w, sched, budget_s=90)
        del ev4, vis_s, vis_q
        torch.cuda.empty_cache()
        return obj

    runners = {"k100": run_k100, "k397_hard": run_k397_hard, "fs_k1000": run_fs_k1000}

    if args.workload in ("k397_hard", "fs_k1000"):       # one secondary alone (profiling); single GPU
        if world != 1:
            raise SystemExit("--workload k397_hard / fs_k1000 runs on one GPU")
        obj = runners[args.workload]()
        print(json.dumps(dict({"metric": "transductive tasks/sec (75-query EM-Dirichlet)", "n_gpus": 1, "higher_is_better": True,
                               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                               "config": {"workload": obj["workload"]}}, **obj)), flush=True)
        return

    head = HEADLINE if args.workload == "k1000" else SECONDARY
    ev, table, labels, idx = prepare(head, world)
    elapsed, prof, acc_mean, per_rank = run_steps(ev, table, labels, idx, args.warmup, args.steps)
    steps = max(args.steps, 1)
    mm_iters = ev.last_method.mm_iters                  # (local batches, iters)
    line = None
    if rank == 0:
        job = (8 if args.scaling == "strong" else world) * head["tasks_per_batch"] * head["batches_per_gpu"]
        tasks = job * steps
        K = head["K"]
        roof = roofline_of(prof, steps, K, head["name"])
        roof["element_updates_reference_semantics_per_step"] = float(K) ** 2 * head["tasks_per_batch"] * float(mm_iters.sum())
        preds = ev.last_task_predictions                 # what the one collective delivered: (batches, tasks per batch, 75) int32
        line = {
            "metric": "transductive tasks/sec (75-query EM-Dirichlet)",
            "value": tasks / elapsed, "unit": "tasks/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": head["text"], "n_class": K, "n_query": N_QUERY, "tasks_per_batch": head["tasks_per_batch"],
                       "batches_per_gpu": head["batches_per_gpu"] * (8 // world if args.scaling == "strong" else 1), "tasks_total": job,
                       "parallelism": f"batch-sharded x{world}, one all_gather of per-task predictions (int32 x 75), accuracies, "
                                      f"per-batch criterions and MM counts",
                       "path": "Evaluator_zero_shot.evaluate_tasks (device gather from a 50-rows-per-class synthetic table, "
                               "engine, accuracy tail, gather)",
                       "mean_accuracy": float(acc_mean), "mm_iters_batch0": mm_iters[0].tolist(),
                       "gathered": {"predictions": list(preds.shape), "criterions": list(ev.last_batch_criterions.shape),
                                    "mm_iters": list(ev.last_batch_mm_iters.shape)}},
            "roofline": roof,
        }
        try:                                             # which unit of the pool this was: the MM launches' durations differ by up to 5 % between boxes
            props = torch.cuda.get_device_properties(dev)
            line["device"] = {"name": props.name, "compute_units": props.multi_processor_count, "hbm_gib": round(props.total_memory / 2 ** 30)}
        except Exception:
            pass
        if dist_on:
            line["ranks_seen"] = dist.get_world_size()
            line["rank_devices"] = rank_devices
            line["backend"] = dist.get_backend()
            line["rank_step_ms"] = {"min": 1e3 * min(per_rank) / steps, "max": 1e3 * max(per_rank) / steps,
                                    "all": [1e3 * t / steps for t in per_rank]}
    del ev, table, labels, idx
    torch.cuda.empty_cache()

    if world == 1 and not args.no_secondary and args.workload == "k1000":
        sec = {name: fn() for name, fn in runners.items()}
        line["secondary"] = sec

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(head, mm_iters[0].tolist(), budget_s=120)
        print(short_line(line, write_record(line)), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
