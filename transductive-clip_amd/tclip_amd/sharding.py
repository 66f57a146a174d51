"""Whole-batch sharding of the task-batch loop across the GPUs of one node.

Reference batches are independent (new method instance per batch, eval_zero_shot.py:151-172)
while tasks inside a batch are coupled by the MM stop test, so the batch is the unit that is
distributed: rank r runs batches {b : b % world == r}.  The data path needs no collective; the
only exchange is ONE gather of the per-task results onto rank 0 (RCCL over xGMI when the
process group is "nccl", gloo in the CPU tests)."""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def my_batches(n_batches, rank=None, world_size=None):
    if rank is None:
        rank, world_size = world()
    return list(range(rank, n_batches, world_size))


# methods whose engine call keeps per-batch records besides predictions and accuracies
_PER_BATCH_RECORDS = {'EM_DIRICHLET': ('criterions', 'mm_iters'), 'HARD_EM_DIRICHLET': ('criterions', 'mm_iters'),
                      'ALPHA_TIM': ('criterions',)}


def method_parts(args, method, logs, n_local, N, Q, dev):
    """This rank's results of one method run over `n_local` batches of N tasks, as the parts gather_packed moves
    (SURVEY.md section 8e): `preds` (N*Q) int32 - the class assigned to every query, after the cluster-to-class
    matching for the zero-shot clustering methods, `acc` (N) f32, and for the EM-Dirichlet classes / ALPHA_TIM the
    per-batch `criterions` (iter) f32 and `mm_iters` (iter) int32.  method=None: a rank without a batch (zero rows of
    the same widths)."""
    records = _PER_BATCH_RECORDS.get(getattr(args, 'name_method', None), ())
    iters = int(getattr(args, 'iter', 0))
    if method is None:
        parts = {'preds': torch.zeros(0, N * Q, dtype=torch.int32), 'acc': torch.zeros(0, N)}
        if 'criterions' in records:
            parts['criterions'] = torch.zeros(0, iters)
        if 'mm_iters' in records:
            parts['mm_iters'] = torch.zeros(0, iters, dtype=torch.int32)
    else:
        p = getattr(method, 'matched_preds', None)
        if p is None:
            p = method.preds
        parts = {'preds': p.reshape(n_local, N * Q).to(torch.int32),
                 'acc': torch.from_numpy(logs['acc'][:, -1].copy()).view(n_local, N).float()}
        if 'criterions' in records:
            parts['criterions'] = torch.as_tensor(method.criterions_per_batch, dtype=torch.float32).view(n_local, iters)
        if 'mm_iters' in records:
            parts['mm_iters'] = torch.as_tensor(method.mm_iters, dtype=torch.int32).view(n_local, iters)
    return {k: v.to(dev) for k, v in parts.items()}


def concat_parts(a, b):
    """rows of `a` followed by rows of `b` (two method runs of one rank, in batch order)"""
    return b if a is None else {k: torch.cat([a[k], b[k]], 0) for k in a}


def gather_packed(parts, n_batches, rank=None, world_size=None):
    """The one collective of a step: every per-batch result of this rank in ONE all_gather.

    parts: dict name -> tensor (n_local_batches, ...) of dtype int32 or float32, rows in my_batches(n_batches) order
    (per-task predictions (N*Q) int32, accuracies (N) f32, criterions (iters) f32, MM counts (iters) int32 ...).
    The rows are bit-cast to int32 and laid side by side in one (batches_per_rank, width) block per rank.
    Returns on rank 0 a dict of (n_batches, ...) CPU tensors in batch order, None elsewhere."""
    if rank is None:
        rank, world_size = world()
    names = sorted(parts)
    for n in names:
        if parts[n].dtype not in (torch.int32, torch.float32):
            raise TypeError(f"gather_packed moves int32 / float32 rows, {n} is {parts[n].dtype}")
    shapes = {n: tuple(parts[n].shape[1:]) for n in names}
    widths = {n: int(torch.Size(shapes[n]).numel()) for n in names}
    n_local = parts[names[0]].shape[0]
    dev = parts[names[0]].device
    flat = [parts[n].to(dev).reshape(n_local, widths[n]).contiguous().view(torch.int32) for n in names]
    local = torch.cat(flat, 1)
    out = gather_batch_results(local, n_batches, rank, world_size)
    if out is None:
        return None
    out = out.cpu()
    res, o = {}, 0
    for n in names:
        block = out[:, o:o + widths[n]].contiguous()
        res[n] = block.view(parts[n].dtype).reshape((n_batches,) + shapes[n])
        o += widths[n]
    return res


def gather_batch_results(local, n_batches, rank=None, world_size=None):
    """local: tensor (n_local_batches, ...) for my_batches(n_batches) in that order.
    Returns on rank 0 the (n_batches, ...) tensor in batch order (None elsewhere).
    One collective: an all_gather of equally padded blocks (the payload is a few hundred kB,
    latency-bound; xGMI bandwidth is irrelevant at this size)."""
    if rank is None:
        rank, world_size = world()
    if world_size == 1:
        return local
    per_rank = (n_batches + world_size - 1) // world_size
    # RCCL gathers device tensors; under gloo (CPU tests, or ranks that share one GPU) the blocks travel through the host
    where = local.device if dist.get_backend() == "nccl" else torch.device("cpu")
    pad = torch.zeros((per_rank,) + tuple(local.shape[1:]), dtype=local.dtype, device=where)
    pad[: local.shape[0]] = local.to(where)
    blocks = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(blocks, pad)
    if rank != 0:
        return None
    out = torch.empty((n_batches,) + tuple(local.shape[1:]), dtype=local.dtype, device=where)
    for r in range(world_size):
        ids = my_batches(n_batches, r, world_size)
        if ids:
            out[ids] = blocks[r][: len(ids)]
    return out


def gather_rank_values(value, device=None):
    """One float per rank, on every rank (bench.py: the ranks' step times, whose maximum is the job's time and whose
    spread is the skew).  `device`: where the exchanged tensor lives (the rank's GPU under nccl, None = CPU under gloo)."""
    rank, world_size = world()
    if world_size == 1:
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    every = [torch.zeros_like(mine) for _ in range(world_size)]
    dist.all_gather(every, mine)
    return [float(t.item()) for t in every]


def _fold48(text):
    """48 bits of a string's SHA-1 as a float: exact in the float64 the ranks exchange"""
    import hashlib
    return float(int.from_bytes(hashlib.sha1(text.encode()).digest()[:6], "big"))


def _node_key(env=None):
    """A number that is the same for the ranks of one node and different across nodes.  The launcher knows the node:
    torch.distributed.run exports GROUP_RANK (the node's rank), and RANK // LOCAL_WORLD_SIZE is the same number for
    launchers that export only those two.  Without either, a hash of the host name (which nodes or containers started
    with one fixed --hostname would share: hence only the fallback)."""
    import socket
    env = os.environ if env is None else env
    if env.get("GROUP_RANK", "").isdigit():
        return float(int(env["GROUP_RANK"]))
    if env.get("RANK", "").isdigit() and env.get("LOCAL_WORLD_SIZE", "").isdigit() and int(env["LOCAL_WORLD_SIZE"]) > 0:
        return float(int(env["RANK"]) // int(env["LOCAL_WORLD_SIZE"]))
    return _fold48(socket.gethostname())


def _device_key(device_index):
    """The PHYSICAL device behind a process-local index: its uuid, else its PCI address.  A launcher that gives every rank
    one GPU through HIP_VISIBLE_DEVICES makes every rank see index 0 - on different devices; the process-local index
    (the fallback, and what CPU tests pass) cannot tell that apart from N ranks on one GPU."""
    try:
        if torch.cuda.is_available() and 0 <= int(device_index) < torch.cuda.device_count():
            p = torch.cuda.get_device_properties(int(device_index))
            uuid = getattr(p, "uuid", None)
            if uuid is not None and set(str(uuid)) - set("0-"):
                return _fold48("uuid:" + str(uuid))
            pci = [getattr(p, n, None) for n in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
            if pci[1] is not None:
                return _fold48("pci:" + ":".join(str(v) for v in pci))
    except Exception:
        pass
    return float(int(device_index))


def check_one_device_per_rank(device_index, device=None, node_key=None, device_key=None):
    """Every rank must drive its own GPU: gathers the ranks' (node, device index, physical device) triples and raises when two
    ranks OF ONE NODE hold the same one (a launcher that did not export LOCAL_RANK, or a script that ignored it, would run N
    ranks on cuda:0 and report N times the single-GPU rate as if it scaled).  Devices repeat across the nodes of a multi-node
    job, so the node (`_node_key`: the launcher's node rank, else a host-name hash) is part of the triple; the physical device
    (`_device_key`: uuid / PCI address) lets ranks that all see index 0 through HIP_VISIBLE_DEVICES pass when the devices
    differ, and the index stays in so that a runtime which reported ONE uuid for all its devices could not make a correct job
    fail - the check errs on the side of running.  `node_key` / `device_key` override both in tests.
    Returns the list of device indices by rank."""
    seen = [int(v) for v in gather_rank_values(float(device_index), device)]
    keys = [int(v) for v in gather_rank_values(_device_key(device_index) if device_key is None else float(device_key), device)]
    nodes = [int(v) for v in gather_rank_values(_node_key() if node_key is None else float(node_key), device)]
    triples = list(zip(nodes, seen, keys))
    if len(set(triples)) != len(triples):
        raise RuntimeError(f"ranks share a GPU: device index per rank = {seen}" +
                           (f" (nodes {nodes})" if len(set(nodes)) > 1 else ""))
    return seen
