"""Multi-GPU plumbing: one process per GPU, utterances sharded across
ranks (they are the independent unit: Harvest/DIO need whole utterances), results
reassembled on every rank with one RCCL all-gather per array over xGMI.

torch.distributed only (backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).
Nothing here touches the data path of a single GPU.
"""
import torch
import torch.distributed as dist


def partition(lengths, world_size):
    """Greedy longest-first assignment of utterances to ranks (work ~ length).
    Returns a list of index lists, one per rank; deterministic on every rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(lengths[i])
    for p in parts:
        p.sort()
    return parts


def _gather_one(t, group, async_op):
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if dist.get_backend(group) == "gloo":        # CPU tests
        chunks = list(out.unbind(0))
        work = dist.all_gather(chunks, t.contiguous(), group=group, async_op=async_op)
    else:
        work = dist.all_gather_into_tensor(out, t.contiguous(), group=group, async_op=async_op)
    return out, work


def all_gather_results(tensors, group=None, async_op=False):
    """All-gather equally-shaped per-rank result tensors (pad the local shard to the
    common shape first).  Returns ([world, ...] tensors, works); with async_op the
    collectives overlap whatever is enqueued next -- call wait_all(works) before use."""
    outs, works = [], []
    for t in tensors:
        o, w = _gather_one(t, group, async_op)
        outs.append(o)
        works.append(w)
    return outs, works


def wait_all(works):
    for w in works:
        if w is not None:
            w.wait()


def pad_shard(t, rows):
    """Pad dim 0 of a shard to `rows` (ranks may own different numbers of utterances)."""
    if t.shape[0] == rows:
        return t
    pad = torch.zeros((rows - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat([t, pad], 0)


def assemble(gathered, parts, n_total):
    """Undo `partition`: gathered [world, rows, ...] -> [n_total, ...] in utterance order."""
    out = torch.empty((n_total,) + tuple(gathered.shape[2:]), dtype=gathered.dtype, device=gathered.device)
    for r, idx in enumerate(parts):
        if idx:
            out[torch.tensor(idx, device=gathered.device)] = gathered[r, :len(idx)]
    return out
