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


_analyzers = {}


def _default_analyzer():
    """one WorldHip (library context + workspace) per device and process, created on first use"""
    from .api import WorldHip
    device = torch.cuda.current_device()
    if device not in _analyzers:
        _analyzers[device] = WorldHip(device=device)
    return _analyzers[device]


def analyze_sharded(x_list, fs, analyze=None, group=None, frame_period=5.0, **options):
    """The whole multi-GPU recipe in one call (BASELINE configs[3]).

    x_list: every utterance of the job as a 1-D float64 tensor -- the same list on every rank (only this
    rank's share is uploaded and analysed).  analyze(x [b, L] on the device, fs, x_len=..., frame_period=...,
    **options) -> (tpos, f0, sp, ap, n_frames) is WorldHip.analyze unless given.
    Returns (f0 [n, F], sp [n, F, nb], ap [n, F, nb], n_frames [n]) for ALL utterances, in input order, on
    every rank: utterances go to ranks longest-first (partition), each rank runs one batched analysis, and
    one all-gather per array over equal-size padded shards reassembles the results.
    """
    from .api import frame_count
    if analyze is None:
        analyze = _default_analyzer().analyze
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    lengths = [int(x.numel()) for x in x_list]
    n_frames = [frame_count(fs, n, frame_period) for n in lengths]
    parts = partition(lengths, world)
    mine, rows, F = parts[rank], max(len(p) for p in parts), max(n_frames)
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    f0 = sp = ap = None
    if mine:
        xb = torch.zeros((len(mine), max(lengths[i] for i in mine)), dtype=torch.float64, device=device)
        for row, i in enumerate(mine):
            xb[row, :lengths[i]] = x_list[i].to(device)
        _, f0, sp, ap, _ = analyze(xb, fs, x_len=[lengths[i] for i in mine], frame_period=frame_period, **options)

    def shard(t, tail):
        """this rank's [len(mine), F_local, ...] result as an equal-size [rows, F, ...] block"""
        block = torch.zeros((rows, F) + tail, dtype=torch.float64, device=device)
        if t is not None:
            block[:t.shape[0], :t.shape[1]] = t
        return block

    if world > 1:                      # every rank must know the bin count, also one that owns no utterance
        nb_t = torch.tensor([sp.shape[-1] if sp is not None else 0], device=device)
        dist.all_reduce(nb_t, op=dist.ReduceOp.MAX, group=group)
        nb = int(nb_t.item())
    else:
        nb = sp.shape[-1] if sp is not None else 0
    blocks = [shard(f0, ()), shard(sp, (nb,)), shard(ap, (nb,))]
    if world > 1:
        gathered, works = all_gather_results(blocks, group=group, async_op=True)
        wait_all(works)
    else:
        gathered = [b.unsqueeze(0) for b in blocks]
    f0_all, sp_all, ap_all = (assemble(g, parts, len(x_list)) for g in gathered)
    return f0_all, sp_all, ap_all, torch.tensor(n_frames, dtype=torch.int32)
