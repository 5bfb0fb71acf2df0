"""Multi-GPU plumbing: one process per GPU, utterances sharded across
ranks (they are the independent unit: Harvest/DIO need whole utterances), results
reassembled on every rank with ONE all-gather over xGMI of one packed block per rank

    row = [ tpos, f0, sp[0 .. nb), ap[0 .. nb) ]          (2 + 2 nb float64)

holding the rank's valid frames back to back (include/world_hip.h: world_hip_pack_results;
no padding to the longest utterance -- only the ranks' total frame counts are equalised,
and longest-first partitioning keeps those within one utterance of each other).

torch.distributed only (backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).
Nothing here touches the data path of a single GPU.  A single PROCESS driving several
GPUs does the same exchange without torch: world_hip_allgather_blocks (peer copies).
"""
import time

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
    if dist.get_backend(group) == "gloo":        # CPU tests; device tensors are staged through the host (gloo has no
        if t.is_cuda:                            # device all-gather): the functional check of bench.py --gpus N on a 1-GPU box
            host = torch.empty((world,) + tuple(t.shape), dtype=t.dtype)
            dist.all_gather(list(host.unbind(0)), t.contiguous().cpu(), group=group)
            out.copy_(host)
            return out, None
        chunks = list(out.unbind(0))
        work = dist.all_gather(chunks, t.contiguous(), group=group, async_op=async_op)
    else:
        work = dist.all_gather_into_tensor(out, t.contiguous(), group=group, async_op=async_op)
    return out, work


def all_gather_results(tensors, group=None, async_op=False):
    """All-gather equally-shaped per-rank tensors.  Returns ([world, ...] tensors, works); with async_op
    the collectives overlap whatever is enqueued next -- call wait_all(works) before use."""
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


class ShardedResult:
    """Every utterance's analysis on this rank, as views into the gathered blocks (nothing is copied again).

    blocks  [world][rows_max][2 + 2 nb] float64 -- rank r's records are blocks[r, :rank_rows[r]]
    where    {utterance index: (rank, first record, n_frames)}
    """

    def __init__(self, blocks, where, n_frames, nb):
        self.blocks, self.where, self.n_frames, self.nb = blocks, where, n_frames, nb

    def __len__(self):
        return len(self.n_frames)

    def utterance(self, i):
        """(tpos [n], f0 [n], sp [n, nb], ap [n, nb]) of utterance i: views, no copy"""
        r, first, n = self.where[i]
        rec = self.blocks[r, first:first + n]
        return rec[:, 0], rec[:, 1], rec[:, 2:2 + self.nb], rec[:, 2 + self.nb:2 + 2 * self.nb]

    def dense(self):
        """(f0 [n_utt, F], sp [n_utt, F, nb], ap [n_utt, F, nb], n_frames) padded to the longest utterance
        (one more full-size copy: for callers that want the batched API's layout back)"""
        n, F = len(self.n_frames), (max(self.n_frames) if self.n_frames else 0)
        dev = self.blocks.device
        f0 = torch.zeros((n, F), dtype=torch.float64, device=dev)
        sp = torch.zeros((n, F, self.nb), dtype=torch.float64, device=dev)
        ap = torch.zeros((n, F, self.nb), dtype=torch.float64, device=dev)
        for i in range(n):
            if i in self.where:
                _, f, s, a = self.utterance(i)
                k = f.shape[0]
                f0[i, :k], sp[i, :k], ap[i, :k] = f, s, a
        return f0, sp, ap, torch.tensor(self.n_frames, dtype=torch.int32)


_analyzers = {}


def _default_analyzer():
    """one WorldHip (library context + workspace) per device and process, created on first use"""
    from .api import WorldHip
    device = torch.cuda.current_device()
    if dist.is_available() and dist.is_initialized() and torch.cuda.device_count() > 1:
        # every rank silently analysing on GPU 0 is the classic mistake: insist on set_device(local_rank)
        import os
        local = os.environ.get("LOCAL_RANK")
        if local is not None and int(local) % torch.cuda.device_count() != device:
            raise RuntimeError(f"rank with LOCAL_RANK={local} is on cuda:{device}: call torch.cuda.set_device(local_rank) first")
    if device not in _analyzers:
        _analyzers[device] = WorldHip(device=device)
    return _analyzers[device]


def _pack(wh, tpos, f0, sp, ap, nf, block, first_row):
    """records of one batched analysis into block[first_row:]: the library's kernel on the GPU, indexing on
    CPU tensors (the gloo tests)"""
    nb = sp.shape[-1]
    if block.is_cuda:
        if wh is None:
            wh = _default_analyzer()
        wh.pack_results(tpos, f0, sp, ap, nf, block, first_row)
        return
    row = first_row
    for u, n in enumerate(int(k) for k in nf):
        rec = block[row:row + n]
        rec[:, 0], rec[:, 1] = (tpos[u, :n] if tpos is not None else 0.0), f0[u, :n]
        rec[:, 2:2 + nb], rec[:, 2 + nb:] = sp[u, :n], ap[u, :n]
        row += n


def analyze_sharded(x_list, fs, analyze=None, group=None, frame_period=5.0, lengths=None, sub_batch=128, gather=True,
                    timings=None, packer=None, bins=None, **options):
    """The whole multi-GPU recipe in one call (BASELINE configs[3]).

    x_list   every utterance of the job as a 1-D float64 tensor: a list (the same on every rank), or -- with
             `lengths` given for ALL utterances -- a dict {index: tensor} that need only hold this rank's share
             (what partition(lengths, world)[rank] names), so that no rank materialises the whole job.
    analyze  (x [b, L] on the device, fs, x_len=..., frame_period=..., **options) -> (tpos, f0, sp, ap, n_frames);
             WorldHip.analyze unless given.  A rank's share runs in batched calls of <= sub_batch utterances.
    bins     spectrogram bins per frame; needed only by a rank that owns no utterance (default: fft/2+1 of fs)
    timings  optional dict: accumulates "compute_ms", "exchange_ms" (host wall clock, device synchronised) and "steps"
    Returns a ShardedResult covering ALL utterances on every rank (gather=False: this rank's only): utterances go
    to ranks longest-first, every rank packs its frames into one block, one all-gather moves the blocks.
    """
    from .api import cheaptrick_fft_size, frame_count
    if analyze is None:
        analyze = _default_analyzer().analyze
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    if lengths is None:
        lengths = [int(x.numel()) for x in x_list]
    n_utt = len(lengths)
    n_frames = [frame_count(fs, n, frame_period) for n in lengths]
    nb = bins or cheaptrick_fft_size(fs) // 2 + 1
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if n_utt == 0:
        return ShardedResult(torch.zeros((world, 0, 2 + 2 * nb), dtype=torch.float64, device=device), {}, [], nb)
    parts = partition(lengths, world)
    mine = parts[rank]
    rank_rows = [sum(n_frames[i] for i in p) for p in parts]
    rows_max = max(rank_rows)
    where = {}
    for r, p in enumerate(parts):
        row = 0
        for i in p:
            where[i] = (r, row, n_frames[i])
            row += n_frames[i]

    def sync():
        if timings is not None and device.type == "cuda":
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    block = None
    row = 0
    for lo in range(0, len(mine), max(1, sub_batch)):
        idx = mine[lo:lo + max(1, sub_batch)]
        xb = torch.zeros((len(idx), max(lengths[i] for i in idx)), dtype=torch.float64, device=device)
        for k, i in enumerate(idx):
            xb[k, :lengths[i]] = x_list[i].to(device)
        tpos, f0, sp, ap, nf = analyze(xb, fs, x_len=[lengths[i] for i in idx], frame_period=frame_period, **options)
        if block is None:
            nb = sp.shape[-1]
            block = torch.zeros((rows_max, 2 + 2 * nb), dtype=torch.float64, device=device)
        _pack(packer, tpos, f0, sp, ap, [n_frames[i] for i in idx], block, row)
        row += sum(n_frames[i] for i in idx)
    if block is None:                          # a rank that owns nothing still takes part in the collective
        block = torch.zeros((rows_max, 2 + 2 * nb), dtype=torch.float64, device=device)
    sync()
    t1 = time.perf_counter()
    if world > 1 and gather:
        (blocks,), works = all_gather_results([block], group=group, async_op=True)
        wait_all(works)
    elif world > 1:
        blocks = None                          # nothing exchanged
    else:
        blocks = block.unsqueeze(0)
    sync()
    if timings is not None:
        timings["compute_ms"] = timings.get("compute_ms", 0.0) + (t1 - t0) * 1e3
        timings["exchange_ms"] = timings.get("exchange_ms", 0.0) + (time.perf_counter() - t1) * 1e3
        timings["steps"] = timings.get("steps", 0) + 1
    if blocks is None:                          # gather=False at world > 1: this rank's utterances only
        local = {i: (0, w[1], w[2]) for i, w in where.items() if w[0] == rank}
        return ShardedResult(block.unsqueeze(0), local, n_frames, nb)
    return ShardedResult(blocks, where, n_frames, nb)
