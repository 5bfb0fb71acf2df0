"""Multi-GPU plumbing: one process per GPU, utterances sharded across
ranks (they are the independent unit: Harvest/DIO need whole utterances), results
reassembled on every rank by all-gathers over xGMI of packed record blocks

    row = [ tpos, f0, sp[0 .. nb), ap[0 .. nb) ]          (2 + 2 nb float64)

A rank's share runs in sub-batches ("chunks"); chunk k's records are written by the
stage kernels STRAIGHT into the rank's slice of chunk k's receive buffer
(include/world_hip.h: world_hip_analyze_packed -- no dense spectrogram, no pack pass),
and chunk k is all-gathered -- in place, asynchronously -- while chunk k + 1 is being
analysed: the exchange hides behind the compute except for the last chunk's.  Only
the chunks' row counts are equalised across ranks (longest-first partitioning keeps
them within an utterance of each other); utterances are never padded.

torch.distributed only (backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).
Nothing here touches the data path of a single GPU.  A single PROCESS driving several
GPUs does the same without torch: world_hip_analyze_sharded (host threads + peer copies).
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


def chunk_sizes(n, sub_batch, taper=True):
    """Sub-batch sizes for a share of n utterances: full sub-batches, and the LAST one tapered into halves
    (sb/2, sb/4, sb/4).  Chunk k's all-gather runs under chunk k + 1's analysis, so only the last chunk's exchange is
    exposed: tapering shrinks it from 1/4 of a 128-utterance share (32, 32, 32, 32) to 1/16 (32, 32, 32, 16, 8, 8).
    Tails of fewer than 8 utterances stay whole -- smaller batches stop filling the chip.  (api.hip: chunk_sizes is the
    same schedule for the C driver.)"""
    sb = max(1, int(sub_batch))
    out = []
    while n > sb:
        out.append(sb)
        n -= sb
    if taper and n >= 8:
        a = (n + 1) // 2
        b = (n - a + 1) // 2
        out += [v for v in (a, b, n - a - b) if v > 0]
    elif n > 0:
        out.append(n)
    return out


def chunks_of(parts, sub_batch, taper=True):
    """parts[r] cut into sub-batches (chunk_sizes of the LARGEST share, the same cuts on every rank):
    chunks[k][r] = rank r's utterances of chunk k (every rank has the same number of chunks; late ones may be
    short or empty on ranks with a smaller share)"""
    n = max((len(p) for p in parts), default=0)
    cuts, lo = [], 0
    for size in chunk_sizes(n, sub_batch, taper):
        cuts.append((lo, lo + size))
        lo += size
    return [[p[a:b] for p in parts] for a, b in cuts]


def _gather_in_place(out, rank, group, async_op):
    """all-gather where rank r's contribution already sits in out[r] (no send buffer, no copy)"""
    if dist.get_backend(group) == "gloo":
        src = out[rank].clone()                  # gloo: CPU tests and the 1-GPU functional check (host staging)
        if out.is_cuda:
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather(list(host.unbind(0)), src.cpu(), group=group)
            out.copy_(host)
            return None
        return dist.all_gather(list(out.unbind(0)), src, group=group, async_op=async_op)
    return dist.all_gather_into_tensor(out, out[rank], group=group, async_op=async_op)


def all_gather_results(tensors, group=None, async_op=False):
    """All-gather equally-shaped per-rank tensors.  Returns ([world, ...] tensors, works); with async_op
    the collectives overlap whatever is enqueued next -- call wait_all(works) before use."""
    outs, works = [], []
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    for t in tensors:
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        out[rank].copy_(t)
        works.append(_gather_in_place(out, rank, group, async_op))
        outs.append(out)
    return outs, works


def wait_all(works):
    for w in works:
        if w is not None:
            w.wait()


WIRE_COLS = {"f64": lambda nb: 2 + 2 * nb, "f32": lambda nb: 2 + nb}
CODED_DIMENSIONS = 60            # mel-cepstral coefficients per frame on the "coded" wire (the reference's usual choice)


def number_of_aperiodicities(fs):
    """GetNumberOfAperiodicities (reference src/codec.cpp:212-215)"""
    return int(min(15000.0, fs / 2.0 - 3000.0) / 3000.0)


def wire_columns(wire, nb, fs=None, dimensions=CODED_DIMENSIONS):
    """doubles per record on the wire: "f64" 2 + 2 nb, "f32" 2 + nb, "coded" 2 + dimensions + bands (include/world_hip.h:
    world_hip_analyze_coded -- [tpos, f0, mel-cepstrum, band aperiodicity], the reference's coders applied on the device)"""
    if wire == "coded":
        if fs is None:
            raise ValueError("the coded wire format needs the sampling rate")
        return 2 + dimensions + number_of_aperiodicities(fs)
    return WIRE_COLS[wire](nb)


def record_views(rec, nb, wire="f64"):
    """(tpos, f0, sp, ap) views of packed records rec [n, cols] (float64 storage).  wire "f64": cols = 2 + 2 nb, all
    doubles; wire "f32": cols = 2 + nb doubles = 16 + 8 nb bytes, the spectra stored as float32 (include/world_hip.h:
    world_hip_analyze_packed) -- sp / ap come back as float32 views of the same memory."""
    if wire == "f64":
        return rec[:, 0], rec[:, 1], rec[:, 2:2 + nb], rec[:, 2 + nb:2 + 2 * nb]
    if isinstance(wire, tuple):                        # ("coded", dimensions): (tpos, f0, mel-cepstrum [n, D], band aperiodicity [n, bands])
        return rec[:, 0], rec[:, 1], rec[:, 2:2 + wire[1]], rec[:, 2 + wire[1]:]
    assert wire == "f32" and rec.shape[-1] == 2 + nb
    f32 = rec.view(torch.float32)                      # [n, 2 cols]: float j of a record = bytes 4 j ..
    return rec[:, 0], rec[:, 1], f32[:, 4:4 + nb], f32[:, 4 + nb:4 + 2 * nb]


class ShardedResult:
    """Every utterance's analysis on this rank, as views into the gathered chunk buffers (nothing is copied again).

    blocks  list over chunks of [world][rows_k][cols] float64 (cols by wire format: WIRE_COLS)
    where   {utterance index: (chunk, rank, first record, n_frames)}
    """

    def __init__(self, blocks, where, n_frames, nb, wire="f64"):
        self.blocks, self.where, self.n_frames, self.nb, self.wire = blocks, where, n_frames, nb, wire

    def __len__(self):
        return len(self.n_frames)

    def utterance(self, i):
        """(tpos [n], f0 [n], sp [n, nb], ap [n, nb]) of utterance i: views, no copy (sp / ap float32 on the f32 wire; on
        the coded wire: mel-cepstrum [n, D] and band aperiodicity [n, bands] -- decode with WorldHip.decode_*)"""
        k, r, first, n = self.where[i]
        return record_views(self.blocks[k][r, first:first + n], self.nb, self.wire)

    def dense(self):
        """(f0 [n_utt, F], sp [n_utt, F, nb], ap [n_utt, F, nb], n_frames) padded to the longest utterance
        (one more full-size copy: for callers that want the batched API's layout back)"""
        n, F = len(self.n_frames), (max(self.n_frames) if self.n_frames else 0)
        dev = self.blocks[0].device if self.blocks else torch.device("cpu")
        f0 = torch.zeros((n, F), dtype=torch.float64, device=dev)
        sp = torch.zeros((n, F, self.nb), dtype=torch.float64, device=dev)
        ap = torch.zeros((n, F, self.nb), dtype=torch.float64, device=dev)
        for i in range(n):
            if i in self.where:
                _, f, s, a = self.utterance(i)
                k = f.shape[0]
                f0[i, :k], sp[i, :k], ap[i, :k] = f, s.to(torch.float64), a.to(torch.float64)
        return f0, sp, ap, torch.tensor(self.n_frames, dtype=torch.int32)


_analyzers = {}
_buffers = {}       # (device, world, rows per chunk, columns) -> the chunks' receive buffers, reused from step to step


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


_lanes = {}


def _default_lanes(packer=None, coded=0):
    """two (stream, analyze_packed) lanes per device and process: the caller's analyser (or the default one) and a second
    library context, each bound to its own stream (coded > 0: analyze_coded with that many coefficients)"""
    import functools
    device = torch.cuda.current_device()
    wh = packer or _default_analyzer()                 # a WorldHip keeps one library context per stream it is used on
    # keyed by the analyser ITSELF (kept alive by the entry): an id() can name another object after garbage collection
    key = (device, wh, coded)
    if key not in _lanes:
        run = functools.partial(wh.analyze_coded, number_of_dimensions=coded) if coded else wh.analyze_packed
        _lanes[key] = [(torch.cuda.Stream(device=device), run), (torch.cuda.Stream(device=device), run)]
    return _lanes[key]


def _store_records(packer, tpos, f0, sp, ap, nf, block):
    """records of one batched analysis into block[0:]: the library's kernel on the GPU, indexing on CPU tensors"""
    nb = sp.shape[-1]
    if 2 + 2 * nb != block.shape[-1]:
        # every rank sized its receive buffers from `bins`: a mismatch here would otherwise surface as a hang of the ranks
        # that own no utterance of this chunk
        raise ValueError(f"analyze() returned {nb} bins per frame, the job was set up for {(block.shape[-1] - 2) // 2}: "
                         "pass bins= to analyze_sharded when the analysis does not use the default fft size")
    if block.is_cuda:
        (packer or _default_analyzer()).pack_results(tpos, f0, sp, ap, nf, block, 0)
        return
    row = 0
    for u, n in enumerate(int(k) for k in nf):
        rec = block[row:row + n]
        rec[:, 0], rec[:, 1] = (tpos[u, :n] if tpos is not None else 0.0), f0[u, :n]
        rec[:, 2:2 + nb], rec[:, 2 + nb:] = sp[u, :n], ap[u, :n]
        row += n


def analyze_sharded(x_list, fs, analyze=None, group=None, frame_period=5.0, lengths=None, sub_batch=32, gather=True,
                    timings=None, packer=None, bins=None, analyze_packed=None, lanes=None, wire="f64", taper=True,
                    own_buffers=False, exchange_single_rank=False, coded_dimensions=CODED_DIMENSIONS, **options):
    """The whole multi-GPU recipe in one call (BASELINE configs[3]).

    x_list   every utterance of the job as a 1-D float64 tensor: a list (the same on every rank), or -- with
             `lengths` given for ALL utterances -- a dict {index: tensor} that need only hold this rank's share
             (what partition(lengths, world)[rank] names), so that no rank materialises the whole job.
    analyze_packed  (x [b, L], fs, block [rows, 2 + 2 nb], x_len=..., frame_period=..., **options) -> n_frames: writes
             the batch's records straight into `block` (WorldHip.analyze_packed: the default on a GPU).
    analyze  alternative for callers without a packed analysis (the CPU tests): (x, fs, x_len=..., frame_period=...,
             **options) -> (tpos, f0, sp, ap, n_frames); its dense results are packed by `packer` / on the host.
    sub_batch  utterances per chunk = per batched call AND per all-gather (chunk k's exchange overlaps chunk k+1's compute)
    lanes    list of (torch.cuda.Stream, analyze_packed) pairs: consecutive chunks alternate between them, so that the serial
             tail of one chunk's analysis (Harvest's contour kernels run one wavefront per utterance: a quarter of a
             32-utterance chunk's time with the chip nearly idle) overlaps the wide kernels of the next chunk.  Default on a
             GPU: two lanes, each with its own library context.
    bins     spectrogram bins per frame (default: fft/2+1 of fs)
    wire     record format on the links: "f64" (default; [tpos, f0, sp[nb], ap[nb]] doubles), "f32" (the spectra
             rounded once to float32 by the stage kernels: half the bytes, 6e-8 relative -- the contract is 1e-4) or "coded"
             ([tpos, f0, mel-cepstrum[coded_dimensions], band aperiodicity]: the reference's CodeSpectralEnvelope /
             CodeAperiodicity applied on the device before anything leaves it -- 31 x fewer bytes at 48 kHz, lossy by design)
    taper    cut the last sub-batch into halves so that the one exposed exchange is small (chunk_sizes)
    exchange_single_rank  run the all-gathers even in a process group of ONE rank (a functional check of the collective
             path -- RCCL, the aliased in-place all_gather_into_tensor -- on a 1-GPU box; never needed for results)
    own_buffers  True: the result owns freshly allocated receive buffers (a caller that keeps step N's result while step
             N + 1 runs); False (default): the buffers are cached per job shape and the NEXT call with the same shape
             overwrites them -- a 16.8 GB set is not reallocated every step
    timings  optional dict, accumulates: "compute_ms" (host wall clock of the analysis calls, device synchronised at the
             end), "exchange_exposed_ms" (device time the compute stream spent waiting for all-gathers after its last
             analysis: what the overlap did NOT hide), "exchange_ms" (= exposed), "steps", "gathered_bytes" (bytes this
             rank RECEIVED from the other ranks), "last_chunk_bytes" (those of the final, exposed all-gather)
    Returns a ShardedResult covering ALL utterances on every rank (gather=False: this rank's only).  Its views point into
    receive buffers that the next call for the same job shape reuses (a 16.8 GB set is not reallocated every step).
    """
    from .api import cheaptrick_fft_size, frame_count
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    if lengths is None:
        lengths = [int(x.numel()) for x in x_list]
    n_utt = len(lengths)
    n_frames = [frame_count(fs, n, frame_period) for n in lengths]
    nb = bins or cheaptrick_fft_size(fs) // 2 + 1
    if wire not in WIRE_COLS and wire != "coded":
        raise ValueError(f"wire format {wire!r}: expected one of {sorted(WIRE_COLS) + ['coded']}")
    if wire != "f64" and analyze is not None:
        raise ValueError("the narrow wire formats are written by the stage kernels: they need analyze_packed, not analyze")
    cols = wire_columns(wire, nb, fs, coded_dimensions)
    coded = coded_dimensions if wire == "coded" else 0
    if coded:
        wire = ("coded", coded)                        # what ShardedResult / record_views need to cut a record
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if n_utt == 0:
        return ShardedResult([], {}, [], nb, wire)
    if analyze is None and analyze_packed is None:
        import functools
        a = packer or _default_analyzer()
        analyze_packed = functools.partial(a.analyze_coded, number_of_dimensions=coded) if coded else a.analyze_packed
        if lanes is None and device.type == "cuda":
            lanes = _default_lanes(packer, coded)
    if lanes is not None and not lanes:
        lanes = None
    parts = partition(lengths, world)
    chunks = chunks_of(parts, sub_batch, taper)
    rows = [[sum(n_frames[i] for i in c[r]) for r in range(world)] for c in chunks]
    rows_k = [max(r) for r in rows]
    where = {}
    for k, c in enumerate(chunks):
        for r in range(world):
            row = 0
            for i in c[r]:
                where[i] = (k, r, row, n_frames[i])
                row += n_frames[i]
    exchange = gather and (world > 1 or (on and exchange_single_rank))
    key = (str(device), world if exchange else 1, tuple(rows_k), cols)
    bufs = None if own_buffers else _buffers.get(key)
    if bufs is None:
        if not own_buffers:
            _buffers.clear()                       # one job shape at a time: a 16.8 GB set is not kept beside the next one
        bufs = [torch.zeros((key[1], max(1, rk), cols), dtype=torch.float64, device=device) for rk in rows_k]
        if not own_buffers:
            _buffers[key] = bufs
    me = rank if exchange else 0
    cuda = device.type == "cuda"
    if timings is not None and cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    works = []
    caller = torch.cuda.current_stream(device) if (lanes and cuda) else None
    if lanes:
        for st, _ in lanes:
            st.wait_stream(caller)                     # the inputs were produced in the caller's stream order

    def one_chunk(k, c, run_packed):
        idx = c[rank]
        block = bufs[k][me]
        if idx:
            xb = torch.zeros((len(idx), max(lengths[i] for i in idx)), dtype=torch.float64, device=device)
            for j, i in enumerate(idx):
                xb[j, :lengths[i]] = x_list[i].to(device)
            xl = [lengths[i] for i in idx]
            if run_packed is not None:
                run_packed(xb, fs, block, x_len=xl, frame_period=frame_period, **options)
            else:
                tpos, f0, sp, ap, _ = analyze(xb, fs, x_len=xl, frame_period=frame_period, **options)
                _store_records(packer, tpos, f0, sp, ap, [n_frames[i] for i in idx], block)
        if exchange:
            works.append(_gather_in_place(bufs[k], rank, group, True))     # in flight while the next chunk is analysed

    for k, c in enumerate(chunks):
        if lanes:
            st, run = lanes[k % len(lanes)]
            with torch.cuda.stream(st):
                one_chunk(k, c, run)
        else:
            one_chunk(k, c, analyze_packed)
    if lanes:
        for st, _ in lanes:
            caller.wait_stream(st)                     # the results are complete in the caller's stream order
    ev0 = ev1 = None
    if timings is not None and cuda:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    wait_all(works)
    if ev1 is not None:
        ev1.record()
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    if timings is not None:
        exposed = ev0.elapsed_time(ev1) if ev0 is not None else 0.0
        timings["compute_ms"] = timings.get("compute_ms", 0.0) + (t1 - t0) * 1e3 - exposed
        timings["exchange_exposed_ms"] = timings.get("exchange_exposed_ms", 0.0) + exposed
        timings["exchange_ms"] = timings.get("exchange_ms", 0.0) + exposed
        timings["steps"] = timings.get("steps", 0) + 1
        if exchange:
            per_chunk = [(world - 1) * max(1, rk) * cols * 8 for rk in rows_k]
            timings["gathered_bytes"] = timings.get("gathered_bytes", 0) + sum(per_chunk)
            timings["last_chunk_bytes"] = timings.get("last_chunk_bytes", 0) + per_chunk[-1]
    if world > 1 and not gather:                  # this rank's utterances only
        local = {i: (w[0], 0, w[2], w[3]) for i, w in where.items() if w[1] == rank}
        return ShardedResult(bufs, local, n_frames, nb, wire)
    return ShardedResult(bufs, where, n_frames, nb, wire)


def frame_ranges(n_frames, world, align=64):
    """[lo, hi) of every rank's share of one utterance's frames: contiguous, equal to within `align` frames (a multiple of
    the wavefront: CheapTrick's serial prefix sums walk 64 frames side by side), empty for ranks beyond the end"""
    per = -(-n_frames // world)
    per = -(-per // align) * align
    return [(min(n_frames, r * per), min(n_frames, (r + 1) * per)) for r in range(world)]


def analyze_long_sharded(x, fs, group=None, frame_period=5.0, wire="f64", harvest=None, spectral_range=None, sub_frames=4096,
                         timings=None, **options):
    """Frame-level sharding of ONE long utterance (last sentence): Harvest needs the whole utterance and is
    cheap next to the spectral stages (a tenth of their time), so EVERY rank runs it (identical F0 on every rank, nothing
    to broadcast); CheapTrick and D4C are independent per frame given F0 (reference src/cheaptrick.cpp:207-216,
    src/d4c.cpp:378-400), so rank r analyses frames frame_ranges()[r] only -- in sub-ranges of `sub_frames`, written by the
    stage kernels as packed records straight into its slice of the sub-range's receive buffer and all-gathered in place
    while the next sub-range is analysed.  The positions in the reference's randn() stream are those of the whole
    utterance, so the reassembled result is BIT-IDENTICAL to a lone analysis (f32 wire: rounded once).

    x        1-D float64 tensor, the same on every rank
    harvest / spectral_range  (tests on CPU) replacements for WorldHip.harvest / WorldHip.spectral_packed_range
    Returns (tpos [n], f0 [n], sp [n, nb], ap [n, nb]) -- sp / ap gathered from all ranks (one copy out of the receive
    buffers: the ranks' ranges are concatenated)."""
    from .api import cheaptrick_fft_size, frame_count
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    if wire not in WIRE_COLS:
        raise ValueError(f"wire format {wire!r}: expected one of {sorted(WIRE_COLS)}")
    n = int(x.numel())
    nf = frame_count(fs, n, frame_period)
    nb = cheaptrick_fft_size(fs) // 2 + 1
    cols = WIRE_COLS[wire](nb)
    cuda = x.is_cuda
    if harvest is None or spectral_range is None:
        wh = _default_analyzer()
        harvest = harvest or (lambda xb: wh.harvest(xb, fs, frame_period=frame_period, **{k: v for k, v in options.items() if k in ("f0_floor", "f0_ceil")}))
        spectral_range = spectral_range or (lambda xb, tp, f0, block, lo, hi, reuse=False: wh.spectral_packed_range(
            xb, fs, tp, f0, [nf], block, lo, hi, reuse_offsets=reuse,
            **{k: v for k, v in options.items() if k in ("q1", "threshold")}))
    if timings is not None and cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    xb = x.reshape(1, -1).contiguous()
    tpos, f0 = harvest(xb)[:2]                                   # [1, nf] each, on every rank
    ranges = frame_ranges(nf, world)
    lo, hi = ranges[rank]
    span = max(h - l for l, h in ranges)
    cuts = [(a, min(span, a + sub_frames)) for a in range(0, span, sub_frames)] or [(0, 0)]
    bufs = [torch.zeros((world, max(1, b - a), cols), dtype=torch.float64, device=x.device) for a, b in cuts]
    works = []
    prepared = False           # the offset scans and D4C's LoveTrain pass cover every frame: once per call, not per sub-range
    for k, (a, b) in enumerate(cuts):
        l, h = min(hi, lo + a), min(hi, lo + b)
        if h > l:
            spectral_range(xb, tpos, f0, bufs[k][rank], l, h, prepared)
            prepared = True
        if world > 1:
            works.append(_gather_in_place(bufs[k], rank, group, True))
    wait_all(works)
    sp_parts, ap_parts = [], []
    for r, (l, h) in enumerate(ranges):
        for k, (a, b) in enumerate(cuts):
            m = min(h, l + b) - min(h, l + a)
            if m > 0:
                _, _, s, p = record_views(bufs[k][r, :m], nb, wire)
                sp_parts.append(s)
                ap_parts.append(p)
    sp = torch.cat(sp_parts) if sp_parts else torch.zeros((0, nb), dtype=torch.float64, device=x.device)
    ap = torch.cat(ap_parts) if ap_parts else torch.zeros((0, nb), dtype=torch.float64, device=x.device)
    if timings is not None:
        if cuda:
            torch.cuda.synchronize()
        timings["ms"] = timings.get("ms", 0.0) + (time.perf_counter() - t0) * 1e3
        timings["steps"] = timings.get("steps", 0) + 1
    return tpos[0, :nf], f0[0, :nf], sp, ap
