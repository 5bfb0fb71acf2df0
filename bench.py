#!/usr/bin/env python3
"""bench.py -- analysis frames/sec of the MI355X WORLD analysis path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--seconds S]

One "step" = one pass of Harvest + CheapTrick + D4C (48 kHz, 5 ms hop, CheapTrick
fft 2048, D4C internal fft 4096) over one batch of synthetic utterances already
resident in HBM, results left in HBM.  The default workload is BASELINE.json
configs[1]: ONE 48 kHz x 10 s utterance per GPU (2001 frames); `--batch B` runs B
utterances per step (the per-GPU share of configs[3] is --batch 128 --seconds 5).
With N > 1 ranks every rank analyses its own utterances (utterances are the
independent unit; weak scaling) and the per-rank f0 / spectrogram /
aperiodicity shards are reassembled on every rank with one RCCL all-gather per
array (north_star), double-buffered so step k's gather overlaps step k+1's compute.

Rank 0 prints ONE JSON line: metric/value (whole-job frames/s), `roofline` for the
kernel that dominates the step (HIP-event timing on the launch stream, taken in a
separate profiled pass) and `cpu_baseline` (the CPU oracle on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with
# more independent jobs in flight than queues, jobs that share a queue run one after the other.  Must be set
# before the runtime initialises (measured on configs[1]: 8 jobs on 16 queues 0.925 ms per job, 6 jobs on the
# default 4 queues 0.98 ms, 8 jobs on 4 queues 1.05 ms).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 48000
FRAME_PERIOD = 5.0
FFT_SIZE = 2048
# compulsory HBM bytes per output frame of the full pipeline at 48 kHz:
# one hop of x read once (240 samples * 8 B) + tpos + f0 + two rows of 1025 doubles written once
BYTES_PER_FRAME = 240 * 8 + 8 + 8 + 2 * 1025 * 8
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)


def cpu_baseline_all_cores(x_np):
    """The reference is re-entrant (no globals), so the fairest whole-box CPU number is one
    analysis per core: P worker PROCESSES (the reference allocates per candidate, threads would
    fight over one heap) each analyse the utterance once, concurrently."""
    from oracle.loader import best_oracle, parallel_analyses
    cores = os.cpu_count() or 1
    procs = max(1, min(cores, 64))
    kind = best_oracle().kind
    frames, dt = parallel_analyses(x_np, FS, FRAME_PERIOD, FFT_SIZE, procs)
    return {"value": frames / dt, "unit": "frames/s", "cores": procs,
            "kind": "reference" if kind == "reference" else "port",
            "sample": f"{procs} concurrent analyses of the same {len(x_np) / FS:.1f} s utterance, one per process, "
                      f"{dt:.1f} s wall (excluding worker start-up)", "host_cores_available": cores}


def measured_traffic(kernel, frames_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, written by tools_profile_summary.py).  FETCH_SIZE is
    doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read);
    WRITE_SIZE is used as reported (uncalibrated).  None when no matching profile exists."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("frames_per_launch") != frames_per_launch:
            return None
        # rocprofv3 prints template arguments ("d4c_groupdelay<4096>"), the library's labels do not
        names = [n for n in t["kernels"] if n == kernel or n.startswith(kernel + "<")]
        if not names:
            return None
        k = t["kernels"][names[0]]
        return int((2.0 * k.get("FETCH_SIZE", 0.0) + k.get("WRITE_SIZE", 0.0)) * 1024.0)
    except (OSError, ValueError, KeyError):
        return None


FP64_VECTOR_PEAK_TFLOPS = 78.6    # MI355X FP64 vector: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz


def measured_fp64(kernels, frames_per_launch):
    """FP64 work per launch from the committed rocprofv3 PMC pass (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64:
    wave-level instructions, 64 lanes each, an FMA counted as two operations) set against the live
    per-launch durations: {kernel: (flop per launch, achieved TFLOP/s)} and the pipeline's flop per frame.
    None when the profile lacks the counters or was taken on another workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if t.get("frames_per_launch") != frames_per_launch:
        return None
    out, total = {}, 0.0
    for name, c in t["kernels"].items():
        if "SQ_INSTS_VALU_FMA_F64" not in c:
            continue
        flop = 64.0 * (c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) +
                       2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
        base = name.split("<")[0]
        if base in kernels and flop > 0:
            k = kernels[base]
            out[base] = (flop, flop / (k["avg_ms"] * 1e-3) / 1e12)
            total += flop * k["launches_per_step"]
    return (out, total / frames_per_launch) if out else None


def cpu_baseline(x_np, reps=1):
    """Time the CPU oracle (the unmodified reference when its in-place build travelled
    here, else this repo's C restatement) on the same utterance, one host core."""
    from oracle.loader import best_oracle
    o = best_oracle()
    t0 = time.perf_counter()
    frames = 0
    for _ in range(reps):
        tp, f0 = o.harvest(x_np, FS, frame_period=FRAME_PERIOD)
        o.cheaptrick(x_np, FS, tp, f0, fft_size=FFT_SIZE)
        o.d4c(x_np, FS, tp, f0, FFT_SIZE)
        frames += len(f0)
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": 1,
            "kind": "reference" if o.kind == "reference" else "port",
            "sample": f"{reps} x ({len(x_np) / FS:.1f} s of 48 kHz audio, Harvest+CheapTrick+D4C, {frames // reps} frames), "
                      f"{getattr(o, 'flags', 'gcc -O2 restatement')}, {dt:.1f} s of CPU time",
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=10.0, help="utterance length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the result all-gather")
    ap.add_argument("--no-extras", action="store_true", help="skip the codec / synthesis legs reported beside the metric")
    ap.add_argument("--streams", type=int, default=8,
                    help="independent analysis jobs in flight per GPU (each step is one job on its own HIP "
                         "stream with its own workspace; 1 = strictly one after the other)")
    args = ap.parse_args()
    args.no_gather_cfg = args.no_gather

    import torch
    import torch.distributed as dist
    from world_amd import distributed as wd
    from world_amd import synth
    from world_amd.api import WorldHip, frame_count

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one rank per GPU; (for a functional check of the N > 1 path on a 1-GPU box, ranks may share a
    # device with WORLD_HIP_BENCH_BACKEND=gloo -- never used for numbers of record)
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(os.environ.get("WORLD_HIP_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    dev = torch.device("cuda", local)
    # synthetic utterances of this rank (distinct per rank and per slot), resident in HBM
    B = args.batch
    xs = [synth.vowel(FS, args.seconds, seed=12345, device=dev) if (rank == 0 and i == 0)
          else synth.utterance(rank * B + i, FS, args.seconds, device=dev) for i in range(B)]
    x = torch.stack(xs).contiguous()
    n = x.shape[1]
    nf = frame_count(FS, n, FRAME_PERIOD)
    # Steps are independent analysis jobs.  `--streams S` keeps S of them in flight: job k
    # runs on HIP stream k % S with its own library context (workspace) and output buffers,
    # so one job's short serial kernels (contour logic, decimation) overlap another job's
    # wide ones.  Every job still does the full work; nothing is cached between steps.
    S = max(1, args.streams)
    nbuf = max(S, 2 if world > 1 else 1)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nbuf)]
    whs = [WorldHip(device=local) for _ in range(nbuf)]
    wh = whs[0]
    sp_bufs = [torch.empty((B, nf, FFT_SIZE // 2 + 1), dtype=torch.float64, device=dev) for _ in range(nbuf)]
    ap_bufs = [torch.empty_like(sp_bufs[0]) for _ in range(nbuf)]
    pending = [None] * nbuf
    counter = [0]
    # one-time initialisation of every slot (workspace allocation, constant tables): not a step
    for k in range(nbuf):
        with torch.cuda.stream(streams[k]):
            whs[k].analyze(x, FS, frame_period=FRAME_PERIOD, sp_out=sp_bufs[k], ap_out=ap_bufs[k])
    torch.cuda.synchronize()

    def step():
        """analysis of this rank's utterances; at N > 1 followed by the asynchronous
        all-gather of (f0, sp, ap) whose completion is awaited when the slot is reused"""
        k = counter[0] % nbuf
        counter[0] += 1
        with torch.cuda.stream(streams[k]):
            if pending[k] is not None:
                wd.wait_all(pending[k][1])
            tpos, f0, sp, ap, _ = whs[k].analyze(x, FS, frame_period=FRAME_PERIOD, sp_out=sp_bufs[k],
                                                 ap_out=ap_bufs[k])
            if world > 1 and not args.no_gather:
                pending[k] = wd.all_gather_results([f0, sp, ap], async_op=True)
        return tpos, f0, sp, ap

    def drain():
        for k in range(nbuf):
            if pending[k] is not None:
                with torch.cuda.stream(streams[k]):
                    wd.wait_all(pending[k][1])
                pending[k] = None

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames_per_step = nf * B * world
    value = frames_per_step * args.steps / dt

    # latency of ONE job with nothing else in flight (not the headline number)
    lat = None
    if rank == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            with torch.cuda.stream(streams[0]):
                whs[0].analyze(x, FS, frame_period=FRAME_PERIOD, sp_out=sp_bufs[0], ap_out=ap_bufs[0])
            torch.cuda.synchronize()
        lat = (time.perf_counter() - t1) / 3 * 1e3

    # ---- roofline leg: per-kernel HIP-event timing of a few extra steps (rank 0) ----
    roofline = None
    kernels = {}
    if rank == 0:
        def lone():
            with torch.cuda.stream(streams[0]):
                whs[0].analyze(x, FS, frame_period=FRAME_PERIOD, sp_out=sp_bufs[0], ap_out=ap_bufs[0])
        prof = wh.profile(lambda: [lone() for _ in range(3)])
        torch.cuda.synchronize()
        kernels = {k: {"launches_per_step": len(v) // 3, "ms_per_step": sum(v) / 3.0, "avg_ms": sum(v) / len(v)}
                   for k, v in prof.items()}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        units = nf * B                              # frames one launch of the dominant kernel covers
        alg_bytes = BYTES_PER_FRAME * units
        achieved = alg_bytes / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(dom, units),
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kernels[dom]["avg_ms"],
                    "note": "FP64-ALU/LDS-bound pipeline (SURVEY.md 8d): ~7 MFLOP/frame (measured, see fp64) vs 18.3 kB/frame"}
        fp64 = measured_fp64(kernels, units)
        if fp64 and dom in fp64[0]:
            flop, tflops = fp64[0][dom]
            roofline["fp64"] = {"flop_per_launch": flop, "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                                "pipeline_flop_per_frame": fp64[1],
                                "pipeline_achieved": fp64[1] * value / world / 1e12,
                                "source": "profiles/pmc_traffic.json (SQ_INSTS_VALU_*_F64) over live durations"}

    # ---- coders behind the path: reported beside the metric, never part of `value` ----
    codec = None
    if rank == 0 and not args.no_extras:
        sp, ap = sp_bufs[0], ap_bufs[0]
        with torch.cuda.stream(streams[0]):
            for _ in range(2):
                mcep = wh.code_spectral_envelope(sp, FS, FFT_SIZE, 60)
                bap = wh.code_aperiodicity(ap, FS, FFT_SIZE)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                mcep = wh.code_spectral_envelope(sp, FS, FFT_SIZE, 60)
                bap = wh.code_aperiodicity(ap, FS, FFT_SIZE)
            e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        cbytes = nf * B * 8 * (2 * (FFT_SIZE // 2 + 1) + mcep.shape[-1] + bap.shape[-1])
        codec = {"workload": f"CodeSpectralEnvelope(60 dims) + CodeAperiodicity on the step's {nf * B} frames",
                 "ms": ms, "frames_per_s": nf * B / (ms * 1e-3),
                 "roofline": {"bound": "hbm", "achieved": cbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": cbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "algorithmic_bytes": cbytes}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle.loader import best_oracle
            orc = best_oracle()
            sp_h, ap_h = sp[0].cpu().numpy(), ap[0].cpu().numpy()
            t1 = time.perf_counter()
            orc.code_spectral_envelope(sp_h, FS, FFT_SIZE, 60)
            orc.code_aperiodicity(ap_h, FS, FFT_SIZE)
            codec["cpu_baseline"] = {"value": nf / (time.perf_counter() - t1), "unit": "frames/s", "cores": 1,
                                     "kind": orc.kind, "sample": f"the same {nf} frames, one call each"}

    # ---- synthesis from the step's device-resident parameters: beside the metric ----
    synthesis = None
    if rank == 0 and not args.no_extras:
        with torch.cuda.stream(streams[0]):
            tpos1, f01, sp1, ap1, nf1 = whs[0].analyze(x, FS, frame_period=FRAME_PERIOD, sp_out=sp_bufs[0],
                                                        ap_out=ap_bufs[0])
            y = wh.synthesis(f01, sp1, ap1, nf1, FFT_SIZE, FRAME_PERIOD, FS, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                y = wh.synthesis(f01, sp1, ap1, nf1, FFT_SIZE, FRAME_PERIOD, FS, n)
            e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        synthesis = {"workload": f"Synthesis() of {B} x {args.seconds:g} s from the analysis outputs in HBM",
                     "ms": ms, "x_realtime": B * args.seconds / (ms * 1e-3),
                     "note": "bound by the bit-faithful serial phase accumulation (one FP64 add per sample, "
                             "36-cycle dependent issue); utterances of a batch share that latency"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle.loader import best_oracle
            orc = best_oracle()
            f0_h, sp_h, ap_h = f01[0].cpu().numpy(), sp1[0].cpu().numpy(), ap1[0].cpu().numpy()
            t1 = time.perf_counter()
            orc.synthesis(f0_h, sp_h, ap_h, FFT_SIZE, FRAME_PERIOD, FS, n)
            synthesis["cpu_baseline"] = {"ms": (time.perf_counter() - t1) * 1e3, "cores": 1, "kind": orc.kind,
                                         "sample": "the same utterance, one call"}

    # ---- the same job through the reference's host-pointer API (the drop-in boundary):
    # x uploaded, results downloaded into row-pointer arrays, one synchronisation per stage.  PCIe-inclusive,
    # reported beside the metric and never part of `value`.
    host_to_host = None
    if rank == 0 and not args.no_extras:
        from world_amd.api import HostAPI
        H = HostAPI()
        x_host = xs[0].cpu().numpy()[:n]

        def host_job():
            tp, f0 = H.harvest(x_host, FS, frame_period=FRAME_PERIOD)
            H.cheaptrick(x_host, FS, tp, f0, fft_size=FFT_SIZE)
            H.d4c(x_host, FS, tp, f0, FFT_SIZE)
            return len(f0)
        host_job()
        t1 = time.perf_counter()
        frames_h = sum(host_job() for _ in range(5))
        dt_h = time.perf_counter() - t1
        host_to_host = {"workload": "Harvest() + CheapTrick() + D4C() on host pointers (libworld_hip.so drop-in symbols), "
                                    "one utterance at a time, PCIe and per-stage synchronisation included",
                        "ms_per_utterance": dt_h / 5 * 1e3, "frames_per_s": frames_h / dt_h}

    cpu = cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        x_host = xs[0].cpu().numpy()
        cpu = cpu_baseline(x_host)
        cpu_all = cpu_baseline_all_cores(x_host)

    barrier()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        out = {
            "metric": "analysis frames/sec (Harvest+CheapTrick+D4C, 48 kHz, 5 ms hop)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: {B} x (48 kHz, {args.seconds:g} s) utterance(s) per GPU, "
                                   f"Harvest+CheapTrick+D4C, fft_size=2048, frame_period=5 ms, inputs/outputs in HBM",
                       "frames_per_step": frames_per_step, "utterances_per_gpu": B, "jobs_in_flight": S,
                       "parallelism": f"utterance-sharded x{world}" + (
                           ", async RCCL all-gather of f0/sp/ap per step" if world > 1 and not args.no_gather_cfg
                           else ", no collective")},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(
                kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "single_job_latency_ms": lat, "codec": codec, "synthesis": synthesis,
            "host_to_host": host_to_host,
            "workspace_bytes": sum(w.workspace_bytes() for w in whs),
        }
        print(json.dumps(out))


if __name__ == "__main__":
    main()
