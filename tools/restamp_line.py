"""Merge freshly collected PMC counters into a bench line that was printed before they were installed.

bench.py takes a kernel's HBM traffic and FP64 work from profiles/pmc_traffic.json and flags them `traffic_stale` when
that file was measured on other kernel sources.  When the line and the counters come from the SAME gpurun call but the
line was printed first (the round's last GPU minutes: bench first, PMC passes after), this recomputes the counter-derived
fields of every roofline object in the line -- `traffic`, `traffic_stale`, `fp64` -- with bench.py's own functions over the
line's own measured durations, and says so in the line (`roofline_counters_note`).  Durations, values and every other
field are left as printed.

    python tools/restamp_line.py gpurun_out/bench_final.json profiles/r04/bench_n1.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def redo(roof, kernels_ms, frames, config):
    kernels = {k: {"avg_ms": v, "launches_per_step": 1, "ms_per_step": v} for k, v in kernels_ms.items()}
    dom = roof["kernel"]
    kernels[dom]["avg_ms"] = roof["avg_launch_ms"]
    roof["traffic"] = bench.measured_traffic(dom, frames, config)
    roof["traffic_stale"] = bench.traffic_stale(config)
    fp = bench.measured_fp64(kernels, frames, config)
    if fp and dom in fp[0] and "fp64" in roof:
        flop, tflops = fp[0][dom]
        pipeline_rate = roof["fp64"].get("pipeline_achieved")
        old_per_frame = roof["fp64"].get("pipeline_flop_per_frame")
        roof["fp64"].update(flop_per_launch=flop, achieved=tflops, frac=tflops / bench.FP64_VECTOR_PEAK_TFLOPS,
                            pipeline_flop_per_frame=fp[1])
        if pipeline_rate and old_per_frame:
            roof["fp64"]["pipeline_achieved"] = pipeline_rate * fp[1] / old_per_frame


def main():
    src, dst = sys.argv[1], sys.argv[2]
    line = json.loads(open(src).read().strip().splitlines()[-1])
    assert line["csrc_hash"] == bench.csrc_hash(), "the line was printed by another tree"
    redo(line["roofline"], line["kernels_ms_per_step"], line["config"]["frames_per_step"], "1")
    for key, leg in line.get("configs", {}).items():
        if isinstance(leg, dict) and "roofline" in leg and "kernels_ms_per_step" in leg:
            redo(leg["roofline"], leg["kernels_ms_per_step"], leg["frames_per_step"], key.split("_")[0])
    line["roofline_counters_note"] = ("traffic / traffic_stale / fp64 of the roofline objects: PMC passes of this tree collected in the same "
                                      "gpurun call AFTER this line was printed, merged by tools/restamp_line.py over the line's own durations")
    open(dst, "w").write(json.dumps(line) + "\n")
    print("stale:", line["roofline"]["traffic_stale"], "traffic:", line["roofline"]["traffic"], "fp64 frac:", line["roofline"]["fp64"]["frac"])
    for key, leg in line.get("configs", {}).items():
        if isinstance(leg, dict) and "roofline" in leg:
            print(" ", key, leg["roofline"].get("traffic_stale"), leg["roofline"].get("traffic"))


if __name__ == "__main__":
    main()
