"""bench.py's roofline bookkeeping (no GPU): the committed PMC profile must be found by the lookups the
bench line uses, for every kernel that can come out as the dominant one."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_profile_feeds_the_bench_line():
    import bench
    with open(os.path.join(ROOT, "profiles", "r01", "bench_n1.json")) as f:
        line = json.load(f)
    frames = line["config"]["frames_per_step"]
    assert frames == 2001 and line["config"]["workload"].startswith("configs[1]")
    kernels = {k: {"avg_ms": v, "launches_per_step": 1, "ms_per_step": v} for k, v in line["kernels_ms_per_step"].items()}
    for dominant in ("hv_refine", "d4c_groupdelay", "d4c_band", "hv_band_events", "ct_frame"):
        traffic = bench.measured_traffic(dominant, frames)
        assert isinstance(traffic, int) and traffic > 1_000_000, dominant          # bytes per launch
        flops, per_frame = bench.measured_fp64(kernels, frames)
        assert dominant in flops and flops[dominant][0] > 1e8 and 0.5 < flops[dominant][1] < 78.6
    assert 5e6 < per_frame < 1e7                                                    # ~7 MFLOP per output frame
    assert bench.measured_traffic("hv_refine", frames + 1) is None                  # another workload: no claim
    # the committed line itself carries every field of the contract
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert line["dtype"] == "f64" and line["vs_baseline"] is None and line["roofline"]["traffic"] is not None
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-12
