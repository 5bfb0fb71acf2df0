"""Helper of tests/test_gpu_parity.py (a subprocess, so that WORLD_HIP_HARVEST_FIR applies to fresh contexts): Harvest of
the golden fixtures through the drop-in C ABI with whatever filter-bank route the environment selects.
Usage: fir_route.py <out.npz>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from util import load_golden                                          # noqa: E402
from world_amd.api import HostAPI                                     # noqa: E402

H = HostAPI()
out = {}
for name in ("vaiueo2d_harvest", "vowel48k_harvest"):
    g = load_golden(name)
    tp, f0 = H.harvest(g["x"], g["fs"], f0_floor=g["f0_floor_est"], frame_period=g["frame_period"])
    out[name + "_tp"], out[name + "_f0"] = tp, f0
np.savez(sys.argv[1], **out)
