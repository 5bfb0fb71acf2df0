"""Development aid: are two builds of the library bit-identical on the analysis path?

    python tools/bitcmp.py r05 tree [more variants ...]

Each named library (`tree` = world_amd/libworld_hip.so, anything else = world_amd/variants/libworld_hip_<name>.so) runs the
same seeded utterances in its own process -- Harvest + CheapTrick + D4C at 48 / 44.1 / 32 / 16 kHz, DIO + StoneMask at 16
and 22.05 kHz, a 96 kHz utterance for the 8192-point shapes -- and prints one SHA-256 per output array; the parent compares
the digests of every variant with the first one's and reports the arrays that differ (with the largest relative difference,
so that "not bit-identical" can be told from "wrong")."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_of(name):
    return os.path.join(ROOT, "world_amd", "libworld_hip.so") if name == "tree" else \
        os.path.join(ROOT, "world_amd", "variants", f"libworld_hip_{name}.so")


def child(out_path):
    import numpy as np
    import torch
    from world_amd import synth
    from world_amd.api import WorldHip
    wh = WorldHip()
    dev = torch.device("cuda", 0)
    arrays = {}
    for fs, sec, n in ((48000, 3.0, 6), (44100, 2.0, 3), (32000, 2.0, 2), (16000, 3.0, 4), (96000, 1.0, 2), (24000, 1.5, 2)):
        x = torch.stack([synth.utterance(100 * n + u, fs, sec, device=dev) for u in range(n)])
        tp, f0, sp, ap, nf = wh.analyze(x, fs)
        torch.cuda.synchronize()
        for k, v in (("tp", tp), ("f0", f0), ("sp", sp), ("ap", ap)):
            arrays[f"harvest{fs}.{k}"] = v.cpu().numpy()
    for fs in (16000, 22050):
        x = torch.stack([synth.utterance(7 + u, fs, 2.0, device=dev) for u in range(3)])
        tp, f0, sp, ap, nf = wh.analyze(x, fs, f0_method="dio")
        torch.cuda.synchronize()
        for k, v in (("f0", f0), ("sp", sp), ("ap", ap)):
            arrays[f"dio{fs}.{k}"] = v.cpu().numpy()
    np.savez(out_path, **arrays)
    print(json.dumps({k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16] for k, v in arrays.items()}))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        return
    import numpy as np
    names = sys.argv[1:] or ["r05", "tree"]
    digests, files = {}, {}
    for name in names:
        out = f"/tmp/bitcmp_{name}.npz"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out], capture_output=True, text=True,
                           env=dict(os.environ, WORLD_HIP_LIB=lib_of(name)), timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            print(name, "FAILED", (r.stderr or r.stdout)[-800:])
            continue
        digests[name], files[name] = json.loads(lines[-1]), out
    base = names[0]
    if base not in digests:
        sys.exit(1)
    ref = np.load(files[base])
    for name in names[1:]:
        if name not in digests:
            continue
        diff = [k for k in digests[base] if digests[name].get(k) != digests[base][k]]
        if not diff:
            print(f"{name}: bit-identical to {base} on all {len(digests[base])} arrays")
            continue
        got = np.load(files[name])
        print(f"{name}: {len(diff)} of {len(digests[base])} arrays differ from {base}")
        for k in diff:
            a, b = ref[k], got[k]
            rel = float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300))) if a.shape == b.shape else float("nan")
            print(f"   {k:22s} max relative difference {rel:.3e}  ({int(np.sum(a != b))} of {a.size} values)")


if __name__ == "__main__":
    main()
