"""Batch command-line tools over the reference's file formats.

The reference ships one-file-at-a-time programs (examples/parameter_io/f0analysis.cpp,
spanalysis.cpp, apanalysis.cpp, readandsynthesis.cpp); these do the same work for MANY files
per call, batched on the GPU, and read / write the same F0 / SPEC / AP / WAV files:

    python -m world_amd.tools analysis a.wav b.wav ... --outdir params      # -> params/a.f0 a.sp a.ap ...
    python -m world_amd.tools synthesis params/a.f0 params/a.sp params/a.ap -o a_resynth.wav

`analysis` keeps the example programs' option letters where they exist (-f/-c/-s of f0analysis,
-q of spanalysis, -t of apanalysis).  Files are grouped by sampling rate; only their PCM bytes
are uploaded (decoded on the device), and with --code-sp / --code-ap the envelopes are coded on
the device before they come back, so the D2H traffic and the files shrink by 10-17x.
There is no CPU path: without a GPU and the built library this exits with an error.
"""
import argparse
import os
import sys

import numpy as np

from .api import FileAPI, WorldHip, cheaptrick_fft_size


def _analysis(a):
    import torch
    wh, files = WorldHip(), FileAPI()
    os.makedirs(a.outdir, exist_ok=True)
    by_rate = {}
    for path in a.wav:
        by_rate.setdefault(wh.wav_layout(path)[0], []).append(path)
    frames = 0
    for fs, group in sorted(by_rate.items()):
        for at in range(0, len(group), a.batch):
            chunk = group[at:at + a.batch]
            waves = [wh.wavread(path)[0] for path in chunk]           # PCM bytes up, FP64 made on the device
            x = torch.zeros((len(chunk), max(w.numel() for w in waves)), dtype=torch.float64, device=wh.device)
            for row, w in enumerate(waves):
                x[row, :w.numel()] = w
            x_len = np.array([w.numel() for w in waves], dtype=np.int32)
            tpos, f0, sp, ap, nf = wh.analyze(x, fs, x_len=x_len, f0_method=a.f0, frame_period=a.s, f0_floor=a.f,
                                              f0_ceil=a.c, q1=a.q, threshold=a.t)
            fft_size = cheaptrick_fft_size(fs, 71.0)
            sp_dims = ap_dims = 0
            if a.code_sp:
                sp, sp_dims = wh.code_spectral_envelope(sp, fs, fft_size, a.code_sp), a.code_sp
            if a.code_ap:
                ap = wh.code_aperiodicity(ap, fs, fft_size)
                ap_dims = ap.shape[-1]
            tpos, f0, sp, ap = (t.cpu().numpy() for t in (tpos, f0, sp, ap))
            for row, path in enumerate(chunk):
                n, stem = int(nf[row]), os.path.join(a.outdir, os.path.splitext(os.path.basename(path))[0])
                files.write_f0(stem + ".f0", a.s, tpos[row, :n], f0[row, :n], text=a.text)
                files.write_spectral_envelope(stem + ".sp", sp[row, :n], fs, a.s, fft_size, sp_dims)
                files.write_aperiodicity(stem + ".ap", ap[row, :n], fs, a.s, fft_size, ap_dims)
                frames += n
    print(f"{len(a.wav)} file(s), {frames} frames -> {a.outdir}")


def _synthesis(a):
    import torch
    wh, files = WorldHip(), FileAPI()
    fs, fft_size = int(files.header(a.sp, "FS  ")), int(files.header(a.sp, "FFT "))
    frame_period = files.header(a.sp, "FP  ")
    read = files.read_f0(a.f0)
    sp, ap = files.read_spectral_envelope(a.sp), files.read_aperiodicity(a.ap)
    if read is None or sp is None or ap is None:
        sys.exit("synthesis: unreadable parameter file")
    f0 = torch.from_numpy(read[1]).to(wh.device)[None]
    sp, ap = torch.from_numpy(sp).to(wh.device)[None], torch.from_numpy(ap).to(wh.device)[None]
    if int(files.header(a.sp, "NOD ")):
        sp = wh.decode_spectral_envelope(sp, fs, fft_size)
    if int(files.header(a.ap, "NOD ")):
        ap = wh.decode_aperiodicity(ap, fs, fft_size)
    n = f0.shape[1]
    y_length = int(n * frame_period / 1000.0 * fs)              # examples/parameter_io/readandsynthesis.cpp:85
    y = wh.synthesis(f0, sp, ap, np.array([n], dtype=np.int32), fft_size, frame_period, fs,
                     np.array([y_length], dtype=np.int32))
    wh.wavwrite(a.o, y[0, :y_length], fs)
    print(f"{n} frames -> {a.o} ({y_length} samples at {fs} Hz)")


def main(argv=None):
    p = argparse.ArgumentParser(prog="python -m world_amd.tools", description=__doc__.split("\n\n")[0])
    sub = p.add_subparsers(dest="tool", required=True)
    an = sub.add_parser("analysis", help="WAV files -> .f0 / .sp / .ap files")
    an.add_argument("wav", nargs="+")
    an.add_argument("--outdir", default=".")
    an.add_argument("--f0", choices=("harvest", "dio"), default="harvest", help="dio = Dio + StoneMask")
    an.add_argument("-f", type=float, default=71.0, help="floor of the F0 range (Hz)")
    an.add_argument("-c", type=float, default=800.0, help="ceiling of the F0 range (Hz)")
    an.add_argument("-s", type=float, default=5.0, help="frame shift (ms)")
    an.add_argument("-q", type=float, default=-0.15, help="CheapTrick q1")
    an.add_argument("-t", type=float, default=0.85, help="D4C threshold")
    an.add_argument("--text", action="store_true", help="write .f0 as text")
    an.add_argument("--code-sp", type=int, default=0, metavar="D", help="store D mel-cepstral coefficients per frame")
    an.add_argument("--code-ap", action="store_true", help="store band aperiodicities")
    an.add_argument("--batch", type=int, default=64, help="utterances per GPU call")
    an.set_defaults(run=_analysis)
    sy = sub.add_parser("synthesis", help=".f0 + .sp + .ap -> WAV")
    sy.add_argument("f0")
    sy.add_argument("sp")
    sy.add_argument("ap")
    sy.add_argument("-o", default="output.wav")
    sy.set_defaults(run=_synthesis)
    a = p.parse_args(argv)
    a.run(a)


if __name__ == "__main__":
    main()
