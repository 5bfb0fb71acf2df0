"""Generate tests/golden/*.npz from the UNMODIFIED reference.

Run in the build container only (needs /root/reference, compiled in place into
oracle/_ref/libworld_ref.so by oracle/Makefile):

    python tests/golden/make_golden.py

Every fixture stores its input samples as int16 (x = q / 32768, the value the
reference's wavread returns, tools/audioio.cpp:236-249), the exact option
values used, and the reference outputs.  Dense spectrogram/aperiodicity are
kept for a subset of frame rows (`rows`) to keep the files small, plus
whole-array checksums.
"""
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.loader import RefOracle, build  # noqa: E402
from world_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def wav_int16(path):
    w = wave.open(path)
    assert w.getsampwidth() == 2 and w.getnchannels() == 1
    return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy(), w.getframerate()


def analyse(R, q, fs, f0_method, f0_floor_est, row_step, frame_period=5.0, q1=-0.15, threshold=0.85):
    x = q.astype(np.float64) / 32768.0
    out = {"q": q, "fs": fs, "f0_method": f0_method, "f0_floor_est": f0_floor_est,
           "frame_period": frame_period, "q1": q1, "threshold": threshold}
    if f0_method == "harvest":
        tp, f0 = R.harvest(x, fs, f0_floor=f0_floor_est, frame_period=frame_period)
    else:
        tp, f0raw = R.dio(x, fs, f0_floor=f0_floor_est, frame_period=frame_period)
        out["f0_dio"] = f0raw
        f0 = R.stonemask(x, fs, tp, f0raw)
    fft_size = R.cheaptrick_fft_size(fs, 71.0)
    sp = R.cheaptrick(x, fs, tp, f0, q1=q1, f0_floor=71.0, fft_size=fft_size)
    ap = R.d4c(x, fs, tp, f0, fft_size, threshold=threshold)
    rows = np.arange(0, len(f0), row_step)
    out.update(tp=tp, f0=f0, fft_size=fft_size, rows=rows, sp_rows=sp[rows], ap_rows=ap[rows],
               sum_log_sp=np.log(sp).sum(), sum_ap=ap.sum(),
               sp_row_sums=np.log(sp).sum(axis=1), ap_row_sums=ap.sum(axis=1))
    return out


def codec_fixture(R):
    """Reference codec (src/codec.cpp) outputs for rows of the analysis fixtures above."""
    out = {}
    for name in ("vaiueo2d_harvest", "vowel48k_harvest", "vowel16k_dio"):
        g = dict(np.load(os.path.join(OUT, name + ".npz")))
        fs, fft = int(g["fs"]), int(g["fft_size"])
        sp, ap = g["sp_rows"][:12], g["ap_rows"][:12]
        for nd in (24, 60):
            coded = R.code_spectral_envelope(sp, fs, fft, nd)
            out[f"{name}.mcep{nd}"] = coded
            out[f"{name}.sp_from_mcep{nd}"] = R.decode_spectral_envelope(coded, fs, fft)[:4]
        bap = R.code_aperiodicity(ap, fs, fft)
        bap_in = bap.copy()
        bap_in[::5] = -0.2                         # frames CheckVUV treats as aperiodic (codec.cpp:31-41)
        out[f"{name}.bap"] = bap
        out[f"{name}.bap_in"] = bap_in
        out[f"{name}.ap_from_bap"] = R.decode_aperiodicity(bap_in, fs, fft)
    np.savez_compressed(os.path.join(OUT, "codec.npz"), **out)


SYNTH_CASES = {"syn16k": (16000, 180, 1024, 5.0, 14400, 0), "syn48k": (48000, 120, 2048, 5.0, 28000, 1),
               "syn22k_hop10": (22050, 90, 1024, 10.0, 19000, 2),
               "syn192k": (192000, 44, 8192, 5.0, 40000, 4)}             # fft_size 8192: the default above 96 kHz (round 5)


def synthesis_fixture(R):
    """Reference Synthesis() (src/synthesis.cpp) on the deterministic parameters of tests/util.synth_params."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import synth_params
    out = {}
    for name, (fs, nf, fft, fp, ylen, seed) in SYNTH_CASES.items():
        f0, sp, ap = synth_params(fs, nf, fft, seed)
        out[name] = R.synthesis(f0, sp, ap, fft, fp, fs, ylen)
    np.savez_compressed(os.path.join(OUT, "synthesis.npz"), **out)


def _wav_bytes(fs, nbit, payload, claim=None, extra=b"", fmt_id=1, channels=1):
    """A WAV file image assembled by hand (the reference only WRITES 16 bit)."""
    import struct
    qb = nbit // 8
    head = b"RIFF" + struct.pack("<I", 36 + len(extra) + len(payload)) + b"WAVEfmt " + \
        struct.pack("<IHHIIHH", 16, fmt_id, channels, fs, fs * qb, qb, nbit)
    return head + extra + b"data" + struct.pack("<I", len(payload) if claim is None else claim) + payload


def fileio_fixture():
    """Audio / parameter files through the reference's own tools/ library,
    oracle/_ref/libworld_tools_ref.so: what it decodes from hand-assembled WAV images and the
    bytes it writes for given arrays."""
    import tempfile
    from world_amd.api import FileAPI
    T = FileAPI(os.path.join(ROOT, "oracle", "_ref", "libworld_tools_ref.so"), hip_runtime=False)
    rng = np.random.default_rng(20240607)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        def through_reference(name, image):
            path = os.path.join(d, name + ".wav")
            open(path, "wb").write(image)
            out[name + "_bytes"] = np.frombuffer(image, dtype=np.uint8)
            out[name + "_length"] = np.int64(T.audio_length(path))
            r = T.wavread(path)
            if r is not None:
                out[name + "_x"], out[name + "_fs"], out[name + "_nbit"] = r[0], np.int64(r[1]), np.int64(r[2])
        for nbit in (8, 16, 24, 32):
            qb = nbit // 8
            pcm = rng.integers(0, 256, size=(200, qb), dtype=np.uint8)
            pcm[0] = 0; pcm[1] = 255; pcm[2] = [0] * (qb - 1) + [128]; pcm[3] = [255] * (qb - 1) + [127]
            through_reference(f"wav{nbit}", _wav_bytes(22050, nbit, pcm.tobytes()))
        pcm = rng.integers(0, 256, size=(120, 2), dtype=np.uint8)
        through_reference("wavlist", _wav_bytes(16000, 16, pcm.tobytes(), extra=b"LIST\x07\x00\x00\x00dat" + b"xdada\x00d"))
        through_reference("wavtrunc", _wav_bytes(8000, 24, pcm.tobytes()[:77], claim=120))
        through_reference("bad_stereo", _wav_bytes(8000, 16, pcm.tobytes(), channels=2))
        through_reference("bad_float", _wav_bytes(8000, 32, pcm.tobytes(), fmt_id=3))
        through_reference("bad_nodata", _wav_bytes(8000, 16, b"")[:36] + b"junkjunkjunk")
        through_reference("bad_riff", b"RIFX" + _wav_bytes(8000, 16, pcm.tobytes())[4:])

        x = np.concatenate([rng.uniform(-1.2, 1.2, 300), [0.0, -0.0, 1.0, -1.0, 1.0000153, -1.0000306, 0.99998, 3e-5, -3e-5,
                                                            1e5, -1e5, 7e4, -7e4, 1e300, -1e300, np.nan, np.inf, -np.inf]])
        path = os.path.join(d, "w.wav")
        T.wavwrite(path, x, 44100)
        out["ww_x"], out["ww_bytes"] = x, np.frombuffer(open(path, "rb").read(), dtype=np.uint8)

        nf = 7
        tpos = np.arange(nf) * 0.005
        f0 = np.array([0.0, 101.25, 99.999999, 440.0 / 3.0, 0.0, 71.5, 800.0])
        for text in (0, 1):
            T.write_f0(path, 5.0, tpos, f0, text=bool(text))
            out[f"f0_file{text}"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        T.write_f0(path, 2.5, tpos, f0)
        out["f0_tpos_2p5"] = T.read_f0(path)[0]
        out["f0_tpos"], out["f0_values"] = tpos, f0
        sp = rng.uniform(1e-9, 2.0, (nf, 9))
        T.write_spectral_envelope(path, sp, 16000, 5.0, 16, 0)
        out["sp"], out["sp_file"] = sp, np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        out["sp_header"] = np.array([T.header(path, k) for k in ("NOF ", "FP  ", "FFT ", "NOD ", "FS  ", "XYZ ")])
        coded = rng.normal(size=(nf, 3))
        T.write_aperiodicity(path, coded, 48000, 2.5, 2048, 3)
        out["ap_coded"], out["ap_file"] = coded, np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        out["ap_read"] = T.read_aperiodicity(path)
    np.savez_compressed(os.path.join(OUT, "fileio.npz"), **out)


def high_rate_fixture(R):
    """192 kHz synthetic vowel, 0.3 s (round 5: CheapTrick fft 8192, D4C's 16384-point transforms, Harvest at its largest
    decimation ratio, 12)"""
    x = synth.vowel(192000, 0.3, seed=192, base_f0=150.0).numpy()
    q = np.round(x * 32768.0).astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "vowel192k_harvest.npz"), **analyse(R, q, 192000, "harvest", 71.0, 8))
    print("vowel192k_harvest.npz", os.path.getsize(os.path.join(OUT, "vowel192k_harvest.npz")) // 1024, "KiB")


def main():
    build()
    R = RefOracle()
    if "--synthesis-only" in sys.argv:
        synthesis_fixture(R)
        print("synthesis.npz", os.path.getsize(os.path.join(OUT, "synthesis.npz")) // 1024, "KiB")
        return
    if "--fileio-only" in sys.argv:
        fileio_fixture()
        print("fileio.npz", os.path.getsize(os.path.join(OUT, "fileio.npz")) // 1024, "KiB")
        return
    if "--high-rate-only" in sys.argv:
        high_rate_fixture(R)
        return
    if "--codec-only" in sys.argv:
        codec_fixture(R)
        print("codec.npz", os.path.getsize(os.path.join(OUT, "codec.npz")) // 1024, "KiB")
        return
    q, fs = wav_int16("/root/reference/test/vaiueo2d.wav")
    # config 0 plumbing of test/test.cpp:89-219 (DIO floor 40 + StoneMask) and its Harvest variant
    np.savez_compressed(os.path.join(OUT, "vaiueo2d_dio.npz"), **analyse(R, q, fs, "dio", 40.0, 4))
    np.savez_compressed(os.path.join(OUT, "vaiueo2d_harvest.npz"), **analyse(R, q, fs, "harvest", 40.0, 2))
    # 48 kHz synthetic vowel, 1.2 s (north-star shapes: fft 2048, D4C 4096, ratio 6)
    x = synth.vowel(48000, 1.2, seed=12345).numpy()
    q48 = np.round(x * 32768.0).astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "vowel48k_harvest.npz"), **analyse(R, q48, 48000, "harvest", 71.0, 8))
    # 16 kHz alt config (config 4 of BASELINE.json): DIO defaults + StoneMask, fft 1024
    x = synth.vowel(16000, 1.5, seed=7, base_f0=180.0).numpy()
    q16 = np.round(x * 32768.0).astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "vowel16k_dio.npz"), **analyse(R, q16, 16000, "dio", 71.0, 6))
    high_rate_fixture(R)
    # primitive known answers
    import ctypes as C
    L = R.lib
    np.savez_compressed(
        os.path.join(OUT, "primitives.npz"),
        randn5=np.array([-1.3276404961943626, -0.62285530939698219, -1.6091805659234524,
                         1.1797650642693043, -0.25188251212239265]),
        interp_x=np.array([1., 2., 4., 7.]), interp_xi=np.array([-1, .5, 1, 1.5, 2, 3.9, 4, 7, 9.]),
        interp_yi=np.array([-10, 5, 10, 15, 20, 39, 40, 70, 90.]))
    codec_fixture(R)
    synthesis_fixture(R)
    fileio_fixture()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
