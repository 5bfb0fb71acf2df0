"""Audio and parameter files (reference tools/audioio.cpp, tools/parameterio.cpp).

Golden vectors: tests/golden/fileio.npz, made by the unmodified reference tools
(tests/golden/make_golden.py::fileio_fixture).  Everything here is bit-exact: the formats are
fixed layouts and the PCM <-> double conversions are exact in FP64.

 * not gpu: the format oracle (oracle/fileformats.py) against the fixtures; the library's
   host-side file functions (headers, F0 / SPEC / AP files: no device work) against the same
   bytes; wavread()/wavwrite() through the host-emulated build of the same sources.
 * gpu: wavread()/wavwrite() and the device-resident decode through libworld_hip.so, and the
   reference's own examples/parameter_io programs linked against libworld_hip.so alone.
"""
import os
import subprocess

import numpy as np
import pytest

from util import GOLDEN, RTOL, max_rel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
WAVS = ["wav8", "wav16", "wav24", "wav32", "wavlist", "wavtrunc"]
BAD = ["bad_stereo", "bad_float", "bad_nodata", "bad_riff"]


@pytest.fixture(scope="module")
def fx():
    return dict(np.load(os.path.join(GOLDEN, "fileio.npz")))


def put(tmp_path, name, data):
    p = os.path.join(str(tmp_path), name)
    with open(p, "wb") as f:
        f.write(bytes(data))
    return p


def slurp(path):
    with open(path, "rb") as f:
        return f.read()


# ---- the format oracle, pinned on the reference's files ---------------------------------------
def test_format_oracle_matches_the_reference_fixtures(fx):
    from oracle import fileformats as ff
    for name in WAVS:
        image = bytes(fx[name + "_bytes"])
        fs, nbit, x = ff.wav_decode(image)
        assert (fs, nbit, len(x)) == (fx[name + "_fs"], fx[name + "_nbit"], fx[name + "_length"])
        assert np.array_equal(x, fx[name + "_x"]), name
    for name in BAD:
        assert ff.wav_layout(bytes(fx[name + "_bytes"])) is None and fx[name + "_length"] == -1
    assert ff.wav_encode(fx["ww_x"], 44100) == bytes(fx["ww_bytes"])
    assert ff.f0_encode(5.0, fx["f0_tpos"], fx["f0_values"]) == bytes(fx["f0_file0"])
    assert ff.f0_encode(5.0, fx["f0_tpos"], fx["f0_values"], text=True) == bytes(fx["f0_file1"])
    tpos, f0, fp = ff.f0_decode(bytes(fx["f0_file0"]))
    assert np.array_equal(f0, fx["f0_values"]) and fp == 5.0
    assert np.array_equal(ff.f0_decode(ff.f0_encode(2.5, tpos, f0))[0], fx["f0_tpos_2p5"])
    assert ff.matrix_encode(b"SPEC", 16000, 5.0, 16, 0, fx["sp"]) == bytes(fx["sp_file"])
    assert ff.matrix_encode(b"AP  ", 48000, 2.5, 2048, 3, fx["ap_coded"]) == bytes(fx["ap_file"])
    assert np.array_equal(ff.matrix_decode(b"SPEC", bytes(fx["sp_file"])), fx["sp"])
    assert np.array_equal(ff.matrix_decode(b"AP  ", bytes(fx["ap_file"])), fx["ap_read"])
    assert ff.matrix_decode(b"AP  ", bytes(fx["sp_file"])) is None
    keys = (b"NOF ", b"FP  ", b"FFT ", b"NOD ", b"FS  ", b"XYZ ")
    assert [ff.header_information(bytes(fx["sp_file"]), k) for k in keys] == list(fx["sp_header"])


# ---- the library's host-side file functions (no device work: run anywhere) ---------------------
def check_parameter_files(F, fx, tmp_path):
    p = os.path.join(str(tmp_path), "out.bin")
    F.write_f0(p, 5.0, fx["f0_tpos"], fx["f0_values"])
    assert slurp(p) == bytes(fx["f0_file0"])
    tpos, f0 = F.read_f0(p)
    assert np.array_equal(f0, fx["f0_values"]) and np.array_equal(tpos, np.arange(7) / 1000.0 * 5.0)
    assert F.header(p, "NOF ") == 7.0 and F.header(p, "FP  ") == 5.0 and F.header(p, "FS  ") == 0.0
    F.write_f0(p, 5.0, fx["f0_tpos"], fx["f0_values"], text=True)
    assert slurp(p) == bytes(fx["f0_file1"])
    F.write_f0(p, 2.5, fx["f0_tpos"], fx["f0_values"])
    assert np.array_equal(F.read_f0(p)[0], fx["f0_tpos_2p5"])
    F.write_spectral_envelope(p, fx["sp"], 16000, 5.0, 16, 0)
    assert slurp(p) == bytes(fx["sp_file"])
    assert [F.header(p, k) for k in ("NOF ", "FP  ", "FFT ", "NOD ", "FS  ", "XYZ ")] == list(fx["sp_header"])
    assert np.array_equal(F.read_spectral_envelope(p), fx["sp"])
    assert F.read_aperiodicity(p) is None and F.read_f0(p) is None          # wrong tag: "Header error."
    F.write_aperiodicity(p, fx["ap_coded"], 48000, 2.5, 2048, 3)
    assert slurp(p) == bytes(fx["ap_file"])
    assert np.array_equal(F.read_aperiodicity(p), fx["ap_read"])
    # a file that ends early leaves the remaining rows as the caller had them
    q = put(tmp_path, "short.sp", bytes(fx["sp_file"])[:48 + 72 * 3 + 16])
    m = F.read_spectral_envelope(q)
    assert np.array_equal(m[:3], fx["sp"][:3]) and np.all(m[4:] == 0.0)
    assert F.read_f0(os.path.join(str(tmp_path), "absent.f0")) is None
    for name in WAVS:
        assert F.audio_length(put(tmp_path, name + ".wav", fx[name + "_bytes"])) == fx[name + "_length"]
    for name in BAD:
        assert F.audio_length(put(tmp_path, name + ".wav", fx[name + "_bytes"])) == -1
    assert F.audio_length(os.path.join(str(tmp_path), "absent.wav")) == 0


def check_audio_files(F, fx, tmp_path):
    for name in WAVS:
        x, fs, nbit = F.wavread(put(tmp_path, name + ".wav", fx[name + "_bytes"]))
        assert (fs, nbit) == (fx[name + "_fs"], fx[name + "_nbit"])
        assert np.array_equal(x, fx[name + "_x"]), name
    for name in BAD:
        assert F.wavread(put(tmp_path, name + ".wav", fx[name + "_bytes"])) is None
    # a "data" tag far into the file (beyond the reader's first window), decoys before it
    from oracle import fileformats as ff
    rng = np.random.default_rng(3)
    junk = bytes(rng.integers(0, 256, 150001, dtype=np.uint8)).replace(b"data", b"dat_") + b"dat"
    image = bytes(fx["wav16_bytes"][:36]) + b"LIST" + len(junk).to_bytes(4, "little") + junk + bytes(fx["wav16_bytes"][36:])
    q = put(tmp_path, "late.wav", image)
    assert F.audio_length(q) == 200
    x, fs, nbit = F.wavread(q)
    assert np.array_equal(x, ff.wav_decode(image)[2]) and np.array_equal(x, fx["wav16_x"])
    p = os.path.join(str(tmp_path), "w.wav")
    F.wavwrite(p, fx["ww_x"], 44100)
    assert slurp(p) == bytes(fx["ww_bytes"])
    F.wavwrite(p, np.zeros(0), 8000)
    assert len(slurp(p)) == 44 and F.audio_length(p) == 0
    # what wavwrite stores, wavread returns: q / 32768 for the quantised samples
    x = np.linspace(-1.0, 1.0, 4001)
    F.wavwrite(p, x, 48000)
    y, fs, nbit = F.wavread(p)
    assert (fs, nbit) == (48000, 16) and np.array_equal(y, np.trunc(x * 32767.0) / 32768.0)


def test_parameter_files_of_the_library_match_the_reference_bytes(fx, tmp_path):
    from world_amd.api import FileAPI
    check_parameter_files(FileAPI(), fx, tmp_path)


def test_reference_tools_agree_with_their_own_fixtures(fx, tmp_path):
    """The same checks on the in-place build of the reference: the checks themselves are sound."""
    from world_amd.api import FileAPI
    lib = os.path.join(REF_DIR, "libworld_tools_ref.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref not built")
    R = FileAPI(lib, hip_runtime=False)
    check_parameter_files(R, fx, tmp_path)
    check_audio_files(R, fx, tmp_path)


def test_emulated_audio_files(fx, tmp_path):
    from world_amd.api import FileAPI
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", emu_dir], check=True)
    E = FileAPI(os.path.join(emu_dir, "libworld_emu.so"), hip_runtime=False)
    check_audio_files(E, fx, tmp_path)
    check_parameter_files(E, fx, tmp_path)


# ---- on the GPU -------------------------------------------------------------------------------
@pytest.mark.gpu
def test_audio_files_on_the_gpu(fx, tmp_path):
    from world_amd.api import FileAPI
    F = FileAPI()
    check_audio_files(F, fx, tmp_path)
    check_parameter_files(F, fx, tmp_path)


@pytest.mark.gpu
def test_device_resident_wav_decode_and_encode(fx, tmp_path):
    import torch
    from oracle import fileformats as ff
    from world_amd.api import WorldHip
    wh = WorldHip()
    rng = np.random.default_rng(5)
    for nbit in (8, 16, 24, 32):
        raw = rng.integers(0, 256, size=300001 * (nbit // 8), dtype=np.uint8)
        image = bytes(fx[f"wav{nbit}_bytes"][:40]) + (len(raw)).to_bytes(4, "little") + raw.tobytes()
        p = put(tmp_path, "big.wav", image)
        x, fs = wh.wavread(p)
        assert fs == 22050 and x.is_cuda and np.array_equal(x.cpu().numpy(), ff.wav_decode(image)[2])
    x16 = torch.from_numpy(rng.integers(-32768, 32768, size=4097, dtype=np.int16)).cuda()
    assert torch.equal(wh.pcm16_to_double(x16), wh.pcm_to_double(x16.view(torch.uint8), 16))
    x = torch.from_numpy(np.concatenate([rng.uniform(-1.5, 1.5, 100000), fx["ww_x"]])).cuda()
    q = wh.double_to_pcm16(x).cpu().numpy()
    assert q.tobytes() == ff.wav_encode(x.cpu().numpy(), 8000)[44:]
    with pytest.raises(OSError):
        wh.wavread(put(tmp_path, "bad.wav", fx["bad_stereo_bytes"]))
    with pytest.raises(RuntimeError):
        wh.pcm_to_double(torch.zeros(8, dtype=torch.uint8).cuda(), 64)


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


@pytest.mark.gpu
def test_reference_example_programs_linked_against_the_library_alone(tmp_path):
    """examples/parameter_io/{f0,sp,ap}analysis.cpp and readandsynthesis.cpp, unmodified, built
    twice by oracle/Makefile: *_ref on the reference's objects, *_hip on libworld_hip.so and
    nothing else.  Same command lines, same files out (headers byte-identical, values to RTOL)."""
    from oracle import fileformats as ff
    from world_amd import synth
    from world_amd.api import FileAPI
    if not os.path.exists(os.path.join(REF_DIR, "f0analysis_hip")):
        pytest.skip("oracle/_ref example programs not built")
    d = str(tmp_path)
    x = synth.vowel(22050, 1.1, seed=3, base_f0=150.0).numpy()
    FileAPI().wavwrite(os.path.join(d, "in.wav"), x, 22050)
    for tag in ("ref", "hip"):
        exe = lambda n: os.path.join(REF_DIR, f"{n}_{tag}")
        run([exe("f0analysis"), "in.wav", "-f", "60", "-o", f"{tag}.f0"], d)
        run([exe("f0analysis"), "in.wav", "-f", "60", "-t", "-o", f"{tag}.txt"], d)
        run([exe("spanalysis"), "in.wav", f"{tag}.f0", "-o", f"{tag}.sp"], d)
        run([exe("apanalysis"), "in.wav", f"{tag}.f0", "-o", f"{tag}.ap"], d)
        run([exe("readandsynthesis"), f"{tag}.f0", f"{tag}.sp", f"{tag}.ap", "-o", f"{tag}.wav"], d)
    ref = {e: slurp(os.path.join(d, "ref." + e)) for e in ("f0", "txt", "sp", "ap", "wav")}
    hip = {e: slurp(os.path.join(d, "hip." + e)) for e in ("f0", "txt", "sp", "ap", "wav")}
    assert hip["f0"][:24] == ref["f0"][:24] and len(hip["f0"]) == len(ref["f0"])
    f_ref, f_hip = ff.f0_decode(ref["f0"])[1], ff.f0_decode(hip["f0"])[1]
    assert np.array_equal(f_ref > 0, f_hip > 0) and np.sum(f_ref > 0) > 100
    assert max_rel(f_hip, f_ref) <= RTOL
    assert hip["txt"] == ref["txt"]                                  # "%.5f %.5f": the printed digits agree
    for e, tag in (("sp", b"SPEC"), ("ap", b"AP  ")):
        assert hip[e][:48] == ref[e][:48] and len(hip[e]) == len(ref[e])
        assert max_rel(ff.matrix_decode(tag, hip[e]), ff.matrix_decode(tag, ref[e])) <= RTOL
    assert hip["wav"][:44] == ref["wav"][:44] and len(hip["wav"]) == len(ref["wav"])
    q_ref = np.frombuffer(ref["wav"][44:], dtype="<i2").astype(int)
    q_hip = np.frombuffer(hip["wav"][44:], dtype="<i2").astype(int)
    assert np.max(np.abs(q_ref)) > 3000
    assert np.max(np.abs(q_hip - q_ref)) <= 1 and np.mean(q_hip != q_ref) < 1e-3   # truncation flips on 1e-9 differences


@pytest.mark.gpu
def test_batch_tools_write_what_the_reference_programs_write(tmp_path):
    """python -m world_amd.tools (many files per GPU call) against the reference's one-file
    programs built from the unmodified reference (oracle/_ref/*_ref)."""
    import sys
    from oracle import fileformats as ff
    from world_amd import synth
    from world_amd.api import FileAPI
    if not os.path.exists(os.path.join(REF_DIR, "f0analysis_ref")):
        pytest.skip("oracle/_ref example programs not built")
    d = str(tmp_path)
    F = FileAPI()
    specs = [("a", 16000, 0.9, 120.0), ("b", 16000, 0.55, 210.0), ("c", 44100, 0.7, 95.0)]
    for name, fs, dur, f0 in specs:
        F.wavwrite(os.path.join(d, name + ".wav"), synth.vowel(fs, dur, seed=len(name) + fs, base_f0=f0).numpy(), fs)
    env = dict(os.environ, PYTHONPATH=ROOT)
    tool = [sys.executable, "-m", "world_amd.tools"]
    subprocess.run(tool + ["analysis", "a.wav", "b.wav", "c.wav", "--outdir", "hip"], cwd=d, env=env, check=True, timeout=600)
    subprocess.run(tool + ["analysis", "a.wav", "c.wav", "--outdir", "coded", "--code-sp", "40", "--code-ap"], cwd=d, env=env,
                   check=True, timeout=600)
    os.makedirs(os.path.join(d, "ref"))
    for name, fs, _, _ in specs:
        exe = lambda n: os.path.join(REF_DIR, n + "_ref")
        run([exe("f0analysis"), name + ".wav", "-o", f"ref/{name}.f0"], d)
        run([exe("spanalysis"), name + ".wav", f"ref/{name}.f0", "-o", f"ref/{name}.sp"], d)
        run([exe("apanalysis"), name + ".wav", f"ref/{name}.f0", "-o", f"ref/{name}.ap"], d)
        ref = {e: slurp(os.path.join(d, "ref", f"{name}.{e}")) for e in ("f0", "sp", "ap")}
        hip = {e: slurp(os.path.join(d, "hip", f"{name}.{e}")) for e in ("f0", "sp", "ap")}
        assert hip["f0"][:24] == ref["f0"][:24] and len(hip["f0"]) == len(ref["f0"])
        assert max_rel(ff.f0_decode(hip["f0"])[1], ff.f0_decode(ref["f0"])[1]) <= RTOL
        for e, tag in (("sp", b"SPEC"), ("ap", b"AP  ")):
            # the envelopes follow each side's own F0 (equal to ~1e-9), hence RTOL and not bit equality
            assert hip[e][:48] == ref[e][:48] and len(hip[e]) == len(ref[e])
            assert max_rel(ff.matrix_decode(tag, hip[e]), ff.matrix_decode(tag, ref[e])) <= RTOL
    # coded files: NOD in the header, rows of NOD values, and they decode back close to the dense ones
    sp40 = slurp(os.path.join(d, "coded", "a.sp"))
    assert ff.header_information(sp40, b"NOD ") == 40.0 and ff.matrix_decode(b"SPEC", sp40).shape[1] == 40
    assert ff.matrix_decode(b"AP  ", slurp(os.path.join(d, "coded", "c.ap"))).shape[1] == 5
    # synthesis from the REFERENCE's parameter files, both ways
    run([os.path.join(REF_DIR, "readandsynthesis_ref"), "ref/c.f0", "ref/c.sp", "ref/c.ap", "-o", "ref/c.wav"], d)
    subprocess.run(tool + ["synthesis", "ref/c.f0", "ref/c.sp", "ref/c.ap", "-o", "hip/c.wav"], cwd=d, env=env, check=True,
                   timeout=600)
    w_ref, w_hip = slurp(os.path.join(d, "ref", "c.wav")), slurp(os.path.join(d, "hip", "c.wav"))
    assert w_hip[:44] == w_ref[:44] and len(w_hip) == len(w_ref)
    q_ref = np.frombuffer(w_ref[44:], dtype="<i2").astype(int)
    q_hip = np.frombuffer(w_hip[44:], dtype="<i2").astype(int)
    assert np.max(np.abs(q_ref)) > 3000 and np.max(np.abs(q_hip - q_ref)) <= 1 and np.mean(q_hip != q_ref) < 1e-3
    # and from the coded files: runs, same length, still the same utterance
    subprocess.run(tool + ["synthesis", "hip/a.f0", "coded/a.sp", "coded/a.ap", "-o", "coded/a.wav"], cwd=d, env=env,
                   check=True, timeout=600)
    assert F.audio_length(os.path.join(d, "coded", "a.wav")) == int(181 * 5.0 / 1000.0 * 16000)


def test_randomised_files_against_the_reference_tools(tmp_path):
    """Random parameter arrays and random WAV images (any of 8/16/24/32 bits, junk chunks of any size before
    `data`, claimed lengths beyond the file's end) through the library's file functions (the PCM decode through
    the host-emulated kernels) and through the unmodified reference tools: same bytes, same samples."""
    from world_amd.api import FileAPI
    lib = os.path.join(REF_DIR, "libworld_tools_ref.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref not built")
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", emu_dir], check=True)
    R, E, F = FileAPI(lib, hip_runtime=False), FileAPI(os.path.join(emu_dir, "libworld_emu.so"), hip_runtime=False), FileAPI()
    rng = np.random.default_rng(99)
    d = str(tmp_path)
    for case in range(120):
        nf, nb = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        f0 = rng.uniform(-5.0, 900.0, nf)
        f0[rng.random(nf) < 0.1] = rng.choice([np.nan, np.inf, -np.inf, 0.0, -0.0])
        tpos, fp = rng.uniform(0.0, 10.0, nf), float(rng.choice([1.0, 2.5, 5.0, 12.34]))
        m = rng.normal(size=(nf, nb)) * 10.0 ** rng.integers(-300, 300)
        fft, nod, fs = 2 * (nb - 1), int(rng.choice([0, nb])), int(rng.integers(1, 200000))
        for L, tag in ((R, "r"), (F, "o")):
            L.write_f0(f"{d}/{tag}.f0", fp, tpos, f0, text=bool(case % 2))
            L.write_spectral_envelope(f"{d}/{tag}.sp", m, fs, fp, max(fft, 2), nod if fft >= 2 else nb)
            L.write_aperiodicity(f"{d}/{tag}.ap", m, fs, fp, max(fft, 2), nod if fft >= 2 else nb)
        for ext in ("f0", "sp", "ap"):
            assert slurp(f"{d}/r.{ext}") == slurp(f"{d}/o.{ext}"), (case, ext)
        if case % 2 == 0:
            a, b = R.read_f0(f"{d}/r.f0"), F.read_f0(f"{d}/r.f0")
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True)
        assert np.array_equal(R.read_spectral_envelope(f"{d}/r.sp"), F.read_spectral_envelope(f"{d}/r.sp"))
        keys = ("NOF ", "FP  ", "FFT ", "NOD ", "FS  ")
        assert [R.header(f"{d}/r.ap", k) for k in keys] == [F.header(f"{d}/r.ap", k) for k in keys]
        # a WAV image
        nbit = int(rng.choice([8, 16, 24, 32]))
        qb, n = nbit // 8, int(rng.integers(0, 300))
        payload = bytes(rng.integers(0, 256, n * qb, dtype=np.uint8))
        junk = bytes(rng.integers(0, 256, int(rng.choice([0, 5, 26, 70000])), dtype=np.uint8)).replace(b"data", b"dat_")
        extra = (b"LIST" + len(junk).to_bytes(4, "little") + junk) if junk else b""
        claim = len(payload) + int(rng.choice([0, 0, qb, 7 * qb + 1]))
        image = (b"RIFF" + (36 + len(extra) + len(payload)).to_bytes(4, "little") + b"WAVEfmt " + (16).to_bytes(4, "little") +
                 (1).to_bytes(2, "little") + (1).to_bytes(2, "little") + fs.to_bytes(4, "little") + (fs * qb % 2**32).to_bytes(4, "little") +
                 qb.to_bytes(2, "little") + nbit.to_bytes(2, "little") + extra + b"data" + claim.to_bytes(4, "little") + payload)
        p = put(tmp_path, "c.wav", image)
        want_n = R.audio_length(p)
        assert F.audio_length(p) == want_n == E.audio_length(p), case
        if want_n > 0 and len(payload) >= qb:       # (with no whole sample in the file the reference decodes stack garbage)
            a, b = R.wavread(p), E.wavread(p)
            assert a[1:] == b[1:] and np.array_equal(a[0], b[0]), (case, nbit, n, claim)
        x = rng.uniform(-1.3, 1.3, int(rng.integers(0, 500)))
        R.wavwrite(f"{d}/r.wav", x, fs % 2**31)
        E.wavwrite(f"{d}/e.wav", x, fs % 2**31)
        assert slurp(f"{d}/r.wav") == slurp(f"{d}/e.wav")
