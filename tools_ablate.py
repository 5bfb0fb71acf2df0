"""Ablation timing on the GPU box (scratch tool): patch a kernel source, rebuild, time, restore."""
import json, subprocess, sys, shutil
SRC = 'world_amd/csrc/d4c.hip'
VARIANTS = {
  'base': [],
  'gd_nofft': [("    block_cfft_dif(Z, plan_c, tw);\n", "")],
  'gd_nowindow': [("    const int wlen = d4c_windowed(x, x_len, fs, cf0, cpos, kBlackman, 4.0, noise + (size_t)c * wdraws,\n                                  Z, true, scratch);\n",
                   "    const int wlen = wdraws; (void)cpos;\n")],
  'gd_nocombine': [("      cplx za = Z[phys], zb = Z[fft_slot(plan_c, (N - k) & (N - 1))];\n", "      cplx za = {1.0 * phys, 2.0}, zb = {3.0, 1.0 * k};\n")],
  'gd_nonorm': [("    pw = block_sum(pw, scratch);\n", "")],
  'gd_nopower': [("    block_rfft(Z, lgn, tw, [&](int k, double re, double im) { B[k] = re * re + im * im; });\n", "")],
  'gd_nosmooth': [("  d4c_smooth(A, cf0 / 2.0, fs, N, Zr, A, scratch);\n  d4c_smooth(A, cf0, fs, N, Zr, B, scratch);\n", "")],
  'band_nofft': [("  block_rfft(Z, lgn, tw, [&](int k, double re, double im) {\n    (void)k;\n", "  for (int k = tid; k <= H; k += nt) { double re = Zr[k], im = 1.0;\n"),
                 ("    ++filled;\n  });\n", "    ++filled;\n  }\n")],
  'band_noselect': [("  block_smallest_sum(key, filled, H + 1, H - bnd, hist, scratch, &part, &tot);\n", "  part = __longlong_as_double((long long)key[0]); tot = 1.0 + filled;\n")],
}
def run(name):
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '12', '--warmup', '3', '--streams', '1', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    k = d['kernels_ms_per_step']
    print(name, 'ms/step %.3f' % d['ms_per_step'], {n: k[n] for n in ('d4c_groupdelay', 'd4c_band', 'd4c_lovetrain')}, flush=True)
orig = open(SRC).read()
try:
    for name in (sys.argv[1:] or VARIANTS):
        s = orig
        for a, b in VARIANTS[name]:
            assert a in s, (name, a)
            s = s.replace(a, b, 1)
        open(SRC, 'w').write(s)
        r = subprocess.run([sys.executable, '-m', 'world_amd.build'], capture_output=True, text=True)
        if r.returncode: print(name, 'BUILD FAILED', r.stderr[-800:]); continue
        run(name)
finally:
    open(SRC, 'w').write(orig)
