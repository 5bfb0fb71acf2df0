// tables.cpp -- host construction of the twiddle and RNG jump-ahead tables.
#include "tables.h"

#include <vector>

namespace world_hip {

void build_twiddles(double2 *out) {
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int k = 0; k < kTwN; ++k) {
    long double a = two_pi * k / kTwN;
    out[k] = make_double2((double)cosl(a), (double)sinl(a));
  }
  // exact values on the axes
  out[0] = make_double2(1.0, 0.0);
  out[kTwN / 4] = make_double2(0.0, 1.0);
  out[kTwN / 2] = make_double2(-1.0, 0.0);
  out[3 * kTwN / 4] = make_double2(0.0, -1.0);
  // dense quarter-wave tables, one per transform size, copied from the entries above (bitwise the same values)
  double *q = reinterpret_cast<double *>(out + kTwN);
  for (int lg = 2; lg <= kTwLog2; ++lg)
    for (int r = 0; r <= (1 << (lg - 2)); ++r) q[quarter_table_offset(lg) + r] = out[(size_t)r << (kTwLog2 - lg)].x;
  if (kQuarterDoubles & 1) q[kQuarterDoubles] = 0.0;
}

namespace {
struct V128 { uint32_t w[4]; };                 // state bits: x = w[0] ... w = w[3]
struct M128 { V128 col[128]; };                 // column c = image of basis vector e_c

// one xorshift128 step (reference src/matlabfunctions.cpp:246-251)
V128 step(V128 s) {
  uint32_t t = s.w[0] ^ (s.w[0] << 11);
  V128 r;
  r.w[0] = s.w[1]; r.w[1] = s.w[2]; r.w[2] = s.w[3];
  r.w[3] = (s.w[3] ^ (s.w[3] >> 19)) ^ (t ^ (t >> 8));
  return r;
}
V128 apply(const M128 &m, const V128 &v) {
  V128 r = {{0, 0, 0, 0}};
  for (int c = 0; c < 128; ++c)
    if ((v.w[c >> 5] >> (c & 31)) & 1u)
      for (int k = 0; k < 4; ++k) r.w[k] ^= m.col[c].w[k];
  return r;
}
M128 mul(const M128 &a, const M128 &b) {        // a * b  (apply b first)
  M128 r;
  for (int c = 0; c < 128; ++c) r.col[c] = apply(a, b.col[c]);
  return r;
}
}  // namespace

void build_jump_tables(uint4 *out) {
  M128 t;
  for (int c = 0; c < 128; ++c) {
    V128 e = {{0, 0, 0, 0}};
    e.w[c >> 5] = 1u << (c & 31);
    t.col[c] = step(e);                          // the step is linear over GF(2)
  }
  M128 m = t;
  for (int i = 1; i < 12; ++i) m = mul(t, m);    // T^12 : one randn() call
  for (int level = 0; level < kJumpLevels; ++level) {
    uint4 *tab = out + (size_t)level * kJumpStride;
    for (int nib = 0; nib < 32; ++nib)
      for (int v = 0; v < 16; ++v) {
        V128 acc = {{0, 0, 0, 0}};
        for (int b = 0; b < 4; ++b)
          if ((v >> b) & 1)
            for (int k = 0; k < 4; ++k) acc.w[k] ^= m.col[4 * nib + b].w[k];
        tab[nib * 16 + v] = make_uint4(acc.w[0], acc.w[1], acc.w[2], acc.w[3]);
      }
    m = mul(m, m);                               // -> 2^(level+1) calls
  }
}

}  // namespace world_hip
