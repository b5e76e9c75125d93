// bc_i8_quant.h -- quantisation of the int8 pre-filter's sweep vector(s), shared by the two places that can do it:
//   * the single-block step kernels (bc_snnls.hip: dev_prep), which PRODUCE the vectors of the next sweep and now also
//     leave their digits behind -- once per step, 1.3 KB;
//   * the prologue of k_sweep_i8 (bc_prefilter_i8.h), where every block of every sweep used to redo the job (two block
//     reductions, two barriers, a few divisions before the first dot4): still the path of standalone argmax sweeps, whose
//     vector does not come out of a step kernel.
// Same arithmetic in both (so the bounds, and with them the candidate counts, do not depend on who quantised).
//
// Record layout (ints): [SP4][4] packed digits of k-group g = {v0 digit 0, v0 digit 1, v1 digit, 0}, then BC_I8Q_HDR floats:
//   fvs0, fvs1 (steps), fev0, fev1 (||v^ - v|| bounds, rounded up), fvn (||v||, rounded up), vbad (1.0: NaN / inf in v)
#pragma once

#define BC_I8Q_HDR 8
#define BC_I8Q_INTS(sp4) (4 * (sp4) + BC_I8Q_HDR)

// running max of |x| in which a NaN element counts as +inf (fmax alone would drop it): either makes the vector "bad"
__device__ __forceinline__ double bc_i8q_absmax(double m, double x) { return (x != x) ? INFINITY : fmax(m, fabs(x)); }

struct bc_i8q_scalars {
  double vstep0, vstep1;
  double inv0, inv1;      // reciprocals of the steps (0 for a zero step): one division per vector instead of one per element
  bool vbad;
};

// (Round 5: ONE fp64 division per vector -- r = 1 / max|v| -- instead of two (step = max / 16256, then 1 / step): a division is a
// ~500-cycle dependent sequence and the single-block step kernels paid eight of them per step for the two digit records
// (tools/fin_stamps.py).  step = max * fl(1 / 16256) and inv = 16256 * r: inv * step = 1 to within 4e-16, and the digit rint(val * inv)
// is still the nearest integer of a value within 1e-12 of val / step -- the bound below only needs that.)
__device__ __forceinline__ bc_i8q_scalars bc_i8q_steps(double vmax0, double vmax1) {
  bc_i8q_scalars q;
  // a NaN / inf in v makes every score NaN in the fp64 kernel: hand all rows over (delta = inf in the sweep)
  q.vbad = !(vmax0 < INFINITY) || !(vmax1 < INFINITY) || vmax0 != vmax0 || vmax1 != vmax1;
  q.vstep0 = vmax0 * (1. / 16256.);   // v0: 14 bits + sign in two digits
  q.vstep1 = vmax1 * (1. / 127.);     // v1: one digit
  // digits are rint(val * inv): |Q step - val| <= step / 2 * (1 + 1e-11), far inside the 1.00001 the sweep's error bound is
  // inflated by
  const double r0 = (vmax0 > 0. && !q.vbad) ? 1. / vmax0 : 0., r1 = (vmax1 > 0. && !q.vbad) ? 1. / vmax1 : 0.;
  q.inv0 = (q.vstep0 > 0.) ? 16256. * r0 : 0.;      // (a denormal max whose step underflows to 0: no digits, as before)
  q.inv1 = (q.vstep1 > 0.) ? 127. * r1 : 0.;
  return q;
}

// the digits of ONE element: (d0, d1) of the score vector's value, e of the second vector's (GIGA)
__device__ __forceinline__ void bc_i8q_elem0(double val, const bc_i8q_scalars& q, int& d0, int& d1) {
  int Q = (int)rint(val * q.inv0);
  Q = Q > 16256 ? 16256 : (Q < -16256 ? -16256 : Q);
  d0 = (int)rint((double)Q * 0.0078125);       // Q / 128, exact
  d1 = Q - 128 * d0;
}
__device__ __forceinline__ int bc_i8q_elem1(double val, const bc_i8q_scalars& q) {
  const int Q = (int)rint(val * q.inv1);
  return Q > 127 ? 127 : (Q < -127 ? -127 : Q);
}

// the four packed words of k-group g (samples 4g .. 4g+3)
template <int MODE>
__device__ __forceinline__ void bc_i8q_group(const double* __restrict__ v, int S, int g, const bc_i8q_scalars& q, unsigned (&w)[4]) {
  w[0] = w[1] = w[2] = w[3] = 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 4 * g + j;
    if (k < S && !q.vbad) {
      int d0, d1;
      bc_i8q_elem0((MODE == 0) ? v[2 * k] : v[k], q, d0, d1);
      w[0] |= ((unsigned)d0 & 0xffu) << (8 * j);
      w[1] |= ((unsigned)d1 & 0xffu) << (8 * j);
      if (MODE == 0) w[2] |= ((unsigned)bc_i8q_elem1(v[2 * k + 1], q) & 0xffu) << (8 * j);
    }
  }
}

struct bc_i8q_hdr {
  float fvs0, fvs1, fev0, fev1, fvn;
  bool vbad;
};

// fp32 copies, each rounded UP where it enters a bound
__device__ __forceinline__ bc_i8q_hdr bc_i8q_header(const bc_i8q_scalars& q, int S, double vn) {
  const double rs = (double)sqrtf((float)S) * (0.5 * (1. + 1e-6));      // >= sqrt(S) / 2 (an fp64 sqrt is another ~500 cycles)
  bc_i8q_hdr h;
  h.fvn = __double2float_ru(vn);
  h.fev0 = __double2float_ru(rs * q.vstep0);
  h.fev1 = __double2float_ru(rs * q.vstep1);
  h.fvs0 = (float)q.vstep0;
  h.fvs1 = (float)q.vstep1;
  h.vbad = q.vbad;
  return h;
}

// One full wave (all 64 lanes active) quantises v into the record `qv`; vn = ||v|| (1 for GIGA's unit vectors).
// m0 / m1: this lane's partial maxima of |v0| / |v1| (bc_i8q_absmax over the elements it produced) -- the wave that writes
// v has every element in a register once, so the maximum costs it one instruction per element and no second pass.
// vsrc: where the digits' pass reads the elements from -- a copy in LDS when the caller has one (a re-read from global
// memory is a round trip to the L2 behind the stores: ~1 us in a single-block kernel), else v itself; either way the
// calling wave wrote it, lane-strided, and has passed a wave-level fence.
template <int MODE>
__device__ __forceinline__ void bc_i8q_wave(const double* __restrict__ vsrc, int S, int SP4, double vn, int* __restrict__ qv, int lane,
                                            double m0, double m1) {
  m0 = bc_wave_max_all(m0);
  m1 = bc_wave_max_all(m1);
  const bc_i8q_scalars q = bc_i8q_steps(m0, m1);
  for (int g = lane; g < SP4; g += 64) {
    unsigned w[4];
    bc_i8q_group<MODE>(vsrc, S, g, q, w);
    reinterpret_cast<int4*>(qv)[g] = make_int4((int)w[0], (int)w[1], (int)w[2], (int)w[3]);
  }
  if (lane == 0) {
    const bc_i8q_hdr h = bc_i8q_header(q, S, vn);
    float* f = reinterpret_cast<float*>(qv + 4 * SP4);
    f[0] = h.fvs0; f[1] = h.fvs1; f[2] = h.fev0; f[3] = h.fev1; f[4] = h.fvn; f[5] = h.vbad ? 1.f : 0.f; f[6] = 0.f; f[7] = 0.f;
  }
}
