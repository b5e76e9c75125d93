// bc_i4_quant.h -- the sweep vector(s) of the two-level pre-filter's FIRST level (bc_prefilter_i4.h) as 4-bit digits.
//
// Each vector gets 8 bits in two signed nibbles: Q = rint(val / vstep) in [-119, 119], vstep = max|v| / 119,
// Q = 16 d0 + d1 with d0 = floor((Q + 8) / 16) in [-7, 7] and d1 in [-8, 7], so that
//   u^ . v^ = scale_i * vstep * (16 * sum_k q d0 + sum_k q d1)
// is computed EXACTLY with v_dot8_i32_i4 on the 4-bit rows (eight products per instruction), and
//   | v^ - v |_2 <= sqrt(S) * vstep / 2.
// Unlike the int8 record (bc_i8_quant.h) GIGA's second vector gets both digits too: with the 4-bit rows' delta ~ 0.1 the
// slope term of the interval, (|s0| + d0) a d1 / c^(3/2), is no longer negligible and a one-digit v1 (vstep = max / 7) tripled
// the first level's candidates (tools/sim_two_level.py).
//
// Record layout (ints): [SP8][4] packed digits of k-group g (samples 8g .. 8g+7, sample 8g+j in bits 4j .. 4j+3)
//   = {v0 d0, v0 d1, v1 d0, v1 d1}, then BC_I4Q_HDR floats: fvs0, fvs1 (steps), fev0, fev1 (||v^ - v|| bounds, rounded up),
//   fvn (||v||, rounded up), vbad (1.0: NaN / inf in v)
#pragma once
#include "bc_i8_quant.h"

#define BC_I4Q_HDR 8
#define BC_I4Q_INTS(sp8) (4 * (sp8) + BC_I4Q_HDR)
#define BC_I4Q_QMAX 119

struct bc_i4q_scalars {
  double vstep0, vstep1;
  double inv0, inv1;
  bool vbad;
};

__device__ __forceinline__ bc_i4q_scalars bc_i4q_steps(double vmax0, double vmax1) {
  bc_i4q_scalars q;
  q.vbad = !(vmax0 < INFINITY) || !(vmax1 < INFINITY) || vmax0 != vmax0 || vmax1 != vmax1;
  q.vstep0 = vmax0 * (1. / (double)BC_I4Q_QMAX);
  q.vstep1 = vmax1 * (1. / (double)BC_I4Q_QMAX);
  // digits are rint(val * inv), one division per vector: see bc_i8q_steps -- |Q step - val| <= step / 2 * (1 + 1e-11)
  const double r0 = (vmax0 > 0. && !q.vbad) ? 1. / vmax0 : 0., r1 = (vmax1 > 0. && !q.vbad) ? 1. / vmax1 : 0.;
  q.inv0 = (q.vstep0 > 0.) ? (double)BC_I4Q_QMAX * r0 : 0.;
  q.inv1 = (q.vstep1 > 0.) ? (double)BC_I4Q_QMAX * r1 : 0.;
  return q;
}

// the two digits of ONE element of vector vv (0: score vector, 1: GIGA's second)
__device__ __forceinline__ void bc_i4q_elem(double val, double inv, int& d0, int& d1) {
  int Q = (int)rint(val * inv);
  Q = Q > BC_I4Q_QMAX ? BC_I4Q_QMAX : (Q < -BC_I4Q_QMAX ? -BC_I4Q_QMAX : Q);
  d0 = (Q + 8) >> 4;              // floor((Q + 8) / 16): arithmetic shift
  d1 = Q - 16 * d0;               // in [-8, 7]
}

// the four packed words of k-group g (samples 8g .. 8g+7)
template <int MODE>
__device__ __forceinline__ void bc_i4q_group(const double* __restrict__ v, int S, int g, const bc_i4q_scalars& q, unsigned (&w)[4]) {
  w[0] = w[1] = w[2] = w[3] = 0u;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g + j;
    if (k < S && !q.vbad) {
#pragma unroll
      for (int vv = 0; vv < (MODE == 0 ? 2 : 1); ++vv) {
        int d0, d1;
        bc_i4q_elem((MODE == 0) ? v[2 * k + vv] : v[k], vv == 0 ? q.inv0 : q.inv1, d0, d1);
        w[2 * vv] |= ((unsigned)d0 & 0xfu) << (4 * j);
        w[2 * vv + 1] |= ((unsigned)d1 & 0xfu) << (4 * j);
      }
    }
  }
}

struct bc_i4q_hdr {
  float fvs0, fvs1, fev0, fev1, fvn;
  bool vbad;
};

__device__ __forceinline__ bc_i4q_hdr bc_i4q_header(const bc_i4q_scalars& q, int S, double vn) {
  const double rs = (double)sqrtf((float)S) * (0.5 * (1. + 1e-6));      // >= sqrt(S) / 2
  bc_i4q_hdr h;
  h.fvn = __double2float_ru(vn);
  h.fev0 = __double2float_ru(rs * q.vstep0);
  h.fev1 = __double2float_ru(rs * q.vstep1);
  h.fvs0 = (float)q.vstep0;
  h.fvs1 = (float)q.vstep1;
  h.vbad = q.vbad;
  return h;
}

// One full wave quantises v into the record `qv4` (same contract as bc_i8q_wave: m0 / m1 are the WAVE-WIDE maxima of |v0| /
// |v1| here, the caller has them from the int8 record's pass).
template <int MODE>
__device__ __forceinline__ void bc_i4q_wave(const double* __restrict__ vsrc, int S, int SP8, double vn, int* __restrict__ qv4, int lane,
                                            double m0_all, double m1_all) {
  const bc_i4q_scalars q = bc_i4q_steps(m0_all, m1_all);
  for (int g = lane; g < SP8; g += 64) {
    unsigned w[4];
    bc_i4q_group<MODE>(vsrc, S, g, q, w);
    reinterpret_cast<int4*>(qv4)[g] = make_int4((int)w[0], (int)w[1], (int)w[2], (int)w[3]);
  }
  if (lane == 0) {
    const bc_i4q_hdr h = bc_i4q_header(q, S, vn);
    float* f = reinterpret_cast<float*>(qv4 + 4 * SP8);
    f[0] = h.fvs0; f[1] = h.fvs1; f[2] = h.fev0; f[3] = h.fev1; f[4] = h.fvn; f[5] = h.vbad ? 1.f : 0.f; f[6] = 0.f; f[7] = 0.f;
  }
}

#if defined(__HIPCC__)      // (device only: the host harness tests/i4_quant_harness.cpp exercises the per-element arithmetic through the group forms)
// One full wave writes BOTH records (the int8 one of bc_i8_quant.h and, if qv4 != nullptr, this one), ONE ELEMENT PER LANE: the
// group-per-lane forms (bc_i8q_wave, bc_i4q_wave) leave 39 / 51 lanes idle at S = 100 and took 2.5k + 3k cycles of the
// single-block step kernels' serial tail (tools/fin_stamps.py); here a lane quantises elements lane, lane + 64, ... and ORs
// their bytes / nibbles into a zeroed LDS image of the two records (ds_or), which the wave then copies out.  Same per-element
// arithmetic (bc_i8q_elem0 / elem1, bc_i4q_elem), hence the same digits.  m0 / m1: wave-wide maxima.  lds: >= 4 * (SP4 + SP8) ints.
template <int MODE>
__device__ __forceinline__ void bc_q_wave_both(const double* __restrict__ vsrc, int S, int SP4, int SP8, double vn, int* __restrict__ qv,
                                               int* __restrict__ qv4, int lane, double m0_all, double m1_all, int* lds) {
  const bc_i8q_scalars q8 = bc_i8q_steps(m0_all, m1_all);
  const bc_i4q_scalars q4 = bc_i4q_steps(m0_all, m1_all);
  const bool two = qv4 != nullptr;
  int* l8 = lds;
  int* l4 = lds + 4 * SP4;
  const int nz = 4 * SP4 + (two ? 4 * SP8 : 0);
  for (int i = lane; i < nz; i += 64) lds[i] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (!q8.vbad)
    for (int k = lane; k < S; k += 64) {
      const double v0 = (MODE == 0) ? vsrc[2 * k] : vsrc[k];
      const double v1 = (MODE == 0) ? vsrc[2 * k + 1] : 0.;
      int d0, d1;
      bc_i8q_elem0(v0, q8, d0, d1);
      const int g = k >> 2, sh = 8 * (k & 3);
      atomicOr(&l8[4 * g + 0], (int)(((unsigned)d0 & 0xffu) << sh));
      atomicOr(&l8[4 * g + 1], (int)(((unsigned)d1 & 0xffu) << sh));
      if (MODE == 0) atomicOr(&l8[4 * g + 2], (int)(((unsigned)bc_i8q_elem1(v1, q8) & 0xffu) << sh));
      if (two) {
        const int g8 = k >> 3, s4 = 4 * (k & 7);
        int a0, a1;
        bc_i4q_elem(v0, q4.inv0, a0, a1);
        atomicOr(&l4[4 * g8 + 0], (int)(((unsigned)a0 & 0xfu) << s4));
        atomicOr(&l4[4 * g8 + 1], (int)(((unsigned)a1 & 0xfu) << s4));
        if (MODE == 0) {
          bc_i4q_elem(v1, q4.inv1, a0, a1);
          atomicOr(&l4[4 * g8 + 2], (int)(((unsigned)a0 & 0xfu) << s4));
          atomicOr(&l4[4 * g8 + 3], (int)(((unsigned)a1 & 0xfu) << s4));
        }
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int g = lane; g < SP4; g += 64) reinterpret_cast<int4*>(qv)[g] = make_int4(l8[4 * g], l8[4 * g + 1], l8[4 * g + 2], l8[4 * g + 3]);
  if (two)
    for (int g = lane; g < SP8; g += 64) reinterpret_cast<int4*>(qv4)[g] = make_int4(l4[4 * g], l4[4 * g + 1], l4[4 * g + 2], l4[4 * g + 3]);
  if (lane == 0) {
    const bc_i8q_hdr h = bc_i8q_header(q8, S, vn);
    float* f = reinterpret_cast<float*>(qv + 4 * SP4);
    f[0] = h.fvs0; f[1] = h.fvs1; f[2] = h.fev0; f[3] = h.fev1; f[4] = h.fvn; f[5] = h.vbad ? 1.f : 0.f; f[6] = 0.f; f[7] = 0.f;
  }
  if (two && lane == 1) {
    const bc_i4q_hdr h = bc_i4q_header(q4, S, vn);
    float* f = reinterpret_cast<float*>(qv4 + 4 * SP8);
    f[0] = h.fvs0; f[1] = h.fvs1; f[2] = h.fev0; f[3] = h.fev1; f[4] = h.fvn; f[5] = h.vbad ? 1.f : 0.f; f[6] = 0.f; f[7] = 0.f;
  }
}
#endif
