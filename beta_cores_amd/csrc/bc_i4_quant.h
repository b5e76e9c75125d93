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
  q.vstep0 = vmax0 / (double)BC_I4Q_QMAX;
  q.vstep1 = vmax1 / (double)BC_I4Q_QMAX;
  // digits are rint(val * inv): see bc_i8_quant.h -- |Q step - val| <= step / 2 * (1 + 1e-11), inside the sweep's 1.00001
  q.inv0 = (q.vstep0 > 0. && !q.vbad) ? 1. / q.vstep0 : 0.;
  q.inv1 = (q.vstep1 > 0. && !q.vbad) ? 1. / q.vstep1 : 0.;
  return q;
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
        const double val = (MODE == 0) ? v[2 * k + vv] : v[k];
        int Q = (int)rint(val * (vv == 0 ? q.inv0 : q.inv1));
        Q = Q > BC_I4Q_QMAX ? BC_I4Q_QMAX : (Q < -BC_I4Q_QMAX ? -BC_I4Q_QMAX : Q);
        const int d0 = (Q + 8) >> 4;              // floor((Q + 8) / 16): arithmetic shift
        const int d1 = Q - 16 * d0;               // in [-8, 7]
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
  const double rs = sqrt((double)S) * 0.5;
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
