// The transcendental bodies of the K1 epilogue: exp(x) and log1p(exp(-a)), table-driven, division-free.
//
// Why not libm: the logistic projections evaluate one log1p(exp(.)) (and the beta-likelihood three more exp) per
// element of Phi, 10^8 times per million rows, on the vector pipe that the fp64 matrix cores keep busy (K1's time is
// the SUM of its MFMA time and its other vector work).  Round 2's written-out bodies (Cody-Waite +
// degree-12 polynomial, 2 atanh series, TWO fp64 divisions) cost ~106 vector instructions per element of the
// logistic log-likelihood (SQ_INSTS_VALU, profiles/r03_k1_pmc_logistic.csv); these cost ~36:
//
//   exp(x):  x = (64 e + j) ln2/64 + r, |r| <= ln2/128;  exp(x) = 2^e * T[j] * (1 + r + r^2/2 + ... + r^5/120)
//            (T[j] = 2^(j/64) from LDS; truncation r^6/720 <= 3.5e-17)
//   log1p(exp(-a)), a >= 0:  u = exp(-a), f = 1 + u in (1, 2];  i = round(256 (f - 1)), c_i = 1 + i/256;
//            ln f = lc_i + log1p(t),  t = f*rc_i - 1 (one fma, |t| <= 2^-9),  rc_i = RN(1/c_i), lc_i = RN(-ln rc_i)
//            (the log of the ROUNDED reciprocal: the identity is exact), log1p(t) to degree 6 (t^7/7 <= 8e-18 t), plus
//            the first-order term (u - (f - 1)) * rc_i for the rounding of 1 + u.  Entry 0 is {1, 0}: for small u the
//            result keeps u's relative accuracy; entry 256 is {1/2, RN(ln 2)}: a = 0 gives exactly RN(ln 2).
//
// Measured against 80-bit arithmetic (tests/test_k1_math_cpu.py, two million arguments each): exp <= 1.0 ulp,
// log1p(exp(-a)) <= 1.6 ulp.  Parity bar of the projections: 1e-11 * (1 + max|f|) against the reference (golden F2).
//
// `tab` points at BC_K1_TAB_DOUBLES doubles laid out as bc_k1_tables.h says (in LDS inside K1).
// Plain C99 / C++: compiled by hipcc for the device and by gcc for the host-side accuracy test.
#ifndef BC_K1_MATH_H
#define BC_K1_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>
#include "bc_k1_tables.h"

#if defined(__HIPCC__)
#define BC_KM __host__ __device__ __forceinline__
#else
#define BC_KM static inline
#endif

// exp(x) for |x| <= 2e4 (beyond 745 / 709 the result is 0 / +inf through ldexp); a NaN gives a NaN.  No clamps: the
// callers bound their arguments once per element instead of twice per exp (a compare-and-select pair on doubles is
// four instructions).
BC_KM unsigned bc_k1_lo32(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return (unsigned)u;
}

BC_KM double bc_exp_tab_core(double x, const double* tab) {
  // z = x * 64/ln2 + 1.5 * 2^52: one rounding puts the nearest integer k in the low mantissa bits (two's complement in
  // the low 32 for |k| < 2^31, i.e. |x| < 2e7) -- no v_rndne, no v_cvt: the integer IS the low register of z
  const double z = fma(x, 92.33248261689366, 6755399441055744.);
  const double kd = z - 6755399441055744.;
  double r = fma(-kd, 0.010830424695086549, x);        // ln2/64, high 32 bits: kd * hi is exact for |kd| < 2^21
  r = fma(-kd, 1.162596423439437e-12, r);              // ln2/64, low part
  const int ki = (int)bc_k1_lo32(z);
  const int j = ki & 63, e = ki >> 6;                  // ki = 64 e + j, j in 0..63 (arithmetic shift)
  double p = fma(r, 1. / 120., 1. / 24.);
  p = fma(p, r, 1. / 6.);
  p = fma(p, r, 0.5);
  p = fma(p, r * r, r);                                // exp(r) - 1
  const double t = tab[j];
  return ldexp(fma(t, p, t), e);
}

// exp(x) for x <= 0 up to rounding (the beta-likelihoods' exp(-b q), q >= 0): only the lower bound needs a clamp, and
// the comparison is written so that a NaN passes through it (three instructions instead of five)
BC_KM double bc_exp_tab_nonpos(double x, const double* tab) {
  const double xc = !(x < -800.) ? x : -800.;
  return bc_exp_tab_core(xc, tab);
}

// exp(x) for any x (NaN in, NaN out)
BC_KM double bc_exp_tab(double x, const double* tab) {
  const double xc = fmin(fmax(x, -800.), 800.);        // (fmax / fmin drop a NaN: restored below)
  const double r = bc_exp_tab_core(xc, tab);
  return (x != x) ? x : r;
}

// log(1 + exp(-a)) for 0 <= a <= 2e4 (a NaN gives a NaN); also hands out u = exp(-a) and f = 1 + u (rounded)
BC_KM double bc_log1p_exp_neg_tab_uf(double a, const double* tab, double* u_out, double* f_out) {
  const double u = bc_exp_tab_core(-a, tab);
  const double f = 1. + u;
  *u_out = u;
  *f_out = f;
  const double fm1 = f - 1.;                           // exact
  // nearest c_i = 1 + i/256, i in 0..256, from the low mantissa bits of fm1 * 256 + 1.5 * 2^52 (fm1 in [0, 1]: the low
  // word IS i; the unsigned min only matters for a NaN, whose arbitrary payload bits must not index past the 257 table
  // entries -- the table is the last piece of the kernels' LDS -- and whose result is a NaN either way)
  const unsigned iu = bc_k1_lo32(fma(fm1, 256., 6755399441055744.));
  const int i = (int)(iu < 256u ? iu : 256u);
  const double rc = tab[BC_K1_EXP_N + 2 * i], lc = tab[BC_K1_EXP_N + 2 * i + 1];
  const double t = fma(f, rc, -1.);
  double q = fma(t, -1. / 6., 1. / 5.);
  q = fma(q, t, -1. / 4.);
  q = fma(q, t, 1. / 3.);
  q = fma(q, t, -1. / 2.);
  q = q * t;
  const double s = fma(q, t, t);                       // log1p(t)
  const double c = (u - fm1) * rc;                     // d/df ln f * (rounding error of 1 + u)
  return lc + (s + c);
}

BC_KM double bc_log1p_exp_neg_tab(double a, const double* tab) {
  double u, f;
  return bc_log1p_exp_neg_tab_uf(a, tab, &u, &f);
}

// 1/f for f in [1, 2]: the hardware's approximate reciprocal (v_rcp_f64; on the host a float division stands in for it)
// and two Newton steps -- five instructions, no scaling or fix-up cases in this range
BC_KM double bc_rcp_1_2(double f) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(f);
#else
  double r = (double)(1.0f / (float)f);
#endif
  double e = fma(-f, r, 1.);
  r = fma(r, e, r);
  e = fma(-f, r, 1.);
  return fma(r, e, r);
}

// The logistic beta-likelihood (model_lr.py:81-86):  -( (b+1)/b (1+e^m)^-b - ((1+e^m)^(-b-1) + (1+e^-m)^(-b-1)) ),
// c0 = (b+1)/b, c1 = -b, c2 = -b-1.  The reference evaluates two exp and three pow per element; here, with
//   L1 = log(1+e^m),  L2 = log(1+e^-m) = L1 - m   (the smaller of the two is Ls = log1p(e^-|m|), the other one adds |m|):
//   (1+e^m)^-b      = exp(c1 L1)                                   exp #1
//   (1+e^-m)^(-b-1) = exp(c2 L2)                                   exp #2
//   (1+e^m)^(-b-1)  = exp(c1 L1) / (1+e^m) = exp(c1 L1) * w / f,   w = 1 (m <= 0) or u = e^-|m| (m > 0),  f = 1 + u
// i.e. exp(-|m|), one log1p, two exp and one reciprocal of f in [1, 2] (round 3 took a third exp for the last line).
// Same saturation as the reference's IEEE overflow semantics (m -> +inf: +1, m -> -inf: -1/b exactly as -(c0 - 1)).  Where
// np.exp overflows the reference's powers flush to EXACTLY 0: for the two powers with exponent -b-1 <= -1 the smooth value is
// below 1e-308 there, but (1+e^m)^-b at m = 709.78 is exp(-709.78 b) -- 1e-31 at b = 0.1 and 8e-4 at b = 0.01 (times c0 = 101) --
// so the first power takes the reference's cutoff: exactly 0 for m > log(DBL_MAX) (golden F22).  NaN in, NaN out.
#define BC_EXP_OVERFLOW_ARG 709.782712893384          // largest m with finite exp(m) (0x40862E42FEFA39EF)
BC_KM double bc_logistic_beta_value(double m, double c0, double c1, double c2, const double* tab) {
  const double am = fmin(fabs(m), 800.);              // beyond that every term has saturated (fmin drops a NaN: restored below)
  double u, f;
  const double Ls = bc_log1p_exp_neg_tab_uf(am, tab, &u, &f);    // log(1 + e^-|m|)
  const double Ll = Ls + am;                          // log(1 + e^+|m|)
  const int neg = m <= 0.;
  const double L1 = neg ? Ls : Ll, L2 = neg ? Ll : Ls;
  const double e1s = bc_exp_tab_core(c1 * L1, tab);   // arguments in [-(b+1) * 801, 0]
  const double e1 = (m > BC_EXP_OVERFLOW_ARG) ? 0. : e1s;          // (1 + inf)^-b == 0 in the reference (model_lr.py:85)
  const double e3 = bc_exp_tab_core(c2 * L2, tab);
  const double e2 = e1 * ((neg ? 1. : u) * bc_rcp_1_2(f));
  const double v = -((c0 * e1) - (e2 + e3));
  return (m != m) ? m : v;
}

// ---- round 5: the same value with ONE exp and the log1p less.  Per element the body above spends exp(-|m|), a log1p and two
// more exp on what are powers of f = 1 + e^-|m| in [1, 2] and of e^-|m| itself:
//   f^-b       = rc_i^b * (1 + t)^-b,  t = f * rc_i - 1, |t| <= 2^-9  (the log table's reduction, bc_log1p_exp_neg_tab_uf)
//              = Tb[i] * (1 + a t + a(a-1)/2 t^2 + ... to t^6),  a = -b;  Tb[i] = exp(-b * lc_i) -- 257 entries that depend on
//                beta: each K1 block builds them once in its LDS prologue (bc_pow_table_entry); truncation C(a,7) 2^-63:
//                1e-15 at b = 8, 2e-12 at b = 32 (the limit the library accepts for this model, BC_K1_POWTAB_MAX_BETA)
//   (e^-|m|)^b = exp(-b |m|)                                                        -- the second and last exp
//   m <= 0:  (1+e^m) = f:       e1 = f^-b,           e2 = f^-b / f,            e3 = (f / u)^(-b-1) = u u^b f^-b / f
//   m  > 0:  (1+e^m) = f / u:   e1 = u^b f^-b,       e2 = u u^b f^-b / f,      e3 = f^-b / f
// so e2 + e3 = (f^-b / f) (1 + u u^b) on both sides.  ~62 vector instructions per element instead of ~83; same saturation
// limits, same cutoff at np.exp's overflow; max abs error against 80-bit arithmetic < 1e-14 for b <= 1 (tests/k1_math_harness.c).
#define BC_K1_POWTAB_MAX_BETA 32.0
// the series' coefficients b1..b6 of (1 + t)^a, a = -beta (host side: they travel as model constants)
BC_KM void bc_powtab_coefs(double a, double (*k)[6]) {
  (*k)[0] = a;
  (*k)[1] = (*k)[0] * (a - 1.) * (1. / 2.);
  (*k)[2] = (*k)[1] * (a - 2.) * (1. / 3.);
  (*k)[3] = (*k)[2] * (a - 3.) * (1. / 4.);
  (*k)[4] = (*k)[3] * (a - 4.) * (1. / 5.);
  (*k)[5] = (*k)[4] * (a - 5.) * (1. / 6.);
}
// entry i of the power table: rc_i^beta = exp(-beta * lc_i) with lc_i = -ln(rc_i) from the log table
BC_KM double bc_pow_table_entry(int i, double c1 /* = -beta */, const double* tab) {
  return bc_exp_tab_core(c1 * tab[BC_K1_EXP_N + 2 * i + 1], tab);
}
// k1..k6: the coefficients above (k1 == c1 == -beta)
BC_KM double bc_logistic_beta_value_pt(double m, double c0, double k1, double k2, double k3, double k4, double k5, double k6,
                                       const double* tab, const double* tb) {
  const double am = fmin(fabs(m), 800.);              // (fmin drops a NaN: restored below)
  const double u = bc_exp_tab_core(-am, tab);         // e^-|m|
  const double ub = bc_exp_tab_core(k1 * am, tab);    // (e^-|m|)^beta
  const double f = 1. + u;
  const double fm1 = f - 1.;                          // exact
  const unsigned iu = bc_k1_lo32(fma(fm1, 256., 6755399441055744.));
  const int i = (int)(iu < 256u ? iu : 256u);
  const double rc = tab[BC_K1_EXP_N + 2 * i], T = tb[i];
  const double t = fma(f, rc, -1.);
  double q = fma(t, k6, k5);
  q = fma(q, t, k4);
  q = fma(q, t, k3);
  q = fma(q, t, k2);
  q = fma(q, t, k1);
  const double fp = fma(T * q, t, T);                 // f^-beta
  const double fpf = fp * bc_rcp_1_2(f);              // f^(-beta-1)
  const double e1s = (m <= 0.) ? fp : ub * fp;        // (1 + e^m)^-beta
  const double e1 = (m > BC_EXP_OVERFLOW_ARG) ? 0. : e1s;          // (1 + inf)^-b == 0 in the reference (model_lr.py:85)
  const double e23 = fma(u * ub, fpf, fpf);           // (1 + e^m)^(-beta-1) + (1 + e^-m)^(-beta-1)
  const double v = -((c0 * e1) - e23);
  return (m != m) ? m : v;
}

#endif  // BC_K1_MATH_H
