// bc_np_pow2.h -- np.power(2., y) with NumPy's bits on AVX-512 hosts.
//
// model_lr.py:85 raises (1 + np.exp(m)) to the powers -beta and -beta-1 with `**`, i.e. np.power, which NumPy >= 1.22
// evaluates with the SVML routine it bundles (`__svml_pow8_ha`, 18 call sites in _multiarray_umath) on x86-64 hosts with
// AVX512_SKX -- the hosts the goldens were generated and are checked on.  Its last bit differs from libm's pow for 5.3 %
// and from np.exp2 for 1.5 % of exponents at base 2.  The projection of a data row z = 0 consists of S copies of
// -((b+1)/b 2^-b - 2 2^(-b-1)), and whether `c - mean` (projector.py:55) is exactly 0 hangs on the last bit of those two
// powers (golden F20), so the C layer needs them with NumPy's bits when the caller does not hand the
// constant in (the Python layer does: likelihoods.LogisticRegression.beta_value_at_zero evaluates it with NumPy itself).
//
// Only BASE 2 is needed, and there the routine's main path collapses: its log2 stage (mantissa in [1/2, 1), a 14-bit
// reciprocal rounded to 5 fraction bits, table + polynomial in R' = (rcp * mant - 1) / 2) returns exactly (hi, lo) = (1, 0)
// for x = 2 (rcp = 2, R' = 0, table entry 0 = {0, 0}, exponent 1), the round-toward-zero double-double product with y is
// y itself, and what remains is the exp2 stage, restated here operation for operation:
//     k = floor(16 y)            (the routine adds the shifter 1.5 * 2^48 + 1023 * 16 rounding DOWN and reads k off the mantissa)
//     r = RD(y - k / 16)         (vreducepd, ROUNDED DOWN where inexact; 0 <= r < 1/16)   j = k mod 16,  N = (k - j) / 16
//     p = ((c5 r + c4) r^2 + (c3 r + c2)) r^2 + (c1 r + c0);   p = p r + Tlo[j];   p = p T[j] + T[j];   result = p 2^N
// with the 16-entry table T[j] = RN(2^(j/16)), its tails Tlo[j] (relative), and the six coefficients of the routine's data
// block (`__svml_dpow_ha_data_internal_avx512` + 0x200 .. 0x840).  The main path covers |y| <= 1021.5; outside it (and for
// NaN) `*covered` is 0 and the caller uses the ordinary pow.  tests/test_np_pow2_cpu.py compiles this header for the host and
// compares it with np.power(2., y) bit for bit on two million exponents (where NumPy dispatches to that routine).
// Plain C99: host code only (model constants are evaluated on the host, bc_project.hip: model_constants).
#ifndef BC_NP_POW2_H
#define BC_NP_POW2_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline double bc_np_pow2_bits(uint64_t u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}

static inline double bc_np_pow2(double y, int* covered) {
  static const uint64_t T[16] = {
      0x3ff0000000000000ull, 0x3ff0b5586cf9890full, 0x3ff172b83c7d517bull, 0x3ff2387a6e756238ull, 0x3ff306fe0a31b715ull, 0x3ff3dea64c123422ull,
      0x3ff4bfdad5362a27ull, 0x3ff5ab07dd485429ull, 0x3ff6a09e667f3bcdull, 0x3ff7a11473eb0187ull, 0x3ff8ace5422aa0dbull, 0x3ff9c49182a3f090ull,
      0x3ffae89f995ad3adull, 0x3ffc199bdd85529cull, 0x3ffd5818dcfba487ull, 0x3ffea4afa2a490daull};
  static const uint64_t TLO[16] = {
      0x0000000000000000ull, 0x3c979aa65d837b6dull, 0xbc801b15eaa59348ull, 0x3c968efde3a8a894ull, 0x3c834d754db0abb6ull, 0x3c859f48a72a4c6dull,
      0x3c7690cebb7aafb0ull, 0x3c9063e1e21c5409ull, 0xbc93b3efbf5e2228ull, 0xbc7b32dcb94da51dull, 0x3c8db72fc1f0eab4ull, 0x3c71affc2b91ce27ull,
      0x3c8c1a7792cb3387ull, 0x3c736eae30af0cb3ull, 0x3c74a385a63d07a7ull, 0xbc8ff7128fd391f0ull};
  *covered = 0;
  if (!(fabs(y) <= 1021.5)) return 0.;             /* rare path of the routine (and NaN): not restated */
  *covered = 1;
  const double kd = floor(y * 16.);                /* exact scaling, exact floor */
  /* vreducepd imm8 = 0x41: r = y - floor(16 y) / 16 with the SUBTRACTION rounded DOWN too (the instruction's rounding control
     applies to both steps).  It is inexact only for y in (-1/32, 0) -- r then has fewer fraction bits than y -- and there
     round-to-nearest would be one ulp above in half of the cases.  TwoSum gives the rounding error's sign. */
  const double c = -kd * 0.0625;
  double r = y + c;
  {
    const double bb = r - c;
    const double e = (c - (r - bb)) + (y - bb);    /* exact: (y + c) - r */
    if (e < 0.) r = nextafter(r, -INFINITY);
  }
  const long k = (long)kd;
  const int j = (int)(k & 15);                     /* two's complement: the non-negative residue */
  const int n = (int)((k - j) / 16);
  const double r2 = r * r;
  double p = fma(bc_np_pow2_bits(0x3f24a1d7f58c2d59ull), r, bc_np_pow2_bits(0x3f55d7472783d279ull));
  const double q = fma(bc_np_pow2_bits(0x3f83b2ad1b14ebaaull), r, bc_np_pow2_bits(0x3fac6b08d4ad8eb9ull));
  const double s = fma(bc_np_pow2_bits(0x3fcebfbdff84554dull), r, bc_np_pow2_bits(0x3fe62e42fefa398bull));
  p = fma(p, r2, q);
  p = fma(p, r2, s);
  p = fma(p, r, bc_np_pow2_bits(TLO[j]));
  const double t = bc_np_pow2_bits(T[j]);
  p = fma(p, t, t);
  return ldexp(p, n);                              /* the routine multiplies by 2^N built from the shifter's bits */
}

#endif  /* BC_NP_POW2_H */
