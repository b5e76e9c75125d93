// exp() with the bits NumPy produces on AVX-512 hosts.
//
// Why: a data row with all-zero features projects to S copies of ONE value c, and whether the reference's
// `lls -= lls.mean(axis=1)` (projector.py:26,55) leaves such a row exactly 0 -- and with it whether the row is an all-zero
// (NaN-candidate) row of the greedy-VI classes (golden F13) -- depends on the LAST BIT of c.  For the
// beta-likelihoods c contains np.exp(...) (model_neurlinr.py:107, gaussian.py:42,55).  On x86-64 hosts with AVX512_SKX
// (the machines the goldens were generated and are checked on) NumPy >= 1.22 evaluates float64 np.exp with Intel's
// SVML routine __svml_exp8_ha that it bundles (numpy/_core/src/umath/svml, linux/avx512/svml_z0_exp_d_ha.s; published under
// BSD-3-Clause as part of NumPy), NOT with glibc's exp(): the two differ in the last bit for ~4.6 % of arguments.
// This header restates that routine's main path operation for operation (same constants, same fma sequence, same
// table), so that K1 can give the FEW constant rows of a projection the reference's bits.  It is not used for ordinary
// rows (their tolerance is 1e-11; they use the short device exp bodies).
//
// Algorithm (Tang-style, 16-entry table):  M = RZ(x*log2(e) + Shifter) puts N = round-toward-zero(16 x log2 e)/16 in
// the low mantissa bits; j = low 4 bits selects Th[j] + Tl[j] ~ 2^(j/16); R = (x - N*ln2_hi) - N*ln2_lo;
// P = ((c7 R + c6) R^2 + (c5 R + c4)) R^2 + (c3 R + c2);  exp(x) = scalef(Th + Th*(P*R + Tl), floor(N)).
// Arguments with |x| >= 707.703 (and NaN) take SVML's scalar call-out in the original; here they return a NaN-boxed
// "not covered" answer through *covered = 0 and the caller falls back to its ordinary exp.
//
// Plain C99 / C++: compiled by hipcc for the device and by gcc for tests/test_np_exp_cpu.py, which checks it bit for
// bit against np.exp on the running machine (skipped where NumPy does not dispatch to SVML).
#ifndef BC_NP_EXP_H
#define BC_NP_EXP_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define BC_HD __host__ __device__ __forceinline__
#else
#define BC_HD static inline
#endif

BC_HD double bc_bits2d(uint64_t u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}

BC_HD uint64_t bc_d2bits(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return u;
}

BC_HD double bc_np_exp(double x, int* covered) {
  // __svml_dexp_ha_data_internal_avx512: Exp_tbl_H (2^(j/16) high), Exp_tbl_L (low), L2E, Shifter, L2H, L2L, poly
  const uint64_t TH[16] = {0x3ff0000000000000ull, 0x3ff0b5586cf9890full, 0x3ff172b83c7d517bull, 0x3ff2387a6e756238ull,
                           0x3ff306fe0a31b715ull, 0x3ff3dea64c123422ull, 0x3ff4bfdad5362a27ull, 0x3ff5ab07dd485429ull,
                           0x3ff6a09e667f3bcdull, 0x3ff7a11473eb0187ull, 0x3ff8ace5422aa0dbull, 0x3ff9c49182a3f090ull,
                           0x3ffae89f995ad3adull, 0x3ffc199bdd85529cull, 0x3ffd5818dcfba487ull, 0x3ffea4afa2a490daull};
  const uint64_t TL[16] = {0x0000000000000000ull, 0x3c979aa65d837b6dull, 0xbc801b15eaa59348ull, 0x3c968efde3a8a894ull,
                           0x3c834d754db0abb6ull, 0x3c859f48a72a4c6dull, 0x3c7690cebb7aafb0ull, 0x3c9063e1e21c5409ull,
                           0xbc93b3efbf5e2228ull, 0xbc7b32dcb94da51dull, 0x3c8db72fc1f0eab4ull, 0x3c71affc2b91ce27ull,
                           0x3c8c1a7792cb3387ull, 0x3c736eae30af0cb3ull, 0x3c74a385a63d07a7ull, 0xbc8ff7128fd391f0ull};
  const double L2E = bc_bits2d(0x3ff71547652b82feull);       // log2(e)
  const double SHIFTER = bc_bits2d(0x42f8000000003ff0ull);   // 1.5 * 2^48 + 1023: ulp = 2^-4
  const double L2H = bc_bits2d(0x3fe62e42fefa39efull);       // ln 2, high
  const double L2L = bc_bits2d(0x3c7abc9e3b39803full);       // ln 2, low
  const double C7 = bc_bits2d(0x3f57411836940c04ull), C6 = bc_bits2d(0x3f81101cbbc265c0ull);
  const double C5 = bc_bits2d(0x3fa55557242d68feull), C4 = bc_bits2d(0x3fc5555553939732ull);
  const double C3 = bc_bits2d(0x3fe000000000d008ull), C2 = bc_bits2d(0x3fefffffffffff70ull);
  const double THRESH = bc_bits2d(0x40861da04cbafe44ull);    // 707.703...: beyond it SVML calls out to scalar code
  *covered = (fabs(x) < THRESH) ? 1 : 0;                     // false for NaN as well
  if (!*covered) return x;
  // M = x*L2E + SHIFTER rounded TOWARD ZERO (vfmadd213pd {rz-sae}).  M > 0, so that is the round-to-nearest result
  // minus one ulp (2^-4) whenever rounding went up; the sign of the exact remainder x*L2E - (M_rn - SHIFTER) tells.
  double m = fma(x, L2E, SHIFTER);
  const double n_rn = m - SHIFTER;                           // exact: a multiple of 2^-4 of small magnitude
  const double rem = fma(x, L2E, -n_rn);                     // x*L2E - n_rn, one rounding, sign exact
  if (rem < 0.) m = m - 0.0625;                              // exact: m is a multiple of 2^-4 near 1.5 * 2^48
  const double n = m - SHIFTER;                              // vsubpd: N = k/16
  const int j = (int)(bc_d2bits(m) & 15u);                   // vpermt2pd index: low 4 mantissa bits
  double r = fma(-n, L2H, x);                                // vfnmadd213pd: x - N*L2H
  r = fma(-L2L, n, r);                                       // vfnmadd231pd: R = (x - N*L2H) - L2L*N
  const double r2 = r * r;
  double p = fma(C7, r, C6);                                 // zmm12 = c7*R + c6
  const double q = fma(C5, r, C4);                           // zmm9
  const double s = fma(C3, r, C2);                           // zmm11
  p = fma(r2, p, q);
  p = fma(r2, p, s);
  const double t = fma(p, r, bc_bits2d(TL[j]));              // zmm3 = P*R + Tl
  const double th = bc_bits2d(TH[j]);
  const double e = fma(th, t, th);                           // Th*zmm3 + Th
  return ldexp(e, (int)floor(n));                            // vscalefpd: e * 2^floor(N)
}

#endif  // BC_NP_EXP_H
