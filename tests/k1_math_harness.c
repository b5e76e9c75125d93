/* Host-side accuracy check of beta_cores_amd/csrc/bc_k1_math.h (the K1 epilogue's exp and log1p(exp(-a))) against
 * 80-bit long double arithmetic: prints the maximum and mean error in ulp over n pseudo-random arguments.
 * Built and driven by tests/test_k1_math_cpu.py. */
#include <stdio.h>
#include <stdlib.h>
#include "bc_k1_math.h"

static double ulp_of(double v) {
  if (v == 0.0) return 4.9406564584124654e-324;
  double a = fabs(v);
  return nextafter(a, INFINITY) - a;
}

static unsigned long long s = 88172645463325252ull;
static double rnd(void) {      /* xorshift64, uniform in [0, 1) */
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  return (double)(s >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 1000000;
  const unsigned long long tabbits[BC_K1_TAB_DOUBLES] = BC_K1_TABLE_INIT;
  double tab[BC_K1_TAB_DOUBLES];
  memcpy(tab, tabbits, sizeof(tab));
  double emax = 0, esum = 0, lmax = 0, lsum = 0, xe = 0, xl = 0;
  for (long k = 0; k < n; ++k) {
    /* exp: arguments over the whole range the projections produce, dense near 0 */
    double x = (k & 1) ? -700.0 * rnd() : -40.0 * rnd() * rnd() * rnd();
    if ((k & 7) == 7) x = 700.0 * rnd();
    const long double we = expl((long double)x);
    const double ge = bc_exp_tab(x, tab);
    const double ee = (double)(fabsl((long double)ge - we) / (long double)ulp_of((double)we));
    if (ee > emax) { emax = ee; xe = x; }
    esum += ee;
    /* log1p(exp(-a)) */
    double a = (k & 1) ? 40.0 * rnd() : 8.0 * rnd() * rnd();
    if ((k & 15) == 15) a = 700.0 * rnd();
    const long double wl = log1pl(expl(-(long double)a));
    const double gl = bc_log1p_exp_neg_tab(a, tab);
    const double el = (double)(fabsl((long double)gl - wl) / (long double)ulp_of((double)wl));
    if (el > lmax) { lmax = el; xl = a; }
    lsum += el;
  }
  const double l0 = bc_log1p_exp_neg_tab(0.0, tab);
  printf("exp: max %.3f ulp (at %.17g) mean %.3f | log1p_exp_neg: max %.3f ulp (at %.17g) mean %.3f | at0 %s\n", emax, xe, esum / n, lmax, xl,
         lsum / n, l0 == 0.6931471805599453 ? "exact-ln2" : "NOT-ln2");
  /* the logistic beta-likelihood body against the reference's formula in 80-bit arithmetic (model_lr.py:85) */
  double bmax = 0.;
  const double betas[3] = {0.1, 0.5, 1.0};
  for (int bi = 0; bi < 3; ++bi) {
    const double b = betas[bi], c0 = (b + 1.) / b, c1 = -b, c2 = -b - 1.;
    for (long i = 0; i < n / 4; ++i) {
      const double r = rnd();
      const double m = (i & 3) == 0 ? (r - 0.5) * 1800. : ((i & 3) == 1 ? (r - 0.5) * 8. : (r - 0.5) * 80.);
      const double got = bc_logistic_beta_value(m, c0, c1, c2, tab);
      const long double em = expl((long double)m), enm = expl(-(long double)m);
      const long double want = -(((long double)c0) * powl(1.L + em, (long double)c1) - (powl(1.L + em, (long double)c2) + powl(1.L + enm, (long double)c2)));
      const double err = (double)fabsl((long double)got - want);
      if (err > bmax) bmax = err;
    }
    /* saturation: exactly the reference's limits */
    if (bc_logistic_beta_value(900., c0, c1, c2, tab) != 1.0) { printf("beta-lik +inf limit BAD\n"); return 1; }
    if (bc_logistic_beta_value(-900., c0, c1, c2, tab) != -(c0 - 1.0)) { printf("beta-lik -inf limit BAD\n"); return 1; }
    if (bc_logistic_beta_value(-47., c0, c1, c2, tab) != -(c0 - 1.0)) { printf("beta-lik saturated row not constant\n"); return 1; }
  }
  {
    /* np.exp's overflow (golden F22): past log(DBL_MAX) the reference's (1 + inf)^-b is exactly 0 and the value exactly 1;
       just below it the smooth formula holds -- at b = 0.01 the two differ by 0.083 */
    const double b = 0.01, c0 = (b + 1.) / b, c1 = -b, c2 = -b - 1.;
    if (bc_logistic_beta_value(709.79, c0, c1, c2, tab) != 1.0 || bc_logistic_beta_value(745., c0, c1, c2, tab) != 1.0 ||
        bc_logistic_beta_value(5000., c0, c1, c2, tab) != 1.0) { printf("beta-lik exp-overflow cutoff BAD\n"); return 1; }
    const double below = bc_logistic_beta_value(709.78, c0, c1, c2, tab);
    const long double wantb = -(((long double)c0) * powl(1.L + expl(709.78L), (long double)c1) - (powl(1.L + expl(709.78L), (long double)c2) + 1.L));
    if (fabsl((long double)below - wantb) > 1e-14L || below > 0.92) { printf("beta-lik below the cutoff BAD %.17g\n", below); return 1; }
  }
  /* round 5: the body K1 runs -- power table + binomial series instead of log1p + exp (bc_logistic_beta_value_pt) */
  {
    const double pb[6] = {0.01, 0.1, 0.5, 1.0, 8.0, 32.0};
    double pmax_small = 0., pmax_big = 0.;
    for (int bi = 0; bi < 6; ++bi) {
      const double b = pb[bi], c0 = (b + 1.) / b;
      double k[6], tb[BC_K1_LOG_N];
      bc_powtab_coefs(-b, &k);
      for (int i = 0; i < BC_K1_LOG_N; ++i) tb[i] = bc_pow_table_entry(i, -b, tab);
      for (long i = 0; i < n / 4; ++i) {
        const double r = rnd();
        const double m = (i & 3) == 0 ? (r - 0.5) * 1400. : ((i & 3) == 1 ? (r - 0.5) * 8. : (r - 0.5) * 80.);
        const double got = bc_logistic_beta_value_pt(m, c0, k[0], k[1], k[2], k[3], k[4], k[5], tab, tb);
        const long double em = expl((long double)m), enm = expl(-(long double)m);
        const long double want = -(((long double)c0) * powl(1.L + em, -(long double)b) - (powl(1.L + em, -(long double)b - 1.L) + powl(1.L + enm, -(long double)b - 1.L)));
        const double err = (double)fabsl((long double)got - want);
        if (b <= 1.) { if (err > pmax_small) pmax_small = err; } else if (err > pmax_big) pmax_big = err;
      }
      /* saturation and np.exp's overflow: exactly the reference's values */
      if (bc_logistic_beta_value_pt(900., c0, k[0], k[1], k[2], k[3], k[4], k[5], tab, tb) != 1.0 ||
          bc_logistic_beta_value_pt(709.79, c0, k[0], k[1], k[2], k[3], k[4], k[5], tab, tb) != 1.0 ||
          bc_logistic_beta_value_pt(-900., c0, k[0], k[1], k[2], k[3], k[4], k[5], tab, tb) != -(c0 - 1.0) ||
          bc_logistic_beta_value_pt(-47., c0, k[0], k[1], k[2], k[3], k[4], k[5], tab, tb) != -(c0 - 1.0) ||
          !isnan(bc_logistic_beta_value_pt(NAN, c0, k[0], k[1], k[2], k[3], k[4], k[5], tab, tb))) {
        printf("beta-lik (power table) limits BAD at beta %g\n", b);
        return 1;
      }
    }
    printf("logistic_beta_value_pt: max abs err %.3g (beta <= 1; values O(1/beta)), %.3g (beta = 8, 32)\n", pmax_small, pmax_big);
    /* beta = 0.01: values reach 100; 80-bit pow itself is good to ~1e-17 relative */
    if (!(pmax_small < 2e-13) || !(pmax_big < 1e-11)) return 1;
  }
  printf("logistic_beta_value: max abs err %.3g (values are O(1..1/beta))\n", bmax);
  if (!(bmax < 1e-14)) return 1;
  /* special values */
  int ok = 1;
  ok &= isnan(bc_logistic_beta_value(NAN, 11., -0.1, -1.1, tab));
  ok &= bc_exp_tab(0.0, tab) == 1.0;
  ok &= bc_exp_tab(-1000.0, tab) == 0.0;
  ok &= isinf(bc_exp_tab(1000.0, tab));
  ok &= isnan(bc_exp_tab(NAN, tab));
  ok &= isnan(bc_log1p_exp_neg_tab(NAN, tab));
  {
    /* NaNs with payloads: whatever low mantissa bits reach the table index, it must stay within the 257 entries (under
       ASan / UBSan in tests/test_sanitized_cpu.py an escape is a hard error, here the result must still be a NaN) */
    const unsigned long long payloads[] = {0x7ff80000000001ffULL, 0x7ff8000000000100ULL, 0x7ff80000ffffffffULL, 0xfff8000012345678ULL,
                                           0x7ff0000000000001ULL, 0x7ff8000000000440ULL};
    for (unsigned k = 0; k < sizeof(payloads) / sizeof(payloads[0]); ++k) {
      double x;
      memcpy(&x, &payloads[k], 8);
      ok &= isnan(bc_log1p_exp_neg_tab(x, tab)) && isnan(bc_exp_tab(x, tab));
    }
  }
  ok &= bc_log1p_exp_neg_tab(800.0, tab) == 0.0;
  ok &= bc_exp_tab_nonpos(-1e300, tab) == 0.0 && isnan(bc_exp_tab_nonpos(NAN, tab)) && bc_exp_tab_nonpos(0.0, tab) == 1.0 && bc_exp_tab_nonpos(-INFINITY, tab) == 0.0;
  ok &= bc_log1p_exp_neg_tab(20000.0, tab) == 0.0;
  ok &= bc_exp_tab(-1e300, tab) == 0.0 && isinf(bc_exp_tab(1e300, tab)) && bc_exp_tab(-INFINITY, tab) == 0.0;
  ok &= bc_log1p_exp_neg_tab(50.0, tab) == exp(-50.0) || fabs(bc_log1p_exp_neg_tab(50.0, tab) / exp(-50.0) - 1.0) < 4e-16;
  printf("special %s\n", ok ? "ok" : "BAD");
  return ok ? 0 : 1;
}
