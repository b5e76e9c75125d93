/* Host-side check of beta_cores_amd/csrc/bc_np_exp.h (the header the K1 kernel uses for the constant rows of the
 * beta-likelihood models): reads n arguments and n expected results (raw doubles) and counts bit mismatches.
 * Built and driven by tests/test_np_exp_cpu.py. */
#include <stdio.h>
#include <stdlib.h>
#include "bc_np_exp.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  const long n = atol(argv[2]);
  double* x = (double*)malloc((size_t)n * 8);
  double* y = (double*)malloc((size_t)n * 8);
  if (fread(x, 8, (size_t)n, f) != (size_t)n || fread(y, 8, (size_t)n, f) != (size_t)n) return 2;
  fclose(f);
  long bad = 0, not_covered = 0;
  for (long i = 0; i < n; ++i) {
    int cov;
    const double e = bc_np_exp(x[i], &cov);
    if (!cov) { not_covered++; continue; }
    if (bc_d2bits(e) != bc_d2bits(y[i])) {
      if (bad < 5) printf("x = %.17g: got %.17g, numpy %.17g\n", x[i], e, y[i]);
      bad++;
    }
  }
  printf("n=%ld mismatches=%ld not_covered=%ld\n", n, bad, not_covered);
  free(x);
  free(y);
  return bad ? 1 : 0;
}
