/* Host-side check of beta_cores_amd/csrc/bc_np_pow2.h against np.power(2., y) (tests/test_np_pow2_cpu.py writes the
 * exponents and NumPy's results to a file): prints mismatches=<n> not_covered=<n>. */
#include <stdio.h>
#include <stdlib.h>
#include "bc_np_pow2.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  const long n = atol(argv[2]);
  double* y = (double*)malloc(sizeof(double) * (size_t)n);
  double* w = (double*)malloc(sizeof(double) * (size_t)n);
  if (fread(y, 8, (size_t)n, f) != (size_t)n || fread(w, 8, (size_t)n, f) != (size_t)n) return 2;
  fclose(f);
  long bad = 0, nc = 0;
  for (long i = 0; i < n; ++i) {
    int cov;
    const double g = bc_np_pow2(y[i], &cov);
    if (!cov) { ++nc; continue; }
    if (memcmp(&g, &w[i], 8) != 0) {
      if (bad < 5) printf("y = %.17g: got %.17g want %.17g\n", y[i], g, w[i]);
      ++bad;
    }
  }
  printf("mismatches=%ld not_covered=%ld\n", bad, nc);
  free(y); free(w);
  return bad ? 1 : 0;
}
