/* Host-side check of beta_cores_amd/csrc/bc_np_sum.h: reads triples (n, c, expected sum) as raw doubles and counts
 * bit mismatches of bc_np_sum_const_any (and of bc_np_sum_const_256 for n <= 256).  Driven by tests/test_np_sum_cpu.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bc_np_sum.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  const long m = atol(argv[2]);
  double* t = (double*)malloc((size_t)m * 3 * 8);
  if (fread(t, 8, (size_t)m * 3, f) != (size_t)m * 3) return 2;
  fclose(f);
  long bad = 0;
  for (long i = 0; i < m; ++i) {
    const int n = (int)t[3 * i];
    const double c = t[3 * i + 1], want = t[3 * i + 2];
    const double got = bc_np_sum_const_any(c, n);
    int ok = memcmp(&got, &want, 8) == 0;
    if (n <= 256) {
      const double g2 = bc_np_sum_const_256(c, n);
      ok = ok && memcmp(&g2, &want, 8) == 0;
    }
    if (!ok) {
      if (bad < 5) printf("n = %d c = %.17g: got %.17g, numpy %.17g\n", n, c, got, want);
      bad++;
    }
  }
  printf("checked=%ld mismatches=%ld\n", m, bad);
  free(t);
  return bad ? 1 : 0;
}
