/* Plain-C consumer of include/beta_cores.h: proves the boundary is a C ABI (no C++/Python/torch needed).
 * Without a GPU (argv[1] absent) it only checks that every entry point links and that argument validation
 * happens before any device work.  With "run" it builds a 10-point GIGA coreset on device 0 and prints it.
 *   gcc -std=c99 -Iinclude tests/c_abi_smoke.c -Lbeta_cores_amd -lbeta_cores -Wl,-rpath,beta_cores_amd -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "beta_cores.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != BC_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bc_last_error()); return 1; } \
  } while (0)

static double lcg(unsigned long long* s) {           /* deterministic pseudo-random in (-1, 1) */
  *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((double)((*s >> 11) & 0xFFFFFFFFFFFFFULL) / 4503599627370496.0) * 2.0 - 1.0;
}

int main(int argc, char** argv) {
  /* every symbol of the header is referenced so that the link fails if one is missing */
  void* syms[] = {(void*)bc_version, (void*)bc_last_error, (void*)bc_ctx_create, (void*)bc_ctx_destroy, (void*)bc_ctx_sync,
                  (void*)bc_ctx_kernel_time, (void*)bc_ctx_kernel_time_reset, (void*)bc_ctx_enable_timing,
                  (void*)bc_data_from_host, (void*)bc_data_from_device, (void*)bc_data_create, (void*)bc_data_upload,
                  (void*)bc_data_gather_rows, (void*)bc_data_zero_feature_keys, (void*)bc_ctx_set_constant_row_values, (void*)bc_data_destroy, (void*)bc_phi_from_host, (void*)bc_phi_create,
                  (void*)bc_project, (void*)bc_project_grad_x, (void*)bc_phi_shape, (void*)bc_phi_colsum, (void*)bc_phi_norms, (void*)bc_phi_norm_stats,
                  (void*)bc_phi_to_host, (void*)bc_phi_gather_rows, (void*)bc_phi_group_sum, (void*)bc_phi_matvec, (void*)bc_phi_destroy,
                  (void*)bc_phi_argmax, (void*)bc_snnls_create, (void*)bc_snnls_destroy, (void*)bc_snnls_prefilter_active, (void*)bc_snnls_prefilter_form,
                  (void*)bc_snnls_prefilter_fallbacks, (void*)bc_snnls_prefilter_stats, (void*)bc_snnls_prefilter_levels,
                  (void*)bc_snnls_set_tolerance, (void*)bc_snnls_bind_exchange, (void*)bc_snnls_record_doubles,
                  (void*)bc_snnls_build_begin, (void*)bc_snnls_step_local, (void*)bc_snnls_step_local_exact, (void*)bc_snnls_step_finish, (void*)bc_snnls_build_end, (void*)bc_snnls_select_local_exact,
                  (void*)bc_snnls_build, (void*)bc_snnls_select, (void*)bc_snnls_select_local, (void*)bc_snnls_select_pick,
                  (void*)bc_snnls_reweight, (void*)bc_snnls_error, (void*)bc_snnls_size, (void*)bc_snnls_weights,
                  (void*)bc_snnls_set_weights, (void*)bc_snnls_columns, (void*)bc_snnls_reset, (void*)bc_snnls_get_flags,
                  (void*)bc_snnls_set_flags, (void*)bc_snnls_trace, (void*)bc_weighted_gram, (void*)bc_comm_load, (void*)bc_comm_unique_id, (void*)bc_comm_create,
                  (void*)bc_comm_destroy, (void*)bc_comm_info, (void*)bc_comm_all_gather, (void*)bc_comm_selftest, (void*)bc_comm_precheck, (void*)bc_comm_abort, (void*)bc_comm_sum_doubles, (void*)bc_phi_colsum_all,
                  (void*)bc_project_colsum, (void*)bc_vi_gradient, (void*)bc_vi_gradient_begin, (void*)bc_vi_gradient_end, (void*)bc_ctx_phase_times, (void*)bc_comm_rank_order_sum_selftest, (void*)bc_weighted_gram_host, (void*)bc_ctx_timing_classes,
                  (void*)bc_snnls_bind_comm, (void*)bc_project_from_host};
  printf("abi %d, %d entry points\n", bc_version(), (int)(sizeof(syms) / sizeof(syms[0])));
  if (bc_ctx_sync(NULL) != BC_INVALID_ARGUMENT || bc_snnls_build(NULL, 1, NULL) != BC_INVALID_ARGUMENT) return 2;
  if (argc < 2 || strcmp(argv[1], "run") != 0) return 0;

  enum { N = 5000, D = 6, S = 40, M = 10 };
  unsigned long long seed = 42;
  double* z = (double*)malloc(sizeof(double) * N * (D + 1));
  double theta[S * D], b[S], sig[1] = {1.0};
  for (int i = 0; i < N * (D + 1); ++i) z[i] = lcg(&seed);
  for (int i = 0; i < S * D; ++i) theta[i] = 0.5 * lcg(&seed);
  bc_ctx* ctx = NULL;
  bc_data* data = NULL;
  bc_phi* phi = NULL;
  bc_snnls* sv = NULL;
  CHECK(bc_ctx_create(0, NULL, &ctx));
  CHECK(bc_data_from_host(ctx, z, N, D + 1, &data));
  CHECK(bc_project(ctx, data, BC_MODEL_LINREG_LL, theta, S, sig, 1, 0, &phi));     /* K1 + K2 */
  CHECK(bc_phi_colsum(phi, b));
  double b2[S];
  CHECK(bc_project_colsum(ctx, data, BC_MODEL_LINREG_LL, theta, S, sig, 1, NULL, b2));    /* store-free K1: the same bits */
  if (memcmp(b, b2, sizeof(b)) != 0) { printf("store-free column sums differ from the materialised ones\n"); return 4; }
  {
    /* the host-array entry point (upload + K1 pipelined): the same column sums, bit for bit */
    bc_data* data3 = NULL;
    bc_phi* phi3 = NULL;
    double b3[S];
    CHECK(bc_project_from_host(ctx, z, N, D + 1, BC_MODEL_LINREG_LL, theta, S, sig, 1, 0, &data3, &phi3));
    CHECK(bc_phi_colsum(phi3, b3));
    if (memcmp(b, b3, sizeof(b)) != 0) { printf("bc_project_from_host column sums differ from bc_project's\n"); return 5; }
    CHECK(bc_phi_destroy(phi3));
    CHECK(bc_data_destroy(data3));
  }
  double nsum = 0.0;
  int64_t nzero = 0;
  CHECK(bc_phi_norm_stats(phi, &nzero, &nsum));
  CHECK(bc_snnls_create(ctx, phi, b, BC_ALG_GIGA, nsum, 0, &sv));
  int limit = 0;
  CHECK(bc_snnls_build(sv, M, &limit));                                             /* fused greedy loop */
  int64_t idx[64], n = 0;
  double val[64], err = 0.0, bnorm = 0.0;
  CHECK(bc_snnls_weights(sv, 64, idx, val, &n));
  CHECK(bc_snnls_error(sv, &err));
  for (int k = 0; k < S; ++k) bnorm += b[k] * b[k];
  printf("coreset of %lld points, error %.6e (||b|| = %.6e), numeric limit %d\n", (long long)n, err, sqrt(bnorm), limit);
  for (int j = 0; j < (int)n; ++j) printf("  row %lld  w = %.12g\n", (long long)idx[j], val[j]);
  int ok = n > 0 && n <= M && err < sqrt(bnorm) && !limit;
  CHECK(bc_snnls_destroy(sv));
  CHECK(bc_phi_destroy(phi));
  CHECK(bc_data_destroy(data));
  CHECK(bc_ctx_destroy(ctx));
  free(z);
  return ok ? 0 : 3;
}
