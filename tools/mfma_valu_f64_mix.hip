// Microbenchmark: fp64 throughput of v_mfma_f64_16x16x4_f64 alone, v_fma_f64 alone, and both
// interleaved in one wave (register-only, no memory).  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NM, int NV>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  double4_t acc[NM > 0 ? NM : 1];
  double v[NV > 0 ? NV : 1];
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = (double4_t){0., 0., 0., 0.};
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = i * 0.5;
  double a = a0 + threadIdx.x * 1e-3, b = b0 + threadIdx.x * 0.5e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_fma(v[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NV>
void run(int blocks_per_cu) {
  double* out;
  (void)hipMalloc(&out, 256 * 8 * 4096 * 8);
  int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  k<NM, NV><<<grid, 256>>>(out, 10, 1.0, 2.0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NM, NV><<<grid, 256>>>(out, iters, 1.0, 2.0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double fm = (double)grid * 4 * iters * NM * 2048.0, fv = (double)grid * 256 * (double)iters * NV * 2.0;
  printf("NM=%2d NV=%2d blocks/CU=%d: %.3f ms  mfma %.1f TF + valu %.1f TF = %.1f TF\n", NM, NV, blocks_per_cu, ms,
         fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
  (void)hipFree(out);
}
int main() {
  run<14, 0>(2);
  run<0, 32>(2);
  run<0, 32>(4);
  run<14, 8>(2);
  run<14, 16>(2);
  run<14, 28>(2);
  run<14, 56>(2);
  run<7, 28>(4);
  return 0;
}
