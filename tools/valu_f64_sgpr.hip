// Microbenchmark for the K1 vector-FMA formulation: lane = 4 data rows, 50 samples per wave,
// theta wave-uniform (scalar loads), x from LDS.  acc[4][50] += x[r][d] * theta[d][s].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int R, int SW>
__global__ __launch_bounds__(256, 2) void k(const double* __restrict__ theta, const double* __restrict__ xin, double* out, int D, int reps) {
  __shared__ double xl[64 * R * 33];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double acc[R][SW];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int s = 0; s < SW; ++s) acc[r][s] = 0.;
  for (int i = threadIdx.x; i < 64 * R * 33; i += 256) xl[i] = xin[i % 1024];
  __syncthreads();
  const double* th = theta + (w & 1) * SW;     // two sample groups (wave-uniform)
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 2
    for (int d = 0; d < D; ++d) {
      double x[R];
#pragma unroll
      for (int r = 0; r < R; ++r) x[r] = xl[(lane * R + r) * 33 + (d & 31)];
      const double* t = th + (size_t)d * (2 * SW);
#pragma unroll
      for (int s = 0; s < SW; ++s) {
        const double tv = t[s];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][s] = __builtin_fma(x[r], tv, acc[r][s]);
      }
    }
  }
  double sum = 0;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int s = 0; s < SW; ++s) sum += acc[r][s];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <int R, int SW>
void run(int bpc) {
  const int D = 128, reps = 40;
  std::vector<double> th(D * 2 * SW, 0.001), x(1024, 0.5);
  double *dth, *dx, *out;
  (void)hipMalloc(&dth, th.size() * 8); (void)hipMalloc(&dx, 8192); (void)hipMalloc(&out, 256 * 8 * 256 * 8);
  (void)hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dx, x.data(), 8192, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int grid = 256 * bpc;
  k<R, SW><<<grid, 256>>>(dth, dx, out, D, 2);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<R, SW><<<grid, 256>>>(dth, dx, out, D, reps);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)grid * 256 * R * SW * 2.0 * D * reps;
  printf("R=%d SW=%d blocks/CU=%d: %.3f ms  %.1f TF\n", R, SW, bpc, ms, fl / ms / 1e9);
}
int main() { run<4, 25>(2); run<8, 13>(1); run<8, 13>(2); run<8, 12>(2); run<6, 17>(2); return 0; }
