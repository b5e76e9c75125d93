// Operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by probing: one-hot A and B lanes,
// which D lanes light up.  Prints, for every D lane, the (A lane, B lane) pairs that feed it.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_4x4x4_layout.hip -o tools/mfma_f64_4x4x4_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out) {   // out[la][lb][64]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1. : 0., b = lane == lb ? 1. : 0.;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., 0, 0, 0);
      out[((size_t)la * 64 + lb) * 64 + lane] = d;
    }
}
int main() {
  double* d;
  hipMalloc(&d, 64 * 64 * 64 * sizeof(double));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(64 * 64 * 64);
  hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  for (int ld = 0; ld < 64; ++ld) {
    printf("D lane %2d <-", ld);
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb)
        if (h[((size_t)la * 64 + lb) * 64 + ld] != 0.) printf(" (a%d,b%d)", la, lb);
    printf("\n");
  }
  return 0;
}
