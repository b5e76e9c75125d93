// Does v_mfma_f64_16x16x4_f64 issue slower when consecutive instructions name the same A / B source registers?
// (tools/mfma_f64_peak.hip, one (a, b) pair for all 12 accumulators: ~104 cycles per MFMA; tools/mfma_valu_coissue.hip,
// 12 distinct pairs: 64.0.)  One wave per SIMD, 12 accumulators, register-only loop; patterns:
//   0  12 distinct A, 12 distinct B                      3  K1 order transposed: (a[st], b[jt]) jt-outer
//   1  one A, one B for all                              4  12 distinct A, one B
//   2  K1's order: (a[st], b[jt]) st-outer, jt-inner     5  one A, 12 distinct B
//   6  K1's operands, order chosen so that neighbours share neither A nor B where possible
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_operand_reuse.hip -o tools/mfma_f64_operand_reuse.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* stamps, int iters, double a0) {
  double4_t acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = (double4_t){0., 0., 0., 0.};
  double a[12], b[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    a[i] = a0 + 1e-3 * threadIdx.x + i;
    b[i] = 2. + 5e-4 * threadIdx.x - i;
    asm volatile("" : "+v"(a[i]), "+v"(b[i]));      // 24 live registers, not folded into one
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      int ia, ib;
      if (P == 0) { ia = i; ib = i; }
      else if (P == 1) { ia = 0; ib = 0; }
      else if (P == 2) { ia = i >> 1; ib = i & 1; }
      else if (P == 3) { ia = i % 6; ib = i / 6; }
      else if (P == 4) { ia = i; ib = 0; }
      else if (P == 5) { ia = 0; ib = i; }
      else { ia = i % 6; ib = (i + i / 6) & 1; }     // a0b0 a1b1 a2b0 a3b1 a4b0 a5b1 | a0b1 a1b0 ...
      const int ic = (P == 2) ? i : (P == 3) ? (i % 6) * 2 + i / 6 : (P == 6) ? (i % 6) * 2 + ((i + i / 6) & 1) : i;
      acc[ic] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ia], b[ib], acc[ic], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  double res = 0.;
#pragma unroll
  for (int i = 0; i < 12; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int P>
static void run(const char* tag, int n_cu, int bpc) {
  const int iters = 4000;
  double* out;
  unsigned long long* st;
  const int grid = n_cu * bpc;
  (void)hipMalloc(&out, (size_t)grid * 256 * sizeof(double));
  (void)hipMalloc(&st, (size_t)grid * 4 * sizeof(unsigned long long));
  hipLaunchKernelGGL(k<P>, dim3(grid), dim3(256), 0, 0, out, st, iters / 4, 1.0);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k<P>, dim3(grid), dim3(256), 0, 0, out, st, iters, 1.0);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)grid * 4);
  (void)hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> m(h.begin(), h.end());
  std::sort(m.begin(), m.end());
  const double cyc = m[m.size() / 2];
  printf("%-58s %d wave(s)/SIMD: %6.1f cycles per MFMA per wave  (SIMD issues one every %5.1f)\n", tag, bpc, cyc / (12.0 * iters), cyc / (12.0 * iters) / bpc);
  (void)hipFree(out);
  (void)hipFree(st);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int n = p.multiProcessorCount;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run<0>("0: 12 distinct A, 12 distinct B", n, bpc);
    run<1>("1: one A, one B", n, bpc);
    run<2>("2: K1 order (A shared by neighbours, B alternates)", n, bpc);
    run<3>("3: jt-outer (B shared by six neighbours)", n, bpc);
    run<4>("4: 12 distinct A, one B", n, bpc);
    run<5>("5: one A, 12 distinct B", n, bpc);
    run<6>("6: K1 operands, neighbours share neither A nor B", n, bpc);
  }
  return 0;
}
