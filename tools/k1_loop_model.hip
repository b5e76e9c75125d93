// K1's contraction loop in isolation: 4 waves per block, every wave 12 x v_mfma_f64_16x16x4_f64 + 2 x v_mfma_f64_4x4x4_4b_f64
// per k-step, operands read from LDS in K1's layout ([128][33] Z slab, [100][34] Theta slab), accumulators in VGPRs.
// What does the loop lose against the matrix pipe's 64 + 16 cycles per instruction, and to what?
//   V: 0 = K1's loop (operand reads at the top of each 2-k-step body, two barriers per 8 k-steps)
//      1 = same without the barriers          2 = operands from registers (no LDS reads, no barriers)
//      3 = K1's loop, next body's operands requested before this body's MFMAs (explicit double buffer)
//   hipcc -O3 --offload-arch=gfx950 tools/k1_loop_model.hip -o tools/k1_loop_model.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define LDZ 33
#define LDT 34
#define NT 6

template <int V>
__global__ __launch_bounds__(256, 2) void k(double* out, unsigned long long* stamps, int chunks, double a0) {
  extern __shared__ double lds[];
  double* Zl = lds;
  double* Tl = lds + 128 * LDZ;
  for (int i = threadIdx.x; i < 128 * LDZ + 100 * LDT; i += 256) lds[i] = a0 + 1e-3 * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int row_base = 32 * w + 2 * j;
  double4_t acc[2][NT];
  double tv[2] = {0., 0.};
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int st = 0; st < NT; ++st) acc[jt][st] = (double4_t){0., 0., 0., 0.};
  const double* zrow0 = Zl + row_base * LDZ + g;
  const double* trow = Tl + j * LDT + g;
  const double* tquad = Tl + (NT * 16 + (j & 3)) * LDT + g;
  double bz[2][2], at[2][NT + 1];
  auto fetch = [&](int kk, int buf) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) bz[buf][jt] = zrow0[jt * LDZ + kk * 4];
#pragma unroll
    for (int st = 0; st < NT; ++st) at[buf][st] = trow[st * 16 * LDT + kk * 4];
    at[buf][NT] = tquad[kk * 4];
  };
  auto mm = [&](int buf) {
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) acc[jt][st] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[buf][st], bz[buf][jt], acc[jt][st], 0, 0, 0);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) tv[jt] = __builtin_amdgcn_mfma_f64_4x4x4f64(at[buf][NT], bz[buf][jt], tv[jt], 0, 0, 0);
  };
  if (V == 2) { fetch(0, 0); fetch(1, 1); }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < chunks; ++c) {
    if (V == 0 || V == 3) __syncthreads();
    if (V == 3) {
      fetch(0, 0);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        if (kk + 1 < 8) fetch(kk + 1, (kk + 1) & 1);
        mm(kk & 1);
      }
    } else {
#pragma unroll 1
      for (int kp = 0; kp < 4; ++kp) {
        if (V != 2) { fetch(2 * kp, 0); fetch(2 * kp + 1, 1); }
        mm(0);
        mm(1);
      }
    }
    if (V == 0 || V == 3) __syncthreads();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  double res = tv[0] + tv[1];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int st = 0; st < NT; ++st) res += acc[jt][st][0] + acc[jt][st][1] + acc[jt][st][2] + acc[jt][st][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (lane == 0) stamps[(size_t)blockIdx.x * 4 + w] = c1 - c0;
}

template <int V>
static void run(const char* tag, int n_cu, int bpc) {
  const int chunks = 2000;
  const int grid = n_cu * bpc;
  const size_t ldsb = (128 * LDZ + 100 * LDT) * sizeof(double);
  double* out;
  unsigned long long* st;
  (void)hipMalloc(&out, (size_t)grid * 256 * sizeof(double));
  (void)hipMalloc(&st, (size_t)grid * 4 * sizeof(unsigned long long));
  (void)hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), ldsb, 0, out, st, chunks / 4, 1.0);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), ldsb, 0, out, st, chunks, 1.0);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)grid * 4);
  (void)hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> m(h.begin(), h.end());
  std::sort(m.begin(), m.end());
  const double per_kstep = m[m.size() / 2] / (8.0 * chunks);
  const double pipe = 12 * 64 + 2 * 16;
  printf("%-64s %d block(s)/CU: %7.1f cycles per k-step per wave; matrix pipe busy %5.1f %%\n", tag, bpc, per_kstep, 100. * pipe * bpc / per_kstep);
  (void)hipFree(out);
  (void)hipFree(st);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int n = p.multiProcessorCount;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run<0>("0: K1 loop (LDS operands, 2 barriers per 8 k-steps)", n, bpc);
    run<1>("1: no barriers", n, bpc);
    run<2>("2: register operands, no barriers", n, bpc);
    run<3>("3: K1 loop, operands of the next k-step requested first", n, bpc);
  }
  return 0;
}
