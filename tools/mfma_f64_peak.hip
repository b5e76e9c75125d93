// fp64 matrix-core ceiling on MI355X (gfx950): v_mfma_f64_16x16x4_f64 in a register-only loop, with the
// in-kernel clock measured beside it (s_memtime = shader cycles, s_memrealtime = 100 MHz), so that the
// achieved TFLOP/s can be split into "cycles per MFMA per SIMD" (pipe rate; 64 = specified) and "clock held
// under this load".  DESIGN.md section 4 prices K1 (the projection kernel) against the number this prints.
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak.bin && tools/mfma_f64_peak.bin
//
// Variants: NACC independent accumulators per wave (K1 holds 12), 1 / 2 / 3 / 4 waves per SIMD (K1 runs 2),
// operands from registers or re-read from LDS before every MFMA (K1's inner loop reads both operands from LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS_OPS>
__global__ __launch_bounds__(256) void k_mfma(double* out, unsigned long long* stamps, int iters, double a0, double b0) {
  __shared__ double ops[2][NACC][64 * 4 + 8];
  double4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0., 0., 0., 0.};
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double a = a0 + 1e-3 * threadIdx.x, b = b0 + 5e-4 * threadIdx.x;
  if (LDS_OPS) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      ops[0][i][w * 64 + lane] = a + i;
      ops[1][i][w * 64 + lane] = b - i;
    }
    __syncthreads();
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (LDS_OPS) {
        const double av = ops[0][i][w * 64 + lane], bv = ops[1][i][w * 64 + lane];
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
      } else {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

// the other fp64 matrix instruction: v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks per instruction, 512 flop)
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(double* out, unsigned long long* stamps, int iters, double a0, double b0) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.;
  double a = a0 + 1e-3 * threadIdx.x, b = b0 + 5e-4 * threadIdx.x;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

// the vector pipe for comparison: v_fma_f64, 32 independent chains per lane
__global__ __launch_bounds__(256) void k_valu(double* out, unsigned long long* stamps, int iters, double a0, double b0) {
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 1e-3 * i;
  const double a = a0 + 1e-9 * threadIdx.x, b = b0 * 1e-3;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

static double median(std::vector<double>& v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

template <typename L>
static void run(const char* tag, int n_cu, int blocks_per_cu, double flop_per_wave_iter, double pipe_cycles_per_wave_iter, int iters, L launch) {
  const int grid = n_cu * blocks_per_cu;
  double* out;
  unsigned long long* st;
  hipMalloc(&out, (size_t)grid * 256 * sizeof(double));
  hipMalloc(&st, (size_t)grid * 2 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch(grid, out, st, iters / 8);                  // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(grid, out, st, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)grid * 2);
  hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int b = 0; b < grid; ++b) {
    clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);       // GHz: cycles per 10 ns tick
    cyc.push_back((double)h[2 * b]);
  }
  const double ghz = median(clk), cycles = median(cyc);
  const double flops = (double)grid * 4 * iters * flop_per_wave_iter;
  // waves per SIMD = blocks_per_cu (each 256-thread block puts one wave on every SIMD)
  const double pipe_util = pipe_cycles_per_wave_iter * iters * blocks_per_cu / cycles;
  printf("%-34s waves/SIMD=%d: %8.3f ms  %6.1f TF  in-kernel clock %.2f GHz  pipe busy %.1f %% of the loop's cycles\n", tag,
         blocks_per_cu, ms, flops / ms / 1e9, ghz, 100. * pipe_util);
  hipFree(out);
  hipFree(st);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.gcnArchName, n_cu);
  const int it = 40000;
#define MF(NACC, LDS, BPC, TAG)                                                                              \
  run(TAG, n_cu, BPC, NACC * 2048.0, NACC * 64.0, it, [](int g, double* o, unsigned long long* s, int n) {     \
    hipLaunchKernelGGL((k_mfma<NACC, LDS>), dim3(g), dim3(256), 0, 0, o, s, n, 1.0, 2.0);                      \
  })
  MF(12, false, 1, "mfma_f64_16x16x4, 12 acc, regs");
  MF(12, false, 2, "mfma_f64_16x16x4, 12 acc, regs");
  MF(12, false, 3, "mfma_f64_16x16x4, 12 acc, regs");
  MF(8, false, 4, "mfma_f64_16x16x4,  8 acc, regs");
  MF(12, true, 1, "mfma_f64_16x16x4, 12 acc, LDS ops");
  MF(12, true, 2, "mfma_f64_16x16x4, 12 acc, LDS ops");
  MF(12, true, 3, "mfma_f64_16x16x4, 12 acc, LDS ops");
  for (int bpc = 1; bpc <= 4; bpc *= 2)
    run("mfma_f64_4x4x4_4b, 16 acc, regs", n_cu, bpc, 16 * 512.0, 16 * 16.0, it * 4, [](int g, double* o, unsigned long long* s, int n) {
      hipLaunchKernelGGL((k_mfma4<16>), dim3(g), dim3(256), 0, 0, o, s, n, 1.0, 2.0);
    });
#define MF4(NACC, BPC)                                                                                               \
  run("mfma_f64_4x4x4_4b, " #NACC " acc (dependent chains)", n_cu, BPC, NACC * 512.0, NACC * 16.0, it * 4, \
      [](int g, double* o, unsigned long long* s, int n) { hipLaunchKernelGGL((k_mfma4<NACC>), dim3(g), dim3(256), 0, 0, o, s, n, 1.0, 2.0); })
  MF4(1, 1); MF4(2, 1); MF4(4, 1); MF4(8, 1); MF4(2, 2); MF4(4, 2);
  for (int bpc = 1; bpc <= 4; bpc *= 2)
    run("v_fma_f64, 32 chains", n_cu, bpc, 32 * 64 * 2.0, 32 * 4.0, it, [](int g, double* o, unsigned long long* s, int n) {
      hipLaunchKernelGGL(k_valu, dim3(g), dim3(256), 0, 0, o, s, n, 0.999, 1.0);
    });
  return 0;
}
