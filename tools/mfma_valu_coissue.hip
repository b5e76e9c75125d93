// Do fp64 MFMAs and ordinary vector instructions of ANOTHER wave on the same SIMD overlap on gfx950?
// 512-thread blocks, one per CU: waves 0-3 (one per SIMD) run a v_mfma_f64_16x16x4_f64 loop, waves 4-7 (their SIMD
// partners) run one of: nothing / v_fma_f64 / v_add_u32 / v_fma_f32 / ds_read_b64.  Cycles of each role alone and
// together tell whether the partner's instructions cost the MFMA wave its pipe time (K1's epilogue and staging run
// beside another block's contraction loop).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_coissue.hip -o tools/mfma_valu_coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(double* out, unsigned long long* stamps, int mfma_iters, int valu_iters, int mode, double a0) {
  __shared__ double lds[512 * 4];
  const int w = threadIdx.x >> 6;
  lds[threadIdx.x] = a0 + threadIdx.x;
  lds[threadIdx.x + 512] = a0 - threadIdx.x;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  double res = 0.;
  if (w < 4) {
    if (mfma_iters > 0) {
      double4_t acc[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = (double4_t){0., 0., 0., 0.};
      double a = a0 + 1e-3 * threadIdx.x, b = 2. + 5e-4 * threadIdx.x;
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a + i, b - i, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else if (mode == 1) {            // fp64 vector FMAs
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 1e-3 * i + threadIdx.x;
    const double a = 0.999 + 1e-9 * threadIdx.x, b = 1e-3;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) res += acc[i];
  } else if (mode == 2) {            // integer vector ops
    unsigned acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = i + threadIdx.x;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = (acc[i] + 0x9e3779b9u) ^ (unsigned)it;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) res += (double)acc[i];
  } else if (mode == 3) {            // fp32 vector FMAs
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 1e-3f * i + threadIdx.x;
    const float a = 0.999f, b = 1e-3f;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) res += acc[i];
  } else if (mode == 4) {            // LDS reads
    double acc = 0.;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += lds[(threadIdx.x + 64 * i + it) & 1023];
    }
    res = acc;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 8 + w] = c1 - c0;
}

static void run(const char* tag, int n_cu, int mfma_iters, int valu_iters, int mode) {
  double* out;
  unsigned long long* st;
  (void)hipMalloc(&out, (size_t)n_cu * 512 * sizeof(double));
  (void)hipMalloc(&st, (size_t)n_cu * 8 * sizeof(unsigned long long));
  hipLaunchKernelGGL(k, dim3(n_cu), dim3(512), 0, 0, out, st, mfma_iters / 4, valu_iters / 4, mode, 1.0);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k, dim3(n_cu), dim3(512), 0, 0, out, st, mfma_iters, valu_iters, mode, 1.0);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)n_cu * 8);
  (void)hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> m, v;
  for (int b = 0; b < n_cu; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? m : v).push_back((double)h[(size_t)b * 8 + w]);
  std::sort(m.begin(), m.end());
  std::sort(v.begin(), v.end());
  printf("%-44s MFMA waves %9.0f cycles (%5.1f per MFMA)   partner waves %9.0f cycles (%5.2f per instruction)\n", tag,
         m[m.size() / 2], mfma_iters ? m[m.size() / 2] / (12.0 * mfma_iters) : 0., v[v.size() / 2],
         valu_iters ? v[v.size() / 2] / (16.0 * valu_iters) : 0.);
  (void)hipFree(out);
  (void)hipFree(st);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  const int MI = 4000;
  run("MFMA alone", n_cu, MI, 0, 0);
  const char* names[5] = {"", "v_fma_f64", "v_add_u32/v_xor", "v_fma_f32", "ds_read_b64"};
  const int vi[5] = {0, 60000, 120000, 120000, 30000};
  for (int mode = 1; mode <= 4; ++mode) {
    char tag[96];
    snprintf(tag, sizeof tag, "%s alone", names[mode]);
    run(tag, n_cu, 0, vi[mode], mode);
    snprintf(tag, sizeof tag, "MFMA  +  %s on the partner wave", names[mode]);
    run(tag, n_cu, MI, vi[mode], mode);
  }
  return 0;
}
