// How long does a kernel wait for its kernel arguments, and does `-mllvm -amdgpu-kernarg-preload-count=N` (the CP puts the first
// N dwords of the scalar / pointer arguments into SGPRs before the wave starts) remove that wait on this part?
//   hipcc --offload-arch=gfx950 -O2 -o tools/kernarg_preload_probe.bin tools/kernarg_preload_probe.hip                                  (plain)
//   hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-kernarg-preload-count=8 -o tools/kernarg_preload_probe_pl.bin tools/kernarg_preload_probe.hip
// Each round: a 1 GB streaming kernel (cools caches and TLBs the way a sweep does), then the probe; prints the mean ticks from
// the first instruction to the arrival of the first load that needs a pointer argument.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_stream(const double* __restrict__ a, double* out, size_t n) {
  double acc = 0.;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(a + i);
  if (acc == 1.2345) out[0] = acc;
}

__global__ void k_probe(const double* __restrict__ p, unsigned long long* __restrict__ out, int slot) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const double v = p[threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(v) : "memory");       // (the input ties the wait behind the load)
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (v == 1.2345) t1 += 1;
  if (threadIdx.x == 0) out[slot] = t1 - t0;
}

int main() {
  const size_t n = (size_t)1 << 27;   // 1 GiB of doubles
  double *a, *small;
  unsigned long long* out;
  hipMalloc(&a, n * 8);
  hipMalloc(&small, 4096);
  hipMalloc(&out, 64 * 8);
  hipMemset(a, 0, n * 8);
  hipMemset(small, 0, 4096);
  for (int r = 0; r < 40; ++r) {
    hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, a, small, n);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, small, out, r);
  }
  hipDeviceSynchronize();
  unsigned long long h[64];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int r = 8; r < 40; ++r) s += (double)h[r];
  printf("mean ticks from kernel start to the first pointer-dependent load's arrival: %.0f  (first rounds: %llu %llu %llu)\n", s / 32, h[0], h[1], h[2]);
  return 0;
}
