// What does ONE dependent global round trip cost a single-block kernel that runs right after a chip-wide streaming
// kernel (the greedy step's situation: a 1 GB sweep, then the one-block step kernel)?
//   chase:  8 dependent loads (each address comes from the previous value) into an 8 GB buffer, s_memtime per hop
//   cases:  after an idle second / right after a 1 GB streaming kernel / the same with a light kernel kept running on a
//           second stream (are the long round trips a clock / power-state effect of an otherwise idle chip?)
//   hipcc -O3 --offload-arch=gfx950 tools/tail_latency.hip -o tools/tail_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>

typedef double bc_d4 __attribute__((ext_vector_type(4)));
__global__ void k_stream(const bc_d4* __restrict__ p, size_t n, double* out) {
  double acc = 0.;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const bc_d4 v = __builtin_nontemporal_load(p + i);
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678) out[0] = acc;
}

__global__ void k_chase(const long long* __restrict__ next, long long start, unsigned long long* stamps, long long* sink) {
  long long i = start;
  stamps[0] = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    i = next[i];
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("" :: "v"(i));
    stamps[h + 1] = __builtin_amdgcn_s_memtime();
  }
  sink[0] = i;
}

__global__ void k_light(const double* __restrict__ p, size_t n, double* out, int iters) {
  double acc = 0.;
  for (int it = 0; it < iters; ++it)
    for (size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678) out[0] = acc;
}

int main() {
  const size_t nchase = (size_t)1 << 30;            // 8 GB of long long
  long long* next;
  if (hipMalloc(&next, nchase * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  // next[i] = a far-away index: fill only the 9 entries the chase visits, stride ~ 900 MB apart
  std::vector<long long> idx(10);
  for (int h = 0; h < 10; ++h) idx[h] = ((long long)h * 113000077LL + 12345) % (long long)nchase;
  for (int h = 0; h < 9; ++h) (void)hipMemcpy(next + idx[h], &idx[h + 1], 8, hipMemcpyHostToDevice);
  bc_d4* big; double* out; unsigned long long* st; long long* sink;
  const size_t nbig = ((size_t)1 << 30) / 32;       // 1 GB
  (void)hipMalloc(&big, nbig * 32); (void)hipMemset(big, 0, nbig * 32);
  (void)hipMalloc(&out, 64); (void)hipMalloc(&st, 16 * 8); (void)hipMalloc(&sink, 8);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  auto report = [&](const char* tag) {
    unsigned long long h[9];
    (void)hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
    printf("%-46s hops:", tag);
    for (int i = 0; i < 8; ++i) printf(" %6llu", h[i + 1] - h[i]);
    printf("  ticks\n");
  };
  for (int rep = 0; rep < 2; ++rep) {
    sleep(1);
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s1, next, idx[0], st, sink); (void)hipStreamSynchronize(s1);
    report("after an idle second");
    hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s1, big, nbig, out);
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s1, next, idx[0], st, sink); (void)hipStreamSynchronize(s1);
    report("right after a 1 GB streaming kernel");
    hipLaunchKernelGGL(k_light, dim3(64), dim3(256), 0, s2, (const double*)big, (size_t)1 << 22, out, 400);
    hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s1, big, nbig, out);
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s1, next, idx[0], st, sink); (void)hipStreamSynchronize(s1);
    report("... with a light kernel running beside it");
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s1, big, nbig, out);
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s1, next, idx[0], st, sink);
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s1, next, idx[0], st, sink); (void)hipStreamSynchronize(s1);
    report("second chase of the same entries (warm TLB/L2)");
  }
  return 0;
}
