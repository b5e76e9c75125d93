// Does the size of the contiguous chunk a wave reads before it jumps change the streaming rate?  2048 resident waves (8 per CU)
// read a 1.04 GB buffer with non-temporal dwordx4 loads, 5 x 1 KB in flight per wave and buffer (the int8 sweep's pattern); a wave
// reads `chunk` KB contiguously, then jumps by (waves x chunk).  chunk = 26 is the int8 sweep's tile, 102 the fp64 sweep's.
//   hipcc --offload-arch=gfx950 -O2 -o tools/stream_chunk_probe.bin tools/stream_chunk_probe.hip && tools/stream_chunk_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const i4* __restrict__ p, size_t n_kb, int chunk_kb, int* out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  int acc = 0;
  for (size_t c0 = wave * chunk_kb; c0 < n_kb; c0 += nwaves * chunk_kb) {
    const i4* q = p + c0 * 64 + lane;                      // 1 KB = 64 lanes x 16 B
    const int len = (int)((n_kb - c0) < (size_t)chunk_kb ? (n_kb - c0) : (size_t)chunk_kb);
    i4 x[5], y[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) x[u] = __builtin_nontemporal_load(q + (size_t)(u < len ? u : 0) * 64);
    for (int k = 0; k < len; k += 5) {
#pragma unroll
      for (int u = 0; u < 5; ++u) y[u] = __builtin_nontemporal_load(q + (size_t)(k + 5 + u < len ? k + 5 + u : 0) * 64);
#pragma unroll
      for (int u = 0; u < 5; ++u) acc += x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
#pragma unroll
      for (int u = 0; u < 5; ++u) x[u] = y[u];
    }
  }
  if (acc == 0x12345678) out[0] = acc;
}

int main() {
  const size_t n_kb = 527280;           // ~0.54 GB (the 4-bit mirror at N = 10M)
  i4* p; int* out;
  hipMalloc(&p, n_kb * 1024); hipMalloc(&out, 4);
  hipMemset(p, 1, n_kb * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int chunks[] = {13, 25, 26, 52, 13, 104, 25};
  for (int ci = 0; ci < 7; ++ci) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_read, dim3(512), dim3(256), 0, 0, p, n_kb, chunks[ci], out);
    hipEventRecord(a);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k_read, dim3(512), dim3(256), 0, 0, p, n_kb, chunks[ci], out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("chunk %3d KB: %.1f us per pass, %.2f TB/s\n", chunks[ci], 1e3 * ms / 20, n_kb * 1024.0 / (ms / 20 * 1e-3) / 1e12);
  }
  return 0;
}
