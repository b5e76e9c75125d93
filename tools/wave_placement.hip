// Where do the waves of a workgroup land?  Launches `grid` blocks of `nw` waves whose kernel holds ~150 VGPRs (so that a SIMD
// takes three of them) and prints, for a few blocks, the SIMD each wave ran on (HW_REG_HW_ID), and how many blocks shared a CU.
//   hipcc --offload-arch=gfx950 -O2 -o tools/wave_placement.bin tools/wave_placement.hip && tools/wave_placement.bin 9 256
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void k_place(unsigned* out, int spin, float* sink) {
  // hold ~150 registers live across the spin loop
  float r[140];
#pragma unroll
  for (int i = 0; i < 140; ++i) r[i] = (float)(threadIdx.x + i);
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  for (int it = 0; it < spin; ++it) {
#pragma unroll
    for (int i = 0; i < 140; ++i) r[i] = r[i] * 1.0001f + r[(i + 1) % 140];
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 140; ++i) acc += r[i];
  if (acc == 12345.678f) sink[0] = acc;
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    out[(blockIdx.x * nw + w) * 2] = hwid;
    out[(blockIdx.x * nw + w) * 2 + 1] = xcc;
  }
}

int main(int argc, char** argv) {
  const int nw = argc > 1 ? atoi(argv[1]) : 9, grid = argc > 2 ? atoi(argv[2]) : 256;
  unsigned* d;
  float* sink;
  hipMalloc(&d, (size_t)grid * nw * 2 * sizeof(unsigned));
  hipMalloc(&sink, 4);
  hipLaunchKernelGGL(k_place, dim3(grid), dim3(64 * nw), 0, 0, d, 20000, sink);
  hipDeviceSynchronize();
  std::vector<unsigned> h((size_t)grid * nw * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu;
  int hist[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < grid; ++b) {
    int simd_cnt[4] = {0, 0, 0, 0};
    unsigned cu_key = 0;
    for (int w = 0; w < nw; ++w) {
      const unsigned id = h[(b * nw + w) * 2], xcc = h[(b * nw + w) * 2 + 1] & 0xf;
      const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 7;
      simd_cnt[simd]++;
      cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      if (b < 4) printf("block %d wave %d: xcc %u se %u sh %u cu %u simd %u\n", b, w, xcc, se, sh, cu, simd);
    }
    per_cu[cu_key]++;
    int mx = 0;
    for (int s = 0; s < 4; ++s) mx = simd_cnt[s] > mx ? simd_cnt[s] : mx;
    hist[mx < 4 ? mx : 4]++;
  }
  printf("blocks by max waves on one SIMD: 1:%d 2:%d 3:%d 4+:%d\n", hist[1], hist[2], hist[3], hist[4]);
  std::map<int, int> cu_hist;
  for (auto& kv : per_cu) cu_hist[kv.second]++;
  printf("distinct CUs used: %zu;", per_cu.size());
  for (auto& kv : cu_hist) printf("  %d CUs ran %d block(s)", kv.second, kv.first);
  printf("\n");
  return 0;
}
