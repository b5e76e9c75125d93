// h2d_bench.hip -- what bounds a host-array upload on this box?  (feeds the design of csrc/bc_upload.hip)
//   hipcc --offload-arch=gfx950 -O2 tools/h2d_bench.hip -o tools/h2d_bench.bin -lpthread && tools/h2d_bench.bin [GiB]
// Measures, for a buffer of the given size:
//   a  pinned -> device, one hipMemcpyAsync                                  (the DMA ceiling of the link)
//   b  pageable -> device, one hipMemcpyAsync                                (what bc_data_from_host did until round 3)
//   c  pageable -> pinned memcpy with T threads, no DMA                      (the host-side staging rate)
//   d  T threads, each: memcpy 8 MiB sub-chunks into its two pinned buffers, DMA on its own stream  (bc_upload.hip's scheme)
//   e  hipHostRegister of the pageable buffer, then one DMA                  (pin in place)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;
  const size_t bytes = (size_t)(gib * (1ull << 30)) & ~(size_t)0xfffff;
  printf("hardware_concurrency %u, buffer %.2f GiB\n", std::thread::hardware_concurrency(), bytes / double(1ull << 30));
  char* dev = nullptr;
  CK(hipMalloc((void**)&dev, bytes));
  char* pageable = (char*)malloc(bytes);
  for (size_t i = 0; i < bytes; i += 4096) pageable[i] = (char)i;      // touch every page
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  {  // a
    char* pinned = nullptr;
    const size_t pb = bytes < ((size_t)1 << 30) ? bytes : ((size_t)1 << 30);
    double t0 = now();
    CK(hipHostMalloc((void**)&pinned, pb, hipHostMallocDefault));
    printf("hipHostMalloc(%.2f GiB): %.1f ms\n", pb / double(1ull << 30), 1e3 * (now() - t0));
    memset(pinned, 1, pb);
    for (int rep = 0; rep < 3; ++rep) {
      t0 = now();
      CK(hipMemcpyAsync(dev, pinned, pb, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      printf("a pinned->device    %.1f GB/s\n", pb / (now() - t0) / 1e9);
    }
    CK(hipHostFree(pinned));
  }
  for (int rep = 0; rep < 2; ++rep) {  // b
    double t0 = now();
    CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    printf("b pageable->device  %.1f GB/s\n", bytes / (now() - t0) / 1e9);
  }
  const int TS[] = {1, 2, 4, 8, 16};
  for (int T : TS) {
    // c: staging rate alone (each thread cycles through its two 8 MiB pinned buffers)
    const size_t sub = (size_t)8 << 20;
    std::vector<char*> stg(2 * T);
    for (auto& p : stg) CK(hipHostMalloc((void**)&p, sub, hipHostMallocDefault));
    std::vector<hipStream_t> sts(T);
    std::vector<hipEvent_t> evs(2 * T);
    for (auto& s : sts) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
    for (int mode = 0; mode < 2; ++mode) {
      double t0 = now();
      std::vector<std::thread> pool;
      const size_t chunk = (size_t)128 << 20;
      const size_t nchunks = (bytes + chunk - 1) / chunk;
      for (int t = 0; t < T; ++t)
        pool.emplace_back([&, t]() {
          (void)hipSetDevice(0);
          int buf = 0;
          bool used[2] = {false, false};
          for (size_t c = t; c < nchunks; c += T) {
            size_t off = c * chunk, left = (bytes - off) < chunk ? (bytes - off) : chunk;
            while (left > 0) {
              const size_t nb = left < sub ? left : sub;
              const int k = 2 * t + buf;
              if (mode == 1 && used[buf]) (void)hipEventSynchronize(evs[k]);
              memcpy(stg[k], pageable + off, nb);
              if (mode == 1) {
                (void)hipMemcpyAsync(dev + off, stg[k], nb, hipMemcpyHostToDevice, sts[t]);
                (void)hipEventRecord(evs[k], sts[t]);
                used[buf] = true;
              }
              off += nb; left -= nb; buf ^= 1;
            }
          }
          if (mode == 1) (void)hipStreamSynchronize(sts[t]);
        });
      for (auto& th : pool) th.join();
      printf("%s T=%2d  %.1f GB/s\n", mode == 0 ? "c memcpy->pinned only  " : "d memcpy->pinned + DMA ", T, bytes / (now() - t0) / 1e9);
    }
    for (auto& p : stg) CK(hipHostFree(p));
    for (auto& s : sts) CK(hipStreamDestroy(s));
    for (auto& e : evs) CK(hipEventDestroy(e));
  }
  {  // e
    double t0 = now();
    hipError_t e = hipHostRegister(pageable, bytes, hipHostRegisterDefault);
    const double treg = now() - t0;
    if (e == hipSuccess) {
      t0 = now();
      CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      const double tc = now() - t0;
      printf("e hipHostRegister %.1f ms (%.1f GB/s), then DMA %.1f GB/s; together %.1f GB/s\n", 1e3 * treg, bytes / treg / 1e9, bytes / tc / 1e9,
             bytes / (treg + tc) / 1e9);
      (void)hipHostUnregister(pageable);
    } else {
      printf("e hipHostRegister failed: %s\n", hipGetErrorString(e));
    }
  }
  return 0;
}
