// Microbenchmark behind profiles/r01_notes.md "fp16 pre-filter sweep -- what bounds it": the streaming loop of
// k_sweep_f16 (bc_prefilter.hip) on synthetic data, in variants selected by the template parameter VAR:
//   0 baseline (load 10 planes, consume, repeat)      1 register double buffering          2 no interval epilogue
//   3 loads only                                      4 / 5 fp32 epilogue (without / with double buffering)
//   6 / 7 dynamic tile queue                          8 one dot product   9 packed-fp16 math (VALU-load probes)
//   10 / 11 continuous pipeline across tiles          12 / 13 alternating sweep direction (nt / temporal loads)
//   14 no per-row bound store   15 no store, no norm read   16 live-byte mask instead of norms, no store
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -o sweep_f16_variants tools/sweep_f16_variants.hip
// Run:   ./sweep_f16_variants [rows]      (default 10M rows, S = 100)
// Findings that shaped the shipped kernels: double buffering +8 %; the 4 B/row bound store costs 15 % and the
// 8 B/row norm read 5 % of the bandwidth; VALU work, tail balance and the Infinity Cache do not matter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define HT 512
struct A { const unsigned char* live; int rev; unsigned* ctr; const _Float16* u; const double* norms; const double* v; float* ub; float* tile_u; double* blk_l; long long n_rows, ptiles; int s; double delta; };

__device__ __forceinline__ void interval(double s0, double s1, double delta, double& U, double& L) {
  const double a = fabs(s1) + delta; const double c = 1. - a * a;
  if (!(s0 == s0) || !(s1 == s1) || !(c > 1e-6)) { U = INFINITY; L = -INFINITY; return; }
  const double f = s0 / sqrt(1. - s1 * s1); const double rc = 1. / sqrt(c);
  const double e = delta * (rc + (fabs(s0) + delta) * a * rc * rc * rc) * 1.001 + 1e-13 * (1. + fabs(f));
  U = f + e; L = f - e;
}
__device__ __forceinline__ void interval32(float s0, float s1, float delta, float& U, float& L) {
  const float a = fabsf(s1) + delta; const float c = 1.f - a * a;
  if (!(s0 == s0) || !(s1 == s1) || !(c > 1e-2f)) { U = INFINITY; L = -INFINITY; return; }
  const float rc = __frsqrt_rn(c);
  const float f = s0 * __frsqrt_rn(1.f - s1 * s1);
  const float e = delta * (rc + (fabsf(s0) + delta) * a * rc * rc * rc) * 1.002f + fabsf(f) * (1e-7f * rc * rc + 6e-7f) + 1e-7f;
  U = f + e; L = f - e;
}

// VAR: 0 = baseline, 1 = register double buffering, 2 = baseline w/o epilogue, 3 = load only, 4 = fp32 epilogue, 5 = prefetch + fp32 epilogue
template <int VAR>
__global__ __launch_bounds__(256) void k(A a) {
  __shared__ double sl[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double best_l = -INFINITY;
  const int S = a.s;
  const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.v);
  constexpr int U = 10;
  constexpr bool DYN = (VAR == 6 || VAR == 7);
  long long t = (long long)blockIdx.x * 4 + wave;
  for (;;) {
    if (DYN) { unsigned q = 0; if (lane == 0) q = atomicAdd(a.ctr, 1u); t = __shfl(q, 0, 64); }
    if (t >= a.ptiles) break;
    const long long tt = ((VAR == 12 || VAR == 13) && a.rev) ? a.ptiles - 1 - t : t;
    const h8* __restrict__ p = reinterpret_cast<const h8*>(a.u + (size_t)tt * S * HT) + lane;
    float a0[8], a1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = a1[j] = 0.f;
    if (VAR == 1 || VAR == 5 || VAR == 7 || VAR == 8 || VAR == 9 || VAR == 12 || VAR == 13 || VAR == 14 || VAR == 15 || VAR == 16) {
      h8 x[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = (VAR == 13) ? p[(size_t)u * 64] : __builtin_nontemporal_load(p + (size_t)u * 64);
      for (int k = 0; k < S; k += U) {     // S % U == 0 assumed in this variant
        const bool more = k + U < S;
        if (more) {
#pragma unroll
          for (int u = 0; u < U; ++u) y[u] = (VAR == 13) ? p[(size_t)(k + U + u) * 64] : __builtin_nontemporal_load(p + (size_t)(k + U + u) * 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double2 vv = v2[k + u];
          const float vx = (float)vv.x, vy = (float)vv.y;
          if (VAR == 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a0[j] = fmaf((float)x[u][j], vx, a0[j]);
          } else if (VAR == 9) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 vh = {(_Float16)vx, (_Float16)vx};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              h2 xx = {x[u][2 * j], x[u][2 * j + 1]};
              h2* acc = reinterpret_cast<h2*>(&a0[j]);
              *acc = xx * vh + *acc;
            }
          } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { a0[j] = fmaf((float)x[u][j], vx, a0[j]); a1[j] = fmaf((float)x[u][j], vy, a1[j]); }
          }
        }
        if (more) {
#pragma unroll
          for (int u = 0; u < U; ++u) x[u] = y[u];
        }
      }
    } else {
      for (int k = 0; k + U <= S; k += U) {
        h8 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p + (size_t)(k + u) * 64);
        if (VAR == 3 || VAR == 6) {
#pragma unroll
          for (int u = 0; u < U; ++u) a0[u & 7] += (float)x[u][0];
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const double2 vv = v2[k + u];
            const float vx = (float)vv.x, vy = (float)vv.y;
#pragma unroll
            for (int j = 0; j < 8; ++j) { a0[j] = fmaf((float)x[u][j], vx, a0[j]); a1[j] = fmaf((float)x[u][j], vy, a1[j]); }
          }
        }
      }
    }
    const long long r = tt * HT + 8 * lane;
    const unsigned lv = (VAR == 16) ? a.live[tt * 64 + lane] : 0u;
    float tmax = -INFINITY;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f4 ub;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = 4 * h + j;
        float uf = -INFINITY;
        if (r + q < a.n_rows && (VAR == 15 || (VAR == 16 ? ((lv >> q) & 1u) != 0u : a.norms[r + q] != 0.))) {
          if (VAR == 2 || VAR == 3 || VAR == 6) { uf = a0[q] + a1[q]; best_l = fmax(best_l, (double)uf); }
          else if (VAR == 4 || VAR == 5) { float Ub, Lb; interval32(a0[q], a1[q], (float)a.delta, Ub, Lb); uf = Ub; best_l = fmax(best_l, (double)Lb); }
          else { double Ub, Lb; interval((double)a0[q], (double)a1[q], a.delta, Ub, Lb); uf = __double2float_ru(Ub); best_l = fmax(best_l, Lb); }
        }
        ub[j] = uf; tmax = fmaxf(tmax, uf);
      }
      if (VAR != 14 && VAR != 15 && VAR != 16) *reinterpret_cast<f4*>(a.ub + r + 4 * h) = ub;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tmax = fmaxf(tmax, __shfl_down(tmax, d, 64));
    if (lane == 0) a.tile_u[t] = tmax;
    if (!DYN) t += (long long)gridDim.x * 4;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best_l = fmax(best_l, __shfl_down(best_l, d, 64));
  if (lane == 0) sl[wave] = best_l;
  __syncthreads();
  if (threadIdx.x == 0) a.blk_l[blockIdx.x] = fmax(fmax(sl[0], sl[1]), fmax(sl[2], sl[3]));
}


template <int VAR>
__global__ __launch_bounds__(256) void kp(A a, const unsigned char* live) {
  __shared__ double sl[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double best_l = -INFINITY;
  const int S = a.s;
  const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.v);
  constexpr int U = 10;
  const int nb = S / U;
  const long long stride = (long long)gridDim.x * 4;
  long long t = (long long)blockIdx.x * 4 + wave;
  if (t < a.ptiles) {
    h8 x[U], y[U];
    const h8* __restrict__ p = reinterpret_cast<const h8*>(a.u + (size_t)t * S * HT) + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
    unsigned lv = live[t * 64 + lane], lvn = 0;
    float a0[8], a1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = a1[j] = 0.f;
    int b = 0;
    for (;;) {
      const bool last = (b + 1 == nb);
      const long long tn = last ? t + stride : t;
      const int bn = last ? 0 : b + 1;
      const bool more = tn < a.ptiles;
      if (more) {
        const h8* __restrict__ pn = reinterpret_cast<const h8*>(a.u + (size_t)tn * S * HT) + lane + (size_t)bn * U * 64;
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(pn + (size_t)u * 64);
        if (last) lvn = live[tn * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double2 vv = v2[b * U + u];
        const float vx = (float)vv.x, vy = (float)vv.y;
#pragma unroll
        for (int j = 0; j < 8; ++j) { a0[j] = fmaf((float)x[u][j], vx, a0[j]); a1[j] = fmaf((float)x[u][j], vy, a1[j]); }
      }
      if (last) {
        const long long r = t * HT + 8 * lane;
        float tmax = -INFINITY;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f4 ub;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = 4 * h + j;
            float uf = -INFINITY;
            const bool ok = (VAR == 11) ? ((lv >> q) & 1u) : (r + q < a.n_rows && a.norms[r + q] != 0.);
            if (ok) {
              double Ub, Lb; interval((double)a0[q], (double)a1[q], a.delta, Ub, Lb); uf = __double2float_ru(Ub); best_l = fmax(best_l, Lb);
            }
            ub[j] = uf; tmax = fmaxf(tmax, uf);
            a0[q] = 0.f; a1[q] = 0.f;
          }
          *reinterpret_cast<f4*>(a.ub + r + 4 * h) = ub;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) tmax = fmaxf(tmax, __shfl_down(tmax, d, 64));
        if (lane == 0) a.tile_u[t] = tmax;
        lv = lvn;
      }
      if (!more) break;
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = y[u];
      t = tn; b = bn;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best_l = fmax(best_l, __shfl_down(best_l, d, 64));
  if (lane == 0) sl[wave] = best_l;
  __syncthreads();
  if (threadIdx.x == 0) a.blk_l[blockIdx.x] = fmax(fmax(sl[0], sl[1]), fmax(sl[2], sl[3]));
}
template <int VAR> float runp(A a, const unsigned char* live, int grid, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kp<VAR>, dim3(grid), dim3(256), 0, 0, a, live);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kp<VAR>, dim3(grid), dim3(256), 0, 0, a, live);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

__global__ void fill(_Float16* u, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    u[i] = (_Float16)(0.1f * (float)((int)((i * 2654435761u) >> 20 & 1023) - 512) / 512.f);
}
template <int VAR> float run(A a, int grid, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) { if (VAR >= 6) hipMemsetAsync(a.ctr, 0, 4, 0); hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, a); }
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) { if (VAR >= 6 && VAR < 8) hipMemsetAsync(a.ctr, 0, 4, 0); a.rev = i & 1; hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, a); }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 10000000; const int S = 100;
  A a; a.n_rows = n; a.s = S; a.ptiles = (n + HT - 1) / HT; a.delta = 4.95e-4;
  _Float16* u; double *norms, *v, *bl; float *ub, *tu;
  size_t ne = (size_t)a.ptiles * S * HT;
  hipMalloc(&u, ne * 2); hipMalloc(&norms, a.ptiles * HT * 8); hipMalloc(&v, 2 * S * 8); hipMalloc(&ub, a.ptiles * HT * 4); hipMalloc(&tu, a.ptiles * 4); hipMalloc(&bl, 65536 * 8); hipMalloc(&a.ctr, 256);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, u, ne);
  std::vector<double> hn(a.ptiles * HT, 1.0), hv(2 * S);
  for (int i = 0; i < 2 * S; ++i) hv[i] = 0.1 * sin(i * 0.7);
  hipMemcpy(norms, hn.data(), hn.size() * 8, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
  a.u = u; a.norms = norms; a.v = v; a.ub = ub; a.tile_u = tu; a.blk_l = bl;
  const double bytes = (double)n * (2.0 * S + 12);
  unsigned char* live; hipMalloc(&live, a.ptiles * 64); hipMemset(live, 0xff, a.ptiles * 64);
  for (int wpc : {8, 12, 16, 24, 32}) {
    long long wmax = 256LL * wpc, rounds = (a.ptiles + wmax - 1) / wmax, waves = (a.ptiles + rounds - 1) / rounds; int grid = (int)((waves + 3) / 4);
    float t = runp<10>(a, live, grid, 50); printf("wpc %2d grid %5d  pipeline norms    %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = runp<11>(a, live, grid, 50); printf("wpc %2d grid %5d  pipeline live     %.4f ms  %.0f GB/s\n", wpc, grid, t, ((double)n * (2.0 * S + 4.125)) / t / 1e6);
  }
  a.live = live;
  for (int wpc : {16}) {
    long long wmax = 256LL * wpc, rounds = (a.ptiles + wmax - 1) / wmax, waves = (a.ptiles + rounds - 1) / rounds; int grid = (int)((waves + 3) / 4);
    float t;
    t = run<0>(a, grid, 50); printf("wpc %2d grid %5d  baseline          %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<1>(a, grid, 50); printf("wpc %2d grid %5d  prefetch          %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<2>(a, grid, 50); printf("wpc %2d grid %5d  no epilogue       %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<3>(a, grid, 50); printf("wpc %2d grid %5d  load only         %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<4>(a, grid, 50); printf("wpc %2d grid %5d  fp32 epilogue     %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<5>(a, grid, 50); printf("wpc %2d grid %5d  prefetch+fp32 epi %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<8>(a, grid, 50); printf("wpc %2d grid %5d  prefetch one dot  %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<9>(a, grid, 50); printf("wpc %2d grid %5d  prefetch pk f16   %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<14>(a, grid, 50); printf("wpc %2d grid %5d  prefetch no ub st %.4f ms  %.0f GB/s\n", wpc, grid, t, ((double)n * (2.0 * S + 8)) / t / 1e6);
    t = run<15>(a, grid, 50); printf("wpc %2d grid %5d  prefetch no ub/nrm %.4f ms  %.0f GB/s\n", wpc, grid, t, ((double)n * (2.0 * S)) / t / 1e6);
    t = run<16>(a, grid, 50); printf("wpc %2d grid %5d  prefetch live/noub %.4f ms  %.0f GB/s\n", wpc, grid, t, ((double)n * (2.0 * S + 0.125)) / t / 1e6);
    t = run<12>(a, grid, 50); printf("wpc %2d grid %5d  prefetch alt dir  %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<13>(a, grid, 50); printf("wpc %2d grid %5d  prefetch alt temp %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    grid = 64 * wpc;
    t = run<6>(a, grid, 50); printf("wpc %2d grid %5d  load only dynamic %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
    t = run<7>(a, grid, 50); printf("wpc %2d grid %5d  prefetch dynamic  %.4f ms  %.0f GB/s\n", wpc, grid, t, bytes / t / 1e6);
  }
  return 0;
}
