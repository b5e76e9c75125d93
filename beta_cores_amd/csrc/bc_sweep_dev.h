// Device-side pieces of K3 shared by the sweep launcher (bc_sweep.hip) and the pre-filter's in-kernel
// fp64 fallback (bc_prefilter.hip): argument block, streaming load, score epilogue, record emission and
// the per-block sweep body.
#pragma once
#include "bc_internal.h"
#include <climits>
#include <cmath>

struct bc_sweep_args {
  const double* tiles;
  const double* norms;
  const double* v;          // mode 0: [S][2] (cdir, xw) interleaved; mode 1: [S]
  const int* skip_flag;     // optional device flag: when non-zero the sweep is a no-op
  long long n_rows;
  long long ntiles;
  long long row_offset;
  double post_div;
  int s;
};

// Phi is read exactly once per sweep and is far larger than the 256 MiB Infinity Cache at the sizes
// that matter: stream it with the non-temporal policy so it does not evict the vectors and norms.
// Measured (N=10M, S=100): 1.30 ms -> 1.15 ms per sweep, 6.2 -> 7.0 TB/s.
#ifndef BC_SWEEP_NO_NT
typedef double bc_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 bc_nt_load(const double2* p) {
  const bc_d2 v = __builtin_nontemporal_load(reinterpret_cast<const bc_d2*>(p));
  return make_double2(v.x, v.y);
}
#define BC_STREAM_LOAD(p) bc_nt_load(p)
#else
#define BC_STREAM_LOAD(p) (*(p))
#endif

template <int MODE>
__device__ __forceinline__ double bc_row_score(double a0, double a1, double nr, double post_div) {
  if (MODE == 0) {
    // giga.py:31-38 on normalised columns
    const double s0 = a0 / nr, s1 = a1 / nr;
    const bool ok = (s1 > -1. + 1e-14) && (1. - s1 * s1 > 0.);
    const double den = ok ? sqrt(1. - s1 * s1) : INFINITY;
    return s0 / den;
  } else {
    return a0 / nr / post_div;
  }
}

// Reduce per-block candidates to the local winner and emit its candidate record (whole block):
//   rec[0] = score, rec[1] = global index (int64 bits), rec[2] = row norm, rec[3] = 1.0 if valid,
//   rec[4..4+S) = Phi[row, :]  (the un-normalised column A[:, f])
__device__ __forceinline__ void bc_emit_record(const double* __restrict__ blk_val, const long long* __restrict__ blk_idx,
                                               int nblk, const double* __restrict__ tiles,
                                               const double* __restrict__ norms, int s, long long row_offset,
                                               bool skip, double* __restrict__ rec, double* sv, long long* si,
                                               long long* win) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double bv = -INFINITY;
  long long bi = LLONG_MAX;
  if (!skip)
    for (int i = threadIdx.x; i < nblk; i += blockDim.x)
      if (bc_better(blk_val[i], blk_idx[i], bv, bi)) { bv = blk_val[i]; bi = blk_idx[i]; }
  bc_wave_argmax(bv, bi);
  __syncthreads();
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (bc_better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
    const bool valid = bi != LLONG_MAX;
    rec[0] = bv;
    reinterpret_cast<long long*>(rec)[1] = valid ? bi : -1;
    rec[2] = valid ? norms[bi - row_offset] : 0.0;
    rec[3] = valid ? 1.0 : 0.0;
    *win = valid ? bi - row_offset : -1;
  }
  __syncthreads();
  const long long r = *win;
  for (int k = threadIdx.x; k < s; k += blockDim.x) rec[BC_REC_HDR + k] = (r >= 0) ? tiles[bc_tile_off(r, k, s)] : 0.0;
}

// One block's share of the sweep: tiles block*4+wave, +nblocks*4, ...  Leaves the block's best
// (score, global row) in thread 0's best_v / best_i (sv/si: 4-entry shared scratch).
template <int MODE>
__device__ __forceinline__ void bc_sweep_block(const bc_sweep_args& a, int block, int nblocks, bool skip, double* sv,
                                               long long* si, double& best_v, long long& best_i) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  best_v = -INFINITY;
  best_i = LLONG_MAX;
  if (!skip) {
    const int S = a.s;
    const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.v);
    const double* __restrict__ v1 = a.v;
    for (long long t = (long long)block * 4 + wave; t < a.ntiles; t += (long long)nblocks * 4) {
      const double2* __restrict__ p = reinterpret_cast<const double2*>(a.tiles + (size_t)t * S * BC_TILE) + lane;
      double a00 = 0., a01 = 0., a10 = 0., a11 = 0.;
      int k = 0;
#ifndef BC_SWEEP_U
#define BC_SWEEP_U 10
#endif
      constexpr int U = BC_SWEEP_U;
      for (; k + U <= S; k += U) {
        double2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = BC_STREAM_LOAD(p + (size_t)(k + u) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (MODE == 0) {
            const double2 vv = v2[k + u];
            a00 = fma(x[u].x, vv.x, a00);
            a01 = fma(x[u].x, vv.y, a01);
            a10 = fma(x[u].y, vv.x, a10);
            a11 = fma(x[u].y, vv.y, a11);
          } else {
            const double vv = v1[k + u];
            a00 = fma(x[u].x, vv, a00);
            a10 = fma(x[u].y, vv, a10);
          }
        }
      }
      for (; k < S; ++k) {
        const double2 x = BC_STREAM_LOAD(p + (size_t)k * 64);
        if (MODE == 0) {
          const double2 vv = v2[k];
          a00 = fma(x.x, vv.x, a00);
          a01 = fma(x.x, vv.y, a01);
          a10 = fma(x.y, vv.x, a10);
          a11 = fma(x.y, vv.y, a11);
        } else {
          const double vv = v1[k];
          a00 = fma(x.x, vv, a00);
          a10 = fma(x.y, vv, a10);
        }
      }
      const long long r = t * BC_TILE + 2 * lane;
      const double2 nr = reinterpret_cast<const double2*>(a.norms)[(size_t)t * 64 + lane];
      if (r < a.n_rows && nr.x != 0.) {
        const double sc = bc_row_score<MODE>(a00, a01, nr.x, a.post_div);
        const long long gi = a.row_offset + r;
        if (bc_better(sc, gi, best_v, best_i)) { best_v = sc; best_i = gi; }
      }
      if (r + 1 < a.n_rows && nr.y != 0.) {
        const double sc = bc_row_score<MODE>(a10, a11, nr.y, a.post_div);
        const long long gi = a.row_offset + r + 1;
        if (bc_better(sc, gi, best_v, best_i)) { best_v = sc; best_i = gi; }
      }
    }
  }
  bc_wave_argmax(best_v, best_i);
  if (lane == 0) { sv[wave] = best_v; si[wave] = best_i; }
  __syncthreads();
  if (threadIdx.x == 0)
    for (int w = 1; w < 4; ++w)
      if (bc_better(sv[w], si[w], best_v, best_i)) { best_v = sv[w]; best_i = si[w]; }
}
