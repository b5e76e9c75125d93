// K3 with a reduced-precision pre-filter: the per-iteration sweep streams a fp16 (default) or fp32 mirror of
// the normalised rows -- a quarter / half of the bytes -- while returning EXACTLY the row the fp64 sweep
// (bc_sweep.hip) would return.
//
//   u[i, :] = round( Phi[i, :] / ||Phi[i, :]|| )       built once per solver (2*N*S or 4*N*S bytes)
//
//   pass A  k_sweep_f16 / k_sweep_f32
//                         streams u (non-temporal) and for every row computes an interval [L_i, U_i] that
//                         provably contains the fp64 kernel's score of that row: the rounding of the inputs and
//                         of the accumulation moves each normalised dot product by at most delta * ||v||
//                         (Cauchy-Schwarz; delta32 = 6.2e-8 with fp64 accumulation, delta16 ~ 4.9e-4 with the
//                         fp32 fma chain, see bc_pref_delta16), propagated through the score formula by the
//                         mean-value theorem.  Rows the bound cannot handle (|s1| close to 1, NaN) get
//                         [-inf, +inf].  Writes U_i (fp32, rounded up), the tile maxima of U and the block maxima
//                         of L.
//   pass B  k_rescore     (block 0) Lmax = max_i L_i; every row with U_i >= Lmax is a candidate -- the true
//                         argmax is always among them, typically a handful of rows; whole tiles are skipped
//                         through their maximum U.  The candidates' scores are recomputed from the fp64 Phi
//                         with the same arithmetic (same fma chain, same epilogue) as k_sweep, the argmax is
//                         taken with NumPy's tie rule, and the candidate record is emitted.
//   If the candidate list overflows, the same launch falls back to the full fp64 sweep: k_rescore runs with one
//   block per CU; block 0 does the work above while the others wait for its verdict (they exit at once in the
//   common case) and, on overflow, all of them sweep the fp64 Phi and block 0 merges their winners.
//
// Algorithmic traffic per row: fp16 2*S + 8 (norm) + 4 (U written) bytes; fp32 4*S + 12.
//
// The fp16 kernel accumulates in fp32 (v_cvt_f32_f16 + v_pk_fma_f32: both dot products of a row in one packed
// FMA) and double-buffers its loads in registers; its planes are padded with zeros to a multiple of BC_HU and
// the sweep vectors carry a zero tail (BC_V_PAD, bc_snnls.hip), so there is no remainder loop.
#include "bc_sweep_dev.h"
#include <cstdlib>
#include <cstring>

#define BC_PTILE 256   // rows per u32 tile: one sample of a tile = 1 KiB = 64 lanes x float4
typedef _Float16 bc_hq2 __attribute__((ext_vector_type(2)));   // int8 mirror: (scale, delta) of a row (bc_prefilter_i8.h)

struct bc_pref {
  bc_ctx* ctx = nullptr;
  bc_phi* phi = nullptr;
  float* u32 = nullptr;       // fp32: [ptiles][S][256]
  _Float16* u16 = nullptr;    // fp16: [ptiles][SP][512]
  unsigned char* live = nullptr;   // fp16: [ptiles][64] live-row mask
  int* u8 = nullptr;          // int8: [ptiles][sp4][256] dwords (bc_prefilter_i8.h)
  bc_hq2* rowq = nullptr;     // int8: (scale, delta) per row, two halfs
  float2* tile_cand = nullptr;   // int8: [ptiles][4]
  int* tile_ncand = nullptr;     // int8: [ptiles]
  int sp4 = 0;                // int8: k-groups stored per tile
  int prec = 32;              // 32, 16 or 8
  int ptile = BC_PTILE;       // rows per pre-filter tile (256 for fp32, 512 for fp16)
  int sp = 0;                 // fp16: planes stored per tile
  float* ub = nullptr;        // [ptiles*256] upper bounds of the last sweep
  float* tile_u = nullptr;    // [ptiles] per-tile maximum of the upper bounds
  double* blk_l = nullptr;    // [grid] block maxima of the lower bounds
  float* blk_u = nullptr;     // [grid] block maxima of the upper bounds (lets the selection skip whole blocks of tiles)
  long long* cand = nullptr;  // [cap] candidate LOCAL rows
  int* ctrl = nullptr;        // [1] the last launch fell back to the fp64 sweep, [2] hand-shake timeout, [3] fallbacks so far,
                              // [4..5] sweeps so far (u64), [6..7] candidates rescored so far (u64)
  unsigned* sync = nullptr;   // in-launch hand-shake of k_rescore (verdict, arrivals)
  unsigned epoch = 0;         // launch sequence number of k_rescore
  int helpers_grid = 1;       // blocks of k_rescore: block 0 + fallback helpers
  int cap = 4096;
  long long ptiles = 0;
  int grid = 1;
  void* slab = nullptr;
};

struct PrefArgs {
  const float* u32;
  const _Float16* u16;
  const unsigned char* live; // fp16: [ptiles][64] bytes, bit q of byte l = row 8*l + q of the tile is live
  double delta;              // per-dot-product bound for unit ||v||
  int sp;                    // fp16: stored planes per tile (S padded to a multiple of BC_HU)
  const double* norms;
  const double* v;
  const int* skip_flag;
  const double* v_norm;      // dot mode: ||v|| lives in the solver state (device); GIGA: null (= 1)
  float* ub;
  float* tile_u;
  double* blk_l;
  float* blk_u;
  long long n_rows, ptiles;
  double post_div;
  int s;
};

typedef float bc_f4 __attribute__((ext_vector_type(4)));

// bound of |fp64-kernel dot - fp32-input dot| for unit ||v||: 2^-24 (input rounding, Cauchy-Schwarz with
// ||u_i|| <= 1 + 2^-24) plus the two fma chains' rounding (S * 2^-53 each), with margin.
#define BC_PREF_DELTA 6.2e-8

template <int MODE>
__device__ __forceinline__ void bc_score_interval(double s0, double s1, double delta, double post_div, double& U, double& L) {
  if (MODE == 0) {
    const double a = fabs(s1) + delta;
    const double c = 1. - a * a;
    if (!(s0 == s0) || !(s1 == s1) || !(c > 1e-6)) {   // NaN, or too close to the validity boundary of giga.py:33
      U = INFINITY;
      L = -INFINITY;
      return;
    }
    const double f = s0 / sqrt(1. - s1 * s1);
    const double rc = 1. / sqrt(c);
    const double e = delta * (rc + (fabs(s0) + delta) * a * rc * rc * rc) * 1.001 + 1e-13 * (1. + fabs(f));
    U = f + e;
    L = f - e;
  } else {
    if (!(s0 == s0)) { U = INFINITY; L = -INFINITY; return; }
    const double f = s0 / post_div;
    const double e = (delta * 1.001 + 1e-13 * fabs(s0)) / fabs(post_div);
    U = f + e;
    L = f - e;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_sweep_f32(PrefArgs a) {
  __shared__ double sl[4];
  __shared__ float su[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double best_l = -INFINITY;
  float umax = -INFINITY;
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;
  if (!skip) {
    const int S = a.s;
    const double delta = (MODE == 0) ? a.delta : a.delta * (*a.v_norm);
    const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.v);
    const double* __restrict__ v1 = a.v;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < a.ptiles; t += (long long)gridDim.x * 4) {
      const bc_f4* __restrict__ p = reinterpret_cast<const bc_f4*>(a.u32 + (size_t)t * S * BC_PTILE) + lane;
      double a0[4] = {0., 0., 0., 0.}, a1[4] = {0., 0., 0., 0.};
      int k = 0;
      constexpr int U = 10;
      for (; k + U <= S; k += U) {
        bc_f4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p + (size_t)(k + u) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (MODE == 0) {
            const double2 vv = v2[k + u];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const double xd = (double)x[u][j];
              a0[j] = fma(xd, vv.x, a0[j]);
              a1[j] = fma(xd, vv.y, a1[j]);
            }
          } else {
            const double vv = v1[k + u];
#pragma unroll
            for (int j = 0; j < 4; ++j) a0[j] = fma((double)x[u][j], vv, a0[j]);
          }
        }
      }
      for (; k < S; ++k) {
        const bc_f4 x = __builtin_nontemporal_load(p + (size_t)k * 64);
        if (MODE == 0) {
          const double2 vv = v2[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double xd = (double)x[j];
            a0[j] = fma(xd, vv.x, a0[j]);
            a1[j] = fma(xd, vv.y, a1[j]);
          }
        } else {
          const double vv = v1[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) a0[j] = fma((double)x[j], vv, a0[j]);
        }
      }
      const long long r = t * BC_PTILE + 4 * lane;
      bc_f4 ub;
      float tmax = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float uf = -INFINITY;
        if (r + j < a.n_rows && a.norms[r + j] != 0.) {
          double Ub, Lb;
          bc_score_interval<MODE>(a0[j], a1[j], delta, a.post_div, Ub, Lb);
          uf = __double2float_ru(Ub);
          best_l = fmax(best_l, Lb);
        }
        ub[j] = uf;
        tmax = fmaxf(tmax, uf);
      }
      *reinterpret_cast<bc_f4*>(a.ub + r) = ub;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) tmax = fmaxf(tmax, __shfl_down(tmax, d, BC_WAVE));
      if (lane == 0) a.tile_u[t] = tmax;       // lets the selection pass skip whole tiles
      umax = fmaxf(umax, tmax);                // (meaningful in lane 0)
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best_l = fmax(best_l, __shfl_down(best_l, d, BC_WAVE));
  if (lane == 0) { sl[wave] = best_l; su[wave] = umax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.blk_l[blockIdx.x] = fmax(fmax(sl[0], sl[1]), fmax(sl[2], sl[3]));
    a.blk_u[blockIdx.x] = fmaxf(fmaxf(su[0], su[1]), fmaxf(su[2], su[3]));
  }
}


#include "bc_prefilter_i8.h"

// ---- fp16 variant.  Tile = 512 rows, [S][512] halfs: one sample of a tile = 1 KiB = 64 lanes x 8 halfs.
#define BC_HTILE 512
typedef _Float16 bc_h8 __attribute__((ext_vector_type(8)));

// bound of |fp64-kernel dot - (fp16 rows, fp32 v, fp32 fma chain) dot| for unit ||v||, ||u|| = 1:
//   rows:   |u^ - u|_2 <= (2^-11 + 2^-23) (RN to half through float, normal range) + sqrt(S) 2^-25 (subnormal halfs)
//   v:      2^-24 (RN to float)
//   chain:  S roundings of 2^-24 relative to sum |u^_i v^_i| <= ||u^|| ||v^||
// every term taken with margin.
static double bc_pref_delta16(int S) { return 4.8845e-4 + 3.0e-8 * sqrt((double)S) + 6.1e-8 * (double)(S + 2); }

#define BC_HU 10   // sample planes per batch; the stored plane count is padded to a multiple (zero planes)

template <int MODE>
__global__ __launch_bounds__(256) void k_sweep_f16(PrefArgs a) {
  __shared__ double sl[4];
  __shared__ float su[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double best_l = -INFINITY;
  float umax = -INFINITY;
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;
  if (!skip) {
    const int SP = a.sp;
    const double delta = (MODE == 0) ? a.delta : a.delta * (*a.v_norm);
    const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.v);
    const double* __restrict__ v1 = a.v;
    constexpr int U = BC_HU;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < a.ptiles; t += (long long)gridDim.x * 4) {
      const bc_h8* __restrict__ p = reinterpret_cast<const bc_h8*>(a.u16 + (size_t)t * SP * BC_HTILE) + lane;
      float a0[8], a1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a0[j] = a1[j] = 0.f;
      // register double buffering: the next batch of planes is in flight while this one is consumed
      bc_h8 x[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
      const unsigned lv = a.live[t * 64 + lane];     // bit q: row 8*lane + q of the tile exists and has a non-zero norm
      for (int k = 0; k < SP; k += U) {
        const bool more = k + U < SP;
        if (more) {
#pragma unroll
          for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(p + (size_t)(k + U + u) * 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int kk = k + u;        // planes S..SP-1 are zero and so is v's tail (BC_V_PAD in bc_snnls.hip)
          if (MODE == 0) {
            const double2 vv = v2[kk];
            const float vx = (float)vv.x, vy = (float)vv.y;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              a0[j] = fmaf((float)x[u][j], vx, a0[j]);
              a1[j] = fmaf((float)x[u][j], vy, a1[j]);
            }
          } else {
            const float vx = (float)v1[kk];
#pragma unroll
            for (int j = 0; j < 8; ++j) a0[j] = fmaf((float)x[u][j], vx, a0[j]);
          }
        }
        if (more) {
#pragma unroll
          for (int u = 0; u < U; ++u) x[u] = y[u];
        }
      }
      // Nothing is written per row: the per-row bounds of the (one or two) tiles that matter are recomputed by
      // k_rescore from the same mirror.  A trickle of 4 B/row stores cost 15 % of the sweep's bandwidth, the
      // 8 B/row norm reads another 5 %; the live mask is one byte per lane and tile.
      float tmax = -INFINITY;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if ((lv >> q) & 1u) {
          double Ub, Lb;
          bc_score_interval<MODE>((double)a0[q], (double)a1[q], delta, a.post_div, Ub, Lb);
          tmax = fmaxf(tmax, __double2float_ru(Ub));
          best_l = fmax(best_l, Lb);
        }
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) tmax = fmaxf(tmax, __shfl_down(tmax, d, BC_WAVE));
      if (lane == 0) a.tile_u[t] = tmax;
      umax = fmaxf(umax, tmax);                // (meaningful in lane 0)
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best_l = fmax(best_l, __shfl_down(best_l, d, BC_WAVE));
  if (lane == 0) { sl[wave] = best_l; su[wave] = umax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.blk_l[blockIdx.x] = fmax(fmax(sl[0], sl[1]), fmax(sl[2], sl[3]));
    a.blk_u[blockIdx.x] = fmaxf(fmaxf(su[0], su[1]), fmaxf(su[2], su[3]));
  }
}

// u16 tile builder: one block per 512-row tile, thread = two rows; planes S..SP-1 are zero.  Also the live mask.
__global__ __launch_bounds__(256) void k_build_u16(const double* __restrict__ tiles, const double* __restrict__ norms,
                                                  long long n_rows, int S, int SP, _Float16* __restrict__ u16,
                                                  unsigned char* __restrict__ live_mask) {
  __shared__ unsigned char flags[BC_HTILE];
  const long long t = blockIdx.x;
  for (int h = 0; h < 2; ++h) {
    const int i = threadIdx.x + 256 * h;
    const long long r = t * BC_HTILE + i;
    const bool live = r < n_rows;
    const double nr = live ? norms[r] : 0.;
    flags[i] = (live && nr != 0.) ? 1 : 0;
    const double* p = tiles + (size_t)(r >> 7) * S * BC_TILE + (r & (BC_TILE - 1));
    _Float16* q = u16 + (size_t)t * SP * BC_HTILE + i;
    for (int k = 0; k < SP; ++k) {
      _Float16 u = (_Float16)0.f;
      if (k < S && live && nr != 0.) u = (_Float16)(float)(p[(size_t)k * BC_TILE] / nr);
      q[(size_t)k * BC_HTILE] = u;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    unsigned b = 0;
    for (int q = 0; q < 8; ++q) b |= (unsigned)flags[8 * threadIdx.x + q] << q;
    live_mask[t * 64 + threadIdx.x] = (unsigned char)b;
  }
}

struct RescoreArgs {
  const double* tiles;
  const double* norms;
  const double* v;
  const int* skip_flag;
  const float* ub;
  const float* tile_u;
  const double* blk_l;
  const float* blk_u;
  const _Float16* u16;       // fp16 mode: the per-row bounds of candidate tiles are recomputed from the mirror
  const unsigned char* live;
  const double* v_norm;
  double delta;
  int sp;
  const float2* tile_cand;   // int8 mode: the sweep left up to 4 (upper bound, row) pairs per tile
  const int* tile_ncand;
  long long* cand;
  int* ctrl;
  double* rec;
  long long row_offset, ptiles;
  double post_div;
  int s, cap, nblk, ptile;
  int tile_rounds;           // tiles per sweep wave = ceil(ptiles / (4 * nblk)): block b swept tiles 4b+w + 4*nblk*i
};

// same per-row arithmetic as bc_sweep.hip (sequential fma chain over k, bc_row_score epilogue)
template <int MODE>
__device__ __forceinline__ double bc_exact_score(const double* __restrict__ tiles, const double* __restrict__ v, long long r,
                                                 int S, double nr, double post_div) {
  const double* p = tiles + (size_t)(r >> 7) * S * BC_TILE + (r & (BC_TILE - 1));
  double a0 = 0., a1 = 0.;
  int k = 0;
  for (; k + 32 <= S; k += 32) {     // 32 independent loads in flight: this kernel is pure latency
    double x[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) x[u] = p[(size_t)(k + u) * BC_TILE];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (MODE == 0) {
        a0 = fma(x[u], v[2 * (k + u)], a0);
        a1 = fma(x[u], v[2 * (k + u) + 1], a1);
      } else {
        a0 = fma(x[u], v[k + u], a0);
      }
    }
  }
  for (; k < S; ++k) {
    const double x = p[(size_t)k * BC_TILE];
    if (MODE == 0) {
      a0 = fma(x, v[2 * k], a0);
      a1 = fma(x, v[2 * k + 1], a1);
    } else {
      a0 = fma(x, v[k], a0);
    }
  }
  if (MODE == 0) {
    const double s0 = a0 / nr, s1 = a1 / nr;
    const bool ok = (s1 > -1. + 1e-14) && (1. - s1 * s1 > 0.);
    const double den = ok ? sqrt(1. - s1 * s1) : INFINITY;
    return s0 / den;
  }
  return a0 / nr / post_div;
}

// The same score, computed by a whole wave for ONE row: the lanes fetch the row (and the sweep vectors) with one
// round of independent loads -- lane l holds elements l, l+64, ... -- and then every lane runs the identical
// sequential fma chain on broadcast values (v_readlane), so the result has the bits of bc_exact_score / k_sweep.
// With a handful of candidates this replaces ~4 dependent load batches per candidate by one.
// The same score, computed by a whole wave for ONE row: the lanes fetch the row and the sweep vectors with one
// round of independent loads -- lane l holds elements l, l+64, ... --, park them in a wave-private LDS strip
// and every lane then runs the identical sequential fma chain on broadcast LDS reads, so the result has the
// bits of bc_exact_score / k_sweep.  With a handful of candidates this replaces four dependent load batches and
// a one-lane chain by one round trip and a pipelined chain.  (Wave-private strip: LDS serves a wave's requests in
// order, the wavefront-scope fences only keep the compiler from reordering.)
template <int MODE>
__device__ __forceinline__ double bc_exact_score_wave(const double* __restrict__ tiles, const double* __restrict__ v, long long r,
                                                      int S, double nr, double post_div, double* strip /* [3][256] */) {
  const int lane = threadIdx.x & 63;
  const double* p = tiles + (size_t)(r >> 7) * S * BC_TILE + (r & (BC_TILE - 1));
  double* sx = strip;
  double* sa = strip + 256;
  double* sb = strip + 512;
  double a0 = 0., a1 = 0.;
  for (int base = 0; base < S; base += 256) {
    double x[4], vx[4], vy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = base + 64 * e + lane;
      const bool in = k < S;
      x[e] = in ? p[(size_t)k * BC_TILE] : 0.;
      if (MODE == 0) {
        vx[e] = in ? v[2 * k] : 0.;
        vy[e] = in ? v[2 * k + 1] : 0.;
      } else {
        vx[e] = in ? v[k] : 0.;
        vy[e] = 0.;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the previous block's reads come first
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sx[64 * e + lane] = x[e];
      sa[64 * e + lane] = vx[e];
      if (MODE == 0) sb[64 * e + lane] = vy[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n = (S - base) < 256 ? (S - base) : 256;
#pragma unroll 8
    for (int kk = 0; kk < n; ++kk) {
      const double xk = sx[kk];
      a0 = fma(xk, sa[kk], a0);
      if (MODE == 0) a1 = fma(xk, sb[kk], a1);
    }
  }
  if (MODE == 0) {
    const double s0 = a0 / nr, s1 = a1 / nr;
    const bool ok = (s1 > -1. + 1e-14) && (1. - s1 * s1 > 0.);
    const double den = ok ? sqrt(1. - s1 * s1) : INFINITY;
    return s0 / den;
  }
  return a0 / nr / post_div;
}

// In-launch hand-shake between block 0 (selection + rescoring) and the helper blocks (fp64 fallback).
//   sync[0]  verdict of launch `epoch`: 4*epoch + 1 = fall back, 4*epoch + 2 = done, nothing to do
//   sync[1]  arrivals of the helper blocks after their share of the fallback sweep (reset by block 0)
// Every wait is bounded, so the grid drains even if the protocol were broken (ctrl[2] records a timeout).
#define BC_RS_SPIN_LIMIT (1 << 24)
#define BC_RS_TILE_LIMIT16 64   // fp16 mode recomputes candidate tiles (~3 us each): past this the fp64 sweep is cheaper
typedef _Float16 bc_h2 __attribute__((ext_vector_type(2)));
// polling load: relaxed (an acquire per poll would invalidate caches 255 blocks x every poll and slow
// block 0 down); the one acquire fence follows once the awaited value has been seen
__device__ __forceinline__ unsigned bc_ld_poll(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one plane of the fp32 chain for a thread's two rows (k_rescore's recomputation of a candidate tile)
template <int MODE>
__device__ __forceinline__ void bc_rs_accumulate(bc_h2 x, const double* __restrict__ v, int k, float (&a0)[2], float (&a1)[2]) {
  if (MODE == 0) {
    const float vx = (float)v[2 * k], vy = (float)v[2 * k + 1];
    a0[0] = fmaf((float)x[0], vx, a0[0]);
    a1[0] = fmaf((float)x[0], vy, a1[0]);
    a0[1] = fmaf((float)x[1], vx, a0[1]);
    a1[1] = fmaf((float)x[1], vy, a1[1]);
  } else {
    const float vx = (float)v[k];
    a0[0] = fmaf((float)x[0], vx, a0[0]);
    a0[1] = fmaf((float)x[1], vx, a0[1]);
  }
}

// passes B + C: Lmax = max of the block lower bounds; candidates = rows whose upper bound reaches it (whole
// tiles are skipped through their maximum); exact fp64 rescoring; record.  Grid = 1 + helpers.
template <int MODE>
__global__ __launch_bounds__(256) void k_rescore(RescoreArgs a, bc_sweep_args sw, double* __restrict__ blk_val,
                                                long long* __restrict__ blk_idx, unsigned* __restrict__ sync,
                                                unsigned epoch) {
  __shared__ double sv[4];
  __shared__ long long si[4];
  __shared__ long long win;
  __shared__ int cnt;
  __shared__ int tcnt;
  __shared__ int tlist[1024];
  __shared__ unsigned verdict;
  __shared__ int bcnt;
  __shared__ int blist[64];                  // sweep blocks whose maximum upper bound reaches Lmax
  __shared__ long long scand[32];            // the first candidates, kept on chip (the usual case has 1-3)
  __shared__ double strips[4][3 * 256];      // bc_exact_score_wave: one strip per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;

  if (blockIdx.x != 0) {
    // ---- helper block: wait for block 0's verdict
    if (threadIdx.x == 0) {
      unsigned v = 0;
      for (int spin = 0; spin < BC_RS_SPIN_LIMIT; ++spin) {
        v = bc_ld_poll(sync);
        if ((v >> 2) == epoch) break;
        __builtin_amdgcn_s_sleep(32);
      }
      verdict = ((v >> 2) == epoch) ? (v & 3u) : 2u;
      if (verdict == 1u) __threadfence();
    }
    __syncthreads();
    if (verdict != 1u) return;
    double bv;
    long long bi;
    bc_sweep_block<MODE>(sw, blockIdx.x, gridDim.x, false, sv, si, bv, bi);
    if (threadIdx.x == 0) {
      blk_val[blockIdx.x] = bv;
      blk_idx[blockIdx.x] = bi;
      __threadfence();                         // release the candidate before arriving
      atomicAdd(sync + 1, 1u);
    }
    return;
  }

  // ---- block 0
  bool overflow = false;
  if (!skip) {
    double lmax = -INFINITY;
    float bu[4];                               // this thread's share of the block upper bounds (nblk <= 1024)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = threadIdx.x + q * 256;
      bu[q] = -INFINITY;
      if (i < a.nblk) {
        lmax = fmax(lmax, a.blk_l[i]);
        bu[q] = a.blk_u[i];
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lmax = fmax(lmax, __shfl_down(lmax, d, BC_WAVE));
    if (lane == 0) sv[wave] = lmax;
    if (threadIdx.x == 0) { cnt = 0; tcnt = 0; bcnt = 0; }
    __syncthreads();
    lmax = fmax(fmax(sv[0], sv[1]), fmax(sv[2], sv[3]));
    // phase B1: tiles whose maximum upper bound reaches Lmax -- first the sweep blocks whose maximum does
    // (usually one or two), then only the tiles those blocks walked
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (bu[q] != -INFINITY && (double)bu[q] >= lmax) {
        const int slot = atomicAdd(&bcnt, 1);
        if (slot < 64) blist[slot] = threadIdx.x + q * 256;
      }
    __syncthreads();
    const int nbl = bcnt;
    if (nbl <= 64) {
      const int per = 4 * a.tile_rounds, total = nbl * per;
      for (int i0 = 0; i0 < total; i0 += 8 * blockDim.x) {
        float tu[8];
        long long tt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = i0 + u * blockDim.x + threadIdx.x;
          tt[u] = -1;
          tu[u] = -INFINITY;
          if (idx < total) {
            const int b = blist[idx / per], q = idx % per;
            const long long t = (long long)b * 4 + (q & 3) + (long long)(q >> 2) * 4 * a.nblk;
            if (t < a.ptiles) { tt[u] = t; tu[u] = a.tile_u[t]; }
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (tt[u] >= 0 && tu[u] != -INFINITY && (double)tu[u] >= lmax) {
            const int slot = atomicAdd(&tcnt, 1);
            if (slot < 1024) tlist[slot] = (int)tt[u];
          }
      }
    } else {
      // many blocks in play: scan all per-tile maxima, 16 independent loads at a time
      for (long long t0 = 0; t0 < a.ptiles; t0 += 16LL * blockDim.x) {
        float tu[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const long long t = t0 + (long long)u * blockDim.x + threadIdx.x;
          tu[u] = t < a.ptiles ? a.tile_u[t] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (tu[u] != -INFINITY && (double)tu[u] >= lmax) {
            const int slot = atomicAdd(&tcnt, 1);
            if (slot < 1024) tlist[slot] = (int)(t0 + (long long)u * blockDim.x + threadIdx.x);
          }
      }
    }
    __syncthreads();
    const int ntl = tcnt;
    if (threadIdx.x == 0) bcnt = 0;            // reused by the int8 branch below
    __syncthreads();
    overflow = ntl > (a.u16 ? BC_RS_TILE_LIMIT16 : 1024);      // too many tiles in play
    if (!overflow && a.tile_cand) {
      // phase B2 (int8 mirror): the pairs the sweep left for each such tile; a tile with more than four local
      // candidates hands over all of its rows
      for (int q = threadIdx.x; q < ntl; q += blockDim.x) {
        const long long t = tlist[q];
        const int n = a.tile_ncand[t];
        float2 prs[4];                           // fetched together with the count: one round trip
#pragma unroll
        for (int i = 0; i < 4; ++i) prs[i] = a.tile_cand[t * 4 + i];
        if (n <= 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float2 pr = prs[i];
            if (i < n && (double)pr.x >= lmax) {
              const int slot = atomicAdd(&cnt, 1);
              const long long row = t * a.ptile + (int)pr.y;
              if (slot < a.cap) a.cand[slot] = row;
              if (slot < 32) scand[slot] = row;
            }
          }
        } else {
          const int o = atomicAdd(&bcnt, 1);      // (bcnt is free again after phase B1)
          if (o < 64) blist[o] = (int)t;
        }
      }
      __syncthreads();
      const int no = bcnt;
      overflow = no > 64;
      if (!overflow) {
        for (int o = 0; o < no; ++o) {
          const long long row = (long long)blist[o] * a.ptile + threadIdx.x;      // ptile == blockDim.x == 256
          if (row < sw.n_rows && a.norms[row] != 0.) {
            const int slot = atomicAdd(&cnt, 1);
            if (slot < a.cap) a.cand[slot] = row;
            if (slot < 32) scand[slot] = row;
          }
        }
        __syncthreads();
        overflow = cnt > a.cap;
      }
    } else if (!overflow && a.u16) {
      // phase B2 (fp16 mirror): recompute the per-row intervals of each such tile from the mirror -- the sweep
      // wrote none.  Thread = two adjacent rows of the tile (one 4-byte load per plane, 1 KiB per plane and
      // block), up to 64 planes in flight; fp32 chain, same interval formula and delta as the sweep.
      const double delta = (MODE == 0) ? a.delta : a.delta * (*a.v_norm);
      for (int q = 0; q < ntl; ++q) {
        const long long t = tlist[q];
        const bc_h2* __restrict__ tp = reinterpret_cast<const bc_h2*>(a.u16 + (size_t)t * a.sp * BC_HTILE) + threadIdx.x;
        float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
        // sp is a multiple of BC_HU = 10: batches of 50 planes (all loads of a batch in flight), then of 10
        int k0 = 0;
        for (; k0 + 50 <= a.sp; k0 += 50) {
          bc_h2 x[50];
#pragma unroll
          for (int u = 0; u < 50; ++u) x[u] = tp[(size_t)(k0 + u) * (BC_HTILE / 2)];
#pragma unroll
          for (int u = 0; u < 50; ++u) bc_rs_accumulate<MODE>(x[u], a.v, k0 + u, a0, a1);
        }
        for (; k0 < a.sp; k0 += BC_HU) {
          bc_h2 x[BC_HU];
#pragma unroll
          for (int u = 0; u < BC_HU; ++u) x[u] = tp[(size_t)(k0 + u) * (BC_HTILE / 2)];
#pragma unroll
          for (int u = 0; u < BC_HU; ++u) bc_rs_accumulate<MODE>(x[u], a.v, k0 + u, a0, a1);
        }
        const unsigned lv = a.live[t * 64 + (threadIdx.x >> 2)];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = 2 * threadIdx.x + j;                    // row within the tile
          if ((lv >> (i & 7)) & 1u) {
            double Ub, Lb;
            bc_score_interval<MODE>((double)a0[j], (double)a1[j], delta, a.post_div, Ub, Lb);
            if (Ub >= lmax) {
              const int slot = atomicAdd(&cnt, 1);
              if (slot < a.cap) a.cand[slot] = t * BC_HTILE + i;
              if (slot < 32) scand[slot] = t * BC_HTILE + i;
            }
          }
        }
      }
      __syncthreads();
      overflow = cnt > a.cap;
    } else if (!overflow) {
      // phase B2 (fp32 mirror): one wave per such tile, all of the tile's stored upper bounds in flight at once
      for (int q = wave; q < ntl; q += 4) {
        const long long t = tlist[q];
        float u8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int i = lane + 64 * e;
          u8[e] = i < a.ptile ? a.ub[t * a.ptile + i] : -INFINITY;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (u8[e] != -INFINITY && (double)u8[e] >= lmax) {
            const int slot = atomicAdd(&cnt, 1);
            if (slot < a.cap) a.cand[slot] = t * a.ptile + lane + 64 * e;
            if (slot < 32) scand[slot] = t * a.ptile + lane + 64 * e;
          }
      }
      __syncthreads();
      overflow = cnt > a.cap;
    }
  }
  // verdict for the helpers (block-uniform: tcnt / cnt are shared)
  if (threadIdx.x == 0) {
    a.ctrl[1] = overflow ? 1 : 0;              // observable: the last launch fell back
    if (overflow) a.ctrl[3] += 1;              // ... and how often since creation
    if (!skip) {                               // diagnostics: sweeps and candidates rescored since creation
      unsigned long long* st = reinterpret_cast<unsigned long long*>(a.ctrl + 4);
      st[0] += 1;
      st[1] += overflow ? 0 : (unsigned long long)cnt;
    }
    if (gridDim.x > 1) {
      __threadfence();
      __hip_atomic_store(sync, 4u * epoch + (overflow ? 1u : 2u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (skip) return;

  if (overflow) {
    // ---- full fp64 sweep by the whole grid, block 0 merges
    double bv;
    long long bi;
    bc_sweep_block<MODE>(sw, 0, gridDim.x, false, sv, si, bv, bi);
    if (threadIdx.x == 0) {
      blk_val[0] = bv;
      blk_idx[0] = bi;
      int ok = 1;
      if (gridDim.x > 1) {
        ok = 0;
        for (int spin = 0; spin < BC_RS_SPIN_LIMIT; ++spin) {
          if (bc_ld_poll(sync + 1) == gridDim.x - 1) { ok = 1; break; }
          __builtin_amdgcn_s_sleep(8);
        }
        sync[1] = 0;                            // ready for the next launch (stream-ordered)
      }
      if (!ok) a.ctrl[2] = 1;
      __threadfence();
    }
    __syncthreads();
    bc_emit_record(blk_val, blk_idx, (int)gridDim.x, a.tiles, a.norms, a.s, a.row_offset, false, a.rec, sv, si, &win);
    return;
  }

  const int count = cnt;
  double bv = -INFINITY;
  long long bi = LLONG_MAX;
  if (count <= 32) {
    // the usual case, a handful of candidates: a wave per candidate
    for (int j = wave; j < count; j += 4) {
      const long long r = scand[j];
      const double sc = bc_exact_score_wave<MODE>(a.tiles, a.v, r, a.s, a.norms[r], a.post_div, strips[wave]);
      const long long gi = a.row_offset + r;
      if (bc_better(sc, gi, bv, bi)) { bv = sc; bi = gi; }
    }
  } else {
    for (int j = threadIdx.x; j < count; j += blockDim.x) {
      const long long r = a.cand[j];
      const double sc = bc_exact_score<MODE>(a.tiles, a.v, r, a.s, a.norms[r], a.post_div);
      const long long gi = a.row_offset + r;
      if (bc_better(sc, gi, bv, bi)) { bv = sc; bi = gi; }
    }
  }
  bc_wave_argmax(bv, bi);
  __syncthreads();
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (bc_better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
    const bool valid = bi != LLONG_MAX;
    a.rec[0] = bv;
    reinterpret_cast<long long*>(a.rec)[1] = valid ? bi : -1;
    a.rec[2] = valid ? a.norms[bi - a.row_offset] : 0.0;
    a.rec[3] = valid ? 1.0 : 0.0;
    win = valid ? bi - a.row_offset : -1;
  }
  __syncthreads();
  const long long r = win;
  for (int k = threadIdx.x; k < a.s; k += blockDim.x) a.rec[BC_REC_HDR + k] = (r >= 0) ? a.tiles[bc_tile_off(r, k, a.s)] : 0.0;
}

// u32 tile builder: one block per 256-row tile, thread = row
__global__ __launch_bounds__(256) void k_build_u32(const double* __restrict__ tiles, const double* __restrict__ norms,
                                                  long long n_rows, int S, float* __restrict__ u32) {
  const long long t = blockIdx.x;
  const long long r = t * BC_PTILE + threadIdx.x;
  const bool live = r < n_rows;
  const double nr = live ? norms[r] : 0.;
  const double* p = tiles + (size_t)(r >> 7) * S * BC_TILE + (r & (BC_TILE - 1));
  float* q = u32 + (size_t)t * S * BC_PTILE + threadIdx.x;
  for (int k = 0; k < S; ++k) {
    float u = 0.f;
    if (live && nr != 0.) u = (float)(p[(size_t)k * BC_TILE] / nr);
    q[(size_t)k * BC_PTILE] = u;
  }
}

// ------------------------------------------------------------------ host side
int bc_pref_create(bc_phi* phi, int prec, bc_pref** out) {
  bc_ctx* ctx = phi->ctx;
  bc_pref* p = new bc_pref();
  p->ctx = ctx;
  p->phi = phi;
  p->prec = prec == 16 ? 16 : (prec == 8 ? 8 : 32);
  if (p->prec == 8 && (phi->s + 3) / 4 > BC_IMAXG - BC_IU) p->prec = 16;      // the digit table of k_sweep_i8 holds S <= ~1260
  p->ptile = p->prec == 16 ? BC_HTILE : (p->prec == 8 ? BC_ITILE : BC_PTILE);
  p->sp = p->prec == 16 ? (phi->s + BC_HU - 1) / BC_HU * BC_HU : phi->s;
  p->sp4 = ((phi->s + 3) / 4 + BC_IU - 1) / BC_IU * BC_IU;
  p->ptiles = (phi->n_rows + p->ptile - 1) / p->ptile;
  if (p->ptiles < 1) p->ptiles = 1;
  // one wave per tile and a grid-stride loop: size the grid so that all its waves are resident at once
  // (4 per SIMD) and every wave walks the same number of tiles -- a 2x over-subscribed grid left waves with
  // 2 or 3 tiles each (79% balance at 10M rows)
  // fp16 / fp32 sweeps: 4 waves per SIMD fit (<= 128 VGPRs); the int8 sweep holds 148 VGPRs: 3 per SIMD
  long long wmax = (long long)ctx->n_cu * (p->prec == 8 ? 12 : 16);
  const char* genv = getenv("BC_PREF_WAVES_PER_CU");
  if (genv && atoi(genv) > 0) wmax = (long long)ctx->n_cu * atoi(genv);
  const long long rounds = (p->ptiles + wmax - 1) / wmax;
  const long long waves = (p->ptiles + rounds - 1) / rounds;
  p->grid = (int)((waves + 3) / 4);
  if (p->grid < 1) p->grid = 1;
  if (p->grid > 1024) p->grid = 1024;       // k_rescore keeps the block bounds in 4 registers per thread
  // the fp64 tiles cover ntiles*128 rows; the unit rows cover ptiles*ptile >= that, reads past the fp64 tiles are masked by `live`
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_u = take(p->prec == 8 ? (size_t)p->ptiles * p->sp4 * BC_ITILE * sizeof(int)
                                       : (size_t)p->ptiles * p->sp * p->ptile * (p->prec == 16 ? sizeof(_Float16) : sizeof(float)));
  const size_t o_ub = take(p->prec != 32 ? 0 : (size_t)p->ptiles * p->ptile * sizeof(float));
  const size_t o_rq = take(p->prec == 8 ? (size_t)p->ptiles * BC_ITILE * sizeof(bc_hq2) : 0);
  const size_t o_tc = take(p->prec == 8 ? (size_t)p->ptiles * 4 * sizeof(float2) : 0);
  const size_t o_tn = take(p->prec == 8 ? (size_t)p->ptiles * sizeof(int) : 0);
  const size_t o_lv = take(p->prec == 16 ? (size_t)p->ptiles * 64 : 0);
  const size_t o_tu = take((size_t)p->ptiles * sizeof(float));
  const size_t o_bl = take((size_t)p->grid * sizeof(double));
  const size_t o_bu = take((size_t)p->grid * sizeof(float));
  const size_t o_c = take((size_t)p->cap * sizeof(long long));
  const size_t o_ctrl = take(256);
  const size_t o_sync = take(256);
  hipError_t e = hipMalloc(&p->slab, off);
  if (e != hipSuccess) { delete p; return bc_hip_fail(e, "hipMalloc(prefilter)", __FILE__, __LINE__); }
  char* base = (char*)p->slab;
  p->u32 = p->prec == 32 ? (float*)(base + o_u) : nullptr;
  p->u16 = p->prec == 16 ? (_Float16*)(base + o_u) : nullptr;
  p->ub = p->prec != 32 ? nullptr : (float*)(base + o_ub);
  if (p->prec == 8) {
    p->u8 = (int*)(base + o_u);
    p->rowq = (bc_hq2*)(base + o_rq);
    p->tile_cand = (float2*)(base + o_tc);
    p->tile_ncand = (int*)(base + o_tn);
  }
  p->live = p->prec == 16 ? (unsigned char*)(base + o_lv) : nullptr;
  p->tile_u = (float*)(base + o_tu);
  p->blk_l = (double*)(base + o_bl);
  p->blk_u = (float*)(base + o_bu);
  p->cand = (long long*)(base + o_c);
  p->ctrl = (int*)(base + o_ctrl);
  p->sync = (unsigned*)(base + o_sync);
  // helpers: one block per CU, all resident next to block 0; never more than the fp64 block-candidate arrays hold
  p->helpers_grid = ctx->n_cu < phi->sweep_blocks ? ctx->n_cu : phi->sweep_blocks;
  { const char* hg = getenv("BC_PREF_HELPERS"); if (hg && atoi(hg) >= 1 && atoi(hg) < p->helpers_grid) p->helpers_grid = atoi(hg); }
  if ((long long)p->helpers_grid * 4 > phi->ntiles) p->helpers_grid = (int)((phi->ntiles + 3) / 4);
  if (p->helpers_grid < 1) p->helpers_grid = 1;
  e = hipMemsetAsync(p->ctrl, 0, 512, ctx->stream);
  if (e == hipSuccess) {
    if (p->prec == 8)
      hipLaunchKernelGGL(k_build_i8, dim3((unsigned)p->ptiles), dim3(256), 0, ctx->stream, phi->tiles, phi->norms,
                         (long long)phi->n_rows, phi->s, p->sp4, p->u8, p->rowq);
    else if (p->prec == 16)
      hipLaunchKernelGGL(k_build_u16, dim3((unsigned)p->ptiles), dim3(256), 0, ctx->stream, phi->tiles, phi->norms,
                         (long long)phi->n_rows, phi->s, p->sp, p->u16, p->live);
    else
      hipLaunchKernelGGL(k_build_u32, dim3((unsigned)p->ptiles), dim3(256), 0, ctx->stream, phi->tiles, phi->norms,
                         (long long)phi->n_rows, phi->s, p->u32);
    e = hipGetLastError();
  }
  if (e != hipSuccess) { (void)hipFree(p->slab); delete p; return bc_hip_fail(e, "prefilter build", __FILE__, __LINE__); }
  *out = p;
  return BC_OK;
}

void bc_pref_destroy(bc_pref* p) {
  if (!p) return;
  if (p->slab) (void)hipFree(p->slab);
  delete p;
}

const int* bc_pref_ctrl(const bc_pref* p) { return p->ctrl; }
void bc_pref_set_cap(bc_pref* p, int cap) { if (cap >= 1 && cap <= 4096) p->cap = cap; }
int bc_pref_precision(const bc_pref* p) { return p->prec; }

// passes A, B, C (and the in-launch fp64 fallback)
int bc_pref_launch(bc_pref* p, int mode, const double* v_dev, const double* v_norm_dev, double post_div,
                   const int* skip_flag, double* rec_dev) {
  bc_ctx* ctx = p->ctx;
  bc_phi* phi = p->phi;
  PrefArgs a;
  a.u32 = p->u32;
  a.u16 = p->u16;
  a.live = p->live;
  a.delta = p->prec == 16 ? bc_pref_delta16(phi->s) : BC_PREF_DELTA;
  a.sp = p->sp;
  a.norms = phi->norms;
  a.v = v_dev;
  a.skip_flag = skip_flag;
  a.v_norm = v_norm_dev;
  a.ub = p->ub;
  a.tile_u = p->tile_u;
  a.blk_l = p->blk_l;
  a.blk_u = p->blk_u;
  a.n_rows = phi->n_rows;
  a.ptiles = p->ptiles;
  a.post_div = post_div;
  a.s = phi->s;
  int rc = bc_timer_begin(ctx, 0);
  if (rc) return rc;
  if (p->prec == 8) {
    I8Args ia;
    ia.u8 = p->u8;
    ia.rowq = p->rowq;
    ia.v = v_dev;
    ia.skip_flag = skip_flag;
    ia.v_norm = v_norm_dev;
    ia.tile_u = p->tile_u;
    ia.tile_cand = p->tile_cand;
    ia.tile_ncand = p->tile_ncand;
    ia.blk_l = p->blk_l;
    ia.blk_u = p->blk_u;
    ia.ptiles = p->ptiles;
    ia.post_div = post_div;
    ia.s = phi->s;
    ia.sp4 = p->sp4;
    if (mode == 0) hipLaunchKernelGGL(k_sweep_i8<0>, dim3(p->grid), dim3(256), 0, ctx->stream, ia);
    else hipLaunchKernelGGL(k_sweep_i8<1>, dim3(p->grid), dim3(256), 0, ctx->stream, ia);
  } else if (p->prec == 16) {
    if (mode == 0) hipLaunchKernelGGL(k_sweep_f16<0>, dim3(p->grid), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_sweep_f16<1>, dim3(p->grid), dim3(256), 0, ctx->stream, a);
  } else {
    if (mode == 0) hipLaunchKernelGGL(k_sweep_f32<0>, dim3(p->grid), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_sweep_f32<1>, dim3(p->grid), dim3(256), 0, ctx->stream, a);
  }
  BC_HIP(hipGetLastError());
  rc = bc_timer_end(ctx, 0);
  if (rc) return rc;
  RescoreArgs r;
  r.tiles = phi->tiles;
  r.norms = phi->norms;
  r.v = v_dev;
  r.skip_flag = skip_flag;
  r.ub = p->ub;
  r.tile_u = p->tile_u;
  r.blk_l = p->blk_l;
  r.blk_u = p->blk_u;
  r.u16 = p->u16;
  r.tile_cand = p->tile_cand;
  r.tile_ncand = p->tile_ncand;
  r.live = p->live;
  r.v_norm = v_norm_dev;
  r.delta = a.delta;
  r.sp = p->sp;
  r.tile_rounds = (int)((p->ptiles + 4LL * p->grid - 1) / (4LL * p->grid));
  r.cand = p->cand;
  r.ctrl = p->ctrl;
  r.rec = rec_dev;
  r.row_offset = phi->row_offset;
  r.ptiles = p->ptiles;
  r.post_div = post_div;
  r.s = phi->s;
  r.cap = p->cap;
  r.nblk = p->grid;
  r.ptile = p->ptile;
  bc_sweep_args sw;
  sw.tiles = phi->tiles;
  sw.norms = phi->norms;
  sw.v = v_dev;
  sw.skip_flag = nullptr;
  sw.n_rows = phi->n_rows;
  sw.ntiles = phi->ntiles;
  sw.row_offset = phi->row_offset;
  sw.post_div = post_div;
  sw.s = phi->s;
  p->epoch = (p->epoch + 1) & 0x3fffffffu;
  if (p->epoch == 0) p->epoch = 1;            // sync[0] starts at 0: epoch 0 would match before any verdict
  const int g = p->helpers_grid;
  if (mode == 0) hipLaunchKernelGGL(k_rescore<0>, dim3(g), dim3(256), 0, ctx->stream, r, sw, phi->blk_val, phi->blk_idx, p->sync, p->epoch);
  else hipLaunchKernelGGL(k_rescore<1>, dim3(g), dim3(256), 0, ctx->stream, r, sw, phi->blk_val, phi->blk_idx, p->sync, p->epoch);
  BC_HIP(hipGetLastError());
  return BC_OK;
}
