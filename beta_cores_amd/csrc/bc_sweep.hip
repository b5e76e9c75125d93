// K3: fused score + argmax sweep over the tiled Phi (the per-iteration hot kernel).
//
// Replaces, in one streaming pass over Phi:
//   giga.py:31-38        scorends = An.T.dot([cdir, xw]); mask; sqrt; divide; argmax
//   frankwolfe.py:16-17  (An.T.dot(residual)).argmax()
//   orthopursuit.py:18-24 dots = An.T.dot(residual); dots.argmax()
//   bcores.py:78-81      corrs = vecs.dot(resid)/rownorm/S ; argmax
//
// Mapping to the hardware: one 64-lane wave owns one 128-row tile at a time; lane l
// owns rows 2l, 2l+1 of the tile, so sample k of the tile is one coalesced 1 KiB
// load (16 B per lane).  The S-vector(s) are wave-uniform and come through the
// scalar cache.  No LDS and no cross-lane traffic inside the sweep; lanes keep a
// running (score, row) best and the block reduces once at the end.
// Algorithmic traffic: 8*S*128 B of Phi + 8*128 B of norms per tile.
#include "bc_internal.h"
#include <climits>
#include <cmath>
#include <cstdlib>

struct bc_sweep_args {
  const double* tiles;
  const double* norms;
  const double* v;          // mode 0: [S][2] (cdir, xw) interleaved; mode 1: [S]
  const int* skip_flag;     // optional device flag: when non-zero the sweep is a no-op
  const int* run_flag;      // optional device flag: when given and ZERO the sweep is a no-op (pre-filter fallback)
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

// FUSED: the last block to arrive also merges the block winners and emits the candidate record.  Only the
// gated pre-filter fallback uses it (one launch instead of two when -- almost always -- it has nothing to
// do); with the full-size grid of a regular sweep the per-block release costs more than the launch it saves.
template <int MODE, bool FUSED>
__global__ __launch_bounds__(256) void k_sweep(bc_sweep_args a, double* __restrict__ blk_val,
                                              long long* __restrict__ blk_idx, unsigned* __restrict__ arrivals,
                                              double* __restrict__ rec) {
  __shared__ double sv[4];
  __shared__ long long si[4];
  __shared__ long long win;
  __shared__ int is_last;
  if (a.run_flag != nullptr && *a.run_flag == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double best_v = -INFINITY;
  long long best_i = LLONG_MAX;
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;
  if (!skip) {
    const int S = a.s;
    const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.v);
    const double* __restrict__ v1 = a.v;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < a.ntiles; t += (long long)gridDim.x * 4) {
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
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (bc_better(sv[w], si[w], best_v, best_i)) { best_v = sv[w]; best_i = si[w]; }
    blk_val[blockIdx.x] = best_v;
    blk_idx[blockIdx.x] = best_i;
    if (FUSED) {
      __threadfence();                                   // release this block's candidate (agent scope)
      const unsigned prev = atomicAdd(arrivals, 1u);
      is_last = prev == gridDim.x - 1;
    }
  }
  if (FUSED) {
    __syncthreads();
    if (!is_last) return;
    __threadfence();                                     // acquire the other blocks' candidates
    if (threadIdx.x == 0) *arrivals = 0;                 // ready for the next launch (stream-ordered)
    bc_emit_record(blk_val, blk_idx, (int)gridDim.x, a.tiles, a.norms, a.s, a.row_offset, skip, rec, sv, si, &win);
  }
}

// (For the regular, full-grid sweep the in-launch "last block reduces" form was measured and dropped: the
// agent-scope release each of the 2048 blocks needs before its arrival costs more than the launch it
// saves -- sweep 156 -> 243 us at 1.25M rows.  profiles/r01_notes.md)
__global__ __launch_bounds__(256) void k_local_winner(const double* __restrict__ blk_val,
                                                     const long long* __restrict__ blk_idx, int nblk,
                                                     const double* __restrict__ tiles,
                                                     const double* __restrict__ norms, int s, long long row_offset,
                                                     const int* skip_flag, const int* run_flag, double* __restrict__ rec) {
  __shared__ double sv[4];
  __shared__ long long si[4];
  __shared__ long long win;
  if (run_flag != nullptr && *run_flag == 0) return;
  const bool skip = skip_flag != nullptr && *skip_flag != 0;
  bc_emit_record(blk_val, blk_idx, nblk, tiles, norms, s, row_offset, skip, rec, sv, si, &win);
}

// host-side launcher shared by bc_phi_argmax and the solver loop
// rec_dev == nullptr: sweep only (the caller reduces p->blk_val / p->blk_idx itself)
int bc_launch_sweep(bc_phi* p, int mode, const double* v_dev, double post_div, const int* skip_flag, double* rec_dev,
                    const int* run_flag) {
  bc_ctx* ctx = p->ctx;
  bc_sweep_args a;
  a.tiles = p->tiles;
  a.norms = p->norms;
  a.v = v_dev;
  a.skip_flag = skip_flag;
  a.run_flag = run_flag;
  a.n_rows = p->n_rows;
  a.ntiles = p->ntiles;
  a.row_offset = p->row_offset;
  a.post_div = post_div;
  a.s = p->s;
  // (a gated fallback launch is a no-op almost always: keep it out of the K3 timer)
  int rc = run_flag ? BC_OK : bc_timer_begin(ctx, 0);
  if (rc) return rc;
  // a gated launch almost never runs: one block per CU keeps its no-op cost at ~1.5 us (it is slower when it does run)
  int grid = p->sweep_blocks;
  if (run_flag && grid > ctx->n_cu) grid = ctx->n_cu;
  if (run_flag && rec_dev) {       // gated fallback: sweep + winner + record in one launch
    if (mode == 0)
      hipLaunchKernelGGL((k_sweep<0, true>), dim3(grid), dim3(256), 0, ctx->stream, a, p->blk_val, p->blk_idx, p->sweep_counter, rec_dev);
    else
      hipLaunchKernelGGL((k_sweep<1, true>), dim3(grid), dim3(256), 0, ctx->stream, a, p->blk_val, p->blk_idx, p->sweep_counter, rec_dev);
    BC_HIP(hipGetLastError());
    return BC_OK;
  }
  if (mode == 0)
    hipLaunchKernelGGL((k_sweep<0, false>), dim3(grid), dim3(256), 0, ctx->stream, a, p->blk_val, p->blk_idx, nullptr, nullptr);
  else
    hipLaunchKernelGGL((k_sweep<1, false>), dim3(grid), dim3(256), 0, ctx->stream, a, p->blk_val, p->blk_idx, nullptr, nullptr);
  BC_HIP(hipGetLastError());
  rc = run_flag ? BC_OK : bc_timer_end(ctx, 0);
  if (rc) return rc;
  if (!rec_dev) return BC_OK;
  hipLaunchKernelGGL(k_local_winner, dim3(1), dim3(256), 0, ctx->stream, p->blk_val, p->blk_idx, grid,
                     p->tiles, p->norms, p->s, (long long)p->row_offset, skip_flag, run_flag, rec_dev);
  BC_HIP(hipGetLastError());
  return BC_OK;
}

extern "C" int bc_phi_argmax(bc_phi* p, int mode, const double* v, double post_div, int64_t* best, double* score) {
  if (!p || !v || (mode != 0 && mode != 1)) { bc_set_error("bc_phi_argmax: bad argument"); return BC_INVALID_ARGUMENT; }
  bc_ctx* ctx = p->ctx;
  const size_t nv = (size_t)(mode == 0 ? 2 : 1) * p->s;
  BC_HIP(hipMemcpyAsync(p->vbuf, v, nv * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  int rc = bc_launch_sweep(p, mode, p->vbuf, post_div, nullptr, p->rec, nullptr);
  if (rc) return rc;
  BC_HIP(hipMemcpyAsync(ctx->pinned, p->rec, BC_REC_HDR * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  if (best) *best = reinterpret_cast<long long*>(ctx->pinned)[1];
  if (score) *score = ctx->pinned[0];
  return BC_OK;
}

// scores[i] = Phi[i,:].v  (test / debugging aid; same access pattern as the sweep)
__global__ __launch_bounds__(256) void k_matvec(const double* __restrict__ tiles, const double* __restrict__ v, int s,
                                               long long ntiles, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += (long long)gridDim.x * 4) {
    const double2* __restrict__ p = reinterpret_cast<const double2*>(tiles + (size_t)t * s * BC_TILE) + lane;
    double a0 = 0., a1 = 0.;
    for (int k = 0; k < s; ++k) {
      const double2 x = p[(size_t)k * 64];
      a0 = fma(x.x, v[k], a0);
      a1 = fma(x.y, v[k], a1);
    }
    reinterpret_cast<double2*>(out)[(size_t)t * 64 + lane] = make_double2(a0, a1);
  }
}

extern "C" int bc_phi_matvec(bc_phi* p, const double* v, double* out) {
  if (!p || !v || (!out && p->n_rows)) { bc_set_error("bc_phi_matvec: bad argument"); return BC_INVALID_ARGUMENT; }
  if (p->n_rows == 0) return BC_OK;
  bc_ctx* ctx = p->ctx;
  double* dout = nullptr;
  BC_HIP(hipMalloc((void**)&dout, (size_t)p->ntiles * BC_TILE * sizeof(double)));
  hipError_t e = hipMemcpyAsync(p->vbuf, v, (size_t)p->s * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_matvec, dim3(p->sweep_blocks), dim3(256), 0, ctx->stream, p->tiles, p->vbuf, p->s,
                       (long long)p->ntiles, dout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)p->n_rows * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(dout);
  if (e != hipSuccess) return bc_hip_fail(e, "bc_phi_matvec", __FILE__, __LINE__);
  return BC_OK;
}
