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
#include "bc_sweep_dev.h"
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256) void k_sweep(bc_sweep_args a, double* __restrict__ blk_val,
                                              long long* __restrict__ blk_idx) {
  __shared__ double sv[4];
  __shared__ long long si[4];
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;
  double best_v;
  long long best_i;
  bc_sweep_block<MODE>(a, blockIdx.x, gridDim.x, skip, sv, si, best_v, best_i);
  if (threadIdx.x == 0) {
    blk_val[blockIdx.x] = best_v;
    blk_idx[blockIdx.x] = best_i;
  }
}

// (An in-launch "last block reduces" form of the winner was measured and dropped for the full-grid sweep:
// the agent-scope release each of the 2048 blocks needs before its arrival costs more than the launch it
// saves -- sweep 156 -> 243 us at 1.25M rows.  profiles/r01_notes.md)
__global__ __launch_bounds__(256) void k_local_winner(const double* __restrict__ blk_val,
                                                     const long long* __restrict__ blk_idx, int nblk,
                                                     const double* __restrict__ tiles,
                                                     const double* __restrict__ norms, int s, long long row_offset,
                                                     const int* skip_flag, double* __restrict__ rec) {
  __shared__ double sv[4];
  __shared__ long long si[4];
  __shared__ long long win;
  const bool skip = skip_flag != nullptr && *skip_flag != 0;
  bc_emit_record(blk_val, blk_idx, nblk, tiles, norms, s, row_offset, skip, rec, sv, si, &win);
}

// host-side launcher shared by bc_phi_argmax and the solver loop
// rec_dev == nullptr: sweep only (the caller reduces p->blk_val / p->blk_idx itself)
int bc_launch_sweep(bc_phi* p, int mode, const double* v_dev, double post_div, const int* skip_flag, double* rec_dev) {
  bc_ctx* ctx = p->ctx;
  bc_sweep_args a;
  a.tiles = p->tiles;
  a.norms = p->norms;
  a.v = v_dev;
  a.skip_flag = skip_flag;
  a.n_rows = p->n_rows;
  a.ntiles = p->ntiles;
  a.row_offset = p->row_offset;
  a.post_div = post_div;
  a.s = p->s;
  int rc = bc_timer_begin(ctx, 0);
  if (rc) return rc;
  const int grid = p->sweep_blocks;
  if (mode == 0) hipLaunchKernelGGL(k_sweep<0>, dim3(grid), dim3(256), 0, ctx->stream, a, p->blk_val, p->blk_idx);
  else hipLaunchKernelGGL(k_sweep<1>, dim3(grid), dim3(256), 0, ctx->stream, a, p->blk_val, p->blk_idx);
  BC_HIP(hipGetLastError());
  rc = bc_timer_end(ctx, 0);
  if (rc) return rc;
  if (!rec_dev) return BC_OK;
  rc = bc_timer_begin(ctx, 3);
  if (rc) return rc;
  hipLaunchKernelGGL(k_local_winner, dim3(1), dim3(256), 0, ctx->stream, p->blk_val, p->blk_idx, grid,
                     p->tiles, p->norms, p->s, (long long)p->row_offset, skip_flag, rec_dev);
  BC_HIP(hipGetLastError());
  return bc_timer_end(ctx, 3);
}

extern "C" int bc_phi_argmax(bc_phi* p, int mode, const double* v, double post_div, int64_t* best, double* score) {
  if (!p || !v || (mode != 0 && mode != 1)) { bc_set_error("bc_phi_argmax: bad argument"); return BC_INVALID_ARGUMENT; }
  bc_ctx* ctx = p->ctx;
  const size_t nv = (size_t)(mode == 0 ? 2 : 1) * p->s;
  BC_HIP(hipMemcpyAsync(p->vbuf, v, nv * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  int rc = bc_launch_sweep(p, mode, p->vbuf, post_div, nullptr, p->rec);
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
