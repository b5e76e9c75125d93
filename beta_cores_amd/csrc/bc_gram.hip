// K4: weighted Gram matrix  Z^T diag(w) Z  of the data rows z = [x (D), y], from which the
// posterior update takes  X^T W X  (D x D) and  X^T (w*y)  (D)   (model_linreg.py:29,31 ==
// model_neurlinr.py:118,120:  (w[:,None]*X).T.dot(X)  and  (w[:,None]*Y[:,None]*X).sum(axis=0)).
//
// This is the one genuinely GEMM-shaped reduction on the path (2*N*Dz^2 flop against 8*N*Dz
// bytes: intensity Dz/4 flop/B), so it runs on the fp64 matrix cores: v_mfma_f64_16x16x4_f64
// with A = (w*Z)^T panel, B = Z panel, both staged through LDS in 16-row slabs (buffer loads,
// register prefetch of the next slab).  Only the upper-triangular BT x BT tiles are computed;
// the row range is split over the grid (split-K) and the per-split partial tiles are summed in
// split order by a second kernel, so the result is run-to-run deterministic.
#include "bc_internal.h"
#include <cmath>
#include <cstring>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

struct GramArgs {
  const double* z;        // [n_rows][dz]
  const double* w;        // [n_rows] or null (all ones)
  double* partial;        // [splits][ntri][BT][BT]
  double* partial_y;      // [splits][nt*BT]   X^T (w*y), accumulated by the diagonal-tile blocks
  long long n_rows;
  long long rows_per_split;   // multiple of KR
  int dz, d, nt;          // d = number of x columns (y is column d), nt = number of BT-wide column tiles over d
  int ntri;               // nt*(nt+1)/2
};

template <int BT>
__global__ __launch_bounds__(256, 2) void k_gram(GramArgs a) {
  constexpr int KR = 16;                 // rows per LDS slab
  constexpr int LDX = BT + 16;           // row stride == 16 (mod 32) doubles: conflict-free 2-row x 16-col reads
  constexpr int MT = BT / 32;            // MFMA tiles per wave per dimension (wave tile = BT/2 x BT/2)
  constexpr int LP = (KR * BT) / 256;    // 8-byte loads per thread per panel per slab
  __shared__ double Al[KR * LDX];
  __shared__ double Bl[KR * LDX];
  __shared__ double Yl[256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int wa = wv >> 1, wb = wv & 1;   // wave position inside the block tile
  // blockIdx.x = split * ntri + tri  (tiles of one split are neighbours => they share rows in L2)
  const int tri = blockIdx.x % a.ntri;
  const long long split = blockIdx.x / a.ntri;
  int ta = 0, rem = tri;                 // tri -> (ta <= tb)
  while (rem >= a.nt - ta) { rem -= a.nt - ta; ++ta; }
  const int tb = ta + rem;
  const int a0 = ta * BT, b0 = tb * BT;
  const long long r_begin = split * a.rows_per_split;
  long long r_end = r_begin + a.rows_per_split;
  if (r_end > a.n_rows) r_end = a.n_rows;

  double4_t acc[MT][MT];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < MT; ++y) acc[x][y] = (double4_t){0., 0., 0., 0.};

  // staging map: thread -> (row lr = idx / BT, col lc = idx % BT) of the slab, LP passes
  const int lc = tid % BT, lr0 = tid / BT;
  constexpr int RPP = 256 / BT;          // rows covered per pass
  const bool a_ok = a0 + lc < a.d, b_ok = b0 + lc < a.d;
  const int ca = a_ok ? a0 + lc : 0, cb = b_ok ? b0 + lc : 0;
  const bool diag = ta == tb;            // the diagonal block of a column panel also owns its X^T (w*y) slice
  double vy = 0.0;
  // raw values of the slab in flight; the weighting and the X^T (w*y) term are applied when the slab is parked in LDS,
  // so that the loads can be requested in slices spread over the previous slab's k-steps (requested in one go after the
  // barrier they held the wave in the issue stage before its first MFMA of the slab -- see K1, profiles/r02_notes.md)
  double va[LP], vb[LP], vw[LP], vyv[LP];
  constexpr int NSL = KR / 4;            // one slice per k-step
  static_assert(LP % NSL == 0 || LP < NSL, "slices");
  auto load_part = [&](long long r0, int part) {
    const long long left = r_end - r0;
    const long long rows_here = left < KR ? (left > 0 ? left : 0) : KR;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.z + (size_t)r0 * a.dz), 0, (int)(rows_here * a.dz * 8), 0x00020000);
#pragma unroll
    for (int q = part * LP / NSL; q < (part + 1) * LP / NSL; ++q) {
      const int lr = lr0 + q * RPP;
      vw[q] = 1.0;
      if (a.w) vw[q] = (r0 + lr < r_end) ? a.w[r0 + lr] : 0.0;
      va[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (lr * a.dz + ca) * 8, 0, 0));
      vb[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (lr * a.dz + cb) * 8, 0, 0));
      vyv[q] = 0.;
      if (diag) vyv[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (lr * a.dz + a.d) * 8, 0, 0));
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int q = 0; q < LP; ++q) {
      const int lr = lr0 + q * RPP;
      const double raq = a_ok ? vw[q] * va[q] : 0.0;     // A panel carries the weights: (w[:,None]*X)
      Al[lr * LDX + lc] = raq;
      Bl[lr * LDX + lc] = b_ok ? vb[q] : 0.0;
      if (diag) vy = fma(raq, vyv[q], vy);               // (w[:,None]*Y[:,None]*X).sum(axis=0), model_linreg.py:31
    }
  };

  if (r_begin < r_end) {
#pragma unroll
    for (int part = 0; part < NSL; ++part) load_part(r_begin, part);
  }
  for (long long r0 = r_begin; r0 < r_end; r0 += KR) {
    store_slab();
    __syncthreads();
    const bool more = r0 + KR < r_end;
#pragma unroll
    for (int kk = 0; kk < KR / 4; ++kk) {
      if (more) load_part(r0 + KR, kk);
      double fa[MT], fb[MT];
#pragma unroll
      for (int x = 0; x < MT; ++x) fa[x] = Al[(kk * 4 + g) * LDX + wa * (BT / 2) + x * 16 + j];
#pragma unroll
      for (int y = 0; y < MT; ++y) fb[y] = Bl[(kk * 4 + g) * LDX + wb * (BT / 2) + y * 16 + j];
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < MT; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[x], fb[y], acc[x][y], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);     // keeps each slice of loads with its k-step
    }
    __syncthreads();
  }
  if (diag) {                            // combine the RPP thread-rows of each column in fixed order
    Yl[tid] = vy;
    __syncthreads();
    if (tid < BT) {
      double t = Yl[tid];
#pragma unroll
      for (int q = 1; q < RPP; ++q) t += Yl[q * BT + tid];
      a.partial_y[(size_t)split * a.nt * BT + ta * BT + tid] = t;
    }
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
  double* out = a.partial + ((size_t)split * a.ntri + tri) * BT * BT;
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < MT; ++y)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = wa * (BT / 2) + x * 16 + g + 4 * reg;
        const int col = wb * (BT / 2) + y * 16 + j;
        out[(size_t)row * BT + col] = acc[x][y][reg];
      }
}

// sum the per-split partial tiles in split order and scatter them (and their mirror images)
// into the dense dz x dz matrix
__global__ __launch_bounds__(256) void k_gram_reduce(const double* __restrict__ partial,
                                                    const double* __restrict__ partial_y, long long splits, int ntri,
                                                    int nt, int bt, int dz, double* __restrict__ out,
                                                    double* __restrict__ out_y) {
  // grid = (ntri, bt*bt/256): one output element per thread, partials summed in split order
  const int tri = blockIdx.x;
  if (tri == 0 && blockIdx.y == 0)
    for (int c = threadIdx.x; c < dz; c += blockDim.x) {
      double acc = 0.0;
      for (long long sp = 0; sp < splits; ++sp) acc += partial_y[(size_t)sp * nt * bt + c];
      out_y[c] = acc;
    }
  int ta = 0, rem = tri;
  while (rem >= nt - ta) { rem -= nt - ta; ++ta; }
  const int tb = ta + rem;
  const int e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e >= bt * bt) return;
  const int r = ta * bt + e / bt, c = tb * bt + e % bt;
  if (r >= dz || c >= dz || (ta == tb && r > c)) return;
  double acc = 0.0;
  const double* p = partial + (size_t)tri * bt * bt + e;
  const size_t stride = (size_t)ntri * bt * bt;
  long long sp = 0;
  for (; sp + 8 <= splits; sp += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(sp + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; sp < splits; ++sp) acc += p[(size_t)sp * stride];
  out[(size_t)r * dz + c] = acc;
  out[(size_t)c * dz + r] = acc;
}

// grow-only device scratch, owned by the context: the sampler calls weighted_post on <= M coreset rows thousands of
// times (bcores.py:39 -> sampler -> weighted_post), a hipMalloc/hipFree pair per call would dominate
static int gram_buf(bc_ctx* ctx, int which, size_t doubles, double** out) {
  int rc = bc_scratch_grow(ctx, &ctx->gram[which], doubles);
  if (rc) return rc;
  *out = ctx->gram[which].p;
  return BC_OK;
}

template <int BT>
static int run_gram(bc_ctx* ctx, const bc_data* data, const double* w_dev, double* out_dev, double* outy_dev) {
  const int dz = data->dz, d = dz - 1;
  const int nt = (d + BT - 1) / BT;
  const int ntri = nt * (nt + 1) / 2;
  const int KR = 16;
  // enough blocks for ~8 per CU, at least 2 slabs per split
  long long want_splits = ((long long)ctx->n_cu * 8 + ntri - 1) / ntri;
  long long max_splits = (data->n_rows + 2 * KR - 1) / (2 * KR);
  long long splits = want_splits < max_splits ? want_splits : max_splits;
  if (splits < 1) splits = 1;
  long long rps = (data->n_rows + splits - 1) / splits;
  rps = ((rps + KR - 1) / KR) * KR;
  if (rps < KR) rps = KR;
  splits = (data->n_rows + rps - 1) / rps;
  if (splits < 1) splits = 1;
  double* partial = nullptr;
  double* partial_y = nullptr;
  int rcb = gram_buf(ctx, 0, (size_t)splits * ntri * BT * BT, &partial);
  if (!rcb) rcb = gram_buf(ctx, 1, (size_t)splits * nt * BT, &partial_y);
  if (rcb) return rcb;
  hipError_t e = hipSuccess;
  GramArgs a;
  a.z = data->z;
  a.w = w_dev;
  a.partial = partial;
  a.partial_y = partial_y;
  a.n_rows = data->n_rows;
  a.rows_per_split = rps;
  a.dz = dz;
  a.d = d;
  a.nt = nt;
  a.ntri = ntri;
  int rc = bc_timer_begin(ctx, 2);
  if (!rc) {
    hipLaunchKernelGGL(k_gram<BT>, dim3((unsigned)(splits * ntri)), dim3(256), 0, ctx->stream, a);
    e = hipGetLastError();
  }
  if (!rc && e == hipSuccess) rc = bc_timer_end(ctx, 2);
  if (!rc && e == hipSuccess) {
    hipLaunchKernelGGL(k_gram_reduce, dim3(ntri, (BT * BT + 255) / 256), dim3(256), 0, ctx->stream, partial, partial_y, splits, ntri, nt, BT, d,
                       out_dev, outy_dev);
    e = hipGetLastError();
  }
  if (e != hipSuccess) return bc_hip_fail(e, "bc_weighted_gram", __FILE__, __LINE__);
  return rc;
}

// Coreset-sized inputs straight from host memory (the samplers call weighted_post on the <= M coreset rows once per
// gradient, bcores.py:39 -> sampler -> model_linreg.py:25-34): rows and weights travel in ONE transfer from the pinned
// staging area, both results come back in ONE, one synchronisation in all -- the per-call latency is what counts here.
extern "C" int bc_weighted_gram_host(bc_ctx* ctx, const double* z_rowmajor, int64_t n_rows, int32_t dz, const double* w,
                                     double* out_xtwx, double* out_xtwy) {
  if (!ctx || !out_xtwx || !out_xtwy || n_rows < 0 || dz < 2 || (n_rows > 0 && !z_rowmajor)) {
    bc_set_error("bc_weighted_gram_host: bad argument (rows must be [x (D >= 1), y])");
    return BC_INVALID_ARGUMENT;
  }
  const int d = dz - 1;
  const size_t n_z = (size_t)n_rows * dz, n_in = n_z + (w ? (size_t)n_rows : 0), n_out = (size_t)d * d + (size_t)d;
  if (n_in > ctx->pinned_doubles || n_out > ctx->pinned_doubles) {
    bc_set_error("bc_weighted_gram_host: meant for coreset-sized inputs (%lld rows x %d); upload larger ones with bc_data_from_host",
                 (long long)n_rows, dz);
    return BC_INVALID_ARGUMENT;
  }
  if (n_rows == 0) {
    memset(out_xtwx, 0, (size_t)d * d * sizeof(double));
    memset(out_xtwy, 0, (size_t)d * sizeof(double));
    return BC_OK;
  }
  BC_HIP(hipSetDevice(ctx->device));
  double* in_dev = nullptr;
  double* out_dev = nullptr;
  int rc = gram_buf(ctx, 4, n_in, &in_dev);
  if (!rc) rc = gram_buf(ctx, 2, n_out, &out_dev);
  if (rc) return rc;
  BC_HIP(hipStreamSynchronize(ctx->stream));            // nothing enqueued may still use the staging area
  memcpy(ctx->pinned, z_rowmajor, n_z * sizeof(double));
  if (w) memcpy(ctx->pinned + n_z, w, (size_t)n_rows * sizeof(double));
  BC_HIP(hipMemcpyAsync(in_dev, ctx->pinned, n_in * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  bc_data view;
  view.ctx = ctx;
  view.n_rows = n_rows;
  view.dz = dz;
  view.z = in_dev;
  view.owned = false;
  const double* w_dev = w ? in_dev + n_z : nullptr;
  rc = d > 64 ? run_gram<128>(ctx, &view, w_dev, out_dev, out_dev + (size_t)d * d) : run_gram<64>(ctx, &view, w_dev, out_dev, out_dev + (size_t)d * d);
  if (rc) return rc;
  BC_HIP(hipMemcpyAsync(ctx->pinned, out_dev, n_out * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(out_xtwx, ctx->pinned, (size_t)d * d * sizeof(double));
  memcpy(out_xtwy, ctx->pinned + (size_t)d * d, (size_t)d * sizeof(double));
  return BC_OK;
}

extern "C" int bc_weighted_gram(bc_ctx* ctx, const bc_data* data, const double* w, double* out_xtwx, double* out_xtwy) {
  if (!ctx || !data || !out_xtwx || !out_xtwy) { bc_set_error("bc_weighted_gram: bad argument"); return BC_INVALID_ARGUMENT; }
  if (data->ctx != ctx) { bc_set_error("bc_weighted_gram: data belongs to another context"); return BC_INVALID_ARGUMENT; }
  const int dz = data->dz, d = dz - 1;
  if (d <= 0) { bc_set_error("bc_weighted_gram: rows must be [x (D >= 1), y]"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  if (data->n_rows == 0) {
    memset(out_xtwx, 0, (size_t)d * d * sizeof(double));
    memset(out_xtwy, 0, (size_t)d * sizeof(double));
    return BC_OK;
  }
  double* w_dev = nullptr;
  double* out_dev = nullptr;
  double* outy_dev = nullptr;
  int rc = gram_buf(ctx, 2, (size_t)d * d, &out_dev);
  if (!rc) rc = gram_buf(ctx, 3, (size_t)d, &outy_dev);
  if (!rc && w) rc = gram_buf(ctx, 4, (size_t)data->n_rows, &w_dev);
  if (rc) return rc;
  if (w) BC_HIP(hipMemcpyAsync(w_dev, w, (size_t)data->n_rows * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  rc = d > 64 ? run_gram<128>(ctx, data, w_dev, out_dev, outy_dev) : run_gram<64>(ctx, data, w_dev, out_dev, outy_dev);
  if (rc) return rc;
  BC_HIP(hipMemcpyAsync(out_xtwx, out_dev, (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipMemcpyAsync(out_xtwy, outy_dev, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  return BC_OK;
}
