// x-gradients of the log-likelihood at the coreset's pseudo-points: the (M, S, W) tensor that
// BatchPSVICoreset moves its points with (bayesiancoresets/coreset/bpsvi.py:39-57), produced the way
// BlackBoxProjector.project(pts, grad=True) does (coreset/projector.py:27-32):
//     glls  = grad_loglikelihood(pts, samples)              (M, S, W)
//     glls -= glls.mean(axis=2)[:, :, np.newaxis]           centred over the LAST axis (the coordinates)
// for the three formulas the reference ships:
//     linear regression   model_linreg.py:12-17 (== model_neurlinr.py:99-100)   W = D + 1
//         1/sigsq * (y - x.th_s) * [th_s, 1]
//     logistic regression model_lr.py:107-114                                    W = D
//         m = -z.th_s ;  (m < 100 ? e^m / (1 + e^m) : 1) * th_s
//     Gaussian location   gaussian.py:17-20                                      W = d
//         th_s.Siginv - x.Siginv
// M is the coreset size (tens to hundreds of rows): the tensor is small (M*S*W doubles) and the work is one dot
// product and one scaled copy per (m, s) -- no tiling; one block per point, one wave per sample.
#include "bc_internal.h"
#include <cmath>
#include <cstring>

struct GradArgs {
  const double* z;       // [m][dz]
  const double* theta;   // [s][d]
  const double* ts;      // gauss: Theta.Siginv [s][d]
  const double* xs;      // gauss: X.Siginv     [m][d]
  double* out;           // [m][s][w]
  int m, s, d, dz, w, model;
  double c0;             // linreg: sigsq
};

// out[r][c] = sum_k in[r][k] * mat[k][c]   (rows x d) . (d x d)
__global__ __launch_bounds__(256) void k_rows_times_mat(const double* __restrict__ in, int ld_in, const double* __restrict__ mat,
                                                       int rows, int d, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * d) return;
  const int r = (int)(i / d), c = (int)(i % d);
  double t = 0.;
  for (int k = 0; k < d; ++k) t = fma(in[(size_t)r * ld_in + k], mat[(size_t)k * d + c], t);
  out[i] = t;
}

__global__ __launch_bounds__(256) void k_grad_x(GradArgs a) {
  extern __shared__ double xrow[];          // this point's row
  const int m = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int k = threadIdx.x; k < a.dz; k += blockDim.x) xrow[k] = a.z[(size_t)m * a.dz + k];
  __syncthreads();
  for (int s = wave; s < a.s; s += nw) {
    const double* th = a.theta + (size_t)s * a.d;
    double* o = a.out + ((size_t)m * a.s + s) * a.w;
    double fac = 0.;
    if (a.model != BC_MODEL_GAUSS_LL) {
      double p = 0.;
      for (int k = lane; k < a.d; k += 64) p = fma(xrow[k], th[k], p);
      p = bc_wave_sum_all(p);
      if (a.model == BC_MODEL_LINREG_LL) {
        fac = 1. / a.c0 * (xrow[a.d] - p);
      } else {
        const double mm = -p;
        fac = (mm < 100.) ? exp(mm) / (1. + exp(mm)) : 1.;
      }
    }
    // the (uncentred) gradient row, its mean over the W coordinates, the centred row
    double part = 0.;
    for (int k = lane; k < a.w; k += 64) {
      double v;
      if (a.model == BC_MODEL_GAUSS_LL) v = a.ts[(size_t)s * a.d + k] - a.xs[(size_t)m * a.d + k];
      else v = fac * (k < a.d ? th[k] : 1.);
      part += v;
    }
    const double mean = bc_wave_sum_all(part) / (double)a.w;
    for (int k = lane; k < a.w; k += 64) {
      double v;
      if (a.model == BC_MODEL_GAUSS_LL) v = a.ts[(size_t)s * a.d + k] - a.xs[(size_t)m * a.d + k];
      else v = fac * (k < a.d ? th[k] : 1.);
      o[k] = v - mean;
    }
  }
}

extern "C" int bc_project_grad_x(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                                 const double* params, int32_t n_params, double* out) {
  if (!ctx || !data || !theta || !out || s <= 0 || (n_params > 0 && !params)) {
    bc_set_error("bc_project_grad_x: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  if (data->ctx != ctx) { bc_set_error("bc_project_grad_x: data belongs to another context"); return BC_INVALID_ARGUMENT; }
  const int dz = data->dz;
  int d = dz, w = dz;
  double c0 = 0.;
  const double* siginv = nullptr;
  switch (model) {
    case BC_MODEL_LINREG_LL:
      if (n_params != 1 || dz < 2) { bc_set_error("bc_project_grad_x: linear regression takes params = {sigsq} and rows [x, y]"); return BC_INVALID_ARGUMENT; }
      d = dz - 1;
      c0 = params[0];
      break;
    case BC_MODEL_LOGISTIC_LL:
      if (n_params != 0) { bc_set_error("bc_project_grad_x: logistic regression takes no parameters"); return BC_INVALID_ARGUMENT; }
      break;
    case BC_MODEL_GAUSS_LL:
      if (n_params != 1 + dz * dz) { bc_set_error("bc_project_grad_x: Gaussian model takes params = {logdetSig, Siginv[d*d]}"); return BC_INVALID_ARGUMENT; }
      siginv = params + 1;
      break;
    default:
      bc_set_error("bc_project_grad_x: model %d has no x-gradient in the reference", model);
      return BC_INVALID_ARGUMENT;
  }
  const int64_t m = data->n_rows;
  if (m == 0) return BC_OK;
  if (m > (1 << 20)) { bc_set_error("bc_project_grad_x: meant for coreset points (m = %lld rows)", (long long)m); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  // device scratch: theta [s][d], (gauss) Siginv [d][d], ts [s][d], xs [m][d], out [m][s][w]
  const size_t n_th = (size_t)s * d, n_sg = siginv ? (size_t)d * d : 0, n_ts = siginv ? n_th : 0, n_xs = siginv ? (size_t)m * d : 0;
  const size_t n_out = (size_t)m * s * w;
  const size_t need = n_th + n_sg + n_ts + n_xs + n_out;
  int rcs = bc_scratch_grow(ctx, &ctx->gradx, need);      // owned by the context, freed with it
  if (rcs) return rcs;
  double* d_th = ctx->gradx.p;
  double* d_sg = d_th + n_th;
  double* d_ts = d_sg + n_sg;
  double* d_xs = d_ts + n_ts;
  double* d_out = d_xs + n_xs;
  BC_HIP(hipMemcpyAsync(d_th, theta, n_th * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (siginv) {
    BC_HIP(hipMemcpyAsync(d_sg, siginv, n_sg * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_rows_times_mat, dim3((unsigned)((n_ts + 255) / 256)), dim3(256), 0, ctx->stream, d_th, d, d_sg, s, d, d_ts);
    hipLaunchKernelGGL(k_rows_times_mat, dim3((unsigned)((n_xs + 255) / 256)), dim3(256), 0, ctx->stream, data->z, dz, d_sg, (int)m, d, d_xs);
    BC_HIP(hipGetLastError());
  }
  GradArgs a;
  memset(&a, 0, sizeof(a));
  a.z = data->z; a.theta = d_th; a.ts = d_ts; a.xs = d_xs; a.out = d_out;
  a.m = (int)m; a.s = s; a.d = d; a.dz = dz; a.w = w; a.model = model; a.c0 = c0;
  const size_t lds = (size_t)dz * sizeof(double);
  if (lds > 64 * 1024) { bc_set_error("bc_project_grad_x: rows of %d doubles do not fit the staging buffer", dz); return BC_INVALID_ARGUMENT; }
  hipLaunchKernelGGL(k_grad_x, dim3((unsigned)m), dim3(256), lds, ctx->stream, a);
  BC_HIP(hipGetLastError());
  BC_HIP(hipMemcpyAsync(out, d_out, n_out * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  return BC_OK;
}
