// Native candidate exchange: RCCL called directly from the library, on the library's stream.
//
// The per-step collective of the sharded greedy loop is ONE all-gather of an (S + 4)-double record per
// rank.  Going through a Python collective per step costs a host round trip plus the
// framework's stream/event bookkeeping (~10 us of dead stream time per step, measured) next to a ~50 us
// sweep; here the whole multi-rank loop  sweep -> rescoring -> ncclAllGather -> finish  is enqueued by
// bc_snnls_build without leaving C.  The communicator is bootstrapped by the host language: rank 0 calls
// bc_comm_unique_id, ships the 128 bytes to the other ranks by whatever channel it has (the Python layer
// uses torch.distributed), every rank calls bc_comm_create.
//
// RCCL is resolved with dlopen at first use -- the library must load (and export its symbols) on machines
// without RCCL, and a process that already carries a RCCL (PyTorch bundles one) should use that one:
// bc_comm_load(path) selects it, otherwise "librccl.so" is searched in the default paths.
#include "bc_internal.h"
#include <dlfcn.h>
#include <cstring>
#include <vector>

namespace {
// the subset of rccl.h this file needs (ABI-stable since NCCL 2.x)
typedef void* nccl_comm_t;
struct nccl_unique_id { char internal[128]; };
enum { NCCL_SUCCESS = 0, NCCL_FLOAT64 = 8 };
typedef int (*fn_get_unique_id)(nccl_unique_id*);
typedef int (*fn_comm_init_rank)(nccl_comm_t*, int, nccl_unique_id, int);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);
typedef int (*fn_comm_abort)(nccl_comm_t);

struct RcclApi {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_error_string error_string = nullptr;
  fn_comm_abort comm_abort = nullptr;      // optional
};
RcclApi g_rccl;

int rccl_load(const char* path) {
  if (g_rccl.lib) return BC_OK;
  const char* candidates[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (const char* c : candidates) {
    if (!c || !*c) continue;
    lib = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) { bc_set_error("RCCL not found: %s", dlerror()); return BC_INVALID_ARGUMENT; }
  RcclApi a;
  a.lib = lib;
  a.get_unique_id = (fn_get_unique_id)dlsym(lib, "ncclGetUniqueId");
  a.comm_init_rank = (fn_comm_init_rank)dlsym(lib, "ncclCommInitRank");
  a.comm_destroy = (fn_comm_destroy)dlsym(lib, "ncclCommDestroy");
  a.all_gather = (fn_all_gather)dlsym(lib, "ncclAllGather");
  a.error_string = (fn_error_string)dlsym(lib, "ncclGetErrorString");
  a.comm_abort = (fn_comm_abort)dlsym(lib, "ncclCommAbort");
  if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_gather) {
    bc_set_error("RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather");
    dlclose(lib);
    return BC_INVALID_ARGUMENT;
  }
  g_rccl = a;
  return BC_OK;
}

int rccl_fail(int res, const char* what) {
  bc_set_error("%s: RCCL error %d (%s)", what, res, g_rccl.error_string ? g_rccl.error_string(res) : "?");
  return -(1000 + res);
}
}  // namespace

struct bc_comm {
  bc_ctx* ctx = nullptr;
  nccl_comm_t nccl = nullptr;
  int rank = 0, world = 1;
  double* sum_buf = nullptr;       // scratch of bc_comm_sum_doubles: [count] send | [world][count] gathered | [count] result
  size_t sum_cap = 0;              // count the scratch holds
};

extern "C" int bc_comm_load(const char* rccl_library_path) { return rccl_load(rccl_library_path); }

extern "C" int bc_comm_unique_id(void* id_out, int32_t capacity) {
  if (!id_out || capacity < 128) { bc_set_error("bc_comm_unique_id: need a 128-byte buffer"); return BC_INVALID_ARGUMENT; }
  int rc = rccl_load(nullptr);
  if (rc) return rc;
  nccl_unique_id id;
  const int res = g_rccl.get_unique_id(&id);
  if (res != NCCL_SUCCESS) return rccl_fail(res, "ncclGetUniqueId");
  memcpy(id_out, id.internal, 128);
  return BC_OK;
}

extern "C" int bc_comm_create(bc_ctx* ctx, const void* id, int32_t rank, int32_t world, bc_comm** out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) {
    bc_set_error("bc_comm_create: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  int rc = rccl_load(nullptr);
  if (rc) return rc;
  BC_HIP(hipSetDevice(ctx->device));
  nccl_unique_id uid;
  memcpy(uid.internal, id, 128);
  bc_comm* c = new bc_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  const int res = g_rccl.comm_init_rank(&c->nccl, world, uid, rank);      // collective: returns once every rank joined
  if (res != NCCL_SUCCESS) { delete c; return rccl_fail(res, "ncclCommInitRank"); }
  *out = c;
  return BC_OK;
}

extern "C" int bc_comm_destroy(bc_comm* c) {
  if (!c) return BC_OK;
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->nccl) (void)g_rccl.comm_destroy(c->nccl);
  if (c->sum_buf) (void)hipFree(c->sum_buf);
  delete c;
  return BC_OK;
}

// Tear the communicator down WITHOUT waiting for outstanding collectives (ncclCommAbort): what a rank does
// when it fails in the middle of a multi-rank loop, so that it exits instead of leaving its peers' next
// all-gather waiting for it forever (they then fail out of RCCL instead of hanging).
extern "C" int bc_comm_abort(bc_comm* c) {
  if (!c) return BC_OK;
  if (c->nccl) {
    if (g_rccl.comm_abort) (void)g_rccl.comm_abort(c->nccl);
    else (void)g_rccl.comm_destroy(c->nccl);
    c->nccl = nullptr;
  }
  return BC_OK;
}

// What bc_comm_create needs LOCALLY (library present, device usable): run it on every rank and agree on the
// outcome BEFORE entering ncclCommInitRank -- a rank that fails here would otherwise leave the others inside the
// collective bootstrap.
extern "C" int bc_comm_precheck(bc_ctx* ctx) {
  if (!ctx) { bc_set_error("bc_comm_precheck: bad argument"); return BC_INVALID_ARGUMENT; }
  int rc = rccl_load(nullptr);
  if (rc) return rc;
  BC_HIP(hipSetDevice(ctx->device));
  void* p = nullptr;
  BC_HIP(hipMalloc(&p, 4096));
  BC_HIP(hipFree(p));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  return BC_OK;
}

bc_ctx* bc_comm_ctx(const bc_comm* c) { return c ? c->ctx : nullptr; }

extern "C" int bc_comm_info(const bc_comm* c, int32_t* rank, int32_t* world) {
  if (!c) return BC_INVALID_ARGUMENT;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return BC_OK;
}

// enqueue on the context's stream; no host synchronisation
int bc_comm_all_gather_dev(bc_comm* c, const double* send_dev, double* recv_dev, size_t count) {
  if (!c->nccl) { bc_set_error("RCCL all-gather: the communicator was aborted"); return -1; }
  const int res = g_rccl.all_gather(send_dev, recv_dev, count, NCCL_FLOAT64, c->nccl, c->ctx->stream);
  if (res != NCCL_SUCCESS) return rccl_fail(res, "ncclAllGather");
  return BC_OK;
}

extern "C" int bc_comm_all_gather(bc_comm* c, const void* send_dev, void* recv_dev, int64_t count) {
  if (!c || !send_dev || !recv_dev || count < 0) { bc_set_error("bc_comm_all_gather: bad argument"); return BC_INVALID_ARGUMENT; }
  return bc_comm_all_gather_dev(c, (const double*)send_dev, (double*)recv_dev, (size_t)count);
}

// Sum of `count` doubles over ranks IN RANK ORDER (the same bits on every rank and on every run, unlike a ring
// all-reduce): all-gather on the context's stream, then one small kernel adds rank 0, 1, ... in turn.  Replaces the
// host -> torch.distributed -> host bounce of the replicated S-vector sums (b = Phi^T 1 of a sharded projection,
// bcores.py:77 per gradient step; hilbert.py:17 once).
__global__ void k_sum_rank_order(const double* __restrict__ gathered, int world, long long count, double* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
    double acc = gathered[i];
    for (int r = 1; r < world; ++r) acc = acc + gathered[(size_t)r * count + i];
    out[i] = acc;
  }
}

// device-to-device part: all-gather + rank-order sum, enqueued on the context's stream; *result_dev stays valid until
// the next sum through this communicator
int bc_comm_sum_dev(bc_comm* c, const double* in_dev, int64_t count, const double** result_dev) {
  if (!c || !in_dev || !result_dev || count <= 0) { bc_set_error("bc_comm_sum_dev: bad argument"); return BC_INVALID_ARGUMENT; }
  if (!c->nccl) { bc_set_error("bc_comm_sum_doubles: the communicator was aborted"); return -1; }
  bc_ctx* ctx = c->ctx;
  BC_HIP(hipSetDevice(ctx->device));
  if ((size_t)count > c->sum_cap) {
    BC_HIP(hipStreamSynchronize(ctx->stream));
    if (c->sum_buf) (void)hipFree(c->sum_buf);
    c->sum_buf = nullptr;
    c->sum_cap = 0;
    BC_HIP(hipMalloc((void**)&c->sum_buf, (size_t)count * (c->world + 1) * sizeof(double)));
    c->sum_cap = (size_t)count;
  }
  double* gathered = c->sum_buf;
  double* result = c->sum_buf + c->sum_cap * c->world;
  int rc = bc_comm_all_gather_dev(c, in_dev, gathered, (size_t)count);
  if (rc) return rc;
  const int blocks = (int)((count + 255) / 256 < 64 ? (count + 255) / 256 : 64);
  hipLaunchKernelGGL(k_sum_rank_order, dim3(blocks), dim3(256), 0, ctx->stream, gathered, c->world, (long long)count, result);
  BC_HIP(hipGetLastError());
  *result_dev = result;
  return BC_OK;
}

extern "C" int bc_comm_sum_doubles(bc_comm* c, const double* in_dev, int64_t count, double* out_host) {
  if (!c || !in_dev || !out_host || count <= 0) { bc_set_error("bc_comm_sum_doubles: bad argument"); return BC_INVALID_ARGUMENT; }
  bc_ctx* ctx = c->ctx;
  if ((size_t)count > ctx->pinned_doubles) { bc_set_error("bc_comm_sum_doubles: at most %zu doubles", ctx->pinned_doubles); return BC_INVALID_ARGUMENT; }
  const double* result = nullptr;
  int rc = bc_comm_sum_dev(c, in_dev, count, &result);
  if (rc) return rc;
  BC_HIP(hipMemcpyAsync(ctx->pinned, result, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(out_host, ctx->pinned, (size_t)count * sizeof(double));
  return BC_OK;
}

// Test hook: the rank-order sum on a FABRICATED gathered buffer ([world][count], host), so that the [world][count]
// indexing and the order of additions are checked for any world size on a single GPU, without a communicator.
extern "C" int bc_comm_rank_order_sum_selftest(bc_ctx* ctx, const double* gathered_host, int32_t world, int64_t count, double* out_host) {
  if (!ctx || !gathered_host || !out_host || world < 1 || count <= 0) { bc_set_error("bc_comm_rank_order_sum_selftest: bad argument"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  double* buf = nullptr;
  BC_HIP(hipMalloc((void**)&buf, (size_t)count * (world + 1) * sizeof(double)));
  hipError_t e = hipMemcpyAsync(buf, gathered_host, (size_t)count * world * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    const int blocks = (int)((count + 255) / 256 < 64 ? (count + 255) / 256 : 64);
    hipLaunchKernelGGL(k_sum_rank_order, dim3(blocks), dim3(256), 0, ctx->stream, buf, world, (long long)count, buf + (size_t)count * world);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out_host, buf + (size_t)count * world, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(buf);
  if (e != hipSuccess) return bc_hip_fail(e, "bc_comm_rank_order_sum_selftest", __FILE__, __LINE__);
  return BC_OK;
}

// Wiring check, run once after creation: every rank contributes a rank-coded pattern and verifies the
// gathered buffer element by element (order, count, dtype, stream).  Collective.
extern "C" int bc_comm_selftest(bc_comm* c) {
  if (!c) return BC_INVALID_ARGUMENT;
  const int n = 16;
  std::vector<double> send(n), recv((size_t)n * c->world, -1.0);
  for (int k = 0; k < n; ++k) send[k] = 1000.0 * c->rank + k + 0.5;
  double *ds = nullptr, *dr = nullptr;
  BC_HIP(hipSetDevice(c->ctx->device));
  BC_HIP(hipMalloc((void**)&ds, n * sizeof(double)));
  hipError_t e = hipMalloc((void**)&dr, (size_t)n * c->world * sizeof(double));
  if (e != hipSuccess) { (void)hipFree(ds); return bc_hip_fail(e, "hipMalloc(selftest)", __FILE__, __LINE__); }
  int rc = BC_OK;
  e = hipMemcpyAsync(ds, send.data(), n * sizeof(double), hipMemcpyHostToDevice, c->ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(dr, 0xff, (size_t)n * c->world * sizeof(double), c->ctx->stream);
  if (e == hipSuccess) rc = bc_comm_all_gather_dev(c, ds, dr, n);
  if (e == hipSuccess && !rc) e = hipMemcpyAsync(recv.data(), dr, recv.size() * sizeof(double), hipMemcpyDeviceToHost, c->ctx->stream);
  if (e == hipSuccess && !rc) e = hipStreamSynchronize(c->ctx->stream);
  (void)hipFree(ds);
  (void)hipFree(dr);
  if (e != hipSuccess) return bc_hip_fail(e, "bc_comm_selftest", __FILE__, __LINE__);
  if (rc) return rc;
  for (int r = 0; r < c->world; ++r)
    for (int k = 0; k < n; ++k)
      if (recv[(size_t)r * n + k] != 1000.0 * r + k + 0.5) {
        bc_set_error("bc_comm_selftest: slot (rank %d, %d) holds %g, expected %g", r, k, recv[(size_t)r * n + k], 1000.0 * r + k + 0.5);
        return -1;
      }
  return BC_OK;
}
