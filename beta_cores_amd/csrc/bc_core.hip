// libbeta_cores: context, Phi storage (row tiles), layout kernels, K2 (column
// sums / norms) and K3 (fused score + argmax sweep).  gfx950 only.
#include "bc_internal.h"
#include "bc_layout.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <climits>

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void bc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int bc_hip_fail(hipError_t e, const char* what, const char* file, int line) {
  bc_set_error("HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file, line, what);
  return -(int)e - 1000;
}

extern "C" const char* bc_last_error(void) { return g_err; }
extern "C" int bc_version(void) { return 210; }

// ------------------------------------------------------------------ context
extern "C" int bc_ctx_create(int device, void* stream, bc_ctx** out) {
  if (!out) { bc_set_error("bc_ctx_create: out is NULL"); return BC_INVALID_ARGUMENT; }
  *out = nullptr;
  int ndev = 0;
  BC_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) {
    bc_set_error("bc_ctx_create: device %d out of range (%d visible)", device, ndev);
    return BC_INVALID_ARGUMENT;
  }
  BC_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  BC_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    bc_set_error("bc_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return BC_INVALID_ARGUMENT;
  }
  bc_ctx* c = new bc_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  c->max_lds = (int)prop.sharedMemPerBlock;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0) c->max_lds = v;
  }
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return bc_hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    c->own_stream = true;
  }
  c->pinned_doubles = 1 << 16;
  hipError_t e = hipHostMalloc((void**)&c->pinned, c->pinned_doubles * sizeof(double), hipHostMallocDefault);
  if (e != hipSuccess) { delete c; return bc_hip_fail(e, "hipHostMalloc", __FILE__, __LINE__); }
  *out = c;
  return BC_OK;
}

static void timer_free(bc_timer& t) {
  for (auto ev : t.start) (void)hipEventDestroy(ev);
  for (auto ev : t.stop) (void)hipEventDestroy(ev);
  t.start.clear();
  t.stop.clear();
  t.used = 0;
}

int bc_scratch_grow(bc_ctx* ctx, bc_scratch* s, size_t doubles) {
  if (doubles <= s->cap) return BC_OK;
  BC_HIP(hipStreamSynchronize(ctx->stream));       // an enqueued kernel may still read the old buffer
  if (s->p) (void)hipFree(s->p);
  s->p = nullptr;
  s->cap = 0;
  const size_t want = doubles + doubles / 2;
  BC_HIP(hipMalloc((void**)&s->p, want * sizeof(double)));
  s->cap = want;
  return BC_OK;
}

extern "C" int bc_ctx_destroy(bc_ctx* ctx) {
  if (!ctx) return BC_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& t : ctx->timers) timer_free(t);
  for (auto& ev : ctx->vi_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->colsum_phi) bc_phi_destroy(ctx->colsum_phi);
  if (ctx->core_phi) bc_phi_destroy(ctx->core_phi);
  bc_scratch* all[] = {&ctx->proj_theta, &ctx->proj_rowaux, &ctx->proj_rowaux2, &ctx->gradx, &ctx->vi_buf, &ctx->const_rows,
                       &ctx->gram[0], &ctx->gram[1], &ctx->gram[2], &ctx->gram[3], &ctx->gram[4]};
  for (bc_scratch* sc : all)
    if (sc->p) (void)hipFree(sc->p);
  if (ctx->proj_pinned) (void)hipHostFree(ctx->proj_pinned);
  if (ctx->vi_pinned) (void)hipHostFree(ctx->vi_pinned);
  if (ctx->vi_side) { (void)hipStreamSynchronize(ctx->vi_side); (void)hipStreamDestroy(ctx->vi_side); }
  if (ctx->vi_ev_staged) (void)hipEventDestroy(ctx->vi_ev_staged);
  if (ctx->vi_ev_core) (void)hipEventDestroy(ctx->vi_ev_core);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  bc_uploader_free(ctx);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return BC_OK;
}

extern "C" int bc_ctx_sync(bc_ctx* ctx) {
  if (!ctx) { bc_set_error("bc_ctx_sync: NULL context"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipStreamSynchronize(ctx->stream));
  return BC_OK;
}

extern "C" int bc_ctx_enable_timing(bc_ctx* ctx, int on) {
  if (!ctx) return BC_INVALID_ARGUMENT;
  ctx->timing = on < 0 ? 0 : on;
  return BC_OK;
}

extern "C" int bc_ctx_timing_classes(bc_ctx* ctx, uint32_t mask) {
  if (!ctx) return BC_INVALID_ARGUMENT;
  ctx->timing_mask = mask & 0x3f;
  return BC_OK;
}

int bc_timer_begin(bc_ctx* ctx, int which) {
  if (!ctx->timing || !((ctx->timing_mask >> which) & 1u)) return BC_OK;
  bc_timer& t = ctx->timers[which];
  t.armed = (t.seq++ % ctx->timing) == 0;
  if (!t.armed) return BC_OK;
  if (t.used == t.start.size()) {
    hipEvent_t a, b;
    BC_HIP(hipEventCreate(&a));
    BC_HIP(hipEventCreate(&b));
    t.start.push_back(a);
    t.stop.push_back(b);
  }
  BC_HIP(hipEventRecord(t.start[t.used], ctx->stream));
  return BC_OK;
}

int bc_timer_end(bc_ctx* ctx, int which) {
  bc_timer& t = ctx->timers[which];
  if (!ctx->timing || !t.armed) return BC_OK;
  t.armed = false;
  BC_HIP(hipEventRecord(t.stop[t.used], ctx->stream));
  t.used++;
  t.launches++;
  return BC_OK;
}

static int timer_collect(bc_ctx* ctx, bc_timer& t) {
  if (t.used == 0) return BC_OK;
  BC_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < t.used; ++i) {
    float ms = 0.f;
    BC_HIP(hipEventElapsedTime(&ms, t.start[i], t.stop[i]));
    t.acc_ms += ms;
  }
  t.used = 0;
  return BC_OK;
}

extern "C" int bc_ctx_kernel_time(bc_ctx* ctx, int which, double* total_ms, int64_t* launches) {
  if (!ctx || which < 0 || which > 5) { bc_set_error("bc_ctx_kernel_time: bad argument"); return BC_INVALID_ARGUMENT; }
  int rc = timer_collect(ctx, ctx->timers[which]);
  if (rc) return rc;
  if (total_ms) *total_ms = ctx->timers[which].acc_ms;
  if (launches) *launches = ctx->timers[which].launches;
  return BC_OK;
}

extern "C" int bc_ctx_kernel_time_reset(bc_ctx* ctx) {
  if (!ctx) return BC_INVALID_ARGUMENT;
  for (auto& t : ctx->timers) {
    int rc = timer_collect(ctx, t);
    if (rc) return rc;
    t.acc_ms = 0.0;
    t.launches = 0;
    t.seq = 0;
  }
  return BC_OK;
}

// ------------------------------------------------------------------ data rows
extern "C" int bc_data_from_host(bc_ctx* ctx, const double* z, int64_t n_rows, int32_t dz, bc_data** out) {
  if (!ctx || !out || n_rows < 0 || dz <= 0 || (n_rows > 0 && !z)) {
    bc_set_error("bc_data_from_host: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  bc_data* d = new bc_data();
  d->ctx = ctx;
  d->n_rows = n_rows;
  d->dz = dz;
  size_t bytes = (size_t)n_rows * dz * sizeof(double);
  if (bytes) {
    hipError_t e = hipMalloc((void**)&d->z, bytes);
    if (e != hipSuccess) { delete d; return bc_hip_fail(e, "hipMalloc(data)", __FILE__, __LINE__); }
    d->cap_rows = n_rows;
    // default: ONE hipMemcpyAsync from the caller's array on ctx->stream, waited for (bc_upload.hip, direct mode: the runtime
    // moves pageable memory at the link rate here); BC_UPLOAD_THREADS = T >= 1 opts into T host threads copying through
    // pinned staging on copy streams of their own, with ctx->stream waiting for every chunk.  Either way the host buffer is
    // only borrowed for the call (every byte has left it when bc_upload_rows returns)
    int rc = bc_upload_rows(ctx, z, d->z, n_rows, dz, bc_upload_default_chunk_rows(n_rows, dz), nullptr);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(d->z); delete d; return rc; }
  }
  *out = d;
  return BC_OK;
}

extern "C" int bc_data_create(bc_ctx* ctx, int64_t cap_rows, int32_t dz, bc_data** out) {
  if (!ctx || !out || cap_rows < 0 || dz <= 0) { bc_set_error("bc_data_create: bad argument"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  bc_data* d = new bc_data();
  d->ctx = ctx;
  d->dz = dz;
  d->n_rows = 0;
  d->cap_rows = cap_rows > 0 ? cap_rows : 1;
  hipError_t e = hipMalloc((void**)&d->z, (size_t)d->cap_rows * dz * sizeof(double));
  if (e != hipSuccess) { delete d; return bc_hip_fail(e, "hipMalloc(data)", __FILE__, __LINE__); }
  *out = d;
  return BC_OK;
}

extern "C" int bc_data_upload(bc_data* d, const double* z, int64_t n_rows) {
  if (!d || n_rows < 0 || (n_rows > 0 && !z) || !d->owned) { bc_set_error("bc_data_upload: bad argument"); return BC_INVALID_ARGUMENT; }
  bc_ctx* ctx = d->ctx;
  if (n_rows > d->cap_rows) {
    BC_HIP(hipStreamSynchronize(ctx->stream));
    if (d->z) (void)hipFree(d->z);
    d->z = nullptr;
    int64_t cap = d->cap_rows * 2 > n_rows ? d->cap_rows * 2 : n_rows;
    BC_HIP(hipMalloc((void**)&d->z, (size_t)cap * d->dz * sizeof(double)));
    d->cap_rows = cap;
  }
  d->n_rows = n_rows;
  if (n_rows > 0) {
    int rc = bc_upload_rows(ctx, z, d->z, n_rows, d->dz, bc_upload_default_chunk_rows(n_rows, d->dz), nullptr);
    if (rc) return rc;                             // (the host buffer is only borrowed for the call: it has been read when this returns)
  }
  return BC_OK;
}

extern "C" int bc_data_from_device(bc_ctx* ctx, const void* z_dev, int64_t n_rows, int32_t dz, bc_data** out) {
  if (!ctx || !out || n_rows < 0 || dz <= 0 || (n_rows > 0 && !z_dev)) {
    bc_set_error("bc_data_from_device: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  bc_data* d = new bc_data();
  d->ctx = ctx;
  d->n_rows = n_rows;
  d->dz = dz;
  d->z = (double*)z_dev;
  d->owned = false;
  *out = d;
  return BC_OK;
}

__global__ void k_gather_data_rows(const double* __restrict__ z, int dz, const long long* __restrict__ idx, long long m,
                                   double* __restrict__ out) {
  const long long j = blockIdx.x;
  if (j >= m) return;
  const long long r = idx[j];
  for (int k = threadIdx.x; k < dz; k += blockDim.x) out[(size_t)j * dz + k] = z[(size_t)r * dz + k];
}

extern "C" int bc_data_gather_rows(bc_data* d, const int64_t* local_idx, int64_t m, double* out) {
  if (!d || m < 0 || (m > 0 && (!local_idx || !out))) { bc_set_error("bc_data_gather_rows: bad argument"); return BC_INVALID_ARGUMENT; }
  if (m == 0) return BC_OK;
  for (int64_t j = 0; j < m; ++j)
    if (local_idx[j] < 0 || local_idx[j] >= d->n_rows) {
      bc_set_error("bc_data_gather_rows: index %lld out of range [0,%lld)", (long long)local_idx[j], (long long)d->n_rows);
      return BC_INVALID_ARGUMENT;
    }
  bc_ctx* ctx = d->ctx;
  long long* didx = nullptr;
  double* dout = nullptr;
  BC_HIP(hipMalloc((void**)&didx, (size_t)m * sizeof(long long)));
  hipError_t e = hipMalloc((void**)&dout, (size_t)m * d->dz * sizeof(double));
  if (e == hipSuccess) e = hipMemcpyAsync(didx, local_idx, (size_t)m * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_gather_data_rows, dim3((unsigned)m), dim3(128), 0, ctx->stream, d->z, d->dz, didx, (long long)m, dout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)m * d->dz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(didx);
  if (dout) (void)hipFree(dout);
  if (e != hipSuccess) return bc_hip_fail(e, "bc_data_gather_rows", __FILE__, __LINE__);
  return BC_OK;
}

extern "C" int bc_data_destroy(bc_data* d) {
  if (!d) return BC_OK;
  if (d->owned && d->z) (void)hipFree(d->z);
  delete d;
  return BC_OK;
}

// ------------------------------------------------------------------ Phi storage
int bc_sweep_grid(const bc_phi* phi) {
  // Memory-bound sweep, one wave per tile and a grid-stride loop.  FEWER resident waves stream better: at N = 10M, S = 100
  // one 4-wave block per CU (each wave keeps 10 KiB in flight) reaches 0.886 of the HBM spec, two 0.873, four 0.857, eight
  // 0.846 (profiles/r02_notes.md); a wave of a narrow Phi has fewer bytes per tile to keep in flight and needs company
  // (S = 16: four blocks per CU are best, one loses 40 %).
  long long per_cu = phi->s >= 64 ? 1 : (phi->s >= 24 ? 2 : 4);
  const char* env = getenv("BC_SWEEP_BLOCKS_PER_CU");
  if (env && atoi(env) > 0) per_cu = atoi(env);
  long long want = (phi->ntiles + 3) / 4;
  long long cap = (long long)phi->ctx->n_cu * per_cu;
  long long g = want < cap ? want : cap;
  return (int)(g < 1 ? 1 : g);
}

static int stat_blocks_for(int64_t ntiles) {
  const int64_t nt = ntiles > 0 ? ntiles : 1;
  return (int)(nt < 256 ? nt : 256);
}

int bc_phi_set_rows(bc_phi* p, int64_t n_rows) {
  const int64_t nt = (n_rows + BC_TILE - 1) / BC_TILE;
  if (nt > p->cap_tiles) return 1;
  p->n_rows = n_rows;
  p->ntiles = nt;
  p->sweep_blocks = bc_sweep_grid(p);
  p->stat_blocks = stat_blocks_for(nt);
  p->part_rows = nt;
  p->stats_valid = false;
  return 0;
}

// ONE device allocation per Phi (tiles + all the small side arrays): creating / destroying a Phi is
// one hipMalloc / hipFree, which matters when small projections are made thousands of times.
int bc_phi_alloc(bc_ctx* ctx, int64_t n_rows, int32_t s, int64_t row_offset, bc_phi** out, int64_t cap_rows, bool stats_only) {
  bc_phi* p = new bc_phi();
  p->ctx = ctx;
  p->s = s;
  p->row_offset = row_offset;
  if (cap_rows < n_rows) cap_rows = n_rows;
  p->cap_tiles = (cap_rows + BC_TILE - 1) / BC_TILE;
  if (p->cap_tiles < 1) p->cap_tiles = 1;
  // sizes at capacity
  p->n_rows = p->cap_tiles * BC_TILE;
  p->ntiles = p->cap_tiles;
  const int sweep_cap = bc_sweep_grid(p);
  const int stat_cap = stat_blocks_for(p->cap_tiles);
  const size_t nt = (size_t)p->cap_tiles;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_tiles = take(stats_only ? 0 : bc_lay_phi_doubles((long long)nt, s) * sizeof(double));
  const size_t o_norms = take(stats_only ? 0 : nt * BC_TILE * sizeof(double));
  const size_t o_colsum = take((size_t)s * sizeof(double));
  const size_t o_tpart = take(nt * s * sizeof(double));
  const size_t o_stats = take(4 * sizeof(double));
  const size_t o_part2 = take((size_t)stat_cap * s * sizeof(double));
  const size_t o_nstat = take((size_t)stat_cap * 2 * sizeof(double));
  const size_t o_bval = take((size_t)sweep_cap * sizeof(double));
  const size_t o_bidx = take((size_t)sweep_cap * sizeof(long long));
  const size_t o_vbuf = take((size_t)2 * s * sizeof(double));
  const size_t o_rec = take((size_t)(s + BC_REC_HDR) * sizeof(double));
  const size_t o_cnt = take(256);
  hipError_t e = hipMalloc(&p->slab, off);
  if (e != hipSuccess) {
    delete p;
    return bc_hip_fail(e, "hipMalloc(phi)", __FILE__, __LINE__);
  }
  char* base = (char*)p->slab;
  p->tiles = stats_only ? nullptr : (double*)(base + o_tiles);
  p->norms = stats_only ? nullptr : (double*)(base + o_norms);
  p->colsum = (double*)(base + o_colsum);
  p->tile_part = (double*)(base + o_tpart);
  p->stats = (double*)(base + o_stats);
  p->part2 = (double*)(base + o_part2);
  p->nstat = (double*)(base + o_nstat);
  p->blk_val = (double*)(base + o_bval);
  p->blk_idx = (long long*)(base + o_bidx);
  p->vbuf = (double*)(base + o_vbuf);
  p->rec = (double*)(base + o_rec);
  p->sweep_counter = (unsigned*)(base + o_cnt);
  e = hipMemsetAsync(p->sweep_counter, 0, 256, ctx->stream);
  if (e != hipSuccess) {
    (void)hipFree(p->slab);
    delete p;
    return bc_hip_fail(e, "hipMemset(phi)", __FILE__, __LINE__);
  }
  (void)bc_phi_set_rows(p, n_rows);
  *out = p;
  return BC_OK;
}

extern "C" int bc_phi_create(bc_ctx* ctx, int64_t cap_rows, int32_t s, bc_phi** out) {
  if (!ctx || !out || cap_rows < 0 || s <= 0) { bc_set_error("bc_phi_create: bad argument"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  bc_phi* p = nullptr;
  int rc = bc_phi_alloc(ctx, 0, s, 0, &p, cap_rows);
  if (rc) return rc;
  BC_HIP(hipMemsetAsync(p->norms, 0, BC_TILE * sizeof(double), ctx->stream));
  BC_HIP(hipMemsetAsync(p->colsum, 0, (size_t)s * sizeof(double), ctx->stream));
  *out = p;
  return BC_OK;
}

extern "C" int bc_phi_destroy(bc_phi* p) {
  if (!p) return BC_OK;
  if (p->slab) (void)hipFree(p->slab);
  if (p->stage) (void)hipFree(p->stage);
  delete p;
  return BC_OK;
}

// row-major (rows x s) -> tile layout, plus row norms and per-tile column partial sums.
// One block per tile; rows are staged through LDS so both sides stay coalesced.
__global__ __launch_bounds__(256) void k_layout_from_rowmajor(const double* __restrict__ src, long long n_rows, int s,
                                                             double* __restrict__ tiles, double* __restrict__ norms,
                                                             double* __restrict__ tile_part) {
  extern __shared__ double lds[];  // [chunk_k][129]
  const long long t = blockIdx.x;
  const long long r0 = t * BC_TILE;
  const int rows = (int)((n_rows - r0) < BC_TILE ? (n_rows - r0) : BC_TILE);
  const int KC = 32;  // samples per pass
  const int tid = threadIdx.x;
  double nrm = 0.0;   // threads 0..127 own one row each for the norm
  for (int k0 = 0; k0 < s; k0 += KC) {
    const int kc = (s - k0) < KC ? (s - k0) : KC;
    // load rows x kc block: thread -> (row = idx / kc, k = idx % kc), coalesced along k
    for (int idx = tid; idx < BC_TILE * kc; idx += blockDim.x) {
      int r = idx / kc, k = idx - r * kc;
      double v = (r < rows) ? src[(size_t)(r0 + r) * s + k0 + k] : 0.0;
      lds[k * (BC_TILE + 1) + r] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < BC_TILE * kc; idx += blockDim.x) {
      int k = idx >> 7, r = idx & (BC_TILE - 1);
      tiles[(size_t)t * s * BC_TILE + (size_t)(k0 + k) * BC_TILE + r] = lds[k * (BC_TILE + 1) + r];
    }
    if (tid < BC_TILE) {
      for (int k = 0; k < kc; ++k) {
        double v = lds[k * (BC_TILE + 1) + tid];
        nrm = fma(v, v, nrm);
      }
    } else if (tid - BC_TILE < kc) {
      // threads 128.. : one sample each, column partial over the tile's rows (fixed order)
      int k = tid - BC_TILE;
      double acc = 0.0;
      for (int r = 0; r < BC_TILE; ++r) acc += lds[k * (BC_TILE + 1) + r];
      tile_part[(size_t)t * s + k0 + k] = acc;
    }
    __syncthreads();
  }
  if (tid < BC_TILE) norms[r0 + tid] = sqrt(nrm);
}

// tile layout -> row-major
__global__ __launch_bounds__(256) void k_layout_to_rowmajor(const double* __restrict__ tiles, long long n_rows, int s,
                                                           double* __restrict__ dst) {
  extern __shared__ double lds[];
  const long long t = blockIdx.x;
  const long long r0 = t * BC_TILE;
  const int rows = (int)((n_rows - r0) < BC_TILE ? (n_rows - r0) : BC_TILE);
  const int KC = 32;
  const int tid = threadIdx.x;
  for (int k0 = 0; k0 < s; k0 += KC) {
    const int kc = (s - k0) < KC ? (s - k0) : KC;
    for (int idx = tid; idx < BC_TILE * kc; idx += blockDim.x) {
      int k = idx >> 7, r = idx & (BC_TILE - 1);
      lds[k * (BC_TILE + 1) + r] = tiles[(size_t)t * s * BC_TILE + (size_t)(k0 + k) * BC_TILE + r];
    }
    __syncthreads();
    for (int idx = tid; idx < BC_TILE * kc; idx += blockDim.x) {
      int r = idx / kc, k = idx - r * kc;
      if (r < rows) dst[(size_t)(r0 + r) * s + k0 + k] = lds[k * (BC_TILE + 1) + r];
    }
    __syncthreads();
  }
}

// K2 tail, stage 1: block b reduces a fixed contiguous range of tiles (per-tile column
// partials, row norms) -> part2[b][s], nstat[b][2].  Fixed assignment => deterministic.
__global__ __launch_bounds__(256) void k_stats_stage1(const double* __restrict__ tile_part, long long ntiles, int s,
                                                     const double* __restrict__ norms, long long n_rows,
                                                     long long chunk, double* __restrict__ part2,
                                                     double* __restrict__ nstat, long long part_rows, long long pchunk) {
  __shared__ double red[32];
  __shared__ double part[256];
  const int tid = threadIdx.x, g = tid >> 6, lane = tid & 63;
  // column partials: block b owns rows [b * pchunk, (b + 1) * pchunk) of tile_part (one row per tile, or per wave of
  // the Theta-resident K1); row norms: tiles [b * chunk, (b + 1) * chunk)
  long long t0 = (long long)blockIdx.x * pchunk;
  long long t1 = t0 + pchunk;
  if (t1 > part_rows) t1 = part_rows;
  for (int k0 = 0; k0 < s; k0 += 64) {
    const int k = k0 + lane;
    double acc = 0.0;
    if (k < s) {
      // the wave's tiles t0+g, t0+g+4, ... are added in that order; eight loads are in flight at a time (a chain of
      // dependent HBM round trips made this kernel 78 us for the 78 125 tiles of a 10M-row projection)
      long long t = t0 + g;
      for (; t + 28 < t1; t += 32) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = tile_part[(size_t)(t + 4 * u) * s + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; t < t1; t += 4) acc += tile_part[(size_t)t * s + k];
    }
    part[tid] = acc;
    __syncthreads();
    if (g == 0 && k < s) part2[(size_t)blockIdx.x * s + k] = ((part[lane] + part[64 + lane]) + part[128 + lane]) + part[192 + lane];
    __syncthreads();
  }
  if (!norms) return;                 // store-free projection: column sums only (block-uniform)
  t0 = (long long)blockIdx.x * chunk;
  t1 = t0 + chunk;
  if (t1 > ntiles) t1 = ntiles;
  double ns = 0.0, nz = 0.0;
  long long r1 = t1 * BC_TILE;
  if (r1 > n_rows) r1 = n_rows;
  for (long long r = t0 * BC_TILE + tid; r < r1; r += blockDim.x) {
    double v = norms[r];
    ns += v;
    nz += (v == 0.0) ? 1.0 : 0.0;
  }
  ns = bc_block_sum(ns, red);
  nz = bc_block_sum(nz, red);
  if (tid == 0) {
    nstat[2 * blockIdx.x] = ns;
    nstat[2 * blockIdx.x + 1] = nz;
  }
}

// stage 2: single block; the nb partials of a column are split over 8 thread groups (fixed
// contiguous ranges) and the group sums are combined in group order => deterministic.
__global__ __launch_bounds__(1024) void k_stats_stage2(const double* __restrict__ part2, const double* __restrict__ nstat,
                                                      int nb, int s, double* __restrict__ colsum,
                                                      double* __restrict__ stats) {
  __shared__ double part[1024];
  const int G = 8, per = (nb + G - 1) / G;
  const int g = threadIdx.x >> 7, lane = threadIdx.x & 127;
  for (int k0 = 0; k0 < s; k0 += 128) {
    const int k = k0 + lane;
    double acc = 0.0;
    if (k < s) {
      const int b0 = g * per, b1 = (b0 + per) < nb ? (b0 + per) : nb;
      int b = b0;
      for (; b + 8 <= b1; b += 8) {          // same order of additions, eight loads in flight (one at a time: ~10 us for nb = 256)
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part2[(size_t)(b + u) * s + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; b < b1; ++b) acc += part2[(size_t)b * s + k];
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && k < s) {
      double t = part[lane];
      for (int gg = 1; gg < G; ++gg) t += part[gg * 128 + lane];
      colsum[k] = t;
    }
    __syncthreads();
  }
  if (!nstat) return;                 // store-free projection: no norm statistics
  // norm statistics: same split over the first 2 x G x ... threads
  double ns = 0.0, nz = 0.0;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
    ns += nstat[2 * b];
    nz += nstat[2 * b + 1];
  }
  part[threadIdx.x] = ns;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    const int m = nb < 1024 ? nb : 1024;
    for (int i = 0; i < m; ++i) t += part[i];
    stats[0] = t;
  }
  __syncthreads();
  part[threadIdx.x] = nz;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    const int m = nb < 1024 ? nb : 1024;
    for (int i = 0; i < m; ++i) t += part[i];
    stats[1] = t;
  }
}

// tile_part -> colsum with the very kernels, grid and order of additions of bc_phi_finish_stats (the column sums of a
// store-free projection carry the bits the materialised one's would), minus the norm statistics and the host copy
int bc_phi_reduce_colsum(bc_phi* p) {
  bc_ctx* ctx = p->ctx;
  const int nb = p->stat_blocks;
  const long long chunk = (p->ntiles + nb - 1) / nb;
  const long long prow = p->part_rows > 0 ? p->part_rows : p->ntiles, pchunk = (prow + nb - 1) / nb;
  hipLaunchKernelGGL(k_stats_stage1, dim3(nb), dim3(256), 0, ctx->stream, p->tile_part, (long long)p->ntiles, p->s,
                     (const double*)nullptr, (long long)p->n_rows, chunk > 0 ? chunk : 1, p->part2, p->nstat, prow,
                     pchunk > 0 ? pchunk : 1);
  BC_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_stats_stage2, dim3(1), dim3(1024), 0, ctx->stream, p->part2, (const double*)nullptr, nb, p->s, p->colsum,
                     p->stats);
  BC_HIP(hipGetLastError());
  return BC_OK;
}

int bc_phi_finish_stats(bc_phi* p) {
  bc_ctx* ctx = p->ctx;
  const int nb = p->stat_blocks;
  const long long chunk = (p->ntiles + nb - 1) / nb;
  const long long prow = p->part_rows > 0 ? p->part_rows : p->ntiles, pchunk = (prow + nb - 1) / nb;
  hipLaunchKernelGGL(k_stats_stage1, dim3(nb), dim3(256), 0, ctx->stream, p->tile_part, (long long)p->ntiles, p->s,
                     p->norms, (long long)p->n_rows, chunk > 0 ? chunk : 1, p->part2, p->nstat, prow,
                     pchunk > 0 ? pchunk : 1);
  BC_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_stats_stage2, dim3(1), dim3(1024), 0, ctx->stream, p->part2, p->nstat, nb, p->s, p->colsum,
                     p->stats);
  BC_HIP(hipGetLastError());
  BC_HIP(hipMemcpyAsync(ctx->pinned, p->stats, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  p->norm_sum = ctx->pinned[0];
  p->zero_rows = (int64_t)ctx->pinned[1];
  p->stats_valid = true;
  return BC_OK;
}

extern "C" int bc_phi_from_host(bc_ctx* ctx, const double* src, int64_t n_rows, int32_t s, int64_t row_offset,
                                bc_phi** out) {
  if (!ctx || !out || n_rows < 0 || s <= 0 || (n_rows > 0 && !src)) {
    bc_set_error("bc_phi_from_host: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  BC_HIP(hipSetDevice(ctx->device));
  bc_phi* p = nullptr;
  int rc = bc_phi_alloc(ctx, n_rows, s, row_offset, &p);
  if (rc) return rc;
  if (n_rows > 0) {
    double* stage = nullptr;
    size_t bytes = (size_t)n_rows * s * sizeof(double);
    hipError_t e = hipMalloc((void**)&stage, bytes);
    if (e != hipSuccess) { bc_phi_destroy(p); return bc_hip_fail(e, "hipMalloc(stage)", __FILE__, __LINE__); }
    e = hipMemcpyAsync(stage, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      size_t lds = (size_t)32 * (BC_TILE + 1) * sizeof(double);
      hipLaunchKernelGGL(k_layout_from_rowmajor, dim3((unsigned)p->ntiles), dim3(256), lds, ctx->stream, stage,
                         (long long)n_rows, s, p->tiles, p->norms, p->tile_part);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) { bc_phi_destroy(p); return bc_hip_fail(e, "upload/layout", __FILE__, __LINE__); }
  } else {
    BC_HIP(hipMemsetAsync(p->norms, 0, BC_TILE * sizeof(double), ctx->stream));
  }
  rc = bc_phi_finish_stats(p);
  if (rc) { bc_phi_destroy(p); return rc; }
  *out = p;
  return BC_OK;
}

// Segmented row sums (grouped selection, bcores.py:46-50,56-61 / sparsevi.py:44-48,54-59):
//   dst[g, :] = sum_{j in [offsets[g], offsets[g+1])} Phi[members[j], :]
// accumulated sequentially in member order, which is what `vecs[idcs].sum(axis=0)` does in NumPy (a reduction over
// the outer axis adds row after row).  One block per group, thread = sample; the reads of a member row are 1 KiB
// apart per sample (tile layout) but groups are usually runs of neighbouring rows, which share cache lines.
__global__ __launch_bounds__(128) void k_group_sum(const double* __restrict__ tiles, int s, const long long* __restrict__ members,
                                                  const long long* __restrict__ offsets, double* __restrict__ dst) {
  const long long g = blockIdx.x;
  const long long j0 = offsets[g], j1 = offsets[g + 1];
  for (int k = threadIdx.x; k < s; k += blockDim.x) {
    double acc = 0.0;
    for (long long j = j0; j < j1; ++j) acc += tiles[bc_tile_off(members[j], k, s)];
    dst[(size_t)g * s + k] = acc;
  }
}

extern "C" int bc_phi_group_sum(bc_phi* p, const int64_t* members, const int64_t* offsets, int64_t n_groups, bc_phi** out) {
  if (!p || !out || n_groups < 0 || (n_groups > 0 && (!offsets || (offsets[n_groups] > 0 && !members)))) {
    bc_set_error("bc_phi_group_sum: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  const int64_t total = n_groups > 0 ? offsets[n_groups] : 0;
  for (int64_t g = 0; g < n_groups; ++g)
    if (offsets[g] < 0 || offsets[g + 1] < offsets[g]) { bc_set_error("bc_phi_group_sum: offsets must be non-decreasing from >= 0"); return BC_INVALID_ARGUMENT; }
  for (int64_t j = 0; j < total; ++j)
    if (members[j] < 0 || members[j] >= p->n_rows) {
      bc_set_error("bc_phi_group_sum: member %lld out of range [0,%lld)", (long long)members[j], (long long)p->n_rows);
      return BC_INVALID_ARGUMENT;
    }
  bc_ctx* ctx = p->ctx;
  BC_HIP(hipSetDevice(ctx->device));
  bc_phi* q = nullptr;
  int rc = bc_phi_alloc(ctx, n_groups, p->s, 0, &q);
  if (rc) return rc;
  if (n_groups > 0) {
    long long *dmem = nullptr, *doff = nullptr;
    double* stage = nullptr;
    hipError_t e = hipMalloc((void**)&doff, (size_t)(n_groups + 1) * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc((void**)&dmem, (size_t)(total > 0 ? total : 1) * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc((void**)&stage, (size_t)n_groups * p->s * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(doff, offsets, (size_t)(n_groups + 1) * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && total > 0) e = hipMemcpyAsync(dmem, members, (size_t)total * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_group_sum, dim3((unsigned)n_groups), dim3(128), 0, ctx->stream, p->tiles, p->s, dmem, doff, stage);
      const size_t lds = (size_t)32 * (BC_TILE + 1) * sizeof(double);
      hipLaunchKernelGGL(k_layout_from_rowmajor, dim3((unsigned)q->ntiles), dim3(256), lds, ctx->stream, stage,
                         (long long)n_groups, p->s, q->tiles, q->norms, q->tile_part);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (doff) (void)hipFree(doff);
    if (dmem) (void)hipFree(dmem);
    if (stage) (void)hipFree(stage);
    if (e != hipSuccess) { bc_phi_destroy(q); return bc_hip_fail(e, "bc_phi_group_sum", __FILE__, __LINE__); }
  } else {
    BC_HIP(hipMemsetAsync(q->norms, 0, BC_TILE * sizeof(double), ctx->stream));
  }
  rc = bc_phi_finish_stats(q);
  if (rc) { bc_phi_destroy(q); return rc; }
  *out = q;
  return BC_OK;
}

extern "C" int bc_phi_shape(const bc_phi* p, int64_t* n_rows, int32_t* s, int64_t* row_offset) {
  if (!p) { bc_set_error("bc_phi_shape: NULL phi"); return BC_INVALID_ARGUMENT; }
  if (n_rows) *n_rows = p->n_rows;
  if (s) *s = p->s;
  if (row_offset) *row_offset = p->row_offset;
  return BC_OK;
}

extern "C" int bc_phi_colsum(bc_phi* p, double* out) {
  if (!p || !out) { bc_set_error("bc_phi_colsum: bad argument"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipMemcpyAsync(out, p->colsum, (size_t)p->s * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
  BC_HIP(hipStreamSynchronize(p->ctx->stream));
  return BC_OK;
}

struct bc_comm;
extern "C" int bc_comm_sum_doubles(bc_comm* c, const double* in_dev, int64_t count, double* out_host);
extern "C" int bc_comm_info(const bc_comm* c, int32_t* rank, int32_t* world);
bc_ctx* bc_comm_ctx(const bc_comm* c);

// b = Phi^T 1 over ALL ranks' row shards (hilbert.py:17, bcores.py:77): this shard's column sums are already in HBM
// (K1 fuses them); they are all-gathered and added in rank order without leaving the device.
extern "C" int bc_phi_colsum_all(bc_phi* p, bc_comm* c, double* out) {
  if (!p || !c || !out) { bc_set_error("bc_phi_colsum_all: bad argument"); return BC_INVALID_ARGUMENT; }
  int32_t rk = 0, wd = 0;
  if (bc_comm_info(c, &rk, &wd) != BC_OK) { bc_set_error("bc_phi_colsum_all: bad communicator"); return BC_INVALID_ARGUMENT; }
  if (bc_comm_ctx(c) != p->ctx) { bc_set_error("bc_phi_colsum_all: Phi and communicator belong to different contexts"); return BC_INVALID_ARGUMENT; }
  return bc_comm_sum_doubles(c, p->colsum, p->s, out);
}

extern "C" int bc_phi_norms(bc_phi* p, double* out) {
  if (!p || (!out && p->n_rows)) { bc_set_error("bc_phi_norms: bad argument"); return BC_INVALID_ARGUMENT; }
  if (p->n_rows)
    BC_HIP(hipMemcpyAsync(out, p->norms, (size_t)p->n_rows * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
  BC_HIP(hipStreamSynchronize(p->ctx->stream));
  return BC_OK;
}

extern "C" int bc_phi_norm_stats(bc_phi* p, int64_t* zero_rows, double* norm_sum) {
  if (!p) { bc_set_error("bc_phi_norm_stats: NULL phi"); return BC_INVALID_ARGUMENT; }
  if (!p->stats_valid) {
    int rc = bc_phi_finish_stats(p);
    if (rc) return rc;
  }
  if (zero_rows) *zero_rows = p->zero_rows;
  if (norm_sum) *norm_sum = p->norm_sum;
  return BC_OK;
}

extern "C" int bc_phi_to_host(bc_phi* p, double* out) {
  if (!p || (!out && p->n_rows)) { bc_set_error("bc_phi_to_host: bad argument"); return BC_INVALID_ARGUMENT; }
  if (p->n_rows == 0) return BC_OK;
  bc_ctx* ctx = p->ctx;
  const size_t need = (size_t)p->n_rows * p->s;
  if (need > p->stage_cap) {
    if (p->stage) (void)hipFree(p->stage);
    p->stage = nullptr;
    p->stage_cap = 0;
    const size_t cap = (size_t)p->cap_tiles * BC_TILE * p->s;
    BC_HIP(hipMalloc((void**)&p->stage, (cap > need ? cap : need) * sizeof(double)));
    p->stage_cap = cap > need ? cap : need;
  }
  size_t lds = (size_t)32 * (BC_TILE + 1) * sizeof(double);
  hipLaunchKernelGGL(k_layout_to_rowmajor, dim3((unsigned)p->ntiles), dim3(256), lds, ctx->stream, p->tiles,
                     (long long)p->n_rows, p->s, p->stage);
  BC_HIP(hipGetLastError());
  BC_HIP(hipMemcpyAsync(out, p->stage, need * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  return BC_OK;
}

__global__ void k_gather_rows(const double* __restrict__ tiles, int s, const long long* __restrict__ idx, long long m,
                              double* __restrict__ out) {
  long long j = blockIdx.x;
  if (j >= m) return;
  long long r = idx[j];
  for (int k = threadIdx.x; k < s; k += blockDim.x) out[(size_t)j * s + k] = tiles[bc_tile_off(r, k, s)];
}

extern "C" int bc_phi_gather_rows(bc_phi* p, const int64_t* local_idx, int64_t m, double* out) {
  if (!p || m < 0 || (m > 0 && (!local_idx || !out))) { bc_set_error("bc_phi_gather_rows: bad argument"); return BC_INVALID_ARGUMENT; }
  if (m == 0) return BC_OK;
  for (int64_t j = 0; j < m; ++j)
    if (local_idx[j] < 0 || local_idx[j] >= p->n_rows) {
      bc_set_error("bc_phi_gather_rows: index %lld out of range [0,%lld)", (long long)local_idx[j], (long long)p->n_rows);
      return BC_INVALID_ARGUMENT;
    }
  bc_ctx* ctx = p->ctx;
  long long* didx = nullptr;
  double* dout = nullptr;
  BC_HIP(hipMalloc((void**)&didx, (size_t)m * sizeof(long long)));
  hipError_t e = hipMalloc((void**)&dout, (size_t)m * p->s * sizeof(double));
  if (e == hipSuccess) e = hipMemcpyAsync(didx, local_idx, (size_t)m * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)m), dim3(128), 0, ctx->stream, p->tiles, p->s, didx, (long long)m, dout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)m * p->s * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(didx);
  if (dout) (void)hipFree(dout);
  if (e != hipSuccess) return bc_hip_fail(e, "bc_phi_gather_rows", __FILE__, __LINE__);
  return BC_OK;
}
