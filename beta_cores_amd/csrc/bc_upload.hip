// bc_upload.hip -- host rows -> HBM, pipelined.
//
// The reference's API takes host ndarrays (hilbert.py:11, bcores.py:44), so for a drop-in user the first thing that
// happens to a 10 GB data set is its upload.  Rows travel in CHUNKS on a copy stream of the library's own, and a
// caller-supplied hook is called on the calling thread for every chunk IN ROW ORDER as soon as the chunk's copy is queued,
// with the event that marks its arrival: K1 is launched behind that event on the compute stream (bc_project_from_host,
// bc_project.hip) and projects chunk c while chunks c+1.. are still on the wire.
//
// Two ways of moving a chunk (measured on the MI355X boxes of this pool with tools/h2d_bench.hip and
// tools/upload_probe.py, profiles/r04_notes.md: the link delivers 56-57 GB/s from pinned memory):
//   direct (default, BC_UPLOAD_THREADS unset or 0): hipMemcpyAsync straight from the caller's pageable memory.  The
//     ROCm runtime of this image moves pageable memory at the link rate (50-56 GB/s for 4-10 GB arrays), so nothing is
//     gained by staging it ourselves -- the chunking alone buys the overlap with K1.
//   staged (BC_UPLOAD_THREADS = T >= 1): T host threads copy disjoint chunks into pinned staging buffers (two per thread,
//     8 MiB each) and queue the DMA of each sub-chunk on a copy stream of their own.  45-52 GB/s here (the memcpy into
//     staging competes with the DMA for the same memory controllers); kept for hosts whose pageable path is slow.
//
// Ownership: the host pointer is only read during the call (every byte has been handed to the runtime or copied into
// staging when the call returns); streams, staging buffers and events live in the context and are reused.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "bc_internal.h"

struct bc_uploader {
  int nthreads = 0;
  size_t sub_bytes = 0;
  std::vector<hipStream_t> streams;     // one copy stream per thread
  std::vector<void*> staging;           // two pinned buffers per thread
  std::vector<hipEvent_t> stg_ev;       // "the DMA out of this staging buffer is done"
  std::vector<char> stg_used;
  std::vector<hipEvent_t> landed;       // one per chunk of the current call (grown on demand, reused)
};

static int uploader_threads() {
  const char* env = getenv("BC_UPLOAD_THREADS");
  if (env) {
    const int t = atoi(env);
    return t < 0 ? 0 : (t > 32 ? 32 : t);
  }
  return 0;       // direct mode (see the head of this file)
}

// chunking of a plain upload (no per-chunk consumer): enough chunks to keep every copy thread busy, none below 4 MiB
int64_t bc_upload_default_chunk_rows(int64_t n_rows, int32_t dz) {
  const size_t row_bytes = (size_t)dz * sizeof(double);
  const size_t total = (size_t)n_rows * row_bytes;
  int t = uploader_threads();
  if (t < 1) t = 1;
  size_t cb = total / (size_t)(4 * t);
  if (cb < ((size_t)4 << 20)) cb = (size_t)4 << 20;
  if (cb > ((size_t)64 << 20)) cb = (size_t)64 << 20;
  int64_t rows = (int64_t)(cb / row_bytes);
  return rows < 1 ? 1 : rows;
}

static void uploader_destroy(bc_uploader* u);

void bc_uploader_free(bc_ctx* ctx) {
  bc_uploader* u = ctx->upl;
  if (!u) return;
  uploader_destroy(u);
  ctx->upl = nullptr;
}

// all the resources of an uploader, or none: a failure part-way frees what exists and leaves ctx->upl untouched, so that a
// later call never meets an uploader whose streams / staging buffers / events are fewer than its nthreads promises
static void uploader_destroy(bc_uploader* u) {
  for (auto s : u->streams) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
  for (auto p : u->staging) (void)hipHostFree(p);
  for (auto e : u->stg_ev) (void)hipEventDestroy(e);
  for (auto e : u->landed) (void)hipEventDestroy(e);
  delete u;
}

static int uploader_build(bc_uploader* u, int nthreads) {
  u->nthreads = nthreads;
  u->sub_bytes = (size_t)8 << 20;
  const char* env = getenv("BC_UPLOAD_SUB_MB");
  if (env && atoi(env) > 0) u->sub_bytes = (size_t)atoi(env) << 20;
  for (int t = 0; t < (nthreads > 0 ? nthreads : 1); ++t) {
    hipStream_t s;
    BC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    u->streams.push_back(s);
    for (int k = 0; k < (nthreads > 0 ? 2 : 0); ++k) {
      void* p = nullptr;
      BC_HIP(hipHostMalloc(&p, u->sub_bytes, hipHostMallocDefault));
      u->staging.push_back(p);
      hipEvent_t e;
      BC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
      u->stg_ev.push_back(e);
      u->stg_used.push_back(0);
    }
  }
  return BC_OK;
}

static int uploader_get(bc_ctx* ctx, int nthreads, size_t nchunks, bc_uploader** out) {
  bc_uploader* u = ctx->upl;
  if (u && u->nthreads != nthreads) {
    bc_uploader_free(ctx);
    u = nullptr;
  }
  if (!u) {
    u = new bc_uploader();
    const int rc = uploader_build(u, nthreads);
    if (rc != BC_OK) {
      uploader_destroy(u);
      return rc;
    }
    ctx->upl = u;
  }
  while (u->landed.size() < nchunks) {
    hipEvent_t e;
    BC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    u->landed.push_back(e);
  }
  *out = u;
  return BC_OK;
}

// rows [0, n_rows) of `src` (row-major, dz doubles per row) -> dst_dev, in chunks of chunk_rows rows.
// on_chunk(c, row0, rows, landed): called in chunk order on the calling thread; `landed` is recorded on a copy stream
// behind the chunk's last DMA.  Without a hook ctx->stream is made to wait for every chunk.
int bc_upload_rows(bc_ctx* ctx, const double* src, double* dst_dev, int64_t n_rows, int32_t dz, int64_t chunk_rows,
                   const bc_chunk_hook* on_chunk) {
  if (n_rows <= 0) return BC_OK;
  const size_t row_bytes = (size_t)dz * sizeof(double);
  const size_t total = (size_t)n_rows * row_bytes;
  int nthreads = uploader_threads();
  if (!on_chunk && (nthreads == 0 || total < ((size_t)32 << 20))) {
    // nobody consumes chunks: one copy on the compute stream (the runtime moves pageable memory at the link rate)
    BC_HIP(hipMemcpyAsync(dst_dev, src, total, hipMemcpyHostToDevice, ctx->stream));
    BC_HIP(hipStreamSynchronize(ctx->stream));     // the host buffer is only borrowed for the call
    return BC_OK;
  }
  if (chunk_rows <= 0) chunk_rows = n_rows;
  const int64_t nchunks = (n_rows + chunk_rows - 1) / chunk_rows;
  if (nthreads == 0) {
    // direct mode: chunk c goes out on the copy stream, the hook queues its consumer behind `landed`, and while that runs
    // on the compute stream this thread is already handing chunk c+1 to the runtime
    bc_uploader* u = nullptr;
    int rc = uploader_get(ctx, 0, (size_t)nchunks, &u);
    if (rc) return rc;
    hipStream_t st = u->streams[0];
    for (int64_t c = 0; c < nchunks; ++c) {
      const int64_t r0 = c * chunk_rows;
      const int64_t rows = (n_rows - r0) < chunk_rows ? (n_rows - r0) : chunk_rows;
      hipError_t e = hipMemcpyAsync(reinterpret_cast<char*>(dst_dev) + (size_t)r0 * row_bytes,
                                    reinterpret_cast<const char*>(src) + (size_t)r0 * row_bytes, (size_t)rows * row_bytes,
                                    hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipEventRecord(u->landed[(size_t)c], st);
      if (e != hipSuccess) { (void)hipStreamSynchronize(st); return bc_hip_fail(e, "chunked upload", __FILE__, __LINE__); }
      rc = (*on_chunk)(c, r0, rows, u->landed[(size_t)c]);
      if (rc) { (void)hipStreamSynchronize(st); return rc; }
    }
    // a pageable source may still be read by the runtime until the copies have drained: the buffer is only borrowed
    BC_HIP(hipStreamSynchronize(st));
    return BC_OK;
  }
  if (nthreads > nchunks) nthreads = (int)nchunks;
  bc_uploader* u = nullptr;
  {
    // (the uploader is sized for the configured thread count; fewer chunks than threads just leave some idle)
    int rc = uploader_get(ctx, uploader_threads(), (size_t)nchunks, &u);
    if (rc) return rc;
  }
  std::mutex mu;
  std::condition_variable cv;
  std::vector<char> ready((size_t)nchunks, 0);
  std::atomic<int> err{0};
  hipError_t first_err = hipSuccess;
  const int device = ctx->device;

  auto worker = [&](int t) {
    hipError_t e = hipSetDevice(device);
    hipStream_t st = u->streams[t];
    int buf = 0;
    for (int64_t c = t; c < nchunks; c += nthreads) {
      if (e == hipSuccess && err.load() == 0) {
        const int64_t r0 = c * chunk_rows;
        const int64_t rows = (n_rows - r0) < chunk_rows ? (n_rows - r0) : chunk_rows;
        const char* sp = reinterpret_cast<const char*>(src) + (size_t)r0 * row_bytes;
        char* dp = reinterpret_cast<char*>(dst_dev) + (size_t)r0 * row_bytes;
        size_t left = (size_t)rows * row_bytes;
        while (left > 0 && e == hipSuccess) {
          const size_t nb = left < u->sub_bytes ? left : u->sub_bytes;
          const int k = 2 * t + buf;
          if (u->stg_used[k]) e = hipEventSynchronize(u->stg_ev[k]);     // the previous DMA out of this buffer
          if (e != hipSuccess) break;
          memcpy(u->staging[k], sp, nb);
          e = hipMemcpyAsync(dp, u->staging[k], nb, hipMemcpyHostToDevice, st);
          if (e == hipSuccess) e = hipEventRecord(u->stg_ev[k], st);
          u->stg_used[k] = 1;
          sp += nb; dp += nb; left -= nb;
          buf ^= 1;
        }
        if (e == hipSuccess) e = hipEventRecord(u->landed[(size_t)c], st);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (e != hipSuccess && err.load() == 0) { err.store(1); first_err = e; }
        ready[(size_t)c] = 1;
      }
      cv.notify_all();
    }
  };
  std::vector<std::thread> pool;
  pool.reserve((size_t)nthreads);
  for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker, t);
  int rc = BC_OK;
  for (int64_t c = 0; c < nchunks; ++c) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return ready[(size_t)c] != 0; });
    }
    if (err.load() != 0 || rc) continue;          // keep draining `ready` so that every worker can finish
    const int64_t r0 = c * chunk_rows;
    const int64_t rows = (n_rows - r0) < chunk_rows ? (n_rows - r0) : chunk_rows;
    if (on_chunk) {
      rc = (*on_chunk)(c, r0, rows, u->landed[(size_t)c]);
      if (rc) err.store(2);                        // the workers stop queueing new chunks
    } else {
      hipError_t e = hipStreamWaitEvent(ctx->stream, u->landed[(size_t)c], 0);
      if (e != hipSuccess) { rc = bc_hip_fail(e, "hipStreamWaitEvent(upload)", __FILE__, __LINE__); err.store(2); }
    }
  }
  for (auto& th : pool) th.join();
  if (first_err != hipSuccess) {
    for (int t = 0; t < u->nthreads; ++t) (void)hipStreamSynchronize(u->streams[t]);
    return bc_hip_fail(first_err, "pipelined upload", __FILE__, __LINE__);
  }
  if (rc) {
    // a hook failed: nothing may still be writing into dst_dev when the caller frees it
    for (int t = 0; t < u->nthreads; ++t) (void)hipStreamSynchronize(u->streams[t]);
  }
  return rc;
}
