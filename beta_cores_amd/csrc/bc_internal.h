// Internal definitions shared by the HIP translation units of libbeta_cores.
// gfx950 (MI355X, CDNA4) only: wave = 64 lanes, fp64 throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <functional>
#include <vector>
#include "../../include/beta_cores.h"

#define BC_WAVE 64
#define BC_TILE BC_TILE_ROWS          // rows per Phi tile
#define BC_REC_HDR 4                  // candidate record header doubles: score, gidx(bits), norm, valid

void bc_set_error(const char* fmt, ...);
int bc_hip_fail(hipError_t e, const char* what, const char* file, int line);

#define BC_HIP(call)                                                        \
  do {                                                                      \
    hipError_t _e = (call);                                                 \
    if (_e != hipSuccess) return bc_hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

struct bc_timer {
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
  double acc_ms = 0.0;     // folded-in time of already collected events
  int64_t launches = 0;    // timed launches
  int64_t seq = 0;         // all launches seen (timed or not)
  bool armed = false;      // the launch in progress is being timed
};

// grow-only device buffer owned by a context (freed in bc_ctx_destroy): scratch of calls that are made thousands of
// times on coreset-sized inputs, where a hipMalloc / hipFree pair per call would dominate
struct bc_scratch {
  double* p = nullptr;
  size_t cap = 0;                // doubles
};

// phases of bc_vi_gradient, timed with HIP events when the context's timing is on (bc_ctx_phase_times)
#define BC_VI_PHASES 5           // upload (Theta, coreset rows, w) | K1 of the coreset rows | K1 over the data rows (store-free) |
                                 // column-sum reduction (+ rank-order sum over ranks) | M x S algebra + download
struct bc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int timing = 0;                // 0 = off, n >= 1: every n-th launch of each kernel class is timed
  unsigned timing_mask = 0x7;    // which classes (bit = class): the step stages 3-5 only on request (bc_ctx_timing_classes)
  bc_timer timers[6];            // 0 K3 sweep | 1 K1 projection | 2 K4 Gram + reduce | 3 rescoring / local winner |
                                 // 4 candidate all-gather (RCCL) | 5 step finish
  int n_cu = 256;
  int max_lds = 64 * 1024;       // hipDeviceAttributeMaxSharedMemoryPerBlock (160 KiB on gfx950)
  double* pinned = nullptr;      // small pinned staging area (host)
  size_t pinned_doubles = 0;
  // K1 staging (bc_project.hip): zero-padded Theta, per-sample / per-row extras, Siginv; a pinned host mirror
  bc_scratch proj_theta, proj_rowaux, proj_rowaux2;      // proj_theta: Theta | saux | Siginv | extras, one transfer
  double* proj_pinned = nullptr;
  size_t proj_pinned_cap = 0;
  bc_scratch gradx;              // bc_project_grad_x
  bc_scratch gram[5];            // K4: partial, partial_y, out, out_y, w
  bc_phi* colsum_phi = nullptr;  // store-free K1: a Phi with the per-tile column partials but no tiles / norms
  bc_phi* core_phi = nullptr;    // bc_vi_gradient: the projection of the <= M coreset rows
  bc_scratch vi_buf;             // bc_vi_gradient: grad | resid
  hipEvent_t vi_ev[BC_VI_PHASES + 1] = {};
  double vi_phase_ms[BC_VI_PHASES] = {};
  int64_t vi_calls_timed = 0;
  int64_t vi_pending_m = 0;      // bc_vi_gradient_begin enqueued a gradient of this many rows (bc_vi_gradient_end fetches it)
  int32_t vi_pending_s = 0;
  bool vi_pending_timed = false;
  double* vi_pinned = nullptr;   // pinned landing area of the pending gradient
  hipStream_t vi_side = nullptr;  // bc_vi_gradient: the coreset rows' K1 runs here, beside the data rows' launch on `stream`
  hipEvent_t vi_ev_staged = nullptr, vi_ev_core = nullptr;
  struct bc_uploader* upl = nullptr;   // pipelined host -> HBM uploads (bc_upload.hip): copy streams, pinned staging, events
  // host-evaluated constants of constant rows (bc_ctx_set_constant_row_values): sorted keys, then values, on the device
  bc_scratch const_rows;
  int64_t n_const_rows = 0;
  int const_model = -1;
  double const_params[4] = {0., 0., 0., 0.};
  int const_n_params = 0;
};

// bc_upload.hip: rows of a host array -> dst_dev through pinned staging and several copy threads; the hook (optional) is
// called per chunk in row order with the event that marks the chunk's arrival (nullptr: it has already landed)
typedef std::function<int(int64_t chunk, int64_t row0, int64_t rows, hipEvent_t landed)> bc_chunk_hook;
int bc_upload_rows(bc_ctx* ctx, const double* src, double* dst_dev, int64_t n_rows, int32_t dz, int64_t chunk_rows,
                   const bc_chunk_hook* on_chunk);
int64_t bc_upload_default_chunk_rows(int64_t n_rows, int32_t dz);
void bc_uploader_free(bc_ctx* ctx);

int bc_scratch_grow(bc_ctx* ctx, bc_scratch* s, size_t doubles);   // contents are NOT kept when it grows

int bc_timer_begin(bc_ctx* ctx, int which);
int bc_timer_end(bc_ctx* ctx, int which);

struct bc_data {
  bc_ctx* ctx = nullptr;
  int64_t n_rows = 0;
  int32_t dz = 0;
  double* z = nullptr;           // row-major n_rows x dz
  bool owned = true;
  int64_t cap_rows = 0;          // allocation size in rows (owned buffers)
};

struct bc_phi {
  bc_ctx* ctx = nullptr;
  int64_t n_rows = 0;
  int32_t s = 0;
  int64_t row_offset = 0;
  int64_t ntiles = 0;
  int64_t cap_tiles = 0;         // allocation size in tiles (n_rows may change below it)
  void* slab = nullptr;          // the one device allocation all pointers below point into
  double* stage = nullptr;       // row-major staging for bc_phi_to_host (lazy)
  size_t stage_cap = 0;
  double* tiles = nullptr;       // [ntiles][s][128]
  double* norms = nullptr;       // [ntiles*128]
  double* colsum = nullptr;      // [s]   (valid when stats_valid)
  double* tile_part = nullptr;   // [ntiles][s] per-tile column partial sums (scratch)
  int64_t part_rows = 0;         // rows of tile_part the last producer filled: ntiles, or one per wave (k_project_r)
  double* stats = nullptr;       // device: {norm_sum, zero_rows}
  double* part2 = nullptr;       // [stat_blocks][s] second-level column partials
  double* nstat = nullptr;       // [stat_blocks][2]
  int stat_blocks = 1;
  bool stats_valid = false;
  double norm_sum = 0.0;
  int64_t zero_rows = 0;
  // scratch for standalone argmax sweeps
  double* blk_val = nullptr;
  long long* blk_idx = nullptr;
  int sweep_blocks = 0;
  double* vbuf = nullptr;        // [2*s]
  double* rec = nullptr;         // one candidate record
  unsigned* sweep_counter = nullptr;   // arrival counter of the sweep's last-block reduction (kept at 0 between launches)
};

int bc_phi_alloc(bc_ctx* ctx, int64_t n_rows, int32_t s, int64_t row_offset, bc_phi** out, int64_t cap_rows = 0,
                 bool stats_only = false);   // stats_only: no tiles, no norms (store-free K1)
int bc_phi_set_rows(bc_phi* phi, int64_t n_rows);   // 0 if n_rows fits the capacity (state updated), 1 otherwise
int bc_phi_finish_stats(bc_phi* phi);   // tile_part -> colsum, norm stats (device), then host copy
int bc_phi_reduce_colsum(bc_phi* phi);  // tile_part -> colsum on the device only: no norm statistics, no host copy, no sync
int bc_sweep_grid(const bc_phi* phi);

// ------------------------------------------------------------------ device helpers
#if defined(__HIPCC__)

__device__ __forceinline__ bool bc_better(double av, long long ai, double bv, long long bi) {
  // np.argmax semantics: NaN beats everything, first occurrence wins ties.
  const bool an = av != av, bn = bv != bv;
  if (an | bn) {
    if (an & bn) return ai < bi;
    return an;
  }
  if (av > bv) return true;
  if (av < bv) return false;
  return ai < bi;
}

__device__ __forceinline__ long long bc_shfl_down_ll(long long v, int d) {
  int lo = (int)(v & 0xffffffffLL), hi = (int)(v >> 32);
  lo = __shfl_down(lo, d, BC_WAVE);
  hi = __shfl_down(hi, d, BC_WAVE);
  return ((long long)hi << 32) | (unsigned int)lo;
}

__device__ __forceinline__ void bc_wave_argmax(double& v, long long& i) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    double ov = __shfl_down(v, d, BC_WAVE);
    long long oi = bc_shfl_down_ll(i, d);
    if (bc_better(ov, oi, v, i)) { v = ov; i = oi; }
  }
}

// Wave-wide sum of a double, the total in EVERY lane (all 64 lanes must be active).  Inside a 16-lane row the
// partners come through DPP row rotations (v_mov_b32_dpp: a few cycles each) -- a rotation butterfly, so all lanes of
// a row hold bit-identical sums -- and the four row sums are combined from SGPRs (v_readlane) in a fixed order.  The
// __shfl_down tree this replaces moved each step through ds_bpermute: ~1.5k cycles per reduction, and the
// single-block step kernels do a dozen of them back to back (8 of the 20 us of a greedy step's tail).
template <int CTRL>
__device__ __forceinline__ double bc_dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bc_readlane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bc_wave_sum_all(double v) {
  v += bc_dpp_mov<0x128>(v);   // row_ror:8
  v += bc_dpp_mov<0x124>(v);   // row_ror:4
  v += bc_dpp_mov<0x122>(v);   // row_ror:2
  v += bc_dpp_mov<0x121>(v);   // row_ror:1
  return (bc_readlane(v, 0) + bc_readlane(v, 16)) + (bc_readlane(v, 32) + bc_readlane(v, 48));
}

__device__ __forceinline__ double bc_wave_sum(double v) { return bc_wave_sum_all(v); }

// wave-wide fmax of a double, the maximum in every lane (all 64 lanes active): the same rotation butterfly
__device__ __forceinline__ double bc_wave_max_all(double v) {
  v = fmax(v, bc_dpp_mov<0x128>(v));   // row_ror:8
  v = fmax(v, bc_dpp_mov<0x124>(v));   // row_ror:4
  v = fmax(v, bc_dpp_mov<0x122>(v));   // row_ror:2
  v = fmax(v, bc_dpp_mov<0x121>(v));   // row_ror:1
  return fmax(fmax(bc_readlane(v, 0), bc_readlane(v, 16)), fmax(bc_readlane(v, 32), bc_readlane(v, 48)));
}

// Wave-wide fmaxf, the maximum in every lane (all 64 lanes active): the same rotation butterfly on one register.
template <int CTRL>
__device__ __forceinline__ float bc_dpp_mov_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ float bc_wave_max_f32_all(float v) {
  v = fmaxf(v, bc_dpp_mov_f32<0x128>(v));   // row_ror:8
  v = fmaxf(v, bc_dpp_mov_f32<0x124>(v));   // row_ror:4
  v = fmaxf(v, bc_dpp_mov_f32<0x122>(v));   // row_ror:2
  v = fmaxf(v, bc_dpp_mov_f32<0x121>(v));   // row_ror:1
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// block-wide sum, result broadcast to every thread; red must hold >= 17 doubles
__device__ __forceinline__ double bc_block_sum(double v, double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = bc_wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[w];
    red[16] = t;
  }
  __syncthreads();
  return red[16];
}

// block-wide sum of N values at once (one barrier pair for all of them); red must hold >= 17*N doubles
template <int N>
__device__ __forceinline__ void bc_block_sum_n(double (&v)[N], double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = bc_wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[i * 17 + wave] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[threadIdx.x * 17 + w];
    red[threadIdx.x * 17 + 16] = t;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = red[i * 17 + 16];
}

#include "bc_np_sum.h"   // bc_np_sum_const_*: NumPy's pairwise sum of n equal numbers (the mean of a constant row)

// element (row r, sample k) of a tiled Phi
__device__ __forceinline__ size_t bc_tile_off(long long r, int k, int s) {
  return (size_t)(r >> 7) * (size_t)s * BC_TILE + (size_t)k * BC_TILE + (size_t)(r & (BC_TILE - 1));
}

#endif  // __HIPCC__
