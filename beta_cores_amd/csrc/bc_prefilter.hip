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
//   If the candidate lists overflow, no record is produced; the step is marked and the host re-runs it with the
//   exact fp64 sweep, stream-ordered (bc_rescore_dev.h, bc_snnls.hip: pf_overflow).
//
// Algorithmic traffic per row: fp16 2*S + 8 (norm) + 4 (U written) bytes; fp32 4*S + 12.
//
// The fp16 kernel accumulates in fp32 (v_cvt_f32_f16 + v_pk_fma_f32: both dot products of a row in one packed
// FMA) and double-buffers its loads in registers; its planes are padded with zeros to a multiple of BC_HU and
// the sweep vectors carry a zero tail (BC_V_PAD, bc_snnls.hip), so there is no remainder loop.
#include "bc_rescore_dev.h"
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
  const int* qv = nullptr;    // int8: the owner's pre-quantised sweep vector (bc_i8_quant.h), or nullptr
  int* u8 = nullptr;          // int8: [ptiles][sp4][256] dwords (bc_prefilter_i8.h)
  bc_hq2* rowq = nullptr;     // int8: (scale, delta) per row, two halfs
  float2* tile_cand = nullptr;   // int8: [ptiles][4]
  int* tile_ncand = nullptr;     // int8: [ptiles]
  int2* blk_cand = nullptr;      // int8: [grid][BC_BLK_NC] the sweep blocks' own candidate lists (round 5)
  int* blk_nc = nullptr;         // int8: [grid]
  int sp4 = 0;                // int8: k-groups stored per tile
  int prec = 32;              // 32, 16 or 8
  int ptile = BC_PTILE;       // rows per pre-filter tile (256 for fp32, 512 for fp16)
  int sp = 0;                 // fp16: planes stored per tile
  float* ub = nullptr;        // [ptiles*256] upper bounds of the last sweep
  float* tile_u = nullptr;    // [ptiles] per-tile maximum of the upper bounds
  double* blk_l = nullptr;    // [grid] block maxima of the lower bounds
  float* blk_u = nullptr;     // [grid] block maxima of the upper bounds (lets the selection skip whole blocks of tiles)
  // int8, branch-and-bound form (bc_prefilter_bb.h): the sweep rescores its candidates itself and leaves one record per block
  bool bb = false;
  unsigned bb_seq = 0;             // sweep sequence number: validates the device-wide bound without ever resetting it
  unsigned long long* bb_theta = nullptr;
  BbRec* bb_rec = nullptr;         // [grid]
  double* bb_col = nullptr;        // [grid][S]
  // int8, two-level form (bc_prefilter_i4.h): a 4-bit first level in front of a row-major copy of the int8 mirror
  bool two = false;
  int i4_u = 13;                   // loads per batch of the 4-bit sweep (template parameter)
  int sp8 = 0, g4 = 0, rb = 0;     // k-groups of 8 stored per tile; ceil(S / 4); bytes of an int8 row record
  int* u4 = nullptr;               // [ptiles][sp8][256]
  unsigned short* rowq4 = nullptr; // [ptiles*256]
  unsigned char* r8 = nullptr;     // [ptiles*256][rb]
  const int* qv4 = nullptr;        // the owner's sweep vector in 4-bit digits (bc_i4_quant.h), or nullptr
  long long* hot = nullptr;        // [BC_I4_SEEDS + BC_I4_HOT] seeds
  unsigned l1_seq = 0;             // two-level sweeps launched
  // the host's watch over the first level (bc_pref_adapt): on data whose top scores the 4-bit bounds do not separate from the
  // bulk it passes on a large share of the rows, and the one-level int8 sweep is the faster form
  bool two_active = true;
  unsigned chk_l1 = 0;             // l1_seq and the device's row total at the last check
  unsigned long long chk_rows = 0;
  long long int8_since = 0;        // one-level sweeps since the two-level form was put aside
  long long reprobe = 256;         // ... after which it is tried again (doubled every time it fails)
  double last_share = -1.;         // share of the rows passed on, last window (diagnostic)
  long long sweeps_launched = 0;   // any form: the first sweep has no seeds and runs as the plain int8 sweep
  int grid1 = 1;
  double* l2_blk_l = nullptr;      // [grid1] the two-level sweep's block lists (the int8 sweep keeps its own)
  float* l2_blk_u = nullptr;
  int2* l2_blk_cand = nullptr;
  int* l2_blk_nc = nullptr;
  int2* spill = nullptr;           // [cap] pairs of blocks with more than BC_BLK_NC rows in play
  long long* cand = nullptr;  // [cap] candidate LOCAL rows
  int* ctrl = nullptr;        // [1] the last rescoring overflowed, [3] overflows so far,
                              // [4..5] sweeps so far (u64), [6..7] candidates rescored so far (u64); two-level form: [9] position
                              // in the seeds' ring, [12..13] rows passed on by the first level so far (u64), [14] pairs in the spill
                              // list of the sweep in flight; branch-and-bound form: [16..17] its shared bound
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

// BC_PREF_DELTA (bc_rescore_dev.h): bound of |fp64-kernel dot - fp32-input dot| for unit ||v||: 2^-24 (input rounding,
// Cauchy-Schwarz with ||u_i|| <= 1 + 2^-24) plus the two fma chains' rounding (S * 2^-53 each), with margin.

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
#include "bc_prefilter_bb.h"
#include "bc_prefilter_i4.h"

// ---- fp16 variant.  Tile = 512 rows, [S][512] halfs: one sample of a tile = 1 KiB = 64 lanes x 8 halfs.
typedef _Float16 bc_h8 __attribute__((ext_vector_type(8)));

// bound of |fp64-kernel dot - (fp16 rows, fp32 v, fp32 fma chain) dot| for unit ||v||, ||u|| = 1:
//   rows:   |u^ - u|_2 <= (2^-11 + 2^-23) (RN to half through float, normal range) + sqrt(S) 2^-25 (subnormal halfs)
//   v:      2^-24 (RN to float)
//   chain:  S roundings of 2^-24 relative to sum |u^_i v^_i| <= ||u^|| ||v^||
// every term taken with margin.
static double bc_pref_delta16(int S) { return 4.8845e-4 + 3.0e-8 * sqrt((double)S) + 6.1e-8 * (double)(S + 2); }


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

// passes B + C as their own launch (one block): multi-rank steps (the record goes into the exchange) and the
// step-wise protocol.  On overflow the record carries BC_REC_OVERFLOW in its `valid` slot: whoever consumes the
// gathered records (finish / pick kernels, every rank alike) turns that into "redo this step with the exact sweep".
#define BC_RESCORE_THREADS 512      // (a thread of the rescoring stage holds the candidate lists of two sweep blocks: 1024 blocks)
template <int MODE>
__global__ __launch_bounds__(BC_RESCORE_THREADS) void k_rescore(RescoreArgs a, long long n_rows) {
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;
  if (skip) return;
  const RescorePre pre = bc_rescore_prefetch(a);
  if (bc_rescore_block<MODE>(a, n_rows, a.rec, pre)) {
    if (threadIdx.x == 0) {
      a.rec[0] = -INFINITY;
      reinterpret_cast<long long*>(a.rec)[1] = -1;
      a.rec[2] = 0.0;
      a.rec[3] = BC_REC_OVERFLOW;
    }
  }
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
  bool want_two = prec == 4;
  if (prec == 4) prec = 8;
  p->prec = prec == 16 ? 16 : (prec == 8 ? 8 : 32);
  if (p->prec == 8 && (phi->s + 3) / 4 > BC_IMAXG - BC_IU) p->prec = 16;      // the digit table of k_sweep_i8 holds S <= ~1260
  p->ptile = p->prec == 16 ? BC_HTILE : (p->prec == 8 ? BC_ITILE : BC_PTILE);
  p->sp = p->prec == 16 ? (phi->s + BC_HU - 1) / BC_HU * BC_HU : phi->s;
  p->sp4 = bc_lay_i8_sp4(phi->s);
  p->ptiles = (phi->n_rows + p->ptile - 1) / p->ptile;
  if (p->ptiles < 1) p->ptiles = 1;
  static_assert(BC_BLK_NC == BC_RS_BLK_NC, "bc_rescore_dev.h mirrors the block list length");
  static_assert(BC_ITILE == BC_LAY_ITILE && BC_IU == BC_LAY_IU && BC_TILE == BC_LAY_TILE && BC_IMAXG == BC_LAY_IMAXG, "bc_layout.h mirrors these");
  // one wave per tile and a grid-stride loop: size the grid so that all its waves are resident at once
  // (4 per SIMD) and every wave walks the same number of tiles -- a 2x over-subscribed grid left waves with
  // 2 or 3 tiles each (79% balance at 10M rows)
  // fp16 / fp32 sweeps: 4 waves per SIMD fit (<= 128 VGPRs); the int8 sweep holds 148 VGPRs: 3 per SIMD would fit, 2 are
  // used (8 waves per CU: 0.805 of the HBM spec against 0.798 with 12; 6 -> 0.71, 4 -> 0.57: below 8 the 5 KiB a wave keeps
  // in flight no longer cover the latency)
  long long wmax = (long long)ctx->n_cu * (p->prec == 8 ? 8 : 16);
  const char* genv = getenv("BC_PREF_WAVES_PER_CU");
  if (genv && atoi(genv) > 0) wmax = (long long)ctx->n_cu * atoi(genv);
  const long long rounds = (p->ptiles + wmax - 1) / wmax;
  const long long waves = (p->ptiles + rounds - 1) / rounds;
  p->grid = (int)((waves + 3) / 4);
  if (p->grid < 1) p->grid = 1;
  if (p->grid > 1024) p->grid = 1024;       // k_rescore keeps the block bounds in 4 registers per thread
  // branch-and-bound form of the int8 sweep (bc_prefilter_bb.h): OPT-IN, BC_I8_BB=1.  Measured at N = 10M (profiles/r05_notes.md):
  // the step kernel's tail drops by 4.6 us, the sweep grows by 7-14 us (posts of a wave's last tile are rescored after the
  // stream has ended; every sharing of the bound across blocks costs more than it saves) -- the two-pass form stays the default.
  if (p->prec == 8 && phi->s <= 256 && (long long)p->ptiles * BC_ITILE < 2147483647LL) {
    const char* benv = getenv("BC_I8_BB");
    p->bb = benv ? atoi(benv) != 0 : false;
    if (p->bb) {                               // BC_BB_SW streaming waves (+ the rescoring wave) per block, one block per CU
      p->grid = (int)((waves + BC_BB_SW - 1) / BC_BB_SW);
      if (p->grid < 1) p->grid = 1;
    }
  }
  // two-level form: S <= 256 (digit tables in LDS), local rows as 32-bit integers, not together with the branch-and-bound sweep
  p->two = want_two && p->prec == 8 && !p->bb && phi->s <= 256 && (long long)p->ptiles * BC_ITILE < 2147483647LL;
  if (p->two) {
    p->i4_u = bc_lay_i4_batch(phi->s);
    p->sp8 = bc_lay_i4_sp8(phi->s, p->i4_u);
    p->g4 = (phi->s + 3) / 4;
    p->rb = bc_lay_r8_bytes(phi->s);
    p->grid1 = p->grid;                        // 8 waves per CU, every wave the same number of tiles (as the int8 sweep)
    static_assert(BC_I4_SEEDS == 256, "k_sweep_i4 evaluates one seed per thread");
  }
  // the fp64 tiles cover ntiles*128 rows; the unit rows cover ptiles*ptile >= that, reads past the fp64 tiles are masked by `live`
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_u = take(p->prec == 8 ? bc_lay_i8_words(p->ptiles, p->sp4) * sizeof(int)
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
  const size_t o_bc = take(p->prec == 8 ? (size_t)p->grid * BC_BLK_NC * sizeof(int2) : 0);
  const size_t o_bn = take(p->prec == 8 ? (size_t)p->grid * sizeof(int) : 0);
  const size_t o_bbr = take(p->bb ? (size_t)p->grid * sizeof(BbRec) : 0);
  const size_t o_bbc = take(p->bb ? (size_t)p->grid * phi->s * sizeof(double) : 0);
  const size_t o_u4 = take(p->two ? bc_lay_i4_words(p->ptiles, p->sp8) * sizeof(int) : 0);
  const size_t o_rq4 = take(p->two ? (size_t)p->ptiles * BC_ITILE * sizeof(unsigned short) : 0);
  const size_t o_r8 = take(p->two ? (size_t)p->ptiles * BC_ITILE * p->rb : 0);
  const size_t o_hot = take(p->two ? (BC_I4_SEEDS + BC_I4_HOT) * sizeof(long long) : 0);
  const size_t o_l2l = take(p->two ? (size_t)p->grid1 * sizeof(double) : 0);
  const size_t o_l2u = take(p->two ? (size_t)p->grid1 * sizeof(float) : 0);
  const size_t o_l2c = take(p->two ? (size_t)p->grid1 * BC_BLK_NC * sizeof(int2) : 0);
  const size_t o_l2n = take(p->two ? (size_t)p->grid1 * sizeof(int) : 0);
  const size_t o_sp = take(p->two ? (size_t)4096 * sizeof(int2) : 0);
  hipError_t e = hipMalloc(&p->slab, off);
  if (e != hipSuccess) { delete p; return bc_hip_fail(e, "hipMalloc(prefilter)", __FILE__, __LINE__); }
  char* base = (char*)p->slab;
  if (p->two) {
    p->u4 = (int*)(base + o_u4);
    p->rowq4 = (unsigned short*)(base + o_rq4);
    p->r8 = (unsigned char*)(base + o_r8);
    p->hot = (long long*)(base + o_hot);
    p->l2_blk_l = (double*)(base + o_l2l);
    p->l2_blk_u = (float*)(base + o_l2u);
    p->l2_blk_cand = (int2*)(base + o_l2c);
    p->l2_blk_nc = (int*)(base + o_l2n);
    p->spill = (int2*)(base + o_sp);
  }
  p->u32 = p->prec == 32 ? (float*)(base + o_u) : nullptr;
  p->u16 = p->prec == 16 ? (_Float16*)(base + o_u) : nullptr;
  p->ub = p->prec != 32 ? nullptr : (float*)(base + o_ub);
  if (p->prec == 8) {
    p->u8 = (int*)(base + o_u);
    p->rowq = (bc_hq2*)(base + o_rq);
    p->tile_cand = (float2*)(base + o_tc);
    p->tile_ncand = (int*)(base + o_tn);
    // the sweep blocks' own candidate lists (round 5): OPT-IN, BC_I8_BLKLIST=1.  They take the tile walk out of the rescoring
    // stage (-2k of its 20k cycles) and put ~0.7 us of list building at the end of every sweep block: no gain at N = 10M
    // (5 587-5 655 against 5 640-5 671 it/s) nor at a 1.25M-row shard (50.7-50.9 against 50.2 us per step), profiles/r05_notes.md
    const char* lenv = getenv("BC_I8_BLKLIST");
    if (lenv && atoi(lenv) != 0) {
      p->blk_cand = (int2*)(base + o_bc);
      p->blk_nc = (int*)(base + o_bn);
    }
  }
  p->live = p->prec == 16 ? (unsigned char*)(base + o_lv) : nullptr;
  p->tile_u = (float*)(base + o_tu);
  p->blk_l = (double*)(base + o_bl);
  p->blk_u = (float*)(base + o_bu);
  p->cand = (long long*)(base + o_c);
  p->ctrl = (int*)(base + o_ctrl);
  if (p->bb) {
    p->bb_theta = reinterpret_cast<unsigned long long*>(p->ctrl + 16);      // inside the zeroed control block (sequence 0 = "none")
    p->bb_rec = (BbRec*)(base + o_bbr);
    p->bb_col = (double*)(base + o_bbc);
  }
  e = hipMemsetAsync(p->ctrl, 0, 256, ctx->stream);
  if (e == hipSuccess) {
    if (p->prec == 8)
    {
      const int two_pass = getenv("BC_BUILD_I8_TWO_PASS") ? atoi(getenv("BC_BUILD_I8_TWO_PASS")) : 0;   // (A/B and tests: read per call)
      if (phi->s <= 104 && !two_pass)        // (S <= 104 covers every BASELINE configuration; two-pass kernel for the rest)
        hipLaunchKernelGGL(k_build_i8_r<104>, dim3((unsigned)p->ptiles), dim3(256), 0, ctx->stream, phi->tiles, phi->norms,
                           (long long)phi->n_rows, phi->s, p->sp4, p->u8, p->rowq);
      else
        hipLaunchKernelGGL(k_build_i8, dim3((unsigned)p->ptiles), dim3(256), 0, ctx->stream, phi->tiles, phi->norms,
                           (long long)phi->n_rows, phi->s, p->sp4, p->u8, p->rowq);
    }
    if (p->two && e == hipSuccess) {
      e = hipMemsetAsync(p->hot, 0xff, (BC_I4_SEEDS + BC_I4_HOT) * sizeof(long long), ctx->stream);            // -1: no seeds yet
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_build_i4, dim3((unsigned)p->ptiles), dim3(256), 0, ctx->stream, phi->tiles, phi->norms,
                           (long long)phi->n_rows, phi->s, p->sp8, p->u8, p->rowq, p->sp4, p->g4, p->u4, p->rowq4);
        hipLaunchKernelGGL(k_build_r8, dim3((unsigned)p->ptiles), dim3(256), 0, ctx->stream, p->u8, p->rowq, p->sp4, p->g4, p->rb, p->r8);
      }
    }
    if (p->prec == 8) {
    } else if (p->prec == 16)
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

// pass A: the reduced-precision sweep (bounds per tile / block); fills the argument block of passes B + C
int bc_pref_launch_sweep(bc_pref* p, int mode, const double* v_dev, const double* v_norm_dev, double post_div,
                         const int* skip_flag, double* rec_dev, RescoreArgs* r_out) {
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
  // two-level form: needs both digit records from the owner's step kernels and seeds from an earlier rescoring
  const bool two = p->two && p->two_active && p->qv != nullptr && p->qv4 != nullptr && p->sweeps_launched > 0;
  p->sweeps_launched += 1;
  if (p->two && !p->two_active) p->int8_since += 1;
  if (two) {
    I4Args fa;
    fa.u4 = p->u4;
    fa.rowq4 = p->rowq4;
    fa.qv4 = p->qv4;
    fa.qv8 = p->qv;
    fa.r8 = p->r8;
    fa.hot = p->hot;
    fa.skip_flag = skip_flag;
    fa.blk_l = p->l2_blk_l;
    fa.blk_u = p->l2_blk_u;
    fa.blk_cand = p->l2_blk_cand;
    fa.blk_nc = p->l2_blk_nc;
    fa.ctrl = p->ctrl;
    fa.spill = p->spill;
    fa.spill_cap = 4096;
    fa.ptiles = p->ptiles;
    fa.post_div = post_div;
    p->l1_seq += 1;
    fa.s = phi->s;
    fa.sp8 = p->sp8;
    fa.sp4 = p->sp4;
    fa.g4 = p->g4;
    fa.rb = p->rb;
#define BC_I4_LAUNCH(MODE, UU) hipLaunchKernelGGL((k_sweep_i4<MODE, UU>), dim3(p->grid1), dim3(256), 0, ctx->stream, fa)
#define BC_I4_BY_U(MODE)                                                                          \
    switch (p->i4_u) {                                                                            \
      case 13: BC_I4_LAUNCH(MODE, 13); break;                                                     \
      case 8: BC_I4_LAUNCH(MODE, 8); break;                                                       \
      case 7: BC_I4_LAUNCH(MODE, 7); break;                                                       \
      case 6: BC_I4_LAUNCH(MODE, 6); break;                                                       \
      default: BC_I4_LAUNCH(MODE, 5); break;                                                      \
    }
    if (mode == 0) { BC_I4_BY_U(0) } else { BC_I4_BY_U(1) }
    BC_HIP(hipGetLastError());
    rc = bc_timer_end(ctx, 0);
    if (rc) return rc;
  } else
  if (p->prec == 8) {
    I8Args ia;
    ia.u8 = p->u8;
    ia.rowq = p->rowq;
    ia.v = v_dev;
    ia.qv = p->qv;
    ia.skip_flag = skip_flag;
    ia.v_norm = v_norm_dev;
    ia.tile_u = p->tile_u;
    ia.tile_cand = p->tile_cand;
    ia.tile_ncand = p->tile_ncand;
    ia.blk_l = p->blk_l;
    ia.blk_u = p->blk_u;
    ia.blk_cand = p->bb ? nullptr : p->blk_cand;
    ia.blk_nc = p->bb ? nullptr : p->blk_nc;
    ia.ptiles = p->ptiles;
    ia.post_div = post_div;
    ia.s = phi->s;
    ia.sp4 = p->sp4;
    if (p->bb) {
      I8BbArgs ba;
      ba.a = ia;
      ba.tiles = phi->tiles;
      ba.norms = phi->norms;
      ba.row_offset = phi->row_offset;
      ba.theta = p->bb_theta;
      if (++p->bb_seq == 0u) p->bb_seq = 1u;       // (2^32 sweeps later: the word still holds a larger sequence number -- start over)
      if (p->bb_seq == 1u) BC_HIP(hipMemsetAsync(p->bb_theta, 0, sizeof(unsigned long long), ctx->stream));
      ba.seq = p->bb_seq;
      ba.blk_rec = p->bb_rec;
      ba.blk_col = p->bb_col;
      ba.max_res = p->cap < BC_BB_MAXRES ? p->cap : BC_BB_MAXRES;
      if (mode == 0) hipLaunchKernelGGL(k_sweep_i8_bb<0>, dim3(p->grid), dim3(BC_BB_THREADS), 0, ctx->stream, ba);
      else hipLaunchKernelGGL(k_sweep_i8_bb<1>, dim3(p->grid), dim3(BC_BB_THREADS), 0, ctx->stream, ba);
    } else if (mode == 0) hipLaunchKernelGGL(k_sweep_i8<0>, dim3(p->grid), dim3(256), 0, ctx->stream, ia);
    else hipLaunchKernelGGL(k_sweep_i8<1>, dim3(p->grid), dim3(256), 0, ctx->stream, ia);
  } else if (p->prec == 16) {
    if (mode == 0) hipLaunchKernelGGL(k_sweep_f16<0>, dim3(p->grid), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_sweep_f16<1>, dim3(p->grid), dim3(256), 0, ctx->stream, a);
  } else {
    if (mode == 0) hipLaunchKernelGGL(k_sweep_f32<0>, dim3(p->grid), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_sweep_f32<1>, dim3(p->grid), dim3(256), 0, ctx->stream, a);
  }
  if (!two) {
    BC_HIP(hipGetLastError());
    rc = bc_timer_end(ctx, 0);
    if (rc) return rc;
  }
  RescoreArgs& r = *r_out;
  r.bb.rec = p->bb ? p->bb_rec : nullptr;
  r.bb.col = p->bb_col;
  r.bb.nblk = p->grid;
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
  r.blk_cand = (p->prec == 8 && !p->bb) ? p->blk_cand : nullptr;
  r.blk_nc = (p->prec == 8 && !p->bb) ? p->blk_nc : nullptr;
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
  r.two_level = two ? 1 : 0;
  r.spill = p->spill;
  r.spill_cap = 4096;
  r.hot = p->two ? p->hot + BC_I4_SEEDS : nullptr;      // (the ring behind the sweep blocks' slots)
  if (two) {
    // the two-level sweep's block lists; there are no tiles behind them
    r.blk_l = p->l2_blk_l;
    r.blk_u = p->l2_blk_u;
    r.blk_cand = p->l2_blk_cand;
    r.blk_nc = p->l2_blk_nc;
    r.nblk = p->grid1;
    r.tile_u = nullptr;
    r.tile_cand = nullptr;
    r.tile_ncand = nullptr;
    r.tile_rounds = 0;
  }
  return BC_OK;
}

// The owner's step kernels keep a quantised copy of v_dev up to date (bc_snnls.hip: dev_prep -> bc_i8q_wave): the sweep
// then skips its own quantisation prologue.  Only valid while EVERY launch of this pre-filter sweeps that same v.
void bc_pref_set_qv(bc_pref* p, const int* qv_dev) { p->qv = (p->prec == 8) ? qv_dev : nullptr; }
// ... and a 4-bit copy for the two-level form's first level (bc_i4_quant.h; sp8 must be bc_pref_sp8)
void bc_pref_set_qv4(bc_pref* p, const int* qv4_dev) { p->qv4 = p->two ? qv4_dev : nullptr; }
int bc_pref_sp8(const bc_pref* p) { return p->two ? p->sp8 : 0; }
int bc_pref_two_level(const bc_pref* p) { return p->two ? 1 : 0; }
// The host's watch over the two-level form; called where the host synchronises with the stream anyway (end of a build call,
// every 64 steps of a long one).  BC_TWO_LEVEL_MAX_SHARE: share of the rows the first level may pass on (default 0.04).
int bc_pref_adapt(bc_pref* p) {
  if (!p || !p->two) return BC_OK;
  static const double max_share = getenv("BC_TWO_LEVEL_MAX_SHARE") ? atof(getenv("BC_TWO_LEVEL_MAX_SHARE")) : 0.04;
  const bool probe_due = !p->two_active && p->int8_since >= p->reprobe;
  const unsigned dl1 = p->l1_seq - p->chk_l1;
  if (!probe_due && !(p->two_active && dl1 >= 4)) return BC_OK;
  unsigned long long rows = 0;
  BC_HIP(hipMemcpyAsync(&rows, p->ctrl + 12, sizeof(rows), hipMemcpyDeviceToHost, p->ctx->stream));
  BC_HIP(hipStreamSynchronize(p->ctx->stream));
  if (p->two_active) {
    const double share = (double)(rows - p->chk_rows) / ((double)dl1 * (double)(p->phi->n_rows > 0 ? p->phi->n_rows : 1));
    p->last_share = share;
    if (share > max_share) {
      p->two_active = false;
      p->int8_since = 0;
    } else {
      p->reprobe = 256;
    }
  } else {
    p->two_active = true;                        // try again (the ring of seeds kept turning meanwhile)
    p->reprobe = p->reprobe < 8192 ? 2 * p->reprobe : p->reprobe;
  }
  p->chk_l1 = p->l1_seq;
  p->chk_rows = rows;
  return BC_OK;
}
int bc_pref_two_level_active(const bc_pref* p) { return (p->two && p->two_active) ? 1 : 0; }

long long bc_pref_l1_sweeps(const bc_pref* p) { return (long long)p->l1_seq; }     // (host count: includes sweeps the device skipped)
int bc_pref_sp4(const bc_pref* p) { return p->sp4; }
int bc_pref_bb(const bc_pref* p) { return p->bb ? 1 : 0; }

// passes A, B, C: sweep, then the rescoring as its own one-block launch (record into rec_dev)
int bc_pref_launch(bc_pref* p, int mode, const double* v_dev, const double* v_norm_dev, double post_div,
                   const int* skip_flag, double* rec_dev) {
  RescoreArgs r;
  int rc = bc_pref_launch_sweep(p, mode, v_dev, v_norm_dev, post_div, skip_flag, rec_dev, &r);
  if (rc) return rc;
  rc = bc_timer_begin(p->ctx, 3);
  if (rc) return rc;
  if (p->bb) hipLaunchKernelGGL(k_bb_winner, dim3(1), dim3(256), 0, p->ctx->stream, r.bb, r.s, r.skip_flag, r.ctrl, r.rec);
  else if (mode == 0) hipLaunchKernelGGL(k_rescore<0>, dim3(1), dim3(BC_RESCORE_THREADS), 0, p->ctx->stream, r, (long long)p->phi->n_rows);
  else hipLaunchKernelGGL(k_rescore<1>, dim3(1), dim3(BC_RESCORE_THREADS), 0, p->ctx->stream, r, (long long)p->phi->n_rows);
  BC_HIP(hipGetLastError());
  return bc_timer_end(p->ctx, 3);
}
