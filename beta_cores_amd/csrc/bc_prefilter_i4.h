// K3 pre-filter, TWO-LEVEL form (BC_PREFILTER=4): the per-iteration sweep streams HALF a byte per element.
//
//   level 1  k_sweep_i4 streams a 4-bit mirror of the normalised rows (q in [-7, 7], per-row scale chosen at build time to
//            minimise the MEASURED error delta4_i ~ 0.1) against the sweep vectors in two 4-bit digits each (bc_i4_quant.h),
//            v_dot8_i32_i4: eight exact products per instruction.  Same interval arithmetic as the int8 sweep
//            (bc_score_interval_f32), wider deltas.  A row is passed on when its upper bound reaches
//            theta = max(theta0, the best lower bound the wave has seen so far); both are lower bounds of the best exact
//            score, so the row the fp64 sweep returns -- and every row tied with it -- is always passed on.  theta0 comes from
//            SEEDS: the strongest row of each of the first 256 sweep blocks of the last two-level step (rows spread over the
//            top of the score distribution) and a ring of the rows that reached the exact rescoring lately, evaluated against
//            the NEW vectors from their int8 records by every block in its prologue (one row per thread; the first tile's
//            loads are in flight meanwhile).  Scores move slowly between steps, so theta0 is usually within a few percent of
//            the new maximum and 0.02-0.5 % of the rows are passed on at N = 10M (without seeds: 5-20 %).
//   level 2  the SAME wave re-bounds the rows it passed on -- parked in LDS, one row per lane, at the end of its walk (or
//            when 128 are parked) -- from a ROW-MAJOR copy of the int8 mirror (one 128-byte line per row at S <= 124; the
//            tile-major mirror would cost 25 lines) with the int8 sweep's arithmetic (bc_i8_row_bounds).  The block keeps its
//            best int8 lower bound and the (upper bound, row) pairs that reach it and leaves them in the format of the int8
//            sweep's block lists, which the rescoring stage (bc_rescore_dev.h) consumes.  (First built as a kernel of its own
//            over one global list: 18 us per step at N = 10M for ~15k rows -- launch, three dependent cold round trips --
//            against ~4 us at the end of the sweep's waves.)
//   level 3  the exact fp64 rescoring of the handful of rows left, unchanged.
//
// A block left with more than BC_BLK_NC rows in play ends in the existing "redo this step with the exact sweep" path; the first
// sweep after creation (no seeds yet) is the plain int8 sweep; the host watches the share of rows passed on and goes back to
// the one-level int8 sweep on data where the 4-bit bounds do not separate the top of the score distribution (bc_prefilter.hip).
//
// Bytes per row and sweep: 4 * SP8 + 2 (S = 100: 54 against the int8 mirror's 104).
#pragma once

#include "bc_layout.h"
#include "bc_i4_quant.h"
#include "bc_i8_quant.h"

#define BC_I4_MAXG 48        // k-groups of 8 the 4-bit digit table holds (S <= 256 plus padding)
#define BC_I4_MAXG8 72       // k-groups of 4 of the int8 digit table (seeds, level 2): S <= 256 plus padding
#define BC_I4_PARK 128       // rows a wave parks in LDS before it appends them to the global list
#define BC_I4_HOT 32         // seeds: ring of rows that were in play lately (local row numbers, -1: empty) ...
#define BC_I4_SEEDS 256      // ... behind one slot per refine block (its strongest row): hot[BC_I4_SEEDS + BC_I4_HOT]
#define BC_I4_DEAD 255       // delta code of a dead row (padding or zero norm)
#define BC_I4_UNCERTAIN 254  // delta code of a row the bound cannot cover (NaN / inf in it, or delta too large for the code)
#define BC_L2_LCAP 512       // rows in play a block can hold before it gives up (-> exact redo); rows that cannot reach the
                             // block's best lower bound so far are never stored

struct I4Args {
  const int* u4;                 // [ptiles][sp8][256] dwords of eight nibbles
  const unsigned short* rowq4;   // [ptiles*256] scale code | delta code << 8
  const int* qv4;                // the sweep vectors in 4-bit digits (bc_i4_quant.h record)
  const int* qv8;                // ... and in int8 digits (bc_i8_quant.h record): level 2 and the seeds
  const unsigned char* r8;       // row-major int8 records (bc_layout.h)
  long long* hot;                // [BC_I4_SEEDS + BC_I4_HOT]: block b < BC_I4_SEEDS leaves its strongest row in slot b
  const int* skip_flag;
  double* blk_l;                 // [grid] outputs in the format of the int8 sweep's block lists (bc_prefilter_i8.h)
  float* blk_u;
  int2* blk_cand;                // [grid][BC_BLK_NC]
  int* blk_nc;
  int* ctrl;                     // [12..13] u64 total: rows passed on by level 1 and re-bounded by level 2; [14]: pairs in `spill`
  int2* spill;                   // [spill_cap] a block with more than BC_BLK_NC rows in play appends ALL its pairs here (blk_nc = -1)
  int spill_cap;
  long long ptiles;
  double post_div;
  int s, sp8, sp4, g4, rb;       // sp4: k-groups of the int8 digit record; g4 = ceil(S / 4); rb: bytes of an int8 record
};

__device__ __forceinline__ int bc_f32_ord(float f) {        // order-preserving map float -> int (for LDS atomicMax)
  const int b = __builtin_bit_cast(int, f);
  return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float bc_ord_f32(int o) { return __builtin_bit_cast(float, o >= 0 ? o : o ^ 0x7fffffff); }

// (U, L) of one row from its int8 record, against the int8 digit table in LDS.  hdr = the record's six header floats.
template <int MODE>
__device__ __forceinline__ void bc_r8_interval(const unsigned char* __restrict__ rec, const int (*dig8)[4], const float* hdr, int g4, int rb,
                                               float fpd, float& U, float& L) {
  const bc_i4* __restrict__ p = reinterpret_cast<const bc_i4*>(rec);
  int a0 = 0, a1 = 0, a2 = 0, lastw = 0;
  const int nch = rb >> 7;
  for (int c = 0; c < nch; ++c) {
    bc_i4 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = p[8 * c + u];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int g = 32 * c + 4 * u + j;
        if (g < g4) {
          const bc_i4 dg = *reinterpret_cast<const bc_i4*>(&dig8[g][0]);
          a0 = __builtin_amdgcn_sdot4(w[u][j], dg[0], a0, false);
          a1 = __builtin_amdgcn_sdot4(w[u][j], dg[1], a1, false);
          if (MODE == 0) a2 = __builtin_amdgcn_sdot4(w[u][j], dg[2], a2, false);
        }
      }
    lastw = w[7][3];
  }
  const bc_hq2 rq = __builtin_bit_cast(bc_hq2, lastw);
  const float sc = (float)rq[0], dl = (float)rq[1];
  if (dl < 0.f) { U = -INFINITY; L = -INFINITY; return; }      // dead row
  bc_i8_row_bounds<MODE>(sc, dl, a0, a1, a2, hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], hdr[5] != 0.f, fpd, U, L);
}

// ------------------------------------------------------------------ builders
// quantise one unit row's element with scale code c (scale = c / 1024)
__device__ __forceinline__ int bc_i4_digit(double u, double iscale) {
  int q = (int)rint(u * iscale);
  return q > 7 ? 7 : (q < -7 ? -7 : q);
}

// Loop over a row's elements: unrolled to SMAX with a `k < S` predicate when the values sit in registers (SMAX > 0; a runtime
// index would send the register array to scratch), a plain loop otherwise.
#define BC_I4_FOR_K(SMAX, S, body)                                   \
  if (SMAX > 0) {                                                    \
    _Pragma("unroll") for (int k = 0; k < (SMAX > 0 ? SMAX : 1); ++k) \
      if (k < S) { body }                                            \
  } else {                                                           \
    for (int k = 0; k < S; ++k) { body }                             \
  }

// The scale search (fp32: a heuristic, any scale is valid) and the final measurement (fp64) for one row whose unit values
// come from `getu(k)`.  Returns the packed 16-bit word; `scale_out` is the chosen scale (0: nothing to store).
template <int SMAX, typename F>
__device__ __forceinline__ unsigned short bc_i4_choose(F getu, int S, bool live, bool has_nan, double mx, double& scale_out) {
  scale_out = 0.;
  if (!live) return (unsigned short)(BC_I4_DEAD << 8);
  if (has_nan || !(mx > 0.)) return (unsigned short)(BC_I4_UNCERTAIN << 8);
  int cb = (int)ceil(mx * (1024. / 7.));            // the smallest code that never clips
  cb = cb < 1 ? 1 : (cb > 253 ? 253 : cb);
  int bestc = cb;
  float beste = INFINITY;
#pragma unroll 1
  for (int f = 0; f < 8; ++f) {
    int c = (int)((float)cb * (1.f - 0.1f * (float)f) + 0.5f);       // cb x {1, .9, .8, .7, .6, .5, .4, .3}
    c = c < 1 ? 1 : c;
    const float sc = (float)c * (1.f / 1024.f), isc = 1024.f / (float)c;
    float e = 0.f;
    BC_I4_FOR_K(SMAX, S, {
      const float u = (float)getu(k);
      float q = rintf(u * isc);
      q = q > 7.f ? 7.f : (q < -7.f ? -7.f : q);
      const float d = q * sc - u;
      e = fmaf(d, d, e);
    })
    if (e < beste) { beste = e; bestc = c; }
  }
  const double scale = (double)bestc * (1. / 1024.);
  const double iscale = 1024. / (double)bestc;
  double err2 = 0.;
  BC_I4_FOR_K(SMAX, S, {
    const double u = getu(k);
    const double d = (double)bc_i4_digit(u, iscale) * scale - u;
    err2 = fma(d, d, err2);
  })
  const double dd = sqrt(err2) * (1. + 1e-6) + 1e-12;
  const int dc = (int)ceil(dd * 512.);               // delta = dc / 512 >= dd
  if (!(dc <= 250)) return (unsigned short)(BC_I4_UNCERTAIN << 8);
  scale_out = scale;
  return (unsigned short)(bestc | (dc << 8));
}

// one block per 256-row tile, thread = row.  SMAX > 0: S <= SMAX and the row's values are loaded ONCE into registers (all loads
// in flight, as k_build_i8_r); SMAX == 0: any S, every pass re-reads the row from Phi (L2-resident: 100 KB per 128-row tile).
template <int SMAX>
__global__ __launch_bounds__(256) void k_build_i4(const double* __restrict__ tiles, const double* __restrict__ norms, long long n_rows,
                                                 int S, int SP8, int* __restrict__ u4, unsigned short* __restrict__ rowq4) {
  const long long t = blockIdx.x;
  const long long r = t * BC_ITILE + threadIdx.x;
  const bool live = r < n_rows && norms[r < n_rows ? r : 0] != 0.;
  const double inr = live ? 1. / norms[r] : 0.;       // (one reciprocal per row: see k_build_i8)
  const long long rr = bc_lay_i8_src_row(r, n_rows);  // (rows past the end read row 0 and are dead)
  const double* p = tiles + bc_lay_phi_elem(rr, 0, S);
  double ur[SMAX > 0 ? SMAX : 1];
  if (SMAX > 0) {
#pragma unroll
    for (int k = 0; k < (SMAX > 0 ? SMAX : 1); ++k) ur[k] = (k < S && n_rows > 0) ? p[(size_t)k * BC_TILE] * inr : 0.;
  }
  auto getu = [&](int k) -> double { return SMAX > 0 ? ur[SMAX > 0 ? k : 0] : p[(size_t)k * BC_TILE] * inr; };
  double mx = 0.;
  bool has_nan = false;
  if (live) {
    BC_I4_FOR_K(SMAX, S, {
      const double u = getu(k);
      has_nan |= !(fabs(u) <= 1.7976931348623157e308);
      mx = fmax(mx, fabs(u));
    })
  }
  double scale;
  const unsigned short code = bc_i4_choose<SMAX>(getu, S, live, has_nan, mx, scale);
  const double iscale = scale > 0. ? 1. / scale : 0.;
  int* q = u4 + bc_lay_i4_word(t, 0, threadIdx.x, SP8);
  if (SMAX > 0) {
#pragma unroll
    for (int g = 0; g < (SMAX + 7) / 8; ++g) {
      if (g < SP8) {
        unsigned w = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * g + j;
          if (k < (SMAX > 0 ? SMAX : 1)) {
            const int d = bc_i4_digit(ur[k < (SMAX > 0 ? SMAX : 1) ? k : 0], iscale);
            w |= (k < S && scale > 0.) ? ((unsigned)d & 0xfu) << (4 * j) : 0u;
          }
        }
        q[(size_t)g * BC_ITILE] = (int)w;
      }
    }
    for (int g = (SMAX + 7) / 8; g < SP8; ++g) q[(size_t)g * BC_ITILE] = 0;
  } else {
    for (int g = 0; g < SP8; ++g) {
      unsigned w = 0;
      if (scale > 0.) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * g + j;
          if (k < S) w |= ((unsigned)bc_i4_digit(getu(k), iscale) & 0xfu) << (4 * j);
        }
      }
      q[(size_t)g * BC_ITILE] = (int)w;
    }
  }
  rowq4[r] = code;
}

// row-major copy of the int8 mirror: one block per 256-row tile, thread = row
__global__ __launch_bounds__(256) void k_build_r8(const int* __restrict__ u8, const bc_hq2* __restrict__ rowq, int SP4, int G4, int RB,
                                                 unsigned char* __restrict__ r8) {
  const long long t = blockIdx.x;
  const long long r = t * BC_ITILE + threadIdx.x;
  const int* src = u8 + bc_lay_i8_word(t, 0, threadIdx.x, SP4);
  bc_i4* dst = reinterpret_cast<bc_i4*>(r8 + (size_t)r * RB);
  const int last = RB / 4 - 1;
  for (int c = 0; c < RB / 16; ++c) {
    bc_i4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = 4 * c + j;
      w[j] = g < G4 ? src[(size_t)g * BC_ITILE] : (g == last ? __builtin_bit_cast(int, rowq[r]) : 0);
    }
    dst[c] = w;
  }
}

// ------------------------------------------------------------------ levels 1 + 2
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_sweep_i4(I4Args a) {
  constexpr int NV = (MODE == 0) ? 4 : 2;           // (vector, digit): v0 d0, v0 d1 [, v1 d0, v1 d1]
  __shared__ __attribute__((aligned(16))) int dig4[BC_I4_MAXG][4];
  __shared__ __attribute__((aligned(16))) int dig8[BC_I4_MAXG8][4];
  __shared__ float hdr8[8];
  __shared__ float sl[4];
  __shared__ int park[4][BC_I4_PARK];                  // rows a wave has passed on and not yet re-bounded
  __shared__ float l_u[BC_L2_LCAP];                    // the block's rows in play: int8 upper bound, row
  __shared__ int l_row[BC_L2_LCAP];
  __shared__ int s_n, s_umax, s_on, s_refined, s_refined_base;
  __shared__ unsigned long long s_best;                // (ordered int8 lower bound << 32 | row): the block's strongest row
  __shared__ int2 s_out[BC_BLK_NC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;
  if (skip) return;
  const int SP8 = a.sp8;
  // the first tile's loads and the seed rows' numbers do not depend on the prologue: in flight before it
  long long t = (long long)blockIdx.x * 4 + wave;
  bc_i4 x[U], y[U];
  uint2 rq = make_uint2(0xffffffffu, 0xffffffffu);    // (dead)
  if (t < a.ptiles) {
    const bc_i4* __restrict__ p0 = reinterpret_cast<const bc_i4*>(a.u4 + (size_t)t * SP8 * BC_ITILE) + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p0 + (size_t)u * 64);
    rq = reinterpret_cast<const uint2*>(a.rowq4 + t * BC_ITILE)[lane];
  }
  const long long hot_row = a.hot[threadIdx.x];                                              // (BC_I4_SEEDS == blockDim.x)
  const long long ring_row = (threadIdx.x < BC_I4_HOT) ? a.hot[BC_I4_SEEDS + threadIdx.x] : -1;
  // ---- prologue: both digit records into LDS
  for (int g = threadIdx.x; g < SP8; g += blockDim.x)
    *reinterpret_cast<bc_i4*>(&dig4[g][0]) = reinterpret_cast<const bc_i4*>(a.qv4)[g];
  for (int g = threadIdx.x; g < a.sp4; g += blockDim.x)
    *reinterpret_cast<bc_i4*>(&dig8[g][0]) = reinterpret_cast<const bc_i4*>(a.qv8)[g];
  if (threadIdx.x < 8) hdr8[threadIdx.x] = reinterpret_cast<const float*>(a.qv8 + 4 * a.sp4)[threadIdx.x];
  if (threadIdx.x == 0) { s_n = 0; s_best = 0ull; s_umax = bc_f32_ord(-INFINITY); s_on = 0; s_refined = 0; }
  const float* hf = reinterpret_cast<const float*>(a.qv4 + 4 * SP8);
  const float fvs0 = hf[0], fvs1 = hf[1], fev0 = hf[2], fev1 = hf[3], fvn = hf[4];
  const bool vbad = hf[5] != 0.f;
  const float fpd = (float)a.post_div;
  __syncthreads();
  // ---- seeds: the int8 lower bound of a row under the NEW vectors is a lower bound of the best exact score
  {
    float seed = -INFINITY;
    if (hot_row >= 0) {
      float Us, Ls;
      bc_r8_interval<MODE>(a.r8 + (size_t)hot_row * a.rb, dig8, hdr8, a.g4, a.rb, fpd, Us, Ls);
      if (Ls == Ls) seed = Ls;
    }
    if (ring_row >= 0) {                               // (wave 0 only)
      float Us, Ls;
      bc_r8_interval<MODE>(a.r8 + (size_t)ring_row * a.rb, dig8, hdr8, a.g4, a.rb, fpd, Us, Ls);
      if (Ls == Ls) seed = fmaxf(seed, Ls);
    }
    seed = bc_wave_max_f32_all(seed);
    if (lane == 0) sl[wave] = seed;
  }
  __syncthreads();
  const float theta0 = fmaxf(fmaxf(sl[0], sl[1]), fmaxf(sl[2], sl[3]));

  const long long tstride = (long long)gridDim.x * 4;
  float wave_l = -INFINITY;                            // best 4-bit lower bound of the tiles this wave has finished
  int np = 0;                                          // rows parked by this wave
  // ---- level 2: the parked rows, one per lane, from their int8 records
  auto refine = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < np; base += 64) {
      if (base + lane < np) {
        const int row = park[wave][base + lane];
        float U8, L8;
        bc_r8_interval<MODE>(a.r8 + (size_t)row * a.rb, dig8, hdr8, a.g4, a.rb, fpd, U8, L8);
        if (U8 != -INFINITY) {                         // (a dead row cannot have been passed on; kept for symmetry)
          // key: the lower bound in an unsigned order (-inf -> 0x007fffff > 0: an all-uncertain block still names a row)
          const unsigned long long key = ((unsigned long long)((unsigned)bc_f32_ord(L8) ^ 0x80000000u) << 32) | (unsigned)row;
          const unsigned long long was = atomicMax(&s_best, key);
          atomicMax(&s_umax, bc_f32_ord(U8));
          // only a row whose upper bound reaches the best lower bound seen SO FAR can reach the final one: the others are
          // dropped here, which keeps the list short however many rows the block re-bounds
          const unsigned long long cur = was > key ? was : key;
          if (U8 >= bc_ord_f32((int)((unsigned)(cur >> 32) ^ 0x80000000u))) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < BC_L2_LCAP) { l_u[slot] = U8; l_row[slot] = row; }
          }
        }
      }
    }
    if (lane == 0) atomicAdd(&s_refined, np);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    np = 0;
  };
  for (; t < a.ptiles; t += tstride) {
    const bc_i4* __restrict__ p = reinterpret_cast<const bc_i4*>(a.u4 + (size_t)t * SP8 * BC_ITILE) + lane;
    int acc[4][NV];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < NV; ++c) acc[j][c] = 0;
    const uint2 cq = rq;                               // this tile's codes x 4 rows
    for (int g0 = 0; g0 < SP8; g0 += U) {
      const bool more = g0 + U < SP8;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(p + (size_t)(g0 + U + u) * 64);
      } else if (t + tstride < a.ptiles) {
        const long long tn = t + tstride;
        const bc_i4* __restrict__ pn = reinterpret_cast<const bc_i4*>(a.u4 + (size_t)tn * SP8 * BC_ITILE) + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(pn + (size_t)u * 64);
        rq = reinterpret_cast<const uint2*>(a.rowq4 + tn * BC_ITILE)[lane];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bc_i4 dg = *reinterpret_cast<const bc_i4*>(&dig4[g0 + u][0]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c = 0; c < NV; ++c) acc[j][c] = __builtin_amdgcn_sdot8(x[u][j], dg[c], acc[j][c], false);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = y[u];
    }
    // ---- per-row intervals (4 rows per lane)
    const unsigned codes[4] = {cq.x & 0xffffu, cq.x >> 16, cq.y & 0xffffu, cq.y >> 16};
    const float theta = fmaxf(theta0, wave_l);
    float Ub[4];
    float tl = -INFINITY;
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned dc = codes[j] >> 8;
      Ub[j] = -INFINITY;
      if (dc != BC_I4_DEAD) {
        float Lb;
        if (vbad || dc == BC_I4_UNCERTAIN) {
          Ub[j] = INFINITY;
          Lb = -INFINITY;
        } else {
          const float sc = (float)(codes[j] & 0xffu) * (1.f / 1024.f), dr = (float)dc * (1.f / 512.f);      // exact
          // u^.v^ : exact integers (|16 A0 + A1| < 2^24), scaled in fp32 (relative error < 2e-7, covered by the 4e-7 terms)
          const float s0 = sc * fvs0 * (16.f * (float)acc[j][0] + (float)acc[j][1]);
          const float s1 = (MODE == 0) ? sc * fvs1 * (16.f * (float)acc[j][NV - 2] + (float)acc[j][NV - 1]) : 0.f;
          const float delta0 = (dr * fvn + (1.f + dr) * fev0) * 1.00001f + 4e-7f * fabsf(s0) + 1e-12f;
          const float delta1 = (MODE == 0) ? (dr * fvn + (1.f + dr) * fev1) * 1.00001f + 4e-7f * fabsf(s1) + 1e-12f : 0.f;
          bc_score_interval_f32<MODE>(s0, s1, delta0, delta1, fpd, Ub[j], Lb);
        }
        tl = fmaxf(tl, Lb);
        any |= Ub[j] >= theta;
      }
    }
    // ---- rows whose upper bound reaches theta (a lower bound of the best exact score) are parked for level 2
    if (__ballot(any) != 0ull) {
      const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned long long m = __ballot(Ub[j] >= theta && Ub[j] != -INFINITY);
        const int cnt = __popcll(m);
        if (cnt != 0) {                                // (wave-uniform)
          if (np + cnt > BC_I4_PARK) refine();         // (cnt <= 64 <= BC_I4_PARK)
          if ((m >> lane) & 1ull) park[wave][np + __popcll(m & below)] = (int)(t * BC_ITILE + 4 * lane + j);
          np += cnt;
        }
      }
    }
    wave_l = fmaxf(wave_l, bc_wave_max_f32_all(tl));
  }
  if (np > 0) refine();
  __syncthreads();
  // ---- the block's list: the rows in play whose int8 upper bound reaches the block's best int8 lower bound
  const int n = s_n;
  const unsigned long long bk = s_best;
  const float best = bk != 0ull ? bc_ord_f32((int)((unsigned)(bk >> 32) ^ 0x80000000u)) : -INFINITY;
  if (n <= BC_L2_LCAP)
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      if (l_u[i] >= best) {
        const int slot = atomicAdd(&s_on, 1);
        if (slot < BC_BLK_NC) s_out[slot] = make_int2(__builtin_bit_cast(int, l_u[i]), l_row[i]);
      }
  __syncthreads();
  const int on = s_on;
  if ((int)threadIdx.x < on && threadIdx.x < BC_BLK_NC) a.blk_cand[(size_t)blockIdx.x * BC_BLK_NC + threadIdx.x] = s_out[threadIdx.x];
  bool spilled = false;
  if (on > BC_BLK_NC && n <= BC_L2_LCAP) {             // (block-uniform, rare) too many for the block's own list: the spill list
    if (threadIdx.x == 0) { s_refined_base = atomicAdd(&a.ctrl[14], on); s_on = 0; }
    __syncthreads();
    const int gb = s_refined_base;
    spilled = gb + on <= a.spill_cap;
    if (spilled)
      for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (l_u[i] >= best) a.spill[gb + atomicAdd(&s_on, 1)] = make_int2(__builtin_bit_cast(int, l_u[i]), l_row[i]);
  }
  if (threadIdx.x == 0) {
    a.blk_l[blockIdx.x] = (double)best;
    a.blk_u[blockIdx.x] = bc_ord_f32(s_umax);
    a.blk_nc[blockIdx.x] = (n > BC_L2_LCAP || (on > BC_BLK_NC && !spilled)) ? -2 : (on > BC_BLK_NC ? -1 : on);
    if (bk != 0ull && blockIdx.x < BC_I4_SEEDS) a.hot[blockIdx.x] = (long long)(unsigned)(bk & 0xffffffffull);
    if (s_refined > 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.ctrl + 12), (unsigned long long)s_refined);
  }
}
