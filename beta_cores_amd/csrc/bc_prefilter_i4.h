// K3 pre-filter, TWO-LEVEL form (BC_PREFILTER=4): the per-iteration sweep streams HALF a byte per element.
//
//   level 1  k_sweep_i4 streams a 4-bit mirror of the normalised rows (q in [-7, 7], per-row scale chosen at build time to
//            minimise the MEASURED error delta4_i ~ 0.1) against the sweep vectors in two 4-bit digits each (bc_i4_quant.h),
//            v_dot8_i32_i4: eight exact products per instruction.  Same interval arithmetic as the int8 sweep
//            (bc_score_interval_f32), wider deltas.  A row is passed on when its upper bound reaches
//            theta = max(theta0, the best lower bound the wave has seen so far); both are lower bounds of the best exact
//            score, so the row the fp64 sweep returns -- and every row tied with it -- is always passed on.  theta0 comes from
//            SEEDS: the strongest row of each of the first 224 sweep blocks of the last two-level step (rows spread over the
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
#define BC_I4_SEEDS 256      // ... behind one slot per sweep block (its strongest row; the last BC_I4_HOT slots are not read): hot[BC_I4_SEEDS + BC_I4_HOT]
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

// bc_score_interval_f32 (bc_prefilter_i8.h) with the hardware's approximate reciprocal / square roots (v_rcp_f32, v_sqrt_f32,
// v_rsq_f32: 1 ulp) instead of the correctly rounded sequences (~12 instructions each: the 4-bit sweep evaluates a row per 13
// loaded dwords, and its epilogue cost as much as its dot products).  Every use is an UPPER bound inflated by 1.000001 (an ulp
// is 1.2e-7) or enters f, whose evaluation error term is raised from 6e-7 to 1e-6 for the rsq's ulp.  ifpd >= 1 / |post_div|.
template <int MODE>
__device__ __forceinline__ void bc_score_interval_f32_hw(float s0, float s1, float delta0, float delta1, float post_div, float ifpd, float& U, float& L) {
  if (MODE == 0) {
    const float a = fabsf(s1) + delta1;
    const float c = 1.f - a * a;
    const bool bad = !(s0 == s0) || !(s1 == s1) || !(c > 1e-3f);
    const float cc = bad ? 1.f : c;
    const float rc2 = __builtin_amdgcn_rcpf(cc) * 1.000001f;              // >= 1/c
    const float rc = __builtin_amdgcn_sqrtf(rc2) * 1.000001f;            // >= 1/sqrt(c)
    const float g = 1.f - s1 * s1;
    const float f = s0 * __builtin_amdgcn_rsqf(bad ? 1.f : g);
    const float e = (delta0 * rc + (fabsf(s0) + delta0) * a * delta1 * rc * rc2) * 1.003f + fabsf(f) * (6.1e-8f * rc2 + 1e-6f) + 2e-7f;
    U = bad ? INFINITY : f + e;
    L = bad ? -INFINITY : f - e;
  } else {
    const bool bad = !(s0 == s0);
    const float f = s0 * __builtin_amdgcn_rcpf(post_div);                // within 2 ulp of s0 / post_div: the 1e-6 |f| below
    const float e = (delta0 * 1.002f + 3e-7f * fabsf(s0)) * ifpd + 1e-6f * fabsf(f) + 1e-30f;
    U = bad ? INFINITY : f + e;
    L = bad ? -INFINITY : f - e;
  }
}

// ------------------------------------------------------------------ builders
// quantise one unit row's element with scale code c (scale = c / 1024)
__device__ __forceinline__ int bc_i4_digit(double u, double iscale) {
  int q = (int)rint(u * iscale);
  return q > 7 ? 7 : (q < -7 ? -7 : q);
}

// The 4-bit mirror: one block per 256-row tile, thread = row.
//   Phase A, the scale: a heuristic search over eight candidates (the smallest code that never clips, times 1 .. 0.3) for the
//     one with the least squared error -- evaluated on the row's INT8 digits (25 coalesced dwords per row instead of a hundred
//     doubles; their 0.6 % error is irrelevant for a heuristic: ANY scale is valid, the bound only needs the error measured).
//   Phase B, digits and delta: ONE streaming pass over the fp64 row, u = Phi / ||Phi|| as in k_build_i8: q = clamp(rint(u /
//     scale)), delta^2 += (q scale - u)^2 in fp64; delta rounded UP to its 8-bit code.
// (First version: the row held in 208 registers, the search on fp64 -> fp32 conversions: 334 VGPRs, one wave per SIMD, 13 ms at
// N = 10M against 1.9 ms for the int8 mirror; this one streams.)
__global__ __launch_bounds__(256) void k_build_i4(const double* __restrict__ tiles, const double* __restrict__ norms, long long n_rows,
                                                 int S, int SP8, const int* __restrict__ u8, const bc_hq2* __restrict__ rowq, int SP4, int G4,
                                                 int* __restrict__ u4, unsigned short* __restrict__ rowq4) {
  const long long t = blockIdx.x;
  const long long r = t * BC_ITILE + threadIdx.x;
  const bc_hq2 rq8 = rowq[r];
  const float sc8 = (float)rq8[0], d8 = (float)rq8[1];
  const bool dead = d8 < 0.f, unc = d8 != d8;
  int* q = u4 + bc_lay_i4_word(t, 0, threadIdx.x, SP8);
  if (dead || unc) {                                   // (per-lane branch: such rows are rare)
    for (int g = 0; g < SP8; ++g) q[(size_t)g * BC_ITILE] = 0;
    rowq4[r] = (unsigned short)((dead ? BC_I4_DEAD : BC_I4_UNCERTAIN) << 8);
    return;
  }
  // ---- phase A
  const int* src = u8 + bc_lay_i8_word(t, 0, threadIdx.x, SP4);
  int cb = (int)ceilf(127.f * sc8 * (1024.f / 7.f));    // |u| <= 127 scale8
  cb = cb < 1 ? 1 : (cb > 253 ? 253 : cb);
  int bestc = cb;
  float beste = INFINITY;
#pragma unroll 1
  for (int f = 0; f < 8; ++f) {
    int c = (int)((float)cb * (1.f - 0.1f * (float)f) + 0.5f);       // cb x {1, .9, .8, .7, .6, .5, .4, .3}
    c = c < 1 ? 1 : c;
    const float sc = (float)c * (1.f / 1024.f);
    const float ratio = sc8 * (1024.f / (float)c);       // int8 digit -> units of the 4-bit step
    float e0 = 0.f, e1 = 0.f;
    for (int g = 0; g < G4; ++g) {
      const int w = src[(size_t)g * BC_ITILE];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = (float)((w << (24 - 8 * j)) >> 24) * ratio;     // the element in 4-bit steps
        float qq = rintf(x);
        qq = qq > 7.f ? 7.f : (qq < -7.f ? -7.f : qq);
        const float d = qq - x;
        if (j & 1) e1 = fmaf(d, d, e1); else e0 = fmaf(d, d, e0);
      }
    }
    const float e = (e0 + e1) * sc * sc;
    if (e < beste) { beste = e; bestc = c; }
  }
  // ---- phase B
  const double scale = (double)bestc * (1. / 1024.);
  const double iscale = 1024. / (double)bestc;
  const double inr = 1. / norms[r];                     // (live: r < n_rows and a non-zero norm; one reciprocal per row, see k_build_i8)
  const double* p = tiles + bc_lay_phi_elem(r, 0, S);
  double err2 = 0.;
  for (int g = 0; g < SP8; ++g) {
    unsigned w = 0;
    double u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * g + j;
      u[j] = k < S ? p[(size_t)k * BC_TILE] * inr : 0.;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int qi = bc_i4_digit(u[j], iscale);          // (padding: u = 0 -> digit 0, no error)
      const double d = (double)qi * scale - u[j];
      err2 = fma(d, d, err2);
      w |= ((unsigned)qi & 0xfu) << (4 * j);
    }
    q[(size_t)g * BC_ITILE] = (int)w;
  }
  const double dd = sqrt(err2) * (1. + 1e-6) + 1e-12;
  const int dc = (int)ceil(dd * 512.);                   // delta = dc / 512 >= dd
  // (a NaN / inf element makes the int8 row "uncertain" already; should the measured error not fit the code, the row is uncertain too)
  rowq4[r] = (dc <= 250) ? (unsigned short)(bestc | (dc << 8)) : (unsigned short)(BC_I4_UNCERTAIN << 8);
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

// diagnostic build (-DBC_I4_STAMPS, tools/build_variant.sh): s_memtime of every block's thread 0 at the phase boundaries
#ifdef BC_I4_STAMPS
__device__ unsigned long long g_i4_stamps[1024][8];
#define I4STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_i4_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int bc_debug_i4_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_i4_stamps), sizeof(g_i4_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define I4STAMP(i) do { } while (0)
#endif

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
  I4STAMP(0);
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;
  if (skip) return;
  const int SP8 = a.sp8;
  // the first tile's loads and the seed rows' numbers do not depend on the prologue: in flight before it
  long long t = (long long)blockIdx.x * 4 + wave;
  const long long tstride = (long long)gridDim.x * 4;
  // the seed rows' numbers first
  // one seed per thread: the strongest rows of the first 224 sweep blocks and the ring's 32 (a second evaluation by wave 0 for
  // the ring kept the whole block at its second barrier: every wave now does exactly one)
  const long long hot_row = a.hot[threadIdx.x < BC_I4_SEEDS - BC_I4_HOT ? threadIdx.x : threadIdx.x + BC_I4_HOT];      // (BC_I4_SEEDS == blockDim.x)
  bc_i4 x[U], y[U];
  uint2 rq = make_uint2(0xffffffffu, 0xffffffffu), rq_next = rq;    // (dead)
  // the digit records are requested BEFORE the tiles (loads return in order): the barrier that publishes them in LDS then
  // waits for 2 KB, not for the wave's first 26 KB -- it is an LDS-only barrier (s_waitcnt lgkmcnt(0) + s_barrier; a
  // __syncthreads() carries a vmcnt(0)), and the seeds' records are requested ~10 us earlier, right behind the tiles
  bc_i4 d4v = {0, 0, 0, 0}, d8v = {0, 0, 0, 0};
  float h8v = 0.f;
  if ((int)threadIdx.x < SP8) d4v = reinterpret_cast<const bc_i4*>(a.qv4)[threadIdx.x];             // (SP8 <= 48, sp4 <= 72 < blockDim)
  if ((int)threadIdx.x < a.sp4) d8v = reinterpret_cast<const bc_i4*>(a.qv8)[threadIdx.x];
  if (threadIdx.x < 8) h8v = reinterpret_cast<const float*>(a.qv8 + 4 * a.sp4)[threadIdx.x];
  const float* hf = reinterpret_cast<const float*>(a.qv4 + 4 * SP8);
  const float fvs0 = hf[0], fvs1 = hf[1], fev0 = hf[2], fev1 = hf[3], fvn = hf[4];
  const bool vbad = hf[5] != 0.f;
  // (unconditional, clamped to a valid tile: inside an `if` the compiler can no longer count the loads behind the digit
  // records and waits for everything before it touches them)
  {
    const long long ta = t < a.ptiles ? t : 0;
    const long long t2r = (U >= SP8) ? t + tstride : t;
    const int g2 = (U >= SP8) ? 0 : U;
    const long long tb = t2r < a.ptiles ? t2r : 0;
    const uint2 rqa = reinterpret_cast<const uint2*>(a.rowq4 + ta * BC_ITILE)[lane];
    const uint2 rqb = reinterpret_cast<const uint2*>(a.rowq4 + tb * BC_ITILE)[lane];
    const bc_i4* __restrict__ p0 = reinterpret_cast<const bc_i4*>(a.u4 + (size_t)ta * SP8 * BC_ITILE) + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p0 + (size_t)u * 64);
    // ... and the second batch: two batches per wave cover the prologue
    const bc_i4* __restrict__ p1 = reinterpret_cast<const bc_i4*>(a.u4 + (size_t)tb * SP8 * BC_ITILE) + (size_t)g2 * 64 + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(p1 + (size_t)u * 64);
    if (t < a.ptiles) rq = rqa;
    if (g2 == 0 && t2r < a.ptiles) rq_next = rqb;
  }
  // ---- prologue: both digit records into LDS
  if ((int)threadIdx.x < SP8) *reinterpret_cast<bc_i4*>(&dig4[threadIdx.x][0]) = d4v;
  if ((int)threadIdx.x < a.sp4) *reinterpret_cast<bc_i4*>(&dig8[threadIdx.x][0]) = d8v;
  if (threadIdx.x < 8) hdr8[threadIdx.x] = h8v;
  if (threadIdx.x == 0) { s_n = 0; s_best = 0ull; s_umax = bc_f32_ord(-INFINITY); s_on = 0; s_refined = 0; }
  const float fpd = (float)a.post_div;
  const float ifpd = __frcp_rn(fabsf(fpd)) * 1.000001f;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // LDS-only barrier: the tiles stay in flight
  I4STAMP(6);
  // ---- seeds: the int8 lower bound of a row under the NEW vectors is a lower bound of the best exact score
  {
    float seed = -INFINITY;
    if (hot_row >= 0) {
      float Us, Ls;
      bc_r8_interval<MODE>(a.r8 + (size_t)hot_row * a.rb, dig8, hdr8, a.g4, a.rb, fpd, Us, Ls);
      if (Ls == Ls) seed = Ls;
    }
    seed = bc_wave_max_f32_all(seed);
    if (lane == 0) sl[wave] = seed;
  }
  __syncthreads();
  const float theta0 = fmaxf(fmaxf(sl[0], sl[1]), fmaxf(sl[2], sl[3]));
  I4STAMP(1);

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
  // ---- per-row intervals of a finished tile (4 rows per lane), branch-free: dead / uncertain rows are patched in at the end
  auto epilogue = [&](long long tt, const uint2 cq, const int (&acc)[4][NV]) __attribute__((always_inline)) {
    const unsigned codes[4] = {cq.x & 0xffffu, cq.x >> 16, cq.y & 0xffffu, cq.y >> 16};
    const float theta = fmaxf(theta0, wave_l);
    float Ub[4];
    float tl = -INFINITY;
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned dc = codes[j] >> 8;
      const float sc = (float)(codes[j] & 0xffu) * (1.f / 1024.f), dr = (float)dc * (1.f / 512.f);      // exact
      // u^.v^ : exact integers (|16 A0 + A1| < 2^24), scaled in fp32 (relative error < 2e-7, covered by the 4e-7 terms)
      const float s0 = sc * fvs0 * (16.f * (float)acc[j][0] + (float)acc[j][1]);
      const float s1 = (MODE == 0) ? sc * fvs1 * (16.f * (float)acc[j][NV - 2] + (float)acc[j][NV - 1]) : 0.f;
      const float delta0 = (dr * fvn + (1.f + dr) * fev0) * 1.00001f + 4e-7f * fabsf(s0) + 1e-12f;
      const float delta1 = (MODE == 0) ? (dr * fvn + (1.f + dr) * fev1) * 1.00001f + 4e-7f * fabsf(s1) + 1e-12f : 0.f;
      float Uj, Lj;
      bc_score_interval_f32_hw<MODE>(s0, s1, delta0, delta1, fpd, ifpd, Uj, Lj);
      const bool dead = dc == BC_I4_DEAD, unc = vbad || dc == BC_I4_UNCERTAIN;
      Uj = dead ? -INFINITY : (unc ? INFINITY : Uj);
      Lj = (dead || unc) ? -INFINITY : Lj;
      Ub[j] = Uj;
      tl = fmaxf(tl, Lj);
      any |= Uj >= theta;                              // (-inf >= theta only when theta is -inf: filtered below)
    }
    // rows whose upper bound reaches theta (a lower bound of the best exact score) are parked for level 2
    if (__ballot(any) != 0ull) {
      const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned long long m = __ballot(Ub[j] >= theta && Ub[j] != -INFINITY);
        const int cnt = __popcll(m);
        if (cnt != 0) {                                // (wave-uniform)
          if (np + cnt > BC_I4_PARK) refine();         // (cnt <= 64 <= BC_I4_PARK)
          // (Tried: touching the row's int8 record here, so that level 2 finds it in the L2 -- loads return in order, so the
          // cold touch holds up the next batch: stream phase 131k -> 178k cycles.)
          if ((m >> lane) & 1ull) park[wave][np + __popcll(m & below)] = (int)(tt * BC_ITILE + 4 * lane + j);
          np += cnt;
        }
      }
    }
    wave_l = fmaxf(wave_l, bc_wave_max_f32_all(tl));
  };
  // ---- the stream, as a sequence of batches (tile, first k-group) consumed from two register buffers in turn: while one is
  // consumed (and, at a tile's end, its rows are evaluated) the other one's loads are in flight; nothing is copied
  auto issue = [&](bc_i4 (&buf)[U], long long tt, int g0) __attribute__((always_inline)) {
    const bc_i4* __restrict__ p = reinterpret_cast<const bc_i4*>(a.u4 + (size_t)tt * SP8 * BC_ITILE) + (size_t)g0 * 64 + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
  };
  int acc[4][NV];
  auto zero = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < NV; ++c) acc[j][c] = 0;
  };
  auto dots = [&](const bc_i4 (&buf)[U], int g0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bc_i4 dg = *reinterpret_cast<const bc_i4*>(&dig4[g0 + u][0]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < NV; ++c) acc[j][c] = __builtin_amdgcn_sdot8(buf[u][j], dg[c], acc[j][c], false);
    }
  };
  zero();
  long long tc = t;                                    // the batch being consumed: tile, first k-group
  int gc = 0;
  bool primed = true;                                  // (the second batch is already in flight)
  while (tc < a.ptiles) {
    // --- half 1: consume x, y in flight
    long long tn = tc;
    int gn = gc + U;
    if (gn >= SP8) { gn = 0; tn = tc + tstride; }
    if (tn < a.ptiles && !primed) {
      if (gn == 0) rq_next = reinterpret_cast<const uint2*>(a.rowq4 + tn * BC_ITILE)[lane];
      issue(y, tn, gn);
    }
    primed = false;
    dots(x, gc);
    if (gc + U >= SP8) { epilogue(tc, rq, acc); zero(); rq = rq_next; }
    tc = tn; gc = gn;
    if (tc >= a.ptiles) break;
    // --- half 2: consume y, x in flight
    tn = tc;
    gn = gc + U;
    if (gn >= SP8) { gn = 0; tn = tc + tstride; }
    if (tn < a.ptiles) {
      if (gn == 0) rq_next = reinterpret_cast<const uint2*>(a.rowq4 + tn * BC_ITILE)[lane];
      issue(x, tn, gn);
    }
    dots(y, gc);
    if (gc + U >= SP8) { epilogue(tc, rq, acc); zero(); rq = rq_next; }
    tc = tn; gc = gn;
  }
  I4STAMP(2);
  if (np > 0) refine();
  I4STAMP(3);
  __syncthreads();
  I4STAMP(4);
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
  I4STAMP(5);
}
