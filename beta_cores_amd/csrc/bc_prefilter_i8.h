// K3 pre-filter, int8 mirror (BC_PREFILTER=8): one byte per element.
//
//   q[i, k] = rint( u[i, k] / scale_i ),  scale_i = max_k |u[i, k]| / 127,   u = Phi[i, :] / ||Phi[i, :]||
//   rowq[i] = (scale_i, delta_i) as two halfs, delta_i >= || q[i, :] * scale_i - u[i, :] ||_2 measured at build time
//             (scale_i is rounded UP to a half first and the row is quantised with that value, so it is exact)
//
// The sweep vector that carries the score (GIGA: the residual direction cdir, s0 = u.cdir; FW: the residual) is quantised
// per launch to 14 bits + sign, v^_k = (128 d0_k + d1_k) * vstep with int8 digits d0 in [-127, 127], d1 in [-64, 64]
// and vstep = max|v| / 16256, so that
//   u^ . v^ = scale_i * vstep * (128 * sum_k q d0 + sum_k q d1)
// is computed EXACTLY with v_dot4_i32_i8 (four products per instruction, no conversions in the loop), and
//   | u^.v^ - u.v |  <=  delta_i ||v||  +  (1 + delta_i) * sqrt(S) * vstep / 2
// bounds the distance to the fp64 kernel's dot product (Cauchy-Schwarz twice).  GIGA's second vector (the weights'
// direction y^, s1 = u.y^) only enters through 1 / sqrt(1 - s1^2), whose slope s1 / (1 - s1^2)^(3/2) is small for all
// but the rows nearly parallel to y^: it gets ONE digit (vstep = max|v| / 127) and an error bound of its own
// (delta1 ~ 2 delta0), which widens the score interval by ~|s0 s1| delta1, a tenth of delta0 -- and takes a quarter
// of the dot4 instructions out of a loop that is bound by them (profiles/r02_notes.md).  The interval formula is that
// of the fp16 / fp32 mirrors (bc_prefilter.hip) with the two deltas kept apart.
//
// Layout: tiles of 256 rows, [SP4][256] dwords, dword (g, r) = samples 4g..4g+3 of row r; lane l of the wave that
// owns the tile holds rows 4l..4l+3, so one k-group of a tile is one 1 KiB dwordx4 load.  SP4 = ceil(S/4) rounded
// up to a multiple of BC_IU (zero groups).  Bytes per row: 4*SP4 + 8.
//
// Candidates: the sweep writes nothing per row.  Per tile it stores the maximum upper bound and up to four
// (upper bound, row) pairs -- the rows whose upper bound reaches the TILE's own best lower bound, a superset of
// the rows that can reach the global one.  k_rescore reads the pairs of the tiles whose maximum reaches Lmax;
// a tile with more than four such rows hands over all of its rows.
#pragma once

#include "bc_layout.h"
#include "bc_i8_quant.h"
#define BC_ITILE 256      /* == BC_LAY_ITILE (bc_layout.h) */
#ifndef BC_IU
#define BC_IU 5          // k-groups per batch (5 KiB in flight per wave and buffer)
#endif
#define BC_IMAXG 320     // k-groups the digit table holds: S <= 1280
#define BC_IFLUSH 32     // tiles a wave parks in LDS before it writes their results out
#ifndef BC_I8_DEPTH
#define BC_I8_DEPTH 1    // batches requested ahead of the one being consumed (2: tools/build_variant.sh experiment, S >= 37 only)
#endif
typedef int bc_i4 __attribute__((ext_vector_type(4)));
typedef _Float16 bc_hq8 __attribute__((ext_vector_type(8)));   // ... of a lane's four rows

struct I8Args {
  const int* u8;            // [ptiles][sp4][256]
  const bc_hq2* rowq;       // [ptiles*256] (scale, delta) halfs; delta < 0: dead row (padding or zero norm), NaN: uncertain row
  const double* v;
  const int* qv;            // v already quantised (bc_i8_quant.h record) or nullptr: the prologue quantises v itself
  const int* skip_flag;
  const double* v_norm;     // dot mode: ||v|| from the solver state
  float* tile_u;            // [ptiles]
  float2* tile_cand;        // [ptiles][4] (upper bound, row-in-tile as float)
  int* tile_ncand;          // [ptiles]
  double* blk_l;
  float* blk_u;
  int2* blk_cand;           // [grid][BC_BLK_NC] (upper bound bits, local row): the block's rows whose upper bound reaches the BLOCK's
  int* blk_nc;              // best lower bound -- a superset of those that reach the global one; count, or -1: "scan my tiles"
  long long ptiles;
  double post_div;
  int s, sp4;
};
#define BC_BLK_NC 8

// The interval of bc_score_interval evaluated in fp32.  The int8 mirror's delta is ~1e-2, five orders of magnitude
// above fp32 rounding, so single precision costs nothing in selectivity; every fp32 evaluation error is covered
// explicitly: g = 1 - s1^2 carries an absolute error <= 2^-23, i.e. a relative one <= 2^-23 / c (c <= g), rsq and
// the products add <= 4e-7, and the slope term is inflated by another 0.2 %.  Rows with c <= 1e-3 are "uncertain".
// |s0* - s0| <= delta0, |s1* - s1| <= delta1:  |f* - f| <= delta0 / sqrt(c) + (|s0| + delta0) * a * delta1 / c^(3/2)  with
// a = |s1| + delta1, c = 1 - a^2 (mean value theorem on 1 / sqrt(1 - s^2), whose derivative is <= a / c^(3/2) on [-a, a]).
template <int MODE>
__device__ __forceinline__ void bc_score_interval_f32(float s0, float s1, float delta0, float delta1, float post_div, float& U, float& L) {
  if (MODE == 0) {
    const float a = fabsf(s1) + delta1;
    const float c = 1.f - a * a;
    if (!(s0 == s0) || !(s1 == s1) || !(c > 1e-3f)) { U = INFINITY; L = -INFINITY; return; }
    const float rc2 = __frcp_rn(c) * 1.000001f;             // >= 1/c
    const float rc = sqrtf(rc2) * 1.000001f;                // >= 1/sqrt(c)
    const float f = s0 * __frsqrt_rn(1.f - s1 * s1);
    const float e = (delta0 * rc + (fabsf(s0) + delta0) * a * delta1 * rc * rc2) * 1.003f + fabsf(f) * (6.1e-8f * rc2 + 6e-7f) + 2e-7f;
    U = f + e;
    L = f - e;
  } else {
    if (!(s0 == s0)) { U = INFINITY; L = -INFINITY; return; }
    const float ip = __frcp_rn(fabsf(post_div)) * 1.000001f;
    const float f = s0 / post_div;
    const float e = (delta0 * 1.002f + 3e-7f * fabsf(s0)) * ip + 2e-7f * fabsf(f) + 1e-30f;
    U = f + e;
    L = f - e;
  }
}

// The interval of one LIVE row from its exact integer dot products (a0 = sum q d0, a1 = sum q d1 of the score vector, a2 = sum q d
// of GIGA's second vector): shared by the sweep below and by the second level of the two-level form (bc_prefilter_i4.h), which
// evaluates the same rows from a row-major copy of the same digits -- the same arithmetic, hence the same bounds.
template <int MODE>
__device__ __forceinline__ void bc_i8_row_bounds(float sc, float dr, int a0, int a1, int a2, float fvs0, float fvs1, float fev0, float fev1,
                                                 float fvn, bool vbad, float fpd, float& U, float& L) {
  // u^.v^ : exact integers, scaled in fp32 (relative error < 3e-7, covered below)
  const float s0 = sc * fvs0 * (128.f * (float)a0 + (float)a1);
  const float s1 = (MODE == 0) ? sc * fvs1 * (float)a2 : 0.f;
  const float delta0 = (dr * fvn + (1.f + dr) * fev0) * 1.00001f + 4e-7f * fabsf(s0) + 1e-12f;
  const float delta1 = (MODE == 0) ? (dr * fvn + (1.f + dr) * fev1) * 1.00001f + 4e-7f * fabsf(s1) + 1e-12f : 0.f;
  if (vbad || dr != dr) { U = INFINITY; L = -INFINITY; }
  else bc_score_interval_f32<MODE>(s0, s1, delta0, delta1, fpd, U, L);
}

// one block per 256-row tile, thread = row
__global__ __launch_bounds__(256) void k_build_i8(const double* __restrict__ tiles, const double* __restrict__ norms,
                                                 long long n_rows, int S, int SP4, int* __restrict__ u8,
                                                 bc_hq2* __restrict__ rowq) {
  const long long t = blockIdx.x;
  const long long r = t * BC_ITILE + threadIdx.x;
  const bool live = r < n_rows && norms[r < n_rows ? r : 0] != 0.;
  const double nr = live ? norms[r] : 1.;
  // u = Phi / ||Phi|| through ONE reciprocal per row instead of two fp64 divisions per element (u, then u / scale): 3.71 ->
  // 3.23 ms at 10M rows.  Nothing downstream needs the correctly rounded quotient: delta_i is MEASURED below against the
  // values actually stored, u's relative error of 2^-52 is far inside the (1 + 1e-6) inflation of delta, and any integer
  // q with |q| <= 127 is a valid digit.  (Holding the row in registers between the two passes instead of re-reading it:
  // 4.24 ms, slower.)
  const double inr = 1. / nr;
  const double* p = tiles + bc_lay_phi_elem(r, 0, S);      // (only dereferenced for live rows)
  double mx = 0.;
  bool has_nan = false;
  if (live)
    for (int k = 0; k < S; ++k) {
      const double u = p[(size_t)k * BC_TILE] * inr;
      has_nan |= !(fabs(u) <= 1.7976931348623157e308);   // NaN or inf
      mx = fmax(mx, fabs(u));
    }
  const bool ok = live && !has_nan && mx > 0.;
  // the scale is a half, rounded UP (so |q| <= 127 still holds); the row is quantised with exactly that value
  _Float16 hs = (_Float16)0.f;
  if (ok) {
    hs = (_Float16)(float)(mx / 127.);
    if ((double)(float)hs < mx / 127.) hs = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, hs) + 1));   // next half up (positive)
    if ((float)hs < 6.2e-5f) hs = (_Float16)6.2e-5f;     // keep it a normal half (rows with tiny maxima do not exist for unit rows)
  }
  const double scale = (double)(float)hs;
  const double iscale = ok ? 1. / scale : 0.;
  double err2 = 0.;
  int* q = u8 + bc_lay_i8_word(t, 0, threadIdx.x, SP4);
  for (int g = 0; g < SP4; ++g) {
    unsigned w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 4 * g + j;
      int qi = 0;
      if (ok && k < S) {
        const double u = p[(size_t)k * BC_TILE] * inr;
        qi = (int)rint(u * iscale);
        qi = qi > 127 ? 127 : (qi < -127 ? -127 : qi);
        const double d = (double)qi * scale - u;
        err2 = fma(d, d, err2);
      }
      w |= ((unsigned)qi & 0xffu) << (8 * j);
    }
    q[(size_t)g * BC_ITILE] = (int)w;
  }
  bc_hq2 rq;
  if (!live) {
    rq = (bc_hq2){(_Float16)0.f, (_Float16)-1.f};
  } else if (!ok) {
    rq = (bc_hq2){(_Float16)0.f, __builtin_bit_cast(_Float16, (unsigned short)0x7e00)};   // NaN: kept, with [-inf, inf]
  } else {
    // delta as a half rounded up (with room for the fp64 evaluation above)
    const double dd = sqrt(err2) * (1. + 1e-6) + 1e-12;
    _Float16 hd = (_Float16)(float)dd;
    if ((double)(float)hd < dd) hd = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, hd) + 1));
    rq = (bc_hq2){hs, hd};
  }
  rowq[r - 0] = rq;
}

// S <= SMAX: ONE pass over Phi -- thread = row, the row's S values are loaded once (all of them in flight: no per-lane
// condition on the loads, every row of a tile is readable memory) and stay in registers between the maximum and the
// quantisation: 1.9 ms for 10M rows.  The two-pass kernel above moves 17 GB there (the second pass misses the L2): 3.2 ms.
template <int SMAX>
__global__ __launch_bounds__(256) void k_build_i8_r(const double* __restrict__ tiles, const double* __restrict__ norms,
                                                   long long n_rows, int S, int SP4, int* __restrict__ u8,
                                                   bc_hq2* __restrict__ rowq) {
  const long long t = blockIdx.x;
  const long long r = t * BC_ITILE + threadIdx.x;
  const bool live = r < n_rows && norms[r < n_rows ? r : 0] != 0.;
  const double inr = live ? 1. / norms[r] : 0.;       // (one reciprocal per row: see k_build_i8)
  // rows past the end have no tile behind them when the number of 128-row tiles is odd: they read row 0 instead (and are dead)
  const long long rr = bc_lay_i8_src_row(r, n_rows);
  const double* p = tiles + bc_lay_phi_elem(rr, 0, S);
  double ur[SMAX];
#pragma unroll
  for (int k = 0; k < SMAX; ++k) ur[k] = (k < S && n_rows > 0) ? p[(size_t)k * BC_TILE] : 0.;      // (uniform condition)
  double mx = 0.;
  bool has_nan = false;
#pragma unroll
  for (int k = 0; k < SMAX; ++k) {
    ur[k] *= inr;
    has_nan |= !(fabs(ur[k]) <= 1.7976931348623157e308);   // NaN or inf (a dead row's garbage too: it is dead either way)
    mx = fmax(mx, fabs(ur[k]));
  }
  const bool ok = live && !has_nan && mx > 0.;
  _Float16 hs = (_Float16)0.f;
  if (ok) {
    hs = (_Float16)(float)(mx / 127.);
    if ((double)(float)hs < mx / 127.) hs = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, hs) + 1));   // next half up (positive)
    if ((float)hs < 6.2e-5f) hs = (_Float16)6.2e-5f;
  }
  const double scale = (double)(float)hs;
  const double iscale = ok ? 1. / scale : 0.;
  double err2 = 0.;
  int* q = u8 + bc_lay_i8_word(t, 0, threadIdx.x, SP4);
#pragma unroll
  for (int g = 0; g < SMAX / 4; ++g) {
    if (g < SP4) {                                        // (uniform)
      unsigned w = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * g + j;
        const double u = ur[k];
        int qi = (int)rint(u * iscale);
        qi = qi > 127 ? 127 : (qi < -127 ? -127 : qi);
        const double d = (double)qi * scale - u;
        const bool use = ok && k < S;
        err2 = use ? fma(d, d, err2) : err2;
        w |= use ? ((unsigned)qi & 0xffu) << (8 * j) : 0u;
      }
      q[(size_t)g * BC_ITILE] = (int)w;
    }
  }
  for (int g = SMAX / 4; g < SP4; ++g) q[(size_t)g * BC_ITILE] = 0;          // (zero groups of the padding)
  bc_hq2 rq;
  if (!live) {
    rq = (bc_hq2){(_Float16)0.f, (_Float16)-1.f};
  } else if (!ok) {
    rq = (bc_hq2){(_Float16)0.f, __builtin_bit_cast(_Float16, (unsigned short)0x7e00)};   // NaN: kept, with [-inf, inf]
  } else {
    const double dd = sqrt(err2) * (1. + 1e-6) + 1e-12;
    _Float16 hd = (_Float16)(float)dd;
    if ((double)(float)hd < dd) hd = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, hd) + 1));
    rq = (bc_hq2){hs, hd};
  }
  rowq[r] = rq;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_sweep_i8(I8Args a) {
  constexpr int NV = (MODE == 0) ? 3 : 2;           // (vector, digit) combinations: v0 d0, v0 d1 [, v1 single digit]
  __shared__ __attribute__((aligned(16))) int dig[BC_IMAXG][4];   // packed digits of k-group g: [v0 d0, v0 d1, v1 d, 0]
  __shared__ double vmx[2][4];
  __shared__ float sl[4];
  __shared__ float su[4];
  // per-tile results are parked here and written out in bursts of BC_IFLUSH tiles per wave (for 10M rows: once, after the
  // wave's last tile): 39k x 3 small scattered stores interleaved with the read stream cost the HBM channels their
  // read/write turn-arounds all through the sweep
  __shared__ float s_tu[4][BC_IFLUSH];
  __shared__ int s_nc[4][BC_IFLUSH];
  __shared__ __attribute__((aligned(16))) float2 s_cd[4][BC_IFLUSH][4];
  // round 5: the block's own short list of candidates (see I8Args::blk_cand), so that the rescoring stage finds the rows in
  // play in its FIRST round of loads instead of walking blocks -> tiles -> pairs (a dependent round trip, ~2 us of every step)
  __shared__ int bl_n, bl_scan;
  __shared__ int2 bl_list[BC_BLK_NC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float best_l = -INFINITY;
  float umax = -INFINITY;
  int ti_last = 0;                                   // tiles still parked in LDS when the wave has walked its last tile
  long long t_first = 0, t_step = 0;
  bool flushed_early = false;
  if (threadIdx.x == 0) { bl_n = 0; bl_scan = 0; }
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;
  if (!skip) {
    const int S = a.s, SP4 = a.sp4;
    constexpr int U = BC_IU;
    // the first tile's loads do not depend on the prologue: put them in flight before it
    long long t = (long long)blockIdx.x * 4 + wave;
    bc_i4 x[U], y[U];
    bc_hq8 rq = {(_Float16)0.f, (_Float16)-1.f, (_Float16)0.f, (_Float16)-1.f, (_Float16)0.f, (_Float16)-1.f, (_Float16)0.f, (_Float16)-1.f};
#if BC_I8_DEPTH == 2
    bc_i4 z[U];                                        // two batches ahead (needs SP4 >= 2 U: the launcher checks)
    bc_hq8 rq_n = rq;
#endif
    if (t < a.ptiles) {
      const bc_i4* __restrict__ p0 = reinterpret_cast<const bc_i4*>(a.u8 + (size_t)t * SP4 * BC_ITILE) + lane;
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p0 + (size_t)u * 64);
      rq = reinterpret_cast<const bc_hq8*>(a.rowq + t * BC_ITILE)[lane];
#if BC_I8_DEPTH == 2
#pragma unroll
      for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(p0 + (size_t)(U + u) * 64);
#endif
    }
    // ---- prologue.  With a.qv (the step kernel that produced v left its digits behind, bc_i8_quant.h): copy 1.3 KB into
    // LDS, one barrier.  Without (standalone argmax sweeps): quantise v here, every block the same tiny job.
    float fvn, fev0, fev1, fvs0, fvs1;
    bool vbad;
    if (a.qv != nullptr) {
      for (int g = threadIdx.x; g < SP4; g += blockDim.x)
        *reinterpret_cast<bc_i4*>(&dig[g][0]) = reinterpret_cast<const bc_i4*>(a.qv)[g];
      const float* hf = reinterpret_cast<const float*>(a.qv + 4 * SP4);
      fvs0 = hf[0]; fvs1 = hf[1]; fev0 = hf[2]; fev1 = hf[3]; fvn = hf[4];
      vbad = hf[5] != 0.f;
      __syncthreads();
    } else {
      double m0 = 0., m1 = 0.;
      for (int k = threadIdx.x; k < S; k += blockDim.x) {
        if (MODE == 0) {
          m0 = bc_i8q_absmax(m0, a.v[2 * k]);
          m1 = bc_i8q_absmax(m1, a.v[2 * k + 1]);
        } else {
          m0 = bc_i8q_absmax(m0, a.v[k]);
        }
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        m0 = fmax(m0, __shfl_down(m0, d, BC_WAVE));
        m1 = fmax(m1, __shfl_down(m1, d, BC_WAVE));
      }
      if (lane == 0) { vmx[0][wave] = m0; vmx[1][wave] = m1; }
      __syncthreads();
      const bc_i8q_scalars q = bc_i8q_steps(fmax(fmax(vmx[0][0], vmx[0][1]), fmax(vmx[0][2], vmx[0][3])),
                                            fmax(fmax(vmx[1][0], vmx[1][1]), fmax(vmx[1][2], vmx[1][3])));
      for (int g = threadIdx.x; g < SP4; g += blockDim.x) {
        unsigned w[4];
        bc_i8q_group<MODE>(a.v, S, g, q, w);
        dig[g][0] = (int)w[0]; dig[g][1] = (int)w[1]; dig[g][2] = (int)w[2]; dig[g][3] = (int)w[3];
      }
      __syncthreads();
      const bc_i8q_hdr h = bc_i8q_header(q, S, (MODE == 0) ? 1. : *a.v_norm);
      fvn = h.fvn; fev0 = h.fev0; fev1 = h.fev1; fvs0 = h.fvs0; fvs1 = h.fvs1;
      vbad = h.vbad;
    }
    const float fpd = (float)a.post_div;

    // Continuous pipeline: while the last batch of a tile is consumed and its rows are evaluated, the first batch
    // (and the row constants) of the wave's NEXT tile are already in flight.
    const long long tstride = (long long)gridDim.x * 4;
    int ti = 0;                                      // tiles parked since the last flush
    long long t_park = t;                            // ... the first of them
    auto flush = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < ti) {
        const long long tt = t_park + (long long)lane * tstride;
        a.tile_u[tt] = s_tu[wave][lane];
        a.tile_ncand[tt] = s_nc[wave][lane];
        const float4 c01 = *reinterpret_cast<const float4*>(&s_cd[wave][lane][0]);
        const float4 c23 = *reinterpret_cast<const float4*>(&s_cd[wave][lane][2]);
        float4* dst = reinterpret_cast<float4*>(a.tile_cand + tt * 4);
        dst[0] = c01;
        dst[1] = c23;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      ti = 0;
    };
    for (; t < a.ptiles; t += tstride) {
      const bc_i4* __restrict__ p = reinterpret_cast<const bc_i4*>(a.u8 + (size_t)t * SP4 * BC_ITILE) + lane;
      int acc[4][NV];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < NV; ++c) acc[j][c] = 0;
      const bc_hq8 cq = rq;                            // this tile's (scale, delta) x 4 rows
      for (int g0 = 0; g0 < SP4; g0 += U) {
#if BC_I8_DEPTH == 2
        // two batches ahead: batch g0 / U + 2 of this tile, or batch 0 / 1 of the wave's next tile (with its row constants)
        const int nb = g0 + 2 * U;
        if (nb < SP4) {
#pragma unroll
          for (int u = 0; u < U; ++u) z[u] = __builtin_nontemporal_load(p + (size_t)(nb + u) * 64);
        } else if (t + tstride < a.ptiles) {
          const long long tn = t + tstride;
          const bc_i4* __restrict__ pn = reinterpret_cast<const bc_i4*>(a.u8 + (size_t)tn * SP4 * BC_ITILE) + lane;
#pragma unroll
          for (int u = 0; u < U; ++u) z[u] = __builtin_nontemporal_load(pn + (size_t)(nb - SP4 + u) * 64);
          if (nb == SP4) rq_n = reinterpret_cast<const bc_hq8*>(a.rowq + tn * BC_ITILE)[lane];
        }
#else
        const bool more = g0 + U < SP4;
        if (more) {
#pragma unroll
          for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(p + (size_t)(g0 + U + u) * 64);
        } else if (t + tstride < a.ptiles) {
          const long long tn = t + tstride;
          const bc_i4* __restrict__ pn = reinterpret_cast<const bc_i4*>(a.u8 + (size_t)tn * SP4 * BC_ITILE) + lane;
#pragma unroll
          for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(pn + (size_t)u * 64);
          rq = reinterpret_cast<const bc_hq8*>(a.rowq + tn * BC_ITILE)[lane];
        }
#endif
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bc_i4 dg = *reinterpret_cast<const bc_i4*>(&dig[g0 + u][0]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < NV; ++c) acc[j][c] = __builtin_amdgcn_sdot4(x[u][j], dg[c], acc[j][c], false);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = y[u];        // (after the last batch: the next tile's first batch, if any)
#if BC_I8_DEPTH == 2
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = z[u];
#endif
      }
#if BC_I8_DEPTH == 2
      rq = rq_n;
#endif
      // ---- per-row intervals (4 rows per lane), fp32 with explicit slack (bc_score_interval_f32)
      const float sc[4] = {(float)cq[0], (float)cq[2], (float)cq[4], (float)cq[6]};
      const float dl[4] = {(float)cq[1], (float)cq[3], (float)cq[5], (float)cq[7]};
      float Ub[4], Lb[4];
      float tl = -INFINITY, tmax = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Ub[j] = -INFINITY;
        Lb[j] = -INFINITY;
        if (!(dl[j] < 0.f)) {                           // live (a NaN delta counts as live and yields [-inf, inf])
          bc_i8_row_bounds<MODE>(sc[j], dl[j], acc[j][0], acc[j][1], acc[j][NV - 1], fvs0, fvs1, fev0, fev1, fvn, vbad, fpd, Ub[j], Lb[j]);
          tl = fmaxf(tl, Lb[j]);
          tmax = fmaxf(tmax, Ub[j]);
        }
      }
      tl = bc_wave_max_f32_all(tl);                    // (DPP row rotations + v_readlane: the ds_bpermute tree of
      tmax = bc_wave_max_f32_all(tmax);                //  __shfl_xor was a dozen dependent LDS round trips per tile)
      // local candidates: rows whose upper bound reaches the tile's best lower bound (ballot compaction, <= 4 kept)
      int base = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool c = Ub[j] >= tl && Ub[j] != -INFINITY;
        const unsigned long long m = __ballot(c);
        if (c) {
          const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
          if (slot < 4) s_cd[wave][ti][slot] = make_float2(Ub[j], (float)(4 * lane + j));
        }
        base += __popcll(m);
      }
      if (lane == 0) {
        s_tu[wave][ti] = tmax;
        s_nc[wave][ti] = base;
      }
      if (ti == 0) t_park = t;
      if (++ti == BC_IFLUSH) {
        flush();
        if (t + tstride < a.ptiles) flushed_early = true;      // (a flush that happens to be the wave's last leaves all in LDS)
        else ti_last = BC_IFLUSH;
      }
      best_l = fmaxf(best_l, tl);
      umax = fmaxf(umax, tmax);
    }
    if (ti > 0) {
      ti_last = ti;
      flush();
    }
    t_first = t_park;
    t_step = tstride;
  }
  if (lane == 0) { sl[wave] = best_l; su[wave] = umax; }
  __syncthreads();
  const float blk_best_l = fmaxf(fmaxf(sl[0], sl[1]), fmaxf(sl[2], sl[3]));
  if (threadIdx.x == 0) {
    a.blk_l[blockIdx.x] = (double)blk_best_l;
    a.blk_u[blockIdx.x] = fmaxf(fmaxf(su[0], su[1]), fmaxf(su[2], su[3]));
  }
  if (a.blk_nc == nullptr) return;
  // ---- the block's candidates: parked pairs whose upper bound reaches the block's best lower bound.  The parked results of
  // the wave's last <= BC_IFLUSH tiles are still in LDS (the flush only copied them out); a wave that flushed before that, or a
  // tile with more than four pairs, makes the rescoring stage walk this block's tiles as before (blk_nc = -1).
  if (flushed_early && lane == 0) bl_scan = 1;
  if (!flushed_early && lane < ti_last) {
    const int nc = s_nc[wave][lane];
    const long long tt = t_first + (long long)lane * t_step;
    if (nc > 4) {
      if (s_tu[wave][lane] >= blk_best_l) bl_scan = 1;       // (only if the tile is in play at all: 1-2 % of all tiles hold > 4 pairs)
    } else {
      for (int i = 0; i < nc; ++i) {
        const float2 pr = s_cd[wave][lane][i];
        if (pr.x >= blk_best_l) {
          const int slot = atomicAdd(&bl_n, 1);
          if (slot < BC_BLK_NC) bl_list[slot] = make_int2(__float_as_int(pr.x), (int)(tt * BC_ITILE + (long long)pr.y));
        }
      }
    }
  }
  __syncthreads();
  const int n = bl_n;
  if ((int)threadIdx.x < n && threadIdx.x < BC_BLK_NC) a.blk_cand[(size_t)blockIdx.x * BC_BLK_NC + threadIdx.x] = bl_list[threadIdx.x];
  if (threadIdx.x == 0) a.blk_nc[blockIdx.x] = (bl_scan != 0 || n > BC_BLK_NC) ? -1 : n;
}
