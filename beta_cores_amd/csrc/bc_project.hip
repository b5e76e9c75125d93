// K1: per-datapoint (beta-)log-likelihood projection, fused with row-centring (projector.py:26,55),
// row norms (giga.py:10) and per-tile column sums (K2, hilbert.py:17 / bcores.py:77).
//
//   Phi[i, s] = f(z_i, theta_s) - mean_s f(z_i, theta_.)
//
// The contraction P[s, i] = sum_d Theta[s, d] * Z[i, d] is a dense (S x D)(D x 128) product
// per 128-row tile; it runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) with Theta as the
// A operand and the Z tile as the B operand, so that the accumulator of a lane holds, for ONE
// data row, samples {g + 4*reg + 16*tile}: the model formula, the row mean and the row norm are
// then computed in registers with two cross-lane adds, and the tile is stored straight into
// the [S][128] layout the K3 sweep streams.  Z and Theta are staged through LDS in D-chunks
// (coalesced global loads, register prefetch of the next chunk while the MFMAs of the current
// one run).  Algorithmic traffic per tile: 8*128*Dz B read + 8*128*S B written.
//
// Formula sources (expression order kept, -ffp-contract=off):
//   model_linreg.py:4-10 / model_neurlinr.py:90-97,102-110 / model_lr.py:72-86 / gaussian.py:7-15,34-62
#include "bc_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double bc_d2v __attribute__((ext_vector_type(2)));
typedef unsigned int bc_u4v __attribute__((ext_vector_type(4)));
typedef unsigned int bc_u2v __attribute__((ext_vector_type(2)));
// Z is read once and Phi written once per projection; the non-temporal policy on both streams
// (aux = 2 is the `nt` bit of buffer loads on gfx950) was measured and changes nothing here (the kernel
// is MFMA-bound: 3.18 ms vs 3.05-3.11 ms at N=4M, D=128), so it stays off unless built with -DBC_K1_NT.
#ifdef BC_K1_NT
#define BC_K1_Z_AUX 2
__device__ __forceinline__ void bc_store2(double* p, double x, double y) {
  bc_d2v v = {x, y};
  __builtin_nontemporal_store(v, reinterpret_cast<bc_d2v*>(p));
}
#else
#define BC_K1_Z_AUX 0
__device__ __forceinline__ void bc_store2(double* p, double x, double y) { *reinterpret_cast<double2*>(p) = make_double2(x, y); }
#endif

struct ProjArgs {
  const double* z;        // [n_rows][dz]
  const double* theta;    // [nt*16][dk]  zero padded            (MFMA kernel)
  const double* theta_v;  // [dk][4][SW]  zero padded            (vector-FMA kernel: d-major, one sample quarter per wave)
  const double* saux;     // [nt*16] per-sample extra (gauss: theta^T Siginv theta)
  const double* rowaux;   // [n_rows] per-row extra (gauss: x^T Siginv x) or null
  double* tiles;
  double* norms;
  double* tile_part;
  long long n_rows;
  int dz, d, dk, s, model;
  int s_total, s_off;     // RAW passes (S > 256): this launch fills samples [s_off, s_off + s) of s_total, un-centred
  double c[8];            // model constants, see model_constants()
#ifdef BC_K1_STAMPS       // diagnostic build: s_memtime of wave 0 at phase boundaries, 32 slots per tile
  unsigned long long* stamps;
#endif
};
#ifdef BC_K1_STAMPS
#define KSTAMP(i) do { if (a.stamps && threadIdx.x == 0) { a.stamps[(size_t)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0 || (i) == 24) a.stamps[(size_t)blockIdx.x * 32 + ((i) == 0 ? 30 : 31)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define KSTAMP(i) do { } while (0)
#endif

// ---- the two transcendental building blocks of the logistic models, written out (the libm versions are general-purpose:
// ~150-200 instructions per log1p(exp(m)); these are ~55, which is what the logistic projections are bound by).
// exp(x) for x <= 0: Cody-Waite reduction x = k ln2 + r, |r| <= ln2/2, degree-12 Taylor polynomial in Horner form
// (next term r^13/13! <= 1.7e-16), v_ldexp.  x below -745 gives 0 like exp().
__device__ __forceinline__ double bc_exp_nonpos(double x) {
  x = (x < -800.) ? -800. : x;                       // (a NaN stays a NaN: the result is NaN like exp()'s)
  const double k = rint(x * 1.4426950408889634);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1. / 479001600.;
  p = fma(p, r, 1. / 39916800.);
  p = fma(p, r, 1. / 3628800.);
  p = fma(p, r, 1. / 362880.);
  p = fma(p, r, 1. / 40320.);
  p = fma(p, r, 1. / 5040.);
  p = fma(p, r, 1. / 720.);
  p = fma(p, r, 1. / 120.);
  p = fma(p, r, 1. / 24.);
  p = fma(p, r, 1. / 6.);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.);
  p = fma(p, r, 1.);
  return ldexp(p, (int)k);
}

// log1p(exp(-a)) for a >= 0 (NaN in, NaN out): u = exp(-a) in (0, 1], f = 1 + u in (1, 2] halved above sqrt(2),
// log f' = 2 atanh(t), t = (f' - 1) / (f' + 1), |t| <= 0.172 (odd series to t^21), plus the first-order correction for
// the rounding of 1 + u.  Measured against 80-bit arithmetic over m in [-800, 100): at most 4.4 ulp, 0.4 on average
// (log1p(exp(m)) of glibc: 1.6 / 0.25); exactly log 2 at a = 0.
__device__ __forceinline__ double bc_log1p_exp_neg(double a) {
  const double u = bc_exp_nonpos(-a);
  const double f = 1. + u;
  const bool hi = f > 1.4142135623730951;
  const double fp = hi ? 0.5 * f : f;
  const double t = (fp - 1.) / (fp + 1.);
  const double t2 = t * t;
  double q = 2. / 21.;
  q = fma(q, t2, 2. / 19.);
  q = fma(q, t2, 2. / 17.);
  q = fma(q, t2, 2. / 15.);
  q = fma(q, t2, 2. / 13.);
  q = fma(q, t2, 2. / 11.);
  q = fma(q, t2, 2. / 9.);
  q = fma(q, t2, 2. / 7.);
  q = fma(q, t2, 2. / 5.);
  q = fma(q, t2, 2. / 3.);
  q = fma(q, t2, 2.);
  const double c = (u - (f - 1.)) / f;
  return fma(q, t, hi ? 0.6931471805599453 : 0.) + c;
}

// model_lr.py:81-86 evaluates two exp and three pow per element; here the powers go through
//   L1 = log(1+e^m),  L2 = log(1+e^-m) = L1 - m   (the smaller of the two is log1p(e^-|m|), the other one adds |m|)
//   (1+e^m)^a = exp(a L1),  (1+e^-m)^a = exp(a L2)          (a < 0, L >= 0: arguments <= 0)
// i.e. one log1p(exp) and three exp.  Same saturation as the reference's IEEE overflow semantics (m -> +inf: +1,
// m -> -inf: -1/b); where the reference flushes (1+inf)^a to exactly 0 this gives e^(a m) < 1e-30: far below the
// 1e-11 of the parity tolerance.
// (With the general libm bodies -- one log1p, four exp -- inlined, the epilogue's fully unrolled loops over the
// accumulators exceeded the unroller's budget, stayed rolled, indexed the accumulator array dynamically and so pushed it
// into scratch memory: 438 scratch stores inside the contraction loop of the S = 100 kernel, 4.9 ms per 1M rows.  With
// the short bodies above the loops unroll again: 0.92 ms.)
__device__ __forceinline__ double bc_logistic_beta_value(double m, double c0, double c1, double c2) {
  const double am = fabs(m);
  const double Ls = bc_log1p_exp_neg(am);            // log(1 + e^-|m|)
  const double Ll = Ls + am;                          // log(1 + e^+|m|)
  const double L1 = (m <= 0.) ? Ls : Ll, L2 = (m <= 0.) ? Ll : Ls;
  return -((c0 * bc_exp_nonpos(c1 * L1)) - (bc_exp_nonpos(c2 * L1) + bc_exp_nonpos(c2 * L2)));
}

template <int MODEL>
__device__ __forceinline__ double bc_model_value(double p, double ra, double sa, const double* c) {
  switch (MODEL) {
    // (2p)*y is evaluated as p*(2y): doubling is exact, so the product rounds to the same double, and 2y -- like y*y --
    // is a per-row value that stays out of the per-sample code
    case BC_MODEL_LINREG_LL: {            // c0 - c1*(y^2 - 2*p*y + p^2)
      const double q = (ra * ra - p * (2. * ra)) + p * p;
      return c[0] - c[1] * q;
    }
    case BC_MODEL_LINREG_BETA: {          // k0*(k1*exp(k2*q) + k3)
      const double q = (ra * ra - p * (2. * ra)) + p * p;
      return c[0] * (c[1] * exp(c[2] * q) + c[3]);
    }
    case BC_MODEL_LOGISTIC_LL: {          // m = -z.th ; m < 100 ? -log1p(exp(m)) : -m ;  log1p(e^m) = max(m, 0) + log1p(e^-|m|)
      const double m = -p;
      return (m < 100.) ? -(fmax(m, 0.) + bc_log1p_exp_neg(fabs(m))) : -m;
    }
    case BC_MODEL_LOGISTIC_BETA:          // -( (b+1)/b*(1+e^m)^-b - ((1+e^m)^(-b-1) + (1+e^-m)^(-b-1)) ), out of line (below)
      return bc_logistic_beta_value(-p, c[0], c[1], c[2]);
    case BC_MODEL_GAUSS_LL: {             // cc - 1/2*(xSx + tSt - 2*xSt)
      const double q = (ra + sa) - 2. * p;
      return c[0] - 1. / 2. * q;
    }
    case BC_MODEL_GAUSS_BETA: {           // 1/b*exp(-.5*b*q) - (1+b)^(-.5d-1)
      const double q = (ra + sa) - 2. * p;
      return c[0] * exp(c[1] * q) - c[2];
    }
    default: {                            // BC_MODEL_GAUSS_BETA_GRAD, gaussian.py:46-62
      const double q = (ra + sa) - 2. * p;
      const double gq = exp(c[1] * q);
      const double t1 = c[3] * (c[0] * gq - c[2]);
      const double t2 = c[4] * gq;
      const double t3 = c[5] * q * gq;
      return ((t1 - t2) - t3) - c[6];
    }
  }
}

// NT = number of 16-sample accumulator tiles, KC = D-chunk staged per LDS pass,
// JT = 16-row sub-tiles per wave (2 -> 4 waves per 128-row tile, 1 -> 8 waves; the latter keeps
// the accumulators of a 200+-sample projection within the register file).
// TL = 0 or 4 "tail" samples beyond the NT tiles (S <= 16*NT + TL): one sample QUAD contracted with
// v_mfma_f64_4x4x4_4b_f64 (see the loop).  S = 100 (every BASELINE config) thus runs 6 tiles + 1 quad = exactly
// 100 samples instead of 7 tiles with 12 padded ones.
template <int MODEL, int NT, int KC, int JT, bool RAW = false, int TL = 0>
__global__ __launch_bounds__(128 / (16 * JT) * 64, (JT == 1 && NT <= 8) ? 4 : 2) void k_project(ProjArgs a) {
  static_assert(TL == 0 || (TL == 4 && !RAW), "tail: exactly one extra sample quad");
  constexpr int NTHR = 128 / (16 * JT) * 64;
  constexpr int NR = NT * 16 + TL;                // rows of (padded) Theta this kernel contracts with
  constexpr int LDZ = KC + 1;    // odd stride: rows (2j, 2j+1) of a lane pair hit distinct banks
  constexpr int LDT = KC + 2;
  constexpr int ZP = (128 * KC) / NTHR;           // 8-byte loads of Z per thread per chunk
  constexpr int TN = NR * KC / 2;                 // 16-byte loads of Theta per chunk (whole block)
  constexpr int TP = (TN + NTHR - 1) / NTHR;
  extern __shared__ double lds[];
  double* Zl = lds;                    // [128][LDZ]
  double* Tl = lds + 128 * LDZ;        // [NR][LDT]   (reused for the column partials after the loop)
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long tile = blockIdx.x;
  const long long r0 = tile * BC_TILE;
  const int S = a.s;
  const int row_base = (JT == 2) ? 32 * w + 2 * j : 16 * w + j;   // this lane's first data row in the tile

  double4_t acc[JT][NT];     // written by the first k-step
  double tv[JT];

  // Staging through buffer loads: a wave-uniform descriptor per operand (SGPRs), ONE 32-bit
  // per-thread byte offset shared by all passes, and a scalar offset per pass -- no 64-bit
  // address VGPRs.  The Z descriptor covers exactly this tile's valid rows, so rows past the end
  // of the data read as 0 (hardware range check); columns past D are clamped to a valid column
  // and multiply the zero padding of Theta.
  static_assert(NTHR % KC == 0 && NTHR % (KC / 2) == 0 && KC % 8 == 0, "staging map");
  constexpr int ZROWS = NTHR / KC;          // rows of Z covered by one pass
  constexpr int TROWS = NTHR / (KC / 2);    // rows of Theta covered by one pass
  const int zc = tid % KC, zrw = tid / KC;
  const int tc = (tid % (KC / 2)) * 2, trw = tid / (KC / 2);
  const long long rows_here = (a.n_rows - r0) < BC_TILE ? (a.n_rows - r0) : BC_TILE;
  const auto zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.z + (size_t)r0 * a.dz), 0,
                                                       (int)(rows_here * a.dz * 8), 0x00020000);
  const auto trsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.theta, 0, NR * a.dk * 8, 0x00020000);
  const int toff = (trw * a.dk + tc) * 8;
  double zr[ZP];
  double2 tr[TP];
  // Loads of one chunk in NPART slices: the first chunk is requested in one go, every later one in slices spread over
  // the contraction of the chunk before it.  (Issued in one go after the barrier, the 23 loads of a chunk held the
  // wave in the issue stage for 2-4k cycles -- the CU's memory pipeline takes them at ~20 B per cycle -- before its
  // first MFMA of the chunk: 15 % of the tile's time with nothing on the matrix pipe from this wave.)
  constexpr int NPART = KC / 8;                        // pairs of k-steps per chunk
  constexpr int NSL = KC / 4 > 2 ? KC / 4 - 2 : 1;     // slices: one per k-step, none in the chunk's last two (their
                                                       // loads would not be back when the chunk is written to LDS)
  auto load_part = [&](int d0, int part) __attribute__((always_inline)) {
    const int col = min(d0 + zc, a.d - 1);
    const int voff = (zrw * a.dz + col) * 8;
#pragma unroll
    for (int q = part * ZP / NSL; q < (part + 1) * ZP / NSL; ++q)
      zr[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(zrsrc, voff, q * ZROWS * a.dz * 8, BC_K1_Z_AUX));
#pragma unroll
    for (int q = part * TP / NSL; q < (part + 1) * TP / NSL; ++q)   // rows past NR are outside the descriptor and read as 0
      tr[q] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(trsrc, toff, (q * TROWS * a.dk + d0) * 8, 0));
  };
  auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < ZP; ++q) Zl[(q * ZROWS + zrw) * LDZ + zc] = zr[q];
#pragma unroll
    for (int q = 0; q < TP; ++q) {
      const bool ok = (q + 1) * NTHR <= TN || tid + q * NTHR < TN;
      if (ok) *reinterpret_cast<double2*>(Tl + (q * TROWS + trw) * LDT + tc) = tr[q];
    }
  };

  const int nchunks = a.dk / KC;
  KSTAMP(0);
#pragma unroll
  for (int part = 0; part < NSL; ++part) load_part(0, part);
  // per-row extra (y / x^T Siginv x): requested now, consumed in the epilogue (a dependent load there cost its
  // full memory latency per tile)
  double ra_pf[JT];
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
    const long long gr = r0 + row_base + jt;
    ra_pf[jt] = 0.;
    if (gr < a.n_rows) {
      if (MODEL == BC_MODEL_LINREG_LL || MODEL == BC_MODEL_LINREG_BETA) ra_pf[jt] = a.z[(size_t)gr * a.dz + a.d];
      else if (MODEL >= BC_MODEL_GAUSS_LL) ra_pf[jt] = a.rowaux[gr];
    }
  }
  const double4_t zero4 = {0., 0., 0., 0.};
  for (int c = 0; c < nchunks; ++c) {
    store_chunk();
    KSTAMP(1 + 5 * c);
    __syncthreads();
    KSTAMP(2 + 5 * c);
    const bool more = c + 1 < nchunks;
    const double* zrow0 = Zl + row_base * LDZ + g;
    const double* trow = Tl + j * LDT + g;
    const double* tquad = Tl + (NT * 16 + (j & 3)) * LDT + g;
    // one k-step (4 features): NT*JT 16x16x4 products + the tail quad.  FIRST: the very first step of the tile starts
    // the accumulators from the instruction's inline-constant 0 (no zero-fill of 100+ VGPRs per tile).
    // (always_inline: left to its heuristics the compiler keeps some of these lambdas out of line in the largest
    // instantiations -- the beta-logistic one -- and the accumulators they capture by reference then live in scratch:
    // 438 scratch stores inside the contraction, 2.35 -> 4.9 ms per 1M rows)
    auto kstep = [&](int kk, auto first) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first)::value;
      double bz[JT];
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) bz[jt] = zrow0[jt * LDZ + kk * 4];
#pragma unroll
      for (int st = 0; st < NT; ++st) {
        const double at = trow[st * 16 * LDT + kk * 4];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
          acc[jt][st] = __builtin_amdgcn_mfma_f64_16x16x4f64(at, bz[jt], FIRST ? zero4 : acc[jt][st], 0, 0, 0);
      }
      if (TL > 0) {
        // the 25th sample quad (S in 97..100) on v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per
        // instruction; lane (g, q = 4*blk + t) supplies A_blk[t][g], B_blk[g][t] and receives D_blk[g][t]
        // (tools/mfma_f64_4x4x4_layout.hip).  With the same Theta quad in all four blocks and the sub-tile's 16
        // rows spread over (blk, t), the lane receives, for ITS row, sample 16*NT + g: the accumulator layout of
        // the 16x16x4 tiles, from the B operand they already hold.  (Round 1 contracted these four samples on the
        // vector pipe: 8 v_fma_f64 + 4 operand reads per k-step and 8 shuffles per tile instead of 2 + 1 + 0.)
        const double at = tquad[kk * 4];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) tv[jt] = __builtin_amdgcn_mfma_f64_4x4x4f64(at, bz[jt], FIRST ? 0. : tv[jt], 0, 0, 0);
      }
    };
#pragma unroll
    for (int kp = 0; kp < NPART; ++kp) {
      if (more && 2 * kp < NSL) load_part((c + 1) * KC, 2 * kp);
      if (kp == 0 && c == 0) kstep(0, std::true_type{});
      else kstep(2 * kp, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);     // keeps each slice of loads with its k-step
      if (more && 2 * kp + 1 < NSL) load_part((c + 1) * KC, 2 * kp + 1);
      kstep(2 * kp + 1, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
    }
    KSTAMP(4 + 5 * c);
    __syncthreads();
    KSTAMP(5 + 5 * c);
  }
  const int s_tail = NT * 16 + g;

  // ---- epilogue: lane holds, for data rows (row_base + jt), samples s = 16*st + g + 4*reg
  if (RAW) {
    // S > 256: write the un-centred model values of this sample range; k_center_tiles finishes the job
    double* rbase = a.tiles + (size_t)tile * a.s_total * BC_TILE + row_base;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      const long long gr = r0 + row_base + jt;
      const bool live = gr < a.n_rows;
      const double ra = ra_pf[jt];
#pragma unroll
      for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int s = 16 * st + g + 4 * reg;
          if (s < S) {
            const double v = live ? bc_model_value<MODEL>(acc[jt][st][reg], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s] : 0., a.c) : 0.;
            rbase[(size_t)(a.s_off + s) * BC_TILE + jt] = v;
          }
        }
    }
    return;
  }
  // column partials reuse the staging LDS (all of it: Zl and Tl are dead after the loop)
  constexpr bool LDSCP = (JT == 2) && (NTHR / 64) * NR * 17 <= 128 * LDZ + NR * LDT;
  double* colpart = LDSCP ? lds : Tl;   // LDSCP: [waves][NR][17], else [waves][NR]
  if (TL > 0) {
    // 96 < S <= 100 (every BASELINE config): all samples of the NT tiles are real ones and every lane holds some, so
    // the `s < S` predicates vanish.  Rows past the end of the shard (last tile only) read as zeros, give finite
    // model values, and are zeroed after the fact under a block-uniform branch instead of a select per element.
    // "All S values of the row are equal" is not tracked per element either: such a row shows up afterwards as a
    // centred row with a vanishing norm and is then examined exactly (below).
    const bool full_tile = rows_here == BC_TILE;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      const double ra = ra_pf[jt];
      double sum = 0.;
#pragma unroll
      for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const double v = bc_model_value<MODEL>(acc[jt][st][reg], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[16 * st + g + 4 * reg] : 0., a.c);
          acc[jt][st][reg] = v;
          sum += v;
        }
      {
        const double v = (s_tail < S) ? bc_model_value<MODEL>(tv[jt], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s_tail] : 0., a.c) : 0.;
        tv[jt] = v;
        sum += v;
      }
      sum += __shfl_xor(sum, 16, BC_WAVE);
      sum += __shfl_xor(sum, 32, BC_WAVE);
      double mean = sum / (double)S;                 // lls.mean(axis=1), tree order
      double sq = 0.;
#pragma unroll
      for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const double v = acc[jt][st][reg] - mean;
          acc[jt][st][reg] = v;
          sq = fma(v, v, sq);
        }
      {
        const double v = (s_tail < S) ? tv[jt] - mean : 0.;
        tv[jt] = v;
        sq = fma(v, v, sq);
      }
      sq += __shfl_xor(sq, 16, BC_WAVE);
      sq += __shfl_xor(sq, 32, BC_WAVE);
      // A row whose S values are all the same number c (a data row with all-zero features): the reference subtracts
      // NumPy's rounded mean of S copies of c, which is c only for some (c, S) -- otherwise the row keeps a tiny
      // constant residue, a non-zero norm, and is NOT one of the "all-zero rows" dropped at hilbert.py:16.  The
      // tree-order sum above rounds differently and would flip that zero / non-zero status, so such rows are
      // re-centred with NumPy's order.  Every constant row lands here: its centred values are a few ulp of c, i.e.
      // sq <= S*(8 eps c)^2, a thousand times inside the bound below (and NaN rows never do: they stay NaN as in the
      // reference).  Inside the bound each v was within 1e-11 of the mean, so v - mean was exact (Sterbenz) and
      // mean + (v - mean) gives v back exactly: "all v equal" is decided, exactly, on the centred values.
      const double tiny = 1e-12 * mean;
      const bool suspect = sq <= (double)S * (tiny * tiny);
      if (__builtin_amdgcn_ballot_w64(suspect) != 0ull) {
        double d0 = acc[jt][0][0];
        asm volatile("" : "+v"(d0));                 // keeps the 25 compares below out of the straight-line code
        bool same = suspect;
#pragma unroll
        for (int st = 0; st < NT; ++st)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) same &= (acc[jt][st][reg] == d0);
        if (s_tail < S) same &= (tv[jt] == d0);
        int ok = same ? 1 : 0;
        ok &= (d0 == __shfl_xor(d0, 16, BC_WAVE)) ? 1 : 0;
        ok &= __shfl_xor(ok, 16, BC_WAVE);
        ok &= (d0 == __shfl_xor(d0, 32, BC_WAVE)) ? 1 : 0;
        ok &= __shfl_xor(ok, 32, BC_WAVE);
        if (ok) {                                    // the four lanes of a constant row take this together
          const double cval = mean + d0;
          mean = bc_np_sum_const_256(cval, S) / (double)S;
          const double v = cval - mean;
          sq = 0.;
#pragma unroll
          for (int st = 0; st < NT; ++st)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              acc[jt][st][reg] = v;
              sq = fma(v, v, sq);
            }
          const double vt = (s_tail < S) ? v : 0.;
          tv[jt] = vt;
          sq = fma(vt, vt, sq);
        }
        // the four lanes of a row agree on `ok`: a recomputed row adds up its four new partial sums, every other
        // row keeps the total it had
        double sq2 = ok ? sq : 0.;
        sq2 += __shfl_xor(sq2, 16, BC_WAVE);
        sq2 += __shfl_xor(sq2, 32, BC_WAVE);
        if (ok) sq = sq2;
      }
      if (!full_tile && !(r0 + row_base + jt < a.n_rows)) {
#pragma unroll
        for (int st = 0; st < NT; ++st) acc[jt][st] = (double4_t){0., 0., 0., 0.};
        tv[jt] = 0.;
        sq = 0.;
      }
      if (g == 0) a.norms[r0 + row_base + jt] = sqrt(sq);
    }
  } else {
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
    const long long gr = r0 + row_base + jt;
    const bool live = gr < a.n_rows;
    const double ra = ra_pf[jt];
    double sum = 0., vmin = INFINITY, vmax = -INFINITY;
    // TL > 0 kernels (96 < S <= 100): every sample of the NT tiles is a real one and every lane holds some, so the
    // `s < S` predicates vanish and "all S values equal" is tracked with compares against the lane's first value
    // (fmin / fmax cost three instructions each with their canonicalisation; a NaN makes the row non-constant,
    // as in the reference, where a NaN row stays NaN).
    double vref = 0.;
    bool differs = false;
#pragma unroll
    for (int st = 0; st < NT; ++st) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int s = 16 * st + g + 4 * reg;
        double v = 0.;
        if (TL > 0) {
          if (live) v = bc_model_value<MODEL>(acc[jt][st][reg], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s] : 0., a.c);
          if (st == 0 && reg == 0) vref = v;
          differs |= (v != vref);
        } else if (s < S && live) {
          v = bc_model_value<MODEL>(acc[jt][st][reg], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s] : 0., a.c);
          vmin = fmin(vmin, v);
          vmax = fmax(vmax, v);
        }
        acc[jt][st][reg] = v;
        sum += v;
      }
    }
    if (TL > 0) {
      double v = 0.;
      if (s_tail < S && live) {
        v = bc_model_value<MODEL>(tv[jt], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s_tail] : 0., a.c);
        differs |= (v != vref);
      }
      tv[jt] = v;
      sum += v;
    }
    sum += __shfl_xor(sum, 16, BC_WAVE);
    sum += __shfl_xor(sum, 32, BC_WAVE);
    bool constant_row;
    double cval;
    if (TL > 0) {
      int df = differs ? 1 : 0;
      df |= (vref != __shfl_xor(vref, 16, BC_WAVE)) ? 1 : 0;
      df |= __shfl_xor(df, 16, BC_WAVE);
      df |= (vref != __shfl_xor(vref, 32, BC_WAVE)) ? 1 : 0;
      df |= __shfl_xor(df, 32, BC_WAVE);
      constant_row = df == 0;
      cval = vref;
    } else {
      vmin = fmin(vmin, __shfl_xor(vmin, 16, BC_WAVE));
      vmin = fmin(vmin, __shfl_xor(vmin, 32, BC_WAVE));
      vmax = fmax(vmax, __shfl_xor(vmax, 16, BC_WAVE));
      vmax = fmax(vmax, __shfl_xor(vmax, 32, BC_WAVE));
      constant_row = vmin == vmax;
      cval = vmax;
    }
    // a row whose S values are all the same number c (a data row with all-zero features): the reference subtracts
    // NumPy's rounded mean of S copies of c, which is c only for some (c, S) -- otherwise the row keeps a tiny constant
    // residue, a non-zero norm, and is NOT one of the "all-zero rows" dropped at hilbert.py:16.  The tree-order sum
    // above would round differently and flip that zero / non-zero status, so such rows use NumPy's order.
    const double mean = (constant_row ? bc_np_sum_const_256(cval, S) : sum) / (double)S;   // lls.mean(axis=1)
    double sq = 0.;
#pragma unroll
    for (int st = 0; st < NT; ++st) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int s = 16 * st + g + 4 * reg;
        double v = acc[jt][st][reg];
        v = ((TL > 0 || s < S) && live) ? v - mean : 0.;
        acc[jt][st][reg] = v;
        sq = fma(v, v, sq);
      }
    }
    if (TL > 0) {
      const double v = (s_tail < S && live) ? tv[jt] - mean : 0.;
      tv[jt] = v;
      sq = fma(v, v, sq);
    }
    sq += __shfl_xor(sq, 16, BC_WAVE);
    sq += __shfl_xor(sq, 32, BC_WAVE);
    if (g == 0) a.norms[r0 + row_base + jt] = sqrt(sq);
  }
  }
  KSTAMP(21);
  // store the tile (JT == 2: two adjacent rows per lane -> 16-byte stores, 256 B contiguous per 16 lanes) through a
  // buffer descriptor of exactly this tile's S*128 doubles: one per-lane byte offset for all stores, the sample's
  // offset as the instruction's scalar operand (no 64-bit address arithmetic per store), samples >= S dropped by the
  // hardware range check.  Column partials: one LDS base per lane, constant offsets.
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.tiles + (size_t)tile * S * BC_TILE), 0,
                                                       S * BC_TILE * 8, 0x00020000);
  const int woff = (g * BC_TILE + row_base) * 8;
  double* cpl = LDSCP ? colpart + (w * NR + g) * 17 + j : colpart + w * NR + g;
  auto put = [&](int s0, double v0, double v1) __attribute__((always_inline)) {       // sample s0 + g of this lane's row(s)
    double cp;
    if (JT == 2) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bc_u4v, (bc_d2v){v0, v1}), wrsrc, woff, s0 * BC_TILE * 8, BC_K1_Z_AUX);
      cp = v0 + v1;
    } else {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bc_u2v, v0), wrsrc, woff, s0 * BC_TILE * 8, BC_K1_Z_AUX);
      cp = v0;
    }
    // per-tile column partial (K2): sum over the tile's rows.  JT == 2 kernels park each lane's pair sum in LDS
    // ([wave][sample][16 row pairs], rows padded to 17) and let one thread per sample add them up in a fixed
    // order -- a 4-step fp64 shuffle reduction per value cost ~8 % of the kernel (0.25 ms per 4M rows).
    if (LDSCP) {
      cpl[s0 * 17] = cp;
    } else {
      cp += __shfl_xor(cp, 1, BC_WAVE);
      cp += __shfl_xor(cp, 2, BC_WAVE);
      cp += __shfl_xor(cp, 4, BC_WAVE);
      cp += __shfl_xor(cp, 8, BC_WAVE);
      if (j == 0) cpl[s0] = cp;
    }
  };
#pragma unroll
  for (int st = 0; st < NT; ++st)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) put(16 * st + 4 * reg, acc[0][st][reg], acc[JT - 1][st][reg]);
  if (TL > 0) put(NT * 16, tv[0], tv[JT - 1]);
  KSTAMP(22);
  __syncthreads();
  KSTAMP(23);
  constexpr int NW = NTHR / 64;
  for (int s = tid; s < S; s += NTHR) {
    double t = 0.;
    if (LDSCP) {
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) {
        const double* row = colpart + (ww * NR + s) * 17;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) t += row[jj];
      }
    } else {
      t = colpart[s];
#pragma unroll
      for (int ww = 1; ww < NW; ++ww) t += colpart[ww * NR + s];
    }
    a.tile_part[(size_t)tile * S + s] = t;
  }
  KSTAMP(24);
}

// S > 256, second stage: centre the rows of one tile (subtract the mean over all s_total samples; constant
// rows get NumPy's rounded mean, see bc_np_sum_const_*), write them back, emit the row norms and the tile's column partial sums.
__global__ __launch_bounds__(256) void k_center_tiles(double* __restrict__ tiles, double* __restrict__ norms,
                                                     double* __restrict__ tile_part, long long n_rows, int S) {
  __shared__ double lds[32 * (BC_TILE + 1)];
  const long long t = blockIdx.x;
  double* tp = tiles + (size_t)t * S * BC_TILE;
  const int tid = threadIdx.x, r = tid & (BC_TILE - 1), half = tid >> 7;   // two threads per row: even / odd chunks of 32 samples
  const bool live = t * BC_TILE + r < n_rows;
  __shared__ double psum[256], pmin[256], pmax[256];
  double sum = 0., vmin = INFINITY, vmax = -INFINITY;
  for (int s = half; s < S; s += 2) {
    const double v = tp[(size_t)s * BC_TILE + r];
    sum += v;
    vmin = fmin(vmin, v);
    vmax = fmax(vmax, v);
  }
  psum[tid] = sum; pmin[tid] = vmin; pmax[tid] = vmax;
  __syncthreads();
  const double tot = psum[r] + psum[128 + r];
  const double mn = fmin(pmin[r], pmin[128 + r]), mx = fmax(pmax[r], pmax[128 + r]);
  const double mean = ((mn == mx) ? bc_np_sum_const_any(mx, S) : tot) / (double)S;   // constant row: NumPy's rounding of the mean
  __syncthreads();
  double sq = 0.;
  for (int s0 = 0; s0 < S; s0 += 32) {
    const int kc = (S - s0) < 32 ? (S - s0) : 32;
    for (int k = half; k < kc; k += 2) {
      double v = tp[(size_t)(s0 + k) * BC_TILE + r];
      v = live ? v - mean : 0.;
      tp[(size_t)(s0 + k) * BC_TILE + r] = v;
      lds[k * (BC_TILE + 1) + r] = v;
      sq = fma(v, v, sq);
    }
    __syncthreads();
    if (tid < kc) {
      double acc = 0.0;
      for (int rr = 0; rr < BC_TILE; ++rr) acc += lds[tid * (BC_TILE + 1) + rr];
      tile_part[(size_t)t * S + s0 + tid] = acc;
    }
    __syncthreads();
  }
  psum[tid] = sq;
  __syncthreads();
  if (tid < BC_TILE) norms[t * BC_TILE + tid] = sqrt(psum[tid] + psum[128 + tid]);
}

// ---------------------------------------------------------------------------------------------
// K1, vector-FMA formulation (S <= 100): measured on MI355X the fp64 vector pipe sustains ~56 TF in
// this register blocking against ~47 TF for v_mfma_f64_16x16x4_f64 (tools/valu_f64_sgpr.hip,
// tools/mfma_f64_peak.hip; both DVFS-limited), and it needs no padding of S to a multiple of 16.
//   block = 256 rows (two Phi tiles), 4 waves; wave w owns the sample quarter [w*SW, (w+1)*SW);
//   lane l owns rows 2l, 2l+1 (tile A) and 128+2l, 129+2l (tile B)  ->  acc[4][SW] in VGPRs;
//   per feature d: 2 x ds_read_b128 fetch the lane's four x values from the LDS slab [KC][258],
//   theta[d][quarter] is wave-uniform: s_load into SGPRs, used directly as the FMA's scalar operand.
// The Z slab is staged with buffer loads (rows past N read as 0) and prefetched one slab ahead.
template <int MODEL, int SW>
__global__ __launch_bounds__(256, 2) void k_project_v(ProjArgs a) {
  constexpr int KC = 8;
  constexpr int LDR = 258;               // slab row length (256 rows + 2): keeps the 16-byte reads aligned
  __shared__ double Zl[KC * LDR];
  __shared__ double rs0[4 * 256];        // cross-wave row reductions: sum / sum of squares
  __shared__ double rs1[4 * 256];        // min
  __shared__ double rs2[4 * 256];        // max
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long blk = blockIdx.x;
  const long long r0 = blk * 256;
  const int S = a.s;

  double acc[4][SW];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int j = 0; j < SW; ++j) acc[rr][j] = 0.;

  const int zc = tid % KC, zrw = tid / KC;       // staging: thread -> (row zrw + 32*q, column zc)
  const long long rows_left = a.n_rows - r0;
  const long long rows_here = rows_left < 256 ? rows_left : 256;
  const auto zrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.z + (size_t)r0 * a.dz), 0,
                                                       (int)(rows_here * a.dz * 8), 0x00020000);
  double zr[KC];
  auto load_chunk = [&](int d0) {
    const int col = min(d0 + zc, a.d - 1);
    const int voff = (zrw * a.dz + col) * 8;
#pragma unroll
    for (int q = 0; q < KC; ++q)
      zr[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(zrsrc, voff, q * 32 * a.dz * 8, BC_K1_Z_AUX));
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int q = 0; q < KC; ++q) Zl[zc * LDR + zrw + 32 * q] = zr[q];
  };

  const int nchunks = a.dk / KC;
  load_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
    store_chunk();
    __syncthreads();
    if (c + 1 < nchunks) load_chunk((c + 1) * KC);
    const double* __restrict__ th = a.theta_v + ((size_t)c * KC * 4 + w) * SW;   // wave-uniform
#pragma unroll 1
    for (int dd = 0; dd < KC; ++dd) {
      const double2 xa = *reinterpret_cast<const double2*>(Zl + dd * LDR + 2 * lane);
      const double2 xb = *reinterpret_cast<const double2*>(Zl + dd * LDR + 128 + 2 * lane);
      const double* __restrict__ t = th + (size_t)dd * 4 * SW;
#pragma unroll
      for (int j = 0; j < SW; ++j) {
        const double tv = t[j];
        acc[0][j] = fma(xa.x, tv, acc[0][j]);
        acc[1][j] = fma(xa.y, tv, acc[1][j]);
        acc[2][j] = fma(xb.x, tv, acc[2][j]);
        acc[3][j] = fma(xb.y, tv, acc[3][j]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue
  const int lrow[4] = {2 * lane, 2 * lane + 1, 128 + 2 * lane, 129 + 2 * lane};
  bool live[4];
  double ra[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const long long gr = r0 + lrow[rr];
    live[rr] = gr < a.n_rows;
    ra[rr] = 0.;
    if (live[rr]) {
      if (MODEL == BC_MODEL_LINREG_LL || MODEL == BC_MODEL_LINREG_BETA) ra[rr] = a.z[(size_t)gr * a.dz + a.d];
      else if (MODEL >= BC_MODEL_GAUSS_LL) ra[rr] = a.rowaux[gr];
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    double sum = 0., vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < SW; ++j) {
      const int sidx = w * SW + j;
      double v = 0.;
      if (sidx < S && live[rr]) {
        v = bc_model_value<MODEL>(acc[rr][j], ra[rr], (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[sidx] : 0., a.c);
        vmin = fmin(vmin, v);
        vmax = fmax(vmax, v);
      }
      acc[rr][j] = v;
      sum += v;
    }
    rs0[w * 256 + lrow[rr]] = sum;
    rs1[w * 256 + lrow[rr]] = vmin;
    rs2[w * 256 + lrow[rr]] = vmax;
  }
  __syncthreads();
  double mean[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = lrow[rr];
    const double tot = ((rs0[r] + rs0[256 + r]) + rs0[512 + r]) + rs0[768 + r];
    const double mn = fmin(fmin(rs1[r], rs1[256 + r]), fmin(rs1[512 + r], rs1[768 + r]));
    const double mx = fmax(fmax(rs2[r], rs2[256 + r]), fmax(rs2[512 + r], rs2[768 + r]));
    mean[rr] = ((mn == mx) ? bc_np_sum_const_256(mx, S) : tot) / (double)S;      // constant row: NumPy's rounding of the mean
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    double sq = 0.;
#pragma unroll
    for (int j = 0; j < SW; ++j) {
      const int sidx = w * SW + j;
      double v = acc[rr][j];
      v = (sidx < S && live[rr]) ? v - mean[rr] : 0.;
      acc[rr][j] = v;
      sq = fma(v, v, sq);
    }
    rs0[w * 256 + lrow[rr]] = sq;
  }
  __syncthreads();
  const long long tileA = 2 * blk, tileB = 2 * blk + 1;
  const bool hasB = tileB * BC_TILE < a.n_rows;
  if (w == 0) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = lrow[rr];
      if (rr < 2 || hasB) a.norms[r0 + r] = sqrt(((rs0[r] + rs0[256 + r]) + rs0[512 + r]) + rs0[768 + r]);
    }
  }
  double* tA = a.tiles + (size_t)tileA * S * BC_TILE + 2 * lane;
  double* tB = a.tiles + (size_t)tileB * S * BC_TILE + 2 * lane;
#pragma unroll
  for (int j = 0; j < SW; ++j) {
    const int sidx = w * SW + j;
    if (sidx < S) {
      bc_store2(tA + (size_t)sidx * BC_TILE, acc[0][j], acc[1][j]);
      if (hasB) bc_store2(tB + (size_t)sidx * BC_TILE, acc[2][j], acc[3][j]);
    }
    // column partials (K2): one per 128-row tile, summed over the wave's 64 lanes
    double cpa = bc_wave_sum(acc[0][j] + acc[1][j]);
    double cpb = bc_wave_sum(acc[2][j] + acc[3][j]);
    if (lane == 0 && sidx < S) {
      a.tile_part[(size_t)tileA * S + sidx] = cpa;
      if (hasB) a.tile_part[(size_t)tileB * S + sidx] = cpb;
    }
  }
}

template <int MODEL>
static int launch_project_v(bc_ctx* ctx, const ProjArgs& a, long long ntiles, int sw) {
  const unsigned grid = (unsigned)((ntiles + 1) / 2);
  if (sw <= 16) hipLaunchKernelGGL((k_project_v<MODEL, 16>), dim3(grid), dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL((k_project_v<MODEL, 25>), dim3(grid), dim3(256), 0, ctx->stream, a);
  BC_HIP(hipGetLastError());
  return BC_OK;
}

// x^T Siginv x per row, in the reference's order: (x * (x.dot(Siginv))).sum(axis=1)   (gaussian.py:10).
// Siginv is staged in LDS when it fits (use_lds), otherwise read through the caches (wave-uniform loads).
__global__ __launch_bounds__(256) void k_row_quadform(const double* __restrict__ z, long long n_rows, int d,
                                                     const double* __restrict__ siginv, double* __restrict__ out,
                                                     int use_lds) {
  extern __shared__ double sl[];   // Siginv [d][d] when use_lds
  const double* __restrict__ sg = siginv;
  if (use_lds) {
    for (int i = threadIdx.x; i < d * d; i += blockDim.x) sl[i] = siginv[i];
    __syncthreads();
    sg = sl;
  }
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (long long)gridDim.x * blockDim.x) {
    const double* x = z + (size_t)r * d;
    double tot = 0.;
    for (int aa = 0; aa < d; ++aa) {
      double t = 0.;
      for (int bb = 0; bb < d; ++bb) t = fma(x[bb], sg[bb * d + aa], t);
      tot += x[aa] * t;
    }
    out[r] = tot;
  }
}

// ------------------------------------------------------------------ host side
static int bc_project_wide(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                           const double* params, int32_t n_params, int64_t row_offset, bc_phi** inout);

struct ProjScratch {
  double* theta = nullptr;
  size_t theta_cap = 0;
  double* saux = nullptr;
  size_t saux_cap = 0;
  double* theta_v = nullptr;
  size_t theta_v_cap = 0;
  double* rowaux = nullptr;
  size_t rowaux_cap = 0;
  double* siginv = nullptr;
  size_t siginv_cap = 0;
  double* pinned = nullptr;
  size_t pinned_cap = 0;
};

static ProjScratch g_scr[16];   // per device

static int grow_dev(double** p, size_t* cap, size_t need) {
  if (need <= *cap) return BC_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  BC_HIP(hipMalloc((void**)p, need * sizeof(double)));
  *cap = need;
  return BC_OK;
}

static int grow_pinned(ProjScratch& sc, size_t need) {
  if (need <= sc.pinned_cap) return BC_OK;
  if (sc.pinned) (void)hipHostFree(sc.pinned);
  sc.pinned = nullptr;
  sc.pinned_cap = 0;
  BC_HIP(hipHostMalloc((void**)&sc.pinned, need * sizeof(double), hipHostMallocDefault));
  sc.pinned_cap = need;
  return BC_OK;
}

// model constants, evaluated in the same expression order as the Python sources
static int model_constants(int model, const double* p, int np, int d, double* c, const double** siginv) {
  const double pi = 3.141592653589793;
  *siginv = nullptr;
  memset(c, 0, 8 * sizeof(double));
  switch (model) {
    case BC_MODEL_LINREG_LL: {
      if (np != 1) return BC_INVALID_ARGUMENT;
      const double sigsq = p[0];
      c[0] = -1. / 2. * log(2. * pi * sigsq);
      c[1] = 1. / (2. * sigsq);
      return BC_OK;
    }
    case BC_MODEL_LINREG_BETA: {
      if (np != 2) return BC_INVALID_ARGUMENT;
      const double sigsq = p[0], beta = p[1];
      c[0] = 1. / pow(2 * pi * sigsq, beta / 2.);
      c[1] = -(beta + 1.) / beta;
      c[2] = -beta / (2. * sigsq);
      c[3] = 1. / sqrt(1. + beta);
      return BC_OK;
    }
    case BC_MODEL_LOGISTIC_LL:
      return np == 0 ? BC_OK : BC_INVALID_ARGUMENT;
    case BC_MODEL_LOGISTIC_BETA: {
      if (np != 1) return BC_INVALID_ARGUMENT;
      const double beta = p[0];
      c[0] = (beta + 1.) / beta;
      c[1] = -beta;
      c[2] = -beta - 1.;
      return BC_OK;
    }
    case BC_MODEL_GAUSS_LL: {
      if (np != 1 + d * d) return BC_INVALID_ARGUMENT;
      const double logdet = p[0];
      c[0] = -(double)d / 2 * log(2 * pi) - 1. / 2. * logdet;
      *siginv = p + 1;
      return BC_OK;
    }
    case BC_MODEL_GAUSS_BETA:
    case BC_MODEL_GAUSS_BETA_GRAD: {
      if (np != 2 + d * d) return BC_INVALID_ARGUMENT;
      const double beta = p[0], logdet = p[1], dd = (double)d;
      c[0] = 1. / beta;
      c[1] = -.5 * beta;
      c[2] = pow(1 + beta, -.5 * dd - 1);
      c[3] = log(pow(2 * pi, -.5 * dd) * pow(exp(logdet), -.5));
      c[4] = 1. / pow(beta, 2);
      c[5] = 1. / (2. * beta);
      c[6] = pow(1 + beta, -.5 * dd - 1.) * log(1. + beta);
      *siginv = p + 2;
      return BC_OK;
    }
  }
  return BC_INVALID_ARGUMENT;
}

template <int MODEL, int NT, int KC, int JT, bool RAW = false, int TL = 0>
static int launch_project(bc_ctx* ctx, const ProjArgs& a, long long ntiles) {
  size_t lds = (size_t)(128 * (KC + 1) + (NT * 16 + TL) * (KC + 2)) * sizeof(double);
#ifdef BC_K1_STAMPS
  if (getenv("BC_K1_EXTRA_LDS")) lds += (size_t)atoi(getenv("BC_K1_EXTRA_LDS"));   // diagnostic: fewer blocks per CU
#endif
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    BC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_project<MODEL, NT, KC, JT, RAW, TL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL((k_project<MODEL, NT, KC, JT, RAW, TL>), dim3((unsigned)ntiles), dim3(128 / (16 * JT) * 64), lds, ctx->stream, a);
  BC_HIP(hipGetLastError());
  return BC_OK;
}

template <int MODEL>
static int launch_project_nt(bc_ctx* ctx, const ProjArgs& a, long long ntiles, int ntsel) {
  switch (ntsel) {
    case 4: return launch_project<MODEL, 4, 32, 2>(ctx, a, ntiles);
    case 6: return launch_project<MODEL, 6, 32, 2, false, 4>(ctx, a, ntiles);      // 96 < S <= 100: 6 tiles + 1 sample quad
    case 7: {
      static const int jt1 = getenv("BC_K1_JT1") ? atoi(getenv("BC_K1_JT1")) : 0;
      if (jt1) return launch_project<MODEL, 7, 32, 1>(ctx, a, ntiles);
      return launch_project<MODEL, 7, 32, 2>(ctx, a, ntiles);
    }
    case 13: return launch_project<MODEL, 13, 16, 1>(ctx, a, ntiles);
    default: return launch_project<MODEL, 16, 16, 1>(ctx, a, ntiles);
  }
}

// One launch of the projection.  raw == false: the whole of Phi (s == s_total <= 256), centred, with
// norms and column partials.  raw == true: samples [s_off, s_off + s) of s_total, un-centred.
static int project_impl(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                        const double* params, int32_t n_params, int64_t row_offset, bc_phi** inout,
                        int32_t s_total, int32_t s_off, bool raw) {
  if (!ctx || !data || !theta || !inout || s <= 0 || (n_params > 0 && !params)) {
    bc_set_error("bc_project: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  if (data->ctx != ctx) { bc_set_error("bc_project: data belongs to another context"); return BC_INVALID_ARGUMENT; }
  if (model < 0 || model > BC_MODEL_GAUSS_BETA_GRAD) { bc_set_error("bc_project: unknown model %d", model); return BC_INVALID_ARGUMENT; }
  if (s > 256) { bc_set_error("bc_project: internal: a single pass handles at most 256 samples"); return BC_INVALID_ARGUMENT; }
  const bool has_y = (model == BC_MODEL_LINREG_LL || model == BC_MODEL_LINREG_BETA);
  const int d = data->dz - (has_y ? 1 : 0);
  if (d <= 0) { bc_set_error("bc_project: data rows too short for this model"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  ProjArgs a;
  memset(&a, 0, sizeof(a));
#ifdef BC_K1_STAMPS
  a.stamps = getenv("BC_K1_STAMP_PTR") ? (unsigned long long*)strtoull(getenv("BC_K1_STAMP_PTR"), nullptr, 10) : nullptr;
#endif
  const double* siginv = nullptr;
  if (model_constants(model, params, n_params, d, a.c, &siginv) != BC_OK) {
    bc_set_error("bc_project: model %d expects a different number of parameters than %d (d = %d)", model, n_params, d);
    return BC_INVALID_ARGUMENT;
  }
  // output handle: reuse buffers when the shape matches
  bc_phi* phi = *inout;
  if (phi && (phi->ctx != ctx || phi->s != s_total || bc_phi_set_rows(phi, data->n_rows) != 0)) {
    bc_set_error("bc_project: *inout has a different S or too little row capacity; pass NULL to allocate");
    return BC_INVALID_ARGUMENT;
  }
  bool fresh = false;
  if (!phi) {
    int rc = bc_phi_alloc(ctx, data->n_rows, s_total, row_offset, &phi);
    if (rc) return rc;
    fresh = true;
  }
  phi->row_offset = row_offset;
  phi->stats_valid = false;

  const int nt = (s + 15) / 16;
  static const int no_tail = getenv("BC_K1_NOTAIL") ? atoi(getenv("BC_K1_NOTAIL")) : 0;
  const bool tail = !raw && s > 96 && s <= 100 && !no_tail;      // 6 MFMA tiles + 4 vector-pipe samples
  const int NTsel = raw ? 16 : tail ? 6 : nt <= 4 ? 4 : nt <= 7 ? 7 : nt <= 13 ? 13 : 16;
  const int NRsel = NTsel * 16 + (tail ? 4 : 0);                 // rows of the zero-padded Theta / saux
  // The vector-FMA kernel is parity-clean but measured SLOWER than the MFMA kernel in round 1
  // (N=4M, D=128: 3.73 ms vs 3.11 ms; its scalar theta loads are not software-pipelined yet), so it is
  // opt-in: BC_K1_VALU=1.
  static const int want_valu = getenv("BC_K1_VALU") ? atoi(getenv("BC_K1_VALU")) : 0;
  const bool use_valu = s <= 100 && want_valu && !raw;    // vector-FMA kernel: sample quarters of <= 25
  const int SWv = s <= 64 ? 16 : 25;
  const int KC = NTsel <= 7 ? 32 : 16;                    // dk is a multiple of 8 (the VALU kernel's chunk) either way
  const int dk = ((d + KC - 1) / KC) * KC;
  ProjScratch& sc = g_scr[ctx->device & 15];
  const size_t th_n = (size_t)NRsel * dk, sa_n = (size_t)NRsel;
  const size_t thv_n = use_valu ? (size_t)dk * 4 * SWv : 0;
  int rc = grow_dev(&sc.theta, &sc.theta_cap, th_n);
  if (!rc) rc = grow_dev(&sc.saux, &sc.saux_cap, sa_n);
  if (!rc && use_valu) rc = grow_dev(&sc.theta_v, &sc.theta_v_cap, thv_n);
  if (!rc) rc = grow_pinned(sc, th_n + sa_n + (siginv ? (size_t)d * d : 0) + thv_n);
  if (rc) { if (fresh) bc_phi_destroy(phi); return rc; }
  // make sure an earlier launch is no longer reading the pinned staging area
  BC_HIP(hipStreamSynchronize(ctx->stream));
  double* hth = sc.pinned;
  double* hsa = sc.pinned + th_n;
  memset(hth, 0, (th_n + sa_n) * sizeof(double));
  if (siginv) {
    // Theta' = (Siginv . Theta^T)^T  so that the contraction yields x^T Siginv theta (gaussian.py:12),
    // tSt = (th * (th.dot(Siginv))).sum(axis=1)                                       (gaussian.py:11)
    for (int q = 0; q < s; ++q) {
      const double* th = theta + (size_t)q * d;
      double tst = 0.;
      for (int aa = 0; aa < d; ++aa) {
        double m1 = 0., m2 = 0.;
        for (int bb = 0; bb < d; ++bb) {
          m1 += siginv[(size_t)aa * d + bb] * th[bb];   // (Siginv . th^T)[aa]
          m2 += th[bb] * siginv[(size_t)bb * d + aa];   // (th . Siginv)[aa]
        }
        hth[(size_t)q * dk + aa] = m1;
        tst += th[aa] * m2;
      }
      hsa[q] = tst;
    }
  } else {
    for (int q = 0; q < s; ++q) memcpy(hth + (size_t)q * dk, theta + (size_t)q * d, (size_t)d * sizeof(double));
  }
  hipError_t e = hipMemcpyAsync(sc.theta, hth, th_n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(sc.saux, hsa, sa_n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && use_valu) {
    // d-major copy for the scalar loads of the vector-FMA kernel: [dk][4 quarters][SW], zero padded
    double* hv = sc.pinned + th_n + sa_n + (siginv ? (size_t)d * d : 0);
    memset(hv, 0, thv_n * sizeof(double));
    for (int q = 0; q < s; ++q) {
      const int wq = q / SWv, jq = q % SWv;
      for (int aa = 0; aa < d; ++aa) hv[((size_t)aa * 4 + wq) * SWv + jq] = hth[(size_t)q * dk + aa];
    }
    e = hipMemcpyAsync(sc.theta_v, hv, thv_n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  }
  a.rowaux = nullptr;
  if (e == hipSuccess && siginv && data->n_rows > 0) {
    rc = grow_dev(&sc.rowaux, &sc.rowaux_cap, (size_t)data->n_rows);
    if (!rc) rc = grow_dev(&sc.siginv, &sc.siginv_cap, (size_t)d * d);
    if (rc) { if (fresh) bc_phi_destroy(phi); return rc; }
    double* hsi = sc.pinned + th_n + sa_n;
    memcpy(hsi, siginv, (size_t)d * d * sizeof(double));
    e = hipMemcpyAsync(sc.siginv, hsi, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      const int use_lds = (size_t)d * d * sizeof(double) <= 60 * 1024;
      long long blocks = (data->n_rows + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(k_row_quadform, dim3((unsigned)blocks), dim3(256), use_lds ? (size_t)d * d * sizeof(double) : 0,
                         ctx->stream, data->z, (long long)data->n_rows, d, sc.siginv, sc.rowaux, use_lds);
      e = hipGetLastError();
      a.rowaux = sc.rowaux;
    }
  }
  if (e != hipSuccess) { if (fresh) bc_phi_destroy(phi); return bc_hip_fail(e, "bc_project staging", __FILE__, __LINE__); }

  a.z = data->z;
  a.theta = sc.theta;
  a.theta_v = sc.theta_v;
  a.saux = sc.saux;
  a.tiles = phi->tiles;
  a.norms = phi->norms;
  a.tile_part = phi->tile_part;
  a.n_rows = data->n_rows;
  a.dz = data->dz;
  a.d = d;
  a.dk = dk;
  a.s = s;
  a.s_total = s_total;
  a.s_off = s_off;
  a.model = model;
  rc = BC_OK;
  if (phi->ntiles > 0) {
    rc = bc_timer_begin(ctx, 1);
    if (!rc && raw) {
      switch (model) {
        case BC_MODEL_LINREG_LL: rc = launch_project<BC_MODEL_LINREG_LL, 16, 16, 1, true>(ctx, a, phi->ntiles); break;
        case BC_MODEL_LINREG_BETA: rc = launch_project<BC_MODEL_LINREG_BETA, 16, 16, 1, true>(ctx, a, phi->ntiles); break;
        case BC_MODEL_LOGISTIC_LL: rc = launch_project<BC_MODEL_LOGISTIC_LL, 16, 16, 1, true>(ctx, a, phi->ntiles); break;
        case BC_MODEL_LOGISTIC_BETA: rc = launch_project<BC_MODEL_LOGISTIC_BETA, 16, 16, 1, true>(ctx, a, phi->ntiles); break;
        case BC_MODEL_GAUSS_LL: rc = launch_project<BC_MODEL_GAUSS_LL, 16, 16, 1, true>(ctx, a, phi->ntiles); break;
        case BC_MODEL_GAUSS_BETA: rc = launch_project<BC_MODEL_GAUSS_BETA, 16, 16, 1, true>(ctx, a, phi->ntiles); break;
        default: rc = launch_project<BC_MODEL_GAUSS_BETA_GRAD, 16, 16, 1, true>(ctx, a, phi->ntiles); break;
      }
    } else if (!rc && use_valu) {
      switch (model) {
        case BC_MODEL_LINREG_LL: rc = launch_project_v<BC_MODEL_LINREG_LL>(ctx, a, phi->ntiles, SWv); break;
        case BC_MODEL_LINREG_BETA: rc = launch_project_v<BC_MODEL_LINREG_BETA>(ctx, a, phi->ntiles, SWv); break;
        case BC_MODEL_LOGISTIC_LL: rc = launch_project_v<BC_MODEL_LOGISTIC_LL>(ctx, a, phi->ntiles, SWv); break;
        case BC_MODEL_LOGISTIC_BETA: rc = launch_project_v<BC_MODEL_LOGISTIC_BETA>(ctx, a, phi->ntiles, SWv); break;
        case BC_MODEL_GAUSS_LL: rc = launch_project_v<BC_MODEL_GAUSS_LL>(ctx, a, phi->ntiles, SWv); break;
        case BC_MODEL_GAUSS_BETA: rc = launch_project_v<BC_MODEL_GAUSS_BETA>(ctx, a, phi->ntiles, SWv); break;
        default: rc = launch_project_v<BC_MODEL_GAUSS_BETA_GRAD>(ctx, a, phi->ntiles, SWv); break;
      }
    } else if (!rc) {
      switch (model) {
        case BC_MODEL_LINREG_LL: rc = launch_project_nt<BC_MODEL_LINREG_LL>(ctx, a, phi->ntiles, NTsel); break;
        case BC_MODEL_LINREG_BETA: rc = launch_project_nt<BC_MODEL_LINREG_BETA>(ctx, a, phi->ntiles, NTsel); break;
        case BC_MODEL_LOGISTIC_LL: rc = launch_project_nt<BC_MODEL_LOGISTIC_LL>(ctx, a, phi->ntiles, NTsel); break;
        case BC_MODEL_LOGISTIC_BETA: rc = launch_project_nt<BC_MODEL_LOGISTIC_BETA>(ctx, a, phi->ntiles, NTsel); break;
        case BC_MODEL_GAUSS_LL: rc = launch_project_nt<BC_MODEL_GAUSS_LL>(ctx, a, phi->ntiles, NTsel); break;
        case BC_MODEL_GAUSS_BETA: rc = launch_project_nt<BC_MODEL_GAUSS_BETA>(ctx, a, phi->ntiles, NTsel); break;
        default: rc = launch_project_nt<BC_MODEL_GAUSS_BETA_GRAD>(ctx, a, phi->ntiles, NTsel); break;
      }
    }
    if (!rc) rc = bc_timer_end(ctx, 1);
  } else {
    e = hipMemsetAsync(phi->norms, 0, BC_TILE * sizeof(double), ctx->stream);
    if (e != hipSuccess) rc = bc_hip_fail(e, "memset", __FILE__, __LINE__);
  }
  if (!rc && !raw) rc = bc_phi_finish_stats(phi);
  if (rc) { if (fresh) bc_phi_destroy(phi); return rc; }
  *inout = phi;
  return BC_OK;
}

// S > 256: passes of <= 256 samples write un-centred values, then one centring pass over Phi.
static int bc_project_wide(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                           const double* params, int32_t n_params, int64_t row_offset, bc_phi** inout) {
  const bool has_y = (model == BC_MODEL_LINREG_LL || model == BC_MODEL_LINREG_BETA);
  const int d = data->dz - (has_y ? 1 : 0);
  bc_phi* phi = *inout;
  const bool fresh = phi == nullptr;
  for (int s_off = 0; s_off < s; s_off += 256) {
    const int cs = (s - s_off) < 256 ? (s - s_off) : 256;
    int rc = project_impl(ctx, data, model, theta + (size_t)s_off * d, cs, params, n_params, row_offset, &phi, s, s_off, true);
    if (rc) { if (fresh && phi) bc_phi_destroy(phi); return rc; }
  }
  if (phi->ntiles > 0) {
    hipLaunchKernelGGL(k_center_tiles, dim3((unsigned)phi->ntiles), dim3(256), 0, ctx->stream, phi->tiles, phi->norms,
                       phi->tile_part, (long long)phi->n_rows, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (fresh) bc_phi_destroy(phi); return bc_hip_fail(e, "k_center_tiles", __FILE__, __LINE__); }
  }
  int rc = bc_phi_finish_stats(phi);
  if (rc) { if (fresh) bc_phi_destroy(phi); return rc; }
  *inout = phi;
  return BC_OK;
}

extern "C" int bc_project(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                          const double* params, int32_t n_params, int64_t row_offset, bc_phi** inout) {
  if (!ctx || !data || !theta || !inout || s <= 0 || (n_params > 0 && !params)) {
    bc_set_error("bc_project: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  if (data->ctx != ctx) { bc_set_error("bc_project: data belongs to another context"); return BC_INVALID_ARGUMENT; }
  if (model < 0 || model > BC_MODEL_GAUSS_BETA_GRAD) { bc_set_error("bc_project: unknown model %d", model); return BC_INVALID_ARGUMENT; }
  if (s > 256) return bc_project_wide(ctx, data, model, theta, s, params, n_params, row_offset, inout);
  return project_impl(ctx, data, model, theta, s, params, n_params, row_offset, inout, s, 0, false);
}
