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
#include "bc_np_exp.h"
#include "bc_np_pow2.h"
#include "bc_layout.h"
#include "bc_k1_math.h"
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
  const double* saux;     // [nt*16] per-sample extra (gauss: theta^T Siginv theta)
  const double* rowaux;   // [n_rows] per-row extra (gauss: x^T Siginv x) or null
  double* tiles;
  double* norms;
  double* tile_part;
  long long n_rows;
  int dz, d, dk, s, model;
  int s_total, s_off;     // RAW passes (S > 256): this launch fills samples [s_off, s_off + s) of s_total, un-centred
  long long ngroups;      // k_project_r: 32-row groups of this launch
  int part_init;          // k_project_r: start the per-wave column partials from tile_part instead of 0 (a later chunk of
                          // a chunked projection, bc_project_from_host: the same sums in the same order as ONE launch)
  double c[8];            // model constants, see model_constants()
  // constant rows whose model value the HOST evaluated (bc_ctx_set_constant_row_values: hosts whose NumPy does not take the
  // SVML exp bc_np_exp.h restates): sorted keys (LINREG_BETA: the row's y), the values, how many; 0: none
  const double* ck;
  const double* cv;
  int nck;
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

// ---- lookup tables of the epilogue's exp / log1p(exp(-a)) bodies (bc_k1_math.h): 578 doubles in global memory, copied
// into LDS by every block of a model that needs them (behind the staging buffers; ~4.6 KB next to the 61 KB those take)
__device__ const unsigned long long g_k1_tab_bits[BC_K1_TAB_DOUBLES] = BC_K1_TABLE_INIT;

template <int MODEL>
constexpr bool bc_model_uses_tables();
// doubles of LDS the tables take: the static ones, plus the per-launch power table of the logistic beta-likelihood
template <int MODEL>
constexpr int bc_model_tab_doubles() { return !bc_model_uses_tables<MODEL>() ? 0 : BC_K1_TAB_DOUBLES + (MODEL == BC_MODEL_LOGISTIC_BETA ? 264 : 0); }
template <int MODEL>
constexpr bool bc_model_uses_tables() {
  return MODEL == BC_MODEL_LINREG_BETA || MODEL == BC_MODEL_LOGISTIC_LL || MODEL == BC_MODEL_LOGISTIC_BETA ||
         MODEL == BC_MODEL_GAUSS_BETA || MODEL == BC_MODEL_GAUSS_BETA_GRAD;
}

// (the logistic beta-likelihood's body, bc_logistic_beta_value, lives in bc_k1_math.h: compiled for the host too, where
// tests/k1_math_harness.c measures it against 80-bit arithmetic)
// BC_K1_GROUP (build-time): how many elements of a row the epilogue lets the scheduler interleave (see the S = 100
// epilogue); the beta-logistic element is four transcendental bodies by itself
#ifndef BC_K1_GROUP
#define BC_K1_GROUP 2
#endif
template <int MODEL>
__device__ __forceinline__ double bc_model_value(double p, double ra, double sa, const double* c, const double* tab) {
  switch (MODEL) {
    // (2p)*y is evaluated as p*(2y): doubling is exact, so the product rounds to the same double, and 2y -- like y*y --
    // is a per-row value that stays out of the per-sample code
    case BC_MODEL_LINREG_LL: {            // c0 - c1*(y^2 - 2*p*y + p^2)
      const double q = (ra * ra - p * (2. * ra)) + p * p;
      return c[0] - c[1] * q;
    }
    case BC_MODEL_LINREG_BETA: {          // k0*(k1*exp(k2*q) + k3)
      const double q = (ra * ra - p * (2. * ra)) + p * p;
      return c[0] * (c[1] * bc_exp_tab_nonpos(c[2] * q, tab) + c[3]);
    }
    case BC_MODEL_LOGISTIC_LL: {          // m = -z.th ; m < 100 ? -log1p(exp(m)) : -m ;  log1p(e^m) = max(m, 0) + log1p(e^-|m|)
      const double m = -p;
      // (|m| is bounded for the body: past 800 it returns 0 either way; fmin drops a NaN, which the other branch keeps)
      return (m < 100.) ? -(fmax(m, 0.) + bc_log1p_exp_neg_tab(fmin(fabs(m), 800.), tab)) : -m;
    }
    case BC_MODEL_LOGISTIC_BETA:          // -( (b+1)/b*(1+e^m)^-b - ((1+e^m)^(-b-1) + (1+e^-m)^(-b-1)) ): bc_k1_math.h, the form with the
                                          // per-launch power table behind the static tables (c[1], c[4..7], c[2]: its series)
      return bc_logistic_beta_value_pt(-p, c[0], c[1], c[4], c[5], c[6], c[7], c[2], tab, tab + BC_K1_TAB_DOUBLES);
    case BC_MODEL_GAUSS_LL: {             // cc - 1/2*(xSx + tSt - 2*xSt)
      const double q = (ra + sa) - 2. * p;
      return c[0] - 1. / 2. * q;
    }
    case BC_MODEL_GAUSS_BETA: {           // 1/b*exp(-.5*b*q) - (1+b)^(-.5d-1)
      const double q = (ra + sa) - 2. * p;
      return c[0] * bc_exp_tab_nonpos(c[1] * q, tab) - c[2];
    }
    default: {                            // BC_MODEL_GAUSS_BETA_GRAD, gaussian.py:46-62
      const double q = (ra + sa) - 2. * p;
      const double gq = bc_exp_tab_nonpos(c[1] * q, tab);
      const double t1 = c[3] * (c[0] * gq - c[2]);
      const double t2 = c[4] * gq;
      const double t3 = c[5] * q * gq;
      return ((t1 - t2) - t3) - c[6];
    }
  }
}

// exp() carrying NumPy's bits on AVX-512 hosts (bc_np_exp.h) where that routine covers the argument, the ordinary one
// in the far tails (|x| >= 707.7: the results there are below the last bit of anything they are added to)
__device__ __forceinline__ double bc_exp_like_numpy(double x) {
  int covered;
  const double e = bc_np_exp(x, &covered);
  return covered ? e : exp(x);
}

// The model value of a CONSTANT row (all S values equal: a data row with all-zero features) with the reference's bits:
// which of these rows the reference's centring leaves exactly 0 depends on the last bit of the constant (section 7 of
// DESIGN.md, golden F13), so their np.exp() is restated exactly; the formulas without a transcendental are already
// bit-identical; the logistic log-likelihood's log1p(exp(0)) = RN(log 2) matches as it is; the logistic beta-likelihood's
// constant at m = 0 (a data row z = 0: two np.power(2, .) calls, model_lr.py:85) is handed in by the caller as c[3] --
// the host layer evaluates the reference's expression with NumPy itself (likelihoods.LogisticRegression.params) --
// and rows that are constant because every sample saturated (m << 0: -((b+1)/b - 1); m >> 0: 1) need no power at all.
template <int MODEL>
__device__ __forceinline__ double bc_model_value_np(double p, double ra, double sa, const double* c, const double* tab) {
  switch (MODEL) {
    case BC_MODEL_LINREG_BETA: {
      const double q = (ra * ra - p * (2. * ra)) + p * p;
      return c[0] * (c[1] * bc_exp_like_numpy(c[2] * q) + c[3]);
    }
    case BC_MODEL_GAUSS_BETA: {
      const double q = (ra + sa) - 2. * p;
      return c[0] * bc_exp_like_numpy(c[1] * q) - c[2];
    }
    case BC_MODEL_GAUSS_BETA_GRAD: {
      const double q = (ra + sa) - 2. * p;
      const double gq = bc_exp_like_numpy(c[1] * q);
      const double t1 = c[3] * (c[0] * gq - c[2]);
      const double t2 = c[4] * gq;
      const double t3 = c[5] * q * gq;
      return ((t1 - t2) - t3) - c[6];
    }
    case BC_MODEL_LOGISTIC_BETA:
      if (p == 0. && c[3] == c[3]) return c[3];
      return bc_model_value<MODEL>(p, ra, sa, c, tab);
    default:
      return bc_model_value<MODEL>(p, ra, sa, c, tab);
  }
}
template <int MODEL>
constexpr bool bc_model_has_np_exp() { return MODEL == BC_MODEL_LINREG_BETA || MODEL == BC_MODEL_GAUSS_BETA || MODEL == BC_MODEL_GAUSS_BETA_GRAD; }
// models whose constant rows are re-evaluated with the reference's bits (bc_model_value_np)
template <int MODEL>
constexpr bool bc_model_const_fixup() { return bc_model_has_np_exp<MODEL>() || MODEL == BC_MODEL_LOGISTIC_BETA; }

// the constant of a constant row from the contraction value `p` of the lane's first sample: every lane of the row
// evaluates its own (they agree up to the last bit), the lane with g == 0 decides
template <int MODEL>
__device__ __forceinline__ double bc_const_row_value(double devval, double p, double ra, double sa, const double* c, int lane, const double* tab,
                                                     const double* ck = nullptr, const double* cv = nullptr, int nck = 0) {
  if (!bc_model_const_fixup<MODEL>()) return devval;
  double v = bc_model_value_np<MODEL>(p, ra, sa, c, tab);
  if (MODEL == BC_MODEL_LINREG_BETA && nck > 0) {
    // the caller's own evaluation of the reference's expression for rows with all-zero features (their value depends on y
    // alone): binary search on y.  (Rare branch of a rare branch; the value still has to agree with the device's to 1e-13.)
    int lo = 0, hi = nck - 1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const double k = ck[mid];
      if (k == ra) { v = cv[mid]; break; }
      if (k < ra) lo = mid + 1; else hi = mid - 1;
    }
  }
  v = __shfl(v, lane & 15, BC_WAVE);
  // the restated value is the same number as the device's own up to the last bits; anything else means the row is
  // constant for another reason than equal arguments (it then keeps the device's value)
  return (fabs(v - devval) <= 1e-13 * fabs(devval)) ? v : devval;
}

// Row statistics of one wave's accumulators, shared by the staged kernel (k_project) and the Theta-resident one
// (k_project_r): on entry acc / tv hold the contraction values p of the lane's JT data rows (row0 + jt; samples
// 16*st + g + 4*reg, tail sample 16*NT + g); on exit the centred model values.  Writes the row norms (STORE).
// full_tile: every row of the tile is a real one (no masking).
template <int MODEL, int NT, int JT, int TL, bool STORE>
__device__ __forceinline__ void k1_row_stats(double4_t (&acc)[JT][NT], double (&tv)[JT], const double (&ra_pf)[JT], const ProjArgs& a,
                                             const int S, const int lane, const int g, const double* tabl, const long long row0,
                                             const bool full_tile) {
  const int s_tail = NT * 16 + g;
  if (TL > 0) {
    // 96 < S <= 100 (every BASELINE config): all samples of the NT tiles are real ones and every lane holds some, so
    // the `s < S` predicates vanish.  Rows past the end of the shard (last tile only) read as zeros, give finite
    // model values, and are zeroed after the fact under a block-uniform branch instead of a select per element.
    // "All S values of the row are equal" is not tracked per element either: such a row shows up afterwards as a
    // centred row with a vanishing norm and is then examined exactly (below).
    #pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      const double ra = ra_pf[jt];
      const double p00 = acc[jt][0][0];          // the contraction value of the lane's first sample (constant rows, below)
      double sum = 0.;
#pragma unroll
      for (int st = 0; st < NT; ++st) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const double v = bc_model_value<MODEL>(acc[jt][st][reg], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[16 * st + g + 4 * reg] : 0., a.c, tabl);
          acc[jt][st][reg] = v;
          sum += v;
          // Pin every BC_K1_GROUP elements: left to itself the compiler splits the table-driven bodies in two stages -- index
          // and LDS read of all 25 elements of the row first, polynomials afterwards -- keeps every intermediate alive in
          // between and spills ~900 VGPRs (the beta-logistic instantiation).  The empty asm consumes the finished values
          // (ordering the arithmetic) and its memory clobber keeps the next group's table reads behind it.
          if (bc_model_uses_tables<MODEL>() && ((4 * st + reg + 1) % BC_K1_GROUP) == 0) asm volatile("" : "+v"(acc[jt][st][reg]), "+v"(sum) :: "memory");
        }
      }
      {
        const double v = (s_tail < S) ? bc_model_value<MODEL>(tv[jt], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s_tail] : 0., a.c, tabl) : 0.;
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
          const double cval = bc_const_row_value<MODEL>(mean + d0, p00, ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[g] : 0., a.c, lane, tabl, a.ck, a.cv, a.nck);
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
      if (!full_tile && !(row0 + jt < a.n_rows)) {
#pragma unroll
        for (int st = 0; st < NT; ++st) acc[jt][st] = (double4_t){0., 0., 0., 0.};
        tv[jt] = 0.;
        sq = 0.;
      }
      if (STORE && g == 0) a.norms[row0 + jt] = sqrt(sq);
    }
  } else {
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
    const long long gr = row0 + jt;
    const bool live = gr < a.n_rows;
    const double ra = ra_pf[jt];
    const double p00 = acc[jt][0][0];            // the contraction value of the lane's first sample (constant rows, below)
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
          if (live) v = bc_model_value<MODEL>(acc[jt][st][reg], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s] : 0., a.c, tabl);
          if (st == 0 && reg == 0) vref = v;
          differs |= (v != vref);
        } else if (s < S && live) {
          v = bc_model_value<MODEL>(acc[jt][st][reg], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s] : 0., a.c, tabl);
          vmin = fmin(vmin, v);
          vmax = fmax(vmax, v);
        }
        acc[jt][st][reg] = v;
        sum += v;
      }
      if (bc_model_uses_tables<MODEL>()) asm volatile("" : "+v"(sum) :: "memory");     // see the S = 100 path above
    }
    if (TL > 0) {
      double v = 0.;
      if (s_tail < S && live) {
        v = bc_model_value<MODEL>(tv[jt], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s_tail] : 0., a.c, tabl);
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
    if (bc_model_const_fixup<MODEL>() && __builtin_amdgcn_ballot_w64(constant_row && live) != 0ull) {
      const double cnp = bc_const_row_value<MODEL>(cval, p00, ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[g] : 0., a.c, lane, tabl, a.ck, a.cv, a.nck);
      if (constant_row && live) {                // the reference's bits for the constant: every element of the row IS it
        cval = cnp;
#pragma unroll
        for (int st = 0; st < NT; ++st)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) acc[jt][st][reg] = cnp;
        if (TL > 0) tv[jt] = cnp;
      }
    }
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
    if (STORE && g == 0) a.norms[row0 + jt] = sqrt(sq);
  }
  }
}

// NT = number of 16-sample accumulator tiles, KC = D-chunk staged per LDS pass,
// JT = 16-row sub-tiles per wave (2 -> 4 waves per 128-row tile, 1 -> 8 waves; the latter keeps
// the accumulators of a 200+-sample projection within the register file).
// TL = 0 or 4 "tail" samples beyond the NT tiles (S <= 16*NT + TL): one sample QUAD contracted with
// v_mfma_f64_4x4x4_4b_f64 (see the loop).  S = 100 (every BASELINE config) thus runs 6 tiles + 1 quad = exactly
// 100 samples instead of 7 tiles with 12 padded ones.
// STORE = false: the store-free mode of the gradient loop (bcores.py:141-146 needs `vecs.sum(axis=0)` only): the same
// contraction, formula, centring and per-tile column partials -- bit for bit -- but neither the tile nor the row norms
// are written; algorithmic traffic 8*128*Dz B per tile.
template <int MODEL, int NT, int KC, int JT, bool RAW = false, int TL = 0, bool STORE = true>
__global__ __launch_bounds__(128 / (16 * JT) * 64, (JT == 1 && NT <= 8) ? 4 : 2) void k_project(ProjArgs a) {
  static_assert(TL == 0 || (TL == 4 && !RAW), "tail: exactly one extra sample quad");
  static_assert(STORE || !RAW, "the raw passes of S > 256 exist to be stored");
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
  double* tabl = lds + 128 * LDZ + NR * LDT;      // lookup tables of the epilogue's exp / log bodies (models that have one)
  if (bc_model_uses_tables<MODEL>()) {
    for (int i = threadIdx.x; i < BC_K1_TAB_DOUBLES; i += NTHR) tabl[i] = __builtin_bit_cast(double, g_k1_tab_bits[i]);
    if (MODEL == BC_MODEL_LOGISTIC_BETA)          // the power table of this launch's beta (bc_k1_math.h), from the global tables
      for (int i = threadIdx.x; i < BC_K1_LOG_N; i += NTHR)
        tabl[BC_K1_TAB_DOUBLES + i] = bc_pow_table_entry(i, a.c[1], reinterpret_cast<const double*>(g_k1_tab_bits));
  }                                               // visible after the first barrier of the contraction loop
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
            const double v = live ? bc_model_value_np<MODEL>(acc[jt][st][reg], ra, (MODEL >= BC_MODEL_GAUSS_LL) ? a.saux[s] : 0., a.c, tabl) : 0.;
            rbase[(size_t)(a.s_off + s) * BC_TILE + jt] = v;
          }
        }
    }
    return;
  }
  // column partials reuse the staging LDS (all of it: Zl and Tl are dead after the loop)
  constexpr bool LDSCP = (JT == 2) && (NTHR / 64) * NR * 17 <= 128 * LDZ + NR * LDT;
  double* colpart = LDSCP ? lds : Tl;   // LDSCP: [waves][NR][17], else [waves][NR]
  k1_row_stats<MODEL, NT, JT, TL, STORE>(acc, tv, ra_pf, a, S, lane, g, tabl, r0 + row_base, rows_here == BC_TILE);
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
      if (STORE) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bc_u4v, (bc_d2v){v0, v1}), wrsrc, woff, s0 * BC_TILE * 8, BC_K1_Z_AUX);
      cp = v0 + v1;
    } else {
      if (STORE) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bc_u2v, v0), wrsrc, woff, s0 * BC_TILE * 8, BC_K1_Z_AUX);
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

// ---------------------------------------------------------------------------------------------
// K1, Theta-RESIDENT formulation (large shards, D <= ~160 at S = 100).
//
// What the staged kernel above pays per 128-row tile besides its MFMAs: the whole of Theta (S x D, 102 KB at the
// headline shape) is re-staged through registers into LDS for EVERY tile, the Z tile is staged the same way, and four
// waves meet at two barriers per D-chunk (profiles/r02_notes.md: staging writes 3.4k, first-chunk wait up to 7.7k of a
// tile's ~38k cycles).  Here:
//   * one 512-thread block per CU keeps Theta in LDS for the whole launch (100 x 130 doubles = 104 KB at D = 128),
//     permuted so that the A-operand reads stay conflict-free (below);
//   * the B operand never touches LDS: lane (j, g) of a wave reads 32 contiguous bytes of ITS data row straight from
//     global memory (two dwordx4 per 16 columns; the four lanes of a row cover one 128-byte line), one 16-column stage
//     ahead of the MFMAs that consume it -- the k index of a 16x16x4 step is lane-group g, so "which column is k" is a free
//     choice as long as Theta uses the same one: k-step (c, t) contracts columns {16c + 4g + t}, and Theta[., 16c + 4g + t]
//     sits at LDS position 16c + 4t + g;
//   * waves are independent: a wave owns 32-row groups (wave id + 8 * gridDim * i), no barrier after the set-up, the
//     next group's first stage and y values are requested before the epilogue of the current one;
//   * column partials (K2) are accumulated per WAVE over all its groups in LDS (one S-vector per wave, written once at
//     the end: tile_part holds gridDim * 8 rows instead of one per tile); assignment of groups to waves is static, so the
//     sums are run-to-run deterministic.
// Same accumulator layout, row statistics (k1_row_stats), Phi layout and norms as the staged kernel.
template <int MODEL, int NT, int TL, bool STORE>
__global__ __launch_bounds__(512, 2) void k_project_r(ProjArgs a) {
  constexpr int JT = 2;
  constexpr int NR = NT * 16 + TL;
  constexpr int TRS = 5 * 4 * 17;             // transposition scratch of one wave: 5 values x 4 lane groups x (16 + 1)
  extern __shared__ double lds[];
  const int S = a.s;
  const int dk = a.dk;                        // multiple of 32: an even number of 16-column stages
  const int ldt = dk + 2;
  double* Tl = lds;                           // [NR][ldt], columns permuted inside every block of 16
  double* csum = Tl + NR * ldt;               // [8][NR] per-wave column partials
  double* trs = csum + 8 * NR;                // [8][TRS]
  double* tabl = trs + 8 * TRS;               // lookup tables of the transcendental bodies (models that have one)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;

  // ---- set-up: Theta -> LDS (wave w takes rows w, w + 8, ...), zero the accumulators, tables
  for (int i = w; i < NR; i += 8)
    for (int col = lane; col < dk; col += 64) Tl[i * ldt + (col & ~15) + 4 * (col & 3) + ((col >> 2) & 3)] = a.theta[(size_t)i * dk + col];
  if (a.part_init) {
    // a chunk of a chunked projection: wave (b, w) continues the partial it left in row (8 b + w) of tile_part.  Chunks
    // start at multiples of 8 * gridDim groups, so the wave meets the same groups in the same order as in one launch
    // over all rows and its additions are the same ones.
    for (int i = tid; i < 8 * NR; i += 512) {
      const int ww = i / NR, ss = i - ww * NR;
      csum[i] = ss < S ? a.tile_part[((size_t)blockIdx.x * 8 + ww) * S + ss] : 0.;
    }
  } else {
    for (int i = tid; i < 8 * NR; i += 512) csum[i] = 0.;
  }
  if (bc_model_uses_tables<MODEL>()) {
    for (int i = tid; i < BC_K1_TAB_DOUBLES; i += 512) tabl[i] = __builtin_bit_cast(double, g_k1_tab_bits[i]);
    if (MODEL == BC_MODEL_LOGISTIC_BETA)
      for (int i = tid; i < BC_K1_LOG_N; i += 512)
        tabl[BC_K1_TAB_DOUBLES + i] = bc_pow_table_entry(i, a.c[1], reinterpret_cast<const double*>(g_k1_tab_bits));
  }
  __syncthreads();

  const double* trow = Tl + j * ldt + g;
  const double* tquad = Tl + (NT * 16 + (j & 3)) * ldt + g;
  double* mycs = csum + w * NR;
  double* mytr = trs + w * TRS;
  const long long wstride = (long long)gridDim.x * 8;
  const int nstage = dk >> 4;
  const bool colmask = (a.d & 31) != 0;       // columns in [d, dk) exist: their Z values are zeroed (Theta's are zero already)
  const int voff0 = ((2 * j) * a.dz + 4 * g) * 8;

  auto rsrc_of = [&](long long grp) __attribute__((always_inline)) {
    const long long row0 = grp * 32;
    long long rows = a.n_rows - row0;
    rows = rows > 32 ? 32 : rows;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(a.z + (size_t)row0 * a.dz), 0, (int)(rows * a.dz * 8), 0x00020000);
  };
  // one 16-column stage of the lane's two rows: b[jt][t] = Z[row0 + 2j + jt][16c + 4g + t]   (rows past N read as 0)
  auto issue = [&](auto rs, int c, double (&b)[JT][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      const bc_d2v lo = __builtin_bit_cast(bc_d2v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff0 + jt * a.dz * 8, c * 128, BC_K1_Z_AUX));
      const bc_d2v hi = __builtin_bit_cast(bc_d2v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff0 + jt * a.dz * 8 + 16, c * 128, BC_K1_Z_AUX));
      b[jt][0] = lo[0]; b[jt][1] = lo[1]; b[jt][2] = hi[0]; b[jt][3] = hi[1];
    }
  };
  auto load_ra = [&](long long grp, double (&ra)[JT]) __attribute__((always_inline)) {
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      const long long gr = grp * 32 + 2 * j + jt;
      ra[jt] = 0.;
      if (gr < a.n_rows) {
        if (MODEL == BC_MODEL_LINREG_LL || MODEL == BC_MODEL_LINREG_BETA) ra[jt] = a.z[(size_t)gr * a.dz + a.d];
        else if (MODEL >= BC_MODEL_GAUSS_LL) ra[jt] = a.rowaux[gr];
      }
    }
  };

  double4_t acc[JT][NT];
  double tv[JT];
  const double4_t zero4 = {0., 0., 0., 0.};
  // the four k-steps of stage c; FIRST: the accumulators start from the instruction's inline-constant 0
  auto kstage = [&](int c, double (&b)[JT][4], auto first) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first)::value;
    if (colmask && 16 * c + 16 > a.d) {       // wave-uniform: only the last stages of a D that is not a multiple of 32
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int t = 0; t < 4; ++t) b[jt][t] = (16 * c + 4 * g + t < a.d) ? b[jt][t] : 0.;
    }
    const double* tr0 = trow + 16 * c;
    const double* tq0 = tquad + 16 * c;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int st = 0; st < NT; ++st) {
        const double at = tr0[st * 16 * ldt + 4 * t];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
          acc[jt][st] = __builtin_amdgcn_mfma_f64_16x16x4f64(at, b[jt][t], (FIRST && t == 0) ? zero4 : acc[jt][st], 0, 0, 0);
      }
      if (TL > 0) {
        const double at = tq0[4 * t];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) tv[jt] = __builtin_amdgcn_mfma_f64_4x4x4f64(at, b[jt][t], (FIRST && t == 0) ? 0. : tv[jt], 0, 0, 0);
      }
    }
  };

#ifdef BC_K1_STAMPS
  unsigned long long ph[4] = {0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime(), tstart = tprev;
#define RSTAMP(i) do { const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[i] += tn - tprev; tprev = tn; } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif
  long long grp = (long long)blockIdx.x * 8 + w;
  double bA[JT][4], bB[JT][4], ra_next[JT] = {0., 0.};
  if (grp < a.ngroups) {
    issue(rsrc_of(grp), 0, bA);
    load_ra(grp, ra_next);
  }
  while (grp < a.ngroups) {
    const long long nxt = grp + wstride;
    const bool has_next = nxt < a.ngroups;
    const auto rs = rsrc_of(grp);
    double ra_pf[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) ra_pf[jt] = ra_next[jt];
    // ---- contraction: stage c + 1 is in flight while stage c is consumed; the last stage overlaps the NEXT group's first
    issue(rs, 1, bB);
    kstage(0, bA, std::true_type{});
    for (int c = 1; c + 1 < nstage; c += 2) {
      issue(rs, c + 1, bA);
      kstage(c, bB, std::false_type{});
      issue(rs, c + 2, bB);
      kstage(c + 1, bA, std::false_type{});
    }
    if (has_next) {
      issue(rsrc_of(nxt), 0, bA);
      load_ra(nxt, ra_next);
    }
    kstage(nstage - 1, bB, std::false_type{});
    RSTAMP(0);

    // ---- epilogue: model values, centring, norms (shared with the staged kernel)
    const long long row0 = grp * 32 + 2 * j;
    k1_row_stats<MODEL, NT, JT, TL, STORE>(acc, tv, ra_pf, a, S, lane, g, tabl, row0, grp * 32 + 32 <= a.n_rows);
    RSTAMP(1);
    const long long tile = grp >> 2;
    const int row_base = 32 * (int)(grp & 3) + 2 * j;
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.tiles + (size_t)tile * S * BC_TILE), 0, S * BC_TILE * 8, 0x00020000);
    const int woff = (g * BC_TILE + row_base) * 8;
    // stores + column partials.  The 25 pair sums of a lane go through the wave's transposition scratch five at a time:
    // [value][g][16 row pairs] -> lanes 0..19 add up one (value, g) each in row order and add the total to the wave's
    // running column sum.  LDS operations of one wave execute in issue order: no barrier, only the compiler is held.
    auto flush = [&](int v0idx, int nvals) __attribute__((always_inline)) {
      asm volatile("" ::: "memory");
      if (lane < 4 * nvals) {
        const double* src = mytr + lane * 17;
        double t = 0.;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) t += src[jj];
        const int val = v0idx + (lane >> 2), gg = lane & 3;          // value index 0..24 -> sample
        const int smp = (val < NT * 4) ? 16 * (val >> 2) + 4 * (val & 3) + gg : NT * 16 + gg;
        if (smp < S) mycs[smp] += t;
      }
      asm volatile("" ::: "memory");
    };
    int nq = 0;                                                        // values parked since the last flush
    int vbase = 0;
    auto put = [&](int vidx, int s0, double v0, double v1) __attribute__((always_inline)) {
      if (STORE) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bc_u4v, (bc_d2v){v0, v1}), wrsrc, woff, s0 * BC_TILE * 8, BC_K1_Z_AUX);
      mytr[((vidx - vbase) * 4 + g) * 17 + j] = v0 + v1;
    };
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int vidx = 4 * st + reg;
        put(vidx, 16 * st + 4 * reg, acc[0][st][reg], acc[1][st][reg]);
        if (++nq == 5) { flush(vbase, 5); vbase += 5; nq = 0; }
      }
    if (TL > 0) {
      put(4 * NT, NT * 16, tv[0], tv[1]);
      ++nq;
    }
    if (nq > 0) flush(vbase, nq);
    RSTAMP(2);
    grp = nxt;
  }
#ifdef BC_K1_STAMPS
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 8 + w) * 8;
    o[0] = ph[0]; o[1] = ph[1]; o[2] = ph[2]; o[3] = __builtin_amdgcn_s_memtime() - tstart; o[4] = tstart;
  }
#endif
  // ---- the wave's column sums: row (blockIdx.x * 8 + w) of tile_part
  asm volatile("" ::: "memory");
  double* outp = a.tile_part + ((size_t)blockIdx.x * 8 + w) * S;
  for (int s = lane; s < S; s += 64) outp[s] = mycs[s];
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

static int grow_pinned(bc_ctx* ctx, size_t need) {
  if (need <= ctx->proj_pinned_cap) return BC_OK;
  BC_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->proj_pinned) (void)hipHostFree(ctx->proj_pinned);
  ctx->proj_pinned = nullptr;
  ctx->proj_pinned_cap = 0;
  BC_HIP(hipHostMalloc((void**)&ctx->proj_pinned, need * sizeof(double), hipHostMallocDefault));
  ctx->proj_pinned_cap = need;
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
      // params = [beta] or [beta, value of the beta-likelihood at m = 0 with the caller's np.power bits] (see
      // bc_model_value_np); without the second one the library evaluates that value itself with the restated np.power
      if (np != 1 && np != 2) return BC_INVALID_ARGUMENT;
      const double beta = p[0];
      if (!(beta <= BC_K1_POWTAB_MAX_BETA)) {
        bc_set_error("bc_project: the logistic beta-likelihood takes 0 < beta <= %g on the device (got %g)", BC_K1_POWTAB_MAX_BETA, beta);
        return BC_INVALID_ARGUMENT;
      }
      c[0] = (beta + 1.) / beta;
      {
        double k[6];
        bc_powtab_coefs(-beta, &k);        // (1 + t)^-beta to t^6: c[1] = -beta, c[4..7], c[2]
        c[1] = k[0];
        c[4] = k[1];
        c[5] = k[2];
        c[6] = k[3];
        c[7] = k[4];
        c[2] = k[5];
      }
      if (np == 2) {
        c[3] = p[1];
      } else {
        // no constant from the caller: the reference's expression at m = 0 with np.power's bits at base 2 restated
        // (bc_np_pow2.h: what NumPy computes on AVX-512 hosts, the hosts the goldens come from)
        int cov1 = 0, cov2 = 0;
        double p1 = bc_np_pow2(-beta, &cov1), p2 = bc_np_pow2(-beta - 1., &cov2);
        if (!cov1) p1 = pow(2., -beta);
        if (!cov2) p2 = pow(2., -beta - 1.);
        c[3] = -(((beta + 1.) / beta) * p1 - (p2 + p2));
      }
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

enum { PROJ_FULL = 0, PROJ_RAW = 1, PROJ_COLSUM = 2 };

template <int MODEL, int NT, int KC, int JT, bool RAW = false, int TL = 0, bool STORE = true>
static int launch_project(bc_ctx* ctx, const ProjArgs& a, long long ntiles) {
  size_t lds = (size_t)(128 * (KC + 1) + (NT * 16 + TL) * (KC + 2)) * sizeof(double);
  if (bc_model_uses_tables<MODEL>()) lds += (size_t)bc_model_tab_doubles<MODEL>() * sizeof(double);
#ifdef BC_K1_STAMPS
  if (getenv("BC_K1_EXTRA_LDS")) lds += (size_t)atoi(getenv("BC_K1_EXTRA_LDS"));   // diagnostic: fewer blocks per CU
#endif
  if (lds > (size_t)ctx->max_lds) {
    bc_set_error("bc_project: this instantiation stages %zu bytes of LDS per block, the device allows %d", lds, ctx->max_lds);
    return BC_INVALID_ARGUMENT;
  }
  static unsigned attr_done = 0;            // per instantiation, one bit per device ordinal
  const unsigned bit = 1u << (ctx->device & 31);
  if (!(attr_done & bit) && lds > 64 * 1024) {
    BC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_project<MODEL, NT, KC, JT, RAW, TL, STORE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done |= bit;
  }
  hipLaunchKernelGGL((k_project<MODEL, NT, KC, JT, RAW, TL, STORE>), dim3((unsigned)ntiles), dim3(128 / (16 * JT) * 64), lds, ctx->stream, a);
  BC_HIP(hipGetLastError());
  return BC_OK;
}

template <int MODEL, bool STORE>
static int launch_project_nt(bc_ctx* ctx, const ProjArgs& a, long long ntiles, int ntsel) {
  switch (ntsel) {
    case 4: return launch_project<MODEL, 4, 32, 2, false, 0, STORE>(ctx, a, ntiles);
    case 6: return launch_project<MODEL, 6, 32, 2, false, 4, STORE>(ctx, a, ntiles);      // 96 < S <= 100: 6 tiles + 1 sample quad
    case 7: return launch_project<MODEL, 7, 32, 2, false, 0, STORE>(ctx, a, ntiles);
    case 13: return launch_project<MODEL, 13, 16, 1, false, 0, STORE>(ctx, a, ntiles);
    default: return launch_project<MODEL, 16, 16, 1, false, 0, STORE>(ctx, a, ntiles);
  }
}

template <bool STORE>
static int launch_project_model(bc_ctx* ctx, const ProjArgs& a, long long ntiles, int model, int ntsel) {
  switch (model) {
    case BC_MODEL_LINREG_LL: return launch_project_nt<BC_MODEL_LINREG_LL, STORE>(ctx, a, ntiles, ntsel);
    case BC_MODEL_LINREG_BETA: return launch_project_nt<BC_MODEL_LINREG_BETA, STORE>(ctx, a, ntiles, ntsel);
    case BC_MODEL_LOGISTIC_LL: return launch_project_nt<BC_MODEL_LOGISTIC_LL, STORE>(ctx, a, ntiles, ntsel);
    case BC_MODEL_LOGISTIC_BETA: return launch_project_nt<BC_MODEL_LOGISTIC_BETA, STORE>(ctx, a, ntiles, ntsel);
    case BC_MODEL_GAUSS_LL: return launch_project_nt<BC_MODEL_GAUSS_LL, STORE>(ctx, a, ntiles, ntsel);
    case BC_MODEL_GAUSS_BETA: return launch_project_nt<BC_MODEL_GAUSS_BETA, STORE>(ctx, a, ntiles, ntsel);
    default: return launch_project_nt<BC_MODEL_GAUSS_BETA_GRAD, STORE>(ctx, a, ntiles, ntsel);
  }
}

// ---- Theta-resident kernel: one 512-thread block per CU, LDS = Theta + per-wave column sums + transposition scratch (+ tables)
static size_t project_r_lds_bytes(int nr, int dk, int table_doubles) {
  return ((size_t)nr * (dk + 2) + 8 * (size_t)nr + 8 * (5 * 4 * 17) + (size_t)table_doubles) * sizeof(double);
}

template <int MODEL, int NT, int TL, bool STORE>
static int launch_project_r(bc_ctx* ctx, const ProjArgs& a, int grid) {
  const size_t lds = project_r_lds_bytes(NT * 16 + TL, a.dk, bc_model_tab_doubles<MODEL>());
  static unsigned attr_done = 0;
  const unsigned bit = 1u << (ctx->device & 31);
  if (!(attr_done & bit)) {
    BC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_project_r<MODEL, NT, TL, STORE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, ctx->max_lds));
    attr_done |= bit;
  }
  hipLaunchKernelGGL((k_project_r<MODEL, NT, TL, STORE>), dim3((unsigned)grid), dim3(512), lds, ctx->stream, a);
  BC_HIP(hipGetLastError());
  return BC_OK;
}

template <int MODEL, bool STORE>
static int launch_project_r_nt(bc_ctx* ctx, const ProjArgs& a, int grid, int ntsel) {
  switch (ntsel) {
    case 4: return launch_project_r<MODEL, 4, 0, STORE>(ctx, a, grid);
    case 6: return launch_project_r<MODEL, 6, 4, STORE>(ctx, a, grid);
    default:
      if constexpr (bc_model_has_np_exp<MODEL>()) {      // not instantiated (would spill), never selected (project_r_grid)
        bc_set_error("bc_project: internal: no resident kernel for this model at S in 101..112");
        return BC_INVALID_ARGUMENT;
      } else {
        return launch_project_r<MODEL, 7, 0, STORE>(ctx, a, grid);
      }
  }
}

template <bool STORE>
static int launch_project_r_model(bc_ctx* ctx, const ProjArgs& a, int grid, int model, int ntsel) {
  switch (model) {
    case BC_MODEL_LINREG_LL: return launch_project_r_nt<BC_MODEL_LINREG_LL, STORE>(ctx, a, grid, ntsel);
    case BC_MODEL_LINREG_BETA: return launch_project_r_nt<BC_MODEL_LINREG_BETA, STORE>(ctx, a, grid, ntsel);
    case BC_MODEL_LOGISTIC_LL: return launch_project_r_nt<BC_MODEL_LOGISTIC_LL, STORE>(ctx, a, grid, ntsel);
    case BC_MODEL_LOGISTIC_BETA: return launch_project_r_nt<BC_MODEL_LOGISTIC_BETA, STORE>(ctx, a, grid, ntsel);
    case BC_MODEL_GAUSS_LL: return launch_project_r_nt<BC_MODEL_GAUSS_LL, STORE>(ctx, a, grid, ntsel);
    case BC_MODEL_GAUSS_BETA: return launch_project_r_nt<BC_MODEL_GAUSS_BETA, STORE>(ctx, a, grid, ntsel);
    default: return launch_project_r_nt<BC_MODEL_GAUSS_BETA_GRAD, STORE>(ctx, a, grid, ntsel);
  }
}

static int launch_project_raw(bc_ctx* ctx, const ProjArgs& a, long long ntiles, int model) {
  switch (model) {
    case BC_MODEL_LINREG_LL: return launch_project<BC_MODEL_LINREG_LL, 16, 16, 1, true>(ctx, a, ntiles);
    case BC_MODEL_LINREG_BETA: return launch_project<BC_MODEL_LINREG_BETA, 16, 16, 1, true>(ctx, a, ntiles);
    case BC_MODEL_LOGISTIC_LL: return launch_project<BC_MODEL_LOGISTIC_LL, 16, 16, 1, true>(ctx, a, ntiles);
    case BC_MODEL_LOGISTIC_BETA: return launch_project<BC_MODEL_LOGISTIC_BETA, 16, 16, 1, true>(ctx, a, ntiles);
    case BC_MODEL_GAUSS_LL: return launch_project<BC_MODEL_GAUSS_LL, 16, 16, 1, true>(ctx, a, ntiles);
    case BC_MODEL_GAUSS_BETA: return launch_project<BC_MODEL_GAUSS_BETA, 16, 16, 1, true>(ctx, a, ntiles);
    default: return launch_project<BC_MODEL_GAUSS_BETA_GRAD, 16, 16, 1, true>(ctx, a, ntiles);
  }
}

// ---- a projection in two steps: the plan (model constants, Theta zero-padded and uploaded: once per Theta) and the
// launches that use it (one per set of rows: bc_vi_gradient projects the data rows and the coreset rows with one plan)
struct ProjPlan {
  ProjArgs a;             // constants, theta, saux, d, dk, s, model filled in; the per-launch fields are set by plan_launch
  int model = 0, s = 0, d = 0, ntsel = 0;
  bool raw = false;
  const double* siginv_dev = nullptr;      // Gaussian models: Siginv [d][d] on the device
};

static bool model_has_y(int model) { return model == BC_MODEL_LINREG_LL || model == BC_MODEL_LINREG_BETA; }

// `extra` (optional): further host arrays shipped in the same transfer (bc_vi_gradient: the coreset rows and their
// weights); extra_dev[i] receives the device address of extra_src[i].  ONE pinned staging area, ONE device buffer, ONE
// hipMemcpyAsync per plan: a copy call costs ~5 us of host time, a gradient step of the 1M-row configuration 350.
static int plan_stage(bc_ctx* ctx, int model, const double* theta, int32_t s, const double* params, int32_t n_params,
                      int dz, bool raw, ProjPlan* pl, int n_extra = 0, const double* const* extra_src = nullptr,
                      const size_t* extra_n = nullptr, double** extra_dev = nullptr) {
  if (s <= 0 || s > 256) { bc_set_error("bc_project: internal: a single pass handles 1..256 samples"); return BC_INVALID_ARGUMENT; }
  const int d = dz - (model_has_y(model) ? 1 : 0);
  if (d <= 0) { bc_set_error("bc_project: data rows too short for this model"); return BC_INVALID_ARGUMENT; }
  ProjArgs& a = pl->a;
  memset(&a, 0, sizeof(a));
#ifdef BC_K1_STAMPS
  a.stamps = getenv("BC_K1_STAMP_PTR") ? (unsigned long long*)strtoull(getenv("BC_K1_STAMP_PTR"), nullptr, 10) : nullptr;
#endif
  const double* siginv = nullptr;
  if (model_constants(model, params, n_params, d, a.c, &siginv) != BC_OK) {
    bc_set_error("bc_project: model %d expects a different number of parameters than %d (d = %d)", model, n_params, d);
    return BC_INVALID_ARGUMENT;
  }
  if (ctx->n_const_rows > 0 && ctx->const_model == model && ctx->const_n_params == n_params &&
      memcmp(ctx->const_params, params, (size_t)n_params * sizeof(double)) == 0) {
    a.ck = ctx->const_rows.p;                   // host-evaluated constants for exactly this model and these parameters
    a.cv = ctx->const_rows.p + ctx->n_const_rows;
    a.nck = (int)ctx->n_const_rows;
  }
  const int nt = (s + 15) / 16;
  static const int no_tail = getenv("BC_K1_NOTAIL") ? atoi(getenv("BC_K1_NOTAIL")) : 0;
  const bool tail = !raw && s > 96 && s <= 100 && !no_tail;      // 6 MFMA tiles + one sample quad
  const int NTsel = raw ? 16 : tail ? 6 : nt <= 4 ? 4 : nt <= 7 ? 7 : nt <= 13 ? 13 : 16;
  const int NRsel = NTsel * 16 + (tail ? 4 : 0);                 // rows of the zero-padded Theta / saux
  const int KC = NTsel <= 7 ? 32 : 16;
  const int dk = ((d + KC - 1) / KC) * KC;
  const size_t th_n = (size_t)NRsel * dk, sa_n = (size_t)NRsel, sg_n = siginv ? (size_t)d * d : 0;
  const size_t head_n = (th_n + sa_n + sg_n + 1) & ~(size_t)1;
  size_t total = head_n;
  for (int i = 0; i < n_extra; ++i) total += (extra_n[i] + 1) & ~(size_t)1;      // 16-byte aligned pieces
  int rc = bc_scratch_grow(ctx, &ctx->proj_theta, total);
  if (!rc) rc = grow_pinned(ctx, total);
  if (rc) return rc;
  // make sure an earlier launch is no longer reading the pinned staging area
  BC_HIP(hipStreamSynchronize(ctx->stream));
  double* hth = ctx->proj_pinned;
  double* hsa = ctx->proj_pinned + th_n;
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
    memcpy(ctx->proj_pinned + th_n + sa_n, siginv, sg_n * sizeof(double));
  } else {
    for (int q = 0; q < s; ++q) memcpy(hth + (size_t)q * dk, theta + (size_t)q * d, (size_t)d * sizeof(double));
  }
  size_t off = head_n;
  for (int i = 0; i < n_extra; ++i) {
    memcpy(ctx->proj_pinned + off, extra_src[i], extra_n[i] * sizeof(double));
    extra_dev[i] = ctx->proj_theta.p + off;
    off += (extra_n[i] + 1) & ~(size_t)1;
  }
  BC_HIP(hipMemcpyAsync(ctx->proj_theta.p, ctx->proj_pinned, total * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  a.theta = ctx->proj_theta.p;
  a.saux = ctx->proj_theta.p + th_n;
  a.d = d;
  a.dk = dk;
  a.s = s;
  a.model = model;
  pl->model = model;
  pl->s = s;
  pl->d = d;
  pl->ntsel = NTsel;
  pl->raw = raw;
  pl->siginv_dev = siginv ? ctx->proj_theta.p + th_n + sa_n : nullptr;
  return BC_OK;
}

// The resident kernel serves large shards whose Theta fits one CU's LDS next to the per-wave scratch: at least 8 tiles
// per wave slot of the grid (tile_part then has room for the per-wave partial rows, and the grid is full), S <= 112.
// BC_K1_STAGED=1 forces the staged kernel (A/B measurements).
static bool bc_model_has_np_exp_rt(int model) { return model == BC_MODEL_LINREG_BETA || model == BC_MODEL_GAUSS_BETA || model == BC_MODEL_GAUSS_BETA_GRAD; }

static int project_r_grid(const bc_ctx* ctx, const ProjPlan& pl, const bc_phi* phi, int mode) {
  const char* env = getenv("BC_K1_STAGED");          // read per call: tests toggle it inside one process
  if ((env && atoi(env) > 0) || mode == PROJ_RAW) return 0;
  if (pl.ntsel != 4 && pl.ntsel != 6 && pl.ntsel != 7) return 0;
  // S in 101..112 with an exp in the epilogue: those three instantiations need 4-6 VGPRs more than the 256 a wave of a
  // 512-thread block may hold (they would spill): the staged kernel serves them
  if (pl.ntsel == 7 && bc_model_has_np_exp_rt(pl.model)) return 0;
  if (pl.model == BC_MODEL_LOGISTIC_BETA && !(env && atoi(env) < 0)) return 0;      // measured slower there (0.80 vs 0.74 ms at N = 1M, D = 128): four
                                                                                       // transcendental bodies per element; BC_K1_STAGED=-1 forces the resident kernel
  const int nr = pl.ntsel * 16 + (pl.ntsel == 6 ? 4 : 0);
  if (project_r_lds_bytes(nr, pl.a.dk, BC_K1_TAB_DOUBLES + 264) > (size_t)ctx->max_lds) return 0;
  const long long grid = ctx->n_cu;
  if (phi->ntiles < grid * 8) return 0;
  return (int)grid;
}

// One launch of a staged projection over `data`'s rows.  mode PROJ_FULL: the whole of Phi (s == s_total <= 256),
// centred, with norms and column partials.  PROJ_RAW: samples [s_off, s_off + s) of s_total, un-centred.
// PROJ_COLSUM: `phi` is a stats-only Phi: column partials only.  rowaux: the scratch that receives x^T Siginv x.
static int plan_launch(bc_ctx* ctx, const ProjPlan& pl, const bc_data* data, bc_phi* phi, int mode, int32_t s_total,
                       int32_t s_off, bc_scratch* rowaux) {
  ProjArgs a = pl.a;
  a.rowaux = nullptr;
  if (pl.siginv_dev && data->n_rows > 0) {
    int rc = bc_scratch_grow(ctx, rowaux, (size_t)data->n_rows);
    if (rc) return rc;
    const int d = pl.d;
    const int use_lds = (size_t)d * d * sizeof(double) <= 60 * 1024;
    long long blocks = (data->n_rows + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_row_quadform, dim3((unsigned)blocks), dim3(256), use_lds ? (size_t)d * d * sizeof(double) : 0,
                       ctx->stream, data->z, (long long)data->n_rows, d, pl.siginv_dev, rowaux->p, use_lds);
    BC_HIP(hipGetLastError());
    a.rowaux = rowaux->p;
  }
  a.z = data->z;
  a.tiles = phi->tiles;
  a.norms = phi->norms;
  a.tile_part = phi->tile_part;
  a.n_rows = data->n_rows;
  a.dz = data->dz;
  a.s_total = s_total;
  a.s_off = s_off;
  if (phi->ntiles <= 0) {
    if (phi->norms) BC_HIP(hipMemsetAsync(phi->norms, 0, BC_TILE * sizeof(double), ctx->stream));
    return BC_OK;
  }
  int rc = bc_timer_begin(ctx, 1);
  if (rc) return rc;
  const int rgrid = project_r_grid(ctx, pl, phi, mode);
  phi->part_rows = phi->ntiles;                 // rows of tile_part this launch fills (one per tile, or one per wave)
  if (rgrid > 0) {
    a.ngroups = (data->n_rows + 31) / 32;
    phi->part_rows = (int64_t)rgrid * 8;
    rc = mode == PROJ_COLSUM ? launch_project_r_model<false>(ctx, a, rgrid, pl.model, pl.ntsel)
                             : launch_project_r_model<true>(ctx, a, rgrid, pl.model, pl.ntsel);
  } else if (mode == PROJ_RAW) rc = launch_project_raw(ctx, a, phi->ntiles, pl.model);
  else if (mode == PROJ_COLSUM) rc = launch_project_model<false>(ctx, a, phi->ntiles, pl.model, pl.ntsel);
  else rc = launch_project_model<true>(ctx, a, phi->ntiles, pl.model, pl.ntsel);
  if (!rc) rc = bc_timer_end(ctx, 1);
  return rc;
}

static int project_check(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                         const double* params, int32_t n_params, const char* who) {
  if (!ctx || !data || !theta || s <= 0 || (n_params > 0 && !params)) { bc_set_error("%s: bad argument", who); return BC_INVALID_ARGUMENT; }
  if (data->ctx != ctx) { bc_set_error("%s: data belongs to another context", who); return BC_INVALID_ARGUMENT; }
  if (model < 0 || model > BC_MODEL_GAUSS_BETA_GRAD) { bc_set_error("%s: unknown model %d", who, model); return BC_INVALID_ARGUMENT; }
  return BC_OK;
}

// raw == false: the whole of Phi (s == s_total <= 256).  raw == true: samples [s_off, s_off + s) of s_total, un-centred.
static int project_impl(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                        const double* params, int32_t n_params, int64_t row_offset, bc_phi** inout,
                        int32_t s_total, int32_t s_off, bool raw) {
  int rc = project_check(ctx, data, model, theta, s, params, n_params, "bc_project");
  if (rc) return rc;
  if (!inout) { bc_set_error("bc_project: bad argument"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  ProjPlan pl;
  rc = plan_stage(ctx, model, theta, s, params, n_params, data->dz, raw, &pl);
  if (rc) return rc;
  // output handle: reuse buffers when the shape matches
  bc_phi* phi = *inout;
  if (phi && (phi->ctx != ctx || phi->s != s_total || !phi->tiles || bc_phi_set_rows(phi, data->n_rows) != 0)) {
    bc_set_error("bc_project: *inout has a different S or too little row capacity; pass NULL to allocate");
    return BC_INVALID_ARGUMENT;
  }
  bool fresh = false;
  if (!phi) {
    rc = bc_phi_alloc(ctx, data->n_rows, s_total, row_offset, &phi);
    if (rc) return rc;
    fresh = true;
  }
  phi->row_offset = row_offset;
  phi->stats_valid = false;
  rc = plan_launch(ctx, pl, data, phi, raw ? PROJ_RAW : PROJ_FULL, s_total, s_off, &ctx->proj_rowaux);
  if (!rc && !raw) rc = bc_phi_finish_stats(phi);
  if (rc) { if (fresh) bc_phi_destroy(phi); return rc; }
  *inout = phi;
  return BC_OK;
}

// S > 256: passes of <= 256 samples write un-centred values, then one centring pass over Phi.
static int bc_project_wide(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                           const double* params, int32_t n_params, int64_t row_offset, bc_phi** inout) {
  const int d = data->dz - (model_has_y(model) ? 1 : 0);
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
  int rc = project_check(ctx, data, model, theta, s, params, n_params, "bc_project");
  if (rc) return rc;
  if (!inout) { bc_set_error("bc_project: bad argument"); return BC_INVALID_ARGUMENT; }
  if (s > 256) return bc_project_wide(ctx, data, model, theta, s, params, n_params, row_offset, inout);
  return project_impl(ctx, data, model, theta, s, params, n_params, row_offset, inout, s, 0, false);
}

// ---------------------------------------------------------------------------------------------
// Host rows -> Phi, pipelined (hilbert.py:11-17 and bcores.py:44 hand the projector HOST arrays): the rows are uploaded in
// chunks (bc_upload.hip) and K1 runs on chunk c behind the event that marks its arrival while chunks c+1.. are still on
// the wire.  Rows are independent, so Phi and the norms are the resident path's bit for bit; the column sums too, because
// ---- constant rows with the CALLER's bits (hosts whose NumPy does not evaluate np.exp with the SVML routine bc_np_exp.h
// restates: the reference's own bits for a constant row's value are then NumPy's on THAT host).
// A data row with all-zero features projects to S equal values that depend on its y alone (model_neurlinr.py:102-110 with
// x = 0); the host layer finds those rows (bc_data_zero_feature_keys), evaluates the reference's expression for their y's
// with its own NumPy and hands (y, value) pairs over; K1's constant-row branch then takes the value from here.
__global__ __launch_bounds__(256) void k_zero_feature_keys(const double* __restrict__ z, long long n_rows, int dz, int d,
                                                          double* __restrict__ out, unsigned long long* __restrict__ count,
                                                          long long cap) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const double* row = z + (size_t)r * dz;
  for (int k = 0; k < d; ++k)
    if (row[k] != 0.) return;                   // (almost every row leaves at k = 0)
  const unsigned long long slot = atomicAdd(count, 1ull);
  if ((long long)slot < cap) out[slot] = row[d];
}

extern "C" int bc_data_zero_feature_keys(const bc_data* data, int32_t d, int64_t cap, double* out_keys, int64_t* out_n) {
  if (!data || !out_n || d <= 0 || d >= data->dz || cap < 0 || (cap > 0 && !out_keys)) {
    bc_set_error("bc_data_zero_feature_keys: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  bc_ctx* ctx = data->ctx;
  *out_n = 0;
  if (data->n_rows == 0) return BC_OK;
  BC_HIP(hipSetDevice(ctx->device));
  int rc = bc_scratch_grow(ctx, &ctx->proj_rowaux2, (size_t)cap + 2);
  if (rc) return rc;
  double* buf = ctx->proj_rowaux2.p;
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(buf + cap);
  BC_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(k_zero_feature_keys, dim3((unsigned)((data->n_rows + 255) / 256)), dim3(256), 0, ctx->stream, data->z,
                     (long long)data->n_rows, data->dz, d, buf, cnt, (long long)cap);
  BC_HIP(hipGetLastError());
  unsigned long long n = 0;
  BC_HIP(hipMemcpyAsync(&n, cnt, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  *out_n = (int64_t)n;                          // may exceed cap: the caller then knows the list is truncated
  const int64_t take = (int64_t)n < cap ? (int64_t)n : cap;
  if (take > 0) {
    BC_HIP(hipMemcpyAsync(out_keys, buf, (size_t)take * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    BC_HIP(hipStreamSynchronize(ctx->stream));
  }
  return BC_OK;
}

extern "C" int bc_ctx_set_constant_row_values(bc_ctx* ctx, int model, const double* params, int32_t n_params, const double* keys,
                                              const double* values, int64_t n) {
  if (!ctx || n < 0 || (n > 0 && (!keys || !values || !params)) || n_params < 0 || n_params > 4) {
    bc_set_error("bc_ctx_set_constant_row_values: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  if (n == 0) { ctx->n_const_rows = 0; ctx->const_model = -1; return BC_OK; }
  if (model != BC_MODEL_LINREG_BETA) {
    bc_set_error("bc_ctx_set_constant_row_values: only the linear-regression beta-likelihood takes host-evaluated constants");
    return BC_INVALID_ARGUMENT;
  }
  for (int64_t i = 1; i < n; ++i)
    if (!(keys[i - 1] < keys[i])) { bc_set_error("bc_ctx_set_constant_row_values: keys must be strictly increasing"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  BC_HIP(hipStreamSynchronize(ctx->stream));    // no launch in flight may still read the old table
  int rc = bc_scratch_grow(ctx, &ctx->const_rows, (size_t)(2 * n));
  if (rc) return rc;
  BC_HIP(hipMemcpyAsync(ctx->const_rows.p, keys, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  BC_HIP(hipMemcpyAsync(ctx->const_rows.p + n, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));    // the host arrays are only borrowed for the call
  ctx->n_const_rows = n;
  ctx->const_model = model;
  ctx->const_n_params = n_params;
  memcpy(ctx->const_params, params, (size_t)n_params * sizeof(double));
  return BC_OK;
}

// the chunks keep the partial sums' association: the staged kernel writes one partial row per 128-row tile (chunks are
// tile-aligned), the Theta-resident kernel's per-wave partials are CONTINUED from chunk to chunk (ProjArgs::part_init;
// chunks start at multiples of 8 * n_cu groups = 65 536 rows, so every wave adds the same groups in the same order as in
// one launch over all rows).  Which of the two kernels runs is decided once, from the total row count, exactly as
// bc_project decides it.  *out_data receives the resident rows (the caller keeps or destroys them).
static int launch_chunk(bc_ctx* ctx, const ProjPlan& pl, const bc_data* data, bc_phi* phi, int64_t row0, int64_t rows, int rgrid,
                        bool first, bc_scratch* rowaux) {
  ProjArgs a = pl.a;
  a.rowaux = nullptr;
  if (pl.siginv_dev) {
    const int d = pl.d;
    const int use_lds = (size_t)d * d * sizeof(double) <= 60 * 1024;
    long long blocks = (rows + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_row_quadform, dim3((unsigned)blocks), dim3(256), use_lds ? (size_t)d * d * sizeof(double) : 0,
                       ctx->stream, data->z + (size_t)row0 * data->dz, (long long)rows, d, pl.siginv_dev, rowaux->p + row0, use_lds);
    BC_HIP(hipGetLastError());
    a.rowaux = rowaux->p + row0;
  }
  const int64_t tile0 = row0 / BC_TILE, ntiles = (rows + BC_TILE - 1) / BC_TILE;
  a.z = data->z + (size_t)row0 * data->dz;
  a.tiles = phi->tiles + (size_t)tile0 * phi->s * BC_TILE;
  a.norms = phi->norms + row0;
  a.n_rows = rows;
  a.dz = data->dz;
  a.s_total = phi->s;
  a.s_off = 0;
  // timer class 1 (K1), one span per chunk launch: recorded on ctx->stream behind the wait for the chunk's arrival, so the
  // span is the kernel, not the transfer it waited for
  int rc = bc_timer_begin(ctx, 1);
  if (rc) return rc;
  if (rgrid > 0) {
    a.tile_part = phi->tile_part;                 // one row per wave, shared by all chunks
    a.part_init = first ? 0 : 1;
    a.ngroups = (rows + 31) / 32;
    rc = launch_project_r_model<true>(ctx, a, rgrid, pl.model, pl.ntsel);
  } else {
    a.tile_part = phi->tile_part + (size_t)tile0 * phi->s;
    rc = launch_project_model<true>(ctx, a, ntiles, pl.model, pl.ntsel);
  }
  if (!rc) rc = bc_timer_end(ctx, 1);
  return rc;
}

extern "C" int bc_project_from_host(bc_ctx* ctx, const double* z_host, int64_t n_rows, int32_t dz, int model, const double* theta,
                                    int32_t s, const double* params, int32_t n_params, int64_t row_offset, bc_data** out_data,
                                    bc_phi** inout) {
  if (!ctx || !z_host || n_rows <= 0 || dz <= 0 || !theta || s <= 0 || !out_data || !inout || (n_params > 0 && !params)) {
    bc_set_error("bc_project_from_host: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  if (model < 0 || model > BC_MODEL_GAUSS_BETA_GRAD) { bc_set_error("bc_project_from_host: unknown model %d", model); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  bc_data* data = new bc_data();
  data->ctx = ctx;
  data->n_rows = n_rows;
  data->dz = dz;
  data->cap_rows = n_rows;
  {
    hipError_t e = hipMalloc((void**)&data->z, (size_t)n_rows * dz * sizeof(double));
    if (e != hipSuccess) { delete data; return bc_hip_fail(e, "hipMalloc(data)", __FILE__, __LINE__); }
  }
  auto drop_data = [&]() { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(data->z); delete data; };
  int rc;
  if (s > 256) {
    // the wide path makes several passes over all rows: plain (still multi-threaded) upload, then bc_project
    rc = bc_upload_rows(ctx, z_host, data->z, n_rows, dz, bc_upload_default_chunk_rows(n_rows, dz), nullptr);
    if (!rc) rc = bc_project_wide(ctx, data, model, theta, s, params, n_params, row_offset, inout);
    if (rc) { drop_data(); return rc; }
    *out_data = data;
    return BC_OK;
  }
  ProjPlan pl;
  rc = plan_stage(ctx, model, theta, s, params, n_params, dz, false, &pl);
  if (rc) { drop_data(); return rc; }
  bc_phi* phi = *inout;
  if (phi && (phi->ctx != ctx || phi->s != s || !phi->tiles || bc_phi_set_rows(phi, n_rows) != 0)) {
    bc_set_error("bc_project_from_host: *inout has a different S or too little row capacity; pass NULL to allocate");
    drop_data();
    return BC_INVALID_ARGUMENT;
  }
  bool fresh = false;
  if (!phi) {
    rc = bc_phi_alloc(ctx, n_rows, s, row_offset, &phi);
    if (rc) { drop_data(); return rc; }
    fresh = true;
  }
  phi->row_offset = row_offset;
  phi->stats_valid = false;
  if (pl.siginv_dev) {
    rc = bc_scratch_grow(ctx, &ctx->proj_rowaux, (size_t)n_rows);
    if (rc) { if (fresh) bc_phi_destroy(phi); drop_data(); return rc; }
  }
  const int rgrid = project_r_grid(ctx, pl, phi, PROJ_FULL);
  phi->part_rows = rgrid > 0 ? (int64_t)rgrid * 8 : phi->ntiles;
  // chunk = a multiple of (8 * rgrid) 32-row groups for the resident kernel, of 128-row tiles for the staged one; ~128 MiB
  const int64_t unit = bc_lay_chunk_unit(rgrid);
  const char* env = getenv("BC_PIPE_CHUNK_ROWS");      // tests: force several chunks on small inputs
  const int64_t chunk_rows = bc_lay_chunk_rows(dz, unit, env ? atoll(env) : 0);
  bool first = true;
  bc_chunk_hook hook = [&](int64_t, int64_t row0, int64_t rows, hipEvent_t landed) -> int {
    if (landed) BC_HIP(hipStreamWaitEvent(ctx->stream, landed, 0));
    const int r = launch_chunk(ctx, pl, data, phi, row0, rows, rgrid, first, &ctx->proj_rowaux);
    first = false;
    return r;
  };
  rc = bc_upload_rows(ctx, z_host, data->z, n_rows, dz, chunk_rows, &hook);
  if (!rc) rc = bc_phi_finish_stats(phi);
  if (rc) { if (fresh) { (void)hipStreamSynchronize(ctx->stream); bc_phi_destroy(phi); } drop_data(); return rc; }
  *inout = phi;
  *out_data = data;
  return BC_OK;
}

// ---------------------------------------------------------------------------------------------
// Store-free projection and the fused gradient of the greedy-VI weight optimisation.
//
// bcores.py:141-146 / sparsevi.py:129-134: every one of the opt_itrs ADAM steps of every build step evaluates
//     vecs = project_f(data, beta) ; resid = sum_scaling * vecs.sum(axis=0) - w.dot(corevecs) ; grad = -corevecs.dot(resid) / S
// i.e. of the N x S projection only its S column sums are used.  The materialising path writes 8*N*S bytes to read S
// numbers back; here K1 keeps its per-tile column partials and stores nothing else (k_project<..., STORE = false>).
int bc_comm_sum_dev(bc_comm* c, const double* in_dev, int64_t count, const double** result_dev);   // bc_comm.hip
bc_ctx* bc_comm_ctx(const bc_comm* c);

// the context's stats-only Phi, sized for n_rows x s
static int colsum_phi_for(bc_ctx* ctx, int64_t n_rows, int32_t s, bc_phi** out) {
  bc_phi* p = ctx->colsum_phi;
  if (p && (p->s != s || bc_phi_set_rows(p, n_rows) != 0)) {
    BC_HIP(hipStreamSynchronize(ctx->stream));
    bc_phi_destroy(p);
    ctx->colsum_phi = p = nullptr;
  }
  if (!p) {
    int rc = bc_phi_alloc(ctx, n_rows, s, 0, &p, 0, true);
    if (rc) return rc;
    ctx->colsum_phi = p;
  }
  p->stats_valid = false;
  *out = p;
  return BC_OK;
}

extern "C" int bc_project_colsum(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                                 const double* params, int32_t n_params, bc_comm* comm, double* out_s) {
  int rc = project_check(ctx, data, model, theta, s, params, n_params, "bc_project_colsum");
  if (rc) return rc;
  if (!out_s) { bc_set_error("bc_project_colsum: bad argument"); return BC_INVALID_ARGUMENT; }
  if (comm && bc_comm_ctx(comm) != ctx) { bc_set_error("bc_project_colsum: the communicator belongs to another context"); return BC_INVALID_ARGUMENT; }
  if (s > 256) { bc_set_error("bc_project_colsum: at most 256 samples (S = %d): project and take bc_phi_colsum", s); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  ProjPlan pl;
  rc = plan_stage(ctx, model, theta, s, params, n_params, data->dz, false, &pl);
  if (rc) return rc;
  bc_phi* phi = nullptr;
  rc = colsum_phi_for(ctx, data->n_rows, s, &phi);
  if (!rc) rc = plan_launch(ctx, pl, data, phi, PROJ_COLSUM, s, 0, &ctx->proj_rowaux);
  if (!rc) rc = bc_phi_reduce_colsum(phi);
  if (rc) return rc;
  const double* res = phi->colsum;
  if (comm) {
    rc = bc_comm_sum_dev(comm, phi->colsum, s, &res);
    if (rc) return rc;
  }
  BC_HIP(hipMemcpyAsync(ctx->pinned, res, (size_t)s * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(out_s, ctx->pinned, (size_t)s * sizeof(double));
  return BC_OK;
}

// resid[k] = scale * colsum[k] - sum_i w[i] * C[i, k]   (bcores.py:145: sum_scaling*vecs.sum(axis=0) - w.dot(corevecs))
// grad[i]  = -(sum_k C[i, k] * resid[k]) / S            (bcores.py:146: -corevecs.dot(resid)/corevecs.shape[1])
// C = the projected coreset rows in the tiled layout (row i, sample k at bc_tile_off(i, k, S)): neighbouring threads
// read neighbouring rows, so both passes are coalesced.  One block; m is the coreset size (tens to a few thousand).
__global__ __launch_bounds__(256) void k_vi_gradient(const double* __restrict__ colsum, const double* __restrict__ core,
                                                    const double* __restrict__ w, int m, int s, double scale,
                                                    double* __restrict__ resid_out, double* __restrict__ grad_out) {
  extern __shared__ double rs[];      // [s]
  for (int k = threadIdx.x; k < s; k += blockDim.x) {
    double acc = 0.;
    int i = 0;
    for (; i + 8 <= m; i += 8) {           // same fma chain, eight loads in flight (one block, pure latency)
      double c[8], ww[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { c[u] = core[bc_tile_off(i + u, k, s)]; ww[u] = w[i + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fma(ww[u], c[u], acc);
    }
    for (; i < m; ++i) acc = fma(w[i], core[bc_tile_off(i, k, s)], acc);
    const double r = scale * colsum[k] - acc;
    rs[k] = r;
    resid_out[k] = r;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    double acc = 0.;
    int k = 0;
    for (; k + 8 <= s; k += 8) {
      double c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = core[bc_tile_off(i, k + u, s)];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fma(c[u], rs[k + u], acc);
    }
    for (; k < s; ++k) acc = fma(core[bc_tile_off(i, k, s)], rs[k], acc);
    grad_out[i] = -acc / (double)s;
  }
}

extern "C" int bc_vi_gradient_begin(bc_ctx* ctx, const bc_data* data, const double* core_rows, int64_t m, int model,
                                    const double* theta, int32_t s, const double* params, int32_t n_params, const double* w,
                                    double sum_scaling, bc_comm* comm) {
  int rc = project_check(ctx, data, model, theta, s, params, n_params, "bc_vi_gradient");
  if (rc) return rc;
  if (m <= 0 || !core_rows || !w) { bc_set_error("bc_vi_gradient: needs a non-empty coreset (m = %lld)", (long long)m); return BC_INVALID_ARGUMENT; }
  if (s > 256) { bc_set_error("bc_vi_gradient: at most 256 samples (S = %d)", s); return BC_INVALID_ARGUMENT; }
  if (comm && bc_comm_ctx(comm) != ctx) { bc_set_error("bc_vi_gradient: the communicator belongs to another context"); return BC_INVALID_ARGUMENT; }
  const int dz = data->dz;
  const size_t n_core = (size_t)m * dz, n_down = (size_t)m + (size_t)s;
  if (ctx->vi_pending_m > 0) { bc_set_error("bc_vi_gradient_begin: a gradient is already pending on this context (call bc_vi_gradient_end first)"); return BC_INVALID_ARGUMENT; }
  if (n_down > ctx->pinned_doubles) {
    bc_set_error("bc_vi_gradient: coreset of %lld rows x %d exceeds the staging area", (long long)m, dz);
    return BC_INVALID_ARGUMENT;
  }
  BC_HIP(hipSetDevice(ctx->device));
  const bool timed = ctx->timing != 0;
  if (timed)
    for (auto& ev : ctx->vi_ev)
      if (!ev) BC_HIP(hipEventCreate(&ev));
  auto mark = [&](int i) -> int {
    if (timed) BC_HIP(hipEventRecord(ctx->vi_ev[i], ctx->stream));
    return BC_OK;
  };
  // --- stage Theta, the coreset rows and w: one pinned area, one transfer (plan_stage synchronises with the stream first)
  ProjPlan pl;
  rc = mark(0);
  if (rc) return rc;
  const double* esrc[2] = {core_rows, w};
  const size_t en[2] = {n_core, (size_t)m};
  double* edev[2] = {nullptr, nullptr};
  rc = plan_stage(ctx, model, theta, s, params, n_params, dz, false, &pl, 2, esrc, en, edev);
  if (rc) return rc;
  bc_data core_view;                     // the coreset rows where the transfer put them (borrowed)
  core_view.ctx = ctx;
  core_view.n_rows = m;
  core_view.dz = dz;
  core_view.z = edev[0];
  core_view.owned = false;
  bc_data* cd = &core_view;
  const double* d_w = edev[1];
  rc = bc_scratch_grow(ctx, &ctx->vi_buf, (size_t)m + (size_t)s);
  if (rc) return rc;
  double* d_grad = ctx->vi_buf.p;
  double* d_resid = d_grad + m;
  rc = mark(1);
  if (rc) return rc;
  // --- the <= M coreset rows (materialised: the M x S algebra below reads them), then the data rows (store-free)
  bc_phi* cphi = ctx->core_phi;
  if (cphi && (cphi->s != s || bc_phi_set_rows(cphi, m) != 0)) {
    BC_HIP(hipStreamSynchronize(ctx->stream));
    bc_phi_destroy(cphi);
    ctx->core_phi = cphi = nullptr;
  }
  if (!cphi) {
    int64_t cap = 256;
    while (cap < m) cap *= 2;
    rc = bc_phi_alloc(ctx, m, s, 0, &cphi, cap);
    if (rc) return rc;
    ctx->core_phi = cphi;
  }
  cphi->stats_valid = false;
  // The coreset rows' launch (one tile or a few: ~20 us of dependent latency, no work to speak of) runs on a side stream
  // BESIDE the data rows' launch instead of in front of it; the algebra kernel waits for both (0.441 -> 0.427 ms per native
  // call at N = 1M, D = 64).  The instrumented pass (timing on) keeps everything on one stream, so that its five phases stay a
  // partition of the call.
  const bool beside = !timed;
  if (beside && !ctx->vi_side) {
    BC_HIP(hipStreamCreateWithFlags(&ctx->vi_side, hipStreamNonBlocking));
    BC_HIP(hipEventCreateWithFlags(&ctx->vi_ev_staged, hipEventDisableTiming));
    BC_HIP(hipEventCreateWithFlags(&ctx->vi_ev_core, hipEventDisableTiming));
  }
  const int saved_timing = ctx->timing;
  ctx->timing = 0;                      // the coreset rows' launch is not a K1 sample of the kernel timer
  if (beside) {
    BC_HIP(hipEventRecord(ctx->vi_ev_staged, ctx->stream));             // Theta, the coreset rows and w have landed
    BC_HIP(hipStreamWaitEvent(ctx->vi_side, ctx->vi_ev_staged, 0));
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->vi_side;
    rc = plan_launch(ctx, pl, cd, cphi, PROJ_FULL, s, 0, &ctx->proj_rowaux);
    ctx->stream = main_stream;
    if (!rc) {
      const hipError_t e = hipEventRecord(ctx->vi_ev_core, ctx->vi_side);
      if (e != hipSuccess) rc = bc_hip_fail(e, "hipEventRecord(vi_ev_core)", __FILE__, __LINE__);
    }
  } else {
    rc = plan_launch(ctx, pl, cd, cphi, PROJ_FULL, s, 0, &ctx->proj_rowaux);
  }
  ctx->timing = saved_timing;
  // Everything after the side launch runs inside `rest`: on ANY failure in it the side stream is joined before the error
  // leaves this function -- the next call's plan_stage / bc_scratch_grow synchronise ctx->stream only and would otherwise
  // overwrite proj_theta / free core_phi under a coreset-row kernel that is still reading them.
  auto rest = [&]() -> int {
    int rc = mark(2);
    bc_phi* phi = nullptr;
    if (!rc) rc = colsum_phi_for(ctx, data->n_rows, s, &phi);
    // (x^T Siginv x of the data rows, Gaussian models, goes to a scratch of its own: the coreset rows' is still in use)
    if (!rc) rc = plan_launch(ctx, pl, data, phi, PROJ_COLSUM, s, 0, &ctx->proj_rowaux2);
    if (!rc) rc = mark(3);
    if (!rc) rc = bc_phi_reduce_colsum(phi);
    if (rc) return rc;
    const double* colsum = phi->colsum;
    if (comm) {
      rc = bc_comm_sum_dev(comm, phi->colsum, s, &colsum);
      if (rc) return rc;
    }
    rc = mark(4);
    if (rc) return rc;
    if (beside) BC_HIP(hipStreamWaitEvent(ctx->stream, ctx->vi_ev_core, 0));
    hipLaunchKernelGGL(k_vi_gradient, dim3(1), dim3(256), (size_t)s * sizeof(double), ctx->stream, colsum, cphi->tiles, d_w,
                       (int)m, s, sum_scaling, d_resid, d_grad);
    BC_HIP(hipGetLastError());
    // the result lands in a pinned area of its own: whatever the host does between _begin and _end (it may well call into this
    // library, whose other entry points stage through ctx->pinned) cannot overwrite it
    if (!ctx->vi_pinned) BC_HIP(hipHostMalloc((void**)&ctx->vi_pinned, ctx->pinned_doubles * sizeof(double), hipHostMallocDefault));
    BC_HIP(hipMemcpyAsync(ctx->vi_pinned, d_grad, n_down * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return mark(5);
  };
  if (!rc) rc = rest();
  if (rc) {
    if (beside) (void)hipStreamSynchronize(ctx->vi_side);
    return rc;
  }
  ctx->vi_pending_m = m;
  ctx->vi_pending_s = s;
  ctx->vi_pending_timed = timed;
  return BC_OK;
}

// second half: wait for the enqueued gradient and hand it out (whatever the host did meanwhile -- e.g. drawing the next
// sample matrix's normals -- ran beside the GPU)
extern "C" int bc_vi_gradient_end(bc_ctx* ctx, double* out_grad, double* out_resid) {
  if (!ctx || !out_grad) { bc_set_error("bc_vi_gradient_end: bad argument"); return BC_INVALID_ARGUMENT; }
  if (ctx->vi_pending_m <= 0) { bc_set_error("bc_vi_gradient_end: no gradient is pending on this context"); return BC_INVALID_ARGUMENT; }
  const int64_t m = ctx->vi_pending_m;
  const int32_t s = ctx->vi_pending_s;
  const bool timed = ctx->vi_pending_timed;
  ctx->vi_pending_m = 0;
  BC_HIP(hipSetDevice(ctx->device));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(out_grad, ctx->vi_pinned, (size_t)m * sizeof(double));
  if (out_resid) memcpy(out_resid, ctx->vi_pinned + m, (size_t)s * sizeof(double));
  if (timed) {
    for (int i = 0; i < BC_VI_PHASES; ++i) {
      float ms = 0.f;
      BC_HIP(hipEventElapsedTime(&ms, ctx->vi_ev[i], ctx->vi_ev[i + 1]));
      ctx->vi_phase_ms[i] += ms;
    }
    ctx->vi_calls_timed++;
  }
  return BC_OK;
}

extern "C" int bc_vi_gradient(bc_ctx* ctx, const bc_data* data, const double* core_rows, int64_t m, int model,
                              const double* theta, int32_t s, const double* params, int32_t n_params, const double* w,
                              double sum_scaling, bc_comm* comm, double* out_grad, double* out_resid) {
  if (!out_grad) { bc_set_error("bc_vi_gradient: bad argument"); return BC_INVALID_ARGUMENT; }
  int rc = bc_vi_gradient_begin(ctx, data, core_rows, m, model, theta, s, params, n_params, w, sum_scaling, comm);
  if (rc) return rc;
  return bc_vi_gradient_end(ctx, out_grad, out_resid);
}

extern "C" int bc_ctx_phase_times(bc_ctx* ctx, double* out_ms, int32_t n, int64_t* calls, int reset) {
  if (!ctx || (n > 0 && !out_ms)) { bc_set_error("bc_ctx_phase_times: bad argument"); return BC_INVALID_ARGUMENT; }
  for (int i = 0; i < n; ++i) out_ms[i] = i < BC_VI_PHASES ? ctx->vi_phase_ms[i] : 0.;
  if (calls) *calls = ctx->vi_calls_timed;
  if (reset) {
    for (auto& v : ctx->vi_phase_ms) v = 0.;
    ctx->vi_calls_timed = 0;
  }
  return BC_OK;
}
