// K4: weighted Gram matrix  Z^T diag(w) Z  of the data rows z = [x (D), y], from which the
// posterior update takes  X^T W X  (D x D) and  X^T (w*y)  (D)   (model_linreg.py:29,31 ==
// model_neurlinr.py:118,120:  (w[:,None]*X).T.dot(X)  and  (w[:,None]*Y[:,None]*X).sum(axis=0)).
//
// This is the one genuinely GEMM-shaped reduction on the path (2*N*Dz^2 flop against 8*N*Dz
// bytes: intensity Dz/4 flop/B), so it runs on the fp64 matrix cores: v_mfma_f64_16x16x4_f64
// with A = (w*Z)^T panel, B = Z panel, both staged through LDS in 16-row slabs (range-checked
// buffer loads, register prefetch of the next slab).  Only the upper-triangular BT x BT tiles are
// computed, and of a DIAGONAL tile only the MFMA sub-tiles on or above its diagonal (GramDiag);
// the row range is split over the grid (split-K) and the per-split partial tiles are summed in
// split order by two small kernels (k_gram_reduce1 / 2), so the result is run-to-run
// deterministic.  Coreset-sized inputs (the samplers, once per gradient) take one launch,
// k_gram_small, with no partials at all.
#include "bc_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

struct GramArgs {
  const double* z;        // [n_rows][dz]
  const double* w;        // [n_rows] or null (all ones)
  double* partial;        // [splits][ntri][BT][BT]
  double* partial_y;      // [splits][nt*BT]   X^T (w*y), accumulated by the diagonal-tile blocks
  long long n_rows;
  long long rows_per_split;   // multiple of KR
  long long splits;
  int dz, d, nt;          // d = number of x columns (y is column d), nt = number of BT-wide column tiles over d
  int ntri;               // nt*(nt+1)/2
};

// A DIAGONAL tile (ta == tb) is symmetric itself: of its (BT/16)^2 MFMA sub-tiles only those on or above the diagonal are
// needed -- 36 of 64 at BT = 128, 10 of 16 at BT = 64.  They are dealt to the four waves 9 / 9 / 9 / 9 (3 / 2 / 2 / 3) so that each
// wave's set touches few distinct 16-column strips of the two LDS panels (8 or 7 ds_reads per k-step, as many as the square
// wave tile needs for 16 MFMAs).  At D <= 128 the whole Gram matrix is ONE diagonal tile: 9 MFMAs per wave and k-step instead of 16.
template <int BT, int W> struct GramDiag;
template <> struct GramDiag<128, 0> { static constexpr int n = 9; static constexpr int x[9] = {0, 0, 0, 1, 1, 1, 2, 2, 3}; static constexpr int y[9] = {0, 1, 2, 1, 2, 3, 2, 3, 3}; };
template <> struct GramDiag<128, 1> { static constexpr int n = 9; static constexpr int x[9] = {0, 0, 0, 0, 0, 1, 1, 1, 1}; static constexpr int y[9] = {3, 4, 5, 6, 7, 4, 5, 6, 7}; };
template <> struct GramDiag<128, 2> { static constexpr int n = 9; static constexpr int x[9] = {2, 2, 2, 2, 3, 3, 3, 3, 4}; static constexpr int y[9] = {4, 5, 6, 7, 4, 5, 6, 7, 7}; };
template <> struct GramDiag<128, 3> { static constexpr int n = 9; static constexpr int x[9] = {4, 4, 4, 5, 5, 5, 6, 6, 7}; static constexpr int y[9] = {4, 5, 6, 5, 6, 7, 6, 7, 7}; };
template <> struct GramDiag<64, 0> { static constexpr int n = 3; static constexpr int x[3] = {0, 0, 1}; static constexpr int y[3] = {0, 1, 1}; };
template <> struct GramDiag<64, 1> { static constexpr int n = 2; static constexpr int x[3] = {0, 0, 0}; static constexpr int y[3] = {2, 3, 3}; };
template <> struct GramDiag<64, 2> { static constexpr int n = 2; static constexpr int x[3] = {1, 1, 1}; static constexpr int y[3] = {2, 3, 3}; };
template <> struct GramDiag<64, 3> { static constexpr int n = 3; static constexpr int x[3] = {2, 2, 3}; static constexpr int y[3] = {2, 3, 3}; };

// one k-step (4 rows of the slab) of wave W's share of a diagonal tile, in two halves so that the fragments of the NEXT
// k-step can be requested from LDS before the MFMAs of the current one are issued; acc[i] belongs to sub-tile (x[i], y[i])
template <int BT, int W>
__device__ __forceinline__ void gram_diag_frags(const double* __restrict__ Arow, const double* __restrict__ Brow, int j,
                                                double (&fa)[BT / 16], double (&fb)[BT / 16]) {
  using M = GramDiag<BT, W>;
  constexpr int NS = BT / 16;
  unsigned mx = 0, my = 0;
#pragma unroll
  for (int i = 0; i < M::n; ++i) { mx |= 1u << M::x[i]; my |= 1u << M::y[i]; }
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    fa[c] = 0.;
    fb[c] = 0.;
    if ((mx >> c) & 1u) fa[c] = Arow[c * 16 + j];
    if ((my >> c) & 1u) fb[c] = Brow[c * 16 + j];
  }
}
template <int BT, int W, int NA>
__device__ __forceinline__ void gram_diag_mma(const double (&fa)[BT / 16], const double (&fb)[BT / 16], double4_t (&acc)[NA]) {
  using M = GramDiag<BT, W>;
#pragma unroll
  for (int i = 0; i < M::n; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[M::x[i]], fb[M::y[i]], acc[i], 0, 0, 0);
}

template <int BT, int W, int NA>
__device__ __forceinline__ void gram_diag_store(double* __restrict__ out, int g, int j, const double4_t (&acc)[NA]) {
  using M = GramDiag<BT, W>;
#pragma unroll
  for (int i = 0; i < M::n; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) out[(size_t)(M::x[i] * 16 + g + 4 * reg) * BT + M::y[i] * 16 + j] = acc[i][reg];
}

__device__ const double g_gram_ones[16] = {1., 1., 1., 1., 1., 1., 1., 1., 1., 1., 1., 1., 1., 1., 1., 1.};

// One (split, tile) of the Gram matrix.  V = -1: off-diagonal tile, every wave a square BT/2 x BT/2 wave tile; V = 0..3: diagonal
// tile, wave V's share of its upper triangle (GramDiag).  The variant is block- (V < 0) or wave-uniform, chosen by a scalar
// branch in k_gram; every variant passes the same barriers.
template <int BT, int V>
__device__ __forceinline__ void gram_tile(const GramArgs& a, int ta, int tb, long long split, int tri, double* __restrict__ Al,
                                          double* __restrict__ Bl, double* __restrict__ Yl) {
  constexpr int KR = BT == 64 ? 32 : 16; // rows per LDS slab (BT = 64: three MFMAs per wave and k-step at most -- twice the rows per barrier pair)
  constexpr int LDX = BT + 16;           // row stride == 16 (mod 32) doubles: conflict-free 2-row x 16-col reads
  constexpr int MT = BT / 32;            // MFMA tiles per wave per dimension (wave tile = BT/2 x BT/2)
  constexpr int LP = (KR * BT) / 256;    // 8-byte loads per thread per panel per slab
  constexpr bool diag = V >= 0;          // the diagonal block of a column panel also owns its X^T (w*y) slice
  constexpr int NA = diag ? GramDiag<BT, diag ? V : 0>::n : MT * MT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int wa = V < 0 ? (tid >> 7) : 0, wb = V < 0 ? ((tid >> 6) & 1) : 0;   // wave position inside an off-diagonal block tile
  const int a0 = ta * BT, b0 = tb * BT;
  const long long r_begin = split * a.rows_per_split;
  long long r_end = r_begin + a.rows_per_split;
  if (r_end > a.n_rows) r_end = a.n_rows;

  double4_t acc[NA];
#pragma unroll
  for (int x = 0; x < NA; ++x) acc[x] = (double4_t){0., 0., 0., 0.};

  // staging map: thread -> (row lr = idx / BT, col lc = idx % BT) of the slab, LP passes
  const int lc = tid % BT, lr0 = tid / BT;
  constexpr int RPP = 256 / BT;          // rows covered per pass
  // Every VALU instruction of the staging code adds to the kernel time (an fp64 MFMA wave owns its SIMD's issue,
  // profiles/r02_notes.md), so nothing is selected or compared per element: columns past d (the zero padding of the last
  // tile) are read through an offset the descriptor's range check rejects, rows past the end likewise (z AND w: both
  // descriptors end at the last row), and a rejected load returns 0.
  constexpr int OOB = 0x40000000;
  const int oa = a0 + lc < a.d ? (lr0 * a.dz + a0 + lc) * 8 : OOB;
  const int ob = b0 + lc < a.d ? (lr0 * a.dz + b0 + lc) * 8 : OOB;
  const int oy = (lr0 * a.dz + a.d) * 8;
  const bool weighted = a.w != nullptr;
  double vy = 0.0;
  // raw values of the slab in flight; the weighting and the X^T (w*y) term are applied when the slab is parked in LDS,
  // so that the loads can be requested in slices spread over the previous slab's k-steps (requested in one go after the
  // barrier they held the wave in the issue stage before its first MFMA of the slab -- see K1, profiles/r02_notes.md)
  double va[LP], vb[LP], vw[LP], vyv[LP];
  constexpr int NSL = KR / 4;            // one slice per k-step
  static_assert(LP % NSL == 0 || LP < NSL, "slices");
  auto load_part = [&](long long r0, int part) {
    const long long left = r_end - r0;
    const int rows_here = left < KR ? (left > 0 ? (int)left : 0) : KR;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.z + (size_t)r0 * a.dz), 0, rows_here * a.dz * 8, 0x00020000);
    // (no weights: sixteen ones stand in for them -- a base pointer chosen by the scalar unit, no branch and no constant moves)
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)(weighted ? a.w + r0 : g_gram_ones), 0, rows_here * 8, 0x00020000);
#pragma unroll
    for (int q = part * LP / NSL; q < (part + 1) * LP / NSL; ++q) {
      vw[q] = 1.0;    // (BT = 64 is bound by its loads, not by the matrix pipe: no stand-in loads there)
      if ((BT == 128 && !(diag && !weighted)) || weighted) vw[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rw, (lr0 + q * RPP) * 8, 0, 0));
      va[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, oa + q * RPP * a.dz * 8, 0, 0));
      if (diag) {                          // both panels are the same columns: one load
        vyv[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, oy + q * RPP * a.dz * 8, 0, 0));
      } else {
        vb[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, ob + q * RPP * a.dz * 8, 0, 0));
      }
    }
  };
  // A diagonal tile WITHOUT weights (w = 1: the full-data posterior of the drivers' `sampler_optimal`) has A == B: one panel
  // is parked and both fragment sets are read from it -- eight ds_write, eight multiplications by one and eight stand-in
  // loads fewer per thread and slab, the same numbers (1.0 * x is x).  Block-uniform condition.
  const bool one_panel = diag && !weighted;
  auto store_slab = [&]() {
    if (one_panel) {
#pragma unroll
      for (int q = 0; q < LP; ++q) {
        Al[(lr0 + q * RPP) * LDX + lc] = va[q];
        vy = fma(va[q], vyv[q], vy);
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < LP; ++q) {
      const int lr = lr0 + q * RPP;
      const double raq = vw[q] * va[q];                  // A panel carries the weights: (w[:,None]*X)
      Al[lr * LDX + lc] = raq;
      Bl[lr * LDX + lc] = diag ? va[q] : vb[q];
      if (diag) vy = fma(raq, vyv[q], vy);               // (w[:,None]*Y[:,None]*X).sum(axis=0), model_linreg.py:31
    }
  };

  if (r_begin < r_end) {
#pragma unroll
    for (int part = 0; part < NSL; ++part) load_part(r_begin, part);
  }
  for (long long r0 = r_begin; r0 < r_end; r0 += KR) {
    store_slab();
    __syncthreads();
    const bool more = r0 + KR < r_end;
    // fragments of k-step kk + 1 are requested before the MFMAs of k-step kk are issued
    constexpr int NF = diag ? BT / 16 : MT;
    double fa[2][NF], fb[2][NF];
    auto frags = [&](int kk, double (&xa)[NF], double (&xb)[NF]) {
      const double* Arow = Al + (kk * 4 + g) * LDX;
      const double* Brow = (one_panel ? Al : Bl) + (kk * 4 + g) * LDX;
      if constexpr (diag) {
        gram_diag_frags<BT, V>(Arow, Brow, j, xa, xb);
      } else {
#pragma unroll
        for (int x = 0; x < MT; ++x) xa[x] = Arow[wa * (BT / 2) + x * 16 + j];
#pragma unroll
        for (int y = 0; y < MT; ++y) xb[y] = Brow[wb * (BT / 2) + y * 16 + j];
      }
    };
    frags(0, fa[0], fb[0]);
#pragma unroll
    for (int kk = 0; kk < KR / 4; ++kk) {
      if (more) load_part(r0 + KR, kk);
      if (kk + 1 < KR / 4) frags(kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
      if constexpr (diag) {
        gram_diag_mma<BT, V>(fa[kk & 1], fb[kk & 1], acc);
      } else {
#pragma unroll
        for (int x = 0; x < MT; ++x)
#pragma unroll
          for (int y = 0; y < MT; ++y)
            acc[x * MT + y] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk & 1][x], fb[kk & 1][y], acc[x * MT + y], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);     // keeps each slice of loads with its k-step
    }
    __syncthreads();
  }
  if (diag) {                            // combine the RPP thread-rows of each column in fixed order
    Yl[tid] = vy;
    __syncthreads();
    if (tid < BT) {
      double t = Yl[tid];
#pragma unroll
      for (int q = 1; q < RPP; ++q) t += Yl[q * BT + tid];
      a.partial_y[(size_t)split * a.nt * BT + ta * BT + tid] = t;
    }
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
  double* out = a.partial + ((size_t)split * a.ntri + tri) * BT * BT;
  if constexpr (diag) {                  // sub-tiles below the diagonal are neither computed nor written (nor read: k_gram_reduce*)
    gram_diag_store<BT, V>(out, g, j, acc);
  } else {
#pragma unroll
    for (int x = 0; x < MT; ++x)
#pragma unroll
      for (int y = 0; y < MT; ++y)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = wa * (BT / 2) + x * 16 + g + 4 * reg;
          const int col = wb * (BT / 2) + y * 16 + j;
          out[(size_t)row * BT + col] = acc[x * MT + y][reg];
        }
  }
}

template <int BT>
__global__ __launch_bounds__(256, BT == 64 ? 3 : 2) void k_gram(GramArgs a) {
  __shared__ double Al[(BT == 64 ? 32 : 16) * (BT + 16)];
  __shared__ double Bl[(BT == 64 ? 32 : 16) * (BT + 16)];
  __shared__ double Yl[256];
  // XCD-aware block -> (split, tile) map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share
  // one, each XCD has its own L2), and the ntri tiles of one row split all read the same rows: they are given block
  // numbers that are equal modulo 8 and adjacent in that XCD's queue, so the rows are fetched from HBM once per split
  // instead of once per tile (D = 512: ten tiles per split; with the tiles of a split on neighbouring block numbers,
  // i.e. on ten different XCDs, the kernel ran 13.5 ms instead of 12.85 at eight rounds of blocks, 19.8 instead of 18.6 at one).
  const long long q = blockIdx.x >> 3;
  const int tri = (int)(q % a.ntri);
  const long long split = (q / a.ntri) * 8 + (blockIdx.x & 7);
  if (split >= a.splits) return;
  int ta = 0, rem = tri;                 // tri -> (ta <= tb)
  while (rem >= a.nt - ta) { rem -= a.nt - ta; ++ta; }
  const int tb = ta + rem;
  if (ta != tb) { gram_tile<BT, -1>(a, ta, tb, split, tri, Al, Bl, Yl); return; }
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {     // (scalar branch: one variant per wave)
    case 0: gram_tile<BT, 0>(a, ta, tb, split, tri, Al, Bl, Yl); break;
    case 1: gram_tile<BT, 1>(a, ta, tb, split, tri, Al, Bl, Yl); break;
    case 2: gram_tile<BT, 2>(a, ta, tb, split, tri, Al, Bl, Yl); break;
    default: gram_tile<BT, 3>(a, ta, tb, split, tri, Al, Bl, Yl); break;
  }
}

// ---------------------------------------------------------------------------------------------
// D <= 128 without weights (the drivers' full-data posterior, `sampler_optimal`: ONE diagonal tile, A == B == the raw rows):
// the slab goes from global memory STRAIGHT INTO LDS (`global_load_lds_dwordx4`: a wave instruction moves one 1 KB panel row,
// no VGPR round trip, no ds_write), which frees the staging registers of gram_tile (228 -> 113 VGPRs: FOUR resident blocks per
// CU instead of two) and lets the slabs be double-buffered: one barrier per slab instead of two, and the rows of slab s + 1
// are requested a whole slab of MFMAs ahead instead of being waited for at the next ds_write.  X^T y comes from the fragments the waves
// hold anyway (each wave owns two of the eight 16-column strips), with y[row] read beside them.
// Same k-step order over the rows as gram_tile -> the same bits in every Gram element; X^T y is summed in another order
// (lane-private partial sums over the row groups, combined at the end) and agrees to rounding.
// Measured at N = 10M, D = 128 on one box (tools/k4_bench.py, Gram + reduction; profiles/r05_notes.md): 16-row slabs with four
// resident blocks per CU 3.16 ms, three blocks 3.17, 32-row slabs x 2 blocks 3.23-3.26, 64 rows x 1 block 3.28; k_gram 3.51.
#ifndef BC_GD_KR
#define BC_GD_KR 16
#endif
#ifndef BC_GD_BLOCKS
#define BC_GD_BLOCKS 4         // resident blocks per CU: 4 x 37 KB of LDS, 113 VGPRs (four waves per SIMD)
#endif
#define BC_GD_LDX 144
template <int V>
__device__ __forceinline__ void gram_dma_tile(const GramArgs& a, long long split, double* __restrict__ panel /* [2][KR][LDX] */,
                                              double* __restrict__ ys /* [2][KR] */) {
  constexpr int BT = 128, KR = BC_GD_KR, LDX = BC_GD_LDX;
  constexpr int NA = GramDiag<BT, V>::n;
  // the two strips of X^T y this wave accumulates: both are in the fragment set it loads for its MFMAs
  constexpr int YS0 = V == 0 ? 0 : (V == 1 ? 4 : (V == 2 ? 2 : 6)), YS1 = YS0 + 1;
  constexpr bool ys_in_a = V == 2;          // (wave 2 holds strips 2, 3 as A fragments; the others hold theirs as B fragments)
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const long long r_begin = split * a.rows_per_split;
  long long r_end = r_begin + a.rows_per_split;
  if (r_end > a.n_rows) r_end = a.n_rows;
  double4_t acc[NA];
#pragma unroll
  for (int x = 0; x < NA; ++x) acc[x] = (double4_t){0., 0., 0., 0.};
  double vy0 = 0., vy1 = 0.;

  // wave V moves rows 8V .. 8V+7 of a slab: lane l the 16 bytes [2l, 2l+1] of the row; wave 0 also the 32 y values (two
  // dwords each).  Rows past the end of the split are zero-filled by hand (there is no range check on this path).
  auto request = [&](long long r0, int buf) {
    double* pb = panel + (size_t)buf * KR * LDX;
#pragma unroll
    for (int q = 0; q < KR / 4; ++q) {
      const int lr = V * (KR / 4) + q;
      const long long r = r0 + lr;
      double* dst = pb + lr * LDX;
      if (r < r_end) {
        __builtin_amdgcn_global_load_lds((const void*)(a.z + (size_t)r * a.dz + 2 * lane),
                                         (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
      } else {
        dst[2 * lane] = 0.;
        dst[2 * lane + 1] = 0.;
      }
    }
    if (V == 0) {
#pragma unroll
      for (int h = 0; h < KR; h += 32) {                             // 32 rows' y values per instruction
        double* yb = ys + buf * KR + h;
        const long long r = r0 + h + (lane >> 1);
        const long long rc = r < r_end ? r : r_begin;               // (clamped: zeroed below)
        if (h + (lane >> 1) < KR)
          __builtin_amdgcn_global_load_lds((const void*)(reinterpret_cast<const int*>(a.z + (size_t)rc * a.dz + a.d) + (lane & 1)),
                                           (void __attribute__((address_space(3)))*)yb, 4, 0, 0);
      }
    }
  };
  // (y of rows past the end: the x values of those rows are zero, so whatever y holds there multiplies a zero -- unless it is
  // a NaN/inf: overwrite)
  auto patch_y = [&](long long r0, int buf) {
    if (V == 0 && r0 + KR > r_end) {
      double* yb = ys + buf * KR;
      for (int i = lane; i < KR; i += 64)
        if (r0 + i >= r_end) yb[i] = 0.;
    }
  };

  if (r_begin < r_end) request(r_begin, 0);
  int buf = 0;
  for (long long r0 = r_begin; r0 < r_end; r0 += KR, buf ^= 1) {
    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0) (lgkm / exp untouched): this wave's share of the slab has landed
    patch_y(r0, buf);
    __syncthreads();                             // everybody's share has, and nobody reads the other buffer any more
    if (r0 + KR < r_end) request(r0 + KR, buf ^ 1);
    const double* pb = panel + (size_t)buf * KR * LDX;
    const double* yb = ys + buf * KR;
    constexpr int NF = BT / 16;
    double fa[2][NF], fb[2][NF], fy[2];
    auto frags = [&](int kk, double (&xa)[NF], double (&xb)[NF], double& y) {
      const double* row = pb + (kk * 4 + g) * LDX;
      gram_diag_frags<BT, V>(row, row, j, xa, xb);
      y = yb[kk * 4 + g];
    };
    frags(0, fa[0], fb[0], fy[0]);
#pragma unroll
    for (int kk = 0; kk < KR / 4; ++kk) {
      if (kk + 1 < KR / 4) frags(kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1], fy[(kk + 1) & 1]);
      gram_diag_mma<BT, V>(fa[kk & 1], fb[kk & 1], acc);
      const double x0 = ys_in_a ? fa[kk & 1][YS0] : fb[kk & 1][YS0], x1 = ys_in_a ? fa[kk & 1][YS1] : fb[kk & 1][YS1];
      vy0 = fma(x0, fy[kk & 1], vy0);
      vy1 = fma(x1, fy[kk & 1], vy1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // X^T y: the four row groups g of a strip column, combined in a fixed order
  vy0 += __shfl_xor(vy0, 16, 64);
  vy0 += __shfl_xor(vy0, 32, 64);
  vy1 += __shfl_xor(vy1, 16, 64);
  vy1 += __shfl_xor(vy1, 32, 64);
  if (g == 0) {
    double* py = a.partial_y + (size_t)split * a.nt * BT;
    py[YS0 * 16 + j] = vy0;
    py[YS1 * 16 + j] = vy1;
  }
  double* out = a.partial + (size_t)split * a.ntri * BT * BT;
  gram_diag_store<BT, V>(out, g, j, acc);
}

__global__ __launch_bounds__(256, BC_GD_BLOCKS) void k_gram_dma(GramArgs a) {
  extern __shared__ double gd_lds[];            // [2][KR][LDX] panel + [2][KR] y
  double* panel = gd_lds;
  double* ys = gd_lds + 2 * BC_GD_KR * BC_GD_LDX;
  const long long split = (long long)(blockIdx.x >> 3) * 8 + (blockIdx.x & 7);     // (as k_gram with one tile per split)
  if (split >= a.splits) return;
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {     // (scalar branch: one variant per wave)
    case 0: gram_dma_tile<0>(a, split, panel, ys); break;
    case 1: gram_dma_tile<1>(a, split, panel, ys); break;
    case 2: gram_dma_tile<2>(a, split, panel, ys); break;
    default: gram_dma_tile<3>(a, split, panel, ys); break;
  }
}

// Sum of the per-split partial tiles IN SPLIT ORDER, in two levels so that it is parallel: level 1 adds runs of SEG
// consecutive splits (one thread per element and run, eight loads in flight), level 2 adds the run sums in run order
// and scatters the tile (and its mirror image) into the dense d x d matrix.  The association is fixed by (splits, SEG):
// run-to-run deterministic.  (One level with one thread per element walked 2 048 splits serially: 0.81 ms at N = 10M,
// D = 128, 13 % on top of the Gram kernel itself.)
#define BC_GRAM_SEG 16
__global__ __launch_bounds__(256) void k_gram_reduce1(const double* __restrict__ partial, long long splits, long long nruns,
                                                     size_t tile_elems /* ntri*bt*bt */, int nt, int bt, double* __restrict__ part2) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const long long run = blockIdx.y;
  if (e >= tile_elems) return;
  {
    // sub-tiles below the diagonal of a diagonal tile were never written (k_gram) and are never read (k_gram_reduce2)
    const int tri = (int)(e / ((size_t)bt * bt)), ee = (int)(e % ((size_t)bt * bt));
    int ta = 0, rem = tri;
    while (rem >= nt - ta) { rem -= nt - ta; ++ta; }
    if (rem == 0 && (ee / bt) / 16 > (ee % bt) / 16) return;
  }
  const long long s0 = run * BC_GRAM_SEG;
  long long s1 = s0 + BC_GRAM_SEG;
  if (s1 > splits) s1 = splits;
  const double* p = partial + e;
  double acc = 0.0;
  long long sp = s0;
  for (; sp + 8 <= s1; sp += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(sp + u) * tile_elems];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; sp < s1; ++sp) acc += p[(size_t)sp * tile_elems];
  part2[(size_t)run * tile_elems + e] = acc;
}

__global__ __launch_bounds__(256) void k_gram_reduce2(const double* __restrict__ part2, const double* __restrict__ partial_y,
                                                     long long splits, long long nruns, int ntri, int nt, int bt, int dz,
                                                     double* __restrict__ out, double* __restrict__ out_y) {
  // grid = (ntri, bt*bt/256): one output element per thread
  const int tri = blockIdx.x;
  // X^T (w*y): 32 columns per block of tile 0 (blockIdx.y = chunk), the splits in eight contiguous runs summed in split order
  // with eight loads in flight, the run sums added in run order -- the association is fixed by (splits), run-to-run
  // deterministic.  (One thread per column walking all splits with one load in flight: 0.18 ms at 512 splits, 0.33 ms at
  // D = 512 -- the tail of the whole Gram.)
  // (the chunk loop strides by the grid: bt*bt/256 blocks cover 2048 columns at BT = 128 in one trip, any D in several)
  if (tri == 0) {
    __shared__ double ys[8][32];
    for (int c0 = blockIdx.y * 32; c0 < dz; c0 += gridDim.y * 32) {       // block-uniform trip count
      const int cl = threadIdx.x & 31, part = threadIdx.x >> 5, c = c0 + cl;
      const long long len = (splits + 7) / 8;
      const long long s0 = part * len;
      long long s1 = s0 + len;
      if (s1 > splits) s1 = splits;
      double acc = 0.0;
      if (c < dz) {
        const double* py = partial_y + c;
        const size_t st = (size_t)nt * bt;
        long long sp = s0;
        for (; sp + 8 <= s1; sp += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = py[(size_t)(sp + u) * st];
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; sp < s1; ++sp) acc += py[(size_t)sp * st];
      }
      ys[part][cl] = acc;
      __syncthreads();
      if (part == 0 && c < dz) {
        double t = ys[0][cl];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += ys[q][cl];
        out_y[c] = t;
      }
      __syncthreads();
    }
  }
  int ta = 0, rem = tri;
  while (rem >= nt - ta) { rem -= nt - ta; ++ta; }
  const int tb = ta + rem;
  const int e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e >= bt * bt) return;
  const int r = ta * bt + e / bt, c = tb * bt + e % bt;
  if (r >= dz || c >= dz || (ta == tb && r > c)) return;
  const size_t tile_elems = (size_t)ntri * bt * bt;
  const double* p = part2 + (size_t)tri * bt * bt + e;
  double acc = 0.0;
  long long run = 0;
  for (; run + 8 <= nruns; run += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(run + u) * tile_elems];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; run < nruns; ++run) acc += p[(size_t)run * tile_elems];
  out[(size_t)r * dz + c] = acc;
  out[(size_t)c * dz + r] = acc;
}

// Coreset-sized Gram (the samplers: <= a few hundred rows, once per gradient of the beta-Cores loop): ONE launch, no partials
// and no reduction.  Block (ta <= tb) owns the 16 x 16 tile of Z^T diag(w) Z over ALL dz columns (y included: column d of the
// tile row is X^T (w*y)); the rows are parked in LDS 128 at a time and summed in row order.
#define BC_GRAM_SMALL_ROWS 512
__global__ __launch_bounds__(256) void k_gram_small(const double* __restrict__ z, const double* __restrict__ w, int n_rows, int dz,
                                                   double* __restrict__ out, double* __restrict__ out_y) {
  const int ta = blockIdx.x, tb = blockIdx.y;
  if (ta > tb) return;
  __shared__ double za[128][17], zb[128][16];
  const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
  const int r = ta * 16 + tr, c = tb * 16 + tc, d = dz - 1;
  double acc = 0.;
  for (int i0 = 0; i0 < n_rows; i0 += 128) {
    const int rows = n_rows - i0 < 128 ? n_rows - i0 : 128;
    for (int e = threadIdx.x; e < rows * 16; e += 256) {
      const int i = e >> 4, k = e & 15;
      const double wi = w ? w[i0 + i] : 1.0;
      const int ca = ta * 16 + k, cb = tb * 16 + k;
      za[i][k] = ca < dz ? wi * z[(size_t)(i0 + i) * dz + ca] : 0.;      // (w[:,None]*X)
      zb[i][k] = cb < dz ? z[(size_t)(i0 + i) * dz + cb] : 0.;
    }
    __syncthreads();
    for (int i = 0; i < rows; ++i) acc = fma(za[i][tr], zb[i][tc], acc);
    __syncthreads();
  }
  if (r > c || c >= dz || r >= d) return;
  if (c == d) { out_y[r] = acc; return; }
  out[(size_t)r * d + c] = acc;
  out[(size_t)c * d + r] = acc;
}

// grow-only device scratch, owned by the context: the sampler calls weighted_post on <= M coreset rows thousands of
// times (bcores.py:39 -> sampler -> weighted_post), a hipMalloc/hipFree pair per call would dominate
static int gram_buf(bc_ctx* ctx, int which, size_t doubles, double** out) {
  int rc = bc_scratch_grow(ctx, &ctx->gram[which], doubles);
  if (rc) return rc;
  *out = ctx->gram[which].p;
  return BC_OK;
}

template <int BT>
static int run_gram(bc_ctx* ctx, const bc_data* data, const double* w_dev, double* out_dev, double* outy_dev) {
  const int dz = data->dz, d = dz - 1;
  const int nt = (d + BT - 1) / BT;
  const int ntri = nt * (nt + 1) / 2;
  // one unweighted diagonal tile whose rows hold at least 128 doubles: the LDS-DMA kernel (k_gram_dma); BC_GRAM_DMA=0: the
  // register-staged k_gram (A/B)
  const int dma_env = getenv("BC_GRAM_DMA") ? atoi(getenv("BC_GRAM_DMA")) : 1;      // (read per call: tests flip it)
  const bool use_dma = BT == 128 && ntri == 1 && w_dev == nullptr && dz >= 128 && dma_env != 0 && data->n_rows >= 4096;
  const int KR = use_dma ? BC_GD_KR : (BT == 64 ? 32 : 16);
  // Row splits, in units of the 2 * n_cu resident block slots (BC_GRAM_WAVES overrides), at least 2 slabs per split.
  // One tile (D <= 128): exactly one block per slot -- every further split writes, and the reduction reads back, another
  // 128 KB tile (8 per slot were 268 MB of partials at N = 10M, D = 128: 6.06 ms with one, 7.76 ms with eight).  Several
  // tiles (D = 512: ten): short blocks, eight rounds -- 12.9 ms against 18.6 with one round (8-12 rounds: the same; 16-48:
  // 13.1-14.6).  Long blocks let the ten tile-blocks of a split drift apart by more rows than their XCD's L2 holds, so the
  // shared rows are fetched again; static priorities or a start stagger change nothing (profiles/r03_notes.md).
  static const int waves_env = getenv("BC_GRAM_WAVES") ? atoi(getenv("BC_GRAM_WAVES")) : 0;
  const int waves = waves_env > 0 ? waves_env : (ntri == 1 ? 1 : 8);
  const int slots = use_dma ? BC_GD_BLOCKS : (BT == 64 ? 3 : 2);      // resident blocks per CU (the kernels' launch bounds)
  long long want_splits = ((long long)ctx->n_cu * slots * (waves > 0 ? waves : 1) + ntri - 1) / ntri;
  long long max_splits = (data->n_rows + 2 * KR - 1) / (2 * KR);
  long long splits = want_splits < max_splits ? want_splits : max_splits;
  if (splits < 1) splits = 1;
  long long rps = (data->n_rows + splits - 1) / splits;
  rps = ((rps + KR - 1) / KR) * KR;
  if (rps < KR) rps = KR;
  splits = (data->n_rows + rps - 1) / rps;
  if (splits < 1) splits = 1;
  const long long nruns = (splits + BC_GRAM_SEG - 1) / BC_GRAM_SEG;
  const size_t tile_elems = (size_t)ntri * BT * BT;
  double* partial = nullptr;
  double* partial_y = nullptr;
  // [splits] partial tiles, then [nruns] run sums behind them
  int rcb = gram_buf(ctx, 0, (size_t)(splits + nruns) * tile_elems, &partial);
  if (!rcb) rcb = gram_buf(ctx, 1, (size_t)splits * nt * BT, &partial_y);
  if (rcb) return rcb;
  hipError_t e = hipSuccess;
  GramArgs a;
  a.z = data->z;
  a.w = w_dev;
  a.partial = partial;
  a.partial_y = partial_y;
  a.n_rows = data->n_rows;
  a.rows_per_split = rps;
  a.splits = splits;
  a.dz = dz;
  a.d = d;
  a.nt = nt;
  a.ntri = ntri;
  int rc = bc_timer_begin(ctx, 2);
  if (!rc) {
    if (use_dma) {
      const size_t lds = (size_t)(2 * BC_GD_KR * BC_GD_LDX + 2 * BC_GD_KR) * sizeof(double);      // 37 KB at 16-row slabs
      static unsigned attr_done_mask = 0;
      if (!((attr_done_mask >> (ctx->device & 31)) & 1u)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_dma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done_mask |= 1u << (ctx->device & 31);
      }
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gram_dma, dim3((unsigned)(((splits + 7) / 8) * 8)), dim3(256), lds, ctx->stream, a);
        e = hipGetLastError();
      }
    } else {
      hipLaunchKernelGGL(k_gram<BT>, dim3((unsigned)(((splits + 7) / 8) * 8 * ntri)), dim3(256), 0, ctx->stream, a);
      e = hipGetLastError();
    }
  }
  if (!rc && e == hipSuccess) {
    double* part2 = partial + (size_t)splits * tile_elems;
    hipLaunchKernelGGL(k_gram_reduce1, dim3((unsigned)((tile_elems + 255) / 256), (unsigned)nruns), dim3(256), 0, ctx->stream, partial, splits,
                       nruns, tile_elems, nt, BT, part2);
    hipLaunchKernelGGL(k_gram_reduce2, dim3(ntri, (BT * BT + 255) / 256), dim3(256), 0, ctx->stream, part2, partial_y, splits, nruns, ntri,
                       nt, BT, d, out_dev, outy_dev);
    e = hipGetLastError();
  }
  if (!rc && e == hipSuccess) rc = bc_timer_end(ctx, 2);      // the timed region is the whole K4: Gram kernel + both reduction levels
  if (e != hipSuccess) return bc_hip_fail(e, "bc_weighted_gram", __FILE__, __LINE__);
  return rc;
}

// Coreset-sized inputs straight from host memory (the samplers call weighted_post on the <= M coreset rows once per
// gradient, bcores.py:39 -> sampler -> model_linreg.py:25-34): rows and weights travel in ONE transfer from the pinned
// staging area, both results come back in ONE, one synchronisation in all -- the per-call latency is what counts here.
extern "C" int bc_weighted_gram_host(bc_ctx* ctx, const double* z_rowmajor, int64_t n_rows, int32_t dz, const double* w,
                                     double* out_xtwx, double* out_xtwy) {
  if (!ctx || !out_xtwx || !out_xtwy || n_rows < 0 || dz < 2 || (n_rows > 0 && !z_rowmajor)) {
    bc_set_error("bc_weighted_gram_host: bad argument (rows must be [x (D >= 1), y])");
    return BC_INVALID_ARGUMENT;
  }
  const int d = dz - 1;
  const size_t n_z = (size_t)n_rows * dz, n_in = n_z + (w ? (size_t)n_rows : 0), n_out = (size_t)d * d + (size_t)d;
  if (n_in > ctx->pinned_doubles || n_out > ctx->pinned_doubles) {
    bc_set_error("bc_weighted_gram_host: meant for coreset-sized inputs (%lld rows x %d); upload larger ones with bc_data_from_host",
                 (long long)n_rows, dz);
    return BC_INVALID_ARGUMENT;
  }
  if (n_rows == 0) {
    memset(out_xtwx, 0, (size_t)d * d * sizeof(double));
    memset(out_xtwy, 0, (size_t)d * sizeof(double));
    return BC_OK;
  }
  BC_HIP(hipSetDevice(ctx->device));
  double* in_dev = nullptr;
  double* out_dev = nullptr;
  int rc = gram_buf(ctx, 4, n_in, &in_dev);
  if (!rc) rc = gram_buf(ctx, 2, n_out, &out_dev);
  if (rc) return rc;
  BC_HIP(hipStreamSynchronize(ctx->stream));            // nothing enqueued may still use the staging area
  memcpy(ctx->pinned, z_rowmajor, n_z * sizeof(double));
  if (w) memcpy(ctx->pinned + n_z, w, (size_t)n_rows * sizeof(double));
  BC_HIP(hipMemcpyAsync(in_dev, ctx->pinned, n_in * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  bc_data view;
  view.ctx = ctx;
  view.n_rows = n_rows;
  view.dz = dz;
  view.z = in_dev;
  view.owned = false;
  const double* w_dev = w ? in_dev + n_z : nullptr;
  if (n_rows <= BC_GRAM_SMALL_ROWS) {
    const int nb = (dz + 15) / 16;
    rc = bc_timer_begin(ctx, 2);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gram_small, dim3(nb, nb), dim3(256), 0, ctx->stream, (const double*)in_dev, w_dev, (int)n_rows, dz, out_dev,
                       out_dev + (size_t)d * d);
    BC_HIP(hipGetLastError());
    rc = bc_timer_end(ctx, 2);
  } else {
    rc = d > 64 ? run_gram<128>(ctx, &view, w_dev, out_dev, out_dev + (size_t)d * d) : run_gram<64>(ctx, &view, w_dev, out_dev, out_dev + (size_t)d * d);
  }
  if (rc) return rc;
  BC_HIP(hipMemcpyAsync(ctx->pinned, out_dev, n_out * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(out_xtwx, ctx->pinned, (size_t)d * d * sizeof(double));
  memcpy(out_xtwy, ctx->pinned + (size_t)d * d, (size_t)d * sizeof(double));
  return BC_OK;
}

extern "C" int bc_weighted_gram(bc_ctx* ctx, const bc_data* data, const double* w, double* out_xtwx, double* out_xtwy) {
  if (!ctx || !data || !out_xtwx || !out_xtwy) { bc_set_error("bc_weighted_gram: bad argument"); return BC_INVALID_ARGUMENT; }
  if (data->ctx != ctx) { bc_set_error("bc_weighted_gram: data belongs to another context"); return BC_INVALID_ARGUMENT; }
  const int dz = data->dz, d = dz - 1;
  if (d <= 0) { bc_set_error("bc_weighted_gram: rows must be [x (D >= 1), y]"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipSetDevice(ctx->device));
  if (data->n_rows == 0) {
    memset(out_xtwx, 0, (size_t)d * d * sizeof(double));
    memset(out_xtwy, 0, (size_t)d * sizeof(double));
    return BC_OK;
  }
  double* w_dev = nullptr;
  double* out_dev = nullptr;
  double* outy_dev = nullptr;
  int rc = gram_buf(ctx, 2, (size_t)d * d, &out_dev);
  if (!rc) rc = gram_buf(ctx, 3, (size_t)d, &outy_dev);
  if (!rc && w) rc = gram_buf(ctx, 4, (size_t)data->n_rows, &w_dev);
  if (rc) return rc;
  if (w) BC_HIP(hipMemcpyAsync(w_dev, w, (size_t)data->n_rows * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  rc = d > 64 ? run_gram<128>(ctx, data, w_dev, out_dev, outy_dev) : run_gram<64>(ctx, data, w_dev, out_dev, outy_dev);
  if (rc) return rc;
  BC_HIP(hipMemcpyAsync(out_xtwx, out_dev, (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipMemcpyAsync(out_xtwy, outy_dev, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  return BC_OK;
}
