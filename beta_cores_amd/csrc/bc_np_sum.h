// The sum of n EQUAL doubles exactly as NumPy's pairwise summation rounds it: the mean of a constant row.
//
// `lls.mean(axis=1)` (projector.py:26,55) reduces a contiguous axis with NumPy's pairwise sum: blocks of <= 128 elements
// are added with eight running sums over strides of 8, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then a sequential
// tail; longer rows are split in two, the left part n/2 rounded down to a multiple of 8, recursively.  The rounded mean of S
// equal numbers c is in general NOT c, so the reference centres a constant row (a data row with all-zero features) to a
// tiny constant residue (c - mean) rather than to 0, keeps it at hilbert.py:16 and applies no index shift (golden F12).  On
// equal inputs all eight running sums coincide, so the sum is a function of (c, n) only; it is evaluated for constant
// rows only.
//
// Plain C99 / C++: compiled by hipcc for the device and by gcc for tests/test_np_sum_cpu.py, which checks it bit for bit
// against np.full(n, c).sum() for every n up to 3000 and random larger ones.
#ifndef BC_NP_SUM_H
#define BC_NP_SUM_H

#if defined(__HIPCC__)
#define BC_NS __host__ __device__ __forceinline__
#define BC_NS_UNROLL _Pragma("unroll")
#else
#define BC_NS static inline
#define BC_NS_UNROLL
#endif

BC_NS double bc_np_sum_const_leaf(double c, int n) {   // n <= 128
  if (n < 8) {
    double r = 0.;
    for (int i = 0; i < n; ++i) r += c;
    return r;
  }
  double r = c;
  const int nb = n >> 3;
  for (int i = 1; i < nb; ++i) r += c;
  double res = ((r + r) + (r + r)) + ((r + r) + (r + r));
  for (int i = nb << 3; i < n; ++i) res += c;
  return res;
}

// n <= 256 (two levels of halving always reach blocks of <= 128): the K1 kernels
BC_NS double bc_np_sum_const_256(double c, int n) {
  if (n <= 128) return bc_np_sum_const_leaf(c, n);
  int n2 = n >> 1;
  n2 -= n2 & 7;
  const int m = n - n2;                        // <= 135
  const double left = bc_np_sum_const_leaf(c, n2);
  if (m <= 128) return left + bc_np_sum_const_leaf(c, m);
  int m2 = m >> 1;
  m2 -= m2 & 7;
  return left + (bc_np_sum_const_leaf(c, m2) + bc_np_sum_const_leaf(c, m - m2));
}

// n <= 8192 (one buffer of NumPy's reduction machinery).  The halving tree has at most THREE distinct block sizes per
// level (the "round the left half down to a multiple of 8" rule keeps every level's sizes within three values; checked
// for every n by the host test), and the sum of a block depends on its size only.  So the tree is walked by LEVELS: the
// distinct sizes of every level top-down, then their sums bottom-up -- fixed-size arrays with compile-time indices
// (registers on the device; the explicit recursion stack this replaces lived in 656 B of scratch per thread).
#define BC_NS_LEVELS 8                         /* 8192 = 128 * 2^6: at most 7 levels */
BC_NS double bc_np_sum_const_8192(double c, int n) {
  if (n <= 128) return bc_np_sum_const_leaf(c, n);
  int sz[BC_NS_LEVELS][3];
BC_NS_UNROLL
  for (int l = 0; l < BC_NS_LEVELS; ++l) sz[l][0] = sz[l][1] = sz[l][2] = 0;
  sz[0][0] = n;
BC_NS_UNROLL
  for (int l = 0; l + 1 < BC_NS_LEVELS; ++l) {
    int cnt = 0;
BC_NS_UNROLL
    for (int k = 0; k < 3; ++k) {
      const int x = sz[l][k];
      if (x > 128) {
        int x2 = x >> 1;
        x2 -= x2 & 7;
        const int pair[2] = {x2, x - x2};
BC_NS_UNROLL
        for (int q = 0; q < 2; ++q) {
          const int v = pair[q];
          if (v != sz[l + 1][0] && v != sz[l + 1][1] && v != sz[l + 1][2]) {
            if (cnt == 0) sz[l + 1][0] = v;
            else if (cnt == 1) sz[l + 1][1] = v;
            else sz[l + 1][2] = v;             // (a fourth distinct size cannot occur)
            ++cnt;
          }
        }
      }
    }
  }
  double val[3] = {0., 0., 0.}, nxt[3] = {0., 0., 0.};
BC_NS_UNROLL
  for (int l = BC_NS_LEVELS - 1; l >= 0; --l) {
BC_NS_UNROLL
    for (int k = 0; k < 3; ++k) {
      const int x = sz[l][k];
      double v = 0.;
      if (x > 0 && x <= 128) {
        v = bc_np_sum_const_leaf(c, x);
      } else if (x > 128 && l + 1 < BC_NS_LEVELS) {
        int x2 = x >> 1;
        x2 -= x2 & 7;
        const int xr = x - x2;
        const double a = (x2 == sz[l + 1][0]) ? nxt[0] : ((x2 == sz[l + 1][1]) ? nxt[1] : nxt[2]);
        const double b = (xr == sz[l + 1][0]) ? nxt[0] : ((xr == sz[l + 1][1]) ? nxt[1] : nxt[2]);
        v = a + b;
      }
      val[k] = v;
    }
BC_NS_UNROLL
    for (int k = 0; k < 3; ++k) nxt[k] = val[k];
  }
  return val[0];
}

// Any n (the S > 256 centring pass).  NumPy reduces in buffers of 8192 elements (np.getbufsize()): every buffer is summed
// pairwise, the buffer sums are added to the running total one after the other.
BC_NS double bc_np_sum_const_any(double c, int n) {
  if (n <= 8192) return bc_np_sum_const_8192(c, n);
  const double t = bc_np_sum_const_8192(c, 8192);
  double acc = t;
  for (int i = 1; i < n / 8192; ++i) acc += t;
  if (n % 8192) acc += bc_np_sum_const_8192(c, n % 8192);
  return acc;
}

#endif  // BC_NP_SUM_H
