// Consumer side of the branch-and-bound int8 sweep (bc_prefilter_bb.h): the per-block records the sweep's rescoring waves
// left behind -> the winner's candidate record.  Included by the fused step kernel (bc_snnls.hip) and by k_bb_winner
// (bc_prefilter.hip) through bc_rescore_dev.h.
#pragma once

#define BC_BB_FLAG_OVF 1
struct BbRec {
  double score;                // exact fp64 score of the block's best candidate (-inf: none)
  long long gidx;              // its global row (LLONG_MAX: none)
  double norm;
  int nres;                    // exact rescorings this block ran
  int flags;
};


// ---- consumer side: the per-block records -> the winner's candidate record.
struct BbArgs {
  const BbRec* rec;            // nullptr: the two-pass form is in use
  const double* col;
  int nblk;
};
#define BC_BB_PER 4            // block records per thread of the consumer (nblk <= 1024, >= 256 threads)
struct BbPre {
  double sc[BC_BB_PER], nr[BC_BB_PER];
  long long gi[BC_BB_PER];
  int nres[BC_BB_PER], fl[BC_BB_PER];
};

__device__ __forceinline__ BbPre bc_bb_prefetch(const BbArgs& a) {
  BbPre p;
#pragma unroll
  for (int q = 0; q < BC_BB_PER; ++q) {
    const int i = threadIdx.x + q * blockDim.x;
    p.sc[q] = -INFINITY; p.nr[q] = 0.; p.gi[q] = LLONG_MAX; p.nres[q] = 0; p.fl[q] = 0;
    if (i < a.nblk) {
      const BbRec r = a.rec[i];
      p.sc[q] = r.score; p.nr[q] = r.norm; p.gi[q] = r.gidx; p.nres[q] = r.nres; p.fl[q] = r.flags;
    }
  }
  return p;
}

// Whole block (any size that is a multiple of 64, nblk <= BC_BB_PER * blockDim.x); `rec` may be LDS or global memory.
// Returns 1 on overflow (block-uniform; no record then), else 0 with rec = [score, global row, norm, valid, column(S)].
// ctrl: the pre-filter's counters ([1] last launch overflowed, [3] overflows, u64 [4..5] sweeps, [6..7] rows rescored).
__device__ __forceinline__ int bc_bb_pick(const BbArgs& a, const BbPre& pre, int s, int* ctrl, double* __restrict__ rec) {
  __shared__ double sv[16], sn[16];
  __shared__ long long si[16];
  __shared__ int sb[16], sres[16], sovf[16];
  __shared__ int win_blk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double bv = -INFINITY, bn = 0.;
  long long bi = LLONG_MAX;
  int bb = -1, nres = 0, ovf = 0;
#pragma unroll
  for (int q = 0; q < BC_BB_PER; ++q) {
    nres += pre.nres[q];
    ovf |= pre.fl[q] & BC_BB_FLAG_OVF;
    if (pre.gi[q] != LLONG_MAX && bc_better(pre.sc[q], pre.gi[q], bv, bi)) {
      bv = pre.sc[q]; bi = pre.gi[q]; bn = pre.nr[q]; bb = threadIdx.x + q * blockDim.x;
    }
  }
  // wave argmax on (score, row); the lane that owns the winner then says which block and norm go with it
  double wv = bv;
  long long wi = bi;
  bc_wave_argmax(wv, wi);
  wv = __shfl(wv, 0, BC_WAVE);
  {
    const int lo = __shfl((int)(wi & 0xffffffffLL), 0, BC_WAVE), hi = __shfl((int)(wi >> 32), 0, BC_WAVE);
    wi = ((long long)hi << 32) | (unsigned int)lo;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    nres += __shfl_xor(nres, d, BC_WAVE);
    ovf |= __shfl_xor(ovf, d, BC_WAVE);
  }
  if (wi != LLONG_MAX && bi == wi) { sb[wave] = bb; sn[wave] = bn; }      // rows are unique across block records
  if (lane == 0) { sv[wave] = wv; si[wave] = wi; sres[wave] = nres; sovf[wave] = ovf; }
  __syncthreads();
  int tot = 0, anyovf = 0;
  for (int w = 0; w < nw; ++w) { tot += sres[w]; anyovf |= sovf[w]; }
  if (threadIdx.x == blockDim.x - 1) {
    ctrl[1] = anyovf ? 1 : 0;
    if (anyovf) ctrl[3] += 1;
    unsigned long long* st = reinterpret_cast<unsigned long long*>(ctrl + 4);
    st[0] += 1;
    st[1] += anyovf ? 0ull : (unsigned long long)tot;
  }
  if (anyovf) return 1;
  if (threadIdx.x == 0) {
    int bw = 0;
    double v = sv[0];
    long long i = si[0];
    for (int w = 1; w < nw; ++w)
      if (bc_better(sv[w], si[w], v, i)) { v = sv[w]; i = si[w]; bw = w; }
    const bool valid = i != LLONG_MAX;
    rec[0] = valid ? v : -INFINITY;
    reinterpret_cast<long long*>(rec)[1] = valid ? i : -1;
    rec[2] = valid ? sn[bw] : 0.0;
    rec[3] = valid ? 1.0 : 0.0;
    win_blk = valid ? sb[bw] : -1;
  }
  __syncthreads();
  const int wb = win_blk;
  for (int k = threadIdx.x; k < s; k += blockDim.x) rec[BC_REC_HDR + k] = (wb >= 0) ? a.col[(size_t)wb * s + k] : 0.0;
  return 0;
}

