// Rescoring stage of the pre-filtered K3 sweep (passes B + C of bc_prefilter.hip) as a block-level device function,
// shared by the stand-alone k_rescore launch (bc_prefilter.hip: multi-rank steps, step-wise protocol) and by the
// single-rank greedy loop's fused rescoring + finish kernel (bc_snnls.hip).
//
//   Lmax = max of the sweep's per-block lower bounds; every row whose upper bound reaches Lmax is a candidate --
//   the true argmax is always among them -- found hierarchically (blocks -> tiles -> rows); the candidates are
//   rescored from the fp64 Phi with the arithmetic of k_sweep (same fma chain, same epilogue), the argmax uses
//   NumPy's tie rule on (score, global index), and the candidate record is written.
//
// If the candidate lists overflow (thousands of exact duplicates, say) NO record is produced and the function
// returns 1: the caller marks the step as "redo with the exact sweep" and the HOST re-runs that one step through
// the fp64 kernel, stream-ordered (bc_snnls.hip: pf_overflow).  Round 1 handled the overflow inside the launch with
// helper blocks spinning on a flag; that needed all of them co-resident and could time out silently.
#pragma once
#include "bc_sweep_dev.h"

#define BC_PREF_DELTA 6.2e-8
#define BC_HTILE 512   // rows per fp16 tile
#define BC_HU 10       // fp16 sample planes per batch; the stored plane count is padded to a multiple (zero planes)
#define BC_RS_TILE_LIMIT16 64   // fp16 mode recomputes candidate tiles (~3 us each): past this the fp64 sweep is cheaper
#define BC_REC_OVERFLOW (-1.0)  // rec[3] of a record that says "this rank's pre-filter overflowed: redo the step exactly"
typedef _Float16 bc_h2 __attribute__((ext_vector_type(2)));

template <int MODE>
__device__ __forceinline__ void bc_score_interval(double s0, double s1, double delta, double post_div, double& U, double& L) {
  if (MODE == 0) {
    const double a = fabs(s1) + delta;
    const double c = 1. - a * a;
    if (!(s0 == s0) || !(s1 == s1) || !(c > 1e-6)) {   // NaN, or too close to the validity boundary of giga.py:33
      U = INFINITY;
      L = -INFINITY;
      return;
    }
    const double f = s0 / sqrt(1. - s1 * s1);
    const double rc = 1. / sqrt(c);
    const double e = delta * (rc + (fabs(s0) + delta) * a * rc * rc * rc) * 1.001 + 1e-13 * (1. + fabs(f));
    U = f + e;
    L = f - e;
  } else {
    if (!(s0 == s0)) { U = INFINITY; L = -INFINITY; return; }
    const double f = s0 / post_div;
    const double e = (delta * 1.001 + 1e-13 * fabs(s0)) / fabs(post_div);
    U = f + e;
    L = f - e;
  }
}


#include "bc_bb_pick.h"

struct RescoreArgs {
  BbArgs bb;                 // branch-and-bound sweep (bc_prefilter_bb.h): per-block records instead of per-tile bounds
  const double* tiles;
  const double* norms;
  const double* v;
  const int* skip_flag;
  const float* ub;
  const float* tile_u;
  const double* blk_l;
  const float* blk_u;
  const _Float16* u16;       // fp16 mode: the per-row bounds of candidate tiles are recomputed from the mirror
  const unsigned char* live;
  const double* v_norm;
  double delta;
  int sp;
  const float2* tile_cand;   // int8 mode: the sweep left up to 4 (upper bound, row) pairs per tile
  const int* tile_ncand;
  const int2* blk_cand;      // int8 mode, round 5: ... and up to BC_RS_BLK_NC (upper bound bits, local row) pairs per BLOCK
  const int* blk_nc;         //   (count; -1: walk the block's tiles); nullptr: no such lists
  long long* cand;
  int* ctrl;
  double* rec;
  long long row_offset, ptiles;
  double post_div;
  int s, cap, nblk, ptile;
  int tile_rounds;           // tiles per sweep wave = ceil(ptiles / (4 * nblk)): block b swept tiles 4b+w + 4*nblk*i
  // two-level form (bc_prefilter_i4.h): the block lists come from the sweep's second level -- there are no tiles behind them; a
  // block with more rows in play than its list holds left them in the spill list, one that could not keep them at all means
  // "redo the step exactly"
  int two_level;
  const int2* spill;         // two-level form: pairs of the blocks whose list says -1 (count in ctrl[14], reset here)
  int spill_cap;
  long long* hot;            // ring of BC_RS_HOT rows that were in play lately (the first level's seeds), position in ctrl[9]; or nullptr
};
#define BC_RS_HOT 32         /* == BC_I4_HOT (bc_prefilter_i4.h) */

#ifndef FSTAMP      // diagnostic builds define it before including this header (bc_snnls.hip, -DBC_FIN_STAMPS)
#define FSTAMP(i) do { } while (0)
#endif
#ifndef FDBG
#define FDBG(i, v) do { } while (0)
#endif

// same per-row arithmetic as bc_sweep.hip (sequential fma chain over k, bc_row_score epilogue)
template <int MODE>
__device__ __forceinline__ double bc_exact_score(const double* __restrict__ tiles, const double* __restrict__ v, long long r,
                                                 int S, double nr, double post_div) {
  const double* p = tiles + (size_t)(r >> 7) * S * BC_TILE + (r & (BC_TILE - 1));
  double a0 = 0., a1 = 0.;
  int k = 0;
  for (; k + 32 <= S; k += 32) {     // 32 independent loads in flight: this kernel is pure latency
    double x[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) x[u] = p[(size_t)(k + u) * BC_TILE];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (MODE == 0) {
        a0 = fma(x[u], v[2 * (k + u)], a0);
        a1 = fma(x[u], v[2 * (k + u) + 1], a1);
      } else {
        a0 = fma(x[u], v[k + u], a0);
      }
    }
  }
  for (; k < S; ++k) {
    const double x = p[(size_t)k * BC_TILE];
    if (MODE == 0) {
      a0 = fma(x, v[2 * k], a0);
      a1 = fma(x, v[2 * k + 1], a1);
    } else {
      a0 = fma(x, v[k], a0);
    }
  }
  if (MODE == 0) {
    const double s0 = a0 / nr, s1 = a1 / nr;
    const bool ok = (s1 > -1. + 1e-14) && (1. - s1 * s1 > 0.);
    const double den = ok ? sqrt(1. - s1 * s1) : INFINITY;
    return s0 / den;
  }
  return a0 / nr / post_div;
}

// The same score, computed by a whole wave for ONE row: the lanes fetch the row and the sweep vectors with one
// round of independent loads -- lane l holds elements l, l+64, ... --, park them in a wave-private LDS strip and
// then run the sequential fma chains on broadcast LDS reads, so the result has the bits of bc_exact_score / k_sweep.
// The strip holds PAIRS: half A [k] = (x_k, v0_k), half B [k] = (x_k, v1_k); lanes 0-31 run the first chain on half A,
// lanes 32-63 the second one on half B -- one 16-byte LDS read and one fma per step and lane (three 8-byte reads and
// two fmas when every lane ran both chains: the chain was LDS-issue bound, ~5k of the 10k cycles of this phase).
// With a handful of candidates this replaces four dependent load batches and a one-lane chain by one round trip and
// a pipelined chain.  (Wave-private strip: LDS serves a wave's requests in order, the wavefront-scope fences only
// keep the compiler from reordering.)
// CH: pairs fetched from the strip ahead of the fmas that consume them (registers: 4 CH dwords).  With 8 (a `#pragma unroll`)
// the chain ran at 33 cycles per step -- the LDS latency showed through every eighth step; entries S..255 of the strip are
// zeros, so the trip count is rounded up to a multiple of CH without a branch (fma(0, 0, acc) == acc).
template <int MODE, int CH = 16>
__device__ __forceinline__ double bc_exact_score_wave(const double* __restrict__ tiles, const double* __restrict__ v, long long r,
                                                      int S, double nr, double post_div, double* strip /* [2][256][2] */,
                                                      double* rowcopy = nullptr /* [S]: the row, if the caller wants it */) {
  const int lane = threadIdx.x & 63;
  const double* p = tiles + (size_t)(r >> 7) * S * BC_TILE + (r & (BC_TILE - 1));
  double2* sA = reinterpret_cast<double2*>(strip);
  double2* sB = sA + 256;
  const double2* mine = (MODE == 0 && lane >= 32) ? sB : sA;
  double acc = 0.;
  for (int base = 0; base < S; base += 256) {
    double x[4], vx[4], vy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = base + 64 * e + lane;
      const bool in = k < S;
      x[e] = in ? p[(size_t)k * BC_TILE] : 0.;
      if (MODE == 0) {
        vx[e] = in ? v[2 * k] : 0.;
        vy[e] = in ? v[2 * k + 1] : 0.;
      } else {
        vx[e] = in ? v[k] : 0.;
        vy[e] = 0.;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the previous block's reads come first
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sA[64 * e + lane] = make_double2(x[e], vx[e]);
      if (MODE == 0) sB[64 * e + lane] = make_double2(x[e], vy[e]);
      if (rowcopy && base + 64 * e + lane < S) rowcopy[base + 64 * e + lane] = x[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    FSTAMP(14);
    const int n = (S - base) < 256 ? (S - base) : 256;
    const int npad = (n + CH - 1) / CH * CH;              // <= 256: CH divides 256
    for (int k0 = 0; k0 < npad; k0 += CH) {
      double2 xv[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) xv[u] = mine[k0 + u];
#pragma unroll
      for (int u = 0; u < CH; ++u) acc = fma(xv[u].x, xv[u].y, acc);
    }
  }
  if (MODE == 0) {
    const double a0 = bc_readlane(acc, 0), a1 = bc_readlane(acc, 32);
    const double s0 = a0 / nr, s1 = a1 / nr;
    const bool ok = (s1 > -1. + 1e-14) && (1. - s1 * s1 > 0.);
    const double den = ok ? sqrt(1. - s1 * s1) : INFINITY;
    return s0 / den;
  }
  return acc / nr / post_div;
}

// one plane of the fp32 chain for a thread's two rows (k_rescore's recomputation of a candidate tile)
template <int MODE>
__device__ __forceinline__ void bc_rs_accumulate(bc_h2 x, const double* __restrict__ v, int k, float (&a0)[2], float (&a1)[2]) {
  if (MODE == 0) {
    const float vx = (float)v[2 * k], vy = (float)v[2 * k + 1];
    a0[0] = fmaf((float)x[0], vx, a0[0]);
    a1[0] = fmaf((float)x[0], vy, a1[0]);
    a0[1] = fmaf((float)x[1], vx, a0[1]);
    a1[1] = fmaf((float)x[1], vy, a1[1]);
  } else {
    const float vx = (float)v[k];
    a0[0] = fmaf((float)x[0], vx, a0[0]);
    a0[1] = fmaf((float)x[1], vx, a0[1]);
  }
}


// The block-level rescoring.  Any block size that is a multiple of 64 and >= 256; every thread of the block must
// call it (it contains barriers).  `rec` may point to LDS or global memory.  Returns 1 on overflow (block-uniform).
#ifndef FSTAMP
#define FSTAMP(i) do { } while (0)
#endif
// The sweep blocks' bounds a thread of the rescoring block looks at (threads 0..255, blocks t, t+256, ...): a caller
// with other loads to wait for requests them in the same round (bc_rescore_prefetch) instead of paying a round trip
// of their own at the start of the rescoring.
#define BC_RS_BLK_NC 8      /* == BC_BLK_NC (bc_prefilter_i8.h) */
#define BC_RS_BLK_PER 2     // block lists a thread looks at: blocks t, t + blockDim (nblk <= 1024, blockDim >= 512)
struct RescorePre {
  double l[4];
  float u[4];
  // the sweep blocks' own candidate lists (int8 mirror): thread t holds those of blocks t, t + blockDim
  int nc[BC_RS_BLK_PER];
  float bu[BC_RS_BLK_PER];
  int2 c[BC_RS_BLK_PER][BC_RS_BLK_NC];
};

__device__ __forceinline__ RescorePre bc_rescore_prefetch(const RescoreArgs& a) {
  RescorePre p;
  // (clamped indices instead of branches: every thread loads, nothing separates the requests -- with the loads inside
  // `if`s the compiler waited for the first bound before it issued the rest)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = threadIdx.x + q * 256;
    const bool ok = threadIdx.x < 256 && i < a.nblk;
    const unsigned ic = ok ? (unsigned)i : 0u;
    const double l = a.blk_l[ic];
    const float u = a.blk_u[ic];
    p.l[q] = ok ? l : -INFINITY;
    p.u[q] = ok ? u : -INFINITY;
  }
#pragma unroll
  for (int q = 0; q < BC_RS_BLK_PER; ++q) {
    const int i = threadIdx.x + q * blockDim.x;
    p.nc[q] = 0;
    p.bu[q] = -INFINITY;
    if (a.blk_nc != nullptr && i < a.nblk) {
      p.nc[q] = a.blk_nc[i];
      p.bu[q] = a.blk_u[i];
      const int4* src = reinterpret_cast<const int4*>(a.blk_cand + (size_t)i * BC_RS_BLK_NC);
#pragma unroll
      for (int e = 0; e < BC_RS_BLK_NC / 2; ++e) {      // loaded whatever the count says: no dependent second round
        const int4 v = src[e];
        p.c[q][2 * e] = make_int2(v.x, v.y);
        p.c[q][2 * e + 1] = make_int2(v.z, v.w);
      }
    }
  }
  return p;
}

template <int MODE>
__device__ __forceinline__ int bc_rescore_block(const RescoreArgs& a, long long n_rows, double* __restrict__ rec, const RescorePre& mine) {
  __shared__ double sv[16];
  __shared__ long long si[16];
  __shared__ int cnt, tcnt, bcnt, ocnt;
  __shared__ int tlist[1024];
  __shared__ int blist[64];                  // sweep blocks whose maximum upper bound reaches Lmax
  __shared__ long long scand[32];            // the first candidates, kept on chip (the usual case has 1-3)
  __shared__ __attribute__((aligned(16))) double strips[4][4 * 256];   // bc_exact_score_wave: one strip per scoring wave
  __shared__ double currow[4][256];          // ... and the row it is scoring
  __shared__ double bestrow[4][256];         // the row of each scoring wave's best candidate (S <= 256): the record's column
  __shared__ double snorm[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  bool overflow = false;
  double lmax = -INFINITY;
  float bu[4];                               // this thread's share of the block upper bounds (nblk <= 1024)
  {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lmax = fmax(lmax, mine.l[q]);             // (-inf where there is no block: fmax ignores it, a NaN bound too -- as before)
      bu[q] = mine.u[q];
    }
  }
  // the sweep blocks left their own candidate lists (int8 mirror): they came in with the first round of loads
  const bool lists = a.blk_nc != nullptr && a.nblk <= BC_RS_BLK_PER * (int)blockDim.x;
  lmax = bc_wave_max_all(lmax);                // (DPP rotations + readlane: the six ds_bpermute steps of a __shfl_down tree cost ~0.7k cycles)
  if (lane == 0) sv[wave] = lmax;
  if (threadIdx.x == 0) { cnt = 0; tcnt = 0; bcnt = 0; ocnt = 0; }
  __syncthreads();
  lmax = sv[0];
  for (int w = 1; w < nw; ++w) lmax = fmax(lmax, sv[w]);
  FSTAMP(10);
  // phase B1: tiles whose maximum upper bound reaches Lmax -- first the sweep blocks whose maximum does
  // (usually one or two), then only the tiles those blocks walked
  if (lists) {
    // every row whose upper bound reaches Lmax is in its block's list (it reaches the block's best lower bound a fortiori);
    // only a block that could not keep a list (-1: a tile with more than four pairs, more than BC_RS_BLK_NC rows, or tiles
    // flushed out of LDS early) goes through the tile walk below
    // (almost every thread finds nothing: one branch-free pass decides whether it has anything to do at all)
    // (Tried and dropped: the thread that finds a candidate touching its fp64 row right away, to start the ~2 us TLB miss of
    // that first access into the 8 GB Phi early -- __syncthreads() carries a vmcnt(0), so the next barrier waits for the touch.)
    // (a float reaches the double lmax exactly when it reaches lmax rounded UP to a float: the sixteen comparisons of this
    // pre-check run in fp32)
    const float lmf = __double2float_ru(lmax);
    bool any = false;
#pragma unroll
    for (int q = 0; q < BC_RS_BLK_PER; ++q) {
      any |= mine.nc[q] < 0 && mine.bu[q] != -INFINITY && mine.bu[q] >= lmf;
#pragma unroll
      for (int e = 0; e < BC_RS_BLK_NC; ++e) any |= e < mine.nc[q] && __int_as_float(mine.c[q][e].x) >= lmf;
    }
    if (any) {
#pragma unroll
      for (int q = 0; q < BC_RS_BLK_PER; ++q) {
        if (mine.nc[q] < 0) {
          if (mine.bu[q] != -INFINITY && (double)mine.bu[q] >= lmax) {
            const int slot = atomicAdd(&bcnt, 1);
            if (slot < 64) blist[slot] = threadIdx.x + q * blockDim.x;
            if (mine.nc[q] < -1) atomicAdd(&ocnt, 1);      // (two-level form: a block that could not keep its rows)
          }
        } else {
#pragma unroll
          for (int e = 0; e < BC_RS_BLK_NC; ++e) {
            const float ub = __int_as_float(mine.c[q][e].x);
            if (e < mine.nc[q] && (double)ub >= lmax) {
              const int slot = atomicAdd(&cnt, 1);
              const long long row = mine.c[q][e].y;
              if (slot < a.cap) a.cand[slot] = row;
              if (slot < 32) scand[slot] = row;
            }
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (bu[q] != -INFINITY && (double)bu[q] >= lmax) {
        const int slot = atomicAdd(&bcnt, 1);
        if (slot < 64) blist[slot] = threadIdx.x + q * 256;
      }
  }
  FSTAMP(23);
  __syncthreads();
  FSTAMP(24);
  const int nbl = bcnt;
  FDBG(20, nbl);
  FDBG(21, cnt);
  FDBG(22, lists ? 1 : 0);
  bool merged = false;                       // int8 mirror, few blocks in play: phases B1 and B2 in one round of loads
  if (a.two_level) {
    // the lists ARE the candidates.  A block in play whose list says -1 left its pairs in the spill list (scanned below);
    // -2 (it could not keep them at all) cannot be walked: the step is redone with the exact sweep
    overflow = !lists;
  } else if (nbl <= 64 && a.tile_cand) {
    merged = true;
    const int per = 4 * a.tile_rounds, total = nbl * per;
    for (int i0 = 0; i0 < total; i0 += 4 * blockDim.x) {
      float tu[4];
      long long tt[4];
      int nc[4];
      float2 prs[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + u * blockDim.x + threadIdx.x;
        tt[u] = -1;
        tu[u] = -INFINITY;
        nc[u] = 0;
        if (idx < total) {
          const int b = blist[idx / per], q = idx % per;
          const long long t = (long long)b * 4 + (q & 3) + (long long)(q >> 2) * 4 * a.nblk;
          if (t < a.ptiles) {
            tt[u] = t;
            tu[u] = a.tile_u[t];
            nc[u] = a.tile_ncand[t];
#pragma unroll
            for (int i = 0; i < 4; ++i) prs[u][i] = a.tile_cand[t * 4 + i];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (tt[u] >= 0 && tu[u] != -INFINITY && (double)tu[u] >= lmax) {
          atomicAdd(&tcnt, 1);
          if (nc[u] <= 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float2 pr = prs[u][i];
              if (i < nc[u] && (double)pr.x >= lmax) {
                const int slot = atomicAdd(&cnt, 1);
                const long long row = tt[u] * a.ptile + (int)pr.y;
                if (slot < a.cap) a.cand[slot] = row;
                if (slot < 32) scand[slot] = row;
              }
            }
          } else {
            const int o = atomicAdd(&ocnt, 1);   // more than four local candidates: the tile hands over all its rows
            if (o < 64) tlist[o] = (int)tt[u];
          }
        }
    }
  } else if (nbl <= 64) {
    const int per = 4 * a.tile_rounds, total = nbl * per;
    for (int i0 = 0; i0 < total; i0 += 8 * blockDim.x) {
      float tu[8];
      long long tt[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + u * blockDim.x + threadIdx.x;
        tt[u] = -1;
        tu[u] = -INFINITY;
        if (idx < total) {
          const int b = blist[idx / per], q = idx % per;
          const long long t = (long long)b * 4 + (q & 3) + (long long)(q >> 2) * 4 * a.nblk;
          if (t < a.ptiles) { tt[u] = t; tu[u] = a.tile_u[t]; }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (tt[u] >= 0 && tu[u] != -INFINITY && (double)tu[u] >= lmax) {
          const int slot = atomicAdd(&tcnt, 1);
          if (slot < 1024) tlist[slot] = (int)tt[u];
        }
    }
  } else {
    // many blocks in play: scan all per-tile maxima, 16 independent loads at a time
    for (long long t0 = 0; t0 < a.ptiles; t0 += 16LL * blockDim.x) {
      float tu[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long long t = t0 + (long long)u * blockDim.x + threadIdx.x;
        tu[u] = t < a.ptiles ? a.tile_u[t] : -INFINITY;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (tu[u] != -INFINITY && (double)tu[u] >= lmax) {
          const int slot = atomicAdd(&tcnt, 1);
          if (slot < 1024) tlist[slot] = (int)(t0 + (long long)u * blockDim.x + threadIdx.x);
        }
    }
  }
  int ntl = 0;
  if (!a.two_level) {                        // (block-uniform; the two-level form walked nothing: its counters are final since the last barrier)
    __syncthreads();
    ntl = tcnt;
    if (threadIdx.x == 0 && !merged) bcnt = 0; // reused by the int8 branch below
    __syncthreads();
  }
  FSTAMP(11);
  overflow = overflow || ntl > (a.u16 ? BC_RS_TILE_LIMIT16 : 1024);      // too many tiles in play
  if (a.two_level) {
    overflow = overflow || ocnt > 0;
    if (!overflow && nbl > 0) {                // (block-uniform, rare) some block in play spilled: scan the spill list
      const int ns = a.ctrl[14];
      overflow = ns > a.spill_cap;
      if (!overflow)
        for (int i = threadIdx.x; i < ns; i += blockDim.x) {
          const int2 pr = a.spill[i];
          if ((double)__int_as_float(pr.x) >= lmax) {
            const int slot = atomicAdd(&cnt, 1);
            if (slot < a.cap) a.cand[slot] = pr.y;
            if (slot < 32) scand[slot] = pr.y;
          }
        }
      __syncthreads();
    }
    overflow = overflow || cnt > a.cap;
  } else if (merged) {
    // the pairs were consumed with the tile maxima; what is left are the tiles that hand over all of their rows
    const int no = ocnt;
    overflow = overflow || no > 64;
    if (!overflow) {
      for (int o = 0; o < no; ++o) {
        const long long row = (long long)tlist[o] * a.ptile + threadIdx.x;      // ptile == 256 rows: threads 0..255
        if ((int)threadIdx.x < a.ptile && row < n_rows && a.norms[row] != 0.) {
          const int slot = atomicAdd(&cnt, 1);
          if (slot < a.cap) a.cand[slot] = row;
          if (slot < 32) scand[slot] = row;
        }
      }
      __syncthreads();
      overflow = cnt > a.cap;
    }
  } else if (!overflow && a.tile_cand) {
    // phase B2 (int8 mirror): the pairs the sweep left for each such tile; a tile with more than four local
    // candidates hands over all of its rows
    for (int q = threadIdx.x; q < ntl; q += blockDim.x) {
      const long long t = tlist[q];
      const int n = a.tile_ncand[t];
      float2 prs[4];                           // fetched together with the count: one round trip
#pragma unroll
      for (int i = 0; i < 4; ++i) prs[i] = a.tile_cand[t * 4 + i];
      if (n <= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float2 pr = prs[i];
          if (i < n && (double)pr.x >= lmax) {
            const int slot = atomicAdd(&cnt, 1);
            const long long row = t * a.ptile + (int)pr.y;
            if (slot < a.cap) a.cand[slot] = row;
            if (slot < 32) scand[slot] = row;
          }
        }
      } else {
        const int o = atomicAdd(&bcnt, 1);      // (bcnt is free again after phase B1)
        if (o < 64) blist[o] = (int)t;
      }
    }
    __syncthreads();
    const int no = bcnt;
    overflow = no > 64;
    if (!overflow) {
      for (int o = 0; o < no; ++o) {
        const long long row = (long long)blist[o] * a.ptile + threadIdx.x;      // ptile == 256 rows: threads 0..255
        if ((int)threadIdx.x < a.ptile && row < n_rows && a.norms[row] != 0.) {
          const int slot = atomicAdd(&cnt, 1);
          if (slot < a.cap) a.cand[slot] = row;
          if (slot < 32) scand[slot] = row;
        }
      }
      __syncthreads();
      overflow = cnt > a.cap;
    }
  } else if (!overflow && a.u16) {
    // phase B2 (fp16 mirror): recompute the per-row intervals of each such tile from the mirror -- the sweep
    // wrote none.  Thread = two adjacent rows of the tile (one 4-byte load per plane, 1 KiB per plane and
    // block), up to 64 planes in flight; fp32 chain, same interval formula and delta as the sweep.
    const double delta = (MODE == 0) ? a.delta : a.delta * (*a.v_norm);
    if (threadIdx.x < 256)
    for (int q = 0; q < ntl; ++q) {
      const long long t = tlist[q];
      const bc_h2* __restrict__ tp = reinterpret_cast<const bc_h2*>(a.u16 + (size_t)t * a.sp * BC_HTILE) + threadIdx.x;
      float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
      // sp is a multiple of BC_HU = 10: batches of 50 planes (all loads of a batch in flight), then of 10
      int k0 = 0;
      for (; k0 + 50 <= a.sp; k0 += 50) {
        bc_h2 x[50];
#pragma unroll
        for (int u = 0; u < 50; ++u) x[u] = tp[(size_t)(k0 + u) * (BC_HTILE / 2)];
#pragma unroll
        for (int u = 0; u < 50; ++u) bc_rs_accumulate<MODE>(x[u], a.v, k0 + u, a0, a1);
      }
      for (; k0 < a.sp; k0 += BC_HU) {
        bc_h2 x[BC_HU];
#pragma unroll
        for (int u = 0; u < BC_HU; ++u) x[u] = tp[(size_t)(k0 + u) * (BC_HTILE / 2)];
#pragma unroll
        for (int u = 0; u < BC_HU; ++u) bc_rs_accumulate<MODE>(x[u], a.v, k0 + u, a0, a1);
      }
      const unsigned lv = a.live[t * 64 + (threadIdx.x >> 2)];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = 2 * threadIdx.x + j;                    // row within the tile
        if ((lv >> (i & 7)) & 1u) {
          double Ub, Lb;
          bc_score_interval<MODE>((double)a0[j], (double)a1[j], delta, a.post_div, Ub, Lb);
          if (Ub >= lmax) {
            const int slot = atomicAdd(&cnt, 1);
            if (slot < a.cap) a.cand[slot] = t * BC_HTILE + i;
            if (slot < 32) scand[slot] = t * BC_HTILE + i;
          }
        }
      }
    }
    __syncthreads();
    overflow = cnt > a.cap;
  } else if (!overflow) {
    // phase B2 (fp32 mirror): one wave per such tile, all of the tile's stored upper bounds in flight at once
    for (int q = wave; q < ntl; q += nw) {
      const long long t = tlist[q];
      float u8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = lane + 64 * e;
        u8[e] = i < a.ptile ? a.ub[t * a.ptile + i] : -INFINITY;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (u8[e] != -INFINITY && (double)u8[e] >= lmax) {
          const int slot = atomicAdd(&cnt, 1);
          if (slot < a.cap) a.cand[slot] = t * a.ptile + lane + 64 * e;
          if (slot < 32) scand[slot] = t * a.ptile + lane + 64 * e;
        }
    }
    __syncthreads();
    overflow = cnt > a.cap;
  }
  if (threadIdx.x == blockDim.x - 1) {         // (the last wave scores no candidate: its read-modify-writes delay nobody)
    if (a.two_level) a.ctrl[14] = 0;           // the spill list is consumed (or abandoned)
    a.ctrl[1] = overflow ? 1 : 0;              // observable: the last launch overflowed
    if (overflow) a.ctrl[3] += 1;              // ... and how often since creation
    unsigned long long* st = reinterpret_cast<unsigned long long*>(a.ctrl + 4);   // diagnostics: sweeps, candidates rescored
    st[0] += 1;
    st[1] += overflow ? 0 : (unsigned long long)cnt;
  }
  if (overflow) return 1;
  FSTAMP(12);

  const int count = cnt;
  // seeds of the two-level form's next sweeps: the rows in play go into the ring (the last wave: it scores nothing)
  if (a.hot != nullptr && wave == nw - 1) {
    const int pos = a.ctrl[9];
    const int n = count < BC_RS_HOT ? count : BC_RS_HOT;
    if (lane < n) a.hot[(pos + lane) & (BC_RS_HOT - 1)] = scand[lane];
    if (lane == 0) a.ctrl[9] = (pos + n) & (BC_RS_HOT - 1);
  }
  double bv = -INFINITY, bnorm = 0.;
  long long bi = LLONG_MAX;
  const bool keep_rows = count <= 32 && a.s <= 256;      // the winner's row stays on chip: no reload for the record
  if (count <= 32) {
    // the usual case, a handful of candidates: a wave per candidate (the first four waves)
    if (wave < 4)
      for (int j = wave; j < count; j += 4) {
        const long long r = scand[j];
        const double nr = a.norms[r];
        const double sc = bc_exact_score_wave<MODE>(a.tiles, a.v, r, a.s, nr, a.post_div, strips[wave], keep_rows ? currow[wave] : nullptr);
        const long long gi = a.row_offset + r;
        if (bc_better(sc, gi, bv, bi)) {
          bv = sc; bi = gi; bnorm = nr;
          if (keep_rows) {                               // (wave-uniform branch; currow still holds the row)
            for (int k = lane; k < a.s; k += BC_WAVE) bestrow[wave][k] = currow[wave][k];
          }
        }
      }
  } else {
    for (int j = threadIdx.x; j < count; j += blockDim.x) {
      const long long r = a.cand[j];
      const double sc = bc_exact_score<MODE>(a.tiles, a.v, r, a.s, a.norms[r], a.post_div);
      const long long gi = a.row_offset + r;
      if (bc_better(sc, gi, bv, bi)) { bv = sc; bi = gi; }
    }
  }
  FSTAMP(13);
  if (!keep_rows) bc_wave_argmax(bv, bi);     // (keep_rows: every lane of a scoring wave already holds the wave's best)
  // (sv / si were last read right after the Lmax barrier, several barriers ago: no barrier needed before they are rewritten)
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; snorm[wave] = bnorm; }
  __syncthreads();
  // every thread combines the (at most eight) wave results itself -- the same loop, the same order, hence the same winner in
  // every thread -- instead of thread 0 doing it between two more barriers
  int bw = 0;
  {
    double cv = sv[0];
    long long ci = si[0];
    const int nsc = (count <= 32) ? (nw < 4 ? nw : 4) : nw;      // a handful of candidates: only waves 0..3 scored
    for (int w = 1; w < nsc; ++w)
      if (bc_better(sv[w], si[w], cv, ci)) { cv = sv[w]; ci = si[w]; bw = w; }
    bv = cv;
    bi = ci;
  }
  const bool valid = bi != LLONG_MAX;
  if (threadIdx.x == 0) {
    rec[0] = bv;
    reinterpret_cast<long long*>(rec)[1] = valid ? bi : -1;
    rec[2] = valid ? (keep_rows ? snorm[bw] : a.norms[bi - a.row_offset]) : 0.0;
    rec[3] = valid ? 1.0 : 0.0;
  }
  const long long r = valid ? bi - a.row_offset : -1;
  const int swave = bw;
  if (keep_rows && r >= 0) {
    const double* src = bestrow[swave];
    for (int k = threadIdx.x; k < a.s; k += blockDim.x) rec[BC_REC_HDR + k] = src[k];
  } else {
    for (int k = threadIdx.x; k < a.s; k += blockDim.x) rec[BC_REC_HDR + k] = (r >= 0) ? a.tiles[bc_tile_off(r, k, a.s)] : 0.0;
  }
  return 0;
}
