// Sparse-NNLS solver state on the device: the guarded greedy loop of
// bayesiancoresets/snnls/snnls.py:31-79 run WITHOUT a host round trip per iteration,
// plus the step-wise (_select / _reweight) protocol of snnls.py:102-106.
//
// Per iteration:   K3 sweep (bc_sweep.hip)  ->  k_local_winner  -> [host all-gather when
// world > 1]  ->  k_step_finish (single block): global winner, closed-form reweight
// (giga.py:40-64 / frankwolfe.py:19-40), monotone-error guard with revert, the
// retry-once-then-stop state machine, and the S-vector prep for the next sweep
// (giga.py:20-30 / frankwolfe.py:16).  All replicated state (sparse w, selected
// columns, xw = A.w) lives in device memory; every rank runs the identical finish
// kernel on identical gathered records, so ranks stay bit-identical.
#ifdef BC_FIN_STAMPS          // diagnostic build: phase time stamps of the single-block step kernels
#include <hip/hip_runtime.h>
__device__ unsigned long long g_fin_stamps[64];
#define FSTAMP(i) do { if (threadIdx.x == 0) g_fin_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define FDBG(i, v) do { if (threadIdx.x == 0) g_fin_stamps[i] = (unsigned long long)(v); } while (0)
extern "C" int bc_debug_fin_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_stamps), sizeof(g_fin_stamps)) == hipSuccess ? 0 : -1;
}
#endif
#include "bc_rescore_dev.h"
#include "bc_layout.h"
#include "bc_i8_quant.h"
#include "bc_i4_quant.h"
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>

int bc_launch_sweep(bc_phi* p, int mode, const double* v_dev, double post_div, const int* skip_flag, double* rec_dev);
#define BC_PREF_DEFAULT_PREC 8
#define BC_V_PAD 16            // >= BC_HU - 1 (bc_prefilter.hip): doubles of zero padding after the sweep vector(s)
struct bc_comm;
int bc_comm_all_gather_dev(bc_comm* c, const double* send_dev, double* recv_dev, size_t count);
extern "C" int bc_comm_info(const bc_comm* c, int32_t* rank, int32_t* world);
struct bc_pref;
int bc_pref_create(bc_phi* phi, int prec, bc_pref** out);
int bc_pref_precision(const bc_pref* p);
void bc_pref_destroy(bc_pref* p);
void bc_pref_set_cap(bc_pref* p, int cap);
const int* bc_pref_ctrl(const bc_pref* p);
void bc_pref_set_qv(bc_pref* p, const int* qv_dev);
int bc_pref_sp4(const bc_pref* p);
void bc_pref_set_qv4(bc_pref* p, const int* qv4_dev);
int bc_pref_sp8(const bc_pref* p);
int bc_pref_two_level(const bc_pref* p);
int bc_pref_adapt(bc_pref* p);
int bc_pref_two_level_active(const bc_pref* p);
int bc_pref_launch(bc_pref* p, int mode, const double* v_dev, const double* v_norm_dev, double post_div,
                   const int* skip_flag, double* rec_dev);
int bc_pref_launch_sweep(bc_pref* p, int mode, const double* v_dev, const double* v_norm_dev, double post_div,
                         const int* skip_flag, double* rec_dev, RescoreArgs* r_out);

struct SnnlsState {
  long long nnz;        // length of the (idx, val) list, selection order; val may be 0
  long long npos;       // count(val > 0)  == SparseNNLS.size()
  long long iter;       // consumed iterations since reset (trace length)
  long long sel_f;      // last picked global index, -1 if none
  double sel_score;
  double sel_norm;
  double err_cur;       // ||A.w - b||_2 for the current w
  double xw_sq;         // ||A.w||^2 for the current w
  int skip;             // select_fail | reached_limit : makes the next sweep a no-op
  int reached_limit;
  int retried;
  int select_fail;      // the reference's _select would raise (giga.py:28-29)
  int sel_valid;
  int last_status;      // status of the last step-wise call
  int overflow;         // list capacity exceeded (host bug guard)
  int pf_overflow;      // the pre-filter's candidate lists overflowed in this step: nothing was consumed, the host
                        // re-runs the step with the exact fp64 sweep (sticky until it does; later launches are no-ops)
  double v_norm;        // ||v|| of the dot-mode sweep vector (scales the pre-filter's error bound)
};

struct SnnlsDev {
  SnnlsState* st;
  long long* idx;
  double* val;
  double* prev_val;
  double* cols;       // [cap][s]
  double* colnorm;    // [cap]
  long long cap;
  double* b;
  double* bn;
  double* xw;
  double* xw_prev;
  double* v;          // sweep vectors: GIGA [s][2], else [s]
  int* qv;            // the same, quantised for the int8 pre-filter's sweep (bc_i8_quant.h record) -- or nullptr
  int sp4;            // k-groups of that record
  int* qv4;           // two-level pre-filter: the same in 4-bit digits for its first level (bc_i4_quant.h record) -- or nullptr
  int sp8;            // k-groups of that record
  double* xf;         // picked column
  const double* cand_all;
  const double* tiles;
  const double* norms;
  long long n_rows, row_offset;
  long long* tr_f;
  int* tr_status;
  double* tr_err;
  long long tr_cap;
  double bnorm, tol, norm_sum;
  int s, world, rec_len;
  // single-rank fast path: the finish kernel reduces the sweep's per-block candidates itself
  const double* blk_val;
  const long long* blk_idx;
  int nblk;
  int fuse_winner;
};

struct bc_snnls {
  bc_ctx* ctx = nullptr;
  bc_phi* phi = nullptr;
  int alg = 0;
  SnnlsDev d;
  // second buffer set for set_weights (swapped in after the rebuild kernel)
  long long* idx2 = nullptr;
  double* val2 = nullptr;
  double* cols2 = nullptr;
  double* colnorm2 = nullptr;
  // The small state arrays (st, b, bn, xw, xw_prev, v, xf, the send record) live in ONE allocation and the active
  // lists (both buffer sets) in another: the single-block step kernel starts every greedy step with cold TLBs (a 1 GB
  // sweep ran in between), and its first round of loads paid one page-table walk per separately allocated array.
  void* state_slab = nullptr;
  void* list_slab = nullptr;
  bc_pref* pref = nullptr;         // reduced-precision pre-filter of the sweep (large shards), see bc_prefilter.hip
  double* cand_send = nullptr;     // this rank's candidate record (S + 4 doubles)
  bool cand_send_owned = true;     // false once the host bound its own exchange buffers
  bc_comm* comm = nullptr;         // native RCCL exchange (bc_comm.hip): the loop all-gathers by itself
  double* cand_all_owned = nullptr;
  long long nnz_upper = 0;   // host-side upper bound on the list length
  long long iter_upper = 0;
  RescoreArgs rs;                 // argument block of the rescoring stage for the step in flight (fused finish)
  bool rs_pending = false;        // step_local ran the sweep only: step_finish must run the fused rescoring + finish
  bool exact_step = false;        // the step in flight uses the exact fp64 sweep (redo after a pre-filter overflow)
  bool pref_suspended = false;    // repeated overflows: the rest of this build call sweeps in fp64
  int consec_overflow = 0;
};

// ------------------------------------------------------------------ device building blocks (single block)
// The S-vector algebra of a step (dot products, step sizes, the next sweep vectors) is done by WAVE 0 alone:
// lane l owns elements l, l+64, ... of every vector and reductions are shuffle trees, so a step needs a handful
// of block barriers instead of three per reduction.  Only the O(nnz*S) work -- rebuilding xw from the list,
// scaling / searching the list -- uses the whole block.  Every kernel goes through the same helpers, so the
// fused loop and the step-wise protocol produce the same bits.
#define BC_FIN_THREADS 512

// xw = sum_j val[j] * cols[j], err = ||xw - b||, ||xw||^2 and the positive count.  The list is split over
// G = blockDim/S thread groups (fixed split => deterministic), partial vectors are combined in group order.
#define BC_PF_NC 16     // list columns per thread requested up front by the _pf kernels (32 was measured: no gain)
// PF: the first BC_PF_NC terms of every thread's share come from `c16` (cols prefetched at kernel start, slot u <->
// list entry g + u*G) except the entry appended in this very step (`at_new`), whose column is P.xf.
template <bool PF>
__device__ void dev_xw_err_t(const SnnlsDev& P, SnnlsState& S, double* red, const double (&c16)[BC_PF_NC], long long at_new) {
  __shared__ double part[BC_FIN_THREADS];
  __shared__ int npos_sh;
  const int s = P.s;
  const long long nnz = S.nnz;
  if (threadIdx.x == 0) npos_sh = 0;
  __syncthreads();
  int np = 0;
  if (s <= (int)blockDim.x) {
    const int G = blockDim.x / s;
    const int g = threadIdx.x / s, k = threadIdx.x - g * s;
    double acc = 0.0;
    if (g < G) {
      // 16 independent loads in flight per thread: the list lives in global memory (L2) and a dependent
      // load per term would cost a full memory latency each.  The k == 0 thread of each group also counts
      // the positive weights it walks over (every list slot belongs to exactly one group).
      long long j = g;
      if (PF) {
#pragma unroll
        for (int u = 0; u < BC_PF_NC; ++u) {
          if (j < nnz) {
            const double vj = P.val[j];
            const double cv = (j == at_new) ? P.xf[k] : c16[u];
            acc = fma(vj, cv, acc);
            np += (vj > 0.) ? 1 : 0;
            j += G;
          }
        }
      }
      for (; j + 15 * (long long)G < nnz; j += 16 * (long long)G) {
        double v8[16], c8[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          v8[u] = P.val[j + u * (long long)G];
          c8[u] = P.cols[(size_t)(j + u * (long long)G) * s + k];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          acc = fma(v8[u], c8[u], acc);
          np += (v8[u] > 0.) ? 1 : 0;
        }
      }
      for (; j + 3 * (long long)G < nnz; j += 4 * (long long)G) {
        double v4[4], c4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v4[u] = P.val[j + u * (long long)G];
          c4[u] = P.cols[(size_t)(j + u * (long long)G) * s + k];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc = fma(v4[u], c4[u], acc);
          np += (v4[u] > 0.) ? 1 : 0;
        }
      }
      for (; j < nnz; j += G) {
        const double vj = P.val[j];
        acc = fma(vj, P.cols[(size_t)j * s + k], acc);
        np += (vj > 0.) ? 1 : 0;
      }
      if (k == 0 && np) atomicAdd(&npos_sh, np);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < BC_WAVE) {
      double e = 0.0, q = 0.0;
      for (int kk = threadIdx.x; kk < s; kk += BC_WAVE) {
        double t = part[kk];
        for (int gg = 1; gg < G; ++gg) t += part[gg * s + kk];
        P.xw[kk] = t;
        const double d = t - P.b[kk];
        e = fma(d, d, e);
        q = fma(t, t, q);
      }
      e = bc_wave_sum(e);
      q = bc_wave_sum(q);
      if (threadIdx.x == 0) {
        S.err_cur = sqrt(e);
        S.xw_sq = q;
        S.npos = npos_sh;
      }
    }
  } else {
    for (long long j = threadIdx.x; j < nnz; j += blockDim.x) np += (P.val[j] > 0.) ? 1 : 0;
    if (np) atomicAdd(&npos_sh, np);
    double e = 0.0, q = 0.0;
    for (int k = threadIdx.x; k < s; k += blockDim.x) {
      double acc = 0.0;
      for (long long j = 0; j < nnz; ++j) acc = fma(P.val[j], P.cols[(size_t)j * s + k], acc);
      P.xw[k] = acc;
      const double d = acc - P.b[k];
      e = fma(d, d, e);
      q = fma(acc, acc, q);
    }
    double r2[2] = {e, q};
    bc_block_sum_n<2>(r2, red);
    if (threadIdx.x == 0) {
      S.err_cur = sqrt(r2[0]);
      S.xw_sq = r2[1];
      S.npos = npos_sh;
    }
  }
  __syncthreads();
}

__device__ void dev_xw_err(const SnnlsDev& P, SnnlsState& S, double* red) {
  const double none[BC_PF_NC] = {0.};
  dev_xw_err_t<false>(P, S, red, none, -1);
}

// vectors for the next sweep.  GIGA: giga.py:21-30 ; FW / OMP: residual b - A.w          (wave 0 + one barrier)
// With an int8 pre-filter (P.qv) the same wave also leaves the vectors' int8 digits behind (bc_i8_quant.h): the maxima are
// taken while the elements are in registers, and the digits' pass reads them back from a 4 KB LDS copy (S <= BC_VQ_MAX;
// larger S re-reads v from global memory).
#define BC_VQ_MAX 256
#ifndef BC_QD_INTS
#define BC_QD_INTS 512
#endif
template <int ALG>
__device__ void dev_prep(const SnnlsDev& P, SnnlsState& S, double* red) {
  __shared__ double vq[2 * BC_VQ_MAX];
  __shared__ int qd[BC_QD_INTS];           // LDS image of the two digit records (S <= ~256)
  const int s = P.s;
  if (threadIdx.x < BC_WAVE) {
    const int lane = threadIdx.x;
    const bool quant = P.qv != nullptr;
    const bool lds_copy = quant && s <= BC_VQ_MAX;
    double vnorm = 1.;                  // ||v|| of the sweep vector (GIGA's are unit vectors)
    double m0 = 0., m1 = 0.;            // this lane's max |v0|, |v1|
    if (ALG == BC_ALG_GIGA) {
      double nw = sqrt(S.xw_sq);
      nw = (nw == 0.) ? 1. : nw;
      double bd = 0.0;
      for (int k = lane; k < s; k += BC_WAVE) {
        const double xn = P.xw[k] / nw;
        P.v[2 * k + 1] = xn;
        if (lds_copy) vq[2 * k + 1] = xn;
        m1 = bc_i8q_absmax(m1, xn);
        bd = fma(P.bn[k], xn, bd);
      }
      bd = bc_wave_sum_all(bd);
      double cn = 0.0;
      for (int k = lane; k < s; k += BC_WAVE) {
        const double c = P.bn[k] - bd * (P.xw[k] / nw);
        cn = fma(c, c, cn);
      }
      cn = sqrt(bc_wave_sum_all(cn));
      const bool fail = cn < P.tol;
      for (int k = lane; k < s; k += BC_WAVE) {
        const double c = P.bn[k] - bd * (P.xw[k] / nw);
        const double v0 = fail ? c : c / cn;
        P.v[2 * k] = v0;
        if (lds_copy) vq[2 * k] = v0;
        m0 = bc_i8q_absmax(m0, v0);
      }
      if (lane == 0) S.select_fail = fail ? 1 : 0;
    } else {
      double vn = 0.0;
      for (int k = lane; k < s; k += BC_WAVE) {
        const double r = P.b[k] - P.xw[k];
        P.v[k] = r;
        if (lds_copy) vq[k] = r;
        m0 = bc_i8q_absmax(m0, r);
        vn = fma(r, r, vn);
      }
      vn = bc_wave_sum(vn);
      vnorm = sqrt(vn);
      if (lane == 0) {
        S.select_fail = 0;
        S.v_norm = vnorm;
      }
    }
    if (lane == 0) S.skip = S.select_fail | S.reached_limit | S.pf_overflow;
    FSTAMP(15);
    if (quant) {
      // this wave wrote v (and its LDS copy) itself: a wave-level fence orders the digits' reads behind those stores
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const double* vsrc = lds_copy ? vq : P.v;
      m0 = bc_wave_max_all(m0);
      m1 = bc_wave_max_all(m1);
      if (4 * (P.sp4 + P.sp8) <= BC_QD_INTS) {
        // one element per lane, both records in one pass (bc_i4_quant.h: bc_q_wave_both)
        if (ALG == BC_ALG_GIGA) bc_q_wave_both<0>(vsrc, s, P.sp4, P.sp8, 1., P.qv, P.qv4, lane, m0, m1, qd);
        else bc_q_wave_both<1>(vsrc, s, P.sp4, P.sp8, vnorm, P.qv, P.qv4, lane, m0, m1, qd);
      } else {
        if (ALG == BC_ALG_GIGA) bc_i8q_wave<0>(vsrc, s, P.sp4, 1., P.qv, lane, m0, m1);
        else bc_i8q_wave<1>(vsrc, s, P.sp4, vnorm, P.qv, lane, m0, m1);
        if (P.qv4 != nullptr) {
          if (ALG == BC_ALG_GIGA) bc_i4q_wave<0>(vsrc, s, P.sp8, 1., P.qv4, lane, m0, m1);
          else bc_i4q_wave<1>(vsrc, s, P.sp8, vnorm, P.qv4, lane, m0, m1);
        }
      }
    }
  }
  __syncthreads();
}

// winner over the gathered candidate records: max score, lowest global index on ties.
// OMP additionally weighs the active set's negative direction (orthopursuit.py:25-35).
template <int ALG>
__device__ void dev_pick(const SnnlsDev& P, SnnlsState& S, double* red) {
  __shared__ int src_rec;
  __shared__ long long src_list;
  const int s = P.s;
  if (P.world == 1 && ALG != BC_ALG_OMP) {
    // one record: nothing to reduce (the shuffle argmax below costs ~3.5k cycles of ds_bpermute latency)
    const double* rec = P.cand_all;
    const bool valid = rec[3] != 0.0;
    if (threadIdx.x == 0) {
      S.sel_valid = valid ? 1 : 0;
      S.sel_f = valid ? reinterpret_cast<const long long*>(rec)[1] : -1;
      S.sel_score = valid ? rec[0] : -INFINITY;
      if (valid) S.sel_norm = rec[2];
    }
    if (P.xf != rec + BC_REC_HDR) {                 // (block-uniform; the fused kernel aliases the two)
      if (valid)
        for (int k = threadIdx.x; k < s; k += blockDim.x) P.xf[k] = rec[BC_REC_HDR + k];
    }
    __syncthreads();
    return;
  }
  if (threadIdx.x < BC_WAVE) {
    // wave 0: one record header per lane (independent loads), shuffle argmax on (score, global index)
    // carrying the record slot; the winner's column is copied straight from its record
    const int lane = threadIdx.x;
    double bv = -INFINITY;
    long long bi = LLONG_MAX;
    int br = -1;
    for (int r = lane; r < P.world; r += BC_WAVE) {
      const double* rec = P.cand_all + (size_t)r * P.rec_len;
      const double sc = rec[0], ok = rec[3];
      const long long gi = reinterpret_cast<const long long*>(rec)[1];
      if (ok != 0.0 && bc_better(sc, gi, bv, bi)) { bv = sc; bi = gi; br = r; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const double ov = __shfl_down(bv, d, BC_WAVE);
      const long long oi = bc_shfl_down_ll(bi, d);
      const int orr = __shfl_down(br, d, BC_WAVE);
      if (bc_better(ov, oi, bv, bi)) { bv = ov; bi = oi; br = orr; }
    }
    br = __shfl(br, 0, BC_WAVE);
    const bool valid = br >= 0;
    const double* win = P.cand_all + (size_t)(valid ? br : 0) * P.rec_len;
    if (lane == 0) {
      S.sel_valid = valid ? 1 : 0;
      S.sel_f = valid ? bi : -1;
      S.sel_score = bv;
      if (valid) S.sel_norm = win[2];
      src_rec = br;
      src_list = -1;
    }
    if (ALG != BC_ALG_OMP && valid)
      for (int k = lane; k < s; k += BC_WAVE) P.xf[k] = win[BC_REC_HDR + k];
  }
  __syncthreads();
  if (ALG != BC_ALG_OMP) return;
  if (ALG == BC_ALG_OMP && S.sel_valid && S.npos > 0) {
    // neg = max over active j of -(An[:,j] . residual)
    __shared__ double nv[16];
    __shared__ long long ni[16], nj[16];
    double bv = -INFINITY;
    long long bi = LLONG_MAX, bj = -1;
    for (long long j = threadIdx.x; j < S.nnz; j += blockDim.x) {
      if (!(P.val[j] > 0.)) continue;
      double acc = 0.0;
      for (int k = 0; k < s; ++k) acc = fma(P.cols[(size_t)j * s + k], P.v[k], acc);
      const double d = -(acc / P.colnorm[j]);
      if (bc_better(d, P.idx[j], bv, bi)) { bv = d; bi = P.idx[j]; bj = j; }
    }
    // block argmax carrying the list slot
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      const double ov = __shfl_down(bv, dlt, BC_WAVE);
      const long long oi = bc_shfl_down_ll(bi, dlt), oj = bc_shfl_down_ll(bj, dlt);
      if (bc_better(ov, oi, bv, bi)) { bv = ov; bi = oi; bj = oj; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { nv[wave] = bv; ni[wave] = bi; nj[wave] = bj; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int nwv = (blockDim.x + 63) >> 6;
      for (int w = 1; w < nwv; ++w)
        if (bc_better(nv[w], ni[w], bv, bi)) { bv = nv[w]; bi = ni[w]; bj = nj[w]; }
      if (!(S.sel_score >= bv)) {   // orthopursuit.py:31 `if pos >= neg` else take the active point
        S.sel_f = bi;
        S.sel_score = bv;
        S.sel_norm = P.colnorm[bj];
        src_rec = -1;
        src_list = bj;
      }
    }
    __syncthreads();
  }
  if (S.sel_valid) {
    const double* src = (src_rec >= 0) ? P.cand_all + (size_t)src_rec * P.rec_len + BC_REC_HDR
                                       : P.cols + (size_t)src_list * s;
    for (int k = threadIdx.x; k < s; k += blockDim.x) P.xf[k] = src[k];
  }
  __syncthreads();
}

// single-rank variant of dev_pick: reduce the sweep's per-block candidates here (no record round trip)
__device__ void dev_pick_blocks(const SnnlsDev& P, SnnlsState& S) {
  __shared__ double sv[16];
  __shared__ long long si[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double bv = -INFINITY;
  long long bi = LLONG_MAX;
  for (int i = threadIdx.x; i < P.nblk; i += blockDim.x)
    if (bc_better(P.blk_val[i], P.blk_idx[i], bv, bi)) { bv = P.blk_val[i]; bi = P.blk_idx[i]; }
  bc_wave_argmax(bv, bi);
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nwv = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nwv; ++w)
      if (bc_better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
    const bool valid = bi != LLONG_MAX;
    S.sel_valid = valid ? 1 : 0;
    S.sel_f = valid ? bi : -1;
    S.sel_score = bv;
    S.sel_norm = valid ? P.norms[bi - P.row_offset] : 0.0;
  }
  __syncthreads();
  if (S.sel_valid) {
    const long long r = S.sel_f - P.row_offset;
    for (int k = threadIdx.x; k < P.s; k += blockDim.x) P.xf[k] = P.tiles[bc_tile_off(r, k, P.s)];
  }
  __syncthreads();
}

// closed-form step sizes.  Returns 1 when the reference would raise NumericalPrecisionError
// (w untouched), else 0 with (alpha, beta) set.                                    (wave 0 + one barrier)
template <int ALG>
__device__ int dev_step_sizes(const SnnlsDev& P, const SnnlsState& S, double* red, double& alpha, double& beta) {
  __shared__ double ab[2];
  __shared__ int ab_fail;
  const int s = P.s;
  if (threadIdx.x < BC_WAVE) {
    const int lane = threadIdx.x;
    int fail = 0;
    double al = 0., be = 0.;
    if (ALG == BC_ALG_GIGA) {   // giga.py:42-61
      double nw = sqrt(S.xw_sq);
      nw = (nw == 0.) ? 1. : nw;
      double ff = 0.;
      for (int k = lane; k < s; k += BC_WAVE) ff = fma(P.xf[k], P.xf[k], ff);
      const double nf = sqrt(bc_wave_sum_all(ff));
      double r0 = 0., r1 = 0., r2 = 0.;
      for (int k = lane; k < s; k += BC_WAVE) {
        const double fn = P.xf[k] / nf, wn = P.xw[k] / nw;
        r0 = fma(P.bn[k], fn, r0);
        r1 = fma(P.bn[k], wn, r1);
        r2 = fma(wn, fn, r2);
      }
      const double bxf = bc_wave_sum_all(r0), bxw = bc_wave_sum_all(r1), xwxf = bc_wave_sum_all(r2);
      const double gA = bxf - bxw * xwxf;
      const double gB = bxw - bxf * xwxf;
      if (gA <= 0. || gB < 0.) {
        fail = 1;
      } else {
        const double a = gB / (gA + gB) / nw;
        const double b = gA / (gA + gB) / nf;
        double nx = 0.;
        for (int k = lane; k < s; k += BC_WAVE) {
          const double x = a * P.xw[k] + b * P.xf[k];
          nx = fma(x, x, nx);
        }
        nx = sqrt(bc_wave_sum_all(nx));
        double xb = 0.;
        for (int k = lane; k < s; k += BC_WAVE) {
          const double x = a * P.xw[k] + b * P.xf[k];
          xb = fma(x / nx, P.bn[k], xb);
        }
        xb = bc_wave_sum_all(xb);
        const double scale = P.bnorm / nx * xb;
        al = a * scale;
        be = b * scale;
      }
    } else {                    // frankwolfe.py:20-37
      const double nf = S.sel_norm;
      if (S.npos == 0) {
        al = 0.;
        be = P.norm_sum / nf;
      } else {
        const double c = P.norm_sum / nf;
        double num = 0., den = 0.;
        for (int k = lane; k < s; k += BC_WAVE) {
          const double d = c * P.xf[k] - P.xw[k];
          num = fma(d, P.b[k] - P.xw[k], num);
          den = fma(d, d, den);
        }
        num = bc_wave_sum_all(num);
        den = bc_wave_sum_all(den);
        if (num < 0. || den == 0. || num > den) {
          fail = 1;
        } else {
          al = 1. - num / den;
          be = c * num / den;
        }
      }
    }
    if (lane == 0) {
      ab[0] = al;
      ab[1] = be;
      ab_fail = fail;
    }
  }
  __syncthreads();
  alpha = ab[0];
  beta = ab[1];
  return ab_fail;
}

// w = alpha*w ; w[f] = max(0, w[f] + beta)   (giga.py:63-64, frankwolfe.py:39-40)
__device__ void dev_apply(const SnnlsDev& P, SnnlsState& S, double alpha, double beta) {
  __shared__ int slot;
  const int s = P.s;
  const long long f = S.sel_f;
  if (threadIdx.x == 0) slot = INT_MAX;
  __syncthreads();
  for (long long j = threadIdx.x; j < S.nnz; j += blockDim.x) {
    P.val[j] = alpha * P.val[j];
    if (P.idx[j] == f) atomicMin(&slot, (int)j);
  }
  __syncthreads();
  long long at = slot == INT_MAX ? -1 : slot;
  if (at >= 0) {
    if (threadIdx.x == 0) {
      const double nv = P.val[at] + beta;
      P.val[at] = nv > 0. ? nv : 0.;
    }
  } else {
    const double nv = (0. + beta) > 0. ? (0. + beta) : 0.;
    if (nv > 0.) {
      if (S.nnz < P.cap) {
        at = S.nnz;
        for (int k = threadIdx.x; k < s; k += blockDim.x) P.cols[(size_t)at * s + k] = P.xf[k];
        if (threadIdx.x == 0) {
          P.idx[at] = f;
          P.val[at] = nv;
          P.colnorm[at] = S.sel_norm;
        }
        __syncthreads();
        if (threadIdx.x == 0) S.nnz = at + 1;
      } else if (threadIdx.x == 0) {
        S.overflow = 1;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void dev_trace(const SnnlsDev& P, SnnlsState& S, long long f, int status) {
  if (threadIdx.x == 0) {
    if (S.iter < P.tr_cap) {
      P.tr_f[S.iter] = f;
      P.tr_status[S.iter] = status;
      P.tr_err[S.iter] = S.err_cur;
    }
    S.iter += 1;
  }
}

// any gathered record carrying the overflow marker?  (block-uniform; every rank sees the same records)
__device__ bool dev_records_overflow(const double* cand_all, int world, int rec_len) {
  __shared__ int ovf;
  if (threadIdx.x == 0) ovf = 0;
  __syncthreads();
  for (int r = threadIdx.x; r < world; r += blockDim.x)
    if (cand_all[(size_t)r * rec_len + 3] < 0.) ovf = 1;
  __syncthreads();
  return ovf != 0;
}

// ------------------------------------------------------------------ kernels (grid = 1 block)
template <int ALG>
__global__ __launch_bounds__(BC_FIN_THREADS) void k_prep(SnnlsDev P, int reset_retry) {
  __shared__ SnnlsState S;
  __shared__ double red[64];
  if (threadIdx.x == 0) {
    S = *P.st;
    if (reset_retry) S.retried = 0;
  }
  __syncthreads();
  dev_prep<ALG>(P, S, red);
  if (threadIdx.x == 0) *P.st = S;
}

// one guarded greedy iteration, snnls.py:41-74.  The S-vectors (b, bn, xw, xf, previous xw) live in
// LDS for the duration of the kernel (lds_vecs != 0): one parallel load at the start instead of a
// global-memory round trip per use.
template <int ALG>
__global__ __launch_bounds__(BC_FIN_THREADS) void k_step_finish(SnnlsDev P0, int lds_vecs) {
  extern __shared__ double vec_lds[];
  __shared__ SnnlsState S;
  __shared__ double red[64];
  __shared__ int sh_fail;
  if (threadIdx.x == 0) S = *P0.st;
  SnnlsDev P = P0;
  const int s = P.s;
  if (lds_vecs) {
    P.b = vec_lds;
    P.bn = vec_lds + s;
    P.xw = vec_lds + 2 * s;
    P.xf = vec_lds + 3 * s;
    P.xw_prev = vec_lds + 4 * s;
    for (int k = threadIdx.x; k < s; k += blockDim.x) {
      P.b[k] = P0.b[k];
      P.bn[k] = P0.bn[k];
      P.xw[k] = P0.xw[k];
    }
  }
  __syncthreads();
  if (S.reached_limit || S.pf_overflow) return;      // snnls.py:32-34 / :73-74; a pending exact redo consumes nothing
  if (!P.fuse_winner && !S.select_fail && dev_records_overflow(P0.cand_all, P.world, P.rec_len)) {
    if (threadIdx.x == 0) { S.pf_overflow = 1; S.skip = 1; *P0.st = S; }
    return;
  }
  const bool guard = S.npos > 0;                     // snnls.py:44-45 (check_error_monotone is True for GIGA/FW)
  int fail = S.select_fail;                          // _select raised
  long long f = -1;
  if (!fail) {
    if (P.fuse_winner) dev_pick_blocks(P, S);
    else dev_pick<ALG>(P, S, red);
    if (!S.sel_valid) fail = 1;
    f = S.sel_f;
  }
  if (!fail) {
    double alpha = 0., beta = 0.;
    fail = dev_step_sizes<ALG>(P, S, red, alpha, beta);
    if (!fail) {
      // keep what is needed to revert (snnls.py:46-47)
      const long long nnz0 = S.nnz, npos0 = S.npos;
      const double err0 = S.err_cur, xwsq0 = S.xw_sq;
      // (no barrier needed here: each thread later rescales exactly the val[j] it saved, and xw is only
      //  overwritten behind dev_xw_err's barriers)
      for (long long j = threadIdx.x; j < nnz0; j += blockDim.x) P.prev_val[j] = P.val[j];
      for (int k = threadIdx.x; k < s; k += blockDim.x) P.xw_prev[k] = P.xw[k];
      dev_apply(P, S, alpha, beta);
      dev_xw_err(P, S, red);
      if (guard) {
        if (S.err_cur > err0) {                      // snnls.py:58-61
          for (long long j = threadIdx.x; j < nnz0; j += blockDim.x) P.val[j] = P.prev_val[j];
          for (int k = threadIdx.x; k < s; k += blockDim.x) P.xw[k] = P.xw_prev[k];
          __syncthreads();
          if (threadIdx.x == 0) {
            S.nnz = nnz0;
            S.npos = npos0;
            S.err_cur = err0;
            S.xw_sq = xwsq0;
          }
          fail = 1;
        } else if (threadIdx.x == 0) {
          S.retried = 0;                             // snnls.py:62
        }
        __syncthreads();
      }
    }
  }
  if (threadIdx.x == 0) {
    if (fail) {                                      // snnls.py:63-72
      if (S.retried) S.reached_limit = 1;
      else S.retried = 1;
    }
    sh_fail = fail;
  }
  __syncthreads();
  dev_trace(P, S, f, sh_fail);
  __syncthreads();
  dev_prep<ALG>(P, S, red);
  if (lds_vecs)
    for (int k = threadIdx.x; k < s; k += blockDim.x) P0.xw[k] = P.xw[k];
  if (threadIdx.x == 0) {
    S.sel_valid = 0;      // the picked column lived in LDS: a later step-wise reweight must fetch it again
    *P0.st = S;
  }
}

// The same step with ONE round of global loads up front: state, S-vectors, all candidate records (with their
// columns), the weight list (val, idx) and each thread's first 16 list columns are requested before anything is
// consumed, then the step runs out of LDS / registers and the results are written back at the end.  The plain
// kernel above pays a memory latency per phase (state -> records -> column -> list -> columns ...), ~7 dependent
// round trips; both produce the same bits (same device functions, same orders).  Used when the list and the records
// fit the LDS budget (nnz_hint + 1 <= BC_PF_MAXNNZ, world * rec_len <= BC_PF_MAXREC, S <= blockDim).
#define BC_PF_MAXNNZ 1024
#define BC_PF_MAXREC 2048
// RS: the pre-filter's rescoring stage (bc_rescore_dev.h) runs INSIDE this launch, between the up-front loads and
// the step: its record goes straight into LDS (single rank, no exchange).  One launch per greedy step besides the
// sweep instead of two (k_rescore 11.7 us + k_step_finish_pf 12.6 us in round 1).
// RS = 2: the branch-and-bound sweep (bc_prefilter_bb.h) already rescored its candidates: the "rescoring stage" is the argmax
// over its per-block records (bc_bb_pick), one round of loads that travels with the other up-front loads.
template <int ALG, int RS>
__global__ __launch_bounds__(BC_FIN_THREADS) void k_step_finish_pf(SnnlsDev P0, int nnz_hint, RescoreArgs ra, long long n_rows) {
  extern __shared__ double pf_lds[];
  __shared__ SnnlsState S;
  __shared__ double red[64];
  __shared__ int sh_fail;
  const int s = P0.s;
  const int nrec = RS ? P0.rec_len : (P0.fuse_winner ? 0 : P0.world * P0.rec_len);
  FSTAMP(0);
  double* l_b = pf_lds;
  double* l_bn = l_b + s;
  double* l_xw = l_bn + s;
  double* l_xf = l_xw + s;
  double* l_xwp = l_xf + s;
  double* l_rec = l_xwp + s;
  double* l_val = l_rec + nrec;
  long long* l_idx = reinterpret_cast<long long*>(l_val + (nnz_hint + 1));
  // ---- one round of loads.  Everything is requested into REGISTERS first and parked in LDS afterwards: written as
  // "load, store to LDS" group by group the compiler waited for each group before it issued the next -- six dependent
  // round trips (state, vectors, list, bounds, ...) instead of one: 6.6k of the step's 45k cycles (tools/fin_stamps.py).
  // Indices are clamped instead of branched around, so that nothing separates the requests.
  const int k0 = (int)threadIdx.x < s ? (int)threadIdx.x : 0;                 // (s <= blockDim: one element per thread)
  const double r_b = P0.b[k0], r_bn = P0.bn[k0], r_xw = P0.xw[k0];
  double r_rec[BC_PF_MAXREC / BC_FIN_THREADS];
  if (!RS) {
#pragma unroll
    for (int u = 0; u < BC_PF_MAXREC / BC_FIN_THREADS; ++u) {
      const int i = threadIdx.x + u * BC_FIN_THREADS;
      r_rec[u] = (nrec > 0) ? P0.cand_all[i < nrec ? i : 0] : 0.;
    }
  }
  double r_val[BC_PF_MAXNNZ / BC_FIN_THREADS];
  long long r_idx[BC_PF_MAXNNZ / BC_FIN_THREADS];
#pragma unroll
  for (int u = 0; u < BC_PF_MAXNNZ / BC_FIN_THREADS; ++u) {
    const int j = threadIdx.x + u * BC_FIN_THREADS;
    const int jc = j < nnz_hint ? j : 0;                                      // (the list slab always holds slot 0)
    r_val[u] = P0.val[jc];
    r_idx[u] = P0.idx[jc];
  }
  RescorePre pre;
  BbPre bbpre;
  if (RS == 1) pre = bc_rescore_prefetch(ra);        // the sweep blocks' bounds travel with the other up-front loads
  if (RS == 2) bbpre = bc_bb_prefetch(ra.bb);        // ... or their records
  const int G = blockDim.x / s;
  const int g = threadIdx.x / s, kk = threadIdx.x - g * s;
  double c16[BC_PF_NC];
#pragma unroll
  for (int u = 0; u < BC_PF_NC; ++u) {
    long long j = (g < G ? g : 0) + (long long)u * G;
    j = j < nnz_hint ? j : (nnz_hint > 0 ? nnz_hint - 1 : 0);      // clamped: always a valid slot, unused when past the list
    c16[u] = P0.cols[(size_t)j * s + (g < G ? kk : 0)];
  }
  FSTAMP(25);
  // ---- park
  if (threadIdx.x == 0) S = *P0.st;                  // (requested behind the others: it arrives with them)
  if ((int)threadIdx.x < s) {
    l_b[threadIdx.x] = r_b;
    l_bn[threadIdx.x] = r_bn;
    l_xw[threadIdx.x] = r_xw;
  }
  if (!RS) {
#pragma unroll
    for (int u = 0; u < BC_PF_MAXREC / BC_FIN_THREADS; ++u) {
      const int i = threadIdx.x + u * BC_FIN_THREADS;
      if (i < nrec) l_rec[i] = r_rec[u];
    }
  }
#pragma unroll
  for (int u = 0; u < BC_PF_MAXNNZ / BC_FIN_THREADS; ++u) {
    const int j = threadIdx.x + u * BC_FIN_THREADS;
    if (j < nnz_hint) {
      l_val[j] = r_val[u];
      l_idx[j] = r_idx[u];
    }
  }
  __syncthreads();
  SnnlsDev P = P0;
  P.b = l_b;
  P.bn = l_bn;
  P.xw = l_xw;
  P.xf = l_xf;
  P.xw_prev = l_xwp;
  P.val = l_val;
  P.idx = l_idx;
  if (nrec) P.cand_all = l_rec;
  if (RS) { P.world = 1; P.fuse_winner = 0; }
  // one record, already in LDS (the rescoring stage of this launch wrote it): its column IS the picked column -- dev_pick
  // then has nothing to copy (and no barrier to pass)
  if (RS && ALG != BC_ALG_OMP) P.xf = l_rec + BC_REC_HDR;
  FSTAMP(1);
  if (S.reached_limit || S.pf_overflow) return;      // snnls.py:32-34 / :73-74; a pending exact redo consumes nothing
  bool pf_ovf = false;
  if (RS) {
    if (RS == 2) {
      if (!S.skip) pf_ovf = bc_bb_pick(ra.bb, bbpre, s, ra.ctrl, l_rec) != 0;
    } else if (!S.skip) {
      pf_ovf = bc_rescore_block<(ALG == BC_ALG_GIGA) ? 0 : 1>(ra, n_rows, l_rec, pre) != 0;
    }
    __syncthreads();
    FSTAMP(2);
  } else if (!P.fuse_winner && !S.select_fail) {
    pf_ovf = dev_records_overflow(l_rec, P.world, P.rec_len);
  }
  if (pf_ovf) {
    if (threadIdx.x == 0) { S.pf_overflow = 1; S.skip = 1; *P0.st = S; }
    return;
  }
  if (S.nnz > nnz_hint) {                            // the host's bound on the list length was wrong: refuse
    if (threadIdx.x == 0) { S.overflow = 1; *P0.st = S; }
    return;
  }
  const bool guard = S.npos > 0;                     // snnls.py:44-45
  int fail = S.select_fail;                          // _select raised
  long long f = -1;
  bool wrote = false;
  const long long nnz0 = S.nnz, npos0 = S.npos;
  const double err0 = S.err_cur, xwsq0 = S.xw_sq;
  if (!fail) {
    if (P.fuse_winner) dev_pick_blocks(P, S);
    else dev_pick<ALG>(P, S, red);
    if (!S.sel_valid) fail = 1;
    f = S.sel_f;
  }
  FSTAMP(3);
  if (!fail) {
    double alpha = 0., beta = 0.;
    fail = dev_step_sizes<ALG>(P, S, red, alpha, beta);
    FSTAMP(4);
    if (!fail) {
      for (int k = threadIdx.x; k < s; k += blockDim.x) P.xw_prev[k] = P.xw[k];
      dev_apply(P, S, alpha, beta);                  // on the LDS copy of (val, idx); a new column goes to global cols
      FSTAMP(5);
      const long long at_new = (S.nnz == nnz0 + 1) ? nnz0 : -1;
      dev_xw_err_t<true>(P, S, red, c16, at_new);
      FSTAMP(6);
      wrote = true;
      if (guard) {
        if (S.err_cur > err0) {                      // snnls.py:58-61: nothing was written back yet, just drop it
          for (int k = threadIdx.x; k < s; k += blockDim.x) P.xw[k] = P.xw_prev[k];
          __syncthreads();
          if (threadIdx.x == 0) {
            S.nnz = nnz0;
            S.npos = npos0;
            S.err_cur = err0;
            S.xw_sq = xwsq0;
          }
          fail = 1;
          wrote = false;
        } else if (threadIdx.x == 0) {
          S.retried = 0;                             // snnls.py:62
        }
        __syncthreads();
      }
    }
  }
  if (threadIdx.x == 0) {
    if (fail) {                                      // snnls.py:63-72
      if (S.retried) S.reached_limit = 1;
      else S.retried = 1;
    }
    sh_fail = fail;
  }
  __syncthreads();
  FSTAMP(7);
  dev_trace(P, S, f, sh_fail);
  __syncthreads();
  dev_prep<ALG>(P, S, red);
  FSTAMP(8);
  // ---- write back
  for (int k = threadIdx.x; k < s; k += blockDim.x) P0.xw[k] = P.xw[k];
  if (wrote) {
    const long long nnz = S.nnz;
    for (long long j = threadIdx.x; j < nnz; j += blockDim.x) P0.val[j] = l_val[j];
    if (nnz == nnz0 + 1 && threadIdx.x == 0) P0.idx[nnz0] = l_idx[nnz0];
  }
  if (threadIdx.x == 0) {
    S.sel_valid = 0;
    *P0.st = S;
  }
  FSTAMP(9);
}

template <int ALG>
__global__ __launch_bounds__(BC_FIN_THREADS) void k_pick(SnnlsDev P) {
  __shared__ SnnlsState S;
  __shared__ double red[64];
  if (threadIdx.x == 0) S = *P.st;
  __syncthreads();
  if (!S.select_fail && dev_records_overflow(P.cand_all, P.world, P.rec_len)) {
    if (threadIdx.x == 0) { S.last_status = BC_RETRY_EXACT; *P.st = S; }     // some rank's pre-filter overflowed
    return;
  }
  if (!S.select_fail) dev_pick<ALG>(P, S, red);
  if (threadIdx.x == 0) {
    S.last_status = S.select_fail ? BC_NUMERICAL_PRECISION : (S.sel_valid ? BC_OK : BC_INVALID_ARGUMENT);
    *P.st = S;
  }
}

// step-wise _reweight(f): no guard here, the host loop owns it (snnls.py:53-61)
template <int ALG>
__global__ __launch_bounds__(BC_FIN_THREADS) void k_reweight(SnnlsDev P, long long f) {
  __shared__ SnnlsState S;
  __shared__ double red[64];
  __shared__ int found;
  const int s = P.s;
  if (threadIdx.x == 0) {
    S = *P.st;
    int fd = 0;   // 0: not found, 1: xf already holds it, 2+r: record r, -1: local shard
    if (S.sel_valid && S.sel_f == f) fd = 1;
    if (!fd)
      for (int r = 0; r < P.world; ++r) {
        const double* rec = P.cand_all + (size_t)r * P.rec_len;
        if (rec[3] != 0.0 && reinterpret_cast<const long long*>(rec)[1] == f) { fd = 2 + r; break; }
      }
    if (!fd && f >= P.row_offset && f < P.row_offset + P.n_rows) fd = -1;
    found = fd;
  }
  __syncthreads();
  if (found == 0) {
    if (threadIdx.x == 0) { S.last_status = BC_INVALID_ARGUMENT; *P.st = S; }
    return;
  }
  if (found >= 2) {
    const double* rec = P.cand_all + (size_t)(found - 2) * P.rec_len;
    for (int k = threadIdx.x; k < s; k += blockDim.x) P.xf[k] = rec[BC_REC_HDR + k];
    if (threadIdx.x == 0) S.sel_norm = rec[2];
  } else if (found == -1) {
    const long long r = f - P.row_offset;
    for (int k = threadIdx.x; k < s; k += blockDim.x) P.xf[k] = P.tiles[bc_tile_off(r, k, s)];
    if (threadIdx.x == 0) S.sel_norm = P.norms[r];
  }
  if (threadIdx.x == 0) { S.sel_f = f; S.sel_valid = 1; }
  __syncthreads();
  double alpha = 0., beta = 0.;
  const int fail = dev_step_sizes<ALG>(P, S, red, alpha, beta);
  if (!fail) {
    dev_apply(P, S, alpha, beta);
    dev_xw_err(P, S, red);
  }
  if (threadIdx.x == 0) {
    S.last_status = fail ? BC_NUMERICAL_PRECISION : BC_OK;
    *P.st = S;
  }
}

// rebuild the sparse list from (idx, val[, cols]); columns not supplied are looked up in the
// old list, then in the last candidate records, then in the local shard.
__global__ __launch_bounds__(BC_FIN_THREADS) void k_set_weights(SnnlsDev P, long long n, const long long* __restrict__ nidx,
                                                    const double* __restrict__ nval, const double* __restrict__ ncols,
                                                    long long* idx2, double* val2, double* cols2, double* colnorm2) {
  __shared__ SnnlsState S;
  __shared__ double red[64];
  __shared__ int kind, hit;
  __shared__ long long where;
  const int s = P.s;
  if (threadIdx.x == 0) { S = *P.st; S.last_status = BC_OK; }
  __syncthreads();
  for (long long j = 0; j < n; ++j) {
    const long long f = nidx[j];
    if (threadIdx.x == 0) { kind = ncols ? 1 : 0; where = -1; hit = INT_MAX; }
    __syncthreads();
    if (!ncols) {
      for (long long q = threadIdx.x; q < S.nnz; q += blockDim.x)
        if (P.idx[q] == f) atomicMin(&hit, (int)q);
      __syncthreads();
      if (threadIdx.x == 0) {
        int kd = 0;
        long long w = -1;
        if (hit != INT_MAX) { kd = 2; w = hit; }
        if (!kd)
          for (int r = 0; r < P.world; ++r) {
            const double* rec = P.cand_all + (size_t)r * P.rec_len;
            if (rec[3] != 0.0 && reinterpret_cast<const long long*>(rec)[1] == f) { kd = 3; w = r; break; }
          }
        if (!kd && f >= P.row_offset && f < P.row_offset + P.n_rows) { kd = 4; w = f - P.row_offset; }
        if (!kd) S.last_status = BC_INVALID_ARGUMENT;
        kind = kd;
        where = w;
      }
    }
    __syncthreads();
    const int kd = kind;
    const long long w = where;
    for (int k = threadIdx.x; k < s; k += blockDim.x) {
      double v = 0.0;
      if (kd == 1) v = ncols[(size_t)j * s + k];
      else if (kd == 2) v = P.cols[(size_t)w * s + k];
      else if (kd == 3) v = P.cand_all[(size_t)w * P.rec_len + BC_REC_HDR + k];
      else if (kd == 4) v = P.tiles[bc_tile_off(w, k, s)];
      cols2[(size_t)j * s + k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double nrm = 0.0;
      for (int k = 0; k < s; ++k) nrm = fma(cols2[(size_t)j * s + k], cols2[(size_t)j * s + k], nrm);
      colnorm2[j] = sqrt(nrm);
      idx2[j] = f;
      val2[j] = nval[j];
    }
    __syncthreads();
  }
  // from here on the new buffers are the list
  SnnlsDev Q = P;
  Q.idx = idx2;
  Q.val = val2;
  Q.cols = cols2;
  Q.colnorm = colnorm2;
  if (threadIdx.x == 0) { S.nnz = n; S.sel_valid = 0; }
  __syncthreads();
  dev_xw_err(Q, S, red);
  if (threadIdx.x == 0) *P.st = S;
}

__global__ __launch_bounds__(BC_FIN_THREADS) void k_reset(SnnlsDev P) {
  __shared__ double red[64];
  __shared__ SnnlsState S;
  if (threadIdx.x == 0) {
    memset(&S, 0, sizeof(S));
    S.sel_f = -1;
  }
  __syncthreads();
  dev_xw_err(P, S, red);   // nnz = 0 -> xw = 0, err = ||b||
  if (threadIdx.x == 0) *P.st = S;
}

// ------------------------------------------------------------------ host side
static int fetch_state(bc_snnls* h, SnnlsState* out) {
  bc_ctx* ctx = h->ctx;
  BC_HIP(hipMemcpyAsync(ctx->pinned, h->d.st, sizeof(SnnlsState), hipMemcpyDeviceToHost, ctx->stream));
  BC_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(out, ctx->pinned, sizeof(SnnlsState));
  if (out->overflow) {
    bc_set_error("bc_snnls: active-list capacity exceeded on device (internal error)");
    return BC_INVALID_ARGUMENT;
  }
  return BC_OK;
}

static void free_lists(bc_snnls* h) {
  if (h->list_slab) (void)hipFree(h->list_slab);
  h->list_slab = nullptr;
  h->d.idx = h->idx2 = nullptr;
  h->d.val = h->val2 = h->d.prev_val = nullptr;
  h->d.cols = h->cols2 = nullptr;
  h->d.colnorm = h->colnorm2 = nullptr;
}

static int ensure_capacity(bc_snnls* h, long long need) {
  if (need <= h->d.cap) return BC_OK;
  bc_ctx* ctx = h->ctx;
  long long ncap = std::max<long long>(need, std::max<long long>(256, h->d.cap * 2));
  const int s = h->d.s;
  // one slab: [idx | val | prev_val | colnorm | cols] of the live set first (what the step kernel reads), then the
  // second set (bc_snnls_set_weights rebuilds into it and swaps)
  const size_t n8 = (size_t)ncap * 8, ncols = (size_t)ncap * s * 8;
  const size_t total = 7 * n8 + 2 * ncols;
  char* slab = nullptr;
  hipError_t e = hipMalloc((void**)&slab, total);
  if (e != hipSuccess) return bc_hip_fail(e, "hipMalloc(active list)", __FILE__, __LINE__);
  long long* idx = (long long*)slab;
  double* val = (double*)(slab + n8);
  double* pv = (double*)(slab + 2 * n8);
  double* cn = (double*)(slab + 3 * n8);
  double* cols = (double*)(slab + 4 * n8);
  long long* idx2 = (long long*)(slab + 4 * n8 + ncols);
  double* val2 = (double*)(slab + 5 * n8 + ncols);
  double* cn2 = (double*)(slab + 6 * n8 + ncols);
  double* cols2 = (double*)(slab + 7 * n8 + ncols);
  if (h->d.cap > 0) {
    BC_HIP(hipStreamSynchronize(ctx->stream));
    BC_HIP(hipMemcpy(idx, h->d.idx, h->d.cap * sizeof(long long), hipMemcpyDeviceToDevice));
    BC_HIP(hipMemcpy(val, h->d.val, h->d.cap * sizeof(double), hipMemcpyDeviceToDevice));
    BC_HIP(hipMemcpy(cn, h->d.colnorm, h->d.cap * sizeof(double), hipMemcpyDeviceToDevice));
    BC_HIP(hipMemcpy(cols, h->d.cols, (size_t)h->d.cap * s * sizeof(double), hipMemcpyDeviceToDevice));
  }
  free_lists(h);
  h->list_slab = slab;
  h->d.idx = idx; h->idx2 = idx2;
  h->d.val = val; h->d.prev_val = pv; h->val2 = val2;
  h->d.colnorm = cn; h->colnorm2 = cn2;
  h->d.cols = cols; h->cols2 = cols2;
  h->d.cap = ncap;
  return BC_OK;
}

static int ensure_trace(bc_snnls* h, long long need) {
  if (need <= h->d.tr_cap) return BC_OK;
  long long ncap = std::max<long long>(need, std::max<long long>(1024, h->d.tr_cap * 2));
  long long* f = nullptr;
  int* st = nullptr;
  double* er = nullptr;
  BC_HIP(hipMalloc((void**)&f, ncap * sizeof(long long)));
  BC_HIP(hipMalloc((void**)&st, ncap * sizeof(int)));
  BC_HIP(hipMalloc((void**)&er, ncap * sizeof(double)));
  if (h->d.tr_cap > 0) {
    BC_HIP(hipStreamSynchronize(h->ctx->stream));
    BC_HIP(hipMemcpy(f, h->d.tr_f, h->d.tr_cap * sizeof(long long), hipMemcpyDeviceToDevice));
    BC_HIP(hipMemcpy(st, h->d.tr_status, h->d.tr_cap * sizeof(int), hipMemcpyDeviceToDevice));
    BC_HIP(hipMemcpy(er, h->d.tr_err, h->d.tr_cap * sizeof(double), hipMemcpyDeviceToDevice));
    (void)hipFree(h->d.tr_f); (void)hipFree(h->d.tr_status); (void)hipFree(h->d.tr_err);
  }
  h->d.tr_f = f; h->d.tr_status = st; h->d.tr_err = er; h->d.tr_cap = ncap;
  return BC_OK;
}

extern "C" int bc_snnls_destroy(bc_snnls* h) {
  if (!h) return BC_OK;
  (void)hipStreamSynchronize(h->ctx->stream);
  bc_pref_destroy(h->pref);
  h->pref = nullptr;
  free_lists(h);
  void* ptrs[] = {h->state_slab, h->cand_all_owned, h->d.tr_f, h->d.tr_status, h->d.tr_err};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete h;
  return BC_OK;
}

// Two-level pre-filter (bc_prefilter_i4.h) wanted?  BC_PREFILTER=4 forces it, =8 keeps the one-level int8 sweep; by default it
// is used from BC_TWO_LEVEL_MIN_ROWS rows: measured per step at S = 100 (tools/two_level_probe.py, us, two-level / one-level):
// 1.25M rows 43.4 / 42.5, 2.5M 53.6 / 62, 5M 72.5 / 102, 10M 114 / 177.  S <= 256.
#define BC_TWO_LEVEL_MIN_ROWS 2000000LL
static bool bc_want_two_level(int preq, long long n_rows, int s) {
  if (s > 256 || n_rows <= 0) return false;
  if (preq == 4) return true;
  if (preq >= 0 && preq != 1) return false;          // an explicit BC_PREFILTER precision other than 4
  const char* tenv = getenv("BC_TWO_LEVEL_MIN_ROWS");
  const long long min_rows = tenv ? atoll(tenv) : BC_TWO_LEVEL_MIN_ROWS;
  return n_rows >= min_rows;
}

#define LAUNCH1(kern, ...)                                                         \
  do {                                                                             \
    hipLaunchKernelGGL(kern, dim3(1), dim3(BC_FIN_THREADS), 0, h->ctx->stream, __VA_ARGS__);   \
    BC_HIP(hipGetLastError());                                                     \
  } while (0)

#define BY_ALG(kern, ...)                                                          \
  do {                                                                             \
    if (h->alg == BC_ALG_GIGA) LAUNCH1(kern<BC_ALG_GIGA>, __VA_ARGS__);             \
    else if (h->alg == BC_ALG_FW) LAUNCH1(kern<BC_ALG_FW>, __VA_ARGS__);            \
    else LAUNCH1(kern<BC_ALG_OMP>, __VA_ARGS__);                                    \
  } while (0)

extern "C" int bc_snnls_create(bc_ctx* ctx, bc_phi* phi, const double* b, int alg, double norm_sum,
                               int allow_zero_rows, bc_snnls** out) {
  if (!ctx || !phi || !b || !out || alg < 0 || alg > 2) { bc_set_error("bc_snnls_create: bad argument"); return BC_INVALID_ARGUMENT; }
  if (phi->ctx != ctx) { bc_set_error("bc_snnls_create: phi belongs to another context"); return BC_INVALID_ARGUMENT; }
  int64_t zr = 0;
  int rc = bc_phi_norm_stats(phi, &zr, nullptr);
  if (rc) return rc;
  if (zr > 0 && !allow_zero_rows) {
    bc_set_error("A must not have any 0 columns (%lld zero-norm rows)", (long long)zr);   // giga.py:11-12
    return BC_INVALID_ARGUMENT;
  }
  const int s = phi->s;
  double bnorm = 0.0;
  for (int k = 0; k < s; ++k) bnorm += b[k] * b[k];
  bnorm = sqrt(bnorm);
  if (alg == BC_ALG_GIGA && bnorm == 0.) {
    bc_set_error("norm of b must be > 0");   // giga.py:16-17
    return BC_NUMERICAL_PRECISION;
  }
  bc_snnls* h = new bc_snnls();
  h->ctx = ctx;
  h->phi = phi;
  h->alg = alg;
  memset(&h->d, 0, sizeof(h->d));
  SnnlsDev& d = h->d;
  d.s = s;
  d.world = 1;
  d.rec_len = s + BC_REC_HDR;
  d.bnorm = bnorm;
  d.tol = 1e-12;
  d.norm_sum = norm_sum;
  d.tiles = phi->tiles;
  d.norms = phi->norms;
  d.n_rows = phi->n_rows;
  d.row_offset = phi->row_offset;
  {
    // one slab (see bc_snnls::state_slab): st | b | bn | xw | xw_prev | xf | send record | v (zero tail: the fp16
    // sweep reads whole plane batches), every piece 256-byte aligned
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_st = 0, o_b = up(sizeof(SnnlsState)), o_bn = o_b + up(s * 8), o_xw = o_bn + up(s * 8), o_xwp = o_xw + up(s * 8),
                 o_xf = o_xwp + up(s * 8), o_rec = o_xf + up(s * 8), o_v = o_rec + up(d.rec_len * 8),
                 o_qv = o_v + up(2 * (s + BC_V_PAD) * 8), o_qv4 = o_qv + up((size_t)BC_I8Q_INTS(bc_lay_i8_sp4(s)) * sizeof(int)),
                 total = o_qv4 + up((size_t)BC_I4Q_INTS(bc_lay_i4_sp8(s, bc_lay_i4_batch(s))) * sizeof(int));
    char* slab = nullptr;
    hipError_t e0 = hipMalloc((void**)&slab, total);
    if (e0 == hipSuccess) e0 = hipMemsetAsync(slab, 0, total, ctx->stream);
    if (e0 != hipSuccess) { if (slab) (void)hipFree(slab); bc_snnls_destroy(h); return bc_hip_fail(e0, "hipMalloc(snnls)", __FILE__, __LINE__); }
    h->state_slab = slab;
    d.st = (SnnlsState*)(slab + o_st);
    d.b = (double*)(slab + o_b);
    d.bn = (double*)(slab + o_bn);
    d.xw = (double*)(slab + o_xw);
    d.xw_prev = (double*)(slab + o_xwp);
    d.xf = (double*)(slab + o_xf);
    h->cand_send = (double*)(slab + o_rec);
    d.v = (double*)(slab + o_v);
    // the int8 pre-filter (created below, after k_reset has produced the first sweep vector) reads v as digits: the step
    // kernels keep that record current from the start when such a pre-filter is going to exist
    const char* penv = getenv("BC_PREFILTER");
    const int preq = penv ? atoi(penv) : -1;
    const bool pwant = penv ? preq != 0 : phi->n_rows >= 163840;
    const int pprec = (preq == 8 || preq == 16 || preq == 32) ? preq : BC_PREF_DEFAULT_PREC;      // (4: the int8 mirror behind a 4-bit first level)
    const bool ptwo = bc_want_two_level(preq, phi->n_rows, s);
    const char* qenv = getenv("BC_I8_QV");               // =0: sweeps quantise in their own prologue (A/B, tests)
    d.sp4 = bc_lay_i8_sp4(s);
    d.qv = (pwant && pprec == 8 && phi->n_rows > 0 && (s + 3) / 4 <= BC_LAY_IMAXG - BC_LAY_IU && !(qenv && atoi(qenv) == 0)) ? (int*)(slab + o_qv) : nullptr;
    d.sp8 = bc_lay_i4_sp8(s, bc_lay_i4_batch(s));
    d.qv4 = (d.qv && ptwo) ? (int*)(slab + o_qv4) : nullptr;
  }
  hipError_t e = hipSuccess;
  d.cand_all = h->cand_send;
  rc = ensure_capacity(h, 256);
  if (!rc) rc = ensure_trace(h, 1024);
  if (rc) { bc_snnls_destroy(h); return rc; }
  // bn = b / ||b|| element-wise like giga.py:18
  std::vector<double> tmp(2 * s);
  for (int k = 0; k < s; ++k) {
    tmp[k] = b[k];
    tmp[s + k] = bnorm != 0. ? b[k] / bnorm : 0.;
  }
  e = hipMemcpyAsync(d.b, tmp.data(), s * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d.bn, tmp.data() + s, s * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(h->cand_send, 0, d.rec_len * sizeof(double), ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_reset, dim3(1), dim3(BC_FIN_THREADS), 0, ctx->stream, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { bc_snnls_destroy(h); return bc_hip_fail(e, "snnls init", __FILE__, __LINE__); }
  // reduced-precision pre-filter: worth its extra launches once the sweep is long enough.
  // BC_PREFILTER = 0 (off) / 1 (on, default precision) / 8 / 16 / 32 (on, that storage precision)
  const char* env = getenv("BC_PREFILTER");
  const int req = env ? atoi(env) : -1;
  const bool want = env ? req != 0 : phi->n_rows >= 163840;   // measured break-even ~131k rows at S = 100 (int8 mirror, profiles/r01_notes.md)
  if (want && phi->n_rows > 0) {
    const int cprec = (req == 8 || req == 16 || req == 32) ? req : BC_PREF_DEFAULT_PREC;
    rc = bc_pref_create(phi, (cprec == 8 && d.qv4) ? 4 : cprec, &h->pref);
    if (rc) { bc_snnls_destroy(h); return rc; }
    const char* cap = getenv("BC_PREFILTER_CAP");
    if (cap) bc_pref_set_cap(h->pref, atoi(cap));
    if (d.qv && bc_pref_sp4(h->pref) == d.sp4) bc_pref_set_qv(h->pref, d.qv);
    if (d.qv4 && bc_pref_two_level(h->pref) && bc_pref_sp8(h->pref) == d.sp8) bc_pref_set_qv4(h->pref, d.qv4);
  }
  *out = h;
  return BC_OK;
}

extern "C" int bc_snnls_prefilter_active(const bc_snnls* h, int* on) {
  if (!h || !on) return BC_INVALID_ARGUMENT;
  *on = h->pref ? bc_pref_precision(h->pref) : 0;   // 0 = off, else the storage precision (16 / 32)
  return BC_OK;
}

int bc_pref_bb(const bc_pref* p);
extern "C" int bc_snnls_prefilter_form(const bc_snnls* h, int* form) {
  if (!h || !form) { bc_set_error("bc_snnls_prefilter_form: bad argument"); return BC_INVALID_ARGUMENT; }
  *form = !h->pref ? 0 : (bc_pref_bb(h->pref) ? 2 : (bc_pref_two_level_active(h->pref) ? 3 : 1));      // (3 put aside by the watch: 1)
  return BC_OK;
}

extern "C" int bc_snnls_prefilter_fallbacks(const bc_snnls* h, int64_t* n) {
  if (!h || !n) return BC_INVALID_ARGUMENT;
  *n = 0;
  if (!h->pref) return BC_OK;
  int ctrl[4] = {0, 0, 0, 0};
  BC_HIP(hipMemcpyAsync(ctrl, bc_pref_ctrl(h->pref), sizeof(ctrl), hipMemcpyDeviceToHost, h->ctx->stream));
  BC_HIP(hipStreamSynchronize(h->ctx->stream));
  *n = ctrl[3];
  return BC_OK;
}

extern "C" int bc_snnls_prefilter_stats(const bc_snnls* h, int64_t* sweeps, int64_t* candidates, int64_t* fallbacks) {
  if (!h) return BC_INVALID_ARGUMENT;
  if (sweeps) *sweeps = 0;
  if (candidates) *candidates = 0;
  if (fallbacks) *fallbacks = 0;
  if (!h->pref) return BC_OK;
  int ctrl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  BC_HIP(hipMemcpyAsync(ctrl, bc_pref_ctrl(h->pref), sizeof(ctrl), hipMemcpyDeviceToHost, h->ctx->stream));
  BC_HIP(hipStreamSynchronize(h->ctx->stream));
  unsigned long long st[2];
  memcpy(st, ctrl + 4, sizeof(st));
  if (sweeps) *sweeps = (int64_t)st[0];
  if (candidates) *candidates = (int64_t)st[1];
  if (fallbacks) *fallbacks = ctrl[3];
  return BC_OK;
}

long long bc_pref_l1_sweeps(const bc_pref* p);
extern "C" int bc_snnls_prefilter_levels(const bc_snnls* h, int64_t* l1_sweeps, int64_t* listed, int64_t* refined) {
  if (!h) { bc_set_error("bc_snnls_prefilter_levels: bad argument"); return BC_INVALID_ARGUMENT; }
  if (l1_sweeps) *l1_sweeps = 0;
  if (listed) *listed = 0;
  if (refined) *refined = 0;
  if (!h->pref || !bc_pref_two_level(h->pref)) return BC_OK;
  int ctrl[16];
  BC_HIP(hipMemcpyAsync(ctrl, bc_pref_ctrl(h->pref), sizeof(ctrl), hipMemcpyDeviceToHost, h->ctx->stream));
  BC_HIP(hipStreamSynchronize(h->ctx->stream));
  unsigned long long st[2];
  memcpy(st, ctrl + 10, sizeof(st));
  if (l1_sweeps) *l1_sweeps = (int64_t)bc_pref_l1_sweeps(h->pref);
  (void)st[0];
  if (listed) *listed = (int64_t)st[1];            // (every row the first level passes on is re-bounded by the same wave)
  if (refined) *refined = (int64_t)st[1];
  return BC_OK;
}

extern "C" int bc_snnls_set_tolerance(bc_snnls* h, double tol) {
  if (!h) return BC_INVALID_ARGUMENT;
  h->d.tol = tol;
  return BC_OK;
}

extern "C" int bc_snnls_record_doubles(const bc_snnls* h, int32_t* n) {
  if (!h || !n) return BC_INVALID_ARGUMENT;
  *n = h->d.rec_len;
  return BC_OK;
}

extern "C" int bc_snnls_bind_exchange(bc_snnls* h, int world, void* cand_send_dev, void* cand_all_dev) {
  if (!h || world < 1 || (world > 1 && (!cand_send_dev || !cand_all_dev))) {
    bc_set_error("bc_snnls_bind_exchange: bad argument");
    return BC_INVALID_ARGUMENT;
  }
  BC_HIP(hipStreamSynchronize(h->ctx->stream));
  if (cand_send_dev && cand_all_dev) {
    h->cand_send = (double*)cand_send_dev;      // borrowed (typically torch tensors); the own record stays in the state slab
    h->cand_send_owned = false;
    h->d.cand_all = (const double*)cand_all_dev;
  } else {
    h->d.cand_all = h->cand_send;
  }
  h->d.world = world;
  return BC_OK;
}

extern "C" int bc_snnls_bind_comm(bc_snnls* h, bc_comm* c) {
  if (!h || !c) { bc_set_error("bc_snnls_bind_comm: bad argument"); return BC_INVALID_ARGUMENT; }
  if (!h->cand_send_owned || h->comm) { bc_set_error("bc_snnls_bind_comm: an exchange is already bound"); return BC_INVALID_ARGUMENT; }
  int32_t rank = 0, world = 1;
  bc_comm_info(c, &rank, &world);
  BC_HIP(hipStreamSynchronize(h->ctx->stream));
  const size_t bytes = (size_t)world * h->d.rec_len * sizeof(double);
  BC_HIP(hipMalloc((void**)&h->cand_all_owned, bytes));
  BC_HIP(hipMemsetAsync(h->cand_all_owned, 0, bytes, h->ctx->stream));
  h->d.cand_all = h->cand_all_owned;
  h->d.world = world;
  h->comm = c;
  return BC_OK;
}

// the record all-gather between the local sweep and the replicated finish / pick (native exchange only)
static int exchange(bc_snnls* h) {
  if (!h->comm) return BC_OK;
  int rc = bc_timer_begin(h->ctx, 4);
  if (!rc) rc = bc_comm_all_gather_dev(h->comm, h->cand_send, h->cand_all_owned, (size_t)h->d.rec_len);
  if (!rc) rc = bc_timer_end(h->ctx, 4);
  return rc;
}

static int mode_of(const bc_snnls* h) { return h->alg == BC_ALG_GIGA ? 0 : 1; }

static int launch_prep(bc_snnls* h, int reset_retry) {
  BY_ALG(k_prep, h->d, reset_retry);
  return BC_OK;
}

// the finish kernel may take the sweep's block candidates directly when nobody else needs the record
static bool fuse_winner(const bc_snnls* h) { return h->d.world == 1 && h->cand_send_owned && !h->pref && !h->comm; }
static bool use_pref(const bc_snnls* h) { return h->pref && !h->pref_suspended && !h->exact_step; }

#define BC_RS_MAX_DYN_LDS (48 * 1024)   // dynamic LDS the fused kernel may ask for (on top of ~42 KB static: rescoring strips, lists)
// one-round-of-loads finish kernel (k_step_finish_pf) possible for the step in flight?
static bool finish_pf_ok(const bc_snnls* h, long long nrec) {
  static const int no_pf = getenv("BC_FINISH_NOPF") ? atoi(getenv("BC_FINISH_NOPF")) : 0;
  return !no_pf && h->d.s <= BC_FIN_THREADS && h->nnz_upper + 1 <= BC_PF_MAXNNZ && nrec <= BC_PF_MAXREC && h->nnz_upper + 1 <= h->d.cap;
}
// ... with the rescoring stage inside the same launch (single rank, pre-filtered sweep)?
static bool fused_rescore_ok(const bc_snnls* h) {
  static const int no_fuse = getenv("BC_FINISH_NOFUSE") ? atoi(getenv("BC_FINISH_NOFUSE")) : 0;
  if (no_fuse || !use_pref(h) || h->d.world != 1 || h->comm || !h->cand_send_owned) return false;
  if (!finish_pf_ok(h, h->d.rec_len)) return false;
  const size_t lds = ((size_t)5 * h->d.s + h->d.rec_len + 2 * (size_t)(h->nnz_upper + 1)) * sizeof(double);
  // static LDS of the fused kernel (rescoring strips 32 KB, current / best row 16 KB, tile list 4 KB, ...) + the dynamic part
  // must fit the device's per-block limit (160 KB on gfx950; the two-launch path is taken otherwise)
  return lds <= BC_RS_MAX_DYN_LDS && lds + 64 * 1024 <= (size_t)h->ctx->max_lds;      // (63.5 KB static, of which 4 KB dev_prep copy of the sweep vectors and 2 KB image of the digit records)
}

static int launch_sweep(bc_snnls* h, bool with_record, bool allow_fused) {
  h->rs_pending = false;
  if (use_pref(h)) {
    // reduced-precision pre-filter -> candidates -> exact fp64 rescoring into the record
    if (allow_fused && fused_rescore_ok(h)) {
      h->rs_pending = true;       // step_finish runs rescoring + finish as one launch
      return bc_pref_launch_sweep(h->pref, mode_of(h), h->d.v, &h->d.st->v_norm, 1.0, &h->d.st->skip, h->cand_send, &h->rs);
    }
    return bc_pref_launch(h->pref, mode_of(h), h->d.v, &h->d.st->v_norm, 1.0, &h->d.st->skip, h->cand_send);
  }
  const bool rec = with_record || h->exact_step || h->pref != nullptr;     // (pref suspended: the finish reads the record)
  return bc_launch_sweep(h->phi, mode_of(h), h->d.v, 1.0, &h->d.st->skip, rec ? h->cand_send : nullptr);
}

__global__ void k_clear_pf_overflow(SnnlsState* st) {
  st->pf_overflow = 0;
  st->skip = st->select_fail | st->reached_limit;
}

// ---- fused loop
extern "C" int bc_snnls_build_begin(bc_snnls* h, int itrs) {
  if (!h || itrs < 0) { bc_set_error("bc_snnls_build_begin: bad argument"); return BC_INVALID_ARGUMENT; }
  if (h->alg == BC_ALG_OMP) {
    bc_set_error("bc_snnls_build*: OrthoPursuit refits with a host NNLS every step; use the step-wise protocol");
    return BC_INVALID_ARGUMENT;
  }
  int rc = ensure_capacity(h, h->nnz_upper + itrs);
  if (!rc) rc = ensure_trace(h, h->iter_upper + itrs);
  if (rc) return rc;
  h->pref_suspended = false;
  h->consec_overflow = 0;
  h->exact_step = false;
  return launch_prep(h, 1);
}

extern "C" int bc_snnls_step_local(bc_snnls* h) {
  if (!h) return BC_INVALID_ARGUMENT;
  h->exact_step = false;
  return launch_sweep(h, !fuse_winner(h), true);
}

extern "C" int bc_snnls_step_local_exact(bc_snnls* h) {
  if (!h) return BC_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_clear_pf_overflow, dim3(1), dim3(1), 0, h->ctx->stream, h->d.st);
  BC_HIP(hipGetLastError());
  h->exact_step = true;
  return launch_sweep(h, true, false);
}

extern "C" int bc_snnls_step_finish(bc_snnls* h) {
  if (!h) return BC_INVALID_ARGUMENT;
  SnnlsDev d = h->d;
  const bool rec_based = h->exact_step || h->pref != nullptr;
  d.fuse_winner = (fuse_winner(h) && !rec_based) ? 1 : 0;
  d.blk_val = h->phi->blk_val;
  d.blk_idx = h->phi->blk_idx;
  d.nblk = h->phi->sweep_blocks;
  const bool fused = h->rs_pending;
  h->rs_pending = false;
  h->exact_step = false;
  const long long nrec = fused ? d.rec_len : (d.fuse_winner ? 0 : (long long)d.world * d.rec_len);
  int rct = bc_timer_begin(h->ctx, 5);
  if (rct) return rct;
  if (fused || finish_pf_ok(h, nrec)) {
    const int hint = (int)h->nnz_upper;
    const size_t lds = ((size_t)5 * d.s + nrec + 2 * (size_t)(hint + 1)) * sizeof(double);
    const long long n_rows = h->phi->n_rows;
    if (fused) {
      static unsigned attr_done_mask = 0;  // 42 KB static + up to 48 KB dynamic LDS: above the 64 KB default limit; per device
      const bool attr_done = (attr_done_mask >> (h->ctx->device & 31)) & 1u;
      if (!attr_done) {
        BC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_finish_pf<BC_ALG_GIGA, 1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, BC_RS_MAX_DYN_LDS));
        BC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_finish_pf<BC_ALG_FW, 1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, BC_RS_MAX_DYN_LDS));
        BC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_finish_pf<BC_ALG_GIGA, 2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, BC_RS_MAX_DYN_LDS));
        BC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_finish_pf<BC_ALG_FW, 2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, BC_RS_MAX_DYN_LDS));
        attr_done_mask |= 1u << (h->ctx->device & 31);
      }
      const bool bb = h->rs.bb.rec != nullptr;      // the sweep in flight was the branch-and-bound one (bc_prefilter_bb.h)
      if (h->alg == BC_ALG_GIGA) {
        if (bb) hipLaunchKernelGGL((k_step_finish_pf<BC_ALG_GIGA, 2>), dim3(1), dim3(BC_FIN_THREADS), lds, h->ctx->stream, d, hint, h->rs, n_rows);
        else hipLaunchKernelGGL((k_step_finish_pf<BC_ALG_GIGA, 1>), dim3(1), dim3(BC_FIN_THREADS), lds, h->ctx->stream, d, hint, h->rs, n_rows);
      } else {
        if (bb) hipLaunchKernelGGL((k_step_finish_pf<BC_ALG_FW, 2>), dim3(1), dim3(BC_FIN_THREADS), lds, h->ctx->stream, d, hint, h->rs, n_rows);
        else hipLaunchKernelGGL((k_step_finish_pf<BC_ALG_FW, 1>), dim3(1), dim3(BC_FIN_THREADS), lds, h->ctx->stream, d, hint, h->rs, n_rows);
      }
    } else {
      if (h->alg == BC_ALG_GIGA)
        hipLaunchKernelGGL((k_step_finish_pf<BC_ALG_GIGA, 0>), dim3(1), dim3(BC_FIN_THREADS), lds, h->ctx->stream, d, hint, h->rs, n_rows);
      else
        hipLaunchKernelGGL((k_step_finish_pf<BC_ALG_FW, 0>), dim3(1), dim3(BC_FIN_THREADS), lds, h->ctx->stream, d, hint, h->rs, n_rows);
    }
    BC_HIP(hipGetLastError());
    h->nnz_upper += 1;
    h->iter_upper += 1;
    return bc_timer_end(h->ctx, 5);
  }
  const int lds_vecs = d.s <= 1024 ? 1 : 0;
  const size_t lds = lds_vecs ? (size_t)5 * d.s * sizeof(double) : 0;
  if (h->alg == BC_ALG_GIGA)
    hipLaunchKernelGGL(k_step_finish<BC_ALG_GIGA>, dim3(1), dim3(BC_FIN_THREADS), lds, h->ctx->stream, d, lds_vecs);
  else
    hipLaunchKernelGGL(k_step_finish<BC_ALG_FW>, dim3(1), dim3(BC_FIN_THREADS), lds, h->ctx->stream, d, lds_vecs);
  BC_HIP(hipGetLastError());
  h->nnz_upper += 1;
  h->iter_upper += 1;
  return bc_timer_end(h->ctx, 5);
}

extern "C" int bc_snnls_build_end(bc_snnls* h, int* reached_numeric_limit, int* iterations_consumed, int* pending_exact) {
  if (!h) return BC_INVALID_ARGUMENT;
  SnnlsState st;
  int rc = fetch_state(h, &st);
  if (rc) return rc;
  h->nnz_upper = st.nnz;
  h->iter_upper = st.iter;
  if (h->pref) {                                  // (the stream is idle: the two-level form's watch costs one small copy)
    rc = bc_pref_adapt(h->pref);
    if (rc) return rc;
  }
  if (reached_numeric_limit) *reached_numeric_limit = st.reached_limit;
  if (iterations_consumed) *iterations_consumed = (int)st.iter;
  if (pending_exact) *pending_exact = st.pf_overflow;
  if (st.pf_overflow) {
    // all ranks see the same replicated state, so they take this branch together
    if (++h->consec_overflow >= 3) h->pref_suspended = true;     // e.g. a design full of exact duplicates: stop trying
  } else {
    h->consec_overflow = 0;
  }
  return BC_OK;
}

extern "C" int bc_snnls_build(bc_snnls* h, int itrs, int* reached_numeric_limit) {
  if (!h) return BC_INVALID_ARGUMENT;
  if (h->d.world != 1 && !h->comm) {
    bc_set_error("bc_snnls_build: world > 1 without a bound bc_comm: the host must all-gather between step_local and step_finish");
    return BC_INVALID_ARGUMENT;
  }
  const long long iter0 = h->iter_upper;
  int rc = bc_snnls_build_begin(h, itrs);
  if (rc) return rc;
  int left = itrs, lim = 0;
  while (true) {
    for (int i = 0; i < left && !rc; ++i) {
      rc = bc_snnls_step_local(h);
      if (!rc) rc = exchange(h);
      if (!rc) rc = bc_snnls_step_finish(h);
      // a long call: let the two-level form's watch look every 64 steps (a stream synchronisation: ~0.5 us per step)
      if (!rc && h->pref && bc_pref_two_level(h->pref) && (i & 63) == 63 && i + 1 < left) rc = bc_pref_adapt(h->pref);
    }
    if (rc) return rc;
    int consumed = 0, pending = 0;
    rc = bc_snnls_build_end(h, &lim, &consumed, &pending);
    if (rc) return rc;
    if (!pending) break;
    // the pre-filter overflowed at step `consumed - iter0`: that step and everything enqueued after it were no-ops.
    // Redo it through the exact sweep, then go on with what is left.
    rc = bc_snnls_step_local_exact(h);
    if (!rc) rc = exchange(h);
    if (!rc) rc = bc_snnls_step_finish(h);
    if (rc) return rc;
    left = itrs - (int)(consumed - iter0) - 1;
    if (left <= 0) {
      rc = bc_snnls_build_end(h, &lim, nullptr, nullptr);
      if (rc) return rc;
      break;
    }
  }
  if (reached_numeric_limit) *reached_numeric_limit = lim;
  return BC_OK;
}

// ---- step-wise protocol
extern "C" int bc_snnls_select_local(bc_snnls* h) {
  if (!h) return BC_INVALID_ARGUMENT;
  int rc = launch_prep(h, 0);
  if (rc) return rc;
  h->exact_step = false;
  h->pref_suspended = false;
  return launch_sweep(h, true, false);
}

extern "C" int bc_snnls_select_local_exact(bc_snnls* h) {
  if (!h) return BC_INVALID_ARGUMENT;
  h->exact_step = true;
  const int rc = launch_sweep(h, true, false);
  h->exact_step = false;
  return rc;
}

extern "C" int bc_snnls_select_pick(bc_snnls* h, int64_t* f) {
  if (!h || !f) return BC_INVALID_ARGUMENT;
  BY_ALG(k_pick, h->d);
  SnnlsState st;
  int rc = fetch_state(h, &st);
  if (rc) return rc;
  *f = st.sel_f;
  if (st.last_status == BC_RETRY_EXACT) return BC_RETRY_EXACT;                   // no error text: the caller re-sweeps exactly
  if (st.last_status == BC_NUMERICAL_PRECISION) bc_set_error("cdirnrm < TOL");   // giga.py:28-29
  if (st.last_status == BC_INVALID_ARGUMENT) bc_set_error("bc_snnls_select: no selectable row");
  return st.last_status;
}

extern "C" int bc_snnls_select(bc_snnls* h, int64_t* f) {
  if (!h || !f) return BC_INVALID_ARGUMENT;
  if (h->d.world != 1 && !h->comm) {
    bc_set_error("bc_snnls_select: world > 1 without a bound bc_comm: use select_local / all-gather / select_pick");
    return BC_INVALID_ARGUMENT;
  }
  int rc = bc_snnls_select_local(h);
  if (!rc) rc = exchange(h);
  if (rc) return rc;
  rc = bc_snnls_select_pick(h, f);
  if (rc != BC_RETRY_EXACT) return rc;
  rc = bc_snnls_select_local_exact(h);          // a pre-filter overflowed on some rank: the same step, exact sweep
  if (!rc) rc = exchange(h);
  if (rc) return rc;
  return bc_snnls_select_pick(h, f);
}

extern "C" int bc_snnls_reweight(bc_snnls* h, int64_t f) {
  if (!h) return BC_INVALID_ARGUMENT;
  if (h->alg == BC_ALG_OMP) {
    bc_set_error("bc_snnls_reweight: OrthoPursuit reweights on the host (orthopursuit.py:37-42); use set_weights");
    return BC_INVALID_ARGUMENT;
  }
  int rc = ensure_capacity(h, h->nnz_upper + 1);
  if (rc) return rc;
  if (h->alg == BC_ALG_GIGA) LAUNCH1(k_reweight<BC_ALG_GIGA>, h->d, (long long)f);
  else LAUNCH1(k_reweight<BC_ALG_FW>, h->d, (long long)f);
  SnnlsState st;
  rc = fetch_state(h, &st);
  if (rc) return rc;
  h->nnz_upper = st.nnz;
  if (st.last_status == BC_NUMERICAL_PRECISION) bc_set_error("precision loss in the closed-form step");
  if (st.last_status == BC_INVALID_ARGUMENT) bc_set_error("bc_snnls_reweight: column %lld is not available on this rank", (long long)f);
  return st.last_status;
}

extern "C" int bc_snnls_error(bc_snnls* h, double* err) {
  if (!h || !err) return BC_INVALID_ARGUMENT;
  SnnlsState st;
  int rc = fetch_state(h, &st);
  if (rc) return rc;
  *err = st.err_cur;
  return BC_OK;
}

extern "C" int bc_snnls_size(bc_snnls* h, int64_t* n) {
  if (!h || !n) return BC_INVALID_ARGUMENT;
  SnnlsState st;
  int rc = fetch_state(h, &st);
  if (rc) return rc;
  *n = st.npos;
  return BC_OK;
}

extern "C" int bc_snnls_weights(bc_snnls* h, int64_t cap, int64_t* idx, double* val, int64_t* n) {
  if (!h || !n) return BC_INVALID_ARGUMENT;
  SnnlsState st;
  int rc = fetch_state(h, &st);
  if (rc) return rc;
  *n = st.nnz;
  if (st.nnz == 0 || (!idx && !val)) return BC_OK;
  if (cap < st.nnz) { bc_set_error("bc_snnls_weights: capacity %lld < nnz %lld", (long long)cap, (long long)st.nnz); return BC_INVALID_ARGUMENT; }
  if (idx) BC_HIP(hipMemcpyAsync(idx, h->d.idx, st.nnz * sizeof(long long), hipMemcpyDeviceToHost, h->ctx->stream));
  if (val) BC_HIP(hipMemcpyAsync(val, h->d.val, st.nnz * sizeof(double), hipMemcpyDeviceToHost, h->ctx->stream));
  BC_HIP(hipStreamSynchronize(h->ctx->stream));
  return BC_OK;
}

extern "C" int bc_snnls_columns(bc_snnls* h, int64_t cap, double* cols, int64_t* n) {
  if (!h || !n) return BC_INVALID_ARGUMENT;
  SnnlsState st;
  int rc = fetch_state(h, &st);
  if (rc) return rc;
  *n = st.nnz;
  if (st.nnz == 0 || !cols) return BC_OK;
  if (cap < st.nnz) { bc_set_error("bc_snnls_columns: capacity too small"); return BC_INVALID_ARGUMENT; }
  BC_HIP(hipMemcpyAsync(cols, h->d.cols, (size_t)st.nnz * h->d.s * sizeof(double), hipMemcpyDeviceToHost, h->ctx->stream));
  BC_HIP(hipStreamSynchronize(h->ctx->stream));
  return BC_OK;
}

extern "C" int bc_snnls_set_weights(bc_snnls* h, int64_t n, const int64_t* idx, const double* val, const double* cols) {
  if (!h || n < 0 || (n > 0 && (!idx || !val))) { bc_set_error("bc_snnls_set_weights: bad argument"); return BC_INVALID_ARGUMENT; }
  bc_ctx* ctx = h->ctx;
  int rc = ensure_capacity(h, std::max<long long>(n, h->nnz_upper));
  if (rc) return rc;
  const int s = h->d.s;
  long long* didx = nullptr;
  double *dval = nullptr, *dcols = nullptr;
  hipError_t e = hipSuccess;
  if (n > 0) {
    e = hipMalloc((void**)&didx, n * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc((void**)&dval, n * sizeof(double));
    if (e == hipSuccess && cols) e = hipMalloc((void**)&dcols, (size_t)n * s * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(didx, idx, n * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dval, val, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && cols) e = hipMemcpyAsync(dcols, cols, (size_t)n * s * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_set_weights, dim3(1), dim3(BC_FIN_THREADS), 0, ctx->stream, h->d, (long long)n, didx, dval, dcols, h->idx2,
                       h->val2, h->cols2, h->colnorm2);
    e = hipGetLastError();
  }
  SnnlsState st;
  if (e == hipSuccess) rc = fetch_state(h, &st);
  if (didx) (void)hipFree(didx);
  if (dval) (void)hipFree(dval);
  if (dcols) (void)hipFree(dcols);
  if (e != hipSuccess) return bc_hip_fail(e, "bc_snnls_set_weights", __FILE__, __LINE__);
  if (rc) return rc;
  std::swap(h->d.idx, h->idx2);
  std::swap(h->d.val, h->val2);
  std::swap(h->d.cols, h->cols2);
  std::swap(h->d.colnorm, h->colnorm2);
  h->nnz_upper = n;
  if (st.last_status != BC_OK) {
    bc_set_error("bc_snnls_set_weights: a column was not supplied and is not available on this rank");
    return st.last_status;
  }
  return BC_OK;
}

extern "C" int bc_snnls_reset(bc_snnls* h) {
  if (!h) return BC_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_reset, dim3(1), dim3(BC_FIN_THREADS), 0, h->ctx->stream, h->d);
  BC_HIP(hipGetLastError());
  BC_HIP(hipStreamSynchronize(h->ctx->stream));
  h->nnz_upper = 0;
  h->iter_upper = 0;
  return BC_OK;
}

extern "C" int bc_snnls_get_flags(bc_snnls* h, int* reached_numeric_limit) {
  if (!h) return BC_INVALID_ARGUMENT;
  SnnlsState st;
  int rc = fetch_state(h, &st);
  if (rc) return rc;
  if (reached_numeric_limit) *reached_numeric_limit = st.reached_limit;
  return BC_OK;
}

__global__ void k_set_flag(SnnlsState* st, int reached) {
  st->reached_limit = reached;
  st->skip = st->select_fail | reached;
}

extern "C" int bc_snnls_set_flags(bc_snnls* h, int reached_numeric_limit) {
  if (!h) return BC_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, h->ctx->stream, h->d.st, reached_numeric_limit ? 1 : 0);
  BC_HIP(hipGetLastError());
  BC_HIP(hipStreamSynchronize(h->ctx->stream));
  return BC_OK;
}

extern "C" int bc_snnls_trace(bc_snnls* h, int64_t cap, int64_t* f, int32_t* status, double* err, int64_t* n) {
  if (!h || !n) return BC_INVALID_ARGUMENT;
  SnnlsState st;
  int rc = fetch_state(h, &st);
  if (rc) return rc;
  long long m = std::min<long long>(st.iter, h->d.tr_cap);
  *n = m;
  if (m == 0 || (!f && !status && !err)) return BC_OK;
  if (cap < m) { bc_set_error("bc_snnls_trace: capacity too small"); return BC_INVALID_ARGUMENT; }
  hipStream_t sm = h->ctx->stream;
  if (f) BC_HIP(hipMemcpyAsync(f, h->d.tr_f, m * sizeof(long long), hipMemcpyDeviceToHost, sm));
  if (status) BC_HIP(hipMemcpyAsync(status, h->d.tr_status, m * sizeof(int), hipMemcpyDeviceToHost, sm));
  if (err) BC_HIP(hipMemcpyAsync(err, h->d.tr_err, m * sizeof(double), hipMemcpyDeviceToHost, sm));
  BC_HIP(hipStreamSynchronize(sm));
  return BC_OK;
}
