// K3 pre-filter, int8 mirror, BRANCH-AND-BOUND form (round 5): the exact fp64 rescoring of the candidates runs INSIDE the
// sweep, beside the stream, instead of after it.
//
// The two-pass form (bc_prefilter_i8.h + bc_rescore_dev.h) streams the mirror, leaves per-tile bounds behind, and a single
// block then walks a chain of dependent round trips: Lmax over the blocks -> tiles in play -> candidate rows -> fetch each
// candidate's fp64 row (a cold TLB miss into the 8 GB Phi: ~3 us) -> the sequential fma chain -> winner.  In-kernel stamps
// (tools/fin_stamps.py, profiles/r05_notes.md) put that chain at 20k of the 45k cycles of a greedy step's tail.
//
// Here every sweep block carries one wave more than it has streaming waves.  A streaming wave that finishes a tile POSTS the rows whose upper bound
// reaches theta -- a running lower bound on the best exact score anywhere -- into an LDS mailbox and streams on; the fifth
// wave takes the posts, fetches the fp64 row, runs exactly the chain of k_sweep (bc_exact_score_wave: the bits of the fp64
// sweep) and keeps the block's best (score, row, norm, column).  theta = max of every lower bound L_i seen so far and of
// every exact score computed so far: per block in LDS, per device in one 64-bit word (sweep sequence number in the high
// half, so it never needs a reset; relaxed agent-scope atomic max / load: a stale value only means a few more posts).
//   * The row r* the fp64 sweep would return is never lost: U(r*) >= f(r*) >= every exact score and every L_i >= theta
//     at any time, so r* is posted, rescored by its block, and is that block's best under the (score, lowest index) rule.
//   * The consumer (bc_bb_pick: the fused step kernel, or k_bb_winner in front of an exchange) takes the argmax of the
//     per-block records -- one round of loads -- and copies the winner's column from the compact per-block slot.
//   * The first tile of a wave is held back one tile (theta is -inf until the first round of tiles has been pushed: every
//     wave would post its own first-tile best); afterwards posts are immediate.
//   * Overflow (more than BC_BB_QCAP posts or `max_res` exact rescorings in one block: thousands of duplicated rows) sets
//     a flag in the block record; the consumer turns it into the existing "redo this step with the exact fp64 sweep".
// MEASURED (round 5, N = 10M, S = 100, profiles/r05_notes.md) and NOT the default: the step kernel's tail drops from 45k to
// 34k cycles (-4.6 us) as intended, but the sweep itself grows from 157 us to 165 us with blocks that share nothing but exact
// scores (one rescoring per block and sweep), to 172 us when they also forward their lower bounds, and to 205-224 us when
// they poll the device-wide word -- posts of a wave's last tile are rescored after the stream has ended, and every form of
// sharing the bound costs more than it saves.  Opt-in: BC_I8_BB=1; covered by tests/test_gpu_prefilter.py ('bb').
#pragma once

#define BC_BB_QCAP 256         // posts a block can take per sweep
#define BC_BB_MAXRES 48        // exact rescorings a block may run per sweep
#ifndef BC_BB_SW
#define BC_BB_SW 8             // streaming waves per block: with the rescoring wave nine waves, ONE block per CU (3 + 2 + 2 + 2 on its
                               // SIMDs at 161 VGPRs; two five-wave blocks do not fit side by side and the sweep ran at half rate)
#endif
#define BC_BB_THREADS (64 * (BC_BB_SW + 1))
#ifndef BC_BB_MINWAVES
#define BC_BB_MINWAVES 1
#endif
#ifndef BC_BB_STAGGER
#define BC_BB_STAGGER 0
#endif

struct I8BbArgs {
  I8Args a;                    // mirror, digits, v, skip flag (the per-tile outputs are unused)
  const double* tiles;         // fp64 Phi
  const double* norms;
  long long row_offset;
  unsigned long long* theta;   // (seq << 32) | key of the device-wide lower bound
  unsigned seq;
  BbRec* blk_rec;              // [grid]
  double* blk_col;             // [grid][S] the best candidate's row
  int max_res;
};

// order-preserving map float -> unsigned (0 is below every real value and means "none")
__device__ __forceinline__ unsigned bc_f32_key(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float bc_key_f32(unsigned k) {
  if (k == 0u) return -INFINITY;
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float bc_bb_theta_load(const unsigned long long* theta, unsigned seq) {
  const unsigned long long g = __hip_atomic_load(theta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return ((unsigned)(g >> 32) == seq) ? bc_key_f32((unsigned)g) : -INFINITY;
}
__device__ __forceinline__ void bc_bb_theta_push(unsigned long long* theta, unsigned seq, unsigned key) {
  (void)__hip_atomic_fetch_max(theta, ((unsigned long long)seq << 32) | key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(BC_BB_THREADS, BC_BB_MINWAVES) void k_sweep_i8_bb(I8BbArgs b) {
  const I8Args& a = b.a;
  constexpr int NV = (MODE == 0) ? 3 : 2;
  __shared__ __attribute__((aligned(16))) int dig[BC_IMAXG][4];
  __shared__ double vmx[2][BC_BB_SW + 1];
  __shared__ int2 mbox[BC_BB_QCAP];            // (upper bound bits, local row); row < 0: not written yet
  __shared__ int q_head, done_cnt, q_ovf;
  __shared__ unsigned s_theta;
  __shared__ __attribute__((aligned(16))) double strip[4 * 256];
  __shared__ double currow[256], bestrow[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool skip = a.skip_flag != nullptr && *a.skip_flag != 0;
  if (skip) return;
  const int S = a.s, SP4 = a.sp4;
  constexpr int U = BC_IU;
  const bool streamer = wave < BC_BB_SW;
  // the first tile's loads do not depend on the prologue: put them in flight before it
  long long t = (long long)blockIdx.x * BC_BB_SW + wave;
  bc_i4 x[U], y[U];
  bc_hq8 rq = {(_Float16)0.f, (_Float16)-1.f, (_Float16)0.f, (_Float16)-1.f, (_Float16)0.f, (_Float16)-1.f, (_Float16)0.f, (_Float16)-1.f};
  if (streamer && t < a.ptiles) {
    const bc_i4* __restrict__ p0 = reinterpret_cast<const bc_i4*>(a.u8 + (size_t)t * SP4 * BC_ITILE) + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p0 + (size_t)u * 64);
    rq = reinterpret_cast<const bc_hq8*>(a.rowq + t * BC_ITILE)[lane];
  }
  if (threadIdx.x < BC_BB_QCAP) mbox[threadIdx.x] = make_int2(0, -1);
  if (threadIdx.x == 0) { q_head = 0; done_cnt = 0; q_ovf = 0; s_theta = 0u; }
  // ---- prologue: the digits of the sweep vector (as k_sweep_i8)
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
    double vm0 = vmx[0][0], vm1 = vmx[1][0];
#pragma unroll
    for (int w = 1; w <= BC_BB_SW; ++w) { vm0 = fmax(vm0, vmx[0][w]); vm1 = fmax(vm1, vmx[1][w]); }
    const bc_i8q_scalars q = bc_i8q_steps(vm0, vm1);
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

  if (!streamer) {
    // ------------------------------------------------------------ the rescoring wave
    // It is also the block's only link to the device-wide bound.  Measured (profiles/r05_notes.md): a load of that word in the
    // streaming waves' tile loop drains their loads in flight (vmcnt retires in order), and 256 rescoring waves POLLING the
    // one line (agent-scope loads bypass the L2s) make its memory channel the slowest of the stream -- 157 -> 205-224 us per
    // sweep.  So: the streamers talk to the block's LDS word alone; this wave forwards that word upwards whenever it grew
    // (a fire-and-forget atomic max), and READS the device's word only when it is about to spend a rescoring on a row.
    int tail = 0, nres = 0, flags = 0;
    double bv = -INFINITY, bnorm = 0.;
    long long bi = LLONG_MAX;
    unsigned pushed = 0u;                                    // the largest key this block has sent up
    for (;;) {
      const int done = *(volatile int*)&done_cnt;           // BEFORE the head: once all streamers are done it is final
      int head = *(volatile int*)&q_head;
      head = head < BC_BB_QCAP ? head : BC_BB_QCAP;
      {
        const unsigned lk = *(volatile unsigned*)&s_theta;
        if (lk > pushed) {
          pushed = lk;
          if (lane == 0) bc_bb_theta_push(b.theta, b.seq, lk);
        }
      }
      if (tail < head) {
        const int row = *(volatile int*)&mbox[tail].y;
        if (row < 0) { __builtin_amdgcn_s_sleep(1); continue; }      // slot reserved, entry not written yet
        const float ub = __int_as_float(*(volatile int*)&mbox[tail].x);
        ++tail;
        float th = bc_key_f32(*(volatile unsigned*)&s_theta);
        if (!(ub >= th)) continue;                            // overtaken since it was posted
        const float gth = bc_bb_theta_load(b.theta, b.seq);   // what the other blocks know by now
        if (gth > th) {
          th = gth;
          if (lane == 0) atomicMax(&s_theta, bc_f32_key(gth));
          if (!(ub >= th)) continue;
        }
        if (nres >= b.max_res) { flags |= BC_BB_FLAG_OVF; continue; }
        ++nres;
        const long long r = row;
        const double nr = b.norms[r];
        const double sc = bc_exact_score_wave<MODE, 4>(b.tiles, a.v, r, S, nr, a.post_div, strip, currow);      // (4: this kernel's register budget is the stream's)
        const long long gi = b.row_offset + r;
        if (bc_better(sc, gi, bv, bi)) {
          bv = sc; bi = gi; bnorm = nr;
          for (int k = lane; k < S; k += BC_WAVE) bestrow[k] = currow[k];
        }
        if (sc == sc && fabs(sc) < INFINITY) {                // an exact score is a lower bound on the maximum
          const float fl = __double2float_rd(sc);
          const unsigned key = bc_f32_key(fl);
          if (lane == 0) {
            atomicMax(&s_theta, key);
            if (fl > th) bc_bb_theta_push(b.theta, b.seq, key);
          }
          if (fl > th && key > pushed) pushed = key;
        }
      } else if (done == BC_BB_SW) {
        break;
      } else {
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) {
      BbRec r;
      r.score = bv;
      r.gidx = bi;
      r.norm = bnorm;
      r.nres = nres;
      r.flags = flags | ((*(volatile int*)&q_ovf) ? BC_BB_FLAG_OVF : 0);
      b.blk_rec[blockIdx.x] = r;
    }
    if (bi != LLONG_MAX)
      for (int k = lane; k < S; k += BC_WAVE) b.blk_col[(size_t)blockIdx.x * S + k] = bestrow[k];
    return;
  }

  // -------------------------------------------------------------- the streaming waves
#if BC_BB_STAGGER > 0
  for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(BC_BB_STAGGER);      // (experiment: take the block's waves out of lock-step)
#endif
  const float fpd = (float)a.post_div;
  const long long tstride = (long long)gridDim.x * BC_BB_SW;
  float best_l = -INFINITY;                            // this wave's best lower bound so far (what it has pushed)
  float pU[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // the first tile's upper bounds, held back one tile
  long long pt = -1;
  auto post = [&](const float (&ub)[4], long long tile, float th) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool c = ub[j] >= th && ub[j] != -INFINITY;
      const unsigned long long m = __ballot(c);
      if (m != 0ull) {                                 // (wave-uniform)
        int base = 0;
        if (lane == 0) base = atomicAdd(&q_head, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (c) {
          const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
          if (slot < BC_BB_QCAP) mbox[slot] = make_int2(__float_as_int(ub[j]), (int)(tile * BC_ITILE + 4 * lane + j));
          else q_ovf = 1;
        }
      }
    }
  };
  for (; t < a.ptiles; t += tstride) {
    const bc_i4* __restrict__ p = reinterpret_cast<const bc_i4*>(a.u8 + (size_t)t * SP4 * BC_ITILE) + lane;
    int acc[4][NV];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < NV; ++c) acc[j][c] = 0;
    const bc_hq8 cq = rq;
    for (int g0 = 0; g0 < SP4; g0 += U) {
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
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bc_i4 dg = *reinterpret_cast<const bc_i4*>(&dig[g0 + u][0]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c = 0; c < NV; ++c) acc[j][c] = __builtin_amdgcn_sdot4(x[u][j], dg[c], acc[j][c], false);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = y[u];
    }
    // ---- per-row intervals (4 rows per lane): the arithmetic of k_sweep_i8
    const float sc[4] = {(float)cq[0], (float)cq[2], (float)cq[4], (float)cq[6]};
    const float dl[4] = {(float)cq[1], (float)cq[3], (float)cq[5], (float)cq[7]};
    float Ub[4], Lb[4];
    float tl = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Ub[j] = -INFINITY;
      Lb[j] = -INFINITY;
      if (!(dl[j] < 0.f)) {
        const float dr = dl[j];
        const float s0 = sc[j] * fvs0 * (128.f * (float)acc[j][0] + (float)acc[j][1]);
        const float s1 = (MODE == 0) ? sc[j] * fvs1 * (float)acc[j][NV - 1] : 0.f;
        const float delta0 = (dr * fvn + (1.f + dr) * fev0) * 1.00001f + 4e-7f * fabsf(s0) + 1e-12f;
        const float delta1 = (MODE == 0) ? (dr * fvn + (1.f + dr) * fev1) * 1.00001f + 4e-7f * fabsf(s1) + 1e-12f : 0.f;
        if (vbad || dr != dr) { Ub[j] = INFINITY; Lb[j] = -INFINITY; }
        else bc_score_interval_f32<MODE>(s0, s1, delta0, delta1, fpd, Ub[j], Lb[j]);
        tl = fmaxf(tl, Lb[j]);
      }
    }
    tl = bc_wave_max_f32_all(tl);
    if (tl > best_l) {                                   // (wave-uniform) a better lower bound: tell the block (its rescoring wave
      best_l = tl;                                       // passes it on to the device)
      if (lane == 0) atomicMax(&s_theta, bc_f32_key(tl));
    }
    const float th = fmaxf(best_l, bc_key_f32(*(volatile unsigned*)&s_theta));
    if (pt < 0) {
      // the wave's first tile: theta has seen nothing but this tile -- hold its rows back until the next tile ends
      pt = t;
#pragma unroll
      for (int j = 0; j < 4; ++j) pU[j] = Ub[j];
    } else {
      if (pt != LLONG_MAX) {
        post(pU, pt, th);
        pt = LLONG_MAX;
      }
      post(Ub, t, th);
    }
  }
  if (pt >= 0 && pt != LLONG_MAX) {
    // a wave with a single tile: nothing more to wait for
    const float th = fmaxf(best_l, bc_key_f32(*(volatile unsigned*)&s_theta));
    post(pU, pt, th);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane == 0) atomicAdd(&done_cnt, 1);
}

// in front of an exchange (multi-rank steps) and in the step-wise protocol: the record goes to global memory
__global__ __launch_bounds__(256) void k_bb_winner(BbArgs a, int s, const int* skip_flag, int* ctrl, double* rec) {
  if (skip_flag != nullptr && *skip_flag != 0) return;
  const BbPre pre = bc_bb_prefetch(a);
  if (bc_bb_pick(a, pre, s, ctrl, rec)) {
    if (threadIdx.x == 0) {
      rec[0] = -INFINITY;
      reinterpret_cast<long long*>(rec)[1] = -1;
      rec[2] = 0.0;
      rec[3] = BC_REC_OVERFLOW;
    }
  }
}
