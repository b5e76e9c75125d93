/*
 * beta_cores.h -- C ABI of the MI355X (gfx950) sparse-NNLS coreset hot path.
 *
 * The reference (dionman/beta-cores, package `bayesiancoresets`) is pure
 * Python/NumPy and has no FFI of its own; its plug-in boundary is the duck-typed
 * class protocol  SparseNNLS / Projector / Coreset.  This header is the layer a
 * maintainer would bind (ctypes stub: INTEGRATION.md) underneath those classes.
 * Each entry point names the reference call site(s) it replaces.
 *
 * Conventions
 *   - every function returns an int status:
 *        BC_OK                   0
 *        BC_NUMERICAL_PRECISION  1   (host raises NumericalPrecisionError,
 *                                     bayesiancoresets/util/errors.py:1)
 *        BC_INVALID_ARGUMENT     2   (host raises ValueError)
 *        < 0                         HIP failure (host raises RuntimeError)
 *     bc_last_error() returns a thread-local message for the last non-OK status.
 *   - host pointers are borrowed for the duration of the call only; device
 *     memory is owned by the handles; handles are not thread-safe.
 *   - all arithmetic is IEEE double (the reference computes in float64);
 *     indices are int64 GLOBAL row numbers (row_offset + local row).
 *   - one context == one GPU == one process (ranks are separate processes).
 *     The per-step candidate exchange is issued by THIS library: ncclAllGather
 *     (RCCL) on the context's stream, from inside bc_snnls_build /
 *     bc_snnls_select, once a communicator is bound with bc_snnls_bind_comm
 *     (bc_comm_* below).  bc_snnls_bind_exchange is the second transport: the
 *     host moves the records between step_local / step_finish (gloo CPU tests,
 *     several ranks sharing one GPU).
 *
 * Phi layout in HBM ("row tiles"): rows are grouped in tiles of 128; inside a
 * tile the element (row r, sample s) lives at  tile*S*128 + s*128 + (r%128),
 * so that a 64-lane wavefront reading one sample of one tile issues a single
 * fully-coalesced 1 KiB load (16 B per lane = two adjacent rows).  Padding rows
 * of the last tile are zero and have norm 0.
 */
#ifndef BETA_CORES_H
#define BETA_CORES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BC_OK 0
#define BC_NUMERICAL_PRECISION 1
#define BC_INVALID_ARGUMENT 2
#define BC_RETRY_EXACT 3 /* bc_snnls_select_pick only: a rank's pre-filter overflowed, redo the sweep exactly
                            (bc_snnls_select_local_exact, all-gather, select_pick); never leaves bc_snnls_select */

#define BC_TILE_ROWS 128

/* solver kinds (bayesiancoresets/snnls/__init__.py:1-4) */
#define BC_ALG_GIGA 0 /* giga.py:6-64 */
#define BC_ALG_FW 1   /* frankwolfe.py:5-40 */
#define BC_ALG_OMP 2  /* orthopursuit.py:7-42 (select on device, NNLS refit on host) */

/* likelihood models for bc_project (K1).  params layout per model below. */
#define BC_MODEL_LINREG_LL 0     /* model_linreg.py:4-10 == model_neurlinr.py:90-97 ; params = {sigsq}          ; Z = [x(D), y] */
#define BC_MODEL_LINREG_BETA 1   /* model_neurlinr.py:102-110                       ; params = {sigsq, beta}    ; Z = [x(D), y] */
#define BC_MODEL_LOGISTIC_LL 2   /* model_lr.py:72-79                               ; params = {}               ; Z = y*x (D)   */
#define BC_MODEL_LOGISTIC_BETA 3 /* model_lr.py:81-86                               ; params = {beta[, c0]}     ; Z = y*x (D)   */
/*   c0 (optional): the formula's value at m = 0 carrying the caller's np.power bits; K1 uses it for the constant projection row of
 *   a data row z = 0 (whether that row centres to exactly 0 hangs on its last bit).  Omitted: the library evaluates it with its
 *   restatement of np.power at base 2 (csrc/bc_np_pow2.h: NumPy's bits on AVX-512 hosts). */
#define BC_MODEL_GAUSS_LL 4      /* gaussian.py:7-15  ; params = {logdetSig, Siginv[d*d]}        ; Z = x (d) */
#define BC_MODEL_GAUSS_BETA 5    /* gaussian.py:34-44 ; params = {beta, logdetSig, Siginv[d*d]}  ; Z = x (d) */
#define BC_MODEL_GAUSS_BETA_GRAD 6 /* gaussian.py:46-62 ; params as GAUSS_BETA (d/dbeta, projector.py:56-61) */

typedef struct bc_ctx bc_ctx;
typedef struct bc_data bc_data;
typedef struct bc_phi bc_phi;
typedef struct bc_snnls bc_snnls;

/* ---- library / context ------------------------------------------------ */
int bc_version(void);
const char* bc_last_error(void);
/* device: HIP ordinal.  stream: a hipStream_t to launch on (e.g. torch's current
 * stream so RCCL collectives issued by the host order with the kernels), or NULL
 * to let the library create its own. */
int bc_ctx_create(int device, void* stream, bc_ctx** out);
int bc_ctx_destroy(bc_ctx* ctx);
int bc_ctx_sync(bc_ctx* ctx);
/* time (ms) spent inside the dominant kernels since the last reset, measured
 * with HIP events on the launch stream; which: 0 = K3 score/argmax sweep,
 * 1 = K1 projection, 2 = K4 XtWX (Gram kernel + its split-order reduction), and the other stages of a greedy step:
 * 3 = rescoring / local winner, 4 = candidate all-gather (RCCL), 5 = step finish.  launches returns the number of
 * TIMED launches.
 * bc_ctx_enable_timing(ctx, n): 0 = off (default), n >= 1 = time every n-th launch of each class
 * (an event pair costs ~11 us of stream time on MI355X, which matters next to a 50 us sweep). */
int bc_ctx_kernel_time(bc_ctx* ctx, int which, double* total_ms, int64_t* launches);
int bc_ctx_kernel_time_reset(bc_ctx* ctx);
int bc_ctx_enable_timing(bc_ctx* ctx, int on);
/* which classes bc_ctx_enable_timing applies to (bit i = class i of bc_ctx_kernel_time); default 0x7: the three dominant
 * kernels.  The stage timers 3-5 cost an event pair each per timed step: switch them on for a diagnostic pass only. */
int bc_ctx_timing_classes(bc_ctx* ctx, uint32_t mask);
/* accumulated GPU time (HIP events, ms) of the BC_VI_PHASES = 5 phases of the bc_vi_gradient calls made while timing
 * was on: upload (Theta, coreset rows, w) | K1 of the coreset rows | store-free K1 over the data rows | column-sum
 * reduction (+ rank-order sum over ranks) | M x S algebra + download; *calls = number of timed calls; reset != 0 clears. */
int bc_ctx_phase_times(bc_ctx* ctx, double* out_ms, int32_t n, int64_t* calls, int reset);

/* ---- native candidate exchange (RCCL on the context's stream) ---------- */
/* the one data-path collective of the sharded greedy loop is an all-gather of an (S+4)-double
 * record per rank and step.  With a bc_comm bound to a solver (bc_snnls_bind_comm) the library issues
 * ncclAllGather itself, so bc_snnls_build / bc_snnls_select run the multi-rank loop without returning to
 * the host language between steps.  Bootstrap: rank 0 calls bc_comm_unique_id and ships the 128 bytes to
 * all ranks (any channel), every rank calls bc_comm_create (collective).  RCCL is resolved with dlopen:
 * bc_comm_load(path) picks a specific library (e.g. the one PyTorch bundles), otherwise "librccl.so". */
typedef struct bc_comm bc_comm;
int bc_comm_load(const char* rccl_library_path);
int bc_comm_unique_id(void* id_out, int32_t capacity /* >= 128 */);
int bc_comm_create(bc_ctx* ctx, const void* id, int32_t rank, int32_t world, bc_comm** out);
int bc_comm_destroy(bc_comm* c);
int bc_comm_info(const bc_comm* c, int32_t* rank, int32_t* world);
/* all-gather `count` doubles per rank between device buffers, enqueued on the context's stream */
int bc_comm_all_gather(bc_comm* c, const void* send_dev, void* recv_dev, int64_t count);
/* collective wiring check: rank-coded pattern gathered and verified element by element */
int bc_comm_selftest(bc_comm* c);
/* everything bc_comm_create needs locally (RCCL found, device usable): call on every rank and agree on the
 * result BEFORE the collective bootstrap, so that a rank which cannot join never leaves the others inside it */
int bc_comm_precheck(bc_ctx* ctx);
/* tear down without waiting for outstanding collectives (ncclCommAbort): a rank that fails mid-loop exits
 * instead of leaving its peers blocked */
int bc_comm_abort(bc_comm* c);
/* sum of `count` doubles over ranks, added IN RANK ORDER on the device (bit-stable), result on the host:
 * replaces vecs.sum(axis=0) of a row-sharded projection (hilbert.py:17, bcores.py:77) */
int bc_comm_sum_doubles(bc_comm* c, const double* in_dev, int64_t count, double* out_host);
/* test hook: the device kernel behind bc_comm_sum_doubles on a fabricated gathered buffer (host, [world][count]), so
 * that its indexing and order of additions can be checked for any world size on one GPU, without a communicator */
int bc_comm_rank_order_sum_selftest(bc_ctx* ctx, const double* gathered_host, int32_t world, int64_t count, double* out_host);

/* ---- data rows (Z) resident on the device ----------------------------- */
/* replaces the `data`/`pts` ndarray argument of Projector.project (projector.py:23,51) */
int bc_data_from_host(bc_ctx* ctx, const double* z_rowmajor, int64_t n_rows, int32_t dz, bc_data** out);
/* borrow an existing device buffer (row-major n_rows x dz doubles); not freed by destroy */
int bc_data_from_device(bc_ctx* ctx, const void* z_dev, int64_t n_rows, int32_t dz, bc_data** out);
/* re-usable slot for SMALL inputs that change every call (the <= M coreset points and the sub-sampled
 * rows that BetaCoreset / SparseVI project thousands of times, bcores.py:52-54,63-64): create once with a
 * row capacity, then upload in place (the buffer grows if n_rows exceeds the capacity). */
int bc_data_create(bc_ctx* ctx, int64_t cap_rows, int32_t dz, bc_data** out);
int bc_data_upload(bc_data* d, const double* z_rowmajor, int64_t n_rows);
/* rows by LOCAL index -> m x dz row-major on the host (`pts = data[idcs]`, hilbert.py:33) */
int bc_data_gather_rows(bc_data* d, const int64_t* local_idx, int64_t m, double* out);
/* Constant rows with the CALLER's bits.  A data row whose d features are all exactly 0 projects to S equal values (a
 * "constant row"); whether the reference's centring (projector.py:26 / :55) leaves it exactly 0 depends on the last bit of
 * that value, and for the beta-likelihood of the linear regression the value contains np.exp (model_neurlinr.py:107).  The
 * library restates the routine NumPy takes on AVX-512 hosts (csrc/bc_np_exp.h); on any other host the reference's bits are
 * that host's NumPy's, so the host layer (coreset/projector.py) evaluates the expression itself:
 *   bc_data_zero_feature_keys: the y (column d) of every row whose first d columns are all zero -> out_keys[0 .. min(*out_n, cap));
 *     *out_n = how many there are (a value above cap says the list is truncated).
 *   bc_ctx_set_constant_row_values: (y, value) pairs, y strictly increasing, for `model` (only BC_MODEL_LINREG_BETA) with
 *     exactly these `params`; later bc_project* calls of that model and those parameters on this context take a constant
 *     row's value from the table when its y is in it (and the value agrees with the device's own to 1e-13).  n = 0 clears. */
int bc_data_zero_feature_keys(const bc_data* data, int32_t d, int64_t cap, double* out_keys, int64_t* out_n);
int bc_ctx_set_constant_row_values(bc_ctx* ctx, int model, const double* params, int32_t n_params, const double* keys,
                                   const double* values, int64_t n);
int bc_data_destroy(bc_data* d);

/* ---- Phi: the N x S matrix of row-centred (beta-)log-likelihoods ------- */
/* upload a C-contiguous n_rows x S host array (what `vecs` is at hilbert.py:11-17;
 * the solver's A = vecs.T is a view of it).  row_offset = global index of row 0. */
int bc_phi_from_host(bc_ctx* ctx, const double* phi_rowmajor, int64_t n_rows, int32_t s,
                     int64_t row_offset, bc_phi** out);
/* an empty Phi with room for cap_rows rows: bc_project re-uses it (*inout) for any n_rows <= cap_rows */
int bc_phi_create(bc_ctx* ctx, int64_t cap_rows, int32_t s, bc_phi** out);
/* K1: Phi = f(Z, Theta[, beta]) - rowmean  (projector.py:24-26, :53-55 + the model
 * formula named by `model`).  theta: host, S x D row-major.  If *inout is non-NULL
 * and has the same S and enough row capacity its buffers are reused (BetaCoreset
 * re-projects every gradient call, bcores.py:141-146). Fuses row norms and column sums (K2). */
int bc_project(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
               const double* params, int32_t n_params, int64_t row_offset, bc_phi** inout);
/* K1 straight from a HOST array, upload and projection pipelined: what `HilbertCoreset(data_ndarray, projector)` does at
 * hilbert.py:11 (`ll_projector.project(data)`, projector.py:23-26) and the greedy-VI classes at bcores.py:44.  The rows go
 * to HBM in ~128 MiB chunks, each one `hipMemcpyAsync` straight from the caller's (pageable) array on a copy stream of the
 * library's own, and K1 runs on chunk c -- behind the event of its arrival -- while chunk c+1.. are on the wire.  (Opt-in,
 * BC_UPLOAD_THREADS = T >= 1: T host threads copy the chunks through pinned 8 MiB staging buffers, each on its own copy
 * stream -- for hosts whose pageable copies are slow; bc_data_from_host / bc_data_upload then use the same uploader, by
 * default they issue one plain hipMemcpyAsync on the context's stream and wait for it.)  K1 launches made here count in
 * timer class 1 like bc_project's (one span per chunk).  Phi, norms AND column sums are bit-identical to
 * bc_data_from_host + bc_project (the chunks keep the column partials' order of additions).  *out_data: the resident rows
 * (caller destroys); *inout as for bc_project.  z_host is only read during the call. */
int bc_project_from_host(bc_ctx* ctx, const double* z_host, int64_t n_rows, int32_t dz, int model, const double* theta, int32_t s,
                         const double* params, int32_t n_params, int64_t row_offset, bc_data** out_data, bc_phi** inout);
/* K1 WITHOUT materialising Phi: only b = Phi^T 1 of the projection of `data`'s rows, S <= 256 doubles to the host.
 * What every gradient of the greedy-VI weight optimisation needs of the N x S projection (bcores.py:141-146,
 * sparsevi.py:129-134: `vecs.sum(axis=0)` inside grd).  Same contraction, formula, centring, per-tile column partials
 * and reduction order as bc_project + bc_phi_colsum -- the result is bit-identical to theirs -- but neither the tiles
 * nor the row norms are written: 8*N*Dz bytes of HBM traffic instead of 8*N*(Dz + S).  comm != NULL: the sum over all
 * ranks' row shards (rank order, as bc_phi_colsum_all). */
int bc_project_colsum(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                      const double* params, int32_t n_params, bc_comm* comm, double* out_s);
/* One gradient of the greedy-VI weight optimisation in ONE call and ONE host synchronisation (bcores.py:141-146,
 * sparsevi.py:129-134, full-data tangent space):
 *     vecs = project(data) ; corevecs = project(core_rows)
 *     resid = sum_scaling * vecs.sum(axis=0) - w.dot(corevecs) ; grad = -corevecs.dot(resid) / S
 * core_rows: host, m x dz row-major (the coreset's points, m >= 1); w: host, m weights; theta / params as bc_project;
 * out_grad: m doubles; out_resid: S doubles or NULL.  The data rows go through the store-free K1, the m coreset rows
 * through the materialising one, the M x S algebra runs on the device; comm as in bc_project_colsum. */
int bc_vi_gradient(bc_ctx* ctx, const bc_data* data, const double* core_rows, int64_t m, int model,
                   const double* theta, int32_t s, const double* params, int32_t n_params, const double* w,
                   double sum_scaling, bc_comm* comm, double* out_grad, double* out_resid);
/* The same in two halves: _begin enqueues everything (including the download) and returns; _end waits and hands the result
 * out.  Host work placed between the two runs beside the GPU (the samplers draw the next sample matrix's normals there). */
int bc_vi_gradient_begin(bc_ctx* ctx, const bc_data* data, const double* core_rows, int64_t m, int model,
                         const double* theta, int32_t s, const double* params, int32_t n_params, const double* w,
                         double sum_scaling, bc_comm* comm);
int bc_vi_gradient_end(bc_ctx* ctx, double* out_grad, double* out_resid);
/* x-gradients of the log-likelihood at `data`'s rows (the coreset's pseudo-points), centred over the coordinate axis:
 * what BlackBoxProjector.project(pts, grad=True) returns next to the projection (projector.py:27-32) and
 * BatchPSVICoreset moves its points with (bpsvi.py:39-57).  model: BC_MODEL_LINREG_LL (model_linreg.py:12-17,
 * W = D+1), BC_MODEL_LOGISTIC_LL (model_lr.py:107-114, W = D) or BC_MODEL_GAUSS_LL (gaussian.py:17-20, W = d), params as
 * for bc_project.  out: host, M x S x W row-major (M = rows of `data`, W = its row width). */
int bc_project_grad_x(bc_ctx* ctx, const bc_data* data, int model, const double* theta, int32_t s,
                      const double* params, int32_t n_params, double* out);
int bc_phi_shape(const bc_phi* phi, int64_t* n_rows, int32_t* s, int64_t* row_offset);
/* b = Phi^T 1 over the local rows (hilbert.py:17 `vecs.sum(axis=0)`, bcores.py:77) */
int bc_phi_colsum(bc_phi* phi, double* out_s);
/* the same summed over all ranks' row shards through a communicator (bc_comm_sum_doubles; bc_comm is declared above) */
int bc_phi_colsum_all(bc_phi* phi, bc_comm* c, double* out_s);
/* row norms sqrt(sum_s Phi[i,s]^2) (giga.py:10, hilbert.py:16, bcores.py:78) */
int bc_phi_norms(bc_phi* phi, double* out_n);
/* number of all-zero rows (dropped at hilbert.py:16 / bcores.py:67) and sum of norms (frankwolfe.py:21,24) */
int bc_phi_norm_stats(bc_phi* phi, int64_t* zero_rows, double* norm_sum);
int bc_phi_to_host(bc_phi* phi, double* out_rowmajor);
/* Grouped (batch) selection, bcores.py:46-50,56-61 / sparsevi.py:44-48,54-59: the reference projects each
 * group's rows and sums them, `vecs = [proj(data[groups[i]]).sum(axis=0) for i in ...]`.  Here the data are
 * projected once (K1) and the groups are summed on the device: *out gets n_groups rows,
 * row g = sum over j in [offsets[g], offsets[g+1]) of Phi[members[j], :], accumulated in member order (NumPy's
 * order for an axis-0 sum).  members are LOCAL row numbers of `p`; a group may be empty (zero row). */
int bc_phi_group_sum(bc_phi* p, const int64_t* members, const int64_t* offsets, int64_t n_groups, bc_phi** out);

/* rows by LOCAL index -> m x S row-major (A[:, f] at giga.py:45, frankwolfe.py:27) */
int bc_phi_gather_rows(bc_phi* phi, const int64_t* local_idx, int64_t m, double* out);
/* out[j] = sum_i weights[i] * Phi[i, :]  is not needed by the reference; what it
 * needs is Phi . v for ONE S-vector: scores[i] = Phi[i,:].v  (bcores.py:78 before
 * the row-norm division).  Mostly a debugging / test aid. */
int bc_phi_matvec(bc_phi* phi, const double* v_s, double* out_n);
int bc_phi_destroy(bc_phi* phi);

/* ---- K3-only entry: one fused score + argmax sweep --------------------- */
/* mode 0 (GIGA, giga.py:31-38): v = [cdir, xw] interleaved (2*S doubles).
 * mode 1 (dot, frankwolfe.py:16-17, orthopursuit.py:18-19, bcores.py:78-81):
 *        v = residual (S doubles); score = Phi[i,:].v / norm[i] / post_div.
 * Zero-norm rows are skipped.  Returns the best local row as a GLOBAL index
 * (-1 if there is no valid row) and its score. */
int bc_phi_argmax(bc_phi* phi, int mode, const double* v, double post_div, int64_t* best, double* score);

/* ---- sparse-NNLS solver state (bayesiancoresets/snnls/snnls.py:8-106) -- */
/* b: host, S doubles, the GLOBAL right-hand side (replicated on every rank).
 * norm_sum: GLOBAL sum of row norms (FrankWolfe, frankwolfe.py:21,24); pass the
 * value from bc_phi_norm_stats when there is one rank.
 * Raises BC_INVALID_ARGUMENT if a local row has zero norm (giga.py:11-12) unless
 * allow_zero_rows != 0 (HilbertCoreset keeps dropped rows in place, masked). */
int bc_snnls_create(bc_ctx* ctx, bc_phi* phi, const double* b, int alg, double norm_sum,
                    int allow_zero_rows, bc_snnls** out);
int bc_snnls_destroy(bc_snnls* h);
/* *on = 0, or the storage precision (8 / 16 / 32 bits per element) of the mirror of Phi this solver's sweeps
 * stream through the reduced-precision pre-filter (bc_prefilter.hip): int8 for shards of >= 163840 rows by
 * default -- from 2 000 000 rows behind a 4-bit first level (form 3 below; *on still says 8: the int8 digits are what bounds the
 * rows it passes on) --; BC_PREFILTER=0 / 4 / 8 / 16 / 32 in the environment forces it.  Selections and weights are identical either way
 * (candidates are rescored from the fp64 Phi with the arithmetic of the fp64 sweep). */
int bc_snnls_prefilter_active(const bc_snnls* h, int* on);
/* *form = 0: fp64 sweeps; 1: two-pass pre-filter (reduced-precision sweep, then one block rescoring the candidates from the
 * fp64 Phi); 2: branch-and-bound int8 sweep (csrc/bc_prefilter_bb.h: a fifth wave per sweep block rescoring the candidates
 * beside the stream; opt-in with BC_I8_BB=1 -- measured slower than form 1 at every size tried, kept for the record and its
 * tests); 3: two-level pre-filter (csrc/bc_prefilter_i4.h: a 4-bit mirror streamed first -- half a byte per element --, the rows
 * it cannot exclude re-bounded from row-major int8 records, then the fp64 rescoring; BC_PREFILTER=4, S <= 256).  Same call sites as bc_snnls_prefilter_active (giga.py:31-38, frankwolfe.py:16-17): the row returned is the
 * fp64 sweep's in every form. */
int bc_snnls_prefilter_form(const bc_snnls* h, int* form);
/* Diagnostic: how many sweeps since creation overflowed the pre-filter's candidate lists (thousands of exactly
 * duplicated rows, say).  Such a step consumes nothing: it is marked on the device, the rest of the enqueued
 * launches become no-ops, and the host re-runs it with the exact fp64 sweep (stream-ordered, no in-launch hand-shake). */
int bc_snnls_prefilter_fallbacks(const bc_snnls* h, int64_t* n);
/* Diagnostic: sweeps run through the pre-filter since creation, rows it handed to the exact fp64 rescoring in
 * total (candidates / sweeps = how selective the reduced-precision bounds are on this data), and fallbacks. */
int bc_snnls_prefilter_stats(const bc_snnls* h, int64_t* sweeps, int64_t* candidates, int64_t* fallbacks);
/* Diagnostic of the two-level form (prefilter_form 3, csrc/bc_prefilter_i4.h: a 4-bit first level in front of the int8
 * records): first-level sweeps run, rows they listed in total, and rows the second level re-bounded from the int8 records
 * (listed / (sweeps * rows) = what the seeds leave of the stream; refined / sweeps = the gather's size).  All zero for the
 * other forms.  Same call sites as bc_snnls_prefilter_stats. */
int bc_snnls_prefilter_levels(const bc_snnls* h, int64_t* l1_sweeps, int64_t* listed, int64_t* refined);
/* bayesiancoresets/util/__init__.py:4-7 (TOL, set_tolerance); default 1e-12 */
int bc_snnls_set_tolerance(bc_snnls* h, double tol);
/* multi-rank: device buffers (world*(S+4) and (S+4) doubles) through which the
 * host all-gathers the per-rank candidate records between step_local and
 * step_finish.  With world == 1 nothing needs binding. */
int bc_snnls_bind_exchange(bc_snnls* h, int world, void* cand_send_dev, void* cand_all_dev);
/* native exchange: the solver all-gathers its candidate records through `c` by itself (owns the buffers);
 * afterwards bc_snnls_build and bc_snnls_select work for world > 1.  Exclusive with bc_snnls_bind_exchange. */
int bc_snnls_bind_comm(bc_snnls* h, bc_comm* c);
int bc_snnls_record_doubles(const bc_snnls* h, int32_t* n);

/* fused greedy loop, all on device, no host round trip per iteration
 * (snnls.py:31-79 incl. the monotone guard, revert, retry-once-then-stop).   */
int bc_snnls_build_begin(bc_snnls* h, int itrs);      /* resets the per-call retry flag (snnls.py:40) */
int bc_snnls_step_local(bc_snnls* h);                 /* K3 sweep + local winner -> cand_send */
int bc_snnls_step_local_exact(bc_snnls* h);           /* the same through the fp64 sweep: redo of a step whose pre-filter overflowed */
int bc_snnls_step_finish(bc_snnls* h);                /* winner over cand_all, reweight, guard, prep next */
/* pending_exact (may be NULL): 1 when the enqueued steps stopped at a pre-filter overflow -- the caller runs ONE
 * step as step_local_exact / all-gather / step_finish and then continues with the remaining iterations
 * (iterations_consumed tells how many are done); bc_snnls_build does all of that by itself. */
int bc_snnls_build_end(bc_snnls* h, int* reached_numeric_limit, int* iterations_consumed, int* pending_exact);
/* = begin; itrs x (step_local; step_finish); end   -- single-rank convenience */
int bc_snnls_build(bc_snnls* h, int itrs, int* reached_numeric_limit);

/* step-wise protocol for SparseNNLS subclasses (snnls.py:102-106).
 * select: BC_NUMERICAL_PRECISION when the reference's _select would raise
 * (giga.py:28-29).  For world > 1 call select_local, all-gather, select_pick. */
int bc_snnls_select(bc_snnls* h, int64_t* f);
int bc_snnls_select_local(bc_snnls* h);
int bc_snnls_select_local_exact(bc_snnls* h);         /* after select_pick returned BC_RETRY_EXACT */
int bc_snnls_select_pick(bc_snnls* h, int64_t* f);
/* reweight with column f.  The column is taken from the last select's candidate
 * records when f is among them, else gathered from the local shard
 * (BC_INVALID_ARGUMENT if f is not local and world > 1). */
int bc_snnls_reweight(bc_snnls* h, int64_t f);
int bc_snnls_error(bc_snnls* h, double* err);                       /* snnls.py:28-29 */
int bc_snnls_size(bc_snnls* h, int64_t* nnz_positive);              /* snnls.py:21-22 */
/* sparse view of w: entries in selection order (values may be 0). cap = capacity of the arrays. */
int bc_snnls_weights(bc_snnls* h, int64_t cap, int64_t* idx, double* val, int64_t* n);
/* replace w (snnls.py:59 revert, :88 optimize, orthopursuit.py:41).  cols: n x S
 * row-major columns A[:, idx[j]] or NULL to gather them from the local shard. */
int bc_snnls_set_weights(bc_snnls* h, int64_t n, const int64_t* idx, const double* val, const double* cols);
/* columns of the current active list, n x S (for the host NNLS refit, snnls.py:87) */
int bc_snnls_columns(bc_snnls* h, int64_t cap, double* cols, int64_t* n);
int bc_snnls_reset(bc_snnls* h);                                    /* snnls.py:18-20 */
int bc_snnls_get_flags(bc_snnls* h, int* reached_numeric_limit);
int bc_snnls_set_flags(bc_snnls* h, int reached_numeric_limit);
/* per-iteration trace of the fused loop since the last reset: f (or -1), status, error */
int bc_snnls_trace(bc_snnls* h, int64_t cap, int64_t* f, int32_t* status, double* err, int64_t* n);

/* ---- K4: X^T diag(w) X and X^T (w*y) (model_linreg.py:29,31) ----------- */
/* data rows are [x(D), y]; w: host, n_rows doubles (NULL = all ones).
 * out_xtwx: D x D row-major, out_xtwy: D.  Local rows only; the host sums over ranks. */
int bc_weighted_gram(bc_ctx* ctx, const bc_data* data, const double* w, double* out_xtwx, double* out_xtwy);
/* the same for coreset-sized rows in HOST memory (what the samplers pass, bcores.py:39 -> sampler -> weighted_post on the
 * <= M coreset points, once per gradient): one transfer in, one out, one synchronisation.  z: n_rows x dz row-major. */
int bc_weighted_gram_host(bc_ctx* ctx, const double* z_rowmajor, int64_t n_rows, int32_t dz, const double* w,
                          double* out_xtwx, double* out_xtwy);

#ifdef __cplusplus
}
#endif
#endif /* BETA_CORES_H */
