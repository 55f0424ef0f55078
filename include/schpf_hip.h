/*
 * schpf_hip.h -- C ABI of libschpf_hip.so, the MI355X (gfx950) engine for the scHPF CAVI
 * hot path.
 *
 * The reference (simslab/scHPF 0.5.0) has no FFI: its seam for this path is the set of
 * numba-compiled Python callables in schpf/hpf_numba.py, imported by schpf/scHPF_.py:21
 * and schpf/loss.py:13 and called only from scHPF._fit (scHPF_.py:642-715) and
 * loss.pois_llh_pointwise (loss.py:136-138).  Every entry point below names the
 * reference interface it replaces.  The binding a maintainer would add on the
 * reference side is a ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; all array arguments of the stateless functions and of
 *     set/get_state are HOST pointers to C-contiguous row-major buffers owned by the
 *     caller for the duration of the call (nothing is retained);
 *   - `dtype` selects the model precision T: SCHPF_F32 or SCHPF_F64 (the reference's
 *     scHPF(dtype=...) / hpf_numba.py:30,80);  indices are int32 (SciPy COO default);
 *   - every function returns 0 on success, non-zero on failure (SCHPF_ERR_NO_MEMORY when the
 *     device or the host ran out of memory, 1 for everything else); schpf_last_error()
 *     returns a message for the calling thread.  The library never aborts the process;
 *   - one context = one GPU = one host thread at a time.  Work is enqueued on the
 *     context's HIP stream; calls that return data to the host synchronise it.
 */
#ifndef SCHPF_HIP_H
#define SCHPF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden and an export list (csrc/libschpf_hip.map): the functions declared
 * between this pragma and its pop are everything the shared object exports */
#pragma GCC visibility push(default)

#define SCHPF_F32 0
#define SCHPF_F64 1

/* which variational distribution (scHPF_.py:264-267) */
#define SCHPF_XI 0
#define SCHPF_THETA 1
#define SCHPF_ETA 2
#define SCHPF_BETA 3

/* status of a call that failed because hipMalloc (or a host allocation of the plan builders) found no memory */
#define SCHPF_ERR_NO_MEMORY 2

/* element type of the COO values handed to schpf_upload_coo */
#define SCHPF_VAL_I32 0
#define SCHPF_VAL_I64 1
#define SCHPF_VAL_F32 2
#define SCHPF_VAL_F64 3

/* step flags (keyword arguments of scHPF._fit, scHPF_.py:526-530) */
#define SCHPF_FREEZE_GENES 1u   /* freeze_genes=True: skip the eta/beta block (project()) */
#define SCHPF_SIMULTANEOUS 2u   /* beta_theta_simultaneous=True (scHPF_.py:666-685)      */
#define SCHPF_CELLS_FIRST 8u    /* minibatch order (scHPF_.py:688-704): xi/theta block first (theta.rate
                                   from the current beta), then the gene block from the NEW theta    */
#define SCHPF_LOCAL_GENE 16u    /* schpf_step_local: only the gene-side sweep (+ packing)           */
#define SCHPF_LOCAL_CELL 32u    /* schpf_step_local: only the cell-side sweep; neither bit = both.
                                   Lets the host start the all-reduce of the gene-side sums and
                                   overlap it with the cell-side sweep.                             */
#define SCHPF_SHARDED 4u        /* cells are sharded over several GPUs: gene-side sums go
                                   through the exchange buffer (all-reduced by the host) */

typedef struct schpf_ctx schpf_ctx;

const char *schpf_last_error(void);
int schpf_device_count(int *count);
const char *schpf_version(void);

/* ---------------------------------------------------------------------------------
 * Stateless operator mirrors: array in, array out, caller's COO order.
 * ------------------------------------------------------------------------------- */

/* hpf_numba.psi / hpf_numba.cgammaln (hpf_numba.py:16-22): double -> double. */
int schpf_digamma(int64_t n, const double *x, double *out);
int schpf_gammaln(int64_t n, const double *x, double *out);

/* compute_Xphi_data(X_data, X_row, X_col, theta_vi_shape, theta_vi_rate, beta_vi_shape,
 * beta_vi_rate) -> Xphi (nnz, K)                              hpf_numba.py:54-114.
 * x: (nnz,) of T;  row/col: (nnz,) int32;  theta_*: (ncells, K);  beta_*: (ngenes, K). */
int schpf_xphi(int dtype, int64_t nnz, int ncells, int ngenes, int nfactors, const void *x,
               const int32_t *row, const int32_t *col, const void *theta_shape,
               const void *theta_rate, const void *beta_shape, const void *beta_rate, void *out);

/* compute_pois_llh(...) -> llh (nnz,)                         hpf_numba.py:24-51. */
int schpf_pois_llh_pointwise(int dtype, int64_t nnz, int ncells, int ngenes, int nfactors,
                             const void *x, const int32_t *row, const int32_t *col,
                             const void *theta_shape, const void *theta_rate,
                             const void *beta_shape, const void *beta_rate, void *out);

/* compute_loading_shape_update(Xphi_data, X_keep, nkeep, shape_prior) -> (nkeep, K)
 *                                                             hpf_numba.py:128-156. */
int schpf_shape_update(int dtype, int64_t nnz, int nfactors, const void *xphi,
                       const int32_t *keep, int nkeep, double shape_prior, void *out);

/* compute_loading_rate_update(prior_vi_shape, prior_vi_rate, other_loading_vi_shape,
 * other_loading_vi_rate) -> (n, K)                            hpf_numba.py:159-177.
 * prior_*: (n,);  other_*: (m, K). */
int schpf_rate_update(int dtype, int n, int m, int nfactors, const void *prior_shape,
                      const void *prior_rate, const void *other_shape, const void *other_rate,
                      void *out);

/* compute_capacity_rate_update(loading_vi_shape, loading_vi_rate, prior_rate) -> (n,)
 *                                                             hpf_numba.py:180-188. */
int schpf_capacity_rate_update(int dtype, int n, int nfactors, const void *shape,
                               const void *rate, double prior_rate, void *out);

/* ---------------------------------------------------------------------------------
 * The engine: device-resident state for scHPF._fit's loop (scHPF_.py:642-715).
 * ------------------------------------------------------------------------------- */

/* Create a context on HIP device `device`.  `stream`: a hipStream_t to enqueue on; NULL lets the
 * library create a (non-blocking) stream of its own; SCHPF_STREAM_DEFAULT selects the device's
 * null stream -- torch.cuda.current_stream().cuda_stream is 0 for it, which a caller must map to
 * SCHPF_STREAM_DEFAULT when it wants its collectives ordered with the engine's kernels.
 * ncells is the number of LOCAL cells when cells are sharded. */
#define SCHPF_STREAM_DEFAULT ((void *)1)
int schpf_create(schpf_ctx **out, int device, void *stream, int dtype, int ncells, int ngenes,
                 int nfactors);
int schpf_destroy(schpf_ctx *ctx);

/* The count matrix X (scipy coo_matrix: X.row, X.col, X.data), any order, duplicates kept
 * as separate observations like the reference (hpf_numba.py:98-112).  Validates on the host,
 * copies the triples to the device and builds both sweep plans there (the host
 * builder, SCHPF_DEVICE_PLAN=0, gives the same plans bit for bit).  Values must be finite and
 * >= 0 -- the reference takes any X.data (hpf_numba.py:98-112).  They are stored as float32: UMI
 * counts are exact; other values are rounded (relative 6e-8) and counted in schpf_upload_info.
 * Explicitly stored zeros add nothing to the updates and -r each to the loss, as in the reference
 * (hpf_numba.py:43-50).  Host pointers are not retained. */
int schpf_upload_coo(schpf_ctx *ctx, int64_t nnz, const int32_t *row, const int32_t *col,
                     const void *val, int val_kind);

/* a, c (shape priors of theta, beta) and bp, dp (rate hyper-priors; scHPF_.py:847-879).
 * ap/cp only enter through the constant xi/eta shapes (scHPF_.py:616-618), which the
 * caller sets with schpf_set_state. */
int schpf_set_hypers(schpf_ctx *ctx, double a, double c, double bp, double dp);

/* vi_shape / vi_rate of one HPF_Gamma (scHPF_.py:27-81); (n,) for xi/eta, (n, K) else. */
int schpf_set_state(schpf_ctx *ctx, int which, const void *shape, const void *rate);
int schpf_get_state(schpf_ctx *ctx, int which, void *shape, void *rate);

/* t == 0 of a fit with reinit=True (scHPF_.py:652-655): X*phi with phi ~ Dirichlet(1_K).
 * _host: the caller drew it (NumPy, seed-compatible with the reference) and passes
 *        Xphi_data, (nnz, K) float64, in the order of the uploaded COO.
 * _device: counter-based generator on the GPU (not NumPy-compatible; for matrices whose
 *        nnz*K host draw is impractical).
 * The next schpf_step / schpf_step_local consumes it instead of computing responsibilities. */
int schpf_init_phi_host(schpf_ctx *ctx, const double *xphi);
int schpf_init_phi_device(schpf_ctx *ctx, uint64_t seed);

/* One CAVI iteration, the body of the loop at scHPF_.py:657-714 (non-batched order):
 * responsibilities -> [beta.shape, beta.rate, eta.rate] -> [theta.shape, theta.rate,
 * xi.rate].  schpf_step = schpf_step_local + schpf_step_finish on one GPU. */
int schpf_step(schpf_ctx *ctx, unsigned flags);

/* n iterations in one call -- the iterations between two loss checks of scHPF._fit
 * (scHPF_.py:642-718 with check_freq).  Same result as n schpf_step calls; from the second call
 * with the same (flags, n) the iterations are replayed as one hipGraph (SCHPF_GRAPH=0 disables). */
int schpf_steps(schpf_ctx *ctx, unsigned flags, int n);

/* Sharded form: _local runs both sweeps and packs the gene-side sums [G*K] followed by
 * the local sum_i E[theta_ik] [K] into the exchange buffer (device memory, dtype T);
 * the host all-reduces (sum) that buffer over the ranks (RCCL), then calls _finish. */
int schpf_step_local(schpf_ctx *ctx, unsigned flags);
int schpf_exchange_buffer(schpf_ctx *ctx, void **device_ptr, int64_t *count);
int schpf_step_finish(schpf_ctx *ctx, unsigned flags);

/* Loss terms of mean_negative_pois_llh (loss.py:142-168) for the CURRENT state over the
 * local nonzeros:  llh_sum = sum x*log(r) - r,  gammaln_sum = sum lgamma(x+1).
 * mean negative llh = -(llh_sum - gammaln_sum) / nnz  (sum the three over ranks first). */
int schpf_loss_terms(schpf_ctx *ctx, double *llh_sum, double *gammaln_sum, int64_t *nnz);

int schpf_synchronize(schpf_ctx *ctx);

/* Cells sharded over the GPUs of a node, the collective inside the library (RCCL over xGMI, bound
 * at run time to the copy of librccl.so already in the process -- PyTorch's when torch is loaded
 * -- else $SCHPF_RCCL_PATH, else /opt/rocm/lib).  One context per GPU (per process, or per host
 * thread of one process).  schpf_comm_unique_id: 128 bytes from ONE rank, handed to all others by
 * the host (a file, MPI, torch.distributed, a Python list between threads); schpf_comm_init: every
 * rank, collectively (it blocks until all `world` ranks have called it).  schpf_steps_sharded =
 * n x [schpf_step_local(gene side) -> all-reduce of the exchange buffer on the communicator's own
 * stream, under the cell-side sweep -> schpf_step_finish]; freeze_genes needs no exchange.
 * schpf_loss_terms_all = schpf_loss_terms summed over the ranks. */
/* Before schpf_upload_coo, optional: tell the context that its iterations will be sharded ones (the two
 * sweeps then run as two launches, and the plans of a small row block are cut into tasks that fill the
 * GPU once per launch instead of once per pair of launches). */
int schpf_hint_sharded(schpf_ctx *ctx, int on);
/* Before schpf_upload_coo, optional: this context's matrix is replaced every iteration or so (the host-slicing
 * fall-back of minibatch CAVI, scHPF_.py:643-650): plans are then built the cheapest way -- windows cut by index, no
 * balancing pass (plan.h BALANCED WINDOWS pays for itself over tens of iterations on one matrix, not over one). */
int schpf_hint_transient(schpf_ctx *ctx, int on);

/* Minibatch CAVI without re-uploads (the reference re-slices X each iteration: X[batch_ix], scHPF_.py:643-650,
 * with util.minibatch_ix_generator :218-231).  schpf_keep_rows(ctx, 1) BEFORE schpf_upload_coo makes the engine
 * keep, beside its plans, a (row, col)-sorted copy of the matrix in HBM.  schpf_upload_rows(batch, source, rows, n)
 * then makes `batch`'s matrix the rows rows[0..n) of `source`'s (local cell i = source cell rows[i]; n must be the
 * ncells `batch` was created with, same device, dtype, ngenes, nfactors): gathered and planned on the device,
 * nothing crosses PCIe but the n row numbers.  A batch engine has no loss constants: schpf_loss_terms on it fails,
 * evaluate the loss on the source. */
int schpf_keep_rows(schpf_ctx *ctx, int on);
int schpf_upload_rows(schpf_ctx *ctx, schpf_ctx *source, const int32_t *rows, int n_rows);
int schpf_comm_unique_id(void *out128);
int schpf_comm_init(schpf_ctx *ctx, const void *unique_id128, int rank, int world);
int schpf_comm_destroy(schpf_ctx *ctx);
int schpf_steps_sharded(schpf_ctx *ctx, unsigned flags, int n);
int schpf_loss_terms_all(schpf_ctx *ctx, double *llh_sum, double *gammaln_sum, int64_t *nnz);

/* The hipStream_t the context enqueues on (0 = the null stream), so that a caller can order its
 * own work -- the all-reduce of the exchange buffer -- with the engine's kernels. */
int schpf_stream_handle(schpf_ctx *ctx, void **stream);

/* HIP-event timing of the sweep kernel launches on the context's stream (bench.py).
 * ms[0] = cell sweep, ms[1] = gene sweep, ms[2] = loss sweep, ms[3] = gamma updates;
 * launches[] likewise.  When both sweeps of an iteration run as ONE launch (the default for
 * schpf_step) that launch is counted under ms[0]/launches[0] and ms[1] stays 0.
 * Reading synchronises the stream and resets the counters. */
int schpf_profile_enable(schpf_ctx *ctx, int enable);
int schpf_profile_read(schpf_ctx *ctx, double ms[4], int64_t launches[4]);

/* Shader clock (MHz) the device sustained UNDER the sweep launches on this context since the last call: workgroup 0 of
 * every tile-plan sweep launch stamps the shader-cycle counter and the constant-rate counter on entry and exit
 * (sweep_impl.h clock_probe_*); launches = how many launches the figure averages (0: none ran, shader_mhz = 0).
 * Synchronises the stream and resets the accumulators.  bench.py's roofline.sclk_mhz. */
int schpf_profile_clock(schpf_ctx *ctx, double *shader_mhz, int64_t *launches);

/* Bytes ONE iteration moves through the LDS and streams from HBM, computed from the tile plans (all zero for the
 * gather plan): info = {LDS reads of the nonzeros alone (2 * nnz table rows of KP values), LDS reads of every stored
 * step slot (padding included), LDS writes of the window stagings cell side, gene side, entry-stream bytes of both
 * plans in HBM, partial-row bytes written, the plan the loss pass sweeps (0 cell-major, 1 gene-major: the model of
 * capi.hip loss_tasks / loss_side), the tasks it runs as}.  bench.py's roofline.lds and loss_pass. */
int schpf_sweep_bytes(schpf_ctx *ctx, int64_t info[8]);

/* Plan facts for reports: info[0..] = KP, KL, LPC, chunk_len (tile plan: minus the rows per LDS
 * window / ring slot), windows_cell, windows_gene, n_chunks_cell, n_chunks_gene, n_waves_cell,
 * n_waves_gene, stored entry slots cell, gene, ring slots cell, gene (1 = window schedule),
 * bytes per ring slot, waves per workgroup */
int schpf_plan_info(schpf_ctx *ctx, int64_t info[16]);

/* Facts about the uploaded matrix: info = {nnz, values that were rounded to float32, explicitly
 * stored zeros, bit 0: the packed 8-byte entry format is in use | bit 1: a row-sorted copy is resident
 * (schpf_keep_rows took effect: device-built tile plans)}. */
int schpf_upload_info(schpf_ctx *ctx, int64_t info[4]);

/* Row sums (per cell) and column sums (per gene) of a host COO matrix: the inputs of the empirical
 * hyperparameters bp = ap * mean/var(cell sums), dp = cp * mean/var(gene sums)
 * (scHPF_.py:847-879, where they are X.sum(1) / X.sum(0)).  Host-side, multi-threaded, no GPU
 * needed; sums are accumulated in double (exact for counts).  row_sums[ncells], col_sums[ngenes]. */
int schpf_coo_marginals(int64_t nnz, const int32_t *row, const int32_t *col, const void *val, int val_kind,
                        int ncells, int ngenes, double *row_sums, double *col_sums);

/* Test hook (host only, no GPU needed): build one sweep plan from (major, minor, val) and expand
 * it back into per-nonzero records in storage order -- the major/minor/val it will be processed
 * with, the partials row (natural chunk id) it accumulates into and the wavefront that streams
 * it -- plus cptr[n_major + 1] and stats = {n_chunks, n_slices, n_waves, stored entry slots}. */
int schpf_debug_plan_expand(int64_t nnz, const int32_t *major, const int32_t *minor, const float *val,
                            int n_major, int n_minor, int lpc, int chunk_len, int n_windows,
                            int32_t *out_major, int32_t *out_minor, float *out_val,
                            int32_t *out_natid, int32_t *out_wave, int32_t *out_cptr,
                            int64_t stats[4]);

/* Same for the tile plan (LDS-staged sweep): per stored nonzero the major/minor/val, the partial
 * row (task * groups_per_block + group) it accumulates into and its task; pfirst/pcount[n_major];
 * stats = {n_tasks, n_blocks, n_windows, pstride, stored entry slots, windows_per_task, LDS passes that read a
 * row, extra LDS cycles of those passes under the bank model of plan.cpp (rows of one class are serialised)};
 * $SCHPF_BANK_ORDER picks the order inside a segment (plan.h TileShape::bank_order).
 * ring <= 1: window schedule with win_rows rows per window; ring <= -2: the half-window schedule with
 * -ring slots of slot_bytes (a multiple of 16; table rows are 160 bytes here) -- the hook then also
 * checks that every entry of an epoch points into a slot readable in that epoch. */
int schpf_debug_tile_expand(int64_t nnz, const int32_t *major, const int32_t *minor, const float *val,
                            int n_major, int n_minor, int lpc, int waves_per_block, int win_rows,
                            int target_tasks, int ring, int slot_bytes, int32_t *out_major,
                            int32_t *out_minor, float *out_val, int32_t *out_prow, int32_t *out_task,
                            int32_t *out_pfirst, int32_t *out_pcount, int64_t stats[8]);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* SCHPF_HIP_H */
