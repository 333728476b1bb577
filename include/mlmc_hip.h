/*
 * mlmc_hip.h -- C ABI of libmlmc_hip.so: MI355X (gfx950) moment estimation and maximum-entropy
 * PDF reconstruction for multilevel Monte Carlo.
 *
 * The reference (GeoMop/MLMC, /root/reference) is pure Python and has no FFI of its own; the
 * boundary it offers for this path is a set of Python call signatures.
 * Each entry point below names the reference interface it replaces (file:line relative to
 * /root/reference).  Host code (the mlmc_amd Python package) binds these with ctypes; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success, non-zero on error (message through
 * mlmc_last_error(), thread-local).  The caller owns every buffer passed in; the library owns
 * its device scratch.  `mem_kind` says where a caller buffer lives (host or the bound device).
 * One process binds one device (one process per GPU) and the library works on ONE stream with shared workspaces.
 * Thread safety: every entry point below (all but mlmc_last_error / mlmc_abi_version) takes one library-wide lock, so
 * calls from several host threads are safe and are serialised in arrival order (bindings such as ctypes release the GIL
 * during a call).  Distinct handles may be used from distinct threads concurrently; a multi-call sequence on ONE
 * accumulator (reset ... push ... finalize) belongs to one thread at a time -- the one-call forms mlmc_accum_estimate /
 * mlmc_accum_estimate_packed hold the lock for the whole estimate.
 * All floating point is IEEE fp64, all counts int64.  No CPU fallback exists: without a HIP
 * device every compute entry point fails.
 */
#ifndef MLMC_HIP_H
#define MLMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLMC_ABI_VERSION 6   /* 2: strides in mlmc_expr_eval, chaining flags, mlmc_accum_estimate_packed; 3: mlmc_wait_event;
                              * 4: x_lo / x_hi in mlmc_basis_desc, mlmc_expr_state, mlmc_accum_kernel_flops;
                              * 5: mlmc_accum_aux_kernel_time; 6: mlmc_linearization_table */

/* basis kinds -- mlmc/moments.py: Legendre :174-229, Monomial :111-130, Fourier :133-171;
 * IDENTITY = the quantity itself (estimate_mean of a plain quantity, quantity_estimate.py:22-80);
 * SPLINE = cubic B-spline moments.  The reference imports scipy's BSpline (moments.py:3) but defines no spline class
 * (SURVEY fact 2): phi_0 = 1, phi_r = B_r (r = 1..size-1) of the clamped uniform cubic B-spline basis B_0..B_{size-1}
 * on ref_domain, pinned against scipy.interpolate.BSpline ("parity unpinned" with respect to the reference). */
enum { MLMC_LEGENDRE = 0, MLMC_MONOMIAL = 1, MLMC_FOURIER = 2, MLMC_IDENTITY = 3, MLMC_SPLINE = 4 };
/* accumulation modes */
enum {
    MLMC_MODE_MOMENTS = 0,  /* qe.moments + estimate_mean    (quantity_estimate.py:96-119, :22-80)  K = R      */
    MLMC_MODE_COV = 1       /* qe.covariance + estimate_mean (quantity_estimate.py:122-156, :22-80) K = R * R  */
};
enum { MLMC_HOST = 0, MLMC_DEVICE = 1 };

typedef struct mlmc_basis mlmc_basis;
typedef struct mlmc_accum mlmc_accum;

/* Plain-data image of a reference Moments object (mlmc/moments.py:10-39):
 * t = (x - shift) * scale + ref0 (after log(x) if is_log); values with t outside [ref0, ref1]
 * are NaN-masked when is_clip (Moments.clip, moments.py:58-67).  `matrix` (row-major
 * [out_size][size], host pointer, may be NULL) is TransformedMoments._transform (moments.py:232-259).
 * x_lo / x_hi (used when is_log and is_clip): the keep / drop decision of a sample under log=True is taken on the RAW
 * value, keep <=> x_lo <= x <= x_hi, where x_lo is the smallest and x_hi the largest double whose
 * t = (log(x) - shift) * scale + ref0, computed with the CALLER's log, lies in [ref0, ref1].  The map x -> t is monotone,
 * so the caller finds both by bisection over the doubles (<= 64 steps each); sample counts are then bit-identical to the
 * caller's own NumPy / libm path, whatever the last bit of the device's log is (moments.py:27-39,58-73 use np.log).
 * x_lo == x_hi == 0 asks the library to bisect with the C library's log() of the host. */
typedef struct {
    int32_t kind;
    int32_t size;       /* number of basis functions R of the underlying family */
    double shift;       /* Moments._linear_shift */
    double scale;       /* Moments._linear_scale */
    double ref0, ref1;  /* Moments.ref_domain */
    int32_t is_log;     /* Moments._is_log */
    int32_t is_clip;    /* Moments._is_clip (safe_eval) */
    int32_t out_size;   /* rows of `matrix`; 0 = no linear transform */
    int32_t reserved;
    const double *matrix;
    double x_lo, x_hi;  /* raw-value keep interval under is_log && is_clip (see above) */
} mlmc_basis_desc;

/* ---- runtime -------------------------------------------------------------------------- */
/* Bind this process to HIP device `device` (>= 0). flags: bit0 = record HIP-event timing of the
 * accumulation kernels (read back with mlmc_accum_kernel_time). */
int mlmc_init(int device, int flags);
void mlmc_shutdown(void);
/* Run all later work on the caller's HIP stream (e.g. the stream RCCL collectives are enqueued on), so that
 * mlmc_accum_finalize_packed(..., MLMC_DEVICE) followed by an all-reduce needs no host synchronisation in between. */
int mlmc_set_stream(void *hip_stream);
/* Make the library's stream wait for a HIP event recorded on another stream of the same device (asynchronous for the
 * host): the feed of a storage that arrives in many chunks (SampleStorageHDF: one `collected_values[chunk_slice]` read per
 * chunk, mlmc/tool/hdf5.py:353-376) copies chunk k + 1 from pinned host memory on a copy stream while the kernels of chunk k
 * run here; every later launch of the library is ordered behind the event. */
int mlmc_wait_event(void *hip_event);
/* Wait until everything the library has enqueued on its stream is done (asynchronous entry points: mlmc_accum_push
 * with device buffers, mlmc_expr_eval, mlmc_accum_finalize_packed(MLMC_DEVICE)). */
int mlmc_synchronize(void);
const char *mlmc_last_error(void);
int mlmc_abi_version(void);
/* name[<=256], CU count, wavefront size, total HBM bytes of the bound device */
int mlmc_device_info(char *name, int name_len, int *n_cu, int *wave_size, int64_t *hbm_bytes);

/* ---- moment functions ----------------------------------------------------------------- */
int mlmc_basis_create(const mlmc_basis_desc *desc, mlmc_basis **out);
void mlmc_basis_destroy(mlmc_basis *b);
/* Moments.eval_all(value, size) (moments.py:90-93; legvander/polyvander/Fourier/TransformedMoments
 * ._eval_all :122-126,:145-162,:195-197,:256-259): out[i * size + r] for i < n, r < size.
 * Masked values give NaN in every column, as in the reference. */
int mlmc_basis_eval(const mlmc_basis *b, const double *x, int64_t n, int32_t size, double *out, int mem_kind);

/* ---- level-difference accumulation (the hot loop) ------------------------------------- */
/* One accumulator = one estimate_mean() call (quantity_estimate.py:22-80) over `n_levels` levels of a
 * quantity with `n_comp` (= M) scalar components; rows K = n_comp * R (MOMENTS) or n_comp * R * R (COV),
 * row index m * R + r / m * R * R + i * R + j (mom_at_bottom / cov_at_bottom = True layout). */
/* mode may carry MLMC_MODE_MEAN_ONLY: the caller will read only the sums s (level means), so the passes that exist
 * solely for sp are skipped -- the covariance accumulates one Gram matrix (D^T S) instead of three, TransformedMoments
 * skip their diff-Gram pass; the skipped sp come back as NaN.  (Estimate.construct_density, estimator.py:304-331, uses
 * only the means of both of its estimates.) */
#define MLMC_MODE_MEAN_ONLY 0x100
int mlmc_accum_create(const mlmc_basis *b, int32_t n_levels, int32_t mode, int32_t n_comp, mlmc_accum **out);
void mlmc_accum_destroy(mlmc_accum *a);
int mlmc_accum_reset(mlmc_accum *a);
/* One chunk of level `level`: fine[m * n + k], coarse[m * n + k] (coarse == NULL at level 0:
 * SampleStorage.sample_pairs_level returns [M, n, 1] there, sample_storage.py:261-285).
 * Replaces eval_moments/eval_cov + mask_nan_samples + the two np.sum of quantity_estimate.py:43-65.
 * Asynchronous on the library's stream.  Host buffers are staged before the call returns; DEVICE buffers must stay
 * valid until the next mlmc_accum_finalize / mlmc_accum_reset: chunks of different levels are gathered and processed
 * by ONE kernel launch (one grid for a whole multi-level estimate). */
int mlmc_accum_push(mlmc_accum *a, int32_t level, const double *fine, const double *coarse, int64_t n, int mem_kind);
/* Writes per level l: n[l] (kept samples), n_rm[l] (NaN-masked samples), s[l * K + k] = sum of
 * level differences, sp[l * K + k] = sum of squared differences (quantity_estimate.py:46-47,64-65).
 * Outputs in host or device memory (device: for an RCCL all-reduce over ranks before the host reads them).
 * Synchronises the stream; idempotent. */
int mlmc_accum_finalize(mlmc_accum *a, int64_t *n, int64_t *n_rm, double *s, double *sp, int mem_kind);
/* reset + push of n_chunks chunks + finalize in one call (one host round trip per estimate): chunk k is
 * (levels[k], fine[k], coarse[k] or NULL, n[k]); all chunk buffers of one kind (mem_kind).  Results as mlmc_accum_finalize
 * with MLMC_HOST outputs. */
int mlmc_accum_estimate(mlmc_accum *a, int32_t n_chunks, const int32_t *levels, const double *const *fine,
                        const double *const *coarse, const int64_t *n_samples, int mem_kind, int64_t *n, int64_t *n_rm,
                        double *s, double *sp);
/* The same with the results as mlmc_accum_finalize_packed leaves them (one fp64 buffer n | n_rm | s | sp in host or
 * device memory): the rank-local half of a multi-GPU estimate, stream-ordered before the all-reduce when device. */
int mlmc_accum_estimate_packed(mlmc_accum *a, int32_t n_chunks, const int32_t *levels, const double *const *fine,
                               const double *const *coarse, const int64_t *n_samples, int mem_kind, double *packed,
                               int packed_kind);
/* Same results in ONE fp64 buffer [n(L) | n_rm(L) | s(L*K) | sp(L*K)] (counts as exact doubles): a single packed
 * all-reduce (RCCL) then carries everything a multi-GPU estimate has to exchange.  With MLMC_DEVICE the call is
 * asynchronous (stream-ordered); with MLMC_HOST it synchronises. */
int mlmc_accum_finalize_packed(mlmc_accum *a, double *packed, int mem_kind);
/* HIP-event time (ms) and launch count of the dominant accumulation kernel since create or since the previous
 * call of this function (it returns the totals and clears them; mlmc_accum_reset leaves them alone; needs
 * mlmc_init flag bit0); also the algorithmic HBM bytes those launches had to read.  Waits for the last launch. */
int mlmc_accum_kernel_time(mlmc_accum *a, double *ms, int64_t *launches, int64_t *alg_bytes);
/* Matrix-core flops the covariance launches counted by mlmc_accum_kernel_time have EXECUTED since create or since the
 * previous call of this function: one v_mfma_f64_16x16x4_f64 is 2 * 16 * 16 * 4 flops and covers four samples, i.e. 512
 * flops per 16 x 16 output tile and sample, times the tiles the kernels' tile lists name.  Symmetric Gram tiles are computed
 * once (R = 64, pair level, mean + variance: 16 + 16 + 10 = 42 tiles instead of 48), so the figure is BELOW the reference-
 * form count 6 R^2 per pair (quantity_estimate.py:131-147); it is the numerator of a physical matrix-pipe fraction.
 * Returns the total and clears it. */
int mlmc_accum_kernel_flops(mlmc_accum *a, int64_t *mfma_flops);
/* A MLMC_MODE_COV accumulator WITH variances of 17..128 plain Legendre or monomial moments splits the work of its large chunks
 * (>= 10^5 samples at 33..128 moments, >= 1.5 x 10^6 at 17..32; MLMC_HIP_LINEARIZE_MIN_N overrides; smaller chunks keep all
 * three Gram matrices, the level sums are additive): the matrix cores
 * accumulate only the two Gram matrices of the VARIANCE (G1, G2: 26 instead of 42 tiles per pair at R = 64), and the MEAN
 * (quantity_estimate.py:131-147 + :59-65: level sums of f_i f_j - c_i c_j) comes from the level sums of the 2 R - 1 moments
 * of the same family through the product linearisation phi_i phi_j = sum_k c_ijk phi_k (Legendre: Adams' formula, non-negative
 * coefficients that sum to one) -- one mean-only pass of the moments kernel over the same chunks with the same keep / drop
 * decisions, contracted on the device at finalize.  Same outputs, same counts; the means agree with the direct sums to rounding
 * (a convex combination of sums instead of sums of products).  A large chunk WITHOUT coarse values (level 0) of <= 64 such
 * moments needs no matrix pass at all: with one value per sample the second moments linearise too, (phi_i phi_j)^2 =
 * sum_k c2_ijk phi_k with k < 4 R - 3, so the level sums of 4 R - 3 moments (two windows of the same mean-only kernel, which
 * then also does the counting) give both sum f_i f_j and sum (f_i f_j)^2; MLMC_HIP_LINEARIZE_LEVEL0=0 keeps level 0 on the matrix
 * cores.  MLMC_HIP_LINEARIZE=0 in the environment keeps all three Gram matrices of every chunk on the matrix cores.  This call
 * reports the HIP-event time, launches and algorithmic bytes of the auxiliary moments passes (zeros when the accumulator has
 * none), like mlmc_accum_kernel_time does for the matrix-core launches. */
int mlmc_accum_aux_kernel_time(mlmc_accum *a, double *ms, int64_t *launches, int64_t *alg_bytes);
/* The coefficient tables of those linearisations, as the accumulators use them (host arithmetic only: needs no device, e.g.
 * for checking them against exact rational values): squares == 0: phi_i phi_j = sum_k c_ijk phi_k, K = 2 R - 1;
 * squares != 0: (phi_i phi_j)^2 = sum_k c2_ijk phi_k, K = 4 R - 3.  out [K][R * R] (k-major), out_len >= K * R * R.
 * kind: MLMC_LEGENDRE or MLMC_MONOMIAL, 1 <= R <= 128 (squares: R <= 64). */
int mlmc_linearization_table(int32_t kind, int32_t R, int32_t squares, double *out, int64_t out_len);

/* ---- maximum-entropy density (mlmc/tool/simple_distribution.py:9-327) ------------------ */
typedef struct {
    double tol;          /* gradient-norm tolerance (estimate_density_minimize tol, :50) */
    int32_t max_it;      /* Newton iterations (reference: trust-ncg maxiter 20, :60) */
    int32_t n_intervals; /* composite Gauss-Legendre sub-intervals on [a, b] (0 = default 64) */
    int32_t gauss_degree;/* points per sub-interval (reference uses 21, :45); 0 = 21 */
    int32_t reserved;
    double stab_penalty; /* Distribution._stab_penalty (distribution.py:236); 0 for SimpleDistribution */
    double penalty_coef; /* end-point decay penalty coefficient (distribution.py:47 = 10; simple: 0) */
    int32_t decay_left, decay_right; /* force_decay flags */
} mlmc_maxent_opts;

typedef struct {
    int32_t nit;
    int32_t success;
    double fun;        /* final functional value */
    double grad_norm;  /* ||gradient||_2 at the solution (result.fun_norm, :93) */
    double moment0;    /* integral of the density before the normalisation fix (:81-86) */
    int32_t n_quad;
    int32_t reserved;
} mlmc_maxent_info;

/* Minimise F(l) = sum_i mu_i l_i / sigma_i + int exp(-phi(x).l/sigma) dx (simple_distribution.py:259-327)
 * starting from lambda_io (size R1 = number of moments used, <= basis out size); on return lambda_io holds
 * the multipliers (normalisation fix of :86 NOT applied; see moment0), grad_out / hess_out (may be NULL) the final
 * gradient [R1] and Hessian [R1 * R1].  prev_lambda/n_prev: Distribution's stabilisation term (distribution.py:358-359), may be NULL/0. */
int mlmc_maxent_solve(const mlmc_basis *b, const double *mu, const double *sigma, int32_t R1, double a, double bnd_b,
                      const mlmc_maxent_opts *opts, const double *prev_lambda, int32_t n_prev, double *lambda_io,
                      double *grad_out, double *hess_out, mlmc_maxent_info *info);
/* SimpleDistribution.density (:96-105): out[i] = exp(clip(-phi(x_i).lambda/sigma, -200, 200)) */
int mlmc_density_eval(const mlmc_basis *b, const double *lambda, const double *sigma, int32_t R1, const double *x,
                      int64_t n, double *out, int mem_kind);
/* integral of the density over [lo_i, hi_i] by `degree`-point Gauss-Legendre per interval (cdf :108-125) */
int mlmc_density_integrate(const mlmc_basis *b, const double *lambda, const double *sigma, int32_t R1, const double *lo,
                           const double *hi, int64_t n, int32_t degree, double *out);

/* ---- sample percentiles (Estimate.estimate_domain, mlmc/estimator.py:275-302) -------------------------- */
/* out[i] = np.percentile(x[~isnan(x)], q_percent[i]) (NumPy "linear" method), bit-identical: exact order statistics by
 * a radix select on the device + NumPy's interpolation formula.  n_valid (may be NULL) = number of non-NaN values. */
int mlmc_percentiles(const double *x, int64_t n, const double *q_percent, int32_t nq, double *out, int64_t *n_valid,
                     int mem_kind);

/* ---- quantity expressions (mlmc/quantity/quantity.py:35-512: arithmetic, NumPy ufuncs, comparisons, select) ------
 * A lazily built Quantity tree over one storage is lowered by the host into a straight-line register program that a
 * per-sample byte-code kernel evaluates in ONE pass over the stored rows: every node of the tree is fused, nothing but
 * the result rows is written.  A register holds the fine and the coarse value of one row of one sample.
 * Comparisons follow Quantity._process_mask (:250-262): the result is one flag per sample, true only if the condition
 * holds for the fine AND the coarse value; MLMC_X_SELECT marks the flags that `Quantity.select` (:137-164) applies. */
enum {
    MLMC_X_LOAD = 0,   /* dst = stored row a                       */
    MLMC_X_CONST,      /* dst = imm                                */
    MLMC_X_STORE,      /* result row b = reg a                     */
    MLMC_X_SELECT,     /* keep the sample only if reg a != 0       */
    MLMC_X_ADD, MLMC_X_SUB, MLMC_X_MUL, MLMC_X_DIV,
    MLMC_X_MOD,        /* np.remainder (floored, sign of divisor)  */
    MLMC_X_POW, MLMC_X_MAXIMUM, MLMC_X_MINIMUM, MLMC_X_FMAX, MLMC_X_FMIN, MLMC_X_ATAN2, MLMC_X_HYPOT, MLMC_X_FMOD,
    MLMC_X_NEG, MLMC_X_ABS, MLMC_X_SQRT, MLMC_X_SQUARE, MLMC_X_RECIP, MLMC_X_EXP, MLMC_X_EXP2, MLMC_X_EXPM1,
    MLMC_X_LOG, MLMC_X_LOG2, MLMC_X_LOG10, MLMC_X_LOG1P, MLMC_X_SIN, MLMC_X_COS, MLMC_X_TAN, MLMC_X_ASIN, MLMC_X_ACOS,
    MLMC_X_ATAN, MLMC_X_SINH, MLMC_X_COSH, MLMC_X_TANH, MLMC_X_FLOOR, MLMC_X_CEIL, MLMC_X_TRUNC, MLMC_X_RINT,
    MLMC_X_SIGN, MLMC_X_CBRT,
    MLMC_X_LT, MLMC_X_LE, MLMC_X_GT, MLMC_X_GE, MLMC_X_EQ, MLMC_X_NE,   /* per-sample flag (fine AND coarse), 0.0 / 1.0 */
    MLMC_X_AND, MLMC_X_OR, MLMC_X_NOT, MLMC_X_XOR,                        /* on flags */
    MLMC_X_N_OPS
};
/* flags or-ed into `op` of an arithmetic (ADD..FMOD) or comparison instruction: the operand is `imm`, not a register */
#define MLMC_X_IMM_A 0x4000
#define MLMC_X_IMM_B 0x8000
/* chaining: the result of the latest value-producing instruction (everything but STORE / SELECT) also stays in VGPRs.
 * A_PREV / B_PREV: the operand is that result (the register index is ignored); NO_WB on a producing instruction: the
 * result is read only through such chained operands and is not written to a register (`dst` is ignored).  Optional --
 * a program without these flags computes the same rows, with every value passing through the LDS register file. */
#define MLMC_X_A_PREV 0x2000
#define MLMC_X_B_PREV 0x1000
#define MLMC_X_NO_WB 0x0800
#define MLMC_X_OP_MASK 0x07ff
typedef struct {
    uint16_t op, dst, a, b;   /* registers < n_regs; LOAD: a = input row; STORE: b = output row */
    double imm;               /* CONST value, or the immediate operand */
} mlmc_expr_instr;
#define MLMC_EXPR_MAX_REGS 16
#define MLMC_EXPR_MAX_INSTR 4096
typedef struct mlmc_expr mlmc_expr;
int mlmc_expr_create(const mlmc_expr_instr *prog, int32_t n_instr, int32_t n_regs, int32_t n_in_rows, int32_t n_out_rows,
                     mlmc_expr **out);
void mlmc_expr_destroy(mlmc_expr *e);
/* Evaluate for n samples.  rows_in: host array of n_in_rows DEVICE pointers to the first fine value of each stored
 * row; value (sample i, fine) = row[i * sample_stride], (sample i, coarse) = row[i * sample_stride + side_stride] (doubles).
 * Two layouts occur: a row uploaded on its own -- interleaved (fine, coarse) pairs [n][2], strides (2, 1), level 0 [n],
 * stride 1 -- and a row inside an uploaded storage block [n][2][M] (the layout of the reference's Memory storage and HDF5
 * `collected_values`, sample_storage.py:169-184): pointer block + m, strides (2 M, M): fine / coarse are de-interleaved on
 * the device, the host never reshuffles.
 * fine_out / coarse_out: device buffers of n_out_rows * n doubles (coarse_out ignored without has_coarse).  Without
 * MLMC_X_SELECT the result rows are [n_out_rows][n]; with it the selected samples are compacted in order and the rows
 * are [n_out_rows][*n_selected] contiguous.  n_selected (host) receives the surviving sample count; the call
 * synchronises only when the program selects. */
int mlmc_expr_eval(mlmc_expr *e, const double *const *rows_in, int32_t has_coarse, int64_t n, int64_t sample_stride,
                   int64_t side_stride, double *fine_out, double *coarse_out, int64_t *n_selected);
/* Which form evaluates this program: *state = 0 the byte-code interpreter (the program has not reached its compile threshold
 * yet), 2 its own compiled kernel (hiprtc, gfx950), -1 no compiled form (hiprtc missing, compile or load error: the
 * interpreter keeps the program), -2 compiled forms are not applicable (program too long); *compiled_launches = evaluations of
 * THIS handle that ran the compiled kernel.  Either pointer may be NULL.  (MLMC_EXPR_JIT=0 keeps the interpreter,
 * MLMC_EXPR_JIT_AFTER=k compiles at the (k+1)-th evaluation, MLMC_EXPR_JIT_VERBOSE=1 prints the hiprtc log on failure.) */
int mlmc_expr_state(mlmc_expr *e, int32_t *state, int64_t *compiled_launches);
/* HIP-event time (ms), launches and algorithmic bytes (8 B per value of every referenced stored row and every result
 * row) of the evaluation kernel since create or the previous call; same contract as mlmc_accum_kernel_time. */
int mlmc_expr_kernel_time(mlmc_expr *e, double *ms, int64_t *launches, int64_t *alg_bytes);

/* ---- bootstrap sub-sampling (Quantity.pick_samples, mlmc/quantity/quantity.py:308-325; Estimate.est_bootstrap,
 * mlmc/estimator.py:171-205) ------------------------------------------------------------------------------------
 * out[r][j] = in[r][idx_j], j < k, idx_j uniform in [0, n) with replacement (RNG.choice(chunk, size=k, axis=1)); the
 * same idx_j for every row and for fine and coarse.  idx_j comes from Philox4x32-10 keyed by `seed` with counter j, so a
 * draw is reproducible from (seed, n, k) alone; the reference's generator is an unseeded module global, parity is
 * statistical only.  All pointers are DEVICE pointers: fine / coarse [n_rows][n] (coarse may be NULL),
 * fine_out / coarse_out [n_rows][k].  Asynchronous on the library's stream. */
int mlmc_subsample_gather(const double *fine, const double *coarse, int32_t n_rows, int64_t n, int64_t k, uint64_t seed,
                          double *fine_out, double *coarse_out);

/* ---- synthetic samples in HBM (mlmc/sim/synth_simulation.py:37-46,75-131; seeding mlmc/sampling_pool.py:75-84; sample
 * ids mlmc/sampler.py:120) -------------------------------------------------------------------------------------------
 * Samples first_sample .. first_sample + n - 1 of level `level_id` exactly as Sampler + SynthSimulation (distr =
 * scipy.stats.norm(loc, scale), result_format of synth_simulation.py:136-145: 24 stored rows) produce them: md5 of the sample id
 * -> MT19937 -> two legacy Box-Muller normals -> x + h sqrt(1e-4 + |x|).  rows (host array): the stored rows wanted
 * (0..23 = [quantity][time][location][component]); out (host array of DEVICE pointers): one buffer per row in the
 * storage layout -- interleaved (fine, coarse) pairs [n][2] when coarse_step != 0, else fine only [n] (level 0).
 * Asynchronous on the library's stream. */
int mlmc_synth_generate(int32_t level_id, int64_t first_sample, int64_t n, double fine_step, double coarse_step,
                        double loc, double scale, int32_t n_rows, const int32_t *rows, double *const *out);
/* seeds_host[i] = SamplingPool.compute_seed("L{level:02d}_S{first + i:07d}") (first uint32 of the md5 digest); synchronous */
int mlmc_synth_seeds(int32_t level_id, int64_t first_sample, int64_t n, uint32_t *seeds_host);

#ifdef __cplusplus
}
#endif
#endif /* MLMC_HIP_H */
