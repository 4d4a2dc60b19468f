/*
 * seerhip.h -- flat C ABI of libseerhip.so, the MI355X (gfx950) per-variant association engine.
 *
 * The reference (pyseer) is pure Python and has NO FFI for this path; the boundary it exposes is the
 * Python call signature between its variant stream and its per-variant test:
 *     pyseer/model.py:202  fixed_effects_regression(variant,p,k,m,c,af,pattern,lineage_effects,lin,pret,lrtt,
 *                                                   null_res,null_firth,kstrains,nkstrains,continuous) -> Seer
 *     pyseer/lmm.py:125    fit_lmm(lmm,h2,variants,variant_mat,lineage_effects,lineage_clusters,covariates,
 *                                  continuous,filter_pvalue,lrt_pvalue) -> [LMM]
 *     pyseer/lmm.py:228    fit_lmm_block(lmm,h2,variant_block) -> {beta,bse,frac_h2,p_values}
 * Each entry point below names the reference call it replaces.  INTEGRATION.md shows the ctypes stub a
 * pyseer maintainer would add (pyseer_amd/_abi.py is that stub, in full).
 *
 * Conventions
 *  - return 0 on success, a negative SH_E* code otherwise; sh_last_error() gives a thread-local message.
 *  - no CPU fallback: every compute entry point needs a gfx950 device (sh_create fails without one).
 *  - host-pointer calls (sh_*_batch) copy in/out and synchronise; *_dev calls take DEVICE pointers, are
 *    asynchronous on the context's stream (sh_set_stream) and touch no host memory.
 *  - a context is bound to one device and is NOT thread-safe; use one context per device per host thread.
 *  - presence bits: V rows x row_bytes bytes, variant-major; bit (i & 7) of byte (i >> 3) of row v is the
 *    presence of sample i in variant v (LSB first).  Bits at i >= n_samples are ignored.
 *  - flags[v]: bits 0..8 = notes in the order of docs/usage.rst:553-566 (SH_NOTE_*), bit 18 = SH_FLAG_FIRTH_SENSITIVE, bit 16 = Seer/LMM.prefilter,
 *    bit 17 = Seer/LMM.filter.
 */
#ifndef SEERHIP_H
#define SEERHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SH_ABI_VERSION 2

#define SH_OK          0
#define SH_EINVAL     -1   /* bad argument / call order */
#define SH_ENODEV     -2   /* no gfx950 device / device error */
#define SH_ENOMEM     -3
#define SH_EH2        -4   /* h2 outside [0,1): reference raises KeyError('beta') (lmm_cov.py:667-670) */
#define SH_ESHAPE     -5   /* shape mismatch: reference AssertionError (lmm_cov.py:674) */
#define SH_EHIP       -6   /* HIP runtime failure */

/* notes (pyseer/model.py:253-386, pyseer/lmm.py:160-204) */
#define SH_NOTE_AF_FILTER        (1u << 0)
#define SH_NOTE_PRE_FILTER       (1u << 1)
#define SH_NOTE_BAD_CHISQ        (1u << 2)
#define SH_NOTE_HIGH_BSE         (1u << 3)
#define SH_NOTE_PERFECT_SEP      (1u << 4)
#define SH_NOTE_MATRIX_INV       (1u << 5)
#define SH_NOTE_FIRTH_FAIL       (1u << 6)
#define SH_NOTE_MISSING_DATA     (1u << 7)
#define SH_NOTE_LRT_FILTER       (1u << 8)
#define SH_FLAG_PREFILTER        (1u << 16)
#define SH_FLAG_FILTER           (1u << 17)
/* A row fitted by fit_firth (model.py:414-504) on which the REFERENCE's own answer is fragile -- it may depend on the order of the reference's
 * floating-point sums.  Set when
 *   (a) the variant all but separates the phenotype: a cell of its 2 x 2 table holds at most one sample.  Every `firth-fail` the reference
 *       itself was seen to return (tests/golden/glm_exit_firthfail_*.npz: 13 rows, N = 300 ... 5000) is such a row: near its stop
 *       `firth_likelihood(new) > firth_likelihood(old)` (model.py:467) compares values one ulp apart, for 1000 halvings if the last bits
 *       fall that way, and the reference fits the same row with the samples in another order.  Nothing order-independent tells those rows
 *       from their neighbours that the reference fits (profiles/r05/firthfail_trace.txt), so the whole class is marked;
 *   (b) the iteration took 12 or more accepted steps (linear convergence), or its stop rule (model.py:477-479) was met within 1e-8 of the
 *       limit: the answer then moves by a whole step (~1e-4) with the last bit of a norm;
 *   (c) this library itself reports firth-fail (SEERHIP_ROUTE firth_literal / firth_strict).
 * Informational: set beside the row's statistics, never a reason to filter; < 1e-3 of the rows of a forced-Firth run on random k-mers.
 * DESIGN.md section 6. */
#define SH_FLAG_FIRTH_SENSITIVE  (1u << 18)

typedef struct sh_ctx sh_ctx;

int         sh_abi_version(void);
const char *sh_last_error(void);
int         sh_device_count(void);
/* Start the HIP runtime on `device` and load this library's code object (a no-op launch): about 0.8 s of driver work on first use, which
 * a caller can run on a side thread while it reads its inputs (the reference has no counterpart: its workers start with the interpreter).
 * Optional: every other entry point initialises what it needs. */
int         sh_warmup(int device);

/* How host threads wait for the device: sleeping != 0 asks for waits that SLEEP (hipDeviceScheduleBlockingSync, offered to each device at its
 * first sh_warmup / sh_create; a device that something else initialised first keeps its mode).  The runtime's default is to spin: one CPU per
 * waiting stream -- fine for a benchmark, not for a job of several device streams under a CPU quota (the command line asks for sleeping
 * waits).  Call before the first sh_warmup / sh_create.  Process-wide. */
void        sh_set_wait_mode(int sleeping);

/* one context per device; n_samples = len(p) of the reference */
sh_ctx *sh_create(int device, int n_samples);
void    sh_destroy(sh_ctx *ctx);
/* enqueue all later work on this hipStream_t (NULL = the device's default stream) */
int     sh_set_stream(sh_ctx *ctx, void *hip_stream);
int     sh_synchronize(sh_ctx *ctx);
/* HIP-event timing of the dominant kernel(s) of each later *_batch_dev call, recorded on the context's stream: the contraction kernel
 * (k_lmm_quadform_i8w, or k_lmm_quadform_i8 where the former's conditions do not hold; main pass) for the LMM; every fit kernel of a fixed-effects batch (Newton, Firth rounds, final pass; the bit repack is outside).
 * sh_get_timing synchronises on the recorded events and returns their sum. */
int     sh_set_timing(sh_ctx *ctx, int on);
int     sh_get_timing(sh_ctx *ctx, double *total_ms, int64_t *launches);
/* Pattern de-duplication (pyseer/input.py:710 hash_pattern, scripts/count_patterns.py): when on, every later
 * *_batch / *_batch_dev call tests each DISTINCT packed row once (exact row comparison, the hash only finds candidates) and fans
 * the result out to all variants that share it.  Outputs are unchanged.  Costs one stream synchronisation per batch. */
int     sh_set_dedup(sh_ctx *ctx, int on);
int     sh_dedup_info(sh_ctx *ctx, int64_t *unique_last_batch);
/* AF filter of the variant stream (pyseer/input.py:608,693): keep min_af <= count/n <= max_af (inclusive);
 * others get SH_NOTE_AF_FILTER | SH_FLAG_PREFILTER and NaN statistics.  Default: disabled (0, 1). */
int     sh_set_af_filter(sh_ctx *ctx, double min_af, double max_af);

/* ---------------------------------------------------------------------------------------------
 * LMM  (replaces pyseer/lmm.py:125 fit_lmm + :228 fit_lmm_block over pyseer/fastlmm/lmm_cov.py:165,597,686)
 * U: n x k row-major (lmm.U; arr_0 of the --save-lmm cache), S: k (lmm.S), y: n (lmm.Y),
 * C: n x D row-major covariates with the intercept LAST (lmm.X, pyseer/lmm.py:95-99), h2 (lmm.py:115).
 * continuous selects the prefilter test (model.py:52-55 vs :57-68); pret/lrtt = filter_pvalue/lrt_pvalue
 * (compared with >=, lmm.py:174,201).  n_limbs = 3..7 int8 limbs of the fixed-point kernel matrix; 0 = automatic:
 * the smallest of 4 and 5 whose typical a-posteriori bound on x^T K^-1 x is at most a quarter of the tolerance of sh_set_lmm_tol (1e-8; a variant
 * whose own bound exceeds the tolerance is re-contracted with the extra limbs whichever count was chosen; sh_lmm_info / sh_lmm_bound report both).
 * With pret < 1 a pre-filtered variant is not fitted and carries NaN statistics (fit_lmm); with pret >= 1 every AF-passing variant
 * carries its statistics (fit_lmm_block) and masking by the prefilter flag is the caller's.
 * ------------------------------------------------------------------------------------------- */
int sh_lmm_setup(sh_ctx *ctx, const double *U, const double *S, int k, const double *y,
                 const double *C, int D, double h2, int continuous, double pret, double lrtt, int n_limbs);
/* Multi-GPU (the counterpart of the reference's --cpu N workers sharing one LMM object, pyseer/__main__.py:541-568):
 * copy the per-run LMM state of `src` (after sh_lmm_setup) device-to-device into `dst`, a context on another (or the same) device with
 * the same n_samples.  One set-up per run instead of one per GPU; the per-variant path has no inter-GPU traffic at all. */
int sh_lmm_share(sh_ctx *dst, sh_ctx *src);
/* per variant: prep, pvalue, beta, bse, frac_h2 (each V doubles) + flags.  Statistics are written for every
 * variant that passes the AF filter (the caller applies the NaN masking implied by flags, lmm.py:176-217). */
int sh_lmm_batch(sh_ctx *ctx, const uint8_t *bits, int64_t row_bytes, int64_t V,
                 double *prep, double *pvalue, double *beta, double *bse, double *frac_h2, uint32_t *flags);
/* Pipelined form of the host-pointer batches (the counterpart of the reference's pool.imap over blocks of variants, pyseer/__main__.py:541-568:
 * block k+1 is submitted while block k's results are still being produced).  sh_lmm_batch_async / sh_glm_batch_async take the arguments of
 * sh_lmm_batch / sh_glm_batch, stage all of `bits` (the caller may release it on return) and queue the batch, but return while its LAST
 * chunk is still on the device: the result arrays of a call are complete when the NEXT sh_*_batch / sh_*_batch_async call on the context
 * returns (it copies them back after queueing its own first chunk, so the device never idles between calls), or after sh_wait(ctx).
 * The caller keeps the result arrays alive until then.  Results are bit-identical to the synchronous calls. */
int sh_lmm_batch_async(sh_ctx *ctx, const uint8_t *bits, int64_t row_bytes, int64_t V,
                       double *prep, double *pvalue, double *beta, double *bse, double *frac_h2, uint32_t *flags);
int sh_wait(sh_ctx *ctx);
/* Announce the rows of the batch AFTER the next one: called before batch k with the `bits` of batch k+1 (same row_bytes), it lets batch k's
 * copy thread upload the first chunk of batch k+1 while batch k's last chunk runs, so that batch k+1 starts on the device at once (the
 * fixed-effects launch code blocks on its read-backs: without the announcement every call exposes its first upload, 2.8 of 12 ms per 262 144
 * rows).  `bits` must stay valid AND UNCHANGED until batch k+1 has been called with exactly this pointer (the rows may be uploaded at any
 * time in between; the call only probes two 4 KiB windows of them before it trusts the early upload); a batch called with other rows
 * ignores the announcement, and so does anything after sh_*_setup or sh_set_stream.  Optional; results do not depend on it. */
int sh_prefetch_rows(sh_ctx *ctx, const uint8_t *bits, int64_t row_bytes, int64_t V);
/* device-resident variant: d_bits (V*row_bytes bytes), d_out (5*V doubles, SoA in the order above), d_flags (V). */
int sh_lmm_batch_dev(sh_ctx *ctx, const void *d_bits, int64_t row_bytes, int64_t V, void *d_out, void *d_flags);
/* introspection for tests/benchmarks: dominant-kernel work of the last batch */
int sh_lmm_info(sh_ctx *ctx, int *n_limbs, int64_t *int8_macs_per_variant, double *quant_scale);
/* Accuracy of the fixed-point contraction that replaces the reference's fp64 U.T.dot(A) (pyseer/fastlmm/lmm_cov.py:186-193) and
 * computeAKA (:885-900).  The kernel matrix G is held as n_limbs int8 limbs; the only error of x^T K^-1 x is that quantisation, and for a
 * stored row with m' carriers it is bounded by err_norm * ulp * m', err_norm = an upper bound on the spectral norm of the (symmetrised)
 * quantisation error in units of ulp, certified at set-up (sh_lmm_bound_estimate below).  A variant whose relative bound err_norm*ulp*m'/xKx exceeds `tol`
 * (default 1e-8, sh_set_lmm_tol; 0 = never) is contracted again with `extra_limbs` more limbs (up to 56 bits in all = fp64) inside the
 * same call.  bound_typical: the bound of a variant carried by half of the samples on an unstructured population; bound_max_last /
 * refined_last: the largest final bound and the number of re-contracted variants of the LAST batch (synchronises the stream). */
int sh_set_lmm_tol(sh_ctx *ctx, double tol);
int sh_lmm_bound(sh_ctx *ctx, double *err_norm_ulp, double *ulp, double *tol, int *extra_limbs, double *bound_typical,
                 double *bound_max_last, int64_t *refined_last);
/* err_norm (sh_lmm_bound) is a CERTIFIED upper bound since round 3: |E|_2 <= trace(E^(2p))^(1/(2p)) with 2p = 2^(squarings+1), E^(2p) by repeated
 * squaring on the fp64 matrix pipe (64th power at n <= 16384: within 4.4 % of the norm for a Wigner-like error matrix).  The power iteration of
 * rounds 1-2 (which approaches the norm from below and can stall between near-equal or opposite eigenvalues) remains as the estimate reported here.
 * The reference needs neither: it contracts in fp64 (pyseer/fastlmm/lmm_cov.py:186-193, 885-900). */
int sh_lmm_bound_estimate(sh_ctx *ctx, double *power_iteration_ulp, int *squarings);
/* The two norm routines of sh_lmm_setup on a caller-supplied symmetric n x n fp32 matrix (row-major): diagnostic / test entry. */
int sh_spectral_bound_f32(sh_ctx *ctx, const float *A, int n, int squarings, double *upper, double *power_iteration);

/* ---------------------------------------------------------------------------------------------
 * Fixed effects (replaces pyseer/model.py:202 fixed_effects_regression = a1 prefilter + statsmodels Logit newton /
 * OLS + model.py:414 fit_firth).  y: n; W: n x q row-major = [m | c] (MDS components then covariates, no
 * intercept, no variant column; model.py:274-297); null_llf / null_firth from fit_null (model.py:73-148).
 * force_firth != 0 sends every variant through fit_firth (benchmark config C4).
 * ------------------------------------------------------------------------------------------- */
int sh_glm_setup(sh_ctx *ctx, const double *y, const double *W, int q, int continuous,
                 double null_llf, double null_firth, double pret, double lrtt, int force_firth);
/* outputs: prep, pvalue, kbeta, bse, intercept (V each), betas (V*q row-major), flags (V) */
int sh_glm_batch(sh_ctx *ctx, const uint8_t *bits, int64_t row_bytes, int64_t V,
                 double *prep, double *pvalue, double *kbeta, double *bse, double *intercept,
                 double *betas, uint32_t *flags);
/* pipelined form: see sh_lmm_batch_async */
int sh_glm_batch_async(sh_ctx *ctx, const uint8_t *bits, int64_t row_bytes, int64_t V,
                       double *prep, double *pvalue, double *kbeta, double *bse, double *intercept, double *betas, uint32_t *flags);
/* d_out: (5+q)*V doubles SoA: prep,pvalue,kbeta,bse,intercept,betas[0..q) ; d_flags: V.
 * Device memory: the context keeps per-variant workspaces sized for the largest batch seen (logistic with 1..14 covariates:
 * (3(q+2) + 1.5 (q+2)(q+3)/2 + 6) x 8 bytes per variant, 0.9 GB for 2^20 variants at q = 10); sh_glm_batch cuts host batches into
 * 2^18-variant chunks.  A variant's result does not depend on what else is in its batch or on the batch size. */
int sh_glm_batch_dev(sh_ctx *ctx, const void *d_bits, int64_t row_bytes, int64_t V, void *d_out, void *d_flags);
/* The same batch on one of the context's LANES (csrc/lanes_api.inc; the counterpart of the reference's pool of `--cpu N` workers over
 * blocks of variants, pyseer/__main__.py:541-568): a lane is a worker thread with its own stream and per-batch workspaces, set up from the
 * arguments sh_glm_setup was called with.  The call records an event on the context's stream (the rows are ready there), hands the batch to
 * the next free lane and returns; batches of consecutive calls run side by side on the device (one stream of fixed-effects batches leaves it
 * idle between its per-variant kernels and behind the list-length read-backs of its launch code).  d_out / d_flags of a call are complete
 * after sh_wait(ctx) -- which also reports the first error of a lane -- and bit-identical to sh_glm_batch_dev's (a batch is still one
 * sh_glm_batch_dev on one stream).  At most 2 x lanes batches are accepted before the call blocks.  sh_set_lanes: 1..8, default 3; changing
 * it waits for the lanes and tears them down (they are created at the next use).  The job stream (sh_job_*) of a fixed-effects model
 * computes its blocks on the lanes too. */
int sh_glm_batch_dev_async(sh_ctx *ctx, const void *d_bits, int64_t row_bytes, int64_t V, void *d_out, void *d_flags);
int sh_set_lanes(sh_ctx *ctx, int n);
int sh_get_lanes(sh_ctx *ctx);

/* ---------------------------------------------------------------------------------------------
 * Lineage effect (replaces pyseer/model.py:151 fit_lineage_effect): logistic regression of each VARIANT on
 * [1, lin, cov]; lin: n x l row-major (MDS components or cluster indicators, pyseer/__main__.py:417-432), cov: n x j
 * (or NULL, j = 0).  max_lineage[v] = argmax_j |beta_j|/bse_j over the l lineage columns, -1 = None (separation or
 * singular, model.py:193-197).  1 + l + j <= 16 in this build.
 * ------------------------------------------------------------------------------------------- */
int sh_lineage_setup(sh_ctx *ctx, const double *lin, int l, const double *cov, int j);
int sh_lineage_batch(sh_ctx *ctx, const uint8_t *bits, int64_t row_bytes, int64_t V, int32_t *max_lineage);

/* ---------------------------------------------------------------------------------------------
 * Similarity (kinship) matrix from variant presence: replaces pyseer/similarity.py:99-113 (G assembled from
 * load_var_block, pyseer/input.py:678-707, then np.matmul(G, G.T)).  K[i][j] = number of variants carried by both
 * samples among those passing sh_set_af_filter (filtered variants are all-zero columns in the reference, input.py:690-697).
 * sh_sim_begin zeroes the accumulator; accumulate any number of batches (host or device rows, same packed layout as
 * sh_lmm_batch); sh_sim_finish writes the dense N*N row-major fp64 matrix (exact integers) to host memory.
 * --------------------------------------------------------------------------------------------- */
int sh_sim_begin(sh_ctx *ctx);
int sh_sim_accumulate(sh_ctx *ctx, const uint8_t *bits, int64_t row_bytes, int64_t V);
int sh_sim_accumulate_dev(sh_ctx *ctx, const void *d_bits, int64_t row_bytes, int64_t V);
int sh_sim_finish(sh_ctx *ctx, double *K);

/* ---------------------------------------------------------------------------------------------
 * Native k-mer text reader / packer (host code; replaces the k-mer branch of pyseer/input.py:301 read_variant for the GPU
 * feed).  Lines "KMER | sample:count sample:count ..." (gzip or plain) -> packed presence rows over `sample_names`
 * (= p.index, phenotype order), carrier counts (af = count / n, input.py:446) and the variant names.
 * ------------------------------------------------------------------------------------------- */
typedef struct sh_reader sh_reader;
sh_reader  *sh_reader_open(const char *path, const char *const *sample_names, int n_samples);
void        sh_reader_close(sh_reader *r);
const char *sh_reader_error(void);
/* how many readers the caller is about to run at once (`--kmers a.gz b.gz ...` = one per file, pyseer/__main__.py:526 reads ONE): readers
 * opened afterwards share the usable CPUs (cgroup quota) between them instead of starting a full worker pool each.  Process-wide hint. */
void        sh_reader_set_concurrency(int n_readers);
/* parses up to max_variants lines; returns how many (0 = end of file, -1 = error, -2 = the variant names of this block need more
 * than names_cap bytes: nothing was consumed, sh_reader_names_needed() gives the size to retry with -- the reference has no limit on
 * name length, and unitig names run to tens of kilobases).  bits: max_variants*row_bytes;
 * names: concatenated variant names, name_off[v] .. name_off[v+1] (max_variants+1 offsets). */
int64_t     sh_reader_next(sh_reader *r, int64_t max_variants, uint8_t *bits, int64_t row_bytes, int32_t *counts,
                           char *names, int64_t names_cap, int64_t *name_off);
int64_t     sh_reader_names_needed(sh_reader *r);
/* bytes of inflated text currently buffered (bounded by one block of lines + one read slab; for tests) */
int64_t     sh_reader_buffered(sh_reader *r);
/* gzip members decoded on several threads (csrc/inflate_par.h): chunks accepted so far that started from a SEARCHED block head (0: the
 * stream was decoded by one thread -- small file, stored/fixed blocks only, SEERHIP_ROUTE reader=serial, BGZF, plain text). */
int64_t     sh_reader_par_chunks(sh_reader *r);

/* introspection: how many variants of the LAST batch went through the Firth kernel / its pinv slow path */
int sh_glm_info(sh_ctx *ctx, int64_t *firth_routed, int64_t *pinv_routed);

/* ---------------------------------------------------------------------------------------------
 * Result sink: array-backed results -> the reference's TSV rows (pyseer/utils.py:39-105 format_output, the print loop of
 * pyseer/__main__.py:805-827).  CPU only.  For every selected row v = sel[r], in order:
 *   name \t cols[0][v] \t ... \t cols[ncol-1][v] [\t betas[v*q .. v*q+q) if betas_valid[v]] [\t label | NA] \t notes \n
 * numbers as '%.2E', non-finite -> empty field; notes = names of flag bits 0..8 joined by ','.
 * names/name_off: concatenated variant names as sh_reader_next delivers them.  lineage may be NULL (no column).
 * Returns the number of bytes written, or -(needed+1) when cap is too small (nothing written), -1 on bad arguments.
 * Formatting runs on the calling thread and the idle workers of the process-wide host pool; any number of threads may call at once
 * (each owns its text buffers: no lock).
 * --------------------------------------------------------------------------------------------- */
int64_t sh_format_rows(const char *names, const int64_t *name_off, const int64_t *sel, int64_t nsel,
                       const double *const *cols, int ncol, const double *betas, int q, const uint8_t *betas_valid,
                       const int32_t *lineage, const char *const *lineage_labels, int n_labels,
                       const uint32_t *flags, char *out, int64_t cap);


/* ---------------------------------------------------------------------------------------------
 * Job stream (round 5): blocks of a variant stream in, the TSV text of the rows the run PRINTS and the run's counters out.
 * Replaces, per block, the reference's print loops pyseer/__main__.py:571-593 (fixed effects) and :805-827 (LMM) over the tuples of
 * fixed_effects_regression / fit_lmm, with format_output (pyseer/utils.py:39-105) on the rows that pass: rows that are not pre-filtered and
 * not filtered (or every row with print_filtered, LMM blocks then listing their filtered rows first and NaN-masked as pyseer/lmm.py:160-217
 * leaves them).  Selection, the counters and the compaction of the printed rows' statistics run on the device (csrc/job_kernels.hip); the host
 * formats only printed rows.  The stream is pipelined three blocks deep: sh_job_submit queues the upload of block k and the kernels of block
 * k-1 and returns; sh_job_collect waits (asleep between polls of the block's event) for the OLDEST uncollected block and returns its text, valid until the
 * calling thread's next sh_job_collect / sh_format_* call.  At most sh_job_depth(job) blocks may be submitted and not yet collected.
 *   bits/row_bytes/V: as sh_lmm_batch;  counts[v]: carriers of variant v (af = counts / n_samples, pyseer/input.py:446);
 *   names/name_off: concatenated variant names as sh_reader_next delivers them.  All four must stay valid until the block is collected.
 *   rows_are_dma != 0: `bits` lies in pinned or registered host memory (sh_host_register) and is read by the device where it lies;
 *   otherwise the rows are copied through a pinned slab by the calling thread and the process-wide host pool.
 *   counters[0..2] of sh_job_collect: pre-filtered, tested, printed variants of the block (the reference's stderr summary, __main__.py:595-599).
 * sh_job_set_lineage (round 6; before the first block): fit_lineage_effect (pyseer/model.py:151-199) inside the stream, on the design of
 *   sh_lineage_setup, for the rows the run PRINTS, its label (labels[index], or NA) in the reference's column (pyseer/utils.py:88-95).  Fixed
 *   effects: every printed row the reference reaches the fit with (model.py:379-382: not pre-filtered, no firth-fail).  LMM: ONE fit per block
 *   -- of the block's LAST variant -- given to every row that passed, which is what pyseer/lmm.py:209-213 computes (it calls the fit with the
 *   stale `k` of its loading loop), so blocks must be the reference's (--block_size); per_variant != 0 fits each passing row's own variant.
 * sh_job_set_patterns (--output-patterns): hash_pattern (pyseer/input.py:710-723) -- base64(md5(the presence vector as int64)) + "\n" -- of every
 *   TESTED variant of a block (not pre-filtered; print loops pyseer/__main__.py:584-585, 794-795, 818-819), computed on the device (one lane
 *   per variant; the md5 of 8 N bytes is 60 us of a CPU core per variant at N = 5000); sh_job_patterns returns the text of the block last
 *   collected (25 bytes per tested variant, in input order), valid until the next sh_job_collect.
 * sh_job_set_samples (--print-samples): the two sample lists of pyseer/utils.py:96-98 (carriers, then the others, comma-joined) in front of
 *   the notes of every PRINTED row, read from the row's bits on the host: names / name_off = the n_samples sample names (concatenated, n + 1
 *   offsets) in the context's sample order, order[i] = the sample printed i-th (the reference sorts the names).
 * --------------------------------------------------------------------------------------------- */
typedef struct sh_job sh_job;
sh_job *sh_job_open(sh_ctx *ctx, int lmm, int print_filtered);
void    sh_job_close(sh_job *job);
int     sh_job_submit(sh_job *job, const uint8_t *bits, int64_t row_bytes, int64_t V, const int32_t *counts,
                      const char *names, const int64_t *name_off, int rows_are_dma);
int     sh_job_collect(sh_job *job, const char **text, int64_t *nbytes, int64_t *counters);
int64_t sh_job_pending(sh_job *job);
int     sh_job_depth(sh_job *job);     /* blocks that may be submitted and not yet collected: 3 (LMM), 2 + lanes (fixed effects) */
int     sh_job_set_lineage(sh_job *job, const char *const *labels, int n_labels, int per_variant);
int     sh_job_set_patterns(sh_job *job, int on);
int     sh_job_patterns(sh_job *job, const char **text, int64_t *nbytes);
int     sh_job_set_samples(sh_job *job, const char *names, const int64_t *name_off, const int32_t *order, int n);
/* The whole block loop of one stream of a packed cache (pyseer_amd/input.py PackedCacheWriter: the `--save-packed` / `--load-packed` file) in
 * one call -- the loop over load_var_block / fit / print of pyseer/__main__.py:541-593, 777-827 for part `part_i` of `part_n` contiguous ranges
 * of the cache's rows (the reference's `--cpu N`; one part per device).  Stored blocks are merged to at least block_rows rows (never split)
 * exactly as the single stream would cut them; rows go to the device by DMA from registered windows of the file's mapping (use_dma != 0; a
 * block merged from several stored blocks range by range, each to its place) or through pinned slabs; the text of the printed rows is written to out_fd, the pattern text (sh_job_set_patterns) to pat_fd, in input order.
 * counters[0..3] += pre-filtered, tested, printed variants and blocks.  *stop != 0 (may be NULL) ends the stream at the next block.  The
 * calling thread holds no interpreter lock: streams of several contexts run side by side on threads of one process. */
int     sh_job_run_packed(sh_job *job, const char *path, int part_i, int part_n, int64_t block_rows, int use_dma, int out_fd, int pat_fd,
                          int64_t *counters, const volatile int *stop);
/* the formatter behind sh_job_collect, callable on its own (tests): nsel compacted records -- idx[r] = the variant's index into names / counts,
 * flags[r], cols[c][r] (c < ncol), slopes betas[j * betas_stride + r] printed where betas_valid[r] -- as
 *   name \t counts[idx]/n_samples \t cols... [\t betas...] [\t lineage label] \t notes \n ; lineage[r] = index into lineage_labels or -1 (NA), NULL =
 *   no lineage column; *text is owned by the calling thread (valid until its next call). */
int64_t sh_format_records(const char *names, const int64_t *name_off, const int32_t *counts, int n_samples, const int32_t *idx, int64_t nsel,
                          const double *const *cols, int ncol, const double *betas, int64_t betas_stride, int q, const uint8_t *betas_valid,
                          const int32_t *lineage, const char *const *lineage_labels, int n_labels, const uint32_t *flags, const char **text);
/* Make [p, p + nbytes) (e.g. a window of a read-only file mapping: the packed cache) readable by the device where it lies, so that the rows
 * cross PCIe by DMA with no CPU copy (hipHostRegister on the enclosing pages; measured 4 ns of CPU per 632-byte row against 21 for the
 * copy into pinned staging, profiles/r05/host_feed_probe.txt).  `device`: the device whose stream will read it (the registration itself is portable).
 * Unregister with the same p once the rows have been collected. */
int     sh_host_register(const void *p, int64_t nbytes, int device);
int     sh_host_unregister(const void *p);
/* The host CPU budget (csrc/host_pool.h): every host-side helper of this library -- readers, the stager of pageable rows, the formatter --
 * draws on ONE persistent pool sized to the CPUs the process may use (affinity mask cut by the cgroup quota), the counterpart of the
 * reference's `--cpu N`.  sh_set_host_threads overrides the detected number (0 = detect); sh_set_host_streams says how many device streams
 * (or readers) the caller runs at once, so per-stream helpers take their share (generalises sh_reader_set_concurrency).
 * sh_host_cpu_seconds: thread CPU seconds spent per host stage so far ("stage=seconds,..."); returns the length, or -(needed+1). */
int     sh_host_cpus(void);
void    sh_set_host_threads(int n);
void    sh_set_host_streams(int n);
int     sh_host_pool_workers(void);
int     sh_host_cpu_seconds(char *buf, int cap);
/* the largest number of threads that were inside sh_format_rows / sh_format_records at the same time (reset != 0 clears it): tests */
int     sh_format_concurrency_max(int reset);

#ifdef __cplusplus
}
#endif
#endif
