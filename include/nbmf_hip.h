/*
 * nbmf_hip.h — C ABI of libnbmf_hip.so: the MI355X (gfx950) implementation of the NBMF-MM
 * multiplicative-update inner loop.
 *
 * The reference (siddC/nbmf_mm) has no FFI of its own: its seam is the Python call
 *   nbmf_mm_solver(...)            src/nbmf_mm/_solver.py:61-216   (outer loop)
 *   nbmf_mm_update_beta_dir(...)   src/nbmf_mm/_solver.py:5-59     (one MM iteration)
 *   NBMFMM.transform(...)          src/nbmf_mm/_base.py:162-199    (simplex-factor-only loop)
 * Each entry point below names the reference lines it replaces.  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add to call this library from _solver.py.
 *
 * Conventions
 *   - every function returns 0 (NBMF_OK) or a negative NBMF_ERR_* code; nbmf_last_error() gives the
 *     message of the last failure on the calling thread;
 *   - host buffers are caller-owned, C-contiguous (row-major), float64 unless stated, and are copied
 *     during the call; device memory is owned by the context;
 *   - INTERNAL LAYOUT (that of _solver.py after its orientation transpose, :113-136): the data matrix
 *     Y is m x n, the simplex factor W is k x m (columns sum to 1), the Beta factor H is k x n;
 *   - a context is not thread-safe; distinct contexts are independent; one context = one GPU.
 *     Row-sharded multi-GPU runs use one process (and one context) per GPU, joined by
 *     nbmf_comm_init (RCCL all-reduce of the k x n H-step products each iteration).
 */
#ifndef NBMF_HIP_H
#define NBMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBMF_OK 0
#define NBMF_ERR_ARG (-1)       /* bad argument (shape, NULL, unsupported k) */
#define NBMF_ERR_HIP (-2)       /* a HIP runtime call failed, or no gfx950 device */
#define NBMF_ERR_RANGE (-3)     /* data outside [0,1]: the caller raises ValueError("X must be binary"), _base.py:90-91 */
#define NBMF_ERR_STATE (-4)     /* call order violated (e.g. run before upload) */
#define NBMF_ERR_COMM (-5)      /* RCCL failure */

#define NBMF_MASK_NONE 0
#define NBMF_MASK_F64 1         /* float64 weights; Y*mask semantics of _solver.py:30-32 */
#define NBMF_MASK_U8 2          /* bool / uint8, nonzero = observed */

#define NBMF_PROJ_NORMALIZE 0   /* the reference path: /n then column renormalise, _solver.py:54,57 */
#define NBMF_PROJ_DUCHI 1       /* README.md:27-35 extension: per-row observed count, Euclidean projection */

#define NBMF_FLAG_BINARY_PATH 1 /* out_flags bit: data was {0,1} (and mask {0,1}) -> 1 byte/entry storage */

#define NBMF_MAX_K 512        /* up to 128 components run in one fused sweep; more run as slices of 128 sharing a
                                 stored Theta (8 bytes per entry of extra device memory) */

typedef struct nbmf_ctx nbmf_ctx;

/* library / device ------------------------------------------------------------------------ */
int nbmf_abi_version(void);
/* First 12 hex digits of the SHA-256 over the sources this binary was compiled from (csrc/nbmf_hip.hip, the .inc files beside it sorted
 * by name, include/nbmf_hip.h), stamped by csrc/Makefile; "unstamped" for a build that did not go through it.  bench.py
 * prints it on its line and tools/refresh_profiles.sh refuses a profile whose library is not the tree's. */
const char* nbmf_source_hash(void);
const char* nbmf_last_error(void);
int nbmf_device_count(int* count);
/* The PCI bus id of HIP device `device` ("0000:75:00.0"): which PHYSICAL card an index means in this process.  A launcher may
 * narrow every rank's view to its own card, which is then device 0 everywhere; ranks compare these strings, not indices, to
 * tell whether they share a card (bench.py: RCCL refuses two ranks on one device).  No reference counterpart. */
int nbmf_device_bus_id(int device, char* buf, int len);

/* Create a context for an m x n internal problem with k components on HIP device `device`.
 * Replaces the implicit NumPy allocations of _solver.py:109-136. */
int nbmf_create(int64_t m, int64_t n, int k, int device, nbmf_ctx** out);
int nbmf_destroy(nbmf_ctx* ctx);

/* Hyper-parameters of nbmf_mm_solver (_solver.py:66-67,74) plus the projection extension.  Any positive
 * normal eps is accepted.  The fastest form of the binary path's sweeps needs 1e-12 <= eps < 2^-22 (the reference's
 * default, 1e-8, is in there) and factors inside the range a fit keeps; outside it the sweeps take the reference's
 * own selects, one reciprocal per entry and, below 1e-70, a log-likelihood that cannot underflow (one frexp per
 * entry instead of one per trip): same results, 10-20 % slower. */
int nbmf_set_hyper(nbmf_ctx* ctx, double alpha, double beta, double eps, int projection);

/* Upload the data matrix and optional mask (replaces the per-iteration Y*mask, Y.T*mask.T,
 * (1-Y).T*mask.T products of _solver.py:22-32 by a one-time pack into MFMA tile order).
 *   transposed = 0: x is the m x n internal matrix, leading dimension ldx (elements);
 *   transposed = 1: x is n x m (the user's V under orientation="dir-beta", _solver.py:113-120)
 *                   and Y = x^T; the mask has the same shape/orientation as x.
 * mask may be NULL (mask_kind NBMF_MASK_NONE).  out_flags (may be NULL) receives NBMF_FLAG_*.
 * Returns NBMF_ERR_RANGE if any entry of x is outside [0,1] or not finite. */
int nbmf_upload(nbmf_ctx* ctx, const double* x, int64_t ldx, int transposed,
                const void* mask, int mask_kind, int64_t ldmask, int* out_flags);

/* The same upload from a host array of the given element type: NBMF_DATA_F64 (nbmf_upload), NBMF_DATA_F32, or NBMF_DATA_U8 -- uint8 /
 * bool data, one byte per entry, values 0 and 1 (anything else is out of range: NBMF_ERR_RANGE, the "X must be
 * binary" of _base.py:90-91).  The reference converts every input to float64 first (check_array(dtype=float64),
 * _base.py:83; `Y * mask` at _solver.py:30 would do it anyway); a caller that holds its binary matrix as bool or
 * uint8 hands it over as it is -- BASELINE configs[4]'s 360k x 17k is 6.1 GB that way instead of 49 GB -- and the
 * pack kernel reads the bytes (nbmf_upload_v(ctx, ptr, dtype{f64,u8}, ld); 7 "Memory at config 5").
 * ldx in ELEMENTS.  The fit is bit for bit that of the float64 upload of the same values (tested). */
#define NBMF_DATA_F64 0
#define NBMF_DATA_U8 1
#define NBMF_DATA_F32 2   /* float32 data: converted on the device (exactly, as the reference's dtype=float64 conversion does): 4 bytes per entry over PCIe */
int nbmf_upload_v(nbmf_ctx* ctx, const void* x, int x_kind, int64_t ldx, int transposed,
                  const void* mask, int mask_kind, int64_t ldmask, int* out_flags);

/* Sparse upload of BINARY data: the user's matrix given as a CSR pattern (canonical: sorted or not, but no
 * duplicate entries; every stored entry means 1), optionally with a second CSR pattern of the OBSERVED entries
 * (mask_indptr NULL = everything observed).  Replaces `Y = Y.toarray()` of _solver.py:28-29 / _base.py:86-87
 * plus the pack: the dense matrix is never formed, on the host or on the device (1 byte per entry and image, as
 * for dense binary data).  indptr has rows+1 entries of the user's matrix (m, or n when transposed != 0). */
int nbmf_upload_csr(nbmf_ctx* ctx, const int64_t* indptr, const int32_t* indices, int64_t nnz, int transposed,
                    const int64_t* mask_indptr, const int32_t* mask_indices, int64_t mask_nnz, int* out_flags);

/* Measurement helper with no reference counterpart: fill the context with synthetic BINARY data generated on
 * the device instead of nbmf_upload -- entry (i, j) of the internal m x n matrix is 1 with probability
 * `density` and observed with probability `observed`, from a counter-based hash of (seed, i*n + j)
 * (splitmix64; nbmf_mm_amd/_hip.py:synthetic_reference regenerates it in NumPy).  For shapes whose float64
 * host array would be impractical (BASELINE configs[4]: 360k x 17k = 49 GB). */
int nbmf_generate(nbmf_ctx* ctx, uint64_t seed, double density, double observed);

/* Generate this context's SLICE of a larger synthetic matrix: entry (i, j) of the context is entry (row0 + i,
 * col0 + j) of a global matrix with n_global columns, i.e. the hash counter is (row0 + i) * n_global + (col0 + j).
 * With row0 = col0 = 0 and n_global = n this is nbmf_generate.  For multi-GPU runs at sizes no host array can
 * hold (BASELINE configs[3], 262144 x 8192 over 8 ranks): every rank generates the rows it owns of the SAME matrix
 * a single context would generate whole, so sharded and unsharded runs can be compared entry for entry. */
int nbmf_generate_slice(nbmf_ctx* ctx, uint64_t seed, double density, double observed, int64_t row0, int64_t col0,
                        int64_t n_global);

/* Storage path of the NEXT nbmf_upload.  NBMF_STORAGE_AUTO (default): the cheapest path the data allows -- byte
 * codes for binary data with a binary (or no) mask, doubles otherwise.  NBMF_STORAGE_F64: never the
 * byte codes, i.e. the arithmetic of the reference for real-valued V in [0, 1] (_base.py:90 accepts it; its tests
 * feed np.random.rand) even when the values happen to be binary.  NBMF_STORAGE_F64_WEIGHTS: doubles plus a tile of
 * float64 weights per data tile (the Y * mask of _solver.py:30-32 with a real-valued mask; all-ones without a mask).
 * Measurement and test hook with no reference counterpart: results agree with the automatic path to rounding. */
#define NBMF_STORAGE_AUTO 0
#define NBMF_STORAGE_F64 1
#define NBMF_STORAGE_F64_WEIGHTS 2
int nbmf_set_storage(nbmf_ctx* ctx, int storage);

/* Number of observed entries held by this context: Y.size or count_nonzero(mask), _solver.py:151,155. */
int nbmf_get_n_obs(nbmf_ctx* ctx, double* n_obs);

/* Factors in internal layout: W is k x m, H is k x n, C-contiguous (_solver.py:132-136; the caller
 * has already drawn / transposed / column-normalised them). */
int nbmf_set_factors(nbmf_ctx* ctx, const double* W_kxm, const double* H_kxn);
int nbmf_get_factors(nbmf_ctx* ctx, double* W_kxm, double* H_kxn);

/* The hot loop, _solver.py:143-175: up to max_iter MM iterations (H-step, W-step, loss) with the
 * relative-change stop rule (:169-174, checked on device).  losses must hold max_iter doubles;
 * *n_iter receives iteration+1 (:215).  Factors stay on the device (nbmf_get_factors). */
int nbmf_run(nbmf_ctx* ctx, int max_iter, double tol, double* losses, int* n_iter);

/* n_problems INDEPENDENT fits of the data this context holds, each with its own Beta prior (alpha[p], beta[p]) and its
 * own initial factors (W0: n_problems x k x m, H0: n_problems x k x n, internal layout as nbmf_set_factors), the
 * other hyper-parameters (eps, projection) as set: the loops of the reference's experiment driver -- the 36-point
 * (alpha, beta) grids and the K sweeps of examples/reproduce_magron2022.py:75-340 over train_nbmf_mm (:49-73) -- and
 * the n_init restarts of README.md:144.  Outputs per problem: losses (row p of an n_problems x max_iter array, the
 * first n_iter[p] entries), n_iter[p], final factors (W_out, H_out, shaped like the inputs).
 * Small problems (the reference's datasets: 50 x 85 ... 1226 x 285) run as many at a time as the chip holds, in ONE
 * persistent launch per group -- each fit keeps its CUs, its own barrier words and stops on its own; larger ones run
 * one after the other.  Either way problem p's results are bit for bit those of nbmf_set_hyper(alpha[p], beta[p]) +
 * nbmf_set_factors + nbmf_run + nbmf_get_factors.  The context's own factors are left as they were when the
 * single-launch path served the batch, and hold the last problem's otherwise. */
int nbmf_run_batch(nbmf_ctx* ctx, int n_problems, const double* alpha, const double* beta, const double* W0, const double* H0,
                   int max_iter, double tol, double* losses, int* n_iter, double* W_out, double* H_out);
/* Diagnostics: persistent launches nbmf_run_batch made on this context and the problems they served. */
int nbmf_batch_stats(nbmf_ctx* ctx, int* launches, int* problems);
/* Diagnostics: how the two sweeps of an iteration are cut (valid after an upload / generate): chunks of the H-pass and of
 * the W-pass and the row blocks (16 rows of the swept image) per chunk; a sweep launches (column strips / 4) x chunks
 * workgroups.  Any pointer may be NULL. */
int nbmf_sweep_info(nbmf_ctx* ctx, int* h_chunks, int* h_blocks_per_chunk, int* w_chunks, int* w_blocks_per_chunk);
/* Diagnostics, process-wide: sweeps launched in the two-state W variant (binary data observed everywhere, no pad rows in the
 * swept dimension: the bracket of _solver.py:53 with two states per entry instead of three) and in the ragged-K variant
 * (fewer components than the layout holds), and runs that RESUMED after a sweep could not assemble the previous iteration's
 * loss (_solver.py:148-162) within its bound -- a GPU shared with a tenant that saturates it -- and went on with the loss
 * in a launch of its own.  Tests use it to make sure the variant they mean to test is the one that ran.
 * Any pointer may be NULL. */
int nbmf_variant_stats(long long* full_w_launches, long long* ragged_k_launches, long long* loss_assembly_recoveries);

/* Progress reports while nbmf_run works (the `verbose` prints of _solver.py:165-166 need the losses as they
 * arrive, not after the run): with a callback set, nbmf_run synchronises after every `every` iterations and
 * hands over the losses that have become final since the last report -- iterations [first, first+count),
 * `losses` pointing at the first of them.  The loss of iteration t is settled by the sweep of iteration t+1
 * (SURVEY N3), so reports trail the device by one iteration; the values are those nbmf_run returns.
 * fn == NULL (the default) switches reporting off: no intermediate synchronisation. */
typedef void (*nbmf_progress_fn)(void* user, int first, int count, const double* losses);
int nbmf_set_progress(nbmf_ctx* ctx, nbmf_progress_fn fn, void* user, int every);

/* Diagnostics: how many nbmf_run calls of this context were served by the single-launch path for small problems
 * (one persistent kernel runs the whole loop of _solver.py:143-175), and how many of those gave
 * up at a grid barrier and were redone by the five-kernel path.  Environment: NBMF_PERSISTENT=0 switches the
 * single-launch path off. */
int nbmf_small_stats(nbmf_ctx* ctx, int* runs, int* aborted);

/* Diagnostics, process-wide: fits (nbmf_run calls and problems of nbmf_run_batch) the single-launch engine has served to
 * the end, persistent launches that gave up (their fits were redone by the launches), and nbmf_run calls the
 * launch-per-kernel engine has served.  The GPU tests that run every case once per engine read them to make sure the
 * engine they name is the one that ran (a persistent kernel that silently gave up would otherwise pass as the other).
 * Any pointer may be NULL. */
int nbmf_engine_stats(long long* persistent_served, long long* persistent_aborted, long long* launches_served);

/* n_steps repetitions of the simplex-factor update with the Beta factor frozen: the loop body of
 * NBMFMM.transform, _base.py:178-193 (always "normalize", eps as set by nbmf_set_hyper). */
int nbmf_w_only_steps(nbmf_ctx* ctx, int n_steps);

/* Loss of the current factors, _solver.py:148-162. */
int nbmf_loss(nbmf_ctx* ctx, double* loss);

/* Sum over entries of Ym*log(Theta+eps) + (1-Ym)*log(1-Theta+eps) for the current factors (summed over
 * ranks when a communicator is attached): the numerator of NBMFMM.score, _base.py:239-247 (divide by
 * nbmf_get_n_obs), and the data term of the loss, _solver.py:150-154.  clip_theta != 0 clips Theta to [0, 1]
 * first, as NBMFMM.inverse_transform does (_base.py:210) before score evaluates it. */
int nbmf_loglik(nbmf_ctx* ctx, int clip_theta, double* loglik);

/* Strictly masked variant: sum over OBSERVED entries only of mask*(Y log(Theta+eps) + (1-Y) log(1-Theta+eps)),
 * the held-out log-likelihood of examples/reproduce_magron2022.py:40-47 (compute_perplexity: divide by
 * nbmf_get_n_obs, negate, exponentiate).  A Theta-only sweep (no back-products). */
int nbmf_loglik_strict(nbmf_ctx* ctx, double* loglik);

/* Multi-GPU, one process (and one context) per GPU.  The internal matrix Y is split over the ranks along
 *   shard_axis 0: its ROWS    -> W[:, rows] is local, H is replicated; each iteration all-reduces the k x n
 *                 H-step products [P1 | P2 | loglik] (2*K*N+1 doubles) before the H-update;
 *   shard_axis 1: its COLUMNS -> H[:, cols] is local, W is replicated; each iteration all-reduces the k x m
 *                 W-step bracket (K*M doubles) before the W-update, and [loglik, prior A, prior B] (3 doubles)
 *                 before the stop test.
 * (beta-dir with V split by rows, or dir-beta with V split by columns, is axis 0; the other two
 * combinations are axis 1.)  Rank 0 calls nbmf_comm_unique_id and distributes the 128 bytes; every rank,
 * after nbmf_upload of its shard, calls nbmf_comm_init.  The global observed count (and, for axis 1, the
 * global column count and per-row observed counts) are reduced there.  No reference counterpart (the
 * reference is single-process). */
int nbmf_comm_unique_id(void* id128);
int nbmf_comm_init(nbmf_ctx* ctx, const void* id128, int nranks, int rank, int shard_axis);

/* Same sharded run with the all-reduce done by the caller on a host buffer (sum over ranks, in place,
 * return 0 on success).  Used by the tests (two ranks sharing one GPU, gloo) and as a fallback
 * transport; the device work and the control flow are identical to the RCCL path. */
typedef int (*nbmf_host_allreduce_fn)(void* user, double* buf, int64_t count);
int nbmf_comm_init_host(nbmf_ctx* ctx, nbmf_host_allreduce_fn fn, void* user, int nranks, int rank, int shard_axis);

/* Peer transport: the same exchanges done by the library's own kernels straight over xGMI (each rank maps
 * every other rank's exchange arena through HIP IPC).  Axis 0 fuses the H-update into a reduce-scatter: each
 * rank sums its 1/R column slice of [P1 | P2] in rank order, updates that slice of H and stores it into every
 * rank's arena, so K*N instead of 2*K*N doubles travel back and the update is not repeated R times; axis 1
 * all-reduces the W-step bracket in place (two-shot).  Sums are in rank order -> bitwise reproducible and
 * identical on all ranks.  Every wait is bounded (NBMF_PEER_TIMEOUT_MS, default 30000): a stalled job ends
 * with NBMF_ERR_COMM, never with a hung GPU.
 *   1. every rank: nbmf_peer_export(ctx, shard_axis, handle)   -> NBMF_PEER_HANDLE_BYTES opaque bytes
 *   2. all-gather the handle blocks in rank order (any host channel)
 *   3. every rank: nbmf_comm_init_peer(ctx, all_handles, nranks, rank, shard_axis)
 *      (maps the arenas, runs a known-answer exchange, reduces the global counts; on failure the context is
 *       left unattached so the caller can fall back to nbmf_comm_init)
 * One process per rank, or several ranks (contexts, one host thread each) in one process: a handle block also
 * names the exporting process and the arena's address there, and a rank of the same process uses the memory
 * directly (HIP IPC does not map a handle inside the process that exported it; across devices of one process peer
 * access is enabled instead).  At most 16 ranks.  No reference counterpart. */
#define NBMF_PEER_HANDLE_BYTES 192
int nbmf_peer_export(nbmf_ctx* ctx, int shard_axis, void* handle);
int nbmf_comm_init_peer(nbmf_ctx* ctx, const void* handles, int nranks, int rank, int shard_axis);

/* Row split (shard_axis 0) only: cut the per-iteration exchange into `panels` column panels (1 or 2).  With 2, the
 * second panel's exchange travels on a side stream while the first panel's H-update is applied and the W-pass
 * starts on the chunks that read only those columns.  0 = the default (1 panel, or 2 if the
 * environment says NBMF_OVERLAP=1).  Takes effect at the next nbmf_comm_init*; every rank must choose the same. */
int nbmf_set_exchange_panels(nbmf_ctx* ctx, int panels);

/* Bound on every wait of the peer transport attached NEXT (milliseconds; 0 = the default: NBMF_PEER_TIMEOUT_MS from
 * the environment, else 30000).  A caller that is only probing a transport (bench.py times each candidate over a few
 * iterations before the run) sets a short bound for the probe and the default for the run, so that a transport which
 * cannot work on this machine costs seconds, not a timeout per exchange. */
int nbmf_set_peer_timeout_ms(nbmf_ctx* ctx, double ms);

/* Cancel from another thread (the ONLY call that may be made on a context while its owner thread is inside one): sticky;
 * the owner's nbmf_run returns NBMF_ERR_STATE at its next iteration boundary, and what it has already enqueued is cut
 * short on the device by the run's own stop flag (the word the kernels honour after convergence, _solver.py:169-175).
 * The reference's loop is interrupted by KeyboardInterrupt between NumPy calls; this is what a host thread that joins
 * rank threads (NBMF(n_gpus=N)) uses for the same purpose.  The context can only be destroyed afterwards. */
int nbmf_cancel(nbmf_ctx* ctx);

/* Drop the attached communicator (RCCL, host or peer): the context is a single-GPU context over its own
 * shard again and another nbmf_comm_init* may follow.  Collective in effect: every rank must do the same. */
int nbmf_comm_detach(nbmf_ctx* ctx);

/* What the attached transport itself reports (no reference counterpart: diagnostics of the sharded loop that replaces
 * _solver.py:143-175 on several GPUs).  kind: 0 none, 1 RCCL, 2 peer, 3 host.  nranks_seen: RCCL -- ncclCommCount of the
 * communicator (-1 if this librccl has no such symbol); peer -- arenas mapped, this rank's included; host -- the nranks
 * it was given.  remote: RCCL -- ncclCommCuDevice (the device RCCL bound the communicator to, -1 if unknown); peer / host
 * -- how many of the other ranks' buffers are reached from outside this context.  Any pointer may be NULL. */
int nbmf_comm_info(nbmf_ctx* ctx, int* kind, int* nranks_seen, int* remote);

/* Measurement: HIP-event timing of the two fused pass kernels on the context's stream.  enable: 0 = off, 1 = every
 * sweep, n > 1 = the sweeps of every n-th iteration of a run (a timed dispatch costs ~2.5 us: at sub-millisecond
 * iterations a sample of the launches keeps the measurement from slowing what it measures). */
int nbmf_timing_enable(nbmf_ctx* ctx, int enable);
/* ms summed over launches since enable, and launch counts; any pointer may be NULL. */
int nbmf_timing_get(nbmf_ctx* ctx, double* hpass_ms, int* hpass_launches, double* wpass_ms, int* wpass_launches);
int nbmf_synchronize(nbmf_ctx* ctx);
/* hipDeviceSynchronize() on `device`: everything queued on that GPU by this process has finished (bench.py
 * brackets its timed region with it). */
int nbmf_device_synchronize(int device);

/* Self-test hook used by the GPU tests: applies one of the pass kernel's scalar device routines to n
 * caller-supplied values (op 0: Newton reciprocal used on the binary path; op 1: the natural logarithm
 * of the general path; op 2: the general path's quotient, evaluated as (1 - 0.75 x) / x; ops 3, 4: the two quotients
 * of an entry from one shared reciprocal, (1 - 0.75 x) / x and 0.75 x / ((1 - (x - 1e-8)) + 1e-8); op 5: 1/x by the
 * binary path's shared reciprocal of four, groups of four consecutive values; op 6: op 1 with the product sweeps' table; op 7: the table-free logarithm of the sweeps' running products) so the host can compare with IEEE 1/x,
 * log(x) and the IEEE quotients (accuracy contracts: tests/test_gpu_parity.py). */
#define NBMF_SELFTEST_RCP 0
#define NBMF_SELFTEST_LOG 1
#define NBMF_SELFTEST_DIV 2
#define NBMF_SELFTEST_DIV_PAIR_A 3
#define NBMF_SELFTEST_DIV_PAIR_B 4
#define NBMF_SELFTEST_RCP_OF_FOUR 5
int nbmf_selftest_unary(int device, int op, int n, const double* x, double* y);

#define NBMF_SELFTEST_LOG_1024 6  /* ... with the 1024-entry table of the product sweeps (series to r^4/4) */
#define NBMF_SELFTEST_LOG_PRODUCT 7  /* log of a sweep's running product: no table, no memory access */

/* Measurement helper with no reference counterpart: the f64 matrix rate of `device` as THIS machine delivers it -- a
 * loop of nothing but v_mfma_f64_16x16x4_f64 (four independent accumulators in VGPRs, two waves on every SIMD), one
 * launch of about target_ms milliseconds after a warm-up.  bench.py prints it beside the datasheet figure
 * (roofline.peak_measured; SURVEY 8d asks for the measured peak of the box that ran the bench).  tflops: 2048 flop per
 * MFMA and wave / time; cycles_per_mfma: SIMD cycles per MFMA at the nominal 2.4 GHz (64 = the datasheet's 78.6 TFLOP/s
 * on 1024 SIMDs); launch_ms: the timed launch.  Any pointer may be NULL. */
int nbmf_selftest_mfma_peak(int device, double target_ms, double* tflops, double* cycles_per_mfma, double* launch_ms);

#ifdef __cplusplus
}
#endif
#endif /* NBMF_HIP_H */
