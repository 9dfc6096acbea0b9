/* ptdeco_hip.h -- C ABI of libptdeco_hip.so (gfx950 / MI355X).
 *
 * The reference (TCLResearchEurope/ptdeco v0.5.9) has no FFI layer: its hot
 * path is a sequence of ATen calls inside src/ptdeco/{dwain,falor}/decomposition.py.
 * Each entry point below replaces one group of those call sites; the citation
 * after each declaration names the reference lines it stands in for
 * (dwain.py = src/ptdeco/dwain/decomposition.py, falor.py = src/ptdeco/falor/decomposition.py,
 * losses.py = src/ptdeco/utils/losses_primitives.py).
 *
 * Conventions
 *  - plain C types only; every pointer except `stream` is a DEVICE pointer
 *  - matrices are row-major with an explicit leading dimension in ELEMENTS
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on it
 *    unless stated otherwise, and never allocates or frees device memory: the
 *    caller passes workspaces sized by the matching *_workspace_bytes query
 *  - return value: 0 = ok, negative = ptd_status; ptd_last_error() gives the
 *    message of the last failing call on the calling thread
 *  - re-entrant across streams, devices and host threads.  Mutable state is kept per DEVICE and is
 *    advisory only (it selects between equivalent kernels, never results): lazily loaded code objects,
 *    the device facts queried once (CU count, architecture, occupancy of the whole-chip kernels), the
 *    ptd_set_concurrent_chains hint, the count of eigendecompositions this library has in flight on the
 *    device, and a back-off counter after a whole-chip kernel timed out
 */
#ifndef PTDECO_HIP_H
#define PTDECO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTD_ABI_VERSION 4   /* 2: ptd_nsr workspaces are initialised once (ptd_nsr_workspace_init); ptd_chol_inverse
                             * 3: ptd_stream_pair_wall_us, ptd_streams_wall_us, ptd_syrk_accumulate_multi
                             * 4: ptd_eigh_topk_batched, ptd_eigh_factored_prepare / _finish; ptd_band_reduce and the
                             *    two-stage reduction behind it are gone (2.2 x behind the default for three rounds) */

typedef enum { PTD_F32 = 0, PTD_F64 = 1, PTD_BF16 = 2 } ptd_dtype;

typedef enum {
  PTD_OK = 0,
  PTD_ERR_INVALID = -1,     /* bad argument (shape, stride, dtype, alignment, null) */
  PTD_ERR_UNSUPPORTED = -2, /* dtype / layout combination not implemented */
  PTD_ERR_WORKSPACE = -3,   /* workspace too small */
  PTD_ERR_LAUNCH = -4,      /* HIP runtime reported an error */
  PTD_ERR_NOCONV = -5       /* eigensolver did not converge in max sweeps */
} ptd_status;

int ptd_version(void);
const char* ptd_last_error(void);

/* Hint: the caller is about to run `chains` independent eigendecompositions at once, each on its own stream
 * (the reference has no counterpart: torch.linalg.eigh calls are serial, dwain.py:155-163).  With chains > 1 the
 * solver avoids kernels that claim a whole XCD or the whole chip for milliseconds (the resident kernels of the
 * tridiagonalisation), which shorten one chain and stall the others.  Applies to the CURRENT device (hipGetDevice of the
 * calling thread), default 1; returns the previous value.  Without the hint the library still notices its own
 * overlapping calls on a device (they take the blocked path), and the whole-chip kernels only run at all on an
 * unpartitioned 256-CU gfx950 whose occupancy query admits them; every inter-workgroup wait in them is bounded by a
 * 10 ms wall-clock time-out after which the reduction is repeated on the blocked path. */
int ptd_set_concurrent_chains(int chains);

/* Host query (round 5, synchronous): do two streams sit on DIFFERENT hardware queues?  The ROCm runtime maps the
 * streams of a process onto 4 hardware queues per priority level; two chains of dependent launches whose streams
 * share one are executed one packet after the other (measured: the seven eigendecompositions of a Llama block on four
 * streams take 187 ms when the four queues are distinct and 226-300 ms when two chains share one).  Launches one
 * single-wave kernel that holds its queue for `spin_us` microseconds on each stream and returns the wall time of the
 * pair in *wall_us: about spin_us when they ran side by side, about 2 x spin_us when they were serialised.  Both
 * streams are synchronised before and after.  No reference counterpart (torch.linalg.eigh calls are serial,
 * dwain.py:155-163); used by ptdeco_amd._engine.chain_streams to pick the streams of concurrent eigendecompositions. */
int ptd_stream_pair_wall_us(void* stream_a, void* stream_b, int spin_us, double* wall_us);
/* The same for `count` streams at once (a HOST array of hipStream_t): one kernel of `spin_us` on each; about spin_us when
 * all of them ran side by side, a multiple when some were serialised.  run_concurrently re-checks its streams with it at
 * every call (0.2 ms): which streams share a hardware queue was seen to CHANGE within a process (bench.py: four streams
 * verified distinct early on, two of them serialised a minute later). */
int ptd_streams_wall_us(void* const* streams, int count, int spin_us, double* wall_us);

/* A HIP stream with a hardware queue OF ITS OWN (ABI 4): hipExtStreamCreateWithCUMask over CUs [cu_first, cu_first +
 * cu_count) of the current device, cu_count = 0: every CU.  The runtime multiplexes ordinary streams onto four hardware
 * queues per priority (see above) and which streams share one cannot be told without measuring; a stream created with a
 * CU mask gets a queue that no other stream uses.  ptdeco_amd._engine runs the lanes of a precompute pass -- independent
 * chains of dependent launches, dwain.py:580-633's eigendecompositions -- on such streams, so that their overlap does not
 * depend on the creation order of every stream in the process.  *stream_out is a hipStream_t; the caller destroys it
 * with ptd_stream_destroy (or hipStreamDestroy).  No reference counterpart. */
int ptd_stream_create_dedicated(int cu_first, int cu_count, void** stream_out);
int ptd_stream_destroy(void* stream);

/* ---- covariance accumulation ------------------------------------------- */

/* E[i][j] += scale * sum_t Y[t][i] * Y[t][j]  for i >= j  (LOWER triangle only;
 * ptd_cov_finalize mirrors it).  Y is [T, n] (f32 or bf16), E is [n, n] (f64 or f32).
 * The product is accumulated in f32 on the matrix cores and promoted on the add.
 * Replaces `Eyyt += einsum("bp,bq->pq", y, y) / T`: dwain.py:147-152, falor.py:160. */
int ptd_syrk_accumulate(const void* y, int64_t T, int64_t n, int64_t ldy, int y_dtype,
                        void* E, int64_t ldE, int E_dtype, double scale, void* stream);

/* The same sum over `steps` calibration steps in ONE pass over E (round 5, ABI 3):
 *   E[i][j] += sum_s ( scale * sum_t Y_s[t][i] * Y_s[t][j] ),  i >= j,
 * `ys` a HOST array of `steps` device pointers to [T, n] matrices of one dtype and row pitch.  dwain.py:147-152 is
 * called once per calibration step (D times per layer) and each call reads and writes the live triangle of the f64
 * accumulator: 8 n (n + 1) bytes against 2 T n of bf16 activations -- at n = 4096, T = 2048 that traffic, not the
 * matrix cores, bounds the call.  Here a tile's f64 sum stays in registers across the steps (each step's f32 product
 * is promoted at its end, in step order) and E is read and written once.  bf16: one launch per 8 steps; f32 (bound by
 * the matrix cores): step by step, identical to `steps` calls of ptd_syrk_accumulate. */
int ptd_syrk_accumulate_multi(const void* const* ys, int steps, int64_t T, int64_t n, int64_t ldy, int y_dtype,
                              void* E, int64_t ldE, int E_dtype, double scale, void* stream);

/* ey[j] += scale * sum_t Y[t][j].   Replaces `Ey += y.mean(dim=0)`: falor.py:161. */
int ptd_colsum_accumulate(const void* y, int64_t T, int64_t n, int64_t ldy, int y_dtype,
                          void* ey, int ey_dtype, double scale, void* stream);

/* C = sym(E) / steps - (ey ? (ey / steps)(ey / steps)^T : 0);
 * C[i][i] += damp_factor * mean(diag(C)).   C is always f64 [n, n], full symmetric.
 * sym(E) reads the lower triangle of E; ey may be NULL, else it has E's dtype.
 * ws needs ptd_cov_finalize_workspace_bytes(n).
 * Replaces dwain.py:158-160, 207, 242 and falor.py:192-205. */
size_t ptd_cov_finalize_workspace_bytes(int64_t n);
int ptd_cov_finalize(const void* E, int64_t ldE, int E_dtype, const void* ey, int ey_dtype,
                     int64_t n, double steps, double damp_factor, double* C, int64_t ldC,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- symmetric eigendecomposition ---------------------------------------- */

/* All eigenpairs of the symmetric positive semi-definite f64 matrix A [n, n]
 * (full storage), eigenvalues ascending in evals[n], eigenvectors in the COLUMNS
 * of evecs [n, n] (row-major, ld ldv) -- the layout torch.linalg.eigh returns.
 * A is not modified.  Two solvers (env PTD_EIGH_METHOD = auto | tridiag | jacobi, default auto):
 *   tridiagonal route (n >= 256 in auto): blocked Householder reduction whose per-column
 *     SYMV streams the trailing matrix from HBM / Infinity Cache, eigenvalues by multisection,
 *     eigenvectors by inverse iteration, compact-WY back-transformation on the f64 matrix
 *     cores.  Requested eigenvalues closer than 1e-10 |A|, or chains of more than 48 closer than 1e-7 |A| (a
 *     dominant outlier above a dense bulk), get all computed vectors orthonormalised by one Cholesky-QR pass; a
 *     cluster at the BOTTOM of the spectrum that reaches into the request (a rank-deficient covariance on its
 *     damping floor) is completed with an orthonormal basis of the complement of the vectors above it, each of
 *     which is an eigenvector of the cluster.  Refused -- and handed to the solver below -- only when requested
 *     eigenvalues elsewhere are closer than 1e-13 |A| (PTD_EIGH_LONG_CHAINS=0, PTD_EIGH_NULL_COMPLETION=0:
 *     the earlier, stricter rule);
 *   one-sided block Jacobi on the f64 matrix cores: any PSD matrix.
 * NOT fully asynchronous: synchronises `stream` to read small decisions back (cluster
 * check, Jacobi convergence flag).  sweeps_out (host pointer, may be NULL) receives the number
 * of Jacobi sweeps (0 for the tridiagonal route).
 * Replaces `torch.linalg.eigh`: dwain.py:162, falor.py:207. */
size_t ptd_eigh_workspace_bytes(int64_t n);
int ptd_eigh(const double* A, int64_t lda, int64_t n, double* evals, double* evecs, int64_t ldv,
             void* ws, size_t ws_bytes, int* sweeps_out, void* stream);

/* Same, but only the eigenvectors of the k LARGEST eigenvalues are formed: evecs is [n, k]
 * (ld ldv >= k), column c holds the eigenvector of evals[n - k + c].  evals always has n entries:
 * with all_values != 0 every eigenvalue is computed; with all_values == 0 the tridiagonal route
 * computes only evals[n - k - 1 .. n) and fills the rest with NaN (the Jacobi route always
 * returns all of them) -- the drivers never read eigenvalues.  dwain never looks above the largest
 * candidate rank it evaluates (dwain.py:407-421, 424-426), so most of the inverse iterations and of
 * the back-transformation are skipped, and a dense cluster at the low end of the spectrum no
 * longer forces the Jacobi fallback.
 * With all_values == 0, n >= 2048 (a multiple of 128) and 7 k <= 2 n the eigenpairs come from a
 * Chebyshev-filtered subspace iteration on the f64 matrix cores (only evals[n - k .. n) are set then,
 * NaN below; every returned pair has passed a residual check |A v - lambda v| <= 1e-10 |lambda_max|,
 * eigenvector signs: largest entry positive); where that route does not apply -- a flat spectrum, a
 * breakdown -- the direct reduction answers with the same contract.  A is read only. */
int ptd_eigh_topk(const double* A, int64_t lda, int64_t n, int64_t k, int all_values, double* evals,
                  double* evecs, int64_t ldv, void* ws, size_t ws_bytes, int* sweeps_out, void* stream);

/* The f32 face of ptd_eigh_topk: A [n, n] f32 (full storage, symmetric), evals[n] and evecs [n, k] f32, same conventions.
 * The arithmetic is f64 on a converted copy (the routes above), the results are rounded to f32 -- at least as accurate as
 * the f32 `torch.linalg.eigh` the reference runs with decompose_in_float64=False (dwain.py:224-233, 162; falor's
 * use_float64=False, falor.py:181-186, 207). */
size_t ptd_eigh_f32_workspace_bytes(int64_t n, int64_t k);
int ptd_eigh_topk_f32(const float* A, int64_t lda, int64_t n, int64_t k, int all_values, float* evals, float* evecs,
                      int64_t ldv, void* ws, size_t ws_bytes, int* sweeps_out, void* stream);

/* ptd_eigh_topk for `count` matrices of ONE order and ONE k in a single call (ABI 4): As / evals / evecs are HOST arrays
 * of `count` device pointers, every matrix [n, n] with leading dimension lda, every evecs [n, k] with ldv; results and
 * contract per matrix exactly as ptd_eigh_topk.  Replaces the LOOP of `torch.linalg.eigh` calls of dwain's precompute
 * pass (dwain.py:580-633 calls get_eigenvectors -> :162 once per layer of a split): the direct reduction is a chain
 * of ~2 n dependent launches per matrix, so the matrices advance column by column in lockstep and every launch
 * serves all of them (blockIdx.y = matrix) -- one host thread, one stream, one host synchronisation for the batch,
 * where round 5 ran one thread and one stream per layer.  Requests the filtered route serves (see above) and single
 * matrices are solved one after the other exactly as ptd_eigh_topk would; two matrices are batched from n = 512 on
 * (PTD_EIGH_BATCH_MIN_N), three or more always; PTD_EIGH_BATCHED=0: never.  `all_values` is a set of flags: bit 0 = every
 * eigenvalue is wanted (as in ptd_eigh_topk), bit 1 (PTD_EIGH_FLAG_DIRECT) = take the direct reduction also where the
 * filtered route would serve the request, so that the matrices are batched: for a caller whose pass already runs
 * latency-bound reductions of this order on other streams, beside which the filter's f64 products only share the matrix
 * cores (two Llama blocks: q / o as two filtered problems each 75-127 ms in the pass, as one more batch 40).  stats (HOST pointer, may be NULL): the
 * ptd_eigh_profiled figures of the batch (method 1: `sweeps` = matrices per launch, work[0] = algorithmic bytes of
 * all of them) or of the last matrix when solved one by one.  Workspace: ptd_eigh_batched_workspace_bytes. */
#define PTD_EIGH_FLAG_ALL_VALUES 1
#define PTD_EIGH_FLAG_DIRECT 2
size_t ptd_eigh_batched_workspace_bytes(int64_t n, int64_t k, int count);
struct ptd_eigh_stats_s;
int ptd_eigh_topk_batched(const double* const* As, int64_t lda, int count, int64_t n, int64_t k, int all_values,
                          double* const* evals, double* const* evecs, int64_t ldv, void* ws, size_t ws_bytes,
                          struct ptd_eigh_stats_s* stats, void* stream);

/* The filtered route remembers LATE declines (a breakdown after its products were spent) per calling thread, device and
 * shape, and sends the next requests of that shape straight to the direct route (1, 2, 4, ... of them after the second
 * decline in a row).  This call clears the calling thread's memory: the drivers issue it at the start of every
 * decompose_in_place, so that the route a layer takes depends on that call's own sequence of requests only (two runs in
 * one process produce identical results).  No reference counterpart. */
void ptd_eigh_forget_declines(void);

/* The solver ptd_eigh_topk would try FIRST for this request: 3 = filtered subspace iteration (chip-filling f64
 * products: concurrent chains gain nothing), 1 = direct tridiagonal reduction (a latency-bound chain of short
 * launches: independent matrices overlap well on separate streams), 0 = Jacobi.  A host-side query; the filtered
 * route may still decline at run time (flat spectrum) and hand over to the direct one. */
int ptd_eigh_route(int64_t n, int64_t k, int all_values);

/* Top-k eigenpairs of C = W Ex W^T without forming C, for a layer that widens its input
 * (n_o > n_i; Llama gate / up: 4096 -> 14336): W [n_o, n_i] (f32, bf16 or f64, ld ldw),
 * Ex [n_i, n_i] f64 full symmetric = sum over steps of x^T x / T / steps (the INPUT second
 * moment; C is then exactly the feature covariance the reference accumulates, dwain.py:147-152,
 * since y = x W^T).  U [n_o, k] f64 gets the eigenvectors of the k largest eigenvalues
 * (ascending, same convention as ptd_eigh_topk), evals_k[k] (may be NULL) the eigenvalues of C.
 * G = W^T W = L L^T, B = L^T Ex L, B s = lambda s, u = W L^-T s: a n_i^2 eigenproblem plus
 * five f64 MFMA products.  Returns PTD_ERR_UNSUPPORTED if W^T W is not numerically positive
 * definite (then accumulate Y^T Y and call ptd_eigh_topk). */
size_t ptd_eigh_factored_workspace_bytes(int64_t n_o, int64_t n_i, int64_t k);
int ptd_eigh_factored(const void* W, int64_t ldw, int w_dtype, int64_t n_o, int64_t n_i,
                      const double* Ex, int64_t ldx, int64_t k, double* evals_k, double* U, int64_t ldu,
                      void* ws, size_t ws_bytes, void* stream);

/* ptd_eigh_factored in two halves around an eigendecomposition the CALLER runs (ABI 4), so that the inner n_i-sized
 * problems of several layers (Llama gate, up) and the same-sized covariances of others (down) share one
 * ptd_eigh_topk_batched call.  prepare: G = W^T W = L L^T, B = L^T Ex L; *B_out (HOST pointers to results) receives the
 * device address of B [np, np] (f64, full symmetric, ld np) inside the workspace and *np_out = np = n_i rounded up to
 * 64 (the padding carries exact, tiny eigenvalues that never reach the top k).  Synchronises the stream once (the
 * Cholesky status); PTD_ERR_UNSUPPORTED as ptd_eigh_factored.  finish: evals [np] ascending and S [np, k] (ld lds)
 * = eigenvectors of B's k largest eigenvalues -> U [n_o, k] = W L^-T S, evals_k (may be NULL).  The SAME workspace
 * (ptd_eigh_factored_workspace_bytes(n_o, n_i, k)), untouched in between.  Same reference lines as ptd_eigh_factored
 * (dwain.py:147-163). */
int ptd_eigh_factored_prepare(const void* W, int64_t ldw, int w_dtype, int64_t n_o, int64_t n_i, const double* Ex,
                              int64_t ldx, int64_t k, void* ws, size_t ws_bytes, double** B_out, int64_t* np_out,
                              void* stream);
int ptd_eigh_factored_finish(int64_t n_o, int64_t n_i, int64_t k, const double* evals, const double* S, int64_t lds,
                             double* evals_k, double* U, int64_t ldu, void* ws, size_t ws_bytes, void* stream);

/* Same, with per-phase device timing (HIP events on `stream` around the launches of each
 * phase; a few percent slower, for bench.py's roofline lines).  `stats` is a HOST pointer.
 *   method 0 (Jacobi):       phase 0 jac_gram, 1 jac_inner, 2 jac_update            (work = f64 flops executed)
 *   method 1 (tridiagonal):  phase 0 the per-column SYMV launches (work = algorithmic BYTES: the full
 *                            trailing square a one-stage SYMV streams, 8/3 n^3 in total; ms from
 *                            dispatch-attached events on every 8th launch, scaled), 1 the rest of the
 *                            reduction (alpha kernels, rank-2k updates, launch gaps), 2 work only
 *                            (flops of the rank-2k updates), 3 eigenvalues + inverse iteration +
 *                            back-transformation
 *   method 3 (filtered subspace iteration, the route of ptd_eigh_topk with all_values = 0, n >= 2048, 7 k <= 2 n):
 *                            ms = {Lanczos bounds, filter rounds (products with C + Cholesky-QR passes), the
 *                            Rayleigh-Ritz eigenproblem of order launches[2], Ritz products + residual check};
 *                            launches = {Lanczos steps, products with C, subspace dimension m, 0};
 *                            work[1] = flop of the products with C (2 n^2 m each) */
typedef struct ptd_eigh_stats_s {
  int method;
  int sweeps;        /* Jacobi sweeps; 0 for the tridiagonal route (ptd_eigh_topk_batched: matrices per launch) */
  int launches[4];
  float ms[4];       /* summed device time of the phase's launches */
  double work[4];
  float total_ms;    /* first launch to last launch of the call */
} ptd_eigh_stats;
int ptd_eigh_profiled(const double* A, int64_t lda, int64_t n, int64_t k, int all_values, double* evals,
                      double* evecs, int64_t ldv, void* ws, size_t ws_bytes, ptd_eigh_stats* stats, void* stream);

/* Diagnostic: Householder tridiagonalisation T = Q^T A Q of a symmetric f64 matrix (full
 * storage) and the eigenvalues of T by bisection.  d[n], e[n] (e[n-1] unused), evals[n]
 * ascending; any of the three may be NULL. */
size_t ptd_tridiagonalize_workspace_bytes(int64_t n);
int ptd_tridiagonalize(const double* A, int64_t lda, int64_t n, double* d, double* e, double* evals,
                       void* ws, size_t ws_bytes, void* stream);

/* Diagnostic: the Cholesky sweep of the filtered eigensolver's orthonormalisation passes on its own (the
 * step that replaces nothing in the reference -- torch.linalg.eigh hides it -- but is the latency-critical
 * part of ptd_eigh_topk's filtered route).  G [m, m] row-major f64, symmetric positive definite, lower
 * 64 x 64 tiles read, DESTROYED; Wt [m, m] receives L^-T (upper triangular, G = L L^T), so that X Wt has
 * orthonormal columns when G = X^T X.  m a multiple of 64 in [64, 8192].  Synchronises the stream once;
 * PTD_ERR_UNSUPPORTED when a pivot is not positive. */
size_t ptd_chol_inverse_workspace_bytes(int64_t m);
int ptd_chol_inverse(double* G, int64_t m, double* Wt, void* ws, size_t ws_bytes, void* stream);

/* ---- dense products (layer output, factor construction) ----------------- */

/* C[M,N] = alpha * sum_k A(m,k) * B(k,n) (+ bias[n]),  f32 or bf16 operands,
 * f32 accumulation on the matrix cores.  Operands are addressed with explicit
 * element strides: A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; exactly
 * one stride of each operand must be 1.  C is row-major [M, N] with ld ldc, of
 * dtype c_dtype (f32, or bf16 when the inputs are bf16; f64 x f64 -> f64 without bias
 * is also available, it is what the eigensolver uses internally).
 * Replaces `x @ weight.T` (dwain.py:194, 239; falor.py:159), `orig_weight.T @ uk`
 * and `(U @ V).T` (dwain.py:427-429, 511; falor.py:347-348). */
int ptd_gemm(const void* A, int64_t sam, int64_t sak, const void* B, int64_t sbk, int64_t sbn,
             void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int ab_dtype, int c_dtype,
             double alpha, const void* bias, void* stream);

/* The same product with a workspace (ABI 4).  A product with few output tiles -- T = 1576 rows of a ViT batch
 * against a 768-column layer are 78 tiles of 128 x 128 for 256 CUs -- has its K range split over workgroups; the
 * partial tiles go to f32 slabs in the workspace and a second launch adds them in index order (results do not
 * depend on scheduling).  ptd_gemm_workspace_bytes returns 0 where no split would be made; ws may be NULL
 * (= ptd_gemm). */
size_t ptd_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int ab_dtype, int c_dtype);
int ptd_gemm_ws(const void* A, int64_t sam, int64_t sak, const void* B, int64_t sbk, int64_t sbn,
                void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int ab_dtype, int c_dtype,
                double alpha, const void* bias, void* ws, size_t ws_bytes, void* stream);

/* y[T,n_o] = (x[T,n_i] @ A[r,n_i]^T) @ B[n_o,r]^T (+ bias[n_o]).  The workspace holds the
 * [T, r] intermediate of the operand dtype and, for f32 operands with a small rank, the partial
 * tiles of the first product's K split (added in a fixed order: results do not depend on
 * scheduling).  The decomposed layer's forward: dwain.py:74-85 / falor.py:84-95 (two nn.Linear),
 * dwain.py:126-144 (two 1x1 convs, x viewed as [B*H*W, C]). */
size_t ptd_lowrank_forward_workspace_bytes(int64_t T, int64_t n_i, int64_t r, int dtype);
int ptd_lowrank_forward(const void* x, int64_t ldx, int64_t T, int64_t n_i, const void* A, int64_t lda,
                        int64_t r, const void* B, int64_t ldb, int64_t n_o, const void* bias, void* y,
                        int64_t ldy, void* ws, size_t ws_bytes, int dtype, void* stream);

/* The same pair for a 1x1-convolution input in NCHW layout, without the NHWC copy the reference makes
 * (`permute(0,2,3,1).reshape(-1,C)`, dwain.py:116; falor.py:126): per image b, x_b = x + b*n_i*hw is an
 * [n_i, hw] matrix with the hw = H*W pixels contiguous; h_b[r,hw] = A x_b, y_b[n_o,hw] = B h_b + bias[:,None],
 * y written straight into NCHW (contiguous [batch, n_o, hw]).  The workspace holds h.  Replaces the two
 * nn.Conv2d(kernel_size=1) of dwain.py:126-144 / falor.py:136-153. */
size_t ptd_lowrank_forward_nchw_workspace_bytes(int64_t batch, int64_t hw, int64_t r, int dtype);
int ptd_lowrank_forward_nchw(const void* x, int64_t batch, int64_t n_i, int64_t hw, const void* A, int64_t lda,
                             int64_t r, const void* B, int64_t ldb, int64_t n_o, const void* bias, void* y,
                             void* ws, size_t ws_bytes, int dtype, void* stream);

/* ---- rank-selection metrics ------------------------------------------------ */

/* out[0] (f64) = mean_c( mean_r (x-y)^2 / (var_r(y) + eps) ), x,y viewed as [R, C],
 * var unbiased.  Replaces calc_per_channel_noise_to_signal_ratio (losses.py:10-22)
 * for non_channel_dim = all leading dims. */
/* (ABI 2) The workspace carries the arrival counters of the one-launch reduction: initialise it ONCE with
 * ptd_nsr_workspace_init (any size >= the query for the shapes it will serve); ptd_nsr leaves it initialised, so the
 * same workspace serves any number of stream-ordered calls.  Calls that may overlap need a workspace each. */
size_t ptd_nsr_workspace_bytes(int64_t R, int64_t C);
int ptd_nsr_workspace_init(void* ws, size_t ws_bytes, void* stream);
int ptd_nsr(const void* x, const void* y, int64_t R, int64_t C, int dtype, double eps, double* out,
            void* ws, size_t ws_bytes, void* stream);

/* out[0] (f64) = mean_b max(KL(t_b || s_b), KL(s_b || t_b)) over softmax(dim=-1) of
 * logits s, t [B, C].  Replaces calc_kl_loss (losses.py:48-63). */
size_t ptd_sym_kl_workspace_bytes(int64_t B);
int ptd_sym_kl(const void* s, const void* t, int64_t B, int64_t C, int dtype, double* out, void* ws,
               size_t ws_bytes, void* stream);

/* rows[b] (f64) = KL(p_b || q_b) = sum_c p log(p / q) over softmax(dim=-1) of logits q, p [B, C].
 * Replaces calc_kl_divergence(q_logits, p_logits) (losses.py:48-54). */
int ptd_kl_rows(const void* q, const void* p, int64_t B, int64_t C, int dtype, double* rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PTDECO_HIP_H */
