// extern "C" surface of libptdeco_hip.so (declared in include/ptdeco_hip.h).
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "kernels.h"

namespace ptd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace ptd

using namespace ptd;

// one wave that holds its hardware queue for `ticks` of the 100 MHz wall clock (ptd_stream_pair_wall_us)
__global__ void stream_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" {

int ptd_version(void) { return PTD_ABI_VERSION; }

const char* ptd_last_error(void) { return g_err; }

int ptd_set_concurrent_chains(int chains) { return concurrent_chains_exchange(chains < 1 ? 1 : chains); }

int ptd_streams_wall_us(void* const* streams, int count, int spin_us, double* wall_us) {
  PTD_REQUIRE(streams && wall_us && count >= 1 && count <= 64 && spin_us >= 1 && spin_us <= 100000,
              "ptd_streams_wall_us: bad argument");
  for (int i = 0; i < count; ++i) PTD_CHECK_HIP(hipStreamSynchronize(static_cast<hipStream_t>(streams[i])));
  const long long ticks = (long long)spin_us * 100;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < count; ++i)
    hipLaunchKernelGGL(stream_spin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(streams[i]), ticks);
  for (int i = 0; i < count; ++i) PTD_CHECK_HIP(hipStreamSynchronize(static_cast<hipStream_t>(streams[i])));
  *wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  PTD_CHECK_LAUNCH("ptd_streams_wall_us");
  return PTD_OK;
}

int ptd_stream_create_dedicated(int cu_first, int cu_count, void** stream_out) {
  PTD_REQUIRE(stream_out && cu_first >= 0 && cu_count >= 0, "ptd_stream_create_dedicated: bad argument");
  int dev = 0, cus = 0;
  PTD_CHECK_HIP(hipGetDevice(&dev));
  PTD_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (cu_count == 0) { cu_first = 0; cu_count = cus; }
  PTD_REQUIRE(cu_first + cu_count <= cus, "ptd_stream_create_dedicated: CUs [%d, %d) of %d", cu_first, cu_first + cu_count, cus);
  uint32_t mask[32] = {0};
  for (int c = cu_first; c < cu_first + cu_count && c < 1024; ++c) mask[c >> 5] |= 1u << (c & 31);
  hipStream_t st = nullptr;
  PTD_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)((cus + 31) / 32), mask));
  *stream_out = st;
  return PTD_OK;
}

int ptd_stream_destroy(void* stream) {
  PTD_CHECK_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return PTD_OK;
}

int ptd_stream_pair_wall_us(void* stream_a, void* stream_b, int spin_us, double* wall_us) {
  void* pair[2] = {stream_a, stream_b};
  return ptd_streams_wall_us(pair, 2, spin_us, wall_us);
}

int ptd_syrk_accumulate(const void* y, int64_t T, int64_t n, int64_t ldy, int y_dtype, void* E, int64_t ldE,
                        int E_dtype, double scale, void* stream) {
  PTD_REQUIRE(y && E, "ptd_syrk_accumulate: null pointer");
  PTD_REQUIRE(T >= 0 && n >= 0 && ldy >= n && ldE >= n, "ptd_syrk_accumulate: bad shape T=%lld n=%lld ldy=%lld ldE=%lld",
              (long long)T, (long long)n, (long long)ldy, (long long)ldE);
  PTD_REQUIRE(E_dtype == PTD_F64 || E_dtype == PTD_F32, "ptd_syrk_accumulate: E must be f64 or f32");
  PTD_REQUIRE(T < (1ll << 31) && n < (1ll << 31), "ptd_syrk_accumulate: dimension too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (y_dtype == PTD_F32)
    return syrk_f32(static_cast<const float*>(y), T, n, ldy, E, ldE, E_dtype == PTD_F64, scale, st);
  if (y_dtype == PTD_BF16)
    return syrk_bf16(static_cast<const unsigned short*>(y), T, n, ldy, E, ldE, E_dtype == PTD_F64, scale, st);
  set_error("ptd_syrk_accumulate: y dtype must be f32 or bf16");
  return PTD_ERR_UNSUPPORTED;
}

int ptd_syrk_accumulate_multi(const void* const* ys, int steps, int64_t T, int64_t n, int64_t ldy, int y_dtype, void* E,
                              int64_t ldE, int E_dtype, double scale, void* stream) {
  PTD_REQUIRE(ys && E && steps >= 0 && steps <= 4096, "ptd_syrk_accumulate_multi: bad argument");
  for (int s = 0; s < steps; ++s) PTD_REQUIRE(ys[s], "ptd_syrk_accumulate_multi: null pointer (step %d)", s);
  PTD_REQUIRE(T >= 0 && n >= 0 && ldy >= n && ldE >= n, "ptd_syrk_accumulate_multi: bad shape T=%lld n=%lld ldy=%lld ldE=%lld",
              (long long)T, (long long)n, (long long)ldy, (long long)ldE);
  PTD_REQUIRE(E_dtype == PTD_F64 || E_dtype == PTD_F32, "ptd_syrk_accumulate_multi: E must be f64 or f32");
  PTD_REQUIRE(T < (1ll << 31) && n < (1ll << 31), "ptd_syrk_accumulate_multi: dimension too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (y_dtype == PTD_BF16)
    return syrk_bf16_multi(reinterpret_cast<const unsigned short* const*>(ys), steps, T, n, ldy, E, ldE,
                           E_dtype == PTD_F64, scale, st);
  if (y_dtype == PTD_F32) {
    // the f32 product is bound by the matrix cores, not by the accumulator's read-modify-write: step by step
    for (int s = 0; s < steps; ++s) {
      const int rc = syrk_f32(static_cast<const float*>(ys[s]), T, n, ldy, E, ldE, E_dtype == PTD_F64, scale, st);
      if (rc != PTD_OK) return rc;
    }
    return PTD_OK;
  }
  set_error("ptd_syrk_accumulate_multi: y dtype must be f32 or bf16");
  return PTD_ERR_UNSUPPORTED;
}

int ptd_colsum_accumulate(const void* y, int64_t T, int64_t n, int64_t ldy, int y_dtype, void* ey, int ey_dtype,
                          double scale, void* stream) {
  return colsum_accumulate(y, T, n, ldy, y_dtype, ey, ey_dtype, scale, static_cast<hipStream_t>(stream));
}

size_t ptd_cov_finalize_workspace_bytes(int64_t n) { return cov_finalize_workspace_bytes(n); }

int ptd_cov_finalize(const void* E, int64_t ldE, int E_dtype, const void* ey, int ey_dtype, int64_t n, double steps,
                     double damp_factor, double* C, int64_t ldC, void* ws, size_t ws_bytes, void* stream) {
  return cov_finalize(E, ldE, E_dtype, ey, ey_dtype, n, steps, damp_factor, C, ldC, ws, ws_bytes,
                      static_cast<hipStream_t>(stream));
}

// Solver choice (see include/ptdeco_hip.h).  Read on every call so a process can switch.
static int eigh_method() {
  const char* e = getenv("PTD_EIGH_METHOD");
  if (!e || !strcmp(e, "auto")) return 2;
  if (!strcmp(e, "tridiag")) return 1;
  return 0;
}

size_t ptd_eigh_workspace_bytes(int64_t n) {
  return std::max(std::max(eigh_workspace_bytes(n), tridiag_workspace_bytes(n)), eigh_filtered_workspace_bytes(n));
}

static int eigh_dispatch(const double* A, int64_t lda, int64_t n, int64_t k, double* evals, double* evecs,
                         int64_t ldv, void* ws, size_t ws_bytes, int* sweeps_out, bool all_values,
                         ptd_eigh_stats* stats, hipStream_t st, bool direct = false) {
  const int method = direct && eigh_method() == 2 ? 1 : eigh_method();
  // a quarter of the spectrum of a large matrix: filtered subspace iteration on the f64 matrix cores; it declines
  // (flat spectrum, breakdown, residual above tolerance) with PTD_ERR_UNSUPPORTED and the direct route below runs
  if (method == 2 && A && evals && evecs && ws && lda >= n && k >= 1 && k <= n && ldv >= k && (lda % 2) == 0 &&
      (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
      eigh_filtered_applies(n, k, all_values) && ws_bytes >= eigh_filtered_workspace_bytes(n, k) &&
      !eigh_filtered_backed_off(n, k)) {
    const int rc = eigh_filtered(A, lda, n, k, evals, evecs, ldv, ws, ws_bytes, stats, st);
    if (rc != PTD_ERR_UNSUPPORTED && rc != PTD_ERR_WORKSPACE) {
      if (sweeps_out) *sweeps_out = 0;
      return rc;
    }
  }
  if (method != 0 && (method == 1 || n >= 256) && A && evals && evecs && ws && n >= 2 && lda >= n && k >= 1 &&
      k <= n && ldv >= k) {
    const char* ct = getenv("PTD_EIGH_CLUSTER_TOL");
    const int rc = eigh_tridiag(A, lda, n, k, evals, evecs, ldv, ws, ws_bytes, ct ? atof(ct) : 1e-10, all_values, stats, st);
    if (rc != PTD_ERR_UNSUPPORTED) {
      if (sweeps_out) *sweeps_out = 0;
      return rc;
    }
  }
  return eigh_jacobi(A, lda, n, k, evals, evecs, ldv, ws, ws_bytes, sweeps_out, stats, st);
}

void ptd_eigh_forget_declines(void) { eigh_filtered_forget_declines(); }

int ptd_eigh_route(int64_t n, int64_t k, int all_values) {
  // what eigh_dispatch would try first for this request (the filtered route may still decline at run time)
  const int method = eigh_method();
  if (method == 2 && eigh_filtered_applies(n, k, all_values != 0)) return 3;
  if (method != 0 && (method == 1 || n >= 256) && n >= 2) return 1;
  return 0;
}

// ---- several matrices of one order in one call
// count == 2 matrices are batched from this order on (two matrices of order 1024 take 8.3 ms either way -- a shared
// blocked column of ~8 us against two resident ones of ~4 -- but the batched pair keeps that time beside another
// stream's work, where single calls lose their resident kernels: 9 ms each); count >= 3: always
static int batch_min_n() {
  const char* e = getenv("PTD_EIGH_BATCH_MIN_N");
  return e ? atoi(e) : 512;
}

static bool batch_route(int count, int64_t n, int64_t k, bool all_values, bool direct = false) {
  const int method = eigh_method();
  if (count < 2 || method == 0 || n < 256) return false;
  // (the filtered route's f64 products fill the chip: one by one -- unless the caller asks for the direct reduction)
  if (!direct && method == 2 && eigh_filtered_applies(n, k, all_values)) return false;
  const char* off = getenv("PTD_EIGH_BATCHED");
  if (off && atoi(off) == 0) return false;
  return count >= 3 || n >= batch_min_n();
}

size_t ptd_eigh_batched_workspace_bytes(int64_t n, int64_t k, int count) {
  (void)k;
  return std::max(ptd_eigh_workspace_bytes(n), tridiag_batched_workspace_bytes(n, count));
}

int ptd_eigh_topk_batched(const double* const* As, int64_t lda, int count, int64_t n, int64_t k, int all_values,
                          double* const* evals, double* const* evecs, int64_t ldv, void* ws, size_t ws_bytes,
                          ptd_eigh_stats* stats, void* stream) {
  PTD_REQUIRE(As && evals && evecs && ws && count >= 1 && count <= 64 && n >= 1 && lda >= n && k >= 1 && k <= n && ldv >= k,
              "ptd_eigh_topk_batched: bad argument");
  for (int b = 0; b < count; ++b)
    PTD_REQUIRE(As[b] && evals[b] && evecs[b], "ptd_eigh_topk_batched: null pointer (matrix %d)", b);
  if (ws_bytes < ptd_eigh_batched_workspace_bytes(n, k, count)) {
    set_error("ptd_eigh_topk_batched: workspace %zu < required %zu bytes", ws_bytes,
              ptd_eigh_batched_workspace_bytes(n, k, count));
    return PTD_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats) memset(stats, 0, sizeof(*stats));
  // all_values is a set of flags: bit 0 = every eigenvalue, bit 1 (PTD_EIGH_FLAG_DIRECT) = the direct reduction also
  // where the filtered route would serve the request (a caller whose pass already runs latency-bound reductions of
  // this order beside this call: the filter's products would share the matrix cores with them)
  const bool direct = (all_values & 2) != 0;
  all_values &= 1;
  if (!batch_route(count, n, k, all_values != 0, direct)) {
    // one by one, each with the whole chip (the filtered route's products fill it; a single direct reduction keeps
    // its resident kernels); stats describe the LAST matrix
    for (int b = 0; b < count; ++b) {
      const int rc = eigh_dispatch(As[b], lda, n, k, evals[b], evecs[b], ldv, ws, ws_bytes, nullptr, all_values != 0,
                                   stats, st, direct);
      if (rc != PTD_OK) return rc;
    }
    return PTD_OK;
  }
  int rcs[64];
  const char* ct = getenv("PTD_EIGH_CLUSTER_TOL");
  int rc = eigh_tridiag_batched(As, lda, count, n, k, evals, evecs, ldv, ws, ws_bytes, ct ? atof(ct) : 1e-10,
                                all_values != 0, rcs, stats, st);
  if (rc != PTD_OK) return rc;
  for (int b = 0; b < count; ++b)
    if (rcs[b] == PTD_ERR_UNSUPPORTED) {     // clustered beyond what the tridiagonal route serves: Jacobi needs no gap
      rc = eigh_jacobi(As[b], lda, n, k, evals[b], evecs[b], ldv, ws, ws_bytes, nullptr, nullptr, st);
      if (rc != PTD_OK) return rc;
    }
  return PTD_OK;
}

size_t ptd_eigh_factored_workspace_bytes(int64_t n_o, int64_t n_i, int64_t k) {
  return eigh_factored_workspace_bytes(n_o, n_i, k);
}

int ptd_eigh_factored_prepare(const void* W, int64_t ldw, int w_dtype, int64_t n_o, int64_t n_i, const double* Ex,
                              int64_t ldx, int64_t k, void* ws, size_t ws_bytes, double** B_out, int64_t* np_out,
                              void* stream) {
  return eigh_factored_prepare(W, ldw, w_dtype, n_o, n_i, Ex, ldx, k, ws, ws_bytes, B_out, np_out,
                               static_cast<hipStream_t>(stream));
}

int ptd_eigh_factored_finish(int64_t n_o, int64_t n_i, int64_t k, const double* evals, const double* S, int64_t lds,
                             double* evals_k, double* U, int64_t ldu, void* ws, size_t ws_bytes, void* stream) {
  return eigh_factored_finish(n_o, n_i, k, evals, S, lds, evals_k, U, ldu, ws, ws_bytes,
                              static_cast<hipStream_t>(stream));
}

int ptd_eigh_factored(const void* W, int64_t ldw, int w_dtype, int64_t n_o, int64_t n_i, const double* Ex,
                      int64_t ldx, int64_t k, double* evals_k, double* U, int64_t ldu, void* ws, size_t ws_bytes,
                      void* stream) {
  return eigh_factored(W, ldw, w_dtype, n_o, n_i, Ex, ldx, k, evals_k, U, ldu, ws, ws_bytes,
                       static_cast<hipStream_t>(stream));
}

int ptd_eigh(const double* A, int64_t lda, int64_t n, double* evals, double* evecs, int64_t ldv, void* ws,
             size_t ws_bytes, int* sweeps_out, void* stream) {
  return eigh_dispatch(A, lda, n, n, evals, evecs, ldv, ws, ws_bytes, sweeps_out, true, nullptr,
                       static_cast<hipStream_t>(stream));
}

int ptd_eigh_topk(const double* A, int64_t lda, int64_t n, int64_t k, int all_values, double* evals, double* evecs,
                  int64_t ldv, void* ws, size_t ws_bytes, int* sweeps_out, void* stream) {
  return eigh_dispatch(A, lda, n, k, evals, evecs, ldv, ws, ws_bytes, sweeps_out, all_values != 0, nullptr,
                       static_cast<hipStream_t>(stream));
}

// The f32 face of ptd_eigh_topk (decompose_in_float64=False runs torch.linalg.eigh on an f32 matrix, dwain.py:224-233 +
// 162): f32 in, f32 out, the arithmetic in between in f64 -- the routes above on a converted copy -- so the result is
// the f32 rounding of the f64 eigenpairs of the f32 matrix (at least as accurate as an f32 LAPACK call).
size_t ptd_eigh_f32_workspace_bytes(int64_t n, int64_t k) {
  if (n < 1) return 0;
  k = std::max<int64_t>(1, std::min(k, n));
  return align_up((size_t)n * n * 8, 256) + align_up((size_t)n * 8, 256) + align_up((size_t)n * k * 8, 256) +
         ptd_eigh_workspace_bytes(n);
}

int ptd_eigh_topk_f32(const float* A, int64_t lda, int64_t n, int64_t k, int all_values, float* evals, float* evecs,
                      int64_t ldv, void* ws, size_t ws_bytes, int* sweeps_out, void* stream) {
  PTD_REQUIRE(A && evals && evecs && ws && n >= 1 && lda >= n && k >= 1 && k <= n && ldv >= k,
              "ptd_eigh_topk_f32: bad argument");
  if (ws_bytes < ptd_eigh_f32_workspace_bytes(n, k)) {
    set_error("ptd_eigh_topk_f32: workspace %zu < required %zu bytes", ws_bytes, ptd_eigh_f32_workspace_bytes(n, k));
    return PTD_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(ws);
  double* A64 = reinterpret_cast<double*>(base);
  double* w64 = reinterpret_cast<double*>(base + align_up((size_t)n * n * 8, 256));
  double* v64 = reinterpret_cast<double*>(base + align_up((size_t)n * n * 8, 256) + align_up((size_t)n * 8, 256));
  char* rest = base + align_up((size_t)n * n * 8, 256) + align_up((size_t)n * 8, 256) + align_up((size_t)n * k * 8, 256);
  int rc = convert_f32_to_f64(A, lda, A64, n, n, n, st);
  if (rc != PTD_OK) return rc;
  rc = eigh_dispatch(A64, n, n, k, w64, v64, k, rest, ws_bytes - (size_t)(rest - base), sweeps_out, all_values != 0, nullptr, st);
  if (rc != PTD_OK) return rc;
  rc = convert_f64_to_f32(w64, n, evals, n, 1, n, st);      // (NaN where the route did not compute an eigenvalue stays NaN)
  if (rc != PTD_OK) return rc;
  return convert_f64_to_f32(v64, k, evecs, ldv, n, k, st);
}

int ptd_eigh_profiled(const double* A, int64_t lda, int64_t n, int64_t k, int all_values, double* evals,
                      double* evecs, int64_t ldv, void* ws, size_t ws_bytes, ptd_eigh_stats* stats, void* stream) {
  PTD_REQUIRE(stats, "ptd_eigh_profiled: stats must not be null");
  return eigh_dispatch(A, lda, n, k, evals, evecs, ldv, ws, ws_bytes, nullptr, all_values != 0, stats,
                       static_cast<hipStream_t>(stream));
}

size_t ptd_chol_inverse_workspace_bytes(int64_t m) { return chol_inverse_workspace_bytes(m); }

int ptd_chol_inverse(double* G, int64_t m, double* Wt, void* ws, size_t ws_bytes, void* stream) {
  return chol_inverse(G, m, Wt, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

size_t ptd_tridiagonalize_workspace_bytes(int64_t n) { return tridiag_workspace_bytes(n); }

int ptd_tridiagonalize(const double* A, int64_t lda, int64_t n, double* d, double* e, double* evals, void* ws,
                       size_t ws_bytes, void* stream) {
  return tridiagonalize_f64(A, lda, n, d, e, evals, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

size_t ptd_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int ab_dtype, int c_dtype) {
  if (ab_dtype == PTD_F32 && c_dtype == PTD_F32) return gemm_f32_workspace_bytes(M, N, K);
  if (ab_dtype == PTD_BF16 && (c_dtype == PTD_BF16 || c_dtype == PTD_F32)) return gemm_bf16_workspace_bytes(M, N, K);
  return 0;
}

int ptd_gemm(const void* A, int64_t sam, int64_t sak, const void* B, int64_t sbk, int64_t sbn, void* C, int64_t ldc,
             int64_t M, int64_t N, int64_t K, int ab_dtype, int c_dtype, double alpha, const void* bias,
             void* stream) {
  return ptd_gemm_ws(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, ab_dtype, c_dtype, alpha, bias, nullptr, 0, stream);
}

int ptd_gemm_ws(const void* A, int64_t sam, int64_t sak, const void* B, int64_t sbk, int64_t sbn, void* C, int64_t ldc,
                int64_t M, int64_t N, int64_t K, int ab_dtype, int c_dtype, double alpha, const void* bias, void* ws,
                size_t ws_bytes, void* stream) {
  PTD_REQUIRE(A && B && C, "ptd_gemm: null pointer");
  PTD_REQUIRE(M >= 0 && N >= 0 && K >= 0 && ldc >= N, "ptd_gemm: bad shape");
  PTD_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "ptd_gemm: dimension too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!ws) ws_bytes = 0;
  if (ab_dtype == PTD_F32 && c_dtype == PTD_F32)
    return gemm_f32(static_cast<const float*>(A), sam, sak, static_cast<const float*>(B), sbk, sbn,
                    static_cast<float*>(C), ldc, M, N, K, alpha, static_cast<const float*>(bias), ws, ws_bytes, st);
  if (ab_dtype == PTD_BF16 && (c_dtype == PTD_BF16 || c_dtype == PTD_F32))
    return gemm_bf16(static_cast<const unsigned short*>(A), sam, sak, static_cast<const unsigned short*>(B), sbk,
                     sbn, C, ldc, M, N, K, c_dtype == PTD_BF16, alpha, static_cast<const unsigned short*>(bias),
                     ws, ws_bytes, st);
  if (ab_dtype == PTD_F64 && c_dtype == PTD_F64 && !bias)
    return gemm_f64(static_cast<const double*>(A), sam, sak, static_cast<const double*>(B), sbk, sbn,
                    static_cast<double*>(C), ldc, M, N, K, alpha, false, 1, st);
  set_error("ptd_gemm: unsupported dtype combination ab=%d c=%d", ab_dtype, c_dtype);
  return PTD_ERR_UNSUPPORTED;
}

static size_t elt_bytes(int dtype) { return dtype == PTD_F32 ? 4 : 2; }

// bf16 ranks that are not a multiple of 128 (a dwain search ends at 32 .. 96 on wide layers: Llama gate / up): the first
// product's output would have fewer than 128 columns -- a handful of 128-row tiles for the whole chip (x A^T at
// T = 2048, r = 32: 98 us on 16 workgroups, the library 20) -- and K = r of the second product no whole 64-deep step.
// The pair then runs on the rank padded with zeros: A's rows to a multiple of 128 (a copy in the workspace: the first
// product takes the split-K path of a 128-column tile row and leaves h [T, r128] with exact zeros behind column r),
// K of the second product to a multiple of 64 (B is read in place: its pieces behind column r are fetched from the
// row's start, h is zero there).  The results are those of the unpadded products: the padding adds exact zeros.
static int64_t lowrank_pad128(int64_t r, int dtype) {
  static const bool off = getenv("PTD_LOWRANK_PAD") && atoi(getenv("PTD_LOWRANK_PAD")) == 0;
  if (off || dtype != PTD_BF16 || r % 8 != 0 || r % 128 == 0 || r > 1024) return r;
  return r <= 64 ? 64 : (int64_t)align_up((size_t)r, 128);     // (one column of 128 x 64 tiles serves a rank up to 64)
}

size_t ptd_lowrank_forward_workspace_bytes(int64_t T, int64_t n_i, int64_t r, int dtype) {
  const int64_t rp = lowrank_pad128(r, dtype);
  size_t b = align_up((size_t)T * (size_t)rp * elt_bytes(dtype), 256);
  if (rp != r) b += align_up((size_t)rp * (size_t)n_i * elt_bytes(dtype), 256);
  if (dtype == PTD_F32) b += gemm_f32_workspace_bytes(T, r, n_i);
  else b += gemm_bf16_workspace_bytes(T, rp, n_i);
  return b;
}

int ptd_lowrank_forward(const void* x, int64_t ldx, int64_t T, int64_t n_i, const void* A, int64_t lda, int64_t r,
                        const void* B, int64_t ldb, int64_t n_o, const void* bias, void* y, int64_t ldy, void* ws,
                        size_t ws_bytes, int dtype, void* stream) {
  PTD_REQUIRE(x && A && B && y && ws, "ptd_lowrank_forward: null pointer");
  PTD_REQUIRE(ldx >= n_i && lda >= n_i && ldb >= r && ldy >= n_o, "ptd_lowrank_forward: bad leading dimension");
  PTD_REQUIRE(dtype == PTD_F32 || dtype == PTD_BF16, "ptd_lowrank_forward: dtype must be f32 or bf16");
  const int64_t rp = lowrank_pad128(r, dtype);
  if (rp != r && n_i % 8 == 0 && lda % 8 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
      ws_bytes >= ptd_lowrank_forward_workspace_bytes(T, n_i, r, dtype)) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    typedef unsigned short u16;
    const size_t hp_bytes = align_up((size_t)T * (size_t)rp * 2, 256), ap_bytes = align_up((size_t)rp * (size_t)n_i * 2, 256);
    u16* hp = static_cast<u16*>(ws);
    u16* Ap = reinterpret_cast<u16*>(static_cast<char*>(ws) + hp_bytes);
    char* rest = static_cast<char*>(ws) + hp_bytes + ap_bytes;
    // the first product on the padded rank: A read in place -- its rows behind r are fetched from row 0 and the columns
    // they would produce are written as zeros (GemmBf16Args::nvalid); where no LDS-DMA kernel serves the shape, on a
    // zero-padded copy of A
    int rc = gemm_bf16(static_cast<const u16*>(x), ldx, 1, static_cast<const u16*>(A), 1, lda, hp, rp, T, rp, n_i, true, 1.0,
                       nullptr, rest, ws_bytes - hp_bytes - ap_bytes, st, 0, r);
    if (rc == PTD_ERR_UNSUPPORTED) {
      rc = pad_rows_bf16(static_cast<const u16*>(A), lda, r, n_i, Ap, rp, st);
      if (rc != PTD_OK) return rc;
      rc = gemm_bf16(static_cast<const u16*>(x), ldx, 1, Ap, 1, n_i, hp, rp, T, rp, n_i, true, 1.0, nullptr, rest,
                     ws_bytes - hp_bytes - ap_bytes, st);
    }
    if (rc != PTD_OK) return rc;
    // y = h B^T + bias with K padded to whole 64-deep steps where a short-K kernel serves the shape, else with K = r
    const int64_t k64 = (int64_t)align_up((size_t)r, 64);
    if (k64 != r && k64 <= 256) {
      rc = gemm_bf16(hp, rp, 1, static_cast<const u16*>(B), 1, ldb, y, ldy, T, n_o, k64, true, 1.0,
                     static_cast<const u16*>(bias), nullptr, 0, st, r);
      if (rc != PTD_ERR_UNSUPPORTED) return rc;
    }
    return ptd_gemm(hp, rp, 1, B, 1, ldb, y, ldy, T, n_o, r, dtype, dtype, 1.0, bias, stream);
  }
  const size_t h_bytes = align_up((size_t)T * (size_t)r * elt_bytes(dtype), 256);
  if (ws_bytes < h_bytes) {
    set_error("ptd_lowrank_forward: workspace %zu < required %zu bytes", ws_bytes, h_bytes);
    return PTD_ERR_WORKSPACE;
  }
  void* h = ws;
  // h = x A^T : A(m,k) = x[m*ldx + k], B(k,n) = A[n*lda + k];   y = h B^T + bias
  int rc;
  if (dtype == PTD_F32) {
    // a small rank leaves the first product with few output tiles: its K range is split over
    // workgroups through the rest of the workspace (deterministic two-pass reduction)
    rc = gemm_f32(static_cast<const float*>(x), ldx, 1, static_cast<const float*>(A), 1, lda, static_cast<float*>(h), r,
                  T, r, n_i, 1.0, nullptr, static_cast<char*>(ws) + h_bytes, ws_bytes - h_bytes,
                  static_cast<hipStream_t>(stream));
  } else {
    rc = gemm_bf16(static_cast<const unsigned short*>(x), ldx, 1, static_cast<const unsigned short*>(A), 1, lda, h, r,
                   T, r, n_i, true, 1.0, nullptr, static_cast<char*>(ws) + h_bytes, ws_bytes - h_bytes,
                   static_cast<hipStream_t>(stream));
  }
  if (rc != PTD_OK) return rc;
  return ptd_gemm(h, r, 1, B, 1, ldb, y, ldy, T, n_o, r, dtype, dtype, 1.0, bias, stream);
}

size_t ptd_lowrank_forward_nchw_workspace_bytes(int64_t batch, int64_t hw, int64_t r, int dtype) {
  return align_up((size_t)batch * (size_t)r * (size_t)hw * elt_bytes(dtype), 256);
}

int ptd_lowrank_forward_nchw(const void* x, int64_t batch, int64_t n_i, int64_t hw, const void* A, int64_t lda,
                             int64_t r, const void* B, int64_t ldb, int64_t n_o, const void* bias, void* y, void* ws,
                             size_t ws_bytes, int dtype, void* stream) {
  PTD_REQUIRE(x && A && B && y && ws, "ptd_lowrank_forward_nchw: null pointer");
  PTD_REQUIRE(batch >= 0 && n_i >= 1 && hw >= 1 && r >= 1 && n_o >= 1 && lda >= n_i && ldb >= r,
              "ptd_lowrank_forward_nchw: bad shape");
  PTD_REQUIRE(dtype == PTD_F32 || dtype == PTD_BF16, "ptd_lowrank_forward_nchw: dtype must be f32 or bf16");
  if (ws_bytes < ptd_lowrank_forward_nchw_workspace_bytes(batch, hw, r, dtype)) {
    set_error("ptd_lowrank_forward_nchw: workspace too small");
    return PTD_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // per image b: h_b[r, hw] = A x_b (x_b = x + b n_i hw viewed [n_i, hw], pixels contiguous), y_b[n_o, hw] = B h_b + bias
  if (dtype == PTD_F32) {
    int rc = gemm_f32_batched(static_cast<const float*>(A), lda, 1, 0, static_cast<const float*>(x), hw, 1, n_i * hw,
                              static_cast<float*>(ws), hw, r * hw, r, hw, n_i, batch, 1.0, nullptr, st);
    if (rc != PTD_OK) return rc;
    return gemm_f32_batched(static_cast<const float*>(B), ldb, 1, 0, static_cast<const float*>(ws), hw, 1, r * hw,
                            static_cast<float*>(y), hw, n_o * hw, n_o, hw, r, batch, 1.0,
                            static_cast<const float*>(bias), st);
  }
  typedef unsigned short u16;
  int rc = gemm_bf16_batched(static_cast<const u16*>(A), lda, 1, 0, static_cast<const u16*>(x), hw, 1, n_i * hw,
                             static_cast<u16*>(ws), hw, r * hw, r, hw, n_i, batch, 1.0, nullptr, st);
  if (rc != PTD_OK) return rc;
  return gemm_bf16_batched(static_cast<const u16*>(B), ldb, 1, 0, static_cast<const u16*>(ws), hw, 1, r * hw,
                           static_cast<u16*>(y), hw, n_o * hw, n_o, hw, r, batch, 1.0, static_cast<const u16*>(bias),
                           st);
}

size_t ptd_nsr_workspace_bytes(int64_t R, int64_t C) { return nsr_workspace_bytes(R, C); }

int ptd_nsr_workspace_init(void* ws, size_t ws_bytes, void* stream) {
  return nsr_workspace_init(ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int ptd_nsr(const void* x, const void* y, int64_t R, int64_t C, int dtype, double eps, double* out, void* ws,
            size_t ws_bytes, void* stream) {
  return nsr(x, y, R, C, dtype, eps, out, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

size_t ptd_sym_kl_workspace_bytes(int64_t B) { return sym_kl_workspace_bytes(B); }

int ptd_sym_kl(const void* s, const void* t, int64_t B, int64_t C, int dtype, double* out, void* ws, size_t ws_bytes,
               void* stream) {
  return sym_kl(s, t, B, C, dtype, out, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int ptd_kl_rows(const void* q, const void* p, int64_t B, int64_t C, int dtype, double* rows, void* stream) {
  return kl_rows(q, p, B, C, dtype, rows, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// C++-linkage views of the solver selection for the other translation units (eigh_factored.hip)
namespace ptd {
size_t eigh_select_workspace_bytes(int64_t n) { return ptd_eigh_workspace_bytes(n); }
int eigh_select(const double* A, int64_t lda, int64_t n, int64_t k, double* evals, double* evecs, int64_t ldv,
                void* ws, size_t ws_bytes, int* sweeps_out, ptd_eigh_stats* stats, hipStream_t st) {
  // the factored route reads only the k largest eigenvalues of its reduced problem
  return eigh_dispatch(A, lda, n, k, evals, evecs, ldv, ws, ws_bytes, sweeps_out, false, stats, st);
}
}  // namespace ptd
