// Internal entry points behind the C ABI (one per .hip translation unit).
#pragma once

#include "common.h"

namespace ptd {

// gemm_f32.hip
int gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
             int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha, const float* bias, void* ws, size_t ws_bytes,
             hipStream_t st);
size_t gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K);
int gemm_f32_batched(const float* A, int64_t sam, int64_t sak, int64_t zsa, const float* B, int64_t sbk, int64_t sbn,
                     int64_t zsb, float* C, int64_t ldc, int64_t zsc, int64_t M, int64_t N, int64_t K, int64_t batch,
                     double alpha, const float* bias_rows, hipStream_t st);
int syrk_f32(const float* Y, int64_t T, int64_t n, int64_t ldy, void* E, int64_t ldE, bool e_f64, double scale,
             hipStream_t st);

// gemm_bf16.hip
int gemm_bf16(const unsigned short* A, int64_t sam, int64_t sak, const unsigned short* B, int64_t sbk, int64_t sbn,
              void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, bool c_bf16, double alpha,
              const unsigned short* bias, void* ws, size_t ws_bytes, hipStream_t st, int64_t b_kvalid = 0,
              int64_t b_nvalid = 0);
size_t gemm_bf16_workspace_bytes(int64_t M, int64_t N, int64_t K);
// [rows_out][cols] <- the first `rows` rows of src (row pitch ld), zero rows behind them
int pad_rows_bf16(const unsigned short* src, int64_t ld, int64_t rows, int64_t cols, unsigned short* dst, int64_t rows_out,
                  hipStream_t st);
int gemm_bf16_batched(const unsigned short* A, int64_t sam, int64_t sak, int64_t zsa, const unsigned short* B,
                      int64_t sbk, int64_t sbn, int64_t zsb, unsigned short* C, int64_t ldc, int64_t zsc, int64_t M,
                      int64_t N, int64_t K, int64_t batch, double alpha, const unsigned short* bias_rows,
                      hipStream_t st);
int syrk_bf16(const unsigned short* Y, int64_t T, int64_t n, int64_t ldy, void* E, int64_t ldE, bool e_f64,
              double scale, hipStream_t st);
int syrk_bf16_multi(const unsigned short* const* Ys, int steps, int64_t T, int64_t n, int64_t ldy, void* E, int64_t ldE,
                    bool e_f64, double scale, hipStream_t st);

// eigh_jacobi.hip
size_t eigh_workspace_bytes(int64_t n);
int eigh_jacobi(const double* A, int64_t lda, int64_t n, int64_t k, double* evals, double* evecs, int64_t ldv,
                void* ws, size_t ws_bytes, int* sweeps_out, ptd_eigh_stats* stats, hipStream_t st);

// gemm_f64.hip
int gemm_f64(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn, double* C,
             int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha, bool beta1, int ksplit, hipStream_t st);

int gemm_f64_slabs(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn, double* C,
                   int64_t ldc, int64_t slab, int64_t M, int64_t N, int64_t K, double alpha, int ksplit, int* nslabs,
                   bool lower_only, hipStream_t st);

// eigh_tridiag.hip
size_t tridiag_workspace_bytes(int64_t n);
// ptd_set_concurrent_chains: how many eigendecompositions the caller runs at once on the current device (returns the
// previous value)
int concurrent_chains_exchange(int chains);
int eigh_tridiag(const double* A, int64_t lda, int64_t n, int64_t k, double* evals, double* evecs, int64_t ldv,
                 void* ws, size_t ws_bytes, double cluster_tol, bool all_values, ptd_eigh_stats* stats, hipStream_t st);
int tridiagonalize_f64(const double* A, int64_t lda, int64_t n, double* d_out, double* e_out, double* evals_out,
                       void* ws, size_t ws_bytes, hipStream_t st);

// `count` matrices of one order per launch (blockIdx.y = matrix); rcs[b] = PTD_OK | PTD_ERR_UNSUPPORTED per matrix
size_t tridiag_batched_workspace_bytes(int64_t n, int count);
int eigh_tridiag_batched(const double* const* As, int64_t lda, int count, int64_t n, int64_t k, double* const* evals,
                         double* const* evecs, int64_t ldv, void* ws, size_t ws_bytes, double cluster_tol,
                         bool all_values, int* rcs, ptd_eigh_stats* stats, hipStream_t st);

// eigh_filtered.hip: top-k eigenpairs by Chebyshev-filtered subspace iteration (f64 MFMA products); declines with
// PTD_ERR_UNSUPPORTED when the spectrum does not suit it
bool eigh_filtered_applies(int64_t n, int64_t k, bool all_values);
size_t eigh_filtered_workspace_bytes(int64_t n);              // an upper bound over every k the route accepts at n
size_t eigh_filtered_workspace_bytes(int64_t n, int64_t k);   // what eigh_filtered(n, k) needs (0: does not apply)
bool eigh_filtered_backed_off(int64_t n, int64_t k);          // a recent late decline of this shape (calling thread, device)
void eigh_filtered_forget_declines();                          // ... forgotten: ptd_eigh_forget_declines
int eigh_filtered(const double* A, int64_t lda, int64_t n, int64_t k, double* evals, double* evecs, int64_t ldv,
                  void* ws, size_t ws_bytes, ptd_eigh_stats* stats, hipStream_t st);

size_t chol_inverse_workspace_bytes(int64_t m);
int chol_inverse(double* G, int64_t m, double* Wt, void* ws, size_t ws_bytes, hipStream_t st);

// eigh_factored.hip
size_t eigh_factored_workspace_bytes(int64_t n_o, int64_t n_i, int64_t k);
int eigh_factored(const void* W, int64_t ldw, int w_dtype, int64_t n_o, int64_t n_i, const double* Ex, int64_t ldx,
                  int64_t k, double* evals_k, double* U, int64_t ldu, void* ws, size_t ws_bytes, hipStream_t st);
// the same in two halves around an eigendecomposition the CALLER runs (so that the inner problems of several layers
// can share one batched call): prepare leaves B = L^T Ex L [np, np] (np = n_i rounded up to 64) in the workspace and
// returns its address; finish takes the eigenvalues [np] / top-k eigenvectors S [np, k] of B
int eigh_factored_prepare(const void* W, int64_t ldw, int w_dtype, int64_t n_o, int64_t n_i, const double* Ex,
                          int64_t ldx, int64_t k, void* ws, size_t ws_bytes, double** B_out, int64_t* np_out,
                          hipStream_t st);
int eigh_factored_finish(int64_t n_o, int64_t n_i, int64_t k, const double* evals, const double* S, int64_t lds,
                         double* evals_k, double* U, int64_t ldu, void* ws, size_t ws_bytes, hipStream_t st);

// reduce.hip
size_t cov_finalize_workspace_bytes(int64_t n);
int cov_finalize(const void* E, int64_t ldE, int E_dtype, const void* ey, int ey_dtype, int64_t n, double steps,
                 double damp_factor, double* C, int64_t ldC, void* ws, size_t ws_bytes, hipStream_t st);
int colsum_accumulate(const void* y, int64_t T, int64_t n, int64_t ldy, int y_dtype, void* ey, int ey_dtype,
                      double scale, hipStream_t st);
size_t nsr_workspace_bytes(int64_t R, int64_t C);
int nsr_workspace_init(void* ws, size_t ws_bytes, hipStream_t st);
int nsr(const void* x, const void* y, int64_t R, int64_t C, int dtype, double eps, double* out, void* ws,
        size_t ws_bytes, hipStream_t st);
int convert_f32_to_f64(const float* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t st);
int convert_f64_to_f32(const double* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t st);
size_t sym_kl_workspace_bytes(int64_t B);
int sym_kl(const void* s, const void* t, int64_t B, int64_t C, int dtype, double* out, void* ws, size_t ws_bytes,
           hipStream_t st);
int kl_rows(const void* q, const void* p, int64_t B, int64_t C, int dtype, double* rows, hipStream_t st);

}  // namespace ptd
