// Blocked f64 Cholesky factorisation A = L L^T on the gfx950 matrix cores, used to
// precondition the Jacobi eigensolver (Veselic-Hari): the one-sided Jacobi method run on
// L^T works with singular values sqrt(lambda) instead of lambda, i.e. half the dynamic
// range in the row norms, and L^T L is one LR step closer to diagonal than A.
//
// Right-looking, block size 64, in place on the lower triangle of an np x np buffer
// (np multiple of 64).  Per block column k three launches:
//   diag    one workgroup: L_kk = chol(A_kk) in LDS, plus L_kk^-1 by forward substitution
//   trsm    one workgroup per 64-row tile below:  X_i = A_ik L_kk^-T        (f64 MFMA)
//   update  one workgroup per lower-triangle tile: A_ij -= X_i X_j^T        (f64 MFMA)
// A non-positive pivot raises *fail (the matrix is not numerically positive definite);
// the caller then falls back to running Jacobi on A itself.
#include "chol_tile.h"
#include "common.h"

namespace ptd {

namespace {

constexpr int CB = 64;
constexpr int CP = 66;  // LDS pitch (doubles) of a 64 x 64 operand tile

// L_kk = chol(A_kk) and L_kk^-1 of the diagonal block, one workgroup.  Round 4: the hierarchical tile of chol_tile.h
// (16 x 16 leaves on one wave with cross-lane moves, everything else 16 x 16 x 16 products on the matrix cores) instead
// of a column-by-column factorisation in LDS with three barriers a column and a forward substitution by one thread per
// column: 161 -> ~20 us per block (64 blocks at n = 4096: 10 of the 91 ms of ptd_eigh_factored at 14336 x 4096).
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ L, int np, int k,
                                                        double* __restrict__ Linv, int* __restrict__ fail) {
  __shared__ __attribute__((aligned(16))) double T[FB * FQ];
  __shared__ __attribute__((aligned(16))) double Xs[FB * FQ];
  __shared__ __attribute__((aligned(16))) double Dv[8 * 16 * LP];
  const int tid = threadIdx.x;
  double* blk = L + ((int64_t)k * CB) * np + (int64_t)k * CB;
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e >> 6, c = e & 63;
    T[r * FQ + c] = (c <= r) ? blk[(int64_t)r * np + c] : 0.0;
  }
  __syncthreads();
  fs_chol_inv_tile2<true>(T, Dv, Dv + 4 * 16 * LP, Xs, Linv, fail, tid);   // Linv (global) and L in T
  __syncthreads();
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e >> 6, c = e & 63;
    blk[(int64_t)r * np + c] = (c <= r) ? T[r * FQ + c] : 0.0;  // upper part of the diagonal block becomes 0
  }
}

// C[64 x 64] (+)= sign * P Q^T with P, Q 64 x 64 row-major tiles held in LDS (pitch CP); the
// four waves own 32 x 32 quadrants as 2 x 2 f64 MFMA tiles.  acc layout = v_mfma_f64_16x16x4.
__device__ __forceinline__ void tile_abt(const double* __restrict__ Ps, const double* __restrict__ Qs, int wr, int wc,
                                         int lane, f64x4 (&acc)[2][2]) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < CB; kk += 4) {
    const double a0 = Ps[(wr * 32 + l15) * CP + kk + l4];
    const double a1 = Ps[(wr * 32 + 16 + l15) * CP + kk + l4];
    const double b0 = Qs[(wc * 32 + l15) * CP + kk + l4];
    const double b1 = Qs[(wc * 32 + 16 + l15) * CP + kk + l4];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
}

__device__ __forceinline__ void load_tile64(double* __restrict__ dst, const double* __restrict__ src, int64_t ld,
                                            int tid) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int idx = tid + 256 * p;
    const int r = idx >> 5, c2 = (idx & 31) * 2;
    *reinterpret_cast<double2*>(&dst[r * CP + c2]) = *reinterpret_cast<const double2*>(src + (int64_t)r * ld + c2);
  }
}

// X_i = A_ik * Linv^T for the row tiles i = k+1 .. nblk-1 (blockIdx.x = i - k - 1)
__global__ __launch_bounds__(256) void chol_trsm_kernel(double* __restrict__ L, int np, int k,
                                                        const double* __restrict__ Linv) {
  __shared__ __attribute__((aligned(16))) double As[CB * CP];
  __shared__ __attribute__((aligned(16))) double Bs[CB * CP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int i = k + 1 + blockIdx.x;
  double* tile = L + ((int64_t)i * CB) * np + (int64_t)k * CB;
  load_tile64(As, tile, np, tid);
  load_tile64(Bs, Linv, CB, tid);
  __syncthreads();
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  tile_abt(As, Bs, wr, wc, lane, acc);  // X[r][c] = sum_j A[r][j] * Linv[c][j]
  __syncthreads();
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * 32 + a * 16 + l4 + 4 * r, col = wc * 32 + b * 16 + l15;
        tile[(int64_t)row * np + col] = acc[a][b][r];
      }
}

// A_ij -= X_i X_j^T for k < j <= i (blockIdx.x enumerates the lower-triangle tiles)
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ L, int np, int k) {
  __shared__ __attribute__((aligned(16))) double Ps[CB * CP];
  __shared__ __attribute__((aligned(16))) double Qs[CB * CP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int t = blockIdx.x;
  int ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
  while (ti * (ti + 1) / 2 > t) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int i = k + 1 + ti, j = k + 1 + tj;
  load_tile64(Ps, L + ((int64_t)i * CB) * np + (int64_t)k * CB, np, tid);
  load_tile64(Qs, L + ((int64_t)j * CB) * np + (int64_t)k * CB, np, tid);
  __syncthreads();
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  tile_abt(Ps, Qs, wr, wc, lane, acc);
  double* c = L + ((int64_t)i * CB) * np + (int64_t)j * CB;
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * 32 + a * 16 + l4 + 4 * r, col = wc * 32 + b * 16 + l15;
        c[(int64_t)row * np + col] -= acc[a][b][r];
      }
}

}  // namespace

// In-place blocked Cholesky of the lower triangle of L (np x np, np % 64 == 0).
// linv_ws: 64 x 64 doubles per diagonal block when linv_stride == 4096 (all inverses L_kk^-1
// are kept, for a later triangular solve), or one reused 64 x 64 buffer when linv_stride == 0;
// fail: device int, must be zero on entry.
int cholesky_f64(double* L, int np, double* linv_ws, int linv_stride, int* fail, hipStream_t st) {
  const int nblk = np / CB;
  for (int k = 0; k < nblk; ++k) {
    double* linv = linv_ws + (size_t)k * linv_stride;
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, st, L, np, k, linv, fail);
    const int below = nblk - k - 1;
    if (below > 0) {
      hipLaunchKernelGGL(chol_trsm_kernel, dim3(below), dim3(256), 0, st, L, np, k, linv);
      hipLaunchKernelGGL(chol_update_kernel, dim3(below * (below + 1) / 2), dim3(256), 0, st, L, np, k);
    }
  }
  PTD_CHECK_LAUNCH("cholesky_f64");
  return PTD_OK;
}

}  // namespace ptd
