// Inverse Cholesky factor of a 64 x 64 symmetric positive definite tile held in LDS -- shared by the Cholesky sweep of
// the filtered eigensolver (eigh_filtered.hip) and the blocked Cholesky factorisation (chol.hip).
#pragma once

#include "common.h"

namespace ptd {

namespace {

constexpr int FB = 64;          // tile size of the Cholesky factorisation / triangular inverse
constexpr int FQ = 65;
__device__ __forceinline__ double fs_rcp(double d) {
  double p = __builtin_amdgcn_rcp(d);
  p = fma(fma(-d, p, 1.0), p, p);
  p = fma(fma(-d, p, 1.0), p, p);
  return p;
}
// ---- the same result, hierarchically (round 4): 16 x 16 leaves on ONE wave, everything else on the matrix cores
// The tile is cut into 4 x 4 blocks of 16 x 16.  Blocked right-looking Cholesky: the leaf forms Linv_bb of the current
// diagonal block, the panel L_ib = A_ib Linv_bb^T and the trailing update A_ij -= L_ib L_jb^T are 16 x 16 x 16 products
// (four v_mfma_f64_16x16x4_f64 each).  Then X = L^-1 by block columns, wave j owning column j:
// X_jj = Linv_jj, X_ij = -Linv_ii sum_{k=j}^{i-1} L_ik X_kj.
// The leaf keeps the block in the registers of one wave -- lane 4 r + q holds row r, columns q, q + 4, q + 8, q + 12, the
// full symmetric row as in fs_chol_inv_tile above (A[r][c] until column c is eliminated, X[r][c] afterwards) -- and a
// column step is: row j's entries of the lane's column class, the pivot and the lane's own entry of column j fetched
// with cross-lane moves (ds_bpermute: no LDS round trip, no barrier), one reciprocal, four fmas.  64 columns are 64
// such steps of ~0.1 us instead of 32 barrier-separated two-column steps of ~1.1 us.
constexpr int LP = 17;          // LDS pitch of a 16 x 16 block
// lout != nullptr: the Cholesky factor L of the block is written there too (row-major, pitch ldl, zeros above the
// diagonal; it may be the block A itself: every lane has its row in registers before the first store)
__device__ __forceinline__ bool fs_leaf16(const double* __restrict__ A, int lda, double* __restrict__ out, int lane,
                                          double* lout = nullptr, int ldl = 0) {
  const int r = lane >> 2, q = lane & 3;
  double y[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = q + 4 * s;
    y[s] = (c <= r) ? A[r * lda + c] : A[c * lda + r];
  }
  bool bad = false;
  double myd = 1.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int jq = j & 3, js = j >> 2;
    double P[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) P[s] = __shfl(y[s], 4 * j + q);      // row j, this lane's column class
    const double d = __shfl(y[js], 4 * j + jq);                       // the pivot A[j][j]
    const double a = __shfl(y[js], (lane & ~3) | jq);                 // this row's entry of column j
    const bool ok = d > 0.0 && d < INFINITY;
    const double p = ok ? fs_rcp(d) : 1.0;
    const double m = (r > j) ? a * p : 0.0;
    bad = bad || !ok;
    myd = (r == j && ok) ? d : myd;
#pragma unroll
    for (int s = 0; s < 4; ++s) y[s] = fma(-m, P[s], y[s]);
    if (q == jq) {
      // L[r][j] = A[r][j] / sqrt(d) = m sqrt(d): the lanes that own column j write it (0 above the diagonal)
      if (lout) { const double sd = sqrt(d); lout[r * ldl + j] = (r > j) ? m * sd : (r == j ? sd : 0.0); }
      y[js] = (r == j) ? 1.0 : -m;                                    // the slot changes hands: X[r][j] = -Lhat[r][j]
    }
  }
  const double rs = 1.0 / sqrt(myd);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = q + 4 * s;
    out[r * LP + c] = (c <= r) ? y[s] * rs : 0.0;
  }
  return bad;
}
// acc += A B^T / acc += A B for 16 x 16 row-major LDS blocks (one wave); C/D map: col = lane & 15, row = (lane >> 4) + 4 reg
__device__ __forceinline__ f64x4 fs_blk_abt(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                            int lane, f64x4 acc) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 16; kk += 4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l15 * lda + kk + l4], B[l15 * ldb + kk + l4], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f64x4 fs_blk_ab(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                           int lane, f64x4 acc) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 16; kk += 4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l15 * lda + kk + l4], B[(kk + l4) * ldb + l15], acc, 0, 0, 0);
  return acc;
}
// T: the tile in LDS (pitch FQ, lower triangle read, destroyed); Dv: 4 x 16 x LP, Sc: 4 x 16 x LP, Xs: 64 x FQ doubles of
// LDS; out: L^-1, 64 x 64 row-major in memory (zeros above the diagonal).  256 threads.
// KEEP_L: T holds L afterwards (lower triangle incl. the diagonal blocks, zeros above the diagonal inside them).
template <bool KEEP_L = false>
__device__ __forceinline__ void fs_chol_inv_tile2(double* __restrict__ T, double* __restrict__ Dv, double* __restrict__ Sc,
                                                  double* __restrict__ Xs, double* __restrict__ out,
                                                  int* __restrict__ fail, int tid) {
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  bool bad = false;
#pragma unroll 1
  for (int b = 0; b < 4; ++b) {
    if (w == 0) {
      double* Abb = T + (16 * b) * FQ + 16 * b;
      bad = fs_leaf16(Abb, FQ, Dv + b * 16 * LP, lane, KEEP_L ? Abb : nullptr, FQ) || bad;
    }
    __syncthreads();
    if (w < 3 - b) {                                       // panel: L_ib = A_ib Linv_bb^T over A_ib
      double* Aib = T + 16 * (b + 1 + w) * FQ + 16 * b;
      const f64x4 acc = fs_blk_abt(Aib, FQ, Dv + b * 16 * LP, LP, lane, f64x4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
      for (int q = 0; q < 4; ++q) Aib[(l4 + 4 * q) * FQ + l15] = acc[q];
    }
    __syncthreads();
    const int nt = 3 - b, cnt = nt * (nt + 1) / 2;         // trailing blocks (i, j), b < j <= i; (b+1, b+1) first
    for (int t = w; t < cnt; t += 4) {
      int ii = 0, jj = t;
      while (jj > ii) { jj -= ii + 1; ++ii; }
      const int i = b + 1 + ii, j = b + 1 + jj;
      const f64x4 acc = fs_blk_abt(T + 16 * i * FQ + 16 * b, FQ, T + 16 * j * FQ + 16 * b, FQ, lane,
                                   f64x4{0.0, 0.0, 0.0, 0.0});
      double* Aij = T + 16 * i * FQ + 16 * j;
#pragma unroll
      for (int q = 0; q < 4; ++q) Aij[(l4 + 4 * q) * FQ + l15] -= acc[q];
    }
    __syncthreads();
  }
  {                                                        // X = L^-1: wave j owns block column j
    const int j = w;
    for (int e = lane; e < 256; e += 64) Xs[(16 * j + (e >> 4)) * FQ + 16 * j + (e & 15)] = Dv[j * 16 * LP + (e >> 4) * LP + (e & 15)];
    double* S = Sc + w * 16 * LP;
    for (int i = j + 1; i < 4; ++i) {
      f64x4 s = f64x4{0.0, 0.0, 0.0, 0.0};
      for (int k = j; k < i; ++k) s = fs_blk_ab(T + 16 * i * FQ + 16 * k, FQ, Xs + 16 * k * FQ + 16 * j, FQ, lane, s);
#pragma unroll
      for (int q = 0; q < 4; ++q) S[(l4 + 4 * q) * LP + l15] = s[q];
      const f64x4 x = fs_blk_ab(Dv + i * 16 * LP, LP, S, LP, lane, f64x4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
      for (int q = 0; q < 4; ++q) Xs[(16 * i + l4 + 4 * q) * FQ + 16 * j + l15] = -x[q];
    }
  }
  __syncthreads();
  for (int e = tid; e < FB * FB; e += 256) {
    const int rr = e >> 6, cc = e & 63;
    out[e] = ((cc >> 4) <= (rr >> 4)) ? Xs[rr * FQ + cc] : 0.0;
  }
  if (bad && tid == 0) atomicExch(fail, 1);
}


}  // namespace

}  // namespace ptd
