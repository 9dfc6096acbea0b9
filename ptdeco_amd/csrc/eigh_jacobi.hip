// Symmetric eigendecomposition (f64) for the damped covariance matrices of
// dwain / falor:  one-sided block Jacobi on the gfx950 matrix cores
// (v_mfma_f64_16x16x4_f64).
//
// The rows g_i of G (initially A itself) are rotated until they are mutually
// orthogonal: G_final = W^T A with W orthogonal and G G^T diagonal, hence
// W^T A^2 W is diagonal, the rows of G_final are lambda_i * w_i^T, and for a
// positive semi-definite A   lambda_i = |g_i|,  w_i = g_i / |g_i|.
// Rows (not columns) are used because A is symmetric and rows are contiguous.
//
// Rows are grouped into blocks of 32.  A sweep visits every pair of blocks once,
// in (nb - 1) rounds of nb / 2 disjoint pairs (round-robin tournament); per round
// three kernels run over all pairs in parallel:
//   gram    S_p = R R^T for the 64 rows R of the pair, split over column chunks   [MFMA]
//   inner   sums the chunks, diagonalises S_p = Q Theta Q^T with a cyclic two-sided
//           Jacobi held in LDS (63 rounds of 32 disjoint rotations per inner sweep)
//   update  R <- Q^T R, streamed in 64-column tiles                              [MFMA]
// A pair whose 64 rows are already orthogonal to working precision is skipped.
// The sweep loop ends when a full sweep applied no rotation.
//
// Memory: G is n_pad x n_pad f64 (n_pad = n rounded up to 64), zero padded; a
// padded row never rotates (its inner products are exactly 0) and is dropped at
// the end by index.  At n = 4096 G is 134 MB and stays resident in the 256 MB
// Infinity Cache between the three kernels of a round.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace ptd {

int cholesky_f64(double* L, int np, double* linv_ws, int linv_stride, int* fail, hipStream_t st);  // chol.hip

namespace {

constexpr int JB = 32;        // rows per block
constexpr int JP = 2 * JB;    // rows per pair
constexpr int GRAM_KC = 32;   // columns per LDS tile in the gram kernel
constexpr int GRAM_PITCH = GRAM_KC + 2;
constexpr int UPD_PITCH = 66;

// Round-robin tournament: the two members of slot `k` in round `r` of a
// tournament over `m` players (m even).  Every unordered pair meets exactly once
// over rounds 0 .. m-2.
__host__ __device__ __forceinline__ void rr_pair(int m, int r, int k, int& a, int& b) {
  const int q = m - 1;
  if (k == 0) {
    a = q;
    b = r % q;
  } else {
    a = (r + k) % q;
    b = (r - k + q) % q;
  }
}

__device__ __forceinline__ int pair_row(int blkI, int blkJ, int r) {
  return r < JB ? blkI * JB + r : blkJ * JB + (r - JB);
}

// ---------------------------------------------------------------------------
// setup: symmetric permutation by descending diagonal (largest rows first, de Rijk), the
// padded working copy, and G0 = L^T after the Cholesky factorisation.
__global__ void jac_diag_kernel(const double* __restrict__ A, int64_t lda, int n, double* __restrict__ diag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) diag[i] = A[(int64_t)i * lda + i];
}

__global__ void jac_perm_kernel(const double* __restrict__ diag, int n, int* __restrict__ perm) {
  __shared__ double tile[256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double vi = i < n ? diag[i] : 0.0;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    __syncthreads();
    tile[threadIdx.x] = (j0 + (int)threadIdx.x < n) ? diag[j0 + threadIdx.x] : -INFINITY;
    __syncthreads();
    const int lim = min(256, n - j0);
    for (int jj = 0; jj < lim; ++jj) {
      const double vj = tile[jj];
      rank += (vj > vi) || (vj == vi && j0 + jj < i);
    }
  }
  if (i < n) perm[rank] = i;  // position `rank` of the permuted matrix holds original index i
}

// B[i][j] = A[perm[i]][perm[j]] for i, j < n; identity on the padding
__global__ void jac_gather_kernel(const double* __restrict__ A, int64_t lda, int n, const int* __restrict__ perm,
                                  double* __restrict__ B, int np) {
  const int64_t total = (int64_t)np * np;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / np), j = (int)(e % np);
    B[e] = (i < n && j < n) ? A[(int64_t)perm[i] * lda + perm[j]] : (i == j ? 1.0 : 0.0);
  }
}

// G = L^T (upper triangular): G[r][c] = c >= r ? L[c][r] : 0
__global__ void jac_transpose_kernel(const double* __restrict__ L, double* __restrict__ G, int np) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x, by = blockIdx.y;  // output tile: rows by*32.., cols bx*32..
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (bx >= by) {
    for (int rr = ty; rr < 32; rr += 8) tile[rr][tx] = L[(int64_t)(bx * 32 + rr) * np + by * 32 + tx];
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = by * 32 + rr, c = bx * 32 + tx;
    G[(int64_t)r * np + c] = (bx >= by && c >= r) ? tile[tx][rr] : 0.0;
  }
}

// ---------------------------------------------------------------------------
// gram: grid (ksplit, pairs).  Spart[pair][split] = R[:, cols] R[:, cols]^T  (64 x 64)
__global__ __launch_bounds__(256) void jac_gram_kernel(const double* __restrict__ G, int np, int nb, int round,
                                                       int kcols, double* __restrict__ Spart) {
  __shared__ __attribute__((aligned(16))) double T[JP * GRAM_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int pair = blockIdx.y, split = blockIdx.x, nsplit = gridDim.x;
  int bi, bj;
  rr_pair(nb, round, pair, bi, bj);
  const int c0 = split * kcols;
  const int ntile = kcols / GRAM_KC;

  // this thread's 4 x (2 doubles) of a 64 x 32 tile: row = idx >> 4, col pair = idx & 15
  const double* src[4];
  int dst[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = tid + 256 * p;
    const int row = idx >> 4, c2 = (idx & 15) * 2;
    src[p] = G + (int64_t)pair_row(bi, bj, row) * np + c0 + c2;
    dst[p] = row * GRAM_PITCH + c2;
  }
  double2 reg[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) reg[p] = *reinterpret_cast<const double2*>(src[p]);

  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  const int l15 = lane & 15, l4 = lane >> 4;
  for (int t = 0; t < ntile; ++t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<double2*>(&T[dst[p]]) = reg[p];
    __syncthreads();
    if (t + 1 < ntile) {
#pragma unroll
      for (int p = 0; p < 4; ++p) reg[p] = *reinterpret_cast<const double2*>(src[p] + (t + 1) * GRAM_KC);
    }
#pragma unroll
    for (int kk = 0; kk < GRAM_KC; kk += 4) {
      // A[m][k] = R[m][k];  B[k][n] = R[n][k]: both operands read "row on lane & 15, k on lane >> 4"
      const double a0 = T[(wr * 32 + l15) * GRAM_PITCH + kk + l4];
      const double a1 = T[(wr * 32 + 16 + l15) * GRAM_PITCH + kk + l4];
      const double b0 = T[(wc * 32 + l15) * GRAM_PITCH + kk + l4];
      const double b1 = T[(wc * 32 + 16 + l15) * GRAM_PITCH + kk + l4];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // f64 16x16x4 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* out = Spart + ((int64_t)pair * nsplit + split) * (JP * JP);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * 32 + i * 16 + l4 + 4 * r;
        const int col = wc * 32 + j * 16 + l15;
        out[row * JP + col] = acc[i][j][r];
      }
}

// ---------------------------------------------------------------------------
// inner: one workgroup per pair.  S = sum of the gram chunks; if its scaled
// off-diagonal part is below tol the pair is skipped, else S is diagonalised by a
// cyclic Jacobi in LDS and the accumulated rotations Q (S = Q Theta Q^T) are stored.
constexpr int IN_T = 512;        // threads
constexpr int SP = JP + 1;       // LDS pitch of S and Q

__device__ __forceinline__ void atomic_max_pos_double(double* addr, double v) {
  // non-negative doubles order like their bit patterns
  atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

__global__ __launch_bounds__(IN_T) void jac_inner_kernel(const double* __restrict__ Spart, int nsplit, double tol,
                                                         int max_inner_sweeps, int cross_only,
                                                         double* __restrict__ Qout, int* __restrict__ skip,
                                                         double* __restrict__ conv) {
  __shared__ double S[JP * SP];
  __shared__ double Q[JP * SP];
  __shared__ double rc[JB], rs[JB];
  __shared__ int rp[JB], rq[JB];
  __shared__ double red[IN_T / 64];
  __shared__ int rotated;

  const int tid = threadIdx.x;
  const int pair = blockIdx.x;
  const double* sp = Spart + (int64_t)pair * nsplit * (JP * JP);
  for (int e = tid; e < JP * JP; e += IN_T) {
    double v = 0.0;
    for (int s = 0; s < nsplit; ++s) v += sp[(int64_t)s * (JP * JP) + e];
    const int i = e >> 6, j = e & 63;
    S[i * SP + j] = v;
    Q[i * SP + j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  // symmetrise (the two triangles come from different MFMA tiles but are sums of the
  // same products in the same order; this only guards against future changes)
  for (int e = tid; e < JP * JP; e += IN_T) {
    const int i = e >> 6, j = e & 63;
    if (i < j) {
      const double v = 0.5 * (S[i * SP + j] + S[j * SP + i]);
      S[i * SP + j] = v;
      S[j * SP + i] = v;
    }
  }
  __syncthreads();

  // scaled off-diagonal measure of the incoming S
  double off = 0.0;
  for (int e = tid; e < JP * JP; e += IN_T) {
    const int i = e >> 6, j = e & 63;
    if (i < j) {
      const double d = S[i * SP + i] * S[j * SP + j];
      const double v = fabs(S[i * SP + j]);
      if (v > 0.0) off = fmax(off, d > 0.0 ? v / sqrt(d) : 1.0);
    }
  }
  for (int o = 32; o > 0; o >>= 1) off = fmax(off, __shfl_xor(off, o));
  if ((tid & 63) == 0) red[tid >> 6] = off;
  __syncthreads();
  off = 0.0;
  for (int w = 0; w < IN_T / 64; ++w) off = fmax(off, red[w]);
  if (tid == 0) {
    atomic_max_pos_double(conv, off);
    skip[pair] = (off <= tol) ? 1 : 0;
  }
  if (off <= tol) return;

  for (int sweep = 0; sweep < max_inner_sweeps; ++sweep) {
    if (tid == 0) rotated = 0;
    __syncthreads();
    // cross_only: the two 32-row blocks are each internally orthogonal already (they were fully
    // treated at round 0 of this outer sweep), so only the 32 x 32 cross pairs are rotated:
    // 32 rounds of pairs (k, 32 + (k + round) % 32) instead of the 63-round tournament.
    const int nrounds = cross_only ? JB : JP - 1;
    for (int round = 0; round < nrounds; ++round) {
      if (tid < JB) {
        int p, q;
        if (cross_only) {
          p = tid;
          q = JB + ((tid + round) & (JB - 1));
        } else {
          rr_pair(JP, round, tid, p, q);
        }
        const double app = S[p * SP + p], aqq = S[q * SP + q], apq = S[p * SP + q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > tol * sqrt(fabs(app * aqq)) && apq != 0.0) {
          const double tau = (aqq - app) / (2.0 * apq);
          const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + t * t);
          s = t * c;
          rotated = 1;
        }
        rp[tid] = p; rq[tid] = q; rc[tid] = c; rs[tid] = s;
      }
      __syncthreads();
      // S <- J^T S J on the 32 x 32 grid of 2 x 2 blocks, J_k = [[c, s], [-s, c]] on (p_k, q_k)
      for (int blk = tid; blk < JB * JB; blk += IN_T) {
        const int ka = blk >> 5, kb = blk & 31;
        const int pa = rp[ka], qa = rq[ka], pb = rp[kb], qb = rq[kb];
        const double ca = rc[ka], sa = rs[ka], cb = rc[kb], sb = rs[kb];
        const double x11 = S[pa * SP + pb], x12 = S[pa * SP + qb];
        const double x21 = S[qa * SP + pb], x22 = S[qa * SP + qb];
        // columns: [p', q'] = [c p - s q, s p + c q]
        const double y11 = cb * x11 - sb * x12, y12 = sb * x11 + cb * x12;
        const double y21 = cb * x21 - sb * x22, y22 = sb * x21 + cb * x22;
        // rows; the rotated pair's own off-diagonal is zero by construction: write it exactly
        const bool diag = (ka == kb) && (sa != 0.0);
        S[pa * SP + pb] = ca * y11 - sa * y21;
        S[pa * SP + qb] = diag ? 0.0 : ca * y12 - sa * y22;
        S[qa * SP + pb] = diag ? 0.0 : sa * y11 + ca * y21;
        S[qa * SP + qb] = sa * y12 + ca * y22;
      }
      // Q <- Q J
      for (int it = tid; it < JP * JB; it += IN_T) {
        const int row = it >> 5, k = it & 31;
        const int p = rp[k], q = rq[k];
        const double c = rc[k], s = rs[k];
        const double xp = Q[row * SP + p], xq = Q[row * SP + q];
        Q[row * SP + p] = c * xp - s * xq;
        Q[row * SP + q] = s * xp + c * xq;
      }
      __syncthreads();
    }
    const int any = rotated;
    __syncthreads();
    if (!any) break;
  }

  double* qo = Qout + (int64_t)pair * (JP * JP);
  for (int e = tid; e < JP * JP; e += IN_T) qo[e] = Q[(e >> 6) * SP + (e & 63)];
}

// ---------------------------------------------------------------------------
// update: grid (chunks, pairs).  R[:, cols] <- Q^T R[:, cols] in 64-column tiles.
__global__ __launch_bounds__(256) void jac_update_kernel(double* __restrict__ G, int np, int nb, int round,
                                                         int tiles_per_wg, const double* __restrict__ Qall,
                                                         const int* __restrict__ skip) {
  __shared__ __attribute__((aligned(16))) double Qs[JP * UPD_PITCH];
  __shared__ __attribute__((aligned(16))) double Rs[JP * UPD_PITCH];
  const int pair = blockIdx.y;
  if (skip[pair]) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  int bi, bj;
  rr_pair(nb, round, pair, bi, bj);

  const double* qg = Qall + (int64_t)pair * (JP * JP);
  // this thread's 8 x (2 doubles) of a 64 x 64 tile: row = idx >> 5, col pair = idx & 31
  int lo[8];
  int64_t go[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int idx = tid + 256 * p;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    lo[p] = row * UPD_PITCH + c2;
    go[p] = (int64_t)pair_row(bi, bj, row) * np + c2;
    *reinterpret_cast<double2*>(&Qs[lo[p]]) = *reinterpret_cast<const double2*>(qg + row * JP + c2);
  }

  const int tile0 = blockIdx.x * tiles_per_wg;
  const int ntile = min(tiles_per_wg, np / JP - tile0);
  double2 reg[8];
  if (ntile > 0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) reg[p] = *reinterpret_cast<const double2*>(G + go[p] + (int64_t)tile0 * JP);
  }
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int t = 0; t < ntile; ++t) {
    const int64_t cbase = (int64_t)(tile0 + t) * JP;
    __syncthreads();  // previous tile's MFMA reads of Rs are done
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<double2*>(&Rs[lo[p]]) = reg[p];
    __syncthreads();
    if (t + 1 < ntile) {
#pragma unroll
      for (int p = 0; p < 8; ++p) reg[p] = *reinterpret_cast<const double2*>(G + go[p] + cbase + JP);
    }
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < JP; kk += 4) {
      // out[m][c] = sum_k Q[k][m] R[k][c]:  A[m][k] = Q[k][m],  B[k][c] = R[k][c]
      const double a0 = Qs[(kk + l4) * UPD_PITCH + wr * 32 + l15];
      const double a1 = Qs[(kk + l4) * UPD_PITCH + wr * 32 + 16 + l15];
      const double b0 = Rs[(kk + l4) * UPD_PITCH + wc * 32 + l15];
      const double b1 = Rs[(kk + l4) * UPD_PITCH + wc * 32 + 16 + l15];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wr * 32 + i * 16 + l4 + 4 * r;
          const int col = wc * 32 + j * 16 + l15;
          G[(int64_t)pair_row(bi, bj, row) * np + cbase + col] = acc[i][j][r];
        }
  }
}

// ---------------------------------------------------------------------------
// post-processing: eigenvalue = row norm, ascending order, eigenvectors to columns
__global__ void jac_norm_kernel(const double* __restrict__ G, int np, int n, double* __restrict__ norms,
                                double* __restrict__ lambdas, int squared) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int j = lane; j < np; j += 64) {
    const double v = G[(int64_t)row * np + j];
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    norms[row] = sqrt(s);
    lambdas[row] = squared ? s : sqrt(s);  // rows of L^T carry sqrt(lambda), rows of A carry lambda
  }
}

__global__ void jac_rank_kernel(const double* __restrict__ norms, const double* __restrict__ lambdas, int n,
                                int* __restrict__ inv, double* __restrict__ evals) {
  __shared__ double tile[256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double vi = i < n ? norms[i] : 0.0;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    __syncthreads();
    tile[threadIdx.x] = (j0 + (int)threadIdx.x < n) ? norms[j0 + threadIdx.x] : INFINITY;
    __syncthreads();
    const int lim = min(256, n - j0);
    for (int jj = 0; jj < lim; ++jj) {
      const double vj = tile[jj];
      rank += (vj < vi) || (vj == vi && j0 + jj < i);
    }
  }
  if (i < n) {
    inv[rank] = i;
    evals[rank] = lambdas[i];
  }
}

// evecs[perm[c]][k] = G[inv[k]][c] / norm(inv[k]); 32 x 32 tiles through LDS
__global__ void jac_scatter_kernel(const double* __restrict__ G, int np, int n, int kvec,
                                   const int* __restrict__ inv, const double* __restrict__ norms,
                                   const int* __restrict__ perm, double* __restrict__ evecs, int64_t ldv) {
  __shared__ double tile[32][33];
  const int koff = n - kvec;  // output column kk holds the eigenvector of sorted index koff + kk
  const int k0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int kk = ty; kk < 32; kk += 8) {
    const int k = k0 + kk, c = c0 + tx;
    double v = 0.0;
    if (k < kvec && c < n) {
      const int src = inv[koff + k];
      const double nr = norms[src];
      v = nr > 0.0 ? G[(int64_t)src * np + c] / nr : (c == src ? 1.0 : 0.0);
    }
    tile[kk][tx] = v;
  }
  __syncthreads();
  for (int cc = ty; cc < 32; cc += 8) {
    const int c = c0 + cc, k = k0 + tx;
    if (c < n && k < kvec) evecs[(int64_t)perm[c] * ldv + k] = tile[tx][cc];
  }
}

struct JacobiPlan {
  int np, nb, pairs, ksplit, kcols, chunks, tiles_per_wg;
  size_t off_G, off_L, off_S, off_Q, off_skip, off_conv, off_norms, off_lam, off_inv, off_perm, off_linv, total;
};

JacobiPlan make_plan(int64_t n) {
  JacobiPlan p{};
  p.np = (int)align_up((size_t)std::max<int64_t>(n, 1), JP);
  p.nb = p.np / JB;
  p.pairs = p.nb / 2;
  // gram: pairs x ksplit workgroups, aim for >= 512; each split is a multiple of GRAM_KC columns
  const int max_split = p.np / GRAM_KC;
  int ks = 1;
  while (ks * 2 <= max_split && p.pairs * ks < 512 && (p.np / GRAM_KC) % (ks * 2) == 0) ks *= 2;
  p.ksplit = ks;
  p.kcols = p.np / ks;
  // update: pairs x chunks workgroups over np / 64 column tiles
  const int tiles = p.np / JP;
  int ch = 1;
  while (ch * 2 <= tiles && p.pairs * ch < 512) ch *= 2;
  p.chunks = ch;
  p.tiles_per_wg = (int)ceil_div(tiles, ch);
  p.chunks = (int)ceil_div(tiles, p.tiles_per_wg);
  size_t o = 0;
  p.off_G = o; o += align_up((size_t)p.np * p.np * 8, 256);
  p.off_L = o; o += align_up((size_t)p.np * p.np * 8, 256);
  p.off_S = o; o += align_up((size_t)p.pairs * p.ksplit * JP * JP * 8, 256);
  p.off_Q = o; o += align_up((size_t)p.pairs * JP * JP * 8, 256);
  p.off_skip = o; o += align_up((size_t)p.pairs * 4, 256);
  p.off_conv = o; o += 256;  // [0] convergence measure (f64), [8] cholesky failure flag (int)
  p.off_norms = o; o += align_up((size_t)p.np * 8, 256);
  p.off_lam = o; o += align_up((size_t)p.np * 8, 256);
  p.off_inv = o; o += align_up((size_t)p.np * 4, 256);
  p.off_perm = o; o += align_up((size_t)p.np * 4, 256);
  p.off_linv = o; o += 64 * 64 * 8;
  p.total = o;
  return p;
}

}  // namespace

size_t eigh_workspace_bytes(int64_t n) { return make_plan(n).total; }

int eigh_jacobi(const double* A, int64_t lda, int64_t n, int64_t k, double* evals, double* evecs, int64_t ldv,
                void* ws, size_t ws_bytes, int* sweeps_out, ptd_eigh_stats* stats, hipStream_t st) {
  PTD_REQUIRE(n >= 1 && n <= 32768, "ptd_eigh: n=%lld out of range [1, 32768]", (long long)n);
  PTD_REQUIRE(k >= 1 && k <= n, "ptd_eigh: k=%lld out of range [1, n]", (long long)k);
  PTD_REQUIRE(lda >= n && ldv >= k, "ptd_eigh: leading dimension too small");
  PTD_REQUIRE(A && evals && evecs && ws, "ptd_eigh: null pointer");
  PTD_REQUIRE(aligned16(ws), "ptd_eigh: workspace must be 16-byte aligned");
  const JacobiPlan p = make_plan(n);
  if (ws_bytes < p.total) {
    set_error("ptd_eigh: workspace %zu < required %zu bytes", ws_bytes, p.total);
    return PTD_ERR_WORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  double* G = reinterpret_cast<double*>(base + p.off_G);
  double* Sp = reinterpret_cast<double*>(base + p.off_S);
  double* Q = reinterpret_cast<double*>(base + p.off_Q);
  int* skip = reinterpret_cast<int*>(base + p.off_skip);
  double* conv = reinterpret_cast<double*>(base + p.off_conv);
  double* norms = reinterpret_cast<double*>(base + p.off_norms);
  int* inv = reinterpret_cast<int*>(base + p.off_inv);

  const double tol = 2.0 * std::sqrt((double)p.np) * 2.220446049250313e-16;
  const int max_sweeps = 40;
  // A sweep whose largest scaled off-diagonal entry (measured before its rotations) is below
  // `quad` leaves entries of order quad^2 / relative gap behind (quadratic convergence of the
  // cyclic Jacobi method): far below the rounding noise of the f64 inner products themselves
  // (~ sqrt(n) eps), so no further "verification" sweep is run.
  const double quad = 1e-9;
  static const int inner_sweeps = getenv("PTD_JACOBI_INNER_SWEEPS") ? atoi(getenv("PTD_JACOBI_INNER_SWEEPS")) : 1;
  static const bool debug = getenv("PTD_JACOBI_DEBUG") != nullptr;
  static const bool cross_only = getenv("PTD_JACOBI_FULL_INNER") == nullptr;

  // optional per-phase timing: one event pair per launch, read back at the sweep's sync
  std::vector<hipEvent_t> ev;
  hipEvent_t ev_first = nullptr, ev_last = nullptr;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    ev.resize((size_t)4 * (p.nb - 1));
    for (auto& e : ev) PTD_CHECK_HIP(hipEventCreate(&e));
    PTD_CHECK_HIP(hipEventCreate(&ev_first));
    PTD_CHECK_HIP(hipEventCreate(&ev_last));
    PTD_CHECK_HIP(hipEventRecord(ev_first, st));
  }
  auto mark = [&](int r, int k) {
    if (stats) (void)hipEventRecord(ev[(size_t)4 * r + k], st);
  };

  double* Lb = reinterpret_cast<double*>(base + p.off_L);
  double* lambdas = reinterpret_cast<double*>(base + p.off_lam);
  int* perm = reinterpret_cast<int*>(base + p.off_perm);
  double* linv = reinterpret_cast<double*>(base + p.off_linv);
  int* fail = reinterpret_cast<int*>(base + p.off_conv + 8);
  static const bool no_chol = getenv("PTD_JACOBI_NO_CHOLESKY") != nullptr;

  // G0: Cholesky-preconditioned start (L^T of the diagonally sorted matrix), or the sorted
  // matrix itself when it is not numerically positive definite
  PTD_CHECK_HIP(hipMemsetAsync(conv, 0, 16, st));
  hipLaunchKernelGGL(jac_diag_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, A, lda, (int)n, norms);
  hipLaunchKernelGGL(jac_perm_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, norms, (int)n, perm);
  bool use_chol = !no_chol;
  if (use_chol) {
    hipLaunchKernelGGL(jac_gather_kernel, dim3(2048), dim3(256), 0, st, A, lda, (int)n, perm, Lb, p.np);
    PTD_CHECK_LAUNCH("jac_gather");
    int rc = cholesky_f64(Lb, p.np, linv, 0, fail, st);
    if (rc != PTD_OK) return rc;
    int h_fail = 0;
    PTD_CHECK_HIP(hipMemcpyAsync(&h_fail, fail, 4, hipMemcpyDeviceToHost, st));
    PTD_CHECK_HIP(hipStreamSynchronize(st));
    use_chol = (h_fail == 0);
    if (debug) fprintf(stderr, "[ptd_eigh] n=%lld cholesky %s\n", (long long)n, use_chol ? "ok" : "failed -> plain start");
  }
  if (use_chol) {
    hipLaunchKernelGGL(jac_transpose_kernel, dim3(p.np / 32, p.np / 32), dim3(256), 0, st, Lb, G, p.np);
  } else {
    hipLaunchKernelGGL(jac_gather_kernel, dim3(2048), dim3(256), 0, st, A, lda, (int)n, perm, G, p.np);
  }
  PTD_CHECK_LAUNCH("jac_setup");

  int sweeps = 0;
  bool converged = false;
  for (; sweeps < max_sweeps && !converged;) {
    PTD_CHECK_HIP(hipMemsetAsync(conv, 0, 8, st));
    for (int r = 0; r < p.nb - 1; ++r) {
      mark(r, 0);
      hipLaunchKernelGGL(jac_gram_kernel, dim3(p.ksplit, p.pairs), dim3(256), 0, st, G, p.np, p.nb, r, p.kcols,
                         Sp);
      mark(r, 1);
      hipLaunchKernelGGL(jac_inner_kernel, dim3(p.pairs), dim3(IN_T), 0, st, Sp, p.ksplit, tol, inner_sweeps,
                         (cross_only && r > 0) ? 1 : 0, Q, skip, conv);
      mark(r, 2);
      hipLaunchKernelGGL(jac_update_kernel, dim3(p.chunks, p.pairs), dim3(256), 0, st, G, p.np, p.nb, r,
                         p.tiles_per_wg, Q, skip);
      mark(r, 3);
    }
    PTD_CHECK_LAUNCH("jacobi sweep");
    double h_conv = 0.0;
    PTD_CHECK_HIP(hipMemcpyAsync(&h_conv, conv, 8, hipMemcpyDeviceToHost, st));
    PTD_CHECK_HIP(hipStreamSynchronize(st));
    ++sweeps;
    converged = (h_conv <= quad);
    if (debug) fprintf(stderr, "[ptd_eigh] n=%lld sweep %d max scaled off-diagonal %.3e\n", (long long)n, sweeps, h_conv);
    if (stats) {
      for (int r = 0; r < p.nb - 1; ++r)
        for (int k = 0; k < 3; ++k) {
          float ms = 0.f;
          (void)hipEventElapsedTime(&ms, ev[(size_t)4 * r + k], ev[(size_t)4 * r + k + 1]);
          stats->ms[k] += ms;
          stats->launches[k] += 1;
        }
      const double pair_flops = 2.0 * JP * JP * (double)p.np;  // one 64 x 64 x np product
      stats->work[0] += (double)(p.nb - 1) * p.pairs * pair_flops;
      stats->work[2] += (double)(p.nb - 1) * p.pairs * pair_flops;  // upper bound: skipped pairs do none
    }
  }
  if (stats) {
    stats->sweeps = sweeps;
    stats->method = 0;
    PTD_CHECK_HIP(hipEventRecord(ev_last, st));
    PTD_CHECK_HIP(hipEventSynchronize(ev_last));
    (void)hipEventElapsedTime(&stats->total_ms, ev_first, ev_last);
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipEventDestroy(ev_first);
    (void)hipEventDestroy(ev_last);
  }
  if (sweeps_out) *sweeps_out = sweeps;

  hipLaunchKernelGGL(jac_norm_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st, G, p.np, (int)n, norms,
                     lambdas, use_chol ? 1 : 0);
  hipLaunchKernelGGL(jac_rank_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, norms, lambdas, (int)n,
                     inv, evals);
  hipLaunchKernelGGL(jac_scatter_kernel, dim3((unsigned)ceil_div(k, 32), (unsigned)ceil_div(n, 32)), dim3(256), 0,
                     st, G, p.np, (int)n, (int)k, inv, norms, perm, evecs, ldv);
  PTD_CHECK_LAUNCH("jacobi post");
  if (!converged) {
    set_error("ptd_eigh: no convergence after %d sweeps", sweeps);
    return PTD_ERR_NOCONV;
  }
  return PTD_OK;
}

}  // namespace ptd
