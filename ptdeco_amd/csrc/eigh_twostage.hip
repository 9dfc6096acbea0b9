// Two-stage tridiagonalisation of a symmetric f64 matrix (n a multiple of 32):
//
//   stage 1  dense -> band (bandwidth TB = 32): per panel a tall-skinny QR by CholeskyQR2 + Householder
//            reconstruction (every step a Gram product, a 32 x 32 factorisation or a left multiplication by
//            a 32 x 32 matrix: no per-column launches), then the two-sided block update of the trailing
//            square as two f64-MFMA products (A22 V, and the rank-2*32 update A22 -= V X^T + X V^T);
//   stage 2  band -> tridiagonal by bulge chasing: one wave per sweep, the sweeps pipelined two block
//            positions apart over a band kept in L2 / Infinity Cache, hand-offs between waves through
//            per-sweep progress counters (sc1 stores / loads, bounded spins);
//   back     eigenvectors of T -> eigenvectors of A:  Z = Q1 (Q2 Y).  Q2 (the n^2 / 64 short reflectors of
//            stage 2) is applied with a column chunk of Y resident in LDS, Q1 (compact-WY panels) as f64
//            MFMA products.
//
// Why: the one-stage reduction (eigh_tridiag.hip) needs two dependent launches per column (8190 for
// n = 4096) and streams the trailing matrix once per column; here stage 1 touches it three times per 32
// columns on the matrix cores and the per-column work happens on a 2 MB band.
// The algorithm and its index conventions are prototyped line by line in tools/twostage_proto.py.
//
// Storage: everything "wide".  The panel of block column j0 is Pt = A[j0 : j0+32, r0 : n] (r0 = j0 + 32), the
// transpose of the tall panel by symmetry (both triangles of the trailing square are kept current).  Its
// reflectors V (m x 32, unit lower trapezoidal) are stored transposed IN PLACE over it: Vt[c][i] = V[i][c].
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "twostage.h"

namespace ptd {

int gemm_f64(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn, double* C,
             int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha, bool beta1, int ksplit, hipStream_t st);
int gemm_f64_pair(const double* A, const double* B, const double* A2, const double* B2, int64_t sam, int64_t sak,
                  int64_t sbk, int64_t sbn, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha,
                  double* row0_out, hipStream_t st);

namespace {

constexpr int TB = TS_BAND;       // bandwidth, panel width, reflector length of stage 2
constexpr int LDB = TS_LDBAND;    // band row [i][k], k = j - i + 2 TB (k = 2 TB is the diagonal)
constexpr int CH = 128;           // columns of a wide panel per workgroup
constexpr int SP = 33;            // LDS pitch of a 32 x 32 matrix

__device__ __forceinline__ double wave_sum64(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------------------ stage 1
// G += X[:, c0 : c0 + cw] Y[:, ...]^T for 32-row wide panels staged in LDS as [32][CH + 1]
__device__ __forceinline__ void gram_accumulate(const double* __restrict__ Xs, const double* __restrict__ Ys, int cw,
                                                double* __restrict__ G, int tid) {
  const int ti = tid >> 4, tj = tid & 15;  // 2 x 2 entries per thread: rows 2 ti.., columns 2 tj..
  double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
  const double* x0 = Xs + (2 * ti) * (CH + 1);
  const double* x1 = x0 + (CH + 1);
  const double* y0 = Ys + (2 * tj) * (CH + 1);
  const double* y1 = y0 + (CH + 1);
#pragma unroll 4
  for (int c = 0; c < cw; ++c) {
    const double xa = x0[c], xb = x1[c], ya = y0[c], yb = y1[c];
    a00 += xa * ya; a01 += xa * yb; a10 += xb * ya; a11 += xb * yb;
  }
  atomicAdd(&G[(2 * ti) * TB + 2 * tj], a00);
  atomicAdd(&G[(2 * ti) * TB + 2 * tj + 1], a01);
  atomicAdd(&G[(2 * ti + 1) * TB + 2 * tj], a10);
  atomicAdd(&G[(2 * ti + 1) * TB + 2 * tj + 1], a11);
}

// G (32 x 32, zeroed by the caller) += X Y^T over the m columns of two wide panels (X == Y allowed)
__global__ __launch_bounds__(256) void ts_gram_kernel(const double* __restrict__ X, const double* __restrict__ Y,
                                                      int64_t ld, int m, double* __restrict__ G) {
  extern __shared__ double sm[];
  double* Xs = sm;
  double* Ys = (X == Y) ? sm : sm + TB * (CH + 1);
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * CH, cw = min(CH, m - c0);
  for (int e = tid; e < TB * CH; e += 256) {
    const int r = e / CH, c = e % CH;
    const bool ok = c < cw;
    Xs[r * (CH + 1) + c] = ok ? X[(int64_t)r * ld + c0 + c] : 0.0;
    if (X != Y) Ys[r * (CH + 1) + c] = ok ? Y[(int64_t)r * ld + c0 + c] : 0.0;
  }
  __syncthreads();
  gram_accumulate(Xs, Ys, cw, G, tid);
}

// In-LDS factorisations of a 32 x 32 matrix held as S[32][SP], 256 threads.
// Cholesky (lower, in place in the lower triangle); returns false (to all threads) on a non-positive pivot.
__device__ bool chol32(double* S, int tid, int* flag_s) {
  if (tid == 0) *flag_s = 0;
  __syncthreads();
  for (int k = 0; k < TB; ++k) {
    if (tid == 0) {
      const double piv = S[k * SP + k];
      if (!(piv > 0.0)) { *flag_s = 1; S[k * SP + k] = 1.0; }
      else S[k * SP + k] = sqrt(piv);
    }
    __syncthreads();
    if (tid > k && tid < TB) S[tid * SP + k] /= S[k * SP + k];
    __syncthreads();
    // trailing update of the lower triangle: rows i > k, columns k < j <= i
    for (int e = tid; e < TB * TB; e += 256) {
      const int i = e >> 5, j = e & 31;
      if (j > k && i >= j) S[i * SP + j] -= S[i * SP + k] * S[j * SP + k];
    }
    __syncthreads();
  }
  return *flag_s == 0;
}

// X = L^-1 for a lower triangular L (unit_diag: the diagonal is taken as 1); one thread per column, result in
// the lower triangle of X[32][SP], zeros above
__device__ void tri_lower_inverse32(const double* L, double* X, bool unit_diag, int tid) {
  if (tid < TB) {
    const int j = tid;
    double x[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < TB; ++k)
        if (k < i) acc -= (k >= j ? L[i * SP + k] * x[k] : 0.0);
      x[i] = (i < j) ? 0.0 : (unit_diag ? acc : acc / L[i * SP + i]);
    }
#pragma unroll
    for (int i = 0; i < TB; ++i) X[i * SP + j] = x[i];
  }
}

// C = op(A) op(B) for 32 x 32 LDS matrices; element (i, k) of op(A) is A[i*sai + k*sak] etc.
__device__ __forceinline__ void mm32(const double* A, int sai, int sak, const double* B, int sbk, int sbj, double* C,
                                     double alpha, int tid) {
  for (int e = tid; e < TB * TB; e += 256) {
    const int i = e >> 5, j = e & 31;
    double acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < TB; ++k) acc += A[i * sai + k * sak] * B[k * sbk + j * sbj];
    C[i * SP + j] = alpha * acc;
  }
}

// K2: G -> L1 = chol(G) (lower) and L1inv; status[0] is set on breakdown
__global__ __launch_bounds__(256) void ts_chol_kernel(const double* __restrict__ G, double* __restrict__ L1,
                                                      double* __restrict__ L1inv, int* __restrict__ status) {
  __shared__ double S[TB * SP], X[TB * SP];
  __shared__ int flag;
  const int tid = threadIdx.x;
  for (int e = tid; e < TB * TB; e += 256) S[(e >> 5) * SP + (e & 31)] = G[e];
  __syncthreads();
  if (!chol32(S, tid, &flag) && tid == 0) atomicExch(status, 1);
  tri_lower_inverse32(S, X, false, tid);
  __syncthreads();
  for (int e = tid; e < TB * TB; e += 256) {
    const int i = e >> 5, j = e & 31;
    L1[e] = (j <= i) ? S[i * SP + j] : 0.0;
    L1inv[e] = X[i * SP + j];
  }
}

// Left multiplication of the columns [c_lo, m) of a wide panel by a 32 x 32 matrix Lm (row-major, read from
// memory into LDS):  X[:, c] <- Lm X[:, c].  One thread per column.  Optionally accumulates the Gram matrix of
// the result into G and zeroes the same columns of a second wide panel.
__global__ __launch_bounds__(256) void ts_lmul_kernel(const double* __restrict__ Lm, double* __restrict__ X,
                                                      int64_t ld, int c_lo, int m, double* __restrict__ G,
                                                      double* __restrict__ zero_out) {
  extern __shared__ double sm[];
  double* Ls = sm;                  // [32][32]
  double* Os = sm + TB * TB;        // [32][CH + 1] when G
  const int tid = threadIdx.x;
  for (int e = tid; e < TB * TB; e += 256) Ls[e] = Lm[e];
  __syncthreads();
  const int c0 = blockIdx.x * CH;
  const int c = c0 + tid;
  const bool act = tid < CH && c >= c_lo && c < m;
  double x[TB], o[TB];
  if (act) {
#pragma unroll
    for (int k = 0; k < TB; ++k) x[k] = X[(int64_t)k * ld + c];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < TB; ++k) acc += Ls[i * TB + k] * x[k];
      o[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < TB; ++i) X[(int64_t)i * ld + c] = o[i];
  }
  if (zero_out && tid < CH && c < m) {
#pragma unroll
    for (int i = 0; i < TB; ++i) zero_out[(int64_t)i * ld + c] = 0.0;
  }
  if (G) {
    if (tid < CH) {
#pragma unroll
      for (int i = 0; i < TB; ++i) Os[i * (CH + 1) + tid] = act ? o[i] : 0.0;
    }
    __syncthreads();
    gram_accumulate(Os, Os, CH, G, tid);
  }
}

// K4: second Cholesky pass + Householder reconstruction of the panel (single workgroup).
//   in : G2 = Q1t Q1t^T, L1 (first pass), Q1t_top = At[0:32, 0:32] (At = the wide panel, leading dimension ld)
//   out: Vt_top written over Q1t_top, MT (Vt_rest = MT Q1t_rest), T (32 x 32 upper, row-major), the band entries of
//        R~ = D R2 R1 (the sub-diagonal block of the band), status on breakdown
__global__ __launch_bounds__(256) void ts_hr_kernel(const double* __restrict__ G2, const double* __restrict__ L1,
                                                    double* __restrict__ At, int64_t ld, double* __restrict__ MT,
                                                    double* __restrict__ T, double* __restrict__ band_row0,
                                                    int* __restrict__ status) {
  __shared__ double S[TB * SP], R2i[TB * SP], W[TB * SP], X[TB * SP], Y[TB * SP], L1s[TB * SP];
  __shared__ double Dg[TB];
  __shared__ int flag;
  const int tid = threadIdx.x;
  for (int e = tid; e < TB * TB; e += 256) {
    S[(e >> 5) * SP + (e & 31)] = G2[e];
    L1s[(e >> 5) * SP + (e & 31)] = L1[e];
  }
  __syncthreads();
  if (!chol32(S, tid, &flag) && tid == 0) atomicExch(status, 1);   // S lower = L2, R2 = L2^T
  tri_lower_inverse32(S, R2i, false, tid);                          // R2i = L2^-1 ; R2^-1 = R2i^T
  __syncthreads();
  // Qtop[i][j] = sum_k Q1top[i][k] R2inv[k][j] = sum_k Q1t[k][i] L2inv[j][k]
  for (int e = tid; e < TB * TB; e += 256) X[(e >> 5) * SP + (e & 31)] = At[(int64_t)(e >> 5) * ld + (e & 31)];  // X[k][i] = Q1t[k][i]
  __syncthreads();
  mm32(X, 1, SP, R2i, 1, SP, W, 1.0, tid);   // op(A)(i,k) = X[k][i]; op(B)(k,j) = R2i[j][k]
  __syncthreads();
  // LU without pivoting of Qtop - D, D_ii = -sign(pivot candidate): |pivot| >= 1
  for (int i = 0; i < TB; ++i) {
    if (tid == 0) {
      const double piv = W[i * SP + i];
      const double dg = piv >= 0.0 ? -1.0 : 1.0;
      Dg[i] = dg;
      W[i * SP + i] = piv - dg;
    }
    __syncthreads();
    if (tid > i && tid < TB) W[tid * SP + i] /= W[i * SP + i];
    __syncthreads();
    for (int e = tid; e < TB * TB; e += 256) {
      const int r = e >> 5, c = e & 31;
      if (r > i && c > i) W[r * SP + c] -= W[r * SP + i] * W[i * SP + c];
    }
    __syncthreads();
  }
  // W = strict lower L (unit) + upper U.  Uinv^T: invert the lower triangular U^T
  for (int e = tid; e < TB * TB; e += 256) {
    const int i = e >> 5, j = e & 31;
    Y[i * SP + j] = (j <= i) ? W[j * SP + i] : 0.0;   // Y = U^T (lower)
  }
  __syncthreads();
  tri_lower_inverse32(Y, X, false, tid);               // X = (U^T)^-1 = (U^-1)^T
  __syncthreads();
  // M = R2inv Uinv ;  MT[j][i] = M[i][j] = sum_k R2inv[i][k] Uinv[k][j] = sum_k R2i[k][i] X[j][k]
  mm32(X, SP, 1, R2i, SP, 1, Y, 1.0, tid);             // Y[j][i] = sum_k X[j][k] R2i[k][i]  = MT
  __syncthreads();
  for (int e = tid; e < TB * TB; e += 256) MT[e] = Y[(e >> 5) * SP + (e & 31)];
  // Vt_top[c][i] = L[i][c]
  for (int e = tid; e < TB * TB; e += 256) {
    const int c = e >> 5, i = e & 31;
    At[(int64_t)c * ld + i] = (i > c) ? W[i * SP + c] : (i == c ? 1.0 : 0.0);
  }
  __syncthreads();
  // Linv of the unit lower L
  for (int e = tid; e < TB * TB; e += 256) {
    const int i = e >> 5, j = e & 31;
    Y[i * SP + j] = (j < i) ? W[i * SP + j] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  tri_lower_inverse32(Y, X, true, tid);                // X = L^-1
  __syncthreads();
  // T[i][j] = - sum_k U[i][k] D[k] Linv[j][k]
  for (int e = tid; e < TB * TB; e += 256) {
    const int i = e >> 5, j = e & 31;
    double acc = 0.0;
    for (int k = i; k <= j; ++k) acc += W[i * SP + k] * Dg[k] * X[j * SP + k];
    T[e] = (j >= i) ? -acc : 0.0;
  }
  // R~[i][j] = D[i] sum_k R2[i][k] R1[k][j] = D[i] sum_{k=i..j} L2[k][i] L1[j][k]   (j >= i), band entry
  // (r0 + i, j0 + j): k = j - i + TB of band row r0 + i
  for (int e = tid; e < TB * TB; e += 256) {
    const int i = e >> 5, j = e & 31;
    if (j < i) continue;
    double acc = 0.0;
    for (int k = i; k <= j; ++k) acc += S[k * SP + i] * L1s[j * SP + k];
    band_row0[(int64_t)i * LDB + (j - i + TB)] = Dg[i] * acc;
  }
}

// K7b: Xt[:, c] = T^T W0t[:, c] - 1/2 C Vt[:, c],  C = T^T Z0 T  (symmetric); one thread per column
__global__ __launch_bounds__(256) void ts_x_kernel(const double* __restrict__ T, const double* __restrict__ Z0,
                                                   const double* __restrict__ Vt, const double* __restrict__ W0t,
                                                   double* __restrict__ Xt, int64_t ld, int m) {
  __shared__ double Ts[TB * SP], Zs[TB * SP], Es[TB * SP], Cs[TB * SP];
  const int tid = threadIdx.x;
  for (int e = tid; e < TB * TB; e += 256) {
    Ts[(e >> 5) * SP + (e & 31)] = T[e];
    Zs[(e >> 5) * SP + (e & 31)] = Z0[e];
  }
  __syncthreads();
  mm32(Zs, SP, 1, Ts, SP, 1, Es, 1.0, tid);    // E = Z0 T
  __syncthreads();
  mm32(Ts, 1, SP, Es, SP, 1, Cs, 1.0, tid);    // C = T^T E
  __syncthreads();
  const int c = blockIdx.x * CH + tid;
  if (tid >= CH || c >= m) return;
  double w[TB], v[TB];
#pragma unroll
  for (int k = 0; k < TB; ++k) { w[k] = W0t[(int64_t)k * ld + c]; v[k] = Vt[(int64_t)k * ld + c]; }
#pragma unroll
  for (int i = 0; i < TB; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < TB; ++k) acc += Ts[k * SP + i] * w[k] - 0.5 * Cs[i * SP + k] * v[k];
    Xt[(int64_t)i * ld + c] = acc;
  }
}

// band[i][k] for the entries of the diagonal blocks (lower triangles); the sub-diagonal blocks were written by
// ts_hr_kernel.  One thread per (i, j) of a block.
__global__ __launch_bounds__(256) void ts_band_diag_kernel(const double* __restrict__ A, int64_t ld, int n,
                                                           double* __restrict__ band) {
  const int blk = blockIdx.x;
  for (int e = threadIdx.x; e < TB * TB; e += 256) {
    const int i = e >> 5, j = e & 31;
    const int gi = blk * TB + i, gj = blk * TB + j;
    if (j <= i && gi < n) band[(int64_t)gi * LDB + (gj - gi + 2 * TB)] = A[(int64_t)gi * ld + gj];
  }
}

// ------------------------------------------------------------------------------------------------ stage 2
__device__ __forceinline__ double ld_sc1(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int PROG_DONE = 1 << 30;

// Wait until sweep s - 1 has completed `need` tasks (or all of them).  Bounded: on a time-out (a worker that
// is not resident: the launch is sized to be co-resident) the failure flag is raised and every worker leaves.
__device__ __forceinline__ bool chase_wait(const int* prog, int s, int need, int* fail) {
  if (s == 0) return true;
  const int* f = prog + (s - 1);
  for (long spin = 0; spin < 40000000L; ++spin) {
    const int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= need) return true;
    if ((spin & 1023) == 1023 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// Householder vector of x (one entry per lane of the first half wave, zero beyond the length): returns v_i for the
// lane's row, tau and beta (uniform).  LAPACK dlarfg convention, v_0 = 1.
__device__ __forceinline__ void householder32(double x, int i, bool first_half, double& v, double& tau, double& beta) {
  const double alpha = __shfl(x, 0);
  const double sq = (first_half && i >= 1) ? x * x : 0.0;
  const double sigma = wave_sum64(sq);
  if (sigma == 0.0) {
    tau = 0.0; beta = alpha; v = (i == 0) ? 1.0 : 0.0;
  } else {
    const double nrm = sqrt(alpha * alpha + sigma);
    beta = alpha >= 0.0 ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    v = (i == 0) ? 1.0 : x * sc;
  }
  v = __shfl(v, i);  // both half waves hold row i: the second half takes it from the first
}

// Bulge chasing, one wave per sweep (tools/twostage_proto.py: band_to_tridiag).  Lane l = (row i = l & 31,
// half h = l >> 5) holds columns 16 h .. 16 h + 15 of row i of the 32 x 32 blocks.  All band accesses are
// agent-scope (sc1) 8-byte loads / stores: the band is handed from wave to wave through L2.
__global__ __launch_bounds__(64) void ts_chase_kernel(double* __restrict__ band, int n, int* __restrict__ prog,
                                                      int* __restrict__ fail, double* __restrict__ V2, int64_t ldv2,
                                                      double* __restrict__ tau2, int npos) {
  __shared__ double Bs[TB * SP];
  __shared__ double vb[TB], wb[TB], ub[TB];
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
#define BAND(gi, gj) band[(int64_t)(gi) * LDB + ((gj) - (gi) + 2 * TB)]
  for (int s = blockIdx.x; s < n - 2; s += gridDim.x) {
    int r = s + 1;
    int ln = min(TB, n - r);
    if (ln < 2) break;
    // ---- task 0: reflector from column s, two-sided on the diagonal block
    if (!chase_wait(prog, s, 2, fail)) return;
    asm volatile("" ::: "memory");
    double x = (h == 0 && i < ln) ? ld_sc1(&BAND(r + i, s)) : 0.0;
    double v, tau, beta;
    householder32(x, i, h == 0, v, tau, beta);
    if (i >= ln) v = 0.0;
    if (h == 0 && i < ln) st_sc1(&BAND(r + i, s), i == 0 ? beta : 0.0);
    int p = 0;
    for (;;) {
      // two-sided update of the diagonal block D (rows / columns r .. r + ln) with (v, tau)
      double Dv[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = 16 * h + jj;
        double val = 0.0;
        if (i < ln && j < ln) val = (j <= i) ? ld_sc1(&BAND(r + i, r + j)) : ld_sc1(&BAND(r + j, r + i));
        Dv[jj] = val;
      }
      if (h == 0) vb[i] = v;
      __syncthreads();
      double part = 0.0;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) part += Dv[jj] * vb[16 * h + jj];
      part += __shfl_xor(part, 32);
      double w = tau * part;
      const double wv = wave_sum64(h == 0 ? w * v : 0.0);
      w -= 0.5 * tau * wv * v;
      if (h == 0) wb[i] = w;
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = 16 * h + jj;
        Dv[jj] -= v * wb[j] + w * vb[j];
        if (j <= i && i < ln) st_sc1(&BAND(r + i, r + j), Dv[jj]);
      }
      if (h == 0 && i < ln) V2[(int64_t)s * ldv2 + r + i] = v;
      if (lane == 0) tau2[(int64_t)s * npos + p] = tau;
      // publish: every store of this task has left the wave, then the counter
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ++p;
      if (lane == 0) __hip_atomic_store(prog + s, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // ---- next block position
      const int c0 = r;
      r = c0 + ln;           // (ln == TB whenever another block follows)
      ln = min(TB, n - r);
      if (ln <= 0) break;
      if (!chase_wait(prog, s, p + 2, fail)) return;
      asm volatile("" ::: "memory");
      // B = A[r : r + ln, c0 : c0 + TB]: right-apply the previous reflector (v over the columns)
      double Bv[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = 16 * h + jj;
        Bv[jj] = (i < ln) ? ld_sc1(&BAND(r + i, c0 + j)) : 0.0;
      }
      // (vb still holds v indexed by the previous block's row = this block's column)
      part = 0.0;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) part += Bv[jj] * vb[16 * h + jj];
      part += __shfl_xor(part, 32);
      const double t = tau * part;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) Bv[jj] -= t * vb[16 * h + jj];
      // new reflector from the first column of B
      double vn, taun, betan;
      householder32(h == 0 ? Bv[0] : 0.0, i, h == 0, vn, taun, betan);
      if (i >= ln) vn = 0.0;
      if (ln < 2) { taun = 0.0; }
      // left-apply to the columns 1 .. of B: u_j = sum_i vn_i B_ij through an LDS transpose
      __syncthreads();   // vb reads above are done before it is overwritten below
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) Bs[i * SP + 16 * h + jj] = Bv[jj];
      if (h == 0) wb[i] = vn;
      __syncthreads();
      {
        const int j = i, g = h;   // lane -> column j, row half g
        double up = 0.0;
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) up += wb[16 * g + ii] * Bs[(16 * g + ii) * SP + j];
        up += __shfl_xor(up, 32);
        if (g == 0) ub[j] = up;
      }
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = 16 * h + jj;
        if (j == 0) Bv[jj] = (i == 0) ? betan : 0.0;   // H x = beta e1
        else Bv[jj] -= taun * vn * ub[j];
        if (i < ln) st_sc1(&BAND(r + i, c0 + j), Bv[jj]);
      }
      v = vn; tau = taun;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(prog + s, PROG_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#undef BAND
}

// d, e of the tridiagonal band
__global__ void ts_extract_tridiag_kernel(const double* __restrict__ band, int n, double* __restrict__ d,
                                          double* __restrict__ e) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    d[i] = band[(int64_t)i * LDB + 2 * TB];
    e[i] = (i + 1 < n) ? band[(int64_t)(i + 1) * LDB + 2 * TB - 1] : 0.0;
  }
}

// -------------------------------------------------------------------------------------- back-transformation
// Y[:, c0 : c0 + CQ] <- Q2 Y (the reflectors of stage 2 in reverse order of their generation), the column chunk
// resident in LDS as [CQ][n + 1].  Within a sweep the reflectors touch disjoint rows: wave w takes positions
// w, w + 8, ...; lane (i, cp) holds row i of the reflector and the column pair cp.
template <int CQ>
__global__ __launch_bounds__(512) void ts_apply_q2_kernel(const double* __restrict__ V2, int64_t ldv2,
                                                          const double* __restrict__ tau2, int npos, int n,
                                                          double* __restrict__ Y, int64_t ldy, int nvec) {
  extern __shared__ double Ys[];   // [CQ][n + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * CQ;
  const int pitch = n + 1;
  for (int e = tid; e < n * CQ; e += 512) {
    const int r = e / CQ, c = e % CQ;
    Ys[c * pitch + r] = (c0 + c < nvec) ? Y[(int64_t)r * ldy + c0 + c] : 0.0;
  }
  __syncthreads();
  constexpr int PAIRS = CQ / 2;           // column pairs; lanes beyond 32 * PAIRS idle (CQ = 2)
  const int i = lane & 31, cp = lane >> 5;
  const bool lane_on = cp < PAIRS;
  for (int s = n - 3; s >= 0; --s) {
    const int first = s + 1;
    const int np_s = (n - first + TB - 1) / TB;
    for (int p = wave; p < np_s; p += 8) {
      const int r = first + p * TB;
      const int ln = min(TB, n - r);
      const double tau = tau2[(int64_t)s * npos + p];
      const double v = (i < ln) ? V2[(int64_t)s * ldv2 + r + i] : 0.0;
      double y0 = 0.0, y1 = 0.0;
      if (lane_on && i < ln) { y0 = Ys[(2 * cp) * pitch + r + i]; y1 = Ys[(2 * cp + 1) * pitch + r + i]; }
      double d0 = v * y0, d1 = v * y1;
      for (int o = 16; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); }
      if (lane_on && i < ln) {
        Ys[(2 * cp) * pitch + r + i] = y0 - tau * v * d0;
        Ys[(2 * cp + 1) * pitch + r + i] = y1 - tau * v * d1;
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < n * CQ; e += 512) {
    const int r = e / CQ, c = e % CQ;
    if (c0 + c < nvec) Y[(int64_t)r * ldy + c0 + c] = Ys[c * pitch + r];
  }
}

// TVt_p = T_p Vt_p for every panel: one thread per column of a panel (blockIdx.y = panel)
__global__ __launch_bounds__(256) void ts_tv_kernel(const double* __restrict__ A, int64_t ld, int n,
                                                    const double* __restrict__ Tall, double* __restrict__ TV,
                                                    int64_t ldt) {
  __shared__ double Ts[TB * TB];
  const int p = blockIdx.y, tid = threadIdx.x;
  const int j0 = p * TB, r0 = j0 + TB, m = n - r0;
  const int c = blockIdx.x * 256 + tid;
  if (blockIdx.x * 256 >= m) return;
  for (int e = tid; e < TB * TB; e += 256) Ts[e] = Tall[(int64_t)p * TB * TB + e];
  __syncthreads();
  if (c >= m) return;
  const double* Vt = A + (int64_t)j0 * ld + r0;
  double* out = TV + (int64_t)j0 * ldt + r0;
  double v[TB];
#pragma unroll
  for (int k = 0; k < TB; ++k) v[k] = Vt[(int64_t)k * ld + c];
#pragma unroll
  for (int i = 0; i < TB; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < TB; ++k) acc += Ts[i * TB + k] * v[k];   // (T upper triangular: zeros below)
    out[(int64_t)i * ldt + c] = acc;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
TwoStagePlan twostage_plan(int64_t n, int64_t ld) {
  TwoStagePlan p{};
  p.n = (int)n;
  p.ld = ld;
  p.npanels = (int)(n / TB) - 1;
  p.npos = (int)(n / TB) + 1;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
  p.off_zero = o;
  p.off_G1 = take((size_t)std::max(p.npanels, 1) * TB * TB * 8);
  p.off_G2 = take((size_t)std::max(p.npanels, 1) * TB * TB * 8);
  p.off_Z0 = take((size_t)std::max(p.npanels, 1) * TB * TB * 8);
  p.off_band = take((size_t)(n + 1) * LDB * 8);
  p.off_prog = take((size_t)(n + 8) * 4);
  p.off_status = take(256);
  p.zero_bytes = o - p.off_zero;
  p.off_L1 = take((size_t)TB * TB * 8);
  p.off_L1inv = take((size_t)TB * TB * 8);
  p.off_MT = take((size_t)TB * TB * 8);
  p.off_T = take((size_t)std::max(p.npanels, 1) * TB * TB * 8);
  p.off_W0t = take((size_t)TB * ld * 8);
  p.off_Xt = take((size_t)TB * ld * 8);
  p.off_tau2 = take((size_t)n * p.npos * 8);
  p.off_W2 = take((size_t)TB * n * 8);
  p.total = o;
  return p;
}

bool twostage_supported(int64_t n) {
  static const int mode = getenv("PTD_EIGH_STAGES") ? atoi(getenv("PTD_EIGH_STAGES")) : 2;
  return mode == 2 && n % TB == 0 && n >= 4 * TB && n <= 8192;
}

// A (working copy, n x n, both triangles, leading dimension ld; destroyed: afterwards it holds the reflector
// panels of stage 1) -> d, e.  V2 ([n][ldv2], zeroed here) receives the reflectors of stage 2.
int twostage_reduce(const TwoStagePlan& p, char* base, double* Aw, double* V2, int64_t ldv2, double* d, double* e,
                    hipEvent_t mid, hipStream_t st) {
  return twostage_reduce_stages(p, base, Aw, V2, ldv2, d, e, mid, 2, st);
}

int twostage_reduce_stages(const TwoStagePlan& p, char* base, double* Aw, double* V2, int64_t ldv2, double* d, double* e,
                           hipEvent_t mid, int stages, hipStream_t st) {
  const int n = p.n;
  const int64_t ld = p.ld;
  double* G1 = reinterpret_cast<double*>(base + p.off_G1);
  double* G2 = reinterpret_cast<double*>(base + p.off_G2);
  double* Z0 = reinterpret_cast<double*>(base + p.off_Z0);
  double* band = reinterpret_cast<double*>(base + p.off_band);
  int* prog = reinterpret_cast<int*>(base + p.off_prog);
  int* status = reinterpret_cast<int*>(base + p.off_status);
  double* L1 = reinterpret_cast<double*>(base + p.off_L1);
  double* L1inv = reinterpret_cast<double*>(base + p.off_L1inv);
  double* MT = reinterpret_cast<double*>(base + p.off_MT);
  double* Tall = reinterpret_cast<double*>(base + p.off_T);
  double* W0t = reinterpret_cast<double*>(base + p.off_W0t);
  double* Xt = reinterpret_cast<double*>(base + p.off_Xt);
  double* tau2 = reinterpret_cast<double*>(base + p.off_tau2);
  PTD_CHECK_HIP(hipMemsetAsync(base + p.off_zero, 0, p.zero_bytes, st));
  PTD_CHECK_HIP(hipMemsetAsync(tau2, 0, (size_t)n * p.npos * 8, st));  // (every entry of V2 that is read is written)
  const size_t lds_gram1 = (size_t)TB * (CH + 1) * 8, lds_gram2 = 2 * lds_gram1;
  const size_t lds_lmul = (size_t)TB * TB * 8, lds_lmul_g = lds_lmul + lds_gram1;
  for (int pn = 0; pn < p.npanels; ++pn) {
    const int j0 = pn * TB, r0 = j0 + TB, m = n - r0;
    double* Pt = Aw + (int64_t)j0 * ld + r0;          // wide panel [32][m]
    double* A22 = Aw + (int64_t)r0 * ld + r0;
    double* g1 = G1 + (size_t)pn * TB * TB;
    double* g2 = G2 + (size_t)pn * TB * TB;
    double* z0 = Z0 + (size_t)pn * TB * TB;
    double* Tp = Tall + (size_t)pn * TB * TB;
    const unsigned nch = (unsigned)ceil_div(m, CH);
    hipLaunchKernelGGL(ts_gram_kernel, dim3(nch), dim3(256), lds_gram1, st, Pt, Pt, ld, m, g1);
    hipLaunchKernelGGL(ts_chol_kernel, dim3(1), dim3(256), 0, st, g1, L1, L1inv, status);
    hipLaunchKernelGGL(ts_lmul_kernel, dim3(nch), dim3(256), lds_lmul_g, st, L1inv, Pt, ld, 0, m, g2, (double*)nullptr);
    hipLaunchKernelGGL(ts_hr_kernel, dim3(1), dim3(256), 0, st, g2, L1, Pt, ld, MT, Tp, band + (int64_t)r0 * LDB, status);
    hipLaunchKernelGGL(ts_lmul_kernel, dim3(nch), dim3(256), lds_lmul, st, MT, Pt, ld, TB, m, (double*)nullptr, W0t);
    // W0t (32 x m) = Vt A22  (split K, atomics into the zeroed W0t)
    const int tiles = (int)ceil_div(m, 64);
    int ks = (int)std::min<int64_t>(32, std::max<int64_t>(1, 768 / tiles));
    ks = (int)std::min<int64_t>(ks, std::max<int64_t>(1, m / 64));
    int rc = gemm_f64(Pt, ld, 1, A22, ld, 1, W0t, ld, TB, m, m, 1.0, true, ks, st);
    if (rc != PTD_OK) return rc;
    hipLaunchKernelGGL(ts_gram_kernel, dim3(nch), dim3(256), lds_gram2, st, Pt, W0t, ld, m, z0);
    hipLaunchKernelGGL(ts_x_kernel, dim3(nch), dim3(256), 0, st, Tp, z0, Pt, W0t, Xt, ld, m);
    // A22 -= V X^T + X V^T
    rc = gemm_f64_pair(Pt, Xt, Xt, Pt, 1, ld, ld, 1, A22, ld, m, m, TB, -1.0, nullptr, st);
    if (rc != PTD_OK) return rc;
  }
  hipLaunchKernelGGL(ts_band_diag_kernel, dim3((unsigned)(n / TB)), dim3(256), 0, st, Aw, ld, n, band);
  PTD_CHECK_LAUNCH("twostage stage 1");
  if (mid) PTD_CHECK_HIP(hipEventRecord(mid, st));
  if (stages < 2) return PTD_OK;
  // stage 2: one single-wave workgroup per sweep in flight (at most n / 64 sweeps overlap)
  static const int workers_env = getenv("PTD_CHASE_WORKERS") ? atoi(getenv("PTD_CHASE_WORKERS")) : 0;
  const int workers = workers_env > 0 ? workers_env : (int)std::min<int64_t>(192, std::max<int64_t>(8, n / (2 * TB) + 8));
  hipLaunchKernelGGL(ts_chase_kernel, dim3((unsigned)workers), dim3(64), 0, st, band, n, prog, status + 1, V2, ldv2, tau2,
                     p.npos);
  hipLaunchKernelGGL(ts_extract_tridiag_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, band, n, d, e);
  PTD_CHECK_LAUNCH("twostage stage 2");
  return PTD_OK;
}

int* twostage_status(const TwoStagePlan& p, char* base) { return reinterpret_cast<int*>(base + p.off_status); }

// Y (n x nvec, eigenvectors of T in columns) <- Q1 Q2 Y.  TV: scratch [n][ld] (T_p Vt_p of every panel).
int twostage_backtransform(const TwoStagePlan& p, char* base, const double* Aw, const double* V2, int64_t ldv2,
                           double* TV, double* Y, int64_t ldy, int nvec, hipEvent_t mid, hipStream_t st) {
  const int n = p.n;
  const int64_t ld = p.ld;
  const double* tau2 = reinterpret_cast<const double*>(base + p.off_tau2);
  const double* Tall = reinterpret_cast<const double*>(base + p.off_T);
  double* W2 = reinterpret_cast<double*>(base + p.off_W2);
  if (n <= 4096) {
    constexpr int CQ = 4;
    const size_t lds = (size_t)CQ * (n + 1) * 8;
    PTD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ts_apply_q2_kernel<CQ>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ts_apply_q2_kernel<CQ>, dim3((unsigned)ceil_div(nvec, CQ)), dim3(512), lds, st, V2, ldv2, tau2,
                       p.npos, n, Y, ldy, nvec);
  } else {
    constexpr int CQ = 2;
    const size_t lds = (size_t)CQ * (n + 1) * 8;
    PTD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ts_apply_q2_kernel<CQ>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ts_apply_q2_kernel<CQ>, dim3((unsigned)ceil_div(nvec, CQ)), dim3(512), lds, st, V2, ldv2, tau2,
                       p.npos, n, Y, ldy, nvec);
  }
  PTD_CHECK_LAUNCH("twostage Q2");
  if (mid) PTD_CHECK_HIP(hipEventRecord(mid, st));
  const int64_t ldt = n;
  // Q1 = H_0 H_1 ... : apply the last panel first,  Y[r0:, :] -= Vt_p^T (T_p Vt_p Y[r0:, :])
  hipLaunchKernelGGL(ts_tv_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)p.npanels), dim3(256), 0, st, Aw, ld, n,
                     Tall, TV, ldt);
  for (int pn = p.npanels - 1; pn >= 0; --pn) {
    const int j0 = pn * TB, r0 = j0 + TB, m = n - r0;
    const double* Vt = Aw + (int64_t)j0 * ld + r0;
    const double* TVt = TV + (int64_t)j0 * ldt + r0;
    PTD_CHECK_HIP(hipMemsetAsync(W2, 0, (size_t)TB * nvec * 8, st));
    int rc = gemm_f64(TVt, ldt, 1, Y + (int64_t)r0 * ldy, ldy, 1, W2, nvec, TB, nvec, m, 1.0, true, 16, st);
    if (rc != PTD_OK) return rc;
    rc = gemm_f64(Vt, 1, ld, W2, nvec, 1, Y + (int64_t)r0 * ldy, ldy, m, nvec, TB, -1.0, true, 1, st);
    if (rc != PTD_OK) return rc;
  }
  PTD_CHECK_LAUNCH("twostage Q1");
  return PTD_OK;
}

}  // namespace ptd
