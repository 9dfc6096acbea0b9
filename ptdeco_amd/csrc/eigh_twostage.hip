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
constexpr int SP = 33;            // LDS pitch of a 32 x 32 matrix

__device__ __forceinline__ double wave_sum64(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------------------ stage 1
// value of `v` in lane `src` (wave-uniform, here always a compile-time constant of an unrolled loop)
__device__ __forceinline__ double readlane_d(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Factorisations of a 32 x 32 matrix held in LDS as S[32][SP].  The sequential part runs on ONE wave with the
// matrix in registers (lane i = row i, right-looking, fully unrolled: the pivot row / column entries travel by
// v_readlane, there is no LDS traffic and no barrier inside); the other waves of the workgroup wait at the
// barrier that follows.  (The first version -- in LDS, three workgroup barriers per column -- took 51 us for a
// Cholesky factorisation + inverse and 137 us for the reconstruction kernel: 24 ms per n = 4096 matrix.)
// Cholesky (lower, in place in the lower triangle); returns false (to all threads) on a non-positive pivot.
__device__ bool chol32(double* S, int tid, int* flag_s) {
  if (tid < 64) {
    const int i = tid & 31;
    double w[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) w[j] = S[i * SP + j];
    int bad = 0;
#pragma unroll
    for (int k = 0; k < TB; ++k) {
      double piv = readlane_d(w[k], k);
      if (!(piv > 0.0)) { bad = 1; piv = 1.0; }
      const double d = sqrt(piv);
      w[k] = (i == k) ? d : w[k] / d;                 // column k of L (rows i > k; the rest is never read)
#pragma unroll
      for (int j = k + 1; j < TB; ++j) w[j] -= w[k] * readlane_d(w[k], j);   // S[i][j] -= L[i][k] L[j][k]
    }
    if (tid < TB) {
#pragma unroll
      for (int j = 0; j < TB; ++j) S[i * SP + j] = w[j];
    }
    if (tid == 0) *flag_s = bad;
  }
  __syncthreads();
  return *flag_s == 0;
}

// LU without pivoting of W - diag(D), D_kk = -sign of the pivot candidate (|pivot| >= 1: stable), in place:
// strict lower = multipliers L (unit diagonal implied), upper = U.  Same one-wave register scheme.
// With R2t (the lower Cholesky factor L2 = R2^T of the second pass, may be null = identity) the matrix factored is
// W - diag(D) R2: row k of D R2 is inserted when D_k becomes known (it has no entries left of the diagonal).  Its
// factors are L and U R2 of the unscaled problem W R2^-1 - diag(D), with the same signs D.
__device__ void lu_signed32(double* W, double* Dg, const double* R2t, int tid) {
  if (tid < 64) {
    const int i = tid & 31;
    double w[TB], r2[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      w[j] = W[i * SP + j];
      r2[j] = R2t ? ((j >= i) ? R2t[j * SP + i] : 0.0) : ((i == j) ? 1.0 : 0.0);   // row i of R2 (upper)
    }
#pragma unroll
    for (int k = 0; k < TB; ++k) {
      const double cand = readlane_d(w[k], k);
      const double dg = cand >= 0.0 ? -1.0 : 1.0;
      const double piv = cand - dg * readlane_d(r2[k], k);
      if (tid == 0) Dg[k] = dg;
      if (i == k) {
#pragma unroll
        for (int j = k; j < TB; ++j) w[j] -= dg * r2[j];   // row k of -D R2
      } else if (i > k) w[k] = w[k] / piv;             // multiplier
#pragma unroll
      for (int j = k + 1; j < TB; ++j) {
        const double ukj = readlane_d(w[j], k);       // U[k][j]
        if (i > k) w[j] -= w[k] * ukj;
      }
    }
    if (tid < TB) {
#pragma unroll
      for (int j = 0; j < TB; ++j) W[i * SP + j] = w[j];
    }
  }
  __syncthreads();
}

// X = L^-1 for a lower triangular L in LDS (unit_diag: the diagonal is taken as 1), zeros above the diagonal of X.
// One wave, lane i = row i of L and of X in registers: forward elimination of [L | I], row k of X is final at
// step k and travels to the rows below by v_readlane (no LDS traffic, no barrier inside; ends with a workgroup
// barrier).  (One thread per column reading L from LDS element by element: 40 us per call, latency bound.)
__device__ void tri_lower_inverse32_wave(const double* L, double* X, bool unit_diag, int lane);

__device__ void tri_lower_inverse32(const double* L, double* X, bool unit_diag, int tid) {
  if (tid < 64) tri_lower_inverse32_wave(L, X, unit_diag, tid);
  __syncthreads();
}

// (one wave; no barrier: the caller synchronises)
__device__ void tri_lower_inverse32_wave(const double* L, double* X, bool unit_diag, int tid) {
  {
    const int i = tid & 31;
    double l[TB], x[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) { l[j] = L[i * SP + j]; x[j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int k = 0; k < TB; ++k) {
      if (!unit_diag) {
        const double dinv = 1.0 / l[k];               // (only lane k's value is used)
#pragma unroll
        for (int j = 0; j <= k; ++j) x[j] = (i == k) ? x[j] * dinv : x[j];
      }
#pragma unroll
      for (int j = 0; j <= k; ++j) {
        const double xkj = readlane_d(x[j], k);       // X[k][j]
        if (i > k) x[j] -= l[k] * xkj;
      }
    }
    if (tid < TB) {
#pragma unroll
      for (int j = 0; j < TB; ++j) X[i * SP + j] = x[j];
    }
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

// K4: second Cholesky pass + Householder reconstruction of the panel (single workgroup).
//   in : G2 = Q1t Q1t^T, L1 (first pass), Q1t_top = At[0:32, 0:32] (At = the wide panel, leading dimension ld)
//   out: Vt_top written over Q1t_top, MT (Vt_rest = MT Q1t_rest), T (32 x 32 upper, row-major), the band entries of
//        R~ = D R2 R1 (the sub-diagonal block of the band), status on breakdown
// With Q = Q1 R2^-1:  LU(Q_top - D) = L U  <=>  LU(Q1_top - D R2) = L (U R2), so ONE factorisation of the scaled
// matrix gives L and U~ = U R2, and V_rest = Q_rest U^-1 = Q1_rest U~^-1: no R2^-1 on the way to V.  The three
// triangular inverses that remain (U~^T for MT, L and L2 for T = -U~ R2^-1 D L^-T) are independent and run on
// three waves at once.  Chain: Cholesky, LU, inverses, two 32^3 products.
__global__ __launch_bounds__(256) void ts_hr_kernel(const double* __restrict__ G2, const double* __restrict__ L1,
                                                    double* __restrict__ At, int64_t ld, double* __restrict__ MT,
                                                    double* __restrict__ T, double* __restrict__ band_row0,
                                                    int* __restrict__ status) {
  __shared__ double S[TB * SP], W[TB * SP], Ut[TB * SP], Lu[TB * SP], Xu[TB * SP], Xl[TB * SP], X2[TB * SP];
  __shared__ double Dg[TB];
  __shared__ int flag;
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int e = tid; e < TB * TB; e += 256) {
    S[(e >> 5) * SP + (e & 31)] = G2[e];
    W[(e & 31) * SP + (e >> 5)] = At[(int64_t)(e >> 5) * ld + (e & 31)];   // W[i][k] = Q1top[i][k] = Q1t[k][i]
  }
  __syncthreads();
  if (!chol32(S, tid, &flag) && tid == 0) atomicExch(status, 1);   // S lower = L2, R2 = L2^T
  lu_signed32(W, Dg, S, tid);                                       // W = strict lower L (unit) + upper U~
  for (int e = tid; e < TB * TB; e += 256) {
    const int i = e >> 5, j = e & 31;
    Ut[i * SP + j] = (j <= i) ? W[j * SP + i] : 0.0;                // U~^T (lower)
    Lu[i * SP + j] = (j < i) ? W[i * SP + j] : (i == j ? 1.0 : 0.0);
    // Vt_top[c][i] = L[i][c]
    At[(int64_t)i * ld + j] = (j > i) ? W[j * SP + i] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  if (wave == 0) tri_lower_inverse32_wave(Ut, Xu, false, tid & 63);       // Xu = (U~^T)^-1 = (U~^-1)^T = MT
  else if (wave == 1) tri_lower_inverse32_wave(Lu, Xl, true, tid & 63);   // Xl = L^-1
  else if (wave == 2) tri_lower_inverse32_wave(S, X2, false, tid & 63);   // X2 = L2^-1 ; R2^-1 = X2^T
  __syncthreads();
  for (int e = tid; e < TB * TB; e += 256) MT[e] = Xu[(e >> 5) * SP + (e & 31)];
  // Z[a][j] = sum_b R2inv[a][b] D_b Linv[j][b] = sum_b X2[b][a] D_b Xl[j][b]   (into Ut, dead by now)
  for (int e = tid; e < TB * TB; e += 256) {
    const int a = e >> 5, j = e & 31;
    double acc = 0.0;
#pragma unroll 8
    for (int b = 0; b < TB; ++b) acc += X2[b * SP + a] * Dg[b] * Xl[j * SP + b];
    Ut[a * SP + j] = acc;
  }
  __syncthreads();
  // T = -U~ Z (upper triangular);  R~[i][j] = D[i] sum_{k=i..j} L2[k][i] L1[j][k]  (j >= i): band entry
  // (r0 + i, j0 + j) at k = j - i + TB of band row r0 + i
  for (int e = tid; e < TB * TB; e += 256) {
    const int i = e >> 5, j = e & 31;
    double acc = 0.0, accr = 0.0;
#pragma unroll 8
    for (int k = 0; k < TB; ++k) {
      acc += (k >= i ? W[i * SP + k] : 0.0) * Ut[k * SP + j];
      accr += (k >= i && k <= j) ? S[k * SP + i] * L1[j * TB + k] : 0.0;
    }
    T[e] = (j >= i) ? -acc : 0.0;
    if (j >= i) band_row0[(int64_t)i * LDB + (j - i + TB)] = Dg[i] * accr;
  }
}

// ---- panel operations on the f64 matrix cores (v_mfma_f64_16x16x4_f64: lane l holds A[m = l & 15][k = l >> 4],
// B[k = l >> 4][n = l & 15], D[row = (l >> 4) + 4 reg][col = l & 15]).  The 32 x 32 operand sits in registers as MFMA
// fragments, a wave owns a 16-column tile (left multiplication) or a 32-column slice (Gram product) and reads the
// panel straight from memory.  (First version: VALU, one thread per column, the 32 x 32 operand broadcast from LDS
// element by element -- two LDS reads per fma, 32 workgroups: 13 - 35 us per call at m <= 4096 against 5 - 8 us.)

// G (32 x 32, zeroed by the caller) += X Y^T over the m columns (m % 32 == 0) of two wide panels; 4 waves x 32 columns
__global__ __launch_bounds__(256) void ts_gram_mfma_kernel(const double* __restrict__ X, const double* __restrict__ Y,
                                                           int64_t ld, int m, double* __restrict__ G) {
  __shared__ double red[4][TB * TB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int c0 = (blockIdx.x * 4 + wave) * 32;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  if (c0 < m) {
    double xa[2][8], ya[2][8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int c = c0 + 4 * ks + l4;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xa[t][ks] = X[(int64_t)(16 * t + l15) * ld + c];
        ya[t][ks] = (X == Y) ? xa[t][ks] : Y[(int64_t)(16 * t + l15) * ld + c];
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[a][ks], ya[b][ks], acc[a][b], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][(16 * a + l4 + 4 * r) * TB + 16 * b + l15] = acc[a][b][r];
  __syncthreads();
  for (int e = tid; e < TB * TB; e += 256) {
    const double v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    atomicAdd(&G[e], v);
  }
}

// X[:, c] <- Lm X[:, c] for the columns [c_lo, m) of a wide panel (c_lo, m multiples of 16); a wave owns 16 columns.
// Optionally accumulates the Gram matrix of the result into G and zeroes the same columns of a second panel.
__global__ __launch_bounds__(256) void ts_lmul_mfma_kernel(const double* __restrict__ Lm, double* __restrict__ X,
                                                           int64_t ld, int c_lo, int m, double* __restrict__ G,
                                                           double* __restrict__ zero_out) {
  __shared__ double tile[4][TB * 17];
  __shared__ double red[4][TB * TB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int c0 = (blockIdx.x * 4 + wave) * 16;
  const bool act = c0 >= c_lo && c0 < m;
  f64x4 acc[2] = {f64x4{0.0, 0.0, 0.0, 0.0}, f64x4{0.0, 0.0, 0.0, 0.0}};
  if (act) {
    double la[2][8], xb[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      la[0][ks] = Lm[l15 * TB + 4 * ks + l4];
      la[1][ks] = Lm[(16 + l15) * TB + 4 * ks + l4];
      xb[ks] = X[(int64_t)(4 * ks + l4) * ld + c0 + l15];
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[0][ks], xb[ks], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[1][ks], xb[ks], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) X[(int64_t)(16 * t + l4 + 4 * r) * ld + c0 + l15] = acc[t][r];
  }
  if (zero_out && c0 < m) {
#pragma unroll
    for (int r = 0; r < 8; ++r) zero_out[(int64_t)(l4 + 4 * r) * ld + c0 + l15] = 0.0;
  }
  if (G) {
    // Gram matrix of the wave's 32 x 16 result tile: through LDS into the operand layout (k = column of the tile)
    double* tl = tile[wave];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) tl[(16 * t + l4 + 4 * r) * 17 + l15] = act ? acc[t][r] : 0.0;
    __syncthreads();
    f64x4 g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) g[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double f0 = tl[l15 * 17 + 4 * ks + l4], f1 = tl[(16 + l15) * 17 + 4 * ks + l4];
      g[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, g[0][0], 0, 0, 0);
      g[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, g[0][1], 0, 0, 0);
      g[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f0, g[1][0], 0, 0, 0);
      g[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, g[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(16 * a + l4 + 4 * r) * TB + 16 * b + l15] = g[a][b][r];
    __syncthreads();
    for (int e = tid; e < TB * TB; e += 256) atomicAdd(&G[e], (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]));
  }
}

// Xt[:, c] = T^T W0t[:, c] - 1/2 C Vt[:, c],  C = T^T Z0 T (symmetric); a wave owns 16 columns
__global__ __launch_bounds__(256) void ts_x_mfma_kernel(const double* __restrict__ T, const double* __restrict__ Z0,
                                                        const double* __restrict__ Vt, const double* __restrict__ W0t,
                                                        double* __restrict__ Xt, int64_t ld, int m) {
  __shared__ double Ts[TB * SP], Zs[TB * SP], Es[TB * SP], Cs[TB * SP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int e = tid; e < TB * TB; e += 256) {
    Ts[(e >> 5) * SP + (e & 31)] = T[e];
    Zs[(e >> 5) * SP + (e & 31)] = Z0[e];
  }
  __syncthreads();
  mm32(Zs, SP, 1, Ts, SP, 1, Es, 1.0, tid);    // E = Z0 T
  __syncthreads();
  mm32(Ts, 1, SP, Es, SP, 1, Cs, -0.5, tid);   // -1/2 C = -1/2 T^T E
  __syncthreads();
  const int c0 = (blockIdx.x * 4 + wave) * 16;
  if (c0 >= m) return;
  f64x4 acc[2] = {f64x4{0.0, 0.0, 0.0, 0.0}, f64x4{0.0, 0.0, 0.0, 0.0}};
  double ta[2][8], ca[2][8], wb[8], vb[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int k = 4 * ks + l4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ta[t][ks] = Ts[k * SP + 16 * t + l15];       // (T^T)[i][k] = T[k][i]
      ca[t][ks] = Cs[(16 * t + l15) * SP + k];
    }
    wb[ks] = W0t[(int64_t)k * ld + c0 + l15];
    vb[ks] = Vt[(int64_t)k * ld + c0 + l15];
  }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[t][ks], wb[ks], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[t][ks], vb[ks], acc[t], 0, 0, 0);
    }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) Xt[(int64_t)(16 * t + l4 + 4 * r) * ld + c0 + l15] = acc[t][r];
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

// sum over the 16 lanes of a DPP row (quad_perm xor 1, xor 2, row_ror 4, row_ror 8): VALU only, no LDS crossbar
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_mov_d<0xB1>(v);
  v += dpp_mov_d<0x4E>(v);
  v += dpp_mov_d<0x124>(v);
  v += dpp_mov_d<0x128>(v);
  return v;
}

// ------------------------------------------------------------------------------------------------ stage 2
__device__ __forceinline__ double ld_sc1(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int PROG_DONE = 1 << 30;

// Bulge chasing (tools/twostage_proto.py: band_to_tridiag) as a systolic pipeline.
//
// Task (s, p) -- sweep s (column s), block position p -- touches band rows r_p .. r_p + 31, r_p = s + 1 + 32 p, and
// may run once (s, p - 1) and (s - 1, p + 1) are done: consecutive sweeps follow each other two positions apart,
// so the critical path is 2 n task times whatever the parallelism.  The first version ran one wave per sweep over
// a band in L2 (sc1 loads / stores, a progress counter per sweep): a task then costs two global round trips and a
// store drain, 7 us per sweep, 57 ms at n = 4096.  Here a workgroup owns CG consecutive sweeps (one wave each) and
// keeps the part of the band they are working on in LDS: a circular window of WROWS band rows that a fourth
// wave fills ahead of the first sweep and drains behind the last one.  Sweep-to-sweep hand-offs inside a group are
// LDS progress counters; only group-to-group hand-offs go through global memory (rows stored with sc1, a
// per-group counter published behind `s_waitcnt vmcnt(0)`, polled with sc1 loads, every spin bounded).
// Rows of different tasks that may run concurrently are disjoint (a band row belongs to exactly one block row).
constexpr int CG = 3;             // sweeps (waves) per group
constexpr int WROWS = 256;        // LDS window, band rows (slot = row % WROWS)
// window pitch in doubles = the pitch of the band in memory: a column of a block is (row, k = j - i + 64), i.e.
// stride WP - 1 = 65 doubles over the lanes -- odd in 8-byte units, conflict-free (pitch 65 gave stride 64: every
// lane of a column access on one bank)
constexpr int WP = LDB;
constexpr int CHASE_THREADS = 64 * (CG + 2);   // CG sweep waves, a loader wave, a storer wave

struct ChaseShared {
  double win[WROWS * WP];
  double vb[CG][TB], wb[CG][TB];
  volatile int prog[CG];       // completed tasks of each sweep of the current group
  volatile int loaded;         // band rows < loaded are in the window
  volatile int stored;         // band rows < stored have left the window
  volatile int abort_flag;
};

__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// x(i, h = 0) + x(i, h = 1) in both half waves: two v_permlane32_swap + one add (a ds_bpermute shuffle costs an LDS
// round trip on the dependent chain of every task)
__device__ __forceinline__ double half_sum(double x) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
}
// the value of lane (i, h = 0) in both half waves
__device__ __forceinline__ double from_first_half(double x) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(r1[0], r0[0]);
}
// sum over the 32 lanes of the first half wave (wave-uniform result): DPP row sums + two v_readlane
__device__ __forceinline__ double sum_first_half(double x) {
  const double r = row16_sum(x);
  return readlane_d(r, 0) + readlane_d(r, 16);
}

// Householder vector of x (one entry per lane of the first half wave, zero beyond the length): returns v_i for the
// lane's row (both half waves), tau and beta (uniform).  LAPACK dlarfg convention, v_0 = 1.
__device__ __forceinline__ void householder32w(double x, int i, bool first_half, double& v, double& tau, double& beta) {
  const double alpha = readlane_d(x, 0);
  const double sigma = sum_first_half((first_half && i >= 1) ? x * x : 0.0);
  if (sigma == 0.0) {
    tau = 0.0; beta = alpha; v = (i == 0) ? 1.0 : 0.0;
  } else {
    const double nrm = sqrt(alpha * alpha + sigma);
    beta = alpha >= 0.0 ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    v = (i == 0) ? 1.0 : x * sc;
  }
  v = from_first_half(v);
}

// one sweep of the chase by one wave, band rows in the LDS window
__device__ void chase_sweep_wave(ChaseShared& sh, const int w, const int s, const int n,
                                 double* __restrict__ V2, const int64_t ldv2, double* __restrict__ tau2,
                                 const int npos, const int dbg) {
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  double* vb = sh.vb[w];
  double* wb = sh.wb[w];
#define WIN(gi, gj) sh.win[((gi) % WROWS) * WP + ((gj) - (gi) + 2 * TB)]
  // wait until the rows of task p are there: wave 0 waits for the loader, the others for the sweep ahead
  auto wait_rows = [&](int p, int last_row) -> bool {
    for (long spin = 0; spin < 200000000L; ++spin) {
      const bool ok = (w == 0) ? (sh.loaded > last_row) : (sh.prog[w - 1] >= p + 2);
      if (ok) return true;
      if (sh.abort_flag) return false;
      __builtin_amdgcn_s_sleep(1);
    }
    sh.abort_flag = 1;
    return false;
  };
  int r = s + 1;
  int ln = min(TB, n - r);
  if (dbg & 1) {   // timing experiment: the hand-off protocol only, no arithmetic
    int p = 0;
    for (;;) {
      if (!wait_rows(p, r + ln - 1)) return;
      ++p;
      if (lane == 0) sh.prog[w] = p;
      r += ln;
      ln = min(TB, n - r);
      if (ln <= 0) break;
    }
    if (lane == 0) sh.prog[w] = PROG_DONE;
    return;
  }
  if (!wait_rows(0, r + ln - 1)) return;
  lds_order();
  double x = (h == 0 && i < ln) ? WIN(r + i, s) : 0.0;
  double v, tau, beta;
  householder32w(x, i, h == 0, v, tau, beta);
  if (i >= ln) v = 0.0;
  if (h == 0 && i < ln) WIN(r + i, s) = (i == 0) ? beta : 0.0;
  if (h == 0) vb[i] = v;
  int p = 0;
  for (;;) {
    // ---- two-sided update of the diagonal block D (rows / columns r .. r + ln) with (v, tau); vb holds v
    double Dv[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = 16 * h + jj;
      double val = 0.0;
      if (i < ln && j < ln) val = (j <= i) ? WIN(r + i, r + j) : WIN(r + j, r + i);
      Dv[jj] = val;
    }
    lds_order();
    double part = 0.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) part += Dv[jj] * vb[16 * h + jj];
    double wv_ = tau * half_sum(part);
    const double dotwv = sum_first_half(wv_ * v);
    wv_ -= 0.5 * tau * dotwv * v;
    if (h == 0) wb[i] = wv_;
    lds_order();
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = 16 * h + jj;
      Dv[jj] -= v * wb[j] + wv_ * vb[j];
      if (j <= i && i < ln) WIN(r + i, r + j) = Dv[jj];
    }
    if (h == 0 && i < ln) V2[(int64_t)s * ldv2 + r + i] = v;
    if (lane == 0) tau2[(int64_t)s * npos + p] = tau;
    lds_order();
    ++p;
    if (lane == 0) sh.prog[w] = p;
    // ---- next block position
    const int c0 = r;
    r = c0 + ln;
    ln = min(TB, n - r);
    if (ln <= 0) break;
    if (!wait_rows(p, r + ln - 1)) return;
    lds_order();
    // B = A[r : r + ln, c0 : c0 + 32] by rows: right-apply the previous reflector (vb over the columns)
    double Bv[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) Bv[jj] = (i < ln) ? WIN(r + i, c0 + 16 * h + jj) : 0.0;
    part = 0.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) part += Bv[jj] * vb[16 * h + jj];
    const double t = tau * half_sum(part);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) Bv[jj] -= t * vb[16 * h + jj];
    // new reflector from the first column of B;  H x = beta e1
    double vn, taun, betan;
    householder32w(h == 0 ? Bv[0] : 0.0, i, h == 0, vn, taun, betan);
    if (i >= ln) vn = 0.0;
    if (ln < 2) taun = 0.0;
    lds_order();   // (every read of the old vb is done)
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = 16 * h + jj;
      if (i < ln) WIN(r + i, c0 + j) = (j == 0) ? ((i == 0) ? betan : 0.0) : Bv[jj];
    }
    if (h == 0) vb[i] = vn;
    lds_order();
    // left-apply to the columns 1 .. 31 by columns, through the window (no private transposition buffer): lane
    // (j, g) holds rows 16 g .. 16 g + 15 of column j; a column access has stride 65 doubles: conflict-free
    {
      const int j = i, g = h;
      double cv[16];
#pragma unroll
      for (int ii = 0; ii < 16; ++ii) cv[ii] = (16 * g + ii < ln) ? WIN(r + 16 * g + ii, c0 + j) : 0.0;
      double up = 0.0;
#pragma unroll
      for (int ii = 0; ii < 16; ++ii) up += vb[16 * g + ii] * cv[ii];
      const double u = taun * half_sum(up);
      if (j >= 1) {
#pragma unroll
        for (int ii = 0; ii < 16; ++ii)
          if (16 * g + ii < ln) WIN(r + 16 * g + ii, c0 + j) = cv[ii] - u * vb[16 * g + ii];
      }
    }
    lds_order();
    v = vn; tau = taun;
  }
  lds_order();
  if (lane == 0) sh.prog[w] = PROG_DONE;
#undef WIN
}

// Loader wave of a group: brings 32-row chunks of the band into the window ahead of the first sweep, once the
// previous group has published them (band rows are contiguous in memory: a chunk is one flat, fully coalesced range).
__device__ void chase_loader_wave(ChaseShared& sh, const int grp, const int s_base, const int n,
                                  const double* __restrict__ band, const int* __restrict__ gprog, int* __restrict__ fail,
                                  const int dbg) {
  const int lane = threadIdx.x & 63;
  const int base_row = s_base + 1;
  int loaded = base_row, chunk = 0;
  long idle = 0;
  while (loaded < n) {
    const int hi = min(n, base_row + TB * (chunk + 1));
    bool ready = hi - sh.stored <= WROWS - 8;
    if (ready && grp > 0)
      ready = __hip_atomic_load(gprog + grp - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= chunk + 2;
    if (ready) {
      asm volatile("" ::: "memory");
      constexpr int UB = 34;   // 32 rows x 66 doubles = 33 wave loads: the whole chunk in flight
      if (!(dbg & 2)) {
        const int e0 = loaded * LDB, e1 = hi * LDB;
        double tmp[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int e = e0 + 64 * u + lane;
          tmp[u] = (e < e1) ? ld_sc1(&band[e]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int e = e0 + 64 * u + lane;
          const int row = e / LDB, k = e - row * LDB;
          if (e < e1) sh.win[(row % WROWS) * WP + k] = tmp[u];
        }
      }
      lds_order();
      loaded = hi;
      ++chunk;
      if (lane == 0) sh.loaded = loaded;
      idle = 0;
    } else {
      if (sh.abort_flag) return;
      if (++idle > 100000000L || ((idle & 1023) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        sh.abort_flag = 1;
        __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}

// Storer wave of a group: writes the rows the last sweep is done with back to memory and publishes the group's
// progress (the next group's loader polls it).
__device__ void chase_storer_wave(ChaseShared& sh, const int grp, const int ga, const int s_base, const int n,
                                  double* __restrict__ band, int* __restrict__ gprog, const int dbg) {
  const int lane = threadIdx.x & 63;
  const int s_last = s_base + ga - 1;
  int stored = s_base + 1, pub = 0;
  for (long idle = 0; idle < 400000000L; ++idle) {
    const int pl = sh.prog[ga - 1];
    if (pl > pub) {
      const int fin = (pl >= PROG_DONE) ? n : min(n, s_last + 1 + TB * pl);   // rows < fin are final
      lds_order();
      if (!(dbg & 2)) {
        for (int e = stored * LDB + lane; e < fin * LDB; e += 64) {
          const int row = e / LDB, k = e - row * LDB;
          st_sc1(&band[e], sh.win[(row % WROWS) * WP + k]);
        }
      }
      stored = max(stored, fin);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pub = pl;
      if (lane == 0) {
        sh.stored = stored;
        __hip_atomic_store(gprog + grp, pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (pl >= PROG_DONE) return;
      idle = 0;
    } else {
      if (sh.abort_flag) return;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  sh.abort_flag = 1;
}

__global__ __launch_bounds__(CHASE_THREADS) void ts_chase_kernel(double* __restrict__ band, int n,
                                                                 int* __restrict__ gprog, int* __restrict__ fail,
                                                                 double* __restrict__ V2, int64_t ldv2,
                                                                 double* __restrict__ tau2, int npos, int dbg) {
  extern __shared__ char chase_smem[];
  ChaseShared& sh = *reinterpret_cast<ChaseShared*>(chase_smem);
  const int wave = threadIdx.x >> 6;
  const int nsweeps = n - 2;
  const int ngroups = (nsweeps + CG - 1) / CG;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int s_base = grp * CG;
    const int ga = min(CG, nsweeps - s_base);     // active sweeps of this group
    if (threadIdx.x == 0) {
      for (int q = 0; q < CG; ++q) sh.prog[q] = 0;
      sh.loaded = 0;
      sh.stored = s_base + 1;
      sh.abort_flag = 0;
    }
    __syncthreads();
    if (wave < ga) chase_sweep_wave(sh, wave, s_base + wave, n, V2, ldv2, tau2, npos, dbg);
    else if (wave == CG) chase_loader_wave(sh, grp, s_base, n, band, gprog, fail, dbg);
    else if (wave == CG + 1) chase_storer_wave(sh, grp, ga, s_base, n, band, gprog, dbg);
    __syncthreads();
    if (sh.abort_flag) {
      if (threadIdx.x == 0) __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
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
// Y[:, c0 : c0 + 4] <- Q2 Y (the reflectors of stage 2 in reverse order of their generation), the column chunk
// resident in LDS as [4][n + 41].  Within a sweep the reflectors touch disjoint rows.  A quad of lanes owns one
// (reflector, column): lane sub of the quad holds rows sub, sub + 4, .., sub + 28, so a dot product is eight fmas
// and two quad_perm DPP adds; a wave (16 quads) handles 4 reflectors x 4 columns per step, wave w of 16 the positions
// 4 (w + 16 q) + ...  The four lanes that need the same reflector entries (one per column) each load a quarter
// of them and pass the rest around with row_ror DPP moves -- loaded four times over, the reflectors saturate the CU's
// vector-memory path (147 KB per sweep: 2.3 us per sweep whatever else changed) -- and the entries of the next THREE
// sweeps are in flight while one is applied (12 registers per sweep).
// History at n = 4096, k = 1024: one reflector per wave step, 32-lane ds_bpermute reductions, dependent loads 24 ms;
// DPP reductions + one-sweep prefetch 11.4 ms; 8 rows per lane 9.5 ms.
template <int MAXQ>
__global__ __launch_bounds__(1024) void ts_apply_q2_kernel(const double* __restrict__ V2, int64_t ldv2,
                                                           const double* __restrict__ tau2, int npos, int n,
                                                           double* __restrict__ Y, int64_t ldy, int nvec) {
  // rows n .. n + 39 are zero padding, the landing zone of reflector rows beyond the matrix (V2 is zero there too:
  // twostage_reduce clears it), so the inner loops carry no per-element predicate
  extern __shared__ double Ys[];
  constexpr int CQ = 4, PPI = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * CQ;
  const int pitch = n + 41;
  for (int e = tid; e < pitch * CQ; e += 1024) {
    const int r = e / CQ, c = e % CQ;
    Ys[c * pitch + r] = (r < n && c0 + c < nvec) ? Y[(int64_t)r * ldy + c0 + c] : 0.0;
  }
  const int sub = lane & 3, col = (lane >> 2) & 3, ps = lane >> 4;
  // logical element j = 2 k + e of this lane is reflector row sub + 4 ((2 (col - k) + e) & 7): k = 0 are the two
  // entries the lane loads itself, k = 1, 2, 3 the own entries of the lane k columns down (row_ror k * 4 brings the
  // value of lane l - 4 k of the 16-lane row)
  int yoff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) yoff[j] = col * pitch + sub + 4 * ((2 * (col - (j >> 1)) + (j & 1)) & 7);
  const double* vbase = V2 + sub + 4 * (2 * col);
  int poff[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) poff[q] = 1 + (PPI * (wave + 16 * q) + ps) * TB;
  struct Set { double v0[MAXQ], v1[MAXQ], t[MAXQ]; int row[MAXQ]; };
  auto fetch = [&](int s, Set& f) {
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int r0 = s + poff[q];
      const bool on = r0 < n;
      f.row[q] = on ? r0 : n;                    // absent reflector: zero rows of V2 / Ys, tau = 0
      f.t[q] = on ? tau2[(int64_t)s * npos + PPI * (wave + 16 * q) + ps] : 0.0;
      const double* vp = vbase + (int64_t)s * ldv2 + f.row[q];
      f.v0[q] = vp[0];
      f.v1[q] = vp[4];
    }
  };
  auto apply = [&](const Set& f) {
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      double v[8];
      v[0] = f.v0[q]; v[1] = f.v1[q];
      v[2] = dpp_mov_d<0x124>(v[0]); v[3] = dpp_mov_d<0x124>(v[1]);   // row_ror 4: from the lane one column down
      v[4] = dpp_mov_d<0x128>(v[0]); v[5] = dpp_mov_d<0x128>(v[1]);
      v[6] = dpp_mov_d<0x12C>(v[0]); v[7] = dpp_mov_d<0x12C>(v[1]);
      int yo[8];   // (indices into the shared array, one per element and opaque: single 8-byte DS accesses)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        yo[j] = yoff[j] + f.row[q];
        asm volatile("" : "+v"(yo[j]));
      }
      double y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = Ys[yo[j]];
      double d = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) d += v[j] * y[j];
      d += dpp_mov_d<0xB1>(d);
      d += dpp_mov_d<0x4E>(d);
      d *= f.t[q];
#pragma unroll
      for (int j = 0; j < 8; ++j) Ys[yo[j]] = y[j] - v[j] * d;
    }
  };
  // (__syncthreads() waits for vmcnt(0) too, i.e. for the prefetched reflectors: every earlier version of this
  // kernel paid a full global-load latency per sweep because of it, whatever else was changed)
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  Set f0, f1, f2, f3;
  fetch(n - 3, f0);
  if (n - 4 >= 0) fetch(n - 4, f1);
  if (n - 5 >= 0) fetch(n - 5, f2);
  __syncthreads();
  // four sweeps per iteration: the set fetched three sweeps ago is used in place (no register copies)
  for (int s = n - 3; s >= 0; s -= 4) {
    if (s - 3 >= 0) fetch(s - 3, f3);
    apply(f0);
    lds_barrier();
    if (s - 1 < 0) break;
    if (s - 4 >= 0) fetch(s - 4, f0);
    apply(f1);
    lds_barrier();
    if (s - 2 < 0) break;
    if (s - 5 >= 0) fetch(s - 5, f1);
    apply(f2);
    lds_barrier();
    if (s - 3 < 0) break;
    if (s - 6 >= 0) fetch(s - 6, f2);
    apply(f3);
    lds_barrier();
  }
  __syncthreads();
  for (int e = tid; e < n * CQ; e += 1024) {
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
#pragma unroll 1
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

// Opt-in (PTD_EIGH_STAGES=2, read on every call): as measured on MI355X the two-stage route is correct to
// 3e-15 but not yet faster than the one-stage reduction at n = 4096 (101 ms against 72 ms, DESIGN.md section 3:
// the bulge chase is bound by the instruction issue rate of the single wave that executes a task).
bool twostage_supported(int64_t n) {
  const char* e = getenv("PTD_EIGH_STAGES");
  return e && atoi(e) == 2 && n % TB == 0 && n >= 4 * TB && n <= 4096;   // (Q2 keeps 4 columns of Y in LDS)
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
  PTD_CHECK_HIP(hipMemsetAsync(tau2, 0, (size_t)n * p.npos * 8, st));
  PTD_CHECK_HIP(hipMemsetAsync(V2, 0, (size_t)n * ldv2 * 8, st));   // (the Q2 kernel reads rows past n as zeros)
  for (int pn = 0; pn < p.npanels; ++pn) {
    const int j0 = pn * TB, r0 = j0 + TB, m = n - r0;
    double* Pt = Aw + (int64_t)j0 * ld + r0;          // wide panel [32][m]
    double* A22 = Aw + (int64_t)r0 * ld + r0;
    double* g1 = G1 + (size_t)pn * TB * TB;
    double* g2 = G2 + (size_t)pn * TB * TB;
    double* z0 = Z0 + (size_t)pn * TB * TB;
    double* Tp = Tall + (size_t)pn * TB * TB;
    const unsigned ng = (unsigned)ceil_div(m, 128), nl = (unsigned)ceil_div(m, 64);
    hipLaunchKernelGGL(ts_gram_mfma_kernel, dim3(ng), dim3(256), 0, st, Pt, Pt, ld, m, g1);
    hipLaunchKernelGGL(ts_chol_kernel, dim3(1), dim3(256), 0, st, g1, L1, L1inv, status);
    hipLaunchKernelGGL(ts_lmul_mfma_kernel, dim3(nl), dim3(256), 0, st, L1inv, Pt, ld, 0, m, g2, (double*)nullptr);
    hipLaunchKernelGGL(ts_hr_kernel, dim3(1), dim3(256), 0, st, g2, L1, Pt, ld, MT, Tp, band + (int64_t)r0 * LDB, status);
    hipLaunchKernelGGL(ts_lmul_mfma_kernel, dim3(nl), dim3(256), 0, st, MT, Pt, ld, TB, m, (double*)nullptr, W0t);
    // W0t (32 x m) = Vt A22  (split K, atomics into the zeroed W0t)
    const int tiles = (int)ceil_div(m, 64);
    int ks = (int)std::min<int64_t>(32, std::max<int64_t>(1, 768 / tiles));
    ks = (int)std::min<int64_t>(ks, std::max<int64_t>(1, m / 64));
    int rc = gemm_f64(Pt, ld, 1, A22, ld, 1, W0t, ld, TB, m, m, 1.0, true, ks, st);
    if (rc != PTD_OK) return rc;
    hipLaunchKernelGGL(ts_gram_mfma_kernel, dim3(ng), dim3(256), 0, st, Pt, W0t, ld, m, z0);
    hipLaunchKernelGGL(ts_x_mfma_kernel, dim3(nl), dim3(256), 0, st, Tp, z0, Pt, W0t, Xt, ld, m);
    // A22 -= V X^T + X V^T
    rc = gemm_f64_pair(Pt, Xt, Xt, Pt, 1, ld, ld, 1, A22, ld, m, m, TB, -1.0, nullptr, st);
    if (rc != PTD_OK) return rc;
  }
  hipLaunchKernelGGL(ts_band_diag_kernel, dim3((unsigned)(n / TB)), dim3(256), 0, st, Aw, ld, n, band);
  PTD_CHECK_LAUNCH("twostage stage 1");
  if (mid) PTD_CHECK_HIP(hipEventRecord(mid, st));
  if (stages < 2) return PTD_OK;
  // stage 2: a group of CG sweeps per workgroup; a group trails its predecessor by ~2 CG + 2 block positions, so at
  // most ~n / (32 (2 CG + 2)) groups overlap: one workgroup per CU, all co-resident (the spins are bounded anyway)
  static const int workers_env = getenv("PTD_CHASE_WORKERS") ? atoi(getenv("PTD_CHASE_WORKERS")) : 0;
  const int workers = workers_env > 0 ? workers_env : (int)std::min<int64_t>(128, std::max<int64_t>(4, n / (TB * 2 * CG) + 4));
  PTD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ts_chase_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChaseShared)));
  hipLaunchKernelGGL(ts_chase_kernel, dim3((unsigned)workers), dim3(CHASE_THREADS), sizeof(ChaseShared), st, band, n,
                     prog, status + 1, V2, ldv2, tau2, p.npos, getenv("PTD_CHASE_DBG") ? atoi(getenv("PTD_CHASE_DBG")) : 0);
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
  {
    const size_t lds = (size_t)4 * (n + 41) * 8;
    PTD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ts_apply_q2_kernel<2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((ts_apply_q2_kernel<2>), dim3((unsigned)ceil_div(nvec, 4)), dim3(1024), lds, st, V2, ldv2, tau2,
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
