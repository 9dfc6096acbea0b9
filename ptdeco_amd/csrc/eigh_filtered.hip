// Top-k eigenpairs of a symmetric positive semi-definite f64 matrix by Chebyshev-filtered subspace iteration:
// the eigensolver of the rank search when it asks for a quarter of the spectrum (dwain on a square layer: the 1024
// largest of 4096, dwain.py:155-163 + 407-421), built from dense f64 products on the matrix cores instead of a
// Householder reduction of the whole matrix.
//
//   1. Lanczos (4 chains x 32 steps, one SYMV-with-4-vectors launch per step): spectral bounds [lo, hi] and the
//      density of states; from it the cut `a` with about m = k + k / 4 eigenvalues above it and an estimate of
//      lambda_k.  If lambda_k is too close to `a` for the filter to separate them in a sensible number of products
//      (a flat spectrum), the route declines and the caller reduces the matrix directly (eigh_tridiag).
//   2. X [n, m] random; rounds of X <- T_d((2 C - (a + lo)) / (a - lo)) X by the three-term recurrence (one product
//      C Y per degree, scaled so that the top eigenvalue stays O(1)), each followed by Cholesky-QR passes
//      (G = X^T X, shifted on the first pass of a round since cond(X) grows like 150^d; G = L L^T; X <- X L^-T).
//      Eigendirections above `a` grow like (x + sqrt(x^2 - 1))^d, x > 1 their position relative to the damped
//      interval; those inside stay bounded by 1.
//   3. Rayleigh-Ritz: Z = C X, H = X^T Z (m x m), H y = theta y by the dense solver at order m (eigh_tridiag: at
//      m = 1280 the whole reduction is resident on chip), V = X Y_k.
//   4. Residual check ||C v - theta v|| = ||Z y - V theta|| for every returned pair (no further product with C);
//      above the tolerance the route declines as well.
// Every step is a product on v_mfma_f64_16x16x4_f64 (gemm_f64.hip) except the small factorisations; n = 4096,
// k = 1024: 15 products with C (0.19 TFLOP each) instead of 4/3 n^3 flop of latency-bound Householder columns.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "chol_tile.h"
#include "common.h"
#include "kernels.h"

namespace ptd {

namespace {

constexpr int LZ_NV = 4;        // Lanczos chains (independent start vectors)
constexpr int LZ_STEPS = 32;
constexpr int FP = 66;          // LDS pitch of a 64 x 64 tile

// ------------------------------------------------------------------------------------------------ small kernels
__device__ __forceinline__ double hash_uniform(unsigned long long i, unsigned long long seed) {
  unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ULL + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;   // [-1, 1)
}

__global__ void fs_random_kernel(double* __restrict__ X, int64_t total, unsigned long long seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    X[i] = hash_uniform((unsigned long long)i, seed);
}

// Lanczos start: NV interleaved unit vectors [n][NV], previous vectors and W zero
__global__ __launch_bounds__(1024) void fs_lanczos_init_kernel(double* __restrict__ Qc, double* __restrict__ Qp,
                                                               double* __restrict__ W, int n) {
  __shared__ double red[1024];
  const int v = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int i = tid; i < n; i += 1024) {
    const double x = hash_uniform((unsigned long long)i * LZ_NV + v, 0x51ED27ULL);
    Qc[(int64_t)i * LZ_NV + v] = x;
    s += x * x;
  }
  red[tid] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double inv = 1.0 / sqrt(red[0]);
  for (int i = tid; i < n; i += 1024) {
    Qc[(int64_t)i * LZ_NV + v] *= inv;
    Qp[(int64_t)i * LZ_NV + v] = 0.0;
    W[(int64_t)i * LZ_NV + v] = 0.0;
  }
}

// W[r][0..3] = sum_k A[r][k] Q[k][0..3]: a wave owns 4 rows, lanes stride over k; Q (n x 4, 32 n bytes) sits in LDS
// when it fits (n <= 4096), else it is read through the caches.  HBM-bound: A is streamed once per launch.
template <bool QLDS>
__global__ __launch_bounds__(256) void fs_symv4_kernel(const double* __restrict__ A, int64_t lda, int n,
                                                       const double* __restrict__ Q, double* __restrict__ W) {
  extern __shared__ __attribute__((aligned(16))) char fs_smem[];
  double* qs = reinterpret_cast<double*>(fs_smem);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (QLDS) {
    for (int i = tid; i < n * (LZ_NV / 2); i += 256)
      reinterpret_cast<double2*>(qs)[i] = reinterpret_cast<const double2*>(Q)[i];
    __syncthreads();
  }
  const double* q = QLDS ? qs : Q;
  for (int r0 = (blockIdx.x * 4 + wid) * 4; r0 < n; r0 += gridDim.x * 16) {
    double acc[4][LZ_NV];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int v = 0; v < LZ_NV; ++v) acc[i][v] = 0.0;
    // a lane takes two consecutive k per trip (16-byte loads of A), four trips unrolled: 16 row loads in flight per lane
    constexpr int TRIPS = 4;
    const int n2 = n - n % (128 * TRIPS);
    // every wave would sweep the same 4-KiB column window of its rows at the same time (rows are 8 n bytes apart: the
    // same few memory channels for the whole chip): wave w starts w windows into the row and wraps around
    // (n = 4096: 38 -> 33 us per launch, 4 TB/s; a finer rotation and more loads in flight changed nothing)
    const int nwin = n2 / (128 * TRIPS);
    const int rot = nwin > 0 ? ((blockIdx.x * 4 + wid) % nwin) * (128 * TRIPS) : 0;
    for (int t = 0; t < n2; t += 128 * TRIPS) {
      const int kb = t + rot >= n2 ? t + rot - n2 : t + rot;
      const int k0 = kb + 2 * lane;
      double2 av[TRIPS][4];
#pragma unroll
      for (int u = 0; u < TRIPS; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          av[u][i] = (r0 + i < n) ? *reinterpret_cast<const double2*>(A + (int64_t)(r0 + i) * lda + k0 + 128 * u)
                                  : double2{0.0, 0.0};
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) {
        const int k = k0 + 128 * u;
        const double2 qa01 = *reinterpret_cast<const double2*>(q + (int64_t)k * LZ_NV);
        const double2 qa23 = *reinterpret_cast<const double2*>(q + (int64_t)k * LZ_NV + 2);
        const double2 qb01 = *reinterpret_cast<const double2*>(q + (int64_t)(k + 1) * LZ_NV);
        const double2 qb23 = *reinterpret_cast<const double2*>(q + (int64_t)(k + 1) * LZ_NV + 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i][0] += av[u][i].x * qa01.x + av[u][i].y * qb01.x;
          acc[i][1] += av[u][i].x * qa01.y + av[u][i].y * qb01.y;
          acc[i][2] += av[u][i].x * qa23.x + av[u][i].y * qb23.x;
          acc[i][3] += av[u][i].x * qa23.y + av[u][i].y * qb23.y;
        }
      }
    }
    for (int k = n2 + lane; k < n; k += 64) {
      const double2 q01 = *reinterpret_cast<const double2*>(q + (int64_t)k * LZ_NV);
      const double2 q23 = *reinterpret_cast<const double2*>(q + (int64_t)k * LZ_NV + 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double a1 = (r0 + i < n) ? A[(int64_t)(r0 + i) * lda + k] : 0.0;
        acc[i][0] += a1 * q01.x;
        acc[i][1] += a1 * q01.y;
        acc[i][2] += a1 * q23.x;
        acc[i][3] += a1 * q23.y;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int v = 0; v < LZ_NV; ++v) {
        double s = acc[i][v];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0 && r0 + i < n) W[(int64_t)(r0 + i) * LZ_NV + v] = s;
      }
  }
}

// One Lanczos step of chain v = blockIdx.x: alpha = q.w, w -= alpha q + beta_prev q_prev, beta = |w|, (q_prev, q) <-
// (q, w / beta).  ab: [NV][2][STEPS] (alphas, betas).
__global__ __launch_bounds__(1024) void fs_lanczos_step_kernel(double* __restrict__ Qc, double* __restrict__ Qp,
                                                               const double* __restrict__ W, int n,
                                                               double* __restrict__ ab, int step) {
  __shared__ double red[1024];
  __shared__ double bc;
  const int v = blockIdx.x, tid = threadIdx.x;
  double* alpha = ab + (size_t)v * 2 * LZ_STEPS;
  double* beta = alpha + LZ_STEPS;
  const double bprev = step > 0 ? beta[step - 1] : 0.0;
  double s = 0.0;
  for (int i = tid; i < n; i += 1024) s += Qc[(int64_t)i * LZ_NV + v] * W[(int64_t)i * LZ_NV + v];
  red[tid] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double al = red[0];
  __syncthreads();
  double s2 = 0.0;
  // (n <= 8 * 1024 elements per thread kept in registers would be nicer; the vectors are L2 resident)
  for (int i = tid; i < n; i += 1024) {
    const int64_t at = (int64_t)i * LZ_NV + v;
    const double w = W[at] - al * Qc[at] - bprev * Qp[at];
    s2 += w * w;
  }
  red[tid] = s2;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const double b = sqrt(red[0]);
    alpha[step] = al;
    beta[step] = b;
    bc = b;
  }
  __syncthreads();
  const double inv = bc > 0.0 ? 1.0 / bc : 0.0;
  for (int i = tid; i < n; i += 1024) {
    const int64_t at = (int64_t)i * LZ_NV + v;
    const double qc = Qc[at];
    const double w = W[at] - al * qc - bprev * Qp[at];
    Qp[at] = qc;
    Qc[at] = w * inv;
  }
}

// Out = b * Y + c * X (the part of a Chebyshev step that is not the product; the product is added by the GEMM)
__global__ void fs_axpby_kernel(double* __restrict__ Out, const double* __restrict__ Y, const double* __restrict__ X,
                                double b, double c, int64_t total2) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total2; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 y = reinterpret_cast<const double2*>(Y)[i];
    double2 o;
    if (X) {
      const double2 x = reinterpret_cast<const double2*>(X)[i];
      o.x = b * y.x + c * x.x;
      o.y = b * y.y + c * x.y;
    } else {
      o.x = b * y.x;
      o.y = b * y.y;
    }
    reinterpret_cast<double2*>(Out)[i] = o;
  }
}

// G = sum of `ns` slabs in index order (a deterministic K split)
__global__ void fs_sum_slabs_kernel(double* __restrict__ G, const double* __restrict__ S, int64_t slab, int ns,
                                    int64_t total2) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total2; i += (int64_t)gridDim.x * blockDim.x) {
    double2 acc = reinterpret_cast<const double2*>(S)[i];
    for (int q = 1; q < ns; ++q) {
      const double2 v = reinterpret_cast<const double2*>(S + q * slab)[i];
      acc.x += v.x;
      acc.y += v.y;
    }
    reinterpret_cast<double2*>(G)[i] = acc;
  }
}

// G[i][i] += shift_rel * trace(G)  (one workgroup)
__global__ __launch_bounds__(1024) void fs_shift_kernel(double* __restrict__ G, int m, double shift_rel) {
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int i = tid; i < m; i += 1024) s += G[(int64_t)i * m + i];
  red[tid] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double sh = shift_rel * red[0];
  for (int i = tid; i < m; i += 1024) G[(int64_t)i * m + i] += sh;
}

// H[j][i] = H[i][j] for j > i: the upper triangle is the mirror of the lower one (the product computed only the
// tiles that touch the lower triangle)
__global__ __launch_bounds__(256) void fs_symmetrize_kernel(double* __restrict__ H, int m) {
  __shared__ double t[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;      // lower tile (bi, bj), bj <= bi, is copied to (bj, bi)
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    const int i = bi * 32 + r, j = bj * 32 + tx;
    t[r][tx] = (i < m && j < m) ? H[(int64_t)i * m + j] : 0.0;
  }
  __syncthreads();
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    const int i = bj * 32 + r, j = bi * 32 + tx;   // element (i, j) of the mirror tile = lower element (j, i) = t[tx][r]
    if (i < m && j < m && j > i) H[(int64_t)i * m + j] = t[tx][r];
  }
}

// Column statistics of the Ritz vectors in ONE row-contiguous pass over T = C V and V (a workgroup owns 64 columns of a
// slab of rows: 512-byte row segments instead of one 8-byte element per 8-KiB stride): per (slab, column) the sum of
// squares of T[:, c] - V[:, c] lam[c] and the entry of largest modulus with its row (lowest row wins a tie).
constexpr int FS_SLABS = 16;
__global__ __launch_bounds__(256) void fs_colstat_kernel(const double* __restrict__ T, int64_t ldt,
                                                         const double* __restrict__ V, int64_t ldv,
                                                         const double* __restrict__ lam, int n, int k,
                                                         double* __restrict__ stat) {
  __shared__ double ss[4][64], sb[4][64];
  __shared__ int si[4][64];
  const int tid = threadIdx.x, cl = tid & 63, rq = tid >> 6;
  const int c = blockIdx.x * 64 + cl, slab = blockIdx.y;
  const int rows = (n + FS_SLABS - 1) / FS_SLABS;
  const int r0 = slab * rows, r1 = min(n, r0 + rows);
  double s = 0.0, best = -1.0;
  int at = r0;
  if (c < k) {
    const double l = lam[c];
    for (int i = r0 + rq; i < r1; i += 4) {
      const double v = V[(int64_t)i * ldv + c];
      const double r = T[(int64_t)i * ldt + c] - v * l;
      s += r * r;
      const double a = fabs(v);
      if (a > best) { best = a; at = i; }
    }
  }
  ss[rq][cl] = s; sb[rq][cl] = best; si[rq][cl] = at;
  __syncthreads();
  if (rq == 0 && c < k) {
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      s += ss[q][cl];
      if (sb[q][cl] > best || (sb[q][cl] == best && si[q][cl] < at)) { best = sb[q][cl]; at = si[q][cl]; }
    }
    double* o = stat + ((int64_t)slab * k + c) * 3;
    o[0] = s; o[1] = best; o[2] = (double)at;
  }
}

// per column, slabs in index order (a fixed summation order): the residual norm and |lam| into out[0], out[1] (bit
// patterns of non-negative doubles order like integers), and the sign that makes the column's entry of largest
// modulus positive -- so the sign does not hang on the last bits of the Ritz problem
__global__ __launch_bounds__(256) void fs_colfinal_kernel(const double* __restrict__ stat, const double* __restrict__ V,
                                                          int64_t ldv, const double* __restrict__ lam, int k,
                                                          unsigned long long* __restrict__ out, double* __restrict__ sgn) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= k) return;
  double s = 0.0, best = -1.0;
  int at = 0;
  for (int slab = 0; slab < FS_SLABS; ++slab) {
    const double* o = stat + ((int64_t)slab * k + c) * 3;
    s += o[0];
    if (o[1] > best) { best = o[1]; at = (int)o[2]; }
  }
  atomicMax(out, (unsigned long long)__double_as_longlong(sqrt(s)));
  atomicMax(out + 1, (unsigned long long)__double_as_longlong(fabs(lam[c])));
  sgn[c] = V[(int64_t)at * ldv + c] < 0.0 ? -1.0 : 1.0;
}

__global__ __launch_bounds__(256) void fs_flip_kernel(double* __restrict__ V, int64_t ldv, int n, int k,
                                                      const double* __restrict__ sgn) {
  const int64_t total = (int64_t)n * k;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % k);
    const int64_t r = i / k;
    if (sgn[c] < 0.0) V[r * ldv + c] = -V[r * ldv + c];
  }
}

// ------------------------------------------------------------------------------------ triangular inverse
// acc (+)= P Q^T for 64 x 64 row-major LDS tiles of pitch FP; the four waves own 32 x 32 quadrants (2 x 2 MFMA tiles)
__device__ __forceinline__ void fs_tile_abt(const double* __restrict__ Ps, const double* __restrict__ Qs, int wr, int wc,
                                            int lane, f64x4 (&acc)[2][2]) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < FB; kk += 4) {
    const double a0 = Ps[(wr * 32 + l15) * FP + kk + l4];
    const double a1 = Ps[(wr * 32 + 16 + l15) * FP + kk + l4];
    const double b0 = Qs[(wc * 32 + l15) * FP + kk + l4];
    const double b1 = Qs[(wc * 32 + 16 + l15) * FP + kk + l4];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// ---- Cholesky factor and its inverse of a 64 x 64 SPD tile, one workgroup of 256 threads
// Thread (r = tid & 63, g = tid >> 6) keeps row r in REGISTERS, columns c = g mod 4: ONE array y[16] that holds the FULL
// symmetric row A[r][c] while column c is still to be eliminated and X[r][c] (X = Lhat^-1, Lhat = L D^-1 unit lower) from
// then on.  Right-looking, ONE barrier per column j.  Because both triangles are kept current, row j of the registers IS
// everything step j needs from elsewhere: its entries c > j are the column A[c][j] = A[j][c], its entries c < j the row
// X[j][c] of the inverse (final by then), entry j the pivot.  So the four threads that hold row j publish one 64-entry LDS
// line P (double buffered) and every thread does, with m = P[r] / P[j] for r > j (zero for r <= j),
//   y[c] -= m P[c]            every column c != j   (c > j: A[r][c] -= L[r][j] L[c][j];   c < j: X[r][c] -= Lhat[r][j] X[j][c])
//   y[j]  = r == j ? 1 : -m   the slot changes hands: X[r][j] = -Lhat[r][j]
// 16 broadcast reads at fixed offsets, 16 FMAs, one reciprocal; the slot indices are static although the loop over j is
// rolled.  At the end L^-1 = D^-1 X.  (History of this step, per 64 x 64 tile: the tile in LDS with read-modify-write
// updates 60 us; two register arrays with masked multipliers 65 us; one array, column published by its owners and the
// row of X by row j, addresses selected per slot 50 us -- 270 instructions a column, half of them scalar address
// arithmetic --; one wave with `v_readlane` broadcasts, fully unrolled: the compiler spilled 35 k instructions.)
// A: LDS tile (pitch FQ, lower triangle read); buf: 512 doubles of LDS.  A non-positive pivot raises *fail.
__device__ __forceinline__ void fs_chol_inv_tile(const double* __restrict__ A, double* __restrict__ buf,
                                                 double* __restrict__ out, int* __restrict__ fail, int tid) {
  const int r = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  double y[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int c = g + 4 * s;
    y[s] = (c <= r) ? A[r * FQ + c] : A[c * FQ + r];      // the full row, mirrored from the lower triangle
  }
  __syncthreads();   // (buf may alias the tile's neighbours; everyone has its row)
  bool bad = false;
  double myd = 1.0;
  // Two columns per barrier: rows j and j + 1 publish their lines P and R as they stand BEFORE column j is eliminated;
  // every thread forms the multipliers of column j, row j + 1 as column j leaves it (Q = R - l P, slot j = -l: the
  // very operations the row's own lane performs), the pivot and multipliers of column j + 1 from Q, and applies both
  // updates -- the same fma sequence, bit for bit, as two one-column steps, for one barrier and one LDS round trip.
#pragma unroll 1
  for (int j = 0; j < FB; j += 2) {
    const int i = j >> 2, gj = j & 3;
    double* P = buf + ((j >> 1) & 1) * 128;
    double* R = P + 64;
    if (r == j || r == j + 1) {
      double* dst = (r == j) ? P : R;
#pragma unroll
      for (int s = 0; s < 16; ++s) dst[g + 4 * s] = y[s];
    }
    __syncthreads();
    const double d0 = P[j];
    const bool ok0 = d0 > 0.0 && d0 < INFINITY;
    const double p0 = ok0 ? fs_rcp(d0) : 1.0;
    const double pj1 = P[j + 1];
    const double l = pj1 * p0;                               // multiplier of row j + 1 in column j
    const double m0 = (r > j) ? P[r] * p0 : 0.0;
    const double d1 = fma(-l, pj1, R[j + 1]);                // pivot of column j + 1
    const bool ok1 = d1 > 0.0 && d1 < INFINITY;
    const double p1 = ok1 ? fs_rcp(d1) : 1.0;
    const double m1 = (r > j + 1) ? fma(-l, P[r], R[r]) * p1 : 0.0;
    bad = bad || !ok0 || !ok1;
    myd = (r == j && ok0) ? d0 : myd;
    myd = (r == j + 1 && ok1) ? d1 : myd;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double pc = P[g + 4 * s];
      const double t = fma(-m0, pc, y[s]);
      y[s] = fma(-m1, fma(-l, pc, R[g + 4 * s]), t);
    }
    if (g == gj) {        // (wave-uniform) slot j: X[r][j] = -Lhat[r][j] after column j, then column j + 1 acts on it (Q[j] = -l)
      const double xj = fma(-m1, -l, (r == j) ? 1.0 : -m0);
#pragma unroll
      for (int s = 0; s < 16; ++s) y[s] = (s == i) ? xj : y[s];
    }
    if (g == gj + 1) {    // slot j + 1
      const double xj = (r == j + 1) ? 1.0 : -m1;
#pragma unroll
      for (int s = 0; s < 16; ++s) y[s] = (s == i) ? xj : y[s];
    }
  }
  const double rs = 1.0 / sqrt(myd);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int c = g + 4 * s;
    out[r * FB + c] = (c <= r) ? y[s] * rs : 0.0;
  }
  if (bad && tid == 0) atomicExch(fail, 1);
}

__device__ __forceinline__ void fs_load_tile(double* __restrict__ dst, const double* __restrict__ src, int64_t ld, int tid) {
#pragma unroll
  for (int p2 = 0; p2 < 8; ++p2) {
    const int idx = tid + 256 * p2;
    const int r = idx >> 5, c2 = (idx & 31) * 2;
    *reinterpret_cast<double2*>(&dst[r * FP + c2]) = *reinterpret_cast<const double2*>(src + (int64_t)r * ld + c2);
  }
}
// accumulators -> LDS tile (row-major)
__device__ __forceinline__ void fs_store_acc(double* __restrict__ dst, const f64x4 (&acc)[2][2], int wr, int wc, int lane,
                                             double sign) {
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dst[(wr * 32 + a * 16 + l4 + 4 * r) * FP + wc * 32 + b * 16 + l15] = sign * acc[a][b][r];
}
__device__ __forceinline__ void fs_zero_acc(f64x4 (&acc)[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
}

// Linv_0 of the first diagonal tile of G
__global__ __launch_bounds__(256) void fs_diag0_kernel(const double* __restrict__ G, int m, double* __restrict__ linv,
                                                       int* __restrict__ fail, int leaf) {
  __shared__ __attribute__((aligned(16))) double T[FB * FQ];
  __shared__ __attribute__((aligned(16))) double Xs[FB * FQ];
  __shared__ __attribute__((aligned(16))) double Dv[8 * 16 * LP];
  for (int e = threadIdx.x; e < FB * FB; e += 256) T[(e >> 6) * FQ + (e & 63)] = G[(int64_t)(e >> 6) * m + (e & 63)];
  __syncthreads();
  if (leaf) fs_chol_inv_tile2(T, Dv, Dv + 4 * 16 * LP, Xs, linv, fail, threadIdx.x);
  else fs_chol_inv_tile(T, Xs, linv, fail, threadIdx.x);
}

// W work = identity
__global__ void fs_identity_kernel(double* __restrict__ W, int m) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)m * m; i += (int64_t)gridDim.x * blockDim.x)
    W[i] = (i / m == i % m) ? 1.0 : 0.0;
}

// Step k of the Cholesky sweep G = L L^T that also forms W = L^-T (upper triangular) without ever storing L:
// with P = Linv_kk (inverse of the factor of the current diagonal tile, from the previous step),
//   G tile (i, j), k < j <= i:   G_ij -= (G_ik P^T) (G_jk P^T)^T;   the workgroup of (k+1, k+1) then factors its tile:
//                                Linv_{k+1}
//   W tile (r, j), r <= k < j:   Ww_rj -= (Ww_rk P^T) (G_jk P^T)^T        (Ww starts as the identity: the same right-
//   W tile (r, k), r <= k:       Wout_rk = Ww_rk P^T                        looking substitution applied to I gives L^-T)
// Every tile a step writes is read by no other workgroup of that step (column k of G and of Ww is read-only, the final
// column goes to Wout), so the steps need no synchronisation beyond their launch order.
__global__ __launch_bounds__(256) void fs_sweep_kernel(double* __restrict__ G, int m, double* __restrict__ Ww,
                                                       double* __restrict__ Wout, double* __restrict__ linv,
                                                       int* __restrict__ fail, int k, int leaf) {
  __shared__ __attribute__((aligned(16))) double Ps[FB * FP];
  __shared__ __attribute__((aligned(16))) double Qs[FB * FP];
  __shared__ __attribute__((aligned(16))) double Ss[FB * FP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nb = m / FB, nt = nb - k - 1;
  const double* P = linv + (size_t)k * FB * FB;
  int idx = blockIdx.x;
  f64x4 acc[2][2];
  fs_load_tile(Qs, P, FB, tid);                              // Qs = Linv_kk (row-major): X Qs^T = X Linv^T
  if (idx < k + 1) {
    // ---- final column block k of W, row tile r
    const int r = idx;
    fs_load_tile(Ps, Ww + (int64_t)r * FB * m + (int64_t)k * FB, m, tid);
    __syncthreads();
    fs_zero_acc(acc);
    fs_tile_abt(Ps, Qs, wr, wc, lane, acc);
    double* dst = Wout + (int64_t)r * FB * m + (int64_t)k * FB;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          dst[(int64_t)(wr * 32 + a * 16 + l4 + 4 * q) * m + wc * 32 + b * 16 + l15] = acc[a][b][q];
    return;
  }
  idx -= k + 1;
  const bool wrole = idx < (k + 1) * nt;
  int ti, tj;   // G role: tile (ti, tj); W role: row tile ti of Ww, column tj
  if (wrole) {
    ti = idx / nt;
    tj = k + 1 + idx % nt;
  } else {
    idx -= (k + 1) * nt;
    int t = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);
    while (t * (t + 1) / 2 > idx) --t;
    while ((t + 1) * (t + 2) / 2 <= idx) ++t;
    ti = k + 1 + t;
    tj = k + 1 + (idx - t * (t + 1) / 2);
  }
  // Ss = G_jk P^T  (= L_jk)
  fs_load_tile(Ps, G + (int64_t)tj * FB * m + (int64_t)k * FB, m, tid);
  __syncthreads();
  fs_zero_acc(acc);
  fs_tile_abt(Ps, Qs, wr, wc, lane, acc);
  __syncthreads();
  fs_store_acc(Ss, acc, wr, wc, lane, 1.0);
  // Ps = (row operand) P^T: G_ik P^T or Ww_rk P^T
  const bool same = !wrole && ti == tj;
  if (!same) {
    const double* src = wrole ? Ww + (int64_t)ti * FB * m + (int64_t)k * FB : G + (int64_t)ti * FB * m + (int64_t)k * FB;
    fs_load_tile(Ps, src, m, tid);
    __syncthreads();
    fs_zero_acc(acc);
    fs_tile_abt(Ps, Qs, wr, wc, lane, acc);
    __syncthreads();
    fs_store_acc(Ps, acc, wr, wc, lane, 1.0);
  }
  __syncthreads();
  fs_zero_acc(acc);
  fs_tile_abt(same ? Ss : Ps, Ss, wr, wc, lane, acc);        // acc[r][c] = sum_q Row[r][q] L_jk[c][q]
  double* dst = (wrole ? Ww : G) + (int64_t)ti * FB * m + (int64_t)tj * FB;
  const bool diag = same && ti == k + 1;
  double upd[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        upd[a][b][q] = dst[(int64_t)(wr * 32 + a * 16 + l4 + 4 * q) * m + wc * 32 + b * 16 + l15];
  if (!diag) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          dst[(int64_t)(wr * 32 + a * 16 + l4 + 4 * q) * m + wc * 32 + b * 16 + l15] = upd[a][b][q] - acc[a][b][q];
    return;
  }
  // ---- the next diagonal tile: updated value into LDS (pitch FQ), factor + inverse by the whole workgroup
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        Ps[(wr * 32 + a * 16 + l4 + 4 * q) * FQ + wc * 32 + b * 16 + l15] = upd[a][b][q] - acc[a][b][q];
  __syncthreads();
  if (leaf) fs_chol_inv_tile2(Ps, Qs, Qs + 4 * 16 * LP, Ss, linv + (size_t)(k + 1) * FB * FB, fail, tid);
  else fs_chol_inv_tile(Ps, Ss, linv + (size_t)(k + 1) * FB * FB, fail, tid);
}

// ------------------------------------------------------------------------------------------------ host side
// eigenvalues of the symmetric tridiagonal (d, e) by implicit QL, with the first and the last components of the
// normalised eigenvectors in z and y (on entry z = e_1, y = e_n).  d, e, z, y of length n; e[i] couples i and i + 1.
// false: no convergence.
bool tridiag_ql_first_row(int n, double* d, double* e, double* z, double* y) {
  if (n <= 0) return true;
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 80) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + copysign(r, g));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          e[i + 1] = r = hypot(f, g);
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
          f = z[i + 1];
          z[i + 1] = s * z[i] + c * f;
          z[i] = c * z[i] - s * f;
          f = y[i + 1];
          y[i + 1] = s * y[i] + c * f;
          y[i] = c * y[i] - s * f;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  return true;
}

// G = L L^T (lower tiles of the m x m matrix G, destroyed) and Wt = L^-T (upper triangular) together, one launch per
// 64-column panel; Hw: m x m work matrix, linv: (m / 64) tiles of 64 x 64, *fail raised on a non-positive pivot
int chol_sweep(double* G, int m, double* Hw, double* Wt, double* linv, int* fail, int leaf, hipStream_t st) {
  PTD_CHECK_HIP(hipMemsetAsync(Wt, 0, (size_t)m * m * 8, st));
  hipLaunchKernelGGL(fs_identity_kernel, dim3(1024), dim3(256), 0, st, Hw, m);
  hipLaunchKernelGGL(fs_diag0_kernel, dim3(1), dim3(256), 0, st, G, m, linv, fail, leaf);
  const int nb = m / FB;
  for (int kk = 0; kk < nb; ++kk) {
    const int nt = nb - kk - 1;
    const int wgs = (kk + 1) + (kk + 1) * nt + nt * (nt + 1) / 2;
    hipLaunchKernelGGL(fs_sweep_kernel, dim3(wgs), dim3(256), 0, st, G, m, Hw, Wt, linv, fail, kk, leaf);
  }
  return PTD_OK;
}

struct FilterPlan {
  int m;
  size_t off_X, off_Y, off_Z, off_H, off_G, off_W, off_linv, off_Yk, off_lam, off_lz, off_ab, off_flags, off_eigh;
  size_t eigh_bytes, total;
};

int filter_block(int64_t n, int64_t k) {
  // k + a fraction of k more (at least 128), a multiple of 64 (the panels of the Cholesky sweep; the products pick
  // their column tile from 64 / 80 / 96 / 128).  The fraction trades products for the Rayleigh-Ritz problem: 1/4
  // (m = 1280 at k = 1024) takes 16 products and an order-1280 eigenproblem, 1/2 (m = 1536) 9 wider products, four
  // Cholesky-QR passes instead of six and an order-1536 eigenproblem.  PTD_EIGH_FILTER_OVERSAMPLE overrides.
  const char* e = getenv("PTD_EIGH_FILTER_OVERSAMPLE");
  const double f = e ? std::min(1.0, std::max(0.05, atof(e))) : 0.25;
  const int64_t want = k + std::max<int64_t>((int64_t)ceil((double)k * f), 128);
  return (int)std::min<int64_t>(align_up((size_t)want, 64), n / 128 * 128);
}

FilterPlan filter_plan(int64_t n, int64_t k) {
  FilterPlan p{};
  p.m = filter_block(n, k);
  const size_t m = (size_t)p.m;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
  p.off_X = take((size_t)n * m * 8);
  p.off_Y = take((size_t)n * m * 8);
  p.off_Z = take((size_t)n * m * 8);
  p.off_H = take(m * m * 8);
  p.off_G = take(m * m * 8);
  p.off_W = take(m * m * 8);
  p.off_linv = take((m / FB) * FB * FB * 8);
  p.off_Yk = take(m * (size_t)k * 8);
  p.off_lam = take(m * 8);
  p.off_lz = take((size_t)3 * n * LZ_NV * 8);
  p.off_ab = take((size_t)LZ_NV * 2 * LZ_STEPS * 8);
  p.off_flags = take(256);
  p.eigh_bytes = tridiag_workspace_bytes((int64_t)m);
  p.off_eigh = take(p.eigh_bytes);
  p.total = o;
  return p;
}

double env_double(const char* name, double dflt) {
  const char* e = getenv(name);
  return e ? atof(e) : dflt;
}

}  // namespace

// Diagnostic (ptd_chol_inverse): the Cholesky sweep of the orthonormalisation passes on its own.
size_t chol_inverse_workspace_bytes(int64_t m) {
  return align_up((size_t)m * m * 8, 256) + align_up((size_t)(m / FB) * FB * FB * 8, 256) + 256;
}
int chol_inverse(double* G, int64_t m, double* Wt, void* ws, size_t ws_bytes, hipStream_t st) {
  PTD_REQUIRE(G && Wt && ws && m >= FB && m % FB == 0 && m <= 8192, "ptd_chol_inverse: m must be a multiple of 64 in [64, 8192]");
  if (ws_bytes < chol_inverse_workspace_bytes(m)) {
    set_error("ptd_chol_inverse: workspace too small");
    return PTD_ERR_WORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  double* Hw = reinterpret_cast<double*>(base);
  double* linv = reinterpret_cast<double*>(base + align_up((size_t)m * m * 8, 256));
  int* fail = reinterpret_cast<int*>(base + align_up((size_t)m * m * 8, 256) + align_up((size_t)(m / FB) * FB * FB * 8, 256));
  PTD_CHECK_HIP(hipMemsetAsync(fail, 0, 256, st));
  const int leaf = env_double("PTD_EIGH_FILTER_LEAF", 1.0) != 0.0 ? 1 : 0;
  int rc = chol_sweep(G, (int)m, Hw, Wt, linv, fail, leaf, st);
  if (rc != PTD_OK) return rc;
  PTD_CHECK_LAUNCH("ptd_chol_inverse");
  int h = 0;
  PTD_CHECK_HIP(hipMemcpyAsync(&h, fail, sizeof(int), hipMemcpyDeviceToHost, st));
  PTD_CHECK_HIP(hipStreamSynchronize(st));
  if (h) {
    set_error("ptd_chol_inverse: the matrix is not numerically positive definite");
    return PTD_ERR_UNSUPPORTED;
  }
  return PTD_OK;
}

// ---- late declines are remembered (per calling thread, device and shape)
// A decline after the density estimate costs 1.5 ms of a 55 ms direct reduction; a LATE one -- a Cholesky breakdown, a
// residual that does not come down, clustered Ritz values -- has spent some fifteen products and a Rayleigh-Ritz solve
// first.  The layers of one model share shapes and tend to share the character of their spectra, so after a late
// decline THAT FOLLOWS ANOTHER (no success in between) the next 1, 2, 4, ... 64 requests of the same (device, n, k) go
// straight to the direct route; a success resets the count.
// The memory belongs to the CALLING THREAD and the caller can clear it (ptd_eigh_forget_declines: the drivers do at
// the start of every decompose_in_place): which route a layer takes is then a function of the sequence of requests
// that one caller made since -- not of what other threads of the process were solving, nor of an earlier run (round 5
// kept one table per process, written by whichever concurrent chain finished first).
namespace {
struct DeclineNote { int device; int64_t n, k; int fails; int skip; };
thread_local std::vector<DeclineNote> g_declines;
DeclineNote* decline_note(int device, int64_t n, int64_t k, bool create) {
  for (auto& d : g_declines)
    if (d.device == device && d.n == n && d.k == k) return &d;
  if (!create) return nullptr;
  if (g_declines.size() >= 256) g_declines.clear();
  g_declines.push_back(DeclineNote{device, n, k, 0, 0});
  return &g_declines.back();
}
int current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}
bool backoff_enabled() {     // PTD_EIGH_FILTER_BACKOFF=0: every request is tried afresh (tests)
  const char* e = getenv("PTD_EIGH_FILTER_BACKOFF");
  return !(e && atoi(e) == 0);
}
}  // namespace

// true: the filtered route should not be tried for this request (a recent late decline of the same shape)
bool eigh_filtered_backed_off(int64_t n, int64_t k) {
  if (!backoff_enabled()) return false;
  DeclineNote* d = decline_note(current_device(), n, k, false);
  if (!d || d->skip <= 0) return false;
  --d->skip;
  return true;
}
static void note_outcome(int64_t n, int64_t k, bool late_decline) {
  if (!backoff_enabled()) return;
  DeclineNote* d = decline_note(current_device(), n, k, late_decline);
  if (!d) return;
  if (late_decline) {
    // (the first late decline of a run changes nothing: layers of one shape alternate in a model -- Llama's q and o
    // are both 4096 x 4096 -- and one of the two declining must not send the other to the direct route; from the
    // second consecutive one on, 1, 2, 4, ... requests are skipped)
    d->fails = std::min(d->fails + 1, 8);
    d->skip = d->fails >= 2 ? 1 << (d->fails - 2) : 0;
  } else {
    d->fails = 0;
    d->skip = 0;
  }
}

void eigh_filtered_forget_declines() { g_declines.clear(); }

// The widest block the route accepts at order n: n / 2 (the subspace must stay well below the matrix order);
// PTD_EIGH_FILTER_BLOCK_EIGHTHS=5 admits 5 n / 8 -- k = n / 2 with a quarter more: 2560 of 4096 -- (experiments)
static int64_t filter_max_block(int64_t n) {
  const char* e = getenv("PTD_EIGH_FILTER_BLOCK_EIGHTHS");
  const int eighths = e ? std::min(6, std::max(1, atoi(e))) : 4;
  return n * eighths / 8;
}

bool eigh_filtered_applies(int64_t n, int64_t k, bool all_values) {
  const char* e = getenv("PTD_EIGH_FILTERED");
  const int mode = e ? atoi(e) : 1;       // 0 off, 1 auto, 2 whenever the shapes allow (tests)
  if (mode == 0 || all_values) return false;
  if (n % 128 != 0 || k < 32) return false;
  if (filter_block(n, k) > filter_max_block(n)) return false;
  if (mode >= 2) return n >= 512;
  // up to 2/7 of the spectrum: at k = n / 3 (n = 4096: a block of 1728 columns, 27 ms of filter rounds, a 14-ms
  // Rayleigh-Ritz problem and, on covariance spectra, a second attempt) the route took 68 ms against 58 ms direct
  return n >= 2048 && 7 * k <= 2 * n;
}

size_t eigh_filtered_workspace_bytes(int64_t n, int64_t k) {
  if (n % 128 != 0 || n < 512 || k < 32 || filter_block(n, k) > filter_max_block(n)) return 0;
  return filter_plan(n, k).total;
}

size_t eigh_filtered_workspace_bytes(int64_t n) {
  // an upper bound over every request the route can accept at this order (PTD_EIGH_FILTERED=2 and
  // PTD_EIGH_FILTER_OVERSAMPLE included): the plan is monotone in the block size and in k, and no block exceeds n / 2
  if (n % 128 != 0 || n < 512) return 0;
  size_t worst = 0;
  for (int64_t k = 32; k <= filter_max_block(n); k += 32)
    if (filter_block(n, k) <= filter_max_block(n)) worst = std::max(worst, filter_plan(n, k).total);
  return worst;
}

// PTD_OK, or PTD_ERR_UNSUPPORTED when the route declines (flat spectrum, a breakdown, residual above tolerance): the
// outputs are then unspecified and the caller takes the direct route.
int eigh_filtered(const double* A, int64_t lda, int64_t n, int64_t k, double* evals, double* evecs, int64_t ldv,
                  void* ws, size_t ws_bytes, ptd_eigh_stats* stats, hipStream_t st) {
  const FilterPlan p = filter_plan(n, k);
  if (ws_bytes < p.total) {
    set_error("eigh_filtered: workspace %zu < required %zu bytes", ws_bytes, p.total);
    return PTD_ERR_WORKSPACE;
  }
  const bool debug = getenv("PTD_JACOBI_DEBUG") != nullptr;
  const int m = p.m;
  char* base = static_cast<char*>(ws);
  double* X = reinterpret_cast<double*>(base + p.off_X);
  double* Y = reinterpret_cast<double*>(base + p.off_Y);
  double* Z = reinterpret_cast<double*>(base + p.off_Z);
  double* H = reinterpret_cast<double*>(base + p.off_H);
  double* G = reinterpret_cast<double*>(base + p.off_G);
  double* Wt = reinterpret_cast<double*>(base + p.off_W);
  double* linv = reinterpret_cast<double*>(base + p.off_linv);
  double* Yk = reinterpret_cast<double*>(base + p.off_Yk);
  double* lam = reinterpret_cast<double*>(base + p.off_lam);
  double* Qc = reinterpret_cast<double*>(base + p.off_lz);
  double* Qp = Qc + (size_t)n * LZ_NV;
  double* Wl = Qp + (size_t)n * LZ_NV;
  double* ab = reinterpret_cast<double*>(base + p.off_ab);
  int* fail = reinterpret_cast<int*>(base + p.off_flags);
  unsigned long long* resid = reinterpret_cast<unsigned long long*>(base + p.off_flags + 64);

  // (the statistics events live in a guard: every return path below destroys them)
  struct Events {
    hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ~Events() {
      for (auto& x : e)
        if (x) (void)hipEventDestroy(x);
    }
  } events;
  hipEvent_t (&ev)[6] = events.e;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->method = 3;
    for (auto& e : ev) PTD_CHECK_HIP(hipEventCreate(&e));
    PTD_CHECK_HIP(hipEventRecord(ev[0], st));
  }
  bool spent = false;     // products with C have been spent: a decline from here on is a late one
  // a decline is not an error: the caller answers on the direct route and the thread's last error stays as it was
  auto decline = [&](const char* why) {
    if (debug) fprintf(stderr, "[eigh_filtered] n=%lld k=%lld declines: %s\n", (long long)n, (long long)k, why);
    if (spent) note_outcome(n, k, true);
    return PTD_ERR_UNSUPPORTED;
  };

  // ---- 1. Lanczos: bounds and density of states
  PTD_CHECK_HIP(hipMemsetAsync(base + p.off_flags, 0, 256, st));
  hipLaunchKernelGGL(fs_lanczos_init_kernel, dim3(LZ_NV), dim3(1024), 0, st, Qc, Qp, Wl, (int)n);
  const bool qlds = (size_t)n * LZ_NV * 8 <= 128 * 1024;
  if (qlds)
    PTD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fs_symv4_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  const int symv_grid = (int)std::min<int64_t>(256, ceil_div(n, 16));
  for (int s = 0; s < LZ_STEPS; ++s) {
    if (qlds)
      hipLaunchKernelGGL((fs_symv4_kernel<true>), dim3(symv_grid), dim3(256), (size_t)n * LZ_NV * 8, st, A, lda, (int)n,
                         Qc, Wl);
    else
      hipLaunchKernelGGL((fs_symv4_kernel<false>), dim3(symv_grid), dim3(256), 0, st, A, lda, (int)n, Qc, Wl);
    hipLaunchKernelGGL(fs_lanczos_step_kernel, dim3(LZ_NV), dim3(1024), 0, st, Qc, Qp, Wl, (int)n, ab, s);
  }
  PTD_CHECK_LAUNCH("eigh_filtered lanczos");
  double h_ab[LZ_NV * 2 * LZ_STEPS];
  PTD_CHECK_HIP(hipMemcpyAsync(h_ab, ab, sizeof(h_ab), hipMemcpyDeviceToHost, st));
  // (the random start of the filter does not depend on the bounds: queue it behind the copy)
  hipLaunchKernelGGL(fs_random_kernel, dim3(2048), dim3(256), 0, st, X, (int64_t)n * m, 0xC0FFEEULL);
  PTD_CHECK_HIP(hipStreamSynchronize(st));
  if (stats) PTD_CHECK_HIP(hipEventRecord(ev[1], st));

  std::vector<std::pair<double, double>> nodes;   // (theta, weight), weights of one chain sum to 1 / NV
  double lo = INFINITY, hi = -INFINITY;
  for (int v = 0; v < LZ_NV; ++v) {
    double d[LZ_STEPS], e[LZ_STEPS], z[LZ_STEPS], y[LZ_STEPS];
    int len = LZ_STEPS;
    for (int i = 0; i < LZ_STEPS; ++i) {
      d[i] = h_ab[v * 2 * LZ_STEPS + i];
      e[i] = h_ab[v * 2 * LZ_STEPS + LZ_STEPS + i];
      z[i] = i == 0 ? 1.0 : 0.0;
      y[i] = 0.0;
      if (!std::isfinite(d[i]) || !std::isfinite(e[i])) return decline("non-finite Lanczos coefficients");
      if (i < len - 1 && !(e[i] > 0.0)) len = i + 1;   // an invariant subspace: the chain ends here
    }
    const double blast = e[len - 1];
    y[len - 1] = 1.0;
    if (!tridiag_ql_first_row(len, d, e, z, y)) return decline("Lanczos tridiagonal did not converge");
    for (int i = 0; i < len; ++i) {
      if (!std::isfinite(d[i]) || !std::isfinite(z[i]) || !std::isfinite(y[i])) return decline("non-finite Ritz values");
      nodes.emplace_back(d[i], z[i] * z[i] / LZ_NV);
    }
    // a Ritz pair's residual is |beta_last| |last component|: an eigenvalue lies within that distance of theta_i
    for (int i = 0; i < len; ++i) {
      lo = std::min(lo, d[i] - fabs(blast * y[i]));
      hi = std::max(hi, d[i] + fabs(blast * y[i]));
    }
  }
  std::sort(nodes.begin(), nodes.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
  // theta with an estimated `count` eigenvalues above it (piecewise linear in the cumulative weight)
  auto quantile = [&](double count) {
    double cum = 0.0, prev_cum = 0.0, prev_theta = nodes.front().first;
    for (const auto& nd : nodes) {
      prev_cum = cum;
      cum += nd.second * (double)n;
      if (cum >= count) {
        const double t = cum > prev_cum ? (count - prev_cum) / (cum - prev_cum) : 1.0;
        return prev_theta + t * (nd.first - prev_theta);
      }
      prev_theta = nd.first;
    }
    return nodes.back().first;
  };
  const double theta_max = nodes.front().first;
  const double a_cut = quantile((double)k + 0.9 * (double)(m - k));
  const double lam_k = quantile(1.05 * (double)k);   // (a little inside: the density estimate is good to ~10 %)
  lo -= 0.01 * std::max(a_cut - lo, 0.0);   // (an eigenvalue just below `lo` would grow like one just above `a`)
  if (!(hi > 0.0) || !(a_cut > lo) || !(lam_k > a_cut) || (a_cut - lo) < 1e-10 * fabs(hi))
    return decline("degenerate spectral bounds");
  const double ee = 0.5 * (a_cut - lo), cc = 0.5 * (a_cut + lo);
  const double xk = (lam_k - cc) / ee;
  const double growth = xk + sqrt(xk * xk - 1.0);
  const double tol = env_double("PTD_EIGH_FILTER_TOL", 1e-10);     // residual / |lambda_max| accepted
  // (up to 32 products the filter still wins: 29 products at m = 1280 cost ~19 ms, six rounds of passes ~12 ms, the
  // Rayleigh-Ritz problem and the rest ~11.5 -- 42 ms against 55 ms for the direct reduction at n = 4096, k = 1024)
  const int max_products = (int)env_double("PTD_EIGH_FILTER_MAX_PRODUCTS", 32);
  // measured on covariance spectra: the residual falls like 0.05 g^-d with g about three quarters of the way from 1
  // to the asymptotic factor (rounds restart the polynomial; the neighbours of the cut grow a little as well)
  const double g_eff = 1.0 + 0.9 * (growth - 1.0);
  int degree = (int)ceil(log(0.1 / tol) / log(g_eff));    // (aims a factor two below the tolerance)
  degree = std::max(degree, 4);
  // test hook: PTD_EIGH_FILTER_FORCE_DEGREE=<d> spends d products before the first Rayleigh-Ritz step whatever the
  // estimate says (an under-provisioned first attempt exercises the retry rounds)
  if (const char* fd = getenv("PTD_EIGH_FILTER_FORCE_DEGREE")) degree = std::max(2, atoi(fd));
  if (debug)
    fprintf(stderr, "[eigh_filtered] n=%lld k=%lld m=%d: lo %.3e a %.3e lambda_k~%.3e hi %.3e growth %.2f/product -> "
                    "%d products\n", (long long)n, (long long)k, m, lo, a_cut, lam_k, hi, growth, degree);
  if (!(growth > 1.0) || degree > max_products) return decline("spectrum too flat for the filter");
  // A round multiplies cond(X) by about g_top^d, g_top the growth per product at the top of the spectrum (~150 on
  // covariance spectra: six products reach 1e13, which the shifted Cholesky-QR pass still takes; a spectrum with a wide
  // gap right below lambda_k grows by 1e4 and more per product and broke the pass down at four).  Rounds of at most
  // dmax = floor(log 2e13 / log g_top) <= 6 products, the last one at most 4 (it sets the final accuracy).
  const double x_top = (hi - cc) / ee;
  const double g_top = x_top + sqrt(std::max(x_top * x_top - 1.0, 0.0));
  // (round 4: 2e13 instead of 3e14 -- at g_top = 255 six products are 2.7e14, and the shifted pass broke down on the
  // third layer of bench.py's stack AFTER 17 products: a late decline, 82 ms for that layer instead of 28)
  const int dmax = (int)std::max(1.0, std::min(6.0, floor(log(2e13) / log(std::max(g_top, 1.0 + 1e-9)))));
  if (debug) fprintf(stderr, "[eigh_filtered] growth at the top %.3g per product: rounds of at most %d\n", g_top, dmax);
  std::vector<int> rounds;
  {
    const int last = std::min(std::min(4, dmax), degree);
    int rest = degree - last;
    const int nr = (int)ceil_div(rest, dmax);
    for (int r = 0; r < nr; ++r) {
      const int d = (int)ceil_div(rest, nr - r);
      rounds.push_back(d);
      rest -= d;
    }
    rounds.push_back(last);
  }

  // ---- 2. filter rounds
  spent = true;
  const int64_t tot2 = (int64_t)n * m / 2;
  const double shift_rel = env_double("PTD_EIGH_FILTER_SHIFT", 6e-13);
  int products = 0, rc = PTD_OK;
  // Out = P^T Q (m x m) with the K range split into slabs that are added in index order: the m x m tiles alone do not
  // fill the chip, and atomics would make the run-to-run rounding -- and with it eigenvector signs -- differ
  const int ksplit_mm = (int)std::max<int64_t>(1, std::min<int64_t>(3, (int64_t)n / m));
  auto gram = [&](const double* P, const double* Q, double* Out, double* scratch) -> int {
    int ns = 1;
    const int64_t slab = (int64_t)m * m;
    // (P^T Q is symmetric for both uses: tiles entirely above the diagonal are not computed; their entries of Out are
    // unspecified and never read: the sweep walks the lower 64 x 64 tiles, H is mirrored from its lower triangle)
    int r2 = gemm_f64_slabs(P, 1, m, Q, m, 1, scratch, m, slab, m, m, n, 1.0, ksplit_mm, &ns, true, st);
    if (r2 != PTD_OK) return r2;
    hipLaunchKernelGGL(fs_sum_slabs_kernel, dim3(1024), dim3(256), 0, st, Out, scratch, slab, ns, slab / 2);
    return PTD_OK;
  };
  const bool checksum = getenv("PTD_FILTER_CHECKSUM") != nullptr;
  // diagonal tiles of the Cholesky sweep: 16 x 16 register leaves + matrix-core blocks (default) or the round-3 form
  const int leaf = env_double("PTD_EIGH_FILTER_LEAF", 1.0) != 0.0 ? 1 : 0;
  auto dump = [&](const char* what, const double* buf, size_t count) {
    if (!checksum) return;
    std::vector<double> hbuf(count);
    (void)hipMemcpyAsync(hbuf.data(), buf, count * 8, hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    unsigned long long x = 0;
    for (size_t i = 0; i < count; ++i) {
      unsigned long long b;
      memcpy(&b, &hbuf[i], 8);
      x = (x ^ b) * 0x100000001B3ULL;
    }
    fprintf(stderr, "[eigh_filtered] checksum %-28s %016llx\n", what, x);
  };
  auto chol_pass = [&](bool shifted) -> int {
    // G = X^T X (K split: the tiles of an m x m product do not fill the chip); sweep: G = L L^T and W = L^-T together
    // (one launch per 64-column panel, L itself is never stored); X <- X W
    int r2 = gram(X, X, G, Z);
    if (r2 != PTD_OK) return r2;
    dump("gram", G, (size_t)m * m);
    if (shifted) hipLaunchKernelGGL(fs_shift_kernel, dim3(1), dim3(1024), 0, st, G, m, shift_rel);
    r2 = chol_sweep(G, m, H, Wt, linv, fail, leaf, st);                            // (H is free until Rayleigh-Ritz)
    if (r2 != PTD_OK) return r2;
    dump("W = L^-T", Wt, (size_t)m * m);
    r2 = gemm_f64(X, m, 1, Wt, m, 1, Y, m, n, m, m, 1.0, false, 1, st);
    std::swap(X, Y);
    dump("X W", X, (size_t)n * m);
    return r2;
  };
  // X <- T_d(..) X, then a shifted Cholesky-QR pass (cond(X) ~ 150^d down to ~1e3) and a clean one (orthonormal to ~1e-11)
  auto filter_round = [&](int d) -> int {
    // Y_1 = s1 / e (C - c) X;  Y_{j+1} = 2 s_{j+1} / e (C - c) Y_j - s_j s_{j+1} Y_{j-1},  s_{j+1} = 1 / (2 / s1 - s_j)
    const double s1 = ee / (hi - cc);
    double sg = s1;
    hipLaunchKernelGGL(fs_axpby_kernel, dim3(2048), dim3(256), 0, st, Y, X, (const double*)nullptr, -cc * s1 / ee, 0.0,
                       tot2);
    int r2 = gemm_f64(A, lda, 1, X, m, 1, Y, m, n, m, n, s1 / ee, true, 1, st);
    if (r2 != PTD_OK) return r2;
    ++products;
    // (X, Y) = (Y_{j-1}, Y_j); Z receives Y_{j+1}
    for (int j = 2; j <= d; ++j) {
      const double sn = 1.0 / (2.0 / s1 - sg);
      hipLaunchKernelGGL(fs_axpby_kernel, dim3(2048), dim3(256), 0, st, Z, Y, X, -2.0 * sn * cc / ee, -sg * sn, tot2);
      r2 = gemm_f64(A, lda, 1, Y, m, 1, Z, m, n, m, n, 2.0 * sn / ee, true, 1, st);
      if (r2 != PTD_OK) return r2;
      ++products;
      double* t = X; X = Y; Y = Z; Z = t;
      sg = sn;
    }
    std::swap(X, Y);   // X = the filtered block
    dump("filtered block", X, (size_t)n * m);
    r2 = chol_pass(true);
    if (r2 == PTD_OK) r2 = chol_pass(false);
    return r2;
  };
  for (size_t ri = 0; ri < rounds.size(); ++ri) {
    rc = filter_round(rounds[ri]);
    if (rc != PTD_OK) return rc;
  }
  PTD_CHECK_LAUNCH("eigh_filtered filter");
  if (stats) PTD_CHECK_HIP(hipEventRecord(ev[2], st));

  double res = 0.0, lmax = 1.0;
  int filter_products = products;
  for (int attempt = 0;; ++attempt) {
    // ---- 3. Rayleigh-Ritz
    rc = gemm_f64(A, lda, 1, X, m, 1, Z, m, n, m, n, 1.0, false, 1, st);           // Z = C X
    if (rc != PTD_OK) return rc;
    ++products;
    rc = gram(X, Z, H, Y);                                                         // H = X^T Z
    if (rc != PTD_OK) return rc;
    hipLaunchKernelGGL(fs_symmetrize_kernel, dim3((unsigned)ceil_div(m, 32), (unsigned)ceil_div(m, 32)), dim3(256), 0,
                       st, H, m);
    if (stats && attempt == 0) PTD_CHECK_HIP(hipEventRecord(ev[3], st));
    dump("H", H, (size_t)m * m);
    rc = eigh_tridiag(H, m, m, k, lam, Yk, k, base + p.off_eigh, p.eigh_bytes, 1e-10, false, nullptr, st);
    if (rc == PTD_ERR_UNSUPPORTED) return decline("clustered Ritz values");
    if (rc != PTD_OK) return rc;
    if (stats && attempt == 0) PTD_CHECK_HIP(hipEventRecord(ev[4], st));
    dump("Y_k", Yk, (size_t)m * k);
    rc = gemm_f64(X, m, 1, Yk, k, 1, evecs, ldv, n, k, m, 1.0, false, 1, st);      // V = X Y_k
    if (rc != PTD_OK) return rc;

    // ---- 4. residuals: C V - V theta = Z Y_k - V theta
    double* T = Y;
    rc = gemm_f64(Z, m, 1, Yk, k, 1, T, k, n, k, m, 1.0, false, 1, st);
    if (rc != PTD_OK) return rc;
    PTD_CHECK_HIP(hipMemsetAsync(resid, 0, 16, st));
    // (G and W are free outside the orthonormalisation passes: column statistics and signs live there)
    hipLaunchKernelGGL(fs_colstat_kernel, dim3((unsigned)ceil_div(k, 64), FS_SLABS), dim3(256), 0, st, T, (int64_t)k, evecs,
                       ldv, lam + (m - k), (int)n, (int)k, G);
    hipLaunchKernelGGL(fs_colfinal_kernel, dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, st, G, evecs, ldv, lam + (m - k),
                       (int)k, resid, Wt);
    struct { int fail; int pad[15]; unsigned long long res, lmax; } h{};
    PTD_CHECK_HIP(hipMemcpyAsync(&h, base + p.off_flags, sizeof(h), hipMemcpyDeviceToHost, st));
    PTD_CHECK_HIP(hipStreamSynchronize(st));
    memcpy(&res, &h.res, 8);
    memcpy(&lmax, &h.lmax, 8);
    if (debug)
      fprintf(stderr, "[eigh_filtered] %d products, chol fail %d, max residual %.3e = %.2e |lambda_max| (theta_max %.3e)\n",
              products, h.fail, res, res / std::max(lmax, 1e-300), theta_max);
    if (h.fail) return decline("Cholesky breakdown in an orthonormalisation pass");
    if (res <= tol * lmax) break;
    // not there yet (the density estimate put lambda_k too high): one more round, sized by the rate measured so far --
    // at most three extra rounds (four Rayleigh-Ritz steps in all)
    const double rel = res / std::max(lmax, 1e-300);
    if (attempt >= 3 || !(rel < 1e-2)) return decline("residual above the tolerance");
    const double rate = pow(0.05 / rel, 1.0 / (double)filter_products);     // over ALL filter products so far
    const int extra = std::min(dmax, std::max(std::min(2, dmax), (int)ceil(log(rel / (0.1 * tol)) / log(std::max(rate, 1.5)))));
    if (debug) fprintf(stderr, "[eigh_filtered] measured %.2f per product: %d more\n", rate, extra);
    // what the rate measured so far says the residual still needs, against what the remaining rounds can deliver (and
    // the product budget): a spectrum that converges at 1.4 per product from 1e-5 needs ~35 more -- three capped rounds
    // (and three Rayleigh-Ritz solves) later the answer would be the same decline
    const double need = log(rel / tol) / log(std::max(rate, 1.05));
    if (need > (double)((3 - attempt) * dmax) || (double)products + need > (double)max_products + dmax)
      return decline("residual above the tolerance and out of reach at the measured rate");
    const int before = products;
    rc = filter_round(extra);
    if (rc != PTD_OK) return rc;
    filter_products += products - before;
  }
  hipLaunchKernelGGL(fs_flip_kernel, dim3(1024), dim3(256), 0, st, evecs, ldv, (int)n, (int)k, Wt);   // signs of the accepted attempt
  // eigenvalues: the k largest at the end of evals[n], NaN below (the convention of ptd_eigh_topk with all_values = 0)
  PTD_CHECK_HIP(hipMemsetAsync(evals, 0xFF, (size_t)(n - k) * 8, st));
  PTD_CHECK_HIP(hipMemcpyAsync(evals + (n - k), lam + (m - k), (size_t)k * 8, hipMemcpyDeviceToDevice, st));
  if (stats) {
    PTD_CHECK_HIP(hipEventRecord(ev[5], st));
    PTD_CHECK_HIP(hipEventSynchronize(ev[5]));
  }
  if (stats) {
    float t01 = 0, t12 = 0, t23 = 0, t34 = 0, t45 = 0;
    (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
    (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
    (void)hipEventElapsedTime(&t23, ev[2], ev[3]);
    (void)hipEventElapsedTime(&t34, ev[3], ev[4]);
    (void)hipEventElapsedTime(&t45, ev[4], ev[5]);
    // ms: 0 Lanczos bounds, 1 filter rounds (products with C + orthonormalisation), 2 the m x m eigenproblem,
    // 3 Rayleigh-Ritz products + back-multiplication + residuals; work[1] = flop of the products with C
    stats->ms[0] = t01;
    stats->ms[1] = t12;
    stats->ms[2] = t34;
    stats->ms[3] = t23 + t45;
    stats->launches[0] = LZ_STEPS;
    stats->launches[1] = products;
    stats->launches[2] = m;
    stats->work[1] = 2.0 * (double)n * (double)n * (double)m * (double)products;
    stats->total_ms = t01 + t12 + t23 + t34 + t45;
  }
  note_outcome(n, k, false);
  return PTD_OK;
}

}  // namespace ptd
