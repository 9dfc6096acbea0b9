// Symmetric eigensolver, tridiagonalisation route (f64):
//   1. blocked Householder reduction A -> T (panels of 64 reflectors): per column one
//      HBM / Infinity-Cache bound SYMV over the trailing matrix (a one-stage reduction streams
//      8/3 n^3 bytes; the symmetric kernel reads the lower triangle only) and one small kernel
//      for everything between two SYMVs, per panel one rank-2*64 update on the f64 matrix cores;
//   2. eigenvalues of T by multisection on Sturm counts (one wave per eigenvalue);
//   3. eigenvectors of T by inverse iteration (one eigenvector per thread, pivoted LU of the
//      shifted tridiagonal), tight clusters re-orthogonalised;
//   4. back-transformation Z = Q Y with compact-WY block reflectors (f64 MFMA GEMMs).
// The full matrix (both triangles) is kept current, so the column a panel step needs is a contiguous row.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include <hip/hip_ext.h>

#include "common.h"
#include "kernels.h"

namespace ptd {

int gemm_f64(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn, double* C,
             int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha, bool beta1, int ksplit, hipStream_t st);
int gemm_f64_pair(const double* A, const double* B, const double* A2, const double* B2, int64_t sam, int64_t sak,
                  int64_t sbk, int64_t sbn, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha,
                  double* row0_out, hipStream_t st, int batch = 1, int64_t bstride_elems = 0);
int gemm_f64_slabs(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn, double* C,
                   int64_t ldc, int64_t slab, int64_t M, int64_t N, int64_t K, double alpha, int ksplit, int* nslabs,
                   bool lower_only, hipStream_t st);

namespace {

constexpr int NB = 64;          // reflectors per panel
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Per column j (reflector i of its panel) TWO multi-workgroup launches:
//   symv(j)     x_j = base_j - delta_j v_{j-1} on the fly;  unscaled products  sd[r] = A[r][j+2:] . x_j[j+2:]
//               for the trailing rows r and, as extra rows of the same product, for W_k and V_k;
//               q[r] = row[j+1];  partial sums of x_j^2  (the reflector's norm)
//   alpha(j+1)  everything between two SYMVs: reflector scalars of column j, v_j, the finalised
//               w_{j-1}, w_raw_j = tau (p - V c1 - W c2), partial sums of w_raw_j . v_j, and the
//               updated column j+1 ("base", see below)
// Global dependencies per column are thereby cut to two (the norm and W^T v / V^T v), the third
// one (alpha2 = -tau/2 w_raw.v, which turns w_raw into w) is deferred algebraically:
//   w_j = w_raw_j + alpha2_j v_j  enters column j+1 as  x_{j+1} = base_{j+1} - 2 alpha2_j v_j
// (v_j[j+1] = 1), and enters W^T v as  c1[i-1] = w_raw . v + alpha2 (v_{j-1} . v);  both
// corrections are applied by the consumer, which can sum the producer's partial dot products.
// colbuf is indexed by the GLOBAL row / column index (zero padding beyond n keeps 16-byte loads
// aligned and harmless).

// sum of `count` partial values, computed by one full wave (call with all 64 lanes of a wave)
__device__ __forceinline__ double wave_total(const double* __restrict__ v, int count, int lane) {
  double t = 0.0;
  for (int k = lane; k < count; k += 64) t += v[k];
  return wave_sum_d(t);
}

struct ColState {
  double delta;  // 2 alpha2 of the previous column of the panel (0 for the panel's first column)
  double alpha;  // x_j[j+1]
};

// Batched launches (eigh_tridiag_batched): blockIdx.y is the matrix, every workspace pointer of the launch moves by
// y * bstride bytes (the workspaces of the batch are copies of one plan, bstride apart; bstride = 0: one matrix).
template <typename T>
__device__ __forceinline__ T* batch_ptr(T* p, size_t boff) {
  return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + boff);
}
#define PTD_BATCH(P) P = batch_ptr(P, boff_)

// first column of a panel: base = A[j][j:]
__global__ void sytrd_colinit_kernel(const double* __restrict__ A, int64_t ld, int n, int j,
                                     double* __restrict__ colbuf, size_t bstride) {
  const size_t boff_ = (size_t)blockIdx.y * bstride;
  PTD_BATCH(A); PTD_BATCH(colbuf);
  const int r = j + blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) colbuf[r] = A[(int64_t)j * ld + r];
}

constexpr int SROWS = 4;  // rows per workgroup; its four waves split the columns

// symv: rows 0 .. m-1 are matrix rows j+1 .. n-1, rows m .. m+i-1 are W_k (k = i-1: w_raw of the
// previous column), rows m+i .. m+2i-1 are V_k.
__global__ __launch_bounds__(256) void sytrd_symv_kernel(const double* __restrict__ A, int64_t ld, int n, int j,
                                                         int i, const double* __restrict__ colbuf,
                                                         const double* __restrict__ Vp,
                                                         const double* __restrict__ Wp, int64_t ldv,
                                                         const double* __restrict__ wraw_prev,
                                                         const double* __restrict__ partial2, int nparts2,
                                                         const double* __restrict__ taus,
                                                         double* __restrict__ sd, double* __restrict__ qv,
                                                         double* __restrict__ cb, double* __restrict__ px2,
                                                         ColState* __restrict__ cs, size_t bstride) {
  __shared__ double part[4][SROWS];
  __shared__ double delta_s;
  const size_t boff_ = (size_t)blockIdx.y * bstride;
  PTD_BATCH(A); PTD_BATCH(colbuf); PTD_BATCH(Vp); PTD_BATCH(Wp); PTD_BATCH(wraw_prev); PTD_BATCH(partial2);
  PTD_BATCH(taus); PTD_BATCH(sd); PTD_BATCH(qv); PTD_BATCH(cb); PTD_BATCH(px2); PTD_BATCH(cs);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (wid == 0) {
    double dl = 0.0;
    if (i > 0) {
      const double t = wave_total(partial2, nparts2, lane);
      dl = -taus[j - 1] * t;  // 2 * alpha2_{j-1}
    }
    if (lane == 0) delta_s = dl;
  }
  __syncthreads();
  const double delta = delta_s;
  const double* vprev = Vp + (int64_t)max(i - 1, 0) * ldv;  // multiplied by delta == 0 when i == 0
  const int m = n - j - 1;
  const int total = m + 2 * i;
  const int g0 = blockIdx.x * SROWS;
  const double* rp[SROWS];
  double acc[SROWS];
#pragma unroll
  for (int t = 0; t < SROWS; ++t) {
    const int g = min(g0 + t, total - 1);
    if (g < m) rp[t] = A + (int64_t)(j + 1 + g) * ld;
    else if (g < m + i) rp[t] = (g - m == i - 1) ? wraw_prev : Wp + (int64_t)(g - m) * ldv;
    else rp[t] = Vp + (int64_t)(g - m - i) * ldv;
    acc[t] = 0.0;
  }
  const int cstart = (j + 2) & ~1;  // even start: 16-byte aligned loads
  int c = cstart + 2 * lane + 128 * wid;
  for (; c + 512 < n; c += 1024) {
    double2 xa = *reinterpret_cast<const double2*>(colbuf + c);
    double2 xb = *reinterpret_cast<const double2*>(colbuf + c + 512);
    const double2 va = *reinterpret_cast<const double2*>(vprev + c);
    const double2 vb = *reinterpret_cast<const double2*>(vprev + c + 512);
    xa.x -= delta * va.x; xa.y -= delta * va.y;
    xb.x -= delta * vb.x; xb.y -= delta * vb.y;
    if (c < j + 2) xa.x = 0.0;
    double2 ra[SROWS], rb[SROWS];
#pragma unroll
    for (int t = 0; t < SROWS; ++t) {
      ra[t] = *reinterpret_cast<const double2*>(rp[t] + c);
      rb[t] = *reinterpret_cast<const double2*>(rp[t] + c + 512);
    }
#pragma unroll
    for (int t = 0; t < SROWS; ++t) acc[t] += (ra[t].x * xa.x + ra[t].y * xa.y) + (rb[t].x * xb.x + rb[t].y * xb.y);
  }
  for (; c < n; c += 512) {
    double2 xa = *reinterpret_cast<const double2*>(colbuf + c);
    const double2 va = *reinterpret_cast<const double2*>(vprev + c);
    xa.x -= delta * va.x; xa.y -= delta * va.y;
    if (c < j + 2) xa.x = 0.0;
    if (c + 1 >= n) xa.y = 0.0;
#pragma unroll
    for (int t = 0; t < SROWS; ++t) {
      const double2 r2 = *reinterpret_cast<const double2*>(rp[t] + c);
      acc[t] += r2.x * xa.x + r2.y * xa.y;
    }
  }
#pragma unroll
  for (int t = 0; t < SROWS; ++t) {
    const double sdot = wave_sum_d(acc[t]);
    if (lane == 0) part[wid][t] = sdot;
  }
  __syncthreads();
  if (wid == 0) {
    double x2 = 0.0;
    if (lane < SROWS) {
      const int t = lane, g = g0 + t;
      if (g < total) {
        const double sdot = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        if (g < m) {
          const int row = j + 1 + g;
          sd[row] = sdot;
          qv[row] = A[(int64_t)row * ld + j + 1];
          if (row >= j + 2) {
            const double xr = colbuf[row] - delta * vprev[row];
            x2 = xr * xr;
          }
        } else if (g < m + i) {
          const int k = g - m;
          const double* row = (k == i - 1) ? wraw_prev : Wp + (int64_t)k * ldv;
          cb[k] = sdot;
          cb[NB + k] = row[j + 1];
        } else {
          const int k = g - m - i;
          cb[2 * NB + k] = sdot;
          cb[3 * NB + k] = Vp[(int64_t)k * ldv + j + 1];
        }
      }
    }
    x2 = wave_sum_d(x2);
    if (lane == 0) {
      px2[blockIdx.x] = x2;
      if (blockIdx.x == 0) {
        cs->delta = delta;
        cs->alpha = colbuf[j + 1] - delta * vprev[j + 1];
      }
    }
  }
}

// ---- symmetric SYMV: only the lower triangle of the trailing matrix is read.
// The trailing square is cut on an ABSOLUTE grid of TR x TC tiles; a lower-triangle tile (I, J)
// contributes  A_IJ x_J  to the rows of I ("row part", reduced over the lanes of a wave) and
// A_IJ^T x_I  to the columns of J ("column part", accumulated per lane), strictly below the
// diagonal for the latter.  Partial results go to rowpart[J][r] / colpart[I][c]; the consumer
// (alpha kernel) adds, for row r, rowpart[J0 .. r/TC][r] + colpart[r/TR .. nI-1][r] in a fixed
// order, so the result does not depend on scheduling.  Tile rows come in groups of TQ that share
// their range of J; group u (counted from the first group that touches row j+1) has TQ (u+1) tiles.
// Behind the tiles the same launch carries the V_k / W_k dot products (4 rows per workgroup) and
// the strided column read q[r] = A[r][j+1] with the partial sums of x^2 (256 rows per workgroup).
constexpr int TR = 64, TC = 256, TQ = TC / TR;

struct SymPart {
  double* rowpart;
  double* colpart;
  int64_t ldp;
  int nI;       // tile rows: ceil(n / TR)
  int enabled;
};

// p[16] per lane -> sum over the 64 lanes of row (lane >> 2), returned in every lane
__device__ __forceinline__ double transpose_reduce16(double (&p)[16], int lane) {
  {
    const bool hi = lane & 32;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const double send = hi ? p[t] : p[t + 8];
      const double keep = hi ? p[t + 8] : p[t];
      p[t] = keep + __shfl_xor(send, 32);
    }
  }
  {
    const bool hi = lane & 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double send = hi ? p[t] : p[t + 4];
      const double keep = hi ? p[t + 4] : p[t];
      p[t] = keep + __shfl_xor(send, 16);
    }
  }
  {
    const bool hi = lane & 8;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double send = hi ? p[t] : p[t + 2];
      const double keep = hi ? p[t + 2] : p[t];
      p[t] = keep + __shfl_xor(send, 8);
    }
  }
  {
    const bool hi = lane & 4;
    const double send = hi ? p[0] : p[1];
    const double keep = hi ? p[1] : p[0];
    p[0] = keep + __shfl_xor(send, 4);
  }
  p[0] += __shfl_xor(p[0], 2);
  p[0] += __shfl_xor(p[0], 1);
  return p[0];
}

template <bool GEN>
__device__ __forceinline__ void symv2_tile_body(const double2 (&a0)[16], const double2 (&a1)[16], const double2 xa,
                                                const double2 xb, const double* __restrict__ xs, int rbase, int nclamp,
                                                int cA, int cB, double (&p)[16], double (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const double xr = xs[t];
    double r0 = a0[t].x, r1 = a0[t].y, r2 = a1[t].x, r3 = a1[t].y;  // row part: c <= R
    double c0 = r0, c1 = r1, c2 = r2, c3 = r3;                      // column part: c < R
    if (GEN) {
      const int R = min(rbase + t, nclamp);
      r0 = (cA <= R) ? r0 : 0.0;     c0 = (cA < R) ? c0 : 0.0;
      r1 = (cA + 1 <= R) ? r1 : 0.0; c1 = (cA + 1 < R) ? c1 : 0.0;
      r2 = (cB <= R) ? r2 : 0.0;     c2 = (cB < R) ? c2 : 0.0;
      r3 = (cB + 1 <= R) ? r3 : 0.0; c3 = (cB + 1 < R) ? c3 : 0.0;
    }
    p[t] = (r0 * xa.x + r1 * xa.y) + (r2 * xb.x + r3 * xb.y);
    acc[0] += c0 * xr; acc[1] += c1 * xr; acc[2] += c2 * xr; acc[3] += c3 * xr;
  }
}

#ifdef PTD_RES_PROF
__device__ int symv_dbg;      // probe build: 1 return at once, 2 return when the tile has arrived, 4 no partial-sum stores
extern "C" void ptd_debug_symv(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(symv_dbg), &v, sizeof(int)); }
#define SYMV_DBG(BIT) (symv_dbg & (BIT))
#else
#define SYMV_DBG(BIT) false
#endif
__global__ __launch_bounds__(256) void sytrd_symv2_kernel(const double* __restrict__ A, int64_t ld, int n, int j, int i,
                                                          const double* __restrict__ colbuf,
                                                          const double* __restrict__ Vp,
                                                          const double* __restrict__ Wp, int64_t ldv,
                                                          const double* __restrict__ wraw_prev,
                                                          const double* __restrict__ partial2, int nparts2,
                                                          const double* __restrict__ taus, SymPart sp, int ntiles,
                                                          int nextra, double* __restrict__ qv,
                                                          double* __restrict__ cb, double* __restrict__ px2,
                                                          ColState* __restrict__ cs, size_t bstride) {
  const size_t boff_ = (size_t)blockIdx.y * bstride;
  PTD_BATCH(A); PTD_BATCH(colbuf); PTD_BATCH(Vp); PTD_BATCH(Wp); PTD_BATCH(wraw_prev); PTD_BATCH(partial2);
  PTD_BATCH(taus); PTD_BATCH(sp.rowpart); PTD_BATCH(sp.colpart); PTD_BATCH(qv); PTD_BATCH(cb); PTD_BATCH(px2);
  PTD_BATCH(cs);
  __shared__ __attribute__((aligned(16))) double xs[TR];
  __shared__ __attribute__((aligned(16))) double colred[4][TC];
  __shared__ double part[4][SROWS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x;
  const double* vprev = Vp + (int64_t)max(i - 1, 0) * ldv;  // multiplied by delta == 0 when i == 0
  if (SYMV_DBG(1)) return;
  if (b < ntiles) {
    // ---- tile (I, J)
    const int g0 = (j + 1) / TC;
    int u = (int)((sqrtf(2.f * (float)b + 1.f) - 1.f) * 0.5f);
    while (TQ / 2 * u * (u + 1) > b) --u;
    while (TQ / 2 * (u + 1) * (u + 2) <= b) ++u;
    const int rem = b - TQ / 2 * u * (u + 1);
    const int I = TQ * (g0 + u) + rem / (u + 1), J = g0 + rem % (u + 1);
    const int R0 = I * TR, C0 = J * TC;
    if (R0 + TR - 1 < j + 1 || R0 >= n) return;
    const int cA = C0 + 2 * lane, cB = cA + 128;
    const int rbase = R0 + 16 * wid;
    double2 a0[16], a1[16];
    // A 128-column half of the tile is not fetched when nothing in it is read: left of column j + 1 (the tile column
    // that holds it starts at a multiple of 256: on average 128 dead columns over all m rows) or right of the tile's
    // last row (the upper halves of the 256 x 256 diagonal blocks).  Wave-uniform tests; 0.64 -> 0.59 of the full-square
    // bytes at n = 4096.
    // ... and inside a live half a lane whose two columns lie left of column j + 1 asks for nothing (whole 128-byte lines
    // at the left edge of the first tile column are then not requested), nor -- in the tiles on the diagonal -- for the
    // rows above its columns.
    const bool liveA = (C0 + 127 >= j + 1) && (C0 <= R0 + TR - 1) && (cA + 1 >= j + 1);
    const bool liveB = (C0 + 255 >= j + 1) && (C0 + 128 <= R0 + TR - 1) && (cB + 1 >= j + 1);
    if (C0 + TC - 1 > R0) {      // a tile on the diagonal: elements with c > r are never read
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int rr = min(rbase + t, n - 1);
        const double* rowp = A + (int64_t)rr * ld;
        a0[t] = (liveA && cA <= rr) ? *reinterpret_cast<const double2*>(rowp + cA) : double2{0.0, 0.0};
        a1[t] = (liveB && cB <= rr) ? *reinterpret_cast<const double2*>(rowp + cB) : double2{0.0, 0.0};
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const double* rowp = A + (int64_t)min(rbase + t, n - 1) * ld;
        a0[t] = liveA ? *reinterpret_cast<const double2*>(rowp + cA) : double2{0.0, 0.0};
        a1[t] = liveB ? *reinterpret_cast<const double2*>(rowp + cB) : double2{0.0, 0.0};
      }
    }
    if (SYMV_DBG(2)) {
      double sink = 0.0;
#pragma unroll
      for (int t = 0; t < 16; ++t) sink += a0[t].x + a0[t].y + a1[t].x + a1[t].y;
      if (sink == 1.2345e-300) xs[0] = sink;
      return;
    }
    double2 xa = *reinterpret_cast<const double2*>(colbuf + cA);
    double2 xb = *reinterpret_cast<const double2*>(colbuf + cB);
    const double2 va = *reinterpret_cast<const double2*>(vprev + cA);
    const double2 vb = *reinterpret_cast<const double2*>(vprev + cB);
    double xrow = 0.0, vrow = 0.0;
    const int Rx = R0 + tid;
    const bool rowok = tid < TR && Rx >= j + 2 && Rx < n;
    if (rowok) { xrow = colbuf[Rx]; vrow = vprev[Rx]; }
    double delta = 0.0;
    if (i > 0) delta = -taus[j - 1] * wave_total(partial2, nparts2, lane);  // 2 * alpha2_{j-1}
    if (tid < TR) xs[tid] = rowok ? xrow - delta * vrow : 0.0;
    xa.x = (cA >= j + 2) ? xa.x - delta * va.x : 0.0;
    xa.y = (cA + 1 >= j + 2) ? xa.y - delta * va.y : 0.0;
    xb.x = (cB >= j + 2) ? xb.x - delta * vb.x : 0.0;
    xb.y = (cB + 1 >= j + 2) ? xb.y - delta * vb.y : 0.0;
    __syncthreads();
    double p[16], acc[4] = {0.0, 0.0, 0.0, 0.0};
    const bool general = (C0 + TC - 1 > R0) || (C0 + TC > n);
    if (general) symv2_tile_body<true>(a0, a1, xa, xb, xs + 16 * wid, rbase, n - 1, cA, cB, p, acc);
    else symv2_tile_body<false>(a0, a1, xa, xb, xs + 16 * wid, rbase, n - 1, cA, cB, p, acc);
    const double rowsum = transpose_reduce16(p, lane);
    if ((lane & 3) == 0) sp.rowpart[(int64_t)J * sp.ldp + rbase + (lane >> 2)] = rowsum;
    *reinterpret_cast<double2*>(&colred[wid][2 * lane]) = double2{acc[0], acc[1]};
    *reinterpret_cast<double2*>(&colred[wid][128 + 2 * lane]) = double2{acc[2], acc[3]};
    __syncthreads();
    sp.colpart[(int64_t)I * sp.ldp + C0 + tid] =
        (colred[0][tid] + colred[1][tid]) + (colred[2][tid] + colred[3][tid]);
    return;
  }
  double delta = 0.0;
  if (i > 0) delta = -taus[j - 1] * wave_total(partial2, nparts2, lane);
  if (b < ntiles + nextra) {
    // ---- dot products of W_k (k < i; k == i-1 is w_raw of the previous column) and V_k with x
    const int total = 2 * i;
    const int g0 = (b - ntiles) * SROWS;
    const double* rp[SROWS];
    double acc[SROWS];
#pragma unroll
    for (int t = 0; t < SROWS; ++t) {
      const int g = min(g0 + t, total - 1);
      if (g < i) rp[t] = (g == i - 1) ? wraw_prev : Wp + (int64_t)g * ldv;
      else rp[t] = Vp + (int64_t)(g - i) * ldv;
      acc[t] = 0.0;
    }
    const int cstart = (j + 2) & ~1;
    for (int c = cstart + 2 * lane + 128 * wid; c < n; c += 512) {
      double2 xa = *reinterpret_cast<const double2*>(colbuf + c);
      const double2 va = *reinterpret_cast<const double2*>(vprev + c);
      xa.x -= delta * va.x; xa.y -= delta * va.y;
      if (c < j + 2) xa.x = 0.0;
      if (c + 1 >= n) xa.y = 0.0;
#pragma unroll
      for (int t = 0; t < SROWS; ++t) {
        const double2 r2 = *reinterpret_cast<const double2*>(rp[t] + c);
        acc[t] += r2.x * xa.x + r2.y * xa.y;
      }
    }
#pragma unroll
    for (int t = 0; t < SROWS; ++t) {
      const double sdot = wave_sum_d(acc[t]);
      if (lane == 0) part[wid][t] = sdot;
    }
    __syncthreads();
    if (tid < SROWS && g0 + tid < total) {
      const int g = g0 + tid;
      const double sdot = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
      if (g < i) {
        const double* row = (g == i - 1) ? wraw_prev : Wp + (int64_t)g * ldv;
        cb[g] = sdot;
        cb[NB + g] = row[j + 1];
      } else {
        const int k = g - i;
        cb[2 * NB + k] = sdot;
        cb[3 * NB + k] = Vp[(int64_t)k * ldv + j + 1];
      }
    }
    return;
  }
  // ---- q[r] = A[r][j+1] and the partial sums of x^2 over 256 rows
  {
    const int chunk = b - ntiles - nextra;
    const int r = j + 1 + chunk * 256 + tid;
    double x2 = 0.0;
    if (r < n) {
      qv[r] = A[(int64_t)r * ld + j + 1];
      if (r >= j + 2) {
        const double xr = colbuf[r] - delta * vprev[r];
        x2 = xr * xr;
      }
    }
    x2 = wave_sum_d(x2);
    if (lane == 0) part[wid][0] = x2;
    __syncthreads();
    if (tid == 0) {
      px2[chunk] = (part[0][0] + part[1][0]) + (part[2][0] + part[3][0]);
      if (chunk == 0) {
        cs->delta = delta;
        cs->alpha = colbuf[j + 1] - delta * vprev[j + 1];
      }
    }
  }
}

// alpha(jn): finishes column j = jn - 1 (panel index i = in - 1 >= 0) and, if do_next, forms the
// updated column jn.  Grid ceil((n - jn) / 64) x 256 threads; lane = row, the four waves split k.
__global__ __launch_bounds__(256) void sytrd_alpha_kernel(const double* __restrict__ A, int64_t ld, int n, int jn,
                                                          int in, int do_next, double* __restrict__ Vp,
                                                          double* __restrict__ Wp, int64_t ldv,
                                                          const double* __restrict__ wraw_prev,
                                                          double* __restrict__ wraw_cur,
                                                          double* __restrict__ colbuf, const double* __restrict__ sd,
                                                          const double* __restrict__ qv,
                                                          const double* __restrict__ cb,
                                                          const double* __restrict__ px2, int npx2,
                                                          const ColState* __restrict__ cs,
                                                          double* __restrict__ partial2, double* __restrict__ d,
                                                          double* __restrict__ e, double* __restrict__ taus,
                                                          SymPart sp, size_t bstride) {
  const size_t boff_ = (size_t)blockIdx.y * bstride;
  PTD_BATCH(A); PTD_BATCH(Vp); PTD_BATCH(Wp); PTD_BATCH(wraw_prev); PTD_BATCH(wraw_cur); PTD_BATCH(colbuf);
  PTD_BATCH(sd); PTD_BATCH(qv); PTD_BATCH(cb); PTD_BATCH(px2); PTD_BATCH(cs); PTD_BATCH(partial2); PTD_BATCH(d);
  PTD_BATCH(e); PTD_BATCH(taus); PTD_BATCH(sp.rowpart); PTD_BATCH(sp.colpart);
  __shared__ double c1[NB], c2[NB], wj1[NB], vj1[NB];
  __shared__ double part1[4][64], part2[4][64];
  __shared__ double vs[64], wfs[64];
  __shared__ double red[4], red2[4];
  __shared__ double sdp[4][64];
  __shared__ double wrj_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = jn - 1, i = in - 1;
  const int r = jn + blockIdx.x * 64 + lane;
  // ---- phase 0: every global operand of the prologue is requested up front (one latency)
  double px = 0.0;
  for (int q = tid; q < npx2; q += 256) px += px2[q];
  const double delta = cs->delta, alpha = cs->alpha;
  double r_sdW = 0.0, r_qW = 0.0, r_sdV = 0.0, r_qV = 0.0, r_vj1 = 0.0, r_wj1 = 0.0, r_vpjn = 0.0;
  if (tid < i) {
    r_sdW = cb[tid]; r_qW = cb[NB + tid]; r_sdV = cb[2 * NB + tid]; r_qV = cb[3 * NB + tid];
    r_vj1 = Vp[(int64_t)tid * ldv + jn];
    r_wj1 = (tid == i - 1) ? wraw_prev[jn] : Wp[(int64_t)tid * ldv + jn];
    if (tid == i - 1) r_vpjn = Vp[(int64_t)(i - 1) * ldv + jn];
  }
  double r_col = 0.0, r_vprev = 0.0, r_wprev = 0.0, r_qv = 0.0, r_a = 0.0;
  if (wid == 0 && r < n) {
    r_col = colbuf[r];
    if (i > 0) { r_vprev = Vp[(int64_t)(i - 1) * ldv + r]; r_wprev = wraw_prev[r]; }
    r_qv = qv[r];
    if (do_next) r_a = A[(int64_t)jn * ld + r];
  }
  // the panel rows V_k[r], W_k[r] of the two corrections (wave wid takes k = wid, wid + 4, ...):
  // requested here, used after the second barrier (W_{i-1} is finalised in this launch: wfs)
  double pv[NB / 4], pw[NB / 4];
#pragma unroll
  for (int q = 0; q < NB / 4; ++q) {
    const int k = wid + 4 * q;
    pv[q] = 0.0; pw[q] = 0.0;
    if (k < i && r < n) {
      pv[q] = Vp[(int64_t)k * ldv + r];
      if (k < i - 1) pw[q] = Wp[(int64_t)k * ldv + r];
    }
  }
  // unscaled SYMV products of this workgroup's rows and of row jn: either sd[] itself or, after a
  // symmetric SYMV, the fixed-order sum of its row / column partials (the four waves split the list)
  double sdacc = 0.0, sdjn = 0.0;
  if (sp.enabled) {
    const int J0 = jn / TC;
    if (r < n) {
      const int nrow = r / TC - J0 + 1, i0 = r / TR, cnt = nrow + sp.nI - i0;
      // (up to 16 + n / 64 partials a row: at n <= 4096 every wave asks for all of its 20 at once -- in batches of 8 the
      // three dependent round trips to memory were ~0.8 us of this launch)
      constexpr int SDU = 20;
      for (int k0 = wid; k0 < cnt; k0 += 4 * SDU) {
        double v[SDU];
#pragma unroll
        for (int q = 0; q < SDU; ++q) {
          const int k = k0 + 4 * q;
          v[q] = 0.0;
          if (k < cnt)
            v[q] = (k < nrow) ? sp.rowpart[(int64_t)(J0 + k) * sp.ldp + r]
                              : sp.colpart[(int64_t)(i0 + k - nrow) * sp.ldp + r];
        }
        double s4[SDU / 4];
#pragma unroll
        for (int q = 0; q < SDU / 4; ++q) s4[q] = (v[4 * q] + v[4 * q + 1]) + (v[4 * q + 2] + v[4 * q + 3]);
        sdacc += ((s4[0] + s4[1]) + (s4[2] + s4[3])) + s4[4];
      }
    }
    const int i0 = jn / TR, cnt = 1 + sp.nI - i0;
    for (int k = tid; k < cnt; k += 256)
      sdjn += (k == 0) ? sp.rowpart[(int64_t)J0 * sp.ldp + jn] : sp.colpart[(int64_t)(i0 + k - 1) * sp.ldp + jn];
  } else {
    if (wid == 0 && r < n) sdacc = sd[r];
    if (tid == 0) sdjn = sd[jn];
  }
  const double qv_jn = qv[jn];
  px = wave_sum_d(px);
  sdjn = wave_sum_d(sdjn);
  if (lane == 0) { red[wid] = px; red2[wid] = sdjn; }
  sdp[wid][lane] = sdacc;
  __syncthreads();
  const double sd_jn = (red2[0] + red2[1]) + (red2[2] + red2[3]);
  const double r_sd = (sdp[0][lane] + sdp[1][lane]) + (sdp[2][lane] + sdp[3][lane]);
  // ---- reflector scalars of column j (every thread computes the same values)
  const double xn2 = (red[0] + red[1]) + (red[2] + red[3]);
  double tau, beta, scale;
  if (xn2 == 0.0) {
    tau = 0.0; beta = alpha; scale = 0.0;
  } else {
    const double nrm = sqrt(alpha * alpha + xn2);
    beta = alpha >= 0.0 ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  const double a2prev = 0.5 * delta;
  if (blockIdx.x == 0 && tid == 0) {
    taus[j] = tau;
    e[j] = beta;
    d[j] = colbuf[j] - delta;  // x_j[j] = base_j[j] - delta v_{j-1}[j], v_{j-1}[j] = 1 (delta = 0 if i == 0)
  }
  if (tid < NB) {
    // coefficient k is stored at (k & 3) * 16 + (k >> 2): the sixteen a wave needs below (k = wid, wid + 4, ...)
    // are contiguous, and the entries k >= i are ZERO (their registers were initialised to zero), so the panel
    // dot products need neither a bound check nor a branch per term
    const double c2v = r_qV + scale * r_sdV;
    double c1v = r_qW + scale * r_sdW;
    double wj = r_wj1;
    if (tid == i - 1) {  // W_{i-1} = w_raw + alpha2 v_{j-1}
      c1v += a2prev * c2v;
      wj += a2prev * r_vpjn;
    }
    const int pos = (tid & 3) * 16 + (tid >> 2);
    c1[pos] = c1v; c2[pos] = c2v; vj1[pos] = r_vj1; wj1[pos] = wj;
  }
  if (wid == 0) {
    double vr = 0.0, wf = 0.0;
    if (r < n) {
      const double xr = r_col - delta * r_vprev;
      vr = (r == jn) ? 1.0 : xr * scale;
      Vp[(int64_t)i * ldv + r] = vr;
      if (i > 0) {
        wf = r_wprev + a2prev * r_vprev;
        Wp[(int64_t)(i - 1) * ldv + r] = wf;
      }
    }
    vs[lane] = vr;
    wfs[lane] = wf;
  }
  __syncthreads();
  if (wid == 1) {  // w_raw_j[jn], needed by every row of the next column
    double sacc = 0.0;
    static_assert(NB == 64, "one coefficient per lane");
    const int pos = (lane & 3) * 16 + (lane >> 2);   // lane = k; zeros beyond k = i - 1
    sacc = vj1[pos] * c1[pos] + wj1[pos] * c2[pos];
    sacc = wave_sum_d(sacc);
    if (lane == 0) wrj_s = tau * ((qv_jn + scale * sd_jn) - sacc);
  }
  // straight-line: with a bound check per term the sixteen iterations each paid an LDS round trip behind a
  // branch (2,467 of the launch's ~13,000 cycles, clock64); here the 64 coefficient reads issue back to back
  double a1 = 0.0, a2 = 0.0;
  const double wfl = wfs[lane];
  const double* cc1 = c1 + wid * 16;
  const double* cc2 = c2 + wid * 16;
  const double* cw = wj1 + wid * 16;
  const double* cv = vj1 + wid * 16;
#pragma unroll
  for (int q = 0; q < NB / 4; ++q) {
    const int k = wid + 4 * q;
    const double vk = pv[q];                           // zero for k >= i
    const double wk = (k == i - 1) ? wfl : pw[q];      // pw is zero for k >= i - 1
    a1 += vk * cc1[q] + wk * cc2[q];
    a2 += vk * cw[q] + wk * cv[q];
  }
  part1[wid][lane] = a1;
  part2[wid][lane] = a2;
  __syncthreads();
  if (wid == 0) {
    double dot = 0.0;
    if (r < n) {
      const double s1 = (part1[0][lane] + part1[1][lane]) + (part1[2][lane] + part1[3][lane]);
      const double wr = tau * ((r_qv + scale * r_sd) - s1);
      wraw_cur[r] = wr;
      dot = wr * vs[lane];
      if (do_next) {
        const double s2 = (part2[0][lane] + part2[1][lane]) + (part2[2][lane] + part2[3][lane]);
        colbuf[r] = r_a - s2 - (vs[lane] * wrj_s + wr);
      }
    }
    dot = wave_sum_d(dot);
    if (lane == 0) partial2[blockIdx.x] = dot;
  }
}

// end of panel: w_last = w_raw + alpha2 v_last on the rows the rank-2k update reads
__global__ void sytrd_wfix_kernel(int n, int r0, int ilast, int jlast, const double* __restrict__ Vp,
                                  double* __restrict__ Wp, int64_t ldv, const double* __restrict__ wraw,
                                  const double* __restrict__ partial2, int nparts2,
                                  const double* __restrict__ taus, size_t bstride) {
  const size_t boff_ = (size_t)blockIdx.y * bstride;
  PTD_BATCH(Vp); PTD_BATCH(Wp); PTD_BATCH(wraw); PTD_BATCH(partial2); PTD_BATCH(taus);
  __shared__ double a2s;
  if (threadIdx.x < 64) {
    const double t = wave_total(partial2, nparts2, threadIdx.x);
    if (threadIdx.x == 0) a2s = -0.5 * taus[jlast] * t;
  }
  __syncthreads();
  const int r = r0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) Wp[(int64_t)ilast * ldv + r] = wraw[r] + a2s * Vp[(int64_t)ilast * ldv + r];
}

// ------------------------------------------------------------------------------------- resident kernels
// The last columns of the reduction with the trailing block resident in registers: what a column costs there is no
// longer its bytes but its global dependencies -- two dependent launches of ~5 us each on the blocked path, whatever
// their size.  The unblocked reduction (dsytd2) needs ONE all-to-all hand-off per column (see the loop), and what a
// hand-off costs depends on where it happens:
//   * workgroups of one XCD share an L2, the vector L1 is write-through: a plain store of one CU is seen by a
//     device-scope (sc1) load of another as soon as it is acknowledged, without the write-back / invalidate of an
//     agent-scope release / acquire pair (tools/probes/xcd_barrier_probe.hip: a barrier through that L2 1.2 us, with
//     fences 5.8 us).  <32, 768, false>: 256 workgroups of 150 KB LDS (one per CU), the 32 with blockIdx.x % 8 == 0 --
//     XCC 0, checked against the hardware id -- take the last 768 columns: 4.6 us per column against 11.5.
//   * across XCDs entries and sequence numbers are written through to memory (sc1 stores): a hand-off costs 5.3 us.
//     <256, 2048, true>: every CU, from a trailing order of 2048 down to 1024 (then the block goes back to memory for
//     the one-XCD kernels): 9.0 us per column against ~16.4 blocked.  <32, 1024, false> (four rows a wave, 249 VGPRs,
//     no spill) takes 1024 .. 769 on one XCD at ~4.9 us per column and hands over to <32, 768, false>, which is the
//     faster of the two below 768 (round 4: n = 1280 9.25 -> 8.52 ms, n = 1024 5.78 -> 4.97 ms).
//   * sytrd_resident3_kernel (further down): the same on every CU for trailing orders 3072 .. 2049, three half rows
//     per wave: ~9.1 us per column (14.7 before round 5 took the predicates out of its pass).
//   * sytrd_resident4_kernel<3328> / <3584> (further down): trailing orders 3584 .. 3073 on four waves a CU with 512
//     registers a lane, quarter rows: ~11.5 / ~13.5 us per column against ~22 blocked.
// (Round-5 figures per column, tools/probes/res_prof.py: whole chip <256, 2048> 7.8 us, one XCD 4.0 and 3.2 us.)
// Every spin is bounded; a time-out or an XCC mismatch sets the status word and the host repeats the reduction on the
// blocked path.  ptd_eigh_topk n = 4096, k = 1024: 68.2 -> 54.3 ms (with the twisted-factorisation eigenvectors);
// n = 768: 9.0 -> 4.3 ms.
constexpr int RES_MAX = 768;                             // one XCD: 32 workgroups
constexpr int RES_MID = 1024;                            // one XCD, four rows a wave: trailing orders 1024 .. 769
constexpr int RES_WG = 32;
constexpr int RESG_MAX = 2048;                           // the whole chip: 256 workgroups, hand-offs through memory
constexpr int RESG_WG = 256;
constexpr int RES_T = 512;
constexpr int RES_XS = 3840 + 64;                        // one exchange vector (the widest kernel: R4C_MAX)
constexpr size_t RES_LDS = 150 * 1024;   // two vectors and scalars; sized so that a CU takes exactly one workgroup
// Waits are bounded in TIME (wall_clock64: the 100 MHz constant clock), not in polls: 10 ms is three orders above
// the longest legitimate wait (a hand-off: microseconds) and short against the blocked repeat it triggers.  The clock
// and the shared failure word are looked at every RES_CHECK polls; one workgroup's time-out therefore releases every
// other workgroup within RES_CHECK polls instead of each of them running into a time-out of its own.
constexpr unsigned long long RES_TIMEOUT_TICKS = 1000000ULL;
constexpr int RES_CHECK = 256;
constexpr int RES_MAX_POLLS = 20000000;   // second guard, in polls (seconds): the wait ends even if the clock did not advance

#ifdef PTD_RES_PROF
__device__ unsigned long long res_prof[5][16];
#define RES_T0(K) unsigned long long prof_t = wall_clock64(); const int prof_k = (K); const bool prof_on = blockIdx.x == ((K) >= 2 ? 16 : 17) && threadIdx.x == 0
#define RES_MARK(I) do { if (prof_on) { const unsigned long long nw = wall_clock64(); res_prof[prof_k][I] += nw - prof_t; prof_t = nw; } } while (0)
extern "C" void ptd_debug_res_prof(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[80] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(res_prof), z, sizeof(z)); }
  else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(res_prof), 80 * sizeof(unsigned long long));
}
#else
#define RES_T0(K)
#define RES_MARK(I)
#endif
struct ResCtl {
  unsigned long long fp[RESG_WG];       // per workgroup: sequence number of its last published entries
  unsigned reg;    unsigned pad1[31];   // registration (agent scope)
  int fail;        int pad2[31];        // 1 time-out, 2 XCC mismatch, 3 the test hook PTD_SYTRD_RESIDENT=2
  int xcc[32];
};

// Exchanged vectors are read with device-scope (sc1) loads: they bypass this CU's L1 and are served by the XCD's
// L2, where the other workgroups' plain stores are (probe: 8 KB per workgroup and barrier for +0.5 us; read with
// returning atomics instead, the 32 workgroups' requests for one line serialise in the L2: +1.9 us, and 500 us when
// the idle lanes shared a dummy address).
__device__ __forceinline__ double res_ld_f64(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Hand-off: a workgroup stores its entries, drains the stores (they are in the L2 then), and publishes a sequence
// number; a consumer wave watches the 32 numbers (lane s: workgroup s) and then reads the vector.  Against a counter
// barrier followed by the read this saves one L2 round trip per exchange, and nothing serialises on one address.
// false after a time-out.
// one look at the clock and at the failure word; false = give up (after marking the time-out)
__device__ __forceinline__ bool res_alive(ResCtl* c, unsigned long long& t_start) {
  if (__hip_atomic_load(&c->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
  const unsigned long long now = wall_clock64();
  if (t_start == 0) t_start = now;
  if (now - t_start > RES_TIMEOUT_TICKS) {
    __hip_atomic_store(&c->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
  }
  return true;
}
template <int NWG>
__device__ __forceinline__ bool res_wait(const unsigned long long* F, unsigned long long seq, int lane, ResCtl* c) {
  unsigned long long t_start = 0;
  for (int spin = 1;; ++spin) {
    bool ok = true;
#pragma unroll
    for (int q = 0; q < (NWG + 63) / 64; ++q)
      ok = ok && __hip_atomic_load(F + ((lane + 64 * q) & (NWG - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= seq;
    if (__all(ok)) return true;
    if (spin % RES_CHECK == 0 && !res_alive(c, t_start)) return false;
    if (spin >= RES_MAX_POLLS) {
      __hip_atomic_store(&c->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}
// GLOBAL: the consumers sit on other XCDs -- entries and number are written through to memory (sc1)
template <bool GLOBAL>
__device__ __forceinline__ void res_st_f64(double* p, double v) {
  if (GLOBAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool GLOBAL>
__device__ __forceinline__ void res_publish(unsigned long long* F, int slot, unsigned long long seq) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's entries have arrived ...
  __syncthreads();
  if (threadIdx.x == 0) {                            // ... before the number
    if (GLOBAL) __hip_atomic_store(F + slot, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(F + slot, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// sum over the 64 lanes of a wave without the LDS crossbar: DPP adds inside each row of 16 lanes, then the four row
// sums by v_readlane in a fixed order (wave-uniform result)
template <int CTRL>
__device__ __forceinline__ double res_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double res_readlane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double res_wave_sum(double v) {
  v += res_dpp<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += res_dpp<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += res_dpp<0x124>(v);   // row_ror 4
  v += res_dpp<0x128>(v);   // row_ror 8
  return (res_readlane(v, 0) + res_readlane(v, 16)) + (res_readlane(v, 32) + res_readlane(v, 48));
}

// The workgroup's rows live in REGISTERS: wave w holds rows q = w, w + 8, .. (global row slot + NWG q), lane l their
// columns l + 64 k -- the layout of the product and of the rank-2 update, which therefore touch no memory at all
// (kept in LDS, streaming the 147 KB slice through the LDS pipe twice a column took 2.9 us of a 7.3 us column).
// LDS holds the two vectors of the column and the scalars.
// (The same kernel on all 256 CUs -- 2048 resident columns, hand-offs written through to memory with sc1 stores -- is
// correct and slower than the blocked path: ~14 us per column, n = 2048 31.3 vs 22.2 ms; a hand-off that leaves the
// XCD costs 5-6 us.)
//   <32, 768, false>    the workgroups of XCC 0 (blockIdx.x % 8 == 0 of a grid of 256), hand-offs through its L2
//   <256, 2048, true>   every CU of the chip, hand-offs written through to memory; stops after `ncols` columns and
//                       leaves the trailing block in Aw for the kernel above
template <int NWG, int MAXM, bool GLOBAL>
__global__ __launch_bounds__(RES_T) void sytrd_resident_kernel(double* __restrict__ Aw, int64_t ld, int n, int t0, int ncols,
                                                               double* __restrict__ Vall, double* __restrict__ taus,
                                                               double* __restrict__ d, double* __restrict__ e,
                                                               ResCtl* __restrict__ ctl, double* __restrict__ Xbuf,
                                                               unsigned long long epoch) {
  constexpr int NW = RES_T / 64;                    // waves
  constexpr int RI = MAXM / NWG / NW;               // rows per wave
  constexpr int CK = MAXM / 64;                     // columns per lane and row
  constexpr int CT = MAXM / RES_T + (MAXM % RES_T != 0);   // vector entries formed per thread
  static_assert(RI >= 1 && RI * NW * NWG == MAXM, "rows must divide evenly");
  extern __shared__ __attribute__((aligned(16))) char res_smem[];
  if (!GLOBAL && (blockIdx.x & 7) != 0) return;
  double* vs = reinterpret_cast<double*>(res_smem);
  double* wv = vs + MAXM;
  double* scr = wv + MAXM;                    // [0, 8) per-wave sums of x^2, [8] alpha, [16, 24) per-wave sums of p v
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int slot = GLOBAL ? blockIdx.x : blockIdx.x >> 3;
  const int m = n - t0;
  if (!GLOBAL) {
    // ---- the 32 workgroups must sit on one XCC
    const int my_xcc = (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID[3:0]
    if (tid == 0) {
      __hip_atomic_store(&ctl->xcc[slot], my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&ctl->reg, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int ok = 1;
      unsigned long long t_start = 0;
      for (int spin = 1;; ++spin) {
        if (__hip_atomic_load(&ctl->reg, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)NWG) break;
        if ((spin % RES_CHECK == 0 && !res_alive(ctl, t_start)) || spin >= RES_MAX_POLLS) {
          __hip_atomic_store(&ctl->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
      if (ok)
        for (int q = 0; q < NWG; ++q)
          if (__hip_atomic_load(&ctl->xcc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != my_xcc) {
            ok = 0;
            __hip_atomic_store(&ctl->fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
      flag = ok;
    }
    __syncthreads();
    if (!flag) return;
  }
  // ---- this thread's entries (zero beyond m: they meet zeros of v but must be finite)
  double a[RI][CK];
  int lr[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    lr[i] = slot + NWG * (wid + NW * i);
    const double* src = Aw + (int64_t)(t0 + min(lr[i], m - 1)) * ld + t0;
#pragma unroll
    for (int k = 0; k < CK; ++k) {
      const int c = lane + 64 * k;
      a[i][k] = (lr[i] < m && c < m) ? src[c] : 0.0;
    }
  }
  // One all-to-all hand-off per column.  With v_j[j+1] = 1 the updated column j+1 is
  //   A'[r][j+1] = A[r][j+1] - v[r] w[j+1] - w[r] = (A[r][j+1] - p[r]) - v[r] (p[j+1] - 2 hk),   hk = tau/2 p^T v,
  // so a workgroup publishes, next to its entries of p = tau A v, the entries b[r] = A[r][j+1] - p[r] of its own rows,
  // and whoever holds all of p, b and v forms p^T v, w AND the next column (its entry r = j+1 is the next diagonal
  // element) without a second exchange.  Sequence numbers are unique over calls (epoch), so a number left in a cache
  // by an earlier call can never pass; the vectors alternate between two buffers (a fast workgroup publishes column
  // j + 1 while a slow one still reads column j).
  double xv[CT];                                            // column jl below its diagonal (entries tid + 512 t), and
  double dnext = 0.0;                                       // the diagonal entry in front of it
  {
    // column 0 is gathered as it lies: b = A[:, 0], p = 0
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < RI; ++i)
        if (lr[i] < m) res_st_f64<GLOBAL>(Xbuf + RES_XS + lr[i], a[i][0]);
    res_publish<GLOBAL>(ctl->fp, slot, epoch);
    if (!res_wait<NWG>(ctl->fp, epoch, lane, ctl)) return;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int c = tid + RES_T * t;
      xv[t] = (c >= 1 && c < m) ? res_ld_f64(Xbuf + RES_XS + c) : 0.0;
    }
    dnext = res_ld_f64(Xbuf + RES_XS);
    // per-wave sums of x^2 (rows 2 ..) and alpha = x[1] of column 0; later columns get theirs where x is formed
    double sq = 0.0;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int c = tid + RES_T * t;
      if (c >= 2) sq += xv[t] * xv[t];
      if (c == 1) scr[8] = xv[t];
    }
    sq = res_wave_sum(sq);
    if (lane == 0) scr[wid] = sq;
    __syncthreads();
  }
  // The rank-2 update of column jl - 1 is applied in the same pass over the registers as the product of column jl
  // (v_{jl} is known before the update: it comes from the hand-off), so the matrix is walked once per column:
  // a <- a - v_r w_c - w_r v_c, then acc += a v'_c.  vreg / vrow_ / wrow_ hold the pending reflector.
  double vreg[CK], vrow_[RI], wrow_[RI];
#pragma unroll
  for (int k = 0; k < CK; ++k) vreg[k] = 0.0;
  for (int c = tid; c < MAXM; c += RES_T) wv[c] = 0.0;     // (the first pass applies an update of zeros)
#pragma unroll
  for (int i = 0; i < RI; ++i) { vrow_[i] = 0.0; wrow_[i] = 0.0; }
  bool pending = false;
  RES_T0(GLOBAL ? 0 : (MAXM == RES_MID ? 2 : 3));
  for (int jl = 0; jl < ncols; ++jl) {
    RES_MARK(7);
    const unsigned long long seq = epoch + (unsigned long long)jl + 1;
    double* Pb = Xbuf + ((jl + 1) & 1) * 2 * RES_XS;        // this column's p entries, then its b entries
    double* Bb = Pb + RES_XS;
    const int k1 = (jl + 1) >> 6, l1 = (jl + 1) & 63;       // where column jl + 1 sits in the registers
    // ---- the reflector of column jl: the same arithmetic in every workgroup
    double vv[CT];
    double xn2 = 0.0;                                       // (scr[0 .. 8] were written behind the last barrier)
#pragma unroll
    for (int w = 0; w < NW; ++w) xn2 += scr[w];
    const double alpha = scr[8];
    double tau, beta, scale;
    if (xn2 == 0.0) {
      tau = 0.0; beta = alpha; scale = 0.0;
    } else {
      const double nrm = sqrt(alpha * alpha + xn2);
      beta = alpha >= 0.0 ? -nrm : nrm;
      tau = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    {
      double* vrow = Vall + (int64_t)(t0 + jl) * ld + t0;
      const bool writer = slot == (jl & (NWG - 1));
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int c = tid + RES_T * t;
        vv[t] = (c == jl + 1) ? 1.0 : ((c > jl + 1) ? xv[t] * scale : 0.0);
        if (c < MAXM) vs[c] = vv[t];
        if (writer && c > jl && c < m) vrow[c] = vv[t];
      }
      if (writer && tid == 0) { taus[t0 + jl] = tau; e[t0 + jl] = beta; d[t0 + jl] = dnext; }
    }
    __syncthreads();
    RES_MARK(0);
    // ---- the pending update, then p = tau A v and b = A[:, jl + 1] - p for this workgroup's rows
    {
      double acc[RI];
#pragma unroll
      for (int i = 0; i < RI; ++i) acc[i] = 0.0;
      // Straight arithmetic on every live register column: where a row or a column has retired, or no update is
      // pending, v and w of the update are zero (v_j and w_j vanish up to index j; tau = 0 gives w = 0) and the new v is
      // zero over retired columns, so neither a predicate per row nor a test per column is needed (they were 4 x the
      // arithmetic: 2.6 us of a 5.5-us column in <32, 1024>, tools/probes/res_prof.py).
      const int kold = jl >> 6;                             // first live register column of the pending update
#pragma unroll
      for (int k = 0; k < CK; ++k) {
        if (k < kold) continue;                             // (wave-uniform; also bounds how far the LDS reads are hoisted)
        const double vnew = vs[lane + 64 * k];
        const double wk = wv[lane + 64 * k];
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          a[i][k] -= vrow_[i] * wk + wrow_[i] * vreg[k];
          acc[i] += a[i][k] * vnew;
        }
        vreg[k] = vnew;
      }
#pragma unroll
      for (int i = 0; i < RI; ++i) vrow_[i] = vs[min(lr[i], MAXM - 1)];
      // (the rows' entries of column jl + 1: one wave-uniform test per register column instead of a select per row and
      // column)
      double aj1[RI];
#pragma unroll
      for (int i = 0; i < RI; ++i) aj1[i] = 0.0;
#pragma unroll
      for (int k = 0; k < CK; ++k)
        if (k == k1)
#pragma unroll
          for (int i = 0; i < RI; ++i) aj1[i] = a[i][k];
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const double pi = tau * res_wave_sum(acc[i]);
        if (lane == l1 && lr[i] > jl && lr[i] < m) {
          res_st_f64<GLOBAL>(Pb + lr[i], pi);
          res_st_f64<GLOBAL>(Bb + lr[i], aj1[i] - pi);
        }
      }
    }
    RES_MARK(1);
    res_publish<GLOBAL>(ctl->fp, slot, seq);
    RES_MARK(2);
    if (GLOBAL) {
      // one wave watches the 256 numbers (eight waves of 256 workgroups polling four lines were a storm of their own)
      if (wid == 0) { const bool ok = res_wait<NWG>(ctl->fp, seq, lane, ctl); if (lane == 0) flag = ok; }
      __syncthreads();
      if (!flag) return;
    } else if (!res_wait<NWG>(ctl->fp, seq, lane, ctl)) return;
    RES_MARK(3);
    double pv_[CT], bv_[CT];
    {
      double dp = 0.0;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int c = tid + RES_T * t;
        const bool ok = c > jl && c < m;
        pv_[t] = ok ? res_ld_f64(Pb + c) : 0.0;
        bv_[t] = ok ? res_ld_f64(Bb + c) : 0.0;
        dp += pv_[t] * vv[t];
        if (c == jl + 1) scr[9] = pv_[t];
      }
      dp = res_wave_sum(dp);                                // p^T v from the whole vectors
      if (lane == 0) scr[16 + wid] = dp;
    }
    __syncthreads();
    RES_MARK(4);
    double dot = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) dot += scr[16 + w];
    const double hk = 0.5 * tau * dot;
    const double g = scr[9] - 2.0 * hk;                     // p[jl + 1] - 2 hk
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int c = tid + RES_T * t;
      if (c < MAXM) wv[c] = pv_[t] - hk * vv[t];            // zero outside (jl, m): p and v are
      const double xn = bv_[t] - vv[t] * g;                 // the updated column jl + 1, rows jl + 1 ..
      if (c == jl + 1) scr[10] = xn;                        // .. whose first entry is the next diagonal element
      xv[t] = (c >= jl + 2) ? xn : 0.0;
    }
    {
      // the next reflector's sums travel through the same barrier: alpha = x[jl + 2], x^2 over the rows below
      double sq = 0.0;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int c = tid + RES_T * t;
        if (c >= jl + 3) sq += xv[t] * xv[t];
        if (c == jl + 2) scr[8] = xv[t];
      }
      sq = res_wave_sum(sq);
      if (lane == 0) scr[wid] = sq;
    }
    __syncthreads();
    RES_MARK(5);
    dnext = scr[10];
    // ---- the update A -= v w^T + w v^T waits for the next column's pass (or the one behind the loop)
#pragma unroll
    for (int i = 0; i < RI; ++i) wrow_[i] = wv[min(lr[i], MAXM - 1)];
    pending = tau != 0.0;
    // (vs / wv / scr are rewritten only behind the barriers of the next column)
  }
  if (ncols == m - 1) {
    if (slot == 0 && tid == 0) d[t0 + m - 1] = dnext;
  } else {
    // the trailing block goes back to memory for the next kernel (both triangles, as it lies in the registers),
    // with the last column's update applied on the way
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int k = 0; k < CK; ++k) {
        const int c = lane + 64 * k;
        if (pending) a[i][k] -= vrow_[i] * wv[c] + wrow_[i] * vreg[k];
        if (lr[i] >= ncols && lr[i] < m && c >= ncols && c < m) Aw[(int64_t)(t0 + lr[i]) * ld + t0 + c] = a[i][k];
      }
  }
}

// The whole-chip kernel for trailing orders up to 3072 (12 rows a workgroup).  A row does not fit one wave's
// registers next to everything else, so a wave holds three HALF rows -- half-row h = wave + 8 i is row h / 2 of the
// workgroup (global row slot + 256 (h / 2)), columns (h & 1) 1536 + lane + 64 k: 72 doubles a thread -- and the
// product of a row is the sum of two waves' partial sums (LDS).  v, w and the next column live in LDS (3 x 24 KB);
// nothing of the matrix does.  As the columns retire left to right the waves of the left halves fall idle after
// 1536 columns; the hand-offs, not the arithmetic, set the pace.  Run from a trailing order of at most 3072 down to
// 2048, where sytrd_resident_kernel<256, 2048, true> takes over (one row per wave, operands in registers).
constexpr int R3_MAX = 3072;
constexpr int R3_HALF = R3_MAX / 2;
constexpr int R3_CK = R3_HALF / 64;       // columns per lane and half row
constexpr int R3_HR = 3;                  // half rows per wave (24 half rows over 8 waves)
constexpr int R3_CT = R3_MAX / RES_T;     // vector entries formed per thread

__global__ __launch_bounds__(RES_T) void sytrd_resident3_kernel(double* __restrict__ Aw, int64_t ld, int n, int t0,
                                                                int ncols, double* __restrict__ Vall,
                                                                double* __restrict__ taus, double* __restrict__ d,
                                                                double* __restrict__ e, ResCtl* __restrict__ ctl,
                                                                double* __restrict__ Xbuf, unsigned long long epoch) {
  constexpr int NWG = RESG_WG, NW = RES_T / 64;
  extern __shared__ __attribute__((aligned(16))) char res_smem[];
  double* vsb = reinterpret_cast<double*>(res_smem);   // two v vectors: column jl's in half jl & 1, the pending update's in the other
  double* wv = vsb + 2 * R3_MAX;
  double* xs = wv + R3_MAX;                   // the current column below its diagonal
  double* part = xs + R3_MAX;                 // [24] partial row products
  double* scr = part + 32;                    // [0, 8) per-wave sums of x^2, [8] alpha, [9] p[jl+1], [10] next diagonal, [16, 24) p^T v
  double* aj = scr + 32;                      // [12] A[row q][jl + 1] (after the pending update)
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int slot = blockIdx.x;
  const int m = n - t0;
  const int half = wid & 1, cbase = half * R3_HALF + lane;
  double a[R3_HR][R3_CK];
  int lr[R3_HR];
#pragma unroll
  for (int i = 0; i < R3_HR; ++i) {
    lr[i] = slot + NWG * ((wid + NW * i) >> 1);
    const double* src = Aw + (int64_t)(t0 + min(lr[i], m - 1)) * ld + t0;
#pragma unroll
    for (int k = 0; k < R3_CK; ++k) {
      const int c = cbase + 64 * k;
      a[i][k] = (lr[i] < m && c < m) ? src[c] : 0.0;
    }
  }
  auto wait_all = [&](unsigned long long seq) -> bool {
    if (wid == 0) { const bool ok = res_wait<NWG>(ctl->fp, seq, lane, ctl); if (lane == 0) flag = ok; }
    __syncthreads();
    return flag != 0;
  };
  double dnext;
  {
    // column 0 is gathered as it lies: b = A[:, 0], p = 0
    if (half == 0 && lane == 0)
#pragma unroll
      for (int i = 0; i < R3_HR; ++i)
        if (lr[i] < m) res_st_f64<true>(Xbuf + RES_XS + lr[i], a[i][0]);
    res_publish<true>(ctl->fp, slot, epoch);
    if (!wait_all(epoch)) return;
    // its x, the sums of its reflector (alpha = x[1], x^2 over the rows below), and zeros where the first pass looks
    // for a pending update
    double sq = 0.0;
#pragma unroll
    for (int t = 0; t < R3_CT; ++t) {
      const int c = tid + RES_T * t;
      const double x = (c >= 1 && c < m) ? res_ld_f64(Xbuf + RES_XS + c) : 0.0;
      xs[c] = x;
      if (c >= 2) sq += x * x;
      if (c == 1) scr[8] = x;
      vsb[R3_MAX + c] = 0.0;
      wv[c] = 0.0;
    }
    dnext = res_ld_f64(Xbuf + RES_XS);
    sq = res_wave_sum(sq);
    if (lane == 0) scr[wid] = sq;
    __syncthreads();
  }
  // The rank-2 update of column jl - 1 rides in the product pass of column jl (see sytrd_resident_kernel); the old v of
  // the lane's columns cannot stay in registers here: the two v vectors alternate between two LDS buffers.  The pass is
  // straight-line code over all 24 register columns and three half rows: where a row or a column has retired, or no
  // update is pending, v and w of the update are ZERO there (v_j and w_j vanish up to index j, and tau = 0 gives w = 0),
  // and the new v is zero over the retired columns -- the arithmetic is a no-op without a predicate.  (With a branch
  // per register column and a predicate per row the pass took 5.9 us of a 13.8-us column: 517 branches and 27 spilled
  // registers in the loop, tools/probes/res_prof.py.)
  double vrow_[R3_HR], wrow_[R3_HR];
#pragma unroll
  for (int i = 0; i < R3_HR; ++i) { vrow_[i] = 0.0; wrow_[i] = 0.0; }
  RES_T0(1);
  for (int jl = 0; jl < ncols; ++jl) {
    RES_MARK(7);
    const unsigned long long seq = epoch + (unsigned long long)jl + 1;
    double* Pb = Xbuf + ((jl + 1) & 1) * 2 * RES_XS;
    double* Bb = Pb + RES_XS;
    double* vs = vsb + (jl & 1) * R3_MAX;
    const double* vo = vsb + ((jl & 1) ^ 1) * R3_MAX;
    // ---- the reflector of column jl: the same arithmetic in every workgroup (its sums came through the last barrier)
    double xn2 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) xn2 += scr[w];
    const double alpha = scr[8];
    double tau, beta, scale;
    if (xn2 == 0.0) {
      tau = 0.0; beta = alpha; scale = 0.0;
    } else {
      const double nrm = sqrt(alpha * alpha + xn2);
      beta = alpha >= 0.0 ? -nrm : nrm;
      tau = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    {
      double* vrow = Vall + (int64_t)(t0 + jl) * ld + t0;
      const bool writer = slot == (jl & (NWG - 1));
      // (two consecutive entries a thread: 16-byte LDS accesses)
#pragma unroll
      for (int t = 0; t < (R3_MAX / 2 + RES_T - 1) / RES_T; ++t) {
        const int c = 2 * (tid + RES_T * t);
        if (c >= R3_MAX) break;
        const double2 x2 = *reinterpret_cast<const double2*>(xs + c);
        double2 v2;
        v2.x = (c == jl + 1) ? 1.0 : ((c > jl + 1) ? x2.x * scale : 0.0);
        v2.y = (c + 1 == jl + 1) ? 1.0 : ((c + 1 > jl + 1) ? x2.y * scale : 0.0);
        *reinterpret_cast<double2*>(vs + c) = v2;
        if (writer) {
          if (c > jl && c + 1 < m) *reinterpret_cast<double2*>(vrow + c) = v2;
          else {
            if (c > jl && c < m) vrow[c] = v2.x;
            if (c + 1 > jl && c + 1 < m) vrow[c + 1] = v2.y;
          }
        }
      }
      if (writer && tid == 0) { taus[t0 + jl] = tau; e[t0 + jl] = beta; d[t0 + jl] = dnext; }
    }
    __syncthreads();
    RES_MARK(0);
    // ---- the pending update, then the partial products of the half rows
    {
      double acc[R3_HR];
#pragma unroll
      for (int i = 0; i < R3_HR; ++i) acc[i] = 0.0;
      const int kold = (jl - half * R3_HALF) >> 6;     // register columns below hold only retired columns (negative: none)
      const int half1 = (jl + 1) >= R3_HALF ? 1 : 0;
      const int l1 = (jl + 1 - half1 * R3_HALF) & 63, k1 = (jl + 1 - half1 * R3_HALF) >> 6;
#pragma unroll
      for (int k = 0; k < R3_CK; ++k) {
        if (k < kold) continue;          // (wave-uniform; the branch also keeps the compiler from hoisting all 72 LDS reads)
        const int c = cbase + 64 * k;
        const double vnew = vs[c], wk = wv[c], vold = vo[c];
#pragma unroll
        for (int i = 0; i < R3_HR; ++i) {
          a[i][k] -= vrow_[i] * wk + wrow_[i] * vold;
          acc[i] += a[i][k] * vnew;
        }
      }
      if (half == half1) {                   // (wave-uniform) the updated entries of column jl + 1, for the b values
#pragma unroll
        for (int k = 0; k < R3_CK; ++k)
          if (k == k1 && lane == l1)
#pragma unroll
            for (int i = 0; i < R3_HR; ++i) aj[(wid + NW * i) >> 1] = a[i][k];
      }
#pragma unroll
      for (int i = 0; i < R3_HR; ++i) {
        const double sacc = res_wave_sum(acc[i]);
        if (lane == 0) part[wid + NW * i] = sacc;
        vrow_[i] = vs[lr[i]];
      }
    }
    __syncthreads();
    RES_MARK(1);
    // ---- p = tau A v and b = A[:, jl + 1] - p, stored by the lane that holds column jl + 1 of the row
    if (tid < 12) {                        // row q = tid of the workgroup: its two half products, its entry of column jl + 1
      const int row = slot + NWG * tid;
      if (row > jl && row < m) {
        const double pi = tau * (part[2 * tid] + part[2 * tid + 1]);
        res_st_f64<true>(Pb + row, pi);
        res_st_f64<true>(Bb + row, aj[tid] - pi);
      }
    }
    RES_MARK(6);
    res_publish<true>(ctl->fp, slot, seq);
    RES_MARK(2);
    if (!wait_all(seq)) return;
    RES_MARK(3);
    // (p and b from memory into LDS -- where w and the next column are formed from them -- by LDS-DMA at device scope:
    // see sytrd_resident4_kernel)
    {
      constexpr int NCH = R3_MAX * 8 / 1024;
      const unsigned lds_w = (unsigned)(size_t)(lds_void_t*)wv, lds_x = (unsigned)(size_t)(lds_void_t*)xs;
      const unsigned voff = (unsigned)lane * 16u;
      const int swid = __builtin_amdgcn_readfirstlane(wid);
      for (int i = swid; i < NCH; i += NW) {
        const char* gp = reinterpret_cast<const char*>(Pb) + i * 1024;
        const char* gb = reinterpret_cast<const char*>(Bb) + i * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1"
                     : : "s"(lds_w + i * 1024), "v"(voff), "s"(gp) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1"
                     : : "s"(lds_x + i * 1024), "v"(voff), "s"(gb) : "memory", "m0");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    {
      double dp = 0.0;
#pragma unroll
      for (int t = 0; t < (R3_MAX / 2 + RES_T - 1) / RES_T; ++t) {
        const int c = 2 * (tid + RES_T * t);
        if (c >= R3_MAX) break;
        const double2 p2 = *reinterpret_cast<const double2*>(wv + c), v2 = *reinterpret_cast<const double2*>(vs + c);
        const double p0 = (c > jl && c < m) ? p2.x : 0.0;      // (entries of retired rows are two columns old)
        const double p1 = (c + 1 > jl && c + 1 < m) ? p2.y : 0.0;
        dp += p0 * v2.x + p1 * v2.y;
        if (c == jl + 1) scr[9] = p0;
        if (c + 1 == jl + 1) scr[9] = p1;
      }
      dp = res_wave_sum(dp);
      if (lane == 0) scr[16 + wid] = dp;
    }
    __syncthreads();
    RES_MARK(4);
    double dot = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) dot += scr[16 + w];
    const double hk = 0.5 * tau * dot;
    const double g = scr[9] - 2.0 * hk;
    {
      // w, the next column and -- through the same barrier -- the sums of ITS reflector: alpha = x[jl + 2], x^2 below
      double sq = 0.0;
#pragma unroll
      for (int t = 0; t < (R3_MAX / 2 + RES_T - 1) / RES_T; ++t) {
        const int c = 2 * (tid + RES_T * t);
        if (c >= R3_MAX) break;
        const double2 v2 = *reinterpret_cast<const double2*>(vs + c), p2 = *reinterpret_cast<const double2*>(wv + c),
                      b2 = *reinterpret_cast<const double2*>(xs + c);
        const bool ok0 = c > jl && c < m, ok1 = c + 1 > jl && c + 1 < m;
        const double p0 = ok0 ? p2.x : 0.0, p1 = ok1 ? p2.y : 0.0, b0 = ok0 ? b2.x : 0.0, b1 = ok1 ? b2.y : 0.0;
        *reinterpret_cast<double2*>(wv + c) = double2{p0 - hk * v2.x, p1 - hk * v2.y};
        const double x0 = b0 - v2.x * g, x1 = b1 - v2.y * g;
        if (c == jl + 1) scr[10] = x0;
        if (c + 1 == jl + 1) scr[10] = x1;
        if (c == jl + 2) scr[8] = x0;
        if (c + 1 == jl + 2) scr[8] = x1;
        *reinterpret_cast<double2*>(xs + c) = double2{(c >= jl + 2) ? x0 : 0.0, (c + 1 >= jl + 2) ? x1 : 0.0};
        if (c >= jl + 3) sq += x0 * x0;
        if (c + 1 >= jl + 3) sq += x1 * x1;
      }
      sq = res_wave_sum(sq);
      if (lane == 0) scr[wid] = sq;
    }
    __syncthreads();
    RES_MARK(5);
    dnext = scr[10];
    // ---- the update waits for the next column's pass (vs of this column is its `vo`)
#pragma unroll
    for (int i = 0; i < R3_HR; ++i) wrow_[i] = wv[lr[i]];
  }
  if (ncols == m - 1) {
    if (slot == 0 && tid == 0) d[t0 + m - 1] = dnext;
  } else {
    const double* vo = vsb + ((ncols - 1) & 1) * R3_MAX;
#pragma unroll
    for (int i = 0; i < R3_HR; ++i)
#pragma unroll
      for (int k = 0; k < R3_CK; ++k) {
        const int c = cbase + 64 * k;
        if (ncols > 0) a[i][k] -= vrow_[i] * wv[c] + wrow_[i] * vo[c];
        if (lr[i] >= ncols && lr[i] < m && c >= ncols && c < m) Aw[(int64_t)(t0 + lr[i]) * ld + t0 + c] = a[i][k];
      }
  }
}

// The whole-chip kernel in FRONT of the half-row one: trailing orders 3328 .. 3073, 13 rows a workgroup on FOUR waves
// (one per SIMD: 512 registers a lane).  Wave w holds column quarter w (832 columns = 13 register columns) of all 13
// rows of the workgroup -- 169 doubles a lane, of which the compiler keeps what does not fit the 256 architectural
// registers in the accumulation file -- so a lane reads v, w and the old v of its 13 columns once for 13 rows (39 LDS
// reads a column where the half-row kernel has 72), and a row's product is the sum over the 256 lanes that hold a
// piece of it: every lane leaves its 13 partial sums in LDS and sixteen threads per row add them up.  A column moved
// from the blocked path (two launches, ~21 us + the panel updates) into this kernel costs what a hand-off costs plus
// its pass; the kernel stops at 3072, where the half-row kernel's pass is cheaper (72 doubles in plain registers).
// MAXM = 3584 (14 rows, 196 doubles a lane) in front of MAXM = 3328: its rows are passed in four groups.
constexpr int R4_MAX = 3328, R4B_MAX = 3584, R4C_MAX = 3840;
constexpr int R4_T = 256;
static_assert(R4C_MAX + 64 <= RES_XS, "exchange vectors");

template <int MAXM>
__global__ __launch_bounds__(R4_T) void sytrd_resident4_kernel(double* __restrict__ Aw, int64_t ld, int n, int t0,
                                                               int ncols, double* __restrict__ Vall,
                                                               double* __restrict__ taus, double* __restrict__ d,
                                                               double* __restrict__ e, ResCtl* __restrict__ ctl,
                                                               double* __restrict__ Xbuf, unsigned long long epoch) {
  constexpr int NWG = RESG_WG, NW = R4_T / 64;
  constexpr int R4_ROWS = MAXM / RESG_WG;   // rows of a workgroup
  constexpr int R4_Q = MAXM / 4;            // columns of a wave's quarter
  constexpr int R4_CK = R4_Q / 64;          // register columns per lane and row
  constexpr int R4_CT = MAXM / R4_T;        // vector entries formed per thread
  static_assert(R4_ROWS * RESG_WG == MAXM && R4_CK * 256 == MAXM && R4_ROWS <= 16, "13 or 14 rows");
  static_assert((4 * MAXM + 64) * 8 <= (int)RES_LDS && R4_ROWS * R4_T <= MAXM, "LDS");
  extern __shared__ __attribute__((aligned(16))) char res_smem[];
  double* vsb = reinterpret_cast<double*>(res_smem);   // two v vectors: column jl's in half jl & 1, the pending update's in the other
  double* wv = vsb + 2 * MAXM;
  double* xs = wv + MAXM;                   // the current column below its diagonal
  double* red = xs;                           // [rows][256] every lane's partial row products: between the reflector's read of
                                              // the column and the gather of b the space of xs is free
  double* aj = xs + MAXM;                     // [16] A[row q][jl + 1] (after the pending update)
  double* scr = aj + 16;                      // [0, 4) per-wave sums of x^2, [8] alpha, [9] p[jl+1], [10] next diagonal, [16, 20) p^T v
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int slot = blockIdx.x;
  const int m = n - t0;
  const int cbase = wid * R4_Q + lane;
  double a[R4_ROWS][R4_CK];
#pragma unroll
  for (int q = 0; q < R4_ROWS; ++q) {
    const int row = slot + NWG * q;
    const double* src = Aw + (int64_t)(t0 + min(row, m - 1)) * ld + t0;
#pragma unroll
    for (int k = 0; k < R4_CK; ++k) {
      const int c = cbase + 64 * k;
      a[q][k] = (row < m && c < m) ? src[c] : 0.0;
    }
  }
  auto wait_all = [&](unsigned long long seq) -> bool {
    if (wid == 0) { const bool ok = res_wait<NWG>(ctl->fp, seq, lane, ctl); if (lane == 0) flag = ok; }
    __syncthreads();
    return flag != 0;
  };
  double dnext;
  {
    // column 0 is gathered as it lies: b = A[:, 0], p = 0
    if (wid == 0 && lane == 0)
#pragma unroll
      for (int q = 0; q < R4_ROWS; ++q)
        if (slot + NWG * q < m) res_st_f64<true>(Xbuf + RES_XS + slot + NWG * q, a[q][0]);
    res_publish<true>(ctl->fp, slot, epoch);
    if (!wait_all(epoch)) return;
    double sq = 0.0;
#pragma unroll
    for (int t = 0; t < R4_CT; ++t) {
      const int c = tid + R4_T * t;
      const double x = (c >= 1 && c < m) ? res_ld_f64(Xbuf + RES_XS + c) : 0.0;
      xs[c] = x;
      if (c >= 2) sq += x * x;
      if (c == 1) scr[8] = x;
      vsb[MAXM + c] = 0.0;
      wv[c] = 0.0;
    }
    dnext = res_ld_f64(Xbuf + RES_XS);
    sq = res_wave_sum(sq);
    if (lane == 0) scr[wid] = sq;
    __syncthreads();
  }
  RES_T0(4);
  for (int jl = 0; jl < ncols; ++jl) {
    RES_MARK(7);
    const unsigned long long seq = epoch + (unsigned long long)jl + 1;
    double* Pb = Xbuf + ((jl + 1) & 1) * 2 * RES_XS;
    double* Bb = Pb + RES_XS;
    double* vs = vsb + (jl & 1) * MAXM;
    const double* vo = vsb + ((jl & 1) ^ 1) * MAXM;
    // ---- the reflector of column jl: the same arithmetic in every workgroup (its sums came through the last barrier)
    double xn2 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) xn2 += scr[w];
    const double alpha = scr[8];
    double tau, beta, scale;
    if (xn2 == 0.0) {
      tau = 0.0; beta = alpha; scale = 0.0;
    } else {
      const double nrm = sqrt(alpha * alpha + xn2);
      beta = alpha >= 0.0 ? -nrm : nrm;
      tau = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    {
      double* vrow = Vall + (int64_t)(t0 + jl) * ld + t0;
      const bool writer = slot == (jl & (NWG - 1));
      // (two consecutive entries a thread: 16-byte LDS accesses)
#pragma unroll
      for (int t = 0; t < (MAXM / 2 + R4_T - 1) / R4_T; ++t) {
        const int c = 2 * (tid + R4_T * t);
        if (c >= MAXM) break;
        const double2 x2 = *reinterpret_cast<const double2*>(xs + c);
        double2 v2;
        v2.x = (c == jl + 1) ? 1.0 : ((c > jl + 1) ? x2.x * scale : 0.0);
        v2.y = (c + 1 == jl + 1) ? 1.0 : ((c + 1 > jl + 1) ? x2.y * scale : 0.0);
        *reinterpret_cast<double2*>(vs + c) = v2;
        if (writer) {
          if (c > jl && c + 1 < m) *reinterpret_cast<double2*>(vrow + c) = v2;
          else {
            if (c > jl && c < m) vrow[c] = v2.x;
            if (c + 1 > jl && c + 1 < m) vrow[c + 1] = v2.y;
          }
        }
      }
      if (writer && tid == 0) { taus[t0 + jl] = tau; e[t0 + jl] = beta; d[t0 + jl] = dnext; }
    }
    __syncthreads();
    RES_MARK(0);
    // ---- the pending update, then every lane's partial products of the 13 rows (straight arithmetic: see the half-row
    // kernel); the lane that holds column jl + 1 leaves the updated entries of that column for the b values
    // (the rows in two groups: a group's values of the pending v and w -- read from LDS, where they still lie -- and its
    // accumulators are all that lives beside the matrix; v, w and the old v of the lane's columns are read once per
    // group)
    {
      const int kold = (jl - wid * R4_Q) >> 6;         // register columns below hold only retired columns (negative: none)
      const int w1 = (jl + 1) / R4_Q;
      const int l1 = (jl + 1 - w1 * R4_Q) & 63, k1 = (jl + 1 - w1 * R4_Q) >> 6;
      constexpr int GR = R4_ROWS >= 15 ? 3 : (R4_ROWS >= 14 ? (R4_ROWS + 2) / 3 : R4_ROWS);     // (14 rows: three groups -- what fits the registers without scratch)
#pragma unroll
      for (int g0 = 0; g0 < R4_ROWS; g0 += GR) {
        double vr[GR], wr[GR], acc[GR];
#pragma unroll
        for (int q = 0; q < GR; ++q) {
          const int row = slot + NWG * min(g0 + q, R4_ROWS - 1);
          vr[q] = vo[row]; wr[q] = wv[row]; acc[q] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < R4_CK; ++k) {
          if (k < kold) continue;          // (wave-uniform)
          const int c = cbase + 64 * k;
          const double vnew = vs[c], wk = wv[c], vold = vo[c];
#pragma unroll
          for (int q = 0; q < GR; ++q)
            if (g0 + q < R4_ROWS) {
              a[g0 + q][k] -= vr[q] * wk + wr[q] * vold;
              acc[q] += a[g0 + q][k] * vnew;
            }
        }
        if (wid == w1) {                     // (wave-uniform) the updated entries of column jl + 1, for the b values
#pragma unroll
          for (int k = 0; k < R4_CK; ++k)
            if (k == k1 && lane == l1)
#pragma unroll
              for (int q = 0; q < GR; ++q)
                if (g0 + q < R4_ROWS) aj[g0 + q] = a[g0 + q][k];
        }
#pragma unroll
        for (int q = 0; q < GR; ++q)
          if (g0 + q < R4_ROWS) red[(g0 + q) * R4_T + tid] = acc[q];
      }
    }
    __syncthreads();
    RES_MARK(1);
    // ---- p = tau A v and b = A[:, jl + 1] - p: sixteen threads add up a row's 256 partial sums
    if (tid < 16 * R4_ROWS) {
      const int q = tid >> 4, sub = tid & 15;
      double sacc = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc += red[q * R4_T + sub + 16 * i];
      sacc += res_dpp<0xB1>(sacc);
      sacc += res_dpp<0x4E>(sacc);
      sacc += res_dpp<0x124>(sacc);
      sacc += res_dpp<0x128>(sacc);
      const int row = slot + NWG * q;
      if (sub == 0 && row > jl && row < m) {
        const double pi = tau * sacc;
        res_st_f64<true>(Pb + row, pi);
        res_st_f64<true>(Bb + row, aj[q] - pi);
      }
    }
    RES_MARK(6);
    res_publish<true>(ctl->fp, slot, seq);
    RES_MARK(2);
    if (!wait_all(seq)) return;
    RES_MARK(3);
    // ---- the gathered p and b go from memory into LDS -- where w and the next column are formed from them -- by LDS-DMA
    // at device scope: no registers (28 of them would not fit beside 14 rows), and one round trip for both vectors
    {
      constexpr int NCH = MAXM * 8 / 1024;      // 1-KiB pieces of a vector (64 lanes x 16 bytes an instruction)
      static_assert(NCH * 1024 == MAXM * 8, "whole pieces");
      const unsigned lds_w = (unsigned)(size_t)(lds_void_t*)wv, lds_x = (unsigned)(size_t)(lds_void_t*)xs;
      const unsigned voff = (unsigned)lane * 16u;
      const int swid = __builtin_amdgcn_readfirstlane(wid);
      for (int i = swid; i < NCH; i += NW) {
        const char* gp = reinterpret_cast<const char*>(Pb) + i * 1024;
        const char* gb = reinterpret_cast<const char*>(Bb) + i * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1"
                     : : "s"(lds_w + i * 1024), "v"(voff), "s"(gp) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1"
                     : : "s"(lds_x + i * 1024), "v"(voff), "s"(gb) : "memory", "m0");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    {
      double dp = 0.0;
#pragma unroll
      for (int t = 0; t < (MAXM / 2 + R4_T - 1) / R4_T; ++t) {
        const int c = 2 * (tid + R4_T * t);
        if (c >= MAXM) break;
        const double2 p2 = *reinterpret_cast<const double2*>(wv + c), v2 = *reinterpret_cast<const double2*>(vs + c);
        const double p0 = (c > jl && c < m) ? p2.x : 0.0;      // (entries of retired rows are two columns old)
        const double p1 = (c + 1 > jl && c + 1 < m) ? p2.y : 0.0;
        dp += p0 * v2.x + p1 * v2.y;
        if (c == jl + 1) scr[9] = p0;
        if (c + 1 == jl + 1) scr[9] = p1;
      }
      dp = res_wave_sum(dp);
      if (lane == 0) scr[16 + wid] = dp;
    }
    __syncthreads();
    RES_MARK(4);
    double dot = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) dot += scr[16 + w];
    const double hk = 0.5 * tau * dot;
    const double g = scr[9] - 2.0 * hk;
    {
      double sq = 0.0;
#pragma unroll
      for (int t = 0; t < (MAXM / 2 + R4_T - 1) / R4_T; ++t) {
        const int c = 2 * (tid + R4_T * t);
        if (c >= MAXM) break;
        const double2 v2 = *reinterpret_cast<const double2*>(vs + c), p2 = *reinterpret_cast<const double2*>(wv + c),
                      b2 = *reinterpret_cast<const double2*>(xs + c);
        const bool ok0 = c > jl && c < m, ok1 = c + 1 > jl && c + 1 < m;
        const double p0 = ok0 ? p2.x : 0.0, p1 = ok1 ? p2.y : 0.0, b0 = ok0 ? b2.x : 0.0, b1 = ok1 ? b2.y : 0.0;
        *reinterpret_cast<double2*>(wv + c) = double2{p0 - hk * v2.x, p1 - hk * v2.y};
        const double x0 = b0 - v2.x * g, x1 = b1 - v2.y * g;
        if (c == jl + 1) scr[10] = x0;
        if (c + 1 == jl + 1) scr[10] = x1;
        if (c == jl + 2) scr[8] = x0;
        if (c + 1 == jl + 2) scr[8] = x1;
        *reinterpret_cast<double2*>(xs + c) = double2{(c >= jl + 2) ? x0 : 0.0, (c + 1 >= jl + 2) ? x1 : 0.0};
        if (c >= jl + 3) sq += x0 * x0;
        if (c + 1 >= jl + 3) sq += x1 * x1;
      }
      sq = res_wave_sum(sq);
      if (lane == 0) scr[wid] = sq;
    }
    __syncthreads();
    RES_MARK(5);
    dnext = scr[10];
  }
  if (ncols == m - 1) {
    if (slot == 0 && tid == 0) d[t0 + m - 1] = dnext;
  } else {
    const double* vo = vsb + ((ncols - 1) & 1) * MAXM;
#pragma unroll
    for (int q = 0; q < R4_ROWS; ++q)
#pragma unroll
      for (int k = 0; k < R4_CK; ++k) {
        const int c = cbase + 64 * k, row = slot + NWG * q;
        if (ncols > 0) a[q][k] -= vo[row] * wv[c] + wv[row] * vo[c];
        if (row >= ncols && row < m && c >= ncols && c < m) Aw[(int64_t)(t0 + row) * ld + t0 + c] = a[q][k];
      }
  }
}

// very last diagonal entry: d[n-1] = base[n-1] - delta (delta = 0 if the column opens a panel)
__global__ void sytrd_last_kernel(int n, int i, const double* __restrict__ colbuf,
                                  const double* __restrict__ partial2, int nparts2,
                                  const double* __restrict__ taus, double* __restrict__ d, size_t bstride) {
  const size_t boff_ = (size_t)blockIdx.y * bstride;
  PTD_BATCH(colbuf); PTD_BATCH(partial2); PTD_BATCH(taus); PTD_BATCH(d);
  double dl = 0.0;
  if (i > 0) dl = -taus[n - 2] * wave_total(partial2, nparts2, threadIdx.x);
  if (threadIdx.x == 0) d[n - 1] = colbuf[n - 1] - dl;
}

// ---- Sturm-count bisection: thread k finds the k-th smallest eigenvalue of T(d, e)
__global__ void tridiag_bounds_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                      double* __restrict__ out /* gl, gu, pivmin, tnorm */) {
  __shared__ double rl[16], ru[16], re[16];
  const int tid = threadIdx.x;
  double gl = INFINITY, gu = -INFINITY, emax = 0.0;
  for (int k = tid; k < n; k += blockDim.x) {
    const double el = k > 0 ? fabs(e[k - 1]) : 0.0, er = k < n - 1 ? fabs(e[k]) : 0.0;
    gl = fmin(gl, d[k] - el - er);
    gu = fmax(gu, d[k] + el + er);
    emax = fmax(emax, er);
  }
  for (int o = 32; o > 0; o >>= 1) {
    gl = fmin(gl, __shfl_xor(gl, o));
    gu = fmax(gu, __shfl_xor(gu, o));
    emax = fmax(emax, __shfl_xor(emax, o));
  }
  if ((tid & 63) == 0) { rl[tid >> 6] = gl; ru[tid >> 6] = gu; re[tid >> 6] = emax; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
      gl = fmin(gl, rl[w]); gu = fmax(gu, ru[w]); emax = fmax(emax, re[w]);
    }
    const double tnorm = fmax(fabs(gl), fabs(gu));
    const double eps = 2.220446049250313e-16;
    gl -= 2.0 * tnorm * eps * n + 2.0 * 2.2250738585072014e-308;
    gu += 2.0 * tnorm * eps * n + 2.0 * 2.2250738585072014e-308;
    out[0] = gl; out[1] = gu;
    out[2] = 2.2250738585072014e-308 * fmax(1.0, emax * emax);
    out[3] = tnorm;
  }
}

// Number of eigenvalues of T below x (Sturm sequence), division free: the count is the number of sign
// changes in p_0 = 1, p_1 = d_0 - x, p_{k+1} = (d_k - x) p_k - e_{k-1}^2 p_{k-1} (the leading principal
// minors of T - x I).  The quotient form q_k = p_k / p_{k-1} puts a reciprocal, its Newton step and a
// clamp on the critical path of every row (~210 cycles per row measured: 3.7 ms per matrix with the chip
// idle behind one wave per eigenvalue); here the dependent chain is ONE fma per row, everything else
// (d_k - x, e^2 p_{k-1}, the sign bookkeeping) is off it.  Entries are scaled by 1 / ||T|| so that a minor
// grows by at most ~3x per row (scale_tridiag_kernel prepares d / ||T|| and (e / ||T||)^2 once), and both running minors are renormalised by a power of two every 8 rows
// (exponent arithmetic only: no rounding).  An exactly zero minor counts as positive, which gives the
// same total as LAPACK's "pivot = -pivmin" rule (the next minor then has the sign opposite to the one
// before the zero).
template <typename Row>   // Row(k) -> {d_k / |T|, (e_{k-1} / |T|)^2}
__device__ __forceinline__ int sturm_count(Row row, int n, double xs) {
  double pm = 1.0, p = row(0).x - xs;
  int cnt = p < 0.0;
  int k = 1;
  for (; k + 8 <= n; k += 8) {
    // sign changes of the eight steps as bits: the high words' signs xor-ed, shifted into `bits` (v_xor + v_alignbit a
    // row instead of two f64 compares and their mask arithmetic -- the loop is bound by its VALU instruction count,
    // ~10 a row: 134 cycles a row-step with two waves a SIMD), counted once per group
    unsigned bits = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double2 t = row(k + u);
      const double pn = fma(t.x - xs, p, -(t.y * pm));
      bits = __builtin_amdgcn_alignbit(bits, (unsigned)(__double2hiint(pn) ^ __double2hiint(p)), 31);
      pm = p; p = pn;
    }
    cnt += __builtin_popcount(bits);
    // renormalise: max(|p|, |pm|) back to [1, 2); two zero minors in a row (a split matrix hit exactly)
    // restart the sequence below the split
    const double m = fmax(fabs(p), fabs(pm));
    if (m == 0.0) { p = 1.0; pm = 0.0; }
    else {
      const int ex = -ilogb(m);
      p = ldexp(p, ex); pm = ldexp(pm, ex);
    }
  }
  for (; k < n; ++k) {
    const double2 t = row(k);
    const double pn = fma(t.x - xs, p, -(t.y * pm));
    cnt += (pn < 0.0) != (p < 0.0);
    pm = p; p = pn;
  }
  return cnt;
}

// One wave per eigenvalue index k: every round the 64 lanes count the eigenvalues below 64
// interior points of the current bracket (multisection), which shrinks it 65-fold -- 9 rounds
// from the Gershgorin interval to machine precision instead of 53 bisection steps.
// With LDS = true (n <= 4096) the workgroup first copies the scaled tridiagonal into LDS as {d_k, e_{k-1}^2} pairs
// and the four waves read every row from there (one broadcast ds_read_b128, pipelined by the compiler): read as
// wave-uniform scalar loads from memory, every batch of 8 rows waited for a scalar-memory round trip.
template <bool LDS>
__global__ __launch_bounds__(256) void tridiag_bisect_kernel(const double* __restrict__ ds,
                                                             const double* __restrict__ es2, int n,
                                                             const double* __restrict__ bounds,
                                                             double* __restrict__ lam, const int first) {
  extern __shared__ double2 tri[];
  if (LDS) {
    for (int r = threadIdx.x; r < n; r += 256) tri[r] = double2{ds[r], r > 0 ? es2[r - 1] : 0.0};
    __syncthreads();
  }
  const int k = first + blockIdx.x * 4 + (threadIdx.x >> 6);  // eigenvalue indices first .. n-1
  const int lane = threadIdx.x & 63;
  if (k >= n) return;
  double lo = bounds[0], hi = bounds[1];
  const double pivmin = bounds[2];
  const double tnorm = bounds[3];
  const double sc = tnorm > 0.0 ? 1.0 / tnorm : 1.0;
  const double eps = 2.220446049250313e-16;
  for (int round = 0; round < 16; ++round) {
    const double h = (hi - lo) / 65.0;
    if (hi - lo <= 2.0 * eps * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin || !(h > 0.0)) break;
    const double x = lo + (double)(lane + 1) * h;
    int cnt;
    if (LDS) cnt = sturm_count([&](int r) { return tri[r]; }, n, x * sc);
    else cnt = sturm_count([&](int r) { return double2{ds[r], r > 0 ? es2[r - 1] : 0.0}; }, n, x * sc);
    // lanes whose point is still <= lambda_k form a prefix; only the prefix is trusted (in floating point
    // the count of the product form need not be monotone within rounding distance of an eigenvalue)
    const unsigned long long ok = __ballot(cnt <= k);
    const int c = (~ok == 0ull) ? 64 : __builtin_ctzll(~ok);
    const double nlo = lo + (double)c * h;
    hi = (c == 64) ? hi : fmin(hi, lo + (double)(c + 1) * h);
    lo = fmax(lo, nlo);
  }
  if (lane == 0) lam[k] = 0.5 * (lo + hi);
}

// ---- inverse iteration: thread k computes the eigenvector of T for lam[k].
// (T - lam I) = P L U with partial pivoting between neighbouring rows (U has two
// superdiagonals); the factors of all vectors are interleaved [row][vector] so that the 64
// lanes of a wave touch consecutive addresses.  Three solves from a hashed start vector, the
// iterate lives in the output matrix Y [n][ldy] (column k), normalised to unit 2-norm.
struct InvitWs {
  double* U1i;  // reciprocal pivots
  double* U2;
  double* U3;
  double* Lm;
  unsigned char* sw;
};

__device__ __forceinline__ double hash_uniform(unsigned int i, unsigned int k) {
  unsigned long long z = ((unsigned long long)i << 32 | k) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;  // uniform in [-1, 1)
}

__global__ __launch_bounds__(64) void tridiag_invit_kernel(const double* __restrict__ d, const double* __restrict__ e,
                                                          int n, const double* __restrict__ lam,
                                                          const double* __restrict__ bounds, int nvec, InvitWs ws,
                                                          double* __restrict__ Y, int64_t ldy, const int niter,
                                                          const int* __restrict__ list, const int* __restrict__ count,
                                                          const int list_first) {
  // with a work list (the vectors the twisted-factorisation kernel gave up on): entry list_first + blockIdx.x * 64 +
  // lane of it (the entries below list_first are the wave kernel's).  An entry k >= 0 restarts from the twisted vector
  // in column k of Y, an entry ~k < 0 from the hashed vector (the twisted vector is zero, not finite or not an
  // eigenvector to rounding: tridiag_twist_kernel).
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  bool from_memory = false;
  if (list) {
    k += list_first;
    if (k >= *count) return;
    k = list[k];
    from_memory = k >= 0;
    if (k < 0) k = ~k;
  } else if (k >= nvec) {
    return;
  }
  const double tnorm = bounds[3];
  const double tiny = fmax(2.220446049250313e-16 * tnorm, 2.2250738585072014e-308 * 4.0);
  const double lk = lam[k];
  const int64_t S = nvec;  // stride between consecutive rows of the interleaved factor arrays
  double* __restrict__ U1i = ws.U1i + k;
  double* __restrict__ U2 = ws.U2 + k;
  double* __restrict__ U3 = ws.U3 + k;
  double* __restrict__ Lm = ws.Lm + k;
  unsigned char* __restrict__ sw = ws.sw + k;
  double* __restrict__ y = Y + k;

  // A vector from the work list starts from what the twisted-factorisation kernel left in Y -- an eigenvector up to a
  // residual of ~1e-13 |T| that was only too large for its gap -- and ONE inverse iteration from there leaves the
  // neighbours' components at (residual / gap) x (eigenvalue error / gap) ~ 1e-16: three passes over the rows
  // (factorisation, forward, backward) instead of seven from a random start (3.4 -> ~1.5 ms at n = 4096; a Llama
  // block's three (4096, 2048) problems refuse about twenty vectors each).
  const int niter_eff = from_memory ? 1 : niter;
  // factorisation, with the forward sweep of the FIRST iteration riding along: the start vector is a hash of
  // (row, vector), so P L^-1 y0 needs nothing from memory and a separate latency-bound pass is saved
  double u = d[0] - lk, v = n > 1 ? e[0] : 0.0, w = 0.0;
  double cur0 = hash_uniform(0u, (unsigned)k);
  for (int i = 0; i < n - 1; ++i) {
    const double b = e[i], a1 = d[i + 1] - lk, c1 = (i + 2 < n) ? e[i + 1] : 0.0;
    double nxt = hash_uniform((unsigned)(i + 1), (unsigned)k);
    double m;
    if (fabs(u) >= fabs(b)) {
      if (fabs(u) < tiny) u = (u < 0.0) ? -tiny : tiny;
      const double ui = 1.0 / u;
      m = b * ui;
      U1i[i * S] = ui; U2[i * S] = v; U3[i * S] = w; Lm[i * S] = m; sw[i * S] = 0;
      u = a1 - m * v; v = c1 - m * w; w = 0.0;
    } else {
      const double bi = 1.0 / b;
      m = u * bi;
      U1i[i * S] = bi; U2[i * S] = a1; U3[i * S] = c1; Lm[i * S] = m; sw[i * S] = 1;
      u = v - m * a1; v = w - m * c1; w = 0.0;
      const double t = cur0; cur0 = nxt; nxt = t;
    }
    if (!from_memory) {
      y[(int64_t)i * ldy] = cur0;
      cur0 = nxt - m * cur0;
    }
  }
  if (fabs(u) < tiny) u = (u < 0.0) ? -tiny : tiny;
  U1i[(int64_t)(n - 1) * S] = 1.0 / u;

  // Each step of the two substitution sweeps depends on the previous one only through one
  // register, so the operands of 8 steps are fetched together (one latency per 8 steps).
  constexpr int UB = 12;  // 4 arrays x 12 rows = 48 loads per lane in flight (the counter holds 63; 16 rows stall, 15 = 12)
  double carry = 1.0;  // scale of the iterate in memory, applied on the next read
  for (int it = 0; it < niter_eff; ++it) {
    // forward: apply the row interchanges and L^-1 (done above for the first iteration of a hashed start)
    const bool fused = it == 0 && !from_memory;
    double cur = fused ? cur0 : y[0] * carry;
    for (int i0 = fused ? n : 0; i0 < n - 1; i0 += UB) {
      double yn[UB], lm[UB];
      unsigned char s8[UB];
#pragma unroll
      for (int q = 0; q < UB; ++q) {
        const int i = min(i0 + q, n - 2);
        yn[q] = y[(int64_t)(i + 1) * ldy];
        lm[q] = Lm[i * S];
        s8[q] = sw[i * S];
      }
#pragma unroll
      for (int q = 0; q < UB; ++q) {
        const int i = i0 + q;
        if (i < n - 1) {
          double nxt = yn[q] * carry;
          if (s8[q]) { const double t = cur; cur = nxt; nxt = t; }
          y[(int64_t)i * ldy] = cur;
          cur = nxt - lm[q] * cur;
        }
      }
    }
    // backward: U x = y
    double x2 = 0.0, x1 = cur * U1i[(int64_t)(n - 1) * S];
    y[(int64_t)(n - 1) * ldy] = x1;
    double ss = x1 * x1, big = fabs(x1);
    for (int i0 = n - 2; i0 >= 0; i0 -= UB) {
      double yv[UB], u1[UB], u2[UB], u3[UB];
#pragma unroll
      for (int q = 0; q < UB; ++q) {
        const int i = max(i0 - q, 0);
        yv[q] = y[(int64_t)i * ldy];
        u1[q] = U1i[i * S];
        u2[q] = U2[i * S];
        u3[q] = U3[i * S];
      }
#pragma unroll
      for (int q = 0; q < UB; ++q) {
        const int i = i0 - q;
        if (i >= 0) {
          const double x0 = (yv[q] - u2[q] * x1 - u3[q] * x2) * u1[q];
          y[(int64_t)i * ldy] = x0;
          x2 = x1; x1 = x0;
          big = fmax(big, fabs(x0));
          ss += x0 * x0;
        }
      }
    }
    // normalisation factor (guard the sum of squares against overflow through the max entry)
    if (big > 1e140 || !(ss < INFINITY)) {
      double s2 = 0.0;
      for (int i = 0; i < n; ++i) { const double t = y[(int64_t)i * ldy] / big; s2 += t * t; }
      carry = 1.0 / (big * sqrt(s2));
    } else {
      carry = ss > 0.0 ? 1.0 / sqrt(ss) : 0.0;
    }
  }
  // the final scaling is left to invit_scale_kernel (fully parallel over the matrix; done here it is one more
  // latency-bound pass of n / UB round trips per lane): the factor goes to row 0 of this vector's Lm column
  Lm[0] = carry;
}

// ---- inverse iteration, one WAVE per vector (round 5): the work list of the twisted-factorisation kernel.
// The kernel above runs one vector per LANE: every row step waits for memory once per 12 rows and the launch takes
// 3.4 ms at n = 4096 whether the list holds two vectors or two thousand (a (4096, 2048) problem refuses about twenty).
// Here a wave owns ONE vector.  The arithmetic is the lane kernel's, step for step -- pivoted LU of T - lambda I, the
// forward sweep riding in the factorisation, back substitution, 2-norm -- but the rows are fetched 64 at a time (lane
// l holds row i0 + l: one coalesced request per array and chunk, the next chunk's already in flight), the 64 dependent
// steps of a chunk run on wave-uniform values broadcast from the owning lane with v_readlane, and lane q keeps row q's
// results until the chunk is stored.  The factors of list entry j live in rows [j n, (j + 1) n) of the factor arrays
// (the multipliers behind the first nvec entries of Lm, which are the scale factors invit_scale_kernel reads).
__device__ __forceinline__ double bcast_lane(double x, int q) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), q);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), q);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void tridiag_invit_wave_kernel(const double* __restrict__ d, const double* __restrict__ e,
                                                               int n, const double* __restrict__ lam,
                                                               const double* __restrict__ bounds, int nvec, InvitWs ws,
                                                               double* Y, int64_t ldy, const int niter,
                                                               const int* __restrict__ list,
                                                               const int* __restrict__ count, const int wave_max) {
  const int j = blockIdx.x;
  if (j >= min(*count, wave_max)) return;
  int k = list[j];
  const bool from_memory = k >= 0;       // (wave-uniform)
  if (k < 0) k = ~k;
  const int lane = threadIdx.x;
  const double tnorm = bounds[3];
  const double tiny = fmax(2.220446049250313e-16 * tnorm, 2.2250738585072014e-308 * 4.0);
  const double lk = lam[k];
  double* U1i = ws.U1i + (int64_t)j * n;
  double* U2 = ws.U2 + (int64_t)j * n;
  double* U3 = ws.U3 + (int64_t)j * n;
  double* Lm = ws.Lm + nvec + (int64_t)j * n;
  unsigned char* sw = ws.sw + (int64_t)j * n;
  double* y = Y + k;
  const int niter_eff = from_memory ? 1 : niter;

  // factorisation + the forward sweep of the first iteration
  double u = d[0] - lk, v = n > 1 ? e[0] : 0.0, w = 0.0;
  double cur0 = from_memory ? y[0] : hash_uniform(0u, (unsigned)k);
  // chunk operands of lane l: row i = i0 + l; b = e[i], a1 = d[i + 1] - lambda, c1 = e[i + 1], start value of row i + 1
  auto load_chunk = [&](int i0, double& eb, double& da, double& ec, double& ys) {
    const int i = i0 + lane;
    const bool live = i < n - 1;
    eb = live ? e[i] : 1.0;
    da = live ? d[i + 1] - lk : 0.0;
    ec = (i + 2 < n) ? e[i + 1] : 0.0;
    ys = !live ? 0.0 : (from_memory ? y[(int64_t)(i + 1) * ldy] : hash_uniform((unsigned)(i + 1), (unsigned)k));
  };
  double eb, da, ec, ys;
  load_chunk(0, eb, da, ec, ys);
  for (int i0 = 0; i0 < n - 1; i0 += 64) {
    double eb_n = 0.0, da_n = 0.0, ec_n = 0.0, ys_n = 0.0;
    if (i0 + 64 < n - 1) load_chunk(i0 + 64, eb_n, da_n, ec_n, ys_n);
    double o_u1 = 0.0, o_u2 = 0.0, o_u3 = 0.0, o_lm = 0.0, o_y = 0.0;
    int o_sw = 0;
    const int cnt = min(64, n - 1 - i0);
    for (int q = 0; q < cnt; ++q) {
      const double b = bcast_lane(eb, q), a1 = bcast_lane(da, q), c1 = bcast_lane(ec, q);
      double nxt = bcast_lane(ys, q);
      double m, r1, r2, r3;
      int s;
      if (fabs(u) >= fabs(b)) {
        if (fabs(u) < tiny) u = (u < 0.0) ? -tiny : tiny;
        const double ui = 1.0 / u;
        m = b * ui;
        r1 = ui; r2 = v; r3 = w; s = 0;
        u = a1 - m * v; v = c1 - m * w; w = 0.0;
      } else {
        const double bi = 1.0 / b;
        m = u * bi;
        r1 = bi; r2 = a1; r3 = c1; s = 1;
        u = v - m * a1; v = w - m * c1; w = 0.0;
        const double t = cur0; cur0 = nxt; nxt = t;
      }
      if (lane == q) { o_u1 = r1; o_u2 = r2; o_u3 = r3; o_lm = m; o_sw = s; o_y = cur0; }
      cur0 = nxt - m * cur0;
    }
    const int i = i0 + lane;
    if (i < n - 1) {
      U1i[i] = o_u1; U2[i] = o_u2; U3[i] = o_u3; Lm[i] = o_lm; sw[i] = (unsigned char)o_sw;
      y[(int64_t)i * ldy] = o_y;
    }
    eb = eb_n; da = da_n; ec = ec_n; ys = ys_n;
  }
  if (fabs(u) < tiny) u = (u < 0.0) ? -tiny : tiny;
  const double u1_last = 1.0 / u;

  double carry = 1.0;   // scale of the iterate in memory, applied on the next read
  for (int it = 0; it < niter_eff; ++it) {
    double cur = cur0;
    if (it > 0) {
      // forward: row interchanges and L^-1 on the iterate in memory
      cur = y[0] * carry;
      for (int i0 = 0; i0 < n - 1; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < n - 1;
        const double yn = live ? y[(int64_t)(i + 1) * ldy] * carry : 0.0;
        const double lm = live ? Lm[i] : 0.0;
        const int s8 = live ? (int)sw[i] : 0;
        double o_y = 0.0;
        const int cnt = min(64, n - 1 - i0);
        for (int q = 0; q < cnt; ++q) {
          double nxt = bcast_lane(yn, q);
          const double m = bcast_lane(lm, q);
          if (__builtin_amdgcn_readlane(s8, q)) { const double t = cur; cur = nxt; nxt = t; }
          if (lane == q) o_y = cur;
          cur = nxt - m * cur;
        }
        if (live) y[(int64_t)i * ldy] = o_y;
      }
    }
    // backward: U x = y
    double x2 = 0.0, x1 = cur * u1_last;
    if (lane == 0) y[(int64_t)(n - 1) * ldy] = x1;
    double ss = x1 * x1, big = fabs(x1);
    for (int i0 = n - 2; i0 >= 0; i0 -= 64) {
      const int i = i0 - lane;
      const bool live = i >= 0;
      const double yv = live ? y[(int64_t)i * ldy] : 0.0;
      const double u1 = live ? U1i[i] : 0.0, u2 = live ? U2[i] : 0.0, u3 = live ? U3[i] : 0.0;
      double o_x = 0.0;
      const int cnt = min(64, i0 + 1);
      for (int q = 0; q < cnt; ++q) {
        const double x0 = (bcast_lane(yv, q) - bcast_lane(u2, q) * x1 - bcast_lane(u3, q) * x2) * bcast_lane(u1, q);
        if (lane == q) o_x = x0;
        x2 = x1; x1 = x0;
        big = fmax(big, fabs(x0));
        ss += x0 * x0;
      }
      if (live) y[(int64_t)i * ldy] = o_x;
    }
    // normalisation factor (guard the sum of squares against overflow through the max entry)
    if (big > 1e140 || !(ss < INFINITY)) {
      double s2 = 0.0;
      for (int i = lane; i < n; i += 64) { const double t = y[(int64_t)i * ldy] / big; s2 += t * t; }
      for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
      carry = 1.0 / (big * sqrt(s2));
    } else {
      carry = ss > 0.0 ? 1.0 / sqrt(ss) : 0.0;
    }
  }
  if (lane == 0) ws.Lm[k] = carry;      // invit_scale_kernel multiplies column k by this
}

// ---- eigenvectors of T by twisted factorisations, one WAVE per eigenvector, the recurrences as scans ----
// The inverse iteration above runs one eigenvector per LANE: 4 n sequential steps each, a memory round trip per 12
// rows, 16 waves on a chip of 1024 SIMDs (3.3 ms at n = 4096, k = 1024).  Here lane c owns rows 64 c .. 64 c + 63.
//   forward pivots   s_0 = a_0, s_i = a_i - b_{i-1} / s_{i-1}           (a = (d - lambda) / |T|, b = (e / |T|)^2)
//   backward pivots  t_{n-1} = a_{n-1}, t_i = a_i - b_i / t_{i+1}
//   gamma_i = s_i + t_i - a_i;  twist r = argmin |gamma|;  z_r = 1,
//   z_i = -(e_i / s_i) z_{i+1} above r,  z_i = -(e_{i-1} / t_i) z_{i-1} below      (Parlett & Dhillon; LAPACK dlar1v)
// A pivot recurrence is a Moebius map, i.e. the three-term recurrence of the leading (trailing) minors
// p_{i+1} = a_i p_i - b_{i-1} p_{i-1}: a lane multiplies the 2 x 2 matrices of its 64 rows, a wave scan of those
// products (rescaled by powers of two) gives every lane the minors just before its first row, hence its first
// pivot, and the lane then runs the ordinary recurrence over its own rows.  z is a running product of links: the
// same pattern with (mantissa, exponent) pairs, since entries far from the twist underflow legitimately.
// The result is checked row by row against (T - lambda) z = gamma_r e_r.  A residual rho leaves a component
// ~rho / gap along the neighbouring eigenvector, so a vector is accepted when rho <= TW_ORTH x its gap (and
// rho <= TW_TOL); the others -- the members of tight clusters, mostly -- go on a list that the inverse-iteration kernel
// works off afterwards (empty for spectra with relative gaps above ~1e-5, e.g. the top of a covariance spectrum).
constexpr int TW_MAXN = 4096;
constexpr double TW_TOL = 2e-13;
constexpr double TW_ORTH = 1e-9;
constexpr double TW_PIVMIN = 1e-290;

// dsT[il * 64 + c] = d[64 c + il] / |T|, esT likewise for e: lane c's rows as coalesced loads
__global__ void tw_prepare_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                  const double* __restrict__ bounds, double* __restrict__ dsT,
                                  double* __restrict__ esT, int* __restrict__ count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int R = (n + 63) / 64;
  if (t == 0) *count = 0;
  if (t >= R * 64) return;
  const int il = t >> 6, c = t & 63, i = c * R + il;
  const double sc = bounds[3] > 0.0 ? 1.0 / bounds[3] : 1.0;
  dsT[t] = i < n ? d[i] * sc : 0.0;
  esT[t] = i + 1 < n ? e[i] * sc : 0.0;
}

struct M22 { double a, b, c, d; };   // [[a, b], [c, d]], defined up to a positive factor
__device__ __forceinline__ void m22_renorm(M22& m) {
  const double mx = fmax(fmax(fabs(m.a), fabs(m.b)), fmax(fabs(m.c), fabs(m.d)));
  if (mx > 0.0 && mx < INFINITY) {
    int ex;
    (void)frexp(mx, &ex);
    m.a = ldexp(m.a, -ex); m.b = ldexp(m.b, -ex); m.c = ldexp(m.c, -ex); m.d = ldexp(m.d, -ex);
  }
}
__device__ __forceinline__ M22 m22_mul(const M22& x, const M22& y) {   // x y
  return M22{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
__device__ __forceinline__ M22 m22_shfl(const M22& m, int src) {
  return M22{__shfl(m.a, src), __shfl(m.b, src), __shfl(m.c, src), __shfl(m.d, src)};
}
// (mantissa, exponent) products
__device__ __forceinline__ void me_mul(double& m, int& e, double f) {
  m *= f;
  if (m != 0.0 && fabs(m) < INFINITY) { int ex; m = frexp(m, &ex); e += ex; }
}

__global__ __launch_bounds__(64) void tridiag_twist_kernel(const double* __restrict__ dsT, const double* __restrict__ esT,
                                                           int n, const double* __restrict__ lam,
                                                           const double* __restrict__ bounds, int nvec,
                                                           double* __restrict__ Y, int64_t ldy,
                                                           double* __restrict__ scale_out, int* __restrict__ fail_list,
                                                           int* __restrict__ fail_count, int has_below,
                                                           int break_every) {
  extern __shared__ __attribute__((aligned(16))) char tw_smem[];
  double* S = reinterpret_cast<double*>(tw_smem);        // [R][64]: forward pivots, then z mantissas, then z
  double* Tt = S + 64 * 64;                               // [R][64]: backward pivots, then z exponents
  const int k = blockIdx.x, lane = threadIdx.x;
  const int R = (n + 63) / 64;
  const int i0 = lane * R;                                // this lane's rows i0 .. i0 + R - 1
  const double tnorm = bounds[3];
  double ls = tnorm > 0.0 ? lam[k] / tnorm : lam[k];
#define TW_A(il) (dsT[(il) * 64 + lane] - ls)
#define TW_E(il) (esT[(il) * 64 + lane])
  // e of the row above this lane's first row (links the chunks)
  const double e_above = __shfl_up(esT[(R - 1) * 64 + lane], 1);   // lane - 1's last row; lane 0: unused
  // Two passes: the first one's twisted factorisation yields the Rayleigh-quotient correction of the shift,
  // lambda + gamma_r / z^T z (dlarrv's RQI step) -- the multisection leaves ~1e-14 |T| of error in lambda, and a vector
  // computed at a shift that far off carries that much residual; at the corrected shift it is at rounding level.
  double ss = 0.0, nrm = 0.0;
  int r = 0;
  for (int pass = 0; pass < 2; ++pass) {
  // ---- forward: product of this lane's row matrices, scan, first pivot, pivots
  M22 m{1.0, 0.0, 0.0, 1.0};
  for (int il = 0; il < R; ++il) {
    const int i = i0 + il;
    if (i >= n) break;
    const double a = TW_A(il);
    const double ep = il == 0 ? (lane == 0 ? 0.0 : e_above) : TW_E(il - 1);
    const double b = ep * ep;
    m = M22{a * m.a - b * m.c, a * m.b - b * m.d, m.a, m.b};
    if ((il & 7) == 7) m22_renorm(m);
  }
  m22_renorm(m);
  for (int off = 1; off < 64; off <<= 1) {
    const M22 o = m22_shfl(m, max(lane - off, 0));
    if (lane >= off) { m = m22_mul(m, o); m22_renorm(m); }
  }
  {
    // state before this lane's first row = first column of the product up to lane - 1: (p_{i0}, p_{i0-1})
    const double pn = __shfl_up(m.a, 1), pd = __shfl_up(m.c, 1);
    double sprev_inv = (lane == 0 || pn == 0.0) ? 0.0 : pd / pn;   // 1 / s_{i0-1}; p_{i0} = 0: pivot 0 -> handled as pivmin
    if (lane > 0 && pn == 0.0) sprev_inv = -1.0 / TW_PIVMIN;
    double s = 0.0;
    for (int il = 0; il < R; ++il) {
      const int i = i0 + il;
      if (i >= n) break;
      const double a = TW_A(il);
      const double ep = il == 0 ? (lane == 0 ? 0.0 : e_above) : TW_E(il - 1);
      const double b = ep * ep;
      s = (il == 0) ? a - b * sprev_inv : a - b / s;
      if (!(fabs(s) >= TW_PIVMIN)) s = -TW_PIVMIN;
      S[il * 64 + lane] = s;
    }
  }
  // ---- backward: the same from the bottom; gamma and its smallest magnitude on the way
  M22 w{1.0, 0.0, 0.0, 1.0};
  for (int il = R - 1; il >= 0; --il) {
    const int i = i0 + il;
    if (i >= n) continue;
    const double a = TW_A(il);
    const double e_ = TW_E(il);                            // zero for i = n - 1
    const double b = e_ * e_;
    w = M22{a * w.a - b * w.c, a * w.b - b * w.d, w.a, w.b};
    if ((il & 7) == 0) m22_renorm(w);
  }
  m22_renorm(w);
  for (int off = 1; off < 64; off <<= 1) {
    const M22 o = m22_shfl(w, min(lane + off, 63));
    if (lane + off < 64) { w = m22_mul(w, o); m22_renorm(w); }
  }
  double gbest = INFINITY, gsign = 0.0;
  int rbest = 0;
  {
    const double qn = __shfl_down(w.a, 1), qd = __shfl_down(w.c, 1);   // (q_{i1}, q_{i1+1}) of the rows below
    double tnext_inv = (lane == 63 || qn == 0.0) ? 0.0 : qd / qn;
    if (lane < 63 && qn == 0.0) tnext_inv = -1.0 / TW_PIVMIN;
    double t = 0.0;
    bool first = true;
    for (int il = R - 1; il >= 0; --il) {
      const int i = i0 + il;
      if (i >= n) continue;
      const double a = TW_A(il);
      const double e_ = TW_E(il);
      const double b = e_ * e_;
      t = first ? a - b * tnext_inv : a - b / t;
      first = false;
      if (!(fabs(t) >= TW_PIVMIN)) t = -TW_PIVMIN;
      Tt[il * 64 + lane] = t;
      const double gs = S[il * 64 + lane] + t - a;
      const double g = fabs(gs);
      if (g < gbest) { gbest = g; rbest = i; gsign = gs; }  // (NaN never wins)
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double og = __shfl_xor(gbest, off), os = __shfl_xor(gsign, off);
    const int orr = __shfl_xor(rbest, off);
    if (og < gbest || (og == gbest && orr < rbest)) { gbest = og; rbest = orr; gsign = os; }
  }
  r = rbest;
  const int cr = r / R;
  // ---- z as running products of links, (mantissa, exponent) per row; S <- mantissa, Tt <- exponent
  int* Ex = reinterpret_cast<int*>(Tt);                    // one int per slot (the doubles of Tt are consumed first)
  double vm = 1.0; int ve = 0;                              // this lane's total link product towards the twist
  if (lane <= cr) {                                         // rows above the twist: z_i = f_i z_{i+1}, f_i = -e_i / s_i
    double pm = 1.0; int pe = 0;
    for (int il = R - 1; il >= 0; --il) {
      const int i = i0 + il;
      if (i >= n || i > r) continue;
      if (i == r) { S[il * 64 + lane] = 1.0; Ex[2 * (il * 64 + lane)] = 0; continue; }
      const double f = -TW_E(il) / S[il * 64 + lane];
      me_mul(pm, pe, f);
      S[il * 64 + lane] = pm; Ex[2 * (il * 64 + lane)] = pe;
    }
    vm = pm; ve = pe;
  }
  if (lane >= cr) {                                         // rows below: z_i = g_i z_{i-1}, g_i = -e_{i-1} / t_i
    double pm = 1.0; int pe = 0;
    for (int il = 0; il < R; ++il) {
      const int i = i0 + il;
      if (i >= n) break;
      if (i <= r) continue;
      const double ep = il == 0 ? e_above : TW_E(il - 1);
      const double g = -ep / Tt[il * 64 + lane];
      me_mul(pm, pe, g);
      S[il * 64 + lane] = pm; Ex[2 * (il * 64 + lane)] = pe;
    }
    if (lane > cr) { vm = pm; ve = pe; }
  }
  // base factor of a lane: the product of the totals of the lanes between it and the twist lane (the twist lane's
  // partial products above / below r count as its totals in the two directions)
  double upm = vm; int upe = ve;                            // totals used by the lanes ABOVE (smaller index)
  double dnm = vm; int dne = ve;                            // totals used by the lanes BELOW
  if (lane == cr) {
    // vm / ve currently hold the ABOVE partial if rows above r exist in this lane, recompute both explicitly
    double am = 1.0; int ae = 0, be = 0; double bm = 1.0;
    const int ilr = r - i0;
    if (ilr > 0) { am = S[0 * 64 + lane]; ae = Ex[2 * (0 * 64 + lane)]; }
    const int last = min(R - 1, n - 1 - i0);
    if (last > ilr) { bm = S[last * 64 + lane]; be = Ex[2 * (last * 64 + lane)]; }
    upm = am; upe = ae; dnm = bm; dne = be;
  }
  if (lane > cr) { upm = 1.0; upe = 0; }
  if (lane < cr) { dnm = 1.0; dne = 0; }
  // exclusive suffix product of upm over lanes (what the lanes above multiply by), exclusive prefix product of dnm
  double bum = 1.0; int bue = 0, bde = 0; double bdm = 1.0;
  {
    double sm = upm; int se = upe;                          // inclusive suffix
    for (int off = 1; off < 64; off <<= 1) {
      const double om = __shfl_down(sm, off); const int oe = __shfl_down(se, off);
      if (lane + off < 64) { sm *= om; se += oe; if (sm != 0.0 && fabs(sm) < INFINITY) { int ex; sm = frexp(sm, &ex); se += ex; } }
    }
    bum = __shfl_down(sm, 1); bue = __shfl_down(se, 1);
    if (lane == 63) { bum = 1.0; bue = 0; }
    double qm = dnm; int qe = dne;                          // inclusive prefix
    for (int off = 1; off < 64; off <<= 1) {
      const double om = __shfl_up(qm, off); const int oe = __shfl_up(qe, off);
      if (lane >= off) { qm *= om; qe += oe; if (qm != 0.0 && fabs(qm) < INFINITY) { int ex; qm = frexp(qm, &ex); qe += ex; } }
    }
    bdm = __shfl_up(qm, 1); bde = __shfl_up(qe, 1);
    if (lane == 0) { bdm = 1.0; bde = 0; }
  }
  // ---- assemble: exponent range, scaling, 2-norm
  int emax = -(1 << 30);
  for (int il = 0; il < R; ++il) {
    const int i = i0 + il;
    if (i >= n) break;
    double fm = 1.0; int fe = 0;
    if (i < r && lane < cr) { fm = bum; fe = bue; }
    if (i > r && lane > cr) { fm = bdm; fe = bde; }
    double zm = S[il * 64 + lane] * fm;
    int ze = Ex[2 * (il * 64 + lane)] + fe;
    if (zm != 0.0 && fabs(zm) < INFINITY) { int ex; zm = frexp(zm, &ex); ze += ex; } else if (!(fabs(zm) < INFINITY)) { zm = 0.0; ze = -(1 << 29); }
    S[il * 64 + lane] = zm; Ex[2 * (il * 64 + lane)] = ze;
    if (zm != 0.0) emax = max(emax, ze);
  }
  for (int off = 32; off > 0; off >>= 1) emax = max(emax, __shfl_xor(emax, off));
  ss = 0.0;
  for (int il = 0; il < R; ++il) {
    const int i = i0 + il;
    if (i >= n) break;
    const int de = Ex[2 * (il * 64 + lane)] - emax;
    const double z = de < -1100 ? 0.0 : ldexp(S[il * 64 + lane], de);
    S[il * 64 + lane] = z;
    ss += z * z;
  }
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  nrm = ss > 0.0 ? 1.0 / sqrt(ss) : 0.0;
  if (pass == 0) {
    // z^T z in the scale where z_r = 1 is ss 4^emax; the correction is taken only while it is a refinement
    const double corr = (ss > 0.0 && emax < 400) ? ldexp(gsign / ss, -2 * emax) : 0.0;
    if (fabs(corr) <= 1e-11) ls += corr;
    __syncthreads();   // (the pivots of the second pass overwrite S / Tt)
  }
  }
  __syncthreads();   // (one wave; the barrier keeps the compiler from moving the neighbour reads above the writes)
  // ---- residual of (T - lambda) z, in units of |T| with |z|_2 = 1, and the output
  // (test hook PTD_TWIST_TEST_BREAK=m: every m-th vector leaves as a zero / NaN column on the list, as after a breakdown)
  const bool force_break = break_every > 0 && (k % break_every) == 0;
  double res = 0.0;
  for (int il = 0; il < R; ++il) {
    const int i = i0 + il;
    if (i >= n) break;
    const double z = S[il * 64 + lane];
    const double zu = il > 0 ? S[(il - 1) * 64 + lane] : (lane > 0 ? S[(R - 1) * 64 + lane - 1] : 0.0);
    const bool has_dn = i + 1 < n;
    const double zd = !has_dn ? 0.0 : (il + 1 < R ? S[(il + 1) * 64 + lane] : S[0 * 64 + lane + 1]);
    const double eu = il == 0 ? (lane == 0 ? 0.0 : e_above) : TW_E(il - 1);
    const double rr = i == r ? 0.0 : fabs(eu * zu + TW_A(il) * z + TW_E(il) * zd) * nrm;
    res = fmax(res, rr);
    Y[(int64_t)i * ldy + k] = force_break ? (((k / break_every) & 1) ? NAN : 0.0) : z * nrm;
  }
  for (int off = 32; off > 0; off >>= 1) res = fmax(res, __shfl_xor(res, off));
  if (lane == 0) {
    scale_out[k] = 1.0;                                     // invit_scale_kernel multiplies by this
    // gap to the neighbours in units of |T| (lam[-1] exists when an eigenvalue below the requested ones was computed)
    double gap = INFINITY;
    if (k + 1 < nvec) gap = fmin(gap, lam[k + 1] - lam[k]);
    if (k > 0 || has_below) gap = fmin(gap, lam[k] - lam[k - 1]);
    gap = tnorm > 0.0 ? gap / tnorm : gap;
    const double tol = fmin(TW_TOL, TW_ORTH * gap);
    // Refused.  A finite, non-zero vector that is an eigenvector to rounding and only too close to a neighbour for its
    // residual is restarted FROM (entry k); a breakdown (zero or non-finite vector, residual above TW_TOL) is restarted
    // from the hashed vector (entry ~k): one inverse iteration from a zero or NaN column would return it unchanged.
    const bool broken = !(ss > 0.0) || !(ss < INFINITY) || !(res <= TW_TOL) || force_break;
    if (broken || !(res <= tol)) fail_list[atomicAdd(fail_count, 1)] = broken ? ~k : k;
  }
#undef TW_A
#undef TW_E
}

// Y[:, k] *= scale[k]: the normalisation of the inverse-iteration vectors
__global__ void invit_scale_kernel(double* __restrict__ Y, int64_t ldy, int n, int nvec, const double* __restrict__ scale) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nvec) return;
  const double c = scale[k];
  for (int i = blockIdx.y; i < n; i += gridDim.y) Y[(int64_t)i * ldy + k] *= c;
}

// smallest gap between consecutive eigenvalues relative to |T|, and the longest chain of
// consecutive gaps below `ortol`: decides whether inverse iteration + chain
// re-orthogonalisation is safe and cheap (single workgroup, strided scan with carried runs)
__global__ void min_gap_kernel(const double* __restrict__ lam, int n, const double* __restrict__ bounds,
                               double ortol, double* __restrict__ out) {
  __shared__ double red[16];
  __shared__ int redc[16];
  const double tnorm = fmax(bounds[3], 2.2250738585072014e-308);
  const int nt = blockDim.x;
  // thread t scans the contiguous gap range [lo, hi): runs crossing a range boundary are
  // under-counted by at most a factor 2 per boundary, harmless for a threshold decision, but
  // to keep it exact a run is extended backwards into the previous range when it starts at lo
  const int ngaps = n - 1;
  const int per = (ngaps + nt - 1) / nt;
  const int lo = threadIdx.x * per, hi = min(ngaps, lo + per);
  double g = INFINITY;
  int longest = 0;
  if (lo < hi) {
    int run = 0;
    // extend backwards: gaps lo-1, lo-2, ... that are close belong to the run that enters this range
    for (int q = lo - 1; q >= 0 && (lam[q + 1] - lam[q]) < ortol * tnorm; --q) ++run;
    for (int q = lo; q < hi; ++q) {
      const double gap = lam[q + 1] - lam[q];
      g = fmin(g, gap);
      run = gap < ortol * tnorm ? run + 1 : 0;
      longest = max(longest, run);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    g = fmin(g, __shfl_xor(g, o));
    longest = max(longest, __shfl_xor(longest, o));
  }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = g; redc[threadIdx.x >> 6] = longest; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { g = fmin(g, red[w]); longest = max(longest, redc[w]); }
    out[0] = g / tnorm;
    out[1] = (double)longest;  // members of the longest chain minus one
  }
}

// Chains of consecutive eigenvalues closer than ortol |T|: inverse iteration leaves their
// vectors non-orthogonal at the level eps / gap, so they are orthogonalised explicitly
// (modified Gram-Schmidt along the chain; one workgroup per chain, chains are rare and short).
__global__ __launch_bounds__(256) void tridiag_chain_mgs_kernel(const double* __restrict__ lam, int n, int nvec,
                                                                const double* __restrict__ bounds, double ortol,
                                                                double* __restrict__ Y, int64_t ldy) {
  __shared__ double red[4];
  __shared__ double bc;
  const int k = blockIdx.x, tid = threadIdx.x;
  const double thr = ortol * fmax(bounds[3], 2.2250738585072014e-308);
  const bool closeL = k > 0 && (lam[k] - lam[k - 1]) < thr;
  const bool closeR = k < nvec - 1 && (lam[k + 1] - lam[k]) < thr;
  if (closeL || !closeR) return;  // not the first member of a chain
  for (int kk = k + 1; kk < nvec && (lam[kk] - lam[kk - 1]) < thr; ++kk) {
    for (int p = k; p <= kk; ++p) {  // p == kk: normalisation pass
      double s = 0.0;
      for (int i = tid; i < n; i += 256) s += Y[(int64_t)i * ldy + p] * Y[(int64_t)i * ldy + kk];
      s = wave_sum_d(s);
      if ((tid & 63) == 0) red[tid >> 6] = s;
      __syncthreads();
      if (tid == 0) bc = (red[0] + red[1]) + (red[2] + red[3]);
      __syncthreads();
      const double dot = bc;
      if (p < kk) {
        for (int i = tid; i < n; i += 256) Y[(int64_t)i * ldy + kk] -= dot * Y[(int64_t)i * ldy + p];
      } else {
        const double sc = dot > 0.0 ? 1.0 / sqrt(dot) : 0.0;
        for (int i = tid; i < n; i += 256) Y[(int64_t)i * ldy + kk] *= sc;
      }
      __syncthreads();
    }
  }
}

// ---- compact WY without forming T:  the block reflector of a panel is I - V^T T V with
// T^-1 = striu(V V^T) + diag(1 / tau)  (Joffrain et al., "Accumulating Householder
// transformations, revisited"), so W2 = T W1 is one 64-row upper-triangular solve per column
// of W1.  One thread per column, the column lives in registers, the 64 x 64 matrix in LDS
// (every lane reads the same entry: broadcast).  tau_i == 0 (identity reflector) gives a zero row.
// ---- compact-WY factors of ALL panels in three launches (they depend only on V and tau):
//   wy_gram      partial Gram matrices  G_p = V_p V_p^T  (V_p: the panel's 64 reflectors as rows)
//   wy_tfactor   T_p = (striu(G_p) + diag(1 / tau))^-1  (upper triangular; tau_i = 0 gives a zero row / column)
//   wy_tv        TV_p = T_p V_p
// so that the back-transformation of a panel is  Y -= V_p^T (TV_p Y): two products instead of a Gram
// product, a product, a triangular solve and a product.
constexpr int GCH = 512;  // columns of V_p per wy_gram workgroup

__global__ __launch_bounds__(256) void wy_gram_kernel(const double* __restrict__ Vall, int64_t ld, int n,
                                                      double* __restrict__ Gpart, int nchunks) {
  // (f64 matrix cores: wave w forms rows 16 w .. 16 w + 15 of the 64 x 64 Gram block, four 16 x 16 tiles, from a
  // 128-column image of the panel in LDS; MFMA operand map: a = A[row = lane & 15][k = lane >> 4],
  // b = B[k = lane >> 4][col = lane & 15], d[r] = D[row = (lane >> 4) + 4 r][col = lane & 15])
  __shared__ double Vs[NB][129];
  const int panel = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int r0 = panel * NB + 1;
  const double* Vp = Vall + (int64_t)panel * NB * ld;
  f64x4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) acc[cb] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int cbeg = r0 + chunk * GCH, cend = min(n, cbeg + GCH);
  for (int c0 = cbeg; c0 < cend; c0 += 128) {
    for (int e = tid; e < NB * 128; e += 256) {
      const int r = e >> 7, c = e & 127;
      Vs[r][c] = (c0 + c < cend) ? Vp[(int64_t)r * ld + c0 + c] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < 128; kk += 4) {
      const double av = Vs[16 * wid + l15][kk + l4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Vs[16 * cb + l15][kk + l4], acc[cb], 0, 0, 0);
    }
    __syncthreads();
  }
  double* out = Gpart + ((int64_t)panel * nchunks + chunk) * NB * NB;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(16 * wid + l4 + 4 * r) * NB + 16 * cb + l15] = acc[cb][r];
}

__global__ __launch_bounds__(256) void wy_tfactor_kernel(const double* __restrict__ Gpart, int nchunks,
                                                         const double* __restrict__ taus, int n,
                                                         double* __restrict__ Tall) {
  __shared__ double Ti[NB][NB + 1];  // striu(G), then the inverse is built column by column in registers
  __shared__ double tl[NB];
  const int panel = blockIdx.x, tid = threadIdx.x;
  const int j0 = panel * NB, cols = min(NB, n - j0);
  for (int e = tid; e < NB * NB; e += 256) {
    double g = 0.0;
    for (int c = 0; c < nchunks; ++c) g += Gpart[((int64_t)panel * nchunks + c) * NB * NB + e];  // fixed order
    const int r = e >> 6, q = e & 63;
    Ti[r][q] = (r < cols && q < cols && q > r) ? g : 0.0;
  }
  if (tid < NB) tl[tid] = tid < cols ? taus[j0 + tid] : 0.0;
  __syncthreads();
  if (tid < NB) {
    // column j of T solves (striu(G) + diag(1 / tau)) t = e_j by back substitution
    const int j = tid;
    double t[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) t[i] = 0.0;
#pragma unroll
    for (int i = NB - 1; i >= 0; --i) {
      if (i <= j) {
        double sacc = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int q = i + 1; q < NB; ++q) sacc -= Ti[i][q] * t[q];
        t[i] = sacc * tl[i];
      }
    }
    double* out = Tall + (int64_t)panel * NB * NB;
#pragma unroll
    for (int i = 0; i < NB; ++i) out[i * NB + j] = t[i];
  }
}

__global__ __launch_bounds__(256) void wy_tv_kernel(const double* __restrict__ Vall, int64_t ld, int n,
                                                    const double* __restrict__ Tall, double* __restrict__ TVall) {
  __shared__ double Ts[NB][NB + 1];
  __shared__ double Vs[NB][129];
  const int panel = blockIdx.y, tid = threadIdx.x;
  const int r0 = (panel & ~3) * NB;         // (from the first column of the panel's GROUP of four: zeros up to the panel's
  const int c0 = r0 + blockIdx.x * 128;     // own column j0, which the grouped products of the back-transformation read)
  if (c0 >= n) return;
  const double* Vp = Vall + (int64_t)panel * NB * ld;
  for (int e = tid; e < NB * NB; e += 256) Ts[e >> 6][e & 63] = Tall[(int64_t)panel * NB * NB + e];
  for (int e = tid; e < NB * 128; e += 256) {
    const int r = e >> 7, c = e & 127;
    Vs[r][c] = (c0 + c < n) ? Vp[(int64_t)r * ld + c0 + c] : 0.0;
  }
  __syncthreads();
  // f64 matrix cores (operand map: see wy_gram_kernel): wave w forms the 64 rows of columns 32 w .. 32 w + 31
  const int lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  f64x4 acc[4][2];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int kk = 0; kk < NB; kk += 4) {
    double av[4], bv[2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) av[rb] = Ts[16 * rb + l15][kk + l4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) bv[cb] = Vs[kk + l4][32 * wid + 16 * cb + l15];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rb], bv[cb], acc[rb][cb], 0, 0, 0);
  }
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = c0 + 32 * wid + 16 * cb + l15;
      if (col < n)
#pragma unroll
        for (int r = 0; r < 4; ++r) TVall[((int64_t)panel * NB + 16 * rb + l4 + 4 * r) * ld + col] = acc[rb][cb][r];
    }
}

// ---- four panels as ONE block reflector of 256 rows.  A panel's two products of the back-transformation have K = 64 or
// M = 64: 25 us each for a 5 / 10-us stream over Y, 64 times.  For consecutive panels a < b < c < d (applied d first)
//   Q_a Q_b Q_c Q_d Y = Y - [V_a; V_b; V_c; V_d]^T [W_a; W_b; W_c; W_d],   W_d = TV_d Y,  W_c = TV_c Y - S_cd W_d,
//   W_b = TV_b Y - S_bc W_c - S_bd W_d, ...   with S_xy = TV_x V_y^T (64 x 64),
// so with TV'_d = TV_d, TV'_c = TV_c - S_cd TV'_d, TV'_b = TV_b - S_bc TV'_c - S_bd TV'_d, ... the group is applied as
// Y -= V_g^T (TV'_g Y): two products of 256 rows / K = 256 per FOUR panels.  Three launches for all groups up front:
//   wy_cross      partial S_xy over column chunks (the six pairs of every group)
//   wy_cross_sum  their sums in chunk order
//   wy_merge      the substitution above on 64-column slabs of TV (in place)
constexpr int WYG = 4;        // panels per group
constexpr int XCH = 1024;     // columns per wy_cross workgroup
__device__ __constant__ const int wy_pair_x[6] = {0, 0, 0, 1, 1, 2};
__device__ __constant__ const int wy_pair_y[6] = {1, 2, 3, 2, 3, 3};

__global__ __launch_bounds__(256) void wy_cross_kernel(const double* __restrict__ TVall, const double* __restrict__ Vall,
                                                       int64_t ld, int n, int npanels, double* __restrict__ Spart,
                                                       int nchunks) {
  __shared__ double As[NB][65];
  __shared__ double Bs[NB][65];
  const int chunk = blockIdx.x, q = blockIdx.y, g = blockIdx.z, tid = threadIdx.x;
  const int px = WYG * g + wy_pair_x[q], py = WYG * g + wy_pair_y[q];
  if (py >= npanels) return;
  const int cbeg = WYG * NB * g + chunk * XCH, cend = min(n, cbeg + XCH);
  // (f64 matrix cores, operand map as in wy_gram_kernel: wave w forms rows 16 w .. of S_xy)
  const int lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  f64x4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) acc[cb] = f64x4{0.0, 0.0, 0.0, 0.0};
  const double* Ax = TVall + (int64_t)px * NB * ld;
  const double* By = Vall + (int64_t)py * NB * ld;
  for (int c0 = cbeg; c0 < cend; c0 += 64) {
    for (int e = tid; e < NB * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      const bool ok = c0 + c < cend;
      As[r][c] = ok ? Ax[(int64_t)r * ld + c0 + c] : 0.0;
      Bs[r][c] = ok ? By[(int64_t)r * ld + c0 + c] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < 64; kk += 4) {
      const double av = As[16 * wid + l15][kk + l4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[16 * cb + l15][kk + l4], acc[cb], 0, 0, 0);
    }
    __syncthreads();
  }
  double* out = Spart + (((int64_t)g * 6 + q) * nchunks + chunk) * NB * NB;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(16 * wid + l4 + 4 * r) * NB + 16 * cb + l15] = acc[cb][r];
}

__global__ __launch_bounds__(256) void wy_cross_sum_kernel(const double* __restrict__ Spart, int n, int nchunks,
                                                           double* __restrict__ S) {
  const int gq = blockIdx.x, g = gq / 6;
  const int used = (int)((n - WYG * NB * g + XCH - 1) / XCH);     // chunks of this group that lie inside the matrix
  for (int e = threadIdx.x; e < NB * NB; e += 256) {
    double t = 0.0;
    for (int c = 0; c < used; ++c) t += Spart[((int64_t)gq * nchunks + c) * NB * NB + e];   // fixed order
    S[(int64_t)gq * NB * NB + e] = t;
  }
}

// one 64-column slab of a group's TV rows: P_x <- TV_x - sum_{y > x} S_xy P_y, x = gsize - 2 .. 0 (P_y: the finished slabs)
__global__ __launch_bounds__(256) void wy_merge_kernel(double* __restrict__ TVall, int64_t ld, int n, int npanels,
                                                       const double* __restrict__ S) {
  extern __shared__ __attribute__((aligned(16))) double mg_smem[];
  double (*P)[NB][65] = reinterpret_cast<double (*)[NB][65]>(mg_smem);          // [3]: finished slabs of panels 1 .. 3
  double (*Ss)[65] = reinterpret_cast<double (*)[65]>(mg_smem + 3 * NB * 65);
  const int g = blockIdx.y, tid = threadIdx.x;
  const int gsize = min(WYG, npanels - WYG * g);
  const int c0 = WYG * NB * g + blockIdx.x * 64;
  if (c0 >= n || gsize < 2) return;
  const int lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int ncol = min(64, n - c0);
  // the last panel of the group is its own P
  {
    const double* src = TVall + (int64_t)(WYG * g + gsize - 1) * NB * ld + c0;
    for (int e = tid; e < NB * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      P[gsize - 2][r][c] = c < ncol ? src[(int64_t)r * ld + c] : 0.0;
    }
  }
  // (f64 matrix cores, operand map as in wy_gram_kernel: wave w holds rows 16 w .. of the slab, four 16 x 16 tiles)
  for (int x = gsize - 2; x >= 0; --x) {
    double* dst = TVall + (int64_t)(WYG * g + x) * NB * ld + c0;
    f64x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * cb + l15;
        acc[cb][r] = c < ncol ? dst[(int64_t)(16 * wid + l4 + 4 * r) * ld + c] : 0.0;
      }
    for (int y = x + 1; y < gsize; ++y) {
      int q = 0;
      for (int t = 0; t < 6; ++t) if (wy_pair_x[t] == x && wy_pair_y[t] == y) q = t;
      __syncthreads();                              // (the previous S block and, the first time, P are complete / consumed)
      for (int e = tid; e < NB * NB; e += 256) Ss[e >> 6][e & 63] = S[((int64_t)g * 6 + q) * NB * NB + e];
      __syncthreads();
      const double (*Py)[65] = P[y - 1];
#pragma unroll 4
      for (int kk = 0; kk < NB; kk += 4) {
        const double av = -Ss[16 * wid + l15][kk + l4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Py[kk + l4][16 * cb + l15], acc[cb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * cb + l15, row = 16 * wid + l4 + 4 * r;
        if (c < ncol) dst[(int64_t)row * ld + c] = acc[cb][r];
        if (x >= 1) P[x - 1][row][c] = acc[cb][r];
      }
  }
}

// T / ||T|| for the Sturm counts: ds = d / ||T||, es2 = (e / ||T||)^2 (bounds[3] = ||T|| from tridiag_bounds_kernel)
__global__ void scale_tridiag_kernel(const double* __restrict__ d, const double* __restrict__ e, int n,
                                     const double* __restrict__ bounds, double* __restrict__ ds,
                                     double* __restrict__ es2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const double tnorm = bounds[3];
  const double s = tnorm > 0.0 ? 1.0 / tnorm : 1.0;
  if (k < n) {
    ds[k] = d[k] * s;
    const double es = e[k] * s;
    es2[k] = es * es;
  }
}

__global__ void copy_pad_kernel(const double* __restrict__ A, int64_t lda, int n, double* __restrict__ B,
                                int64_t ldb) {
  const int64_t total = (int64_t)n * n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / n), jj = (int)(t % n);
    B[(int64_t)i * ldb + jj] = A[(int64_t)i * lda + jj];
  }
}

}  // namespace

struct TridiagPlan {
  int n;
  int64_t ld;      // leading dimension of the working copy and of the V / W panels
  int npanels;
  size_t off_A, off_V, off_W, off_col, off_p, off_part, off_refl, off_d, off_e, off_e2, off_ds, off_tau, off_bounds, off_lam;
  size_t off_u1, off_u2, off_u3, off_lm, off_sw, off_G, off_T, off_W1, off_W2, off_wraw, off_wraw2, off_part2, off_cbuf;
  size_t off_qv, off_px2, off_rowpart, off_colpart, off_gpart, off_tall, off_res, off_twd, off_twe, off_twl;
  int64_t ldp;     // leading dimension of the symmetric SYMV's partial-result arrays
  size_t total;
};

TridiagPlan tridiag_plan(int64_t n) {
  TridiagPlan p{};
  p.n = (int)n;
  // rows padded to whole SYMV tiles (the symmetric kernel reads TC-wide segments) once the matrix is
  // large enough for it to be used
  p.ld = (int64_t)align_up((size_t)n, n >= 512 ? TC : 8);
  if (n >= 512) {
    const char* pad = getenv("PTD_LD_PAD");
    p.ld += pad ? atoi(pad) / 2 * 2 : 64;  // rows rotate over the memory channels
  }
  p.ldp = (int64_t)align_up((size_t)n, TC);
  p.npanels = (int)ceil_div(n, NB);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
  // working copy of A; once the reduction is done it holds TV_p = T_p V_p of every panel (npanels * 64 rows)
  p.off_A = take((size_t)p.npanels * NB * p.ld * 8);
  p.off_V = take((size_t)p.npanels * NB * p.ld * 8);
  p.off_W = take((size_t)NB * p.ld * 8);
  p.off_col = take((size_t)(p.ldp + 8) * 8);
  p.off_p = take((size_t)(n + 8) * 8);
  p.off_part = take((size_t)(ceil_div(n, 64) + 8) * 8);
  p.off_refl = take(256);
  p.off_d = take((size_t)n * 8);
  p.off_e = take((size_t)n * 8);
  p.off_e2 = take((size_t)n * 8);
  p.off_ds = take((size_t)n * 8);
  p.off_tau = take((size_t)n * 8);
  p.off_bounds = take(64);
  p.off_lam = take((size_t)n * 8);
  // inverse-iteration factors alias the (dead by then) working copy of A when they fit; they
  // are n x n each, the same size as A
  p.off_u1 = take((size_t)n * n * 8);
  p.off_u2 = take((size_t)n * n * 8);
  p.off_u3 = take((size_t)n * n * 8);
  p.off_lm = take((size_t)n * n * 8);
  p.off_sw = take((size_t)n * n);
  p.off_G = take((size_t)NB * NB * 8);
  p.off_T = take((size_t)NB * NB * 8);
  p.off_W1 = take((size_t)NB * n * 8);
  p.off_W2 = take((size_t)NB * n * 8);
  p.off_wraw = take((size_t)(n + 8) * 8);
  p.off_wraw2 = take((size_t)(n + 8) * 8);
  p.off_part2 = take((size_t)(ceil_div(n, 64) + 8) * 8);
  p.off_cbuf = take((size_t)4 * NB * 8);
  p.off_qv = take((size_t)(n + 8) * 8);
  p.off_px2 = take((size_t)(ceil_div(n + 2 * NB, SROWS) + 8) * 8);
  p.off_rowpart = take((size_t)ceil_div(n, TC) * p.ldp * 8);
  p.off_colpart = take((size_t)ceil_div(n, TR) * p.ldp * 8);
  p.off_gpart = take((size_t)p.npanels * ceil_div(n, GCH) * NB * NB * 8);
  p.off_tall = take((size_t)p.npanels * NB * NB * 8);
  p.off_res = take(8192 + 4 * (size_t)RES_XS * 8);   // resident tail: control block, two pairs of exchange vectors
  p.off_twd = take((size_t)TW_MAXN * 8);             // twisted factorisations: d / |T|, e / |T| by lane chunks,
  p.off_twe = take((size_t)TW_MAXN * 8);
  p.off_twl = take((size_t)(n + 16) * 4);            // ... [0] count, [16 ..] list of the vectors left to inverse iteration
  p.total = o;
  return p;
}

// A (n x n, full symmetric) -> d, e, tau and the reflector panels in the workspace
// optional event timing of the SYMV launches (two events per column, summed by the caller)
struct SymvTimer {
  std::vector<hipEvent_t> ev;  // two per SAMPLED column: index 2 * (j / stride)
  int stride = 1;              // every stride-th column is timed (timing every launch slows the chain)
  int limit = INT32_MAX;       // columns from here on have no SYMV launch (resident tail)
  bool sampled(int j) const { return j < limit && j % stride == stride / 2; }
  hipEvent_t start(int j) const { return ev[2 * (size_t)(j / stride)]; }
  hipEvent_t stop(int j) const { return ev[2 * (size_t)(j / stride) + 1]; }
};

// ---- when the resident kernels may run: per-device facts and state
// They assume the whole chip: 256 workgroups of 150 KB LDS that must ALL be resident at once (one per CU) and, for the
// tail, 32 of them on one XCC.  That is a fact about the device (an unpartitioned MI355X: 256 CUs = 8 XCCs x 32; a CPX /
// CU-masked partition has fewer), about the kernel (the occupancy query must admit one such workgroup per CU) and
// about what else runs (another chain of this process on the device would share the CUs).  Queried once per device;
// the moving parts are per device as well: the caller's hint (ptd_set_concurrent_chains), the number of
// eigendecompositions this library has in flight there, and a back-off after a time-out.
constexpr int MAX_DEVICES = 64;
struct DeviceState {
  std::once_flag probed;
  int cus = 0;
  bool gfx950 = false;
  bool occupancy_ok = false;
  std::atomic<int> chains{1};     // ptd_set_concurrent_chains
  std::atomic<int> inflight{0};   // eigendecompositions of this process between entry and exit on this device
  std::atomic<int> skip{0};       // calls that stay on the blocked path after a time-out
  std::atomic<int> failures{0};
};
DeviceState g_dev[MAX_DEVICES];

DeviceState& device_state() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return g_dev[(dev < 0 || dev >= MAX_DEVICES) ? 0 : dev];
}

void probe_device(DeviceState& ds) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
  ds.cus = prop.multiProcessorCount;
  ds.gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
  // (the attribute belongs to this device's copy of the function; it is set again before every launch)
  const void* kernels[7] = {reinterpret_cast<const void*>(sytrd_resident_kernel<RES_WG, RES_MAX, false>),
                            reinterpret_cast<const void*>(sytrd_resident_kernel<RES_WG, RES_MID, false>),
                            reinterpret_cast<const void*>(sytrd_resident_kernel<RESG_WG, RESG_MAX, true>),
                            reinterpret_cast<const void*>(sytrd_resident3_kernel),
                            reinterpret_cast<const void*>(sytrd_resident4_kernel<R4_MAX>),
                            reinterpret_cast<const void*>(sytrd_resident4_kernel<R4B_MAX>),
                            reinterpret_cast<const void*>(sytrd_resident4_kernel<R4C_MAX>)};
  bool ok = true;
  for (const void* f : kernels) {
    int blocks = 0;
    const int threads = (f == kernels[4] || f == kernels[5] || f == kernels[6]) ? R4_T : RES_T;
    ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS) == hipSuccess &&
         hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, f, threads, RES_LDS) == hipSuccess && blocks >= 1;
  }
  ds.occupancy_ok = ok;
}

// PTD_SYTRD_RESIDENT: 0 off; 1 the one-XCD tail only; 2 test hook, see sytrd_f64; 3 whole-chip kernel from a trailing
// order of 2048, then the one-XCD tail; 4 the half-row whole-chip kernel from 3072 in front of those; 5 (default) the
// four-wave quarter-row kernel from 3328 in front of that; 6 the same kernel with 14 rows from 3584 in front; 7
// (default) and with 15 rows from 3840 in front of that
int resident_mode() {
  const char* env = getenv("PTD_SYTRD_RESIDENT");
  return env ? atoi(env) : 7;
}
// first column of the resident part: the first panel boundary with a trailing order the kernels take; n itself
// (= nothing resident) when they may not run here and now
int resident_start(int n) {
  const int mode = resident_mode();
  if (mode == 0 || n < 128) return n;
  DeviceState& ds = device_state();
  std::call_once(ds.probed, probe_device, std::ref(ds));
  // test hook, read at every call: PTD_SYTRD_FAKE_CUS=<count> stands for a device with that many CUs
  const char* fake = getenv("PTD_SYTRD_FAKE_CUS");
  const int cus = fake ? atoi(fake) : ds.cus;
  if (cus != RESG_WG || !ds.gfx950 || !ds.occupancy_ok) return n;
  // several chains at once -- announced (ptd_set_concurrent_chains) or seen (another call of this process is between
  // entry and exit on this device): the kernels would hold XCC 0, or the chip, for milliseconds while the other
  // chains' launches queue behind them (2-block Llama stack, three chains: 1.50 s with, 1.38 s without), and two of
  // them at once could never all be resident
  if (ds.chains.load(std::memory_order_relaxed) > 1 || ds.inflight.load(std::memory_order_relaxed) > 1) return n;
  // after a time-out (someone else's work held CUs): blocked path for a while, doubling per failure
  int left = ds.skip.load(std::memory_order_relaxed);
  while (left > 0)
    if (ds.skip.compare_exchange_weak(left, left - 1, std::memory_order_relaxed)) return n;
  const int cap = n <= RES_MAX ? RES_MAX : (mode >= 7 ? R4C_MAX : (mode == 6 ? R4B_MAX : (mode == 5 ? R4_MAX : (mode == 4 ? R3_MAX : (mode == 3 ? RESG_MAX : RES_MAX)))));
  return n <= cap ? 0 : (int)align_up((size_t)(n - cap), NB);
}
void resident_failed(int status) {
  if (status == 3) return;   // the test hook is not evidence about the device
  DeviceState& ds = device_state();
  const int f = std::min(ds.failures.fetch_add(1) + 1, 14);
  ds.skip.store(32 << f, std::memory_order_relaxed);
}
struct InflightGuard {
  DeviceState& ds;
  InflightGuard() : ds(device_state()) { ds.inflight.fetch_add(1); }
  ~InflightGuard() { ds.inflight.fetch_sub(1); }
};

template <typename T>
static inline T* batch_host(T* p, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(p) + off); }

int* resident_status(const TridiagPlan& p, char* base) { return &reinterpret_cast<ResCtl*>(base + p.off_res)->fail; }

// count > 1: `count` matrices of order n in workspaces bstride bytes apart, one launch per kernel and column for all of
// them (blocked path to the end: see eigh_tridiag_batched)
int sytrd_f64(const TridiagPlan& p, char* base, SymvTimer* timer, bool resident, hipStream_t st, int count = 1,
              size_t bstride = 0) {
  const int n = p.n;
  const unsigned nb = (unsigned)std::max(count, 1);
  if (nb > 1) resident = false;
  if (nb == 1) bstride = 0;
  const int64_t ld = p.ld;
  double* Aw = reinterpret_cast<double*>(base + p.off_A);
  double* Vall = reinterpret_cast<double*>(base + p.off_V);
  double* Wp = reinterpret_cast<double*>(base + p.off_W);
  double* colbuf = reinterpret_cast<double*>(base + p.off_col);
  double* pbuf = reinterpret_cast<double*>(base + p.off_p);
  double* d = reinterpret_cast<double*>(base + p.off_d);
  double* e = reinterpret_cast<double*>(base + p.off_e);
  double* taus = reinterpret_cast<double*>(base + p.off_tau);

  double* wr[2] = {reinterpret_cast<double*>(base + p.off_wraw), reinterpret_cast<double*>(base + p.off_wraw2)};
  double* partial2 = reinterpret_cast<double*>(base + p.off_part2);
  double* cbuf = reinterpret_cast<double*>(base + p.off_cbuf);
  double* sd = pbuf;                                             // unscaled SYMV products
  double* qv = reinterpret_cast<double*>(base + p.off_qv);       // A[r][j+1]
  double* px2 = reinterpret_cast<double*>(base + p.off_px2);
  ColState* cs = reinterpret_cast<ColState*>(base + p.off_refl);
  // symmetric (lower-triangle) SYMV for trailing orders >= sym_min; PTD_SYMV=full switches it off
  SymPart sp{reinterpret_cast<double*>(base + p.off_rowpart), reinterpret_cast<double*>(base + p.off_colpart), p.ldp,
             (int)ceil_div(n, TR), 0};
  const char* symv_env = getenv("PTD_SYMV");
  const char* symv_min_env = getenv("PTD_SYMV_MIN");
  const int sym_min = (symv_env && !strcmp(symv_env, "full")) ? INT32_MAX
                      : std::max(512, symv_min_env ? atoi(symv_min_env) : 1024);
  bool prev_sym = false;  // how the open column's SYMV was computed
  for (unsigned b = 0; b < nb; ++b) {
    const size_t bo = b * bstride;
    PTD_CHECK_HIP(hipMemsetAsync(batch_host(Vall, bo), 0, (size_t)p.npanels * NB * ld * 8, st));
    PTD_CHECK_HIP(hipMemsetAsync(batch_host(taus, bo), 0, (size_t)n * 8, st));
    PTD_CHECK_HIP(hipMemsetAsync(batch_host(e, bo), 0, (size_t)n * 8, st));
    PTD_CHECK_HIP(hipMemsetAsync(batch_host(colbuf, bo), 0, (size_t)(p.ldp + 8) * 8, st));
    // W panel: entries below a row's first written column are multiplied by zero but must be finite;
    // later panels find the previous panel's (finite) values there
    PTD_CHECK_HIP(hipMemsetAsync(batch_host(Wp, bo), 0, (size_t)NB * ld * 8, st));
    PTD_CHECK_HIP(hipMemsetAsync(batch_host(wr[0], bo), 0, (size_t)(n + 8) * 8, st));
    PTD_CHECK_HIP(hipMemsetAsync(batch_host(wr[1], bo), 0, (size_t)(n + 8) * 8, st));
  }
  bool colbuf_ready = false;
  const int t_res = resident ? resident_start(n) : n;
  if (timer) timer->limit = t_res;
  ResCtl* rctl = reinterpret_cast<ResCtl*>(base + p.off_res);
  for (unsigned b = 0; b < nb; ++b) PTD_CHECK_HIP(hipMemsetAsync(batch_host(rctl, b * bstride), 0, sizeof(ResCtl), st));
  for (int pn = 0; pn < p.npanels; ++pn) {
    const int j0 = pn * NB;
    if (j0 == t_res && n - j0 >= 2) {
      // the rest of the reduction in one launch, the trailing block resident in the LDS of one XCD
      double* X = reinterpret_cast<double*>(base + p.off_res + 8192);
      // (set at every call: the attribute belongs to the current device's copy of the function)
      const bool attr =
          hipFuncSetAttribute(reinterpret_cast<const void*>(sytrd_resident_kernel<RES_WG, RES_MAX, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS) == hipSuccess &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(sytrd_resident_kernel<RES_WG, RES_MID, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS) == hipSuccess &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(sytrd_resident_kernel<RESG_WG, RESG_MAX, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS) == hipSuccess &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(sytrd_resident3_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS) == hipSuccess &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(sytrd_resident4_kernel<R4_MAX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS) == hipSuccess &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(sytrd_resident4_kernel<R4B_MAX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS) == hipSuccess &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(sytrd_resident4_kernel<R4C_MAX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS) == hipSuccess;
      PTD_REQUIRE(attr, "sytrd_f64: cannot reserve the LDS of the resident tail");
      static std::atomic<unsigned long long> calls{0};
      static_assert(RESG_MAX < (1 << 16), "sequence numbers of one launch: epoch .. epoch + m");
      int t1 = j0, t2 = j0, t3 = j0, t4 = j0, t5 = j0;
      if (n - j0 > R4B_MAX) {
        // 15 rows a workgroup: down to a trailing order of 3584
        const unsigned long long epoch = (calls.fetch_add(1) + 1) << 16;
        t5 = n - R4B_MAX;
        hipLaunchKernelGGL((sytrd_resident4_kernel<R4C_MAX>), dim3(RESG_WG), dim3(R4_T), RES_LDS, st, Aw, ld, n, j0,
                           t5 - j0, Vall, taus, d, e, rctl, X, epoch);
      }
      t4 = t5;
      if (n - t5 > R4_MAX) {
        // quarter rows on four waves, 14 rows a workgroup: down to a trailing order of 3328
        const unsigned long long epoch = (calls.fetch_add(1) + 1) << 16;
        t4 = n - R4_MAX;
        hipLaunchKernelGGL((sytrd_resident4_kernel<R4B_MAX>), dim3(RESG_WG), dim3(R4_T), RES_LDS, st, Aw, ld, n, t5,
                           t4 - t5, Vall, taus, d, e, rctl, X, epoch);
      }
      t3 = t4;
      if (n - t4 > R3_MAX) {
        // 13 rows a workgroup: down to a trailing order of 3072
        const unsigned long long epoch = (calls.fetch_add(1) + 1) << 16;
        t3 = n - R3_MAX;
        hipLaunchKernelGGL((sytrd_resident4_kernel<R4_MAX>), dim3(RESG_WG), dim3(R4_T), RES_LDS, st, Aw, ld, n, t4,
                           t3 - t4, Vall, taus, d, e, rctl, X, epoch);
      }
      t2 = t3;
      if (n - t3 > RESG_MAX) {
        // half rows: down to a trailing order of 2048
        const unsigned long long epoch = (calls.fetch_add(1) + 1) << 16;
        t2 = n - RESG_MAX;
        hipLaunchKernelGGL(sytrd_resident3_kernel, dim3(RESG_WG), dim3(RES_T), RES_LDS, st, Aw, ld, n, t3, t2 - t3, Vall,
                           taus, d, e, rctl, X, epoch);
      }
      const int one_xcd = resident_mode() == 1 ? RES_MAX : RES_MID;   // (mode 1: the 768 tail alone, as it was measured)
      if (n - t2 > one_xcd) {
        // every CU first: down to a trailing order of 1024
        const unsigned long long epoch = (calls.fetch_add(1) + 1) << 16;   // > any sequence number of an earlier launch
        t1 = n - one_xcd;
        hipLaunchKernelGGL((sytrd_resident_kernel<RESG_WG, RESG_MAX, true>), dim3(RESG_WG), dim3(RES_T), RES_LDS, st, Aw, ld,
                           n, t2, t1 - t2, Vall, taus, d, e, rctl, X, epoch);
      }
      if (n - t1 > RES_MAX) {
        // one XCD from 1024 down to 768 with four rows a wave (4.9 us a column against 9 on every CU; the same kernel
        // is slower than the three-row form below 768 -- 4.6 against 3.9 ms at n = 768 -- so it hands over there)
        const unsigned long long epoch = (calls.fetch_add(1) + 1) << 16;
        const int t15 = n - RES_MAX;
        hipLaunchKernelGGL((sytrd_resident_kernel<RES_WG, RES_MID, false>), dim3(8 * RES_WG), dim3(RES_T), RES_LDS, st, Aw,
                           ld, n, t1, t15 - t1, Vall, taus, d, e, rctl, X, epoch);
        PTD_CHECK_HIP(hipMemsetAsync(&rctl->reg, 0, sizeof(unsigned), st));   // the next launch registers afresh
        t1 = t15;
      }
      const unsigned long long epoch = (calls.fetch_add(1) + 1) << 16;
      hipLaunchKernelGGL((sytrd_resident_kernel<RES_WG, RES_MAX, false>), dim3(8 * RES_WG), dim3(RES_T), RES_LDS, st, Aw, ld,
                         n, t1, n - t1 - 1, Vall, taus, d, e, rctl, X, epoch);
      // test hook: PTD_SYTRD_RESIDENT=2 reports the tail as failed, so that the caller's repeat on the blocked path runs
      if (resident_mode() == 2)
        PTD_CHECK_HIP(hipMemsetAsync(&rctl->fail, 3, 1, st));
      break;
    }
    const int cols = std::min(NB, n - j0);
    double* Vp = Vall + (size_t)pn * NB * ld;
    int nparts2 = 0, npx2 = 0;
    bool open = false;  // a column whose SYMV ran and whose reflector is not finished yet
    for (int i = 0; i < cols; ++i) {
      const int j = j0 + i;
      if (i == 0) {
        if (!colbuf_ready)  // the previous panel's trailing update has already left row j in colbuf
          hipLaunchKernelGGL(sytrd_colinit_kernel, dim3((unsigned)ceil_div(n - j, 256), nb), dim3(256), 0, st, Aw, ld, n,
                             j, colbuf, bstride);
        colbuf_ready = false;
      } else {
        const int blocks = (int)ceil_div(n - j, 64);
        SymPart spa = sp;
        spa.enabled = prev_sym;
        hipLaunchKernelGGL(sytrd_alpha_kernel, dim3(blocks, nb), dim3(256), 0, st, Aw, ld, n, j, i, 1, Vp, Wp, ld,
                           wr[(i & 1)], wr[(i - 1) & 1], colbuf, sd, qv, cbuf, px2, npx2, cs, partial2, d, e, taus,
                           spa, bstride);
        nparts2 = blocks;
      }
      const int m = n - j - 1;
      if (m >= sym_min && n >= 512) {
        const int U = (int)ceil_div(n, TC) - (j + 1) / TC;  // groups of tile rows from the one holding row j+1
        const int ntiles = TQ / 2 * U * (U + 1);
        const int nextra = (int)ceil_div(2 * i, SROWS);
        npx2 = (int)ceil_div(m, 256);
        const dim3 grid((unsigned)(ntiles + nextra + npx2), nb);
        const bool timed = timer && timer->sampled(j);
        hipEvent_t ev0 = timed ? timer->start(j) : nullptr, ev1 = timed ? timer->stop(j) : nullptr;
        hipExtLaunchKernelGGL(sytrd_symv2_kernel, grid, dim3(256), 0, st, ev0, ev1, 0, Aw, ld, n, j, i, colbuf, Vp, Wp,
                              ld, wr[(i + 1) & 1], partial2, nparts2, taus, sp, ntiles, nextra, qv, cbuf, px2, cs, bstride);
        prev_sym = true;
        open = true;
      } else if (m > 0) {
        prev_sym = false;
        const int rows = m + 2 * i;
        npx2 = (int)ceil_div(rows, SROWS);
        if (timer && timer->sampled(j)) {
          // start / stop events attached to the dispatch itself: the same begin / end timestamps
          // of the kernel's completion signal that rocprofv3 reports
          hipExtLaunchKernelGGL(sytrd_symv_kernel, dim3(npx2, nb), dim3(256), 0, st, timer->start(j),
                                timer->stop(j), 0, Aw, ld, n, j, i, colbuf, Vp, Wp, ld,
                                wr[(i + 1) & 1], partial2, nparts2, taus, sd, qv, cbuf, px2, cs, bstride);
        } else {
          hipLaunchKernelGGL(sytrd_symv_kernel, dim3(npx2, nb), dim3(256), 0, st, Aw, ld, n, j, i, colbuf, Vp, Wp, ld,
                             wr[(i + 1) & 1], partial2, nparts2, taus, sd, qv, cbuf, px2, cs, bstride);
        }
        open = true;
      } else {
        hipLaunchKernelGGL(sytrd_last_kernel, dim3(1, nb), dim3(64), 0, st, n, i, colbuf, partial2, nparts2, taus, d, bstride);
        open = false;
      }
    }
    const int t0 = j0 + cols;  // first row / column after the panel
    if (open) {
      // finish the panel's last column (no next column to form), then finalise its w
      const int blocks = (int)ceil_div(n - t0, 64);
      SymPart spa = sp;
      spa.enabled = prev_sym;
      hipLaunchKernelGGL(sytrd_alpha_kernel, dim3(std::max(blocks, 1), nb), dim3(256), 0, st, Aw, ld, n, t0, cols, 0, Vp,
                         Wp, ld, wr[(cols & 1)], wr[(cols - 1) & 1], colbuf, sd, qv, cbuf, px2, npx2, cs, partial2,
                         d, e, taus, spa, bstride);
      const int mt = n - t0;
      if (mt > 0) {
        hipLaunchKernelGGL(sytrd_wfix_kernel, dim3((unsigned)ceil_div(mt, 256), nb), dim3(256), 0, st, n, t0, cols - 1,
                           t0 - 1, Vp, Wp, ld, wr[(cols - 1) & 1], partial2, std::max(blocks, 1), taus, bstride);
        // trailing update A[T0:, T0:] -= V W^T + W V^T
        double* At = Aw + (int64_t)t0 * ld + t0;
        // ... and the updated row t0 goes straight into colbuf: the next panel's first column (colinit)
        const int rc = gemm_f64_pair(Vp + t0, Wp + t0, Wp + t0, Vp + t0, 1, ld, ld, 1, At, ld, mt, mt, cols, -1.0,
                                     colbuf + t0, st, (int)nb, (int64_t)(bstride / 8));
        colbuf_ready = true;
        if (rc != PTD_OK) return rc;
      }
    }
  }
  PTD_CHECK_LAUNCH("sytrd_f64");
  return PTD_OK;
}

// eigenvalues first .. n-1 of T (ascending order) into lam[first ..]; lam[0 .. first) becomes NaN
int tridiag_eigenvalues(const TridiagPlan& p, char* base, int first, hipStream_t st) {
  const int n = p.n;
  double* d = reinterpret_cast<double*>(base + p.off_d);
  double* e = reinterpret_cast<double*>(base + p.off_e);
  double* e2 = reinterpret_cast<double*>(base + p.off_e2);
  double* bounds = reinterpret_cast<double*>(base + p.off_bounds);
  double* lam = reinterpret_cast<double*>(base + p.off_lam);
  double* ds = reinterpret_cast<double*>(base + p.off_ds);
  hipLaunchKernelGGL(tridiag_bounds_kernel, dim3(1), dim3(1024), 0, st, d, e, n, bounds);
  hipLaunchKernelGGL(scale_tridiag_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, d, e, n, bounds, ds, e2);
  if (first > 0) PTD_CHECK_HIP(hipMemsetAsync(lam, 0xFF, (size_t)first * 8, st));  // all-ones = NaN
  if (n <= 4096)
    hipLaunchKernelGGL((tridiag_bisect_kernel<true>), dim3((unsigned)ceil_div(n - first, 4)), dim3(256), (size_t)n * 16, st,
                       ds, e2, n, bounds, lam, first);
  else
    hipLaunchKernelGGL((tridiag_bisect_kernel<false>), dim3((unsigned)ceil_div(n - first, 4)), dim3(256), 0, st, ds, e2,
                       n, bounds, lam, first);
  PTD_CHECK_LAUNCH("tridiag_eigenvalues");
  return PTD_OK;
}


// eigenvectors of T for all eigenvalues into Y (= evecs, [n][ldv]), then Y <- Q Y
namespace {
// out = slab_0 + slab_1 + ... (index order: a deterministic K split); slab q at S + q * total
__global__ void sum_slabs_kernel(double* __restrict__ out, const double* __restrict__ S, int64_t total, int ns) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    double acc = S[i];
    for (int q = 1; q < ns; ++q) acc += S[(int64_t)q * total + i];
    out[i] = acc;
  }
}

// G (mp x mp) <- identity outside its leading nvec x nvec block (which the Gram product fills)
__global__ void gram_pad_kernel(double* __restrict__ G, int mp, int nvec) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)mp * mp; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / mp), c = (int)(e % mp);
    if (r >= nvec || c >= nvec) G[e] = r == c ? 1.0 : 0.0;
  }
}
}  // namespace

// long_chains: more than 48 consecutive eigenvalues closer than ortol |T| (a dominant outlier above a dense bulk makes
// every spacing of the bulk small RELATIVE TO |T|): the chain-by-chain modified Gram-Schmidt would take O(L^2) dependent
// passes, so ALL vectors are orthonormalised at once by one Cholesky-QR pass instead (Gram matrix, Cholesky sweep, one
// product: ~2 ms at k = 1024).  The computed vectors are orthogonal to ~eps |T| / gap <= 1e-6 already, so the Gram
// matrix is I + O(1e-6) and one pass is exact to rounding; the triangular factor mixes a vector only with its
// predecessors in eigenvalue order, by those same tiny amounts.
int tridiag_vectors_and_backtransform(const TridiagPlan& p, char* base, int nvec, double* Y, int64_t ldy,
                                      double ortol, int niter, hipStream_t st, bool long_chains = false) {
  const int n = p.n;
  const int64_t ld = p.ld;
  double* d = reinterpret_cast<double*>(base + p.off_d);
  double* e = reinterpret_cast<double*>(base + p.off_e);
  double* lam = reinterpret_cast<double*>(base + p.off_lam);
  double* bounds = reinterpret_cast<double*>(base + p.off_bounds);
  double* taus = reinterpret_cast<double*>(base + p.off_tau);
  double* Vall = reinterpret_cast<double*>(base + p.off_V);
  double* W2 = reinterpret_cast<double*>(base + p.off_W2);
  InvitWs ws;
  ws.U1i = reinterpret_cast<double*>(base + p.off_u1);
  ws.U2 = reinterpret_cast<double*>(base + p.off_u2);
  ws.U3 = reinterpret_cast<double*>(base + p.off_u3);
  ws.Lm = reinterpret_cast<double*>(base + p.off_lm);
  ws.sw = reinterpret_cast<unsigned char*>(base + p.off_sw);
  const double* lamk = lam + (n - nvec);  // the nvec largest eigenvalues, ascending
  // (the recurrences are sequential in the row index and latency bound -- one round trip per 12 rows --, every wave
  // runs the same chain, so fewer lanes per wave / more waves do not shorten the launch: measured 64 = 32 = 16 = 8)
  static const bool no_twist = getenv("PTD_EIGH_TWIST") && atoi(getenv("PTD_EIGH_TWIST")) == 0;
  if (n <= TW_MAXN && n >= 64 && !no_twist) {
    double* dsT = reinterpret_cast<double*>(base + p.off_twd);
    double* esT = reinterpret_cast<double*>(base + p.off_twe);
    int* cnt = reinterpret_cast<int*>(base + p.off_twl);
    int* list = cnt + 16;
    const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(tridiag_twist_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess;
    PTD_REQUIRE(attr, "tridiag_twist: cannot reserve LDS");
    hipLaunchKernelGGL(tw_prepare_kernel, dim3(TW_MAXN / 256), dim3(256), 0, st, d, e, n, bounds, dsT, esT, cnt);
    const char* brk = getenv("PTD_TWIST_TEST_BREAK");
    hipLaunchKernelGGL(tridiag_twist_kernel, dim3((unsigned)nvec), dim3(64), 65536, st, dsT, esT, n, lamk, bounds, nvec,
                       Y, ldy, ws.Lm, list, cnt, nvec < n ? 1 : 0, brk ? atoi(brk) : 0);
    // the vectors whose residual was refused (normally a handful): inverse iteration, one WAVE per list entry for the
    // first wave_max entries (both launches exit at once on an empty list); a list longer than that -- a spectrum that
    // is one dense cluster -- leaves the rest to the one-vector-per-lane kernel
    static const bool no_wave = getenv("PTD_EIGH_INVIT_WAVE") && atoi(getenv("PTD_EIGH_INVIT_WAVE")) == 0;
    const int wave_max = no_wave ? 0 : std::max(0, std::min(nvec - 1, 2048));
    if (wave_max > 0)
      hipLaunchKernelGGL(tridiag_invit_wave_kernel, dim3((unsigned)wave_max), dim3(64), 0, st, d, e, n, lamk, bounds,
                         nvec, ws, Y, ldy, niter, list, cnt, wave_max);
    if (nvec > wave_max)
      hipLaunchKernelGGL(tridiag_invit_kernel, dim3((unsigned)ceil_div(nvec - wave_max, 64)), dim3(64), 0, st, d, e, n,
                         lamk, bounds, nvec, ws, Y, ldy, niter, list, cnt, wave_max);
    if (getenv("PTD_JACOBI_DEBUG")) {
      int h = 0;
      PTD_CHECK_HIP(hipMemcpyAsync(&h, cnt, 4, hipMemcpyDeviceToHost, st));
      PTD_CHECK_HIP(hipStreamSynchronize(st));
      fprintf(stderr, "[eigh_tridiag] twisted factorisations: %d of %d vectors left to inverse iteration\n", h, nvec);
    }
  } else {
    hipLaunchKernelGGL(tridiag_invit_kernel, dim3((unsigned)ceil_div(nvec, 64)), dim3(64), 0, st, d, e, n, lamk, bounds,
                       nvec, ws, Y, ldy, niter, (const int*)nullptr, (const int*)nullptr, 0);
  }
  hipLaunchKernelGGL(invit_scale_kernel, dim3((unsigned)ceil_div(nvec, 256), 256), dim3(256), 0, st, Y, ldy, n, nvec,
                     ws.Lm);
  if (ortol > 0.0 && !long_chains)
    hipLaunchKernelGGL(tridiag_chain_mgs_kernel, dim3((unsigned)nvec), dim3(256), 0, st, lamk, n, nvec, bounds, ortol,
                       Y, ldy);
  PTD_CHECK_LAUNCH("tridiag_invit");
  if (ortol > 0.0 && long_chains) {
    // (the inverse-iteration buffers u1 .. lm are dead behind the kernels above; the working copy of A is free until
    // the back-transformation fills it with T_p V_p)
    const int64_t mp = (int64_t)align_up((size_t)nvec, 64);
    double* G = reinterpret_cast<double*>(base + p.off_u1);
    double* Wt = G + mp * mp;
    char* cws = reinterpret_cast<char*>(Wt + mp * mp);
    const size_t cbytes = chol_inverse_workspace_bytes(mp);
    // (the working copy of A is free here: the new block goes there)
    char* after = cws + align_up(cbytes, 256);
    double* Ynew = reinterpret_cast<double*>(base + p.off_A);
    const size_t used = (size_t)(after - (base + p.off_u1));
    if (used > (size_t)(p.off_sw - p.off_u1) || (size_t)n * nvec * 8 > (size_t)(p.off_V - p.off_A)) {
      set_error("eigh_tridiag: no room to orthonormalise %d clustered vectors", nvec);
      return PTD_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(gram_pad_kernel, dim3(1024), dim3(256), 0, st, G, (int)mp, nvec);
    int rc = gemm_f64(Y, 1, ldy, Y, ldy, 1, G, mp, nvec, nvec, n, 1.0, false, 1, st);
    if (rc != PTD_OK) return rc;
    rc = chol_inverse(G, mp, Wt, cws, cbytes, st);
    if (rc != PTD_OK) return rc == PTD_ERR_WORKSPACE ? PTD_ERR_UNSUPPORTED : rc;
    rc = gemm_f64(Y, ldy, 1, Wt, mp, 1, Ynew, nvec, n, nvec, nvec, 1.0, false, 1, st);
    if (rc != PTD_OK) return rc;
    PTD_CHECK_HIP(hipMemcpy2DAsync(Y, (size_t)ldy * 8, Ynew, (size_t)nvec * 8, (size_t)nvec * 8, (size_t)n,
                                   hipMemcpyDeviceToDevice, st));
  }
  // Y <- Q_0 Q_1 ... Q_last Y : apply the panels' block reflectors from the last to the first,
  // Q_p Y = Y - V_p^T (TV_p Y) with TV_p = T_p V_p formed for all panels up front
  double* TVall = reinterpret_cast<double*>(base + p.off_A);  // the working copy of A is dead by now
  {
    double* Gpart = reinterpret_cast<double*>(base + p.off_gpart);
    double* Tall = reinterpret_cast<double*>(base + p.off_tall);
    const int nchunks = (int)ceil_div(n, GCH);
    hipLaunchKernelGGL(wy_gram_kernel, dim3((unsigned)nchunks, (unsigned)p.npanels), dim3(256), 0, st, Vall, ld, n,
                       Gpart, nchunks);
    hipLaunchKernelGGL(wy_tfactor_kernel, dim3((unsigned)p.npanels), dim3(256), 0, st, Gpart, nchunks, taus, n, Tall);
    hipLaunchKernelGGL(wy_tv_kernel, dim3((unsigned)ceil_div(n, 128), (unsigned)p.npanels), dim3(256), 0, st, Vall, ld,
                       n, Tall, TVall);
    PTD_CHECK_LAUNCH("wy factors");
  }
  // groups of four panels as one block reflector (see wy_cross_kernel); PTD_EIGH_WY_GROUP=0: panel by panel
  static const bool no_group = getenv("PTD_EIGH_WY_GROUP") && atoi(getenv("PTD_EIGH_WY_GROUP")) == 0;
  const int ngroups = (int)ceil_div(p.npanels, WYG);
  const int xchunks = (int)ceil_div(n, XCH);
  // (the factors of the inverse iteration are dead by now: the group's W, the partial and the summed S live there)
  double* Wg = reinterpret_cast<double*>(base + p.off_u1);
  double* Spart = Wg + (size_t)WYG * NB * nvec;
  double* Sall = Spart + (size_t)ngroups * 6 * xchunks * NB * NB;
  const bool grouped = !no_group && n >= 2048 &&      // (below, the three extra launches cost what the fewer products save)
                       ((size_t)WYG * NB * nvec + (size_t)ngroups * 6 * (xchunks + 1) * NB * NB) * 8 <= (size_t)(p.off_u2 - p.off_u1);
  if (grouped) {
    constexpr int MERGE_LDS = (3 * NB * 65 + NB * 65) * 8;
    const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(wy_merge_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, MERGE_LDS) == hipSuccess;
    PTD_REQUIRE(attr, "tridiag_backtransform: cannot reserve LDS");
    hipLaunchKernelGGL(wy_cross_kernel, dim3((unsigned)xchunks, 6, (unsigned)ngroups), dim3(256), 0, st, TVall, Vall, ld, n,
                       p.npanels, Spart, xchunks);
    hipLaunchKernelGGL(wy_cross_sum_kernel, dim3((unsigned)(6 * ngroups)), dim3(256), 0, st, Spart, n, xchunks, Sall);
    hipLaunchKernelGGL(wy_merge_kernel, dim3((unsigned)ceil_div(n, 64), (unsigned)ngroups), dim3(256), MERGE_LDS, st, TVall,
                       ld, n, p.npanels, Sall);
    PTD_CHECK_LAUNCH("wy groups");
  }
  const int step = grouped ? WYG : 1;
  for (int gi = (p.npanels - 1) / step; gi >= 0; --gi) {
    const int pn = gi * step;                          // first panel of the group
    const int j0 = pn * NB;
    const int rows = std::min(step * NB, n - j0);      // reflectors of the group
    if (n - (j0 + 1) <= 0) continue;          // (j0 + 1: the first row any reflector of this group touches)
    // the products start at the 128-row boundary at or below j0: V_p and T_p V_p are zero in the columns before a panel's
    // first reflector (Vall is cleared before the reduction, wy_tv writes zeros from the group's first column), so the
    // extra rows add nothing -- and the operands are 16-byte aligned and the row count a multiple of 128 where n is, which
    // is what the LDS-DMA kernel asks for (from row j0 + 1 both products ran on the generic 64 x 64 kernel)
    const int r0 = j0 & ~127;
    const int mr = n - r0;
    const double* Vp = Vall + (size_t)pn * NB * ld;
    const double* TVp = TVall + (size_t)pn * NB * ld;
    double* Wp2 = grouped ? Wg : W2;
    // W = TV_g Y  (rows x nvec): the K range (up to n rows) is cut into slabs -- 16 for a 64-row panel, 8 for a group's
    // 256 rows: 4 x the tiles per range -- that are summed IN INDEX ORDER (round 5 added the ranges with f64 atomics:
    // the eigenvectors then differed in their last bits from run to run).  The slabs live in the second inverse-iteration
    // buffer (n x n doubles, dead by now); the split shrinks where they would not fit.
    int ksplit = grouped ? 8 : 16;
    while (ksplit > 1 && (size_t)ksplit * rows * nvec > (size_t)n * n) ksplit /= 2;
    int rc;
    if (ksplit > 1) {
      double* slabs = reinterpret_cast<double*>(base + p.off_u2);
      int nslabs = 0;
      rc = gemm_f64_slabs(TVp + r0, ld, 1, Y + (int64_t)r0 * ldy, ldy, 1, slabs, nvec, (int64_t)rows * nvec, rows, nvec, mr,
                          1.0, ksplit, &nslabs, false, st);
      if (rc != PTD_OK) return rc;
      const int64_t total = (int64_t)rows * nvec;
      hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 4096)), dim3(256), 0, st,
                         Wp2, slabs, total, nslabs);
    } else {
      rc = gemm_f64(TVp + r0, ld, 1, Y + (int64_t)r0 * ldy, ldy, 1, Wp2, nvec, rows, nvec, mr, 1.0, false, 1, st);
    }
    if (rc != PTD_OK) return rc;
    // Y[r0:, :] -= V_g^T W
    rc = gemm_f64(Vp + r0, 1, ld, Wp2, nvec, 1, Y + (int64_t)r0 * ldy, ldy, mr, nvec, rows, -1.0, true, 1, st);
    if (rc != PTD_OK) return rc;
  }
  PTD_CHECK_LAUNCH("tridiag_backtransform");
  return PTD_OK;
}

// ---- a cluster at the BOTTOM of the spectrum that reaches into the requested eigenvalues
// A rank-deficient covariance with the drivers' damping (C + d I, d = 0.01 mean diag) has n - r eigenvalues equal to d
// to working precision; when the request asks for more than r eigenvectors, k - r of them belong to that cluster.
// Inverse iteration cannot separate them -- and does not have to: EVERY unit vector orthogonal to the eigenvectors of
// the eigenvalues above the cluster is an eigenvector of it (the cluster holds all the rest of the spectrum), with a
// residual of the cluster's width.  So the r vectors above are computed as usual and the block is completed with an
// orthonormal basis of their complement: random columns, two projections against the computed vectors, two
// Cholesky-QR passes (all f64 matrix-core products).  Before this the whole matrix went to the Jacobi solver:
// ~0.5 s at n = 4096 against ~65 ms.
namespace {
__global__ void random_fill_kernel(double* __restrict__ R, int64_t total, unsigned long long seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long z = seed + (unsigned long long)i * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    R[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
}
}  // namespace

// V [n][ldv]: columns [m, k) hold orthonormal vectors; columns [0, m) receive orthonormal vectors orthogonal to them.
// PTD_ERR_UNSUPPORTED when the workspace is too small or a Gram matrix is not positive definite.
static int complete_basis(double* V, int64_t ldv, int64_t n, int64_t m, int64_t k, char* ws, size_t ws_bytes,
                          hipStream_t st) {
  const int64_t kz = k - m;
  const int64_t mp = (int64_t)align_up((size_t)m, 64);  // leading dimension and order of the padded Gram matrix
  if (m > n - kz) return PTD_ERR_UNSUPPORTED;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
  const size_t off_r0 = take((size_t)n * mp * 8), off_r1 = take((size_t)n * mp * 8);
  const size_t off_p = take((size_t)std::max<int64_t>(kz, 1) * mp * 8);
  const size_t off_g = take((size_t)mp * mp * 8), off_w = take((size_t)mp * mp * 8);
  const size_t chol_bytes = chol_inverse_workspace_bytes(mp);
  const size_t off_c = take(chol_bytes);
  if (o > ws_bytes) return PTD_ERR_UNSUPPORTED;
  double* R0 = reinterpret_cast<double*>(ws + off_r0);
  double* R1 = reinterpret_cast<double*>(ws + off_r1);
  double* P = reinterpret_cast<double*>(ws + off_p);
  double* G = reinterpret_cast<double*>(ws + off_g);
  double* Wt = reinterpret_cast<double*>(ws + off_w);
  const double* Z = V + m;
  // m random columns (rows of length mp: the padding columns stay unread -- every product below takes m of them)
  hipLaunchKernelGGL(random_fill_kernel, dim3(2048), dim3(256), 0, st, R0, n * mp, 0x5DEECE66DULL);
  PTD_CHECK_LAUNCH("complete_basis (fill)");
  int rc = PTD_OK;
  for (int pass = 0; pass < 2 && kz > 0; ++pass) {
    // P = Z^T R, R -= Z P
    rc = gemm_f64(Z, 1, ldv, R0, mp, 1, P, mp, kz, m, n, 1.0, false, 1, st);
    if (rc != PTD_OK) return rc;
    rc = gemm_f64(Z, ldv, 1, P, mp, 1, R0, mp, n, m, kz, -1.0, true, 1, st);
    if (rc != PTD_OK) return rc;
  }
  double* cur = R0;
  double* nxt = R1;
  for (int pass = 0; pass < 2; ++pass) {
    // G = R^T R = L L^T (padded with the identity to whole 64 x 64 tiles), R <- R L^-T
    hipLaunchKernelGGL(gram_pad_kernel, dim3(1024), dim3(256), 0, st, G, (int)mp, (int)m);
    rc = gemm_f64(cur, 1, mp, cur, mp, 1, G, mp, m, m, n, 1.0, false, 1, st);
    if (rc != PTD_OK) return rc;
    rc = chol_inverse(G, mp, Wt, ws + off_c, chol_bytes, st);
    if (rc != PTD_OK) return rc == PTD_ERR_WORKSPACE ? PTD_ERR_UNSUPPORTED : rc;
    rc = gemm_f64(cur, mp, 1, Wt, mp, 1, nxt, mp, n, m, m, 1.0, false, 1, st);
    if (rc != PTD_OK) return rc;
    std::swap(cur, nxt);
  }
  PTD_CHECK_HIP(hipMemcpy2DAsync(V, (size_t)ldv * 8, cur, (size_t)mp * 8, (size_t)m * 8, (size_t)n,
                                 hipMemcpyDeviceToDevice, st));
  return PTD_OK;
}

// ---- after the reduction: the decisions that depend on the spectrum of T, then eigenvectors and back-transformation.
// h_gap = {smallest relative gap among the requested eigenvalues, longest chain of gaps below ortol} as min_gap_kernel
// left them (read back by the caller behind ONE host synchronisation, which the batched driver shares between its
// matrices).  PTD_ERR_UNSUPPORTED (outputs untouched) when requested eigenvalues are closer than the route serves.
constexpr double TRIDIAG_ORTOL = 1e-7;  // neighbours closer than this (relative to |T|) are re-orthogonalised

static int tridiag_finish(const TridiagPlan& p, char* base, int64_t n, int64_t k, double* evals, double* evecs,
                          int64_t ldv, double cluster_tol, bool all_values, double (&h_gap)[2], hipStream_t st) {
  const double ortol = TRIDIAG_ORTOL;
  double* lam = reinterpret_cast<double*>(base + p.off_lam);
  double* bounds = reinterpret_cast<double*>(base + p.off_bounds);
  int rc = PTD_OK;
  if (getenv("PTD_JACOBI_DEBUG"))
    fprintf(stderr, "[eigh_tridiag] n=%lld min relative gap %.3e, longest chain of gaps below %.0e: %d\n",
            (long long)n, h_gap[0], ortol, (int)h_gap[1]);
  int64_t kreal = k;     // requested eigenvalues above a cluster at the bottom of the spectrum (k: no such cluster)
  static const bool no_long_chains = getenv("PTD_EIGH_LONG_CHAINS") && atoi(getenv("PTD_EIGH_LONG_CHAINS")) == 0;
  // Neighbours closer than cluster_tol |T| (1e-10) or chains of more than 48 close ones used to be refused.  With every
  // computed vector orthonormalised by one Cholesky-QR pass (tridiag_vectors_and_backtransform, long_chains) the limit
  // is where two inverse-iteration vectors stop being independent: eigenvalues are exact to ~1e-16 |T|, so at a gap of
  // 1e-13 |T| the mutual contamination is still ~1e-3 and the Gram matrix well conditioned.  Below that only a
  // cluster at the bottom of the spectrum is served (complete_basis).
  const double hard_tol = 1e-3 * cluster_tol;
  bool force_qr = false;
  if (n > 1 && (!(h_gap[0] > cluster_tol) || h_gap[1] > 48.0) && !no_long_chains && h_gap[0] > hard_tol) force_qr = true;
  if (n > 1 && !force_qr && (!(h_gap[0] > cluster_tol) || h_gap[1] > 48.0)) {
    // clustered.  The one case served here: the cluster is the bottom of the spectrum (see complete_basis)
    static const bool no_completion = getenv("PTD_EIGH_NULL_COMPLETION") && atoi(getenv("PTD_EIGH_NULL_COMPLETION")) == 0;
    bool served = false;
    if (!no_completion && k < n) {
      if (!all_values) {
        rc = tridiag_eigenvalues(p, base, 0, st);
        if (rc != PTD_OK) return rc;
      }
      std::vector<double> hl((size_t)n);
      double hb[4] = {0, 0, 0, 0};
      PTD_CHECK_HIP(hipMemcpyAsync(hl.data(), lam, (size_t)n * 8, hipMemcpyDeviceToHost, st));
      PTD_CHECK_HIP(hipMemcpyAsync(hb, bounds, 32, hipMemcpyDeviceToHost, st));
      PTD_CHECK_HIP(hipStreamSynchronize(st));
      const double tnorm = std::max(hb[3], 2.2250738585072014e-308);
      int64_t jt = 0;                                   // the cluster: indices 0 .. jt, within 1e-9 |T| of lambda_0
      while (jt + 1 < n && hl[(size_t)jt + 1] - hl[0] <= 1e-9 * tnorm) ++jt;
      if (jt >= n - k && jt >= 1 && jt + 1 < n + 1) {
        double gmin = INFINITY;
        int run = 0, longest = 0;
        for (int64_t i = jt + 1; i < n; ++i) {
          const double gap = hl[(size_t)i] - hl[(size_t)i - 1];
          gmin = std::min(gmin, gap);
          run = (i > jt + 1 && gap < ortol * tnorm) ? run + 1 : 0;
          longest = std::max(longest, run);
        }
        if (jt + 1 == n || (gmin / tnorm > cluster_tol && longest <= 48) || (!no_long_chains && gmin / tnorm > hard_tol)) {
          force_qr = jt + 1 < n && !(gmin / tnorm > cluster_tol && longest <= 48);
          kreal = n - 1 - jt;
          h_gap[0] = jt + 1 == n ? 1.0 : gmin / tnorm;
          h_gap[1] = (double)longest;
          served = true;
          if (getenv("PTD_JACOBI_DEBUG"))
            fprintf(stderr, "[eigh_tridiag] n=%lld k=%lld: %lld eigenvalues within 1e-9 |T| of the smallest: %lld vectors "
                    "computed, %lld completed from their complement\n", (long long)n, (long long)k, (long long)(jt + 1),
                    (long long)kreal, (long long)(k - kreal));
        }
      }
    }
    if (!served) {
      set_error("eigh_tridiag: clustered eigenvalues (min relative gap %.3e, chain of %d)", h_gap[0], (int)h_gap[1] + 1);
      return PTD_ERR_UNSUPPORTED;
    }
  }
  // Inverse iteration from a random start with eigenvalues exact to working precision: an iteration damps the
  // component along a neighbour by ~eps |T| / gap, so with every relative gap above 1e-5 two iterations leave it
  // below 1e-21; closer spectra keep the third (and the re-orthogonalisation above 1e-7)
  static const int force_iter = getenv("PTD_INVIT_ITERS") ? atoi(getenv("PTD_INVIT_ITERS")) : 0;
  const int niter = force_iter > 0 ? force_iter : (h_gap[0] > 1e-5 ? 2 : 3);
  if (kreal > 0) {
    rc = tridiag_vectors_and_backtransform(p, base, (int)kreal, evecs + (k - kreal), ldv, h_gap[1] > 0.0 ? ortol : 0.0,
                                           niter, st, force_qr);
    if (rc != PTD_OK) return rc;
  }
  if (kreal < k) {
    // (the inverse-iteration factors u1 .. lm, four n x n buffers in a row, are dead behind the back-transformation)
    rc = complete_basis(evecs, ldv, n, k - kreal, k, base + p.off_u1, (size_t)(p.off_sw - p.off_u1), st);
    if (rc != PTD_OK) {
      if (rc == PTD_ERR_UNSUPPORTED) set_error("eigh_tridiag: the complement of the computed eigenvectors could not be completed");
      return rc;
    }
  }
  PTD_CHECK_HIP(hipMemcpyAsync(evals, lam, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  return PTD_OK;
}

// Eigenvalues (all n) and the eigenvectors of the k largest ones ([n][k], ascending) through
// the tridiagonal route.  Returns PTD_ERR_UNSUPPORTED (and
// leaves the outputs untouched) when two eigenvalues are closer than `cluster_tol` * |T|: the
// caller then uses the Jacobi solver, which needs no gap.
int eigh_tridiag(const double* A, int64_t lda, int64_t n, int64_t k, double* evals, double* evecs, int64_t ldv,
                 void* ws, size_t ws_bytes, double cluster_tol, bool all_values, ptd_eigh_stats* stats,
                 hipStream_t st) {
  const TridiagPlan p = tridiag_plan(n);
  if (ws_bytes < p.total) {
    set_error("eigh_tridiag: workspace %zu < required %zu bytes", ws_bytes, p.total);
    return PTD_ERR_WORKSPACE;
  }
  InflightGuard in_flight;
  char* base = static_cast<char*>(ws);
  double* Aw = reinterpret_cast<double*>(base + p.off_A);
  SymvTimer timer;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->method = 1;
    timer.stride = n >= 512 ? 8 : 1;
    timer.ev.resize(2 * (size_t)(n / timer.stride + 1));
    for (auto& e : timer.ev) PTD_CHECK_HIP(hipEventCreate(&e));
    PTD_CHECK_HIP(hipEventCreate(&e0));
    PTD_CHECK_HIP(hipEventCreate(&e1));
    PTD_CHECK_HIP(hipEventCreate(&e2));
    PTD_CHECK_HIP(hipEventRecord(e0, st));
  }
  auto cleanup = [&]() {
    for (auto& e : timer.ev) (void)hipEventDestroy(e);
    if (e0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); }
  };
  bool resident = true;   // the resident tail of the reduction; its status word is read below
  const int first = (int)std::max<int64_t>(0, n - k - 1);
  double* lam = reinterpret_cast<double*>(base + p.off_lam);
  double* bounds = reinterpret_cast<double*>(base + p.off_bounds);
  double h_gap[2] = {0.0, 0.0};
  int rc = PTD_OK;
  for (;;) {
    hipLaunchKernelGGL(copy_pad_kernel, dim3(2048), dim3(256), 0, st, A, lda, (int)n, Aw, p.ld);
    rc = sytrd_f64(p, base, stats ? &timer : nullptr, resident, st);
    if (rc != PTD_OK) { cleanup(); return rc; }
    if (stats) PTD_CHECK_HIP(hipEventRecord(e1, st));
    // only the gaps that touch one of the k requested (largest) eigenvalues matter, and unless the caller
    // wants every eigenvalue only those k + 1 are computed (one wave each: 5.4 ms for all 4096)
    rc = tridiag_eigenvalues(p, base, all_values ? 0 : first, st);
    if (rc != PTD_OK) { cleanup(); return rc; }
    hipLaunchKernelGGL(min_gap_kernel, dim3(1), dim3(1024), 0, st, lam + first, (int)n - first, bounds, TRIDIAG_ORTOL,
                       bounds + 4);
    int h_res = 0;
    PTD_CHECK_HIP(hipMemcpyAsync(h_gap, bounds + 4, 16, hipMemcpyDeviceToHost, st));
    PTD_CHECK_HIP(hipMemcpyAsync(&h_res, resident_status(p, base), 4, hipMemcpyDeviceToHost, st));
    PTD_CHECK_HIP(hipStreamSynchronize(st));
    if (h_res) {
      if (getenv("PTD_JACOBI_DEBUG"))
        fprintf(stderr, "[eigh_tridiag] resident tail gave up (status %d): blocked path to the end\n", h_res);
      resident_failed(h_res);
      resident = false;
      continue;
    }
    break;
  }
  rc = tridiag_finish(p, base, n, k, evals, evecs, ldv, cluster_tol, all_values, h_gap, st);
  if (rc != PTD_OK) { cleanup(); return rc; }
  if (stats) {
    PTD_CHECK_HIP(hipEventRecord(e2, st));
    PTD_CHECK_HIP(hipEventSynchronize(e2));
    float t_red = 0.f, t_tail = 0.f;
    (void)hipEventElapsedTime(&t_red, e0, e1);
    (void)hipEventElapsedTime(&t_tail, e1, e2);
    stats->total_ms = t_red + t_tail;
    // SYMV launches: every stride-th column carries events; the columns of one stride block have
    // nearly the same trailing order, so the block's time is stride x its sample
    double timed_ms = 0.0;
    for (int64_t j = 0; j < std::min<int64_t>(n - 1, timer.limit); ++j) {
      if (timer.sampled((int)j)) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, timer.start((int)j), timer.stop((int)j));
        const int64_t blk0 = j / timer.stride * timer.stride;
        const int64_t cols = std::min<int64_t>(blk0 + timer.stride, n - 1) - blk0;  // SYMV columns in this block
        timed_ms += (double)ms * (double)cols;
      }
      stats->launches[0] += 1;
      const double m = (double)(n - j - 1);
      stats->work[0] += 8.0 * m * (m - 1.0);  // trailing rows j+1.., columns j+2..: the bytes the SYMV must read
    }
    stats->ms[0] = (float)timed_ms;
    stats->ms[1] = t_red - stats->ms[0];   // per-column vector kernels + rank-2k updates + launch gaps
    stats->launches[1] = 2 * (int)n;
    stats->ms[3] = t_tail;
    stats->work[2] = 4.0 / 3.0 * (double)n * (double)n * (double)n;  // rank-2k updates of the full trailing square
    cleanup();
  }
  return PTD_OK;
}

size_t tridiag_workspace_bytes(int64_t n) { return tridiag_plan(n).total; }

// ---- several matrices of one order per launch
// dwain's precompute pass (dwain.py:580-633) is a loop of independent torch.linalg.eigh calls (:162) on the layers of a
// split.  One direct reduction is a chain of ~2 n dependent launches (alpha, SYMV) whose cost per column is launch
// latency once the trailing block is small and the stream of the triangle while it is large; matrices of the same
// order advance column by column in lockstep, so ONE launch per kernel and column serves all of them: blockIdx.y is
// the matrix, every workspace pointer moves by y * bstride (the workspaces are `count` copies of one plan, bstride
// apart).  The latency one chain cannot hide is shared by the batch -- by construction, on one stream, from one host
// thread -- where round 5 interleaved the chains of concurrent host threads on streams probed for distinct hardware
// queues.  The resident kernels (one matrix in the registers of the whole chip) do not take part: `count` of them would
// run one after the other at 8-18 us a column and matrix; the batched columns cost less per matrix from count = 2 on.
size_t tridiag_batched_workspace_bytes(int64_t n, int count) {
  return align_up(tridiag_plan(n).total, 4096) * (size_t)std::max(count, 1);
}

// rcs[b]: PTD_OK or PTD_ERR_UNSUPPORTED (matrix b's requested eigenvalues are clustered beyond what the route serves:
// its outputs are untouched, the caller hands it to the Jacobi solver); any other failure is returned at once.
// stats (optional): method 1, the figures of the WHOLE batch -- ms[0] / launches[0] / work[0] the sampled SYMV launches
// (each serves `count` matrices: work[0] counts their bytes together).
int eigh_tridiag_batched(const double* const* As, int64_t lda, int count, int64_t n, int64_t k, double* const* evals,
                         double* const* evecs, int64_t ldv, void* ws, size_t ws_bytes, double cluster_tol,
                         bool all_values, int* rcs, ptd_eigh_stats* stats, hipStream_t st) {
  const TridiagPlan p = tridiag_plan(n);
  const size_t bstride = align_up(p.total, 4096);
  if (ws_bytes < bstride * (size_t)count) {
    set_error("eigh_tridiag_batched: workspace %zu < required %zu bytes", ws_bytes, bstride * (size_t)count);
    return PTD_ERR_WORKSPACE;
  }
  InflightGuard in_flight;
  char* base = static_cast<char*>(ws);
  SymvTimer timer;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->method = 1;
    timer.stride = n >= 512 ? 8 : 1;
    timer.ev.resize(2 * (size_t)(n / timer.stride + 1));
    for (auto& e : timer.ev) PTD_CHECK_HIP(hipEventCreate(&e));
    PTD_CHECK_HIP(hipEventCreate(&e0));
    PTD_CHECK_HIP(hipEventCreate(&e1));
    PTD_CHECK_HIP(hipEventCreate(&e2));
    PTD_CHECK_HIP(hipEventRecord(e0, st));
  }
  auto cleanup = [&]() {
    for (auto& e : timer.ev) (void)hipEventDestroy(e);
    if (e0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); }
  };
  for (int b = 0; b < count; ++b)
    hipLaunchKernelGGL(copy_pad_kernel, dim3(2048), dim3(256), 0, st, As[b], lda, (int)n,
                       reinterpret_cast<double*>(base + b * bstride + p.off_A), p.ld);
  int rc = sytrd_f64(p, base, stats ? &timer : nullptr, false, st, count, bstride);
  if (rc != PTD_OK) { cleanup(); return rc; }
  if (stats) PTD_CHECK_HIP(hipEventRecord(e1, st));
  const int first = (int)std::max<int64_t>(0, n - k - 1);
  std::vector<double> gaps(2 * (size_t)count, 0.0);
  for (int b = 0; b < count; ++b) {
    char* bb = base + b * bstride;
    rc = tridiag_eigenvalues(p, bb, all_values ? 0 : first, st);
    if (rc != PTD_OK) { cleanup(); return rc; }
    double* lam = reinterpret_cast<double*>(bb + p.off_lam);
    double* bounds = reinterpret_cast<double*>(bb + p.off_bounds);
    hipLaunchKernelGGL(min_gap_kernel, dim3(1), dim3(1024), 0, st, lam + first, (int)n - first, bounds, TRIDIAG_ORTOL,
                       bounds + 4);
    PTD_CHECK_HIP(hipMemcpyAsync(&gaps[2 * (size_t)b], bounds + 4, 16, hipMemcpyDeviceToHost, st));
  }
  PTD_CHECK_HIP(hipStreamSynchronize(st));     // the ONE host synchronisation of the batch (clustered spectra aside)
  for (int b = 0; b < count; ++b) {
    double h_gap[2] = {gaps[2 * (size_t)b], gaps[2 * (size_t)b + 1]};
    rcs[b] = tridiag_finish(p, base + b * bstride, n, k, evals[b], evecs[b], ldv, cluster_tol, all_values, h_gap, st);
    if (rcs[b] != PTD_OK && rcs[b] != PTD_ERR_UNSUPPORTED) { cleanup(); return rcs[b]; }
  }
  if (stats) {
    PTD_CHECK_HIP(hipEventRecord(e2, st));
    PTD_CHECK_HIP(hipEventSynchronize(e2));
    float t_red = 0.f, t_tail = 0.f;
    (void)hipEventElapsedTime(&t_red, e0, e1);
    (void)hipEventElapsedTime(&t_tail, e1, e2);
    stats->total_ms = t_red + t_tail;
    stats->sweeps = count;      // (Jacobi sweeps are zero on this route: the field carries the batch size here)
    double timed_ms = 0.0;
    for (int64_t j = 0; j < n - 1; ++j) {
      if (n - j - 1 < 1) break;
      if (timer.sampled((int)j)) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, timer.start((int)j), timer.stop((int)j));
        const int64_t blk0 = j / timer.stride * timer.stride;
        const int64_t cols = std::min<int64_t>(blk0 + timer.stride, n - 1) - blk0;
        timed_ms += (double)ms * (double)cols;
      }
      stats->launches[0] += 1;
      const double m = (double)(n - j - 1);
      stats->work[0] += (double)count * 8.0 * m * (m - 1.0);
    }
    stats->ms[0] = (float)timed_ms;
    stats->ms[1] = t_red - stats->ms[0];
    stats->launches[1] = 2 * (int)n;
    stats->ms[3] = t_tail;
    stats->work[2] = (double)count * 4.0 / 3.0 * (double)n * (double)n * (double)n;
    cleanup();
  }
  return PTD_OK;
}

int concurrent_chains_exchange(int chains) { return device_state().chains.exchange(chains); }

// Diagnostic entry: tridiagonalise A and return (d, e, tau) and the eigenvalues of T.
int tridiagonalize_f64(const double* A, int64_t lda, int64_t n, double* d_out, double* e_out, double* evals_out,
                       void* ws, size_t ws_bytes, hipStream_t st) {
  PTD_REQUIRE(A && ws && n >= 1 && lda >= n, "ptd_tridiagonalize: bad argument");
  const TridiagPlan p = tridiag_plan(n);
  if (ws_bytes < p.total) {
    set_error("ptd_tridiagonalize: workspace %zu < required %zu bytes", ws_bytes, p.total);
    return PTD_ERR_WORKSPACE;
  }
  InflightGuard in_flight;
  char* base = static_cast<char*>(ws);
  double* Aw = reinterpret_cast<double*>(base + p.off_A);
  hipLaunchKernelGGL(copy_pad_kernel, dim3(2048), dim3(256), 0, st, A, lda, (int)n, Aw, p.ld);
  int rc = sytrd_f64(p, base, nullptr, true, st);
  if (rc != PTD_OK) return rc;
  int h_res = 0;
  PTD_CHECK_HIP(hipMemcpyAsync(&h_res, resident_status(p, base), 4, hipMemcpyDeviceToHost, st));
  PTD_CHECK_HIP(hipStreamSynchronize(st));
  if (h_res) {   // the resident tail gave up: once more on the blocked path
    resident_failed(h_res);
    hipLaunchKernelGGL(copy_pad_kernel, dim3(2048), dim3(256), 0, st, A, lda, (int)n, Aw, p.ld);
    rc = sytrd_f64(p, base, nullptr, false, st);
  }
  if (rc != PTD_OK) return rc;
  rc = tridiag_eigenvalues(p, base, 0, st);
  if (rc != PTD_OK) return rc;
  if (d_out) PTD_CHECK_HIP(hipMemcpyAsync(d_out, base + p.off_d, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  if (e_out) PTD_CHECK_HIP(hipMemcpyAsync(e_out, base + p.off_e, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  if (evals_out) PTD_CHECK_HIP(hipMemcpyAsync(evals_out, base + p.off_lam, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  return PTD_OK;
}

}  // namespace ptd
