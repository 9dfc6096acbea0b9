// Top-k eigenvectors of  C = W Ex W^T  (W [n_o, n_i], n_o > n_i, Ex [n_i, n_i] symmetric PSD)
// without ever forming the n_o x n_o matrix -- the covariance of y = x W^T for a layer that
// widens its input (Llama gate / up: 4096 -> 14336).  Exact algebra, all in f64:
//     G = W^T W = L L^T                       (blocked Cholesky, chol.hip)
//     B = L^T Ex L                            (n_i x n_i, symmetric)
//     B s = lambda s                          (ptd_eigh route, top k)
//     u = W L^-T s                            => C u = lambda u,  |u| = |s| = 1
// so a 14336^2 eigenproblem (1.6 GB matrix, 8/3 n^3 = 7.9e12 bytes of SYMV traffic) becomes a
// 4096^2 one plus five f64 MFMA products.  Damping C + damp I shifts eigenvalues only.
// Fails with PTD_ERR_UNSUPPORTED when G is not numerically positive definite (W rank deficient):
// the caller then decomposes C directly.
#include <algorithm>
#include <cstring>

#include "common.h"
#include "kernels.h"

namespace ptd {

int cholesky_f64(double* L, int np, double* linv_ws, int linv_stride, int* fail, hipStream_t st);
int eigh_select(const double* A, int64_t lda, int64_t n, int64_t k, double* evals, double* evecs, int64_t ldv,
                void* ws, size_t ws_bytes, int* sweeps_out, ptd_eigh_stats* stats, hipStream_t st);  // api.hip
size_t eigh_select_workspace_bytes(int64_t n);

namespace {

constexpr int FB = 64;

template <typename T>
__device__ __forceinline__ double to_f64(T v);
template <>
__device__ __forceinline__ double to_f64<float>(float v) { return (double)v; }
template <>
__device__ __forceinline__ double to_f64<double>(double v) { return v; }
template <>
__device__ __forceinline__ double to_f64<unsigned short>(unsigned short v) { return (double)bf16_to_f32(v); }

// D[r][c] = (r < rows && c < cols) ? S[r][c] : (r == c ? diag_pad : 0)
template <typename T>
__global__ void widen_pad_kernel(const T* __restrict__ S, int64_t lds, int rows, int cols, double* __restrict__ D,
                                 int64_t ldd, int prow, int pcol, double diag_pad) {
  const int64_t total = (int64_t)prow * pcol;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / pcol), c = (int)(e % pcol);
    D[(int64_t)r * ldd + c] = (r < rows && c < cols) ? to_f64(S[(int64_t)r * lds + c]) : (r == c ? diag_pad : 0.0);
  }
}

// zero the strict upper triangle (the Cholesky kernels leave stale values of G there)
__global__ void tril_kernel(double* __restrict__ L, int np) {
  const int64_t total = (int64_t)np * np;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / np), c = (int)(e % np);
    if (c > r) L[e] = 0.0;
  }
}

// B <- (B + B^T) / 2 on the n x n leading block; identity on the padding (keeps B PSD, the
// padded eigenvalues are exact and tiny so they never reach the top k)
// (mode 0: only the padding is written -- the matrix holds its lower triangle alone; mode 1: the average; mode 2: the
// upper triangle becomes the mirror image of the lower one, which alone was computed)
__global__ void symmetrize_kernel(double* __restrict__ B, int np, int n, double pad_diag, int mode) {
  const int64_t total = (int64_t)np * np;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / np), c = (int)(e % np);
    if (r >= n || c >= n) {
      B[e] = (r == c) ? pad_diag : 0.0;
    } else if (c > r && mode == 1) {
      const double v = 0.5 * (B[e] + B[(int64_t)c * np + r]);
      B[e] = v;
      B[(int64_t)c * np + r] = v;
    } else if (c > r && mode == 2) {
      B[e] = B[(int64_t)c * np + r];
    }
  }
}

// G = sum of `ns` slabs in index order (a deterministic K split), lower triangle only
__global__ void sum_slabs_lower_kernel(double* __restrict__ G, const double* __restrict__ S, int64_t slab, int ns,
                                       int np) {
  const int half = np / 2;
  const int64_t total2 = (int64_t)np * half;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total2; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / half), c2 = (int)(i % half);
    if (2 * c2 > r) continue;
    double2 acc = reinterpret_cast<const double2*>(S)[i];
    for (int q = 1; q < ns; ++q) {
      const double2 v = reinterpret_cast<const double2*>(S + q * slab)[i];
      acc.x += v.x;
      acc.y += v.y;
    }
    reinterpret_cast<double2*>(G)[i] = acc;
  }
}

// min and max of diag(L): max / min bounds cond(L) = cond(W) from below
__global__ void diag_range_kernel(const double* __restrict__ L, int np, int n, double* __restrict__ out) {
  __shared__ double rmin[16], rmax[16];
  double lo = INFINITY, hi = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = L[(int64_t)i * np + i];
    lo = fmin(lo, v);
    hi = fmax(hi, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o));
    hi = fmax(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) { rmin[threadIdx.x >> 6] = lo; rmax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fmin(lo, rmin[w]); hi = fmax(hi, rmax[w]); }
    out[0] = lo;
    out[1] = hi;
  }
}

// Inverse of the 256 x 256 diagonal blocks of L (lower triangular) from the inverses of their four 64 x 64 diagonal
// tiles (`linv`, from the factorisation): X_ii = linv_i,  X_ij = -linv_i sum_{k = j}^{i - 1} L_ik X_kj  for i > j --
// 16 products of 64 x 64 tiles a block on the f64 matrix cores, one workgroup per block.  With it the back substitution
// T = L^-T S runs in np / 256 steps of two products with K = 256 instead of np / 64 steps of thin ones (128 launches,
// 4.8 ms at np = 4096, k = 2048).  out: [np / 256][256][256], upper blocks zero.
__global__ __launch_bounds__(256) void tri_inv256_kernel(const double* __restrict__ L, int np,
                                                         const double* __restrict__ linv, double* __restrict__ out) {
  __shared__ double As[FB][FB + 1];
  __shared__ double Bs[FB][FB + 1];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  double* O = out + (size_t)g * 256 * 256;
  const double* Lg = L + ((size_t)g * 256) * np + (size_t)g * 256;     // the block's corner in L
  auto load = [&](double (*dst)[FB + 1], const double* src, int ld) {
    for (int e = tid; e < FB * FB; e += 256) dst[e >> 6][e & 63] = src[(size_t)(e >> 6) * ld + (e & 63)];
  };
  auto store = [&](const f64x4 (&acc)[4], int bi, int bj, double sign) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        O[(size_t)(64 * bi + 16 * wid + l4 + 4 * r) * 256 + 64 * bj + 16 * cb + l15] = sign * acc[cb][r];
  };
  auto mma = [&](f64x4 (&acc)[4]) {       // acc += As Bs (operand map: a = A[row = lane & 15][k = lane >> 4], ...)
#pragma unroll 4
    for (int kk = 0; kk < FB; kk += 4) {
      const double av = As[16 * wid + l15][kk + l4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[kk + l4][16 * cb + l15], acc[cb], 0, 0, 0);
    }
  };
  // zeros above the diagonal, the tiles' own inverses on it
  for (int e = tid; e < 256 * 256; e += 256) {
    const int r = e >> 8, c = e & 255;
    if ((c >> 6) > (r >> 6)) O[e] = 0.0;
    else if ((c >> 6) == (r >> 6)) O[e] = linv[((size_t)(4 * g + (r >> 6)) * FB + (r & 63)) * FB + (c & 63)];
  }
  __syncthreads();
  for (int j = 0; j < 3; ++j)
    for (int i = j + 1; i < 4; ++i) {
      f64x4 acc[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = f64x4{0.0, 0.0, 0.0, 0.0};
      for (int k = j; k < i; ++k) {
        __syncthreads();                                   // (the operands of the previous product are consumed)
        load(As, Lg + (size_t)(64 * i) * np + 64 * k, np);
        load(Bs, O + (size_t)(64 * k) * 256 + 64 * j, 256);     // X_kj (finished: k < i; written by this workgroup)
        __syncthreads();
        mma(acc);
      }
      __syncthreads();
      // X_ij = -linv_i acc
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Bs[16 * wid + l4 + 4 * r][16 * cb + l15] = acc[cb][r];
      load(As, linv + (size_t)(4 * g + i) * FB * FB, FB);
      __syncthreads();
      f64x4 x[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) x[cb] = f64x4{0.0, 0.0, 0.0, 0.0};
      mma(x);
      store(x, i, j, -1.0);
      __threadfence_block();
    }
}

}  // namespace

struct FactoredPlan {
  int np;  // n_i rounded up to 64
  size_t off_W, off_G, off_P, off_B, off_linv, off_fail, off_S, off_T, off_evals, off_eigh, total;
  size_t eigh_bytes;
};

static FactoredPlan factored_plan(int64_t n_o, int64_t n_i, int64_t k) {
  FactoredPlan p{};
  p.np = (int)align_up((size_t)n_i, FB);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
  p.off_W = take((size_t)n_o * p.np * 8);
  p.off_G = take((size_t)p.np * p.np * 8);
  p.off_P = take((size_t)p.np * p.np * 8);
  p.off_B = take((size_t)p.np * p.np * 8);
  p.off_linv = take((size_t)(p.np / FB) * FB * FB * 8);
  p.off_fail = take(256);
  p.off_S = take((size_t)p.np * k * 8);
  p.off_T = take((size_t)p.np * k * 8);
  p.off_evals = take((size_t)p.np * 8);
  p.eigh_bytes = eigh_select_workspace_bytes(p.np);
  p.off_eigh = take(p.eigh_bytes);
  p.total = o;
  return p;
}

size_t eigh_factored_workspace_bytes(int64_t n_o, int64_t n_i, int64_t k) { return factored_plan(n_o, n_i, k).total; }

// ---- first half: everything up to B = L^T Ex L (left in the workspace, [np][np])
int eigh_factored_prepare(const void* W, int64_t ldw, int w_dtype, int64_t n_o, int64_t n_i, const double* Ex,
                          int64_t ldx, int64_t k, void* ws, size_t ws_bytes, double** B_out, int64_t* np_out,
                          hipStream_t st) {
  PTD_REQUIRE(W && Ex && ws, "ptd_eigh_factored: null pointer");
  PTD_REQUIRE(n_i >= 1 && n_o >= n_i && k >= 1 && k <= n_i && ldw >= n_i && ldx >= n_i,
              "ptd_eigh_factored: bad shape n_o=%lld n_i=%lld k=%lld", (long long)n_o, (long long)n_i, (long long)k);
  const FactoredPlan p = factored_plan(n_o, n_i, k);
  if (ws_bytes < p.total) {
    set_error("ptd_eigh_factored: workspace %zu < required %zu bytes", ws_bytes, p.total);
    return PTD_ERR_WORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  const int np = p.np;
  double* W64 = reinterpret_cast<double*>(base + p.off_W);   // [n_o][np], zero padded columns
  double* G = reinterpret_cast<double*>(base + p.off_G);     // G, then L (lower)
  double* P = reinterpret_cast<double*>(base + p.off_P);
  double* B = reinterpret_cast<double*>(base + p.off_B);
  double* linv = reinterpret_cast<double*>(base + p.off_linv);
  int* fail = reinterpret_cast<int*>(base + p.off_fail);

  // W in f64 (zero padded to np columns)
  if (w_dtype == PTD_F32)
    hipLaunchKernelGGL((widen_pad_kernel<float>), dim3(4096), dim3(256), 0, st, (const float*)W, ldw, (int)n_o,
                       (int)n_i, W64, (int64_t)np, (int)n_o, np, 0.0);
  else if (w_dtype == PTD_BF16)
    hipLaunchKernelGGL((widen_pad_kernel<unsigned short>), dim3(4096), dim3(256), 0, st, (const unsigned short*)W,
                       ldw, (int)n_o, (int)n_i, W64, (int64_t)np, (int)n_o, np, 0.0);
  else if (w_dtype == PTD_F64)
    hipLaunchKernelGGL((widen_pad_kernel<double>), dim3(4096), dim3(256), 0, st, (const double*)W, ldw, (int)n_o,
                       (int)n_i, W64, (int64_t)np, (int)n_o, np, 0.0);
  else {
    set_error("ptd_eigh_factored: unsupported weight dtype");
    return PTD_ERR_UNSUPPORTED;
  }
  PTD_CHECK_LAUNCH("widen W");
  // G = W^T W  (+ identity on the padding so the Cholesky stays positive definite).  Only the 128 x 128 tiles that
  // touch the lower triangle are computed (the factorisation reads nothing else and tril_kernel clears the rest
  // afterwards): 4.8e11 flop at n_o = 14336, n_i = 4096 become 2.5e11.  The 528 tiles left are one round of the 512
  // workgroup slots plus a nearly empty second one, so the K range is cut into slabs (summed in index order) until the
  // last round is a small share of the launch; the slabs live in the buffers of the later steps (P, B, S, T, eigh).
  int nslabs = 1, ksplit = 1;
  const int64_t slab_elems = (int64_t)np * np;
  {
    const int64_t avail = (int64_t)((p.total - p.off_P) / ((size_t)slab_elems * 8));
    const int tm = np / 128;
    const int64_t tiles = (int64_t)tm * (tm + 1) / 2;
    double best = 0.0;
    for (int ks = 1; ks <= (int)std::min<int64_t>(avail, 16); ++ks) {
      const int64_t kc = (int64_t)align_up((size_t)ceil_div(n_o, (int64_t)ks), 16);
      const int64_t ns = ceil_div(n_o, kc);
      const double cost = (double)ceil_div(tiles * ns, (int64_t)512) * (double)kc / (double)n_o + 0.01 * (ns > 1 ? ns : 0);
      if (ks == 1 || cost < best) { best = cost; ksplit = ks; }
    }
  }
  double* slabs = ksplit > 1 ? P : G;
  int rc = gemm_f64_slabs(W64, 1, np, W64, np, 1, slabs, np, slab_elems, np, np, n_o, 1.0, ksplit, &nslabs, true, st);
  if (rc != PTD_OK) return rc;
  if (ksplit > 1)
    hipLaunchKernelGGL(sum_slabs_lower_kernel, dim3(2048), dim3(256), 0, st, G, slabs, slab_elems, nslabs, np);
  if (np > n_i) hipLaunchKernelGGL(symmetrize_kernel, dim3(2048), dim3(256), 0, st, G, np, (int)n_i, 1.0, 0);
  PTD_CHECK_HIP(hipMemsetAsync(fail, 0, 16, st));
  rc = cholesky_f64(G, np, linv, FB * FB, fail, st);
  if (rc != PTD_OK) return rc;
  double* drange = reinterpret_cast<double*>(base + p.off_fail + 16);
  hipLaunchKernelGGL(diag_range_kernel, dim3(1), dim3(1024), 0, st, G, np, (int)n_i, drange);
  struct { int fail; int pad; double pad2; double lo, hi; } h{};
  PTD_CHECK_HIP(hipMemcpyAsync(&h, fail, 32, hipMemcpyDeviceToHost, st));
  PTD_CHECK_HIP(hipStreamSynchronize(st));
  // the eigenvectors inherit an error of order eps * cond(W)^2 from G = W^T W: refuse beyond
  // cond(W) ~ 1e4 (diag(L) ratio is a lower bound of it), the caller then takes the direct route
  if (h.fail || !(h.lo > 1e-4 * h.hi)) {
    set_error("ptd_eigh_factored: W^T W is not safely positive definite (diag(L) range %.3e .. %.3e)", h.lo, h.hi);
    return PTD_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(tril_kernel, dim3(2048), dim3(256), 0, st, G, np);  // G now holds L, upper = 0
  // Ex (n_i x n_i) widened into B's buffer first, P = Ex L, B = L^T P
  hipLaunchKernelGGL((widen_pad_kernel<double>), dim3(4096), dim3(256), 0, st, Ex, ldx, (int)n_i, (int)n_i, B,
                     (int64_t)np, np, np, 0.0);
  // P = Ex L, in four column panels: L is lower triangular, so panel q (columns from c0 = q np / 4) needs only the rows
  // of L -- the K range -- from c0 on (137 -> 86 Gflop at np = 4096)
  {
    const int npan = (np % 512 == 0) ? 4 : 1, pw = np / npan;
    for (int q = 0; q < npan; ++q) {
      const int c0 = q * pw;
      rc = gemm_f64(B + c0, np, 1, G + (size_t)c0 * np + c0, np, 1, P + c0, np, np, pw, np - c0, 1.0, false, 1, st);
      if (rc != PTD_OK) return rc;
    }
  }
  // B = L^T P is symmetric: its lower tiles alone are formed (half the flops), the upper triangle is their mirror image
  {
    int ns = 0;
    rc = gemm_f64_slabs(G, 1, np, P, np, 1, B, np, (int64_t)np * np, np, np, np, 1.0, 1, &ns, true, st);
    if (rc != PTD_OK) return rc;
  }
  hipLaunchKernelGGL(symmetrize_kernel, dim3(2048), dim3(256), 0, st, B, np, (int)n_i, 0.0, 2);
  PTD_CHECK_LAUNCH("factored products");
  if (B_out) *B_out = B;
  if (np_out) *np_out = np;
  return PTD_OK;
}

// ---- second half: u = W L^-T s for the top-k eigenvectors S [np][lds >= k] of B (evals [np], ascending; only the last
// k are read).  The workspace is the one eigh_factored_prepare filled (W64, L and linv are read from it).
int eigh_factored_finish(int64_t n_o, int64_t n_i, int64_t k, const double* evals, const double* Sin, int64_t lds,
                         double* evals_k, double* U, int64_t ldu, void* ws, size_t ws_bytes, hipStream_t st) {
  PTD_REQUIRE(Sin && U && ws && evals && ldu >= k && lds >= k, "ptd_eigh_factored_finish: bad argument");
  const FactoredPlan p = factored_plan(n_o, n_i, k);
  if (ws_bytes < p.total) {
    set_error("ptd_eigh_factored_finish: workspace %zu < required %zu bytes", ws_bytes, p.total);
    return PTD_ERR_WORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  const int np = p.np;
  double* W64 = reinterpret_cast<double*>(base + p.off_W);
  double* G = reinterpret_cast<double*>(base + p.off_G);
  double* P = reinterpret_cast<double*>(base + p.off_P);
  double* linv = reinterpret_cast<double*>(base + p.off_linv);
  double* S = reinterpret_cast<double*>(base + p.off_S);     // [np][k] eigenvectors of B (destroyed by the substitution)
  double* T = reinterpret_cast<double*>(base + p.off_T);     // [np][k] L^-T S
  if (Sin != S)
    PTD_CHECK_HIP(hipMemcpy2DAsync(S, (size_t)k * 8, Sin, (size_t)lds * 8, (size_t)k * 8, (size_t)np,
                                   hipMemcpyDeviceToDevice, st));
  int rc = PTD_OK;
  // T = L^-T S : blocked back substitution with the inverses of the diagonal blocks -- 256 x 256 ones where np allows
  // (tri_inv256_kernel; they live in P, which is dead by now)
  static const bool no_inv256 = getenv("PTD_FACTORED_INV256") && atoi(getenv("PTD_FACTORED_INV256")) == 0;
  const int nblk = (np % 256 == 0 && np >= 512 && !no_inv256) ? 0 : np / FB;
  if (nblk == 0) {
    double* Li = P;
    hipLaunchKernelGGL(tri_inv256_kernel, dim3((unsigned)(np / 256)), dim3(256), 0, st, G, np, linv, Li);
    for (int g = np / 256 - 1; g >= 0; --g) {
      // T_g = (L_gg^-1)^T S_g
      rc = gemm_f64(Li + (size_t)g * 256 * 256, 1, 256, S + (size_t)g * 256 * k, k, 1, T + (size_t)g * 256 * k, k, 256, k,
                    256, 1.0, false, 1, st);
      if (rc != PTD_OK) return rc;
      // S[0 : 256 g] -= L[the block's rows, 0 : 256 g]^T T_g
      if (g > 0) {
        rc = gemm_f64(G + (size_t)g * 256 * np, 1, np, T + (size_t)g * 256 * k, k, 1, S, k, (int64_t)g * 256, k, 256, -1.0,
                      true, 1, st);
        if (rc != PTD_OK) return rc;
      }
    }
  }
  for (int b = nblk - 1; b >= 0; --b) {
    // T_b = (L_bb^-1)^T S_b
    rc = gemm_f64(linv + (size_t)b * FB * FB, 1, FB, S + (size_t)b * FB * k, k, 1, T + (size_t)b * FB * k, k, FB, k, FB,
                  1.0, false, 1, st);
    if (rc != PTD_OK) return rc;
    // S[0 : b*64] -= L[b-block rows, 0 : b*64]^T T_b
    if (b > 0) {
      rc = gemm_f64(G + (size_t)b * FB * np, 1, np, T + (size_t)b * FB * k, k, 1, S, k, (int64_t)b * FB, k, FB, -1.0,
                    true, 1, st);
      if (rc != PTD_OK) return rc;
    }
  }
  // U = W T  [n_o, k]
  rc = gemm_f64(W64, np, 1, T, k, 1, U, ldu, n_o, k, np, 1.0, false, 1, st);
  if (rc != PTD_OK) return rc;
  if (evals_k)
    PTD_CHECK_HIP(hipMemcpyAsync(evals_k, evals + (np - k), (size_t)k * 8, hipMemcpyDeviceToDevice, st));
  return PTD_OK;
}

int eigh_factored(const void* W, int64_t ldw, int w_dtype, int64_t n_o, int64_t n_i, const double* Ex, int64_t ldx,
                  int64_t k, double* evals_k, double* U, int64_t ldu, void* ws, size_t ws_bytes, hipStream_t st) {
  PTD_REQUIRE(U && ldu >= k, "ptd_eigh_factored: bad output");
  double* B = nullptr;
  int64_t np = 0;
  int rc = eigh_factored_prepare(W, ldw, w_dtype, n_o, n_i, Ex, ldx, k, ws, ws_bytes, &B, &np, st);
  if (rc != PTD_OK) return rc;
  const FactoredPlan p = factored_plan(n_o, n_i, k);
  char* base = static_cast<char*>(ws);
  double* S = reinterpret_cast<double*>(base + p.off_S);
  double* evals = reinterpret_cast<double*>(base + p.off_evals);
  // top-k eigenvectors of B
  rc = eigh_select(B, np, np, k, evals, S, k, base + p.off_eigh, p.eigh_bytes, nullptr, nullptr, st);
  if (rc != PTD_OK) return rc;
  return eigh_factored_finish(n_o, n_i, k, evals, S, k, evals_k, U, ldu, ws, ws_bytes, st);
}

}  // namespace ptd
