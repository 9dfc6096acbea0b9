// f64 dense products on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64, 78.6 TFLOP/s
// peak), used inside the symmetric eigensolver (rank-2k trailing updates of the
// tridiagonalisation, block-reflector back-transformation, Rayleigh-Ritz products).
//
//   C[M,N] = alpha * sum_k A(m,k) B(k,n) + beta * C        (beta in {0, 1}; optional split-K
//                                                           with f64 atomics when beta == 1)
// Operands are addressed by element strides (exactly one stride of each operand is 1), so
// transposed views cost nothing.  64x64 output tile per 256-thread workgroup, four waves each
// own a 32x32 quadrant as 2x2 MFMA tiles; K step 16, register-staged double buffering.  LDS
// image [k][m] with pitch 80 doubles: the one-f64-per-lane operand read (16 consecutive m of
// row k, k = lane >> 4) then touches all 64 banks exactly once per half wave.
#include <algorithm>

#include "common.h"

namespace ptd {

namespace {

constexpr int DM = 64, DN = 64, DK = 16, DP = 80;

struct GemmF64Args {
  const double* A;
  int64_t sam, sak;
  const double* B;
  int64_t sbk, sbn;
  double* C;
  int64_t ldc;
  int M, N, K;
  double alpha;
  int beta1;   // 1: accumulate into C
  int atomic;  // 1: split K, accumulate with atomics (requires beta1)
  int kchunk;
  int tiles_m;
  const double* A2;  // optional second operand pair with the same strides: C gets A B + A2 B2 in one pass
  const double* B2;  // over C (the symmetric rank-2k update V^T W + W^T V)
  double* row0_out;  // optional: the updated first row of C is also written here (the next panel's first column)
};

// this thread's 2 x (2 doubles) of a 64 (r) x 16 (k) operand tile
template <bool KC>
__device__ __forceinline__ void fetch64(const double* __restrict__ P, int64_t s, int r_lim, int k_lim, int tid,
                                        double2 (&v)[2]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int idx = tid + 256 * p;
    const int r = KC ? (idx >> 3) : (idx & 31) * 2;
    const int k = KC ? (idx & 7) * 2 : (idx >> 5);
    const double* q = KC ? P + (int64_t)r * s + k : P + (int64_t)k * s + r;
    const bool ok0 = r < r_lim && k < k_lim;
    const bool ok1 = KC ? (r < r_lim && k + 1 < k_lim) : (r + 1 < r_lim && k < k_lim);
    v[p].x = ok0 ? q[0] : 0.0;
    v[p].y = ok1 ? q[1] : 0.0;
  }
}

template <bool KC>
__device__ __forceinline__ void stash64(double* __restrict__ L, int tid, const double2 (&v)[2]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int idx = tid + 256 * p;
    if (KC) {
      const int r = idx >> 3, k = (idx & 7) * 2;
      L[k * DP + r] = v[p].x;
      L[(k + 1) * DP + r] = v[p].y;
    } else {
      const int k = idx >> 5, r = (idx & 31) * 2;
      *reinterpret_cast<double2*>(&L[k * DP + r]) = v[p];
    }
  }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f64_kernel(const GemmF64Args a) {
  __shared__ __attribute__((aligned(16))) double As[DK * DP];
  __shared__ __attribute__((aligned(16))) double Bs[DK * DP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int ti = blockIdx.x % a.tiles_m, tj = blockIdx.x / a.tiles_m;
  const int m0 = ti * DM, n0 = tj * DN;
  const int kbeg = blockIdx.y * a.kchunk;
  const int kend = min(a.K, kbeg + a.kchunk);
  const int nk = (kend - kbeg + DK - 1) / DK;
  const double* Ap = a.A + (int64_t)m0 * a.sam + (int64_t)kbeg * a.sak;
  const double* Bp = a.B + (int64_t)n0 * a.sbn + (int64_t)kbeg * a.sbk;
  const int64_t sa = AKC ? a.sam : a.sak, sb = BKC ? a.sbn : a.sbk;
  const int64_t astep = (int64_t)DK * a.sak, bstep = (int64_t)DK * a.sbk;
  const int m_lim = a.M - m0, n_lim = a.N - n0;

  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  double2 ra[2], rb[2];
  const int l15 = lane & 15, l4 = lane >> 4;
  const int npass = a.A2 ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  if (pass == 1) {
    Ap = a.A2 + (int64_t)m0 * a.sam + (int64_t)kbeg * a.sak;
    Bp = a.B2 + (int64_t)n0 * a.sbn + (int64_t)kbeg * a.sbk;
  }
  if (nk > 0) {
    fetch64<AKC>(Ap, sa, m_lim, kend - kbeg, tid, ra);
    fetch64<BKC>(Bp, sb, n_lim, kend - kbeg, tid, rb);
    stash64<AKC>(As, tid, ra);
    stash64<BKC>(Bs, tid, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      const int k_lim = kend - kbeg - (kt + 1) * DK;
      fetch64<AKC>(Ap + (kt + 1) * astep, sa, m_lim, k_lim, tid, ra);
      fetch64<BKC>(Bp + (kt + 1) * bstep, sb, n_lim, k_lim, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < DK; kk += 4) {
      const double a0 = As[(kk + l4) * DP + wr * 32 + l15];
      const double a1 = As[(kk + l4) * DP + wr * 32 + 16 + l15];
      const double b0 = Bs[(kk + l4) * DP + wc * 32 + l15];
      const double b1 = Bs[(kk + l4) * DP + wc * 32 + 16 + l15];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      stash64<AKC>(As, tid, ra);
      stash64<BKC>(Bs, tid, rb);
      __syncthreads();
    }
  }
  }
  // C/D map of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + i * 16 + l4 + 4 * r;
        const int col = n0 + wc * 32 + j * 16 + l15;
        if (row >= a.M || col >= a.N) continue;
        double* c = a.C + (int64_t)row * a.ldc + col;
        const double v = a.alpha * acc[i][j][r];
        if (a.atomic) {
          atomicAdd(c, v);
        } else {
          const double nv = a.beta1 ? *c + v : v;
          *c = nv;
          if (a.row0_out && row == 0) a.row0_out[col] = nv;
        }
      }
}

}  // namespace

// C = alpha * op(A) op(B) + (beta1 ? C : 0).  ksplit > 1 needs beta1 (C must hold the addend).
static int gemm_f64_impl(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn,
                         const double* A2, const double* B2, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                         double alpha, bool beta1, int ksplit, double* row0_out, hipStream_t st);

int gemm_f64(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn, double* C,
             int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha, bool beta1, int ksplit, hipStream_t st) {
  return gemm_f64_impl(A, sam, sak, B, sbk, sbn, nullptr, nullptr, C, ldc, M, N, K, alpha, beta1, ksplit, nullptr, st);
}

// C += alpha * (A B + A2 B2), both pairs with the same strides, in ONE pass over C
int gemm_f64_pair(const double* A, const double* B, const double* A2, const double* B2, int64_t sam, int64_t sak,
                  int64_t sbk, int64_t sbn, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha,
                  double* row0_out, hipStream_t st) {
  return gemm_f64_impl(A, sam, sak, B, sbk, sbn, A2, B2, C, ldc, M, N, K, alpha, true, 1, row0_out, st);
}

static int gemm_f64_impl(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn,
                         const double* A2, const double* B2, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                         double alpha, bool beta1, int ksplit, double* row0_out, hipStream_t st) {
  if (M <= 0 || N <= 0) return PTD_OK;
  GemmF64Args a{};
  a.A = A; a.sam = sam; a.sak = sak;
  a.B = B; a.sbk = sbk; a.sbn = sbn;
  a.A2 = A2; a.B2 = B2;
  a.row0_out = row0_out;
  a.C = C; a.ldc = ldc;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.alpha = alpha; a.beta1 = beta1 ? 1 : 0;
  a.tiles_m = (int)ceil_div(M, DM);
  if (ksplit < 1 || !beta1) ksplit = 1;
  a.kchunk = (int)align_up((size_t)ceil_div(std::max<int64_t>(K, 1), ksplit), DK);
  ksplit = (int)ceil_div(std::max<int64_t>(K, 1), a.kchunk);
  a.atomic = ksplit > 1;
  const bool akc = (sak == 1), bkc = (sbk == 1);
  dim3 grid((unsigned)(a.tiles_m * ceil_div(N, DN)), (unsigned)ksplit);
  if (akc && bkc) hipLaunchKernelGGL((gemm_f64_kernel<true, true>), grid, dim3(256), 0, st, a);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_f64_kernel<true, false>), grid, dim3(256), 0, st, a);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_f64_kernel<false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_f64_kernel<false, false>), grid, dim3(256), 0, st, a);
  PTD_CHECK_LAUNCH("gemm_f64");
  return PTD_OK;
}

}  // namespace ptd
