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
#include <cstdlib>

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
  int64_t slab;  // > 0: split K without atomics -- K range y writes its partial product to C + y * slab (elements)
  int lower;     // 1: tiles that lie entirely above the diagonal are skipped (symmetric results: X^T X)
  int kchunk;
  int tiles_m;
  const double* A2;  // optional second operand pair with the same strides: C gets A B + A2 B2 in one pass
  const double* B2;  // over C (the symmetric rank-2k update V^T W + W^T V)
  double* row0_out;  // optional: the updated first row of C is also written here (the next panel's first column)
  int64_t bstride;   // 64 x 64 kernel only: blockIdx.z is a batch index, every operand pointer moves by z * bstride elements
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
  const int64_t zoff = (int64_t)blockIdx.z * a.bstride;
  const int ti = blockIdx.x % a.tiles_m, tj = blockIdx.x / a.tiles_m;
  const int m0 = ti * DM, n0 = tj * DN;
  const int kbeg = blockIdx.y * a.kchunk;
  const int kend = min(a.K, kbeg + a.kchunk);
  const int nk = (kend - kbeg + DK - 1) / DK;
  const double* Ap = a.A + zoff + (int64_t)m0 * a.sam + (int64_t)kbeg * a.sak;
  const double* Bp = a.B + zoff + (int64_t)n0 * a.sbn + (int64_t)kbeg * a.sbk;
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
    Ap = a.A2 + zoff + (int64_t)m0 * a.sam + (int64_t)kbeg * a.sak;
    Bp = a.B2 + zoff + (int64_t)n0 * a.sbn + (int64_t)kbeg * a.sbk;
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
        double* c = a.C + zoff + (int64_t)blockIdx.y * a.slab + (int64_t)row * a.ldc + col;
        const double v = a.alpha * acc[i][j][r];
        if (a.atomic) {
          atomicAdd(c, v);
        } else {
          const double nv = a.beta1 ? *c + v : v;
          *c = nv;
          if (a.row0_out && row == 0) a.row0_out[zoff + col] = nv;
        }
      }
}


// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA kernel for the large products of the eigensolver (C X, X^T X, X R of the filtered subspace iteration; the
// operands of eigh_factored): B is [K][N] with N contiguous, A is either [M][K] with K contiguous (AM = false) or
// [K][M] with M contiguous (AM = true: A^T B of a row-major A).  v_mfma_f64_16x16x4_f64 takes 64 cycles for 2048
// flop, so a wave needs only one 8-byte LDS read per operand fragment and 16 passes of matrix work: what bounds the
// 64 x 64 kernel above (38 TFLOP/s on 4096 x 4096 x 1280) is not LDS or issue but the bytes it pulls through L2 (a
// 64 x 64 tile re-reads 16 KiB per 16-deep K step for 16 MFMAs a wave) and two full barriers with a register-staged
// write pass per step.  Here: 128 x (16 NT) tile, 4 waves stacked in M (32 rows x 16 NT columns = 2 x NT
// accumulators each), K step 16, both operands staged by global_load_lds (16 B per lane, 1 KiB per instruction) into
// a ring of three buffers: ONE barrier per K step -- wait for the own pieces of step t (those of step t + 1 stay in
// flight: counted vmcnt), raw s_barrier (everyone's pieces of t have landed and everyone has finished reading step
// t - 1), issue the DMA of step t + 2 into the buffer step t - 1 used, multiply step t.  Two workgroups per CU (NT <=
// 5: 78 KiB each) cover each other's barrier stalls.
//   A image (AM = false): [128 rows][16 k] = 128-byte rows; the 16-byte chunks of row r are XOR-swizzled with
//     (r >> 1) & 7 on the SOURCE address, so the 16 rows x 16 B a half wave reads for one fragment (lane: row l & 15,
//     k = kk + (l >> 4)) cover all 64 banks once.
//   B image, and A when AM: [16 k][W] rows of W = 16 NT (or 128) doubles as they lie in memory; a half wave reads 16
//     consecutive doubles of two neighbouring k rows: conflict-free when the row stride is an odd multiple of 128 B,
//     otherwise chunk index ^= 8 (k & 1) on the source address moves odd rows by half a bank row.
constexpr int GM = 128, GK = 16;

template <int NT, bool AM>
struct GldsCfg {
  static constexpr int BN = 16 * NT;
  static constexpr int A_BYTES = GM * GK * 8;                    // 16 KiB either layout
  static constexpr int B_BYTES = GK * BN * 8;
  static constexpr int A_PIECES = A_BYTES / 1024;                // 16: 4 per wave
  static constexpr int B_PIECES = B_BYTES / 1024;                // 2 NT
  static constexpr int B_PER_WAVE = (B_PIECES + 3) / 4;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int PER_WAVE = 4 + B_PER_WAVE;                // DMA instructions a wave issues per stage
  // ring depth: three buffers (the DMA two K steps ahead) where two workgroups still fit a CU, else two
  static constexpr int ST = (3 * STAGE <= 80 * 1024) ? 3 : 2;
};

template <int NT, bool AM>
__global__ __launch_bounds__(256, 2) void gemm_f64_glds_kernel(const GemmF64Args a) {
  using Cfg = GldsCfg<NT, AM>;
  constexpr int BN = Cfg::BN;
  constexpr int ST = Cfg::ST;
  __shared__ __attribute__((aligned(16))) char lds[ST * Cfg::STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // a contiguous run of tiles per XCD (blocks b and b + 8 share one), row-major over (M tile, N tile) inside it
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles_n = a.N / BN;
  int ti, tj;
  if (a.lower) {
    // only the tiles that touch the lower triangle exist in the grid (launch_glds counts them the same way): row ti
    // has min(tiles_n, ceil((ti + 1) GM / BN)) of them.  Launching the full grid and leaving early would hand the
    // last XCD -- the bottom rows -- nearly all of its tiles, and the launch would take as long as the full product.
    int t = wg;
    ti = 0;
    for (;;) {
      const int c = min(tiles_n, ((ti + 1) * GM + BN - 1) / BN);
      if (t < c) break;
      t -= c;
      ++ti;
    }
    tj = t;
  } else {
    ti = wg / tiles_n;
    tj = wg % tiles_n;
  }
  const int m0 = ti * GM, n0 = tj * BN;
  const int kbeg = blockIdx.y * a.kchunk;
  const int kend = min(a.K, kbeg + a.kchunk);
  const int nk = (kend - kbeg) / GK;

  // ---- DMA sources of this lane (advance by one K step per stage)
  const double* srcA[4];
  const double* srcB[Cfg::B_PER_WAVE];
  int64_t stepA, stepB = (int64_t)GK * a.sbk;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = wid + 4 * j;                       // piece: bytes [1024 p, 1024 p + 1024) of the A image
    if (!AM) {
      const int r = 8 * p + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      srcA[j] = a.A + (int64_t)(m0 + r) * a.sam + kbeg + 2 * c;
    } else {
      const int k = p, c = lane ^ (8 * (k & 1));     // row k of 128 doubles = one piece; 64 chunks
      srcA[j] = a.A + (int64_t)(kbeg + k) * a.sak + m0 + 2 * c;
    }
  }
  stepA = AM ? (int64_t)GK * a.sak : (int64_t)GK;
#pragma unroll
  for (int j = 0; j < Cfg::B_PER_WAVE; ++j) {
    int p = wid + 4 * j;
    if (p >= Cfg::B_PIECES) p -= 4;                  // surplus slot: the wave repeats its previous piece (same bytes)
    const int o = p * 1024 + lane * 16;              // byte offset in the B image
    const int k = o / (BN * 8);
    int c = (o % (BN * 8)) >> 4;
    if ((NT & 1) == 0) c ^= 8 * (k & 1);
    srcB[j] = a.B + (int64_t)(kbeg + k) * a.sbk + n0 + 2 * c;
  }
  auto stage = [&](int buf, int kt) {
    char* base = lds + buf * Cfg::STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[j] + kt * stepA),
                                       (__attribute__((address_space(3))) void*)(base + (wid + 4 * j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < Cfg::B_PER_WAVE; ++j) {
      int p = wid + 4 * j;
      if (p >= Cfg::B_PIECES) p -= 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[j] + kt * stepB),
                                       (__attribute__((address_space(3))) void*)(base + Cfg::A_BYTES + p * 1024), 16, 0, 0);
    }
  };

  // ---- fragment addresses
  const int l15 = lane & 15, l4 = lane >> 4;
  int offA[2][4];   // [M sub-tile][k sub-step]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int k = 4 * ks + l4;
      if (!AM) {
        const int r = wid * 32 + t * 16 + l15;
        offA[t][ks] = r * 128 + (((k >> 1) ^ ((r >> 1) & 7)) << 4) + (k & 1) * 8;
      } else {
        const int m = wid * 32 + t * 16 + l15;       // image [k][128]: chunk = m >> 1, swizzled with 8 (k & 1)
        offA[t][ks] = k * 1024 + ((((m >> 1) ^ (8 * (k & 1)))) << 4) + (m & 1) * 8;
      }
    }
  int offB[4];      // column tile 0; tile j adds 128 bytes (16 doubles), before the swizzle
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) offB[ks] = Cfg::A_BYTES + (4 * ks + l4) * (BN * 8);
  const int bsw = ((NT & 1) == 0) ? 8 * (l4 & 1) : 0;   // (k & 1) = l4 & 1 since 4 ks is even

  f64x4 acc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[t][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  if (nk > 0) stage(0, 0);
  if (ST == 3 && nk > 1) stage(1, 1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (ST == 3 && kt + 1 < nk) {
      if (Cfg::PER_WAVE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (Cfg::PER_WAVE == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + ST - 1 < nk) stage(cur == 0 ? ST - 1 : cur - 1, kt + ST - 1);
    const char* base = lds + cur * Cfg::STAGE;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double af[2], bf[NT];
#pragma unroll
      for (int t = 0; t < 2; ++t) af[t] = *reinterpret_cast<const double*>(base + offA[t][ks]);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int chunk = (8 * j + (l15 >> 1)) ^ bsw;
        bf[j] = *reinterpret_cast<const double*>(base + offB[ks] + (chunk << 4) + (l15 & 1) * 8);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t], bf[j], acc[t][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this step's reads are done before the next barrier)
    cur = cur == ST - 1 ? 0 : cur + 1;
  }

  // C/D map: col = lane & 15, row = (lane >> 4) + 4 reg.  With beta1 the old values of a row block are requested
  // together (written as `*c += v` per element they are 8 NT dependent round trips)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    double old[NT][4];
    const bool add = a.beta1 && !a.atomic;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wid * 32 + t * 16 + l4 + 4 * r;
        old[j][r] = add ? a.C[(int64_t)blockIdx.y * a.slab + (int64_t)row * a.ldc + n0 + j * 16 + l15] : 0.0;
      }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wid * 32 + t * 16 + l4 + 4 * r;
        double* c = a.C + (int64_t)blockIdx.y * a.slab + (int64_t)row * a.ldc + n0 + j * 16 + l15;
        const double v = a.alpha * acc[t][j][r];
        if (a.atomic) atomicAdd(c, v);
        else *c = old[j][r] + v;
      }
  }
}

template <int NT, bool AM>
static void launch_glds(const GemmF64Args& a, int ksplit, hipStream_t st) {
  const int tiles_m = a.M / GM, tiles_n = a.N / (16 * NT);
  int tiles = tiles_m * tiles_n;
  if (a.lower) {
    tiles = 0;
    for (int ti = 0; ti < tiles_m; ++ti) tiles += std::min(tiles_n, ((ti + 1) * GM + 16 * NT - 1) / (16 * NT));
  }
  const dim3 grid((unsigned)tiles, (unsigned)ksplit);
  hipLaunchKernelGGL((gemm_f64_glds_kernel<NT, AM>), grid, dim3(256), 0, st, a);
}

// picks the column-tile width (and reports whether the LDS-DMA kernel applies at all)
static int glds_nt(int64_t M, int64_t N, int64_t K, int ksplit) {
  if (M % GM || K % GK || N % 16 || M < GM || N < 64) return 0;
  int best = 0;
  double best_cost = 0.0;
  for (int nt : {8, 6, 5, 4}) {
    if (N % (16 * nt)) continue;
    const int64_t tiles = (M / GM) * (N / (16 * nt)) * ksplit;
    const int64_t rounds = ceil_div(tiles, 512);     // two workgroups per CU
    // time ~ rounds x tile width; narrower tiles re-read A more often (a 10 % handicap per step below 8)
    const double cost = (double)rounds * nt * (1.0 + 0.04 * (8 - nt));
    if (!best || cost < best_cost) { best = nt; best_cost = cost; }
  }
  return best;
}

}  // namespace

// C = alpha * op(A) op(B) + (beta1 ? C : 0).  ksplit > 1 needs beta1 (C must hold the addend).
static int gemm_f64_impl(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn,
                         const double* A2, const double* B2, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                         double alpha, bool beta1, int ksplit, double* row0_out, hipStream_t st, int64_t slab = 0,
                         bool lower_only = false, int batch = 1, int64_t bstride = 0);

int gemm_f64(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn, double* C,
             int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha, bool beta1, int ksplit, hipStream_t st) {
  return gemm_f64_impl(A, sam, sak, B, sbk, sbn, nullptr, nullptr, C, ldc, M, N, K, alpha, beta1, ksplit, nullptr, st);
}

// Split K without atomics: K range y of `ksplit` writes alpha * A(:, range) B(range, :) to C + y * slab; the caller adds
// the slabs in index order (a deterministic sum).  Returns the number of slabs written in *nslabs.
int gemm_f64_slabs(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn, double* C,
                   int64_t ldc, int64_t slab, int64_t M, int64_t N, int64_t K, double alpha, int ksplit, int* nslabs,
                   bool lower_only, hipStream_t st) {
  const int kchunk = (int)align_up((size_t)ceil_div(std::max<int64_t>(K, 1), std::max(ksplit, 1)), DK);
  *nslabs = (int)ceil_div(std::max<int64_t>(K, 1), kchunk);
  return gemm_f64_impl(A, sam, sak, B, sbk, sbn, nullptr, nullptr, C, ldc, M, N, K, alpha, false, -ksplit, nullptr, st,
                       slab, lower_only);
}

// C += alpha * (A B + A2 B2), both pairs with the same strides, in ONE pass over C
// batch > 1: the same update on `batch` operand sets that lie bstride ELEMENTS apart (every pointer moves alike)
int gemm_f64_pair(const double* A, const double* B, const double* A2, const double* B2, int64_t sam, int64_t sak,
                  int64_t sbk, int64_t sbn, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha,
                  double* row0_out, hipStream_t st, int batch, int64_t bstride) {
  return gemm_f64_impl(A, sam, sak, B, sbk, sbn, A2, B2, C, ldc, M, N, K, alpha, true, 1, row0_out, st, 0, false, batch,
                       bstride);
}

static int gemm_f64_impl(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn,
                         const double* A2, const double* B2, double* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                         double alpha, bool beta1, int ksplit, double* row0_out, hipStream_t st, int64_t slab,
                         bool lower_only, int batch, int64_t bstride) {
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
  if (slab > 0) ksplit = -ksplit;          // (slabs: the K split needs no addend in C)
  else if (ksplit < 1 || !beta1) ksplit = 1;
  a.kchunk = (int)align_up((size_t)ceil_div(std::max<int64_t>(K, 1), ksplit), DK);
  ksplit = (int)ceil_div(std::max<int64_t>(K, 1), a.kchunk);
  a.atomic = ksplit > 1 && slab == 0;
  a.slab = slab;
  a.lower = 0;
  a.bstride = batch > 1 ? bstride : 0;
  const bool akc = (sak == 1), bkc = (sbk == 1);
  // the LDS-DMA kernel: B rows N-contiguous, A rows K- or M-contiguous, 16-byte aligned everything
  static const bool no_glds = getenv("PTD_GEMM_F64_NO_GLDS") != nullptr;
  if (!no_glds && batch <= 1 && !A2 && !row0_out && sbn == 1 && (sak == 1 || sam == 1) && M >= 256 && N >= 64 && K >= 64 &&
      a.kchunk % GK == 0 && (sak == 1 ? sam : sak) % 2 == 0 && sbk % 2 == 0 && ldc % 2 == 0 &&
      ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 15) == 0) {
    const int nt = glds_nt(M, N, K, ksplit);
    const bool am = sak != 1;
    if (nt) {
      a.lower = lower_only ? 1 : 0;     // (only this kernel skips; the 64 x 64 kernel computes every tile)
      switch (nt * 2 + (am ? 1 : 0)) {
        case 16: launch_glds<8, false>(a, ksplit, st); break;
        case 17: launch_glds<8, true>(a, ksplit, st); break;
        case 12: launch_glds<6, false>(a, ksplit, st); break;
        case 13: launch_glds<6, true>(a, ksplit, st); break;
        case 10: launch_glds<5, false>(a, ksplit, st); break;
        case 11: launch_glds<5, true>(a, ksplit, st); break;
        case 8: launch_glds<4, false>(a, ksplit, st); break;
        default: launch_glds<4, true>(a, ksplit, st); break;
      }
      PTD_CHECK_LAUNCH("gemm_f64 (lds-dma)");
      return PTD_OK;
    }
  }
  dim3 grid((unsigned)(a.tiles_m * ceil_div(N, DN)), (unsigned)ksplit, (unsigned)std::max(batch, 1));
  if (akc && bkc) hipLaunchKernelGGL((gemm_f64_kernel<true, true>), grid, dim3(256), 0, st, a);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_f64_kernel<true, false>), grid, dim3(256), 0, st, a);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_f64_kernel<false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_f64_kernel<false, false>), grid, dim3(256), 0, st, a);
  PTD_CHECK_LAUNCH("gemm_f64");
  return PTD_OK;
}

}  // namespace ptd
