// f32 dense products on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact
// f32 fma chain, 157 TFLOP/s peak).
//
//   gemm_f32   C[M,N] = alpha * sum_k A(m,k) B(k,n) (+ bias)          -> ptd_gemm
//   syrk_f32   E[i,j] += scale * sum_t Y[t,i] Y[t,j], i >= j            -> ptd_syrk_accumulate
//
// One 256-thread workgroup owns a 128x128 output tile; its 4 waves each own a
// 64x64 quadrant as 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs per lane).
// The K loop runs in steps of 32: the next step's operands are fetched from HBM
// into registers while the matrix cores consume the current step from LDS
// (register-staged double buffering, one LDS image, two barriers per step).
//
// LDS image of an operand tile is [k][m] (m contiguous): the 32x32x2 MFMA wants
// lane l to hold A[m = l & 31][k = l >> 5], i.e. 32 consecutive words per half
// wave -> conflict free for any pitch.  The pitch is chosen for the WRITE side:
//   k-contiguous operand (x, W of an nn.Linear): coalesced 16-B global loads
//     along k, transposed on the way into LDS with four ds_write_b32; pitch 129
//     spreads a half wave's (8 k-quads x 4 rows) over all 32 banks;
//   m-contiguous operand (Y^T of the covariance product): rows of the tile are
//     contiguous in memory, written with ds_write_b128; pitch 132 keeps 16-B
//     alignment.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace ptd {

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_PITCH_MAX = 132;

enum { EPI_STORE = 0, EPI_ACC_F64 = 1, EPI_ACC_F32 = 2 };

struct GemmF32Args {
  const float* A;
  int64_t sam, sak;
  const float* B;
  int64_t sbk, sbn;
  void* C;
  int64_t ldc;
  int M, N, K;
  float alpha;
  double scale;
  const float* bias;
  int tiles_m;
  int tri;     // 1: blockIdx.x enumerates lower-triangle tiles (ti >= tj)
  int kchunk;  // K range of blockIdx.y is [y*kchunk, min(K, (y+1)*kchunk)); multiple of BK
  int atomic;  // accumulate with atomics (split K)
  int vecA, vecB;
  int64_t cslab;  // EPI_STORE split K: blockIdx.y writes its partial tile to C + y * cslab
  int64_t zsa, zsb, zsc;  // batched products (generic kernel only): blockIdx.z advances A, B, C by these element strides
  int bias_rows;          // bias indexed by the output ROW (an NCHW 1x1 convolution's channel) instead of the column
};

// Fetch this thread's share of a TS x 32 operand tile (TS = 128: four 4-element groups, TS = 64: two).
//   KC  (k contiguous):  element (r, k) at P[r * s + k];   thread -> row idx>>3, k-quad idx&7
//   !KC (r contiguous):  element (r, k) at P[k * s + r];   thread -> k idx / (TS/4), r-quad idx % (TS/4)
template <bool KC, int TS>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ P, int64_t s, int r_lim, int k_lim,
                                           bool vec, int tid, f32x4 (&v)[TS / 32]) {
#pragma unroll
  for (int p = 0; p < TS / 32; ++p) {
    const int idx = tid + 256 * p;
    if (KC) {
      const int r = idx >> 3, k = (idx & 7) * 4;
      const float* q = P + (int64_t)r * s + k;
      if (vec && r < r_lim && k + 3 < k_lim) {
        v[p] = *reinterpret_cast<const f32x4*>(q);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[p][j] = (r < r_lim && k + j < k_lim) ? q[j] : 0.f;
      }
    } else {
      const int k = idx / (TS / 4), r = (idx % (TS / 4)) * 4;
      const float* q = P + (int64_t)k * s + r;
      if (vec && k < k_lim && r + 3 < r_lim) {
        v[p] = *reinterpret_cast<const f32x4*>(q);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[p][j] = (k < k_lim && r + j < r_lim) ? q[j] : 0.f;
      }
    }
  }
}

template <bool KC, int TS>
__device__ __forceinline__ void stash_tile(float* __restrict__ L, int tid, const f32x4 (&v)[TS / 32]) {
  constexpr int S = KC ? 129 : 132;
#pragma unroll
  for (int p = 0; p < TS / 32; ++p) {
    const int idx = tid + 256 * p;
    if (KC) {
      const int r = idx >> 3, k = (idx & 7) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) L[(k + j) * S + r] = v[p][j];
    } else {
      const int k = idx / (TS / 4), r = (idx % (TS / 4)) * 4;
      *reinterpret_cast<f32x4*>(&L[k * S + r]) = v[p];
    }
  }
}

// One TS x TS output tile (ti, tj in units of TS) by the 4 waves of a workgroup: TS = 128 -> 64 x 64
// per wave (2 x 2 MFMA tiles), TS = 64 -> 32 x 32 per wave.
template <bool AKC, bool BKC, int EPI, int TS>
__device__ __forceinline__ void gemm_f32_tile(const GemmF32Args& a, const int ti, const int tj, float* lds) {
  constexpr int MI = TS / 64;
  float* As = lds;
  float* Bs = lds + BK * LDS_PITCH_MAX;
  constexpr int SA = AKC ? 129 : 132;
  constexpr int SB = BKC ? 129 : 132;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  const int m0 = ti * TS, n0 = tj * TS;
  const int kbeg = blockIdx.y * a.kchunk;
  const int kend = min(a.K, kbeg + a.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const float* Ap = a.A + (int64_t)blockIdx.z * a.zsa + (int64_t)m0 * a.sam + (int64_t)kbeg * a.sak;
  const float* Bp = a.B + (int64_t)blockIdx.z * a.zsb + (int64_t)n0 * a.sbn + (int64_t)kbeg * a.sbk;
  const int64_t sa = AKC ? a.sam : a.sak;  // stride of the non-contiguous index
  const int64_t sb = BKC ? a.sbn : a.sbk;
  const int64_t astep = (int64_t)BK * a.sak, bstep = (int64_t)BK * a.sbk;
  const int m_lim = a.M - m0, n_lim = a.N - n0;

  f32x16 acc[MI][MI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[TS / 32], rb[TS / 32];
  if (nk > 0) {
    fetch_tile<AKC, TS>(Ap, sa, m_lim, kend - kbeg, a.vecA, tid, ra);
    fetch_tile<BKC, TS>(Bp, sb, n_lim, kend - kbeg, a.vecB, tid, rb);
    stash_tile<AKC, TS>(As, tid, ra);
    stash_tile<BKC, TS>(Bs, tid, rb);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      const int k_lim = kend - kbeg - (kt + 1) * BK;
      fetch_tile<AKC, TS>(Ap + (kt + 1) * astep, sa, m_lim, k_lim, a.vecA, tid, ra);
      fetch_tile<BKC, TS>(Bp + (kt + 1) * bstep, sb, n_lim, k_lim, a.vecB, tid, rb);
    }
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float* ar = As + (kk + kh) * SA + wm * (TS / 2) + l31;
      const float* br = Bs + (kk + kh) * SB + wn * (TS / 2) + l31;
      float av[MI], bv[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) { av[i] = ar[32 * i]; bv[i] = br[32 * i]; }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      stash_tile<AKC, TS>(As, tid, ra);
      stash_tile<BKC, TS>(Bs, tid, rb);
      __syncthreads();
    }
  }

  // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int col = n0 + wn * (TS / 2) + j * 32 + (lane & 31);
      if (EPI != EPI_STORE) {
        const int row0 = m0 + wm * (TS / 2) + i * 32 + 4 * (lane >> 5);
        if (EPI == EPI_ACC_F64)
          accumulate_block(reinterpret_cast<double*>(a.C), a.ldc, row0, col, a.M, a.N, a.tri != 0, a.atomic != 0, a.scale, acc[i][j]);
        else
          accumulate_block(reinterpret_cast<float*>(a.C), a.ldc, row0, col, a.M, a.N, a.tri != 0, a.atomic != 0, a.scale, acc[i][j]);
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (TS / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= a.M || col >= a.N) continue;
        if (a.tri && col > row) continue;
        const float v = acc[i][j][r];
        if (EPI == EPI_STORE) {
          float o = a.alpha * v;
          if (a.bias) o += a.bias[a.bias_rows ? row : col];
          reinterpret_cast<float*>(a.C)[(int64_t)blockIdx.z * a.zsc + (int64_t)blockIdx.y * a.cslab +
                                        (int64_t)row * a.ldc + col] = o;
        } else if (EPI == EPI_ACC_F64) {
          double* e = reinterpret_cast<double*>(a.C) + (int64_t)row * a.ldc + col;
          const double d = a.scale * (double)v;
          if (a.atomic) atomicAdd(e, d); else *e += d;
        } else {
          float* e = reinterpret_cast<float*>(a.C) + (int64_t)row * a.ldc + col;
          const float d = (float)(a.scale * (double)v);
          if (a.atomic) atomicAdd(e, d); else *e += d;
        }
      }
    }
}

template <bool AKC, bool BKC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmF32Args a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * LDS_PITCH_MAX];
  int ti, tj;
  if (a.tri) {
    const int t = blockIdx.x;
    ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    tj = t - ti * (ti + 1) / 2;
  } else {
    ti = blockIdx.x % a.tiles_m;
    tj = blockIdx.x / a.tiles_m;
  }
  gemm_f32_tile<AKC, BKC, EPI, 128>(a, ti, tj, lds);
}

// Covariance product with a two-size tile schedule: the strictly-lower 128 x 128 tiles first, then
// every diagonal tile as its three lower 64 x 64 quarters.  n = 4096 has 528 lower tiles for 512
// resident workgroups (256 CUs x 2): with equal tiles the last 16 run alone after everyone else; the
// small workgroups instead queue behind the large ones and finish inside their shadow.
template <int EPI>
__global__ __launch_bounds__(256, 2) void syrk_f32_mixed_kernel(const GemmF32Args a, const int nbig, const int ndiag) {
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * LDS_PITCH_MAX];
  const int b = blockIdx.x;
  if (b >= nbig && b < nbig + ndiag) {  // diagonal tiles kept whole
    gemm_f32_tile<false, false, EPI, 128>(a, b - nbig, b - nbig, lds);
  } else if (b < nbig) {
    // strictly lower: ti > tj, t = ti (ti - 1) / 2 + tj
    int ti = (int)((sqrtf(8.f * (float)b + 1.f) + 1.f) * 0.5f);
    while (ti * (ti - 1) / 2 > b) --ti;
    while ((ti + 1) * ti / 2 <= b) ++ti;
    const int tj = b - ti * (ti - 1) / 2;
    gemm_f32_tile<false, false, EPI, 128>(a, ti, tj, lds);
  } else {
    const int s = b - nbig - ndiag, d = ndiag + s / 3, q = s % 3;
    const int si = 2 * d + (q > 0), sj = 2 * d + (q > 1);
    if (si * 64 >= a.M) return;
    gemm_f32_tile<false, false, EPI, 64>(a, si, sj, lds);
  }
}

// second pass of the EPI_STORE split K: C = alpha * (slab_0 + slab_1 + ...) + bias, slabs added in
// index order (deterministic), four outputs per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int ksplit, int64_t cslab,
                                                            int M, int N, float alpha, const float* __restrict__ bias,
                                                            float* __restrict__ C, int64_t ldc) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // quad index over M x ceil(N/4)
  const int nq = (N + 3) >> 2;
  const int row = (int)(q / nq), c0 = (int)(q % nq) * 4;
  if (row >= M) return;
  const float* p = slabs + (int64_t)row * N + c0;
  if (c0 + 3 < N && (N & 3) == 0) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(p);
    for (int s = 1; s < ksplit; ++s) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + (int64_t)s * cslab);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float o = alpha * acc[j];
      if (bias) o += bias[c0 + j];
      C[(int64_t)row * ldc + c0 + j] = o;
    }
  } else {
    for (int j = 0; j < 4 && c0 + j < N; ++j) {
      float acc = p[j];
      for (int s = 1; s < ksplit; ++s) acc += p[(int64_t)s * cslab + j];
      float o = alpha * acc;
      if (bias) o += bias[c0 + j];
      C[(int64_t)row * ldc + c0 + j] = o;
    }
  }
}

template <int EPI>
void launch_f32(const GemmF32Args& a, bool akc, bool bkc, dim3 grid, hipStream_t st) {
  if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<true, true, EPI>), grid, dim3(256), 0, st, a);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_f32_kernel<true, false, EPI>), grid, dim3(256), 0, st, a);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<false, true, EPI>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false, EPI>), grid, dim3(256), 0, st, a);
}

}  // namespace

// K split of a product with few output tiles (a skinny operand, e.g. x U with a small rank): the
// partial tiles go to slabs in a caller-provided workspace and a second launch adds them in a fixed
// ---- 256 x 256 tile, 8 waves, LDS-DMA in flight across raw barriers: the schedule of
// gemm_bf16_nt_8ph_kernel (gemm_bf16.hip, hazard analysis there) with 4-byte elements.  For the
// nn.Linear layout (both operands k-contiguous) with M, N % 256 = 0, K % 64 = 0 and >= 192 tiles: the
// layer's own y = x W^T of the rank search.  A K step is 32 deep, so an image row is the same 128
// bytes and the half tiles, the swizzle and the wait counts carry over unchanged; a 16-byte fragment
// is 4 consecutive k of one row and feeds FOUR v_mfma_f32_32x32x2_f32 (lanes 0-31 hold k 0..3 of an
// 8-k group, lanes 32-63 k 4..7; A and B use the same map, and the order of the k sum is free).  A
// phase is 32 MFMAs of 16 passes (2048 cycles) against the same handful of LDS / DMA instructions as
// in the bf16 kernel, so the matrix pipe stays busy.  B fragment first: transposed accumulators, 16-byte
// epilogue writes.  The 128^2 register-staged kernel above reaches 125 TFLOP/s at 4096^3 (two barriers
// and a write pass per K step).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // (m0 is named as clobbered on purpose: the loads set it)
template <bool STAGGER>
__global__ __launch_bounds__(512, 1) void gemm_f32_nt_8ph_kernel(const GemmF32Args a) {
  __shared__ __attribute__((aligned(16))) char lds[8 * 16384];  // slot ((op*2 + d)*2 + h) * 16 KiB: A below 64 KiB, B above
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles_m = a.tiles_m, tiles_n = nwg / tiles_m;
  const int width = 8 * tiles_n, first = (wg / width) * 8;
  const int gsz = min(tiles_m - first, 8);
  const int ti = first + (wg % width) % gsz, tj = (wg % width) / gsz;
  const int m0 = ti * 256, n0 = tj * 256;
  const int nk = a.K / 32;

  // staging: a wave instruction moves 8 rows x 128 B; wave w owns pieces 2w, 2w + 1 of every half tile.  Addresses are a
  // wave-uniform tile base plus ONE 32-bit per-lane byte offset per piece (the scalar-base form of the load, written out
  // with m0 beside it: the builtin's 64-bit per-lane address arithmetic costs registers and VALU work in every phase)
  const int srow = lane >> 3, spos = lane & 7;
  const int sr0 = wid * 16 + srow, sr1 = sr0 + 8;
  const unsigned voa0 = (unsigned)((sr0 * a.sam + (spos ^ ((sr0 >> 1) & 7)) * 4) * 4);
  const unsigned voa1 = (unsigned)((sr1 * a.sam + (spos ^ ((sr1 >> 1) & 7)) * 4) * 4);
  const unsigned vob0 = (unsigned)((sr0 * a.sbn + (spos ^ ((sr0 >> 1) & 7)) * 4) * 4);
  const unsigned vob1 = (unsigned)((sr1 * a.sbn + (spos ^ ((sr1 >> 1) & 7)) * 4) * 4);
  const char* const baseA = reinterpret_cast<const char*>(a.A + (int64_t)m0 * a.sam);
  const char* const baseB = reinterpret_cast<const char*>(a.B + (int64_t)n0 * a.sbn);
  const int64_t halfA = 512 * a.sam, halfB = 512 * a.sbn;   // bytes
  const unsigned mypiece = (unsigned)(size_t)(lds_void_t*)lds + wid * 2048;
#define PTD_DMA(LDSADDR, VOFF, SBASE)                                                                    \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                          \
               : : "s"(LDSADDR), "v"(VOFF), "s"(SBASE) : "memory", "m0")
#define PTD_STAGE(D, OP, H, KT)                                                                          \
  do {                                                                                                   \
    const unsigned slot_ = mypiece + ((((OP) * 2 + (D)) * 2 + (H)) << 14);                               \
    const char* base_ = ((OP) ? baseB + (H) * halfB : baseA + (H) * halfA) + (int64_t)(KT) * 128;        \
    PTD_DMA(slot_, ((OP) ? vob0 : voa0), base_);                                                         \
    PTD_DMA(slot_ + 1024, ((OP) ? vob1 : voa1), base_);                                                  \
  } while (0)

  const int fr = lane & 31, fh = lane >> 5, sw = (fr >> 1) & 7;
  int offA[4], offB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int ch = ((2 * ks + fh) ^ sw) << 4;
    offA[ks] = (wr * 64 + fr) * 128 + ch;
    offB[ks] = 65536 + (wc * 32 + fr) * 128 + ch;
  }
  f32x4 af[2][4], b0[4], b1[4];
  f32x16 acc[2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][mt][r] = 0.f;

#define PTD_READ_A(D, H)                                                                                 \
  _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_) _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) \
      af[mt_][ks_] = *reinterpret_cast<const f32x4*>(lds + (((D) * 2 + (H)) << 14) + mt_ * 4096 + offA[ks_])
#define PTD_READ_B(D, H, DST)                                                                            \
  _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                     \
      DST[ks_] = *reinterpret_cast<const f32x4*>(lds + (((D) * 2 + (H)) << 14) + offB[ks_])
#define PTD_QUAD(I, J, BREG)                                                                             \
  do {                                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)  \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_)                                              \
            acc[I][J][mt_] = __builtin_amdgcn_mfma_f32_32x32x2f32(BREG[ks_][e_], af[mt_][ks_][e_], acc[I][J][mt_], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  } while (0)
#define PTD_SYNC_IN(WAIT)                                                                                \
  do {                                                                                                   \
    asm volatile("s_waitcnt vmcnt(" #WAIT ")" ::: "memory");                                             \
    __builtin_amdgcn_s_barrier();                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
  } while (0)
#define PTD_SYNC_OUT()                                                                                   \
  do {                                                                                                   \
    asm volatile("" ::: "memory");                                                                       \
    __builtin_amdgcn_s_barrier();                                                                        \
  } while (0)

  PTD_STAGE(0, 1, 0, 0); PTD_STAGE(0, 0, 0, 0); PTD_STAGE(0, 1, 1, 0); PTD_STAGE(0, 0, 1, 0);
  PTD_STAGE(1, 1, 0, 1); PTD_STAGE(1, 0, 0, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  int kt = 0;
  for (; kt + 2 < nk; kt += 2) {
    PTD_READ_B(0, 0, b0); PTD_READ_A(0, 0); PTD_STAGE(1, 1, 1, kt + 1); PTD_SYNC_IN(8); PTD_QUAD(0, 0, b0); PTD_SYNC_OUT();
    PTD_READ_B(0, 1, b1);                   PTD_STAGE(1, 0, 1, kt + 1); PTD_SYNC_IN(8); PTD_QUAD(0, 1, b1); PTD_SYNC_OUT();
    PTD_READ_A(0, 1);                       PTD_STAGE(0, 1, 0, kt + 2); PTD_SYNC_IN(8); PTD_QUAD(1, 1, b1); PTD_SYNC_OUT();
                                            PTD_STAGE(0, 0, 0, kt + 2); PTD_SYNC_IN(8); PTD_QUAD(1, 0, b0); PTD_SYNC_OUT();
    PTD_READ_B(1, 0, b0); PTD_READ_A(1, 0); PTD_STAGE(0, 1, 1, kt + 2); PTD_SYNC_IN(8); PTD_QUAD(0, 0, b0); PTD_SYNC_OUT();
    PTD_READ_B(1, 1, b1);                   PTD_STAGE(0, 0, 1, kt + 2); PTD_SYNC_IN(8); PTD_QUAD(0, 1, b1); PTD_SYNC_OUT();
    PTD_READ_A(1, 1);                       PTD_STAGE(1, 1, 0, kt + 3); PTD_SYNC_IN(8); PTD_QUAD(1, 1, b1); PTD_SYNC_OUT();
                                            PTD_STAGE(1, 0, 0, kt + 3); PTD_SYNC_IN(8); PTD_QUAD(1, 0, b0); PTD_SYNC_OUT();
  }
  {
    PTD_READ_B(0, 0, b0); PTD_READ_A(0, 0); PTD_STAGE(1, 1, 1, kt + 1); PTD_SYNC_IN(8); PTD_QUAD(0, 0, b0); PTD_SYNC_OUT();
    PTD_READ_B(0, 1, b1);                   PTD_STAGE(1, 0, 1, kt + 1); PTD_SYNC_IN(8); PTD_QUAD(0, 1, b1); PTD_SYNC_OUT();
    PTD_READ_A(0, 1);                                                   PTD_SYNC_IN(6); PTD_QUAD(1, 1, b1); PTD_SYNC_OUT();
                                                                        PTD_SYNC_IN(4); PTD_QUAD(1, 0, b0); PTD_SYNC_OUT();
    PTD_READ_B(1, 0, b0); PTD_READ_A(1, 0);                             PTD_SYNC_IN(2); PTD_QUAD(0, 0, b0); PTD_SYNC_OUT();
    PTD_READ_B(1, 1, b1);                                               PTD_SYNC_IN(0); PTD_QUAD(0, 1, b1); PTD_SYNC_OUT();
    PTD_READ_A(1, 1);                                                   PTD_SYNC_IN(0); PTD_QUAD(1, 1, b1); PTD_SYNC_OUT();
                                                                        PTD_SYNC_IN(0); PTD_QUAD(1, 0, b0); PTD_SYNC_OUT();
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();
#undef PTD_DMA
#undef PTD_STAGE
#undef PTD_READ_A
#undef PTD_READ_B
#undef PTD_QUAD
#undef PTD_SYNC_IN
#undef PTD_SYNC_OUT

  // epilogue: lane l holds output row (l & 31) and four consecutive columns per register group; the four
  // 128 x 128 quadrants leave through an LDS image as 16-byte row-contiguous stores
  constexpr int CP = 128 * 4 + 16;
  static_assert(128 * CP <= 8 * 16384, "the C image must fit the staging buffers");
  float* Cp = static_cast<float*>(a.C);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (i + j) __syncthreads();
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int lr = wr * 64 + mt * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int lc = wc * 32 + 8 * g + 4 * (lane >> 5);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = a.alpha * acc[i][j][mt][4 * g + e] + (a.bias ? a.bias[n0 + j * 128 + lc + e] : 0.f);
          *reinterpret_cast<f32x4*>(lds + lr * CP + lc * 4) = v;
        }
      }
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 128 * 32 / 512; ++p) {
        const int q = tid + 512 * p;
        const int lr = q >> 5, ch = q & 31;
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + lr * CP + ch * 16);
        *reinterpret_cast<f32x4*>(Cp + (int64_t)(m0 + i * 128 + lr) * a.ldc + n0 + j * 128 + ch * 4) = v;
      }
    }
}
#pragma clang diagnostic pop

// K split of the generic kernel for products with few output tiles: the partial tiles are added in index
// order, so the result does not depend on scheduling.  Returns 1 when splitting does not pay.
int gemm_f32_ksplit(int64_t M, int64_t N, int64_t K) {
  // 512 workgroups are resident at once (two per CU); a workgroup keeps at least four K steps (its prologue and
  // its 64 KiB of partial tile are paid once).  T = 1576 rows of a ViT batch against 768 columns are 78 tiles:
  // unsplit, 78 CUs worked through K = 3072 alone (270 us; six ranges of 16 steps: 80).
  const int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
  if (tiles >= 256 || K < 8 * BK) return 1;
  const int64_t ks = std::min<int64_t>(std::min<int64_t>(512 / tiles, K / (4 * BK)), 16);
  return (int)std::max<int64_t>(ks, 1);
}

size_t gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  const int ks = gemm_f32_ksplit(M, N, K);
  return ks > 1 ? (size_t)ks * (size_t)M * (size_t)N * sizeof(float) : 0;
}

int gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
             int64_t ldc, int64_t M, int64_t N, int64_t K, double alpha, const float* bias, void* ws, size_t ws_bytes,
             hipStream_t st) {
  PTD_REQUIRE((sam == 1) != (sak == 1) || (M == 1 || K == 1), "ptd_gemm: exactly one stride of A must be 1");
  PTD_REQUIRE((sbk == 1) != (sbn == 1) || (N == 1 || K == 1), "ptd_gemm: exactly one stride of B must be 1");
  if (M == 0 || N == 0) return PTD_OK;
  GemmF32Args a{};
  a.A = A; a.sam = sam; a.sak = sak;
  a.B = B; a.sbk = sbk; a.sbn = sbn;
  a.C = C; a.ldc = ldc;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.alpha = (float)alpha; a.scale = 1.0; a.bias = bias;
  a.tiles_m = (int)ceil_div(M, BM);
  a.tri = 0; a.kchunk = (int)align_up((size_t)(K > 0 ? K : 1), BK); a.atomic = 0;
  const bool akc = (sak == 1), bkc = (sbk == 1);
  a.vecA = aligned16(A) && ((akc ? sam : sak) % 4 == 0);
  a.vecB = aligned16(B) && ((bkc ? sbn : sbk) % 4 == 0);
  static const int mode_8ph = getenv("PTD_GEMM_8PH") ? atoi(getenv("PTD_GEMM_8PH")) : 2;  // 0 off, 1 lockstep, 2 staggered
  if (mode_8ph && akc && bkc && a.vecA && a.vecB && aligned16(C) && ldc % 4 == 0 && M % 256 == 0 && N % 256 == 0 &&
      K % 64 == 0 && K >= 128 && (M / 256) * (N / 256) >= 192) {
    a.tiles_m = (int)(M / 256);
    dim3 g8((unsigned)((M / 256) * (N / 256)), 1);
    if (mode_8ph == 1) hipLaunchKernelGGL((gemm_f32_nt_8ph_kernel<false>), g8, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((gemm_f32_nt_8ph_kernel<true>), g8, dim3(512), 0, st, a);
    PTD_CHECK_LAUNCH("gemm_f32 (256x256)");
    return PTD_OK;
  }
  const int tiles = (int)(a.tiles_m * ceil_div(N, BN));
  int ksplit = ws ? gemm_f32_ksplit(M, N, K) : 1;
  if (ksplit > 1 && (size_t)ksplit * (size_t)M * (size_t)N * sizeof(float) > ws_bytes) ksplit = 1;
  if (ksplit > 1) {
    a.kchunk = (int)align_up((size_t)ceil_div(K, ksplit), BK);
    ksplit = (int)ceil_div(K, a.kchunk);
  }
  if (ksplit > 1) {
    a.C = ws; a.ldc = N; a.cslab = (int64_t)M * N;
    a.alpha = 1.f; a.bias = nullptr;
    launch_f32<EPI_STORE>(a, akc, bkc, dim3((unsigned)tiles, (unsigned)ksplit), st);
    const int64_t quads = M * ceil_div(N, 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div(quads, 256)), dim3(256), 0, st,
                       static_cast<const float*>(ws), ksplit, a.cslab, (int)M, (int)N, (float)alpha, bias, C, ldc);
  } else {
    launch_f32<EPI_STORE>(a, akc, bkc, dim3((unsigned)tiles, 1), st);
  }
  PTD_CHECK_LAUNCH("gemm_f32");
  return PTD_OK;
}

// `batch` independent products C_z = alpha A_z B_z (+ bias per output row) with A, B, C advanced by the
// element strides zsa, zsb, zsc (0 = shared operand) per z: the NCHW 1x1-convolution pair, whose per-image
// operand x[b] is a [C, H W] matrix with the pixels contiguous.
int gemm_f32_batched(const float* A, int64_t sam, int64_t sak, int64_t zsa, const float* B, int64_t sbk, int64_t sbn,
                     int64_t zsb, float* C, int64_t ldc, int64_t zsc, int64_t M, int64_t N, int64_t K, int64_t batch,
                     double alpha, const float* bias_rows, hipStream_t st) {
  PTD_REQUIRE((sam == 1) != (sak == 1) || (M == 1 || K == 1), "ptd_gemm: exactly one stride of A must be 1");
  PTD_REQUIRE((sbk == 1) != (sbn == 1) || (N == 1 || K == 1), "ptd_gemm: exactly one stride of B must be 1");
  PTD_REQUIRE(batch >= 0 && batch < 65536, "ptd_gemm: batch out of range");
  if (M == 0 || N == 0 || batch == 0) return PTD_OK;
  GemmF32Args a{};
  a.A = A; a.sam = sam; a.sak = sak; a.zsa = zsa;
  a.B = B; a.sbk = sbk; a.sbn = sbn; a.zsb = zsb;
  a.C = C; a.ldc = ldc; a.zsc = zsc;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.alpha = (float)alpha; a.scale = 1.0; a.bias = bias_rows; a.bias_rows = 1;
  a.tiles_m = (int)ceil_div(M, BM);
  a.kchunk = (int)align_up((size_t)(K > 0 ? K : 1), BK);
  const bool akc = (sak == 1), bkc = (sbk == 1);
  a.vecA = aligned16(A) && ((akc ? sam : sak) % 4 == 0) && zsa % 4 == 0;
  a.vecB = aligned16(B) && ((bkc ? sbn : sbk) % 4 == 0) && zsb % 4 == 0;
  launch_f32<EPI_STORE>(a, akc, bkc, dim3((unsigned)(a.tiles_m * ceil_div(N, BN)), 1, (unsigned)batch), st);
  PTD_CHECK_LAUNCH("gemm_f32 (batched)");
  return PTD_OK;
}

int syrk_f32(const float* Y, int64_t T, int64_t n, int64_t ldy, void* E, int64_t ldE, bool e_f64, double scale,
             hipStream_t st) {
  if (n == 0 || T == 0) return PTD_OK;
  GemmF32Args a{};
  a.A = Y; a.sam = 1; a.sak = ldy;
  a.B = Y; a.sbk = ldy; a.sbn = 1;
  a.C = E; a.ldc = ldE;
  a.M = (int)n; a.N = (int)n; a.K = (int)T;
  a.alpha = 1.f; a.scale = scale; a.bias = nullptr;
  const int nt = (int)ceil_div(n, BM);
  a.tiles_m = nt;
  a.tri = 1;
  const int tiles = nt * (nt + 1) / 2;
  // few output tiles (small n): split the token axis so the launch still fills 256 CUs
  int ksplit = 1;
  if (tiles < 192) {
    ksplit = (int)std::min<int64_t>(ceil_div(512, tiles), ceil_div(T, 4 * BK));
    if (ksplit < 1) ksplit = 1;
  }
  a.kchunk = (int)align_up((size_t)ceil_div(T, ksplit), BK);
  ksplit = (int)ceil_div(T, a.kchunk);
  a.atomic = ksplit > 1;
  a.vecA = a.vecB = aligned16(Y) && (ldy % 4 == 0);
  if (ksplit == 1 && !getenv("PTD_SYRK_UNIFORM")) {
    // as many diagonal tiles are quartered as exceed a whole number of resident rounds (512 = 256 CUs x 2)
    const int nbig = nt * (nt - 1) / 2;
    const char* pe = getenv("PTD_SYRK_PEEL");
    const int npeel = std::min(nt, pe ? atoi(pe) : tiles % 512);
    const int ndiag = nt - npeel;
    dim3 grid((unsigned)(nbig + ndiag + 3 * npeel), 1);
    if (e_f64) hipLaunchKernelGGL((syrk_f32_mixed_kernel<EPI_ACC_F64>), grid, dim3(256), 0, st, a, nbig, ndiag);
    else hipLaunchKernelGGL((syrk_f32_mixed_kernel<EPI_ACC_F32>), grid, dim3(256), 0, st, a, nbig, ndiag);
    PTD_CHECK_LAUNCH("syrk_f32");
    return PTD_OK;
  }
  dim3 grid((unsigned)tiles, (unsigned)ksplit);
  if (e_f64) hipLaunchKernelGGL((gemm_f32_kernel<false, false, EPI_ACC_F64>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false, EPI_ACC_F32>), grid, dim3(256), 0, st, a);
  PTD_CHECK_LAUNCH("syrk_f32");
  return PTD_OK;
}

}  // namespace ptd
