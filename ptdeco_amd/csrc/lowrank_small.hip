// The decomposed layer's forward at SMALL ranks (bf16, r = 32 / 64 / 96 / 128), one launch:
//     y[T, n_o] = (x[T, n_i] A[r, n_i]^T) B[n_o, r]^T (+ bias)            dwain.py:74-85 (two nn.Linear), falor.py:84-95
// A dwain search on a wide layer ends at such ranks (the Llama-3-8B-shaped runs: most gate / up / down layers at 32 of
// 4096), and every later model forward runs those pairs -- at full depth they are ~40 % of the rank search's time.  As two
// products the pair is three launches (x A^T with a K split, the reduction of its slabs, h B^T): 23-29 us at T = 2048
// against an HBM bound of 4-10 us, because a [T, r] intermediate is a handful of tiles for the whole chip.
// Here a workgroup owns 16 rows of x, and both operand layouts are the matrix cores' own: v_mfma_f32_16x16x32_bf16
// takes, per lane, 8 consecutive k of row (lane & 15) -- 16 bytes of a K-contiguous row -- for BOTH operands, so x, A and
// B go from memory into the operand registers directly (no LDS image, no transposition):
//   1. h[16, r] = x A^T: the four waves split K, every wave streams its quarter of the 16 x-rows and of A's r rows
//      (A comes out of L2: every workgroup reads all of it), partial sums meet in LDS in wave order (deterministic);
//      h is rounded to bf16, as the first nn.Linear of the reference pair rounds its output;
//   2. y[16, n_o] = h B^T: the waves walk 64-column chunks of B's rows (r K-contiguous values each: for r = 32 sixteen
//      rows are ONE contiguous KiB), the B fragment is the FIRST operand so that a lane holds four consecutive output
//      columns of one row, which leave through a per-wave LDS patch as 16-byte row-contiguous stores.
// With few row blocks (T / 16 < ~512) the columns of y are split over `nsplit` workgroups per row block, each of which
// repeats step 1 (x and A then come from L2 / Infinity Cache: 0.5 GFLOP and 16 MB at T = 2048, n_i = 4096).
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace ptd {

namespace {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, hw_bf16x2));
}

struct SmallArgs {
  const u16* x; int64_t ldx; int T, n_i;
  const u16* A; int64_t lda;
  const u16* B; int64_t ldb; int n_o;
  const u16* bias;
  u16* y; int64_t ldy;
  int nsplit;
};

template <int R32>   // r = 32 R32
__global__ __launch_bounds__(256) void lowrank_small_kernel(const SmallArgs a) {
  constexpr int R = 32 * R32, JB = R / 16, HP = R + 8;     // HP: bf16 pitch of h (16-byte rows, rotated banks)
  __shared__ __attribute__((aligned(16))) float part[4][16][R + 4];
  __shared__ __attribute__((aligned(16))) u16 hs[16][HP];
  __shared__ __attribute__((aligned(16))) u16 patch[4][16][64 + 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int rb = blockIdx.x / a.nsplit, sp = blockIdx.x % a.nsplit;
  const int m0 = rb * 16;

  // ---- 1. h = x A^T, K split over the waves
  {
    const int kw = a.n_i / 4, k0 = wid * kw;
    const u16* xp = a.x + (int64_t)(m0 + l15) * a.ldx + k0 + 8 * lq;
    const u16* ap = a.A + (int64_t)l15 * a.lda + k0 + 8 * lq;
    f32x4 acc[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;                               // K steps requested together
    const int steps = kw / 32;
    int ks = 0;
    for (; ks + U <= steps; ks += U) {
      s16x8 xf[U], af[U][JB];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xf[u] = *reinterpret_cast<const s16x8*>(xp + 32 * (ks + u));
#pragma unroll
        for (int j = 0; j < JB; ++j) af[u][j] = *reinterpret_cast<const s16x8*>(ap + (int64_t)j * 16 * a.lda + 32 * (ks + u));
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][j], xf[u], acc[j], 0, 0, 0);
    }
    for (; ks < steps; ++ks) {
      const s16x8 xf = *reinterpret_cast<const s16x8*>(xp + 32 * ks);
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const s16x8 af = *reinterpret_cast<const s16x8*>(ap + (int64_t)j * 16 * a.lda + 32 * ks);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf, acc[j], 0, 0, 0);
      }
    }
    // first operand = A's rows: the lane holds h[row l15][columns 16 j + 4 lq .. + 3]
#pragma unroll
    for (int j = 0; j < JB; ++j) *reinterpret_cast<f32x4*>(&part[wid][l15][16 * j + 4 * lq]) = acc[j];
  }
  __syncthreads();
  for (int e = tid; e < 16 * R; e += 256) {
    const int m = e / R, j = e % R;
    const float s = ((part[0][m][j] + part[1][m][j]) + part[2][m][j]) + part[3][m][j];
    hs[m][j] = (u16)(pack2(s, 0.f) & 0xffffu);
  }
  __syncthreads();

  // ---- 2. y = h B^T over this workgroup's 64-column chunks
  s16x8 hf[R32];
#pragma unroll
  for (int ks = 0; ks < R32; ++ks) hf[ks] = *reinterpret_cast<const s16x8*>(&hs[l15][32 * ks + 8 * lq]);
  const int chunks = a.n_o / 64;
  const int cbeg = (int)((int64_t)chunks * sp / a.nsplit), cend = (int)((int64_t)chunks * (sp + 1) / a.nsplit);
  const u16* bp = a.B + (int64_t)l15 * a.ldb + 8 * lq;
  auto load_chunk = [&](int c, s16x8 (&bf)[4][R32]) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int ks = 0; ks < R32; ++ks)
        bf[cb][ks] = *reinterpret_cast<const s16x8*>(bp + (int64_t)(64 * c + 16 * cb) * a.ldb + 32 * ks);
  };
  s16x8 cur[4][R32], nxt[4][R32];
  int c = cbeg + wid;
  if (c < cend) load_chunk(c, cur);
  for (; c < cend; c += 4) {
    const bool more = c + 4 < cend;
    if (more) load_chunk(c + 4, nxt);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < R32; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[cb][ks], hf[ks], acc, 0, 0, 0);
      // first operand = B's rows: the lane holds y[row l15][columns 64 c + 16 cb + 4 lq .. + 3]
      if (a.bias) {
        const u16* bq = a.bias + 64 * c + 16 * cb + 4 * lq;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += bf16_to_f32(bq[e]);
      }
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 pk = {pack2(acc[0], acc[1]), pack2(acc[2], acc[3])};
      *reinterpret_cast<u32x2*>(&patch[wid][l15][16 * cb + 4 * lq]) = pk;
    }
    // the wave's own patch: 16 rows x 128 bytes leave as 16-byte pieces (LDS operations of one wave complete in order)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = (lane >> 3) + 8 * p, ch = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(&patch[wid][row][8 * ch]);
      *reinterpret_cast<uint4*>(a.y + (int64_t)(m0 + row) * a.ldy + 64 * c + 8 * ch) = v;
    }
    if (more) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int ks = 0; ks < R32; ++ks) cur[cb][ks] = nxt[cb][ks];
    }
  }
}

}  // namespace

bool lowrank_small_applies(int64_t T, int64_t n_i, int64_t r, int64_t n_o, int64_t ldx, int64_t lda, int64_t ldb,
                           int64_t ldy, const void* x, const void* A, const void* B, const void* y) {
  static const bool off = getenv("PTD_LOWRANK_SMALL") && atoi(getenv("PTD_LOWRANK_SMALL")) == 0;
  if (off) return false;
  return r % 32 == 0 && r >= 32 && r <= 128 && T % 16 == 0 && T >= 16 && n_i % 128 == 0 && n_o % 64 == 0 && ldx % 8 == 0 &&
         lda % 8 == 0 && ldb % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(A) && aligned16(B) && aligned16(y) &&
         T < (1ll << 31) && n_i < (1ll << 31) && n_o < (1ll << 31);
}

int lowrank_small_bf16(const void* x, int64_t ldx, int64_t T, int64_t n_i, const void* A, int64_t lda, int64_t r,
                       const void* B, int64_t ldb, int64_t n_o, const void* bias, void* y, int64_t ldy, hipStream_t st) {
  SmallArgs a{static_cast<const u16*>(x), ldx, (int)T, (int)n_i, static_cast<const u16*>(A), lda, static_cast<const u16*>(B),
              ldb, (int)n_o, static_cast<const u16*>(bias), static_cast<u16*>(y), ldy, 1};
  const int64_t blocks = T / 16, chunks = n_o / 64;
  // enough workgroups for two rounds of the chip: split the columns when the row blocks are few
  const char* ns = getenv("PTD_LOWRANK_SMALL_NSPLIT");
  int64_t nsplit = ns ? atoi(ns) : ceil_div(768, blocks);
  nsplit = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nsplit, 16), chunks));
  a.nsplit = (int)nsplit;
  const dim3 grid((unsigned)(blocks * nsplit));
  switch (r / 32) {
    case 1: hipLaunchKernelGGL((lowrank_small_kernel<1>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((lowrank_small_kernel<2>), grid, dim3(256), 0, st, a); break;
    case 3: hipLaunchKernelGGL((lowrank_small_kernel<3>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((lowrank_small_kernel<4>), grid, dim3(256), 0, st, a); break;
  }
  PTD_CHECK_LAUNCH("lowrank_small_bf16");
  return PTD_OK;
}

}  // namespace ptd
