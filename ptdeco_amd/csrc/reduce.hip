// HBM-bound helpers of the covariance / rank-selection path:
//   cov_finalize   C = sym(E) / steps - mean term, Tikhonov damping on the diagonal
//   colsum         ey += scale * column sums of Y                         (falor Ey)
//   nsr            per-channel noise-to-signal ratio, one pass over x and y
//   sym_kl         symmetric-max KL of two logit matrices, one wave per row
// All accumulate in f64; inputs are f32 or bf16 and are read once, coalesced.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace ptd {

namespace {

template <typename T>
__device__ __forceinline__ double ld(const T* p, int64_t i);
template <>
__device__ __forceinline__ double ld<float>(const float* p, int64_t i) { return (double)p[i]; }
template <>
__device__ __forceinline__ double ld<double>(const double* p, int64_t i) { return p[i]; }
template <>
__device__ __forceinline__ double ld<unsigned short>(const unsigned short* p, int64_t i) {
  return (double)bf16_to_f32(p[i]);
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------ finalize
template <typename TE, typename TY>
__global__ void diag_mean_kernel(const TE* __restrict__ E, int64_t ldE, const TY* __restrict__ ey, int n,
                                 double steps, double damp_factor, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double c = ld(E, (int64_t)i * ldE + i) / steps;
    if (ey) {
      const double m = ld(ey, i) / steps;
      c -= m * m;
    }
    s += c;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    out[0] = damp_factor * (t / (double)n);
  }
}

template <typename TE, typename TY>
__global__ void cov_finalize_kernel(const TE* __restrict__ E, int64_t ldE, const TY* __restrict__ ey, int n,
                                    double steps, const double* __restrict__ damp, double* __restrict__ C,
                                    int64_t ldC) {
  __shared__ double tile[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int hi = max(bi, bj), lo = min(bi, bj);  // source tile in the lower triangle
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) {
    const int i = hi * 32 + rr, j = lo * 32 + tx;
    tile[rr][tx] = (i < n && j < n) ? ld(E, (int64_t)i * ldE + j) : 0.0;
  }
  __syncthreads();
  const double d = damp[0];
  for (int rr = ty; rr < 32; rr += 8) {
    const int i = bi * 32 + rr, j = bj * 32 + tx;
    if (i >= n || j >= n) continue;
    double e;
    if (bi > bj) e = tile[rr][tx];
    else if (bi < bj) e = tile[tx][rr];
    else e = (rr >= tx) ? tile[rr][tx] : tile[tx][rr];
    double c = e / steps;
    if (ey) c -= (ld(ey, i) / steps) * (ld(ey, j) / steps);
    if (i == j) c += d;
    C[(int64_t)i * ldC + j] = c;
  }
}

// -------------------------------------------------------------------- colsum
template <typename TY, typename TO>
__global__ void colsum_kernel(const TY* __restrict__ Y, int64_t T, int n, int64_t ldy, TO* __restrict__ ey,
                              double scale, int rows_per_block) {
  __shared__ double red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = min(T, r0 + rows_per_block);
  double s = 0.0;
  if (col < n)
    for (int64_t r = r0 + ty; r < r1; r += 4) s += ld(Y, r * ldy + col);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < n) {
    const double t = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    atomicAdd(&ey[col], (TO)(scale * t));
  }
}

// ----------------------------------------------------------------------- nsr
// Threads are laid out as Rt row lanes x Ct columns (Ct = min(C, 256), Rt = 256 / Ct) so that
// consecutive threads touch consecutive addresses for every C, including C == 1.
template <typename T>
__global__ __launch_bounds__(256) void nsr_partial_kernel(const T* __restrict__ x, const T* __restrict__ y, int64_t R,
                                                          int64_t C, int Ct, int Rt, int64_t rows_per_chunk,
                                                          double* __restrict__ part, double* __restrict__ out) {
  __shared__ double sm[3][256];
  // (a workspace that was never initialised leaves the final kernel without a last block: the result then stays NaN
  // instead of whatever the caller's buffer held)
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) out[0] = __builtin_nan("");
  const int tid = threadIdx.x;
  const int cl = tid % Ct, rl = tid / Ct;
  const int64_t c = (int64_t)blockIdx.x * Ct + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = min(R, r0 + rows_per_chunk);
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const bool active = rl < Rt && c < C;
  if (active) {
    const double pivot = ld(y, c);  // first row: keeps the variance sum well conditioned
    for (int64_t r = r0 + rl; r < r1; r += Rt) {
      const double yv = ld(y, r * C + c), xv = ld(x, r * C + c);
      const double dy = yv - pivot, dx = xv - yv;
      s1 += dy;
      s2 += dy * dy;
      s3 += dx * dx;
    }
  }
  sm[0][tid] = s1; sm[1][tid] = s2; sm[2][tid] = s3;
  __syncthreads();
  if (active && rl == 0) {
    for (int k = 1; k < Rt; ++k) {
      s1 += sm[0][k * Ct + cl];
      s2 += sm[1][k * Ct + cl];
      s3 += sm[2][k * Ct + cl];
    }
    // partials as [chunk][3][C]: the final pass reads them coalesced
    double* o = part + (int64_t)blockIdx.y * 3 * C + c;
    o[0] = s1; o[C] = s2; o[2 * C] = s3;
  }
}

// The same sums with 16-byte loads (C a multiple of V = 16 / sizeof(T), 16-byte aligned operands): a lane owns V
// consecutive channels, a wave 64 V of them, the four waves of a block are four row lanes; U = 8 rows of x and of y are
// requested per trip (sixteen 16-byte loads in flight per lane) before anything is consumed.  HBM-bound: the one-element-
// per-lane form above keeps too few bytes in flight per CU (C2 logits: 3.4 TB/s).
template <typename T, int V>
__device__ __forceinline__ void nsr_unpack(const uint4& q, double (&v)[V]);
template <>
__device__ __forceinline__ void nsr_unpack<float, 4>(const uint4& q, double (&v)[4]) {
  v[0] = (double)__uint_as_float(q.x); v[1] = (double)__uint_as_float(q.y);
  v[2] = (double)__uint_as_float(q.z); v[3] = (double)__uint_as_float(q.w);
}
template <>
__device__ __forceinline__ void nsr_unpack<unsigned short, 8>(const uint4& q, double (&v)[8]) {
  v[0] = (double)__uint_as_float(q.x << 16); v[1] = (double)__uint_as_float(q.x & 0xFFFF0000u);
  v[2] = (double)__uint_as_float(q.y << 16); v[3] = (double)__uint_as_float(q.y & 0xFFFF0000u);
  v[4] = (double)__uint_as_float(q.z << 16); v[5] = (double)__uint_as_float(q.z & 0xFFFF0000u);
  v[6] = (double)__uint_as_float(q.w << 16); v[7] = (double)__uint_as_float(q.w & 0xFFFF0000u);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void nsr_partial_vec_kernel(const T* __restrict__ x, const T* __restrict__ y, int64_t R,
                                                              int64_t C, int64_t rows_per_chunk,
                                                              double* __restrict__ part, double* __restrict__ out) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) out[0] = __builtin_nan("");   // (see nsr_partial_kernel)
  constexpr int U = 32 / V;   // rows per trip: 8 (f32), 4 (bf16: eight channels a lane, twice the accumulators)
  __shared__ double sm[3][V][64];   // one quantity at a time: [row lane 1..3][channel of the lane][lane]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * V;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = min(R, r0 + rows_per_chunk);
  const bool active = c0 < C;
  double s1[V], s2[V], s3[V], pv[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s1[i] = s2[i] = s3[i] = pv[i] = 0.0;
  if (active) {
    nsr_unpack<T, V>(*reinterpret_cast<const uint4*>(y + c0), pv);   // first row: the pivot of the variance sums
    // full trips: U rows of y and of x requested before anything is consumed, no branch in between (with a guard per
    // row inside the trip the compiler sank every load into its guarded block: one load in flight per lane)
    int64_t r = r0 + w;
    for (; r + 4 * (U - 1) < r1; r += 4 * U) {
      uint4 qy[U], qx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        qy[u] = *reinterpret_cast<const uint4*>(y + (r + 4 * u) * C + c0);
        qx[u] = *reinterpret_cast<const uint4*>(x + (r + 4 * u) * C + c0);
      }
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise pairs each load with its use to save registers)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        double yv[V], xv[V];
        nsr_unpack<T, V>(qy[u], yv);
        nsr_unpack<T, V>(qx[u], xv);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const double dy = yv[i] - pv[i], dx = xv[i] - yv[i];
          s1[i] += dy;
          s2[i] = fma(dy, dy, s2[i]);
          s3[i] = fma(dx, dx, s3[i]);
        }
      }
    }
    for (; r < r1; r += 4) {                                         // the rows of a last, partial trip
      double yv[V], xv[V];
      nsr_unpack<T, V>(*reinterpret_cast<const uint4*>(y + r * C + c0), yv);
      nsr_unpack<T, V>(*reinterpret_cast<const uint4*>(x + r * C + c0), xv);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const double dy = yv[i] - pv[i], dx = xv[i] - yv[i];
        s1[i] += dy;
        s2[i] = fma(dy, dy, s2[i]);
        s3[i] = fma(dx, dx, s3[i]);
      }
    }
  }
  // row lanes 1..3 hand their sums to lane 0, one quantity at a time, added in lane order
  double* o = part + (int64_t)blockIdx.y * 3 * C + c0;
#pragma unroll
  for (int qn = 0; qn < 3; ++qn) {
    double (&s)[V] = qn == 0 ? s1 : (qn == 1 ? s2 : s3);
    if (qn) __syncthreads();
    if (w > 0)
#pragma unroll
      for (int i = 0; i < V; ++i) sm[w - 1][i][lane] = s[i];
    __syncthreads();
    if (w == 0 && active) {
#pragma unroll
      for (int i = 0; i < V; ++i) o[(int64_t)qn * C + i] = ((s[i] + sm[0][i][lane]) + sm[1][i][lane]) + sm[2][i][lane];
    }
  }
}

// 64 channels x 4 chunk lanes per block: the chunk partials of a channel are added in a fixed order (lane q takes
// chunks q, q + 4, ..., then lanes 0 .. 3), the block leaves the sum of its channels' ratios in blocksum[blockIdx.x],
// and the LAST block to finish (ticket counter) adds the block sums in index order and writes the result:
// deterministic, and one launch instead of three (the old per-channel loop over strided partials took 21 us).
// The ticket is zero on entry (ptd_nsr_workspace_init, once per workspace) and zero again on return.
// (Round 4 tried the whole reduction in ONE launch -- the last chunk block of a column tile adding its chunks, the last
// tile adding the tiles: two dependent ticket + agent-acquire + load rounds behind the stream cost 8-13 us against
// 1.7 us of kernel boundary + 4 us of this kernel; with an agent-scope release per block instead of write-through
// stores 0.25 ms; with tickets claimed by compare-and-swap in an uninitialised workspace 0.1-0.6 ms.)
__global__ __launch_bounds__(256) void nsr_final_kernel(const double* __restrict__ part, int64_t R, int64_t C,
                                                        int nchunk, double eps, double* __restrict__ blocksum,
                                                        unsigned int* __restrict__ ticket, double* __restrict__ out) {
  __shared__ double sm[3][4][64];
  __shared__ double red[4];
  __shared__ bool last;
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (c < C)
#pragma unroll 8
    for (int k = q; k < nchunk; k += 4) {
      const double* p = part + (int64_t)k * 3 * C + c;
      s1 += p[0]; s2 += p[C]; s3 += p[2 * C];
    }
  sm[0][q][cl] = s1; sm[1][q][cl] = s2; sm[2][q][cl] = s3;
  __syncthreads();
  double acc = 0.0;
  if (q == 0 && c < C) {
    s1 = (sm[0][0][cl] + sm[0][1][cl]) + (sm[0][2][cl] + sm[0][3][cl]);
    s2 = (sm[1][0][cl] + sm[1][1][cl]) + (sm[1][2][cl] + sm[1][3][cl]);
    s3 = (sm[2][0][cl] + sm[2][1][cl]) + (sm[2][2][cl] + sm[2][3][cl]);
    const double n = (double)R;
    const double var = (s2 - s1 * s1 / n) / (n - 1.0);  // unbiased, like torch.std
    acc = (s3 / n) / (var + eps);
  }
  if (q == 0) {
    acc = wave_sum(acc);
    if (cl == 0) {
      // written through to memory and acknowledged before the ticket is drawn; the last block reads the sums past its
      // caches (device-scope loads).  A __threadfence() here is a write-back of the whole L2 -- behind the layer
      // products of a metric forward it made this 4-us kernel take 16 (rocprofv3, C2 workload).
      // What this relies on: the hand-off of MI355X_MICROARCH.md's table "Hand-offs measured with sc1 loads
      // in place of the acquire", first row, in every cell -- ONE lane of each storing workgroup makes the workgroup's
      // only store (8 bytes, agent scope = sc1, to hipMalloc memory), waits vmcnt(0) for its acknowledgement, then adds
      // to ONE unsharded agent-scope counter; the workgroup whose add came last (the value returned) loads the sums with
      // agent-scope (sc1) loads after its add has returned, its other waves behind the workgroup barrier that lane
      // joins; one workgroup per CU (the launch asks for more than half a CU's LDS).  Measured on gfx950 / ROCm 7.2, not
      // an architectural guarantee: tests/test_kernels_gpu.py::test_nsr_back_to_back_with_blocks_on_every_xcd compares
      // hundreds of alternating calls bit for bit.
      __hip_atomic_store(&blocksum[blockIdx.x], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
  }
  __syncthreads();
  if (!last) return;
  double t = 0.0;
  for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) t += __hip_atomic_load(&blocksum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = wave_sum(t);
  if (cl == 0) red[q] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (double)C;
    *ticket = 0u;   // ready for the next call on this workspace
  }
}

// -------------------------------------------------------------------- sym_kl
// SYM: rows[b] = max(KL(t || s), KL(s || t)) (calc_kl_loss before its mean); otherwise rows[b] = KL(t || s),
// the reference's calc_kl_divergence(q_logits = s, p_logits = t) (losses_primitives.py:48-54).
template <typename T, bool SYM>
__global__ void sym_kl_rows_kernel(const T* __restrict__ s, const T* __restrict__ t, int64_t B, int64_t C,
                                   double* __restrict__ rows) {
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B) return;
  const int lane = threadIdx.x & 63;
  const T* sr = s + row * C;
  const T* tr = t + row * C;
  double ms = -INFINITY, mt = -INFINITY;
  for (int64_t c = lane; c < C; c += 64) {
    ms = fmax(ms, ld(sr, c));
    mt = fmax(mt, ld(tr, c));
  }
  for (int o = 32; o > 0; o >>= 1) {
    ms = fmax(ms, __shfl_xor(ms, o));
    mt = fmax(mt, __shfl_xor(mt, o));
  }
  double zs = 0.0, zt = 0.0;
  for (int64_t c = lane; c < C; c += 64) {
    zs += exp(ld(sr, c) - ms);
    zt += exp(ld(tr, c) - mt);
  }
  zs = wave_sum(zs);
  zt = wave_sum(zt);
  const double ls = ms + log(zs), lt = mt + log(zt);
  double kts = 0.0, kst = 0.0;  // KL(t || s), KL(s || t)
  for (int64_t c = lane; c < C; c += 64) {
    const double a = ld(sr, c) - ls, b = ld(tr, c) - lt;  // log-probabilities
    kts += exp(b) * (b - a);
    kst += exp(a) * (a - b);
  }
  kts = wave_sum(kts);
  kst = wave_sum(kst);
  if (lane == 0) rows[row] = SYM ? fmax(kts, kst) : kts;
}

__global__ void mean_kernel(const double* __restrict__ v, int64_t n, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    out[0] = t / (double)n;
  }
}

struct NsrPlan {
  int Ct, Rt, coltiles, nchunk;
  int64_t rows_per_chunk;
};

// the 16-byte form: `vec` channels per lane, 64 lanes per column tile, four row lanes; chunks of at least 32 rows, about
// 1024 blocks (the partial sums are 24 C bytes per chunk, written once and read once by the final kernel;
// PTD_NSR_BLOCKS overrides the target)
NsrPlan nsr_plan_vec(int64_t R, int64_t C, int vec) {
  NsrPlan p{};
  p.Ct = 64 * vec;
  p.Rt = 4;
  p.coltiles = (int)ceil_div(C, p.Ct);
  // rows of 64 KiB and more (vocabulary-sized logits): the blocks that run side by side should be COLUMN neighbours --
  // together they sweep whole rows -- so few row chunks (2048 x 128256 bf16: 170 us with 251 blocks, 216 with 2008);
  // narrow rows (the 4096-wide C2 logits) want more blocks in flight
  static const int forced = [] { const char* e = getenv("PTD_NSR_BLOCKS"); return e ? std::max(1, atoi(e)) : 0; }();
  const int blocks = forced ? forced : (C * (16 / vec) >= 65536 ? 256 : 1024);
  const int64_t want = std::max<int64_t>(1, blocks / p.coltiles);
  p.nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(want, ceil_div(R, 32)));
  p.rows_per_chunk = ceil_div(R, p.nchunk);
  p.nchunk = (int)ceil_div(R, p.rows_per_chunk);
  return p;
}

NsrPlan nsr_plan(int64_t R, int64_t C) {
  NsrPlan p{};
  p.Ct = (int)std::min<int64_t>(C, 256);
  p.Rt = 256 / p.Ct;
  p.coltiles = (int)ceil_div(C, p.Ct);
  const int64_t want = std::max<int64_t>(1, 1024 / p.coltiles);
  const int64_t min_rows = (int64_t)p.Rt * 16;
  p.nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(want, ceil_div(R, min_rows)));
  p.rows_per_chunk = ceil_div(R, p.nchunk);
  p.nchunk = (int)ceil_div(R, p.rows_per_chunk);
  return p;
}

}  // namespace

size_t cov_finalize_workspace_bytes(int64_t) { return 256; }

template <typename TE, typename TY>
static int cov_finalize_t(const TE* E, int64_t ldE, const TY* ey, int64_t n, double steps, double damp_factor,
                          double* C, int64_t ldC, double* ws, hipStream_t st) {
  hipLaunchKernelGGL((diag_mean_kernel<TE, TY>), dim3(1), dim3(1024), 0, st, E, ldE, ey, (int)n, steps,
                     damp_factor, ws);
  const unsigned nt = (unsigned)ceil_div(n, 32);
  hipLaunchKernelGGL((cov_finalize_kernel<TE, TY>), dim3(nt, nt), dim3(256), 0, st, E, ldE, ey, (int)n, steps, ws,
                     C, ldC);
  PTD_CHECK_LAUNCH("cov_finalize");
  return PTD_OK;
}

int cov_finalize(const void* E, int64_t ldE, int E_dtype, const void* ey, int ey_dtype, int64_t n, double steps,
                 double damp_factor, double* C, int64_t ldC, void* ws, size_t ws_bytes, hipStream_t st) {
  PTD_REQUIRE(E && C && ws && n >= 1 && ldE >= n && ldC >= n && steps > 0, "ptd_cov_finalize: bad argument");
  PTD_REQUIRE(ws_bytes >= 256, "ptd_cov_finalize: workspace too small");
  PTD_REQUIRE(!ey || ey_dtype == E_dtype, "ptd_cov_finalize: ey dtype must match E dtype");
  double* w = static_cast<double*>(ws);
  if (E_dtype == PTD_F64)
    return cov_finalize_t((const double*)E, ldE, (const double*)ey, n, steps, damp_factor, C, ldC, w, st);
  if (E_dtype == PTD_F32)
    return cov_finalize_t((const float*)E, ldE, (const float*)ey, n, steps, damp_factor, C, ldC, w, st);
  set_error("ptd_cov_finalize: E dtype must be f32 or f64");
  return PTD_ERR_UNSUPPORTED;
}

template <typename TY, typename TO>
static int colsum_t(const TY* y, int64_t T, int64_t n, int64_t ldy, TO* ey, double scale, hipStream_t st) {
  const int64_t colt = ceil_div(n, 64);
  int64_t nchunk = std::max<int64_t>(1, std::min<int64_t>(ceil_div(T, 64), ceil_div(1024, colt)));
  const int64_t rpb = ceil_div(T, nchunk);
  nchunk = ceil_div(T, rpb);
  hipLaunchKernelGGL((colsum_kernel<TY, TO>), dim3((unsigned)colt, (unsigned)nchunk), dim3(256), 0, st, y, T,
                     (int)n, ldy, ey, scale, (int)rpb);
  PTD_CHECK_LAUNCH("colsum");
  return PTD_OK;
}

int colsum_accumulate(const void* y, int64_t T, int64_t n, int64_t ldy, int y_dtype, void* ey, int ey_dtype,
                      double scale, hipStream_t st) {
  PTD_REQUIRE(y && ey && T >= 1 && n >= 1 && ldy >= n, "ptd_colsum_accumulate: bad argument");
  if (y_dtype == PTD_F32 && ey_dtype == PTD_F64) return colsum_t((const float*)y, T, n, ldy, (double*)ey, scale, st);
  if (y_dtype == PTD_F32 && ey_dtype == PTD_F32) return colsum_t((const float*)y, T, n, ldy, (float*)ey, scale, st);
  if (y_dtype == PTD_BF16 && ey_dtype == PTD_F64)
    return colsum_t((const unsigned short*)y, T, n, ldy, (double*)ey, scale, st);
  if (y_dtype == PTD_BF16 && ey_dtype == PTD_F32)
    return colsum_t((const unsigned short*)y, T, n, ldy, (float*)ey, scale, st);
  set_error("ptd_colsum_accumulate: unsupported dtype combination");
  return PTD_ERR_UNSUPPORTED;
}

// workspace: [ticket of the final kernel (a fixed 256-byte head: a workspace serves calls of different shapes, and what
// one call uses for partial sums must never be where another looks for its zeroed counter)][one sum per 64 channels]
// [partial sums [chunk][3][C]]
static size_t nsr_ticket_bytes() { return 256; }
size_t nsr_workspace_bytes(int64_t R, int64_t C) {
  // (the dtype is not known here: room for whichever plan has more chunks)
  const int nchunk = std::max(std::max(nsr_plan(R, C).nchunk, nsr_plan_vec(R, C, 4).nchunk), nsr_plan_vec(R, C, 8).nchunk);
  return nsr_ticket_bytes() + align_up((size_t)ceil_div(C, 64) * 8, 256) + align_up((size_t)nchunk * C * 3 * 8, 256);
}

int nsr_workspace_init(void* ws, size_t ws_bytes, hipStream_t st) {
  PTD_REQUIRE(ws && ws_bytes > 0, "ptd_nsr_workspace_init: bad argument");
  PTD_CHECK_HIP(hipMemsetAsync(ws, 0, ws_bytes, st));
  return PTD_OK;
}

int nsr(const void* x, const void* y, int64_t R, int64_t C, int dtype, double eps, double* out, void* ws,
        size_t ws_bytes, hipStream_t st) {
  PTD_REQUIRE(x && y && out && ws && R >= 1 && C >= 1, "ptd_nsr: bad argument");
  if (ws_bytes < nsr_workspace_bytes(R, C)) {
    set_error("ptd_nsr: workspace too small");
    return PTD_ERR_WORKSPACE;
  }
  NsrPlan p = nsr_plan(R, C);
  const int vec = dtype == PTD_F32 ? 4 : (dtype == PTD_BF16 ? 8 : 0);
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  const bool use_vec = vec && C % vec == 0 && C >= 64 && aligned;
  if (use_vec) p = nsr_plan_vec(R, C, vec);
  char* base = static_cast<char*>(ws);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(base);
  double* blocksum = reinterpret_cast<double*>(base + nsr_ticket_bytes());
  double* part = reinterpret_cast<double*>(base + nsr_ticket_bytes() + align_up((size_t)ceil_div(C, 64) * 8, 256));
  dim3 grid((unsigned)p.coltiles, (unsigned)p.nchunk);
  if (use_vec) {
    if (dtype == PTD_F32)
      hipLaunchKernelGGL((nsr_partial_vec_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)x, (const float*)y, R,
                         C, p.rows_per_chunk, part, out);
    else
      hipLaunchKernelGGL((nsr_partial_vec_kernel<unsigned short, 8>), grid, dim3(256), 0, st, (const unsigned short*)x,
                         (const unsigned short*)y, R, C, p.rows_per_chunk, part, out);
  } else if (dtype == PTD_F32) {
    hipLaunchKernelGGL((nsr_partial_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (const float*)y, R, C,
                       p.Ct, p.Rt, p.rows_per_chunk, part, out);
  } else if (dtype == PTD_BF16) {
    hipLaunchKernelGGL((nsr_partial_kernel<unsigned short>), grid, dim3(256), 0, st, (const unsigned short*)x,
                       (const unsigned short*)y, R, C, p.Ct, p.Rt, p.rows_per_chunk, part, out);
  } else if (dtype == PTD_F64) {
    hipLaunchKernelGGL((nsr_partial_kernel<double>), grid, dim3(256), 0, st, (const double*)x, (const double*)y, R,
                       C, p.Ct, p.Rt, p.rows_per_chunk, part, out);
  } else {
    set_error("ptd_nsr: unsupported dtype");
    return PTD_ERR_UNSUPPORTED;
  }
  const unsigned fblocks = (unsigned)ceil_div(C, 64);
  // (81 KiB of dynamic LDS nobody touches: at most ONE of these workgroups per CU, the condition of the hand-off's row)
  static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(nsr_final_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 81 * 1024) == hipSuccess;
  hipLaunchKernelGGL(nsr_final_kernel, dim3(fblocks), dim3(256), lds_ok ? 81 * 1024 : 0, st, part, R, C, p.nchunk, eps,
                     blocksum, ticket, out);
  PTD_CHECK_LAUNCH("nsr");
  return PTD_OK;
}

// ---------------------------------------------------------------- f32 <-> f64 (the f32 face of the eigensolver)
template <typename TS, typename TD>
__global__ void convert_kernel(const TS* __restrict__ src, int64_t lds, TD* __restrict__ dst, int64_t ldd, int64_t rows,
                               int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    dst[r * ldd + c] = (TD)src[r * lds + c];
  }
}

int convert_f32_to_f64(const float* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return PTD_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(4096, ceil_div(rows * cols, 256));
  hipLaunchKernelGGL((convert_kernel<float, double>), dim3(grid), dim3(256), 0, st, src, lds, dst, ldd, rows, cols);
  PTD_CHECK_LAUNCH("convert f32 -> f64");
  return PTD_OK;
}

int convert_f64_to_f32(const double* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return PTD_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(4096, ceil_div(rows * cols, 256));
  hipLaunchKernelGGL((convert_kernel<double, float>), dim3(grid), dim3(256), 0, st, src, lds, dst, ldd, rows, cols);
  PTD_CHECK_LAUNCH("convert f64 -> f32");
  return PTD_OK;
}

size_t sym_kl_workspace_bytes(int64_t B) { return align_up((size_t)std::max<int64_t>(B, 1) * 8, 256); }

int sym_kl(const void* s, const void* t, int64_t B, int64_t C, int dtype, double* out, void* ws, size_t ws_bytes,
           hipStream_t st) {
  PTD_REQUIRE(s && t && out && ws && B >= 1 && C >= 1, "ptd_sym_kl: bad argument");
  if (ws_bytes < sym_kl_workspace_bytes(B)) {
    set_error("ptd_sym_kl: workspace too small");
    return PTD_ERR_WORKSPACE;
  }
  double* rows = static_cast<double*>(ws);
  const unsigned grid = (unsigned)ceil_div(B, 4);
  if (dtype == PTD_F32)
    hipLaunchKernelGGL((sym_kl_rows_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)s,
                       (const float*)t, B, C, rows);
  else if (dtype == PTD_BF16)
    hipLaunchKernelGGL((sym_kl_rows_kernel<unsigned short, true>), dim3(grid), dim3(256), 0, st,
                       (const unsigned short*)s, (const unsigned short*)t, B, C, rows);
  else {
    set_error("ptd_sym_kl: unsupported dtype");
    return PTD_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, st, rows, B, out);
  PTD_CHECK_LAUNCH("sym_kl");
  return PTD_OK;
}

int kl_rows(const void* q, const void* p, int64_t B, int64_t C, int dtype, double* rows, hipStream_t st) {
  PTD_REQUIRE(q && p && rows && B >= 1 && C >= 1, "ptd_kl_rows: bad argument");
  const unsigned grid = (unsigned)ceil_div(B, 4);
  if (dtype == PTD_F32)
    hipLaunchKernelGGL((sym_kl_rows_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)q,
                       (const float*)p, B, C, rows);
  else if (dtype == PTD_BF16)
    hipLaunchKernelGGL((sym_kl_rows_kernel<unsigned short, false>), dim3(grid), dim3(256), 0, st,
                       (const unsigned short*)q, (const unsigned short*)p, B, C, rows);
  else {
    set_error("ptd_kl_rows: unsupported dtype");
    return PTD_ERR_UNSUPPORTED;
  }
  PTD_CHECK_LAUNCH("kl_rows");
  return PTD_OK;
}

}  // namespace ptd
