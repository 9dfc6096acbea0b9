// Probe: cost of a barrier among the 32 workgroups of ONE XCD (counter in that XCD's L2: workgroup-scope atomic,
// polled with L1-bypassing sc0 loads), with and without an 8 KB vector exchanged per barrier, against the same
// barrier at agent scope (sc1).  One 256-thread workgroup per CU (128 KB of LDS each), every workgroup reads its
// XCC id; only XCC 0's take part.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/xcd_barrier_probe.hip -o gpurun_out/xcd_probe && gpurun_out/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_sc0(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
// loads that are served by the XCD's L2, never by this CU's L1: a returning atomic add of zero, written as asm so the
// compiler cannot turn the idempotent read-modify-write into an (L1-cacheable) atomic load
__device__ __forceinline__ unsigned ld_l2(unsigned* p) {
  unsigned v; const unsigned z = 0;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(z) : "memory");
  return v;
}
__device__ __forceinline__ double ld_l2_d(double* p) {
  double v; const unsigned long long z = 0;
  asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(z) : "memory");
  return v;
}
__device__ __forceinline__ double ld_sc0_d(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

struct Ctl {
  unsigned count;      // registration
  unsigned pad0[31];
  unsigned arrive;     // barrier counter (monotonic)
  unsigned pad1[31];
  unsigned fail;
};

template <int MODE>   // 0, 2: L2-local (workgroup-scope arrive, returning-atomic polls), 1: agent scope (sc1)
__device__ __forceinline__ bool xbarrier(Ctl* c, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE != 1) __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spin = 0;
    for (;;) {
      // MODE 0: the poll is an atomic OR of 0 (atomics always execute in L2; an sc0 load may hit a stale L1 line)
      const unsigned v = MODE != 1 ? ld_l2(&c->arrive) : __hip_atomic_load(&c->arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= target) break;
      if (++spin > 100000L) { c->fail = 1; break; }
    }
  }
  __syncthreads();
  return true;
}

template <int MODE, int EXCH>
__global__ __launch_bounds__(256) void probe(Ctl* c, double* vec, int iters, long long* out, int* xcc_of) {
  extern __shared__ char smem[];
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
  if (threadIdx.x == 0) xcc_of[blockIdx.x] = (int)xcc;
  if (xcc != 0) return;
  __shared__ unsigned slot_s;
  if (threadIdx.x == 0) slot_s = __hip_atomic_fetch_add(&c->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned slot = slot_s;
  if (slot >= 32) return;
  // wait for the 32 of this XCD
  if (threadIdx.x == 0) {
    long spin = 0;
    while (__hip_atomic_load(&c->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u) if (++spin > 100000L) { c->fail = 2; break; }
  }
  __syncthreads();
  if (__hip_atomic_load(&c->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  double acc = 0.0;
  const long long t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    if (EXCH) {
      // my 32 doubles of the 1024-vector of this iteration (double buffered by parity)
      double* v = vec + (it & 1) * 1024;
      if (threadIdx.x < 32) v[slot * 32 + threadIdx.x] = (double)(it + slot) + acc * 1e-30;
    }
    xbarrier<MODE>(c, 32u * (unsigned)it);
    if (EXCH) {
      double* v = vec + (it & 1) * 1024;
      double s = 0.0;
      for (int q = threadIdx.x; q < 1024; q += 256)
        s += MODE == 0 ? ld_l2_d(v + q) : __hip_atomic_load(v + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // MODE 2: sc1 loads behind the L2-local barrier
      acc += s;
    }
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[2 * slot] = t1 - t0; out[2 * slot + 1] = (long long)acc; }
  if (EXCH && threadIdx.x == 1) smem[0] = (char)acc;
}

template <int MODE, int EXCH>
void run(const char* name, Ctl* c, double* vec, long long* out, int* xcc_of, int iters) {
  CK(hipMemset(c, 0, sizeof(Ctl)));
  CK(hipMemset(out, 0, 64 * sizeof(long long)));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, EXCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipLaunchKernelGGL((probe<MODE, EXCH>), dim3(256), dim3(256), 128 * 1024, 0, c, vec, iters, out, xcc_of);
  CK(hipDeviceSynchronize());
  Ctl h; CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
  std::vector<long long> o(64); CK(hipMemcpy(o.data(), out, 64 * sizeof(long long), hipMemcpyDeviceToHost));
  std::vector<int> x(256); CK(hipMemcpy(x.data(), xcc_of, 256 * sizeof(int), hipMemcpyDeviceToHost));
  int on0 = 0, modok = 0;
  for (int b = 0; b < 256; ++b) { on0 += x[b] == 0; modok += x[b] == (b & 7); }
  long long mx = 0; for (int s = 0; s < 32; ++s) mx = o[2 * s] > mx ? o[2 * s] : mx;
  // wall_clock64 ticks at 100 MHz
  printf("%-34s registered %u fail %u  workgroups on XCC 0: %d  (xcc == block %% 8 for %d of 256)  %.3f us per barrier  checksum %lld\n",
         name, h.count, h.fail, on0, modok, (double)mx / 100.0 / iters, o[1]);
}

int main() {
  Ctl* c; double* vec; long long* out; int* xcc_of;
  CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&vec, 2048 * 8)); CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&xcc_of, 256 * 4));
  CK(hipMemset(vec, 0, 2048 * 8));
  const int iters = 500; setvbuf(stdout, nullptr, _IONBF, 0);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>("L2-local barrier", c, vec, out, xcc_of, iters);
    run<0, 1>("L2-local barrier + 8 KB exchange", c, vec, out, xcc_of, iters);
    run<2, 1>("L2-local barrier + sc1 loads", c, vec, out, xcc_of, iters);
    run<1, 0>("agent-scope barrier", c, vec, out, xcc_of, iters);
    run<1, 1>("agent-scope barrier + 8 KB exchange", c, vec, out, xcc_of, iters);
  }
  return 0;
}
