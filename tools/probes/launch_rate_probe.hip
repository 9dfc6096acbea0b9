// Probe (round 5): what bounds SEVERAL chains of short dependent launches issued from several host threads?
//  (1) which of S streams share a hardware queue: pairwise, two 300-us single-workgroup spin kernels on streams a, b
//      finish in ~300 us when the streams sit on different queues and in ~600 us when they share one;
//  (2) aggregated launch rate of T threads x own stream x L dependent launches of a short kernel (1 or 256 WGs);
//  (3) the same chains captured once as hipGraphs and replayed (instantiate cost, replay wall per launch).
//   hipcc --offload-arch=gfx950 -O3 -pthread tools/probes/launch_rate_probe.hip -o gpurun_out/launch_probe && gpurun_out/launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin_kernel(long long ticks, unsigned* sink) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (sink && threadIdx.x == 0 && ticks < 0) *sink = 1;
}
__global__ void short_kernel(double* x, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] * 0.999 + 1.0;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double pair_time(hipStream_t a, hipStream_t b, long long ticks) {
  CK(hipDeviceSynchronize());
  double t0 = now_us();
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks, (unsigned*)nullptr);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, ticks, (unsigned*)nullptr);
  CK(hipStreamSynchronize(a));
  CK(hipStreamSynchronize(b));
  return now_us() - t0;
}

static void matrix(const char* title, std::vector<hipStream_t>& st, long long ticks) {
  printf("%s: pair wall in us (spin %lld ticks of 100 MHz = %.0f us each)\n", title, ticks, ticks / 100.0);
  int S = (int)st.size();
  for (int a = 0; a < S; ++a) {
    printf("  s%-2d", a);
    for (int b = 0; b < S; ++b) {
      if (b <= a) { printf("      ."); continue; }
      printf(" %6.0f", pair_time(st[a], st[b], ticks));
    }
    printf("\n");
  }
  fflush(stdout);
}

static void chain(hipStream_t s, double* buf, int n, int wgs, int launches) {
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(short_kernel, dim3(wgs), dim3(256), 0, s, buf, n);
}

int main(int argc, char** argv) {
  int S = argc > 1 ? atoi(argv[1]) : 8;
  CK(hipSetDevice(0));
  long long ticks = 30000;   // wall_clock64 runs at 100 MHz: 300 us
  std::vector<hipStream_t> st(S);
  for (int i = 0; i < S; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  pair_time(st[0], st[1], 1000);
  matrix("fresh streams", st, ticks);
  // use them all once, then again
  for (int i = 0; i < S; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[i], 100LL, (unsigned*)nullptr);
  CK(hipDeviceSynchronize());
  matrix("after use", st, ticks);
  // priority streams
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("priority range: least %d greatest %d\n", lo, hi);
  std::vector<hipStream_t> hp(4);
  for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithPriority(&hp[i], hipStreamNonBlocking, hi));
  matrix("4 high-priority streams", hp, ticks);
  std::vector<hipStream_t> mix = {st[0], st[1], hp[0], hp[1], (hipStream_t)0};
  matrix("mix: s0 s1 hp0 hp1 null", mix, ticks);

  // ---- launch rates ----
  const int n = 256 * 256;
  const int L = 20000;
  std::vector<double*> bufs(8);
  for (auto& b : bufs) { CK(hipMalloc(&b, n * sizeof(double))); CK(hipMemset(b, 0, n * sizeof(double))); }
  for (int wgs : {1, 256}) {
    for (int T : {1, 2, 3, 4, 6, 8}) {
      if (T > S) continue;
      CK(hipDeviceSynchronize());
      double t0 = now_us();
      std::vector<std::thread> th;
      std::vector<double> host_done(T);
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] { CK(hipSetDevice(0)); chain(st[t], bufs[t], n, wgs, L); host_done[t] = now_us(); });
      for (auto& x : th) x.join();
      double t_host = now_us() - t0;
      CK(hipDeviceSynchronize());
      double t_all = now_us() - t0;
      printf("eager  wgs %3d threads %d: host issue %.1f ms, all done %.1f ms -> %.2f us per launch per chain, %.2f us aggregated\n",
             wgs, T, t_host / 1e3, t_all / 1e3, t_all / L, t_all / (double(L) * T));
      fflush(stdout);
    }
  }
  // one thread driving T streams round robin (no thread contention in the runtime)
  for (int T : {2, 4}) {
    CK(hipDeviceSynchronize());
    double t0 = now_us();
    for (int i = 0; i < L; ++i)
      for (int t = 0; t < T; ++t) hipLaunchKernelGGL(short_kernel, dim3(256), dim3(256), 0, st[t], bufs[t], n);
    double t_host = now_us() - t0;
    CK(hipDeviceSynchronize());
    double t_all = now_us() - t0;
    printf("eager  wgs 256, ONE thread over %d streams: host issue %.1f ms, all done %.1f ms -> %.2f us aggregated\n", T,
           t_host / 1e3, t_all / 1e3, t_all / (double(L) * T));
  }
  // ---- graphs ----
  const int G = 4096;   // launches per graph
  std::vector<hipGraphExec_t> execs(8);
  for (int t = 0; t < 8 && t < S; ++t) {
    double t0 = now_us();
    hipGraph_t g;
    CK(hipStreamBeginCapture(st[t], hipStreamCaptureModeThreadLocal));
    chain(st[t], bufs[t], n, 256, G);
    CK(hipStreamEndCapture(st[t], &g));
    double t1 = now_us();
    CK(hipGraphInstantiate(&execs[t], g, nullptr, nullptr, 0));
    double t2 = now_us();
    if (t == 0) printf("graph of %d launches: capture %.1f ms, instantiate %.1f ms\n", G, (t1 - t0) / 1e3, (t2 - t1) / 1e3);
    CK(hipGraphDestroy(g));
  }
  for (int T : {1, 2, 3, 4, 6, 8}) {
    if (T > S) continue;
    const int reps = 5;
    CK(hipDeviceSynchronize());
    double t0 = now_us();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] { CK(hipSetDevice(0)); for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(execs[t], st[t])); });
    for (auto& x : th) x.join();
    double t_host = now_us() - t0;
    CK(hipDeviceSynchronize());
    double t_all = now_us() - t0;
    printf("graph  wgs 256 threads %d: host issue %.1f ms, all done %.1f ms -> %.2f us per launch per chain, %.2f us aggregated\n",
           T, t_host / 1e3, t_all / 1e3, t_all / (double(G) * reps), t_all / (double(G) * reps * T));
    fflush(stdout);
  }
  return 0;
}
