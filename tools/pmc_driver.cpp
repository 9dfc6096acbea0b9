// Torch-free driver for rocprofv3 counter passes: one Householder tridiagonalisation of a random
// symmetric n x n f64 matrix through the C ABI (ptd_tridiagonalize), so the per-dispatch PMC rows
// of sytrd_symv_kernel can be set beside its algorithmic bytes.
//   hipcc -O2 -o tools/pmc_driver tools/pmc_driver.cpp -Iinclude -Lptdeco_amd -lptdeco_hip -Wl,-rpath,'$ORIGIN/../ptdeco_amd'
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/pmc_driver 4096
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ptdeco_hip.h"

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
  const int reps = argc > 2 ? atoi(argv[2]) : 1;
  std::vector<double> h((size_t)n * n);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (double)(s >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  };
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j <= i; ++j) {
      const double v = rnd() + (i == j ? (double)n * 0.01 : 0.0);
      h[i * n + j] = v;
      h[j * n + i] = v;
    }
  double *A, *d, *e;
  void* ws;
  const size_t wsb = ptd_tridiagonalize_workspace_bytes(n);
  if (hipMalloc(&A, h.size() * 8) != hipSuccess || hipMalloc(&d, n * 8) != hipSuccess ||
      hipMalloc(&e, n * 8) != hipSuccess || hipMalloc(&ws, wsb) != hipSuccess) {
    fprintf(stderr, "hipMalloc failed\n");
    return 2;
  }
  (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  for (int r = 0; r < reps; ++r) {
    const int rc = ptd_tridiagonalize(A, n, n, d, e, nullptr, ws, wsb, nullptr);
    if (rc != 0) {
      fprintf(stderr, "ptd_tridiagonalize rc=%d: %s\n", rc, ptd_last_error());
      return 1;
    }
  }
  (void)hipDeviceSynchronize();
  std::vector<double> hd(n);
  (void)hipMemcpy(hd.data(), d, n * 8, hipMemcpyDeviceToHost);
  double tr = 0.0, tr0 = 0.0;
  for (int64_t i = 0; i < n; ++i) { tr += hd[i]; tr0 += h[i * n + i]; }
  printf("n=%lld trace(T)=%.9e trace(A)=%.9e\n", (long long)n, tr, tr0);
  return 0;
}
