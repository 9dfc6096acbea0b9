// Torch-free driver for rocprofv3 counter passes: one Householder tridiagonalisation of a random
// symmetric n x n f64 matrix through the C ABI (ptd_tridiagonalize), so the per-dispatch PMC rows
// of sytrd_symv_kernel can be set beside its algorithmic bytes.
//   hipcc -O2 -o tools/pmc_driver tools/pmc_driver.cpp -Iinclude -Lptdeco_amd -lptdeco_hip -Wl,-rpath,'$ORIGIN/../ptdeco_amd'
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/pmc_driver 4096
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ptdeco_hip.h"

// mode "mfma": the f32 and bf16 covariance SYRK, the f32 layer-output GEMM and the bf16 GEMM at 4096^3, for
//   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "syrk|gemm" -- tools/pmc_driver mfma
static int run_mfma() {
  const int64_t n = 4096;
  std::vector<float> h((size_t)n * n);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((double)(s >> 11) * (2.0 / 9007199254740992.0) - 1.0); }
  std::vector<unsigned short> hb(h.size());
  for (size_t i = 0; i < h.size(); ++i) { unsigned u; memcpy(&u, &h[i], 4); hb[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
  float *X, *W, *Y; double* E; unsigned short *Xb, *Wb, *Yb;
  if (hipMalloc(&X, n * n * 4) || hipMalloc(&W, n * n * 4) || hipMalloc(&Y, n * n * 4) || hipMalloc(&E, n * n * 8) ||
      hipMalloc(&Xb, n * n * 2) || hipMalloc(&Wb, n * n * 2) || hipMalloc(&Yb, n * n * 2)) return 2;
  (void)hipMemcpy(X, h.data(), n * n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, h.data(), n * n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(Xb, hb.data(), n * n * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(Wb, hb.data(), n * n * 2, hipMemcpyHostToDevice);
  (void)hipMemset(E, 0, n * n * 8);
  for (int r = 0; r < 3; ++r) {
    int rc = ptd_syrk_accumulate(X, n, n, n, 0, E, n, 1, 1.0 / n, nullptr);
    rc |= ptd_gemm(X, n, 1, W, 1, n, Y, n, n, n, n, 0, 0, 1.0, nullptr, nullptr);      // y = x W^T, f32
    rc |= ptd_gemm(Xb, n, 1, Wb, 1, n, Yb, n, n, n, n, 2, 2, 1.0, nullptr, nullptr);    // bf16
    rc |= ptd_syrk_accumulate(Xb, n, n, n, 2, E, n, 1, 1.0 / n, nullptr);                // bf16 covariance product
    if (rc) { fprintf(stderr, "mfma mode rc=%d: %s\n", rc, ptd_last_error()); return 1; }
  }
  // the bf16 decomposed forward at rank 256 (BASELINE configs[4]): h = x A^T (16384 x 256 x 4096), y = h B^T (16384 x 4096 x 256)
  const int64_t T = 16384, r = 256;
  unsigned short *Xf, *Af, *Hf, *Bf, *Yf;
  if (hipMalloc(&Xf, T * n * 2) || hipMalloc(&Af, r * n * 2) || hipMalloc(&Hf, T * r * 2) || hipMalloc(&Bf, n * r * 2) ||
      hipMalloc(&Yf, T * n * 2)) return 2;
  for (int64_t off = 0; off < T * n; off += n * n) (void)hipMemcpy(Xf + off, hb.data(), n * n * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(Af, hb.data(), r * n * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(Bf, hb.data(), n * r * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    int rc = ptd_gemm(Xf, n, 1, Af, 1, n, Hf, r, T, r, n, 2, 2, 1.0 / 64, nullptr, nullptr);
    rc |= ptd_gemm(Hf, r, 1, Bf, 1, r, Yf, n, T, n, r, 2, 2, 1.0, nullptr, nullptr);   // alpha 1, no bias: the plain path
    if (rc) { fprintf(stderr, "mfma mode (forward) rc=%d: %s\n", rc, ptd_last_error()); return 1; }
  }
  (void)hipDeviceSynchronize();
  printf("mfma mode done\n");
  return 0;
}

// mode "syrk": the bf16 covariance product (f64 accumulator) at n = T = 4096 and at the Llama-3-8B calibration shapes
// (T = 2048 tokens per step; n = 4096: q / o / down outputs and the shared input moment of gate / up, n = 1024: k / v,
// n = 14336: the output side of gate / up, which the factored route never forms), three launches per shape in this
// order -- tools/pmc_syrk_summary.py groups the dispatches by it
static int run_syrk() {
  const int64_t shapes[4][2] = {{4096, 4096}, {4096, 2048}, {1024, 2048}, {14336, 2048}};
  int64_t maxy = 0, maxn = 0;
  for (auto& sh : shapes) { maxy = sh[0] * sh[1] > maxy ? sh[0] * sh[1] : maxy; maxn = sh[0] > maxn ? sh[0] : maxn; }
  std::vector<unsigned short> hb((size_t)maxy);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  for (auto& v : hb) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const float f = (float)((double)(s >> 11) * (2.0 / 9007199254740992.0) - 1.0);
    unsigned u; memcpy(&u, &f, 4);
    v = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
  }
  unsigned short* Y; double* E;
  if (hipMalloc(&Y, 8 * maxy * 2) || hipMalloc(&E, maxn * maxn * 8)) return 2;      // eight distinct step matrices
  for (int st = 0; st < 8; ++st) (void)hipMemcpy(Y + st * maxy, hb.data(), maxy * 2, hipMemcpyHostToDevice);
  (void)hipMemset(E, 0, maxn * maxn * 8);
  for (auto& sh : shapes) {
    const int64_t n = sh[0], T = sh[1];
    for (int r = 0; r < 3; ++r) {
      const int rc = ptd_syrk_accumulate(Y, T, n, n, PTD_BF16, E, n, PTD_F64, 1.0 / (double)T, nullptr);
      if (rc) { fprintf(stderr, "syrk mode rc=%d: %s\n", rc, ptd_last_error()); return 1; }
    }
  }
  // round 5: the multi-step entry at the calibration shapes -- 8 steps of 2048 rows in one pass over E (eight
  // distinct step matrices)
  for (int q = 1; q < 4; ++q) {
    const int64_t n = shapes[q][0], T = shapes[q][1];
    const void* ys[8];
    for (int st = 0; st < 8; ++st) ys[st] = Y + st * maxy;
    for (int r = 0; r < 3; ++r) {
      const int rc = ptd_syrk_accumulate_multi(ys, 8, T, n, n, PTD_BF16, E, n, PTD_F64, 1.0 / (double)T, nullptr);
      if (rc) { fprintf(stderr, "syrk multi rc=%d: %s\n", rc, ptd_last_error()); return 1; }
    }
  }
  (void)hipDeviceSynchronize();
  printf("syrk mode done\n");
  return 0;
}

// mode "eigh": ptd_eigh_topk(n = 4096, k = 1024, top-k only) on a covariance with a decaying spectrum -- the filtered
// subspace-iteration route -- for counter passes over its dominant kernel:
//   rocprofv3 --pmc FETCH_SIZE --kernel-include-regex gemm_f64_glds --kernel-trace --output-format csv -d out -- tools/pmc_driver eigh
static int run_eigh() {
  const int64_t n = 4096, T = 8192, k = 1024;
  std::vector<float> h((size_t)T * n);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  for (int64_t t = 0; t < T; ++t)
    for (int64_t c = 0; c < n; ++c) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const double u = (double)(s >> 11) * (2.0 / 9007199254740992.0) - 1.0;
      h[t * n + c] = (float)(u * pow(10.0, -2.0 * (double)c / (double)(n - 1)));
    }
  float* Y; double *E, *C, *evals, *evecs; void *ws, *ws2;
  const size_t wsb = ptd_eigh_workspace_bytes(n), ws2b = ptd_cov_finalize_workspace_bytes(n);
  if (hipMalloc(&Y, h.size() * 4) || hipMalloc(&E, n * n * 8) || hipMalloc(&C, n * n * 8) || hipMalloc(&evals, n * 8) ||
      hipMalloc(&evecs, n * k * 8) || hipMalloc(&ws, wsb) || hipMalloc(&ws2, ws2b ? ws2b : 256)) return 2;
  (void)hipMemcpy(Y, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(E, 0, n * n * 8);
  int rc = ptd_syrk_accumulate(Y, T, n, n, PTD_F32, E, n, PTD_F64, 1.0 / (double)T, nullptr);
  rc |= ptd_cov_finalize(E, n, PTD_F64, nullptr, PTD_F64, n, 1.0, 0.01, C, n, ws2, ws2b, nullptr);
  if (rc) { fprintf(stderr, "eigh mode (covariance) rc=%d: %s\n", rc, ptd_last_error()); return 1; }
  for (int rep = 0; rep < 2; ++rep) {
    rc = ptd_eigh_topk(C, n, n, k, 0, evals, evecs, k, ws, wsb, nullptr, nullptr);
    if (rc) { fprintf(stderr, "eigh mode rc=%d: %s\n", rc, ptd_last_error()); return 1; }
  }
  (void)hipDeviceSynchronize();
  std::vector<double> he(n);
  (void)hipMemcpy(he.data(), evals, n * 8, hipMemcpyDeviceToHost);
  printf("eigh mode done: lambda_max %.6e lambda_k %.6e\n", he[n - 1], he[n - k]);
  return 0;
}

// mode "batched [count]": ptd_eigh_topk_batched on `count` (default 2) symmetric matrices of order 4096, k = 2048 -- the unit
// of the headline step's direct lane (blockIdx.y = matrix through the blocked reduction): every column has a SYMV launch
//   rocprofv3 --pmc FETCH_SIZE --kernel-include-regex sytrd_symv --kernel-trace --output-format csv -d out -- tools/pmc_driver batched 2
static int run_batched(int count) {
  const int64_t n = 4096, k = 2048;
  std::vector<double> h((size_t)n * n);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (double)(s >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  };
  std::vector<const double*> As(count);
  std::vector<double*> evals(count), evecs(count);
  for (int b = 0; b < count; ++b) {
    for (int64_t i = 0; i < n; ++i)
      for (int64_t j = 0; j <= i; ++j) {
        const double v = rnd() + (i == j ? (double)n * 0.01 * (1.0 + (double)i / (double)n) : 0.0);
        h[i * n + j] = v;
        h[j * n + i] = v;
      }
    double *A, *w, *v;
    if (hipMalloc(&A, h.size() * 8) || hipMalloc(&w, n * 8) || hipMalloc(&v, n * k * 8)) return 2;
    (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    As[b] = A; evals[b] = w; evecs[b] = v;
  }
  void* ws;
  const size_t wsb = ptd_eigh_batched_workspace_bytes(n, k, count);
  if (hipMalloc(&ws, wsb)) return 2;
  const int rc = ptd_eigh_topk_batched(As.data(), n, count, n, k, 0, evals.data(), evecs.data(), k, ws, wsb, nullptr, nullptr);
  if (rc) { fprintf(stderr, "batched mode rc=%d: %s\n", rc, ptd_last_error()); return 1; }
  (void)hipDeviceSynchronize();
  printf("batched mode done (%d matrices)\n", count);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "batched")) return run_batched(argc > 2 ? atoi(argv[2]) : 2);
  if (argc > 1 && !strcmp(argv[1], "mfma")) return run_mfma();
  if (argc > 1 && !strcmp(argv[1], "eigh")) return run_eigh();
  if (argc > 1 && !strcmp(argv[1], "syrk")) return run_syrk();
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
