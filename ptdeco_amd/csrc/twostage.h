// Interface between eigh_tridiag.hip (solver driver) and eigh_twostage.hip (two-stage reduction).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ptd {

constexpr int TS_BAND = 32;               // bandwidth after stage 1 = panel width = reflector length of stage 2
constexpr int TS_LDBAND = 2 * TS_BAND + 2;  // doubles per band row: [i][k], k = j - i + 2 TS_BAND

struct TwoStagePlan {
  int n, npanels, npos;
  int64_t ld;
  size_t off_zero, zero_bytes;  // region cleared at the start of every reduction
  size_t off_G1, off_G2, off_Z0, off_band, off_prog, off_status;
  size_t off_L1, off_L1inv, off_MT, off_T, off_W0t, off_Xt, off_tau2, off_W2;
  size_t total;
};

bool twostage_supported(int64_t n);
TwoStagePlan twostage_plan(int64_t n, int64_t ld);
// Aw: working copy (n x n, leading dimension ld, both triangles), destroyed; V2: [n][ldv2] reflectors of stage 2;
// d, e: the tridiagonal.  `mid` (optional) is recorded between the two stages.
int twostage_reduce(const TwoStagePlan& p, char* base, double* Aw, double* V2, int64_t ldv2, double* d, double* e,
                    hipEvent_t mid, hipStream_t st);
int twostage_reduce_stages(const TwoStagePlan& p, char* base, double* Aw, double* V2, int64_t ldv2, double* d, double* e,
                           hipEvent_t mid, int stages, hipStream_t st);
int* twostage_status(const TwoStagePlan& p, char* base);  // device: [0] Cholesky breakdown, [1] chase time-out
// Y (n x nvec, eigenvectors of T in columns) <- Q1 Q2 Y; TV: scratch of n * n doubles
int twostage_backtransform(const TwoStagePlan& p, char* base, const double* Aw, const double* V2, int64_t ldv2,
                           double* TV, double* Y, int64_t ldy, int nvec, hipEvent_t mid, hipStream_t st);

}  // namespace ptd
