// Shared helpers for the gfx950 kernels of libptdeco_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ptdeco_hip.h"

namespace ptd {

void set_error(const char* fmt, ...);

#define PTD_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ptd::set_error(__VA_ARGS__);      \
      return PTD_ERR_INVALID;           \
    }                                   \
  } while (0)

#define PTD_CHECK_LAUNCH(what)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      ptd::set_error("%s: %s", what, hipGetErrorString(e_));                \
      return PTD_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

#define PTD_CHECK_HIP(expr)                                                 \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess) {                                                 \
      ptd::set_error("%s: %s", #expr, hipGetErrorString(e_));               \
      return PTD_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(unsigned short v) {
  return __uint_as_float(static_cast<unsigned int>(v) << 16);
}

// E[row][col] += scale * acc for one 32 x 32 MFMA block (lane l: column col, rows row0 + (r & 3) + 8 (r >> 2) of
// register r), lower triangle only when `tri`.  The 16 old values of a lane are requested together: written as
// `*e += d` per register they are 16 dependent round trips (the compiler may not move a load above the store before
// it), which made the epilogue of a covariance tile as long as its K loop.
template <typename ET>
__device__ __forceinline__ void accumulate_block(ET* __restrict__ E, const int64_t ld, const int row0, const int col,
                                                 const int M, const int N, const bool tri, const bool atomic,
                                                 const double scale, const f32x16& acc) {
  ET old[16];
  if (!atomic) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2);
      const bool ok = row < M && col < N && !(tri && col > row);
      old[r] = ok ? E[(int64_t)row * ld + col] : (ET)0;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2);
    const bool ok = row < M && col < N && !(tri && col > row);
    if (!ok) continue;
    const ET d = (ET)(scale * (double)acc[r]);
    ET* e = E + (int64_t)row * ld + col;
    if (atomic) atomicAdd(e, d); else *e = old[r] + d;
  }
}

}  // namespace ptd
