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

}  // namespace ptd
