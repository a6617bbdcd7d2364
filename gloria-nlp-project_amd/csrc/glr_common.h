// Shared device helpers for libglr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/glr.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

#define GLR_WAVE 64

#define GLR_CHECK_LAUNCH()                              \
  do {                                                  \
    if (hipGetLastError() != hipSuccess) return GLR_ELAUNCH; \
  } while (0)

// Per-device launch facts, looked up once per (kernel, device) instead of once per launch: hipFuncSetAttribute and the
// occupancy query cost microseconds of host time each, which a 32-pair training step pays ~250 times.  The tables are
// immutable after their first fill (benign race: two threads may both fill an entry with the same value).
#include <atomic>
constexpr int GLR_MAX_DEV = 16;
inline int glr_cur_dev() {
  int dev = 0;
  return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < GLR_MAX_DEV) ? dev : -1;
}
struct GlrLdsAttr {                 // one object per kernel: largest dynamic-LDS size already granted on each device
  std::atomic<int> have[GLR_MAX_DEV];
  GlrLdsAttr() { for (auto& h : have) h.store(0, std::memory_order_relaxed); }
};
inline int glr_ensure_lds(GlrLdsAttr& a, const void* fn, int lds) {
  const int dev = glr_cur_dev();
  if (dev >= 0 && a.have[dev].load(std::memory_order_relaxed) >= lds) return GLR_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return GLR_ELAUNCH;
  if (dev >= 0) a.have[dev].store(lds, std::memory_order_relaxed);
  return GLR_OK;
}
inline int glr_dev_cus() {          // compute units of the current device
  static std::atomic<int> n_cu[GLR_MAX_DEV];
  const int dev = glr_cur_dev();
  int n = dev >= 0 ? n_cu[dev].load(std::memory_order_relaxed) : 0;
  if (n == 0) {
    int v = 0;
    n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev >= 0 ? dev : 0) == hipSuccess && v > 0) ? v : 256;
    if (dev >= 0) n_cu[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
struct GlrOccupancy {               // one object per kernel: resident workgroups per CU at `threads` threads, no dynamic LDS
  std::atomic<int> occ[GLR_MAX_DEV];
  GlrOccupancy() { for (auto& o : occ) o.store(0, std::memory_order_relaxed); }
  int get(const void* fn, int threads) {
    const int dev = glr_cur_dev();
    int v = dev >= 0 ? occ[dev].load(std::memory_order_relaxed) : 0;
    if (v == 0) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, threads, 0) != hipSuccess || v < 1) v = 2;
      if (dev >= 0) occ[dev].store(v, std::memory_order_relaxed);
    }
    return v;
  }
};

__device__ __forceinline__ unsigned short f2bf(float x) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round to nearest even, NaN preserved)
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x));
}
__device__ __forceinline__ float bf2f(unsigned short x) {
  return __builtin_bit_cast(float, ((unsigned)x) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Operand traits of the two arithmetic modes of the MFMA kernels.
//   F32 : fp32 operands, v_mfma_f32_32x32x2_f32 (exact fp32 fma chain) - the 1e-4 parity mode
//   BF16: bf16 operands, v_mfma_f32_32x32x16_bf16, fp32 accumulate
// A "fragment" is the 16 bytes one lane feeds per k-step: lane (row = lane&31, h = lane>>5)
// holds K elements [8h, 8h+8) (bf16, one MFMA of K=16) or [4h, 4h+4) of an 8-wide k-group
// (fp32, four MFMAs of K=2; the k order inside the group is permuted identically for A and B).
struct OpF32 {
  static constexpr int ESZ = 4;    // bytes per element
  typedef f32x4 frag;
  static __device__ __forceinline__ frag ld(const unsigned char* p) {
    return __builtin_bit_cast(f32x4, *reinterpret_cast<const uint4*>(p));
  }
  static __device__ __forceinline__ void mma(const frag& a, const frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(const unsigned char* p) { return *reinterpret_cast<const float*>(p); }
  static __device__ __forceinline__ void from_f32(unsigned char* p, float v) { *reinterpret_cast<float*>(p) = v; }
};
struct OpBF16 {
  static constexpr int ESZ = 2;
  typedef bf16x8 frag;
  static __device__ __forceinline__ frag ld(const unsigned char* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
  }
  static __device__ __forceinline__ void mma(const frag& a, const frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(const unsigned char* p) {
    return bf2f(*reinterpret_cast<const unsigned short*>(p));
  }
  static __device__ __forceinline__ void from_f32(unsigned char* p, float v) {
    *reinterpret_cast<unsigned short*>(p) = f2bf(v);
  }
};

// generic scalar load / store by dtype code (pack kernels)
__device__ __forceinline__ float ld_any(const void* base, size_t idx, int dtype) {
  return dtype == GLR_F32 ? reinterpret_cast<const float*>(base)[idx]
                          : bf2f(reinterpret_cast<const unsigned short*>(base)[idx]);
}
__device__ __forceinline__ void st_any(void* base, size_t idx, int dtype, float v) {
  if (dtype == GLR_F32) reinterpret_cast<float*>(base)[idx] = v;
  else reinterpret_cast<unsigned short*>(base)[idx] = f2bf(v);
}
