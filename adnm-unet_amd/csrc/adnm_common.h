// Shared device/host helpers for libadnm_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "../../include/adnm_hip.h"

#define ADNM_WAVE 64

void adnm_set_error(const char* fmt, ...);

// Opt-in per-kernel timing (see core.hip): declare one right before a launch, in its own scope:
//   { ADNM_PROF("kernel_name", stream, algorithmic_bytes); kernel<<<...>>>(...); }
struct AdnmProfScope {
  AdnmProfScope(const char* name, hipStream_t st, double bytes);
  ~AdnmProfScope();
  hipStream_t st_;
  long idx_;
};
#define ADNM_PROF(name, st, bytes) AdnmProfScope adnm_prof_scope__(name, st, (double)(bytes))

// out[c] = sum_{r<rows} part[r*n + c], c in [0,n), scattered to up to 4 contiguous output segments
// (segment k holds columns [off_k, off_k+len_k); a NULL pointer skips it).  Deterministic tree order.
struct AdnmFoldSeg {
  float* ptr;
  int len;
};
void adnm_launch_fold(const char* prof_name, const float* part, int rows, int n, AdnmFoldSeg s0, AdnmFoldSeg s1, AdnmFoldSeg s2,
                      AdnmFoldSeg s3, hipStream_t st);

// LDS-tiled NT / NN GEMM (lgemm.hip), reached through adnm_skgemm.  b_oc: second operand contiguous along the output axis (op NN);
// nbs: cross-workgroup split of the reduction (combined inside the launch).  adnm_take_tickets (skgemm.hip): `n` zeroed arrival
// counters from the per-device ring, NULL on failure.
int64_t adnm_lgemm_ws_bytes(int64_t I, int64_t J, int64_t R, int nbs);
int adnm_lgemm_launch(bool b_oc, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, float* c, int64_t ldc, void* ws,
                      int64_t ws_bytes, int64_t I, int64_t J, int64_t R, int nbs, int prec, hipStream_t st);
int* adnm_take_tickets(int n, hipStream_t st);

#define ADNM_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      adnm_set_error(__VA_ARGS__);         \
      return ADNM_EINVAL;                  \
    }                                      \
  } while (0)

#define ADNM_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      adnm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return ADNM_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

// More than 64 KB of dynamic LDS needs hipFuncSetAttribute on the function — per DEVICE (HIP keeps function attributes per
// device, and the reference's nn.DataParallel flow, train.py:99-102, drives several devices from one process).  `done` is the
// call site's own bitmask of devices already opted in; setting the attribute twice is harmless, so a race only repeats the call.
#include <atomic>
static inline int adnm_allow_lds(const void* fn, size_t smem, std::atomic<uint64_t>& done, const char* name) {
  if (smem <= 64 * 1024) return ADNM_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return ADNM_OK;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    adnm_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes on device %d: %s", name, smem, dev, hipGetErrorString(e));
    return ADNM_ELAUNCH;
  }
  done.fetch_or(bit, std::memory_order_relaxed);
  return ADNM_OK;
}
#define ADNM_ALLOW_LDS(kernel, smem, name)                                             \
  do {                                                                                 \
    static std::atomic<uint64_t> lds_done__{0};                                        \
    if (int rc__ = adnm_allow_lds((const void*)(kernel), (smem), lds_done__, name)) return rc__; \
  } while (0)

static inline int64_t adnm_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t adnm_align(int64_t a, int64_t b) { return adnm_cdiv(a, b) * b; }

// ---- storage types -------------------------------------------------------------------------
struct bf16x4 {
  uint16_t v[4];
};

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  __hip_bfloat16 b = __float2bfloat16(f);  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return *reinterpret_cast<uint16_t*>(&b);
}

template <typename T>
struct Io;
template <>
struct Io<float> {
  static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <>
struct Io<uint16_t> {  // bf16 storage
  static __device__ __forceinline__ float4 ld4(const uint16_t* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
  }
  static __device__ __forceinline__ void st4(uint16_t* p, float4 v) {
    uint2 r;
    r.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    r.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(p) = r;
  }
  static __device__ __forceinline__ float ld(const uint16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_bf16(v); }
};

// ---- MFMA precision of the GEMM-shaped kernels (tsgemm, skgemm, conv3): ADNM_MFMA_F32 = v_mfma_f32_16x16x4_f32 (exact fp32, an fmaf
// chain); ADNM_MFMA_BF16 = operands rounded to bf16 (RNE) on the way into v_mfma_f32_16x16x16_bf16, fp32 accumulation — the arithmetic
// of BASELINE's bf16 configs at 8x the matrix rate.  Lane (i, kq) of a 16x16x16 bf16 operand holds k = 4*kq .. 4*kq+3: exactly the
// four consecutive reduction steps one float4 fragment load provides, so both precisions share every loader.
using adnm_f32x4 = __attribute__((ext_vector_type(4))) float;
using adnm_bf16x4 = __attribute__((ext_vector_type(4))) short;
__device__ __forceinline__ adnm_bf16x4 adnm_pack_bf16(float a, float b, float c, float d) {
  using bf4 = __attribute__((ext_vector_type(4))) __bf16;
  const adnm_f32x4 v = {a, b, c, d};
  const bf4 h = __builtin_convertvector(v, bf4);   // two v_cvt_pk_bf16_f32
  return __builtin_bit_cast(adnm_bf16x4, h);
}
// acc += A(16 x 16 k-steps) . B for one 16x16 block: a[e], b[e] = the operands of reduction step 4*kq + e of this lane
template <bool BF16>
__device__ __forceinline__ adnm_f32x4 adnm_mfma16(const float (&a)[4], const float (&b)[4], adnm_f32x4 acc) {
  if (BF16) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(adnm_pack_bf16(a[0], a[1], a[2], a[3]), adnm_pack_bf16(b[0], b[1], b[2], b[3]), acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  return acc;
}

// ---- math ----------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_gradf_(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float geluf_(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_gradf_(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
// softplus with torch's threshold (x > 20 -> x).  log1p(e), e = exp(x) > 0, without libm's ~130-instruction log1pf: for e < 1/4 the
// alternating series to e^8 (truncation < 4.3e-7 of the value), above it log(1 + e) on the hardware log (1 + e >= 1.25 keeps the
// rounding of the sum below 1e-7 of the result).  Relative error <= 1e-6 over the whole range.
__device__ __forceinline__ float softplusf_(float x) {
  if (x > 20.0f) return x;
  const float e = __expf(x);
  if (e < 0.25f) {
    const float p = fmaf(e, fmaf(e, fmaf(e, fmaf(e, fmaf(e, fmaf(e, fmaf(e, -0.125f, 1.0f / 7.0f), -1.0f / 6.0f), 0.2f), -0.25f), 1.0f / 3.0f), -0.5f), 1.0f);
    return e * p;
  }
  return __logf(1.0f + e);
}

template <int ACT>
__device__ __forceinline__ float act_fwd(float x) {
  if (ACT == ADNM_ACT_SILU) return siluf_(x);
  if (ACT == ADNM_ACT_GELU) return geluf_(x);
  return x;
}
template <int ACT>
__device__ __forceinline__ float act_grad(float x) {
  if (ACT == ADNM_ACT_SILU) return silu_gradf_(x);
  if (ACT == ADNM_ACT_GELU) return gelu_gradf_(x);
  return 1.0f;
}

// ---- reductions ----------------------------------------------------------------------------
// sum across the lanes whose index differs only in bits [lo_bit, 6): i.e. lanes l, l^lo, l^2lo, ...
__device__ __forceinline__ float wave_sum_from(float v, int lo) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1)
    if (o >= lo) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_from(v, 1); }
// sum across groups of `width` adjacent lanes (width power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1)
    if (o < width) v += __shfl_xor(v, o, 64);
  return v;
}
