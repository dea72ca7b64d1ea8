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
// nbs: cross-workgroup split of the reduction (combined inside the launch).  A split launch's workspaces are the CALLER's: ws = ordinary
// device memory [arrival counters: one int per output tile, zero when idle, rounded up to 256 B | slabs]; slabs_uc: optional uncached
// device memory (adnm_uncached_alloc) for the slabs alone — a slab store is then at the device-wide coherence point once it has
// completed and the ticket needs no fences (ws only has to hold the counters).
int64_t adnm_lgemm_ws_bytes(int64_t I, int64_t J, int64_t R, int nbs);
int adnm_lgemm_launch(bool b_oc, const float* a, int64_t lda, const void* b, int64_t ldb, int b_dtype, const float* b_scale, const float* bias, float* c,
                      int64_t ldc, void* ws, int64_t ws_bytes, void* slabs_uc, int64_t slabs_uc_bytes, int64_t I, int64_t J, int64_t R, int nbs, int prec,
                      float* q, hipStream_t st);
static inline int64_t adnm_ticket_bytes(int64_t ntiles) { return (ntiles * 4 + 255) / 256 * 256; }

// Deferred LEAF launches (core.hip, include/adnm_hip.h: adnm_leafq_*).  A weight-gradient kernel is a leaf of the backward pass: nothing reads
// its result before the optimiser.  While the calling thread has bound a leaf queue (AND a fold queue: the fold of a leaf's partials must
// stay behind it) such an entry point stores its fully prepared launch here instead of making it; adnm_leafq_flush launches the stored
// problems grouped — many per launch — through the per-kind function below.  args: the kernel's argument struct, copied byte for byte.
struct AdnmLeaf {
  int kind, grid, prec;
  const char* prof;
  double bytes;
  alignas(16) unsigned char args[200];
};
enum { ADNM_LEAF_SKGEMM_TN = 0, ADNM_LEAF_DWCONV_WGRAD_K3 = 1, ADNM_LEAF_DWCONV_WGRAD_K5 = 2, ADNM_LEAF_TSGEMM_TN = 3, ADNM_LEAF_KINDS = 4 };
bool adnm_leafq_active();                                                    // would a push be queued?
bool adnm_leafq_push(const AdnmLeaf& leaf);                                  // false: no queue bound on this thread (launch now)
int adnm_skgemm_tn_launch_multi(const AdnmLeaf* const* items, int n, hipStream_t st);   // skgemm.hip
int adnm_dwconv_wgrad_launch_multi(const AdnmLeaf* const* items, int n, int K, hipStream_t st);   // dwconv.hip
int adnm_tsgemm_tn_launch_multi(const AdnmLeaf* const* items, int n, hipStream_t st);   // tsgemm.hip

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
  static __device__ __forceinline__ float rt(float v) { return v; }   // the value a store + load round trip would give back
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
  static __device__ __forceinline__ float rt(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// ---- storage of a GEMM's WEIGHT operand (include/adnm_hip.h: b_dtype of adnm_skgemm): the fp32 master values, or the narrow shadow copy
// the optimiser pass keeps beside them — bf16 (ADNM_MFMA_BF16) or per-tensor scaled OCP e4m3 (the fp8 modes).  4 consecutive elements at
// element offset `off` -> floats: the loaders of the GEMM kernels are otherwise unchanged (same lane -> k map), only the bytes shrink.  The
// decoded values round-trip exactly through the fragment builders (a bf16 / e4m3 value is a fixed point of its own rounding).
template <int BT>
__device__ __forceinline__ float4 adnm_ldb4(const void* base, int64_t off) {
  if constexpr (BT == ADNM_B_BF16) {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + off);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
  } else if constexpr (BT == ADNM_B_FP8) {
    using f2 = __attribute__((ext_vector_type(2))) float;
    const int r = *reinterpret_cast<const int*>(reinterpret_cast<const uint8_t*>(base) + off);
    const f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(r, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(r, true);
    return make_float4(lo[0], lo[1], hi[0], hi[1]);
  } else {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
  }
}

// ---- MFMA precision ladder of the GEMM-shaped kernels (tsgemm, skgemm / lgemm, conv3) — all on gfx950's own instructions:
//   ADNM_MFMA_F32  = v_mfma_f32_16x16x4_f32   (exact fp32: an fmaf chain; the bit-level parity path);
//   ADNM_MFMA_BF16 = v_mfma_f32_16x16x32_bf16 (operands rounded to bf16, RNE, on the way in; fp32 accumulation) — BASELINE's bf16 configs;
//   ADNM_MFMA_FP8  = v_mfma_f32_16x16x32_{fp8,bf8}_fp8 (config 5): per-tensor scaled OCP e4m3 for activations and weights, e5m2 ("bf8")
//                    for gradients, saturating, fp32 accumulation; the accumulator is multiplied by 1 / (scale_a * scale_b) on the way out.
// One MFMA step = 32 reduction steps = 8 per lane.  Every kernel feeds BOTH operands of a step through the same lane -> k map (a dot
// product does not care which k a lane holds as long as the two operands agree), so a lane's 8 values are simply "two float4 fragment
// loads" in all three precisions and the loaders are shared.
using adnm_f32x4 = __attribute__((ext_vector_type(4))) float;
using adnm_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// Quantisation record of one GEMM call site (device memory, 8 floats = 32 B; layout documented in include/adnm_hip.h):
// scale_a multiplies the first ("activation-like") operand, scale_b the weight; amax_* collect max |value| of what the launch read
// (atomic max on the bit pattern: non-negative floats order like unsigned ints) when record != 0.
struct AdnmQuant {
  float scale_a, scale_b, amax_a, amax_b, fmax_a, fmax_b, record, pad;
};

__device__ __forceinline__ adnm_bf16x8 adnm_pack_bf16x8(const float (&v)[8]) {
  using f8 = __attribute__((ext_vector_type(8))) float;
  const f8 t = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  return __builtin_convertvector(t, adnm_bf16x8);   // four v_cvt_pk_bf16_f32
}
// 8 values -> 8 OCP fp8 (BF8 = false: e4m3, |x| <= 448; true: e5m2, |x| <= 57344), RNE, saturating (v_med3 first: the conversion
// itself would turn an overflow into NaN).  The values are already multiplied by the tensor's scale.
template <bool BF8>
__device__ __forceinline__ long adnm_pack_f8x8(const float (&v)[8]) {
  constexpr float kMax = BF8 ? 57344.f : 448.f;
  float c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_fmed3f(v[i], kMax, -kMax);
  int lo = 0, hi = 0;
  if (BF8) {
    lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], lo, false), lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_bf8_f32(c[4], c[5], hi, false), hi = __builtin_amdgcn_cvt_pk_bf8_f32(c[6], c[7], hi, true);
  } else {
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false), lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false), hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  }
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}
// acc += P(32 reduction steps) . Q for one 16 x 16 block: p[e], q[e] = the operands of this lane's e-th reduction step of the group.
// P_BF8 / Q_BF8: that operand is a gradient (e5m2) in the fp8 mode.  In the fp8 mode p / q are already scaled.
template <int PREC, bool P_BF8 = false, bool Q_BF8 = false>
__device__ __forceinline__ adnm_f32x4 adnm_mfma32(const float (&p)[8], const float (&q)[8], adnm_f32x4 acc) {
  if constexpr (PREC == ADNM_MFMA_BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(adnm_pack_bf16x8(p), adnm_pack_bf16x8(q), acc, 0, 0, 0);
  } else if constexpr (PREC == ADNM_MFMA_FP8) {
    const long a = adnm_pack_f8x8<P_BF8>(p), b = adnm_pack_f8x8<Q_BF8>(q);
    if constexpr (P_BF8 && Q_BF8) return __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a, b, acc, 0, 0, 0);
    else if constexpr (P_BF8) return __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(a, b, acc, 0, 0, 0);
    else if constexpr (Q_BF8) return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(a, b, acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p[e], q[e], acc, 0, 0, 0);
    return acc;
  }
}
// the same for ONE group of 16 reduction steps (4 per lane): the odd group at the end of a reduction.  The narrow modes run the 32-step
// instruction on a zero upper half (products with zero operands add nothing).
template <int PREC, bool P_BF8 = false, bool Q_BF8 = false>
__device__ __forceinline__ adnm_f32x4 adnm_mfma16(const float (&p)[4], const float (&q)[4], adnm_f32x4 acc) {
  if constexpr (PREC == ADNM_MFMA_F32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p[e], q[e], acc, 0, 0, 0);
    return acc;
  } else {
    const float p8[8] = {p[0], p[1], p[2], p[3], 0.f, 0.f, 0.f, 0.f}, q8[8] = {q[0], q[1], q[2], q[3], 0.f, 0.f, 0.f, 0.f};
    return adnm_mfma32<PREC, P_BF8, Q_BF8>(p8, q8, acc);
  }
}
// A lane's packed operand of one 32-step group: built ONCE per loaded fragment (scale, saturate, convert), used by every MFMA that reads it
template <int PREC>
struct AdnmFrag {
  float v[8];
};
template <>
struct AdnmFrag<ADNM_MFMA_BF16> {
  adnm_bf16x8 v;
};
template <>
struct AdnmFrag<ADNM_MFMA_FP8> {
  long v;
};
// lo = the lane's 4 reduction steps of the group's first half, hi = of its second half (hi == nullptr: an odd last half, zeros)
template <int PREC, bool BF8>
__device__ __forceinline__ AdnmFrag<PREC> adnm_make_frag(const float (&lo)[4], const float* hi, float scale) {
  AdnmFrag<PREC> f;
  float t[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) t[e] = lo[e], t[4 + e] = hi ? hi[e] : 0.f;
  if constexpr (PREC == ADNM_MFMA_BF16) {
    f.v = adnm_pack_bf16x8(t);
  } else if constexpr (PREC == ADNM_MFMA_FP8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] *= scale;
    f.v = adnm_pack_f8x8<BF8>(t);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = t[e];
  }
  return f;
}
// HALF: only the first 16 steps of the group are real (fp32: skip the zero half's four MFMAs; the narrow modes multiply zeros)
template <int PREC, bool P_BF8, bool Q_BF8, bool HALF = false>
__device__ __forceinline__ adnm_f32x4 adnm_mma(const AdnmFrag<PREC>& p, const AdnmFrag<PREC>& q, adnm_f32x4 acc) {
  if constexpr (PREC == ADNM_MFMA_BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.v, q.v, acc, 0, 0, 0);
  } else if constexpr (PREC == ADNM_MFMA_FP8) {
    if constexpr (P_BF8 && Q_BF8) return __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(p.v, q.v, acc, 0, 0, 0);
    else if constexpr (P_BF8) return __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(p.v, q.v, acc, 0, 0, 0);
    else if constexpr (Q_BF8) return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(p.v, q.v, acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(p.v, q.v, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int e = 0; e < (HALF ? 4 : 8); ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p.v[e], q.v[e], acc, 0, 0, 0);
    return acc;
  }
}

__device__ __forceinline__ float adnm_amax4(float m, float a, float b, float c, float d) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(a), fabsf(b))), fmaxf(fabsf(c), fabsf(d)));
}
// wave-wide max of a non-negative value, then ONE atomic max per wave on the record's slot (order-independent: deterministic)
__device__ __forceinline__ void adnm_amax_commit(float* slot, float m) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(m));
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
