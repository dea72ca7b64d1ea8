// Fused step glue over FLAT fp32 buffers: global gradient norm, clip_grad_norm_ scaling
// (train.py:140) and the AdamW update (train_untils.py:35-42: betas (0.9,0.999), eps 1e-9, decoupled weight decay)
// in three launches instead of torch's ~100 multi-tensor launches over 669 tensors.  HBM-bound: 7 floats of
// traffic per parameter (read p,g,m,v; write p,m,v) + one extra read of g for the norm.
// The step counter and the bias corrections live on the device (state[0..3]) so the sequence is graph-replayable.
#include "adnm_common.h"

namespace {
constexpr int kBlock = 256;
constexpr int kNormBlocks = 1024;

__global__ __launch_bounds__(kBlock) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float sm[kBlock / 64];
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float t = g[(n4 << 2) + threadIdx.x];
    acc = fmaf(t, t, acc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// state: [0] step, [1] sum of squares of the gradient, [2] 1 - beta1^step, [3] sqrt(1 - beta2^step)
__global__ void optim_tick_kernel(float* __restrict__ state, float beta1, float beta2) {
  const float step = state[0] + 1.0f;
  state[0] = step;
  state[2] = 1.0f - powf(beta1, step);
  state[3] = sqrtf(1.0f - powf(beta2, step));
}

// NT: streaming (non-temporal) accesses for g / m / v — read and written exactly once per step — and U float4 per lane and stream in
// flight; p keeps normal accesses (the next forward reads it)
template <bool NT, int U>
__global__ __launch_bounds__(kBlock) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, const float* __restrict__ state, float lr,
                                                       float beta1, float beta2, float eps, float wd, float max_norm) {
  float coef = 1.0f;
  if (max_norm > 0.f) {  // torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (norm + 1e-6), max=1)
    coef = max_norm / (sqrtf(state[1]) + 1e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
  }
  const float bc1 = state[2], bc2s = state[3];
  const float decay = 1.0f - lr * wd, step_size = lr / bc1;
  const int64_t n4 = n >> 2;
  using f4 = __attribute__((ext_vector_type(4))) float;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n4; i0 += U * stride) {
    f4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {   // every load of the trip first
      const int64_t i = i0 + u * stride;
      if (i < n4) {
        pp[u] = reinterpret_cast<const f4*>(p)[i];
        if (NT) {
          gg[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(g) + i);
          mm[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(m) + i);
          vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(v) + i);
        } else {
          gg[u] = reinterpret_cast<const f4*>(g)[i], mm[u] = reinterpret_cast<const f4*>(m)[i], vv[u] = reinterpret_cast<const f4*>(v)[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= n4) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gk = gg[u][k] * coef;
        float P = pp[u][k] * decay;                               // p.mul_(1 - lr*wd)
        const float M = mm[u][k] + (1.0f - beta1) * (gk - mm[u][k]);      // exp_avg.lerp_(g, 1-beta1)
        const float V = vv[u][k] * beta2 + (1.0f - beta2) * gk * gk;      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
        const float denom = sqrtf(V) / bc2s + eps;
        P -= step_size * (M / denom);                             // p.addcdiv_(exp_avg, denom, value=-lr/bc1)
        pp[u][k] = P, mm[u][k] = M, vv[u][k] = V;
      }
      reinterpret_cast<f4*>(p)[i] = pp[u];
      if (NT) {
        __builtin_nontemporal_store(mm[u], reinterpret_cast<f4*>(m) + i);
        __builtin_nontemporal_store(vv[u], reinterpret_cast<f4*>(v) + i);
      } else {
        reinterpret_cast<f4*>(m)[i] = mm[u], reinterpret_cast<f4*>(v)[i] = vv[u];
      }
    }
  }
}
// wire format of the gradient all-reduce (reduce_dtype = bf16): one streaming pass each way, 8 elements per lane
__global__ __launch_bounds__(kBlock) void cast_f32_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n, float scale) {
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kBlock) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(a.x * scale) | ((uint32_t)f32_to_bf16(a.y * scale) << 16);
    o.y = (uint32_t)f32_to_bf16(a.z * scale) | ((uint32_t)f32_to_bf16(a.w * scale) << 16);
    o.z = (uint32_t)f32_to_bf16(b.x * scale) | ((uint32_t)f32_to_bf16(b.y * scale) << 16);
    o.w = (uint32_t)f32_to_bf16(b.z * scale) | ((uint32_t)f32_to_bf16(b.w * scale) << 16);
    reinterpret_cast<uint4*>(dst)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(n8 << 3) + threadIdx.x] = f32_to_bf16(src[(n8 << 3) + threadIdx.x] * scale);
}
__global__ __launch_bounds__(kBlock) void cast_bf16_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, int64_t n, float scale) {
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kBlock) {
    const uint4 v = reinterpret_cast<const uint4*>(src)[i];
    reinterpret_cast<float4*>(dst)[2 * i] = make_float4(scale * __uint_as_float(v.x << 16), scale * __uint_as_float(v.x & 0xffff0000u),
                                                        scale * __uint_as_float(v.y << 16), scale * __uint_as_float(v.y & 0xffff0000u));
    reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4(scale * __uint_as_float(v.z << 16), scale * __uint_as_float(v.z & 0xffff0000u),
                                                            scale * __uint_as_float(v.w << 16), scale * __uint_as_float(v.w & 0xffff0000u));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(n8 << 3) + threadIdx.x] = scale * bf16_to_f32(src[(n8 << 3) + threadIdx.x]);
}
}  // namespace

static int cast_launch(const void* src, void* dst, int64_t n, float scale, adnm_stream_t stream, bool to_bf16) {
  ADNM_REQUIRE(src && dst && n > 0, "cast: null pointer or n <= 0");
  ADNM_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "cast: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = adnm_cdiv(adnm_cdiv(n, 8), kBlock);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  ADNM_PROF(to_bf16 ? "cast_f32_bf16" : "cast_bf16_f32", st, 6.0 * n);
  if (to_bf16) cast_f32_bf16_kernel<<<(unsigned)blocks, kBlock, 0, st>>>((const float*)src, (uint16_t*)dst, n, scale);
  else cast_bf16_f32_kernel<<<(unsigned)blocks, kBlock, 0, st>>>((const uint16_t*)src, (float*)dst, n, scale);
  ADNM_CHECK_LAUNCH("cast");
  return ADNM_OK;
}
extern "C" int adnm_cast_f32_bf16(const void* src, void* dst, int64_t n, float scale, adnm_stream_t stream) {
  return cast_launch(src, dst, n, scale, stream, true);
}
extern "C" int adnm_cast_bf16_f32(const void* src, void* dst, int64_t n, float scale, adnm_stream_t stream) {
  return cast_launch(src, dst, n, scale, stream, false);
}

extern "C" int64_t adnm_adamw_ws_bytes(void) { return kNormBlocks * (int64_t)sizeof(float); }

extern "C" int adnm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float* state, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float max_norm, void* ws, int64_t ws_bytes, adnm_stream_t stream) {
  ADNM_REQUIRE(p && g && m && v && state, "adamw_step: null pointer");
  ADNM_REQUIRE(n > 0 && n % 4 == 0, "adamw_step: n=%lld must be a positive multiple of 4 (pad the flat buffers)", (long long)n);
  if (!ws || ws_bytes < adnm_adamw_ws_bytes()) {
    adnm_set_error("adamw_step: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_adamw_ws_bytes());
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  { ADNM_PROF("grad_sumsq", st, 4.0 * n); sumsq_partial_kernel<<<kNormBlocks, kBlock, 0, st>>>(g, n, part); }
  adnm_launch_fold("grad_sumsq_fold", part, kNormBlocks, 1, {state + 1, 1}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  optim_tick_kernel<<<1, 1, 0, st>>>(state, beta1, beta2);
  int64_t blocks = adnm_cdiv(n / 4, kBlock);
  if (blocks > 4096) blocks = 4096;
  {   // measured in one session (two runs each): plain accesses 0.446 ms, non-temporal g / m / v 0.414, + two float4 per lane 0.400 / 0.47 without
    ADNM_PROF("adamw_update", st, 4.0 * n * 7);
    adamw_kernel<true, 1><<<(unsigned)blocks, kBlock, 0, st>>>(p, g, m, v, n, state, lr, beta1, beta2, eps, weight_decay, max_norm);
  }
  ADNM_CHECK_LAUNCH("adamw_step");
  return ADNM_OK;
}
