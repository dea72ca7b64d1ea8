// Stand-alone activations of the conv stack that are not an epilogue of another kernel:
//   act_fwd / act_bwd — GELU (exact, erf) or SiLU over a flat fp32 array: Mlp.act1 between fc1 and fc2 (model_untils.py:52-70) when the
//                       producing GEMM could not take it as its epilogue, and the backward dpre = dy * act'(pre) of the fused ones;
//   swish_fwd / bwd   — Swish with a learnable slope, y = x * sigmoid(beta * x) (model_untils.py:162-169; OutProj.conv2's activation):
//                       one pass each way, d beta through per-workgroup partials + the shared deterministic fold.
#include "adnm_common.h"

namespace {
constexpr int kBlock = 256;
inline unsigned grid_for(int64_t n4) {
  const int64_t g = adnm_cdiv(n4, kBlock);
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

template <int ACT>
__global__ __launch_bounds__(kBlock) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(act_fwd<ACT>(v.x), act_fwd<ACT>(v.y), act_fwd<ACT>(v.z), act_fwd<ACT>(v.w));
  }
}
template <int ACT>
__global__ __launch_bounds__(kBlock) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dpre, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i], p = reinterpret_cast<const float4*>(pre)[i];
    reinterpret_cast<float4*>(dpre)[i] = make_float4(g.x * act_grad<ACT>(p.x), g.y * act_grad<ACT>(p.y), g.z * act_grad<ACT>(p.z), g.w * act_grad<ACT>(p.w));
  }
}

__global__ __launch_bounds__(kBlock) void swish_fwd_kernel(const float* __restrict__ x, const float* __restrict__ beta, float* __restrict__ y, int64_t n4) {
  const float b = beta[0];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(v.x * sigmoidf_(b * v.x), v.y * sigmoidf_(b * v.y), v.z * sigmoidf_(b * v.z), v.w * sigmoidf_(b * v.w));
  }
}
// dx = dy * s (1 + b x (1 - s)),  d beta = sum dy * x^2 * s (1 - s),  s = sigmoid(b x)
__global__ __launch_bounds__(kBlock) void swish_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ beta,
                                                           float* __restrict__ dx, float* __restrict__ part, int64_t n4) {
  __shared__ float sm[kBlock / 64];
  const float b = beta[0];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i], v = reinterpret_cast<const float4*>(x)[i];
    const float ge[4] = {g.x, g.y, g.z, g.w}, ve[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s = sigmoidf_(b * ve[e]), t = s * (1.0f - s);
      o[e] = ge[e] * (s + b * ve[e] * t);
      acc = fmaf(ge[e] * ve[e] * ve[e], t, acc);
    }
    reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
inline unsigned swish_blocks(int64_t n4) {
  const int64_t g = adnm_cdiv(n4, kBlock * 4);
  return (unsigned)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
}  // namespace

extern "C" int adnm_act_fwd(const float* x, float* y, int64_t n, int act, adnm_stream_t stream) {
  ADNM_REQUIRE(x && y && n > 0 && n % 4 == 0 && al16(x) && al16(y), "act_fwd: needs 16-byte aligned fp32 arrays of a multiple of 4 elements (n=%lld)", (long long)n);
  ADNM_REQUIRE(act == ADNM_ACT_GELU || act == ADNM_ACT_SILU, "act_fwd: activation %d not in {silu, gelu}", act);
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("act_fwd", st, 8.0 * n);
  if (act == ADNM_ACT_GELU) act_fwd_kernel<ADNM_ACT_GELU><<<grid_for(n / 4), kBlock, 0, st>>>(x, y, n / 4);
  else act_fwd_kernel<ADNM_ACT_SILU><<<grid_for(n / 4), kBlock, 0, st>>>(x, y, n / 4);
  ADNM_CHECK_LAUNCH("act_fwd");
  return ADNM_OK;
}

extern "C" int adnm_act_bwd(const float* dy, const float* pre, float* dpre, int64_t n, int act, adnm_stream_t stream) {
  ADNM_REQUIRE(dy && pre && dpre && n > 0 && n % 4 == 0 && al16(dy) && al16(pre) && al16(dpre),
               "act_bwd: needs 16-byte aligned fp32 arrays of a multiple of 4 elements (n=%lld)", (long long)n);
  ADNM_REQUIRE(act == ADNM_ACT_GELU || act == ADNM_ACT_SILU, "act_bwd: activation %d not in {silu, gelu}", act);
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("act_bwd", st, 12.0 * n);
  if (act == ADNM_ACT_GELU) act_bwd_kernel<ADNM_ACT_GELU><<<grid_for(n / 4), kBlock, 0, st>>>(dy, pre, dpre, n / 4);
  else act_bwd_kernel<ADNM_ACT_SILU><<<grid_for(n / 4), kBlock, 0, st>>>(dy, pre, dpre, n / 4);
  ADNM_CHECK_LAUNCH("act_bwd");
  return ADNM_OK;
}

extern "C" int adnm_swish_fwd(const float* x, const float* beta, float* y, int64_t n, adnm_stream_t stream) {
  ADNM_REQUIRE(x && beta && y && n > 0 && n % 4 == 0 && al16(x) && al16(y), "swish_fwd: needs 16-byte aligned fp32 arrays of a multiple of 4 elements (n=%lld)",
               (long long)n);
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("swish_fwd", st, 8.0 * n);
  swish_fwd_kernel<<<grid_for(n / 4), kBlock, 0, st>>>(x, beta, y, n / 4);
  ADNM_CHECK_LAUNCH("swish_fwd");
  return ADNM_OK;
}

extern "C" int64_t adnm_swish_bwd_ws_bytes(int64_t n) { return n > 0 ? (int64_t)swish_blocks(n / 4) * (int64_t)sizeof(float) : 0; }

extern "C" int adnm_swish_bwd(const float* dy, const float* x, const float* beta, float* dx, float* dbeta, void* ws, int64_t ws_bytes, int64_t n,
                              adnm_stream_t stream) {
  ADNM_REQUIRE(dy && x && beta && dx && dbeta && n > 0 && n % 4 == 0 && al16(dy) && al16(x) && al16(dx),
               "swish_bwd: needs 16-byte aligned fp32 arrays of a multiple of 4 elements (n=%lld)", (long long)n);
  if (!ws || ws_bytes < adnm_swish_bwd_ws_bytes(n)) {
    adnm_set_error("swish_bwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_swish_bwd_ws_bytes(n));
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = swish_blocks(n / 4);
  {
    ADNM_PROF("swish_bwd", st, 12.0 * n);
    swish_bwd_kernel<<<grid, kBlock, 0, st>>>(dy, x, beta, dx, (float*)ws, n / 4);
  }
  ADNM_CHECK_LAUNCH("swish_bwd");
  adnm_launch_fold("swish_bwd_fold", (const float*)ws, (int)grid, 1, {dbeta, 1}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("swish_bwd_fold");
  return ADNM_OK;
}
