// Small fused element streams of the hot path.  HBM-bound, 16 B per lane, grid-stride.
#include "adnm_common.h"

namespace {
constexpr int kBlock = 256;

// FeedForward gate (model_untils.py:194-195): y = gelu(x1) * sigmoid(x2), x1|x2 = column halves of h.
template <typename T>
__global__ __launch_bounds__(kBlock) void gate_fwd_kernel(const T* __restrict__ h, int64_t ldh, T* __restrict__ y, int64_t ldy,
                                                          int64_t M, int F) {
  const int F4 = F >> 2;
  const int64_t total = M * F4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t m = i / F4;
    const int f = (int)(i % F4) * 4;
    const float4 a = Io<T>::ld4(h + m * ldh + f), b = Io<T>::ld4(h + m * ldh + F + f);
    Io<T>::st4(y + m * ldy + f, make_float4(geluf_(a.x) * sigmoidf_(b.x), geluf_(a.y) * sigmoidf_(b.y), geluf_(a.z) * sigmoidf_(b.z),
                                            geluf_(a.w) * sigmoidf_(b.w)));
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gate_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ h, int64_t ldh,
                                                          T* __restrict__ dh, int64_t lddh, int64_t M, int F) {
  const int F4 = F >> 2;
  const int64_t total = M * F4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t m = i / F4;
    const int f = (int)(i % F4) * 4;
    const float4 a = Io<T>::ld4(h + m * ldh + f), b = Io<T>::ld4(h + m * ldh + F + f), g = Io<T>::ld4(dy + m * lddy + f);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, gv[4] = {g.x, g.y, g.z, g.w};
    float d1[4], d2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float s = sigmoidf_(bv[k]);
      d1[k] = gv[k] * s * gelu_gradf_(av[k]);
      d2[k] = gv[k] * geluf_(av[k]) * s * (1.f - s);
    }
    Io<T>::st4(dh + m * lddh + f, make_float4(d1[0], d1[1], d1[2], d1[3]));
    Io<T>::st4(dh + m * lddh + F + f, make_float4(d2[0], d2[1], d2[2], d2[3]));
  }
}

unsigned grid_for(int64_t total) {
  int64_t g = adnm_cdiv(total, kBlock);
  return (unsigned)(g < 2048 ? (g < 1 ? 1 : g) : 2048);
}
}  // namespace

extern "C" int adnm_gate_fwd(const void* h, int64_t ldh, void* y, int64_t ldy, int64_t M, int64_t F, int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(h && y, "gate_fwd: null pointer");
  ADNM_REQUIRE(M > 0 && F > 0 && F % 4 == 0 && ldh >= 2 * F && ldy >= F && ldh % 4 == 0 && ldy % 4 == 0, "gate_fwd: bad shape M=%lld F=%lld",
               (long long)M, (long long)F);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "gate_fwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ADNM_F32) { ADNM_PROF("gate_fwd", st, 4.0 * M * F * 3); gate_fwd_kernel<float><<<grid_for(M * F / 4), kBlock, 0, st>>>((const float*)h, ldh, (float*)y, ldy, M, (int)F); }
  else { ADNM_PROF("gate_fwd", st, 2.0 * M * F * 3); gate_fwd_kernel<uint16_t><<<grid_for(M * F / 4), kBlock, 0, st>>>((const uint16_t*)h, ldh, (uint16_t*)y, ldy, M, (int)F); }
  ADNM_CHECK_LAUNCH("gate_fwd");
  return ADNM_OK;
}

extern "C" int adnm_gate_bwd(const void* dy, int64_t lddy, const void* h, int64_t ldh, void* dh, int64_t lddh, int64_t M, int64_t F,
                             int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(dy && h && dh, "gate_bwd: null pointer");
  ADNM_REQUIRE(M > 0 && F > 0 && F % 4 == 0 && ldh >= 2 * F && lddh >= 2 * F && lddy >= F && ldh % 4 == 0 && lddh % 4 == 0 && lddy % 4 == 0,
               "gate_bwd: bad shape M=%lld F=%lld", (long long)M, (long long)F);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "gate_bwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ADNM_F32)
    { ADNM_PROF("gate_bwd", st, 4.0 * M * F * 5); gate_bwd_kernel<float><<<grid_for(M * F / 4), kBlock, 0, st>>>((const float*)dy, lddy, (const float*)h, ldh, (float*)dh, lddh, M, (int)F); }
  else
    { ADNM_PROF("gate_bwd", st, 2.0 * M * F * 5); gate_bwd_kernel<uint16_t><<<grid_for(M * F / 4), kBlock, 0, st>>>((const uint16_t*)dy, lddy, (const uint16_t*)h, ldh, (uint16_t*)dh, lddh, M, (int)F); }
  ADNM_CHECK_LAUNCH("gate_bwd");
  return ADNM_OK;
}
