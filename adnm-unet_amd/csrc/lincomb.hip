// Fused scalar/channel-affine mixes of the token stream:  y = gamma[c] * (s0*x0 + s1*x1 + s2*x2)
// with s_k learnable 0-d parameters and gamma a learnable per-channel vector.  Covers, in ONE pass each way,
// the reference's chains of broadcast mul/add ops and their scalar-gradient reductions:
//   Block.forward            x = beta1*x + beta2*mixer(..), x = beta3*x + beta4*ffn(..), x*gamma   (ADNMUNet.py:152,158,161)
//   Attention.forward        same pattern (ADNMUNet.py:226,232,234)
//   WTLayer / PatchEmbed / OutProj   alpha*wtconv(x) + beta*x, (.)*gamma   (model_untils.py:306-310,418-421,881-883)
//   EncoderToDecoder         (alpha1*x1 + alpha2*x2 + alpha3*x3) * gamma   (model_untils.py:785-787)
// HBM-bound: (K+1)*M*C elements forward, (2K+1)*M*C backward (the reference's unfused chain moves ~3x that and
// launches ~10 kernels).  Same lane mapping and deterministic two-stage reduction as rownorm.hip.
#include "adnm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxPartBlocks = 1024;  // 4 workgroups per CU: enough waves in flight to stream at HBM rate (256 left 1 per CU: ~2 TB/s)

inline int lanes_per_row(int64_t d) {
  int l = 1;
  while (l < 64 && (int64_t)l * 4 < d) l <<= 1;
  return l;
}

struct Ops {
  const void* x[3];
  int64_t ld[3];
  const float* s[3];
};

template <typename T, int K>
__global__ __launch_bounds__(kBlock) void lincomb_fwd_kernel(Ops o, const float* __restrict__ gamma, T* __restrict__ y, int64_t ldy,
                                                             int64_t M, int C) {
  const int C4 = C >> 2;
  const int64_t total = M * C4;
  float sc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) sc[k] = o.s[k] ? *o.s[k] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t m = i / C4;
    const int c = (int)(i % C4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float4 v = Io<T>::ld4((const T*)o.x[k] + m * o.ld[k] + c);
      acc.x = fmaf(sc[k], v.x, acc.x); acc.y = fmaf(sc[k], v.y, acc.y);
      acc.z = fmaf(sc[k], v.z, acc.z); acc.w = fmaf(sc[k], v.w, acc.w);
    }
    if (gamma) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      acc.x *= g.x; acc.y *= g.y; acc.z *= g.z; acc.w *= g.w;
    }
    Io<T>::st4(y + m * ldy + c, acc);
  }
}

struct Grads {
  void* dx[3];
  int64_t ld[3];
};

// partial row per block: [dgamma(C) | ds0 | ds1 | ds2]
template <typename T, int K, int IT>
__global__ __launch_bounds__(kBlock) void lincomb_bwd_kernel(const T* __restrict__ dy, int64_t lddy, Ops o, const float* __restrict__ gamma,
                                                             Grads gr, float* __restrict__ part, int64_t M, int C, int lpr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // waves x (C+4)
  const int lane_in_row = threadIdx.x & (lpr - 1);
  const int rows_per_block = kBlock / lpr;
  float sc[K], ds[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    sc[k] = o.s[k] ? *o.s[k] : 1.f;
    ds[k] = 0.f;
  }
  float4 gm[IT], ag[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (i * lpr + lane_in_row) * 4;
    gm[i] = (gamma && c < C) ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // two rows per trip with ALL loads issued before the first store: the dx pointers are opaque to the compiler (no restrict
  // through the struct), so without this it orders every load behind the previous row's stores
  const int64_t rstride = (int64_t)gridDim.x * rows_per_block;
  for (int64_t row = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x / lpr); row < M; row += 2 * rstride) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (i * lpr + lane_in_row) * 4;
      if (c >= C) continue;
      float4 g[2], v[2][K];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t r = row + u * rstride;
        const bool ok = r < M;
        const int64_t rr = ok ? r : row;
        g[u] = Io<T>::ld4(dy + rr * lddy + c);
        if (!ok) g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < K; ++k) v[u][k] = Io<T>::ld4((const T*)o.x[k] + rr * o.ld[k] + c);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t r = row + u * rstride;
        if (r >= M) continue;
        const float4 gg = make_float4(g[u].x * gm[i].x, g[u].y * gm[i].y, g[u].z * gm[i].z, g[u].w * gm[i].w);
        float4 mix = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float4 vv = v[u][k];
          ds[k] += gg.x * vv.x + gg.y * vv.y + gg.z * vv.z + gg.w * vv.w;
          mix.x = fmaf(sc[k], vv.x, mix.x); mix.y = fmaf(sc[k], vv.y, mix.y);
          mix.z = fmaf(sc[k], vv.z, mix.z); mix.w = fmaf(sc[k], vv.w, mix.w);
          if (gr.dx[k]) Io<T>::st4((T*)gr.dx[k] + r * gr.ld[k] + c, make_float4(sc[k] * gg.x, sc[k] * gg.y, sc[k] * gg.z, sc[k] * gg.w));
        }
        ag[i].x = fmaf(g[u].x, mix.x, ag[i].x); ag[i].y = fmaf(g[u].y, mix.y, ag[i].y);
        ag[i].z = fmaf(g[u].z, mix.z, ag[i].z); ag[i].w = fmaf(g[u].w, mix.w, ag[i].w);
      }
    }
  }
  const int sstride = C + 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    ag[i].x = wave_sum_from(ag[i].x, lpr); ag[i].y = wave_sum_from(ag[i].y, lpr);
    ag[i].z = wave_sum_from(ag[i].z, lpr); ag[i].w = wave_sum_from(ag[i].w, lpr);
    const int c = (i * lpr + lane_in_row) * 4;
    if (lane < lpr && c < C) *reinterpret_cast<float4*>(smem + wave * sstride + c) = ag[i];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    ds[k] = wave_sum(ds[k]);
    if (lane == 0) smem[wave * sstride + C + k] = ds[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C + 3; c += kBlock) {
    float t = 0.f;
    if (c < C + K) {
#pragma unroll
      for (int wv = 0; wv < kBlock / 64; ++wv) t += smem[wv * sstride + c];
    }
    part[(int64_t)blockIdx.x * (C + 3) + c] = t;
  }
}

int bwd_blocks(int64_t M, int64_t C) {
  const int lpr = lanes_per_row(C);
  int64_t nb = adnm_cdiv(M, kBlock / lpr);
  return (int)(nb < kMaxPartBlocks ? nb : kMaxPartBlocks);
}

int count_ops(const void* x0, const void* x1, const void* x2) { return x2 ? 3 : (x1 ? 2 : 1); }

template <typename T>
int run_fwd(Ops o, int K, const float* gamma, void* y, int64_t ldy, int64_t M, int64_t C, hipStream_t st) {
  int64_t g = adnm_cdiv(M * (C / 4), kBlock);
  const unsigned grid = (unsigned)(g < 2048 ? g : 2048);
  ADNM_PROF("lincomb_fwd", st, (double)sizeof(T) * M * C * (K + 1));
  if (K == 1) lincomb_fwd_kernel<T, 1><<<grid, kBlock, 0, st>>>(o, gamma, (T*)y, ldy, M, (int)C);
  else if (K == 2) lincomb_fwd_kernel<T, 2><<<grid, kBlock, 0, st>>>(o, gamma, (T*)y, ldy, M, (int)C);
  else lincomb_fwd_kernel<T, 3><<<grid, kBlock, 0, st>>>(o, gamma, (T*)y, ldy, M, (int)C);
  ADNM_CHECK_LAUNCH("lincomb_fwd");
  return ADNM_OK;
}

template <typename T, int K>
void run_bwd_k(const void* dy, int64_t lddy, Ops o, const float* gamma, Grads gr, float* part, int64_t M, int64_t C, int nblk, int it, int lpr,
               size_t smem, hipStream_t st) {
#define BWD(IT) lincomb_bwd_kernel<T, K, IT><<<nblk, kBlock, smem, st>>>((const T*)dy, lddy, o, gamma, gr, part, M, (int)C, lpr)
  if (it <= 1) BWD(1);
  else if (it <= 2) BWD(2);
  else if (it <= 4) BWD(4);
  else BWD(8);
#undef BWD
}

template <typename T>
int run_bwd(const void* dy, int64_t lddy, Ops o, int K, const float* gamma, Grads gr, float* ds0, float* ds1, float* ds2, float* dgamma,
            float* part, int64_t M, int64_t C, hipStream_t st) {
  const int lpr = lanes_per_row(C);
  const int it = (int)adnm_cdiv(C, (int64_t)lpr * 4);
  const int nblk = bwd_blocks(M, C);
  const size_t smem = (size_t)(kBlock / 64) * (C + 4) * sizeof(float);
  int ndx = 0;
  for (int k = 0; k < K; ++k) ndx += gr.dx[k] ? 1 : 0;
  {
    ADNM_PROF("lincomb_bwd", st, (double)sizeof(T) * M * C * (1 + K + ndx));
    if (K == 1) run_bwd_k<T, 1>(dy, lddy, o, gamma, gr, part, M, C, nblk, it, lpr, smem, st);
    else if (K == 2) run_bwd_k<T, 2>(dy, lddy, o, gamma, gr, part, M, C, nblk, it, lpr, smem, st);
    else run_bwd_k<T, 3>(dy, lddy, o, gamma, gr, part, M, C, nblk, it, lpr, smem, st);
  }
  ADNM_CHECK_LAUNCH("lincomb_bwd");
  adnm_launch_fold("lincomb_bwd_fold", part, nblk, (int)C + 3, {dgamma, (int)C}, {ds0, 1}, {ds1, 1}, {ds2, 1}, st);
  ADNM_CHECK_LAUNCH("lincomb_bwd_fold");
  return ADNM_OK;
}

int check(const char* who, const void* x0, int64_t ld0, const void* x1, int64_t ld1, const void* x2, int64_t ld2, int64_t M, int64_t C, int dtype) {
  ADNM_REQUIRE(x0 && (x1 || !x2), "%s: operands must be filled in order", who);
  ADNM_REQUIRE(M > 0 && C >= 4 && C % 4 == 0 && C <= 2048, "%s: C=%lld must be a multiple of 4 in [4,2048]", who, (long long)C);
  ADNM_REQUIRE(ld0 >= C && ld0 % 4 == 0 && (!x1 || (ld1 >= C && ld1 % 4 == 0)) && (!x2 || (ld2 >= C && ld2 % 4 == 0)),
               "%s: row strides must be >= C and multiples of 4", who);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "%s: bad dtype %d", who, dtype);
  return ADNM_OK;
}

}  // namespace

extern "C" int adnm_lincomb_fwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, const void* x2, int64_t ld2, const float* s0,
                                const float* s1, const float* s2, const float* gamma, void* y, int64_t ldy, int64_t M, int64_t C, int dtype,
                                adnm_stream_t stream) {
  if (int rc = check("lincomb_fwd", x0, ld0, x1, ld1, x2, ld2, M, C, dtype)) return rc;
  ADNM_REQUIRE(y && ldy >= C && ldy % 4 == 0, "lincomb_fwd: bad output");
  Ops o{{x0, x1, x2}, {ld0, ld1, ld2}, {s0, s1, s2}};
  const int K = count_ops(x0, x1, x2);
  hipStream_t st = (hipStream_t)stream;
  return dtype == ADNM_F32 ? run_fwd<float>(o, K, gamma, y, ldy, M, C, st) : run_fwd<uint16_t>(o, K, gamma, y, ldy, M, C, st);
}

extern "C" int64_t adnm_lincomb_bwd_ws_bytes(int64_t M, int64_t C) {
  if (M <= 0 || C <= 0) return 0;
  return (int64_t)bwd_blocks(M, C) * (C + 3) * (int64_t)sizeof(float);
}

extern "C" int adnm_lincomb_bwd(const void* dy, int64_t lddy, const void* x0, int64_t ld0, const void* x1, int64_t ld1, const void* x2,
                                int64_t ld2, const float* s0, const float* s1, const float* s2, const float* gamma, void* dx0, int64_t lddx0,
                                void* dx1, int64_t lddx1, void* dx2, int64_t lddx2, float* ds0, float* ds1, float* ds2, float* dgamma, void* ws,
                                int64_t ws_bytes, int64_t M, int64_t C, int dtype, adnm_stream_t stream) {
  if (int rc = check("lincomb_bwd", x0, ld0, x1, ld1, x2, ld2, M, C, dtype)) return rc;
  ADNM_REQUIRE(dy && lddy >= C && lddy % 4 == 0, "lincomb_bwd: bad dy");
  ADNM_REQUIRE((!dx0 || lddx0 >= C) && (!dx1 || lddx1 >= C) && (!dx2 || lddx2 >= C), "lincomb_bwd: bad gradient strides");
  if (!ws || ws_bytes < adnm_lincomb_bwd_ws_bytes(M, C)) {
    adnm_set_error("lincomb_bwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_lincomb_bwd_ws_bytes(M, C));
    return ADNM_EWORKSPACE;
  }
  Ops o{{x0, x1, x2}, {ld0, ld1, ld2}, {s0, s1, s2}};
  Grads gr{{dx0, dx1, dx2}, {lddx0, lddx1, lddx2}};
  const int K = count_ops(x0, x1, x2);
  hipStream_t st = (hipStream_t)stream;
  return dtype == ADNM_F32 ? run_bwd<float>(dy, lddy, o, K, gamma, gr, ds0, ds1, ds2, dgamma, (float*)ws, M, C, st)
                           : run_bwd<uint16_t>(dy, lddy, o, K, gamma, gr, ds0, ds1, ds2, dgamma, (float*)ws, M, C, st);
}
