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

// IntensityGate (model_untils.py:523-532): y = silu(enhance * (x - threshold)), both learnable scalars.
template <typename T>
__global__ __launch_bounds__(kBlock) void igate_fwd_kernel(const T* __restrict__ x, const float* __restrict__ enh, const float* __restrict__ thr,
                                                           T* __restrict__ y, int64_t n4) {
  const float a = *enh, t = *thr;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = Io<T>::ld4(x + i * 4);
    Io<T>::st4(y + i * 4, make_float4(siluf_(a * (v.x - t)), siluf_(a * (v.y - t)), siluf_(a * (v.z - t)), siluf_(a * (v.w - t))));
  }
}

// part[blockIdx.x] = {sum g*silu'(z)*(x-t), sum g*silu'(z)}
template <typename T>
__global__ __launch_bounds__(kBlock) void igate_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ enh,
                                                           const float* __restrict__ thr, T* __restrict__ dx, float* __restrict__ part,
                                                           int64_t n4) {
  __shared__ float sm[2][kBlock / 64];
  const float a = *enh, t = *thr;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = Io<T>::ld4(x + i * 4), g = Io<T>::ld4(dy + i * 4);
    const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = xv[k] - t, gp = gv[k] * silu_gradf_(a * d);
      o[k] = gp * a;
      s1 = fmaf(gp, d, s1);
      s2 += gp;
    }
    Io<T>::st4(dx + i * 4, make_float4(o[0], o[1], o[2], o[3]));
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s1; sm[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
    part[blockIdx.x * 2 + 1] = -a * ((sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]));
  }
}

// Head merge of Block / Attention / WTLayer (ADNMUNet.py:124-131, :214-221; model_untils.py:398-400):
//   y = cat((a1*x, a2*r), -1) [+ cat((a3*f, a4*f), -1)]      x, r, f: (M,d) -> y: (M,2d)
template <typename T>
__global__ __launch_bounds__(kBlock) void catmix_fwd_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ r, int64_t ldr,
                                                            const T* __restrict__ f, int64_t ldf, const float* __restrict__ a1,
                                                            const float* __restrict__ a2, const float* __restrict__ a3,
                                                            const float* __restrict__ a4, T* __restrict__ y, int64_t M, int d) {
  const int d4 = d >> 2;
  const float s1 = a1 ? *a1 : 1.f, s2 = a2 ? *a2 : 1.f, s3 = (f && a3) ? *a3 : 1.f, s4 = (f && a4) ? *a4 : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M * d4; i += (int64_t)gridDim.x * kBlock) {
    const int64_t m = i / d4;
    const int c = (int)(i % d4) * 4;
    const float4 xv = Io<T>::ld4(x + m * ldx + c), rv = Io<T>::ld4(r + m * ldr + c);
    float4 l = make_float4(s1 * xv.x, s1 * xv.y, s1 * xv.z, s1 * xv.w), q = make_float4(s2 * rv.x, s2 * rv.y, s2 * rv.z, s2 * rv.w);
    if (f) {
      const float4 fv = Io<T>::ld4(f + m * ldf + c);
      l = make_float4(fmaf(s3, fv.x, l.x), fmaf(s3, fv.y, l.y), fmaf(s3, fv.z, l.z), fmaf(s3, fv.w, l.w));
      q = make_float4(fmaf(s4, fv.x, q.x), fmaf(s4, fv.y, q.y), fmaf(s4, fv.z, q.z), fmaf(s4, fv.w, q.w));
    }
    Io<T>::st4(y + m * 2 * d + c, l);
    Io<T>::st4(y + m * 2 * d + d + c, q);
  }
}

// part[block] = {sum dyL*x, sum dyR*r, sum dyL*f, sum dyR*f}
template <typename T>
__global__ __launch_bounds__(kBlock) void catmix_bwd_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx,
                                                            const T* __restrict__ r, int64_t ldr, const T* __restrict__ f, int64_t ldf,
                                                            const float* __restrict__ a1, const float* __restrict__ a2,
                                                            const float* __restrict__ a3, const float* __restrict__ a4, T* __restrict__ dx,
                                                            T* __restrict__ dr, T* __restrict__ df, float* __restrict__ part, int64_t M, int d) {
  __shared__ float sm[4][kBlock / 64];
  const int d4 = d >> 2;
  const float s1 = a1 ? *a1 : 1.f, s2 = a2 ? *a2 : 1.f, s3 = (f && a3) ? *a3 : 1.f, s4 = (f && a4) ? *a4 : 1.f;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M * d4; i += (int64_t)gridDim.x * kBlock) {
    const int64_t m = i / d4;
    const int c = (int)(i % d4) * 4;
    const float4 gl = Io<T>::ld4(dy + m * lddy + c), gr = Io<T>::ld4(dy + m * lddy + d + c);
    const float4 xv = Io<T>::ld4(x + m * ldx + c), rv = Io<T>::ld4(r + m * ldr + c);
    acc[0] += gl.x * xv.x + gl.y * xv.y + gl.z * xv.z + gl.w * xv.w;
    acc[1] += gr.x * rv.x + gr.y * rv.y + gr.z * rv.z + gr.w * rv.w;
    if (dx) Io<T>::st4(dx + m * d + c, make_float4(s1 * gl.x, s1 * gl.y, s1 * gl.z, s1 * gl.w));
    if (dr) Io<T>::st4(dr + m * d + c, make_float4(s2 * gr.x, s2 * gr.y, s2 * gr.z, s2 * gr.w));
    if (f) {
      const float4 fv = Io<T>::ld4(f + m * ldf + c);
      acc[2] += gl.x * fv.x + gl.y * fv.y + gl.z * fv.z + gl.w * fv.w;
      acc[3] += gr.x * fv.x + gr.y * fv.y + gr.z * fv.z + gr.w * fv.w;
      if (df)
        Io<T>::st4(df + m * d + c, make_float4(fmaf(s3, gl.x, s4 * gr.x), fmaf(s3, gl.y, s4 * gr.y), fmaf(s3, gl.z, s4 * gr.z),
                                               fmaf(s3, gl.w, s4 * gr.w)));
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    acc[k] = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) sm[k][threadIdx.x >> 6] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = (sm[threadIdx.x][0] + sm[threadIdx.x][1]) + (sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

// enRainfallLoss (models/loss.py:30-57 of the reference), value and d/dpred in one pass.
//   e = w(pred>=t) * |pred-t| * (1 + [t>=0.7] alpha exp(t))  +  [t>=0.7 and pred<t] gamma (exp(alpha (t-pred)) - 1);  loss = sum e / n
__global__ __launch_bounds__(kBlock) void rainloss_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, float* __restrict__ grad,
                                                          float* __restrict__ part, int64_t n, float omega, float alpha, float gamma) {
  __shared__ float sm[kBlock / 64];
  const float inv_n = 1.0f / (float)n;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float p = pred[i], t = tgt[i], diff = p - t;
    const bool over = p >= t, heavy = t >= 0.7f;
    const float w = over ? 1.0f - omega : omega;
    const float wi = heavy ? alpha * __expf(t) : 0.f;
    float e = w * fabsf(diff) * (1.0f + wi);
    float g = w * (1.0f + wi) * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
    if (gamma != 0.f && heavy && !over) {
      const float ex = __expf(alpha * (t - p));
      e += gamma * (ex - 1.0f);
      g -= gamma * alpha * ex;
    }
    acc += e;
    grad[i] = g * inv_n;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) * inv_n;
}

// nn.Conv1d(1, 1, 3, padding=1) over the channel axis of (B, n) (get_all_att, model_untils.py:548,592): one workgroup.
// fwd: y = w0 x[i-1] + w1 x[i] + w2 x[i+1] + b.  bwd: dx, dw[3], db in the same pass (block reduction, no partials).
// One workgroup of 1024 lanes, four elements in flight per lane: this is ~8.5 K elements behind a chain of dependent loads (as a
// 256-lane loop of 34 trips it took 30 us).
constexpr int kC1Threads = 1024;
__global__ __launch_bounds__(kC1Threads) void conv1d3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ dy, float* __restrict__ y, float* __restrict__ dx,
                                                             float* __restrict__ dwb, int B, int n) {
  __shared__ float sm[4][kC1Threads / 64];
  const float w0 = w[0], w1 = w[1], w2 = w[2], bv = bias ? bias[0] : 0.f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, ab = 0.f;
  const int total = B * n;
  for (int base = threadIdx.x; base < total; base += 4 * kC1Threads) {
    float xm[4], xc[4], xp[4], gm[4], gc[4], gp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // all loads of the four elements first
      const int idx = base + u * kC1Threads, i = idx % n;
      const bool ok = idx < total;
      xm[u] = ok && i > 0 ? x[idx - 1] : 0.f;
      xc[u] = ok ? x[idx] : 0.f;
      xp[u] = ok && i + 1 < n ? x[idx + 1] : 0.f;
      if (dy) {
        gm[u] = ok && i > 0 ? dy[idx - 1] : 0.f;
        gc[u] = ok ? dy[idx] : 0.f;
        gp[u] = ok && i + 1 < n ? dy[idx + 1] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * kC1Threads;
      if (idx >= total) break;
      if (!dy) {
        y[idx] = fmaf(w0, xm[u], fmaf(w1, xc[u], fmaf(w2, xp[u], bv)));
      } else {
        dx[idx] = fmaf(w0, gp[u], fmaf(w1, gc[u], w2 * gm[u]));
        a0 = fmaf(gc[u], xm[u], a0); a1 = fmaf(gc[u], xc[u], a1); a2 = fmaf(gc[u], xp[u], a2); ab += gc[u];
      }
    }
  }
  if (!dy) return;
  float acc[4] = {a0, a1, a2, ab};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    acc[k] = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) sm[k][threadIdx.x >> 6] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < kC1Threads / 64; ++q) t += sm[threadIdx.x][q];
    dwb[threadIdx.x] = t;
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

extern "C" int64_t adnm_igate_bwd_ws_bytes(int64_t n) { return (int64_t)(n / 4 / kBlock + 1 < 512 ? n / 4 / kBlock + 1 : 512) * 2 * sizeof(float); }

extern "C" int adnm_igate_fwd(const void* x, const float* enhance, const float* threshold, void* y, int64_t n, int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(x && enhance && threshold && y, "igate_fwd: null pointer");
  ADNM_REQUIRE(n > 0 && n % 4 == 0, "igate_fwd: element count %lld must be a positive multiple of 4", (long long)n);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "igate_fwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("igate_fwd", st, (dtype == ADNM_F32 ? 8.0 : 4.0) * n);
  if (dtype == ADNM_F32) igate_fwd_kernel<float><<<grid_for(n / 4), kBlock, 0, st>>>((const float*)x, enhance, threshold, (float*)y, n / 4);
  else igate_fwd_kernel<uint16_t><<<grid_for(n / 4), kBlock, 0, st>>>((const uint16_t*)x, enhance, threshold, (uint16_t*)y, n / 4);
  ADNM_CHECK_LAUNCH("igate_fwd");
  return ADNM_OK;
}

extern "C" int adnm_igate_bwd(const void* dy, const void* x, const float* enhance, const float* threshold, void* dx, float* denhance,
                              float* dthreshold, void* ws, int64_t ws_bytes, int64_t n, int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(dy && x && enhance && threshold && dx && denhance && dthreshold, "igate_bwd: null pointer");
  ADNM_REQUIRE(n > 0 && n % 4 == 0, "igate_bwd: element count %lld must be a positive multiple of 4", (long long)n);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "igate_bwd: bad dtype %d", dtype);
  if (!ws || ws_bytes < adnm_igate_bwd_ws_bytes(n)) {
    adnm_set_error("igate_bwd: workspace too small");
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int64_t g = n / 4 / kBlock + 1;
  const unsigned grid = (unsigned)(g < 512 ? g : 512);
  {
    ADNM_PROF("igate_bwd", st, (dtype == ADNM_F32 ? 12.0 : 6.0) * n);
    if (dtype == ADNM_F32) igate_bwd_kernel<float><<<grid, kBlock, 0, st>>>((const float*)dy, (const float*)x, enhance, threshold, (float*)dx, (float*)ws, n / 4);
    else igate_bwd_kernel<uint16_t><<<grid, kBlock, 0, st>>>((const uint16_t*)dy, (const uint16_t*)x, enhance, threshold, (uint16_t*)dx, (float*)ws, n / 4);
  }
  adnm_launch_fold("igate_bwd_fold", (const float*)ws, (int)grid, 2, {denhance, 1}, {dthreshold, 1}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("igate_bwd");
  return ADNM_OK;
}

// EncoderToDecoder's entry (model_untils.py:761-763): y = IntensityGate(x + gama * res) with res the channel-attention gate of
// the skip, one value per (sample, channel) broadcast over the L tokens: y[b,l,c] = silu(enh * (x[b,l,c] + gama * res[b,c] - thr)).
// One pass each way instead of the reference's mul / add / sub / mul / silu chain and autograd's broadcast reductions.
namespace {
__global__ __launch_bounds__(kBlock) void igate_res_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gama,
                                                               const float* __restrict__ enh, const float* __restrict__ thr, float* __restrict__ y,
                                                               int B, int L, int C4, int per_token) {
  const float a = *enh, t = *thr, gm = *gama;
  const int64_t n4 = (int64_t)B * L * C4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const int c = (int)(i % C4), b = (int)(i / ((int64_t)L * C4));
    const float4 v = Io<float>::ld4(x + i * 4), r = Io<float>::ld4(res + (per_token ? i : (int64_t)b * C4 + c) * 4);
    Io<float>::st4(y + i * 4, make_float4(siluf_(a * (fmaf(gm, r.x, v.x) - t)), siluf_(a * (fmaf(gm, r.y, v.y) - t)),
                                          siluf_(a * (fmaf(gm, r.z, v.z) - t)), siluf_(a * (fmaf(gm, r.w, v.w) - t))));
  }
}

// workgroup = (sample b, 16 channel quads) x 16 token slices: thread (sl, cl) takes tokens sl, sl+16, ... of its quad; the 16 slices are
// summed through LDS, so d res (already times gama) is complete when the launch ends — the bridge's backward reads it next — and only
// the three scalar gradients {d gama, d enhance, d threshold} go through per-workgroup partials and the (deferrable) fold.
__global__ __launch_bounds__(kBlock) void igate_res_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ res,
                                                               const float* __restrict__ gama, const float* __restrict__ enh,
                                                               const float* __restrict__ thr, float* __restrict__ dx, float* __restrict__ dres,
                                                               float* __restrict__ spart, int B, int L, int C4, int per_token) {
  __shared__ float4 sm[16][16];
  __shared__ float ss[3][kBlock / 64];
  const float a = *enh, t = *thr, gm = *gama;
  const int chunks = (C4 + 15) / 16;
  const int b = blockIdx.x / chunks, cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int cq = (blockIdx.x % chunks) * 16 + cl;
  const bool valid = cq < C4;
  const int c = valid ? cq : 0;
  float4 r4 = Io<float>::ld4(res + (per_token ? 0 : ((int64_t)b * C4 + c) * 4));
  float dr[4] = {0.f, 0.f, 0.f, 0.f}, s_g = 0.f, s_e = 0.f, s_t = 0.f;
  for (int l = valid ? sl : L; l < L; l += 16) {
    const int64_t off = (((int64_t)b * L + l) * C4 + c) * 4;
    const float4 v = Io<float>::ld4(x + off), g = Io<float>::ld4(dy + off);
    if (per_token) r4 = Io<float>::ld4(res + off);
    const float rv[4] = {r4.x, r4.y, r4.z, r4.w};
    const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = fmaf(gm, rv[k], xv[k]) - t, gp = gv[k] * silu_gradf_(a * d);
      o[k] = gp * a;
      dr[k] += o[k];
      s_g = fmaf(o[k], rv[k], s_g);
      s_e = fmaf(gp, d, s_e);
      s_t += gp;
    }
    Io<float>::st4(dx + off, make_float4(o[0], o[1], o[2], o[3]));
    if (per_token) Io<float>::st4(dres + off, make_float4(gm * o[0], gm * o[1], gm * o[2], gm * o[3]));
  }
  sm[sl][cl] = make_float4(dr[0], dr[1], dr[2], dr[3]);
  s_g = wave_sum(s_g), s_e = wave_sum(s_e), s_t = wave_sum(s_t);
  if ((threadIdx.x & 63) == 0) ss[0][threadIdx.x >> 6] = s_g, ss[1][threadIdx.x >> 6] = s_e, ss[2][threadIdx.x >> 6] = s_t;
  __syncthreads();
  if (sl == 0 && valid && !per_token) {
    float4 acc = sm[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) acc.x += sm[k][cl].x, acc.y += sm[k][cl].y, acc.z += sm[k][cl].z, acc.w += sm[k][cl].w;
    Io<float>::st4(dres + ((int64_t)b * C4 + c) * 4, make_float4(gm * acc.x, gm * acc.y, gm * acc.z, gm * acc.w));
  }
  if (threadIdx.x == 0) {
    float* sp = spart + (int64_t)blockIdx.x * 4;
    sp[0] = (ss[0][0] + ss[0][1]) + (ss[0][2] + ss[0][3]);
    sp[1] = (ss[1][0] + ss[1][1]) + (ss[1][2] + ss[1][3]);
    sp[2] = -a * ((ss[2][0] + ss[2][1]) + (ss[2][2] + ss[2][3]));
    sp[3] = 0.f;
  }
}
int64_t igate_res_blocks(int64_t B, int64_t C) { return B * adnm_cdiv(C / 4, 16); }
}  // namespace

extern "C" int64_t adnm_igate_res_bwd_ws_bytes(int64_t B, int64_t L, int64_t C) {
  if (B < 1 || L < 1 || C < 4 || C % 4) return -1;
  return igate_res_blocks(B, C) * 4 * (int64_t)sizeof(float);
}

extern "C" int adnm_igate_res_fwd(const float* x, const float* res, int per_token, const float* gama, const float* enhance, const float* threshold,
                                  float* y, int64_t B, int64_t L, int64_t C, adnm_stream_t stream) {
  ADNM_REQUIRE(x && res && gama && enhance && threshold && y, "igate_res_fwd: null pointer");
  ADNM_REQUIRE(B > 0 && L > 0 && C > 0 && C % 4 == 0 && B * L * C < (1ll << 31), "igate_res_fwd: bad shape B=%lld L=%lld C=%lld (4 | C)", (long long)B,
               (long long)L, (long long)C);
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("igate_fwd", st, 4.0 * (2.0 * B * L * C + B * C));
  igate_res_fwd_kernel<<<grid_for(B * L * C / 4), kBlock, 0, st>>>(x, res, gama, enhance, threshold, y, (int)B, (int)L, (int)(C / 4), per_token);
  ADNM_CHECK_LAUNCH("igate_res_fwd");
  return ADNM_OK;
}

extern "C" int adnm_igate_res_bwd(const float* dy, const float* x, const float* res, int per_token, const float* gama, const float* enhance,
                                  const float* threshold, float* dx, float* dres, float* dgama, float* denhance, float* dthreshold, void* ws, int64_t ws_bytes, int64_t B,
                                  int64_t L, int64_t C, adnm_stream_t stream) {
  ADNM_REQUIRE(dy && x && res && gama && enhance && threshold && dx && dres && dgama && denhance && dthreshold, "igate_res_bwd: null pointer");
  ADNM_REQUIRE(B > 0 && L > 0 && C > 0 && C % 4 == 0 && B * L * C < (1ll << 31), "igate_res_bwd: bad shape B=%lld L=%lld C=%lld (4 | C)", (long long)B,
               (long long)L, (long long)C);
  if (!ws || ws_bytes < adnm_igate_res_bwd_ws_bytes(B, L, C)) {
    adnm_set_error("igate_res_bwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_igate_res_bwd_ws_bytes(B, L, C));
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t blocks = igate_res_blocks(B, C);
  float* spart = (float*)ws;
  {
    ADNM_PROF("igate_bwd", st, 4.0 * (3.0 * B * L * C + 2.0 * B * C));
    igate_res_bwd_kernel<<<(unsigned)blocks, kBlock, 0, st>>>(dy, x, res, gama, enhance, threshold, dx, dres, spart, (int)B, (int)L, (int)(C / 4),
                                                               per_token);
  }
  ADNM_CHECK_LAUNCH("igate_res_bwd");
  adnm_launch_fold("igate_bwd_fold", spart, (int)blocks, 4, {dgama, 1}, {denhance, 1}, {dthreshold, 1}, {nullptr, 0}, st);   // deferrable
  ADNM_CHECK_LAUNCH("igate_res_bwd");
  return ADNM_OK;
}

// Elementwise product of two token matrices (row strides lda / ldb): VSSD's gate LayerNorm(y) * z (Vssd.py:280-281), one pass each way.
namespace {
__global__ __launch_bounds__(kBlock) void emul_fwd_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                          float* __restrict__ y, int64_t M, int C4) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M * C4; i += (int64_t)gridDim.x * kBlock) {
    const int64_t m = i / C4;
    const int c = (int)(i - m * C4) * 4;
    const float4 u = Io<float>::ld4(a + m * lda + c), v = Io<float>::ld4(b + m * ldb + c);
    Io<float>::st4(y + i * 4, make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w));
  }
}
__global__ __launch_bounds__(kBlock) void emul_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a, int64_t lda,
                                                          const float* __restrict__ b, int64_t ldb, float* __restrict__ da, float* __restrict__ db,
                                                          int64_t M, int C4) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M * C4; i += (int64_t)gridDim.x * kBlock) {
    const int64_t m = i / C4;
    const int c = (int)(i - m * C4) * 4;
    const float4 g = Io<float>::ld4(dy + i * 4), u = Io<float>::ld4(a + m * lda + c), v = Io<float>::ld4(b + m * ldb + c);
    Io<float>::st4(da + i * 4, make_float4(g.x * v.x, g.y * v.y, g.z * v.z, g.w * v.w));
    Io<float>::st4(db + i * 4, make_float4(g.x * u.x, g.y * u.y, g.z * u.z, g.w * u.w));
  }
}
}  // namespace

extern "C" int adnm_emul_fwd(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t M, int64_t C, adnm_stream_t stream) {
  ADNM_REQUIRE(a && b && y && M > 0 && C > 0 && C % 4 == 0 && lda >= C && ldb >= C && lda % 4 == 0 && ldb % 4 == 0, "emul_fwd: bad arguments (4 | C, 4 | strides)");
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("emul_fwd", st, 12.0 * M * C);
  emul_fwd_kernel<<<grid_for(M * C / 4), kBlock, 0, st>>>(a, lda, b, ldb, y, M, (int)(C / 4));
  ADNM_CHECK_LAUNCH("emul_fwd");
  return ADNM_OK;
}
extern "C" int adnm_emul_bwd(const float* dy, const float* a, int64_t lda, const float* b, int64_t ldb, float* da, float* db, int64_t M, int64_t C,
                             adnm_stream_t stream) {
  ADNM_REQUIRE(dy && a && b && da && db && M > 0 && C > 0 && C % 4 == 0 && lda >= C && ldb >= C && lda % 4 == 0 && ldb % 4 == 0, "emul_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("emul_bwd", st, 20.0 * M * C);
  emul_bwd_kernel<<<grid_for(M * C / 4), kBlock, 0, st>>>(dy, a, lda, b, ldb, da, db, M, (int)(C / 4));
  ADNM_CHECK_LAUNCH("emul_bwd");
  return ADNM_OK;
}

// Channel pad / crop of a token matrix: y[m, c] = x[m, c] for c < min(Cin, Cout), 0 for Cin <= c < Cout.  The 5-frame input stage is run on
// 8 channels (every stencil kernel moves 16-byte channel quads): pad on the way in, crop on the way out, and each is the other's backward.
namespace {
__global__ __launch_bounds__(kBlock) void chancopy_kernel(const float* __restrict__ x, int64_t ldx, int Cin, float* __restrict__ y, int Cout, int64_t M) {
  const int64_t total = M * Cout;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t m = i / Cout;
    const int c = (int)(i - m * Cout);
    y[i] = c < Cin ? x[m * ldx + c] : 0.f;
  }
}
}  // namespace

extern "C" int adnm_chancopy(const float* x, int64_t ldx, int64_t Cin, float* y, int64_t Cout, int64_t M, adnm_stream_t stream) {
  ADNM_REQUIRE(x && y && M > 0 && Cin > 0 && Cout > 0 && ldx >= Cin && M * Cout < (1ll << 40), "chancopy: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("chancopy", st, 4.0 * M * ((Cin < Cout ? Cin : Cout) + Cout));
  chancopy_kernel<<<grid_for(M * Cout), kBlock, 0, st>>>(x, ldx, (int)Cin, y, (int)Cout, M);
  ADNM_CHECK_LAUNCH("chancopy");
  return ADNM_OK;
}

namespace {
unsigned catmix_blocks(int64_t M, int64_t d) {
  const int64_t g = adnm_cdiv(M * (d / 4), kBlock);
  return (unsigned)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}
int catmix_check(const char* who, const void* x, int64_t ldx, const void* r, int64_t ldr, const void* f, int64_t ldf, int64_t M, int64_t d, int dtype) {
  ADNM_REQUIRE(x && r, "%s: null pointer", who);
  ADNM_REQUIRE(M > 0 && d >= 4 && d % 4 == 0, "%s: d=%lld must be a positive multiple of 4", who, (long long)d);
  ADNM_REQUIRE(ldx >= d && ldr >= d && ldx % 4 == 0 && ldr % 4 == 0 && (!f || (ldf >= d && ldf % 4 == 0)), "%s: bad row stride", who);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "%s: bad dtype %d", who, dtype);
  return ADNM_OK;
}
}  // namespace

extern "C" int adnm_catmix_fwd(const void* x, int64_t ldx, const void* r, int64_t ldr, const void* f, int64_t ldf, const float* a1, const float* a2,
                               const float* a3, const float* a4, void* y, int64_t M, int64_t d, int dtype, adnm_stream_t stream) {
  if (int rc = catmix_check("catmix_fwd", x, ldx, r, ldr, f, ldf, M, d, dtype)) return rc;
  ADNM_REQUIRE(y, "catmix_fwd: null output");
  hipStream_t st = (hipStream_t)stream;
  const double es = dtype == ADNM_F32 ? 4.0 : 2.0;
  ADNM_PROF("catmix_fwd", st, es * M * d * (f ? 5 : 4));
  if (dtype == ADNM_F32)
    catmix_fwd_kernel<float><<<grid_for(M * (d / 4)), kBlock, 0, st>>>((const float*)x, ldx, (const float*)r, ldr, (const float*)f, ldf, a1, a2, a3, a4,
                                                                       (float*)y, M, (int)d);
  else
    catmix_fwd_kernel<uint16_t><<<grid_for(M * (d / 4)), kBlock, 0, st>>>((const uint16_t*)x, ldx, (const uint16_t*)r, ldr, (const uint16_t*)f, ldf,
                                                                          a1, a2, a3, a4, (uint16_t*)y, M, (int)d);
  ADNM_CHECK_LAUNCH("catmix_fwd");
  return ADNM_OK;
}

extern "C" int64_t adnm_catmix_bwd_ws_bytes(int64_t M, int64_t d) { return (int64_t)catmix_blocks(M, d) * 4 * sizeof(float); }

extern "C" int adnm_catmix_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* r, int64_t ldr, const void* f, int64_t ldf,
                               const float* a1, const float* a2, const float* a3, const float* a4, void* dx, void* dr, void* df, float* da,
                               void* ws, int64_t ws_bytes, int64_t M, int64_t d, int dtype, adnm_stream_t stream) {
  if (int rc = catmix_check("catmix_bwd", x, ldx, r, ldr, f, ldf, M, d, dtype)) return rc;
  ADNM_REQUIRE(dy && da && lddy >= 2 * d && lddy % 4 == 0, "catmix_bwd: dy must be (M, 2d) with a row stride that is a multiple of 4");
  if (!ws || ws_bytes < adnm_catmix_bwd_ws_bytes(M, d)) {
    adnm_set_error("catmix_bwd: workspace too small");
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = catmix_blocks(M, d);
  const double es = dtype == ADNM_F32 ? 4.0 : 2.0;
  {
    ADNM_PROF("catmix_bwd", st, es * M * d * (f ? 8 : 6));
    if (dtype == ADNM_F32)
      catmix_bwd_kernel<float><<<grid, kBlock, 0, st>>>((const float*)dy, lddy, (const float*)x, ldx, (const float*)r, ldr, (const float*)f, ldf, a1, a2,
                                                        a3, a4, (float*)dx, (float*)dr, (float*)df, (float*)ws, M, (int)d);
    else
      catmix_bwd_kernel<uint16_t><<<grid, kBlock, 0, st>>>((const uint16_t*)dy, lddy, (const uint16_t*)x, ldx, (const uint16_t*)r, ldr,
                                                           (const uint16_t*)f, ldf, a1, a2, a3, a4, (uint16_t*)dx, (uint16_t*)dr, (uint16_t*)df,
                                                           (float*)ws, M, (int)d);
  }
  adnm_launch_fold("catmix_bwd_fold", (const float*)ws, (int)grid, 4, {da, 4}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("catmix_bwd");
  return ADNM_OK;
}

namespace {
unsigned rainloss_blocks(int64_t n) {
  const int64_t g = adnm_cdiv(n, kBlock * 4);
  return (unsigned)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}
}  // namespace
extern "C" int64_t adnm_rainloss_ws_bytes(int64_t n) { return (int64_t)rainloss_blocks(n) * sizeof(float); }

extern "C" int adnm_rainloss(const float* pred, const float* target, float* loss, float* grad, void* ws, int64_t ws_bytes, int64_t n,
                             float omega_t, float alpha, float gamma, adnm_stream_t stream) {
  ADNM_REQUIRE(pred && target && loss && grad, "rainloss: null pointer");
  ADNM_REQUIRE(n > 0, "rainloss: empty input");
  if (!ws || ws_bytes < adnm_rainloss_ws_bytes(n)) {
    adnm_set_error("rainloss: workspace too small");
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = rainloss_blocks(n);
  {
    ADNM_PROF("rainloss", st, 12.0 * n);
    rainloss_kernel<<<grid, kBlock, 0, st>>>(pred, target, grad, (float*)ws, n, omega_t, alpha, gamma);
  }
  adnm_launch_fold("rainloss_fold", (const float*)ws, (int)grid, 1, {loss, 1}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("rainloss");
  return ADNM_OK;
}

extern "C" int adnm_conv1d3_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t n, adnm_stream_t stream) {
  ADNM_REQUIRE(x && w && y, "conv1d3_fwd: null pointer");
  ADNM_REQUIRE(B > 0 && n > 0 && B * n <= (1 << 20), "conv1d3_fwd: B*n = %lld outside (0, 2^20]", (long long)(B * n));
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("conv1d3", st, 8.0 * B * n);
  conv1d3_kernel<<<1, kC1Threads, 0, st>>>(x, w, bias, nullptr, y, nullptr, nullptr, (int)B, (int)n);
  ADNM_CHECK_LAUNCH("conv1d3_fwd");
  return ADNM_OK;
}

extern "C" int adnm_conv1d3_bwd(const float* dy, const float* x, const float* w, float* dx, float* dwb, int64_t B, int64_t n, adnm_stream_t stream) {
  ADNM_REQUIRE(dy && x && w && dx && dwb, "conv1d3_bwd: null pointer");
  ADNM_REQUIRE(B > 0 && n > 0 && B * n <= (1 << 20), "conv1d3_bwd: B*n = %lld outside (0, 2^20]", (long long)(B * n));
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("conv1d3_bwd", st, 12.0 * B * n);
  conv1d3_kernel<<<1, kC1Threads, 0, st>>>(x, w, nullptr, dy, nullptr, dx, dwb, (int)B, (int)n);
  ADNM_CHECK_LAUNCH("conv1d3_bwd");
  return ADNM_OK;
}
