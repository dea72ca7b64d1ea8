// Row norms over the channel axis of token tensors: RMSNorm / LayerNorm / BiasFree_LayerNorm with the
// scalar affine of Block.forward fused (reference: mamba_ssm RMSNorm bound at ADNMUNet.py:278 and used
// at :149,155; nn.LayerNorm at ADNssd.py:456 / Vssd.py:280; BiasFree_LayerNorm model_untils.py:43-48).
//
// HBM-bound: 2*M*d elements forward, 3*M*d backward.  A row is owned by LPR = min(64, pow2(d/4)) adjacent
// lanes, 16 B per lane per step, so a wave64 covers 64/LPR rows with fully coalesced 1 KiB accesses
// (d=32 -> 8 rows per wave).  Statistics by xor-shuffles inside the LPR group; no LDS in forward.
// Backward is one persistent pass: per-lane fp32 accumulators for dw/db/dscale/dshift over a grid-stride
// row loop, one LDS fold per block, per-block partials to the workspace, and a tiny deterministic
// finalize kernel (no atomics -> bitwise reproducible gradients).
#include "adnm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxPartBlocks = 1024;  // 4 workgroups per CU: enough waves in flight to stream at HBM rate (256 left 1 per CU: ~2 TB/s)

__host__ __device__ inline int lanes_per_row(int64_t d) {
  int l = 1;
  while (l < 64 && (int64_t)l * 4 < d) l <<= 1;
  return l;
}

template <typename T, bool MEAN, int IT>
__global__ __launch_bounds__(kBlock) void rownorm_fwd_kernel(const T* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ b,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ y,
                                                             int64_t ldy, float* __restrict__ mu_out,
                                                             float* __restrict__ rstd_out, int64_t M, int d,
                                                             float eps, int lpr) {
  const int lane_in_row = threadIdx.x & (lpr - 1);
  const int rows_per_block = kBlock / lpr;
  const int64_t row = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x / lpr);
  const bool live = row < M;
  const float sc = scale ? *scale : 1.0f;
  const float sh = shift ? *shift : 0.0f;
  float4 v[IT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (i * lpr + lane_in_row) * 4;
    v[i] = (live && c < d) ? Io<T>::ld4(x + row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += MEAN ? (v[i].x + v[i].y + v[i].z + v[i].w) : (v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w);
  }
  s = group_sum(s, lpr);
  float mu = 0.f, var;
  if (MEAN) {
    mu = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (i * lpr + lane_in_row) * 4;
      if (c < d) {
        float a0 = v[i].x - mu, a1 = v[i].y - mu, a2 = v[i].z - mu, a3 = v[i].w - mu;
        q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
      }
    }
    var = group_sum(q, lpr) / (float)d;
  } else {
    var = s / (float)d;
  }
  const float rstd = rsqrtf(var + eps);
  if (live && lane_in_row == 0) {
    if (MEAN) mu_out[row] = mu;
    rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (i * lpr + lane_in_row) * 4;
    if (live && c < d) {
      const float4 ww = *reinterpret_cast<const float4*>(w + c);
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b) bb = *reinterpret_cast<const float4*>(b + c);
      float4 o;
      o.x = sc * ((v[i].x - mu) * rstd * ww.x + bb.x) + sh;
      o.y = sc * ((v[i].y - mu) * rstd * ww.y + bb.y) + sh;
      o.z = sc * ((v[i].z - mu) * rstd * ww.z + bb.z) + sh;
      o.w = sc * ((v[i].w - mu) * rstd * ww.w + bb.w) + sh;
      Io<T>::st4(y + row * ldy + c, o);
    }
  }
}

// partial layout per block: [dw(d) | db(d) | dscale | dshift]
template <typename T, bool MEAN, int IT>
__global__ __launch_bounds__(kBlock) void rownorm_bwd_kernel(
    const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx, const float* __restrict__ w,
    const float* __restrict__ b, const float* __restrict__ scale, const float* __restrict__ mu_in,
    const float* __restrict__ rstd_in, T* __restrict__ dx, int64_t lddx, const T* __restrict__ dres, int64_t lddres,
    float* __restrict__ part, int64_t M, int d, int lpr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // (waves) x (2d+4)
  const int lane_in_row = threadIdx.x & (lpr - 1);
  const int rows_per_block = kBlock / lpr;
  const float sc = scale ? *scale : 1.0f;
  float4 aw[IT], ab[IT], ww[IT], bb[IT];
  float a_scale = 0.f, a_shift = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    aw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (i * lpr + lane_in_row) * 4;
    ww[i] = c < d ? *reinterpret_cast<const float4*>(w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    bb[i] = (b && c < d) ? *reinterpret_cast<const float4*>(b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x / lpr); row < M;
       row += (int64_t)gridDim.x * rows_per_block) {
    const float mu = MEAN ? mu_in[row] : 0.f;
    const float rstd = rstd_in[row];
    float4 g[IT], xh[IT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (i * lpr + lane_in_row) * 4;
      if (c < d) {
        const float4 xv = Io<T>::ld4(x + row * ldx + c);
        const float4 dv = Io<T>::ld4(dy + row * lddy + c);
        xh[i] = make_float4((xv.x - mu) * rstd, (xv.y - mu) * rstd, (xv.z - mu) * rstd, (xv.w - mu) * rstd);
        a_shift += dv.x + dv.y + dv.z + dv.w;
        a_scale += dv.x * (xh[i].x * ww[i].x + bb[i].x) + dv.y * (xh[i].y * ww[i].y + bb[i].y) +
                   dv.z * (xh[i].z * ww[i].z + bb[i].z) + dv.w * (xh[i].w * ww[i].w + bb[i].w);
        const float4 du = make_float4(dv.x * sc, dv.y * sc, dv.z * sc, dv.w * sc);
        ab[i].x += du.x; ab[i].y += du.y; ab[i].z += du.z; ab[i].w += du.w;
        aw[i].x += du.x * xh[i].x; aw[i].y += du.y * xh[i].y; aw[i].z += du.z * xh[i].z; aw[i].w += du.w * xh[i].w;
        g[i] = make_float4(du.x * ww[i].x, du.y * ww[i].y, du.z * ww[i].z, du.w * ww[i].w);
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      } else {
        g[i] = xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    s2 = group_sum(s2, lpr) / (float)d;
    s1 = MEAN ? group_sum(s1, lpr) / (float)d : 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (i * lpr + lane_in_row) * 4;
      if (c < d) {
        float4 o;
        o.x = rstd * (g[i].x - s1 - xh[i].x * s2);
        o.y = rstd * (g[i].y - s1 - xh[i].y * s2);
        o.z = rstd * (g[i].z - s1 - xh[i].z * s2);
        o.w = rstd * (g[i].w - s1 - xh[i].w * s2);
        if (dres) {  // gradient arriving through the residual path of the pre-norm block
          const float4 e = Io<T>::ld4(dres + row * lddres + c);
          o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
        }
        Io<T>::st4(dx + row * lddx + c, o);
      }
    }
  }
  // fold the row groups that share a wave, then the waves of the block (LDS), then write the partial
  const int stride = 2 * d + 2;
  const int sstride = 2 * d + 4;  // LDS row stride, keeps float4 slots 16-B aligned
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    aw[i].x = wave_sum_from(aw[i].x, lpr); aw[i].y = wave_sum_from(aw[i].y, lpr);
    aw[i].z = wave_sum_from(aw[i].z, lpr); aw[i].w = wave_sum_from(aw[i].w, lpr);
    ab[i].x = wave_sum_from(ab[i].x, lpr); ab[i].y = wave_sum_from(ab[i].y, lpr);
    ab[i].z = wave_sum_from(ab[i].z, lpr); ab[i].w = wave_sum_from(ab[i].w, lpr);
    const int c = (i * lpr + lane_in_row) * 4;
    if (lane < lpr && c < d) {
      *reinterpret_cast<float4*>(smem + wave * sstride + c) = aw[i];
      *reinterpret_cast<float4*>(smem + wave * sstride + d + c) = ab[i];
    }
  }
  a_scale = wave_sum(a_scale);
  a_shift = wave_sum(a_shift);
  if (lane == 0) {
    smem[wave * sstride + 2 * d] = a_scale;
    smem[wave * sstride + 2 * d + 1] = a_shift;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < stride; c += kBlock) {
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; ++wv) t += smem[wv * sstride + c];
    part[(int64_t)blockIdx.x * stride + c] = t;
  }
}

int bwd_blocks(int64_t M, int64_t d) {
  const int lpr = lanes_per_row(d);
  int64_t nb = adnm_cdiv(M, kBlock / lpr);
  return (int)(nb < kMaxPartBlocks ? nb : kMaxPartBlocks);
}

template <typename T, bool MEAN>
int launch_fwd(const void* x, int64_t ldx, const float* w, const float* b, const float* scale, const float* shift,
               void* y, int64_t ldy, float* mu, float* rstd, int64_t M, int64_t d, float eps, hipStream_t st) {
  const int lpr = lanes_per_row(d);
  const int it = (int)adnm_cdiv(d, (int64_t)lpr * 4);
  const dim3 grid((unsigned)adnm_cdiv(M, kBlock / lpr));
#define FWD(IT)                                                                                               \
  rownorm_fwd_kernel<T, MEAN, IT><<<grid, kBlock, 0, st>>>((const T*)x, ldx, w, b, scale, shift, (T*)y, ldy, mu, \
                                                            rstd, M, (int)d, eps, lpr)
  ADNM_PROF("rownorm_fwd", st, (double)sizeof(T) * M * d * 2);
  if (it <= 1) FWD(1);
  else if (it <= 2) FWD(2);
  else if (it <= 4) FWD(4);
  else if (it <= 8) FWD(8);
  else FWD(16);
#undef FWD
  ADNM_CHECK_LAUNCH("rownorm_fwd");
  return ADNM_OK;
}

template <typename T, bool MEAN>
int launch_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* w, const float* b,
               const float* scale, const float* mu, const float* rstd, void* dx, int64_t lddx, float* dw, float* db,
               float* dscale, float* dshift, const void* dres, int64_t lddres, float* part, int64_t M, int64_t d, hipStream_t st) {
  const int lpr = lanes_per_row(d);
  const int it = (int)adnm_cdiv(d, (int64_t)lpr * 4);
  const int nblk = bwd_blocks(M, d);
  const size_t smem = (size_t)(kBlock / 64) * (2 * d + 4) * sizeof(float);
#define BWD(IT)                                                                                                  \
  rownorm_bwd_kernel<T, MEAN, IT><<<nblk, kBlock, smem, st>>>((const T*)dy, lddy, (const T*)x, ldx, w, b, scale, mu, \
                                                              rstd, (T*)dx, lddx, (const T*)dres, lddres, part, M, (int)d, lpr)
  {
  ADNM_PROF("rownorm_bwd", st, (double)sizeof(T) * M * d * (dres ? 4 : 3));
  if (it <= 1) BWD(1);
  else if (it <= 2) BWD(2);
  else if (it <= 4) BWD(4);
  else if (it <= 8) BWD(8);
  else BWD(16);
#undef BWD
  }
  ADNM_CHECK_LAUNCH("rownorm_bwd");
  adnm_launch_fold("rownorm_bwd_fold", part, nblk, (int)(2 * d + 2), {dw, (int)d}, {db, (int)d}, {dscale, 1}, {dshift, 1}, st);
  ADNM_CHECK_LAUNCH("rownorm_bwd_finalize");
  return ADNM_OK;
}

}  // namespace

extern "C" int adnm_rownorm_fwd(const void* x, int64_t ldx, const float* w, const float* b, const float* scale,
                                const float* shift, void* y, int64_t ldy, float* mu, float* rstd, int64_t M,
                                int64_t d, float eps, int subtract_mean, int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(x && w && y && rstd, "rownorm_fwd: null pointer");
  ADNM_REQUIRE(!subtract_mean || mu, "rownorm_fwd: mu buffer required when subtract_mean");
  ADNM_REQUIRE(M >= 0 && d >= 4 && d % 4 == 0 && d <= 4096, "rownorm_fwd: d=%lld must be a multiple of 4 in [4,4096]", (long long)d);
  ADNM_REQUIRE(ldx >= d && ldy >= d && ldx % 4 == 0 && ldy % 4 == 0, "rownorm_fwd: row strides must be >= d and multiples of 4");
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "rownorm_fwd: bad dtype %d", dtype);
  if (M == 0) return ADNM_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ADNM_F32)
    return subtract_mean ? launch_fwd<float, true>(x, ldx, w, b, scale, shift, y, ldy, mu, rstd, M, d, eps, st)
                         : launch_fwd<float, false>(x, ldx, w, b, scale, shift, y, ldy, mu, rstd, M, d, eps, st);
  return subtract_mean ? launch_fwd<uint16_t, true>(x, ldx, w, b, scale, shift, y, ldy, mu, rstd, M, d, eps, st)
                       : launch_fwd<uint16_t, false>(x, ldx, w, b, scale, shift, y, ldy, mu, rstd, M, d, eps, st);
}

extern "C" int64_t adnm_rownorm_bwd_ws_bytes(int64_t M, int64_t d) {
  if (M <= 0 || d <= 0) return 0;
  return (int64_t)bwd_blocks(M, d) * (2 * d + 2) * (int64_t)sizeof(float);
}

extern "C" int adnm_rownorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* w,
                                const float* b, const float* scale, const float* mu, const float* rstd, void* dx,
                                int64_t lddx, float* dw, float* db, float* dscale, float* dshift, const void* dres,
                                int64_t lddres, void* ws, int64_t ws_bytes, int64_t M, int64_t d, int subtract_mean,
                                int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(dy && x && w && rstd && dx && dw, "rownorm_bwd: null pointer");
  ADNM_REQUIRE(!subtract_mean || mu, "rownorm_bwd: mu required when subtract_mean");
  ADNM_REQUIRE(M > 0 && d >= 4 && d % 4 == 0 && d <= 4096, "rownorm_bwd: d=%lld must be a multiple of 4 in [4,4096]", (long long)d);
  ADNM_REQUIRE(ldx >= d && lddy >= d && lddx >= d && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0,
               "rownorm_bwd: row strides must be >= d and multiples of 4");
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "rownorm_bwd: bad dtype %d", dtype);
  ADNM_REQUIRE(!dres || (lddres >= d && lddres % 4 == 0), "rownorm_bwd: bad residual-gradient stride");
  if (ws_bytes < adnm_rownorm_bwd_ws_bytes(M, d) || !ws) {
    adnm_set_error("rownorm_bwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_rownorm_bwd_ws_bytes(M, d));
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  if (dtype == ADNM_F32)
    return subtract_mean ? launch_bwd<float, true>(dy, lddy, x, ldx, w, b, scale, mu, rstd, dx, lddx, dw, db, dscale, dshift, dres, lddres, part, M, d, st)
                         : launch_bwd<float, false>(dy, lddy, x, ldx, w, b, scale, mu, rstd, dx, lddx, dw, db, dscale, dshift, dres, lddres, part, M, d, st);
  return subtract_mean ? launch_bwd<uint16_t, true>(dy, lddy, x, ldx, w, b, scale, mu, rstd, dx, lddx, dw, db, dscale, dshift, dres, lddres, part, M, d, st)
                       : launch_bwd<uint16_t, false>(dy, lddy, x, ldx, w, b, scale, mu, rstd, dx, lddx, dw, db, dscale, dshift, dres, lddres, part, M, d, st);
}
