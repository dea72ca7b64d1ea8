// Residual mix + the NEXT sub-block's pre-norm in one pass each way (reference ADNMUNet.py:149-158: x = beta1*x + beta2*mixer(norm1(x));
// xn = scale2*norm2(x) + shift2; ... and the same pair around the FFN, :226-232 in Attention):
//
//   mix[m,c] = gamma[c] * (s0*x0[m,c] + s1*x1[m,c])                    (lincomb.hip)
//   xn[m,c]  = scale * ((mix - mu_m) * rstd_m * w[c] + b[c]) + shift    (rownorm.hip; RMSNorm when !MEAN)
//
// The two kernels this replaces are each a few microseconds of launch floor at the deep stages and stream the same rows at the wide
// ones.  Forward writes mix AND xn from one read of x0, x1 (the value of `mix` is bitwise the one lincomb_fwd stores); backward takes
// d xn and the gradient arriving on mix through the residual path, recomputes mix from x0, x1 (which it needs for d s_k anyway), and
// writes d x0, d x1 directly: the intermediate d mix never exists in memory.  fp32 token rows; same lane mapping (LPR lanes per
// row, 16 B per lane per step) and deterministic two-stage parameter-gradient reduction as rownorm.hip / lincomb.hip.
#include "adnm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxPartBlocks = 1024;

inline int lanes_per_row(int64_t d) {
  int l = 1;
  while (l < 64 && (int64_t)l * 4 < d) l <<= 1;
  return l;
}

struct MixIn {
  const float* x[2];
  int64_t ld[2];
  const float* s[2];
  const float* gamma;
  const float *w, *b, *scale, *shift;
};

// products that must round exactly as the forward's stores did: keep the compiler from contracting them into a neighbouring add
__device__ __forceinline__ float4 mul4_rn(float4 a, float4 b) { return make_float4(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y), __fmul_rn(a.z, b.z), __fmul_rn(a.w, b.w)); }

template <bool MEAN, int IT>
__global__ __launch_bounds__(kBlock) void mixnorm_fwd_kernel(MixIn in, float* __restrict__ ymix, int64_t ldm, float* __restrict__ yn, int64_t ldn,
                                                             float* __restrict__ mu_out, float* __restrict__ rstd_out, int64_t M, int d, float eps,
                                                             int lpr) {
  const int lane_in_row = threadIdx.x & (lpr - 1);
  const int rows_per_block = kBlock / lpr;
  const int64_t row = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x / lpr);
  const bool live = row < M;
  const float s0 = in.s[0] ? *in.s[0] : 1.f, s1 = in.s[1] ? *in.s[1] : 1.f;
  const float sc = in.scale ? *in.scale : 1.0f, sh = in.shift ? *in.shift : 0.0f;
  float4 v[IT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (i * lpr + lane_in_row) * 4;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && c < d) {
      const float4 a = *reinterpret_cast<const float4*>(in.x[0] + row * in.ld[0] + c), e = *reinterpret_cast<const float4*>(in.x[1] + row * in.ld[1] + c);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);   // the operation order of lincomb_fwd_kernel
      acc.x = fmaf(s0, a.x, acc.x), acc.y = fmaf(s0, a.y, acc.y), acc.z = fmaf(s0, a.z, acc.z), acc.w = fmaf(s0, a.w, acc.w);
      acc.x = fmaf(s1, e.x, acc.x), acc.y = fmaf(s1, e.y, acc.y), acc.z = fmaf(s1, e.z, acc.z), acc.w = fmaf(s1, e.w, acc.w);
      if (in.gamma) acc = mul4_rn(acc, *reinterpret_cast<const float4*>(in.gamma + c));
      v[i] = acc;
      *reinterpret_cast<float4*>(ymix + row * ldm + c) = acc;
    }
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
        const float a0 = v[i].x - mu, a1 = v[i].y - mu, a2 = v[i].z - mu, a3 = v[i].w - mu;
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
      const float4 ww = *reinterpret_cast<const float4*>(in.w + c);
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in.b) bb = *reinterpret_cast<const float4*>(in.b + c);
      float4 o;
      o.x = sc * ((v[i].x - mu) * rstd * ww.x + bb.x) + sh;
      o.y = sc * ((v[i].y - mu) * rstd * ww.y + bb.y) + sh;
      o.z = sc * ((v[i].z - mu) * rstd * ww.z + bb.z) + sh;
      o.w = sc * ((v[i].w - mu) * rstd * ww.w + bb.w) + sh;
      *reinterpret_cast<float4*>(yn + row * ldn + c) = o;
    }
  }
}

struct MixGrads {
  float* dx[2];
  int64_t ld[2];
};

// partials per workgroup: part_mix [dgamma(d) | ds0 | ds1 | 0]  (the layout of lincomb_bwd),  part_norm [dw(d) | db(d) | dscale | dshift]
template <bool MEAN, int IT>
__global__ __launch_bounds__(kBlock) void mixnorm_bwd_kernel(const float* __restrict__ dyn, int64_t lddyn, const float* __restrict__ dres, int64_t lddres,
                                                             MixIn in, const float* __restrict__ mu_in, const float* __restrict__ rstd_in, MixGrads gr,
                                                             float* __restrict__ part_mix, float* __restrict__ part_norm, int64_t M, int d, int lpr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // waves x (3d + 8)
  const int lane_in_row = threadIdx.x & (lpr - 1);
  const int rows_per_block = kBlock / lpr;
  const float s0 = in.s[0] ? *in.s[0] : 1.f, s1 = in.s[1] ? *in.s[1] : 1.f;
  const float sc = in.scale ? *in.scale : 1.0f;
  float4 aw[IT], ab[IT], ag[IT], ww[IT], bb[IT], gm[IT];
  float a_scale = 0.f, a_shift = 0.f, ds0 = 0.f, ds1 = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    aw[i] = ab[i] = ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (i * lpr + lane_in_row) * 4;
    ww[i] = c < d ? *reinterpret_cast<const float4*>(in.w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    bb[i] = (in.b && c < d) ? *reinterpret_cast<const float4*>(in.b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    gm[i] = (in.gamma && c < d) ? *reinterpret_cast<const float4*>(in.gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  for (int64_t row = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x / lpr); row < M; row += (int64_t)gridDim.x * rows_per_block) {
    const float mu = MEAN ? mu_in[row] : 0.f;
    const float rstd = rstd_in[row];
    float4 xa[IT], xb[IT], dv[IT], er[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {   // every load of the row first
      const int c = (i * lpr + lane_in_row) * 4;
      const bool ok = c < d;
      const int cc = ok ? c : 0;
      xa[i] = *reinterpret_cast<const float4*>(in.x[0] + row * in.ld[0] + cc);
      xb[i] = *reinterpret_cast<const float4*>(in.x[1] + row * in.ld[1] + cc);
      dv[i] = *reinterpret_cast<const float4*>(dyn + row * lddyn + cc);
      er[i] = dres ? *reinterpret_cast<const float4*>(dres + row * lddres + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 g[IT], xh[IT], pre[IT];
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (i * lpr + lane_in_row) * 4;
      if (c < d) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        acc.x = fmaf(s0, xa[i].x, acc.x), acc.y = fmaf(s0, xa[i].y, acc.y), acc.z = fmaf(s0, xa[i].z, acc.z), acc.w = fmaf(s0, xa[i].w, acc.w);
        acc.x = fmaf(s1, xb[i].x, acc.x), acc.y = fmaf(s1, xb[i].y, acc.y), acc.z = fmaf(s1, xb[i].z, acc.z), acc.w = fmaf(s1, xb[i].w, acc.w);
        pre[i] = acc;
        const float4 mix = in.gamma ? mul4_rn(acc, gm[i]) : acc;   // bitwise what the forward stored
        xh[i] = make_float4(__fsub_rn(mix.x, mu) * rstd, __fsub_rn(mix.y, mu) * rstd, __fsub_rn(mix.z, mu) * rstd, __fsub_rn(mix.w, mu) * rstd);
        const float4 q = dv[i];
        a_shift += q.x + q.y + q.z + q.w;
        a_scale += q.x * (xh[i].x * ww[i].x + bb[i].x) + q.y * (xh[i].y * ww[i].y + bb[i].y) + q.z * (xh[i].z * ww[i].z + bb[i].z) +
                   q.w * (xh[i].w * ww[i].w + bb[i].w);
        const float4 du = make_float4(q.x * sc, q.y * sc, q.z * sc, q.w * sc);
        ab[i].x += du.x, ab[i].y += du.y, ab[i].z += du.z, ab[i].w += du.w;
        aw[i].x += du.x * xh[i].x, aw[i].y += du.y * xh[i].y, aw[i].z += du.z * xh[i].z, aw[i].w += du.w * xh[i].w;
        g[i] = make_float4(du.x * ww[i].x, du.y * ww[i].y, du.z * ww[i].z, du.w * ww[i].w);
        t1 += g[i].x + g[i].y + g[i].z + g[i].w;
        t2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      } else {
        g[i] = xh[i] = pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    t2 = group_sum(t2, lpr) / (float)d;
    t1 = MEAN ? group_sum(t1, lpr) / (float)d : 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (i * lpr + lane_in_row) * 4;
      if (c < d) {
        float4 G;   // d mix: the row norm's input gradient + what arrives on the residual path
        G.x = rstd * (g[i].x - t1 - xh[i].x * t2) + er[i].x;
        G.y = rstd * (g[i].y - t1 - xh[i].y * t2) + er[i].y;
        G.z = rstd * (g[i].z - t1 - xh[i].z * t2) + er[i].z;
        G.w = rstd * (g[i].w - t1 - xh[i].w * t2) + er[i].w;
        const float4 gg = make_float4(G.x * gm[i].x, G.y * gm[i].y, G.z * gm[i].z, G.w * gm[i].w);
        ds0 += gg.x * xa[i].x + gg.y * xa[i].y + gg.z * xa[i].z + gg.w * xa[i].w;
        ds1 += gg.x * xb[i].x + gg.y * xb[i].y + gg.z * xb[i].z + gg.w * xb[i].w;
        ag[i].x = fmaf(G.x, pre[i].x, ag[i].x), ag[i].y = fmaf(G.y, pre[i].y, ag[i].y), ag[i].z = fmaf(G.z, pre[i].z, ag[i].z), ag[i].w = fmaf(G.w, pre[i].w, ag[i].w);
        if (gr.dx[0]) *reinterpret_cast<float4*>(gr.dx[0] + row * gr.ld[0] + c) = make_float4(s0 * gg.x, s0 * gg.y, s0 * gg.z, s0 * gg.w);
        if (gr.dx[1]) *reinterpret_cast<float4*>(gr.dx[1] + row * gr.ld[1] + c) = make_float4(s1 * gg.x, s1 * gg.y, s1 * gg.z, s1 * gg.w);
      }
    }
  }
  // fold the row groups that share a wave, then the waves of the block (LDS), then write the two partial rows
  const int sstride = 3 * d + 8;   // [dw | db | dgamma | dscale dshift ds0 ds1 ...]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    aw[i].x = wave_sum_from(aw[i].x, lpr), aw[i].y = wave_sum_from(aw[i].y, lpr), aw[i].z = wave_sum_from(aw[i].z, lpr), aw[i].w = wave_sum_from(aw[i].w, lpr);
    ab[i].x = wave_sum_from(ab[i].x, lpr), ab[i].y = wave_sum_from(ab[i].y, lpr), ab[i].z = wave_sum_from(ab[i].z, lpr), ab[i].w = wave_sum_from(ab[i].w, lpr);
    ag[i].x = wave_sum_from(ag[i].x, lpr), ag[i].y = wave_sum_from(ag[i].y, lpr), ag[i].z = wave_sum_from(ag[i].z, lpr), ag[i].w = wave_sum_from(ag[i].w, lpr);
    const int c = (i * lpr + lane_in_row) * 4;
    if (lane < lpr && c < d) {
      *reinterpret_cast<float4*>(smem + wave * sstride + c) = aw[i];
      *reinterpret_cast<float4*>(smem + wave * sstride + d + c) = ab[i];
      *reinterpret_cast<float4*>(smem + wave * sstride + 2 * d + c) = ag[i];
    }
  }
  a_scale = wave_sum(a_scale), a_shift = wave_sum(a_shift), ds0 = wave_sum(ds0), ds1 = wave_sum(ds1);
  if (lane == 0) *reinterpret_cast<float4*>(smem + wave * sstride + 3 * d) = make_float4(a_scale, a_shift, ds0, ds1);
  __syncthreads();
  const int nn = 2 * d + 2, nm = d + 3;
  for (int c = threadIdx.x; c < nn + nm; c += kBlock) {
    // destination column -> LDS column
    int src;
    if (c < nn) src = c < 2 * d ? c : 3 * d + (c - 2 * d);            // dw, db | dscale, dshift
    else {
      const int e = c - nn;
      src = e < d ? 2 * d + e : (e < d + 2 ? 3 * d + 2 + (e - d) : -1);   // dgamma | ds0, ds1 | (unused third scalar)
    }
    float t = 0.f;
    if (src >= 0) {
#pragma unroll
      for (int wv = 0; wv < kBlock / 64; ++wv) t += smem[wv * sstride + src];
    }
    if (c < nn) part_norm[(int64_t)blockIdx.x * nn + c] = t;
    else part_mix[(int64_t)blockIdx.x * nm + (c - nn)] = t;
  }
}

int bwd_blocks(int64_t M, int64_t d) {
  const int lpr = lanes_per_row(d);
  const int64_t nb = adnm_cdiv(M, kBlock / lpr);
  return (int)(nb < kMaxPartBlocks ? nb : kMaxPartBlocks);
}

int check(const char* who, const float* x0, int64_t ld0, const float* x1, int64_t ld1, int64_t M, int64_t d) {
  ADNM_REQUIRE(x0 && x1, "%s: null operand", who);
  ADNM_REQUIRE(M > 0 && d >= 4 && d % 4 == 0 && d <= 1024, "%s: d=%lld must be a multiple of 4 in [4, 1024]", who, (long long)d);
  ADNM_REQUIRE(ld0 >= d && ld1 >= d && ld0 % 4 == 0 && ld1 % 4 == 0, "%s: row strides must be >= d and multiples of 4", who);
  return ADNM_OK;
}
}  // namespace

extern "C" int adnm_mixnorm_fwd(const float* x0, int64_t ld0, const float* x1, int64_t ld1, const float* s0, const float* s1, const float* gamma,
                                const float* w, const float* b, const float* scale, const float* shift, float* ymix, int64_t ldm, float* yn,
                                int64_t ldn, float* mu, float* rstd, int64_t M, int64_t d, float eps, int subtract_mean, adnm_stream_t stream) {
  if (int rc = check("mixnorm_fwd", x0, ld0, x1, ld1, M, d)) return rc;
  ADNM_REQUIRE(w && ymix && yn && rstd && (!subtract_mean || mu), "mixnorm_fwd: null pointer");
  ADNM_REQUIRE(ldm >= d && ldn >= d && ldm % 4 == 0 && ldn % 4 == 0, "mixnorm_fwd: output row strides must be >= d and multiples of 4");
  hipStream_t st = (hipStream_t)stream;
  const int lpr = lanes_per_row(d);
  const int it = (int)adnm_cdiv(d, (int64_t)lpr * 4);
  const dim3 grid((unsigned)adnm_cdiv(M, kBlock / lpr));
  MixIn in;
  in.x[0] = x0, in.x[1] = x1, in.ld[0] = ld0, in.ld[1] = ld1, in.s[0] = s0, in.s[1] = s1, in.gamma = gamma, in.w = w, in.b = b, in.scale = scale, in.shift = shift;
  ADNM_PROF("mixnorm_fwd", st, 4.0 * M * d * 4);
#define FWD(MEAN, IT) mixnorm_fwd_kernel<MEAN, IT><<<grid, kBlock, 0, st>>>(in, ymix, ldm, yn, ldn, mu, rstd, M, (int)d, eps, lpr)
  if (subtract_mean) {
    if (it <= 1) FWD(true, 1);
    else if (it <= 2) FWD(true, 2);
    else FWD(true, 4);
  } else {
    if (it <= 1) FWD(false, 1);
    else if (it <= 2) FWD(false, 2);
    else FWD(false, 4);
  }
#undef FWD
  ADNM_CHECK_LAUNCH("mixnorm_fwd");
  return ADNM_OK;
}

extern "C" int64_t adnm_mixnorm_bwd_ws_bytes(int64_t M, int64_t d) {
  if (M <= 0 || d <= 0) return 0;
  return (int64_t)bwd_blocks(M, d) * (3 * d + 5) * (int64_t)sizeof(float);
}

extern "C" int adnm_mixnorm_bwd(const float* dyn, int64_t lddyn, const float* dres, int64_t lddres, const float* x0, int64_t ld0, const float* x1,
                                int64_t ld1, const float* s0, const float* s1, const float* gamma, const float* w, const float* b, const float* scale,
                                const float* mu, const float* rstd, float* dx0, int64_t lddx0, float* dx1, int64_t lddx1, float* ds0, float* ds1,
                                float* dgamma, float* dw, float* db, float* dscale, float* dshift, void* ws, int64_t ws_bytes, int64_t M, int64_t d,
                                int subtract_mean, adnm_stream_t stream) {
  if (int rc = check("mixnorm_bwd", x0, ld0, x1, ld1, M, d)) return rc;
  ADNM_REQUIRE(dyn && w && rstd && (!subtract_mean || mu), "mixnorm_bwd: null pointer");
  ADNM_REQUIRE(lddyn >= d && lddyn % 4 == 0 && (!dres || (lddres >= d && lddres % 4 == 0)), "mixnorm_bwd: bad gradient strides");
  ADNM_REQUIRE((!dx0 || (lddx0 >= d && lddx0 % 4 == 0)) && (!dx1 || (lddx1 >= d && lddx1 % 4 == 0)), "mixnorm_bwd: bad input-gradient strides");
  if (!ws || ws_bytes < adnm_mixnorm_bwd_ws_bytes(M, d)) {
    adnm_set_error("mixnorm_bwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_mixnorm_bwd_ws_bytes(M, d));
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int lpr = lanes_per_row(d);
  const int it = (int)adnm_cdiv(d, (int64_t)lpr * 4);
  const int nblk = bwd_blocks(M, d);
  const size_t smem = (size_t)(kBlock / 64) * (3 * d + 8) * sizeof(float);
  float* part_mix = (float*)ws;                               // [nblk][d + 3]
  float* part_norm = part_mix + (int64_t)nblk * (d + 3);      // [nblk][2d + 2]
  MixIn in;
  in.x[0] = x0, in.x[1] = x1, in.ld[0] = ld0, in.ld[1] = ld1, in.s[0] = s0, in.s[1] = s1, in.gamma = gamma, in.w = w, in.b = b, in.scale = scale, in.shift = nullptr;
  MixGrads gr;
  gr.dx[0] = dx0, gr.dx[1] = dx1, gr.ld[0] = lddx0, gr.ld[1] = lddx1;
  {
    ADNM_PROF("mixnorm_bwd", st, 4.0 * M * d * (3 + (dres ? 1 : 0) + (dx0 ? 1 : 0) + (dx1 ? 1 : 0)));
#define BWD(MEAN, IT) \
  mixnorm_bwd_kernel<MEAN, IT><<<nblk, kBlock, smem, st>>>(dyn, lddyn, dres, lddres, in, mu, rstd, gr, part_mix, part_norm, M, (int)d, lpr)
    if (subtract_mean) {
      if (it <= 1) BWD(true, 1);
      else if (it <= 2) BWD(true, 2);
      else BWD(true, 4);
    } else {
      if (it <= 1) BWD(false, 1);
      else if (it <= 2) BWD(false, 2);
      else BWD(false, 4);
    }
#undef BWD
  }
  ADNM_CHECK_LAUNCH("mixnorm_bwd");
  // the mix's fold FIRST: an accumulate mask announced with adnm_foldq_accumulate_next (bit 0 = gamma, 1 + k = scalar k, as for
  // adnm_lincomb_bwd) applies to the next fold queued
  adnm_launch_fold("mixnorm_bwd_fold", part_mix, nblk, (int)d + 3, {dgamma, (int)d}, {ds0, 1}, {ds1, 1}, {nullptr, 1}, st);
  adnm_launch_fold("mixnorm_bwd_fold", part_norm, nblk, (int)(2 * d + 2), {dw, (int)d}, {db, (int)d}, {dscale, 1}, {dshift, 1}, st);
  ADNM_CHECK_LAUNCH("mixnorm_bwd_fold");
  return ADNM_OK;
}
