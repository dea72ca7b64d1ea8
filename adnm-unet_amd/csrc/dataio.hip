// The data formats on either side of the hot path.
//
//   radar_ingest  — the input pipeline of datasets/Shanghai.py:52-59,121: uint8 radar frames (25, H0, W0) in 0..70 -> float / 255 ->
//                   transforms.Resize((S, S)) (bilinear, align_corners=False, no antialias: what torchvision's tensor Resize hands to
//                   F.interpolate) -> (B, T, 1, S, S) fp32, in ONE pass over bytes that arrive by DMA from pinned host memory, instead of
//                   a float copy 4x the size followed by torch's resize and a pageable, synchronous H2D copy (train.py:134).
//   eval_counts   — SimplifiedEvaluator.evaluate (datasets/Shanghai_metrics.py:49-152) without the round trip to numpy: per frame the
//                   contingency counts TP / FN / FP / TN at every threshold on the uint16-truncated, value_scale'd, [0,1]-clipped fields
//                   (:45-47,103-112) and the sums |d|, d^2 of the scaled float fields (:114-120) from which MAE / MSE / RMSE / PSNR follow.
#include "adnm_common.h"
#include <math.h>

namespace {
constexpr int kBlock = 256;
constexpr int kMaxThr = 8;

__global__ __launch_bounds__(kBlock) void radar_ingest_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t frames, int H0, int W0,
                                                              int S, float scale_y, float scale_x, float mul) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= frames * S * S) return;
  const int x = (int)(i % S), y = (int)((i / S) % S);
  const int64_t f = i / ((int64_t)S * S);
  // torch's area_pixel_compute_source_index(align_corners=False): src = scale * (dst + 0.5) - 0.5, clamped at 0
  float sy = scale_y * (y + 0.5f) - 0.5f, sx = scale_x * (x + 0.5f) - 0.5f;
  sy = sy < 0.f ? 0.f : sy;
  sx = sx < 0.f ? 0.f : sx;
  const int y0 = (int)sy < H0 - 1 ? (int)sy : H0 - 1, x0 = (int)sx < W0 - 1 ? (int)sx : W0 - 1;
  const int y1 = y0 + (y0 < H0 - 1), x1 = x0 + (x0 < W0 - 1);
  const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
  const uint8_t* p = src + f * (int64_t)H0 * W0;
  const float v00 = p[(int64_t)y0 * W0 + x0], v01 = p[(int64_t)y0 * W0 + x1], v10 = p[(int64_t)y1 * W0 + x0], v11 = p[(int64_t)y1 * W0 + x1];
  // the reference divides by 255 BEFORE resizing; interpolation is linear, so the order only moves the rounding: kept as the reference has it
  dst[i] = hy * (hx * (v00 * mul) + lx * (v01 * mul)) + ly * (hx * (v10 * mul) + lx * (v11 * mul));
}

struct Thr {
  float t[kMaxThr];
  int n;
};

// part[blk][frame][4*nthr + 2]; blocks along a frame: gridDim.x, frames: gridDim.y
__global__ __launch_bounds__(kBlock) void eval_counts_kernel(const float* __restrict__ truth, const float* __restrict__ pred, float* __restrict__ part,
                                                             int64_t hw, float value_scale, Thr thr) {
  __shared__ float sm[kBlock / 64][4 * kMaxThr + 2];
  const int64_t f = blockIdx.y;
  const float* t = truth + f * hw;
  const float* p = pred + f * hw;
  float cnt[4 * kMaxThr], sa = 0.f, sq = 0.f;
#pragma unroll
  for (int k = 0; k < 4 * kMaxThr; ++k) cnt[k] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < hw; i += (int64_t)gridDim.x * kBlock) {
    const float tc = fminf(fmaxf(t[i], 0.f), 1.f), pc = fminf(fmaxf(p[i], 0.f), 1.f);
    const float ts = tc * value_scale, ps = pc * value_scale;
    const float d = ps - ts;
    sa += fabsf(d);
    sq = fmaf(d, d, sq);
    const float ti = floorf(ts), pi = floorf(ps);   // .astype(np.uint16) of a non-negative float: truncation
#pragma unroll
    for (int k = 0; k < kMaxThr; ++k)
      if (k < thr.n) {
        const bool o = ti >= thr.t[k], s = pi >= thr.t[k];
        cnt[4 * k + 0] += (o && s) ? 1.f : 0.f;     // TP
        cnt[4 * k + 1] += (o && !s) ? 1.f : 0.f;    // FN
        cnt[4 * k + 2] += (!o && s) ? 1.f : 0.f;    // FP
        cnt[4 * k + 3] += (!o && !s) ? 1.f : 0.f;   // TN
      }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nout = 4 * thr.n + 2;
#pragma unroll
  for (int k = 0; k < 4 * kMaxThr; ++k) {
    const float v = wave_sum(cnt[k]);
    if (lane == 0 && k < 4 * thr.n) sm[wave][k] = v;
  }
  sa = wave_sum(sa);
  sq = wave_sum(sq);
  if (lane == 0) {
    sm[wave][4 * thr.n] = sa;
    sm[wave][4 * thr.n + 1] = sq;
  }
  __syncthreads();
  if ((int)threadIdx.x < nout)
    part[((int64_t)blockIdx.x * gridDim.y + f) * nout + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// SSIM of SimplifiedEvaluator.cal_ssim (datasets/Shanghai_metrics.py:132-152): per frame, the mean over the VALID region of
//   ((2 mu1 mu2 + C1)(2 s12 + C2)) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)),   mu / s = 11x11 Gaussian (sigma 1.5) windowed moments of the
// value_scale'd clipped fields, computed in float64 as the reference does.  A workgroup = a 16 x 16 tile of output pixels of one frame: the
// 26 x 26 input patch of both fields is staged in LDS, the five moment maps are filtered separably (rows, then columns) through LDS, and the
// tile's sum of the SSIM map goes to part[frame][tile] for the shared fold.
constexpr int kSsimT = 16, kSsimR = 5, kSsimP = kSsimT + 2 * kSsimR;
struct GaussWin {
  double k[2 * kSsimR + 1];
};
__global__ __launch_bounds__(kSsimT * kSsimT) void eval_ssim_kernel(const float* __restrict__ truth, const float* __restrict__ pred, float* __restrict__ part,
                                                                  int H, int W, int tiles_x, int tiles, float value_scale, GaussWin g) {
  __shared__ double sp[kSsimP][kSsimP], st[kSsimP][kSsimP];   // pred, truth patches (scaled, clipped)
  __shared__ double sh[5][kSsimP][kSsimT];                     // row-filtered moments: p, t, p^2, t^2, p t
  __shared__ double red[kSsimT * kSsimT / 64];
  const int f = blockIdx.y, tile = blockIdx.x, ty0 = (tile / tiles_x) * kSsimT, tx0 = (tile % tiles_x) * kSsimT;
  const int Ho = H - 2 * kSsimR, Wo = W - 2 * kSsimR;
  const float* tp = truth + (int64_t)f * H * W;
  const float* pp = pred + (int64_t)f * H * W;
  for (int i = threadIdx.x; i < kSsimP * kSsimP; i += kSsimT * kSsimT) {
    const int r = i / kSsimP, c = i % kSsimP, y = ty0 + r, x = tx0 + c;   // input pixel of valid-output (ty0, tx0) + offset
    double a = 0.0, b = 0.0;
    if (y < H && x < W) {
      a = (double)(fminf(fmaxf(pp[(int64_t)y * W + x], 0.f), 1.f) * value_scale);   // float32 product (as numpy), then float64
      b = (double)(fminf(fmaxf(tp[(int64_t)y * W + x], 0.f), 1.f) * value_scale);
    }
    sp[r][c] = a, st[r][c] = b;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSsimP * kSsimT; i += kSsimT * kSsimT) {
    const int r = i / kSsimT, c = i % kSsimT;
    double m[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j <= 2 * kSsimR; ++j) {
      const double a = sp[r][c + j], b = st[r][c + j], w = g.k[j];
      m[0] += w * a, m[1] += w * b, m[2] += w * a * a, m[3] += w * b * b, m[4] += w * a * b;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) sh[q][r][c] = m[q];
  }
  __syncthreads();
  const int r = threadIdx.x / kSsimT, c = threadIdx.x % kSsimT;
  double v = 0.0;
  if (ty0 + r < Ho && tx0 + c < Wo) {
    double m[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j <= 2 * kSsimR; ++j)
#pragma unroll
      for (int q = 0; q < 5; ++q) m[q] += g.k[j] * sh[q][r + j][c];
    const double C1 = (0.01 * value_scale) * (0.01 * value_scale), C2 = (0.03 * value_scale) * (0.03 * value_scale);
    const double mu12 = m[0] * m[1], s1 = m[2] - m[0] * m[0], s2 = m[3] - m[1] * m[1], s12 = m[4] - mu12;
    v = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((m[0] * m[0] + m[1] * m[1] + C1) * (s1 + s2 + C2));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)tile * gridDim.y + f] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

inline int eval_blocks(int64_t hw) {
  int64_t b = adnm_cdiv(hw, kBlock * 16);
  return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}
}  // namespace

extern "C" int adnm_radar_ingest(const void* src_u8, float* dst, int64_t frames, int64_t H0, int64_t W0, int64_t S, float mul, adnm_stream_t stream) {
  ADNM_REQUIRE(src_u8 && dst, "radar_ingest: null pointer");
  ADNM_REQUIRE(frames > 0 && H0 > 0 && W0 > 0 && S > 0 && H0 < 32768 && W0 < 32768 && S < 32768, "radar_ingest: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = frames * S * S;
  ADNM_PROF("radar_ingest", st, (double)frames * H0 * W0 + 4.0 * total);
  radar_ingest_kernel<<<(unsigned)adnm_cdiv(total, kBlock), kBlock, 0, st>>>((const uint8_t*)src_u8, dst, frames, (int)H0, (int)W0, (int)S, (float)H0 / (float)S,
                                                                            (float)W0 / (float)S, mul);
  ADNM_CHECK_LAUNCH("radar_ingest");
  return ADNM_OK;
}

extern "C" int64_t adnm_eval_counts_ws_bytes(int64_t frames, int64_t hw, int64_t nthr) {
  if (frames <= 0 || hw <= 0 || nthr <= 0 || nthr > kMaxThr) return 0;
  return (int64_t)eval_blocks(hw) * frames * (4 * nthr + 2) * (int64_t)sizeof(float);
}

// out: (frames, 4*nthr + 2) fp32 = [TP, FN, FP, TN] per threshold, then sum |d|, sum d^2 of the value_scale'd clipped fields.  OVERWRITES out.
extern "C" int adnm_eval_counts(const float* truth, const float* pred, float* out, const float* thresholds_host, int64_t nthr, float value_scale,
                                void* ws, int64_t ws_bytes, int64_t frames, int64_t hw, adnm_stream_t stream) {
  ADNM_REQUIRE(truth && pred && out && thresholds_host, "eval_counts: null pointer");
  ADNM_REQUIRE(frames > 0 && frames <= 65535 && hw > 0 && hw < (1ll << 24) && nthr > 0 && nthr <= kMaxThr,
               "eval_counts: bad shape (frames <= 65535, pixels per frame < 2^24 so that counts are exact in fp32, <= 8 thresholds)");
  if (!ws || ws_bytes < adnm_eval_counts_ws_bytes(frames, hw, nthr)) {
    adnm_set_error("eval_counts: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_eval_counts_ws_bytes(frames, hw, nthr));
    return ADNM_EWORKSPACE;
  }
  Thr thr;
  thr.n = (int)nthr;
  for (int k = 0; k < kMaxThr; ++k) thr.t[k] = k < nthr ? thresholds_host[k] : 0.f;
  hipStream_t st = (hipStream_t)stream;
  const int nb = eval_blocks(hw), nout = 4 * (int)nthr + 2;
  {
    ADNM_PROF("eval_counts", st, 8.0 * frames * hw);
    eval_counts_kernel<<<dim3(nb, (unsigned)frames), kBlock, 0, st>>>(truth, pred, (float*)ws, hw, value_scale, thr);
  }
  ADNM_CHECK_LAUNCH("eval_counts");
  adnm_launch_fold("eval_counts_fold", (const float*)ws, nb, (int)(frames * nout), {out, (int)(frames * nout)}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("eval_counts_fold");
  return ADNM_OK;
}


extern "C" int64_t adnm_eval_ssim_ws_bytes(int64_t frames, int64_t H, int64_t W) {
  if (frames <= 0 || H <= 2 * kSsimR || W <= 2 * kSsimR) return 0;
  return adnm_cdiv(H - 2 * kSsimR, kSsimT) * adnm_cdiv(W - 2 * kSsimR, kSsimT) * frames * (int64_t)sizeof(float);
}

// out[frame] = SUM of the SSIM map of frame `frame` over its (H - 10) x (W - 10) valid region (SimplifiedEvaluator.cal_ssim,
// datasets/Shanghai_metrics.py:132-152, takes the mean: the caller divides) on the [0,1]-clipped, value_scale'd fields.  OVERWRITES out.
extern "C" int adnm_eval_ssim(const float* truth, const float* pred, float* out, float value_scale, void* ws, int64_t ws_bytes, int64_t frames, int64_t H,
                              int64_t W, adnm_stream_t stream) {
  ADNM_REQUIRE(truth && pred && out, "eval_ssim: null pointer");
  ADNM_REQUIRE(frames > 0 && frames <= 65535 && H > 2 * kSsimR && W > 2 * kSsimR && H < 32768 && W < 32768,
               "eval_ssim: needs frames of more than 10 x 10 pixels (11 x 11 window, valid region), got %lld x %lld", (long long)H, (long long)W);
  if (!ws || ws_bytes < adnm_eval_ssim_ws_bytes(frames, H, W)) {
    adnm_set_error("eval_ssim: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_eval_ssim_ws_bytes(frames, H, W));
    return ADNM_EWORKSPACE;
  }
  GaussWin g;
  double sum = 0.0;
  for (int j = 0; j <= 2 * kSsimR; ++j) sum += (g.k[j] = exp(-(double)((j - kSsimR) * (j - kSsimR)) / (2.0 * 1.5 * 1.5)));
  for (int j = 0; j <= 2 * kSsimR; ++j) g.k[j] /= sum;
  const int tx = (int)adnm_cdiv(W - 2 * kSsimR, kSsimT), ty = (int)adnm_cdiv(H - 2 * kSsimR, kSsimT);
  hipStream_t st = (hipStream_t)stream;
  {
    ADNM_PROF("eval_ssim", st, 8.0 * frames * H * W);
    eval_ssim_kernel<<<dim3((unsigned)(tx * ty), (unsigned)frames), kSsimT * kSsimT, 0, st>>>(truth, pred, (float*)ws, (int)H, (int)W, tx, tx * ty, value_scale, g);
  }
  ADNM_CHECK_LAUNCH("eval_ssim");
  adnm_launch_fold("eval_ssim_fold", (const float*)ws, tx * ty, (int)frames, {out, (int)frames}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("eval_ssim_fold");
  return ADNM_OK;
}
