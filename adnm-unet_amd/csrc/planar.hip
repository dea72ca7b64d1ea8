// Per-plane ops of the U-Net conv stack on the channels-last token layout:
//   * Haar (db1) analysis / synthesis butterflies of WTConv2d (WTConv2d.py:31-51, 111-141)        [K3]
//   * InstanceNorm2d(affine=False) with the external scalar scale/shift and optional GELU fused
//     (model_untils.py:90,113 around nn.InstanceNorm2d at :284,371,741,814)                        [K8]
// All are HBM-bound element streams: lane = 4 adjacent channels (16 B), consecutive lanes = consecutive
// channel quads of a pixel, so every access is a coalesced row segment.
#include "adnm_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---------------------------------------------------------------------------------------------- Haar
// x:(B,H,W,C) pixel stride ldx, channel stride CX -> y:(B,h2,w2,4C) contiguous, channel = c*4+k.
template <typename T, int CX>
__global__ __launch_bounds__(kBlock) void haar_dwt_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int B, int H,
                                                          int W, int C) {
  const int C4 = C >> 2, h2 = (H + 1) >> 1, w2 = (W + 1) >> 1;
  const int64_t total = (int64_t)B * h2 * w2 * C4;
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % C4);
  int64_t t = idx / C4;
  const int j = (int)(t % w2);
  t /= w2;
  const int i = (int)(t % h2);
  const int b = (int)(t / h2);
  float v[4][4];  // [pixel a,b,c,d][channel]
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int hh = 2 * i + (q >> 1), ww = 2 * j + (q & 1);
    const bool in = hh < H && ww < W;  // odd sizes are zero-padded bottom/right (WTConv2d.py:114-116)
    const T* p = x + (((int64_t)b * H + hh) * W + ww) * ldx + (int64_t)cg * 4 * CX;
    if (CX == 1) {
      const float4 f = in ? Io<T>::ld4(p) : f4zero();
      v[q][0] = f.x; v[q][1] = f.y; v[q][2] = f.z; v[q][3] = f.w;
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) v[q][m] = in ? Io<T>::ld4(p + m * CX).x : 0.f;  // CX == 4: LL of the previous level
    }
  }
  T* o = y + (((int64_t)b * h2 + i) * w2 + j) * (4 * (int64_t)C) + (int64_t)cg * 16;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float a = v[0][m], bb = v[1][m], c = v[2][m], d = v[3][m];
    Io<T>::st4(o + m * 4, make_float4(0.5f * (a + bb + c + d), 0.5f * (a + bb - c - d), 0.5f * (a - bb + c - d), 0.5f * (a - bb - c + d)));
  }
}

// s:(B,h,w,4C) (+ ll_add:(B,h,w,C) on the LL band) -> y:(B,H,W,C), cropped to H,W.  ya / yb (optional, (B,H,W,C)): added to the result —
// the last synthesis step of WTConv2d's backward sums its three input-gradient paths (wavelet pyramid, base conv, the tensor's other
// consumer) in this one pass instead of two separate adds.
// UP > 0: the whole synthesis CASCADE (WTConv2d.py:128-141) in one launch.  Haar synthesis is local — output pixel (y, x) of a level
// depends on ONE sub-band pixel of that level and on one value of the level above — so instead of materialising the intermediate
// reconstructions (one 5 us launch each), a thread derives its LL addend from the coarser levels' sub-band tensors up1 (and up2 above
// it): one 64-byte read per coarser level, served by L2 (a coarser level is 1/4 the size and shared by 4 / 16 threads).  ll_add then
// belongs to the COARSEST level of the call.  Same operations in the same order as the chain of single-level launches: bitwise equal.
__device__ __forceinline__ float idwt_pick(const float4 f, const float ll, const int q) {   // output (q >> 1, q & 1) of the 2 x 2 block
  const float sy = (q & 2) ? -1.f : 1.f, sz = (q & 1) ? -1.f : 1.f, sw = (q == 1 || q == 2) ? -1.f : 1.f;
  return 0.5f * (ll + sy * f.y + sz * f.z + sw * f.w);
}

template <typename T, int UP>
__global__ __launch_bounds__(kBlock) void haar_idwt_kernel(const T* __restrict__ s, const T* __restrict__ up1, const T* __restrict__ up2,
                                                           const T* __restrict__ ll_add, const T* __restrict__ ya, const T* __restrict__ yb,
                                                           T* __restrict__ y, int B, int H, int W, int C) {
  const int C4 = C >> 2, h2 = (H + 1) >> 1, w2 = (W + 1) >> 1;
  const int64_t total = (int64_t)B * h2 * w2 * C4;
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % C4);
  int64_t t = idx / C4;
  const int j = (int)(t % w2);
  t /= w2;
  const int i = (int)(t % h2);
  const int b = (int)(t / h2);
  const int64_t pix = ((int64_t)b * h2 + i) * w2 + j;
  const T* p = s + pix * (4 * (int64_t)C) + (int64_t)cg * 16;
  float addv[4] = {0.f, 0.f, 0.f, 0.f};
  if (UP == 0) {
    if (ll_add) {
      const float4 add = Io<T>::ld4(ll_add + pix * C + cg * 4);
      addv[0] = add.x, addv[1] = add.y, addv[2] = add.z, addv[3] = add.w;
    }
  } else {
    const int h3 = (h2 + 1) >> 1, w3 = (w2 + 1) >> 1;
    const int64_t pix1 = ((int64_t)b * h3 + (i >> 1)) * w3 + (j >> 1);
    float add1[4] = {0.f, 0.f, 0.f, 0.f};
    if (UP == 2) {
      const int h4 = (h3 + 1) >> 1, w4 = (w3 + 1) >> 1;
      const int64_t pix2 = ((int64_t)b * h4 + (i >> 2)) * w4 + (j >> 2);
      float add2[4] = {0.f, 0.f, 0.f, 0.f};
      if (ll_add) {
        const float4 add = Io<T>::ld4(ll_add + pix2 * C + cg * 4);
        add2[0] = add.x, add2[1] = add.y, add2[2] = add.z, add2[3] = add.w;
      }
      const T* p2 = up2 + pix2 * (4 * (int64_t)C) + (int64_t)cg * 16;
      const int q2 = ((i >> 1) & 1) * 2 + ((j >> 1) & 1);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float4 f = Io<T>::ld4(p2 + m * 4);
        add1[m] = Io<T>::rt(idwt_pick(f, f.x + add2[m], q2));
      }
    } else if (ll_add) {
      const float4 add = Io<T>::ld4(ll_add + pix1 * C + cg * 4);
      add1[0] = add.x, add1[1] = add.y, add1[2] = add.z, add1[3] = add.w;
    }
    const T* p1 = up1 + pix1 * (4 * (int64_t)C) + (int64_t)cg * 16;
    const int q1 = (i & 1) * 2 + (j & 1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float4 f = Io<T>::ld4(p1 + m * 4);
      addv[m] = Io<T>::rt(idwt_pick(f, f.x + add1[m], q1));
    }
  }
  float o[4][4];  // [pixel][channel]
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float4 f = Io<T>::ld4(p + m * 4);
    const float ll = f.x + addv[m];
    o[0][m] = 0.5f * (ll + f.y + f.z + f.w);
    o[1][m] = 0.5f * (ll + f.y - f.z - f.w);
    o[2][m] = 0.5f * (ll - f.y + f.z - f.w);
    o[3][m] = 0.5f * (ll - f.y - f.z + f.w);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int hh = 2 * i + (q >> 1), ww = 2 * j + (q & 1);
    if (hh < H && ww < W) {
      const int64_t at = (((int64_t)b * H + hh) * W + ww) * C + cg * 4;
      float4 v = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
      if (ya) { const float4 a = Io<T>::ld4(ya + at); v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w; }
      if (yb) { const float4 a = Io<T>::ld4(yb + at); v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w; }
      Io<T>::st4(y + at, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------- max pooling
// nn.MaxPool2d on tokens: DownSample's 2x2/stride-2 (model_untils.py:472-487) and EncoderToDecoder's stride-1
// (1x3, 3x1, 3x3, -inf padding, model_untils.py:690-719).  Backward is a GATHER: every input pixel re-derives the
// first arg-max (row-major scan, strict >, as ATen) of each window that contains it, so there are no atomics and no
// saved index tensor (ATen's max_pool_backward_nhwc takes 65 us per call here).
struct PoolGeo {
  int kh, kw, s, ph, pw, Ho, Wo;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, PoolGeo g) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)B * g.Ho * g.Wo * C4;
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % C4);
  int64_t t = idx / C4;
  const int j = (int)(t % g.Wo);
  t /= g.Wo;
  const int i = (int)(t % g.Ho);
  const int b = (int)(t / g.Ho);
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int a = 0; a < g.kh; ++a) {
    const int hh = i * g.s - g.ph + a;
    if (hh < 0 || hh >= H) continue;
    for (int c = 0; c < g.kw; ++c) {
      const int ww = j * g.s - g.pw + c;
      if (ww < 0 || ww >= W) continue;
      const float4 v = Io<T>::ld4(x + (((int64_t)b * H + hh) * W + ww) * C + cg * 4);
      m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y; m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
    }
  }
  Io<T>::st4(y + (((int64_t)b * g.Ho + i) * g.Wo + j) * C + cg * 4, m);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void maxpool_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ dxa,
                                                             T* __restrict__ dx, int B, int H, int W, int C, PoolGeo g) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)B * H * W * C4;
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % C4);
  int64_t t = idx / C4;
  const int w = (int)(t % W);
  t /= W;
  const int h = (int)(t % H);
  const int b = (int)(t / H);
  const T* xb = x + (int64_t)b * H * W * C + cg * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // windows (qi,qj) with qi*s - ph <= h <= qi*s - ph + kh - 1
  int qi0 = h + g.ph - g.kh + 1;
  qi0 = qi0 <= 0 ? 0 : (qi0 + g.s - 1) / g.s;
  int qj0 = w + g.pw - g.kw + 1;
  qj0 = qj0 <= 0 ? 0 : (qj0 + g.s - 1) / g.s;
  const int qi1 = (h + g.ph) / g.s, qj1 = (w + g.pw) / g.s;
  for (int qi = qi0; qi <= qi1 && qi < g.Ho; ++qi)
    for (int qj = qj0; qj <= qj1 && qj < g.Wo; ++qj) {
      float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int arg[4] = {-1, -1, -1, -1};
      for (int a = 0; a < g.kh; ++a) {
        const int hh = qi * g.s - g.ph + a;
        if (hh < 0 || hh >= H) continue;
        for (int c = 0; c < g.kw; ++c) {
          const int ww = qj * g.s - g.pw + c;
          if (ww < 0 || ww >= W) continue;
          const float4 v = Io<T>::ld4(xb + ((int64_t)hh * W + ww) * C);
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (vv[k] > best[k] || arg[k] < 0) { best[k] = vv[k]; arg[k] = hh * W + ww; }
        }
      }
      const float4 gq = Io<T>::ld4(dy + (((int64_t)b * g.Ho + qi) * g.Wo + qj) * C + cg * 4);
      const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (arg[k] == h * W + w) acc[k] += gv[k];
    }
  const int64_t at = (((int64_t)b * H + h) * W + w) * C + cg * 4;
  if (dxa) {   // the gradient of x's other consumer (the encoder stage's output is also a skip tensor): summed here, not by an autograd add
    const float4 e = Io<T>::ld4(dxa + at);
    acc[0] += e.x, acc[1] += e.y, acc[2] += e.z, acc[3] += e.w;
  }
  Io<T>::st4(dx + at, make_float4(acc[0], acc[1], acc[2], acc[3]));
}

// ---------------------------------------------------------------------------------------------- InstanceNorm
struct IGeo {
  int cgb, gx, nchunk, pix_per_chunk;
};
IGeo igeo(int64_t HW, int64_t C) {
  IGeo g;
  const int64_t C4 = C / 4;
  g.cgb = 1;
  while (g.cgb < 64 && g.cgb < C4) g.cgb <<= 1;
  g.gx = (int)adnm_cdiv(C4, g.cgb);
  const int slots = kBlock / g.cgb;
  int64_t ppc = (int64_t)slots * 8;   // more, smaller chunks were measured SLOWER: every apply workgroup re-merges all chunk partials
  int64_t nch = adnm_cdiv(HW, ppc);
  if (nch > 128) {
    nch = 128;
    ppc = adnm_align(adnm_cdiv(HW, nch), slots);
    nch = adnm_cdiv(HW, ppc);
  }
  g.nchunk = (int)nch;
  g.pix_per_chunk = (int)ppc;
  return g;
}

// block-level fold of NV float4 accumulators over the pixel slots: wave shuffles, then LDS across waves.
// Result valid in wave 0, lanes < cgb.
template <int NV>
__device__ __forceinline__ void fold_slots(float4 (&v)[NV], int cgb, float* smem) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    v[k].x = wave_sum_from(v[k].x, cgb); v[k].y = wave_sum_from(v[k].y, cgb);
    v[k].z = wave_sum_from(v[k].z, cgb); v[k].w = wave_sum_from(v[k].w, cgb);
  }
  if (wave > 0 && lane < cgb) {
#pragma unroll
    for (int k = 0; k < NV; ++k) *reinterpret_cast<float4*>(smem + (((wave - 1) * NV + k) * 64 + lane) * 4) = v[k];
  }
  __syncthreads();
  if (wave == 0 && lane < cgb) {
    for (int wv = 0; wv < kBlock / 64 - 1; ++wv)
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(smem + ((wv * NV + k) * 64 + lane) * 4);
        v[k].x += t.x; v[k].y += t.y; v[k].z += t.z; v[k].w += t.w;
      }
  }
}

// Sum of the nchunk chunk partials [S1 | S2] of (b, channel quad), computed ONCE per workgroup: the pixel slots split the
// chunks, fold_slots adds them up, wave 0 publishes through LDS.  (Every thread walking all chunks itself made the apply
// kernels prologue-bound: 13 us for a 16 MB map.)  All threads of the block must call it.
__device__ __forceinline__ void merge_partials(const float* __restrict__ part, int b, int nchunk, int C, int c, bool cv, int cgb, float4& s1,
                                               float4& s2) {
  __shared__ __attribute__((aligned(16))) float msm[3 * 2 * 64 * 4];
  __shared__ __attribute__((aligned(16))) float mout[2][64][4];
  const int cgl = threadIdx.x & (cgb - 1), slot = threadIdx.x / cgb, slots = kBlock / cgb;
  float4 v[2] = {f4zero(), f4zero()};
  if (cv) {
    for (int k = slot; k < nchunk; k += slots) {
      const float* src = part + ((int64_t)b * nchunk + k) * 2 * C;
      const float4 a = *reinterpret_cast<const float4*>(src + c), q = *reinterpret_cast<const float4*>(src + C + c);
      v[0].x += a.x; v[0].y += a.y; v[0].z += a.z; v[0].w += a.w;
      v[1].x += q.x; v[1].y += q.y; v[1].z += q.z; v[1].w += q.w;
    }
  }
  fold_slots<2>(v, cgb, msm);
  if ((threadIdx.x >> 6) == 0 && (threadIdx.x & 63) < cgb) {
    *reinterpret_cast<float4*>(&mout[0][cgl][0]) = v[0];
    *reinterpret_cast<float4*>(&mout[1][cgl][0]) = v[1];
  }
  __syncthreads();
  s1 = *reinterpret_cast<const float4*>(&mout[0][cgl][0]);
  s2 = *reinterpret_cast<const float4*>(&mout[1][cgl][0]);
}

// pass 1 of forward: shifted sums.  part[(b,chunk), {S1,S2}, c] with shift K = x[b,0,c]
template <typename T>
__global__ __launch_bounds__(kBlock) void instnorm_stats_kernel(const T* __restrict__ x, float* __restrict__ part, int64_t HW, int C,
                                                                int cgb, int ppc) {
  __shared__ __attribute__((aligned(16))) float smem[3 * 2 * 64 * 4];
  const int C4 = C >> 2;
  const int cgl = threadIdx.x & (cgb - 1), slot = threadIdx.x / cgb, slots = kBlock / cgb;
  const int cg = blockIdx.x * cgb + cgl;
  const bool cv = cg < C4;
  const int c = cv ? cg * 4 : 0;
  const int b = blockIdx.z;
  const T* xb = x + (int64_t)b * HW * C + c;
  const float4 K = Io<T>::ld4(xb);
  float4 acc[2] = {f4zero(), f4zero()};
  const int64_t p0 = (int64_t)blockIdx.y * ppc;
  const int64_t p1 = p0 + ppc < HW ? p0 + ppc : HW;
  if (cv)
    for (int64_t p = p0 + slot; p < p1; p += slots) {
      const float4 v = Io<T>::ld4(xb + p * C);
      const float d0 = v.x - K.x, d1 = v.y - K.y, d2 = v.z - K.z, d3 = v.w - K.w;
      acc[0].x += d0; acc[0].y += d1; acc[0].z += d2; acc[0].w += d3;
      acc[1].x = fmaf(d0, d0, acc[1].x); acc[1].y = fmaf(d1, d1, acc[1].y);
      acc[1].z = fmaf(d2, d2, acc[1].z); acc[1].w = fmaf(d3, d3, acc[1].w);
    }
  fold_slots<2>(acc, cgb, smem);
  if ((threadIdx.x >> 6) == 0 && (threadIdx.x & 63) < cgb && cv) {
    float* dst = part + ((int64_t)b * gridDim.y + blockIdx.y) * 2 * C;
    *reinterpret_cast<float4*>(dst + c) = acc[0];
    *reinterpret_cast<float4*>(dst + C + c) = acc[1];
  }
}

__device__ __forceinline__ float4 act4(float4 v, int act) {
  if (act == ADNM_ACT_GELU) return make_float4(geluf_(v.x), geluf_(v.y), geluf_(v.z), geluf_(v.w));
  if (act == ADNM_ACT_SILU) return make_float4(siluf_(v.x), siluf_(v.y), siluf_(v.z), siluf_(v.w));
  return v;
}
__device__ __forceinline__ float4 actg4(float4 v, int act) {
  if (act == ADNM_ACT_GELU) return make_float4(gelu_gradf_(v.x), gelu_gradf_(v.y), gelu_gradf_(v.z), gelu_gradf_(v.w));
  if (act == ADNM_ACT_SILU) return make_float4(silu_gradf_(v.x), silu_gradf_(v.y), silu_gradf_(v.z), silu_gradf_(v.w));
  return make_float4(1.f, 1.f, 1.f, 1.f);
}

// pass 2 of forward: merge chunk partials -> mu,rstd (chunk 0 of each (b, channel block) saves them), normalise.
template <typename T>
__global__ __launch_bounds__(kBlock) void instnorm_apply_kernel(const T* __restrict__ x, const float* __restrict__ part,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                T* __restrict__ y, float* __restrict__ mu_out,
                                                                float* __restrict__ rstd_out, int64_t HW, int C, int cgb, int ppc,
                                                                int nchunk, float eps, int act) {
  const int C4 = C >> 2;
  const int cgl = threadIdx.x & (cgb - 1), slot = threadIdx.x / cgb, slots = kBlock / cgb;
  const int cg = blockIdx.x * cgb + cgl;
  const bool cv = cg < C4;
  const int c = cv ? cg * 4 : 0;
  const int b = blockIdx.z;
  float4 s1, s2;
  merge_partials(part, b, nchunk, C, c, cv, cgb, s1, s2);
  if (!cv) return;
  const T* xb = x + (int64_t)b * HW * C + c;
  T* yb = y + (int64_t)b * HW * C + c;
  const float4 K = Io<T>::ld4(xb);
  const float inv = 1.0f / (float)HW;
  const float4 m = make_float4(s1.x * inv, s1.y * inv, s1.z * inv, s1.w * inv);  // mean of (x-K)
  const float4 mu = make_float4(K.x + m.x, K.y + m.y, K.z + m.z, K.w + m.w);
  float4 rstd;
  rstd.x = rsqrtf(fmaxf(s2.x * inv - m.x * m.x, 0.f) + eps);
  rstd.y = rsqrtf(fmaxf(s2.y * inv - m.y * m.y, 0.f) + eps);
  rstd.z = rsqrtf(fmaxf(s2.z * inv - m.z * m.z, 0.f) + eps);
  rstd.w = rsqrtf(fmaxf(s2.w * inv - m.w * m.w, 0.f) + eps);
  if (blockIdx.y == 0 && slot == 0) {
    *reinterpret_cast<float4*>(mu_out + (int64_t)b * C + c) = mu;
    *reinterpret_cast<float4*>(rstd_out + (int64_t)b * C + c) = rstd;
  }
  const float sc = scale ? *scale : 1.f, sh = shift ? *shift : 0.f;
  const int64_t p0 = (int64_t)blockIdx.y * ppc;
  const int64_t p1 = p0 + ppc < HW ? p0 + ppc : HW;
  for (int64_t p = p0 + slot; p < p1; p += slots) {
    const float4 v = Io<T>::ld4(xb + p * C);
    float4 o = make_float4(sc * (v.x - mu.x) * rstd.x + sh, sc * (v.y - mu.y) * rstd.y + sh, sc * (v.z - mu.z) * rstd.z + sh,
                           sc * (v.w - mu.w) * rstd.w + sh);
    Io<T>::st4(yb + p * C, act4(o, act));
  }
}

// backward pass 1: per (b,chunk,c) sums of dpre and dpre*xhat, dpre = dy * act'(scale*xhat+shift)
template <typename T>
__global__ __launch_bounds__(kBlock) void instnorm_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    const float* __restrict__ mu_in, const float* __restrict__ rstd_in,
                                                                    float* __restrict__ part, float* __restrict__ spart, int64_t HW, int C,
                                                                    int cgb, int ppc, int act) {
  __shared__ __attribute__((aligned(16))) float smem[3 * 2 * 64 * 4];
  const int C4 = C >> 2;
  const int cgl = threadIdx.x & (cgb - 1), slot = threadIdx.x / cgb, slots = kBlock / cgb;
  const int cg = blockIdx.x * cgb + cgl;
  const bool cv = cg < C4;
  const int c = cv ? cg * 4 : 0;
  const int b = blockIdx.z;
  const T* xb = x + (int64_t)b * HW * C + c;
  const T* db = dy + (int64_t)b * HW * C + c;
  const float4 mu = *reinterpret_cast<const float4*>(mu_in + (int64_t)b * C + c);
  const float4 rs = *reinterpret_cast<const float4*>(rstd_in + (int64_t)b * C + c);
  const float sc = scale ? *scale : 1.f, sh = shift ? *shift : 0.f;
  float4 acc[2] = {f4zero(), f4zero()};
  const int64_t p0 = (int64_t)blockIdx.y * ppc;
  const int64_t p1 = p0 + ppc < HW ? p0 + ppc : HW;
  if (cv)
    for (int64_t p = p0 + slot; p < p1; p += slots) {
      const float4 v = Io<T>::ld4(xb + p * C), g = Io<T>::ld4(db + p * C);
      const float4 xh = make_float4((v.x - mu.x) * rs.x, (v.y - mu.y) * rs.y, (v.z - mu.z) * rs.z, (v.w - mu.w) * rs.w);
      const float4 ag = actg4(make_float4(sc * xh.x + sh, sc * xh.y + sh, sc * xh.z + sh, sc * xh.w + sh), act);
      const float4 dp = make_float4(g.x * ag.x, g.y * ag.y, g.z * ag.z, g.w * ag.w);
      acc[0].x += dp.x; acc[0].y += dp.y; acc[0].z += dp.z; acc[0].w += dp.w;
      acc[1].x = fmaf(dp.x, xh.x, acc[1].x); acc[1].y = fmaf(dp.y, xh.y, acc[1].y);
      acc[1].z = fmaf(dp.z, xh.z, acc[1].z); acc[1].w = fmaf(dp.w, xh.w, acc[1].w);
    }
  fold_slots<2>(acc, cgb, smem);
  if ((threadIdx.x >> 6) == 0) {
    const bool mine = (threadIdx.x & 63) < cgb && cv;
    if (mine) {
      float* dst = part + ((int64_t)b * gridDim.y + blockIdx.y) * 2 * C;
      *reinterpret_cast<float4*>(dst + c) = acc[0];
      *reinterpret_cast<float4*>(dst + C + c) = acc[1];
    }
    // this workgroup's share of the two scalar gradients (d shift = sum dpre, d scale = sum dpre * xhat over everything): one partial row
    // per workgroup for the shared — deferrable — fold
    if (spart) {
      float a = mine ? (acc[0].x + acc[0].y) + (acc[0].z + acc[0].w) : 0.f, q = mine ? (acc[1].x + acc[1].y) + (acc[1].z + acc[1].w) : 0.f;
      a = wave_sum(a), q = wave_sum(q);
      if (threadIdx.x == 0) {
        float* sp = spart + (((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2;
        sp[0] = a, sp[1] = q;
      }
    }
  }
}

// backward pass 2: dx = rstd*scale*(dpre - mean(dpre) - xhat*mean(dpre*xhat))
template <typename T>
__global__ __launch_bounds__(kBlock) void instnorm_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                    const float* __restrict__ part, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, const float* __restrict__ mu_in,
                                                                    const float* __restrict__ rstd_in, T* __restrict__ dx, int64_t HW,
                                                                    int C, int cgb, int ppc, int nchunk, int act) {
  const int C4 = C >> 2;
  const int cgl = threadIdx.x & (cgb - 1), slot = threadIdx.x / cgb, slots = kBlock / cgb;
  const int cg = blockIdx.x * cgb + cgl;
  const bool cv = cg < C4;
  const int c = cv ? cg * 4 : 0;
  const int b = blockIdx.z;
  float4 s1, s2;
  merge_partials(part, b, nchunk, C, c, cv, cgb, s1, s2);
  if (!cv) return;
  const T* xb = x + (int64_t)b * HW * C + c;
  const T* db = dy + (int64_t)b * HW * C + c;
  T* ob = dx + (int64_t)b * HW * C + c;
  const float4 mu = *reinterpret_cast<const float4*>(mu_in + (int64_t)b * C + c);
  const float4 rs = *reinterpret_cast<const float4*>(rstd_in + (int64_t)b * C + c);
  const float sc = scale ? *scale : 1.f, sh = shift ? *shift : 0.f;
  const float inv = 1.0f / (float)HW;
  s1.x *= inv; s1.y *= inv; s1.z *= inv; s1.w *= inv;
  s2.x *= inv; s2.y *= inv; s2.z *= inv; s2.w *= inv;
  const int64_t p0 = (int64_t)blockIdx.y * ppc;
  const int64_t p1 = p0 + ppc < HW ? p0 + ppc : HW;
  for (int64_t p = p0 + slot; p < p1; p += slots) {
    const float4 v = Io<T>::ld4(xb + p * C), g = Io<T>::ld4(db + p * C);
    const float4 xh = make_float4((v.x - mu.x) * rs.x, (v.y - mu.y) * rs.y, (v.z - mu.z) * rs.z, (v.w - mu.w) * rs.w);
    const float4 ag = actg4(make_float4(sc * xh.x + sh, sc * xh.y + sh, sc * xh.z + sh, sc * xh.w + sh), act);
    float4 o;
    o.x = rs.x * sc * (g.x * ag.x - s1.x - xh.x * s2.x);
    o.y = rs.y * sc * (g.y * ag.y - s1.y - xh.y * s2.y);
    o.z = rs.z * sc * (g.z * ag.z - s1.z - xh.z * s2.z);
    o.w = rs.w * sc * (g.w * ag.w - s1.w - xh.w * s2.w);
    Io<T>::st4(ob + p * C, o);
  }
}


}  // namespace

extern "C" int adnm_haar_dwt(const void* x, int64_t ldx, int64_t cx, void* y, int64_t B, int64_t H, int64_t W, int64_t C, int dtype,
                             adnm_stream_t stream) {
  ADNM_REQUIRE(x && y, "haar_dwt: null pointer");
  ADNM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "haar_dwt: bad shape (C=%lld must be a multiple of 4)", (long long)C);
  ADNM_REQUIRE(cx == 1 || cx == 4, "haar_dwt: channel stride %lld not in {1,4}", (long long)cx);
  ADNM_REQUIRE(ldx >= C * cx && ldx % 4 == 0, "haar_dwt: pixel stride too small");
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "haar_dwt: bad dtype %d", dtype);
  const int64_t total = B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  const unsigned grid = (unsigned)adnm_cdiv(total, kBlock);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ADNM_F32) {
    if (cx == 1) { ADNM_PROF("haar_dwt", st, 4.0 * B * H * W * C * 2); haar_dwt_kernel<float, 1><<<grid, kBlock, 0, st>>>((const float*)x, ldx, (float*)y, (int)B, (int)H, (int)W, (int)C); }
    else { ADNM_PROF("haar_dwt", st, 4.0 * B * H * W * C * 2); haar_dwt_kernel<float, 4><<<grid, kBlock, 0, st>>>((const float*)x, ldx, (float*)y, (int)B, (int)H, (int)W, (int)C); }
  } else {
    if (cx == 1) { ADNM_PROF("haar_dwt", st, 2.0 * B * H * W * C * 2); haar_dwt_kernel<uint16_t, 1><<<grid, kBlock, 0, st>>>((const uint16_t*)x, ldx, (uint16_t*)y, (int)B, (int)H, (int)W, (int)C); }
    else { ADNM_PROF("haar_dwt", st, 2.0 * B * H * W * C * 2); haar_dwt_kernel<uint16_t, 4><<<grid, kBlock, 0, st>>>((const uint16_t*)x, ldx, (uint16_t*)y, (int)B, (int)H, (int)W, (int)C); }
  }
  ADNM_CHECK_LAUNCH("haar_dwt");
  return ADNM_OK;
}

extern "C" int adnm_haar_idwt(const void* s, const void* up1, const void* up2, const void* ll_add, const void* y_add1, const void* y_add2, void* y,
                              int64_t B, int64_t H, int64_t W, int64_t C, int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(s && y, "haar_idwt: null pointer");
  ADNM_REQUIRE(up1 || !up2, "haar_idwt: the coarser levels must be filled in order (up2 without up1)");
  ADNM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "haar_idwt: bad shape (C=%lld must be a multiple of 4)", (long long)C);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "haar_idwt: bad dtype %d", dtype);
  const int64_t total = B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  const unsigned grid = (unsigned)adnm_cdiv(total, kBlock);
  hipStream_t st = (hipStream_t)stream;
  const double units = 2.0 + (up1 ? 0.25 : 0.0) + (up2 ? 0.0625 : 0.0) + (ll_add ? (up2 ? 0.015625 : (up1 ? 0.0625 : 0.25)) : 0.0) + (y_add1 ? 1.0 : 0.0) + (y_add2 ? 1.0 : 0.0);
  const int up = up2 ? 2 : (up1 ? 1 : 0);
#define IDWT(T, UP) haar_idwt_kernel<T, UP><<<grid, kBlock, 0, st>>>((const T*)s, (const T*)up1, (const T*)up2, (const T*)ll_add, (const T*)y_add1, (const T*)y_add2, (T*)y, (int)B, (int)H, (int)W, (int)C)
  if (dtype == ADNM_F32) {
    ADNM_PROF("haar_idwt", st, 4.0 * B * H * W * C * units);
    if (up == 0) IDWT(float, 0);
    else if (up == 1) IDWT(float, 1);
    else IDWT(float, 2);
  } else {
    ADNM_PROF("haar_idwt", st, 2.0 * B * H * W * C * units);
    if (up == 0) IDWT(uint16_t, 0);
    else if (up == 1) IDWT(uint16_t, 1);
    else IDWT(uint16_t, 2);
  }
#undef IDWT
  ADNM_CHECK_LAUNCH("haar_idwt");
  return ADNM_OK;
}

static int pool_geo(const char* who, int64_t B, int64_t H, int64_t W, int64_t C, int kh, int kw, int stride, int dtype, PoolGeo* g) {
  ADNM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad shape (C=%lld must be a multiple of 4)", who, (long long)C);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "%s: bad dtype %d", who, dtype);
  if (stride == 1) {
    ADNM_REQUIRE((kh == 1 || kh == 3) && (kw == 1 || kw == 3), "%s: stride-1 pooling supports 1x3, 3x1, 3x3 ('same' padding)", who);
    *g = PoolGeo{kh, kw, 1, kh / 2, kw / 2, (int)H, (int)W};
  } else {
    ADNM_REQUIRE(stride == kh && stride == kw && stride >= 2 && stride <= 4, "%s: strided pooling needs kernel == stride in [2,4]", who);
    *g = PoolGeo{kh, kw, stride, 0, 0, (int)(H / stride), (int)(W / stride)};
    ADNM_REQUIRE(g->Ho > 0 && g->Wo > 0, "%s: input smaller than the window", who);
  }
  return ADNM_OK;
}

extern "C" int adnm_maxpool_fwd(const void* x, void* y, int64_t B, int64_t H, int64_t W, int64_t C, int kh, int kw, int stride, int dtype,
                                adnm_stream_t stream) {
  ADNM_REQUIRE(x && y, "maxpool_fwd: null pointer");
  PoolGeo g;
  if (int rc = pool_geo("maxpool_fwd", B, H, W, C, kh, kw, stride, dtype, &g)) return rc;
  const int64_t total = B * g.Ho * g.Wo * (C / 4);
  const unsigned grid = (unsigned)adnm_cdiv(total, kBlock);
  hipStream_t st = (hipStream_t)stream;
  const double es = dtype == ADNM_F32 ? 4.0 : 2.0;
  ADNM_PROF("maxpool_fwd", st, es * C * B * ((double)H * W + (double)g.Ho * g.Wo));
  if (dtype == ADNM_F32) maxpool_fwd_kernel<float><<<grid, kBlock, 0, st>>>((const float*)x, (float*)y, (int)B, (int)H, (int)W, (int)C, g);
  else maxpool_fwd_kernel<uint16_t><<<grid, kBlock, 0, st>>>((const uint16_t*)x, (uint16_t*)y, (int)B, (int)H, (int)W, (int)C, g);
  ADNM_CHECK_LAUNCH("maxpool_fwd");
  return ADNM_OK;
}

extern "C" int adnm_maxpool_bwd(const void* dy, const void* x, const void* dx_add, void* dx, int64_t B, int64_t H, int64_t W, int64_t C, int kh, int kw, int stride,
                                int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(dy && x && dx, "maxpool_bwd: null pointer");
  PoolGeo g;
  if (int rc = pool_geo("maxpool_bwd", B, H, W, C, kh, kw, stride, dtype, &g)) return rc;
  const int64_t total = B * H * W * (C / 4);
  const unsigned grid = (unsigned)adnm_cdiv(total, kBlock);
  hipStream_t st = (hipStream_t)stream;
  const double es = dtype == ADNM_F32 ? 4.0 : 2.0;
  ADNM_PROF("maxpool_bwd", st, es * C * B * ((dx_add ? 3.0 : 2.0) * H * W + (double)g.Ho * g.Wo));
  if (dtype == ADNM_F32)
    maxpool_bwd_kernel<float><<<grid, kBlock, 0, st>>>((const float*)dy, (const float*)x, (const float*)dx_add, (float*)dx, (int)B, (int)H, (int)W, (int)C, g);
  else
    maxpool_bwd_kernel<uint16_t><<<grid, kBlock, 0, st>>>((const uint16_t*)dy, (const uint16_t*)x, (const uint16_t*)dx_add, (uint16_t*)dx, (int)B, (int)H, (int)W, (int)C, g);
  ADNM_CHECK_LAUNCH("maxpool_bwd");
  return ADNM_OK;
}

extern "C" int64_t adnm_instnorm_ws_bytes(int64_t B, int64_t HW, int64_t C) {
  if (B <= 0 || HW <= 0 || C < 4) return 0;
  const IGeo g = igeo(HW, C);
  return (B * g.nchunk * 2 * C + B * g.nchunk * (int64_t)g.gx * 2) * (int64_t)sizeof(float);   // channel partials + per-workgroup scalar partials
}

static int instnorm_check(const char* who, int64_t B, int64_t HW, int64_t C, int act, int dtype, void* ws, int64_t ws_bytes) {
  ADNM_REQUIRE(B > 0 && HW > 1 && C > 0 && C % 4 == 0 && B <= 65535, "%s: bad shape B=%lld HW=%lld C=%lld", who, (long long)B, (long long)HW,
               (long long)C);
  ADNM_REQUIRE(act == ADNM_ACT_NONE || act == ADNM_ACT_GELU || act == ADNM_ACT_SILU, "%s: bad activation %d", who, act);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "%s: bad dtype %d", who, dtype);
  if (!ws || ws_bytes < adnm_instnorm_ws_bytes(B, HW, C)) {
    adnm_set_error("%s: workspace %lld < %lld bytes", who, (long long)ws_bytes, (long long)adnm_instnorm_ws_bytes(B, HW, C));
    return ADNM_EWORKSPACE;
  }
  return ADNM_OK;
}

extern "C" int adnm_instnorm_fwd(const void* x, const float* scale, const float* shift, void* y, float* mu, float* rstd, void* ws,
                                 int64_t ws_bytes, int64_t B, int64_t HW, int64_t C, float eps, int act, int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(x && y && mu && rstd, "instnorm_fwd: null pointer");
  if (int rc = instnorm_check("instnorm_fwd", B, HW, C, act, dtype, ws, ws_bytes)) return rc;
  const IGeo g = igeo(HW, C);
  const dim3 grid(g.gx, g.nchunk, (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  if (dtype == ADNM_F32) {
    { ADNM_PROF("instnorm_stats", st, 4.0 * B * HW * C); instnorm_stats_kernel<float><<<grid, kBlock, 0, st>>>((const float*)x, part, HW, (int)C, g.cgb, g.pix_per_chunk); }
    { ADNM_PROF("instnorm_apply", st, 4.0 * B * HW * C * 2); instnorm_apply_kernel<float><<<grid, kBlock, 0, st>>>((const float*)x, part, scale, shift, (float*)y, mu, rstd, HW, (int)C, g.cgb,
                                                          g.pix_per_chunk, g.nchunk, eps, act); }
  } else {
    { ADNM_PROF("instnorm_stats", st, 2.0 * B * HW * C); instnorm_stats_kernel<uint16_t><<<grid, kBlock, 0, st>>>((const uint16_t*)x, part, HW, (int)C, g.cgb, g.pix_per_chunk); }
    { ADNM_PROF("instnorm_apply", st, 2.0 * B * HW * C * 2); instnorm_apply_kernel<uint16_t><<<grid, kBlock, 0, st>>>((const uint16_t*)x, part, scale, shift, (uint16_t*)y, mu, rstd, HW, (int)C,
                                                             g.cgb, g.pix_per_chunk, g.nchunk, eps, act); }
  }
  ADNM_CHECK_LAUNCH("instnorm_fwd");
  return ADNM_OK;
}

extern "C" int adnm_instnorm_bwd(const void* dy, const void* x, const float* scale, const float* shift, const float* mu,
                                 const float* rstd, void* dx, float* dscale, float* dshift, void* ws, int64_t ws_bytes, int64_t B,
                                 int64_t HW, int64_t C, int act, int dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(dy && x && mu && rstd && dx, "instnorm_bwd: null pointer");
  if (int rc = instnorm_check("instnorm_bwd", B, HW, C, act, dtype, ws, ws_bytes)) return rc;
  const IGeo g = igeo(HW, C);
  const dim3 grid(g.gx, g.nchunk, (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  float* spart = (dscale || dshift) ? part + B * g.nchunk * 2 * C : nullptr;
  if (dtype == ADNM_F32) {
    { ADNM_PROF("instnorm_bwd_stats", st, 4.0 * B * HW * C * 2); instnorm_bwd_stats_kernel<float><<<grid, kBlock, 0, st>>>((const float*)dy, (const float*)x, scale, shift, mu, rstd, part, spart, HW, (int)C,
                                                              g.cgb, g.pix_per_chunk, act); }
    { ADNM_PROF("instnorm_bwd_apply", st, 4.0 * B * HW * C * 3); instnorm_bwd_apply_kernel<float><<<grid, kBlock, 0, st>>>((const float*)dy, (const float*)x, part, scale, shift, mu, rstd, (float*)dx,
                                                              HW, (int)C, g.cgb, g.pix_per_chunk, g.nchunk, act); }
  } else {
    { ADNM_PROF("instnorm_bwd_stats", st, 2.0 * B * HW * C * 2); instnorm_bwd_stats_kernel<uint16_t><<<grid, kBlock, 0, st>>>((const uint16_t*)dy, (const uint16_t*)x, scale, shift, mu, rstd, part, spart, HW,
                                                                 (int)C, g.cgb, g.pix_per_chunk, act); }
    { ADNM_PROF("instnorm_bwd_apply", st, 2.0 * B * HW * C * 3); instnorm_bwd_apply_kernel<uint16_t><<<grid, kBlock, 0, st>>>((const uint16_t*)dy, (const uint16_t*)x, part, scale, shift, mu, rstd,
                                                                 (uint16_t*)dx, HW, (int)C, g.cgb, g.pix_per_chunk, g.nchunk, act); }
  }
  ADNM_CHECK_LAUNCH("instnorm_bwd");
  // d shift / d scale: parameter gradients through the shared fold (deferrable: the caller may have bound a fold queue)
  if (spart) adnm_launch_fold("instnorm_bwd_scalar", spart, (int)(B * g.nchunk * g.gx), 2, {dshift, 1}, {dscale, 1}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("instnorm_bwd");
  return ADNM_OK;
}
