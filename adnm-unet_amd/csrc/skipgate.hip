// K12 — the pooled gating core of EncoderToDecoder (reference models/model_untils.py:767-787), fused.
//
//   p_k = maxpool_k(x) + avgpool_k(x)                    k = 0: 3x1 window, 1: 1x3, 2: 3x3   (stride 1, same size)
//   c_k = conv_k(p_k) + b_k                              grouped conv, 4 channels per group; kernel 1x3 / 3x1 / 3x3
//   y_k = silu(enh_k * (x * gelu(c_k) * fw_k + fb_k - thr_k))      fw/fb/enh/thr: ffd13+act_func13 for k = 0, 1 (the
//                                                                   reference reuses them), ffd33+act_func33 for k = 2
//   out = gamma * (alpha1 y_0 + alpha2 y_1 + alpha3 y_2)
//
// The reference runs ~30 library launches forward and ~70 backward for this on 4x4 .. 16x16 feature maps with
// 256..1024 channels — pure launch latency on MI355X.  Here: 2 launches forward, 5-6 backward.  Tokens are channels-last
// (B, H, W, C) fp32; a lane owns one conv GROUP (4 channels = one float4), so the grouped conv is a 4x4 matrix per tap
// in registers, read straight from PyTorch's (C, 4, kh, kw) weight layout.  C % 4 == 0: lanes past the last (slice, group) pair
// stay in their wave (the in-wave reductions need every lane) and contribute zeros.
#include "adnm_common.h"

namespace {
constexpr int kBlock = 256;
constexpr int kVec = 8;    // per-channel gradient vectors: gamma, fw13, fb13, fw33, fb33, b0, b1, b2
constexpr int kScal = 8;   // scalar gradients: alpha1..3, enh13, thr13, enh33, thr33, (pad)
constexpr int kWeightsPerChannel = 4 * (3 + 3 + 9);

struct Geo {
  int B, H, W, C, C4;
  int64_t npix;
};

struct Params {  // device pointers, order documented in include/adnm_hip.h (adnm_skipgate_fwd)
  const float* w[3];
  const float* b[3];
  const float* fw[2];
  const float* fb[2];
  const float* enh[2];
  const float* thr[2];
  const float* alpha[3];
  const float* gamma;
};

template <int K>
struct Taps {
  static constexpr int T = K == 2 ? 9 : 3;
  static __device__ __forceinline__ int di(int t) { return K == 0 ? 0 : (K == 1 ? t - 1 : t / 3 - 1); }
  static __device__ __forceinline__ int dj(int t) { return K == 0 ? t - 1 : (K == 1 ? 0 : t % 3 - 1); }
};

__device__ __forceinline__ void decode(int64_t idx, const Geo& g, int& b, int& h, int& w, int& cg) {
  cg = (int)(idx % g.C4);
  int64_t t = idx / g.C4;
  w = (int)(t % g.W);
  t /= g.W;
  h = (int)(t % g.H);
  b = (int)(t / g.H);
}
__device__ __forceinline__ void f4(const float4& v, float o[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ float4 mk4(const float o[4]) { return make_float4(o[0], o[1], o[2], o[3]); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float o[4]) { *reinterpret_cast<float4*>(p) = mk4(o); }

// ------------------------------------------------------------------------------------------------ pools
__global__ __launch_bounds__(kBlock) void skip_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ p0, float* __restrict__ p1,
                                                               float* __restrict__ p2, Geo g) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= g.npix * g.C4) return;
  int b, h, w, cg;
  decode(idx, g, b, h, w, cg);
  float mx[3][4], sm[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) mx[k][c] = -INFINITY, sm[k][c] = 0.f;
#pragma unroll
  for (int a = -1; a <= 1; ++a)
#pragma unroll
    for (int c = -1; c <= 1; ++c) {
      const int hh = h + a, ww = w + c;
      if (hh < 0 || hh >= g.H || ww < 0 || ww >= g.W) continue;
      float v[4];
      f4(ld4(x + (((int64_t)b * g.H + hh) * g.W + ww) * g.C + cg * 4), v);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (c == 0) { mx[0][q] = v[q] > mx[0][q] ? v[q] : mx[0][q]; sm[0][q] += v[q]; }
        if (a == 0) { mx[1][q] = v[q] > mx[1][q] ? v[q] : mx[1][q]; sm[1][q] += v[q]; }
        mx[2][q] = v[q] > mx[2][q] ? v[q] : mx[2][q];
        sm[2][q] += v[q];
      }
    }
  float o[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    o[0][q] = mx[0][q] + sm[0][q] * (1.0f / 3.0f);   // count_include_pad=True: the divisor is the window size
    o[1][q] = mx[1][q] + sm[1][q] * (1.0f / 3.0f);
    o[2][q] = mx[2][q] + sm[2][q] * (1.0f / 9.0f);
  }
  const int64_t off = idx * 4;
  st4(p0 + off, o[0]);
  st4(p1 + off, o[1]);
  st4(p2 + off, o[2]);
}

// d x of the three pool sums, as a gather: pixel p collects, from every window q that contains it, g[q]/|window| (avg)
// and g[q] if p is that window's first arg-max (row-major scan, strict >, as ATen's max_pool2d) — decided from a 5x5
// register tile of x, no index tensor, no atomics.
template <int RH, int RW>
__device__ __forceinline__ void pool_bwd_one(const float (&xv)[5][5][4], const bool (&ok)[5][5], const float* __restrict__ gk, const Geo& g,
                                             int b, int h, int w, int cg, float acc[4]) {
  constexpr float inv = 1.0f / ((2 * RH + 1) * (2 * RW + 1));
#pragma unroll
  for (int di = -RH; di <= RH; ++di)
#pragma unroll
    for (int dj = -RW; dj <= RW; ++dj) {
      if (!ok[2 + di][2 + dj]) continue;   // window centre q = p + (di, dj) must be a real pixel
      float gq[4];
      f4(ld4(gk + (((int64_t)b * g.H + h + di) * g.W + w + dj) * g.C + cg * 4), gq);
      bool win[4] = {true, true, true, true};
      const int my = (-di + RH) * (2 * RW + 1) + (-dj + RW);   // p's position in q's row-major scan
#pragma unroll
      for (int a = -RH; a <= RH; ++a)
#pragma unroll
        for (int c = -RW; c <= RW; ++c) {
          const int pos = (a + RH) * (2 * RW + 1) + (c + RW);
          if (pos == my || !ok[2 + di + a][2 + dj + c]) continue;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float e = xv[2 + di + a][2 + dj + c][q], me = xv[2][2][q];
            win[q] = win[q] && (pos < my ? me > e : me >= e);
          }
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += gq[q] * inv + (win[q] ? gq[q] : 0.f);
    }
}

__global__ __launch_bounds__(kBlock) void skip_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g0, const float* __restrict__ g1,
                                                               const float* __restrict__ g2, const float* __restrict__ dxa, float* __restrict__ dx,
                                                               Geo g) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= g.npix * g.C4) return;
  int b, h, w, cg;
  decode(idx, g, b, h, w, cg);
  float xv[5][5][4];
  bool ok[5][5];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int hh = h + a - 2, ww = w + c - 2;
      ok[a][c] = hh >= 0 && hh < g.H && ww >= 0 && ww < g.W;
      if (ok[a][c]) f4(ld4(x + (((int64_t)b * g.H + hh) * g.W + ww) * g.C + cg * 4), xv[a][c]);
      else xv[a][c][0] = xv[a][c][1] = xv[a][c][2] = xv[a][c][3] = 0.f;
    }
  float acc[4];
  f4(ld4(dxa + idx * 4), acc);
  pool_bwd_one<1, 0>(xv, ok, g0, g, b, h, w, cg, acc);
  pool_bwd_one<0, 1>(xv, ok, g1, g, b, h, w, cg, acc);
  pool_bwd_one<1, 1>(xv, ok, g2, g, b, h, w, cg, acc);
  st4(dx + idx * 4, acc);
}

// ------------------------------------------------------------------------------------------------ gated branches, forward
// The grouped-conv weights of 16 adjacent groups, staged ONCE per workgroup in LDS.  A workgroup = 16 groups x 16 pixels: the 16 groups'
// 16 T weights of branch K are one contiguous run of memory (coalesced 16-byte loads), every pixel lane of a group then reads the same LDS
// word (a broadcast) and the 16 group lanes hit 16 different banks (group pitch 241 words).  Before, every (pixel, group) lane fetched its
// group's 960 bytes itself — as 240 dword loads (31 us per launch), then as 60 16-byte loads (17 us), each touching 64 cache lines.
constexpr int kGW = 16, kGP = kBlock / kGW;          // groups / pixels per workgroup
constexpr int kWPitch = 4 * 4 * (3 + 3 + 9) + 1;      // 240 weights per group + 1
constexpr int kWOff[3] = {0, 48, 96};                // branch K's weights inside a group's LDS row
template <int K>
__device__ __forceinline__ void stage_weights(const float* __restrict__ wk, int cg0, int C4, float* __restrict__ wl) {
  constexpr int T = Taps<K>::T, PER = 16 * T;
  const int ng = C4 - cg0 < kGW ? C4 - cg0 : kGW;
  const float* src = wk + (int64_t)cg0 * PER;
  for (int e4 = threadIdx.x; e4 < ng * PER / 4; e4 += kBlock) {
    const float4 v = ld4(src + 4 * e4);
    const int gl = (4 * e4) / PER, e = (4 * e4) % PER;
    float* d = wl + gl * kWPitch + kWOff[K] + e;
    d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
  }
}

template <int K>
__device__ __forceinline__ void conv4(const float* __restrict__ p, const float* __restrict__ wl, const float* __restrict__ bk, const Geo& g, int b,
                                      int h, int w, int cg, float c[4]) {
  constexpr int T = Taps<K>::T;
  f4(ld4(bk + cg * 4), c);
  const float* wg = wl + kWOff[K];   // this lane's group row of the staged weights
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int hh = h + Taps<K>::di(t), ww = w + Taps<K>::dj(t);
    if (hh < 0 || hh >= g.H || ww < 0 || ww >= g.W) continue;
    float pv[4];
    f4(ld4(p + (((int64_t)b * g.H + hh) * g.W + ww) * g.C + cg * 4), pv);
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[o] = fmaf(wg[(o * 4 + i) * T + t], pv[i], c[o]);
  }
}

__global__ __launch_bounds__(kBlock) void skip_branch_fwd_kernel(const float* __restrict__ x, const float* __restrict__ p0,
                                                                 const float* __restrict__ p1, const float* __restrict__ p2, Params P,
                                                                 float* __restrict__ c0, float* __restrict__ c1, float* __restrict__ c2,
                                                                 float* __restrict__ out, Geo g) {
  __shared__ float wlds[kGW * kWPitch];
  const int nchunk = (g.C4 + kGW - 1) / kGW;
  const int cg0 = ((int)blockIdx.x % nchunk) * kGW, cg = cg0 + (threadIdx.x & (kGW - 1));
  const int64_t pix = (int64_t)((int)blockIdx.x / nchunk) * kGP + (threadIdx.x / kGW);
  stage_weights<0>(P.w[0], cg0, g.C4, wlds);
  stage_weights<1>(P.w[1], cg0, g.C4, wlds);
  stage_weights<2>(P.w[2], cg0, g.C4, wlds);
  __syncthreads();
  if (cg >= g.C4 || pix >= g.npix) return;
  const int64_t idx = pix * g.C4 + cg;
  int b, h, w, cgd;
  decode(idx, g, b, h, w, cgd);
  const float* wl = wlds + (threadIdx.x & (kGW - 1)) * kWPitch;
  float c[3][4], xv[4], gam[4], o[4] = {0.f, 0.f, 0.f, 0.f};
  conv4<0>(p0, wl, P.b[0], g, b, h, w, cg, c[0]);
  conv4<1>(p1, wl, P.b[1], g, b, h, w, cg, c[1]);
  conv4<2>(p2, wl, P.b[2], g, b, h, w, cg, c[2]);
  f4(ld4(x + idx * 4), xv);
  f4(ld4(P.gamma + cg * 4), gam);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int f = k >> 1;   // 0: ffd13 / act_func13, 1: ffd33 / act_func33
    float fw[4], fb[4];
    f4(ld4(P.fw[f] + cg * 4), fw);
    f4(ld4(P.fb[f] + cg * 4), fb);
    const float enh = *P.enh[f], thr = *P.thr[f], al = *P.alpha[k];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float z = fmaf(xv[q] * geluf_(c[k][q]), fw[q], fb[q]);
      o[q] = fmaf(al, siluf_(enh * (z - thr)), o[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] *= gam[q];
  st4(c0 + idx * 4, c[0]);
  st4(c1 + idx * 4, c[1]);
  st4(c2 + idx * 4, c[2]);
  st4(out + idx * 4, o);
}

// ------------------------------------------------------------------------------------------------ gated branches, backward (pointwise)
// thread = (pixel slice s, group cg); pixels s, s+S, ...  Writes d c_k, the direct d x term, and per-slice partials of the
// 8 per-channel vectors; the 7 scalars are summed over the wave's 64 groups first (one slice per wave: C4 % 64 == 0).
__global__ __launch_bounds__(kBlock) void skip_branch_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                                 const float* __restrict__ c0, const float* __restrict__ c1,
                                                                 const float* __restrict__ c2, Params P, float* __restrict__ dc0,
                                                                 float* __restrict__ dc1, float* __restrict__ dc2, float* __restrict__ dxa,
                                                                 float* __restrict__ vpart, float* __restrict__ spart, int S, Geo g) {
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool valid = tid < (int64_t)S * g.C4;   // an idle lane keeps going: its wave's scalar sums are shuffled across all 64 lanes
  const int cg = valid ? (int)(tid % g.C4) : 0, s = (int)(tid / g.C4);
  float gam[4], fw[2][4], fb[2][4], enh[2], thr[2], al[3];
  f4(ld4(P.gamma + cg * 4), gam);
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    f4(ld4(P.fw[f] + cg * 4), fw[f]);
    f4(ld4(P.fb[f] + cg * 4), fb[f]);
    enh[f] = *P.enh[f];
    thr[f] = *P.thr[f];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) al[k] = *P.alpha[k];
  float vacc[kVec][4], sacc[kScal];
#pragma unroll
  for (int j = 0; j < kVec; ++j) vacc[j][0] = vacc[j][1] = vacc[j][2] = vacc[j][3] = 0.f;
#pragma unroll
  for (int j = 0; j < kScal; ++j) sacc[j] = 0.f;
  const float* cs[3] = {c0, c1, c2};
  float* dcs[3] = {dc0, dc1, dc2};
  for (int64_t pix = valid ? s : g.npix; pix < g.npix; pix += S) {
    const int64_t off = (pix * g.C4 + cg) * 4;
    float G[4], xv[4], dxv[4] = {0.f, 0.f, 0.f, 0.f}, ysum[4] = {0.f, 0.f, 0.f, 0.f};
    f4(ld4(dout + off), G);
    f4(ld4(x + off), xv);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int f = k >> 1;
      float c[4], dc[4];
      f4(ld4(cs[k] + off), c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = geluf_(c[q]), v = xv[q] * u, z = fmaf(v, fw[f][q], fb[f][q]), sarg = enh[f] * (z - thr[f]);
        const float y = siluf_(sarg), qg = gam[q] * G[q];
        ysum[q] = fmaf(al[k], y, ysum[q]);
        sacc[k] = fmaf(qg, y, sacc[k]);                       // d alpha_k
        const float ds = al[k] * qg * silu_gradf_(sarg);
        sacc[3 + 2 * f] = fmaf(ds, z - thr[f], sacc[3 + 2 * f]);   // d enh
        sacc[4 + 2 * f] += ds;                                // d thr (times -enh at the end)
        const float dz = ds * enh[f];
        vacc[1 + 2 * f][q] = fmaf(dz, v, vacc[1 + 2 * f][q]);  // d fw
        vacc[2 + 2 * f][q] += dz;                             // d fb
        const float dv = dz * fw[f][q];
        dxv[q] = fmaf(dv, u, dxv[q]);
        dc[q] = dv * xv[q] * gelu_gradf_(c[q]);
        vacc[5 + k][q] += dc[q];                              // d conv bias
      }
      st4(dcs[k] + off, dc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) vacc[0][q] = fmaf(G[q], ysum[q], vacc[0][q]);   // d gamma
    st4(dxa + off, dxv);
  }
  sacc[4] *= -enh[0];
  sacc[6] *= -enh[1];
  if (valid) {
#pragma unroll
    for (int j = 0; j < kVec; ++j) st4(vpart + ((int64_t)s * kVec + j) * g.C + cg * 4, vacc[j]);
  }
  // one partial row per WAVE (whatever slices its lanes belong to: the rows are only ever summed)
#pragma unroll
  for (int j = 0; j < kScal; ++j) sacc[j] = wave_sum(sacc[j]);
  if ((threadIdx.x & 63) == 0) {
    float* sp = spart + (tid >> 6) * kScal;
#pragma unroll
    for (int j = 0; j < kScal; ++j) sp[j] = sacc[j];
  }
}

// ------------------------------------------------------------------------------------------------ grouped conv, backward
// One launch, two roles.  Blocks [0, ndg): data gradient g_k = conv_k^T(d c_k), thread = (pixel, group).
// Blocks [ndg, ...): weight gradient, thread = one weight element (PyTorch's flat index) x pixel slice.
template <int K>
__device__ __forceinline__ void dgrad4(const float* __restrict__ dc, const float* __restrict__ wl, const Geo& g, int b, int h, int w, int cg,
                                       float o[4]) {
  constexpr int T = Taps<K>::T;
  o[0] = o[1] = o[2] = o[3] = 0.f;
  const float* wg = wl + kWOff[K];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int hh = h - Taps<K>::di(t), ww = w - Taps<K>::dj(t);   // the output pixel that read us through tap t
    if (hh < 0 || hh >= g.H || ww < 0 || ww >= g.W) continue;
    float dv[4];
    f4(ld4(dc + (((int64_t)b * g.H + hh) * g.W + ww) * g.C + cg * 4), dv);
#pragma unroll
    for (int oo = 0; oo < 4; ++oo)
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = fmaf(wg[(oo * 4 + i) * T + t], dv[oo], o[i]);
  }
}

// Weight gradient of branch K for 16 groups: a wave = 16 pixel sub-slices x 4 groups, every lane keeps its group's
// whole 4x4xT tap matrix in registers (coalesced float4 loads of d c and p), the 16 sub-slices are summed with
// in-row lane shuffles, and the lane of sub-slice 0 writes the group's 16T floats in PyTorch's (o, i, t) order.
template <int K>
__device__ __forceinline__ void wgrad_block(const float* __restrict__ dc, const float* __restrict__ p, float* __restrict__ dw, const Geo& g,
                                            int gchunk, int outer, int SW) {
  constexpr int T = Taps<K>::T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 15, cg_raw = gchunk * 16 + wave * 4 + (lane >> 4);
  const bool valid = cg_raw < g.C4;   // the last chunk of a narrow map (C4 % 16 != 0): whole 16-lane rows idle, shuffles stay in-row
  const int cg = valid ? cg_raw : 0;
  float acc[16 * T];
#pragma unroll
  for (int e = 0; e < 16 * T; ++e) acc[e] = 0.f;
  for (int64_t pix = valid ? sub + 16 * (int64_t)outer : g.npix; pix < g.npix; pix += 16 * (int64_t)SW) {
    const int w = (int)(pix % g.W), h = (int)((pix / g.W) % g.H);
    float dv[4];
    f4(ld4(dc + pix * g.C + cg * 4), dv);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int di = Taps<K>::di(t), dj = Taps<K>::dj(t);
      const int hh = h + di, ww = w + dj;
      if (hh < 0 || hh >= g.H || ww < 0 || ww >= g.W) continue;
      float pv[4];
      f4(ld4(p + (pix + (int64_t)di * g.W + dj) * g.C + cg * 4), pv);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[(o * 4 + i) * T + t] = fmaf(dv[o], pv[i], acc[(o * 4 + i) * T + t]);
    }
  }
#pragma unroll
  for (int e = 0; e < 16 * T; ++e) acc[e] = group_sum(acc[e], 16);
  if (sub == 0 && valid) {
    float* out = dw + (int64_t)cg * 16 * T;
#pragma unroll
    for (int e = 0; e < 16 * T; e += 4) *reinterpret_cast<float4*>(out + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
  }
}

__global__ __launch_bounds__(kBlock) void skip_conv_bwd_kernel(const float* __restrict__ dc0, const float* __restrict__ dc1,
                                                               const float* __restrict__ dc2, const float* __restrict__ p0,
                                                               const float* __restrict__ p1, const float* __restrict__ p2, Params P,
                                                               float* __restrict__ g0, float* __restrict__ g1, float* __restrict__ g2,
                                                               float* __restrict__ dw, int ndg, int SW, Geo g) {
  __shared__ float wlds[kGW * kWPitch];
  if ((int)blockIdx.x < ndg) {   // workgroup = 16 groups x 16 pixels, the groups' weights staged in LDS (see stage_weights)
    const int nchunk = (g.C4 + kGW - 1) / kGW;
    const int cg0 = ((int)blockIdx.x % nchunk) * kGW, cg = cg0 + (threadIdx.x & (kGW - 1));
    const int64_t pix = (int64_t)((int)blockIdx.x / nchunk) * kGP + (threadIdx.x / kGW);
    stage_weights<0>(P.w[0], cg0, g.C4, wlds);
    stage_weights<1>(P.w[1], cg0, g.C4, wlds);
    stage_weights<2>(P.w[2], cg0, g.C4, wlds);
    __syncthreads();
    if (cg >= g.C4 || pix >= g.npix) return;
    const int64_t idx = pix * g.C4 + cg;
    int b, h, w, cgd;
    decode(idx, g, b, h, w, cgd);
    const float* wl = wlds + (threadIdx.x & (kGW - 1)) * kWPitch;
    float o[4];
    dgrad4<0>(dc0, wl, g, b, h, w, cg, o);
    st4(g0 + idx * 4, o);
    dgrad4<1>(dc1, wl, g, b, h, w, cg, o);
    st4(g1 + idx * 4, o);
    dgrad4<2>(dc2, wl, g, b, h, w, cg, o);
    st4(g2 + idx * 4, o);
    return;
  }
  // weight-gradient role: block -> (branch, chunk of 16 groups, outer pixel slice); partial row `outer` of (SW, 60 C)
  const int nchunk = (g.C4 + 15) / 16;
  int r = (int)blockIdx.x - ndg;
  const int k = r / (nchunk * SW);
  r -= k * nchunk * SW;
  const int outer = r / nchunk, gchunk = r % nchunk;
  float* row = dw + (int64_t)outer * g.C * kWeightsPerChannel;
  if (k == 0) wgrad_block<0>(dc0, p0, row, g, gchunk, outer, SW);
  else if (k == 1) wgrad_block<1>(dc1, p1, row + (int64_t)g.C * 12, g, gchunk, outer, SW);
  else wgrad_block<2>(dc2, p2, row + (int64_t)g.C * 24, g, gchunk, outer, SW);
}

// ------------------------------------------------------------------------------------------------ host side
int make_geo(const char* who, int64_t B, int64_t H, int64_t W, int64_t C, Geo* g) {
  ADNM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "%s: empty shape", who);
  ADNM_REQUIRE(C % 4 == 0, "%s: C=%lld must be a multiple of 4 (one lane per 4-channel conv group)", who, (long long)C);
  ADNM_REQUIRE(B * H * W * C < (1ll << 31), "%s: tensor too large", who);
  g->B = (int)B, g->H = (int)H, g->W = (int)W, g->C = (int)C, g->C4 = (int)(C / 4);
  g->npix = B * H * W;
  return ADNM_OK;
}
int load_params(const char* who, const float* const* params, Params* P) {
  ADNM_REQUIRE(params, "%s: null parameter table", who);
  for (int i = 0; i < 18; ++i) ADNM_REQUIRE(params[i], "%s: parameter %d is null", who, i);
  for (int k = 0; k < 3; ++k) P->w[k] = params[2 * k], P->b[k] = params[2 * k + 1];
  P->fw[0] = params[6], P->fb[0] = params[7], P->fw[1] = params[8], P->fb[1] = params[9];
  P->enh[0] = params[10], P->thr[0] = params[11], P->enh[1] = params[12], P->thr[1] = params[13];
  P->alpha[0] = params[14], P->alpha[1] = params[15], P->alpha[2] = params[16];
  P->gamma = params[17];
  return ADNM_OK;
}
int slices_pointwise(const Geo& g) {   // pixels per lane of the pointwise backward = npix / slices (ADNM_SKIP_PPL: measurement aid)
  static const int ppl = [] {
    const char* e = getenv("ADNM_SKIP_PPL");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 1;   // measured in one session (3 gates per step): 4 pixels per lane 73 us, 2: 48, 1: 37
  }();
  return (int)(g.npix < ppl ? 1 : (g.npix / ppl > 1024 ? 1024 : g.npix / ppl));
}
int64_t scal_rows(const Geo& g, int S) { return adnm_cdiv((int64_t)S * g.C4, kBlock) * (kBlock / 64); }   // one per launched wave
int slices_wgrad(const Geo& g) { return (int)(g.npix / 64 < 1 ? 1 : (g.npix / 64 > 16 ? 16 : g.npix / 64)); }   // x 16 in-wave sub-slices
struct WsLayout {
  int64_t tensors, vpart, spart, wpart, total;   // float offsets
};
WsLayout ws_layout(const Geo& g) {
  WsLayout L;
  const int64_t n = g.npix * g.C;
  const int S = slices_pointwise(g), SW = slices_wgrad(g);
  L.tensors = 0;
  L.vpart = 7 * n;
  L.spart = L.vpart + (int64_t)S * kVec * g.C;
  L.wpart = L.spart + scal_rows(g, S) * kScal;
  L.total = L.wpart + (SW > 1 ? (int64_t)SW * g.C * kWeightsPerChannel : 0);
  return L;
}
}  // namespace

extern "C" int64_t adnm_skipgate_grad_floats(int64_t C) { return C * (kWeightsPerChannel + kVec) + kScal; }

extern "C" int64_t adnm_skipgate_bwd_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t C) {
  Geo g;
  if (make_geo("skipgate_bwd_ws_bytes", B, H, W, C, &g)) return -1;
  return ws_layout(g).total * (int64_t)sizeof(float);
}

extern "C" int adnm_skipgate_fwd(const float* x, const float* const* params, float* pooled, float* conv, float* out, int64_t B,
                                 int64_t H, int64_t W, int64_t C, adnm_stream_t stream) {
  ADNM_REQUIRE(x && pooled && conv && out, "skipgate_fwd: null pointer");
  Geo g;
  Params P;
  if (int rc = make_geo("skipgate_fwd", B, H, W, C, &g)) return rc;
  if (int rc = load_params("skipgate_fwd", params, &P)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = g.npix * g.C;
  const unsigned grid = (unsigned)adnm_cdiv(g.npix * g.C4, kBlock);
  {
    ADNM_PROF("skip_pool_fwd", st, 4.0 * 4 * n);
    skip_pool_fwd_kernel<<<grid, kBlock, 0, st>>>(x, pooled, pooled + n, pooled + 2 * n, g);
  }
  {
    ADNM_PROF("skip_branch_fwd", st, 4.0 * (8 * n + kWeightsPerChannel * C));
    const unsigned gridb = (unsigned)(adnm_cdiv(g.C4, kGW) * adnm_cdiv(g.npix, kGP));
    skip_branch_fwd_kernel<<<gridb, kBlock, 0, st>>>(x, pooled, pooled + n, pooled + 2 * n, P, conv, conv + n, conv + 2 * n, out, g);
  }
  ADNM_CHECK_LAUNCH("skipgate_fwd");
  return ADNM_OK;
}

extern "C" int adnm_skipgate_bwd(const float* dout, const float* x, const float* const* params, const float* pooled,
                                 const float* conv, float* dx, float* dparams, void* ws, int64_t ws_bytes, int64_t B, int64_t H, int64_t W,
                                 int64_t C, adnm_stream_t stream) {
  ADNM_REQUIRE(dout && x && pooled && conv && dx && dparams, "skipgate_bwd: null pointer");
  Geo g;
  Params P;
  if (int rc = make_geo("skipgate_bwd", B, H, W, C, &g)) return rc;
  if (int rc = load_params("skipgate_bwd", params, &P)) return rc;
  const WsLayout L = ws_layout(g);
  if (!ws || ws_bytes < L.total * (int64_t)sizeof(float)) {
    adnm_set_error("skipgate_bwd: workspace of %lld bytes needed, %lld given", (long long)(L.total * sizeof(float)), (long long)ws_bytes);
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = g.npix * g.C;
  float* f = (float*)ws;
  float *dc0 = f, *dc1 = f + n, *dc2 = f + 2 * n, *dxa = f + 3 * n, *g0 = f + 4 * n, *g1 = f + 5 * n, *g2 = f + 6 * n;
  float *vpart = f + L.vpart, *spart = f + L.spart, *wpart = f + L.wpart;
  const int S = slices_pointwise(g), SW = slices_wgrad(g);
  const int nwt = g.C * kWeightsPerChannel;
  float *dwgt = dparams, *dvec = dparams + nwt, *dscal = dvec + (int64_t)kVec * g.C;
  {
    ADNM_PROF("skip_branch_bwd", st, 4.0 * 9 * n);
    skip_branch_bwd_kernel<<<(unsigned)adnm_cdiv((int64_t)S * g.C4, kBlock), kBlock, 0, st>>>(dout, x, conv, conv + n, conv + 2 * n, P, dc0, dc1, dc2,
                                                                                             dxa, vpart, spart, S, g);
  }
  adnm_launch_fold("skip_vec_fold", vpart, S, kVec * g.C, {dvec, kVec * g.C}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  adnm_launch_fold("skip_scal_fold", spart, (int)scal_rows(g, S), kScal, {dscal, kScal}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  const int ndg = (int)(adnm_cdiv(g.C4, kGW) * adnm_cdiv(g.npix, kGP));
  {
    ADNM_PROF("skip_conv_bwd", st, 4.0 * (9 * n + 2.0 * nwt));
    skip_conv_bwd_kernel<<<(unsigned)(ndg + 3 * ((g.C4 + 15) / 16) * SW), kBlock, 0, st>>>(
        dc0, dc1, dc2, pooled, pooled + n, pooled + 2 * n, P, g0, g1, g2, SW > 1 ? wpart : dwgt, ndg, SW, g);
  }
  if (SW > 1) adnm_launch_fold("skip_wgrad_fold", wpart, SW, nwt, {dwgt, nwt}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  {
    ADNM_PROF("skip_pool_bwd", st, 4.0 * 6 * n);
    skip_pool_bwd_kernel<<<(unsigned)adnm_cdiv(g.npix * g.C4, kBlock), kBlock, 0, st>>>(x, g0, g1, g2, dxa, dx, g);
  }
  ADNM_CHECK_LAUNCH("skipgate_bwd");
  return ADNM_OK;
}
