// K4 (and the depthwise part of K3) — depthwise KxK "same" convolution applied directly on the channels-last
// token layout (B, H*W, C), replacing the reference's  view->permute->contiguous->nn.Conv2d(groups=C)->permute
// round trips (ADNssd.py:331-334,343-346,388-390; Vssd.py:232-234; model_untils.py:180-188,203-211;
// WTConv2d.py:81,86,123,146).
//
// HBM-bound stencil: 2*B*H*W*C elements of traffic.  Lane = 4 adjacent channels (16 B) x a strip of TW=4
// pixels along W, consecutive lanes = consecutive channel quads, so every tap is a fully coalesced row read
// and the K-1 halo rows/cols are re-served by L1/L2.  Weights are tap-major (KH*KW, C) so a tap is one float4.
// Backward = (a) dpre = dy * act'(conv(x)) [recomputed, not stored by forward], (b) dx = correlation of dpre
// with the flipped taps (same kernel, FLIP), (c) dw/db by a persistent pass with per-lane register
// accumulators, per-wave partials in the workspace and a deterministic fold (no atomics).
#include "adnm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int TW = 4;

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void fma4(float4& a, const float4& w, const float4& x) {
  a.x = fmaf(w.x, x.x, a.x); a.y = fmaf(w.y, x.y, a.y); a.z = fmaf(w.z, x.z, a.z); a.w = fmaf(w.w, x.w, a.w);
}
__device__ __forceinline__ float4 apply_act(float4 v, int act) {
  if (act == ADNM_ACT_SILU) return make_float4(siluf_(v.x), siluf_(v.y), siluf_(v.z), siluf_(v.w));
  if (act == ADNM_ACT_GELU) return make_float4(geluf_(v.x), geluf_(v.y), geluf_(v.z), geluf_(v.w));
  return v;
}
__device__ __forceinline__ float4 apply_act_grad(float4 v, int act) {
  if (act == ADNM_ACT_SILU) return make_float4(silu_gradf_(v.x), silu_gradf_(v.y), silu_gradf_(v.z), silu_gradf_(v.w));
  if (act == ADNM_ACT_GELU) return make_float4(gelu_gradf_(v.x), gelu_gradf_(v.y), gelu_gradf_(v.z), gelu_gradf_(v.w));
  return make_float4(1.f, 1.f, 1.f, 1.f);
}

// A lane's channel vector: ONE 16-byte access per pixel whatever the storage type — 4 fp32 channels or 8 bf16 channels (half the bytes AND
// half the load / store instructions per channel: a bf16 path that kept 4 channels = 8 bytes per lane moved half the bytes per request and
// ran no faster than fp32, profiles/r03_bf16_storage_ab.txt).  Arithmetic is fp32 in both.
template <typename T>
struct CVec;
template <>
struct CVec<float> {
  static constexpr int N = 4, Q = 1;
  float4 q[1];
};
template <>
struct CVec<uint16_t> {
  static constexpr int N = 8, Q = 2;
  float4 q[2];
};
template <typename T>
__device__ __forceinline__ CVec<T> cv_zero() {
  CVec<T> v;
#pragma unroll
  for (int i = 0; i < CVec<T>::Q; ++i) v.q[i] = f4zero();
  return v;
}
__device__ __forceinline__ CVec<float> cv_ld(const float* p) {
  CVec<float> v;
  v.q[0] = *reinterpret_cast<const float4*>(p);
  return v;
}
__device__ __forceinline__ CVec<uint16_t> cv_ld(const uint16_t* p) {
  const uint4 r = *reinterpret_cast<const uint4*>(p);
  CVec<uint16_t> v;
  v.q[0] = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
  v.q[1] = make_float4(__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u), __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u));
  return v;
}
__device__ __forceinline__ void cv_st(float* p, const CVec<float>& v) { *reinterpret_cast<float4*>(p) = v.q[0]; }
__device__ __forceinline__ void cv_st(uint16_t* p, const CVec<uint16_t>& v) {
  uint4 r;
  r.x = (uint32_t)f32_to_bf16(v.q[0].x) | ((uint32_t)f32_to_bf16(v.q[0].y) << 16);
  r.y = (uint32_t)f32_to_bf16(v.q[0].z) | ((uint32_t)f32_to_bf16(v.q[0].w) << 16);
  r.z = (uint32_t)f32_to_bf16(v.q[1].x) | ((uint32_t)f32_to_bf16(v.q[1].y) << 16);
  r.w = (uint32_t)f32_to_bf16(v.q[1].z) | ((uint32_t)f32_to_bf16(v.q[1].w) << 16);
  *reinterpret_cast<uint4*>(p) = r;
}
template <typename T>
__device__ __forceinline__ CVec<T> cv_ldf(const float* p) {   // N consecutive fp32 values (a tap, a bias)
  CVec<T> v;
#pragma unroll
  for (int i = 0; i < CVec<T>::Q; ++i) v.q[i] = *reinterpret_cast<const float4*>(p + 4 * i);
  return v;
}
template <typename T>
__device__ __forceinline__ void cv_stf(float* p, const CVec<T>& v) {
#pragma unroll
  for (int i = 0; i < CVec<T>::Q; ++i) *reinterpret_cast<float4*>(p + 4 * i) = v.q[i];
}
template <typename T>
__device__ __forceinline__ void cv_fma(CVec<T>& a, const CVec<T>& w, const CVec<T>& x) {
#pragma unroll
  for (int i = 0; i < CVec<T>::Q; ++i) fma4(a.q[i], w.q[i], x.q[i]);
}
template <typename T>
__device__ __forceinline__ void cv_add(CVec<T>& a, const CVec<T>& x) {
#pragma unroll
  for (int i = 0; i < CVec<T>::Q; ++i) a.q[i].x += x.q[i].x, a.q[i].y += x.q[i].y, a.q[i].z += x.q[i].z, a.q[i].w += x.q[i].w;
}

// MODE 0: y = act(conv(x) + bias) (+ addend)         [forward]
// MODE 1: y = dy * act'(conv(x) + bias)              [backward step (a); `aux` = dy with pixel stride ldaux]
// MODE 2: y = conv_flipped(x)                        [backward step (b); x = dpre]
template <typename T, int K, int MODE, bool WCM>
__global__ __launch_bounds__(kBlock) void dwconv_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias, const T* __restrict__ aux,
                                                        int64_t ldaux, T* __restrict__ y, int64_t ldy, int B, int H, int W,
                                                        int C, int act) {
  constexpr int R = K / 2, N = CVec<T>::N, Q = CVec<T>::Q;
  using V = CVec<T>;
  const int C4 = C / N;   // channel vectors per pixel
  const int WT = (W + TW - 1) / TW;
  const int64_t total = (int64_t)B * H * WT * C4;
  // XCD-aware block order: consecutive blockIdx round-robin over the 8 XCDs (private L2 each), but the halo rows of a
  // stencil are shared by VERTICALLY adjacent tiles.  Give every XCD one contiguous band of tiles so that re-reads of
  // the K-1 halo rows hit that XCD's own L2 (bijective for any grid size).
  const unsigned nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const unsigned bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int64_t idx = (int64_t)bid * kBlock + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % C4);
  int64_t t = idx / C4;
  const int wt = (int)(t % WT);
  t /= WT;
  const int h = (int)(t % H);
  const int b = (int)(t / H);
  const int c = cg * N;
  const int w0 = wt * TW;
  float wcmv[WCM ? N * K * K : 1];
  if (WCM) {
#pragma unroll
    for (int q = 0; q < N * K * K / 4; ++q) {
      const float4 t4 = *reinterpret_cast<const float4*>(wgt + (int64_t)c * (K * K) + 4 * q);
      wcmv[4 * q] = t4.x; wcmv[4 * q + 1] = t4.y; wcmv[4 * q + 2] = t4.z; wcmv[4 * q + 3] = t4.w;
    }
  }
  V acc[TW];
#pragma unroll
  for (int i = 0; i < TW; ++i) acc[i] = cv_zero<T>();
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int hh = h + i - R;
    if (hh < 0 || hh >= H) continue;
    const T* xr = x + ((int64_t)b * H + hh) * W * ldx + c;
    V row[TW + K - 1];
#pragma unroll
    for (int j = 0; j < TW + K - 1; ++j) {
      const int ww = w0 + j - R;
      row[j] = (ww >= 0 && ww < W) ? cv_ld(xr + (int64_t)ww * ldx) : cv_zero<T>();
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int tap = (MODE == 2) ? ((K - 1 - i) * K + (K - 1 - j)) : (i * K + j);
      // weights: tap-major (K*K, C): N consecutive floats per tap; or, wcm, nn.Conv2d's own channel-major (C, K*K): the lane's N channels
      // are N*K*K consecutive floats, fetched once (wcmv) and picked apart at compile-time indices
      V wv;
      if (WCM) {
#pragma unroll
        for (int qq = 0; qq < Q; ++qq)
          wv.q[qq] = make_float4(wcmv[(WCM ? (4 * qq) * K * K : 0) + tap * WCM], wcmv[(WCM ? (4 * qq + 1) * K * K : 0) + tap * WCM],
                                 wcmv[(WCM ? (4 * qq + 2) * K * K : 0) + tap * WCM], wcmv[(WCM ? (4 * qq + 3) * K * K : 0) + tap * WCM]);
      } else {
        wv = cv_ldf<T>(wgt + (int64_t)tap * C + c);
      }
#pragma unroll
      for (int p = 0; p < TW; ++p) cv_fma(acc[p], wv, row[p + j]);
    }
  }
  V bv = cv_zero<T>();
  if (MODE != 2 && bias) bv = cv_ldf<T>(bias + c);
#pragma unroll
  for (int p = 0; p < TW; ++p) {
    const int ww = w0 + p;
    if (ww >= W) break;
    const int64_t pix = ((int64_t)b * H + h) * W + ww;
    V v = acc[p];
    if (MODE == 0) {
      cv_add(v, bv);
#pragma unroll
      for (int qq = 0; qq < Q; ++qq) v.q[qq] = apply_act(v.q[qq], act);
      if (aux) cv_add(v, cv_ld(aux + pix * ldaux + c));
    } else if (MODE == 1) {
      cv_add(v, bv);
      const V d = cv_ld(aux + pix * ldaux + c);
#pragma unroll
      for (int qq = 0; qq < Q; ++qq) {
        const float4 g = apply_act_grad(v.q[qq], act);
        v.q[qq] = make_float4(d.q[qq].x * g.x, d.q[qq].y * g.y, d.q[qq].z * g.z, d.q[qq].w * g.w);
      }
    }
    cv_st(y + pix * ldy + c, v);
  }
}

// (c) weight / bias gradients.  block = K waves, wave i owns tap ROW i of every channel quad it sees, so a lane
// keeps only K float4 accumulators (not K*K) and reads ONE input row per tile; the K waves of a block share the
// tile's dpre strip through L1.  lane = (channel quad, pixel slot); grid = (C4/cgb, npb).  No cross-wave reduction:
// wave i writes taps [i*K, i*K+K) of the block's partial row; the middle wave also writes the bias gradient.
// part[blockIdx.y, tap, c]  (tap index K*K holds dbias)
constexpr int kWgSlices = 2;  // waves per tap row: more memory-level parallelism without more partial rows

// (bx, by, nby): this workgroup's position in ITS problem's (channel-quad block, pixel block) grid — blockIdx / gridDim of the single-problem
// launch, a block range of the grouped launch (dwconv_wgrad_multi_kernel: the tap gradients of a backward pass are leaves).
template <typename T, int K>
__device__ __forceinline__ void dwconv_wgrad_body(const T* __restrict__ dpre, int64_t ldd, const T* __restrict__ x, int64_t ldx,
                                                  float* __restrict__ part, int B, int H, int W, int C, int cgb, int wcm, const int bx,
                                                  const int by, const int nby) {
  constexpr int R = K / 2;
  constexpr int NT = K * K, N = CVec<T>::N;
  using V = CVec<T>;
  const int C4 = C / N;   // channel vectors per pixel
  __shared__ __attribute__((aligned(16))) float red[K][K + 1][64][N];
  const int wv = threadIdx.x >> 6;
  const int i = wv % K;            // tap row of this wave
  const int sl = wv / K;           // tile slice of this wave
  const int lane = threadIdx.x & 63;
  const int cgl = lane & (cgb - 1);
  const int slot = lane / cgb;
  const int slots = 64 / cgb;
  const int cg = bx * cgb + cgl;
  const bool cv = cg < C4;
  const int c = cv ? cg * N : 0;
  const int WT = (W + TW - 1) / TW;
  const int64_t tiles = (int64_t)B * H * WT;
  V aw[K];
  V ab = cv_zero<T>();
#pragma unroll
  for (int k = 0; k < K; ++k) aw[k] = cv_zero<T>();
  if (cv) {
    // Each block owns a CONTIGUOUS range of tiles (XCD-aware: the ranges of one XCD are adjacent, so the x / dpre rows
    // shared by its K tap-row waves and by vertically adjacent tiles are served by that XCD's L2), and keeps two tiles
    // in flight per lane (memory-level parallelism).
    const unsigned nb = (unsigned)nby, q8 = nb >> 3, r8 = nb & 7, xcd = (unsigned)by & 7, loc = (unsigned)by >> 3;
    const int64_t bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int64_t tpb = (tiles + nb - 1) / nb;
    const int64_t tbeg = bid * tpb, tend = (tbeg + tpb < tiles) ? tbeg + tpb : tiles;
    const int64_t stride = (int64_t)slots * kWgSlices;
    for (int64_t t0 = tbeg + sl * slots + slot; t0 < tend; t0 += 2 * stride) {
      V g[2][TW], row[2][TW + K - 1];
      bool live[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t t = t0 + u * stride;
        const int wt = (int)(t % WT);
        const int h = (int)((t / WT) % H);
        const int b = (int)(t / ((int64_t)WT * H));
        const int hh = h + i - R;
        live[u] = t < tend && hh >= 0 && hh < H;
        const int w0 = wt * TW;
#pragma unroll
        for (int p = 0; p < TW; ++p) {
          const int ww = w0 + p;
          g[u][p] = (live[u] && ww < W) ? cv_ld(dpre + (((int64_t)b * H + h) * W + ww) * ldd + c) : cv_zero<T>();
        }
        const T* xr = x + ((int64_t)b * H + (live[u] ? hh : 0)) * W * ldx + c;
#pragma unroll
        for (int j = 0; j < TW + K - 1; ++j) {
          const int ww = w0 + j - R;
          row[u][j] = (live[u] && ww >= 0 && ww < W) ? cv_ld(xr + (int64_t)ww * ldx) : cv_zero<T>();
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (i == R) {
#pragma unroll
          for (int p = 0; p < TW; ++p) cv_add(ab, g[u][p]);
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
          for (int p = 0; p < TW; ++p) cv_fma(aw[j], g[u][p], row[u][p + j]);
      }
    }
  }
  // fold the pixel slots of a wave, then the tile slices of the block (LDS), then write the block's partial row
#pragma unroll
  for (int j = 0; j <= K; ++j) {
    V& v = j < K ? aw[j] : ab;
#pragma unroll
    for (int qq = 0; qq < V::Q; ++qq) {
      v.q[qq].x = wave_sum_from(v.q[qq].x, cgb); v.q[qq].y = wave_sum_from(v.q[qq].y, cgb);
      v.q[qq].z = wave_sum_from(v.q[qq].z, cgb); v.q[qq].w = wave_sum_from(v.q[qq].w, cgb);
    }
  }
  if (sl == 1) {
#pragma unroll
    for (int j = 0; j <= K; ++j) cv_stf<T>(&red[i][j][lane][0], j < K ? aw[j] : ab);
  }
  __syncthreads();
  if (sl == 0) {
    float* dst = part + (int64_t)by * (NT + 1) * C;
#pragma unroll
    for (int j = 0; j <= K; ++j) {
      if (j == K && i != R) break;
      V v = j < K ? aw[j] : ab;
      cv_add(v, cv_ldf<T>(&red[i][j][lane][0]));
      if (lane < cgb && cv) {
        if (wcm && j < K) {   // channel-major partial row: column c*K*K + tap, so the fold emits nn.Conv2d's (C, K*K) layout
          float* q = dst + (int64_t)c * NT + i * K + j;
#pragma unroll
          for (int qq = 0; qq < V::Q; ++qq) {
            q[(4 * qq) * NT] = v.q[qq].x; q[(4 * qq + 1) * NT] = v.q[qq].y; q[(4 * qq + 2) * NT] = v.q[qq].z; q[(4 * qq + 3) * NT] = v.q[qq].w;
          }
        } else {
          cv_stf<T>(dst + (int64_t)(j < K ? i * K + j : NT) * C + c, v);
        }
      }
    }
  }
}

template <typename T, int K>
__global__ __launch_bounds__(64 * K * kWgSlices) void dwconv_wgrad_kernel(const T* __restrict__ dpre, int64_t ldd, const T* __restrict__ x,
                                                              int64_t ldx, float* __restrict__ part, int B, int H, int W, int C,
                                                              int cgb, int wcm) {
  dwconv_wgrad_body<T, K>(dpre, ldd, x, ldx, part, B, H, W, C, cgb, wcm, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// grouped form: up to kMaxWg queued problems per launch (fp32 tokens), descriptors in the kernel argument segment
constexpr int kMaxWg = 16;
struct WgLeaf {
  const float* dpre;
  int64_t ldd;
  const float* x;
  int64_t ldx;
  float* part;
  int B, H, W, C, cgb, wcm, gx, npb;
};
struct MultiWg {
  int count, blk_end[kMaxWg];
  WgLeaf p[kMaxWg];
};
template <int K>
__global__ __launch_bounds__(64 * K * kWgSlices) void dwconv_wgrad_multi_kernel(MultiWg by_value) {
  (void)by_value;
  const auto& m = *(const __attribute__((address_space(4))) MultiWg*)__builtin_amdgcn_kernarg_segment_ptr();
  int k = 0;
  while (k + 1 < m.count && (int)blockIdx.x >= m.blk_end[k]) ++k;
  const int lin = (int)blockIdx.x - (k ? m.blk_end[k - 1] : 0);
  const int gx = m.p[k].gx;
  dwconv_wgrad_body<float, K>(m.p[k].dpre, m.p[k].ldd, m.p[k].x, m.p[k].ldx, m.p[k].part, m.p[k].B, m.p[k].H, m.p[k].W, m.p[k].C, m.p[k].cgb,
                              m.p[k].wcm, lin % gx, lin / gx, m.p[k].npb);
}

// (c') 3x3 weight / bias gradients, "column walker": a lane owns one channel quad and one 4-pixel-wide strip of SEG rows and
// walks DOWN it with a rolling window of three x rows in registers, so every x row is fetched once per strip (not once per
// tap row and per vertically adjacent tile) and the dpre tile once (not once per tap-row wave): ~2.4x less L1/L2 traffic
// than the tap-row version above, which stays for 5x5 (its 25 accumulators + 5-row window do not fit in registers).
// block = cgb channel quads x (256 / cgb) strips; part[blockIdx.y, tap, c] as above.
template <typename T>
__global__ __launch_bounds__(kBlock) void dwconv_wgrad3_roll_kernel(const T* __restrict__ dpre, int64_t ldd, const T* __restrict__ x, int64_t ldx,
                                                                    float* __restrict__ part, int B, int H, int W, int C, int cgb, int SEG,
                                                                    int nseg, int wcm) {
  constexpr int K = 3, NT = 9, N = CVec<T>::N;
  using V = CVec<T>;
  __shared__ __attribute__((aligned(16))) float red[kBlock / 64 - 1][NT + 1][64][N];
  const int C4 = C / N;   // channel vectors per pixel
  const int WT = (W + TW - 1) / TW;
  const int cgl = threadIdx.x & (cgb - 1), sp = threadIdx.x / cgb, spb = kBlock / cgb;
  const int cg = blockIdx.x * cgb + cgl;
  const int64_t S = (int64_t)B * nseg * WT, sidx = (int64_t)blockIdx.y * spb + sp;
  const bool cv = cg < C4, live = cv && sidx < S;
  const int c = cv ? cg * N : 0;
  V aw[NT], ab = cv_zero<T>();
#pragma unroll
  for (int k = 0; k < NT; ++k) aw[k] = cv_zero<T>();
  if (live) {
    const int wt = (int)(sidx % WT), seg = (int)((sidx / WT) % nseg), b = (int)(sidx / ((int64_t)WT * nseg));
    const int h0 = seg * SEG, h1 = h0 + SEG < H ? h0 + SEG : H, w0 = wt * TW;
    auto load_x = [&](int hh, V (&r)[TW + 2]) {
      const bool ok = hh >= 0 && hh < H;
      const T* xr = x + ((int64_t)b * H + (ok ? hh : 0)) * W * ldx + c;
#pragma unroll
      for (int j = 0; j < TW + 2; ++j) {
        const int ww = w0 + j - 1;
        r[j] = (ok && ww >= 0 && ww < W) ? cv_ld(xr + (int64_t)ww * ldx) : cv_zero<T>();
      }
    };
    auto step = [&](int h, const V (&r0)[TW + 2], const V (&r1)[TW + 2], const V (&r2)[TW + 2]) {
      V g[TW];
#pragma unroll
      for (int p = 0; p < TW; ++p) {
        const int ww = w0 + p;
        g[p] = ww < W ? cv_ld(dpre + (((int64_t)b * H + h) * W + ww) * ldd + c) : cv_zero<T>();
        cv_add(ab, g[p]);
      }
#pragma unroll
      for (int j = 0; j < K; ++j)
#pragma unroll
        for (int p = 0; p < TW; ++p) {
          cv_fma(aw[j], g[p], r0[p + j]);
          cv_fma(aw[K + j], g[p], r1[p + j]);
          cv_fma(aw[2 * K + j], g[p], r2[p + j]);
        }
    };
    V ra[TW + 2], rb[TW + 2], rc[TW + 2];
    load_x(h0 - 1, ra);
    load_x(h0, rb);
    for (int h = h0; h < h1; h += 3) {   // three rows per trip so the window rotates by renaming, not by copying
      load_x(h + 1, rc);
      step(h, ra, rb, rc);
      if (h + 1 < h1) {
        load_x(h + 2, ra);
        step(h + 1, rb, rc, ra);
      }
      if (h + 2 < h1) {
        load_x(h + 3, rb);
        step(h + 2, rc, ra, rb);
      }
    }
  }
  // strips that share a wave (lanes differing in bits >= log2 cgb), then the waves through LDS
#pragma unroll
  for (int k = 0; k <= NT; ++k) {
    V& v = k < NT ? aw[k] : ab;
#pragma unroll
    for (int qq = 0; qq < V::Q; ++qq) {
      v.q[qq].x = wave_sum_from(v.q[qq].x, cgb); v.q[qq].y = wave_sum_from(v.q[qq].y, cgb);
      v.q[qq].z = wave_sum_from(v.q[qq].z, cgb); v.q[qq].w = wave_sum_from(v.q[qq].w, cgb);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k <= NT; ++k) cv_stf<T>(&red[wave - 1][k][lane][0], k < NT ? aw[k] : ab);
  }
  __syncthreads();
  if (wave == 0 && lane < cgb) {
    // lanes 0..cgb-1 of wave 0 hold channel vectors cgl = lane; lane l of every other wave holds the same vector iff (l & (cgb-1)) == lane,
    // and after wave_sum_from all lanes of a vector hold the wave's total, so reading lane `lane` of each wave is enough
    float* dst = part + (int64_t)blockIdx.y * (NT + 1) * C;
#pragma unroll
    for (int k = 0; k <= NT; ++k) {
      V v = k < NT ? aw[k] : ab;
#pragma unroll
      for (int wv = 0; wv < kBlock / 64 - 1; ++wv) cv_add(v, cv_ldf<T>(&red[wv][k][lane][0]));
      if (cv) {
        if (wcm && k < NT) {
          float* q = dst + (int64_t)c * NT + k;
#pragma unroll
          for (int qq = 0; qq < V::Q; ++qq) {
            q[(4 * qq) * NT] = v.q[qq].x; q[(4 * qq + 1) * NT] = v.q[qq].y; q[(4 * qq + 2) * NT] = v.q[qq].z; q[(4 * qq + 3) * NT] = v.q[qq].w;
          }
        } else {
          cv_stf<T>(dst + (int64_t)k * C + c, v);
        }
      }
    }
  }
}

// (c'') 5x5 weight / bias gradients, the same column walker (fp32 tokens).  The tap-row kernel above moves 7.5x its algorithmic bytes
// through L1 at 5x5 (every tap-row wave re-reads the d pre tile, every x row is fetched by five of them): 0.22 ms per step at config 2.
// Here a lane owns one channel quad and one TWR-pixel-wide strip of SEG rows, keeps a rolling window of FIVE x rows (TWR + 4 wide) in
// registers and all 25 tap accumulators: per row it loads TWR + 4 + TWR float4 for 25 TWR fma4 — ~4 units of L1 traffic per pixel
// instead of 15.  ~230 VGPRs: two waves per SIMD, which is what a workgroup of the grouped launch gets anyway.
constexpr int kWalkTW = 1;
template <int K, int TWR>
__device__ __forceinline__ void dwconv_wgrad_walk_body(const float* __restrict__ dpre, int64_t ldd, const float* __restrict__ x, int64_t ldx,
                                                       float* __restrict__ part, int B, int H, int W, int C, int cgb, int SEG, int nseg, int wcm,
                                                       const int bx, const int by) {
  constexpr int R = K / 2, NT = K * K, XW = TWR + K - 1;
  __shared__ __attribute__((aligned(16))) float red[kBlock / 64 - 1][64][4];
  const int C4 = C >> 2;
  const int WT = (W + TWR - 1) / TWR;
  const int cgl = threadIdx.x & (cgb - 1), sp = threadIdx.x / cgb, spb = kBlock / cgb;
  const int cg = bx * cgb + cgl;
  const int64_t S = (int64_t)B * nseg * WT, sidx = (int64_t)by * spb + sp;
  const bool cv = cg < C4, live = cv && sidx < S;
  const int c = cv ? cg * 4 : 0;
  float4 aw[NT], ab = f4zero();
#pragma unroll
  for (int k = 0; k < NT; ++k) aw[k] = f4zero();
  if (live) {
    const int wt = (int)(sidx % WT), seg = (int)((sidx / WT) % nseg), b = (int)(sidx / ((int64_t)WT * nseg));
    const int h0 = seg * SEG, h1 = h0 + SEG < H ? h0 + SEG : H, w0 = wt * TWR;
    float4 win[K][XW];
    auto load_x = [&](int hh, float4 (&r)[XW]) {
      const bool ok = hh >= 0 && hh < H;
      const float* xr = x + ((int64_t)b * H + (ok ? hh : 0)) * W * ldx + c;
#pragma unroll
      for (int j = 0; j < XW; ++j) {
        const int ww = w0 + j - R;
        r[j] = (ok && ww >= 0 && ww < W) ? *reinterpret_cast<const float4*>(xr + (int64_t)ww * ldx) : f4zero();
      }
    };
#pragma unroll
    for (int i = 0; i < K - 1; ++i) load_x(h0 - R + i, win[i]);
    // one row: tap row 0 (the window's oldest row) first, so that the NEXT row's loads — into that slot — can be issued under the other
    // four tap rows' 80 fma4
    auto step = [&](int hr, int j) {
      float4 g[TWR];
#pragma unroll
      for (int p = 0; p < TWR; ++p) {
        const int ww = w0 + p;
        g[p] = ww < W ? *reinterpret_cast<const float4*>(dpre + (((int64_t)b * H + hr) * W + ww) * ldd + c) : f4zero();
        ab.x += g[p].x, ab.y += g[p].y, ab.z += g[p].z, ab.w += g[p].w;
      }
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int t = 0; t < K; ++t)
#pragma unroll
          for (int p = 0; p < TWR; ++p) fma4(aw[i * K + t], g[p], win[(j + i) % K][p + t]);
    };
    // (a branch-free copy of the full trips lets the compiler hoist every load of a trip: 384 VGPRs = one wave per SIMD, measured 1.5x slower)
    for (int h = h0; h < h1; h += K) {   // K rows per trip so the window rotates by renaming, not by copying
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (h + j < h1) {
          load_x(h + j + R, win[(j + K - 1) % K]);
          step(h + j, j);
        }
      }
    }
  }
  // strips that share a wave (lanes differing in bits >= log2 cgb), then the waves through LDS — one tap at a time (26 float4 per lane
  // at once would be 80 KB of LDS), then the workgroup's partial row
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* dst = part + (int64_t)by * (NT + 1) * C;
#pragma unroll
  for (int k = 0; k <= NT; ++k) {
    float4 v = k < NT ? aw[k] : ab;
    v.x = wave_sum_from(v.x, cgb), v.y = wave_sum_from(v.y, cgb), v.z = wave_sum_from(v.z, cgb), v.w = wave_sum_from(v.w, cgb);
    if (wave > 0) *reinterpret_cast<float4*>(&red[wave - 1][lane][0]) = v;
    __syncthreads();
    if (wave == 0 && lane < cgb) {
#pragma unroll
      for (int wv = 0; wv < kBlock / 64 - 1; ++wv) {
        const float4 o = *reinterpret_cast<const float4*>(&red[wv][lane][0]);   // lane l of every wave holds quad l & (cgb - 1)'s wave total
        v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
      }
      if (cv) {
        if (wcm && k < NT) {
          float* q = dst + (int64_t)c * NT + k;
          q[0] = v.x, q[NT] = v.y, q[2 * NT] = v.z, q[3 * NT] = v.w;
        } else {
          *reinterpret_cast<float4*>(dst + (int64_t)k * C + c) = v;
        }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void dwconv_wgrad5_walk_kernel(const float* __restrict__ dpre, int64_t ldd, const float* __restrict__ x, int64_t ldx,
                                                                    float* __restrict__ part, int B, int H, int W, int C, int cgb, int SEG, int nseg,
                                                                    int wcm) {
  dwconv_wgrad_walk_body<5, kWalkTW>(dpre, ldd, x, ldx, part, B, H, W, C, cgb, SEG, nseg, wcm, (int)blockIdx.x, (int)blockIdx.y);
}

// grouped form of the 5x5 walker
struct WalkLeaf {
  const float* dpre;
  int64_t ldd;
  const float* x;
  int64_t ldx;
  float* part;
  int B, H, W, C, cgb, wcm, gx, npb, seg, nseg;
};
struct MultiWalk {
  int count, blk_end[kMaxWg];
  WalkLeaf p[kMaxWg];
};
__global__ __launch_bounds__(kBlock) void dwconv_wgrad5_walk_multi_kernel(MultiWalk by_value) {
  (void)by_value;
  const auto& m = *(const __attribute__((address_space(4))) MultiWalk*)__builtin_amdgcn_kernarg_segment_ptr();
  int k = 0;
  while (k + 1 < m.count && (int)blockIdx.x >= m.blk_end[k]) ++k;
  const int lin = (int)blockIdx.x - (k ? m.blk_end[k - 1] : 0);
  const int gx = m.p[k].gx;
  dwconv_wgrad_walk_body<5, kWalkTW>(m.p[k].dpre, m.p[k].ldd, m.p[k].x, m.p[k].ldx, m.p[k].part, m.p[k].B, m.p[k].H, m.p[k].W, m.p[k].C, m.p[k].cgb,
                                     m.p[k].seg, m.p[k].nseg, m.p[k].wcm, lin % gx, lin / gx);
}

struct WGeo {
  int cgb, gx, npb, rows;
  int seg, nseg;   // 3x3 column walker: rows per strip, strips per image column (0 = tap-row kernel)
};
WGeo wgeo(int64_t B, int64_t H, int64_t W, int64_t C, int K, int cn = 4) {   // cn: channels per lane (4 fp32 / 8 bf16)
  WGeo g;
  const int64_t C4 = C / cn;
  g.cgb = 1;
  while (g.cgb < 64 && g.cgb < C4) g.cgb <<= 1;
  g.gx = (int)adnm_cdiv(C4, g.cgb);
  const int slots = 64 / g.cgb;
  const int64_t tiles = B * H * adnm_cdiv(W, TW);
  // tiles per lane: 8 on the big maps; on the deep ones fewer (down to 2 = one trip of the two-tiles-in-flight loop), so that the
  // grid still has ~256 workgroups instead of a handful of lanes walking a long dependent chain
  // (queued for the grouped launch the other problems fill the chip: longer walks per lane = fewer workgroup tails and fewer partial
  // rows for the fold; ADNM_DW_WG_TPL: measurement aid)
  static const int grouped_tpl = [] {
    const char* e = getenv("ADNM_DW_WG_TPL");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 16;
  }();
  const bool walker = K == 3 && H * W >= 128 * 128;
  const int64_t tmax = !walker && adnm_leafq_active() ? grouped_tpl : 8;
  int64_t tpl = (tiles * g.gx) / ((int64_t)slots * 256);
  if (tmax > 8 && tpl >= 8) tpl = tmax;
  tpl = tpl < 2 ? 2 : (tpl > tmax ? tmax : tpl);
  int64_t npb = adnm_cdiv(tiles, (int64_t)slots * tpl);
  int64_t cap = (4 << 20) / ((int64_t)(K * K + 1) * C * 4);     // keep the partials under ~4 MB
  if (cap > 1024) cap = 1024;
  if (cap < 32) cap = 32;
  if (npb > cap) npb = cap;
  if (npb < 1) npb = 1;
  g.npb = (int)npb;
  g.rows = g.npb;
  g.seg = g.nseg = 0;
  if (walker) {   // column walker — measured: 37.6 -> 29.7 us on the 128x128x128 stencil, slower than the tap-row
                                         // kernel on 64x64 and smaller maps (short strips: halo rows and the block reduction dominate)
    const int64_t WT = adnm_cdiv(W, TW);
    int64_t seg = (H * B * WT * C4) / 65536;
    seg = seg < 2 ? 2 : (seg > 16 ? 16 : seg);
    if (seg > H) seg = H;
    g.seg = (int)seg;
    g.nseg = (int)adnm_cdiv(H, seg);
    const int64_t S = B * g.nseg * WT;
    g.npb = (int)adnm_cdiv(S, kBlock / g.cgb);
    g.rows = g.npb;
  }
  return g;
}

// geometry of the 5x5 column walker (fp32 tokens): strips of SEG rows; a strip that starts inside an image re-reads 4 halo rows, so
// strips are at least 16 rows long, and as long as ~64 K lanes' worth of strips allow
WGeo wgeo_walk5(int64_t B, int64_t H, int64_t W, int64_t C) {
  WGeo g;
  const int64_t C4 = C / 4;
  g.cgb = 1;
  while (g.cgb < 64 && g.cgb < C4) g.cgb <<= 1;
  g.gx = (int)adnm_cdiv(C4, g.cgb);
  const int64_t WT = adnm_cdiv(W, kWalkTW);
  static const int seg_min = [] {   // measurement aid
    const char* e = getenv("ADNM_DW_WALK_SEG");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 16;
  }();
  int64_t seg = (H * B * WT * C4) / 65536;
  seg = seg < seg_min ? seg_min : (seg > 64 ? 64 : seg);
  if (seg > H) seg = H;
  g.seg = (int)seg;
  g.nseg = (int)adnm_cdiv(H, seg);
  g.npb = (int)adnm_cdiv(B * g.nseg * WT, kBlock / g.cgb);
  g.rows = g.npb;
  return g;
}

int check(const char* who, const void* x, int64_t B, int64_t H, int64_t W, int64_t C, int KH, int KW, int act, int dtype) {
  ADNM_REQUIRE(x, "%s: null pointer", who);
  ADNM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: shape B=%lld H=%lld W=%lld C=%lld (C must be a multiple of 4)", who,
               (long long)B, (long long)H, (long long)W, (long long)C);
  ADNM_REQUIRE(KH == KW && (KH == 3 || KH == 5), "%s: kernel %dx%d not in {3x3, 5x5}", who, KH, KW);
  ADNM_REQUIRE(act >= ADNM_ACT_NONE && act <= ADNM_ACT_GELU, "%s: bad activation %d", who, act);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "%s: bad dtype %d", who, dtype);
  ADNM_REQUIRE(dtype == ADNM_F32 || C % 8 == 0, "%s: bf16 tokens move 8 channels (16 bytes) per lane: C=%lld must be a multiple of 8", who, (long long)C);
  ADNM_REQUIRE(B * H * W * C < (1ll << 40), "%s: tensor too large", who);
  return ADNM_OK;
}

template <typename T, int MODE>
void launch_conv(const void* x, int64_t ldx, const float* wgt, const float* bias, const void* aux, int64_t ldaux, void* y, int64_t ldy,
                 int64_t B, int64_t H, int64_t W, int64_t C, int K, int act, int wcm, hipStream_t st) {
  const int64_t total = B * H * adnm_cdiv(W, TW) * (C / CVec<T>::N);
  const unsigned grid = (unsigned)adnm_cdiv(total, kBlock);
#define ADNM_DWCONV(KK, WCMV)                                                                                                       \
  dwconv_kernel<T, KK, MODE, WCMV><<<grid, kBlock, 0, st>>>((const T*)x, ldx, wgt, bias, (const T*)aux, ldaux, (T*)y, ldy, (int)B, (int)H, \
                                                            (int)W, (int)C, act)
  ADNM_PROF(K == 3 ? "dwconv_k3" : "dwconv_k5", st, (double)sizeof(T) * B * H * W * C * (MODE == 1 ? 3 : (aux ? 3 : 2)));
  if (K == 3) {
    if (wcm) ADNM_DWCONV(3, true);
    else ADNM_DWCONV(3, false);
  } else {
    if (wcm) ADNM_DWCONV(5, true);
    else ADNM_DWCONV(5, false);
  }
#undef ADNM_DWCONV
}

template <typename T>
void launch_wgrad(const void* dpre, int64_t ldd, const void* x, int64_t ldx, float* part, float* dwgt, float* dbias, int64_t B, int64_t H,
                  int64_t W, int64_t C, int K, int wcm, hipStream_t st) {
  if (K == 5 && sizeof(T) == 4) {   // the 5x5 column walker (fp32 tokens); a leaf of the backward pass: may wait for the grouped launch
    const WGeo g = wgeo_walk5(B, H, W, C);
    static_assert(sizeof(WalkLeaf) <= sizeof(AdnmLeaf::args), "WalkLeaf must fit a leaf record");
    WalkLeaf w{(const float*)dpre, ldd, (const float*)x, ldx, part, (int)B, (int)H, (int)W, (int)C, g.cgb, wcm, g.gx, g.npb, g.seg, g.nseg};
    AdnmLeaf leaf;
    leaf.kind = ADNM_LEAF_DWCONV_WGRAD_K5, leaf.grid = g.gx * g.npb, leaf.prec = 0, leaf.prof = "dwconv_wgrad_k5", leaf.bytes = 4.0 * B * H * W * C * 2;
    memcpy(leaf.args, &w, sizeof(w));
    if (!adnm_leafq_push(leaf)) {
      ADNM_PROF("dwconv_wgrad_k5", st, 4.0 * B * H * W * C * 2);
      dwconv_wgrad5_walk_kernel<<<dim3(g.gx, g.npb), kBlock, 0, st>>>((const float*)dpre, ldd, (const float*)x, ldx, part, (int)B, (int)H, (int)W, (int)C, g.cgb,
                                                                     g.seg, g.nseg, wcm);
    }
    adnm_launch_fold("dwconv_wgrad_fold", part, g.rows, 26 * (int)C, {dwgt, 25 * (int)C}, {dbias, (int)C}, {nullptr, 0}, {nullptr, 0}, st);
    return;
  }
  const WGeo g = wgeo(B, H, W, C, K, CVec<T>::N);
  const dim3 grid(g.gx, g.npb);
  if (K == 3 && g.seg > 0)
    { ADNM_PROF("dwconv_wgrad_k3", st, (double)sizeof(T) * B * H * W * C * 2); dwconv_wgrad3_roll_kernel<T><<<grid, kBlock, 0, st>>>((const T*)dpre, ldd, (const T*)x, ldx, part, (int)B, (int)H, (int)W, (int)C, g.cgb, g.seg, g.nseg, wcm); }
  else {
    bool queued = false;
    if (sizeof(T) == 4) {   // a leaf of the backward pass: may wait for the grouped launch (fp32 tokens)
      static_assert(sizeof(WgLeaf) <= sizeof(AdnmLeaf::args), "WgLeaf must fit a leaf record");
      WgLeaf w{(const float*)dpre, ldd, (const float*)x, ldx, part, (int)B, (int)H, (int)W, (int)C, g.cgb, wcm, (int)g.gx, (int)g.npb};
      AdnmLeaf leaf;
      leaf.kind = K == 3 ? ADNM_LEAF_DWCONV_WGRAD_K3 : ADNM_LEAF_DWCONV_WGRAD_K5, leaf.grid = (int)(g.gx * g.npb), leaf.prec = 0;
      leaf.prof = K == 3 ? "dwconv_wgrad_k3" : "dwconv_wgrad_k5", leaf.bytes = 4.0 * B * H * W * C * 2;
      memcpy(leaf.args, &w, sizeof(w));
      queued = adnm_leafq_push(leaf);
    }
    if (!queued) {
      if (K == 3)
        { ADNM_PROF("dwconv_wgrad_k3", st, (double)sizeof(T) * B * H * W * C * 2); dwconv_wgrad_kernel<T, 3><<<grid, 64 * 3 * kWgSlices, 0, st>>>((const T*)dpre, ldd, (const T*)x, ldx, part, (int)B, (int)H, (int)W, (int)C, g.cgb, wcm); }
      else
        { ADNM_PROF("dwconv_wgrad_k5", st, (double)sizeof(T) * B * H * W * C * 2); dwconv_wgrad_kernel<T, 5><<<grid, 64 * 5 * kWgSlices, 0, st>>>((const T*)dpre, ldd, (const T*)x, ldx, part, (int)B, (int)H, (int)W, (int)C, g.cgb, wcm); }
    }
  }
  adnm_launch_fold("dwconv_wgrad_fold", part, g.rows, (K * K + 1) * (int)C, {dwgt, K * K * (int)C}, {dbias, (int)C}, {nullptr, 0}, {nullptr, 0}, st);
}

}  // namespace

extern "C" int adnm_dwconv_fwd(const void* x, int64_t ldx, const float* wgt, const float* bias, const void* addend, int64_t ldadd,
                               void* y, int64_t ldy, int64_t B, int64_t H, int64_t W, int64_t C, int KH, int KW, int act, int wlayout,
                               int dtype, adnm_stream_t stream) {
  if (int rc = check("dwconv_fwd", x, B, H, W, C, KH, KW, act, dtype)) return rc;
  ADNM_REQUIRE(wgt && y, "dwconv_fwd: null pointer");
  ADNM_REQUIRE(wlayout == 0 || wlayout == 1, "dwconv_fwd: weight layout %d not in {0 tap-major, 1 channel-major}", wlayout);
  const int al = dtype == ADNM_BF16 ? 8 : 4;   // elements per 16-byte lane access
  ADNM_REQUIRE(ldx >= C && ldy >= C && ldx % al == 0 && ldy % al == 0 && (!addend || (ldadd >= C && ldadd % al == 0)) &&
                   ((uintptr_t)x | (uintptr_t)y | (uintptr_t)addend) % 16 == 0,
               "dwconv_fwd: pixel strides must be >= C, rows and strides 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ADNM_F32) launch_conv<float, 0>(x, ldx, wgt, bias, addend, ldadd, y, ldy, B, H, W, C, KH, act, wlayout, st);
  else launch_conv<uint16_t, 0>(x, ldx, wgt, bias, addend, ldadd, y, ldy, B, H, W, C, KH, act, wlayout, st);
  ADNM_CHECK_LAUNCH("dwconv_fwd");
  return ADNM_OK;
}

extern "C" int64_t adnm_dwconv_bwd_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t C, int KH, int KW) {
  if (B <= 0 || H <= 0 || W <= 0 || C < 4) return 0;
  int64_t rows = wgeo(B, H, W, C, KH).rows;
  if (C % 8 == 0) {   // (the query does not know the storage type: bf16 tokens move 8 channels per lane, another geometry)
    const int64_t r8 = wgeo(B, H, W, C, KH, 8).rows;
    rows = r8 > rows ? r8 : rows;
  }
  if (KH == 5) {   // (the query does not know the storage type: fp32 tokens take the column walker's geometry)
    const int64_t r5 = wgeo_walk5(B, H, W, C).rows;
    rows = r5 > rows ? r5 : rows;
  }
  return rows * (KH * KW + 1) * C * (int64_t)sizeof(float);
}

extern "C" int adnm_dwconv_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* wgt, const float* bias,
                               void* dpre, void* dx, int64_t lddx, float* dwgt, float* dbias, void* ws, int64_t ws_bytes, int64_t B,
                               int64_t H, int64_t W, int64_t C, int KH, int KW, int act, int wlayout, int dtype, adnm_stream_t stream) {
  if (int rc = check("dwconv_bwd", x, B, H, W, C, KH, KW, act, dtype)) return rc;
  ADNM_REQUIRE(dy && wgt && dx, "dwconv_bwd: null pointer");
  ADNM_REQUIRE(wlayout == 0 || wlayout == 1, "dwconv_bwd: weight layout %d not in {0 tap-major, 1 channel-major}", wlayout);
  ADNM_REQUIRE(act == ADNM_ACT_NONE || dpre, "dwconv_bwd: dpre scratch required when an activation is fused");
  const int al = dtype == ADNM_BF16 ? 8 : 4;   // elements per 16-byte lane access
  ADNM_REQUIRE(ldx >= C && lddy >= C && lddx >= C && ldx % al == 0 && lddy % al == 0 && lddx % al == 0 &&
                   ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dpre) % 16 == 0,
               "dwconv_bwd: pixel strides must be >= C, rows and strides 16-byte aligned");
  if (!ws || ws_bytes < adnm_dwconv_bwd_ws_bytes(B, H, W, C, KH, KW)) {
    adnm_set_error("dwconv_bwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_dwconv_bwd_ws_bytes(B, H, W, C, KH, KW));
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const void* g = dy;
  int64_t ldg = lddy;
  if (dtype == ADNM_F32) {
    if (act != ADNM_ACT_NONE) {
      launch_conv<float, 1>(x, ldx, wgt, bias, dy, lddy, dpre, C, B, H, W, C, KH, act, wlayout, st);
      g = dpre; ldg = C;
    }
    launch_conv<float, 2>(g, ldg, wgt, nullptr, nullptr, 0, dx, lddx, B, H, W, C, KH, 0, wlayout, st);
    if (dwgt) launch_wgrad<float>(g, ldg, x, ldx, (float*)ws, dwgt, dbias, B, H, W, C, KH, wlayout, st);
  } else {
    if (act != ADNM_ACT_NONE) {
      launch_conv<uint16_t, 1>(x, ldx, wgt, bias, dy, lddy, dpre, C, B, H, W, C, KH, act, wlayout, st);
      g = dpre; ldg = C;
    }
    launch_conv<uint16_t, 2>(g, ldg, wgt, nullptr, nullptr, 0, dx, lddx, B, H, W, C, KH, 0, wlayout, st);
    if (dwgt) launch_wgrad<uint16_t>(g, ldg, x, ldx, (float*)ws, dwgt, dbias, B, H, W, C, KH, wlayout, st);
  }
  ADNM_CHECK_LAUNCH("dwconv_bwd");
  return ADNM_OK;
}

// The weight (and bias) gradient alone: g = the gradient at the conv's output BEFORE the fused activation (adnm_dwconv_bwd's dpre, or dy
// itself without one).  Lets the caller run it on another stream than the input-gradient chain (adnm_dwconv_bwd with dwgt = NULL).
extern "C" int adnm_dwconv_wgrad(const void* g, int64_t ldg, const void* x, int64_t ldx, float* dwgt, float* dbias, void* ws, int64_t ws_bytes,
                                 int64_t B, int64_t H, int64_t W, int64_t C, int KH, int KW, int wlayout, int dtype, adnm_stream_t stream) {
  if (int rc = check("dwconv_wgrad", x, B, H, W, C, KH, KW, ADNM_ACT_NONE, dtype)) return rc;
  ADNM_REQUIRE(g && dwgt, "dwconv_wgrad: null pointer");
  ADNM_REQUIRE(wlayout == 0 || wlayout == 1, "dwconv_wgrad: weight layout %d not in {0 tap-major, 1 channel-major}", wlayout);
  const int al = dtype == ADNM_BF16 ? 8 : 4;   // elements per 16-byte lane access
  ADNM_REQUIRE(ldx >= C && ldg >= C && ldx % al == 0 && ldg % al == 0 && ((uintptr_t)x | (uintptr_t)g) % 16 == 0,
               "dwconv_wgrad: pixel strides must be >= C, rows and strides 16-byte aligned");
  if (!ws || ws_bytes < adnm_dwconv_bwd_ws_bytes(B, H, W, C, KH, KW)) {
    adnm_set_error("dwconv_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_dwconv_bwd_ws_bytes(B, H, W, C, KH, KW));
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ADNM_F32) launch_wgrad<float>(g, ldg, x, ldx, (float*)ws, dwgt, dbias, B, H, W, C, KH, wlayout, st);
  else launch_wgrad<uint16_t>(g, ldg, x, ldx, (float*)ws, dwgt, dbias, B, H, W, C, KH, wlayout, st);
  ADNM_CHECK_LAUNCH("dwconv_wgrad");
  return ADNM_OK;
}


// the queued depthwise tap-gradient problems of one kernel size, kMaxWg per launch
int adnm_dwconv_wgrad_launch_multi(const AdnmLeaf* const* items, int n, int K, hipStream_t st) {
  if (K == 5) {   // every queued 5x5 problem is a column-walker problem (launch_wgrad)
    for (int i = 0; i < n;) {
      MultiWalk m;
      m.count = 0;
      int blocks = 0;
      double bytes = 0;
      for (; i < n && m.count < kMaxWg; ++i) {
        memcpy(&m.p[m.count], items[i]->args, sizeof(WalkLeaf));
        blocks += items[i]->grid;
        m.blk_end[m.count++] = blocks;
        bytes += items[i]->bytes;
      }
      for (int k = m.count; k < kMaxWg; ++k) m.blk_end[k] = blocks;
      ADNM_PROF("dwconv_wgrad_k5", st, bytes);
      dwconv_wgrad5_walk_multi_kernel<<<(unsigned)blocks, kBlock, 0, st>>>(m);
    }
    ADNM_CHECK_LAUNCH("dwconv_wgrad (grouped)");
    return ADNM_OK;
  }
  for (int i = 0; i < n;) {
    MultiWg m;
    m.count = 0;
    int blocks = 0;
    double bytes = 0;
    for (; i < n && m.count < kMaxWg; ++i) {
      memcpy(&m.p[m.count], items[i]->args, sizeof(WgLeaf));
      blocks += items[i]->grid;
      m.blk_end[m.count++] = blocks;
      bytes += items[i]->bytes;
    }
    for (int k = m.count; k < kMaxWg; ++k) m.blk_end[k] = blocks;
    ADNM_PROF("dwconv_wgrad_k3", st, bytes);
    dwconv_wgrad_multi_kernel<3><<<(unsigned)blocks, 64 * 3 * kWgSlices, 0, st>>>(m);
  }
  ADNM_CHECK_LAUNCH("dwconv_wgrad (grouped)");
  return ADNM_OK;
}
