// K3 — one ANALYSIS LEVEL of WTConv2d (reference models/WTConv2d.py:111-124) as one kernel:
//
//   sub = DWT_haar(cur)                      (B, h2, w2, 4C), channel = c*4 + k, k = LL, (rows differ), (cols differ), (diagonal); odd sizes
//                                            zero-padded right / bottom (:114-116)
//   tag = depthwise KxK 'same' conv of sub   per-channel wavelet_scale folded into the taps (tap-major (K*K, 4C))
//
// and, with flipped taps, one level of the BACKWARD pass: dm = DWT(d r) (the reconstruction's backward), d sub = conv^T(dm).
// Before: a Haar kernel that writes `sub` and a stencil kernel that reads it back through L1/L2 ten times over — two launches of a few
// microseconds each per level, 36 pairs per step at config 2.  Here a workgroup owns an 8 x 8 tile of sub-band pixels and up to 32 input
// channels (= 128 sub-band channels): phase 1 forms the tile plus its K/2 halo in LDS (every sub-band value is computed once per workgroup,
// from coalesced 16-byte loads of the 2 x 2 input blocks) and writes the tile's interior to `sub` (the next level and the backward pass
// need it); phase 2 runs the KxK stencil out of LDS (conflict-free ds_read_b128: 32 lanes = the 128 channels of one pixel) and stores `tag`.
#include "adnm_common.h"

namespace {
constexpr int kBlock = 256, kT = 8;

struct LvArgs {
  const float* x;
  int64_t ldx;       // pixel stride of the input (floats)
  const float* taps; // (K*K, 4C) tap-major
  float* sub;        // (B, h2, w2, 4C)
  float* tag;        // (B, h2, w2, 4C)
  int B, H, W, C, h2, w2, tiles_x, CB;
};

// CX: channel stride of the input (1: a plain (B,H,W,C) tensor; 4: the LL band of the previous level's sub-band tensor); K: stencil size;
// FLIP: correlation with the flipped taps (the transposed conv of the backward pass)
template <int CX, int K, bool FLIP>
__global__ __launch_bounds__(kBlock) void wt_level_kernel(LvArgs a) {
  constexpr int R = K / 2, TP = kT + 2 * R;
  extern __shared__ __attribute__((aligned(16))) float sT[];   // [TP * TP pixels][4 * CB sub-band channels]
  const int CB = a.CB, CB4 = CB >> 2, S = 4 * CB;
  const int tile = blockIdx.x, ty0 = (tile / a.tiles_x) * kT, tx0 = (tile % a.tiles_x) * kT;
  const int c0 = blockIdx.y * CB, b = blockIdx.z;
  const int C4all = 4 * a.C;
  // the stencil's taps of this thread's channel quad: requested first, so that they arrive under phase 1
  const int sq = threadIdx.x % CB, pstep = kBlock / CB;
  float4 wv[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) wv[t] = *reinterpret_cast<const float4*>(a.taps + (int64_t)(FLIP ? K * K - 1 - t : t) * C4all + 4 * (c0 + sq));
  // ---- phase 1: the tile + halo of sub-band pixels; item = (halo pixel, input-channel quad).  All of a thread's items are LOADED before
  // the first is used: one memory round trip per workgroup instead of one per item (the small maps are pure latency)
  constexpr int kIt = (TP * TP * 8 + kBlock - 1) / kBlock;   // items per thread at the widest channel block (CB = 32)
  float v[kIt][4][4];   // [item][pixel a, b, c, d of the 2 x 2 block][channel of the quad]
  const int nitems = TP * TP * CB4;
#pragma unroll
  for (int r = 0; r < kIt; ++r) {
    const int item = threadIdx.x + r * kBlock;
    const int cq = item % CB4, hp = item / CB4, hy = hp / TP, hx = hp - hy * TP;
    const int i = ty0 + hy - R, j = tx0 + hx - R;   // sub-band pixel
    const bool inside = item < nitems && i >= 0 && i < a.h2 && j >= 0 && j < a.w2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int hh = 2 * i + (q >> 1), ww = 2 * j + (q & 1);
      const bool in = inside && hh < a.H && ww < a.W;   // odd sizes: zero-padded bottom / right; outside the map: the conv's zero padding
      const float* p = a.x + (((int64_t)b * a.H + (in ? hh : 0)) * a.W + (in ? ww : 0)) * a.ldx + (int64_t)(c0 + 4 * (in ? cq : 0)) * CX;
      if (CX == 1) {
        const float4 f = in ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[r][q][0] = f.x, v[r][q][1] = f.y, v[r][q][2] = f.z, v[r][q][3] = f.w;
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) v[r][q][m] = in ? p[m * CX] : 0.f;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kIt; ++r) {
    const int item = threadIdx.x + r * kBlock;
    if (item >= nitems) break;
    const int cq = item % CB4, hp = item / CB4, hy = hp / TP, hx = hp - hy * TP;
    const int i = ty0 + hy - R, j = tx0 + hx - R;
    const bool inside = i >= 0 && i < a.h2 && j >= 0 && j < a.w2;
    float* dst = sT + hp * S + 16 * cq;
    float* gsub = a.sub + (((int64_t)b * a.h2 + i) * a.w2 + j) * C4all + 4 * (c0 + 4 * cq);
    const bool interior = inside && hy >= R && hy < R + kT && hx >= R && hx < R + kT;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float p0 = v[r][0][m], p1 = v[r][1][m], p2 = v[r][2][m], p3 = v[r][3][m];
      const float4 o = make_float4(0.5f * (p0 + p1 + p2 + p3), 0.5f * (p0 + p1 - p2 - p3), 0.5f * (p0 - p1 + p2 - p3), 0.5f * (p0 - p1 - p2 + p3));
      *reinterpret_cast<float4*>(dst + 4 * m) = o;
      if (interior) *reinterpret_cast<float4*>(gsub + 4 * m) = o;
    }
  }
  __syncthreads();
  // ---- phase 2: the stencil out of LDS; item = (interior pixel, sub-band channel quad = one input channel); a thread keeps its quad
  for (int pl = threadIdx.x / CB; pl < kT * kT; pl += pstep) {
    const int py = pl / kT, px = pl - py * kT;
    const int i = ty0 + py, j = tx0 + px;
    if (i >= a.h2 || j >= a.w2) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < K; ++dy)
#pragma unroll
      for (int dx = 0; dx < K; ++dx) {
        const float4 s = *reinterpret_cast<const float4*>(sT + ((py + dy) * TP + px + dx) * S + 4 * sq);
        const float4 w = wv[dy * K + dx];
        acc.x = fmaf(w.x, s.x, acc.x), acc.y = fmaf(w.y, s.y, acc.y), acc.z = fmaf(w.z, s.z, acc.z), acc.w = fmaf(w.w, s.w, acc.w);
      }
    *reinterpret_cast<float4*>(a.tag + (((int64_t)b * a.h2 + i) * a.w2 + j) * C4all + 4 * (c0 + sq)) = acc;
  }
}

template <int CX, int K>
int launch_level(const LvArgs& a, bool flip, hipStream_t st) {
  constexpr int TP = kT + 2 * (K / 2);
  const size_t smem = (size_t)TP * TP * 4 * a.CB * sizeof(float);
  const dim3 grid((unsigned)(a.tiles_x * ((a.h2 + kT - 1) / kT)), (unsigned)(a.C / a.CB), (unsigned)a.B);
  if (flip) {
    ADNM_ALLOW_LDS((wt_level_kernel<CX, K, true>), smem, "wt_level");
    wt_level_kernel<CX, K, true><<<grid, kBlock, smem, st>>>(a);
  } else {
    ADNM_ALLOW_LDS((wt_level_kernel<CX, K, false>), smem, "wt_level");
    wt_level_kernel<CX, K, false><<<grid, kBlock, smem, st>>>(a);
  }
  return ADNM_OK;
}
}  // namespace

// x: (B, H, W) pixel rows of stride ldx floats, channel c at column c * cx (cx = 1, or 4 = the LL band of a sub-band tensor);
// taps: (K*K, 4C) tap-major fp32; sub, tag: (B, ceil(H/2), ceil(W/2), 4C) contiguous, both OVERWRITTEN.  flip != 0: flipped taps.
extern "C" int adnm_wt_level(const float* x, int64_t ldx, int64_t cx, const float* taps, float* sub, float* tag, int64_t B, int64_t H, int64_t W,
                             int64_t C, int K, int flip, adnm_stream_t stream) {
  ADNM_REQUIRE(x && taps && sub && tag, "wt_level: null pointer");
  ADNM_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (K == 3 || K == 5) && (cx == 1 || cx == 4), "wt_level: bad arguments (4 | C, K in {3, 5}, cx in {1, 4})");
  ADNM_REQUIRE(ldx >= C * cx && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0, "wt_level: input rows must be 16-byte aligned and hold C * cx columns");
  ADNM_REQUIRE(B * H * W * C < (1ll << 31), "wt_level: tensor too large");
  LvArgs a;
  a.x = x, a.ldx = ldx, a.taps = taps, a.sub = sub, a.tag = tag;
  a.B = (int)B, a.H = (int)H, a.W = (int)W, a.C = (int)C, a.h2 = (int)((H + 1) / 2), a.w2 = (int)((W + 1) / 2);
  a.tiles_x = (a.w2 + kT - 1) / kT;
  // input channels per workgroup: as many as divide C (<= 32: the LDS tile), but on the small maps fewer, so that the grid still has ~256
  // workgroups — 8 x 8 sub-band tiles of a 32 x 32 map with 32-channel blocks are 64 workgroups walking five items each on a 256-CU part
  a.CB = C % 32 == 0 ? 32 : (C % 16 == 0 ? 16 : (C % 8 == 0 ? 8 : 4));
  {
    const int64_t tiles = (int64_t)a.tiles_x * ((a.h2 + kT - 1) / kT) * B;
    while (a.CB > 4 && tiles * (C / a.CB) < 256) a.CB >>= 1;
  }
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("wt_level", st, 4.0 * B * (H * W * C + 2.0 * a.h2 * a.w2 * 4 * C));
  int rc;
  if (cx == 1) rc = K == 5 ? launch_level<1, 5>(a, flip != 0, st) : launch_level<1, 3>(a, flip != 0, st);
  else rc = K == 5 ? launch_level<4, 5>(a, flip != 0, st) : launch_level<4, 3>(a, flip != 0, st);
  if (rc != ADNM_OK) return rc;
  ADNM_CHECK_LAUNCH("wt_level");
  return ADNM_OK;
}
