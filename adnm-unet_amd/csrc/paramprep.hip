// Parameter-side preparation for the fused modules, one launch each way instead of the ~20-60 tiny torch ops
// (index_select, outer products, scale folds, transposes and their autograd) they would otherwise cost per module
// and per step.  These kernels touch PARAMETERS only (kilobytes to a few MB), never activations.
//
//  * ADN-SSD mixer (models/ADNssd.py of the reference): builds, from the reference-layout parameters,
//      w_in   = in_proj.weight with rows reordered to [z | x' | B' | C' | dt]   (replaces the index_select gathers of
//               ADNssd.py:329-341,375-386: even/odd halves become alternating heads, B'/C' = [even | odd])
//      cw     = tap-major effective 3x3 taps of every xBC channel in that order: conv2d taps for even channels,
//               outer(conv_31_*, conv_13_*) for the four asymmetric chains (ADNssd.py:343-346)
//      czw    = tap-major conv2d_z taps; ln_w/ln_b = norm.weight/bias permuted like x';
//      w_out  = alpha1 * out_proj.weight with its y-columns permuted alike (ADNssd.py:459)
//    and the exact transpose of that map for the gradients.
//  * WTConv2d (models/WTConv2d.py): tap-major taps with base_scale / wavelet_scale folded in (WTConv2d.py:123,146),
//    channels zero-padded to a multiple of 4.
#include "adnm_common.h"

namespace {
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;

struct AdnDims {
  int dm, di, gn, P, nh;  // d_model, d_inner, ngroups*d_state, headdim, nheads
  int ldcw, ldcz;         // row strides (floats) of the two tap images cw (9, di + 2 gn) and czw (9, di): their widths, or one common
                          // stride when both live side by side in a (9, 2 di + 2 gn) image [czw | cw] (one stencil launch for z and xBC)
};

// kernel-order xBC channel -> reference xBC channel, and back
template <typename D>
__device__ __forceinline__ int adn_fwd_map(int ch, const D& d) {
  const int half = d.gn >> 1;
  if (ch < d.di) {
    const int h = ch / d.P, p = ch - h * d.P;
    return 2 * ((h >> 1) * d.P + p) + (h & 1);
  }
  int q = ch - d.di, base = d.di;
  if (q >= d.gn) { q -= d.gn; base += d.gn; }
  const int e = q / half, n = q - e * half;
  return base + 2 * n + e;
}
template <typename D>
__device__ __forceinline__ int adn_inv_map(int co, const D& d) {
  const int half = d.gn >> 1;
  if (co < d.di) {
    const int e = co & 1, m = co >> 1, j = m / d.P, p = m - j * d.P;
    return (2 * j + e) * d.P + p;
  }
  int q = co - d.di, base = d.di;
  if (q >= d.gn) { q -= d.gn; base += d.gn; }
  return base + (q & 1) * half + (q >> 1);
}

struct AdnParams {      // reference-layout parameters (or their gradients)
  float *w_in, *conv2d, *c31[4], *c13[4], *conv2d_z, *ln_w, *ln_b, *w_out, *alpha1;  // chains: x1, bc1, x2, bc2
};
struct AdnPrepped {     // kernel-layout tensors (or their gradients)
  float *w_in, *cw, *czw, *ln_w, *ln_b, *w_out;
  // forward only: NARROW copies of the two big matrices for the weight-streaming GEMMs (adnm_skgemm b_dtype) instead of the fp32 ones
  // (ndt = ADNM_B_BF16 / ADNM_B_FP8; NULL = keep fp32).  fp8: w_in_n = e4m3(w_in * *s_in); w_out carries alpha1, so its effective scale
  // *s_out / |alpha1| is published at s_out_eff for the GEMMs that read w_out_n.
  void *w_in_n, *w_out_n;
  const float *s_in, *s_out;
  float* s_out_eff;
  int ndt;
};
__device__ __forceinline__ void adn_store_narrow(void* dst, int64_t off, int ndt, float4 v, float scale) {
  if (ndt == ADNM_B_BF16) {
    Io<uint16_t>::st4(reinterpret_cast<uint16_t*>(dst) + off, v);
  } else {
    const float c0 = __builtin_amdgcn_fmed3f(v.x * scale, 448.f, -448.f), c1 = __builtin_amdgcn_fmed3f(v.y * scale, 448.f, -448.f);
    const float c2 = __builtin_amdgcn_fmed3f(v.z * scale, 448.f, -448.f), c3 = __builtin_amdgcn_fmed3f(v.w * scale, 448.f, -448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c0, c1, w, false), w = __builtin_amdgcn_cvt_pk_fp8_f32(c2, c3, w, true);
    *reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(dst) + off) = w;
  }
}

// chain of reference xBC channel `co` (odd): which (c31,c13) pair and which row
template <typename D>
__device__ __forceinline__ void adn_chain_of(int co, const D& d, int& chain, int& row) {
  const int o = (co - 1) >> 1;          // index in the odd part O
  const int i = o >> 1;                 // index in Oe / Oo
  const int nx = d.di >> 2;
  const int odd = o & 1;                // 0: Oe (x1 / bc1), 1: Oo (x2 / bc2)
  if (i < nx) { chain = odd ? 2 : 0; row = i; }
  else { chain = odd ? 3 : 1; row = i - nx; }
}

// The two big matrices (in_proj.weight: a ROW permutation; out_proj.weight: a column permutation of its first half, scaled by alpha1 —
// 19 + 17 MB at the deepest mixer) move as float4s, one quad per work item; the small tap / norm tensors one element per work item.
// (bid, nblk): this workgroup's index and the workgroup count of ITS mixer — the whole grid for the single-mixer launch, a block range
// of the grouped launch (adnm_adnprep_*_multi: every mixer of a model stage in one launch)
template <typename P, typename O, typename D>
__device__ __forceinline__ void adn_prep_fwd_body(const P& p, const O& o, const D& d, int bid, int nblk) {
  const int cx = d.di + 2 * d.gn, dinp = 2 * d.di + 2 * d.gn + d.nh, dm4 = d.dm >> 2, oq = d.di >> 1;   // oq: quads per out_proj row (2 di / 4)
  const int64_t n0 = (int64_t)dinp * dm4, n1 = n0 + 9 * cx, n2 = n1 + 9 * d.di, n3 = n2 + 2 * d.di, n4 = n3 + (int64_t)d.dm * oq;
  const float a1 = *p.alpha1;
  float sc_in = 1.f, sc_out = 1.f;
  if (o.ndt == ADNM_B_FP8) {
    if (o.w_in_n) sc_in = *o.s_in;
    if (o.w_out_n) {
      sc_out = *o.s_out / fmaxf(fabsf(a1), 1e-30f);
      if (bid == 0 && threadIdx.x == 0) *o.s_out_eff = sc_out;
    }
  }
  for (int64_t i = (int64_t)bid * kBlock + threadIdx.x; i < n4; i += (int64_t)nblk * kBlock) {
    if (i < n0) {
      const int r = (int)(i / dm4), c = (int)(i - (int64_t)r * dm4) * 4;
      int src = r;
      if (r >= d.di && r < d.di + cx) src = d.di + adn_fwd_map(r - d.di, d);
      const float4 v = *reinterpret_cast<const float4*>(p.w_in + (int64_t)src * d.dm + c);
      if (o.w_in_n) adn_store_narrow(o.w_in_n, (int64_t)r * d.dm + c, o.ndt, v, sc_in);
      else *reinterpret_cast<float4*>(o.w_in + (int64_t)r * d.dm + c) = v;
    } else if (i < n1) {
      const int j = (int)(i - n0), t = j / cx, ch = j - t * cx;
      const int co = adn_fwd_map(ch, d);
      float v;
      if ((co & 1) == 0) v = p.conv2d[(co >> 1) * 9 + t];
      else {
        int chain, row;
        adn_chain_of(co, d, chain, row);
        v = p.c31[chain][row * 3 + t / 3] * p.c13[chain][row * 3 + t % 3];
      }
      o.cw[t * d.ldcw + ch] = v;
    } else if (i < n2) {
      const int j = (int)(i - n1), t = j / d.di, c = j - t * d.di;
      o.czw[t * d.ldcz + c] = p.conv2d_z[c * 9 + t];
    } else if (i < n3) {
      const int j = (int)(i - n2);
      if (j < d.di) o.ln_w[j] = p.ln_w[adn_fwd_map(j, d)];
      else o.ln_b[j - d.di] = p.ln_b[adn_fwd_map(j - d.di, d)];
    } else {
      const int64_t j = i - n3;
      const int r = (int)(j / oq), k = (int)(j - (int64_t)r * oq) * 4;
      const float* row = p.w_out + (int64_t)r * 2 * d.di;
      float4 v;
      if (k >= d.di) v = *reinterpret_cast<const float4*>(row + k);
      else v = make_float4(row[adn_fwd_map(k, d)], row[adn_fwd_map(k + 1, d)], row[adn_fwd_map(k + 2, d)], row[adn_fwd_map(k + 3, d)]);
      const float4 av = make_float4(a1 * v.x, a1 * v.y, a1 * v.z, a1 * v.w);
      if (o.w_out_n) adn_store_narrow(o.w_out_n, (int64_t)r * 2 * d.di + k, o.ndt, av, sc_out);
      else *reinterpret_cast<float4*>(o.w_out + (int64_t)r * 2 * d.di + k) = av;
    }
  }
}
__global__ __launch_bounds__(kBlock) void adn_prep_fwd_kernel(AdnParams p, AdnPrepped o, AdnDims d) { adn_prep_fwd_body(p, o, d, blockIdx.x, gridDim.x); }

// thread per reference-parameter element: gathers its gradient from the prepped gradients.  part[blockIdx.x] =
// this block's share of d alpha1 = sum g_w_out * out_proj.weight(permuted).
template <typename P, typename G, typename D>
__device__ __forceinline__ void adn_prep_bwd_body(const P& p, const G& g, const P& dp, const D& d, float* __restrict__ part, int bid, int nblk) {
  __shared__ float sm[kBlock / 64];
  const int cx = d.di + 2 * d.gn, dinp = 2 * d.di + 2 * d.gn + d.nh, ce = cx >> 1, nx = d.di >> 2, nbc = d.gn >> 1;
  const int nchain = 2 * (nx + nbc) * 3;  // per kind (c31 / c13): x1,bc1,x2,bc2 rows x 3 taps
  const int dm4 = d.dm >> 2, oq = d.di >> 1;   // the two big matrices as float4 quads (see adn_prep_fwd_kernel)
  const int64_t n0 = (int64_t)dinp * dm4, n1 = n0 + (int64_t)ce * 9, n2 = n1 + 2 * nchain, n3 = n2 + 9 * d.di, n4 = n3 + 2 * d.di,
                n5 = n4 + (int64_t)d.dm * oq;
  const float a1 = *p.alpha1;
  float acc = 0.f;
  for (int64_t i = (int64_t)bid * kBlock + threadIdx.x; i < n5; i += (int64_t)nblk * kBlock) {
    if (i < n0) {
      const int r = (int)(i / dm4), c = (int)(i - (int64_t)r * dm4) * 4;
      int src = r;
      if (r >= d.di && r < d.di + cx) src = d.di + adn_inv_map(r - d.di, d);
      *reinterpret_cast<float4*>(dp.w_in + (int64_t)r * d.dm + c) = *reinterpret_cast<const float4*>(g.w_in + (int64_t)src * d.dm + c);
    } else if (i < n1) {
      const int j = (int)(i - n0), m = j / 9, t = j - m * 9;
      dp.conv2d[j] = g.cw[t * d.ldcw + adn_inv_map(2 * m, d)];
    } else if (i < n2) {
      int j = (int)(i - n1);
      const int kind = j / nchain;  // 0: conv_31 (3x1, tap a = row), 1: conv_13 (1x3, tap b = column)
      j -= kind * nchain;
      // layout inside a kind: [x1 (nx*3) | bc1 (nbc*3) | x2 (nx*3) | bc2 (nbc*3)]
      int chain, row, tap;
      if (j < nx * 3) { chain = 0; row = j / 3; tap = j % 3; }
      else if (j < (nx + nbc) * 3) { j -= nx * 3; chain = 1; row = j / 3; tap = j % 3; }
      else if (j < (2 * nx + nbc) * 3) { j -= (nx + nbc) * 3; chain = 2; row = j / 3; tap = j % 3; }
      else { j -= (2 * nx + nbc) * 3; chain = 3; row = j / 3; tap = j % 3; }
      const int idx = (chain & 1) ? nx + row : row;          // index in Oe / Oo
      const int o_ = 2 * idx + (chain >> 1);                 // index in O
      const int ch = adn_inv_map(2 * o_ + 1, d);
      float s = 0.f;
      if (kind == 0) {
#pragma unroll
        for (int b = 0; b < 3; ++b) s += g.cw[(tap * 3 + b) * d.ldcw + ch] * p.c13[chain][row * 3 + b];
        dp.c31[chain][row * 3 + tap] = s;
      } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) s += g.cw[(a * 3 + tap) * d.ldcw + ch] * p.c31[chain][row * 3 + a];
        dp.c13[chain][row * 3 + tap] = s;
      }
    } else if (i < n3) {
      const int j = (int)(i - n2), c = j / 9, t = j - c * 9;
      dp.conv2d_z[j] = g.czw[t * d.ldcz + c];
    } else if (i < n4) {
      const int j = (int)(i - n3);
      if (j < d.di) dp.ln_w[j] = g.ln_w[adn_inv_map(j, d)];
      else dp.ln_b[j - d.di] = g.ln_b[adn_inv_map(j - d.di, d)];
    } else {
      const int64_t j = i - n4;
      const int r = (int)(j / oq), k = (int)(j - (int64_t)r * oq) * 4;
      const float* grow = g.w_out + (int64_t)r * 2 * d.di;
      float4 gv;
      if (k >= d.di) gv = *reinterpret_cast<const float4*>(grow + k);
      else gv = make_float4(grow[adn_inv_map(k, d)], grow[adn_inv_map(k + 1, d)], grow[adn_inv_map(k + 2, d)], grow[adn_inv_map(k + 3, d)]);
      const int64_t at = (int64_t)r * 2 * d.di + k;
      const float4 w = *reinterpret_cast<const float4*>(p.w_out + at);
      *reinterpret_cast<float4*>(dp.w_out + at) = make_float4(a1 * gv.x, a1 * gv.y, a1 * gv.z, a1 * gv.w);
      acc = fmaf(gv.x, w.x, fmaf(gv.y, w.y, fmaf(gv.z, w.z, fmaf(gv.w, w.w, acc))));
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[bid] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ __launch_bounds__(kBlock) void adn_prep_bwd_kernel(AdnParams p, AdnPrepped g, AdnParams dp, AdnDims d, float* __restrict__ part) {
  adn_prep_bwd_body(p, g, dp, d, part, blockIdx.x, gridDim.x);
}

// Grouped launches: up to kMaxMulti mixers per launch, descriptors by value in the kernel argument segment (read through the constant
// address space: the descriptor index is a run-time, workgroup-uniform value).
constexpr int kMaxMulti = 8;
struct MultiAdnFwd {
  int count, blk_end[kMaxMulti];
  AdnParams p[kMaxMulti];
  AdnPrepped o[kMaxMulti];
  AdnDims d[kMaxMulti];
};
struct MultiAdnBwd {
  int count, blk_end[kMaxMulti];
  AdnParams p[kMaxMulti];
  AdnPrepped g[kMaxMulti];
  AdnParams dp[kMaxMulti];
  AdnDims d[kMaxMulti];
  float* part[kMaxMulti];
};
#define ADNM_KERNARG(T) (*(const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr())
__global__ __launch_bounds__(kBlock) void adn_prep_fwd_multi_kernel(MultiAdnFwd by_value) {
  (void)by_value;
  const auto& m = ADNM_KERNARG(MultiAdnFwd);
  int k = 0;
  while (k + 1 < m.count && (int)blockIdx.x >= m.blk_end[k]) ++k;
  const int b0 = k ? m.blk_end[k - 1] : 0;
  adn_prep_fwd_body(m.p[k], m.o[k], m.d[k], (int)blockIdx.x - b0, m.blk_end[k] - b0);
}
__global__ __launch_bounds__(kBlock) void adn_prep_bwd_multi_kernel(MultiAdnBwd by_value) {
  (void)by_value;
  const auto& m = ADNM_KERNARG(MultiAdnBwd);
  int k = 0;
  while (k + 1 < m.count && (int)blockIdx.x >= m.blk_end[k]) ++k;
  const int b0 = k ? m.blk_end[k - 1] : 0;
  adn_prep_bwd_body(m.p[k], m.g[k], m.dp[k], m.d[k], m.part[k], (int)blockIdx.x - b0, m.blk_end[k] - b0);
}

// ------------------------------------------------------------------------------------------------ WTConv2d
struct WtPtrs {
  float* w[5];  // [0] base conv, [1..] level convs   (reference layout (Cg, K*K), Cg = C or 4C)
  float* s[5];  // per-channel scales
  float* t[5];  // tap-major outputs (K*K, Cgp)
  float *bias, *bias_t;
};

template <int KK, typename P>
__device__ __forceinline__ void wt_prep_fwd_body(const P& p, int C, int Cp, int levels, int bid, int nblk) {
  // one thread per (group, padded channel): writes its K*K taps
  const int per0 = Cp, perl = 4 * Cp;
  const int total = per0 + levels * perl;
  for (int i = bid * kBlock + threadIdx.x; i < total; i += nblk * kBlock) {
    int g = 0, c = i;
    if (i >= per0) { g = 1 + (i - per0) / perl; c = (i - per0) % perl; }
    const int cg = g == 0 ? C : 4 * C, cgp = g == 0 ? Cp : 4 * Cp;
    const bool live = c < cg;
    const float sc = live ? p.s[g][c] : 0.f;
    float wv[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) wv[t] = live ? p.w[g][c * KK + t] : 0.f;   // all taps in flight, then the strided stores
#pragma unroll
    for (int t = 0; t < KK; ++t) p.t[g][t * cgp + c] = wv[t] * sc;
    if (g == 0 && p.bias_t) p.bias_t[c] = live ? p.bias[c] * sc : 0.f;
  }
}

template <int KK>
__global__ __launch_bounds__(kBlock) void wt_prep_fwd_kernel(WtPtrs p, int C, int Cp, int levels) { wt_prep_fwd_body<KK>(p, C, Cp, levels, blockIdx.x, gridDim.x); }

// g: gradients of the tap-major tensors (same struct, fields t / bias_t); d: gradients of the parameters (w, s, bias)
// KK is a template parameter so that the tap loop unrolls: all K*K gradient / weight loads of a channel are in flight together
// (as a runtime loop of dependent load -> store pairs this took 15 us for ~400 channels)
template <int KK, typename P>
__device__ __forceinline__ void wt_prep_bwd_body(const P& p, const P& g, const P& d, int C, int Cp, int levels, int bid, int nblk) {
  const int per0 = C, perl = 4 * C;
  const int total = per0 + levels * perl;
  for (int i = bid * kBlock + threadIdx.x; i < total; i += nblk * kBlock) {
    int gi = 0, c = i;
    if (i >= per0) { gi = 1 + (i - per0) / perl; c = (i - per0) % perl; }
    const int cgp = gi == 0 ? Cp : 4 * Cp;
    const float sc = p.s[gi][c];
    float ds = 0.f;
    float gv[KK], wv[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      gv[t] = g.t[gi][t * cgp + c];
      wv[t] = p.w[gi][c * KK + t];
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      d.w[gi][c * KK + t] = gv[t] * sc;
      ds = fmaf(gv[t], wv[t], ds);
    }
    if (gi == 0 && p.bias) {
      const float gb = g.bias_t[c];
      d.bias[c] = gb * sc;
      ds = fmaf(gb, p.bias[c], ds);
    }
    d.s[gi][c] = ds;
  }
}

template <int KK>
__global__ __launch_bounds__(kBlock) void wt_prep_bwd_kernel(WtPtrs p, WtPtrs g, WtPtrs d, int C, int Cp, int levels) {
  wt_prep_bwd_body<KK>(p, g, d, C, Cp, levels, blockIdx.x, gridDim.x);
}

// grouped launches of the WTConv2d preparation (every WTConv2d of a model stage in one launch); K is per descriptor (3 or 5)
struct WtDims {
  int C, Cp, K, levels;
};
struct MultiWtFwd {
  int count, blk_end[kMaxMulti];
  WtPtrs p[kMaxMulti];
  WtDims d[kMaxMulti];
};
struct MultiWtBwd {
  int count, blk_end[kMaxMulti];
  WtPtrs p[kMaxMulti], g[kMaxMulti], dp[kMaxMulti];
  WtDims d[kMaxMulti];
};
__global__ __launch_bounds__(kBlock) void wt_prep_fwd_multi_kernel(MultiWtFwd by_value) {
  (void)by_value;
  const auto& m = ADNM_KERNARG(MultiWtFwd);
  int k = 0;
  while (k + 1 < m.count && (int)blockIdx.x >= m.blk_end[k]) ++k;
  const int b0 = k ? m.blk_end[k - 1] : 0, bid = (int)blockIdx.x - b0, nblk = m.blk_end[k] - b0;
  if (m.d[k].K == 3) wt_prep_fwd_body<9>(m.p[k], m.d[k].C, m.d[k].Cp, m.d[k].levels, bid, nblk);
  else wt_prep_fwd_body<25>(m.p[k], m.d[k].C, m.d[k].Cp, m.d[k].levels, bid, nblk);
}
__global__ __launch_bounds__(kBlock) void wt_prep_bwd_multi_kernel(MultiWtBwd by_value) {
  (void)by_value;
  const auto& m = ADNM_KERNARG(MultiWtBwd);
  int k = 0;
  while (k + 1 < m.count && (int)blockIdx.x >= m.blk_end[k]) ++k;
  const int b0 = k ? m.blk_end[k - 1] : 0, bid = (int)blockIdx.x - b0, nblk = m.blk_end[k] - b0;
  if (m.d[k].K == 3) wt_prep_bwd_body<9>(m.p[k], m.g[k], m.dp[k], m.d[k].C, m.d[k].Cp, m.d[k].levels, bid, nblk);
  else wt_prep_bwd_body<25>(m.p[k], m.g[k], m.dp[k], m.d[k].C, m.d[k].Cp, m.d[k].levels, bid, nblk);
}

int adn_check(const char* who, AdnDims d) {
  ADNM_REQUIRE(d.dm > 0 && d.dm % 4 == 0 && d.di > 0 && d.di % 8 == 0 && d.gn > 0 && d.gn % 4 == 0 && d.P > 0 && d.di % d.P == 0 && d.nh == d.di / d.P && d.nh % 2 == 0,
               "%s: unsupported ADN-SSD dimensions dm=%d di=%d gn=%d P=%d nh=%d", who, d.dm, d.di, d.gn, d.P, d.nh);
  return ADNM_OK;
}

unsigned grid_for(int64_t n) {
  int64_t g = adnm_cdiv(n, kBlock);
  return (unsigned)(g < 1 ? 1 : (g > kMaxBlocks ? kMaxBlocks : g));
}

}  // namespace

// pointer tables are passed as plain arrays to keep the ABI free of structs:
//   params[17]  = {in_proj.weight, conv2d.weight, conv_31_x1, conv_31_bc1, conv_31_x2, conv_31_bc2, conv_13_x1, conv_13_bc1,
//                  conv_13_x2, conv_13_bc2, conv2d_z.weight, norm.weight, norm.bias, out_proj.weight, alpha1, 0, 0}
//   prepped[6]  = {w_in, cw, czw, ln_w, ln_b, w_out}
static AdnParams adn_params(float* const* a) {
  AdnParams p;
  p.w_in = a[0]; p.conv2d = a[1];
  for (int k = 0; k < 4; ++k) { p.c31[k] = a[2 + k]; p.c13[k] = a[6 + k]; }
  p.conv2d_z = a[10]; p.ln_w = a[11]; p.ln_b = a[12]; p.w_out = a[13]; p.alpha1 = a[14];
  return p;
}
static AdnPrepped adn_prepped(float* const* a) {
  AdnPrepped o;
  o.w_in = a[0]; o.cw = a[1]; o.czw = a[2]; o.ln_w = a[3]; o.ln_b = a[4]; o.w_out = a[5];
  o.w_in_n = o.w_out_n = nullptr, o.s_in = o.s_out = nullptr, o.s_out_eff = nullptr, o.ndt = ADNM_B_F32;
  return o;
}

extern "C" int adnm_adnprep_fwd(float* const* params, float* const* prepped, int64_t d_model, int64_t d_inner, int64_t gn, int64_t headdim,
                                int64_t tap_ld, adnm_stream_t stream) {
  ADNM_REQUIRE(params && prepped, "adnprep_fwd: null table");
  for (int k = 0; k < 15; ++k) ADNM_REQUIRE(params[k], "adnprep_fwd: params[%d] is null", k);
  for (int k = 0; k < 6; ++k) ADNM_REQUIRE(prepped[k], "adnprep_fwd: prepped[%d] is null", k);
  ADNM_REQUIRE(tap_ld == 0 || tap_ld >= d_inner + 2 * gn, "adnprep: tap_ld %lld < the width of the tap image", (long long)tap_ld);
  AdnDims d{(int)d_model, (int)d_inner, (int)gn, (int)headdim, (int)(d_inner / (headdim > 0 ? headdim : 1)),
            (int)(tap_ld ? tap_ld : d_inner + 2 * gn), (int)(tap_ld ? tap_ld : d_inner)};
  if (int rc = adn_check("adnprep_fwd", d)) return rc;
  const int64_t n = (int64_t)(2 * d.di + 2 * d.gn + d.nh) * d.dm + 9 * (d.di + 2 * d.gn) + 9 * d.di + 2 * d.di + (int64_t)d.dm * 2 * d.di;
  hipStream_t st = (hipStream_t)stream;
  { ADNM_PROF("adn_prep_fwd", st, 8.0 * n); adn_prep_fwd_kernel<<<grid_for(n), kBlock, 0, st>>>(adn_params(params), adn_prepped(prepped), d); }
  ADNM_CHECK_LAUNCH("adnprep_fwd");
  return ADNM_OK;
}

extern "C" int64_t adnm_adnprep_bwd_ws_bytes(void) { return kMaxBlocks * (int64_t)sizeof(float); }

extern "C" int adnm_adnprep_bwd(float* const* params, float* const* gprepped, float* const* dparams, int64_t d_model, int64_t d_inner,
                                int64_t gn, int64_t headdim, int64_t tap_ld, void* ws, int64_t ws_bytes, adnm_stream_t stream) {
  ADNM_REQUIRE(params && gprepped && dparams, "adnprep_bwd: null table");
  for (int k = 0; k < 15; ++k) ADNM_REQUIRE(params[k] && dparams[k], "adnprep_bwd: params/dparams[%d] is null", k);
  for (int k = 0; k < 6; ++k) ADNM_REQUIRE(gprepped[k], "adnprep_bwd: gprepped[%d] is null", k);
  if (!ws || ws_bytes < adnm_adnprep_bwd_ws_bytes()) {
    adnm_set_error("adnprep_bwd: workspace too small");
    return ADNM_EWORKSPACE;
  }
  ADNM_REQUIRE(tap_ld == 0 || tap_ld >= d_inner + 2 * gn, "adnprep: tap_ld %lld < the width of the tap image", (long long)tap_ld);
  AdnDims d{(int)d_model, (int)d_inner, (int)gn, (int)headdim, (int)(d_inner / (headdim > 0 ? headdim : 1)),
            (int)(tap_ld ? tap_ld : d_inner + 2 * gn), (int)(tap_ld ? tap_ld : d_inner)};
  if (int rc = adn_check("adnprep_bwd", d)) return rc;
  const int cx = d.di + 2 * d.gn;
  const int64_t n = (int64_t)(2 * d.di + 2 * d.gn + d.nh) * d.dm + (int64_t)(cx / 2) * 9 + 2 * 2 * (d.di / 4 + d.gn / 2) * 3 + 9 * d.di + 2 * d.di +
                    (int64_t)d.dm * 2 * d.di;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(n);
  { ADNM_PROF("adn_prep_bwd", st, 8.0 * n); adn_prep_bwd_kernel<<<grid, kBlock, 0, st>>>(adn_params(params), adn_prepped(gprepped), adn_params(dparams), d, (float*)ws); }
  adnm_launch_fold("adn_prep_bwd_fold", (const float*)ws, (int)grid, 1, {dparams[14], 1}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("adnprep_bwd");
  return ADNM_OK;
}

// WTConv2d tables (levels <= 4): w[1+levels], s[1+levels], t[1+levels]; bias / bias_t may be NULL together.
static WtPtrs wt_ptrs(float* const* w, float* const* s, float* const* t, float* bias, float* bias_t, int levels) {
  WtPtrs p;
  for (int k = 0; k < 5; ++k) {
    p.w[k] = (w && k <= levels) ? w[k] : nullptr;
    p.s[k] = (s && k <= levels) ? s[k] : nullptr;
    p.t[k] = (t && k <= levels) ? t[k] : nullptr;
  }
  p.bias = bias;
  p.bias_t = bias_t;
  return p;
}

extern "C" int adnm_wtprep_fwd(float* const* w, float* const* s, float* bias, float* const* taps, float* bias_t, int64_t C, int64_t Cp,
                               int64_t K, int64_t levels, adnm_stream_t stream) {
  ADNM_REQUIRE(w && s && taps && C > 0 && Cp >= C && Cp % 4 == 0 && levels >= 0 && levels <= 4 && (K == 3 || K == 5) && (!bias == !bias_t),
               "wtprep_fwd: bad arguments (C=%lld Cp=%lld K=%lld levels=%lld)", (long long)C, (long long)Cp, (long long)K, (long long)levels);
  for (int k = 0; k <= levels; ++k) ADNM_REQUIRE(w[k] && s[k] && taps[k], "wtprep_fwd: table entry %d is null", k);
  hipStream_t st = (hipStream_t)stream;
  const int total = (int)(Cp + levels * 4 * Cp);
  {
    ADNM_PROF("wt_prep_fwd", st, 8.0 * total * K * K);
    const WtPtrs pp = wt_ptrs(w, s, taps, bias, bias_t, (int)levels);
    if (K == 3) wt_prep_fwd_kernel<9><<<grid_for(total), kBlock, 0, st>>>(pp, (int)C, (int)Cp, (int)levels);
    else wt_prep_fwd_kernel<25><<<grid_for(total), kBlock, 0, st>>>(pp, (int)C, (int)Cp, (int)levels);
  }
  ADNM_CHECK_LAUNCH("wtprep_fwd");
  return ADNM_OK;
}

extern "C" int adnm_wtprep_bwd(float* const* w, float* const* s, float* bias, float* const* gtaps, float* gbias_t, float* const* dw,
                               float* const* ds, float* dbias, int64_t C, int64_t Cp, int64_t K, int64_t levels, adnm_stream_t stream) {
  ADNM_REQUIRE(w && s && gtaps && dw && ds && C > 0 && Cp >= C && levels >= 0 && levels <= 4 && (K == 3 || K == 5), "wtprep_bwd: bad arguments");
  ADNM_REQUIRE((!bias == !gbias_t) && (!bias == !dbias), "wtprep_bwd: bias pointers must be all set or all null");
  for (int k = 0; k <= levels; ++k) ADNM_REQUIRE(w[k] && s[k] && gtaps[k] && dw[k] && ds[k], "wtprep_bwd: table entry %d is null", k);
  hipStream_t st = (hipStream_t)stream;
  const int total = (int)(C + levels * 4 * C);
  {
    ADNM_PROF("wt_prep_bwd", st, 12.0 * total * K * K);
    const WtPtrs pp = wt_ptrs(w, s, nullptr, bias, nullptr, (int)levels), gp = wt_ptrs(nullptr, nullptr, gtaps, nullptr, gbias_t, (int)levels),
                 dp = wt_ptrs(dw, ds, nullptr, dbias, nullptr, (int)levels);
    if (K == 3) wt_prep_bwd_kernel<9><<<grid_for(total), kBlock, 0, st>>>(pp, gp, dp, (int)C, (int)Cp, (int)levels);
    else wt_prep_bwd_kernel<25><<<grid_for(total), kBlock, 0, st>>>(pp, gp, dp, (int)C, (int)Cp, (int)levels);
  }
  ADNM_CHECK_LAUNCH("wtprep_bwd");
  return ADNM_OK;
}


// ---- grouped forms: every ADN-SSD mixer / every WTConv2d of a model stage in ONE launch each way (the per-module launches above are a
// few microseconds of work each: 36 launches per step at config 2).  Tables are concatenated per module: params[15 * i ..], prepped[6 * i ..],
// dims[5 * i ..] = {d_model, d_inner, gn, headdim, tap_ld}; at most 8 modules per launch (longer lists are cut into several launches).
namespace {
AdnDims adn_dims(const int64_t* v) {
  const int64_t tap_ld = v[4];
  return AdnDims{(int)v[0], (int)v[1], (int)v[2], (int)v[3], (int)(v[1] / (v[3] > 0 ? v[3] : 1)), (int)(tap_ld ? tap_ld : v[1] + 2 * v[2]),
                 (int)(tap_ld ? tap_ld : v[1])};
}
int64_t adn_items_fwd(const AdnDims& d) {
  return (int64_t)(2 * d.di + 2 * d.gn + d.nh) * d.dm + 9 * (d.di + 2 * d.gn) + 9 * d.di + 2 * d.di + (int64_t)d.dm * 2 * d.di;
}
int64_t adn_items_bwd(const AdnDims& d) {
  const int cx = d.di + 2 * d.gn;
  return (int64_t)(2 * d.di + 2 * d.gn + d.nh) * d.dm + (int64_t)(cx / 2) * 9 + 2 * 2 * (d.di / 4 + d.gn / 2) * 3 + 9 * d.di + 2 * d.di + (int64_t)d.dm * 2 * d.di;
}
// workgroups of one module inside a grouped launch: enough to stream its matrices, capped so that 8 modules stay a sane grid
unsigned multi_blocks(int64_t items) {
  int64_t g = adnm_cdiv(items / 4 + 1, kBlock);   // the big matrices move as float4 quads
  return (unsigned)(g < 1 ? 1 : (g > 512 ? 512 : g));
}
}  // namespace

extern "C" int adnm_adnprep_fwd_multi(int64_t n, float* const* params, float* const* prepped, const int64_t* dims, float* const* narrow,
                                      int narrow_dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(n >= 1 && params && prepped && dims, "adnprep_fwd_multi: bad arguments");
  ADNM_REQUIRE(!narrow || narrow_dtype == ADNM_B_BF16 || narrow_dtype == ADNM_B_FP8, "adnprep_fwd_multi: narrow copies are bf16 (1) or scaled e4m3 (2)");
  hipStream_t st = (hipStream_t)stream;
  for (int64_t i0 = 0; i0 < n; i0 += kMaxMulti) {
    MultiAdnFwd m;
    m.count = (int)(n - i0 < kMaxMulti ? n - i0 : kMaxMulti);
    int blocks = 0;
    double bytes = 0;
    for (int k = 0; k < m.count; ++k) {
      const int64_t i = i0 + k;
      for (int j = 0; j < 15; ++j) ADNM_REQUIRE(params[15 * i + j], "adnprep_fwd_multi: params[%lld][%d] is null", (long long)i, j);
      m.p[k] = adn_params(params + 15 * i), m.o[k] = adn_prepped(prepped + 6 * i), m.d[k] = adn_dims(dims + 5 * i);
      if (narrow) {   // narrow[5 i ..] = {w_in_n, w_out_n, s_in, s_out, s_out_eff}: a narrow copy replaces the fp32 matrix of the same name
        float* const* t = narrow + 5 * i;
        m.o[k].w_in_n = t[0], m.o[k].w_out_n = t[1], m.o[k].s_in = t[2], m.o[k].s_out = t[3], m.o[k].s_out_eff = t[4], m.o[k].ndt = narrow_dtype;
        ADNM_REQUIRE(narrow_dtype != ADNM_B_FP8 || ((!t[0] || t[2]) && (!t[1] || (t[3] && t[4]))), "adnprep_fwd_multi: an fp8 copy needs its scale pointers");
      }
      for (int j = 0; j < 6; ++j)
        ADNM_REQUIRE(prepped[6 * i + j] || (j == 0 && m.o[k].w_in_n) || (j == 5 && m.o[k].w_out_n), "adnprep_fwd_multi: prepped[%lld][%d] is null", (long long)i, j);
      if (int rc = adn_check("adnprep_fwd_multi", m.d[k])) return rc;
      blocks += (int)multi_blocks(adn_items_fwd(m.d[k]));
      m.blk_end[k] = blocks;
      bytes += 8.0 * adn_items_fwd(m.d[k]);
    }
    for (int k = m.count; k < kMaxMulti; ++k) m.blk_end[k] = blocks;
    ADNM_PROF("adn_prep_fwd", st, bytes);
    adn_prep_fwd_multi_kernel<<<(unsigned)blocks, kBlock, 0, st>>>(m);
  }
  ADNM_CHECK_LAUNCH("adnprep_fwd_multi");
  return ADNM_OK;
}

extern "C" int64_t adnm_adnprep_bwd_multi_ws_bytes(int64_t n) { return n * 512 * (int64_t)sizeof(float); }

extern "C" int adnm_adnprep_bwd_multi(int64_t n, float* const* params, float* const* gprepped, float* const* dparams, const int64_t* dims,
                                      void* ws, int64_t ws_bytes, adnm_stream_t stream) {
  ADNM_REQUIRE(n >= 1 && params && gprepped && dparams && dims, "adnprep_bwd_multi: bad arguments");
  if (!ws || ws_bytes < adnm_adnprep_bwd_multi_ws_bytes(n)) {
    adnm_set_error("adnprep_bwd_multi: workspace too small");
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  for (int64_t i0 = 0; i0 < n; i0 += kMaxMulti) {
    MultiAdnBwd m;
    m.count = (int)(n - i0 < kMaxMulti ? n - i0 : kMaxMulti);
    int blocks = 0, nb[kMaxMulti];
    double bytes = 0;
    for (int k = 0; k < m.count; ++k) {
      const int64_t i = i0 + k;
      for (int j = 0; j < 15; ++j) ADNM_REQUIRE(params[15 * i + j] && dparams[15 * i + j], "adnprep_bwd_multi: params/dparams[%lld][%d] is null", (long long)i, j);
      for (int j = 0; j < 6; ++j) ADNM_REQUIRE(gprepped[6 * i + j], "adnprep_bwd_multi: gprepped[%lld][%d] is null", (long long)i, j);
      m.p[k] = adn_params(params + 15 * i), m.g[k] = adn_prepped(gprepped + 6 * i), m.dp[k] = adn_params(dparams + 15 * i), m.d[k] = adn_dims(dims + 5 * i);
      if (int rc = adn_check("adnprep_bwd_multi", m.d[k])) return rc;
      m.part[k] = (float*)ws + 512 * i;
      nb[k] = (int)multi_blocks(adn_items_bwd(m.d[k]));
      blocks += nb[k];
      m.blk_end[k] = blocks;
      bytes += 8.0 * adn_items_bwd(m.d[k]);
    }
    for (int k = m.count; k < kMaxMulti; ++k) m.blk_end[k] = blocks;
    {
      ADNM_PROF("adn_prep_bwd", st, bytes);
      adn_prep_bwd_multi_kernel<<<(unsigned)blocks, kBlock, 0, st>>>(m);
    }
    for (int k = 0; k < m.count; ++k)
      adnm_launch_fold("adn_prep_bwd_fold", m.part[k], nb[k], 1, {dparams[15 * (i0 + k) + 14], 1}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  }
  ADNM_CHECK_LAUNCH("adnprep_bwd_multi");
  return ADNM_OK;
}

// WTConv2d, grouped.  Tables per module i: w[5 i ..], s[5 i ..], taps[5 i ..] (entries 0 .. levels used), bias[i], bias_t[i] (both NULL or both
// set), dims[4 i ..] = {C, Cp, K, levels}.
extern "C" int adnm_wtprep_fwd_multi(int64_t n, float* const* w, float* const* s, float* const* bias, float* const* taps, float* const* bias_t,
                                     const int64_t* dims, adnm_stream_t stream) {
  ADNM_REQUIRE(n >= 1 && w && s && bias && taps && bias_t && dims, "wtprep_fwd_multi: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int64_t i0 = 0; i0 < n; i0 += kMaxMulti) {
    MultiWtFwd m;
    m.count = (int)(n - i0 < kMaxMulti ? n - i0 : kMaxMulti);
    int blocks = 0;
    double bytes = 0;
    for (int k = 0; k < m.count; ++k) {
      const int64_t i = i0 + k, C = dims[4 * i], Cp = dims[4 * i + 1], K = dims[4 * i + 2], levels = dims[4 * i + 3];
      ADNM_REQUIRE(C > 0 && Cp >= C && Cp % 4 == 0 && levels >= 0 && levels <= 4 && (K == 3 || K == 5) && (!bias[i] == !bias_t[i]),
                   "wtprep_fwd_multi: bad arguments for module %lld (C=%lld Cp=%lld K=%lld levels=%lld)", (long long)i, (long long)C, (long long)Cp,
                   (long long)K, (long long)levels);
      for (int j = 0; j <= levels; ++j) ADNM_REQUIRE(w[5 * i + j] && s[5 * i + j] && taps[5 * i + j], "wtprep_fwd_multi: table entry %lld/%d is null", (long long)i, j);
      m.p[k] = wt_ptrs(w + 5 * i, s + 5 * i, taps + 5 * i, bias[i], bias_t[i], (int)levels);
      m.d[k] = WtDims{(int)C, (int)Cp, (int)K, (int)levels};
      const int total = (int)(Cp + levels * 4 * Cp);
      blocks += (int)grid_for(total);
      m.blk_end[k] = blocks;
      bytes += 8.0 * total * K * K;
    }
    for (int k = m.count; k < kMaxMulti; ++k) m.blk_end[k] = blocks;
    ADNM_PROF("wt_prep_fwd", st, bytes);
    wt_prep_fwd_multi_kernel<<<(unsigned)blocks, kBlock, 0, st>>>(m);
  }
  ADNM_CHECK_LAUNCH("wtprep_fwd_multi");
  return ADNM_OK;
}

extern "C" int adnm_wtprep_bwd_multi(int64_t n, float* const* w, float* const* s, float* const* bias, float* const* gtaps, float* const* gbias_t,
                                     float* const* dw, float* const* ds, float* const* dbias, const int64_t* dims, adnm_stream_t stream) {
  ADNM_REQUIRE(n >= 1 && w && s && bias && gtaps && gbias_t && dw && ds && dbias && dims, "wtprep_bwd_multi: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int64_t i0 = 0; i0 < n; i0 += kMaxMulti) {
    MultiWtBwd m;
    m.count = (int)(n - i0 < kMaxMulti ? n - i0 : kMaxMulti);
    int blocks = 0;
    double bytes = 0;
    for (int k = 0; k < m.count; ++k) {
      const int64_t i = i0 + k, C = dims[4 * i], Cp = dims[4 * i + 1], K = dims[4 * i + 2], levels = dims[4 * i + 3];
      ADNM_REQUIRE(C > 0 && Cp >= C && levels >= 0 && levels <= 4 && (K == 3 || K == 5), "wtprep_bwd_multi: bad arguments for module %lld", (long long)i);
      ADNM_REQUIRE((!bias[i] == !gbias_t[i]) && (!bias[i] == !dbias[i]), "wtprep_bwd_multi: bias pointers of module %lld must be all set or all null", (long long)i);
      for (int j = 0; j <= levels; ++j)
        ADNM_REQUIRE(w[5 * i + j] && s[5 * i + j] && gtaps[5 * i + j] && dw[5 * i + j] && ds[5 * i + j], "wtprep_bwd_multi: table entry %lld/%d is null", (long long)i, j);
      m.p[k] = wt_ptrs(w + 5 * i, s + 5 * i, nullptr, bias[i], nullptr, (int)levels);
      m.g[k] = wt_ptrs(nullptr, nullptr, gtaps + 5 * i, nullptr, gbias_t[i], (int)levels);
      m.dp[k] = wt_ptrs(dw + 5 * i, ds + 5 * i, nullptr, dbias[i], nullptr, (int)levels);
      m.d[k] = WtDims{(int)C, (int)Cp, (int)K, (int)levels};
      const int total = (int)(C + levels * 4 * C);
      blocks += (int)grid_for(total);
      m.blk_end[k] = blocks;
      bytes += 12.0 * total * K * K;
    }
    for (int k = m.count; k < kMaxMulti; ++k) m.blk_end[k] = blocks;
    ADNM_PROF("wt_prep_bwd", st, bytes);
    wt_prep_bwd_multi_kernel<<<(unsigned)blocks, kBlock, 0, st>>>(m);
  }
  ADNM_CHECK_LAUNCH("wtprep_bwd_multi");
  return ADNM_OK;
}
