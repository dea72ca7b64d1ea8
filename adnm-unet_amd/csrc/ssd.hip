// K1 — the SSD "linear-attention duality" reduction that every Mamba2 mixer of ADNM-UNet executes
// (non_casual_linear_attn: ADNssd.py:252-299 single-group branch :278-284, Vssd.py:161-208 grouped :194-206),
// with softplus(dt + dt_bias) (ADNssd.py:318) fused in and, where a token row fits one head block, the mixer's
// LayerNorm(d_inner) (ADNssd.py:456) as the epilogue of pass 2:
//
//   w[b,l,h]    = softplus(dt_raw + dt_bias[h]) * exp(A_log[h])
//   KV[b,h,n,p] = sum_l Bm[b,l,g,n] * x[b,l,h,p] * w[b,l,h]               (pass 1: global reduction over L)
//   y[b,l,h,p]  = sum_n Cm[b,l,g,n] * KV[b,h,n,p] + D[h] * x[b,l,h,p]      (pass 2: streaming)      g = h % G
//
// There is no recurrence on this branch, so there is nothing to scan: both passes (and the backward pass) are HBM-bound
// streams over the token rows whose arithmetic is three small dense contractions — exactly GEMM-shaped, so they run on
// the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, an fmaf chain) and the vector ALU only applies w / D:
//
//   pass 1   KV_g[N x cols]  += Bm_g^T[N x 16 tok] . (x.w)[16 tok x cols]      (the reduction dimension is the token)
//   pass 2   y[16 tok x cols] = Cm_g[16 tok x N]   . KV_g[N x cols]            (+ D x)
//   backward t = Bm_g . dKV_g,  dCm_g = dy_g . KV_g^T,  dBm_g = (x.w)_g . dKV_g^T   (+ the per-token scalar chain to dt)
//
// Geometry: a workgroup = 4 waves on one batch item and one HEAD BLOCK = 64 consecutive x columns (16 heads of 4, or 8
// of 8) = four 16-column MFMA blocks, each belonging to one B/C group.  A wave walks 16-token TILES: the tile's rows are
// read from HBM with coalesced 16-byte loads (a whole 256-byte x row per 16 lanes), staged in the wave's own LDS slab
// (no workgroup barrier in the loop), and the MFMA operands are read back from there in fragment order; results go the
// same way back (LDS -> coalesced row stores).  KV / dKV of the head block live in registers as MFMA B operands for
// the whole workgroup lifetime.  Cross-workgroup sums (pass 1 over token chunks; the per-head statistics of backward)
// go through fp32 partials in the caller's workspace and a small deterministic fold — never atomics, so a step is
// bitwise reproducible.  Deep maps (L <= 512) take one workgroup per (batch, head block): no partials, no fold.
#include "adnm_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kTile = 16;         // tokens per wave tile = one MFMA block edge
constexpr int kCols = 64;         // x columns per head block
constexpr int kSX = kCols + 4;    // LDS row stride of a 64-column tile (+ one 16-byte access: conflict-free fragment reads)

// column (inside the 64-column head block) of MFMA column j of column block c.  Column blocks are grouped by B/C
// group: blocks [g*4/G, (g+1)*4/G) hold the heads hl = g + G*m of group g, head-major (u = m*P + p).
template <int P, int G>
__device__ __forceinline__ int memcol(int c, int j) {
  constexpr int CPG = 4 / G;
  const int g = c / CPG, u = 16 * (c % CPG) + j;
  return (g + G * (u / P)) * P + (u % P);
}
// column of element u (0 .. 64/G-1) of group g in the same order
template <int P, int G>
__device__ __forceinline__ int groupcol(int g, int u) {
  return (g + G * (u / P)) * P + (u % P);
}

struct Geo {
  int nhb;      // head blocks (grid.y)
  int tok;      // tokens per workgroup
  int nchunk;   // workgroups along L
};
// tiles per wave: enough waves to fill the chip (>= 2048), at most 8 tiles per wave; small maps: one workgroup per (b, head block)
inline Geo make_geo(int64_t B, int64_t L, int64_t H, int64_t P, bool reduction, bool apply = false) {
  Geo g;
  g.nhb = (int)adnm_cdiv(H * P, kCols);
  const int64_t tiles = adnm_cdiv(L, kTile);
  if (reduction && tiles <= 32) {   // no partials, no fold launch
    g.tok = (int)(tiles * kTile);
    g.nchunk = 1;
    return g;
  }
  // target wave count (ADNM_SSD_WAVES_* : measurement aids).  Measured at the refiner shape (tools/kbench_ssd.py, forward pair / backward
  // pair per call): the apply pass is fastest with 4096 waves = one 16-token tile per wave, every tile's loads in flight at once
  // (30.1 -> 28.5 us); the backward pass (236 VGPRs, 2 waves per SIMD) and the reduction pass (partials + fold grow with the
  // workgroup count) with 2048 (41.8 vs 46.3 us; 30.1 vs 33.1 us)
  static const int64_t waves_a = [] { const char* e = getenv("ADNM_SSD_WAVES_APPLY"); const int v = e ? atoi(e) : 0; return (int64_t)(v > 0 ? v : 4096); }();
  static const int64_t waves_b = [] { const char* e = getenv("ADNM_SSD_WAVES_BWD"); const int v = e ? atoi(e) : 0; return (int64_t)(v > 0 ? v : 2048); }();
  static const int64_t waves_r = [] { const char* e = getenv("ADNM_SSD_WAVES_RED"); const int v = e ? atoi(e) : 0; return (int64_t)(v > 0 ? v : 2048); }();
  int64_t tpw = (B * g.nhb * tiles) / (reduction ? waves_r : (apply ? waves_a : waves_b));
  tpw = tpw < 1 ? 1 : (tpw > 8 ? 8 : tpw);
  g.tok = (int)(kWaves * tpw * kTile);
  g.nchunk = (int)adnm_cdiv(L, g.tok);
  return g;
}

// ---- coalesced tile movers.  A 64-column tile: lane f = lane + 64 r (r < 4) owns columns 4q .. 4q+3 (q = lane & 15) of
// token (lane >> 4) + 4 r: 16 lanes = one 256-byte row.  A KW-column tile (Bm / Cm rows): f = lane + 64 r over 16 * KW/4.
template <typename T>
__device__ __forceinline__ float4 ld_row4(const T* base, int64_t row, int64_t ld, int col, bool ok) {
  return ok ? Io<T>::ld4(base + row * ld + col) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void lds_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 lds_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <typename T, int KW>
__device__ __forceinline__ void stage_k_tile(const T* __restrict__ K, int64_t ldk, int64_t row0, int nvalid, float* sK, int lane) {
  constexpr int SK = KW + 4, Q = KW / 4, TOT = kTile * Q, IT = (TOT + 63) / 64;
  float4 v[IT];
#pragma unroll
  for (int r = 0; r < IT; ++r) {
    const int f = lane + 64 * r, t = f / Q, qq = f % Q;
    v[r] = ld_row4<T>(K, row0 + t, ldk, 4 * qq, f < TOT && t < nvalid);
  }
#pragma unroll
  for (int r = 0; r < IT; ++r) {
    const int f = lane + 64 * r, t = f / Q, qq = f % Q;
    if (f < TOT) lds_st4(sK + t * SK + 4 * qq, v[r]);
  }
}

// the same in two halves: issue the loads of a tile (registers), write them to the wave's slab later — the next tile's loads fly under
// the current tile's MFMAs
template <typename T, int KW>
struct KTile {
  static constexpr int SK = KW + 4, Q = KW / 4, TOT = kTile * Q, IT = (TOT + 63) / 64;
  float4 v[IT];
  __device__ __forceinline__ void load(const T* __restrict__ K, int64_t ldk, int64_t row0, int nvalid, int lane) {
#pragma unroll
    for (int r = 0; r < IT; ++r) {
      const int f = lane + 64 * r, t = f / Q, qq = f % Q;
      v[r] = ld_row4<T>(K, row0 + t, ldk, 4 * qq, f < TOT && t < nvalid);
    }
  }
  __device__ __forceinline__ void store(float* sK, int lane) const {
#pragma unroll
    for (int r = 0; r < IT; ++r) {
      const int f = lane + 64 * r, t = f / Q, qq = f % Q;
      if (f < TOT) lds_st4(sK + t * SK + 4 * qq, v[r]);
    }
  }
};

// ================================================================================================ pass 1
// out[b, chunk, h, n, p] = sum_{l in chunk} K[b,l,g,n] * V[b,l,h,p] * (WEIGHTED ? w[b,l,h] : 1)
// used for KV (K = Bm, V = x, weighted) and for dKV in backward (K = Cm, V = dy, unweighted).
template <typename T, int P, int N, int G, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void ssd_kv_kernel(const T* __restrict__ V, int64_t ldv, const T* __restrict__ K, int64_t ldk,
                                                        const T* __restrict__ dt_raw, int64_t lddt, int64_t dt_hs,
                                                        const float* __restrict__ dt_bias, const float* __restrict__ A_log, int64_t p_hs,
                                                        float* __restrict__ out, int64_t out_bstride, int64_t out_cstride, int64_t L, int H,
                                                        int tok_per_block) {
  constexpr int KW = G * N, SK = KW + 4, HBK = kCols / P, CPG = 4 / G;
  constexpr int kSlab = kTile * kSX + kTile * SK;   // floats per wave (>= 1024 / kWaves * ... : the reduction reuses 4 x 1024)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sX = smem + wave * kSlab;
  float* sK = sX + kTile * kSX;
  const int b = blockIdx.z, hb0 = blockIdx.y * HBK, col0 = hb0 * P;
  const int64_t l_begin = (int64_t)blockIdx.x * tok_per_block;
  const int64_t l_end = l_begin + tok_per_block < L ? l_begin + tok_per_block : L;
  const int q = lane & 15, hl = (4 * q) / P, h = hb0 + hl;
  const bool hv = h < H;
  float a = 0.f, bias = 0.f;
  if (WEIGHTED && hv) {
    a = __expf(A_log[h * p_hs]);
    bias = dt_bias[h * p_hs];
  }
  const int i15 = lane & 15, kk = lane >> 4;
  int mc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) mc[c] = memcol<P, G>(c, i15);
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (int)((l_end - l_begin + kTile - 1) / kTile);
  // two tiles in flight per wave: the loads of tile t + 1 are issued before tile t is staged and multiplied (a wave used to pay one
  // full memory round trip per tile: 2 tiles per wave at the refiner shape)
  struct Tile {
    float4 xv[4];
    float z[4];
    KTile<T, KW> k;
    int nvalid;
  };
  auto load_tile = [&](int tile, Tile& tr) {
    const int64_t t0 = l_begin + (int64_t)tile * kTile;
    tr.nvalid = (int)(l_end - t0 < kTile ? l_end - t0 : kTile);
    const int64_t row0 = (int64_t)b * L + t0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = kk + 4 * r;
      const bool ok = hv && t < tr.nvalid;
      tr.xv[r] = ld_row4<T>(V, row0 + t, ldv, col0 + 4 * q, ok);
      tr.z[r] = (WEIGHTED && ok) ? Io<T>::ld(dt_raw + (row0 + t) * lddt + (int64_t)h * dt_hs) : 0.f;
    }
    tr.k.load(K, ldk, row0, tr.nvalid, lane);
  };
  auto process = [&](Tile& tr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (WEIGHTED) {
        const bool ok = hv && kk + 4 * r < tr.nvalid;
        const float w = ok ? softplusf_(tr.z[r] + bias) * a : 0.f;
        tr.xv[r].x *= w; tr.xv[r].y *= w; tr.xv[r].z *= w; tr.xv[r].w *= w;
      }
    }
    tr.k.store(sK, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_st4(sX + (kk + 4 * r) * kSX + 4 * q, tr.xv[r]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int tok = 4 * s + kk;
      float av[G];
#pragma unroll
      for (int g = 0; g < G; ++g) av[g] = i15 < N ? sK[tok * SK + g * N + i15] : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c / CPG], sX[tok * kSX + mc[c]], acc[c], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  };
  {
    Tile ta, tb;
    int tile = wave;
    if (tile < ntiles) load_tile(tile, ta);
    for (; tile < ntiles; tile += 2 * kWaves) {
      const bool second = tile + kWaves < ntiles;
      if (second) load_tile(tile + kWaves, tb);
      process(ta);
      if (second) {
        if (tile + 2 * kWaves < ntiles) load_tile(tile + 2 * kWaves, ta);
        process(tb);
      }
    }
  }
  // sum the 4 waves through LDS (fixed order), then one store of the head block's N x 64 state.  MFMA D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
  __syncthreads();
  float* red = smem;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 1024 + (c * 4 + r) * 64 + lane] = acc[c][r];
  __syncthreads();
  float* dst = out + (int64_t)b * out_bstride + (int64_t)blockIdx.x * out_cstride;
#pragma unroll
  for (int k = 0; k < 1024 / kBlock; ++k) {
    const int e = threadIdx.x + kBlock * k;
    const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
    const int cr = e >> 6, ln = e & 63, c = cr >> 2, r = cr & 3;
    const int n = (ln >> 4) * 4 + r, col = memcol<P, G>(c, ln & 15), hh = hb0 + col / P;
    if (n < N && hh < H) dst[((int64_t)hh * N + n) * P + (col % P)] = v;
  }
}

// out[b, e] = sum_k part[b, k, e],  e in [0, E): 4 k-slices per workgroup, 4 independent loads in flight per lane
__global__ __launch_bounds__(256) void ssd_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int nk, int64_t E) {
  __shared__ float sm[4][64];
  const int b = blockIdx.y;
  const int e_local = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + e_local;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (e < E) {
    const float* p = part + (int64_t)b * nk * E + e;
    int k = ks;
    for (; k + 12 < nk; k += 16) {
      t0 += p[(int64_t)k * E];
      t1 += p[(int64_t)(k + 4) * E];
      t2 += p[(int64_t)(k + 8) * E];
      t3 += p[(int64_t)(k + 12) * E];
    }
    for (; k < nk; k += 4) t0 += p[(int64_t)k * E];
  }
  sm[ks][e_local] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (ks == 0 && e < E) out[(int64_t)b * E + e] = (sm[0][e_local] + sm[1][e_local]) + (sm[2][e_local] + sm[3][e_local]);
}

// ================================================================================================ pass 2
// y = Cm . KV + D x   (+ LayerNorm over the token row as the epilogue when the row is one head block: H*P == 64)
template <typename T, int P, int N, int G, bool LN>
__global__ __launch_bounds__(kBlock) void ssd_apply_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ Cm, int64_t ldc,
                                                           const float* __restrict__ D, int64_t p_hs, const float* __restrict__ kv,
                                                           T* __restrict__ y, int64_t ldy, const float* __restrict__ ln_w,
                                                           const float* __restrict__ ln_b, T* __restrict__ yn, int64_t ldyn,
                                                           float* __restrict__ mu, float* __restrict__ rstd, float eps, int64_t L, int H,
                                                           int tok_per_block) {
  constexpr int KW = G * N, SK = KW + 4, HBK = kCols / P, CPG = 4 / G, NS = N / 4;
  constexpr int kSlab = kTile * kSX + kTile * SK;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sX = smem + wave * kSlab;
  float* sC = sX + kTile * kSX;
  const int b = blockIdx.z, hb0 = blockIdx.y * HBK, col0 = hb0 * P;
  const int64_t l_begin = (int64_t)blockIdx.x * tok_per_block;
  const int64_t l_end = l_begin + tok_per_block < L ? l_begin + tok_per_block : L;
  const int q = lane & 15, i15 = lane & 15, kk = lane >> 4;
  const bool cv = col0 + 4 * q < H * P;   // this lane's 4 columns exist
  // KV as MFMA B operands: kvr[c][s] = KV[h(c, j)][n = kk*NS + s][p(c, j)]  (the k index is permuted identically on the A side)
  float kvr[4][NS], Dh[4];
  int mc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    mc[c] = memcol<P, G>(c, i15);
    const int hh = hb0 + mc[c] / P;
    const bool ok = hh < H;
    Dh[c] = ok ? D[hh * p_hs] : 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) kvr[c][s] = ok ? kv[(((int64_t)b * H + hh) * N + kk * NS + s) * P + (mc[c] % P)] : 0.f;
  }
  float4 lw = make_float4(0.f, 0.f, 0.f, 0.f), lb = lw;
  if (LN) {
    lw = *reinterpret_cast<const float4*>(ln_w + 4 * q);
    lb = *reinterpret_cast<const float4*>(ln_b + 4 * q);
  }
  const int ntiles = (int)((l_end - l_begin + kTile - 1) / kTile);
  for (int tile = wave; tile < ntiles; tile += kWaves) {
    const int64_t t0 = l_begin + (int64_t)tile * kTile;
    const int nvalid = (int)(l_end - t0 < kTile ? l_end - t0 : kTile);
    const int64_t row0 = (int64_t)b * L + t0;
    float4 xv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xv[r] = ld_row4<T>(x, row0 + kk + 4 * r, ldx, col0 + 4 * q, cv && kk + 4 * r < nvalid);
    stage_k_tile<T, KW>(Cm, ldc, row0, nvalid, sC, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_st4(sX + (kk + 4 * r) * kSX + 4 * q, xv[r]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // A[i = token][k]: lane (i, kk) holds Cm[token i][g*N + kk*NS + s] for step s — one 8- or 16-byte LDS read
      float av[NS];
      const float* ap = sC + i15 * SK + (c / CPG) * N + kk * NS;
      if (NS == 4) {
        const float4 t = lds_ld4(ap);
        av[0] = t.x; av[1] = t.y; av[NS - 2] = t.z; av[NS - 1] = t.w;
      } else {
        av[0] = ap[0]; av[1] = ap[1];
      }
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], kvr[c][s], acc, 0, 0, 0);
      // D layout: row = token kk*4 + r, column j -> this lane's column mc[c]; the result replaces x in the slab
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* p = sX + (kk * 4 + r) * kSX + mc[c];
        *p = fmaf(Dh[c], *p, acc[r]);
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = kk + 4 * r;
      const float4 v = lds_ld4(sX + t * kSX + 4 * q);
      const bool ok = cv && t < nvalid;
      if (ok) Io<T>::st4(y + (row0 + t) * ldy + col0 + 4 * q, v);
      if (LN) {   // the 16 lanes of a token hold its whole row: statistics by xor-shuffles inside the 16-lane group
        float s1 = (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) s1 += __shfl_xor(s1, o, 64);
        const float m = s1 * (1.0f / kCols);
        const float dx0 = v.x - m, dx1 = v.y - m, dx2 = v.z - m, dx3 = v.w - m;
        float s2 = (dx0 * dx0 + dx1 * dx1) + (dx2 * dx2 + dx3 * dx3);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) s2 += __shfl_xor(s2, o, 64);
        const float rs = rsqrtf(s2 * (1.0f / kCols) + eps);
        if (ok) {
          Io<T>::st4(yn + (row0 + t) * ldyn + 4 * q, make_float4(fmaf(dx0 * rs, lw.x, lb.x), fmaf(dx1 * rs, lw.y, lb.y), fmaf(dx2 * rs, lw.z, lb.z),
                                                                fmaf(dx3 * rs, lw.w, lb.w)));
          if (q == 0) {
            mu[row0 + t] = m;
            rstd[row0 + t] = rs;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ================================================================================================ backward
// Per (token, head), with z = dt_raw + dt_bias, w = softplus(z) a:
//   t[p]   = sum_n Bm[n] dKV[n][p]          dx[p] = D dy[p] + w t[p]
//   dw     = sum_p t[p] x[p]                ddt_raw = dw * a * sigmoid(z)
//   dCm[n] = sum_{h in g} sum_p dy[p] KV[n][p]        dBm[n] = sum_{h in g} w sum_p dKV[n][p] x[p]
//   dD += sum_p dy x ;  ddt_bias += ddt_raw ;  dA_log += dw * w
// All three contractions are computed TRANSPOSED (rows = columns / states, MFMA columns = the 16 tokens of the tile), so a
// lane (token j = lane & 15, kk = lane >> 4) ends up with FOUR CONSECUTIVE memory columns of its token in every accumulator:
// t, x, dy of one (token, head) meet in one lane as float4s (no cross-lane sums for P = 4), and dBm / dCm / dx leave as 16-byte
// pieces.  LDS is only the transposer between the coalesced row layout of the global loads / stores and that fragment layout.
// hpart: per (b, token block) partial of [dD | ddt_bias | dA_log]; bcpart: per head-block partial of [dBm | dCm] rows when
// H spans several head blocks (else written straight out).
template <typename T, int P, int N, int G>
__global__ __launch_bounds__(kBlock) void ssd_bwd_kernel(
    const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx, const T* __restrict__ Bm, int64_t ldb,
    const T* __restrict__ dt_raw, int64_t lddt, int64_t dt_hs, const float* __restrict__ dt_bias, const float* __restrict__ A_log,
    const float* __restrict__ D, int64_t p_hs, const float* __restrict__ kv, const float* __restrict__ dkv, T* __restrict__ dx,
    int64_t lddx, T* __restrict__ dBm, int64_t lddb, T* __restrict__ dCm, int64_t lddc, T* __restrict__ ddt_raw, int64_t ldddt,
    float* __restrict__ hpart, float* __restrict__ bcpart, int64_t L, int H, int tok_per_block, int nblk, int nhb) {
  constexpr int KW = G * N, HBK = kCols / P, CPG = 4 / G, NS = N / 4, LPH = P / 4;   // LPH: kk-lanes that share a head
  // row stride of the per-(token, head) scalars sW / sS: the fragment-side reads walk the 16 tokens of a tile (one per lane) at a fixed
  // head — with a stride of HBK = 16 words that is an 8-way bank conflict per 32-lane half (round 2 measured 37 % of this kernel's LDS
  // cycles as conflict cycles); 17 spreads the tokens over distinct banks
  constexpr int kSW = HBK + 1;
  constexpr int kSlab = 2 * kTile * kSX + 2 * kTile * kSW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sDY = smem + wave * kSlab;
  float* sX = sDY + kTile * kSX;          // x, then dx in place
  float* sW = sX + kTile * kSX;           // w per (token, head of the block)
  float* sS = sW + kTile * kSW;           // a * sigmoid(z), then ddt_raw in place
  const int b = blockIdx.z, hb0 = blockIdx.y * HBK, col0 = hb0 * P;
  const int64_t l_begin = (int64_t)blockIdx.x * tok_per_block;
  const int64_t l_end = l_begin + tok_per_block < L ? l_begin + tok_per_block : L;
  const int q = lane & 15, j = lane & 15, kk = lane >> 4;
  const int hl_q = (4 * q) / P, h_q = hb0 + hl_q;
  const bool hv = h_q < H;
  const float a = hv ? __expf(A_log[h_q * p_hs]) : 0.f, bias = hv ? dt_bias[h_q * p_hs] : 0.f;
  // fragment-layout constants of this lane: column block c -> its 4 consecutive columns cb[c] .. cb[c]+3 of head hf[c]
  int cb[4], hf[4];
  float Dh[4];
  bool hok[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    cb[c] = memcol<P, G>(c, 4 * kk);
    hf[c] = cb[c] / P;
    hok[c] = hb0 + hf[c] < H;
    Dh[c] = hok[c] ? D[(hb0 + hf[c]) * p_hs] : 0.f;
  }
  // register-resident MFMA A operands.  dkvA[c][s] = dKV[h(c, i)][n = kk*NS + s][p(c, i)]  (rows i = the block's 16 columns, k = n);
  // kvC / dkvB[c][e] = (d)KV[head of column memcol(c, 4 kk + e)][n = i][its p]              (rows i = n, k = the group's columns)
  float dkvA[4][NS], kvC[4][4], dkvB[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = memcol<P, G>(c, j), hh = hb0 + col / P;
#pragma unroll
    for (int s = 0; s < NS; ++s) dkvA[c][s] = hh < H ? dkv[(((int64_t)b * H + hh) * N + kk * NS + s) * P + (col % P)] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col2 = cb[c] + e;
      const bool ok = hok[c] && j < N;
      const int64_t o = (((int64_t)b * H + hb0 + hf[c]) * N + j) * P + (col2 % P);
      kvC[c][e] = ok ? kv[o] : 0.f;
      dkvB[c][e] = ok ? dkv[o] : 0.f;
    }
  }
  float stD[4] = {0.f, 0.f, 0.f, 0.f}, stB[4] = {0.f, 0.f, 0.f, 0.f}, stA[4] = {0.f, 0.f, 0.f, 0.f};
  const int ntiles = (int)((l_end - l_begin + kTile - 1) / kTile);
  for (int tile = wave; tile < ntiles; tile += kWaves) {
    const int64_t t0 = l_begin + (int64_t)tile * kTile;
    const int nvalid = (int)(l_end - t0 < kTile ? l_end - t0 : kTile);
    const int64_t row0 = (int64_t)b * L + t0;
    {   // coalesced row-layout loads -> LDS (the transposer)
      float4 gv[4], xv[4];
      float wv[4], sv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = kk + 4 * r;
        const bool ok = hv && t < nvalid;
        gv[r] = ld_row4<T>(dy, row0 + t, lddy, col0 + 4 * q, ok);
        xv[r] = ld_row4<T>(x, row0 + t, ldx, col0 + 4 * q, ok);
        const float z = ok ? Io<T>::ld(dt_raw + (row0 + t) * lddt + (int64_t)h_q * dt_hs) + bias : 0.f;
        wv[r] = ok ? softplusf_(z) * a : 0.f;
        sv[r] = ok ? a * sigmoidf_(z) : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = kk + 4 * r;
        lds_st4(sDY + t * kSX + 4 * q, gv[r]);
        lds_st4(sX + t * kSX + 4 * q, xv[r]);
        sW[t * kSW + hl_q] = wv[r];    // P = 8: the two lanes of a head write the same value
        sS[t * kSW + hl_q] = sv[r];
      }
    }
    // Bm of token j in fragment order: n = kk*NS + s (the permutation dkvA uses)
    float bf[G][NS];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const T* bp = Bm + (row0 + j) * ldb + g * N + kk * NS;
      if (NS == 4) {
        const float4 t = j < nvalid ? Io<T>::ld4(bp) : make_float4(0.f, 0.f, 0.f, 0.f);
        bf[g][0] = t.x; bf[g][1] = t.y; bf[g][NS - 2] = t.z; bf[g][NS - 1] = t.w;
      } else {
        bf[g][0] = j < nvalid ? Io<T>::ld(bp) : 0.f;
        bf[g][1] = j < nvalid ? Io<T>::ld(bp + 1) : 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();
    f32x4 aC[G], aB[G];
#pragma unroll
    for (int g = 0; g < G; ++g) aC[g] = aB[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 gf = lds_ld4(sDY + j * kSX + cb[c]), xf = lds_ld4(sX + j * kSX + cb[c]);
      const float wf = sW[j * kSW + hf[c]], sf = sS[j * kSW + hf[c]];
      const float ge[4] = {gf.x, gf.y, gf.z, gf.w}, xe[4] = {xf.x, xf.y, xf.z, xf.w};
      // dCm^T += KV_g . dy_g^T,  dBm^T += dKV_g . (x w)_g^T     (rows = n, columns = tokens; k = this block's 16 columns)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        aC[c / CPG] = __builtin_amdgcn_mfma_f32_16x16x4f32(kvC[c][e], ge[e], aC[c / CPG], 0, 0, 0);
        aB[c / CPG] = __builtin_amdgcn_mfma_f32_16x16x4f32(dkvB[c][e], xe[e] * wf, aB[c / CPG], 0, 0, 0);
      }
      // t^T = dKV_c^T . Bm_g^T     (rows = the block's columns, columns = tokens): reg r <-> column cb[c] + r
      f32x4 tt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) tt = __builtin_amdgcn_mfma_f32_16x16x4f32(dkvA[c][s], bf[c / CPG][s], tt, 0, 0, 0);
      float dw = (tt[0] * xe[0] + tt[1] * xe[1]) + (tt[2] * xe[2] + tt[3] * xe[3]);
      float dd = (ge[0] * xe[0] + ge[1] * xe[1]) + (ge[2] * xe[2] + ge[3] * xe[3]);
#pragma unroll
      for (int o = 1; o < LPH; o <<= 1) {   // P = 8: a head's 8 columns sit in two kk-lanes
        dw += __shfl_xor(dw, 16 * o, 64);
        dd += __shfl_xor(dd, 16 * o, 64);
      }
      const float dz = dw * sf;
      lds_st4(sX + j * kSX + cb[c], make_float4(fmaf(wf, tt[0], Dh[c] * ge[0]), fmaf(wf, tt[1], Dh[c] * ge[1]), fmaf(wf, tt[2], Dh[c] * ge[2]),
                                                fmaf(wf, tt[3], Dh[c] * ge[3])));
      if ((cb[c] % P) == 0) {   // one lane per (token, head)
        sS[j * kSW + hf[c]] = dz;
        stD[c] += dd;
        stB[c] += dz;
        stA[c] += dw * wf;
      }
    }
    // dBm / dCm rows: lane (token j, kk) holds n = 4 kk .. 4 kk + 3 of every group
    if (j < nvalid && 4 * kk < N) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 vb = make_float4(aB[g][0], aB[g][1], aB[g][2], aB[g][3]), vc = make_float4(aC[g][0], aC[g][1], aC[g][2], aC[g][3]);
        if (nhb == 1) {
          Io<T>::st4(dBm + (row0 + j) * lddb + g * N + 4 * kk, vb);
          Io<T>::st4(dCm + (row0 + j) * lddc + g * N + 4 * kk, vc);
        } else {
          float* dst = bcpart + (((int64_t)blockIdx.y * gridDim.z + b) * L + (t0 + j)) * (2 * KW);
          *reinterpret_cast<float4*>(dst + g * N + 4 * kk) = vb;
          *reinterpret_cast<float4*>(dst + KW + g * N + 4 * kk) = vc;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // coalesced stores of dx rows and of ddt_raw
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = kk + 4 * r;
      if (hv && t < nvalid) Io<T>::st4(dx + (row0 + t) * lddx + col0 + 4 * q, lds_ld4(sX + t * kSX + 4 * q));
    }
    {
      constexpr int IT = (kTile * HBK + 63) / 64;
#pragma unroll
      for (int r = 0; r < IT; ++r) {
        const int f = lane + 64 * r, t = f / HBK, hh = f % HBK;
        if (f < kTile * HBK && t < nvalid && hb0 + hh < H) Io<T>::st(ddt_raw + (row0 + t) * ldddt + (int64_t)(hb0 + hh) * dt_hs, sS[t * kSW + hh]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // per-head statistics: the lane with (first column of the head) holds the sums of its token column j over the wave's tiles;
  // fold the 16 tokens by shuffles, the waves through LDS, one partial row per workgroup
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      stD[c] += __shfl_xor(stD[c], o, 64);
      stB[c] += __shfl_xor(stB[c], o, 64);
      stA[c] += __shfl_xor(stA[c], o, 64);
    }
  }
  __syncthreads();
  float* sred = smem;   // [kWaves][3][HBK]
  if (j == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if ((cb[c] % P) == 0) {
        sred[(wave * 3 + 0) * HBK + hf[c]] = stD[c];
        sred[(wave * 3 + 1) * HBK + hf[c]] = stB[c];
        sred[(wave * 3 + 2) * HBK + hf[c]] = stA[c];
      }
  }
  __syncthreads();
  if (threadIdx.x < 3 * HBK) {
    const int k = threadIdx.x / HBK, hl = threadIdx.x % HBK, hh = hb0 + hl;
    if (hh < H) {
      const float v = (sred[(0 * 3 + k) * HBK + hl] + sred[(1 * 3 + k) * HBK + hl]) + (sred[(2 * 3 + k) * HBK + hl] + sred[(3 * 3 + k) * HBK + hl]);
      hpart[(((int64_t)b * nblk + blockIdx.x) * 3 + k) * H + hh] = v;
    }
  }
}

// sums the per-head-block [dBm | dCm] partials (only when H spans several head blocks)
template <typename T>
__global__ void ssd_bc_fold_kernel(const float* __restrict__ bcpart, int nhb, int64_t rows, int GN, T* __restrict__ dBm,
                                   int64_t lddb, T* __restrict__ dCm, int64_t lddc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * 2 * GN) return;
  const int64_t row = i / (2 * GN);
  const int c = (int)(i % (2 * GN));
  float t = 0.f;
  for (int k = 0; k < nhb; ++k) t += bcpart[((int64_t)k * rows + row) * (2 * GN) + c];
  if (c < GN) Io<T>::st(dBm + row * lddb + c, t);
  else Io<T>::st(dCm + row * lddc + (c - GN), t);
}

struct Ws {
  float *part, *dkv, *hpart, *bcpart;
  int64_t bytes;
};

Ws carve(void* ws, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N, int64_t G) {
  const Geo gr = make_geo(B, L, H, P, true), gs = make_geo(B, L, H, P, false);
  Ws w;
  int64_t off = 0;
  auto take = [&](int64_t nfloat) {
    float* p = ws ? (float*)((char*)ws + off) : nullptr;
    off += adnm_align(nfloat * 4, 256);
    return p;
  };
  w.part = take(gr.nchunk > 1 ? B * gr.nchunk * H * N * P : 0);
  w.dkv = take(B * H * N * P);
  w.hpart = take(B * gs.nchunk * 3 * H);
  w.bcpart = take(gs.nhb > 1 ? (int64_t)gs.nhb * B * L * 2 * G * N : 0);
  w.bytes = off;
  return w;
}

int check_common(const char* who, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N, int64_t G, int dtype) {
  ADNM_REQUIRE(B > 0 && L > 0 && H > 0, "%s: empty shape B=%lld L=%lld H=%lld", who, (long long)B, (long long)L, (long long)H);
  ADNM_REQUIRE((P == 4 && (N == 8 || N == 16)) || (P == 8 && N == 8),
               "%s: (headdim P=%lld, states-per-group N=%lld) not in {(4,8),(4,16),(8,8)}", who, (long long)P, (long long)N);
  ADNM_REQUIRE(G == 1 || G == 2 || G == 4, "%s: groups G=%lld not in {1,2,4}", who, (long long)G);
  ADNM_REQUIRE(B <= 65535 && adnm_cdiv(H * P, kCols) <= 65535, "%s: grid too large", who);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "%s: bad dtype %d", who, dtype);
  return ADNM_OK;
}

// (headdim, state, groups) instantiations: the MFMA column-block <-> group map needs them at compile time
#define DISPATCH_PNG(FN, ...)                                                                                   \
  do {                                                                                                          \
    if (P == 4 && N == 8) { if (G == 1) FN<T, 4, 8, 1>(__VA_ARGS__); else if (G == 2) FN<T, 4, 8, 2>(__VA_ARGS__); else FN<T, 4, 8, 4>(__VA_ARGS__); }       \
    else if (P == 4 && N == 16) { if (G == 1) FN<T, 4, 16, 1>(__VA_ARGS__); else if (G == 2) FN<T, 4, 16, 2>(__VA_ARGS__); else FN<T, 4, 16, 4>(__VA_ARGS__); } \
    else { if (G == 1) FN<T, 8, 8, 1>(__VA_ARGS__); else if (G == 2) FN<T, 8, 8, 2>(__VA_ARGS__); else FN<T, 8, 8, 4>(__VA_ARGS__); }                        \
  } while (0)

template <typename T, int P, int N, int G>
int run_kv(bool weighted, const void* V, int64_t ldv, const void* K, int64_t ldk, const void* dt_raw, int64_t lddt, int64_t dt_hs,
           const float* dt_bias, const float* A_log, int64_t p_hs, float* part, float* out, int64_t B, int64_t L, int64_t H, hipStream_t st) {
  const Geo g = make_geo(B, L, H, P, true);
  const dim3 grid(g.nchunk, g.nhb, (unsigned)B);
  constexpr int KW = G * N, kSlab = kTile * kSX + kTile * (KW + 4);
  const size_t smem = sizeof(float) * (size_t)(kWaves * kSlab > 4096 ? kWaves * kSlab : 4096);
  const int64_t E = H * N * P;
  float* dst = g.nchunk > 1 ? part : out;
  const int64_t bs = g.nchunk > 1 ? (int64_t)g.nchunk * E : E, cs = g.nchunk > 1 ? E : 0;
  if (weighted) {
    ADNM_ALLOW_LDS((ssd_kv_kernel<T, P, N, G, true>), smem, "ssd_kv");
    ADNM_PROF("ssd_kv", st, (double)sizeof(T) * B * L * (H * P + (double)g.nhb * G * N + H));
    ssd_kv_kernel<T, P, N, G, true><<<grid, kBlock, smem, st>>>((const T*)V, ldv, (const T*)K, ldk, (const T*)dt_raw, lddt, dt_hs, dt_bias, A_log,
                                                                p_hs, dst, bs, cs, L, (int)H, g.tok);
  } else {
    ADNM_ALLOW_LDS((ssd_kv_kernel<T, P, N, G, false>), smem, "ssd_dkv");
    ADNM_PROF("ssd_dkv", st, (double)sizeof(T) * B * L * (H * P + (double)g.nhb * G * N));
    ssd_kv_kernel<T, P, N, G, false><<<grid, kBlock, smem, st>>>((const T*)V, ldv, (const T*)K, ldk, nullptr, 0, 0, nullptr, nullptr, 0, dst, bs, cs, L,
                                                                 (int)H, g.tok);
  }
  if (g.nchunk > 1) {
    ADNM_PROF("ssd_fold", st, 4.0 * B * E * (g.nchunk + 1));
    ssd_fold_kernel<<<dim3((unsigned)adnm_cdiv(E, 64), (unsigned)B), 256, 0, st>>>(part, out, g.nchunk, E);
  }
  return ADNM_OK;
}

template <typename T, int P, int N, int G>
int run_apply(const void* x, int64_t ldx, const void* Cm, int64_t ldc, const float* D, int64_t p_hs, const float* kv, void* y, int64_t ldy,
              const float* ln_w, const float* ln_b, void* yn, int64_t ldyn, float* mu, float* rstd, float eps, int64_t B, int64_t L, int64_t H,
              hipStream_t st) {
  const Geo g = make_geo(B, L, H, P, false, true);
  constexpr int KW = G * N, kSlab = kTile * kSX + kTile * (KW + 4);
  const size_t smem = sizeof(float) * (size_t)kWaves * kSlab;
  const dim3 grid(g.nchunk, g.nhb, (unsigned)B);
  if (yn) {
    ADNM_ALLOW_LDS((ssd_apply_kernel<T, P, N, G, true>), smem, "ssd_apply_ln");
    ADNM_PROF("ssd_apply_ln", st, (double)sizeof(T) * B * L * (3 * H * P + G * N) + 8.0 * B * L);
    ssd_apply_kernel<T, P, N, G, true><<<grid, kBlock, smem, st>>>((const T*)x, ldx, (const T*)Cm, ldc, D, p_hs, kv, (T*)y, ldy, ln_w, ln_b, (T*)yn, ldyn,
                                                                   mu, rstd, eps, L, (int)H, g.tok);
  } else {
    ADNM_ALLOW_LDS((ssd_apply_kernel<T, P, N, G, false>), smem, "ssd_apply");
    ADNM_PROF("ssd_apply", st, (double)sizeof(T) * B * L * (2 * H * P + (double)g.nhb * G * N));
    ssd_apply_kernel<T, P, N, G, false><<<grid, kBlock, smem, st>>>((const T*)x, ldx, (const T*)Cm, ldc, D, p_hs, kv, (T*)y, ldy, nullptr, nullptr, nullptr,
                                                                    0, nullptr, nullptr, 0.f, L, (int)H, g.tok);
  }
  return ADNM_OK;
}

template <typename T, int P, int N, int G>
int run_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* Bm, int64_t ldb, const void* dt_raw, int64_t lddt,
            int64_t dt_hs, const float* dt_bias, const float* A_log, const float* D, int64_t p_hs, const float* kv, const float* dkv, void* dx,
            int64_t lddx, void* dBm, int64_t lddb, void* dCm, int64_t lddc, void* ddt_raw, int64_t ldddt, float* hpart, float* bcpart,
            float* ddt_bias, float* dA_log, float* dD, int64_t B, int64_t L, int64_t H, hipStream_t st) {
  const Geo g = make_geo(B, L, H, P, false);
  constexpr int HBK = kCols / P, kSlab = 2 * kTile * kSX + 2 * kTile * (HBK + 1);
  const size_t smem = sizeof(float) * (size_t)kWaves * kSlab;
  const dim3 grid(g.nchunk, g.nhb, (unsigned)B);
  {
    ADNM_ALLOW_LDS((ssd_bwd_kernel<T, P, N, G>), smem, "ssd_bwd");
    ADNM_PROF("ssd_bwd", st, (double)sizeof(T) * B * L * (3 * H * P + (double)g.nhb * 3 * G * N + 2 * H));
    ssd_bwd_kernel<T, P, N, G><<<grid, kBlock, smem, st>>>((const T*)dy, lddy, (const T*)x, ldx, (const T*)Bm, ldb, (const T*)dt_raw, lddt, dt_hs, dt_bias,
                                                           A_log, D, p_hs, kv, dkv, (T*)dx, lddx, (T*)dBm, lddb, (T*)dCm, lddc, (T*)ddt_raw, ldddt, hpart,
                                                           bcpart, L, (int)H, g.tok, g.nchunk, g.nhb);
  }
  if (g.nhb > 1) {
    const int64_t tot = B * L * 2 * G * N;
    ADNM_PROF("ssd_bc_fold", st, 4.0 * tot * g.nhb);
    ssd_bc_fold_kernel<T><<<(unsigned)adnm_cdiv(tot, 256), 256, 0, st>>>(bcpart, g.nhb, B * L, (int)(G * N), (T*)dBm, lddb, (T*)dCm, lddc);
  }
  adnm_launch_fold("ssd_head_fold", hpart, (int)(B * g.nchunk), 3 * (int)H, {dD, (int)H}, {ddt_bias, (int)H}, {dA_log, (int)H}, {nullptr, 0}, st);
  return ADNM_OK;
}

}  // namespace

extern "C" int64_t adnm_ssd_ws_bytes(int64_t B, int64_t L, int64_t H, int64_t P, int64_t N, int64_t G) {
  if (B <= 0 || L <= 0 || H <= 0) return 0;
  return carve(nullptr, B, L, H, P, N, G).bytes;
}

extern "C" int adnm_ssd_reduce_fwd(const void* x, int64_t ldx, const void* Bm, int64_t ldb, const void* Cm, int64_t ldc,
                                   const void* dt_raw, int64_t lddt, int64_t dt_hstride, const float* dt_bias,
                                   const float* A_log, const float* D, int64_t p_hstride, void* y, int64_t ldy, float* kv,
                                   const float* ln_w, const float* ln_b, void* yn, int64_t ldyn, float* ln_mu, float* ln_rstd, float ln_eps,
                                   void* ws, int64_t ws_bytes, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N,
                                   int64_t G, int dtype, adnm_stream_t stream) {
  if (int rc = check_common("ssd_reduce_fwd", B, L, H, P, N, G, dtype)) return rc;
  ADNM_REQUIRE(x && Bm && Cm && dt_raw && dt_bias && A_log && D && y && kv, "ssd_reduce_fwd: null pointer");
  ADNM_REQUIRE(ldx >= H * P && ldy >= H * P && ldb >= G * N && ldc >= G * N && lddt >= (H - 1) * dt_hstride + 1,
               "ssd_reduce_fwd: row strides smaller than the rows they address");
  ADNM_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0, "ssd_reduce_fwd: row strides must be multiples of 4");
  if (yn) {
    ADNM_REQUIRE(H * P == kCols, "ssd_reduce_fwd: the fused LayerNorm epilogue needs the token row to be one head block (H*P == 64), got %lld",
                 (long long)(H * P));
    ADNM_REQUIRE(ln_w && ln_b && ln_mu && ln_rstd && ldyn >= H * P && ldyn % 4 == 0, "ssd_reduce_fwd: incomplete LayerNorm arguments");
  }
  const Ws w = carve(ws, B, L, H, P, N, G);
  if (!ws || ws_bytes < w.bytes) {
    adnm_set_error("ssd_reduce_fwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)w.bytes);
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int rc = ADNM_OK;
  if (dtype == ADNM_F32) {
    using T = float;
    DISPATCH_PNG(rc = run_kv, true, x, ldx, Bm, ldb, dt_raw, lddt, dt_hstride, dt_bias, A_log, p_hstride, w.part, kv, B, L, H, st);
    if (rc) return rc;
    DISPATCH_PNG(rc = run_apply, x, ldx, Cm, ldc, D, p_hstride, kv, y, ldy, ln_w, ln_b, yn, ldyn, ln_mu, ln_rstd, ln_eps, B, L, H, st);
  } else {
    using T = uint16_t;
    DISPATCH_PNG(rc = run_kv, true, x, ldx, Bm, ldb, dt_raw, lddt, dt_hstride, dt_bias, A_log, p_hstride, w.part, kv, B, L, H, st);
    if (rc) return rc;
    DISPATCH_PNG(rc = run_apply, x, ldx, Cm, ldc, D, p_hstride, kv, y, ldy, ln_w, ln_b, yn, ldyn, ln_mu, ln_rstd, ln_eps, B, L, H, st);
  }
  if (rc) return rc;
  ADNM_CHECK_LAUNCH("ssd_reduce_fwd");
  return ADNM_OK;
}

extern "C" int adnm_ssd_reduce_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* Bm, int64_t ldb,
                                   const void* Cm, int64_t ldc, const void* dt_raw, int64_t lddt, int64_t dt_hstride,
                                   const float* dt_bias, const float* A_log, const float* D, int64_t p_hstride,
                                   const float* kv, void* dx, int64_t lddx, void* dBm, int64_t lddb, void* dCm, int64_t lddc,
                                   void* ddt_raw, int64_t ldddt, float* ddt_bias, float* dA_log, float* dD, void* ws,
                                   int64_t ws_bytes, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N, int64_t G,
                                   int dtype, adnm_stream_t stream) {
  if (int rc = check_common("ssd_reduce_bwd", B, L, H, P, N, G, dtype)) return rc;
  ADNM_REQUIRE(dy && x && Bm && Cm && dt_raw && dt_bias && A_log && D && kv && dx && dBm && dCm && ddt_raw && ddt_bias && dA_log && dD,
               "ssd_reduce_bwd: null pointer");
  ADNM_REQUIRE(ldx >= H * P && lddy >= H * P && lddx >= H * P && ldb >= G * N && ldc >= G * N && lddb >= G * N && lddc >= G * N,
               "ssd_reduce_bwd: row strides smaller than the rows they address");
  ADNM_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && lddb % 4 == 0 && lddc % 4 == 0,
               "ssd_reduce_bwd: row strides must be multiples of 4");
  const Ws w = carve(ws, B, L, H, P, N, G);
  if (!ws || ws_bytes < w.bytes) {
    adnm_set_error("ssd_reduce_bwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)w.bytes);
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int rc = ADNM_OK;
  if (dtype == ADNM_F32) {
    using T = float;
    DISPATCH_PNG(rc = run_kv, false, dy, lddy, Cm, ldc, nullptr, 0, 0, nullptr, nullptr, 0, w.part, w.dkv, B, L, H, st);
    if (rc) return rc;
    DISPATCH_PNG(rc = run_bwd, dy, lddy, x, ldx, Bm, ldb, dt_raw, lddt, dt_hstride, dt_bias, A_log, D, p_hstride, kv, w.dkv, dx, lddx, dBm, lddb, dCm,
                 lddc, ddt_raw, ldddt, w.hpart, w.bcpart, ddt_bias, dA_log, dD, B, L, H, st);
  } else {
    using T = uint16_t;
    DISPATCH_PNG(rc = run_kv, false, dy, lddy, Cm, ldc, nullptr, 0, 0, nullptr, nullptr, 0, w.part, w.dkv, B, L, H, st);
    if (rc) return rc;
    DISPATCH_PNG(rc = run_bwd, dy, lddy, x, ldx, Bm, ldb, dt_raw, lddt, dt_hstride, dt_bias, A_log, D, p_hstride, kv, w.dkv, dx, lddx, dBm, lddb, dCm,
                 lddc, ddt_raw, ldddt, w.hpart, w.bcpart, ddt_bias, dA_log, dD, B, L, H, st);
  }
  if (rc) return rc;
  ADNM_CHECK_LAUNCH("ssd_reduce_bwd");
  return ADNM_OK;
}
