// K1 — the SSD "linear-attention duality" reduction that every Mamba2 mixer of ADNM-UNet executes
// (non_casual_linear_attn: ADNssd.py:252-299 single-group branch :278-284, Vssd.py:161-208 grouped :194-206),
// with softplus(dt + dt_bias) (ADNssd.py:318) fused in:
//
//   w[b,l,h]    = softplus(dt_raw + dt_bias[h]) * exp(A_log[h])
//   KV[b,h,n,p] = sum_l Bm[b,l,g,n] * x[b,l,h,p] * w[b,l,h]               (pass 1: global reduction over L)
//   y[b,l,h,p]  = sum_n Cm[b,l,g,n] * KV[b,h,n,p] + D[h] * x[b,l,h,p]      (pass 2: streaming)      g = h % G
//
// There is no recurrence on this branch, so there is nothing to scan: it is a two-pass, HBM-bound reduction
// (~2.3 FMA per byte).  Mapping: one lane per (token, head); the HB = min(64, pow2(H)) lanes of a token are
// adjacent, so a token row is read as one contiguous 16*HB-byte segment and the shared Bm/Cm row is a
// same-address broadcast.  Each lane keeps the N x P state of ITS head in registers (64 VGPRs at N=16, P=4)
// for all the tokens it visits; no LDS in the forward inner loops.  Cross-lane sums over the tokens of a wave in
// pass 1 are xor-shuffles; the backward pass keeps the states in LDS instead and gives a lane one (token, group), so
// its sums over heads stay in registers (see ssd_bwd_kernel); cross-block sums go through
// fp32 partials in the caller's workspace and a small deterministic second kernel — never atomics, so the
// result is bitwise reproducible run to run.
#include "adnm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

__host__ __device__ inline int heads_per_block(int64_t H) {
  int l = 1;
  while (l < 64 && l < H) l <<= 1;
  return l;
}

struct Geo {
  int hb;         // lanes (heads) per token inside a block
  int slots;      // tokens in flight per block step
  int nhb;        // head blocks (grid.y)
  int tok1;       // tokens per block, reduction passes
  int nchunk;     // blocks along L, reduction passes
  int tok2;       // tokens per block, streaming passes
  int nblk2;      // blocks along L, streaming passes
};

struct BwdGeo {
  int nhb, tok, nblk;   // head blocks of kBwdHeadsHost heads, tokens per block, blocks along L
};
constexpr int kBwdHeadsHost = 16;
inline BwdGeo make_bwd_geo(int64_t B, int64_t L, int64_t H, int64_t G) {
  BwdGeo g;
  g.nhb = (int)adnm_cdiv(H, kBwdHeadsHost);
  const int64_t slots = kBlock / G;
  int64_t per_lane = (B * L * g.nhb) / (slots * 1024);   // aim at ~1024 blocks, 1..8 tokens per lane
  per_lane = per_lane < 1 ? 1 : (per_lane > 8 ? 8 : per_lane);
  g.tok = (int)(slots * per_lane);
  g.nblk = (int)adnm_cdiv(L, g.tok);
  return g;
}

inline Geo make_geo(int64_t L, int64_t H, int64_t B = 4) {
  Geo g;
  g.hb = heads_per_block(H);
  g.slots = kBlock / g.hb;
  g.nhb = (int)adnm_cdiv(H, g.hb);
  // tokens a lane walks in the reduction passes: 16 on the big maps; on the deep 4x4 .. 16x16 maps fewer, so that the grid
  // still has ~256 workgroups (a 16-step dependent load chain on 16 workgroups cost 18 us for a 64-token reduction)
  int64_t per_lane = (B * L * g.nhb) / ((int64_t)g.slots * 256);
  per_lane = per_lane < 2 ? 2 : (per_lane > 16 ? 16 : per_lane);
  g.tok1 = g.slots * (int)per_lane;
  g.nchunk = (int)adnm_cdiv(L, g.tok1);
  g.tok2 = g.slots * 8;
  g.nblk2 = (int)adnm_cdiv(L, g.tok2);
  return g;
}

template <typename T, int P>
__device__ __forceinline__ void load_vec(const T* p, float (&v)[P]) {
#pragma unroll
  for (int i = 0; i < P; i += 4) {
    float4 t = Io<T>::ld4(p + i);
    v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
  }
}
template <typename T, int P>
__device__ __forceinline__ void store_vec(T* p, const float (&v)[P]) {
#pragma unroll
  for (int i = 0; i < P; i += 4) Io<T>::st4(p + i, make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]));
}

// ------------------------------------------------------------------------------------------------
// Reduction pass: part[b, chunk, h, n, p] = sum_{l in chunk} K[b,l,g,n] * V[b,l,h,p] * (WEIGHTED ? w : 1)
// used for KV (K=Bm, V=x, weighted) and for dKV in backward (K=Cm, V=dy, unweighted).
template <typename T, int P, int N, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void ssd_outer_reduce_kernel(
    const T* __restrict__ V, int64_t ldv, const T* __restrict__ K, int64_t ldk, const T* __restrict__ dt_raw,
    int64_t lddt, int64_t dt_hs, const float* __restrict__ dt_bias, const float* __restrict__ A_log, int64_t p_hs,
    float* __restrict__ part, int64_t L, int H, int G, int hb, int tok_per_block, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [hb][N*P]
  const int hl = threadIdx.x & (hb - 1);
  const int slot = threadIdx.x / hb;
  const int slots = kBlock / hb;
  const int h = blockIdx.y * hb + hl;
  const bool hv = h < H;
  const int b = blockIdx.z;
  const int64_t l0 = (int64_t)blockIdx.x * tok_per_block;
  const int64_t l1 = (l0 + tok_per_block < L) ? l0 + tok_per_block : L;
  float acc[N][P];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int p = 0; p < P; ++p) acc[n][p] = 0.f;
  float a = 0.f, bias = 0.f;
  if (WEIGHTED && hv) {
    a = __expf(A_log[h * p_hs]);
    bias = dt_bias[h * p_hs];
  }
  const int g = hv ? (h % G) : 0;
  if (hv) {
    for (int64_t l = l0 + slot; l < l1; l += slots) {
      const int64_t row = (int64_t)b * L + l;
      float v[P], k[N];
      load_vec<T, P>(V + row * ldv + (int64_t)h * P, v);
      load_vec<T, N>(K + row * ldk + g * N, k);
      if (WEIGHTED) {
        const float w = softplusf_(Io<T>::ld(dt_raw + row * lddt + h * dt_hs) + bias) * a;
#pragma unroll
        for (int p = 0; p < P; ++p) v[p] *= w;
      }
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[n][p] = fmaf(k[n], v[p], acc[n][p]);
    }
  }
  // fold the token slots that share a wave, then the waves of the block through LDS
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int p = 0; p < P; ++p) acc[n][p] = wave_sum_from(acc[n][p], hb);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int wv = 1; wv < kWaves; ++wv) {
    if (wave == wv && lane < hb) {
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int p = 0; p < P; p += 4)
          *reinterpret_cast<float4*>(smem + ((n * P + p) / 4 * hb + lane) * 4) =
              make_float4(acc[n][p], acc[n][p + 1], acc[n][p + 2], acc[n][p + 3]);
    }
    __syncthreads();
    if (wave == 0 && lane < hb) {
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int p = 0; p < P; p += 4) {
          float4 t = *reinterpret_cast<const float4*>(smem + ((n * P + p) / 4 * hb + lane) * 4);
          acc[n][p] += t.x; acc[n][p + 1] += t.y; acc[n][p + 2] += t.z; acc[n][p + 3] += t.w;
        }
    }
    __syncthreads();
  }
  if (wave == 0 && lane < hb && hv) {
    float* dst = part + (((int64_t)b * nchunk + blockIdx.x) * H + h) * (N * P);
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int p = 0; p < P; p += 4)
        *reinterpret_cast<float4*>(dst + n * P + p) = make_float4(acc[n][p], acc[n][p + 1], acc[n][p + 2], acc[n][p + 3]);
  }
}

// out[b, e] = sum_k part[b, k, e],  e in [0, E)
__global__ __launch_bounds__(256) void ssd_fold_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                       int nk, int64_t E) {
  __shared__ float sm[4][64];
  const int b = blockIdx.y;
  const int e_local = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + e_local;
  float t = 0.f;
  if (e < E)
    for (int k = ks; k < nk; k += 4) t += part[((int64_t)b * nk + k) * E + e];
  sm[ks][e_local] = t;
  __syncthreads();
  if (ks == 0 && e < E) out[(int64_t)b * E + e] = (sm[0][e_local] + sm[1][e_local]) + (sm[2][e_local] + sm[3][e_local]);
}

// ------------------------------------------------------------------------------------------------
// Streaming pass: y = Cm . KV + D x
template <typename T, int P, int N>
__global__ __launch_bounds__(kBlock) void ssd_apply_kernel(const T* __restrict__ x, int64_t ldx,
                                                           const T* __restrict__ Cm, int64_t ldc,
                                                           const float* __restrict__ D, int64_t p_hs,
                                                           const float* __restrict__ kv, T* __restrict__ y,
                                                           int64_t ldy, int64_t L, int H, int G, int hb,
                                                           int tok_per_block) {
  const int hl = threadIdx.x & (hb - 1);
  const int slot = threadIdx.x / hb;
  const int slots = kBlock / hb;
  const int h = blockIdx.y * hb + hl;
  if (h >= H) return;
  const int b = blockIdx.z;
  const int64_t l0 = (int64_t)blockIdx.x * tok_per_block;
  const int64_t l1 = (l0 + tok_per_block < L) ? l0 + tok_per_block : L;
  float s[N][P];
  const float* src = kv + ((int64_t)b * H + h) * (N * P);
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int p = 0; p < P; p += 4) {
      float4 t = *reinterpret_cast<const float4*>(src + n * P + p);
      s[n][p] = t.x; s[n][p + 1] = t.y; s[n][p + 2] = t.z; s[n][p + 3] = t.w;
    }
  const float Dh = D[h * p_hs];
  const int g = h % G;
  for (int64_t l = l0 + slot; l < l1; l += slots) {
    const int64_t row = (int64_t)b * L + l;
    float v[P], c[N], o[P];
    load_vec<T, P>(x + row * ldx + (int64_t)h * P, v);
    load_vec<T, N>(Cm + row * ldc + g * N, c);
#pragma unroll
    for (int p = 0; p < P; ++p) o[p] = Dh * v[p];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int p = 0; p < P; ++p) o[p] = fmaf(c[n], s[n][p], o[p]);
    store_vec<T, P>(y + row * ldy + (int64_t)h * P, o);
  }
}

// ------------------------------------------------------------------------------------------------
// Backward streaming pass.  Per (token, head):
//   t[p]   = sum_n Bm[n] dKV[n][p]          dx[p] = D dy[p] + w t[p]
//   dw     = sum_p t[p] x[p]                ddt_raw = dw * a * sigmoid(dt_raw + bias)
//   dCm[n] = sum_{h in g} sum_p dy[p] KV[n][p]        dBm[n] = sum_{h in g} w sum_p dKV[n][p] x[p]
//   dD += sum_p dy x ;  ddt_bias += ddt_raw ;  dA_log += dw * w
// Mapping: a block owns kBwdHeads heads of one batch item; their KV / dKV states sit in LDS (every lane of a group reads
// the same address: a broadcast).  A lane owns one (token, group) and walks the heads of its group, so the sums over heads
// that dBm / dCm need are plain register accumulations — the earlier (token, head)-per-lane version spent its time in the
// 96 cross-lane shuffles per token that this removes.  hpart: per (b, token block) partial of [dD | ddt_bias | dA_log];
// bcpart: per head-block partial of [dBm | dCm] rows when H spans several head blocks (else written straight out).
constexpr int kBwdHeads = 16;

template <typename T, int P, int N, int G>
__global__ __launch_bounds__(kBlock) void ssd_bwd_kernel(
    const T* __restrict__ dy, int64_t lddy, const T* __restrict__ x, int64_t ldx, const T* __restrict__ Bm, int64_t ldb,
    const T* __restrict__ dt_raw, int64_t lddt, int64_t dt_hs, const float* __restrict__ dt_bias, const float* __restrict__ A_log,
    const float* __restrict__ D, int64_t p_hs, const float* __restrict__ kv, const float* __restrict__ dkv, T* __restrict__ dx,
    int64_t lddx, T* __restrict__ dBm, int64_t lddb, T* __restrict__ dCm, int64_t lddc, T* __restrict__ ddt_raw, int64_t ldddt,
    float* __restrict__ hpart, float* __restrict__ bcpart, int64_t L, int H, int tok_per_block, int nblk, int nhb) {
  constexpr int HPL = kBwdHeads / G;      // heads per lane
  constexpr int SST = N * P + 4;          // LDS floats per head: +4 keeps float4 alignment and staggers the banks of the G groups
  __shared__ __attribute__((aligned(16))) float sS[kBwdHeads * SST];
  __shared__ __attribute__((aligned(16))) float sD[kBwdHeads * SST];
  __shared__ float sc[kBwdHeads][3];                  // exp(A_log), dt_bias, D
  __shared__ float sred[kWaves][kBwdHeads * 3];
  __shared__ float sacc[HPL * 3][kBlock];             // per-thread [dD, ddt_bias, dA_log] of each of its heads (column = thread: no conflicts)
  const int b = blockIdx.z, h0 = blockIdx.y * kBwdHeads;
  for (int i = threadIdx.x; i < kBwdHeads * N * P / 4; i += kBlock) {
    const int hd = i / (N * P / 4), r = i % (N * P / 4);
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
    if (h0 + hd < H) {
      u = *reinterpret_cast<const float4*>(kv + ((int64_t)b * H + h0 + hd) * (N * P) + r * 4);
      v = *reinterpret_cast<const float4*>(dkv + ((int64_t)b * H + h0 + hd) * (N * P) + r * 4);
    }
    *reinterpret_cast<float4*>(sS + hd * SST + r * 4) = u;
    *reinterpret_cast<float4*>(sD + hd * SST + r * 4) = v;
  }
  if (threadIdx.x < kBwdHeads) {
    const int h = h0 + threadIdx.x;
    const bool ok = h < H;
    sc[threadIdx.x][0] = ok ? __expf(A_log[h * p_hs]) : 0.f;
    sc[threadIdx.x][1] = ok ? dt_bias[h * p_hs] : 0.f;
    sc[threadIdx.x][2] = ok ? D[h * p_hs] : 0.f;
  }
  __syncthreads();
  const int g = threadIdx.x & (G - 1), slot = threadIdx.x / G;
  constexpr int slots = kBlock / G;
  const int64_t l0 = (int64_t)blockIdx.x * tok_per_block;
  const int64_t l1 = (l0 + tok_per_block < L) ? l0 + tok_per_block : L;
#pragma unroll
  for (int j = 0; j < HPL * 3; ++j) sacc[j][threadIdx.x] = 0.f;
  for (int64_t l = l0 + slot; l < l1; l += slots) {
    const int64_t row = (int64_t)b * L + l;
    float kb[N], dBv[N], dCv[N];
    load_vec<T, N>(Bm + row * ldb + g * N, kb);
#pragma unroll
    for (int n = 0; n < N; ++n) dBv[n] = dCv[n] = 0.f;
#pragma unroll 1
    for (int j = 0; j < HPL; ++j) {   // not unrolled: keeps the live state at one head (the accumulators above live in LDS for that)
      const int hl = g + G * j, h = h0 + hl;
      if (h >= H) continue;
      float v[P], gy[P], t[P], o[P];
      load_vec<T, P>(x + row * ldx + (int64_t)h * P, v);
      load_vec<T, P>(dy + row * lddy + (int64_t)h * P, gy);
      const float a = sc[hl][0], z = Io<T>::ld(dt_raw + row * lddt + h * dt_hs) + sc[hl][1], Dh = sc[hl][2];
      const float w = softplusf_(z) * a;
      const float* S = sS + hl * SST;
      const float* Dk = sD + hl * SST;
      float dd = 0.f, dw = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        t[p] = 0.f;
        dd = fmaf(gy[p], v[p], dd);
      }
#pragma unroll
      for (int n = 0; n < N; ++n) {
        float sv[P], dv[P];
#pragma unroll
        for (int p = 0; p < P; p += 4) {
          const float4 s4 = *reinterpret_cast<const float4*>(S + n * P + p), d4 = *reinterpret_cast<const float4*>(Dk + n * P + p);
          sv[p] = s4.x; sv[p + 1] = s4.y; sv[p + 2] = s4.z; sv[p + 3] = s4.w;
          dv[p] = d4.x; dv[p + 1] = d4.y; dv[p + 2] = d4.z; dv[p + 3] = d4.w;
        }
        float cb = 0.f, cc = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          t[p] = fmaf(kb[n], dv[p], t[p]);
          cb = fmaf(dv[p], v[p], cb);
          cc = fmaf(gy[p], sv[p], cc);
        }
        dBv[n] = fmaf(cb, w, dBv[n]);
        dCv[n] += cc;
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        o[p] = fmaf(w, t[p], Dh * gy[p]);
        dw = fmaf(t[p], v[p], dw);
      }
      store_vec<T, P>(dx + row * lddx + (int64_t)h * P, o);
      const float dz = dw * a * sigmoidf_(z);
      Io<T>::st(ddt_raw + row * ldddt + h * dt_hs, dz);
      sacc[j * 3][threadIdx.x] += dd;
      sacc[j * 3 + 1][threadIdx.x] += dz;
      sacc[j * 3 + 2][threadIdx.x] += dw * w;
    }
    if (nhb == 1) {
      store_vec<T, N>(dBm + row * lddb + g * N, dBv);
      store_vec<T, N>(dCm + row * lddc + g * N, dCv);
    } else {
      float* dst = bcpart + (((int64_t)blockIdx.y * gridDim.z + b) * L + l) * (2 * G * N);
#pragma unroll
      for (int n = 0; n < N; n += 4) {
        *reinterpret_cast<float4*>(dst + g * N + n) = make_float4(dBv[n], dBv[n + 1], dBv[n + 2], dBv[n + 3]);
        *reinterpret_cast<float4*>(dst + G * N + g * N + n) = make_float4(dCv[n], dCv[n + 1], dCv[n + 2], dCv[n + 3]);
      }
    }
  }
  // per-head statistics: sum over the token slots of the wave (lanes of equal g), then over the waves through LDS
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = 0; j < HPL; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float v = wave_sum_from(sacc[j * 3 + k][threadIdx.x], G);
      if (lane < G) sred[wave][(g + G * j) * 3 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < kBwdHeads * 3) {
    const int hl = threadIdx.x / 3, k = threadIdx.x % 3, h = h0 + hl;
    if (h < H) hpart[(((int64_t)b * nblk + blockIdx.x) * 3 + k) * H + h] = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
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
  const Geo g = make_geo(L, H, B);
  Ws w;
  int64_t off = 0;
  auto take = [&](int64_t nfloat) {
    float* p = ws ? (float*)((char*)ws + off) : nullptr;
    off += adnm_align(nfloat * 4, 256);
    return p;
  };
  w.part = take(B * g.nchunk * H * N * P);
  w.dkv = take(B * H * N * P);
  const BwdGeo bg = make_bwd_geo(B, L, H, G);
  w.hpart = take(B * bg.nblk * 3 * H);
  w.bcpart = take(bg.nhb > 1 ? (int64_t)bg.nhb * B * L * 2 * G * N : 0);
  w.bytes = off;
  return w;
}

int check_common(const char* who, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N, int64_t G, int dtype) {
  ADNM_REQUIRE(B > 0 && L > 0 && H > 0, "%s: empty shape B=%lld L=%lld H=%lld", who, (long long)B, (long long)L, (long long)H);
  ADNM_REQUIRE((P == 4 && (N == 8 || N == 16)) || (P == 8 && N == 8),
               "%s: (headdim P=%lld, states-per-group N=%lld) not in {(4,8),(4,16),(8,8)}", who, (long long)P, (long long)N);
  ADNM_REQUIRE(G == 1 || G == 2 || G == 4, "%s: groups G=%lld not in {1,2,4}", who, (long long)G);
  ADNM_REQUIRE(B <= 65535 && adnm_cdiv(H, 64) <= 65535, "%s: grid too large", who);
  ADNM_REQUIRE(dtype == ADNM_F32 || dtype == ADNM_BF16, "%s: bad dtype %d", who, dtype);
  return ADNM_OK;
}

// supported (headdim, state) pairs: N*P <= 64 keeps the per-lane state in registers
#define DISPATCH_PN(FN, ...)                                \
  do {                                                      \
    if (P == 4 && N == 8) FN<T, 4, 8>(__VA_ARGS__);         \
    else if (P == 4 && N == 16) FN<T, 4, 16>(__VA_ARGS__);  \
    else FN<T, 8, 8>(__VA_ARGS__);                          \
  } while (0)

template <typename T, int P, int N>
void run_outer(bool weighted, const void* V, int64_t ldv, const void* K, int64_t ldk, const void* dt_raw, int64_t lddt,
               int64_t dt_hs, const float* dt_bias, const float* A_log, int64_t p_hs, float* part, float* out, int64_t B,
               int64_t L, int64_t H, int64_t G, hipStream_t st) {
  const Geo g = make_geo(L, H, B);
  const dim3 grid(g.nchunk, g.nhb, (unsigned)B);
  const size_t smem = (size_t)g.hb * N * P * sizeof(float);
  if (weighted)
    { ADNM_PROF("ssd_outer_reduce", st, (double)sizeof(T) * B * L * (H * P + G * N + H)); ssd_outer_reduce_kernel<T, P, N, true><<<grid, kBlock, smem, st>>>((const T*)V, ldv, (const T*)K, ldk, (const T*)dt_raw, lddt,
                                                                       dt_hs, dt_bias, A_log, p_hs, part, L, (int)H, (int)G,
                                                                       g.hb, g.tok1, g.nchunk); }
  else
    { ADNM_PROF("ssd_outer_reduce", st, (double)sizeof(T) * B * L * (H * P + G * N)); ssd_outer_reduce_kernel<T, P, N, false><<<grid, kBlock, smem, st>>>((const T*)V, ldv, (const T*)K, ldk, nullptr, 0, 0, nullptr,
                                                                        nullptr, 0, part, L, (int)H, (int)G, g.hb, g.tok1,
                                                                        g.nchunk); }
  const int64_t E = H * N * P;
  { ADNM_PROF("ssd_fold", st, 4.0 * B * E * (g.nchunk + 1)); ssd_fold_kernel<<<dim3((unsigned)adnm_cdiv(E, 64), (unsigned)B), 256, 0, st>>>(part, out, g.nchunk, E); }
}

template <typename T, int P, int N>
void run_apply(const void* x, int64_t ldx, const void* Cm, int64_t ldc, const float* D, int64_t p_hs, const float* kv, void* y,
               int64_t ldy, int64_t B, int64_t L, int64_t H, int64_t G, hipStream_t st) {
  const Geo g = make_geo(L, H, B);
  { ADNM_PROF("ssd_apply", st, (double)sizeof(T) * B * L * (2 * H * P + G * N)); ssd_apply_kernel<T, P, N><<<dim3(g.nblk2, g.nhb, (unsigned)B), kBlock, 0, st>>>((const T*)x, ldx, (const T*)Cm, ldc, D, p_hs, kv,
                                                                                  (T*)y, ldy, L, (int)H, (int)G, g.hb, g.tok2); }
}

template <typename T, int P, int N>
void run_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* Bm, int64_t ldb, const void* Cm, int64_t ldc,
             const void* dt_raw, int64_t lddt, int64_t dt_hs, const float* dt_bias, const float* A_log, const float* D,
             int64_t p_hs, const float* kv, const float* dkv, void* dx, int64_t lddx, void* dBm, int64_t lddb, void* dCm,
             int64_t lddc, void* ddt_raw, int64_t ldddt, float* hpart, float* bcpart, float* ddt_bias, float* dA_log, float* dD,
             int64_t B, int64_t L, int64_t H, int64_t G, hipStream_t st) {
  (void)Cm;
  (void)ldc;
  const BwdGeo g = make_bwd_geo(B, L, H, G);
  const dim3 grid(g.nblk, g.nhb, (unsigned)B);
#define ADNM_SSD_BWD(GG)                                                                                                              \
  ssd_bwd_kernel<T, P, N, GG><<<grid, kBlock, 0, st>>>((const T*)dy, lddy, (const T*)x, ldx, (const T*)Bm, ldb, (const T*)dt_raw, lddt, \
                                                       dt_hs, dt_bias, A_log, D, p_hs, kv, dkv, (T*)dx, lddx, (T*)dBm, lddb, (T*)dCm,   \
                                                       lddc, (T*)ddt_raw, ldddt, hpart, bcpart, L, (int)H, g.tok, g.nblk, g.nhb)
  {
    ADNM_PROF("ssd_bwd", st, (double)sizeof(T) * B * L * (3 * H * P + 4 * G * N + 2 * H));
    if (G == 1) ADNM_SSD_BWD(1);
    else if (G == 2) ADNM_SSD_BWD(2);
    else ADNM_SSD_BWD(4);
  }
#undef ADNM_SSD_BWD
  if (g.nhb > 1) {
    const int64_t tot = B * L * 2 * G * N;
    { ADNM_PROF("ssd_bc_fold", st, 4.0 * tot * g.nhb); ssd_bc_fold_kernel<T><<<(unsigned)adnm_cdiv(tot, 256), 256, 0, st>>>(bcpart, g.nhb, B * L, (int)(G * N), (T*)dBm, lddb, (T*)dCm,
                                                                         lddc); }
  }
  adnm_launch_fold("ssd_head_fold", hpart, (int)(B * g.nblk), 3 * (int)H, {dD, (int)H}, {ddt_bias, (int)H}, {dA_log, (int)H}, {nullptr, 0}, st);
}

}  // namespace

extern "C" int64_t adnm_ssd_ws_bytes(int64_t B, int64_t L, int64_t H, int64_t P, int64_t N, int64_t G) {
  if (B <= 0 || L <= 0 || H <= 0) return 0;
  return carve(nullptr, B, L, H, P, N, G).bytes;
}

extern "C" int adnm_ssd_reduce_fwd(const void* x, int64_t ldx, const void* Bm, int64_t ldb, const void* Cm, int64_t ldc,
                                   const void* dt_raw, int64_t lddt, int64_t dt_hstride, const float* dt_bias,
                                   const float* A_log, const float* D, int64_t p_hstride, void* y, int64_t ldy, float* kv,
                                   void* ws, int64_t ws_bytes, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N,
                                   int64_t G, int dtype, adnm_stream_t stream) {
  if (int rc = check_common("ssd_reduce_fwd", B, L, H, P, N, G, dtype)) return rc;
  ADNM_REQUIRE(x && Bm && Cm && dt_raw && dt_bias && A_log && D && y && kv, "ssd_reduce_fwd: null pointer");
  ADNM_REQUIRE(ldx >= H * P && ldy >= H * P && ldb >= G * N && ldc >= G * N && lddt >= (H - 1) * dt_hstride + 1,
               "ssd_reduce_fwd: row strides smaller than the rows they address");
  ADNM_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0, "ssd_reduce_fwd: row strides must be multiples of 4");
  const Ws w = carve(ws, B, L, H, P, N, G);
  if (!ws || ws_bytes < w.bytes) {
    adnm_set_error("ssd_reduce_fwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)w.bytes);
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ADNM_F32) {
    using T = float;
    DISPATCH_PN(run_outer, true, x, ldx, Bm, ldb, dt_raw, lddt, dt_hstride, dt_bias, A_log, p_hstride, w.part, kv, B, L, H, G, st);
    DISPATCH_PN(run_apply, x, ldx, Cm, ldc, D, p_hstride, kv, y, ldy, B, L, H, G, st);
  } else {
    using T = uint16_t;
    DISPATCH_PN(run_outer, true, x, ldx, Bm, ldb, dt_raw, lddt, dt_hstride, dt_bias, A_log, p_hstride, w.part, kv, B, L, H, G, st);
    DISPATCH_PN(run_apply, x, ldx, Cm, ldc, D, p_hstride, kv, y, ldy, B, L, H, G, st);
  }
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
  if (dtype == ADNM_F32) {
    using T = float;
    DISPATCH_PN(run_outer, false, dy, lddy, Cm, ldc, nullptr, 0, 0, nullptr, nullptr, 0, w.part, w.dkv, B, L, H, G, st);
    DISPATCH_PN(run_bwd, dy, lddy, x, ldx, Bm, ldb, Cm, ldc, dt_raw, lddt, dt_hstride, dt_bias, A_log, D, p_hstride, kv, w.dkv, dx,
                lddx, dBm, lddb, dCm, lddc, ddt_raw, ldddt, w.hpart, w.bcpart, ddt_bias, dA_log, dD, B, L, H, G, st);
  } else {
    using T = uint16_t;
    DISPATCH_PN(run_outer, false, dy, lddy, Cm, ldc, nullptr, 0, 0, nullptr, nullptr, 0, w.part, w.dkv, B, L, H, G, st);
    DISPATCH_PN(run_bwd, dy, lddy, x, ldx, Bm, ldb, Cm, ldc, dt_raw, lddt, dt_hstride, dt_bias, A_log, D, p_hstride, kv, w.dkv, dx,
                lddx, dBm, lddb, dCm, lddc, ddt_raw, ldddt, w.hpart, w.bcpart, ddt_bias, dA_log, dD, B, L, H, G, st);
  }
  ADNM_CHECK_LAUNCH("ssd_reduce_bwd");
  return ADNM_OK;
}
