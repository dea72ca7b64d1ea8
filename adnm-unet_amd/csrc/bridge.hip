// The heads of Channel_Att_Bridge (reference models/model_untils.py:744-750, :594-613), grouped: after the global average pool and the
// Conv1d over the concatenated channel axis, every skip i gets
//     gate_i[b, :] = IntensityGate( att[b, :] . W_i^T + bias_i )        att: (B, S), S = sum of the skips' channels (2144), W_i: (C_i, S)
// i.e. up to seven nn.Linear + silu(enhance * (z - threshold)) pairs on M = B rows (4 at config 2): weight-streaming GEMVs.  As separate
// short GEMMs they cost, per head, a forward launch, a gate launch, two gradient GEMMs, the gate's backward and its fold — ~25 launches
// for the three live heads, each a few microseconds of work.  Here: ONE forward launch for all heads (a wave per output feature, the B
// rows of att in LDS, the weight row streamed once with 16-byte loads) and TWO backward launches (a workgroup per feature range: the
// weight-gradient rows are written as they are formed, the att gradient is accumulated in registers over the range and folded over the
// workgroups; the shared enhance / threshold gradients through the same fold).  HBM-bound: 4 * sum(C_i) * S bytes each way.
#include "adnm_common.h"

namespace {
constexpr int kMaxHeads = 8, kMaxB = 8;
constexpr int kFwdThreads = 256, kBwdThreads = 256, kColsPerThread = 3;   // backward: a thread owns float4 columns t, t+256, t+512

struct Heads {
  const float* W[kMaxHeads];
  const float* bias[kMaxHeads];
  float* z[kMaxHeads];          // pre-activation (B, C_i), saved for backward
  float* y[kMaxHeads];          // gate (B, C_i)                                  (backward: dy)
  float* dW[kMaxHeads];         // backward only
  float* dbias[kMaxHeads];
  int cend[kMaxHeads];          // exclusive prefix of the head widths
  int n;
};

__device__ __forceinline__ void locate(const Heads& h, int f, int& head, int& local) {
  head = 0;
  while (head + 1 < h.n && f >= h.cend[head]) ++head;
  local = f - (head ? h.cend[head - 1] : 0);
}

// one wave per output feature; att (B x S) staged in LDS once per workgroup.  BMAX: compile-time bound of B (registers)
template <int BMAX>
__global__ __launch_bounds__(kFwdThreads) void bridge_heads_fwd_kernel(const float* __restrict__ att, Heads h, const float* __restrict__ enh,
                                                                       const float* __restrict__ thr, int B, int S, int per_wave) {
  extern __shared__ __attribute__((aligned(16))) float satt[];   // [B][S]
  for (int i = threadIdx.x; i < B * S / 4; i += kFwdThreads) reinterpret_cast<float4*>(satt)[i] = reinterpret_cast<const float4*>(att)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int total = h.cend[h.n - 1], S4 = S >> 2;
  const float a = *enh, t = *thr;
  const int f0 = (blockIdx.x * (kFwdThreads / 64) + wave) * per_wave;
  for (int f = f0; f < f0 + per_wave && f < total; ++f) {
    int head, local;
    locate(h, f, head, local);
    const int C = h.cend[head] - (head ? h.cend[head - 1] : 0);
    const float4* wrow = reinterpret_cast<const float4*>(h.W[head] + (int64_t)local * S);
    float acc[BMAX];
#pragma unroll
    for (int b = 0; b < BMAX; ++b) acc[b] = 0.f;
    for (int c = lane; c < S4; c += 64) {
      const float4 w = wrow[c];
#pragma unroll
      for (int b = 0; b < BMAX; ++b)
        if (b < B) {
          const float4 x = reinterpret_cast<const float4*>(satt + b * S)[c];
          acc[b] = fmaf(w.x, x.x, fmaf(w.y, x.y, fmaf(w.z, x.z, fmaf(w.w, x.w, acc[b]))));
        }
    }
#pragma unroll
    for (int b = 0; b < BMAX; ++b)
      if (b < B) acc[b] = wave_sum(acc[b]);
    if (lane == 0) {
      const float bv = h.bias[head] ? h.bias[head][local] : 0.f;
      for (int b = 0; b < B; ++b) {
        const float z = acc[b] + bv;
        h.z[head][(int64_t)b * C + local] = z;
        h.y[head][(int64_t)b * C + local] = siluf_(a * (z - t));
      }
    }
  }
}

// one workgroup per range of `per_wg` output features.  Per feature: dz[b] = dy[b] * silu'(enh (z[b] - thr)) * enh (every thread, from
// broadcast loads), dW row = sum_b dz[b] * att[b, :] (written), datt[b, :] += dz[b] * W row (registers), dbias = sum_b dz[b].
// Partial row of the workgroup: [B * S of datt | d enhance | d threshold | pad, pad].
template <int BMAX>
__global__ __launch_bounds__(kBwdThreads) void bridge_heads_bwd_kernel(const float* __restrict__ att, Heads h, const float* __restrict__ enh,
                                                                       const float* __restrict__ thr, float* __restrict__ part, int B, int S,
                                                                       int per_wg) {
  const int total = h.cend[h.n - 1], S4 = S >> 2;
  const float a = *enh, t = *thr;
  float4 xa[BMAX][kColsPerThread], da[BMAX][kColsPerThread];
  int col[kColsPerThread];
#pragma unroll
  for (int u = 0; u < kColsPerThread; ++u) {
    col[u] = threadIdx.x + u * kBwdThreads;
#pragma unroll
    for (int b = 0; b < BMAX; ++b) {
      da[b][u] = make_float4(0.f, 0.f, 0.f, 0.f);
      xa[b][u] = (b < B && col[u] < S4) ? reinterpret_cast<const float4*>(att + (int64_t)b * S)[col[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float s_e = 0.f, s_t = 0.f;
  const int f0 = blockIdx.x * per_wg;
  for (int f = f0; f < f0 + per_wg && f < total; ++f) {
    int head, local;
    locate(h, f, head, local);
    const int C = h.cend[head] - (head ? h.cend[head - 1] : 0);
    float dz[BMAX], dbs = 0.f;
#pragma unroll
    for (int b = 0; b < BMAX; ++b) {
      dz[b] = 0.f;
      if (b < B) {
        const float zz = h.z[head][(int64_t)b * C + local], g = h.y[head][(int64_t)b * C + local];   // (y holds dy here)
        const float d = zz - t, gp = g * silu_gradf_(a * d);
        dz[b] = gp * a;
        dbs += dz[b];
        s_e = fmaf(gp, d, s_e);   // every thread computes the same scalars; thread 0's copy is the one that is stored
        s_t += gp;
      }
    }
    const float4* wrow = reinterpret_cast<const float4*>(h.W[head] + (int64_t)local * S);
    float4* dwrow = reinterpret_cast<float4*>(h.dW[head] + (int64_t)local * S);
#pragma unroll
    for (int u = 0; u < kColsPerThread; ++u) {
      if (col[u] >= S4) continue;
      const float4 w = wrow[col[u]];
      float4 dw = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int b = 0; b < BMAX; ++b)
        if (b < B) {
          dw.x = fmaf(dz[b], xa[b][u].x, dw.x), dw.y = fmaf(dz[b], xa[b][u].y, dw.y), dw.z = fmaf(dz[b], xa[b][u].z, dw.z), dw.w = fmaf(dz[b], xa[b][u].w, dw.w);
          da[b][u].x = fmaf(dz[b], w.x, da[b][u].x), da[b][u].y = fmaf(dz[b], w.y, da[b][u].y), da[b][u].z = fmaf(dz[b], w.z, da[b][u].z),
          da[b][u].w = fmaf(dz[b], w.w, da[b][u].w);
        }
      dwrow[col[u]] = dw;
    }
    if (threadIdx.x == 0 && h.dbias[head]) h.dbias[head][local] = dbs;
  }
  float* row = part + (int64_t)blockIdx.x * ((int64_t)B * S + 4);
#pragma unroll
  for (int u = 0; u < kColsPerThread; ++u) {
    if (col[u] >= S4) continue;
#pragma unroll
    for (int b = 0; b < BMAX; ++b)
      if (b < B) reinterpret_cast<float4*>(row + (int64_t)b * S)[col[u]] = da[b][u];
  }
  if (threadIdx.x == 0) {
    float* sc = row + (int64_t)B * S;
    sc[0] = s_e, sc[1] = -a * s_t, sc[2] = 0.f, sc[3] = 0.f;
  }
}

int load_heads(const char* who, int nheads, const float* const* W, const float* const* bias, const int64_t* C, Heads* h) {
  ADNM_REQUIRE(nheads >= 1 && nheads <= kMaxHeads && W && C, "%s: 1..%d heads", who, kMaxHeads);
  int end = 0;
  for (int i = 0; i < nheads; ++i) {
    ADNM_REQUIRE(W[i] && C[i] > 0 && C[i] < (1 << 20), "%s: head %d has no weight / a bad width", who, i);
    end += (int)C[i];
    h->W[i] = W[i], h->bias[i] = bias ? bias[i] : nullptr, h->cend[i] = end;
    h->z[i] = h->y[i] = h->dW[i] = h->dbias[i] = nullptr;
  }
  for (int i = nheads; i < kMaxHeads; ++i) h->W[i] = h->bias[i] = nullptr, h->z[i] = h->y[i] = h->dW[i] = h->dbias[i] = nullptr, h->cend[i] = end;
  h->n = nheads;
  return ADNM_OK;
}
int bwd_blocks(int total) {
  int per = (total + 511) / 512;   // ~512 workgroups: two per CU
  if (per < 1) per = 1;
  return per;
}
}  // namespace

extern "C" int adnm_bridge_heads_fwd(const float* att, const float* const* W, const float* const* bias, const int64_t* C, int nheads,
                                     const float* enhance, const float* threshold, float* const* z, float* const* y, int64_t B, int64_t S,
                                     adnm_stream_t stream) {
  ADNM_REQUIRE(att && enhance && threshold && z && y, "bridge_heads_fwd: null pointer");
  ADNM_REQUIRE(B >= 1 && B <= kMaxB && S >= 4 && S % 4 == 0 && S <= 4 * kBwdThreads * kColsPerThread && B * S * 4 <= 160 * 1024,
               "bridge_heads_fwd: needs 1 <= B <= %d sample rows and 4 | S <= %d pooled channels, got B=%lld S=%lld", kMaxB,
               4 * kBwdThreads * kColsPerThread, (long long)B, (long long)S);
  Heads h;
  if (int rc = load_heads("bridge_heads_fwd", nheads, W, bias, C, &h)) return rc;
  for (int i = 0; i < nheads; ++i) {
    ADNM_REQUIRE(z[i] && y[i], "bridge_heads_fwd: head %d has no output", i);
    h.z[i] = z[i], h.y[i] = y[i];
  }
  const int total = h.cend[nheads - 1];
  hipStream_t st = (hipStream_t)stream;
  const int per_wave = 1;
  const size_t smem = (size_t)B * S * sizeof(float);
  const unsigned grid = (unsigned)adnm_cdiv(total, (kFwdThreads / 64) * per_wave);
  ADNM_PROF("bridge_heads_fwd", st, 4.0 * ((double)total * S + (double)B * (S + 2.0 * total)));
  if (B <= 4) {
    ADNM_ALLOW_LDS(bridge_heads_fwd_kernel<4>, smem, "bridge_heads_fwd");
    bridge_heads_fwd_kernel<4><<<grid, kFwdThreads, smem, st>>>(att, h, enhance, threshold, (int)B, (int)S, per_wave);
  } else {
    ADNM_ALLOW_LDS(bridge_heads_fwd_kernel<kMaxB>, smem, "bridge_heads_fwd");
    bridge_heads_fwd_kernel<kMaxB><<<grid, kFwdThreads, smem, st>>>(att, h, enhance, threshold, (int)B, (int)S, per_wave);
  }
  ADNM_CHECK_LAUNCH("bridge_heads_fwd");
  return ADNM_OK;
}

extern "C" int64_t adnm_bridge_heads_bwd_ws_bytes(int64_t total, int64_t B, int64_t S) {
  if (total < 1 || B < 1 || S < 4) return -1;
  return adnm_cdiv(total, bwd_blocks((int)total)) * (B * S + 4) * (int64_t)sizeof(float);
}

// dy[i]: (B, C_i) gate gradients; z[i]: the saved pre-activations.  Outputs (all OVERWRITTEN): datt (B, S); dW[i] (C_i, S), dbias[i] (C_i)
// (or NULL); denhance, dthreshold (1 each).  datt / denhance / dthreshold come out of ONE fold of the per-workgroup partial rows.
extern "C" int adnm_bridge_heads_bwd(const float* att, const float* const* W, const int64_t* C, int nheads, const float* enhance, const float* threshold,
                                     const float* const* z, const float* const* dy, float* datt, float* const* dW, float* const* dbias,
                                     float* denhance, float* dthreshold, void* ws, int64_t ws_bytes, int64_t B, int64_t S, adnm_stream_t stream) {
  ADNM_REQUIRE(att && enhance && threshold && z && dy && datt && dW && denhance && dthreshold, "bridge_heads_bwd: null pointer");
  ADNM_REQUIRE(B >= 1 && B <= kMaxB && S >= 4 && S % 4 == 0 && S <= 4 * kBwdThreads * kColsPerThread,
               "bridge_heads_bwd: needs 1 <= B <= %d sample rows and 4 | S <= %d pooled channels, got B=%lld S=%lld", kMaxB,
               4 * kBwdThreads * kColsPerThread, (long long)B, (long long)S);
  Heads h;
  if (int rc = load_heads("bridge_heads_bwd", nheads, W, nullptr, C, &h)) return rc;
  for (int i = 0; i < nheads; ++i) {
    ADNM_REQUIRE(z[i] && dy[i] && dW[i], "bridge_heads_bwd: head %d misses a tensor", i);
    h.z[i] = const_cast<float*>(z[i]), h.y[i] = const_cast<float*>(dy[i]), h.dW[i] = dW[i], h.dbias[i] = dbias ? dbias[i] : nullptr;
  }
  const int total = h.cend[nheads - 1], per_wg = bwd_blocks(total);
  const int64_t blocks = adnm_cdiv(total, per_wg), rowlen = B * S + 4;
  if (!ws || ws_bytes < blocks * rowlen * 4) {
    adnm_set_error("bridge_heads_bwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)(blocks * rowlen * 4));
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  {
    ADNM_PROF("bridge_heads_bwd", st, 4.0 * (2.0 * total * S + (double)B * (2.0 * S + 2.0 * total)));
    if (B <= 4) bridge_heads_bwd_kernel<4><<<(unsigned)blocks, kBwdThreads, 0, st>>>(att, h, enhance, threshold, (float*)ws, (int)B, (int)S, per_wg);
    else bridge_heads_bwd_kernel<kMaxB><<<(unsigned)blocks, kBwdThreads, 0, st>>>(att, h, enhance, threshold, (float*)ws, (int)B, (int)S, per_wg);
  }
  ADNM_CHECK_LAUNCH("bridge_heads_bwd");
  // the att gradient is read by the Conv1d's backward right away: this fold is never deferred (the caller binds no queue around this call)
  adnm_launch_fold("bridge_heads_fold", (const float*)ws, (int)blocks, (int)rowlen, {datt, (int)(B * S)}, {denhance, 1}, {dthreshold, 1}, {nullptr, 2}, st);
  ADNM_CHECK_LAUNCH("bridge_heads_bwd");
  return ADNM_OK;
}

// ---- the bridge's global average pools, grouped (Channel_Att_Bridge.forward, model_untils.py:570-590 of the reference: avgpool of each of
// the 7 skips, concatenated along the channel axis).  Before: one pool launch + one fold per skip + a torch.cat — 11 launches of a few
// microseconds for ~20 MB of reads.  Here: ONE launch over all skips writes per-slice partial rows straight into the concatenated layout
// (part[slice][b][off_k + c], already scaled by 1 / L_k), one fold sums the slices into att (B, S).  Backward likewise one launch:
// dx_k[b,l,c] = dxa_k[b,l,c] + datt[b, off_k + c] / L_k (dxa_k: the gradient of the skip's other consumers, or NULL).
namespace {
constexpr int kMaxPool = 8, kPoolSlices = 64, kPoolThreads = 256;
struct PoolDesc {
  const float* x;     // fwd: (B, L, C) contiguous;  bwd: dxa or NULL
  float* dx;          // bwd only
  int L, C, off, ql;  // ql: channel-quad lanes of a workgroup (8 or 16); 256 / ql row lanes
};
struct MultiPool {
  int count, B, S;
  int blk_end[kMaxPool];
  PoolDesc d[kMaxPool];
};

// workgroup = ql channel quads x (256 / ql) row lanes of ONE (skip, sample, slice); 4 rows in flight per lane
__global__ __launch_bounds__(kPoolThreads) void bridge_pool_fwd_kernel(MultiPool mp_by_value, float* __restrict__ part) {
  __shared__ float4 sm[kPoolThreads];
  (void)mp_by_value;
  const __attribute__((address_space(4))) MultiPool& mp = *(const __attribute__((address_space(4))) MultiPool*)__builtin_amdgcn_kernarg_segment_ptr();
  int k = 0;
  while (k + 1 < mp.count && (int)blockIdx.x >= mp.blk_end[k]) ++k;
  const int local = (int)blockIdx.x - (k ? mp.blk_end[k - 1] : 0);
  const int L = mp.d[k].L, C = mp.d[k].C, ql = mp.d[k].ql, C4 = C >> 2;
  const int qblocks = (C4 + ql - 1) / ql;
  const int qb = local % qblocks, s = (local / qblocks) % kPoolSlices, b = local / (qblocks * kPoolSlices);
  const int lq = threadIdx.x % ql, rl = threadIdx.x / ql, rls = kPoolThreads / ql;
  const int q = qb * ql + lq;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < C4) {
    const float* xb = mp.d[k].x + (int64_t)b * L * C + q * 4;
    const int step = rls * kPoolSlices;
    int r = s * rls + rl;
    for (; r + 3 * step < L; r += 4 * step) {
      const float4 v0 = *reinterpret_cast<const float4*>(xb + (int64_t)r * C), v1 = *reinterpret_cast<const float4*>(xb + (int64_t)(r + step) * C);
      const float4 v2 = *reinterpret_cast<const float4*>(xb + (int64_t)(r + 2 * step) * C), v3 = *reinterpret_cast<const float4*>(xb + (int64_t)(r + 3 * step) * C);
      acc.x += (v0.x + v1.x) + (v2.x + v3.x), acc.y += (v0.y + v1.y) + (v2.y + v3.y), acc.z += (v0.z + v1.z) + (v2.z + v3.z), acc.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; r < L; r += step) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)r * C);
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0 && q < C4) {
    for (int j = 1; j < rls; ++j) {   // fixed order
      const float4 v = sm[j * ql + lq];
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    const float inv_l = 1.0f / (float)L;
    *reinterpret_cast<float4*>(part + ((int64_t)s * mp.B + b) * mp.S + mp.d[k].off + q * 4) = make_float4(acc.x * inv_l, acc.y * inv_l, acc.z * inv_l, acc.w * inv_l);
  }
}

__global__ __launch_bounds__(kPoolThreads) void bridge_pool_bwd_kernel(MultiPool mp_by_value, const float* __restrict__ datt) {
  (void)mp_by_value;
  const __attribute__((address_space(4))) MultiPool& mp = *(const __attribute__((address_space(4))) MultiPool*)__builtin_amdgcn_kernarg_segment_ptr();
  int k = 0;
  while (k + 1 < mp.count && (int)blockIdx.x >= mp.blk_end[k]) ++k;
  const int first = k ? mp.blk_end[k - 1] : 0, nblk = mp.blk_end[k] - first;
  const int L = mp.d[k].L, C4 = mp.d[k].C >> 2;
  const float* __restrict__ dxa = mp.d[k].x;
  float* __restrict__ dx = mp.d[k].dx;
  const float* dm = datt + mp.d[k].off;
  const float inv_l = 1.0f / (float)L;
  const int64_t total4 = (int64_t)mp.B * L * C4;
  for (int64_t i = (int64_t)((int)blockIdx.x - first) * kPoolThreads + threadIdx.x; i < total4; i += (int64_t)nblk * kPoolThreads) {
    const int q = (int)(i % C4);
    const int64_t b = i / ((int64_t)C4 * L);
    const float4 m = *reinterpret_cast<const float4*>(dm + b * mp.S + q * 4);
    float4 v = dxa ? *reinterpret_cast<const float4*>(dxa + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = fmaf(m.x, inv_l, v.x), v.y = fmaf(m.y, inv_l, v.y), v.z = fmaf(m.z, inv_l, v.z), v.w = fmaf(m.w, inv_l, v.w);
    *reinterpret_cast<float4*>(dx + i * 4) = v;
  }
}

int load_pools(const char* who, int n, const int64_t* L, const int64_t* C, int64_t B, MultiPool* mp) {
  ADNM_REQUIRE(n >= 1 && n <= kMaxPool && L && C, "%s: 1 .. %d skips", who, kMaxPool);
  ADNM_REQUIRE(B >= 1 && B <= 65535, "%s: bad B=%lld", who, (long long)B);
  int off = 0;
  for (int k = 0; k < n; ++k) {
    ADNM_REQUIRE(L[k] > 0 && C[k] >= 4 && C[k] % 4 == 0 && B * L[k] * C[k] < (1ll << 31), "%s: skip %d: bad shape L=%lld C=%lld (4 | C)", who, k,
                 (long long)L[k], (long long)C[k]);
    mp->d[k].L = (int)L[k], mp->d[k].C = (int)C[k], mp->d[k].off = off, mp->d[k].ql = C[k] >= 64 ? 16 : 8;
    off += (int)C[k];
  }
  mp->count = n, mp->B = (int)B, mp->S = off;
  return ADNM_OK;
}
}  // namespace

extern "C" int64_t adnm_bridge_pool_ws_bytes(int64_t B, int64_t S) { return B < 1 || S < 4 ? -1 : (int64_t)kPoolSlices * B * S * (int64_t)sizeof(float); }

// x[k]: (B, L[k], C[k]) contiguous fp32; att: (B, S = sum C) OVERWRITTEN with the token means, skip k at columns [off_k, off_k + C[k]).
extern "C" int adnm_bridge_pool_fwd(int64_t n, const float* const* x, const int64_t* L, const int64_t* C, float* att, void* ws, int64_t ws_bytes, int64_t B,
                                    adnm_stream_t stream) {
  ADNM_REQUIRE(x && att, "bridge_pool_fwd: null pointer");
  MultiPool mp;
  if (int rc = load_pools("bridge_pool_fwd", (int)n, L, C, B, &mp)) return rc;
  if (!ws || ws_bytes < adnm_bridge_pool_ws_bytes(B, mp.S)) {
    adnm_set_error("bridge_pool_fwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_bridge_pool_ws_bytes(B, mp.S));
    return ADNM_EWORKSPACE;
  }
  int blocks = 0;
  double bytes = 0;
  for (int k = 0; k < (int)n; ++k) {
    ADNM_REQUIRE(x[k], "bridge_pool_fwd: skip %d is NULL", k);
    mp.d[k].x = x[k], mp.d[k].dx = nullptr;
    blocks += (int)adnm_cdiv(mp.d[k].C / 4, mp.d[k].ql) * kPoolSlices * (int)B;
    mp.blk_end[k] = blocks;
    bytes += 4.0 * B * ((double)mp.d[k].L + kPoolSlices) * mp.d[k].C;
  }
  for (int k = (int)n; k < kMaxPool; ++k) mp.blk_end[k] = blocks;
  hipStream_t st = (hipStream_t)stream;
  {
    ADNM_PROF("bridge_pool_fwd", st, bytes);
    bridge_pool_fwd_kernel<<<(unsigned)blocks, kPoolThreads, 0, st>>>(mp, (float*)ws);
  }
  ADNM_CHECK_LAUNCH("bridge_pool_fwd");
  // the conv over the pooled channels reads att right away: the caller binds no fold queue around this call
  adnm_launch_fold("bridge_pool_fold", (const float*)ws, kPoolSlices, (int)(B * mp.S), {att, (int)(B * mp.S)}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("bridge_pool_fold");
  return ADNM_OK;
}

// dx[k] (B, L[k], C[k]) = dxa[k] (same shape, or NULL) + datt[:, off_k : off_k + C[k]] / L[k] broadcast over the tokens; dx[k] NULL skips a skip
extern "C" int adnm_bridge_pool_bwd(int64_t n, const float* const* dxa, const float* datt, const int64_t* L, const int64_t* C, float* const* dx, int64_t B,
                                    adnm_stream_t stream) {
  ADNM_REQUIRE(dxa && datt && dx && (reinterpret_cast<uintptr_t>(datt) & 15) == 0, "bridge_pool_bwd: null / misaligned pointer");
  MultiPool mp;
  if (int rc = load_pools("bridge_pool_bwd", (int)n, L, C, B, &mp)) return rc;
  int blocks = 0;
  double bytes = 0;
  for (int k = 0; k < (int)n; ++k) {
    mp.d[k].x = dxa[k], mp.d[k].dx = dx[k];
    if (dx[k]) {
      const int64_t need = adnm_cdiv(B * mp.d[k].L * (mp.d[k].C / 4), kPoolThreads);
      blocks += (int)(need < 1024 ? need : 1024);
      bytes += 4.0 * B * mp.d[k].L * mp.d[k].C * (dxa[k] ? 2 : 1);
    }
    mp.blk_end[k] = blocks;
  }
  for (int k = (int)n; k < kMaxPool; ++k) mp.blk_end[k] = blocks;
  if (!blocks) return ADNM_OK;
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("bridge_pool_bwd", st, bytes);
  bridge_pool_bwd_kernel<<<(unsigned)blocks, kPoolThreads, 0, st>>>(mp, datt);
  ADNM_CHECK_LAUNCH("bridge_pool_bwd");
  return ADNM_OK;
}
