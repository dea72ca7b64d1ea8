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
