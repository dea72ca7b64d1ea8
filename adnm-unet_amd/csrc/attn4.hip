// K16 — StandardAttention (models/ADNssd.py:26-47 of the reference) for its head size of 4, fused.
//   q, k, v = to_qkv(x).chunk(3);  out = softmax(q k^T * scale) v      per (batch, head), heads = dim / 4, L = 64 .. 1024 tokens
// The reference (and the library path) materialises the (B*heads, L, L) score tensor four times forward and six times backward
// (33 MB each at L = 256: bmm, scale, softmax, bmm + their autograd twins, ~145 us per block); with 4-wide heads the whole
// K / V of a head is 8 KB, so a workgroup keeps them in LDS and never writes a score.  qkv is the (B, L, 3*inner) output of
// to_qkv as it is: head h of q / k / v is the float4 at column h*4 (+ inner, + 2*inner).
//   forward : grid (B*heads, L/64), 64 queries x 4 key slices per workgroup, two passes over the slice (max, then exp-sum),
//             slices merged with in-quad shuffles; saves lse = max + log(sum) per query.
//   backward: same grid; the workgroup recomputes p = exp(s - lse) twice: once per query row (dQ), once per key row (dK, dV).
#include "adnm_common.h"

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
__device__ __forceinline__ void axpy4(float4& acc, float s, const float4& v) {
  acc.x = fmaf(s, v.x, acc.x); acc.y = fmaf(s, v.y, acc.y); acc.z = fmaf(s, v.z, acc.z); acc.w = fmaf(s, v.w, acc.w);
}
__device__ __forceinline__ float quad_add(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  return v;
}

__global__ __launch_bounds__(kBlock) void attn4_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse, int L,
                                                           int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) float4 sm[];   // K[L], V[L]
  float4* Ks = sm;
  float4* Vs = sm + L;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int inner = heads * 4, ld = 3 * inner;
  const float* base = qkv + (int64_t)b * L * ld + h * 4;
  for (int j = threadIdx.x; j < L; j += kBlock) {
    Ks[j] = *reinterpret_cast<const float4*>(base + (int64_t)j * ld + inner);
    Vs[j] = *reinterpret_cast<const float4*>(base + (int64_t)j * ld + 2 * inner);
  }
  __syncthreads();
  const int qi = blockIdx.y * 64 + (threadIdx.x >> 2), slice = threadIdx.x & 3;
  const bool live = qi < L;
  float4 q = live ? *reinterpret_cast<const float4*>(base + (int64_t)qi * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
  q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
  float m = -INFINITY;
  for (int j = slice; j < L; j += 4) m = fmaxf(m, dot4(q, Ks[j]));
  m = quad_max(m);
  float l = 0.f;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = slice; j < L; j += 4) {
    const float p = __expf(dot4(q, Ks[j]) - m);
    l += p;
    axpy4(o, p, Vs[j]);
  }
  l = quad_add(l);
  o.x = quad_add(o.x); o.y = quad_add(o.y); o.z = quad_add(o.z); o.w = quad_add(o.w);
  if (live && slice == 0) {
    const float r = 1.0f / l;
    *reinterpret_cast<float4*>(out + ((int64_t)b * L + qi) * inner + h * 4) = make_float4(o.x * r, o.y * r, o.z * r, o.w * r);
    lse[((int64_t)b * heads + h) * L + qi] = m + __logf(l);
  }
}

__global__ __launch_bounds__(kBlock) void attn4_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ qkv, const float* __restrict__ out,
                                                           const float* __restrict__ lse, float* __restrict__ dqkv, int L, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) float4 sm[];   // Q[L], K[L], V[L], dO[L], then lse[L], D[L] as floats
  float4* Qs = sm;
  float4* Ks = sm + L;
  float4* Vs = sm + 2 * L;
  float4* Gs = sm + 3 * L;
  float* Ls = reinterpret_cast<float*>(sm + 4 * L);
  float* Ds = Ls + L;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int inner = heads * 4, ld = 3 * inner;
  const float* base = qkv + (int64_t)b * L * ld + h * 4;
  for (int j = threadIdx.x; j < L; j += kBlock) {
    Qs[j] = *reinterpret_cast<const float4*>(base + (int64_t)j * ld);
    Ks[j] = *reinterpret_cast<const float4*>(base + (int64_t)j * ld + inner);
    Vs[j] = *reinterpret_cast<const float4*>(base + (int64_t)j * ld + 2 * inner);
    const float4 g = *reinterpret_cast<const float4*>(dout + ((int64_t)b * L + j) * inner + h * 4);
    const float4 o = *reinterpret_cast<const float4*>(out + ((int64_t)b * L + j) * inner + h * 4);
    Gs[j] = g;
    Ds[j] = dot4(g, o);
    Ls[j] = lse[((int64_t)b * heads + h) * L + j];
  }
  __syncthreads();
  const int r = blockIdx.y * 64 + (threadIdx.x >> 2), slice = threadIdx.x & 3;
  const bool live = r < L;
  const int rc = live ? r : L - 1;
  float* dst = dqkv + ((int64_t)b * L + rc) * ld + h * 4;
  {  // row r as a QUERY: dQ_r = scale * sum_j p_rj (dO_r . v_j - D_r) k_j
    const float4 q = Qs[rc], g = Gs[rc];
    const float lr = Ls[rc], dr = Ds[rc];
    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = slice; j < L; j += 4) {
      const float p = __expf(scale * dot4(q, Ks[j]) - lr);
      axpy4(dq, p * (dot4(g, Vs[j]) - dr), Ks[j]);
    }
    dq.x = quad_add(dq.x); dq.y = quad_add(dq.y); dq.z = quad_add(dq.z); dq.w = quad_add(dq.w);
    if (live && slice == 0) *reinterpret_cast<float4*>(dst) = make_float4(dq.x * scale, dq.y * scale, dq.z * scale, dq.w * scale);
  }
  {  // row r as a KEY: dV_r = sum_i p_ir dO_i ; dK_r = scale * sum_i p_ir (dO_i . v_r - D_i) q_i
    const float4 k = Ks[rc], v = Vs[rc];
    float4 dk = make_float4(0.f, 0.f, 0.f, 0.f), dv = dk;
    for (int i = slice; i < L; i += 4) {
      const float p = __expf(scale * dot4(Qs[i], k) - Ls[i]);
      axpy4(dv, p, Gs[i]);
      axpy4(dk, p * (dot4(Gs[i], v) - Ds[i]), Qs[i]);
    }
    dk.x = quad_add(dk.x); dk.y = quad_add(dk.y); dk.z = quad_add(dk.z); dk.w = quad_add(dk.w);
    dv.x = quad_add(dv.x); dv.y = quad_add(dv.y); dv.z = quad_add(dv.z); dv.w = quad_add(dv.w);
    if (live && slice == 0) {
      *reinterpret_cast<float4*>(dst + inner) = make_float4(dk.x * scale, dk.y * scale, dk.z * scale, dk.w * scale);
      *reinterpret_cast<float4*>(dst + 2 * inner) = dv;
    }
  }
}

int check(const char* who, int64_t B, int64_t L, int64_t heads) {
  ADNM_REQUIRE(B > 0 && L > 0 && heads > 0 && B * heads <= (1ll << 30), "%s: bad shape B=%lld L=%lld heads=%lld", who, (long long)B, (long long)L,
               (long long)heads);
  ADNM_REQUIRE(L <= 2048, "%s: L=%lld > 2048 tokens does not fit the workgroup's LDS", who, (long long)L);
  return ADNM_OK;
}
}  // namespace

extern "C" int adnm_attn4_fwd(const float* qkv, float* out, float* lse, int64_t B, int64_t L, int64_t heads, float scale, adnm_stream_t stream) {
  ADNM_REQUIRE(qkv && out && lse, "attn4_fwd: null pointer");
  if (int rc = check("attn4_fwd", B, L, heads)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)L * 2 * sizeof(float4);
  ADNM_ALLOW_LDS(attn4_fwd_kernel, smem, "attn4_fwd");
  ADNM_PROF("attn4_fwd", st, 4.0 * B * L * heads * (16.0 + 1.0));
  attn4_fwd_kernel<<<dim3((unsigned)(B * heads), (unsigned)adnm_cdiv(L, 64)), kBlock, smem, st>>>(qkv, out, lse, (int)L, (int)heads, scale);
  ADNM_CHECK_LAUNCH("attn4_fwd");
  return ADNM_OK;
}

extern "C" int adnm_attn4_bwd(const float* dout, const float* qkv, const float* out, const float* lse, float* dqkv, int64_t B, int64_t L,
                              int64_t heads, float scale, adnm_stream_t stream) {
  ADNM_REQUIRE(dout && qkv && out && lse && dqkv, "attn4_bwd: null pointer");
  if (int rc = check("attn4_bwd", B, L, heads)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)L * (4 * sizeof(float4) + 2 * sizeof(float));
  ADNM_ALLOW_LDS(attn4_bwd_kernel, smem, "attn4_bwd");
  ADNM_PROF("attn4_bwd", st, 4.0 * B * L * heads * (12.0 * 2 + 4 * 2 + 1));
  attn4_bwd_kernel<<<dim3((unsigned)(B * heads), (unsigned)adnm_cdiv(L, 64)), kBlock, smem, st>>>(dout, qkv, out, lse, dqkv, (int)L, (int)heads, scale);
  ADNM_CHECK_LAUNCH("attn4_bwd");
  return ADNM_OK;
}
