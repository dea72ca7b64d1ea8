// Fused step glue over FLAT fp32 buffers: global gradient norm, clip_grad_norm_ scaling
// (train.py:140) and the AdamW update (train_untils.py:35-42: betas (0.9,0.999), eps 1e-9, decoupled weight decay)
// in three launches instead of torch's ~100 multi-tensor launches over 669 tensors.  HBM-bound: 7 floats of
// traffic per parameter (read p,g,m,v; write p,m,v) + one extra read of g for the norm.
// The step counter and the bias corrections live on the device (state[0..3]) so the sequence is graph-replayable.
#include "adnm_common.h"

namespace {
constexpr int kBlock = 256;
constexpr int kNormBlocks = 1024;

__global__ __launch_bounds__(kBlock) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float sm[kBlock / 64];
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float t = g[(n4 << 2) + threadIdx.x];
    acc = fmaf(t, t, acc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// state: [0] step, [1] sum of squares of the gradient, [2] 1 - beta1^step, [3] sqrt(1 - beta2^step)
__global__ void optim_tick_kernel(float* __restrict__ state, float beta1, float beta2) {
  const float step = state[0] + 1.0f;
  state[0] = step;
  state[2] = 1.0f - powf(beta1, step);
  state[3] = sqrtf(1.0f - powf(beta2, step));
}

// one element of torch.optim.AdamW's update, written so that every kernel variant rounds alike (no contraction left to the context)
__device__ __forceinline__ void adamw_elem(float& P, float g, float& M, float& V, float coef, float decay, float step_size, float beta1, float beta2,
                                           float bc2s, float eps) {
#pragma clang fp contract(off)
  const float gk = g * coef;
  P = P * decay;                                      // p.mul_(1 - lr*wd)
  M = M + (1.0f - beta1) * (gk - M);                  // exp_avg.lerp_(g, 1-beta1)
  V = V * beta2 + ((1.0f - beta2) * gk) * gk;         // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
  const float denom = sqrtf(V) / bc2s + eps;
  P = P - step_size * (M / denom);                    // p.addcdiv_(exp_avg, denom, value=-lr/bc1)
}

// NT: streaming (non-temporal) accesses for g / m / v — read and written exactly once per step — and U float4 per lane and stream in
// flight; p keeps normal accesses (the next forward reads it)
// SH16: also write the bf16 SHADOW of the updated parameters (element i of sh16 = bf16(p[i])): the narrow copy the weight-streaming GEMMs
// read in the bf16 configuration (adnm_skgemm b_dtype ADNM_B_BF16) — half the bytes of the pass that re-reads 72 M parameters twice a step,
// for 2 more bytes per parameter here (the values are in registers anyway)
template <bool NT, int U, bool SH16>
__global__ __launch_bounds__(kBlock) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, const float* __restrict__ state, float lr,
                                                       float beta1, float beta2, float eps, float wd, float max_norm, uint16_t* __restrict__ sh16,
                                                       const float* __restrict__ hyper) {
  if (hyper) lr = hyper[0], max_norm = hyper[1];   // device-resident learning rate / clip threshold: a captured launch follows the host's schedule
  float coef = 1.0f;
  if (max_norm > 0.f) {  // torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (norm + 1e-6), max=1)
    coef = max_norm / (sqrtf(state[1]) + 1e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
  }
  const float bc1 = state[2], bc2s = state[3];
  const float decay = 1.0f - lr * wd, step_size = lr / bc1;
  const int64_t n4 = n >> 2;
  using f4 = __attribute__((ext_vector_type(4))) float;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n4; i0 += U * stride) {
    f4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {   // every load of the trip first
      const int64_t i = i0 + u * stride;
      if (i < n4) {
        pp[u] = reinterpret_cast<const f4*>(p)[i];
        if (NT) {
          gg[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(g) + i);
          mm[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(m) + i);
          vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(v) + i);
        } else {
          gg[u] = reinterpret_cast<const f4*>(g)[i], mm[u] = reinterpret_cast<const f4*>(m)[i], vv[u] = reinterpret_cast<const f4*>(v)[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= n4) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float P = pp[u][k], M = mm[u][k], V = vv[u][k];
        adamw_elem(P, gg[u][k], M, V, coef, decay, step_size, beta1, beta2, bc2s, eps);
        pp[u][k] = P, mm[u][k] = M, vv[u][k] = V;
      }
      reinterpret_cast<f4*>(p)[i] = pp[u];
      if (SH16) Io<uint16_t>::st4(sh16 + 4 * i, make_float4(pp[u][0], pp[u][1], pp[u][2], pp[u][3]));
      if (NT) {
        __builtin_nontemporal_store(mm[u], reinterpret_cast<f4*>(m) + i);
        __builtin_nontemporal_store(vv[u], reinterpret_cast<f4*>(v) + i);
      } else {
        reinterpret_cast<f4*>(m)[i] = mm[u], reinterpret_cast<f4*>(v)[i] = vv[u];
      }
    }
  }
}
// The fp8 configuration's optimiser pass (and, with UPD = false, the stand-alone shadow pass: no g / m / v, no update).  Besides the AdamW
// update it writes sh8[i] = e4m3(p[i] * scale_b(record of i's tensor)) — the scaled fp8 SHADOW the weight-streaming GEMMs read (adnm_skgemm
// b_dtype ADNM_B_FP8) — or, with SH = 1, the bf16 shadow; and while a tensor's record says so it collects max |p| into the record's amax_b,
// from which adnm_quant_update makes the next scale (delayed scaling: the shadow and the GEMMs always see the same scale, include/adnm_hip.h).
// Tensors are the segments of the flat buffer: seg_end[s] = end of segment s in float4 units (tensors are 16-byte aligned in the flat
// layout), seg_rec[s] = row of its record in the table, or < 0 (not a GEMM weight: no scale, the shadow bytes are never read).
// A workgroup owns a CONTIGUOUS range, so a wave stays inside one tensor for many trips and commits one atomic max per tensor it crosses.
template <bool UPD, int SH>
__global__ __launch_bounds__(kBlock) void adamw_seg_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, int64_t n4, int64_t per_block, const float* __restrict__ state,
                                                           float lr, float beta1, float beta2, float eps, float wd, float max_norm,
                                                           void* __restrict__ shadow, const int* __restrict__ seg_end, const int* __restrict__ seg_rec,
                                                           int nseg, AdnmQuant* __restrict__ tab, int collect, const float* __restrict__ hyper) {
  if (UPD && hyper) lr = hyper[0], max_norm = hyper[1];
  float coef = 1.0f, bc1 = 1.f, bc2s = 1.f;
  if (UPD) {
    if (max_norm > 0.f) {
      coef = max_norm / (sqrtf(state[1]) + 1e-6f);
      coef = coef > 1.0f ? 1.0f : coef;
    }
    bc1 = state[2], bc2s = state[3];
  }
  const float decay = 1.0f - lr * wd, step_size = lr / bc1;
  using f4 = __attribute__((ext_vector_type(4))) float;
  const int64_t lo = (int64_t)blockIdx.x * per_block, hi = lo + per_block < n4 ? lo + per_block : n4;
  int seg = 0;
  {   // first segment whose end lies beyond this thread's first quad
    const int64_t i0 = lo + threadIdx.x;
    int a = 0, b = nseg - 1;
    while (a < b) {
      const int mid = (a + b) >> 1;
      if ((int64_t)seg_end[mid] > i0) b = mid; else a = mid + 1;
    }
    seg = a;
  }
  int rec = -2;        // the record this WAVE is collecting for (-2: none yet)
  float amax = 0.f, scale = 1.f;
  auto commit = [&]() {   // wave-uniform: every lane holds the same `rec`
    if (rec >= 0 && collect && tab[rec].record != 0.f) adnm_amax_commit(&tab[rec].amax_b, amax);   // (the record says when: calibration steps)
    amax = 0.f;
  };
  for (int64_t i = lo + threadIdx.x; i < lo + per_block; i += kBlock) {   // (uniform trip count: the wave votes below need every lane)
    const bool in = i < hi;
    while (in && seg + 1 < nseg && i >= (int64_t)seg_end[seg]) ++seg;
    const int r = in ? (SH == 2 ? seg_rec[seg] : -1) : -3;   // (the bf16 shadow has no records: the table may be NULL)
    f4 pp = {0.f, 0.f, 0.f, 0.f};
    if (in) {
      pp = reinterpret_cast<const f4*>(p)[i];
      if (UPD) {
        const f4 gg = __builtin_nontemporal_load(reinterpret_cast<const f4*>(g) + i);
        f4 mm = __builtin_nontemporal_load(reinterpret_cast<const f4*>(m) + i), vv = __builtin_nontemporal_load(reinterpret_cast<const f4*>(v) + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float P = pp[k], M = mm[k], V = vv[k];
          adamw_elem(P, gg[k], M, V, coef, decay, step_size, beta1, beta2, bc2s, eps);
          pp[k] = P, mm[k] = M, vv[k] = V;
        }
        reinterpret_cast<f4*>(p)[i] = pp;
        __builtin_nontemporal_store(mm, reinterpret_cast<f4*>(m) + i);
        __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(v) + i);
      }
    }
    // which record do this wave's lanes belong to?  Uniform (the common case: a tensor is thousands of quads): keep collecting in
    // registers; a change of tensor commits the old one's maximum with ONE atomic.  A trip that straddles tensors commits per lane.
    const int r0 = __builtin_amdgcn_readfirstlane(r);
    const bool uniform = __all(r == r0 || !in) && __builtin_amdgcn_readfirstlane(in ? 1 : 0);
    float s_here = 1.f;
    if (uniform) {
      if (r0 != rec) {
        commit();
        rec = r0;
        scale = rec >= 0 ? tab[rec].scale_b : 1.f;
      }
      s_here = scale;
      if (in) amax = adnm_amax4(amax, pp[0], pp[1], pp[2], pp[3]);
    } else {
      commit();
      rec = -2;
      if (in && r >= 0) {
        s_here = tab[r].scale_b;
        const float mx = adnm_amax4(0.f, pp[0], pp[1], pp[2], pp[3]);
        if (collect && mx > 0.f && tab[r].record != 0.f) atomicMax(reinterpret_cast<unsigned int*>(&tab[r].amax_b), __float_as_uint(mx));
      }
    }
    if (in && shadow) {
      if (SH == 1) {
        Io<uint16_t>::st4(reinterpret_cast<uint16_t*>(shadow) + 4 * i, make_float4(pp[0], pp[1], pp[2], pp[3]));
      } else {
        const float c0 = __builtin_amdgcn_fmed3f(pp[0] * s_here, 448.f, -448.f), c1 = __builtin_amdgcn_fmed3f(pp[1] * s_here, 448.f, -448.f);
        const float c2 = __builtin_amdgcn_fmed3f(pp[2] * s_here, 448.f, -448.f), c3 = __builtin_amdgcn_fmed3f(pp[3] * s_here, 448.f, -448.f);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(c0, c1, w, false), w = __builtin_amdgcn_cvt_pk_fp8_f32(c2, c3, w, true);
        reinterpret_cast<int*>(shadow)[i] = w;
      }
    }
  }
  commit();
}

// wire format of the gradient all-reduce (reduce_dtype = bf16): one streaming pass each way, 8 elements per lane
__global__ __launch_bounds__(kBlock) void cast_f32_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n, float scale) {
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kBlock) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(a.x * scale) | ((uint32_t)f32_to_bf16(a.y * scale) << 16);
    o.y = (uint32_t)f32_to_bf16(a.z * scale) | ((uint32_t)f32_to_bf16(a.w * scale) << 16);
    o.z = (uint32_t)f32_to_bf16(b.x * scale) | ((uint32_t)f32_to_bf16(b.y * scale) << 16);
    o.w = (uint32_t)f32_to_bf16(b.z * scale) | ((uint32_t)f32_to_bf16(b.w * scale) << 16);
    reinterpret_cast<uint4*>(dst)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(n8 << 3) + threadIdx.x] = f32_to_bf16(src[(n8 << 3) + threadIdx.x] * scale);
}
__global__ __launch_bounds__(kBlock) void cast_bf16_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, int64_t n, float scale) {
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kBlock) {
    const uint4 v = reinterpret_cast<const uint4*>(src)[i];
    reinterpret_cast<float4*>(dst)[2 * i] = make_float4(scale * __uint_as_float(v.x << 16), scale * __uint_as_float(v.x & 0xffff0000u),
                                                        scale * __uint_as_float(v.y << 16), scale * __uint_as_float(v.y & 0xffff0000u));
    reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4(scale * __uint_as_float(v.z << 16), scale * __uint_as_float(v.z & 0xffff0000u),
                                                            scale * __uint_as_float(v.w << 16), scale * __uint_as_float(v.w & 0xffff0000u));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(n8 << 3) + threadIdx.x] = scale * bf16_to_f32(src[(n8 << 3) + threadIdx.x]);
}
}  // namespace

static int cast_launch(const void* src, void* dst, int64_t n, float scale, adnm_stream_t stream, bool to_bf16) {
  ADNM_REQUIRE(src && dst && n > 0, "cast: null pointer or n <= 0");
  ADNM_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "cast: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = adnm_cdiv(adnm_cdiv(n, 8), kBlock);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  ADNM_PROF(to_bf16 ? "cast_f32_bf16" : "cast_bf16_f32", st, 6.0 * n);
  if (to_bf16) cast_f32_bf16_kernel<<<(unsigned)blocks, kBlock, 0, st>>>((const float*)src, (uint16_t*)dst, n, scale);
  else cast_bf16_f32_kernel<<<(unsigned)blocks, kBlock, 0, st>>>((const uint16_t*)src, (float*)dst, n, scale);
  ADNM_CHECK_LAUNCH("cast");
  return ADNM_OK;
}
extern "C" int adnm_cast_f32_bf16(const void* src, void* dst, int64_t n, float scale, adnm_stream_t stream) {
  return cast_launch(src, dst, n, scale, stream, true);
}
extern "C" int adnm_cast_bf16_f32(const void* src, void* dst, int64_t n, float scale, adnm_stream_t stream) {
  return cast_launch(src, dst, n, scale, stream, false);
}

extern "C" int64_t adnm_adamw_ws_bytes(void) { return kNormBlocks * (int64_t)sizeof(float); }

namespace {
int seg_launch(bool upd, float* p, const float* g, float* m, float* v, int64_t n, float* state, float lr, float beta1, float beta2, float eps,
               float wd, float max_norm, void* shadow, int shadow_dtype, const int* seg_end, const int* seg_rec, int64_t nseg, float* tab,
               int collect, const float* hyper, hipStream_t st) {
  ADNM_REQUIRE(seg_end && seg_rec && nseg >= 1 && nseg < (1 << 24) && (tab || shadow_dtype != ADNM_B_FP8),
               "adamw / shadow pass: the fp8 shadow needs the segment tables and the record table");
  ADNM_REQUIRE(n / 4 < (1ll << 31), "adamw / shadow pass: more than 2^31 quads");
  const int64_t n4 = n / 4;
  int64_t blocks = adnm_cdiv(n4, (int64_t)kBlock * 8);   // >= 8 trips per thread, <= 2048 workgroups
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const int64_t per_block = adnm_cdiv(adnm_cdiv(n4, blocks), kBlock) * kBlock;
  AdnmQuant* t = reinterpret_cast<AdnmQuant*>(tab);
#define SEG(UPDV, SHV) adamw_seg_kernel<UPDV, SHV><<<(unsigned)blocks, kBlock, 0, st>>>(p, g, m, v, n4, per_block, state, lr, beta1, beta2, eps, wd, \
                                                                                       max_norm, shadow, seg_end, seg_rec, (int)nseg, t, collect, hyper)
  if (upd) {
    if (shadow_dtype == ADNM_B_BF16) SEG(true, 1); else SEG(true, 2);
  } else {
    if (shadow_dtype == ADNM_B_BF16) SEG(false, 1); else SEG(false, 2);
  }
#undef SEG
  return ADNM_OK;
}
}  // namespace

extern "C" int adnm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float* state, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float max_norm, void* ws, int64_t ws_bytes, void* shadow, int shadow_dtype,
                               const int* seg_end, const int* seg_rec, int64_t nseg, float* wtab, const float* hyper, adnm_stream_t stream) {
  ADNM_REQUIRE(p && g && m && v && state, "adamw_step: null pointer");
  ADNM_REQUIRE(n > 0 && n % 4 == 0, "adamw_step: n=%lld must be a positive multiple of 4 (pad the flat buffers)", (long long)n);
  ADNM_REQUIRE(!shadow || shadow_dtype == ADNM_B_BF16 || shadow_dtype == ADNM_B_FP8, "adamw_step: the shadow is bf16 (1) or scaled e4m3 (2)");
  if (!ws || ws_bytes < adnm_adamw_ws_bytes()) {
    adnm_set_error("adamw_step: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_adamw_ws_bytes());
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  { ADNM_PROF("grad_sumsq", st, 4.0 * n); sumsq_partial_kernel<<<kNormBlocks, kBlock, 0, st>>>(g, n, part); }
  adnm_launch_fold("grad_sumsq_fold", part, kNormBlocks, 1, {state + 1, 1}, {nullptr, 0}, {nullptr, 0}, {nullptr, 0}, st);
  optim_tick_kernel<<<1, 1, 0, st>>>(state, beta1, beta2);
  int64_t blocks = adnm_cdiv(n / 4, kBlock);
  if (blocks > 4096) blocks = 4096;
  {   // measured in one session (two runs each): plain accesses 0.446 ms, non-temporal g / m / v 0.414, + two float4 per lane 0.400 / 0.47 without
    ADNM_PROF("adamw_update", st, 4.0 * n * 7 + (shadow ? (shadow_dtype == ADNM_B_BF16 ? 2.0 : 1.0) * n : 0.0));
    // with the segment tables at hand the contiguous-range kernel takes the bf16 shadow too (measured in one trace: 0.36 ms against 0.41 ms
    // for the interleaved kernel with the extra 2-byte stream)
    if (shadow && (shadow_dtype == ADNM_B_FP8 || (seg_end && seg_rec && nseg >= 1))) {
      if (int rc = seg_launch(true, p, g, m, v, n, state, lr, beta1, beta2, eps, weight_decay, max_norm, shadow, shadow_dtype, seg_end, seg_rec, nseg, wtab, 1, hyper, st)) return rc;
    } else if (shadow) {
      adamw_kernel<true, 1, true><<<(unsigned)blocks, kBlock, 0, st>>>(p, g, m, v, n, state, lr, beta1, beta2, eps, weight_decay, max_norm, (uint16_t*)shadow, hyper);
    } else {
      adamw_kernel<true, 1, false><<<(unsigned)blocks, kBlock, 0, st>>>(p, g, m, v, n, state, lr, beta1, beta2, eps, weight_decay, max_norm, nullptr, hyper);
    }
  }
  ADNM_CHECK_LAUNCH("adamw_step");
  return ADNM_OK;
}

// The shadow alone, from the parameters as they are (after the trainer has moved them into the flat buffer, after load_state_dict):
// shadow = NULL with collect != 0 only collects max |p| per record (the first calibration).
extern "C" int adnm_shadow_refresh(const float* p, int64_t n, void* shadow, int shadow_dtype, const int* seg_end, const int* seg_rec, int64_t nseg,
                                   float* wtab, int collect, adnm_stream_t stream) {
  ADNM_REQUIRE(p && n > 0 && n % 4 == 0, "shadow_refresh: bad arguments");
  ADNM_REQUIRE(shadow_dtype == ADNM_B_BF16 || shadow_dtype == ADNM_B_FP8, "shadow_refresh: the shadow is bf16 (1) or scaled e4m3 (2)");
  hipStream_t st = (hipStream_t)stream;
  ADNM_PROF("shadow_refresh", st, 4.0 * n + (shadow ? (shadow_dtype == ADNM_B_BF16 ? 2.0 : 1.0) * n : 0.0));
  if (shadow_dtype == ADNM_B_BF16) {
    ADNM_REQUIRE(shadow, "shadow_refresh: no destination");
    return adnm_cast_f32_bf16(p, shadow, n, 1.0f, stream);
  }
  if (int rc = seg_launch(false, const_cast<float*>(p), nullptr, nullptr, nullptr, n, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, shadow, shadow_dtype, seg_end,
                          seg_rec, nseg, wtab, collect, nullptr, st))
    return rc;
  ADNM_CHECK_LAUNCH("shadow_refresh");
  return ADNM_OK;
}
