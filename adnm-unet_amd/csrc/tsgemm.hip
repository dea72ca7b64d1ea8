// K6 — the 1x1 / Linear projections of the full-resolution stages as "tall-skinny" fp32 GEMMs on the matrix cores.
//   Reference call sites: Mamba2.in_proj / out_proj (ADNssd.py:309,461), FeedForward.project_in / project_out
//   (model_untils.py:193,196), Mlp.fc1/fc2 (:64,67), decoder6_s (ADNMUNet.py:634), OutProj's 1x1 (model_untils.py:831).
// Shapes: M = B*H*W tokens is huge (65 536 at config 2), K and N are tiny (32..256).  Library GEMMs pick tiles for
// square problems (measured here: 93 us for 65536x32 @ 32x208, 196 us for the 208x32 weight gradient with a 65 536-long
// reduction under hipBLASLt); the work is 63 MB of mandatory traffic = 12 us at 5 TB/s and 0.9 GFLOP.
//
// Design: v_mfma_f32_16x16x4_f32 (exact fp32, bit-for-bit an fmaf chain — the parity tolerance is untouched).
//   * nt kernel (forward and input gradient):  Y[M,N] = X[M,K] . Wp[N,K]^T (+bias).  The whole (strided) weight is
//     staged ONCE per workgroup into LDS in B-fragment order ([col block][k step][lane]) so every B fragment is one
//     conflict-free ds_read_b32; a wave then streams 16-row blocks of X: K/16 float4 loads per lane (the k index is
//     permuted identically for A and B, which a dot product does not care about), NB*K/4 MFMAs, NB*4 accumulators.
//     The input gradient dX = dY . W is the same kernel with the weight read through swapped strides.
//   * tn kernel (weight gradient):  dW[N,K] = sum_m dY[m,N]^T X[m,K]: per-wave accumulators for the whole N x K result
//     (<= 32 blocks of 16x16), a 64-B segment of every dY / X row per MFMA, waves folded through LDS, one fp32
//     partial per workgroup, deterministic fold (no atomics).
#include "adnm_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kBlock = 512;  // 8 waves share one LDS copy of the weight (nt) / fold through one LDS buffer (tn)
constexpr int kWaves = 8;

// ---------------------------------------------------------------------------------------------- nt: Y = X . Wp^T
// Wp[n][k] is read at w[n * ws_n + k * ws_k]  (forward: ws_n=K, ws_k=1; input gradient: ws_n=1, ws_k=K_of_weight)
// PREC = ADNM_MFMA_*; A_BF8: in the fp8 mode the rows of X are a gradient (e5m2).
template <int PREC>
struct WImg {   // bytes of one (column block, reduction group, lane) entry of the LDS weight image, reduction steps per entry
  static constexpr int kEntry = PREC == ADNM_MFMA_F32 ? 4 : (PREC == ADNM_MFMA_BF16 ? 16 : 8);
  static constexpr int kSteps = PREC == ADNM_MFMA_F32 ? 1 : 8;   // per lane
};
// AT / CT: storage type of the token rows read / written (float, or uint16_t = bf16: the wide intermediates of the full-resolution level
// are kept in bf16 when the matrix-core precision is bf16 anyway — half the bytes of the kernels that are bound by exactly those bytes).
template <int KQ, int NB, int PREC, bool A_BF8, typename AT, typename CT>
__global__ __launch_bounds__(kBlock) void tsgemm_nt_kernel(const AT* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                           int64_t ws_n, int64_t ws_k, const float* __restrict__ bias,
                                                           CT* __restrict__ y, int64_t ldy, int64_t M, int N, int K, AdnmQuant* q) {
  // fp32: [NB][K/4][64] floats, one per (column block, k step, lane).  bf16 / fp8: [NB][KP][64] entries of 8 bf16 / 8 fp8 = the lane's
  // eight reduction steps of ONE v_mfma_f32_16x16x32 (KP = ceil(KQ / 2) groups of 32 steps; an odd last half is zeros)
  extern __shared__ __attribute__((aligned(16))) float wl[];
  uint16_t* const wh = reinterpret_cast<uint16_t*>(wl);
  uint8_t* const wb = reinterpret_cast<uint8_t*>(wl);
  constexpr int KP = (KQ + 1) / 2;
  const int ksteps = K >> 2;
  constexpr int kTStride = 68;   // floats per row of a wave's 16 x 64 output staging tile (64 + pad: conflict-free float4 writes)
  const int wbytes = PREC == ADNM_MFMA_F32 ? NB * (K >> 2) * 64 * 4 : NB * KP * 64 * WImg<PREC>::kEntry;
  float* const tbuf = wl + ((wbytes + 15) / 16) * 4;   // behind the weight image
  float q_sa = 1.f, q_sb = 1.f, amax_a = 0.f, amax_b = 0.f;
  bool rec = false;
  if (q) {
    if (PREC == ADNM_MFMA_FP8) q_sa = q->scale_a, q_sb = q->scale_b;
    rec = q->record != 0.f;
  }
  // stage the weight in fragment order: entry (cb, s, lane=(j,kk)) = Wp[cb*16+j][16*(s/4) + 4*kk + (s%4)].
  // Global reads run along the weight's contiguous axis (k for the forward layout, n for the transposed one).
  {
    const int Kp = PREC == ADNM_MFMA_F32 ? K : KP * 32;   // the narrow images cover whole 32-step groups: steps >= K are zeros
    const int total = NB * 16 * Kp;
    constexpr int kU = 8;   // loads issued back to back before the first LDS write: one HBM round trip per 4096 elements
    for (int base = 0; base < total; base += kBlock * kU) {
      float v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * kBlock + threadIdx.x;
        int n, k;
        if (ws_k == 1) { n = idx / Kp; k = idx - n * Kp; }
        else { k = idx / (NB * 16); n = idx - k * (NB * 16); }
        v[u] = (idx < total && n < N && k < K) ? w[(int64_t)n * ws_n + (int64_t)k * ws_k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int idx = base + u * kBlock + threadIdx.x;
        if (idx >= total) continue;
        int n, k;
        if (ws_k == 1) { n = idx / Kp; k = idx - n * Kp; }
        else { k = idx / (NB * 16); n = idx - k * (NB * 16); }
        const int cb = n >> 4, j = n & 15, qq = k >> 4, kk = (k >> 2) & 3, e = k & 3;
        amax_b = fmaxf(amax_b, fabsf(v[u]));
        if (PREC == ADNM_MFMA_BF16) wh[((cb * KP + (qq >> 1)) * 64 + kk * 16 + j) * 8 + (qq & 1) * 4 + e] = f32_to_bf16(v[u]);
        else if (PREC == ADNM_MFMA_FP8)
          wb[((cb * KP + (qq >> 1)) * 64 + kk * 16 + j) * 8 + (qq & 1) * 4 + e] =
              (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[u] * q_sb, 448.f, -448.f), 0.f, 0, false) & 0xff);
        else wl[(cb * ksteps + (qq * 4 + e)) * 64 + kk * 16 + j] = v[u];
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t nrb = (M + 15) >> 4;
  const int64_t stride = (int64_t)gridDim.x * kWaves;
  int64_t rb = (int64_t)blockIdx.x * kWaves + wave;
  // the first row block's loads are in flight while the weight is being staged; inside the loop the NEXT block's
  // loads are issued before the current block's MFMAs, so a wave never waits for HBM with an idle matrix pipe
  float4 xa[KQ], xn[KQ];
  auto fetch = [&](int64_t b, float4 (&dst)[KQ]) {
    const int64_t row = b * 16 + i;
    const bool rv = b < nrb && row < M;
#pragma unroll
    for (int qq = 0; qq < KQ; ++qq) dst[qq] = rv ? Io<AT>::ld4(x + row * ldx + 16 * qq + 4 * kk) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  const bool vec_ok = (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & (4 * sizeof(CT) - 1)) == 0 && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
  constexpr bool kPrefetch = KQ < 16 && NB < 16;   // the widest variants have no registers to spare for a second row block
  const float inv = 1.0f / (q_sa * q_sb);
  fetch(rb, xa);
  __syncthreads();
  for (; rb < nrb; rb += stride) {
    if (kPrefetch) fetch(rb + stride, xn);
    f32x4 acc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rec) {
#pragma unroll
      for (int qq = 0; qq < KQ; ++qq) amax_a = adnm_amax4(amax_a, xa[qq].x, xa[qq].y, xa[qq].z, xa[qq].w);
    }
    if constexpr (PREC != ADNM_MFMA_F32) {
#pragma unroll
      for (int qp = 0; qp < KP; ++qp) {
        const float lo[4] = {xa[2 * qp].x, xa[2 * qp].y, xa[2 * qp].z, xa[2 * qp].w};
        float hi[4] = {0.f, 0.f, 0.f, 0.f};
        if (2 * qp + 1 < KQ) hi[0] = xa[(2 * qp + 1) % KQ].x, hi[1] = xa[(2 * qp + 1) % KQ].y, hi[2] = xa[(2 * qp + 1) % KQ].z, hi[3] = xa[(2 * qp + 1) % KQ].w;
        const AdnmFrag<PREC> xf = adnm_make_frag<PREC, A_BF8>(lo, hi, q_sa);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
          AdnmFrag<PREC> wf;
          if constexpr (PREC == ADNM_MFMA_BF16) wf.v = __builtin_bit_cast(adnm_bf16x8, *reinterpret_cast<const uint4*>(wh + ((cb * KP + qp) * 64 + lane) * 8));
          else wf.v = *reinterpret_cast<const long*>(wb + ((cb * KP + qp) * 64 + lane) * 8);
          // operands swapped (W fragment first): the accumulator tile is Y^T, a lane ends up with FOUR CONSECUTIVE output columns of one row
          acc[cb] = adnm_mma<PREC, false, A_BF8>(wf, xf, acc[cb]);
        }
      }
      if (PREC == ADNM_MFMA_FP8) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[cb] = acc[cb] * inv;
      }
    } else {
#pragma unroll
      for (int qq = 0; qq < KQ; ++qq) {
        const float ae[4] = {xa[qq].x, xa[qq].y, xa[qq].z, xa[qq].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int s = qq * 4 + e;
#pragma unroll
          // operands swapped (W fragment as A, X fragment as B): the accumulator tile is Y^T, i.e. a lane ends up with FOUR
          // CONSECUTIVE output columns of one row -> float4 stores (8 per row block instead of 32 scalar ones)
          for (int cb = 0; cb < NB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(cb * ksteps + s) * 64 + lane], ae[e], acc[cb], 0, 0, 0);
        }
      }
    }
    // C^T layout: column (lane & 15) -> row of Y, row (lane >> 4) * 4 + reg -> column of Y: a lane holds four consecutive output columns
    // of row i per column block, so a direct store instruction writes 16 rows x 64 B — half-line pieces, which is what bounds the
    // output-heavy shapes (32 -> 128 / 208 features).  With 16-byte-aligned rows the tile goes through a wave-private LDS buffer instead,
    // four column blocks (64 columns) at a time, and is stored as 4 rows x 256 contiguous bytes per instruction (+ bias on the way out).
    if (vec_ok && (N & 3) == 0) {
      float* const tb = tbuf + wave * (16 * kTStride);
#pragma unroll
      for (int g0 = 0; g0 < NB; g0 += 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (g0 + c < NB) *reinterpret_cast<f32x4*>(tb + i * kTStride + c * 16 + kk * 4) = acc[g0 + c];
        __builtin_amdgcn_wave_barrier();
        const int col = g0 * 16 + (lane & 15) * 4;
        if (col < N) {
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias) bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int r = (lane >> 4) + 4 * t;
            const int64_t orow = rb * 16 + r;
            const float4 v = *reinterpret_cast<const float4*>(tb + r * kTStride + (lane & 15) * 4);
            if (orow < M) Io<CT>::st4(y + orow * ldy + col, make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w));
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      const int64_t orow = rb * 16 + i;
      if (orow < M) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
          const int n = cb * 16 + kk * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < N) Io<CT>::st(y + orow * ldy + n + r, acc[cb][r] + (bias ? bias[n + r] : 0.f));
        }
      }
    }
    if (kPrefetch) {
#pragma unroll
      for (int qq = 0; qq < KQ; ++qq) xa[qq] = xn[qq];
    } else {
      fetch(rb + stride, xa);
    }
  }
  if (rec) {   // amax of the rows this workgroup streamed (8 waves -> one atomic through LDS) and, from workgroup 0, of the weight
    __syncthreads();
    float* const red = wl;   // the weight image is idle by now
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax_a = fmaxf(amax_a, __shfl_xor(amax_a, o, 64)), amax_b = fmaxf(amax_b, __shfl_xor(amax_b, o, 64));
    if (lane == 0) red[wave] = amax_a, red[kWaves + wave] = amax_b;
    __syncthreads();
    if (threadIdx.x == 0) {
      float ma = 0.f, mb = 0.f;
#pragma unroll
      for (int k = 0; k < kWaves; ++k) ma = fmaxf(ma, red[k]), mb = fmaxf(mb, red[kWaves + k]);
      if (ma > 0.f) atomicMax(reinterpret_cast<unsigned int*>(&q->amax_a), __float_as_uint(ma));
      if (mb > 0.f && blockIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(&q->amax_b), __float_as_uint(mb));
    }
  }
}

// ---------------------------------------------------------------------------------------------- tn: dW = dY^T . X
// part[blockIdx.x][n][k]; A = dY^T (16 n x 4 m), B = X (4 m x 16 k)
// (bx, by, nbx): this workgroup's row-band index, its N-tile and the number of row bands of ITS problem — the grid of the single-problem
// launch, a block range of the grouped launch (tsgemm_tn_multi_kernel)
template <int NBN, int NBK, typename YT, typename XT>
__device__ __forceinline__ void tsgemm_tn_body(const YT* __restrict__ dy, int64_t lddy, const XT* __restrict__ x, int64_t ldx, float* __restrict__ part,
                                               float* __restrict__ bpart, int64_t M, int N, int K, const int bx, const int by, const int nbx,
                                               float* __restrict__ sm) {   // sm: 4 x (NBN*NBK*256 + NBN*16) floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int nb0 = by * NBN * 16;  // first output row (n) of this block's tile
  f32x4 acc[NBN][NBK];
  float bsum[NBN];
#pragma unroll
  for (int a = 0; a < NBN; ++a) {
    bsum[a] = 0.f;
#pragma unroll
    for (int b = 0; b < NBK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t nmb = (M + 3) >> 2;
  const int64_t stride = (int64_t)nbx * kWaves;
  // D rounds of operand loads in flight per wave: one round is (NBN + NBK) x 256 B — with one round ahead a CU had ~30 KB outstanding,
  // i.e. the kernel ran at latency x bytes-in-flight (2.3 TB/s at the 65 536-token level) instead of at the HBM rate
  constexpr int D = NBN + NBK <= 6 ? 4 : (NBN + NBK <= 10 ? 3 : (NBN * NBK <= 16 ? 2 : 1));   // (the 26-block tile has no registers to spare)
  float ra[D + 1][NBN], rb[D + 1][NBK];
  // UNCONDITIONAL loads from clamped (valid) addresses: a predicated load becomes a branch around it, and across branches the compiler
  // gives up counting outstanding loads — every round then waited for ALL loads in flight (s_waitcnt vmcnt(0)) and the ring was worth
  // nothing.  A row past M is clamped to the last row and its dY values are zeroed when the round is CONSUMED (0 x finite = 0; the bias
  // sum takes the zeroed value too); a column past N / K is clamped to the last column and lands in accumulator rows / columns that are
  // never stored.  Addresses are uniform base + 32-bit lane offset (host-checked: M * ld < 2^31): one register per load in flight.
  const uint32_t cn = (uint32_t)(nb0 + i), ck = (uint32_t)i, nmax = (uint32_t)(N - 1), kmax = (uint32_t)(K - 1);
  const uint32_t lddy32 = (uint32_t)lddy, ldx32 = (uint32_t)ldx, mlast = (uint32_t)(M - 1);
  auto fetch = [&](int64_t mb_, float (&a_)[NBN], float (&b_)[NBK]) {
    const int64_t m64 = mb_ * 4 + kk;
    const uint32_t m = m64 < M ? (uint32_t)m64 : mlast;
    const uint32_t oa = m * lddy32, ob = m * ldx32;
#pragma unroll
    for (int a = 0; a < NBN; ++a) a_[a] = Io<YT>::ld(dy + (oa + min(cn + 16u * a, nmax)));
#pragma unroll
    for (int b = 0; b < NBK; ++b) b_[b] = Io<XT>::ld(x + (ob + min(ck + 16u * b, kmax)));
  };
  int64_t mb = (int64_t)bx * kWaves + wave;
#pragma unroll
  for (int s = 0; s < D; ++s) fetch(mb + s * stride, ra[s], rb[s]);
  for (; mb < nmb; mb += (D + 1) * stride) {
#pragma unroll
    for (int s = 0; s <= D; ++s) {   // slot s holds round mb + s * stride
      fetch(mb + (s + D) * stride, ra[(s + D) % (D + 1)], rb[(s + D) % (D + 1)]);
      const bool mv = (mb + s * stride) * 4 + kk < M;
      float av[NBN];
#pragma unroll
      for (int a = 0; a < NBN; ++a) av[a] = mv ? ra[s][a] : 0.f;
#pragma unroll
      for (int a = 0; a < NBN; ++a) bsum[a] += av[a];
#pragma unroll
      for (int a = 0; a < NBN; ++a)
#pragma unroll
        for (int b = 0; b < NBK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], rb[s][b], acc[a][b], 0, 0, 0);
    }
  }
  // bias gradient: lanes with the same i (n) but different kk hold different m's
#pragma unroll
  for (int a = 0; a < NBN; ++a) bsum[a] += __shfl_xor(bsum[a], 16, 64), bsum[a] += __shfl_xor(bsum[a], 32, 64);
  // fold the 8 waves as a tree through 4 LDS buffers (3 rounds; upper half writes, lower half adds) — deterministic order
  constexpr int kAcc = NBN * NBK * 256, kBuf = kAcc + NBN * 16;
#pragma unroll
  for (int half = kWaves / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      float* buf = sm + (wave - half) * kBuf;
#pragma unroll
      for (int a = 0; a < NBN; ++a)
#pragma unroll
        for (int b = 0; b < NBK; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) buf[((a * NBK + b) * 4 + r) * 64 + lane] = acc[a][b][r];
      if (kk == 0)
#pragma unroll
        for (int a = 0; a < NBN; ++a) buf[kAcc + a * 16 + i] = bsum[a];
    }
    __syncthreads();
    if (wave < half) {
      const float* buf = sm + wave * kBuf;
#pragma unroll
      for (int a = 0; a < NBN; ++a) {
#pragma unroll
        for (int b = 0; b < NBK; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][b][r] += buf[((a * NBK + b) * 4 + r) * 64 + lane];
        if (kk == 0) bsum[a] += buf[kAcc + a * 16 + i];
      }
    }
    __syncthreads();
  }
  if (wave == 0) {
    const int64_t rowlen = (int64_t)N * K + (bpart ? N : 0);   // one partial row = [dW tile rows | bias sums]: ONE fold for both
    float* dst = part + (int64_t)bx * rowlen;
#pragma unroll
    for (int a = 0; a < NBN; ++a) {
#pragma unroll
      for (int b = 0; b < NBK; ++b) {
        const int k = b * 16 + i;  // C layout: col = lane&15 -> k, row = kk*4+r -> n
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nb0 + a * 16 + kk * 4 + r;
          if (n < N && k < K) dst[(int64_t)n * K + k] = acc[a][b][r];
        }
      }
      if (bpart && kk == 0 && nb0 + a * 16 + i < N) dst[(int64_t)N * K + nb0 + a * 16 + i] = bsum[a];
    }
  }
}

template <int NBN, int NBK, typename YT, typename XT>
__global__ __launch_bounds__(kBlock) void tsgemm_tn_kernel(const YT* __restrict__ dy, int64_t lddy, const XT* __restrict__ x, int64_t ldx,
                                                           float* __restrict__ part, float* __restrict__ bpart, int64_t M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  tsgemm_tn_body<NBN, NBK, YT, XT>(dy, lddy, x, ldx, part, bpart, M, N, K, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, sm);
}

// The weight-gradient GEMMs of a backward stage in ONE launch.  Each of them is a leaf (nothing reads dW before the optimiser) and, at
// the 65 536-token level, a SHORT kernel: 25 - 60 MB streamed in ~8 us between a launch ramp and a tail (LDS tree over the 8 waves, one
// wave's stores) of comparable length — 20 launches per step at config 2 ran at 2.3 TB/s.  Queued (core.hip: adnm_leafq_*) and launched
// together, one problem's tail overlaps the next one's stream.  fp32 storage only; the descriptor table lives in the kernarg segment.
constexpr int kMaxTn = 24;
struct TnArgs {
  const float* dy;
  const float* x;
  float* part;
  float* bpart;
  int64_t lddy, ldx, M;
  int N, K, a, b, nblk, ntiles;
};
struct MultiTn {
  int count;
  int blk_end[kMaxTn];
  TnArgs p[kMaxTn];
};
__global__ __launch_bounds__(kBlock) void tsgemm_tn_multi_kernel(MultiTn m_by_value) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  (void)m_by_value;
  const __attribute__((address_space(4))) MultiTn& m = *(const __attribute__((address_space(4))) MultiTn*)__builtin_amdgcn_kernarg_segment_ptr();
  int k = 0;
  while (k + 1 < m.count && (int)blockIdx.x >= m.blk_end[k]) ++k;
  const int local = (int)blockIdx.x - (k ? m.blk_end[k - 1] : 0);
  const float* dy = m.p[k].dy;
  const float* x = m.p[k].x;
  float* part = m.p[k].part;
  float* bpart = m.p[k].bpart;
  const int64_t lddy = m.p[k].lddy, ldx = m.p[k].ldx, M = m.p[k].M;
  const int N = m.p[k].N, K = m.p[k].K, nblk = m.p[k].nblk;
  const int bx = local % nblk, by = local / nblk;
  const int variant = m.p[k].a * 32 + m.p[k].b;   // workgroup-uniform
#define TN(A, B) case A * 32 + B: tsgemm_tn_body<A, B, float, float>(dy, lddy, x, ldx, part, bpart, M, N, K, bx, by, nblk, sm); break
  switch (variant) {
    TN(1, 1); TN(1, 2); TN(1, 4); TN(1, 8); TN(1, 13); TN(1, 16); TN(2, 1); TN(2, 2); TN(2, 4); TN(2, 8); TN(2, 13);
    TN(4, 1); TN(4, 2); TN(4, 4); TN(8, 1); TN(8, 2); TN(13, 1); TN(13, 2); TN(16, 1);
    default: break;
  }
#undef TN
}

int rows_per_wave(const char* name, int dflt) {   // measurement aid: ADNM_TS_NT_RPW / ADNM_TS_TN_RPW override the row blocks per wave
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  return v > 0 ? v : dflt;
}
int nt_blocks(int64_t M, bool bf16) {
  // fp32: ~2 row blocks per wave.  bf16: one — the kernel needs half the registers, so every row block of the 65 536-token level is
  // resident at once (4 workgroups per CU) and the whole input is requested in one round trip
  static const int rpw32 = rows_per_wave("ADNM_TS_NT_RPW", 2), rpw16 = rows_per_wave("ADNM_TS_NT_RPW16", 1);
  const int rpw = bf16 ? rpw16 : rpw32;
  int64_t b = adnm_cdiv(adnm_cdiv(M, 16), kWaves * rpw);
  if (b > 1024) b = 1024;
  return (int)(b < 1 ? 1 : b);
}
// row bands (= partial rows for the fold) of one problem.  Launched alone a problem needs one band per CU; inside the grouped launch the
// other problems fill the chip, and fewer, longer bands amortise a band's tail (LDS tree + store) and halve the fold's reads: measured
// in one session at config 2, 20 problems: 256 bands 328 us, 128 bands 266 us, 64 bands 248 us (ADNM_TS_TN_BLOCKS overrides the grouped figure)
int tn_blocks(int64_t M, bool grouped = false) {
  static const int gcap = rows_per_wave("ADNM_TS_TN_BLOCKS", 64);
  const int cap = grouped ? gcap : 256;
  int64_t b = adnm_cdiv(adnm_cdiv(M, 4), kWaves * 8);  // >= 8 MFMA rounds per wave
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

template <int KQ, int NB>
int launch_nt(const void* x, int64_t ldx, const float* w, int64_t ws_n, int64_t ws_k, const float* bias, void* y, int64_t ldy, int64_t M,
               int N, int K, int prec, AdnmQuant* q, int a_bf16, int c_bf16, hipStream_t st) {
  const bool narrow = prec != ADNM_MFMA_F32;
  const size_t entries = narrow ? (size_t)NB * ((KQ + 1) / 2) * 64 : (size_t)NB * (K / 4) * 64;
  const size_t wbytes = (entries * (prec == ADNM_MFMA_F32 ? 4 : (prec == ADNM_MFMA_BF16 ? 16 : 8)) + 15) / 16 * 16;
  const size_t smem = wbytes + (size_t)kWaves * 16 * 68 * sizeof(float);   // weight image + one 16 x 64 output staging tile per wave
  ADNM_PROF("tsgemm_nt", st, (double)M * (K * (a_bf16 ? 2.0 : 4.0) + N * (c_bf16 ? 2.0 : 4.0)) + 4.0 * (double)N * K);
#define TS_GO(PRECV, BF8V, ATV, CTV)                                                                                                     \
  do {                                                                                                                                   \
    ADNM_ALLOW_LDS((tsgemm_nt_kernel<KQ, NB, PRECV, BF8V, ATV, CTV>), smem, "tsgemm_nt"); /* > 64 KB of dynamic LDS for the widest weights */ \
    tsgemm_nt_kernel<KQ, NB, PRECV, BF8V, ATV, CTV><<<nt_blocks(M, narrow), kBlock, smem, st>>>((const ATV*)x, ldx, w, ws_n, ws_k, bias, (CTV*)y, ldy, M, N, K, q); \
  } while (0)
  if (a_bf16 || c_bf16) {   // bf16 token storage exists for the bf16 matrix-core precision only (host-checked)
    if (a_bf16 && c_bf16) TS_GO(ADNM_MFMA_BF16, false, uint16_t, uint16_t);
    else if (a_bf16) TS_GO(ADNM_MFMA_BF16, false, uint16_t, float);
    else TS_GO(ADNM_MFMA_BF16, false, float, uint16_t);
  } else if (prec == ADNM_MFMA_BF16) TS_GO(ADNM_MFMA_BF16, false, float, float);
  else if (prec == ADNM_MFMA_FP8) TS_GO(ADNM_MFMA_FP8, false, float, float);
  else if (prec == ADNM_MFMA_FP8_GRAD) TS_GO(ADNM_MFMA_FP8, true, float, float);
  else TS_GO(ADNM_MFMA_F32, false, float, float);
#undef TS_GO
  return ADNM_OK;
}

template <int NBN, int NBK>
int launch_tn(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* part, float* bpart, int64_t M, int N, int K, int nblk,
               int ntiles, int y_bf16, int x_bf16, hipStream_t st) {
  const size_t smem = (size_t)(kWaves / 2) * (NBN * NBK * 256 + NBN * 16) * sizeof(float);
  ADNM_PROF("tsgemm_tn", st, (double)M * (K * (x_bf16 ? 2.0 : 4.0) + N * (y_bf16 ? 2.0 : 4.0)) + 4.0 * (double)N * K);
#define TN_GO(YTV, XTV)                                                                                                              \
  do {                                                                                                                               \
    ADNM_ALLOW_LDS((tsgemm_tn_kernel<NBN, NBK, YTV, XTV>), smem, "tsgemm_tn"); /* > 64 KB of dynamic LDS: per-device opt-in */         \
    tsgemm_tn_kernel<NBN, NBK, YTV, XTV><<<dim3(nblk, ntiles), kBlock, smem, st>>>((const YTV*)dy, lddy, (const XTV*)x, ldx, part, bpart, M, N, K); \
  } while (0)
  if (y_bf16 && x_bf16) TN_GO(uint16_t, uint16_t);
  else if (y_bf16) TN_GO(uint16_t, float);
  else if (x_bf16) TN_GO(float, uint16_t);
  else TN_GO(float, float);
#undef TN_GO
  return ADNM_OK;
}

// tile the N axis so that one block keeps at most 32 accumulator blocks: returns NBN (blocks of 16 rows per tile)
// (<= 16 blocks = 64 accumulator VGPRs, 26 for the 13-wide in_proj case — tiling that one to 2 x (8, 2) re-reads X and measured 9 %
// slower; the 32-block variants spill to scratch)
inline int tn_tile(int nbn_total, int nbk) {
  int t = nbk == 13 ? 2 : 16 / nbk;
  if (t > 16) t = 16;
  while (t > 1 && (t & (t - 1))) --t;  // power of two -> an instantiated NBN
  if (t >= nbn_total) return -1;        // fits without tiling
  return t < 1 ? 1 : t;
}

inline int pick(int v, const int* opts, int n) {
  for (int k = 0; k < n; ++k)
    if (v <= opts[k]) return opts[k];
  return -1;
}
const int kKQ[] = {1, 2, 4, 8, 13, 16};
const int kNB[] = {1, 2, 4, 8, 13, 16};

}  // namespace

extern "C" int adnm_tsgemm_supported(int64_t M, int64_t N, int64_t K) {
  if (M < 1 || N < 1 || K < 16 || K % 16 || K > 256 || N > 256) return 0;
  const int nb = pick((int)adnm_cdiv(N, 16), kNB, 6);
  if (pick((int)(K / 16), kKQ, 6) != K / 16) return 0;  // K/16 must be one of the instantiated depths
  if (N * K > 8192) return 0;  // the weight must stay a small, register/LDS-resident operand (refiner-family shapes)
  return nb > 0 && nb * (K / 16) <= 32 ? 1 : 0;   // <= 32 accumulator/operand blocks per wave: the instantiated, spill-free variants
}

// Y[M,N] = X[M,K] . Wp^T (+bias), Wp[n][k] = w[n*ws_n + k*ws_k].  K % 16 == 0, K <= 256, N <= 256.
extern "C" int adnm_tsgemm_nt(const void* x, int64_t ldx, const float* w, int64_t ws_n, int64_t ws_k, const float* bias, void* y,
                              int64_t ldy, int64_t M, int64_t N, int64_t K, int prec, float* q, int x_dtype, int y_dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(x && w && y, "tsgemm_nt: null pointer");
  ADNM_REQUIRE(prec >= ADNM_MFMA_F32 && prec <= ADNM_MFMA_FP8_GRAD, "tsgemm_nt: bad prec %d", prec);
  ADNM_REQUIRE((prec != ADNM_MFMA_FP8 && prec != ADNM_MFMA_FP8_GRAD) || q, "tsgemm_nt: the fp8 modes need a quantisation record");
  ADNM_REQUIRE(adnm_tsgemm_supported(M, N, K), "tsgemm_nt: unsupported shape M=%lld N=%lld K=%lld (need K%%16==0, K<=256, N<=256)", (long long)M,
               (long long)N, (long long)K);
  ADNM_REQUIRE(ldx >= K && ldx % 4 == 0 && ldy >= N, "tsgemm_nt: bad row strides");
  ADNM_REQUIRE((x_dtype == ADNM_F32 || x_dtype == ADNM_BF16) && (y_dtype == ADNM_F32 || y_dtype == ADNM_BF16), "tsgemm_nt: bad storage dtype");
  ADNM_REQUIRE((x_dtype == ADNM_F32 && y_dtype == ADNM_F32) || prec == ADNM_MFMA_BF16, "tsgemm_nt: bf16 token storage needs prec = ADNM_MFMA_BF16");
  ADNM_REQUIRE(((uintptr_t)x & (x_dtype == ADNM_BF16 ? 7 : 15)) == 0, "tsgemm_nt: input rows must be aligned to 4 elements");
  const int kq = pick((int)(K / 16), kKQ, 6), nb = pick((int)adnm_cdiv(N, 16), kNB, 6);
  ADNM_REQUIRE(kq == K / 16, "tsgemm_nt: K/16=%lld not in {1,2,4,8,13,16}", (long long)(K / 16));
  hipStream_t st = (hipStream_t)stream;
  // only the variants adnm_tsgemm_supported admits (N*K <= 8192, i.e. KQ*NB <= 32 blocks: <= 64 accumulator + 64 operand VGPRs,
  // no scratch) are instantiated
  int rc = ADNM_EINVAL;
#define NT(KQ, NB) if (kq == KQ && nb == NB) rc = launch_nt<KQ, NB>(x, ldx, w, ws_n, ws_k, bias, y, ldy, M, (int)N, (int)K, prec, reinterpret_cast<AdnmQuant*>(q), x_dtype == ADNM_BF16, y_dtype == ADNM_BF16, st)
  NT(1, 1); NT(1, 2); NT(1, 4); NT(1, 8); NT(1, 13); NT(1, 16);
  NT(2, 1); NT(2, 2); NT(2, 4); NT(2, 8); NT(2, 13); NT(2, 16);
  NT(4, 1); NT(4, 2); NT(4, 4); NT(4, 8);
  NT(8, 1); NT(8, 2); NT(8, 4);
  NT(13, 1); NT(13, 2);
  NT(16, 1); NT(16, 2);
#undef NT
  ADNM_REQUIRE(rc == ADNM_OK, "tsgemm_nt: no kernel variant for K/16=%d, N/16=%d", kq, nb);
  ADNM_CHECK_LAUNCH("tsgemm_nt");
  return ADNM_OK;
}

namespace {
bool tn_variant(int a, int b) {   // the (N/16, K/16) tiles the grouped kernel's switch holds (= every variant of the single-problem dispatch)
  static const int v[][2] = {{1, 1}, {1, 2}, {1, 4}, {1, 8}, {1, 13}, {1, 16}, {2, 1}, {2, 2}, {2, 4}, {2, 8}, {2, 13}, {4, 1}, {4, 2}, {4, 4}, {8, 1}, {8, 2}, {13, 1}, {13, 2}, {16, 1}};
  for (const auto& e : v)
    if (e[0] == a && e[1] == b) return true;
  return false;
}
}  // namespace

// the queued weight-gradient problems, kMaxTn per launch
int adnm_tsgemm_tn_launch_multi(const AdnmLeaf* const* items, int n, hipStream_t st) {
  int i = 0;
  while (i < n) {
    MultiTn m;
    m.count = 0;
    int blocks = 0;
    double bytes = 0;
    size_t smem = 0;
    for (; i < n && m.count < kMaxTn; ++i) {
      memcpy(&m.p[m.count], items[i]->args, sizeof(TnArgs));
      const TnArgs& t = m.p[m.count];
      const size_t need = (size_t)(kWaves / 2) * (t.a * t.b * 256 + t.a * 16) * sizeof(float);
      smem = need > smem ? need : smem;
      blocks += items[i]->grid;
      m.blk_end[m.count++] = blocks;
      bytes += items[i]->bytes;
    }
    for (int k = m.count; k < kMaxTn; ++k) m.blk_end[k] = blocks;
    ADNM_PROF("tsgemm_tn", st, bytes);
    ADNM_ALLOW_LDS(tsgemm_tn_multi_kernel, smem, "tsgemm_tn (grouped)");
    tsgemm_tn_multi_kernel<<<(unsigned)blocks, kBlock, smem, st>>>(m);
  }
  ADNM_CHECK_LAUNCH("tsgemm_tn (grouped)");
  return ADNM_OK;
}

extern "C" int adnm_tsgemm_tn_supported(int64_t M, int64_t N, int64_t K) {
  if (M < 1 || N < 1 || K < 1 || N > 1024 || K > 256) return 0;
  const int b = pick((int)adnm_cdiv(K, 16), kNB, 6);
  return b > 0 ? 1 : 0;  // N is tiled over blockIdx.y when N/16 * K/16 > 26 accumulator blocks
}
extern "C" int64_t adnm_tsgemm_tn_ws_bytes(int64_t M, int64_t N, int64_t K) { return (int64_t)tn_blocks(M) * (N * K + N) * (int64_t)sizeof(float); }

// dW[N,K] = dY[M,N]^T . X[M,K]  (+ dbias[N] = column sums of dY when dbias != NULL).  OVERWRITES dW / dbias.
extern "C" int adnm_tsgemm_tn(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, float* dbias, void* ws, int64_t ws_bytes,
                              int64_t M, int64_t N, int64_t K, int dy_dtype, int x_dtype, adnm_stream_t stream) {
  ADNM_REQUIRE(dy && x && dw, "tsgemm_tn: null pointer");
  ADNM_REQUIRE(adnm_tsgemm_tn_supported(M, N, K), "tsgemm_tn: unsupported shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  ADNM_REQUIRE(lddy >= N && ldx >= K, "tsgemm_tn: bad row strides");
  ADNM_REQUIRE((dy_dtype == ADNM_F32 || dy_dtype == ADNM_BF16) && (x_dtype == ADNM_F32 || x_dtype == ADNM_BF16), "tsgemm_tn: bad storage dtype");
  if (!ws || ws_bytes < adnm_tsgemm_tn_ws_bytes(M, N, K)) {
    adnm_set_error("tsgemm_tn: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_tsgemm_tn_ws_bytes(M, N, K));
    return ADNM_EWORKSPACE;
  }
  const int b = pick((int)adnm_cdiv(K, 16), kNB, 6);
  const int nbn_total = (int)adnm_cdiv(N, 16);
  int a = pick(nbn_total, kNB, 6), ntiles = 1;
  if (a < 0 || a * b > 26) {
    a = tn_tile(nbn_total, b);
    if (a < 0) a = 1;
    ntiles = (int)adnm_cdiv(nbn_total, a);
  }
  ADNM_REQUIRE(M * lddy < (1ll << 31) && M * ldx < (1ll << 31), "tsgemm_tn: operands beyond 2^31 elements (32-bit lane offsets)");
  const bool leafy = dy_dtype == ADNM_F32 && x_dtype == ADNM_F32 && tn_variant(a, b) && adnm_leafq_active();
  const int nblk = tn_blocks(M, leafy);
  float* part = (float*)ws;
  float* bpart = dbias ? part : nullptr;   // non-null = "emit the bias sums at the end of every partial row"
  hipStream_t st = (hipStream_t)stream;
  int rc = ADNM_EINVAL;
  if (leafy) {   // a leaf: waits for the grouped launch
    static_assert(sizeof(TnArgs) <= sizeof(AdnmLeaf::args), "TnArgs must fit a leaf record");
    TnArgs t;
    t.dy = (const float*)dy, t.x = (const float*)x, t.part = part, t.bpart = bpart, t.lddy = lddy, t.ldx = ldx, t.M = M;
    t.N = (int)N, t.K = (int)K, t.a = a, t.b = b, t.nblk = nblk, t.ntiles = ntiles;
    AdnmLeaf leaf;
    leaf.kind = ADNM_LEAF_TSGEMM_TN, leaf.grid = nblk * ntiles, leaf.prec = ADNM_MFMA_F32, leaf.prof = "tsgemm_tn";
    leaf.bytes = 4.0 * ((double)M * (K + N) + (double)N * K);
    memcpy(leaf.args, &t, sizeof(TnArgs));
    if (adnm_leafq_push(leaf)) rc = ADNM_OK;
  }
#define TN(A, B) if (rc != ADNM_OK) if (a == A && b == B) rc = launch_tn<A, B>(dy, lddy, x, ldx, part, bpart, M, (int)N, (int)K, nblk, ntiles, dy_dtype == ADNM_BF16, x_dtype == ADNM_BF16, st)
  TN(1, 1); TN(1, 2); TN(1, 4); TN(1, 8); TN(1, 13); TN(1, 16); TN(2, 1); TN(2, 2); TN(2, 4); TN(2, 8); TN(2, 13);
  TN(4, 1); TN(4, 2); TN(4, 4); TN(8, 1); TN(8, 2); TN(13, 1); TN(13, 2); TN(16, 1);
#undef TN
  if (rc != ADNM_OK) {
    if (rc == ADNM_EINVAL) adnm_set_error("tsgemm_tn: no kernel variant for N/16=%d, K/16=%d", a, b);
    return rc;
  }
  ADNM_CHECK_LAUNCH("tsgemm_tn");
  adnm_launch_fold("tsgemm_tn_fold", part, nblk, (int)(N * K + (dbias ? N : 0)), {dw, (int)(N * K)}, {dbias, dbias ? (int)N : 0}, {nullptr, 0},
                   {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("tsgemm_tn_fold");
  return ADNM_OK;
}
