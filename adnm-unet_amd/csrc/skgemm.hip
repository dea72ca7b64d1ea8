// K6b — the Linear layers of the DEEP stages (4x4 .. 16x16 maps: M = 64 .. 1024 token rows, 128 .. 4672 features) as
// "short" fp32 GEMMs on the matrix cores.  Reference call sites: Mamba2.in_proj / out_proj (ADNssd.py:309,461),
// FeedForward.project_in / project_out (model_untils.py:193,196), Mlp.fc1/fc2 (:64,67), ConvFFD (:217,221),
// Block.out_proj (ADNMUNet.py:163), StandardAttention.to_qkv / to_out (ADNssd.py:33-34), Channel_Att_Bridge.att* (:744-750).
//
// With M = 64 every weight element is used for 64 rows = 32 FLOP per byte: the fp32 MFMA peak (157 TFLOP/s) and the HBM
// peak meet there, so the job is to stream each weight ONCE at full bandwidth while all 1024 SIMDs issue MFMAs.  Library
// GEMMs pick 32x32..256x256 macro tiles and leave most of the chip idle on these shapes (measured 13 us average, 123 us
// worst).  Here ONE WAVE is the unit of work: a 64x64 output tile (4x4 blocks of v_mfma_f32_16x16x4_f32, exact fp32)
// over one slice of the reduction axis; tiles x slices are spread over the whole chip.  The 2/4/8 waves of a workgroup
// that share a tile sum their slices through LDS (tree, fixed order); what is left of the split goes through fp32
// partials and the shared deterministic fold (no atomics).  A wave's operands come straight from global memory as
// 16-byte loads, double-buffered in registers.
//
//   op NT:  C[M,N] = A[M,K] . W[N,K]^T (+ bias)      forward            both operands contiguous along the reduction
//   op NN:  C[M,K] = A[M,N] . W[N,K]                 input gradient     W contiguous along the OUTPUT axis
//   op TN:  C[N,K] = A[M,N]^T . X[M,K], db = sum_m A  weight gradient    both contiguous along the output axes
//
// An operand that is contiguous along the reduction gives a lane 4 consecutive reduction steps of one tile row per float4;
// one contiguous along the output axis gives a lane ONE step of 4 interleaved tile rows (tile row 4*l+t of block t), so
// the 64-wide tile is a permutation of 64 consecutive rows/columns and the epilogue stores whole float4s.
#include "adnm_common.h"

namespace {

using f32x4 = adnm_f32x4;
constexpr int kBlock = 512;
constexpr int kWaves = kBlock / 64;
constexpr int kT = 4;
constexpr int kBuf = 64 * 64 + 4 * 64;   // one wave's accumulators (+ its bias-gradient lanes) in LDS   // 4 x 4 blocks of 16 x 16 per wave

struct SkArgs {
  const float* A;
  int64_t sa_i, sa_r;   // A(i, r) = A[i*sa_i + r*sa_r]
  const float* B;
  int64_t sb_r, sb_j;   // B(r, j) = B[r*sb_r + j*sb_j]
  const float* bias;    // added to slice 0 (NT), or NULL
  float* C;             // output, or the partial base when nslices > 1
  int64_t ldc, slice_stride;
  float* bsum;          // TN: column sums of A (bias gradient), or NULL
  int64_t bsum_stride;
  int I, J, R;
  int tiles_j, ntiles, nbs, wpt, chunks_per_wave, nchunks;   // nbs block-slices x wpt waves per tile
};

// A_RC / B_RC: operand contiguous along the reduction axis
template <bool A_RC, bool B_RC, bool BF16>
__global__ __launch_bounds__(kBlock) void skgemm_kernel(SkArgs p) {
  __shared__ __attribute__((aligned(16))) float red[(kWaves / 2) * kBuf];
  const int wave = threadIdx.x >> 6;
  const int gw = wave % p.wpt, group = wave / p.wpt;                 // slice within the tile's wave group, group within the block
  const int ts = blockIdx.x * (kWaves / p.wpt) + group;
  const bool active = ts < p.ntiles * p.nbs;
  const int tile = active ? ts % p.ntiles : 0, bslice = active ? ts / p.ntiles : 0;
  const int i0 = (tile / p.tiles_j) * 16 * kT, j0 = (tile % p.tiles_j) * 16 * kT;
  const int lane = threadIdx.x & 63, l15 = lane & 15, kk = lane >> 4;
  const int c0 = active ? (bslice * p.wpt + gw) * p.chunks_per_wave : 0;
  const int c1 = !active ? 0 : (c0 + p.chunks_per_wave < p.nchunks ? c0 + p.chunks_per_wave : p.nchunks);

  // fetch one chunk (16 reduction steps' worth = 4 MFMA steps) of both operands: v[t][e] = value of tile block t at step e.
  // Out-of-range rows / columns read a clamped (valid) address: their products land in accumulator rows / columns that are
  // never stored.  Reduction steps past R (only the TN op has such a tail) are zeroed on the A side when they are USED,
  // so nothing in here waits for a load and all 8 stay in flight under the previous chunk's MFMAs.
  auto fetch = [&](int c, float (&av)[kT][4], float (&bv)[kT][4]) {
    const int r0 = c * 16;
    // operands contiguous along the reduction: a ragged last chunk (R % 16 != 0; R % 4 == 0) re-reads a valid quad; its products are
    // zeroed on the A side when they are used
    const int rq4 = r0 + 4 * kk < p.R ? r0 + 4 * kk : 0;
    if (A_RC) {
#pragma unroll
      for (int t = 0; t < kT; ++t) {
        const int row = i0 + 16 * t + l15, rc = row < p.I ? row : p.I - 1;
        const float4 v = *reinterpret_cast<const float4*>(p.A + (int64_t)rc * p.sa_i + rq4);
        av[t][0] = v.x; av[t][1] = v.y; av[t][2] = v.z; av[t][3] = v.w;
      }
    } else {
      const int row = i0 + 4 * l15, rc = row < p.I ? row : p.I - 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rr = r0 + 4 * kk + e, rq = rr < p.R ? rr : p.R - 1;
        const float4 v = *reinterpret_cast<const float4*>(p.A + (int64_t)rq * p.sa_r + rc);
        av[0][e] = v.x; av[1][e] = v.y; av[2][e] = v.z; av[3][e] = v.w;
      }
    }
    if (B_RC) {
#pragma unroll
      for (int t = 0; t < kT; ++t) {
        const int col = j0 + 16 * t + l15, cc = col < p.J ? col : p.J - 1;
        const float4 v = *reinterpret_cast<const float4*>(p.B + (int64_t)cc * p.sb_j + rq4);
        bv[t][0] = v.x; bv[t][1] = v.y; bv[t][2] = v.z; bv[t][3] = v.w;
      }
    } else {
      const int col = j0 + 4 * l15, cc = col < p.J ? col : p.J - 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rr = r0 + 4 * kk + e, rq = rr < p.R ? rr : p.R - 1;
        const float4 v = *reinterpret_cast<const float4*>(p.B + (int64_t)rq * p.sb_r + cc);
        bv[0][e] = v.x; bv[1][e] = v.y; bv[2][e] = v.z; bv[3][e] = v.w;
      }
    }
  };

  f32x4 acc[kT][kT];
#pragma unroll
  for (int a = 0; a < kT; ++a)
#pragma unroll
    for (int b = 0; b < kT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[kT] = {0.f, 0.f, 0.f, 0.f};
  // ping-pong register buffers, two chunks per trip: the loads of chunk c+1 are issued BEFORE the 64 MFMAs of chunk c (the
  // scheduling barriers keep the compiler from sinking them below), so HBM latency hides under the matrix pipe
  float a0[kT][4], b0[kT][4], a1[kT][4], b1[kT][4];
  auto compute = [&](int c, float (&av)[kT][4], const float (&bv)[kT][4]) {
    if (c * 16 + 16 > p.R) {   // reduction tail (wave-uniform branch): zero the A side, the B side may hold anything finite
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c * 16 + 4 * kk + (A_RC ? 0 : e) >= p.R) av[0][e] = av[1][e] = av[2][e] = av[3][e] = 0.f;
    }
    if (!A_RC && p.bsum) {
#pragma unroll
      for (int t = 0; t < kT; ++t) bs[t] += (av[t][0] + av[t][1]) + (av[t][2] + av[t][3]);
    }
#pragma unroll
    for (int a = 0; a < kT; ++a)
#pragma unroll
      for (int b = 0; b < kT; ++b) {
        // B_RC (the NT op): operands swapped, so the accumulator block is C^T and a lane ends up with four CONSECUTIVE output
        // columns of one row (float4 stores); otherwise the 4 interleaved column blocks already give that.
        // av[.][e] / bv[.][e] = reduction step 4*kk + e of the chunk: one bf16 MFMA (ADNM_MFMA_BF16) or four fp32 ones
        if (B_RC) acc[a][b] = adnm_mfma16<BF16>(bv[b], av[a], acc[a][b]);
        else acc[a][b] = adnm_mfma16<BF16>(av[a], bv[b], acc[a][b]);
      }
  };
  // fetches are unconditional (past the end they re-read the last chunk): a branch around them would make the compiler's
  // wait-count bookkeeping assume the worst path and wait for the NEW loads before the first MFMA
  const int clast = p.nchunks - 1;
  fetch(c0 < clast ? c0 : clast, a0, b0);
  for (int c = c0; c < c1; c += 2) {
    fetch(c + 1 < clast ? c + 1 : clast, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    compute(c, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    fetch(c + 2 < clast ? c + 2 : clast, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < c1) compute(c + 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }

  // sum the wpt slices of the tile: upper half of the group writes, lower half adds (log2(wpt) rounds, fixed order)
  for (int half = p.wpt >> 1; half >= 1; half >>= 1) {
    if (gw >= half && gw < 2 * half) {
      float* buf = red + (group * (p.wpt >> 1) + (gw - half)) * kBuf;
#pragma unroll
      for (int a = 0; a < kT; ++a)
#pragma unroll
        for (int b = 0; b < kT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) buf[((a * kT + b) * 4 + r) * 64 + lane] = acc[a][b][r];
#pragma unroll
      for (int t = 0; t < kT; ++t) buf[64 * 64 + t * 64 + lane] = bs[t];
    }
    __syncthreads();
    if (gw < half) {
      const float* buf = red + (group * (p.wpt >> 1) + gw) * kBuf;
#pragma unroll
      for (int a = 0; a < kT; ++a)
#pragma unroll
        for (int b = 0; b < kT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][b][r] += buf[((a * kT + b) * 4 + r) * 64 + lane];
#pragma unroll
      for (int t = 0; t < kT; ++t) bs[t] += buf[64 * 64 + t * 64 + lane];
    }
    __syncthreads();
  }
  if (!active || gw != 0) return;

  // epilogue.  MFMA C layout: column = lane & 15, row = (lane >> 4) * 4 + reg (local to the 16 x 16 block)
  float* Cp = p.C + (int64_t)bslice * p.slice_stride;
  const bool add_bias = p.bias && bslice == 0;
  if (B_RC) {
    // C^T blocks: column (lane & 15) -> row i of C, row (lane >> 4) * 4 + reg -> column j of C
    const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(Cp) & 15) == 0 && (!add_bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
#pragma unroll
    for (int a = 0; a < kT; ++a) {
      const int ig = i0 + 16 * a + l15;
      if (ig >= p.I) continue;
#pragma unroll
      for (int b = 0; b < kT; ++b) {
        const int jg = j0 + 16 * b + 4 * kk;
        if (vec_ok && jg + 3 < p.J) {
          float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (add_bias) bv4 = *reinterpret_cast<const float4*>(p.bias + jg);
          *reinterpret_cast<float4*>(Cp + (int64_t)ig * p.ldc + jg) =
              make_float4(acc[a][b][0] + bv4.x, acc[a][b][1] + bv4.y, acc[a][b][2] + bv4.z, acc[a][b][3] + bv4.w);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (jg + r < p.J) Cp[(int64_t)ig * p.ldc + jg + r] = acc[a][b][r] + (add_bias ? p.bias[jg + r] : 0.f);
        }
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < kT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = kk * 4 + r;
        const int ig = A_RC ? i0 + 16 * a + il : i0 + 4 * il + a;
        if (ig >= p.I) continue;
        const int jg = j0 + 4 * l15;
        if (jg < p.J) *reinterpret_cast<float4*>(Cp + (int64_t)ig * p.ldc + jg) = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
      }
  }
  if (!A_RC && p.bsum && (tile % p.tiles_j) == 0) {   // bias gradient: the kk lanes hold different reduction rows
#pragma unroll
    for (int t = 0; t < kT; ++t) bs[t] += __shfl_xor(bs[t], 16, 64), bs[t] += __shfl_xor(bs[t], 32, 64);
    const int ig = i0 + 4 * l15;
    if (kk == 0 && ig < p.I) *reinterpret_cast<float4*>(p.bsum + (int64_t)bslice * p.bsum_stride + ig) = make_float4(bs[0], bs[1], bs[2], bs[3]);
  }
}

struct Plan {
  int tiles_i, tiles_j, ntiles, nchunks, wpt, nbs, cpw;
};
// ~2048 waves when the reduction is long enough (>= 1 chunk = 64 MFMAs per wave): up to 8 waves of a block share a tile
// (summed in LDS), further slices go through partials kept below ~2 MB.  direct: the output cannot take partials.
// critical: the result is consumed by the next launch (NT forward, NN input gradient), so the fold of a cross-workgroup split
// sits on the critical path (~6.5 us of launch + kernel): split only when one workgroup per tile would take longer than that
// saves (a wave issues a 16-step chunk in ~0.85 us).  The weight-gradient op's fold is batched and deferred: it always splits.
Plan make_plan(int64_t I, int64_t J, int64_t R, bool direct, bool critical) {
  Plan pl;
  pl.tiles_i = (int)adnm_cdiv(I, 64);
  pl.tiles_j = (int)adnm_cdiv(J, 64);
  pl.ntiles = pl.tiles_i * pl.tiles_j;
  pl.nchunks = (int)adnm_cdiv(R, 16);
  int want = (int)adnm_cdiv(2048, pl.ntiles);
  if (want > pl.nchunks) want = pl.nchunks;
  if (want < 1) want = 1;
  pl.wpt = 1;
  while (pl.wpt * 2 <= want && pl.wpt < kWaves) pl.wpt *= 2;
  int64_t nbs = adnm_cdiv(want, pl.wpt);
  const int64_t by_mem = (int64_t)(2 << 20) / (I * J * 4);
  if (nbs > by_mem) nbs = by_mem;
  if (critical && 0.85 * pl.nchunks / pl.wpt <= 13.0) nbs = 1;
  if (nbs < 1 || direct) nbs = 1;
  pl.cpw = (int)adnm_cdiv(pl.nchunks, nbs * pl.wpt);
  pl.nbs = (int)adnm_cdiv(adnm_cdiv(pl.nchunks, pl.cpw), pl.wpt);
  return pl;
}

int shape_ok(int op, int64_t M, int64_t N, int64_t K) {
  if (M < 1 || N < 4 || K < 4 || M > 65536 || N > 16384 || K > 16384) return 0;
  if (op == ADNM_SKGEMM_NT) return K % 4 == 0;
  if (op == ADNM_SKGEMM_NN) return N % 4 == 0 && K % 4 == 0;
  if (op == ADNM_SKGEMM_TN) return N % 4 == 0 && K % 4 == 0;
  return 0;
}
void dims(int op, int64_t M, int64_t N, int64_t K, int64_t* I, int64_t* J, int64_t* R) {
  if (op == ADNM_SKGEMM_NT) *I = M, *J = N, *R = K;
  else if (op == ADNM_SKGEMM_NN) *I = M, *J = K, *R = N;
  else *I = N, *J = K, *R = M;
}
}  // namespace

extern "C" int adnm_skgemm_supported(int op, int64_t M, int64_t N, int64_t K) { return shape_ok(op, M, N, K); }

extern "C" int64_t adnm_skgemm_ws_bytes(int op, int64_t M, int64_t N, int64_t K) {
  if (!shape_ok(op, M, N, K)) return -1;
  int64_t I, J, R;
  dims(op, M, N, K, &I, &J, &R);
  const Plan pl = make_plan(I, J, R, false, op != ADNM_SKGEMM_TN);
  return pl.nbs > 1 ? (int64_t)pl.nbs * (I * J + I) * (int64_t)sizeof(float) : 16;
}

extern "C" int adnm_skgemm(int op, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, float* c, int64_t ldc, float* dbias,
                           void* ws, int64_t ws_bytes, int64_t M, int64_t N, int64_t K, int prec, adnm_stream_t stream) {
  ADNM_REQUIRE(prec == ADNM_MFMA_F32 || prec == ADNM_MFMA_BF16, "skgemm: bad prec %d", prec);
  ADNM_REQUIRE(a && b && c, "skgemm: null pointer");
  ADNM_REQUIRE(shape_ok(op, M, N, K), "skgemm: unsupported op/shape op=%d M=%lld N=%lld K=%lld", op, (long long)M, (long long)N, (long long)K);
  int64_t I, J, R;
  dims(op, M, N, K, &I, &J, &R);
  ADNM_REQUIRE(ldc >= J && ldc % 4 == 0, "skgemm: bad output row stride");
  ADNM_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= (op == ADNM_SKGEMM_NT ? K : N) && ldb >= K, "skgemm: bad operand row strides");
  ADNM_REQUIRE(!(bias && op != ADNM_SKGEMM_NT) && !(dbias && op != ADNM_SKGEMM_TN), "skgemm: bias only with NT, dbias only with TN");
  ADNM_REQUIRE(((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16 == 0, "skgemm: operands must be 16-byte aligned");
  const Plan pl = make_plan(I, J, R, ldc != J, op != ADNM_SKGEMM_TN);   // a strided output (column slice of a wider buffer) takes no partials
  const bool split = pl.nbs > 1;
  if (split && (!ws || ws_bytes < adnm_skgemm_ws_bytes(op, M, N, K))) {
    adnm_set_error("skgemm: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_skgemm_ws_bytes(op, M, N, K));
    return ADNM_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  SkArgs p;
  p.A = a, p.B = b, p.bias = bias;
  if (op == ADNM_SKGEMM_NT) p.sa_i = lda, p.sa_r = 1, p.sb_r = 1, p.sb_j = ldb;
  else if (op == ADNM_SKGEMM_NN) p.sa_i = lda, p.sa_r = 1, p.sb_r = ldb, p.sb_j = 1;
  else p.sa_i = 1, p.sa_r = lda, p.sb_r = ldb, p.sb_j = 1;
  p.I = (int)I, p.J = (int)J, p.R = (int)R;
  p.tiles_j = pl.tiles_j, p.ntiles = pl.ntiles, p.nbs = pl.nbs, p.wpt = pl.wpt, p.chunks_per_wave = pl.cpw, p.nchunks = pl.nchunks;
  float* part = (float*)ws;
  p.C = split ? part : c;
  p.ldc = split ? J : ldc;
  const int64_t rowlen = I * J + (dbias ? I : 0);   // a partial row = [output tile rows | bias sums]: one fold for both
  p.slice_stride = split ? rowlen : 0;
  p.bsum = dbias ? (split ? part + I * J : dbias) : nullptr;
  p.bsum_stride = split ? rowlen : 0;
  const unsigned grid = (unsigned)adnm_cdiv((int64_t)pl.ntiles * pl.nbs, kWaves / pl.wpt);
  {
    ADNM_PROF(op == ADNM_SKGEMM_NT ? "skgemm_nt" : (op == ADNM_SKGEMM_NN ? "skgemm_nn" : "skgemm_tn"), st, 4.0 * ((double)M * (K + N) + (double)N * K));
    if (prec == ADNM_MFMA_BF16) {
      if (op == ADNM_SKGEMM_NT) skgemm_kernel<true, true, true><<<grid, kBlock, 0, st>>>(p);
      else if (op == ADNM_SKGEMM_NN) skgemm_kernel<true, false, true><<<grid, kBlock, 0, st>>>(p);
      else skgemm_kernel<false, false, true><<<grid, kBlock, 0, st>>>(p);
    } else {
      if (op == ADNM_SKGEMM_NT) skgemm_kernel<true, true, false><<<grid, kBlock, 0, st>>>(p);
      else if (op == ADNM_SKGEMM_NN) skgemm_kernel<true, false, false><<<grid, kBlock, 0, st>>>(p);
      else skgemm_kernel<false, false, false><<<grid, kBlock, 0, st>>>(p);
    }
  }
  ADNM_CHECK_LAUNCH("skgemm");
  if (split) {
    adnm_launch_fold("skgemm_fold", part, pl.nbs, (int)rowlen, {c, (int)(I * J)}, {dbias, dbias ? (int)I : 0}, {nullptr, 0}, {nullptr, 0}, st);
    ADNM_CHECK_LAUNCH("skgemm_fold");
  }
  return ADNM_OK;
}
