// K6b — the Linear layers of the DEEP stages (4x4 .. 32x32 maps: M = 4 .. 1024 token rows, 128 .. 4672 features) as "short" GEMMs on
// the matrix cores, fp32 in memory.  Reference call sites: Mamba2.in_proj / out_proj (ADNssd.py:309,461), FeedForward.project_in /
// project_out (model_untils.py:193,196), Mlp.fc1/fc2 (:64,67), ConvFFD (:217,221), Block.out_proj (ADNMUNet.py:163),
// StandardAttention.to_qkv / to_out (ADNssd.py:33-34), Channel_Att_Bridge.att* (:744-750), UpSample's ConvTranspose2d as a GEMM.
//
//   op NT:  C[M,N] = A[M,K] . W[N,K]^T (+ bias)      forward            both operands contiguous along the reduction ("RC")
//   op NN:  C[M,K] = A[M,N] . W[N,K]                 input gradient     W contiguous along the OUTPUT axis ("OC")
//   op TN:  C[N,K] = A[M,N]^T . X[M,K], db = sum_m A  weight gradient    both contiguous along the output axes
//
// These problems are small (0.1 - 2 GFLOP, 1 - 20 MB) and every one sits on the step's critical path, so what decides the time is
// how many CUs pull operand bytes at once and how many bytes each keeps in flight — not the MFMA rate.  Two kernels share the work
// (adnm_skgemm picks one per shape, statically: the same choice in every process and on every rank):
//
//   * this file, the REGISTER-STREAMING kernel: one wave owns a (16 TM) x (16 TN) output tile over one slice of the reduction and
//     loads its MFMA operands straight from global memory into registers (16-byte loads, three register buffers = two 16 KC-step
//     chunks in flight under the MFMAs of the third), no LDS staging and no barrier in the loop.  The 2 / 4 / 8 waves of a workgroup
//     that share a tile take different slices and sum them through LDS (tree, fixed order).  Small outputs take small tiles
//     (TM x TN = 1x1, 2x2: more tiles -> more CUs; the operands they re-read come from L2), large ones 4x4.  A reduction that still
//     leaves CUs idle is split over workgroups: the weight-gradient op (TN) writes fp32 partials for the shared fold kernel, whose
//     launch is batched with the other parameter-gradient folds and deferred off the critical path; NT / NN combine their slabs
//     INSIDE the launch (arrival tickets; the last workgroup of a tile adds the slabs in slice order: bitwise reproducible).
//   * lgemm.hip, the LDS-TILED kernel, for outputs of >= ~128 64x64 tiles, where sharing each operand tile between four waves halves
//     the L2 -> CU traffic that bounds those shapes.
//
// An operand that is contiguous along the reduction gives a lane 4 consecutive reduction steps of one tile row per float4; one
// contiguous along the output axis gives a lane ONE step of 4 interleaved tile rows (tile row 4*l+t of block t; such an operand is
// always 4 blocks = 64 wide), so the 64-wide tile is a permutation of 64 consecutive rows/columns and the epilogue stores float4s.
// Which reduction steps a lane feeds to which MFMA is a permutation shared by both operands ("k-permutation").
#include "adnm_common.h"

namespace {

using f32x4 = adnm_f32x4;
constexpr int kBlock = 512;
constexpr int kWaves = kBlock / 64;

struct SkArgs {
  const float* A;
  int64_t sa_i, sa_r;   // A(i, r) = A[i*sa_i + r*sa_r]
  const void* B;        // fp32, or the weight's narrow shadow (BT of skgemm_body: bf16 / scaled e4m3)
  int64_t sb_r, sb_j;   // B(r, j) = B[r*sb_r + j*sb_j]   (element strides)
  const float* b_scale; // fp8 shadow: the weight's scale (value = e4m3 / *b_scale); else NULL
  const float* bias;    // NT: added once, or NULL
  float* C;             // output; with nbs > 1 the slab base ([slice][I*J (+I)])
  int64_t ldc, slice_stride;
  float* bsum;          // TN: column sums of A (bias gradient), or NULL
  int64_t bsum_stride;
  int I, J, R;
  int tiles_j, ntiles, nbs, wpt, chunks_per_wave, nchunks;   // nbs workgroup-slices x wpt waves per tile; a chunk = 16 KC steps
  int* tickets;         // NT / NN with nbs > 1: one arrival counter per output tile (zero when idle); the result goes to `out`
  int uc;               // the slabs live in uncached memory (the caller said so: ws_uncached): no agent-scope fences around the ticket
  float* out;
  int64_t ldo;
  AdnmQuant* q;         // quantisation record (fp8 scales, amax collection) or NULL
};

// A_RC / B_RC: operand contiguous along the reduction axis.  TM x TN blocks of 16 x 16 per wave, KC 16-step groups per chunk.
// PREC: ADNM_MFMA_* (the matrix-core precision); A_BF8: in the fp8 mode the A operand is a gradient (e5m2).
// (bid, nblk): this workgroup's index and the workgroup count of ITS problem — the whole grid of the single-problem launch, a block range of
// the grouped weight-gradient launch (skgemm_tn_multi_kernel).  Args: SkArgs, possibly in the constant address space.
template <bool A_RC, bool B_RC, int PREC, bool A_BF8, int TM, int TN, int KC, int BT = ADNM_B_F32, typename Args>
__device__ __forceinline__ void skgemm_body(const Args& p, const int bid, const int nblk) {
  static_assert(BT == ADNM_B_F32 || (BT == ADNM_B_BF16 && PREC == ADNM_MFMA_BF16) || (BT == ADNM_B_FP8 && PREC == ADNM_MFMA_FP8),
                "a narrow weight shadow feeds the matrix-core precision it was made for");
  static_assert((A_RC || TM == 4) && (B_RC || TN == 4), "an operand contiguous along the output axis is 4 interleaved blocks wide");
  constexpr int kAcc = TM * TN * 256;        // one wave's accumulators ...
  constexpr int kBuf = kAcc + 4 * 64;        // ... + its bias-gradient lanes, in LDS
  __shared__ __attribute__((aligned(16))) float red[(kWaves / 2) * kBuf];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and the compiler should know it
  const int gw = wave % p.wpt, group = wave / p.wpt;                 // slice within the tile's wave group, group within the block
  int ts = bid * (kWaves / p.wpt) + group;
  const bool active = ts < p.ntiles * p.nbs;
  int tile, bslice;
  if (p.tickets) {
    // one tile slice per workgroup.  Workgroups go round-robin over the 8 XCDs: give each XCD a contiguous run of (tile, slice) pairs so
    // that the slices of a tile (whose slabs the last of them reads back) and neighbouring tiles (which share A rows) meet in one L2.
    // A speed choice only: the combine below is correct for any placement.
    if ((nblk & 7) == 0) ts = (bid & 7) * (nblk >> 3) + (bid >> 3);
    tile = ts / p.nbs, bslice = ts % p.nbs;
  } else {
    tile = active ? ts % p.ntiles : 0, bslice = active ? ts / p.ntiles : 0;
  }
  const int i0 = (tile / p.tiles_j) * 16 * TM, j0 = (tile % p.tiles_j) * 16 * TN;
  const int lane = threadIdx.x & 63, l15 = lane & 15, kk = lane >> 4;
  const int c0 = active ? (bslice * p.wpt + gw) * p.chunks_per_wave : 0;
  const int c1 = !active ? 0 : (c0 + p.chunks_per_wave < p.nchunks ? c0 + p.chunks_per_wave : p.nchunks);

  // fetch one chunk (KC groups of 16 reduction steps = 4 MFMA steps each) of both operands: v[t][s][e] = value of tile block t at
  // step e of the lane's quad in group s.  Out-of-range rows / columns read a clamped (valid) address: their products land in
  // accumulator rows / columns that are never stored.  Reduction steps past R are zeroed on the A side when they are USED, so nothing
  // in here waits for a load and every load stays in flight under the previous chunk's MFMAs.
  // `live` (wave-uniform): a fetch past the wave's last chunk still issues its loads (a branch around them would make the compiler's
  // wait counts assume the worst path), but all lanes read ONE 16-byte word: a single broadcast request instead of 16 line requests.
  auto fetch = [&](int c, bool live, float (&av)[TM][KC][4], float (&bv)[TN][KC][4]) {
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int r0 = (c * KC + s) * 16;
      // operands contiguous along the reduction: a group past R (R % 4 == 0) re-reads a valid quad
      const int rq4 = r0 + 4 * kk < p.R ? r0 + 4 * kk : 0;
      if constexpr (A_RC) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = i0 + 16 * t + l15, rc = row < p.I ? row : p.I - 1;
          const float4 v = *reinterpret_cast<const float4*>(p.A + (live ? (int64_t)rc * p.sa_i + rq4 : 0));
          av[t][s][0] = v.x, av[t][s][1] = v.y, av[t][s][2] = v.z, av[t][s][3] = v.w;
        }
      } else {
        const int row = i0 + 4 * l15, rc = row < p.I ? row : p.I - 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rr = r0 + 4 * kk + e, rq = rr < p.R ? rr : p.R - 1;
          const float4 v = *reinterpret_cast<const float4*>(p.A + (live ? (int64_t)rq * p.sa_r + rc : 0));
          av[0][s][e] = v.x, av[1 % TM][s][e] = v.y, av[2 % TM][s][e] = v.z, av[3 % TM][s][e] = v.w;
        }
      }
      if constexpr (B_RC) {
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int col = j0 + 16 * t + l15, cc = col < p.J ? col : p.J - 1;
          const float4 v = adnm_ldb4<BT>(p.B, live ? (int64_t)cc * p.sb_j + rq4 : 0);
          bv[t][s][0] = v.x, bv[t][s][1] = v.y, bv[t][s][2] = v.z, bv[t][s][3] = v.w;
        }
      } else {
        const int col = j0 + 4 * l15, cc = col < p.J ? col : p.J - 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rr = r0 + 4 * kk + e, rq = rr < p.R ? rr : p.R - 1;
          const float4 v = adnm_ldb4<BT>(p.B, live ? (int64_t)rq * p.sb_r + cc : 0);
          bv[0][s][e] = v.x, bv[1 % TN][s][e] = v.y, bv[2 % TN][s][e] = v.z, bv[3 % TN][s][e] = v.w;
        }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  // fp8: per-tensor scales from the call site's quantisation record; the accumulators are un-scaled in the epilogue.  rec_a / rec_b
  // (wave-uniform): this wave collects max |value| of the A rows / B columns it reads — only the first tile column / tile row do, so every
  // element is seen once per reduction slice and the atomics stay a handful per launch.
  // (an fp8 SHADOW is already scaled and rounded: its values pass through the fragment builder with scale 1 and come out bit for bit;
  // q_sbo = the scale the accumulators are divided by)
  float q_sa = 1.f, q_sb = 1.f, q_sbo = 1.f, amax_a = 0.f, amax_b = 0.f;
  bool rec_a = false, rec_b = false;
  if (p.q) {
    if (PREC == ADNM_MFMA_FP8) q_sa = p.q->scale_a, q_sb = q_sbo = p.q->scale_b;
    const bool rec = p.q->record != 0.f && active;
    rec_a = rec && (tile % p.tiles_j) == 0, rec_b = BT == ADNM_B_F32 && rec && (tile / p.tiles_j) == 0;
  }
  if (BT == ADNM_B_FP8) q_sb = 1.f, q_sbo = *p.b_scale;
  auto compute = [&](int c, float (&av)[TM][KC][4], const float (&bv)[TN][KC][4]) {
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int r0 = (c * KC + s) * 16;
      if (r0 + 16 > p.R) {   // reduction tail (wave-uniform branch): zero the A side, the B side may hold anything finite
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (r0 + 4 * kk + (A_RC ? 0 : e) >= p.R) {
#pragma unroll
            for (int t = 0; t < TM; ++t) av[t][s][e] = 0.f;
          }
      }
      if (!A_RC && p.bsum) {
#pragma unroll
        for (int t = 0; t < TM; ++t) bs[t] += (av[t][s][0] + av[t][s][1]) + (av[t][s][2] + av[t][s][3]);
      }
      if (rec_a) {
#pragma unroll
        for (int t = 0; t < TM; ++t) amax_a = adnm_amax4(amax_a, av[t][s][0], av[t][s][1], av[t][s][2], av[t][s][3]);
      }
      if (rec_b && r0 + 16 <= p.R) {   // (B's tail steps are not zeroed: skip the group that holds them; the weight's amax comes from the rest)
#pragma unroll
        for (int t = 0; t < TN; ++t) amax_b = adnm_amax4(amax_b, bv[t][s][0], bv[t][s][1], bv[t][s][2], bv[t][s][3]);
      }
    }
    // one MFMA step = 32 reduction steps = two 16-step groups (an odd last group runs on a zero upper half); every fragment is scaled /
    // rounded / packed once, then used by the TN (TM) MFMAs that read it
#pragma unroll
    for (int s = 0; s < KC; s += 2) {
      constexpr bool kHalf = (KC & 1) != 0;   // KC == 1: the chunk is one group
      AdnmFrag<PREC> fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = adnm_make_frag<PREC, A_BF8>(av[a][s], kHalf ? nullptr : av[a][(s + 1) % KC], q_sa);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = adnm_make_frag<PREC, false>(bv[b][s], kHalf ? nullptr : bv[b][(s + 1) % KC], q_sb);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          // B_RC (the NT op): operands swapped, so the accumulator block is C^T and a lane ends up with four CONSECUTIVE output
          // columns of one row (float4 stores); otherwise the 4 interleaved column blocks already give that.
          if (B_RC) acc[a][b] = adnm_mma<PREC, false, A_BF8, kHalf>(fb[b], fa[a], acc[a][b]);
          else acc[a][b] = adnm_mma<PREC, A_BF8, false, kHalf>(fa[a], fb[b], acc[a][b]);
        }
    }
  };
  // three register buffers, three chunks per trip: the loads of chunks c+1 and c+2 are in flight under the MFMAs of chunk c (the
  // scheduling barriers keep the compiler from sinking them below), so two HBM/L2 round trips overlap per wave
  float a0[TM][KC][4], b0[TN][KC][4], a1[TM][KC][4], b1[TN][KC][4], a2[TM][KC][4], b2[TN][KC][4];
  const int cl = c1 > c0 ? c1 - 1 : 0;
  auto fetch_at = [&](int c, float (&av)[TM][KC][4], float (&bv)[TN][KC][4]) { fetch(c < c1 ? c : cl, c < c1, av, bv); };
  fetch_at(c0, a0, b0);
  __builtin_amdgcn_sched_barrier(0);
  fetch_at(c0 + 1, a1, b1);
  __builtin_amdgcn_sched_barrier(0);
  for (int c = c0; c < c1; c += 3) {
    fetch_at(c + 2, a2, b2);
    __builtin_amdgcn_sched_barrier(0);
    compute(c, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    fetch_at(c + 3, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < c1) compute(c + 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    fetch_at(c + 4, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 2 < c1) compute(c + 2, a2, b2);
    __builtin_amdgcn_sched_barrier(0);
  }

  if (rec_a) adnm_amax_commit(&p.q->amax_a, amax_a);
  if (rec_b) adnm_amax_commit(&p.q->amax_b, amax_b);
  if (PREC == ADNM_MFMA_FP8) {   // back to the operands' own scale (before slices are summed, bias is added or slabs are written)
    const float inv = 1.0f / (q_sa * q_sbo);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = acc[a][b] * inv;
  }
  // sum the wpt slices of the tile: upper half of the group writes, lower half adds (log2(wpt) rounds, fixed order)
  for (int half = p.wpt >> 1; half >= 1; half >>= 1) {
    if (gw >= half && gw < 2 * half) {
      float* buf = red + (group * (p.wpt >> 1) + (gw - half)) * kBuf;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) buf[((a * TN + b) * 4 + r) * 64 + lane] = acc[a][b][r];
      if (!A_RC) {
#pragma unroll
        for (int t = 0; t < 4; ++t) buf[kAcc + t * 64 + lane] = bs[t];
      }
    }
    __syncthreads();
    if (gw < half) {
      const float* buf = red + (group * (p.wpt >> 1) + gw) * kBuf;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][b][r] += buf[((a * TN + b) * 4 + r) * 64 + lane];
      if (!A_RC) {
#pragma unroll
        for (int t = 0; t < 4; ++t) bs[t] += buf[kAcc + t * 64 + lane];
      }
    }
    __syncthreads();
  }
  if (!p.tickets && (!active || gw != 0)) return;

  if (gw == 0) {
    // epilogue.  MFMA C layout: column = lane & 15, row = (lane >> 4) * 4 + reg (local to the 16 x 16 block)
    float* Cp = p.C + (int64_t)bslice * p.slice_stride;
    const bool add_bias = p.bias && bslice == 0 && !p.tickets;
    if constexpr (B_RC) {
      // C^T blocks: column (lane & 15) -> row i of C, row (lane >> 4) * 4 + reg -> column j of C
      const bool vec_ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(Cp) & 15) == 0 && (!add_bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int ig = i0 + 16 * a + l15;
        if (ig >= p.I) continue;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
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
    } else {   // TN == 4: the four column blocks are interleaved
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int il = kk * 4 + r;
          const int ig = A_RC ? i0 + 16 * a + il : i0 + 4 * il + a;
          if (ig >= p.I) continue;
          const int jg = j0 + 4 * l15;
          if (jg < p.J)
            *reinterpret_cast<float4*>(Cp + (int64_t)ig * p.ldc + jg) = make_float4(acc[a][0][r], acc[a][1 % TN][r], acc[a][2 % TN][r], acc[a][3 % TN][r]);
        }
    }
    if (!A_RC && p.bsum && (tile % p.tiles_j) == 0) {   // bias gradient: the kk lanes hold different reduction rows
#pragma unroll
      for (int t = 0; t < 4; ++t) bs[t] += __shfl_xor(bs[t], 16, 64), bs[t] += __shfl_xor(bs[t], 32, 64);
      const int ig = i0 + 4 * l15;
      if (kk == 0 && ig < p.I) *reinterpret_cast<float4*>(p.bsum + (int64_t)bslice * p.bsum_stride + ig) = make_float4(bs[0], bs[1], bs[2], bs[3]);
    }
  }
  if (!p.tickets) return;

  // In-launch combine of the nbs slabs of this tile (one workgroup = one slice, wpt == kWaves; wave 0 has just stored this slice's slab
  // with plain stores).  Publish: wave 0 drains its stores, lane 0 releases at agent scope and draws a ticket; the workgroup that draws
  // the last one acquires at agent scope and all of its waves add the slabs in slice order 0 .. nbs-1 (its own included, re-read from
  // memory), so the sum does not depend on which slice arrived last: the result is bitwise reproducible.  The counter goes back to zero
  // for the next launch on the stream.  Correct wherever the slices ran (any CU / XCD).
  int* flag = reinterpret_cast<int*>(red);   // the reduction buffer is idle by now (last read before the barrier above)
  if (wave == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      if (!p.uc) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const int drawn = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (drawn == p.nbs - 1) {
        if (!p.uc) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      *flag = drawn;
    }
  }
  __syncthreads();
  if (*flag != p.nbs - 1) return;
  {
    constexpr int kQuads = 4 * TN;   // float4 columns of the tile; a thread takes one quad of every (kBlock / kQuads)-th row
    const int jg = j0 + 4 * (threadIdx.x % kQuads);
    if (jg >= p.J) return;   // J % 4 == 0 (host: no split otherwise)
    for (int il = threadIdx.x / kQuads; il < 16 * TM; il += kBlock / kQuads) {
      const int ig = i0 + il;
      if (ig >= p.I) break;
      const float* src = p.C + (int64_t)ig * p.ldc + jg;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int s = 0; s < p.nbs; ++s) {
        const float4 u = *reinterpret_cast<const float4*>(src + (int64_t)s * p.slice_stride);
        sum.x += u.x, sum.y += u.y, sum.z += u.z, sum.w += u.w;
      }
      if (p.bias) sum.x += p.bias[jg], sum.y += p.bias[jg + 1], sum.z += p.bias[jg + 2], sum.w += p.bias[jg + 3];
      *reinterpret_cast<float4*>(p.out + (int64_t)ig * p.ldo + jg) = sum;
    }
  }
}

template <bool A_RC, bool B_RC, int PREC, bool A_BF8, int TM, int TN, int KC, int BT>
__global__ __launch_bounds__(kBlock) void skgemm_kernel(SkArgs p) {
  skgemm_body<A_RC, B_RC, PREC, A_BF8, TM, TN, KC, BT>(p, (int)blockIdx.x, (int)gridDim.x);
}

// Grouped weight gradients: the TN problems of a backward pass are leaves (nothing reads dW before the optimiser) and individually small
// (a few tiles x a few reduction slices: 70 launches of ~10 us at config 2), so a caller that has bound a leaf queue gets them QUEUED and
// launched up to kMaxSk at a time — every workgroup looks up its problem in the descriptor table (kernel arguments, constant address space)
// and runs the unchanged tile code on it.  Same arithmetic per problem: bitwise the same results as separate launches.
constexpr int kMaxSk = 16;
struct MultiSk {
  int count, blk_end[kMaxSk];
  SkArgs p[kMaxSk];
};
static_assert(sizeof(MultiSk) <= 4096, "the descriptor table must fit the kernel argument segment");
template <int PREC>
__global__ __launch_bounds__(kBlock) void skgemm_tn_multi_kernel(MultiSk by_value) {
  (void)by_value;
  const auto& m = *(const __attribute__((address_space(4))) MultiSk*)__builtin_amdgcn_kernarg_segment_ptr();
  int k = 0;
  while (k + 1 < m.count && (int)blockIdx.x >= m.blk_end[k]) ++k;
  const int b0 = k ? m.blk_end[k - 1] : 0;
  skgemm_body<false, false, PREC, false, 4, 4, 1>(m.p[k], (int)blockIdx.x - b0, m.blk_end[k] - b0);
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
enum { KERNEL_STREAM = 0, KERNEL_LDS = 1 };
struct Plan {
  int kernel;            // KERNEL_STREAM (this file) or KERNEL_LDS (lgemm.hip); < 0: invalid forced configuration
  int tm, tn, kc;        // wave tile (16 tm) x (16 tn), 16 kc reduction steps per chunk
  int tiles_i, tiles_j, ntiles, nchunks, wpt, nbs, cpw;
  bool combine;          // nbs > 1 and the slabs are summed inside the launch (tickets) instead of by the fold kernel
};

void finish(Plan& pl, int64_t I, int64_t J, int64_t R) {
  pl.tiles_i = (int)adnm_cdiv(I, 16 * pl.tm);
  pl.tiles_j = (int)adnm_cdiv(J, 16 * pl.tn);
  pl.ntiles = pl.tiles_i * pl.tiles_j;
  pl.nchunks = (int)adnm_cdiv(R, 16 * pl.kc);
}
void slice(Plan& pl) {   // chunks per wave for (wpt, nbs), and the slice count that is really needed for it
  if (pl.nbs < 1) pl.nbs = 1;
  pl.cpw = (int)adnm_cdiv(pl.nchunks, pl.nbs * pl.wpt);
  pl.nbs = (int)adnm_cdiv(adnm_cdiv(pl.nchunks, pl.cpw), pl.wpt);
}

// measurement aid (tools/kbench_gemm.py sweeps it): ADNM_SK_FORCE="kernel,tm,tn,kc,wpt,nbs" overrides the NT / NN plan
struct Forced {
  bool on = false;
  int v[6] = {0, 0, 0, 0, 0, 0};
  Forced() {
    const char* e = getenv("ADNM_SK_FORCE");
    on = e && sscanf(e, "%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6;
  }
};
const Forced& forced() {
  static const Forced f;
  return f;
}
int use_table() {   // measurement aid: ADNM_SK_TABLE=0 plans every shape by the rules (what shapes outside the table get)
  static const int v = [] {
    const char* e = getenv("ADNM_SK_TABLE");
    return e && e[0] == '0' ? 0 : 1;
  }();
  return v;
}
bool config_ok(int op, const Plan& pl) {
  if (pl.kernel == KERNEL_LDS) return op != ADNM_SKGEMM_TN;
  if (pl.kernel != KERNEL_STREAM || pl.wpt < 1 || pl.wpt > kWaves || (pl.wpt & (pl.wpt - 1))) return false;
  const int tm = pl.tm, tn = pl.tn, kc = pl.kc;
  if (op == ADNM_SKGEMM_NT) return (tm == 1 && tn == 1 && kc == 4) || (tm == 2 && tn == 2 && kc == 2) || (tm == 4 && tn == 4 && kc == 1);
  if (op == ADNM_SKGEMM_NN) return tn == 4 && ((tm == 1 && kc == 2) || (tm == 2 && kc == 2) || (tm == 4 && kc == 1));
  return tm == 4 && tn == 4 && kc == 1;
}

// Forward / input-gradient ops (NT, NN): the result is consumed by the next launch, so the whole chip should work on it and no second
// launch should stand between.  The shapes of BASELINE config 2 take the configuration that measured fastest on an MI355X
// (skgemm_tuned.inc, made by tools/tune_gemm.py from tools/kbench_gemm.py sweeps); every other shape follows the rules below, which
// summarise that table:
//   NT: the LDS-tiled kernel when there are >= 128 tiles of 64 x 64 or the reduction is <= 128 steps; with a reduction >= 1024 steps
//       and fewer tiles, the LDS-tiled kernel split ~192 / tiles ways (combined in the launch); otherwise the streaming kernel with the
//       largest of the 32 x 32 / 16 x 16 wave tiles that still gives >= 128 tiles, 8 waves per tile.
//   NN: the LDS-tiled kernel for >= 256 tiles or a reduction <= 128 steps; otherwise the streaming kernel, 16 (32 for >= 100 tiles of
//       >= 256 rows) x 64 wave tiles, 8 waves per tile.
struct Tuned {
  int op, I, J, R, bf16;   // op 0 NT, 1 NN, 2 TN
  int cfg[6];   // kernel, tm, tn, kc, wpt, nbs
};
const Tuned kTuned[] = {
#include "skgemm_tuned.inc"
};
const Tuned* tuned(int op, int64_t I, int64_t J, int64_t R, bool bf16) {
  for (const Tuned& t : kTuned)
    if (t.op == op && t.I == I && t.J == J && t.R == R && t.bf16 == (int)bf16) return &t;
  return nullptr;
}
// Weight-gradient op (TN): 64 x 64 tiles (both operands are contiguous along their output axes).  The (waves per tile, slices) pair of a
// config-2 shape comes from the measured table; otherwise ~2048 waves when the reduction is long enough (>= 1 chunk = 64 MFMAs per wave):
// up to 8 waves of a block share a tile (summed in LDS), further slices go through partials kept below ~2 MB and the shared fold
// kernel, whose launch is batched with the other parameter-gradient folds and deferred off the critical path.
// direct: the output cannot take partials.
Plan plan_deferred(int64_t I, int64_t J, int64_t R, bool direct, bool bf16) {
  Plan pl;
  pl.kernel = KERNEL_STREAM, pl.combine = false, pl.tm = pl.tn = 4, pl.kc = 1;
  finish(pl, I, J, R);
  const bool force = forced().on;
  const Tuned* tu = force || use_table() == 0 ? nullptr : tuned(ADNM_SKGEMM_TN, I, J, R, bf16);
  if (force || tu) {
    const int* v = force ? forced().v : tu->cfg;
    pl.wpt = v[4], pl.nbs = v[5];
    if (v[0] != KERNEL_STREAM || v[1] != 4 || v[2] != 4 || v[3] != 1 || pl.wpt < 1 || pl.wpt > kWaves || (pl.wpt & (pl.wpt - 1))) {
      pl.kernel = -1;
      return pl;
    }
    while (pl.wpt > 1 && pl.wpt > pl.nchunks) pl.wpt >>= 1;
  } else {
    // waves per problem: alone, a weight gradient needs every CU (2048 waves); queued for the grouped launch the other problems fill
    // the chip, and fewer reduction slices mean fewer partial slabs to write and fold (ADNM_SK_TN_WAVES: measurement aid)
    static const int grouped_waves = [] {
      const char* e = getenv("ADNM_SK_TN_WAVES");
      const int v = e ? atoi(e) : 0;
      return v > 0 ? v : 256;
    }();
    int want = (int)adnm_cdiv(adnm_leafq_active() ? grouped_waves : 2048, pl.ntiles);
    if (want > pl.nchunks) want = pl.nchunks;
    if (want < 1) want = 1;
    pl.wpt = 1;
    while (pl.wpt * 2 <= want && pl.wpt < kWaves) pl.wpt *= 2;
    pl.nbs = (int)adnm_cdiv(want, pl.wpt);
  }
  if (!force && !tu) {
    const int64_t by_mem = (int64_t)(2 << 20) / (I * J * 4);
    if (pl.nbs > by_mem) pl.nbs = (int)by_mem;
  }
  if (pl.nbs > 64) pl.nbs = 64;
  if (pl.nbs < 1 || direct) pl.nbs = 1;
  slice(pl);
  return pl;
}
Plan plan_critical(int op, int64_t I, int64_t J, int64_t R, bool can_split, bool bf16) {
  Plan pl;
  pl.combine = false;
  const bool force = forced().on;
  const Tuned* tu = force || use_table() == 0 ? nullptr : tuned(op, I, J, R, bf16);
  if (force || tu) {
    const int* v = force ? forced().v : tu->cfg;
    pl.kernel = v[0], pl.tm = v[1], pl.tn = v[2], pl.kc = v[3], pl.wpt = v[4], pl.nbs = v[5];
    if (!config_ok(op, pl)) pl.kernel = -1;
  } else {
    const int64_t t64 = adnm_cdiv(I, 64) * adnm_cdiv(J, 64);
    pl.nbs = 1, pl.wpt = kWaves;
    if (op == ADNM_SKGEMM_NT) {
      pl.kernel = (t64 >= 128 || R <= 128 || (R >= 1024 && t64 >= 8)) ? KERNEL_LDS : KERNEL_STREAM;
      if (pl.kernel == KERNEL_LDS && R >= 1024 && t64 <= 128) {
        int want = (int)(192 / t64);
        while (pl.nbs * 2 <= want && pl.nbs < 8) pl.nbs *= 2;
      }
      pl.tm = pl.tn = 2, pl.kc = 2;
      if (adnm_cdiv(I, 32) * adnm_cdiv(J, 32) < 128) pl.tm = pl.tn = 1, pl.kc = 4;
    } else {
      pl.kernel = (t64 >= 256 || R <= 128) ? KERNEL_LDS : KERNEL_STREAM;
      pl.tn = 4, pl.kc = 2, pl.tm = (I >= 256 && t64 >= 100) ? 2 : 1;
    }
  }
  if (pl.kernel != KERNEL_STREAM) {
    if (!can_split || pl.nbs < 1) pl.nbs = 1;
    return pl;
  }
  finish(pl, I, J, R);
  while (pl.wpt > 1 && pl.wpt > pl.nchunks) pl.wpt >>= 1;
  if (!can_split || pl.wpt != kWaves) pl.nbs = 1;
  if (pl.nbs > 32) pl.nbs = 32;
  slice(pl);
  pl.combine = pl.nbs > 1;
  return pl;
}
Plan make_plan(int op, int64_t I, int64_t J, int64_t R, bool direct, bool bf16) {
  return op == ADNM_SKGEMM_TN ? plan_deferred(I, J, R, direct, bf16) : plan_critical(op, I, J, R, !direct && J % 4 == 0, bf16);
}

}  // namespace

namespace {
int shape_ok(int op, int64_t M, int64_t N, int64_t K) {
  if (M < 1 || N < 4 || K < 4 || M > (1 << 20) || N > 16384 || K > 16384 || M * N >= (1ll << 31) || M * K >= (1ll << 31)) return 0;   // (int element offsets)
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

template <bool A_RC, bool B_RC, int TM, int TN, int KC>
void launch(int prec, int bt, unsigned grid, hipStream_t st, const SkArgs& p) {
  if (bt == ADNM_B_BF16) skgemm_kernel<A_RC, B_RC, ADNM_MFMA_BF16, false, TM, TN, KC, ADNM_B_BF16><<<grid, kBlock, 0, st>>>(p);
  else if (bt == ADNM_B_FP8 && prec == ADNM_MFMA_FP8) skgemm_kernel<A_RC, B_RC, ADNM_MFMA_FP8, false, TM, TN, KC, ADNM_B_FP8><<<grid, kBlock, 0, st>>>(p);
  else if (bt == ADNM_B_FP8) skgemm_kernel<A_RC, B_RC, ADNM_MFMA_FP8, true, TM, TN, KC, ADNM_B_FP8><<<grid, kBlock, 0, st>>>(p);
  else if (prec == ADNM_MFMA_BF16) skgemm_kernel<A_RC, B_RC, ADNM_MFMA_BF16, false, TM, TN, KC, ADNM_B_F32><<<grid, kBlock, 0, st>>>(p);
  else if (prec == ADNM_MFMA_FP8) skgemm_kernel<A_RC, B_RC, ADNM_MFMA_FP8, false, TM, TN, KC, ADNM_B_F32><<<grid, kBlock, 0, st>>>(p);
  else if (prec == ADNM_MFMA_FP8_GRAD) skgemm_kernel<A_RC, B_RC, ADNM_MFMA_FP8, true, TM, TN, KC, ADNM_B_F32><<<grid, kBlock, 0, st>>>(p);
  else skgemm_kernel<A_RC, B_RC, ADNM_MFMA_F32, false, TM, TN, KC, ADNM_B_F32><<<grid, kBlock, 0, st>>>(p);
}
}  // namespace

extern "C" int adnm_skgemm_supported(int op, int64_t M, int64_t N, int64_t K) { return shape_ok(op, M, N, K); }

namespace {
int64_t ws_need(const Plan& pl, int64_t I, int64_t J, int64_t R) {
  if (pl.kernel == KERNEL_LDS) return adnm_lgemm_ws_bytes(I, J, R, pl.nbs);
  if (pl.kernel < 0) return 16;
  if (pl.nbs <= 1) return 16;
  // [arrival counters (in-launch combine only) | nbs slabs of I*J (+ I bias-gradient sums: op TN)]
  return (pl.combine ? adnm_ticket_bytes(pl.ntiles) : 0) + (int64_t)pl.nbs * (I * J + I) * (int64_t)sizeof(float);
}
}  // namespace

// enough for either precision (the tuned configuration of a shape may differ between them)
extern "C" int64_t adnm_skgemm_ws_bytes(int op, int64_t M, int64_t N, int64_t K) {
  if (!shape_ok(op, M, N, K)) return -1;
  int64_t I, J, R;
  dims(op, M, N, K, &I, &J, &R);
  const int64_t a = ws_need(make_plan(op, I, J, R, false, false), I, J, R), b = ws_need(make_plan(op, I, J, R, false, true), I, J, R);
  return a > b ? a : b;
}

// the counter part of a split NT / NN launch's workspace (what ws must hold when the slabs go to uncached space); 0: never split
extern "C" int64_t adnm_skgemm_counter_bytes(int op, int64_t M, int64_t N, int64_t K) {
  if (!shape_ok(op, M, N, K)) return -1;
  if (op == ADNM_SKGEMM_TN) return 0;
  int64_t I, J, R, need = 0;
  dims(op, M, N, K, &I, &J, &R);
  for (int bf = 0; bf < 2; ++bf) {
    const Plan pl = make_plan(op, I, J, R, false, bf != 0);
    if (pl.kernel < 0) continue;
    int64_t ntiles = pl.ntiles, nbs = pl.nbs;
    if (pl.kernel == KERNEL_LDS) {
      ntiles = adnm_cdiv(I, 64) * adnm_cdiv(J, 64);
      nbs = adnm_lgemm_ws_bytes(I, J, R, pl.nbs) > 16 ? 2 : 1;
    }
    if (nbs > 1 && adnm_ticket_bytes(ntiles) > need) need = adnm_ticket_bytes(ntiles);
  }
  return need;
}

extern "C" int adnm_skgemm(int op, const float* a, int64_t lda, const void* b, int64_t ldb, int b_dtype, const float* b_scale, const float* bias,
                           float* c, int64_t ldc, float* dbias, void* ws, int64_t ws_bytes, void* slabs_uc, int64_t slabs_uc_bytes, int64_t M, int64_t N, int64_t K, int prec, float* q,
                           adnm_stream_t stream) {
  ADNM_REQUIRE(prec >= ADNM_MFMA_F32 && prec <= ADNM_MFMA_FP8_GRAD, "skgemm: bad prec %d", prec);
  const bool fp8 = prec == ADNM_MFMA_FP8 || prec == ADNM_MFMA_FP8_GRAD;
  ADNM_REQUIRE(!fp8 || q, "skgemm: the fp8 modes need a quantisation record");
  // the weight-gradient op keeps bf16 operands in the fp8 configuration (its two operands are the forward's activation and the output
  // gradient; the record of the forward call site does not describe that pair)
  if (fp8 && op == ADNM_SKGEMM_TN) prec = ADNM_MFMA_BF16, q = nullptr;
  ADNM_REQUIRE(a && b && c, "skgemm: null pointer");
  ADNM_REQUIRE(b_dtype == ADNM_B_F32 || (op != ADNM_SKGEMM_TN && ((b_dtype == ADNM_B_BF16 && prec == ADNM_MFMA_BF16) || (b_dtype == ADNM_B_FP8 && fp8 && b_scale))),
               "skgemm: a narrow weight operand (b_dtype %d) needs op NT / NN and the matching prec (bf16 shadow: ADNM_MFMA_BF16; fp8 shadow: an fp8 mode "
               "and its scale), got op %d prec %d", b_dtype, op, prec);
  ADNM_REQUIRE(shape_ok(op, M, N, K), "skgemm: unsupported op/shape op=%d M=%lld N=%lld K=%lld", op, (long long)M, (long long)N, (long long)K);
  int64_t I, J, R;
  dims(op, M, N, K, &I, &J, &R);
  ADNM_REQUIRE(ldc >= J && ldc % 4 == 0, "skgemm: bad output row stride");
  ADNM_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= (op == ADNM_SKGEMM_NT ? K : N) && ldb >= K, "skgemm: bad operand row strides");
  ADNM_REQUIRE(!(bias && op != ADNM_SKGEMM_NT) && !(dbias && op != ADNM_SKGEMM_TN), "skgemm: bias only with NT, dbias only with TN");
  ADNM_REQUIRE(((uintptr_t)a | (uintptr_t)c) % 16 == 0 && (uintptr_t)b % (b_dtype == ADNM_B_F32 ? 16 : (b_dtype == ADNM_B_BF16 ? 8 : 4)) == 0,
               "skgemm: operands must be aligned to 4 elements (16 bytes of fp32)");
  hipStream_t st = (hipStream_t)stream;
  const bool bf = prec != ADNM_MFMA_F32;   // the narrow modes share the tile plan measured for bf16
  Plan pl = make_plan(op, I, J, R, false, bf);
  // the deferred fold writes whole contiguous rows: a strided output (column slice of a wider buffer) takes no partials there
  if (op == ADNM_SKGEMM_TN && ldc != J) pl = make_plan(op, I, J, R, true, bf);
  ADNM_REQUIRE(pl.kernel >= 0, "skgemm: ADNM_SK_FORCE names a configuration this op has no kernel for");
  const char* scope = op == ADNM_SKGEMM_NT ? "skgemm_nt" : (op == ADNM_SKGEMM_NN ? "skgemm_nn" : "skgemm_tn");
  // algorithmic bytes: activations in and out in fp32, the weight in the storage it is read from (fp32 / bf16 shadow / fp8 shadow)
  const double algo_bytes = 4.0 * (double)M * (K + N) + (b_dtype == ADNM_B_BF16 ? 2.0 : (b_dtype == ADNM_B_FP8 ? 1.0 : 4.0)) * (double)N * K;
  if (pl.kernel == KERNEL_LDS) {
    ADNM_PROF(scope, st, algo_bytes);
    const int rc = adnm_lgemm_launch(op == ADNM_SKGEMM_NN, a, lda, b, ldb, b_dtype, b_scale, bias, c, ldc, ws, ws_bytes, slabs_uc, slabs_uc_bytes, I, J, R, pl.nbs, prec, q, st);
    if (rc != ADNM_OK) return rc;
    ADNM_CHECK_LAUNCH("skgemm");
    return ADNM_OK;
  }
  const bool split = pl.nbs > 1;
  // split NT / NN: ws = [arrival counters | slabs], or just the counters when the caller also hands over uncached slab space
  const int64_t slab_bytes = split ? (int64_t)pl.nbs * (I * J + (dbias ? I : 0)) * (int64_t)sizeof(float) : 0;
  const bool uc = split && pl.combine && slabs_uc && slabs_uc_bytes >= slab_bytes;
  const int64_t need = !split ? 0 : (uc ? adnm_ticket_bytes(pl.ntiles) : ws_need(pl, I, J, R));
  if (split && (!ws || ws_bytes < need)) {
    adnm_set_error("skgemm: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    return ADNM_EWORKSPACE;
  }
  SkArgs p;
  p.A = a, p.B = b, p.bias = bias, p.b_scale = b_scale;
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
  p.tickets = nullptr, p.out = c, p.ldo = ldc, p.uc = 0;
  p.q = reinterpret_cast<AdnmQuant*>(q);
  if (pl.combine) {
    // arrival counters (zero when idle) at the head of ws, in ordinary memory (agent-scope atomics); the slabs behind them (fenced
    // protocol) or in the caller's uncached space (a completed store is at the coherence point: no fences around the ticket)
    ADNM_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)slabs_uc & 255) == 0, "skgemm: a split launch needs 256-byte aligned workspaces");
    p.tickets = (int*)ws;
    p.C = uc ? (float*)slabs_uc : (float*)((char*)ws + adnm_ticket_bytes(pl.ntiles));
    p.uc = uc ? 1 : 0;
  }
  const unsigned grid = (unsigned)adnm_cdiv((int64_t)pl.ntiles * pl.nbs, kWaves / pl.wpt);
  bool queued = false;
  if (op == ADNM_SKGEMM_TN && (prec == ADNM_MFMA_F32 || prec == ADNM_MFMA_BF16)) {   // a leaf: may wait for the grouped launch
    static_assert(sizeof(SkArgs) <= sizeof(AdnmLeaf::args), "SkArgs must fit a leaf record");
    AdnmLeaf leaf;
    leaf.kind = ADNM_LEAF_SKGEMM_TN, leaf.grid = (int)grid, leaf.prec = prec, leaf.prof = scope, leaf.bytes = algo_bytes;
    memcpy(leaf.args, &p, sizeof(SkArgs));
    queued = adnm_leafq_push(leaf);
  }
  if (!queued) {
    ADNM_PROF(scope, st, algo_bytes);
    if (op == ADNM_SKGEMM_NT) {
      if (pl.tm == 1) launch<true, true, 1, 1, 4>(prec, b_dtype, grid, st, p);
      else if (pl.tm == 2) launch<true, true, 2, 2, 2>(prec, b_dtype, grid, st, p);
      else launch<true, true, 4, 4, 1>(prec, b_dtype, grid, st, p);
    } else if (op == ADNM_SKGEMM_NN) {
      if (pl.tm == 1) launch<true, false, 1, 4, 2>(prec, b_dtype, grid, st, p);
      else if (pl.tm == 2) launch<true, false, 2, 4, 2>(prec, b_dtype, grid, st, p);
      else launch<true, false, 4, 4, 1>(prec, b_dtype, grid, st, p);
    } else {
      launch<false, false, 4, 4, 1>(prec, b_dtype, grid, st, p);
    }
  }
  ADNM_CHECK_LAUNCH("skgemm");
  if (split && !pl.combine) {
    adnm_launch_fold("skgemm_fold", part, pl.nbs, (int)rowlen, {c, (int)(I * J)}, {dbias, dbias ? (int)I : 0}, {nullptr, 0}, {nullptr, 0}, st);
    ADNM_CHECK_LAUNCH("skgemm_fold");
  }
  return ADNM_OK;
}


// the queued weight-gradient problems, kMaxSk per launch (fp32 and bf16 problems in separate launches)
int adnm_skgemm_tn_launch_multi(const AdnmLeaf* const* items, int n, hipStream_t st) {
  for (int prec = ADNM_MFMA_F32; prec <= ADNM_MFMA_BF16; ++prec) {
    int i = 0;
    while (i < n) {
      MultiSk m;
      m.count = 0;
      int blocks = 0;
      double bytes = 0;
      for (; i < n && m.count < kMaxSk; ++i) {
        if (items[i]->prec != prec) continue;
        memcpy(&m.p[m.count], items[i]->args, sizeof(SkArgs));
        blocks += items[i]->grid;
        m.blk_end[m.count++] = blocks;
        bytes += items[i]->bytes;
      }
      if (!m.count) break;
      for (int k = m.count; k < kMaxSk; ++k) m.blk_end[k] = blocks;
      ADNM_PROF("skgemm_tn", st, bytes);
      if (prec == ADNM_MFMA_BF16) skgemm_tn_multi_kernel<ADNM_MFMA_BF16><<<(unsigned)blocks, kBlock, 0, st>>>(m);
      else skgemm_tn_multi_kernel<ADNM_MFMA_F32><<<(unsigned)blocks, kBlock, 0, st>>>(m);
    }
  }
  ADNM_CHECK_LAUNCH("skgemm_tn (grouped)");
  return ADNM_OK;
}
