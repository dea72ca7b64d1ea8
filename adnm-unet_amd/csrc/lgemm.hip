// K6b (forward / input-gradient half) — the Linear layers of the deep stages as LDS-tiled fp32-storage GEMMs on the matrix cores:
//
//   op NT:  C[M,N] = A[M,K] . W[N,K]^T (+ bias)      forward            both operands contiguous along the reduction ("RC")
//   op NN:  C[M,K] = A[M,N] . W[N,K]                 input gradient     W contiguous along the OUTPUT axis ("OC")
//
// Reference call sites: Mamba2.in_proj / out_proj (ADNssd.py:309,461), FeedForward.project_in / project_out
// (model_untils.py:193,196), Mlp.fc1/fc2 (:64,67), ConvFFD (:217,221), Block.out_proj (ADNMUNet.py:163),
// StandardAttention.to_qkv / to_out (ADNssd.py:33-34), Channel_Att_Bridge.att* (:744-750), UpSample's ConvTranspose2d as a GEMM.
//
// Shapes: M = 4 .. 1024 token rows, 128 .. 4672 features.  At M = 64 the weight is streamed once from HBM (32 FLOP per byte);
// at M >= 256 the operands sit in L2 and the L2 -> CU fill rate (64 B/clk/CU) is the ceiling.  Both want the same thing: every
// CU busy, whole 128-byte lines per row, several tiles in flight per CU.
//
//   * workgroup = 4 waves (2 x 2), output tile 64 x 64, reduction step 32: each operand tile is 64 rows x 128 B, fetched as one
//     16-byte load per lane (8 lanes cover a row's line), held in a register ring D tiles deep (D x 16 KB in flight per workgroup,
//     3-4 workgroups per CU), then written to one of two LDS buffers — rounded to bf16 on the way in the ADNM_MFMA_BF16 mode;
//   * LDS images need no transposes: an RC tile is [row][k] with a padded row (24 / 40 words: conflict-free ds_read_b128 of a lane's
//     8 bf16 / 4 fp32 reduction steps), an OC tile keeps its memory order [k][column] (bf16: two k per word) with 68-word rows
//     (conflict-free ds_read_b32).  Which reduction steps a lane feeds to which MFMA is a permutation shared by both operands;
//   * when the caller asks for it the reduction is split over workgroups and the slabs of a tile are combined INSIDE the launch: the
//     workgroup that draws the tile's last ticket adds all slabs in slice order (bitwise reproducible, no atomics on data, no
//     second launch on the critical path).
#include "adnm_common.h"

namespace {

using f32x4 = adnm_f32x4;
constexpr int kThreads = 256, kTile = 64, kStep = 32;

// LDS images by matrix-core precision: bf16 pairs (rounded once, when a tile is staged); fp8 BYTES (scaled / saturated / converted once, when
// a tile is staged — or copied as they are from the weight's e4m3 shadow), both operands row-major [row][k] with 40-byte rows, the OC
// operand transposed by byte stores on the way in: a fragment is one ds_read_b64 and the tile a quarter of the fp32 image's LDS bytes;
// fp32 keeps fp32 images.
template <int PREC>
struct Geo {
  static constexpr bool BF16 = PREC == ADNM_MFMA_BF16, FP8 = PREC == ADNM_MFMA_FP8;
  static constexpr int rc_stride = FP8 ? 10 : (BF16 ? 24 : 40);   // words per RC row (32 reduction steps + pad)
  static constexpr int rc_words = kTile * rc_stride;
  static constexpr int oc_stride = 68;               // words per OC row (64 columns + pad)
  static constexpr int oc_rows = BF16 ? kStep / 2 : kStep;
  static constexpr int oc_words = FP8 ? rc_words : oc_rows * oc_stride;   // (fp8: the OC operand is staged transposed, as an RC image)
};

struct LgArgs {
  const float* A;
  int64_t lda;          // A(i, r) = A[i*lda + r]
  const void* B;        // fp32, or the weight's narrow shadow (BT: bf16 / scaled e4m3)
  int64_t ldb;          // RC: B(r, j) = B[j*ldb + r];  OC: B(r, j) = B[r*ldb + j]   (element stride)
  const float* b_scale; // fp8 shadow: the weight's scale (value = e4m3 / *b_scale); else NULL
  const float* bias;
  float* C;
  int64_t ldc;
  float* slab;          // nbs > 1: [tile][slice][64][64]
  int* tickets;
  int uc;               // the slabs live in uncached memory (the caller said so: ws_uncached): no agent-scope fences around the ticket
  int I, J, R;
  int tiles_j, nbs, kt_per_slice, nkt;
  AdnmQuant* q;         // quantisation record (fp8 scales, amax collection) or NULL
};

// by value on purpose: `ok ? a : b` on two float4 OBJECTS selects an address and pushes both into scratch
__device__ __forceinline__ float4 keep_if(bool ok, float4 v) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  using bf2 = __attribute__((ext_vector_type(2))) __bf16;
  using f2 = __attribute__((ext_vector_type(2))) float;
  const f2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
}

template <bool B_OC, int PREC, bool A_BF8, int D, int BT>
__global__ __launch_bounds__(kThreads) void lgemm_kernel(LgArgs p) {
  static_assert(BT == ADNM_B_F32 || (BT == ADNM_B_BF16 && PREC == ADNM_MFMA_BF16) || (BT == ADNM_B_FP8 && PREC == ADNM_MFMA_FP8),
                "a narrow weight shadow feeds the matrix-core precision it was made for");
  constexpr bool BF16 = PREC == ADNM_MFMA_BF16, FP8 = PREC == ADNM_MFMA_FP8;
  using G = Geo<PREC>;
  constexpr int kAW = G::rc_words, kBW = B_OC ? G::oc_words : G::rc_words;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * (kAW + kBW)];
  uint32_t* const ldsA = lds;
  uint32_t* const ldsB = lds + 2 * kAW;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wi = wave >> 1, wj = wave & 1;

  // workgroup -> (tile, slice).  Workgroups go round-robin over the 8 XCDs: give each XCD a contiguous run of (tile, slice) pairs, so the
  // slices of a tile (whose slabs the last of them reads back) and neighbouring tiles (which share A rows) meet in one L2.  A speed
  // choice only: the combine is correct for any placement.
  int lin = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) lin = (lin & 7) * (nblk >> 3) + (lin >> 3);
  const int tile = lin / p.nbs, slice = lin - tile * p.nbs;
  const int i0 = (tile / p.tiles_j) * kTile, j0 = (tile % p.tiles_j) * kTile;
  const int kt0 = slice * p.kt_per_slice;
  const int nkt = p.nkt - kt0 < p.kt_per_slice ? p.nkt - kt0 : p.kt_per_slice;

  // ---- loaders.  RC: lane -> (row tid>>3 (+32), reduction quad tid&7).  OC: lane -> (column quad tid&15, reduction row tid>>4 ...).
  // Rows / columns outside the matrix read a clamped (valid) address and land in accumulator rows / columns that are never stored;
  // reduction steps outside [0, R) and tiles past the slice read the operand's first 16 bytes (one broadcast request) and are zeroed
  // when they are written to LDS, so nothing in the loop waits for a load before it has to.
  const int q = tid & 7, lrow = tid >> 3;
  const int cq = tid & 15, lr = tid >> 4;
  const float* a_row[2];
  int b_off[2];   // element offset of this lane's B row / column quad (the operand may be a 2- or 1-byte shadow: offsets, not pointers)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ia = i0 + lrow + 32 * u;
    a_row[u] = p.A + (int64_t)(ia < p.I ? ia : p.I - 1) * p.lda + 4 * q;
    if (!B_OC) {
      const int jb = j0 + lrow + 32 * u;
      b_off[u] = (int)((int64_t)(jb < p.J ? jb : p.J - 1) * p.ldb + 4 * q);
    } else {
      const int jb = j0 + 4 * cq;
      b_off[u] = (int)((jb + 4 <= p.J ? jb : p.J - 4) + (int64_t)(BF16 ? 2 * lr + u : lr + 16 * u) * p.ldb);
    }
  }
  float4 ra[D][2], rb[D][2];
  auto a_ok = [&](int kt) { return kt < nkt && (kt0 + kt) * kStep + 4 * q < p.R; };
  auto b_ok = [&](int kt, int u) {
    if (!B_OC) return kt < nkt && (kt0 + kt) * kStep + 4 * q < p.R;
    return kt < nkt && (kt0 + kt) * kStep + (BF16 ? 2 * lr + u : lr + 16 * u) < p.R;
  };
  // element offsets from a lane's own row pointer back to the operand's first word (the "dead" address): selecting a 32-bit offset
  // is one v_cndmask, selecting between two pointers makes the compiler branch around the address arithmetic
  int a_dead[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) a_dead[u] = (int)(p.A - a_row[u]);
  auto load = [&](int kt, float4 (&av)[2], float4 (&bv)[2]) {
    const int r0 = (kt0 + kt) * kStep;
    const bool oka = a_ok(kt);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      av[u] = *reinterpret_cast<const float4*>(a_row[u] + (oka ? r0 : a_dead[u]));
      const int64_t bo = b_ok(kt, u) ? b_off[u] + (B_OC ? r0 * (int)p.ldb : r0) : 0;
      if constexpr (BT == ADNM_B_FP8)   // the shadow's bytes go into the fp8 image as they are: carried as the bit pattern of .x
        bv[u] = make_float4(__int_as_float(*reinterpret_cast<const int*>(reinterpret_cast<const uint8_t*>(p.B) + bo)), 0.f, 0.f, 0.f);
      else bv[u] = adnm_ldb4<BT>(p.B, bo);
    }
  };
  // fp8: per-tensor scales from the call site's quantisation record (the accumulators are un-scaled in the epilogue); rec_a / rec_b
  // (workgroup-uniform): collect max |value| of the A rows / B columns this workgroup stages — the first tile column / tile row only
  // (an fp8 SHADOW is already scaled and rounded: it passes through the fragment builder with scale 1; q_sbo divides the accumulators)
  float q_sa = 1.f, q_sb = 1.f, q_sbo = 1.f, amax_a = 0.f, amax_b = 0.f;
  bool rec_a = false, rec_b = false;
  if (p.q) {
    if (PREC == ADNM_MFMA_FP8) q_sa = p.q->scale_a, q_sb = q_sbo = p.q->scale_b;
    const bool rec = p.q->record != 0.f;
    rec_a = rec && (tile % p.tiles_j) == 0, rec_b = BT == ADNM_B_F32 && rec && (tile / p.tiles_j) == 0;
  }
  if (BT == ADNM_B_FP8) q_sb = 1.f, q_sbo = *p.b_scale;
  auto stage = [&](int kt, int buf, const float4 (&av)[2], const float4 (&bv)[2]) {
    uint32_t* const sa = ldsA + buf * kAW;
    uint32_t* const sb = ldsB + buf * kBW;
    const bool oka = a_ok(kt);
    if (rec_a) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4 v = keep_if(oka, av[u]);
        amax_a = adnm_amax4(amax_a, v.x, v.y, v.z, v.w);
      }
    }
    if (rec_b) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4 v = keep_if(b_ok(kt, u), bv[u]);
        amax_b = adnm_amax4(amax_b, v.x, v.y, v.z, v.w);
      }
    }
    if constexpr (FP8) {
      // bytes: 4 reduction steps of a row = one word.  A: scaled, saturated, e4m3 (e5m2 when it is a gradient); B: the same from fp32
      // values, or the shadow's own bytes
      auto q4 = [](float4 v, float sc, bool bf8) {
        const float c0 = __builtin_amdgcn_fmed3f(v.x * sc, bf8 ? 57344.f : 448.f, bf8 ? -57344.f : -448.f);
        const float c1 = __builtin_amdgcn_fmed3f(v.y * sc, bf8 ? 57344.f : 448.f, bf8 ? -57344.f : -448.f);
        const float c2 = __builtin_amdgcn_fmed3f(v.z * sc, bf8 ? 57344.f : 448.f, bf8 ? -57344.f : -448.f);
        const float c3 = __builtin_amdgcn_fmed3f(v.w * sc, bf8 ? 57344.f : 448.f, bf8 ? -57344.f : -448.f);
        int w = 0;
        if (bf8) w = __builtin_amdgcn_cvt_pk_bf8_f32(c0, c1, w, false), w = __builtin_amdgcn_cvt_pk_bf8_f32(c2, c3, w, true);
        else w = __builtin_amdgcn_cvt_pk_fp8_f32(c0, c1, w, false), w = __builtin_amdgcn_cvt_pk_fp8_f32(c2, c3, w, true);
        return (uint32_t)w;
      };
#pragma unroll
      for (int u = 0; u < 2; ++u) sa[(lrow + 32 * u) * G::rc_stride + q] = q4(keep_if(oka, av[u]), q_sa, A_BF8);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4 v = keep_if(b_ok(kt, u), bv[u]);
        const uint32_t w = BT == ADNM_B_FP8 ? (uint32_t)__float_as_int(v.x) : q4(v, q_sb, false);
        if (!B_OC) {
          sb[(lrow + 32 * u) * G::rc_stride + q] = w;
        } else {   // 4 columns of reduction row lr + 16 u -> byte (column, k) of the transposed image
          uint8_t* const sb8 = reinterpret_cast<uint8_t*>(sb);
          const int k = lr + 16 * u;
#pragma unroll
          for (int c = 0; c < 4; ++c) sb8[(4 * cq + c) * (4 * G::rc_stride) + k] = (uint8_t)(w >> (8 * c));
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 v = keep_if(oka, av[u]);
      uint32_t* dst = sa + (lrow + 32 * u) * G::rc_stride;
      if (BF16) *reinterpret_cast<uint2*>(dst + 2 * q) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
      else *reinterpret_cast<float4*>(dst + 4 * q) = v;
    }
    if (!B_OC) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4 v = keep_if(b_ok(kt, u), bv[u]);
        uint32_t* dst = sb + (lrow + 32 * u) * G::rc_stride;
        if (BF16) *reinterpret_cast<uint2*>(dst + 2 * q) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
        else *reinterpret_cast<float4*>(dst + 4 * q) = v;
      }
    } else if (BF16) {   // rows 2*lr and 2*lr+1 share a word per column
      const float4 v0 = keep_if(b_ok(kt, 0), bv[0]), v1 = keep_if(b_ok(kt, 1), bv[1]);
      *reinterpret_cast<uint4*>(sb + lr * G::oc_stride + 4 * cq) = make_uint4(pack2(v0.x, v1.x), pack2(v0.y, v1.y), pack2(v0.z, v1.z), pack2(v0.w, v1.w));
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) *reinterpret_cast<float4*>(sb + (lr + 16 * u) * G::oc_stride + 4 * cq) = keep_if(b_ok(kt, u), bv[u]);
    }
  };

  // accumulators: block (a, b) of the wave's 32 x 32 quadrant, held TRANSPOSED (operands swapped) so that a lane owns four consecutive
  // output columns of one row: C[i0 + 32 wi + 16 a + l15][j0 + 32 wj + 16 b + 4 kq + 0..3]
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const uint32_t* const sa = ldsA + buf * kAW;
    const uint32_t* const sb = ldsB + buf * kBW;
    if constexpr (FP8) {
      AdnmFrag<ADNM_MFMA_FP8> fa[2], fb[2];   // 8 bytes = reduction steps 8 kq .. 8 kq + 7 of the tile: one ds_read_b64, one fp8 MFMA per block pair
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a].v = *reinterpret_cast<const long*>(sa + (32 * wi + 16 * a + l15) * G::rc_stride + 2 * kq);
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b].v = *reinterpret_cast<const long*>(sb + (32 * wj + 16 * b + l15) * G::rc_stride + 2 * kq);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = adnm_mma<ADNM_MFMA_FP8, false, A_BF8>(fb[b], fa[a], acc[a][b]);
    } else if constexpr (BF16) {
      uint4 fa[2], fb[2];   // 8 bf16 = reduction steps 8 kq .. 8 kq + 7 of the tile: ONE v_mfma_f32_16x16x32_bf16 per block pair
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = *reinterpret_cast<const uint4*>(sa + (32 * wi + 16 * a + l15) * G::rc_stride + 4 * kq);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (!B_OC) fb[b] = *reinterpret_cast<const uint4*>(sb + (32 * wj + 16 * b + l15) * G::rc_stride + 4 * kq);
        else {
          const uint32_t* s = sb + (4 * kq) * G::oc_stride + 32 * wj + 16 * b + l15;
          fb[b] = make_uint4(s[0], s[G::oc_stride], s[2 * G::oc_stride], s[3 * G::oc_stride]);
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(adnm_bf16x8, fb[b]), __builtin_bit_cast(adnm_bf16x8, fa[a]), acc[a][b], 0, 0, 0);
    } else {
      // fp32 images: lane (l15, kq) takes reduction steps 16 g + 4 kq + e of the tile's 32 (g = 0, 1): 8 per lane = one 32-step MFMA group
      float fa[2][2][4], fb[2][2][4];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const float4 v = *reinterpret_cast<const float4*>(sa + (32 * wi + 16 * a + l15) * G::rc_stride + 16 * g + 4 * kq);
          fa[a][g][0] = v.x, fa[a][g][1] = v.y, fa[a][g][2] = v.z, fa[a][g][3] = v.w;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (!B_OC) {
            const float4 v = *reinterpret_cast<const float4*>(sb + (32 * wj + 16 * b + l15) * G::rc_stride + 16 * g + 4 * kq);
            fb[b][g][0] = v.x, fb[b][g][1] = v.y, fb[b][g][2] = v.z, fb[b][g][3] = v.w;
          } else {
            const float* s = reinterpret_cast<const float*>(sb) + (16 * g + 4 * kq) * G::oc_stride + 32 * wj + 16 * b + l15;
#pragma unroll
            for (int e = 0; e < 4; ++e) fb[b][g][e] = s[e * G::oc_stride];
          }
        }
      }
      AdnmFrag<PREC> pa[2], pb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) pa[a] = adnm_make_frag<PREC, A_BF8>(fa[a][0], fa[a][1], q_sa);
#pragma unroll
      for (int b = 0; b < 2; ++b) pb[b] = adnm_make_frag<PREC, false>(fb[b][0], fb[b][1], q_sb);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = adnm_mma<PREC, false, A_BF8>(pb[b], pa[a], acc[a][b]);
    }
  };

  // ---- main loop: register ring D tiles deep, two LDS buffers, one barrier per tile.  A wave that writes tile t+2 into the buffer tile t
  // was read from has passed the barrier of tile t+1, which every wave reaches only after its reads of tile t.
#pragma unroll
  for (int s = 0; s < D; ++s) {
    load(s, ra[s], rb[s]);
    __builtin_amdgcn_sched_barrier(0);   // keep the ring's load ORDER: the loop's wait counts are derived from it
  }
  for (int t = 0; t < nkt; t += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {   // no branch on t + s < nkt: a tile past the slice is all zeros (and costs ~0.1 us), a branch here would
      stage(t + s, s & 1, ra[s], rb[s]);   // make the compiler's wait counts drain the ring
      load(t + s + D, ra[s], rb[s]);
      __syncthreads();
      compute(s & 1);
      __builtin_amdgcn_sched_barrier(0);   // or the scheduler hoists the NEXT slot's zero-selects (= a wait for its loads) up here
    }
  }

  if (rec_a) adnm_amax_commit(&p.q->amax_a, amax_a);
  if (rec_b) adnm_amax_commit(&p.q->amax_b, amax_b);
  if (PREC == ADNM_MFMA_FP8) {   // back to the operands' own scale, before bias / slabs
    const float inv = 1.0f / (q_sa * q_sbo);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = acc[a][b] * inv;
  }
  // ---- epilogue
  const bool split = p.nbs > 1;
  float* const slab = split ? p.slab + ((int64_t)tile * p.nbs + slice) * (kTile * kTile) : nullptr;
  const bool vec_ok = (p.ldc & 3) == 0 && (p.J & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
                      (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int il = 32 * wi + 16 * a + l15;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int jl = 32 * wj + 16 * b + 4 * kq;
      const f32x4 v = acc[a][b];
      if (split) {
        *reinterpret_cast<float4*>(slab + il * kTile + jl) = make_float4(v[0], v[1], v[2], v[3]);
      } else if (i0 + il < p.I) {
        const int jg = j0 + jl;
        float* dst = p.C + (int64_t)(i0 + il) * p.ldc + jg;
        if (vec_ok && jg + 3 < p.J) {
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + jg);
          *reinterpret_cast<float4*>(dst) = make_float4(v[0] + bv.x, v[1] + bv.y, v[2] + bv.z, v[3] + bv.w);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (jg + r < p.J) dst[r] = v[r] + (p.bias ? p.bias[jg + r] : 0.f);
        }
      }
    }
  }
  if (!split) return;

  // In-launch combine of the nbs slabs of this tile.  Publish: every wave drains its slab stores, the workgroup meets, one lane
  // (releases at agent scope unless the slabs are in uncached memory — ws_uncached — and) draws a ticket; the workgroup that
  // draws the last one (acquires at agent scope and) adds the slabs in
  // slice order 0 .. nbs-1 (its own included, re-read from memory), so the sum does not depend on which slice arrived last.  The
  // counter goes back to zero for the next launch on the stream.  Correct wherever the slices ran (any CU / XCD).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* const flag = reinterpret_cast<int*>(lds);
  if (tid == 0) {
    if (!p.uc) {   // slabs in ordinary (L2-cached) memory: write this XCD's L2 back before the ticket, invalidate before reading the others'
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
  __syncthreads();
  if (*flag != p.nbs - 1) return;
  {
    const int il = tid >> 2, jl = (tid & 3) * 16;   // 64 rows x 4 lanes x 4 float4
    const int ig = i0 + il, jg = j0 + jl;
    if (ig >= p.I || jg >= p.J) return;
    const float* src = p.slab + (int64_t)tile * p.nbs * (kTile * kTile) + il * kTile + jl;
    float4 sum[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) sum[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
    for (int s = 0; s < p.nbs; ++s) {
      float4 v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const float4*>(src + (int64_t)s * (kTile * kTile) + 4 * c);
#pragma unroll
      for (int c = 0; c < 4; ++c) sum[c].x += v[c].x, sum[c].y += v[c].y, sum[c].z += v[c].z, sum[c].w += v[c].w;
    }
    float* dst = p.C + (int64_t)ig * p.ldc + jg;   // split => J % 4 == 0, ldc % 4 == 0, C 16-byte aligned (host checks)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (jg + 4 * c >= p.J) break;
      float4 o = sum[c];
      if (p.bias) {
        const float* bp = p.bias + jg + 4 * c;
        o.x += bp[0], o.y += bp[1], o.z += bp[2], o.w += bp[3];
      }
      *reinterpret_cast<float4*>(dst + 4 * c) = o;
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
constexpr int kRingDepth = 4;

struct LgPlan {
  int tiles_i, tiles_j, ntiles, nkt, nbs, kt_per_slice;
};
// nbs: the cross-workgroup split the caller (skgemm.hip's planner) asks for; clamped so that every slice has at least one step tile
LgPlan make_plan(int64_t I, int64_t J, int64_t R, int nbs) {
  LgPlan pl;
  pl.tiles_i = (int)adnm_cdiv(I, kTile);
  pl.tiles_j = (int)adnm_cdiv(J, kTile);
  pl.ntiles = pl.tiles_i * pl.tiles_j;
  pl.nkt = (int)adnm_cdiv(R, kStep);
  if (nbs > pl.nkt) nbs = pl.nkt;
  if (nbs > 32) nbs = 32;
  if (nbs < 1 || J % 4 != 0) nbs = 1;
  pl.kt_per_slice = (int)adnm_cdiv(pl.nkt, nbs);
  pl.nbs = (int)adnm_cdiv(pl.nkt, pl.kt_per_slice);
  return pl;
}
}  // namespace

// Shared with skgemm.hip (adnm_skgemm routes op NT / NN here).  b_oc: the second operand is contiguous along the OUTPUT axis (op NN).
int64_t adnm_lgemm_ws_bytes(int64_t I, int64_t J, int64_t R, int nbs) {
  const LgPlan pl = make_plan(I, J, R, nbs);
  return pl.nbs > 1 ? adnm_ticket_bytes(pl.ntiles) + (int64_t)pl.ntiles * pl.nbs * kTile * kTile * (int64_t)sizeof(float) : 16;
}

int adnm_lgemm_launch(bool b_oc, const float* a, int64_t lda, const void* b, int64_t ldb, int b_dtype, const float* b_scale, const float* bias, float* c,
                      int64_t ldc, void* ws, int64_t ws_bytes, void* slabs_uc, int64_t slabs_uc_bytes, int64_t I, int64_t J, int64_t R, int nbs, int prec,
                      float* q, hipStream_t st) {
  const LgPlan pl = make_plan(I, J, R, nbs);
  LgArgs p;
  p.A = a, p.lda = lda, p.B = b, p.ldb = ldb, p.b_scale = b_scale, p.bias = bias, p.C = c, p.ldc = ldc;
  ADNM_REQUIRE(J * ldb < (1ll << 31) && R * ldb < (1ll << 31), "skgemm: weight beyond 2^31 elements (32-bit lane offsets)");
  p.I = (int)I, p.J = (int)J, p.R = (int)R;
  p.tiles_j = pl.tiles_j, p.nbs = pl.nbs, p.kt_per_slice = pl.kt_per_slice, p.nkt = pl.nkt;
  p.slab = nullptr, p.tickets = nullptr, p.uc = 0;
  p.q = reinterpret_cast<AdnmQuant*>(q);
  if (pl.nbs > 1) {
    // arrival counters (zero when idle) at the head of ws, in ordinary memory; the slabs behind them (fenced protocol) or in the caller's
    // uncached space (no fences around the ticket)
    const int64_t slab_bytes = (int64_t)pl.ntiles * pl.nbs * kTile * kTile * (int64_t)sizeof(float);
    const bool uc = slabs_uc && slabs_uc_bytes >= slab_bytes;
    const int64_t need = uc ? adnm_ticket_bytes(pl.ntiles) : adnm_lgemm_ws_bytes(I, J, R, nbs);
    if (!ws || ws_bytes < need) {
      adnm_set_error("skgemm: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
      return ADNM_EWORKSPACE;
    }
    ADNM_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)slabs_uc & 255) == 0, "skgemm: a split launch needs 256-byte aligned workspaces");
    p.tickets = (int*)ws;
    p.slab = uc ? (float*)slabs_uc : (float*)((char*)ws + adnm_ticket_bytes(pl.ntiles));
    p.uc = uc ? 1 : 0;
  }
  const unsigned grid = (unsigned)(pl.ntiles * pl.nbs);
#define LG(OC)                                                                                                                             \
  do {                                                                                                                                     \
    if (b_dtype == ADNM_B_BF16) lgemm_kernel<OC, ADNM_MFMA_BF16, false, kRingDepth, ADNM_B_BF16><<<grid, kThreads, 0, st>>>(p);             \
    else if (b_dtype == ADNM_B_FP8 && prec == ADNM_MFMA_FP8) lgemm_kernel<OC, ADNM_MFMA_FP8, false, kRingDepth, ADNM_B_FP8><<<grid, kThreads, 0, st>>>(p); \
    else if (b_dtype == ADNM_B_FP8) lgemm_kernel<OC, ADNM_MFMA_FP8, true, kRingDepth, ADNM_B_FP8><<<grid, kThreads, 0, st>>>(p);            \
    else if (prec == ADNM_MFMA_BF16) lgemm_kernel<OC, ADNM_MFMA_BF16, false, kRingDepth, ADNM_B_F32><<<grid, kThreads, 0, st>>>(p);         \
    else if (prec == ADNM_MFMA_FP8) lgemm_kernel<OC, ADNM_MFMA_FP8, false, kRingDepth, ADNM_B_F32><<<grid, kThreads, 0, st>>>(p);           \
    else if (prec == ADNM_MFMA_FP8_GRAD) lgemm_kernel<OC, ADNM_MFMA_FP8, true, kRingDepth, ADNM_B_F32><<<grid, kThreads, 0, st>>>(p);       \
    else lgemm_kernel<OC, ADNM_MFMA_F32, false, kRingDepth, ADNM_B_F32><<<grid, kThreads, 0, st>>>(p);                                      \
  } while (0)
  if (!b_oc) LG(false);
  else LG(true);
#undef LG
  return ADNM_OK;
}
