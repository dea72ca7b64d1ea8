// K5 — the dense 3x3 'same' convolutions of the U-Net conv stack as NHWC implicit GEMMs on the matrix cores.
// Reference call sites: PatchEmbed.conv2 5->32 (model_untils.py:259-273), WTLayer.conv 32->64 / 64->128 / 256->64 / 128->32 /
// 64->32 (:376-387), OutProj.conv[0] 32->64 and conv2 20->20 (:818-849) — nn.Conv2d(k=3, s=1, p=1) [+ bias] [+ GELU].
//
//   forward   out[p, n]   = act( sum_{tap, k} in[p + tap, k] * W[n, tap, k] + bias[n] )       p = pixel, n = Cout, k = Cin
//   dgrad     din[p, k]   = sum_{tap, n} dpre[p - tap, n] * W[n, tap, k]                      dpre = dout * act'(pre)
//   wgrad     dW[n,tap,k] = sum_p dpre[p, n] * in[p + tap, k],   dbias[n] = sum_p dpre[p, n]
//
// One gather-GEMM kernel serves forward and dgrad (dgrad = the same conv over dpre with the weight read through swapped
// strides and flipped taps); wgrad is a second kernel.  v_mfma_f32_16x16x4_f32 (exact fp32).  Layout choices:
//   * the batch is one TALL image: B images of H rows with ONE zero row between them, so "same" zero padding in y, image
//     boundaries and small maps (4x4 ... 16x16: a 16-pixel MFMA block = 16/W rows x W columns) are all the same tile code;
//   * a workgroup (4 waves) owns a tile of 8 pixel blocks and up to 64 output channels; the input tile + halo of a 16-channel
//     chunk is staged ONCE in LDS (coalesced 64-byte channel runs) and read 9 times (taps) as MFMA B operands (one 16-byte
//     LDS read = 4 reduction steps: the k index is permuted identically on the weight side); weights come straight from
//     global memory / L2 as A operands (16-byte loads when the reduction axis is contiguous), so the accumulator block is
//     out^T and a lane stores 4 consecutive channels of one pixel;
//   * deep maps have few pixels and long reductions: the channel chunks are split over blockIdx.z into fp32 partials and a
//     small epilogue kernel sums them (fixed order) and applies bias / activation;
//   * wgrad: each WAVE owns a quarter of the tile's pixels and all 9 taps x up to 64 output channels x 16 input channels as
//     register accumulators; per-wave partial rows + the shared deterministic fold (batchable: a parameter gradient).
#include "adnm_common.h"

namespace {

using f32x4 = adnm_f32x4;
constexpr int kBlock = 256, kWaves = 4;
constexpr int CK = 16, CKP = CK + 4;   // channels per staged chunk, LDS pitch of a pixel
constexpr int MB = 2;                  // pixel blocks per wave
constexpr int kTilePix = kWaves * MB * 16;

struct Geo {
  int TW, RB, TH;      // pixel-block width, rows per pixel block, tile rows
  int tiles_x, tiles_y, VR;
};
inline Geo make_geo(int64_t B, int64_t H, int64_t W) {
  Geo g;
  g.TW = (W == 4 || W == 8) ? (int)W : 16;
  g.RB = 16 / g.TW;
  g.TH = kWaves * MB * g.RB;
  g.VR = (int)(B * (H + 1));   // tall image: every image is followed by one zero row
  g.tiles_x = (int)adnm_cdiv(W, g.TW);
  g.tiles_y = (int)adnm_cdiv(g.VR, g.TH);
  return g;
}

struct ConvArgs {
  const float* in;  int64_t ldin;     // (B*H*W, K) pixel rows
  const float* in2; int64_t ldin2;    // dgrad: pre-activation of the forward output; the staged value is in * act'(in2)
  const float* w; int64_t sn, st, sk; int flip;   // W(n, tap, k) = w[n*sn + (flip ? 8-tap : tap)*st + k*sk]
  const float* bias;
  float* out; int64_t ldo;            // act(acc + bias)
  float* pre; int64_t ldpre;          // acc + bias (saved for backward), or NULL
  float* part;                        // nsplit > 1: partials [z][B*H*W][N]
  int B, H, W, K, N, nsplit, chunks_per_split;
  int TW, RB, TH, VR, tiles_x;
  int vec_in, vec_w, vec_out;         // 16-byte access is legal for the input rows / the weight's reduction axis / the output rows
  AdnmQuant* q;                       // quantisation record (fp8 scales, amax collection) or NULL
};

// pixel row of tall-image position (v, x), or -1 outside the images
__device__ __forceinline__ int64_t pixel_of(int v, int x, int B, int H, int W, int VR) {
  if (v < 0 || v >= VR || x < 0 || x >= W) return -1;
  const int b = v / (H + 1), y = v - b * (H + 1);
  return y < H ? ((int64_t)b * H + y) * W + x : -1;
}

// ---- stage the (TH+2) x (TW+2) tile of channels [c0, c0+16) of (in [* act'(in2)]) into LDS, zeros outside
// (scale: the fp8 mode stages value * scale; amax: running max |value| of what this thread staged, before scaling)
template <int ACT>
__device__ __forceinline__ void stage_tile(const float* __restrict__ in, int64_t ldin, const float* __restrict__ in2, int64_t ldin2, bool vec,
                                           int K, int B, int H, int W, int VR, int TW, int TH, float* sIn, int v0, int x0, int c0,
                                           float scale = 1.f, float* amax = nullptr) {
  const int TWp = TW + 2, PT = (TH + 2) * TWp;
  for (int it = threadIdx.x; it < PT * 4; it += kBlock) {
    const int pix = it >> 2, q = it & 3, r = pix / TWp, c = pix - r * TWp, ch = c0 + 4 * q;
    const int64_t p = pixel_of(v0 - 1 + r, x0 - 1 + c, B, H, W, VR);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (p >= 0 && ch < K) {
      if (vec && ch + 3 < K) {
        const float4 t = *reinterpret_cast<const float4*>(in + p * ldin + ch);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        if (ACT != ADNM_ACT_NONE) {
          const float4 u = *reinterpret_cast<const float4*>(in2 + p * ldin2 + ch);
          v[0] *= act_grad<ACT>(u.x); v[1] *= act_grad<ACT>(u.y); v[2] *= act_grad<ACT>(u.z); v[3] *= act_grad<ACT>(u.w);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ch + e < K) {
            v[e] = in[p * ldin + ch + e];
            if (ACT != ADNM_ACT_NONE) v[e] *= act_grad<ACT>(in2[p * ldin2 + ch + e]);
          }
      }
    }
    if (amax) *amax = adnm_amax4(*amax, v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(sIn + pix * CKP + 4 * q) = make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
  }
}

// ================================================================================================ forward / dgrad
// epilogue shared by the two gather-GEMM kernels: bias, activation, pre-activation copy, or the split run's partial tile.
// (Measured, round 4: parking the tile in LDS to write whole pixel rows — NB*64 contiguous bytes instead of 64 — gained 4 us on the
// full-resolution GELU convs and lost 1-2 us on every other one to its two barriers: not kept.  tools/kbench_conv.py)
template <int NB, int ACT_OUT>
__device__ __forceinline__ void conv3_epilogue(const ConvArgs& a, const f32x4 (&acc)[NB][MB], int wave, int kk, int dyj, int dxj, int x0, int v0,
                                               int n0) {
  // D layout: row = channel (kk*4 + reg) of the block, column = pixel j
  const int64_t M = (int64_t)a.B * a.H * a.W;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int64_t p = pixel_of(v0 + (wave * MB + mb) * a.RB + dyj, x0 + dxj, a.B, a.H, a.W, a.VR);
    if (p < 0) continue;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = n0 + nb * 16 + 4 * kk;
      if (n >= a.N) continue;
      float v[4] = {acc[nb][mb][0], acc[nb][mb][1], acc[nb][mb][2], acc[nb][mb][3]};
      if (a.nsplit > 1) {   // N % 4 == 0 is required for split runs (host-checked)
        *reinterpret_cast<float4*>(a.part + ((int64_t)blockIdx.z * M + p) * a.N + n) = make_float4(v[0], v[1], v[2], v[3]);
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (a.bias && n + r < a.N) v[r] += a.bias[n + r];
      if (a.vec_out && n + 3 < a.N) {
        if (a.pre) *reinterpret_cast<float4*>(a.pre + p * a.ldpre + n) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(a.out + p * a.ldo + n) = make_float4(act_fwd<ACT_OUT>(v[0]), act_fwd<ACT_OUT>(v[1]), act_fwd<ACT_OUT>(v[2]), act_fwd<ACT_OUT>(v[3]));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) {
            if (a.pre) a.pre[p * a.ldpre + n + r] = v[r];
            a.out[p * a.ldo + n + r] = act_fwd<ACT_OUT>(v[r]);
          }
      }
    }
  }
}

// LDS: [input tile (TH+2)(TW+2) x CKP] [weights of the chunk: 9 taps x NB*16 channels x CKP]
// PREC = ADNM_MFMA_*; A_BF8: in the fp8 mode the pixel rows are a gradient (e5m2) — the dgrad use.
template <int NB, int ACT_IN, int ACT_OUT, int PREC, bool A_BF8>
__global__ __launch_bounds__(kBlock) void conv3_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // fp8: the tile and the weights are staged already multiplied by their per-tensor scales (the accumulators are un-scaled in the
  // epilogue); rec_a / rec_b (workgroup-uniform): collect max |value| of the pixels / weights staged — the first channel group /
  // first pixel tile only, so every element is seen once (halo pixels twice, harmless for a max)
  float q_sa = 1.f, q_sb = 1.f, amax_a = 0.f, amax_b = 0.f;
  bool rec_a = false, rec_b = false;
  if (a.q) {
    if (PREC == ADNM_MFMA_FP8) q_sa = a.q->scale_a, q_sb = a.q->scale_b;
    const bool rec = a.q->record != 0.f;
    rec_a = rec && blockIdx.y == 0, rec_b = rec && blockIdx.x == 0;
  }
  const int TWp = a.TW + 2;
  float* sIn = smem;
  float* sW = smem + (a.TH + 2) * TWp * CKP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, kk = lane >> 4;
  const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
  const int x0 = tx * a.TW, v0 = ty * a.TH, n0 = blockIdx.y * NB * 16;
  const int dyj = j / a.TW, dxj = j - dyj * a.TW;
  int base[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) base[mb] = (((wave * MB + mb) * a.RB + dyj) * TWp + dxj) * CKP + 4 * kk;
  f32x4 acc[NB][MB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nchunks = (a.K + CK - 1) / CK;
  const int cbeg = blockIdx.z * a.chunks_per_split;
  const int cend = cbeg + a.chunks_per_split < nchunks ? cbeg + a.chunks_per_split : nchunks;
  // Register-staged pipeline over the 16-channel chunks: the global loads of chunk c+1 (input tile + halo, the chunk's weights) are
  // issued right after chunk c went to LDS and fly under its 9-tap MFMA loop — one exposed memory round trip per workgroup instead of one
  // per chunk (these kernels ran at ~1 TB/s on the 128x128 maps: 2-4 chunks x two barriers with the loads in between).
  constexpr int kItIn = 4;                                   // input items per thread: (TH+2)(TW+2) pixels x 4 channel quads / 256 <= 3.2
  constexpr int kItW = (9 * NB * 64 + kBlock - 1) / kBlock;   // weight items per thread
  const int PT = (a.TH + 2) * TWp;
  float rin[kItIn][4], rw[kItW][4];
  auto load_chunk = [&](int c) {
    const int c0 = c * CK;
#pragma unroll
    for (int u = 0; u < kItIn; ++u) {
      const int it = threadIdx.x + u * kBlock;
      const int pix = it >> 2, q = it & 3, r = pix / TWp, cc = pix - r * TWp, ch = c0 + 4 * q;
      const int64_t p = it < PT * 4 ? pixel_of(v0 - 1 + r, x0 - 1 + cc, a.B, a.H, a.W, a.VR) : -1;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (p >= 0 && ch < a.K) {
        if (a.vec_in && ch + 3 < a.K) {
          const float4 t = *reinterpret_cast<const float4*>(a.in + p * a.ldin + ch);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          if (ACT_IN != ADNM_ACT_NONE) {
            const float4 g = *reinterpret_cast<const float4*>(a.in2 + p * a.ldin2 + ch);
            v[0] *= act_grad<ACT_IN>(g.x); v[1] *= act_grad<ACT_IN>(g.y); v[2] *= act_grad<ACT_IN>(g.z); v[3] *= act_grad<ACT_IN>(g.w);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ch + e < a.K) {
              v[e] = a.in[p * a.ldin + ch + e];
              if (ACT_IN != ADNM_ACT_NONE) v[e] *= act_grad<ACT_IN>(a.in2[p * a.ldin2 + ch + e]);
            }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) rin[u][e] = v[e];
    }
    // weights of this chunk: sW[(tap*NB*16 + nl)*CKP + kq*4 .. +3] = W(n0 + nl, tap, c*16 + 4 kq ..)
#pragma unroll
    for (int u = 0; u < kItW; ++u) {
      const int it = threadIdx.x + u * kBlock;
      const int kq = it & 3, nl = (it >> 2) % (NB * 16), tap = it / (NB * 64);
      const int n = n0 + nl, k0 = c0 + 4 * kq;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (it < 9 * NB * 64 && n < a.N && k0 < a.K) {
        const float* wp = a.w + (int64_t)n * a.sn + (int64_t)(a.flip ? 8 - tap : tap) * a.st + (int64_t)k0 * a.sk;
        if (a.vec_w && k0 + 3 < a.K) {
          const float4 t = *reinterpret_cast<const float4*>(wp);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + e < a.K) v[e] = wp[(int64_t)e * a.sk];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) rw[u][e] = v[e];
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int u = 0; u < kItIn; ++u) {
      const int it = threadIdx.x + u * kBlock;
      if (it >= PT * 4) break;
      if (rec_a) amax_a = adnm_amax4(amax_a, rin[u][0], rin[u][1], rin[u][2], rin[u][3]);
      *reinterpret_cast<float4*>(sIn + (it >> 2) * CKP + 4 * (it & 3)) = make_float4(rin[u][0] * q_sa, rin[u][1] * q_sa, rin[u][2] * q_sa, rin[u][3] * q_sa);
    }
#pragma unroll
    for (int u = 0; u < kItW; ++u) {
      const int it = threadIdx.x + u * kBlock;
      if (it >= 9 * NB * 64) break;
      const int kq = it & 3, nl = (it >> 2) % (NB * 16), tap = it / (NB * 64);
      if (rec_b) amax_b = adnm_amax4(amax_b, rw[u][0], rw[u][1], rw[u][2], rw[u][3]);
      *reinterpret_cast<float4*>(sW + (tap * NB * 16 + nl) * CKP + 4 * kq) = make_float4(rw[u][0] * q_sb, rw[u][1] * q_sb, rw[u][2] * q_sb, rw[u][3] * q_sb);
    }
  };
  if (cbeg < cend) load_chunk(cbeg);
  for (int c = cbeg; c < cend; ++c) {
    __syncthreads();   // every wave has finished reading the previous chunk's LDS images
    store_chunk();
    __syncthreads();
    if (c + 1 < cend) load_chunk(c + 1);   // in flight under this chunk's MFMAs
    // one MFMA step = 32 reduction steps = the chunk's 16 channels of TWO taps (tap 8 runs on a zero upper half)
#pragma unroll
    for (int tp = 0; tp < 9; tp += 2) {
      float wa[NB][2][4], xb[MB][2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tap = tp + h < 9 ? tp + h : 8;
        const int toff = ((tap / 3) * TWp + (tap % 3)) * CKP;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float4 t = *reinterpret_cast<const float4*>(sW + (tap * NB * 16 + nb * 16 + j) * CKP + 4 * kk);
          wa[nb][h][0] = t.x; wa[nb][h][1] = t.y; wa[nb][h][2] = t.z; wa[nb][h][3] = t.w;
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float4 t = *reinterpret_cast<const float4*>(sIn + base[mb] + toff);
          xb[mb][h][0] = t.x; xb[mb][h][1] = t.y; xb[mb][h][2] = t.z; xb[mb][h][3] = t.w;
        }
      }
      const bool pair = tp + 1 < 9;
      AdnmFrag<PREC> fw[NB], fx[MB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) fw[nb] = adnm_make_frag<PREC, false>(wa[nb][0], pair ? wa[nb][1] : nullptr, 1.f);   // (already scaled)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) fx[mb] = adnm_make_frag<PREC, A_BF8>(xb[mb][0], pair ? xb[mb][1] : nullptr, 1.f);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          if (tp + 1 < 9) acc[nb][mb] = adnm_mma<PREC, false, A_BF8, false>(fw[nb], fx[mb], acc[nb][mb]);
          else acc[nb][mb] = adnm_mma<PREC, false, A_BF8, true>(fw[nb], fx[mb], acc[nb][mb]);
        }
    }
  }
  if (rec_a) adnm_amax_commit(&a.q->amax_a, amax_a);
  if (rec_b) adnm_amax_commit(&a.q->amax_b, amax_b);
  if (PREC == ADNM_MFMA_FP8) {   // back to the operands' own scale (before partials are written, bias is added)
    const float inv = 1.0f / (q_sa * q_sb);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = acc[nb][mb] * inv;
  }
  conv3_epilogue<NB, ACT_OUT>(a, acc, wave, kk, dyj, dxj, x0, v0, n0);
}

// ---- the bf16 configuration: bf16 IMAGES in LDS (round 4).  conv3_kernel keeps fp32 images and rounds where a lane forms its fragment —
// every staged value is read by 9 taps (x the NB / MB blocks that share it), so it was converted nine times over, and a fragment cost two
// 16-byte LDS reads.  Here a value is rounded ONCE when its tile is staged; the same 80-byte pixel pitch now holds 32 channels, a lane's
// fragment of a 32-step MFMA is ONE ds_read_b128 (8 channels of one tap; conflict-free as before: 5 is coprime with 16), a chunk is 32
// channels of all nine taps (no half-empty step for the ninth tap), and there are half as many chunk rounds (two barriers each).
// Same results as rounding per use up to the fp32 summation order inside the accumulator.
template <int NB, int ACT_IN, int ACT_OUT>
__global__ __launch_bounds__(kBlock) void conv3_bf16_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CN = 2 * CK;   // channels per chunk
  float amax_a = 0.f, amax_b = 0.f;
  bool rec_a = false, rec_b = false;
  if (a.q) {   // a calibrating pass of the fp8 configuration runs bf16 operands and collects the amax of what it staged
    const bool rec = a.q->record != 0.f;
    rec_a = rec && blockIdx.y == 0, rec_b = rec && blockIdx.x == 0;
  }
  const int TWp = a.TW + 2;
  float* sIn = smem;
  float* sW = smem + (a.TH + 2) * TWp * CKP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, kk = lane >> 4;
  const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
  const int x0 = tx * a.TW, v0 = ty * a.TH, n0 = blockIdx.y * NB * 16;
  const int dyj = j / a.TW, dxj = j - dyj * a.TW;
  int base[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) base[mb] = (((wave * MB + mb) * a.RB + dyj) * TWp + dxj) * CKP + 4 * kk;
  f32x4 acc[NB][MB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this split's channel range (chunks_per_split counts 16-channel units and is even for split runs: plan_split)
  const int kbeg = blockIdx.z * a.chunks_per_split * CK;
  const int kend = kbeg + a.chunks_per_split * CK < a.K ? kbeg + a.chunks_per_split * CK : a.K;
  constexpr int kItIn = 4;                                    // (TH+2)(TW+2) pixels x 4 channel octets / 256 <= 3.2
  constexpr int kItW = (9 * NB * 64 + kBlock - 1) / kBlock;   // 9 taps x NB*16 output channels x 4 octets
  const int PT = (a.TH + 2) * TWp;
  float rin[kItIn][8], rw[kItW][8];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int u = 0; u < kItIn; ++u) {
      const int it = threadIdx.x + u * kBlock;
      const int pix = it >> 2, q = it & 3, r = pix / TWp, cc = pix - r * TWp, ch = c0 + 8 * q;
      const int64_t p = it < PT * 4 ? pixel_of(v0 - 1 + r, x0 - 1 + cc, a.B, a.H, a.W, a.VR) : -1;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (p >= 0 && ch < kend) {
        if (a.vec_in && ch + 7 < kend) {
          const float4 t0 = *reinterpret_cast<const float4*>(a.in + p * a.ldin + ch), t1 = *reinterpret_cast<const float4*>(a.in + p * a.ldin + ch + 4);
          v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
          if (ACT_IN != ADNM_ACT_NONE) {   // (applied here, not when the chunk goes to LDS: between the barriers it was 35 % slower)
            const float4 g0 = *reinterpret_cast<const float4*>(a.in2 + p * a.ldin2 + ch), g1 = *reinterpret_cast<const float4*>(a.in2 + p * a.ldin2 + ch + 4);
            v[0] *= act_grad<ACT_IN>(g0.x); v[1] *= act_grad<ACT_IN>(g0.y); v[2] *= act_grad<ACT_IN>(g0.z); v[3] *= act_grad<ACT_IN>(g0.w);
            v[4] *= act_grad<ACT_IN>(g1.x); v[5] *= act_grad<ACT_IN>(g1.y); v[6] *= act_grad<ACT_IN>(g1.z); v[7] *= act_grad<ACT_IN>(g1.w);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (ch + e < kend) {
              v[e] = a.in[p * a.ldin + ch + e];
              if (ACT_IN != ADNM_ACT_NONE) v[e] *= act_grad<ACT_IN>(a.in2[p * a.ldin2 + ch + e]);
            }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) rin[u][e] = v[e];
    }
    // weights of this chunk: the octet at sW[(tap*NB*16 + nl)*CKP + 4 kq] = W(n0 + nl, tap, c0 + 8 kq ..)
#pragma unroll
    for (int u = 0; u < kItW; ++u) {
      const int it = threadIdx.x + u * kBlock;
      const int kq = it & 3, nl = (it >> 2) % (NB * 16), tap = it / (NB * 64);
      const int n = n0 + nl, k0 = c0 + 8 * kq;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (it < 9 * NB * 64 && n < a.N && k0 < kend) {
        const float* wp = a.w + (int64_t)n * a.sn + (int64_t)(a.flip ? 8 - tap : tap) * a.st + (int64_t)k0 * a.sk;
        if (a.vec_w && k0 + 7 < kend) {
          const float4 t0 = *reinterpret_cast<const float4*>(wp), t1 = *reinterpret_cast<const float4*>(wp + 4);
          v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (k0 + e < kend) v[e] = wp[(int64_t)e * a.sk];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) rw[u][e] = v[e];
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int u = 0; u < kItIn; ++u) {
      const int it = threadIdx.x + u * kBlock;
      if (it >= PT * 4) break;
      if (rec_a) amax_a = adnm_amax4(adnm_amax4(amax_a, rin[u][0], rin[u][1], rin[u][2], rin[u][3]), rin[u][4], rin[u][5], rin[u][6], rin[u][7]);
      *reinterpret_cast<adnm_bf16x8*>(sIn + (it >> 2) * CKP + 4 * (it & 3)) = adnm_pack_bf16x8(rin[u]);
    }
#pragma unroll
    for (int u = 0; u < kItW; ++u) {
      const int it = threadIdx.x + u * kBlock;
      if (it >= 9 * NB * 64) break;
      const int kq = it & 3, nl = (it >> 2) % (NB * 16), tap = it / (NB * 64);
      if (rec_b) amax_b = adnm_amax4(adnm_amax4(amax_b, rw[u][0], rw[u][1], rw[u][2], rw[u][3]), rw[u][4], rw[u][5], rw[u][6], rw[u][7]);
      *reinterpret_cast<adnm_bf16x8*>(sW + (tap * NB * 16 + nl) * CKP + 4 * kq) = adnm_pack_bf16x8(rw[u]);
    }
  };
  if (kbeg < kend) load_chunk(kbeg);
  for (int c0 = kbeg; c0 < kend; c0 += CN) {
    __syncthreads();   // every wave has finished reading the previous chunk's LDS images
    store_chunk();
    __syncthreads();
    if (c0 + CN < kend) load_chunk(c0 + CN);   // in flight under this chunk's MFMAs
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = ((tap / 3) * TWp + (tap % 3)) * CKP;
      adnm_bf16x8 fw[NB], fx[MB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) fw[nb] = *reinterpret_cast<const adnm_bf16x8*>(sW + (tap * NB * 16 + nb * 16 + j) * CKP + 4 * kk);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) fx[mb] = *reinterpret_cast<const adnm_bf16x8*>(sIn + base[mb] + toff);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nb], fx[mb], acc[nb][mb], 0, 0, 0);
    }
  }
  if (rec_a) adnm_amax_commit(&a.q->amax_a, amax_a);
  if (rec_b) adnm_amax_commit(&a.q->amax_b, amax_b);
  conv3_epilogue<NB, ACT_OUT>(a, acc, wave, kk, dyj, dxj, x0, v0, n0);
}

// split runs: out = act(sum_z part[z] + bias) (+ pre), one float4 per thread
template <int ACT_OUT>
__global__ __launch_bounds__(256) void conv3_join_kernel(const float* __restrict__ part, int nsplit, int64_t M, int N, const float* __restrict__ bias,
                                                         float* __restrict__ out, int64_t ldo, float* __restrict__ pre, int64_t ldpre) {
  const int nq = N >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * nq) return;
  const int64_t p = i / nq;
  const int n = (int)(i - p * nq) * 4;
  float4 s = *reinterpret_cast<const float4*>(part + p * N + n);
  for (int z = 1; z < nsplit; ++z) {
    const float4 t = *reinterpret_cast<const float4*>(part + ((int64_t)z * M + p) * N + n);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  if (bias) { s.x += bias[n]; s.y += bias[n + 1]; s.z += bias[n + 2]; s.w += bias[n + 3]; }
  if (pre) *reinterpret_cast<float4*>(pre + p * ldpre + n) = s;
  *reinterpret_cast<float4*>(out + p * ldo + n) = make_float4(act_fwd<ACT_OUT>(s.x), act_fwd<ACT_OUT>(s.y), act_fwd<ACT_OUT>(s.z), act_fwd<ACT_OUT>(s.w));
}

// ================================================================================================ wgrad
struct WgArgs {
  const float* dout; int64_t lddo;    // (M, N)
  const float* pre;  int64_t ldpre;   // pre-activation of the forward output (ACT != NONE)
  const float* in;   int64_t ldin;    // (M, K)
  float* part; int64_t rowlen;        // partial rows [gridDim.x][N*9*K (+ N)]
  int want_bias;
  int B, H, W, K, N, ntiles;
  int TW, RB, TH, VR, tiles_x;
  int vec_in, vec_do;
};

template <int NB, int ACT, int PREC>   // PREC: fp32 or bf16 (the fp8 configuration keeps bf16 operands for the weight gradient)
__global__ __launch_bounds__(kBlock) void conv3_wgrad_kernel(WgArgs a) {
  constexpr int DP = NB * 16 + 16;   // pitch of a dpre pixel row: (DP mod 32) == 16 -> conflict-free A-operand reads
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int TWp = a.TW + 2;
  float* sIn = smem;
  float* sD = smem + (a.TH + 2) * TWp * CKP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, kk = lane >> 4;
  const int c0 = blockIdx.y * CK, n0 = blockIdx.z * NB * 16;
  f32x4 acc[9][NB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  // Register-staged pipeline over the pixel tiles (as in conv3_kernel): the loads of the NEXT tile (input tile + halo, dout [* act'(pre)])
  // fly under the current tile's MFMAs.
  constexpr int kItIn = 4, kItD = (kTilePix * NB * 4 + kBlock - 1) / kBlock;
  const int PT = (a.TH + 2) * TWp;
  float rin[kItIn][4], rd[kItD][4];
  auto load_tile = [&](int tile) {
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x, x0 = tx * a.TW, v0 = ty * a.TH;
#pragma unroll
    for (int u = 0; u < kItIn; ++u) {
      const int it = threadIdx.x + u * kBlock;
      const int pix = it >> 2, q = it & 3, r = pix / TWp, cc = pix - r * TWp, ch = c0 + 4 * q;
      const int64_t p = it < PT * 4 ? pixel_of(v0 - 1 + r, x0 - 1 + cc, a.B, a.H, a.W, a.VR) : -1;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (p >= 0 && ch < a.K) {
        if (a.vec_in && ch + 3 < a.K) {
          const float4 t = *reinterpret_cast<const float4*>(a.in + p * a.ldin + ch);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ch + e < a.K) v[e] = a.in[p * a.ldin + ch + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) rin[u][e] = v[e];
    }
    // dpre tile: pixel block pb (0..7), pixel jj of it -> row pb*16 + jj; NB*16 channels from n0
#pragma unroll
    for (int u = 0; u < kItD; ++u) {
      const int it = threadIdx.x + u * kBlock;
      const int pxl = it / (NB * 4), q = it - pxl * (NB * 4), n = n0 + 4 * q;
      const int pb = pxl >> 4, jj = pxl & 15, dy_ = jj / a.TW, dx_ = jj - dy_ * a.TW;
      const int64_t p = it < kTilePix * NB * 4 ? pixel_of(v0 + pb * a.RB + dy_, x0 + dx_, a.B, a.H, a.W, a.VR) : -1;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (p >= 0 && n < a.N) {
        if (a.vec_do && n + 3 < a.N) {
          const float4 t = *reinterpret_cast<const float4*>(a.dout + p * a.lddo + n);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          if (ACT != ADNM_ACT_NONE) {
            const float4 g = *reinterpret_cast<const float4*>(a.pre + p * a.ldpre + n);
            v[0] *= act_grad<ACT>(g.x); v[1] *= act_grad<ACT>(g.y); v[2] *= act_grad<ACT>(g.z); v[3] *= act_grad<ACT>(g.w);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < a.N) {
              v[e] = a.dout[p * a.lddo + n + e];
              if (ACT != ADNM_ACT_NONE) v[e] *= act_grad<ACT>(a.pre[p * a.ldpre + n + e]);
            }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) rd[u][e] = v[e];
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int u = 0; u < kItIn; ++u) {
      const int it = threadIdx.x + u * kBlock;
      if (it >= PT * 4) break;
      *reinterpret_cast<float4*>(sIn + (it >> 2) * CKP + 4 * (it & 3)) = make_float4(rin[u][0], rin[u][1], rin[u][2], rin[u][3]);
    }
#pragma unroll
    for (int u = 0; u < kItD; ++u) {
      const int it = threadIdx.x + u * kBlock;
      if (it >= kTilePix * NB * 4) break;
      const int pxl = it / (NB * 4), q = it - pxl * (NB * 4);
      *reinterpret_cast<float4*>(sD + pxl * DP + 4 * q) = make_float4(rd[u][0], rd[u][1], rd[u][2], rd[u][3]);
    }
  };
  if ((int)blockIdx.x < a.ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();   // every wave has finished reading the previous tile's LDS images
    store_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) load_tile(tile + gridDim.x);   // in flight under this tile's MFMAs
    // reduction = the pixels of the wave's MB = 2 pixel blocks: lane (., kk) takes pixels 4 kk .. 4 kk + 3 of each (step e) — 8 per lane =
    // one 32-step MFMA group
    static_assert(MB == 2, "the weight-gradient step pairs the wave's two pixel blocks");
    float av[NB][2][4];
    const float* bp[2][4];
#pragma unroll
    for (int pbw = 0; pbw < MB; ++pbw) {
      const int pb = wave * MB + pbw;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int jj = 4 * kk + e, pxl = pb * 16 + jj, dy_ = jj / a.TW, dx_ = jj - dy_ * a.TW;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) av[nb][pbw][e] = sD[pxl * DP + nb * 16 + j];   // A[i = channel][k = pixel]
        bp[pbw][e] = sIn + ((pb * a.RB + dy_) * TWp + dx_) * CKP + j;                   // B[k = pixel][j = input channel], tap (0,0)
      }
    }
    AdnmFrag<PREC> fa[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) fa[nb] = adnm_make_frag<PREC, false>(av[nb][0], av[nb][1], 1.f);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = ((tap / 3) * TWp + (tap % 3)) * CKP;
      const float b0[4] = {bp[0][0][toff], bp[0][1][toff], bp[0][2][toff], bp[0][3][toff]};
      const float b1[4] = {bp[1][0][toff], bp[1][1][toff], bp[1][2][toff], bp[1][3][toff]};
      const AdnmFrag<PREC> fb = adnm_make_frag<PREC, false>(b0, b1, 1.f);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[tap][nb] = adnm_mma<PREC, false, false>(fa[nb], fb, acc[tap][nb]);
    }
    if (a.want_bias && blockIdx.y == 0 && lane < NB * 16) {
#pragma unroll 8
      for (int p = 0; p < MB * 16; ++p) bsum += sD[(wave * MB * 16 + p) * DP + lane];
    }
  }
  // sum the 4 waves through LDS (fixed order), then ONE partial row per workgroup: element (n, tap, k) at (n*9 + tap)*K + k;
  // D layout: row = channel kk*4 + reg, column = input channel j
  __syncthreads();
  float* red = smem;   // [kWaves][9*NB*4][64] floats (+ bias lanes) — fits: 4 * (72 + 1) * 64 * 4 B = 74.8 KB for NB = 2
  constexpr int kPer = 9 * NB * 4;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * (kPer + 1) + (tap * NB + nb) * 4 + r) * 64 + lane] = acc[tap][nb][r];
  red[(wave * (kPer + 1) + kPer) * 64 + lane] = bsum;
  __syncthreads();
  float* dst = a.part + (int64_t)blockIdx.x * a.rowlen;
  for (int e = threadIdx.x; e < (kPer + 1) * 64; e += kBlock) {
    const float v = (red[e] + red[(kPer + 1) * 64 + e]) + (red[2 * (kPer + 1) * 64 + e] + red[3 * (kPer + 1) * 64 + e]);
    const int slot = e >> 6, ln = e & 63;
    if (slot == kPer) {
      if (a.want_bias && blockIdx.y == 0 && ln < NB * 16 && n0 + ln < a.N) dst[(int64_t)a.N * 9 * a.K + n0 + ln] = v;
      continue;
    }
    const int r = slot & 3, nb = (slot >> 2) % NB, tap = slot / (4 * NB);
    const int n = n0 + nb * 16 + (ln >> 4) * 4 + r, k = c0 + (ln & 15);
    if (n < a.N && k < a.K) dst[((int64_t)n * 9 + tap) * a.K + k] = v;
  }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int check_shape(const char* who, int64_t B, int64_t H, int64_t W, int64_t K, int64_t N) {
  ADNM_REQUIRE(B > 0 && H > 0 && W > 0 && K > 0 && N > 0, "%s: empty shape", who);
  ADNM_REQUIRE(B * (H + 1) < (1ll << 30) && B * H * W < (1ll << 31) && K <= 65536 && N <= 65536, "%s: shape too large", who);
  return ADNM_OK;
}

struct Split {
  int nb, ngroups, nsplit, cps;
};
inline Split plan_split(const Geo& g, int64_t K, int64_t N) {
  Split s;
  s.nb = N > 32 ? 4 : (N > 16 ? 2 : 1);
  s.ngroups = (int)adnm_cdiv(N, s.nb * 16);
  const int nchunks = (int)adnm_cdiv(K, CK);
  const int64_t blocks = (int64_t)g.tiles_x * g.tiles_y * s.ngroups;
  int want = 1;
  if (blocks < 256 && nchunks >= 4 && N % 4 == 0) {   // deep maps: few pixels, long reductions
    want = (int)adnm_cdiv(512, blocks);
    if (want > nchunks / 2) want = nchunks / 2;
    if (want > 16) want = 16;
    if (want < 1) want = 1;
  }
  s.cps = (int)adnm_cdiv(nchunks, want);
  if (want > 1 && (s.cps & 1)) ++s.cps;   // whole 32-channel chunks per split: the bf16 kernel's chunk is two of these units
  s.nsplit = (int)adnm_cdiv(nchunks, s.cps);
  return s;
}

template <int ACT_IN, int ACT_OUT>
int launch_conv(const ConvArgs& a, const Geo& g, const Split& s, int prec, hipStream_t st, const char* prof, double bytes) {
  const size_t smem = sizeof(float) * ((size_t)(g.TH + 2) * (g.TW + 2) * CKP + (size_t)9 * s.nb * 16 * CKP);
  const dim3 grid((unsigned)(g.tiles_x * g.tiles_y), (unsigned)s.ngroups, (unsigned)s.nsplit);
  ADNM_PROF(prof, st, bytes);
#define CV(PRECV, BF8V)                                                                                  \
  do {                                                                                                   \
    if (s.nb == 4) conv3_kernel<4, ACT_IN, ACT_OUT, PRECV, BF8V><<<grid, kBlock, smem, st>>>(a);         \
    else if (s.nb == 2) conv3_kernel<2, ACT_IN, ACT_OUT, PRECV, BF8V><<<grid, kBlock, smem, st>>>(a);    \
    else conv3_kernel<1, ACT_IN, ACT_OUT, PRECV, BF8V><<<grid, kBlock, smem, st>>>(a);                   \
  } while (0)
  static const bool fp32_images = getenv("ADNM_CONV3_FP32_IMAGES") && atoi(getenv("ADNM_CONV3_FP32_IMAGES")) != 0;   // measurement aid: the round-3 kernel
  if (prec == ADNM_MFMA_BF16 && fp32_images) CV(ADNM_MFMA_BF16, false);
  else if (prec == ADNM_MFMA_BF16) {
    if (s.nb == 4) conv3_bf16_kernel<4, ACT_IN, ACT_OUT><<<grid, kBlock, smem, st>>>(a);
    else if (s.nb == 2) conv3_bf16_kernel<2, ACT_IN, ACT_OUT><<<grid, kBlock, smem, st>>>(a);
    else conv3_bf16_kernel<1, ACT_IN, ACT_OUT><<<grid, kBlock, smem, st>>>(a);
  } else if (prec == ADNM_MFMA_FP8) CV(ADNM_MFMA_FP8, false);
  else if (prec == ADNM_MFMA_FP8_GRAD) CV(ADNM_MFMA_FP8, true);
  else CV(ADNM_MFMA_F32, false);
#undef CV
  return ADNM_OK;
}

void fill_geo(ConvArgs& a, const Geo& g) { a.TW = g.TW, a.RB = g.RB, a.TH = g.TH, a.VR = g.VR, a.tiles_x = g.tiles_x; }

}  // namespace

extern "C" int64_t adnm_conv3_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t K, int64_t N) {
  if (B <= 0 || H <= 0 || W <= 0 || K <= 0 || N <= 0) return 0;
  const Geo g = make_geo(B, H, W);
  const Split s = plan_split(g, K, N);
  return s.nsplit > 1 ? (int64_t)s.nsplit * B * H * W * N * (int64_t)sizeof(float) : 16;
}

// out = act(conv3x3(in, w) + bias); pre (optional) receives the value before the activation.
// w element (n, ky, kx, k) at w[n*ws_n + (ky*3+kx)*ws_tap + k*ws_k] — both nn.Conv2d's (Cout,Cin,3,3) layout (ws_n=9K, ws_tap=1, ws_k=9)
// and the channels-last one (Cout,3,3,Cin) (ws_n=9K, ws_tap=K, ws_k=1) are read in place.
extern "C" int adnm_conv3_fwd(const float* in, int64_t ldin, const float* w, int64_t ws_n, int64_t ws_tap, int64_t ws_k, const float* bias,
                              float* out, int64_t ldo, float* pre, int64_t ldpre, void* ws, int64_t ws_bytes, int64_t B, int64_t H, int64_t W,
                              int64_t K, int64_t N, int act, int prec, float* q, adnm_stream_t stream) {
  if (int rc = check_shape("conv3_fwd", B, H, W, K, N)) return rc;
  ADNM_REQUIRE(in && w && out, "conv3_fwd: null pointer");
  ADNM_REQUIRE(prec >= ADNM_MFMA_F32 && prec <= ADNM_MFMA_FP8, "conv3_fwd: bad prec %d", prec);
  ADNM_REQUIRE(prec != ADNM_MFMA_FP8 || q, "conv3_fwd: the fp8 mode needs a quantisation record");
  ADNM_REQUIRE(ldin >= K && ldo >= N && (!pre || ldpre >= N), "conv3_fwd: row strides smaller than the rows");
  ADNM_REQUIRE(act == ADNM_ACT_NONE || act == ADNM_ACT_GELU, "conv3_fwd: activation %d not in {none, gelu}", act);
  const Geo g = make_geo(B, H, W);
  const Split s = plan_split(g, K, N);
  if (s.nsplit > 1 && (!ws || ws_bytes < adnm_conv3_ws_bytes(B, H, W, K, N))) {
    adnm_set_error("conv3_fwd: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_conv3_ws_bytes(B, H, W, K, N));
    return ADNM_EWORKSPACE;
  }
  ConvArgs a{};
  a.in = in, a.ldin = ldin, a.in2 = nullptr, a.ldin2 = 0;
  a.w = w, a.sn = ws_n, a.st = ws_tap, a.sk = ws_k, a.flip = 0;
  a.bias = bias, a.out = out, a.ldo = ldo, a.pre = pre, a.ldpre = ldpre, a.part = (float*)ws;
  a.B = (int)B, a.H = (int)H, a.W = (int)W, a.K = (int)K, a.N = (int)N, a.nsplit = s.nsplit, a.chunks_per_split = s.cps;
  fill_geo(a, g);
  a.vec_in = al16(in) && ldin % 4 == 0 && K % 4 == 0;
  a.vec_w = ws_k == 1 && al16(w) && ws_n % 4 == 0 && ws_tap % 4 == 0 && K % 4 == 0;
  a.vec_out = al16(out) && ldo % 4 == 0 && N % 4 == 0 && (!pre || (al16(pre) && ldpre % 4 == 0));
  a.q = reinterpret_cast<AdnmQuant*>(q);
  hipStream_t st = (hipStream_t)stream;
  const double bytes = 4.0 * ((double)B * H * W * (K + N * (pre ? 2 : 1)) + 9.0 * K * N);
  if (act == ADNM_ACT_GELU && s.nsplit == 1) launch_conv<ADNM_ACT_NONE, ADNM_ACT_GELU>(a, g, s, prec, st, "conv3_fwd", bytes);
  else launch_conv<ADNM_ACT_NONE, ADNM_ACT_NONE>(a, g, s, prec, st, "conv3_fwd", bytes);
  ADNM_CHECK_LAUNCH("conv3_fwd");
  if (s.nsplit > 1) {
    const int64_t M = B * H * W, nthreads = M * (N / 4);
    ADNM_PROF("conv3_join", st, 4.0 * M * N * (s.nsplit + 1 + (pre ? 1 : 0)));
    if (act == ADNM_ACT_GELU)
      conv3_join_kernel<ADNM_ACT_GELU><<<(unsigned)adnm_cdiv(nthreads, 256), 256, 0, st>>>((const float*)ws, s.nsplit, M, (int)N, bias, out, ldo, pre, ldpre);
    else
      conv3_join_kernel<ADNM_ACT_NONE><<<(unsigned)adnm_cdiv(nthreads, 256), 256, 0, st>>>((const float*)ws, s.nsplit, M, (int)N, bias, out, ldo, pre, ldpre);
    ADNM_CHECK_LAUNCH("conv3_join");
  }
  return ADNM_OK;
}

// din = conv3x3^T(dout * act'(pre), w): the input gradient of adnm_conv3_fwd (K = Cin, N = Cout of the forward conv, same w strides).
extern "C" int adnm_conv3_dgrad(const float* dout, int64_t lddo, const float* pre, int64_t ldpre, int act, const float* w, int64_t ws_n,
                                int64_t ws_tap, int64_t ws_k, float* din, int64_t lddin, void* ws, int64_t ws_bytes, int64_t B, int64_t H,
                                int64_t W, int64_t K, int64_t N, int prec, float* q, adnm_stream_t stream) {
  if (int rc = check_shape("conv3_dgrad", B, H, W, K, N)) return rc;
  ADNM_REQUIRE(dout && w && din, "conv3_dgrad: null pointer");
  ADNM_REQUIRE(prec >= ADNM_MFMA_F32 && prec <= ADNM_MFMA_FP8_GRAD, "conv3_dgrad: bad prec %d", prec);
  ADNM_REQUIRE((prec != ADNM_MFMA_FP8 && prec != ADNM_MFMA_FP8_GRAD) || q, "conv3_dgrad: the fp8 modes need a quantisation record");
  if (prec == ADNM_MFMA_FP8) prec = ADNM_MFMA_FP8_GRAD;   // the pixel rows of this op are always a gradient
  ADNM_REQUIRE(act == ADNM_ACT_NONE || (act == ADNM_ACT_GELU && pre), "conv3_dgrad: activation %d needs the saved pre-activation", act);
  ADNM_REQUIRE(lddo >= N && lddin >= K && (!pre || ldpre >= N), "conv3_dgrad: row strides smaller than the rows");
  const Geo g = make_geo(B, H, W);
  const Split s = plan_split(g, N, K);   // reduction over the forward's output channels, K columns out
  if (s.nsplit > 1 && (!ws || ws_bytes < adnm_conv3_ws_bytes(B, H, W, N, K))) {
    adnm_set_error("conv3_dgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)adnm_conv3_ws_bytes(B, H, W, N, K));
    return ADNM_EWORKSPACE;
  }
  ConvArgs a{};
  a.in = dout, a.ldin = lddo, a.in2 = pre, a.ldin2 = ldpre;
  a.w = w, a.sn = ws_k, a.st = ws_tap, a.sk = ws_n, a.flip = 1;   // W'(k, tap, n) = W(n, 8 - tap, k)
  a.bias = nullptr, a.out = din, a.ldo = lddin, a.pre = nullptr, a.ldpre = 0, a.part = (float*)ws;
  a.B = (int)B, a.H = (int)H, a.W = (int)W, a.K = (int)N, a.N = (int)K, a.nsplit = s.nsplit, a.chunks_per_split = s.cps;
  fill_geo(a, g);
  a.vec_in = al16(dout) && lddo % 4 == 0 && N % 4 == 0 && (!pre || (al16(pre) && ldpre % 4 == 0));
  a.vec_w = a.sk == 1 && al16(w) && a.sn % 4 == 0 && a.st % 4 == 0 && N % 4 == 0;
  a.vec_out = al16(din) && lddin % 4 == 0 && K % 4 == 0;
  a.q = reinterpret_cast<AdnmQuant*>(q);
  hipStream_t st = (hipStream_t)stream;
  const double bytes = 4.0 * ((double)B * H * W * (K + N * (act != ADNM_ACT_NONE ? 2 : 1)) + 9.0 * K * N);
  if (act == ADNM_ACT_GELU) launch_conv<ADNM_ACT_GELU, ADNM_ACT_NONE>(a, g, s, prec, st, "conv3_dgrad", bytes);
  else launch_conv<ADNM_ACT_NONE, ADNM_ACT_NONE>(a, g, s, prec, st, "conv3_dgrad", bytes);
  ADNM_CHECK_LAUNCH("conv3_dgrad");
  if (s.nsplit > 1) {
    const int64_t M = B * H * W, nthreads = M * (K / 4);
    ADNM_PROF("conv3_join", st, 4.0 * M * K * (s.nsplit + 1));
    conv3_join_kernel<ADNM_ACT_NONE><<<(unsigned)adnm_cdiv(nthreads, 256), 256, 0, st>>>((const float*)ws, s.nsplit, M, (int)K, nullptr, din, lddin, nullptr, 0);
    ADNM_CHECK_LAUNCH("conv3_join");
  }
  return ADNM_OK;
}

namespace {
struct WgPlan {
  int nb, cogroups, cichunks, rows;
  int64_t rowlen;
};
inline WgPlan plan_wgrad(const Geo& g, int64_t K, int64_t N, bool bias) {
  WgPlan p;
  p.nb = N > 16 ? 2 : 1;                                                   // 32 output channels per workgroup: 18 accumulator blocks per wave
  p.cogroups = (int)adnm_cdiv(N, p.nb * 16);
  p.cichunks = (int)adnm_cdiv(K, CK);
  p.rowlen = N * 9 * K + (bias ? N : 0);
  const int64_t ntiles = (int64_t)g.tiles_x * g.tiles_y;
  int64_t r = adnm_cdiv(768, (int64_t)p.cichunks * p.cogroups);              // ~768 workgroups (3 per CU)
  const int64_t by_mem = (int64_t)(8 << 20) / (p.rowlen * 4);              // partials capped at ~8 MB
  if (r > by_mem) r = by_mem;
  if (r > ntiles) r = ntiles;
  if (r > 256) r = 256;
  p.rows = (int)(r < 1 ? 1 : r);
  return p;
}
}  // namespace

extern "C" int64_t adnm_conv3_wgrad_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t K, int64_t N) {
  if (B <= 0 || H <= 0 || W <= 0 || K <= 0 || N <= 0) return 0;
  const WgPlan p = plan_wgrad(make_geo(B, H, W), K, N, true);
  return (int64_t)p.rows * p.rowlen * (int64_t)sizeof(float);
}

// dw[n][tap][k] (contiguous, = the channels-last weight layout) and dbias[n] (optional) of adnm_conv3_fwd.  OVERWRITES both.
extern "C" int adnm_conv3_wgrad(const float* dout, int64_t lddo, const float* pre, int64_t ldpre, int act, const float* in, int64_t ldin,
                                float* dw, float* dbias, void* ws, int64_t ws_bytes, int64_t B, int64_t H, int64_t W, int64_t K, int64_t N,
                                int prec, adnm_stream_t stream) {
  if (prec == ADNM_MFMA_FP8 || prec == ADNM_MFMA_FP8_GRAD) prec = ADNM_MFMA_BF16;   // the weight gradient keeps bf16 operands in the fp8 configuration
  if (int rc = check_shape("conv3_wgrad", B, H, W, K, N)) return rc;
  ADNM_REQUIRE(dout && in && dw, "conv3_wgrad: null pointer");
  ADNM_REQUIRE(act == ADNM_ACT_NONE || (act == ADNM_ACT_GELU && pre), "conv3_wgrad: activation %d needs the saved pre-activation", act);
  ADNM_REQUIRE(lddo >= N && ldin >= K && (!pre || ldpre >= N), "conv3_wgrad: row strides smaller than the rows");
  ADNM_REQUIRE(N * 9 * K + N < (1ll << 31), "conv3_wgrad: weight too large");
  const Geo g = make_geo(B, H, W);
  const WgPlan p = plan_wgrad(g, K, N, dbias != nullptr);
  if (!ws || ws_bytes < (int64_t)p.rows * p.rowlen * 4) {
    adnm_set_error("conv3_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)((int64_t)p.rows * p.rowlen * 4));
    return ADNM_EWORKSPACE;
  }
  WgArgs a{};
  a.dout = dout, a.lddo = lddo, a.pre = pre, a.ldpre = ldpre, a.in = in, a.ldin = ldin;
  a.part = (float*)ws, a.rowlen = p.rowlen, a.want_bias = dbias != nullptr;
  a.B = (int)B, a.H = (int)H, a.W = (int)W, a.K = (int)K, a.N = (int)N, a.ntiles = g.tiles_x * g.tiles_y;
  a.TW = g.TW, a.RB = g.RB, a.TH = g.TH, a.VR = g.VR, a.tiles_x = g.tiles_x;
  a.vec_in = al16(in) && ldin % 4 == 0 && K % 4 == 0;
  a.vec_do = al16(dout) && lddo % 4 == 0 && N % 4 == 0 && (!pre || (al16(pre) && ldpre % 4 == 0));
  hipStream_t st = (hipStream_t)stream;
  size_t smem = sizeof(float) * ((size_t)(g.TH + 2) * (g.TW + 2) * CKP + (size_t)kTilePix * (p.nb * 16 + 16));
  const size_t red = sizeof(float) * (size_t)kWaves * (9 * p.nb * 4 + 1) * 64;   // the cross-wave sum reuses the same LDS
  if (smem < red) smem = red;
  const dim3 grid((unsigned)p.rows, (unsigned)p.cichunks, (unsigned)p.cogroups);
  {
    ADNM_PROF("conv3_wgrad", st, 4.0 * ((double)B * H * W * (K + N * (act != ADNM_ACT_NONE ? 2 : 1)) + 9.0 * K * N));
#define WG1(NBV, ACTV, BFV)                                                                   \
  do {                                                                                        \
    ADNM_ALLOW_LDS((conv3_wgrad_kernel<NBV, ACTV, BFV>), smem, "conv3_wgrad");                \
    conv3_wgrad_kernel<NBV, ACTV, BFV><<<grid, kBlock, smem, st>>>(a);                        \
  } while (0)
#define WG(NBV)                                                                               \
  do {                                                                                        \
    if (act == ADNM_ACT_GELU) { if (prec == ADNM_MFMA_BF16) WG1(NBV, ADNM_ACT_GELU, ADNM_MFMA_BF16); else WG1(NBV, ADNM_ACT_GELU, ADNM_MFMA_F32); } \
    else { if (prec == ADNM_MFMA_BF16) WG1(NBV, ADNM_ACT_NONE, ADNM_MFMA_BF16); else WG1(NBV, ADNM_ACT_NONE, ADNM_MFMA_F32); }                     \
  } while (0)
    if (p.nb == 2) WG(2);
    else WG(1);
#undef WG
#undef WG1
  }
  ADNM_CHECK_LAUNCH("conv3_wgrad");
  adnm_launch_fold("conv3_wgrad_fold", (const float*)ws, p.rows, (int)p.rowlen, {dw, (int)(N * 9 * K)}, {dbias, dbias ? (int)N : 0}, {nullptr, 0},
                   {nullptr, 0}, st);
  ADNM_CHECK_LAUNCH("conv3_wgrad_fold");
  return ADNM_OK;
}
