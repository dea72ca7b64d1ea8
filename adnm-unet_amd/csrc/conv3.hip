// K5 — the dense 3x3 'same' convolutions of the U-Net conv stack as NHWC implicit GEMMs on the matrix cores.
// Reference call sites: PatchEmbed.conv2 5->32 (model_untils.py:259-273), WTLayer.conv 32->64 / 64->128 / 256->64 / 128->32 /
// 64->32 (:376-387), OutProj.conv[0] 32->64 and conv2 20->20 (:818-849) — nn.Conv2d(k=3, s=1, p=1) [+ bias] [+ GELU].
//
//   forward   out[p, n]  = act( sum_{tap, k} in[p + tap, k] * W[n, tap, k] + bias[n] )        p = pixel, n = Cout, k = Cin
//   dgrad     din[p, k]  = sum_{tap, n} dpre[p - tap, n] * W[n, tap, k]                       dpre = dout * act'(pre)
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

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kBlock = 256, kWaves = 4;
constexpr int CK = 16, CKP = CK + 4;   // channels per staged chunk, LDS pitch of a pixel
constexpr int MB = 2;                  // pixel blocks per wave (forward / dgrad)
constexpr int kMaxNB = 4;              // output-channel blocks per workgroup (64 channels)

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
  int act;                            // forward: epilogue activation; dgrad: the activation whose derivative multiplies `in`
  const float* w; int64_t sn, st, sk; int flip;   // W(n, tap, k) = w[n*sn + (flip ? 8-tap : tap)*st + k*sk]
  const float* bias;
  float* out; int64_t ldo;            // act(acc + bias)
  float* pre; int64_t ldpre;          // acc + bias (saved for backward), or NULL
  float* part;                        // nsplit > 1: partials [z][B*H*W][N]
  int B, H, W, K, N, nsplit, chunks_per_split;
  int TW, RB, TH;
};

template <int ACT>
__device__ __forceinline__ float stage_val(float v, float p) { return ACT == ADNM_ACT_NONE ? v : v * act_grad<ACT>(p); }

// ---- stage the (TH+2) x (TW+2) input tile of channels [c0, c0+16) of the tall image into LDS (zeros outside)
template <int ACT>
__device__ __forceinline__ void stage_tile(const ConvArgs& a, float* sIn, int v0, int x0, int c0) {
  const int TWp = a.TW + 2, PT = (a.TH + 2) * TWp;
  const bool quad_ok = (a.K & 3) == 0;
  for (int it = threadIdx.x; it < PT * 4; it += kBlock) {
    const int pix = it >> 2, q = it & 3, r = pix / TWp, c = pix - r * TWp;
    const int v = v0 - 1 + r, x = x0 - 1 + c, ch = c0 + 4 * q;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v >= 0 && v < a.VR_dummy_guard_never_used()) {}
    (void)val;
  }
}

}  // namespace
