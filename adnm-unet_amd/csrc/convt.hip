// K9 — UpSample: nn.ConvTranspose2d(C, C, k=3, s=2, p=1, output_padding=1) (model_untils.py:120-158,490-520) as a GEMM over the
// INPUT pixels plus a 4-phase gather.  With s = 2 an output pixel (oy, ox) receives the taps whose parity matches:
//   out[b, oy, ox, co] = bias[co] + sum over (ky, kx) with (oy + 1 - ky), (ox + 1 - kx) even of  in[b, (oy+1-ky)/2, (ox+1-kx)/2, :] . W[:, co, ky, kx]
// i.e. 1, 2, 2 or 4 taps depending on the parities of (oy, ox).  The products of EVERY input pixel with EVERY tap are one dense GEMM
//   cols[M = B*H*W, 9*Cout] = X[M, Cin] . Wf[Cin, 9*Cout]            (the short-GEMM MFMA kernel, csrc/skgemm.hip, op NN)
// on the weight as it lies ((Cin, 3, 3, Cout) memory order = the flat trainer's layout; nn.ConvTranspose2d's own (Cin, Cout, 3, 3) order is
// read through the column strides below), and this file holds the two data-movement kernels around it:
//   col2im:  out  <- the per-phase gather of cols (+ bias)                       forward
//   im2col:  dcols[m, tap, co] <- dout[b, 2iy-1+ky, 2ix-1+kx, co] (0 outside)    backward: dX = dcols . Wf^T (op NT), dWf = X^T . dcols (op TN)
// Both are pure streams (4 floats per lane along co when the columns are (tap, co)-ordered).
#include "adnm_common.h"

namespace {
constexpr int kBlock = 256;

// column of (tap, co) in a cols row: tap*ct + co*cc
template <bool VEC>
__global__ __launch_bounds__(kBlock) void convt_col2im_kernel(const float* __restrict__ cols, int64_t ldc, int ct, int cc, const float* __restrict__ bias,
                                                              float* __restrict__ out, int64_t ldo, int B, int H, int W, int C) {
  const int cq = VEC ? C / 4 : C;
  const int64_t total = (int64_t)B * (2 * H) * (2 * W) * cq;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cq) * (VEC ? 4 : 1);
  int64_t p = i / cq;
  const int ox = (int)(p % (2 * W));
  p /= 2 * W;
  const int oy = (int)(p % (2 * H)), b = (int)(p / (2 * H));
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int e = 0; e < (VEC ? 4 : 1); ++e) acc[e] = bias[c + e];
  }
  // taps of matching parity: ky in {1} (oy even) or {0, 2} (oy odd); the same in x
  for (int ky = (oy & 1) ? 0 : 1; ky < 3; ky += 2) {
    const int iy2 = oy + 1 - ky;
    if (iy2 < 0 || iy2 >= 2 * H) continue;
    for (int kx = (ox & 1) ? 0 : 1; kx < 3; kx += 2) {
      const int ix2 = ox + 1 - kx;
      if (ix2 < 0 || ix2 >= 2 * W) continue;
      const int64_t m = ((int64_t)b * H + (iy2 >> 1)) * W + (ix2 >> 1);
      const float* src = cols + m * ldc + (int64_t)(ky * 3 + kx) * ct + (int64_t)c * cc;
      if (VEC) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
      } else {
        acc[0] += *src;
      }
    }
  }
  float* dst = out + (((int64_t)b * 2 * H + oy) * 2 * W + ox) * ldo + c;
  if (VEC) *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  else *dst = acc[0];
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void convt_im2col_kernel(const float* __restrict__ dout, int64_t lddo, float* __restrict__ dcols, int64_t ldc, int ct,
                                                              int cc, int B, int H, int W, int C) {
  const int cq = VEC ? C / 4 : C;
  const int64_t total = (int64_t)B * H * W * 9 * cq;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cq) * (VEC ? 4 : 1);
  int64_t r = i / cq;
  const int tap = (int)(r % 9);
  const int64_t m = r / 9;
  const int ix = (int)(m % W), iy = (int)((m / W) % H), b = (int)(m / ((int64_t)W * H));
  const int oy = 2 * iy - 1 + tap / 3, ox = 2 * ix - 1 + tap % 3;
  const bool ok = oy >= 0 && oy < 2 * H && ox >= 0 && ox < 2 * W;
  float* dst = dcols + m * ldc + (int64_t)tap * ct + (int64_t)c * cc;
  const float* src = dout + (((int64_t)b * 2 * H + oy) * 2 * W + ox) * lddo + c;
  if (VEC) *reinterpret_cast<float4*>(dst) = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
  else *dst = ok ? *src : 0.f;
}

int check(const char* who, const void* a, const void* b, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ct, int64_t cc) {
  ADNM_REQUIRE(a && b, "%s: null pointer", who);
  ADNM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && B * H * W * 36 * C < (1ll << 40), "%s: bad shape", who);
  ADNM_REQUIRE((ct == C && cc == 1) || (ct == 1 && cc == 9), "%s: column order must be (tap, co) [ct=C, cc=1] or (co, tap) [ct=1, cc=9]", who);
  return ADNM_OK;
}
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
}  // namespace

extern "C" int adnm_convt_col2im(const float* cols, int64_t ldc, int64_t col_tap_stride, int64_t col_c_stride, const float* bias, float* out, int64_t ldo,
                                 int64_t B, int64_t H, int64_t W, int64_t C, adnm_stream_t stream) {
  if (int rc = check("convt_col2im", cols, out, B, H, W, C, col_tap_stride, col_c_stride)) return rc;
  ADNM_REQUIRE(ldc >= 9 * C && ldo >= C, "convt_col2im: row strides smaller than the rows");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = col_c_stride == 1 && C % 4 == 0 && ldc % 4 == 0 && ldo % 4 == 0 && al16(cols) && al16(out) && (!bias || al16(bias));
  const int64_t total = B * 2 * H * 2 * W * (vec ? C / 4 : C);
  ADNM_PROF("convt_col2im", st, 4.0 * B * H * W * C * (9.0 + 4.0));
  if (vec) convt_col2im_kernel<true><<<(unsigned)adnm_cdiv(total, kBlock), kBlock, 0, st>>>(cols, ldc, (int)col_tap_stride, 1, bias, out, ldo, (int)B, (int)H, (int)W, (int)C);
  else convt_col2im_kernel<false><<<(unsigned)adnm_cdiv(total, kBlock), kBlock, 0, st>>>(cols, ldc, (int)col_tap_stride, (int)col_c_stride, bias, out, ldo, (int)B, (int)H, (int)W, (int)C);
  ADNM_CHECK_LAUNCH("convt_col2im");
  return ADNM_OK;
}

extern "C" int adnm_convt_im2col(const float* dout, int64_t lddo, float* dcols, int64_t ldc, int64_t col_tap_stride, int64_t col_c_stride, int64_t B,
                                 int64_t H, int64_t W, int64_t C, adnm_stream_t stream) {
  if (int rc = check("convt_im2col", dout, dcols, B, H, W, C, col_tap_stride, col_c_stride)) return rc;
  ADNM_REQUIRE(ldc >= 9 * C && lddo >= C, "convt_im2col: row strides smaller than the rows");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = col_c_stride == 1 && C % 4 == 0 && ldc % 4 == 0 && lddo % 4 == 0 && al16(dcols) && al16(dout);
  const int64_t total = B * H * W * 9 * (vec ? C / 4 : C);
  ADNM_PROF("convt_im2col", st, 4.0 * B * H * W * C * (9.0 + 4.0));
  if (vec) convt_im2col_kernel<true><<<(unsigned)adnm_cdiv(total, kBlock), kBlock, 0, st>>>(dout, lddo, dcols, ldc, (int)col_tap_stride, 1, (int)B, (int)H, (int)W, (int)C);
  else convt_im2col_kernel<false><<<(unsigned)adnm_cdiv(total, kBlock), kBlock, 0, st>>>(dout, lddo, dcols, ldc, (int)col_tap_stride, (int)col_c_stride, (int)B, (int)H, (int)W, (int)C);
  ADNM_CHECK_LAUNCH("convt_im2col");
  return ADNM_OK;
}
