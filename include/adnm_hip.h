/*
 * adnm_hip.h — C-ABI of libadnm_hip.so, the MI355X (gfx950) kernels behind the ADNM-UNet
 * training hot path.
 *
 * The reference (kanyu369/ADNM-UNet) has no FFI of its own: its hot ops are torch / mamba_ssm
 * calls inside nn.Module.forward().  Each entry point below names the reference call site it
 * replaces (file:line under the reference tree).  The Python modules in
 * adnm-unet_amd/models/ keep the reference's nn.Module surface and call these through
 * ctypes (adnm-unet_amd/adnm_hip/lib.py); INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch allocator); no compute entry point
 *    allocates, frees, synchronises or touches the host copy of anything (the two setup-time helpers
 *    adnm_uncached_alloc / adnm_uncached_free exist so that the CALLER can own the uncached workspace
 *    of split GEMM launches: see adnm_skgemm);
 *  - kernels are enqueued on `stream` (torch.cuda.current_stream().cuda_stream) and are
 *    hipGraph-capturable; workspace is passed in, its size comes from the *_ws_bytes helper;
 *  - token tensors are channels-last ("BLD" = (B, H*W, C) = NHWC), the layout the reference
 *    keeps between stages (ADNMUNet.py:119, model_untils.py:21-27);
 *  - `dtype` selects the storage type of activations: ADNM_F32 (0) or ADNM_BF16 (1);
 *    parameters, statistics, reductions and workspaces are always fp32;
 *  - return value 0 = launched; negative = rejected (nothing launched), text via
 *    adnm_last_error() (thread-local).  The compute entry points keep no global mutable state: they are re-entrant
 *    per thread / stream / device (the reference's nn.DataParallel calls them from one thread per device).  The only
 *    process-wide state is the opt-in profiler's record list (adnm_prof_*, mutex-guarded), per-device "dynamic LDS
 *    limit raised" flags (atomics; setting one twice is harmless) and the thread-local error string and fold- / leaf-queue
 *    bindings.  There are no library-owned device buffers.
 */
#ifndef ADNM_HIP_H
#define ADNM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* adnm_stream_t; /* hipStream_t */

enum { ADNM_F32 = 0, ADNM_BF16 = 1 };
enum { ADNM_ACT_NONE = 0, ADNM_ACT_SILU = 1, ADNM_ACT_GELU = 2 };
enum { ADNM_OK = 0, ADNM_EINVAL = -1, ADNM_ELAUNCH = -2, ADNM_EWORKSPACE = -3 };
/* `prec` of the GEMM-shaped entry points (tsgemm, skgemm, conv3) — the precision ladder, storage stays fp32:
 *   ADNM_MFMA_F32   exact fp32 MFMA (v_mfma_f32_16x16x4_f32: the parity path);
 *   ADNM_MFMA_BF16  operands rounded to bf16 on the way into v_mfma_f32_16x16x32_bf16, fp32 accumulation (BASELINE configs 2-4);
 *   ADNM_MFMA_FP8   BASELINE config 5: per-tensor scaled OCP fp8 operands into v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulation:
 *                   first operand (the activation rows) e4m3, second (the weight) e4m3;
 *   ADNM_MFMA_FP8_GRAD  the same with the first operand in e5m2 (a gradient: output gradients into the input-gradient GEMMs).
 * The fp8 modes need a quantisation record `q` (device memory, 8 floats): {scale_a, scale_b, amax_a, amax_b, fmax_a, fmax_b, record, -}.
 * An operand value v enters the MFMA as fp8(clamp(v * scale)), the accumulator is multiplied by 1 / (scale_a * scale_b) before bias /
 * activation.  With q != NULL and record != 0 (any prec) the launch also collects amax_a / amax_b = max |v| of the operands it read
 * (atomic max, order-independent); adnm_quant_update turns them into the next scales (delayed per-tensor scaling: a tensor's scale
 * comes from an earlier step's amax, so quantisation costs no extra pass over the activations).  q == NULL: scales 1, nothing recorded. */
enum { ADNM_MFMA_F32 = 0, ADNM_MFMA_BF16 = 1, ADNM_MFMA_FP8 = 2, ADNM_MFMA_FP8_GRAD = 3 };
/* storage of a GEMM's weight operand (b_dtype of adnm_skgemm): fp32 master values or the narrow shadow kept beside them */
enum { ADNM_B_F32 = 0, ADNM_B_BF16 = 1, ADNM_B_FP8 = 2 };

/* One pass over a table of `n` quantisation records (n * 8 floats), once per training step:
 *   state = {step counter, period}: the counter advances; records collect amax (record = 1) during the steps where counter % period == 0;
 *   after such a step: scale_x = fmax_x / (amax_x * headroom) for every operand with amax_x > 0 (else unchanged), amax_x = 0.
 * fmax_a / fmax_b are set by the caller when a record is created (448 for e4m3 operands, 57344 for e5m2).  headroom >= 1 leaves room
 * for the tensor to grow between calibrations (2 = one binade). */
int adnm_quant_update(float* table, int64_t n, float* state, float headroom, adnm_stream_t stream);

const char* adnm_last_error(void);
int adnm_abi_version(void);

/* Opt-in measurement aid (off by default): when enabled every kernel launch of the library is bracketed by
 * hipEventRecord on its own stream.  adnm_prof_collect synchronises those events and writes one line per kernel
 * name "name\tlaunches\ttotal_ms\talgorithmic_bytes\n" into buf (returns the full length).  bench.py uses it
 * for the roofline figures; nothing else calls it. */
int adnm_prof_enable(int on);
int64_t adnm_prof_collect(char* buf, int64_t buflen);

/* Deferred second-stage folds.  Every cross-workgroup reduction of the library is "fp32 partials + a deterministic fold launch".
 * For parameter gradients (nothing reads them before clip_grad_norm_ / the optimiser, train.py:140-144) the caller may batch
 * those fold launches: create a queue, bind it on the calling thread around *_bwd entry points (their folds are then QUEUED, the
 * partial workspaces and destinations must stay alive), and flush it — one launch per 16 queued folds — before anything reads
 * the results.  Unbound (the default) every fold is launched at once.  The queue is a caller-owned host object; the binding is
 * thread-local (like adnm_last_error), so concurrent threads / devices do not see each other's queues.  Results are bitwise
 * the same either way. */
void* adnm_foldq_create(void);
int adnm_foldq_destroy(void* q);
int adnm_foldq_bind(void* q); /* NULL unbinds */
int64_t adnm_foldq_pending(void* q);
int adnm_foldq_flush(void* q, adnm_stream_t stream);
int adnm_foldq_clear(void* q); /* drop the queued folds without launching them (error path of the caller) */
/* The NEXT fold queued on this thread adds to its destination segment q (instead of overwriting it) for every bit q of mask: a parameter
 * used by two autograd nodes (Block's beta1 / beta2 feed both residual mixes, ADNMUNet.py:152,158) gets the second node's contribution
 * added by the fold itself — flushes launch the overwriting folds first — instead of by a separate autograd add per node.  Only
 * meaningful while a queue is bound (ignored otherwise: an immediate fold always overwrites). */
int adnm_foldq_accumulate_next(int mask);

/* Deferred leaf launches.  The weight-gradient GEMMs of a backward pass (ADNM_SKGEMM_TN) are leaves — nothing reads dW before
 * clip_grad_norm_ / the optimiser (train.py:140-144) — and individually small (70 launches of ~10 us at config 2).  While the calling
 * thread has bound BOTH a leaf queue and a fold queue, adnm_skgemm(TN) stores its prepared launch instead of making it (operands,
 * workspace and destination must stay alive); adnm_leafq_flush launches the stored problems grouped, 16 per launch, and must be called
 * BEFORE adnm_foldq_flush (the folds of split problems read the partials those launches write).  Results are bitwise the same either way. */
void* adnm_leafq_create(void);
int adnm_leafq_destroy(void* q);
int adnm_leafq_bind(void* q); /* NULL unbinds */
int64_t adnm_leafq_pending(void* q);
int adnm_leafq_flush(void* q, adnm_stream_t stream);
int adnm_leafq_clear(void* q);

/* ---------------------------------------------------------------- row norms (K2, K7)
 * y = scale * ( xhat * w + b ) + shift,  xhat = (x - mu) * rstd
 *   RMSNorm   (mamba_ssm RMSNorm bound at ADNMUNet.py:278, used ADNMUNet.py:149,155): subtract_mean=0, b=NULL
 *   LayerNorm (ADNssd.py:456, Vssd.py:280):                                           subtract_mean=1
 *   BiasFree_LayerNorm (model_untils.py:43-48):                                       subtract_mean=1, b=NULL
 * scale/shift are the scalar learnable affine of Block.forward (ADNMUNet.py:149,155) as
 * 1-element device tensors, or NULL (= 1 / 0).  x:(M,d) with row stride ldx, y row stride ldy
 * (elements).  mu,rstd: (M) fp32 statistics saved for backward (mu unused for RMS). d % 4 == 0. */
int adnm_rownorm_fwd(const void* x, int64_t ldx, const float* w, const float* b, const float* scale,
                     const float* shift, void* y, int64_t ldy, float* mu, float* rstd, int64_t M, int64_t d,
                     float eps, int subtract_mean, int dtype, adnm_stream_t stream);
/* dx:(M,d) stride lddx; dw,db:(d); dscale,dshift:(1) — any of db/dscale/dshift may be NULL.
 * dres (optional, (M,d) stride lddres): the gradient reaching x through the residual path of a pre-norm block
 * (ADNMUNet.py:152,158: x' = beta1*x + beta2*f(norm(x))) — added into dx in the same pass, so autograd's separate add disappears.
 * ws: fp32 workspace of adnm_rownorm_bwd_ws_bytes(M,d) bytes. Parameter grads are OVERWRITTEN. */
int64_t adnm_rownorm_bwd_ws_bytes(int64_t M, int64_t d);
int adnm_rownorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* w, const float* b,
                     const float* scale, const float* mu, const float* rstd, void* dx, int64_t lddx, float* dw,
                     float* db, float* dscale, float* dshift, const void* dres, int64_t lddres, void* ws, int64_t ws_bytes,
                     int64_t M, int64_t d, int subtract_mean, int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- SSD reduction form (K1)
 * non_casual_linear_attn (ADNssd.py:252-299, Vssd.py:161-208):
 *   dt = softplus(dt_raw + dt_bias)                      (ADNssd.py:318 fused here)
 *   KV[b,h,n,p] = sum_l Bm[b,l,g(h),n] * x[b,l,h,p] * dt[b,l,h] * exp(A_log[h])
 *   y[b,l,h,p]  = sum_n Cm[b,l,g(h),n] * KV[b,h,n,p] + D[h] * x[b,l,h,p],     g(h) = h % G
 * x:(B,L,H,P) row(token) stride ldx; Bm,Cm:(B,L,G*N) strides ldb,ldc; dt_raw:(B,L,H) stride lddt with
 * per-head element stride dt_hstride (2 for the even/odd head split of ADNssd.py:375-378);
 * dt_bias,A_log,D indexed [h*p_hstride] likewise.  y stride ldy.  kv:(B,H,N,P) fp32 (saved for backward).
 * (P,N) in {(4,8),(4,16),(8,8)}, G in {1,2,4}.  The three contractions run on v_mfma_f32_16x16x4_f32 (exact fp32).
 * Optional fused epilogue (yn != NULL; needs H*P == 64, i.e. the token row is one head block — the refiner mixers): the mixer's
 * LayerNorm(d_inner) of ADNssd.py:456, yn = (y - mean) * rstd * ln_w + ln_b over the H*P columns of the row, written with row
 * stride ldyn next to y itself; ln_mu, ln_rstd:(B*L) fp32 statistics saved for adnm_rownorm_bwd.  yn == NULL: the ln_* arguments
 * are ignored. */
int64_t adnm_ssd_ws_bytes(int64_t B, int64_t L, int64_t H, int64_t P, int64_t N, int64_t G);
int adnm_ssd_reduce_fwd(const void* x, int64_t ldx, const void* Bm, int64_t ldb, const void* Cm, int64_t ldc,
                        const void* dt_raw, int64_t lddt, int64_t dt_hstride, const float* dt_bias,
                        const float* A_log, const float* D, int64_t p_hstride, void* y, int64_t ldy, float* kv,
                        const float* ln_w, const float* ln_b, void* yn, int64_t ldyn, float* ln_mu, float* ln_rstd,
                        float ln_eps, void* ws, int64_t ws_bytes, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N,
                        int64_t G, int dtype, adnm_stream_t stream);
/* gradients: dx,dBm,dCm,ddt_raw share the strides of their primals (they may alias slices of one
 * wide gradient buffer); ddt_bias,dA_log,dD:(H) contiguous fp32, OVERWRITTEN. */
int adnm_ssd_reduce_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* Bm, int64_t ldb,
                        const void* Cm, int64_t ldc, const void* dt_raw, int64_t lddt, int64_t dt_hstride,
                        const float* dt_bias, const float* A_log, const float* D, int64_t p_hstride,
                        const float* kv, void* dx, int64_t lddx, void* dBm, int64_t lddb, void* dCm, int64_t lddc,
                        void* ddt_raw, int64_t ldddt, float* ddt_bias, float* dA_log, float* dD, void* ws,
                        int64_t ws_bytes, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N, int64_t G,
                        int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- SSD chunked scan (K1b) — PARITY UNPINNED
 * The `linear_attn_duality=False` branch (ADNssd.py:413-454, Vssd.py:245-275): mamba_ssm's
 * mamba_chunk_scan_combined(x, dt, A, B, C, chunk_size, D, z=None) — un-vendored Triton in the reference:
 *   d_t = exp(dt_t*A_h), S_t = d_t S_{t-1} + dt_t B_t (x) x_t, y_t = C_t.S_t + D_h x_t,  dt = softplus(dt_raw+dt_bias),
 *   A_h = -exp(A_log[h]); head h reads K/Q group h / (H/G).  reverse=1 scans the sequence backwards (the reference
 * flips the odd half, ADNssd.py:425-435).  x/y/dx: element (row, h, p) at ptr[row*ld + h*hstride + p];
 * dt_raw (row, h) at ptr[row*lddt + h*dt_hstride]; dt_bias/A_log/D at [h*p_hstride].  S_in: (B,H,ceil(L/chunk),P,N)
 * fp32 entering states, written by fwd and read by bwd.  P = 4, N in {8,16}, fp32 only. */
int64_t adnm_ssd_scan_ws_bytes(int64_t B, int64_t L, int64_t H, int64_t N, int64_t chunk, int backward);
int adnm_ssd_scan_fwd(const void* x, int64_t ldx, int64_t x_hstride, const void* Bm, int64_t ldb, const void* Cm,
                      int64_t ldc, const void* dt_raw, int64_t lddt, int64_t dt_hstride, const float* dt_bias,
                      const float* A_log, const float* D, int64_t p_hstride, void* y, int64_t ldy, int64_t y_hstride,
                      float* S_in, void* ws, int64_t ws_bytes, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N,
                      int64_t G, int64_t chunk, int reverse, int dtype, adnm_stream_t stream);
int adnm_ssd_scan_bwd(const void* dy, int64_t lddy, int64_t dy_hstride, const void* x, int64_t ldx, int64_t x_hstride,
                      const void* Bm, int64_t ldb, const void* Cm, int64_t ldc, const void* dt_raw, int64_t lddt,
                      int64_t dt_hstride, const float* dt_bias, const float* A_log, const float* D, int64_t p_hstride,
                      const float* S_in, void* dx, int64_t lddx, int64_t dx_hstride, void* dBm, int64_t lddb, void* dCm,
                      int64_t lddc, void* ddt_raw, int64_t ldddt, int64_t ddt_hstride, float* ddt_bias, float* dA_log,
                      float* dD, void* ws, int64_t ws_bytes, int64_t B, int64_t L, int64_t H, int64_t P, int64_t N,
                      int64_t G, int64_t chunk, int reverse, int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- depthwise conv, NHWC (K4, part of K3)
 * y[b,h,w,c] = act( sum_{i,j} wgt[c,i,j] * x[b,h+i-KH/2,w+j-KW/2,c] + bias[c] ) (+ addend[b,h,w,c])
 * replaces the depthwise nn.Conv2d calls at ADNssd.py:334,343-346,389, Vssd.py:233,
 * model_untils.py:180-188 (FeedForward.dwconv), :203-211 (ConvFFD.dw_conv), WTConv2d.py:81,86 —
 * applied directly on the token layout, so the reference's BLD<->BCHW permute().contiguous()
 * copies disappear.  x pixel stride ldx, y pixel stride ldy, addend pixel stride ldadd (elements);
 * wgt, fp32: wlayout 0 = TAP-MAJOR (KH,KW,C) (one float4 per tap and channel quad: what the parameter-prep kernels
 * emit), wlayout 1 = nn.Conv2d's own (C,1,KH,KW) (no transposes around the call); KH == KW in {3,5}; C % 4 == 0
 * (the 5-channel input stage is zero-padded to 8 channels by the caller). */
int adnm_dwconv_fwd(const void* x, int64_t ldx, const float* wgt, const float* bias, const void* addend,
                    int64_t ldadd, void* y, int64_t ldy, int64_t B, int64_t H, int64_t W, int64_t C, int KH,
                    int KW, int act, int wlayout, int dtype, adnm_stream_t stream);
/* dpre:(B,H,W,C) contiguous scratch in activation dtype (ignored when act==NONE).
 * dwgt: same layout as wgt (wlayout), dbias:(C) or NULL: OVERWRITTEN.  dwgt == NULL skips the weight-gradient pass
 * (constant taps, e.g. the average pools of EncoderToDecoder expressed as a depthwise conv). */
int64_t adnm_dwconv_bwd_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t C, int KH, int KW);
int adnm_dwconv_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* wgt,
                    const float* bias, void* dpre, void* dx, int64_t lddx, float* dwgt, float* dbias, void* ws,
                    int64_t ws_bytes, int64_t B, int64_t H, int64_t W, int64_t C, int KH, int KW, int act,
                    int wlayout, int dtype, adnm_stream_t stream);
/* the weight / bias gradient of adnm_dwconv_bwd alone (g = its dpre, or dy when no activation is fused), e.g. on another stream than the
 * input-gradient chain; adnm_dwconv_bwd with dwgt = NULL then skips it.  Workspace: adnm_dwconv_bwd_ws_bytes. */
int adnm_dwconv_wgrad(const void* g, int64_t ldg, const void* x, int64_t ldx, float* dwgt, float* dbias, void* ws, int64_t ws_bytes,
                      int64_t B, int64_t H, int64_t W, int64_t C, int KH, int KW, int wlayout, int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- Haar butterflies, NHWC (K3)
 * wavelet_transform / inverse_wavelet_transform with db1 (WTConv2d.py:31-51): per 2x2 block
 * [[a,b],[c,d]]: LL=(a+b+c+d)/2, k1=(a+b-c-d)/2, k2=(a-b+c-d)/2, k3=(a-b-c+d)/2; output channel c*4+k.
 * dwt: x:(B,H,W,C) with pixel stride ldx and channel stride cx (4 when x is the LL band of a previous
 * level) -> y:(B,ceil(H/2),ceil(W/2),4C) contiguous; odd H/W are zero-padded (WTConv2d.py:114-116).
 * idwt: s:(B,h,w,4C) contiguous, optional ll_add:(B,h,w,C) contiguous added to the LL band
 * (WTConv2d.py:136) -> y:(B,H,W,C) contiguous with H in {2h-1,2h}, W in {2w-1,2w} (crop, :141).
 * The butterfly is its own transpose, so dwt's backward is idwt and vice versa. */
int adnm_haar_dwt(const void* x, int64_t ldx, int64_t cx, void* y, int64_t B, int64_t H, int64_t W, int64_t C,
                  int dtype, adnm_stream_t stream);
/* One ANALYSIS LEVEL of WTConv2d fused (WTConv2d.py:111-124): sub = DWT(x) and tag = depthwise KxK 'same' conv of sub (taps tap-major
 * (K*K, 4C) fp32, wavelet_scale folded in) in ONE launch (csrc/wtlevel.hip: the sub-band tile + halo lives in LDS).  x: (B,H,W) pixel rows
 * of stride ldx, channel c at column c*cx (cx = 4: the LL band of the previous level's sub-band tensor); sub, tag: (B,ceil(H/2),ceil(W/2),4C)
 * contiguous fp32, OVERWRITTEN.  flip != 0: flipped taps — one level of the backward pass (dm = DWT(d r), d sub = conv^T(dm)).  K in {3, 5}. */
int adnm_wt_level(const float* x, int64_t ldx, int64_t cx, const float* taps, float* sub, float* tag, int64_t B, int64_t H, int64_t W,
                  int64_t C, int K, int flip, adnm_stream_t stream);
/* y_add1 / y_add2 (optional, (B,H,W,C) like y): added to the result in the same pass — WTConv2d's backward sums its input-gradient
 * paths (pyramid + base conv + the input's other consumer) there instead of in separate adds.
 * up1 / up2 (optional): the sub-band tensors of the next one / two COARSER levels ((B, ceil(h/2), ceil(w/2), 4C) with h, w the sub-band
 * size of the level below): the synthesis cascade (WTConv2d.py:128-141) in ONE launch — the LL addend of `s` is derived from them instead of
 * being read; ll_add then belongs to the coarsest level given.  Bitwise equal to the chain of single-level calls. */
int adnm_haar_idwt(const void* s, const void* up1, const void* up2, const void* ll_add, const void* y_add1, const void* y_add2, void* y,
                   int64_t B, int64_t H, int64_t W, int64_t C, int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- max pooling, NHWC (K11)
 * nn.MaxPool2d on tokens: stride == kernel in [2,4] (DownSample, model_untils.py:472-487; floor mode) or stride 1 with
 * kernel 1x3 / 3x1 / 3x3 and -inf 'same' padding (EncoderToDecoder, model_untils.py:690-719).  x:(B,H,W,C),
 * y:(B,Ho,Wo,C) contiguous.  bwd recomputes the first arg-max of every window (ATen's tie rule): no index tensor; dx_add (optional, like
 * x): the gradient of x's other consumer, added in the same pass (an encoder stage's output is pooled AND kept as a skip tensor). */
int adnm_maxpool_fwd(const void* x, void* y, int64_t B, int64_t H, int64_t W, int64_t C, int kh, int kw, int stride,
                     int dtype, adnm_stream_t stream);
int adnm_maxpool_bwd(const void* dy, const void* x, const void* dx_add, void* dx, int64_t B, int64_t H, int64_t W, int64_t C, int kh,
                     int kw, int stride, int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- InstanceNorm2d, NHWC (K8)
 * y = act( scale * (x - mean_{hw}) * rsqrt(var_{hw} + eps) + shift ), per (b,c) plane, no affine,
 * external scalar scale/shift (model_untils.py:90,113,155; nn.InstanceNorm2d at :284,371,741,814).
 * x,y:(B,HW,C) contiguous; mu,rstd:(B,C) fp32 saved for backward. act in {NONE, GELU}. */
int64_t adnm_instnorm_ws_bytes(int64_t B, int64_t HW, int64_t C);
int adnm_instnorm_fwd(const void* x, const float* scale, const float* shift, void* y, float* mu, float* rstd,
                      void* ws, int64_t ws_bytes, int64_t B, int64_t HW, int64_t C, float eps, int act, int dtype,
                      adnm_stream_t stream);
int adnm_instnorm_bwd(const void* dy, const void* x, const float* scale, const float* shift, const float* mu,
                      const float* rstd, void* dx, float* dscale, float* dshift, void* ws, int64_t ws_bytes,
                      int64_t B, int64_t HW, int64_t C, int act, int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- gated FFN activation (part of K6's epilogue)
 * FeedForward.forward (model_untils.py:194-195): h:(M,2F) -> y[m,f] = gelu(h[m,f]) * sigmoid(h[m,F+f]).
 * bwd: dh:(M,2F) from dy:(M,F).  F % 4 == 0; row strides in elements. */
int adnm_gate_fwd(const void* h, int64_t ldh, void* y, int64_t ldy, int64_t M, int64_t F, int dtype,
                  adnm_stream_t stream);
int adnm_gate_bwd(const void* dy, int64_t lddy, const void* h, int64_t ldh, void* dh, int64_t lddh, int64_t M,
                  int64_t F, int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- IntensityGate (K11)
 * y = silu(enhance * (x - threshold)) with both scalars learnable (model_untils.py:523-532; 4 per EncoderToDecoder
 * + the bridge's gates).  n contiguous elements, n % 4 == 0.  bwd also gives d enhance, d threshold (OVERWRITTEN). */
int adnm_igate_fwd(const void* x, const float* enhance, const float* threshold, void* y, int64_t n, int dtype,
                   adnm_stream_t stream);
int64_t adnm_igate_bwd_ws_bytes(int64_t n);
int adnm_igate_bwd(const void* dy, const void* x, const float* enhance, const float* threshold, void* dx, float* denhance,
                   float* dthreshold, void* ws, int64_t ws_bytes, int64_t n, int dtype, adnm_stream_t stream);
/* EncoderToDecoder's entry, fused (model_untils.py:761-763: x = self.act(x + self.gama * res), act = IntensityGate):
 *   y[b,l,c] = silu(enhance * (x[b,l,c] + gama * res[b,c] - threshold)),   x, y:(B,L,C) fp32 contiguous, res:(B,C) — the
 * channel-attention gate of Channel_Att_Bridge, one value per (sample, channel), which the reference expands over the tokens
 * (model_untils.py:604-613); per_token != 0: res is the expanded (B,L,C) tensor itself (the reference's own call form).
 * gama / enhance / threshold: 1-element fp32.  C % 4 == 0.
 * Backward: dx:(B,L,C), dres: the shape of res (complete when the launch ends), dgama/denhance/dthreshold:(1) through per-workgroup partials
 * + one fold (deferrable: parameter gradients).  All OVERWRITTEN. */
int adnm_igate_res_fwd(const float* x, const float* res, int per_token, const float* gama, const float* enhance,
                       const float* threshold, float* y, int64_t B, int64_t L, int64_t C, adnm_stream_t stream);
int64_t adnm_igate_res_bwd_ws_bytes(int64_t B, int64_t L, int64_t C);
int adnm_igate_res_bwd(const float* dy, const float* x, const float* res, int per_token, const float* gama, const float* enhance,
                       const float* threshold, float* dx, float* dres, float* dgama, float* denhance, float* dthreshold, void* ws,
                       int64_t ws_bytes, int64_t B, int64_t L, int64_t C, adnm_stream_t stream);


/* ---------------------------------------------------------------- EncoderToDecoder's pooled gating core (K12)
 * model_untils.py:767-787 of the reference, x = the block's normalised tokens (B, H, W, C) fp32 contiguous, C % 4 == 0:
 *   p_k = MaxPool_k(x) + AvgPool_k(x)      k = 0: (3,1) window, 1: (1,3), 2: (3,3); stride 1, count_include_pad
 *   c_k = conv_k(p_k)                      conv13pool (1,3) / conv31pool (3,1) / conv33pool (3,3), groups = C/4, with bias
 *   y_k = IntensityGate_k(ffd_k(x * GELU(c_k)))    ffd13 + act_func13 for k = 0 AND 1 (as the reference), ffd33 + act_func33 for k = 2
 *   out = gamma * (alpha1 y_0 + alpha2 y_1 + alpha3 y_2)
 * params: HOST array of 18 device pointers
 *   [0..5]  conv13pool.weight (C,4,1,3), .bias, conv31pool.weight (C,4,3,1), .bias, conv33pool.weight (C,4,3,3), .bias
 *   [6..9]  ffd13.weight (C), ffd13.bias, ffd33.weight, ffd33.bias          (1x1 depthwise = per-channel affine)
 *   [10..13] act_func13.enhance, .threshold, act_func33.enhance, .threshold  (scalars)
 *   [14..17] alpha1, alpha2, alpha3 (scalars), gamma (C)
 * pooled, conv: (3, B, H, W, C) each, written by fwd and handed back to bwd.
 * dparams (OVERWRITTEN), adnm_skipgate_grad_floats(C) floats:
 *   [d w0 12C | d w1 12C | d w2 36C | d gamma | d ffd13.w | d ffd13.b | d ffd33.w | d ffd33.b | d b0 | d b1 | d b2 (C each) |
 *    d alpha1..3, d enh13, d thr13, d enh33, d thr33, pad] */
int adnm_skipgate_fwd(const float* x, const float* const* params, float* pooled, float* conv, float* out, int64_t B, int64_t H, int64_t W,
                      int64_t C, adnm_stream_t stream);
int64_t adnm_skipgate_grad_floats(int64_t C);
int64_t adnm_skipgate_bwd_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t C);
int adnm_skipgate_bwd(const float* dout, const float* x, const float* const* params, const float* pooled, const float* conv, float* dx,
                      float* dparams, void* ws, int64_t ws_bytes, int64_t B, int64_t H, int64_t W, int64_t C, adnm_stream_t stream);

/* ---------------------------------------------------------------- soft-max attention with 4-wide heads (K16)
 * StandardAttention.forward (models/ADNssd.py:38-46 of the reference) between to_qkv and to_out:
 *   out[b,l,h*4+d] = sum_j softmax_j(scale * q[b,h,l,:] . k[b,h,j,:]) v[b,h,j,d],   heads = inner/4, L <= 2048
 * qkv: (B, L, 3*inner) fp32 contiguous = to_qkv's output as it is ([q | k | v], head h at columns h*4..h*4+3 of each third);
 * out: (B, L, inner); lse: (B, heads, L) log-sum-exp of the scaled scores, saved for backward.  No (L, L) tensor is
 * materialised.  bwd: dqkv (B, L, 3*inner) OVERWRITTEN. */
int adnm_attn4_fwd(const float* qkv, float* out, float* lse, int64_t B, int64_t L, int64_t heads, float scale, adnm_stream_t stream);
int adnm_attn4_bwd(const float* dout, const float* qkv, const float* out, const float* lse, float* dqkv, int64_t B, int64_t L, int64_t heads,
                   float scale, adnm_stream_t stream);

/* ---------------------------------------------------------------- Channel_Att_Bridge's Conv1d (K15; the pooling itself: adnm_bridge_pool_*)
 * model_untils.py:592 of the reference.  conv1d3: nn.Conv1d(1,1,3,padding=1) over the concatenated channel axis
 * of (B,n) (get_all_att); bwd writes dx and dwb = [dw0, dw1, dw2, dbias] (OVERWRITTEN). */
/* Elementwise product y = a * b of two (M, C) fp32 token matrices with row strides lda / ldb (y, da, db contiguous): VSSD's output gate
 * LayerNorm(y) * z (Vssd.py:280-281).  4 | C. */
int adnm_emul_fwd(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t M, int64_t C, adnm_stream_t stream);
int adnm_emul_bwd(const float* dy, const float* a, int64_t lda, const float* b, int64_t ldb, float* da, float* db, int64_t M, int64_t C,
                  adnm_stream_t stream);
/* Channel pad / crop of a token matrix (row stride ldx): y[m, c] = x[m, c] for c < min(Cin, Cout), 0 for Cin <= c < Cout; y:(M, Cout)
 * contiguous.  The 5-frame input stage (PatchEmbed.conv1: WTConv2d on 5 channels, model_untils.py:259) runs on 8 channels; pad and crop
 * are each other's backward. */
int adnm_chancopy(const float* x, int64_t ldx, int64_t Cin, float* y, int64_t Cout, int64_t M, adnm_stream_t stream);
int adnm_conv1d3_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t n, adnm_stream_t stream);
int adnm_conv1d3_bwd(const float* dy, const float* x, const float* w, float* dx, float* dwb, int64_t B, int64_t n, adnm_stream_t stream);
/* The heads of Channel_Att_Bridge, grouped (model_untils.py:744-750 the seven nn.Linear(sum C, C_i), :594-613 sigmoid1 = IntensityGate):
 *   z_i[b, :] = att[b, :] . W_i^T + bias_i,   gate_i = silu(enhance * (z_i - threshold)),   att:(B, S) fp32 contiguous, W_i:(C_i, S).
 * W / bias / z / y / dy / dW / dbias: HOST arrays of nheads device pointers; C: HOST array of the head widths.  1 <= B <= 8, 4 | S <= 3072.
 * One forward launch for all heads (a wave per output feature); backward = one launch (weight-gradient rows written as they are formed,
 * dbias, per-workgroup partials of datt / denhance / dthreshold) + one fold that is never deferred (datt is read next).
 * All outputs OVERWRITTEN; dbias may be NULL (or hold NULL entries). */
int adnm_bridge_heads_fwd(const float* att, const float* const* W, const float* const* bias, const int64_t* C, int nheads,
                          const float* enhance, const float* threshold, float* const* z, float* const* y, int64_t B, int64_t S,
                          adnm_stream_t stream);
int64_t adnm_bridge_heads_bwd_ws_bytes(int64_t total, int64_t B, int64_t S);
int adnm_bridge_heads_bwd(const float* att, const float* const* W, const int64_t* C, int nheads, const float* enhance,
                          const float* threshold, const float* const* z, const float* const* dy, float* datt, float* const* dW,
                          float* const* dbias, float* denhance, float* dthreshold, void* ws, int64_t ws_bytes, int64_t B, int64_t S,
                          adnm_stream_t stream);

/* The bridge's global average pools, grouped (Channel_Att_Bridge.forward, model_untils.py:570-590: AdaptiveAvgPool2d(1) of each skip, then
 * torch.cat along the channels): ONE launch + one fold for all n <= 8 skips.  x[k]: (B, L[k], C[k]) contiguous fp32 tokens, 4 | C[k];
 * att: (B, S = sum C[k]) OVERWRITTEN, skip k at columns [sum_{j<k} C[j], ...).  bwd: dx[k] = dxa[k] (or 0 when NULL) + datt[:, cols of k] / L[k]
 * broadcast over the tokens, one launch for all skips (dx[k] NULL skips skip k). */
int64_t adnm_bridge_pool_ws_bytes(int64_t B, int64_t S);
int adnm_bridge_pool_fwd(int64_t n, const float* const* x, const int64_t* L, const int64_t* C, float* att, void* ws, int64_t ws_bytes, int64_t B,
                         adnm_stream_t stream);
int adnm_bridge_pool_bwd(int64_t n, const float* const* dxa, const float* datt, const int64_t* L, const int64_t* C, float* const* dx, int64_t B,
                         adnm_stream_t stream);

/* out[c] = sum_r x[r*n + c] for a contiguous (rows, n) fp32 matrix — nn.Linear's bias gradient (autograd's sum over the token
 * rows) when the weight gradient is computed elsewhere (the transposed conv's).  Deterministic (fixed tree), OVERWRITES out; matrices
 * of >= 2048 rows are summed in two stages through `ws` (adnm_colsum_ws_bytes), which must stay alive until a bound fold queue is
 * flushed. */
int64_t adnm_colsum_ws_bytes(int64_t rows, int64_t n);
int adnm_colsum(const float* x, float* out, int64_t rows, int64_t n, void* ws, int64_t ws_bytes, adnm_stream_t stream);

/* ---------------------------------------------------------------- short GEMMs of the deep stages (K6b, MFMA)
 * nn.Linear with M <= 2^20 token rows (M x features < 2^31) and up to 16384 features: Mamba2.in_proj/out_proj (ADNssd.py:309,461),
 * FeedForward.project_in/out (model_untils.py:193,196), Mlp (:64,67), ConvFFD (:217,221), Block.out_proj (ADNMUNet.py:163),
 * StandardAttention.to_qkv/to_out (ADNssd.py:33-34), Channel_Att_Bridge.att* (model_untils.py:744-750).  fp32, row-major:
 *   ADNM_SKGEMM_NT: c[M,N] = a[M,K] . b[N,K]^T (+ bias[N])           forward          K % 4 == 0
 *   ADNM_SKGEMM_NN: c[M,K] = a[M,N] . b[N,K]                         input gradient   N % 4 == 0, K % 4 == 0
 *   ADNM_SKGEMM_TN: c[N,K] = a[M,N]^T . b[M,K]; dbias[N] = sum_m a   weight gradient  N % 4 == 0, K % 4 == 0
 * lda / ldb / ldc: row strides in elements (multiples of 4); bias only with NT, dbias only with TN; c / dbias OVERWRITTEN.
 * prec / q: the precision ladder and the call site's quantisation record (see ADNM_MFMA_*); a = the "first operand", b = the weight.
 * TN (the weight gradient) runs on bf16 operands in the fp8 modes.
 * b_dtype (NT / NN): storage of the weight operand b — ADNM_B_F32 (the fp32 master values, rounded / scaled on the way into the MFMA),
 *   ADNM_B_BF16 (its bf16 shadow, prec ADNM_MFMA_BF16 only) or ADNM_B_FP8 (its per-tensor scaled OCP e4m3 shadow, the fp8 modes only;
 *   b_scale = device pointer to the scale the shadow was made with: value = e4m3 / *b_scale; q's scale_b / amax_b are then unused).
 *   The shadows are written by the optimiser pass (adnm_adamw_step) or the parameter-prep kernels; ldb counts ELEMENTS in every case.
 *   A narrow weight halves / quarters the bytes of the weight-streaming shapes and gives bit for bit the result of the fp32 operand
 *   rounded the same way.
 * Workspace (adnm_skgemm_ws_bytes; 16 = the shape is not split): a reduction that would leave CUs idle is split over workgroups.
 *   TN: ws holds fp32 partials for the shared fold — ordinary device memory.
 *   NT / NN: the slabs are combined INSIDE the launch (arrival counters; the last workgroup of a tile adds the slabs in slice order:
 *     bitwise reproducible).  ws = [arrival counters: one int per output tile, rounded up to 256 B | slabs], 256-byte aligned ordinary
 *     device memory; the counters must be ZERO when the launch starts and are zero again when it ends, so one workspace serves every
 *     launch of a stream (launches on a stream are ordered) but must not be shared by launches that can run concurrently (another
 *     stream, another captured graph replayed beside this one).  slabs_uc (optional, NULL = none): uncached device memory
 *     (adnm_uncached_alloc) of at least the slab part; the slabs then live there — a completed slab store is at the device-wide
 *     coherence point, so the arrival ticket needs no agent-scope release / acquire (= no per-workgroup write-back / invalidate of an
 *     XCD's L2) — and ws only has to hold the counters (adnm_skgemm_counter_bytes).  Same sharing rule.  Both protocols give bitwise
 *     identical results.
 * adnm_uncached_alloc: `bytes` of zero-filled uncached device memory on the current device (hipExtMallocWithFlags +
 * hipDeviceMallocUncached), NULL on failure; a host-side setup call (it synchronises; not under stream capture). */
#define ADNM_SKGEMM_NT 0
#define ADNM_SKGEMM_NN 1
#define ADNM_SKGEMM_TN 2
int adnm_skgemm_supported(int op, int64_t M, int64_t N, int64_t K);
int64_t adnm_skgemm_ws_bytes(int op, int64_t M, int64_t N, int64_t K);   /* counters + slabs (NT / NN), partials (TN) */
int adnm_skgemm(int op, const float* a, int64_t lda, const void* b, int64_t ldb, int b_dtype, const float* b_scale, const float* bias, float* c,
                int64_t ldc, float* dbias, void* ws, int64_t ws_bytes, void* slabs_uc, int64_t slabs_uc_bytes, int64_t M, int64_t N, int64_t K,
                int prec, float* q, adnm_stream_t stream);
int64_t adnm_skgemm_counter_bytes(int op, int64_t M, int64_t N, int64_t K);
void* adnm_uncached_alloc(int64_t bytes);
int adnm_uncached_free(void* ptr);

/* ---------------------------------------------------------------- enRainfallLoss (K14)
 * models/loss.py:30-57 of the reference (train_untils.py:43 builds it with omega_t 0.57, alpha 0.25, gamma 0):
 * loss (1 float) and grad = d loss / d pred (n floats) in one pass over contiguous fp32 pred / target. */
int64_t adnm_rainloss_ws_bytes(int64_t n);
int adnm_rainloss(const float* pred, const float* target, float* loss, float* grad, void* ws, int64_t ws_bytes, int64_t n, float omega_t,
                  float alpha, float gamma, adnm_stream_t stream);

/* ---------------------------------------------------------------- head merge (K13)
 * y = cat((a1*x, a2*r), -1) [+ cat((a3*f, a4*f), -1)] — Block / Attention merge (ADNMUNet.py:124-131, :214-221) and WTLayer's
 * (model_untils.py:398-400).  x, r, f: (M,d) with row strides; f may be NULL; a_k: device scalars (NULL = 1); y: (M,2d)
 * contiguous.  bwd: dx, dr, df (M,d) contiguous (NULL skips), da[4] OVERWRITTEN (da[2..3] = 0 without f). */
int adnm_catmix_fwd(const void* x, int64_t ldx, const void* r, int64_t ldr, const void* f, int64_t ldf, const float* a1, const float* a2,
                    const float* a3, const float* a4, void* y, int64_t M, int64_t d, int dtype, adnm_stream_t stream);
int64_t adnm_catmix_bwd_ws_bytes(int64_t M, int64_t d);
int adnm_catmix_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* r, int64_t ldr, const void* f, int64_t ldf,
                    const float* a1, const float* a2, const float* a3, const float* a4, void* dx, void* dr, void* df, float* da, void* ws,
                    int64_t ws_bytes, int64_t M, int64_t d, int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- fused scalar / channel-affine mixes
 * y[m,c] = gamma[c] * ( s0*x0[m,c] + s1*x1[m,c] + s2*x2[m,c] )      x1,x2 optional (NULL), s_k NULL = 1, gamma NULL = 1
 * replaces the broadcast mul/add chains around the learnable scalars of Block.forward (ADNMUNet.py:152,158,161),
 * Attention.forward (:226,232,234), WTLayer / PatchEmbed / OutProj (model_untils.py:306-310,418-421,881-883) and
 * EncoderToDecoder (:785-787).  bwd: dx_k = s_k*gamma*dy (NULL skips), ds_k = sum dy*gamma*x_k, dgamma[c] = sum_m dy*mix;
 * ds_k / dgamma are OVERWRITTEN (NULL skips).  C % 4 == 0, C <= 2048. */
int adnm_lincomb_fwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, const void* x2, int64_t ld2,
                     const float* s0, const float* s1, const float* s2, const float* gamma, void* y, int64_t ldy,
                     int64_t M, int64_t C, int dtype, adnm_stream_t stream);
int64_t adnm_lincomb_bwd_ws_bytes(int64_t M, int64_t C);
int adnm_lincomb_bwd(const void* dy, int64_t lddy, const void* x0, int64_t ld0, const void* x1, int64_t ld1,
                     const void* x2, int64_t ld2, const float* s0, const float* s1, const float* s2,
                     const float* gamma, void* dx0, int64_t lddx0, void* dx1, int64_t lddx1, void* dx2, int64_t lddx2,
                     float* ds0, float* ds1, float* ds2, float* dgamma, void* ws, int64_t ws_bytes, int64_t M, int64_t C,
                     int dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- residual mix + the next pre-norm, one pass each way
 * mix = gamma * (s0*x0 + s1*x1)  (adnm_lincomb_fwd, bitwise)   and   xn = scale * ((mix - mu) * rstd * w + b) + shift  (adnm_rownorm_fwd)
 * — Block.forward's `x = beta1*x + beta2*mixer(..); xn = scale2*norm2(x) + shift2` and the pair around the FFN (ADNMUNet.py:149-158), the
 * same pairs in Attention.forward (:226-232).  fp32 rows, d % 4 == 0, d <= 1024.  bwd: dyn = d xn, dres = the gradient arriving on mix through
 * the residual path (NULL: none); writes dx0 / dx1 (NULL skips) and OVERWRITES ds0, ds1, dgamma, dw, db, dscale, dshift (NULL skips); an
 * accumulate mask announced with adnm_foldq_accumulate_next applies to {dgamma, ds0, ds1} as for adnm_lincomb_bwd. */
int adnm_mixnorm_fwd(const float* x0, int64_t ld0, const float* x1, int64_t ld1, const float* s0, const float* s1, const float* gamma,
                     const float* w, const float* b, const float* scale, const float* shift, float* ymix, int64_t ldm, float* yn,
                     int64_t ldn, float* mu, float* rstd, int64_t M, int64_t d, float eps, int subtract_mean, adnm_stream_t stream);
int64_t adnm_mixnorm_bwd_ws_bytes(int64_t M, int64_t d);
int adnm_mixnorm_bwd(const float* dyn, int64_t lddyn, const float* dres, int64_t lddres, const float* x0, int64_t ld0, const float* x1,
                     int64_t ld1, const float* s0, const float* s1, const float* gamma, const float* w, const float* b, const float* scale,
                     const float* mu, const float* rstd, float* dx0, int64_t lddx0, float* dx1, int64_t lddx1, float* ds0, float* ds1,
                     float* dgamma, float* dw, float* db, float* dscale, float* dshift, void* ws, int64_t ws_bytes, int64_t M, int64_t d,
                     int subtract_mean, adnm_stream_t stream);

/* ---------------------------------------------------------------- fused step glue on flat fp32 buffers (§8f rank 1)
 * clip_grad_norm_(max_norm) (train.py:140) + AdamW (train_untils.py:35-42) over n parameters laid out flat:
 *   state[1] = sum g^2;  coef = min(1, max_norm / (sqrt(state[1]) + 1e-6))  (max_norm <= 0: no clipping)
 *   state[0] += 1 (step);  p *= 1 - lr*wd;  m = lerp(m, coef*g, 1-beta1);  v = beta2*v + (1-beta2)(coef*g)^2
 *   p -= lr/(1-beta1^step) * m / (sqrt(v)/sqrt(1-beta2^step) + eps)        — torch.optim.AdamW's exact update order.
 * state: 4 device floats [step, sumsq, bc1, sqrt(bc2)], zero-initialised by the caller; n % 4 == 0.
 * shadow (optional, NULL = none): the narrow copy of the updated parameters the weight-streaming GEMMs read (adnm_skgemm b_dtype), written
 *   in the same pass — the values are in registers anyway, so the step pays 2 (bf16) or 1 (fp8) more bytes per parameter and the two
 *   passes that re-read the 72 M parameters every step (forward, input gradients) read a half / a quarter of the bytes:
 *   shadow_dtype ADNM_B_BF16: shadow[i] = bf16(p[i]), n uint16;
 *   shadow_dtype ADNM_B_FP8:  shadow[i] = e4m3(clamp(p[i] * scale_b(record of i's tensor))), n bytes.  Tensors = the segments of the flat
 *     buffer: seg_end[s] = END of segment s in units of 4 elements (ascending; tensors are 16-byte aligned in the flat layout),
 *     seg_rec[s] = row of its 8-float quantisation record in wtab (the layout of `q`; only scale_b / amax_b / fmax_b / record are used)
 *     or < 0 for a tensor no GEMM reads (its shadow bytes are unspecified).  While a record's `record` flag is set the pass collects
 *     max |p| of the UPDATED values into amax_b; adnm_quant_update (run over wtab BEFORE this pass in a step) turns it into the next
 *     scale_b, with which this pass writes the shadow and the next step's GEMMs read it: shadow and scale never disagree.
 * hyper (optional, NULL = use the arguments): two device floats {lr, max_norm} read by the kernels INSTEAD of the by-value arguments, so that
 *   a launch captured in a hipGraph (the tail graph of a multi-GPU step) follows the host's learning-rate schedule and adaptive clip
 *   threshold (train.py:122-130) — the host rewrites the two floats when they change.
 * adnm_shadow_refresh: the shadow alone from the parameters as they are (after they moved into the flat buffer, after a checkpoint load);
 *   shadow = NULL (fp8) with collect != 0: only collect max |p| (the first calibration). */
int64_t adnm_adamw_ws_bytes(void);
int adnm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float* state, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float max_norm, void* ws, int64_t ws_bytes, void* shadow, int shadow_dtype,
                    const int* seg_end, const int* seg_rec, int64_t nseg, float* wtab, const float* hyper, adnm_stream_t stream);
int adnm_shadow_refresh(const float* p, int64_t n, void* shadow, int shadow_dtype, const int* seg_end, const int* seg_rec, int64_t nseg,
                        float* wtab, int collect, adnm_stream_t stream);

/* ---------------------------------------------------------------- dense 3x3 'same' convolution, NHWC, on MFMA (K5)
 * nn.Conv2d(k=3, s=1, p=1) [+ bias] [+ GELU] of the U-Net conv stack: PatchEmbed.conv2 (model_untils.py:259-273), WTLayer.conv
 * (:376-387), OutProj.conv[0] / conv2 (:818-849).  Implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32).
 *   in:(B*H*W, K) pixel rows (row stride ldin), out:(B*H*W, N) (ldo); K = Cin, N = Cout (any sizes; 16-byte paths need % 4).
 *   w element (n, ky, kx, k) at w[n*ws_n + (ky*3+kx)*ws_tap + k*ws_k]: nn.Conv2d's own (Cout,Cin,3,3) layout is
 *   (ws_n, ws_tap, ws_k) = (9K, 1, 9), the channels-last one (Cout,3,3,Cin) is (9K, K, 1) — both are read in place.
 *   act: ADNM_ACT_NONE | ADNM_ACT_GELU applied in the epilogue; pre (optional, row stride ldpre) receives conv + bias BEFORE the
 *   activation — the tensor autograd would have saved for the reference's separate GELU; dgrad / wgrad take it back.
 *   dgrad: din = conv^T(dout * act'(pre));  wgrad: dw[n][tap][k] contiguous (= the channels-last weight layout), dbias optional;
 *   both OVERWRITE.  Workspaces: split-K partials of the deep maps (fwd / dgrad: adnm_conv3_ws_bytes with the (K, N) of THAT
 *   launch's reduction / output), per-wave partial rows (wgrad). */
int64_t adnm_conv3_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t K, int64_t N);
int adnm_conv3_fwd(const float* in, int64_t ldin, const float* w, int64_t ws_n, int64_t ws_tap, int64_t ws_k, const float* bias,
                   float* out, int64_t ldo, float* pre, int64_t ldpre, void* ws, int64_t ws_bytes, int64_t B, int64_t H, int64_t W,
                   int64_t K, int64_t N, int act, int prec, float* q, adnm_stream_t stream);
int adnm_conv3_dgrad(const float* dout, int64_t lddo, const float* pre, int64_t ldpre, int act, const float* w, int64_t ws_n,
                     int64_t ws_tap, int64_t ws_k, float* din, int64_t lddin, void* ws, int64_t ws_bytes, int64_t B, int64_t H,
                     int64_t W, int64_t K, int64_t N, int prec, float* q, adnm_stream_t stream);
int64_t adnm_conv3_wgrad_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t K, int64_t N);
int adnm_conv3_wgrad(const float* dout, int64_t lddo, const float* pre, int64_t ldpre, int act, const float* in, int64_t ldin,
                     float* dw, float* dbias, void* ws, int64_t ws_bytes, int64_t B, int64_t H, int64_t W, int64_t K, int64_t N,
                     int prec, adnm_stream_t stream);

/* ---------------------------------------------------------------- stride-2 transposed conv of UpSample (K9)
 * nn.ConvTranspose2d(C, C, k=3, s=2, p=1, output_padding=1) (model_untils.py:120-158,490-520) = one GEMM over the input pixels
 * (cols[B*H*W, 9C] = X . Wf, adnm_skgemm op NN on the weight as it lies) + these two data-movement kernels:
 *   col2im: out[b, oy, ox, :] = bias + the taps of cols whose parity matches (oy, ox) (1, 2, 2 or 4 of them: the 4 phases);
 *   im2col: dcols[(b,iy,ix), tap, :] = dout[b, 2iy-1+ky, 2ix-1+kx, :] (0 outside): then dX = dcols . Wf^T (op NT), dWf = X^T . dcols (op TN).
 * Column (tap, c) of a cols row sits at tap*col_tap_stride + c*col_c_stride: (C, 1) for the (Cin, 3, 3, Cout) weight order,
 * (1, 9) for nn.ConvTranspose2d's own (Cin, Cout, 3, 3).  out / dout: (B*2H*2W, C) pixel rows. */
int adnm_convt_col2im(const float* cols, int64_t ldc, int64_t col_tap_stride, int64_t col_c_stride, const float* bias, float* out,
                      int64_t ldo, int64_t B, int64_t H, int64_t W, int64_t C, adnm_stream_t stream);
int adnm_convt_im2col(const float* dout, int64_t lddo, float* dcols, int64_t ldc, int64_t col_tap_stride, int64_t col_c_stride,
                      int64_t B, int64_t H, int64_t W, int64_t C, adnm_stream_t stream);

/* ---------------------------------------------------------------- the data formats either side of the path
 * adnm_radar_ingest: datasets/Shanghai.py:52-59,121 — uint8 frames (frames, H0, W0), value * mul (1/255), bilinear resize to (S, S)
 *   (align_corners=False, no antialias = torchvision's tensor Resize) -> fp32 (frames, S, S).  src_u8 is a DEVICE pointer (the
 *   caller moves the bytes, e.g. by an async copy from pinned host memory).
 * adnm_eval_counts: datasets/Shanghai_metrics.py:49-152 — per frame [TP, FN, FP, TN] at each threshold on
 *   uint16(clip(x,0,1) * value_scale) fields (truth = "obs", pred = "sim") and sum |d|, sum d^2 of the scaled clipped float fields.
 *   thresholds_host: nthr <= 8 floats in HOST memory (they become kernel arguments).  out: (frames, 4*nthr + 2) fp32, OVERWRITTEN. */
int adnm_radar_ingest(const void* src_u8, float* dst, int64_t frames, int64_t H0, int64_t W0, int64_t S, float mul, adnm_stream_t stream);
int64_t adnm_eval_counts_ws_bytes(int64_t frames, int64_t hw, int64_t nthr);
int adnm_eval_counts(const float* truth, const float* pred, float* out, const float* thresholds_host, int64_t nthr, float value_scale,
                     void* ws, int64_t ws_bytes, int64_t frames, int64_t hw, adnm_stream_t stream);
/* adnm_eval_ssim: SimplifiedEvaluator.cal_ssim (datasets/Shanghai_metrics.py:132-152) — out[frame] = SUM over the valid (H-10) x (W-10)
 * region of the SSIM map (11x11 Gaussian window, sigma 1.5, C1 = (0.01 s)^2, C2 = (0.03 s)^2, float64 arithmetic like the reference) of
 * the [0,1]-clipped fields times value_scale s; the caller divides by the region's area (the reference takes the mean).  truth, pred:
 * (frames, H, W) fp32 contiguous.  The window is the normalised sampled Gaussian cv2.getGaussianKernel(11, 1.5) documents. */
int64_t adnm_eval_ssim_ws_bytes(int64_t frames, int64_t H, int64_t W);
int adnm_eval_ssim(const float* truth, const float* pred, float* out, float value_scale, void* ws, int64_t ws_bytes, int64_t frames,
                   int64_t H, int64_t W, adnm_stream_t stream);

/* ---------------------------------------------------------------- stand-alone activations
 * act_fwd / act_bwd: y = act(x), dpre = dy * act'(pre) over flat fp32 arrays (n % 4 == 0), act in {ADNM_ACT_SILU, ADNM_ACT_GELU}: nn.GELU
 *   between Mlp.fc1 and fc2 (model_untils.py:52-70) and the backward of the GELUs fused into GEMM / conv epilogues.
 * swish_fwd / bwd: Swish with a learnable slope, y = x * sigmoid(beta * x) (model_untils.py:162-169), beta a 1-element device tensor;
 *   bwd OVERWRITES dx and dbeta (1 element). */
int adnm_act_fwd(const float* x, float* y, int64_t n, int act, adnm_stream_t stream);
int adnm_act_bwd(const float* dy, const float* pre, float* dpre, int64_t n, int act, adnm_stream_t stream);
int adnm_swish_fwd(const float* x, const float* beta, float* y, int64_t n, adnm_stream_t stream);
int64_t adnm_swish_bwd_ws_bytes(int64_t n);
int adnm_swish_bwd(const float* dy, const float* x, const float* beta, float* dx, float* dbeta, void* ws, int64_t ws_bytes, int64_t n,
                   adnm_stream_t stream);

/* Wire format of the data-parallel gradient all-reduce that replaces nn.DataParallel's reduce_add_coalesced (train.py:99-102;
 * SURVEY.md §8e): dst[i] = (bf16)(scale * src[i]) before the collective, dst[i] = scale * (float)src[i] after it (scale = 1/world
 * folds the average in).  n elements, both buffers 16-byte aligned. */
int adnm_cast_f32_bf16(const void* src, void* dst, int64_t n, float scale, adnm_stream_t stream);
int adnm_cast_bf16_f32(const void* src, void* dst, int64_t n, float scale, adnm_stream_t stream);

/* ---------------------------------------------------------------- tall-skinny fp32 GEMMs on MFMA (K6)
 * The Linear / 1x1 projections of the full-resolution stages (ADNssd.py:309,461; model_untils.py:64,67,193,196,831;
 * ADNMUNet.py:634): M = B*H*W tokens, K,N <= 256.  v_mfma_f32_16x16x4_f32: exact fp32 (an fmaf chain).
 *   nt: Y[M,N] = X[M,K] . Wp^T (+bias[N]),  Wp[n][k] = w[n*ws_n + k*ws_k]   (forward: ws_n=K, ws_k=1;
 *       input gradient dX = dY . W: call with X:=dY, N:=K_w, K:=N_w, ws_n=1, ws_k=K_w).  K % 16 == 0.
 *   tn: dW[N,K] = dY[M,N]^T . X[M,K], dbias[N] = column sums of dY (NULL skips); OVERWRITTEN; ws from *_ws_bytes.
 * *_supported() return 1 when the shape fits the kernels (callers use the library GEMM otherwise). */
int adnm_tsgemm_supported(int64_t M, int64_t N, int64_t K);
/* x_dtype / y_dtype (ADNM_F32 | ADNM_BF16): storage type of the token rows read / written; bf16 storage needs prec = ADNM_MFMA_BF16
 * (the wide intermediates of the full-resolution level are kept in bf16 in the bf16 configuration: half the bytes). */
int adnm_tsgemm_nt(const void* x, int64_t ldx, const float* w, int64_t ws_n, int64_t ws_k, const float* bias, void* y,
                   int64_t ldy, int64_t M, int64_t N, int64_t K, int prec, float* q, int x_dtype, int y_dtype, adnm_stream_t stream);
int adnm_tsgemm_tn_supported(int64_t M, int64_t N, int64_t K);
int64_t adnm_tsgemm_tn_ws_bytes(int64_t M, int64_t N, int64_t K);
int adnm_tsgemm_tn(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, float* dbias, void* ws,
                   int64_t ws_bytes, int64_t M, int64_t N, int64_t K, int dy_dtype, int x_dtype, adnm_stream_t stream);

/* ---------------------------------------------------------------- parameter-side preparation (one launch each way)
 * ADN-SSD mixer: reference-layout parameters -> kernel-layout tensors (row-permuted in_proj, effective 3x3 taps of
 * the conv2d / asymmetric 1x3o3x1 chains (ADNssd.py:334,343-346) in tap-major order, permuted LayerNorm weights,
 * alpha1 * column-permuted out_proj (ADNssd.py:459)) and the transpose of that map for the gradients.
 *   params[15]  = {in_proj.weight, conv2d.weight, conv_31_x1, conv_31_bc1, conv_31_x2, conv_31_bc2, conv_13_x1,
 *                  conv_13_bc1, conv_13_x2, conv_13_bc2, conv2d_z.weight, norm.weight, norm.bias, out_proj.weight, alpha1}
 *   prepped[6]  = {w_in (d_in_proj,dm), cw (9,di+2gn), czw (9,di), ln_w (di), ln_b (di), w_out (dm,2di)}
 * All fp32 device pointers; the tables themselves are host arrays.  gn = ngroups*d_state.
 * tap_ld: row stride (floats) of BOTH tap images, 0 = dense (di+2gn resp. di).  With tap_ld = 2di+2gn, czw = T and cw = T + di are
 * the two column ranges of ONE (9, 2di+2gn) image T = [czw | cw]: the taps of a single depthwise launch over [z | xBC]. */
int adnm_adnprep_fwd(float* const* params, float* const* prepped, int64_t d_model, int64_t d_inner, int64_t gn,
                     int64_t headdim, int64_t tap_ld, adnm_stream_t stream);
int64_t adnm_adnprep_bwd_ws_bytes(void);
int adnm_adnprep_bwd(float* const* params, float* const* gprepped, float* const* dparams, int64_t d_model,
                     int64_t d_inner, int64_t gn, int64_t headdim, int64_t tap_ld, void* ws, int64_t ws_bytes,
                     adnm_stream_t stream);
/* WTConv2d: taps[k] (K*K, Cgp) = tap-major( w[k] (Cg,K*K) * s[k] (Cg) ), zero-padded from C to Cp channels
 * (Cg = C for k = 0, the base conv; 4C for the level convs k = 1..levels), bias_t = bias * s[0]
 * (WTConv2d.py:123,146).  bwd: dw, ds, dbias from the tap gradients. */
int adnm_wtprep_fwd(float* const* w, float* const* s, float* bias, float* const* taps, float* bias_t, int64_t C,
                    int64_t Cp, int64_t K, int64_t levels, adnm_stream_t stream);
int adnm_wtprep_bwd(float* const* w, float* const* s, float* bias, float* const* gtaps, float* gbias_t,
                    float* const* dw, float* const* ds, float* dbias, int64_t C, int64_t Cp, int64_t K, int64_t levels,
                    adnm_stream_t stream);
/* Grouped forms: every ADN-SSD mixer / every WTConv2d of a model stage in ONE launch each way (36 per-module launches of a few
 * microseconds each at config 2).  Tables are concatenated per module i: params[15 i ..], prepped / gprepped[6 i ..], dparams[15 i ..],
 * dims[5 i ..] = {d_model, d_inner, gn, headdim, tap_ld};  WTConv2d: w / s / taps / gtaps / dw / ds[5 i ..] (entries 0 .. levels), bias[i],
 * bias_t[i], gbias_t[i], dbias[i] (all NULL or all set per module), dims[4 i ..] = {C, Cp, K, levels}.  More than 8 modules are cut into
 * several launches.  Same results as the per-module entry points, bit for bit.
 * adnm_adnprep_fwd_multi, narrow (optional, NULL = none): narrow[5 i ..] = {w_in_n, w_out_n, s_in, s_out, s_out_eff} — NARROW copies of the
 * two big matrices of mixer i for the weight-streaming GEMMs (adnm_skgemm b_dtype = narrow_dtype) INSTEAD of the fp32 ones (prepped[6 i]
 * resp. prepped[6 i + 5] may then be NULL): ADNM_B_BF16: bf16 values; ADNM_B_FP8: w_in_n = e4m3(w_in * *s_in), w_out_n = e4m3(w_out * e)
 * with e = *s_out / |alpha1| written to *s_out_eff (s_in / s_out: the scale_b of in_proj.weight's / out_proj.weight's quantisation
 * records — a row / column permutation keeps a tensor's maximum, alpha1 scales it). */
int adnm_adnprep_fwd_multi(int64_t n, float* const* params, float* const* prepped, const int64_t* dims, float* const* narrow, int narrow_dtype,
                           adnm_stream_t stream);
int64_t adnm_adnprep_bwd_multi_ws_bytes(int64_t n);
int adnm_adnprep_bwd_multi(int64_t n, float* const* params, float* const* gprepped, float* const* dparams, const int64_t* dims, void* ws,
                           int64_t ws_bytes, adnm_stream_t stream);
int adnm_wtprep_fwd_multi(int64_t n, float* const* w, float* const* s, float* const* bias, float* const* taps, float* const* bias_t,
                          const int64_t* dims, adnm_stream_t stream);
int adnm_wtprep_bwd_multi(int64_t n, float* const* w, float* const* s, float* const* bias, float* const* gtaps, float* const* gbias_t,
                          float* const* dw, float* const* ds, float* const* dbias, const int64_t* dims, adnm_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* ADNM_HIP_H */
