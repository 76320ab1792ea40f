/* automoe_hip.h -- C ABI of libautomoe_hip.so: the gfx950 (MI355X) kernels behind the AutoMoE
 * data-parallel train-step hot path.
 *
 * The reference (immanuel-peter/self-driving-model) is pure Python; it has no FFI of its own.
 * Its "operator API" for this path is torch.nn / torchvision / scipy calls, so every entry point
 * below names the reference call site (file:line under /root/reference) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C symbols, raw device pointers + sizes + a hipStream_t (passed as void*); no torch types;
 *   - every call returns int: AM_OK (0) or a negative AM_ERR_* code; nothing throws across the ABI;
 *   - the caller owns every buffer (scratch included: am_conv_wgrad_workspace_bytes / am_conv_wgrad_ws); the library
 *     allocates nothing, and calls on different streams are independent.  What it keeps beyond the call: the process-wide
 *     tuning table of am_set_tuning() (A/B switches, set before launching; no environment variable is read anywhere), a
 *     per-host-thread record of the last conv kernel launched (am_conv_last_variant), idempotent per-device "large-LDS
 *     attribute set" flags, and the diagnostic counters behind am_diag_ring_clock (written by workgroup 0 of a ring launch);
 *   - all launches are asynchronous on `stream`; no call synchronises the device;
 *   - activations are NHWC ("pixel-major") with element type `dtype` (AM_F32 or AM_F16);
 *     accumulation is always fp32 (fp64 for BatchNorm statistics).
 */
#ifndef AUTOMOE_HIP_H
#define AUTOMOE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AM_OK 0
#define AM_ERR_ARG (-1)         /* bad argument (null pointer, bad size, unsupported shape) */
#define AM_ERR_LAUNCH (-2)      /* the HIP runtime refused the launch */
#define AM_ERR_UNSUPPORTED (-3) /* shape/dtype combination not built */

#define AM_F32 0
#define AM_F16 1

#define AM_MAX_TAPS 16
#define AM_STATS_REPLICAS 16

typedef void* am_stream_t; /* hipStream_t */

int am_version(void);

/* ------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM ("gather-GEMM").
 * Replaces torch.nn.Conv2d forward / input-gradient at: ResNet-18 trunk convs
 * (models/experts/bdd_detection_expert.py:9-10 via torchvision), expert heads
 * (bdd_detection_expert.py:12-16, bdd_segmentation_expert.py:13-17, bdd_drivable_expert.py:13-17),
 * EasyBackbone convs (models/policy/trajectory_head.py:9-22), and their autograd backward
 * (training/train_bdd100k_ddp.py:97, training/train_gating_network.py:101).
 *
 * One geometry struct describes forward, stride-1 dgrad and each parity class of a strided
 * dgrad: output row m -> (image n, my, mx) on an MH x MW sub-grid;
 *   output pixel  = (my*oys + oy0, mx*oxs + ox0) of an OH x OW image, pixel stride ldo elements;
 *   tap t gathers `krun` contiguous elements starting at input pixel (my*iys + dy[t], mx*ixs + dx[t]),
 *   channel x_coff, of an IH x IW image with pixel stride ldi elements; chunks (16 B) whose pixel
 *   falls outside the image read as zero (that is the conv zero padding).
 * GEMM view: M = B*MH*MW rows, K = ntaps*krun, N output channels;
 *   y[m, n] = act( sum_k gather(m, k) * w[n, k] + bias[n] ).
 * Packed weights `w`: row-major [am_conv_npad(N)][ntaps*krun] of `dtype`, zero padded rows.
 * krun*sizeof(dtype) must be a multiple of 64.  pix_shift: log2(elements per input pixel) when a
 * run spans several pixels (first-layer trick: 8 pixels x 8 channels), 31 when it covers one.
 * osplit (0 = off; needs N == 2*osplit, f16, a shape the LDS-DMA ring kernels cover, else AM_ERR_UNSUPPORTED): output
 * columns [0, osplit) of row m are stored at its output pixel, columns [osplit, N) at that pixel + osplit_stride elements.
 * This is how the input gradient of a 3x3 / stride-2 / pad-1 convolution runs as ONE gather-GEMM instead of four parity-
 * class launches: a row is the 2x2 block of dX pixels (2my.., 2mx..), its N = 4*Cin columns are ordered (py, px, ci), the
 * four taps are the 2x2 neighbourhood of dY pixels it depends on (weights zero where a class does not use a tap); with
 * osplit = 2*Cin and osplit_stride = one image row the block leaves as two contiguous 2*Cin-element segments (full
 * cache lines) instead of four strided Cin-element ones.
 * `stats` (optional, may be NULL): [AM_STATS_REPLICAS][2][N] fp64, zeroed by the caller;
 * receives per-channel sum and sum of squares of the pre-bias accumulator (BatchNorm batch
 * statistics, torch.nn.BatchNorm2d in train mode) -- bias-free so the variance is shift-free.
 * ------------------------------------------------------------------------------------------ */
typedef struct am_conv_geom {
  int32_t B, MH, MW;
  int32_t IH, IW, ldi, x_coff;
  int32_t OH, OW, ldo, y_coff;
  int32_t oys, oy0, oxs, ox0;
  int32_t iys, ixs;
  int32_t ntaps, krun, pix_shift;
  int32_t N;
  int16_t dy[AM_MAX_TAPS];
  int16_t dx[AM_MAX_TAPS];
  int32_t osplit, osplit_stride; /* 0, or: columns [osplit, N) of an output row land osplit_stride elements after its pixel */
} am_conv_geom;

int am_conv_npad(int N);
int am_conv_gemm(const am_conv_geom* g, int dtype, const void* x, const void* w, const float* bias,
                 int relu, void* y, double* stats, am_stream_t stream);

/* Fused first layer for frozen (no-gradient) stems in train-mode BatchNorm, f16, space-to-depth geometry only
 * (conv -> BatchNorm2d(batch statistics) -> ReLU of bdd_*_expert.py:9-11 without writing the raw conv output):
 *   mode 1: accumulate the BatchNorm statistics of the conv output into `stats`, write nothing;
 *   mode 2: y = relu(conv * scale[n] + shift[n]) with scale/shift from am_bn_finalize;
 *   mode 3: y = MaxPool2d(3,2,1)(relu(conv * scale[n] + shift[n])), y = [B,(OH-1)/2+1,(OW-1)/2+1,ldo] (ResNet stem + maxpool);
 *   mode 4: the whole frozen stem in ONE pass over the image: `scale` = the BatchNorm weight gamma[64] (only its signs are
 *           used), y = MaxPool2d(3,2,1)(s[n] * conv) with s[n] = -1 where gamma[n] < 0 else +1 (the f16-rounded conv output
 *           pooled as it is), and `stats` += the BatchNorm sums of s * conv.  MaxPool commutes with the per-channel monotone
 *           map v -> relu(v * scale + shift) (torchvision resnet.py: conv1 -> bn1 -> relu -> maxpool, bdd_*_expert.py:9-11), so
 *           the consumers apply it to the POOLED map: am_bn_finalize_signed turns the sums into scale >= 0 / shift for s * conv,
 *           am_conv_gemm_prebn / am_bn_apply2 (relu bit 1) / am_bn_apply consume (y, scale, shift).  Equal bit for bit to
 *           am_conv_gemm -> am_bn_apply(relu) -> am_maxpool3x3s2_fwd given the same statistics.
 * Returns AM_ERR_UNSUPPORTED when the geometry / size is not covered (caller uses am_conv_gemm + am_bn_apply). */
int am_conv_first_fused(const am_conv_geom* g, int dtype, int mode, const void* x, const void* w, const float* scale,
                        const float* shift, void* y, double* stats, am_stream_t stream);

/* am_conv_gemm on relu(x * pre_scale[c] + pre_shift[c]): the BatchNorm(+ReLU) of the PRODUCING layer (scale / shift from
 * am_bn_finalize, am_bn_apply's fp32 arithmetic) is applied while the input tile is staged in LDS, so that layer's normalised
 * output never goes through HBM (ResNet BasicBlock conv1 -> bn1 -> relu -> conv2 with nothing else reading the middle tensor,
 * torchvision resnet.py BasicBlock.forward as used by bdd_*_expert.py:9-11).  Zero padding applies to the transformed tensor.
 * No bias / ReLU epilogue; `stats` as in am_conv_gemm.  Returns AM_ERR_UNSUPPORTED unless the layer is a dense 3x3 / stride 1 /
 * pad 1 f16 convolution whose kernel stages an input patch in LDS: 64 -> 64 (weights-in-registers kernel) or Cin <= 256 ->
 * 64 < N <= 128 (halo-staged kernel), large enough for it (caller: am_bn_apply + am_conv_gemm). */
int am_conv_gemm_prebn(const am_conv_geom* g, int dtype, const void* x, const float* pre_scale, const float* pre_shift,
                       const void* w, void* y, double* stats, am_stream_t stream);

/* am_conv_gemm with a residual epilogue, y = act(conv(x, w) + bias[n] + res): the end of a ResNet BasicBlock in inference
 * (torchvision resnet.py BasicBlock.forward `out += identity; out = relu(out)` as used by bdd_*_expert.py:9-11 through
 * inference/run_automoe.py:34-53; eval-mode BatchNorm folded into w / bias by the caller), so no normalise + add pass and no
 * raw conv output.  `res` has y's geometry (same pixel stride and channel offset).  The conv + bias is rounded to f16 before
 * the residual is added (the same two roundings as am_conv_gemm followed by am_bn_apply).  f16 only; returns
 * AM_ERR_UNSUPPORTED for shapes outside the LDS-DMA ring / weights-in-registers kernels (caller: am_conv_gemm + am_bn_apply). */
int am_conv_gemm_res(const am_conv_geom* g, int dtype, const void* x, const void* w, const float* bias, const void* res,
                     int relu, void* y, am_stream_t stream);

/* Diagnostic (bench.py roofline leg, kernel tests): which kernel the last am_conv_gemm / am_conv_first_fused / am_conv_wgrad call
 * made by the CALLING HOST THREAD launched (thread-local record).
 * 0 none, 1 conv_ring_k<256,256>, 2 conv_ring_k<256,128>, 3 conv3x3_c64n64_duo_k, 4 (retired), 5 (retired),
 * 6 conv_gemm2_k, 7 (retired), 8 conv_gemm_k (register-staged), 9 conv_s2d_k, 10 conv_s2d_pool_k,
 * 11 conv_ring16_k<256,256>, 12 conv_ring16_k<256,128>, 13 wgrad_ring_k, 14 conv_wgrad_k (register-staged), 15 conv_s2d_wgrad_k,
 * 16 conv_halo_k, 17 conv_patch_wgrad_k, 18 conv_band16_k, 19 conv_ring16_k<128,256>. */
int am_conv_last_variant(void);

/* Process-wide A/B switches between kernels that compute the same result (diagnostic / tuning use: tests pin a kernel, bench
 * compares two).  The only mutable state the library keeps besides the per-thread am_conv_last_variant() record.  Set before
 * launching; returns the previous value (AM_ERR_ARG for an unknown key).
 *   AM_TUNE_RING   0: conv_ring_k (v_mfma_f32_32x32x16_f16), 1: conv_ring16_k (16x16x32, transposed product, pieces issued as a
 *                  block), 2: the same with static issue priority for waves 4-7, 3: pieces placed by wave age, 4 (default): fragment
 *                  reads and pieces interleaved with the MFMAs of their half K-step.
 *   AM_TUNE_RING128_MIN_TILES   fewest 256x128 tiles (M/256 * N/128) for which conv_ring_k<256,128> is dispatched.
 *   AM_TUNE_WGRAD_RING   1: wgrad_ring_k where its shape conditions hold, 2: the same with v_mfma_f32_16x16x32_f16 on its
 *                  256-channel tile (measured: no gain per step), 0: always the register-staged conv_wgrad_k.
 *   AM_TUNE_WGRAD_MAX_SLABS   am_conv_wgrad_ws keeps one slab per pixel chunk up to this many chunks; beyond it the chunks add
 *                  atomically into ONE zero-filled slab (0: always; a huge value: never).
 *   AM_TUNE_RING_SHORT_K   contractions of at most this many 32-element K-steps take the 256x128 ring tile even when N >= 256;
 *   AM_TUNE_HALO_MIN_TILES 3x3 / stride-1 layers with 64 < N <= 128 take the halo-staged kernel (conv_halo_k) from this many 8x32-pixel
 *                          tiles on (default 256; a huge value sends them to the ring kernel).
 *   AM_TUNE_PATCH_WGRAD_MIN_TILES the weight gradient of 64 -> 64 channel 3x3 / stride-1 layers takes the patch-staged kernel
 *                          (conv_patch_wgrad_k) from this many 8x32-pixel tiles on (default 512; a huge value: never); the same
 *                          bound, in 8x16-pixel tiles, for the 128 -> 128 channel form.
 *   AM_TUNE_PATCH_WGRAD_C128 1 (default): 128 -> 128 channel 3x3 / stride-1 layers take conv_patch_wgrad_k<128> (three workgroups
 *                          per tile stream, one per horizontal tap), 0: wgrad_ring_k.
 *   AM_TUNE_DUO_MFMA16     conv3x3_c64n64_duo_k with v_mfma_f32_16x16x32_f16 and a 160-byte patch pitch (1) or with
 *                          v_mfma_f32_32x32x16_f16 and a 144-byte pitch (0).
 *   AM_TUNE_BAND_MIN_TILES 3x3 / stride-1 layers with N a multiple of 256 take the row-band halo kernel (conv_band16_k) from this
 *                          many 256x256 tiles on (default 200: conv_ring16_k's gate; a huge value sends them to the ring kernels).
 *   AM_TUNE_RING_DIAG      1: conv_ring16_k launches its diagnostic instantiation (workgroup 0 stamps the K-loop's cycle / wall
 *                          counters for am_diag_ring_clock); 0 (default): the production kernel carries no stamp.
 *   AM_TUNE_RING16_M128_MIN_TILES problems with N >= 256 that make fewer than 200 tiles of 256x256 (layer 4 at B <= 16, the 512 -> 256
 *                          heads) take conv_ring16_k's 128x256 tile from this many 128x256 tiles on (default 200: +3-6 % over conv_ring_k<256,128>
 *                          there, slower below; a huge value: never). */
#define AM_TUNE_RING 0
#define AM_TUNE_RING128_MIN_TILES 1
#define AM_TUNE_WGRAD_RING 2
#define AM_TUNE_WGRAD_MAX_SLABS 3
#define AM_TUNE_RING_SHORT_K 4
#define AM_TUNE_HALO_MIN_TILES 5
#define AM_TUNE_PATCH_WGRAD_MIN_TILES 6
#define AM_TUNE_PATCH_WGRAD_C128 7
#define AM_TUNE_DUO_MFMA16 8
#define AM_TUNE_BAND_MIN_TILES 9
#define AM_TUNE_RING_DIAG 10
#define AM_TUNE_RING16_M128_MIN_TILES 11
#define AM_TUNE_COUNT 12
int am_set_tuning(int key, int value);
int am_get_tuning(int key);

/* Diagnostic (bench.py roofline leg): what workgroup 0 of the last 256x256 ring launch measured with s_memtime --
 * out[0] shader-clock cycles of its K-loop, out[1] ticks of the constant 100 MHz clock (s_memrealtime) over the same span,
 * out[2] K-steps (the MFMA floor is 1024 cycles per K-step), out[3] cycles from kernel entry to the K-loop (address set-up, the
 * first two tiles' flight), out[4] cycles from the K-loop's end to the last output store issued (statistics, staging, stores);
 * out[3], out[4] are 0 for the conv_ring_k generation.  out[0] / out[1] * 100 MHz = the clock the chip held under that kernel.
 * `out` holds 5 values.  Synchronises `stream`. */
int am_diag_ring_clock(long long* out, am_stream_t stream);

/* Weight gradient of the same gather-GEMM (torch conv2d backward w.r.t. weight):
 *   dw[n, t*krun + r] += scale * sum_m dy[m, n] * gather(m, t, r),  fp32, atomically accumulated,
 * dw row-major [>=N rows][ntaps*krun] (caller zeroes it, e.g. zero_grad).  `dy` is read at the
 * geometry's OUTPUT pixels (pixel stride ldo, channel y_coff).  `scale` undoes loss scaling. */
int am_conv_wgrad(const am_conv_geom* g, int dtype, const void* x, const void* dy, float scale,
                  float* dw, am_stream_t stream);
/* Workspace form of am_conv_wgrad (the caller owns scratch memory, sized by a query): every pixel chunk of the split
 * contraction stores its partial dW tile with plain stores into its own slab of `workspace`; a second pass sums the slabs and
 * writes dw_oihw[n][c][t] (the nn.Conv2d weight layout, c < cin, t = kh*KW + kw), = or += by `accumulate` -- no device-scope fp32
 * atomics (they bound the atomic form at ~1.3 TB/s of added bytes), no zero-filled staging tensor, no re-layout pass, and a
 * bitwise reproducible sum.  Geometries whose contraction splits into more than AM_TUNE_WGRAD_MAX_SLABS pixel chunks (tiny dW,
 * millions of pixels) keep the atomics but aim them at ONE slab the call zero-fills itself, and share the second pass (not
 * bitwise reproducible there).  am_conv_wgrad_workspace_bytes() gives the size for a geometry (0: that geometry runs a kernel
 * without a workspace form -- the 3-channel first layers -- and am_conv_wgrad_ws returns AM_ERR_UNSUPPORTED: use am_conv_wgrad).
 * `workspace` must be 16-byte aligned; it needs no initialisation. */
int am_conv_wgrad_workspace_bytes(const am_conv_geom* g, int dtype, long long* bytes);
int am_conv_wgrad_ws(const am_conv_geom* g, int dtype, const void* x, const void* dy, float scale, void* workspace,
                     long long workspace_bytes, float* dw_oihw, int cin, int accumulate, am_stream_t stream);
/* am_bn_bwd_apply + am_conv_wgrad in one launch for a layer whose input needs no gradient (the 3-channel first layers:
 * trajectory_head.py:10-12 conv -> BN -> ReLU on the image): `dy` is the gradient w.r.t. the BN(+ReLU) output, `raw` the conv
 * output, `yout` the BN+ReLU output (ReLU mask; NULL without ReLU), mean/rstd the saved batch statistics, coef = [3][N] from
 * am_bn_bwd_finalize; the gradient w.r.t. the conv output is formed on the fly with am_bn_bwd_apply's arithmetic and never
 * written.  Returns AM_ERR_UNSUPPORTED unless the geometry is a space-to-depth first layer in f16 (caller: the two-step form). */
int am_conv_wgrad_bn(const am_conv_geom* g, int dtype, const void* x, const void* dy, const void* yout, const void* raw,
                     const float* mean, const float* rstd, const float* coef, int relu, float scale, float* dw,
                     am_stream_t stream);
/* am_conv_wgrad_bn for BatchNorm + ReLU layers WITHOUT a residual (every first layer): the ReLU mask is the sign of the layer's own
 * normalised output raw * bn_scale[n] + bn_shift[n] (scale / shift as am_bn_finalize wrote them in forward, am_bn_apply's fp32
 * arithmetic), recomputed from the conv output that is read anyway -- the activation tensor is not read. */
int am_conv_wgrad_bn_sign(const am_conv_geom* g, int dtype, const void* x, const void* dy, const void* raw, const float* mean,
                          const float* rstd, const float* coef, const float* bn_scale, const float* bn_shift, float scale,
                          float* dw, am_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * BatchNorm2d around the conv (torch.nn.BatchNorm2d, eps 1e-5, momentum 0.1: ResNet-18 via
 * bdd_*_expert.py:9-11; models/policy/trajectory_head.py:10-22).  P = pixels (B*H*W), NHWC rows with
 * leading dimension ld* (elements).
 *   am_bn_finalize   training: batch mean/var from the conv's fp64 sums (+conv bias), updates the
 *                    running stats (unbiased var), emits scale = gamma*rstd, shift = beta - mean*scale
 *                    and saves mean/rstd for backward.  eval: scale/shift from the running stats.
 *   am_bn_apply      y = act(x*scale + shift (+ residual))      (BasicBlock add + ReLU fused)
 *   am_bn_bwd_*      dz = dy * (yout > 0); sums = [sum dz, sum dz*xhat] (fp64 replicas);
 *                    finalize: dgamma += gscale*sum dz*xhat, dbeta += gscale*sum dz, coef[3][C];
 *                    apply: dx = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)), dz_out = dz.
 *   am_bias_relu_bwd heads (conv + bias + ReLU, no BN): dz = dy*(yout>0), dbias += gscale*colsum(dz).
 * ------------------------------------------------------------------------------------------ */
int am_bn_finalize(const double* stats, int nrep, double count, const float* conv_bias, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                   int training, float* scale, float* shift, float* save_mean, float* save_rstd, int C,
                   am_stream_t stream);
/* am_bn_finalize (training) for sums taken on x' = s * x, s[c] = -1 where gamma[c] < 0 else +1 (am_conv_first_fused mode 4):
 * scale = |gamma| * rstd, shift = beta - mean(x') * scale -- the affine map for x' --, running_mean / running_var updated
 * with the statistics of x itself (mean(x) = s * mean(x'), var(x) = var(x')). */
int am_bn_finalize_signed(const double* stats, int nrep, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                          int C, am_stream_t stream);
int am_bn_apply(int dtype, const void* x, int ldx, const float* scale, const float* shift, const void* res, int ldr,
                int relu, void* y, int ldy, long long P, int C, am_stream_t stream);
/* am_bn_apply whose residual is itself a raw conv output normalised on the fly (the downsample branch of a strided
 * BasicBlock: y = relu(bn2(conv2) + bn_d(conv_d)), torchvision resnet.py BasicBlock.forward): res' = round(res*res_scale +
 * res_shift) as a separate am_bn_apply pass would have stored it.  res_scale/res_shift both NULL: am_bn_apply.
 * relu: bit 0 = ReLU on the sum (as am_bn_apply), bit 1 = ReLU on the transformed residual before it is rounded (the residual
 * is a BatchNorm + ReLU output that was never written: the pooled stem of am_conv_first_fused mode 4 feeding layer1's block). */
int am_bn_apply2(int dtype, const void* x, int ldx, const float* scale, const float* shift, const void* res, int ldr,
                 const float* res_scale, const float* res_shift, int relu, void* y, int ldy, long long P, int C,
                 am_stream_t stream);
int am_bn_bwd_reduce(int dtype, const void* dy, int lddy, const void* yout, int ldyo, const void* x, int ldx,
                     const float* mean, const float* rstd, int relu, double* sums, long long P, int C,
                     am_stream_t stream);
int am_bn_bwd_finalize(const double* sums, int nrep, double count, const float* gamma, const float* rstd,
                       float gscale, float* dgamma, float* dbeta, float* coef, int C, am_stream_t stream);
int am_bn_bwd_apply(int dtype, const void* dy, int lddy, const void* yout, int ldyo, const void* x, int ldx,
                    const float* mean, const float* rstd, const float* coef, int relu, void* dx, int lddx,
                    void* dz_out, int lddz, long long P, int C, am_stream_t stream);
/* am_bn_bwd_reduce / am_bn_bwd_apply for BatchNorm + ReLU layers WITHOUT a residual (bn1 of a BasicBlock, stems): the ReLU mask
 * (y > 0) is the sign of x * scale[c] + shift[c] -- the layer's own normalised output, scale / shift as am_bn_finalize wrote them in
 * forward -- recomputed from the conv output `x` both passes read anyway, so the activation tensor is not read: 2 instead of 3
 * tensor reads in the reduce pass, 2 + 1 write instead of 3 + 1 in the apply pass.  (A positive value below half the smallest
 * f16 subnormal was stored as 0 by am_bn_apply and masked by the *_bwd_* entries above; here it passes: fp32 autograd's rule.) */
int am_bn_bwd_reduce_sign(int dtype, const void* dy, int lddy, const void* x, int ldx, const float* mean, const float* rstd,
                          const float* scale, const float* shift, double* sums, long long P, int C, am_stream_t stream);
int am_bn_bwd_apply_sign(int dtype, const void* dy, int lddy, const void* x, int ldx, const float* mean, const float* rstd,
                         const float* coef, const float* scale, const float* shift, void* dx, int lddx, long long P, int C,
                         am_stream_t stream);
int am_bias_relu_bwd(int dtype, const void* dy, int lddy, const void* yout, int ldyo, int relu, void* dz_out,
                     int lddz, float* dbias, float gscale, long long P, int C, int Cvalid, am_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Spatial, HBM-bound kernels.
 *   am_nchw_to_nhwc / am_nhwc_to_nchw   boundary layout change: the reference API is NCHW fp32
 *       ([B,3,H,W] images, [B,C,h,w] head outputs); inside, activations are NHWC `dtype` with the
 *       channel count padded to `ld`.
 *   am_maxpool3x3s2_*    nn.MaxPool2d(3,2,1) of the ResNet stem (first max in scan order wins).
 *   am_gap_nhwc_*        nn.AdaptiveAvgPool2d(1) on NHWC activations (trajectory_head.py:25).
 *   am_gap_plane_*       the same over NCHW fp32 planes (expert_extractors.py:28,62,89).
 *   am_bilinear_up_*     F.interpolate(mode='bilinear', align_corners=False)
 *                        (bdd_segmentation_expert.py:22, bdd_drivable_expert.py:22): low-res NHWC
 *                        `dtype` -> full-res NCHW fp32 and its adjoint (x mul x *dev_scale).
 *   am_ce2d_*            nn.CrossEntropyLoss(ignore_index) over [B,C,H,W] logits and int64 targets
 *                        (train_bdd100k_ddp.py:58,193); acc2 = {sum of losses, valid count} fp64 on
 *                        the device; backward reads the count and grad_out from device memory.
 * ------------------------------------------------------------------------------------------ */
int am_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int ld, float mul,
                    am_stream_t stream);
/* Image boundary for the 3-channel first layers: NCHW fp32 [B,C,H,W] (H, W even) -> space-to-depth(2) NHWC
 * [B,H/2,W/2,16] `dtype`, channel (py*2+px)*C + c = img[b,c,2Y+py,2X+px].  A stride-2 KxK conv on the image becomes a
 * stride-1 conv with 16 input channels on this tensor (7x7 -> 4x4 taps, 5x5 -> 3x3 taps). */
int am_image_s2d(int dtype, const float* src, void* dst, int B, int C, int H, int W, am_stream_t stream);
/* The same boundary for raw frames: uint8 NCHW [B,C,H,W] -> (u/255 - mean[c]) / std[c] computed in fp32 exactly as the
 * reference's loader does (dataloaders/bdd_detection_loader.py:54 `read_image(...).float() / 255.0`; optional torchvision
 * Normalize, train_bdd100k_ddp.py:471-473) -> space-to-depth NHWC `dtype` [B,ceil(H/2),ceil(W/2),16] (odd sizes: the missing
 * row / column is zero).  mean/std are HOST arrays of C floats, both NULL for /255 only. */
int am_image_u8_s2d(int dtype, const uint8_t* src, void* dst, int B, int C, int H, int W, const float* mean,
                    const float* stdv, am_stream_t stream);
int am_nhwc_to_nchw(int dtype, const void* src, float* dst, int B, int C, int H, int W, int ld, float mul,
                    am_stream_t stream);
int am_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* argmax, int B, int IH, int IW, int C,
                        am_stream_t stream);
/* BatchNorm (scale / shift from am_bn_finalize) + ReLU + MaxPool2d(3,2,1) of a raw conv output in one pass (the ResNet stem of a
 * TRAINABLE expert: torchvision resnet.py conv1 -> bn1 -> relu -> maxpool): y = pool(relu(x * scale[c] + shift[c])) with the
 * activation rounded to `dtype` before the comparison, exactly what am_bn_apply + am_maxpool3x3s2_fwd give, without writing or
 * re-reading the normalised map.  argmax as am_maxpool3x3s2_fwd (its backward is am_maxpool3x3s2_bwd; the BatchNorm backward then
 * takes the ReLU mask from the sign of x * scale + shift: am_bn_bwd_*_sign). */
int am_bn_relu_maxpool3x3s2_fwd(int dtype, const void* x, const float* scale, const float* shift, void* y, uint8_t* argmax,
                                int B, int IH, int IW, int C, am_stream_t stream);
int am_maxpool3x3s2_bwd(int dtype, const void* dy, const uint8_t* argmax, void* dx, int B, int IH, int IW, int C,
                        am_stream_t stream);
/* am_maxpool3x3s2_bwd for a pooled conv -> BatchNorm -> ReLU layer (the ResNet stem), fused with that BatchNorm's backward reduce pass:
 * while a pixel's gradient is in registers the kernel reads the raw conv output there and accumulates `sums` (the layout
 * am_bn_bwd_reduce fills, zeroed by the caller; ReLU mask = sign of raw * scale + shift as am_bn_bwd_reduce_sign), so that pass over
 * the two full-resolution tensors is not run.  AM_ERR_UNSUPPORTED when the channel count does not give every thread a fixed
 * 16-byte channel chunk (caller: am_maxpool3x3s2_bwd + am_bn_bwd_reduce_sign). */
int am_maxpool3x3s2_bwd_bn(int dtype, const void* dy, const uint8_t* argmax, void* dx, int B, int IH, int IW, int C,
                           const void* raw, const float* mean, const float* rstd, const float* scale, const float* shift,
                           double* sums, am_stream_t stream);
int am_gap_nhwc_fwd(int dtype, const void* x, int ld, float* out, int B, int P, int C, am_stream_t stream);
int am_gap_nhwc_bwd(int dtype, const float* dout, void* dx, int ld, int B, int P, int C, float mul,
                    am_stream_t stream);
int am_gap_plane_fwd(const float* x, float* out, long long planes, long long HW, am_stream_t stream);
int am_gap_plane_bwd(const float* dout, float* dx, long long planes, long long HW, am_stream_t stream);
int am_bilinear_up_fwd(int dtype, const void* low, int ld, float* out, int B, int C, int h, int w, int H, int W,
                       am_stream_t stream);
int am_bilinear_up_bwd(int dtype, const float* dout, void* dlow, int ld, int B, int C, int h, int w, int H, int W,
                       float mul, const float* dev_scale, am_stream_t stream);
/* Extractor seam fused: AdaptiveAvgPool2d(1)(F.interpolate(low)) == sum_{y,x} cy[y]*cx[x]*low[b,y,x,c] with
 * cy/cx = am_bilinear_colsum (column sums of the separable interpolation weights / output size); the full-resolution
 * logits (70 MB/img for the segmentation expert) are never written.  bwd: dlow = g[b,c]*cy[y]*cx[x]*mul. */
int am_bilinear_colsum(float* coef, int in_size, int out_size, am_stream_t stream);
int am_upsample_gap_fwd(int dtype, const void* low, int ld, const float* cy, const float* cx, float* out, int B, int C,
                        int h, int w, am_stream_t stream);
int am_upsample_gap_bwd(int dtype, const float* g, const float* cy, const float* cx, void* dlow, int ld, int B, int C,
                        int h, int w, float mul, am_stream_t stream);
int am_ce2d_fwd(const float* logits, const long long* target, int B, int C, long long HW, long long ignore_index,
                double* acc2, am_stream_t stream);
int am_ce2d_bwd(const float* logits, const long long* target, int B, int C, long long HW, long long ignore_index,
                const double* acc2, const float* grad_out, float* dlogits, am_stream_t stream);

/* Dense-expert training loss fused (bdd_segmentation_expert.py:22 F.interpolate + train_bdd100k_ddp.py:89-100 CrossEntropyLoss(
 * ignore_index=255)): loss sum / valid count (acc2, as am_ce2d_fwd) and the UNNORMALISED gradient with respect to the
 * low-resolution NHWC logits, G[B,h,w,C] fp32 = sum over output pixels of (softmax - onehot) * interpolation weight, in one pass
 * over the labels; the [B,C,H,W] logits are never written.  Deterministic (every G cell has one owner, fixed summation order).
 * C = 3 or 19 (AM_ERR_UNSUPPORTED otherwise: caller uses am_bilinear_up_* + am_ce2d_*).
 * bwd: dlow[b,y,x,c] = G * grad_out[0] / max(count, 1) * mul in `dtype` at pixel stride ld. */
int am_upsample_ce2d_fwd(int dtype, const void* low, int ld, const long long* target, int B, int C, int h, int w, int H, int W,
                         long long ignore_index, double* acc2, float* G, am_stream_t stream);
int am_upsample_ce2d_bwd(int dtype, const float* G, const double* acc2, const float* grad_out, float mul, void* dlow, int ld,
                         int B, int C, int h, int w, am_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * MoE tail (fp32): nn.Linear / ReLU / Dropout / LayerNorm of expert_extractors.py:30-34,
 * context_features.py:143-149, gating_network.py:13-20,38-44,93-100,168, trajectory_head.py:26,44-53,
 * and the gate (gating_network.py:149-166): weights = softmax(logits/T) or sigmoid-normalise, with
 * optional top-k masking; combined = sum_e w[:,e] * P_e.
 * Linear: W is [N][K] row-major (nn.Linear.weight); `yact` (optional) is the post-ReLU output whose
 * sign masks dy in the backward kernels; bwd_weight ACCUMULATES into dW / dbias.
 * ------------------------------------------------------------------------------------------ */
#define AM_MAX_EXPERTS 8
int am_linear_fwd(const float* x, int ldx, const float* W, const float* bias, float* y, int ldy, int M, int N,
                  int K, int relu, am_stream_t stream);
int am_linear_bwd_input(const float* dy, int lddy, const float* yact, int ldya, const float* W, float* dx, int lddx,
                        int M, int N, int K, int accumulate, am_stream_t stream);
int am_linear_bwd_weight(const float* dy, int lddy, const float* yact, int ldya, const float* x, int ldx, float* dW,
                         float* dbias, int M, int N, int K, am_stream_t stream);
int am_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* y, int ldy,
                     float* mean, float* rstd, int M, int D, am_stream_t stream);
int am_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* mean,
                     const float* rstd, float* dx, int lddx, float* dgamma, float* dbeta, int M, int D,
                     am_stream_t stream);
int am_gate_combine_fwd(const float* logits, const float* const* processed, int E, int ldp, float temperature,
                        int use_softmax, int top_k, float* weights, float* combined, int B, int D,
                        am_stream_t stream);
int am_gate_combine_bwd(const float* logits, const float* const* processed, int E, int ldp, float temperature,
                        int use_softmax, int top_k, const float* dcombined, const float* dweights_ext,
                        float* dlogits, float* const* dprocessed, int B, int D, am_stream_t stream);
/* dev_step (optional device int64): mixed into the seed on the device, so a captured hipGraph draws a fresh mask per replay */
int am_dropout_fwd(const float* x, float* y, uint8_t* mask, long long n, float p, unsigned long long seed,
                   const long long* dev_step, am_stream_t stream);
int am_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, long long n, float p, am_stream_t stream);

/* Grouped forms ("the MoE tail in a handful of launches").  The tail of AutoMoE.forward (automoe.py:189-233 ->
 * expert_extractors.py:30-34, context_features.py:143-149, gating_network.py:13-20,38-44,93-100,168, trajectory_head.py:44-53) is
 * ~13 DEPENDENT stages, each with 2-5 INDEPENDENT branches: one extractor MLP / output processor per expert, the context
 * encoders, the two policy heads.  One call = one launch for all branches of a stage (backward: two launches -- input
 * gradients, parameter gradients); the dependent stages stay separate launches (a cut at every all-to-all seam).
 *   linear:    y = dropout_p(relu?(x W^T + b)); the reference's Linear -> ReLU -> Dropout triples run as one epilogue.  The dropout
 *              mask is the counter-based hash of am_dropout_fwd on (seed, *dev_step, m*N + n); it is not stored: y == 0 <=> dropped
 *              or rectified, so backward takes dz = dy * gscale * (yact > 0) with yact = y and gscale = 1 / (1 - p).
 *   backward:  dx (= or +=, dx_accumulate) = dz W;  dW += dz^T x;  dbias += colsum(dz).  NULL dx / dW skip that gradient.
 *   layernorm: am_layernorm_fwd / _bwd per member (dgamma / dbeta accumulate).
 * count <= AM_TAIL_MAX_GROUP members, all with the same row count M (the batch). */
#define AM_TAIL_MAX_GROUP 8
typedef struct am_tail_linear {
  const float* x; const float* W; const float* bias; float* y;                 /* forward */
  const float* dy; const float* yact; float* dx; float* dW; float* dbias;      /* backward (yact, dx, dW, dbias may be NULL) */
  int32_t ldx, ldy, lddy, ldya, lddx;
  int32_t N, K, relu, dx_accumulate;
  float drop_p, gscale;
  uint64_t seed;
} am_tail_linear;
typedef struct am_tail_layernorm {
  const float* x; const float* gamma; const float* beta; float* y; float* mean; float* rstd;   /* forward (mean / rstd saved) */
  const float* dy; float* dx; float* dgamma; float* dbeta;                                     /* backward (dx or dgamma+dbeta may be NULL) */
  int32_t ldx, ldy, lddy, lddx, D;
  float eps;
} am_tail_layernorm;
int am_moe_tail_linear_fwd(const am_tail_linear* group, int count, int M, const long long* dev_step, am_stream_t stream);
int am_moe_tail_linear_bwd(const am_tail_linear* group, int count, int M, am_stream_t stream);
int am_moe_tail_layernorm_fwd(const am_tail_layernorm* group, int count, int M, am_stream_t stream);
int am_moe_tail_layernorm_bwd(const am_tail_layernorm* group, int count, int M, am_stream_t stream);

/* Gating-stage objective (training/train_gating_network.py:21-76 compute_gating_losses) and its gradient in one launch:
 * parts6 = {ade, fde, speed, smoothness, load_balancing, entropy}, total = coef6 . parts6 (one float);
 * g_* = d total / d input (dense).
 * coef6 is a HOST array.  wp/twp [B,T,2] dense; spd/tspd [B,S] with row strides ld_* (S = 0: no speed term); w [B,E], E <= 64. */
int am_gating_losses(const float* wp, const float* twp, int B, int T, const float* spd, const float* tspd, int ld_spd,
                     int ld_tspd, int S, const float* w, int E, const float* coef6, int use_lb, int use_ent,
                     float* total, float* parts6, float* g_wp, float* g_spd, float* g_w, am_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Hungarian matcher (training/hungarian_matcher.py:20-85).
 *   am_match_cost    per image b and query q: C = w_bbox*L1(cxcywh) + w_class*(-softmax prob at the
 *                    GT label) + w_giou*(-GIoU), summed in that order in fp32; written TRANSPOSED,
 *                    cost[b][j][q] (j < n_tgt[b], leading dims Nmax and Q), so the solver reads rows.
 *   am_lsap_batched  scipy.optimize.linear_sum_assignment per image, fp32 costs promoted to fp64:
 *                    cost(i,j) at cost[b*batch_stride + i*row_stride + j*col_stride], i < nr,
 *                    j < nc_per[b].  Outputs int64 pairs sorted by row (row_idx/col_idx [B][kmax]),
 *                    count[b] = min(nr, nc), status[b] = 0 ok, -1 infeasible, -2 NaN/-inf entry
 *                    (scipy raises ValueError for both).  Bit-exact with scipy, ties included.
 * ------------------------------------------------------------------------------------------ */
int am_match_cost(const float* logits, const float* boxes, const int64_t* tgt_labels, const float* tgt_boxes,
                  const int32_t* n_tgt, int B, int Q, int C, int Nmax, float w_class, float w_bbox, float w_giou,
                  float* cost, am_stream_t stream);
/* am_match_cost for boxes of D numbers (hungarian_matcher.py:47-68): D = 4 as above; D = 7 = [cx,cy,cz,w,l,h,yaw] with the
 * L1 term over all seven and the axis-aligned BEV GIoU on (cx -+ w/2, cy -+ l/2); any other D: no GIoU term. */
int am_match_cost_d(const float* logits, const float* boxes, int D, const int64_t* tgt_labels, const float* tgt_boxes,
                    const int32_t* n_tgt, int B, int Q, int C, int Nmax, float w_class, float w_bbox, float w_giou,
                    float* cost, am_stream_t stream);
int am_lsap_batched(const float* cost, int B, int nr, const int32_t* nc_per, int nc_max, long long batch_stride,
                    long long row_stride, long long col_stride, int64_t* row_idx, int64_t* col_idx, int kmax,
                    int32_t* count, int32_t* status, am_stream_t stream);
/* am_lsap_batched with a caller-owned workspace (am_lsap_batched_workspace_bytes; 0 bytes: the plain form runs): problems whose
 * short side has at most 32 rows and whose long side at most 1024 (the detection loss, hungarian_matcher.py:76-82: <= 32 boxes
 * against 920 queries) are solved by the split solver -- per-row sorted candidate lists made on the whole chip, then one wave per
 * image with the <= 32 assigned columns in registers (lsap.hip) -- and any image in which a tie between candidates could matter,
 * or which is infeasible, is solved again by the general kernel in the same call, so the results are those of am_lsap_batched
 * (bit-exact with scipy, ties included).  The workspace is scratch: nothing in it survives the call. */
int am_lsap_batched_workspace_bytes(int B, int nr, int nc_max, long long* bytes);
int am_lsap_batched_ws(const float* cost, int B, int nr, const int32_t* nc_per, int nc_max, long long batch_stride,
                       long long row_stride, long long col_stride, int64_t* row_idx, int64_t* col_idx, int kmax,
                       int32_t* count, int32_t* status, void* workspace, long long workspace_bytes, am_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Step glue on flat fp32 buffers (train_bdd100k_ddp.py:98-99, train_gating_network.py:103-105):
 * global grad norm kept on the device, clip_grad_norm_(max_norm) folded into the AdamW update
 * (torch.optim.AdamW arithmetic), non-finite norm => step skipped and counted.
 * ------------------------------------------------------------------------------------------ */
int am_sumsq_accumulate(const float* x, long long n, double* acc, am_stream_t stream);
/* hyper-parameters as doubles: torch.optim.AdamW derives 1 - beta, lr / (1 - beta1^t), 1 - lr*wd in double and rounds each to
 * fp32 once; doing the same keeps the moments interchangeable with a torch optimizer's state (checkpoint resume both ways). */
int am_adamw_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int step, float max_norm, const double* norm_sq, int* skipped,
                  am_stream_t stream);
int am_scale_inplace(float* x, long long n, float mul, const double* denom, am_stream_t stream);

/* Gradient exchange -- SURVEY 8(b) sketched an `am_allreduce_bucket` (ncclAllReduce behind an event, on a side stream).  It is
 * deliberately NOT an entry of this library: the RCCL communicator belongs to torch.distributed's process group (c10d
 * ProcessGroupNCCL, the transport of the reference's DistributedDataParallel, train_bdd100k_ddp.py:497); c10d does not hand out
 * its ncclComm_t, and a second communicator here would duplicate RCCL's bootstrap, buffers and streams.  What the wrapper would
 * have contained -- record an event on the compute stream when a bucket of the flat gradient buffer is complete, wait for it
 * and all-reduce the bucket on a side stream, join before am_adamw_step -- is training/ddp.py on dist.all_reduce(async_op=True);
 * with the "nccl" backend the whole sequence is captured into the step's hipGraph (tests/test_hip_multigpu.py). */

/* Packed conv operands from the fp32 master weight (hip/conv.py pack_fwd / pack_dgrad layouts; the reference keeps plain
 * OIHW nn.Conv2d weights, e.g. models/policy/trajectory_head.py:10-22): dst[i] = idx[i] < 0 ? 0 : (dtype)src[idx[i]].
 * One launch rebuilds every layout of a layer (or of a whole model) after an optimizer step.  idx and dst 16-byte aligned
 * (AM_ERR_ARG otherwise): the kernel works in vectors of eight elements. */
int am_gather_cast(int dtype, const float* src, const int* idx, void* dst, long long n, am_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AUTOMOE_HIP_H */
