/* deadtrees_hip.h — C ABI of libdeadtrees_hip.so (MI355X / gfx950 only).
 *
 * The reference (cwerner/deadtrees) has NO native code and no FFI: its hot path is
 * `self.model(img)` (deadtrees/network/segmodel.py:214,235,280; deployment/inference.py:60) on
 * smp.Unet(resnet34) plus the loss callables of deadtrees/loss (segmodel.py:169-200), executed
 * by ATen.  Each entry point below replaces one ATen op family that this path launches
 * (SURVEY.md §2.3 K1..K22); the comment on each names the reference call site it serves.
 *
 * Conventions
 *   - plain pointers + sizes; all tensors are DEVICE pointers, fp32 unless stated, NHWC
 *     ("[B,H,W,C]") for activations, HWIO ("[kh][kw][Cin][Cout]") for conv weights;
 *     logits / labels / distance maps at the module boundary are NCHW like the reference.
 *   - inputs are borrowed, outputs are caller-allocated, no ownership transfer, no hidden
 *     allocation, no host synchronisation: every call only enqueues kernels on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream) and is hipGraph-capturable.
 *   - return 0 on success, a negative DT_E* code otherwise; dt_last_error() gives the message
 *     (thread-local).  Shape preconditions are validated on the host BEFORE any launch.
 */
#ifndef DEADTREES_HIP_H
#define DEADTREES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DT_OK 0
#define DT_EINVAL (-22)  /* bad shape / argument            */
#define DT_ENOSYS (-38)  /* configuration not implemented   */
#define DT_EHIP (-5)     /* HIP runtime / launch failure    */

const char* dt_last_error(void);
int dt_version(void);
/* number of HIP devices visible; <0 on error.  Does not create a context. */
int dt_device_count(void);

/* ------------------------------------------------------------------ convolution (K1,K5,K6,K7,K9,K10,K21)
 * One descriptor drives forward, data-gradient and weight-gradient kernels.
 * The LOGICAL input is cat([src0', src1], channel) of size [B,Hin,Win,C0+C1] where src0' is
 *   mode0 = 0 : src0 itself                                   [B,Hin,Win,C0]
 *   mode0 = 1 : nearest x2 upsample of src0                   [B,Hin/2,Win/2,C0]  (F.interpolate, K9)
 *   mode0 = 2 : zero-insertion x2 of src0 (transposed conv)   [B,Hin/2,Win/2,C0]  (stride-2 dgrad)
 * src1 (C1 may be 0) is always direct (the U-Net skip, torch.cat K10).
 * Output [B,Ho,Wo,Cout]; channels [0,cout_split) go to out0 (leading dim cout_split), the rest to
 * out1 (leading dim Cout-cout_split); cout_split = 0 -> everything to out0.
 */
typedef struct dt_conv_desc {
  int32_t B, Hin, Win;
  int32_t C0, C1;
  int32_t mode0;
  int32_t Ho, Wo, Cout;
  int32_t ksize;       /* 1, 3 or 7 (square) */
  int32_t stride;      /* 1 or 2             */
  int32_t pad;
  int32_t cout_split;  /* 0 or a multiple of 32 */
  int32_t accumulate;  /* !=0: out0 += result (gradient accumulation on residual / skip joins) */
} dt_conv_desc;

/* rows P of the BatchNorm partial-statistics buffer dt_conv2d writes: stats is [2][P][Cout]
 * (plane 0 = sum, plane 1 = sum of squares, one row per workgroup tile). */
int dt_conv2d_stat_rows(const dt_conv_desc* d);

/* y = conv(x, w) — replaces ATen conv2d reached from smp encoder/decoder (segmodel.py:214).
 * stats may be NULL.  fp32 MFMA (v_mfma_f32_32x32x2_f32): exact fp32 fma chains. */
int dt_conv2d(const dt_conv_desc* d, const float* src0, const float* src1, const float* w_hwio,
              float* out0, float* out1, float* stats, const float* in_scale, const float* in_shift, void* stream);
/* in_scale/in_shift (both NULL or both [C0]): source 0 is read as relu(src0*in_scale[c]+in_shift[c]) while it is
 * staged into LDS — the BatchNorm-apply + ReLU of the producing layer fused into its consumer, so that activation
 * never exists in HBM (zero padding stays zero).  Same pair on dt_conv2d_wgrad. */

/* tile configuration dt_conv2d selects for d (kernel conv_fwd_kernel<ksize,stride,tw,tn,ck>): used to
 * attribute profiler rows and roofline numbers to launches. */
int dt_conv2d_config(const dt_conv_desc* d, int* tw, int* tn, int* ck);
int dt_conv2d_uses_zi(const dt_conv_desc* d);  /* 1: parity-class tiles of the transposed (stride-2) data gradient */

/* ---- Winograd F(2x2,3x3) form of the 3x3 stride-1 pad-1 layers (same ATen conv2d / convolution_backward(input)
 * calls as dt_conv2d; 2.25x fewer multiplies, results within ~1e-6 relative of the direct form instead of an exact fma
 * chain).  u = dt_winograd_weights(w_hwio): G g G^T in the order [16 positions][Cin/8][2][Cout][4], 16*Cin*Cout floats.
 * Supported when C0, C1 are multiples of 8 and C0 + C1 of 16, Cout and cout_split multiples of 64, mode0 in {0,1}, every
 * operand below 2 GiB (dt_conv2d_winograd_supported; callers fall back to dt_conv2d otherwise).
 * stats rows: dt_conv2d_winograd_stat_rows (16x16-pixel tiles). */
int dt_conv2d_winograd_supported(const dt_conv_desc* d);
int dt_conv2d_winograd_stat_rows(const dt_conv_desc* d);
int dt_winograd_weights(const float* w_hwio, float* u, int Cin, int Cout, void* stream);
int dt_conv2d_winograd(const dt_conv_desc* d, const float* src0, const float* src1, const float* u, float* out0,
                       float* out1, float* stats, const float* in_scale, const float* in_shift, void* stream);
/* inference form (eval-mode BatchNorm): out = relu(conv(src) * scale + shift [+ residual]) — the ATen chain conv2d ->
 * batch_norm(eval) -> (add) -> relu_ of a ResNet / decoder block (reference call site deployment/inference.py:60) in ONE
 * kernel; scale / shift per output channel (dt_bn_eval_affine), residual (optional) in the layout of out.  Same
 * arithmetic as dt_conv2d_winograd followed by dt_bn_act: bit-identical.  No split outputs, joins or statistics. */
int dt_conv2d_winograd_affine(const dt_conv_desc* d, const float* src0, const float* src1, const float* u, float* out,
                              const float* scale, const float* shift, const float* residual, void* stream);
/* The same one-launch inference form for the narrow full-resolution decoder layers (3x3 stride 1 pad 1, Cin and Cout in
 * {16, 32}, maps of at least 8 x 32 pixels; dt_conv2d_narrow_supported tells): out = relu(conv(x) * scale + shift), with
 * x = src0 or — in_scale / in_shift given — relu(src0 * in_scale + in_shift) (the producer's virtual activation). */
int dt_conv2d_narrow_supported(const dt_conv_desc* d);
int dt_conv2d_narrow_affine(const dt_conv_desc* d, const float* src0, const float* w_hwio, float* out, const float* scale,
                            const float* shift, const float* in_scale, const float* in_shift, void* stream);
/* ... and for every other layer dt_conv2d takes (stem 7x7 / 2, stride-2 3x3, 1x1 down-sample, concat layers): out =
 * conv(x) * scale + shift, through ReLU when relu != 0 — bit-identical to dt_conv2d followed by dt_bn_act(relu).  No
 * input transform, split, join or statistics. */
int dt_conv2d_affine(const dt_conv_desc* d, const float* src0, const float* src1, const float* w_hwio, float* out,
                     const float* scale, const float* shift, int relu, void* stream);
/* all eligible layers in one launch: int32 table rows (w_off, u_off, Cin, Cout, first_block), blocks of a layer =
 * ceil(Cout/64) * ceil(Cin/16); `weights` = the flat parameter buffer (forward images) or its dt_weight_images mode-0
 * image with Cin/Cout swapped (data-gradient images). */
int dt_winograd_weight_images(const float* weights, float* u, const int32_t* table, int n_layers, int total_blocks,
                              void* stream);

/* wd[kh'][kw'][co][ci] = w[K-1-kh'][K-1-kw'][ci][co]: weights of the data-gradient convolution. */
int dt_weight_flip_transpose(const float* w_hwio, float* wd, int ksize, int Cin, int Cout, void* stream);

/* dW[kh][kw][ci][co] = sum_{b,oy,ox} x[b,oy*s+kh-p,ox*s+kw-p,ci] * dy[b,oy,ox,co]  (autograd of conv2d, K21).
 * workspace: fp32 scratch of at least dt_conv2d_wgrad_workspace(d) bytes (split-K partials, reduced
 * in a fixed order -> run-to-run deterministic). */
size_t dt_conv2d_wgrad_workspace(const dt_conv_desc* d);
int dt_conv2d_wgrad(const dt_conv_desc* d, const float* src0, const float* src1, const float* dy,
                    float* dw_hwio, float* workspace, size_t workspace_bytes, const float* in_scale,
                    const float* in_shift, void* stream);
/* Winograd F(2x2,3x3) form of dt_conv2d_wgrad for 3x3 stride-1 pad-1 layers whose C0, C1 and Cout are multiples of 64
 * (same arguments; 2.25x fewer multiplies, result within ~1e-6 relative of the direct form; deterministic). */
int dt_conv2d_wgrad_winograd_supported(const dt_conv_desc* d);
size_t dt_conv2d_wgrad_winograd_workspace(const dt_conv_desc* d);
int dt_conv2d_wgrad_winograd(const dt_conv_desc* d, const float* src0, const float* src1, const float* dy,
                             float* dw_hwio, float* workspace, size_t workspace_bytes, const float* in_scale,
                             const float* in_shift, void* stream);

/* ------------------------------------------------------------------ BatchNorm / ReLU / residual (K2,K3,K8,K21) */
/* stats[2][P][C] -> batch mean / biased var over `count` elements; writes mean, invstd, and the fused
 * affine scale = gamma*invstd, shift = beta - mean*scale; updates running stats with `momentum`
 * (unbiased var), as torch BatchNorm2d(train).  fp64 accumulation, fixed order. */
/* The stats buffer (also the one dt_conv2d writes) must hold dt_bn_stats_floats(P,C) floats: the
 * [2][P][C] partial rows plus a scratch tail used by the first reduction stage when P is large. */
int64_t dt_bn_stats_floats(int P, int C);
int dt_bn_finalize(float* stats, int P, int C, double count, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var,
                   float* mean, float* invstd, float* scale, float* shift, void* stream);
/* eval mode: scale/shift from running stats. */
int dt_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int C, float* scale, float* shift, void* stream);
/* eval mode with a backward pass to follow (frozen-BatchNorm fine-tuning, F.batch_norm(training=False) backward):
 * mean = running_mean, invstd = 1/sqrt(running_var + eps) for dt_bn_bwd_reduce / dt_bn_bwd_apply_frozen. */
int dt_bn_eval_stats(const float* running_mean, const float* running_var, float eps, int C, float* mean,
                     float* invstd, void* stream);
/* out = act( y*scale[c]+shift[c] + (res ? res*rscale[c]+rshift[c] : 0) ), act = ReLU if relu == 1;
 * relu == 2: ReLU on the main branch only, out = relu(y*scale+shift) + residual (the ResUnet decoder block,
 * network/extra/resunet/decoder.py:40-52).  rscale/rshift NULL -> identity residual.  n_pix = B*H*W. */
int dt_bn_act(const float* y, const float* scale, const float* shift, const float* res,
              const float* rscale, const float* rshift, float* out, int64_t n_pix, int C, int relu,
              void* stream);
/* BN backward, pass 1: g = dout * (out>0 if out_act else 1); partial sums of g and g*xhat per channel
 * -> red[2][P][C] with P = dt_bn_bwd_rows(n_pix); `red` holds dt_bn_bwd_red_floats(n_pix,C) floats. */
int dt_bn_bwd_rows(int64_t n_pix, int C);
int64_t dt_bn_bwd_red_floats(int64_t n_pix, int C);  /* size of `red` in floats (rows + reduction scratch) */
/* ReLU mask of g: from out_act when given, else recomputed as (y*act_scale+act_shift > 0) when act_scale is
 * given (virtual activation), else none (BatchNorm without ReLU: the downsample branch). */
int dt_bn_bwd_reduce(const float* dout, const float* out_act, const float* y, const float* mean,
                     const float* invstd, const float* act_scale, const float* act_shift, float* red,
                     int64_t n_pix, int C, void* stream);
/* pass 2: reduces red -> dgamma, dbeta (fp64, fixed order) and writes
 * dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat));  if dres != NULL also dres (+)= g. */
int dt_bn_bwd_apply(const float* dout, const float* out_act, const float* y, const float* mean,
                    const float* invstd, const float* gamma, const float* act_scale, const float* act_shift,
                    float* red, int P, float* dgamma, float* dbeta, float* dy, float* dres, int dres_accumulate,
                    int64_t n_pix, int C, void* stream);
/* the same pass for eval-mode (frozen-statistics) BatchNorm: dgamma = sum g*xhat, dbeta = sum g as above, but
 * dy = gamma*invstd*g (mean / invstd do not depend on the batch) — ATen batch_norm_backward with train=False. */
int dt_bn_bwd_apply_frozen(const float* dout, const float* out_act, const float* y, const float* mean,
                           const float* invstd, const float* gamma, const float* act_scale, const float* act_shift,
                           float* red, int P, float* dgamma, float* dbeta, float* dy, float* dres,
                           int dres_accumulate, int64_t n_pix, int C, void* stream);

/* torch.cat of NHWC tensors along the channels and its backward (the dense skip connections of the Unet++ decoder:
 * reference network/extra/efficientunetplusplus/decoder.py:170-177, same wiring as smp UnetPlusPlusDecoder).
 * to_wide = 1: wide[n, offset : offset + C_narrow] = narrow[n, :] (src = narrow, dst = wide);
 * to_wide = 0: narrow[n, :] (+)= wide[n, offset : offset + C_narrow] (src = wide, dst = narrow; += when accumulate). */
int dt_channel_slice(const float* src, float* dst, int64_t n_pix, int C_narrow, int C_wide, int offset, int to_wide,
                     int accumulate, void* stream);

/* per-channel sums of g [n_pix][C] -> out[C]: the bias gradient of a biased convolution (ATen convolution_backward's
 * third output; here the 1x1 identity_conv of the ResUnet decoder).  workspace: dt_channel_sums_workspace floats. */
int64_t dt_channel_sums_workspace(int64_t n_pix, int C);
int dt_channel_sums(const float* g, float* workspace, int64_t n_pix, int C, float* out, void* stream);
/* bf16 twin (g bf16, fp32 sums; C a multiple of 8, at most 256): the identity_conv bias gradient under AMP */
int64_t dt_channel_sums_bf16_workspace(int64_t n_pix, int C);
int dt_channel_sums_bf16(const void* g_bf16, float* workspace, int64_t n_pix, int C, float* out, void* stream);

/* ------------------------------------------------------------------ pooling / resampling (K4,K9 bwd) */
/* max_pool2d(k=3,s=2,p=1) NHWC; argmax (uint8 window position, first max in scan order like ATen). */
int dt_maxpool3x3s2(const float* x, float* out, uint8_t* argmax, int B, int H, int W, int C, void* stream);
int dt_maxpool3x3s2_bwd(const float* dout, const uint8_t* argmax, float* dx, int accumulate, int B,
                        int H, int W, int C, void* stream);
/* backward of nearest x2 upsample: dx[b,y,x,c] = sum of the 2x2 block of dup; dup is [B,2H,2W,C]. */
int dt_upsample2x_bwd(const float* dup, float* dx, int accumulate, int B, int H, int W, int C, void* stream);
/* layout shuttles at the module boundary */
int dt_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, void* stream);
int dt_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, void* stream);
/* uint8 HWC tile -> normalised fp32 NHWC ((x/255-mean)/std), reference data/deadtreedata.py:148-154 */
/* mean/std are HOST arrays of Cdst floats (passed by value to the kernel). */
int dt_normalize_u8(const uint8_t* src, float* dst, int64_t n_pix, int Csrc, int Cdst, const float* mean,
                    const float* std, void* stream);
/* Tiled-inference input in one gather (reference deployment/tiler.py:121-134 zero-pad + utils/data_handling.py:9-20
 * make_blocks_vectorized + scripts/inference.py:94-96 per-tile Normalize): band-major uint8 raster [Csrc][h][w] ->
 * fp32 NHWC sub-tiles [n_blocks][d][d][Cdst] for the blocks first_block .. of the row-major block grid that is nbx blocks
 * wide; pixels beyond the raster are the tiler's zero padding (normalised like a zero byte). */
int dt_split_normalize_u8(const uint8_t* raster_chw, float* dst_nhwc, int Csrc, int h, int w, int d, int nbx,
                          int first_block, int n_blocks, int Cdst, const float* mean, const float* stdv, void* stream);
/* scripts/inference.py:60-62 is_valid_tile: flag[0] (int32, zero it first) = 1 iff some byte of band[n] is neither 0 nor
 * 255 (a raster whose first band is all 0 / 255 is skipped by the reference's driver). */
int dt_band_has_data(const uint8_t* band, int64_t n, int32_t* flag, void* stream);

/* Training augmentation on the device, data/deadtreedata.py:128-146 `train_transform`:
 * OneOf(HorizontalFlip, VerticalFlip), RandomRotate90, RandomBrightnessContrast(brightness_by_max=False),
 * Normalize, ToTensorV2 in ONE gather pass uint8 NHWC [B,H,W,Csrc] -> fp32 NHWC [B,H,W,Cdst].
 * The host draws the per-sample parameters: geo int32 [B][2] = (flip 0 none / 1 horizontal / 2 vertical,
 * rot k = np.rot90 count, H == W when k is odd), bc fp32 [B][2] = (alpha, beta); (1, 0) leaves the pixel values
 * unchanged, else lut(v) = uint8(clip(v*alpha + beta*mean(image), 0, 255)) as albumentations does for uint8.
 * sums_scratch: B uint64 (per-image pixel sums for the mean).  mean/std: HOST arrays of Cdst floats.
 * dt_augment_labels applies the same geometric map to int64 [B,H,W] masks / land-use maps. */
int dt_augment_normalize_u8(const uint8_t* src, float* dst, const int32_t* geo, const float* bc, uint64_t* sums_scratch,
                            int B, int H, int W, int Csrc, int Cdst, const float* mean, const float* std, void* stream);
int dt_augment_labels(const int64_t* src, int64_t* dst, const int32_t* geo, int B, int H, int W, void* stream);

/* Data gradient of a 3x3 stride-1 layer with the BatchNorm-backward REDUCTION of the layer it feeds fused into the
 * epilogue (instead of a separate dt_bn_bwd_reduce pass over the tensor it has just written): out0 = conv(src0, w)
 * like dt_conv2d (no concat / split / accumulate / upsample), and red[2][P][Cout] (P = dt_conv2d_stat_rows(desc))
 * receives per workgroup  sum g  and  sum g * xhat  with  g = out0 * [y*act_scale + act_shift > 0],
 * xhat = (y - mean) * invstd  — exactly what dt_bn_bwd_reduce computes from (dout = out0, y) with a virtual
 * activation.  `red` must hold dt_bn_stats_floats(P, Cout) floats; pass it with P to dt_bn_bwd_apply. */
typedef struct dt_bn_bwd_fuse {
  const float* y;          /* raw conv output of the BatchNorm layer, same shape as out0 */
  const float* mean;
  const float* invstd;
  const float* act_scale;  /* scale / shift of that BatchNorm (its ReLU mask is recomputed from y) */
  const float* act_shift;
  const float* act;        /* optional: the STORED activation (block outputs, relu(bn(y) + identity)); when given the
                              mask is act > 0 and act_scale / act_shift are not read.  With it the convolution may be a
                              gradient join (desc->accumulate = 1): the sums are taken over out0 AFTER the add. */
} dt_bn_bwd_fuse;
int dt_conv2d_bn_bwd(const dt_conv_desc* desc, const float* src0, const float* w, float* out0, float* red,
                     const dt_bn_bwd_fuse* fuse, void* stream);
/* Winograd form of dt_conv2d_bn_bwd: same arguments, u = dt_winograd_weights of the flipped / transposed weights. */
int dt_conv2d_winograd_bn_bwd(const dt_conv_desc* d, const float* src0, const float* u, float* out0, float* red,
                              const dt_bn_bwd_fuse* fuse, void* stream);
/* Data gradient of a decoder conv1 — y = conv3x3(cat(nearest_upsample_x2(x), skip)) — with the up-sampling's backward in
 * the Winograd epilogue (the 2x2 outputs of a Winograd tile are one source pixel's four gradients): `desc` = the stride-1
 * data-gradient descriptor (C0 = the convolution's output channels, Cout = channels of x + skip, cout_split = channels of
 * x; multiples of 64), u = dt_winograd_weights of the flipped / transposed weights.  gx [B, Hin/2, Win/2, cout_split] =
 * gradient of x (2x2-summed, never stored at full resolution), red [2][P][cout_split] = BatchNorm-backward sums of the
 * layer that produced x (fuse: its raw output at gx's resolution, mean, invstd, act_scale, act_shift), dskip [B, Hin,
 * Win, Cout - cout_split] = gradient of the skip.  Replaces dt_conv2d_winograd (split outputs) + dt_upsample2x_bwd_bn
 * (reference: autograd of torch.cat + F.interpolate(nearest, x2) + Conv2d in smp's DecoderBlock.forward). */
int dt_conv2d_winograd_upsampled_dgrad_supported(const dt_conv_desc* desc);
int dt_conv2d_winograd_upsampled_dgrad_rows(const dt_conv_desc* desc);
int dt_conv2d_winograd_upsampled_dgrad(const dt_conv_desc* desc, const float* dy, const float* u, float* gx, float* dskip,
                                       float* red, const dt_bn_bwd_fuse* fuse, int launches /* reserved: pass 3 */,
                                       void* stream);

/* ------------------------------------------------------------------ segmentation head (K11,K12,K19) */
/* logits[B,K,H,W] (NCHW) = conv3x3(x[B,H,W,Cin], w[K][3][3][Cin]) + bias; optional uint8/int64 argmax
 * class map (ties -> lowest index, torch.argmax) — smp SegmentationHead + inference.py:62. */
int dt_head_fwd(const float* x, const float* w_ohwi, const float* bias, float* logits_nchw,
                int64_t* argmax_i64, uint8_t* argmax_u8, int B, int H, int W, int Cin, int K, void* stream);
int dt_head_bwd_rows(int B, int H, int W);
/* floats the `red` scratch of dt_head_bwd / dt_head_bwd_finalize must hold (partial rows + reduction stage) */
int64_t dt_head_bwd_red_floats(int B, int H, int W, int Cin, int K);
/* dx[B,H,W,Cin] and one partial row of dW/dbias per workgroup into red; dt_head_bwd_finalize sums the rows
 * (two-stage, fixed order) into dw_ohwi[K][3][3][Cin] and dbias[K]. */
int dt_head_bwd(const float* x, const float* w_ohwi, const float* dlogits_nchw, float* dx, float* red,
                int B, int H, int W, int Cin, int K, void* stream);
int dt_head_bwd_finalize(float* red, int P, float* dw_ohwi, float* dbias, int Cin, int K, void* stream);

/* ------------------------------------------------------------------ losses & metrics (K12-K18) */
#define DT_LOSS_NACC 10
/* per (b,k) accumulators, fp64 [B][K][DT_LOSS_NACC]:
 *   0 count(t)  1 sum p*t  2 sum p  3 sum (1-p)^gamma * t * log(p+1e-10)  4 sum t*log(p+1e-10)
 *   5 sum p*dist  6 sum t*[p>0.5]  7 sum [p>0.5]
 *   8 sum t * wass, wass = sum_l M[label][l] * softmax(p)_l  (GWDICE, loss/gwdl.py:131-178; only when wass_m given:
 *     K x K label-distance matrix, row-major; note the SECOND softmax over the probabilities, segmodel.py:176)
 *   9 sum t * V(pixel), V = gw_possum from dt_gwdice_possum (the cross-sample sum the reference's broadcasting
 *     in gwdl.py:180-198 produces); wass_m and gw_possum are both NULL unless GWDICE is on.
 * from logits (softmax fused, never materialising the int32 one-hot of losses.py:124-141).
 * labels int64 [B,H,W]; dist may be NULL.  Also optionally writes probs[B,K,H,W].
 * acc must have room for dt_seg_loss_acc_doubles(B,K,H,W) doubles: the [B][K][NACC] result first,
 * per-workgroup partial rows behind it (fixed-order second stage, no atomics).
 * Labels outside [0,K) (the assert of losses.py:129) set err_flag[0]=1 instead of aborting the kernel. */
int64_t dt_seg_loss_acc_doubles(int B, int K, int H, int W);
int dt_seg_loss_fwd(const float* logits, const int64_t* labels, const float* dist, const float* wass_m,
                    const float* gw_possum, float gamma, double* acc, float* probs, int32_t* err_flag, int B, int K, int H, int W, void* stream);
/* dlogits = softmax-backward of g, g_k = a[b,k]*t_k + c[b,k] + wf*focal'(p_k)*t_k + wb[k]*dist_k,
 * scaled by gscale[0] (device scalar: upstream grad); coef fp32 [B][K][2] = (a,c); wfocal = [wf/M, gamma].
 * GWDICE (all NULL when unused): wass_m [K][K]; d loss / d wass of pixel (b, s) = wass_coef[b] + gw_posgrad[s]
 * (fp32 [B] and [H*W], the latter from dt_gwdice_posgrad); chained through the second softmax before it joins g. */
int dt_seg_loss_bwd(const float* logits, const int64_t* labels, const float* dist, const float* coef,
                    const float* wfocal, const float* wbound, const float* gscale, const float* wass_m,
                    const float* wass_coef, const float* gw_posgrad, float* dlogits, int B, int K, int H, int W,
                    void* stream);
/* The scalar algebra of SemSegment.calculate_loss / log_metrics (segmodel.py:169-208; gdl.py:15-27,
 * losses.py:187-291, gwdl.py:110-138, smp Fscore) on the [B][K][NACC] sums of dt_seg_loss_fwd, fp64, on the device:
 *   parts fp32 [8] = dice_loss, boundary_loss, focal_loss, ce_loss, dice (Fscore w/o background), dice_with_bg,
 *                    total_loss, total_loss (slot 7 = the differentiable scalar callers hand to backward);
 *   coef [B][K][2], wfocal [2], wbound [K] and (GWDICE) wass_a [B] (-> dt_gwdice_posgrad), wass_c [B]: the inputs of
 *   dt_seg_loss_bwd.  dice_kind 0 GDICE / 1 DICE / 2 GWDICE / 3 no dice term (single-loss callables); boundary_weight = alpha for BOUNDARY-RAMPED, else 1. */
typedef struct dt_loss_cfg {
  int32_t dice_kind, use_boundary, use_focal;
  float boundary_weight, gamma;
} dt_loss_cfg;
int dt_seg_loss_algebra(const double* acc, const dt_loss_cfg* cfg, int B, int K, int H, int W, float* parts, float* coef,
                        float* wfocal, float* wbound, float* wass_a, float* wass_c, void* stream);
/* GWDICE position passes.  loss/gwdl.py:180-198 broadcasts alpha[B,1,S] * (1 - wass)[B,S] to [B,B,S], so sample
 * i's generalised true positives are sum_s alpha_i(s) * V(s), V(s) = sum_j (1 - wass_j(s)) over the whole batch
 * (equal to the published formula only for B = 1); reproduced because the reference trains with it.
 *   dt_gwdice_possum : possum[H*W] = V                                   (forward, before dt_seg_loss_fwd)
 *   dt_gwdice_posgrad: posgrad[H*W] = sum_i sample_coef[i] * [label_i(s) > 0]   (backward, before dt_seg_loss_bwd) */
int dt_gwdice_possum(const float* logits, const int64_t* labels, const float* wass_m, float* possum, int B, int K,
                     int H, int W, void* stream);
int dt_gwdice_posgrad(const int64_t* labels, const float* sample_coef, float* posgrad, int B, int H, int W,
                      void* stream);

/* K x K confusion counts accumulated on the device (eval reductions, segmodel.py:291-309,337-365):
 * counts int64 [2][K][K] (+=): plane 0 all pixels, plane 1 pixels with lu == 1 (lu may be NULL);
 * rows = target, columns = prediction.  Give the prediction as int64 OR uint8 (the other pointer NULL). */
int dt_confusion_matrix(const int64_t* pred_i64, const uint8_t* pred_u8, const int64_t* target, const int64_t* lu,
                        int K, int64_t n, int64_t* counts, int32_t* err_flag, void* stream);

/* Ensemble vote (deployment/inference.py:65-116 PyTorchEnsembleInference.run): per-pixel torch.mode over the
 * uint8 class maps of M models, maps[M][n]; ties -> the smallest class (torch.mode).  n % 4 == 0, 2 <= K <= 8.
 * Writes uint8 and/or int64 maps (either pointer may be NULL); classes >= K set err_flag[0]. */
int dt_ensemble_vote(const uint8_t* maps, int M, int64_t n, int K, uint8_t* out_u8, int64_t* out_i64,
                     int32_t* err_flag, void* stream);

/* Signed Euclidean distance maps of the boundary loss, computed on the device in place of the
 * loader's scipy pass: data/deadtreedata.py:182-185 -> loss/losses.py:159-178 one_hot2dist(resolution=[1,1]).
 * labels int64 [B,H,W] -> dist fp32 [B,K,H,W]; per class: floor(edt to the class) outside it, 1 - floor(edt to the
 * outside) inside it, all zero when the class is absent (the int32 truncation of the reference included).
 * Exact integer arithmetic: bit-identical to the reference.  workspace: dt_signed_distmap_workspace() BYTES.
 * Labels outside [0,K) set err_flag[0] (the assert of losses.py:129). */
int64_t dt_signed_distmap_workspace(int B, int K, int H, int W);
int dt_signed_distmap(const int64_t* labels, float* dist, void* workspace, int32_t* err_flag, int B, int K, int H,
                      int W, void* stream);

/* ------------------------------------------------------------------ bf16 storage / fp32 accumulate (inference leg)
 * BASELINE configs[2] precision: activations NHWC bf16, weights [tap][Cout][Cin] bf16 (dt_pack_weights_bf16),
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulators, one rounding at the store.  Same descriptor semantics as
 * dt_conv2d (mode0 0/1/2, concat, split outputs, accumulate, BatchNorm partial statistics from the fp32
 * accumulators, fused input BatchNorm-apply + ReLU). */
int dt_conv2d_bf16_stat_rows(const dt_conv_desc* d);
int dt_conv2d_bf16_config(const dt_conv_desc* d, int* tile_w, int* tile_n, int* chunk_k, int* row_tiles);  /* conv_fwd_bf16_kernel<k,s,tw,tn,tf> */
int dt_conv2d_bf16(const dt_conv_desc* d, const void* src0, const void* src1, const void* w_bf16, void* out,
                   void* out1, float* stats, const float* in_scale, const float* in_shift, void* stream);
int dt_pack_weights_bf16(const float* w_hwio, void* out_bf16, int ksize, int Cin, int Cout, void* stream);
/* weights of the bf16 data-gradient convolution ([tap'][Cin][Cout] = HWIO with the taps reversed) */
int dt_pack_dgrad_weights_bf16(const float* w_hwio, void* out_bf16, int ksize, int Cin, int Cout, void* stream);
/* fp32 dW[kh][kw][ci][co] from bf16 x (same virtual-input modes as dt_conv2d_wgrad) and bf16 dy; operands are
 * transposed out of the pixel-major LDS tiles by ds_read_b64_tr_b16; split-K slabs reduced in fixed order. */
size_t dt_conv2d_wgrad_bf16_workspace(const dt_conv_desc* d);
int dt_conv2d_wgrad_bf16(const dt_conv_desc* d, const void* src0, const void* src1, const void* dy, float* dw_hwio,
                         float* workspace, size_t workspace_bytes, const float* in_scale, const float* in_shift,
                         void* stream);
/* out(bf16) = act(y*scale+shift + res'), y fp32 (y_is_f32) or bf16, res bf16 (optional affine); relu: 0 none, 1 ReLU after
 * the sum, 2 ReLU on the main branch only — relu(y*scale+shift) + res' (ResUnet decoder block, like dt_bn_act) */
int dt_bn_act_bf16(const void* y, int y_is_f32, const float* scale, const float* shift, const void* res,
                   const float* rscale, const float* rshift, void* out, int64_t n_pix, int C, int relu, void* stream);
int dt_maxpool3x3s2_bf16(const void* x, void* out, int B, int H, int W, int C, void* stream);
/* fp32 conv (dt_conv2d semantics, no split/accumulate/concat) storing its output as bf16 — the 7x7 stem of the
 * bf16 path keeps fp32 operands (K = 147) but feeds bf16 activations; statistics from the fp32 accumulators. */
int dt_conv2d_out_bf16(const dt_conv_desc* d, const float* src0, const float* w_hwio, void* out_bf16, float* stats,
                       void* stream);
/* stem weight gradient with the bf16 dy of the bf16 path (x fp32 image, dW fp32) */
int dt_conv2d_wgrad_stem_dy_bf16(const dt_conv_desc* d, const float* src0, const void* dy_bf16, float* dw_hwio,
                                 float* workspace, size_t workspace_bytes, void* stream);
/* head with bf16 decoder activations in (forward) / bf16 activation gradient out (backward); weights, logits,
 * dlogits and parameter gradients stay fp32 */
int dt_head_fwd_bf16(const void* x_bf16, const float* w_ohwi, const float* bias, float* logits_nchw,
                     int64_t* argmax_i64, uint8_t* argmax_u8, int B, int H, int W, int Cin, int K, void* stream);
int dt_head_bwd_bf16(const void* x_bf16, const float* w_ohwi, const float* dlogits_nchw, void* dx_bf16, float* red,
                     int B, int H, int W, int Cin, int K, void* stream);
int dt_bf16_to_f32(const void* x, float* out, int64_t n, void* stream);
int dt_f32_to_bf16(const float* x, void* out, int64_t n, void* stream);
/* bf16 training passes: same contracts as their fp32 namesakes, tensors bf16, sums fp32/fp64.  `red` holds
 * dt_bn_stats_floats(dt_bn_bwd_rows_bf16(n_pix), C) floats. */
int dt_bn_bwd_rows_bf16(int64_t n_pix);
int dt_bn_bwd_reduce_bf16(const void* dout, const void* out_act, const void* y, const float* mean,
                          const float* invstd, const float* act_scale, const float* act_shift, float* red,
                          int64_t n_pix, int C, void* stream);
int dt_bn_bwd_apply_bf16(const void* dout, const void* out_act, const void* y, const float* mean, const float* invstd,
                         const float* gamma, const float* act_scale, const float* act_shift, float* red, int P,
                         float* dgamma, float* dbeta, void* dy, void* dres, int dres_accumulate, int64_t n_pix, int C,
                         void* stream);
int dt_maxpool3x3s2_bf16_amax(const void* x, void* out, uint8_t* argmax, int B, int H, int W, int C, void* stream);
int dt_maxpool3x3s2_bwd_bf16(const void* dout, const uint8_t* argmax, void* dx, int accumulate, int B, int H, int W,
                             int C, void* stream);
int dt_upsample2x_bwd_bf16(const void* dup, void* dx, int B, int H, int W, int C, void* stream);
/* the same with accumulate != 0: dx = bf16(dx + sums) — a node of the Unet++ dense decoder collects several consumers */
int dt_upsample2x_bwd_acc_bf16(const void* dup, void* dx, int accumulate, int B, int H, int W, int C, void* stream);
/* bf16 twin of dt_channel_slice (channel counts and offset multiples of 8): torch.cat of the dense skip connections of
 * smp UnetPlusPlusDecoder.forward under AMP and its backward; the accumulating form adds in fp32 and rounds once */
int dt_channel_slice_bf16(const void* src, void* dst, int64_t n_pix, int C_narrow, int C_wide, int offset, int to_wide,
                          int accumulate, void* stream);

/* Every weight image of a network in one launch, table-driven over the flat parameter buffer: table int32
 * [n_layers][5] = (w_off, taps, Cin, Cout, first_tile) on the DEVICE, first_tile = running sum of
 * taps*ceil(Cin/32)*ceil(Cout/32); total_tiles = that sum over all layers.  The image of layer l lands at out + w_off
 * (in elements of the output type).  mode 0 = dt_weight_flip_transpose (fp32), 1 = dt_pack_weights_bf16,
 * 2 = dt_pack_dgrad_weights_bf16; 3 / 4 = the forward / data-gradient bf16 images in the chunked layout
 * [tap][K/32][N][32] that the LDS-DMA staged kernels read (dt_conv2d_bf16 reports them with mt == 8 in
 * dt_conv2d_bf16_config: pass THAT image to such a launch); layers with K or N not a multiple of 32 are skipped. */
int dt_weight_images(const float* params, void* out, const int32_t* table, int n_layers, int total_tiles, int mode,
                     void* stream);
/* modes 1 - 4 (the four bf16 images of a bf16 training step: forward / data-gradient, plain / chunked for the LDS-DMA
 * kernels) from one read of the parameters, each into its own n_params-element buffer */
int dt_weight_images_bf16_all(const float* params, void* fwd, void* dgrad, void* fwd_chunked, void* dgrad_chunked,
                              const int32_t* table, int n_layers, int total_tiles, void* stream);

/* The 7x7 / stride-2 stem on the bf16 convolution kernels (smp encoder conv1, reached from network/segmodel.py:214):
 * dt_stem_s2d_bf16 turns the fp32 NHWC image [B,H,W,Cin<=4] into its 2x2 space-to-depth form [B,H/2,W/2,16] bf16
 * (channel (a*2+b)*4 + c, unused channels zero), dt_stem_pack_weights_bf16 the HWIO 7x7 weights into the matching
 * [16 taps][Cout][16] image; dt_conv2d_bf16 with ksize = 4, stride 1, pad = 2 (C0 = 16, Ho = Hin, Wo = Win,
 * Cout %% 64 == 0) is then exactly the 7x7/2 convolution on bf16-rounded operands. */
int dt_stem_s2d_bf16(const float* x_nhwc, void* out_bf16, int B, int H, int W, int Cin, void* stream);
int dt_stem_pack_weights_bf16(const float* w_hwio_7x7, void* out_bf16, int Cin, int Cout, void* stream);
/* weight gradient of the stem in that form: dt_conv2d_wgrad_bf16 with ksize = 4 (C0 = 16, pad 2) gives
 * dw4 fp32 [16 taps][16][Cout]; dt_stem_unpack_wgrad gathers it into the HWIO 7x7 gradient [7][7][Cin][Cout]. */
int dt_stem_unpack_wgrad(const float* dw4, float* dw_hwio_7x7, int Cin, int Cout, void* stream);

/* bf16 twin of dt_conv2d_bn_bwd: fuse->y points at the bf16 raw output of the BatchNorm layer (cast to const
 * float*); the sums use the rounded bf16 gradient and the mask of bf16(y*scale+shift), like dt_bn_bwd_reduce_bf16.
 * P = dt_conv2d_bf16_stat_rows(desc). */
int dt_conv2d_bf16_bn_bwd(const dt_conv_desc* desc, const void* src0, const void* w_bf16, void* out, float* red,
                          const dt_bn_bwd_fuse* fuse, void* stream);

/* bf16: data gradient of y = conv3x3(nearest_upsample_x2(x)) for the narrow decoder layer in ONE launch: `desc`, dy and
 * w_bf16 as for dt_conv2d_bf16_bn_bwd (the full-resolution data-gradient form, plain store); the 2x2 sums of the
 * up-sampling's backward are taken on the fp32 accumulators, gx [B, Ho/2, Wo/2, Cout] (bf16) is stored and red
 * [2][P][Cout], P = dt_conv2d_bf16_stat_rows(desc), receives the BatchNorm-backward sums of the layer that produced x
 * (fuse: its bf16 raw output at gx's resolution, mean, invstd, act_scale / act_shift) like dt_upsample2x_bwd_bn_bf16
 * — which it replaces together with the full-resolution dt_conv2d_bf16 launch (reference: autograd of
 * F.interpolate(nearest, x2) + Conv2d under AMP, smp unet/decoder.py DecoderBlock.forward via segmodel.py:30-57). */
int dt_conv2d_bf16_upsampled_dgrad_supported(const dt_conv_desc* desc);
int dt_conv2d_bf16_upsampled_dgrad(const dt_conv_desc* desc, const void* dy, const void* w_bf16, void* gx, float* red,
                                   const dt_bn_bwd_fuse* fuse, void* stream);

/* Max-pool 3x3 / 2 backward (dt_maxpool3x3s2_bwd / _bf16) with the BatchNorm-backward sums of the layer whose (virtual)
 * activation was pooled — the stem — taken from the gradient the pass writes (after the optional join with the skip
 * gradient already in dx): red[2][P][C], P = dt_maxpool3x3s2_bwd_bn[_bf16]_rows(...) (0: odd map or channel count not
 * covered — use the plain pass + dt_bn_bwd_reduce), -> dt_bn_bwd_apply.  bf16: sums from the rounded gradient and the mask
 * of bf16(y * scale + shift), like dt_bn_bwd_reduce_bf16. */
int dt_maxpool3x3s2_bwd_bn_rows(int B, int H, int W, int C);
int dt_maxpool3x3s2_bwd_bn(const float* dout, const uint8_t* argmax, float* dx, int accumulate, const dt_bn_bwd_fuse* fuse,
                           float* red, int B, int H, int W, int C, void* stream);
int dt_maxpool3x3s2_bwd_bn_bf16_rows(int B, int H, int W, int C);
int dt_maxpool3x3s2_bwd_bn_bf16(const void* dout, const uint8_t* argmax, void* dx, int accumulate, const dt_bn_bwd_fuse* fuse,
                                float* red, int B, int H, int W, int C, void* stream);

/* Nearest x2 upsample backward (2x2 sums, like dt_upsample2x_bwd) with the BatchNorm-backward reduction of the layer
 * whose (virtual) activation was upsampled fused in: dx[B,H,W,C] is that layer's output gradient, fuse->y its raw
 * output; red[2][P][C], P = dt_upsample2x_bwd_bn_rows(...), holds dt_bn_stats_floats(P, C) floats -> dt_bn_bwd_apply.
 * The _bf16 twins take bf16 tensors (fuse->y bf16) and follow dt_bn_bwd_reduce_bf16's arithmetic. */
int dt_upsample2x_bwd_bn_rows(int B, int H, int W, int C);
int dt_upsample2x_bwd_bn(const float* dup, float* dx, const dt_bn_bwd_fuse* fuse, float* red, int B, int H, int W, int C,
                         void* stream);
/* Data gradient of y = conv3x3(nearest_upsample_x2(x)) in ONE kernel (sub-pixel form: a 4x4 stride-2 convolution over
 * dY with 16 combined weight matrices, 16 tap products per source pixel instead of 36): replaces the chain
 * dt_conv2d (data-gradient form) -> dt_upsample2x_bwd(_bn) for the narrow decoder layer without a skip (reference:
 * autograd of F.interpolate(x, scale_factor=2, mode="nearest") + Conv2d, deadtrees/network/segmodel.py:30-57 via
 * segmentation_models_pytorch 0.2.1 unet/decoder.py DecoderBlock.forward).  `desc` describes the FORWARD convolution
 * (mode0 = 1, C0 = channels of x, Cout); w_hwio the forward weights; gx[B, Hin/2, Win/2, C0].  With `fuse` (y, mean,
 * invstd, act_scale, act_shift of the layer that produced x) red[2][P][C0], P = dt_conv2d_upsampled_dgrad_rows(desc),
 * receives the BatchNorm-backward sums like dt_upsample2x_bwd_bn; fuse = NULL: plain gradient. */
int dt_conv2d_upsampled_dgrad_supported(const dt_conv_desc* desc);
int dt_conv2d_upsampled_dgrad_rows(const dt_conv_desc* desc);
int dt_conv2d_upsampled_dgrad(const dt_conv_desc* desc, const float* dy, const float* w_hwio, float* gx, float* red,
                              const dt_bn_bwd_fuse* fuse, void* stream);
int dt_upsample2x_bwd_bn_bf16_rows(int B, int H, int W, int C);
int dt_upsample2x_bwd_bn_bf16(const void* dup, void* dx, const dt_bn_bwd_fuse* fuse, float* red, int B, int H, int W,
                              int C, void* stream);

/* ------------------------------------------------------------------ optimiser (K22) */
/* sum of squares of g[n] -> partial[rows]; rows = dt_sumsq_rows(n) */
int dt_sumsq_rows(int64_t n);
int dt_sumsq(const float* g, int64_t n, double* partial, void* stream);
/* norm[0] = sqrt(sum partial) ; clipcoef[0] = min(1, max_norm/(norm+1e-6)) * gscale  (clip_grad_norm_).
 * skip_flag (may be NULL): set to 1 when the norm is NaN/Inf — a non-finite gradient must not reach Adam's moments even
 * when the loss itself stayed finite (the reference's Lightning loop would have seen the NaN loss of the next step;
 * here the update is dropped like a non-finite loss, segmodel.py:220-222).  Never cleared here. */
int dt_clip_coef(const double* partial, int rows, float max_norm, float gscale, float* norm, float* clipcoef,
                 int32_t* skip_flag, void* stream);
/* torch.optim.Adam step on a flat buffer (segmodel.py:420-425): g' = g*clipcoef[0];
 * skipped entirely when skip_flag[0] != 0 (non-finite loss: segmodel.py:220-222). */
int dt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, double beta1, double beta2,
                 float eps, float bias_c1, float bias_c2, const float* clipcoef, const int32_t* skip_flag,
                 void* stream);
/* skip_flag[0] = !isfinite(loss[0])  (segmodel.py:220-222: training_step returns None) */
int dt_skip_from_loss(const float* loss, int32_t* skip_flag, void* stream);
/* device-resident step count: t += (skip ? 0 : 1); hyper[3] = (lr_dev[0], 1 - beta1^t, 1 - beta2^t) for
 * dt_adam_step_dev — a skipped step does not advance the bias correction (torch.optim.Adam is not called then).
 * The betas travel as doubles: torch computes 1 - beta^t in Python doubles ((double)(float)0.999 is off by 1.3e-8). */
int dt_adam_advance(double* t_dev, const int32_t* skip_flag, const double* lr_dev, double beta1, double beta2,
                    float* hyper, void* stream);
/* the same step with the per-step scalars on the device: hyper fp32 [3] = (lr, 1 - beta1^t, 1 - beta2^t), so a
 * training step captured in a HIP graph replays with the current learning rate and bias corrections. */
int dt_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, double beta1,
                     double beta2, float eps, const float* clipcoef, const int32_t* skip_flag, void* stream);

/* ------------------------------------------------------------------ library options */
/* Kernel-selection switches (host side, process wide; every choice computes the same values):
 *   "bf16_dma": 0 = register-staged bf16 convolutions only, 1 (default) = the LDS-DMA staged 512-pixel kernel where
 *               its tiles fill the chip, 2 = wherever its shape conditions hold.  Also read from the environment
 *               variable DT_BF16_DMA at first use. */
int dt_set_option(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* DEADTREES_HIP_H */
