// Argument block shared by the bf16 convolution kernels (conv_bf16.hip, conv_bf16_dma.hip).
#pragma once
#include "common.h"

struct ConvBfArgs {
  const __bf16* src0;
  const __bf16* src1;
  const __bf16* w;  // [tap][Cout][Cin]
  const float* in_scale;
  const float* in_shift;
  __bf16* out;
  __bf16* out1;     // channels >= cout_split (decoder concat data gradient)
  float* stats;     // [2][P][Cout] BatchNorm partial sums from the fp32 accumulators, or null
  // fused BatchNorm-backward reduction (dt_conv2d_bf16_bn_bwd): bnb.y (bf16, shape of `out`) set -> `stats` receives
  // sum g, sum g*xhat of the layer this data gradient belongs to instead of sum v, sum v^2
  dt_bn_bwd_fuse bnb;
  int B, Hin, Win, C0, C1, mode0, Ho, Wo, Cout, pad, tiles_x, tiles_y, n_tiles, P, cout_split, accumulate;
  int pstats = 0;   // conv_bf16_dma.hip: BatchNorm partial sums accumulated over a workgroup's tiles, one row per workgroup
};

// ---- LDS-DMA staged, double-buffered 3x3 stride-1 kernel (conv_bf16_dma.hip): 512-pixel x 64-channel tiles
int dt_conv_bf16_dma_supported(const dt_conv_desc* d);
int dt_conv_bf16_dma_launch(ConvBfArgs a, hipStream_t st);   // fills tiles_x / tiles_y / n_tiles / P itself
int dt_conv_bf16_dma_stat_rows(const dt_conv_desc* d);

// ---- LDS-DMA staged, double-buffered, persistent bf16 weight gradient of the 3x3 stride-1 layers with 64-channel
// blocks and a plain (untransformed) input (conv_bf16_wgrad_dma.hip); launch -> number of split-K slabs in `ws`
int dt_wgrad_bf16_dma_supported(const dt_conv_desc* d);
size_t dt_wgrad_bf16_dma_workspace(const dt_conv_desc* d);
int dt_wgrad_bf16_dma_launch(const dt_conv_desc* d, const void* src0, const void* src1, const void* dy, float* ws,
                             hipStream_t st);

// ---- lean persistent kernel for the narrow full-resolution decoder layers (conv_bf16_narrow.hip): Cin, Cout in {16, 32},
// weights in registers, 16-wide N; reported by dt_conv2d_bf16_config as mt = 16
int dt_conv_bf16_narrow_supported(const dt_conv_desc* d);
int dt_conv_bf16_narrow_grid(const dt_conv_desc* d, int tf, int bnb);   // persistent workgroups of the variant
int dt_conv_bf16_narrow_rows(const dt_conv_desc* d);                    // rows of the statistics buffer (>= any grid)
int dt_conv_bf16_narrow_launch(const dt_conv_desc* d, ConvBfArgs a, hipStream_t st, bool upsample_bwd = false);
int dt_wgrad_bf16_narrow_supported(const dt_conv_desc* d);
size_t dt_wgrad_bf16_narrow_workspace(const dt_conv_desc* d);
int dt_wgrad_bf16_narrow_launch(const dt_conv_desc* d, const void* src0, const void* dy, float* ws, const float* in_scale,
                                const float* in_shift, hipStream_t st);   // -> slabs written, or a negative code
