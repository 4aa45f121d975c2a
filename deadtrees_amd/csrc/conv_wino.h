// Winograd F(2x2,3x3) fp32 convolution (conv_wino.hip): internal launch interface used by conv_fwd.hip's dispatch.
#pragma once
#include "common.h"

extern "C" int dt_conv2d_winograd_supported(const dt_conv_desc* d);
extern "C" int dt_conv2d_winograd_stat_rows(const dt_conv_desc* d);
int dt_conv_wino_launch(const dt_conv_desc* d, const float* src0, const float* src1, const float* u, float* out0,
                        float* out1, float* stats, const float* in_scale, const float* in_shift,
                        const dt_bn_bwd_fuse* fuse, hipStream_t st, bool affine = false);
