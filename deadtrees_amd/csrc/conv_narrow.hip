// 16-channel-granular convolution kernels for the full-resolution decoder layers (dec.4: 32->16, 16->16
// at 512x512) on v_mfma_f32_16x16x4_f32.
//
// The 32x32x2 tiles of conv_fwd.hip / conv_wgrad.hip pad Cout=16 (and Cin=16) to 32 and waste half (or 3/4)
// of the matrix-core work on exactly the layers with the most pixels (dec.4.conv1 2.4 GFLOP,
// dec.4.conv2 1.2 GFLOP per tile).  16x16x4 has the same FLOP/cycle (2048 FLOP / 32 cycles per SIMD) at a
// 16-wide N, so these layers run without padding.  Same structure otherwise: im2col-free, channel-planar LDS
// halo tile reused by the 9 taps, virtual nearest-upsample in the LDS fill, BatchNorm partial statistics in the
// epilogue; split-K weight gradient with deterministic slab reduction.
// Reference ops replaced: ATen conv2d / convolution_backward of smp UnetDecoder block 4
// (deadtrees/network/segmodel.py:214; twin deadtrees/network/extra/resunet/decoder.py:40-52).
#include "common.h"

struct NarrowArgs {
  const float* src0;
  const float* in_scale;  // optional fused BatchNorm-apply + ReLU of the producer layer (see conv_fwd.hip)
  const float* in_shift;
  const float* w;      // [9][Cin][Cout]
  float* out;
  float* stats;        // [2][P][Cout] or null
  dt_bn_bwd_fuse bnb;  // fused BatchNorm-backward reduction (see conv_fwd.hip): stats = sum g, sum g*xhat
  int B, Hin, Win, Cin, mode0, Ho, Wo, Cout, tiles_x, tiles_y, P;
};

// ------------------------------------------------------------------ forward / data gradient, Cout <= 16
// tile 8 x 32 output pixels x 16 channels per 4-wave workgroup; wave w owns rows 2w, 2w+1 (4 M-tiles of 16 px)
#define N16_TW 32
#define N16_TH 8
#define N16_HW (N16_TW + 2)
#define N16_HH (N16_TH + 2)
#define N16_PLANE 368  // >= 340, == 16 (mod 32): the two 16-lane runs of a 32-lane read group tile all 32 banks
#define N16_CK 16

__device__ __forceinline__ int n16_plane_base(int c) { return c * N16_PLANE + 8 * ((c >> 2) & 3); }

__global__ __launch_bounds__(256, 4) void conv_fwd_n16_kernel(const NarrowArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[N16_CK * N16_PLANE + 32 + 9 * N16_CK * 16];
  float* lds_in = lds;
  float* lds_w = lds + N16_CK * N16_PLANE + 32;
  const int sp = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
  const int oy0 = ty * N16_TH, ox0 = tx * N16_TW;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;

  // per-lane LDS offsets of the 4 M-tiles (rows 2w,2w+1 x halves 0,1), channel kq of a 4-channel k-step
  int abase[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int py = wave * 2 + (t >> 1), px = (t & 1) * 16 + m;
    abase[t] = kq * N16_PLANE + py * N16_HW + px;
  }
  const int bbase = kq * 16 + m;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fill bookkeeping: 340 halo pixels x 4 channel quads; lane -> (quad fastest, pixel)
  constexpr int IN_TOTAL = N16_HH * N16_HW * 4, IN_IT = (IN_TOTAL + 255) / 256;
  const int qi = tid & 3, pix0 = tid >> 2;
  const int Hs = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws = a.mode0 ? (a.Win >> 1) : a.Win;
  int pidx[IN_IT];
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int pix = pix0 + it * 64;
    const int hy = pix / N16_HW, hx = pix - hy * N16_HW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win && pix < N16_HH * N16_HW;
    const int sy = a.mode0 ? (iy >> 1) : iy, sx = a.mode0 ? (ix >> 1) : ix;
    pidx[it] = ok ? (b * Hs + sy) * Ws + sx : -1;
  }
  const int lds_q_base = n16_plane_base(4 * qi);

  for (int c0 = 0; c0 < a.Cin; c0 += N16_CK) {
    f32x4 rin[IN_IT];
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pidx[it] >= 0 && c0 + 4 * qi < a.Cin)
        v = *reinterpret_cast<const f32x4*>(a.src0 + (size_t)pidx[it] * a.Cin + c0 + 4 * qi);
      rin[it] = v;
    }
    // weights of this chunk: [9][16][16] (zero-padded rows/cols)
    f32x4 rw[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int idx = tid + it * 256;       // < 576 float4
      const int q = idx & 3, row = idx >> 2;  // row = tap*16 + k
      const int tap = row >> 4, k = row & 15;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < 576 && c0 + k < a.Cin && 4 * q < a.Cout)
        v = *reinterpret_cast<const f32x4*>(a.w + ((size_t)tap * a.Cin + c0 + k) * a.Cout + 4 * q);
      rw[it] = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int pix = pix0 + it * 64;
      if (pix < N16_HH * N16_HW) {
        f32x4 v = rin[it];
        if (a.in_scale != nullptr && pidx[it] >= 0 && c0 + 4 * qi < a.Cin) {
          v = v * *reinterpret_cast<const f32x4*>(a.in_scale + c0 + 4 * qi) +
              *reinterpret_cast<const f32x4*>(a.in_shift + c0 + 4 * qi);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] < 0.f ? 0.f : v[k];
        }
        float* d = lds_in + lds_q_base + pix;
        d[0] = v[0];
        d[N16_PLANE] = v[1];
        d[2 * N16_PLANE] = v[2];
        d[3 * N16_PLANE] = v[3];
      }
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int idx = tid + it * 256;
      if (idx < 576) *reinterpret_cast<f32x4*>(lds_w + idx * 4) = rw[it];
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
#pragma unroll
      for (int c4 = 0; c4 < N16_CK / 4; ++c4) {
        const float bv = lds_w[bbase + (tap * 16 + c4 * 4) * 16];
        const int coff = c4 * 4 * N16_PLANE + 8 * c4 + kh * N16_HW + kw;   // plane base incl. the +8*quad skew
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float av = lds_in[abase[t] + coff];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // ---- epilogue: D col = lane&15 (channel), row = 4*(lane>>4) + reg (pixel within the M-tile)
  float s1 = 0.f, s2 = 0.f;
  const bool nok = m < a.Cout;
  const bool bnb = a.bnb.y != nullptr;   // uniform
  float b_mu = 0.f, b_is = 0.f, b_sc = 0.f, b_sh = 0.f;
  if (bnb && nok) {
    b_mu = a.bnb.mean[m];
    b_is = a.bnb.invstd[m];
    b_sc = a.bnb.act_scale[m];
    b_sh = a.bnb.act_shift[m];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int oy = oy0 + wave * 2 + (t >> 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ox = ox0 + (t & 1) * 16 + 4 * kq + i;
      if (nok && oy < a.Ho && ox < a.Wo) {
        const float v = acc[t][i];
        const size_t o = (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + m;
        if (bnb) {
          const float yv = a.bnb.y[o];
          const float g = (yv * b_sc + b_sh) > 0.f ? v : 0.f;
          s1 += g;
          s2 += g * ((yv - b_mu) * b_is);
        } else {
          s1 += v;
          s2 += v * v;
        }
        a.out[o] = v;
      }
    }
  }
  if (a.stats != nullptr) {
    s1 += __shfl_xor(s1, 16, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    __syncthreads();
    float* red = lds;  // [2][4][16]
    if (lane < 16) {
      red[wave * 16 + lane] = s1;
      red[64 + wave * 16 + lane] = s2;
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, c = tid & 15;
      if (c < a.Cout) {
        const float* rr = red + which * 64 + c;
        a.stats[((size_t)which * a.P + sp) * a.Cout + c] = (rr[0] + rr[16]) + (rr[32] + rr[48]);
      }
    }
  }
}

extern "C" int dt_conv2d_n16_supported(const dt_conv_desc* d) {
  return d && d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->Cout <= 16 && (d->Cout & 3) == 0 && d->C1 == 0 &&
         (d->C0 & 3) == 0 && d->C0 <= 64 && d->mode0 != 2 && d->cout_split == 0 && d->accumulate == 0 && d->Wo > 16;
}

int dt_conv2d_n16_rows(const dt_conv_desc* d) { return d->B * dt_cdiv(d->Ho, N16_TH) * dt_cdiv(d->Wo, N16_TW); }

int dt_conv2d_n16_launch(const dt_conv_desc* d, const float* src0, const float* w, float* out, float* stats,
                         const float* in_scale, const float* in_shift, hipStream_t st, const dt_bn_bwd_fuse* fuse) {
  NarrowArgs a;
  a.bnb = fuse ? *fuse : dt_bn_bwd_fuse{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a.src0 = src0; a.w = w; a.out = out; a.stats = stats; a.in_scale = in_scale; a.in_shift = in_shift;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.Cin = d->C0; a.mode0 = d->mode0;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.tiles_x = dt_cdiv(d->Wo, N16_TW); a.tiles_y = dt_cdiv(d->Ho, N16_TH);
  a.P = d->B * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL(conv_fwd_n16_kernel, dim3(a.P), dim3(256), 0, st, a);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ weight gradient, Cin <= 32 and Cout <= 32
// M = 16 input channels, N = 16 output channels, K = 4 pixels per MFMA; one accumulator per (tap, ci-tile, co-tile).
struct NarrowWgArgs {
  const float* src0;
  const float* in_scale;
  const float* in_shift;
  const float* dy;
  float* ws;           // [parts][9][Cin][Cout]
  int B, Hin, Win, Cin, mode0, Ho, Wo, Cout, tiles_x, tiles_y, T, ksplit;
};

#define W16_TW 32
#define W16_TH 4   // one output row per wave per tile
#define W16_HW (W16_TW + 2)
#define W16_HH (W16_TH + 2)

template <int CIT, int COT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_n16_kernel(const NarrowWgArgs a) {
  constexpr int CIW = CIT * 16, COW = COT * 16;
  constexpr int X_ELEMS = W16_HH * W16_HW * CIW, Y_ELEMS = W16_TH * W16_TW * COW;
  __shared__ __attribute__((aligned(16))) float lds[X_ELEMS + Y_ELEMS + 64];
  float* lx = lds;
  float* ly = lds + X_ELEMS;
  const int ks = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kq = lane >> 4;
  f32x4 acc[9][CIT][COT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < CIT; ++i)
#pragma unroll
      for (int j = 0; j < COT; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int QX = CIW / 4, QY = COW / 4;
  constexpr int X_TOTAL = W16_HH * W16_HW * QX, X_IT = (X_TOTAL + 255) / 256;
  constexpr int Y_TOTAL = W16_TH * W16_TW * QY, Y_IT = (Y_TOTAL + 255) / 256;
  const int qx = tid % QX, px0 = tid / QX, qy = tid % QY, py0 = tid / QY;
  const int Hs = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws = a.mode0 ? (a.Win >> 1) : a.Win;
  f32x4 x_sc = {1.f, 1.f, 1.f, 1.f}, x_sh = {0.f, 0.f, 0.f, 0.f};
  if (a.in_scale != nullptr && 4 * qx < a.Cin) {
    x_sc = *reinterpret_cast<const f32x4*>(a.in_scale + 4 * qx);
    x_sh = *reinterpret_cast<const f32x4*>(a.in_shift + 4 * qx);
  }
  const int xb = wave * W16_HW * CIW + kq * CIW + m;   // row `wave` of the tile, pixel kq of the 4-pixel k-step
  const int yb = wave * W16_TW * COW + kq * COW + m;

  for (int tile = ks; tile < a.T; tile += a.ksplit) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * W16_TH, ox0 = tx * W16_TW;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    f32x4 rx[X_IT], ry[Y_IT];
    unsigned xvalid = 0;
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int pix = px0 + it * (256 / QX);
      const int hy = pix / W16_HW, hx = pix - hy * W16_HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pix < W16_HH * W16_HW && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win && 4 * qx < a.Cin) {
        const int sy = a.mode0 ? (iy >> 1) : iy, sx = a.mode0 ? (ix >> 1) : ix;
        v = *reinterpret_cast<const f32x4*>(a.src0 + (((size_t)b * Hs + sy) * Ws + sx) * a.Cin + 4 * qx);
        xvalid |= 1u << it;
      }
      rx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int pix = py0 + it * (256 / QY);
      const int oy = oy0 + pix / W16_TW, ox = ox0 + pix % W16_TW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pix < W16_TH * W16_TW && oy < a.Ho && ox < a.Wo && 4 * qy < a.Cout)
        v = *reinterpret_cast<const f32x4*>(a.dy + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + 4 * qy);
      ry[it] = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int pix = px0 + it * (256 / QX);
      if (pix < W16_HH * W16_HW) {
        f32x4 v = rx[it];
        if (a.in_scale != nullptr && ((xvalid >> it) & 1u)) {   // fused BN-apply + ReLU; padding stays zero
          v = v * x_sc + x_sh;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] < 0.f ? 0.f : v[k];
        }
        *reinterpret_cast<f32x4*>(lx + pix * CIW + 4 * qx) = v;
      }
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int pix = py0 + it * (256 / QY);
      if (pix < W16_TH * W16_TW) *reinterpret_cast<f32x4*>(ly + pix * COW + 4 * qy) = ry[it];
    }
    __syncthreads();
#pragma unroll
    for (int j4 = 0; j4 < W16_TW / 4; ++j4) {
      float bv[COT];
#pragma unroll
      for (int j = 0; j < COT; ++j) bv[j] = ly[yb + (j4 * 4) * COW + j * 16];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int kh = t / 3, kw = t % 3;
#pragma unroll
        for (int i = 0; i < CIT; ++i) {
          const float av = lx[xb + (kh * W16_HW + j4 * 4 + kw) * CIW + i * 16];
#pragma unroll
          for (int j = 0; j < COT; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[t][i][j], 0, 0, 0);
        }
      }
    }
  }
  // slab of this (k-split, wave): D row = 4*(lane>>4)+reg = ci within tile, col = lane&15 = co within tile
  const int part = ks * 4 + wave;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < CIT; ++i)
#pragma unroll
      for (int j = 0; j < COT; ++j) {
        const int co = j * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = i * 16 + 4 * kq + r;
          if (ci < a.Cin && co < a.Cout)
            a.ws[(((size_t)part * 9 + t) * a.Cin + ci) * a.Cout + co] = acc[t][i][j][r];
        }
      }
}

extern "C" int dt_conv2d_wgrad_n16_supported(const dt_conv_desc* d) {
  const int Cin = d ? d->C0 + d->C1 : 0;
  return d && d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->C1 == 0 && (Cin & 3) == 0 && Cin <= 32 &&
         (d->Cout & 3) == 0 && d->Cout <= 32 && !(Cin > 16 && d->Cout > 16) && d->mode0 != 2 && d->Wo > 16;
}

int dt_wgrad_n16_cfg(const dt_conv_desc* d, int* ksplit, int* parts) {
  const int T = d->B * dt_cdiv(d->Ho, W16_TH) * dt_cdiv(d->Wo, W16_TW);
  int ks = 512;
  if (ks > T) ks = T;
  *ksplit = ks;
  *parts = ks * 4;
  return T;
}

int dt_wgrad_n16_launch(const dt_conv_desc* d, const float* src0, const float* dy, float* ws, const float* in_scale,
                        const float* in_shift, hipStream_t st) {
  NarrowWgArgs a;
  a.src0 = src0; a.dy = dy; a.ws = ws; a.in_scale = in_scale; a.in_shift = in_shift;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.Cin = d->C0; a.mode0 = d->mode0;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.tiles_x = dt_cdiv(d->Wo, W16_TW); a.tiles_y = dt_cdiv(d->Ho, W16_TH);
  int parts;
  a.T = dt_wgrad_n16_cfg(d, &a.ksplit, &parts);
  const int cit = d->C0 > 16 ? 2 : 1, cot = d->Cout > 16 ? 2 : 1;
  if (cit == 1 && cot == 1)
    hipLaunchKernelGGL((conv_wgrad_n16_kernel<1, 1>), dim3(a.ksplit), dim3(256), 0, st, a);
  else if (cit == 2 && cot == 1)
    hipLaunchKernelGGL((conv_wgrad_n16_kernel<2, 1>), dim3(a.ksplit), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_n16_kernel<1, 2>), dim3(a.ksplit), dim3(256), 0, st, a);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
