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

// ---- sub-pixel weight gradient of the up-sampled layer (dec4.conv1): the transpose of conv3x3_f32_upc_kernel in the
// weights.  With y[2Y+a][2X+b] = sum_{dy,dx} W'_ab[dy][dx] x[Y+a-1+dy][X+b-1+dx] (W' = sums of the 3x3 taps, see there):
//     dW'_ab[dy][dx] = sum_{Y,X} x[Y+a-1+dy][X+b-1+dx] (x) dY[2Y+a][2X+b]      16 products per source pixel (36 direct)
//     dW[kh][kw]     = sum over the (a,dy) with kh in R(a,dy) and the (b,dx) with kw in R(b,dx) of dW'_ab[dy][dx]
// The 16 accumulators of a lane hold the same (ci, co) element, so the fold to nine taps is lane-local and the partial
// slabs leave in the [part][9][Cin][Cout] layout of conv_wgrad_n16_kernel (same split-K reduction, fixed order).
// K step = 4 source pixels of a row: nine shifted x fragments (A, M = ci) and four parity fragments of dY (B, N = co)
// feed 16 CIT MFMAs; dY is kept parity-split in LDS ([row][column parity][32 columns]) and the pixel pitches (16 / 48
// floats) make every ds_read_b32 of a fragment conflict-free.  Persistent over the workgroup's tiles with the next
// tile's loads in flight.  Measured: DESIGN.md 10.
#define UW_TW 32            // source (low-resolution) columns per tile
#define UW_TH 4             // source rows per tile: one per wave
#define UW_HW (UW_TW + 2)
#define UW_HH (UW_TH + 2)

template <int CIT>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_f32_upc_kernel(const NarrowWgArgs a) {
  constexpr int CIW = 16 * CIT, COW = 16;
  constexpr int XP = CIT == 1 ? 16 : 48;        // floats per source pixel in LDS (bank rule: kq * XP + [0, 16) disjoint mod 64)
  constexpr int X_ELEMS = UW_HH * UW_HW * XP, Y_ELEMS = 2 * UW_TH * 2 * UW_TW * COW;
  __shared__ __attribute__((aligned(16))) float lds[X_ELEMS + Y_ELEMS];
  float* lx = lds;
  float* ly = lds + X_ELEMS;
  const int ks = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int Hs = a.Hin >> 1, Ws = a.Win >> 1;
  f32x4 acc[2][2][2][2][CIT];                   // [a][b][dy][dx][ci block]
#pragma unroll
  for (int e = 0; e < 16; ++e)
#pragma unroll
    for (int i = 0; i < CIT; ++i) acc[e >> 3][(e >> 2) & 1][(e >> 1) & 1][e & 1][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int QX = CIW / 4, QY = COW / 4;
  constexpr int X_TOTAL = UW_HH * UW_HW * QX, X_IT = (X_TOTAL + 255) / 256;
  constexpr int Y_TOTAL = 2 * UW_TH * 2 * UW_TW * QY, Y_IT = Y_TOTAL / 256;
  static_assert(Y_TOTAL % 256 == 0, "dY fill");
  const int qx = tid % QX, px0 = tid / QX, qy = tid % QY, py0 = tid / QY;
  f32x4 x_sc = {1.f, 1.f, 1.f, 1.f}, x_sh = {0.f, 0.f, 0.f, 0.f};
  const bool tf = a.in_scale != nullptr;
  if (tf) {
    x_sc = *reinterpret_cast<const f32x4*>(a.in_scale + 4 * qx);
    x_sh = *reinterpret_cast<const f32x4*>(a.in_shift + 4 * qx);
  }
  // fragment bases: x halo pixel (wave + 1 + ro, 4 j4 + kq + 1 + co); dY (row 2 wave + a, parity b, column 4 j4 + kq)
  const int xb = (wave * UW_HW + kq) * XP + m;
  const int yb = ((2 * wave) * 2 * UW_TW + kq) * COW + m;

  f32x4 rx[X_IT], ry[Y_IT];
  unsigned xvalid = 0;
  auto issue_loads = [&](int tile) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int sy0 = ty * UW_TH - 1, sx0 = tx * UW_TW - 1;
    xvalid = 0;
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int pix = px0 + it * (256 / QX);
      const int hy = pix / UW_HW, hx = pix - hy * UW_HW;
      const int sy = sy0 + hy, sx = sx0 + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pix < UW_HH * UW_HW && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws) {
        v = *reinterpret_cast<const f32x4*>(a.src0 + (((size_t)b * Hs + sy) * Ws + sx) * CIW + 4 * qx);
        xvalid |= 1u << it;
      }
      rx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int pix = py0 + it * (256 / QY);                      // (row, column) of the 8 x 64 output tile
      const int oy = 2 * ty * UW_TH + pix / (2 * UW_TW), ox = 2 * tx * UW_TW + pix % (2 * UW_TW);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (oy < a.Ho && ox < a.Wo) v = *reinterpret_cast<const f32x4*>(a.dy + (((size_t)b * a.Ho + oy) * a.Wo + ox) * COW + 4 * qy);
      ry[it] = v;
    }
  };

  if (ks < a.T) issue_loads(ks);
  for (int tile = ks; tile < a.T; tile += a.ksplit) {
    __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int pix = px0 + it * (256 / QX);
      if (pix < UW_HH * UW_HW) {
        f32x4 v = rx[it];
        if (tf && ((xvalid >> it) & 1u)) {   // the producer's BatchNorm + ReLU; padding stays zero
          v = v * x_sc + x_sh;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] < 0.f ? 0.f : v[k];
        }
        *reinterpret_cast<f32x4*>(lx + pix * XP + 4 * qx) = v;
      }
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int pix = py0 + it * (256 / QY);
      const int row = pix / (2 * UW_TW), col = pix % (2 * UW_TW);
      *reinterpret_cast<f32x4*>(ly + ((row * 2 + (col & 1)) * UW_TW + (col >> 1)) * COW + 4 * qy) = ry[it];
    }
    __syncthreads();
    if (tile + a.ksplit < a.T) issue_loads(tile + a.ksplit);

#pragma unroll 2
    for (int j4 = 0; j4 < UW_TW / 4; ++j4) {
      float bv[2][2], av[3][3][CIT];
#pragma unroll
      for (int pa = 0; pa < 2; ++pa)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) bv[pa][pb] = ly[yb + ((pa * 2 + pb) * UW_TW + 4 * j4) * COW];
#pragma unroll
      for (int ro = 0; ro < 3; ++ro)
#pragma unroll
        for (int co = 0; co < 3; ++co)
#pragma unroll
          for (int i = 0; i < CIT; ++i) av[ro][co][i] = lx[xb + (ro * UW_HW + 4 * j4 + co) * XP + 16 * i];
#pragma unroll
      for (int pa = 0; pa < 2; ++pa)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
              for (int i = 0; i < CIT; ++i)
                acc[pa][pb][dy][dx][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[pa + dy][pb + dx][i], bv[pa][pb],
                                                                              acc[pa][pb][dy][dx][i], 0, 0, 0);
    }
  }
  // ---- fold to the nine taps (fixed order) and write this (k-split, wave)'s slab: D row = ci 4 kq + r, column = co m
  const int part = ks * 4 + wave;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      // tap row kh lies in R(a, dy) for: kh 0 -> (0,0), (1,0); kh 1 -> (0,1), (1,0); kh 2 -> (0,1), (1,1)
      const int a0 = kh == 0 ? 0 : (kh == 1 ? 0 : 0), d0 = kh == 0 ? 0 : 1;
      const int a1 = 1, d1 = kh == 2 ? 1 : 0;
      const int b0 = 0, e0 = kw == 0 ? 0 : 1;
      const int b1 = 1, e1 = kw == 2 ? 1 : 0;
#pragma unroll
      for (int i = 0; i < CIT; ++i) {
        const f32x4 v = (acc[a0][b0][d0][e0][i] + acc[a0][b1][d0][e1][i]) + (acc[a1][b0][d1][e0][i] + acc[a1][b1][d1][e1][i]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          a.ws[(((size_t)part * 9 + kh * 3 + kw) * CIW + 16 * i + 4 * kq + r) * COW + m] = v[r];
      }
    }
}

static bool nl_upc_enabled();
static int wg_upc(const dt_conv_desc* d) {   // 1: the sub-pixel weight-gradient kernel takes this (n16-supported) layer
  static const int on = [] {
    const char* e = getenv("DT_FP32_SUBPIXEL_WGRAD");
    return (e == nullptr || e[0] != '0') ? 1 : 0;
  }();
  return on && nl_upc_enabled() && d->mode0 == 1 && d->C1 == 0 && (d->C0 == 16 || d->C0 == 32) && d->Cout == 16 &&
         ((d->Hin | d->Win) & 1) == 0 && d->Ho == d->Hin && d->Wo == d->Win && d->Win >= 64;
}

int dt_wgrad_n16_cfg(const dt_conv_desc* d, int* ksplit, int* parts) {
  const int T = wg_upc(d) ? d->B * dt_cdiv(d->Hin / 2, UW_TH) * dt_cdiv(d->Win / 2, UW_TW)
                          : d->B * dt_cdiv(d->Ho, W16_TH) * dt_cdiv(d->Wo, W16_TW);
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
  if (wg_upc(d)) {
    a.tiles_x = dt_cdiv(d->Win / 2, UW_TW); a.tiles_y = dt_cdiv(d->Hin / 2, UW_TH);
    if (d->C0 == 16)
      hipLaunchKernelGGL((conv3x3_wgrad_f32_upc_kernel<1>), dim3(a.ksplit), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv3x3_wgrad_f32_upc_kernel<2>), dim3(a.ksplit), dim3(256), 0, st, a);
    DT_LAUNCH_CHECK();
    return DT_OK;
  }
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

// ================================================================================================================
// Round 3: lean persistent fp32 kernel for the narrow full-resolution decoder layers — Cin, Cout in {16, 32}, 3x3
// stride 1 (dec3.conv2, dec4.conv1 / conv2 and their data gradients; in inference the same layers).  The fp32 twin of
// conv_bf16_narrow.hip: the round-1/2 kernels above and conv_fwd_kernel<3, 1, 32, 32, 8, *> keep the matrix pipe busy
// 39-57 % on these MFMA-bound layers (PMC: ~500 non-MFMA vector instructions per wave and 256-pixel tile, weights
// re-staged through LDS per tile, one ds_read_b32 per MFMA).  Here:
//   * weights in REGISTERS for the whole kernel (v_mfma_f32_16x16x4_f32: one register per K step of 4 channels and N
//     block: 36 registers for 16 -> 16, 144 for 32 -> 32);
//   * the K order inside a 16-channel block is permuted (lane group kq supplies channels 4 kq .. 4 kq + 3, one per K
//     step) so that ONE ds_read_b128 per tap and block feeds four MFMAs; pixel rows at a 96- / 160-byte pitch are
//     conflict-free for those reads (brute-forced against the bank rule);
//   * persistent over 8 x 32-pixel tiles, the next tile's input in flight in registers (not for 32 -> 32: its 144 weight
//     registers leave no room, and its 18k MFMA cycles per tile cover a load by themselves);
//   * BatchNorm statistics / fused BatchNorm-backward sums accumulated in registers over all tiles of a workgroup: one
//     row per workgroup; interior tiles test no bounds.
struct NarrowLeanArgs {
  const float* src0;
  const float* in_scale;
  const float* in_shift;
  const float* w;      // [9][Cin][Cout]
  float* out;
  float* stats;        // [2][P][Cout] or null
  dt_bn_bwd_fuse bnb;
  int B, Hin, Win, mode0, tiles_x, tiles_y, P;
};

#define NL_PIX (N16_HH * N16_HW)   // 340 halo pixels of an 8 x 32 tile

// EPI: 0 store (+ BatchNorm statistics); 1 store + fused BatchNorm-backward sums (virtual activation); 4 inference:
// out = relu(conv * scale + shift) — eval-mode BatchNorm + ReLU on the accumulators (bn_act's mul, add, NaN-keeping ReLU)
template <int CB, int NB, bool TF, int EPI>
__global__ __launch_bounds__(256, (CB * NB == 1) ? 3 : 2) void conv3x3_f32_narrow_kernel(
    const NarrowLeanArgs a, const int total_tiles) {
  constexpr int CIN = 16 * CB, COUT = 16 * NB;
  constexpr int PITCH = CB == 1 ? 96 : 160;                 // bytes per halo pixel
  constexpr int SLOTS = 4 * CB;                             // 16-byte slots (4 channels) per pixel
  constexpr int FILL = (NL_PIX * SLOTS + 255) / 256;
  constexpr bool PREFETCH = CB * NB < 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NL_PIX * PITCH > 4096 ? NL_PIX * PITCH : 4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int Hs = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws = a.mode0 ? (a.Win >> 1) : a.Win;
  const int NG = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + NG - 1) / NG;
  auto tile_of = [&](int round) { return (int)xcd_remap(blockIdx.x + (unsigned)round * NG, (unsigned)total_tiles); };

  // ---- weights -> registers: K step (tap, block cb, j): B[k = kq][n] = w[tap][16 cb + 4 kq + j][16 nb + n]
  float wreg[9][CB][4][NB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          wreg[t][cb][j][nb] = a.w[((size_t)t * CIN + 16 * cb + 4 * kq + j) * COUT + 16 * nb + m];

  // ---- fragment base: M block mb of wave w = output row 2 w + (mb >> 1), columns 16 (mb & 1) + m; slot kq of block cb
  const int abase = (2 * wave * N16_HW + m) * PITCH + 16 * kq;

  // ---- fill roles: element e = tid + 256 it = (halo pixel e / SLOTS, slot e % SLOTS); only the pixel index is kept
  // per iteration (registers are the scarce resource here), row / column are re-derived per tile
  const int f_pix0 = tid / SLOTS, f_slot = tid % SLOTS;   // pixel of iteration it = f_pix0 + it * (256 / SLOTS)
  const int f_ch = 4 * f_slot;
  f32x4 tf_sc = {1.f, 1.f, 1.f, 1.f}, tf_sh = {0.f, 0.f, 0.f, 0.f};
  if constexpr (TF) {
    tf_sc = *reinterpret_cast<const f32x4*>(a.in_scale + f_ch);
    tf_sh = *reinterpret_cast<const f32x4*>(a.in_shift + f_ch);
  }
  f32x4 rin[FILL];
  unsigned rvalid = 0;
  auto issue_loads = [&](int round) {
    const int sp = tile_of(round);
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int iy0 = ty * N16_TH - 1, ix0 = tx * N16_TW - 1;
    rvalid = 0;
#pragma unroll
    for (int it = 0; it < FILL; ++it) {
      const int pix = f_pix0 + it * (256 / SLOTS);
      const int hy = pix / N16_HW, hx = pix - hy * N16_HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = pix < NL_PIX && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      const int sy = a.mode0 ? (iy >> 1) : iy, sx = a.mode0 ? (ix >> 1) : ix;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(a.src0 + ((size_t)(b * Hs + sy) * Ws + sx) * CIN + f_ch);
      rvalid |= (ok ? 1u : 0u) << it;
      rin[it] = v;
    }
  };

  float s1[NB], s2[NB];
  float b_mu[NB], b_is[NB], b_sc[NB], b_sh[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    s1[nb] = s2[nb] = 0.f;
    b_mu[nb] = b_is[nb] = b_sc[nb] = b_sh[nb] = 0.f;
    if constexpr (EPI == 1) {
      b_mu[nb] = a.bnb.mean[16 * nb + m];
      b_is[nb] = a.bnb.invstd[16 * nb + m];
    }
    if constexpr (EPI == 1 || EPI == 4) {
      b_sc[nb] = a.bnb.act_scale[16 * nb + m];
      b_sh[nb] = a.bnb.act_shift[16 * nb + m];
    }
  }
  const bool want_stats = a.stats != nullptr;

  if (PREFETCH && my_tiles > 0) issue_loads(0);
  for (int round = 0; round < my_tiles; ++round) {
    if constexpr (!PREFETCH) issue_loads(round);
    __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
    for (int it = 0; it < FILL; ++it) {
      const int pix = f_pix0 + it * (256 / SLOTS);
      if (pix < NL_PIX) {
        f32x4 v = rin[it];
        if constexpr (TF) {
          if ((rvalid >> it) & 1u) {   // the arithmetic of the direct kernels' staging (mul, add, ReLU); padding stays zero
            v = v * tf_sc + tf_sh;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] < 0.f ? 0.f : v[k];
          }
        }
        *reinterpret_cast<f32x4*>(lds + pix * PITCH + 16 * f_slot) = v;
      }
    }
    __syncthreads();
    if (PREFETCH && round + 1 < my_tiles) issue_loads(round + 1);

    const int sp = tile_of(round);
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * N16_TH, ox0 = tx * N16_TW;
    const bool interior = oy0 + N16_TH <= a.Hin && ox0 + N16_TW <= a.Win;
    // output addresses: one base per lane and tile + the row stride; the rest are immediates
    const size_t o00 = (((size_t)b * a.Hin + oy0 + 2 * wave) * a.Win + ox0 + 4 * kq) * COUT + m;
    const size_t orow = (size_t)a.Win * COUT;

    f32x4 acc[4][NB];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // taps in kh -> kw -> channel order like the direct kernels; within a block the channel order is (j, kq) — an
    // exact fp32 fma chain either way, the rounding sequence differs from conv_fwd_n16_kernel's (kq, j) order
    // groups g = (tap, block): 4 fragment reads (one per M block) feed 16 NB MFMAs; the reads of group g + 1 are issued
    // before the MFMAs of group g (two register sets) and the scheduler is fenced per group — left alone, hipcc hoists
    // dozens of the 36 CB reads and spills (168 registers + 290 B of scratch for 32 -> 16)
    auto frag = [&](int g, f32x4 (&av)[4]) {
      const int t = g / CB, cb = g % CB;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        av[mb] = *reinterpret_cast<const f32x4*>(lds + abase + (((mb >> 1) + t / 3) * N16_HW + 16 * (mb & 1) + t % 3) * PITCH + 64 * cb);
    };
    f32x4 fav[2][4];
    frag(0, fav[0]);
#pragma unroll
    for (int g = 0; g < 9 * CB; ++g) {
      if (g + 1 < 9 * CB) frag(g + 1, fav[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      // K step j outermost: consecutive MFMAs go to DIFFERENT accumulators (v_mfma_f32_16x16x4_f32 issues every 32
      // cycles but a dependent one waits 40: with the M block outermost the four K steps of a block chained on one
      // accumulator and the loop ran 25 % slower)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fav[g & 1][mb][j], wreg[g / CB][g % CB][j][nb], acc[mb][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: D col = channel 16 nb + m, row = pixel column 4 kq + i of the M block
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int oy = oy0 + 2 * wave + (mb >> 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ox = ox0 + 16 * (mb & 1) + 4 * kq + i;
        if (interior || (oy < a.Hin && ox < a.Win)) {
          const size_t o = o00 + (mb >> 1) * orow + (16 * (mb & 1) + i) * COUT;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            float v = acc[mb][nb][i];
            if constexpr (EPI == 4) {
              v = v * b_sc[nb] + b_sh[nb];
              v = v < 0.f ? 0.f : v;
            }
            if constexpr (EPI == 1) {
              // (requesting y ahead of the MFMA loop — 16 NB registers — changed nothing: 792 / 787 vs 788 / 786 tiles/s)
              const float yv = a.bnb.y[o + 16 * nb];
              const float g = (yv * b_sc[nb] + b_sh[nb]) > 0.f ? v : 0.f;
              s1[nb] += g;
              s2[nb] += g * ((yv - b_mu[nb]) * b_is[nb]);
            } else if constexpr (EPI == 0) {
              s1[nb] += v;
              s2[nb] += v * v;
            }
            a.out[o + 16 * nb] = v;
          }
        }
      }
    }
  }

  // ---- one row of partial sums per workgroup; rows beyond the grid (the buffer has a.P rows) are written as zeros
  if (want_stats) {
    float* red = reinterpret_cast<float*>(lds);   // [2][4 waves][NB][16]
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float u1 = s1[nb], u2 = s2[nb];
      u1 += __shfl_xor(u1, 16, 64);
      u2 += __shfl_xor(u2, 16, 64);
      u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 32, 64);
      if (kq == 0) {
        red[(wave * NB + nb) * 16 + m] = u1;
        red[512 + (wave * NB + nb) * 16 + m] = u2;
      }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
      const int which = tid / COUT, c = tid % COUT, nb = c >> 4, n = c & 15;
      const float* r = red + which * 512 + nb * 16 + n;
      a.stats[((size_t)which * a.P + blockIdx.x) * COUT + c] = (r[0] + r[NB * 16]) + (r[2 * NB * 16] + r[3 * NB * 16]);
      for (int q = (int)blockIdx.x + NG; q < a.P; q += NG) a.stats[((size_t)which * a.P + q) * COUT + c] = 0.f;
    }
  }
}

static bool nl_enabled() {
  static const int on = [] {
    const char* e = getenv("DT_FP32_NARROW");
    return (e == nullptr || e[0] != '0') ? 1 : 0;
  }();
  return on != 0;
}

extern "C" int dt_conv2d_narrow_supported(const dt_conv_desc* d) {
  if (!nl_enabled() || d == nullptr) return 0;
  if (d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->C1 != 0 || d->cout_split != 0 || d->accumulate != 0) return 0;
  if ((d->C0 != 16 && d->C0 != 32) || (d->Cout != 16 && d->Cout != 32)) return 0;
  if (d->mode0 != 0 && d->mode0 != 1) return 0;
  if (d->Ho != d->Hin || d->Wo != d->Win || d->Wo < 32 || d->Ho < 8) return 0;
  return 1;
}

static int nl_tiles(const dt_conv_desc* d) { return d->B * dt_cdiv(d->Ho, N16_TH) * dt_cdiv(d->Wo, N16_TW); }

// rows of the statistics buffer: the same for every variant of a layer shape (an upper bound of the persistent grid)
int dt_conv2d_narrow_rows(const dt_conv_desc* d) {
  const int t = nl_tiles(d);
  return t < 8 * 256 ? t : 8 * 256;
}

template <class K>
static int nl_occupancy(K kernel) {
  hipFuncAttributes at;
  if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(kernel)) != hipSuccess) return 2;
  const int regs = ((at.numRegs + 7) / 8) * 8;
  const int by_regs = regs > 0 ? 512 / regs : 8;
  const int by_lds = at.sharedSizeBytes > 0 ? (int)(163840 / at.sharedSizeBytes) : 8;
  int occ = by_regs < by_lds ? by_regs : by_lds;
  return occ > 8 ? 8 : (occ < 1 ? 1 : occ);
}

template <int CB, int NB>
static int nl_launch(const NarrowLeanArgs& a, int total, bool tf, int epi, hipStream_t st) {
  // variants: 0 plain, 1 input transform, 2 BatchNorm-backward sums, 3 inference affine, 4 input transform + affine
  static int occ[5] = {0, 0, 0, 0, 0};
  const int v = epi == 1 ? 2 : (epi == 4 ? (tf ? 4 : 3) : (tf ? 1 : 0));
#define NL_CASE(vv, TFv, EPIv)                                                                              \
  if (v == vv) {                                                                                            \
    if (occ[vv] == 0) occ[vv] = nl_occupancy(conv3x3_f32_narrow_kernel<CB, NB, TFv, EPIv>);                   \
    const int grid = total < occ[vv] * 256 ? total : occ[vv] * 256;                                          \
    hipLaunchKernelGGL((conv3x3_f32_narrow_kernel<CB, NB, TFv, EPIv>), dim3((unsigned)grid), dim3(256), 0, st, a, total); \
    return DT_OK;                                                                                           \
  }
  NL_CASE(0, false, 0)
  NL_CASE(1, true, 0)
  NL_CASE(2, false, 1)
  NL_CASE(3, false, 4)
  NL_CASE(4, true, 4)
#undef NL_CASE
  return DT_EINVAL;
}

int dt_conv2d_narrow_launch_upc(const dt_conv_desc* d, const NarrowLeanArgs& a, int total, bool tf, int epi, hipStream_t st);

static bool nl_upc_enabled();
// 1 when the (supported) narrow layer runs in its sub-pixel form
int dt_conv2d_narrow_subpixel(const dt_conv_desc* d) {
  return d->mode0 == 1 && nl_upc_enabled() && ((d->Hin | d->Win) & 1) == 0 && d->C0 * d->Cout <= 512;
}
static bool nl_upc_enabled() {
  static const int on = [] {
    const char* e = getenv("DT_FP32_SUBPIXEL");
    return (e == nullptr || e[0] != '0') ? 1 : 0;
  }();
  return on != 0;
}

int dt_conv2d_narrow_launch(const dt_conv_desc* d, const float* src0, const float* w, float* out, float* stats,
                            const float* in_scale, const float* in_shift, hipStream_t st, const dt_bn_bwd_fuse* fuse,
                            bool affine) {
  DT_REQUIRE(dt_conv2d_narrow_supported(d), "conv_narrow: layer shape not supported");
  const bool tf = in_scale != nullptr, bnb = !affine && fuse != nullptr && fuse->y != nullptr;
  DT_REQUIRE(!bnb || (fuse->act == nullptr && fuse->act_scale && fuse->act_shift && stats),
             "conv_narrow: the fused BatchNorm-backward sums take a virtual activation and a stats buffer");
  DT_REQUIRE(!(tf && bnb), "conv_narrow: no input transform on the BatchNorm-backward form");
  DT_REQUIRE(!affine || (fuse && fuse->act_scale && fuse->act_shift && stats == nullptr),
             "conv_narrow: the inference epilogue needs scale / shift and takes no statistics");
  NarrowLeanArgs a;
  a.bnb = fuse ? *fuse : dt_bn_bwd_fuse{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a.src0 = src0; a.w = w; a.out = out; a.stats = stats; a.in_scale = in_scale; a.in_shift = in_shift;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.mode0 = d->mode0;
  a.tiles_x = dt_cdiv(d->Wo, N16_TW); a.tiles_y = dt_cdiv(d->Ho, N16_TH);
  a.P = dt_conv2d_narrow_rows(d);
  const int total = nl_tiles(d), epi = affine ? 4 : (bnb ? 1 : 0);
  int rc;
  if (epi != 1 && dt_conv2d_narrow_subpixel(d)) {
    // nearest-upsampled input: the sub-pixel form (4 combined taps per output parity instead of 9); 32 -> 32 would
    // need 256 weight registers and keeps the 9-tap form
    rc = dt_conv2d_narrow_launch_upc(d, a, total, tf, epi, st);
    if (rc != DT_OK) return rc;
    DT_LAUNCH_CHECK();
    return DT_OK;
  }
  if (d->C0 == 16 && d->Cout == 16) rc = nl_launch<1, 1>(a, total, tf, epi, st);
  else if (d->C0 == 16) rc = nl_launch<1, 2>(a, total, tf, epi, st);
  else if (d->Cout == 16) rc = nl_launch<2, 1>(a, total, tf, epi, st);
  else rc = nl_launch<2, 2>(a, total, tf, epi, st);
  if (rc != DT_OK) return rc;
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// inference form: out = relu(conv(src) * scale + shift) in ONE launch for the narrow decoder layers — the ATen chain
// conv2d -> batch_norm(eval) -> relu_ of smp's Conv2dReLU (deadtrees/network/extra/modules.py:74-92 twin); same
// arithmetic as dt_conv2d followed by dt_bn_act (bit-identical), the raw output is never stored
extern "C" int dt_conv2d_narrow_affine(const dt_conv_desc* d, const float* src0, const float* w_hwio, float* out,
                                       const float* scale, const float* shift, const float* in_scale,
                                       const float* in_shift, void* stream) {
  DT_REQUIRE(d && src0 && w_hwio && out && scale && shift, "conv_narrow_affine: null pointer");
  DT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv_narrow_affine: in_scale/in_shift must come together");
  dt_bn_bwd_fuse f{nullptr, nullptr, nullptr, scale, shift, nullptr};
  return dt_conv2d_narrow_launch(d, src0, w_hwio, out, nullptr, in_scale, in_shift, (hipStream_t)stream, &f, true);
}

// ================================================================================================================
// Round 3: the same layers when the input is NEAREST-UPSAMPLED (mode0 = 1: dec4.conv1, 32 -> 16 over up(dec3)) — the
// "sub-pixel" form.  A 3x3 convolution over an image in which every 2x2 block repeats one source pixel reads only 2x2
// DISTINCT source pixels per output pixel: for output parity (a, b) = (row & 1, column & 1)
//     y[2Y + a][2X + b] = sum_{dy, dx in {0,1}} W'_ab[dy][dx] . x[Y + a - 1 + dy][X + b - 1 + dx],
//     W'_ab[dy][dx] = sum_{kh in R(a,dy)} sum_{kw in R(b,dx)} W[kh][kw],   R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}
// — 4 taps instead of 9: 2.25x fewer multiplies on a layer that is MFMA-bound in fp32 (exact algebra, zero padding
// included: an out-of-range source pixel is exactly the padding of the up-sampled image; the pre-summed weights round
// once more, ~1e-7 relative).  The 16 combined matrices live in registers (64 CB NB of them); a workgroup's 8 x 32
// output tile comes from a 6 x 18 LOW-RESOLUTION halo (17 KB of LDS instead of 54); an M block = 16 output pixels of
// one parity in one output row, so the 16 (parity, tap) products of a source row need only 9 distinct fragment reads.
template <int CB, int NB, bool TF, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_f32_upc_kernel(const NarrowLeanArgs a, const int total_tiles) {
  constexpr int CIN = 16 * CB, COUT = 16 * NB;
  constexpr int PITCH = CB == 1 ? 96 : 160;
  constexpr int SLOTS = 4 * CB;
  constexpr int LH = N16_TH / 2 + 2, LW = N16_TW / 2 + 2, LPIX = LH * LW;   // 6 x 18 low-resolution halo pixels
  constexpr int FILL = (LPIX * SLOTS + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LPIX * PITCH > 4096 ? LPIX * PITCH : 4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int Hs = a.Hin >> 1, Ws = a.Win >> 1;
  const int NG = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + NG - 1) / NG;
  auto tile_of = [&](int round) { return (int)xcd_remap(blockIdx.x + (unsigned)round * NG, (unsigned)total_tiles); };

  // ---- combined weights -> registers: wc[a][dy][b][dx][cb][j][nb], built from the lane's 9 original taps
  float wc[2][2][2][2][CB][4][NB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float w9[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) w9[t / 3][t % 3] = a.w[((size_t)t * CIN + 16 * cb + 4 * kq + j) * COUT + 16 * nb + m];
        // rows first: R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}
        float rw[2][2][3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          rw[0][0][kw] = w9[0][kw];
          rw[0][1][kw] = w9[1][kw] + w9[2][kw];
          rw[1][0][kw] = w9[0][kw] + w9[1][kw];
          rw[1][1][kw] = w9[2][kw];
        }
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            wc[pa][dy][0][0][cb][j][nb] = rw[pa][dy][0];
            wc[pa][dy][0][1][cb][j][nb] = rw[pa][dy][1] + rw[pa][dy][2];
            wc[pa][dy][1][0][cb][j][nb] = rw[pa][dy][0] + rw[pa][dy][1];
            wc[pa][dy][1][1][cb][j][nb] = rw[pa][dy][2];
          }
      }

  // wave w = low-resolution row Y = w of the tile; fragment (r, c): halo pixel (w + r, m + c), slot kq of block cb
  const int abase = (wave * LW + m) * PITCH + 16 * kq;
  const int f_pix0 = tid / SLOTS, f_slot = tid % SLOTS, f_ch = 4 * f_slot;
  f32x4 tf_sc = {1.f, 1.f, 1.f, 1.f}, tf_sh = {0.f, 0.f, 0.f, 0.f};
  if constexpr (TF) {
    tf_sc = *reinterpret_cast<const f32x4*>(a.in_scale + f_ch);
    tf_sh = *reinterpret_cast<const f32x4*>(a.in_shift + f_ch);
  }
  f32x4 rin[FILL];
  unsigned rvalid = 0;
  auto issue_loads = [&](int round) {
    const int sp = tile_of(round);
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int y0 = ty * (N16_TH / 2) - 1, x0 = tx * (N16_TW / 2) - 1;
    rvalid = 0;
#pragma unroll
    for (int it = 0; it < FILL; ++it) {
      const int pix = f_pix0 + it * (256 / SLOTS);
      const int hy = pix / LW, hx = pix - hy * LW;
      const int sy = y0 + hy, sx = x0 + hx;
      const bool ok = pix < LPIX && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(a.src0 + ((size_t)(b * Hs + sy) * Ws + sx) * CIN + f_ch);
      rvalid |= (ok ? 1u : 0u) << it;
      rin[it] = v;
    }
  };
  float s1[NB], s2[NB], b_sc[NB], b_sh[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    s1[nb] = s2[nb] = 0.f;
    b_sc[nb] = b_sh[nb] = 0.f;
    if constexpr (EPI == 4) {
      b_sc[nb] = a.bnb.act_scale[16 * nb + m];
      b_sh[nb] = a.bnb.act_shift[16 * nb + m];
    }
  }
  const bool want_stats = a.stats != nullptr;

  if (my_tiles > 0) issue_loads(0);
  for (int round = 0; round < my_tiles; ++round) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < FILL; ++it) {
      const int pix = f_pix0 + it * (256 / SLOTS);
      if (pix < LPIX) {
        f32x4 v = rin[it];
        if constexpr (TF) {
          if ((rvalid >> it) & 1u) {
            v = v * tf_sc + tf_sh;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] < 0.f ? 0.f : v[k];
          }
        }
        *reinterpret_cast<f32x4*>(lds + pix * PITCH + 16 * f_slot) = v;
      }
    }
    __syncthreads();
    if (round + 1 < my_tiles) issue_loads(round + 1);

    f32x4 acc[2][2][NB];   // [row parity a][column parity b]
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[pa][pb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments f = (cb, r, c), one ahead in registers; fragment (r, c) serves every (a, dy) with a + dy = r and every
    // (b, dx) with b + dx = c.  Taps in (dy, dx) order per parity = kh -> kw order of the direct form.
    auto frag = [&](int f) -> f32x4 {
      const int cb = f / 9, r = (f % 9) / 3, c = f % 3;
      return *reinterpret_cast<const f32x4*>(lds + abase + (r * LW + c) * PITCH + 64 * cb);
    };
    f32x4 fav[2];
    fav[0] = frag(0);
#pragma unroll
    for (int f = 0; f < 9 * CB; ++f) {
      if (f + 1 < 9 * CB) fav[(f + 1) & 1] = frag(f + 1);
      __builtin_amdgcn_sched_barrier(0);
      const int cb = f / 9, r = (f % 9) / 3, c = f % 3;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            const int dy = r - pa, dx = c - pb;
            if (dy >= 0 && dy < 2 && dx >= 0 && dx < 2) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[pa][pb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fav[f & 1][j], wc[pa][dy][pb][dx][cb][j][nb],
                                                                       acc[pa][pb][nb], 0, 0, 0);
            }
          }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: M block (a, b): output row 2 w + a, columns 2 (4 kq + i) + b of the tile
    const int sp = tile_of(round);
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * N16_TH, ox0 = tx * N16_TW;
    const bool interior = oy0 + N16_TH <= a.Hin && ox0 + N16_TW <= a.Win;
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
      const int oy = oy0 + 2 * wave + pa;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ox = ox0 + 2 * (4 * kq + i) + pb;
          if (interior || (oy < a.Hin && ox < a.Win)) {
            const size_t o = (((size_t)b * a.Hin + oy) * a.Win + ox) * COUT + m;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              float v = acc[pa][pb][nb][i];
              if constexpr (EPI == 4) {
                v = v * b_sc[nb] + b_sh[nb];
                v = v < 0.f ? 0.f : v;
              } else {
                s1[nb] += v;
                s2[nb] += v * v;
              }
              a.out[o + 16 * nb] = v;
            }
          }
        }
    }
  }
  if (want_stats) {
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float u1 = s1[nb], u2 = s2[nb];
      u1 += __shfl_xor(u1, 16, 64);
      u2 += __shfl_xor(u2, 16, 64);
      u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 32, 64);
      if (kq == 0) {
        red[(wave * NB + nb) * 16 + m] = u1;
        red[512 + (wave * NB + nb) * 16 + m] = u2;
      }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
      const int which = tid / COUT, c = tid % COUT, nb = c >> 4, n = c & 15;
      const float* r = red + which * 512 + nb * 16 + n;
      a.stats[((size_t)which * a.P + blockIdx.x) * COUT + c] = (r[0] + r[NB * 16]) + (r[2 * NB * 16] + r[3 * NB * 16]);
      for (int q = (int)blockIdx.x + NG; q < a.P; q += NG) a.stats[((size_t)which * a.P + q) * COUT + c] = 0.f;
    }
  }
}

// the sub-pixel form serves the up-sampled layers without fused BatchNorm-backward sums (forward / inference of dec4.conv1)
template <int CB, int NB>
static int nl_launch_upc(const NarrowLeanArgs& a, int total, bool tf, int epi, hipStream_t st) {
  static int occ[4] = {0, 0, 0, 0};
  const int v = (epi == 4 ? 2 : 0) + (tf ? 1 : 0);
#define NLU_CASE(vv, TFv, EPIv)                                                                          \
  if (v == vv) {                                                                                         \
    if (occ[vv] == 0) occ[vv] = nl_occupancy(conv3x3_f32_upc_kernel<CB, NB, TFv, EPIv>);                   \
    const int grid = total < occ[vv] * 256 ? total : occ[vv] * 256;                                       \
    hipLaunchKernelGGL((conv3x3_f32_upc_kernel<CB, NB, TFv, EPIv>), dim3((unsigned)grid), dim3(256), 0, st, a, total); \
    return DT_OK;                                                                                        \
  }
  NLU_CASE(0, false, 0)
  NLU_CASE(1, true, 0)
  NLU_CASE(2, false, 4)
  NLU_CASE(3, true, 4)
#undef NLU_CASE
  return DT_EINVAL;
}

int dt_conv2d_narrow_launch_upc(const dt_conv_desc* d, const NarrowLeanArgs& a, int total, bool tf, int epi, hipStream_t st) {
  if (d->C0 == 16 && d->Cout == 16) return nl_launch_upc<1, 1>(a, total, tf, epi, st);
  if (d->C0 == 16) return nl_launch_upc<1, 2>(a, total, tf, epi, st);
  if (d->Cout == 16) return nl_launch_upc<2, 1>(a, total, tf, epi, st);
  // 32 -> 32 would need 256 weight registers (it spills): dt_conv2d_narrow_subpixel keeps that shape on the 9-tap kernel
  dt_set_error("conv_narrow: no sub-pixel form for 32 -> 32 channels");
  return DT_ENOSYS;
}

// ================================================================================================================
// Data gradient of the same up-sampled layer, sub-pixel form: the transpose of conv3x3_f32_upc_kernel.  The reference
// chain convolution_backward(input) of the 3x3 convolution at full resolution (9 taps x 4 pixels per source pixel)
// followed by the backward of F.interpolate(nearest, x2) (a 2x2 sum) collapses into ONE 4x4 stride-2 convolution over
// dY with 16 combined weight matrices:
//     g[Y][X] = sum_{r, c = 0..3} V[r][c] . dY[2Y - 1 + r][2X - 1 + c],   V[r][c] = W'_{a(r) b(c)}[dy(r)][dx(c)]^T,
//     r = 3 - a - 2 dy  (r = 0: (a, dy) = (1, 1); 1: (0, 1); 2: (1, 0); 3: (0, 0)), columns alike
// — 16 tap products per source pixel instead of 36, and the full-resolution gradient of the up-sampled tensor
// (32 channels at 512x512: 1 GB written and read back per 32-tile step) never exists.  The BatchNorm-backward sums of
// the layer the gradient belongs to (what dt_upsample2x_bwd_bn fused into the 2x2-sum pass) ride in the epilogue.
// dY tile in LDS split by column parity ([row][parity][17 columns]) so that the stride-2 fragment reads are unit-stride.
struct UpcDgradArgs {
  const float* dy;     // [B][H][W][CO] gradient of the convolution's output (full resolution)
  const float* w;      // [9][CI][CO] the FORWARD weights (HWIO)
  float* gx;           // [B][H/2][W/2][CI] gradient of the low-resolution input
  float* stats;        // [2][P][CI] fused BatchNorm-backward sums, or null
  dt_bn_bwd_fuse bnb;
  int B, H, W, tiles_x, tiles_y, P;
};

template <int CIB /* CI / 16 */, int COB /* CO / 16 */, bool BNB>
__global__ __launch_bounds__(256, 2) void conv3x3_f32_upc_dgrad_kernel(const UpcDgradArgs a, const int total_tiles) {
  constexpr int CI = 16 * CIB, CO = 16 * COB;
  constexpr int PITCH = COB == 1 ? 96 : 160;
  constexpr int SLOTS = 4 * COB;
  constexpr int TR = 10, TC = 34, HC = 17;                 // dY tile: 10 rows x 34 columns = [row][parity][17]
  constexpr int TPIX = TR * TC;
  constexpr int FILL = (TPIX * SLOTS + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char lds[TPIX * PITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int Hs = a.H >> 1, Ws = a.W >> 1;
  const int NG = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + NG - 1) / NG;
  auto tile_of = [&](int round) { return (int)xcd_remap(blockIdx.x + (unsigned)round * NG, (unsigned)total_tiles); };

  // ---- V[r][c] -> registers: B operand of K step (r, c, block cob, j): B[k = kq][n] = V[r][c][co = 16 cob + 4 kq + j][ci = 16 nb + m]
  float vreg[4][4][COB][4][CIB];
#pragma unroll
  for (int cob = 0; cob < COB; ++cob)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < CIB; ++nb) {
        float w9[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) w9[t / 3][t % 3] = a.w[((size_t)t * CI + 16 * nb + m) * CO + 16 * cob + 4 * kq + j];
        // row classes r = 0..3 <-> (a, dy) = (1,1), (0,1), (1,0), (0,0) <-> tap rows {2}, {1,2}, {0,1}, {0}
        float rw[4][3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          rw[0][kw] = w9[2][kw];
          rw[1][kw] = w9[1][kw] + w9[2][kw];
          rw[2][kw] = w9[0][kw] + w9[1][kw];
          rw[3][kw] = w9[0][kw];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          vreg[r][0][cob][j][nb] = rw[r][2];
          vreg[r][1][cob][j][nb] = rw[r][1] + rw[r][2];
          vreg[r][2][cob][j][nb] = rw[r][0] + rw[r][1];
          vreg[r][3][cob][j][nb] = rw[r][0];
        }
      }

  // wave w = low-resolution row w of the 4 x 16 tile; tap (r, c): dY row 2 w + r, column 2 m + c of the tile
  const int abase = ((2 * wave) * 2 * HC + m) * PITCH + 16 * kq;
  const int f_pix0 = tid / SLOTS, f_slot = tid % SLOTS, f_ch = 4 * f_slot;
  f32x4 rin[FILL];
  auto issue_loads = [&](int round) {
    const int sp = tile_of(round);
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int y0 = 8 * ty - 1, x0 = 32 * tx - 1;
#pragma unroll
    for (int it = 0; it < FILL; ++it) {
      const int pix = f_pix0 + it * (256 / SLOTS);          // linear (row, column) index of the tile
      const int hy = pix / TC, hx = pix - hy * TC;
      const int oy = y0 + hy, ox = x0 + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pix < TPIX && (unsigned)oy < (unsigned)a.H && (unsigned)ox < (unsigned)a.W)
        v = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)(b * a.H + oy) * a.W + ox) * CO + f_ch);
      rin[it] = v;
    }
  };
  float s1[CIB], s2[CIB], b_mu[CIB], b_is[CIB], b_sc[CIB], b_sh[CIB];
#pragma unroll
  for (int nb = 0; nb < CIB; ++nb) {
    s1[nb] = s2[nb] = b_mu[nb] = b_is[nb] = b_sc[nb] = b_sh[nb] = 0.f;
    if constexpr (BNB) {
      b_mu[nb] = a.bnb.mean[16 * nb + m];
      b_is[nb] = a.bnb.invstd[16 * nb + m];
      b_sc[nb] = a.bnb.act_scale[16 * nb + m];
      b_sh[nb] = a.bnb.act_shift[16 * nb + m];
    }
  }

  if (my_tiles > 0) issue_loads(0);
  for (int round = 0; round < my_tiles; ++round) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < FILL; ++it) {
      const int pix = f_pix0 + it * (256 / SLOTS);
      if (pix < TPIX) {
        const int hy = pix / TC, hx = pix - hy * TC;
        *reinterpret_cast<f32x4*>(lds + ((hy * 2 + (hx & 1)) * HC + (hx >> 1)) * PITCH + 16 * f_slot) = rin[it];
      }
    }
    __syncthreads();
    if (round + 1 < my_tiles) issue_loads(round + 1);

    const int sp = tile_of(round);
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int Y = 4 * ty + wave;
    f32x4 acc[CIB];
#pragma unroll
    for (int nb = 0; nb < CIB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto frag = [&](int f) -> f32x4 {
      const int cob = f / 16, r = (f % 16) / 4, c = f % 4;
      return *reinterpret_cast<const f32x4*>(lds + abase + ((r * 2 + (c & 1)) * HC + (c >> 1)) * PITCH + 64 * cob);
    };
    f32x4 fav[2];
    fav[0] = frag(0);
#pragma unroll
    for (int f = 0; f < 16 * COB; ++f) {
      if (f + 1 < 16 * COB) fav[(f + 1) & 1] = frag(f + 1);
      __builtin_amdgcn_sched_barrier(0);
      const int cob = f / 16, r = (f % 16) / 4, c = f % 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < CIB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fav[f & 1][j], vreg[r][c][cob][j][nb], acc[nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: low-resolution pixel (4 ty + w, 16 tx + 4 kq + i), channel 16 nb + m
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int X = 16 * tx + 4 * kq + i;
      if (Y < Hs && X < Ws) {
        const size_t o = (((size_t)b * Hs + Y) * Ws + X) * CI + m;
#pragma unroll
        for (int nb = 0; nb < CIB; ++nb) {
          const float v = acc[nb][i];
          if constexpr (BNB) {
            const float yv = a.bnb.y[o + 16 * nb];
            const float g = (yv * b_sc[nb] + b_sh[nb]) > 0.f ? v : 0.f;
            s1[nb] += g;
            s2[nb] += g * ((yv - b_mu[nb]) * b_is[nb]);
          }
          a.gx[o + 16 * nb] = v;
        }
      }
    }
  }
  if (BNB && a.stats != nullptr) {
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < CIB; ++nb) {
      float u1 = s1[nb], u2 = s2[nb];
      u1 += __shfl_xor(u1, 16, 64);
      u2 += __shfl_xor(u2, 16, 64);
      u1 += __shfl_xor(u1, 32, 64);
      u2 += __shfl_xor(u2, 32, 64);
      if (kq == 0) {
        red[(wave * CIB + nb) * 16 + m] = u1;
        red[512 + (wave * CIB + nb) * 16 + m] = u2;
      }
    }
    __syncthreads();
    if (tid < 2 * CI) {
      const int which = tid / CI, c = tid % CI, nb = c >> 4, n = c & 15;
      const float* r = red + which * 512 + nb * 16 + n;
      a.stats[((size_t)which * a.P + blockIdx.x) * CI + c] = (r[0] + r[CIB * 16]) + (r[2 * CIB * 16] + r[3 * CIB * 16]);
      for (int q = (int)blockIdx.x + NG; q < a.P; q += NG) a.stats[((size_t)which * a.P + q) * CI + c] = 0.f;
    }
  }
}

// `d` describes the FORWARD convolution (mode0 = 1, C0 = CI low-resolution input channels, Cout = CO, Hin x Win = the
// full-resolution map)
extern "C" int dt_conv2d_upsampled_dgrad_supported(const dt_conv_desc* d) {
  static const int on = [] {
    const char* e = getenv("DT_FP32_SUBPIXEL_DGRAD");
    return (e == nullptr || e[0] != '0') ? 1 : 0;
  }();
  if (!on || !nl_enabled() || !nl_upc_enabled() || d == nullptr) return 0;
  if (d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->mode0 != 1 || d->C1 != 0 || d->cout_split != 0) return 0;
  if ((d->C0 != 16 && d->C0 != 32) || (d->Cout != 16 && d->Cout != 32) || d->C0 * d->Cout > 512) return 0;
  if (((d->Hin | d->Win) & 1) != 0 || d->Ho != d->Hin || d->Wo != d->Win || d->Win < 32 || d->Hin < 8) return 0;
  return 1;
}

static int upcd_tiles(const dt_conv_desc* d) { return d->B * dt_cdiv(d->Hin / 2, 4) * dt_cdiv(d->Win / 2, 16); }

extern "C" int dt_conv2d_upsampled_dgrad_rows(const dt_conv_desc* d) {
  if (!dt_conv2d_upsampled_dgrad_supported(d)) return 0;
  const int t = upcd_tiles(d);
  return t < 8 * 256 ? t : 8 * 256;
}

template <int CIB, int COB>
static int upcd_launch(const UpcDgradArgs& a, int total, bool bnb, hipStream_t st) {
  static int occ[2] = {0, 0};
  if (bnb) {
    if (occ[1] == 0) occ[1] = nl_occupancy(conv3x3_f32_upc_dgrad_kernel<CIB, COB, true>);
    const int grid = total < occ[1] * 256 ? total : occ[1] * 256;
    hipLaunchKernelGGL((conv3x3_f32_upc_dgrad_kernel<CIB, COB, true>), dim3((unsigned)grid), dim3(256), 0, st, a, total);
  } else {
    if (occ[0] == 0) occ[0] = nl_occupancy(conv3x3_f32_upc_dgrad_kernel<CIB, COB, false>);
    const int grid = total < occ[0] * 256 ? total : occ[0] * 256;
    hipLaunchKernelGGL((conv3x3_f32_upc_dgrad_kernel<CIB, COB, false>), dim3((unsigned)grid), dim3(256), 0, st, a, total);
  }
  return DT_OK;
}

// gx = d loss / d x for y = conv3x3(nearest_upsample_x2(x)): replaces dt_conv2d (data-gradient form) + dt_upsample2x_bwd
// (_bn) of the generic chain; red (with fuse) receives the BatchNorm-backward sums like dt_upsample2x_bwd_bn
extern "C" int dt_conv2d_upsampled_dgrad(const dt_conv_desc* d, const float* dy, const float* w_hwio, float* gx, float* red,
                                         const dt_bn_bwd_fuse* fuse, void* stream) {
  DT_REQUIRE(d && dy && w_hwio && gx, "conv_upsampled_dgrad: null pointer");
  DT_REQUIRE(dt_conv2d_upsampled_dgrad_supported(d), "conv_upsampled_dgrad: layer shape not supported");
  const bool bnb = fuse != nullptr && fuse->y != nullptr;
  DT_REQUIRE(!bnb || (red && fuse->mean && fuse->invstd && fuse->act_scale && fuse->act_shift),
             "conv_upsampled_dgrad: the fused sums need red, mean, invstd and the activation's scale / shift");
  UpcDgradArgs a;
  a.bnb = fuse ? *fuse : dt_bn_bwd_fuse{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a.dy = dy; a.w = w_hwio; a.gx = gx; a.stats = bnb ? red : nullptr;
  a.B = d->B; a.H = d->Hin; a.W = d->Win;
  a.tiles_x = dt_cdiv(d->Win / 2, 16); a.tiles_y = dt_cdiv(d->Hin / 2, 4);
  a.P = dt_conv2d_upsampled_dgrad_rows(d);
  const int total = upcd_tiles(d);
  hipStream_t st = (hipStream_t)stream;
  if (d->C0 == 16 && d->Cout == 16) upcd_launch<1, 1>(a, total, bnb, st);
  else if (d->C0 == 32 && d->Cout == 16) upcd_launch<2, 1>(a, total, bnb, st);
  else upcd_launch<1, 2>(a, total, bnb, st);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
