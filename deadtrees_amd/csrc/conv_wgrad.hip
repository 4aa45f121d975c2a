// Weight gradient of the NHWC convolutions on the fp32 matrix cores of gfx950.
//
// Replaces the ATen convolution_backward(weight) launched by autograd for every conv of
// smp.Unet(resnet34) (reference: Lightning's loss.backward() on the graph built at
// deadtrees/network/segmodel.py:214).  dW[tap][ci][co] = sum over pixels of x[pix+tap][ci]*dy[pix][co]
// is a GEMM whose K dimension is the pixel index: M = 32 input channels, N = 32 output channels,
// K = 2 pixels per v_mfma_f32_32x32x2_f32.  Both operands are staged once per pixel tile in LDS
// (pixel-major, channel-contiguous: the 32 lanes of a fragment read 32 consecutive dwords) and the x
// halo tile is re-used by all KS*KS taps.  The pixel dimension is split over workgroups (and
// optionally over the waves of a workgroup); each split writes one fp32 partial slab and a second
// kernel sums the slabs in a fixed order -> run-to-run deterministic, no float atomics.
#include "common.h"

struct WgradArgs {
  const float* src0;
  const float* src1;
  const float* in_scale;  // optional fused BatchNorm-apply + ReLU on source 0 (see conv_fwd.hip)
  const float* in_shift;
  const float* dy;
  const __bf16* dy_bf16;  // stem only: dy given in bf16 (bf16 training path)
  float* ws;  // [parts][taps][Cin][Cout]
  int B, Hin, Win, C0, C1, mode0;
  int Ho, Wo, Cout, pad;
  int tiles_x, tiles_y, T;  // pixel tiles
  int ci_blocks, co_blocks, ksplit;
};

template <int KS, int STRIDE, int TW, int TPX>
struct WGeom {
  static constexpr int TH = TPX / TW;
  static constexpr int LS = (KS == 1) ? 1 : STRIDE;
  static constexpr int GS = (KS == 1) ? STRIDE : 1;
  static constexpr int HALO_H = (TH - 1) * LS + KS;
  static constexpr int HALO_W = (TW - 1) * LS + KS;
  static constexpr int TAPS = KS * KS;
};

template <int KS, int STRIDE, int TW, int TPX, int WCI, int WCO, bool TF = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
  using G = WGeom<KS, STRIDE, TW, TPX>;
  constexpr int WK = 4 / (WCI * WCO);
  constexpr int CIW = WCI * 32, COW = WCO * 32;
  constexpr int X_ELEMS = G::HALO_H * G::HALO_W * CIW;
  constexpr int Y_ELEMS = TPX * COW;
  constexpr int ROWS = G::TH / WK;
  static_assert(G::TH % WK == 0, "rows must split evenly over the K-waves");
  __shared__ __attribute__((aligned(16))) float lds[X_ELEMS + Y_ELEMS];
  float* lx = lds;
  float* ly = lds + X_ELEMS;

  const int wgid = (int)xcd_remap(blockIdx.x, gridDim.x);  // (ci,co) blocks of one k-split share an XCD's L2
  const int blk = wgid % (a.ci_blocks * a.co_blocks);
  const int ks = wgid / (a.ci_blocks * a.co_blocks);
  const int cib = blk / a.co_blocks, cob = blk % a.co_blocks;
  const int ci0 = cib * CIW, co0 = cob * COW;
  const int Cin = a.C0 + a.C1;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = lane >> 5, r = lane & 31;
  const int wk = wave / (WCI * WCO);
  const int wci = (wave % (WCI * WCO)) / WCO, wco = wave % WCO;

  const int xbase = (wk * G::LS * G::HALO_W + s * G::LS) * CIW + wci * 32 + r;
  const int ybase = (wk * TW + s) * COW + wco * 32 + r;

  f32x16 acc[G::TAPS];
#pragma unroll
  for (int t = 0; t < G::TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // Software-pipelined staging (same scheme as conv_fwd): the global loads of the next pixel tile are in
  // flight, in registers, while the matrix cores work on the current one.
  constexpr int QX = CIW / 4, QY = COW / 4;
  constexpr int X_TOTAL = G::HALO_H * G::HALO_W * QX, X_IT = (X_TOTAL + 255) / 256;
  constexpr int Y_TOTAL = TPX * QY, Y_IT = (Y_TOTAL + 255) / 256;
  static_assert(256 % QX == 0 && 256 % QY == 0, "lane->quad mapping must be iteration invariant");
  const int qx = tid % QX, px0 = tid / QX;
  const int qy = tid % QY, py0 = tid / QY;
  // channel quad of x handled by this lane: source select is tile invariant
  const int cx = ci0 + 4 * qx;
  const bool x_use0 = cx < a.C0;
  const float* xsrc = x_use0 ? a.src0 : a.src1;
  const int xC = x_use0 ? a.C0 : a.C1;
  const int xcc = x_use0 ? cx : cx - a.C0;
  const int xmode = x_use0 ? a.mode0 : 0;
  const int xHs = xmode ? (a.Hin >> 1) : a.Hin, xWs = xmode ? (a.Win >> 1) : a.Win;
  const bool x_ch_ok = cx < Cin;
  // fused BatchNorm-apply + ReLU on source 0: scale/shift of this lane's channel quad parked in LDS so that no
  // registers are held across the MFMA loop; applied when the staged registers are written to LDS
  const bool x_tf = TF && x_use0 && x_ch_ok;
  __shared__ __attribute__((aligned(16))) float lds_tf[TF ? 2 * CIW : 4];
  if (TF && tid < CIW && ci0 + tid < a.C0) {
    lds_tf[tid] = a.in_scale[ci0 + tid];
    lds_tf[CIW + tid] = a.in_shift[ci0 + tid];
  }
  unsigned xvalid = 0;
  const int cy = co0 + 4 * qy;
  const bool y_ch_ok = cy < a.Cout;
  f32x4 rx[X_IT], ry[Y_IT];
  auto issue_loads = [&](int tile) {
    const int tx = tile % a.tiles_x;
    const int ty = (tile / a.tiles_x) % a.tiles_y;
    const int b = tile / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * G::TH, ox0 = tx * TW;
    const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int pix = px0 + it * (256 / QX);
      const int hy = pix / G::HALO_W, hx = pix - hy * G::HALO_W;
      const int iy = iy0 + hy * G::GS, ix = ix0 + hx * G::GS;
      bool ok = x_ch_ok && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win &&
                (X_TOTAL % 256 == 0 || pix < G::HALO_H * G::HALO_W);
      if (xmode == 2) ok = ok && (((iy | ix) & 1) == 0);
      const int sy = xmode ? (iy >> 1) : iy, sx = xmode ? (ix >> 1) : ix;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(xsrc + (((size_t)b * xHs + sy) * xWs + sx) * xC + xcc);
      if constexpr (TF) {
        if (it == 0) xvalid = 0;
        xvalid |= (ok ? 1u : 0u) << it;
      }
      rx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int pix = py0 + it * (256 / QY);
      const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (y_ch_ok && oy < a.Ho && ox < a.Wo && (Y_TOTAL % 256 == 0 || pix < TPX))
        v = *reinterpret_cast<const f32x4*>(a.dy + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + cy);
      ry[it] = v;
    }
  };
  auto write_lds = [&]() {
    f32x4 x_sc = {1.f, 1.f, 1.f, 1.f}, x_sh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (TF) {
      if (x_tf) {
        x_sc = *reinterpret_cast<const f32x4*>(lds_tf + 4 * qx);
        x_sh = *reinterpret_cast<const f32x4*>(lds_tf + CIW + 4 * qx);
      }
    }
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int pix = px0 + it * (256 / QX);
      if (X_TOTAL % 256 == 0 || pix < G::HALO_H * G::HALO_W) {
        f32x4 v = rx[it];
        if (TF && x_tf && ((xvalid >> it) & 1u)) {   // padding stays zero
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
      if (Y_TOTAL % 256 == 0 || pix < TPX) *reinterpret_cast<f32x4*>(ly + pix * COW + 4 * qy) = ry[it];
    }
  };
  // register budget: 16*TAPS accumulators + 4 per staged float4; prefetch across the MFMAs only when both fit
  // in the 256 registers that two waves per SIMD leave each (otherwise: batched loads, then write, then MFMA)
  constexpr bool PREFETCH = (16 * G::TAPS + 4 * (X_IT + Y_IT)) <= 190;
  if (PREFETCH && ks < a.T) issue_loads(ks);
  for (int tile = ks; tile < a.T; tile += a.ksplit) {
    __syncthreads();
    if (!PREFETCH) issue_loads(tile);
    write_lds();
    __syncthreads();
    if (PREFETCH && tile + a.ksplit < a.T) issue_loads(tile + a.ksplit);
#pragma unroll
    for (int yy = 0; yy < ROWS; ++yy) {
#pragma unroll
      for (int j = 0; j < TW / 2; ++j) {
        const float bv = ly[ybase + (yy * WK * TW + 2 * j) * COW];
#pragma unroll
        for (int t = 0; t < G::TAPS; ++t) {
          const int kh = t / KS, kw = t % KS;
          const float av = lx[xbase + ((yy * WK * G::LS + kh) * G::HALO_W + 2 * j * G::LS + kw) * CIW];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // ---- write the partial slab of this (k-split, k-wave)
  const int part = ks * WK + wk;
  const int co = co0 + wco * 32 + r;
  if (co < a.Cout) {
#pragma unroll
    for (int t = 0; t < G::TAPS; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = ci0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * s;
        if (ci < Cin) a.ws[(((size_t)part * G::TAPS + t) * Cin + ci) * a.Cout + co] = acc[t][i];
      }
    }
  }
}

// dW[e] = sum_part ws[part][e], fixed order, fp64 accumulate
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           int parts, int64_t E) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
    double s = 0.0;
    for (int p = 0; p < parts; ++p) s += (double)ws[(size_t)p * E + e];
    dw[e] = (float)s;
  }
}

// ---------------------------------------------------------------- stem (7x7 / stride 2, Cin <= 4)
// M index = kw*4 + ci (28 of 32 rows used), one accumulator per kh: the [pixel][4] LDS image makes the
// 7 kw-taps x 4 channels of a window row 28 consecutive dwords.
#define STEM_TW 32
#define STEM_TH 4
__global__ __launch_bounds__(256, 2) void conv_wgrad_stem_kernel(const WgradArgs a) {
  constexpr int KS = 7, TW = STEM_TW, TH = STEM_TH;
  constexpr int HALO_H = (TH - 1) * 2 + KS, HALO_W = (TW - 1) * 2 + KS;  // 13 x 69
  constexpr int X_ELEMS = HALO_H * HALO_W * 4 + 32;                       // +32: rows 28..31 read past the end
  constexpr int COW = 64;
  __shared__ __attribute__((aligned(16))) float lds[X_ELEMS + TH * TW * COW];
  float* lx = lds;
  float* ly = lds + X_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = lane >> 5, r = lane & 31;
  const int cob = blockIdx.x % a.co_blocks, ks = blockIdx.x / a.co_blocks;
  const int co0 = cob * COW;
  // waves: 2 co-tiles x 2 k-waves
  const int wco = wave & 1, wk = wave >> 1;
  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int Cin = a.C0;
  for (int tile = ks; tile < a.T; tile += a.ksplit) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * 2 - a.pad, ix0 = ox0 * 2 - a.pad;
    // batched staging: all loads of the tile in flight before the first LDS write
    constexpr int XT = HALO_H * HALO_W * 4 + 32, X_IT = (XT + 255) / 256;
    constexpr int YT = TH * TW * (COW / 4), Y_IT = (YT + 255) / 256;
    float rx[X_IT];
    f32x4 ry[Y_IT];
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int idx = tid + it * 256;
      const int c = idx & 3, pix = idx >> 2;
      const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
      const int iy = iy0 + hy, ix = ix0 + hx;
      float v = 0.f;
      if (pix < HALO_H * HALO_W && c < Cin && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win)
        v = a.src0[(((size_t)b * a.Hin + iy) * a.Win + ix) * Cin + c];
      rx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int idx = tid + it * 256;
      const int q = idx % (COW / 4), pix = idx / (COW / 4);
      const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
      const int c = co0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < YT && oy < a.Ho && ox < a.Wo && c < a.Cout) {
        const size_t o = (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + c;
        if (a.dy_bf16 != nullptr) {
          typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
          const bf16x4_t q = *reinterpret_cast<const bf16x4_t*>(a.dy_bf16 + o);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = (float)q[k];
        } else {
          v = *reinterpret_cast<const f32x4*>(a.dy + o);
        }
      }
      ry[it] = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < XT) lx[idx] = rx[it];
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < YT) *reinterpret_cast<f32x4*>(ly + (idx / (COW / 4)) * COW + 4 * (idx % (COW / 4))) = ry[it];
    }
    __syncthreads();
#pragma unroll
    for (int yy = 0; yy < TH / 2; ++yy) {
      const int y = yy * 2 + wk;
#pragma unroll
      for (int j = 0; j < TW / 2; ++j) {
        const float bv = ly[(y * TW + 2 * j + s) * COW + wco * 32 + r];
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
          const float av = lx[((y * 2 + kh) * HALO_W + (2 * j + s) * 2) * 4 + r];
          acc[kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[kh], 0, 0, 0);
        }
      }
    }
  }
  const int part = ks * 2 + wk;
  const int co = co0 + wco * 32 + r;
  if (co < a.Cout) {
#pragma unroll
    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i & 3) + 8 * (i >> 2) + 4 * s;
        const int kw = m >> 2, ci = m & 3;
        if (kw < KS && ci < Cin)
          a.ws[(((size_t)part * 49 + kh * 7 + kw) * Cin + ci) * a.Cout + co] = acc[kh][i];
      }
  }
}

// ---------------------------------------------------------------- host side
struct WgCfg {
  int tw, tpx, wci, wco, wk, ksplit, parts, T, tiles_x, tiles_y, ci_blocks, co_blocks;
  int rb, parts2;  // first reduction stage: blocks of rb slabs -> parts2 slabs (parts2 == parts: single stage)
  bool stem, narrow;
};

static int wg_validate(const dt_conv_desc* d) {
  DT_REQUIRE(d != nullptr, "wgrad: null descriptor");
  DT_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->C0 > 0 && d->C1 >= 0 && d->Cout > 0, "wgrad: bad sizes");
  DT_REQUIRE((d->ksize == 3 && (d->stride == 1 || d->stride == 2)) || (d->ksize == 1 && (d->stride == 1 || d->stride == 2)) ||
                 (d->ksize == 7 && d->stride == 2 && d->C1 == 0 && d->C0 <= 4 && d->mode0 == 0),
             "wgrad: ksize/stride (%d,%d) unsupported", d->ksize, d->stride);
  DT_REQUIRE(d->ksize == 7 || ((d->C0 & 3) == 0 && (d->C1 & 3) == 0), "wgrad: channels must be multiples of 4");
  DT_REQUIRE((d->Cout & 3) == 0, "wgrad: Cout must be a multiple of 4");
  DT_REQUIRE(d->mode0 >= 0 && d->mode0 <= 2, "wgrad: mode0");
  DT_REQUIRE(d->mode0 == 0 || ((d->Hin & 1) == 0 && (d->Win & 1) == 0), "wgrad: mode0 needs even Hin/Win");
  const int ho = (d->Hin + 2 * d->pad - d->ksize) / d->stride + 1;
  const int wo = (d->Win + 2 * d->pad - d->ksize) / d->stride + 1;
  DT_REQUIRE(ho == d->Ho && wo == d->Wo, "wgrad: Ho/Wo mismatch");
  return DT_OK;
}

static WgCfg wg_cfg(const dt_conv_desc* d) {
  WgCfg c;
  const int Cin = d->C0 + d->C1;
  c.stem = d->ksize == 7;
  c.narrow = dt_conv2d_wgrad_n16_supported(d) != 0;
  if (c.narrow) {
    c.tw = 32; c.tpx = 128; c.wci = c.wco = 1; c.wk = 4; c.ci_blocks = c.co_blocks = 1;
    c.tiles_x = c.tiles_y = 0;
    c.T = dt_wgrad_n16_cfg(d, &c.ksplit, &c.parts);
    c.rb = 1;
    c.parts2 = c.parts;
    if (c.parts > 16) {
      c.rb = dt_cdiv(c.parts, 16);
      c.parts2 = dt_cdiv(c.parts, c.rb);
    }
    return c;
  }
  if (c.stem) {
    c.tw = STEM_TW; c.tpx = STEM_TW * STEM_TH; c.wci = 1; c.wco = 2; c.wk = 2;
    c.ci_blocks = 1; c.co_blocks = dt_cdiv(d->Cout, 64);
  } else {
    c.tw = d->Wo > 16 ? 32 : 16;
    const bool ci_wide = Cin > 32, co_wide = d->Cout > 32;
    if (d->stride == 2 && d->ksize == 3) {
      // one arrangement for stride 2 (every such conv of the net has Cout >= 128); a narrower Cout is masked
      c.wci = 1; c.wco = 2; c.wk = 2; c.tpx = 64;
    } else if (ci_wide && co_wide) { c.wci = 2; c.wco = 2; c.wk = 1; c.tpx = 64; }
    else if (!ci_wide && co_wide) { c.wci = 1; c.wco = 2; c.wk = 2; c.tpx = 128; }
    else if (ci_wide && !co_wide) { c.wci = 2; c.wco = 1; c.wk = 2; c.tpx = 128; }
    else { c.wci = 1; c.wco = 1; c.wk = 4; c.tpx = 128; }
    c.ci_blocks = dt_cdiv(Cin, c.wci * 32);
    c.co_blocks = dt_cdiv(d->Cout, c.wco * 32);
  }
  const int th = c.tpx / c.tw;
  c.tiles_x = dt_cdiv(d->Wo, c.tw);
  c.tiles_y = dt_cdiv(d->Ho, th);
  c.T = d->B * c.tiles_x * c.tiles_y;
  int ks = 512 / (c.ci_blocks * c.co_blocks);
  if (ks < 1) ks = 1;
  if (ks > c.T) ks = c.T;
  c.ksplit = ks;
  c.parts = ks * c.wk;
  c.rb = 1;
  c.parts2 = c.parts;
  if (c.parts > 16) {
    c.rb = dt_cdiv(c.parts, 16);
    c.parts2 = dt_cdiv(c.parts, c.rb);
  }
  return c;
}

extern "C" size_t dt_conv2d_wgrad_workspace(const dt_conv_desc* d) {
  if (wg_validate(d) != DT_OK) return 0;
  WgCfg c = wg_cfg(d);
  const size_t E = (size_t)d->ksize * d->ksize * (d->C0 + d->C1) * d->Cout;
  return ((size_t)c.parts + (c.parts2 != c.parts ? c.parts2 : 0)) * E * sizeof(float);
}

template <int KS, int STRIDE, int TW, int TPX, int WCI, int WCO>
static int wg_launch(const WgradArgs& a, int grid, hipStream_t st) {
  if constexpr (KS == 3 && STRIDE == 1) {
    if (a.in_scale != nullptr) {
      hipLaunchKernelGGL((conv_wgrad_kernel<KS, STRIDE, TW, TPX, WCI, WCO, true>), dim3(grid), dim3(256), 0, st, a);
      DT_LAUNCH_CHECK();
      return DT_OK;
    }
  }
  hipLaunchKernelGGL((conv_wgrad_kernel<KS, STRIDE, TW, TPX, WCI, WCO>), dim3(grid), dim3(256), 0, st, a);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

template <int KS, int STRIDE, int TW>
static int wg_dispatch(const WgradArgs& a, const WgCfg& c, int grid, hipStream_t st) {
  if (c.wci == 2 && c.wco == 2) return wg_launch<KS, STRIDE, TW, 64, 2, 2>(a, grid, st);
  if (c.wci == 1 && c.wco == 2 && c.tpx == 128) return wg_launch<KS, STRIDE, TW, 128, 1, 2>(a, grid, st);
  if (c.wci == 1 && c.wco == 2 && c.tpx == 64) return wg_launch<KS, STRIDE, TW, 64, 1, 2>(a, grid, st);
  if (c.wci == 2 && c.wco == 1) return wg_launch<KS, STRIDE, TW, 128, 2, 1>(a, grid, st);
  return wg_launch<KS, STRIDE, TW, 128, 1, 1>(a, grid, st);
}

static int wgrad_impl(const dt_conv_desc* d, const float* src0, const float* src1, const float* dy, const void* dy_bf16,
                      float* dw, float* workspace, size_t workspace_bytes, const float* in_scale,
                      const float* in_shift, void* stream);

extern "C" int dt_conv2d_wgrad(const dt_conv_desc* d, const float* src0, const float* src1, const float* dy,
                               float* dw, float* workspace, size_t workspace_bytes, const float* in_scale,
                               const float* in_shift, void* stream) {
  return wgrad_impl(d, src0, src1, dy, nullptr, dw, workspace, workspace_bytes, in_scale, in_shift, stream);
}

// stem (7x7/2, fp32 image) weight gradient with dy in bf16 — the bf16 training path
extern "C" int dt_conv2d_wgrad_stem_dy_bf16(const dt_conv_desc* d, const float* src0, const void* dy_bf16, float* dw,
                                            float* workspace, size_t workspace_bytes, void* stream) {
  DT_REQUIRE(d && d->ksize == 7 && dy_bf16, "wgrad_stem_dy_bf16: 7x7 stem only");
  return wgrad_impl(d, src0, nullptr, reinterpret_cast<const float*>(dy_bf16), dy_bf16, dw, workspace, workspace_bytes,
                    nullptr, nullptr, stream);
}

static int wgrad_impl(const dt_conv_desc* d, const float* src0, const float* src1, const float* dy, const void* dy_bf16,
                      float* dw, float* workspace, size_t workspace_bytes, const float* in_scale,
                      const float* in_shift, void* stream) {
  int rc = wg_validate(d);
  if (rc != DT_OK) return rc;
  DT_REQUIRE(src0 && dy && dw && workspace, "wgrad: null pointer");
  DT_REQUIRE(d->C1 == 0 || src1, "wgrad: src1 missing");
  WgCfg c = wg_cfg(d);
  const int taps = d->ksize * d->ksize;
  const int64_t E = (int64_t)taps * (d->C0 + d->C1) * d->Cout;
  DT_REQUIRE(workspace_bytes >= dt_conv2d_wgrad_workspace(d), "wgrad: workspace too small (%zu < %zu)",
             workspace_bytes, dt_conv2d_wgrad_workspace(d));
  WgradArgs a;
  DT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "wgrad: in_scale/in_shift must come together");
  DT_REQUIRE(in_scale == nullptr || (d->ksize == 3 && d->stride == 1 && d->mode0 != 2),
             "wgrad: input transform needs a 3x3 stride-1 layer without zero-insertion");
  a.src0 = src0; a.src1 = src1; a.dy = dy; a.ws = workspace; a.in_scale = in_scale; a.in_shift = in_shift;
  a.dy_bf16 = (const __bf16*)dy_bf16;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.pad = d->pad;
  a.tiles_x = c.tiles_x; a.tiles_y = c.tiles_y; a.T = c.T;
  a.ci_blocks = c.ci_blocks; a.co_blocks = c.co_blocks; a.ksplit = c.ksplit;
  hipStream_t st = (hipStream_t)stream;
  const int grid = c.ci_blocks * c.co_blocks * c.ksplit;
  if (c.narrow) {
    rc = dt_wgrad_n16_launch(d, src0, dy, workspace, in_scale, in_shift, st);
  } else if (c.stem) {
    hipLaunchKernelGGL(conv_wgrad_stem_kernel, dim3(grid), dim3(256), 0, st, a);
    DT_LAUNCH_CHECK();
  } else if (d->ksize == 3 && d->stride == 1) {
    rc = c.tw == 32 ? wg_dispatch<3, 1, 32>(a, c, grid, st) : wg_dispatch<3, 1, 16>(a, c, grid, st);
  } else if (d->ksize == 3 && d->stride == 2) {
    // only the (1,2,k2,tpx64) arrangement is generated for stride 2
    rc = c.tw == 32 ? wg_launch<3, 2, 32, 64, 1, 2>(a, grid, st) : wg_launch<3, 2, 16, 64, 1, 2>(a, grid, st);
  } else if (d->stride == 2) {
    rc = c.tw == 32 ? wg_dispatch<1, 2, 32>(a, c, grid, st) : wg_dispatch<1, 2, 16>(a, c, grid, st);
  } else {   // 1x1 stride 1: the identity_conv of the ResUnet decoder (resunet/decoder.py:36-38)
    rc = c.tw == 32 ? wg_dispatch<1, 1, 32>(a, c, grid, st) : wg_dispatch<1, 1, 16>(a, c, grid, st);
  }
  if (rc != DT_OK) return rc;
  const float* slabs = workspace;
  int nslabs = c.parts;
  if (c.parts2 != c.parts) {
    float* stage = workspace + (size_t)c.parts * E;
    rc = dt_reduce_rows_launch(workspace, stage, 1, c.parts, (int)E, c.rb, st);
    if (rc != DT_OK) return rc;
    slabs = stage;
    nslabs = c.parts2;
  }
  int64_t g = (E + 255) / 256;
  if (g > 256 * 8) g = 256 * 8;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, slabs, dw, nslabs, E);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
