// Direct (im2col-free) NHWC convolution on the fp32 matrix cores of gfx950.
//
// Replaces ATen conv2d on the reference hot path (smp.Unet(resnet34) forward reached from
// deadtrees/network/segmodel.py:214,235,280 and deployment/inference.py:60) and, through the
// input modes, also F.interpolate(nearest x2) + torch.cat of the smp decoder block (in-tree twin:
// deadtrees/network/extra/resunet/decoder.py:41-43) and the data-gradient convolutions of autograd.
//
// Design (MI355X-first):
//   * one workgroup (4 waves) owns a 256-pixel output tile (TH x TW) x TN output channels;
//   * the input halo tile for CK input channels is staged ONCE in LDS (channel-planar, so the
//     32 pixels of an MFMA row-fragment are 32 consecutive dwords: conflict-free ds_read_b32) and is
//     re-used by all KS*KS taps — no im2col buffer ever exists in HBM;
//   * virtual nearest-upsample / zero-insertion / channel concat are index arithmetic in the LDS fill;
//   * the contraction over input channels runs on v_mfma_f32_32x32x2_f32: exact fp32 fma chains at
//     the fp32 peak (157 TF) with one VGPR per operand (MI355X_MICROARCH.md §Matrix cores);
//   * epilogue: coalesced 128-B row stores + per-channel sum / sum-of-squares partials for the
//     following BatchNorm (wave64 shuffle -> LDS -> one row per workgroup; reduced later in fixed
//     order, so results are run-to-run deterministic — no float atomics).
#include "common.h"

struct ConvArgs {
  const float* src0;
  const float* src1;
  const float* w;
  const float* in_scale;  // optional per-channel affine + ReLU applied to source 0 while staging (fused BatchNorm
  const float* in_shift;  // apply of the producer layer: z = relu(y*scale+shift) is never materialised)
  float* out0;
  __bf16* out0_bf16;      // when set (bf16 training path, stem): out0 is written as bf16 instead (no split/accumulate)
  float* out1;
  float* stats;
  const float* aff_scale;  // inference epilogue (dt_conv2d_affine): out = [relu](conv * scale + shift), eval-mode BatchNorm
  const float* aff_shift;  // on the accumulators (bn_act's mul, add, NaN-keeping ReLU); no statistics
  int aff_relu;
  // fused BatchNorm-backward reduction (dt_conv2d_bn_bwd): when bnb.y is set, `stats` receives sum g, sum g*xhat
  dt_bn_bwd_fuse bnb;
  int B, Hin, Win, C0, C1, mode0;
  int Ho, Wo, Cout, cout_split, pad, accumulate;
  int tiles_x, tiles_y, n_tiles, P;
  int sp_tiles;
  int pack;   // 1: a tile = four images of <= 8 x 8 output pixels (layer4.0.conv1 of a 256-pixel tile): rows 8 k .. 8 k + 7 of
              // the 8 x 32 tile belong to image 4 sp + k — without it three quarters of every M tile were padding
};

__host__ __device__ constexpr int plane_pad(int n, int mod8) {
  // smallest p >= n with p % 8 == mod8
  int p = n;
  while ((p & 7) != mod8) ++p;
  return p;
}

template <int KS, int STRIDE, int TW, int CK>
struct ConvGeom {
  static constexpr int TH = 256 / TW;
  static constexpr int LS = (KS == 1) ? 1 : STRIDE;  // pixel step inside the LDS tile
  static constexpr int GS = (KS == 1) ? STRIDE : 1;  // pixel step in global memory per LDS pixel
  // 8-wide stride-2 3x3 tiles (32 rows) can hold four images of at most 8 x 8 output pixels (ConvArgs::pack): each
  // image's 17 halo rows (input rows -1 .. 15) are kept apart, 4 x 17 = 68 rows instead of 65
  static constexpr bool PACKABLE = KS == 3 && STRIDE == 2 && TW == 8;
  static constexpr int HALO_H = PACKABLE ? 68 : (TH - 1) * LS + KS;
  static constexpr int HALO_W = (TW - 1) * LS + KS;
  static constexpr int PMOD = (CK >= 32) ? 1 : (CK == 16 ? 2 : (CK == 8 ? 4 : 0));
  static constexpr int PLANE = plane_pad(HALO_H * HALO_W, PMOD);
  static constexpr int TAPS = KS * KS;
};

// Stage CK channels [c0, c0+CK) of the logical input halo tile into LDS, channel-planar.
template <int KS, int STRIDE, int TW, int CK>
__device__ __forceinline__ void fill_input_planar(float* __restrict__ lds_in, const ConvArgs& a, int b,
                                                  int iy0, int ix0, int c0) {
  using G = ConvGeom<KS, STRIDE, TW, CK>;
  const float* src;
  int C, cc, mode;
  if (c0 < a.C0) {
    src = a.src0; C = a.C0; cc = c0; mode = a.mode0;
  } else {
    src = a.src1; C = a.C1; cc = c0 - a.C0; mode = 0;
  }
  const int Hs = mode ? (a.Hin >> 1) : a.Hin;
  const int Ws = mode ? (a.Win >> 1) : a.Win;
  const int tid = threadIdx.x;
  if ((C & 3) == 0 && (CK & 3) == 0) {
    constexpr int Q = CK / 4;
    constexpr int TOTAL = G::HALO_H * G::HALO_W * Q;
    for (int idx = tid; idx < TOTAL; idx += 256) {
      const int q = idx % Q;
      const int pix = idx / Q;
      const int hy = pix / G::HALO_W, hx = pix - hy * G::HALO_W;
      const int iy = iy0 + hy * G::GS, ix = ix0 + hx * G::GS;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win && (cc + 4 * q) < C;
      if (mode == 2) ok = ok && (((iy | ix) & 1) == 0);
      if (ok) {
        const int sy = mode ? (iy >> 1) : iy, sx = mode ? (ix >> 1) : ix;
        v = *reinterpret_cast<const f32x4*>(src + (((size_t)b * Hs + sy) * Ws + sx) * C + cc + 4 * q);
      }
      float* d = lds_in + (4 * q) * G::PLANE + pix;
      d[0] = v[0];
      d[G::PLANE] = v[1];
      d[2 * G::PLANE] = v[2];
      d[3 * G::PLANE] = v[3];
    }
  } else {
    constexpr int TOTAL = G::HALO_H * G::HALO_W * CK;
    for (int idx = tid; idx < TOTAL; idx += 256) {
      const int c = idx % CK;
      const int pix = idx / CK;
      const int hy = pix / G::HALO_W, hx = pix - hy * G::HALO_W;
      const int iy = iy0 + hy * G::GS, ix = ix0 + hx * G::GS;
      float v = 0.f;
      bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win && (cc + c) < C;
      if (mode == 2) ok = ok && (((iy | ix) & 1) == 0);
      if (ok) {
        const int sy = mode ? (iy >> 1) : iy, sx = mode ? (ix >> 1) : ix;
        v = src[(((size_t)b * Hs + sy) * Ws + sx) * C + cc + c];
      }
      lds_in[c * G::PLANE + pix] = v;
    }
  }
}

// Stage weights [tap][c0..c0+CK)[n0..n0+TN) into LDS as [tap][CK][TN].
template <int TAPS, int CK, int TN>
__device__ __forceinline__ void fill_weights(float* __restrict__ lds_w, const float* __restrict__ w, int Cin,
                                             int Cout, int c0, int n0) {
  const int tid = threadIdx.x;
  if ((Cout & 3) == 0) {
    constexpr int Q = TN / 4;
    constexpr int TOTAL = TAPS * CK * Q;
    for (int idx = tid; idx < TOTAL; idx += 256) {
      const int q = idx % Q;
      const int row = idx / Q;  // tap*CK + k
      const int tap = row / CK, k = row - tap * CK;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c0 + k < Cin && n0 + 4 * q < Cout)
        v = *reinterpret_cast<const f32x4*>(w + ((size_t)tap * Cin + c0 + k) * Cout + n0 + 4 * q);
      *reinterpret_cast<f32x4*>(lds_w + row * TN + 4 * q) = v;
    }
  } else {
    constexpr int TOTAL = TAPS * CK * TN;
    for (int idx = tid; idx < TOTAL; idx += 256) {
      const int n = idx % TN;
      const int row = idx / TN;
      const int tap = row / CK, k = row - tap * CK;
      float v = 0.f;
      if (c0 + k < Cin && n0 + n < Cout) v = w[((size_t)tap * Cin + c0 + k) * Cout + n0 + n];
      lds_w[row * TN + n] = v;
    }
  }
}

// all KS*KS taps x CK input channels of one staged chunk on the matrix cores
template <int KS, int TN, int CK, int PLANE, int HALO_W>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ lds_in, const float* __restrict__ lds_w,
                                          const int (&abase)[2], int bbase, f32x16 (&acc)[2][TN / 32]) {
  constexpr int NT = TN / 32;
#pragma unroll
  for (int tap = 0; tap < KS * KS; ++tap) {
    const int kh = tap / KS, kw = tap % KS;
#pragma unroll
    for (int kk = 0; kk < CK / 2; ++kk) {
      float av[2], bv[NT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) av[mt] = lds_in[abase[mt] + 2 * kk * PLANE + kh * HALO_W + kw];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = lds_w[bbase + (tap * CK + 2 * kk) * TN + 32 * j];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[j], acc[mt][j], 0, 0, 0);
    }
  }
}

// explicitly software-pipelined variant: the fragments of step i+1 are loaded before the MFMAs of step i
template <int KS, int TN, int CK, int PLANE, int HALO_W>
__device__ __forceinline__ void mma_chunk_v1(const float* __restrict__ lds_in, const float* __restrict__ lds_w,
                                             const int (&abase)[2], int bbase, f32x16 (&acc)[2][TN / 32]) {
  constexpr int NT = TN / 32;
  constexpr int STEPS = KS * KS * (CK / 2);
  float av[2][2], bv[2][NT];
  auto load = [&](int step, int buf) {
    const int tap = step / (CK / 2), kk = step % (CK / 2);
    const int kh = tap / KS, kw = tap % KS;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) av[buf][mt] = lds_in[abase[mt] + 2 * kk * PLANE + kh * HALO_W + kw];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[buf][j] = lds_w[bbase + (tap * CK + 2 * kk) * TN + 32 * j];
  };
  load(0, 0);
#pragma unroll
  for (int step = 0; step < STEPS; ++step) {
    const int cur = step & 1;
    if (step + 1 < STEPS) load(step + 1, cur ^ 1);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mt], bv[cur][j], acc[mt][j], 0, 0, 0);
  }
}

// zero-insertion variant: per row-tile tap mask (wave-uniform), B fragments re-read per tile
template <int KS, int TN, int CK, int PLANE, int HALO_W>
__device__ __forceinline__ void mma_chunk_zi(const float* __restrict__ lds_in, const float* __restrict__ lds_w,
                                             const int (&abase)[2], int bbase, f32x16 (&acc)[2][TN / 32], int wave,
                                             int pad) {
  constexpr int NT = TN / 32;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int py = mt, px = (wave & 1) ? (1 - mt) : mt;
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
      const int kh = tap / KS, kw = tap % KS;
      if ((((py + kh - pad) | (px + kw - pad)) & 1) != 0) continue;   // structurally zero input: skip the tap
#pragma unroll
      for (int kk = 0; kk < CK / 2; ++kk) {
        const float av = lds_in[abase[mt] + 2 * kk * PLANE + kh * HALO_W + kw];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float bv = lds_w[bbase + (tap * CK + 2 * kk) * TN + 32 * j];
          acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt][j], 0, 0, 0);
        }
      }
    }
  }
}

// narrow-input stride-1 layers (Cin <= 32: one or two chunks, no pipelining depth) run the CK=8 variant at
// 3 waves/SIMD when TN = 32 (with 64 output channels per tile 168 VGPRs spill); everything else 2 waves/SIMD
// (measured per layer with scripts/bench_conv.py)
// Zero-insertion (transposed-conv) tiles, ZI = true: 3/4 of the zero-inserted input is structurally zero, so an
// output pixel of parity class (py,px) only sees the taps with (py+kh-pad, px+kw-pad) both even: 1/2/2/4 of the
// 9 taps (1x1: only the even/even class).  Each 32-pixel MFMA row tile is therefore built from pixels of ONE
// parity class (2 rows of the same parity x 16 same-parity columns) and skips the dead taps with wave-uniform
// branches; every wave gets one light and one heavy class (5 or 4 taps instead of 18).
template <bool ZI, int TW>
__device__ __forceinline__ void tile_pixel(int wave, int mt, int m, int& row, int& col, int& py, int& px) {
  if constexpr (ZI) {
    const int pair = wave >> 1;
    py = mt;                                  // wave's tile 0: even rows, tile 1: odd rows
    px = (wave & 1) ? (1 - mt) : mt;          // even waves: (E,E)+(O,O); odd waves: (E,O)+(O,E)
    row = py + 2 * (2 * pair + (m >> 4));
    col = px + 2 * (m & 15);
  } else {
    const int p = (wave * 2 + mt) * 32 + m;
    row = p / TW;
    col = p % TW;
    py = px = 0;
  }
}

#define DT_TF_MAXC 512   // channels of source 0 that may carry a fused input transform

template <int KS, int STRIDE, int TW, int TN, int CK, bool ZI = false, bool TF = false>
__global__ __launch_bounds__(256, (KS == 3 && STRIDE == 1 && CK == 8 && TN == 32) ? 3 : 2) void conv_fwd_kernel(const ConvArgs a) {
  static_assert(!ZI || (TW == 32 && STRIDE == 1), "zero-insertion tiles are 8 x 32");
  static_assert(!TF || (KS != 7 && !ZI), "input transform: regular tiles only");
  // TF: per-channel scale/shift of source 0 live in LDS (no registers held across the MFMA loop)
  __shared__ __attribute__((aligned(16))) float lds_tf[TF ? 2 * DT_TF_MAXC : 4];
  if constexpr (TF) {
    for (int i = threadIdx.x; i < a.C0; i += 256) {
      lds_tf[i] = a.in_scale[i];
      lds_tf[DT_TF_MAXC + i] = a.in_shift[i];
    }
  }
  using G = ConvGeom<KS, STRIDE, TW, CK>;
  constexpr int NT = TN / 32;
  constexpr int IN_ELEMS = CK * G::PLANE;
  constexpr int W_ELEMS = G::TAPS * CK * TN;
  __shared__ __attribute__((aligned(16))) float lds[IN_ELEMS + W_ELEMS];
  float* lds_in = lds;
  float* lds_w = lds + IN_ELEMS;
  static_assert((IN_ELEMS & 3) == 0, "weight region must stay 16-byte aligned");

  const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);   // n-tiles of one spatial tile share an XCD's L2
  const int nt = wg % a.n_tiles;
  const int sp = wg / a.n_tiles;
  const int tx = sp % a.tiles_x;
  const int ty = (sp / a.tiles_x) % a.tiles_y;
  const int b = sp / (a.tiles_x * a.tiles_y);
  const int oy0 = ty * G::TH, ox0 = tx * TW, n0 = nt * TN;
  const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane >> 5, r = lane & 31;

  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int py, px, cy, cx;
    tile_pixel<ZI, TW>(wave, mt, r, py, px, cy, cx);
    abase[mt] = s * G::PLANE + py * G::LS * G::HALO_W + px * G::LS;
    if (G::PACKABLE && a.pack) abase[mt] += (py >> 3) * G::HALO_W;   // image k's halo starts at row 17 k, not 16 k
  }
  const int bbase = s * TN + r;

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][j][i] = 0.f;

  const int Cin = a.C0 + a.C1;
  if constexpr (KS == 7) {
    // stem (Cin = 3 or 4, a single chunk): scalar staging, but with every load of the tile in flight at once
    constexpr int IN_TOTAL = G::HALO_H * G::HALO_W * CK;
    constexpr int IN_IT = (IN_TOTAL + 255) / 256;
    constexpr int W_TOTAL = G::TAPS * CK * TN;
    constexpr int W_IT = (W_TOTAL + 255) / 256;
    const int tid = threadIdx.x;
    float rin[IN_IT], rw[W_IT];
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int idx = tid + it * 256;
      const int c = idx % CK, pix = idx / CK;
      const int hy = pix / G::HALO_W, hx = pix - hy * G::HALO_W;
      const int iy = iy0 + hy * G::GS, ix = ix0 + hx * G::GS;
      float v = 0.f;
      if (idx < IN_TOTAL && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win && c < Cin)
        v = a.src0[(((size_t)b * a.Hin + iy) * a.Win + ix) * Cin + c];
      rin[it] = v;
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int idx = tid + it * 256;
      const int n = idx % TN, row = idx / TN;
      const int tap = row / CK, k = row - tap * CK;
      float v = 0.f;
      if (idx < W_TOTAL && k < Cin && n0 + n < a.Cout) v = a.w[((size_t)tap * Cin + k) * a.Cout + n0 + n];
      rw[it] = v;
    }
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < IN_TOTAL) lds_in[(idx % CK) * G::PLANE + idx / CK] = rin[it];
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < W_TOTAL) lds_w[idx] = rw[it];
    }
    __syncthreads();
    mma_chunk<KS, TN, CK, G::PLANE, G::HALO_W>(lds_in, lds_w, abase, bbase, acc);
  } else {
    // Software-pipelined staging: the global loads of chunk c+1 are issued (into registers) before the
    // MFMAs of chunk c and written to LDS after them, so HBM/L2 latency hides under the matrix work.
    constexpr int QI = CK / 4;                       // float4 per pixel
    constexpr int IN_TOTAL = G::HALO_H * G::HALO_W * QI;
    constexpr int IN_IT = (IN_TOTAL + 255) / 256;
    constexpr int QW = TN / 4;
    constexpr int W_TOTAL = G::TAPS * CK * QW;
    constexpr int W_IT = (W_TOTAL + 255) / 256;
    static_assert(256 % QI == 0 && 256 % QW == 0, "lane->quad mapping must be iteration invariant");
    const int tid = threadIdx.x;
    const int qi = tid % QI, pix0 = tid / QI;
    const int qw = tid % QW, row0 = tid / QW;
    int pidx0[IN_IT], pidx1[IN_IT];   // pixel index in source 0 / source 1, -1 = zero (padding / inserted zero)
    const int Hs0 = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws0 = a.mode0 ? (a.Win >> 1) : a.Win;
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int pix = pix0 + it * (256 / QI);
      const int hy = pix / G::HALO_W, hx = pix - hy * G::HALO_W;
      int iy = iy0 + hy * G::GS, bb = b;
      const int ix = ix0 + hx * G::GS;
      if (G::PACKABLE && a.pack) {          // halo rows 17 k .. 17 k + 16 = input rows -1 .. 15 of image 4 sp + k
        bb = 4 * sp + hy / 17;
        iy = hy % 17 - a.pad;
      }
      const bool inb = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win && pix < G::HALO_H * G::HALO_W &&
                       bb < a.B;
      bool ok0 = inb;
      if (a.mode0 == 2) ok0 = ok0 && (((iy | ix) & 1) == 0);
      const int sy = a.mode0 ? (iy >> 1) : iy, sx = a.mode0 ? (ix >> 1) : ix;
      pidx0[it] = ok0 ? (bb * Hs0 + sy) * Ws0 + sx : -1;
      pidx1[it] = inb ? (bb * a.Hin + iy) * a.Win + ix : -1;
    }
    int woff[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int row = row0 + it * (256 / QW);
      const int tap = row / CK, k = row - tap * CK;
      woff[it] = (row < G::TAPS * CK && n0 + 4 * qw < a.Cout) ? (tap * Cin + k) * a.Cout + n0 + 4 * qw : -1;
    }
    f32x4 rin[IN_IT], rw[W_IT];
    auto issue_loads = [&](int c0) {
      const bool use0 = c0 < a.C0;
      const float* src = use0 ? a.src0 : a.src1;
      const int C = use0 ? a.C0 : a.C1;
      const int cc = (use0 ? c0 : c0 - a.C0) + 4 * qi;

#pragma unroll
      for (int it = 0; it < IN_IT; ++it) {
        const int p = use0 ? pidx0[it] : pidx1[it];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p >= 0 && cc < C) v = *reinterpret_cast<const f32x4*>(src + (size_t)p * C + cc);
        rin[it] = v;
      }
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int k = (row0 + it * (256 / QW)) % CK;
        if (woff[it] >= 0 && c0 + k < Cin) v = *reinterpret_cast<const f32x4*>(a.w + (size_t)c0 * a.Cout + woff[it]);
        rw[it] = v;
      }
    };
    auto write_lds = [&](int c0) {
      bool tf_on = false;
      f32x4 tf_sc = {1.f, 1.f, 1.f, 1.f}, tf_sh = {0.f, 0.f, 0.f, 0.f};
      if constexpr (TF) {
        const int cc = c0 + 4 * qi;
        tf_on = cc < a.C0;
        if (tf_on) {
          tf_sc = *reinterpret_cast<const f32x4*>(lds_tf + cc);
          tf_sh = *reinterpret_cast<const f32x4*>(lds_tf + DT_TF_MAXC + cc);
        }
      }
#pragma unroll
      for (int it = 0; it < IN_IT; ++it) {
        const int pix = pix0 + it * (256 / QI);
        if (IN_TOTAL % 256 == 0 || pix < G::HALO_H * G::HALO_W) {
          f32x4 v = rin[it];
          if (TF && tf_on && pidx0[it] >= 0) {   // zero padding stays zero: only real pixels get the affine + ReLU
            v = v * tf_sc + tf_sh;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] < 0.f ? 0.f : v[k];
          }
          float* d = lds_in + (4 * qi) * G::PLANE + pix;
          d[0] = v[0];
          d[G::PLANE] = v[1];
          d[2 * G::PLANE] = v[2];
          d[3 * G::PLANE] = v[3];
        }
      }
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        const int row = row0 + it * (256 / QW);
        if (W_TOTAL % 256 == 0 || row < G::TAPS * CK) *reinterpret_cast<f32x4*>(lds_w + row * TN + 4 * qw) = rw[it];
      }
    };
    issue_loads(0);
    for (int c0 = 0; c0 < Cin; c0 += CK) {
      __syncthreads();   // every wave is done reading the previous chunk
      write_lds(c0);
      __syncthreads();
      if (c0 + CK < Cin) issue_loads(c0 + CK);
      if constexpr (ZI)
        mma_chunk_zi<KS, TN, CK, G::PLANE, G::HALO_W>(lds_in, lds_w, abase, bbase, acc, wave, a.pad);
      else
        mma_chunk_v1<KS, TN, CK, G::PLANE, G::HALO_W>(lds_in, lds_w, abase, bbase, acc);
    }
  }

  // ---------------- epilogue: store + BatchNorm partial statistics
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + 32 * j + r;
    float* outp = a.out0;
    int ld = a.Cout, nn = n;
    if (a.cout_split > 0) {
      if (n0 >= a.cout_split) {
        outp = a.out1; ld = a.Cout - a.cout_split; nn = n - a.cout_split;
      } else {
        ld = a.cout_split;
      }
    }
    const bool nok = n < a.Cout;
    const bool bnb = a.bnb.y != nullptr;   // uniform
    float b_mu = 0.f, b_is = 0.f, b_sc = 0.f, b_sh = 0.f;
    if (bnb && nok) {
      b_mu = a.bnb.mean[n];
      b_is = a.bnb.invstd[n];
      if (a.bnb.act == nullptr) {
        b_sc = a.bnb.act_scale[n];
        b_sh = a.bnb.act_shift[n];
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      size_t off[16];
      bool ok[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int mrow = (i & 3) + 8 * (i >> 2) + 4 * s;
        int ty_, tx_, cy, cx;
        tile_pixel<ZI, TW>(wave, mt, mrow, ty_, tx_, cy, cx);
        int oy = oy0 + ty_, bb = b;
        const int ox = ox0 + tx_;
        if (G::PACKABLE && a.pack) {
          bb = 4 * sp + (ty_ >> 3);
          oy = ty_ & 7;
        }
        ok[i] = nok && oy < a.Ho && ox < a.Wo && bb < a.B;
        off[i] = (((size_t)bb * a.Ho + oy) * a.Wo + ox) * ld + nn;
      }
      if (a.accumulate && outp == a.out0) {
        float prev[16];   // all 16 loads in flight before the first add (gradient accumulation joins)
#pragma unroll
        for (int i = 0; i < 16; ++i) prev[i] = ok[i] ? outp[off[i]] : 0.f;
        if (bnb) {
          // join + BatchNorm-backward sums of the block-output layer: mask from its stored activation
          float yv[16], zv[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            yv[i] = ok[i] ? a.bnb.y[off[i]] : 0.f;
            zv[i] = ok[i] ? a.bnb.act[off[i]] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (ok[i]) {
              const float v = acc[mt][j][i] + prev[i];
              const float g = zv[i] > 0.f ? v : 0.f;
              s1[j] += g;
              s2[j] += g * ((yv[i] - b_mu) * b_is);
              outp[off[i]] = v;
            }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (ok[i]) outp[off[i]] = acc[mt][j][i] + prev[i];
        }
      } else if (bnb) {
        // BatchNorm-backward partial sums of the layer this gradient belongs to (its raw output y at the same
        // positions; 128-byte rows like the stores): g = v * [relu mask], sum g and sum g * xhat
        float yv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) yv[i] = ok[i] ? a.bnb.y[off[i]] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (ok[i]) {
            const float v = acc[mt][j][i];
            const float g = (yv[i] * b_sc + b_sh) > 0.f ? v : 0.f;   // virtual activation (act == nullptr: host check)
            s1[j] += g;
            s2[j] += g * ((yv[i] - b_mu) * b_is);
            outp[off[i]] = v;
          }
      } else if (a.aff_scale != nullptr) {
        const float e_sc = nok ? a.aff_scale[n] : 0.f, e_sh = nok ? a.aff_shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (ok[i]) {
            float v = acc[mt][j][i] * e_sc + e_sh;
            if (a.aff_relu) v = v < 0.f ? 0.f : v;
            outp[off[i]] = v;
          }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (ok[i]) {
            const float v = acc[mt][j][i];
            s1[j] += v;
            s2[j] += v * v;
            if (a.out0_bf16 != nullptr)
              a.out0_bf16[off[i]] = (__bf16)v;
            else
              outp[off[i]] = v;
          }
      }
    }
  }
  if (a.stats != nullptr) {
    __syncthreads();  // all waves done with lds_in / lds_w
    float* red = lds;  // [2][4 waves][TN]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
      const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (s == 0) {
        red[wave * TN + 32 * j + r] = t1;
        red[4 * TN + wave * TN + 32 * j + r] = t2;
      }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (G::PACKABLE && a.pack) {
      // wave w holds image 4 sp + w (in the pixel order of a tile of its own): one row per IMAGE, the same partial sums
      // as without packing
      for (int idx = t; idx < 8 * TN; idx += 256) {
        const int which = idx / (4 * TN), wv = (idx / TN) & 3, c = idx % TN, img = 4 * sp + wv;
        if (img < a.B && n0 + c < a.Cout) a.stats[((size_t)which * a.P + img) * a.Cout + n0 + c] = red[which * 4 * TN + wv * TN + c];
      }
    } else if (t < 2 * TN) {
      const int which = t / TN, c = t % TN;
      if (n0 + c < a.Cout) {
        const float* rr = red + which * 4 * TN + c;
        const float v = (rr[0] + rr[TN]) + (rr[2 * TN] + rr[3 * TN]);
        a.stats[((size_t)which * a.P + sp) * a.Cout + n0 + c] = v;
      }
    }
  }
}

// ---------------------------------------------------------------- host dispatch
struct ConvCfg {
  int tw, tn;
};

// stride-2 3x3 layers with maps of at most 8 x 8 output pixels: four images share an 8 x 32 tile (ConvArgs::pack)
static int conv_packs(const dt_conv_desc* d) {
  return d->ksize == 3 && d->stride == 2 && d->pad == 1 && d->mode0 == 0 && d->C1 == 0 && d->Ho <= 8 && d->Wo <= 8 &&
         d->Hin <= 16 && d->B >= 2;
}

static ConvCfg pick_cfg(const dt_conv_desc* d) {
  ConvCfg c;
  c.tw = d->Wo > 16 ? 32 : (d->Wo > 8 ? 16 : 8);
  int tn = d->Cout >= 64 ? 64 : 32;
  if (d->cout_split > 0 && (d->cout_split % 64) != 0) tn = 32;
  // keep >= 2 workgroups per CU in flight when the grid is small (deep layers: few spatial tiles)
  if (tn == 64) {
    const int th = 256 / c.tw;
    const long sp = conv_packs(d) ? dt_cdiv(d->B, 4) : (long)d->B * dt_cdiv(d->Ho, th) * dt_cdiv(d->Wo, c.tw);
    const long wgs = sp * dt_cdiv(d->Cout, 64);
    if (wgs < 512) tn = 32;
  }
  c.tn = tn;
  return c;
}

static int validate(const dt_conv_desc* d) {
  DT_REQUIRE(d != nullptr, "conv: null descriptor");
  DT_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->C0 > 0 && d->C1 >= 0 && d->Cout > 0, "conv: bad sizes");
  DT_REQUIRE(d->ksize == 1 || d->ksize == 3 || d->ksize == 7, "conv: ksize %d unsupported", d->ksize);
  DT_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported", d->stride);
  DT_REQUIRE(d->mode0 >= 0 && d->mode0 <= 2, "conv: mode0 %d", d->mode0);
  DT_REQUIRE(d->mode0 == 0 || ((d->Hin & 1) == 0 && (d->Win & 1) == 0), "conv: mode0 needs even Hin/Win");
  const int ho = (d->Hin + 2 * d->pad - d->ksize) / d->stride + 1;
  const int wo = (d->Win + 2 * d->pad - d->ksize) / d->stride + 1;
  DT_REQUIRE(ho == d->Ho && wo == d->Wo, "conv: Ho/Wo (%d,%d) != expected (%d,%d)", d->Ho, d->Wo, ho, wo);
  DT_REQUIRE(d->C1 == 0 || (d->C0 % 16) == 0, "conv: concat needs C0 %% 16 == 0");
  DT_REQUIRE(d->cout_split == 0 || ((d->cout_split % 32) == 0 && d->cout_split < d->Cout),
             "conv: cout_split must be a multiple of 32 below Cout");
  DT_REQUIRE(d->ksize != 7 || (d->stride == 2 && d->C1 == 0 && d->C0 <= 4), "conv: 7x7 only as the stem");
  return DT_OK;
}

extern "C" int dt_conv2d_stat_rows(const dt_conv_desc* d) {
  if (validate(d) != DT_OK) return DT_EINVAL;
  if (dt_conv2d_narrow_supported(d)) return dt_conv2d_narrow_rows(d);   // one row per persistent workgroup (upper bound)
  ConvCfg c = pick_cfg(d);
  return d->B * dt_cdiv(d->Ho, 256 / c.tw) * dt_cdiv(d->Wo, c.tw);
}

template <int KS, int STRIDE, int TW, int TN, int CK, bool ZI = false>
static int launch(const ConvArgs& a, hipStream_t st) {
  const long grid = (long)a.sp_tiles * a.n_tiles;
  if constexpr (KS == 3 && STRIDE == 1 && !ZI) {
    if (a.in_scale != nullptr) {
      hipLaunchKernelGGL((conv_fwd_kernel<KS, STRIDE, TW, TN, CK, false, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
      DT_LAUNCH_CHECK();
      return DT_OK;
    }
  } else {
    if (a.in_scale != nullptr) {
      dt_set_error("conv: fused input transform is only built for 3x3 stride-1 layers");
      return DT_ENOSYS;
    }
  }
  hipLaunchKernelGGL((conv_fwd_kernel<KS, STRIDE, TW, TN, CK, ZI>), dim3((unsigned)grid), dim3(256), 0, st, a);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

template <int KS, int STRIDE, int CK>
static int launch_tw_tn(const ConvArgs& a, const ConvCfg& c, hipStream_t st) {
  if (c.tn == 64) {
    if (c.tw == 32) return launch<KS, STRIDE, 32, 64, CK>(a, st);
    if (c.tw == 16) return launch<KS, STRIDE, 16, 64, CK>(a, st);
    return launch<KS, STRIDE, 8, 64, CK>(a, st);
  }
  if (c.tw == 32) return launch<KS, STRIDE, 32, 32, CK>(a, st);
  if (c.tw == 16) return launch<KS, STRIDE, 16, 32, CK>(a, st);
  return launch<KS, STRIDE, 8, 32, CK>(a, st);
}

static int conv2d_impl(const dt_conv_desc* d, const float* src0, const float* src1, const float* w, float* out0,
                       float* out1, float* stats, const float* in_scale, const float* in_shift, void* out_bf16,
                       void* stream, const dt_bn_bwd_fuse* fuse = nullptr, const float* aff_scale = nullptr,
                       const float* aff_shift = nullptr, int aff_relu = 0);

// inference: out = [relu](conv(x) * scale + shift) in one launch — eval-mode BatchNorm (+ ReLU) of the layers that are
// neither Winograd layers (dt_conv2d_winograd_affine) nor narrow ones (dt_conv2d_narrow_affine): the stem, the stride-2
// 3x3 and the 1x1 down-sample convolutions, dec3.conv1 (reference: model.eval() forward of
// segmentation_models_pytorch Unet via deadtrees/network/segmodel.py:151-153 -> Conv2d + BatchNorm2d [+ ReLU])
extern "C" int dt_conv2d_affine(const dt_conv_desc* d, const float* src0, const float* src1, const float* w, float* out,
                                const float* scale, const float* shift, int relu, void* stream) {
  DT_REQUIRE(d && scale && shift, "conv_affine: null pointer");
  DT_REQUIRE(d->cout_split == 0 && d->accumulate == 0, "conv_affine: no split / join");
  return conv2d_impl(d, src0, src1, w, out, nullptr, nullptr, nullptr, nullptr, nullptr, stream, nullptr, scale, shift, relu);
}

extern "C" int dt_conv2d(const dt_conv_desc* d, const float* src0, const float* src1, const float* w,
                         float* out0, float* out1, float* stats, const float* in_scale, const float* in_shift,
                         void* stream) {
  return conv2d_impl(d, src0, src1, w, out0, out1, stats, in_scale, in_shift, nullptr, stream);
}

extern "C" int dt_conv2d_bn_bwd(const dt_conv_desc* d, const float* src0, const float* w, float* out0, float* red,
                                const dt_bn_bwd_fuse* fuse, void* stream) {
  DT_REQUIRE(d && fuse && red && fuse->y && fuse->mean && fuse->invstd, "conv_bn_bwd: null pointer");
  DT_REQUIRE(fuse->act != nullptr || (fuse->act_scale && fuse->act_shift),
             "conv_bn_bwd: give the stored activation or the scale/shift of a virtual one");
  DT_REQUIRE(d->ksize == 3 && d->stride == 1 && d->mode0 == 0 && d->C1 == 0 && d->cout_split == 0,
             "conv_bn_bwd: plain 3x3 stride-1 data gradients only");
  DT_REQUIRE((d->accumulate != 0) == (fuse->act != nullptr),
             "conv_bn_bwd: gradient joins (accumulate) go with a stored activation, plain stores with a virtual one");
  DT_REQUIRE(!(d->accumulate && dt_conv2d_n16_supported(d)), "conv_bn_bwd: no join on the 16-wide kernels");
  return conv2d_impl(d, src0, nullptr, w, out0, nullptr, red, nullptr, nullptr, nullptr, stream, fuse);
}

static int conv2d_impl(const dt_conv_desc* d, const float* src0, const float* src1, const float* w, float* out0,
                       float* out1, float* stats, const float* in_scale, const float* in_shift, void* out_bf16,
                       void* stream, const dt_bn_bwd_fuse* fuse, const float* aff_scale, const float* aff_shift,
                       int aff_relu) {
  int rc = validate(d);
  if (rc != DT_OK) return rc;
  DT_REQUIRE(src0 && w && out0, "conv: null pointer");
  DT_REQUIRE(d->C1 == 0 || src1, "conv: src1 missing");
  DT_REQUIRE(d->cout_split == 0 || out1, "conv: out1 missing");
  DT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv: in_scale/in_shift must come together");
  DT_REQUIRE(in_scale == nullptr || (d->ksize == 3 && d->stride == 1 && d->mode0 != 2 && d->C0 <= DT_TF_MAXC),
             "conv: input transform needs a 3x3 stride-1 layer with C0 <= %d and no zero-insertion", DT_TF_MAXC);
  if (aff_scale != nullptr && aff_relu && src1 == nullptr && dt_conv2d_narrow_supported(d))
    return dt_conv2d_narrow_affine(d, src0, w, out0, aff_scale, aff_shift, nullptr, nullptr, stream);
  if (aff_scale == nullptr && out_bf16 == nullptr && dt_conv2d_narrow_supported(d))   // Cin, Cout in {16, 32} at full resolution: the lean kernel
    return dt_conv2d_narrow_launch(d, src0, w, out0, stats, in_scale, in_shift, (hipStream_t)stream, fuse);
  if (aff_scale == nullptr && dt_conv2d_n16_supported(d))
    return dt_conv2d_n16_launch(d, src0, w, out0, stats, in_scale, in_shift, (hipStream_t)stream, fuse);
  ConvCfg c = pick_cfg(d);
  ConvArgs a;
  a.bnb = fuse ? *fuse : dt_bn_bwd_fuse{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a.src0 = src0; a.src1 = src1; a.w = w; a.out0 = out0; a.out1 = out1; a.stats = stats;
  a.in_scale = in_scale; a.in_shift = in_shift; a.out0_bf16 = (__bf16*)out_bf16;
  a.aff_scale = aff_scale; a.aff_shift = aff_shift; a.aff_relu = aff_relu;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.cout_split = d->cout_split; a.pad = d->pad;
  a.accumulate = d->accumulate;
  a.tiles_x = dt_cdiv(d->Wo, c.tw);
  a.tiles_y = dt_cdiv(d->Ho, 256 / c.tw);
  a.n_tiles = dt_cdiv(d->Cout, c.tn);
  a.pack = conv_packs(d);
  a.P = d->B * a.tiles_x * a.tiles_y;            // statistics rows (packed tiles: still one per image)
  a.sp_tiles = a.pack ? dt_cdiv(d->B, 4) : a.P;  // spatial tiles of the grid
  hipStream_t st = (hipStream_t)stream;
  if (d->mode0 == 2 && d->stride == 1 && c.tw == 32 && d->C0 > 32) {   // transposed conv: parity-class tiles
    if (d->ksize == 3) return c.tn == 64 ? launch<3, 1, 32, 64, 16, true>(a, st) : launch<3, 1, 32, 32, 16, true>(a, st);
    if (d->ksize == 1) return c.tn == 64 ? launch<1, 1, 32, 64, 16, true>(a, st) : launch<1, 1, 32, 32, 16, true>(a, st);
  }
  if (d->ksize == 3 && d->stride == 1)
    return (d->C0 + d->C1 <= 32 && c.tn == 32) ? launch_tw_tn<3, 1, 8>(a, c, st) : launch_tw_tn<3, 1, 16>(a, c, st);
  if (d->ksize == 3 && d->stride == 2) return launch_tw_tn<3, 2, 8>(a, c, st);
  if (d->ksize == 1 && d->stride == 2) return launch_tw_tn<1, 2, 16>(a, c, st);
  if (d->ksize == 1 && d->stride == 1) return launch_tw_tn<1, 1, 16>(a, c, st);
  if (d->ksize == 7) {
    if (c.tw == 32) return launch<7, 2, 32, 64, 4>(a, st);
    if (c.tw == 16) return launch<7, 2, 16, 64, 4>(a, st);
    return launch<7, 2, 8, 64, 4>(a, st);
  }
  dt_set_error("conv: configuration not implemented");
  return DT_ENOSYS;
}

// ---------------------------------------------------------------- weight flip+transpose for dgrad
__global__ void weight_flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wd, int taps,
                                             int Cin, int Cout) {
  // wd[tap'][co][ci] = w[taps-1-tap'][ci][co]; 32x32 LDS transpose tiles
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  const float* wsrc = w + (size_t)(taps - 1 - tap) * Cin * Cout;
  for (int i = ty; i < 32; i += 8) {
    const int ci = ci0 + i, co = co0 + tx;
    tile[i][tx] = (ci < Cin && co < Cout) ? wsrc[(size_t)ci * Cout + co] : 0.f;
  }
  __syncthreads();
  float* wdst = wd + (size_t)tap * Cin * Cout;
  for (int i = ty; i < 32; i += 8) {
    const int co = co0 + i, ci = ci0 + tx;
    if (ci < Cin && co < Cout) wdst[(size_t)co * Cin + ci] = tile[tx][i];
  }
}

extern "C" int dt_weight_flip_transpose(const float* w, float* wd, int ksize, int Cin, int Cout, void* stream) {
  DT_REQUIRE(w && wd && ksize > 0 && Cin > 0 && Cout > 0, "flip_transpose: bad args");
  dim3 grid(dt_cdiv(Cout, 32), dt_cdiv(Cin, 32), ksize * ksize);
  hipLaunchKernelGGL(weight_flip_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wd, ksize * ksize,
                     Cin, Cout);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// which kernel instantiation dt_conv2d launches for a descriptor (profiling / roofline attribution)
// fp32 operands and accumulation, bf16 output (+ fp32 BatchNorm partial statistics): the stem of the bf16 path
extern "C" int dt_conv2d_out_bf16(const dt_conv_desc* d, const float* src0, const float* w_hwio, void* out_bf16,
                                  float* stats, void* stream) {
  DT_REQUIRE(d && d->cout_split == 0 && d->accumulate == 0 && d->C1 == 0 && out_bf16, "conv_out_bf16: bad args");
  DT_REQUIRE(!dt_conv2d_n16_supported(d), "conv_out_bf16: not built for the 16-wide kernels");
  return conv2d_impl(d, src0, nullptr, w_hwio, reinterpret_cast<float*>(out_bf16), nullptr, stats, nullptr, nullptr,
                     out_bf16, stream);
}

// 1 when dt_conv2d runs the parity-class (zero-insertion) tiles for this descriptor
extern "C" int dt_conv2d_uses_zi(const dt_conv_desc* d) {
  if (validate(d) != DT_OK) return 0;
  ConvCfg c = pick_cfg(d);
  return d->mode0 == 2 && d->stride == 1 && c.tw == 32 && d->C0 > 32 && (d->ksize == 3 || d->ksize == 1);
}

extern "C" int dt_conv2d_config(const dt_conv_desc* d, int* tw, int* tn, int* ck) {
  int rc = validate(d);
  if (rc != DT_OK) return rc;
  ConvCfg c = pick_cfg(d);
  if (dt_conv2d_narrow_supported(d)) {   // conv3x3_f32_narrow_kernel<CB, NB, ...>: reported as ck = 1000 + 10 CB + NB
    if (tw) *tw = 32;
    if (tn) *tn = d->Cout;
    if (ck) *ck = (dt_conv2d_narrow_subpixel(d) ? 2000 : 1000) + 10 * (d->C0 / 16) + d->Cout / 16;   // 2000 +: conv3x3_f32_upc_kernel
    return DT_OK;
  }
  if (dt_conv2d_n16_supported(d)) {   // conv_fwd_n16_kernel: 8x32 pixel tile, 16 output channels, CK 16
    if (tw) *tw = 32;
    if (tn) *tn = 16;
    if (ck) *ck = 16;
    return DT_OK;
  }
  if (tw) *tw = c.tw;
  if (tn) *tn = d->ksize == 7 ? 64 : c.tn;
  if (ck) *ck = d->ksize == 7 ? 4 : ((d->ksize == 3 && (d->stride == 2 || (d->C0 + d->C1 <= 32 && c.tn == 32))) ? 8 : 16);
  return DT_OK;
}
