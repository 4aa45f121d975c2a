// bf16-storage / fp32-accumulate forward path (BASELINE configs[2] "bf16 mixed precision", inference leg first).
//
// Same direct, im2col-free structure as conv_fwd.hip, on v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate):
//   * activations NHWC bf16, weights repacked [tap][Cout][Cin] bf16 (the 8 consecutive k of an MFMA B fragment
//     are 16 contiguous bytes);
//   * LDS images are pixel-major / cout-major rows of 32 channels (64 B) at an 80-B pitch: the 16-lane groups of a
//     ds_read_b128 fragment read then cover all 64 banks (pitch/4 = 20 dwords, 20/4 odd) -> conflict-free;
//   * one 16-B ds_read_b128 per lane feeds one MFMA operand (8 bf16); 2x2 MFMA tiles per wave;
//   * nearest-upsample / concat / the producer's BatchNorm-apply + ReLU are applied while staging, like fp32;
//   * fp32 accumulators, one rounding to bf16 at the store (v_cvt_pk_bf16_f32 keeps NaN a NaN).
// Replaces the same ATen ops as conv_fwd.hip for the autocast/AMP configuration the reference trains with
// (protocol.md:27 "AMP"; configs/trainer/default.yaml precision key).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct ConvBfArgs {
  const __bf16* src0;
  const __bf16* src1;
  const __bf16* w;  // [tap][Cout][Cin]
  const float* in_scale;
  const float* in_shift;
  __bf16* out;
  int B, Hin, Win, C0, C1, mode0, Ho, Wo, Cout, pad, tiles_x, tiles_y, n_tiles, P;
};

#define BF_CK 32
#define BF_PITCH 40   // bf16 elements per LDS row (32 used): 80 bytes
#define BF_TF_MAXC 512

template <int KS, int STRIDE, int TW>
struct BfGeom {
  static constexpr int TH = 256 / TW;
  static constexpr int LS = (KS == 1) ? 1 : STRIDE;
  static constexpr int GS = (KS == 1) ? STRIDE : 1;
  static constexpr int HALO_H = (TH - 1) * LS + KS;
  static constexpr int HALO_W = (TW - 1) * LS + KS;
  static constexpr int TAPS = KS * KS;
};

template <int KS, int STRIDE, int TW, int TN, bool TF>
__global__ __launch_bounds__(256, 2) void conv_fwd_bf16_kernel(const ConvBfArgs a) {
  using G = BfGeom<KS, STRIDE, TW>;
  constexpr int NT = TN / 32;
  constexpr int IN_ROWS = G::HALO_H * G::HALO_W;
  constexpr int W_ROWS = G::TAPS * TN;
  __shared__ __attribute__((aligned(16))) __bf16 lds[(IN_ROWS + W_ROWS) * BF_PITCH];
  __shared__ __attribute__((aligned(16))) float lds_tf[TF ? 2 * BF_TF_MAXC : 4];
  __bf16* lds_in = lds;
  __bf16* lds_w = lds + IN_ROWS * BF_PITCH;
  if constexpr (TF) {
    for (int i = threadIdx.x; i < a.C0; i += 256) {
      lds_tf[i] = a.in_scale[i];
      lds_tf[BF_TF_MAXC + i] = a.in_shift[i];
    }
  }
  const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int nt = wg % a.n_tiles, sp = wg / a.n_tiles;
  const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
  const int oy0 = ty * G::TH, ox0 = tx * TW, n0 = nt * TN;
  const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, r = lane & 31;

  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = (wave * 2 + mt) * 32 + r;
    abase[mt] = ((p / TW) * G::LS * G::HALO_W + (p % TW) * G::LS) * BF_PITCH + h * 8;
  }
  const int bbase = r * BF_PITCH + h * 8;

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][j][i] = 0.f;

  // staging bookkeeping: 4 x 16-byte segments (8 channels) per row
  constexpr int IN_TOTAL = IN_ROWS * 4, IN_IT = (IN_TOTAL + 255) / 256;
  constexpr int W_TOTAL = W_ROWS * 4, W_IT = (W_TOTAL + 255) / 256;
  const int q = tid & 3, row0 = tid >> 2;
  const int Cin = a.C0 + a.C1;
  const int Hs0 = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws0 = a.mode0 ? (a.Win >> 1) : a.Win;
  int pidx0[IN_IT], pidx1[IN_IT];
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int pix = row0 + it * 64;
    const int hy = pix / G::HALO_W, hx = pix - hy * G::HALO_W;
    const int iy = iy0 + hy * G::GS, ix = ix0 + hx * G::GS;
    const bool inb = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win && pix < IN_ROWS;
    const int sy = a.mode0 ? (iy >> 1) : iy, sx = a.mode0 ? (ix >> 1) : ix;
    pidx0[it] = inb ? (b * Hs0 + sy) * Ws0 + sx : -1;
    pidx1[it] = inb ? (b * a.Hin + iy) * a.Win + ix : -1;
  }
  int woff[W_IT];
#pragma unroll
  for (int it = 0; it < W_IT; ++it) {
    const int row = row0 + it * 64;  // tap*TN + n
    const int tap = row / TN, n = row - tap * TN;
    woff[it] = (row < W_ROWS && n0 + n < a.Cout) ? (tap * a.Cout + n0 + n) * Cin + 8 * q : -1;
  }
  f32x4 rin[IN_IT], rw[W_IT];   // 16 bytes each (8 bf16), carried as raw bits
  auto issue_loads = [&](int c0) {
    const bool use0 = c0 < a.C0;
    const __bf16* src = use0 ? a.src0 : a.src1;
    const int C = use0 ? a.C0 : a.C1;
    const int cc = (use0 ? c0 : c0 - a.C0) + 8 * q;
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
      if (woff[it] >= 0 && c0 + 8 * q < Cin) v = *reinterpret_cast<const f32x4*>(a.w + (size_t)woff[it] + c0);
      rw[it] = v;
    }
  };
  auto write_lds = [&](int c0) {
    bool tf_on = false;
    float sc[8], sh[8];
    if constexpr (TF) {
      const int cc = c0 + 8 * q;
      tf_on = cc < a.C0;
      if (tf_on) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          sc[k] = lds_tf[cc + k];
          sh[k] = lds_tf[BF_TF_MAXC + cc + k];
        }
      }
    }
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int pix = row0 + it * 64;
      if (IN_TOTAL % 256 == 0 || pix < IN_ROWS) {
        f32x4 raw = rin[it];
        if (TF && tf_on && pidx0[it] >= 0) {
          bf16x8 v = *reinterpret_cast<bf16x8*>(&raw);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float f = (float)v[k] * sc[k] + sh[k];
            f = f < 0.f ? 0.f : f;
            v[k] = (__bf16)f;
          }
          raw = *reinterpret_cast<f32x4*>(&v);
        }
        *reinterpret_cast<f32x4*>(lds_in + pix * BF_PITCH + 8 * q) = raw;
      }
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int row = row0 + it * 64;
      if (W_TOTAL % 256 == 0 || row < W_ROWS) *reinterpret_cast<f32x4*>(lds_w + row * BF_PITCH + 8 * q) = rw[it];
    }
  };

  issue_loads(0);
  for (int c0 = 0; c0 < Cin; c0 += BF_CK) {
    __syncthreads();
    write_lds(c0);
    __syncthreads();
    if (c0 + BF_CK < Cin) issue_loads(c0 + BF_CK);
#pragma unroll
    for (int tap = 0; tap < G::TAPS; ++tap) {
      const int kh = tap / KS, kw = tap % KS;
#pragma unroll
      for (int ks = 0; ks < BF_CK / 16; ++ks) {
        bf16x8 av[2], bv[NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          av[mt] = *reinterpret_cast<const bf16x8*>(lds_in + abase[mt] + (kh * G::HALO_W + kw) * BF_PITCH + ks * 16);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bv[j] = *reinterpret_cast<const bf16x8*>(lds_w + bbase + (tap * TN + 32 * j) * BF_PITCH + ks * 16);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt], bv[j], acc[mt][j], 0, 0, 0);
      }
    }
  }
  // epilogue: D col = lane&31 (channel), row = (i&3) + 8*(i>>2) + 4*(lane>>5) (pixel)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + 32 * j + r;
    if (n >= a.Cout) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int mrow = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int p = (wave * 2 + mt) * 32 + mrow;
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        if (oy < a.Ho && ox < a.Wo) a.out[(((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + n] = (__bf16)acc[mt][j][i];
      }
  }
}

static int bf_validate(const dt_conv_desc* d) {
  DT_REQUIRE(d != nullptr, "conv_bf16: null descriptor");
  DT_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->C0 > 0 && d->C1 >= 0 && d->Cout > 0, "conv_bf16: bad sizes");
  DT_REQUIRE((d->ksize == 3 && (d->stride == 1 || d->stride == 2)) || (d->ksize == 1 && d->stride == 2),
             "conv_bf16: ksize/stride (%d,%d) unsupported", d->ksize, d->stride);
  DT_REQUIRE((d->C0 & 7) == 0 && (d->C1 & 7) == 0, "conv_bf16: channels must be multiples of 8");
  DT_REQUIRE(d->C1 == 0 || (d->C0 % BF_CK) == 0, "conv_bf16: concat needs C0 %% 32 == 0");
  DT_REQUIRE(d->mode0 == 0 || d->mode0 == 1, "conv_bf16: mode0 %d unsupported", d->mode0);
  DT_REQUIRE(d->mode0 == 0 || ((d->Hin & 1) == 0 && (d->Win & 1) == 0), "conv_bf16: mode0 needs even Hin/Win");
  DT_REQUIRE(d->cout_split == 0 && d->accumulate == 0, "conv_bf16: split/accumulate not built");
  const int ho = (d->Hin + 2 * d->pad - d->ksize) / d->stride + 1, wo = (d->Win + 2 * d->pad - d->ksize) / d->stride + 1;
  DT_REQUIRE(ho == d->Ho && wo == d->Wo, "conv_bf16: Ho/Wo mismatch");
  return DT_OK;
}

template <int KS, int STRIDE, int TW, int TN>
static int bf_launch(const ConvBfArgs& a, hipStream_t st) {
  const long grid = (long)a.P * a.n_tiles;
  if (a.in_scale != nullptr)
    hipLaunchKernelGGL((conv_fwd_bf16_kernel<KS, STRIDE, TW, TN, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_fwd_bf16_kernel<KS, STRIDE, TW, TN, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

template <int KS, int STRIDE>
static int bf_dispatch(const ConvBfArgs& a, int tw, int tn, hipStream_t st) {
  if (tn == 64) {
    if (tw == 32) return bf_launch<KS, STRIDE, 32, 64>(a, st);
    if (tw == 16) return bf_launch<KS, STRIDE, 16, 64>(a, st);
    return bf_launch<KS, STRIDE, 8, 64>(a, st);
  }
  if (tw == 32) return bf_launch<KS, STRIDE, 32, 32>(a, st);
  if (tw == 16) return bf_launch<KS, STRIDE, 16, 32>(a, st);
  return bf_launch<KS, STRIDE, 8, 32>(a, st);
}

extern "C" int dt_conv2d_bf16(const dt_conv_desc* d, const void* src0, const void* src1, const void* w_bf16,
                              void* out, const float* in_scale, const float* in_shift, void* stream) {
  int rc = bf_validate(d);
  if (rc != DT_OK) return rc;
  DT_REQUIRE(src0 && w_bf16 && out, "conv_bf16: null pointer");
  DT_REQUIRE(d->C1 == 0 || src1, "conv_bf16: src1 missing");
  DT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv_bf16: in_scale/in_shift must come together");
  DT_REQUIRE(in_scale == nullptr || d->C0 <= BF_TF_MAXC, "conv_bf16: input transform needs C0 <= %d", BF_TF_MAXC);
  const int tw = d->Wo > 16 ? 32 : (d->Wo > 8 ? 16 : 8);
  int tn = d->Cout >= 64 ? 64 : 32;
  if (tn == 64) {
    const long wgs = (long)d->B * dt_cdiv(d->Ho, 256 / tw) * dt_cdiv(d->Wo, tw) * dt_cdiv(d->Cout, 64);
    if (wgs < 512) tn = 32;
  }
  ConvBfArgs a;
  a.src0 = (const __bf16*)src0; a.src1 = (const __bf16*)src1; a.w = (const __bf16*)w_bf16;
  a.in_scale = in_scale; a.in_shift = in_shift; a.out = (__bf16*)out;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.pad = d->pad;
  a.tiles_x = dt_cdiv(d->Wo, tw); a.tiles_y = dt_cdiv(d->Ho, 256 / tw); a.n_tiles = dt_cdiv(d->Cout, tn);
  a.P = d->B * a.tiles_x * a.tiles_y;
  hipStream_t st = (hipStream_t)stream;
  if (d->ksize == 3 && d->stride == 1) return bf_dispatch<3, 1>(a, tw, tn, st);
  if (d->ksize == 3 && d->stride == 2) return bf_dispatch<3, 2>(a, tw, tn, st);
  return bf_dispatch<1, 2>(a, tw, tn, st);
}

// ------------------------------------------------------------------ weights: fp32 HWIO -> bf16 [tap][Cout][Cin]
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int taps, int Cin,
                                         int Cout) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z, ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int ci = ci0 + i, co = co0 + tx;
    tile[i][tx] = (ci < Cin && co < Cout) ? w[((size_t)tap * Cin + ci) * Cout + co] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int co = co0 + i, ci = ci0 + tx;
    if (ci < Cin && co < Cout) out[((size_t)tap * Cout + co) * Cin + ci] = (__bf16)tile[tx][i];
  }
}

extern "C" int dt_pack_weights_bf16(const float* w_hwio, void* out, int ksize, int Cin, int Cout, void* stream) {
  DT_REQUIRE(w_hwio && out && ksize > 0 && Cin > 0 && Cout > 0, "pack_weights_bf16: bad args");
  dim3 grid(dt_cdiv(Cout, 32), dt_cdiv(Cin, 32), ksize * ksize);
  hipLaunchKernelGGL(pack_weights_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, w_hwio, (__bf16*)out,
                     ksize * ksize, Cin, Cout);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ bf16 elementwise passes (inference leg)
// out = act(y*scale + shift + res'), y fp32 (stem output) or bf16, res bf16 with optional affine, out bf16
template <bool Y_F32>
__global__ __launch_bounds__(256) void bn_act_bf16_kernel(const void* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const __bf16* __restrict__ res,
                                                          const float* __restrict__ rscale,
                                                          const float* __restrict__ rshift, __bf16* __restrict__ out,
                                                          int64_t n8, int C8, int relu) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const int c = (int)(i % C8) * 8;
    float v[8];
    if (Y_F32) {
      const f32x4 a = reinterpret_cast<const f32x4*>(y)[2 * i], bq = reinterpret_cast<const f32x4*>(y)[2 * i + 1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = a[k];
        v[4 + k] = bq[k];
      }
    } else {
      const bf16x8 a = reinterpret_cast<const bf16x8*>(y)[i];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (float)a[k];
    }
    bf16x8 rv;
    if (res) rv = reinterpret_cast<const bf16x8*>(res)[i];
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float f = v[k] * scale[c + k] + shift[c + k];
      if (res) {
        float rr = (float)rv[k];
        if (rscale) rr = rr * rscale[c + k] + rshift[c + k];
        f += rr;
      }
      if (relu) f = f < 0.f ? 0.f : f;
      o[k] = (__bf16)f;
    }
    reinterpret_cast<bf16x8*>(out)[i] = o;
  }
}

extern "C" int dt_bn_act_bf16(const void* y, int y_is_f32, const float* scale, const float* shift, const void* res,
                              const float* rscale, const float* rshift, void* out, int64_t n_pix, int C, int relu,
                              void* stream) {
  DT_REQUIRE(y && scale && shift && out && n_pix > 0 && C > 0 && (C & 7) == 0, "bn_act_bf16: bad args (C%%8)");
  DT_REQUIRE((rscale == nullptr) == (rshift == nullptr), "bn_act_bf16: rscale/rshift must come together");
  const int64_t n8 = n_pix * C / 8;
  int64_t g = (n8 + 255) / 256;
  if (g > 4096) g = 4096;
  if (y_is_f32)
    hipLaunchKernelGGL(bn_act_bf16_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, y, scale, shift,
                       (const __bf16*)res, rscale, rshift, (__bf16*)out, n8, C / 8, relu);
  else
    hipLaunchKernelGGL(bn_act_bf16_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, y, scale, shift,
                       (const __bf16*)res, rscale, rshift, (__bf16*)out, n8, C / 8, relu);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

__global__ __launch_bounds__(256) void maxpool_bf16_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ out, int B,
                                                           int H, int W, int C8, int Ho, int Wo) {
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % C8);
    int64_t rr = i / C8;
    const int ox = (int)(rr % Wo);
    rr /= Wo;
    const int oy = (int)(rr % Ho);
    const int b = (int)(rr / Ho);
    float best[8];
    bool first = true;
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy - 1 + kh;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = 2 * ox - 1 + kw;
        if ((unsigned)ix >= (unsigned)W) continue;
        const bf16x8 v = x[(((int64_t)b * H + iy) * W + ix) * C8 + c8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float f = (float)v[k];
          if (first || f > best[k] || f != f) best[k] = f;
        }
        first = false;
      }
    }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (__bf16)best[k];
    out[i] = o;
  }
}

extern "C" int dt_maxpool3x3s2_bf16(const void* x, void* out, int B, int H, int W, int C, void* stream) {
  DT_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "maxpool_bf16: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  int64_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(maxpool_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x,
                     (bf16x8*)out, B, H, W, C / 8, Ho, Wo);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// bf16 -> fp32 (feeds the fp32 head kernel; 16 channels at full resolution)
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16x8* __restrict__ x, f32x4* __restrict__ out, int64_t n8) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const bf16x8 v = x[i];
    f32x4 a, bq;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = (float)v[k];
      bq[k] = (float)v[4 + k];
    }
    out[2 * i] = a;
    out[2 * i + 1] = bq;
  }
}

extern "C" int dt_bf16_to_f32(const void* x, float* out, int64_t n, void* stream) {
  DT_REQUIRE(x && out && n > 0 && (n & 7) == 0, "bf16_to_f32: n must be a multiple of 8");
  int64_t g = (n / 8 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x,
                     (f32x4*)out, n / 8);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
