// Segmentation head (3x3 conv 16->K + bias, logits NCHW, fused argmax) and the fused
// softmax + Dice-family / focal / boundary / F-score reductions with their one-pass backward.
//
// Replaces on the reference hot path:
//   smp SegmentationHead conv (reached from deadtrees/network/segmodel.py:214),
//   logits.softmax(dim=1)                              segmodel.py:216,237,282
//   class2one_hot                                      deadtrees/loss/losses.py:124-141 (never materialised)
//   GeneralizedDiceLoss / DiceLoss / FocalLoss / CrossEntropy / SurfaceLoss reductions
//                                                      loss/gdl.py:10-27, loss/losses.py:187-291
//   smp Fscore threshold + sums                        segmodel.py:145-149,202-208
//   argmax(dim=1)                                      segmodel.py:273,289; deployment/inference.py:62
// All HBM-bound: one read of the 16-channel decoder output / of the logits, wave64 shuffle
// reductions -> LDS -> one fp64 row per workgroup (fixed-order finalize, no float atomics).
#include "common.h"

#include <math.h>

#define HEAD_TW 32
#define HEAD_TH 8
#define HEAD_MAXK 4
#define HEAD_CIN 16

// ------------------------------------------------------------------ head forward
typedef __bf16 hbf16x4 __attribute__((ext_vector_type(4)));

// 4 consecutive channels of pixel `pix_elem_off` from an fp32 or bf16 NHWC tensor
template <bool XB>
__device__ __forceinline__ f32x4 head_load4(const void* x, size_t elem_off) {
  if constexpr (XB) {
    const hbf16x4 q = *reinterpret_cast<const hbf16x4*>(reinterpret_cast<const __bf16*>(x) + elem_off);
    return f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
  } else {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + elem_off);
  }
}

template <int K, bool XB = false>
__global__ __launch_bounds__(256) void head_fwd_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ logits,
                                                       int64_t* __restrict__ am64, uint8_t* __restrict__ am8, int B,
                                                       int H, int W) {
  constexpr int C = HEAD_CIN;
  constexpr int HH = HEAD_TH + 2, HW_ = HEAD_TW + 2;
  __shared__ float tile[HH * HW_][C + 1];  // +1: odd pitch -> conflict-free per-pixel reads
  // the K*9*16 weights are the same for every lane and their indices are compile-time constants after unrolling:
  // read straight from `w` they become scalar loads (SGPR operands of the FMAs) instead of one LDS read per FMA
  const int tiles_x = (W + HEAD_TW - 1) / HEAD_TW, tiles_y = (H + HEAD_TH - 1) / HEAD_TH;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
  const int oy0 = ty * HEAD_TH, ox0 = tx * HEAD_TW;
  const int t = threadIdx.x;
  for (int i = t; i < HH * HW_ * (C / 4); i += 256) {
    const int q = i % (C / 4), pix = i / (C / 4);
    const int hy = pix / HW_, hx = pix % HW_;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
      v = head_load4<XB>(x, (((size_t)b * H + iy) * W + ix) * C + 4 * q);
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[pix][4 * q + k] = v[k];
  }
  __syncthreads();
  const int py = t / HEAD_TW, px = t % HEAD_TW;
  const int oy = oy0 + py, ox = ox0 + px;
  float acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.f;
  // summation order kh -> kw -> cin (then + bias) per output, fp32 fma
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float* tp = tile[(py + kh) * HW_ + px + kw];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float xv = tp[c];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = fmaf(xv, w[(k * 9 + kh * 3 + kw) * C + c], acc[k]);
      }
    }
  if (oy < H && ox < W) {
    int best = 0;
    float bv = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float v = acc[k] + bias[k];
      logits[(((size_t)b * K + k) * H + oy) * W + ox] = v;
      if (k == 0 || v > bv) {  // ties -> lowest index (torch.argmax)
        bv = v;
        best = k;
      }
    }
    const size_t o = ((size_t)b * H + oy) * W + ox;
    if (am64) am64[o] = best;
    if (am8) am8[o] = (uint8_t)best;
  }
}

template <bool XB>
static int head_fwd_launch(const void* x, const float* w, const float* bias, float* logits, int64_t* am64, uint8_t* am8,
                           int B, int H, int W, int Cin, int K, void* stream) {
  DT_REQUIRE(x && w && bias && logits && B > 0 && H > 0 && W > 0, "head_fwd: bad args");
  DT_REQUIRE(Cin == HEAD_CIN, "head_fwd: Cin must be %d (decoder_channels[-1])", HEAD_CIN);
  DT_REQUIRE(K >= 1 && K <= HEAD_MAXK, "head_fwd: K=%d unsupported (1..%d)", K, HEAD_MAXK);
  const int grid = B * dt_cdiv(H, HEAD_TH) * dt_cdiv(W, HEAD_TW);
  hipStream_t st = (hipStream_t)stream;
  switch (K) {
    case 1: hipLaunchKernelGGL((head_fwd_kernel<1, XB>), dim3(grid), dim3(256), 0, st, x, w, bias, logits, am64, am8, B, H, W); break;
    case 2: hipLaunchKernelGGL((head_fwd_kernel<2, XB>), dim3(grid), dim3(256), 0, st, x, w, bias, logits, am64, am8, B, H, W); break;
    case 3: hipLaunchKernelGGL((head_fwd_kernel<3, XB>), dim3(grid), dim3(256), 0, st, x, w, bias, logits, am64, am8, B, H, W); break;
    default: hipLaunchKernelGGL((head_fwd_kernel<4, XB>), dim3(grid), dim3(256), 0, st, x, w, bias, logits, am64, am8, B, H, W); break;
  }
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_head_fwd(const float* x, const float* w, const float* bias, float* logits, int64_t* am64,
                           uint8_t* am8, int B, int H, int W, int Cin, int K, void* stream) {
  return head_fwd_launch<false>(x, w, bias, logits, am64, am8, B, H, W, Cin, K, stream);
}

extern "C" int dt_head_fwd_bf16(const void* x_bf16, const float* w, const float* bias, float* logits, int64_t* am64,
                                uint8_t* am8, int B, int H, int W, int Cin, int K, void* stream) {
  return head_fwd_launch<true>(x_bf16, w, bias, logits, am64, am8, B, H, W, Cin, K, stream);
}

// ------------------------------------------------------------------ head backward
// dx[b,y,x,c] = sum_{k,kh,kw} dl[b,k,y+1-kh,x+1-kw] * w[k][kh][kw][c]
// dW[k][kh][kw][c] = sum_pix x[b,y+kh-1,x+kw-1,c] * dl[b,k,y,x] ; dbias[k] = sum dl
//
// Both are tiny-K GEMMs over an 8x32-pixel tile held in LDS (x halo 10x34x16, dl halo K x 10x34) and run on
// v_mfma_f32_16x16x4_f32 (exact fp32 fma chains; the VALU form of round 1 was issue-bound at 29 % of the HBM rate):
//   dx^T [16 c x 16 pixels]    = w^T [16 c x 9K (k,tap)]      x dl_shifted [9K x 16 pixels]   (ceil(9K/4) MFMAs per 16 pixels;
//        D: lane = pixel, 4 consecutive channels -> ONE 16-byte (fp32) / 8-byte (bf16) store per lane, 1 KiB per wave)
//   dW   [9K (k,tap) x 16 c]   = dl_shifted^T [9K x pixels]   x x [pixels x 16 c]             (K loop = the 10 x 36 halo pixels,
//        4 per MFMA, the waves take every 4th step; accumulators live across the HB_TPW tiles of a workgroup)
// dbias falls out of the dW loop: the A operand of a centre-tap row is dl of the tile's own pixels.
#define HB_HWP 36                         // halo columns 34 + 2 zero columns: 9 K-steps of 4 pixels per halo row
#define HB_HH (HEAD_TH + 2)
#define HB_NPIX (HB_HH * HB_HWP)          // 360
#define HB_TPW 8                          // tiles per workgroup (one partial dW row per workgroup)

static inline int head_bwd_tiles(int B, int H, int W) { return B * dt_cdiv(H, HEAD_TH) * dt_cdiv(W, HEAD_TW); }
extern "C" int dt_head_bwd_rows(int B, int H, int W) { return dt_cdiv(head_bwd_tiles(B, H, W), HB_TPW); }

#define HB_ZP 40                          // pitch of the zero-bordered copy of the tile's own dl (12 rows x 38 columns used)
#define HB_ZN (12 * HB_ZP)

template <int K, bool XB = false>
// K <= 2: 4 workgroups per CU (<= 128 VGPRs; 5 spill: measured 708 -> 927 us); K = 3 / 4 need 134 / 208 registers
__global__ __launch_bounds__(256, K <= 2 ? 4 : (K == 3 ? 3 : 2)) void head_bwd_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ dl, void* __restrict__ dx,
                                                       float* __restrict__ red, int B, int H, int W, int total_tiles) {
  constexpr int C = HEAD_CIN;
  constexpr int NR = 9 * K;                 // real (k, tap) rows
  constexpr int MT = (NR + 15) / 16;        // 16-row M tiles of the dW GEMM
  constexpr int KS = (NR + 3) / 4;          // K steps of the dx GEMM
  constexpr int NW = K * 9 * C + K, NWP = (NW + 3) & ~3;
  constexpr int NSTEP = HB_HH * (HB_HWP / 4), SPW = (NSTEP + 3) / 4;   // 90 K steps of the dW GEMM, <= 23 per wave
  static_assert(4 * MT * 64 * 5 <= HB_NPIX * C, "the cross-wave reduction image fits the x tile");
  __shared__ __attribute__((aligned(16))) float xt[HB_NPIX * C];   // [halo pixel][16 channels]: 16 j + c -> 64 distinct banks
  __shared__ float dlt[K * HB_NPIX];        // dl with its halo (dx reads the neighbours' pixels)
  __shared__ float dlz[K * HB_ZN];          // the tile's OWN dl at (py + 2, px + 2) inside a border of zeros: the dW operand
                                            // dl[hy - kh, hx - kw] needs no range test for any halo pixel / tap
  const int tiles_x = (W + HEAD_TW - 1) / HEAD_TW, tiles_y = (H + HEAD_TH - 1) / HEAD_TH;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: the K-step bookkeeping below runs on the SALU
  const int m = lane & 15, j = lane >> 4;

  // ---- per-lane constants.  dx: A = w^T (row c = m, k index kk = 4 s + j), B = dl at halo (py + 2 - kh, px + 2 - kw)
  float wA[KS];
  int boff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int kk = 4 * s + j;
    const bool ok = kk < NR;
    const int k = ok ? kk / 9 : 0, tap = ok ? kk % 9 : 0;
    wA[s] = ok ? w[kk * C + m] : 0.f;
    boff[s] = k * HB_NPIX + (2 - tap / 3) * HB_HWP + (2 - tap % 3);
  }
  // dW: A row (k, tap) = 16 mt + m at halo pixel (hy, hx) = dlz[k][hy - kh + 2][hx - kw + 2]; rows past 9K read the
  // always-zero corner dlz[0][0][0 .. 3] with a zero stride
  int aoff[MT], amul[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int rg = 16 * mt + m;
    const bool ok = rg < NR;
    const int k = ok ? rg / 9 : 0, tap = ok ? rg % 9 : 0;
    aoff[mt] = ok ? k * HB_ZN + (2 - tap / 3) * HB_ZP + (2 - tap % 3) + j : 0;
    amul[mt] = ok ? 1 : 0;
  }
  const int xoff = j * C + m;
  f32x4 accw[MT];
  float asum[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    accw[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    asum[mt] = 0.f;
  }
  for (int i = t; i < K * HB_ZN; i += 256) dlz[i] = 0.f;   // the border stays zero; the interior is rewritten per tile

  // ---- staging: the tile's global loads are issued as one batch into registers (a loop of load -> LDS store pairs
  // would serialise 6 global latencies per tile), and the NEXT tile's batch is in flight while this tile is multiplied.
  // Element e = t + 256 it: halo position packed once per thread (hy << 8 | hx; 255 = not an element)
  constexpr int NXL = (HB_NPIX * (C / 4) + 255) / 256, NDL = (K * HB_NPIX + 255) / 256;
  int xpos[NXL], dpos[NDL];
#pragma unroll
  for (int it = 0; it < NXL; ++it) {
    const int pix = (t + it * 256) >> 2;
    const int hy = pix / HB_HWP, hx = pix - hy * HB_HWP;
    xpos[it] = (pix < HB_NPIX && hx < HEAD_TW + 2) ? (hy << 8 | hx) : (255 << 8 | 255);
  }
#pragma unroll
  for (int it = 0; it < NDL; ++it) {
    const int i = t + it * 256;
    const int k = i / HB_NPIX, pix = i - k * HB_NPIX;
    const int hy = pix / HB_HWP, hx = pix - hy * HB_HWP;
    dpos[it] = (k < K && hx < HEAD_TW + 2) ? (k << 16 | hy << 8 | hx) : (255 << 8 | 255);
  }
  f32x4 rx[NXL];
  float rd[NDL];
  auto issue_loads = [&](int tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int oy0 = ty * HEAD_TH, ox0 = tx * HEAD_TW;
#pragma unroll
    for (int it = 0; it < NXL; ++it) {
      const int iy = oy0 - 1 + ((xpos[it] >> 8) & 255), ix = ox0 - 1 + (xpos[it] & 255);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      // non-elements (sentinel 0xffff) load nothing whatever the image size: their LDS slots must be exact zeros
      if (xpos[it] != 0xffff && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        v = head_load4<XB>(x, (((size_t)b * H + iy) * W + ix) * C + 4 * (t & 3));
      rx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NDL; ++it) {
      const int iy = oy0 - 1 + ((dpos[it] >> 8) & 255), ix = ox0 - 1 + (dpos[it] & 255);
      float v = 0.f;
      if ((dpos[it] & 0xffff) != 0xffff && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        v = dl[(((size_t)b * K + (dpos[it] >> 16)) * H + iy) * W + ix];
      rd[it] = v;
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int it = 0; it < NXL; ++it) {
      const int i = t + it * 256;
      if (i < HB_NPIX * (C / 4)) *reinterpret_cast<f32x4*>(xt + i * 4) = rx[it];   // pix * 16 + 4 q = 4 i
    }
#pragma unroll
    for (int it = 0; it < NDL; ++it) {
      const int i = t + it * 256;
      if (i < K * HB_NPIX) {
        dlt[i] = rd[it];
        const int hy = (dpos[it] >> 8) & 255, hx = dpos[it] & 255;
        if (hy >= 1 && hy <= HEAD_TH && hx >= 1 && hx <= HEAD_TW)   // output pixel (hy - 1, hx - 1) of this tile
          dlz[(dpos[it] >> 16) * HB_ZN + (hy + 1) * HB_ZP + hx + 1] = rd[it];
      }
    }
  };
  const int tile0 = blockIdx.x * HB_TPW;
  issue_loads(tile0);     // host: grid = ceil(total / HB_TPW), so every workgroup has a first tile
#pragma unroll 1
  for (int ti = 0; ti < HB_TPW; ++ti) {
    const int tile = tile0 + ti;
    if (tile >= total_tiles) break;     // uniform
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int oy0 = ty * HEAD_TH, ox0 = tx * HEAD_TW;
    __syncthreads();                    // the previous tile's reads are done (first tile: the zero fill of dlz)
    write_lds();
    __syncthreads();
    if (ti + 1 < HB_TPW && tile + 1 < total_tiles) issue_loads(tile + 1);
    // ---- dx: wave w owns tile rows 2w, 2w+1 = 4 groups of 16 pixels
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int py = 2 * wave + (g >> 1), px = (g & 1) * 16 + m;
      const int base = py * HB_HWP + px;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) a = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[s], dlt[boff[s] + base], a, 0, 0, 0);
      const int oy = oy0 + py, ox = ox0 + px;
      if (oy < H && ox < W) {
        const size_t o = (((size_t)b * H + oy) * W + ox) * C + 4 * j;
        if constexpr (XB) {
          hbf16x4 v = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
          *reinterpret_cast<hbf16x4*>(reinterpret_cast<__bf16*>(dx) + o) = v;
        } else {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dx) + o) = a;
        }
      }
    }
    // ---- dW / dbias: wave w takes the K steps s = SPW w .. of the 90 (halo row hy = s / 9, columns 4 (s % 9) + j): the step
    // bookkeeping is scalar, a lane adds its constant to two scalar offsets, reads and multiplies
    // (the opaque scalar zero keeps hipcc from hoisting the unrolled loop's 23 x 3 per-lane LDS addresses out of the tile
    // loop: 200+ live registers and 2 waves per SIMD otherwise)
    int zs = 0;
    asm volatile("" : "+s"(zs));
    int s = wave * SPW + zs, hy = s / (HB_HWP / 4), sx = s - hy * (HB_HWP / 4);
#pragma unroll 4
    for (int it = 0; it < SPW; ++it) {
      const bool live = s < NSTEP;                                   // uniform (the last wave has fewer steps)
      const int zo = live ? hy * HB_ZP + 4 * sx : 0, xo = (live ? hy * HB_HWP + 4 * sx : 0) * C;
      const float bv = xt[xo + xoff];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float a = dlz[live ? aoff[mt] + amul[mt] * zo : 0];
        asum[mt] += a;
        accw[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, accw[mt], 0, 0, 0);
      }
      ++s;
      if (++sx == HB_HWP / 4) {
        sx = 0;
        ++hy;
      }
    }
  }
  // ---- one partial row per workgroup: the four waves' accumulators combined in a fixed order
  __syncthreads();
  float* rl = xt;   // [4 waves][MT][64 lanes][5]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float* d = rl + ((wave * MT + mt) * 64 + lane) * 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = accw[mt][i];
    d[4] = asum[mt];
  }
  __syncthreads();
  float* row = red + (size_t)blockIdx.x * NWP;
  for (int o = t; o < NWP; o += 256) {
    float v = 0.f;
    if (o < NR * C) {
      const int rg = o / C, c = o % C;          // D row rg = 16 mt + 4 jj + i lives in lane 16 jj + c, register i
      const int mt = rg >> 4, r16 = rg & 15;
      const int src = (mt * 64 + (r16 >> 2) * 16 + c) * 5 + (r16 & 3);
      v = (rl[src] + rl[src + MT * 320]) + (rl[src + 2 * MT * 320] + rl[src + 3 * MT * 320]);
    } else if (o < NW) {
      const int rg = 9 * (o - NR * C) + 4;      // centre tap of class k
      const int mt = rg >> 4, r16 = rg & 15;
      for (int wv = 0; wv < 4; ++wv)
        for (int jj = 0; jj < 4; ++jj) v += rl[((wv * MT + mt) * 64 + jj * 16 + r16) * 5 + 4];
    }
    row[o] = v;
  }
}

template <bool XB>
static int head_bwd_launch(const void* x, const float* w, const float* dl, void* dx, float* red, int B, int H, int W,
                           int Cin, int K, void* stream) {
  DT_REQUIRE(x && w && dl && dx && red && B > 0 && H > 0 && W > 0, "head_bwd: bad args");
  DT_REQUIRE(Cin == HEAD_CIN && K >= 1 && K <= HEAD_MAXK, "head_bwd: unsupported Cin/K");
  const int total = head_bwd_tiles(B, H, W), grid = dt_head_bwd_rows(B, H, W);
  hipStream_t st = (hipStream_t)stream;
  switch (K) {
    case 1: hipLaunchKernelGGL((head_bwd_kernel<1, XB>), dim3(grid), dim3(256), 0, st, x, w, dl, dx, red, B, H, W, total); break;
    case 2: hipLaunchKernelGGL((head_bwd_kernel<2, XB>), dim3(grid), dim3(256), 0, st, x, w, dl, dx, red, B, H, W, total); break;
    case 3: hipLaunchKernelGGL((head_bwd_kernel<3, XB>), dim3(grid), dim3(256), 0, st, x, w, dl, dx, red, B, H, W, total); break;
    default: hipLaunchKernelGGL((head_bwd_kernel<4, XB>), dim3(grid), dim3(256), 0, st, x, w, dl, dx, red, B, H, W, total); break;
  }
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_head_bwd(const float* x, const float* w, const float* dl, float* dx, float* red, int B, int H,
                           int W, int Cin, int K, void* stream) {
  return head_bwd_launch<false>(x, w, dl, dx, red, B, H, W, Cin, K, stream);
}

extern "C" int dt_head_bwd_bf16(const void* x_bf16, const float* w, const float* dl, void* dx_bf16, float* red, int B,
                                int H, int W, int Cin, int K, void* stream) {
  return head_bwd_launch<true>(x_bf16, w, dl, dx_bf16, red, B, H, W, Cin, K, stream);
}

__global__ __launch_bounds__(256) void colsum_f64_kernel(const float* __restrict__ red, int P, int N, int pitch,
                                                         float* __restrict__ out_a, int na,
                                                         float* __restrict__ out_b) {
  // out[j] = sum_p red[p][j]; 4 columns x 64 row-lanes per workgroup, fp64
  __shared__ double sh[64][5];
  const int j0 = blockIdx.x * 4;
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  double s = 0.0;
  if (j0 + cl < N)
    for (int p = rl; p < P; p += 64) s += (double)red[(size_t)p * pitch + j0 + cl];
  sh[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && j0 + cl < N) {
    double tsum = 0.0;
    for (int i = 0; i < 64; ++i) tsum += sh[i][cl];
    const int j = j0 + cl;
    if (j < na)
      out_a[j] = (float)tsum;
    else
      out_b[j - na] = (float)tsum;
  }
}

#define HEAD_RED_RB 64
static inline int head_pitch(int Cin, int K) { return (K * 9 * Cin + K + 3) & ~3; }

extern "C" int64_t dt_head_bwd_red_floats(int B, int H, int W, int Cin, int K) {
  const int64_t P = dt_head_bwd_rows(B, H, W);
  return (P + dt_reduce_rows_out((int)P, HEAD_RED_RB)) * head_pitch(Cin, K);
}

extern "C" int dt_head_bwd_finalize(float* red, int P, float* dw, float* dbias, int Cin, int K, void* stream) {
  DT_REQUIRE(red && dw && dbias && P > 0 && Cin == HEAD_CIN && K >= 1 && K <= HEAD_MAXK, "head_bwd_finalize: bad args");
  const int na = K * 9 * Cin, N = na + K, pitch = head_pitch(Cin, K);
  hipStream_t st = (hipStream_t)stream;
  const float* rows = red;
  if (P > 256) {
    float* stage = red + (size_t)P * pitch;
    int rc = dt_reduce_rows_launch(red, stage, 1, P, pitch, HEAD_RED_RB, st);
    if (rc != DT_OK) return rc;
    rows = stage;
    P = dt_reduce_rows_out(P, HEAD_RED_RB);
  }
  hipLaunchKernelGGL(colsum_f64_kernel, dim3(dt_cdiv(N, 4)), dim3(256), 0, st, rows, P, N, pitch, dw, na, dbias);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ fused softmax + loss / metric reductions
#define LOSS_PIX_PER_WG 4096

template <int K>
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ labels,
                                                           const float* __restrict__ dist,
                                                           const float* __restrict__ wassM,
                                                           const float* __restrict__ gwV, float gamma,
                                                           double* __restrict__ part, float* __restrict__ probs,
                                                           int32_t* __restrict__ err, int64_t HW, int wg_per_img) {
  // part: [B][wg_per_img][K][NACC] fp64
  const int b = blockIdx.x / wg_per_img, chunk = blockIdx.x % wg_per_img;
  const int64_t p0 = (int64_t)chunk * LOSS_PIX_PER_WG;
  int64_t p1 = p0 + LOSS_PIX_PER_WG;
  if (p1 > HW) p1 = HW;
  float acc[K][DT_LOSS_NACC];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < DT_LOSS_NACC; ++j) acc[k][j] = 0.f;
  const float* lg = logits + (size_t)b * K * HW;
  float Mw[K][K];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int l = 0; l < K; ++l) Mw[k][l] = wassM ? wassM[k * K + l] : 0.f;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
    float z[K], pr[K];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      z[k] = lg[(size_t)k * HW + p];
      m = fmaxf(m, z[k]);
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pr[k] = expf(z[k] - m);
      sum += pr[k];
    }
    const float inv = 1.f / sum;
    const int64_t lab = labels[(size_t)b * HW + p];
    if (lab < 0 || lab >= K) err[0] = 1;
    if (wassM) {
      // GWDICE applies softmax to the PROBABILITIES once more (gwdl.py:104 on segmodel.py:176's y_hat) and
      // weights them with the label-distance row of the pixel's class (gwdl.py:131-178)
      float q[K], qs = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        q[k] = expf(pr[k] * inv);
        qs += q[k];
      }
      const float qi = 1.f / qs;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float wass = 0.f;
#pragma unroll
        for (int l = 0; l < K; ++l) wass += Mw[k][l] * q[l] * qi;
        if (lab == k) {
          acc[k][8] += wass;
          acc[k][9] += gwV[p];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float pk = pr[k] * inv;
      if (probs) probs[((size_t)b * K + k) * HW + p] = pk;
      const float tk = (lab == k) ? 1.f : 0.f;
      const float lp = logf(pk + 1e-10f);
      const float om = 1.f - pk;
      const float wgt = (gamma == 2.f) ? om * om : (gamma == 0.f ? 1.f : powf(om, gamma));
      const float hard = pk > 0.5f ? 1.f : 0.f;
      acc[k][0] += tk;
      acc[k][1] += pk * tk;
      acc[k][2] += pk;
      acc[k][3] += wgt * tk * lp;
      acc[k][4] += tk * lp;
      if (dist) acc[k][5] += pk * dist[((size_t)b * K + k) * HW + p];
      acc[k][6] += tk * hard;
      acc[k][7] += hard;
    }
  }
  __shared__ double sh[4][K * DT_LOSS_NACC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < DT_LOSS_NACC; ++j) {
      const double v = wave_sum_d((double)acc[k][j]);
      if (lane == 0) sh[wave][k * DT_LOSS_NACC + j] = v;
    }
  __syncthreads();
  if (threadIdx.x < K * DT_LOSS_NACC) {
    const int i = threadIdx.x;
    part[(size_t)blockIdx.x * K * DT_LOSS_NACC + i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
  }
}

__global__ void seg_loss_finalize_kernel(const double* __restrict__ part, double* __restrict__ acc, int wg_per_img,
                                         int KN) {
  // acc[b][i] = sum_chunk part[b][chunk][i]  (fixed order)
  const int b = blockIdx.x, i = threadIdx.x;
  if (i < KN) {
    double s = 0.0;
    for (int c = 0; c < wg_per_img; ++c) s += part[((size_t)b * wg_per_img + c) * KN + i];
    acc[(size_t)b * KN + i] = s;
  }
}

extern "C" int64_t dt_seg_loss_acc_doubles(int B, int K, int H, int W) {
  const int64_t wpi = ((int64_t)H * W + LOSS_PIX_PER_WG - 1) / LOSS_PIX_PER_WG;
  return (int64_t)B * (1 + wpi) * K * DT_LOSS_NACC;
}

// scratch for partials lives behind acc: caller passes acc sized [B][K][NACC] + partial area; to keep
// the ABI allocation-free we require acc to have room for B*(1+wg_per_img)*K*NACC doubles.
extern "C" int dt_seg_loss_fwd(const float* logits, const int64_t* labels, const float* dist, const float* wass_m,
                               const float* gw_possum, float gamma, double* acc, float* probs, int32_t* err_flag, int B, int K, int H, int W,
                               void* stream) {
  DT_REQUIRE(logits && labels && acc && err_flag && B > 0 && H > 0 && W > 0, "seg_loss_fwd: bad args");
  DT_REQUIRE(K >= 2 && K <= HEAD_MAXK, "seg_loss_fwd: K=%d unsupported (2..%d)", K, HEAD_MAXK);
  DT_REQUIRE((wass_m == nullptr) == (gw_possum == nullptr), "seg_loss_fwd: wass_m and gw_possum go together");
  const int64_t HW = (int64_t)H * W;
  const int wpi = dt_cdiv(HW, LOSS_PIX_PER_WG);
  double* part = acc + (size_t)B * K * DT_LOSS_NACC;
  hipStream_t st = (hipStream_t)stream;
  const int grid = B * wpi;
  switch (K) {
    case 2: hipLaunchKernelGGL(seg_loss_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, logits, labels, dist, wass_m, gw_possum, gamma, part, probs, err_flag, HW, wpi); break;
    case 3: hipLaunchKernelGGL(seg_loss_fwd_kernel<3>, dim3(grid), dim3(256), 0, st, logits, labels, dist, wass_m, gw_possum, gamma, part, probs, err_flag, HW, wpi); break;
    default: hipLaunchKernelGGL(seg_loss_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, logits, labels, dist, wass_m, gw_possum, gamma, part, probs, err_flag, HW, wpi); break;
  }
  DT_LAUNCH_CHECK();
  hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(B), dim3(64), 0, st, part, acc, wpi, K * DT_LOSS_NACC);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

template <int K>
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ labels,
                                                           const float* __restrict__ dist,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ wfocal,
                                                           const float* __restrict__ wbound,
                                                           const float* __restrict__ gscale,
                                                           const float* __restrict__ wassM,
                                                           const float* __restrict__ wassC,
                                                           const float* __restrict__ gwG,
                                                           float* __restrict__ dlogits, int64_t HW, int64_t total) {
  const float wf = wfocal[0], gamma = wfocal[1];
  const float gs = gscale ? gscale[0] : 1.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / HW, p = i - b * HW;
    const float* lg = logits + (size_t)b * K * HW;
    float z[K], pr[K], g[K];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      z[k] = lg[(size_t)k * HW + p];
      m = fmaxf(m, z[k]);
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pr[k] = expf(z[k] - m);
      sum += pr[k];
    }
    const float inv = 1.f / sum;
    const int64_t lab = labels[i];
    float dot = 0.f;
    float gw[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gw[k] = 0.f;
    if (wassM && lab >= 0 && lab < K) {
      // d loss / d p through the second softmax q = softmax(p):  g_l = (wassC[b] + gwG[p]) * M[lab][l]
      float q[K], qs = 0.f, gq = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        q[k] = expf(pr[k] * inv);
        qs += q[k];
      }
      const float qi = 1.f / qs, c = wassC[b] + gwG[p];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        q[k] *= qi;
        gw[k] = c * wassM[lab * K + k];
        gq += gw[k] * q[k];
      }
#pragma unroll
      for (int k = 0; k < K; ++k) gw[k] = q[k] * (gw[k] - gq);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float pk = pr[k] * inv;
      pr[k] = pk;
      const float tk = (lab == k) ? 1.f : 0.f;
      float gk = coef[((size_t)b * K + k) * 2] * tk + coef[((size_t)b * K + k) * 2 + 1];
      if (wf != 0.f && tk != 0.f) {
        // d/dp [ -(1-p)^gamma * log(p+eps) ] / M
        const float om = 1.f - pk, pe = pk + 1e-10f;
        float d;
        if (gamma == 2.f)
          d = 2.f * om * logf(pe) - om * om / pe;
        else if (gamma == 0.f)
          d = -1.f / pe;
        else
          d = gamma * powf(om, gamma - 1.f) * logf(pe) - powf(om, gamma) / pe;
        gk += wf * d;
      }
      if (dist && wbound) gk += wbound[k] * dist[((size_t)b * K + k) * HW + p];
      gk += gw[k];
      g[k] = gk;
      dot += gk * pk;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) dlogits[((size_t)b * K + k) * HW + p] = gs * pr[k] * (g[k] - dot);
  }
}

extern "C" int dt_seg_loss_bwd(const float* logits, const int64_t* labels, const float* dist, const float* coef,
                               const float* wfocal, const float* wbound, const float* gscale, const float* wass_m,
                               const float* wass_coef, const float* gw_posgrad, float* dlogits, int B, int K, int H,
                               int W, void* stream) {
  DT_REQUIRE(logits && labels && coef && wfocal && dlogits && B > 0 && H > 0 && W > 0, "seg_loss_bwd: bad args");
  DT_REQUIRE(K >= 2 && K <= HEAD_MAXK, "seg_loss_bwd: K=%d unsupported", K);
  DT_REQUIRE((wass_m == nullptr) == (wass_coef == nullptr) && (wass_m == nullptr) == (gw_posgrad == nullptr),
             "seg_loss_bwd: wass_m, wass_coef and gw_posgrad go together");
  const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
  int64_t g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  hipStream_t st = (hipStream_t)stream;
  switch (K) {
    case 2: hipLaunchKernelGGL(seg_loss_bwd_kernel<2>, dim3((unsigned)g), dim3(256), 0, st, logits, labels, dist, coef, wfocal, wbound, gscale, wass_m, wass_coef, gw_posgrad, dlogits, HW, total); break;
    case 3: hipLaunchKernelGGL(seg_loss_bwd_kernel<3>, dim3((unsigned)g), dim3(256), 0, st, logits, labels, dist, coef, wfocal, wbound, gscale, wass_m, wass_coef, gw_posgrad, dlogits, HW, total); break;
    default: hipLaunchKernelGGL(seg_loss_bwd_kernel<4>, dim3((unsigned)g), dim3(256), 0, st, logits, labels, dist, coef, wfocal, wbound, gscale, wass_m, wass_coef, gw_posgrad, dlogits, HW, total); break;
  }
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ scalar algebra of the compound loss
// Everything SemSegment.calculate_loss / log_metrics (segmodel.py:169-208) does AFTER the big reductions — class
// weights of gdl.py:15-23, the dice / focal / boundary quotients of losses.py:187-291, gwdl.py:110-138, both smp
// F-scores — on the [B][K][NACC] sums, in fp64, by one thread: ~60 tiny ATen launches per step become one, and
// the coefficients the backward pass needs (coef, wfocal, wbound, wass_a, wass_c) never leave the device.
#define GW_EPS 2.220446049250313e-16   // np.spacing(1), loss/gwdl.py:92
#define LOSS_EPS 1e-10                 // loss/losses.py:19
__global__ void seg_loss_algebra_kernel(const double* __restrict__ acc, dt_loss_cfg cfg, int B, int K, double HW,
                                        float* __restrict__ parts, float* __restrict__ coef,
                                        float* __restrict__ wfocal, float* __restrict__ wbound,
                                        float* __restrict__ wass_a, float* __restrict__ wass_c) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  auto A = [&](int b, int k, int j) { return acc[((size_t)b * K + k) * DT_LOSS_NACC + j]; };
  for (int i = 0; i < B * K * 2; ++i) coef[i] = 0.f;
  double dice = 0.0;
  if (cfg.dice_kind == 2) {   // GWDICE, weighting_mode "default": alpha = 0 for background
    for (int b = 0; b < B; ++b) {
      double gtp = 0.0, ae = 0.0;
      for (int k = 0; k < K; ++k) {
        if (k > 0) gtp += A(b, k, 9);
        ae += A(b, k, 8);
      }
      const double den = 2.0 * gtp + ae + GW_EPS;
      dice += 1.0 - (2.0 * gtp + GW_EPS) / den;
      if (wass_a) wass_a[b] = (float)(2.0 * ae / (den * den) / B);
      if (wass_c) wass_c[b] = (float)((2.0 * gtp + GW_EPS) / (den * den) / B);
    }
    dice /= B;
  } else if (cfg.dice_kind == 0) {   // GDICE: whole-batch class weights from integer counts
    double N = 0.0, D = 0.0, w[HEAD_MAXK];
    for (int k = 0; k < K; ++k) {
      double S = 0.0, pt = 0.0, ps = 0.0;
      for (int b = 0; b < B; ++b) {
        S += A(b, k, 0);
        pt += A(b, k, 1);
        ps += A(b, k, 2);
      }
      w[k] = 1.0 / (S * S + 1e-9);
      N += w[k] * pt;
      D += w[k] * (S + ps);
    }
    dice = 1.0 - 2.0 * (N + 1e-9) / (D + 1e-9);
    for (int b = 0; b < B; ++b)
      for (int k = 0; k < K; ++k) {
        coef[((size_t)b * K + k) * 2] = (float)(-2.0 * w[k] / (D + 1e-9));
        coef[((size_t)b * K + k) * 2 + 1] = (float)(2.0 * w[k] * (N + 1e-9) / ((D + 1e-9) * (D + 1e-9)));
      }
  } else if (cfg.dice_kind == 1) {   // DICE: per sample, non-background classes
    const double nfg = K - 1;
    for (int b = 0; b < B; ++b)
      for (int k = 1; k < K; ++k) {
        const double I = A(b, k, 1), U = A(b, k, 2) + A(b, k, 0);
        dice += 1.0 - (2.0 * I + LOSS_EPS) / (U + LOSS_EPS);
        coef[((size_t)b * K + k) * 2] = (float)(-2.0 / (U + LOSS_EPS) / (B * nfg));
        coef[((size_t)b * K + k) * 2 + 1] = (float)((2.0 * I + LOSS_EPS) / ((U + LOSS_EPS) * (U + LOSS_EPS)) / (B * nfg));
      }
    dice /= (B * nfg);
  }
  double total = dice, boundary = 0.0, focal = 0.0;
  for (int k = 0; k < K; ++k) wbound[k] = 0.f;
  if (cfg.use_boundary) {
    const double scale = 1.0 / ((double)B * (K - 1) * HW);
    for (int b = 0; b < B; ++b)
      for (int k = 1; k < K; ++k) boundary += A(b, k, 5);
    boundary *= scale;
    total += (double)cfg.boundary_weight * boundary;
    for (int k = 1; k < K; ++k) wbound[k] = (float)((double)cfg.boundary_weight * scale);
  }
  double cnt = 0.0, foc = 0.0, ce = 0.0;
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < K; ++k) {
      cnt += A(b, k, 0);
      foc += A(b, k, 3);
      ce += A(b, k, 4);
    }
  wfocal[0] = 0.f;
  wfocal[1] = cfg.gamma;
  if (cfg.use_focal) {
    const double M = cnt + LOSS_EPS;
    focal = -foc / M;
    total += focal;
    wfocal[0] = (float)(1.0 / M);
  }
  auto fscore = [&](int k0) {
    double tps = 0.0, prs = 0.0, gts = 0.0;
    for (int b = 0; b < B; ++b)
      for (int k = k0; k < K; ++k) {
        tps += A(b, k, 6);
        prs += A(b, k, 7);
        gts += A(b, k, 0);
      }
    return (2.0 * tps + 1e-7) / (2.0 * tps + (gts - tps) + (prs - tps) + 1e-7);
  };
  parts[0] = (float)dice;
  parts[1] = (float)boundary;
  parts[2] = (float)focal;
  parts[3] = (float)(-ce / (cnt + LOSS_EPS));
  parts[4] = (float)fscore(1);
  parts[5] = (float)fscore(0);
  parts[6] = (float)total;
  parts[7] = (float)total;
}

extern "C" int dt_seg_loss_algebra(const double* acc, const dt_loss_cfg* cfg, int B, int K, int H, int W, float* parts,
                                   float* coef, float* wfocal, float* wbound, float* wass_a, float* wass_c,
                                   void* stream) {
  DT_REQUIRE(acc && cfg && parts && coef && wfocal && wbound && B > 0 && H > 0 && W > 0, "seg_loss_algebra: bad args");
  DT_REQUIRE(K >= 2 && K <= HEAD_MAXK, "seg_loss_algebra: K=%d unsupported (2..%d)", K, HEAD_MAXK);
  DT_REQUIRE(cfg->dice_kind >= 0 && cfg->dice_kind <= 3,
             "seg_loss_algebra: dice_kind %d (0 GDICE, 1 DICE, 2 GWDICE, 3 none)", cfg->dice_kind);
  DT_REQUIRE(cfg->dice_kind != 2 || (wass_a && wass_c), "seg_loss_algebra: GWDICE needs wass_a / wass_c");
  hipLaunchKernelGGL(seg_loss_algebra_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, *cfg, B, K,
                     (double)H * (double)W, parts, coef, wfocal, wbound, wass_a, wass_c);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ GWDICE cross-sample position sums
// loss/gwdl.py:180-198 multiplies alpha[B,1,S] with (1 - wass)[B,S]; the shapes broadcast to [B,B,S], so the
// "generalised true positives" of sample i are  sum_s alpha_i(s) * V(s)  with  V(s) = sum_j (1 - wass_j(s))
// over ALL samples j.  The reference trains with that; these two small passes reproduce it:
//   fwd:  V[s] = sum_j (1 - wass_j(s))                       (sequential over j: fixed order)
//   bwd:  G[s] = sum_i a[i] * alpha(label_i(s)),  alpha = [k > 0]   (d loss / d wass_j(s) = G[s] + c[j])
template <int K>
__global__ __launch_bounds__(256) void gwdice_possum_kernel(const float* __restrict__ logits,
                                                            const int64_t* __restrict__ labels,
                                                            const float* __restrict__ wassM, float* __restrict__ V,
                                                            int B, int64_t HW) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  float v = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* lg = logits + (size_t)b * K * HW;
    float z[K], m = -INFINITY, sum = 0.f, qs = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      z[k] = lg[(size_t)k * HW + p];
      m = fmaxf(m, z[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      z[k] = expf(z[k] - m);
      sum += z[k];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      z[k] = expf(z[k] * inv);
      qs += z[k];
    }
    const float qi = 1.f / qs;
    const int64_t lab = labels[(size_t)b * HW + p];
    float wass = 0.f;
    if (lab >= 0 && lab < K) {
#pragma unroll
      for (int k = 0; k < K; ++k) wass += wassM[lab * K + k] * z[k] * qi;
    }
    v += 1.f - wass;
  }
  V[p] = v;
}

__global__ __launch_bounds__(256) void gwdice_posgrad_kernel(const int64_t* __restrict__ labels,
                                                             const float* __restrict__ a, float* __restrict__ G,
                                                             int B, int64_t HW) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  float g = 0.f;
  for (int b = 0; b < B; ++b)
    if (labels[(size_t)b * HW + p] > 0) g += a[b];
  G[p] = g;
}

extern "C" int dt_gwdice_possum(const float* logits, const int64_t* labels, const float* wass_m, float* possum, int B,
                                int K, int H, int W, void* stream) {
  DT_REQUIRE(logits && labels && wass_m && possum && B > 0 && H > 0 && W > 0, "gwdice_possum: bad args");
  DT_REQUIRE(K >= 2 && K <= HEAD_MAXK, "gwdice_possum: K=%d unsupported", K);
  const int64_t HW = (int64_t)H * W;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(dt_cdiv(HW, 256));
  switch (K) {
    case 2: hipLaunchKernelGGL(gwdice_possum_kernel<2>, grid, dim3(256), 0, st, logits, labels, wass_m, possum, B, HW); break;
    case 3: hipLaunchKernelGGL(gwdice_possum_kernel<3>, grid, dim3(256), 0, st, logits, labels, wass_m, possum, B, HW); break;
    default: hipLaunchKernelGGL(gwdice_possum_kernel<4>, grid, dim3(256), 0, st, logits, labels, wass_m, possum, B, HW); break;
  }
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_gwdice_posgrad(const int64_t* labels, const float* sample_coef, float* posgrad, int B, int H, int W,
                                 void* stream) {
  DT_REQUIRE(labels && sample_coef && posgrad && B > 0 && H > 0 && W > 0, "gwdice_posgrad: bad args");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(gwdice_posgrad_kernel, dim3(dt_cdiv(HW, 256)), dim3(256), 0, (hipStream_t)stream, labels,
                     sample_coef, posgrad, B, HW);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ confusion matrices (eval reductions)
// Replaces torchmetrics.functional.confusion_matrix over the concatenated int64 masks of a whole epoch
// (reference deadtrees/network/segmodel.py:291-309, 337-365: 4 passes over N*H*W int64 x3 kept in memory):
// per batch, counts[0][t][p] += 1 for every pixel and counts[1][t][p] += 1 where lu == 1 (forest mask).
// Integer atomics: exact and order independent.
__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ pred, const uint8_t* __restrict__ pred8,
                                                        const int64_t* __restrict__ target,
                                                        const int64_t* __restrict__ lu, int K, int64_t n,
                                                        unsigned long long* __restrict__ counts, int32_t* __restrict__ err) {
  __shared__ unsigned int hist[2 * HEAD_MAXK * HEAD_MAXK];
  for (int i = threadIdx.x; i < 2 * K * K; i += 256) hist[i] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t p = pred ? pred[i] : (int64_t)pred8[i];
    const int64_t t = target[i];
    if (p < 0 || p >= K || t < 0 || t >= K) {
      err[0] = 1;
      continue;
    }
    atomicAdd(&hist[t * K + p], 1u);
    if (lu && lu[i] == 1) atomicAdd(&hist[K * K + t * K + p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * K * K; i += 256)
    if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

extern "C" int dt_confusion_matrix(const int64_t* pred_i64, const uint8_t* pred_u8, const int64_t* target,
                                   const int64_t* lu, int K, int64_t n, int64_t* counts, int32_t* err_flag,
                                   void* stream) {
  DT_REQUIRE((pred_i64 != nullptr) != (pred_u8 != nullptr), "confusion: exactly one of pred_i64 / pred_u8");
  DT_REQUIRE(target && counts && err_flag && n > 0 && K >= 2 && K <= HEAD_MAXK, "confusion: bad args");
  int64_t g = (n + 256 * 16 - 1) / (256 * 16);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pred_i64, pred_u8, target,
                     lu, K, n, (unsigned long long*)counts, err_flag);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ ensemble vote
// deployment/inference.py:65-116 PyTorchEnsembleInference.run: per-pixel torch.mode over the class maps of an odd
// number of models.  torch.mode returns the SMALLEST of the most frequent values (checked against torch on the CPU
// in tests/test_surface_gpu.py); counts live in registers, 4 pixels per lane (one dword per model).
#define VOTE_MAXK 8
__global__ __launch_bounds__(256) void ensemble_vote_kernel(const uint32_t* __restrict__ maps, int M, int64_t n4, int K,
                                                            uint32_t* __restrict__ out_u8, int64_t* __restrict__ out_i64,
                                                            int32_t* __restrict__ err) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    unsigned cnt[4][VOTE_MAXK];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int k = 0; k < VOTE_MAXK; ++k) cnt[p][k] = 0;
    bool bad = false;
    for (int m = 0; m < M; ++m) {
      const uint32_t v = maps[(size_t)m * n4 + i];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const unsigned c = (v >> (8 * p)) & 0xffu;
        bad |= c >= (unsigned)K;
#pragma unroll
        for (int k = 0; k < VOTE_MAXK; ++k) cnt[p][k] += (c == (unsigned)k) ? 1u : 0u;
      }
    }
    if (bad) err[0] = 1;
    uint32_t packed = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned best = 0, bc = cnt[p][0];
#pragma unroll
      for (int k = 1; k < VOTE_MAXK; ++k)
        if (cnt[p][k] > bc) {   // strict: ties keep the smaller class (torch.mode)
          bc = cnt[p][k];
          best = k;
        }
      packed |= best << (8 * p);
      if (out_i64) out_i64[4 * i + p] = (int64_t)best;
    }
    if (out_u8) out_u8[i] = packed;
  }
}

extern "C" int dt_ensemble_vote(const uint8_t* maps, int M, int64_t n, int K, uint8_t* out_u8, int64_t* out_i64,
                                int32_t* err_flag, void* stream) {
  DT_REQUIRE(maps && err_flag && (out_u8 || out_i64) && M > 0 && n > 0, "ensemble_vote: bad args");
  DT_REQUIRE((n & 3) == 0, "ensemble_vote: pixel count must be a multiple of 4 (n=%lld)", (long long)n);
  DT_REQUIRE(K >= 2 && K <= VOTE_MAXK, "ensemble_vote: K=%d unsupported (2..%d)", K, VOTE_MAXK);
  const int64_t n4 = n / 4;
  int64_t g = (n4 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(ensemble_vote_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)maps,
                     M, n4, K, (uint32_t*)out_u8, out_i64, err_flag);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
