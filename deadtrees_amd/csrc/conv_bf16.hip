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

#include "conv_bf16.h"

#define BF_CK 32      // channels per chunk of the wide kernels; LDS rows hold CK channels at a pitch of CK + 8 bf16:
                      // 80 B (CK 32) or 48 B (CK 16) — both keep every 8-lane ds_read_b128 / ds_write_b128 group on
                      // distinct banks
#define BF_TF_MAXC 512
#define BF_NARROW_CIN 32   // up to this many input channels the 16-channel-chunk kernel is used (if Cout < 64)

template <int KS, int STRIDE, int TW, int TPX>
struct BfGeom {
  static constexpr int TH = TPX / TW;
  static constexpr int LS = (KS == 1) ? 1 : STRIDE;
  static constexpr int GS = (KS == 1) ? STRIDE : 1;
  static constexpr int HALO_H = (TH - 1) * LS + KS;
  static constexpr int HALO_W = (TW - 1) * LS + KS;
  static constexpr int TAPS = KS * KS;
};

// CK = 16 (with TN = 32) is the variant for the full-resolution decoder end (Cin <= 32): 30 KB of LDS and ~100
// VGPRs instead of 50 KB / 140, so five workgroups per CU instead of three keep loads in flight — those layers
// are bound by memory latency per workgroup, not by MFMA or LDS.
// MT = MFMA row tiles (32 pixels) per wave: 2 -> 256-pixel workgroup tiles; 4 (with CK = 16) -> 512-pixel tiles.
// The bf16 kernels are bound by global->LDS staging: a 512 x 64 tile stages 248 FLOP per byte
// instead of 161 and needs 0.75 instead of 1 LDS fragment per MFMA; 16-channel chunks keep two workgroups per CU.
// 3x3 stride-2 variants: the halo image alone is 88 KB -> one workgroup per CU, bounds say so.
template <int KS, int STRIDE, int TW, int TN, int CK, int MT, bool TF>
__global__ __launch_bounds__(256, (KS == 3 && STRIDE == 2) ? 1 : 2) void conv_fwd_bf16_kernel(const ConvBfArgs a) {
  constexpr int TPX = 128 * MT;
  using G = BfGeom<KS, STRIDE, TW, TPX>;
  constexpr int BF_PITCH = CK + 8, SEG = CK / 8, ROWS_IT = 256 / SEG;
  constexpr int NT = TN / 32;
  constexpr int IN_ROWS = G::HALO_H * G::HALO_W;
  constexpr int W_ROWS = G::TAPS * TN;
  constexpr int OUT_PITCH = TN + 8;   // bf16 per pixel row of the store-staging image (16-B aligned rows)
  constexpr int OUTF_PITCH = TN + 4;  // fp32 per pixel row of the gradient-join staging image
  constexpr int LDS_MAIN = (IN_ROWS + W_ROWS) * BF_PITCH, LDS_OUT = 256 * OUT_PITCH;
  constexpr int LDS_OUTF = 256 * OUTF_PITCH * 2;   // in bf16 elements
  static_assert(LDS_OUTF >= LDS_OUT, "fp32 staging is the larger image");
  __shared__ __attribute__((aligned(16))) __bf16 lds[LDS_MAIN > LDS_OUTF ? LDS_MAIN : LDS_OUTF];
  constexpr int TFC = (CK == 16 && MT == 2) ? BF_NARROW_CIN : BF_TF_MAXC;   // channels of the input-transform table
  __shared__ __attribute__((aligned(16))) float lds_tf[TF ? 2 * TFC : 4];
  __bf16* lds_in = lds;
  __bf16* lds_w = lds + IN_ROWS * BF_PITCH;
  if constexpr (TF) {
    for (int i = threadIdx.x; i < a.C0; i += 256) {
      lds_tf[i] = a.in_scale[i];
      lds_tf[TFC + i] = a.in_shift[i];
    }
  }
  const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int nt = wg % a.n_tiles, sp = wg / a.n_tiles;
  const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
  const int oy0 = ty * G::TH, ox0 = tx * TW, n0 = nt * TN;
  const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, r = lane & 31;

  // MFMA row tile (wave, mt) covers pixels [(mt/2)*256 + (wave*2 + mt%2)*32, +32): slices of 256 pixels for the epilogue
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int p = (mt >> 1) * 256 + (wave * 2 + (mt & 1)) * 32 + r;
    abase[mt] = ((p / TW) * G::LS * G::HALO_W + (p % TW) * G::LS) * BF_PITCH + h * 8;
  }
  const int bbase = r * BF_PITCH + h * 8;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][j][i] = 0.f;

  // staging bookkeeping: SEG x 16-byte segments (8 channels) per row
  constexpr int IN_TOTAL = IN_ROWS * SEG, IN_IT = (IN_TOTAL + 255) / 256;
  constexpr int W_TOTAL = W_ROWS * SEG, W_IT = (W_TOTAL + 255) / 256;
  const int q = tid % SEG, row0 = tid / SEG;
  const int Cin = a.C0 + a.C1;
  const int Hs0 = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws0 = a.mode0 ? (a.Win >> 1) : a.Win;
  int pidx0[IN_IT], pidx1[IN_IT];
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int pix = row0 + it * ROWS_IT;
    const int hy = pix / G::HALO_W, hx = pix - hy * G::HALO_W;
    const int iy = iy0 + hy * G::GS, ix = ix0 + hx * G::GS;
    const bool inb = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win && pix < IN_ROWS;
    const int sy = a.mode0 ? (iy >> 1) : iy, sx = a.mode0 ? (ix >> 1) : ix;
    bool ok0 = inb;
    if (a.mode0 == 2) ok0 = ok0 && (((iy | ix) & 1) == 0);   // zero-insertion (transposed conv)
    pidx0[it] = ok0 ? (b * Hs0 + sy) * Ws0 + sx : -1;
    pidx1[it] = inb ? (b * a.Hin + iy) * a.Win + ix : -1;
  }
  int woff[W_IT];
#pragma unroll
  for (int it = 0; it < W_IT; ++it) {
    const int row = row0 + it * ROWS_IT;  // tap*TN + n
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
          sh[k] = lds_tf[TFC + cc + k];
        }
      }
    }
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      const int pix = row0 + it * ROWS_IT;
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
      const int row = row0 + it * ROWS_IT;
      if (W_TOTAL % 256 == 0 || row < W_ROWS) *reinterpret_cast<f32x4*>(lds_w + row * BF_PITCH + 8 * q) = rw[it];
    }
  };

  issue_loads(0);
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    __syncthreads();
    write_lds(c0);
    __syncthreads();
    if (c0 + CK < Cin) issue_loads(c0 + CK);
#pragma unroll
    for (int tap = 0; tap < G::TAPS; ++tap) {
      const int kh = tap / KS, kw = tap % KS;
#pragma unroll
      for (int ks = 0; ks < CK / 16; ++ks) {
        if (c0 + 16 * ks >= Cin) continue;   // Cin = 16 (mod 32): the second k-step of the last chunk is all zero
        bf16x8 av[MT], bv[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          av[mt] = *reinterpret_cast<const bf16x8*>(lds_in + abase[mt] + (kh * G::HALO_W + kw) * BF_PITCH + ks * 16);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bv[j] = *reinterpret_cast<const bf16x8*>(lds_w + bbase + (tap * TN + 32 * j) * BF_PITCH + ks * 16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt], bv[j], acc[mt][j], 0, 0, 0);
      }
    }
  }
  // epilogue: D col = lane&31 (channel), row = (i&3) + 8*(i>>2) + 4*(lane>>5) (pixel)
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
  // Stores go through LDS: a lane owns ONE channel of 16 pixels (2-byte scattered accesses, 64 B runs); staged
  // pixel-major, every lane then moves 16 B and a wave covers whole NHWC pixel rows.  Plain stores stage the
  // rounded bf16 values; gradient joins (accumulate) stage the fp32 accumulators, add the bf16 value already in
  // memory and round ONCE.
  const bool second = a.cout_split > 0 && n0 >= a.cout_split;
  const int ld_all = a.cout_split > 0 ? (second ? a.Cout - a.cout_split : a.cout_split) : a.Cout;
  __bf16* outp = second ? a.out1 : a.out;
  const int nn0 = second ? n0 - a.cout_split : n0;
  const bool join = a.accumulate && !second;
  constexpr int SEGS = TN / 8, PER_IT = 256 / SEGS;
  const int seg = tid % SEGS, prow = tid / SEGS;
  const bool bnb = a.bnb.y != nullptr;   // uniform; join (accumulate) <=> mask from the stored activation a.bnb.act
  float q1[8], q2[8], b_mu[8], b_is[8], b_sc[8], b_sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) q1[k] = q2[k] = b_mu[k] = b_is[k] = b_sc[k] = b_sh[k] = 0.f;
  if (bnb && n0 + 8 * seg < a.Cout) {
    auto ld8 = [&](const float* p, float (&v)[8]) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(p + n0 + 8 * seg), hi = *reinterpret_cast<const f32x4*>(p + n0 + 8 * seg + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = lo[k];
        v[4 + k] = hi[k];
      }
    };
    ld8(a.bnb.mean, b_mu);
    ld8(a.bnb.invstd, b_is);
    if (a.bnb.act == nullptr) {
      ld8(a.bnb.act_scale, b_sc);
      ld8(a.bnb.act_shift, b_sh);
    }
  }
  // the tile is written out in slices of 256 pixels (MT / 2 of them) through the same staging image
#pragma unroll
  for (int sl = 0; sl < MT / 2; ++sl) {
    __syncthreads();   // every wave is done with the operand images / the previous slice
    const int pbase = sl * 256;
    if (!join) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int mrow = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int pl = (wave * 2 + m2) * 32 + mrow;
            const int oy = oy0 + (pbase + pl) / TW, ox = ox0 + (pbase + pl) % TW;
            const float v = acc[2 * sl + m2][j][i];
            if (n0 + 32 * j + r < a.Cout && oy < a.Ho && ox < a.Wo) {
              s1[j] += v;
              s2[j] += v * v;
            }
            lds[pl * OUT_PITCH + 32 * j + r] = (__bf16)v;
          }
      __syncthreads();
      if (n0 + 8 * seg < a.Cout) {
#pragma unroll
        for (int it = 0; it < SEGS; ++it) {
          const int pl = prow + it * PER_IT;
          const int oy = oy0 + (pbase + pl) / TW, ox = ox0 + (pbase + pl) % TW;
          if (oy < a.Ho && ox < a.Wo) {
            const size_t o = (((size_t)b * a.Ho + oy) * a.Wo + ox) * ld_all + nn0 + 8 * seg;
            const f32x4 raw = *reinterpret_cast<const f32x4*>(lds + pl * OUT_PITCH + 8 * seg);
            *reinterpret_cast<f32x4*>(outp + o) = raw;
            if (bnb) {   // (plain store: virtual activation, host check)
              // BatchNorm-backward partial sums from the ROUNDED gradient and the layer's raw output y, with the ReLU
              // mask the consumers saw (sign of bf16(y*sc+sh)) — the arithmetic of bn_bwd_reduce_bf16_kernel
              const f32x4 yraw = *reinterpret_cast<const f32x4*>(reinterpret_cast<const __bf16*>(a.bnb.y) + o);
              const bf16x8 gv = *reinterpret_cast<const bf16x8*>(&raw), yv = *reinterpret_cast<const bf16x8*>(&yraw);
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const float yk = (float)yv[k];
                const float act = (float)(__bf16)(yk * b_sc[k] + b_sh[k]);
                const float g = act > 0.f ? (float)gv[k] : 0.f;
                q1[k] += g;
                q2[k] += g * ((yk - b_mu[k]) * b_is[k]);
              }
            }
          }
        }
      }
    } else {
      float* ldsf = reinterpret_cast<float*>(lds);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int mrow = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int pl = (wave * 2 + m2) * 32 + mrow;
            ldsf[pl * OUTF_PITCH + 32 * j + r] = acc[2 * sl + m2][j][i];
          }
      __syncthreads();
      if (n0 + 8 * seg < a.Cout) {
        f32x4 prev[SEGS];
#pragma unroll
        for (int it = 0; it < SEGS; ++it) {
          const int pl = prow + it * PER_IT;
          const int oy = oy0 + (pbase + pl) / TW, ox = ox0 + (pbase + pl) % TW;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (oy < a.Ho && ox < a.Wo)
            v = *reinterpret_cast<const f32x4*>(outp + (((size_t)b * a.Ho + oy) * a.Wo + ox) * ld_all + nn0 + 8 * seg);
          prev[it] = v;
        }
#pragma unroll
        for (int it = 0; it < SEGS; ++it) {
          const int pl = prow + it * PER_IT;
          const int oy = oy0 + (pbase + pl) / TW, ox = ox0 + (pbase + pl) % TW;
          if (oy < a.Ho && ox < a.Wo) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(ldsf + pl * OUTF_PITCH + 8 * seg);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(ldsf + pl * OUTF_PITCH + 8 * seg + 4);
            bf16x8 o = *reinterpret_cast<bf16x8*>(&prev[it]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              o[k] = (__bf16)(lo[k] + (float)o[k]);
              o[4 + k] = (__bf16)(hi[k] + (float)o[4 + k]);
            }
            const size_t oo = (((size_t)b * a.Ho + oy) * a.Wo + ox) * ld_all + nn0 + 8 * seg;
            *reinterpret_cast<bf16x8*>(outp + oo) = o;
            if (bnb) {   // sums over the joined (rounded) gradient, mask from the stored activation
              const f32x4 yraw = *reinterpret_cast<const f32x4*>(reinterpret_cast<const __bf16*>(a.bnb.y) + oo);
              const f32x4 zraw = *reinterpret_cast<const f32x4*>(reinterpret_cast<const __bf16*>(a.bnb.act) + oo);
              const bf16x8 yv = *reinterpret_cast<const bf16x8*>(&yraw), zv = *reinterpret_cast<const bf16x8*>(&zraw);
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const float yk = (float)yv[k];
                const float g = (float)zv[k] > 0.f ? (float)o[k] : 0.f;
                q1[k] += g;
                q2[k] += g * ((yk - b_mu[k]) * b_is[k]);
              }
            }
          }
        }
      }
    }
  }
  if (a.stats != nullptr && bnb) {
    // per-thread sums over its pixels of 8 channels -> per-channel sums over the 256 / SEGS threads of a segment
    __syncthreads();
    float* qs = reinterpret_cast<float*>(lds);   // [2][8][256]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      qs[k * 256 + tid] = q1[k];
      qs[(8 + k) * 256 + tid] = q2[k];
    }
    __syncthreads();
    if (tid < 2 * TN) {
      const int which = tid / TN, c = tid % TN;
      if (n0 + c < a.Cout) {
        const float* col = qs + (which * 8 + (c & 7)) * 256 + (c >> 3);
        float sum = 0.f;
        for (int i = 0; i < PER_IT; ++i) sum += col[i * SEGS];   // fixed order
        a.stats[((size_t)which * a.P + sp) * a.Cout + n0 + c] = sum;
      }
    }
  } else if (a.stats != nullptr) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);  // [2][4 waves][TN]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
      const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (h == 0) {
        red[wave * TN + 32 * j + r] = t1;
        red[4 * TN + wave * TN + 32 * j + r] = t2;
      }
    }
    __syncthreads();
    if (tid < 2 * TN) {
      const int which = tid / TN, c = tid % TN;
      if (n0 + c < a.Cout) {
        const float* rr = red + which * 4 * TN + c;
        a.stats[((size_t)which * a.P + sp) * a.Cout + n0 + c] = (rr[0] + rr[TN]) + (rr[2 * TN] + rr[3 * TN]);
      }
    }
  }
}

static int bf_validate(const dt_conv_desc* d) {
  DT_REQUIRE(d != nullptr, "conv_bf16: null descriptor");
  DT_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->C0 > 0 && d->C1 >= 0 && d->Cout > 0, "conv_bf16: bad sizes");
  if (d->ksize == 4) {
    // the space-to-depth stem (dt_stem_s2d_bf16): 4x4 stride-1 window over 16 channels, 2 rows/cols of padding before
    // and 1 after (pad = 2), same-size output
    DT_REQUIRE(d->stride == 1 && d->pad == 2 && d->C0 == 16 && d->C1 == 0 && d->mode0 == 0 && (d->Cout % 64) == 0 &&
                   d->cout_split == 0 && d->accumulate == 0 && d->Ho == d->Hin && d->Wo == d->Win && d->Wo > 16,
               "conv_bf16: ksize 4 is the space-to-depth stem only (16 channels, pad 2, Cout %% 64 == 0)");
    return DT_OK;
  }
  DT_REQUIRE((d->ksize == 3 && (d->stride == 1 || d->stride == 2)) || (d->ksize == 1 && (d->stride == 2 || d->stride == 1)),
             "conv_bf16: ksize/stride (%d,%d) unsupported", d->ksize, d->stride);
  DT_REQUIRE((d->C0 & 7) == 0 && (d->C1 & 7) == 0 && (d->Cout & 7) == 0, "conv_bf16: channels must be multiples of 8");
  DT_REQUIRE(d->C1 == 0 || (d->C0 % BF_CK) == 0, "conv_bf16: concat needs C0 %% 32 == 0");
  DT_REQUIRE(d->mode0 >= 0 && d->mode0 <= 2, "conv_bf16: mode0 %d unsupported", d->mode0);
  DT_REQUIRE(d->mode0 == 0 || ((d->Hin & 1) == 0 && (d->Win & 1) == 0), "conv_bf16: mode0 needs even Hin/Win");
  DT_REQUIRE(d->cout_split == 0 || ((d->cout_split % 32) == 0 && d->cout_split < d->Cout),
             "conv_bf16: cout_split must be a multiple of 32 below Cout");
  const int ho = (d->Hin + 2 * d->pad - d->ksize) / d->stride + 1, wo = (d->Win + 2 * d->pad - d->ksize) / d->stride + 1;
  DT_REQUIRE(ho == d->Ho && wo == d->Wo, "conv_bf16: Ho/Wo mismatch");
  return DT_OK;
}

template <int KS, int STRIDE, int TW, int TN, int CK, int MT>
static int bf_launch(const ConvBfArgs& a, hipStream_t st) {
  const long grid = (long)a.P * a.n_tiles;
  if (a.in_scale != nullptr)
    hipLaunchKernelGGL((conv_fwd_bf16_kernel<KS, STRIDE, TW, TN, CK, MT, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_fwd_bf16_kernel<KS, STRIDE, TW, TN, CK, MT, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

template <int KS, int STRIDE>
static int bf_dispatch(const ConvBfArgs& a, int tw, int tn, int ck, int mt, hipStream_t st) {
  if constexpr (KS == 3 && STRIDE == 1) {
    if (mt == 4) return bf_launch<KS, STRIDE, 32, 64, 16, 4>(a, st);   // 512-pixel tiles: TW 32, TN 64, CK 16 only
    if (ck == 16) {
      if (tw == 32) return bf_launch<KS, STRIDE, 32, 32, 16, 2>(a, st);
      if (tw == 16) return bf_launch<KS, STRIDE, 16, 32, 16, 2>(a, st);
      return bf_launch<KS, STRIDE, 8, 32, 16, 2>(a, st);
    }
  }
  if (tn == 64) {
    if (tw == 32) return bf_launch<KS, STRIDE, 32, 64, 32, 2>(a, st);
    if (tw == 16) return bf_launch<KS, STRIDE, 16, 64, 32, 2>(a, st);
    return bf_launch<KS, STRIDE, 8, 64, 32, 2>(a, st);
  }
  if (tw == 32) return bf_launch<KS, STRIDE, 32, 32, 32, 2>(a, st);
  if (tw == 16) return bf_launch<KS, STRIDE, 16, 32, 32, 2>(a, st);
  return bf_launch<KS, STRIDE, 8, 32, 32, 2>(a, st);
}

static void bf_cfg(const dt_conv_desc* d, int* tw_, int* tn_, int* ck_, int* mt_) {
  if (d->ksize == 4) {   // space-to-depth stem: one variant
    *tw_ = 32; *tn_ = 64; *ck_ = 16; *mt_ = 2;
    return;
  }
  if (dt_conv_bf16_narrow_supported(d)) {   // conv_bf16_narrow.hip: Cin, Cout in {16, 32} at full resolution (mt = 16)
    *tw_ = 32; *tn_ = 16; *ck_ = 16; *mt_ = 16;
    return;
  }
  if (dt_conv_bf16_dma_supported(d)) {   // conv_bf16_dma.hip: 512-pixel x 64-channel tiles, 8 waves (reported as mt = 8)
    *tw_ = 32; *tn_ = 64; *ck_ = 32; *mt_ = 8;
    return;
  }
  const int tw = d->Wo > 16 ? 32 : (d->Wo > 8 ? 16 : 8);
  int tn = d->Cout >= 64 ? 64 : 32;
  if (d->cout_split > 0 && (d->cout_split % 64) != 0) tn = 32;
  if (tn == 64) {
    const long wgs = (long)d->B * dt_cdiv(d->Ho, 256 / tw) * dt_cdiv(d->Wo, tw) * dt_cdiv(d->Cout, 64);
    if (wgs < 512) tn = 32;
  }
  *tw_ = tw;
  *tn_ = tn;
  *ck_ = (tn == 32 && d->ksize == 3 && d->stride == 1 && d->C1 == 0 && d->C0 <= BF_NARROW_CIN) ? 16 : 32;
  *mt_ = 2;
  if (tn == 64 && tw == 32 && d->ksize == 3 && d->stride == 1 && (d->C0 % 16) == 0 && (d->C1 % 16) == 0 && d->Ho >= 16 &&
      d->Cout >= 128) {
    // 512-pixel tiles when they still fill the chip twice over.  Measured per layer (scripts/bench_conv.py, B=64):
    // +8 % on 128->128 @64x64, +5 % on 384->128 @64x64, nothing on 64-channel outputs (one n-tile) -> Cout >= 128 only
    const long wgs = (long)d->B * dt_cdiv(d->Ho, 16) * dt_cdiv(d->Wo, 32) * dt_cdiv(d->Cout, 64);
    if (wgs >= 1024) {
      *mt_ = 4;
      *ck_ = 16;
    }
  }
}

extern "C" int dt_conv2d_bf16_config(const dt_conv_desc* d, int* tw, int* tn, int* ck, int* mt) {
  if (bf_validate(d) != DT_OK) return DT_EINVAL;
  bf_cfg(d, tw, tn, ck, mt);
  return DT_OK;
}

extern "C" int dt_conv2d_bf16_stat_rows(const dt_conv_desc* d) {
  if (bf_validate(d) != DT_OK) return DT_EINVAL;
  int tw, tn, ck, mt;
  bf_cfg(d, &tw, &tn, &ck, &mt);
  if (mt == 8) return dt_conv_bf16_dma_stat_rows(d);
  if (mt == 16) return dt_conv_bf16_narrow_rows(d);
  return d->B * dt_cdiv(d->Ho, 128 * mt / tw) * dt_cdiv(d->Wo, tw);
}

static int conv2d_bf16_impl(const dt_conv_desc* d, const void* src0, const void* src1, const void* w_bf16, void* out,
                            void* out1, float* stats, const float* in_scale, const float* in_shift, void* stream,
                            const dt_bn_bwd_fuse* fuse);

extern "C" int dt_conv2d_bf16(const dt_conv_desc* d, const void* src0, const void* src1, const void* w_bf16,
                              void* out, void* out1, float* stats, const float* in_scale, const float* in_shift,
                              void* stream) {
  return conv2d_bf16_impl(d, src0, src1, w_bf16, out, out1, stats, in_scale, in_shift, stream, nullptr);
}

extern "C" int dt_conv2d_bf16_bn_bwd(const dt_conv_desc* d, const void* src0, const void* w_bf16, void* out, float* red,
                                     const dt_bn_bwd_fuse* fuse, void* stream) {
  DT_REQUIRE(d && fuse && red && fuse->y && fuse->mean && fuse->invstd, "conv_bf16_bn_bwd: null pointer");
  DT_REQUIRE(fuse->act != nullptr || (fuse->act_scale && fuse->act_shift),
             "conv_bf16_bn_bwd: give the stored activation or the scale/shift of a virtual one");
  DT_REQUIRE(d->ksize == 3 && d->stride == 1 && d->mode0 == 0 && d->C1 == 0 && d->cout_split == 0,
             "conv_bf16_bn_bwd: plain 3x3 stride-1 data gradients only");
  DT_REQUIRE((d->accumulate != 0) == (fuse->act != nullptr),
             "conv_bf16_bn_bwd: gradient joins (accumulate) go with a stored activation, plain stores with a virtual one");
  DT_REQUIRE((((uintptr_t)fuse->mean | (uintptr_t)fuse->invstd | (uintptr_t)fuse->act_scale |
               (uintptr_t)fuse->act_shift) & 15) == 0, "conv_bf16_bn_bwd: per-channel arrays must be 16-byte aligned");
  return conv2d_bf16_impl(d, src0, nullptr, w_bf16, out, nullptr, red, nullptr, nullptr, stream, fuse);
}

// data gradient of a convolution whose input was a nearest x2 up-sampling, for the narrow layers (dec4.conv1): `d`, src0 =
// dy and w_bf16 as for dt_conv2d_bf16_bn_bwd (the full-resolution data-gradient form); the 2x2 sums of the up-sampling's
// backward are taken on the accumulators and gx [B, Ho/2, Wo/2, Cout] is stored with the BatchNorm-backward sums of the
// layer below in red (P = dt_conv2d_bf16_stat_rows(d)) — replaces dt_conv2d_bf16 + dt_upsample2x_bwd_bn_bf16
extern "C" int dt_conv2d_bf16_upsampled_dgrad_supported(const dt_conv_desc* d) {
  static const int on = [] {
    const char* e = getenv("DT_BF16_FUSE_UPSAMPLE_BWD");
    return (e == nullptr || e[0] != '0') ? 1 : 0;
  }();
  return on && d != nullptr && bf_validate(d) == DT_OK && dt_conv_bf16_narrow_supported(d) && d->mode0 == 0 &&
         ((d->Ho | d->Wo) & 1) == 0;
}

extern "C" int dt_conv2d_bf16_upsampled_dgrad(const dt_conv_desc* d, const void* dy, const void* w_bf16, void* gx, float* red,
                                              const dt_bn_bwd_fuse* fuse, void* stream) {
  DT_REQUIRE(d && dy && w_bf16 && gx && red && fuse && fuse->y && fuse->mean && fuse->invstd && fuse->act_scale &&
                 fuse->act_shift && fuse->act == nullptr, "conv_bf16_upsampled_dgrad: null pointer / stored activation");
  DT_REQUIRE(dt_conv2d_bf16_upsampled_dgrad_supported(d), "conv_bf16_upsampled_dgrad: layer shape not supported");
  ConvBfArgs a;
  a.bnb = *fuse;
  a.out1 = nullptr; a.stats = red; a.cout_split = 0; a.accumulate = 0;
  a.src0 = (const __bf16*)dy; a.src1 = nullptr; a.w = (const __bf16*)w_bf16;
  a.in_scale = nullptr; a.in_shift = nullptr; a.out = (__bf16*)gx;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = 0; a.mode0 = 0;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.pad = d->pad;
  a.n_tiles = 1;
  return dt_conv_bf16_narrow_launch(d, a, (hipStream_t)stream, true);
}

static int conv2d_bf16_impl(const dt_conv_desc* d, const void* src0, const void* src1, const void* w_bf16, void* out,
                            void* out1, float* stats, const float* in_scale, const float* in_shift, void* stream,
                            const dt_bn_bwd_fuse* fuse) {
  int rc = bf_validate(d);
  if (rc != DT_OK) return rc;
  DT_REQUIRE(src0 && w_bf16 && out, "conv_bf16: null pointer");
  DT_REQUIRE(d->C1 == 0 || src1, "conv_bf16: src1 missing");
  DT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv_bf16: in_scale/in_shift must come together");
  DT_REQUIRE(in_scale == nullptr || d->C0 <= BF_TF_MAXC, "conv_bf16: input transform needs C0 <= %d", BF_TF_MAXC);
  DT_REQUIRE(d->cout_split == 0 || out1, "conv_bf16: out1 missing");
  int tw, tn, ck, mt;
  bf_cfg(d, &tw, &tn, &ck, &mt);
  ConvBfArgs a;
  a.bnb = fuse ? *fuse : dt_bn_bwd_fuse{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a.out1 = (__bf16*)out1; a.stats = stats; a.cout_split = d->cout_split; a.accumulate = d->accumulate;
  a.src0 = (const __bf16*)src0; a.src1 = (const __bf16*)src1; a.w = (const __bf16*)w_bf16;
  a.in_scale = in_scale; a.in_shift = in_shift; a.out = (__bf16*)out;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.pad = d->pad;
  a.tiles_x = dt_cdiv(d->Wo, tw); a.tiles_y = dt_cdiv(d->Ho, 128 * mt / tw); a.n_tiles = dt_cdiv(d->Cout, tn);
  a.P = d->B * a.tiles_x * a.tiles_y;
  hipStream_t st = (hipStream_t)stream;
  if (mt == 8) return dt_conv_bf16_dma_launch(a, st);
  if (mt == 16) return dt_conv_bf16_narrow_launch(d, a, st);   // (accumulate == 0 there: no stored-activation joins)
  if (d->ksize == 4) {
    DT_REQUIRE(in_scale == nullptr && fuse == nullptr, "conv_bf16: the stem takes no input transform / fused reduction");
    return bf_launch<4, 1, 32, 64, 16, 2>(a, st);
  }
  if (d->ksize == 3 && d->stride == 1) return bf_dispatch<3, 1>(a, tw, tn, ck, mt, st);
  if (d->ksize == 3 && d->stride == 2) return bf_dispatch<3, 2>(a, tw, tn, ck, mt, st);
  if (d->ksize == 1 && d->stride == 1) return bf_dispatch<1, 1>(a, tw, tn, ck, mt, st);
  return bf_dispatch<1, 2>(a, tw, tn, ck, mt, st);
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

// data-gradient weights: the dgrad conv has Cout_d = Cin, Cin_d = Cout, so its [tap'][Cout_d][Cin_d] image is the
// HWIO tensor itself with the taps reversed — one elementwise fp32 -> bf16 pass
__global__ void pack_dgrad_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int taps,
                                               int64_t per_tap) {
  const int64_t total = (int64_t)taps * per_tap;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t tap = i / per_tap, e = i - tap * per_tap;
    out[i] = (__bf16)w[(taps - 1 - tap) * per_tap + e];
  }
}

extern "C" int dt_pack_dgrad_weights_bf16(const float* w_hwio, void* out, int ksize, int Cin, int Cout, void* stream) {
  DT_REQUIRE(w_hwio && out && ksize > 0 && Cin > 0 && Cout > 0, "pack_dgrad_weights_bf16: bad args");
  const int64_t per_tap = (int64_t)Cin * Cout;
  int64_t g = (per_tap * ksize * ksize + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_dgrad_weights_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_hwio,
                     (__bf16*)out, ksize * ksize, per_tap);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ the 7x7 / stride-2 stem on the bf16 kernels
// A 7x7 stride-2 window over C <= 4 channels is a 4x4 stride-1 window over the 2x2 space-to-depth image with
// 4*4 = 16 channels: input row 2u + a (a in {0,1}) of window row kh = 2*ku + a - 1, ku = 0..3 (ku = 0, a = 0 falls
// outside the 7 taps: zero weight).  dt_stem_s2d_bf16 builds that image ([B,H/2,W/2,16] bf16, channel (a*2+b)*4 + c),
// dt_stem_pack_weights_bf16 the matching [16 taps][Cout][16] weights; dt_conv2d_bf16 with ksize = 4, pad = 2 then
// computes the stem with v_mfma_f32_32x32x16_bf16 (K = 256, 147 of them real) instead of fp32 MFMA.
__global__ __launch_bounds__(256) void stem_s2d_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ out, int B,
                                                            int H, int W, int Cin) {
  const int H2 = H >> 1, W2 = W >> 1;
  const int64_t total = (int64_t)B * H2 * W2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % W2);
    const int u = (int)((i / W2) % H2);
    const int b = (int)(i / ((int64_t)W2 * H2));
    bf16x8 lo, hi;
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const float* px = x + (((size_t)b * H + 2 * u + (ab >> 1)) * W + 2 * v + (ab & 1)) * Cin;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const __bf16 val = c < Cin ? (__bf16)px[c] : (__bf16)0.f;
        if (ab < 2) lo[(ab & 1) * 4 + c] = val; else hi[(ab & 1) * 4 + c] = val;
      }
    }
    reinterpret_cast<bf16x8*>(out)[2 * i] = lo;
    reinterpret_cast<bf16x8*>(out)[2 * i + 1] = hi;
  }
}

extern "C" int dt_stem_s2d_bf16(const float* x_nhwc, void* out, int B, int H, int W, int Cin, void* stream) {
  DT_REQUIRE(x_nhwc && out && B > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0 && Cin >= 1 && Cin <= 4,
             "stem_s2d_bf16: needs even H, W and 1..4 channels");
  int64_t g = ((int64_t)B * (H / 2) * (W / 2) + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(stem_s2d_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x_nhwc, (__bf16*)out, B,
                     H, W, Cin);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

__global__ void stem_pack_weights_bf16_kernel(const float* __restrict__ w7, __bf16* __restrict__ out, int Cin, int Cout) {
  // out[(ku*4 + kv)][co][(a*2+b)*4 + c] = w7[2ku+a-1][2kv+b-1][c][co]
  const int total = 16 * Cout * 16;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ch = i & 15, co = (i >> 4) % Cout, tap = i / (16 * Cout);
    const int ku = tap >> 2, kv = tap & 3, a = ch >> 3, b = (ch >> 2) & 1, c = ch & 3;
    const int kh = 2 * ku + a - 1, kw = 2 * kv + b - 1;
    float v = 0.f;
    if (kh >= 0 && kh < 7 && kw >= 0 && kw < 7 && c < Cin) v = w7[((size_t)(kh * 7 + kw) * Cin + c) * Cout + co];
    out[i] = (__bf16)v;
  }
}

extern "C" int dt_stem_pack_weights_bf16(const float* w_hwio, void* out, int Cin, int Cout, void* stream) {
  DT_REQUIRE(w_hwio && out && Cin >= 1 && Cin <= 4 && Cout > 0, "stem_pack_weights_bf16: bad args");
  hipLaunchKernelGGL(stem_pack_weights_bf16_kernel, dim3(dt_cdiv(16 * Cout * 16, 256)), dim3(256), 0, (hipStream_t)stream,
                     w_hwio, (__bf16*)out, Cin, Cout);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// weight gradient of the space-to-depth stem back to the 7x7 HWIO layout:
// dw7[kh][kw][c][co] = dw4[(ku*4 + kv)][(a*2+b)*4 + c][co] with kh = 2ku + a - 1, kw = 2kv + b - 1
__global__ void stem_unpack_wgrad_kernel(const float* __restrict__ dw4, float* __restrict__ dw7, int Cin, int Cout) {
  const int total = 49 * Cin * Cout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int co = i % Cout, c = (i / Cout) % Cin, tap = i / (Cout * Cin);
    const int kh = tap / 7, kw = tap % 7;
    const int ku = (kh + 1) >> 1, a = (kh + 1) & 1, kv = (kw + 1) >> 1, b = (kw + 1) & 1;
    dw7[i] = dw4[((size_t)(ku * 4 + kv) * 16 + (a * 2 + b) * 4 + c) * Cout + co];
  }
}

extern "C" int dt_stem_unpack_wgrad(const float* dw4, float* dw_hwio_7x7, int Cin, int Cout, void* stream) {
  DT_REQUIRE(dw4 && dw_hwio_7x7 && Cin >= 1 && Cin <= 4 && Cout > 0, "stem_unpack_wgrad: bad args");
  hipLaunchKernelGGL(stem_unpack_wgrad_kernel, dim3(dt_cdiv(49 * Cin * Cout, 256)), dim3(256), 0, (hipStream_t)stream,
                     dw4, dw_hwio_7x7, Cin, Cout);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ all weight images of a network in ONE launch
// The per-layer pack / flip kernels above are 4-5 us each and there are ~45 layers: 90-135 launches per step.
// Table-driven variant over the flat parameter buffer: row l = (w_off, taps, Cin, Cout, first_tile); a workgroup
// owns one 32x32 (ci, co) tile of one tap.  mode 0: fp32 [tap'][co][ci] with reversed taps (dt_weight_flip_transpose),
// 1: bf16 [tap][co][ci] (dt_pack_weights_bf16), 2: bf16 HWIO with reversed taps (dt_pack_dgrad_weights_bf16).
// Images are written at the layer's own offset w_off of the output buffer.
__global__ __launch_bounds__(256) void weight_images_kernel(const float* __restrict__ params, void* __restrict__ out,
                                                            const int32_t* __restrict__ table, int n_layers, int mode) {
  __shared__ float tile[32][33];
  __shared__ int32_t row[5];
  if (threadIdx.x == 0) {
    int l = 0;
    while (l + 1 < n_layers && table[(l + 1) * 5 + 4] <= (int)blockIdx.x) ++l;   // <= 64 layers: linear scan
#pragma unroll
    for (int k = 0; k < 5; ++k) row[k] = table[l * 5 + k];
  }
  __syncthreads();
  const int w_off = row[0], taps = row[1], Cin = row[2], Cout = row[3];
  const int t = (int)blockIdx.x - row[4];
  const int cob = (Cout + 31) / 32, cib = (Cin + 31) / 32;
  const int tap = t / (cib * cob), ci0 = ((t / cob) % cib) * 32, co0 = (t % cob) * 32;
  const bool flip = mode == 0 || mode == 2 || mode == 4, transpose = mode == 0 || mode == 1 || mode == 3;
  const float* w = params + w_off + (size_t)(flip ? taps - 1 - tap : tap) * Cin * Cout;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int ci = ci0 + i, co = co0 + tx;
    tile[i][tx] = (ci < Cin && co < Cout) ? w[(size_t)ci * Cout + co] : 0.f;
  }
  __syncthreads();
  const size_t obase = (size_t)w_off + (size_t)tap * Cin * Cout;
  if (mode >= 3) {
    // chunked images for the LDS-DMA kernels (conv_bf16_dma.hip): [tap][K/32][N][32] — the 64 output rows x 32 input
    // channels one workgroup stages per (tap, chunk) are ONE contiguous 4 KiB block, so every 1 KiB DMA piece reads
    // whole cache lines (the [tap][N][K] image gives 64-byte fragments of 16 different lines per piece)
    if ((Cin & 31) || (Cout & 31)) return;
    for (int i = ty; i < 32; i += 8) {
      if (mode == 3)   // forward: N = Cout, K = Cin; row co0 + i holds the 32 input channels ci0 .. ci0 + 31
        reinterpret_cast<__bf16*>(out)[obase + ((size_t)(ci0 >> 5) * Cout + co0 + i) * 32 + tx] = (__bf16)tile[tx][i];
      else             // data gradient: N = Cin, K = Cout; row ci0 + i holds the 32 channels co0 .. co0 + 31
        reinterpret_cast<__bf16*>(out)[obase + ((size_t)(co0 >> 5) * Cin + ci0 + i) * 32 + tx] = (__bf16)tile[i][tx];
    }
    return;
  }
  for (int i = ty; i < 32; i += 8) {
    if (transpose) {
      const int co = co0 + i, ci = ci0 + tx;
      if (ci < Cin && co < Cout) {
        const size_t o = obase + (size_t)co * Cin + ci;
        if (mode == 0) reinterpret_cast<float*>(out)[o] = tile[tx][i];
        else reinterpret_cast<__bf16*>(out)[o] = (__bf16)tile[tx][i];
      }
    } else {
      const int ci = ci0 + i, co = co0 + tx;
      if (ci < Cin && co < Cout) reinterpret_cast<__bf16*>(out)[obase + (size_t)ci * Cout + co] = (__bf16)tile[i][tx];
    }
  }
}

// the four bf16 images a bf16 TRAINING step needs (modes 1, 2, 3, 4 of weight_images_kernel) from ONE read of the
// parameters: four launches re-read the 98 MB of fp32 weights four times (4 x 74 us per step at 2 TB/s)
__global__ __launch_bounds__(256) void weight_images_bf16_all_kernel(const float* __restrict__ params, __bf16* __restrict__ o_fwd,
                                                                     __bf16* __restrict__ o_dgrad, __bf16* __restrict__ o_fwd_c,
                                                                     __bf16* __restrict__ o_dgrad_c,
                                                                     const int32_t* __restrict__ table, int n_layers) {
  __shared__ float tile[32][33];
  __shared__ int32_t row[5];
  if (threadIdx.x == 0) {
    int l = 0;
    while (l + 1 < n_layers && table[(l + 1) * 5 + 4] <= (int)blockIdx.x) ++l;
#pragma unroll
    for (int k = 0; k < 5; ++k) row[k] = table[l * 5 + k];
  }
  __syncthreads();
  const int w_off = row[0], taps = row[1], Cin = row[2], Cout = row[3];
  const int t = (int)blockIdx.x - row[4];
  const int cob = (Cout + 31) / 32, cib = (Cin + 31) / 32;
  const int tap = t / (cib * cob), ci0 = ((t / cob) % cib) * 32, co0 = (t % cob) * 32;
  const float* w = params + w_off + (size_t)tap * Cin * Cout;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int ci = ci0 + i, co = co0 + tx;
    tile[i][tx] = (ci < Cin && co < Cout) ? w[(size_t)ci * Cout + co] : 0.f;
  }
  __syncthreads();
  const size_t ob = (size_t)w_off + (size_t)tap * Cin * Cout;                  // images in the tap order of the source
  const size_t obf = (size_t)w_off + (size_t)(taps - 1 - tap) * Cin * Cout;    // data-gradient images: taps reversed
  const bool chunked = ((Cin | Cout) & 31) == 0;
  for (int i = ty; i < 32; i += 8) {
    {   // [tap][co][ci]
      const int co = co0 + i, ci = ci0 + tx;
      if (ci < Cin && co < Cout) o_fwd[ob + (size_t)co * Cin + ci] = (__bf16)tile[tx][i];
    }
    {   // HWIO, taps reversed
      const int ci = ci0 + i, co = co0 + tx;
      if (ci < Cin && co < Cout) o_dgrad[obf + (size_t)ci * Cout + co] = (__bf16)tile[i][tx];
    }
    if (chunked) {   // [tap][K/32][N][32] for the LDS-DMA kernels
      o_fwd_c[ob + ((size_t)(ci0 >> 5) * Cout + co0 + i) * 32 + tx] = (__bf16)tile[tx][i];
      o_dgrad_c[obf + ((size_t)(co0 >> 5) * Cin + ci0 + i) * 32 + tx] = (__bf16)tile[i][tx];
    }
  }
}

extern "C" int dt_weight_images_bf16_all(const float* params, void* fwd, void* dgrad, void* fwd_chunked, void* dgrad_chunked,
                                         const int32_t* table, int n_layers, int total_tiles, void* stream) {
  DT_REQUIRE(params && fwd && dgrad && fwd_chunked && dgrad_chunked && table && n_layers > 0 && total_tiles > 0,
             "weight_images_bf16_all: bad args");
  hipLaunchKernelGGL(weight_images_bf16_all_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, params,
                     (__bf16*)fwd, (__bf16*)dgrad, (__bf16*)fwd_chunked, (__bf16*)dgrad_chunked, table, n_layers);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_weight_images(const float* params, void* out, const int32_t* table, int n_layers, int total_tiles,
                                int mode, void* stream) {
  DT_REQUIRE(params && out && table && n_layers > 0 && total_tiles > 0 && mode >= 0 && mode <= 4,
             "weight_images: bad args");
  hipLaunchKernelGGL(weight_images_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, params, out,
                     table, n_layers, mode);
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
  // grid stride is a multiple of C8 (host check): one channel group per thread, coefficients loaded once
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(i0 % C8) * 8;
  float sc[8], sf[8], rsc[8], rsf[8];
  auto ldc8 = [](const float* __restrict__ p, int cc, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + cc), b = *reinterpret_cast<const f32x4*>(p + cc + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = a[k];
      v[4 + k] = b[k];
    }
  };
  ldc8(scale, c, sc);
  ldc8(shift, c, sf);
  if (rscale) {
    ldc8(rscale, c, rsc);
    ldc8(rshift, c, rsf);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      rsc[k] = 1.f;
      rsf[k] = 0.f;
    }
  }
  for (int64_t i = i0; i < n8; i += stride) {
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
      float f = v[k] * sc[k] + sf[k];
      if (relu == 2) f = f < 0.f ? 0.f : f;      // ReLU on the main branch only (ResUnet decoder: relu(bn2(.)) + identity_conv)
      if (res) {
        float rr = (float)rv[k];
        if (rscale) rr = rr * rsc[k] + rsf[k];
        f += rr;
      }
      if (relu == 1) f = f < 0.f ? 0.f : f;
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
  DT_REQUIRE(256 % (C / 8) == 0, "bn_act_bf16: C/8 must divide 256 (C=%d)", C);
  DT_REQUIRE((((uintptr_t)scale | (uintptr_t)shift | (uintptr_t)rscale | (uintptr_t)rshift) & 15) == 0,
             "bn_act_bf16: per-channel arrays must be 16-byte aligned");
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

// ------------------------------------------------------------------ weight gradient, bf16 operands
// dW[tap][ci][co] (fp32) = sum_pixels x[pix+tap][ci] * dy[pix][co] on v_mfma_f32_32x32x16_bf16 with K = 16 pixels.
// Both operands need 8 consecutive PIXELS per lane for a fixed channel, i.e. the transpose of the NHWC tile:
// the tiles are staged pixel-major (coalesced from HBM) and read with ds_read_b64_tr_b16, which hands every
// 16-lane group a 4-pixel x 16-channel block column-major (cdna_hip_programming.md T10) — no transposed copy.
// Tap shifts move whole pixel rows of the LDS image, so every transposed read stays 8-byte aligned.
struct WgradBfArgs {
  const __bf16* src0;
  const __bf16* src1;
  const float* in_scale;
  const float* in_shift;
  const __bf16* dy;
  float* ws;  // [parts][taps][Cin][Cout] fp32 slabs
  int B, Hin, Win, C0, C1, mode0, Ho, Wo, Cout, pad, tiles_x, tiles_y, T, ci_blocks, co_blocks, ksplit;
};

#define WB_PITCH 72   // bf16 elements per LDS row (64 used): 144 bytes (multiple of 8 for the transposed reads)
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

// BLK = 64: a workgroup owns a 64ci x 64co block, its 4 waves split the channels 2 x 2.
// BLK = 32 (layers with Cin, Cout <= 32 — the full-resolution decoder end, HBM-bound in bf16): one 32 x 32 block,
// the 4 waves split the PIXEL rows of each tile instead and are summed through LDS once at the end, so no wave
// multiplies zero padding.
template <int KS, int STRIDE, int TW, bool TF, int BLK>
__global__ __launch_bounds__(256, (KS == 3 && STRIDE == 2) ? 1 : 2) void conv_wgrad_bf16_kernel(const WgradBfArgs a) {
  constexpr int TPX = 128, TH = TPX / TW;
  constexpr int LS = (KS == 1) ? 1 : STRIDE, GS = (KS == 1) ? STRIDE : 1;
  constexpr int HALO_H = (TH - 1) * LS + KS, HALO_W = (TW - 1) * LS + KS;
  constexpr int TAPS = KS * KS;
  // KS = 4 (the space-to-depth stem): 16 taps would need 256 accumulator registers, so two workgroups share a
  // (ci, co) block and take 8 taps (two window rows) each
  constexpr int TG = (KS == 4) ? 2 : 1, NTAP = TAPS / TG;
  constexpr int X_ROWS = HALO_H * HALO_W;
  __shared__ __attribute__((aligned(16))) __bf16 lds[(X_ROWS + TPX) * WB_PITCH];
  __shared__ __attribute__((aligned(16))) float lds_tf[TF ? 128 : 4];
  __bf16* lx = lds;
  __bf16* ly = lds + X_ROWS * WB_PITCH;
  const int wgid = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int nblk = a.ci_blocks * a.co_blocks;
  const int blk = wgid % nblk, tg = (wgid / nblk) % TG, ks = wgid / (nblk * TG);
  const int kh0 = tg * (NTAP / KS);    // first window row of this workgroup's tap group
  const int ci0 = (blk / a.co_blocks) * BLK, co0 = (blk % a.co_blocks) * BLK;
  const int Cin = a.C0 + a.C1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wci = BLK == 64 ? (wave >> 1) : 0, wco = BLK == 64 ? (wave & 1) : 0;
  // transposed-read lane roles: group g = lane>>4 -> channel half (g&1), pixel half h = lane>>5; q,p inside the group
  const int h = lane >> 5, gsel = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int xlane = (8 * h + q4) * LS * WB_PITCH + wci * 32 + 16 * gsel + 4 * p4;
  const int ylane = (8 * h + q4) * WB_PITCH + wco * 32 + 16 * gsel + 4 * p4;
  if constexpr (TF) {
    if (tid < BLK && ci0 + tid < a.C0) {
      lds_tf[tid] = a.in_scale[ci0 + tid];
      lds_tf[64 + tid] = a.in_shift[ci0 + tid];
    }
  }
  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // staging: BLK/8 x 16-byte segments (8 channels) per pixel row
  constexpr int SEGS = BLK / 8, ROWS_IT = 256 / SEGS;
  constexpr int X_TOTAL = X_ROWS * SEGS, X_IT = (X_TOTAL + 255) / 256;
  constexpr int Y_TOTAL = TPX * SEGS, Y_IT = (Y_TOTAL + 255) / 256;
  const int q8 = tid % SEGS, prow0 = tid / SEGS;
  const int cx = ci0 + 8 * q8;
  const bool x_use0 = cx < a.C0;
  const __bf16* xsrc = x_use0 ? a.src0 : a.src1;
  const int xC = x_use0 ? a.C0 : a.C1, xcc = x_use0 ? cx : cx - a.C0, xmode = x_use0 ? a.mode0 : 0;
  const int xHs = xmode ? (a.Hin >> 1) : a.Hin, xWs = xmode ? (a.Win >> 1) : a.Win;
  const bool x_ch_ok = cx < Cin;
  const bool x_tf = TF && x_use0 && x_ch_ok;
  const int cy = co0 + 8 * q8;
  const bool y_ch_ok = cy < a.Cout;

  for (int tile = ks; tile < a.T; tile += a.ksplit) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
    f32x4 rx[X_IT], ry[Y_IT];
    unsigned xvalid = 0;
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int pix = prow0 + it * ROWS_IT;
      const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
      const int iy = iy0 + hy * GS, ix = ix0 + hx * GS;
      bool ok = x_ch_ok && pix < X_ROWS && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      if (xmode == 2) ok = ok && (((iy | ix) & 1) == 0);
      const int sy = xmode ? (iy >> 1) : iy, sx = xmode ? (ix >> 1) : ix;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(xsrc + (((size_t)b * xHs + sy) * xWs + sx) * xC + xcc);
      xvalid |= (ok ? 1u : 0u) << it;
      rx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int pix = prow0 + it * ROWS_IT;
      const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (y_ch_ok && pix < TPX && oy < a.Ho && ox < a.Wo)
        v = *reinterpret_cast<const f32x4*>(a.dy + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + cy);
      ry[it] = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int pix = prow0 + it * ROWS_IT;
      if (pix < X_ROWS) {
        f32x4 raw = rx[it];
        if (TF && x_tf && ((xvalid >> it) & 1u)) {
          bf16x8 v = *reinterpret_cast<bf16x8*>(&raw);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float f = (float)v[k] * lds_tf[8 * q8 + k] + lds_tf[64 + 8 * q8 + k];
            f = f < 0.f ? 0.f : f;
            v[k] = (__bf16)f;
          }
          raw = *reinterpret_cast<f32x4*>(&v);
        }
        *reinterpret_cast<f32x4*>(lx + pix * WB_PITCH + 8 * q8) = raw;
      }
    }
#pragma unroll
    for (int it = 0; it < Y_IT; ++it) {
      const int pix = prow0 + it * ROWS_IT;
      if (pix < TPX) *reinterpret_cast<f32x4*>(ly + pix * WB_PITCH + 8 * q8) = ry[it];
    }
    __syncthreads();
    typedef bf16x4v __attribute__((address_space(3))) * lds_ptr;
    constexpr int ROWS_W = BLK == 64 ? TH : TH / 4;   // BLK 32: this wave's share of the tile's pixel rows
    const int row_w0 = BLK == 64 ? 0 : wave * ROWS_W;
#pragma unroll
    for (int rr = 0; rr < ROWS_W; ++rr) {
      const int row = row_w0 + rr;
#pragma unroll
      for (int xs = 0; xs < TW / 16; ++xs) {
        // B fragment: dy pixels (row, 16*xs + 8h .. +7), channels of this wave's co tile
        bf16x8 bv;
        {
          const int e = ylane + (row * TW + 16 * xs) * WB_PITCH;
          const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(ly + e));
          const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(ly + e + 4 * WB_PITCH));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            bv[k] = lo[k];
            bv[4 + k] = hi[k];
          }
        }
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          const int kh = kh0 + t / KS, kw = t % KS;
          const int e = xlane + ((row * LS + kh) * HALO_W + 16 * xs * LS + kw) * WB_PITCH;
          const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(lx + e));
          const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(lx + e + 4 * LS * WB_PITCH));
          bf16x8 av;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            av[k] = lo[k];
            av[4 + k] = hi[k];
          }
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }
  if constexpr (BLK == 32) {
    // sum the four pixel shares in wave 0 (fixed order 1, 2, 3 -> deterministic)
    float* red = reinterpret_cast<float*>(lds);   // [NTAP][16][64] fp32 <= 36 KB, inside the operand images
    static_assert((size_t)NTAP * 16 * 64 * sizeof(float) <= sizeof(lds), "reduction image must fit the tiles");
    for (int w = 1; w < 4; ++w) {
      __syncthreads();
      if (wave == w) {
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) red[(t * 16 + i) * 64 + lane] = acc[t][i];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[t][i] += red[(t * 16 + i) * 64 + lane];
      }
    }
    if (wave != 0) return;
  }
  const int part = ks;
  const int r = lane & 31;
  const int co = co0 + wco * 32 + r;
  if (co < a.Cout) {
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = ci0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (ci < Cin) a.ws[(((size_t)part * TAPS + tg * NTAP + t) * Cin + ci) * a.Cout + co] = acc[t][i];
      }
  }
}

static int wb_cfg(const dt_conv_desc* d, int* tw, int* ksplit, int* T, int* cib, int* cob) {
  const int Cin = d->C0 + d->C1;
  *tw = d->Wo > 16 ? 32 : 16;
  const bool stem = d->ksize == 4;
  const int blk = ((Cin <= 32 && d->Cout <= 32 && d->ksize == 3 && d->stride == 1) || stem) ? 32 : 64;
  *cib = dt_cdiv(Cin, blk);
  *cob = dt_cdiv(d->Cout, blk);
  const int th = 128 / *tw;
  *T = d->B * dt_cdiv(d->Ho, th) * dt_cdiv(d->Wo, *tw);
  int ks = 512 / (*cib * *cob * (stem ? 2 : 1));
  if (ks < 1) ks = 1;
  if (ks > *T) ks = *T;
  *ksplit = ks;
  return DT_OK;
}

// DT_BF16_WGRAD_DMA=0 keeps every layer on the register-staged kernel below (A/B switch; read once)
static bool wb_use_dma() {
  static const int on = [] {
    const char* e = getenv("DT_BF16_WGRAD_DMA");
    return (e == nullptr || e[0] != '0') ? 1 : 0;
  }();
  return on != 0;
}

static int wb_validate(const dt_conv_desc* d) {
  DT_REQUIRE(d != nullptr, "wgrad_bf16: null descriptor");
  if (d->ksize == 4) {   // the space-to-depth stem (see dt_stem_s2d_bf16)
    DT_REQUIRE(d->stride == 1 && d->pad == 2 && d->C0 == 16 && d->C1 == 0 && d->mode0 == 0 && (d->Cout % 32) == 0 &&
                   d->Ho == d->Hin && d->Wo == d->Win && d->Wo > 16,
               "wgrad_bf16: ksize 4 is the space-to-depth stem only");
    return DT_OK;
  }
  DT_REQUIRE((d->ksize == 3 || d->ksize == 1) && (d->stride == 1 || d->stride == 2),
             "wgrad_bf16: ksize/stride (%d,%d) unsupported", d->ksize, d->stride);
  DT_REQUIRE((d->C0 & 7) == 0 && (d->C1 & 7) == 0 && (d->Cout & 7) == 0, "wgrad_bf16: channels must be multiples of 8");
  DT_REQUIRE(d->mode0 >= 0 && d->mode0 <= 1, "wgrad_bf16: mode0");
  const int ho = (d->Hin + 2 * d->pad - d->ksize) / d->stride + 1, wo = (d->Win + 2 * d->pad - d->ksize) / d->stride + 1;
  DT_REQUIRE(ho == d->Ho && wo == d->Wo, "wgrad_bf16: Ho/Wo mismatch");
  return DT_OK;
}

extern "C" size_t dt_conv2d_wgrad_bf16_workspace(const dt_conv_desc* d) {
  if (wb_validate(d) != DT_OK) return 0;
  int tw, ks, T, cib, cob;
  wb_cfg(d, &tw, &ks, &T, &cib, &cob);
  const size_t E = (size_t)d->ksize * d->ksize * (d->C0 + d->C1) * d->Cout;
  size_t need = (size_t)ks * E * sizeof(float);
  if (dt_wgrad_bf16_narrow_supported(d)) {
    const size_t n3 = dt_wgrad_bf16_narrow_workspace(d);
    if (n3 > need) need = n3;
  }
  if (wb_use_dma() && dt_wgrad_bf16_dma_supported(d)) {   // the larger of the two: in_scale decides the kernel at launch
    const size_t n2 = dt_wgrad_bf16_dma_workspace(d);
    if (n2 > need) need = n2;
  }
  return need;
}

// dw[e] = sum_p ws[p][e] in ONE launch whatever the number of split-K slabs: a workgroup owns 64 consecutive elements
// (16 float4 quads) x 16 slab-lanes; every lane sums its slabs p = lane, lane + 16, ... in fp32 (4 independent
// accumulators: memory-level parallelism), the 16 lanes are combined in fp64 in a fixed order -> deterministic
__global__ __launch_bounds__(256) void wgrad_bf16_final_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                               int parts, int64_t E) {
  __shared__ f32x4 sh[256];
  const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t E4 = E >> 2, e4 = (int64_t)blockIdx.x * 16 + q;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (e4 < E4) {
    const f32x4* src = reinterpret_cast<const f32x4*>(ws) + e4;
    int p = rl;
    for (; p + 48 < parts; p += 64) {
      const f32x4 v0 = src[(size_t)p * E4], v1 = src[(size_t)(p + 16) * E4];
      const f32x4 v2 = src[(size_t)(p + 32) * E4], v3 = src[(size_t)(p + 48) * E4];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; p < parts; p += 16) a0 += src[(size_t)p * E4];
  }
  sh[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && e4 < E4) {
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < 16; ++i) {
      const f32x4 v = sh[i * 16 + q];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] += (double)v[k];
    }
    reinterpret_cast<f32x4*>(dw)[e4] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
  }
}

template <int KS, int STRIDE, int TW>
static int wb_launch(const WgradBfArgs& a, int grid, hipStream_t st) {
  const bool narrow = a.C0 + a.C1 <= 32 && a.Cout <= 32;
  if (narrow) {
    if constexpr (KS == 3 && STRIDE == 1) {
      if (a.in_scale != nullptr)
        hipLaunchKernelGGL((conv_wgrad_bf16_kernel<KS, STRIDE, TW, true, 32>), dim3(grid), dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((conv_wgrad_bf16_kernel<KS, STRIDE, TW, false, 32>), dim3(grid), dim3(256), 0, st, a);
      DT_LAUNCH_CHECK();
      return DT_OK;
    }
  }
  if (a.in_scale != nullptr)
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<KS, STRIDE, TW, true, 64>), dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<KS, STRIDE, TW, false, 64>), dim3(grid), dim3(256), 0, st, a);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_conv2d_wgrad_bf16(const dt_conv_desc* d, const void* src0, const void* src1, const void* dy,
                                    float* dw_hwio, float* workspace, size_t workspace_bytes, const float* in_scale,
                                    const float* in_shift, void* stream) {
  int rc = wb_validate(d);
  if (rc != DT_OK) return rc;
  DT_REQUIRE(src0 && dy && dw_hwio && workspace, "wgrad_bf16: null pointer");
  DT_REQUIRE(d->C1 == 0 || src1, "wgrad_bf16: src1 missing");
  DT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "wgrad_bf16: in_scale/in_shift must come together");
  DT_REQUIRE(workspace_bytes >= dt_conv2d_wgrad_bf16_workspace(d), "wgrad_bf16: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t E = (int64_t)d->ksize * d->ksize * (d->C0 + d->C1) * d->Cout;
  if (dt_wgrad_bf16_narrow_supported(d)) {
    // Cin, Cout in {16, 32} at full resolution: the lean persistent kernel (conv_bf16_narrow.hip), one slab per workgroup
    const int parts = dt_wgrad_bf16_narrow_launch(d, src0, dy, workspace, in_scale, in_shift, st);
    if (parts < 0) return parts;
    hipLaunchKernelGGL(wgrad_bf16_final_kernel, dim3((unsigned)((E / 4 + 15) / 16)), dim3(256), 0, st, workspace, dw_hwio,
                       parts, E);
    DT_LAUNCH_CHECK();
    return DT_OK;
  }
  if (in_scale == nullptr && wb_use_dma() && dt_wgrad_bf16_dma_supported(d)) {
    // 3x3 stride-1 layers with 64-channel blocks and a stored (untransformed) input: the LDS-DMA persistent kernel
    const int parts = dt_wgrad_bf16_dma_launch(d, src0, src1, dy, workspace, st);
    if (parts < 0) return parts;
    hipLaunchKernelGGL(wgrad_bf16_final_kernel, dim3((unsigned)((E / 4 + 15) / 16)), dim3(256), 0, st, workspace, dw_hwio,
                       parts, E);
    DT_LAUNCH_CHECK();
    return DT_OK;
  }
  WgradBfArgs a;
  int tw;
  wb_cfg(d, &tw, &a.ksplit, &a.T, &a.ci_blocks, &a.co_blocks);
  a.src0 = (const __bf16*)src0; a.src1 = (const __bf16*)src1; a.dy = (const __bf16*)dy; a.ws = workspace;
  a.in_scale = in_scale; a.in_shift = in_shift;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.pad = d->pad;
  a.tiles_x = dt_cdiv(d->Wo, tw); a.tiles_y = dt_cdiv(d->Ho, 128 / tw);
  const int grid = a.ci_blocks * a.co_blocks * a.ksplit * (d->ksize == 4 ? 2 : 1);
  if (d->ksize == 4) {
    DT_REQUIRE(in_scale == nullptr, "wgrad_bf16: the stem takes no input transform");
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<4, 1, 32, false, 32>), dim3(grid), dim3(256), 0, st, a);
    DT_LAUNCH_CHECK();
    rc = DT_OK;
  } else if (d->ksize == 3 && d->stride == 1) rc = tw == 32 ? wb_launch<3, 1, 32>(a, grid, st) : wb_launch<3, 1, 16>(a, grid, st);
  else if (d->ksize == 3) rc = tw == 32 ? wb_launch<3, 2, 32>(a, grid, st) : wb_launch<3, 2, 16>(a, grid, st);
  else if (d->stride == 2) rc = tw == 32 ? wb_launch<1, 2, 32>(a, grid, st) : wb_launch<1, 2, 16>(a, grid, st);
  else rc = tw == 32 ? wb_launch<1, 1, 32>(a, grid, st) : wb_launch<1, 1, 16>(a, grid, st);   // ResUnet identity_conv
  if (rc != DT_OK) return rc;
  DT_REQUIRE((E & 3) == 0, "wgrad_bf16: weight tensor size must be a multiple of 4");
  const int64_t g = (E / 4 + 15) / 16;
  hipLaunchKernelGGL(wgrad_bf16_final_kernel, dim3((unsigned)g), dim3(256), 0, st, workspace, dw_hwio, a.ksplit, E);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
