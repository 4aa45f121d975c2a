// 3x3 stride-1 bf16 convolution (forward and data gradient) of the NARROW full-resolution decoder layers: Cin, Cout in
// {16, 32} at 256x256 / 512x512 (smp UnetDecoder blocks 3 and 4: dec3.conv2 32->32, dec4.conv1 up(32)->16, dec4.conv2
// 16->16 and their data gradients).  Replaces the same ATen conv2d / convolution_backward(input) calls as conv_bf16.hip
// (deadtrees/network/segmodel.py:214 under the reference's AMP setting, protocol.md:27).
//
// Why a kernel of their own (round 3, measured): these layers move 64 B per pixel and need 4.6-18 kFLOP per pixel — in
// bf16 they are HBM-bound by a wide margin (dec4.conv2 at B = 64: 1.07 GB -> 0.2 ms at 5.3 TB/s), but the general
// register-staged kernel conv_fwd_bf16_kernel<3, 1, 32, 32, 16, 2, *> spends ~2,000 instructions per wave on a
// 256-pixel tile that holds 18 MFMAs (PMC, scripts/pmc_probe.sh: matrix pipe busy 17 %, waves issuing 46 % of the time,
// ~800 VALU instructions per wave and tile) and runs at 0.4 of the HBM roof: per-element bounds tests, 32-channel-wide
// tiles for 16 output channels, run-time epilogue variants, weights re-staged through LDS for every tile.  Here:
//   * persistent workgroups walk 8 x 32-pixel tiles; the WEIGHTS LIVE IN REGISTERS for the whole kernel
//     (v_mfma_f32_16x16x32_bf16: a 16-channel N block needs 4 registers per K step; K step = two taps x 16 input
//     channels, or one tap x 32) — no weight traffic, no weight barrier;
//   * 16-wide N: no padding of 16 output channels to 32;
//   * input halo tile pixel-major in LDS (32-byte pixel rows for 16 channels, 96-byte pitch for 32: both conflict-free
//     for the ds_read_b128 fragment reads, brute-forced against the bank rule of MI355X_MICROARCH.md), every fragment
//     address = one per-lane base + an immediate;
//   * the next tile's input is in flight in registers (3 / 6 x 16 B per thread) while the current tile is multiplied;
//     the producer's BatchNorm + ReLU (virtual activations) is applied on the way into LDS like in conv_bf16.hip;
//   * epilogue: adjacent-channel lane pairs are packed with one DPP swap, the tile is transposed through LDS and leaves
//     as 16-byte stores; BatchNorm statistics (from the fp32 accumulators) or the fused BatchNorm-backward sums are
//     accumulated in registers over ALL tiles of the workgroup and written as ONE row per workgroup;
//   * edge tiles (ragged maps) take a masked path, interior tiles test nothing per element.
// Same arithmetic as conv_fwd_bf16_kernel: bf16 operands, fp32 accumulation over (tap, channel), one rounding at the
// store; the K order differs (taps paired), so results agree to fp32 accumulation-order noise, not bit for bit.
#include "conv_bf16.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define NR_TW 32
#define NR_TH 8
#define NR_HW (NR_TW + 2)
#define NR_HH (NR_TH + 2)
#define NR_PIX (NR_HH * NR_HW)   // 340 halo pixels
#define NR_CUS 256

template <int CB, int NB>
struct NrGeom {
  static constexpr int PITCH = CB == 1 ? 32 : 96;        // bytes per halo pixel in LDS
  static constexpr int SLOTS = 2 * CB;                   // 16-byte slots (8 channels) per pixel
  static constexpr int FILL = (NR_PIX * SLOTS + 255) / 256;
  static constexpr int KS = CB == 1 ? 5 : 9;             // K steps of 32: two taps x 16 channels, or one tap x 32
  static constexpr int OUTP = 32 * NB + 16;              // bytes per pixel of the store-staging image
  static constexpr int SEGS = 2 * NB;                    // 16-byte segments per output pixel
  static constexpr int LDS = NR_PIX * PITCH + 256 * OUTP + 2 * 256 * 8 * 4 * 0;
};

__device__ __forceinline__ float nr_swap_pair(float v) {   // the value of the lane's neighbour in its (even, odd) pair
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// CB = Cin / 16, NB = Cout / 16 (1 or 2 each); TF: BatchNorm + ReLU of the producer applied while staging;
// BNB: fused BatchNorm-backward sums of the layer the written gradient belongs to (virtual activation)
// UPB (with BNB): the convolution's input was a nearest x2 up-sampling (dec4.conv1's data gradient) — the 2x2 sums of its
// backward are taken on the fp32 accumulators (row pair = two M blocks of the wave, column pair = two registers of the
// lane) and the LOW-resolution gradient [B, Ho/2, Wo/2, COUT] is stored, with the sums of the layer below: the
// full-resolution gradient (1 GB written and read back per 64-tile step) and the dt_upsample2x_bwd_bn_bf16 pass disappear
template <int CB, int NB, bool TF, bool BNB, bool UPB = false>
__global__ __launch_bounds__(256, (CB == 2 && NB == 2) ? 2 : (CB == 1 && NB == 1 ? 4 : 3)) void conv3x3_bf16_narrow_kernel(const ConvBfArgs a, const int total_tiles) {
  static_assert(!UPB || BNB, "the up-sample backward form carries the BatchNorm-backward sums");
  using G = NrGeom<CB, NB>;
  constexpr int CIN = 16 * CB, COUT = 16 * NB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NR_PIX * G::PITCH + 256 * G::OUTP];
  unsigned char* lds_in = lds;
  unsigned char* lds_out = lds + NR_PIX * G::PITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kg = lane >> 4;
  const int Hs = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws = a.mode0 ? (a.Win >> 1) : a.Win;
  const int NG = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + NG - 1) / NG;
  auto tile_of = [&](int round) { return (int)xcd_remap(blockIdx.x + (unsigned)round * NG, (unsigned)total_tiles); };

  // ---- weights -> registers, once: B operand of K step s, N block nb: lane (n = m, kg) holds k = 8 kg .. 8 kg + 7
  bf16x8 wreg[G::KS][NB];
#pragma unroll
  for (int s = 0; s < G::KS; ++s)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (CB == 1) {
        const int tap = 2 * s + (kg >> 1);                 // K step = taps (2 s, 2 s + 1) x 16 channels; tap 9 = zeros
        if (tap < 9) v = *reinterpret_cast<const bf16x8*>(a.w + ((size_t)tap * COUT + 16 * nb + m) * CIN + 8 * (kg & 1));
      } else {
        v = *reinterpret_cast<const bf16x8*>(a.w + ((size_t)s * COUT + 16 * nb + m) * CIN + 8 * kg);
      }
      wreg[s][nb] = v;
    }

  // ---- fragment read addresses: M block mb of wave w = output row 2 w + (mb >> 1), columns 16 (mb & 1) + m
  int abase[CB == 1 ? G::KS : 1];
  if constexpr (CB == 1) {
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
      int tap = 2 * s + (kg >> 1);
      tap = tap < 9 ? tap : 8;                             // zero weights: any valid address
      abase[s] = ((2 * wave + tap / 3) * NR_HW + m + tap % 3) * G::PITCH + 16 * (kg & 1);
    }
  } else {
    abase[0] = (2 * wave * NR_HW + m) * G::PITCH + 16 * kg;
  }

  // ---- fill roles: element e = tid + 256 it of the halo image = (pixel e / SLOTS, slot e % SLOTS), fixed per thread
  int f_hy[G::FILL], f_hx[G::FILL], f_dst[G::FILL];
#pragma unroll
  for (int it = 0; it < G::FILL; ++it) {
    const int e = tid + 256 * it;
    const int pix = e / G::SLOTS, slot = e - pix * G::SLOTS;
    f_hy[it] = e < NR_PIX * G::SLOTS ? pix / NR_HW : -100000;     // out of the image: never valid
    f_hx[it] = pix % NR_HW;
    f_dst[it] = pix * G::PITCH + 16 * slot;
  }
  const int f_ch = 8 * ((tid % G::SLOTS));     // SLOTS divides 256: a thread's slot is the same in every iteration
  float tf_sc[TF ? 8 : 1], tf_sh[TF ? 8 : 1];
  if constexpr (TF) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      tf_sc[k] = a.in_scale[f_ch + k];
      tf_sh[k] = a.in_shift[f_ch + k];
    }
  }
  u32x4 rin[G::FILL];
  unsigned rvalid = 0;
  auto issue_loads = [&](int round) {
    const int sp = tile_of(round);
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int iy0 = ty * NR_TH - 1, ix0 = tx * NR_TW - 1;
    rvalid = 0;
#pragma unroll
    for (int it = 0; it < G::FILL; ++it) {
      const int iy = iy0 + f_hy[it], ix = ix0 + f_hx[it];
      const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      const int sy = a.mode0 ? (iy >> 1) : iy, sx = a.mode0 ? (ix >> 1) : ix;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(a.src0 + ((size_t)(b * Hs + sy) * Ws + sx) * CIN + f_ch);
      rvalid |= (ok ? 1u : 0u) << it;
      rin[it] = v;
    }
  };

  // ---- epilogue roles: 16-byte segment sg of pixels prow, prow + PER_IT, ...
  constexpr int PER_IT = 256 / G::SEGS;
  const int sg = tid % G::SEGS, prow = tid / G::SEGS;
  float s1[NB], s2[NB], q1[BNB ? 8 : 1], q2[BNB ? 8 : 1];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) s1[nb] = s2[nb] = 0.f;
  float b_mu[BNB ? 8 : 1], b_is[BNB ? 8 : 1], b_sc[BNB ? 8 : 1], b_sh[BNB ? 8 : 1];
  if constexpr (BNB) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      q1[k] = q2[k] = 0.f;
      b_mu[k] = a.bnb.mean[8 * sg + k];
      b_is[k] = a.bnb.invstd[8 * sg + k];
      b_sc[k] = a.bnb.act_scale[8 * sg + k];
      b_sh[k] = a.bnb.act_shift[8 * sg + k];
    }
  }
  const bool want_stats = a.stats != nullptr;
  const bool odd = (lane & 1) != 0;

  if (my_tiles > 0) issue_loads(0);
  for (int round = 0; round < my_tiles; ++round) {
    // registers -> LDS (the previous tile's fragment reads finished before its barrier C)
#pragma unroll
    for (int it = 0; it < G::FILL; ++it) {
      if (f_hy[it] >= 0) {
        u32x4 raw = rin[it];
        if constexpr (TF) {
          if ((rvalid >> it) & 1u) {   // the arithmetic of conv_fwd_bf16_kernel's staging; padding stays zero
            bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float f = (float)v[k] * tf_sc[k] + tf_sh[k];
              f = f < 0.f ? 0.f : f;
              v[k] = (__bf16)f;
            }
            raw = __builtin_bit_cast(u32x4, v);
          }
        }
        *reinterpret_cast<u32x4*>(lds_in + f_dst[it]) = raw;
      }
    }
    __syncthreads();   // barrier B: the tile is staged (and the previous tile's staging image has been read)
    if (round + 1 < my_tiles) issue_loads(round + 1);

    f32x4 acc[4][NB];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        int off = ((mb >> 1) * NR_HW + 16 * (mb & 1)) * G::PITCH;
        if constexpr (CB == 2) off += ((s / 3) * NR_HW + s % 3) * G::PITCH;
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(lds_in + abase[CB == 1 ? s : 0] + off);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wreg[s][nb], acc[mb][nb], 0, 0, 0);
      }
    }

    // ---- epilogue.  D layout: lane (n = m, g = kg), register i: pixel column 4 g + i of the M block, channel 16 nb + n
    const int sp = tile_of(round);
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * NR_TH, ox0 = tx * NR_TW;
    const bool interior = oy0 + NR_TH <= a.Ho && ox0 + NR_TW <= a.Wo;
    if (want_stats && !BNB) {   // BatchNorm statistics from the fp32 accumulators
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int oy = oy0 + 2 * wave + (mb >> 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acc[mb][nb][i];
            const bool ok = interior || (oy < a.Ho && ox0 + 16 * (mb & 1) + 4 * kg + i < a.Wo);
            s1[nb] += ok ? v : 0.f;
            s2[nb] += ok ? v * v : 0.f;
          }
      }
    }
    // pack channel pairs (n, n + 1) across the lane pair and write the bf16 tile [pixel][channel] to the staging image
    if constexpr (UPB) {
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int mbx = 0; mbx < 2; ++mbx)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          // source pixel (row wave, column 8 mbx + 2 kg + ip) = rows 2 wave, 2 wave + 1 x columns 2 ip, 2 ip + 1 of the block
          const float x0 = (acc[mbx][nb][0] + acc[mbx][nb][1]) + (acc[mbx + 2][nb][0] + acc[mbx + 2][nb][1]);
          const float x1 = (acc[mbx][nb][2] + acc[mbx][nb][3]) + (acc[mbx + 2][nb][2] + acc[mbx + 2][nb][3]);
          const float y0 = nr_swap_pair(x0), y1 = nr_swap_pair(x1);
          const float lo = odd ? y1 : x0, hi = odd ? x1 : y0;
          const bf16x2 pk = {(__bf16)lo, (__bf16)hi};
          const int pl = wave * (NR_TW / 2) + 8 * mbx + 2 * kg + (odd ? 1 : 0);
          *reinterpret_cast<bf16x2*>(lds_out + pl * G::OUTP + 2 * (16 * nb + (m & ~1))) = pk;
        }
    } else
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
          const float x0 = acc[mb][nb][2 * ip], x1 = acc[mb][nb][2 * ip + 1];
          const float y0 = nr_swap_pair(x0), y1 = nr_swap_pair(x1);
          // even lane: channels (n, n + 1) of pixel 2 ip; odd lane: channels (n - 1, n) of pixel 2 ip + 1
          const float lo = odd ? y1 : x0, hi = odd ? x1 : y0;
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          const bf16x2 pk = {(__bf16)lo, (__bf16)hi};
          const int pl = (2 * wave + (mb >> 1)) * NR_TW + 16 * (mb & 1) + 4 * kg + 2 * ip + (odd ? 1 : 0);
          *reinterpret_cast<bf16x2*>(lds_out + pl * G::OUTP + 2 * (16 * nb + (m & ~1))) = pk;
        }
    __syncthreads();   // barrier C: the staging image is complete; every wave is done with the input image
    // UPB: 4 x 16 source pixels (one pass, threads beyond 64 SEGS idle); else the 8 x 32 tile in SEGS passes
    constexpr int OW = UPB ? NR_TW / 2 : NR_TW, OPIX = UPB ? NR_TW * NR_TH / 4 : NR_TW * NR_TH;
    const int oh_ = UPB ? (a.Ho >> 1) : a.Ho, ow_ = UPB ? (a.Wo >> 1) : a.Wo;
    const int py0 = UPB ? (oy0 >> 1) : oy0, px0 = UPB ? (ox0 >> 1) : ox0;
#pragma unroll
    for (int it = 0; it < (OPIX + PER_IT - 1) / PER_IT; ++it) {
      const int pl = prow + it * PER_IT;
      const int oy = py0 + pl / OW, ox = px0 + pl % OW;
      if ((OPIX % PER_IT == 0 || pl < OPIX) && ((!UPB && interior) || (oy < oh_ && ox < ow_))) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(lds_out + pl * G::OUTP + 16 * sg);
        const size_t o = (((size_t)b * oh_ + oy) * ow_ + ox) * COUT + 8 * sg;
        *reinterpret_cast<u32x4*>(a.out + o) = raw;
        if constexpr (BNB) {   // virtual activation — the arithmetic of bn_bwd_reduce_bf16_kernel / conv_bf16_dma.hip
          const u32x4 yraw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(a.bnb.y) + o);
          const bf16x8 gv = __builtin_bit_cast(bf16x8, raw), yv = __builtin_bit_cast(bf16x8, yraw);
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

  // ---- ONE row of partial sums per workgroup: fixed-order reductions, no atomics.  The buffer has a.P rows (the same
  // for every variant of a layer shape: the host sizes it before it knows the variant's grid): rows beyond the grid are
  // written as zeros by the workgroups (row b + k * grid by workgroup b)
  if (want_stats) {
    float* red = reinterpret_cast<float*>(lds);   // >= 2 x 256 x 8 floats = 16 KB: inside the (idle) images
    __syncthreads();
    if constexpr (BNB) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[tid * 8 + k] = q1[k];
        red[2048 + tid * 8 + k] = q2[k];
      }
      __syncthreads();
      if (tid < 2 * COUT) {
        const int which = tid / COUT, c = tid % COUT;
        float t = 0.f;
        for (int r = 0; r < PER_IT; ++r) t += red[which * 2048 + (r * G::SEGS + c / 8) * 8 + (c & 7)];
        a.stats[((size_t)which * a.P + blockIdx.x) * COUT + c] = t;
        for (int r = (int)blockIdx.x + NG; r < a.P; r += NG) a.stats[((size_t)which * a.P + r) * COUT + c] = 0.f;
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float u1 = s1[nb], u2 = s2[nb];
        u1 += __shfl_xor(u1, 16, 64);
        u2 += __shfl_xor(u2, 16, 64);
        u1 += __shfl_xor(u1, 32, 64);
        u2 += __shfl_xor(u2, 32, 64);
        if (kg == 0) {
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
}

// ---------------------------------------------------------------- host side
static bool nr_enabled() {
  static const int on = [] {
    const char* e = getenv("DT_BF16_NARROW");
    return (e == nullptr || e[0] != '0') ? 1 : 0;
  }();
  return on != 0;
}

int dt_conv_bf16_narrow_supported(const dt_conv_desc* d) {
  if (!nr_enabled() || d == nullptr) return 0;
  if (d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->C1 != 0 || d->cout_split != 0 || d->accumulate != 0) return 0;
  if ((d->C0 != 16 && d->C0 != 32) || (d->Cout != 16 && d->Cout != 32)) return 0;
  if (d->mode0 != 0 && d->mode0 != 1) return 0;
  if (d->Ho != d->Hin || d->Wo != d->Win || d->Wo < 32 || d->Ho < 8) return 0;
  return 1;
}

static int nr_tiles(const dt_conv_desc* d) { return d->B * dt_cdiv(d->Ho, NR_TH) * dt_cdiv(d->Wo, NR_TW); }

// Workgroups of an instantiation that fit one CU, from its own code object: registers (512 per lane and SIMD, granule 8,
// one wave of the workgroup per SIMD) and LDS (160 KiB).  The kernel is latency-bound by the bytes it keeps in flight
// (PMC round 3: waves parked 53 % of the time at 3 workgroups per CU = 33 KB in flight per CU -> 3.6 TB/s), so the
// persistent grid takes every slot there is — and exactly those, so that all workgroups walk the same number of tiles.
template <class K>
static int nr_occupancy(K kernel) {
  hipFuncAttributes at;
  if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(kernel)) != hipSuccess) return 2;
  const int regs = ((at.numRegs + 7) / 8) * 8;
  int by_regs = regs > 0 ? 512 / regs : 8;
  const int by_lds = at.sharedSizeBytes > 0 ? (int)(163840 / at.sharedSizeBytes) : 8;
  int occ = by_regs < by_lds ? by_regs : by_lds;
  if (occ > 8) occ = 8;
  if (occ < 1) occ = 1;
  return occ;
}

template <int CB, int NB>
static int nr_occ_of(bool tf, int bnb) {   // bnb: 0 none, 1 fused BatchNorm-backward sums, 2 with the up-sample backward
  static int cache[4] = {0, 0, 0, 0};
  const int v = tf ? 1 : (bnb ? 1 + bnb : 0);
  if (cache[v] == 0)
    cache[v] = tf ? nr_occupancy(conv3x3_bf16_narrow_kernel<CB, NB, true, false>)
                  : (bnb == 2 ? nr_occupancy(conv3x3_bf16_narrow_kernel<CB, NB, false, true, true>)
                     : bnb  ? nr_occupancy(conv3x3_bf16_narrow_kernel<CB, NB, false, true>)
                            : nr_occupancy(conv3x3_bf16_narrow_kernel<CB, NB, false, false>));
  return cache[v];
}

static int nr_per_cu(const dt_conv_desc* d, bool tf, int bnb) {
  if (d->C0 == 16 && d->Cout == 16) return nr_occ_of<1, 1>(tf, bnb);
  if (d->C0 == 16) return nr_occ_of<1, 2>(tf, bnb);
  if (d->Cout == 16) return nr_occ_of<2, 1>(tf, bnb);
  return nr_occ_of<2, 2>(tf, bnb);
}

// rows of the statistics buffer: an upper bound of every variant's persistent grid (8 workgroups per CU), the same for
// all variants of a layer shape (dt_conv2d_bf16_stat_rows is asked before the variant is known)
int dt_conv_bf16_narrow_rows(const dt_conv_desc* d) {
  const int t = nr_tiles(d);
  return t < 8 * NR_CUS ? t : 8 * NR_CUS;
}

int dt_conv_bf16_narrow_grid(const dt_conv_desc* d, int tf, int bnb) {
  const int t = nr_tiles(d), per_cu = nr_per_cu(d, tf != 0, bnb);
  return t < per_cu * NR_CUS ? t : per_cu * NR_CUS;
}

template <int CB, int NB>
static int nr_launch(const ConvBfArgs& a, int grid, int total, bool tf, int bnb, hipStream_t st) {
  const dim3 g((unsigned)grid), blk(256);
  if (tf && bnb) return DT_EINVAL;
  if (tf) hipLaunchKernelGGL((conv3x3_bf16_narrow_kernel<CB, NB, true, false>), g, blk, 0, st, a, total);
  else if (bnb == 2) hipLaunchKernelGGL((conv3x3_bf16_narrow_kernel<CB, NB, false, true, true>), g, blk, 0, st, a, total);
  else if (bnb) hipLaunchKernelGGL((conv3x3_bf16_narrow_kernel<CB, NB, false, true>), g, blk, 0, st, a, total);
  else hipLaunchKernelGGL((conv3x3_bf16_narrow_kernel<CB, NB, false, false>), g, blk, 0, st, a, total);
  return DT_OK;
}

int dt_conv_bf16_narrow_launch(const dt_conv_desc* d, ConvBfArgs a, hipStream_t st, bool upsample_bwd) {
  DT_REQUIRE(dt_conv_bf16_narrow_supported(d), "conv_bf16_narrow: layer shape not supported");
  const bool tf = a.in_scale != nullptr;
  const int bnb = a.bnb.y != nullptr ? (upsample_bwd ? 2 : 1) : 0;
  DT_REQUIRE(!upsample_bwd || (bnb == 2 && ((d->Ho | d->Wo) & 1) == 0),
             "conv_bf16_narrow: the up-sample backward form needs the fused sums and an even map");
  DT_REQUIRE(!bnb || (a.bnb.act == nullptr && a.bnb.act_scale && a.bnb.act_shift && a.stats),
             "conv_bf16_narrow: the fused BatchNorm-backward sums take a virtual activation (scale / shift) and a stats buffer");
  DT_REQUIRE(!(tf && bnb), "conv_bf16_narrow: no input transform on the BatchNorm-backward form");
  a.tiles_x = dt_cdiv(d->Wo, NR_TW);
  a.tiles_y = dt_cdiv(d->Ho, NR_TH);
  const int total = nr_tiles(d), grid = dt_conv_bf16_narrow_grid(d, tf, bnb);
  a.P = dt_conv_bf16_narrow_rows(d);
  int rc;
  if (d->C0 == 16 && d->Cout == 16) rc = nr_launch<1, 1>(a, grid, total, tf, bnb, st);
  else if (d->C0 == 16) rc = nr_launch<1, 2>(a, grid, total, tf, bnb, st);
  else if (d->Cout == 16) rc = nr_launch<2, 1>(a, grid, total, tf, bnb, st);
  else rc = nr_launch<2, 2>(a, grid, total, tf, bnb, st);
  if (rc != DT_OK) return rc;
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ================================================================================================================
// Weight gradient of the same narrow layers: dW[tap][ci][co] (fp32) = sum_pixels x[pix + tap][ci] * dy[pix][co].
// Replaces ATen convolution_backward(weight) for dec3.conv2 / dec4.conv1 / dec4.conv2 under AMP, where the general
// kernel conv_wgrad_bf16_kernel<3, 1, 32, *, 32> (32x32x16 MFMAs on 16-channel operands, 128-pixel tiles between two
// barriers, run-time variants) ran at 0.15-0.5 PFLOP/s on layers that are HBM-bound (dec4.conv2 at B = 64: 1.07 GB).
//   * v_mfma_f32_16x16x32_bf16 with M = 16 input channels, N = 16 output channels, K = 32 PIXELS = one tile row;
//     both operands need 8 consecutive pixels per lane for a fixed channel: pixel-major LDS images read with
//     ds_read_b64_tr_b16.  The pixel order inside a K step is permuted (lane group kg holds columns 4 kg .. 4 kg + 3 and
//     16 + 4 kg .. 16 + 4 kg + 3) — the same for both operands — so that a 32-lane half of a transposed read covers 8
//     CONSECUTIVE pixel rows: 256 contiguous bytes at the 32-byte pitch, 8 distinct bank octets at the 96-byte pitch;
//   * an x fragment (halo row, tap column, channel block) feeds the MFMAs of up to three tap rows; the dy fragments of a
//     row are read once; 9 x CB x NB accumulators (4 registers each) live for the whole kernel;
//   * persistent over 8 x 32-pixel tiles with the next tile in flight in registers; the four waves split the tile's rows
//     and are summed through LDS once at the end (fixed order); one slab per workgroup -> wgrad_bf16_final_kernel.
typedef __bf16 bf16x4n __attribute__((ext_vector_type(4)));
typedef bf16x4n __attribute__((address_space(3)))* nr_trptr;

struct NrWgArgs {
  const __bf16* src0;
  const __bf16* dy;
  const float* in_scale;
  const float* in_shift;
  float* ws;                 // [gridDim.x][9][Cin][Cout] fp32 slabs
  int B, Hin, Win, mode0, tiles_x, tiles_y;
};

template <int CB, int NB, bool TF>
__global__ __launch_bounds__(256, (CB == 2 && NB == 2) ? 2 : 3) void conv3x3_wgrad_bf16_narrow_kernel(const NrWgArgs a,
                                                                                                    const int total_tiles) {
  constexpr int CIN = 16 * CB, COUT = 16 * NB;
  constexpr int PX = CB == 1 ? 32 : 96, PY = NB == 1 ? 32 : 96;       // bytes per pixel row of the two LDS images
  constexpr int XS = 2 * CB, YS = 2 * NB;                             // 16-byte slots per pixel
  constexpr int FX = (NR_PIX * XS + 255) / 256, FY = YS;              // fill iterations (256 output pixels x YS / 256)
  constexpr int LDS_X = NR_PIX * PX, LDS_Y = 256 * PY;
  constexpr int RED = 4 * CB * NB * 256 * 4;                          // one tap of three waves' accumulators... (per wave)
  __shared__ __attribute__((aligned(16))) unsigned char lds[(LDS_X + LDS_Y) > 3 * RED ? (LDS_X + LDS_Y) : 3 * RED];
  unsigned char* lx = lds;
  unsigned char* ly = lds + LDS_X;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int Hs = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws = a.mode0 ? (a.Win >> 1) : a.Win;
  const int NG = gridDim.x;
  const int my_tiles = (total_tiles - (int)blockIdx.x + NG - 1) / NG;
  auto tile_of = [&](int round) { return (int)xcd_remap(blockIdx.x + (unsigned)round * NG, (unsigned)total_tiles); };

  // ---- fill roles (fixed per thread)
  int x_hy[FX], x_hx[FX], x_dst[FX];
#pragma unroll
  for (int it = 0; it < FX; ++it) {
    const int e = tid + 256 * it;
    const int pix = e / XS, slot = e - pix * XS;
    x_hy[it] = e < NR_PIX * XS ? pix / NR_HW : -100000;
    x_hx[it] = pix % NR_HW;
    x_dst[it] = pix * PX + 16 * slot;
  }
  const int x_ch = 8 * (tid % XS), y_ch = 8 * (tid % YS);
  const int y_pix0 = tid / YS;                                         // + it * (256 / YS)
  float tf_sc[TF ? 8 : 1], tf_sh[TF ? 8 : 1];
  if constexpr (TF) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      tf_sc[k] = a.in_scale[x_ch + k];
      tf_sh[k] = a.in_shift[x_ch + k];
    }
  }
  u32x4 rx[FX], ry[FY];
  unsigned xvalid = 0;
  auto issue_loads = [&](int round) {
    const int sp = tile_of(round);
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * NR_TH, ox0 = tx * NR_TW;
    xvalid = 0;
#pragma unroll
    for (int it = 0; it < FX; ++it) {
      const int iy = oy0 - 1 + x_hy[it], ix = ox0 - 1 + x_hx[it];
      const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      const int sy = a.mode0 ? (iy >> 1) : iy, sx = a.mode0 ? (ix >> 1) : ix;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(a.src0 + ((size_t)(b * Hs + sy) * Ws + sx) * CIN + x_ch);
      xvalid |= (ok ? 1u : 0u) << it;
      rx[it] = v;
    }
#pragma unroll
    for (int it = 0; it < FY; ++it) {
      const int pl = y_pix0 + it * (256 / YS);
      const int oy = oy0 + pl / NR_TW, ox = ox0 + pl % NR_TW;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (oy < a.Hin && ox < a.Win) v = *reinterpret_cast<const u32x4*>(a.dy + ((size_t)(b * a.Hin + oy) * a.Win + ox) * COUT + y_ch);
      ry[it] = v;
    }
  };

  // ---- transposed-read lane bases: group kg reads pixel rows 4 kg + q4 (lo) and 16 + 4 kg + q4 (hi) of the K step,
  // 4 channels 4 p4 .. 4 p4 + 3 of the 16-channel block; lane i of the group receives channel i
  const int xlane = (4 * kg + q4) * PX + 8 * p4;
  const int ylane = (4 * kg + q4) * PY + 8 * p4;
  auto tr8 = [&](const unsigned char* p, int pitch) -> bf16x8 {
    const bf16x4n lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((nr_trptr)p);
    const bf16x4n hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((nr_trptr)(p + 16 * pitch));
    bf16x8 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = lo[k];
      v[4 + k] = hi[k];
    }
    return v;
  };

  f32x4 acc[9][CB][NB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[t][cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (my_tiles > 0) issue_loads(0);
  for (int round = 0; round < my_tiles; ++round) {
    __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
    for (int it = 0; it < FX; ++it) {
      if (x_hy[it] >= 0) {
        u32x4 raw = rx[it];
        if constexpr (TF) {
          if ((xvalid >> it) & 1u) {
            bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float f = (float)v[k] * tf_sc[k] + tf_sh[k];
              f = f < 0.f ? 0.f : f;
              v[k] = (__bf16)f;
            }
            raw = __builtin_bit_cast(u32x4, v);
          }
        }
        *reinterpret_cast<u32x4*>(lx + x_dst[it]) = raw;
      }
    }
#pragma unroll
    for (int it = 0; it < FY; ++it)
      *reinterpret_cast<u32x4*>(ly + (y_pix0 + it * (256 / YS)) * PY + 16 * (tid % YS)) = ry[it];
    __syncthreads();
    if (round + 1 < my_tiles) issue_loads(round + 1);

    // this wave's rows r = 0, 1 (tile rows 2 w + r): dy fragments once, x fragments per (halo row, tap column, block)
    bf16x8 fb[2][NB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) fb[r][nb] = tr8(ly + ylane + ((2 * wave + r) * NR_TW) * PY + 32 * nb, PY);
#pragma unroll
    for (int hr = 0; hr < 4; ++hr)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const bf16x8 av = tr8(lx + xlane + ((2 * wave + hr) * NR_HW + kw) * PX + 32 * cb, PX);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int r = hr - kh;
            if (r >= 0 && r < 2) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[3 * kh + kw][cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, fb[r][nb], acc[3 * kh + kw][cb][nb], 0, 0, 0);
            }
          }
        }
  }

  // ---- sum the four waves' accumulators through LDS, one tap at a time (fixed order 1, 2, 3), and write the workgroup's
  // slab: D row (ci) = 4 kg + i, column (co) = lane & 15
  float* red = reinterpret_cast<float*>(lds);
  const int n = lane & 15;
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if (wave != 0) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = 0.f;
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) v = tt == t ? acc[tt][cb][nb][i] : v;   // static register selects
            red[(((wave - 1) * CB * NB + cb * NB + nb) * 4 + i) * 64 + lane] = v;
          }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = 0.f;
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) v = tt == t ? acc[tt][cb][nb][i] : v;
#pragma unroll
            for (int w = 0; w < 3; ++w) v += red[((w * CB * NB + cb * NB + nb) * 4 + i) * 64 + lane];
            const int ci = 16 * cb + 4 * kg + i, co = 16 * nb + n;
            a.ws[(((size_t)blockIdx.x * 9 + t) * CIN + ci) * COUT + co] = v;
          }
    }
  }
}

int dt_wgrad_bf16_narrow_supported(const dt_conv_desc* d) {
  if (!nr_enabled() || d == nullptr) return 0;
  if (d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->C1 != 0) return 0;
  if ((d->C0 != 16 && d->C0 != 32) || (d->Cout != 16 && d->Cout != 32)) return 0;
  if (d->mode0 != 0 && d->mode0 != 1) return 0;
  if (d->Ho != d->Hin || d->Wo != d->Win || d->Wo < 32 || d->Ho < 8) return 0;
  return 1;
}

template <int CB, int NB>
static int nrw_occ(bool tf) {
  static int cache[2] = {0, 0};
  if (cache[tf] == 0)
    cache[tf] = tf ? nr_occupancy(conv3x3_wgrad_bf16_narrow_kernel<CB, NB, true>)
                   : nr_occupancy(conv3x3_wgrad_bf16_narrow_kernel<CB, NB, false>);
  return cache[tf];
}

size_t dt_wgrad_bf16_narrow_workspace(const dt_conv_desc* d) {
  const int t = nr_tiles(d);
  const size_t parts = (size_t)(t < 8 * NR_CUS ? t : 8 * NR_CUS);
  return parts * 9 * d->C0 * d->Cout * sizeof(float);
}

// -> number of slabs written to `ws` (one per workgroup), or a negative error code
int dt_wgrad_bf16_narrow_launch(const dt_conv_desc* d, const void* src0, const void* dy, float* ws, const float* in_scale,
                                const float* in_shift, hipStream_t st) {
  DT_REQUIRE(dt_wgrad_bf16_narrow_supported(d), "wgrad_bf16_narrow: layer shape not supported");
  NrWgArgs a;
  a.src0 = (const __bf16*)src0; a.dy = (const __bf16*)dy; a.in_scale = in_scale; a.in_shift = in_shift; a.ws = ws;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.mode0 = d->mode0;
  a.tiles_x = dt_cdiv(d->Wo, NR_TW); a.tiles_y = dt_cdiv(d->Ho, NR_TH);
  const int total = nr_tiles(d);
  const bool tf = in_scale != nullptr;
  int occ;
  if (d->C0 == 16 && d->Cout == 16) occ = nrw_occ<1, 1>(tf);
  else if (d->C0 == 16) occ = nrw_occ<1, 2>(tf);
  else if (d->Cout == 16) occ = nrw_occ<2, 1>(tf);
  else occ = nrw_occ<2, 2>(tf);
  const int grid = total < occ * NR_CUS ? total : occ * NR_CUS;
  const dim3 g((unsigned)grid), blk(256);
#define NRW_LAUNCH(CBv, NBv)                                                                                     \
  do {                                                                                                           \
    if (tf) hipLaunchKernelGGL((conv3x3_wgrad_bf16_narrow_kernel<CBv, NBv, true>), g, blk, 0, st, a, total);      \
    else hipLaunchKernelGGL((conv3x3_wgrad_bf16_narrow_kernel<CBv, NBv, false>), g, blk, 0, st, a, total);       \
  } while (0)
  if (d->C0 == 16 && d->Cout == 16) NRW_LAUNCH(1, 1);
  else if (d->C0 == 16) NRW_LAUNCH(1, 2);
  else if (d->Cout == 16) NRW_LAUNCH(2, 1);
  else NRW_LAUNCH(2, 2);
#undef NRW_LAUNCH
  DT_LAUNCH_CHECK();
  return grid;
}
