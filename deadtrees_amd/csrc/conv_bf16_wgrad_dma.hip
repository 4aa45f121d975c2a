// bf16 weight gradient of the 3x3 stride-1 layers with 64-channel blocks — LDS-DMA staged, double-buffered, persistent.
//
// Replaces ATen convolution_backward(weight) (autograd of the smp.Unet(resnet34) convolutions reached from
// deadtrees/network/segmodel.py:214-222 under the reference's AMP setting, protocol.md:27) for the layers the round-1
// kernel conv_wgrad_bf16_kernel<3, 1, *, false, 64> (conv_bf16.hip) ran at 0.23 of the bf16 MFMA peak: that kernel
// stages a 128-pixel tile through registers (ds_write_b128, 79 B/clk/CU) between TWO barriers per 72 MFMAs.
//
//   dW[tap][ci][co] (fp32) = sum_pixels x[pix + tap][ci] * dy[pix][co]       v_mfma_f32_32x32x16_bf16, K = 16 pixels
//
// Design (MI355X-first):
//   * persistent 512-thread workgroups, one per CU: a workgroup owns one 64 ci x 64 co block of dW for a contiguous range
//     of 128-pixel tiles (split-K over workgroups); its 8 waves = 2 (ci halves) x 2 (co halves) x 2 (pixel-row halves of
//     the tile), 9 tap accumulators (144 registers) per wave, two waves per SIMD: one wave's DMA issue and LDS reads run
//     beside its partner's MFMAs;
//   * both operands go global -> LDS by DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs, no ds_write);
//     padding / ragged edges copy a 16-byte zero block instead;
//   * two 46 KiB LDS buffers (x halo image 30 KiB + dy image 16 KiB): the DMA of tile t+1 runs under the MFMAs of tile t,
//     ONE barrier per tile;
//   * the images stay pixel-major (what the DMA writes) and are read with ds_read_b64_tr_b16 (hardware transpose: both
//     MFMA operands need 8 consecutive PIXELS per lane).  A DMA writes lane-linear, so rows are unpadded 128-byte pixel
//     rows and the bank-conflict fix is an XOR on the SOURCE address: 16-byte slot s of LDS row q holds channel segment
//     s ^ 4*((q >> 1) & 1) — the four pixel rows of a transposed read then cover the four 64-byte quarters of the 256-byte
//     bank row.  Halo rows are padded to a multiple of 8 pixels (40 / 24) so that a DMA piece never crosses a halo row:
//     all per-piece address arithmetic is scalar; with even pitches the swizzle phase of every fragment read is a
//     compile-time function of the tap column -> three per-lane base addresses + immediate offsets, no address VALU;
//   * an x fragment (halo row, 16-pixel column block, tap column) is read ONCE and feeds up to three MFMAs (the tap rows
//     it serves for different output rows); the dy fragments of the wave's rows stay in registers for the whole tile:
//     88 transposed reads per 72 MFMAs instead of 160;
//   * at the end the two pixel halves are summed through LDS (fixed order) and the block goes to the split-K workspace
//     that wgrad_bf16_final_kernel (conv_bf16.hip) reduces in one launch — deterministic, no atomics.
#include "conv_bf16.h"

#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4w __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3)))* wd_lptr;
typedef const void __attribute__((address_space(1)))* wd_gptr;
typedef bf16x4w __attribute__((address_space(3)))* wd_trptr;

#define WD_OOB 0x80000000u     // byte offset beyond every buffer this kernel takes (host check: operands < 2 GiB)
#define WD_MAX_WGS 256
#define WD_XBYTES (30 * 1024)  // x halo image: 30 pieces of 8 pixel rows x 128 B
#define WD_BUF (46 * 1024)     // + dy image: 16 pieces

__device__ __attribute__((aligned(64))) unsigned wd_zero_block[16];

struct WgDmaArgs {
  const __bf16* src0;
  const __bf16* src1;
  const __bf16* dy;
  float* ws;                   // [parts = ksplit][9][Cin][Cout] fp32 slabs
  int B, Hin, Win, C0, C1, mode0, Cout;
  int tiles_x, tiles_y, T, ci_blocks, co_blocks, ksplit, tps;   // tps: tiles per split
  unsigned bytes0, bytes1, dybytes;
};

template <int K, int N, class F>
__device__ __forceinline__ void wd_static_for(F&& f) {
  if constexpr (K < N) {
    f(std::integral_constant<int, K>{});
    wd_static_for<K + 1, N>(f);
  }
}

// TW = 32: tiles of 4 rows x 32 pixels (halo 6 x 40); TW = 16 (maps at most 16 pixels wide): 8 rows x 16 (halo 10 x 24)
template <int TW>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_bf16_dma_kernel(const WgDmaArgs a) {
  constexpr int ROWS = 128 / TW, XS = TW / 16, HWP = TW + 8, HROWS = ROWS + 2, PPR = HWP / 8;
  constexpr int WROWS = ROWS / 2;            // output rows of one pixel half
  static_assert(HROWS * PPR == 30, "x halo image = 30 DMA pieces");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * WD_BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave >> 2, wci = (wave >> 1) & 1, wco = wave & 1;

  const int nblocks = a.ci_blocks * a.co_blocks;
  const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);   // the blocks of one split (same tiles) share an XCD's L2
  const int blk = wg % nblocks, split = wg / nblocks;
  const int ci0 = 64 * (blk / a.co_blocks), co0 = 64 * (blk % a.co_blocks);
  const bool use0 = ci0 < a.C0;                            // the 64-channel input block lies in one source (host check)
  const int Cs = use0 ? a.C0 : a.C1, cbase = use0 ? ci0 : ci0 - a.C0, mode = use0 ? a.mode0 : 0;
  const int Hs = mode ? (a.Hin >> 1) : a.Hin, Ws = mode ? (a.Win >> 1) : a.Win;

  // ---- DMA lane roles: lane l fills slot (l & 7) of row (l >> 3) of a piece with channel segment slot ^ 4*((row>>1)&1)
  const int l8 = lane >> 3;
  const int seg = (lane & 7) ^ (((lane >> 4) & 1) << 2);
  const unsigned xlane_e = (unsigned)(cbase + seg * 8), ylane_e = (unsigned)(co0 + seg * 8);   // element offsets
  const __bf16* xsrc = use0 ? a.src0 : a.src1;
  const __bf16* zsrc = reinterpret_cast<const __bf16*>(wd_zero_block);

  // ---- staging context: the tile whose operands are DMA-ed next (one tile ahead of the multiplication)
  const int t_begin = split * a.tps, t_end = (t_begin + a.tps < a.T) ? t_begin + a.tps : a.T;
  int s_tx, s_ty, s_b;
  {
    const int per_img = a.tiles_x * a.tiles_y;
    s_b = t_begin / per_img;
    const int rem = t_begin - s_b * per_img;
    s_ty = rem / a.tiles_x;
    s_tx = rem - s_ty * a.tiles_x;
  }
  // piece k of a tile (46 per tile; wave w issues pieces w, w + 8, ...): 0..29 the x halo image (halo row k / PPR, 8-pixel
  // column block k % PPR), 30..45 the dy image.  Everything but the lane's column part is scalar arithmetic.
  auto stage_piece = [&](int i, int buf) {
    const int k = wave + 8 * i;
    if (k >= 46) return;
    unsigned char* dst = lds + buf + k * 1024;
    const __bf16* g = zsrc;     // padding / ragged edges: every such lane copies the same 16 zero bytes
    if (k < 30) {
      const int hy = k / PPR;                                      // scalar (wave-uniform k < 30)
      const int j = k - hy * PPR;
      const int iy = s_ty * ROWS - 1 + hy;
      const int ix = s_tx * TW - 1 + 8 * j + l8;
      const int lim = (s_tx * TW + TW + 1 < a.Win) ? s_tx * TW + TW + 1 : a.Win;   // halo columns 0 .. TW + 1 only
      if ((unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)lim) {
        const int sx = mode ? (ix >> 1) : ix;
        const unsigned e = ((unsigned)((s_b * Hs + (mode ? (iy >> 1) : iy)) * Ws) + (unsigned)sx) * (unsigned)Cs + xlane_e;
        g = xsrc + e;
      }
    } else {
      const int kk = k - 30;
      const int row = kk / (TW / 8), j = kk - row * (TW / 8);
      const int oy = s_ty * ROWS + row;
      const int ox = s_tx * TW + 8 * j + l8;
      if (oy < a.Hin && ox < a.Win) g = a.dy + ((unsigned)((s_b * a.Hin + oy) * a.Win) + (unsigned)ox) * (unsigned)a.Cout + ylane_e;
    }
    // global_load_lds, not buffer_load ... lds: behind the buffer form hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front
    // of the next LDS read (it cannot prove that the read misses the DMA's destination) — every piece's whole latency was
    // exposed (first version of this kernel: 0.7x the register-staged one); the global form is not tracked that way
    __builtin_amdgcn_global_load_lds((wd_gptr)g, (wd_lptr)dst, 16, 0, 0);
  };
  auto stage_advance = [&]() {
    if (++s_tx == a.tiles_x) {
      s_tx = 0;
      if (++s_ty == a.tiles_y) {
        s_ty = 0;
        ++s_b;
      }
    }
  };

  // ---- fragment read bases (bytes inside a buffer); transposed-read lane roles as in conv_wgrad_bf16_kernel: pixel half
  // h, channel half gsel of the wave's 32 channels, pixel row q4 of the 4-row block, 4-channel group p4
  const int h = lane >> 5, gsel = (lane >> 4) & 1, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int xch = 64 * wci + 32 * gsel + 8 * p4, ych = 64 * wco + 32 * gsel + 8 * p4;   // byte inside the 128-B pixel row
  // the row's swizzle phase ((q >> 1) & 1 of its LDS row index q): first pixel of a fragment = even multiple + tap column
  // kw, so the phase is (q4 >> 1) for kw = 0, ((q4 + 1) >> 1) & 1 for kw = 1, (q4 >> 1) ^ 1 for kw = 2 (file header)
  const int rowb = (8 * h + q4) * 128;
  const int abase0 = rowb + (xch ^ (((q4 >> 1) & 1) << 6));
  const int abase1 = rowb + (xch ^ ((((q4 + 1) >> 1) & 1) << 6));
  const int abase2 = rowb + (xch ^ ((((q4 >> 1) & 1) ^ 1) << 6));
  const int bbase = WD_XBYTES + rowb + (ych ^ (((q4 >> 1) & 1) << 6));

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // Transposed fragment reads as inline asm with explicit waits: behind a pending LDS-DMA hipcc (ROCm 7.2) puts
  // `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 builtin (the intrinsic carries no address information, so
  // the read "may alias" the DMA's destination): each piece's whole L2 / HBM latency was exposed right after its issue
  // (first versions of this kernel: 0.65-0.75x the register-staged one).  The ordering that matters is established by
  // hand instead: vmcnt(0) + barrier once per tile (below); LDS returns reads in order, so `lgkmcnt(2)` after issuing the
  // next fragment's two reads means the current fragment has landed.
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  struct Frag { u32x2 lo, hi; };
  const unsigned lds0 = (unsigned)(unsigned long)(wd_lptr)lds;
  auto frag_issue = [&](Frag& fr, unsigned base, auto imm_tag) {
    constexpr int IMM = decltype(imm_tag)::value;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(fr.lo), "=&v"(fr.hi)
                 : "v"(base), "n"(IMM), "n"(IMM + 4 * 128)
                 : "memory");
  };
  auto frag_value = [&](const Frag& fr) -> bf16x8 { return __builtin_bit_cast(bf16x8, fr); };

  // ---- prologue: the first tile's operands
  if (t_begin < t_end) {
#pragma unroll
    for (int i = 0; i < 6; ++i) stage_piece(i, 0);
    stage_advance();
  }
  int step = 0;
  for (int tile = t_begin; tile < t_end; ++tile, ++step) {
    const int cur = (step & 1) * WD_BUF, nxt = WD_BUF - cur;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave's DMA pieces of this tile have
    const bool more = tile + 1 < t_end;                                         // landed; nobody still reads buffer `nxt`
    // per-lane read bases of this tile: buffer + the wave's pixel half (runtime) folded in, the rest are immediates
    const unsigned xoff = lds0 + cur + (wpx * WROWS * HWP) * 128;
    const unsigned xa0 = xoff + abase0, xa1 = xoff + abase1, xa2 = xoff + abase2;
    const unsigned yb = lds0 + cur + bbase + (wpx * WROWS * TW) * 128;
    // dy fragments of this wave's rows: read once per tile
    Frag fbr[WROWS][XS];
    wd_static_for<0, WROWS * XS>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      frag_issue(fbr[i / XS][i % XS], yb, std::integral_constant<int, ((i / XS) * TW + 16 * (i % XS)) * 128>{});
    });
    // x fragments: halo rows hr = 0 .. WROWS + 1 of this wave's half; fragment (hr, xs, kw) serves output row r = hr - kh
    // with tap (kh, kw) for kh = 0..2.  One fragment ahead in registers.
    constexpr int NF = (WROWS + 2) * XS * 3;
    Frag far[2];
    frag_issue(far[0], xa0, std::integral_constant<int, 0>{});
    bf16x8 fb[WROWS][XS];
    wd_static_for<0, NF>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      constexpr int kw = f % 3, xs = (f / 3) % XS, hr = f / (3 * XS);
      if constexpr (f + 1 < NF) {
        constexpr int f1 = f + 1, kw1 = f1 % 3, xs1 = (f1 / 3) % XS, hr1 = f1 / (3 * XS);
        frag_issue(far[f1 & 1], kw1 == 0 ? xa0 : (kw1 == 1 ? xa1 : xa2),
                   std::integral_constant<int, (hr1 * HWP + 16 * xs1 + kw1) * 128>{});
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (f == 0) {   // everything older than the two reads just issued has landed: the dy fragments too
#pragma unroll
        for (int r = 0; r < WROWS; ++r)
#pragma unroll
          for (int x2 = 0; x2 < XS; ++x2) fb[r][x2] = frag_value(fbr[r][x2]);
      }
      const bf16x8 av = frag_value(far[f & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int r = hr - kh;
        if (r >= 0 && r < WROWS)
          acc[3 * kh + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, fb[r][xs], acc[3 * kh + kw], 0, 0, 0);
      }
      // the next tile's DMA pieces go out with the first six fragments (a piece needs no registers)
      if constexpr (f < 6) {
        if (more) stage_piece(f, nxt);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (more) stage_advance();
  }

  // ---- sum the two pixel halves through LDS (two rounds of taps: 4 waves x 5 taps x 4 KiB = 80 KiB), then the block
  // goes to the split's slab of the workspace: D layout row (ci) = (i & 3) + 8 (i >> 2) + 4 h, column (co) = lane & 31
  float* red = reinterpret_cast<float*>(lds);
  const int quad = wave & 3;
  const int Cin = a.C0 + a.C1;
  const int r = lane & 31;
  const int co = co0 + 32 * wco + r;
#pragma unroll 1
  for (int rnd = 0; rnd < 2; ++rnd) {
    const int t0 = rnd * 5, tn = rnd ? 4 : 5;
    __syncthreads();
    if (wpx == 1) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (t < tn)
#pragma unroll
          for (int i = 0; i < 16; ++i) red[((quad * 5 + t) * 16 + i) * 64 + lane] = rnd ? acc[5 + (t < 4 ? t : 0)][i] : acc[t][i];
    }
    __syncthreads();
    if (wpx == 0) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (t < tn)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float v = (rnd ? acc[5 + (t < 4 ? t : 0)][i] : acc[t][i]) + red[((quad * 5 + t) * 16 + i) * 64 + lane];
            const int ci = ci0 + 32 * wci + (i & 3) + 8 * (i >> 2) + 4 * h;
            a.ws[(((size_t)split * 9 + t0 + t) * Cin + ci) * a.Cout + co] = v;
          }
    }
  }
}

// ---------------------------------------------------------------- host side
struct WdCfg {
  int tw, tiles_x, tiles_y, T, ci_blocks, co_blocks, ksplit, tps;
};

static WdCfg wd_cfg(const dt_conv_desc* d) {
  WdCfg c;
  c.tw = d->Wo > 16 ? 32 : 16;
  const int rows = 128 / c.tw;
  c.tiles_x = dt_cdiv(d->Wo, c.tw);
  c.tiles_y = dt_cdiv(d->Ho, rows);
  c.T = d->B * c.tiles_x * c.tiles_y;
  c.ci_blocks = (d->C0 + d->C1) / 64;
  c.co_blocks = d->Cout / 64;
  const int nblocks = c.ci_blocks * c.co_blocks;
  int ks = WD_MAX_WGS / nblocks;
  if (ks < 1) ks = 1;
  if (ks > c.T) ks = c.T;
  c.tps = dt_cdiv(c.T, ks);
  c.ksplit = dt_cdiv(c.T, c.tps);      // no empty splits
  return c;
}

int dt_wgrad_bf16_dma_supported(const dt_conv_desc* d) {
  if (d == nullptr || d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->mode0 < 0 || d->mode0 > 1) return 0;
  if ((d->C0 % 64) != 0 || (d->C1 % 64) != 0 || (d->Cout % 64) != 0 || d->C0 <= 0) return 0;
  if (d->Ho != d->Hin || d->Wo != d->Win) return 0;
  if (d->mode0 == 1 && ((d->Hin | d->Win) & 1)) return 0;
  const size_t px0 = (size_t)d->B * (d->mode0 ? (d->Hin / 2) * (size_t)(d->Win / 2) : (size_t)d->Hin * d->Win);
  if (px0 * d->C0 * 2 >= 0x80000000ull || (size_t)d->B * d->Hin * d->Win * d->C1 * 2 >= 0x80000000ull) return 0;
  if ((size_t)d->B * d->Ho * d->Wo * d->Cout * 2 >= 0x80000000ull) return 0;
  if (((d->C0 + d->C1) / 64) * (d->Cout / 64) > WD_MAX_WGS) return 0;
  return 1;
}

size_t dt_wgrad_bf16_dma_workspace(const dt_conv_desc* d) {
  const WdCfg c = wd_cfg(d);
  return (size_t)c.ksplit * 9 * (d->C0 + d->C1) * d->Cout * sizeof(float);
}

// -> number of split-K slabs written to `ws` (the caller reduces them with wgrad_bf16_final_kernel), or a negative code
int dt_wgrad_bf16_dma_launch(const dt_conv_desc* d, const void* src0, const void* src1, const void* dy, float* ws,
                             hipStream_t st) {
  DT_REQUIRE(dt_wgrad_bf16_dma_supported(d), "wgrad_bf16_dma: layer shape not supported");
  const WdCfg c = wd_cfg(d);
  WgDmaArgs a;
  a.src0 = (const __bf16*)src0; a.src1 = (const __bf16*)src1; a.dy = (const __bf16*)dy; a.ws = ws;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0; a.Cout = d->Cout;
  a.tiles_x = c.tiles_x; a.tiles_y = c.tiles_y; a.T = c.T;
  a.ci_blocks = c.ci_blocks; a.co_blocks = c.co_blocks; a.ksplit = c.ksplit; a.tps = c.tps;
  const size_t px0 = (size_t)d->B * (d->mode0 ? (d->Hin / 2) * (size_t)(d->Win / 2) : (size_t)d->Hin * d->Win);
  a.bytes0 = (unsigned)(px0 * d->C0 * 2);
  a.bytes1 = (unsigned)((size_t)d->B * d->Hin * d->Win * d->C1 * 2);
  a.dybytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->Cout * 2);
  const dim3 g((unsigned)(c.ci_blocks * c.co_blocks * c.ksplit)), blk(512);
  if (c.tw == 32) hipLaunchKernelGGL((conv3x3_wgrad_bf16_dma_kernel<32>), g, blk, 0, st, a);
  else hipLaunchKernelGGL((conv3x3_wgrad_bf16_dma_kernel<16>), g, blk, 0, st, a);
  DT_LAUNCH_CHECK();
  return c.ksplit;
}
