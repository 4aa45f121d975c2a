// 3x3 stride-1 bf16 convolution (forward and data gradient), LDS-DMA staged and double-buffered.
//
// Same arithmetic and epilogues as conv_fwd_bf16_kernel (conv_bf16.hip): v_mfma_f32_32x32x16_bf16, fp32 accumulators,
// channel chunks of 32 in the same order, so the two kernels give bit-identical results.  What differs is how the
// operands reach the matrix pipe.  The register-staged kernel spends, per 32-channel chunk, two barriers and a
// ds_write_b128 pass (79 B/clk/CU, MI355X_MICROARCH.md LDS table) over 73 KB of operands of which 63 % are weights;
// measured it is bound by exactly those staging phases, not by MFMA issue or bytes.  Here:
//   * operands go global -> LDS by DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction, no VGPRs, no ds_write);
//     only a virtual input activation (the producer's BatchNorm + ReLU applied while staging) still passes through
//     registers, because a DMA cannot transform;
//   * two LDS buffers: the DMA of chunk c+1 runs under the MFMAs of chunk c -> ONE barrier per chunk;
//   * 8-wave workgroups (2 waves per SIMD: one wave's DMA issue / ds_reads run beside its partner's MFMAs) on a
//     512-pixel x 64-channel tile: 255 FLOP per staged byte instead of 161;
//   * a DMA writes lane-linear (wave-uniform base + 16 B x lane), so the LDS images are unpadded 64-byte rows and the
//     bank-conflict fix is an XOR swizzle applied to the per-lane SOURCE address and again to the fragment reads
//     (cdna_hip_programming.md 5.4 rule 21): 16-byte slot s of row q holds channel segment s ^ ((q >> 2) & 3);
//     every 16-lane group of a ds_read_b128 then touches 16 distinct slots of the 256-byte bank row;
//   * zero padding / zero insertion: lanes whose pixel lies outside the image read a 64-byte zero block instead.
// Replaces the same ATen ops as conv_bf16.hip (conv2d / convolution_backward(input) under the reference's AMP
// setting, protocol.md:27).
#include "conv_bf16.h"

#include <stdlib.h>
#include <string.h>

#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const void __attribute__((address_space(1)))* dm_gptr;
typedef void __attribute__((address_space(3)))* dm_lptr;

#define DM_TW 32
#define DM_TH 16
#define DM_HW (DM_TW + 2)
#define DM_HH (DM_TH + 2)
#define DM_IN_ROWS (DM_HH * DM_HW)                     // 612 halo pixels, 64 B (32 channels) each
#define DM_IN_PIECES ((DM_IN_ROWS + 15) / 16)          // 39 DMA pieces of 1 KiB
#define DM_TN 64
#define DM_W_PIECES (9 * DM_TN / 16)                   // 36
#define DM_BUF (1024 * (DM_IN_PIECES + DM_W_PIECES))   // 76,800 B per buffer
#define DM_TF_OFF (2 * DM_BUF)                         // scale[512] | shift[512] of the input transform
#define DM_TF_MAXC 512
#define DM_LDS (DM_TF_OFF + 2 * DM_TF_MAXC * 4)        // 157,696 B of the CU's 163,840
#define DM_PPW 5                                       // pieces per wave and operand (8 waves)
#define DM_MAX_WGS 256                                 // CUs of an MI355X

__device__ __attribute__((aligned(64))) unsigned dm_zero_block[16];

__device__ __forceinline__ void dm_dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((dm_gptr)g, (dm_lptr)l, 16, 0, 0);
}

// EPI selects the epilogue at compile time (register pressure: the persistent loop keeps the accumulators, the next
// tile's staging context and the epilogue's state live together): 0 plain store (+ BatchNorm statistics, split
// outputs), 1 plain store + fused BatchNorm-backward sums (virtual activation), 2 gradient join (fp32 staging, split
// outputs), 3 gradient join + fused BatchNorm-backward sums (stored activation).
template <bool TF, int EPI>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16_dma_kernel(const ConvBfArgs a, int total_tiles) {
  constexpr int TN = DM_TN, NT = TN / 32;
  constexpr int OUT_PITCH = TN + 8;    // bf16 per pixel row of the store-staging image
  constexpr int OUTF_PITCH = TN + 4;   // fp32 per pixel row of the gradient-join staging image
  static_assert(256 * OUTF_PITCH * 4 <= DM_BUF, "the epilogue staging image of a 256-pixel half fits one buffer");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[DM_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  float* lds_tf = reinterpret_cast<float*>(lds + DM_TF_OFF);
  if constexpr (TF) {
    for (int i = tid; i < a.C0; i += 512) {
      lds_tf[i] = a.in_scale[i];
      lds_tf[DM_TF_MAXC + i] = a.in_shift[i];
    }
  }
  const int Cin = a.C0 + a.C1;
  const int Hs0 = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws0 = a.mode0 ? (a.Win >> 1) : a.Win;
  const int nchunks = Cin / 32;
  const int G = gridDim.x;

  // ---- staging.  Piece p (1 KiB) = LDS rows 16p .. 16p+15; lane l fills slot (l & 3) of row 16p + (l >> 2) with
  // channel segment seg = slot ^ ((row >> 2) & 3) = (l & 3) ^ ((l >> 4) & 3) of that row's pixel / weight row.
  // The staging context (which tile / chunk the next DMA belongs to) runs one step ahead of the multiplication.
  const int seg = (lane & 3) ^ ((lane >> 4) & 3);
  int pix0[DM_PPW], pix1[DM_PPW], woff[DM_PPW];
  int s_tile = blockIdx.x, s_chunk = 0;        // tile id (before the XCD remap) and chunk of the next staging step
  auto stage_setup = [&](int tile_id) {
    const int wg = (int)xcd_remap((unsigned)tile_id, (unsigned)total_tiles);
    const int nt = wg % a.n_tiles, sp = wg / a.n_tiles;
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int iy0 = ty * DM_TH - a.pad, ix0 = tx * DM_TW - a.pad, n0 = nt * TN;
#pragma unroll
    for (int i = 0; i < DM_PPW; ++i) {
      const int row = (wave + 8 * i) * 16 + (lane >> 2);
      const int hy = row / DM_HW, hx = row - hy * DM_HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool inb = row < DM_IN_ROWS && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      bool ok0 = inb;
      if (a.mode0 == 2) ok0 = ok0 && (((iy | ix) & 1) == 0);   // zero insertion (transposed convolution)
      const int sy = a.mode0 ? (iy >> 1) : iy, sx = a.mode0 ? (ix >> 1) : ix;
      pix0[i] = ok0 ? (b * Hs0 + sy) * Ws0 + sx : -1;
      pix1[i] = inb ? (b * a.Hin + iy) * a.Win + ix : -1;
      // chunked weight image [tap][Cin/32][Cout][32]: weight row = tap * 64 + n; + chunk * Cout * 32 per step
      woff[i] = ((row >> 6) * (Cin >> 5) * a.Cout + n0 + (row & 63)) * 32 + 8 * seg;
    }
  };
  const __bf16* zsrc = reinterpret_cast<const __bf16*>(dm_zero_block);
  f32x4 rin[TF ? DM_PPW : 1];     // a virtual input activation passes through registers
  unsigned rin_valid = 0;
  int rin_c0 = 0;
  // The staging of step (s_tile, s_chunk) into buffer `buf` is 10 pieces per wave — 5 of the weight image, 5 of the
  // input image: DMA (or, for inputs that still need the producer's BatchNorm + ReLU, register loads) — issued one at
  // a time between the MFMA groups of the step being multiplied, then the staging context advances.
  auto stage_piece = [&](int k, int buf) {
    const int c0 = 32 * s_chunk;
    if (k < DM_PPW) {
      const int wp = wave + 8 * k;
      if (wp < DM_W_PIECES)
        dm_dma16(a.w + (size_t)woff[k] + (size_t)s_chunk * a.Cout * 32, lds + buf + (DM_IN_PIECES + wp) * 1024);
      return;
    }
    const int i = k - DM_PPW;
    const bool use0 = c0 < a.C0;
    const __bf16* src = use0 ? a.src0 : a.src1;
    const int C = use0 ? a.C0 : a.C1;
    const int cc = (use0 ? c0 : c0 - a.C0) + 8 * seg;
    const int p = use0 ? pix0[i] : pix1[i];
    const int ip = wave + 8 * i;
    if constexpr (TF) {
      if (i == 0) {
        rin_valid = 0;
        rin_c0 = c0;
      }
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ip < DM_IN_PIECES && p >= 0) {
        v = *reinterpret_cast<const f32x4*>(src + (size_t)p * C + cc);
        rin_valid |= 1u << i;
      }
      rin[TF ? i : 0] = v;
    } else {
      if (ip < DM_IN_PIECES) {
        const __bf16* g = p >= 0 ? src + (size_t)p * C + cc : zsrc;
        dm_dma16(g, lds + buf + ip * 1024);
      }
    }
  };
  auto stage_advance = [&]() {
    if (++s_chunk == nchunks) {
      s_chunk = 0;
      s_tile += G;
      if (s_tile < total_tiles) stage_setup(s_tile);
    }
  };
  // registers -> LDS for the chunk loaded by the last stage_issue (transform applied; padding stays zero)
  auto stage_write = [&](int buf) {
    const int cc = rin_c0 + 8 * seg;
    const bool tf_on = cc < a.C0;
    float sc[8], sh[8];
    if (tf_on) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sc[k] = lds_tf[cc + k];
        sh[k] = lds_tf[DM_TF_MAXC + cc + k];
      }
    }
#pragma unroll
    for (int i = 0; i < DM_PPW; ++i) {
      const int ip = wave + 8 * i;
      if (ip < DM_IN_PIECES) {
        f32x4 raw = rin[TF ? i : 0];
        if (tf_on && ((rin_valid >> i) & 1u)) {
          bf16x8 v = *reinterpret_cast<bf16x8*>(&raw);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float f = (float)v[k] * sc[k] + sh[k];
            f = f < 0.f ? 0.f : f;
            v[k] = (__bf16)f;
          }
          raw = *reinterpret_cast<f32x4*>(&v);
        }
        *reinterpret_cast<f32x4*>(lds + buf + ip * 1024 + lane * 16) = raw;
      }
    }
  };

  // ---- fragment read addresses (bytes inside a buffer).  A: halo pixel q, k-segment g = 2 ks + h -> q*64 + 16*(g ^ sw(q))
  // (recomputed per chunk from qb: 18 precomputed addresses would cost 18 VGPRs for the whole persistent loop)
  int qb = 2 * wave * DM_HW + r;
  const int bbase = DM_IN_PIECES * 1024 + r * 64 + ((((r >> 2) & 3) ^ h) << 4);   // + tap*4096 + j*2048, ^ (ks << 5)

  // epilogue roles: every thread moves 16-byte segment sg of pixel rows prow, prow + 64, ... of a 256-pixel half
  constexpr int SEGS = TN / 8, PER_IT = 512 / SEGS, ITS = 256 / PER_IT;
  const int sg = tid % SEGS, prow = tid / SEGS;
  const int hf = wave >> 2, lw = wave & 3;
  constexpr bool bnb = (EPI & 1) != 0, JOIN = (EPI & 2) != 0;

  // ---- persistent loop over this workgroup's tiles; the chunks of consecutive tiles form ONE pipelined stream:
  // while chunk s is multiplied the DMA of chunk s+1 (possibly the next tile's first) fills the other buffer, so a
  // tile's epilogue and the next tile's first loads overlap.  One barrier per chunk + four per tile.
  if constexpr (TF) __syncthreads();   // the transform table
  if (s_tile < total_tiles) {
    stage_setup(s_tile);
#pragma unroll
    for (int k = 0; k < 2 * DM_PPW; ++k) stage_piece(k, 0);
    stage_advance();
    if constexpr (TF) stage_write(0);
  }
  int step = 0;
  // BatchNorm (backward) partial sums: per tile, or — a.pstats: every tile of this workgroup has the same channel block —
  // kept in registers over all of the workgroup's tiles and written as ONE row per workgroup after the loop (<= 256 rows:
  // the finalize pass reads them directly, no row-reduction launch; no per-tile barriers / LDS passes for the sums)
  float s1[NT], s2[NT], q1[8], q2[8];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) q1[k] = q2[k] = 0.f;
  const bool pstats = a.pstats != 0;
  for (int tile = blockIdx.x; tile < total_tiles; tile += G) {
    const int wg = (int)xcd_remap((unsigned)tile, (unsigned)total_tiles);
    const int nt = wg % a.n_tiles, sp = wg / a.n_tiles;
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * DM_TH, ox0 = tx * DM_TW, n0 = nt * TN;
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][j][i] = 0.f;
    int cur = 0;
    for (int c = 0; c < nchunks; ++c, ++step) {
      cur = (step & 1) * DM_BUF;
      const int nxt = DM_BUF - cur;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of this step have landed
      __syncthreads();                                   // ... everybody's; and nobody still reads buffer `nxt`
      const bool more = s_tile < total_tiles;
      const unsigned char* bufp = lds + cur;
      asm volatile("" : "+v"(qb));   // keep the address arithmetic below inside the loop (see qb)
      // 18 k-steps (tap, ks) of 4 MFMAs; the fragments of step t+1 are requested before the MFMAs of step t are issued
      // (two named register sets, everything unrolled), so a wave's LDS latency hides under its own matrix work and
      // the compiler's waits become counted (lgkmcnt(4)) instead of a drain before every MFMA group
      bf16x8 fa[2][2], fb[2][NT];
      auto frag_load = [&](int t, bf16x8 (&av)[2], bf16x8 (&bv)[NT]) {
        const int tap = t >> 1, ks = t & 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int q = qb + (m + tap / 3) * DM_HW + tap % 3;
          av[m] = *reinterpret_cast<const bf16x8*>(bufp + ((q * 64 + ((((q >> 2) & 3) ^ h) << 4)) ^ (ks << 5)));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bv[j] = *reinterpret_cast<const bf16x8*>(bufp + (bbase ^ (ks << 5)) + tap * 4096 + j * 2048);
      };
      auto frag_mma = [&](const bf16x8 (&av)[2], const bf16x8 (&bv)[NT]) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m], bv[j], acc[m][j], 0, 0, 0);
      };
      // `frag_ready` touches the current fragments: hipcc puts its `s_waitcnt lgkmcnt(0)` THERE, i.e. before the next
      // step's reads are issued, so the wait covers only reads that already had a whole MFMA group to complete
      // (left to itself it drains the reads it has just issued in front of every MFMA group)
      auto frag_ready = [&](const bf16x8 (&av)[2], const bf16x8 (&bv)[NT]) {
        asm volatile("" ::"v"(av[0]), "v"(av[1]), "v"(bv[0]), "v"(bv[1]));
      };
      frag_load(0, fa[0], fb[0]);
#pragma unroll
      for (int t = 0; t < 18; t += 2) {
        frag_ready(fa[0], fb[0]);
        frag_load(t + 1, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);   // reads first: they then have the whole MFMA group to complete
        frag_mma(fa[0], fb[0]);
        if (more && t < 2 * DM_PPW) stage_piece(t, nxt);          // the matrix pipe works on while the wave issues it
        __builtin_amdgcn_sched_barrier(0);
        frag_ready(fa[1], fb[1]);
        if (t + 2 < 18) frag_load(t + 2, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        frag_mma(fa[1], fb[1]);
        if (more && t + 1 < 2 * DM_PPW) stage_piece(t + 1, nxt);
        if (more && t + 1 == 2 * DM_PPW - 1) stage_advance();
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (TF) {
        if (more) stage_write(nxt);
      }
    }

    // ---- epilogue (the arithmetic of conv_fwd_bf16_kernel): D col = lane&31 (channel), row = (i&3) + 8*(i>>2) + 4*h.
    // The buffer just multiplied from (`cur`) is the staging image; the other one is receiving the next step's operands.
    // The tile's two 256-pixel halves (waves 0-3 / 4-7) pass through it one after the other, all 512 threads storing.
    __bf16* st16 = reinterpret_cast<__bf16*>(lds + cur);
    float* st32 = reinterpret_cast<float*>(lds + cur);
    if (!pstats) {
#pragma unroll
      for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
    }
    const bool second = a.cout_split > 0 && n0 >= a.cout_split;
    const int ld_all = a.cout_split > 0 ? (second ? a.Cout - a.cout_split : a.cout_split) : a.Cout;
    __bf16* outp = second ? a.out1 : a.out;
    const int nn0 = second ? n0 - a.cout_split : n0;
    // a split data gradient joins only into its first output (host: EPI 2 <=> accumulate)
    const bool join = JOIN && !second;
    float b_mu[8], b_is[8], b_sc[8], b_sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) b_mu[k] = b_is[k] = b_sc[k] = b_sh[k] = 0.f;
    if (!pstats) {
#pragma unroll
      for (int k = 0; k < 8; ++k) q1[k] = q2[k] = 0.f;
    }
    if (bnb) {
      auto ld8 = [&](const float* p, float (&v)[8]) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p + n0 + 8 * sg), hi = *reinterpret_cast<const f32x4*>(p + n0 + 8 * sg + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k] = lo[k];
          v[4 + k] = hi[k];
        }
      };
      ld8(a.bnb.mean, b_mu);
      ld8(a.bnb.invstd, b_is);
      if constexpr (!JOIN) {
        ld8(a.bnb.act_scale, b_sc);
        ld8(a.bnb.act_shift, b_sh);
      }
    }
    if (!bnb && !JOIN && a.stats != nullptr) {   // BatchNorm statistics from the fp32 accumulators
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int mrow = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int pt = (wave * 2 + m2) * 32 + mrow;
            const float v = acc[m2][j][i];
            if (oy0 + pt / DM_TW < a.Ho && ox0 + pt % DM_TW < a.Wo) {
              s1[j] += v;
              s2[j] += v * v;
            }
          }
    }
#pragma unroll 1
    for (int hh = 0; hh < 2; ++hh) {
      const int pbase = hh * 256;
      __syncthreads();   // the staging image is free: operand reads / the other half's stores are done
      if (hf == hh) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int mrow = (i & 3) + 8 * (i >> 2) + 4 * h;
              const int pl = (lw * 2 + m2) * 32 + mrow;
              if (join) st32[pl * OUTF_PITCH + 32 * j + r] = acc[m2][j][i];
              else st16[pl * OUT_PITCH + 32 * j + r] = (__bf16)acc[m2][j][i];
            }
      }
      __syncthreads();
      if (!JOIN || !join) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int pl = prow + it * PER_IT;
          const int oy = oy0 + (pbase + pl) / DM_TW, ox = ox0 + (pbase + pl) % DM_TW;
          if (oy < a.Ho && ox < a.Wo) {
            const size_t o = (((size_t)b * a.Ho + oy) * a.Wo + ox) * ld_all + nn0 + 8 * sg;
            const f32x4 raw = *reinterpret_cast<const f32x4*>(st16 + pl * OUT_PITCH + 8 * sg);
            *reinterpret_cast<f32x4*>(outp + o) = raw;
            if constexpr (bnb && !JOIN) {   // virtual activation — the arithmetic of bn_bwd_reduce_bf16_kernel
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
      } else if constexpr (JOIN) {
        f32x4 prev[ITS];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int pl = prow + it * PER_IT;
          const int oy = oy0 + (pbase + pl) / DM_TW, ox = ox0 + (pbase + pl) % DM_TW;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (oy < a.Ho && ox < a.Wo)
            v = *reinterpret_cast<const f32x4*>(outp + (((size_t)b * a.Ho + oy) * a.Wo + ox) * ld_all + nn0 + 8 * sg);
          prev[it] = v;
        }
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int pl = prow + it * PER_IT;
          const int oy = oy0 + (pbase + pl) / DM_TW, ox = ox0 + (pbase + pl) % DM_TW;
          if (oy < a.Ho && ox < a.Wo) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(st32 + pl * OUTF_PITCH + 8 * sg);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(st32 + pl * OUTF_PITCH + 8 * sg + 4);
            bf16x8 o = *reinterpret_cast<bf16x8*>(&prev[it]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              o[k] = (__bf16)(lo[k] + (float)o[k]);
              o[4 + k] = (__bf16)(hi[k] + (float)o[4 + k]);
            }
            const size_t oo = (((size_t)b * a.Ho + oy) * a.Wo + ox) * ld_all + nn0 + 8 * sg;
            *reinterpret_cast<bf16x8*>(outp + oo) = o;
            if constexpr (bnb && JOIN) {   // sums over the joined (rounded) gradient, mask from the stored activation
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
    if (pstats) {
      // sums stay in registers until the workgroup's last tile
    } else if (bnb) {
      // per-thread sums over its pixels of 8 channels -> per-channel sums over the 64 threads of a channel segment
      __syncthreads();
      float* qs = reinterpret_cast<float*>(lds + cur);   // [2][8][512]
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        qs[k * 512 + tid] = q1[k];
        qs[(8 + k) * 512 + tid] = q2[k];
      }
      __syncthreads();
      if (tid < 2 * TN) {
        const int which = tid / TN, c = tid % TN;
        const float* col = qs + (which * 8 + (c & 7)) * 512 + (c >> 3);
        float sum = 0.f;
        for (int i = 0; i < PER_IT; ++i) sum += col[i * SEGS];   // fixed order
        a.stats[((size_t)which * a.P + sp) * a.Cout + n0 + c] = sum;
      }
    } else if (!JOIN && a.stats != nullptr) {
      __syncthreads();
      float* red = reinterpret_cast<float*>(lds + cur);  // [2][8 waves][TN]
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
        const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
        if (h == 0) {
          red[wave * TN + 32 * j + r] = t1;
          red[8 * TN + wave * TN + 32 * j + r] = t2;
        }
      }
      __syncthreads();
      if (tid < 2 * TN) {
        const int which = tid / TN, c = tid % TN;
        const float* rr = red + which * 8 * TN + c;
        a.stats[((size_t)which * a.P + sp) * a.Cout + n0 + c] =
            ((rr[0] + rr[TN]) + (rr[2 * TN] + rr[3 * TN])) + ((rr[4 * TN] + rr[5 * TN]) + (rr[6 * TN] + rr[7 * TN]));
      }
    }
    // the next step's barrier orders these LDS reads before the DMA that will overwrite this buffer
  }
  if (pstats && a.stats != nullptr && (bnb || !JOIN)) {
    // one row per workgroup (same arithmetic as the per-tile forms above); the other channel blocks of the row are zeros
    const int nt0 = (int)xcd_remap(blockIdx.x, (unsigned)total_tiles) % a.n_tiles;
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();   // every wave is done with the operand / staging images (no DMA is in flight after the last step)
    if (bnb) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[k * 512 + tid] = q1[k];
        red[(8 + k) * 512 + tid] = q2[k];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
        const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
        if (h == 0) {
          red[wave * TN + 32 * j + r] = t1;
          red[8 * TN + wave * TN + 32 * j + r] = t2;
        }
      }
    }
    __syncthreads();
    if (tid < 2 * TN) {
      const int which = tid / TN, c = tid % TN;
      float sum;
      if (bnb) {
        const float* col = red + (which * 8 + (c & 7)) * 512 + (c >> 3);
        sum = 0.f;
        for (int i = 0; i < PER_IT; ++i) sum += col[i * SEGS];   // fixed order
      } else {
        const float* rr = red + which * 8 * TN + c;
        sum = ((rr[0] + rr[TN]) + (rr[2 * TN] + rr[3 * TN])) + ((rr[4 * TN] + rr[5 * TN]) + (rr[6 * TN] + rr[7 * TN]));
      }
      float* row = a.stats + ((size_t)which * a.P + blockIdx.x) * a.Cout;
      for (int cb = 0; cb < a.n_tiles; ++cb) row[TN * cb + c] = cb == nt0 ? sum : 0.f;
    }
  }
}

// Tried and removed (round 2): a wave-specialised form of this kernel — waves 0-3 only multiply (4x2 accumulator tiles,
// input rows reused across the 3 kernel rows: 0.5 LDS fragment reads per MFMA), waves 4-7 only move data and run the
// previous tile's epilogue beside the next tile's MFMAs.  MEASURED (same box, B=64, scripts/bench_conv.py): 64->64
// @128^2 404 vs 667 TFLOP/s for the symmetric kernel above, 128->128 @64^2 608 vs 796, 256->256 @32^2 817 vs 915,
// 768->256 @32^2 1118 vs 1113: with few chunks per tile the producers' chain drop -> flush -> store drain -> stage at a
// tile boundary (~6 us) is longer than the MFMA step it should hide behind (2.4 us), and in steady state the step time is
// the same 4.3 us — the DMA's LDS writes and the fragment reads contend for the LDS whoever issues them.  At B=64/512^2 a
// training step on it was also not run-to-run bit-identical (tests/test_fullsize_gpu.py), so it is not kept as an option.

// DT_BF16_DMA in the environment: 0 keeps every layer on the register-staged kernels (A/B measurements), 2 uses this
// kernel wherever its shape conditions hold (tests on small batches); default 1: only where its tiles fill the chip
static int g_dm_mode = -1;
static int dm_mode() {
  if (g_dm_mode < 0) {
    const char* e = getenv("DT_BF16_DMA");
    g_dm_mode = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
  }
  return g_dm_mode;
}

extern "C" int dt_set_option(const char* name, int value) {
  DT_REQUIRE(name != nullptr, "set_option: null name");
  if (strcmp(name, "bf16_dma") == 0) {
    DT_REQUIRE(value >= 0 && value <= 2, "set_option: bf16_dma takes 0 (off), 1 (auto) or 2 (wherever the shape allows)");
    g_dm_mode = value;
    return DT_OK;
  }
  dt_set_error("set_option: unknown option '%s'", name);
  return DT_EINVAL;
}

int dt_conv_bf16_dma_supported(const dt_conv_desc* d) {
  const int mode = dm_mode();
  if (mode == 0) return 0;
  if (d->ksize != 3 || d->stride != 1 || d->pad != 1) return 0;
  if ((d->C0 % 32) != 0 || (d->C1 % 32) != 0 || (d->Cout % DM_TN) != 0 || d->C0 > DM_TF_MAXC) return 0;
  if (d->cout_split != 0 && (d->cout_split % DM_TN) != 0) return 0;
  if (d->Wo <= 16 || d->Ho < DM_TH) return 0;          // narrower / shorter maps: the 16- and 8-wide tiles
  if (mode == 2) return 1;
  // enough 512-pixel x 64-channel tiles to fill the chip (the register-staged kernels have smaller tiles)
  const long wgs = (long)d->B * dt_cdiv(d->Ho, DM_TH) * dt_cdiv(d->Wo, DM_TW) * (d->Cout / DM_TN);
  return wgs >= 256;
}

// rows of the partial-sum buffer: one per workgroup when every tile of a (persistent) workgroup has the same channel
// block — tile ids id, id + 256, ... through xcd_remap: 32 % n_tiles == 0 — one per spatial tile otherwise
static int dm_stat_rows(int B, int Ho, int Wo, int Cout, int* pstats) {
  const int sp = B * dt_cdiv(Ho, DM_TH) * dt_cdiv(Wo, DM_TW), nt = Cout / DM_TN;
  const long total = (long)sp * nt;
  *pstats = nt > 0 && (32 % nt) == 0;
  return *pstats ? (int)(total < DM_MAX_WGS ? total : DM_MAX_WGS) : sp;
}

int dt_conv_bf16_dma_stat_rows(const dt_conv_desc* d) {
  int ps;
  return dm_stat_rows(d->B, d->Ho, d->Wo, d->Cout, &ps);
}

int dt_conv_bf16_dma_launch(ConvBfArgs a, hipStream_t st) {
  a.tiles_x = dt_cdiv(a.Wo, DM_TW);
  a.tiles_y = dt_cdiv(a.Ho, DM_TH);
  a.n_tiles = a.Cout / DM_TN;
  const int total = a.B * a.tiles_x * a.tiles_y * a.n_tiles;
  a.P = dm_stat_rows(a.B, a.Ho, a.Wo, a.Cout, &a.pstats);   // rows (= indexing stride) of the statistics buffer
  // persistent workgroups: one per CU (157 KB of LDS each), every one walks tiles id, id + grid, ...
  const int grid = total < DM_MAX_WGS ? total : DM_MAX_WGS;
  const bool bnb = a.bnb.y != nullptr, join = a.accumulate != 0;
  DT_REQUIRE(!bnb || a.stats != nullptr, "conv_bf16_dma: fused BatchNorm-backward sums need the stats buffer");
  DT_REQUIRE(!(a.in_scale != nullptr && bnb), "conv_bf16_dma: no input transform on the BatchNorm-backward form");
  dim3 g((unsigned)grid), blk(512);
  if (a.in_scale != nullptr && join) hipLaunchKernelGGL((conv3x3_bf16_dma_kernel<true, 2>), g, blk, 0, st, a, total);
  else if (a.in_scale != nullptr) hipLaunchKernelGGL((conv3x3_bf16_dma_kernel<true, 0>), g, blk, 0, st, a, total);
  else if (!bnb && !join) hipLaunchKernelGGL((conv3x3_bf16_dma_kernel<false, 0>), g, blk, 0, st, a, total);
  else if (bnb && !join) hipLaunchKernelGGL((conv3x3_bf16_dma_kernel<false, 1>), g, blk, 0, st, a, total);
  else if (!bnb && join) hipLaunchKernelGGL((conv3x3_bf16_dma_kernel<false, 2>), g, blk, 0, st, a, total);
  else hipLaunchKernelGGL((conv3x3_bf16_dma_kernel<false, 3>), g, blk, 0, st, a, total);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
