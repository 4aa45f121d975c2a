// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950 — the 3x3 stride-1 layers of the U-Net
// (forward and data gradient; 96 % of the network's FLOPs) with 2.25x fewer multiplies than the direct form.
//
// Replaces the same ATen conv2d / convolution_backward(input) calls as conv_fwd.hip (smp.Unet(resnet34) reached from
// deadtrees/network/segmodel.py:214,235,280); cuDNN / MIOpen pick the same algorithm for fp32 3x3 layers.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      per 4x4 input tile d -> 2x2 output tile Y  (Lavin & Gray 2016)
//
// Design (MI355X-first):
//   * persistent workgroups, one per CU (4 waves, one per SIMD, 512 registers each; 148 KB of LDS); a tile = 16x16
//     output pixels = 64 Winograd tiles x 64 output channels; a wave owns 32 tiles x 32 channels and keeps all 16
//     transform-domain positions of them in 256 accumulator registers (16 independent 32x32 MFMA tiles);
//   * the 16 element-wise products are 16 GEMMs over input channels on v_mfma_f32_32x32x2_f32;
//   * per 8-channel chunk: every thread loads the 4x4 patch of one tile for 2 channels with 16 buffer loads (padding =
//     an out-of-range offset -> 0: no branches; upsample / concat are index arithmetic, the producer's BatchNorm+ReLU 5
//     VALU ops per pixel pair), transforms it in registers (32 packed adds) and writes the 16 positions to LDS
//     `[pos][k-half][tile][4]`, so a wave's A operand of one position for the WHOLE chunk is one conflict-free
//     ds_read_b128;
//   * the transformed weights U = G g G^T are produced once per step for all layers (dt_winograd_weight_images) in
//     exactly the LDS image order `[pos][chunk][k-half][Cout][4]` and go global -> LDS by DMA (buffer_load ... lds);
//   * both LDS images are double-buffered: one barrier per chunk; everything but the MFMAs is issued in fixed slots
//     between the 64 MFMAs of a chunk: weight DMAs of chunk c+1, patch loads of chunk c+2 (two register sets, across
//     tile boundaries), BatchNorm / transform / LDS writes of chunk c+1;
//   * output transform (24 adds per tile and channel) is lane-local on the accumulators; epilogue = buffer stores with
//     scalar offsets (128-B rows) + per-channel sum / sum-of-squares (or BatchNorm-backward) partials, kept in
//     registers over the workgroup's tiles where every tile of a workgroup has the same channel block.
// Measured (B=32, scripts/bench_wino.py, TF/s of direct-convolution FLOPs, vs conv_fwd.hip): 64->64@128^2 167 vs 103,
// 128->128@64^2 190 vs 110, 256->256@32^2 204 vs 114, 512->512@16^2 209 vs 110, 768->256@32^2 240 vs 127; matrix-pipe
// utilisation 57 % (PMC) on 16/36 of the multiplies.  What limits it and what was tried: DESIGN.md section 5.
#include "common.h"
#include "conv_wino.h"

#include <type_traits>

typedef const void __attribute__((address_space(1)))* wn_gptr;
typedef void __attribute__((address_space(3)))* wn_lptr;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define WN_PS 288                      // floats per (position, k-half) plane: 64 x 4 + 32 (bank skew for the V writes)
#define WN_BUF (32 * WN_PS)            // one operand image of one chunk: 36,864 B
#define WN_TF_MAXC 512
#define WN_MAX_WGS 256                 // CUs of an MI355X
#define WN_OOB 0x80000000u             // byte offset beyond every buffer this kernel takes (host check: < 2 GiB)

struct WinoArgs {
  const float* src0;
  const float* src1;
  const float* u;
  const float* in_scale;
  const float* in_shift;
  float* out0;
  float* out1;
  float* stats;
  dt_bn_bwd_fuse bnb;
  int B, Hin, Win, C0, C1, mode0;
  int Cout, cout_split, accumulate;
  int tiles_x, tiles_y, n_tiles, P, nchunks;
  int nt0;                   // first 64-channel block of this launch (n_tiles counts the blocks it covers); 0 except for the
                             // two launches of dt_conv2d_winograd_upsampled_dgrad
  int stat_ld;               // row length of the statistics buffer (Cout; the up-sampled part's width in EPI 6)
  int pstats;                // 1: BatchNorm partial sums accumulated over the workgroup's tiles, ONE row per workgroup
  int pack;                  // 1: maps of at most 8x8 pixels — FOUR images share a 16x16-pixel tile (one per 8x8 quadrant)
  unsigned bytes0, bytes1;   // sizes of the two sources (buffer descriptors: out-of-range loads return 0)
  unsigned obytes0, obytes1; // sizes of the two outputs
  unsigned ubytes;           // size of the transformed weights
};

// compile-time loop: the slot schedule below indexes register arrays (accumulators, patch pixels) with k — an ordinary
// loop that hipcc declines to unroll completely would push them to scratch memory
template <int K, int N, class F>
__device__ __forceinline__ void wn_static_for(F&& f) {
  if constexpr (K < N) {
    f(std::integral_constant<int, K>{});
    wn_static_for<K + 1, N>(f);
  }
}

__device__ __forceinline__ void wn_dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((wn_gptr)g, (wn_lptr)l, 16, 0, 0);
}

// EPI: 0 store (+ BatchNorm statistics, split outputs); 1 store + fused BatchNorm-backward sums (virtual activation);
//      2 gradient join (out0 += ...); 3 join + BatchNorm-backward sums (stored activation);
//      6 data gradient of a convolution whose first cout_split input channels were a nearest x2 up-sampling (decoder conv1):
//        the 2x2 outputs of a Winograd tile ARE one source pixel's four gradients — their sum (a + b) + (c + d) is stored at
//        half resolution [B, Hin/2, Win/2, cout_split] with the BatchNorm-backward sums of the layer below (virtual
//        activation): a quarter of the stores, no full-resolution gradient, no dt_upsample2x_bwd_bn pass
//      4 inference: out = relu(conv * scale + shift) (eval-mode BatchNorm + ReLU applied to the accumulators: the raw
//        output is never stored and no elementwise pass follows); 5 the same with a residual: relu(conv * scale + shift + res)
//
// Persistent workgroups (one per CU): workgroup g walks the tiles xcd_remap(g + round * grid) — the n-tiles of one
// spatial tile run at the same time on one XCD and share its L2 — as ONE sequence of steps (tile, 8-channel chunk).
// During step s (64 MFMAs of 64 cycles per wave; everything else is issued in their shadow, a few instructions after
// each one) the wave also issues: the weight-plane DMAs of step s+1, the 16 patch-pixel loads of step s+2 (two register
// sets: a full step of latency cover, across tile boundaries too), and the BatchNorm+ReLU / B^T d B transform / LDS
// writes of step s+1.  One barrier per step; at a tile's last step the epilogue runs with the next tile's operands
// already in flight.
// PACK (compile time: the 16x16-map path pays nothing for it): see WinoArgs::pack
template <bool TF, int EPI, bool PACK = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_kernel(const WinoArgs a, const int total_tiles) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * WN_BUF + 256 + (TF ? 2 * WN_TF_MAXC : 0)];
  float* Vb = lds;
  float* Ub = lds + 2 * WN_BUF;
  float* red = lds + 4 * WN_BUF;     // [2][2 m-waves][64] BatchNorm partial sums of a tile
  float* lds_tf = red + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (TF) {
    for (int i = tid; i < a.C0; i += 256) {
      lds_tf[i] = a.in_scale[i];
      lds_tf[WN_TF_MAXC + i] = a.in_shift[i];
    }
  }
  const int wm = wave >> 1, wn = wave & 1, kh = lane >> 5, r = lane & 31;
  const int q = tid & 3, wt = tid >> 2;     // staging role: Winograd tile wt (8 x 8 per workgroup), channel pair q
  const int nch = a.nchunks;
  const int my_tiles = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  auto tile_of_round = [&](int round) { return (int)xcd_remap(blockIdx.x + (unsigned)round * gridDim.x, (unsigned)total_tiles); };

  // ---- load context (runs two steps ahead): byte offsets of the 16 patch pixels in source 0 / source 1; WN_OOB
  // (beyond num_records of the buffer descriptor -> the load returns 0) for padding and past the last tile.
  // One buffer_load_dwordx2 per pixel and chunk: no branches, no 64-bit address arithmetic.
  unsigned off0[16], off1[16];
  unsigned long long l_rmask[4] = {0, 0, 0, 0}, l_cmask[4] = {0, 0, 0, 0};
  int l_round = 0, l_chunk = 0;
  auto setup_offsets = [&](int round) {
    const bool live = round < my_tiles;
    const int tile = live ? tile_of_round(round) : 0;
    const int sp = tile / a.n_tiles;
    int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / tiles_per_img;
    int iy = 16 * ty - 1 + 2 * (wt >> 3), ix = 16 * tx - 1 + 2 * (wt & 7);
    bool img = live;
    if constexpr (PACK) {   // quadrant (wt >> 5, (wt >> 2) & 1) of the 8x8 grid of Winograd tiles = image 4 sp + 2 qy + qx
      b = 4 * sp + 2 * (wt >> 5) + ((wt >> 2) & 1);
      iy = -1 + 2 * ((wt >> 3) & 3);
      ix = -1 + 2 * (wt & 3);
      img = live && b < a.B;
    }
    const int Hs0 = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws0 = a.mode0 ? (a.Win >> 1) : a.Win;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = iy + i, x = ix + j;
        const bool ok = img && (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
        const int p0 = (b * Hs0 + (a.mode0 ? (y >> 1) : y)) * Ws0 + (a.mode0 ? (x >> 1) : x);
        const int p1 = (b * a.Hin + y) * a.Win + x;
        off0[4 * i + j] = ok ? (unsigned)(p0 * a.C0 + 2 * q) * 4u : WN_OOB;
        off1[4 * i + j] = ok ? (unsigned)(p1 * a.C1 + 2 * q) * 4u : WN_OOB;
      }
    if constexpr (TF) {   // validity of the patch rows / columns as lane masks (SGPR pairs): pixel (i,j) = row i & column j
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        l_rmask[i] = __builtin_amdgcn_ballot_w64(img && (unsigned)(iy + i) < (unsigned)a.Hin);
        l_cmask[i] = __builtin_amdgcn_ballot_w64((unsigned)(ix + i) < (unsigned)a.Win);
      }
    }
  };
  auto advance_loads = [&]() {
    if (++l_chunk == nch) {
      l_chunk = 0;
      setup_offsets(++l_round);
    }
  };
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, a.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0, a.bytes1, 0x00020000);
  f32x2 d[2][16], t[16];
  int dc0[2] = {0, 0};               // first channel of the chunk each register set holds
  auto load_pixel = [&](auto set_tag, int i) {
    constexpr int SET = decltype(set_tag)::value;
    const int c0 = __builtin_amdgcn_readfirstlane(8 * l_chunk);
    const bool use0 = c0 < a.C0;   // wave-uniform
    const unsigned cb = (unsigned)(use0 ? c0 : c0 - a.C0) * 4u;
    const u32x2 v = use0 ? __builtin_amdgcn_raw_buffer_load_b64(rs0, off0[i] + cb, 0, 0)
                         : __builtin_amdgcn_raw_buffer_load_b64(rs1, off1[i] + cb, 0, 0);
    d[SET][i] = __builtin_bit_cast(f32x2, v);
    if (i == 0) dc0[SET] = c0;
  };
  // the producer's BatchNorm-apply + ReLU on the real pixels of source 0 (zero padding stays zero: a padded pixel was
  // loaded as exactly 0 and is recognised by the validity bit taken when the set was loaded)
  // 5 VALU instructions per pixel pair (with one wave per SIMD the staging instructions are not free behind the MFMAs:
  // halving them from 10 was worth 2-3 % of the kernel): v_pk_fma_f32, two v_max_f32, two v_cndmask_b32 with the pixel's validity lane mask straight from SGPRs
  // (row mask & column mask of the tile the set was loaded for: scalar ops).  The fused multiply-add differs from the
  // mul + add of bn_act / the direct kernels by at most one rounding (6e-8 relative; the Winograd transform itself 1e-6).
  unsigned long long drm[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, dcm[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x2 tf_sc = {1.f, 1.f}, tf_sh = {0.f, 0.f}, tf_v[2];
  float tf_lo = 0.f;
  // branch-free: a chunk of source 1 (no transform) runs with scale 1, shift 0 and a ReLU floor of -inf
  auto tf_begin = [&](auto set_tag) {
    if constexpr (TF) {
      constexpr int SET = decltype(set_tag)::value;
      const int c0 = __builtin_amdgcn_readfirstlane(dc0[SET]);
      const bool on = c0 < a.C0;
      const int tc = on ? c0 : 0;
      const f32x2 sc = *reinterpret_cast<const f32x2*>(lds_tf + tc + 2 * q);
      const f32x2 sh = *reinterpret_cast<const f32x2*>(lds_tf + WN_TF_MAXC + tc + 2 * q);
      tf_sc[0] = on ? sc[0] : 1.f;
      tf_sc[1] = on ? sc[1] : 1.f;
      tf_sh[0] = on ? sh[0] : 0.f;
      tf_sh[1] = on ? sh[1] : 0.f;
      tf_lo = on ? 0.f : -__builtin_inff();
    }
  };
  // two half-steps per pixel, issued one MFMA apart (a single wave per SIMD: dependent VALU chains need the distance)
  auto tf_a = [&](auto set_tag, int i) {
    if constexpr (TF) {
      constexpr int SET = decltype(set_tag)::value;
      tf_v[i & 1] = __builtin_elementwise_fma(d[SET][i], tf_sc, tf_sh);
    }
  };
  auto tf_b = [&](auto set_tag, int i) {
    if constexpr (TF) {
      constexpr int SET = decltype(set_tag)::value;
      const f32x2 v = tf_v[i & 1];
      // ReLU that keeps NaN like torch.relu / bn_act (v_max would return the non-NaN operand): keep v where the pixel is
      // real AND NOT (v < floor) — an ordered compare, false for NaN — else 0 (a padded pixel was loaded as exactly 0 and
      // stays 0; a source-1 chunk has the floor -inf: nothing is below it).  v_cmp + v_cndmask per element: the VALU count
      // of the v_max form; the mask algebra runs on the scalar unit.
      const unsigned long long m = drm[SET][i >> 2] & dcm[SET][i & 3];
      // !(floor > v) is true for NaN; one asm block: the keep-mask never leaves VCC
      float o0 = v[0], o1 = v[1];
      asm volatile("v_cmp_ngt_f32 vcc, %2, %0\n\ts_and_b64 vcc, vcc, %3\n\tv_cndmask_b32 %0, 0, %0, vcc\n\t"
                   "v_cmp_ngt_f32 vcc, %2, %1\n\ts_and_b64 vcc, vcc, %3\n\tv_cndmask_b32 %1, 0, %1, vcc"
                   : "+v"(o0), "+v"(o1)
                   : "s"(tf_lo), "s"(m)
                   : "vcc");
      d[SET][i][0] = o0;
      d[SET][i][1] = o1;
    }
  };
  auto mark_valid = [&](auto set_tag) {
    if constexpr (TF) {
      constexpr int SET = decltype(set_tag)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        drm[SET][i] = l_rmask[i];
        dcm[SET][i] = l_cmask[i];
      }
    }
  };
  // B^T d B in registers: column j of B^T d, then row i of (B^T d) B -> 4 positions of (tile wt, channels 2q, 2q+1)
  auto transform_col = [&](auto set_tag, int j) {
    constexpr int SET = decltype(set_tag)::value;
    t[0 + j] = d[SET][0 + j] - d[SET][8 + j];
    t[4 + j] = d[SET][4 + j] + d[SET][8 + j];
    t[8 + j] = d[SET][8 + j] - d[SET][4 + j];
    t[12 + j] = d[SET][4 + j] - d[SET][12 + j];
  };
  const int vwoff = (q >> 1) * WN_PS + wt * 4 + 2 * (q & 1);
  auto transform_row_write = [&](float* Vd, int i) {
    float* dst = Vd + vwoff;
    const f32x2 v0 = t[4 * i + 0] - t[4 * i + 2];
    const f32x2 v1 = t[4 * i + 1] + t[4 * i + 2];
    const f32x2 v2 = t[4 * i + 2] - t[4 * i + 1];
    const f32x2 v3 = t[4 * i + 1] - t[4 * i + 3];
    *reinterpret_cast<f32x2*>(dst + (4 * i + 0) * 2 * WN_PS) = v0;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 1) * 2 * WN_PS) = v1;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 2) * 2 * WN_PS) = v2;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 3) * 2 * WN_PS) = v3;
  };
  // ---- weight context (one step ahead): transformed weights of (tile, chunk): 32 planes of 1 KiB, 8 per wave, by DMA
  // (buffer_load_dwordx4 ... lds: per-lane part (n0 + lane) * 16 B in the VGPR offset, everything else scalar)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wave_m = wave_u >> 1;
  const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, a.ubytes, 0x00020000);
  int w_round = 0, w_chunk = 0;
  unsigned u_voff = 0;
  auto set_ug = [&](int round) {
    const int tile = tile_of_round(round < my_tiles ? round : 0);   // past the end: any valid address (never multiplied)
    u_voff = (unsigned)((tile % a.n_tiles + a.nt0) * 64 + lane) * 16u;
  };
  auto dma_plane = [&](float* Ud, int i) {
    const int plane = wave_u * 8 + i;
    const int pos = plane >> 1, k = plane & 1;
    const int ch = __builtin_amdgcn_readfirstlane(w_chunk);
    const int soff = ((pos * nch + ch) * 2 + k) * a.Cout * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (wn_lptr)(Ud + plane * WN_PS), 16, u_voff, soff, 0, 0);
  };
  auto advance_weights = [&]() {
    if (++w_chunk == nch) {
      w_chunk = 0;
      set_ug(++w_round);
    }
  };

  f32x16 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
  float ps1 = 0.f, ps2 = 0.f;

  const int aoff = kh * WN_PS + (32 * wm + r) * 4;
  const int boff = kh * WN_PS + (32 * wn + r) * 4;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  if constexpr (TF) __syncthreads();   // lds_tf
  // ---- prologue: step 0 staged completely, the loads of step 1 in flight
  setup_offsets(0);
  set_ug(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_plane(Ub, i);
  advance_weights();
  mark_valid(S0{});
#pragma unroll
  for (int i = 0; i < 16; ++i) load_pixel(S0{}, i);
  advance_loads();
  mark_valid(S1{});
#pragma unroll
  for (int i = 0; i < 16; ++i) load_pixel(S1{}, i);
  advance_loads();
  tf_begin(S0{});
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    tf_a(S0{}, i);
    tf_b(S0{}, i);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) transform_col(S0{}, j);
#pragma unroll
  for (int i = 0; i < 4; ++i) transform_row_write(Vb, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the weight planes of step 0

  // ---- one step.  PAR = s & 1: operand images PAR are multiplied; images PAR^1 receive step s+1; register set PAR^1
  // (loaded during step s-1) is transformed; register set PAR is loaded for step s+2.
  auto step = [&](auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    using SN = std::integral_constant<int, PAR ^ 1>;
    const float* Vc = Vb + PAR * WN_BUF + aoff;
    const float* Uc = Ub + PAR * WN_BUF + boff;
    float* Vn = Vb + (PAR ^ 1) * WN_BUF;
    float* Un = Ub + (PAR ^ 1) * WN_BUF;
    f32x4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const f32x4*>(Vc);
    fb[0] = *reinterpret_cast<const f32x4*>(Uc);
    wn_static_for<0, 64>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int p = k >> 2, j = k & 3, cur = p & 1;
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][j], fb[cur][j], acc[p], 0, 0, 0);
      if (j == 0 && p + 1 < 16) {
        fa[cur ^ 1] = *reinterpret_cast<const f32x4*>(Vc + (p + 1) * 2 * WN_PS);
        fb[cur ^ 1] = *reinterpret_cast<const f32x4*>(Uc + (p + 1) * 2 * WN_PS);
      }
      if (j != 0 && k >= 22 && k < 34) {      // slots 22..31: the 8 weight planes of step s+1
        constexpr int wi = 3 * p + (j - 1) - 16;   // k = 22, 23, 25, 26, 27, 29, 30, 31 -> 0..7
        if (wi >= 0 && wi < 8) dma_plane(Un, wi);
      }
      if (j != 0 && k < 22) {                 // slots 1..21: the 16 patch pixels of step s+2 (issued first: they come
        constexpr int li = 3 * p + (j - 1);   // from HBM, the weights from L2)
        if (li == 0) mark_valid(par_tag);
        if (li < 16) load_pixel(par_tag, li);
      }
      if (k == 34) tf_begin(SN{});
      if (k >= 35 && k < 51) tf_a(SN{}, k - 35);
      if (k >= 36 && k < 52) tf_b(SN{}, k - 36);
      if (k >= 52 && k < 56) transform_col(SN{}, k - 52);
      if (k >= 56 && k < 60) transform_row_write(Vn, k - 56);
      if (k == 60) {
        advance_weights();
        advance_loads();
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // this wave's weight planes of step s+1 (issued >= 32 MFMAs ago) and patch pixels of step s+2 (>= 42 MFMAs ago)
    // have landed.  A plain vmcnt(0): a counted wait would have to know how many spill accesses hipcc put in between.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- epilogue of one tile: output transform A^T M A (lane-local on the accumulators), stores, BatchNorm sums
  auto epilogue = [&](int round) {
    // everything derived from the tile index is wave-uniform; say so, or hipcc wraps each of the 64 buffer stores in a
    // waterfall loop (readfirstlane / compare / saveexec per store: the scalar-offset operand must be provably uniform)
    const int tile = __builtin_amdgcn_readfirstlane(tile_of_round(round));
    const int nt = tile % a.n_tiles + a.nt0, sp = tile / a.n_tiles;
    const int tx = sp % a.tiles_x, ty = (sp / a.tiles_x) % a.tiles_y, b = sp / tiles_per_img;
    const int oy0 = PACK ? 0 : 16 * ty, ox0 = PACK ? 0 : 16 * tx, n0 = 64 * nt;
    const int n = n0 + 32 * wn + r;
    float* outp = a.out0;
    int ld = a.Cout, nn = n;
    if (a.cout_split > 0) {
      if (n0 >= a.cout_split) {
        outp = a.out1; ld = a.Cout - a.cout_split; nn = n - a.cout_split;
      } else {
        ld = a.cout_split;
      }
    }
    float s1 = 0.f, s2 = 0.f;
    float b_mu = 0.f, b_is = 0.f, b_sc = 0.f, b_sh = 0.f;
    if constexpr (EPI == 1 || EPI == 3 || EPI == 6) {
      if (EPI != 6 || n0 < a.cout_split) {   // (form 6: the skip's tiles have no BatchNorm layer behind them)
        b_mu = a.bnb.mean[n];
        b_is = a.bnb.invstd[n];
        if constexpr (EPI == 1 || EPI == 6) {
          b_sc = a.bnb.act_scale[n];
          b_sh = a.bnb.act_shift[n];
        }
      }
    }
    if constexpr (EPI == 4 || EPI == 5) {
      b_sc = a.bnb.act_scale[n];
      b_sh = a.bnb.act_shift[n];
    }
    const bool second = outp == a.out1 && a.cout_split > 0;
    const bool join = (EPI == 2 || EPI == 3) && !second;   // split data gradients accumulate into out0 only
    // 32-bit byte offsets + buffer descriptors: a pixel outside the map gets WN_OOB -> its loads return 0 and its store
    // is dropped by the range check (no branches); y / act of the fused BatchNorm-backward forms share out0's layout
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, second ? a.obytes1 : a.obytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bnb.y ? a.bnb.y : outp), 0, a.obytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bnb.act ? a.bnb.act : outp), 0, a.obytes0, 0x00020000);
    // addressing: byte offset = [tile base + (row, column) of the output inside the wave's block: scalar, in the
    // buffer op's SGPR offset] + [the lane's own part: 1 VGPR]; a pixel outside the map gets WN_OOB in the VGPR part
    const int ld4 = __builtin_amdgcn_readfirstlane(ld * 4), row4 = __builtin_amdgcn_readfirstlane(a.Win * ld * 4);
    // packed small maps: the wave's 4 x 8 Winograd tiles are the quadrants (wm, 0 | 1) = images 4 sp + 2 wm + kh, each from
    // its pixel (0, 0); otherwise rows 8 wm .. 8 wm + 7 of the tile, columns 8 kh ..
    const int img_px4 = __builtin_amdgcn_readfirstlane(a.Hin * a.Win * ld * 4);
    const int tile_base = __builtin_amdgcn_readfirstlane(
        PACK ? (4 * sp + 2 * wave_m) * img_px4 : ((b * a.Hin + oy0 + 8 * wave_m) * a.Win + ox0) * ld * 4);
    const unsigned lane_base = (unsigned)((PACK ? kh * img_px4 : 8 * kh * ld4) + nn * 4);
    const int xlane = PACK ? 0 : ox0 + 8 * kh, ybase = PACK ? 0 : oy0 + 8 * wave_m;
    const bool img_ok = !PACK || 4 * sp + 2 * wave_m + kh < a.B;
    if (EPI == 6 && !second) {
      // half-resolution addressing: source pixel (oy0 / 2 + 4 wave_m + (i >> 2), ox0 / 2 + 4 kh + (i & 3)), channel nn of
      // cout_split; y of the fused sums has the same layout.  Out-of-range pixels: WN_OOB (loads 0, store dropped).
      const int Hs = a.Hin >> 1, Ws = a.Win >> 1;
      const int ldl4 = __builtin_amdgcn_readfirstlane(ld * 4), rowl4 = __builtin_amdgcn_readfirstlane(Ws * ld * 4);
      const int lbase = __builtin_amdgcn_readfirstlane(((b * Hs + (oy0 >> 1) + 4 * wave_m) * Ws + (ox0 >> 1)) * ld * 4);
      const unsigned llane = (unsigned)(4 * kh * ldl4 + nn * 4);
      const unsigned lbytes = (unsigned)((size_t)a.B * Hs * Ws * ld * 4);
      const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, lbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsyl = __builtin_amdgcn_make_buffer_rsrc((void*)a.bnb.y, 0, lbytes, 0x00020000);
      const int ysrc = (oy0 >> 1) + 4 * wave_m, xsrc = (ox0 >> 1) + 4 * kh;
      // the 16 raw outputs y of the tile's source pixels are requested up front: the first quarter's output transform runs
      // under their latency (loaded per quarter, the epilogue of the up-sampled half was slower than the plain half's)
      float yall[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool ok = ysrc + (i >> 2) < Hs && xsrc + (i & 3) < Ws;
        yall[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsyl, ok ? llane : WN_OOB,
                                                                                 lbase + (i >> 2) * rowl4 + (i & 3) * ldl4, 0));
      }
#pragma unroll
      for (int qr = 0; qr < 4; ++qr) {
        unsigned off[4];
        int soff[4];
        float yv[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = 4 * qr + ii;
          const bool ok = ysrc + (i >> 2) < Hs && xsrc + (i & 3) < Ws;
          off[ii] = ok ? llane : WN_OOB;
          soff[ii] = lbase + (i >> 2) * rowl4 + (i & 3) * ldl4;
          yv[ii] = yall[i];
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = 4 * qr + ii;
          float t0[4], t1[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            t0[j] = acc[0 + j][i] + acc[4 + j][i] + acc[8 + j][i];
            t1[j] = acc[4 + j][i] - acc[8 + j][i] - acc[12 + j][i];
          }
          const float y0 = t0[0] + t0[1] + t0[2], y1 = t0[1] - t0[2] - t0[3];
          const float y2 = t1[0] + t1[1] + t1[2], y3 = t1[1] - t1[2] - t1[3];
          const float v = (y0 + y1) + (y2 + y3);                    // dt_upsample2x_bwd's order: (a + b) + (c + d)
          const bool ok = off[ii] != WN_OOB;
          const float g = (ok && (yv[ii] * b_sc + b_sh) > 0.f) ? v : 0.f;
          s1 += g;
          s2 += g * ((yv[ii] - b_mu) * b_is);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsl, off[ii], soff[ii], 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
#pragma unroll
    for (int qr = 0; qr < 4; ++qr) {
      // 4 accumulator rows (4 Winograd tiles x 4 pixels) at a time: the reads of this quarter are in flight while its
      // output transforms are computed; the scheduling fence keeps hipcc from hoisting all 256 accumulator reads
      unsigned off[16];
      int soff[16];
      float prev[(EPI == 2 || EPI == 3) ? 16 : 1], yv[(EPI == 1 || EPI == 3 || EPI == 5) ? 16 : 1], zv[(EPI == 3) ? 16 : 1];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * qr + ii;   // accumulator row m = (i & 3) + 8 (i >> 2) + 4 kh: tile row i >> 2, tile column (i & 3) + 4 kh
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int py = 2 * (i >> 2) + (e >> 1), px = 2 * (i & 3) + (e & 1);
          const bool ok = img_ok && ybase + py < a.Hin && xlane + px < a.Win;
          off[4 * ii + e] = ok ? lane_base : WN_OOB;
          soff[4 * ii + e] = tile_base + py * row4 + px * ld4;
        }
      }
      if constexpr (EPI == 2 || EPI == 3) {
#pragma unroll
        for (int x = 0; x < 16; ++x)
          prev[x] = join ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rso, off[x], soff[x], 0)) : 0.f;
      }
      if constexpr (EPI == 1 || EPI == 3 || EPI == 5) {   // EPI 5: bnb.y is the residual tensor (layout of out0)
#pragma unroll
        for (int x = 0; x < 16; ++x) yv[x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsy, off[x], soff[x], 0));
      }
      if constexpr (EPI == 3) {
#pragma unroll
        for (int x = 0; x < 16; ++x) zv[x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsz, off[x], soff[x], 0));
      }
      float y[16];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * qr + ii;
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t0[j] = acc[0 + j][i] + acc[4 + j][i] + acc[8 + j][i];
          t1[j] = acc[4 + j][i] - acc[8 + j][i] - acc[12 + j][i];
        }
        y[4 * ii + 0] = t0[0] + t0[1] + t0[2];
        y[4 * ii + 1] = t0[1] - t0[2] - t0[3];
        y[4 * ii + 2] = t1[0] + t1[1] + t1[2];
        y[4 * ii + 3] = t1[1] - t1[2] - t1[3];
      }
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const bool ok = off[x] != WN_OOB;
        float v = y[x];
        if constexpr (EPI == 2 || EPI == 3) v += prev[x];
        if constexpr (EPI == 4 || EPI == 5) {   // the arithmetic of bn_act_kernel: mul, add (, + residual), ReLU that keeps NaN
          v = v * b_sc + b_sh;
          if constexpr (EPI == 5) v += yv[x];
          v = v < 0.f ? 0.f : v;
        }
        if constexpr (EPI == 3) {
          const float g = (ok && zv[x] > 0.f) ? v : 0.f;
          s1 += g;
          s2 += g * ((yv[x] - b_mu) * b_is);
        } else if constexpr (EPI == 1) {
          const float g = (ok && (yv[x] * b_sc + b_sh) > 0.f) ? v : 0.f;
          s1 += g;
          s2 += g * ((yv[x] - b_mu) * b_is);
        } else if constexpr (EPI == 0) {
          const float g = ok ? v : 0.f;
          s1 += g;
          s2 += g * g;
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rso, off[x], soff[x], 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
    if (a.stats != nullptr && a.pstats) {
      ps1 += s1;
      ps2 += s2;
    } else if (a.stats != nullptr && !(EPI == 6 && second)) {   // (form 6: the skip's tiles carry no sums)
      const float u1 = s1 + __shfl_xor(s1, 32, 64);
      const float u2 = s2 + __shfl_xor(s2, 32, 64);
      if (kh == 0) {
        red[wm * 64 + 32 * wn + r] = u1;
        red[128 + wm * 64 + 32 * wn + r] = u2;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        a.stats[((size_t)which * a.P + sp) * a.stat_ld + n0 + c] = red[which * 128 + c] + red[which * 128 + 64 + c];
      }
      // the next step's barrier orders these reads of `red` before the next tile's writes
    }
  };

  // Structured as tiles x chunk pairs (the chunk count is even: host check), not as one flat step loop with a
  // conditional epilogue: across a two-way join hipcc no longer keeps the 256 accumulators in place (hundreds of spills).
  // The staging contexts above run ahead across the tile boundaries regardless.
  for (int round = 0; round < my_tiles; ++round) {
    for (int c = 0; c < nch; c += 2) {
      // every wave has waited for its own weight planes (vmcnt in step / prologue) and waits here for its own V writes
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      step(S0{});
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      step(S1{});
    }
    epilogue(round);
  }
  // one row of BatchNorm partial sums per workgroup: every tile of this workgroup has the same channel block (host
  // check: 32 % n_tiles == 0), so the sums stayed in registers; the other channel blocks of the row are written as zeros
  // (the finalize pass then reads <= 256 rows: no separate row-reduction launch)
  if (a.stats != nullptr && a.pstats) {
    const int nt0 = tile_of_round(0) % a.n_tiles;   // (pstats launches start at channel block 0: a.nt0 == 0)
    const float u1 = ps1 + __shfl_xor(ps1, 32, 64);
    const float u2 = ps2 + __shfl_xor(ps2, 32, 64);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kh == 0) {
      red[wm * 64 + 32 * wn + r] = u1;
      red[128 + wm * 64 + 32 * wn + r] = u2;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      float* row = a.stats + ((size_t)which * a.P + blockIdx.x) * a.stat_ld;
      const int stat_blocks = a.stat_ld / 64;   // (form 6: only the up-sampled part's channel blocks have sums)
      for (int cb = 0; cb < stat_blocks; ++cb)
        row[64 * cb + c] = cb == nt0 ? red[which * 128 + c] + red[which * 128 + 64 + c] : 0.f;
    }
  }
}

// ---------------------------------------------------------------- U = G g G^T in the kernel's LDS image order
// one thread per (input-channel quad, output channel): 9 x 4 weights in, 16 float4 out
__device__ __forceinline__ void wino_weights_tile(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout,
                                                  int co, int cq) {
  if (co >= Cout || 4 * cq >= Cin) return;
  f32x4 g[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int j = 0; j < 4; ++j) g[tap][j] = w[((size_t)tap * Cin + 4 * cq + j) * Cout + co];
  f32x4 t[12];   // G g: 4 x 3
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    t[0 + j] = g[0 + j];
    t[3 + j] = 0.5f * (g[0 + j] + g[3 + j] + g[6 + j]);
    t[6 + j] = 0.5f * (g[0 + j] - g[3 + j] + g[6 + j]);
    t[9 + j] = g[6 + j];
  }
  const int nchunks = Cin / 8;
  const int ch = cq >> 1, k = cq & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 o[4];
    o[0] = t[3 * i];
    o[1] = 0.5f * (t[3 * i] + t[3 * i + 1] + t[3 * i + 2]);
    o[2] = 0.5f * (t[3 * i] - t[3 * i + 1] + t[3 * i + 2]);
    o[3] = t[3 * i + 2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pos = 4 * i + j;
      *reinterpret_cast<f32x4*>(u + ((((size_t)pos * nchunks + ch) * 2 + k) * Cout + co) * 4) = o[j];
    }
  }
}

__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout) {
  wino_weights_tile(w, u, Cin, Cout, blockIdx.x * 64 + (threadIdx.x & 63), blockIdx.y * 4 + (threadIdx.x >> 6));
}

extern "C" int dt_winograd_weights(const float* w_hwio, float* u, int Cin, int Cout, void* stream) {
  DT_REQUIRE(w_hwio && u && Cin > 0 && Cout > 0 && (Cin % 8) == 0, "winograd_weights: Cin must be a multiple of 8");
  dim3 grid(dt_cdiv(Cout, 64), dt_cdiv(Cin / 4, 4));
  hipLaunchKernelGGL(wino_weights_kernel, grid, dim3(256), 0, (hipStream_t)stream, w_hwio, u, Cin, Cout);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// every eligible layer of a network in ONE launch (the per-layer kernel is ~5 us and there are ~60 images per step):
// table row l = (w_off, u_off, Cin, Cout, first_block); w_off into `weights` (the flat parameter buffer, or its
// flipped / transposed data-gradient image from dt_weight_images mode 0 with Cin / Cout swapped), u_off into `u`
__global__ void wino_weight_images_kernel(const float* __restrict__ weights, float* __restrict__ u,
                                          const int32_t* __restrict__ table, int n_layers) {
  __shared__ int32_t row[5];
  if (threadIdx.x == 0) {
    int l = 0;
    while (l + 1 < n_layers && table[(l + 1) * 5 + 4] <= (int)blockIdx.x) ++l;   // <= 64 layers: linear scan
#pragma unroll
    for (int k = 0; k < 5; ++k) row[k] = table[l * 5 + k];
  }
  __syncthreads();
  const int Cin = row[2], Cout = row[3];
  const int t = (int)blockIdx.x - row[4];
  const int cob = (Cout + 63) / 64;
  wino_weights_tile(weights + row[0], u + row[1], Cin, Cout, (t % cob) * 64 + (threadIdx.x & 63),
                    (t / cob) * 4 + (threadIdx.x >> 6));
}

extern "C" int dt_winograd_weight_images(const float* weights, float* u, const int32_t* table, int n_layers,
                                         int total_blocks, void* stream) {
  DT_REQUIRE(weights && u && table && n_layers > 0 && total_blocks > 0, "winograd_weight_images: bad args");
  hipLaunchKernelGGL(wino_weight_images_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, weights,
                     u, table, n_layers);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_conv2d_winograd_supported(const dt_conv_desc* d) {
  if (d == nullptr) return 0;
  if (d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->mode0 == 2) return 0;
  if ((d->C0 % 8) != 0 || (d->C1 % 8) != 0 || ((d->C0 + d->C1) % 16) != 0 || (d->Cout % 64) != 0 || d->C0 > WN_TF_MAXC) return 0;
  if (d->cout_split != 0 && (d->cout_split % 64) != 0) return 0;
  // 32-bit byte offsets with 0x80000000 as the out-of-range marker: every operand below 2 GiB (else the direct kernel)
  const size_t px0 = (size_t)d->B * (d->mode0 ? (d->Hin / 2) * (size_t)(d->Win / 2) : (size_t)d->Hin * d->Win);
  if (px0 * d->C0 * 4 >= 0x80000000ull || (size_t)d->B * d->Hin * d->Win * d->C1 * 4 >= 0x80000000ull) return 0;
  const size_t opx = (size_t)d->B * d->Ho * d->Wo * 4;
  if (opx * (d->cout_split ? d->cout_split : d->Cout) >= 0x80000000ull) return 0;
  if (d->cout_split && opx * (d->Cout - d->cout_split) >= 0x80000000ull) return 0;
  return 1;
}

// rows of the partial-sum buffer: one per workgroup when every tile of a workgroup has the same channel block
// (32 % n_tiles == 0 with the xcd-aware tile order), one per spatial tile otherwise
// maps of at most 8x8 pixels (layer 4 of a 256x256 tile): four images per 16x16-pixel tile instead of one with 3/4 padding
static inline int wn_pack(const dt_conv_desc* d) { return d->Hin <= 8 && d->Win <= 8; }

static int wn_stat_rows(const dt_conv_desc* d, int* pstats) {
  const int sp = wn_pack(d) ? dt_cdiv(d->B, 4) : d->B * dt_cdiv(d->Ho, 16) * dt_cdiv(d->Wo, 16), nt = d->Cout / 64;
  const long total = (long)sp * nt;
  *pstats = nt > 0 && (32 % nt) == 0;
  return *pstats ? (int)(total < WN_MAX_WGS ? total : WN_MAX_WGS) : sp;
}

extern "C" int dt_conv2d_winograd_stat_rows(const dt_conv_desc* d) {
  int ps;
  return wn_stat_rows(d, &ps);
}

int dt_conv_wino_launch(const dt_conv_desc* d, const float* src0, const float* src1, const float* u, float* out0,
                        float* out1, float* stats, const float* in_scale, const float* in_shift,
                        const dt_bn_bwd_fuse* fuse, hipStream_t st, bool affine) {
  DT_REQUIRE(dt_conv2d_winograd_supported(d), "conv_winograd: layer shape not supported");
  DT_REQUIRE(d->Ho == d->Hin && d->Wo == d->Win, "conv_winograd: 3x3 stride 1 pad 1 keeps the map size");
  WinoArgs a;
  a.bnb = fuse ? *fuse : dt_bn_bwd_fuse{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a.src0 = src0; a.src1 = src1; a.u = u; a.in_scale = in_scale; a.in_shift = in_shift;
  a.out0 = out0; a.out1 = out1; a.stats = stats;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0;
  a.Cout = d->Cout; a.cout_split = d->cout_split; a.accumulate = d->accumulate;
  a.tiles_x = dt_cdiv(d->Wo, 16);
  a.tiles_y = dt_cdiv(d->Ho, 16);
  a.n_tiles = d->Cout / 64;
  a.nt0 = 0;
  a.stat_ld = d->Cout;
  a.pack = wn_pack(d);
  a.P = a.pack ? dt_cdiv(d->B, 4) : d->B * a.tiles_x * a.tiles_y;
  const int sp_tiles = a.P;
  a.nchunks = (d->C0 + d->C1) / 8;
  const size_t px0 = (size_t)d->B * (d->mode0 ? (d->Hin / 2) * (size_t)(d->Win / 2) : (size_t)d->Hin * d->Win);
  const size_t b0 = px0 * d->C0 * 4, b1 = (size_t)d->B * d->Hin * d->Win * d->C1 * 4;
  DT_REQUIRE(b0 < 0x80000000ull && b1 < 0x80000000ull, "conv_winograd: a source of 2 GiB or more (use the direct kernel)");
  a.bytes0 = (unsigned)b0;
  a.bytes1 = (unsigned)b1;
  const size_t opx = (size_t)d->B * d->Ho * d->Wo * 4;
  const size_t ob0 = opx * (d->cout_split ? d->cout_split : d->Cout), ob1 = opx * (d->cout_split ? d->Cout - d->cout_split : 0);
  DT_REQUIRE(ob0 < 0x80000000ull && ob1 < 0x80000000ull, "conv_winograd: an output of 2 GiB or more (use the direct kernel)");
  a.obytes0 = (unsigned)ob0;
  a.ubytes = (unsigned)((size_t)16 * (d->C0 + d->C1) * d->Cout * 4);
  a.obytes1 = (unsigned)ob1;
  const bool bnb = !affine && a.bnb.y != nullptr, join = d->accumulate != 0;
  DT_REQUIRE(!bnb || stats != nullptr, "conv_winograd: fused BatchNorm-backward sums need the stats buffer");
  DT_REQUIRE(!affine || (in_scale == nullptr && !join && d->cout_split == 0 && stats == nullptr),
             "conv_winograd: the inference epilogue takes no input transform, join, split or statistics");
  DT_REQUIRE(!(in_scale != nullptr && bnb), "conv_winograd: no input transform on the BatchNorm-backward form");
  const int total = sp_tiles * a.n_tiles;
  a.P = wn_stat_rows(d, &a.pstats);       // rows of the statistics buffer (indexing stride of its two planes)
  dim3 g((unsigned)(total < WN_MAX_WGS ? total : WN_MAX_WGS)), blk(256);   // persistent: one workgroup per CU
  // epilogue form (see the kernel) x input transform x packed small maps
  const int epi = affine ? (a.bnb.y != nullptr ? 5 : 4) : (bnb ? 1 : 0) + (join ? 2 : 0);
  const bool tf = in_scale != nullptr;
  DT_REQUIRE(!tf || epi == 0 || epi == 2, "conv_winograd: the input transform goes with the plain / join epilogues only");
#define WN_LAUNCH(TFv, EPIv)                                                                              \
  do {                                                                                                    \
    if (a.pack) hipLaunchKernelGGL((conv3x3_wino_kernel<TFv, EPIv, true>), g, blk, 0, st, a, total);      \
    else hipLaunchKernelGGL((conv3x3_wino_kernel<TFv, EPIv, false>), g, blk, 0, st, a, total);            \
  } while (0)
  if (tf && epi == 2) WN_LAUNCH(true, 2);
  else if (tf) WN_LAUNCH(true, 0);
  else if (epi == 0) WN_LAUNCH(false, 0);
  else if (epi == 1) WN_LAUNCH(false, 1);
  else if (epi == 2) WN_LAUNCH(false, 2);
  else if (epi == 3) WN_LAUNCH(false, 3);
  else if (epi == 4) WN_LAUNCH(false, 4);
  else WN_LAUNCH(false, 5);
#undef WN_LAUNCH
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_conv2d_winograd(const dt_conv_desc* d, const float* src0, const float* src1, const float* u,
                                  float* out0, float* out1, float* stats, const float* in_scale,
                                  const float* in_shift, void* stream) {
  DT_REQUIRE(d && src0 && u && out0, "conv_winograd: null pointer");
  DT_REQUIRE(d->C1 == 0 || src1, "conv_winograd: src1 missing");
  DT_REQUIRE(d->cout_split == 0 || out1, "conv_winograd: out1 missing");
  DT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv_winograd: in_scale/in_shift must come together");
  return dt_conv_wino_launch(d, src0, src1, u, out0, out1, stats, in_scale, in_shift, nullptr, (hipStream_t)stream, false);
}

// inference form: out = relu(conv(src) * scale + shift [+ residual]) — eval-mode BatchNorm + ReLU (and the residual add of a
// ResNet basic block) applied to the accumulators, the ATen chain conv2d -> batch_norm(eval) -> (add) -> relu_ in one
// kernel; same arithmetic as dt_conv2d_winograd followed by dt_bn_act (bit-identical results)
extern "C" int dt_conv2d_winograd_affine(const dt_conv_desc* d, const float* src0, const float* src1, const float* u,
                                         float* out, const float* scale, const float* shift, const float* residual,
                                         void* stream) {
  DT_REQUIRE(d && src0 && u && out && scale && shift, "conv_winograd_affine: null pointer");
  DT_REQUIRE(d->C1 == 0 || src1, "conv_winograd_affine: src1 missing");
  DT_REQUIRE(d->cout_split == 0 && d->accumulate == 0, "conv_winograd_affine: no split outputs / joins");
  dt_bn_bwd_fuse f{residual, nullptr, nullptr, scale, shift, nullptr};
  return dt_conv_wino_launch(d, src0, src1, u, out, nullptr, nullptr, nullptr, nullptr, &f, (hipStream_t)stream, true);
}

// Data gradient of a decoder conv1 — y = conv3x3(cat(nearest_upsample_x2(x), skip)) — with the up-sampling's backward in
// the epilogue: `d` = the stride-1 data-gradient descriptor (C0 = the convolution's output channels, Cout = x's channels +
// skip's, cout_split = x's channels, multiples of 64), u = dt_winograd_weights of the flipped / transposed weights.
// gx [B, Hin/2, Win/2, cout_split] receives the 2x2-summed gradient of x, red the BatchNorm-backward sums of the layer that
// produced x (fuse: its raw output at gx's resolution + mean / invstd / act_scale / act_shift), dskip [B, Hin, Win,
// Cout - cout_split] the skip's gradient.  One launch, per tile epilogue form 6 or the plain store; replaces
// dt_conv2d_winograd (split outputs) + dt_upsample2x_bwd_bn.  P = dt_conv2d_winograd_upsampled_dgrad_rows(d).
static int wn_updgrad_ok(const dt_conv_desc* d) {
  static const int on = [] {
    const char* e = getenv("DT_FP32_WINO_UPSAMPLE_BWD");
    return (e == nullptr || e[0] != '0') ? 1 : 0;
  }();
  return on && dt_conv2d_winograd_supported(d) && d->mode0 == 0 && d->C1 == 0 && d->accumulate == 0 && d->cout_split > 0 &&
         (d->cout_split % 64) == 0 && ((d->Hin | d->Win) & 1) == 0 && !wn_pack(d);
}

// one partial row per workgroup where every tile of a workgroup has the same channel block (32 % blocks == 0), else one per
// spatial tile
static int wn_updgrad_rows(const dt_conv_desc* d, int* pstats) {
  const int nt = d->Cout / 64, sp = d->B * dt_cdiv(d->Ho, 16) * dt_cdiv(d->Wo, 16);
  const long total = (long)sp * nt;
  *pstats = (32 % nt) == 0;
  return *pstats ? (int)(total < WN_MAX_WGS ? total : WN_MAX_WGS) : sp;
}

extern "C" int dt_conv2d_winograd_upsampled_dgrad_supported(const dt_conv_desc* d) { return d != nullptr && wn_updgrad_ok(d); }

extern "C" int dt_conv2d_winograd_upsampled_dgrad_rows(const dt_conv_desc* d) {
  if (d == nullptr || !wn_updgrad_ok(d)) return 0;
  int ps;
  return wn_updgrad_rows(d, &ps);
}

extern "C" int dt_conv2d_winograd_upsampled_dgrad(const dt_conv_desc* d, const float* dy, const float* u, float* gx,
                                                  float* dskip, float* red, const dt_bn_bwd_fuse* fuse, int launches,
                                                  void* stream) {   // launches: reserved (one launch covers both parts)
  DT_REQUIRE(d && dy && u && gx && red && fuse && fuse->y && fuse->mean && fuse->invstd && fuse->act_scale &&
                 fuse->act_shift && fuse->act == nullptr, "conv_winograd_upsampled_dgrad: null pointer / stored activation");
  DT_REQUIRE(wn_updgrad_ok(d), "conv_winograd_upsampled_dgrad: layer shape not supported");
  DT_REQUIRE(d->cout_split == d->Cout || dskip, "conv_winograd_upsampled_dgrad: dskip missing");
  WinoArgs a;
  a.bnb = *fuse;
  a.src0 = dy; a.src1 = nullptr; a.u = u; a.in_scale = nullptr; a.in_shift = nullptr;
  a.out0 = gx; a.out1 = dskip; a.stats = red;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = 0; a.mode0 = 0;
  a.Cout = d->Cout; a.cout_split = d->cout_split; a.accumulate = 0;
  a.tiles_x = dt_cdiv(d->Wo, 16);
  a.tiles_y = dt_cdiv(d->Ho, 16);
  a.pack = 0;
  a.nchunks = d->C0 / 8;
  const size_t b0 = (size_t)d->B * d->Hin * d->Win * d->C0 * 4;
  const size_t opx = (size_t)d->B * d->Ho * d->Wo * 4;
  DT_REQUIRE(b0 < 0x80000000ull && opx * (d->Cout - d->cout_split) < 0x80000000ull,
             "conv_winograd_upsampled_dgrad: an operand of 2 GiB or more");
  a.bytes0 = (unsigned)b0; a.bytes1 = 0;
  a.obytes0 = (unsigned)(opx / 4 * d->cout_split);       // (half resolution; the EPI 6 epilogue builds its own descriptors)
  a.obytes1 = (unsigned)(opx * (d->Cout - d->cout_split));
  a.ubytes = (unsigned)((size_t)16 * d->C0 * d->Cout * 4);
  const int sp_tiles = d->B * a.tiles_x * a.tiles_y;
  hipStream_t st = (hipStream_t)stream;
  // ONE launch over all channel blocks: the tiles of the up-sampled part (blocks below cout_split / 64) take epilogue form
  // 6 (2x2 sums + BatchNorm-backward sums at half resolution), the skip's tiles the plain store into dskip
  // (two launches over the two parts measured time-neutral against the unfused chain: 806.0 vs 807.2 tiles/s)
  (void)launches;
  a.nt0 = 0; a.n_tiles = d->Cout / 64; a.stat_ld = d->cout_split;
  a.P = wn_updgrad_rows(d, &a.pstats);
  const int total = sp_tiles * a.n_tiles;
  hipLaunchKernelGGL((conv3x3_wino_kernel<false, 6, false>), dim3((unsigned)(total < WN_MAX_WGS ? total : WN_MAX_WGS)),
                     dim3(256), 0, st, a, total);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// data gradient with the BatchNorm-backward sums of the layer the gradient belongs to fused into the epilogue: the
// Winograd form of dt_conv2d_bn_bwd (same arguments; `u` = dt_winograd_weights of the flipped / transposed weights)
extern "C" int dt_conv2d_winograd_bn_bwd(const dt_conv_desc* d, const float* src0, const float* u, float* out0,
                                         float* red, const dt_bn_bwd_fuse* fuse, void* stream) {
  DT_REQUIRE(d && src0 && u && out0 && fuse && red && fuse->y && fuse->mean && fuse->invstd, "conv_winograd_bn_bwd: null pointer");
  DT_REQUIRE(fuse->act != nullptr || (fuse->act_scale && fuse->act_shift),
             "conv_winograd_bn_bwd: give the stored activation or the scale/shift of a virtual one");
  DT_REQUIRE(d->mode0 == 0 && d->C1 == 0 && d->cout_split == 0, "conv_winograd_bn_bwd: plain 3x3 stride-1 data gradients only");
  DT_REQUIRE((d->accumulate != 0) == (fuse->act != nullptr),
             "conv_winograd_bn_bwd: gradient joins (accumulate) go with a stored activation, plain stores with a virtual one");
  return dt_conv_wino_launch(d, src0, nullptr, u, out0, nullptr, red, nullptr, nullptr, fuse, (hipStream_t)stream, false);
}
