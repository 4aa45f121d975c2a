// Winograd F(2x2, 3x3) weight gradient of the 3x3 stride-1 layers on the fp32 matrix cores of gfx950.
//
// Replaces ATen convolution_backward(weight) (autograd of the smp.Unet(resnet34) convolutions reached from
// deadtrees/network/segmodel.py:214-222) for the layers conv_wino.hip runs forward: with
//     Y = A^T [ sum_ci U .* V ] A,   U = G g G^T,   V = B^T d B            (conv_wino.hip)
// the gradient of the transformed weights is a sum over Winograd tiles t of element-wise products,
//     dU[pos][ci][co] = sum_t V[pos][t][ci] * W[pos][t][co],   W = A dY A^T   (4x4 from the 2x2 output gradient),
// i.e. 16 GEMMs with K = tiles (2.25x fewer multiplies than the 9 tap GEMMs of the direct form), and dg = G^T dU G.
//
// Design (MI355X-first; the mirror image of conv_wino.hip):
//   * a workgroup (4 waves, one per SIMD, 512 registers each) owns one 64 ci x 64 co block of dU for a contiguous range
//     of tiles (split-K); a wave keeps 32 ci x 32 co x 16 positions in 256 accumulator registers;
//   * per chunk of 8 tiles every thread loads the 4x4 input patch (upsample / concat / padding / the producer's
//     BatchNorm+ReLU applied on the fly) and the 2x2 output-gradient patch of ONE tile for 2 channels straight from global
//     memory (32 lanes x 8 B = one 256-B row per pixel), transforms both in registers and writes the 16 positions to
//     LDS `[pos][k-half][k-step][64 channels]` with conflict-free ds_write_b64; fragments are conflict-free ds_read_b32;
//   * LDS images double-buffered, one barrier per chunk; loads run two chunks ahead (two register sets);
//   * partial dU blocks go to a workspace; a fixed-order two-stage reduction sums the splits and applies G^T . G
//     (deterministic: no float atomics).
#include "common.h"

#include <type_traits>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define WW_BUF 8192                    // floats of one operand image of one chunk: 16 pos x 2 halves x 4 steps x 64 ch
#define WW_TF_MAXC 512
#define WW_OOB 0x80000000u
#define WW_MAX_WGS 256

struct WinoWgArgs {
  const float* src0;
  const float* src1;
  const float* dy;
  const float* in_scale;
  const float* in_shift;
  float* ws;
  int B, Hin, Win, C0, C1, mode0, Cout;
  int tiles_x, tiles_y, T;
  int ci_blocks, co_blocks, ksplit, cps;   // cps: chunks (of 8 tiles) per split, even
  unsigned bytes0, bytes1, dybytes, wsbytes;
};

template <int K, int N, class F>
__device__ __forceinline__ void ww_static_for(F&& f) {
  if constexpr (K < N) {
    f(std::integral_constant<int, K>{});
    ww_static_for<K + 1, N>(f);
  }
}

// CO32: a layer with 32 output channels (dec3.conv1: 128 -> 32): one 32-channel co block.  The two waves of a ci half
// (wn = 0, 1) would multiply the same 32 x 32 block twice, so they split the chunk's K instead — wave wn takes k-steps
// 2 wn, 2 wn + 1 of every position (32 MFMAs per chunk, the staging work of two of the 64-slot schedule's slots behind each)
// — and write their accumulators as two partial slabs (part 2 split + wn) for the existing fixed-order reduction.
template <bool TF, bool CO32 = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_wgrad_kernel(const WinoWgArgs a) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * WW_BUF + (TF ? 2 * WW_TF_MAXC : 4)];
  float* Vb = lds;                 // [2][WW_BUF] transformed input patches
  float* Wb = lds + 2 * WW_BUF;    // [2][WW_BUF] transformed output gradients
  float* lds_tf = lds + 4 * WW_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (TF) {
    for (int i = tid; i < a.C0; i += 256) {
      lds_tf[i] = a.in_scale[i];
      lds_tf[WW_TF_MAXC + i] = a.in_shift[i];
    }
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave_u >> 1, wn = wave_u & 1, kh = lane >> 5, r = lane & 31;

  const int nblocks = a.ci_blocks * a.co_blocks;
  const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);   // the blocks of one split share an XCD's L2 (same tiles)
  const int blk = wg % nblocks, split = wg / nblocks;
  const int cib = blk / a.co_blocks, cob = blk % a.co_blocks;
  const int ci0 = 64 * cib, co0 = 64 * cob;
  const bool use0 = ci0 < a.C0;                            // the 64-channel input block lies in one source (host check)
  const int cbase = use0 ? ci0 : ci0 - a.C0;
  const int Cs = use0 ? a.C0 : a.C1;
  const int mode = use0 ? a.mode0 : 0;
  const int Hs = mode ? (a.Hin >> 1) : a.Hin, Ws = mode ? (a.Win >> 1) : a.Win;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(use0 ? a.src0 : a.src1), 0,
                                                                       use0 ? a.bytes0 : a.bytes1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);

  // ---- staging role: tile tau of the chunk, channel pair c2 of the 64-channel blocks (input and output side)
  const int tau = tid >> 5, c2 = tid & 31;
  const int tiles_img = a.tiles_x * a.tiles_y;
  const int chunk0 = split * a.cps;

  f32x2 dx[2][16], dg[2][4], t[16];
  unsigned long long xrm[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, xcm[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // row / column validity lane masks
  // ---- load context: the tile this thread stages in the NEXT chunk to be loaded (runs two chunks ahead of the MFMAs).
  // Tile coordinates advance incrementally (8 tiles per chunk) — no divisions in the loop — and the 16 + 4 byte offsets
  // are row part + column part; an invalid row makes the row part WW_OOB (the sum stays beyond every buffer: sources
  // < 2 GiB), an invalid column selects WW_OOB: one v_add + one v_cndmask per load, one load per MFMA slot.
  int l_tx, l_ty, l_b;
  {
    const int tl = chunk0 * 8 + tau;
    l_b = tl / tiles_img;
    const int rem = tl - l_b * tiles_img;
    l_ty = rem / a.tiles_x;
    l_tx = rem - l_ty * a.tiles_x;
  }
  unsigned xrow[4], xcol[4], grow[2], gcol[2];
  bool xcok[4], gcok[2];
  const unsigned xpix = (unsigned)Cs * 4u, xpitch = (unsigned)(Ws * Cs) * 4u;           // bytes per source pixel / row
  const unsigned gpix = (unsigned)a.Cout * 4u, gpitch = (unsigned)(a.Win * a.Cout) * 4u;
  const unsigned xlane = (unsigned)(cbase + 2 * c2) * 4u;
  const unsigned glane = (CO32 && c2 >= 16) ? WW_OOB : (unsigned)(co0 + 2 * c2) * 4u;   // CO32: channels 32 .. 63 do not exist
  // offsets of the tile (l_b, l_ty, l_tx); 24-bit multiplies (full rate): row indices and pitches are < 2^24 (host check)
  auto prep_rows = [&]() {
    const bool live = l_b < a.B;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = 2 * l_ty - 1 + i;
      const bool ok = live && (unsigned)y < (unsigned)a.Hin;
      xrow[i] = ok ? __umul24((unsigned)(l_b * Hs + (mode ? (y >> 1) : y)), xpitch) + xlane : WW_OOB;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int oy = 2 * l_ty + e;
      grow[e] = (live && oy < a.Hin) ? __umul24((unsigned)(l_b * a.Hin + oy), gpitch) + glane : WW_OOB;
    }
  };
  auto prep_cols = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = 2 * l_tx - 1 + j;
      xcok[j] = (unsigned)x < (unsigned)a.Win;
      xcol[j] = __umul24((unsigned)(mode ? (x >> 1) : x) & 0xffffffu, xpix);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ox = 2 * l_tx + e;
      gcok[e] = ox < a.Win;
      gcol[e] = __umul24((unsigned)ox, gpix);
    }
  };
  prep_rows();
  auto prep_chunk = [&](auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
    prep_cols();
    if constexpr (TF) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xrm[SET][i] = __builtin_amdgcn_ballot_w64(xrow[i] != WW_OOB);
        xcm[SET][i] = __builtin_amdgcn_ballot_w64(xcok[i]);
      }
    }
  };
  // the tile of the following chunk (after the chunk's loads are issued): 8 tiles further along the row; at the end of
  // a row of tiles the row offsets are rebuilt
  auto advance_tile = [&]() {
    l_tx += 8;
    if (l_tx >= a.tiles_x) {
      do {
        l_tx -= a.tiles_x;
        if (++l_ty == a.tiles_y) {
          l_ty = 0;
          ++l_b;
        }
      } while (l_tx >= a.tiles_x);
      prep_rows();
    }
  };
  auto load_one = [&](auto set_tag, int idx) {   // idx 0..15: input patch pixel (row idx >> 2, column idx & 3); 16..19: dY
    constexpr int SET = decltype(set_tag)::value;
    if (idx < 16) {
      const int i = idx >> 2, j = idx & 3;
      const unsigned off = xcok[j] ? xrow[i] + xcol[j] : WW_OOB;
      dx[SET][idx] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsx, off, 0, 0));
    } else {
      const int e = idx - 16;
      const unsigned off = gcok[e & 1] ? grow[e >> 1] + gcol[e & 1] : WW_OOB;
      dg[SET][e] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsy, off, 0, 0));
    }
  };
  // the producer's BatchNorm-apply + ReLU on the real pixels of source 0 (zero padding stays zero)
  f32x2 tf_sc = {1.f, 1.f}, tf_sh = {0.f, 0.f}, tf_v[2];
  if constexpr (TF) {
    __syncthreads();
    if (use0) {
      tf_sc = *reinterpret_cast<const f32x2*>(lds_tf + ci0 + 2 * c2);
      tf_sh = *reinterpret_cast<const f32x2*>(lds_tf + WW_TF_MAXC + ci0 + 2 * c2);
    }
  }
  const float tf_lo = (TF && use0) ? 0.f : -__builtin_inff();
  auto tf_a = [&](auto set_tag, int i) {
    if constexpr (TF) {
      constexpr int SET = decltype(set_tag)::value;
      tf_v[i & 1] = __builtin_elementwise_fma(dx[SET][i], tf_sc, tf_sh);   // like conv_wino.hip: 5 VALU per pixel pair
    }
  };
  auto tf_b = [&](auto set_tag, int i) {
    if constexpr (TF) {
      constexpr int SET = decltype(set_tag)::value;
      const f32x2 v = tf_v[i & 1];
      // NaN-keeping ReLU (see conv_wino.hip tf_b): real pixel AND NOT (v < floor) -> v, else 0
      const unsigned long long m = xrm[SET][i >> 2] & xcm[SET][i & 3];
      // !(floor > v) is true for NaN; one asm block: the keep-mask never leaves VCC
      float o0 = v[0], o1 = v[1];
      asm volatile("v_cmp_ngt_f32 vcc, %2, %0\n\ts_and_b64 vcc, vcc, %3\n\tv_cndmask_b32 %0, 0, %0, vcc\n\t"
                   "v_cmp_ngt_f32 vcc, %2, %1\n\ts_and_b64 vcc, vcc, %3\n\tv_cndmask_b32 %1, 0, %1, vcc"
                   : "+v"(o0), "+v"(o1)
                   : "s"(tf_lo), "s"(m)
                   : "vcc");
      dx[SET][i][0] = o0;
      dx[SET][i][1] = o1;
    }
  };
  // LDS image: [pos][k-half][k-step][64 channels]; tile tau is k-half tau >> 2, k-step tau & 3
  const int woff = ((tau >> 2) * 4 + (tau & 3)) * 64 + 2 * c2;
  auto x_col = [&](auto set_tag, int j) {   // column j of B^T d
    constexpr int SET = decltype(set_tag)::value;
    t[0 + j] = dx[SET][0 + j] - dx[SET][8 + j];
    t[4 + j] = dx[SET][4 + j] + dx[SET][8 + j];
    t[8 + j] = dx[SET][8 + j] - dx[SET][4 + j];
    t[12 + j] = dx[SET][4 + j] - dx[SET][12 + j];
  };
  auto x_row_write = [&](float* Vd, int i) {   // row i of (B^T d) B -> 4 positions
    float* dst = Vd + woff;
    const f32x2 v0 = t[4 * i + 0] - t[4 * i + 2];
    const f32x2 v1 = t[4 * i + 1] + t[4 * i + 2];
    const f32x2 v2 = t[4 * i + 2] - t[4 * i + 1];
    const f32x2 v3 = t[4 * i + 1] - t[4 * i + 3];
    *reinterpret_cast<f32x2*>(dst + (4 * i + 0) * 512) = v0;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 1) * 512) = v1;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 2) * 512) = v2;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 3) * 512) = v3;
  };
  // W = A dY A^T with A = [[1,0],[1,1],[1,-1],[0,-1]]: rows (dY0, dY0+dY1, dY0-dY1, -dY1), then the same on columns
  auto g_write = [&](auto set_tag, float* Wd, int i) {   // row i of A dY (i = 0..3) -> 4 positions
    constexpr int SET = decltype(set_tag)::value;
    const f32x2 g00 = dg[SET][0], g01 = dg[SET][1], g10 = dg[SET][2], g11 = dg[SET][3];
    f32x2 r0, r1;   // row i of A dY: (col 0, col 1)
    if (i == 0) { r0 = g00; r1 = g01; }
    else if (i == 1) { r0 = g00 + g10; r1 = g01 + g11; }
    else if (i == 2) { r0 = g00 - g10; r1 = g01 - g11; }
    else { r0 = -g10; r1 = -g11; }
    float* dst = Wd + woff;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 0) * 512) = r0;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 1) * 512) = r0 + r1;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 2) * 512) = r0 - r1;
    *reinterpret_cast<f32x2*>(dst + (4 * i + 3) * 512) = -r1;
  };

  f32x16 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // ---- prologue: chunk 0 staged completely, the loads of chunk 1 in flight
  prep_chunk(S0{});
#pragma unroll
  for (int i = 0; i < 20; ++i) load_one(S0{}, i);
  advance_tile();
  prep_chunk(S1{});
#pragma unroll
  for (int i = 0; i < 20; ++i) load_one(S1{}, i);
  advance_tile();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    tf_a(S0{}, i);
    tf_b(S0{}, i);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) x_col(S0{}, j);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x_row_write(Vb, i);
    g_write(S0{}, Wb, i);
  }

  const int aoff = kh * 256 + 32 * wm + r + (CO32 ? 128 * wn : 0);   // + pos * 512 + step * 64 (CO32: k-steps 2 wn, 2 wn + 1)
  const int boff = kh * 256 + (CO32 ? 128 * wn : 32 * wn) + r;

  // ---- one chunk (8 tiles = 4 k-steps x 2 halves): 16 positions x 4 k-steps = 64 MFMAs of 64 cycles; PAR = chunk & 1:
  // images PAR are multiplied, images PAR^1 receive chunk c+1 (from register set PAR^1), register set PAR is loaded
  // for chunk c+2
  auto step = [&](auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    using SN = std::integral_constant<int, PAR ^ 1>;
    const float* Vc = Vb + PAR * WW_BUF + aoff;
    const float* Wc = Wb + PAR * WW_BUF + boff;
    float* Vn = Vb + (PAR ^ 1) * WW_BUF;
    float* Wn = Wb + (PAR ^ 1) * WW_BUF;
    constexpr int NJ = CO32 ? 2 : 4;                    // k-steps of a position this wave multiplies
    float fa[2][NJ], fb[2][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      fa[0][j] = Vc[j * 64];
      fb[0][j] = Wc[j * 64];
    }
    auto slot_work = [&](auto kc) {                     // the staging work of slot k of the 64-slot schedule
      constexpr int k = decltype(kc)::value;
      if (k == 1) prep_chunk(par_tag);                  // chunk c+2: offsets, then one load per slot
      if (k >= 2 && k < 22) load_one(par_tag, k - 2);
      if (k == 52) advance_tile();
      if (k >= 22 && k < 38) tf_a(SN{}, k - 22);
      if (k >= 23 && k < 39) tf_b(SN{}, k - 23);
      if (k >= 39 && k < 43) x_col(SN{}, k - 39);
      if (k >= 43 && k < 47) x_row_write(Vn, k - 43);
      if (k >= 47 && k < 51) g_write(SN{}, Wn, k - 47);
    };
    ww_static_for<0, 16 * NJ>([&](auto mc) {
      constexpr int mslot = decltype(mc)::value;
      constexpr int p = mslot / NJ, j = mslot % NJ, cur = p & 1;
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][j], fb[cur][j], acc[p], 0, 0, 0);
      if (p + 1 < 16) {   // the next position's k-step j
        fa[cur ^ 1][j] = Vc[(p + 1) * 512 + j * 64];
        fb[cur ^ 1][j] = Wc[(p + 1) * 512 + j * 64];
      }
      if constexpr (CO32) {
        slot_work(std::integral_constant<int, 2 * mslot>{});
        slot_work(std::integral_constant<int, 2 * mslot + 1>{});
      } else {
        slot_work(mc);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loads of chunk c+2 (issued >= 58 MFMAs ago)
  };

  for (int c = 0; c < a.cps; c += 2) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    step(S0{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    step(S1{});
  }

  // ---- partial dU block -> workspace [split][block][pos][64 ci][64 co]
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws, 0, a.wsbytes, 0x00020000);
  const unsigned lane_base = (unsigned)(((32 * wm + 4 * kh) * 64 + (CO32 ? 0 : 32 * wn) + r) * 4);
  const int wg_base = ((CO32 ? 2 * split + wn : split) * nblocks + blk) * (16 * 4096 * 4);   // CO32: columns 32 .. 63 stay unwritten
#pragma unroll
  for (int p = 0; p < 16; ++p) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = (i & 3) + 8 * (i >> 2);
      const float v = acc[p][i];   // (a bit_cast applied directly to the vector element reads element 0)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsw, lane_base,
                                            wg_base + (p * 4096 + m * 64) * 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------- reduction of the splits (+ G^T dU G)
// stage A: sums groups of `rb` splits: ws [ksplit][n] -> ws2 [groups][n], n = nblocks * 16 * 4096
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ ws2,
                                                                int64_t n4, int ksplit, int rb) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int g = blockIdx.y;
  const int s0 = g * rb, s1 = s0 + rb < ksplit ? s0 + rb : ksplit;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int k = s0; k < s1; ++k) s += *reinterpret_cast<const f32x4*>(ws + ((int64_t)k * n4 + i) * 4);
  *reinterpret_cast<f32x4*>(ws2 + ((int64_t)g * n4 + i) * 4) = s;
}

// stage B: one thread per (ci, co): sums `parts` partial blocks per position, dg = G^T dU G, writes dw HWIO
__global__ __launch_bounds__(256) void wino_wgrad_final_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                               int parts, int ci_blocks, int co_blocks, int Cin,
                                                               int Cout) {
  const int blk = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;   // 0..4095 inside the block
  const int ci = (blk / co_blocks) * 64 + (e >> 6), co = (blk % co_blocks) * 64 + (e & 63);
  if (co >= Cout) return;   // the 32-channel layers' blocks are half-filled
  const int64_t n = (int64_t)ci_blocks * co_blocks * 16 * 4096;
  float u[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) u[p] = 0.f;
  for (int k = 0; k < parts; ++k) {
    const float* src = ws + (int64_t)k * n + (int64_t)blk * 16 * 4096 + e;
#pragma unroll
    for (int p = 0; p < 16; ++p) u[p] += src[p * 4096];
  }
  // G^T (3x4) = [[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]]
  float tt[12];   // G^T dU: 3 x 4
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    tt[0 + j] = u[0 + j] + 0.5f * (u[4 + j] + u[8 + j]);
    tt[4 + j] = 0.5f * (u[4 + j] - u[8 + j]);
    tt[8 + j] = 0.5f * (u[4 + j] + u[8 + j]) + u[12 + j];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float g0 = tt[4 * i + 0] + 0.5f * (tt[4 * i + 1] + tt[4 * i + 2]);
    const float g1 = 0.5f * (tt[4 * i + 1] - tt[4 * i + 2]);
    const float g2 = 0.5f * (tt[4 * i + 1] + tt[4 * i + 2]) + tt[4 * i + 3];
    dw[((size_t)(3 * i + 0) * Cin + ci) * Cout + co] = g0;
    dw[((size_t)(3 * i + 1) * Cin + ci) * Cout + co] = g1;
    dw[((size_t)(3 * i + 2) * Cin + ci) * Cout + co] = g2;
  }
}

// ---------------------------------------------------------------- host side
struct WwCfg {
  int tiles_x, tiles_y, T, ci_blocks, co_blocks, ksplit, cps, rb, groups, parts;
};

static WwCfg ww_cfg(const dt_conv_desc* d) {
  WwCfg c;
  c.tiles_x = dt_cdiv(d->Win, 2);
  c.tiles_y = dt_cdiv(d->Hin, 2);
  c.T = d->B * c.tiles_x * c.tiles_y;
  c.ci_blocks = (d->C0 + d->C1) / 64;
  c.co_blocks = d->Cout == 32 ? 1 : d->Cout / 64;
  const int nblocks = c.ci_blocks * c.co_blocks;
  const int chunks = dt_cdiv(c.T, 8);
  int ks = WW_MAX_WGS / nblocks;
  if (ks < 1) ks = 1;
  if (ks > dt_cdiv(chunks, 2)) ks = dt_cdiv(chunks, 2);
  c.cps = 2 * dt_cdiv(chunks, 2 * ks);          // even number of chunks per split
  c.ksplit = dt_cdiv(chunks, c.cps);
  c.parts = d->Cout == 32 ? 2 * c.ksplit : c.ksplit;   // partial slabs (32-channel layers: two K halves per split)
  c.rb = c.parts > 16 ? dt_cdiv(c.parts, 16) : 1;
  c.groups = dt_cdiv(c.parts, c.rb);
  return c;
}

static bool ww_co32(const dt_conv_desc* d) {   // 32 output channels: the K-split form (DT_FP32_WINO_WGRAD_CO32=0 switches it off)
  static const int on = [] {
    const char* e = getenv("DT_FP32_WINO_WGRAD_CO32");
    return (e == nullptr || e[0] != '0') ? 1 : 0;
  }();
  return on && d->Cout == 32;
}

extern "C" int dt_conv2d_wgrad_winograd_supported(const dt_conv_desc* d) {
  if (d == nullptr) return 0;
  if (d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->mode0 == 2) return 0;
  if ((d->C0 % 64) != 0 || (d->C1 % 64) != 0 || ((d->Cout % 64) != 0 && !ww_co32(d)) || d->C0 > WW_TF_MAXC) return 0;
  if (d->Ho != d->Hin || d->Wo != d->Win) return 0;
  const size_t px0 = (size_t)d->B * (d->mode0 ? (d->Hin / 2) * (size_t)(d->Win / 2) : (size_t)d->Hin * d->Win);
  if (px0 * d->C0 * 4 >= 0x80000000ull || (size_t)d->B * d->Hin * d->Win * d->C1 * 4 >= 0x80000000ull) return 0;
  if ((size_t)d->B * d->Ho * d->Wo * d->Cout * 4 >= 0x80000000ull) return 0;
  return 1;
}

extern "C" size_t dt_conv2d_wgrad_winograd_workspace(const dt_conv_desc* d) {
  if (!dt_conv2d_wgrad_winograd_supported(d)) return 0;
  const WwCfg c = ww_cfg(d);
  const size_t slab = (size_t)c.ci_blocks * c.co_blocks * 16 * 4096 * 4;
  return slab * c.parts + (c.rb > 1 ? slab * c.groups : 0);
}

extern "C" int dt_conv2d_wgrad_winograd(const dt_conv_desc* d, const float* src0, const float* src1, const float* dy,
                                        float* dw_hwio, float* workspace, size_t workspace_bytes, const float* in_scale,
                                        const float* in_shift, void* stream) {
  DT_REQUIRE(d && src0 && dy && dw_hwio && workspace, "wgrad_winograd: null pointer");
  DT_REQUIRE(dt_conv2d_wgrad_winograd_supported(d), "wgrad_winograd: layer shape not supported");
  DT_REQUIRE(d->C1 == 0 || src1, "wgrad_winograd: src1 missing");
  DT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "wgrad_winograd: in_scale/in_shift must come together");
  DT_REQUIRE(workspace_bytes >= dt_conv2d_wgrad_winograd_workspace(d), "wgrad_winograd: workspace too small");
  const WwCfg c = ww_cfg(d);
  const size_t slab = (size_t)c.ci_blocks * c.co_blocks * 16 * 4096 * 4;
  DT_REQUIRE(slab * c.parts < 0x100000000ull, "wgrad_winograd: workspace beyond 4 GiB");
  WinoWgArgs a;
  a.src0 = src0; a.src1 = src1; a.dy = dy; a.in_scale = in_scale; a.in_shift = in_shift; a.ws = workspace;
  a.B = d->B; a.Hin = d->Hin; a.Win = d->Win; a.C0 = d->C0; a.C1 = d->C1; a.mode0 = d->mode0; a.Cout = d->Cout;
  a.tiles_x = c.tiles_x; a.tiles_y = c.tiles_y; a.T = c.T;
  a.ci_blocks = c.ci_blocks; a.co_blocks = c.co_blocks; a.ksplit = c.ksplit; a.cps = c.cps;
  const size_t px0 = (size_t)d->B * (d->mode0 ? (d->Hin / 2) * (size_t)(d->Win / 2) : (size_t)d->Hin * d->Win);
  a.bytes0 = (unsigned)(px0 * d->C0 * 4);
  a.bytes1 = (unsigned)((size_t)d->B * d->Hin * d->Win * d->C1 * 4);
  a.dybytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->Cout * 4);
  a.wsbytes = (unsigned)(slab * c.parts);
  hipStream_t st = (hipStream_t)stream;
  dim3 g((unsigned)(c.ci_blocks * c.co_blocks * c.ksplit)), blk(256);
  if (d->Cout == 32) {
    if (in_scale != nullptr) hipLaunchKernelGGL((conv3x3_wino_wgrad_kernel<true, true>), g, blk, 0, st, a);
    else hipLaunchKernelGGL((conv3x3_wino_wgrad_kernel<false, true>), g, blk, 0, st, a);
  } else if (in_scale != nullptr) hipLaunchKernelGGL((conv3x3_wino_wgrad_kernel<true>), g, blk, 0, st, a);
  else hipLaunchKernelGGL((conv3x3_wino_wgrad_kernel<false>), g, blk, 0, st, a);
  DT_LAUNCH_CHECK();
  const float* parts_src = workspace;
  int parts = c.parts;
  if (c.rb > 1) {
    float* ws2 = workspace + slab / 4 * c.parts;
    const int64_t n4 = (int64_t)(slab / 16);
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)dt_cdiv(n4, 256), (unsigned)c.groups), dim3(256), 0, st,
                       workspace, ws2, n4, c.parts, c.rb);
    DT_LAUNCH_CHECK();
    parts_src = ws2;
    parts = c.groups;
  }
  hipLaunchKernelGGL(wino_wgrad_final_kernel, dim3(16, (unsigned)(c.ci_blocks * c.co_blocks)), dim3(256), 0, st, parts_src,
                     dw_hwio, parts, c.ci_blocks, c.co_blocks, d->C0 + d->C1, d->Cout);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
