// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950 — the 3x3 stride-1 layers of the U-Net
// (forward and data gradient; 96 % of the network's FLOPs) with 2.25x fewer multiplies than the direct form.
//
// Replaces the same ATen conv2d / convolution_backward(input) calls as conv_fwd.hip (smp.Unet(resnet34) reached from
// deadtrees/network/segmodel.py:214,235,280); cuDNN / MIOpen pick the same algorithm for fp32 3x3 layers.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      per 4x4 input tile d -> 2x2 output tile Y  (Lavin & Gray 2016)
//
// Design (MI355X-first):
//   * one workgroup (4 waves, one per SIMD, 512 registers each) owns 16x16 output pixels = 64 Winograd tiles x 64
//     output channels; a wave owns 32 tiles x 32 channels and keeps all 16 transform-domain positions of them in
//     256 accumulator registers (16 independent 32x32 MFMA tiles -> no dependent-issue stalls);
//   * the 16 element-wise products are 16 GEMMs over input channels on v_mfma_f32_32x32x2_f32;
//   * per 8-channel chunk: every thread loads the 4x4 patch of one tile for 2 channels straight from global memory
//     (upsample / concat / zero padding / the producer's BatchNorm+ReLU are index arithmetic and two VALU ops here),
//     transforms it in registers (32 adds per channel) and writes the 16 positions to LDS `[pos][k-half][tile][4]`,
//     so a wave's A operand of one position for the WHOLE chunk is one conflict-free ds_read_b128;
//   * the transformed weights U = G g G^T are produced once per step by dt_winograd_weights in exactly the LDS image
//     order `[pos][chunk][k-half][Cout][4]` and go global -> LDS by DMA (global_load_lds_dwordx4, no registers);
//   * both LDS images are double-buffered: one barrier per chunk; the loads + transform of chunk c+1 are issued
//     between the MFMA groups of chunk c;
//   * output transform (24 adds per tile and channel) is lane-local on the accumulators; epilogue = 128-B row stores
//     + per-channel sum / sum-of-squares partials for the following BatchNorm, like conv_fwd.hip.
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
  unsigned bytes0, bytes1;   // sizes of the two sources (buffer descriptors: out-of-range loads return 0)
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
//      2 gradient join (out0 += ...); 3 join + BatchNorm-backward sums (stored activation)
template <bool TF, int EPI>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_kernel(const WinoArgs a) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * WN_BUF + (TF ? 2 * WN_TF_MAXC : 4)];
  float* Vb = lds;
  float* Ub = lds + 2 * WN_BUF;
  float* lds_tf = lds + 4 * WN_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (TF) {
    for (int i = tid; i < a.C0; i += 256) {
      lds_tf[i] = a.in_scale[i];
      lds_tf[WN_TF_MAXC + i] = a.in_shift[i];
    }
  }
  const int wm = wave >> 1, wn = wave & 1, kh = lane >> 5, r = lane & 31;

  const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);   // n-tiles of one spatial tile share an XCD's L2
  const int nt = wg % a.n_tiles;
  const int sp = wg / a.n_tiles;
  const int tx = sp % a.tiles_x;
  const int ty = (sp / a.tiles_x) % a.tiles_y;
  const int b = sp / (a.tiles_x * a.tiles_y);
  const int oy0 = ty * 16, ox0 = tx * 16, n0 = nt * 64;

  // ---- staging role: Winograd tile wt (8 x 8 per workgroup), channel pair q of the chunk.  Byte offsets of the 16
  // patch pixels in source 0 / source 1; WN_OOB (beyond num_records of the buffer descriptor -> the load returns 0)
  // for padding.  One buffer_load_dwordx2 per pixel and chunk, no branches, no 64-bit address arithmetic.
  const int q = tid & 3, wt = tid >> 2;
  unsigned off0[16], off1[16];
  {
    const int iy = oy0 - 1 + 2 * (wt >> 3), ix = ox0 - 1 + 2 * (wt & 7);
    const int Hs0 = a.mode0 ? (a.Hin >> 1) : a.Hin, Ws0 = a.mode0 ? (a.Win >> 1) : a.Win;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = iy + i, x = ix + j;
        const bool ok = (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
        const int p0 = (b * Hs0 + (a.mode0 ? (y >> 1) : y)) * Ws0 + (a.mode0 ? (x >> 1) : x);
        const int p1 = (b * a.Hin + y) * a.Win + x;
        off0[4 * i + j] = ok ? (unsigned)(p0 * a.C0 + 2 * q) * 4u : WN_OOB;
        off1[4 * i + j] = ok ? (unsigned)(p1 * a.C1 + 2 * q) * 4u : WN_OOB;
      }
  }
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, a.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0, a.bytes1, 0x00020000);
  f32x2 d[16], t[16];
  auto load_pixel = [&](int i, int c0) {
    const bool use0 = c0 < a.C0;   // wave-uniform
    const unsigned cb = (unsigned)(use0 ? c0 : c0 - a.C0) * 4u;
    const u32x2 v = use0 ? __builtin_amdgcn_raw_buffer_load_b64(rs0, off0[i] + cb, 0, 0)
                         : __builtin_amdgcn_raw_buffer_load_b64(rs1, off1[i] + cb, 0, 0);
    d[i] = __builtin_bit_cast(f32x2, v);
  };
  // the producer's BatchNorm-apply + ReLU on the real pixels of source 0 (zero padding stays zero)
  auto tf_pixel = [&](int i, int c0) {
    if constexpr (TF) {
      if (c0 < a.C0) {
        const f32x2 sc = *reinterpret_cast<const f32x2*>(lds_tf + c0 + 2 * q);
        const f32x2 sh = *reinterpret_cast<const f32x2*>(lds_tf + WN_TF_MAXC + c0 + 2 * q);
        f32x2 v = d[i] * sc + sh;
        v[0] = v[0] < 0.f ? 0.f : v[0];
        v[1] = v[1] < 0.f ? 0.f : v[1];
        d[i] = off0[i] != WN_OOB ? v : d[i];
      }
    }
  };
  // B^T d B in registers: column j of B^T d, then row i of (B^T d) B -> 4 positions of (tile wt, channels 2q, 2q+1)
  auto transform_col = [&](int j) {
    t[0 + j] = d[0 + j] - d[8 + j];
    t[4 + j] = d[4 + j] + d[8 + j];
    t[8 + j] = d[8 + j] - d[4 + j];
    t[12 + j] = d[4 + j] - d[12 + j];
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
  // transformed weights of chunk `ch`: 32 planes of 1 KiB, 8 per wave, straight into LDS
  const float* ug = a.u + (size_t)(n0 + lane) * 4;
  auto dma_plane = [&](float* Ud, int ch, int i) {
    const int plane = wave * 8 + i;
    const int pos = plane >> 1, k = plane & 1;
    wn_dma16(ug + (((size_t)pos * a.nchunks + ch) * 2 + k) * (size_t)a.Cout * 4, Ud + plane * WN_PS);
  };

  f32x16 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;

  const int aoff = kh * WN_PS + (32 * wm + r) * 4;
  const int boff = kh * WN_PS + (32 * wn + r) * 4;

  if constexpr (TF) __syncthreads();   // lds_tf
  // ---- prologue: chunk 0 (exposed once per tile)
#pragma unroll
  for (int i = 0; i < 16; ++i) load_pixel(i, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_plane(Ub, 0, i);
#pragma unroll
  for (int i = 0; i < 16; ++i) tf_pixel(i, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) transform_col(j);
#pragma unroll
  for (int i = 0; i < 4; ++i) transform_row_write(Vb, i);

  // ---- one chunk: 16 positions x 4 k-steps = 64 MFMAs of 64 cycles; everything else is issued in their shadow, a few
  // instructions after each one (slot k = 4 p + j): the next position's fragments right after the first MFMA of a
  // position; chunk c+1's 16 pixel loads and 8 weight-plane DMAs in slots 1..23; BatchNorm+ReLU of the loaded pixels in
  // 32..39 and the input transform + its 16 LDS writes in 40..47.
  auto chunk = [&](int c, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's weight planes of chunk c have landed
    __syncthreads();                                   // everybody's V / U of chunk c; nobody reads the other images
    const float* Vc = Vb + (c & 1) * WN_BUF + aoff;
    const float* Uc = Ub + (c & 1) * WN_BUF + boff;
    float* Vn = Vb + ((c + 1) & 1) * WN_BUF;
    float* Un = Ub + ((c + 1) & 1) * WN_BUF;
    const int c1 = 8 * (c + 1);
    f32x4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const f32x4*>(Vc);
    fb[0] = *reinterpret_cast<const f32x4*>(Uc);
    wn_static_for<0, 64>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int p = k >> 2, j = k & 3, cur = p & 1;
#if defined(WN_ABLATE) && WN_ABLATE == 3      // throw-away measurement build: staging and fragment reads only
      asm volatile("" ::"v"(fa[cur][j]), "v"(fb[cur][j]));
#else
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][j], fb[cur][j], acc[p], 0, 0, 0);
#endif
      if (j == 0 && p + 1 < 16) {
        fa[cur ^ 1] = *reinterpret_cast<const f32x4*>(Vc + (p + 1) * 2 * WN_PS);
        fb[cur ^ 1] = *reinterpret_cast<const f32x4*>(Uc + (p + 1) * 2 * WN_PS);
      }
      if constexpr (MORE) {
#if !(defined(WN_ABLATE) && WN_ABLATE == 1)   // 1: no staging after the first chunk
        if (j != 0 && k < 24) {
          const int li = 3 * p + (j - 1);   // 0..17
#if !(defined(WN_ABLATE) && WN_ABLATE == 4)   // 4: weight DMA only
          if (li < 16) load_pixel(li, c1);
#endif
#if !(defined(WN_ABLATE) && WN_ABLATE == 5)   // 5: input loads + transform only
          if (li >= 8 && li < 16) dma_plane(Un, c + 1, li - 8);
#endif
        }
#if !(defined(WN_ABLATE) && WN_ABLATE == 4)
        if (k >= 32 && k < 40) {
          tf_pixel(2 * (k - 32), c1);
          tf_pixel(2 * (k - 32) + 1, c1);
        }
        if (k >= 40 && k < 44) transform_col(k - 40);
        if (k >= 44 && k < 48) transform_row_write(Vn, k - 44);
#endif
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  for (int c = 0; c + 1 < a.nchunks; ++c) chunk(c, std::true_type{});
  chunk(a.nchunks - 1, std::false_type{});

#if defined(WN_ABLATE) && WN_ABLATE == 2     // throw-away measurement build: no epilogue
#pragma unroll
  for (int p = 0; p < 16; ++p) asm volatile("" ::"v"(acc[p]));
  return;
#endif
  // ---------------- output transform A^T M A (lane-local) + epilogue
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
  if constexpr (EPI == 1 || EPI == 3) {
    b_mu = a.bnb.mean[n];
    b_is = a.bnb.invstd[n];
    if constexpr (EPI == 1) {
      b_sc = a.bnb.act_scale[n];
      b_sh = a.bnb.act_shift[n];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = (i & 3) + 8 * (i >> 2) + 4 * kh;
    const int t = 32 * wm + m;
    const int oy = oy0 + 2 * (t >> 3), ox = ox0 + 2 * (t & 7);
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t0[j] = acc[0 + j][i] + acc[4 + j][i] + acc[8 + j][i];
      t1[j] = acc[4 + j][i] - acc[8 + j][i] - acc[12 + j][i];
    }
    float y[4];
    y[0] = t0[0] + t0[1] + t0[2];
    y[1] = t0[1] - t0[2] - t0[3];
    y[2] = t1[0] + t1[1] + t1[2];
    y[3] = t1[1] - t1[2] - t1[3];
    size_t off[4];
    bool ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int py = oy + (e >> 1), px = ox + (e & 1);
      ok[e] = py < a.Hin && px < a.Win;
      off[e] = (((size_t)b * a.Hin + py) * a.Win + px) * ld + nn;
    }
    if constexpr (EPI >= 2) {
      const bool join = outp == a.out0;   // split data gradients accumulate into out0 only
      float prev[4], yv[4], zv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) prev[e] = (join && ok[e]) ? outp[off[e]] : 0.f;
      if constexpr (EPI == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          yv[e] = ok[e] ? a.bnb.y[off[e]] : 0.f;
          zv[e] = ok[e] ? a.bnb.act[off[e]] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ok[e]) {
          const float v = y[e] + prev[e];
          if constexpr (EPI == 3) {
            const float g = zv[e] > 0.f ? v : 0.f;
            s1 += g;
            s2 += g * ((yv[e] - b_mu) * b_is);
          }
          outp[off[e]] = v;
        }
    } else if constexpr (EPI == 1) {
      float yv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) yv[e] = ok[e] ? a.bnb.y[off[e]] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ok[e]) {
          const float g = (yv[e] * b_sc + b_sh) > 0.f ? y[e] : 0.f;
          s1 += g;
          s2 += g * ((yv[e] - b_mu) * b_is);
          outp[off[e]] = y[e];
        }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ok[e]) {
          s1 += y[e];
          s2 += y[e] * y[e];
          outp[off[e]] = y[e];
        }
    }
  }
  if (a.stats != nullptr) {
    __syncthreads();   // all waves done with the operand images
    float* red = lds;  // [2][2 m-waves][64]
    const float u1 = s1 + __shfl_xor(s1, 32, 64);
    const float u2 = s2 + __shfl_xor(s2, 32, 64);
    if (kh == 0) {
      red[wm * 64 + 32 * wn + r] = u1;
      red[128 + wm * 64 + 32 * wn + r] = u2;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      a.stats[((size_t)which * a.P + sp) * a.Cout + n0 + c] = red[which * 128 + c] + red[which * 128 + 64 + c];
    }
  }
}

// ---------------------------------------------------------------- U = G g G^T in the kernel's LDS image order
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout) {
  // one thread per (input-channel quad, output channel): 9 x 4 weights in, 16 float4 out
  const int co = blockIdx.x * 64 + (threadIdx.x & 63);
  const int cq = blockIdx.y * 4 + (threadIdx.x >> 6);
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

extern "C" int dt_winograd_weights(const float* w_hwio, float* u, int Cin, int Cout, void* stream) {
  DT_REQUIRE(w_hwio && u && Cin > 0 && Cout > 0 && (Cin % 8) == 0, "winograd_weights: Cin must be a multiple of 8");
  dim3 grid(dt_cdiv(Cout, 64), dt_cdiv(Cin / 4, 4));
  hipLaunchKernelGGL(wino_weights_kernel, grid, dim3(256), 0, (hipStream_t)stream, w_hwio, u, Cin, Cout);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_conv2d_winograd_supported(const dt_conv_desc* d) {
  if (d == nullptr) return 0;
  if (d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->mode0 == 2) return 0;
  if ((d->C0 % 8) != 0 || (d->C1 % 8) != 0 || (d->Cout % 64) != 0 || d->C0 > WN_TF_MAXC) return 0;
  if (d->cout_split != 0 && (d->cout_split % 64) != 0) return 0;
  return 1;
}

extern "C" int dt_conv2d_winograd_stat_rows(const dt_conv_desc* d) {
  return d->B * dt_cdiv(d->Ho, 16) * dt_cdiv(d->Wo, 16);
}

int dt_conv_wino_launch(const dt_conv_desc* d, const float* src0, const float* src1, const float* u, float* out0,
                        float* out1, float* stats, const float* in_scale, const float* in_shift,
                        const dt_bn_bwd_fuse* fuse, hipStream_t st) {
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
  a.P = d->B * a.tiles_x * a.tiles_y;
  a.nchunks = (d->C0 + d->C1) / 8;
  const size_t px0 = (size_t)d->B * (d->mode0 ? (d->Hin / 2) * (size_t)(d->Win / 2) : (size_t)d->Hin * d->Win);
  const size_t b0 = px0 * d->C0 * 4, b1 = (size_t)d->B * d->Hin * d->Win * d->C1 * 4;
  DT_REQUIRE(b0 < 0x80000000ull && b1 < 0x80000000ull, "conv_winograd: a source of 2 GiB or more (use the direct kernel)");
  a.bytes0 = (unsigned)b0;
  a.bytes1 = (unsigned)b1;
  const bool bnb = a.bnb.y != nullptr, join = d->accumulate != 0;
  DT_REQUIRE(!bnb || stats != nullptr, "conv_winograd: fused BatchNorm-backward sums need the stats buffer");
  DT_REQUIRE(!(in_scale != nullptr && bnb), "conv_winograd: no input transform on the BatchNorm-backward form");
  dim3 g((unsigned)((long)a.P * a.n_tiles)), blk(256);
  if (in_scale != nullptr && join) hipLaunchKernelGGL((conv3x3_wino_kernel<true, 2>), g, blk, 0, st, a);
  else if (in_scale != nullptr) hipLaunchKernelGGL((conv3x3_wino_kernel<true, 0>), g, blk, 0, st, a);
  else if (!bnb && !join) hipLaunchKernelGGL((conv3x3_wino_kernel<false, 0>), g, blk, 0, st, a);
  else if (bnb && !join) hipLaunchKernelGGL((conv3x3_wino_kernel<false, 1>), g, blk, 0, st, a);
  else if (!bnb && join) hipLaunchKernelGGL((conv3x3_wino_kernel<false, 2>), g, blk, 0, st, a);
  else hipLaunchKernelGGL((conv3x3_wino_kernel<false, 3>), g, blk, 0, st, a);
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
  return dt_conv_wino_launch(d, src0, src1, u, out0, out1, stats, in_scale, in_shift, nullptr, (hipStream_t)stream);
}
