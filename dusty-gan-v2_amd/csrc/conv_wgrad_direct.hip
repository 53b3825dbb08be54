// Direct (LDS halo-tile) weight gradient of the 3x3 / 1x1 ring-padded convolutions for the
// small-channel, large-image layers of the discriminator (C, O <= 64 at 64x512 .. 32x256), where an
// im2col gather would re-read every input pixel nine times through L2.
//   gw[o, ky, kx, c] = sum_{b, ho, wo} gy[b, ho, wo, o] * xpad[b, ho*s + ky, wo*s + kx, c]
// reference: the weight gradient autograd derives for ops.Conv2d (gans/models/ops/common.py:187-210).
//
// MFMA form: D[m = o][n = c] = sum_k A[k][m] B[k][n] with k = output pixels; both operands are
// pixel-major in memory, so fragments come from the transposed LDS read (ds_read_b64_tr_b16, TnFrag in
// gemm_core.h).  A block owns a 32-column strip of one image, a 32 x 32 (o, c) tile and ALL taps; it walks
// the strip four rows at a time: the gy rows and the input halo tile are staged once per chunk and reused
// by the nine taps.  The 18 (tap, c-fragment) units are dealt round-robin to the four waves (each unit
// keeps both o-fragments, so every B fragment feeds two MFMAs); accumulators stay in registers for the
// whole strip and leave as fp32 atomics (one tile per 2048+ pixels).
#include "gemm_core.h"

namespace {

constexpr int WTO = 32, WTC = 32, WROWS = 4, WCOLS = 32;

struct WGeom {
  int B, H, W, C, O, Ho, Wo, k, stride, pad, ring;
};

template <typename T, int S>
__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(float* __restrict__ gw, const T* __restrict__ gy,
                                                                const T* __restrict__ x, WGeom g, int rows_per_blk) {
  constexpr int KS = TnFrag<T>::KS;
  constexpr int CE = 16 / sizeof(T);
  constexpr int IN_ROWS = (WROWS - 1) * S + 3, IN_COLS = (WCOLS - 1) * S + 3;
  __shared__ __attribute__((aligned(16))) T lds_gy[WROWS * WCOLS * WTO];
  __shared__ __attribute__((aligned(16))) T lds_x[IN_ROWS * IN_COLS * WTC];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int ntaps = g.k * g.k;
  const int col_blocks = (g.Wo + WCOLS - 1) / WCOLS;
  const int w0 = (blockIdx.x % col_blocks) * WCOLS;
  const int rg = blockIdx.x / col_blocks;              // row group within the image
  const int b = blockIdx.y;
  const int ctiles = g.C / WTC;
  const int c0 = (blockIdx.z % ctiles) * WTC;
  const int o0 = (blockIdx.z / ctiles) * WTO;
  const int nunits = ntaps * 2;                        // (tap, c-fragment)

  f32x4 acc[5][2];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    acc[u][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[u][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int h_begin = rg * rows_per_blk;
  const int h_end = (h_begin + rows_per_blk < g.Ho) ? h_begin + rows_per_blk : g.Ho;
  const T* gyb = gy + (int64_t)b * g.Ho * g.Wo * g.O;
  const T* xb = x + (int64_t)b * g.H * g.W * g.C;
  const int off = (g.k == 3) ? g.pad : 0;              // tap (ky,kx) reads input (ho*s + ky - pad, ...)

  for (int h0 = h_begin; h0 < h_end; h0 += WROWS) {
    __syncthreads();
    // gy chunk: [WROWS][WCOLS][WTO], zero outside the image / beyond O
    for (int id = tid; id < WROWS * WCOLS * (WTO / CE); id += 256) {
      const int ch = id % (WTO / CE);
      const int pix = id / (WTO / CE);
      const int r = pix / WCOLS, cix = pix - r * WCOLS;
      const int ho = h0 + r, wo = w0 + cix, o = o0 + ch * CE;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ho < h_end && wo < g.Wo && o < g.O)
        v = *reinterpret_cast<const uint4*>(gyb + ((int64_t)ho * g.Wo + wo) * g.O + o);
      reinterpret_cast<uint4*>(lds_gy)[id] = v;
    }
    // input halo: rows h0*S - off .. , cols w0*S - off ..; ring wrap in W, replicate clamp in H
    for (int id = tid; id < IN_ROWS * IN_COLS * (WTC / CE); id += 256) {
      const int ch = id % (WTC / CE);
      const int pix = id / (WTC / CE);
      const int iy = pix / IN_COLS, ix = pix - iy * IN_COLS;
      int hi = h0 * S - off + iy, wi = w0 * S - off + ix;
      hi = hi < 0 ? 0 : (hi >= g.H ? g.H - 1 : hi);
      wi = g.ring ? floormod(wi, g.W) : (wi < 0 ? 0 : (wi >= g.W ? g.W - 1 : wi));
      reinterpret_cast<uint4*>(lds_x)[id] =
          *reinterpret_cast<const uint4*>(xb + ((int64_t)hi * g.W + wi) * g.C + c0 + ch * CE);
    }
    __syncthreads();
    for (int r = 0; r < WROWS; ++r) {
#pragma unroll
      for (int kb = 0; kb < WCOLS / KS; ++kb) {
        const T* arow = lds_gy + (r * WCOLS + kb * KS) * WTO;
        const uint4 a0 = TnFrag<T>::template read<WTO>(arow, 0, lane);
        const uint4 a1 = TnFrag<T>::template read<WTO>(arow, 16, lane);
#pragma unroll
        for (int ui = 0; ui < 5; ++ui) {
          const int u = wave + ui * 4;
          if (u < nunits) {
            const int tap = u >> 1, nf = u & 1;
            const int ky = tap / g.k, kx = tap - ky * g.k;
            const T* brow = lds_x + ((r * S + ky) * IN_COLS + kb * KS * S + kx) * WTC;
            const uint4 bb = TnFrag<T>::template read<S * WTC>(brow, nf * 16, lane);
            Mfma16<T>::run(acc[ui][0], a0, bb);
            Mfma16<T>::run(acc[ui][1], a1, bb);
          }
        }
      }
    }
  }
  // D layout: column (c) = lane & 15, rows (o) = 4 * (lane >> 4) + r
  const int lr = lane & 15, lc = lane >> 4;
#pragma unroll
  for (int ui = 0; ui < 5; ++ui) {
    const int u = wave + ui * 4;
    if (u >= nunits) continue;
    const int tap = u >> 1, nf = u & 1;
    const int c = c0 + nf * 16 + lr;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + mf * 16 + lc * 4 + r;
        if (o < g.O) atomicAdd(&gw[((int64_t)o * ntaps + tap) * g.C + c], acc[ui][mf][r]);
      }
  }
}

}  // namespace

// gw fp32 [O, k*k, C] (overwritten).  Eligibility: k in {1,3}, pad = (k-1)/2, stride in {1,2}, C % 32 == 0,
// O % 8 == 0; returns DGV2_EINVAL otherwise (callers then use dgv2_conv_wgrad).
extern "C" int dgv2_conv_wgrad_direct(float* gw, const void* gy, const void* x, int B, int H, int W, int C, int O,
                                      int k, int stride, int pad, int ring, int dtype, void* stream) {
  if (!gw || !gy || !x || B <= 0 || H <= 0 || W <= 0 || C <= 0 || O <= 0) return DGV2_EINVAL;
  if ((k != 1 && k != 3) || pad != (k - 1) / 2 || (stride != 1 && stride != 2)) return DGV2_EINVAL;
  const int ce = dtype == DGV2_BF16 ? 8 : 4;
  if (C % WTC || O % ce || !aligned16(gy) || !aligned16(x)) return DGV2_EINVAL;
  WGeom g{B, H, W, C, O, (H + 2 * pad - k) / stride + 1, (W + 2 * pad - k) / stride + 1, k, stride, pad, ring};
  if (g.Ho <= 0 || g.Wo <= 0) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(gw, 0, sizeof(float) * (size_t)O * k * k * C, st);
  if (e != hipSuccess) return (int)e;
  const int col_blocks = (g.Wo + WCOLS - 1) / WCOLS;
  // whole-height strips unless that leaves the chip idle; keep >= 16 rows per block to amortise the atomics
  int row_groups = 1;
  const int tiles = ((O + WTO - 1) / WTO) * (C / WTC);
  while ((int64_t)col_blocks * row_groups * B * tiles < 1024 && g.Ho / (row_groups * 2) >= 16) row_groups *= 2;
  const int rows_per_blk = ((g.Ho + row_groups - 1) / row_groups + WROWS - 1) / WROWS * WROWS;
  dim3 grid(col_blocks * row_groups, B, tiles);
  if (dtype == DGV2_BF16) {
    if (stride == 1)
      conv_wgrad_direct_kernel<bf16_t, 1><<<grid, 256, 0, st>>>(gw, (const bf16_t*)gy, (const bf16_t*)x, g, rows_per_blk);
    else
      conv_wgrad_direct_kernel<bf16_t, 2><<<grid, 256, 0, st>>>(gw, (const bf16_t*)gy, (const bf16_t*)x, g, rows_per_blk);
  } else if (dtype == DGV2_F32 && stride == 1) {
    conv_wgrad_direct_kernel<float, 1><<<grid, 256, 0, st>>>(gw, (const float*)gy, (const float*)x, g, rows_per_blk);
  } else {
    return DGV2_EINVAL;  // fp32 stride 2 would need > 64 KB of static LDS: use dgv2_conv_wgrad
  }
  DGV2_RETURN_LAST();
}
