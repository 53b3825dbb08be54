// upfirdn2d (reference ABI) and the separable-operator ADA resampler.
//
// upfirdn2d: gans/models/ops/upfirdn2d/upfirdn2d_kernel.cu:44-202 -- zero-insert upsample, zero
// pad / crop, true convolution with `kernel`, decimate.  Gather form, one thread per output.
//
// ada_apply: the geometric + colour stage of AdaptiveAugment.forward
// (gans/augment/adaptive_augment.py:471-545).  With the policy of configs/gans/dusty_v2.yaml the
// random transform is axis-aligned (flips, translations, vertical scale), so the whole chain
// pad -> 2x up-FIR -> bilinear grid_sample -> 2x down-FIR -> colour is a separable linear map
//     y = a * (Ay x Cx^T) + c
// per sample: Ay [H,H] dense (reflect padding, scale), Cx circulant (ring padding) given by K
// taps, an integer offset and a flip sign.  The host builds Ay / kx by pushing identities
// through the 1-D chain (see gans/augment/adaptive_augment.py in this repo); this kernel
// applies them in one pass over the image (read once, write once) with static shapes.
#include "common.h"

namespace {

template <typename T>
__global__ void upfirdn2d_kernel(T* __restrict__ out, const T* __restrict__ in, const float* __restrict__ kernel,
                                 int major, int in_h, int in_w, int minor, int kh, int kw, int up_x, int up_y,
                                 int down_x, int down_y, int pad_x0, int pad_y0, int out_h, int out_w) {
  const int64_t total = (int64_t)major * out_h * out_w * minor;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int mi = (int)(t % minor);
    int64_t r = t / minor;
    const int ox = (int)(r % out_w);
    r /= out_w;
    const int oy = (int)(r % out_h);
    const int ma = (int)(r / out_h);
    float acc = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      const int v = oy * down_y + ky - pad_y0;  // row in the zero-stuffed image
      if (v < 0 || v % up_y != 0) continue;
      const int iy = v / up_y;
      if (iy >= in_h) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int u = ox * down_x + kx - pad_x0;
        if (u < 0 || u % up_x != 0) continue;
        const int ix = u / up_x;
        if (ix >= in_w) continue;
        acc += kernel[(kh - 1 - ky) * kw + (kw - 1 - kx)] *
               to_f32(in[(((int64_t)ma * in_h + iy) * in_w + ix) * minor + mi]);
      }
    }
    out[t] = from_f32<T>(acc);
  }
}

constexpr int ADA_TW = 64;  // output columns per block

__global__ __launch_bounds__(256) void ada_apply_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                        const float* __restrict__ Ay, const float* __restrict__ kx,
                                                        const int* __restrict__ off, const int* __restrict__ sgn,
                                                        const float* __restrict__ a, const float* __restrict__ c,
                                                        int H, int W, int K, int transpose) {
  extern __shared__ float tmp[];  // [H][ADA_TW] horizontally filtered tile
  const int b = blockIdx.y;
  const int j0 = blockIdx.x * ADA_TW;
  const float* xb = x + (int64_t)b * H * W;
  const float* kb = kx + (int64_t)b * K;
  const int of = off[b], sg = sgn[b];
  // phase 1: t[h][j] = sum_t kx[t] * x[h][col(j, t)]
  for (int it = threadIdx.x; it < H * ADA_TW; it += blockDim.x) {
    const int jl = it % ADA_TW, h = it / ADA_TW;
    const int j = j0 + jl;
    float acc = 0.f;
    if (j < W) {
      for (int t = 0; t < K; ++t) {
        const int col = transpose ? floormod(sg * (j - of - t), W) : floormod(sg * j + of + t, W);
        acc += kb[t] * xb[(int64_t)h * W + col];
      }
    }
    tmp[it] = acc;
  }
  __syncthreads();
  // phase 2: y[i][j] = a * sum_h Ay[i][h] t[h][j] + c   (Ay^T and no offset for the transpose)
  const float* Ab = Ay + (int64_t)b * H * H;
  const float ab = a[b];
  const float cb = (transpose || !c) ? 0.f : c[b];
  for (int it = threadIdx.x; it < H * ADA_TW; it += blockDim.x) {
    const int jl = it % ADA_TW, i = it / ADA_TW;
    const int j = j0 + jl;
    if (j >= W) continue;
    float acc = 0.f;
    for (int h = 0; h < H; ++h) acc += (transpose ? Ab[h * H + i] : Ab[i * H + h]) * tmp[h * ADA_TW + jl];
    y[((int64_t)b * H + i) * W + j] = ab * acc + cb;
  }
}

}  // namespace

extern "C" int dgv2_upfirdn2d(void* out, const void* in, const float* kernel, int major, int in_h, int in_w, int minor,
                              int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                              int pad_y0, int pad_y1, int dtype, void* stream) {
  if (!out || !in || !kernel || major <= 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 || kw <= 0) return DGV2_EINVAL;
  if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1) return DGV2_EINVAL;
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
  const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
  if (out_h <= 0 || out_w <= 0) return DGV2_EINVAL;
  const int64_t total = (int64_t)major * out_h * out_w * minor;
  hipStream_t st = (hipStream_t)stream;
  DGV2_DISPATCH_DTYPE(dtype, {
    upfirdn2d_kernel<T><<<grid_for(total, 256), 256, 0, st>>>((T*)out, (const T*)in, kernel, major, in_h, in_w, minor,
                                                             kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0,
                                                             out_h, out_w);
  });
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_ada_apply(float* y, const float* x, const float* Ay, const float* kx, const int* off,
                              const int* sgn, const float* a, const float* c, int B, int H, int W, int K,
                              int transpose, void* stream) {
  if (!y || !x || !Ay || !kx || !off || !sgn || !a || B <= 0 || H <= 0 || W <= 0 || K <= 0) return DGV2_EINVAL;
  const size_t lds = sizeof(float) * (size_t)H * ADA_TW;
  if (lds > 64 * 1024) return DGV2_EINVAL;
  dim3 grid((W + ADA_TW - 1) / ADA_TW, B);
  ada_apply_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(y, x, Ay, kx, off, sgn, a, c, H, W, K, transpose);
  DGV2_RETURN_LAST();
}
