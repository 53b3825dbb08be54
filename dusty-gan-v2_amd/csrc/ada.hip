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

// Generic version: any H (dynamic LDS = H * ADA_TW floats).  Kept for H > 64 or H % 4 != 0.
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

// LDS-staged version for H <= 64, H % 4 == 0, K <= 64 (the model's 64-row images): the block stages the
// 64 + K - 1 source columns its taps touch (ONE wrap computation per staged element instead of one integer
// modulo per tap), the sample's H x H operator and the taps in LDS; phase 1 then walks consecutive LDS words,
// phase 2 keeps H/4 output rows per thread in registers so each filtered value is read once per thread and the
// operator rows come as wave-uniform 16-byte broadcasts.
constexpr int ADA_HMAX = 64, ADA_KMAX = 64;
constexpr int ADA_XT = ADA_TW + ADA_KMAX;   // staged source columns (row pitch of the LDS tile)

__global__ __launch_bounds__(256) void ada_apply_lds_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ Ay, const float* __restrict__ kx,
                                                            const int* __restrict__ off, const int* __restrict__ sgn,
                                                            const float* __restrict__ a, const float* __restrict__ c,
                                                            int H, int W, int K, int transpose) {
  __shared__ float xt[ADA_HMAX][ADA_XT];
  __shared__ float tmp[ADA_HMAX][ADA_TW];
  __shared__ __attribute__((aligned(16))) float Al[ADA_HMAX][ADA_HMAX];   // Al[i][h] = operator row of OUTPUT row i
  __shared__ float kl[ADA_KMAX];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int j0 = blockIdx.x * ADA_TW;
  const float* xb = x + (int64_t)b * H * W;
  const float* Ab = Ay + (int64_t)b * H * H;
  const int of = off[b], sg = sgn[b];
  // source column of (jl, t): forward sg*(j0+jl) + of + t, transpose sg*((j0+jl) - of - t)  =  c0 + qa*jl + qb*t,
  // LDS column q = qa*jl + qb*t + qs >= 0
  const int qa = sg, qb = transpose ? -sg : 1;
  const int c0 = transpose ? sg * (j0 - of) : sg * j0 + of;
  const int qs = (qa < 0 ? ADA_TW - 1 : 0) + (qb < 0 ? K - 1 : 0);
  const int nq = ADA_TW + K - 1;   // <= 127
  // staging: a thread owns ONE source column (its wrap computed once) and half of the rows -- no integer division in the
  // loop, every load of a thread independent of the others (the element-per-iteration form with a division and a modulo per
  // element was latency bound: 34 us per launch for 8 MB)
  {
    const int q = tid & 127, hh = tid >> 7;
    if (q < nq) {
      const float* src = xb + floormod(c0 + q - qs, W);
      const int h0 = hh * (H / 2), h1 = hh ? H : H / 2;
#pragma unroll 8
      for (int h = h0; h < h1; ++h) xt[h][q] = src[(int64_t)h * W];
    }
  }
  // the operator: coalesced reads either way; the transpose is taken on the LDS side
  for (int it = tid; it < H * H; it += 256) {
    const int r = it / H, cidx = it - r * H;   // H = 64 on the model's path: shifts
    const float v = Ab[it];
    if (transpose) Al[cidx][r] = v;
    else Al[r][cidx] = v;
  }
  if (tid < K) kl[tid] = kx[(int64_t)b * K + tid];
  __syncthreads();
  // phase 1, four consecutive outputs of a row per item: LDS column of (output o, tap t) = qa * (4 jg + o) + qs + qb * t
  // = pb + o' + t' with o' = o or 3 - o and t' = t or K - 1 - t by the two signs -- a window that slides one source value and
  // one tap per step: two LDS reads per four FMAs (the one-output form read two per FMA and was bound by the LDS issue rate)
  for (int it = tid; it < H * (ADA_TW / 4); it += 256) {
    const int jg = it % (ADA_TW / 4), h = it / (ADA_TW / 4);
    const int pb = (qa > 0 ? 4 * jg : -4 * jg - 3) + qs + (qb > 0 ? 0 : -(K - 1));
    const float* xr = &xt[h][pb];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;
    for (int u = 0; u < K + 3; ++u) {
      const float xv = xr[u];
      k3 = k2; k2 = k1; k1 = k0;
      k0 = u < K ? kl[qb > 0 ? u : K - 1 - u] : 0.f;
      a0 = fmaf(k0, xv, a0);
      a1 = fmaf(k1, xv, a1);
      a2 = fmaf(k2, xv, a2);
      a3 = fmaf(k3, xv, a3);
    }
    float* tr = &tmp[h][4 * jg];
    if (qa > 0) { tr[0] = a0; tr[1] = a1; tr[2] = a2; tr[3] = a3; }
    else { tr[0] = a3; tr[1] = a2; tr[2] = a1; tr[3] = a0; }
  }
  __syncthreads();
  const int jl = tid % ADA_TW, ig = tid / ADA_TW;   // 4 row groups x 64 columns
  const int rpt = H / 4;                             // output rows per thread (<= 16)
  const int j = j0 + jl;
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int h = 0; h < H; h += 4) {
    const float t0 = tmp[h][jl], t1 = tmp[h + 1][jl], t2 = tmp[h + 2][jl], t3 = tmp[h + 3][jl];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < rpt) {
        const float4 av = *reinterpret_cast<const float4*>(&Al[ig * rpt + r][h]);
        acc[r] += av.x * t0 + av.y * t1 + av.z * t2 + av.w * t3;
      }
    }
  }
  if (j < W) {
    const float ab = a[b];
    const float cb = (transpose || !c) ? 0.f : c[b];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r < rpt) y[((int64_t)b * H + ig * rpt + r) * W + j] = ab * acc[r] + cb;
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
  static const bool no_lds = getenv("DGV2_NO_ADA_LDS") != nullptr;   // A/B switch for benchmarking
  if (!no_lds && H <= ADA_HMAX && H % 4 == 0 && K <= ADA_KMAX)
    ada_apply_lds_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(y, x, Ay, kx, off, sgn, a, c, H, W, K, transpose);
  else
    ada_apply_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(y, x, Ay, kx, off, sgn, a, c, H, W, K, transpose);
  DGV2_RETURN_LAST();
}
