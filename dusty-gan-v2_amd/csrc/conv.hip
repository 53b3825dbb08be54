// Ring-padded dense convolutions of the discriminator as implicit GEMMs on the MFMA engines.
// Reference: ops.Conv2d = Pad(circular W, replicate H) + nn.Conv2d (gans/models/ops/common.py:10-24,
// 187-210), used by ResidualBlock / Discriminator (gans/models/dusty_v2.py:325-385).  The padding
// is folded into the im2col address computation; nothing padded is materialised in the forward
// and weight-gradient passes.  The data gradient is produced in the padded domain and folded back
// (transpose of the ring / replicate extension) by a small gather kernel.
#include "gemm_core.h"

namespace {

template <typename T, int TO>
int launch_conv_fwd(void* y, const void* x, const void* w, const ConvGeom& g, const float* bias, int act,
                    float alpha, float scale, hipStream_t st) {
  constexpr int CE = 16 / sizeof(T);
  const int K = g.kh * g.kw * g.C;
  const int npix = g.B * g.Ho * g.Wo;
  DenseRowLoader<T> al{(const T*)w, 0, K, g.O, K, (K % CE == 0) && aligned16(w)};
  Im2colFwdLoader<T> bl{(const T*)x, g, npix, K, (g.C % CE == 0) && aligned16(x)};
  StoreEpilogue<T> epi{(T*)y, 0, g.O, g.O, npix, (g.O % 4 == 0) && aligned16(y), bias, act, alpha, scale};
  dim3 grid((npix + 127) / 128, (g.O + TO - 1) / TO, 1);
  gemm_nn_kernel<T, TO, DenseRowLoader<T>, Im2colFwdLoader<T>, StoreEpilogue<T>><<<grid, 256, 0, st>>>(al, bl, epi, K);
  return 0;
}

template <typename T, int TO>
int launch_conv_dgrad(void* gxp, const void* gy, const void* wt, const ConvGeom& g, hipStream_t st) {
  constexpr int CE = 16 / sizeof(T);
  const int K = g.kh * g.kw * g.O;
  const int Hp = g.H + 2 * g.pad, Wp = g.W + 2 * g.pad;
  const int npix = g.B * Hp * Wp;
  DenseRowLoader<T> al{(const T*)wt, 0, K, g.C, K, (K % CE == 0) && aligned16(wt)};
  Im2colDgradLoader<T> bl{(const T*)gy, g, Hp, Wp, npix, K, (g.O % CE == 0) && aligned16(gy)};
  StoreEpilogue<T> epi{(T*)gxp, 0, g.C, g.C, npix, (g.C % 4 == 0) && aligned16(gxp), nullptr, 0, 0.f, 1.f};
  dim3 grid((npix + 127) / 128, (g.C + TO - 1) / TO, 1);
  gemm_nn_kernel<T, TO, DenseRowLoader<T>, Im2colDgradLoader<T>, StoreEpilogue<T>><<<grid, 256, 0, st>>>(al, bl, epi, K);
  return 0;
}

// gx[b,h,w,:] = sum of gxp over every padded position that the extension maps to (h,w).
template <typename T>
__global__ void pad_fold_kernel(T* __restrict__ gx, const T* __restrict__ gxp, int B, int H, int W, int C, int pad,
                                int ring) {
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const int64_t pix = t / C;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int b = (int)(pix / ((int64_t)W * H));
    float acc = 0.f;
    for (int hp = 0; hp < Hp; ++hp) {
      const int hs = hp - pad;
      const int hm = hs < 0 ? 0 : (hs >= H ? H - 1 : hs);
      if (hm != h) continue;
      for (int wp = 0; wp < Wp; ++wp) {
        const int ws = wp - pad;
        const int wm = ring ? floormod(ws, W) : (ws < 0 ? 0 : (ws >= W ? W - 1 : ws));
        if (wm != w) continue;
        acc += to_f32(gxp[(((int64_t)b * Hp + hp) * Wp + wp) * C + c]);
      }
    }
    gx[t] = from_f32<T>(acc);
  }
}

// Cheap version of the fold for the common case (pad small): candidate positions enumerated
// directly instead of scanning the padded axis.
template <typename T>
__global__ void pad_fold_fast_kernel(T* __restrict__ gx, const T* __restrict__ gxp, int B, int H, int W, int C,
                                     int pad, int ring) {
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const int64_t pix = t / C;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int b = (int)(pix / ((int64_t)W * H));
    // rows: h+pad always; the replicate border rows collapse onto h == 0 / h == H-1
    const int h_lo = (h == 0) ? 0 : h + pad;
    const int h_hi = (h == H - 1) ? Hp - 1 : h + pad;
    float acc = 0.f;
    for (int hp = h_lo; hp <= h_hi; ++hp) {
      const T* row = gxp + ((int64_t)b * Hp + hp) * Wp * C + c;
      if (ring) {
        // wp = w + pad + m*W for every integer m with 0 <= wp < Wp
        for (int wp = (w + pad) % W; wp < Wp; wp += W) acc += to_f32(row[(int64_t)wp * C]);
      } else {
        const int w_lo = (w == 0) ? 0 : w + pad;
        const int w_hi = (w == W - 1) ? Wp - 1 : w + pad;
        for (int wp = w_lo; wp <= w_hi; ++wp) acc += to_f32(row[(int64_t)wp * C]);
      }
    }
    gx[t] = from_f32<T>(acc);
  }
}

template <typename T, int TO, int TJ>
int launch_conv_wgrad(float* gw, const void* gy, const void* x, const ConvGeom& g, int ksplit, hipStream_t st) {
  constexpr int CE = 16 / sizeof(T);
  const int J = g.kh * g.kw * g.C;
  const int64_t K = (int64_t)g.B * g.Ho * g.Wo;
  DenseKLoader<T> al{(const T*)gy, 0, g.O, g.O, (g.O % CE == 0) && aligned16(gy)};
  Im2colWgradLoader<T> bl{(const T*)x, g, J, (g.C % CE == 0) && aligned16(x)};
  const int64_t klen = ((K + ksplit - 1) / ksplit + 31) / 32 * 32;
  dim3 grid((J + TJ - 1) / TJ, (g.O + TO - 1) / TO, ksplit);
  gemm_tn_kernel<T, TO, TJ, DenseKLoader<T>, Im2colWgradLoader<T>><<<grid, 256, 0, st>>>(al, bl, gw, g.O, J, K, klen,
                                                                                        ksplit, 0, J);
  return 0;
}

bool geom_ok(const ConvGeom& g) {
  return g.B > 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.O > 0 && g.kh > 0 && g.kw > 0 && g.stride > 0 && g.pad >= 0 &&
         g.Ho > 0 && g.Wo > 0 && (!g.ring || g.pad <= g.W);
}

ConvGeom make_geom(int B, int H, int W, int C, int O, int kh, int kw, int stride, int pad, int ring) {
  ConvGeom g{B, H, W, C, O, 0, 0, kh, kw, stride, pad, ring};
  g.Ho = (H + 2 * pad - kh) / stride + 1;
  g.Wo = (W + 2 * pad - kw) / stride + 1;
  return g;
}

}  // namespace

extern "C" int dgv2_conv_fwd(void* y, const void* x, const void* w, int B, int H, int W, int C, int O, int kh, int kw,
                             int stride, int pad, int ring, const float* bias, int act, float alpha, float scale,
                             int dtype, void* stream) {
  const ConvGeom g = make_geom(B, H, W, C, O, kh, kw, stride, pad, ring);
  if (!y || !x || !w || !geom_ok(g) || (act != 0 && act != 3)) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  DGV2_DISPATCH_DTYPE(dtype, {
    if (O <= 16) launch_conv_fwd<T, 16>(y, x, w, g, bias, act, alpha, scale, st);
    else if (O <= 32) launch_conv_fwd<T, 32>(y, x, w, g, bias, act, alpha, scale, st);
    else if (O <= 64) launch_conv_fwd<T, 64>(y, x, w, g, bias, act, alpha, scale, st);
    else launch_conv_fwd<T, 128>(y, x, w, g, bias, act, alpha, scale, st);
  });
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_conv_dgrad(void* gx, void* gxp_scratch, const void* gy, const void* wt, int B, int H, int W,
                               int C, int O, int kh, int kw, int stride, int pad, int ring, int dtype, void* stream) {
  const ConvGeom g = make_geom(B, H, W, C, O, kh, kw, stride, pad, ring);
  if (!gx || !gy || !wt || !geom_ok(g) || (pad > 0 && !gxp_scratch)) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  void* gxp = pad > 0 ? gxp_scratch : gx;
  DGV2_DISPATCH_DTYPE(dtype, {
    if (C <= 16) launch_conv_dgrad<T, 16>(gxp, gy, wt, g, st);
    else if (C <= 32) launch_conv_dgrad<T, 32>(gxp, gy, wt, g, st);
    else if (C <= 64) launch_conv_dgrad<T, 64>(gxp, gy, wt, g, st);
    else launch_conv_dgrad<T, 128>(gxp, gy, wt, g, st);
    if (pad > 0) {
      const int64_t total = (int64_t)B * H * W * C;
      if (pad < H && pad < W)
        pad_fold_fast_kernel<T><<<grid_for(total, 256), 256, 0, st>>>((T*)gx, (const T*)gxp, B, H, W, C, pad, ring);
      else
        pad_fold_kernel<T><<<grid_for(total, 256), 256, 0, st>>>((T*)gx, (const T*)gxp, B, H, W, C, pad, ring);
    }
  });
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_conv_wgrad(float* gw, const void* gy, const void* x, int B, int H, int W, int C, int O, int kh,
                               int kw, int stride, int pad, int ring, int dtype, void* stream) {
  const ConvGeom g = make_geom(B, H, W, C, O, kh, kw, stride, pad, ring);
  if (!gw || !gy || !x || !geom_ok(g)) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int J = kh * kw * C;
  const int64_t K = (int64_t)B * g.Ho * g.Wo;
  const int tiles = ((O + 63) / 64) * ((J + 127) / 128);
  int ksplit = 1;
  while (tiles * ksplit < 1024 && K / (ksplit * 2) >= 256) ksplit *= 2;
  if (ksplit > 1) {
    hipError_t e = hipMemsetAsync(gw, 0, sizeof(float) * (size_t)O * J, st);
    if (e != hipSuccess) return (int)e;
  }
  DGV2_DISPATCH_DTYPE(dtype, {
    if (O <= 16) launch_conv_wgrad<T, 16, 128>(gw, gy, x, g, ksplit, st);
    else if (O <= 32) launch_conv_wgrad<T, 32, 128>(gw, gy, x, g, ksplit, st);
    else launch_conv_wgrad<T, 64, 128>(gw, gy, x, g, ksplit, st);
  });
  DGV2_RETURN_LAST();
}
