// Fourier-feature positional encoding of the laser angles and the angle pyramid.
// Reference: gans/models/ops/fourier.py:77-82 (1x1 conv on the 2-channel angle map + sin/cos)
// and gans/models/dusty_v2.py:135-140 (sin/cos -> FIR down-2 -> atan2).
//
// The encoding is written straight into a channel slice [c0, c0+2F) of a wider
// channels-last activation (the concat of dusty_v2.py:157-159 never happens).
#include "common.h"

namespace {

template <typename T>
__global__ void fourier_kernel(T* __restrict__ out, const float* __restrict__ angle, const float* __restrict__ shift,
                               const float* __restrict__ freqs, const float* __restrict__ phase, int B, int Ba,
                               int HW, int F, int ld, int c0) {
  // one thread = one (pixel, group of 8 frequencies); consecutive threads walk the
  // frequency groups of a pixel, so a wave writes contiguous channel runs.
  const int groups = (F + 7) / 8;
  const int64_t total = (int64_t)B * HW * groups;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % groups);
    const int64_t bp = t / groups;
    const int p = (int)(bp % HW);
    const int b = (int)(bp / HW);
    const int ba = Ba == 1 ? 0 : b;
    const float elev = angle[((int64_t)ba * 2 + 0) * HW + p];
    float azim = angle[((int64_t)ba * 2 + 1) * HW + p];
    if (shift) azim += shift[b];
    T* o = out + ((int64_t)b * HW + p) * ld + c0;
    const int f0 = g * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = f0 + j;
      if (f < F) {
        // same association order as conv2d's (w0*x0 + w1*x1) + bias
        const float c = freqs[2 * f] * elev + freqs[2 * f + 1] * azim + phase[f];
        float s, cs;
        sincosf(c, &s, &cs);
        o[f] = from_f32<T>(s);
        o[F + f] = from_f32<T>(cs);
      }
    }
  }
}

// out[b, ch, ho, wo] = atan2( sum taps*sin(in), sum taps*cos(in) ), 4x4 separable taps, stride 2,
// pads (1,1): circular along W if ring else replicate, replicate along H.
__global__ void downsample_angle_kernel(float* __restrict__ out, const float* __restrict__ in,
                                        const float* __restrict__ shift, const float* __restrict__ taps, int B,
                                        int Ba, int H, int W, int ring) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)B * 2 * Ho * Wo;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int wo = (int)(t % Wo);
    const int ho = (int)((t / Wo) % Ho);
    const int ch = (int)((t / ((int64_t)Wo * Ho)) % 2);
    const int b = (int)(t / ((int64_t)Wo * Ho * 2));
    const int ba = Ba == 1 ? 0 : b;
    const float* src = in + ((int64_t)ba * 2 + ch) * H * W;
    const float sh = (shift && ch == 1) ? shift[b] : 0.f;
    float ss = 0.f, cc = 0.f;
    for (int i = 0; i < 4; ++i) {
      int h = 2 * ho + i - 1;
      h = h < 0 ? 0 : (h >= H ? H - 1 : h);
      float rs = 0.f, rc = 0.f;
      for (int j = 0; j < 4; ++j) {
        int w = 2 * wo + j - 1;
        w = ring ? floormod(w, W) : (w < 0 ? 0 : (w >= W ? W - 1 : w));
        float s, c;
        sincosf(src[(int64_t)h * W + w] + sh, &s, &c);
        rs += taps[j] * s;
        rc += taps[j] * c;
      }
      ss += taps[i] * rs;
      cc += taps[i] * rc;
    }
    out[t] = atan2f(ss, cc);
  }
}

}  // namespace

extern "C" int dgv2_fourier_feature(void* out, const float* angle, const float* shift, const float* freqs,
                                    const float* phase, int B, int Ba, int H, int W, int F, int ld, int c0, int dtype,
                                    void* stream) {
  if (!out || !angle || !freqs || !phase || B <= 0 || H <= 0 || W <= 0 || F <= 0) return DGV2_EINVAL;
  if ((Ba != 1 && Ba != B) || c0 < 0 || ld < c0 + 2 * F) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * H * W * ((F + 7) / 8);
  DGV2_DISPATCH_DTYPE(dtype, {
    fourier_kernel<T><<<grid_for(total, 256, 256 * 32), 256, 0, st>>>((T*)out, angle, shift, freqs, phase, B, Ba,
                                                                     H * W, F, ld, c0);
  });
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_downsample_angle(float* out, const float* in, const float* shift, const float* taps, int B,
                                     int Ba, int H, int W, int ring, void* stream) {
  if (!out || !in || !taps || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return DGV2_EINVAL;
  if (Ba != 1 && Ba != B) return DGV2_EINVAL;
  const int64_t total = (int64_t)B * 2 * (H / 2) * (W / 2);
  downsample_angle_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(out, in, shift, taps, B, Ba, H, W,
                                                                                ring);
  DGV2_RETURN_LAST();
}
