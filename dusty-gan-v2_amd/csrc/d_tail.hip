// The last two modules of the discriminator's epilogue on the [B, K] output of its 65536 -> 512 Linear as ONE launch
// forward and ONE backward (reference: ops.FusedLeakyReLU(ch(4)) + ops.EqualLR(nn.Linear(ch(4), 1)),
// gans/models/dusty_v2.py:383-384; fused_leaky_relu fused_act.py:20-59,113-129; EqualLR common.py:158-184):
//   a[b, k] = lrelu_alpha(h[b, k] + b1[k]) * act_scale
//   y[b]    = gain2 * (b2 + scale2 * sum_k w2[k] a[b, k])
// Composed from the generic ops this was bias_act + a scaling launch + a library GEMM with one output column (18 us) forward
// and two more library GEMMs, two elementwise launches, a reduction, a fill and bias_act_bwd backward: 9 - 11 launches of
// ~5 us on 64 - 128 x 512 numbers per step body.  fp32 throughout; sums in a fixed order (no atomics).
#include "common.h"

namespace {

// one block per sample
__global__ __launch_bounds__(256) void d_tail_fwd_kernel(float* __restrict__ y, float* __restrict__ a,
                                                         const float* __restrict__ h, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, int K,
                                                         float alpha, float act_scale, float scale2, float gain2) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  const float* hb = h + (int64_t)b * K;
  float* ab = a + (int64_t)b * K;
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v = hb[k] + (b1 ? b1[k] : 0.f);
    const float o = (v > 0.f ? v : v * alpha) * act_scale;
    ab[k] = o;
    s = fmaf(o, w2[k], s);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) y[b] = gain2 * ((b2 ? b2[0] : 0.f) + scale2 * s);
}

// one thread per feature column k, walking the batch; block 0 also sums gy for the output bias
__global__ __launch_bounds__(256) void d_tail_bwd_kernel(float* __restrict__ gh, float* __restrict__ gb1,
                                                         float* __restrict__ gw2, float* __restrict__ gb2,
                                                         const float* __restrict__ gy, const float* __restrict__ a,
                                                         const float* __restrict__ w2, int B, int K, float alpha,
                                                         float act_scale, float scale2, float gain2) {
  __shared__ float red[16];
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < K) {
    const float c = gain2 * scale2 * w2[k];
    const float up = act_scale, dn = alpha * act_scale;
    float sb = 0.f, sw = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) {
      const float g = gy[b], o = a[(int64_t)b * K + k];
      const float d = g * c * (o > 0.f ? up : dn);     // the mask is the sign of the OUTPUT (fused_bias_act_kernel.cu:19-60)
      gh[(int64_t)b * K + k] = d;
      sb += d;
      sw = fmaf(g, o, sw);
    }
    if (gb1) gb1[k] = sb;
    if (gw2) gw2[k] = gain2 * scale2 * sw;
  }
  if (gb2 && blockIdx.x == 0) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += gy[b];
    s = block_sum(s, red);
    if (threadIdx.x == 0) gb2[0] = gain2 * s;
  }
}

// dst[k] <- lerp(dst[k], mean_b src[b * ld + k], w): the running mean of the mapped latents (base.py:89-97), one launch
__global__ __launch_bounds__(256) void colmean_lerp_kernel(float* __restrict__ dst, const float* __restrict__ src, int B,
                                                           int K, int64_t ld, float w) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
#pragma unroll 8
  for (int b = 0; b < B; ++b) s += src[(int64_t)b * ld + k];
  const float d = dst[k];
  dst[k] = d + w * (s / (float)B - d);
}

}  // namespace

// dst fp32 [K] <- lerp(dst, mean over the B rows of src (fp32, row pitch ld >= K), w)
// replaces: Generator.moving_average_w (gans/models/base.py:89-97: w.mean(0) + lerp) -- a reduction and a lerp launch
extern "C" int dgv2_colmean_lerp(float* dst, const float* src, int B, int K, int64_t ld, float w, void* stream) {
  if (!dst || !src || B < 1 || K < 1 || ld < K) return DGV2_EINVAL;
  colmean_lerp_kernel<<<(K + 255) / 256, 256, 0, (hipStream_t)stream>>>(dst, src, B, K, ld, w);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_d_tail_fwd(float* y, float* a, const float* h, const float* b1, const float* w2, const float* b2, int B,
                               int K, float alpha, float act_scale, float scale2, float gain2, void* stream) {
  if (!y || !a || !h || !w2 || B < 1 || K < 1) return DGV2_EINVAL;
  d_tail_fwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(y, a, h, b1, w2, b2, K, alpha, act_scale, scale2, gain2);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_d_tail_bwd(float* gh, float* gb1, float* gw2, float* gb2, const float* gy, const float* a,
                               const float* w2, int B, int K, float alpha, float act_scale, float scale2, float gain2,
                               void* stream) {
  if (!gh || !gy || !a || !w2 || B < 1 || K < 1) return DGV2_EINVAL;
  d_tail_bwd_kernel<<<(K + 255) / 256, 256, 0, (hipStream_t)stream>>>(gh, gb1, gw2, gb2, gy, a, w2, B, K, alpha, act_scale,
                                                                   scale2, gain2);
  DGV2_RETURN_LAST();
}
