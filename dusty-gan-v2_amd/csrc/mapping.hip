// The generator's mapping network (reference: gans/models/dusty_v2.py:13-29 -- PixelNorm, then `depth` x
// [EqualLR(nn.Linear, gain sqrt 2, lr_mul 0.01) + LeakyReLU(0.2)]) on a [B, 512] latent: 33 MFLOP per layer, i.e. pure
// launch latency as library calls (PixelNorm 5 launches, addmm + leaky_relu per layer; twice that in the backward).
// One launch per layer forward, two per layer backward:
//   forward   y[s, o] = lrelu( c1 * sum_k xh[s, k] W[o, k] + c2 * b[o] ),  xh = x / sqrt(mean_k x^2 + 1e-8) for the first layer
//             (c1 = EqualLR scale * gain * lr_mul on the product, c2 = gain * lr_mul on the bias: common.py:158-184)
//   backward  gp = gy * lrelu'(y);  gx[s, k] = c1 * sum_o gp[s, o] W[o, k];  gW[o, k] = c1 * sum_s gp[s, o] xh[s, k];  gb[o] = c2 * sum_s gp[s, o]
// fp32 throughout (the reference runs the mapping network in fp32).
#include "common.h"

namespace {

constexpr int MAP_SPB = 4;   // samples per block

// grid (O / 64, ceil(B / 4)), 64 threads: thread = output channel, the block's sample rows staged (and normalised) in LDS
template <bool NORM>
__global__ __launch_bounds__(64) void map_fwd_kernel(float* __restrict__ y, float* __restrict__ xh_out,
                                                     const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, int B, int K, int O, float c1, float c2,
                                                     float alpha) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [MAP_SPB][K]
  const int lane = threadIdx.x;
  const int s0 = blockIdx.y * MAP_SPB;
  const int o = blockIdx.x * 64 + lane;
#pragma unroll
  for (int s = 0; s < MAP_SPB; ++s) {
    const int sb = s0 + s < B ? s0 + s : B - 1;
    float ss = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)sb * K + k);
      *reinterpret_cast<float4*>(xs + s * K + k) = v;
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (NORM) {
      ss = wave_sum(ss);
      const float inv = 1.f / sqrtf(ss / (float)K + 1e-8f);
      for (int k = lane * 4; k < K; k += 256) {
        float4 v = *reinterpret_cast<float4*>(xs + s * K + k);
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        *reinterpret_cast<float4*>(xs + s * K + k) = v;
        if (xh_out && blockIdx.x == 0 && s0 + s < B) *reinterpret_cast<float4*>(xh_out + (int64_t)(s0 + s) * K + k) = v;
      }
    }
  }
  __syncthreads();
  if (o >= O) return;
  float acc[MAP_SPB];
#pragma unroll
  for (int s = 0; s < MAP_SPB; ++s) acc[s] = 0.f;
  const float* wr = w + (int64_t)o * K;
  for (int k = 0; k < K; k += 4) {
    const float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
    for (int s = 0; s < MAP_SPB; ++s) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + s * K + k);   // same address on every lane: broadcast
      acc[s] = fmaf(wv.x, xv.x, acc[s]);
      acc[s] = fmaf(wv.y, xv.y, acc[s]);
      acc[s] = fmaf(wv.z, xv.z, acc[s]);
      acc[s] = fmaf(wv.w, xv.w, acc[s]);
    }
  }
  const float bo = b ? b[o] * c2 : 0.f;
#pragma unroll
  for (int s = 0; s < MAP_SPB; ++s) {
    if (s0 + s >= B) break;
    const float v = fmaf(acc[s], c1, bo);
    y[(int64_t)(s0 + s) * O + o] = v > 0.f ? v : v * alpha;
  }
}

// gp = gy * lrelu'(y) and gx = c1 * gp W: grid (K / 64, ceil(B / 4)), 64 threads: thread = input channel k (W rows are
// read coalesced across the wave), gp rows of the block's samples in LDS.  gx may be NULL (first layer: only gp).
__global__ __launch_bounds__(64) void map_bwd_x_kernel(float* __restrict__ gx, float* __restrict__ gp_out,
                                                       const float* __restrict__ gy, const float* __restrict__ y,
                                                       const float* __restrict__ w, int B, int K, int O, float c1,
                                                       float alpha) {
  extern __shared__ __attribute__((aligned(16))) float gs[];   // [MAP_SPB][O]
  const int lane = threadIdx.x;
  const int s0 = blockIdx.y * MAP_SPB;
#pragma unroll
  for (int s = 0; s < MAP_SPB; ++s) {
    const bool live = s0 + s < B;
    for (int o = lane; o < O; o += 64) {
      float v = 0.f;
      if (live) {
        const int64_t i = (int64_t)(s0 + s) * O + o;
        v = gy[i] * (y[i] > 0.f ? 1.f : alpha);
        if (gp_out && blockIdx.x == 0) gp_out[i] = v;
      }
      gs[s * O + o] = v;
    }
  }
  __syncthreads();
  if (!gx) return;
  const int k = blockIdx.x * 64 + lane;
  if (k >= K) return;
  float acc[MAP_SPB];
#pragma unroll
  for (int s = 0; s < MAP_SPB; ++s) acc[s] = 0.f;
  for (int o = 0; o < O; ++o) {
    const float wv = w[(int64_t)o * K + k];
#pragma unroll
    for (int s = 0; s < MAP_SPB; ++s) acc[s] = fmaf(gs[s * O + o], wv, acc[s]);
  }
#pragma unroll
  for (int s = 0; s < MAP_SPB; ++s)
    if (s0 + s < B) gx[(int64_t)(s0 + s) * K + k] = acc[s] * c1;
}

// gW[o, k] = c1 * sum_s gp[s, o] x[s, k], gb[o] = c2 * sum_s gp[s, o]: grid (K / 256, O), 64 threads x 4 k each
__global__ __launch_bounds__(64) void map_bwd_w_kernel(float* __restrict__ gw, float* __restrict__ gb,
                                                       const float* __restrict__ gp, const float* __restrict__ x, int B,
                                                       int K, int O, float c1, float c2) {
  const int o = blockIdx.y;
  const int k = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (k >= K) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float sb = 0.f;
  for (int s = 0; s < B; ++s) {
    const float g = gp[(int64_t)s * O + o];          // same address on every lane
    const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)s * K + k);
    acc.x = fmaf(g, xv.x, acc.x);
    acc.y = fmaf(g, xv.y, acc.y);
    acc.z = fmaf(g, xv.z, acc.z);
    acc.w = fmaf(g, xv.w, acc.w);
    sb += g;
  }
  acc.x *= c1; acc.y *= c1; acc.z *= c1; acc.w *= c1;
  *reinterpret_cast<float4*>(gw + (int64_t)o * K + k) = acc;
  if (gb && k == 0) gb[o] = sb * c2;
}

}  // namespace

// One layer of the mapping network, forward: y [B, O] = lrelu(c1 * xh W^T + c2 * b), xh = PixelNorm(x) when norm != 0
// (then also written to xh_out [B, K] when non-NULL: the backward's operand) else x.  x [B, K], w [O, K], b [O] or NULL,
// fp32; K % 4 == 0, O % 64 == 0, K <= 4096.
// replaces: ops.PixelNorm + EqualLR(nn.Linear) + nn.LeakyReLU, gans/models/dusty_v2.py:13-29, ops/common.py:158-184,213-223.
extern "C" int dgv2_map_layer_fwd(float* y, float* xh_out, const float* x, const float* w, const float* b, int B, int K,
                                  int O, float c1, float c2, float alpha, int norm, void* stream) {
  if (!y || !x || !w || B <= 0 || K <= 0 || O <= 0) return DGV2_EINVAL;
  if ((K & 3) || (O & 63) || K > 4096) return DGV2_ENOTSUP;
  if (!aligned16(x) || !aligned16(w) || (xh_out && !aligned16(xh_out))) return DGV2_EINVAL;
  dim3 grid(O / 64, (B + MAP_SPB - 1) / MAP_SPB);
  const size_t lds = sizeof(float) * MAP_SPB * K;
  hipStream_t st = (hipStream_t)stream;
  if (norm) map_fwd_kernel<true><<<grid, 64, lds, st>>>(y, xh_out, x, w, b, B, K, O, c1, c2, alpha);
  else map_fwd_kernel<false><<<grid, 64, lds, st>>>(y, nullptr, x, w, b, B, K, O, c1, c2, alpha);
  DGV2_RETURN_LAST();
}

// The same layer, backward: gp [B, O] = gy * lrelu'(y) (from the stored output y), gx [B, K] = c1 * gp W (or NULL: the
// first layer's input needs no gradient), gw [O, K] = c1 * gp^T x, gb [O] = c2 * sum_s gp (or NULL).  x: the operand the
// forward contracted (xh_out for the first layer).
extern "C" int dgv2_map_layer_bwd(float* gx, float* gw, float* gb, float* gp, const float* gy, const float* y,
                                  const float* x, const float* w, int B, int K, int O, float c1, float c2, float alpha,
                                  void* stream) {
  if (!gw || !gp || !gy || !y || !x || !w || B <= 0 || K <= 0 || O <= 0) return DGV2_EINVAL;
  if ((K & 63) || (O & 3) || O > 4096) return DGV2_ENOTSUP;
  if (!aligned16(x) || !aligned16(gw)) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 gridx(gx ? K / 64 : 1, (B + MAP_SPB - 1) / MAP_SPB);
  map_bwd_x_kernel<<<gridx, 64, sizeof(float) * MAP_SPB * O, st>>>(gx, gp, gy, y, w, B, K, O, c1, alpha);
  dim3 gridw((K / 4 + 63) / 64, O);
  map_bwd_w_kernel<<<gridw, 64, 0, st>>>(gw, gb, gp, x, B, K, O, c1, c2);
  DGV2_RETURN_LAST();
}
