// Small HBM-bound fp32 kernels at the two ends of the path:
//  * generator output stage (shift cancel + scale + tanh + Gumbel-sigmoid ray-drop blend),
//    reference: gans/models/dusty_v2.py:290-306, gans/models/dusty_v1.py:20-25,
//    gans/models/ops/gumbel.py:23-29;
//  * range-image projection, reference: gans/coords.py:73-185, gans/trainer.py:211-217;
//  * sum of squares (input-magnitude EMA of ModConv2d, gans/models/ops/style.py:100-101).
#include "common.h"

namespace {

constexpr float TWO_PI = 6.283185307179586f;

// The reference cancels the azimuth shift with affine_grid + grid_sample on [v, v]: for a pure
// translation that is a circular linear interpolation at j + s/(2 pi) * W (oracle/ops.py
// ring_shift).  k = integer part, f = fraction; identical for every j of a sample.
__device__ __forceinline__ void shift_split(float shift_rad, int W, int& k, float& f) {
  const float t = shift_rad / TWO_PI * (float)W;
  const float fl = floorf(t);
  k = (int)fl;
  f = t - fl;
}

__global__ void gen_tail_fwd_kernel(float* __restrict__ image, float* __restrict__ image_orig,
                                    float* __restrict__ logit, float* __restrict__ mask,
                                    const float* __restrict__ skip, const float* __restrict__ shift,
                                    const float* __restrict__ u, int B, int H, int W, float out_scale,
                                    float raydrop_const, float temperature) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(t % W);
    const int64_t row = t / W;
    const int b = (int)(row / H);
    float v0, v1;
    if (shift) {
      int k;
      float f;
      shift_split(shift[b], W, k, f);
      const int j0 = floormod(w + k, W), j1 = floormod(w + k + 1, W);
      const float* r = skip + row * W * 2;
      v0 = r[j0 * 2 + 0] * (1.f - f) + r[j1 * 2 + 0] * f;
      v1 = r[j0 * 2 + 1] * (1.f - f) + r[j1 * 2 + 1] * f;
    } else {
      v0 = skip[t * 2 + 0];
      v1 = skip[t * 2 + 1];
    }
    const float img = tanhf(v0 * out_scale);
    const float lg = v1 * out_scale;
    const float uu = u[t];
    const float soft = 1.f / (1.f + expf(-(lg + logf(uu) - log1pf(-uu)) / temperature));
    const float m = soft > 0.5f ? 1.f : 0.f;
    image_orig[t] = img;
    logit[t] = lg;
    mask[t] = m;
    image[t] = img + (1.f - m) * (raydrop_const - img);
  }
}

// First pass of the backward: gradient w.r.t. the shifted/scaled pre-activations, written as
// [B,H,W,2] into `gv` (= g_skip when there is no shift).
__global__ void gen_tail_bwd_kernel(float* __restrict__ gv, const float* __restrict__ g_image,
                                    const float* __restrict__ g_image_orig, const float* __restrict__ g_logit,
                                    const float* __restrict__ g_mask, const float* __restrict__ image_orig,
                                    const float* __restrict__ logit, const float* __restrict__ mask,
                                    const float* __restrict__ u, int64_t total, float out_scale,
                                    float raydrop_const, float temperature) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const float img = image_orig[t];
    const float m = mask[t];
    const float gi = g_image ? g_image[t] : 0.f;
    // image = img + (1-m)(c - img): d/dimg = m, d/dm = img - c
    float g_img = gi * m + (g_image_orig ? g_image_orig[t] : 0.f);
    float gm = gi * (img - raydrop_const) + (g_mask ? g_mask[t] : 0.f);
    // straight-through: dm/dlogit = soft (1 - soft) / temperature
    const float uu = u[t];
    const float soft = 1.f / (1.f + expf(-(logit[t] + logf(uu) - log1pf(-uu)) / temperature));
    float gl = gm * soft * (1.f - soft) / temperature + (g_logit ? g_logit[t] : 0.f);
    gv[t * 2 + 0] = g_img * (1.f - img * img) * out_scale;
    gv[t * 2 + 1] = gl * out_scale;
  }
}

// Transpose of the circular interpolation: g_skip[m] = (1-f) gv[m-k] + f gv[m-k-1].
__global__ void ring_shift_adjoint_kernel(float* __restrict__ g_skip, const float* __restrict__ gv,
                                          const float* __restrict__ shift, int B, int H, int W) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(t % W);
    const int64_t row = t / W;
    const int b = (int)(row / H);
    int k;
    float f;
    shift_split(shift[b], W, k, f);
    const int j0 = floormod(w - k, W), j1 = floormod(w - k - 1, W);
    const float* r = gv + row * W * 2;
    g_skip[t * 2 + 0] = r[j0 * 2 + 0] * (1.f - f) + r[j1 * 2 + 0] * f;
    g_skip[t * 2 + 1] = r[j0 * 2 + 1] * (1.f - f) + r[j1 * 2 + 1] * f;
  }
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ float inv_depth_norm_from_depth(float d, float min_d, float max_d) {
  const bool valid = (d >= min_d) && (d <= max_d) && (d > 0.f);
  return valid ? (1.f / (d + 1e-11f)) * min_d : 0.f;
}

__device__ __forceinline__ float depth_from_inv_depth_norm(float x, float min_d, float max_d) {
  const float inv = x / min_d;
  const bool valid = (inv >= 1.f / max_d) && (inv <= 1.f / min_d) && (inv > 0.f);
  return valid ? 1.f / (inv + 1e-11f) : 0.f;
}

__global__ void coords_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ mask,
                              const float* __restrict__ angle, int B, int HW, float min_d, float max_d,
                              float raydrop_const, int mode) {
  const int64_t total = (int64_t)B * HW;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const float x = in[t];
    if (mode == 0) {
      float v = inv_depth_norm_from_depth(x, min_d, max_d);
      if (mask) {
        const float m = mask[t];
        v = m * (v * 2.f - 1.f) + (1.f - m) * raydrop_const;
      }
      out[t] = v;
    } else if (mode == 1) {
      out[t] = depth_from_inv_depth_norm(x, min_d, max_d);
    } else {
      float d = x;
      if (mode == 2) d = (x > 1e-11f) ? depth_from_inv_depth_norm(x, min_d, max_d) : 0.f;
      const int p = (int)(t % HW);
      const int64_t b = t / HW;
      float se, ce, sa, ca;
      sincosf(angle[p], &se, &ce);
      sincosf(angle[HW + p], &sa, &ca);
      float* o = out + b * 3 * HW + p;
      o[0] = d * ce * ca;
      o[HW] = d * ce * sa;
      o[2 * (int64_t)HW] = d * se;
    }
  }
}

// Per-block partial sums: acc[0 .. SUMSQ_SLOTS) is OVERWRITTEN (slots past the grid with 0) and its entries add
// up to the result -- no zero fill before, no same-address atomics (512 of them cost ~25 us per call).
constexpr int SUMSQ_SLOTS = 512;

template <typename T>
__global__ void sum_squares_kernel(float* __restrict__ acc, const T* __restrict__ x, int64_t N, int C, int ld) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t total = N * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / C;
    const int c = (int)(t % C);
    const float v = to_f32(x[r * ld + c]);
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) acc[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (blockIdx.x == 0)
    for (int k = gridDim.x + threadIdx.x; k < SUMSQ_SLOTS; k += blockDim.x) acc[k] = 0.f;
}

template <typename T>
__global__ void sum_squares_vec_kernel(float* __restrict__ acc, const T* __restrict__ x, int64_t N, int cvecs,
                                       int ld) {
  constexpr int VN = vec16<T>::N;
  __shared__ float red[4];
  float s = 0.f;
  const int64_t total = N * cvecs;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  constexpr int U = 4;   // independent 16-byte loads in flight per thread (512 blocks must cover HBM latency)
  for (int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t0 < total; t0 += stride * U) {
    vec16<T> v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = t0 + u * stride;
      v[u].raw = make_uint4(0, 0, 0, 0);
      if (t < total) {
        const int64_t r = cvecs == ld / VN ? 0 : t / cvecs;   // dense rows: one flat index
        const int64_t off = cvecs == ld / VN ? t * VN : r * ld + (t - r * cvecs) * VN;
        v[u].load(x + off);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        const float f = v[u].get(j);
        s += f * f;
      }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) acc[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (blockIdx.x == 0)
    for (int k = gridDim.x + threadIdx.x; k < SUMSQ_SLOTS; k += blockDim.x) acc[k] = 0.f;
}

}  // namespace

extern "C" int dgv2_gen_tail_fwd(float* image, float* image_orig, float* logit, float* mask, const float* skip,
                                 const float* shift, const float* u, int B, int H, int W, float out_scale,
                                 float raydrop_const, float temperature, void* stream) {
  if (!image || !image_orig || !logit || !mask || !skip || !u || B <= 0 || H <= 0 || W <= 0) return DGV2_EINVAL;
  const int64_t total = (int64_t)B * H * W;
  gen_tail_fwd_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      image, image_orig, logit, mask, skip, shift, u, B, H, W, out_scale, raydrop_const, temperature);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_gen_tail_bwd(float* g_skip, float* scratch, const float* g_image, const float* g_image_orig,
                                 const float* g_logit, const float* g_mask, const float* image_orig,
                                 const float* logit, const float* mask, const float* u, const float* shift, int B,
                                 int H, int W, float out_scale, float raydrop_const, float temperature,
                                 void* stream) {
  if (!g_skip || !image_orig || !logit || !mask || !u || B <= 0 || H <= 0 || W <= 0) return DGV2_EINVAL;
  if (shift && !scratch) return DGV2_EINVAL;
  const int64_t total = (int64_t)B * H * W;
  hipStream_t st = (hipStream_t)stream;
  float* gv = shift ? scratch : g_skip;
  gen_tail_bwd_kernel<<<grid_for(total, 256), 256, 0, st>>>(gv, g_image, g_image_orig, g_logit, g_mask, image_orig,
                                                           logit, mask, u, total, out_scale, raydrop_const,
                                                           temperature);
  if (shift) ring_shift_adjoint_kernel<<<grid_for(total, 256), 256, 0, st>>>(g_skip, gv, shift, B, H, W);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_coords_convert(float* out, const float* in, const float* mask, const float* angle, int B, int H,
                                   int W, float min_depth, float max_depth, float raydrop_const, int mode,
                                   void* stream) {
  if (!out || !in || B <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 3) return DGV2_EINVAL;
  if (mode >= 2 && !angle) return DGV2_EINVAL;
  const int64_t total = (int64_t)B * H * W;
  coords_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(out, in, mask, angle, B, H * W, min_depth,
                                                                      max_depth, raydrop_const, mode);
  DGV2_RETURN_LAST();
}

// acc: fp32 [512], overwritten with per-block partial sums (sum them, or hand them to dgv2_ema_scalar).
extern "C" int dgv2_sum_squares(float* acc, const void* x, int64_t N, int C, int ld, int dtype, void* stream) {
  if (!acc || !x || N <= 0 || C <= 0 || ld < C) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  DGV2_DISPATCH_DTYPE(dtype, {
    constexpr int VN = vec16<T>::N;
    if (C % VN == 0 && ld % VN == 0 && aligned16(x)) {
      const int cvecs = C / VN;
      sum_squares_vec_kernel<T><<<grid_for(N * cvecs, 256 * 8, 512), 256, 0, st>>>(acc, (const T*)x, N, cvecs, ld);
    } else {
      sum_squares_kernel<T><<<grid_for(N * C, 256 * 8, 512), 256, 0, st>>>(acc, (const T*)x, N, C, ld);
    }
  });
  DGV2_RETURN_LAST();
}
