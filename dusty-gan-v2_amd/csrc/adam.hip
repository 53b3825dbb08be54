// Adam over a LIST of parameter tensors in one launch (torch.optim.Adam semantics, no weight decay / amsgrad;
// reference: the optimizers of gans/trainer.py:142-171).  The parameters, their gradients (views of the flat
// all-reduce buffer) and the moment tensors of torch's own optimizer state stay separate tensors; their addresses
// travel by value in the kernel arguments, so the launch is hipGraph-capturable and the optimizer's state_dict keeps
// torch's layout.  Seven fp32 streams per element (p, g, m, v read; p, m, v written): a pure HBM kernel.
// beta1 == 0 (both optimizers of the reference's config, dusty_v2.yaml lr.*.beta1 = 0.0): the first moment IS the gradient
// (torch's lerp(m, g, 1) returns g exactly), so m is written but never read -- six streams (B1ZERO instance).
//   m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// The step counter lives on the device (graph replay): dgv2_adam_prep advances it and leaves the two bias
// corrections of that step in a 4-float scratch that the update kernel reads.
#include "common.h"

namespace {

constexpr int AD_MAX = 72;

struct AdamArgs {
  float* p[AD_MAX];
  const float* g[AD_MAX];
  float* m[AD_MAX];
  float* v[AD_MAX];
  int n[AD_MAX];
};

__global__ void adam_prep_kernel(float* __restrict__ sc, float* __restrict__ step, float b1, float b2) {
  const float s = step[0] + 1.f;
  step[0] = s;
  sc[0] = s;
  sc[1] = 1.f - powf(b1, s);          // bias_correction1
  sc[2] = sqrtf(1.f - powf(b2, s));   // sqrt(bias_correction2)
}

template <bool B1ZERO>
__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a, const float* __restrict__ sc, float lr, float b1,
                                                        float b2, float eps) {
  const int l = blockIdx.y;
  const int n = a.n[l];
  float* __restrict__ p = a.p[l];
  const float* __restrict__ g = a.g[l];
  float* __restrict__ m = a.m[l];
  float* __restrict__ v = a.v[l];
  const float step_size = lr / sc[1], bc2s = sc[2];
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    mm = B1ZERO ? gg : mm + (1.f - b1) * (gg - mm);
    vv = b2 * vv + (1.f - b2) * gg * gg;
    pp -= step_size * mm / (sqrtf(vv) / bc2s + eps);
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const int stride = gridDim.x * 256;
  if (vec) {
    const int n4 = n >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      float4 pp = reinterpret_cast<float4*>(p)[i], vv = reinterpret_cast<float4*>(v)[i];
      const float4 gg = reinterpret_cast<const float4*>(g)[i];
      float4 mm = gg;
      if constexpr (!B1ZERO) mm = reinterpret_cast<float4*>(m)[i];
      upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
      reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int i = (n4 << 2) + blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      float mm = B1ZERO ? 0.f : m[i];
      upd(p[i], g[i], mm, v[i]);
      m[i] = mm;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      float mm = B1ZERO ? 0.f : m[i];
      upd(p[i], g[i], mm, v[i]);
      m[i] = mm;
    }
  }
}

struct LerpArgs {
  float* dst[AD_MAX];
  const float* src[AD_MAX];
  int n[AD_MAX];
};

__global__ __launch_bounds__(256) void lerp_list_kernel(LerpArgs a, float w) {
  const int l = blockIdx.y;
  float* __restrict__ d = a.dst[l];
  const float* __restrict__ s = a.src[l];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n[l]; i += gridDim.x * 256) d[i] += w * (s[i] - d[i]);
}

}  // namespace

// dst[l] <- lerp(dst[l], src[l], weight) for L <= 72 fp32 tensors in one launch: the G_ema update
// (ema_inplace, gans/trainer.py:30-41).  HOST arrays of device pointers.
extern "C" int dgv2_lerp_list(float* const* dst, const float* const* src, const int* n, int L, float weight,
                              void* stream) {
  if (!dst || !src || !n || L < 1 || L > AD_MAX) return DGV2_EINVAL;
  LerpArgs a;
  int nmax = 0;
  for (int l = 0; l < L; ++l) {
    if (!dst[l] || !src[l] || n[l] < 0) return DGV2_EINVAL;
    a.dst[l] = dst[l]; a.src[l] = src[l]; a.n[l] = n[l];
    nmax = n[l] > nmax ? n[l] : nmax;
  }
  dim3 grid(grid_for(nmax, 256, 256), L);
  lerp_list_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a, weight);
  DGV2_RETURN_LAST();
}

// step: fp32 [1] device counter (advanced by one); sc: fp32 [4] scratch receiving (step, 1 - b1^step, sqrt(1 - b2^step)).
extern "C" int dgv2_adam_prep(float* sc, float* step, float b1, float b2, void* stream) {
  if (!sc || !step) return DGV2_EINVAL;
  adam_prep_kernel<<<1, 1, 0, (hipStream_t)stream>>>(sc, step, b1, b2);
  DGV2_RETURN_LAST();
}

// p / g / m / v: HOST arrays of L <= 72 device pointers (fp32 tensors of n[l] elements each); sc from dgv2_adam_prep.
extern "C" int dgv2_adam_step(float* const* p, const float* const* g, float* const* m, float* const* v, const int* n,
                              int L, const float* sc, float lr, float b1, float b2, float eps, void* stream) {
  if (!p || !g || !m || !v || !n || !sc || L < 1 || L > AD_MAX) return DGV2_EINVAL;
  AdamArgs a;
  int nmax = 0;
  for (int l = 0; l < L; ++l) {
    if (!p[l] || !g[l] || !m[l] || !v[l] || n[l] < 0) return DGV2_EINVAL;
    a.p[l] = p[l]; a.g[l] = g[l]; a.m[l] = m[l]; a.v[l] = v[l]; a.n[l] = n[l];
    nmax = n[l] > nmax ? n[l] : nmax;
  }
  dim3 grid(grid_for(nmax / 4 + 1, 256, 1024), L);
  if (b1 == 0.f) adam_step_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(a, sc, lr, b1, b2, eps);
  else adam_step_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(a, sc, lr, b1, b2, eps);
  DGV2_RETURN_LAST();
}
