// Surface normals of a coordinated point map (range image of xyz points).
// Reference: estimate_surface_normal, gans/geometry.py:38-127 -- replicate padding along H, circular along W,
// the 8 neighbours at distance d in the reference's order, neighbour pairs (k, k+2):
//   mode 0 "closest": the pair with the smallest |p1-a| + |p2-a| (first minimum) gives n = (p1-a) x (p2-a)
//   mode 1 "mean":    n = mean_k (p1_k-a) x (p2_k-a)
//   out = n / (|n| + 1e-8)
// The reference gathers three [B,8,H,W,3] tensors with advanced indexing (~30 launches, 25x the input in
// intermediates); here one thread owns a pixel and reads its 9 points (NCHW planes, coalesced along W).
#include "common.h"

namespace {

__device__ __forceinline__ float norm3(float x, float y, float z) {
  // (x^2 + y^2) + z^2 without contraction: the order of the reference's reduction over the last axis
  return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
}

__global__ __launch_bounds__(256) void surface_normal_kernel(float* __restrict__ out, const float* __restrict__ pts,
                                                             int B, int H, int W, int d, int mode) {
  const int64_t total = (int64_t)B * H * W;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int w = (int)(i % W);
  const int h = (int)((i / W) % H);
  const int b = (int)(i / ((int64_t)W * H));
  const int64_t plane = (int64_t)H * W;
  const float* pb = pts + (int64_t)b * 3 * plane;
  const float ax = pb[(int64_t)h * W + w], ay = pb[plane + (int64_t)h * W + w], az = pb[2 * plane + (int64_t)h * W + w];
  const int dh[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, dw[8] = {0, 1, 1, 1, 0, -1, -1, -1};
  float vx[8], vy[8], vz[8], nrm[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int hh = min(max(h + dh[k] * d, 0), H - 1);
    int ww = (w + dw[k] * d) % W;
    ww = ww < 0 ? ww + W : ww;
    const int64_t o = (int64_t)hh * W + ww;
    vx[k] = __fsub_rn(pb[o], ax);
    vy[k] = __fsub_rn(pb[plane + o], ay);
    vz[k] = __fsub_rn(pb[2 * plane + o], az);
    nrm[k] = norm3(vx[k], vy[k], vz[k]);
  }
  float nx = 0.f, ny = 0.f, nz = 0.f;
  if (mode == 0) {
    int best = 0;
    float bd = __fadd_rn(nrm[0], nrm[2]);
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float dk = __fadd_rn(nrm[k], nrm[(k + 2) & 7]);
      if (dk < bd) { bd = dk; best = k; }   // strict: the first minimum wins, as torch.argmin on CPU
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k == best) {
        const int k2 = (k + 2) & 7;
        nx = __fsub_rn(__fmul_rn(vy[k], vz[k2]), __fmul_rn(vz[k], vy[k2]));
        ny = __fsub_rn(__fmul_rn(vz[k], vx[k2]), __fmul_rn(vx[k], vz[k2]));
        nz = __fsub_rn(__fmul_rn(vx[k], vy[k2]), __fmul_rn(vy[k], vx[k2]));
      }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int k2 = (k + 2) & 7;
      nx = __fadd_rn(nx, __fsub_rn(__fmul_rn(vy[k], vz[k2]), __fmul_rn(vz[k], vy[k2])));
      ny = __fadd_rn(ny, __fsub_rn(__fmul_rn(vz[k], vx[k2]), __fmul_rn(vx[k], vz[k2])));
      nz = __fadd_rn(nz, __fsub_rn(__fmul_rn(vx[k], vy[k2]), __fmul_rn(vy[k], vx[k2])));
    }
    nx = nx / 8.f; ny = ny / 8.f; nz = nz / 8.f;
  }
  const float inv = 1.f / __fadd_rn(norm3(nx, ny, nz), 1e-8f);
  float* ob = out + (int64_t)b * 3 * plane + (int64_t)h * W + w;
  ob[0] = nx * inv;
  ob[plane] = ny * inv;
  ob[2 * plane] = nz * inv;
}

}  // namespace

// points / out fp32 [B, 3, H, W] (NCHW as in the module API); d >= 1 (d < W); mode 0 = "closest", 1 = "mean".
extern "C" int dgv2_surface_normal(float* out, const float* points, int B, int H, int W, int d, int mode, void* stream) {
  if (!out || !points || B <= 0 || H <= 0 || W <= 0 || d < 1 || d >= W || (mode != 0 && mode != 1)) return DGV2_EINVAL;
  const int64_t total = (int64_t)B * H * W;
  if (total >= (1LL << 39)) return DGV2_EINVAL;
  surface_normal_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(out, points, B, H, W, d, mode);
  DGV2_RETURN_LAST();
}
