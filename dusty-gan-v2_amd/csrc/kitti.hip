// Point cloud -> range image with scan unfolding, the front end of the real-data path
// (reference: KITTIRaw.load_pts_as_img + __getitem__, gans/datasets/kitti.py:264-279,317-370): per point
//   depth = |xyz|, mask = min_depth <= depth <= max_depth,
//   row   = ring index from the scan order (h, precomputed by the host side from the quadrant sequence) or from the
//           pitch angle (scan_unfolding = False), column = floor(((-atan2(y, x) / pi + 1) / 2 mod 1) * W),
//   the NEAREST point of a pixel wins (the reference sorts by decreasing depth and scatters sequentially),
//   then nearest-neighbour resize to (H, Wout) and multiplication by the mask.
// Here: one pass of 64-bit atomicMin on (depth bits << 32 | point index) per pixel -- depth >= 0, so its float bits
// order like the value; equal depths go to the lower index (the reference's unstable argsort leaves ties open) -- and
// one pass that decodes the winners straight into the decimated [6, H, Wout] output (a W / Wout nearest resize keeps
// every (W / Wout)-th column, so only those columns are ever decoded).  No sort, no host round trip.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void kitti_scatter_kernel(unsigned long long* __restrict__ key,
                                                            const float* __restrict__ pts, const int* __restrict__ row,
                                                            int n, int H, int W, int use_pitch) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float x = pts[4 * i], y = pts[4 * i + 1], z = pts[4 * i + 2];
    const float depth = sqrtf(x * x + y * y + z * z);
    int h;
    if (use_pitch) {   // kitti.py:347-351
      const float fup = 3.f * 0.017453292519943295f, fdown = -25.f * 0.017453292519943295f;
      const float pitch = asinf(z / depth) + fabsf(fdown);
      float gh = 1.f - pitch / (fup - fdown);
      gh = floorf(gh * H);
      h = (int)fminf(fmaxf(gh, 0.f), (float)(H - 1));
    } else {
      h = row[i];
      if (h < 0) h += H;   // the reference's scatter indexes the array with -1 for the 65th ring from the end
    }
    const float yaw = -atan2f(y, x);
    float gw = (yaw / 3.14159265358979323846f + 1.f) * 0.5f;
    gw = gw - floorf(gw);                                  // python's % 1 on a float in [0, 1]
    int w = (int)floorf(gw * W);
    w = w < 0 ? 0 : (w > W - 1 ? W - 1 : w);
    const unsigned long long k = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned)i;
    atomicMin(key + (size_t)h * W + w, k);
  }
}

// out [6, H, Wout]: x, y, z, reflectance, depth, mask of the winning point of pixel (h, w * step) (times the mask)
__global__ __launch_bounds__(256) void kitti_gather_kernel(float* __restrict__ out,
                                                           const unsigned long long* __restrict__ key,
                                                           const float* __restrict__ pts, int H, int W, int Wout,
                                                           float min_depth, float max_depth, int apply_mask) {
  const int step = W / Wout;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < H * Wout; p += gridDim.x * 256) {
    const int h = p / Wout, wo = p - h * Wout;
    const unsigned long long k = key[(size_t)h * W + wo * step];
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k != ~0ull) {
      const int i = (int)(k & 0xffffffffu);
      const float x = pts[4 * i], y = pts[4 * i + 1], z = pts[4 * i + 2];
      const float depth = sqrtf(x * x + y * y + z * z);
      const float m = (depth >= min_depth && depth <= max_depth) ? 1.f : 0.f;
      const float s = apply_mask ? m : 1.f;   // __getitem__ multiplies by the mask (:269); load_pts_as_img does not
      v[0] = x * s; v[1] = y * s; v[2] = z * s; v[3] = pts[4 * i + 3] * s; v[4] = depth * s; v[5] = m;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) out[(size_t)c * H * Wout + p] = v[c];
  }
}

__global__ void kitti_fill_kernel(unsigned long long* __restrict__ key, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) key[i] = ~0ull;
}

}  // namespace

// pts fp32 [n, 4] (x, y, z, reflectance); row int32 [n] ring index per point (scan unfolding; may be -1 = ring H-1 as in
// the reference) or NULL for the pitch-angle rows; key scratch uint64 [H * W]; out fp32 [6, H, Wout], W % Wout == 0.
extern "C" int dgv2_kitti_project(float* out, unsigned long long* key, const float* pts, const int* row, int n, int H,
                                  int W, int Wout, float min_depth, float max_depth, int apply_mask, void* stream) {
  if (!out || !key || !pts || n < 0 || H <= 0 || W <= 0 || Wout <= 0 || W % Wout) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  kitti_fill_kernel<<<grid_for((int64_t)H * W, 256, 1024), 256, 0, st>>>(key, (size_t)H * W);
  if (n > 0) kitti_scatter_kernel<<<grid_for(n, 256, 2048), 256, 0, st>>>(key, pts, row, n, H, W, row == nullptr);
  kitti_gather_kernel<<<grid_for((int64_t)H * Wout, 256, 1024), 256, 0, st>>>(out, key, pts, H, W, Wout, min_depth, max_depth, apply_mask);
  DGV2_RETURN_LAST();
}
