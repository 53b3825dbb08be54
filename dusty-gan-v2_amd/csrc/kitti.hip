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

// Ring index per point from the scan order (kitti.py:328-346): a ring starts where the azimuth passes from the 4th into
// the 1st quadrant (quadrant of point i-1 minus quadrant of point i == 3, the sequence taken cyclically); with L
// delimiters and seg = (number of delimiters at or before the point) - 1, the point's ring counted from the end is
// back = (L - 1) - seg and its row (H - 1) - back; points before the first delimiter and rings older than the (H+1)-th
// from the end get 0, the (H+1)-th gets -1 as in the reference.
// Two launches over blocks of 4096 points: (1) delimiters per block, (2) every block adds up the counts of the blocks
// before it (a few dozen values), scans its own lanes and writes the rows.  The quadrants of a block's points go
// through LDS (coalesced loads, then each lane walks 16 consecutive points).
constexpr int KR_BLOCK = 4096, KR_PER = 16;

__device__ __forceinline__ int kitti_stage_and_count(unsigned char* s_q, const float* __restrict__ pts, int n, int base) {
  for (int i = threadIdx.x; i <= KR_BLOCK; i += 256) {   // s_q[i] = quadrant of point base - 1 + i (cyclic)
    int p = base - 1 + i;
    p = p < 0 ? n - 1 : p;
    unsigned char q = 0;
    if (p < n) {
      const float x = pts[4 * p], y = pts[4 * p + 1];
      q = x >= 0.f ? (y >= 0.f ? 0 : 3) : (y >= 0.f ? 1 : 2);
    }
    s_q[i] = q;
  }
  __syncthreads();
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < KR_PER; ++j) {
    const int i = threadIdx.x * KR_PER + j;
    cnt += (base + i < n) && ((int)s_q[i] - (int)s_q[i + 1] == 3);
  }
  return cnt;
}

__global__ __launch_bounds__(256) void kitti_rows_count_kernel(int* __restrict__ counts, const float* __restrict__ pts, int n) {
  __shared__ unsigned char s_q[KR_BLOCK + 1];
  __shared__ int red[4];
  int cnt = kitti_stage_and_count(s_q, pts, n, blockIdx.x * KR_BLOCK);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void kitti_rows_write_kernel(int* __restrict__ row, const int* __restrict__ counts,
                                                               const float* __restrict__ pts, int n, int H) {
  __shared__ unsigned char s_q[KR_BLOCK + 1];
  __shared__ int wsum[4];
  const int base = blockIdx.x * KR_BLOCK, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cnt = kitti_stage_and_count(s_q, pts, n, base);
  int before = 0, L = 0;
  for (int k = 0; k < (int)gridDim.x; ++k) {
    const int c = counts[k];
    before += k < (int)blockIdx.x ? c : 0;
    L += c;
  }
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int seen = before + inc - cnt;
  for (int w = 0; w < wave; ++w) seen += wsum[w];
#pragma unroll
  for (int j = 0; j < KR_PER; ++j) {
    const int i = threadIdx.x * KR_PER + j;
    if (base + i >= n) break;
    seen += ((int)s_q[i] - (int)s_q[i + 1] == 3);
    const int seg = seen - 1, back = (L - 1) - seg;
    row[base + i] = (seg >= 0 && back <= H) ? (H - 1) - back : 0;
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

extern "C" int dgv2_kitti_rows(int* row, int* counts, const float* pts, int n, int H, void* stream) {
  if (!row || !counts || !pts || n < 0 || H <= 0) return DGV2_EINVAL;
  if (n == 0) return 0;
  const int blocks = (n + KR_BLOCK - 1) / KR_BLOCK;
  kitti_rows_count_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(counts, pts, n);
  kitti_rows_write_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(row, counts, pts, n, H);
  DGV2_RETURN_LAST();
}
