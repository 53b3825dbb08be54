// Point-cloud natives of the evaluation path: furthest point sampling + gather, chamfer nearest
// neighbours and the approximate-matching earth mover's distance.
// reference: gans/sampling/fps/furthest_point_sampling.cu, gans/metrics/distance/cd/chamfer_distance.{cpp,cu},
//            gans/metrics/distance/emd/earth_mover_distance.cu
//
// All of this is fp32 VALU work on xyz triples; nothing here is MFMA-shaped.  The designs below are about where
// the per-cloud state lives:
//   FPS       one workgroup per cloud, the cloud AND its running min-distance array live in registers for the whole
//             selection loop (N <= 16384), or xyz in registers + distances in LDS (N <= 32768); a selection round is
//             one sweep over registers, a 6-step wave reduction of (distance, tie key), one 16-slot LDS exchange that
//             carries the winner's coordinates along (no global read on the critical path) and ONE barrier.
//   chamfer   each lane owns 1 or 4 query points, target points stream through LDS as float4 broadcasts.
//   EMD       one workgroup per cloud pair walks the 9 temperature levels; each of the three sweeps of a level keeps
//             the lane's own point in registers and reads the other set as LDS float4 broadcasts (xyz + weight).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// furthest point sampling
// ------------------------------------------------------------------------------------------------------------------
// The reference's selection (furthest_point_sampling.cu:100-205) is an arg-max of the running min distance with two
// details that the indices depend on: points with |p|^2 <= 1e-3 never take part, and ties go by the shape of its
// block reduction: thread t keeps the first maximum of k = t, t+S, ..., the tree folds slot t+s onto slot t
// (s = S/2 .. 1) keeping the lower slot on equality -- among equal maxima the smallest bit-reversed (k mod S), then
// the smallest k wins, where S = min(2^floor(log2 N), 512) is the block size it launches.  `tie_key` reproduces that
// order for any launch shape.
__device__ __forceinline__ unsigned tie_key(int k, int s_log2) {
  const unsigned low = (unsigned)k & ((1u << s_log2) - 1u);
  const unsigned rev = s_log2 ? __brev(low) >> (32 - s_log2) : 0u;
  return (rev << 16) | (unsigned)(k >> s_log2);
}
__device__ __forceinline__ int key_index(unsigned key, int s_log2) {
  const unsigned rev = key >> 16;
  const unsigned low = s_log2 ? __brev(rev) >> (32 - s_log2) : 0u;
  return (int)(((key & 0xffffu) << s_log2) | low);
}

struct Cand {
  float v;
  unsigned key;
  float x, y, z;
};

__device__ __forceinline__ bool beats(float v, unsigned key, float bv, unsigned bkey) {
  return v > bv || (v == bv && key < bkey);
}

// distances exactly as written in the reference, without fused multiply-adds (the oracle is plain fp32 numpy; HIP's
// __fmul_rn / __fadd_rn are ordinary operators that the default -ffp-contract=fast would still fuse)
__device__ __forceinline__ float sqsum(float dx, float dy, float dz) {
#pragma clang fp contract(off)
  return (dx * dx + dy * dy) + dz * dz;
}
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
  return sqsum(bx - ax, by - ay, bz - az);
}

// Sweep order of a lane's register slots: the lane owns k = tid + p * NT; ties inside a lane must also fall in the
// reference's key order.  With NT >= S the keys ascend with p; with NT = 256 < S = 512 (every n > 512 on the 256-lane
// shapes) k mod S = tid + 256 (p & 1), whose bit reversal is brev(tid) + (p & 1): the even slots precede the odd ones.
template <int NT, int P> __device__ constexpr int visit(int i) {
  return (NT == 256 && P >= 4) ? (i < P / 2 ? 2 * i : 2 * (i - P / 2) + 1) : i;
}

// (value, key) as one orderable 64-bit word: value bits + 1 (distances are >= 0), 0 = nothing eligible; low word ~key
__device__ __forceinline__ unsigned long long pack_cand(float v, unsigned key) {
  return ((unsigned long long)(__float_as_uint(v) + 1u) << 32) | (unsigned)~key;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
  return ((unsigned long long)hi << 32) | lo;
}

template <int NT, int P, bool TEMP_LDS>
__global__ __launch_bounds__(NT) void fps_kernel(int* __restrict__ idxs, const float* __restrict__ xyz, int n, int m,
                                                 int s_log2) {
  constexpr int NW = NT / 64;
  extern __shared__ float temp_lds[];   // TEMP_LDS: [P * NT]
  __shared__ unsigned long long slot_c[2][NW];
  __shared__ float4 slot_p[2][NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* pts = xyz + (size_t)blockIdx.x * n * 3;
  int* out = idxs + (size_t)blockIdx.x * m;

  // running min distance; -1 marks a slot that never takes part (beyond n, or |p|^2 <= 1e-3): min(d, -1) stays -1
  float px[P], py[P], pz[P], tmp[TEMP_LDS ? 1 : P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int k = tid + p * NT;
    const int kc = min(k, n - 1);   // branch-free: out-of-range slots load the last point and are marked dead
    px[p] = pts[3 * kc], py[p] = pts[3 * kc + 1], pz[p] = pts[3 * kc + 2];
    const float mag = sqsum(px[p], py[p], pz[p]);
    const float t0 = (k < n && !((double)mag <= 1e-3)) ? 1e10f : -1.f;
    if (TEMP_LDS) temp_lds[p * NT + tid] = t0;
    else tmp[p] = t0;
  }
  const float x0 = pts[0], y0 = pts[1], z0 = pts[2];
  float cx = x0, cy = y0, cz = z0;   // coordinates of the last selected point
  if (tid == 0) out[0] = 0;

  for (int j = 1; j < m; ++j) {
    float bv = -1.f;
    int bp = -1;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int p = visit<NT, P>(i);
      const float d = sqdist(cx, cy, cz, px[p], py[p], pz[p]);
      float t;
      if (TEMP_LDS) t = temp_lds[p * NT + tid];
      else t = tmp[p];
      const float d2 = fminf(d, t);
      if (TEMP_LDS) temp_lds[p * NT + tid] = d2;
      else tmp[p] = d2;
      const bool take = d2 > bv;
      bv = take ? d2 : bv;
      bp = take ? p : bp;
    }
    // the lane's candidate: coordinates out of the register file, key from its index
    float bx = x0, by = y0, bz = z0;
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (bp == p) bx = px[p], by = py[p], bz = pz[p];
    const unsigned long long mine = bp < 0 ? (unsigned long long)~0u : pack_cand(bv, tie_key(tid + bp * NT, s_log2));
    unsigned long long w = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long ow = shfl_xor_u64(w, o);
      w = ow > w ? ow : w;
    }
    if (mine == w) {   // several lanes only when nothing was eligible in the wave: identical data
      slot_c[j & 1][wave] = w;
      slot_p[j & 1][wave] = make_float4(bx, by, bz, 0.f);
    }
    __syncthreads();
    // every wave reduces the NW published candidates across its lanes
    int ws = lane & (NW - 1);
    w = slot_c[j & 1][ws];
#pragma unroll
    for (int o = NW / 2; o > 0; o >>= 1) {
      const unsigned long long ow = shfl_xor_u64(w, o);
      const int os = __shfl_xor(ws, o, 64);
      if (ow > w || (ow == w && os < ws)) w = ow, ws = os;
    }
    const float4 c = slot_p[j & 1][ws];
    cx = c.x, cy = c.y, cz = c.z;
    if (tid == 0) out[j] = key_index(~(unsigned)w, s_log2);
  }
}

// any N: distances in global memory, points re-read from L2 every round (the reference's layout, 1024 lanes)
__global__ __launch_bounds__(1024) void fps_generic_kernel(int* __restrict__ idxs, float* __restrict__ temp,
                                                           const float* __restrict__ xyz, int n, int m, int s_log2) {
  __shared__ Cand slot[2][16];
  const int tid = threadIdx.x, wave = tid >> 6;
  const float* pts = xyz + (size_t)blockIdx.x * n * 3;
  float* tmp = temp + (size_t)blockIdx.x * n;
  int* out = idxs + (size_t)blockIdx.x * m;
  for (int k = tid; k < n; k += 1024) tmp[k] = 1e10f;
  const float x0 = pts[0], y0 = pts[1], z0 = pts[2];
  float cx = x0, cy = y0, cz = z0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    Cand b = {-1.f, 0u, x0, y0, z0};
    for (int k = tid; k < n; k += 1024) {
      const float x = pts[3 * k], y = pts[3 * k + 1], z = pts[3 * k + 2];
      const float mag = sqsum(x, y, z);
      if ((double)mag <= 1e-3) continue;
      const float d2 = fminf(sqdist(cx, cy, cz, x, y, z), tmp[k]);
      tmp[k] = d2;
      const unsigned key = tie_key(k, s_log2);
      if (beats(d2, key, b.v, b.key)) b = {d2, key, x, y, z};
    }
    float wv = b.v;
    unsigned wk = b.key;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(wv, o, 64);
      const unsigned ok = (unsigned)__shfl_xor((int)wk, o, 64);
      if (beats(ov, ok, wv, wk)) wv = ov, wk = ok;
    }
    Cand* s = slot[j & 1];
    if (b.v == wv && b.key == wk) s[wave] = b;
    __syncthreads();
    Cand w = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
      const Cand c = s[i];
      if (beats(c.v, c.key, w.v, w.key)) w = c;
    }
    cx = w.x, cy = w.y, cz = w.z;
    if (tid == 0) out[j] = key_index(w.key, s_log2);
  }
}

// out[b, c, j] = points[b, c, idx[b, j]]     (furthest_point_sampling.cu:37-63)
__global__ __launch_bounds__(256) void gather_points_kernel(float* __restrict__ out, const float* __restrict__ points,
                                                            const int* __restrict__ idx, int C, int n, int m) {
  const int b = blockIdx.z, c = blockIdx.y;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256)
    out[((size_t)b * C + c) * m + j] = points[((size_t)b * C + c) * n + idx[(size_t)b * m + j]];
}
// grad_points[b, c, idx[b, j]] += grad_out[b, c, j]     (:65-93)
__global__ __launch_bounds__(256) void gather_points_grad_kernel(float* __restrict__ gp, const float* __restrict__ go,
                                                                 const int* __restrict__ idx, int C, int n, int m) {
  const int b = blockIdx.z, c = blockIdx.y;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256)
    atomicAdd(gp + ((size_t)b * C + c) * n + idx[(size_t)b * m + j], go[((size_t)b * C + c) * m + j]);
}

// ------------------------------------------------------------------------------------------------------------------
// chamfer: nearest neighbour of every point of `a` [B, n, 3] in `b` [B, m, 3]; first minimum wins (strict <), the
// squared distance is ((dx*dx + dy*dy) + dz*dz) in fp32 with dx = b - a as in nnsearch, chamfer_distance.cpp:42-66
// ------------------------------------------------------------------------------------------------------------------
template <int Q>
__global__ __launch_bounds__(256) void chamfer_nn_kernel(float* __restrict__ dist, int* __restrict__ idx,
                                                         const float* __restrict__ a, const float* __restrict__ b,
                                                         int n, int m, int b_shared) {
  constexpr int TILE = 1024;
  __shared__ float4 tile[TILE];
  const int bi = blockIdx.y, tid = threadIdx.x;
  const float* pa = a + (size_t)bi * n * 3;
  const float* pb = b + (b_shared ? 0 : (size_t)bi * m * 3);   // b_shared: ONE target set for every cloud
  float ax[Q], ay[Q], az[Q], best[Q];
  int besti[Q];
  const int j0 = blockIdx.x * 256 * Q + tid;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = j0 + q * 256;
    ax[q] = ay[q] = az[q] = 0.f;
    if (j < n) ax[q] = pa[3 * j], ay[q] = pa[3 * j + 1], az[q] = pa[3 * j + 2];
    best[q] = 0.f;
    besti[q] = 0;
  }
  for (int k0 = 0; k0 < m; k0 += TILE) {
    const int cnt = min(TILE, m - k0);
    __syncthreads();
    for (int k = tid; k < cnt; k += 256) tile[k] = make_float4(pb[3 * (k0 + k)], pb[3 * (k0 + k) + 1], pb[3 * (k0 + k) + 2], 0.f);
    __syncthreads();
    if (k0 == 0) {
#pragma unroll
      for (int q = 0; q < Q; ++q) best[q] = sqdist(ax[q], ay[q], az[q], tile[0].x, tile[0].y, tile[0].z);
    }
#pragma unroll 4
    for (int k = 0; k < cnt; ++k) {
      const float4 t = tile[k];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float d = sqdist(ax[q], ay[q], az[q], t.x, t.y, t.z);
        const bool lt = d < best[q];
        best[q] = lt ? d : best[q];
        besti[q] = lt ? k0 + k : besti[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = j0 + q * 256;
    if (j < n) dist[(size_t)bi * n + j] = best[q], idx[(size_t)bi * n + j] = besti[q];
  }
}

// ga[j] += 2 g (a_j - b_idx[j]);  gb[idx[j]] -= the same     (chamfer_distance.cpp:104-124, .cu:144-166)
__global__ __launch_bounds__(256) void chamfer_grad_kernel(float* __restrict__ ga, float* __restrict__ gb,
                                                           const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ gd, const int* __restrict__ idx,
                                                           int n, int m) {
  const int bi = blockIdx.y;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    const size_t ja = (size_t)bi * n + j;
    const size_t jb = (size_t)bi * m + idx[ja];
    const float g = gd[ja] * 2.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t = g * (a[3 * ja + c] - b[3 * jb + c]);
      atomicAdd(ga + 3 * ja + c, t);
      atomicAdd(gb + 3 * jb + c, -t);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// earth mover's distance by approximate matching (earth_mover_distance.cu:3-175): 9 annealing levels
// level = -4^j, j = 7 .. -1; per level, with e_kl = exp(level * |x1_k - x2_l|^2):
//   ratioL_k = remainL_k / (1e-9 + sum_l e_kl remainR_l)
//   s_l = remainR_l sum_k e_kl ratioL_k;  ratioR_l = min(remainR_l / (s_l + 1e-9), 1) remainR_l;  remainR_l = max(0, remainR_l - s_l)
//   w_kl = e_kl ratioL_k ratioR_l;  match[l, k] += w_kl;  remainL_k = max(0, remainL_k - sum_l w_kl)
// ------------------------------------------------------------------------------------------------------------------
constexpr int EMD_NT = 1024;

// acc_k (op) sum over the other set of exp(level d) * w_other, own points k = k0 + tid
template <typename F>
__device__ __forceinline__ void emd_sweep(float4* tile, const float* __restrict__ own, int n_own,
                                          const float* __restrict__ other, const float* __restrict__ w_other,
                                          int n_other, float level, F&& body) {
  const int tid = threadIdx.x;
  for (int k0 = 0; k0 < n_own; k0 += EMD_NT) {
    const int k = k0 + tid;
    float x = 0.f, y = 0.f, z = 0.f;
    if (k < n_own) x = own[3 * k], y = own[3 * k + 1], z = own[3 * k + 2];
    body.begin(k);
    for (int l0 = 0; l0 < n_other; l0 += EMD_NT) {
      const int cnt = min(EMD_NT, n_other - l0);
      __syncthreads();
      if (tid < cnt) tile[tid] = make_float4(other[3 * (l0 + tid)], other[3 * (l0 + tid) + 1], other[3 * (l0 + tid) + 2], w_other[l0 + tid]);
      __syncthreads();
      if (k < n_own) {
#pragma unroll 4
        for (int l = 0; l < cnt; ++l) {
          const float4 t = tile[l];
          const float dx = t.x - x, dy = t.y - y, dz = t.z - z;
          const float e = __expf(level * (dx * dx + dy * dy + dz * dz));
          body.pair(k, l0 + l, e * t.w);
        }
      }
    }
    if (k < n_own) body.end(k);
  }
  __syncthreads();
}

struct SweepL {   // ratioL
  float* ratioL;
  const float* remainL;
  float acc;
  __device__ void begin(int) { acc = 1e-9f; }
  __device__ void pair(int, int, float w) { acc += w; }
  __device__ void end(int k) { ratioL[k] = remainL[k] / acc; }
};
struct SweepR {   // ratioR, remainR
  float* ratioR;
  float* remainR;
  float acc;
  __device__ void begin(int) { acc = 0.f; }
  __device__ void pair(int, int, float w) { acc += w; }
  __device__ void end(int l) {
    const float r = remainR[l];
    const float s = acc * r;
    ratioR[l] = fminf(r / (s + 1e-9f), 1.0f) * r;
    remainR[l] = fmaxf(0.0f, r - s);
  }
};
struct SweepM {   // match, remainL
  float* match;
  float* remainL;
  const float* ratioL;
  int n;
  bool first;
  float acc, rl;
  __device__ void begin(int k) { acc = 0.f; rl = k < n ? ratioL[k] : 0.f; }
  __device__ void pair(int k, int l, float w) {
    w *= rl;
    float* p = match + (size_t)l * n + k;
    *p = first ? w : *p + w;
    acc += w;
  }
  __device__ void end(int k) { remainL[k] = fmaxf(0.0f, remainL[k] - acc); }
};

__global__ __launch_bounds__(EMD_NT) void emd_approxmatch_kernel(float* __restrict__ match, float* __restrict__ temp,
                                                                 const float* __restrict__ xyz1,
                                                                 const float* __restrict__ xyz2, int n, int m) {
  __shared__ float4 tile[EMD_NT];
  const int bi = blockIdx.x, tid = threadIdx.x;
  const float* p1 = xyz1 + (size_t)bi * n * 3;
  const float* p2 = xyz2 + (size_t)bi * m * 3;
  float* mt = match + (size_t)bi * n * m;
  float* remainL = temp + (size_t)bi * (n + m) * 2;
  float* remainR = remainL + n;
  float* ratioL = remainR + m;
  float* ratioR = ratioL + n;
  const float multiL = n >= m ? 1.f : (float)(m / n), multiR = n >= m ? (float)(n / m) : 1.f;   // integer quotients
  for (int k = tid; k < n; k += EMD_NT) remainL[k] = multiL;
  for (int l = tid; l < m; l += EMD_NT) remainR[l] = multiR;
  __syncthreads();
  for (int j = 7; j > -2; --j) {
    const float level = -powf(4.0f, (float)j);
    emd_sweep(tile, p1, n, p2, remainR, m, level, SweepL{ratioL, remainL, 0.f});
    emd_sweep(tile, p2, m, p1, ratioL, n, level, SweepR{ratioR, remainR, 0.f});
    emd_sweep(tile, p1, n, p2, ratioR, m, level, SweepM{mt, remainL, ratioL, n, j == 7, 0.f, 0.f});
  }
}

// cost[b] = sum_{k,l} match[l, k] |x1_k - x2_l|     (:177-226); grid (chunks of l, B), atomics into a zeroed cost
__global__ __launch_bounds__(256) void emd_matchcost_kernel(float* __restrict__ cost, const float* __restrict__ match,
                                                            const float* __restrict__ xyz1,
                                                            const float* __restrict__ xyz2, int n, int m) {
  __shared__ float red[16];
  const int bi = blockIdx.y, tid = threadIdx.x;
  const float* p1 = xyz1 + (size_t)bi * n * 3;
  const float* p2 = xyz2 + (size_t)bi * m * 3;
  const float* mt = match + (size_t)bi * n * m;
  const int l_lo = (int)((int64_t)m * blockIdx.x / gridDim.x), l_hi = (int)((int64_t)m * (blockIdx.x + 1) / gridDim.x);
  float acc = 0.f;
  for (int k = tid; k < n; k += 256) {
    const float x = p1[3 * k], y = p1[3 * k + 1], z = p1[3 * k + 2];
    for (int l = l_lo; l < l_hi; ++l) {
      const float dx = p2[3 * l] - x, dy = p2[3 * l + 1] - y, dz = p2[3 * l + 2] - z;
      acc += mt[(size_t)l * n + k] * sqrtf(dx * dx + dy * dy + dz * dz);
    }
  }
  acc = block_sum(acc, red);
  if (tid == 0) atomicAdd(cost + bi, acc);
}

// grad1[k] = sum_l match[l, k] (x1_k - x2_l) / max(|.|, 1e-10)     (:269-297)
__global__ __launch_bounds__(256) void emd_grad1_kernel(float* __restrict__ grad1, const float* __restrict__ match,
                                                        const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                        int n, int m) {
  __shared__ float4 tile[1024];
  const int bi = blockIdx.y, tid = threadIdx.x;
  const float* p1 = xyz1 + (size_t)bi * n * 3;
  const float* p2 = xyz2 + (size_t)bi * m * 3;
  const float* mt = match + (size_t)bi * n * m;
  const int k = blockIdx.x * 256 + tid;
  float x = 0.f, y = 0.f, z = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
  if (k < n) x = p1[3 * k], y = p1[3 * k + 1], z = p1[3 * k + 2];
  for (int l0 = 0; l0 < m; l0 += 1024) {
    const int cnt = min(1024, m - l0);
    __syncthreads();
    for (int l = tid; l < cnt; l += 256) tile[l] = make_float4(p2[3 * (l0 + l)], p2[3 * (l0 + l) + 1], p2[3 * (l0 + l) + 2], 0.f);
    __syncthreads();
    if (k < n) {
#pragma unroll 4
      for (int l = 0; l < cnt; ++l) {
        const float4 t = tile[l];
        const float dx = x - t.x, dy = y - t.y, dz = z - t.z;
        const float d = mt[(size_t)(l0 + l) * n + k] * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
        gx += dx * d, gy += dy * d, gz += dz * d;
      }
    }
  }
  if (k < n) {
    float* g = grad1 + ((size_t)bi * n + k) * 3;
    g[0] = gx, g[1] = gy, g[2] = gz;
  }
}

// grad2[l] = sum_k match[l, k] (x2_l - x1_k) / max(|.|, 1e-10)     (:232-268); one workgroup per (l, cloud)
__global__ __launch_bounds__(256) void emd_grad2_kernel(float* __restrict__ grad2, const float* __restrict__ match,
                                                        const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                        int n, int m) {
  __shared__ float red[16];
  const int bi = blockIdx.y, tid = threadIdx.x;
  const float* p1 = xyz1 + (size_t)bi * n * 3;
  const float* p2 = xyz2 + (size_t)bi * m * 3;
  const float* mt = match + (size_t)bi * n * m;
  for (int l = blockIdx.x; l < m; l += gridDim.x) {
    const float x = p2[3 * l], y = p2[3 * l + 1], z = p2[3 * l + 2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int k = tid; k < n; k += 256) {
      const float dx = x - p1[3 * k], dy = y - p1[3 * k + 1], dz = z - p1[3 * k + 2];
      const float d = mt[(size_t)l * n + k] * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
      gx += dx * d, gy += dy * d, gz += dz * d;
    }
    gx = block_sum(gx, red);
    __syncthreads();
    gy = block_sum(gy, red);
    __syncthreads();
    gz = block_sum(gz, red);
    __syncthreads();
    if (tid == 0) {
      float* g = grad2 + ((size_t)bi * m + l) * 3;
      g[0] = gx, g[1] = gy, g[2] = gz;
    }
  }
}

int ref_block_log2(int n) {   // log2 of the reference's block size: min(2^floor(log2 n), 512)  (opt_n_threads)
  int s = 0;
  while ((2 << s) <= n && s < 9) ++s;
  return s;
}

template <int NT, int P, bool TL>
int launch_fps(int* idxs, const float* xyz, int B, int n, int m, int s_log2, hipStream_t st) {
  const size_t lds = TL ? (size_t)P * NT * sizeof(float) : 0;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)fps_kernel<NT, P, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)hipErrorUnknown;
  }
  fps_kernel<NT, P, TL><<<B, NT, lds, st>>>(idxs, xyz, n, m, s_log2);
  DGV2_RETURN_LAST();
}

}  // namespace

extern "C" int dgv2_fps_scratch(int64_t* floats, int B, int n) {
  if (!floats || B < 0 || n < 0) return DGV2_EINVAL;
  *floats = n > 32768 ? (int64_t)B * n : 0;
  return 0;
}

extern "C" int dgv2_fps(int* idxs, float* temp, const float* xyz, int B, int n, int m, void* stream) {
  if (!idxs || !xyz || B < 0 || n <= 0 || m < 0 || n >= (1 << 24)) return DGV2_EINVAL;
  if (B == 0 || m == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int s = ref_block_log2(n);
  if (n <= 256) return launch_fps<256, 1, false>(idxs, xyz, B, n, m, s, st);
  if (n <= 512) return launch_fps<256, 2, false>(idxs, xyz, B, n, m, s, st);
  if (n <= 1024) return launch_fps<256, 4, false>(idxs, xyz, B, n, m, s, st);
  if (n <= 2048) return launch_fps<256, 8, false>(idxs, xyz, B, n, m, s, st);
  if (n <= 4096) return launch_fps<256, 16, false>(idxs, xyz, B, n, m, s, st);
  if (n <= 8192) return launch_fps<1024, 8, false>(idxs, xyz, B, n, m, s, st);
  if (n <= 16384) return launch_fps<1024, 16, false>(idxs, xyz, B, n, m, s, st);
  if (n <= 32768) return launch_fps<1024, 32, true>(idxs, xyz, B, n, m, s, st);
  if (!temp) return DGV2_EINVAL;
  fps_generic_kernel<<<B, 1024, 0, st>>>(idxs, temp, xyz, n, m, s);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_gather_points(float* out, const float* points, const int* idx, int B, int C, int n, int m,
                                  void* stream) {
  if (!out || !points || !idx || B < 0 || C <= 0 || n <= 0 || m < 0 || C > 65535 || B > 65535) return DGV2_EINVAL;
  if (B == 0 || m == 0) return 0;
  gather_points_kernel<<<dim3((m + 255) / 256, C, B), 256, 0, (hipStream_t)stream>>>(out, points, idx, C, n, m);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_gather_points_grad(float* grad_points, const float* grad_out, const int* idx, int B, int C, int n,
                                       int m, void* stream) {
  if (!grad_points || !grad_out || !idx || B < 0 || C <= 0 || n <= 0 || m < 0 || C > 65535 || B > 65535) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) return 0;
  if (hipMemsetAsync(grad_points, 0, (size_t)B * C * n * sizeof(float), st) != hipSuccess) return (int)hipErrorUnknown;
  if (m == 0) return 0;
  gather_points_grad_kernel<<<dim3((m + 255) / 256, C, B), 256, 0, st>>>(grad_points, grad_out, idx, C, n, m);
  DGV2_RETURN_LAST();
}

static int chamfer_nn(float* dist, int* idx, const float* a, const float* b, int B, int n, int m, hipStream_t st,
                      int b_shared = 0) {
  if ((int64_t)B * ((n + 1023) / 1024) >= 512)
    chamfer_nn_kernel<4><<<dim3((n + 1023) / 1024, B), 256, 0, st>>>(dist, idx, a, b, n, m, b_shared);
  else chamfer_nn_kernel<1><<<dim3((n + 255) / 256, B), 256, 0, st>>>(dist, idx, a, b, n, m, b_shared);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_nn_search(float* dist, int* idx, const float* xyz, const float* ref, int B, int n, int m,
                              int ref_shared, void* stream) {
  if (!dist || !idx || !xyz || !ref || B < 0 || n <= 0 || m <= 0 || B > 65535) return DGV2_EINVAL;
  if (B == 0) return 0;
  return chamfer_nn(dist, idx, xyz, ref, B, n, m, (hipStream_t)stream, ref_shared != 0);
}

extern "C" int dgv2_chamfer_fwd(float* dist1, int* idx1, float* dist2, int* idx2, const float* xyz1, const float* xyz2,
                                int B, int n, int m, void* stream) {
  if (!dist1 || !idx1 || !dist2 || !idx2 || !xyz1 || !xyz2 || B < 0 || n <= 0 || m <= 0 || B > 65535) return DGV2_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int rc = chamfer_nn(dist1, idx1, xyz1, xyz2, B, n, m, st);
  if (rc != 0) return rc;
  return chamfer_nn(dist2, idx2, xyz2, xyz1, B, m, n, st);
}

extern "C" int dgv2_chamfer_bwd(float* gxyz1, float* gxyz2, const float* xyz1, const float* xyz2, const float* gdist1,
                                const float* gdist2, const int* idx1, const int* idx2, int B, int n, int m,
                                void* stream) {
  if (!gxyz1 || !gxyz2 || !xyz1 || !xyz2 || !gdist1 || !gdist2 || !idx1 || !idx2 || B < 0 || n <= 0 || m <= 0 || B > 65535)
    return DGV2_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(gxyz1, 0, (size_t)B * n * 3 * sizeof(float), st) != hipSuccess) return (int)hipErrorUnknown;
  if (hipMemsetAsync(gxyz2, 0, (size_t)B * m * 3 * sizeof(float), st) != hipSuccess) return (int)hipErrorUnknown;
  chamfer_grad_kernel<<<dim3(min((n + 255) / 256, 256), B), 256, 0, st>>>(gxyz1, gxyz2, xyz1, xyz2, gdist1, idx1, n, m);
  chamfer_grad_kernel<<<dim3(min((m + 255) / 256, 256), B), 256, 0, st>>>(gxyz2, gxyz1, xyz2, xyz1, gdist2, idx2, m, n);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_emd_approxmatch(float* match, float* temp, const float* xyz1, const float* xyz2, int B, int n, int m,
                                    void* stream) {
  if (!match || !temp || !xyz1 || !xyz2 || B < 0 || n <= 0 || m <= 0) return DGV2_EINVAL;
  if (B == 0) return 0;
  emd_approxmatch_kernel<<<B, EMD_NT, 0, (hipStream_t)stream>>>(match, temp, xyz1, xyz2, n, m);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_emd_matchcost(float* cost, const float* match, const float* xyz1, const float* xyz2, int B, int n,
                                  int m, void* stream) {
  if (!cost || !match || !xyz1 || !xyz2 || B < 0 || n <= 0 || m <= 0 || B > 65535) return DGV2_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(cost, 0, (size_t)B * sizeof(float), st) != hipSuccess) return (int)hipErrorUnknown;
  emd_matchcost_kernel<<<dim3(min(m, 32), B), 256, 0, st>>>(cost, match, xyz1, xyz2, n, m);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_emd_matchcost_grad(float* grad1, float* grad2, const float* match, const float* xyz1,
                                       const float* xyz2, int B, int n, int m, void* stream) {
  if (!grad1 || !grad2 || !match || !xyz1 || !xyz2 || B < 0 || n <= 0 || m <= 0 || B > 65535) return DGV2_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  emd_grad1_kernel<<<dim3((n + 255) / 256, B), 256, 0, st>>>(grad1, match, xyz1, xyz2, n, m);
  emd_grad2_kernel<<<dim3(min(m, 1024), B), 256, 0, st>>>(grad2, match, xyz1, xyz2, n, m);
  DGV2_RETURN_LAST();
}
