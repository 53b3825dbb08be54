// Fused bias + activation and the per-channel bias-gradient reduction.
// HBM-bound elementwise work: 16-byte loads/stores per lane, grid-stride loops.
// Reference semantics: gans/models/ops/fused_act/fused_bias_act_kernel.cu:19-65.
#include "common.h"

namespace {

__device__ __forceinline__ float act_apply(float x, float ref, int act, int grad, float alpha) {
  if (act == 3) {
    if (grad == 0) return x > 0.f ? x : x * alpha;
    if (grad == 1) return ref > 0.f ? x : x * alpha;
    return 0.f;
  }
  return grad == 2 ? 0.f : x;  // act == 1 (linear)
}

// Vector kernel: size_x % VN == 0, all pointers 16-byte aligned; when step_b == 1 also
// size_b % VN == 0 so one vector never wraps around the channel axis.
template <typename T>
__global__ void bias_act_vec_kernel(T* __restrict__ y, const T* __restrict__ x, const T* __restrict__ bias,
                                    const T* __restrict__ ref, int64_t nvec, int64_t step_b, int64_t size_b,
                                    int act, int grad, float alpha, float scale) {
  constexpr int VN = vec16<T>::N;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = v * VN;
    vec16<T> xv, rv, ov;
    xv.load(x + i0);
    if (ref) rv.load(ref + i0);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      float xf = xv.get(j);
      if (bias) xf += to_f32(bias[((i0 + j) / step_b) % size_b]);
      const float rf = ref ? rv.get(j) : 0.f;
      ov.set(j, act_apply(xf, rf, act, grad, alpha) * scale);
    }
    ov.store(y + i0);
  }
}

template <typename T>
__global__ void bias_act_scalar_kernel(T* __restrict__ y, const T* __restrict__ x, const T* __restrict__ bias,
                                       const T* __restrict__ ref, int64_t n, int64_t step_b, int64_t size_b,
                                       int act, int grad, float alpha, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float xf = to_f32(x[i]);
    if (bias) xf += to_f32(bias[(i / step_b) % size_b]);
    const float rf = ref ? to_f32(ref[i]) : 0.f;
    y[i] = from_f32<T>(act_apply(xf, rf, act, grad, alpha) * scale);
  }
}

// Column sums of a [rows, C] matrix (channels-last, step_b == 1).  Thread t owns column
// t % C; the 256/C... row-lanes of a block stride over rows; partials meet in LDS, then one
// fp32 atomic per column per block.
template <typename T>
__global__ void colsum_kernel(float* __restrict__ gb, const T* __restrict__ x, int64_t rows, int C) {
  extern __shared__ float red[];
  const int tid = threadIdx.x;
  const int lanes_per_row = blockDim.x / C > 0 ? blockDim.x / C : 1;  // row-lanes per block (C <= blockDim)
  if (C <= (int)blockDim.x) {
    const int c = tid % C;
    const int rl = tid / C;
    float acc = 0.f;
    if (rl < lanes_per_row) {
      for (int64_t r = (int64_t)blockIdx.x * lanes_per_row + rl; r < rows; r += (int64_t)gridDim.x * lanes_per_row)
        acc += to_f32(x[r * C + c]);
    }
    red[tid] = (rl < lanes_per_row) ? acc : 0.f;
    __syncthreads();
    if (tid < C) {
      float s = 0.f;
      for (int k = 0; k < lanes_per_row; ++k) s += red[k * C + tid];
      atomicAdd(&gb[tid], s);
    }
  } else {
    // wide rows: each thread walks columns tid, tid+blockDim, ...; blocks stride over rows
    for (int c = tid; c < C; c += blockDim.x) {
      float acc = 0.f;
      for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) acc += to_f32(x[r * C + c]);
      atomicAdd(&gb[c], acc);
    }
  }
}

// General layout (step_b > 1, e.g. NCHW): one block per contiguous run of step_b elements.
template <typename T>
__global__ void runsum_kernel(float* __restrict__ gb, const T* __restrict__ x, int64_t step_b, int64_t size_b,
                              int64_t nruns) {
  __shared__ float red[4];
  for (int64_t run = blockIdx.x; run < nruns; run += gridDim.x) {
    const T* p = x + run * step_b;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < step_b; i += blockDim.x) acc += to_f32(p[i]);
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
      atomicAdd(&gb[run % size_b], s);
    }
  }
}

}  // namespace

extern "C" int dgv2_fused_bias_act(void* y, const void* x, const void* bias, const void* ref, int64_t size_x,
                                   int64_t step_b, int64_t size_b, int act, int grad, float alpha, float scale,
                                   int dtype, void* stream) {
  if (size_x == 0) return 0;
  if (!y || !x || size_x < 0 || (act != 1 && act != 3) || grad < 0 || grad > 2) return DGV2_EINVAL;
  if (bias && (step_b <= 0 || size_b <= 0)) return DGV2_EINVAL;
  if (!bias) { step_b = 1; size_b = 1; }
  hipStream_t st = (hipStream_t)stream;
  DGV2_DISPATCH_DTYPE(dtype, {
    constexpr int VN = vec16<T>::N;
    const bool vec_ok = (size_x % VN == 0) && aligned16(y) && aligned16(x) && (!ref || aligned16(ref));
    if (vec_ok) {
      const int64_t nvec = size_x / VN;
      bias_act_vec_kernel<T><<<grid_for(nvec, 256), 256, 0, st>>>((T*)y, (const T*)x, (const T*)bias, (const T*)ref,
                                                                 nvec, step_b, size_b, act, grad, alpha, scale);
    } else {
      bias_act_scalar_kernel<T><<<grid_for(size_x, 256), 256, 0, st>>>((T*)y, (const T*)x, (const T*)bias,
                                                                      (const T*)ref, size_x, step_b, size_b, act,
                                                                      grad, alpha, scale);
    }
  });
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_bias_grad(float* gb, const void* x, int64_t size_x, int64_t step_b, int64_t size_b, int dtype,
                              void* stream) {
  if (!gb || size_b <= 0 || step_b <= 0 || size_x < 0) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(gb, 0, sizeof(float) * size_b, st);
  if (e != hipSuccess) return (int)e;
  if (size_x == 0) return 0;
  if (size_x % step_b != 0) return DGV2_EINVAL;
  DGV2_DISPATCH_DTYPE(dtype, {
    if (step_b == 1) {
      const int64_t rows = size_x / size_b;
      if (rows * size_b != size_x) return DGV2_EINVAL;
      const int C = (int)size_b;
      const int lanes = C <= 256 ? 256 / C : 1;
      int grid = grid_for((rows + lanes - 1) / lanes, 1, 1024);
      colsum_kernel<T><<<grid, 256, 256 * sizeof(float), st>>>(gb, (const T*)x, rows, C);
    } else {
      const int64_t nruns = size_x / step_b;
      runsum_kernel<T><<<grid_for(nruns, 1, 4096), 256, 0, st>>>(gb, (const T*)x, step_b, size_b, nruns);
    }
  });
  DGV2_RETURN_LAST();
}
