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

// ------------------------------------------------------------------------------------------------
// Backward of bias + leaky-ReLU in ONE pass over the activation gradient (channels-last [rows, C]):
//   gx = (ref > 0 ? gy : gy * alpha) * scale          (ref = forward output)
//   gb[c] = sum_rows gx[:, c]
// reference: FusedLeakyReLUFunctionBackward.forward, fused_act.py:22-45 = the act kernel (grad = 1)
// followed by a separate grad_input.sum(dim) reduction that re-reads gx.
// Thread t owns channel vector t % cvecs and row lane t / cvecs; partial sums stay in registers over the
// grid-stride loop, meet in LDS, and leave as one fp32 atomic per channel per block.
// ------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(T* __restrict__ gx, float* __restrict__ gb,
                                                           const T* __restrict__ gy, const T* __restrict__ ref,
                                                           int64_t rows, int cvecs, float alpha, float scale,
                                                           float* __restrict__ partial,
                                                           const float* __restrict__ row_scale) {
  constexpr int VN = vec16<T>::N;
  __shared__ float red[256 * VN];
  const int tid = threadIdx.x;
  const int lanes = 256 / cvecs;  // row lanes per block (cvecs divides 256)
  const int cv = tid % cvecs, rl = tid / cvecs;
  float acc[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) acc[j] = 0.f;
  const int64_t C = (int64_t)cvecs * VN;
  // optional per-channel factor on the STORED gradient only (gb sums the unscaled one): the modulated convs keep
  // their input-magnitude factor c[o] out of the weights, y = act(c[o] * acc + b[o]), so d/dacc carries c[o]
  float rs[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) rs[j] = row_scale ? row_scale[cv * VN + j] : 1.f;
  constexpr int U = 4;  // rows in flight per thread (few blocks, so the loop itself must cover the latency)
  const int64_t stride = (int64_t)gridDim.x * lanes;
  for (int64_t r0 = (int64_t)blockIdx.x * lanes + rl; r0 < rows; r0 += stride * U) {
    vec16<T> g[U], f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * stride;
      if (r < rows) {
        g[u].load(gy + r * C + cv * VN);
        f[u].load(ref + r * C + cv * VN);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * stride;
      if (r < rows) {
        vec16<T> o;
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          const float v = (f[u].get(j) > 0.f ? g[u].get(j) : g[u].get(j) * alpha) * scale;
          o.set(j, v);
          acc[j] += o.get(j);  // sum what is actually stored (the reference sums the rounded gx)
          if (row_scale) o.set(j, v * rs[j]);
        }
        o.store(gx + r * C + cv * VN);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VN; ++j) red[tid * VN + j] = acc[j];
  __syncthreads();
  // one thread per CHANNEL: a wave-instruction of atomics then covers 64 consecutive channels, and the
  // grid is small -- same-address float atomics serialise at the memory side (~25 ns per instruction)
  for (int c = tid; c < cvecs * VN; c += 256) {
    const int v = c / VN, j = c - v * VN;
    float s = 0.f;
    for (int k = 0; k < lanes; ++k) s += red[(k * cvecs + v) * VN + j];
    if (partial) partial[(int64_t)blockIdx.x * cvecs * VN + c] = s;   // many-block mode: summed by the reduce kernel
    else atomicAdd(&gb[c], s);
  }
}

// gb[c] = sum_blk partial[blk][c]: one block per channel, 256 threads split the producer blocks
__global__ __launch_bounds__(256) void bias_partial_reduce_kernel(float* __restrict__ gb, const float* __restrict__ partial,
                                                                  int nblk, int C) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s = 0.f;
#pragma unroll 4
  for (int k = threadIdx.x; k < nblk; k += 256) s += partial[(int64_t)k * C + c];
  s = block_sum(s, red);
  if (threadIdx.x == 0) gb[c] = s;
}

}  // namespace

// gx [rows,C] and gb fp32 [C] from gy [rows,C] and the forward output ref [rows,C] (channels-last).
// Returns DGV2_EINVAL when the shape does not fit the fused kernel (caller falls back to
// dgv2_fused_bias_act + dgv2_bias_grad): C must be a multiple of the 16-byte vector with C/vec dividing 256.
// scratch (optional, fp32 [scratch_elems >= 2048 * C]): with it the pass runs on up to 2048 blocks and the
// per-block column sums are folded by a second kernel (no atomics, no zero fill of gb); without it <= 256 blocks
// add their sums with fp32 atomics (same-address atomics would serialise a larger grid).
extern "C" int dgv2_bias_act_bwd(void* gx, float* gb, const void* gy, const void* ref, int64_t rows, int C,
                                 float alpha, float scale, float* scratch, int64_t scratch_elems, int dtype,
                                 void* stream) {
  return dgv2_bias_act_bwd_rs(gx, gb, gy, ref, rows, C, alpha, scale, nullptr, scratch, scratch_elems, dtype, stream);
}

extern "C" int dgv2_bias_act_bwd_rs(void* gx, float* gb, const void* gy, const void* ref, int64_t rows, int C,
                                    float alpha, float scale, const float* row_scale, float* scratch,
                                    int64_t scratch_elems, int dtype, void* stream) {
  if (!gx || !gb || !gy || !ref || rows <= 0 || C <= 0) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int vn = dtype == DGV2_BF16 ? 8 : 4;
  if (C % vn || 256 % (C / vn) || !aligned16(gx) || !aligned16(gy) || !aligned16(ref)) return DGV2_EINVAL;
  const int cvecs = C / vn, lanes = 256 / cvecs;
  const bool many = scratch && scratch_elems >= (int64_t)2048 * C && aligned16(scratch);
  if (!many) {
    hipError_t e = hipMemsetAsync(gb, 0, sizeof(float) * C, st);
    if (e != hipSuccess) return (int)e;
  }
  // 4 rows in flight per thread: >= 8 rows per row-lane before adding blocks (32 in the atomics mode)
  int64_t want = rows / ((int64_t)lanes * (many ? 8 : 32));
  const int cap = many ? 2048 : 256;
  const int grid = (int)(want < 32 ? 32 : (want > cap ? cap : want));
  DGV2_DISPATCH_DTYPE(dtype, {
    bias_act_bwd_kernel<T><<<grid, 256, 0, st>>>((T*)gx, gb, (const T*)gy, (const T*)ref, rows, cvecs, alpha, scale,
                                                 many ? scratch : nullptr, row_scale);
  });
  if (many) bias_partial_reduce_kernel<<<C, 256, 0, st>>>(gb, scratch, grid, C);
  DGV2_RETURN_LAST();
}

// y[i] = (TY)(x[i] * row_scale[i % C]): the activation-free modulated layers (the output heads) hand their output
// gradient on to the GEMMs with the per-channel input-magnitude factor applied and in the compute dtype.
namespace {
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void scale_cast_kernel(TY* __restrict__ y, const TX* __restrict__ x,
                                                         const float* __restrict__ row_scale, int64_t n, int C) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = from_f32<TY>(to_f32(x[i]) * row_scale[i % C]);
}
}  // namespace

extern "C" int dgv2_scale_cast(void* y, const void* x, const float* row_scale, int64_t n, int C, int xdtype, int ydtype,
                               void* stream) {
  if (!y || !x || !row_scale || n <= 0 || C <= 0) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(n, 256, 4096);
  if (xdtype == DGV2_F32 && ydtype == DGV2_F32)
    scale_cast_kernel<float, float><<<grid, 256, 0, st>>>((float*)y, (const float*)x, row_scale, n, C);
  else if (xdtype == DGV2_F32 && ydtype == DGV2_BF16)
    scale_cast_kernel<float, bf16_t><<<grid, 256, 0, st>>>((bf16_t*)y, (const float*)x, row_scale, n, C);
  else if (xdtype == DGV2_BF16 && ydtype == DGV2_BF16)
    scale_cast_kernel<bf16_t, bf16_t><<<grid, 256, 0, st>>>((bf16_t*)y, (const bf16_t*)x, row_scale, n, C);
  else
    return DGV2_EINVAL;
  DGV2_RETURN_LAST();
}
