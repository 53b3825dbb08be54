// Shared device helpers for libdgv2 (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dgv2.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define DGV2_WAVE 64

template <int DT> struct dtype_of;
template <> struct dtype_of<DGV2_F32> { typedef float type; };
template <> struct dtype_of<DGV2_BF16> { typedef bf16_t type; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte vector of T: 4 floats or 8 bf16.
template <typename T> struct vec16 {
  static constexpr int N = 16 / sizeof(T);
  union {
    uint4 raw;
    T e[16 / sizeof(T)];
  };
  __device__ __forceinline__ void load(const T* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  // streaming store for outputs far larger than L2 + MALL that another kernel reads next
  __device__ __forceinline__ void store_nt(T* p) const {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 v = {raw.x, raw.y, raw.z, raw.w};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  }
  __device__ __forceinline__ float get(int i) const { return to_f32(e[i]); }
  __device__ __forceinline__ void set(int i, float v) { e[i] = from_f32<T>(v); }
};

// Two accumulator fragments of the 16x16 MFMA D layout for the SAME 16 pixels and two consecutive
// 16-channel blocks (lane holds channels 4*lc..4*lc+3 of each block at pixel lr, lc = lane >> 4) leave as ONE
// 16-byte bf16 store per lane: lanes lc and lc^1 swap one packed quad, after which an even-lc lane owns
// channels [4*lc, 4*lc+8) of block A and an odd-lc lane channels [4*(lc-1), 4*(lc-1)+8) of block B.  A store
// instruction then writes whole 64-byte runs per pixel instead of four 8-byte pieces of two half-lines.
// Returns the channel offset (relative to block A's first channel) of the 8 channels in `out`.
__device__ __forceinline__ int pack_pair_bf16(const float (&va)[4], const float (&vb)[4], int lc, uint4& out) {
  union { uint2 u; bf16_t e[4]; } pa, pb;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pa.e[r] = (bf16_t)va[r];
    pb.e[r] = (bf16_t)vb[r];
  }
  // v_permlane16_swap(a, b): the odd 16-lane rows of a trade places with the even rows of b -- exactly this exchange
  // (row = lc), with no LDS round trip (the __shfl_xor form was two ds_bpermute + selects per pair, serialised on the
  // LDS latency: a quarter of the epilogue of the 16-fragment data-gradient tiles)
  const auto lo = __builtin_amdgcn_permlane16_swap(pa.u.x, pb.u.x, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(pa.u.y, pb.u.y, false, false);
  out = make_uint4(lo[0], hi[0], lo[1], hi[1]);
  const bool odd = lc & 1;
  return odd ? 16 + 4 * (lc - 1) : 4 * lc;
}

// ---- fp8 (OCP e4m3fn, the gfx950 encoding; max finite 448) ----
struct fp8_t { uint8_t v; };   // storage tag: 16 elements per 16-byte operand chunk
#define DGV2_FP8_MAX 448.0f
// 8 floats -> 8 e4m3 bytes (round to nearest even, saturating at +-448)
__device__ __forceinline__ uint2 pack_fp8x8(const float (&f)[8]) {
  float c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = fminf(fmaxf(f[i], -DGV2_FP8_MAX), DGV2_FP8_MAX);
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}
// 4 e4m3 bytes of a dword -> 4 floats (exact)
__device__ __forceinline__ void unpack_fp8x4(unsigned w, float (&f)[4]) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  const f32x2_ a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
  f[0] = a[0]; f[1] = a[1]; f[2] = b[0]; f[3] = b[1];
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Floor division / modulo for possibly negative numerators (b > 0).
__host__ __device__ __forceinline__ int floordiv(int a, int b) {
  int q = a / b;
  return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}
__host__ __device__ __forceinline__ int floormod(int a, int b) {
  int r = a % b;
  return r < 0 ? r + b : r;
}

// wave-level sum (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-level sum (blockDim.x a multiple of 64, <= 1024; `red` = 16 floats of LDS); the result is valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// sum of squares of the 8 bf16 values packed in a 16-byte quad
__device__ __forceinline__ float sumsq_bf16x8(uint4 q) {
  const unsigned w[4] = {q.x, q.y, q.z, q.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
    s = fmaf(lo, lo, s);
    s = fmaf(hi, hi, s);
  }
  return s;
}

#define DGV2_RETURN_LAST()                 \
  do {                                     \
    hipError_t e__ = hipGetLastError();    \
    return (int)e__;                       \
  } while (0)

#define DGV2_DISPATCH_DTYPE(dtype, ...)                      \
  do {                                                       \
    if ((dtype) == DGV2_F32) {                               \
      typedef float T;                                       \
      __VA_ARGS__;                                           \
    } else if ((dtype) == DGV2_BF16) {                       \
      typedef bf16_t T;                                      \
      __VA_ARGS__;                                           \
    } else {                                                 \
      return DGV2_EINVAL;                                    \
    }                                                        \
  } while (0)

// Zero-fill as an ordinary kernel node.  Every accumulate-by-atomics buffer in this library is cleared
// with it instead of hipMemsetAsync: the launch is an ordinary kernel node under stream capture (memset
// nodes replayed from hipGraphs were observed to misbehave on ROCm 7.0/7.2 for these small buffers).
static __global__ void dgv2_zero_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t dgv2_zero_async(void* p, size_t bytes, hipStream_t st) {
  const size_t n = bytes / 4;  // all callers clear whole fp32 words
  if (n == 0) return hipSuccess;
  size_t g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  dgv2_zero_kernel<<<(int)g, 256, 0, st>>>(reinterpret_cast<uint32_t*>(p), n);
  return hipGetLastError();
}
#define hipMemsetAsync(p, value, bytes, st) dgv2_zero_async((p), (bytes), (st))

// Outputs of at least this many bytes leave with nontemporal stores (DGV2_NT_MIN_MB, default 64; 0 disables).
static inline bool nt_output(int64_t bytes) {
  static const int64_t min_mb = getenv("DGV2_NT_MIN_MB") ? atoll(getenv("DGV2_NT_MIN_MB")) : 64;
  return min_mb > 0 && bytes >= (min_mb << 20);
}

static inline int grid_for(int64_t work, int block, int cap = 256 * 16) {
  int64_t g = (work + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
