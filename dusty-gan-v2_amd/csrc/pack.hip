// Pack / unpack of a LIST of small fp32 matrices into one zero-padded [L, Rmax, Cmax] tensor, one launch for
// the whole list.  Used to run the generator's 19 style affines (EqualLR Linear 512 -> I_l of every ModConv2d,
// gans/models/ops/style.py:30,75) as ONE batched library GEMM instead of 19 small ones per pass, and their
// backward as two: the per-layer parameters stay separate tensors (state-dict layout), so their addresses
// travel by value in the kernel arguments (hipGraph-capturable, no device-side pointer table).
#include "common.h"

namespace {

constexpr int PK_MAX = 48;

struct PackArgs {
  const float* src[PK_MAX];   // pack: inputs (NULL = all-zero block); unpack: unused
  float* dst[PK_MAX];         // unpack: outputs
  int rows[PK_MAX], cols[PK_MAX];
  int L, Rmax, Cmax;
};

__global__ __launch_bounds__(256) void pack2d_kernel(float* __restrict__ packed, PackArgs a) {
  const int l = blockIdx.y;
  const int n = a.Rmax * a.Cmax;
  const float* s = a.src[l];
  const int R = a.rows[l], C = a.cols[l];
  float* d = packed + (int64_t)l * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = i / a.Cmax, c = i - r * a.Cmax;
    d[i] = (s && r < R && c < C) ? s[r * C + c] : 0.f;
  }
}

__global__ __launch_bounds__(256) void unpack2d_kernel(const float* __restrict__ packed, PackArgs a) {
  const int l = blockIdx.y;
  float* d = a.dst[l];
  if (!d) return;
  const int R = a.rows[l], C = a.cols[l];
  const float* s = packed + (int64_t)l * a.Rmax * a.Cmax;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < R * C; i += gridDim.x * 256) {
    const int r = i / C, c = i - r * C;
    d[i] = s[r * a.Cmax + c];
  }
}

bool fill_args(PackArgs& a, const int* rows, const int* cols, int L, int Rmax, int Cmax) {
  if (!rows || !cols || L < 1 || L > PK_MAX || Rmax < 1 || Cmax < 1 || (int64_t)Rmax * Cmax >= (1 << 30)) return false;
  a.L = L; a.Rmax = Rmax; a.Cmax = Cmax;
  for (int l = 0; l < L; ++l) {
    if (rows[l] < 0 || rows[l] > Rmax || cols[l] < 0 || cols[l] > Cmax) return false;
    a.rows[l] = rows[l]; a.cols[l] = cols[l];
  }
  return true;
}

}  // namespace

// packed[l, r, c] = src[l][r * cols[l] + c] for r < rows[l], c < cols[l], else 0.  src: HOST array of L device
// pointers (NULL entries give all-zero blocks); rows / cols: HOST arrays.  L <= 48.
extern "C" int dgv2_pack2d(float* packed, const float* const* src, const int* rows, const int* cols, int L, int Rmax,
                           int Cmax, void* stream) {
  PackArgs a;
  if (!packed || !src || !fill_args(a, rows, cols, L, Rmax, Cmax)) return DGV2_EINVAL;
  for (int l = 0; l < L; ++l) a.src[l] = src[l];
  dim3 grid(grid_for((int64_t)Rmax * Cmax, 256, 64), L);
  pack2d_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(packed, a);
  DGV2_RETURN_LAST();
}

// dst[l][r * cols[l] + c] = packed[l, r, c] (the inverse gather; NULL destinations are skipped).
extern "C" int dgv2_unpack2d(float* const* dst, const float* packed, const int* rows, const int* cols, int L, int Rmax,
                             int Cmax, void* stream) {
  PackArgs a;
  if (!packed || !dst || !fill_args(a, rows, cols, L, Rmax, Cmax)) return DGV2_EINVAL;
  for (int l = 0; l < L; ++l) a.dst[l] = dst[l];
  dim3 grid(grid_for((int64_t)Rmax * Cmax, 256, 64), L);
  unpack2d_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(packed, a);
  DGV2_RETURN_LAST();
}

// ------------------------------------------------------------------------------------------------
// Batched transposes of a LIST of per-sample matrices in one launch: dst[l][b, c, r] = src[l][b, r, c] for
// r < rows[l], c < cols[l] (src rows have leading dimension ld[l] >= cols[l]).  The data gradients of the modulated
// convs contract over the OUTPUT channels of the per-sample weights [B, O, I]: the transposed operand of all layers
// (9 strided copies per generator backward) is produced here up front.  16-bit or 32-bit elements.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int TR_MAX = 32;

struct TrArgs {
  const void* src[TR_MAX];
  void* dst[TR_MAX];
  int rows[TR_MAX], cols[TR_MAX], ld[TR_MAX];
  int blk_end[TR_MAX];   // prefix ends of (b, tile) blocks per matrix list entry
  int B;
};

template <typename E>
__global__ __launch_bounds__(256) void transpose_list_kernel(TrArgs a_by_value) {
  const TrArgs& a = *(const TrArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ E tile[32][33];
  int l = 0;
#pragma unroll
  for (int k = 0; k < TR_MAX; ++k) l += (int)blockIdx.x >= a.blk_end[k] ? 1 : 0;
  const int local = blockIdx.x - (l ? a.blk_end[l - 1] : 0);
  const int R = a.rows[l], C = a.cols[l], ld = a.ld[l];
  const int tr = (R + 31) / 32, tc = (C + 31) / 32;
  const int b = local / (tr * tc), t = local - b * tr * tc;
  const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
  const E* s = reinterpret_cast<const E*>(a.src[l]) + (int64_t)b * R * ld;
  E* d = reinterpret_cast<E*>(a.dst[l]) + (int64_t)b * C * R;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    if (r < R && c < C) tile[ty + i * 8][tx] = s[(int64_t)r * ld + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, r = r0 + tx;
    if (r < R && c < C) d[(int64_t)c * R + r] = tile[tx][ty + i * 8];
  }
}

}  // namespace

// src / dst: HOST arrays of L <= 32 device pointers ([B, rows, ld] -> [B, cols, rows]); elem_size 2 or 4.
extern "C" int dgv2_transpose_list(void* const* dst, const void* const* src, const int* rows, const int* cols,
                                   const int* ld, int L, int B, int elem_size, void* stream) {
  if (!dst || !src || !rows || !cols || !ld || L < 1 || L > TR_MAX || B < 1 || (elem_size != 2 && elem_size != 4))
    return DGV2_EINVAL;
  TrArgs a;
  int64_t n = 0;
  for (int l = 0; l < TR_MAX; ++l) a.blk_end[l] = 0x7fffffff;
  for (int l = 0; l < L; ++l) {
    if (!dst[l] || !src[l] || rows[l] < 1 || cols[l] < 1 || ld[l] < cols[l]) return DGV2_EINVAL;
    a.src[l] = src[l]; a.dst[l] = dst[l]; a.rows[l] = rows[l]; a.cols[l] = cols[l]; a.ld[l] = ld[l];
    n += (int64_t)B * ((rows[l] + 31) / 32) * ((cols[l] + 31) / 32);
    if (n >= (1LL << 31) - 1) return DGV2_EINVAL;
    a.blk_end[l] = (int)n;
  }
  a.B = B;
  hipStream_t st = (hipStream_t)stream;
  if (elem_size == 2) transpose_list_kernel<uint16_t><<<(int)n, 256, 0, st>>>(a);
  else transpose_list_kernel<uint32_t><<<(int)n, 256, 0, st>>>(a);
  DGV2_RETURN_LAST();
}
