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
