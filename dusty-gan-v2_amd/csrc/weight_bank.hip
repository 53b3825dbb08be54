// Compute-dtype copies of ALL conv weights of the discriminator in one launch:
//   wf[l] [O, kh*kw, Cpad] = scale_l * w_l[o, c, t]  (forward layout, channels-last filter, zero padded channels)
//   wt[l] [Cpad, kh*kw, O] = the same values transposed (data-gradient layout)
// from the fp32 master parameters [O, C, kh, kw] (reference: ops.Conv2d + EqualLR runtime scaling,
// gans/models/ops/common.py:158-210).  Before, every conv call spent five tiny launches on scale / permute /
// cast / transpose; the parameters stay separate tensors (state-dict layout), so their addresses travel by value
// in the kernel arguments.
#include "common.h"

namespace {

constexpr int WB_MAX = 32;

struct BankArgs {
  const float* src[WB_MAX];
  void* wf[WB_MAX];
  void* wt[WB_MAX];
  int O[WB_MAX], C[WB_MAX], Cpad[WB_MAX], kk[WB_MAX];
  float scale[WB_MAX];
};

template <typename T>
__global__ __launch_bounds__(256) void weight_bank_kernel(BankArgs a) {
  const int l = blockIdx.y;
  const int O = a.O[l], C = a.C[l], Cp = a.Cpad[l], kk = a.kk[l];
  const float s = a.scale[l];
  const float* src = a.src[l];
  T* wf = reinterpret_cast<T*>(a.wf[l]);
  T* wt = reinterpret_cast<T*>(a.wt[l]);
  const int n = O * kk * Cp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int c = i % Cp, r = i / Cp;
    const int t = r % kk, o = r / kk;
    const float v = c < C ? src[(o * C + c) * kk + t] * s : 0.f;
    wf[i] = from_f32<T>(v);
    wt[(c * kk + t) * O + o] = from_f32<T>(v);
  }
}

}  // namespace

// src / wf / wt: HOST arrays of L <= 32 device pointers; O, C, Cpad, kk (= kh*kw), scale: HOST arrays.
extern "C" int dgv2_conv_weight_bank(void* const* wf, void* const* wt, const float* const* src, const int* O,
                                     const int* C, const int* Cpad, const int* kk, const float* scale, int L,
                                     int dtype, void* stream) {
  if (!wf || !wt || !src || !O || !C || !Cpad || !kk || !scale || L < 1 || L > WB_MAX) return DGV2_EINVAL;
  BankArgs a;
  int nmax = 0;
  for (int l = 0; l < L; ++l) {
    if (!wf[l] || !wt[l] || !src[l] || O[l] < 1 || C[l] < 1 || Cpad[l] < C[l] || kk[l] < 1) return DGV2_EINVAL;
    a.src[l] = src[l]; a.wf[l] = wf[l]; a.wt[l] = wt[l];
    a.O[l] = O[l]; a.C[l] = C[l]; a.Cpad[l] = Cpad[l]; a.kk[l] = kk[l]; a.scale[l] = scale[l];
    const int64_t n = (int64_t)O[l] * kk[l] * Cpad[l];
    if (n >= (1 << 30)) return DGV2_EINVAL;
    nmax = n > nmax ? (int)n : nmax;
  }
  dim3 grid(grid_for(nmax, 256, 256), L);
  hipStream_t st = (hipStream_t)stream;
  DGV2_DISPATCH_DTYPE(dtype, { weight_bank_kernel<T><<<grid, 256, 0, st>>>(a); });
  DGV2_RETURN_LAST();
}
