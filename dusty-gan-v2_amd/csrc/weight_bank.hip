// Compute-dtype copies of ALL conv weights of the discriminator in one launch:
//   wf[l] [O, kh*kw, Cpad] = scale_l * w_l[o, c, t]  (forward layout, channels-last filter, zero padded channels)
//   wt[l] [Cpad, kh*kw, O] = the same values transposed (data-gradient layout)
// from the fp32 master parameters [O, C, kh, kw] (reference: ops.Conv2d + EqualLR runtime scaling,
// gans/models/ops/common.py:158-210).  Before, every conv call spent five tiny launches on scale / permute /
// cast / transpose; the parameters stay separate tensors (state-dict layout), so their addresses travel by value
// in the kernel arguments.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int WB_MAX = 32;

struct BankArgs {
  const float* src[WB_MAX];
  void* wf[WB_MAX];
  void* wt[WB_MAX];
  void* w8[WB_MAX];   // optional (3x3, O % 64 == 0, Cpad % 32 == 0, 2-byte T): conv8.hip's staging image, else nullptr
  void* w8t[WB_MAX];  // optional (3x3, Cpad % 64 == 0, O % 32 == 0): the image of the data gradient's operand, else nullptr
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

// Tiled variant (kk <= 9): a block owns a 64 (o) x 32 (c) x kk tile, reads it with contiguous runs of the master
// layout into LDS and writes both layouts with contiguous runs (the transposed layout's scattered 2-byte stores
// were the whole cost of the straightforward kernel above: 71 us per launch on the dusty_v2 discriminator).
constexpr int WB_TO = 16, WB_TC = 32, WB_KK = 9;   // small tiles: ~1000 blocks; the kernel is latency-, not bandwidth-shaped

// KK: compile-time taps per filter (1 or 9; 0 = read it from the arguments): the index arithmetic below divides by it
// six times per element, and with a run-time divisor those divisions WERE the kernel (71 us; 2.5 M elements)
template <typename T, int KK>
__global__ __launch_bounds__(256) void weight_bank_tiled_kernel(BankArgs a) {
  const int l = blockIdx.y;
  const int O = a.O[l], C = a.C[l], Cp = a.Cpad[l];
  const int kk = KK > 0 ? KK : a.kk[l];
  if (KK > 0 && a.kk[l] != KK) return;   // this launch handles the layers with KK taps
  const int tc = (Cp + WB_TC - 1) / WB_TC, to = (O + WB_TO - 1) / WB_TO;
  if ((int)blockIdx.x >= tc * to) return;
  const int o0 = (blockIdx.x / tc) * WB_TO, c0 = (blockIdx.x % tc) * WB_TC;
  const float s = a.scale[l];
  const float* src = a.src[l];
  T* wf = reinterpret_cast<T*>(a.wf[l]);
  T* wt = reinterpret_cast<T*>(a.wt[l]);
  __shared__ float tile[WB_TO][WB_TC * WB_KK + 1];
  const int run = WB_TC * kk;
  // eight loads in flight per thread: with a rolled loop every iteration waited for its own load (72 serial HBM/L2
  // latencies per block were the whole 60 us of this kernel)
  for (int e0 = threadIdx.x; e0 < WB_TO * run; e0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256;
      const int ec = min(e, WB_TO * run - 1);
      const int o = ec / run, r = ec - o * run;
      const int c = c0 + r / kk;
      const bool ok = e < WB_TO * run && o0 + o < O && c < C;
      const size_t idx = ok ? ((size_t)(o0 + o) * C + c0) * kk + r : 0;
      v[u] = ok ? src[idx] * s : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256;
      if (e < WB_TO * run) {
        const int o = e / run, r = e - o * run;
        tile[o][r] = v[u];
      }
    }
  }
  __syncthreads();
  for (int f = threadIdx.x; f < WB_TO * run; f += 256) {   // forward layout: runs of 32 channels
    const int c = f % WB_TC, r = f / WB_TC;
    const int t = r % kk, o = r / kk;
    if (o0 + o < O && c0 + c < Cp) wf[((size_t)(o0 + o) * kk + t) * Cp + c0 + c] = from_f32<T>(tile[o][c * kk + t]);
  }
  if (KK == 9 && sizeof(T) == 2 && a.w8[l]) {
    // conv8.hip's image: [slab = o / 64][chunk = c / 32][unit id] of 16 bytes (8 channels), id = the kernel's staging
    // slot order for (row = tap * 64 + o % 64, plane = (c % 32) / 8): (row >> 3) * 32 + plane * 8 + (row & 7)
    T* w8 = reinterpret_cast<T*>(a.w8[l]);
    const int nchunks = Cp / WB_TC;
    for (int f = threadIdx.x; f < WB_TO * run; f += 256) {
      const int c = f % WB_TC, r = f / WB_TC;
      const int t = r % kk, o = r / kk;
      if (o0 + o < O) {
        const int og = o0 + o, row = t * 64 + (og & 63), plane = c >> 3;
        const size_t unit = ((size_t)(og >> 6) * nchunks + c0 / WB_TC) * (576 * 4) + (row >> 3) * 32 + plane * 8 + (row & 7);
        w8[unit * 8 + (c & 7)] = from_f32<T>(tile[o][c * kk + t]);
      }
    }
  }
  if (KK == 9 && sizeof(T) == 2 && a.w8t[l]) {
    // the data gradient contracts over o and produces c: slab = c / 64, chunk = o / 32, row = t' * 64 + c % 64 with the
    // kernel's tap order t' = (dy + 1) * 3 + dx + 1, dy = 1 - ky: t' = 8 - (ky * 3 + kx); plane = (o % 32) / 8
    T* w8t = reinterpret_cast<T*>(a.w8t[l]);
    const int nchunks = O / 32;
    for (int g = threadIdx.x; g < WB_TO * run; g += 256) {
      const int o = g % WB_TO, r = g / WB_TO;
      const int t = r % kk, c = r / kk;
      if (o0 + o < O && c0 + c < Cp) {
        const int og = o0 + o, cg = c0 + c, row = (8 - t) * 64 + (cg & 63), plane = (og & 31) >> 3;
        const size_t unit = ((size_t)(cg >> 6) * nchunks + (og >> 5)) * (576 * 4) + (row >> 3) * 32 + plane * 8 + (row & 7);
        w8t[unit * 8 + (og & 7)] = from_f32<T>(tile[o][c * kk + t]);
      }
    }
  }
  if constexpr (KK == 9 && sizeof(T) == 4) {
    // fp32 layers: the images are conv_x3.hip's -- the value split into three bf16 planes (h = bf16(v), m = bf16(v - h),
    // l = bf16(v - h - m)), one image per plane in the layout above ([3][slab][chunk][2304 units of 8 bf16]); the forward
    // image has ceil(Cpad / 32) whole chunks (zeros past C), the transposed one the floor(Cpad / 64) whole slabs
    auto split = [](float v, bf16_t& h, bf16_t& m, bf16_t& l) {
      h = (bf16_t)v;
      const float r1 = v - (float)h;
      m = (bf16_t)r1;
      l = (bf16_t)(r1 - (float)m);
    };
    if (a.w8[l]) {
      bf16_t* w8 = reinterpret_cast<bf16_t*>(a.w8[l]);
      const int nchunks = tc;
      const size_t plane = (size_t)(O >> 6) * nchunks * (576 * 4) * 8;
      for (int f = threadIdx.x; f < WB_TO * run; f += 256) {
        const int c = f % WB_TC, r = f / WB_TC;
        const int t = r % kk, o = r / kk;
        if (o0 + o < O) {
          const int og = o0 + o, row = t * 64 + (og & 63), pl = c >> 3;
          const size_t unit = ((size_t)(og >> 6) * nchunks + c0 / WB_TC) * (576 * 4) + (row >> 3) * 32 + pl * 8 + (row & 7);
          bf16_t h, m, lo;
          split(tile[o][c * kk + t], h, m, lo);
          w8[unit * 8 + (c & 7)] = h;
          w8[plane + unit * 8 + (c & 7)] = m;
          w8[2 * plane + unit * 8 + (c & 7)] = lo;
        }
      }
    }
    if (a.w8t[l]) {
      bf16_t* w8t = reinterpret_cast<bf16_t*>(a.w8t[l]);
      const int nchunks = O / 32, nslab = Cp >> 6;
      const size_t plane = (size_t)nslab * nchunks * (576 * 4) * 8;
      for (int g = threadIdx.x; g < WB_TO * run; g += 256) {
        const int o = g % WB_TO, r = g / WB_TO;
        const int t = r % kk, c = r / kk;
        if (o0 + o < O && c0 + c < nslab * 64) {
          const int og = o0 + o, cg = c0 + c, row = (8 - t) * 64 + (cg & 63), pl = (og & 31) >> 3;
          const size_t unit = ((size_t)(cg >> 6) * nchunks + (og >> 5)) * (576 * 4) + (row >> 3) * 32 + pl * 8 + (row & 7);
          bf16_t h, m, lo;
          split(tile[o][c * kk + t], h, m, lo);
          w8t[unit * 8 + (og & 7)] = h;
          w8t[plane + unit * 8 + (og & 7)] = m;
          w8t[2 * plane + unit * 8 + (og & 7)] = lo;
        }
      }
    }
  }
  for (int g = threadIdx.x; g < WB_TO * run; g += 256) {   // transposed layout: runs of 64 output channels
    const int o = g % WB_TO, r = g / WB_TO;
    const int t = r % kk, c = r / kk;
    if (o0 + o < O && c0 + c < Cp) wt[((size_t)(c0 + c) * kk + t) * O + o0 + o] = from_f32<T>(tile[o][c * kk + t]);
  }
}

}  // namespace

// src / wf / wt: HOST arrays of L <= 32 device pointers; O, C, Cpad, kk (= kh*kw), scale: HOST arrays.
extern "C" int dgv2_conv_weight_bank_ex(void* const* wf, void* const* wt, void* const* w8, void* const* w8t,
                                        const float* const* src, const int* O, const int* C, const int* Cpad, const int* kk,
                                        const float* scale, int L, int dtype, void* stream);

extern "C" int dgv2_conv_weight_bank(void* const* wf, void* const* wt, const float* const* src, const int* O,
                                     const int* C, const int* Cpad, const int* kk, const float* scale, int L,
                                     int dtype, void* stream) {
  return dgv2_conv_weight_bank_ex(wf, wt, nullptr, nullptr, src, O, C, Cpad, kk, scale, L, dtype, stream);
}

// ... with two more, optional outputs per layer: w8[l] != nullptr asks for the staging image of the eight-wave forward
// conv (conv8.hip, dgv2_conv3x3_fwd8): [O / 64][Cpad / 32][2304 units of 16 bytes in the kernel's slot order], the same
// values as wf (needs kk == 9, O % 64 == 0, Cpad % 32 == 0, bf16); w8t[l] != nullptr for the image of its stride-1 data
// gradient (dgv2_conv3x3_dgrad8): [Cpad / 64][O / 32][2304 units], taps in the gradient's order (needs kk == 9,
// Cpad % 64 == 0, O % 32 == 0, bf16).  dtype fp32: the images are conv_x3.hip's -- each value split into three bf16 planes
// (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m)), one image per plane: w8 [3][O / 64][ceil(Cpad / 32)][2304 units of
// 8 bf16] (kk == 9, O % 64 == 0, Cpad % 8 == 0), w8t [3][floor(Cpad / 64)][O / 32][2304 units] (the channels of whole
// 64-channel slabs; kk == 9, Cpad >= 64, O % 32 == 0).  DGV2_EINVAL otherwise; the arrays themselves may be nullptr.
extern "C" int dgv2_conv_weight_bank_ex(void* const* wf, void* const* wt, void* const* w8, void* const* w8t,
                                        const float* const* src, const int* O, const int* C, const int* Cpad, const int* kk,
                                        const float* scale, int L, int dtype, void* stream) {
  if (!wf || !wt || !src || !O || !C || !Cpad || !kk || !scale || L < 1 || L > WB_MAX) return DGV2_EINVAL;
  BankArgs a;
  for (int l = 0; l < L; ++l) {
    a.w8[l] = w8 ? w8[l] : nullptr;
    a.w8t[l] = w8t ? w8t[l] : nullptr;
    if (dtype == DGV2_F32) {   // conv_x3.hip's three-plane images
      if (a.w8[l] && (kk[l] != 9 || O[l] % 64 || Cpad[l] % 8)) return DGV2_EINVAL;
      if (a.w8t[l] && (kk[l] != 9 || Cpad[l] < 64 || O[l] % 32)) return DGV2_EINVAL;
      continue;
    }
    if (a.w8[l] && (kk[l] != 9 || O[l] % 64 || Cpad[l] % 32 || dtype != DGV2_BF16)) return DGV2_EINVAL;
    if (a.w8t[l] && (kk[l] != 9 || Cpad[l] % 64 || O[l] % 32 || dtype != DGV2_BF16)) return DGV2_EINVAL;
  }
  int nmax = 0, tmax = 0, kmax = 0;
  for (int l = 0; l < L; ++l) {
    if (!wf[l] || !wt[l] || !src[l] || O[l] < 1 || C[l] < 1 || Cpad[l] < C[l] || kk[l] < 1) return DGV2_EINVAL;
    a.src[l] = src[l]; a.wf[l] = wf[l]; a.wt[l] = wt[l];
    a.O[l] = O[l]; a.C[l] = C[l]; a.Cpad[l] = Cpad[l]; a.kk[l] = kk[l]; a.scale[l] = scale[l];
    const int64_t n = (int64_t)O[l] * kk[l] * Cpad[l];
    if (n >= (1 << 30)) return DGV2_EINVAL;
    nmax = n > nmax ? (int)n : nmax;
    const int tiles = ((Cpad[l] + WB_TC - 1) / WB_TC) * ((O[l] + WB_TO - 1) / WB_TO);
    tmax = tiles > tmax ? tiles : tmax;
    kmax = kk[l] > kmax ? kk[l] : kmax;
  }
  hipStream_t st = (hipStream_t)stream;
  static const bool no_tiled = getenv("DGV2_NO_WB_TILED") != nullptr;
  if (kmax <= WB_KK && !no_tiled) {
    dim3 tgrid(tmax, L);
    bool has1 = false, has9 = false, other = false;
    for (int l = 0; l < L; ++l) {
      if (kk[l] == 1) has1 = true;
      else if (kk[l] == 9) has9 = true;
      else other = true;
    }
    if (other)
      for (int l = 0; l < L; ++l)
        if (a.w8[l] || a.w8t[l]) return DGV2_ENOTSUP;   // the images come from the 9-tap instance only
    DGV2_DISPATCH_DTYPE(dtype, {
      if (other) {
        weight_bank_tiled_kernel<T, 0><<<tgrid, 256, 0, st>>>(a);
      } else {
        if (has9) weight_bank_tiled_kernel<T, 9><<<tgrid, 256, 0, st>>>(a);
        if (has1) weight_bank_tiled_kernel<T, 1><<<tgrid, 256, 0, st>>>(a);
      }
    });
    DGV2_RETURN_LAST();
  }
  for (int l = 0; l < L; ++l)
    if (a.w8[l] || a.w8t[l]) return DGV2_ENOTSUP;   // the images are written by the tiled kernel only
  dim3 grid(grid_for(nmax, 256, 256), L);
  DGV2_DISPATCH_DTYPE(dtype, { weight_bank_kernel<T><<<grid, 256, 0, st>>>(a); });
  DGV2_RETURN_LAST();
}
