// Direct (LDS halo-tile) convolution on the MFMA cores, channels-last, with a generic TAP LIST.
//
// One kernel covers every dense conv of the discriminator and their data gradients
// (reference: ops.Conv2d in gans/models/ops/common.py:187-210 at gans/models/dusty_v2.py:325-385,
// and the cuDNN dgrad autograd would call):
//   forward 3x3 / 1x1, stride 1 or 2      : taps (ky-1, kx-1), input coord = out * stride + d
//   data gradient, stride 1                : taps (1-ky, 1-kx) on gy with transposed weights
//   data gradient, stride 2                : four parity classes, each a 1/2/2/4-tap conv on gy whose
//                                            outputs are scattered with stride 2 (no zero-tap work)
//   replicate-row border terms of dgrad    : 1-row launches in accumulate mode
// Ring padding = wrap of the W coordinate while the halo tile is staged; H is clamped (forward,
// replicate padding) or zero-filled (gradients).  Nothing padded is ever materialised.
//
// Block = 4 waves; tile = 4 x 32 output positions x TO output channels; wave w owns tile row w.
// Per 64-byte channel chunk the input halo tile and the weight slab [TO][ntaps] are staged in LDS once
// and reused by all taps (9x fewer global loads than an im2col gather); fragments are 16-byte
// ds_read_b128 with the same (idx>>2)&3 XOR swizzle as gemm_core.h.
#include "gemm_core.h"

namespace {

constexpr int DTH = 4;
constexpr int DTW = 32;

struct DConv {
  int B, Hin, Win, Cin;
  int Hg, Wg, O, Hy, Wy;
  int in_stride, ioff_h, ioff_w;
  int out_stride, ooff_h, ooff_w;
  int ntaps, wtaps;
  int dy[9], dx[9], widx[9];
  int dymin, dxmin, rows, cols;
  int hzero, ring, accumulate;
  int ablate;  // benchmarking only (DGV2_ABLATE): 1 skip MFMA loop, 2 skip stores, 4 skip staging
  const float* bias;
  int act;
  float alpha, scale;
};

template <typename T, int TO>
__global__ __launch_bounds__(256) void conv_direct_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                          const T* __restrict__ w, DConv p) {
  constexpr int CE = 16 / sizeof(T);
  constexpr int MF = TO / 16;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int npix = p.rows * p.cols;
  uint4* lds_in = smem;
  uint4* lds_w = smem + npix * 4;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lc = lane >> 4;
  const int tiles_h = (p.Hg + DTH - 1) / DTH;
  const int b = blockIdx.y / tiles_h;
  const int h0 = (blockIdx.y % tiles_h) * DTH;
  const int w0 = blockIdx.x * DTW;
  const int o0 = blockIdx.z * TO;
  const T* xb = x + (int64_t)b * p.Hin * p.Win * p.Cin;

  f32x4 acc[MF][2];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    acc[mf][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[mf][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int nchunks = p.Cin / (4 * CE);
  const int gh_base = h0 * p.in_stride + p.ioff_h + p.dymin;
  const int gw_base = w0 * p.in_stride + p.ioff_w + p.dxmin;
  for (int cc = 0; cc < nchunks; ++cc) {
    const int c0 = cc * 4 * CE;
    __syncthreads();
    if (!(p.ablate & 4))
    for (int id = tid; id < npix * 4; id += 256) {
      const int pix = id >> 2, ch = id & 3;
      const int iy = pix / p.cols, ix = pix - iy * p.cols;
      int gh = gh_base + iy, gw = gw_base + ix;
      bool zero = false;
      if (p.hzero) zero = gh < 0 || gh >= p.Hin;
      gh = gh < 0 ? 0 : (gh >= p.Hin ? p.Hin - 1 : gh);
      gw = p.ring ? floormod(gw, p.Win) : (gw < 0 ? 0 : (gw >= p.Win ? p.Win - 1 : gw));
      uint4 v = make_uint4(0, 0, 0, 0);
      if (!zero) v = *reinterpret_cast<const uint4*>(xb + ((int64_t)gh * p.Win + gw) * p.Cin + c0 + ch * CE);
      lds_in[pix * 4 + (ch ^ ((pix >> 2) & 3))] = v;
    }
    if (!(p.ablate & 4))
    for (int id = tid; id < TO * p.ntaps * 4; id += 256) {
      const int r = id / (p.ntaps * 4);
      const int rem = id - r * (p.ntaps * 4);
      const int t = rem >> 2, ch = rem & 3;
      const int o = o0 + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (o < p.O) v = *reinterpret_cast<const uint4*>(w + ((int64_t)o * p.wtaps + p.widx[t]) * p.Cin + c0 + ch * CE);
      lds_w[(r * p.ntaps + t) * 4 + (ch ^ ((r >> 2) & 3))] = v;
    }
    __syncthreads();
    if (!(p.ablate & 1))
    for (int t = 0; t < p.ntaps; ++t) {
      uint4 a[MF], bb[2];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int r = mf * 16 + lr;
        a[mf] = lds_w[(r * p.ntaps + t) * 4 + (lc ^ ((r >> 2) & 3))];
      }
      const int prow = (wave * p.in_stride + p.dy[t] - p.dymin) * p.cols + p.dx[t] - p.dxmin;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int pix = prow + (nf * 16 + lr) * p.in_stride;
        bb[nf] = lds_in[pix * 4 + (lc ^ ((pix >> 2) & 3))];
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        Mfma16<T>::run(acc[mf][0], a[mf], bb[0]);
        Mfma16<T>::run(acc[mf][1], a[mf], bb[1]);
      }
    }
  }

  const int gh = h0 + wave;
  if (gh >= p.Hg) return;
  if ((p.ablate & 2) && acc[0][0][0] != 12345.678f) return;
  const int yh = gh * p.out_stride + p.ooff_h;
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    const int gw = w0 + nf * 16 + lr;
    if (gw >= p.Wg) continue;
    const int yw = gw * p.out_stride + p.ooff_w;
    T* row = y + (((int64_t)b * p.Hy + yh) * p.Wy + yw) * p.O;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int o = o0 + mf * 16 + lc * 4;
      if (o >= p.O) continue;
      f32x4 v = acc[mf][nf];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (o + r >= p.O) continue;
        float f = v[r];
        if (p.accumulate) f += to_f32(row[o + r]);
        if (p.bias) f += p.bias[o + r];
        if (p.act == 3) f = (f > 0.f ? f : f * p.alpha) * p.scale;
        v[r] = f;
      }
      if (o + 3 < p.O && (p.O & 3) == 0) {
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(row + o) = v;
        } else {
          union { uint2 u; bf16_t e[4]; } pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) pk.e[r] = (bf16_t)v[r];
          *reinterpret_cast<uint2*>(row + o) = pk.u;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (o + r < p.O) row[o + r] = from_f32<T>(v[r]);
      }
    }
  }
}

// Weights-in-registers variant for the small-channel layers (Cin <= 2 K-steps, TO = 32): the A fragments
// of every (chunk, tap) live in VGPRs for the whole block, which then walks TPB tiles along W staging only
// the input halo tile.  Compared with the generic kernel this removes the weight slab from the per-tile
// staging (it was larger than the input tile) and every LDS read of A.
constexpr int WREG_TPB = 4;

template <typename T, int NCH, int NT>
__global__ __launch_bounds__(256) void conv_direct_wreg_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                               const T* __restrict__ w, DConv p) {
  constexpr int CE = 16 / sizeof(T);
  constexpr int TO = 32, MF = 2;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int npix = p.rows * p.cols;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lc = lane >> 4;
  const int tiles_h = (p.Hg + DTH - 1) / DTH;
  const int tiles_w = (p.Wg + DTW - 1) / DTW;
  const int b = blockIdx.y / tiles_h;
  const int h0 = (blockIdx.y % tiles_h) * DTH;
  const int o0 = blockIdx.z * TO;
  const T* xb = x + (int64_t)b * p.Hin * p.Win * p.Cin;

  uint4 a[NCH][NT][MF];
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int o = o0 + mf * 16 + lr;
        a[cc][t][mf] = make_uint4(0, 0, 0, 0);
        if (o < p.O)
          a[cc][t][mf] = *reinterpret_cast<const uint4*>(w + ((int64_t)o * p.wtaps + p.widx[t]) * p.Cin + cc * 4 * CE + lc * CE);
      }

  const int gh_base = h0 * p.in_stride + p.ioff_h + p.dymin;
  const int gh = h0 + wave;
  for (int tw = blockIdx.x * WREG_TPB; tw < tiles_w && tw < (int)(blockIdx.x + 1) * WREG_TPB; ++tw) {
    const int w0 = tw * DTW;
    const int gw_base = w0 * p.in_stride + p.ioff_w + p.dxmin;
    __syncthreads();
    for (int id = tid; id < npix * 4 * NCH; id += 256) {
      const int cc = id / (npix * 4);
      const int rem = id - cc * (npix * 4);
      const int pix = rem >> 2, ch = rem & 3;
      const int iy = pix / p.cols, ix = pix - iy * p.cols;
      int sh = gh_base + iy, sw = gw_base + ix;
      bool zero = false;
      if (p.hzero) zero = sh < 0 || sh >= p.Hin;
      sh = sh < 0 ? 0 : (sh >= p.Hin ? p.Hin - 1 : sh);
      sw = p.ring ? floormod(sw, p.Win) : (sw < 0 ? 0 : (sw >= p.Win ? p.Win - 1 : sw));
      uint4 v = make_uint4(0, 0, 0, 0);
      if (!zero) v = *reinterpret_cast<const uint4*>(xb + ((int64_t)sh * p.Win + sw) * p.Cin + cc * 4 * CE + ch * CE);
      smem[cc * npix * 4 + pix * 4 + (ch ^ ((pix >> 2) & 3))] = v;
    }
    __syncthreads();
    f32x4 acc[MF][2];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      acc[mf][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[mf][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int prow = (wave * p.in_stride + p.dy[t] - p.dymin) * p.cols + p.dx[t] - p.dxmin;
        uint4 bb[2];
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          const int pix = prow + (nf * 16 + lr) * p.in_stride;
          bb[nf] = smem[cc * npix * 4 + pix * 4 + (lc ^ ((pix >> 2) & 3))];
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          Mfma16<T>::run(acc[mf][0], a[cc][t][mf], bb[0]);
          Mfma16<T>::run(acc[mf][1], a[cc][t][mf], bb[1]);
        }
      }
    if (gh < p.Hg) {
      const int yh = gh * p.out_stride + p.ooff_h;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int gw = w0 + nf * 16 + lr;
        if (gw >= p.Wg) continue;
        const int yw = gw * p.out_stride + p.ooff_w;
        T* row = y + (((int64_t)b * p.Hy + yh) * p.Wy + yw) * p.O;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int o = o0 + mf * 16 + lc * 4;
          if (o >= p.O) continue;
          f32x4 v = acc[mf][nf];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (o + r >= p.O) continue;
            float f = v[r];
            if (p.accumulate) f += to_f32(row[o + r]);
            if (p.bias) f += p.bias[o + r];
            if (p.act == 3) f = (f > 0.f ? f : f * p.alpha) * p.scale;
            v[r] = f;
          }
          if (o + 3 < p.O && (p.O & 3) == 0) {
            if constexpr (sizeof(T) == 4) {
              *reinterpret_cast<f32x4*>(row + o) = v;
            } else {
              union { uint2 u; bf16_t e[4]; } pk;
#pragma unroll
              for (int r = 0; r < 4; ++r) pk.e[r] = (bf16_t)v[r];
              *reinterpret_cast<uint2*>(row + o) = pk.u;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (o + r < p.O) row[o + r] = from_f32<T>(v[r]);
          }
        }
      }
    }
  }
}

template <typename T, int NCH, int NT>
int launch_wreg(void* y, const void* x, const void* w, const DConv& p, hipStream_t st) {
  const size_t lds = sizeof(uint4) * (size_t)p.rows * p.cols * 4 * NCH;
  if (lds > 64 * 1024) return DGV2_EINVAL;
  const int tiles_w = (p.Wg + DTW - 1) / DTW;
  dim3 grid((tiles_w + WREG_TPB - 1) / WREG_TPB, ((p.Hg + DTH - 1) / DTH) * p.B, (p.O + 31) / 32);
  conv_direct_wreg_kernel<T, NCH, NT><<<grid, 256, lds, st>>>((T*)y, (const T*)x, (const T*)w, p);
  return 0;
}

template <typename T, int NCH>
int dispatch_wreg(void* y, const void* x, const void* w, const DConv& p, hipStream_t st) {
  switch (p.ntaps) {
    case 1: return launch_wreg<T, NCH, 1>(y, x, w, p, st);
    case 2: return launch_wreg<T, NCH, 2>(y, x, w, p, st);
    case 3: return launch_wreg<T, NCH, 3>(y, x, w, p, st);
    case 4: return launch_wreg<T, NCH, 4>(y, x, w, p, st);
    case 9: return launch_wreg<T, NCH, 9>(y, x, w, p, st);
    default: return -2;
  }
}

template <typename T, int TO>
int launch_direct(void* y, const void* x, const void* w, const DConv& p, hipStream_t st) {
  const size_t lds = sizeof(uint4) * ((size_t)p.rows * p.cols * 4 + (size_t)TO * p.ntaps * 4);
  if (lds > 160 * 1024) return DGV2_EINVAL;
  auto kern = conv_direct_kernel<T, TO>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid((p.Wg + DTW - 1) / DTW, ((p.Hg + DTH - 1) / DTH) * p.B, (p.O + TO - 1) / TO);
  kern<<<grid, 256, lds, st>>>((T*)y, (const T*)x, (const T*)w, p);
  return 0;
}

}  // namespace

// y[b, gh*out_stride+ooff_h, gw*out_stride+ooff_w, o] (=|+=) act( sum_t sum_c
//     x[b, H(gh*in_stride+ioff_h+dy_t), W(gw*in_stride+ioff_w+dx_t), c] * w[o, widx_t, c] + bias[o] )
// for gh < Hg, gw < Wg.  taps_host: HOST pointer to ntaps triples (dy, dx, widx), ntaps <= 9.
// H(): clamp (hzero = 0) or zero outside [0,Hin) (hzero = 1); W(): wrap (ring) or clamp.
// Cin must be a multiple of 32 (bf16) / 16 (fp32).
extern "C" int dgv2_conv_taps(void* y, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg, int Wg,
                              int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w, int out_stride,
                              int ooff_h, int ooff_w, int ntaps, int wtaps, const int* taps_host, int hzero, int ring,
                              int accumulate, const float* bias, int act, float alpha, float scale, int dtype,
                              void* stream) {
  if (!y || !x || !w || !taps_host || ntaps < 1 || ntaps > 9 || wtaps < 1) return DGV2_EINVAL;
  if (B <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Hg <= 0 || Wg <= 0 || O <= 0 || in_stride < 1 || out_stride < 1)
    return DGV2_EINVAL;
  if (act != 0 && act != 3) return DGV2_EINVAL;
  const int kstep = dtype == DGV2_BF16 ? 32 : 16;
  if (Cin % kstep || !aligned16(x) || !aligned16(w)) return DGV2_EINVAL;
  DConv p;
  p.B = B; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hg = Hg; p.Wg = Wg; p.O = O; p.Hy = Hy; p.Wy = Wy;
  p.in_stride = in_stride; p.ioff_h = ioff_h; p.ioff_w = ioff_w;
  p.out_stride = out_stride; p.ooff_h = ooff_h; p.ooff_w = ooff_w;
  p.ntaps = ntaps; p.wtaps = wtaps;
  int dymin = 1 << 30, dymax = -(1 << 30), dxmin = 1 << 30, dxmax = -(1 << 30);
  for (int t = 0; t < 9; ++t) {
    p.dy[t] = p.dx[t] = p.widx[t] = 0;
    if (t < ntaps) {
      p.dy[t] = taps_host[3 * t]; p.dx[t] = taps_host[3 * t + 1]; p.widx[t] = taps_host[3 * t + 2];
      if (p.widx[t] < 0 || p.widx[t] >= wtaps) return DGV2_EINVAL;
      dymin = p.dy[t] < dymin ? p.dy[t] : dymin; dymax = p.dy[t] > dymax ? p.dy[t] : dymax;
      dxmin = p.dx[t] < dxmin ? p.dx[t] : dxmin; dxmax = p.dx[t] > dxmax ? p.dx[t] : dxmax;
    }
  }
  p.dymin = dymin; p.dxmin = dxmin;
  p.rows = (DTH - 1) * in_stride + (dymax - dymin) + 1;
  p.cols = (DTW - 1) * in_stride + (dxmax - dxmin) + 1;
  p.hzero = hzero; p.ring = ring; p.accumulate = accumulate;
  { static const int abl = getenv("DGV2_ABLATE") ? atoi(getenv("DGV2_ABLATE")) : 0; p.ablate = abl; }
  p.bias = bias; p.act = act; p.alpha = alpha; p.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  const int nchunks = Cin / kstep;
  DGV2_DISPATCH_DTYPE(dtype, {
    // small-channel layers: weights in registers, several tiles per block
    rc = -2;
    static const bool no_wreg = getenv("DGV2_NO_WREG") != nullptr;  // A/B switch for benchmarking
    if (no_wreg) {
    } else if (nchunks == 1) rc = dispatch_wreg<T, 1>(y, x, w, p, st);
    else if (nchunks == 2 && sizeof(T) == 2) rc = dispatch_wreg<T, 2>(y, x, w, p, st);
    if (rc == 0) DGV2_RETURN_LAST();
    if (O <= 16) rc = launch_direct<T, 16>(y, x, w, p, st);
    else if (O <= 32) rc = launch_direct<T, 32>(y, x, w, p, st);
    else rc = launch_direct<T, 64>(y, x, w, p, st);  // TO = 64 keeps LDS <= 50 KB: 3 blocks/CU hide the staging latency
  });
  if (rc) return rc;
  DGV2_RETURN_LAST();
}
