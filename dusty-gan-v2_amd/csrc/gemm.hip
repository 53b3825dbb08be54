// Batched channel GEMMs: the contraction of the modulated 1x1 convolution and its gradients.
// Reference: the grouped F.conv2d of ModConv2d.forward (gans/models/ops/style.py:105-118).
#include "gemm_core.h"

namespace {

template <typename T, typename TY, int TO>
int launch_bmm_nn(void* y, const void* x, const void* w, int B, int P, int I, int O, int ldx, int ldy,
                  int64_t wstride, const float* bias, int act, float alpha, float scale, hipStream_t st, float* sumsq,
                  int sumsq_cap, int* sumsq_used, const float* row_scale, const void* resid) {
  constexpr int CE = 16 / sizeof(T);
  DenseRowLoader<T> al{(const T*)w, wstride, I, O, I, (I % CE == 0) && (wstride % CE == 0) && aligned16(w)};
  DenseRowLoader<T> bl{(const T*)x, (int64_t)P * ldx, ldx, P, I, (ldx % CE == 0) && aligned16(x)};
  constexpr int YE = 16 / sizeof(TY);
  StoreEpilogue<TY> epi{(TY*)y, (int64_t)P * ldy, ldy, O, P,
                        (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & (4 * sizeof(TY) - 1)) == 0),
                        bias, act, alpha, scale, nullptr, 0.f, row_scale, (const TY*)resid};
  (void)YE;
  dim3 grid((P + 127) / 128, (O + TO - 1) / TO, B);
  const int64_t nblk = (int64_t)grid.x * grid.y * grid.z;
  if (sumsq && sumsq_used && nblk <= sumsq_cap) {
    epi.sumsq = sumsq;
    *sumsq_used = (int)nblk;
  }
  gemm_nn_kernel<T, TO, DenseRowLoader<T>, DenseRowLoader<T>, StoreEpilogue<TY>><<<grid, 256, 0, st>>>(al, bl, epi, I);
  return 0;
}

template <typename T, typename TY>
int dispatch_bmm_nn(void* y, const void* x, const void* w, int B, int P, int I, int O, int ldx, int ldy,
                    int64_t wstride, const float* bias, int act, float alpha, float scale, hipStream_t st, float* sumsq,
                    int sumsq_cap, int* sumsq_used, const float* row_scale, const void* resid) {
  if (O <= 16) return launch_bmm_nn<T, TY, 16>(y, x, w, B, P, I, O, ldx, ldy, wstride, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale, resid);
  if (O <= 32) return launch_bmm_nn<T, TY, 32>(y, x, w, B, P, I, O, ldx, ldy, wstride, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale, resid);
  if (O <= 64) return launch_bmm_nn<T, TY, 64>(y, x, w, B, P, I, O, ldx, ldy, wstride, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale, resid);
  return launch_bmm_nn<T, TY, 128>(y, x, w, B, P, I, O, ldx, ldy, wstride, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale, resid);
}

template <typename T, int TO, int TJ>
int launch_bmm_tn(float* gw, const void* gy, const void* x, int B, int P, int I, int O, int ldgy, int ldx,
                  int ksplit, hipStream_t st) {
  constexpr int CE = 16 / sizeof(T);
  DenseKLoader<T> al{(const T*)gy, (int64_t)P * ldgy, ldgy, O, (ldgy % CE == 0) && aligned16(gy)};
  DenseKLoader<T> bl{(const T*)x, (int64_t)P * ldx, ldx, I, (ldx % CE == 0) && aligned16(x)};
  const int64_t klen = (((int64_t)P + ksplit - 1) / ksplit + 31) / 32 * 32;
  dim3 grid((I + TJ - 1) / TJ, (O + TO - 1) / TO, B * ksplit);
  gemm_tn_kernel<T, TO, TJ, DenseKLoader<T>, DenseKLoader<T>><<<grid, 256, 0, st>>>(al, bl, gw, O, I, P, klen, ksplit,
                                                                                   (int64_t)O * I, I);
  return 0;
}

template <typename T, typename TY, int TO>
int launch_bmm_nn_cat(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka, int Ks, int O,
                      const float* bias, int act, float alpha, float scale, hipStream_t st, float* sumsq, int sumsq_cap,
                      int* sumsq_used, const float* row_scale) {
  const int K = Ka + Ks;
  DenseRowLoader<T> al{(const T*)w, (int64_t)O * K, K, O, K, true};
  ConcatRowLoader<T> bl{(const T*)xa, (int64_t)P * Ka, Ka, Ka, (const T*)xs, Ks, Ks, P};
  StoreEpilogue<TY> epi{(TY*)y, (int64_t)P * O, O, O, P, (O % 4 == 0), bias, act, alpha, scale, nullptr, 0.f, row_scale};
  dim3 grid(B, (O + TO - 1) / TO, (P + 127) / 128);
  const int64_t nblk = (int64_t)grid.x * grid.y * grid.z;
  if (sumsq && sumsq_used && nblk <= sumsq_cap) {
    epi.sumsq = sumsq;
    *sumsq_used = (int)nblk;
  }
  gemm_nn_kernel<T, TO, DenseRowLoader<T>, ConcatRowLoader<T>, StoreEpilogue<TY>, true><<<grid, 256, 0, st>>>(al, bl, epi, K);
  return 0;
}

template <typename T, typename TY>
int dispatch_bmm_nn_cat(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka, int Ks, int O,
                        const float* bias, int act, float alpha, float scale, hipStream_t st, float* sumsq, int sumsq_cap,
                        int* sumsq_used, const float* row_scale) {
  if (O <= 16) return launch_bmm_nn_cat<T, TY, 16>(y, xa, xs, w, B, P, Ka, Ks, O, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale);
  if (O <= 32) return launch_bmm_nn_cat<T, TY, 32>(y, xa, xs, w, B, P, Ka, Ks, O, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale);
  if (O <= 64) return launch_bmm_nn_cat<T, TY, 64>(y, xa, xs, w, B, P, Ka, Ks, O, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale);
  return launch_bmm_nn_cat<T, TY, 128>(y, xa, xs, w, B, P, Ka, Ks, O, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale);
}

template <typename T, int TO, int TJ>
int launch_bmm_tn_cat(float* gw, const void* gy, const void* xa, const void* xs, int B, int P, int Ca, int Cs, int O,
                      int ksplit, hipStream_t st) {
  constexpr int CE = 16 / sizeof(T);
  const int J = Ca + Cs;
  DenseKLoader<T> al{(const T*)gy, (int64_t)P * O, O, O, (O % CE == 0) && aligned16(gy)};
  ConcatKLoader<T> bl{(const T*)xa, (int64_t)P * Ca, Ca, Ca, (const T*)xs, Cs, Cs};
  const int64_t klen = (((int64_t)P + ksplit - 1) / ksplit + 31) / 32 * 32;
  dim3 grid((J + TJ - 1) / TJ, (O + TO - 1) / TO, B * ksplit);
  gemm_tn_kernel<T, TO, TJ, DenseKLoader<T>, ConcatKLoader<T>><<<grid, 256, 0, st>>>(al, bl, gw, O, J, P, klen, ksplit,
                                                                                  (int64_t)O * J, J);
  return 0;
}

}  // namespace

extern "C" int dgv2_bmm_nn(void* y, const void* x, const void* w, int B, int P, int I, int O, int ldx, int ldy,
                           int64_t wstride, const float* bias, int act, float alpha, float scale, int dtype,
                           int ydtype, void* stream) {
  return dgv2_bmm_nn_sq(y, x, w, B, P, I, O, ldx, ldy, wstride, nullptr, bias, act, alpha, scale, nullptr, dtype, ydtype,
                        nullptr, 0, nullptr, stream);
}

extern "C" int dgv2_bmm_nn_sq(void* y, const void* x, const void* w, int B, int P, int I, int O, int ldx, int ldy,
                              int64_t wstride, const float* row_scale, const float* bias, int act, float alpha,
                              float scale, const void* resid, int dtype, int ydtype, float* sumsq, int sumsq_cap,
                              int* sumsq_used, void* stream) {
  if (sumsq_used) *sumsq_used = 0;
  if (!y || !x || !w || B <= 0 || P <= 0 || I <= 0 || O <= 0 || ldx < I || ldy < O) return DGV2_EINVAL;
  if (act != 0 && act != 3) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DGV2_F32 && ydtype == DGV2_F32)
    dispatch_bmm_nn<float, float>(y, x, w, B, P, I, O, ldx, ldy, wstride, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale, resid);
  else if (dtype == DGV2_BF16 && ydtype == DGV2_BF16)
    dispatch_bmm_nn<bf16_t, bf16_t>(y, x, w, B, P, I, O, ldx, ldy, wstride, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale, resid);
  else if (dtype == DGV2_BF16 && ydtype == DGV2_F32)
    dispatch_bmm_nn<bf16_t, float>(y, x, w, B, P, I, O, ldx, ldy, wstride, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale, resid);
  else
    return DGV2_EINVAL;
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_bmm_tn(float* gw, const void* gy, const void* x, int B, int P, int I, int O, int ldgy, int ldx,
                           int dtype, void* stream) {
  if (!gw || !gy || !x || B <= 0 || P <= 0 || I <= 0 || O <= 0 || ldgy < O || ldx < I) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // enough blocks to fill 256 CUs: split the pixel axis when (batch x tiles) is small
  const int tiles = B * ((O + 63) / 64) * ((I + 127) / 128);
  int ksplit = 1;
  while (tiles * ksplit < 512 && P / (ksplit * 2) >= 512) ksplit *= 2;
  if (ksplit > 1) {
    hipError_t e = hipMemsetAsync(gw, 0, sizeof(float) * (size_t)B * O * I, st);
    if (e != hipSuccess) return (int)e;
  }
  DGV2_DISPATCH_DTYPE(dtype, {
    if (O <= 16) launch_bmm_tn<T, 16, 128>(gw, gy, x, B, P, I, O, ldgy, ldx, ksplit, st);
    else if (O <= 32) launch_bmm_tn<T, 32, 128>(gw, gy, x, B, P, I, O, ldgy, ldx, ksplit, st);
    else launch_bmm_tn<T, 64, 128>(gw, gy, x, B, P, I, O, ldgy, ldx, ksplit, st);
  });
  DGV2_RETURN_LAST();
}

// Level-input conv of the generator with the positional encoding kept batch-shared:
//   y[b,p,o] = act( sum_{k<Ka} xa[b,p,k] w[b,o,k] + sum_{k<Ks} xs[p,k] w[b,o,Ka+k] + bias[o] )
// xa = FIR-upsampled previous activation (per sample), xs = PE of the UNSHIFTED angle grid (one copy for
// the whole batch; the per-sample azimuth shift is a rotation folded into w by the host, see
// gans/models/dusty_v2.py).  Replaces cat(h, PE) + grouped conv (dusty_v2.py:153-161, style.py:105-118).
extern "C" int dgv2_bmm_nn_cat(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka, int Ks,
                               int O, const float* bias, int act, float alpha, float scale, int dtype, int ydtype,
                               void* stream) {
  return dgv2_bmm_nn_cat_sq(y, xa, xs, w, B, P, Ka, Ks, O, nullptr, bias, act, alpha, scale, dtype, ydtype, nullptr, 0, nullptr,
                            stream);
}

extern "C" int dgv2_bmm_nn_cat_sq(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka, int Ks,
                                  int O, const float* row_scale, const float* bias, int act, float alpha, float scale, int dtype, int ydtype,
                                  float* sumsq, int sumsq_cap, int* sumsq_used, void* stream) {
  if (sumsq_used) *sumsq_used = 0;
  if (!y || !xs || !w || (Ka > 0 && !xa) || B <= 0 || P <= 0 || Ka < 0 || Ks <= 0 || O <= 0) return DGV2_EINVAL;
  if (act != 0 && act != 3) return DGV2_EINVAL;
  const int ce = dtype == DGV2_BF16 ? 8 : 4;
  if (Ka % ce || Ks % ce || !aligned16(xs) || (Ka > 0 && !aligned16(xa)) || !aligned16(w) || !aligned16(y))
    return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DGV2_F32 && ydtype == DGV2_F32)
    dispatch_bmm_nn_cat<float, float>(y, xa, xs, w, B, P, Ka, Ks, O, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale);
  else if (dtype == DGV2_BF16 && ydtype == DGV2_BF16)
    dispatch_bmm_nn_cat<bf16_t, bf16_t>(y, xa, xs, w, B, P, Ka, Ks, O, bias, act, alpha, scale, st, sumsq, sumsq_cap, sumsq_used, row_scale);
  else
    return DGV2_EINVAL;
  DGV2_RETURN_LAST();
}

// Weight gradient of the above: gw[b,o,:] = sum_p gy[b,p,o] * [xa[b,p,:] | xs[p,:]]   (fp32 [B,O,Ka+Ks]).
extern "C" int dgv2_bmm_tn_cat(float* gw, const void* gy, const void* xa, const void* xs, int B, int P, int Ka,
                               int Ks, int O, int dtype, void* stream) {
  if (!gw || !gy || !xs || (Ka > 0 && !xa) || B <= 0 || P <= 0 || Ka < 0 || Ks <= 0 || O <= 0) return DGV2_EINVAL;
  const int ce = dtype == DGV2_BF16 ? 8 : 4;
  if (Ka % ce || Ks % ce || !aligned16(xs) || (Ka > 0 && !aligned16(xa))) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int J = Ka + Ks;
  const int tiles = B * ((O + 63) / 64) * ((J + 127) / 128);
  int ksplit = 1;
  while (tiles * ksplit < 512 && P / (ksplit * 2) >= 512) ksplit *= 2;
  if (ksplit > 1) {
    hipError_t e = hipMemsetAsync(gw, 0, sizeof(float) * (size_t)B * O * J, st);
    if (e != hipSuccess) return (int)e;
  }
  DGV2_DISPATCH_DTYPE(dtype, {
    if (O <= 16) launch_bmm_tn_cat<T, 16, 128>(gw, gy, xa, xs, B, P, Ka, Ks, O, ksplit, st);
    else if (O <= 32) launch_bmm_tn_cat<T, 32, 128>(gw, gy, xa, xs, B, P, Ka, Ks, O, ksplit, st);
    else launch_bmm_tn_cat<T, 64, 128>(gw, gy, xa, xs, B, P, Ka, Ks, O, ksplit, st);
  });
  DGV2_RETURN_LAST();
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the output heads: gw[b,o,i] = sum_p gy[b,p,o] x[b,p,i] with O <= 4 output channels (image,
// ray-drop logit).  As a GEMM it fills 2 of 16 MFMA rows and ran at a quarter of the streaming rate; it is a
// weighted column sum of x, so: a thread owns one 16-byte channel vector and a lane of pixels, partial sums meet
// in LDS and leave as one fp32 atomic per (o, i) and block (gw cleared first; B * nsplit blocks).
// ------------------------------------------------------------------------------------------------
namespace {

template <typename T, int O>
__global__ __launch_bounds__(256) void bmm_tn_small_kernel(float* __restrict__ gw, const T* __restrict__ gy,
                                                           const T* __restrict__ x, int P, int I, int ppb) {
  constexpr int VN = vec16<T>::N;
  __shared__ float red[256 * VN];
  const int b = blockIdx.y;
  const int cvecs = I / VN, lanes = 256 / cvecs;
  const int cv = threadIdx.x % cvecs, pl = threadIdx.x / cvecs;
  const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, P);
  float acc[O][VN];
#pragma unroll
  for (int o = 0; o < O; ++o)
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[o][j] = 0.f;
  const T* xb = x + (int64_t)b * P * I;
  const T* gb = gy + (int64_t)b * P * O;
  constexpr int U = 4;   // pixels in flight per thread
  for (int p = p0 + pl; p < p1; p += lanes * U) {
    vec16<T> v[U];
    float g[U][O];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = p + u * lanes;
      const int qc = min(q, p1 - 1);
      v[u].load(xb + (int64_t)qc * I + cv * VN);
#pragma unroll
      for (int o = 0; o < O; ++o) g[u][o] = q < p1 ? to_f32(gb[(int64_t)qc * O + o]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int o = 0; o < O; ++o)
#pragma unroll
        for (int j = 0; j < VN; ++j) acc[o][j] = fmaf(g[u][o], v[u].get(j), acc[o][j]);
  }
#pragma unroll
  for (int o = 0; o < O; ++o) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VN; ++j) red[threadIdx.x * VN + j] = acc[o][j];
    __syncthreads();
    for (int c = threadIdx.x; c < I; c += 256) {
      const int v = c / VN, j = c - v * VN;
      float s = 0.f;
      for (int k = 0; k < lanes; ++k) s += red[(k * cvecs + v) * VN + j];
      atomicAdd(&gw[((int64_t)b * O + o) * I + c], s);
    }
  }
}

}  // namespace

// gw fp32 [B, O, I] (overwritten); gy [B, P, O], x [B, P, I] in `dtype`; 1 <= O <= 4, I a multiple of the 16-byte
// vector with I / vector dividing 256.  DGV2_ENOTSUP otherwise (use dgv2_bmm_tn).
extern "C" int dgv2_bmm_tn_small(float* gw, const void* gy, const void* x, int B, int P, int I, int O, int dtype,
                                 void* stream) {
  if (!gw || !gy || !x || B <= 0 || P <= 0 || I <= 0 || O <= 0) return DGV2_EINVAL;
  const int vn = dtype == DGV2_BF16 ? 8 : (dtype == DGV2_F32 ? 4 : 0);
  if (!vn) return DGV2_EINVAL;
  if (O > 4 || I % vn || 256 % (I / vn) || !aligned16(x)) return DGV2_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(gw, 0, sizeof(float) * (size_t)B * O * I, st);
  if (e != hipSuccess) return (int)e;
  const int lanes = 256 / (I / vn);
  int nsplit = (1024 + B - 1) / B;                       // ~1024 blocks
  const int minpix = lanes * 16;                         // >= 16 pixels per thread
  if ((int64_t)nsplit * minpix > P) nsplit = (P + minpix - 1) / minpix;
  nsplit = nsplit < 1 ? 1 : nsplit;
  const int ppb = (P + nsplit - 1) / nsplit;
  dim3 grid((P + ppb - 1) / ppb, B);
  DGV2_DISPATCH_DTYPE(dtype, {
    switch (O) {
      case 1: bmm_tn_small_kernel<T, 1><<<grid, 256, 0, st>>>(gw, (const T*)gy, (const T*)x, P, I, ppb); break;
      case 2: bmm_tn_small_kernel<T, 2><<<grid, 256, 0, st>>>(gw, (const T*)gy, (const T*)x, P, I, ppb); break;
      case 3: bmm_tn_small_kernel<T, 3><<<grid, 256, 0, st>>>(gw, (const T*)gy, (const T*)x, P, I, ppb); break;
      default: bmm_tn_small_kernel<T, 4><<<grid, 256, 0, st>>>(gw, (const T*)gy, (const T*)x, P, I, ppb); break;
    }
  });
  DGV2_RETURN_LAST();
}

// ------------------------------------------------------------------------------------------------
// Data gradient of the output heads: y[b,p,k] = sum_{o<O} x[b,p,o] w[b,k,o] (+ resid[b,p,k]) with a contraction of
// only O <= 4 terms -- an outer-product stream, not a GEMM (K = 2 padded to 32 wasted 15/16 of the MFMA work and
// ran at 2.6 TB/s of output).  A thread owns one 16-byte vector of k for a lane of pixels; the sample's w rows for
// those k live in registers.
// ------------------------------------------------------------------------------------------------
namespace {

// ACT: y is the gradient of the OUTPUT of an upstream bias + leaky-ReLU layer whose forward output is `ref` (same layout
// as y): apply that layer's activation backward here -- store v * row_scale[k] with v = (sum + resid) * (ref > 0 ? 1 :
// alpha) * ascale, and leave per-block column sums of the rounded v in `partial` [blocks, K] for its bias gradient.
template <typename T, int O, bool ACT>
__global__ __launch_bounds__(256) void bmm_nn_small_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                           const T* __restrict__ w, const T* __restrict__ resid, int P,
                                                           int K, int ppb, const T* __restrict__ ref = nullptr,
                                                           const float* __restrict__ row_scale = nullptr, float alpha = 1.f,
                                                           float ascale = 1.f, float* __restrict__ partial = nullptr) {
  constexpr int VN = vec16<T>::N;
  __shared__ float red[ACT ? 256 * VN : 1];
  const int b = blockIdx.y;
  const int kvecs = K / VN, lanes = 256 / kvecs;
  const int kv = threadIdx.x % kvecs, pl = threadIdx.x / kvecs;
  float wr[VN][O];
#pragma unroll
  for (int j = 0; j < VN; ++j)
#pragma unroll
    for (int o = 0; o < O; ++o) wr[j][o] = to_f32(w[((int64_t)b * K + kv * VN + j) * O + o]);
  const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, P);
  const T* xb = x + (int64_t)b * P * O;
  float rs[VN], bsum[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) {
    rs[j] = (ACT && row_scale) ? row_scale[kv * VN + j] : 1.f;
    bsum[j] = 0.f;
  }
  for (int p = p0 + pl; p < p1; p += lanes) {
    float g[O];
#pragma unroll
    for (int o = 0; o < O; ++o) g[o] = to_f32(xb[(int64_t)p * O + o]);
    const int64_t off = ((int64_t)b * P + p) * K + kv * VN;
    vec16<T> r, f, out;
    if (resid) r.load(resid + off);
    if constexpr (ACT) f.load(ref + off);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      float s = resid ? r.get(j) : 0.f;
#pragma unroll
      for (int o = 0; o < O; ++o) s = fmaf(g[o], wr[j][o], s);
      if constexpr (ACT) {
        const float v = (f.get(j) > 0.f ? s : s * alpha) * ascale;
        out.set(j, v);
        bsum[j] += out.get(j);          // the bias gradient sums the rounded, unscaled gradient
        out.set(j, v * rs[j]);
      } else {
        out.set(j, s);
      }
    }
    out.store(y + off);
  }
  if constexpr (ACT) {
#pragma unroll
    for (int j = 0; j < VN; ++j) red[threadIdx.x * VN + j] = bsum[j];
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += 256) {
      const int v = c / VN, j = c - v * VN;
      float s2 = 0.f;
      for (int t = 0; t < lanes; ++t) s2 += red[(t * kvecs + v) * VN + j];
      partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * K + c] = s2;
    }
  }
}

// gb[c] = sum_blk partial[blk][c] (one wave per channel)
__global__ __launch_bounds__(256) void small_bias_reduce_kernel(float* __restrict__ gb, const float* __restrict__ partial,
                                                             int nblk, int C) {
  // one BLOCK per channel: thousands of per-block partials (one row per producer block) fold in ~nblk/256 steps
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s = 0.f;
#pragma unroll 4
  for (int k = threadIdx.x; k < nblk; k += 256) s += partial[(int64_t)k * C + c];
  s = block_sum(s, red);
  if (threadIdx.x == 0) gb[c] = s;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// The WHOLE backward of a level's output heads in one streaming pass (round 5).  The heads' input x [B, P, K] (the
// trunk's activation) is read once and serves as the operand of the head weight gradient AND as the `ref` of the
// upstream layer's activation backward; the skip gradient gy [B, P, O] fp32 (O <= 4 head channels) is read once:
//   g[p, o]   = T(gy[p, o] * cvec[o])                                  (the heads' accumulator gradient, rounded to T)
//   y[p, k]   = ((sum_o g[p, o] w[b, k, o]) + resid[p, k]) * (x > 0 ? 1 : alpha) * ascale * row_scale[k]
//   gb_up[k]  = sum_{b, p} of the unscaled, rounded y                  (upstream bias gradient)
//   gw[b,o,k] = sum_p g[p, o] x[p, k]                                  (head weight gradient, per sample)
//   gbh[o]    = sum_{b, p} gy[p, o]                                    (head bias gradient, unscaled fp32)
// replaces five launches per level (column sum, scale + cast, dgv2_bmm_nn_small_act, dgv2_bmm_tn_small and their zero
// fills / reducers) that read x twice and gy three times.  Partials per block, folded by ONE reduce launch.
// ------------------------------------------------------------------------------------------------
namespace {

template <typename T, int O>
__global__ __launch_bounds__(256) void head_bwd_kernel(T* __restrict__ y, const float* __restrict__ gyf,
                                                       const float* __restrict__ cvec, const T* __restrict__ w,
                                                       const T* __restrict__ resid, const T* __restrict__ ref, int P, int K,
                                                       int ppb, const float* __restrict__ row_scale, float alpha,
                                                       float ascale, float* __restrict__ p_gb, float* __restrict__ p_gw,
                                                       float* __restrict__ p_gbh) {
  constexpr int VN = vec16<T>::N;
  __shared__ float red[256 * VN];
  const int b = blockIdx.y;
  const int kvecs = K / VN, lanes = 256 / kvecs;
  const int kv = threadIdx.x % kvecs, pl = threadIdx.x / kvecs;
  float wr[VN][O], cv[O];
#pragma unroll
  for (int j = 0; j < VN; ++j)
#pragma unroll
    for (int o = 0; o < O; ++o) wr[j][o] = to_f32(w[((int64_t)b * K + kv * VN + j) * O + o]);
#pragma unroll
  for (int o = 0; o < O; ++o) cv[o] = cvec[o];
  const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, P);
  const float* gb_ = gyf + (int64_t)b * P * O;
  float rs[VN], bsum[VN], hw[O][VN], hb[O];
#pragma unroll
  for (int j = 0; j < VN; ++j) {
    rs[j] = row_scale ? row_scale[kv * VN + j] : 1.f;
    bsum[j] = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) hw[o][j] = 0.f;
  }
#pragma unroll
  for (int o = 0; o < O; ++o) hb[o] = 0.f;
  for (int p = p0 + pl; p < p1; p += lanes) {
    float g[O];
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const float raw = gb_[(int64_t)p * O + o];
      if (kv == 0) hb[o] += raw;                      // one thread per pixel sums the unscaled gradient
      g[o] = to_f32(from_f32<T>(raw * cv[o]));
    }
    const int64_t off = ((int64_t)b * P + p) * K + kv * VN;
    vec16<T> r, f, out;
    if (resid) r.load(resid + off);
    f.load(ref + off);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const float xv = f.get(j);
      float s = resid ? r.get(j) : 0.f;
#pragma unroll
      for (int o = 0; o < O; ++o) {
        s = fmaf(g[o], wr[j][o], s);
        hw[o][j] = fmaf(g[o], xv, hw[o][j]);
      }
      const float v = (xv > 0.f ? s : s * alpha) * ascale;
      out.set(j, v);
      bsum[j] += out.get(j);          // the bias gradient sums the rounded, unscaled gradient
      out.set(j, v * rs[j]);
    }
    out.store(y + off);
  }
  // per-block partials: the `lanes` pixel lanes of a channel vector fold through LDS, one quantity at a time
  const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
  for (int q = 0; q <= O; ++q) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VN; ++j) red[threadIdx.x * VN + j] = q == 0 ? bsum[j] : hw[q > 0 ? q - 1 : 0][j];
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += 256) {
      const int v = c / VN, j = c - v * VN;
      float s2 = 0.f;
      for (int t = 0; t < lanes; ++t) s2 += red[(t * kvecs + v) * VN + j];
      if (q == 0) p_gb[blk * K + c] = s2;
      else p_gw[(blk * O + (q - 1)) * K + c] = s2;
    }
  }
  __syncthreads();
#pragma unroll
  for (int o = 0; o < O; ++o) red[threadIdx.x * O + o] = kv == 0 ? hb[o] : 0.f;
  __syncthreads();
  if (threadIdx.x < O) {
    float s2 = 0.f;
    for (int t = 0; t < 256; ++t) s2 += red[t * O + threadIdx.x];
    p_gbh[blk * O + threadIdx.x] = s2;
  }
}

// blocks [0, K): gb_up[c];  [K, K + B*O): gw[b, o, :] over the sample's nsplit blocks;  [K + B*O, K + B*O + O): gbh[o]
__global__ __launch_bounds__(256) void head_bwd_reduce_kernel(float* __restrict__ gb, float* __restrict__ gw,
                                                              float* __restrict__ gbh, const float* __restrict__ p_gb,
                                                              const float* __restrict__ p_gw, const float* __restrict__ p_gbh,
                                                              int nsplit, int B, int O, int K) {
  __shared__ float red[16];
  const int nblk = nsplit * B;
  int id = blockIdx.x;
  if (id < K) {
    float s = 0.f;
#pragma unroll 4
    for (int k = threadIdx.x; k < nblk; k += 256) s += p_gb[(int64_t)k * K + id];
    s = block_sum(s, red);
    if (threadIdx.x == 0) gb[id] = s;
    return;
  }
  id -= K;
  if (id < B * O) {
    const int b = id / O, o = id - b * O;
    for (int c = threadIdx.x; c < K; c += 256) {
      float s = 0.f;
      for (int k = 0; k < nsplit; ++k) s += p_gw[(((int64_t)b * nsplit + k) * O + o) * K + c];
      gw[((int64_t)b * O + o) * K + c] = s;
    }
    return;
  }
  id -= B * O;
  float s = 0.f;
  for (int k = threadIdx.x; k < nblk; k += 256) s += p_gbh[(int64_t)k * O + id];
  s = block_sum(s, red);
  if (threadIdx.x == 0) gbh[id] = s;
}

}  // namespace

// See the section comment above.  gyf fp32 [B, P, O], cvec fp32 [O], w [B, K, O] / resid / ref / y [B, P, K] in `dtype`;
// gb_up fp32 [K], gw fp32 [B, O, K], gbh fp32 [O] (all overwritten).  scratch: fp32 [>= *blocks_needed * (K + O K + O)];
// y == NULL with blocks_needed: query only.  DGV2_ENOTSUP: O > 4, K not a multiple of the 16-byte vector or K / vector not
// dividing 256 (callers then run the separate launches).
extern "C" int dgv2_head_bwd(void* y, float* gw, float* gbh, float* gb_up, float* scratch, int64_t scratch_elems,
                             int64_t* blocks_needed, const float* gyf, const float* cvec, const void* w, const void* resid,
                             const void* ref, const float* row_scale, float alpha, float ascale, int B, int P, int O, int K,
                             int dtype, void* stream) {
  if (B <= 0 || P <= 0 || O <= 0 || K <= 0) return DGV2_EINVAL;
  const int vn = dtype == DGV2_BF16 ? 8 : (dtype == DGV2_F32 ? 4 : 0);
  if (!vn) return DGV2_EINVAL;
  if (O > 4 || K % vn || 256 % (K / vn)) return DGV2_ENOTSUP;
  const int lanes = 256 / (K / vn);
  int nsplit = (2048 + B - 1) / B;
  const int minpix = lanes * 8;
  if ((int64_t)nsplit * minpix > P) nsplit = (P + minpix - 1) / minpix;
  nsplit = nsplit < 1 ? 1 : nsplit;
  const int ppb = (P + nsplit - 1) / nsplit;
  dim3 grid((P + ppb - 1) / ppb, B);
  const int64_t nblk = (int64_t)grid.x * grid.y;
  if (blocks_needed) *blocks_needed = nblk;
  if (blocks_needed && !y) return 0;   // query only
  if (!y || !gw || !gbh || !gb_up || !scratch || !gyf || !cvec || !w || !ref) return DGV2_EINVAL;
  if (scratch_elems < nblk * ((int64_t)K + (int64_t)O * K + O)) return DGV2_EINVAL;
  if (!aligned16(y) || !aligned16(ref) || (resid && !aligned16(resid))) return DGV2_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  float* p_gb = scratch;
  float* p_gw = p_gb + nblk * K;
  float* p_gbh = p_gw + nblk * O * K;
  DGV2_DISPATCH_DTYPE(dtype, {
    switch (O) {
      case 1: head_bwd_kernel<T, 1><<<grid, 256, 0, st>>>((T*)y, gyf, cvec, (const T*)w, (const T*)resid, (const T*)ref, P, K, ppb, row_scale, alpha, ascale, p_gb, p_gw, p_gbh); break;
      case 2: head_bwd_kernel<T, 2><<<grid, 256, 0, st>>>((T*)y, gyf, cvec, (const T*)w, (const T*)resid, (const T*)ref, P, K, ppb, row_scale, alpha, ascale, p_gb, p_gw, p_gbh); break;
      case 3: head_bwd_kernel<T, 3><<<grid, 256, 0, st>>>((T*)y, gyf, cvec, (const T*)w, (const T*)resid, (const T*)ref, P, K, ppb, row_scale, alpha, ascale, p_gb, p_gw, p_gbh); break;
      default: head_bwd_kernel<T, 4><<<grid, 256, 0, st>>>((T*)y, gyf, cvec, (const T*)w, (const T*)resid, (const T*)ref, P, K, ppb, row_scale, alpha, ascale, p_gb, p_gw, p_gbh); break;
    }
  });
  head_bwd_reduce_kernel<<<K + B * O + O, 256, 0, st>>>(gb_up, gw, gbh, p_gb, p_gw, p_gbh, (int)grid.x, B, O, K);
  DGV2_RETURN_LAST();
}

// y [B, P, K] = x [B, P, O] . w [B, K, O]^T (+ resid [B, P, K]), all in `dtype`; 1 <= O <= 4, K a multiple of the
// 16-byte vector with K / vector dividing 256.  DGV2_ENOTSUP otherwise (use dgv2_bmm_nn).
extern "C" int dgv2_bmm_nn_small(void* y, const void* x, const void* w, const void* resid, int B, int P, int O, int K,
                                 int dtype, void* stream) {
  return dgv2_bmm_nn_small_act(y, x, w, resid, B, P, O, K, nullptr, nullptr, 1.f, 1.f, nullptr, nullptr, 0, nullptr, dtype,
                               stream);
}

// ... followed (ref != NULL) by the activation backward of the upstream layer that produced this operator's input:
//   y = (x.w^T + resid) * (ref > 0 ? 1 : alpha) * ascale * row_scale[k],  gb[k] = column sums of the unscaled result.
// scratch: fp32 [>= *blocks_needed * K]; scratch == NULL only reports *blocks_needed.
extern "C" int dgv2_bmm_nn_small_act(void* y, const void* x, const void* w, const void* resid, int B, int P, int O, int K,
                                     const void* ref, const float* row_scale, float alpha, float ascale, float* gb,
                                     float* scratch, int64_t scratch_elems, int64_t* blocks_needed, int dtype,
                                     void* stream) {
  if (B <= 0 || P <= 0 || O <= 0 || K <= 0) return DGV2_EINVAL;
  if (!blocks_needed && (!y || !x || !w)) return DGV2_EINVAL;
  const int vn = dtype == DGV2_BF16 ? 8 : (dtype == DGV2_F32 ? 4 : 0);
  if (!vn) return DGV2_EINVAL;
  if (O > 4 || K % vn || 256 % (K / vn) || !aligned16(y) || (resid && !aligned16(resid))) return DGV2_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  const int lanes = 256 / (K / vn);
  int nsplit = (2048 + B - 1) / B;
  const int minpix = lanes * 8;
  if ((int64_t)nsplit * minpix > P) nsplit = (P + minpix - 1) / minpix;
  nsplit = nsplit < 1 ? 1 : nsplit;
  const int ppb = (P + nsplit - 1) / nsplit;
  dim3 grid((P + ppb - 1) / ppb, B);
  const int64_t nblk = (int64_t)grid.x * grid.y;
  if (blocks_needed) *blocks_needed = nblk;
  if (blocks_needed && !y) return 0;   // query only
  if (ref) {
    if (!gb || !scratch || scratch_elems < nblk * K || !aligned16(ref)) return DGV2_EINVAL;
    DGV2_DISPATCH_DTYPE(dtype, {
      switch (O) {
        case 1: bmm_nn_small_kernel<T, 1, true><<<grid, 256, 0, st>>>((T*)y, (const T*)x, (const T*)w, (const T*)resid, P, K, ppb, (const T*)ref, row_scale, alpha, ascale, scratch); break;
        case 2: bmm_nn_small_kernel<T, 2, true><<<grid, 256, 0, st>>>((T*)y, (const T*)x, (const T*)w, (const T*)resid, P, K, ppb, (const T*)ref, row_scale, alpha, ascale, scratch); break;
        case 3: bmm_nn_small_kernel<T, 3, true><<<grid, 256, 0, st>>>((T*)y, (const T*)x, (const T*)w, (const T*)resid, P, K, ppb, (const T*)ref, row_scale, alpha, ascale, scratch); break;
        default: bmm_nn_small_kernel<T, 4, true><<<grid, 256, 0, st>>>((T*)y, (const T*)x, (const T*)w, (const T*)resid, P, K, ppb, (const T*)ref, row_scale, alpha, ascale, scratch); break;
      }
    });
    small_bias_reduce_kernel<<<K, 256, 0, st>>>(gb, scratch, (int)nblk, K);
    DGV2_RETURN_LAST();
  }
  DGV2_DISPATCH_DTYPE(dtype, {
    switch (O) {
      case 1: bmm_nn_small_kernel<T, 1, false><<<grid, 256, 0, st>>>((T*)y, (const T*)x, (const T*)w, (const T*)resid, P, K, ppb); break;
      case 2: bmm_nn_small_kernel<T, 2, false><<<grid, 256, 0, st>>>((T*)y, (const T*)x, (const T*)w, (const T*)resid, P, K, ppb); break;
      case 3: bmm_nn_small_kernel<T, 3, false><<<grid, 256, 0, st>>>((T*)y, (const T*)x, (const T*)w, (const T*)resid, P, K, ppb); break;
      default: bmm_nn_small_kernel<T, 4, false><<<grid, 256, 0, st>>>((T*)y, (const T*)x, (const T*)w, (const T*)resid, P, K, ppb); break;
    }
  });
  DGV2_RETURN_LAST();
}
