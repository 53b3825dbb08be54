// fp8 (OCP e4m3) operand support for the decimating branch convs of the discriminator (BASELINE configs[4]: "fp8
// activations on CDNA4 MFMA").  What is e4m3 in HBM: the two tensors of a ResidualBlock (reference:
// gans/models/dusty_v2.py:325-345) that exist only as MFMA operands -- hd = blur(act(conv1(x))), the input of the 3x3
// stride-2 conv2, and xs = blur_down(x), the input of the 1x1 skip conv -- written as e4m3 by the FIR kernels that
// produce them (fir_mfma.hip / resample.hip, unit scale: EqualLR + the activation gain keep them O(1)), and the
// weights of those two convs, e4m3 with a power-of-two per-tensor scale (this file).  The residual stream x and the
// activation outputs that a backward pass reads for their sign stay bf16.
// Kernels here: per-tensor amax -> power-of-two scale -> e4m3 weights in the conv engine's [O, kh*kw, C] layout
// (dgv2_fp8_quant_weights), and e4m3 -> bf16 (dgv2_fp8_dequant) for the weight-gradient stream, which contracts the
// saved e4m3 activations against bf16 gradients.
#include "common.h"

namespace {

struct QList {
  const float* src[16];   // fp32 parameters [O, C, kk] (kk = kh * kw)
  fp8_t* dst[16];         // e4m3 [O, kk, C]
  int O[16], C[16], kk[16];
  float eq[16];           // EqualLR factor of the layer
  int n;
};

// amax[l] = max |w| of tensor l (as the bits of a non-negative float: integer max is the float max)
__global__ __launch_bounds__(256) void fp8_amax_kernel(QList q, unsigned* __restrict__ amax) {
  const int l = blockIdx.y;
  const int64_t n = (int64_t)q.O[l] * q.C[l] * q.kk[l];
  float m = 0.f;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(q.src[l][i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(amax + l, __float_as_uint(m));
}

// dst[o][t][c] = e4m3(src[o][c][t] * s), s = 2^floor(log2(448 / amax)) (1 if amax == 0); descale[l] = eq / s: the conv
// epilogue multiplies its fp32 accumulator by it
__global__ __launch_bounds__(256) void fp8_quant_w_kernel(QList q, const unsigned* __restrict__ amax, float* __restrict__ descale) {
  const int l = blockIdx.y;
  const int O = q.O[l], C = q.C[l], kk = q.kk[l];
  const float am = __uint_as_float(amax[l]);
  const float s = am > 0.f ? exp2f(floorf(log2f(DGV2_FP8_MAX / am))) : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) descale[l] = q.eq[l] / s;
  const int64_t units = (int64_t)O * kk * (C >> 3);   // 8 channels (8 output bytes) per thread
  for (int64_t u = blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(u % (C >> 3));
    const int64_t r = u / (C >> 3);
    const int t = (int)(r % kk), o = (int)(r / kk);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = q.src[l][((int64_t)o * C + c8 * 8 + j) * kk + t] * s;
    *reinterpret_cast<uint2*>(q.dst[l] + ((int64_t)o * kk + t) * C + c8 * 8) = pack_fp8x8(f);
  }
}

__global__ __launch_bounds__(256) void fp8_dequant_kernel(bf16_t* __restrict__ y, const fp8_t* __restrict__ x, int64_t n16,
                                                          float scale) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    const uint4 q = reinterpret_cast<const uint4*>(x)[i];
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
    vec16<bf16_t> a, b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float f[4];
      unpack_fp8x4(w[k], f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k < 2) a.set(4 * k + j, f[j] * scale);
        else b.set(4 * (k - 2) + j, f[j] * scale);
      }
    }
    reinterpret_cast<uint4*>(y)[2 * i] = a.raw;
    reinterpret_cast<uint4*>(y)[2 * i + 1] = b.raw;
  }
}

}  // namespace

// n <= 16 conv weights in one launch pair: w8[l] [O,kk,C] = e4m3(src[l] [O,C,kk] * 2^k_l), 2^k_l the largest power of two
// with amax_l * 2^k_l <= 448; descale[l] = eq[l] / 2^k_l.  amax: n zero-initialised words of device scratch (this entry
// clears them).  C % 8 == 0.  Pointer / size tables are HOST arrays (travel by value in the kernel arguments).
// replaces: the `weight * scale` operand of ops.Conv2d (gans/models/ops/common.py:187-210, EqualLR :158-184) for the
// convs that read e4m3 activations.
extern "C" int dgv2_fp8_quant_weights(void* const* w8, const void* const* src, const int* O, const int* C, const int* kk,
                                      const float* eq, int n, float* descale, void* amax, void* stream) {
  if (!w8 || !src || !O || !C || !kk || !eq || !descale || !amax || n < 1 || n > 16) return DGV2_EINVAL;
  QList q;
  q.n = n;
  int64_t most = 0;
  for (int l = 0; l < 16; ++l) {
    q.src[l] = nullptr; q.dst[l] = nullptr; q.O[l] = q.C[l] = q.kk[l] = 0; q.eq[l] = 1.f;
    if (l < n) {
      if (!w8[l] || !src[l] || O[l] <= 0 || C[l] <= 0 || (C[l] & 7) || kk[l] <= 0 || !aligned16(w8[l])) return DGV2_EINVAL;
      q.src[l] = (const float*)src[l]; q.dst[l] = (fp8_t*)w8[l]; q.O[l] = O[l]; q.C[l] = C[l]; q.kk[l] = kk[l]; q.eq[l] = eq[l];
      const int64_t e = (int64_t)O[l] * C[l] * kk[l];
      most = e > most ? e : most;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = dgv2_zero_async(amax, sizeof(unsigned) * n, st);
  if (e != hipSuccess) return (int)e;
  const int gx = grid_for(most / 8, 256, 256);
  fp8_amax_kernel<<<dim3(gx, n), 256, 0, st>>>(q, (unsigned*)amax);
  fp8_quant_w_kernel<<<dim3(gx, n), 256, 0, st>>>(q, (const unsigned*)amax, descale);
  DGV2_RETURN_LAST();
}

// y (bf16) = x (e4m3) * scale, n elements, n % 16 == 0.
extern "C" int dgv2_fp8_dequant(void* y, const void* x, int64_t n, float scale, void* stream) {
  if (!y || !x || n <= 0 || (n & 15) || !aligned16(x) || !aligned16(y)) return DGV2_EINVAL;
  fp8_dequant_kernel<<<grid_for(n / 16, 256), 256, 0, (hipStream_t)stream>>>((bf16_t*)y, (const fp8_t*)x, n / 16, scale);
  DGV2_RETURN_LAST();
}
