// Minibatch standard deviation + channel concat of the discriminator epilogue as two launches forward, one backward.
// Reference: MinibatchStdDev, gans/models/ops/common.py:226-250 (group members strided through the batch) followed by
// torch.cat([x, stat]) -- about ten elementwise / reduction launches each way on a [B, 4, 32, 512] activation.
//   y   = x.reshape(S, g, m, P, C)            sample b = (s*g + gi)*m + mi   (S = independent sub-batches)
//   sd  = sqrt(var_gi(y, biased) + 1e-8)      [S, m, P, C]
//   st  = mean_{P,C}(sd)                      [S, m]  -> channel C of every member of the group
//   out = [x | st | 0 ...]                    [B, P, Cp]   (Cp >= C + 1: channel padding for the conv engine)
// The statistic is taken in fp32 on the stored values (x.float() in the reference); one feature (mbdis_feat = 1).
#include "common.h"

namespace {

constexpr int MB_G = 8;   // largest group

struct MbGeom {
  int B, P, C, Cp, S, g, m, NS;
};

// N consecutive elements of T as floats (N * sizeof(T) is 16 or 32 bytes, 16-byte aligned)
template <typename T, int N> __device__ __forceinline__ void ldN(const T* p, float (&f)[N]) {
  constexpr int V = vec16<T>::N;
#pragma unroll
  for (int h = 0; h < N / V; ++h) {
    vec16<T> v;
    v.load(p + h * V);
#pragma unroll
    for (int j = 0; j < V; ++j) f[h * V + j] = v.get(j);
  }
}
template <typename T, int N> __device__ __forceinline__ void stN(T* p, const float (&f)[N]) {
  constexpr int V = vec16<T>::N;
#pragma unroll
  for (int h = 0; h < N / V; ++h) {
    vec16<T> v;
#pragma unroll
    for (int j = 0; j < V; ++j) v.set(j, f[h * V + j]);
    v.store(p + h * V);
  }
}
template <typename TA, typename TB> struct mb_vn {
  static constexpr int value = vec16<TA>::N > vec16<TB>::N ? vec16<TA>::N : vec16<TB>::N;
};

// partial[sm, split] = sum over this split's (p, c) positions of sd
template <typename T>
__global__ __launch_bounds__(256) void mbstd_partial_kernel(float* __restrict__ partial, const T* __restrict__ x, MbGeom q) {
  __shared__ float red[16];
  constexpr int VN = vec16<T>::N;
  const int sm = blockIdx.x, split = blockIdx.y;
  const int s = sm / q.m, mi = sm - s * q.m;
  const int64_t PC = (int64_t)q.P * q.C;
  const int64_t nvec = PC / VN;
  float acc = 0.f;
  for (int64_t v = (int64_t)split * 256 + threadIdx.x; v < nvec; v += (int64_t)q.NS * 256) {
    vec16<T> a[MB_G];
    float mu[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) mu[j] = 0.f;
#pragma unroll
    for (int gi = 0; gi < MB_G; ++gi) {
      if (gi < q.g) {
        const int b = (s * q.g + gi) * q.m + mi;
        a[gi].load(x + (int64_t)b * PC + v * VN);
#pragma unroll
        for (int j = 0; j < VN; ++j) mu[j] += a[gi].get(j);
      }
    }
    const float ig = 1.f / q.g;
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const float mean = mu[j] * ig;
      float var = 0.f;   // two-pass, like torch.var
#pragma unroll
      for (int gi = 0; gi < MB_G; ++gi)
        if (gi < q.g) {
          const float d = a[gi].get(j) - mean;
          var = fmaf(d, d, var);
        }
      acc += sqrtf(var * ig + 1e-8f);
    }
  }
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) partial[sm * q.NS + split] = t;
}

// TX -> TY: the cast of the reference's x.float() ahead of its fp32 epilogue (dusty_v2.py:394-395) rides in this pass
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void mbstd_cat_kernel(TY* __restrict__ out, const TX* __restrict__ x,
                                                        const float* __restrict__ partial, MbGeom q) {
  constexpr int VN = mb_vn<TX, TY>::value;
  const int cvp = q.Cp / VN, cvx = q.C / VN;
  const int64_t total = (int64_t)q.B * q.P * cvp;
  const float inv = 1.f / ((float)q.P * q.C);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cvp);
    const int64_t bp = i / cvp;
    float o[VN];
    if (cv < cvx) {
      ldN<TX, VN>(x + bp * q.C + cv * VN, o);
    } else {
#pragma unroll
      for (int j = 0; j < VN; ++j) o[j] = 0.f;
      if (cv == cvx) {
        const int b = (int)(bp / q.P);
        const int sg = b / q.m, mi = b - sg * q.m, s = sg / q.g;   // b = (s*g + gi)*m + mi
        float st = 0.f;
        for (int k = 0; k < q.NS; ++k) st += partial[(s * q.m + mi) * q.NS + k];
        o[0] = st * inv;
      }
    }
    stN<TY, VN>(out + bp * q.Cp + cv * VN, o);
  }
}

// gx[b, p, c] = gout[b, p, c] + gst[s, mi] / (P C) * (x_gi - mu) / (g sd),   gst = sum_{gi, p} gout[b, p, C]
template <typename TX, typename TG>
__global__ __launch_bounds__(256) void mbstd_bwd_kernel(TX* __restrict__ gx, const TG* __restrict__ gout,
                                                        const TX* __restrict__ x, MbGeom q) {
  __shared__ float red[16];
  __shared__ float s_gst;
  constexpr int VN = mb_vn<TX, TG>::value;
  const int sm = blockIdx.x, split = blockIdx.y;
  const int s = sm / q.m, mi = sm - s * q.m;
  const int64_t PC = (int64_t)q.P * q.C;
  float part = 0.f;
  for (int e = threadIdx.x; e < q.g * q.P; e += 256) {
    const int gi = e / q.P, p = e - gi * q.P;
    const int b = (s * q.g + gi) * q.m + mi;
    part += to_f32(gout[((int64_t)b * q.P + p) * q.Cp + q.C]);
  }
  part = block_sum(part, red);
  if (threadIdx.x == 0) s_gst = part / ((float)q.P * q.C);
  __syncthreads();
  const float k = s_gst / q.g;
  const int cvx = q.C / VN;
  const int64_t nvec = PC / VN;
  for (int64_t v = (int64_t)split * 256 + threadIdx.x; v < nvec; v += (int64_t)q.NS * 256) {
    const int64_t p = v / cvx;
    const int cv = (int)(v - p * cvx);
    float a[MB_G][VN];
    float mu[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) mu[j] = 0.f;
#pragma unroll
    for (int gi = 0; gi < MB_G; ++gi) {
      if (gi < q.g) {
        const int b = (s * q.g + gi) * q.m + mi;
        ldN<TX, VN>(x + (int64_t)b * PC + v * VN, a[gi]);
#pragma unroll
        for (int j = 0; j < VN; ++j) mu[j] += a[gi][j];
      }
    }
    const float ig = 1.f / q.g;
    float coef[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      mu[j] *= ig;
      float var = 0.f;
#pragma unroll
      for (int gi = 0; gi < MB_G; ++gi)
        if (gi < q.g) {
          const float d = a[gi][j] - mu[j];
          var = fmaf(d, d, var);
        }
      coef[j] = k / sqrtf(var * ig + 1e-8f);
    }
#pragma unroll
    for (int gi = 0; gi < MB_G; ++gi) {
      if (gi < q.g) {
        const int b = (s * q.g + gi) * q.m + mi;
        float go[VN], o[VN];
        ldN<TG, VN>(gout + ((int64_t)b * q.P + p) * q.Cp + cv * VN, go);
#pragma unroll
        for (int j = 0; j < VN; ++j) o[j] = go[j] + coef[j] * (a[gi][j] - mu[j]);
        stN<TX, VN>(gx + (int64_t)b * PC + v * VN, o);
      }
    }
  }
}

bool mb_ok(const MbGeom& q, int vn) {
  return q.B > 0 && q.P > 0 && q.C > 0 && q.S > 0 && q.g > 0 && q.g <= MB_G && q.m > 0 && q.S * q.g * q.m == q.B &&
         q.C % vn == 0 && q.Cp % vn == 0 && q.Cp > q.C && (int64_t)q.B * q.P * q.Cp < (1LL << 40);
}

int mb_splits(const MbGeom& q, int vn) {
  const int64_t nvec = (int64_t)q.P * q.C / vn;
  int ns = (int)((nvec + 1023) / 1024);          // >= 4 vectors per thread
  const int want = (1024 + q.S * q.m - 1) / (q.S * q.m);   // ~1024 blocks in flight
  ns = ns < want ? ns : want;
  return ns < 1 ? 1 : (ns > 64 ? 64 : ns);
}

}  // namespace

// x [B, P, C] (xdtype), out [B, P, Cp] (ydtype: the same, or fp32 from a bf16 x -- the epilogue's cast in the same
// pass), scratch fp32 [>= 64 * B / group]; B = splits * group * m.
extern "C" int dgv2_mbstd_cat_fwd_x(void* out, float* scratch, const void* x, int B, int P, int C, int Cp, int splits,
                                    int group, int xdtype, int ydtype, void* stream) {
  if (!out || !scratch || !x || splits < 1 || group < 1 || B % (splits * group)) return DGV2_EINVAL;
  if (xdtype != ydtype && !(xdtype == DGV2_BF16 && ydtype == DGV2_F32)) return DGV2_ENOTSUP;
  MbGeom q{B, P, C, Cp, splits, group, B / (splits * group), 1};
  const int vx = xdtype == DGV2_BF16 ? 8 : 4;
  if (!mb_ok(q, vx) || !aligned16(out) || !aligned16(x)) return DGV2_EINVAL;
  q.NS = mb_splits(q, vx);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(q.S * q.m, q.NS);
  const int64_t total = (int64_t)B * P * (Cp / vx);
  const int gc = grid_for(total, 256, 4096);
  if (xdtype == DGV2_F32) {
    mbstd_partial_kernel<float><<<grid, 256, 0, st>>>(scratch, (const float*)x, q);
    mbstd_cat_kernel<float, float><<<gc, 256, 0, st>>>((float*)out, (const float*)x, scratch, q);
  } else if (xdtype == DGV2_BF16) {
    mbstd_partial_kernel<bf16_t><<<grid, 256, 0, st>>>(scratch, (const bf16_t*)x, q);
    if (ydtype == DGV2_BF16)
      mbstd_cat_kernel<bf16_t, bf16_t><<<gc, 256, 0, st>>>((bf16_t*)out, (const bf16_t*)x, scratch, q);
    else
      mbstd_cat_kernel<bf16_t, float><<<gc, 256, 0, st>>>((float*)out, (const bf16_t*)x, scratch, q);
  } else {
    return DGV2_EINVAL;
  }
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_mbstd_cat_fwd(void* out, float* scratch, const void* x, int B, int P, int C, int Cp, int splits,
                                  int group, int dtype, void* stream) {
  return dgv2_mbstd_cat_fwd_x(out, scratch, x, B, P, C, Cp, splits, group, dtype, dtype, stream);
}

// gx [B, P, C] (xdtype, as x) from gout [B, P, Cp] (gdtype: gradient of the concatenated tensor; the padding channels
// carry none) and x.  gdtype == xdtype, or fp32 gout with bf16 x / gx (the adjoint of the fused cast).
extern "C" int dgv2_mbstd_cat_bwd_x(void* gx, const void* gout, const void* x, int B, int P, int C, int Cp, int splits,
                                    int group, int xdtype, int gdtype, void* stream) {
  if (!gx || !gout || !x || splits < 1 || group < 1 || B % (splits * group)) return DGV2_EINVAL;
  if (xdtype != gdtype && !(xdtype == DGV2_BF16 && gdtype == DGV2_F32)) return DGV2_ENOTSUP;
  MbGeom q{B, P, C, Cp, splits, group, B / (splits * group), 1};
  const int vx = xdtype == DGV2_BF16 ? 8 : 4;
  if (!mb_ok(q, vx) || !aligned16(gx) || !aligned16(gout) || !aligned16(x)) return DGV2_EINVAL;
  q.NS = mb_splits(q, vx);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(q.S * q.m, q.NS);
  if (xdtype == DGV2_F32)
    mbstd_bwd_kernel<float, float><<<grid, 256, 0, st>>>((float*)gx, (const float*)gout, (const float*)x, q);
  else if (xdtype == DGV2_BF16 && gdtype == DGV2_BF16)
    mbstd_bwd_kernel<bf16_t, bf16_t><<<grid, 256, 0, st>>>((bf16_t*)gx, (const bf16_t*)gout, (const bf16_t*)x, q);
  else if (xdtype == DGV2_BF16)
    mbstd_bwd_kernel<bf16_t, float><<<grid, 256, 0, st>>>((bf16_t*)gx, (const float*)gout, (const bf16_t*)x, q);
  else
    return DGV2_EINVAL;
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_mbstd_cat_bwd(void* gx, const void* gout, const void* x, int B, int P, int C, int Cp, int splits,
                                  int group, int dtype, void* stream) {
  return dgv2_mbstd_cat_bwd_x(gx, gout, x, B, P, C, Cp, splits, group, dtype, dtype, stream);
}
