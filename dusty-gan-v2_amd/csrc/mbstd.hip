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

template <typename T>
__global__ __launch_bounds__(256) void mbstd_cat_kernel(T* __restrict__ out, const T* __restrict__ x,
                                                        const float* __restrict__ partial, MbGeom q) {
  constexpr int VN = vec16<T>::N;
  const int cvp = q.Cp / VN, cvx = q.C / VN;
  const int64_t total = (int64_t)q.B * q.P * cvp;
  const float inv = 1.f / ((float)q.P * q.C);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cvp);
    const int64_t bp = i / cvp;
    vec16<T> o;
    if (cv < cvx) {
      o.load(x + bp * q.C + cv * VN);
    } else {
      o.raw = make_uint4(0, 0, 0, 0);
      if (cv == cvx) {
        const int b = (int)(bp / q.P);
        const int sg = b / q.m, mi = b - sg * q.m, s = sg / q.g;   // b = (s*g + gi)*m + mi
        float st = 0.f;
        for (int k = 0; k < q.NS; ++k) st += partial[(s * q.m + mi) * q.NS + k];
        o.set(0, st * inv);
      }
    }
    o.store(out + bp * q.Cp + cv * VN);
  }
}

// gx[b, p, c] = gout[b, p, c] + gst[s, mi] / (P C) * (x_gi - mu) / (g sd),   gst = sum_{gi, p} gout[b, p, C]
template <typename T>
__global__ __launch_bounds__(256) void mbstd_bwd_kernel(T* __restrict__ gx, const T* __restrict__ gout,
                                                        const T* __restrict__ x, MbGeom q) {
  __shared__ float red[16];
  __shared__ float s_gst;
  constexpr int VN = vec16<T>::N;
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
    float coef[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      mu[j] *= ig;
      float var = 0.f;
#pragma unroll
      for (int gi = 0; gi < MB_G; ++gi)
        if (gi < q.g) {
          const float d = a[gi].get(j) - mu[j];
          var = fmaf(d, d, var);
        }
      coef[j] = k / sqrtf(var * ig + 1e-8f);
    }
#pragma unroll
    for (int gi = 0; gi < MB_G; ++gi) {
      if (gi < q.g) {
        const int b = (s * q.g + gi) * q.m + mi;
        vec16<T> go, o;
        go.load(gout + ((int64_t)b * q.P + p) * q.Cp + cv * VN);
#pragma unroll
        for (int j = 0; j < VN; ++j) o.set(j, go.get(j) + coef[j] * (a[gi].get(j) - mu[j]));
        o.store(gx + (int64_t)b * PC + v * VN);
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

// x [B, P, C], out [B, P, Cp] (same dtype), scratch fp32 [>= 64 * B / group]; B = splits * group * m.
extern "C" int dgv2_mbstd_cat_fwd(void* out, float* scratch, const void* x, int B, int P, int C, int Cp, int splits,
                                  int group, int dtype, void* stream) {
  if (!out || !scratch || !x || splits < 1 || group < 1 || B % (splits * group)) return DGV2_EINVAL;
  MbGeom q{B, P, C, Cp, splits, group, B / (splits * group), 1};
  const int vn = dtype == DGV2_BF16 ? 8 : 4;
  if (!mb_ok(q, vn) || !aligned16(out) || !aligned16(x)) return DGV2_EINVAL;
  q.NS = mb_splits(q, vn);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(q.S * q.m, q.NS);
  const int64_t total = (int64_t)B * P * (Cp / vn);
  DGV2_DISPATCH_DTYPE(dtype, {
    mbstd_partial_kernel<T><<<grid, 256, 0, st>>>(scratch, (const T*)x, q);
    mbstd_cat_kernel<T><<<grid_for(total, 256, 4096), 256, 0, st>>>((T*)out, (const T*)x, scratch, q);
  });
  DGV2_RETURN_LAST();
}

// gx [B, P, C] from gout [B, P, Cp] (gradient of the concatenated tensor; the padding channels carry none) and x.
extern "C" int dgv2_mbstd_cat_bwd(void* gx, const void* gout, const void* x, int B, int P, int C, int Cp, int splits,
                                  int group, int dtype, void* stream) {
  if (!gx || !gout || !x || splits < 1 || group < 1 || B % (splits * group)) return DGV2_EINVAL;
  MbGeom q{B, P, C, Cp, splits, group, B / (splits * group), 1};
  const int vn = dtype == DGV2_BF16 ? 8 : 4;
  if (!mb_ok(q, vn) || !aligned16(gx) || !aligned16(gout) || !aligned16(x)) return DGV2_EINVAL;
  q.NS = mb_splits(q, vn);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(q.S * q.m, q.NS);
  DGV2_DISPATCH_DTYPE(dtype, { mbstd_bwd_kernel<T><<<grid, 256, 0, st>>>((T*)gx, (const T*)gout, (const T*)x, q); });
  DGV2_RETURN_LAST();
}
