// Per-sample weight preparation of the modulated convolution and its exact backward.
//
// Reference: ModConv2d.forward, gans/models/ops/style.py:72-103 -- about a dozen elementwise /
// reduction ops on [B,O,I] tensors per layer (plus their autograd twins), 19 layers per generator
// pass.  Here the whole chain is two launches forward and three backward, writing the GEMM operand
// directly in the compute dtype:
//
//   demod:     w' = W / max|W|            s' = s_b / max_i|s_b|        t = s' + 1
//              m  = w' * t                d  = rsqrt(sum_i m^2 + 1e-8)  wb = m * d * c
//   no demod:  wb = (W * scale) * (s_b + 1) * c                         (heads)
//   c = 1 / (sqrt(ema_var) + 1e-8)
//   optional rotation of the positional-encoding columns (batch-shared PE, see gemm.hip):
//              [wb_sin, wb_cos] <- [wb_sin cos(d) - wb_cos sin(d), wb_sin sin(d) + wb_cos cos(d)],
//              d = shift_b * f_w  (applied after the demodulation: rotations preserve the norm)
//
// Layout: W fp32 [O,I]; s fp32 [B,I]; wb [B, Otot, I] with this layer's rows at [row_off, row_off+O)
// (the two heads of a level share one buffer / one GEMM).  The PE columns are [cin, cin+2F) with
// F == 256 == blockDim, so a thread owns both members (i, i+F) of every rotation pair.
#include "common.h"

namespace {

constexpr int PB = 256;   // block size == number of PE frequencies
constexpr int MAXJ = 4;   // I <= PB * MAXJ = 1024

struct ModGeom {
  int B, O, I, Otot, row_off;
  int demod, cin, F;          // F = 0: no rotation
  float scale;                // 1/sqrt(I) (used when !demod)
};

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// stats[0] = max|W|, stats[2+2b] = max_i|s_b[i]| (odd slots: number of maxima, taken as 1 -- ties have
// measure zero for real-valued parameters).  Many blocks per operand; the maximum is combined with an
// integer atomicMax on the bit pattern of |x| (non-negative floats order like unsigned ints).  The slots
// are zeroed by a memset node ahead of the launch.
__global__ __launch_bounds__(PB) void mod_stats_kernel(float* __restrict__ stats, const float* __restrict__ W,
                                                       const float* __restrict__ s, int OI, int I, int wblocks) {
  __shared__ float red[4];
  float m = 0.f;
  int slot;
  if ((int)blockIdx.x < wblocks) {
    for (int i = blockIdx.x * PB + threadIdx.x; i < OI; i += wblocks * PB) m = fmaxf(m, fabsf(W[i]));
    slot = 0;
  } else {
    const int b = blockIdx.x - wblocks;
    for (int i = threadIdx.x; i < I; i += PB) m = fmaxf(m, fabsf(s[(int64_t)b * I + i]));
    slot = 2 + 2 * b;
  }
  m = block_max(m, red);
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(stats + slot), __float_as_uint(m));
}

// pre-rotation modulated weight m[j] (j-th owned column i = tid + j*PB) and t
__device__ __forceinline__ void mod_row(const ModGeom& g, const float* __restrict__ W, const float* __restrict__ s,
                                        const float* __restrict__ stats, int b, int o, float (&m)[MAXJ],
                                        float (&t)[MAXJ], float (&wp)[MAXJ]) {
  const float inv_wmax = g.demod ? 1.f / stats[0] : g.scale;
  const float inv_smax = g.demod ? 1.f / stats[2 + 2 * b] : 1.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = threadIdx.x + j * PB;
    m[j] = t[j] = wp[j] = 0.f;
    if (i < g.I) {
      wp[j] = W[(int64_t)o * g.I + i] * inv_wmax;
      t[j] = s[(int64_t)b * g.I + i] * inv_smax + 1.f;
      m[j] = wp[j] * t[j];
    }
  }
}

// rotation pair bookkeeping: column i = tid + j*PB is a "sin" column when cin <= i < cin+F, its "cos"
// partner is i + F = column j+1 of the same thread (F == PB).
__device__ __forceinline__ bool is_sin_col(const ModGeom& g, int j) {
  const int i = threadIdx.x + j * PB;
  return g.F > 0 && i >= g.cin && i < g.cin + g.F;
}

template <typename TO>
__global__ __launch_bounds__(PB) void mod_prep_fwd_kernel(TO* __restrict__ wb, float* __restrict__ dsave,
                                                          const float* __restrict__ W, const float* __restrict__ s,
                                                          const float* __restrict__ stats,
                                                          const float* __restrict__ ema_var,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ fw, ModGeom g) {
  __shared__ float red[4];
  const int o = blockIdx.x, b = blockIdx.y;
  float m[MAXJ], t[MAXJ], wp[MAXJ];
  mod_row(g, W, s, stats, b, o, m, t, wp);
  float d = 1.f;
  if (g.demod) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) ss += m[j] * m[j];
    ss = block_sum(ss, red);
    d = rsqrtf(ss + 1e-8f);
  }
  const float c = 1.f / (sqrtf(ema_var[0]) + 1e-8f);
  if (threadIdx.x == 0) dsave[(int64_t)b * g.O + o] = d;
  const float k = d * c;
  if (g.F > 0 && shift) {
#pragma unroll
    for (int j = 0; j + 1 < MAXJ; ++j)
      if (is_sin_col(g, j)) {
        const int f = threadIdx.x + j * PB - g.cin;
        float sd, cd;
        sincosf(shift[b] * fw[f], &sd, &cd);
        const float ws = m[j], wc = m[j + 1];
        m[j] = ws * cd - wc * sd;
        m[j + 1] = ws * sd + wc * cd;
      }
  }
  TO* out = wb + ((int64_t)b * g.Otot + g.row_off + o) * g.I;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = threadIdx.x + j * PB;
    if (i < g.I) out[i] = from_f32<TO>(m[j] * k);
  }
}

// gradient of the loss w.r.t. the pre-rotation modulated weight m (per owned column), given G = dL/dwb
__device__ __forceinline__ void grad_m(const ModGeom& g, const float* __restrict__ G, const float* __restrict__ shift,
                                       const float* __restrict__ fw, int b, int o, const float (&m)[MAXJ], float d,
                                       float c, float* red, float (&gm)[MAXJ]) {
  const float* Gr = G + ((int64_t)b * g.Otot + g.row_off + o) * g.I;
  float gp[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = threadIdx.x + j * PB;
    gp[j] = i < g.I ? Gr[i] : 0.f;
  }
  if (g.F > 0 && shift) {  // transpose of the rotation
#pragma unroll
    for (int j = 0; j + 1 < MAXJ; ++j)
      if (is_sin_col(g, j)) {
        const int f = threadIdx.x + j * PB - g.cin;
        float sd, cd;
        sincosf(shift[b] * fw[f], &sd, &cd);
        const float gs = gp[j], gc = gp[j + 1];
        gp[j] = gs * cd + gc * sd;
        gp[j + 1] = -gs * sd + gc * cd;
      }
  }
  if (g.demod) {
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) r += gp[j] * m[j];
    r = block_sum(r, red);
    const float k = c * d, d2r = d * d * r;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) gm[j] = k * (gp[j] - m[j] * d2r);
  } else {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) gm[j] = c * gp[j];
  }
}

// A block owns an OG x BG group of (o, b) pairs: gm per pair, then
//   gWraw[o,i] += sum_b gm t_b[i],   gt[b,i] += sum_o gm w'[o,i],   corr += sum gm t w'
// with the sums over the group kept in registers, so the fp32 atomics that combine groups are OG (gt) and
// BG (gWraw) times fewer than one per pair -- same-address float atomics are what bounds this kernel.
template <int OG, int BG>
__global__ __launch_bounds__(PB) void mod_prep_bwd_kernel(float* __restrict__ gWraw, float* __restrict__ gt,
                                                          float* __restrict__ corr, const float* __restrict__ G,
                                                          const float* __restrict__ W, const float* __restrict__ s,
                                                          const float* __restrict__ stats,
                                                          const float* __restrict__ dsave,
                                                          const float* __restrict__ ema_var,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ fw, ModGeom g, int corr_slots) {
  __shared__ float red[4];
  const int o0 = blockIdx.x * OG, b0 = blockIdx.y * BG;
  const float c = 1.f / (sqrtf(ema_var[0]) + 1e-8f);
  float gwacc[OG][MAXJ];
#pragma unroll
  for (int ol = 0; ol < OG; ++ol)
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) gwacc[ol][j] = 0.f;
  float part = 0.f;
#pragma unroll
  for (int bl = 0; bl < BG; ++bl) {
    const int b = b0 + bl;
    if (b >= g.B) break;
    float gtacc[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) gtacc[j] = 0.f;
#pragma unroll
    for (int ol = 0; ol < OG; ++ol) {
      const int o = o0 + ol;
      if (o >= g.O) break;
      float m[MAXJ], t[MAXJ], wp[MAXJ], gm[MAXJ];
      mod_row(g, W, s, stats, b, o, m, t, wp);
      grad_m(g, G, shift, fw, b, o, m, dsave[(int64_t)b * g.O + o], c, red, gm);
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const float gw = gm[j] * t[j];
        gwacc[ol][j] += gw;
        gtacc[j] += gm[j] * wp[j];
        part += gw * wp[j];
      }
    }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int i = threadIdx.x + j * PB;
      if (i < g.I) atomicAdd(&gt[(int64_t)b * g.I + i], gtacc[j]);
    }
  }
#pragma unroll
  for (int ol = 0; ol < OG; ++ol) {
    const int o = o0 + ol;
    if (o >= g.O) break;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int i = threadIdx.x + j * PB;
      if (i < g.I) atomicAdd(&gWraw[(int64_t)o * g.I + i], gwacc[ol][j]);
    }
  }
  if (g.demod) {
    part = block_sum(part, red);
    // one slot per block when the caller provided them (plain store: the buffer was cleared, nobody else writes
    // the slot); 1000+ same-address atomics would otherwise serialise into the longest part of this kernel
    if (threadIdx.x == 0) {
      const int blk = blockIdx.y * gridDim.x + blockIdx.x;
      if (corr_slots > 1) corr[blk] = part;
      else atomicAdd(corr, part);
    }
  }
}

// s' = s / smax:  gs_i = gt_i / smax - [|s_i| == smax] sign(s_i) (sum_j gt_j s_j) / smax^2   (in place on gt)
__global__ __launch_bounds__(PB) void mod_prep_bwd_s_fix_kernel(float* __restrict__ gs, const float* __restrict__ s,
                                                                const float* __restrict__ stats, int I) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float smax = stats[2 + 2 * b];
  float acc[MAXJ], sv[MAXJ], dot = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = threadIdx.x + j * PB;
    acc[j] = i < I ? gs[(int64_t)b * I + i] : 0.f;
    sv[j] = i < I ? s[(int64_t)b * I + i] : 0.f;
    dot += acc[j] * sv[j];
  }
  dot = block_sum(dot, red);
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = threadIdx.x + j * PB;
    if (i < I) {
      float v = acc[j] / smax;
      if (fabsf(sv[j]) == smax) v -= (sv[j] > 0.f ? 1.f : -1.f) * dot / (smax * smax);
      gs[(int64_t)b * I + i] = v;
    }
  }
}

// W' = W * k (k = 1/wmax or scale): gW = gWraw * k, and with demod the max-norm term
//   gW_i -= [|W_i| == wmax] sign(W_i) corr / wmax,  corr = sum gWraw w'.
__global__ __launch_bounds__(256) void mod_prep_bwd_w_fix_kernel(float* __restrict__ gW, const float* __restrict__ W,
                                                                 const float* __restrict__ stats,
                                                                 const float* __restrict__ corr, int ncorr, int OI,
                                                                 int demod, float scale) {
  __shared__ float red[4];
  const float wmax = demod ? stats[0] : 1.f;
  const float k = demod ? 1.f / wmax : scale;
  float csum = 0.f;
  if (demod) {   // fold the per-block partial sums
    for (int t = threadIdx.x; t < ncorr; t += 256) csum += corr[t];
    csum = wave_sum(csum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = csum;
    __syncthreads();
    csum = red[0] + red[1] + red[2] + red[3];
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < OI; i += gridDim.x * blockDim.x) {
    float v = gW[i] * k;
    if (demod) {
      const float w = W[i];
      if (fabsf(w) == wmax) v -= (w > 0.f ? 1.f : -1.f) * csum / wmax;
    }
    gW[i] = v;
  }
}

bool geom_ok(int B, int O, int I, int Otot, int row_off, int cin, int F) {
  return B > 0 && O > 0 && I > 0 && I <= PB * MAXJ && Otot >= row_off + O && row_off >= 0 &&
         (F == 0 || (F == PB && cin >= 0 && cin + 2 * F <= I));
}

}  // namespace

// Input-magnitude EMA of ModConv2d (style.py:98-103) as one scalar kernel:
//   ema <- lerp(ema, (sum(sumsq[0..nsum)) + add) * inv_count, weight)   (when update)
//   snapshot <- ema        (the value this forward pass uses; the buffer itself keeps changing)
static __global__ void ema_scalar_kernel(float* ema, float* snapshot, const float* sumsq, int nsum, float add,
                                         float inv_count, float weight, int update) {
  float s = 0.f;   // one wave: fold the nsum partial sums of dgv2_sum_squares
  if (sumsq)
    for (int k = threadIdx.x; k < nsum; k += 64) s += sumsq[k];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    float v = ema[0];
    if (update) {
      v += weight * ((s + add) * inv_count - v);
      ema[0] = v;
    }
    snapshot[0] = v;
  }
}

extern "C" int dgv2_ema_scalar(float* ema, float* snapshot, const float* sumsq, int nsum, float add, float inv_count,
                               float weight, int update, void* stream) {
  if (!ema || !snapshot || nsum < 0) return DGV2_EINVAL;
  ema_scalar_kernel<<<1, 64, 0, (hipStream_t)stream>>>(ema, snapshot, sumsq, nsum, add, inv_count, weight, update);
  DGV2_RETURN_LAST();
}

// Forward.  stats: fp32 [2 + 2B] scratch (filled here when demod); dsave: fp32 [B,O] (saved for backward).
extern "C" int dgv2_mod_prep_fwd(void* wb, float* dsave, float* stats, const float* W, const float* s,
                                 const float* ema_var, const float* shift, const float* fw, int B, int O, int I,
                                 int Otot, int row_off, int demod, int cin, int F, int wb_dtype, void* stream) {
  if (!wb || !dsave || !stats || !W || !s || !ema_var || !geom_ok(B, O, I, Otot, row_off, cin, F)) return DGV2_EINVAL;
  if (F > 0 && shift && !fw) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  ModGeom g{B, O, I, Otot, row_off, demod, cin, (F > 0 && shift) ? F : 0, 1.f / sqrtf((float)I)};
  if (demod) {
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(float) * (2 + 2 * B), st);
    if (e != hipSuccess) return (int)e;
    const int wblocks = (O * I + PB * 8 - 1) / (PB * 8);
    mod_stats_kernel<<<wblocks + B, PB, 0, st>>>(stats, W, s, O * I, I, wblocks);
  }
  dim3 grid(O, B);
  if (wb_dtype == DGV2_F32)
    mod_prep_fwd_kernel<float><<<grid, PB, 0, st>>>((float*)wb, dsave, W, s, stats, ema_var, shift, fw, g);
  else if (wb_dtype == DGV2_BF16)
    mod_prep_fwd_kernel<bf16_t><<<grid, PB, 0, st>>>((bf16_t*)wb, dsave, W, s, stats, ema_var, shift, fw, g);
  else
    return DGV2_EINVAL;
  DGV2_RETURN_LAST();
}

// Backward.  G fp32 [B,Otot,I] = dL/dwb; outputs gW fp32 [O,I], gs fp32 [B,I]; corr: fp32 [corr_elems] scratch
// (>= 1; with >= one slot per launched block -- O*B is always enough -- the max-norm correction is summed without
// same-address atomics).
extern "C" int dgv2_mod_prep_bwd(float* gW, float* gs, float* corr, const float* G, const float* W, const float* s,
                                 const float* stats, const float* dsave, const float* ema_var, const float* shift,
                                 const float* fw, int B, int O, int I, int Otot, int row_off, int demod, int cin,
                                 int F, int corr_elems, void* stream) {
  if (corr_elems < 1) return DGV2_EINVAL;
  if (!gW || !gs || !corr || !G || !W || !s || !stats || !dsave || !ema_var ||
      !geom_ok(B, O, I, Otot, row_off, cin, F))
    return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  ModGeom g{B, O, I, Otot, row_off, demod, cin, (F > 0 && shift) ? F : 0, 1.f / sqrtf((float)I)};
  hipError_t e;
  if (gs == gW + (size_t)O * I && corr == gs + (size_t)B * I) {   // one allocation [gW | gs | corr]: one clear
    e = hipMemsetAsync(gW, 0, sizeof(float) * ((size_t)O * I + (size_t)B * I + corr_elems), st);
  } else {
    e = hipMemsetAsync(corr, 0, sizeof(float) * corr_elems, st);
    if (e == hipSuccess) e = hipMemsetAsync(gW, 0, sizeof(float) * (size_t)O * I, st);
    if (e == hipSuccess) e = hipMemsetAsync(gs, 0, sizeof(float) * (size_t)B * I, st);
  }
  if (e != hipSuccess) return (int)e;
  // group size: as large as keeps >= ~1024 blocks in flight
  const int64_t pairs = (int64_t)O * B;
  int nblk;
  if (pairs >= 16 * 1024) {
    dim3 grid((O + 3) / 4, (B + 3) / 4);
    nblk = grid.x * grid.y;
    mod_prep_bwd_kernel<4, 4><<<grid, PB, 0, st>>>(gW, gs, corr, G, W, s, stats, dsave, ema_var, shift, fw, g,
                                                   corr_elems >= nblk ? nblk : 1);
  } else if (pairs >= 4 * 1024) {
    dim3 grid((O + 1) / 2, (B + 1) / 2);
    nblk = grid.x * grid.y;
    mod_prep_bwd_kernel<2, 2><<<grid, PB, 0, st>>>(gW, gs, corr, G, W, s, stats, dsave, ema_var, shift, fw, g,
                                                   corr_elems >= nblk ? nblk : 1);
  } else {
    dim3 grid(O, B);
    nblk = O * B;
    mod_prep_bwd_kernel<1, 1><<<grid, PB, 0, st>>>(gW, gs, corr, G, W, s, stats, dsave, ema_var, shift, fw, g,
                                                   corr_elems >= nblk ? nblk : 1);
  }
  const int ncorr = corr_elems >= nblk ? nblk : 1;
  if (demod) mod_prep_bwd_s_fix_kernel<<<B, PB, 0, st>>>(gs, s, stats, I);
  mod_prep_bwd_w_fix_kernel<<<grid_for((int64_t)O * I, 256, 256), 256, 0, st>>>(gW, W, stats, corr, ncorr, O * I,
                                                                              demod, g.scale);
  DGV2_RETURN_LAST();
}
