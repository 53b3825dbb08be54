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
  const float* rot;           // optional table [B, 256] of (sin, cos)(shift[b] * fw[f]) (batched kernels), else nullptr
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

// NT threads own one row of <= 1024 columns (column i = lane + j*NT): NT = 256 = the whole block in the per-layer
// kernels, NT = 64 = one wave in the batched kernels (four rows per block, reductions by shuffles, no barriers)
template <int NT>
__device__ __forceinline__ float row_sum(float v, float* red) {
  if constexpr (NT == 64) return wave_sum(v);
  else return block_sum(v, red);
}

// (sin, cos) of the azimuth rotation of frequency f for sample b: read from the layer's table when the launch
// prepared one (batched kernels: g.rot), evaluated otherwise -- and then only for the PE columns (`need`).
__device__ __forceinline__ void rot_factors(const ModGeom& g, const float* __restrict__ shift,
                                            const float* __restrict__ fw, int b, int f, bool need, float& sd,
                                            float& cd) {
  sd = 0.f;
  cd = 1.f;
  f = min(max(f, 0), 255);
  if (g.rot) {
    const float2 sc = reinterpret_cast<const float2*>(g.rot)[b * 256 + f];
    sd = sc.x;
    cd = sc.y;
  } else if (need) {
    sincosf(shift[b] * fw[f], &sd, &cd);
  }
}

// pre-rotation modulated weight m[j] (j-th owned column i = tid + j*PB) and t
template <int NT>
__device__ __forceinline__ void mod_row(const ModGeom& g, const float* __restrict__ W, const float* __restrict__ s,
                                        const float* __restrict__ stats, int b, int o, float (&m)[(1024 / NT)],
                                        float (&t)[(1024 / NT)], float (&wp)[(1024 / NT)]) {
  const float inv_wmax = g.demod ? 1.f / stats[0] : g.scale;
  const float inv_smax = g.demod ? 1.f / stats[2 + 2 * b] : 1.f;
#pragma unroll
  for (int j = 0; j < (1024 / NT); ++j) {
    // straight-line code (clamped address + select): conditional blocks around register-array updates made the
    // compiler shuffle the whole arrays at every merge point (2000+ moves, 240 VGPRs in the wave-per-row kernels)
    const int i = (threadIdx.x & (NT - 1)) + j * NT;
    const int ic = min(i, g.I - 1);
    const bool in = i < g.I;
    wp[j] = in ? W[(int64_t)o * g.I + ic] * inv_wmax : 0.f;
    t[j] = in ? s[(int64_t)b * g.I + ic] * inv_smax + 1.f : 0.f;
    m[j] = wp[j] * t[j];
  }
}

// rotation pair bookkeeping: column i = tid + j*PB is a "sin" column when cin <= i < cin+F, its "cos"
// partner is i + F = column j+1 of the same thread (F == PB).
template <int NT>
__device__ __forceinline__ bool is_sin_col(const ModGeom& g, int j) {
  const int i = (threadIdx.x & (NT - 1)) + j * NT;
  return g.F > 0 && i >= g.cin && i < g.cin + g.F;
}

// one (o, b) row of the prepared weights; ema_var == nullptr: c = 1 (the caller applies the input-magnitude factor
// in the GEMM epilogue instead, see dgv2_mod_prep_all_fwd)
template <int NT, typename TO>
__device__ __forceinline__ void prep_fwd_row(TO* __restrict__ wb, float* __restrict__ dsave,
                                             const float* __restrict__ W, const float* __restrict__ s,
                                             const float* __restrict__ stats, const float* __restrict__ ema_var,
                                             const float* __restrict__ shift, const float* __restrict__ fw,
                                             const ModGeom& g, int o, int b, float* red) {
  float m[(1024 / NT)], t[(1024 / NT)], wp[(1024 / NT)];
  mod_row<NT>(g, W, s, stats, b, o, m, t, wp);
  float d = 1.f;
  if (g.demod) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < (1024 / NT); ++j) ss += m[j] * m[j];
    ss = row_sum<NT>(ss, red);
    d = rsqrtf(ss + 1e-8f);
  }
  const float c = ema_var ? 1.f / (sqrtf(ema_var[0]) + 1e-8f) : 1.f;
  if ((threadIdx.x & (NT - 1)) == 0) dsave[(int64_t)b * g.O + o] = d;
  const float k = d * c;
  if (g.F > 0 && shift) {
#pragma unroll
    for (int j = 0; j + 256 / NT < (1024 / NT); ++j)
    {
      const bool sc = is_sin_col<NT>(g, j);
      const int f = (threadIdx.x & (NT - 1)) + j * NT - g.cin;
      float sd, cd;
      rot_factors(g, shift, fw, b, f, sc, sd, cd);
      const float ws = m[j], wc = m[j + 256 / NT];
      m[j] = sc ? ws * cd - wc * sd : ws;
      m[j + 256 / NT] = sc ? ws * sd + wc * cd : wc;
    }
  }
  TO* out = wb + ((int64_t)b * g.Otot + g.row_off + o) * g.I;
#pragma unroll
  for (int j = 0; j < (1024 / NT); ++j) {
    const int i = (threadIdx.x & (NT - 1)) + j * NT;
    if (i < g.I) out[i] = from_f32<TO>(m[j] * k);
  }
}

template <typename TO>
__global__ __launch_bounds__(PB) void mod_prep_fwd_kernel(TO* __restrict__ wb, float* __restrict__ dsave,
                                                          const float* __restrict__ W, const float* __restrict__ s,
                                                          const float* __restrict__ stats,
                                                          const float* __restrict__ ema_var,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ fw, ModGeom g) {
  __shared__ float red[4];
  prep_fwd_row<PB, TO>(wb, dsave, W, s, stats, ema_var, shift, fw, g, blockIdx.x, blockIdx.y, red);
}

// gradient of the loss w.r.t. the pre-rotation modulated weight m (per owned column), given G = dL/dwb
template <int NT>
__device__ __forceinline__ void grad_m(const ModGeom& g, const float* __restrict__ G, const float* __restrict__ shift,
                                       const float* __restrict__ fw, int b, int o, const float (&m)[(1024 / NT)], float d,
                                       float c, float* red, float (&gm)[(1024 / NT)]) {
  const float* Gr = G + ((int64_t)b * g.Otot + g.row_off + o) * g.I;
  float gp[(1024 / NT)];
#pragma unroll
  for (int j = 0; j < (1024 / NT); ++j) {
    const int i = (threadIdx.x & (NT - 1)) + j * NT;
    const float v = Gr[min(i, g.I - 1)];
    gp[j] = i < g.I ? v : 0.f;
  }
  if (g.F > 0 && shift) {  // transpose of the rotation
#pragma unroll
    for (int j = 0; j + 256 / NT < (1024 / NT); ++j)
    {
      const bool sc = is_sin_col<NT>(g, j);
      const int f = (threadIdx.x & (NT - 1)) + j * NT - g.cin;
      float sd, cd;
      rot_factors(g, shift, fw, b, f, sc, sd, cd);
      const float gs = gp[j], gc = gp[j + 256 / NT];
      gp[j] = sc ? gs * cd + gc * sd : gs;
      gp[j + 256 / NT] = sc ? -gs * sd + gc * cd : gc;
    }
  }
  if (g.demod) {
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < (1024 / NT); ++j) r += gp[j] * m[j];
    r = row_sum<NT>(r, red);
    const float k = c * d, d2r = d * d * r;
#pragma unroll
    for (int j = 0; j < (1024 / NT); ++j) gm[j] = k * (gp[j] - m[j] * d2r);
  } else {
#pragma unroll
    for (int j = 0; j < (1024 / NT); ++j) gm[j] = c * gp[j];
  }
}

// A block owns an OG x BG group of (o, b) pairs: gm per pair, then
//   gWraw[o,i] += sum_b gm t_b[i],   gt[b,i] += sum_o gm w'[o,i],   corr += sum gm t w'
// with the sums over the group kept in registers, so the fp32 atomics that combine groups are OG (gt) and
// BG (gWraw) times fewer than one per pair -- same-address float atomics are what bounds this kernel.
template <int NT, int OG, int BG>
__device__ __forceinline__ void prep_bwd_group(float* __restrict__ gWraw, float* __restrict__ gt,
                                               float* __restrict__ corr, const float* __restrict__ G,
                                               const float* __restrict__ W, const float* __restrict__ s,
                                               const float* __restrict__ stats, const float* __restrict__ dsave,
                                               const float* __restrict__ ema_var, const float* __restrict__ shift,
                                               const float* __restrict__ fw, const ModGeom& g, int corr_slots,
                                               int o0, int b0, int blk, float* red) {
  const float c = ema_var ? 1.f / (sqrtf(ema_var[0]) + 1e-8f) : 1.f;   // nullptr: G is already dL/d(m d)
  float gwacc[OG][(1024 / NT)];
#pragma unroll
  for (int ol = 0; ol < OG; ++ol)
#pragma unroll
    for (int j = 0; j < (1024 / NT); ++j) gwacc[ol][j] = 0.f;
  float part = 0.f;
#pragma unroll
  for (int bl = 0; bl < BG; ++bl) {
    const int b = b0 + bl;
    if (b >= g.B) break;
    float gtacc[(1024 / NT)];
#pragma unroll
    for (int j = 0; j < (1024 / NT); ++j) gtacc[j] = 0.f;
#pragma unroll
    for (int ol = 0; ol < OG; ++ol) {
      const int o = o0 + ol;
      if (o >= g.O) break;
      float m[(1024 / NT)], t[(1024 / NT)], wp[(1024 / NT)], gm[(1024 / NT)];
      mod_row<NT>(g, W, s, stats, b, o, m, t, wp);
      grad_m<NT>(g, G, shift, fw, b, o, m, dsave[(int64_t)b * g.O + o], c, red, gm);
#pragma unroll
      for (int j = 0; j < (1024 / NT); ++j) {
        const float gw = gm[j] * t[j];
        gwacc[ol][j] += gw;
        gtacc[j] += gm[j] * wp[j];
        part += gw * wp[j];
      }
    }
#pragma unroll
    for (int j = 0; j < (1024 / NT); ++j) {
      const int i = (threadIdx.x & (NT - 1)) + j * NT;
      if (i < g.I) atomicAdd(&gt[(int64_t)b * g.I + i], gtacc[j]);
    }
  }
#pragma unroll
  for (int ol = 0; ol < OG; ++ol) {
    const int o = o0 + ol;
    if (o >= g.O) break;
#pragma unroll
    for (int j = 0; j < (1024 / NT); ++j) {
      const int i = (threadIdx.x & (NT - 1)) + j * NT;
      if (i < g.I) atomicAdd(&gWraw[(int64_t)o * g.I + i], gwacc[ol][j]);
    }
  }
  if (g.demod) {
    part = row_sum<NT>(part, red);
    // one slot per block when the caller provided them (plain store: the buffer was cleared, nobody else writes
    // the slot); 1000+ same-address atomics would otherwise serialise into the longest part of this kernel
    if ((threadIdx.x & (NT - 1)) == 0) {
      if (corr_slots > 1) corr[blk] = part;
      else atomicAdd(corr, part);
    }
  }
}

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
  prep_bwd_group<PB, OG, BG>(gWraw, gt, corr, G, W, s, stats, dsave, ema_var, shift, fw, g, corr_slots, blockIdx.x * OG,
                         blockIdx.y * BG, blockIdx.y * gridDim.x + blockIdx.x, red);
}

// s' = s / smax:  gs_i = gt_i / smax - [|s_i| == smax] sign(s_i) (sum_j gt_j s_j) / smax^2   (in place on gt)
template <int NT>
__device__ __forceinline__ void s_fix_row(float* __restrict__ gs, const float* __restrict__ s,
                                          const float* __restrict__ stats, int I, int b, float* red) {
  const float smax = stats[2 + 2 * b];
  float acc[(1024 / NT)], sv[(1024 / NT)], dot = 0.f;
#pragma unroll
  for (int j = 0; j < (1024 / NT); ++j) {
    const int i = (threadIdx.x & (NT - 1)) + j * NT;
    const int ic = min(i, I - 1);
    const float av = gs[(int64_t)b * I + ic], bv = s[(int64_t)b * I + ic];
    acc[j] = i < I ? av : 0.f;
    sv[j] = i < I ? bv : 0.f;
    dot += acc[j] * sv[j];
  }
  dot = row_sum<NT>(dot, red);
#pragma unroll
  for (int j = 0; j < (1024 / NT); ++j) {
    const int i = (threadIdx.x & (NT - 1)) + j * NT;
    if (i < I) {
      float v = acc[j] / smax;
      if (fabsf(sv[j]) == smax) v -= (sv[j] > 0.f ? 1.f : -1.f) * dot / (smax * smax);
      gs[(int64_t)b * I + i] = v;
    }
  }
}

__global__ __launch_bounds__(PB) void mod_prep_bwd_s_fix_kernel(float* __restrict__ gs, const float* __restrict__ s,
                                                                const float* __restrict__ stats, int I) {
  __shared__ float red[4];
  s_fix_row<PB>(gs, s, stats, I, blockIdx.x, red);
}

// W' = W * k (k = 1/wmax or scale): gW = gWraw * k, and with demod the max-norm term
//   gW_i -= [|W_i| == wmax] sign(W_i) corr / wmax,  corr = sum gWraw w'.
__device__ __forceinline__ void w_fix_part(float* __restrict__ gW, const float* __restrict__ W,
                                           const float* __restrict__ stats, const float* __restrict__ corr,
                                           int ncorr, int OI, int demod, float scale, int blk, int nblk, float* red) {
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
  for (int i = blk * 256 + threadIdx.x; i < OI; i += nblk * 256) {
    float v = gW[i] * k;
    if (demod) {
      const float w = W[i];
      if (fabsf(w) == wmax) v -= (w > 0.f ? 1.f : -1.f) * csum / wmax;
    }
    gW[i] = v;
  }
}

__global__ __launch_bounds__(256) void mod_prep_bwd_w_fix_kernel(float* __restrict__ gW, const float* __restrict__ W,
                                                                 const float* __restrict__ stats,
                                                                 const float* __restrict__ corr, int ncorr, int OI,
                                                                 int demod, float scale) {
  __shared__ float red[4];
  w_fix_part(gW, W, stats, corr, ncorr, OI, demod, scale, blockIdx.x, gridDim.x, red);
}

// ------------------------------------------------------------------------------------------------
// Batched forms: ALL modulated layers of one generator pass in one launch each (19 on dusty_v2).  The per-layer
// operands travel by value in the kernel arguments; a block finds its layer in a prefix table of block counts.
// The input-magnitude factor c = 1/(sqrt(ema_var)+1e-8) is NOT folded into these weights (it depends on the
// activation statistics of the running pass, which would serialise the preparation layer by layer): the GEMM
// epilogues apply it per output channel (row_scale), and the backward takes G = dL/d(m d) directly.
// ------------------------------------------------------------------------------------------------
constexpr int MPA_MAX = 32;

struct PrepAll {
  const float* W[MPA_MAX];
  const float* s[MPA_MAX];
  const float* fw[MPA_MAX];
  void* wb[MPA_MAX];           // fwd: prepared weights;  bwd: G = dL/dwb (fp32)
  float* dsave[MPA_MAX];
  float* gW[MPA_MAX];          // bwd: [gW | gs | corr] of the layer, contiguous
  int O[MPA_MAX], I[MPA_MAX], Otot[MPA_MAX], row_off[MPA_MAX], cin[MPA_MAX], flags[MPA_MAX];  // 1 demod, 2 rotate, 4 bf16
  int blk_end[MPA_MAX];        // prefix ends of this launch's blocks per layer
  int aux[MPA_MAX];            // stats: W blocks;  bwd groups: corr slots;  w_fix: blocks of the layer
  int ncorr[MPA_MAX];
  float* stats;                // [L, 2 + 2B]
  float* rot;                  // [L, B, 256, 2] (sin, cos) of shift[b] * fw[l][f] for the rotating layers
  const float* shift;
  int B, L;
};

// The by-value argument block, read where it lies (the kernarg segment) instead of through the private copy clang
// makes of a dynamically indexed by-value struct (which cost these kernels ~200 VGPRs): uniform indices then
// become scalar loads.
__device__ __forceinline__ const PrepAll& kernarg_view(const PrepAll&) {
  return *(const PrepAll*)__builtin_amdgcn_kernarg_segment_ptr();
}

__device__ __forceinline__ int find_layer(const PrepAll& a, int bid, int& local) {
  // number of prefix ends <= bid, all 32 entries compared at once (entries past L are INT_MAX): a dependent
  // while-loop over the table cost one scalar-load latency per layer, microseconds per block
  int l = 0;
#pragma unroll
  for (int k = 0; k < MPA_MAX; ++k) l += bid >= a.blk_end[k] ? 1 : 0;
  l = min(l, a.L - 1);
  local = bid - (l ? a.blk_end[l - 1] : 0);
  return l;
}

__device__ __forceinline__ ModGeom layer_geom(const PrepAll& a, int l) {
  const bool rotate = (a.flags[l] & 2) && a.shift;
  return ModGeom{a.B, a.O[l], a.I[l], a.Otot[l], a.row_off[l], a.flags[l] & 1, a.cin[l], rotate ? PB : 0,
                 rsqrtf((float)a.I[l]), rotate ? a.rot + (size_t)l * a.B * PB * 2 : nullptr};
}

__global__ __launch_bounds__(PB) void mod_stats_all_kernel(PrepAll a_by_value) {
  const PrepAll& a = kernarg_view(a_by_value);
  __shared__ float red[4];
  int local;
  const int l = find_layer(a, blockIdx.x, local);
  float* stats = a.stats + (size_t)l * (2 + 2 * a.B);
  const int I = a.I[l], OI = a.O[l] * I, wblocks = a.aux[l];
  const float* W = a.W[l];
  const float* s = a.s[l];
  const int nstat = (a.flags[l] & 1) ? wblocks + a.B : 0;
  if (local >= nstat) {   // rotation table of this layer: one block per sample, one thread per frequency
    const int b = local - nstat;
    float sd, cd;
    sincosf(a.shift[b] * a.fw[l][threadIdx.x], &sd, &cd);
    reinterpret_cast<float2*>(a.rot)[((size_t)l * a.B + b) * PB + threadIdx.x] = make_float2(sd, cd);
    return;
  }
  float m = 0.f;
  int slot;
  if (local < wblocks) {
    for (int i = local * PB + threadIdx.x; i < OI; i += wblocks * PB) m = fmaxf(m, fabsf(W[i]));
    slot = 0;
  } else {
    const int b = local - wblocks;
    for (int i = threadIdx.x; i < I; i += PB) m = fmaxf(m, fabsf(s[(int64_t)b * I + i]));
    slot = 2 + 2 * b;
  }
  m = block_max(m, red);
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(stats + slot), __float_as_uint(m));
}

__global__ __launch_bounds__(PB) void mod_prep_all_fwd_kernel(PrepAll a_by_value) {
  const PrepAll& a = kernarg_view(a_by_value);
  __shared__ float red[4];
  int local;
  const int l = find_layer(a, blockIdx.x, local);
  const ModGeom g = layer_geom(a, l);
  const int row = local * 4 + (threadIdx.x >> 6);   // one wave per (o, b) row
  if (row >= g.O * g.B) return;
  const int o = row % g.O, b = row / g.O;
  const float* stats = a.stats + (size_t)l * (2 + 2 * a.B);
  if (a.flags[l] & 4)
    prep_fwd_row<64, bf16_t>((bf16_t*)a.wb[l], a.dsave[l], a.W[l], a.s[l], stats, nullptr, a.shift, a.fw[l], g, o, b, red);
  else
    prep_fwd_row<64, float>((float*)a.wb[l], a.dsave[l], a.W[l], a.s[l], stats, nullptr, a.shift, a.fw[l], g, o, b, red);
}

template <int OG, int BG>
__global__ __launch_bounds__(PB) void mod_prep_all_bwd_kernel(PrepAll a_by_value) {
  const PrepAll& a = kernarg_view(a_by_value);
  __shared__ float red[4];
  int local;
  const int l = find_layer(a, blockIdx.x, local);
  const ModGeom g = layer_geom(a, l);
  // one BLOCK per OG x BG group (a wave per group would hold 16 columns x 4 rows of partial sums per lane and
  // drops to one wave per SIMD); the rotation factors are evaluated (3 per thread), not read from the table
  const int gx = (g.O + OG - 1) / OG;
  float* gW = a.gW[l];
  float* gs = gW + (size_t)g.O * g.I;
  float* corr = gs + (size_t)g.B * g.I;
  prep_bwd_group<PB, OG, BG>(gW, gs, corr, (const float*)a.wb[l], a.W[l], a.s[l], a.stats + (size_t)l * (2 + 2 * a.B),
                             a.dsave[l], nullptr, a.shift, a.fw[l], g, a.aux[l], (local % gx) * OG, (local / gx) * BG,
                             local, red);
}

__global__ __launch_bounds__(PB) void mod_prep_all_s_fix_kernel(PrepAll a_by_value) {
  const PrepAll& a = kernarg_view(a_by_value);
  __shared__ float red[4];
  int local;
  const int l = find_layer(a, blockIdx.x, local);
  float* gs = a.gW[l] + (size_t)a.O[l] * a.I[l];
  const int b = local * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  s_fix_row<64>(gs, a.s[l], a.stats + (size_t)l * (2 + 2 * a.B), a.I[l], b, red);
}

__global__ __launch_bounds__(256) void mod_prep_all_w_fix_kernel(PrepAll a_by_value) {
  const PrepAll& a = kernarg_view(a_by_value);
  __shared__ float red[4];
  int local;
  const int l = find_layer(a, blockIdx.x, local);
  const int O = a.O[l], I = a.I[l];
  float* gW = a.gW[l];
  const float* corr = gW + (size_t)O * I + (size_t)a.B * I;
  w_fix_part(gW, a.W[l], a.stats + (size_t)l * (2 + 2 * a.B), corr, a.ncorr[l], O * I, a.flags[l] & 1,
             rsqrtf((float)I), local, a.aux[l], red);
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
                                         float inv_count, float weight, int update, float* cvec, int ncvec) {
  float s = 0.f;   // one wave: fold the nsum partial sums (up to 8192 from the producing kernels), 4 loads in flight
  if (sumsq) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = threadIdx.x;
    for (; k + 192 < nsum; k += 256) {
      s0 += sumsq[k];
      s1 += sumsq[k + 64];
      s2 += sumsq[k + 128];
      s3 += sumsq[k + 192];
    }
    for (; k < nsum; k += 64) s0 += sumsq[k];
    s = (s0 + s1) + (s2 + s3);
  }
  s = wave_sum(s);
  float v = ema[0];   // every lane computes the same value; lane 0 stores it
  if (update) v += weight * ((s + add) * inv_count - v);
  if (threadIdx.x == 0) {
    if (update) ema[0] = v;
    if (snapshot) snapshot[0] = v;
  }
  const float c = 1.f / (sqrtf(v) + 1e-8f);   // the factor the modulated conv applies to its output rows
  for (int i = threadIdx.x; i < ncvec; i += 64) cvec[i] = c;
}

extern "C" int dgv2_ema_scalar(float* ema, float* snapshot, const float* sumsq, int nsum, float add, float inv_count,
                               float weight, int update, float* cvec, int ncvec, void* stream) {
  if (!ema || (!snapshot && !cvec) || nsum < 0 || ncvec < 0 || (ncvec > 0 && !cvec)) return DGV2_EINVAL;
  ema_scalar_kernel<<<1, 64, 0, (hipStream_t)stream>>>(ema, snapshot, sumsq, nsum, add, inv_count, weight, update, cvec,
                                                       ncvec);
  DGV2_RETURN_LAST();
}

// The same update for a GROUP of layers that share their input (the output heads of a level, dusty_v2.py:32-57: one
// ModConv2d per output, each with its own ema_var buffer): block i updates emas[i] and fills rows[i] entries of cvec
// behind those of the blocks before it.  One launch instead of one per head.
struct EmaGroup {
  float* ema[8];
  int rows[8];
};
static __global__ void ema_scalar_group_kernel(EmaGroup grp, const float* sumsq, int nsum, float add, float inv_count,
                                               float weight, int update, float* cvec) {
  float s = 0.f;
  if (sumsq) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // the summation order of ema_scalar_kernel: same bits
    int k = threadIdx.x;
    for (; k + 192 < nsum; k += 256) {
      s0 += sumsq[k];
      s1 += sumsq[k + 64];
      s2 += sumsq[k + 128];
      s3 += sumsq[k + 192];
    }
    for (; k < nsum; k += 64) s0 += sumsq[k];
    s = (s0 + s1) + (s2 + s3);
  }
  s = wave_sum(s);
  const int i = blockIdx.x;
  float v = grp.ema[i][0];
  if (update) v += weight * ((s + add) * inv_count - v);
  if (threadIdx.x == 0 && update) grp.ema[i][0] = v;
  int off = 0;
  for (int j = 0; j < i; ++j) off += grp.rows[j];
  const float c = 1.f / (sqrtf(v) + 1e-8f);
  for (int r = threadIdx.x; r < grp.rows[i]; r += 64) cvec[off + r] = c;
}

extern "C" int dgv2_ema_scalar_group(float* const* emas, const int* rows, int n, const float* sumsq, int nsum, float add,
                                     float inv_count, float weight, int update, float* cvec, void* stream) {
  if (!emas || !rows || n < 1 || n > 8 || nsum < 0 || !cvec) return DGV2_EINVAL;
  EmaGroup grp;
  for (int i = 0; i < 8; ++i) {
    grp.ema[i] = i < n ? emas[i] : nullptr;
    grp.rows[i] = i < n ? rows[i] : 0;
    if (i < n && (!emas[i] || rows[i] < 0)) return DGV2_EINVAL;
  }
  ema_scalar_group_kernel<<<n, 64, 0, (hipStream_t)stream>>>(grp, sumsq, nsum, add, inv_count, weight, update, cvec);
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

// ------------------------------------------------------------------------------------------------
// Batched preparation of L <= 32 modulated layers (see the kernels above).  All pointer / int arguments are HOST
// arrays of length L; wb[l] is the layer's GEMM operand buffer [B, Otot[l], I[l]] (layers that share a GEMM pass
// the same buffer with different row_off), in bf16 when flags[l] & 4 else fp32; dsave[l] fp32 [B, O[l]];
// stats fp32 [L, 2 + 2B] (filled here, needed by the backward); fw[l] fp32 [256] azimuth frequencies of layers
// with flags[l] & 2 (rotation of the PE columns [cin, cin + 512) by shift[b], shift NULL = no rotation).
// The prepared weights do NOT contain 1/(sqrt(ema_var)+1e-8): pass that factor as row_scale to the GEMM.
// ------------------------------------------------------------------------------------------------
static int fill_common(PrepAll& a, const float* const* W, const float* const* s, const float* const* fw, const int* O,
                       const int* I, const int* Otot, const int* row_off, const int* cin, const int* flags, int B, int L) {
  if (!W || !s || !fw || !O || !I || !Otot || !row_off || !cin || !flags || B < 1 || L < 1 || L > MPA_MAX) return DGV2_EINVAL;
  for (int l = 0; l < L; ++l) {
    const int F = (flags[l] & 2) ? PB : 0;
    if (!W[l] || !s[l] || (F && !fw[l]) || !geom_ok(B, O[l], I[l], Otot[l], row_off[l], cin[l], F)) return DGV2_EINVAL;
    if ((int64_t)O[l] * B >= (1 << 24)) return DGV2_EINVAL;
    a.W[l] = W[l]; a.s[l] = s[l]; a.fw[l] = fw[l];
    a.O[l] = O[l]; a.I[l] = I[l]; a.Otot[l] = Otot[l]; a.row_off[l] = row_off[l]; a.cin[l] = cin[l]; a.flags[l] = flags[l];
    a.aux[l] = 0; a.ncorr[l] = 0; a.gW[l] = nullptr; a.dsave[l] = nullptr; a.wb[l] = nullptr;
  }
  for (int l = 0; l < MPA_MAX; ++l) a.blk_end[l] = 0x7fffffff;   // launches overwrite [0, L)
  a.B = B; a.L = L;
  return 0;
}

extern "C" int dgv2_mod_prep_all_fwd(void* const* wb, float* const* dsave, float* stats, float* rot, const float* const* W,
                                     const float* const* s, const float* const* fw, const int* O, const int* I,
                                     const int* Otot, const int* row_off, const int* cin, const int* flags,
                                     const float* shift, int B, int L, void* stream) {
  PrepAll a;
  if (!wb || !dsave || !stats || !rot) return DGV2_EINVAL;
  int rc = fill_common(a, W, s, fw, O, I, Otot, row_off, cin, flags, B, L);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  a.stats = stats; a.shift = shift; a.rot = rot;
  int nstat = 0;
  for (int l = 0; l < L; ++l) {
    if (!wb[l] || !dsave[l]) return DGV2_EINVAL;
    a.wb[l] = wb[l]; a.dsave[l] = dsave[l];
    if (flags[l] & 1) {
      a.aux[l] = (O[l] * I[l] + PB * 8 - 1) / (PB * 8);
      nstat += a.aux[l] + B;
    }
    if ((flags[l] & 2) && shift) nstat += B;   // rotation-table blocks
    a.blk_end[l] = nstat;
  }
  if (nstat > 0) {
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(float) * (size_t)L * (2 + 2 * B), st);
    if (e != hipSuccess) return (int)e;
    mod_stats_all_kernel<<<nstat, PB, 0, st>>>(a);
  }
  int nblk = 0;
  for (int l = 0; l < L; ++l) {
    nblk += (O[l] * B + 3) / 4;   // one wave per row
    a.blk_end[l] = nblk;
  }
  mod_prep_all_fwd_kernel<<<nblk, PB, 0, st>>>(a);
  DGV2_RETURN_LAST();
}

// Backward of the above.  G[l] fp32 [B, Otot[l], I[l]] = dL/d(prepared weights) (layers sharing a buffer pass the
// same pointer); out[l] = one fp32 allocation [gW (O*I) | gs (B*I) | corr (ncorr[l] >= 1 scratch)], all inside
// `flat` [flat_elems], which is cleared here with ONE launch.  stats / dsave: as left by the forward.
extern "C" int dgv2_mod_prep_all_bwd(float* flat, int64_t flat_elems, float* const* out, const int* ncorr,
                                     const float* const* G, const float* const* W, const float* const* s,
                                     const float* stats, const float* rot, float* const* dsave,
                                     const float* const* fw, const int* O, const int* I, const int* Otot,
                                     const int* row_off, const int* cin, const int* flags, const float* shift, int B,
                                     int L, void* stream) {
  PrepAll a;
  if (!flat || flat_elems < 1 || !out || !ncorr || !G || !stats || !dsave || !rot) return DGV2_EINVAL;
  int rc = fill_common(a, W, s, fw, O, I, Otot, row_off, cin, flags, B, L);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  a.stats = const_cast<float*>(stats); a.shift = shift; a.rot = const_cast<float*>(rot);
  int cls[MPA_MAX], nb[MPA_MAX];
  for (int l = 0; l < L; ++l) {
    if (!out[l] || !G[l] || !dsave[l] || ncorr[l] < 1) return DGV2_EINVAL;
    if (out[l] < flat || out[l] + (size_t)O[l] * I[l] + (size_t)B * I[l] + ncorr[l] > flat + flat_elems) return DGV2_EINVAL;
    a.gW[l] = out[l]; a.wb[l] = const_cast<float*>(G[l]); a.dsave[l] = dsave[l];
    // group size: as large as keeps >= ~1024 blocks in flight (same rule as dgv2_mod_prep_bwd)
    const int64_t pairs = (int64_t)O[l] * B;
    cls[l] = pairs >= 16 * 1024 ? 4 : (pairs >= 4 * 1024 ? 2 : 1);
    nb[l] = ((O[l] + cls[l] - 1) / cls[l]) * ((B + cls[l] - 1) / cls[l]);
    a.ncorr[l] = ncorr[l] >= nb[l] ? nb[l] : 1;
  }
  hipError_t e = hipMemsetAsync(flat, 0, sizeof(float) * (size_t)flat_elems, st);
  if (e != hipSuccess) return (int)e;
  for (int c = 4; c >= 1; c >>= 1) {
    int n = 0;
    for (int l = 0; l < L; ++l) {
      if (cls[l] == c) n += nb[l];
      a.blk_end[l] = n;
      a.aux[l] = a.ncorr[l];
    }
    if (n == 0) continue;
    if (c == 4) mod_prep_all_bwd_kernel<4, 4><<<n, PB, 0, st>>>(a);
    else if (c == 2) mod_prep_all_bwd_kernel<2, 2><<<n, PB, 0, st>>>(a);
    else mod_prep_all_bwd_kernel<1, 1><<<n, PB, 0, st>>>(a);
  }
  int n = 0;
  for (int l = 0; l < L; ++l) {
    if (flags[l] & 1) n += (B + 3) / 4;
    a.blk_end[l] = n;
  }
  if (n > 0) mod_prep_all_s_fix_kernel<<<n, PB, 0, st>>>(a);
  n = 0;
  for (int l = 0; l < L; ++l) {
    a.aux[l] = grid_for((int64_t)O[l] * I[l], 256, 64);
    n += a.aux[l];
    a.blk_end[l] = n;
  }
  mod_prep_all_w_fix_kernel<<<n, 256, 0, st>>>(a);
  DGV2_RETURN_LAST();
}
