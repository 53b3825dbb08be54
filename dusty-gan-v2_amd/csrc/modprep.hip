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

// ------------------------------------------------------------------------------------------------
// V4 forms of the batched kernels (round 5).  The wave-per-row kernels above walk a row with column i = lane + 64 j,
// j < 16 -- sixteen 4-byte loads per operand and sixteen 2-byte stores per lane whatever I is (clamped addresses), and
// the backward combines its OG x BG register tiles with fp32 atomics (0.5 per element: what bound it).  Here a lane owns
// 16-byte CHUNKS: chunk q = lane + 64 jj covers columns [4q, 4q + 4), jj < JJ = ceil(I / 256) (a template parameter),
// so a row is JJ float4 loads per operand and JJ 8-byte stores, and the rotation partner of a sin column (i + 256) is
// chunk q + 64 = the SAME lane's next chunk.  The backward keeps the sums over its samples in registers and leaves
// PARTIAL results per (row group, sample chunk) in scratch with plain stores; the two fix-up kernels, which read every
// gW / gs element anyway, fold the partials: no atomics, no zero fill, run-to-run identical bits.
// Conditions (host-checked, else the kernels above run): I % 4 == 0, rotating layers cin % 4 == 0.
// ------------------------------------------------------------------------------------------------
constexpr int V4_BC = 8;   // samples per unit (forward: consecutive rows share the W row; backward: register sums)
constexpr int V4_OG = 4;   // rows per backward unit

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator*(const float4& a, const float4& b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator+(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(const float4& a, float b) { return make_float4(a.x * b, a.y * b, a.z * b, a.w * b); }
__device__ __forceinline__ float hsum(const float4& a) { return (a.x + a.y) + (a.z + a.w); }

template <typename TO> __device__ __forceinline__ void st4(TO* p, const float4& v);
template <> __device__ __forceinline__ void st4<float>(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float4& v) {
  union { uint2 u; bf16_t e[4]; } q;
  q.e[0] = (bf16_t)v.x; q.e[1] = (bf16_t)v.y; q.e[2] = (bf16_t)v.z; q.e[3] = (bf16_t)v.w;
  *reinterpret_cast<uint2*>(p) = q.u;
}

// (sin, cos) of the four frequencies of sin chunk q of a rotating layer (table [B, 256] of float2, mod_stats_all_kernel)
__device__ __forceinline__ void rot4(const float* __restrict__ rot_lb, int f0, float4& sd, float4& cd) {
  const float4 a = ld4(rot_lb + 2 * f0), b = ld4(rot_lb + 2 * f0 + 4);
  sd = make_float4(a.x, a.z, b.x, b.z);
  cd = make_float4(a.y, a.w, b.y, b.w);
}

template <int JJ, typename TO>
__device__ __forceinline__ void prep_fwd_v4(const PrepAll& a, int l, int o, int b0, int nb) {
  const int lane = threadIdx.x & 63;
  const int I = a.I[l], O = a.O[l], Otot = a.Otot[l], row_off = a.row_off[l];
  const bool demod = a.flags[l] & 1, rotate = (a.flags[l] & 2) && a.shift;
  const int cin4 = a.cin[l] >> 2;
  const float* stats = a.stats + (size_t)l * (2 + 2 * a.B);
  const float inv_wmax = demod ? 1.f / stats[0] : rsqrtf((float)I);
  const float* Wr = a.W[l] + (int64_t)o * I;
  float4 wp[JJ];
  bool in[JJ];
#pragma unroll
  for (int jj = 0; jj < JJ; ++jj) {
    const int q = lane + 64 * jj;
    in[jj] = 4 * q < I;
    wp[jj] = in[jj] ? ld4(Wr + 4 * q) * inv_wmax : f4(0.f);
  }
  TO* wb = (TO*)a.wb[l];
  for (int bi = 0; bi < nb; ++bi) {
    const int b = b0 + bi;
    const float inv_smax = demod ? 1.f / stats[2 + 2 * b] : 1.f;
    const float* sr = a.s[l] + (int64_t)b * I;
    float4 m[JJ];
    float ss = 0.f;
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) {
      const int q = lane + 64 * jj;
      const float4 t = in[jj] ? ld4(sr + 4 * q) * inv_smax + f4(1.f) : f4(0.f);
      m[jj] = wp[jj] * t;
      ss += hsum(m[jj] * m[jj]);
    }
    float d = 1.f;
    if (demod) d = rsqrtf(wave_sum(ss) + 1e-8f);
    if (lane == 0) a.dsave[l][(int64_t)b * O + o] = d;
    if (rotate) {
      const float* rot_lb = a.rot + ((size_t)l * a.B + b) * PB * 2;
#pragma unroll
      for (int jj = 0; jj + 1 < JJ; ++jj) {
        const int q = lane + 64 * jj;
        const bool sc = q >= cin4 && q < cin4 + 64;
        float4 sd, cd;
        rot4(rot_lb, 4 * min(max(q - cin4, 0), 63), sd, cd);
        const float4 ws = m[jj], wc = m[jj + 1];
        if (sc) {
          m[jj] = ws * cd - wc * sd;
          m[jj + 1] = ws * sd + wc * cd;
        }
      }
    }
    TO* out = wb + ((int64_t)b * Otot + row_off + o) * I;
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj)
      if (in[jj]) st4<TO>(out + 4 * (lane + 64 * jj), m[jj] * d);
  }
}

// blocks of layer l: ceil(O / 4) row groups x ceil(B / V4_BC) sample chunks; wave w of a block owns row 4 og + w
__global__ __launch_bounds__(PB) void mod_prep_all_fwd_v4_kernel(PrepAll a_by_value) {
  const PrepAll& a = kernarg_view(a_by_value);
  int local;
  const int l = find_layer(a, blockIdx.x, local);
  const int O = a.O[l], nog = (O + 3) / 4;
  const int o = (local % nog) * 4 + (threadIdx.x >> 6), b0 = (local / nog) * V4_BC;
  if (o >= O) return;
  const int nb = min(V4_BC, a.B - b0);
  const int JJ = (a.I[l] + 255) >> 8;
  if (a.flags[l] & 4) {
    switch (JJ) {
      case 1: prep_fwd_v4<1, bf16_t>(a, l, o, b0, nb); break;
      case 2: prep_fwd_v4<2, bf16_t>(a, l, o, b0, nb); break;
      case 3: prep_fwd_v4<3, bf16_t>(a, l, o, b0, nb); break;
      default: prep_fwd_v4<4, bf16_t>(a, l, o, b0, nb); break;
    }
  } else {
    switch (JJ) {
      case 1: prep_fwd_v4<1, float>(a, l, o, b0, nb); break;
      case 2: prep_fwd_v4<2, float>(a, l, o, b0, nb); break;
      case 3: prep_fwd_v4<3, float>(a, l, o, b0, nb); break;
      default: prep_fwd_v4<4, float>(a, l, o, b0, nb); break;
    }
  }
}

// ---- backward ----
struct PrepAllB {
  const float* W[MPA_MAX];
  const float* s[MPA_MAX];
  const float* G[MPA_MAX];       // dL/d(prepared weights) fp32 [B, Otot, I]
  const float* dsave[MPA_MAX];
  float* out[MPA_MAX];           // [gW (O I) | gs (B I) | ...]
  float* gtp[MPA_MAX];           // scratch: gt partials [nog][B][I]
  float* gwp[MPA_MAX];           // scratch: gW partials [nbc][O][I]
  float* corr[MPA_MAX];          // scratch: one slot per unit
  int O[MPA_MAX], I[MPA_MAX], Otot[MPA_MAX], row_off[MPA_MAX], cin[MPA_MAX], flags[MPA_MAX];
  int blk_end[MPA_MAX];
  const float* stats;
  const float* rot;
  const float* shift;
  int B, L;
};
__device__ __forceinline__ const PrepAllB& kernarg_view(const PrepAllB&) {
  return *(const PrepAllB*)__builtin_amdgcn_kernarg_segment_ptr();
}
__device__ __forceinline__ int find_layer(const PrepAllB& a, int bid, int& local) {
  int l = 0;
#pragma unroll
  for (int k = 0; k < MPA_MAX; ++k) l += bid >= a.blk_end[k] ? 1 : 0;
  l = min(l, a.L - 1);
  local = bid - (l ? a.blk_end[l - 1] : 0);
  return l;
}

// one unit: rows 4 og .. 4 og + 3 against samples b0 .. b0 + nb - 1 (see the header of this section)
template <int JJ>
__device__ __forceinline__ void prep_bwd_v4(const PrepAllB& a, int l, int og, int bc, int unit) {
  const int lane = threadIdx.x & 63;
  const int I = a.I[l], O = a.O[l], Otot = a.Otot[l], row_off = a.row_off[l];
  const bool demod = a.flags[l] & 1, rotate = (a.flags[l] & 2) && a.shift;
  const int cin4 = a.cin[l] >> 2;
  const float* stats = a.stats + (size_t)l * (2 + 2 * a.B);
  const float inv_wmax = demod ? 1.f / stats[0] : rsqrtf((float)I);
  const int o0 = og * V4_OG, b0 = bc * V4_BC, nb = min(V4_BC, a.B - b0);
  const int no = min(V4_OG, O - o0);
  float4 wp[V4_OG][JJ], gw[V4_OG][JJ];
  bool in[JJ];
#pragma unroll
  for (int jj = 0; jj < JJ; ++jj) in[jj] = 4 * (lane + 64 * jj) < I;
#pragma unroll
  for (int ol = 0; ol < V4_OG; ++ol)
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) {
      const int oc = min(o0 + ol, O - 1);
      wp[ol][jj] = (in[jj] && ol < no) ? ld4(a.W[l] + (int64_t)oc * I + 4 * (lane + 64 * jj)) * inv_wmax : f4(0.f);
      gw[ol][jj] = f4(0.f);
    }
  float part = 0.f;
  // this lane's sin chunk (one at most) of a rotating layer
  int jsin = -1;
  if (rotate) {
#pragma unroll
    for (int jj = 0; jj + 1 < JJ; ++jj) {
      const int q = lane + 64 * jj;
      if (q >= cin4 && q < cin4 + 64) jsin = jj;
    }
  }
  const int f0 = 4 * min(max(lane + 64 * max(jsin, 0) - cin4, 0), 63);
  // One step = one sample: t, the four rows' gradients (rows past O: a clamped re-read against zero weights), their
  // demodulation factors and the rotation factors.  The NEXT step's operands are requested before this step's
  // arithmetic (one wave per SIMD at this register count: nothing else would hide the memory latency), and the four
  // rows' reductions run as interleaved butterflies.
  struct Step {
    float4 t[JJ], gp[V4_OG][JJ], sd, cd;
    float dv[V4_OG];
  };
  auto load_step = [&](Step& st, int b) {
    const float inv_smax = demod ? 1.f / stats[2 + 2 * b] : 1.f;
    const float* sr = a.s[l] + (int64_t)b * I;
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) st.t[jj] = in[jj] ? ld4(sr + 4 * (lane + 64 * jj)) * inv_smax + f4(1.f) : f4(0.f);
#pragma unroll
    for (int ol = 0; ol < V4_OG; ++ol) {
      const int oc = min(o0 + ol, O - 1);
      const float* Gr = a.G[l] + ((int64_t)b * Otot + row_off + oc) * I;
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) st.gp[ol][jj] = in[jj] ? ld4(Gr + 4 * (lane + 64 * jj)) : f4(0.f);
      st.dv[ol] = demod ? a.dsave[l][(int64_t)b * O + oc] : 1.f;
    }
    st.sd = f4(0.f);
    st.cd = f4(1.f);
    if (rotate) rot4(a.rot + ((size_t)l * a.B + b) * PB * 2, f0, st.sd, st.cd);
  };
  Step cur, nxt;
  load_step(cur, b0);
  for (int bi = 0; bi < nb; ++bi) {
    const int b = b0 + bi;
    load_step(nxt, min(b + 1, b0 + nb - 1));   // (the last step re-reads itself: no branch around the loads)
    float4 gt[JJ];
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) gt[jj] = f4(0.f);
    if (rotate) {   // transpose of the rotation
#pragma unroll
      for (int ol = 0; ol < V4_OG; ++ol)
#pragma unroll
        for (int jj = 0; jj + 1 < JJ; ++jj) {
          const float4 gs = cur.gp[ol][jj], gc = cur.gp[ol][jj + 1];
          if (jj == jsin) {
            cur.gp[ol][jj] = gs * cur.cd + gc * cur.sd;
            cur.gp[ol][jj + 1] = gc * cur.cd - gs * cur.sd;
          }
        }
    }
    float r[V4_OG];
#pragma unroll
    for (int ol = 0; ol < V4_OG; ++ol) {
      r[ol] = 0.f;
      if (demod) {
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) r[ol] += hsum(cur.gp[ol][jj] * (wp[ol][jj] * cur.t[jj]));
      }
    }
    if (demod) {   // four interleaved butterflies
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1)
#pragma unroll
        for (int ol = 0; ol < V4_OG; ++ol) r[ol] += __shfl_xor(r[ol], sft, 64);
    }
#pragma unroll
    for (int ol = 0; ol < V4_OG; ++ol) {
      const float live = ol < no ? 1.f : 0.f;   // rows past O contribute nothing (their wp is zero; gm must be too)
      const float d = cur.dv[ol], d2r = d * d * r[ol];
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) {
        const float4 gm = demod ? (cur.gp[ol][jj] - (wp[ol][jj] * cur.t[jj]) * d2r) * (d * live) : cur.gp[ol][jj] * live;
        const float4 gwv = gm * cur.t[jj];
        gw[ol][jj] = gw[ol][jj] + gwv;
        gt[jj] = gt[jj] + gm * wp[ol][jj];
        part += hsum(gwv * wp[ol][jj]);
      }
    }
    float* gtr = a.gtp[l] + ((int64_t)og * a.B + b) * I;
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj)
      if (in[jj]) st4<float>(gtr + 4 * (lane + 64 * jj), gt[jj]);
    cur = nxt;
  }
#pragma unroll
  for (int ol = 0; ol < V4_OG; ++ol) {
    if (ol >= no) break;
    float* gwr = a.gwp[l] + ((int64_t)bc * O + o0 + ol) * I;
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj)
      if (in[jj]) st4<float>(gwr + 4 * (lane + 64 * jj), gw[ol][jj]);
  }
  part = wave_sum(part);
  if (lane == 0) a.corr[l][unit] = part;
}

// units of layer l: ceil(O / 4) row groups x ceil(B / V4_BC) sample chunks, one WAVE each (four per block)
__global__ __launch_bounds__(PB) void mod_prep_all_bwd_v4_kernel(PrepAllB a_by_value) {
  const PrepAllB& a = kernarg_view(a_by_value);
  int local;
  const int l = find_layer(a, blockIdx.x, local);
  const int nog = (a.O[l] + V4_OG - 1) / V4_OG, nbc = (a.B + V4_BC - 1) / V4_BC;
  const int unit = local * 4 + (threadIdx.x >> 6);
  if (unit >= nog * nbc) return;
  const int og = unit % nog, bc = unit / nog;
  switch ((a.I[l] + 255) >> 8) {
    case 1: prep_bwd_v4<1>(a, l, og, bc, unit); break;
    case 2: prep_bwd_v4<2>(a, l, og, bc, unit); break;
    case 3: prep_bwd_v4<3>(a, l, og, bc, unit); break;
    default: prep_bwd_v4<4>(a, l, og, bc, unit); break;
  }
}

// gs[b, :] = sum over the row groups' partials (+ the max-norm term of s' = s / smax on demodulating layers): one block
// per (layer, sample); I <= 1024: one float4 chunk per thread
__global__ __launch_bounds__(PB) void mod_prep_all_s_fix_v4_kernel(PrepAllB a_by_value) {
  const PrepAllB& a = kernarg_view(a_by_value);
  __shared__ float red[4];
  int b;
  const int l = find_layer(a, blockIdx.x, b);
  const int I = a.I[l], O = a.O[l], nog = (O + V4_OG - 1) / V4_OG;
  const bool demod = a.flags[l] & 1;
  const int i = threadIdx.x * 4;
  const bool in = i < I;
  float4 acc = f4(0.f);
  if (in) {
    const float* p = a.gtp[l] + (int64_t)b * I + i;
    const int64_t step = (int64_t)a.B * I;
    float4 a0 = f4(0.f), a1 = f4(0.f), a2 = f4(0.f), a3 = f4(0.f);
    int g = 0;
    for (; g + 3 < nog; g += 4) {
      a0 = a0 + ld4(p + (g + 0) * step);
      a1 = a1 + ld4(p + (g + 1) * step);
      a2 = a2 + ld4(p + (g + 2) * step);
      a3 = a3 + ld4(p + (g + 3) * step);
    }
    for (; g < nog; ++g) a0 = a0 + ld4(p + g * step);
    acc = (a0 + a1) + (a2 + a3);
  }
  float* gs = a.out[l] + (size_t)O * I + (int64_t)b * I;
  if (!demod) {
    if (in) st4<float>(gs + i, acc);
    return;
  }
  const float smax = a.stats[(size_t)l * (2 + 2 * a.B) + 2 + 2 * b];
  const float4 sv = in ? ld4(a.s[l] + (int64_t)b * I + i) : f4(0.f);
  float dot = hsum(acc * sv);
  dot = block_sum(dot, red);
  if (in) {
    const float k = dot / (smax * smax);
    float4 v = make_float4(acc.x / smax, acc.y / smax, acc.z / smax, acc.w / smax);
    if (fabsf(sv.x) == smax) v.x -= (sv.x > 0.f ? 1.f : -1.f) * k;
    if (fabsf(sv.y) == smax) v.y -= (sv.y > 0.f ? 1.f : -1.f) * k;
    if (fabsf(sv.z) == smax) v.z -= (sv.z > 0.f ? 1.f : -1.f) * k;
    if (fabsf(sv.w) == smax) v.w -= (sv.w > 0.f ? 1.f : -1.f) * k;
    st4<float>(gs + i, v);
  }
}

// gW = (sum over the sample chunks' partials) * k, minus the max-norm term of W' = W / wmax on demodulating layers
__global__ __launch_bounds__(256) void mod_prep_all_w_fix_v4_kernel(PrepAllB a_by_value, int blocks_per_layer_cap) {
  const PrepAllB& a = kernarg_view(a_by_value);
  __shared__ float red[4];
  int local;
  const int l = find_layer(a, blockIdx.x, local);
  const int O = a.O[l], I = a.I[l], OI = O * I;
  const bool demod = a.flags[l] & 1;
  const int nbc = (a.B + V4_BC - 1) / V4_BC, nog = (O + V4_OG - 1) / V4_OG;
  const int nblk = (l ? a.blk_end[l] - a.blk_end[l - 1] : a.blk_end[0]);
  const float* stats = a.stats + (size_t)l * (2 + 2 * a.B);
  const float wmax = demod ? stats[0] : 1.f;
  const float k = demod ? 1.f / wmax : rsqrtf((float)I);
  float csum = 0.f;
  if (demod) {
    const int nc = nog * nbc;
    for (int t = threadIdx.x; t < nc; t += 256) csum += a.corr[l][t];
    csum = block_sum(csum, red);
  }
  float* gW = a.out[l];
  const float* W = a.W[l];
  const float* p = a.gwp[l];
  for (int i = (local * 256 + threadIdx.x) * 4; i < OI; i += nblk * 1024) {
    float4 v = ld4(p + i);
    for (int c = 1; c < nbc; ++c) v = v + ld4(p + (int64_t)c * OI + i);
    v = v * k;
    if (demod) {
      const float4 w = ld4(W + i);
      const float f = csum / wmax;
      if (fabsf(w.x) == wmax) v.x -= (w.x > 0.f ? 1.f : -1.f) * f;
      if (fabsf(w.y) == wmax) v.y -= (w.y > 0.f ? 1.f : -1.f) * f;
      if (fabsf(w.z) == wmax) v.z -= (w.z > 0.f ? 1.f : -1.f) * f;
      if (fabsf(w.w) == wmax) v.w -= (w.w > 0.f ? 1.f : -1.f) * f;
    }
    st4<float>(gW + i, v);
  }
  (void)blocks_per_layer_cap;
}

bool geom_ok(int B, int O, int I, int Otot, int row_off, int cin, int F) {
  return B > 0 && O > 0 && I > 0 && I <= PB * MAXJ && Otot >= row_off + O && row_off >= 0 &&
         (F == 0 || (F == PB && cin >= 0 && cin + 2 * F <= I));
}

}  // namespace

// Input-magnitude EMA of ModConv2d (style.py:98-103) as one scalar kernel:
//   ema <- lerp(ema, (sum(sumsq[0..nsum)) + add) * inv_count, weight)   (when update)
//   snapshot <- ema        (the value this forward pass uses; the buffer itself keeps changing)
// fold of the nsum partial sums by ONE block of EMA_NT threads, the same bits in both kernels below: thread t sums the
// partials t, t + EMA_NT, ... (four running sums, every load of a thread in flight at once), a wave folds its 64 sums,
// wave 0 adds the waves' results in wave order.  (One wave walking 8192 partials four loads at a time was 32 dependent
// round trips to L2: 4.9 us per launch, 28 launches per iteration.)
constexpr int EMA_NT = 256;
static __device__ __forceinline__ float ema_fold(const float* sumsq, int nsum) {
  __shared__ float s_w[EMA_NT / 64];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (sumsq) {
    int k = threadIdx.x;
    for (; k + 3 * EMA_NT < nsum; k += 4 * EMA_NT) {
      s0 += sumsq[k];
      s1 += sumsq[k + EMA_NT];
      s2 += sumsq[k + 2 * EMA_NT];
      s3 += sumsq[k + 3 * EMA_NT];
    }
    for (; k < nsum; k += EMA_NT) s0 += sumsq[k];
  }
  const float w = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = w;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < EMA_NT / 64; ++i) s += s_w[i];
  return s;   // every thread holds the same value
}

static __global__ __launch_bounds__(EMA_NT) void ema_scalar_kernel(float* ema, float* snapshot, const float* sumsq, int nsum,
                                                                   float add, float inv_count, float weight, int update,
                                                                   float* cvec, int ncvec) {
  const float s = ema_fold(sumsq, nsum);
  float v = ema[0];   // every lane computes the same value; thread 0 stores it
  if (update) v += weight * ((s + add) * inv_count - v);
  __syncthreads();    // every thread has read ema[0]
  if (threadIdx.x == 0) {
    if (update) ema[0] = v;
    if (snapshot) snapshot[0] = v;
  }
  const float c = 1.f / (sqrtf(v) + 1e-8f);   // the factor the modulated conv applies to its output rows
  for (int i = threadIdx.x; i < ncvec; i += EMA_NT) cvec[i] = c;
}

extern "C" int dgv2_ema_scalar(float* ema, float* snapshot, const float* sumsq, int nsum, float add, float inv_count,
                               float weight, int update, float* cvec, int ncvec, void* stream) {
  if (!ema || (!snapshot && !cvec) || nsum < 0 || ncvec < 0 || (ncvec > 0 && !cvec)) return DGV2_EINVAL;
  ema_scalar_kernel<<<1, EMA_NT, 0, (hipStream_t)stream>>>(ema, snapshot, sumsq, nsum, add, inv_count, weight, update, cvec,
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
static __global__ __launch_bounds__(EMA_NT) void ema_scalar_group_kernel(EmaGroup grp, const float* sumsq, int nsum, float add,
                                                                         float inv_count, float weight, int update,
                                                                         float* cvec) {
  const float s = ema_fold(sumsq, nsum);   // the summation order of ema_scalar_kernel: same bits
  const int i = blockIdx.x;
  float v = grp.ema[i][0];
  if (update) v += weight * ((s + add) * inv_count - v);
  __syncthreads();
  if (threadIdx.x == 0 && update) grp.ema[i][0] = v;
  int off = 0;
  for (int j = 0; j < i; ++j) off += grp.rows[j];
  const float c = 1.f / (sqrtf(v) + 1e-8f);
  for (int r = threadIdx.x; r < grp.rows[i]; r += EMA_NT) cvec[off + r] = c;
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
  ema_scalar_group_kernel<<<n, EMA_NT, 0, (hipStream_t)stream>>>(grp, sumsq, nsum, add, inv_count, weight, update, cvec);
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
// the V4 kernels' conditions: whole float4 chunks per row, rotation pairs chunk-aligned; DGV2_NO_PREP_V4: A/B switch
static bool v4_ok(const int* I, const int* cin, const int* flags, int L, bool rotating) {
  static const bool off = getenv("DGV2_NO_PREP_V4") != nullptr;
  if (off) return false;
  for (int l = 0; l < L; ++l)
    if ((I[l] & 3) || (rotating && (flags[l] & 2) && (cin[l] & 3))) return false;
  return true;
}

// fp32 scratch of the V4 backward per layer: gt partials [ceil(O/4)][B][I], gW partials [ceil(B/8)][O][I], one slot per unit
static int64_t v4_bwd_scratch(int O, int I, int B, int64_t* gtp = nullptr, int64_t* gwp = nullptr) {
  const int64_t nog = (O + V4_OG - 1) / V4_OG, nbc = (B + V4_BC - 1) / V4_BC;
  const int64_t n_gt = nog * B * I, n_gw = nbc * O * I, n_c = (nog * nbc + 3) / 4 * 4;
  if (gtp) *gtp = 0;
  if (gwp) *gwp = n_gt;
  return n_gt + n_gw + n_c;
}

extern "C" int dgv2_mod_prep_all_bwd_scratch(int64_t* elems, const int* O, const int* I, int B, int L) {
  if (!elems || !O || !I || B < 1 || L < 1 || L > MPA_MAX) return DGV2_EINVAL;
  int64_t n = 0;
  for (int l = 0; l < L; ++l) {
    if (O[l] < 1 || I[l] < 1) return DGV2_EINVAL;
    n += v4_bwd_scratch(O[l], I[l], B);
  }
  *elems = n;
  return 0;
}

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
  bool al = true;
  for (int l = 0; l < L && al; ++l) al = aligned16(wb[l]) && aligned16(W[l]) && aligned16(s[l]);
  if (al && aligned16(rot) && v4_ok(I, cin, flags, L, shift != nullptr)) {
    for (int l = 0; l < L; ++l) {
      nblk += ((O[l] + 3) / 4) * ((B + V4_BC - 1) / V4_BC);   // four rows (waves) x V4_BC samples per block
      a.blk_end[l] = nblk;
    }
    mod_prep_all_fwd_v4_kernel<<<nblk, PB, 0, st>>>(a);
    DGV2_RETURN_LAST();
  }
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
                                     int L, float* scratch, int64_t scratch_elems, void* stream) {
  PrepAll a;
  if (!flat || flat_elems < 1 || !out || !ncorr || !G || !stats || !dsave || !rot) return DGV2_EINVAL;
  int rc = fill_common(a, W, s, fw, O, I, Otot, row_off, cin, flags, B, L);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  int64_t need = 0;
  for (int l = 0; l < L; ++l) need += v4_bwd_scratch(O[l], I[l], B);
  bool al = true;
  for (int l = 0; l < L && al; ++l)
    al = out[l] && G[l] && aligned16(out[l]) && aligned16(G[l]) && aligned16(W[l]) && aligned16(s[l]);
  if (al && scratch && scratch_elems >= need && aligned16(scratch) && v4_ok(I, cin, flags, L, shift != nullptr)) {
    // V4: partial sums in scratch, folded by the fix-up kernels -- no atomics, nothing to clear
    PrepAllB q;
    int64_t off = 0;
    int n1 = 0;
    for (int l = 0; l < L; ++l) {
      if (!out[l] || !G[l] || !dsave[l]) return DGV2_EINVAL;
      if (out[l] < flat || out[l] + (size_t)O[l] * I[l] + (size_t)B * I[l] > flat + flat_elems) return DGV2_EINVAL;
      int64_t o_gt, o_gw;
      const int64_t n = v4_bwd_scratch(O[l], I[l], B, &o_gt, &o_gw);
      const int64_t nog = (O[l] + V4_OG - 1) / V4_OG, nbc = (B + V4_BC - 1) / V4_BC;
      q.W[l] = W[l]; q.s[l] = s[l]; q.G[l] = G[l]; q.dsave[l] = dsave[l]; q.out[l] = out[l];
      q.gtp[l] = scratch + off + o_gt; q.gwp[l] = scratch + off + o_gw; q.corr[l] = scratch + off + n - (nog * nbc + 3) / 4 * 4;
      q.O[l] = O[l]; q.I[l] = I[l]; q.Otot[l] = Otot[l]; q.row_off[l] = row_off[l]; q.cin[l] = cin[l]; q.flags[l] = flags[l];
      off += n;
      n1 += (int)((nog * nbc + 3) / 4);
      q.blk_end[l] = n1;
    }
    for (int l = L; l < MPA_MAX; ++l) q.blk_end[l] = 0x7fffffff;
    q.stats = stats; q.rot = rot; q.shift = shift; q.B = B; q.L = L;
    mod_prep_all_bwd_v4_kernel<<<n1, PB, 0, st>>>(q);
    int n2 = 0;
    for (int l = 0; l < L; ++l) {
      n2 += B;
      q.blk_end[l] = n2;
    }
    mod_prep_all_s_fix_v4_kernel<<<n2, PB, 0, st>>>(q);
    int n3 = 0;
    for (int l = 0; l < L; ++l) {
      n3 += grid_for(((int64_t)O[l] * I[l] + 1023) / 1024, 1, 64);
      q.blk_end[l] = n3;
    }
    mod_prep_all_w_fix_v4_kernel<<<n3, 256, 0, st>>>(q, 64);
    DGV2_RETURN_LAST();
  }
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
