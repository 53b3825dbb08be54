// Discriminator stem in one pass: BlurVH -> 1x1 conv (2 -> O channels) -> bias + leaky ReLU.
// reference: Discriminator.__init__ layers[0:3] (gans/models/dusty_v2.py:364-367), BlurVH
// (gans/models/ops/common.py:141-155: cat(blur_v(x), blur_h(x)), taps [1,2,1]/4, circular W /
// replicate H), ops.Conv2d 1x1, FusedLeakyReLU.
//
// The input has ONE channel, so the whole stem is 5 loads, 2 FIR sums and 2*O FMAs per pixel: a streaming
// kernel bound by writing the [B,H,W,O] activation once.  Routing it through the resampler (C = 1, not
// vectorisable), a concat and the MFMA conv engine padded to a 32-channel K-step cost ~8x that.
//   forward : y[b,h,w,o] = lrelu(w[o,0] v + w[o,1] u + bias[o]) * scale,
//             v = (x[h-1] + 2 x[h] + x[h+1]) / 4 (h clamped), u = the same along w (wrapped / clamped)
//   backward: gpre = gy * lrelu'(y) * scale;  gb[o] = sum gpre;  gw[o,0] = sum gpre v;  gw[o,1] = sum gpre u;
//             gxb[.,0] = sum_o gpre w[o,0], gxb[.,1] = sum_o gpre w[o,1];  gx = blur_v^T gxb0 + blur_h^T gxb1.
// The per-block partial sums of (gb, gw) go to scratch and a second tiny kernel adds them (same-address float
// atomics from hundreds of blocks onto 96 floats would serialise).
#include "common.h"

namespace {

struct StemGeom {
  int B, H, W, O, ring;
  float alpha, scale;
  int nt;   // streaming output stores (outputs >= DGV2_NT_MIN_MB)
};

__device__ __forceinline__ int stem_wcoord(int w, int W, int ring) {
  if (ring) return w < 0 ? w + W : (w >= W ? w - W : w);
  return w < 0 ? 0 : (w >= W ? W - 1 : w);
}

__device__ __forceinline__ void stem_blur(const float* __restrict__ xb, int h, int w, const StemGeom& g, float& v,
                                          float& u) {
  const float c = xb[h * g.W + w];
  const float up = xb[(h > 0 ? h - 1 : 0) * g.W + w], dn = xb[(h + 1 < g.H ? h + 1 : g.H - 1) * g.W + w];
  const float lf = xb[h * g.W + stem_wcoord(w - 1, g.W, g.ring)], rt = xb[h * g.W + stem_wcoord(w + 1, g.W, g.ring)];
  v = 0.25f * (up + dn) + 0.5f * c;
  u = 0.25f * (lf + rt) + 0.5f * c;
}

// one thread = one pixel x 8 output channels (16 bytes of bf16 / 32 bytes of fp32)
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(T* __restrict__ y, const float* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       StemGeom g) {
  const int G = g.O / 8;
  const int64_t items = (int64_t)g.B * g.H * g.W * G;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;   // multiple of G (host-checked)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = (int)(i % G);
  float w0[8], w1[8], bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    w0[j] = w[(grp * 8 + j) * 2];
    w1[j] = w[(grp * 8 + j) * 2 + 1];
    bs[j] = bias ? bias[grp * 8 + j] : 0.f;
  }
  const int HW = g.H * g.W;
  for (; i < items; i += stride) {
    const int64_t px = i / G;
    const int b = (int)(px / HW), r = (int)(px - (int64_t)b * HW);
    const int h = r / g.W, wc = r - h * g.W;
    float v, u;
    stem_blur(x + (int64_t)b * HW, h, wc, g, v, u);
    T* out = y + px * g.O + grp * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = w0[j] * v + w1[j] * u + bs[j];
      f[j] = (t > 0.f ? t : t * g.alpha) * g.scale;
    }
    if constexpr (sizeof(T) == 2) {
      vec16<T> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.set(j, f[j]);
      if (g.nt) o.store_nt(out);
      else o.store(out);
    } else {
      *reinterpret_cast<float4*>(out) = make_float4(f[0], f[1], f[2], f[3]);
      *reinterpret_cast<float4*>(out + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
  }
}

// The same pass with one thread = FOUR pixels along W x 8 output channels (W % 4 == 0): three 16-byte loads and two
// neighbours for four pixels instead of twenty scalar loads, one coordinate division per four stores -- the
// one-pixel form wrote 268 MB in 75 us (3.6 TB/s), bound by its load latency and address arithmetic, not by the stores.
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd4_kernel(T* __restrict__ y, const float* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        StemGeom g) {
  const int G = g.O / 8, W4 = g.W >> 2;
  const int64_t items = (int64_t)g.B * g.H * W4 * G;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;   // multiple of G (host-checked)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = (int)(i % G);
  float w0[8], w1[8], bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    w0[j] = w[(grp * 8 + j) * 2];
    w1[j] = w[(grp * 8 + j) * 2 + 1];
    bs[j] = bias ? bias[grp * 8 + j] : 0.f;
  }
  const int HW4 = g.H * W4;
  for (; i < items; i += stride) {
    const int64_t q = i / G;
    const int b = (int)(q / HW4), r = (int)(q - (int64_t)b * HW4);
    const int h = r / W4, wc = (r - h * W4) << 2;
    const float* xb = x + ((int64_t)b * g.H) * g.W;
    const float4 c4 = *reinterpret_cast<const float4*>(xb + h * g.W + wc);
    const float4 u4 = *reinterpret_cast<const float4*>(xb + (h > 0 ? h - 1 : 0) * g.W + wc);
    const float4 d4 = *reinterpret_cast<const float4*>(xb + (h + 1 < g.H ? h + 1 : g.H - 1) * g.W + wc);
    const float lf = xb[h * g.W + stem_wcoord(wc - 1, g.W, g.ring)], rt = xb[h * g.W + stem_wcoord(wc + 4, g.W, g.ring)];
    const float c[6] = {lf, c4.x, c4.y, c4.z, c4.w, rt};
    const float up[4] = {u4.x, u4.y, u4.z, u4.w}, dn[4] = {d4.x, d4.y, d4.z, d4.w};
    T* out = y + (((int64_t)b * g.H + h) * g.W + wc) * g.O + grp * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = 0.25f * (up[k] + dn[k]) + 0.5f * c[k + 1];       // the operation order of stem_blur: same bits
      const float u = 0.25f * (c[k] + c[k + 2]) + 0.5f * c[k + 1];
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = w0[j] * v + w1[j] * u + bs[j];
        f[j] = (t > 0.f ? t : t * g.alpha) * g.scale;
      }
      if constexpr (sizeof(T) == 2) {
        vec16<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.set(j, f[j]);
        if (g.nt) o.store_nt(out + k * g.O);
        else o.store(out + k * g.O);
      } else {
        *reinterpret_cast<float4*>(out + k * g.O) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(out + k * g.O + 4) = make_float4(f[4], f[5], f[6], f[7]);
      }
    }
  }
}

// The stem's output feeds the first ResidualBlock twice: conv1, and the skip branch through a decimating blur
// (dusty_v2.py:337-345: self.skip(self.resample(x)), evaluated at the even positions only).  SKIP: the gradient of that
// blurred / decimated image, gsk [B, Hs, Ws, O], is gathered HERE through the ADJOINT tables of the blur (rows of the
// full-resolution grid naming the <= E decimated positions they fed, as dgv2_resample_tab takes them) instead of being
// scattered to a full-resolution tensor by a pass of its own, added to conv1's data gradient by another and read back
// by this one: per iteration 2 x 268 MB of traffic and a launch less at 2B = 128.
template <typename T>
struct StemSkip {
  const T* gsk;
  const int* idx_h; const float* coef_h; const int* cnt_h; int Eh;
  const int* idx_w; const float* coef_w; const int* cnt_w; int Ew;
  int Hs, Ws;
};

// partial[block][3*O]: gb[o], gw[o,0], gw[o,1];  gxb fp32 [B,H,W,2]
template <typename T, bool SKIP>
__global__ __launch_bounds__(256) void stem_bwd_kernel(float* __restrict__ partial, float* __restrict__ gxb,
                                                       const T* __restrict__ gy, const T* __restrict__ y,
                                                       const float* __restrict__ x, const float* __restrict__ w,
                                                       StemGeom g, StemSkip<T> sk) {
  __shared__ float red[4][8][24];
  extern __shared__ __attribute__((aligned(16))) unsigned char stem_tabs[];   // SKIP: idx_h | coef_h | idx_w | coef_w
  const int* t_ih = nullptr; const float* t_ch = nullptr; const int* t_iw = nullptr; const float* t_cw = nullptr;
  if constexpr (SKIP) {
    const int nh_ = g.H * sk.Eh, nw_ = g.W * sk.Ew;
    int* a0 = reinterpret_cast<int*>(stem_tabs);
    float* a1 = reinterpret_cast<float*>(a0 + nh_);
    int* a2 = reinterpret_cast<int*>(a1 + nh_);
    float* a3 = reinterpret_cast<float*>(a2 + nw_);
    for (int t = threadIdx.x; t < nh_; t += 256) {
      a0[t] = sk.idx_h[t];
      a1[t] = sk.coef_h[t];
    }
    for (int t = threadIdx.x; t < nw_; t += 256) {
      a2[t] = sk.idx_w[t];
      a3[t] = sk.coef_w[t];
    }
    __syncthreads();
    t_ih = a0; t_ch = a1; t_iw = a2; t_cw = a3;
  }
  const int G = g.O / 8;                  // 1, 2, 4 or 8: a pixel's threads are adjacent lanes
  const int64_t items = (int64_t)g.B * g.H * g.W * G;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = (int)(i % G);
  float w0[8], w1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    w0[j] = w[(grp * 8 + j) * 2];
    w1[j] = w[(grp * 8 + j) * 2 + 1];
  }
  float acc[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) acc[j] = 0.f;
  const int HW = g.H * g.W;
  const int64_t nloop = (items + stride - 1) / stride;       // every lane runs every iteration (shuffles)
  for (int64_t it = 0; it < nloop; ++it, i += stride) {
    const bool live = i < items;
    const int64_t px = live ? i / G : 0;
    const int b = (int)(px / HW), r = (int)(px - (int64_t)b * HW);
    const int h = r / g.W, wc = r - h * g.W;
    float v, u;
    stem_blur(x + (int64_t)b * HW, h, wc, g, v, u);
    float gp[8], gs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gs[j] = 0.f;
    if constexpr (SKIP) {   // blur_down^T(gsk) at this pixel: <= Eh x Ew decimated positions
      // the tables sit in LDS (copied once per block: a global cnt -> idx -> data chain per pixel made this pass latency
      // bound, 504 us instead of 136); the first 2 x 2 entries -- all there is away from the clamped border rows -- are
      // requested together, unconditionally (unused entries carry coefficient 0 and index 0)
      const T* sb = sk.gsk + (int64_t)b * sk.Hs * sk.Ws * g.O + grp * 8;
      const int* ihp = t_ih + h * sk.Eh;
      const float* chp = t_ch + h * sk.Eh;
      const int* iwp = t_iw + wc * sk.Ew;
      const float* cwp = t_cw + wc * sk.Ew;
      const int e1h = sk.Eh > 1 ? 1 : 0, e1w = sk.Ew > 1 ? 1 : 0;
      const float ch0 = chp[0], ch1 = sk.Eh > 1 ? chp[1] : 0.f, cw0 = cwp[0], cw1 = sk.Ew > 1 ? cwp[1] : 0.f;
      const T* r0 = sb + (int64_t)ihp[0] * sk.Ws * g.O;
      const T* r1 = sb + (int64_t)ihp[e1h] * sk.Ws * g.O;
      const int64_t c0 = (int64_t)iwp[0] * g.O, c1 = (int64_t)iwp[e1w] * g.O;
      auto acc8 = [&](const T* sp, float cf) {
        if constexpr (sizeof(T) == 2) {
          vec16<T> v8;
          v8.load(sp);
#pragma unroll
          for (int j = 0; j < 8; ++j) gs[j] = fmaf(cf, v8.get(j), gs[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) gs[j] = fmaf(cf, (float)sp[j], gs[j]);
        }
      };
      acc8(r0 + c0, ch0 * cw0);
      acc8(r0 + c1, ch0 * cw1);
      acc8(r1 + c0, ch1 * cw0);
      acc8(r1 + c1, ch1 * cw1);
      for (int a = 0; a < sk.Eh; ++a)        // what the 2 x 2 block did not cover (border rows / longer tables)
        for (int c = 0; c < sk.Ew; ++c) {
          if (a < 2 && c < 2) continue;
          const float cf = chp[a] * cwp[c];
          if (cf != 0.f) acc8(sb + (int64_t)ihp[a] * sk.Ws * g.O + (int64_t)iwp[c] * g.O, cf);
        }
    }
    if constexpr (sizeof(T) == 2) {
      vec16<T> a, c;
      a.load(gy + px * g.O + grp * 8);
      c.load(y + px * g.O + grp * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) gp[j] = (a.get(j) + gs[j]) * (c.get(j) > 0.f ? 1.f : g.alpha) * g.scale;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = gy[px * g.O + grp * 8 + j], c = y[px * g.O + grp * 8 + j];
        gp[j] = (a + gs[j]) * (c > 0.f ? 1.f : g.alpha) * g.scale;
      }
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float p = live ? gp[j] : 0.f;
      acc[j] += p;
      acc[8 + j] += p * v;
      acc[16 + j] += p * u;
      s0 += p * w0[j];
      s1 += p * w1[j];
    }
    for (int o = 1; o < G; o <<= 1) {     // sum over the pixel's G adjacent lanes
      s0 += __shfl_xor(s0, o, 64);
      s1 += __shfl_xor(s1, o, 64);
    }
    if (live && grp == 0) *reinterpret_cast<float2*>(gxb + px * 2) = make_float2(s0, s1);
  }
  // fold lanes with equal grp inside the wave (xor offsets >= G keep lane % G), then the four waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 24; ++j) {
    float t = acc[j];
    for (int o = 32; o >= G; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane < G) red[wave][lane][j] = t;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < G * 24; t += 256) {
    const int gi = t / 24, j = t % 24;
    const float s = red[0][gi][j] + red[1][gi][j] + red[2][gi][j] + red[3][gi][j];
    // j = 0..7 gb, 8..15 gw[.,0], 16..23 gw[.,1]
    partial[(int64_t)blockIdx.x * 3 * g.O + (j / 8) * g.O + gi * 8 + (j & 7)] = s;
  }
}

// gb[o] = sum_blk partial[blk][o];  gw[o,c] = sum_blk partial[blk][(1+c)*O + o]
__global__ __launch_bounds__(256) void stem_reduce_kernel(float* __restrict__ gw, float* __restrict__ gb,
                                                          const float* __restrict__ partial, int nblk, int O) {
  __shared__ float red[4];
  const int e = blockIdx.x;   // 0 .. 3*O-1
  float s = 0.f;
  for (int k = threadIdx.x; k < nblk; k += 256) s += partial[(int64_t)k * 3 * O + e];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3];
    if (e < O) gb[e] = t;
    else gw[(e % O) * 2 + (e / O - 1)] = t;
  }
}

// gx = blur_v^T gxb[.,0] + blur_h^T gxb[.,1] (transposes of the clamped / wrapped 3-tap FIRs)
__global__ __launch_bounds__(256) void stem_blur_adj_kernel(float* __restrict__ gx, const float* __restrict__ gxb,
                                                            StemGeom g) {
  const int64_t n = (int64_t)g.B * g.H * g.W;
  const int HW = g.H * g.W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), r = (int)(i - (int64_t)b * HW);
    const int h = r / g.W, w = r - h * g.W;
    const float* gb = gxb + (int64_t)b * HW * 2;
    auto G0 = [&](int hh, int ww) { return gb[(hh * g.W + ww) * 2]; };
    auto G1 = [&](int hh, int ww) { return gb[(hh * g.W + ww) * 2 + 1]; };
    float s = 0.5f * G0(h, w);
    if (h > 0) s += 0.25f * G0(h - 1, w);
    if (h + 1 < g.H) s += 0.25f * G0(h + 1, w);
    if (h == 0) s += 0.25f * G0(0, w);              // row -1 was read as row 0
    if (h == g.H - 1) s += 0.25f * G0(g.H - 1, w);  // row H as row H-1
    s += 0.5f * G1(h, w);
    if (g.ring) {
      s += 0.25f * (G1(h, w > 0 ? w - 1 : g.W - 1) + G1(h, w + 1 < g.W ? w + 1 : 0));
    } else {
      if (w > 0) s += 0.25f * G1(h, w - 1);
      if (w + 1 < g.W) s += 0.25f * G1(h, w + 1);
      if (w == 0) s += 0.25f * G1(h, 0);
      if (w == g.W - 1) s += 0.25f * G1(h, g.W - 1);
    }
    gx[i] = s;
  }
}

constexpr int STEM_BWD_BLOCKS = 1024;

bool stem_ok(int B, int H, int W, int O) {
  return B > 0 && H >= 2 && W >= 2 && (O == 8 || O == 16 || O == 32 || O == 64) && (int64_t)H * W < (1 << 30);
}

}  // namespace

// y [B,H,W,O] (ydtype) = stem(x fp32 [B,H,W]); w fp32 [O,2] (column 0: the blur_v channel), bias fp32 [O] or NULL.
extern "C" int dgv2_stem_fwd(void* y, const float* x, const float* w, const float* bias, int B, int H, int W, int O,
                             int ring, float alpha, float scale, int ydtype, void* stream) {
  if (!y || !x || !w || !stem_ok(B, H, W, O) || !aligned16(y)) return DGV2_EINVAL;
  StemGeom g{B, H, W, O, ring, alpha, scale, nt_output((int64_t)B * H * W * O * (ydtype == DGV2_BF16 ? 2 : 4)) ? 1 : 0};
  hipStream_t st = (hipStream_t)stream;
  static const bool no4 = getenv("DGV2_STEM_NO4") != nullptr;   // A/B switch for benchmarking
  if (W % 4 == 0 && aligned16(x) && !no4) {
    const int64_t items = (int64_t)B * H * (W / 4) * (O / 8);
    const int grid = grid_for(items, 256, 256 * 32);
    if (ydtype == DGV2_BF16) stem_fwd4_kernel<bf16_t><<<grid, 256, 0, st>>>((bf16_t*)y, x, w, bias, g);
    else if (ydtype == DGV2_F32) stem_fwd4_kernel<float><<<grid, 256, 0, st>>>((float*)y, x, w, bias, g);
    else return DGV2_EINVAL;
    DGV2_RETURN_LAST();
  }
  const int64_t items = (int64_t)B * H * W * (O / 8);
  const int grid = grid_for(items, 256, 256 * 32);
  if (ydtype == DGV2_BF16) stem_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((bf16_t*)y, x, w, bias, g);
  else if (ydtype == DGV2_F32) stem_fwd_kernel<float><<<grid, 256, 0, st>>>((float*)y, x, w, bias, g);
  else return DGV2_EINVAL;
  DGV2_RETURN_LAST();
}

// Number of fp32 scratch elements dgv2_stem_bwd needs: per-block partial sums + the two-channel gxb image.
extern "C" int dgv2_stem_bwd_scratch(int64_t* elems, int B, int H, int W, int O) {
  if (!elems || !stem_ok(B, H, W, O)) return DGV2_EINVAL;
  *elems = (int64_t)STEM_BWD_BLOCKS * 3 * O + (int64_t)B * H * W * 2;
  return 0;
}

// gw fp32 [O,2], gb fp32 [O], gx fp32 [B,H,W] (NULL: not needed) from gy, y [B,H,W,O] (dtype), x fp32 [B,H,W].
extern "C" int dgv2_stem_bwd(float* gx, float* gw, float* gb, float* scratch, int64_t scratch_elems, const void* gy,
                             const void* y, const float* x, const float* w, int B, int H, int W, int O, int ring,
                             float alpha, float scale, int dtype, void* stream) {
  return dgv2_stem_bwd_skip(gx, gw, gb, scratch, scratch_elems, gy, y, x, w, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                            nullptr, nullptr, 0, 0, 0, B, H, W, O, ring, alpha, scale, dtype, stream);
}

// ... with gsk [B, Hs, Ws, O] (dtype; NULL: none), the gradient of a decimating blur of y (the first ResidualBlock's skip
// branch), gathered through the blur's ADJOINT tables (idx_h / coef_h / cnt_h: H rows of Eh entries naming rows of gsk;
// idx_w ...: W rows of Ew entries): the gradient of y is then gy + blur^T(gsk), never materialised.
extern "C" int dgv2_stem_bwd_skip(float* gx, float* gw, float* gb, float* scratch, int64_t scratch_elems, const void* gy,
                                  const void* y, const float* x, const float* w, const void* gsk, const int* idx_h,
                                  const float* coef_h, const int* cnt_h, int Eh, const int* idx_w, const float* coef_w,
                                  const int* cnt_w, int Ew, int Hs, int Ws, int B, int H, int W, int O, int ring,
                                  float alpha, float scale, int dtype, void* stream) {
  if (!gw || !gb || !scratch || !gy || !y || !x || !w || !stem_ok(B, H, W, O)) return DGV2_EINVAL;
  if (!aligned16(gy) || !aligned16(y) || !aligned16(scratch)) return DGV2_EINVAL;
  if (gsk && (!idx_h || !coef_h || !cnt_h || !idx_w || !coef_w || !cnt_w || Eh < 1 || Ew < 1 || Hs < 1 || Ws < 1 ||
              !aligned16(gsk)))
    return DGV2_EINVAL;
  const size_t tab_bytes = gsk ? 8 * ((size_t)H * Eh + (size_t)W * Ew) : 0;
  if (tab_bytes > 48 * 1024) return DGV2_ENOTSUP;   // the tables ride in LDS (callers then run the separate passes)
  const int64_t need = (int64_t)STEM_BWD_BLOCKS * 3 * O + (int64_t)B * H * W * 2;
  if (scratch_elems < need) return DGV2_EINVAL;
  StemGeom g{B, H, W, O, ring, alpha, scale, 0};
  const int64_t items = (int64_t)B * H * W * (O / 8);
  const int grid = grid_for(items, 256, STEM_BWD_BLOCKS);
  float* partial = scratch;
  float* gxb = scratch + (int64_t)STEM_BWD_BLOCKS * 3 * O;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DGV2_BF16) {
    StemSkip<bf16_t> sk{(const bf16_t*)gsk, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, Hs, Ws};
    if (gsk) stem_bwd_kernel<bf16_t, true><<<grid, 256, tab_bytes, st>>>(partial, gxb, (const bf16_t*)gy, (const bf16_t*)y, x, w, g, sk);
    else stem_bwd_kernel<bf16_t, false><<<grid, 256, 0, st>>>(partial, gxb, (const bf16_t*)gy, (const bf16_t*)y, x, w, g, sk);
  } else if (dtype == DGV2_F32) {
    StemSkip<float> sk{(const float*)gsk, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, Hs, Ws};
    if (gsk) stem_bwd_kernel<float, true><<<grid, 256, tab_bytes, st>>>(partial, gxb, (const float*)gy, (const float*)y, x, w, g, sk);
    else stem_bwd_kernel<float, false><<<grid, 256, 0, st>>>(partial, gxb, (const float*)gy, (const float*)y, x, w, g, sk);
  } else {
    return DGV2_EINVAL;
  }
  stem_reduce_kernel<<<3 * O, 256, 0, st>>>(gw, gb, partial, grid, O);
  if (gx) stem_blur_adj_kernel<<<grid_for((int64_t)B * H * W, 256, 4096), 256, 0, st>>>(gx, gxb, g);
  DGV2_RETURN_LAST();
}
