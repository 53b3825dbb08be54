// Ring-aware separable FIR resampler, channels-last, forward and exact transpose.
//
// Replaces Resample.forward (gans/models/ops/common.py:105-135): the reference pads,
// zero-stuffs, crops and runs two depthwise conv2d passes, materialising ~4x the data.
// Here each output pixel gathers its (few) non-zero taps directly from the unpadded
// input; ring / replicate extension is folded into the index computation.
//
// Per axis the operator is a sparse matrix R[n, j]: out[n] = sum_j R[n,j] in[j].  A block
// owns a TH x TW tile of outputs; its first threads build, in LDS, the sparse rows of
// R (forward) or of R^T (adjoint) for the tile's rows and columns; then every thread
// accumulates pixel x channel-vector products with 16-byte coalesced loads.
#include "common.h"

namespace {

constexpr int MAXE = 32;  // max non-zeros of one sparse row (k <= 8 taps, edge replicas included)
constexpr int TH = 4;
constexpr int TW = 16;

struct AxisParams {
  int L;      // forward input length
  int Lo;     // forward output length
  int k, up, down, p0;
  int wrap;   // 1 = circular, 0 = replicate
};

struct SparseRow {
  int cnt;
  int idx[MAXE];
  float coef[MAXE];
};

__device__ __forceinline__ int extend(int j, int L, int wrap) {
  if (wrap) return floormod(j, L);
  return j < 0 ? 0 : (j >= L ? L - 1 : j);
}

// forward row n: entries (input index, tap)
__device__ void build_fwd_row(SparseRow& r, const AxisParams& a, const float* taps, int n) {
  int cnt = 0;
  if (n < a.Lo) {
    for (int i = 0; i < a.k; ++i) {
      const int u = n * a.down + i - a.p0;
      if (floormod(u, a.up) != 0) continue;
      const int j = extend(floordiv(u, a.up), a.L, a.wrap);
      r.idx[cnt] = j;
      r.coef[cnt] = taps[i];
      ++cnt;
    }
  }
  r.cnt = cnt;
}

// adjoint row j (an index of the forward INPUT): entries (forward output index n, tap) over every
// virtual position j' of the extended signal that resolves to j.
__device__ void build_adj_row(SparseRow& r, const AxisParams& a, const float* taps, int j) {
  int cnt = 0;
  if (j < a.L) {
    const int jmin = floordiv(-a.p0, a.up);
    const int jmax = floordiv((a.Lo - 1) * a.down + a.k - 1 - a.p0, a.up);
    for (int jv = jmin; jv <= jmax; ++jv) {
      if (extend(jv, a.L, a.wrap) != j) continue;
      for (int i = 0; i < a.k; ++i) {
        const int num = jv * a.up + a.p0 - i;  // = n * down
        if (num < 0 || num % a.down != 0) continue;
        const int n = num / a.down;
        if (n >= a.Lo) continue;
        if (cnt < MAXE) {
          r.idx[cnt] = n;
          r.coef[cnt] = taps[i];
          ++cnt;
        }
      }
    }
  }
  r.cnt = cnt;
}

template <typename T, bool VEC>
__global__ void resample_kernel(T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ taps_h,
                                const float* __restrict__ taps_w, int B, int C, int ldx, int ldy, AxisParams ah,
                                AxisParams aw, int adjoint, int in_h, int in_w, int out_h, int out_w) {
  __shared__ SparseRow rows[TH];
  __shared__ SparseRow cols[TW];
  const int tiles_w = (out_w + TW - 1) / TW;
  const int tiles_h = (out_h + TH - 1) / TH;
  const int tile = blockIdx.x;
  const int b = tile / (tiles_h * tiles_w);
  const int th = (tile / tiles_w) % tiles_h;
  const int tw = tile % tiles_w;
  const int h0 = th * TH, w0 = tw * TW;

  if (threadIdx.x < TH) {
    if (adjoint) build_adj_row(rows[threadIdx.x], ah, taps_h, h0 + threadIdx.x);
    else build_fwd_row(rows[threadIdx.x], ah, taps_h, h0 + threadIdx.x);
  } else if (threadIdx.x >= 64 && threadIdx.x < 64 + TW) {
    const int t = threadIdx.x - 64;
    if (adjoint) build_adj_row(cols[t], aw, taps_w, w0 + t);
    else build_fwd_row(cols[t], aw, taps_w, w0 + t);
  }
  __syncthreads();

  constexpr int VN = VEC ? vec16<T>::N : 1;
  const int cvecs = (C + VN - 1) / VN;
  const int work = TH * TW * cvecs;
  const T* xb = x + (int64_t)b * in_h * in_w * ldx;
  T* yb = y + (int64_t)b * out_h * out_w * ldy;
  for (int it = threadIdx.x; it < work; it += blockDim.x) {
    const int cv = it % cvecs;
    const int pix = it / cvecs;
    const int lw = pix % TW, lh = pix / TW;
    const int ho = h0 + lh, wo = w0 + lw;
    if (ho >= out_h || wo >= out_w) continue;
    const SparseRow& rh = rows[lh];
    const SparseRow& rw = cols[lw];
    float acc[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[j] = 0.f;
    for (int a = 0; a < rh.cnt; ++a) {
      const float ch = rh.coef[a];
      const T* xr = xb + (int64_t)rh.idx[a] * in_w * ldx;
      for (int c = 0; c < rw.cnt; ++c) {
        const float cf = ch * rw.coef[c];
        const T* p = xr + (int64_t)rw.idx[c] * ldx + cv * VN;
        if (VEC) {
          vec16<T> v;
          v.load(p);
#pragma unroll
          for (int j = 0; j < VN; ++j) acc[j] += cf * v.get(j);
        } else {
          acc[0] += cf * to_f32(p[0]);
        }
      }
    }
    T* q = yb + ((int64_t)ho * out_w + wo) * ldy + cv * VN;
    if (VEC) {
      vec16<T> o;
#pragma unroll
      for (int j = 0; j < VN; ++j) o.set(j, acc[j]);
      o.store(q);
    } else {
      q[0] = from_f32<T>(acc[0]);
    }
  }
}

}  // namespace

extern "C" int dgv2_resample(void* y, const void* x, const float* taps_h, const float* taps_w, int B, int H, int W,
                             int C, int Ho, int Wo, int ldx, int ldy, int kh, int up_h, int down_h, int p0_h, int kw,
                             int up_w, int down_w, int p0_w, int ring, int adjoint, int dtype, void* stream) {
  if (!y || !x || !taps_h || !taps_w) return DGV2_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return DGV2_EINVAL;
  if (kh < 1 || kh > 8 || kw < 1 || kw > 8 || up_h < 1 || up_w < 1 || down_h < 1 || down_w < 1) return DGV2_EINVAL;
  if (ldx < C || ldy < C) return DGV2_EINVAL;
  AxisParams ah = {H, Ho, kh, up_h, down_h, p0_h, 0};
  AxisParams aw = {W, Wo, kw, up_w, down_w, p0_w, ring ? 1 : 0};
  const int in_h = adjoint ? Ho : H, in_w = adjoint ? Wo : W;
  const int out_h = adjoint ? H : Ho, out_w = adjoint ? W : Wo;
  const int tiles = B * ((out_h + TH - 1) / TH) * ((out_w + TW - 1) / TW);
  hipStream_t st = (hipStream_t)stream;
  DGV2_DISPATCH_DTYPE(dtype, {
    constexpr int VN = vec16<T>::N;
    const bool vec = (C % VN == 0) && (ldx % VN == 0) && (ldy % VN == 0) && aligned16(x) && aligned16(y);
    if (vec)
      resample_kernel<T, true><<<tiles, 256, 0, st>>>((T*)y, (const T*)x, taps_h, taps_w, B, C, ldx, ldy, ah, aw,
                                                     adjoint, in_h, in_w, out_h, out_w);
    else
      resample_kernel<T, false><<<tiles, 256, 0, st>>>((T*)y, (const T*)x, taps_h, taps_w, B, C, ldx, ldy, ah, aw,
                                                      adjoint, in_h, in_w, out_h, out_w);
  });
  DGV2_RETURN_LAST();
}

// ------------------------------------------------------------------------------------------------
// Table-driven variant: the per-axis sparse rows (index, coefficient, count) are geometry-only, so
// the host builds them once per (spec, size, direction) and caches them on the device.  The kernel
// is then pure streaming: one thread = one output pixel x one 16-byte channel vector, no LDS, no
// barrier, 32-bit index math.  Tables: idx/coef [n_out, E] (row-major), cnt [n_out].
// ------------------------------------------------------------------------------------------------
namespace {

// Few-channel images (the generator's 2-channel output pyramid, C <= 4, packed pixels): a thread owns a whole output
// pixel, so the table rows are fetched once per pixel instead of once per channel and the pixel is one 8/16-byte access.
template <typename T, int CC>
__global__ __launch_bounds__(256) void resample_tab_smallc_kernel(
    T* __restrict__ y, const T* __restrict__ x, const int* __restrict__ idx_h, const float* __restrict__ coef_h,
    const int* __restrict__ cnt_h, int Eh, const int* __restrict__ idx_w, const float* __restrict__ coef_w,
    const int* __restrict__ cnt_w, int Ew, int B, int in_h, int in_w, int out_h, int out_w,
    const T* __restrict__ resid = nullptr, const float* __restrict__ rscale = nullptr,
    const float* __restrict__ rbias = nullptr) {
  struct alignas(sizeof(T) * CC) Px { T e[CC]; };
  float rs[CC], rb[CC];   // dgv2_resample_tab_add_affine: resid enters as rscale[c] * resid + rbias[c]
#pragma unroll
  for (int j = 0; j < CC; ++j) {
    rs[j] = rscale ? rscale[j] : 1.f;
    rb[j] = rbias ? rbias[j] : 0.f;
  }
  const int64_t total = (int64_t)B * out_h * out_w;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int wo = (int)(t % out_w);
    const int64_t r = t / out_w;
    const int ho = (int)(r % out_h);
    const int b = (int)(r / out_h);
    const int nh = cnt_h[ho], nw = cnt_w[wo];
    const Px* xb = reinterpret_cast<const Px*>(x) + (int64_t)b * in_h * in_w;
    float acc[CC];
#pragma unroll
    for (int j = 0; j < CC; ++j) acc[j] = 0.f;
    if (Eh <= 2 && Ew <= 2) {
      // the up-2 FIRs of the output pyramid (<= 2 taps per axis): every table entry and then all four pixels are requested at
      // once -- entries past a row's count name entry 0 with coefficient 0 (same sum: the live terms in the same order) --
      // instead of a count -> index -> pixel chain of dependent loads per tap (27 us for 34 MB at level 4)
      const int e1h = Eh > 1 ? 1 : 0, e1w = Ew > 1 ? 1 : 0;
      const int ih0 = idx_h[ho * Eh], ih1 = idx_h[ho * Eh + e1h], iw0 = idx_w[wo * Ew], iw1 = idx_w[wo * Ew + e1w];
      const float ch0 = coef_h[ho * Eh], ch1 = coef_h[ho * Eh + e1h], cw0 = coef_w[wo * Ew], cw1 = coef_w[wo * Ew + e1w];
      const bool h1 = nh > 1, w1 = nw > 1;
      const Px* r0 = xb + (int64_t)ih0 * in_w;
      const Px* r1 = xb + (int64_t)(h1 ? ih1 : ih0) * in_w;
      const Px v00 = r0[iw0], v01 = r0[w1 ? iw1 : iw0], v10 = r1[iw0], v11 = r1[w1 ? iw1 : iw0];
      const bool l0 = nh > 0 && nw > 0;
#pragma unroll
      for (int j = 0; j < CC; ++j) {
        float s = 0.f;
        if (l0) s += (ch0 * cw0) * to_f32(v00.e[j]);
        if (l0 && w1) s += (ch0 * cw1) * to_f32(v01.e[j]);
        if (l0 && h1) s += (ch1 * cw0) * to_f32(v10.e[j]);
        if (l0 && h1 && w1) s += (ch1 * cw1) * to_f32(v11.e[j]);
        acc[j] = s;
      }
    } else {
      for (int a = 0; a < nh; ++a) {
        const float fa = coef_h[ho * Eh + a];
        const Px* xr = xb + (int64_t)idx_h[ho * Eh + a] * in_w;
        for (int c = 0; c < nw; ++c) {
          const float cf = fa * coef_w[wo * Ew + c];
          const Px v = xr[idx_w[wo * Ew + c]];
#pragma unroll
          for (int j = 0; j < CC; ++j) acc[j] += cf * to_f32(v.e[j]);
        }
      }
    }
    Px o;
    if (resid) {   // y = resid + resample(x): the running image of the generator's skip pyramid (dusty_v2.py:179-180)
      const Px rv = reinterpret_cast<const Px*>(resid)[t];
#pragma unroll
      for (int j = 0; j < CC; ++j) {
        const float rj = rscale ? to_f32(from_f32<T>(fmaf(to_f32(rv.e[j]), rs[j], rb[j]))) : to_f32(rv.e[j]);
        o.e[j] = from_f32<T>(rj + to_f32(from_f32<T>(acc[j])));
      }
    } else {
#pragma unroll
      for (int j = 0; j < CC; ++j) o.e[j] = from_f32<T>(acc[j]);
    }
    reinterpret_cast<Px*>(y)[t] = o;
  }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void resample_tab_kernel(
    T* __restrict__ y, const T* __restrict__ x, const int* __restrict__ idx_h, const float* __restrict__ coef_h,
    const int* __restrict__ cnt_h, int Eh, const int* __restrict__ idx_w, const float* __restrict__ coef_w,
    const int* __restrict__ cnt_w, int Ew, int B, int C, int ldx, int ldy, int in_h, int in_w, int out_h, int out_w) {
  constexpr int VN = VEC ? vec16<T>::N : 1;
  const int cvecs = (C + VN - 1) / VN;
  const int64_t total = (int64_t)B * out_h * out_w * cvecs;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(t % cvecs);
    int64_t r = t / cvecs;
    const int wo = (int)(r % out_w);
    r /= out_w;
    const int ho = (int)(r % out_h);
    const int b = (int)(r / out_h);
    const int nh = cnt_h[ho], nw = cnt_w[wo];
    const int* ih = idx_h + ho * Eh;
    const float* ch = coef_h + ho * Eh;
    const int* iw = idx_w + wo * Ew;
    const float* cw = coef_w + wo * Ew;
    const T* xb = x + (int64_t)b * in_h * in_w * ldx + cv * VN;
    float acc[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[j] = 0.f;
    for (int a = 0; a < nh; ++a) {
      const float fa = ch[a];
      const T* xr = xb + (int64_t)ih[a] * in_w * ldx;
      for (int c = 0; c < nw; ++c) {
        const float cf = fa * cw[c];
        const T* p = xr + iw[c] * ldx;
        if (VEC) {
          vec16<T> v;
          v.load(p);
#pragma unroll
          for (int j = 0; j < VN; ++j) acc[j] += cf * v.get(j);
        } else {
          acc[0] += cf * to_f32(p[0]);
        }
      }
    }
    T* q = y + (((int64_t)b * out_h + ho) * out_w + wo) * ldy + cv * VN;
    if (VEC) {
      vec16<T> o;
#pragma unroll
      for (int j = 0; j < VN; ++j) o.set(j, acc[j]);
      o.store(q);
    } else {
      q[0] = from_f32<T>(acc[0]);
    }
  }
}

// Row-streaming separable variant (the production path: 16-byte-vectorisable tensors, <= 4 taps per
// output column).  A thread owns one output column x one channel vector and walks a strip of SH output
// rows.  Its W taps (<= 4 offsets + coefficients) live in registers for the whole walk; each input row
// the strip needs is filtered along W exactly once (4 global loads) into a 4-slot fp32 ring in LDS that
// only its own thread ever reads back, so there are no barriers; the H taps come from the (wave-uniform)
// row table.  A 4x4 FIR costs ~4 global loads + 64 FMAs per output vector instead of 16 loads + 128 FMAs,
// and all integer div/mod leaves the inner loop.  Ring slot = input row & 3 with a uniform tag check, so any
// table is handled correctly; tables whose rows reuse a window of <= 4 consecutive input rows (every
// up / down / blur FIR of this model and their adjoints) never re-filter a row.
constexpr int RS_RB = 4;

// EW = taps per output column the table can hold (1..4): the horizontal pass is unrolled to exactly that many
// loads / FMA groups (an up-2 FIR has 2, a blur 4)
// ACT: the output is a gradient that continues through a fused bias + leaky-ReLU (the discriminator's conv1 ->
// FusedLeakyReLU -> blur/down chain run backwards): store acc * (ref > 0 ? 1 : alpha) * ascale (ref = the forward
// OUTPUT of that activation, laid out like y) and leave per-block column sums of the stored values in bias_partial
// [gridDim.x, C] for the bias gradient -- FusedLeakyReLUFunctionBackward (fused_act.py:22-45) without its own pass.
template <typename T, int EW, bool ACT = false>
__global__ __launch_bounds__(256) void resample_stream_kernel(
    T* __restrict__ y, const T* __restrict__ x, const int* __restrict__ idx_h, const float* __restrict__ coef_h,
    const int* __restrict__ cnt_h, int Eh, const int* __restrict__ idx_w, const float* __restrict__ coef_w,
    const int* __restrict__ cnt_w, int Ew, int B, int C, int ldx, int ldy, int in_h, int in_w, int out_h, int out_w,
    int SHA, float* __restrict__ sumsq, const T* __restrict__ ref = nullptr, float alpha = 1.f, float ascale = 1.f,
    float* __restrict__ bias_partial = nullptr, uint8_t* __restrict__ y8 = nullptr) {
#ifdef DGV2_ABLATE   // benchmarking builds only (make ABLATE=1): wrong results by design, never in the shipped library
  const int SH = SHA & 0xffff, ablate = (SHA >> 16) & 0xff;   // DGV2_RS_ABLATE
#else
  const int SH = SHA & 0xffff;
  constexpr int ablate = 0;
#endif
  const bool nt = (SHA >> 30) & 1;   // nontemporal output stores (host: output >= DGV2_NT_MIN_MB)
  __shared__ float red[16];
  float ss = 0.f;   // sum of squares of what this thread stores (sumsq != nullptr: one partial per block)
  constexpr int VN = vec16<T>::N;
  // the ring holds the W-filtered rows in the tensor's own dtype: for bf16 that halves its LDS footprint
  // (16 KB per block -> 8 resident blocks per CU instead of 5; this kernel lives on occupancy) at the price of one
  // extra bf16 rounding of the intermediate, far below the bf16 output rounding; fp32 tensors keep fp32
  __shared__ uint4 ring[RS_RB][256];
  const int tid = threadIdx.x;
  const int cvecs = C / VN;
  const int rowvecs = out_w * cvecs;
  const int colblocks = (rowvecs + 255) / 256;
  const int strips = (out_h + SH - 1) / SH;
  int bid = blockIdx.x;
  const int cb = bid % colblocks; bid /= colblocks;
  const int strip = bid % strips;
  const int b = bid / strips;
  const int col = cb * 256 + tid;
  const bool live = col < rowvecs;
  const int wo = live ? col / cvecs : 0;
  const int cv = live ? col - wo * cvecs : 0;

  const int nw = live ? cnt_w[wo] : 0;
  // taps beyond the column's count (and every tap of a dead lane) load a valid address with coefficient 0: the row
  // loads are unconditional -- no zero fill, no exec-masked branch per tap in the row loop
  int xo[EW];
  float cw[EW];
  const int xo_any = nw > 0 ? idx_w[wo * Ew] * ldx + cv * VN : 0;
#pragma unroll
  for (int c = 0; c < EW; ++c) {
    xo[c] = xo_any;
    cw[c] = 0.f;
    if (c < nw) {
      xo[c] = idx_w[wo * Ew + c] * ldx + cv * VN;
      cw[c] = coef_w[wo * Ew + c];
    }
  }
  const T* xb = x + (int64_t)b * in_h * in_w * ldx;
  T* yp = y + (int64_t)b * out_h * out_w * ldy + (int64_t)wo * ldy + cv * VN;
  const T* rp = ACT ? ref + (int64_t)b * out_h * out_w * ldy + (int64_t)wo * ldy + cv * VN : nullptr;
  float bsum[VN];   // ACT: this thread's column sums of what it stores
#pragma unroll
  for (int j = 0; j < VN; ++j) bsum[j] = 0.f;
  const int ho0 = strip * SH;
  const int ho1 = min(ho0 + SH, out_h);
  int t0 = -1, t1 = -1, t2 = -1, t3 = -1;   // ring tags (block-uniform)
  // The strip's slice of the H table (<= 64 entries, host-checked SH * Eh <= 64) lives in one VGPR per field,
  // entry e in lane e; the walk below fetches entries with v_readlane (uniform index) -- no scalar-memory
  // round trip per output row, which is what bounded this kernel before.
  const int lane = threadIdx.x & 63;
  const int nent = (ho1 - ho0) * Eh;
  const int tb_idx = lane < nent ? idx_h[ho0 * Eh + lane] : 0;
  const float tb_cf = lane < nent ? coef_h[ho0 * Eh + lane] : 0.f;
  const int tb_cnt = lane < ho1 - ho0 ? cnt_h[ho0 + lane] : 0;
  int rlast = -1;                           // last input row the strip reads: bound of the prefetch
  for (int i = 0; i < ho1 - ho0; ++i) {
    const int n = __builtin_amdgcn_readlane(tb_cnt, i);
    for (int a = 0; a < n; ++a) rlast = max(rlast, __builtin_amdgcn_readlane(tb_idx, i * Eh + a));
  }
  // input rows are consumed in (mostly) increasing order: after filtering row r the loads of row r + 1 are
  // issued at once and stay in flight behind this row's FMAs / LDS traffic / output store
  int pr = -1;
  vec16<T> pv[EW];
  auto issue = [&](int r) {
    const T* xr = xb + (int64_t)r * in_w * ldx;
#pragma unroll
    for (int c = 0; c < EW; ++c) pv[c].load(xr + xo[c]);
    pr = r;
  };
  for (int ho = ho0; ho < ho1; ++ho) {
    const int n = __builtin_amdgcn_readlane(tb_cnt, ho - ho0);
    float acc[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[j] = 0.f;
    for (int a = 0; a < n; ++a) {
      const int e = (ho - ho0) * Eh + a;
      const int r = __builtin_amdgcn_readlane(tb_idx, e);
      const float fa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tb_cf), e));
      const int s = r & (RS_RB - 1);
      const int tag = s == 0 ? t0 : (s == 1 ? t1 : (s == 2 ? t2 : t3));
      if (tag != r) {
        if (pr != r && !(ablate & 2)) issue(r);
        float h[VN];
#pragma unroll
        for (int j = 0; j < VN; ++j) h[j] = 0.f;
#pragma unroll
        for (int c = 0; c < EW; ++c)
#pragma unroll
          for (int j = 0; j < VN; ++j) h[j] += cw[c] * pv[c].get(j);
        if (r + 1 <= rlast && !(ablate & 2)) issue(r + 1); else pr = -1;
        {
          vec16<T> hv;
#pragma unroll
          for (int j = 0; j < VN; ++j) hv.set(j, h[j]);
          ring[s][tid] = hv.raw;
        }
        t0 = s == 0 ? r : t0; t1 = s == 1 ? r : t1; t2 = s == 2 ? r : t2; t3 = s == 3 ? r : t3;
      }
      {
        vec16<T> hv;
        hv.raw = ring[s][tid];
#pragma unroll
        for (int j = 0; j < VN; ++j) acc[j] += fa * hv.get(j);
      }
    }
    if (live && !((ablate & 1) && acc[0] != 12345.678f)) {
      vec16<T> o;
      if constexpr (ACT) {
        vec16<T> f;
        f.load(rp + (int64_t)ho * out_w * ldy);
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          o.set(j, (f.get(j) > 0.f ? acc[j] : acc[j] * alpha) * ascale);
          bsum[j] += o.get(j);   // the reference sums the rounded gradient
        }
      } else {
#pragma unroll
        for (int j = 0; j < VN; ++j) o.set(j, acc[j]);
      }
      if (y) {   // y == nullptr: statistic only (sum of squares of the result) / e4m3 output only
        if (nt) o.store_nt(yp + (int64_t)ho * out_w * ldy);
        else o.store(yp + (int64_t)ho * out_w * ldy);
      }
      if constexpr (!ACT && sizeof(T) == 2) {
        if (y8) {   // the result as e4m3 (unit scale), same [B, out_h, out_w, ldy] element layout: 8 bytes per thread
          float f[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = acc[j];
          *reinterpret_cast<uint2*>(y8 + ((int64_t)b * out_h + ho) * out_w * ldy + (int64_t)wo * ldy + cv * VN) = pack_fp8x8(f);
        }
      }
      if (sumsq) {
#pragma unroll
        for (int j = 0; j < VN; ++j) ss = fmaf(o.get(j), o.get(j), ss);
      }
    }
  }
  if (sumsq) {
    const float s = block_sum(ss, red);
    if (tid == 0) sumsq[blockIdx.x] = s;
  }
  if constexpr (ACT) {
    // threads with the same channel vector are cvecs apart (cvecs divides 256, host-checked): fold them through LDS,
    // one thread per channel writes this block's slot
    float* fold = reinterpret_cast<float*>(&ring[0][0]);   // the ring is dead by now: 256 * VN floats fit
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VN; ++j) fold[tid * VN + j] = live ? bsum[j] : 0.f;
    __syncthreads();
    const int off = (cb * 256) % cvecs;   // channel vector of thread 0 in this column block
    for (int c = tid; c < C; c += 256) {
      const int v = c / VN, j = c - v * VN;
      // threads t with (off + t) % cvecs == v
      float s2 = 0.f;
      for (int t = (v - off + cvecs) % cvecs; t < 256; t += cvecs) s2 += fold[t * VN + j];
      bias_partial[(int64_t)blockIdx.x * C + c] = s2;
    }
  }
}

template <typename T>
void rs_launch(int Ew, int blocks, hipStream_t st, T* y, const T* x, const int* idx_h, const float* coef_h,
               const int* cnt_h, int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int B, int C, int ldx,
               int ldy, int in_h, int in_w, int out_h, int out_w, int sha, float* sumsq) {
  switch (Ew) {
    case 1: resample_stream_kernel<T, 1><<<blocks, 256, 0, st>>>(y, x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w, sha, sumsq); break;
    case 2: resample_stream_kernel<T, 2><<<blocks, 256, 0, st>>>(y, x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w, sha, sumsq); break;
    case 3: resample_stream_kernel<T, 3><<<blocks, 256, 0, st>>>(y, x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w, sha, sumsq); break;
    default: resample_stream_kernel<T, 4><<<blocks, 256, 0, st>>>(y, x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w, sha, sumsq); break;
  }
}

template <typename T>
void rs_launch_act(int Ew, int blocks, hipStream_t st, T* y, const T* x, const int* idx_h, const float* coef_h,
                   const int* cnt_h, int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int B, int C,
                   int ldx, int ldy, int in_h, int in_w, int out_h, int out_w, int sha, const T* ref, float alpha,
                   float ascale, float* partial) {
  switch (Ew) {
    case 1: resample_stream_kernel<T, 1, true><<<blocks, 256, 0, st>>>(y, x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w, sha, nullptr, ref, alpha, ascale, partial); break;
    case 2: resample_stream_kernel<T, 2, true><<<blocks, 256, 0, st>>>(y, x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w, sha, nullptr, ref, alpha, ascale, partial); break;
    case 3: resample_stream_kernel<T, 3, true><<<blocks, 256, 0, st>>>(y, x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w, sha, nullptr, ref, alpha, ascale, partial); break;
    default: resample_stream_kernel<T, 4, true><<<blocks, 256, 0, st>>>(y, x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w, sha, nullptr, ref, alpha, ascale, partial); break;
  }
}

// gb[c] = sum_blk partial[blk][c] (one wave per channel)
__global__ __launch_bounds__(256) void rs_bias_reduce_kernel(float* __restrict__ gb, const float* __restrict__ partial,
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

// Resampling (normally the ADJOINT tables of a blur/down) followed by the backward of a fused bias + leaky-ReLU:
//   y = R(x) * (ref > 0 ? 1 : alpha) * scale,   gb[c] = sum of y over everything but the channel   (fp32 [C])
// ref: forward output of the activation, [B, out_h, out_w, C] contiguous like y (ldy == C).  scratch: fp32
// [>= blocks * C] with blocks = *blocks_needed reported when scratch is NULL (query call: nothing is launched).
// Returns DGV2_ENOTSUP when the streaming kernel does not cover the geometry (callers run dgv2_resample_tab and
// dgv2_bias_act_bwd).  replaces: Resample adjoint (common.py:105-135) + FusedLeakyReLUFunctionBackward
// (fused_act.py:22-45) of the discriminator's conv1 -> activation -> blur/down chain (dusty_v2.py:325-345).
extern "C" int dgv2_resample_tab_actbwd(void* y, float* gb, float* scratch, int64_t scratch_elems, int64_t* blocks_needed,
                                        const void* x, const void* ref, const int* idx_h, const float* coef_h,
                                        const int* cnt_h, int Eh, const int* idx_w, const float* coef_w,
                                        const int* cnt_w, int Ew, int B, int C, int in_h, int in_w, int out_h,
                                        int out_w, float alpha, float scale, int dtype, void* stream) {
  if (B <= 0 || C <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || Eh <= 0 || Ew <= 0) return DGV2_EINVAL;
  const int vn = dtype == DGV2_BF16 ? 8 : (dtype == DGV2_F32 ? 4 : 0);
  if (!vn) return DGV2_EINVAL;
  if (C % vn || 256 % (C / vn) || Ew > 4 || Eh > 64) return DGV2_ENOTSUP;
  int SH = out_h >= 32 ? 16 : (out_h >= 8 ? 8 : out_h);
  while (SH > 1 && SH * Eh > 64) SH >>= 1;
  const int64_t blocks = (int64_t)B * ((out_h + SH - 1) / SH) * (((int64_t)out_w * (C / vn) + 255) / 256);
  if (blocks >= (1LL << 31)) return DGV2_ENOTSUP;
  if (blocks_needed) *blocks_needed = blocks;
  if (!scratch) return blocks_needed ? 0 : DGV2_EINVAL;
  if (!y || !gb || !x || !ref || !idx_h || !coef_h || !cnt_h || !idx_w || !coef_w || !cnt_w) return DGV2_EINVAL;
  if (scratch_elems < blocks * C || !aligned16(x) || !aligned16(y) || !aligned16(ref)) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  DGV2_DISPATCH_DTYPE(dtype, {
    rs_launch_act<T>(Ew, (int)blocks, st, (T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, B, C, C, C,
                     in_h, in_w, out_h, out_w, SH, (const T*)ref, alpha, scale, scratch);
  });
  rs_bias_reduce_kernel<<<C, 256, 0, st>>>(gb, scratch, (int)blocks, C);
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_resample_tab(void* y, const void* x, const int* idx_h, const float* coef_h, const int* cnt_h,
                                 int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int Ew, int B,
                                 int C, int ldx, int ldy, int in_h, int in_w, int out_h, int out_w, int dtype,
                                 void* stream) {
  return dgv2_resample_tab_sq(y, x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h,
                              out_w, dtype, nullptr, 0, nullptr, stream);
}

// y = resid + resample(x) for packed few-channel images (C in {1, 2, 4}, ld = C): the generator's output pyramid,
// `o + self.resample(skip)` of SynthesisBlock.forward (dusty_v2.py:179-180), in the resampler's own store -- same bits
// as the two-launch form (the resampled value is rounded to the storage type before the add).  DGV2_ENOTSUP otherwise.
extern "C" int dgv2_resample_tab_add(void* y, const void* x, const void* resid, const int* idx_h, const float* coef_h,
                                     const int* cnt_h, int Eh, const int* idx_w, const float* coef_w, const int* cnt_w,
                                     int Ew, int B, int C, int in_h, int in_w, int out_h, int out_w, int dtype,
                                     void* stream) {
  return dgv2_resample_tab_add_affine(y, x, resid, nullptr, nullptr, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, C,
                                      in_h, in_w, out_h, out_w, dtype, stream);
}

// ... with resid entering as rscale[c] * resid + rbias[c] (fp32 [C] each, both or neither): the level's head output
// c * (heads' contraction) + bias formed in the same store that adds the up-sampled running image, when the contraction
// itself came out of conv2's epilogue (dgv2_modconv_pe_fwd_head).
extern "C" int dgv2_resample_tab_add_affine(void* y, const void* x, const void* resid, const float* rscale,
                                            const float* rbias, const int* idx_h, const float* coef_h, const int* cnt_h,
                                            int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int Ew, int B,
                                            int C, int in_h, int in_w, int out_h, int out_w, int dtype, void* stream) {
  if ((!rscale) != (!rbias)) return DGV2_EINVAL;
  if (!y || !x || !resid || !idx_h || !coef_h || !cnt_h || !idx_w || !coef_w || !cnt_w) return DGV2_EINVAL;
  if (B <= 0 || C <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || Eh <= 0 || Ew <= 0) return DGV2_EINVAL;
  if (!(C == 1 || C == 2 || C == 4)) return DGV2_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  DGV2_DISPATCH_DTYPE(dtype, {
    const uintptr_t al = sizeof(T) * C;
    if (reinterpret_cast<uintptr_t>(x) % al || reinterpret_cast<uintptr_t>(y) % al || reinterpret_cast<uintptr_t>(resid) % al)
      return DGV2_ENOTSUP;
    const int g2 = grid_for((int64_t)B * out_h * out_w, 256, 256 * 64);
    if (C == 1) resample_tab_smallc_kernel<T, 1><<<g2, 256, 0, st>>>((T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, in_h, in_w, out_h, out_w, (const T*)resid, rscale, rbias);
    else if (C == 2) resample_tab_smallc_kernel<T, 2><<<g2, 256, 0, st>>>((T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, in_h, in_w, out_h, out_w, (const T*)resid, rscale, rbias);
    else resample_tab_smallc_kernel<T, 4><<<g2, 256, 0, st>>>((T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, in_h, in_w, out_h, out_w, (const T*)resid, rscale, rbias);
  });
  DGV2_RETURN_LAST();
}

// dgv2_resample_tab with the result stored as e4m3 (OCP fp8, unit scale, saturating at +-448) instead of bf16:
// y8 [B,out_h,out_w,C] bytes.  x bf16, C % 8 == 0, Ew <= 4, Eh <= 64 (the streaming kernel); DGV2_ENOTSUP otherwise.
// The FIR arithmetic is the bf16 kernel's (fp32 accumulation, W pass rounded to bf16 in the ring); only the final
// store differs.  fp8.hip says which tensors are kept like this.
extern "C" int dgv2_resample_tab_q8(void* y8, const void* x, const int* idx_h, const float* coef_h, const int* cnt_h,
                                    int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int Ew, int B,
                                    int C, int in_h, int in_w, int out_h, int out_w, void* stream) {
  if (!y8 || !x || !idx_h || !coef_h || !cnt_h || !idx_w || !coef_w || !cnt_w) return DGV2_EINVAL;
  if (B <= 0 || C <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || Eh <= 0 || Ew <= 0) return DGV2_EINVAL;
  if ((C & 7) || Ew > 4 || Eh > 64 || !aligned16(x) || (reinterpret_cast<uintptr_t>(y8) & 7)) return DGV2_ENOTSUP;
  typedef bf16_t T;
  int SH = out_h >= 32 ? 16 : (out_h >= 8 ? 8 : out_h);
  while (SH > 1 && SH * Eh > 64) SH >>= 1;
  const int64_t blocks = (int64_t)B * ((out_h + SH - 1) / SH) * (((int64_t)out_w * (C / 8) + 255) / 256);
  if (blocks >= (1LL << 31)) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  uint8_t* q = (uint8_t*)y8;
#define DGV2_RS_Q8(EW)                                                                                                   \
  resample_stream_kernel<T, EW><<<(int)blocks, 256, 0, st>>>((T*)nullptr, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w,   \
                                                            coef_w, cnt_w, Ew, B, C, C, C, in_h, in_w, out_h, out_w, SH,  \
                                                            nullptr, nullptr, 1.f, 1.f, nullptr, q)
  switch (Ew) {
    case 1: DGV2_RS_Q8(1); break;
    case 2: DGV2_RS_Q8(2); break;
    case 3: DGV2_RS_Q8(3); break;
    default: DGV2_RS_Q8(4); break;
  }
#undef DGV2_RS_Q8
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_resample_tab_sq(void* y, const void* x, const int* idx_h, const float* coef_h, const int* cnt_h,
                                    int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int Ew, int B,
                                    int C, int ldx, int ldy, int in_h, int in_w, int out_h, int out_w, int dtype,
                                    float* sumsq, int sumsq_cap, int* sumsq_used, void* stream) {
  if (sumsq_used) *sumsq_used = 0;
  // y == NULL: only the sum-of-squares partials of the (never stored) result are produced -- the input statistic of a
  // modulated conv whose up-sampling was commuted past the contraction (modconv_up.hip); needs the streaming kernel
  const bool stat_only = !y;
  if ((stat_only && !(sumsq && sumsq_used)) || !x || !idx_h || !coef_h || !cnt_h || !idx_w || !coef_w || !cnt_w) return DGV2_EINVAL;
  if (B <= 0 || C <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || Eh <= 0 || Ew <= 0) return DGV2_EINVAL;
  if (ldx < C || ldy < C) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  DGV2_DISPATCH_DTYPE(dtype, {
    constexpr int VN = vec16<T>::N;
    const bool vec = (C % VN == 0) && (ldx % VN == 0) && (ldy % VN == 0) && aligned16(x) && aligned16(y);
    if (stat_only && !(vec && Ew <= 4 && Eh <= 64)) return DGV2_ENOTSUP;
    const int64_t total = (int64_t)B * out_h * out_w * ((C + (vec ? VN : 1) - 1) / (vec ? VN : 1));
    const int grid = grid_for(total, 256, 256 * 64);
    static const bool no_stream = getenv("DGV2_NO_RSTREAM") != nullptr;   // A/B switch for benchmarking
#ifdef DGV2_ABLATE
    static const int rs_ablate = getenv("DGV2_RS_ABLATE") ? atoi(getenv("DGV2_RS_ABLATE")) : 0;
#else
    constexpr int rs_ablate = 0;
#endif
    if (vec && Ew <= 4 && Eh <= 64 && (!no_stream || stat_only)) {
      static const int sh_env = getenv("DGV2_RS_SH") ? atoi(getenv("DGV2_RS_SH")) : 0;   // experiments
      int SH = out_h >= 32 ? 16 : (out_h >= 8 ? 8 : out_h);
      if (sh_env > 0 && sh_env < SH) SH = sh_env;
      while (SH > 1 && SH * Eh > 64) SH >>= 1;   // the strip's H-table slice must fit one lane-indexed register
      const int64_t blocks = (int64_t)B * ((out_h + SH - 1) / SH) * (((int64_t)out_w * (C / VN) + 255) / 256);
      if (blocks >= (1LL << 31)) return DGV2_EINVAL;
      float* sq = (sumsq && sumsq_used && blocks <= sumsq_cap) ? sumsq : nullptr;   // one partial per block
      if (stat_only && !sq) return DGV2_ENOTSUP;
      if (sq) *sumsq_used = (int)blocks;
      rs_launch<T>(Ew, (int)blocks, st, (T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, B, C, ldx, ldy, in_h,
                   in_w, out_h, out_w,
                   SH | (rs_ablate << 16) | ((y && nt_output((int64_t)B * out_h * out_w * C * sizeof(T))) ? (1 << 30) : 0), sq);
    } else if (!vec && (C == 1 || C == 2 || C == 4) && ldx == C && ldy == C &&
               (reinterpret_cast<uintptr_t>(x) % (sizeof(T) * C)) == 0 && (reinterpret_cast<uintptr_t>(y) % (sizeof(T) * C)) == 0) {
      const int g2 = grid_for((int64_t)B * out_h * out_w, 256, 256 * 64);
      if (C == 1) resample_tab_smallc_kernel<T, 1><<<g2, 256, 0, st>>>((T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, in_h, in_w, out_h, out_w);
      else if (C == 2) resample_tab_smallc_kernel<T, 2><<<g2, 256, 0, st>>>((T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, in_h, in_w, out_h, out_w);
      else resample_tab_smallc_kernel<T, 4><<<g2, 256, 0, st>>>((T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w, cnt_w, Ew, B, in_h, in_w, out_h, out_w);
    } else if (vec)
      resample_tab_kernel<T, true><<<grid, 256, 0, st>>>((T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w,
                                                        cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w);
    else
      resample_tab_kernel<T, false><<<grid, 256, 0, st>>>((T*)y, (const T*)x, idx_h, coef_h, cnt_h, Eh, idx_w, coef_w,
                                                         cnt_w, Ew, B, C, ldx, ldy, in_h, in_w, out_h, out_w);
  });
  DGV2_RETURN_LAST();
}
