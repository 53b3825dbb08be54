// MFMA tile engines shared by the modulated-conv contraction (gemm.hip) and the ring-padded
// dense convolutions of the discriminator (conv.hip).  gfx950 only.
//
// Both engines treat operands as 16-byte K-chunks so that ONE structure serves bf16
// (v_mfma_f32_16x16x32_bf16, 8 elements per chunk) and fp32 (4 x v_mfma_f32_16x16x4_f32, exact
// fp32 fma chains -- the parity mode).  MFMA only requires A and B to agree on which k a lane
// holds, so the fp32 path simply feeds element j of a lane's chunk to step j.
//
//   NN engine:  D[m][n] = sum_k A[m][k] * B[n][k]      (both operands K-contiguous)
//               A rows = output channels (16 per fragment), B rows = pixels.
//   TN engine:  D[m][j] = sum_k A[k][m] * B[k][j]      (both operands K-strided: weight gradients,
//               K = pixels).  bf16 fragments come from ds_read_b64_tr_b16 (hardware transpose).
//
// Loaders are functors returning one 16-byte chunk (zero-filled out of range); the same engines
// run dense rows (1x1 modulated conv) and implicit im2col gathers (3x3 ring convs).
#pragma once
#include "common.h"

template <typename T> struct Mfma16;

template <> struct Mfma16<bf16_t> {
  __device__ static __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; bf16x8 v; } ua, ub;
    ua.u = a;
    ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc, 0, 0, 0);
  }
};

// e4m3 operands: a 16-byte chunk is 16 elements = two K = 32 steps of v_mfma_f32_16x16x32_fp8_fp8 (8 bytes per lane
// each).  Which k a lane's bytes stand for is the same permutation on both operands, so the low / high halves of the
// chunks pair up and the two steps together contract the 64 channels of a 64-byte K-chunk.
template <> struct Mfma16<fp8_t> {
  __device__ static __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    const long a0 = (long)(((unsigned long)a.y << 32) | a.x), a1 = (long)(((unsigned long)a.w << 32) | a.z);
    const long b0 = (long)(((unsigned long)b.y << 32) | b.x), b1 = (long)(((unsigned long)b.w << 32) | b.z);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc, 0, 0, 0);
  }
};

template <> struct Mfma16<float> {
  __device__ static __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; float f[4]; } ua, ub;
    ua.u = a;
    ub.u = b;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.f[j], ub.f[j], acc, 0, 0, 0);
  }
};

// MFMA accumulating IN PLACE, as inline asm.  The unrolled-tap kernels issue MFMAs under wave-uniform branches (dead
// taps); with the builtin the compiler gives every conditional MFMA a fresh destination and copies whole accumulator
// sets around the branches (a second 64-register set, then spills).  The asm's "+v" keeps one set.  The compiler does
// not see inside: fragments arrive through ordinary ds_reads (it places their lgkmcnt waits in front of the asm),
// dependent MFMAs on the same accumulator are interlocked by the hardware, and the one software hazard -- an MFMA
// result read by a VALU instruction -- is covered by mfma_drain() in front of the epilogue.
template <typename T> struct MfmaAsm;
template <> struct MfmaAsm<bf16_t> {
  __device__ static __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; bf16x8 v; } ua, ub;
    ua.u = a;
    ub.u = b;
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ua.v), "v"(ub.v));
  }
  // the LAST MFMA of a group that compiler-generated code may follow (a branch condition, a merged register): its five
  // wait states travel inside the statement, so nothing can be scheduled between the MFMA and them
  __device__ static __forceinline__ void run_pad(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; bf16x8 v; } ua, ub;
    ua.u = a;
    ub.u = b;
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 4" : "+v"(acc) : "v"(ua.v), "v"(ub.v));
  }
};
template <> struct MfmaAsm<fp8_t> {   // see Mfma16<fp8_t>
  __device__ static __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(acc) : "v"(a0), "v"(b0));
    asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(acc) : "v"(a1), "v"(b1));
  }
  __device__ static __forceinline__ void run_pad(f32x4& acc, const uint4& a, const uint4& b) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
    asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(acc) : "v"(a0), "v"(b0));
    asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0\n\ts_nop 4" : "+v"(acc) : "v"(a1), "v"(b1));
  }
};
template <> struct MfmaAsm<float> {
  __device__ static __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a.x), "v"(b.x));
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a.y), "v"(b.y));
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a.z), "v"(b.z));
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a.w), "v"(b.w));
  }
  __device__ static __forceinline__ void run_pad(f32x4& acc, const uint4& a, const uint4& b) {   // 8 passes: eleven wait states
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a.x), "v"(b.x));
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a.y), "v"(b.y));
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a.z), "v"(b.z));
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 10" : "+v"(acc) : "v"(a.w), "v"(b.w));
  }
};
// every MFMA issued so far has written its accumulator when this returns (the longest of the shapes used here takes
// 8 passes = 32 cycles; the fences keep the scheduler from moving accumulator reads in front of the wait)
__device__ __forceinline__ void mfma_drain() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ----------------------------------------------------------------------------------------------
// Loaders (NN engine): chunk (row, kchunk) of a [rows, K] K-contiguous operand.
// ----------------------------------------------------------------------------------------------
template <typename T> struct DenseRowLoader {
  const T* base;
  int64_t batch_stride;  // elements
  int ld, rows, K;
  bool vec;
  __device__ __forceinline__ uint4 load(int batch, int row, int kchunk) const {
    constexpr int CE = 16 / sizeof(T);
    const int k = kchunk * CE;
    if (row >= rows || k >= K) return make_uint4(0, 0, 0, 0);
    const T* p = base + batch * batch_stride + (int64_t)row * ld + k;
    if (vec && k + CE <= K) return *reinterpret_cast<const uint4*>(p);
    vec16<T> v;
    v.raw = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < CE; ++j)
      if (k + j < K) v.e[j] = p[j];
    return v.raw;
  }
};

// Two-part K axis for the level-input conv of the generator: k < Ka comes from the per-sample
// activation `a` [batch][rows][lda]; Ka <= k < Ka+Ks from a batch-SHARED tensor `s` [rows][lds]
// (the positional encoding of the unshifted angle grid, see gemm.hip).  Ka % CE == 0 required.
template <typename T> struct ConcatRowLoader {
  const T* a;
  int64_t a_batch_stride;
  int lda, Ka;
  const T* s;
  int lds_, Ks;
  int rows;
  __device__ __forceinline__ uint4 load(int batch, int row, int kchunk) const {
    constexpr int CE = 16 / sizeof(T);
    const int k = kchunk * CE;
    if (row >= rows) return make_uint4(0, 0, 0, 0);
    if (k < Ka) return *reinterpret_cast<const uint4*>(a + batch * a_batch_stride + (int64_t)row * lda + k);
    const int k2 = k - Ka;
    if (k2 + CE <= Ks) return *reinterpret_cast<const uint4*>(s + (int64_t)row * lds_ + k2);
    return make_uint4(0, 0, 0, 0);
  }
};

struct ConvGeom {
  int B, H, W, C, O, Ho, Wo, kh, kw, stride, pad, ring;
};

__device__ __forceinline__ int conv_src_h(const ConvGeom& g, int hi) { return hi < 0 ? 0 : (hi >= g.H ? g.H - 1 : hi); }
__device__ __forceinline__ int conv_src_w(const ConvGeom& g, int wi) {
  if (g.ring) return floormod(wi, g.W);
  return wi < 0 ? 0 : (wi >= g.W ? g.W - 1 : wi);
}

// Forward im2col: row = output pixel (b, ho, wo); k = (ky, kx, c), c fastest.
template <typename T> struct Im2colFwdLoader {
  const T* x;
  ConvGeom g;
  int npix, K;
  bool vec;  // C % CE == 0 and x 16-byte aligned
  __device__ __forceinline__ const T* src(int b, int ho, int wo, int kk, int& c) const {
    const int tap = kk / g.C;
    c = kk - tap * g.C;
    const int ky = tap / g.kw, kx = tap - ky * g.kw;
    const int hi = conv_src_h(g, ho * g.stride + ky - g.pad);
    const int wi = conv_src_w(g, wo * g.stride + kx - g.pad);
    return x + (((int64_t)b * g.H + hi) * g.W + wi) * g.C;
  }
  __device__ __forceinline__ uint4 load(int, int row, int kchunk) const {
    constexpr int CE = 16 / sizeof(T);
    const int k = kchunk * CE;
    if (row >= npix || k >= K) return make_uint4(0, 0, 0, 0);
    const int wo = row % g.Wo;
    const int t = row / g.Wo;
    const int ho = t % g.Ho;
    const int b = t / g.Ho;
    int c;
    if (vec) {
      const T* p = src(b, ho, wo, k, c);
      return *reinterpret_cast<const uint4*>(p + c);
    }
    vec16<T> v;
    v.raw = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < CE; ++j)
      if (k + j < K) {
        const T* p = src(b, ho, wo, k + j, c);
        v.e[j] = p[c];
      }
    return v.raw;
  }
};

// Data-gradient gather in the PADDED input domain: row = (b, hp, wp) over (H+2pad) x (W+2pad);
// k = (ky, kx, o), o fastest; source gy [B,Ho,Wo,O].  gxp[hp] = sum_ky gy[(hp-ky)/s] w[ky].
template <typename T> struct Im2colDgradLoader {
  const T* gy;
  ConvGeom g;
  int Hp, Wp, npix, K;
  bool vec;  // O % CE == 0
  __device__ __forceinline__ const T* src(int b, int hp, int wp, int kk, int& o) const {
    const int tap = kk / g.O;
    o = kk - tap * g.O;
    const int ky = tap / g.kw, kx = tap - ky * g.kw;
    const int nh = hp - ky, nw = wp - kx;
    if (nh < 0 || nw < 0 || nh % g.stride != 0 || nw % g.stride != 0) return nullptr;
    const int ho = nh / g.stride, wo = nw / g.stride;
    if (ho >= g.Ho || wo >= g.Wo) return nullptr;
    return gy + (((int64_t)b * g.Ho + ho) * g.Wo + wo) * g.O;
  }
  __device__ __forceinline__ uint4 load(int, int row, int kchunk) const {
    constexpr int CE = 16 / sizeof(T);
    const int k = kchunk * CE;
    if (row >= npix || k >= K) return make_uint4(0, 0, 0, 0);
    const int wp = row % Wp;
    const int t = row / Wp;
    const int hp = t % Hp;
    const int b = t / Hp;
    int o;
    if (vec) {
      const T* p = src(b, hp, wp, k, o);
      return p ? *reinterpret_cast<const uint4*>(p + o) : make_uint4(0, 0, 0, 0);
    }
    vec16<T> v;
    v.raw = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < CE; ++j)
      if (k + j < K) {
        const T* p = src(b, hp, wp, k + j, o);
        if (p) v.e[j] = p[o];
      }
    return v.raw;
  }
};

// ----------------------------------------------------------------------------------------------
// NN engine.  Block = 256 threads = 4 waves; tile = TO output channels x 128 pixels; wave w owns
// pixels [32w, 32w+32).  LDS rows are 64 bytes (one K-step), XOR-swizzled by (row>>2)&3 so the
// ds_read_b128 fragment reads of 16 consecutive rows hit 64 distinct banks.
// Epilogue functor: epi(batch, m /*first of 4 consecutive channels*/, n /*pixel*/, f32x4 acc).
// ----------------------------------------------------------------------------------------------
template <typename T, int TO, class ALoad, class BLoad, class Epi, bool BATCH_FAST = false>
__global__ __launch_bounds__(256) void gemm_nn_kernel(ALoad al, BLoad bl, Epi epi, int K) {
  constexpr int TP = 128;
  constexpr int MF = TO / 16;
  constexpr int NF = 2;
  constexpr int CE = 16 / sizeof(T);
  constexpr int ACH = (TO * 4 + 255) / 256;  // A chunks per thread
  __shared__ __attribute__((aligned(16))) uint4 ldsA[2][TO * 4];
  __shared__ __attribute__((aligned(16))) uint4 ldsB[2][TP * 4];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lc = lane >> 4;
  // BATCH_FAST: consecutive blocks are the SAME pixel tile of different samples, so a batch-shared
  // B operand (ConcatRowLoader) is fetched from HBM once and served from L2 to the other samples.
  const int batch = BATCH_FAST ? blockIdx.x : blockIdx.z;
  const int m0 = blockIdx.y * TO;
  const int n0 = (BATCH_FAST ? blockIdx.z : blockIdx.x) * TP;
  const int nk = (K + 4 * CE - 1) / (4 * CE);

  uint4 ra[ACH], rb[2];
  auto swz = [](int row, int ch) { return row * 4 + (ch ^ ((row >> 2) & 3)); };
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + i * 256;
      rb[i] = bl.load(batch, n0 + (id >> 2), kt * 4 + (id & 3));
    }
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int id = tid + i * 256;
      if (id < TO * 4) ra[i] = al.load(batch, m0 + (id >> 2), kt * 4 + (id & 3));
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + i * 256;
      ldsB[buf][swz(id >> 2, id & 3)] = rb[i];
    }
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int id = tid + i * 256;
      if (id < TO * 4) ldsA[buf][swz(id >> 2, id & 3)] = ra[i];
    }
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
    uint4 a[MF], b[NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) a[mf] = ldsA[cur][swz(mf * 16 + lr, lc)];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) b[nf] = ldsB[cur][swz(wave * 32 + nf * 16 + lr, lc)];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) Mfma16<T>::run(acc[mf][nf], a[mf], b[nf]);
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  // D layout of the 16x16 MFMA: column (n) = lane & 15, rows (m) = 4*(lane>>4) + r.
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
      epi(batch, m0 + mf * 16 + lc * 4, n0 + wave * 32 + nf * 16 + lr, acc[mf][nf]);
  if (epi.sumsq) {   // one partial sum of squares of the stored outputs per block (see dgv2_bmm_nn_sq)
    __shared__ float red[16];
    const float s = block_sum(epi.ss, red);
    if (tid == 0) epi.sumsq[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
  }
}

// Store epilogue: Y[batch][n][m..m+3] (channels-last), optional fp32 output for bf16 inputs, and
// an optional fused "+ bias[m], leaky-ReLU, * scale" (the FusedLeakyReLU that follows every trunk
// conv: gans/models/ops/fused_act/fused_bias_act_kernel.cu:19-65 case act=3, grad=0).
template <typename TY> struct StoreEpilogue {
  TY* y;
  int64_t batch_stride;
  int ld, M, N;
  bool vec;
  const float* bias;  // fp32 [M] or nullptr
  int act;            // 0 = none, 3 = leaky ReLU
  float alpha, scale;
  float* sumsq;       // optional: per-block partial sums of squares of the stored (rounded) outputs
  float ss;           // this thread's running sum (kernel-private state, initialise to 0)
  const float* row_scale;  // optional fp32 [M]: y = act(acc * row_scale[m] + bias[m])
  const TY* resid;         // optional, same layout as y: added to the result (after the activation)
  __device__ __forceinline__ void operator()(int batch, int m, int n, f32x4 acc) {
    if (n >= N || m >= M) return;
    if (row_scale) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (m + r < M) acc[r] *= row_scale[m + r];
    }
    if (bias || act) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[r];
        if (bias && m + r < M) v += bias[m + r];
        if (act == 3) v = (v > 0.f ? v : v * alpha) * scale;
        acc[r] = v;
      }
    }
    TY* p = y + batch * batch_stride + (int64_t)n * ld + m;
    if (resid) {
      const TY* rp = resid + batch * batch_stride + (int64_t)n * ld + m;
      if (vec && m + 3 < M) {   // one 8- or 16-byte load (resid shares y's layout and alignment)
        if constexpr (sizeof(TY) == 4) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += rv[r];
        } else {
          union { uint2 u; bf16_t e[4]; } rv;
          rv.u = *reinterpret_cast<const uint2*>(rp);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += (float)rv.e[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (m + r < M) acc[r] += to_f32(rp[r]);
      }
    }
    if (sumsq) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (m + r < M) {
          const float v = to_f32(from_f32<TY>(acc[r]));
          ss = fmaf(v, v, ss);
        }
    }
    if (vec && m + 3 < M) {
      if constexpr (sizeof(TY) == 4) {
        *reinterpret_cast<f32x4*>(p) = acc;
      } else {
        union { uint2 u; bf16_t e[4]; } o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o.e[r] = (bf16_t)acc[r];
        *reinterpret_cast<uint2*>(p) = o.u;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (m + r < M) p[r] = from_f32<TY>(acc[r]);
    }
  }
};

// ----------------------------------------------------------------------------------------------
// Loaders (TN engine): chunk (k, colchunk) = CE consecutive columns of row k of a [K, cols] operand.
// ----------------------------------------------------------------------------------------------
template <typename T> struct DenseKLoader {
  const T* base;
  int64_t batch_stride;
  int ld, cols;
  bool vec;
  __device__ __forceinline__ uint4 load(int batch, int64_t k, int64_t K, int colchunk) const {
    constexpr int CE = 16 / sizeof(T);
    const int col = colchunk * CE;
    if (k >= K || col >= cols) return make_uint4(0, 0, 0, 0);
    const T* p = base + batch * batch_stride + k * ld + col;
    if (vec && col + CE <= cols) return *reinterpret_cast<const uint4*>(p);
    vec16<T> v;
    v.raw = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < CE; ++j)
      if (col + j < cols) v.e[j] = p[j];
    return v.raw;
  }
};

// Column-concatenated operand for the TN engine: columns < Ca from the per-sample `a`, the rest from
// the batch-shared `s` (positional encoding).  Ca % CE == 0 required.
template <typename T> struct ConcatKLoader {
  const T* a;
  int64_t a_batch_stride;
  int lda, Ca;
  const T* s;
  int lds_, Cs;
  __device__ __forceinline__ uint4 load(int batch, int64_t k, int64_t K, int colchunk) const {
    constexpr int CE = 16 / sizeof(T);
    const int col = colchunk * CE;
    if (k >= K) return make_uint4(0, 0, 0, 0);
    if (col < Ca) return *reinterpret_cast<const uint4*>(a + batch * a_batch_stride + k * lda + col);
    const int c2 = col - Ca;
    if (c2 + CE <= Cs) return *reinterpret_cast<const uint4*>(s + k * lds_ + c2);
    return make_uint4(0, 0, 0, 0);
  }
};

// Weight-gradient gather: k = output pixel (b, ho, wo); column = (ky, kx, c) of the padded input.
template <typename T> struct Im2colWgradLoader {
  const T* x;
  ConvGeom g;
  int cols;  // kh*kw*C
  bool vec;  // C % CE == 0
  __device__ __forceinline__ uint4 load(int, int64_t k, int64_t K, int colchunk) const {
    constexpr int CE = 16 / sizeof(T);
    const int col = colchunk * CE;
    if (k >= K || col >= cols) return make_uint4(0, 0, 0, 0);
    const int wo = (int)(k % g.Wo);
    const int64_t t = k / g.Wo;
    const int ho = (int)(t % g.Ho);
    const int b = (int)(t / g.Ho);
    vec16<T> v;
    v.raw = make_uint4(0, 0, 0, 0);
    if (vec) {
      const int tap = col / g.C, c = col - tap * g.C;
      const int ky = tap / g.kw, kx = tap - ky * g.kw;
      const int hi = conv_src_h(g, ho * g.stride + ky - g.pad);
      const int wi = conv_src_w(g, wo * g.stride + kx - g.pad);
      return *reinterpret_cast<const uint4*>(x + (((int64_t)b * g.H + hi) * g.W + wi) * g.C + c);
    }
#pragma unroll
    for (int j = 0; j < CE; ++j)
      if (col + j < cols) {
        const int cc = col + j;
        const int tap = cc / g.C, c = cc - tap * g.C;
        const int ky = tap / g.kw, kx = tap - ky * g.kw;
        const int hi = conv_src_h(g, ho * g.stride + ky - g.pad);
        const int wi = conv_src_w(g, wo * g.stride + kx - g.pad);
        v.e[j] = x[(((int64_t)b * g.H + hi) * g.W + wi) * g.C + c];
      }
    return v.raw;
  }
};

// ----------------------------------------------------------------------------------------------
// TN engine.  Tile = TO rows (m) x TJ columns (j); K-step KS = 32 (bf16) / 16 (fp32) rows of k.
// LDS images are [k][m] and [k][j] exactly as loaded (coalesced along m / j); bf16 fragments are
// read with ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a
// 4 x 16 block and receives column (lane&15), rows 0..3 -- two reads give the 8 k-values a lane of
// v_mfma_f32_16x16x32_bf16 needs.  out is fp32 [batch][M][ldo]; atomic accumulate when split-K.
// ----------------------------------------------------------------------------------------------
template <typename T> struct TnFrag;

template <> struct TnFrag<bf16_t> {
  static constexpr int KS = 32;
  // tile: [KS][ROWLEN] bf16; returns the operand chunk of fragment column block c0 (16 columns)
  template <int ROWLEN>
  __device__ static __forceinline__ uint4 read(const bf16_t* tile, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const bf16_t* a0 = tile + (8 * g + q) * ROWLEN + c0 + 4 * p;
    const bf16_t* a1 = a0 + 4 * ROWLEN;
    union { uint4 u; s16x4 h[2]; } r;
    r.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    r.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    return r.u;
  }
};

template <> struct TnFrag<float> {
  static constexpr int KS = 16;
  template <int ROWLEN>
  __device__ static __forceinline__ uint4 read(const float* tile, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    union { uint4 u; float f[4]; } r;
#pragma unroll
    for (int s = 0; s < 4; ++s) r.f[s] = tile[(4 * s + g) * ROWLEN + c0 + i];
    return r.u;
  }
};

template <typename T, int TO, int TJ, class ALoad, class BLoad>
__global__ __launch_bounds__(256) void gemm_tn_kernel(ALoad al, BLoad bl, float* __restrict__ out, int M, int J,
                                                      int64_t K, int64_t klen, int ksplit, int64_t out_batch_stride,
                                                      int ldo) {
  constexpr int KS = TnFrag<T>::KS;
  constexpr int CE = 16 / sizeof(T);
  constexpr int MF = TO / 16;
  constexpr int NF = TJ / 64;  // 16-column fragments per wave (4 waves split TJ)
  constexpr int ACH_ROW = TO / CE, BCH_ROW = TJ / CE;
  constexpr int ACH = (KS * ACH_ROW + 255) / 256, BCH = (KS * BCH_ROW + 255) / 256;
  __shared__ __attribute__((aligned(16))) T ldsA[2][KS * TO];
  __shared__ __attribute__((aligned(16))) T ldsB[2][KS * TJ];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int batch = blockIdx.z / ksplit;
  const int ks = blockIdx.z % ksplit;
  const int m0 = blockIdx.y * TO;
  const int j0 = blockIdx.x * TJ;
  const int64_t kbeg = ks * klen;
  const int64_t kend = (kbeg + klen < K) ? kbeg + klen : K;
  const int nk = (int)((kend - kbeg + KS - 1) / KS);

  uint4 ra[ACH], rb[BCH];
  auto gload = [&](int kt) {
    const int64_t kb = kbeg + (int64_t)kt * KS;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int id = tid + i * 256;
      if (id < KS * ACH_ROW) ra[i] = al.load(batch, kb + id / ACH_ROW, kend, (m0 / CE) + id % ACH_ROW);
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int id = tid + i * 256;
      if (id < KS * BCH_ROW) rb[i] = bl.load(batch, kb + id / BCH_ROW, kend, (j0 / CE) + id % BCH_ROW);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int id = tid + i * 256;
      if (id < KS * ACH_ROW) reinterpret_cast<uint4*>(ldsA[buf])[id] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int id = tid + i * 256;
      if (id < KS * BCH_ROW) reinterpret_cast<uint4*>(ldsB[buf])[id] = rb[i];
    }
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
    uint4 a[MF], b[NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) a[mf] = TnFrag<T>::template read<TO>(ldsA[cur], mf * 16, lane);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) b[nf] = TnFrag<T>::template read<TJ>(ldsB[cur], wave * (TJ / 4) + nf * 16, lane);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) Mfma16<T>::run(acc[mf][nf], a[mf], b[nf]);
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  const int lr = lane & 15, lc = lane >> 4;
  float* ob = out + batch * out_batch_stride;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int j = j0 + wave * (TJ / 4) + nf * 16 + lr;
      if (j >= J) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mf * 16 + lc * 4 + r;
        if (m < M) {
          float* p = ob + (int64_t)m * ldo + j;
          if (ksplit > 1) atomicAdd(p, acc[mf][nf][r]);
          else *p = acc[mf][nf][r];
        }
      }
    }
}
