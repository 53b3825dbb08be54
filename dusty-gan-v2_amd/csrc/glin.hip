// Grouped small Linear layers of the generator in fp32 on the matrix cores: the mapping network's two 512 x 512 layers
// and the 19 style affines of a pass are each ONE launch with the per-layer parameters' addresses in the kernel
// arguments (state-dict tensors stay separate; capturable: no device pointer tables).
//   reference: MappingNetwork / PixelNorm / EqualLR(nn.Linear) (gans/models/dusty_v2.py:13-29,
//   gans/models/ops/common.py:158-184,213-223) and ModConv2d.mod, the style affine of every modulated conv
//   (gans/models/ops/style.py:30,75).
// These are M = batch (32-64 rows) x N <= 1024 x K = 512 problems: 33-390 MFLOP per launch, i.e. microseconds of
// exact-fp32 MFMA time (v_mfma_f32_16x16x4_f32 = an fmaf chain: the results are the library GEMM's to rounding), but as
// library calls they were ~16 launches per generator forward (pixel norm as five element-wise ops, two addmm + two
// leaky ReLUs, pack + baddbmm + unpack for the affines) and twice that in backward.
// One kernel shape serves all three passes -- out[r, c] = f( alpha * sum_t A(r, t) B(c, t) ) on 16 x 64 tiles, a wave
// per 16 columns, both operands read as float4 runs of their contiguous axis:
//   forward   A = x [B, K] rows, B = W [N, K] rows (both contiguous in t);
//   d-input   A = g [B, N] rows (t = n), B = W^T: column c = k of W [N, K], contiguous in c -> per-t scalar loads;
//   d-weight  A = g^T: row r = n of g [B, N], contiguous in r; B = x^T likewise; t = batch.
#include "common.h"

namespace {

constexpr int GL_MAX = 24;

struct GLGroup {
  const float* a;     // forward: input x (row stride lda); d-input: g_l [B, N]; d-weight: g_l [B, N]
  const float* b;     // forward / d-input: W_l [N, K]; d-weight: x (row stride ldb)
  const float* bias;  // forward: bias [N] or nullptr
  const float* mask;  // d-input / d-weight: the layer's forward OUTPUT (leaky-ReLU backward: g * (y > 0 ? 1 : slope)) or nullptr
  float* out;         // forward: y_l [B, N]; d-input: dx (row stride ldo, ACCUMULATED over the groups that share it); d-weight: dW_l [N, K]
  float* out2;        // d-weight: dbias_l [N] or nullptr
  int N;              // the layer's output features
  int lda, ldb, ldo;  // row strides (elements)
  int tile0;          // first column tile of this group in blockIdx.x
};

struct GLArgs {
  GLGroup g[GL_MAX];
  int L, B, K;
  float alpha, beta;   // y = act(alpha * x W^T + beta * bias)
  float slope;         // leaky ReLU slope (act) / its backward (mask)
  int act;             // forward: 0 none, 1 leaky ReLU
  int prenorm;         // forward: x <- x * rsqrt(mean_k x^2 + 1e-8) per row first (PixelNorm)
  float* rnorm;        // prenorm: the per-row factors [B] are also stored here (for the backward), or nullptr
};

__device__ __forceinline__ f32x4 mfma4(const float4& a, const float4& b, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ int find_group(const GLArgs& p, int tile) {
  int l = 0;
#pragma unroll 1
  for (int i = 1; i < p.L; ++i)
    if (tile >= p.g[i].tile0) l = i;
  return l;
}

// forward: y_l[b, n] = act( alpha * rn[b] * sum_k x[b, k] W_l[n, k] + beta * bias_l[n] )
// A block is a 16 x 64 tile; its four waves SPLIT K (wave w contracts k in [w K/4, (w+1) K/4) for all four 16-column
// groups) and fold through LDS.  The first form gave every wave its own 16 columns and the whole of K: a chain of K / 64
// round trips to L2 with eight loads each and 128 dependent MFMAs -- 16 us for microseconds of work (three such launches
// were 8 % of the generator-only forward).  Here a wave's loads of 128 k are all in flight at once and its four
// accumulators are independent chains.
__global__ __launch_bounds__(256) void glin_fwd_kernel(GLArgs p) {
  __shared__ f32x4 red[3][4][64];
  __shared__ float red_ss[4][16];
  const int l = find_group(p, blockIdx.x);
  const GLGroup& g = p.g[l];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int b0 = blockIdx.y * 16;
  const int n0 = (blockIdx.x - g.tile0) * 64;
  const int kw = p.K >> 2;                        // this wave's share of K (K % 64 == 0: a multiple of 16)
  const int brow = min(b0 + r, p.B - 1);
  const float4* xa = reinterpret_cast<const float4*>(g.a + (int64_t)brow * g.lda + wave * kw) + kq;
  const float4* wb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    wb[c] = reinterpret_cast<const float4*>(g.b + (int64_t)min(n0 + 16 * c + r, g.N - 1) * p.K + wave * kw) + kq;
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float ss = 0.f;
  const int steps = kw >> 4;                      // 16 k per step: element j of lane group kq is k = 16 s + 4 kq + j
  for (int s0 = 0; s0 < steps; s0 += 8) {         // up to 128 k per trip: 40 loads in flight
    float4 a[8], b[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = min(s0 + u, steps - 1);       // (a short last trip re-loads its last step; not accumulated)
      a[u] = xa[4 * s];
#pragma unroll
      for (int c = 0; c < 4; ++c) b[u][c] = wb[c][4 * s];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u < steps) {
        if (p.prenorm) ss = fmaf(a[u].x, a[u].x, fmaf(a[u].y, a[u].y, fmaf(a[u].z, a[u].z, fmaf(a[u].w, a[u].w, ss))));
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = mfma4(a[u], b[u][c], acc[c]);
      }
    }
  }
  if (p.prenorm) {   // row r's partial sum of squares lives in the four lanes (r, kq = 0..3) of every wave
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (kq == 0) red_ss[wave][r] = ss;
  }
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave - 1][c][lane] = acc[c];
  }
  __syncthreads();
  if (wave > 0) return;
  float rn = 1.f;
  if (p.prenorm) {
    rn = rsqrtf((red_ss[0][r] + red_ss[1][r] + red_ss[2][r] + red_ss[3][r]) / (float)p.K + 1e-8f);
    if (p.rnorm && blockIdx.x == 0 && kq == 0 && b0 + r < p.B) p.rnorm[b0 + r] = rn;
  }
  // C layout: column (n) = lane & 15, rows (batch) 4 kq + j
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f32x4 t = acc[c];
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const f32x4 o = red[w][c][lane];
      t[0] += o[0]; t[1] += o[1]; t[2] += o[2]; t[3] += o[3];
    }
    const int n = n0 + 16 * c + r;
    const float bv = (g.bias && n < g.N) ? g.bias[n] * p.beta : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bb = b0 + 4 * kq + j;
      const float rj = p.prenorm ? __shfl(rn, 4 * kq + j, 64) : 1.f;   // the factor of row 4 kq + j sits in lane r = 4 kq + j
      float v = fmaf(t[j] * rj, p.alpha, bv);
      if (p.act == 1) v = v > 0.f ? v : v * p.slope;
      if (bb < p.B && n < g.N) g.out[(int64_t)bb * g.ldo + n] = v;
    }
  }
}

// d-input: dx[b, k] (+)= alpha * sum_l sum_n gm_l[b, n] W_l[n, k],  gm = g * (mask > 0 ? 1 : slope).  The groups of one
// launch all ADD into ONE output (the layers that read one input): the contraction runs over every output feature of every
// layer (6 k for the 19 style affines), so it is cut into CHUNKS of up to 256 features (blockIdx.z; GLGroup::tile0 = the
// group's first chunk): a block contracts one chunk for a 16 x 64 tile of dx with all its loads in flight and leaves a
// partial, glin_din_reduce_kernel folds the chunks (fixed order: reproducible).  The first form walked all layers in one
// wave: a serial chain of ~190 round trips to L2.
constexpr int GL_CHUNK = 256;

__global__ __launch_bounds__(256) void glin_din_kernel(GLArgs p, float* __restrict__ part) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int b0 = blockIdx.y * 16;
  const int k0 = blockIdx.x * 64 + wave * 16;
  const int brow = min(b0 + r, p.B - 1);
  const int l = find_group(p, blockIdx.z);
  const GLGroup& g = p.g[l];
  const int nbeg = (blockIdx.z - g.tile0) * GL_CHUNK, nend = min(nbeg + GL_CHUNK, g.N);
  const float* ga = g.a + (int64_t)brow * g.lda;
  const float* ma = g.mask ? g.mask + (int64_t)brow * g.lda : nullptr;
  const float* wcol = g.b + min(k0 + r, p.K - 1);             // W[n][k0 + r]: contiguous over the 16 lanes of a group
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int steps = (nend - nbeg) >> 4;                         // N % 32 == 0 (host-checked): whole steps of 16
  for (int s0 = 0; s0 < steps; s0 += 8) {
    float4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int nb = nbeg + 16 * min(s0 + u, steps - 1) + 4 * kq;
      a[u] = *reinterpret_cast<const float4*>(ga + nb);
      if (ma) {
        const float4 m = *reinterpret_cast<const float4*>(ma + nb);
        a[u].x *= m.x > 0.f ? 1.f : p.slope; a[u].y *= m.y > 0.f ? 1.f : p.slope;
        a[u].z *= m.z > 0.f ? 1.f : p.slope; a[u].w *= m.w > 0.f ? 1.f : p.slope;
      }
      b[u].x = wcol[(int64_t)(nb + 0) * p.K]; b[u].y = wcol[(int64_t)(nb + 1) * p.K];
      b[u].z = wcol[(int64_t)(nb + 2) * p.K]; b[u].w = wcol[(int64_t)(nb + 3) * p.K];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < steps) acc = mfma4(a[u], b[u], acc);
  }
  const int k = k0 + r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int bb = b0 + 4 * kq + j;
    if (bb < p.B && k < p.K) part[((int64_t)blockIdx.z * p.B + bb) * p.K + k] = acc[j];
  }
}

__global__ __launch_bounds__(256) void glin_din_reduce_kernel(float* __restrict__ dx, int ldx, const float* __restrict__ part,
                                                              int nchunks, int B, int K, float alpha, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * K) return;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += part[(int64_t)c * B * K + i];
  float* q = dx + (int64_t)(i / K) * ldx + i % K;
  *q = accumulate ? *q + s * alpha : s * alpha;
}

// d-weight: dW_l[n, k] = alpha * sum_b gm_l[b, n] x[b, k] (* rn[b] when the forward normalised x),  dbias_l[n] = beta * sum_b gm_l[b, n]
__global__ __launch_bounds__(256) void glin_dw_kernel(GLArgs p, int ktiles) {
  const int tile = blockIdx.x / ktiles, kt = blockIdx.x - tile * ktiles;
  const int l = find_group(p, tile);
  const GLGroup& g = p.g[l];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int n0 = (tile - g.tile0) * 16;                      // 16 output rows (n) per tile, 64 columns (k) per block
  const int k0 = kt * 64 + wave * 16;
  const int n = min(n0 + r, g.N - 1), k = min(k0 + r, p.K - 1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int steps = (p.B + 15) >> 4;
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    float av[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bb = 16 * s + 4 * kq + j;
      const bool ok = bb < p.B;
      const int bc = ok ? bb : 0;
      float a = ok ? g.a[(int64_t)bc * g.lda + n] : 0.f;
      if (g.mask && ok) a *= g.mask[(int64_t)bc * g.lda + n] > 0.f ? 1.f : p.slope;
      float x = ok ? g.b[(int64_t)bc * g.ldb + k] : 0.f;
      if (p.rnorm && ok) x *= p.rnorm[bc];
      av[j] = a;
      bv[j] = x;
      bsum += a;
    }
    acc = mfma4(make_float4(av[0], av[1], av[2], av[3]), make_float4(bv[0], bv[1], bv[2], bv[3]), acc);
  }
  // C: column (k) = lane & 15, rows (n) 4 kq + j
  const int kk = k0 + r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nn = n0 + 4 * kq + j;
    if (nn < g.N && kk < p.K) g.out[(int64_t)nn * p.K + kk] = acc[j] * p.alpha;
  }
  if (g.out2 && kt == 0 && wave == 0) {                      // bias gradient: row n = n0 + r summed over the batch
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (kq == 0 && n0 + r < g.N) g.out2[n0 + r] = bsum * p.beta;
  }
}

int fill(GLArgs& p, int L, const float* const* a, const float* const* b, const float* const* bias, const float* const* mask,
         float* const* out, float* const* out2, const int* N, const int* lda, const int* ldb, const int* ldo, int tile_w) {
  int t = 0;
  for (int l = 0; l < L; ++l) {
    if (!a[l] || !b[l] || !out[l] || N[l] < 1) return -1;
    if (!aligned16(a[l]) || !aligned16(b[l]) || (mask && mask[l] && !aligned16(mask[l]))) return -1;
    p.g[l] = GLGroup{a[l], b[l], bias ? bias[l] : nullptr, mask ? mask[l] : nullptr, out[l], out2 ? out2[l] : nullptr, N[l],
                     lda[l], ldb ? ldb[l] : 0, ldo ? ldo[l] : N[l], t};
    t += (N[l] + tile_w - 1) / tile_w;
  }
  return t;
}

}  // namespace

// y_l [B, N_l] = act( alpha * PN(x_l) W_l^T + beta * bias_l ) for l < L <= 24 in ONE launch (fp32, exact fmaf chains on
// v_mfma_f32_16x16x4_f32).  HOST arrays of device pointers / ints: x_l rows of K floats at stride lda[l] (the style
// vector of layer l inside ws [B, S, K]: lda = S * K), W_l [N_l, K], bias_l [N_l] or NULL, y_l contiguous.
// act: 0 | 1 = leaky ReLU(slope); prenorm != 0: x rows are normalised by rsqrt(mean_k x^2 + 1e-8) first (PixelNorm), the
// factors also stored in rnorm [B] when given.  K % 64 == 0, 16-byte aligned rows.
extern "C" int dgv2_glin_fwd(float* const* y, const float* const* x, const float* const* w, const float* const* bias,
                             const int* N, const int* lda, int L, int B, int K, float alpha, float beta, int act, float slope,
                             int prenorm, float* rnorm, void* stream) {
  if (!y || !x || !w || !N || !lda || L < 1 || L > GL_MAX || B < 1 || K < 64) return DGV2_EINVAL;
  if (K & 63) return DGV2_ENOTSUP;
  GLArgs p;
  for (int l = 0; l < L; ++l)
    if (lda[l] & 3) return DGV2_EINVAL;
  const int tiles = fill(p, L, x, w, bias, nullptr, y, nullptr, N, lda, nullptr, nullptr, 64);
  if (tiles < 0) return DGV2_EINVAL;
  p.L = L; p.B = B; p.K = K; p.alpha = alpha; p.beta = beta; p.slope = slope; p.act = act; p.prenorm = prenorm; p.rnorm = rnorm;
  glin_fwd_kernel<<<dim3(tiles, (B + 15) / 16), 256, 0, (hipStream_t)stream>>>(p);
  DGV2_RETURN_LAST();
}

// dx [B, K] at row stride ldx (=|+=) alpha * sum_l (g_l . act'(y_l)) W_l over the L layers that read this input
// (accumulate != 0: added to what dx holds).  g_l, y_l [B, N_l] contiguous (y_l NULL: no activation), N_l % 32 == 0.
// scratch: fp32 [>= B * K * sum_l ceil(N_l / 256)] (the chunk partials of the contraction over the layers' features).
extern "C" int dgv2_glin_dinput(float* dx, int ldx, float* scratch, int64_t scratch_elems, const float* const* g,
                                const float* const* yact, const float* const* w, const int* N, int L, int B, int K, float alpha,
                                float slope, int accumulate, void* stream) {
  if (!dx || !scratch || !g || !w || !N || L < 1 || L > GL_MAX || B < 1 || K < 16 || (K & 15)) return DGV2_EINVAL;
  GLArgs p;
  float* outs[GL_MAX];
  int lda[GL_MAX], ldo[GL_MAX];
  for (int l = 0; l < L; ++l) {
    if (N[l] & 31) return DGV2_ENOTSUP;
    outs[l] = dx; lda[l] = N[l]; ldo[l] = ldx;
  }
  const int nchunks = fill(p, L, g, w, nullptr, yact, outs, nullptr, N, lda, nullptr, ldo, GL_CHUNK);
  if (nchunks < 0 || scratch_elems < (int64_t)nchunks * B * K || !aligned16(scratch)) return DGV2_EINVAL;
  p.L = L; p.B = B; p.K = K; p.alpha = alpha; p.beta = 0.f; p.slope = slope; p.act = accumulate; p.prenorm = 0; p.rnorm = nullptr;
  hipStream_t st = (hipStream_t)stream;
  glin_din_kernel<<<dim3((K + 63) / 64, (B + 15) / 16, nchunks), 256, 0, st>>>(p, scratch);
  glin_din_reduce_kernel<<<(B * K + 255) / 256, 256, 0, st>>>(dx, ldx, scratch, nchunks, B, K, alpha, accumulate);
  DGV2_RETURN_LAST();
}

// dW_l [N_l, K] = alpha * (g_l . act'(y_l))^T (x_l * rnorm),  dbias_l [N_l] = beta * column sums (dbias / entries NULL: skipped)
// for l < L in ONE launch; x_l rows at stride ldx[l].
extern "C" int dgv2_glin_dweight(float* const* dw, float* const* dbias, const float* const* g, const float* const* yact,
                                 const float* const* x, const int* N, const int* ldx, int L, int B, int K, float alpha,
                                 float beta, float slope, const float* rnorm, void* stream) {
  if (!dw || !g || !x || !N || !ldx || L < 1 || L > GL_MAX || B < 1 || K < 16) return DGV2_EINVAL;
  GLArgs p;
  int lda[GL_MAX];
  for (int l = 0; l < L; ++l) lda[l] = N[l];
  const int tiles = fill(p, L, g, x, nullptr, yact, dw, dbias, N, lda, ldx, nullptr, 16);
  if (tiles < 0) return DGV2_EINVAL;
  p.L = L; p.B = B; p.K = K; p.alpha = alpha; p.beta = beta; p.slope = slope; p.act = 0; p.prenorm = 0;
  p.rnorm = const_cast<float*>(rnorm);
  const int ktiles = (K + 63) / 64;
  glin_dw_kernel<<<dim3(tiles * ktiles), 256, 0, (hipStream_t)stream>>>(p, ktiles);
  DGV2_RETURN_LAST();
}
