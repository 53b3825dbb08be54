// ADA parameter sampling and operator construction in three launches (the Python/torch form of the same
// logic in gans/augment/adaptive_augment.py needs ~160 tiny kernels per call, five calls per iteration).
//
// reference: AdaptiveAugment.sample_affine / sample_color / forward geometry,
//            gans/augment/adaptive_augment.py:386-469, 488-535.
//
//   ada_sample : per sample, compose the axis-aligned affine (sx, tx, sy, ty) from the policy's random
//                flips / integer + fractional translations / vertical scale, and the 4x4 colour matrix
//                (brightness, contrast, luma flip, hue, saturation) collapsed to one channel (a, c).
//                Randomness comes in as uniforms u [B,16] and normals n [B,8] (drawn by the host
//                framework's generator, so hipGraph replay advances them correctly).
//   ada_build  : Ay [B,H,H] = D_y S_y(b) M1y  and the K-tap circular x-filter (kx, off, sgn) = one row of
//                D_x S_x(b) M1x, where M1 = (up-FIR o pad) and D = down-FIR are geometry constants and
//                S(b) is 1-D linear interpolation at the sample's affine positions (zero outside).
#include "common.h"

namespace {

constexpr int NU = 16, NN = 8, NTAPS = 12;

struct AdaPolicy {
  float lr_flip, ud_flip, int_trans, iso_scale, frac_trans, brightness, contrast, luma_flip, hue, saturation;
  float h_trans_factor;
};

struct Mat4 {
  float m[4][4];
};

__device__ __forceinline__ Mat4 mat4_eye() {
  Mat4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) r.m[i][j] = (i == j) ? 1.f : 0.f;
  return r;
}

// C <- (sel * Cc + (1 - sel) * I) @ C
__device__ __forceinline__ void mat4_apply(Mat4& C, const Mat4& Cc, float sel) {
  Mat4 S, R;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) S.m[i][j] = sel * Cc.m[i][j] + (1.f - sel) * ((i == j) ? 1.f : 0.f);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += S.m[i][k] * C.m[k][j];
      R.m[i][j] = s;
    }
  C = R;
}

__global__ void ada_sample_kernel(float* __restrict__ gaff, float* __restrict__ a_out, float* __restrict__ c_out,
                                  const float* __restrict__ u, const float* __restrict__ n,
                                  const float* __restrict__ p_ptr, AdaPolicy pol, int B, int H, int W) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float p = p_ptr[0];
  const float* ub = u + (int64_t)b * NU;
  const float* nb = n + (int64_t)b * NN;
  float sx = 1.f, tx = 0.f, sy = 1.f, ty = 0.f;
  auto compose = [&](float sel, float csx, float csy, float ctx, float cty) {
    csx = sel * csx + (1.f - sel);
    csy = sel * csy + (1.f - sel);
    ctx *= sel;
    cty *= sel;
    tx = csx * tx + ctx;
    sx = csx * sx;
    ty = csy * ty + cty;
    sy = csy * sy;
  };
  auto pick = [&](float uu, float mul) { return (uu < p * mul) ? 1.f : 0.f; };
  if (pol.lr_flip > 0.f) compose(pick(ub[1], pol.lr_flip), 1.f - 2.f * (ub[0] < 0.5f ? 0.f : 1.f), 1.f, 0.f, 0.f);
  if (pol.ud_flip > 0.f) compose(pick(ub[3], pol.ud_flip), 1.f, 1.f - 2.f * (ub[2] < 0.5f ? 0.f : 1.f), 0.f, 0.f);
  if (pol.int_trans > 0.f) {
    const float uh = ub[4] * 0.25f - 0.125f, uw = ub[5] * 0.25f - 0.125f;
    compose(pick(ub[6], pol.int_trans), 1.f, 1.f, rintf(uw * (float)W), rintf(uh * (float)H) * pol.h_trans_factor);
  }
  if (pol.iso_scale > 0.f) compose(pick(ub[7], pol.iso_scale), 1.f, expf(nb[0] * 0.2f * 0.6931471805599453f), 0.f, 0.f);
  if (pol.frac_trans > 0.f)
    compose(pick(ub[8], pol.frac_trans), 1.f, 1.f, nb[2] * 0.125f * (float)W,
            nb[1] * 0.125f * (float)H * pol.h_trans_factor);
  gaff[b * 4 + 0] = sx;
  gaff[b * 4 + 1] = tx;
  gaff[b * 4 + 2] = sy;
  gaff[b * 4 + 3] = ty;

  Mat4 C = mat4_eye();
  const float v = 0.5773502691896258f;  // 1/sqrt(3); luma axis (v, v, v, 0)
  if (pol.brightness > 0.f) {
    Mat4 Cc = mat4_eye();
    const float t = nb[3] * 0.2f;
    Cc.m[0][3] = Cc.m[1][3] = Cc.m[2][3] = t;
    mat4_apply(C, Cc, pick(ub[9], pol.brightness));
  }
  if (pol.contrast > 0.f) {
    Mat4 Cc = mat4_eye();
    const float s = expf(nb[4] * 0.5f * 0.6931471805599453f);
    Cc.m[0][0] = Cc.m[1][1] = Cc.m[2][2] = s;
    mat4_apply(C, Cc, pick(ub[10], pol.contrast));
  }
  if (pol.luma_flip > 0.f) {
    Mat4 Cc = mat4_eye();
    const float i = ub[11] < 0.5f ? 0.f : 1.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) Cc.m[r][q] -= 2.f * v * v * i;
    mat4_apply(C, Cc, pick(ub[12], pol.luma_flip));
  }
  if (pol.hue > 0.f) {
    Mat4 Cc = mat4_eye();
    const float th = (ub[13] * 2.f - 1.f) * 3.14159265358979f;
    const float ct = cosf(th), st = sinf(th);
    const float cross[3][3] = {{0.f, -v, v}, {v, 0.f, -v}, {-v, v, 0.f}};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) Cc.m[r][q] = ct * ((r == q) ? 1.f : 0.f) + st * cross[r][q] + (1.f - ct) * v * v;
    mat4_apply(C, Cc, pick(ub[14], pol.hue));
  }
  if (pol.saturation > 0.f) {
    Mat4 Cc;
    const float s = expf(nb[5] * 0.6931471805599453f);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float vv = (r < 3 && q < 3) ? v * v : 0.f;
        Cc.m[r][q] = vv + (((r == q) ? 1.f : 0.f) - vv) * s;
      }
    mat4_apply(C, Cc, pick(ub[15], pol.saturation));
  }
  // 1-channel collapse: mean of the first three rows, then sum of its first three entries / its 4th entry
  float a = 0.f, c = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    a += C.m[r][0] + C.m[r][1] + C.m[r][2];
    c += C.m[r][3];
  }
  a_out[b] = a / 3.f;
  c_out[b] = c / 3.f;
}

// source position in the padded, 2x upsampled signal of grid_sample output index q (one axis)
__device__ __forceinline__ float sample_pos(float s, float trans, int n_in, int n_out, int q) {
  const float a = 1.f / s, t = -trans / s;
  const float A = a * ((float)n_out / (float)n_in);
  const float Bc = (2.f / (float)n_in) * (0.5f * a + 2.f * t - 0.5f);
  const float xn = (2.f * (float)q + 1.f) / (float)n_out - 1.f;
  const float xs = A * xn + Bc;
  return ((xs + 1.f) * (float)n_in - 1.f) / 2.f;
}

// linear interpolation of column `col` of M1 [n_rows, ld] at fractional row `pos`, zero outside
__device__ __forceinline__ float interp_col(const float* __restrict__ M1, int n_rows, int ld, int col, float pos) {
  // branch-free: rows outside the table load a clamped row with weight 0, so the 24 loads of an output (12 taps) are all in
  // flight at once instead of waiting on one another behind exec-masked branches (the operator-building kernels were
  // latency bound: 20 + 10 us per call, three calls per iteration)
  const float fl = floorf(pos);
  const int m = (int)fl;
  const float f = pos - fl;
  const bool in0 = m >= 0 && m < n_rows, in1 = m + 1 >= 0 && m + 1 < n_rows;
  const int m0 = m < 0 ? 0 : (m >= n_rows ? n_rows - 1 : m);
  const int m1 = m + 1 < 0 ? 0 : (m + 1 >= n_rows ? n_rows - 1 : m + 1);
  const float a0 = M1[(int64_t)m0 * ld + col], a1 = M1[(int64_t)m1 * ld + col];
  float v = 0.f;
  v += (in0 ? 1.f - f : 0.f) * a0;
  v += (in1 ? f : 0.f) * a1;
  return v;
}

// Ay[b, i, h] = sum_t k[t] * interp(M1y[:, h], pos_y(b, 2i + 1 + t));  one block per (i, b), >= H threads
__device__ __forceinline__ void ada_build_ay_block(float* __restrict__ Ay, const float* __restrict__ gaff,
                                                   const float* __restrict__ M1y, const float* __restrict__ taps, int H, int i,
                                                   int b) {
  const float sy = gaff[b * 4 + 2], ty = gaff[b * 4 + 3];
  const int n_in = (H + 2 * (H - 1)) * 2, n_out = (H + 6) * 2;
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) acc += taps[t] * interp_col(M1y, n_in, H, h, sample_pos(sy, ty, n_in, n_out, 2 * i + 1 + t));
    Ay[((int64_t)b * H + i) * H + h] = acc;
  }
}

// one reference row j_ref of the circulant x operator, its peak, and the K taps around the peak; one block of 256 threads per b
__device__ __forceinline__ void ada_build_kx_block(float* __restrict__ kx, int* __restrict__ off, int* __restrict__ sgn,
                                                   const float* __restrict__ gaff, const float* __restrict__ M1x,
                                                   const float* __restrict__ taps, int W, int K, int b, float* row) {
  __shared__ float best_v[4];
  __shared__ int best_i[4];
  const float sx = gaff[b * 4 + 0], tx = gaff[b * 4 + 1];
  const int n_in = (W + 2 * (W - 1)) * 2, n_out = (W + 6) * 2;
  const int j_ref = W / 2;
  float pos[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) pos[t] = sample_pos(sx, tx, n_in, n_out, 2 * j_ref + 1 + t);
  float bv = -1.f;
  int bi = 0;
  for (int m = threadIdx.x; m < W; m += blockDim.x) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) acc += taps[t] * interp_col(M1x, n_in, W, m, pos[t]);
    row[m] = acc;
    if (fabsf(acc) > bv) { bv = fabsf(acc); bi = m; }
  }
  // block arg-max (first maximum wins on ties, like torch.argmax)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { best_v[threadIdx.x >> 6] = bv; best_i[threadIdx.x >> 6] = bi; }
  __syncthreads();
  bv = best_v[0]; bi = best_i[0];
  for (int w = 1; w < 4; ++w)
    if (best_v[w] > bv || (best_v[w] == bv && best_i[w] < bi)) { bv = best_v[w]; bi = best_i[w]; }
  const int sg = sx < 0.f ? -1 : 1;
  const int o0 = bi - sg * j_ref - K / 2;
  for (int t = threadIdx.x; t < K; t += blockDim.x) kx[(int64_t)b * K + t] = row[floormod(o0 + t + sg * j_ref, W)];
  if (threadIdx.x == 0) {
    off[b] = floormod(o0, W);
    sgn[b] = sg;
  }
}

// both operators of a sample batch from ONE launch: blocks (i < H, b) build row i of Ay[b], blocks (H, b) the x kernel of b
// (two launches of 10 + 20 us, three times per iteration, ran one after the other for no reason: neither reads the other)
__global__ __launch_bounds__(256) void ada_build_kernel(float* __restrict__ Ay, float* __restrict__ kx, int* __restrict__ off,
                                                        int* __restrict__ sgn, const float* __restrict__ gaff,
                                                        const float* __restrict__ M1y, const float* __restrict__ M1x,
                                                        const float* __restrict__ taps, int H, int W, int K) {
  extern __shared__ float row[];  // [W] (the x-kernel blocks)
  if ((int)blockIdx.x < H) ada_build_ay_block(Ay, gaff, M1y, taps, H, blockIdx.x, blockIdx.y);
  else ada_build_kx_block(kx, off, sgn, gaff, M1x, taps, W, K, blockIdx.y, row);
}

}  // namespace

// u fp32 [B,16] uniforms in [0,1), n fp32 [B,8] standard normals, p fp32 [1] (device), policy fp32 [11] HOST
// array (lr_flip, ud_flip, int_trans, iso_scale, frac_trans, brightness, contrast, luma_flip, hue, saturation,
// h_trans_factor).  Outputs gaff [B,4] = (sx, tx, sy, ty), a [B], c [B].
extern "C" int dgv2_ada_sample(float* gaff, float* a, float* c, const float* u, const float* n, const float* p,
                               const float* policy_host, int B, int H, int W, void* stream) {
  if (!gaff || !a || !c || !u || !n || !p || !policy_host || B <= 0 || H <= 0 || W <= 0) return DGV2_EINVAL;
  AdaPolicy pol;
  pol.lr_flip = policy_host[0]; pol.ud_flip = policy_host[1]; pol.int_trans = policy_host[2];
  pol.iso_scale = policy_host[3]; pol.frac_trans = policy_host[4]; pol.brightness = policy_host[5];
  pol.contrast = policy_host[6]; pol.luma_flip = policy_host[7]; pol.hue = policy_host[8];
  pol.saturation = policy_host[9]; pol.h_trans_factor = policy_host[10];
  ada_sample_kernel<<<(B + 63) / 64, 64, 0, (hipStream_t)stream>>>(gaff, a, c, u, n, p, pol, B, H, W);
  DGV2_RETURN_LAST();
}

// gaff [B,4]; M1y fp32 [2(3H-2), H], M1x fp32 [2(3W-2), W] (up-FIR o pad chain constants), taps fp32 [12].
// Outputs Ay [B,H,H], kx [B,K], off [B], sgn [B] as consumed by dgv2_ada_apply.
extern "C" int dgv2_ada_build(float* Ay, float* kx, int* off, int* sgn, const float* gaff, const float* M1y,
                              const float* M1x, const float* taps, int B, int H, int W, int K, void* stream) {
  if (!Ay || !kx || !off || !sgn || !gaff || !M1y || !M1x || !taps || B <= 0 || H <= 0 || W <= 0 || K <= 0 || K > W)
    return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(H + 1, B);
  ada_build_kernel<<<grid, 256, sizeof(float) * W, st>>>(Ay, kx, off, sgn, gaff, M1y, M1x, taps, H, W, K);
  DGV2_RETURN_LAST();
}
