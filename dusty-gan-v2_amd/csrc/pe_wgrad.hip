// Weight gradient of the batch-shared positional-encoding columns of the generator's modulated 1x1 convs:
//     gw[b, o, k] = sum_p g[b, p, o] * pe[p, k]          g: [B, P, O] bf16 (gradient at the accumulator), pe: [P, Ks] bf16
// reference: the autograd of ModConv2d's grouped conv (gans/models/ops/style.py:105-118) on the PE channels that
// SynthesisBlock concatenates to every sample (gans/models/dusty_v2.py:153-162).
// Per sample this is a [O x P] . [P x Ks] product with O = 32 ... 128 rows: too few rows for the PE tile a block stages
// to pay for itself, and as a batched library GEMM the 33.5 MB encoding is re-read per sample (226 us at level 4).
// Here the M side of a block's tile is SEVERAL samples -- 128 / O of them, 128 rows -- that contract the SAME pixels, so
// one staged PE tile [32 p x 128 k] feeds four samples' accumulators; the pixel axis is split over blocks (partials +
// one summing launch: no atomics).  Both operands are pixel-major in memory (the contraction index is the slow one):
// fragments come from LDS through the transposing read ds_read_tr16_b64.
#include <type_traits>

#include "gemm_core.h"

namespace {

struct PWGeom {
  int B, P, O, Ks;
  int pchunk;               // pixels per split (multiple of 32)
  int64_t part_stride;      // B * O * Ks
  int64_t ldo;              // row pitch of the output (Ks for the partials of a split launch)
  int nz;                   // splits
};

constexpr int PW_ROW = 128 + 16;   // bf16 elements per LDS row: 288 bytes = 8 dwords mod 64 (see gemm_x3.hip)
constexpr int PW_PLANE = 32 * PW_ROW;

__device__ __forceinline__ uint4 pw_frag(const bf16_t* plane, int f, int lane) {
  return TnFrag<bf16_t>::template read<PW_ROW>(plane, f * 16, lane);
}

// grid (Ks / 128, B * O / 128, splits), 256 threads = 2 x 2 waves of 64 x 64
__global__ __launch_bounds__(256) void pe_wgrad_kernel(float* __restrict__ part, const bf16_t* __restrict__ g,
                                                       const bf16_t* __restrict__ pe, PWGeom q) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2][2 * PW_PLANE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wi = wave >> 1, wj = wave & 1;
  // XCD-aware block order (1-D launch; workgroup n runs on XCD n % 8): the Ks / 128 blocks that contract the SAME rows
  // of g over the same pixels get ids 8 apart, i.e. one XCD's L2 fetches that slice of g once for all of them
  const int nid = blockIdx.x, qid = nid >> 3;
  const int nk = q.Ks >> 7;
  const int rest = (qid / nk) * 8 + (nid & 7);       // (row tile, split), row tile fastest
  const int ny = q.B * q.O / 128;
  if (rest >= ny * q.nz) return;
  const int n0 = (qid % nk) * 128;
  const int row0 = (rest % ny) * 128;         // row = b * O + o
  const int bz = rest / ny;
  const int p_begin = bz * q.pchunk;
  const int p_end = min(p_begin + q.pchunk, q.P);
  const int steps = (p_end - p_begin) / 32;

  // slots: 512 16-byte pieces per operand tile [32 t][128], two per thread
  int64_t ga[2], gb[2];
  int ls[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int id = tid + s * 256;
    const int t = id >> 4, c = id & 15;
    const int r = row0 + 8 * c;               // 8 consecutive rows = 8 consecutive o of one sample (O % 8 == 0)
    const int b = r / q.O, o = r - b * q.O;
    ga[s] = ((int64_t)b * q.P + t) * q.O + o;
    gb[s] = (int64_t)t * q.Ks + n0 + 8 * c;
    ls[s] = t * PW_ROW + 8 * c;
  }
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;   // (HIP's uint4 struct arrays went through scratch memory)
  u32x4_ ra[2][2], rb[2][2];
  auto gload = [&](auto set, int p0) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ra[S][s] = *reinterpret_cast<const u32x4_*>(g + ga[s] + (int64_t)p0 * q.O);
      rb[S][s] = *reinterpret_cast<const u32x4_*>(pe + gb[s] + (int64_t)p0 * q.Ks);
    }
  };
  auto lstore = [&](auto set, int buf) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      *reinterpret_cast<u32x4_*>(&lds[buf][ls[s]]) = ra[S][s];
      *reinterpret_cast<u32x4_*>(&lds[buf][PW_PLANE + ls[s]]) = rb[S][s];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int cur) {
    union U { uint4 u; bf16x8 v; };
    U a[4], b[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      a[f].u = pw_frag(&lds[cur][0], wi * 4 + f, lane);
      b[f].u = pw_frag(&lds[cur][PW_PLANE], wj * 4 + f, lane);
    }
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf].v, b[nf].v, acc[mf][nf], 0, 0, 0);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // tile t: register set t & 1, LDS stage t & 1; the loads of tile st + 2 are issued at the top of step st
  auto step = [&](auto set_cur, int st) {
    constexpr int C = decltype(set_cur)::value;
    if (st + 2 < steps) gload(std::integral_constant<int, C>{}, p_begin + (st + 2) * 32);
    compute(C);
    if (st + 1 < steps) lstore(std::integral_constant<int, C ^ 1>{}, C ^ 1);
    __syncthreads();
  };
  if (steps > 0) {
    gload(S0{}, p_begin);
    if (steps > 1) gload(S1{}, p_begin + 32);
    lstore(S0{}, 0);
  }
  __syncthreads();
  int st = 0;
  for (; st + 1 < steps; st += 2) {
    step(S0{}, st);
    step(S1{}, st + 1);
  }
  if (st < steps) step(S0{}, st);

  // D layout: column (k) = lane & 15, rows = 4 (lane >> 4) + r
  float* ob = part + (int64_t)bz * q.part_stride;
  const int lr = lane & 15, lc = lane >> 4;
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + (wi * 4 + mf) * 16 + lc * 4 + r;
        const int k = n0 + (wj * 4 + nf) * 16 + lr;
        ob[(int64_t)row * q.ldo + k] = acc[mf][nf][r];
      }
}

// out[b, o, col0 + k] = sum_z part[z][b, o, k]  (ldo >= col0 + Ks: the caller's [B, O, Ka + Ks] gradient of the whole weight)
__global__ __launch_bounds__(256) void pe_wgrad_reduce_kernel(float* __restrict__ out, const float* __restrict__ part,
                                                              int64_t rows, int Ks, int64_t ldo, int nz) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= rows * Ks) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < nz; ++z) {
    const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)z * rows * Ks + i);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int64_t r = i / Ks;
  *reinterpret_cast<float4*>(out + r * ldo + (i - r * Ks)) = s;
}

int pw_splits(int B, int P, int O, int Ks) {
  const int tiles = (B * O / 128) * (Ks / 128);
  static const int target = getenv("DGV2_PW_BLOCKS") ? atoi(getenv("DGV2_PW_BLOCKS")) : 512;
  int s = (target + tiles - 1) / tiles;       // ~2 blocks per CU in flight; one split = no partials
  const int maxs = P / 256;                   // at least 8 K-steps per block
  s = s > maxs ? maxs : s;
  return s < 1 ? 1 : s;
}

}  // namespace

// fp32 scratch elements dgv2_pe_wgrad needs (0 and DGV2_ENOTSUP when the shape is not supported)
extern "C" int dgv2_pe_wgrad_scratch(int64_t* elems, int B, int P, int O, int Ks) {
  if (!elems) return DGV2_EINVAL;
  *elems = 0;
  if (B <= 0 || P <= 0 || O <= 0 || Ks <= 0) return DGV2_EINVAL;
  if ((O & 7) || 128 % O || (B * O) % 128 || (Ks & 127) || (P & 31)) return DGV2_ENOTSUP;
  *elems = (int64_t)pw_splits(B, P, O, Ks) * B * O * Ks;
  return 0;
}

// gw[b, o, col0 + k] = sum_p g[b, p, o] * pe[p, k], k < Ks  (gw fp32 [B, O, ldo]: the PE columns of the per-sample weight
// gradient written in place into the layer's [B, O, Ka + Ks] gradient).  g [B, P, O], pe [P, Ks] bf16.  O in {8 ... 128}
// dividing 128, B * O % 128 == 0, Ks % 128 == 0, P % 32 == 0; scratch >= dgv2_pe_wgrad_scratch elements.
// replaces: the PE columns of ModConv2d's weight gradient (autograd of gans/models/ops/style.py:105-118 on the input of
//   gans/models/dusty_v2.py:153-162).
extern "C" int dgv2_pe_wgrad(float* gw, float* scratch, int64_t scratch_elems, const void* g, const void* pe, int B, int P,
                             int O, int Ks, int64_t ldo, int col0, void* stream) {
  if (!gw || !scratch || !g || !pe || B <= 0 || P <= 0) return DGV2_EINVAL;
  if ((O & 7) || O <= 0 || 128 % O || (B * O) % 128 || Ks <= 0 || (Ks & 127) || (P & 31) || (ldo & 3) || (col0 & 3) ||
      ldo < col0 + Ks)
    return DGV2_ENOTSUP;
  if (!aligned16(gw) || !aligned16(scratch) || !aligned16(g) || !aligned16(pe)) return DGV2_EINVAL;
  int splits = pw_splits(B, P, O, Ks);
  int pchunk = ((P / 32 + splits - 1) / splits) * 32;
  splits = (P + pchunk - 1) / pchunk;
  const int64_t n = (int64_t)B * O * Ks;
  if (scratch_elems < (int64_t)splits * n) return DGV2_EINVAL;
  PWGeom q{B, P, O, Ks, pchunk, n, splits > 1 ? (int64_t)Ks : ldo, splits};
  hipStream_t st = (hipStream_t)stream;
  const int64_t rest = ((int64_t)(B * O / 128) * splits + 7) / 8 * 8;
  dim3 grid((unsigned)(rest * (Ks / 128)), 1, 1);
  if (splits == 1) {   // enough tiles to fill the chip: every block writes its finished tile
    pe_wgrad_kernel<<<grid, 256, 0, st>>>(gw + col0, (const bf16_t*)g, (const bf16_t*)pe, q);
    DGV2_RETURN_LAST();
  }
  pe_wgrad_kernel<<<grid, 256, 0, st>>>(scratch, (const bf16_t*)g, (const bf16_t*)pe, q);
  pe_wgrad_reduce_kernel<<<(int)((n / 4 + 255) / 256), 256, 0, st>>>(gw + col0, scratch, (int64_t)B * O, Ks, ldo, splits);
  DGV2_RETURN_LAST();
}
