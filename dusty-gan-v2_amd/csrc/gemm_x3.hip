// fp32 GEMMs on the bf16 matrix cores: every fp32 operand value is split into three bf16 planes
//     x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)      (|x - h - m - l| <= 2^-26 |x|)
// while its tile is staged into LDS, and a product a * b is taken as the six bf16 products
//     a_l b_h + a_h b_l + a_m b_m + a_m b_h + a_h b_m + a_h b_h
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the dropped terms a_m b_l, a_l b_m, a_l b_l are below 2^-25 of
// the product: less than the rounding of an fp32 multiply-add chain itself).  gfx950 runs bf16 MFMA at 16x its fp32
// MFMA rate (2.5 PFLOP/s vs 157 TFLOP/s dense), so six bf16 products cost 3/8 of one fp32 product.
//
// Used for the three GEMMs of the discriminator's 65536 -> 512 Linear (reference: gans/models/dusty_v2.py:381-383 under
// the fp32 island of :394-395), which are weight-bandwidth-bound shapes (134 MB of fp32 weights against 8.6 GFLOP):
//     forward   y[m, n]  = sum_k x[m, k] W[n, k]      both operands K-contiguous, split-K over the chip
//     dgrad     gx[m, k] = sum_n g[m, n] W[n, k]      W as the transposed operand
//     wgrad     gw[n, k] = sum_m g[m, n] x[m, k]      both operands transposed, 134 MB of output
// One kernel, C[i, j] = scale * sum_t A(i, t) B(j, t), with each operand either "direct" (memory [i][t], t contiguous)
// or "transposed" (memory [t][i], i contiguous; fragments through ds_read_tr16_b64).
#include <type_traits>

#include "gemm_core.h"

namespace {

struct X3Geom {
  int I, J;
  int64_t T, lda, ldb, ldo;
  int tchunk;            // contraction range of one split (multiple of 32)
  float scale;
  int64_t part_stride;   // elements between the outputs of two splits (0: one split writes C itself)
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;

// 4 floats -> the three planes' 4 bf16 each
__device__ __forceinline__ void split3(const uint4& v, uint2& h, uint2& m, uint2& l) {
  const float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  union { uint2 u; bf16_t e[4]; } ph, pm, pl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16_t hh = (bf16_t)f[i];
    const float r1 = f[i] - (float)hh;
    const bf16_t mm = (bf16_t)r1;
    const float r2 = r1 - (float)mm;
    ph.e[i] = hh;
    pm.e[i] = mm;
    pl.e[i] = (bf16_t)r2;
  }
  h = ph.u;
  m = pm.u;
  l = pl.u;
}

template <bool TR, int BI> struct X3Op {
  // LDS image of one plane of a [BI x 32] operand tile, in bf16 elements.
  //   direct:     [4 kq][BI rows][8 k] -- a fragment read (lane = row r, k chunk kq) is four contiguous 256-byte runs, any
  //               8 consecutive lanes cover 128 contiguous bytes (row-major [BI][32] put lanes r and r + 4 on the same
  //               banks: 2-way conflicts on every read, 2 us per K-step instead of 0.8);
  //   transposed: [32 t][BI + 16] -- a transposing read touches 4 t-rows x 32 bytes per 16 lanes; a row pitch of 8
  //               dwords mod 64 puts those 16 pieces on 16 different bank pairs.
  static constexpr int ROW = TR ? BI + 16 : 32;
  static constexpr int PLANE = TR ? 32 * ROW : BI * 32;
  static constexpr int SLOTS = BI * 8 / 256;               // 16-byte global loads per thread and tile
  // global element offset of slot `id` (tile origin excluded) and its plane offset
  __device__ static __forceinline__ int64_t goff(int id, int64_t ld) {
    if constexpr (TR) return (int64_t)(id / (BI / 4)) * ld + (id % (BI / 4)) * 4;
    else return (int64_t)(id >> 3) * ld + (id & 7) * 4;
  }
  __device__ static __forceinline__ int loff(int id) {
    if constexpr (TR) return (id / (BI / 4)) * ROW + (id % (BI / 4)) * 4;
    else return ((((id & 7) >> 1) * BI + (id >> 3)) * 8) + (id & 1) * 4;
  }
  // the 8 contraction values of row f * 16 + (lane & 15) a lane of v_mfma_f32_16x16x32_bf16 holds
  __device__ static __forceinline__ uint4 frag(const bf16_t* plane, int f, int lane) {
    if constexpr (TR) return TnFrag<bf16_t>::template read<ROW>(plane, f * 16, lane);
    else return *reinterpret_cast<const uint4*>(plane + ((lane >> 4) * BI + f * 16 + (lane & 15)) * 8);
  }
};

// IL: the split of tile st + 1 is issued inside the MFMA loop of tile st (see `convert` below) instead of as a phase of its
// own between two barriers.  Measured per shape (profiles/round6_mb_linear_x3.txt): the skinny forward at M = 64 gains
// (61 -> 44 us with 128 splits: level with the library's fp32 split-K + sum, and one launch less), the M = 128 forms
// need the registers of two blocks per CU more than the overlap (dgrad 60 -> 82 us at one block per CU) and the
// doubly-transposed weight gradient loses as well (37 -> 52 us): those keep the phase form.
template <bool AT, bool BT, int MF, int NF, bool IL>   // a wave owns (16 MF) x (16 NF) of C; 2 x 2 waves per block
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(float* __restrict__ out, const float* __restrict__ A,
                                                      const float* __restrict__ B, X3Geom g) {
  constexpr int BI = 32 * MF, BJ = 32 * NF;
  using OA = X3Op<AT, BI>;
  using OB = X3Op<BT, BJ>;
  // ONE LDS stage of three planes per operand (48-52 KB): two or three blocks per CU hide each other's phases.  (With
  // two stages and one block per CU -- the split riding inside the MFMA loop -- the data gradient took 78 us instead of
  // 61, the weight gradient 97 instead of 68.)
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wi = wave >> 1, wj = wave & 1;
  const int i0 = blockIdx.y * BI, j0 = blockIdx.x * BJ;
  const int64_t t_begin = (int64_t)blockIdx.z * g.tchunk;
  const int64_t t_end = t_begin + g.tchunk < g.T ? t_begin + g.tchunk : g.T;
  const int steps = (int)((t_end - t_begin) / 32);

  // tile origins: direct operand rows are C's rows / columns, transposed operand rows are the contraction index
  const float* Ab = AT ? A + i0 : A + (int64_t)i0 * g.lda;
  const float* Bb = BT ? B + j0 : B + (int64_t)j0 * g.ldb;
  int64_t ga[OA::SLOTS], gb[OB::SLOTS];
  int la[OA::SLOTS], lb[OB::SLOTS];
#pragma unroll
  for (int s = 0; s < OA::SLOTS; ++s) {
    ga[s] = OA::goff(tid + s * 256, g.lda);
    la[s] = OA::loff(tid + s * 256);
  }
#pragma unroll
  for (int s = 0; s < OB::SLOTS; ++s) {
    gb[s] = OB::goff(tid + s * 256, g.ldb);
    lb[s] = OB::loff(tid + s * 256);
  }
  // Two register sets: the tile of step st + 2 is requested at the top of step st and split into LDS at the end of
  // step st + 1.
  uint4 ra[2][OA::SLOTS], rb[2][OB::SLOTS];
  auto gload = [&](auto set, int64_t t0) {
    constexpr int S = decltype(set)::value;
    const float* ap = AT ? Ab + t0 * g.lda : Ab + t0;
    const float* bp = BT ? Bb + t0 * g.ldb : Bb + t0;
#pragma unroll
    for (int s = 0; s < OA::SLOTS; ++s) ra[S][s] = *reinterpret_cast<const uint4*>(ap + ga[s]);
#pragma unroll
    for (int s = 0; s < OB::SLOTS; ++s) rb[S][s] = *reinterpret_cast<const uint4*>(bp + gb[s]);
  };
  // The split of a tile runs in two halves: `convert` turns the fp32 registers of one operand slot into the three planes'
  // packed words (VALU only: 14 conversions + 8 subtractions per 4 values), `lwrite` stores them.  Round 6: the
  // conversions of tile st + 1 are issued INSIDE the MFMA loop of tile st, a slot per fragment group (step()), so that the
  // vector ALU works while the matrix pipe does -- as one phase between two barriers they cost as much as the MFMAs
  // themselves (~250 VALU instructions against 48 MFMAs per wave and step at M = 64) and nothing overlapped them within
  // a block.
  uint2 ca[OA::SLOTS][3], cb[OB::SLOTS][3];
  auto convert_a = [&](auto set, int s) {
    constexpr int S = decltype(set)::value;
    split3(ra[S][s], ca[s][0], ca[s][1], ca[s][2]);
  };
  auto convert_b = [&](auto set, int s) {
    constexpr int S = decltype(set)::value;
    split3(rb[S][s], cb[s][0], cb[s][1], cb[s][2]);
  };
  auto lwrite = [&]() {
    bf16_t* pa = lds;
    bf16_t* pb = pa + 3 * OA::PLANE;
#pragma unroll
    for (int s = 0; s < OA::SLOTS; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(pa + p * OA::PLANE + la[s]) = ca[s][p];
#pragma unroll
    for (int s = 0; s < OB::SLOTS; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(pb + p * OB::PLANE + lb[s]) = cb[s][p];
  };
  auto lstore = [&](auto set) {
#pragma unroll
    for (int s = 0; s < OA::SLOTS; ++s) convert_a(set, s);
#pragma unroll
    for (int s = 0; s < OB::SLOTS; ++s) convert_b(set, s);
    lwrite();
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the six products of an (mf, nf) fragment are issued product-major over the MF independent accumulators; behind the
  // MFMAs of fragment group nf the conversions of share nf of the NEXT tile's slots (cv = true)
  constexpr int TOT = OA::SLOTS + OB::SLOTS;
  auto compute = [&](auto set_next, auto cv_) {
    constexpr bool cv = decltype(cv_)::value;
    const bf16_t* pa = lds;
    const bf16_t* pb = pa + 3 * OA::PLANE;
    union U { uint4 u; bf16x8 v; };
    U a[3][MF];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) a[p][mf].u = OA::frag(pa + p * OA::PLANE, wi * MF + mf, lane);
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      U b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p].u = OB::frag(pb + p * OB::PLANE, wj * NF + nf, lane);
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[k]][mf].v, b[PB[k]].v, acc[mf][nf], 0, 0, 0);
      if constexpr (cv) {
#pragma unroll
        for (int q = nf * TOT / NF; q < (nf + 1) * TOT / NF; ++q) {
          if (q < OA::SLOTS) convert_a(set_next, q);
          else convert_b(set_next, q - OA::SLOTS);
        }
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using Yes = std::true_type;
  using No = std::false_type;
  // tile t lives in register set t & 1; its split is written to LDS between two barriers
  auto step = [&](auto set_cur, auto more_, int st) {
    constexpr int C = decltype(set_cur)::value;
    constexpr bool more = decltype(more_)::value;     // a tile st + 1 exists (compile time: no branch inside the MFMA loop)
    using SC = std::integral_constant<int, C>;
    using SN = std::integral_constant<int, C ^ 1>;
    // set C was converted during step st - 1 and written to LDS at its end: its registers are free for tile st + 2 once
    // the conversions of set N below no longer share an instruction window with them -- the loads are requested AFTER
    // the MFMA / conversion loop has been issued (they still have a whole step to land)
    if constexpr (IL) {
      compute(SN{}, more_);
      if (st + 2 < steps) gload(SC{}, t_begin + (int64_t)(st + 2) * 32);
      __syncthreads();
      if constexpr (more) lwrite();
    } else {
      if (st + 2 < steps) gload(SC{}, t_begin + (int64_t)(st + 2) * 32);   // set C is free: tile st went to LDS a step ago
      compute(SN{}, No{});
      __syncthreads();
      if constexpr (more) lstore(SN{});
    }
    __syncthreads();
  };
  if (steps > 0) {
    gload(S0{}, t_begin);
    if (steps > 1) gload(S1{}, t_begin + 32);
    lstore(S0{});
  }
  __syncthreads();
  int st = 0;
  for (; st + 2 < steps; st += 2) {
    step(S0{}, Yes{}, st);
    step(S1{}, Yes{}, st + 1);
  }
  if (st + 1 < steps) {
    step(S0{}, Yes{}, st);
    step(S1{}, No{}, st + 1);
  } else if (st < steps) {
    step(S0{}, No{}, st);
  }

  // D layout: column (j) = lane & 15, rows (i) = 4 (lane >> 4) + r
  float* ob = out + (int64_t)blockIdx.z * g.part_stride;
  const int lr = lane & 15, lc = lane >> 4;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + (wi * MF + mf) * 16 + lc * 4 + r;
        const int j = j0 + (wj * NF + nf) * 16 + lr;
        ob[(int64_t)i * g.ldo + j] = acc[mf][nf][r] * g.scale;
      }
}

// out[i] = sum_z part[z][i], n a multiple of 4.  256 threads = 16 float4 columns x 16 split lanes (every lane sums its share
// of the splits with independent loads, LDS folds the 16 lanes in lane order): one thread walking 64 splits was 64
// dependent round trips, 17 us for 17 MB.
__global__ __launch_bounds__(256) void x3_reduce_kernel(float* __restrict__ out, const float* __restrict__ part, int64_t n,
                                                        int nz) {
  __shared__ float4 red[16][16];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = ((int64_t)blockIdx.x * 16 + col) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
#pragma unroll 4
    for (int z = sl; z < nz; z += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)z * n + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[sl][col] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 v = red[k][col];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = s;
  }
}

template <bool AT, bool BT, int MF, bool IL = false>
int x3_launch(float* out, const float* A, const float* B, const X3Geom& g, int splits, hipStream_t st) {
  constexpr int NF = 4;
  using OA = X3Op<AT, 32 * MF>;
  using OB = X3Op<BT, 32 * NF>;
  const size_t lds = 3 * (size_t)(OA::PLANE + OB::PLANE) * sizeof(bf16_t);
  auto kern = gemm_x3_kernel<AT, BT, MF, NF, IL>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid(g.J / (32 * NF), g.I / (32 * MF), splits);
  kern<<<grid, 256, lds, st>>>(out, A, B, g);
  return 0;
}

}  // namespace

// C[i, j] = scale * sum_{t < T} A(i, t) * B(j, t)  in fp32-equivalent arithmetic on the bf16 matrix cores (three-plane
// split, six products; see the top of this file).  a_trans == 0: A is [I, lda] with t contiguous; != 0: A is [T, lda]
// with i contiguous (same for B / ldb / j).  C [I, ldo] fp32.  I a multiple of 64, J of 128, T of 32; lda, ldb
// multiples of 4, all pointers 16-byte aligned.  splits > 1 (split-K): scratch >= splits * I * ldo floats takes the
// partial results and a second launch sums them (ldo == J required); splits <= 1: scratch may be NULL.
// Returns DGV2_ENOTSUP for other shapes.
// replaces: F.linear / its autograd GEMMs for EqualLR(nn.Linear(65536, 512)), gans/models/dusty_v2.py:381-383.
extern "C" int dgv2_gemm_x3(float* c, float* scratch, int64_t scratch_elems, const float* a, const float* b, int I,
                            int J, int64_t T, int a_trans, int b_trans, int64_t lda, int64_t ldb, int64_t ldo,
                            int splits, float scale, void* stream) {
  if (!c || !a || !b || I <= 0 || J <= 0 || T <= 0) return DGV2_EINVAL;
  if ((I % 64) || (J % 128) || (T % 32) || (lda & 3) || (ldb & 3) || ldo < J) return DGV2_ENOTSUP;
  if (!aligned16(c) || !aligned16(a) || !aligned16(b)) return DGV2_EINVAL;
  if (splits < 1) splits = 1;
  const int64_t steps = T / 32;
  if (splits > steps) splits = (int)steps;
  int tchunk = (int)((steps + splits - 1) / splits) * 32;
  splits = (int)((T + tchunk - 1) / tchunk);
  X3Geom g{I, J, T, lda, ldb, ldo, tchunk, scale, 0};
  float* out = c;
  if (splits > 1) {
    if (ldo != J) return DGV2_ENOTSUP;
    if (!scratch || !aligned16(scratch) || scratch_elems < (int64_t)splits * I * J) return DGV2_EINVAL;
    g.part_stride = (int64_t)I * J;
    out = scratch;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool m4 = I % 128 == 0;
  int rc;
#define DGV2_X3(AT, BT) (m4 ? x3_launch<AT, BT, 4>(out, a, b, g, splits, st) : x3_launch<AT, BT, 2>(out, a, b, g, splits, st))
  if (!a_trans && !b_trans) rc = m4 ? x3_launch<false, false, 4>(out, a, b, g, splits, st) : x3_launch<false, false, 2, true>(out, a, b, g, splits, st);
  else if (!a_trans) rc = DGV2_X3(false, true);
  else if (!b_trans) rc = DGV2_X3(true, false);
  else rc = DGV2_X3(true, true);
#undef DGV2_X3
  if (rc) return rc;
  if (splits > 1) {
    const int64_t n = (int64_t)I * J;
    x3_reduce_kernel<<<(int)((n / 4 + 15) / 16), 256, 0, st>>>(c, scratch, n, splits);
  }
  DGV2_RETURN_LAST();
}
