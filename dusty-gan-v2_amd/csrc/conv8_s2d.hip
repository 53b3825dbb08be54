// Eight-wave form of the stride-2 3x3 DATA GRADIENT (ring padding along W, replicate along H, channels-last, bf16):
// the deeper residual blocks of the discriminator, whose four-class conv_pipe_kernel launches re-stage the weight slab
// [9 taps][32 c] for every (4 x 32 pixel tile, 64-byte chunk of gy) stage -- 81 FLOP per staged byte, 85-97 us for
// 38.7 GFLOP (profiles/round5_microbench_tables.txt).
//   reference: the data gradient autograd derives for ops.Conv2d (gans/models/ops/common.py:187-210) at
//   ResidualBlock.conv2 (3x3, stride 2, behind the blur), gans/models/dusty_v2.py:337-345.
//
//   gx[b, 2i + py, 2j + px, c] = sum_{(dy, ky) in T(py)} sum_{(dx, kx) in T(px)} sum_o gy[b, i + dy, wrap(j + dx), o] wt[c, ky*3 + kx, o]
//   T(0) = {(0, 1)}, T(1) = {(1, 0), (0, 2)}   (conv.py:_axis_taps_s2; gy rows past the last one are zero),
//   + the replicate row of the forward padding: output row 0 sees gy row 0 once more through the ky = 0 weights.
//
// Two forms.  FOUR-ROW TILES, ALL FOUR CLASSES IN ONE LAUNCH (PY = 2: 128 accumulator registers fit; the border row's terms
// reuse the ky = 0 weights of the odd rows' taps): the form for 4- and 8-row maps and small batches.  EIGHT-ROW TILES need
// ONE LAUNCH PER OUTPUT ROW PARITY py: its two column classes (px = 0, 1) need 3 (py = 0: + 3 for the border row) or 6 taps,
// two accumulator sets instead of four -- which is what lets a block be conv8.hip's: eight waves = two 64-channel slabs of
// the gradient on one (4 RW) x 32 tile of gy pixels (+ one halo row and column), the tile staged once for 128 channels,
// six weight taps per slab and chunk (49 KB) instead of nine.  py = 1 stages 68 KB per 768 MFMAs (185 FLOP per byte),
// py = 0 44 KB per 384.  Everything else is conv8_kernel: four 16-byte LDS planes, taps as immediate offsets, in-place asm
// MFMAs, the next stage's loads issued a share per tap inside the MFMA loop.  A pixel's 64 channels of one slab are a whole
// 128-byte line: the two column classes need no lane exchange to store whole lines.
#include <string.h>

#include "gemm_core.h"

namespace {

struct S2D {
  int B, Hg, Wg, C, O;       // gy [B, Hg, Wg, O]; gx [B, 2 Hg, 2 Wg, C]
  int gx_, gy_, gz_, xcd;
};

// tap j of mode PY: kernel index ky * 3 + kx of wt, output class, gy offsets (dy, dx).
//   PY = 0 / 1: the two column classes of output row parity PY (class = px); PY = 0: taps 3..5 are the border row's (output
//   row 0 only).  PY = 2: all four classes (class = py * 2 + px) in one launch, four-row tiles only (128 accumulator
//   registers): the border row's terms share the ky = 0 weights of the odd rows' taps -- BCL[t] = the EVEN-row class that tap's
//   weights also feed at output row 0 (through the gy pixel at dy = 0), or -1.
template <int PY> struct S2DTaps;
template <> struct S2DTaps<0> {
  static constexpr int NTAP = 6, NCL = 2, NMAIN = 3;
  static constexpr int W[6] = {4, 3, 5, 1, 0, 2}, CL[6] = {0, 1, 1, 0, 1, 1}, DY[6] = {0, 0, 0, 0, 0, 0}, DX[6] = {0, 1, 0, 0, 1, 0};
  static constexpr int BCL[6] = {-1, -1, -1, -1, -1, -1};
};
template <> struct S2DTaps<1> {
  static constexpr int NTAP = 6, NCL = 2, NMAIN = 6;
  static constexpr int W[6] = {1, 7, 0, 2, 6, 8}, CL[6] = {0, 0, 1, 1, 1, 1}, DY[6] = {1, 0, 1, 1, 0, 0}, DX[6] = {0, 0, 1, 0, 1, 0};
  static constexpr int BCL[6] = {-1, -1, -1, -1, -1, -1};
};
template <> struct S2DTaps<2> {
  static constexpr int NTAP = 9, NCL = 4, NMAIN = 9;
  static constexpr int W[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8}, CL[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
  static constexpr int DY[9] = {0, 0, 0, 1, 0, 1, 1, 0, 0}, DX[9] = {0, 1, 0, 0, 0, 1, 0, 1, 0};
  static constexpr int BCL[9] = {-1, -1, -1, 0, -1, 1, 1, -1, -1};
};

template <int PY, int RW>
struct S2DCfg {
  static constexpr int TH = 4 * RW, NF = 2 * RW, MF = 4;
  static constexpr int IROWS = TH + 1, ICOLS = 33, NPIX = IROWS * ICOLS;
  static constexpr int NI = (NPIX * 4 + 511) / 512;
  static constexpr int COVER = NI * 128;
  static constexpr int PIN = (NPIX + (COVER > NPIX ? COVER - NPIX : 0) + 15) / 16 * 16;   // dead slots land behind the image
  static constexpr int NTAP = S2DTaps<PY>::NTAP, PW = NTAP * 64;
  static constexpr size_t LDS = sizeof(uint4) * 4 * ((size_t)PIN + 2 * PW);
};

template <int PY, int RW>
__global__ __launch_bounds__(512, 2) void conv8_s2d_kernel(bf16_t* __restrict__ gx, const bf16_t* __restrict__ gy,
                                                          const bf16_t* __restrict__ wt, S2D p) {
  using Cf = S2DCfg<PY, RW>;
  using Tp = S2DTaps<PY>;
  constexpr int NCL = Tp::NCL;
  static_assert(PY != 2 || RW == 1, "four classes: four-row tiles (128 accumulator registers)");
  constexpr int TH = Cf::TH, NF = Cf::NF, MF = Cf::MF, ICOLS = Cf::ICOLS, NPIX = Cf::NPIX, NI = Cf::NI, PIN = Cf::PIN;
  constexpr int NTAP = Cf::NTAP, PW = Cf::PW;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];

  const int tid = threadIdx.x;
  const int t256 = tid & 255, gm = tid >> 8;
  const int wave4 = (tid >> 6) & 3, lane = tid & 63;
  const int lr = lane & 15, lc = lane >> 4;
  uint4* const lds_in = smem;
  uint4* const lds_w = smem + 4 * PIN + gm * 4 * PW;

  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd) {   // blocks that share one gy tile get ids 8 apart: same XCD, one fetch into that L2 (conv_direct.hip)
    const int n = blockIdx.x, q = n >> 3;
    const int pt = (q / p.gz_) * 8 + (n & 7);
    if (pt >= p.gx_ * p.gy_) return;
    bz = q % p.gz_;
    bx = pt % p.gx_;
    by = pt / p.gx_;
  }
  const int tiles_h = p.Hg / TH;
  const int b = by / tiles_h;
  const int h0 = (by % tiles_h) * TH;
  const int w0 = bx * 32;
  const int c0 = (bz * 2 + gm) * 64;
  const bf16_t* gyb = gy + (int64_t)b * p.Hg * p.Wg * p.O;
  const int nchunks = p.O / 32;

  // ---- staging slots ----
  const int in_plane = (tid >> 3) & 3;
  const int in_pix0 = ((tid >> 5) << 3) | (tid & 7);            // slot j: + 128 j
  int goff[NI], lrow[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int pix = in_pix0 + 128 * j;
    const int pl = pix < NPIX ? pix : NPIX - 1;                 // dead slots re-load the last pixel ...
    const int iy = pl / ICOLS, ix = pl - iy * ICOLS;
    int gh = h0 + iy;
    gh = gh >= p.Hg ? p.Hg - 1 : gh;                            // the row past the image: read by dead taps only
    int gw = w0 + ix;
    gw = gw >= p.Wg ? gw - p.Wg : gw;                           // ring
    goff[j] = (gh * p.Wg + gw) * p.O + in_plane * 8;
    lrow[j] = pix < NPIX ? pix : NPIX + (pix - NPIX);           // ... into rows behind the image
  }
  const int w_plane = (t256 >> 3) & 3;
  const int w_row0 = ((t256 >> 5) << 3) | (t256 & 7);           // row (channel of the slab) of this thread's weight slots
  const unsigned wlane = (unsigned)((c0 + w_row0) * 9 * p.O + w_plane * 8);   // tap j: + W[j] * O

  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 rin[NI], rwt[NTAP];
  // slots [k0, k1) of a stage's NI + NTAP loads (input slots first)
  auto issue = [&](int kc0, int k0, int k1) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
      if (j >= k0 && j < k1) rin[j] = *reinterpret_cast<const u32x4*>(gyb + kc0 + (unsigned)goff[j]);
#pragma unroll
    for (int j = 0; j < NTAP; ++j)
      if (NI + j >= k0 && NI + j < k1) rwt[j] = *reinterpret_cast<const u32x4*>(wt + kc0 + Tp::W[j] * p.O + wlane);
  };
  uint4* const st_in = lds_in + in_plane * PIN;
  uint4* const st_w = lds_w + w_plane * PW + w_row0;

  f32x4 acc[NCL][MF][NF];
#pragma unroll
  for (int cl = 0; cl < NCL; ++cl)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        acc[cl][mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(acc[cl][mf][nf]));   // written here, not between the first stage's MFMAs (conv8.hip)
      }

  int bpix[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) bpix[nf] = (wave4 * RW + (nf >> 1)) * ICOLS + (nf & 1) * 16 + lr;
  const uint4* const a_base = lds_w + lc * PW + lr;
  const uint4* const b_base = lds_in + lc * PIN;

  // per output row of this wave: PY = 1: the dy = 1 taps read the zero row below the image (dead); PY = 0: the border
  // taps 3..5 belong to output row 0 only (dead everywhere else)
  unsigned dead[RW];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int gi = h0 + wave4 * RW + rr;
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if (Tp::DY[t] == 1 && gi + 1 >= p.Hg) m |= 1u << t;
      if (t >= Tp::NMAIN && gi != 0) m |= 1u << t;
    }
    if (PY == 2 && gi == 0) m |= 1u << 16;   // bit 16: this row is output row 0 (the border terms of BCL apply)
    dead[rr] = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
  }

  issue(0, 0, NI + NTAP);
  for (int s = 0; s < nchunks; ++s) {
    __syncthreads();                // every wave has finished reading stage s-1
#pragma unroll
    for (int j = 0; j < NI; ++j) *reinterpret_cast<u32x4*>(st_in + lrow[j]) = rin[j];
#pragma unroll
    for (int j = 0; j < NTAP; ++j) *reinterpret_cast<u32x4*>(st_w + 64 * j) = rwt[j];
    const bool more = s + 1 < nchunks;
    __syncthreads();                // stage s visible in LDS

    uint4 a[2][MF], bb[NF];
    unsigned dd[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      dd[rr] = dead[rr];
      asm volatile("" : "+s"(dd[rr]));   // the bit tests stay in the loop (conv8.hip)
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) a[0][mf] = a_base[mf * 16];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bb[nf] = b_base[bpix[nf] + Tp::DY[0] * ICOLS + Tp::DX[0]];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        __builtin_amdgcn_sched_barrier(0);
        if (!((dd[nf >> 1] >> t) & 1u)) {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            if (mf < MF - 1) MfmaAsm<bf16_t>::run(acc[Tp::CL[t]][mf][nf], a[t & 1][mf], bb[nf]);
            else MfmaAsm<bf16_t>::run_pad(acc[Tp::CL[t]][mf][nf], a[t & 1][mf], bb[nf]);   // compiler code may follow
          }
        }
        // replicate-row border term (PY = 2): at output row 0 the even-row class BCL sees gy row 0 once more through THIS
        // tap's weights (in registers right now) -- one more pixel-fragment read (dy = 0 instead of 1).  (t is a constant
        // after unrolling: the taps without a border class lose this block.)
        if (Tp::BCL[t] >= 0 && ((dd[nf >> 1] >> 16) & 1u)) {
          constexpr int NCLm1 = NCL - 1;
          const int bcl = Tp::BCL[t] >= 0 ? (Tp::BCL[t] < NCLm1 ? Tp::BCL[t] : NCLm1) : 0;
          const uint4 bx = b_base[bpix[nf] + Tp::DX[t]];
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            if (mf < MF - 1) MfmaAsm<bf16_t>::run(acc[bcl][mf][nf], a[t & 1][mf], bx);
            else MfmaAsm<bf16_t>::run_pad(acc[bcl][mf][nf], a[t & 1][mf], bx);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nf == NF - 1 && more) {   // this tap's share of the next stage's loads (conv8.hip)
          issue((s + 1) * 32, t * (NI + NTAP) / NTAP, (t + 1) * (NI + NTAP) / NTAP);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < NTAP) {
          bb[nf] = b_base[bpix[nf] + Tp::DY[t + 1] * ICOLS + Tp::DX[t + 1]];
          constexpr int APG = MF / NF;   // A fragments re-read per pixel-fragment group
#pragma unroll
          for (int k = 0; k < APG; ++k) a[(t + 1) & 1][nf * APG + k] = a_base[(t + 1) * 64 + (nf * APG + k) * 16];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: class cl of output row 2 i + PY is column 2 j + cl; a pixel's 64 channels of this slab = one line ----
  mfma_drain();
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int gi = h0 + wave4 * RW + (nf >> 1);
    const int gj = w0 + (nf & 1) * 16 + lr;
#pragma unroll
    for (int cl = 0; cl < NCL; ++cl) {
      const int py = PY == 2 ? (cl >> 1) : PY, px = PY == 2 ? (cl & 1) : cl;
      bf16_t* const row = gx + ((((int64_t)b * 2 * p.Hg + 2 * gi + py) * 2 * p.Wg) + 2 * gj + px) * p.C + c0;
#pragma unroll
      for (int mf = 0; mf < MF; mf += 2) {
        float fa[4], fb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          fa[r] = acc[cl][mf][nf][r];
          fb[r] = acc[cl][mf + 1][nf][r];
        }
        uint4 pk;
        const int co = pack_pair_bf16(fa, fb, lc, pk);   // every lane takes part in the exchange
        *reinterpret_cast<uint4*>(row + mf * 16 + co) = pk;
      }
    }
  }
}

template <int PY, int RW>
int launch_s2d(void* gx, const void* gy, const void* wt, S2D p, hipStream_t st) {
  using Cf = S2DCfg<PY, RW>;
  static_assert(Cf::LDS <= 160 * 1024 - 1024, "LDS image");
  static_assert(Cf::MF % Cf::NF == 0, "A fragments are re-read in equal shares behind the pixel-fragment groups");
  auto kern = conv8_s2d_kernel<PY, RW>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cf::LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(p.Wg / 32, (p.Hg / Cf::TH) * p.B, p.C / 128);
  p.xcd = 0;
  if (grid.z >= 2) {
    p.xcd = 1;
    p.gx_ = grid.x; p.gy_ = grid.y; p.gz_ = grid.z;
    const int64_t npt = ((int64_t)grid.x * grid.y + 7) / 8 * 8;
    grid = dim3((unsigned)(npt * grid.z), 1, 1);
  }
  kern<<<grid, 512, Cf::LDS, st>>>((bf16_t*)gx, (const bf16_t*)gy, (const bf16_t*)wt, p);
  return 0;
}

}  // namespace

// The stride-2 data gradient of the 3x3 ring conv (pad 1) on the transposed row weights wt [C, 9, O] (the weight bank's wt):
//   gx [B, 2 Hg, 2 Wg, C] (bf16) from gy [B, Hg, Wg, O], the replicate row of output row 0 included; two launches (output row
//   parities).  DGV2_ENOTSUP where the engine does not cover the geometry (C % 128, O % 32, O >= 64, Hg % 4, Wg % 32, bf16):
//   callers then run dgv2_conv_taps_ex.
extern "C" int dgv2_conv3x3_s2_dgrad8(void* gx, const void* gy, const void* wt, int B, int Hg, int Wg, int C, int O, int dtype,
                                      void* stream) {
  if (!gx || !gy || !wt || B < 1 || Hg < 1 || Wg < 1 || C < 1 || O < 1) return DGV2_EINVAL;
  static const bool off = getenv("DGV2_NO_S2D8") != nullptr;   // A/B switch for benchmarking
  if (off || dtype != DGV2_BF16) return DGV2_ENOTSUP;
  if (C % 128 || O % 32 || O < 64 || Hg % 4 || Wg % 32) return DGV2_ENOTSUP;
  if (!aligned16(gx) || !aligned16(gy) || !aligned16(wt)) return DGV2_ENOTSUP;
  if ((int64_t)Hg * Wg * O >= (1ll << 31) || (int64_t)C * 9 * O >= (1ll << 31)) return DGV2_ENOTSUP;
  S2D p;
  p.B = B; p.Hg = Hg; p.Wg = Wg; p.C = C; p.O = O; p.gx_ = p.gy_ = p.gz_ = 1; p.xcd = 0;
  hipStream_t st = (hipStream_t)stream;
  // eight-row tiles while they still fill the chip, four-row tiles otherwise
  const int64_t blocks8 = Hg % 8 ? 0 : (int64_t)(Wg / 32) * (Hg / 8) * B * (C / 128);
  const int64_t blocks4 = (int64_t)(Wg / 32) * (Hg / 4) * B * (C / 128);
  // measured (profiles/round5_mb_s2d.txt, B = 128 / 64, against conv_pipe_kernel's four-class launch): two launches on
  // eight-row tiles that fill the chip 91 -> 65 us; ONE four-class launch on four-row tiles 87 -> 61, 49 -> 33 us (two launches
  // on four-row tiles: level with conv_pipe); half a chip of blocks 43 -> 75 us: left to conv_pipe
  if (blocks4 < 256) return DGV2_ENOTSUP;
  static const int mode = getenv("DGV2_S2D8_MODE") ? atoi(getenv("DGV2_S2D8_MODE")) : 0;   // experiments: 1 two launches, 2 four classes
  int rc;
  const bool four = mode == 2 || (mode == 0 && blocks8 < 256);
  if (four) {
    rc = launch_s2d<2, 1>(gx, gy, wt, p, st);
  } else if (blocks8 >= 256) {
    rc = launch_s2d<0, 2>(gx, gy, wt, p, st);
    if (!rc) rc = launch_s2d<1, 2>(gx, gy, wt, p, st);
  } else {
    rc = launch_s2d<0, 1>(gx, gy, wt, p, st);
    if (!rc) rc = launch_s2d<1, 1>(gx, gy, wt, p, st);
  }
  if (rc) return rc;
  DGV2_RETURN_LAST();
}
