// Streaming weight gradient of the ring-padded 3x3 / 1x1 convolutions of the discriminator.
//   gw[o, ky, kx, c] = sum_{b, ho, wo} gy[b, ho, wo, o] * xpad[b, ho*s + ky, wo*s + kx, c]
// reference: the weight gradient autograd derives for ops.Conv2d (gans/models/ops/common.py:187-210 at
// gans/models/dusty_v2.py:325-385).
//
// The reduction axis (all output pixels of the batch) is ~10^5..10^6 long while the result is tiny, so the
// kernel is organised around the RESULT: a block owns a (16*MFN) x (16*NFN) (o, c) tile for ALL k*k taps and
// keeps those 9 * MFN * NFN accumulator fragments in registers while it streams its slice of the pixel
// tiles (WR rows x 32 columns each) through LDS: gy rows [pix][o] and the input halo tile [pix][c], both
// pixel-major as they sit in HBM, read back as MFMA operands with the transposed LDS read (TnFrag).
// Every halo tile feeds all nine taps.  The (tap, c-fragment) units are dealt round-robin to the four
// waves and each unit keeps all MFN o-fragments, so one B fragment read feeds MFN MFMAs.
// Staging is software-pipelined like conv_pipe_kernel: the global loads of tile t+1 are in flight during
// the MFMAs of tile t.  Blocks split the pixel tiles (split-K); partial results go to a scratch buffer
// with plain stores and a second kernel sums them (no atomics, no zero-fill of gw).
#include "gemm_core.h"

namespace {

struct WSGeom {
  int B, H, W, C, O, Ho, Wo, k, stride, pad, ring;
  int tiles_h, tiles_w, ntiles, tiles_per_split, ctiles;
#ifdef DGV2_ABLATE   // benchmarking builds only (make ABLATE=1): wrong results by design, never in the shipped library
  int ablate;        // DGV2_WS_ABLATE: 1 skip the partial stores, 2 skip the MFMA loop
#define WS_ABL (g.ablate)
#else
#define WS_ABL 0
#endif
  int x_shared; // x is ONE image [H, W, C] shared by all samples (the batch-shared positional encoding)
  int x_exact;  // conv_wgrad_x3_kernel: channels [0, x_exact) of x are bf16-representable (dgv2.h: dgv2_conv3x3_x3_fwd)
  int* status;  // ... and the caller's device status word a breach of that promise is reported to (dgv2.h: status words)
};

template <typename T, int S, int MFN, int NFN, bool K3>
struct WSCfg {
  static constexpr int KS = TnFrag<T>::KS;
  static constexpr int CE = 16 / sizeof(T);
  static constexpr int TO = 16 * MFN, TC = 16 * NFN;
  // LDS row lengths: a transposed fp32 fragment read touches 4 pixel rows x 16 consecutive channels per instruction;
  // rows of 32 / 64 floats put those 4 rows on the same banks (2- / 4-way conflicts), +16 floats spreads them over
  // all 64 banks
  static constexpr int PAD = sizeof(T) == 4 ? 16 : 0;
  static constexpr int RO = TO + PAD, RC = TC + PAD;
  static constexpr int WR = S == 1 ? 4 : 2;
  static constexpr int KK = K3 ? 3 : 1;
  static constexpr int IN_ROWS = (WR - 1) * S + KK, IN_COLS = 31 * S + KK;
  static constexpr int GV = TO / CE, XV = TC / CE;
  static constexpr int NG = (WR * 32 * GV + 255) / 256, NX = (IN_ROWS * IN_COLS * XV + 255) / 256;
  static constexpr int UNITS = (K3 ? 9 : 1) * NFN, UPW = (UNITS + 3) / 4;
  static constexpr size_t LDS = sizeof(T) * ((size_t)WR * 32 * RO + (size_t)IN_ROWS * IN_COLS * RC);
};

template <typename T, int S, int MFN, int NFN, bool K3>
__global__ __launch_bounds__(256, 1) void conv_wgrad_stream_kernel(float* __restrict__ part, const T* __restrict__ gy,
                                                                   const T* __restrict__ x, WSGeom g) {
  using Cf = WSCfg<T, S, MFN, NFN, K3>;
  constexpr int KS = Cf::KS, CE = Cf::CE, TO = Cf::TO, TC = Cf::TC, WR = Cf::WR;
  constexpr int GV = Cf::GV, XV = Cf::XV, NG = Cf::NG, NX = Cf::NX, UPW = Cf::UPW, UNITS = Cf::UNITS;
  constexpr int IN_COLS = Cf::IN_COLS, KK = Cf::KK, RO = Cf::RO, RC = Cf::RC;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  T* lds_gy = reinterpret_cast<T*>(smem);
  T* lds_x = lds_gy + WR * 32 * RO;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int split = blockIdx.x;
  const int c0 = (blockIdx.y % g.ctiles) * TC;
  const int o0 = (blockIdx.y / g.ctiles) * TO;
  constexpr int ntaps = KK * KK;
  constexpr int n_x = Cf::IN_ROWS * IN_COLS * XV;
  const int off = g.pad;

  // ---- per-slot constants ----
  int gpix[NG];                  // (row << 8) | col of the gy slot's pixel inside the tile, -1 = dead slot
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int id = tid + j * 256;
    const int pix = id / GV, ch = id % GV;
    gpix[j] = (id < WR * 32 * GV && o0 + ch * CE < g.O) ? (((pix >> 5) << 8) | (pix & 31)) : -1;
  }
  int xpos[NX];                  // (iy << 16) | ix of the x slot's pixel inside the halo tile, -1 = dead slot
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int id = tid + j * 256;
    const int pix = id / XV, ch = id % XV;
    const int iy = pix / IN_COLS;
    xpos[j] = (id < n_x && c0 + ch * CE < g.C) ? ((iy << 16) | (pix - iy * IN_COLS)) : -1;
  }

  uint4 rg[NG], rx[NX];
  int tw_n, th_n, b_n;           // (column tile, row tile, image) of the NEXT tile to load
  auto issue = [&]() {           // global loads of that tile into registers, then advance
    const int h0 = th_n * WR, w0 = tw_n * 32;
    const T* gyb = gy + (int64_t)b_n * g.Ho * g.Wo * g.O + o0;
    const T* xb = x + (g.x_shared ? 0 : (int64_t)b_n * g.H * g.W * g.C) + c0;
    if (++tw_n == g.tiles_w) {
      tw_n = 0;
      if (++th_n == g.tiles_h) { th_n = 0; ++b_n; }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int id = tid + j * 256;
      const int ho = h0 + (gpix[j] >> 8), wo = w0 + (gpix[j] & 255);
      rg[j] = make_uint4(0, 0, 0, 0);
      if (gpix[j] >= 0 && ho < g.Ho && wo < g.Wo)
        rg[j] = *reinterpret_cast<const uint4*>(gyb + (ho * g.Wo + wo) * g.O + (id % GV) * CE);
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int id = tid + j * 256;
      int hi = h0 * S - off + (xpos[j] >> 16), wi = w0 * S - off + (xpos[j] & 0xffff);
      hi = hi < 0 ? 0 : (hi >= g.H ? g.H - 1 : hi);
      if (g.ring == 2) wi = wi < 0 ? wi + g.W : (wi >= g.W ? wi - g.W : wi);   // host-checked: one wrap suffices
      else wi = g.ring ? floormod(wi, g.W) : (wi < 0 ? 0 : (wi >= g.W ? g.W - 1 : wi));
      rx[j] = make_uint4(0, 0, 0, 0);
      if (xpos[j] >= 0) rx[j] = *reinterpret_cast<const uint4*>(xb + (hi * g.W + wi) * g.C + (id % XV) * CE);
    }
  };

  f32x4 acc[UPW][MFN];
#pragma unroll
  for (int ui = 0; ui < UPW; ++ui)
#pragma unroll
    for (int mf = 0; mf < MFN; ++mf) acc[ui][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-unit LDS offsets (uniform per wave): tap -> (ky, kx), c-fragment.  Units past the end (only when
  // UNITS is not a multiple of 4) alias unit 0 and accumulate garbage that is never stored.
  int uoff[UPW], unf[UPW];
#pragma unroll
  for (int ui = 0; ui < UPW; ++ui) {
    int u = wave + ui * 4;
    u = u < UNITS ? u : 0;
    const int tap = u / NFN;
    const int ky = tap / KK, kx = tap - ky * KK;
    uoff[ui] = (ky * IN_COLS + kx) * RC;
    unf[ui] = (u % NFN) * 16;
  }

  const int t_begin = split * g.tiles_per_split;
  const int t_end = min(t_begin + g.tiles_per_split, g.ntiles);
  tw_n = t_begin % g.tiles_w;
  th_n = (t_begin / g.tiles_w) % g.tiles_h;
  b_n = t_begin / (g.tiles_w * g.tiles_h);
  if (t_begin < t_end) issue();
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();             // every wave has finished reading tile t-1
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int id = tid + j * 256;
      if (id < WR * 32 * GV) *reinterpret_cast<uint4*>(lds_gy + (id / GV) * RO + (id % GV) * CE) = rg[j];
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int id = tid + j * 256;
      if (id < n_x) *reinterpret_cast<uint4*>(lds_x + (id / XV) * RC + (id % XV) * CE) = rx[j];
    }
    if (t + 1 < t_end) issue();
    __syncthreads();
#pragma unroll
    for (int r = 0; r < WR; ++r) {
#pragma unroll
      for (int kb = 0; kb < 32 / KS; ++kb) {
        uint4 a[MFN];
        const T* arow = lds_gy + (r * 32 + kb * KS) * RO;
        const T* xrow = lds_x + (r * S * IN_COLS + kb * KS * S) * RC;
        uint4 bnext = TnFrag<T>::template read<S * RC>(xrow + uoff[0], unf[0], lane);
#pragma unroll
        for (int mf = 0; mf < MFN; ++mf) a[mf] = TnFrag<T>::template read<RO>(arow, mf * 16, lane);
#pragma unroll
        for (int ui = 0; ui < UPW; ++ui) {
          const uint4 bb = bnext;
          if (ui + 1 < UPW) bnext = TnFrag<T>::template read<S * RC>(xrow + uoff[ui + 1], unf[ui + 1], lane);
#pragma unroll
          for (int mf = 0; mf < MFN; ++mf) Mfma16<T>::run(acc[ui][mf], a[mf], bb);
        }
      }
    }
  }

  // D layout: column (c) = lane & 15, rows (o) = 4 * (lane >> 4) + r
  const int lr = lane & 15, lc = lane >> 4;
  float* pb = part + (int64_t)split * g.O * ntaps * g.C;
#pragma unroll
  for (int ui = 0; ui < UPW; ++ui) {
    const int u = wave + ui * 4;
    if (u >= UNITS) continue;
    const int tap = u / NFN;
    const int c = c0 + unf[ui] + lr;
    if (c >= g.C) continue;
#pragma unroll
    for (int mf = 0; mf < MFN; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + mf * 16 + lc * 4 + r;
        if (o < g.O) pb[((int64_t)o * ntaps + tap) * g.C + c] = acc[ui][mf][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 production kernel.  Same algorithm, two refinements that matter at one wave per SIMD:
//  * conflict-free LDS images.  A transposed fragment read touches the same 32-byte column chunk of 8 pixel
//    rows per 32-lane group; with 64/128-byte rows those rows share banks (4-way conflicts on every read).
//    The 32-byte chunks of a row are therefore XOR-swizzled with bits of the row's COLUMN index (bits 1,3 for
//    128-byte rows, bit 3 for 64-byte rows), and for stride 2 the halo tile is stored as two column planes
//    (even / odd input columns) so that a fragment's pixels are consecutive in LDS for both strides.
//  * compile-time taps.  Wave w owns c-fragment w (NFN = 4) or (o-half, c-fragment) (NFN = 2) for ALL taps,
//    so (ky, kx) are constants of the unrolled MFMA loop: every read is `per-lane offset + immediate`, the
//    per-lane offsets (3 column shifts x 2 halves + the A offsets) are computed once per block.
// ---------------------------------------------------------------------------------------------
// GO = 2 (round 4): EIGHT waves = two groups of four on two neighbouring 64-channel o-slabs and the SAME c-slab: the
// input halo tile -- at stride 2 a 5 x 65 pixel tile, 42 of the 50 KB a block stages per pixel tile -- is staged once
// for 128 output channels (the kernel is bound by its staging, not by the MFMAs: profiles/round4_s2_ablation_start.txt).
template <int S, int MFN, int NFN, bool K3, int GO = 1>
struct WZCfg {
  static constexpr int NTHR = 256 * GO;
  static constexpr int TO = 16 * MFN, TC = 16 * NFN;
  static constexpr int WR = S == 1 ? 4 : 2;
  static constexpr int KK = K3 ? 3 : 1;
  static constexpr int IN_ROWS = (WR - 1) * S + KK, IN_COLS = 31 * S + KK;
  static constexpr int PL = S;                        // column planes per halo row
  static constexpr int PC = (IN_COLS + S - 1) / S;    // columns per plane
  static constexpr int GV = TO / 8, XV = TC / 8;      // 16-byte slots per pixel (gy: per o-group)
  static constexpr int NG = (WR * 32 * GV * GO + NTHR - 1) / NTHR, NX = (IN_ROWS * IN_COLS * XV + NTHR - 1) / NTHR;
  static constexpr int MW = NFN == 4 ? MFN : MFN / 2; // o-fragments per wave
  static constexpr int NT = KK * KK;
  static constexpr int NCS = K3 ? (S == 1 ? 3 : 2) : 1;   // distinct column shifts of the taps
  static constexpr int GY_BYTES = WR * 32 * TO * 2;       // one o-group's gy image
  static constexpr size_t LDS = (size_t)GO * GY_BYTES + (size_t)IN_ROWS * PL * PC * TC * 2;
};

template <int ROWB>
__device__ __forceinline__ int wz_swz(int col) {
  if constexpr (ROWB == 128) return ((col >> 1) & 1) | (((col >> 3) & 1) << 1);
  else return (col >> 3) & 1;
}

__device__ __forceinline__ uint4 wz_tr_read(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union { uint4 u; s16x4 h[2]; } r;
  r.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  r.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
  return r.u;
}

#ifndef DGV2_WS_SPREAD
#define DGV2_WS_SPREAD 1
#endif
template <int S, int MFN, int NFN, bool K3, int GO>
__global__ __launch_bounds__(256 * GO, 2) void conv_wgrad_stream_bf16_kernel(float* __restrict__ part,
                                                                             const bf16_t* __restrict__ gy,
                                                                             const bf16_t* __restrict__ x, WSGeom g) {
  using Cf = WZCfg<S, MFN, NFN, K3, GO>;
  constexpr int TO = Cf::TO, TC = Cf::TC, WR = Cf::WR, KK = Cf::KK, IN_COLS = Cf::IN_COLS, PL = Cf::PL, PC = Cf::PC;
  constexpr int GV = Cf::GV, XV = Cf::XV, NG = Cf::NG, NX = Cf::NX, MW = Cf::MW, NT = Cf::NT, NCS = Cf::NCS;
  constexpr int NTHR = Cf::NTHR, GVT = GV * GO;
  constexpr int n_x = Cf::IN_ROWS * IN_COLS * XV;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* lds_gy = smem;
  uint4* lds_x = smem + GO * Cf::GY_BYTES / 16;
  const char* lbase = reinterpret_cast<const char*>(smem);

  const int tid = threadIdx.x;
  const int wave = (tid >> 6) & 3, grp = tid >> 8, lane = tid & 63;   // grp: the o-group (64-channel slab) of this wave
  const int split = blockIdx.x;
  const int c0 = (blockIdx.y % g.ctiles) * TC;
  const int o0 = (blockIdx.y / g.ctiles) * TO * GO;
  const int off = g.pad;
  const int nf = NFN == 4 ? wave : (wave & 1);
  const int m0w = NFN == 4 ? 0 : (wave >> 1) * MW;

  // ---- per-slot constants: pixel of the slot, and its (swizzled) LDS position ----
  int gpix[NG], gw_[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int id = tid + j * NTHR;
    const int pix = id / GVT, c16t = id % GVT;        // a pixel's GO * TO channels are contiguous in gy
    const int c16 = c16t % GV;
    gpix[j] = (id < WR * 32 * GVT && o0 + c16t * 8 < g.O) ? (((pix >> 5) << 8) | (pix & 31)) : -1;
    gw_[j] = (c16t / GV) * (Cf::GY_BYTES / 16) + pix * GV + ((((c16 >> 1) ^ wz_swz<TO * 2>(pix & 31)) << 1) | (c16 & 1));
  }
  int xpos[NX], xw_[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int id = tid + j * NTHR;
    const int pix = id / XV, c16 = id % XV;
    const int iy = pix / IN_COLS, ix = pix - iy * IN_COLS;
    xpos[j] = (id < n_x && c0 + c16 * 8 < g.C) ? ((iy << 16) | ix) : -1;
    const int col = ix / S, plane = ix % S;
    xw_[j] = ((iy * PL + plane) * PC + col) * XV + ((((c16 >> 1) ^ wz_swz<TC * 2>(col)) << 1) | (c16 & 1));
  }

  uint4 rg[NG], rx[NX];
  int tw_n, th_n, b_n;           // (column tile, row tile, image) of the NEXT tile to load
  int h0_l, w0_l;                // ... of the tile being loaded
  const bf16_t *gyb_l, *xb_l;
  auto issue_begin = [&]() {     // fix the tile the coming loads belong to, then advance
    h0_l = th_n * WR; w0_l = tw_n * 32;
    gyb_l = gy + (int64_t)b_n * g.Ho * g.Wo * g.O + o0;
    xb_l = x + (g.x_shared ? 0 : (int64_t)b_n * g.H * g.W * g.C) + c0;
    if (++tw_n == g.tiles_w) {
      tw_n = 0;
      if (++th_n == g.tiles_h) { th_n = 0; ++b_n; }
    }
  };
  // slots [k0, k1) of the NG + NX loads of that tile (gy slots first).  The asm-MFMA instances issue them a share per
  // (row, tap) of the MFMA loop: a wave that issues a whole tile in one burst waits at the full memory queue until most of
  // it has returned, and its MFMA loop starts when the loads are nearly over (conv8.hip, measured there)
  auto issue_part = [&](int k0, int k1) {
    const int h0 = h0_l, w0 = w0_l;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (j < k0 || j >= k1) continue;
      const int id = tid + j * NTHR;
      const int ho = h0 + (gpix[j] >> 8), wo = w0 + (gpix[j] & 255);
      rg[j] = make_uint4(0, 0, 0, 0);
      if (gpix[j] >= 0 && ho < g.Ho && wo < g.Wo)
        rg[j] = *reinterpret_cast<const uint4*>(gyb_l + (ho * g.Wo + wo) * g.O + (id % GVT) * 8);
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      if (NG + j < k0 || NG + j >= k1) continue;
      const int id = tid + j * NTHR;
      int hi = h0 * S - off + (xpos[j] >> 16), wi = w0 * S - off + (xpos[j] & 0xffff);
      hi = hi < 0 ? 0 : (hi >= g.H ? g.H - 1 : hi);
      if (g.ring == 2) wi = wi < 0 ? wi + g.W : (wi >= g.W ? wi - g.W : wi);   // host-checked: one wrap suffices
      else wi = g.ring ? floormod(wi, g.W) : (wi < 0 ? 0 : (wi >= g.W ? g.W - 1 : wi));
      rx[j] = make_uint4(0, 0, 0, 0);
      if (xpos[j] >= 0) rx[j] = *reinterpret_cast<const uint4*>(xb_l + (hi * g.W + wi) * g.C + (id % XV) * 8);
    }
  };
  auto issue = [&]() { issue_begin(); issue_part(0, NG + NX); };
  // measured (gpurun_out/r7d): the eight-wave stride-2 instances (one block per CU, its waves in lockstep) gain 3-4 %; the
  // four-wave instances (two blocks per CU cover each other's load phase) lose 30-50 % to the predicated loads among the MFMAs
  constexpr bool SPREAD = MFN == 4 && NFN == 4 && GO == 2 && DGV2_WS_SPREAD != 0;

  f32x4 acc[NT][MW];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) acc[t][mw] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- per-lane fragment offsets (bytes from the start of LDS) ----
  const int fg = lane >> 4, fi = lane & 15;
  const int fq = fi >> 2, fp = fi & 3;
  int voffA[MW][2], voffB[NCS][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int L = 8 * fg + fq + 4 * h;          // k index (pixel column inside the tile row) of this lane's 4 values
#pragma unroll
    for (int mw = 0; mw < MW; ++mw)
      voffA[mw][h] = grp * Cf::GY_BYTES + L * TO * 2 + (((m0w + mw) ^ wz_swz<TO * 2>(L)) * 32) + 8 * fp;
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs)
      voffB[cs][h] = GO * Cf::GY_BYTES + (cs + L) * TC * 2 + ((nf ^ wz_swz<TC * 2>(cs + L)) * 32) + 8 * fp;
  }

  const int t_begin = split * g.tiles_per_split;
  const int t_end = min(t_begin + g.tiles_per_split, g.ntiles);
  tw_n = t_begin % g.tiles_w;
  th_n = (t_begin / g.tiles_w) % g.tiles_h;
  b_n = t_begin / (g.tiles_w * g.tiles_h);
  if (t_begin < t_end) issue();
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();             // every wave has finished reading tile t-1
#pragma unroll
    for (int j = 0; j < NG; ++j)
      if (tid + j * NTHR < WR * 32 * GVT) lds_gy[gw_[j]] = rg[j];
#pragma unroll
    for (int j = 0; j < NX; ++j)
      if (tid + j * NTHR < n_x) lds_x[xw_[j]] = rx[j];
    const bool more = t + 1 < t_end;
    if (more) issue_begin();
    if (more && (!SPREAD || (WS_ABL & 2))) issue_part(0, NG + NX);
    __syncthreads();
    if (WS_ABL & 2) continue;
#pragma unroll
    for (int r = 0; r < WR; ++r) {
      uint4 a[MW];
#pragma unroll
      for (int mw = 0; mw < MW; ++mw)
        a[mw] = wz_tr_read(lbase + voffA[mw][0] + r * 32 * TO * 2, lbase + voffA[mw][1] + r * 32 * TO * 2);
      // tap -> (row, plane, column shift) immediates
      auto bread = [&](int tap) {
        const int ky = tap / KK, kx = tap % KK;
        const int plane = kx % S, cs = kx / S;
        const int imm = ((r * S + ky) * PL + plane) * PC * TC * 2;
        return wz_tr_read(lbase + voffB[cs][0] + imm, lbase + voffB[cs][1] + imm);
      };
      uint4 bnext = bread(0);
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        const uint4 bb = bnext;
        if (tap + 1 < NT) bnext = bread(tap + 1);
#pragma unroll
        for (int mw = 0; mw < MW; ++mw) {
          // 64 x 64 tiles in place (inline asm): with the builtin the compiler double-books the 144 accumulator registers
          // (410 VGPRs + AGPRs: one wave per SIMD); in place the kernel fits two blocks per CU
          if constexpr (MFN == 4 && NFN == 4) {
            union { uint4 u; bf16x8 v; } ua, ub;
            ua.u = a[mw];
            ub.u = bb;
            // (the last MFMA of a tap carries five wait states: the loads below bring compiler code -- zero fills,
            // address arithmetic, exec masks -- right behind it; scripts/audit_asm_mfma.py)
            if (mw < MW - 1 || !SPREAD)
              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[tap][mw]) : "v"(ua.v), "v"(ub.v));
            else
              asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 4" : "+v"(acc[tap][mw]) : "v"(ua.v), "v"(ub.v));
          } else {   // the smaller tiles have registers to spare and schedule better with the builtin
            Mfma16<bf16_t>::run(acc[tap][mw], a[mw], bb);
          }
        }
        if constexpr (SPREAD) {
          if (more) {
            constexpr int NSLOT = WR * NT;
            const int k = r * NT + tap;
            __builtin_amdgcn_sched_barrier(0);
            issue_part(k * (NG + NX) / NSLOT, (k + 1) * (NG + NX) / NSLOT);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
  // every MFMA above has written its accumulator before the stores below read it (the hazard recogniser does not see
  // inside the asm)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // D layout: column (c) = lane & 15, rows (o) = 4 * (lane >> 4) + r
  const int lr = lane & 15, lc = lane >> 4;
  float* pb = part + (int64_t)split * g.O * NT * g.C;
  const int c = c0 + nf * 16 + lr;
  if (c < g.C && !((WS_ABL & 1) && acc[0][0][0] != 12345.678f)) {
#pragma unroll
    for (int tap = 0; tap < NT; ++tap)
#pragma unroll
      for (int mw = 0; mw < MW; ++mw)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + grp * TO + (m0w + mw) * 16 + lc * 4 + r;
          if (o < g.O) pb[((int64_t)o * NT + tap) * g.C + c] = acc[tap][mw][r];
        }
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 on the bf16 matrix cores (conv_x3.hip's scheme: every fp32 operand value as three bf16 planes h + m + l, a product as
// six bf16 products on v_mfma_f32_16x16x32_bf16, fp32 accumulation -- fp32-equivalent at 3/8 of the fp32 MFMA's cost):
// the weight gradient of the discriminator's fp32 epilogue conv (3x3, stride 1, ring; reference: the autograd of
// ops.Conv2d(ch(4) + 1, ch(4), 3, 1, 1) at gans/models/dusty_v2.py:377 under the fp32 island of :394-395).
// The bf16 kernel above with both operands split while they are staged (two 16-byte loads -> one 16-byte unit in each of
// three LDS images of the same swizzled layout): a wave owns one 16-channel c-fragment x 64 output channels x 9 taps
// (144 accumulator registers, in place); per 32-pixel tile row it reads 12 gy fragments and 27 halo fragments for 216
// MFMAs -- 5.5 MFMAs per fragment read against the bf16 kernel's 2.8.  Input channels past the last whole 64-channel
// c-tile (the minibatch-stddev channel, 512 + 1) are left to conv_wgrad_x3_tail_kernel.
// ---------------------------------------------------------------------------------------------
struct WXCfg {
  static constexpr int WR = 4, IN_ROWS = 6, IN_COLS = 34;
  static constexpr int NG = 4, NX = (IN_ROWS * IN_COLS * 8 + 255) / 256;
  static constexpr int GYB = WR * 32 * 64 * 2, XB = IN_ROWS * IN_COLS * 64 * 2;   // one plane's image, bytes
  static constexpr size_t LDS = 3 * (size_t)GYB + 3 * (size_t)XB;
};

__device__ __forceinline__ void wx_split8(const float4& lo, const float4& hi, uint4& h, uint4& m, uint4& l) {
  const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  union { uint4 u; bf16_t e[8]; } ph, pm, pl;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bf16_t hh = (bf16_t)f[i];
    const float r1 = f[i] - (float)hh;
    const bf16_t mm = (bf16_t)r1;
    ph.e[i] = hh;
    pm.e[i] = mm;
    pl.e[i] = (bf16_t)(r1 - (float)mm);
  }
  h = ph.u;
  m = pm.u;
  l = pl.u;
}


// XE: the c-tile's input channels are bf16-exact (WSGeom::x_exact): x = x_h, the planes m and l of x are neither staged nor
// multiplied -- three products per multiply (gy_l x_h, gy_m x_h, gy_h x_h), the same sum
template <bool XE>
__device__ __forceinline__ void wx3_tile(f32x4 (&acc)[9][4], const char* lbase, const int (&voffA)[4][2], const int (&voffB)[3][2]) {
  using Cf = WXCfg;
  constexpr int WR = Cf::WR, IN_COLS = Cf::IN_COLS, GYB = Cf::GYB, XB = Cf::XB;
  constexpr int NQ = XE ? 1 : 3;
#pragma unroll
  for (int r = 0; r < WR; ++r) {
    uint4 a[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mw = 0; mw < 4; ++mw)
        a[q][mw] = wz_tr_read(lbase + voffA[mw][0] + q * GYB + r * 32 * 128, lbase + voffA[mw][1] + q * GYB + r * 32 * 128);
    auto bread = [&](int tap, int q) {
      const int ky = tap / 3, kx = tap % 3;
      const int imm = q * XB + (r + ky) * IN_COLS * 128;
      return wz_tr_read(lbase + voffB[kx][0] + imm, lbase + voffB[kx][1] + imm);
    };
    uint4 bn[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) bn[q] = bread(0, q);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      uint4 bb[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) bb[q] = bn[q];
      if (tap + 1 < 9) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) bn[q] = bread(tap + 1, q);
      }
      // smallest terms first: gy_h x_l, gy_h x_m, gy_m x_m, gy_l x_h, gy_m x_h, gy_h x_h  (XE: the last three)
      constexpr int QA[6] = {0, 0, 1, 2, 1, 0}, QB[6] = {2, 1, 1, 0, 0, 0};
#pragma unroll
      for (int k = XE ? 3 : 0; k < 6; ++k) {
        // the four MFMAs of a (plane pair, tap) as one asm statement with eight wait states behind the last (conv_x3.hip,
        // mfma4_bf16: nothing the compiler generates can write an operand register a group's MFMA is still reading)
        union U { uint4 u; bf16x8 v; };
        U a0, a1, a2, a3, ub;
        a0.u = a[QA[k]][0]; a1.u = a[QA[k]][1]; a2.u = a[QA[k]][2]; a3.u = a[QA[k]][3]; ub.u = bb[QB[k]];
        asm volatile(
            "v_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\t"
            "v_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\t"
            "v_mfma_f32_16x16x32_bf16 %2, %6, %8, %2\n\t"
            "v_mfma_f32_16x16x32_bf16 %3, %7, %8, %3\n\t"
            "s_nop 7"
            : "+v"(acc[tap][0]), "+v"(acc[tap][1]), "+v"(acc[tap][2]), "+v"(acc[tap][3])
            : "v"(a0.v), "v"(a1.v), "v"(a2.v), "v"(a3.v), "v"(ub.v));
      }
    }
  }
}

template <bool XE>   // every c-tile of the launch lies below WSGeom::x_exact (host-checked)
__global__ __launch_bounds__(256, 1) void conv_wgrad_x3_kernel(float* __restrict__ part, const float* __restrict__ gy,
                                                               const float* __restrict__ x, WSGeom g) {
  using Cf = WXCfg;
  constexpr int WR = Cf::WR, IN_COLS = Cf::IN_COLS, NG = Cf::NG, NX = Cf::NX, GYB = Cf::GYB, XB = Cf::XB;
  constexpr int n_x = Cf::IN_ROWS * IN_COLS * 8;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* lds_gy = smem;
  uint4* lds_x = smem + 3 * GYB / 16;
  const char* lbase = reinterpret_cast<const char*>(smem);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int split = blockIdx.x;
  const int c0 = (blockIdx.y % g.ctiles) * 64;
  const int o0 = (blockIdx.y / g.ctiles) * 64;
  constexpr bool xe = XE;

  int gpix[NG], gw_[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int id = tid + j * 256;
    const int pix = id >> 3, c16 = id & 7;
    gpix[j] = ((pix >> 5) << 8) | (pix & 31);
    gw_[j] = pix * 8 + ((((c16 >> 1) ^ wz_swz<128>(pix & 31)) << 1) | (c16 & 1));
  }
  int xpos[NX], xw_[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int id = tid + j * 256;
    const int pix = id >> 3, c16 = id & 7;
    const int iy = pix / IN_COLS, ix = pix - iy * IN_COLS;
    xpos[j] = id < n_x ? ((iy << 16) | ix) : -1;
    xw_[j] = (iy * IN_COLS + ix) * 8 + ((((c16 >> 1) ^ wz_swz<128>(ix)) << 1) | (c16 & 1));
  }

  float4 rg[NG][2], rx[NX][2];
  int tw_n, th_n, b_n;
  auto issue = [&]() {
    const int h0 = th_n * WR, w0 = tw_n * 32;
    const float* gyb = gy + (int64_t)b_n * g.Ho * g.Wo * g.O + o0;
    const float* xb = x + (int64_t)b_n * g.H * g.W * g.C + c0;
    if (++tw_n == g.tiles_w) {
      tw_n = 0;
      if (++th_n == g.tiles_h) { th_n = 0; ++b_n; }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int id = tid + j * 256;
      int ho = h0 + (gpix[j] >> 8), wo = w0 + (gpix[j] & 255);
      const bool ok = ho < g.Ho && wo < g.Wo;
      ho = ok ? ho : 0;                       // (a conditional load with a zero alternative: see conv_x3.hip)
      wo = ok ? wo : 0;
      const float4* src = reinterpret_cast<const float4*>(gyb + (ho * g.Wo + wo) * g.O + (id & 7) * 8);
      rg[j][0] = src[0];
      rg[j][1] = src[1];
      if (!ok) rg[j][0] = rg[j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int id = tid + j * 256;
      const int xp = xpos[j] < 0 ? 0 : xpos[j];
      int hi = h0 - 1 + (xp >> 16), wi = w0 - 1 + (xp & 0xffff);
      hi = hi < 0 ? 0 : (hi >= g.H ? g.H - 1 : hi);
      wi = wi < 0 ? wi + g.W : (wi >= g.W ? wi - g.W : wi);        // host-checked: W % 32 == 0, one wrap suffices
      const float4* src = reinterpret_cast<const float4*>(xb + (hi * g.W + wi) * g.C + (id & 7) * 8);
      rx[j][0] = src[0];
      rx[j][1] = src[1];
    }
  };

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mw = 0; mw < 4; ++mw) acc[t][mw] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fg = lane >> 4, fi = lane & 15;
  const int fq = fi >> 2, fp = fi & 3;
  int voffA[4][2], voffB[3][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int L = 8 * fg + fq + 4 * h;
#pragma unroll
    for (int mw = 0; mw < 4; ++mw) voffA[mw][h] = L * 128 + ((mw ^ wz_swz<128>(L)) * 32) + 8 * fp;
#pragma unroll
    for (int cs = 0; cs < 3; ++cs) voffB[cs][h] = 3 * GYB + (cs + L) * 128 + ((wave ^ wz_swz<128>(cs + L)) * 32) + 8 * fp;
  }

  const int t_begin = split * g.tiles_per_split;
  const int t_end = min(t_begin + g.tiles_per_split, g.ntiles);
  tw_n = t_begin % g.tiles_w;
  th_n = (t_begin / g.tiles_w) % g.tiles_h;
  b_n = t_begin / (g.tiles_w * g.tiles_h);
  if (t_begin < t_end) issue();
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();             // every wave has finished reading tile t-1
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      uint4 h, m, l;
      wx_split8(rg[j][0], rg[j][1], h, m, l);
      lds_gy[gw_[j]] = h;
      lds_gy[GYB / 16 + gw_[j]] = m;
      lds_gy[2 * (GYB / 16) + gw_[j]] = l;
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      uint4 h, m, l;
      wx_split8(rx[j][0], rx[j][1], h, m, l);
      if (xpos[j] >= 0) {
        lds_x[xw_[j]] = h;
        if (!xe) {
          lds_x[XB / 16 + xw_[j]] = m;
          lds_x[2 * (XB / 16) + xw_[j]] = l;
        } else if ((m.x | m.y | m.z | m.w) != 0u) {
          atomicOr(g.status, DGV2_STATUS_X_INEXACT);             // (m = bf16(x - h) is zero exactly when x is bf16-representable)
        }
      }
    }
    if (t + 1 < t_end) issue();
    __syncthreads();
    wx3_tile<XE>(acc, lbase, voffA, voffB);
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  const int lr = lane & 15, lc = lane >> 4;
  float* pb = part + (int64_t)split * g.O * 9 * g.C;
  const int c = c0 + wave * 16 + lr;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int mw = 0; mw < 4; ++mw)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + mw * 16 + lc * 4 + r;
        pb[((int64_t)o * 9 + tap) * g.C + c] = acc[tap][mw][r];
      }
}

// ... and the input channels behind the last whole c-tile, exact fp32 (NC of them at a time, c0 <= c < clive).  The sums
// run over every output pixel of the batch while the result is 9 x NC numbers per output channel: a block takes a slice
// of the pixel tiles (its own split axis, TS slices) and 128 output channels, 8 pixel lanes per channel, and leaves
// tpart[ts][o][tap][k]; conv_wgrad_x3_tail_reduce_kernel folds the slices into part[0] (zeros in the other splits and in
// clive <= c < C), which the common reduce then treats like the matrix-core columns.
template <int NC>
__global__ __launch_bounds__(1024) void conv_wgrad_x3_tail_kernel(float* __restrict__ tpart, const float* __restrict__ gy,
                                                                  const float* __restrict__ x, WSGeom g, int c0, int tiles_per_ts) {
  __shared__ float red[8][128];
  const int ts = blockIdx.x, o0 = blockIdx.y * 128;
  const int ol = threadIdx.x & 127, pl = threadIdx.x >> 7;
  const int t_begin = ts * tiles_per_ts, t_end = min(t_begin + tiles_per_ts, g.ntiles);
  float acc[9][NC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[t][k] = 0.f;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tw = tile % g.tiles_w, th = (tile / g.tiles_w) % g.tiles_h, b = tile / (g.tiles_w * g.tiles_h);
    const float* gyb = gy + (int64_t)b * g.Ho * g.Wo * g.O + o0 + ol;
    const float* xb = x + (int64_t)b * g.H * g.W * g.C + c0;
#pragma unroll 4
    for (int p = pl; p < 128; p += 8) {
      const int ho = th * 4 + (p >> 5), wo = tw * 32 + (p & 31);
      if (ho >= g.Ho) continue;                 // (W % 32 == 0: no ragged columns)
      const float gv = gyb[(ho * g.Wo + wo) * g.O];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        int hi = ho + t / 3 - 1, wi = wo + t % 3 - 1;
        hi = hi < 0 ? 0 : (hi >= g.H ? g.H - 1 : hi);
        wi = wi < 0 ? wi + g.W : (wi >= g.W ? wi - g.W : wi);
        const float* xp = xb + (hi * g.W + wi) * g.C;     // wave-uniform
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[t][k] = fmaf(gv, xp[k], acc[t][k]);
      }
    }
  }
  float* out = tpart + ((int64_t)ts * g.O + o0 + ol) * 9 * NC;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      red[pl][ol] = acc[t][k];
      __syncthreads();
      if (pl == 0) {
        float v = red[0][ol];
#pragma unroll
        for (int l = 1; l < 8; ++l) v += red[l][ol];
        out[t * NC + k] = v;
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(256) void conv_wgrad_x3_tail_reduce_kernel(float* __restrict__ part, const float* __restrict__ tpart,
                                                                        int O, int C, int c0, int clive, int NC, int TS, int nsplit) {
  const int nz = C - c0;
  const int64_t n = (int64_t)nsplit * O * 9 * nz;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % nz);
    const int64_t r = i / nz;
    const int t = (int)(r % 9);
    const int o = (int)((r / 9) % O), split = (int)(r / 9 / O);
    float v = 0.f;
    if (split == 0 && k < NC && c0 + k < clive)
      for (int s = 0; s < TS; ++s) v += tpart[(((int64_t)s * O + o) * 9 + t) * NC + k];
    part[(((int64_t)split * O + o) * 9 + t) * C + c0 + k] = v;
  }
}

// gw[i] = sum_k part[k][i].  256 threads = 16 float4 columns x 16 split lanes: every lane sums its share
// of the splits with independent loads, LDS folds the 16 lanes.  n is a multiple of 4.
// scale: factor on the result; kkC > 0: write the PARAMETER's layout [O, C, kh*kw] instead of [O, kh*kw, C]
// (kkC = kh*kw*C, C = Cc), so that the gradient needs no permute / scale pass before the optimizer sees it.
// ldo > 0 (with kkC == 0): row r = i / Cc of an image goes to gw[r * ldo + i % Cc], images ldo * (n / Cc) apart -- the
// caller's gradient of a wider weight, written in place (dgv2_bmm_tn_stream_ld).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ gw, const float* __restrict__ part,
                                                           int64_t n, int nsplit, float scale = 1.f, int kkC = 0,
                                                           int Cc = 0, int64_t ldo = 0) {
  __shared__ float4 red[16][16];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = ((int64_t)blockIdx.x * 16 + col) * 4;
  part += (int64_t)blockIdx.y * nsplit * n;   // per-image mode: image blockIdx.y owns its own nsplit partials
  gw += (int64_t)blockIdx.y * (ldo > 0 ? (n / Cc) * ldo : n);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
#pragma unroll 4
    for (int k = sl; k < nsplit; k += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)k * n + i);
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
    s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
    if (kkC == 0 && ldo > 0) {
      const int64_t r = i / Cc;
      *reinterpret_cast<float4*>(gw + r * ldo + (i - r * Cc)) = s;
    } else if (kkC == 0) {
      *reinterpret_cast<float4*>(gw + i) = s;
    } else {
      const float v4[4] = {s.x, s.y, s.z, s.w};
      const int kk = kkC / Cc;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t lin = i + e;
        const int64_t o = lin / kkC;
        const int r = (int)(lin - o * kkC);
        const int t = r / Cc, c = r - t * Cc;
        gw[(o * Cc + c) * kk + t] = v4[e];
      }
    }
  }
}

struct WSPlan {
  WSGeom g;
  int nsplit, otiles, mfn, nfn;
  int go;   // o-groups per block (bf16 64 x 64 tiles at stride 2: 2 = eight waves sharing the halo tile)
};

bool ws_plan(WSPlan& p, int B, int H, int W, int C, int O, int k, int stride, int pad, int ring, int dtype,
             bool per_image = false) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || O <= 0) return false;
  if ((k != 1 && k != 3) || pad != (k - 1) / 2 || (stride != 1 && stride != 2)) return false;
  const int ce = dtype == DGV2_BF16 ? 8 : 4;
  if (C % ce || O % ce) return false;
  WSGeom& g = p.g;
  g = WSGeom{B, H, W, C, O, (H + 2 * pad - k) / stride + 1, (W + 2 * pad - k) / stride + 1, k, stride, pad, ring};
  if (g.Ho <= 0 || g.Wo <= 0) return false;
  // fp32: the 64 x 64 tile (144 accumulator + 84 staging registers per lane) for the wide 3x3 stride-1 layers -- the
  // fp32 island of D's epilogue, 528 -> 512 at 4 x 32 -- and the small tile for everything else (parity mode)
  const bool wide32 = dtype == DGV2_F32 && k == 3 && stride == 1 && C >= 128 && O >= 128;
  const bool big = dtype == DGV2_BF16 || wide32;
  p.mfn = (big && O > 32) ? 4 : 2;
  p.nfn = (big && C > 32) ? 4 : 2;
  const int wr = stride == 1 ? 4 : 2;
  g.tiles_h = (g.Ho + wr - 1) / wr;
  g.tiles_w = (g.Wo + 31) / 32;
  g.ntiles = B * g.tiles_h * g.tiles_w;
  g.ctiles = (C + 16 * p.nfn - 1) / (16 * p.nfn);
  static const bool no_go2 = getenv("DGV2_WS_NO_GO2") != nullptr;   // A/B switch for benchmarking
  static const bool go2_s1 = getenv("DGV2_WS_GO2_S1") != nullptr;   // experiment: the stride-1 layers as well
  p.go = (!no_go2 && dtype == DGV2_BF16 && p.mfn == 4 && p.nfn == 4 && k == 3 && (stride == 2 || go2_s1) && O % 128 == 0 &&
          !per_image) ? 2 : 1;
  p.otiles = (O + 16 * p.mfn * p.go - 1) / (16 * p.mfn * p.go);
  const int pairs = g.ctiles * p.otiles;
#ifdef DGV2_ABLATE
  static const int abl = getenv("DGV2_WS_ABLATE") ? atoi(getenv("DGV2_WS_ABLATE")) : 0;
  g.ablate = abl;
#endif
  g.x_shared = 0;
  // 64 x 64 bf16 tiles: two blocks per CU (the in-place MFMAs keep the kernel inside 256 registers) -- 512 blocks;
  // measured against 256: 8x64 256->256 115 -> 85 us, 16x128 128->128 117 -> 90 us, 4x32 544->512 156 -> 103 us
  static const int blocks_big = getenv("DGV2_WS_BLOCKS_BIG") ? atoi(getenv("DGV2_WS_BLOCKS_BIG")) : 512;
  static const int blocks_small = getenv("DGV2_WS_BLOCKS_SMALL") ? atoi(getenv("DGV2_WS_BLOCKS_SMALL")) : 512;
  // (eight-wave blocks: one per CU)
  int nsplit = (p.go == 2 ? 256 : wide32 ? 512 : (p.mfn == 4 && p.nfn == 4) ? blocks_big : blocks_small) / pairs;
  nsplit = nsplit < 1 ? 1 : (nsplit > g.ntiles ? g.ntiles : nsplit);
  g.tiles_per_split = (g.ntiles + nsplit - 1) / nsplit;
  p.nsplit = (g.ntiles + g.tiles_per_split - 1) / g.tiles_per_split;
  if (per_image) {
    // every split must lie inside one image: splits per image = the largest divisor of the image's tile count
    // that keeps the launch near its block target
    const int tpi = g.tiles_h * g.tiles_w;
    int spi = nsplit / B;
    spi = spi < 1 ? 1 : (spi > tpi ? tpi : spi);
    while (tpi % spi) --spi;
    g.tiles_per_split = tpi / spi;
    p.nsplit = B * spi;
  }
  return true;
}

template <typename T, int S, int MFN, int NFN, bool K3>
int ws_launch(float* part, const void* gy, const void* x, const WSPlan& p, hipStream_t st) {
  dim3 grid(p.nsplit, p.g.ctiles * p.otiles);
  if constexpr (sizeof(T) == 2) {
    if constexpr (MFN == 4 && NFN == 4 && K3) {
      if (p.go == 2) {
        using Cf2 = WZCfg<S, MFN, NFN, K3, 2>;
        auto kern2 = conv_wgrad_stream_bf16_kernel<S, MFN, NFN, K3, 2>;
        if (Cf2::LDS > 64 * 1024) {
          hipError_t e = hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cf2::LDS);
          if (e != hipSuccess) return (int)e;
        }
        kern2<<<grid, 512, Cf2::LDS, st>>>(part, (const bf16_t*)gy, (const bf16_t*)x, p.g);
        return 0;
      }
    }
    using Cf = WZCfg<S, MFN, NFN, K3>;
    auto kern = conv_wgrad_stream_bf16_kernel<S, MFN, NFN, K3, 1>;
    if (Cf::LDS > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cf::LDS);
      if (e != hipSuccess) return (int)e;
    }
    kern<<<grid, 256, Cf::LDS, st>>>(part, (const bf16_t*)gy, (const bf16_t*)x, p.g);
  } else {
    using Cf = WSCfg<T, S, MFN, NFN, K3>;
    auto kern = conv_wgrad_stream_kernel<T, S, MFN, NFN, K3>;
    if (Cf::LDS > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cf::LDS);
      if (e != hipSuccess) return (int)e;
    }
    kern<<<grid, 256, Cf::LDS, st>>>(part, (const T*)gy, (const T*)x, p.g);
  }
  return 0;
}

template <typename T, int S, bool K3>
int ws_dispatch(float* part, const void* gy, const void* x, const WSPlan& p, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (p.mfn == 4 && p.nfn == 4) return ws_launch<T, S, 4, 4, K3>(part, gy, x, p, st);
    if (p.mfn == 4) return ws_launch<T, S, 4, 2, K3>(part, gy, x, p, st);
    if (p.nfn == 4) return ws_launch<T, S, 2, 4, K3>(part, gy, x, p, st);
  } else if constexpr (S == 1 && K3) {
    if (p.mfn == 4 && p.nfn == 4) return ws_launch<T, S, 4, 4, K3>(part, gy, x, p, st);
  }
  return ws_launch<T, S, 2, 2, K3>(part, gy, x, p, st);
}

template <typename T>
int ws_dispatch_geom(float* part, const void* gy, const void* x, const WSPlan& p, hipStream_t st) {
  if (p.g.k == 3)
    return p.g.stride == 1 ? ws_dispatch<T, 1, true>(part, gy, x, p, st) : ws_dispatch<T, 2, true>(part, gy, x, p, st);
  return p.g.stride == 1 ? ws_dispatch<T, 1, false>(part, gy, x, p, st) : ws_dispatch<T, 2, false>(part, gy, x, p, st);
}

}  // namespace

// Number of fp32 elements of scratch dgv2_conv_wgrad_stream needs for this geometry (0 and DGV2_EINVAL when
// the geometry is not supported).
extern "C" int dgv2_conv_wgrad_stream_scratch(int64_t* elems, int B, int H, int W, int C, int O, int k, int stride,
                                              int pad, int dtype) {
  if (!elems) return DGV2_EINVAL;
  *elems = 0;
  WSPlan p;
  if (!ws_plan(p, B, H, W, C, O, k, stride, pad, 1, dtype)) return DGV2_EINVAL;
  *elems = (int64_t)p.nsplit * O * k * k * C;
  return 0;
}

// Per-sample 1x1 weight gradient (the modulated conv of the generator, ModConv2d autograd, style.py:105-118):
//   gw[b, o, c] = sum_p gy[b, p, o] * x[b, p, c],  p over the H x W pixels of sample b.
// Same streaming kernel; every split stays inside one image and the reduce runs per image.
extern "C" int dgv2_bmm_tn_stream_scratch(int64_t* elems, int B, int H, int W, int C, int O, int dtype) {
  if (!elems) return DGV2_EINVAL;
  *elems = 0;
  WSPlan p;
  if (!ws_plan(p, B, H, W, C, O, 1, 1, 0, 0, dtype, true)) return DGV2_EINVAL;
  *elems = (int64_t)p.nsplit * O * C;
  return 0;
}

extern "C" int dgv2_bmm_tn_stream(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x,
                                  int B, int H, int W, int C, int O, int dtype, void* stream) {
  return dgv2_bmm_tn_stream_x(gw, scratch, scratch_elems, gy, x, 0, B, H, W, C, O, dtype, stream);
}

// x_shared != 0: x is one image [H, W, C] contracted against every sample's gy (the batch-shared positional encoding:
// gw[b, o, c] = sum_p gy[b, p, o] * pe[p, c]).
extern "C" int dgv2_bmm_tn_stream_x(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x,
                                    int x_shared, int B, int H, int W, int C, int O, int dtype, void* stream) {
  return dgv2_bmm_tn_stream_ld(gw, 0, scratch, scratch_elems, gy, x, x_shared, B, H, W, C, O, dtype, stream);
}

// ldo > 0: gw is [B, O, ldo] with ldo >= C (and a multiple of 4): the columns [0, C) of the caller's wider per-sample weight
// gradient, written in place (offset the pointer for another first column); 0: contiguous [B, O, C].
extern "C" int dgv2_bmm_tn_stream_ld(float* gw, int64_t ldo, float* scratch, int64_t scratch_elems, const void* gy,
                                     const void* x, int x_shared, int B, int H, int W, int C, int O, int dtype, void* stream) {
  if (!gw || !scratch || !gy || !x || !aligned16(gy) || !aligned16(x) || !aligned16(scratch) || !aligned16(gw))
    return DGV2_EINVAL;
  if (ldo < 0 || (ldo > 0 && (ldo < C || (ldo & 3)))) return DGV2_EINVAL;
  if (dtype != DGV2_BF16 && dtype != DGV2_F32) return DGV2_EINVAL;
  WSPlan p;
  if (!ws_plan(p, B, H, W, C, O, 1, 1, 0, 0, dtype, true)) return DGV2_EINVAL;
  p.g.x_shared = x_shared != 0;
  const int64_t n = (int64_t)O * C;
  if (scratch_elems < (int64_t)p.nsplit * n || (n & 3)) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int rc = dtype == DGV2_BF16 ? ws_dispatch_geom<bf16_t>(scratch, gy, x, p, st)
                                    : ws_dispatch_geom<float>(scratch, gy, x, p, st);
  if (rc) return rc;
  dim3 grid((unsigned)((n / 4 + 15) / 16), B);
  wgrad_reduce_kernel<<<grid, 256, 0, st>>>(gw, scratch, n, p.nsplit / B, 1.f, 0, C, ldo);
  DGV2_RETURN_LAST();
}

// gw fp32 [O, k*k, C] (overwritten).  k in {1,3}, pad = (k-1)/2, stride in {1,2}, C and O multiples of the
// 16-byte vector (8 bf16 / 4 fp32).  scratch: fp32 [scratch_elems] >= dgv2_conv_wgrad_stream_scratch(...).
extern "C" int dgv2_conv_wgrad_stream(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x,
                                      int B, int H, int W, int C, int O, int k, int stride, int pad, int ring,
                                      int dtype, void* stream) {
  return dgv2_conv_wgrad_stream_pl(gw, scratch, scratch_elems, gy, x, B, H, W, C, O, k, stride, pad, ring, 1.f, 0, dtype,
                                   stream);
}

extern "C" int dgv2_conv_wgrad_stream_pl(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x,
                                         int B, int H, int W, int C, int O, int k, int stride, int pad, int ring,
                                         float scale, int param_layout, int dtype, void* stream) {
  if (!gw || !scratch || !gy || !x || !aligned16(gy) || !aligned16(x) || !aligned16(scratch) || !aligned16(gw))
    return DGV2_EINVAL;
  if (dtype != DGV2_BF16 && dtype != DGV2_F32) return DGV2_EINVAL;
  WSPlan p;
  if (!ws_plan(p, B, H, W, C, O, k, stride, pad, ring, dtype)) return DGV2_EINVAL;
  const int64_t n = (int64_t)O * k * k * C;
  if (scratch_elems < (int64_t)p.nsplit * n) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  // ring == 2 tells the kernel that a single conditional wrap covers every halo column
  if (ring && (p.g.tiles_w * 32 - 1) * stride + k - 1 - pad < 2 * W && pad <= W) p.g.ring = 2;
  rc = dtype == DGV2_BF16 ? ws_dispatch_geom<bf16_t>(scratch, gy, x, p, st) : ws_dispatch_geom<float>(scratch, gy, x, p, st);
  if (rc) return rc;
  wgrad_reduce_kernel<<<(int)((n / 4 + 15) / 16), 256, 0, st>>>(gw, scratch, n, p.nsplit, scale,
                                                                 param_layout ? k * k * C : 0, C);
  DGV2_RETURN_LAST();
}

// The same weight gradient for the fp32 3x3 stride-1 ring conv on the bf16 matrix cores (conv_wgrad_x3_kernel: both operands
// as three bf16 planes, six products per multiply; fp32-equivalent):  gw fp32 [O, 9, C] (or the parameter's layout) from
// gy [B, H, W, O] and x [B, H, W, C] fp32.  Input channels [0, 64 * floor(C / 64)) on the matrix cores, [.., clive) in
// exact fp32 (at most 16), [clive, C) = 0 (padding channels of x).  scratch: dgv2_conv3x3_x3_wgrad_scratch.
// DGV2_ENOTSUP where the kernel does not cover the geometry (O % 128, C < 64, C % 8, clive - 64 * floor(C / 64) > 16,
// W % 32): callers then run dgv2_conv_wgrad_stream_pl.
extern "C" int dgv2_conv3x3_x3_wgrad(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x, int B,
                                     int H, int W, int C, int clive, int x_exact, int O, float scale, int param_layout,
                                     int* status, void* stream) {
  if (!gw || !scratch || !gy || !x || !aligned16(gy) || !aligned16(x) || !aligned16(scratch) || !aligned16(gw))
    return DGV2_EINVAL;
  if (B < 1 || H < 1 || W < 1 || C < 1 || O < 1 || clive < 1 || clive > C || x_exact < 0 || x_exact > C) return DGV2_EINVAL;
  if (x_exact > 0 && !status) return DGV2_EINVAL;
  static const bool off = getenv("DGV2_NO_CONV_X3") != nullptr || getenv("DGV2_NO_WGRAD_X3") != nullptr;
  const int ctiles = C / 64, ntail = clive - ctiles * 64;
  if (off || O % 64 || ctiles < 1 || C % 8 || ntail > 16 || W % 32) return DGV2_ENOTSUP;
  if ((int64_t)B * H * W * (C > O ? C : O) >= (1ll << 31)) return DGV2_ENOTSUP;
  WSPlan p;
  WSGeom& g = p.g;
  g = WSGeom{B, H, W, C, O, H, W, 3, 1, 1, 2};
  g.tiles_h = (H + 3) / 4;
  g.tiles_w = W / 32;
  g.ntiles = B * g.tiles_h * g.tiles_w;
  g.ctiles = ctiles;
  g.x_shared = 0;
  g.x_exact = x_exact;
  g.status = status;
#ifdef DGV2_ABLATE
  g.ablate = 0;
#endif
  const int pairs = ctiles * (O / 64);
  int nsplit = 256 / pairs;                 // one block per CU (127 KB of LDS each)
  nsplit = nsplit < 1 ? 1 : (nsplit > g.ntiles ? g.ntiles : nsplit);
  g.tiles_per_split = (g.ntiles + nsplit - 1) / nsplit;
  nsplit = (g.ntiles + g.tiles_per_split - 1) / g.tiles_per_split;
  const int64_t n = (int64_t)O * 9 * C;
  const bool tail = C > ctiles * 64;
  const int NC = ntail <= 1 ? 1 : (ntail <= 4 ? 4 : 16);
  const int TS = g.ntiles < 64 ? g.ntiles : 64, tiles_per_ts = (g.ntiles + TS - 1) / TS;
  const int64_t tneed = tail ? (int64_t)TS * O * 9 * NC : 0;
  if (scratch_elems < (int64_t)nsplit * n + tneed || O % 128) return DGV2_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  const bool xe = x_exact >= ctiles * 64;
  auto kern = xe ? conv_wgrad_x3_kernel<true> : conv_wgrad_x3_kernel<false>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[xe]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WXCfg::LDS);
    if (e != hipSuccess) return (int)e;
    attr_set[xe] = true;
  }
  kern<<<dim3(nsplit, pairs), 256, WXCfg::LDS, st>>>(scratch, (const float*)gy, (const float*)x, g);
  if (tail) {
    float* tpart = scratch + (int64_t)nsplit * n;
    const dim3 tg(TS, O / 128);
    const int c0 = ctiles * 64;
    if (NC == 1) conv_wgrad_x3_tail_kernel<1><<<tg, 1024, 0, st>>>(tpart, (const float*)gy, (const float*)x, g, c0, tiles_per_ts);
    else if (NC == 4) conv_wgrad_x3_tail_kernel<4><<<tg, 1024, 0, st>>>(tpart, (const float*)gy, (const float*)x, g, c0, tiles_per_ts);
    else conv_wgrad_x3_tail_kernel<16><<<tg, 1024, 0, st>>>(tpart, (const float*)gy, (const float*)x, g, c0, tiles_per_ts);
    conv_wgrad_x3_tail_reduce_kernel<<<grid_for((int64_t)nsplit * O * 9 * (C - c0), 256), 256, 0, st>>>(scratch, tpart, O, C, c0,
                                                                                                      clive, NC, TS, nsplit);
  }
  wgrad_reduce_kernel<<<(int)((n / 4 + 15) / 16), 256, 0, st>>>(gw, scratch, n, nsplit, scale, param_layout ? 9 * C : 0, C);
  DGV2_RETURN_LAST();
}

// fp32 elements of scratch dgv2_conv3x3_x3_wgrad needs (split-K partials + the tail kernel's slices); 0 and DGV2_ENOTSUP
// where the geometry is not covered.
extern "C" int dgv2_conv3x3_x3_wgrad_scratch(int64_t* elems, int B, int H, int W, int C, int clive, int O) {
  if (!elems) return DGV2_EINVAL;
  *elems = 0;
  if (B < 1 || H < 1 || W < 1 || C < 64 || O < 1 || clive < 1 || clive > C) return DGV2_EINVAL;
  const int ctiles = C / 64, ntail = clive - ctiles * 64;
  if (O % 128 || C % 8 || ntail > 16 || W % 32) return DGV2_ENOTSUP;
  const int ntiles = B * ((H + 3) / 4) * (W / 32);
  int nsplit = 256 / (ctiles * (O / 64));
  nsplit = nsplit < 1 ? 1 : (nsplit > ntiles ? ntiles : nsplit);
  const int tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  const int NC = ntail <= 1 ? 1 : (ntail <= 4 ? 4 : 16);
  const int TS = ntiles < 64 ? ntiles : 64;
  *elems = (int64_t)nsplit * O * 9 * C + (C > ctiles * 64 ? (int64_t)TS * O * 9 * NC : 0);
  return 0;
}
