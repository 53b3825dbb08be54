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
  auto issue = [&]() {           // global loads of that tile into registers, then advance
    const int h0 = th_n * WR, w0 = tw_n * 32;
    const bf16_t* gyb = gy + (int64_t)b_n * g.Ho * g.Wo * g.O + o0;
    const bf16_t* xb = x + (g.x_shared ? 0 : (int64_t)b_n * g.H * g.W * g.C) + c0;
    if (++tw_n == g.tiles_w) {
      tw_n = 0;
      if (++th_n == g.tiles_h) { th_n = 0; ++b_n; }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int id = tid + j * NTHR;
      const int ho = h0 + (gpix[j] >> 8), wo = w0 + (gpix[j] & 255);
      rg[j] = make_uint4(0, 0, 0, 0);
      if (gpix[j] >= 0 && ho < g.Ho && wo < g.Wo)
        rg[j] = *reinterpret_cast<const uint4*>(gyb + (ho * g.Wo + wo) * g.O + (id % GVT) * 8);
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int id = tid + j * NTHR;
      int hi = h0 * S - off + (xpos[j] >> 16), wi = w0 * S - off + (xpos[j] & 0xffff);
      hi = hi < 0 ? 0 : (hi >= g.H ? g.H - 1 : hi);
      if (g.ring == 2) wi = wi < 0 ? wi + g.W : (wi >= g.W ? wi - g.W : wi);   // host-checked: one wrap suffices
      else wi = g.ring ? floormod(wi, g.W) : (wi < 0 ? 0 : (wi >= g.W ? g.W - 1 : wi));
      rx[j] = make_uint4(0, 0, 0, 0);
      if (xpos[j] >= 0) rx[j] = *reinterpret_cast<const uint4*>(xb + (hi * g.W + wi) * g.C + (id % XV) * 8);
    }
  };

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
    if (t + 1 < t_end) issue();
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
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[tap][mw]) : "v"(ua.v), "v"(ub.v));
          } else {   // the smaller tiles have registers to spare and schedule better with the builtin
            Mfma16<bf16_t>::run(acc[tap][mw], a[mw], bb);
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

// gw[i] = sum_k part[k][i].  256 threads = 16 float4 columns x 16 split lanes: every lane sums its share
// of the splits with independent loads, LDS folds the 16 lanes.  n is a multiple of 4.
// scale: factor on the result; kkC > 0: write the PARAMETER's layout [O, C, kh*kw] instead of [O, kh*kw, C]
// (kkC = kh*kw*C, C = Cc), so that the gradient needs no permute / scale pass before the optimizer sees it.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ gw, const float* __restrict__ part,
                                                           int64_t n, int nsplit, float scale = 1.f, int kkC = 0,
                                                           int Cc = 0) {
  __shared__ float4 red[16][16];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = ((int64_t)blockIdx.x * 16 + col) * 4;
  part += (int64_t)blockIdx.y * nsplit * n;   // per-image mode: image blockIdx.y owns its own nsplit partials
  gw += (int64_t)blockIdx.y * n;
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
    if (kkC == 0) {
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
  if (!gw || !scratch || !gy || !x || !aligned16(gy) || !aligned16(x) || !aligned16(scratch) || !aligned16(gw))
    return DGV2_EINVAL;
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
  wgrad_reduce_kernel<<<grid, 256, 0, st>>>(gw, scratch, n, p.nsplit / B);
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
