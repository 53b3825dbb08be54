// Eight-wave form of the unrolled 3x3 direct convolution (forward, ring padding, channels-last, bf16 / e4m3):
// the layers of the discriminator whose conv_pipe_kernel blocks are bound by the bytes they stage per FLOP.
//   reference: ops.Conv2d forward (gans/models/ops/common.py:187-210) at ResidualBlock.conv1 (3x3 stride 1) and
//   ResidualBlock.conv2 (3x3 stride 2 behind the blur), gans/models/dusty_v2.py:325-345.
//
// conv_pipe_kernel's block is four waves on ONE (pixel tile, 64-channel slab) pair: per 64-byte K-chunk it stages the
// halo tile AND the weight slab [9 taps][64 o] for 9.4 MFLOP -- at stride 2 a 9 x 65 halo for 4 x 32 outputs, 63 FLOP
// per staged byte, and the measured bound (ablation builds, profiles/round4_s2_ablation_start.txt: without its loads
// the stride-2 forward kernels take 44-54 of their 82-103 us, without the MFMAs the same time as with them).  Here a block is
// EIGHT waves = two groups of four that share what the layer re-stages most:
//   GM = 2 (stride 2): two 64-channel slabs on one 8 x 32 pixel tile -- the 17 x 65 halo tile is staged once for 128
//          output channels: 131 FLOP per staged byte;
//   GN = 2 (stride 1): two neighbouring 8 x 32 pixel tiles on one slab -- the weight slab is staged once for 512 pixels.
// Otherwise the structure is conv_pipe_kernel's unrolled variant (F33): operands in four 16-byte LDS planes
// (conflict-free ds_read_b128 at any row, taps = immediate offsets), a software-pipelined (tile, chunk) walk with
// the next stage's global loads in registers during the MFMA loop, in-place asm MFMAs, bias / leaky ReLU / residual /
// bf16 pairing in the epilogue.  Stride 2 adds one thing: the halo image is split by column parity (even and odd
// columns of a row are two runs PITCH rows apart), so a fragment's 16 pixels (input columns 2j + kx) are CONSECUTIVE
// rows of a plane again -- the pixel-major image made every stride-2 fragment read a 2-way bank conflict.
// Forward convs only (rows clamp); data gradients stay on conv_pipe_kernel.
#include <string.h>

#include "gemm_core.h"

namespace {

struct C8 {
  int B, Hin, Win, Cin, Hg, Wg, O, ldy;
  int wtaps, widx0, wstep;   // weight slot of tap t: widx0 + t * wstep (rows of w: [O][wtaps][Cin])
  int wimg;                  // w is the staging image [O / 64][Cin / chunk][2304 16-byte units in slot order] that
                             // dgv2_conv_weight_bank_ex writes: a slab's chunk is ONE contiguous 36 KB run (whole cache
                             // lines, 1 KB per wave-load) instead of 576 pieces of 64 bytes from 576 different lines
  int tpb;                   // super-tiles (GN pixel tiles each) a block walks along W
  int nt, xcd, gx, gy, gz;
#ifdef DGV2_ABLATE   // benchmarking builds only (make ABLATE=1): wrong results by design, never in the shipped library
  int ablate;        // DGV2_C8_ABLATE: 1 no stores, 2 no MFMA loop, 4 no input loads, 8 no weight loads, 16 no epilogue, 32 no LDS writes
#define C8_ABL (p.ablate)
#else
#define C8_ABL 0
#endif
  const float* bias;
  const float* acc_scale;
  const void* resid;
  int act;
  float alpha, scale;
};

template <typename T, typename TY, int S, int GM, int GN, int RW>
struct C8Cfg {
  static constexpr int NT = 256 * GM * GN, UI = 256 * GM, UW = 256 * GN;
  static constexpr int CE = 16 / sizeof(T);
  static constexpr int TH = 4 * RW, MF = 4, NF = 2 * RW;
  static constexpr int IROWS = (TH - 1) * S + 3, ICOLS = 31 * S + 3, NPIX = IROWS * ICOLS;
  // stride 2: row (iy * 2 + (ix & 1)) * PITCH + (ix >> 1); 36 = 4 mod 16 keeps the even and the odd run of 8 consecutive
  // pixels (one ds_write_b128 lane group) on different banks
  static constexpr int PITCH = S == 2 ? 36 : ICOLS;
  static constexpr int LIVE = S == 2 ? IROWS * 2 * PITCH : NPIX;
  static constexpr int NI = (NPIX * 4 + UI - 1) / UI;
  static constexpr int COVER = NI * UI / 4;                                    // pixel indices the staging slots cover
  static constexpr int PIN = (LIVE + (COVER > NPIX ? COVER - NPIX : 0) + 15) / 16 * 16;   // dead slots land behind the image
  static constexpr int PW = 9 * 64;
  static constexpr int TS = UW / 256;                                          // taps one staging slot step advances
  static constexpr int NW = (9 + TS - 1) / TS;
  static constexpr size_t LDS = sizeof(uint4) * 4 * ((size_t)GN * PIN + (size_t)GM * PW);
};

// HZ: the stride-1 DATA GRADIENT of the same conv (x = gy, w = the transposed weights in the launch's tap order): rows
// outside the image are ZERO, not clamped -- they are staged as whatever the clamped address holds, because every tap that
// would read them is a dead tap of that output row and is skipped (wave-uniform branch around in-place MFMAs) -- and
// the replicate-padding term of a border row (output row 0 sees its own gy row once more through the weights of the
// tap mirrored in dy, likewise the last row) is issued next to that mirrored tap's MFMAs: weights already in
// registers, one more pixel-fragment read (conv_pipe_kernel's F33 = 2 form).
// WIMG: w is the staging image (C8::wimg); a template parameter because the two addressing forms together cost the
// 256-register instances five spilled dwords
#ifndef DGV2_C8_SPREAD
#define DGV2_C8_SPREAD 1
#endif
template <typename T, typename TY, int S, int GM, int GN, int RW, int HZ, bool WIMG>
__global__ __launch_bounds__(256 * GM * GN, 2) void conv8_kernel(TY* __restrict__ y, const T* __restrict__ x,
                                                                 const T* __restrict__ w, C8 p) {
  using Cf = C8Cfg<T, TY, S, GM, GN, RW>;
  static_assert(HZ == 0 || S == 1, "data gradient: stride 1");
  constexpr int UI = Cf::UI, UW = Cf::UW, CE = Cf::CE, TH = Cf::TH, MF = Cf::MF, NF = Cf::NF;
  constexpr int ICOLS = Cf::ICOLS, NPIX = Cf::NPIX, PITCH = Cf::PITCH, LIVE = Cf::LIVE;
  constexpr int NI = Cf::NI, PIN = Cf::PIN, PW = Cf::PW, TS = Cf::TS, NW = Cf::NW;
  static_assert(GM * GN == 2, "two groups of four waves");
  constexpr int SPREAD = DGV2_C8_SPREAD;
  static_assert(MF % NF == 0, "A fragments are re-read in equal shares behind the pixel-fragment groups");
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  __shared__ __attribute__((aligned(16))) float s_bias[GM * 64];
  const float ws = p.acc_scale ? *p.acc_scale : 1.f;

  const int tid = threadIdx.x;
  const int t256 = tid & 255, grp = tid >> 8;
  const int gm = GM == 2 ? grp : 0, gn = GN == 2 ? grp : 0;
  const int wave4 = (tid >> 6) & 3, lane = tid & 63;
  const int lr = lane & 15, lc = lane >> 4;
  uint4* const lds_in = smem + gn * 4 * PIN;
  uint4* const lds_w = smem + GN * 4 * PIN + gm * 4 * PW;

  const int tiles_h = (p.Hg + TH - 1) / TH;
  const int tiles_w = (p.Wg + 31) / 32;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd) {   // blocks that share one pixel tile's input get ids 8 apart: same XCD, one fetch into that L2 (conv_direct.hip)
    const int n = blockIdx.x, q = n >> 3;
    const int pt = (q / p.gz) * 8 + (n & 7);
    if (pt >= p.gx * p.gy) return;
    bz = q % p.gz;
    bx = pt % p.gx;
    by = pt / p.gx;
  }
  const int b = by / tiles_h;
  const int h0 = (by % tiles_h) * TH;
  const int o0 = (bz * GM + gm) * 64;
  const int st0 = bx * p.tpb;                              // first super-tile of this block
  const int nsup = (tiles_w + GN - 1) / GN;
  const int ntile = min(p.tpb, nsup - st0);
  const T* xb = x + (int64_t)b * p.Hin * p.Win * p.Cin;
  constexpr int kchunk = 4 * CE;
  const int nchunks = p.Cin / kchunk;

  // ---- staging slots (hoisted: everything but the W wrap of a tile and the chunk offset) ----
  const int ui = gm * 256 + t256;                          // this thread among the UI stagers of input tile gn
  const int uw = gn * 256 + t256;                          // ... among the UW stagers of weight slab gm
  const int in_plane = (ui >> 3) & 3, w_plane = (uw >> 3) & 3;
  const int in_pix0 = ((ui >> 5) << 3) | (ui & 7);         // slot j: + (UI / 4) * j
  const int w_row0 = ((uw >> 5) << 3) | (uw & 7);          // row of slot 0 in the slab image [tap][o]; slot j: + (UW / 4) * j
  int grow[NI], icol[NI], lrow[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int pix = in_pix0 + (UI / 4) * j;
    const int pl = pix < NPIX ? pix : NPIX - 1;            // dead slots re-load the last pixel ...
    const int iy = pl / ICOLS, ix = pl - iy * ICOLS;
    int gh = h0 * S - 1 + iy;
    gh = gh < 0 ? 0 : (gh >= p.Hin ? p.Hin - 1 : gh);     // replicate rows
    grow[j] = gh * p.Win * p.Cin + in_plane * CE;
    icol[j] = ix;
    const int live_row = S == 2 ? (iy * 2 + (ix & 1)) * PITCH + (ix >> 1) : pix;
    lrow[j] = pix < NPIX ? live_row : LIVE + (pix - NPIX);  // ... into rows behind the image
  }
  int goff[NI];
  auto tile_offsets = [&](int sup) {
    const int gw_base = ((sup * GN + gn) * 32) * S - 1;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int gw = gw_base + icol[j];                           // -1 <= gw < Win + 32 * S * GN + ICOLS (host-checked < 3 * Win)
      gw = gw < 0 ? gw + p.Win : gw;
      gw = gw >= 2 * p.Win ? gw - 2 * p.Win : gw;
      gw = gw >= p.Win ? gw - p.Win : gw;
      goff[j] = grow[j] + gw * p.Cin;
    }
  };
  // weight slot j = tap t0 + TS * j of output row o0 + (w_row0 & 63): a uniform tap pointer + this lane offset
  const int t0 = w_row0 >> 6;
  [[maybe_unused]] const unsigned wlane = (unsigned)(((o0 + (w_row0 & 63)) * p.wtaps + t0 * p.wstep) * p.Cin + w_plane * CE);

  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 rin[NI], rwt[NW];
  // loads [j0, j1) of a stage's input slots / weight slots (the whole stage in the prologue; one ninth per tap inside the
  // MFMA loop: a wave that issues all of a stage's loads in one burst sits in front of the full memory queue until most
  // of them have returned -- its MFMA loop then starts when the loads are nearly over, and the two phases add up)
  auto issue_in = [&](int c0, int j0, int j1) {
    if (C8_ABL & 4) return;
#pragma unroll
    for (int j = 0; j < NI; ++j)
      if (j >= j0 && j < j1) rin[j] = *reinterpret_cast<const u32x4*>(xb + c0 + (unsigned)goff[j]);
  };
  auto issue_w = [&](int c0, int j0, int j1) {
    if (C8_ABL & 8) return;
    if constexpr (WIMG) {
      const u32x4* img = reinterpret_cast<const u32x4*>(w) + ((size_t)(o0 >> 6) * nchunks + c0 / kchunk) * (PW * 4) + uw;
#pragma unroll
      for (int j = 0; j < NW; ++j)
        if (j >= j0 && j < j1 && (TS * (NW - 1) + (TS - 1) < 9 || t0 + TS * j < 9)) rwt[j] = img[UW * j];
    } else {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int t = t0 + TS * j;
        const T* wu = w + (p.widx0 + TS * j * p.wstep) * p.Cin + c0;
        if (j >= j0 && j < j1 && (TS * (NW - 1) + (TS - 1) < 9 || t < 9)) rwt[j] = *reinterpret_cast<const u32x4*>(wu + wlane);
      }
    }
  };
  uint4* const st_in = lds_in + in_plane * PIN;
  uint4* const st_w = lds_w + w_plane * PW + w_row0;

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+v"(acc[mf][nf]));   // written here, not between the first stage's MFMAs (see the epilogue)
    }

  int bpix[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int r = wave4 * RW + (nf >> 1);
    bpix[nf] = (S == 2 ? r * 4 * PITCH : r * ICOLS) + (nf & 1) * 16 + lr;
  }
  if (t256 < 64) s_bias[gm * 64 + t256] = p.bias ? p.bias[o0 + t256] : 0.f;   // (GN = 2: both groups write the same values)
  const uint4* const a_base = lds_w + lc * PW + lr;
  const uint4* const b_base = lds_in + lc * PIN;

  // HZ: taps that read only the zero rows above / below the image for this wave's output rows (tap t = (dy + 1) * 3 + dx + 1)
  unsigned dead[RW];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    dead[rr] = 0u;
    if constexpr (HZ != 0) {
      const int orow = h0 + wave4 * RW + rr;
      unsigned m = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if ((unsigned)(orow + t / 3 - 1) >= (unsigned)p.Hin) m |= 1u << t;
      dead[rr] = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    }
  }

  tile_offsets(st0);
  issue_in(0, 0, NI);
  issue_w(0, 0, NW);
  int tile = 0, cc = 0;
  const int nstage = ntile * nchunks;
  for (int s = 0; s < nstage; ++s) {
    __syncthreads();                // every wave has finished reading stage s-1
    if (!(C8_ABL & 32)) {
#pragma unroll
      for (int j = 0; j < NI; ++j) *reinterpret_cast<u32x4*>(st_in + lrow[j]) = rin[j];
    }
    if ((s == 0 || nchunks > 1) && !(C8_ABL & 32)) {
#pragma unroll
      for (int j = 0; j < NW; ++j)
        if (TS * (NW - 1) + (TS - 1) < 9 || t0 + TS * j < 9) *reinterpret_cast<u32x4*>(st_w + (UW / 4) * j) = rwt[j];
    }
    int ntile_i = tile, ncc = cc + 1;
    if (ncc == nchunks) { ncc = 0; ++ntile_i; }
    const bool more = s + 1 < nstage;
    if (more && ncc == 0) tile_offsets(st0 + ntile_i);
    if (more && (SPREAD == 0 || (C8_ABL & 2))) {
      issue_in(ncc * kchunk, 0, NI);
      if (nchunks > 1) issue_w(ncc * kchunk, 0, NW);
    }
    __syncthreads();                // stage s visible in LDS

    if (!(C8_ABL & 2)) {
      // the nine taps straight-line (conv_pipe_kernel's F33 form): immediate LDS offsets, a pixel fragment is re-read for
      // tap t+1 as soon as its MFMAs of tap t are issued, the weight fragments alternate between two sets
      constexpr int APG = MF / NF;
      uint4 a[2][MF], bb[NF];
      unsigned dd[RW];
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        dd[rr] = 0u;
        if constexpr (HZ != 0) {
          dd[rr] = dead[rr];
          // the bit tests stay in the loop: hoisted, their results live in SGPR pairs that spill (conv_direct.hip)
          asm volatile("" : "+s"(dd[rr]));
        }
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) a[0][mf] = a_base[mf * 16];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) bb[nf] = b_base[bpix[nf]];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          __builtin_amdgcn_sched_barrier(0);
          if (!(HZ && ((dd[nf >> 1] >> t) & 1u))) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
              if (mf < MF - 1) MfmaAsm<T>::run(acc[mf][nf], a[t & 1][mf], bb[nf]);
              else MfmaAsm<T>::run_pad(acc[mf][nf], a[t & 1][mf], bb[nf]);
            }
          }
          if constexpr (HZ != 0) {
            // replicate-row border term: this group's row has its MIRRORED tap dead (t - 6 at the first row, t + 6 at
            // the last): that tap's operand would be the clamped copy of the border row, i.e. the pixel fragment of tap
            // (dy = 0, dx), through the weights in registers right now
            if (t < 3 || t >= 6) {
              const int tm = t < 3 ? t + 6 : t - 6;
              if ((dd[nf >> 1] >> tm) & 1u) {
                const uint4 bx = b_base[bpix[nf] + ICOLS + t % 3];
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                  if (mf < MF - 1) MfmaAsm<T>::run(acc[mf][nf], a[t & 1][mf], bx);
                  else MfmaAsm<T>::run_pad(acc[mf][nf], a[t & 1][mf], bx);
                }
              }
            }
            // The branches of this variant put compiler-generated VALU instructions (v_cndmask of the next group's
            // condition) right behind a group's last MFMA -- into a register that MFMA is still reading as its B operand
            // (found by scripts/audit_asm_mfma.py; the hazard recogniser does not see inside the asm).  Five wait states
            // cover a 4-pass MFMA's source reads: they are part of the group's last statement (run_pad).
          }
          __builtin_amdgcn_sched_barrier(0);
          // Measured against this form (gpurun_out/r7e): the same loads over the first 6 or 4 taps, input slots before
          // weight slots, a share per pixel-fragment group -- each 1-6 % slower on the stride-2 layers.
          if (SPREAD != 0 && nf == NF - 1 && more) {   // this tap's ninth of the next stage's loads
            issue_in(ncc * kchunk, t * NI / 9, (t + 1) * NI / 9);
            if (nchunks > 1) issue_w(ncc * kchunk, t * NW / 9, (t + 1) * NW / 9);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (t + 1 < 9) {
            const int ky = (t + 1) / 3, kx = (t + 1) % 3;
            const int tapoff = S == 2 ? (ky * 2 + (kx & 1)) * PITCH + (kx >> 1) : ky * ICOLS + kx;
            bb[nf] = b_base[bpix[nf] + tapoff];
#pragma unroll
            for (int k = 0; k < APG; ++k) a[(t + 1) & 1][nf * APG + k] = a_base[(t + 1) * 64 + (nf * APG + k) * 16];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    if (cc == nchunks - 1 && !(C8_ABL & 16)) {        // tile finished: epilogue, reset accumulators
      mfma_drain();
      const int w0 = ((st0 + tile) * GN + gn) * 32;
      const TY* rbase = reinterpret_cast<const TY*>(p.resid);
      // The residual (the data gradient's sibling branch) is fetched for the WHOLE tile before the first store: `y` and
      // `resid` may alias as far as the compiler knows, so a load written next to its store stays behind every earlier
      // store -- eight dependent round trips per tile, and a block walks up to eight tiles.
      // (data-gradient instances only: the forward instances are at their register limit and run without a residual)
      constexpr bool RPRE = HZ != 0;
      vec16<TY> vres[RPRE ? NF : 1][MF / 2];
      if (RPRE && rbase) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int gh = h0 + wave4 * RW + (nf >> 1);
          const int gw = w0 + (nf & 1) * 16 + lr;
          const bool live = gh < p.Hg && gw < p.Wg;
          const int64_t off = live ? (((int64_t)b * p.Hg + gh) * p.Wg + gw) * p.ldy + o0 : o0;   // dead lanes: a valid row
#pragma unroll
          for (int mf = 0; mf < MF; mf += 2)
            vres[nf][mf / 2].load(rbase + off + mf * 16 + ((lc & 1) ? 16 + 4 * (lc - 1) : 4 * lc));   // pack_pair_bf16's channel offset
        }
      }
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int gh = h0 + wave4 * RW + (nf >> 1);
        const int gw = w0 + (nf & 1) * 16 + lr;
        const bool live = gh < p.Hg && gw < p.Wg && !((C8_ABL & 1) && acc[0][nf][0] != 12345.678f);
        TY* const row = y + (((int64_t)b * p.Hg + gh) * p.Wg + gw) * p.ldy;
        float f[MF][4];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const float4 b4 = *reinterpret_cast<const float4*>(&s_bias[gm * 64 + mf * 16 + lc * 4]);
          const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = fmaf(acc[mf][nf][r], ws, bq[r]);
            if (p.act == 3) t = fmaxf(t, t * p.alpha) * p.scale;   // leaky ReLU, 0 <= alpha <= 1
            f[mf][r] = t;
          }
        }
#pragma unroll
        for (int mf = 0; mf < MF; mf += 2) {
          uint4 pk;
          const int co = pack_pair_bf16(f[mf], f[mf + 1], lc, pk);   // every lane takes part in the exchange
          if (live) {
            TY* q = row + o0 + mf * 16 + co;
            if (rbase) {
              vec16<TY> va, vr;
              va.raw = pk;
              if constexpr (RPRE) vr = vres[RPRE ? nf : 0][mf / 2];
              else vr.load(rbase + (q - y));
#pragma unroll
              for (int j = 0; j < 8; ++j) va.set(j, va.get(j) + vr.get(j));
              pk = va.raw;
            }
            if (p.nt) {
              const u32x4 v4 = {pk.x, pk.y, pk.z, pk.w};
              __builtin_nontemporal_store(v4, reinterpret_cast<u32x4*>(q));
            } else {
              *reinterpret_cast<uint4*>(q) = pk;
            }
          }
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
          // the zeros are written HERE: sunk into the next stage's tap loop they land between asm MFMAs, in registers
          // one of them is still reading as an operand (scripts/audit_asm_mfma.py found exactly that)
          asm volatile("" : "+v"(acc[mf][nf]));
        }
      }
    }
    tile = ntile_i; cc = ncc;
  }
}

template <typename T, typename TY, int S, int GM, int GN, int RW, int HZ, bool WIMG>
int launch8w(void* y, const void* x, const void* w, C8 p, hipStream_t st) {
  using Cf = C8Cfg<T, TY, S, GM, GN, RW>;
  constexpr int TH = Cf::TH;
  static_assert(Cf::LDS <= 160 * 1024 - 1024, "LDS image");
  const int tiles_w = (p.Wg + 31) / 32, tiles_h = (p.Hg + TH - 1) / TH, tiles_o = p.O / (64 * GM);
  const int nsup = (tiles_w + GN - 1) / GN;
  // ring wrap of the kernel: -1 <= gw < 3 * Win
  if ((nsup * GN * 32 - 1) * S + 2 >= 3 * p.Win) return -2;
  auto kern = conv8_kernel<T, TY, S, GM, GN, RW, HZ, WIMG>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cf::LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  // one block per CU: walk several super-tiles per block while the grid still fills the chip
  const int64_t blocks1 = (int64_t)nsup * tiles_h * p.B * tiles_o;
  int tpb = (int)(blocks1 / 256);
  tpb = tpb < 1 ? 1 : (tpb > nsup ? nsup : tpb);
  tpb = tpb > 8 ? 8 : tpb;
  static const int tpb_env = getenv("DGV2_CONV8_TPB") ? atoi(getenv("DGV2_CONV8_TPB")) : 0;   // experiments
  if (tpb_env > 0) tpb = tpb_env > nsup ? nsup : tpb_env;
  p.tpb = tpb;
  p.nt = (!p.resid && nt_output((int64_t)p.B * p.Hg * p.Wg * p.ldy * sizeof(TY))) ? 1 : 0;
  dim3 grid((nsup + tpb - 1) / tpb, tiles_h * p.B, tiles_o);
  p.xcd = 0;
  if (tiles_o >= 2) {
    p.xcd = 1;
    p.gx = grid.x; p.gy = grid.y; p.gz = grid.z;
    const int64_t npt = ((int64_t)grid.x * grid.y + 7) / 8 * 8;
    grid = dim3((unsigned)(npt * grid.z), 1, 1);
  }
  kern<<<grid, Cf::NT, Cf::LDS, st>>>((TY*)y, (const T*)x, (const T*)w, p);
  return 0;
}

template <typename T, typename TY, int S, int GM, int GN, int RW, int HZ = 0>
int launch8(void* y, const void* x, const void* w, C8 p, hipStream_t st) {
  if (p.wimg) {
    if constexpr (sizeof(T) == 2) return launch8w<T, TY, S, GM, GN, RW, HZ, true>(y, x, w, p, st);
    else return -2;   // no e4m3 image yet
  }
  if constexpr (HZ != 0) return -2;   // the data gradient is reached through dgv2_conv3x3_dgrad8 (image) only
  else return launch8w<T, TY, S, GM, GN, RW, HZ, false>(y, x, w, p, st);
}

}  // namespace

// Called by conv_direct.hip's dispatcher for single-class launches whose taps are the full 3x3 grid in order (f33),
// forward form (rows clamp, ring wrap, pad 1, ioff 0, out_stride 1, overwrite).  Returns -2 when this engine does not
// cover the geometry (the caller goes on to conv_pipe_kernel).  dtype: DGV2_BF16 or DGV2_FP8 (y bf16 either way).
int dgv2_conv8_try(void* y, int ldy, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg, int Wg, int O,
                   int in_stride, int wtaps, int widx0, int wstep, const float* bias, const float* acc_scale,
                   const void* resid, int act, float alpha, float scale, int dtype, hipStream_t st, int wimg, int hzero) {
  static const bool off = getenv("DGV2_NO_CONV8") != nullptr;                 // A/B switch for benchmarking
  static const char* s1env = getenv("DGV2_CONV8_S1");                        // experiments: "gn", "gm", "off"
  if (off || O % 64 || Wg < 32 || Hg < 4) return -2;
  const int kstep = dtype == DGV2_FP8 ? 64 : 32;
  if (Cin % kstep || Cin < 2 * kstep) return -2;                             // the prologue amortises over >= 2 K-chunks
  if (!aligned16(y) || !aligned16(x) || !aligned16(w) || (resid && !aligned16(resid)) || ldy % 8) return -2;
  C8 p;
  p.B = B; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hg = Hg; p.Wg = Wg; p.O = O; p.ldy = ldy;
  p.wtaps = wtaps; p.widx0 = widx0; p.wstep = wstep; p.wimg = wimg;
  p.bias = bias; p.acc_scale = acc_scale; p.resid = resid; p.act = act; p.alpha = alpha; p.scale = scale;
  p.tpb = 1; p.nt = 0; p.xcd = 0; p.gx = p.gy = p.gz = 1;
#ifdef DGV2_ABLATE
  p.ablate = getenv("DGV2_C8_ABLATE") ? atoi(getenv("DGV2_C8_ABLATE")) : 0;
#endif
  int rc = -2;
#define C8_GO(TT, S_, GM_, GN_, RW_) rc = launch8<TT, bf16_t, S_, GM_, GN_, RW_>(y, x, w, p, st)
  if (in_stride == 2) {
    if (Hin != 2 * Hg || Win != 2 * Wg || O % 128 || hzero) return -2;
    if (dtype == DGV2_BF16) { if (Hg >= 8) C8_GO(bf16_t, 2, 2, 1, 2); else C8_GO(bf16_t, 2, 2, 1, 1); }
    else { if (Hg >= 8) C8_GO(fp8_t, 2, 2, 1, 2); else C8_GO(fp8_t, 2, 2, 1, 1); }
  } else if (in_stride == 1 && hzero) {
    // the stride-1 data gradient (the caller has checked the canonical form: taps (dy, dx) in order, weight slots affine,
    // the six replicate-row extras): a border row may not be its own mirror (H >= 2 holds: Hg >= 8)
    if (Hin != Hg || Win != Wg || Hg < 8 || dtype != DGV2_BF16) return -2;
    if (O % 128 == 0) rc = launch8<bf16_t, bf16_t, 1, 2, 1, 2, 1>(y, x, w, p, st);
    else if (Wg >= 64) rc = launch8<bf16_t, bf16_t, 1, 1, 2, 2, 1>(y, x, w, p, st);
  } else if (in_stride == 1) {
    if (Hin != Hg || Win != Wg || Hg < 8 || dtype != DGV2_BF16) return -2;
    // two channel slabs per halo tile where the layer has them (measured: 16x128 128->128 85.9 vs 88.8 us, 8x64 256->256
    // 76.0 vs 80.9, profiles/round4_mb_conv8.txt), two pixel tiles per weight slab otherwise (O = 64)
    if (s1env && !strcmp(s1env, "off")) return -2;
    const bool want_gm = s1env ? !strcmp(s1env, "gm") : (O % 128 == 0);
    if (want_gm) { if (O % 128) return -2; C8_GO(bf16_t, 1, 2, 1, 2); }
    else { if (Wg < 64) return -2; C8_GO(bf16_t, 1, 1, 2, 2); }
  }
#undef C8_GO
  return rc;
}

// The forward 3x3 ring conv (stride 1 or 2, pad 1) on the weight IMAGE of dgv2_conv_weight_bank_ex:
//   y [B, Hin / stride, Win / stride, O] (bf16) = act( conv(x [B, Hin, Win, Cin], w) + resid + bias ) * scale.
// DGV2_ENOTSUP where the eight-wave engine does not cover the geometry (callers then run dgv2_conv_taps on the
// row-layout weights).
extern "C" int dgv2_conv3x3_fwd8(void* y, const void* x, const void* w8, int B, int Hin, int Win, int Cin, int O,
                                 int stride, const float* bias, const void* resid, int act, float alpha, float scale,
                                 int dtype, void* stream) {
  if (!y || !x || !w8 || B < 1 || Hin < 1 || Win < 1 || Cin < 1 || O < 1 || (stride != 1 && stride != 2)) return DGV2_EINVAL;
  if (act != 0 && act != 3) return DGV2_EINVAL;
  if (dtype != DGV2_BF16) return DGV2_ENOTSUP;
  if (Hin % stride || Win % stride) return DGV2_ENOTSUP;
  const int rc = dgv2_conv8_try(y, O, x, w8, B, Hin, Win, Cin, Hin / stride, Win / stride, O, stride, 9, 0, 1, bias, nullptr,
                                resid, act, alpha, scale, dtype, (hipStream_t)stream, 1, 0);
  if (rc == -2) return DGV2_ENOTSUP;
  if (rc) return rc;
  DGV2_RETURN_LAST();
}

// The stride-1 data gradient of the same 3x3 ring conv on the TRANSPOSED staging image (w8t of dgv2_conv_weight_bank_ex):
//   gx [B, H, W, C] (bf16) = sum_{ky,kx,o} gy[B, Hz(h + 1 - ky), wrap(w + 1 - kx), o] * w[o, ky, kx, c]
//                            + the replicate-row terms of rows 0 and H - 1  (+ resid: the gradient of a sibling branch).
// DGV2_ENOTSUP where the engine does not cover the geometry (callers then run dgv2_conv_taps_ex on wt).
extern "C" int dgv2_conv3x3_dgrad8(void* gx, const void* gy, const void* w8t, int B, int H, int W, int C, int O,
                                   const void* resid, int dtype, void* stream) {
  if (!gx || !gy || !w8t || B < 1 || H < 1 || W < 1 || C < 1 || O < 1) return DGV2_EINVAL;
  if (dtype != DGV2_BF16) return DGV2_ENOTSUP;
  // the launch contracts over the conv's OUTPUT channels (its K) and produces the conv's INPUT channels
  const int rc = dgv2_conv8_try(gx, C, gy, w8t, B, H, W, O, H, W, C, 1, 9, 0, 1, nullptr, nullptr, resid, 0, 0.2f, 1.f, dtype,
                                (hipStream_t)stream, 1, 1);
  if (rc == -2) return DGV2_ENOTSUP;
  if (rc) return rc;
  DGV2_RETURN_LAST();
}
