// Direct (LDS halo-tile) convolution on the MFMA cores, channels-last, with a generic TAP LIST.
//
// One engine covers every dense conv of the discriminator and their data gradients
// (reference: ops.Conv2d in gans/models/ops/common.py:187-210 at gans/models/dusty_v2.py:325-385,
// and the cuDNN dgrad autograd would call):
//   forward 3x3 / 1x1, stride 1 or 2      : taps (ky-1, kx-1), input coord = out * stride + d
//   data gradient, stride 1                : taps (1-ky, 1-kx) on gy with transposed weights
//   data gradient, stride 2                : four parity classes, each a 1/2/2/4-tap conv on gy whose
//                                            outputs are scattered with stride 2 (no zero-tap work)
//   replicate-row border terms of dgrad    : 1-row launches in accumulate mode
// Ring padding = wrap of the W coordinate while the halo tile is staged; H is clamped (forward,
// replicate padding) or zero-filled (gradients).  Nothing padded is ever materialised.
//
// Two kernels:
//   conv_pipe_kernel    the production path.  Block = 4 waves, tile = (4*RW) x 32 output positions x TO
//                       channels, wave w owns RW tile rows.  A block walks `tpb` tiles along W and, per tile,
//                       the Cin/32 (bf16) 64-byte channel chunks; that (tile, chunk) sequence is software
//                       pipelined: the global loads of stage s+1 are issued into registers before the MFMA
//                       loop of stage s and written to the (single) LDS buffer after it, so HBM/L2 latency,
//                       MFMA work and the epilogue stores of consecutive stages overlap inside one block.
//                       All per-slot address arithmetic is hoisted out of the stage loop.
//   conv_direct_kernel  the simple synchronous version, kept as the fallback for geometries the pipelined
//                       kernel's fixed register budget does not cover (very wide halos, Win < 32).
// LDS images of conv_pipe_kernel: four 16-byte PLANES per operand -- plane c holds bytes [16c, 16c+16) of every
// pixel's (weight row's) 64-byte K chunk, rows consecutive, plane stride a multiple of 256 bytes.  A fragment read
// (lane (lr, lc) takes row base + lr of plane lc) is then one conflict-free ds_read_b128 at ANY base -- a
// ds_read_b128 is served in the lane groups {lc: lr 0-3, 12-15; lc+1: lr 4-11} etc., which cover 16 consecutive
// 16-byte units modulo the plane stride -- and the address is LINEAR in the row: a tap is a constant byte offset,
// for the full 3x3 grid an instruction immediate (F33), so the MFMA loop carries no address arithmetic at all.
// (The pixel-major image with the (row>>2)&3 XOR of gemm_core.h that this replaces cost 5 VALU per fragment and
// still conflicted on 37 % of its LDS cycles: profiles/round2_sq_counters.txt.)  Staging slot id -> row
// (id>>5)*8 + (id&7), plane (id>>3)&3: a wave still loads 16 whole pixels per instruction and its ds_write_b128
// lane groups (8 consecutive lanes) hit 8 consecutive units of one plane.
// conv_direct_kernel keeps the pixel-major swizzled image.
#include <type_traits>

#include "gemm_core.h"

#ifndef DGV2_S2D_ASM
#define DGV2_S2D_ASM 0
#endif

namespace {

constexpr int DTH = 4;
constexpr int DTW = 32;

struct DConv {
  int B, Hin, Win, Cin;
  int Hg, Wg, O, Hy, Wy;
  int ldy;          // channel stride of y / resid rows (>= O: the launch may write a channel range of a wider tensor)
  int in_stride, ioff_h, ioff_w;
  int out_stride, ooff_h, ooff_w;
  int ntaps, wtaps;
  int dy[9], dx[9], widx[9];
  int dymin, dxmin, rows, cols;
  int hzero, ring, accumulate;
  int tpb;          // tiles per block along W (pipelined kernel)
#ifdef DGV2_ABLATE   // benchmarking builds only (make ABLATE=1): wrong results by design, never in the shipped library
  int ablate;       // DGV2_CP_ABLATE: 1 skip stores, 2 skip the MFMA loops, 4 skip input loads, 8 skip weight loads, 16 skip epilogue
#define CP_ABL (p.ablate)
#else
#define CP_ABL 0
#endif
  int s2d;          // four classes = the canonical stride-2 3x3 data gradient (host-checked tap list): unrolled taps
  int wres;         // tap-list kernels, 2 K-chunks: both chunks' weight slabs stay in LDS for the block's whole walk
  int nt;           // streaming output stores (outputs >= DGV2_NT_MIN_MB that no residual read revisits)
  // XCD-aware block order (1-D launch): workgroup n runs on XCD n % 8 (each XCD has its own L2), so the gz blocks that
  // share one pixel tile's input -- one per output-channel tile -- are given ids 8 apart: same XCD, dispatched back to
  // back, the halo tile is fetched into that L2 once instead of once per channel tile from HBM / MALL
  int xcd, gx, gy, gz;
  // image pairs (4-row maps): the launch sees B/2 stacked pairs of `hper`-row images as 2*hper-row maps, so an 8-row
  // tile covers two images and the weight slab is staged once for both; rows clamp / zero-fill per image and each image
  // brings its own `hrows` halo rows
  int hper, hrows;
  // output classes: taps [cls_t0[c], cls_t0[c+1]) accumulate into class c, written at offsets (cls_ooh, cls_oow)
  // (one class = the plain conv; four = the parity classes of the stride-2 data gradient in ONE launch)
  int ncls, cls_t0[5], cls_ooh[4], cls_oow[4];
  // border extras: tap (x_dy, x_dx, weight slot x_slot) added to class x_cls for output row x_row only
  // (the replicate-padding rows of the data gradient, formerly separate one-row accumulate launches)
  int nx, x_dy[6], x_dx[6], x_slot[6], x_cls[6], x_row[6];
  // The six extras are exactly the replicate-row terms of the stride-1 3x3 data gradient (host-checked: border_ok).
  // The unrolled-tap data-gradient instances (F33 = 2) then run WITHOUT extras (border = 1, nx = 0): the term of a dead
  // tap (dy = -1 at output row 0, dy = +1 at the last row) is the border row itself through the weights of the tap
  // mirrored in dy, so it is issued next to THAT tap's MFMAs (weights in registers, one more pixel-fragment read: tap
  // (0, dx)).  A border row costs what an interior row costs and the per-chunk extras loop is gone.
  int border_ok, border;
  float inv_cols;   // 1 / cols
  const float* bias;
  const float* acc_scale;   // device scalar multiplied onto the accumulators ahead of the residual / bias (e4m3 weights:
                            // EqualLR factor / the tensor's power-of-two scale, fp8.hip), or nullptr
  void* ybase;         // = y (lets the epilogue address resid at the same offset)
  const void* resid;   // optional residual, same layout as y: added before bias / activation
  int act;
  float alpha, scale;
};

// Epilogue shared by both kernels: one accumulator fragment (16 channels x 16 positions) -> y.
// bias4: this lane's four bias values (registers: a global bias read here would make the compiler drain every
// prefetch in flight with vmcnt(0) before each tile's stores), or nullptr
template <typename T>
__device__ __forceinline__ void store_frag(const DConv& p, T* __restrict__ row, int o, f32x4 v,
                                           const float* bias4 = nullptr, float ws = 1.f) {
  const T* rrow = p.resid ? reinterpret_cast<const T*>(p.resid) + (row - reinterpret_cast<T*>(p.ybase)) : nullptr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (o + r >= p.O) continue;
    float f = v[r] * ws;
    if (p.accumulate) f += to_f32(row[o + r]);
    if (rrow) f += to_f32(rrow[o + r]);
    if (bias4) f += bias4[r];
    else if (p.bias) f += p.bias[o + r];
    if (p.act == 3) f = (f > 0.f ? f : f * p.alpha) * p.scale;
    v[r] = f;
  }
  if (o + 3 < p.O && (p.O & 3) == 0) {
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<f32x4*>(row + o) = v;
    } else {
      union { uint2 u; bf16_t e[4]; } pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk.e[r] = (bf16_t)v[r];
      *reinterpret_cast<uint2*>(row + o) = pk.u;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (o + r < p.O) row[o + r] = from_f32<T>(v[r]);
  }
}

// ---------------------------------------------------------------------------------------------
// Pipelined kernel.  NI = input-tile staging slots (16 B each) per thread, NW = weight slots.
// ---------------------------------------------------------------------------------------------
// TO <= 32: a third wave per SIMD (<= 168 VGPRs) keeps more staging loads in flight on the HBM-bound small-channel layers
// F33 != 0: the taps are the full 3x3 grid in (dy, dx) order (stride 1; 3: input stride 2) and O is a multiple of TO (host-checked):
// tap loop unrolled, every LDS offset an immediate, staging loads unconditional.  1 = rows clamp (forward convs),
// 2 = rows outside the image are zero (data gradients): those rows are staged as whatever the clamped address holds,
// because every tap that would read them is a dead tap of that output row and is skipped (the host checks that the
// border extras read real rows).
// TY: type of y / resid (= T, or bf16 for e4m3 operands T = fp8_t)
#ifndef DGV2_CP_SPREAD
#define DGV2_CP_SPREAD 1
#endif
template <typename T, typename TY, int TO, int RW, int NI, int NC, int F33>
__global__ __launch_bounds__(256, (TO <= 32 && NC == 1 && NI <= 6) ? 3 : 2) void conv_pipe_kernel(TY* __restrict__ y, const T* __restrict__ x,
                                                           const T* __restrict__ w, DConv p) {
  constexpr int CE = 16 / sizeof(T);
  const float ws = p.acc_scale ? *p.acc_scale : 1.f;   // uniform address: a scalar load, long before the first epilogue
  constexpr int MF = TO / 16, NF = 2 * RW, TH = 4 * RW;
  constexpr int NW = (TO * 9 * 4 + 255) / 256;
  constexpr int PIN = NI * 64, PW = NW * 64;   // rows per plane: exactly what the staging slots cover
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  __shared__ int s_widx[9], s_tapoff[9], s_t0[5], s_xoff[6], s_xslot[6], s_xcls[6], s_xrow[6];
  __shared__ __attribute__((aligned(16))) float s_bias[TO];
  const int npix = p.rows * p.cols;
  uint4* lds_in = smem;
  uint4* lds_w = smem + 4 * PIN;
  const int n_wrows = TO * p.ntaps;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lc = lane >> 4;
  const int tiles_h = (p.Hg + TH - 1) / TH;
  const int tiles_w = (p.Wg + DTW - 1) / DTW;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd) {
    const int n = blockIdx.x, q = n >> 3;
    const int pt = (q / p.gz) * 8 + (n & 7);      // pixel tile (x fastest); the grid is padded to whole groups of 8
    if (pt >= p.gx * p.gy) return;
    bz = q % p.gz;
    bx = pt % p.gx;
    by = pt / p.gx;
  }
  const int b = by / tiles_h;
  const int h0 = (by % tiles_h) * TH;
  const int o0 = bz * TO;
  const int tw0 = bx * p.tpb;
  const int ntile = min(p.tpb, tiles_w - tw0);
  const T* xb = x + (int64_t)b * p.Hin * p.Win * p.Cin;
  const int kchunk = 4 * CE;
  const int nchunks = p.Cin / kchunk;

  if (tid == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      s_widx[t] = p.widx[t];
      s_tapoff[t] = (p.dy[t] - p.dymin) * p.cols + p.dx[t] - p.dxmin;
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) s_t0[c] = p.cls_t0[c];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      s_xoff[e] = (p.x_dy[e] - p.dymin) * p.cols + p.x_dx[e] - p.dxmin;
      s_xslot[e] = p.x_slot[e];
      s_xcls[e] = p.x_cls[e];
      s_xrow[e] = p.x_row[e];
    }
  }
  __syncthreads();

  // ---- per-slot constants (once per block) ----
  const int slot_row = ((tid >> 5) << 3) | (tid & 7);   // row (pixel / weight row) of this thread's slot 0; slot j: + 64 j
  const int slot_ch = (tid >> 3) & 3;                   // its 16-byte plane
  static_assert(F33 == 0 || TO == 64, "F33: weight slot j of a thread is tap j");
  const unsigned wlane = ((o0 + slot_row) * p.wtaps) * p.Cin + slot_ch * CE;   // F33: element offset of this thread's weight row
  int woff[F33 ? 1 : NW];   // element offset of the slot's 16 bytes in w at chunk 0, or -1
#pragma unroll
  for (int j = 0; j < (F33 ? 0 : NW); ++j) {
    const int row = slot_row + 64 * j;
    const int t = row / TO, r = row & (TO - 1);
    woff[j] = (row < n_wrows && o0 + r < p.O) ? ((o0 + r) * p.wtaps + s_widx[t < 9 ? t : 0]) * p.Cin + slot_ch * CE : -1;
  }
  int ipos[NI];   // (iy << 16) | ix of the slot's pixel inside the halo tile
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int pix = slot_row + 64 * j;
    const int iy = (int)(((float)pix + 0.5f) * p.inv_cols);
    ipos[j] = (iy << 16) | (pix - iy * p.cols);
  }
  // A block's tiles share their rows: the H part of every slot's address (clamp / zero fill) is computed once,
  // per tile only the W coordinate is wrapped.
  int grow[NI];   // clamped row offset (elements) + channel offset of the slot, or -1 (zero fill)
  {
    const int gh_base = h0 * p.in_stride + p.ioff_h + p.dymin;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int gh = gh_base + (ipos[j] >> 16);
      int hi = p.Hin, img0 = 0;
      if (p.hper) {   // image pair: halo row -> (image k of the pair, row inside that image)
        const int iy = ipos[j] >> 16, k = iy >= p.hrows ? 1 : 0;
        gh = iy - k * p.hrows + p.ioff_h + p.dymin;
        hi = p.hper;
        img0 = k * p.hper;
      }
      const bool zero = !F33 && (slot_row + 64 * j >= npix || (p.hzero && (gh < 0 || gh >= hi)));
      gh = gh < 0 ? 0 : (gh >= hi ? hi - 1 : gh);
      grow[j] = zero ? -1 : (img0 + gh) * p.Win * p.Cin + slot_ch * CE;
    }
  }
  int goff[NI];   // element offset inside image b of the slot's 16 bytes at chunk 0, or -1 (zero fill)
  auto tile_offsets = [&](int tw) {
    const int gw_base = tw * DTW * p.in_stride + p.ioff_w + p.dxmin;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int gw = gw_base + (ipos[j] & 0xffff);
      if (p.ring) {   // -Win <= gw < 4*Win (host-checked)
        gw = gw < 0 ? gw + p.Win : gw;
        gw = gw >= 2 * p.Win ? gw - 2 * p.Win : gw;
        gw = gw >= p.Win ? gw - p.Win : gw;
      } else {
        gw = gw < 0 ? 0 : (gw >= p.Win ? p.Win - 1 : gw);
      }
      goff[j] = (!F33 && grow[j] < 0) ? -1 : grow[j] + gw * p.Cin;
    }
  };

  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 rin[NI], rwt[NW];   // native vectors: as HIP's uint4 structs the unconditional F33 loads left both arrays in scratch
  // slots [j0, j1) of a stage (F33: a ninth per tap inside the MFMA loop -- a wave that issues a whole stage in one burst
  // waits at the full memory queue until most of it has returned, and its MFMA loop starts when the loads are nearly over;
  // conv8.hip, measured there)
  auto issue_in = [&](int c0, int j0 = 0, int j1 = 1 << 20) {
    if (CP_ABL & 4) return;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (j < j0 || j >= j1) continue;
      if constexpr (F33 != 0) {
        rin[j] = *reinterpret_cast<const u32x4*>(xb + c0 + (unsigned)goff[j]);
      } else {
        rin[j] = (u32x4){0u, 0u, 0u, 0u};
        if (goff[j] >= 0) rin[j] = *reinterpret_cast<const u32x4*>(xb + goff[j] + c0);
      }
    }
  };
  auto issue_w = [&](int c0, int j0 = 0, int j1 = 1 << 20) {
    if (CP_ABL & 8) return;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      if (j < j0 || j >= j1) continue;
      if constexpr (F33 != 0) {   // uniform pointer of tap j's weights + this thread's 32-bit offset (TO = 64: slot j IS tap j)
        const T* wu = w + (p.widx[0] + j * (p.widx[1] - p.widx[0])) * p.Cin + c0;
        rwt[j] = *reinterpret_cast<const u32x4*>(wu + wlane);
      } else {
        rwt[j] = (u32x4){0u, 0u, 0u, 0u};
        if (woff[j] >= 0) rwt[j] = *reinterpret_cast<const u32x4*>(w + woff[j] + c0);
      }
    }
  };
  uint4* const st_in = lds_in + slot_ch * PIN + slot_row;   // staging destinations of slot 0; slot j: + 64 j
  uint4* const st_w = lds_w + slot_ch * PW + slot_row;

  f32x4 acc[NC][MF][NF];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[c][mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int bpix[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
  {
    const int r = wave * RW + (nf >> 1);   // output row of the tile; image pairs: row r % hper of image r / hper
    const int hrow = p.hper ? (r / p.hper) * p.hrows + r % p.hper : r * p.in_stride;
    bpix[nf] = hrow * p.cols + ((nf & 1) * 16 + lr) * p.in_stride;
  }
  // the tile's bias in LDS: read in the epilogue without a vector-memory wait (a global read there would drain
  // every prefetch in flight) and without holding MF*4 registers through the MFMA loop
  if (tid < TO) s_bias[tid] = (p.bias && o0 + tid < p.O) ? p.bias[o0 + tid] : 0.f;
  __syncthreads();
  const uint4* const a_base0 = lds_w + lc * PW + lr;   // A fragment mf of tap t: a_base[t * TO + mf * 16]
  const uint4* const b_base = lds_in + lc * PIN;       // B fragment nf of tap t: b_base[bpix[nf] + tapoff(t)]

  // Taps that read only the zero rows above / below the image for this wave's output row(s) (hzero: the data
  // gradient of a replicate-padded conv) contribute nothing: skipped, wave-uniformly.  Together with the border
  // extras -- which exist for exactly those rows -- a border row costs what an interior row costs; for the 4-row
  // maps (one row per wave) that removes a third of the MFMA time of waves 0 and 3, i.e. of the block.
  unsigned dead[RW];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int orow = h0 + wave * RW + rr;
    const int gin = (p.hper ? orow % p.hper : orow) * p.in_stride + p.ioff_h;
    const unsigned hin = p.hper ? p.hper : p.Hin;
    unsigned m = 0;
    for (int t = 0; t < p.ntaps; ++t)
      if (p.hzero && (unsigned)(gin + p.dy[t]) >= hin) m |= 1u << t;
    dead[rr] = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
  }
  unsigned dead_all = dead[0];
#pragma unroll
  for (int rr = 1; rr < RW; ++rr) dead_all &= dead[rr];

  // ---- pipeline over (tile, chunk) stages ----
  // wres (host: tap-list form, exactly two K-chunks, LDS to spare): the weight slabs of BOTH chunks are staged once,
  // side by side, and stay for every tile the block walks -- re-staging the slab per (tile, chunk) stage was 36 of
  // the 62 KB a stage of the stride-2 data gradients moved through L2 -> LDS.
  const bool wres = !F33 && p.wres;
  if (wres) {
    for (int c = 0; c < 2; ++c) {
      issue_w(c * kchunk);
#pragma unroll
      for (int j = 0; j < NW; ++j) *reinterpret_cast<u32x4*>(st_w + c * 4 * PW + 64 * j) = rwt[j];
    }
  }
  tile_offsets(tw0);
  issue_in(0);
  if (!wres) issue_w(0);
  int tile = 0, cc = 0;             // stage being computed
  const int nstage = ntile * nchunks;
  for (int s = 0; s < nstage; ++s) {
    __syncthreads();                // every wave has finished reading stage s-1
#pragma unroll
    for (int j = 0; j < NI; ++j) *reinterpret_cast<u32x4*>(st_in + 64 * j) = rin[j];
    if (!wres && (s == 0 || nchunks > 1)) {
#pragma unroll
      for (int j = 0; j < NW; ++j) *reinterpret_cast<u32x4*>(st_w + 64 * j) = rwt[j];
    }
    // prefetch stage s+1
    int ntile_i = tile, ncc = cc + 1;
    if (ncc == nchunks) { ncc = 0; ++ntile_i; }
    const bool more = s + 1 < nstage;
    // (not the four-class tap list of the stride-2 data gradient: with its predicated zero-fill loads among the builtin
    // MFMAs that kernel spills and runs 90 -> 128 us, gpurun_out/r7g)
    constexpr bool SPREAD = F33 != 0 && DGV2_CP_SPREAD != 0;
    if (more && ncc == 0) tile_offsets(tw0 + ntile_i);
    if (more && (!SPREAD || (CP_ABL & 2))) {
      issue_in(ncc * kchunk);
      if (!wres && nchunks > 1) issue_w(ncc * kchunk);
    }
    __syncthreads();                // stage s visible in LDS
    const uint4* const a_base = a_base0 + (wres ? cc * 4 * PW : 0);

    if constexpr (F33) {
      // The nine taps straight-line: immediate LDS offsets; a pixel fragment is re-read for tap t+1 as soon as its
      // four MFMAs of tap t are issued (same registers), the weight fragments alternate between two sets, so every
      // read has three quarters of a tap's MFMA time to land; dead taps are skipped under wave-uniform branches around
      // in-place MFMAs.  The fences pin that order: left alone the scheduler hoists reads of several taps ahead and spills.
      constexpr int COLS = F33 == 3 ? 2 * (DTW - 1) + 3 : DTW + 2;   // halo row length: stride 2 / stride 1
      constexpr int APG = MF / NF;   // A fragments re-read per pixel-fragment group
      static_assert(MF % NF == 0, "A fragments are re-read in equal shares behind the pixel-fragment groups");
      uint4 a[2][MF], bb[NF];
      unsigned dd[RW];
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        dd[rr] = F33 == 2 ? dead[rr] : 0u;
        // the bit tests stay in the loop: hoisted, their 27 results live in SGPR pairs that spill
        if constexpr (F33 == 2) asm volatile("" : "+s"(dd[rr]));
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
          if (!(F33 == 2 && ((dd[nf >> 1] >> t) & 1u))) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
              if (mf < MF - 1) MfmaAsm<T>::run(acc[0][mf][nf], a[t & 1][mf], bb[nf]);
              else MfmaAsm<T>::run_pad(acc[0][mf][nf], a[t & 1][mf], bb[nf]);   // compiler code may follow: wait states inside
            }
          }
          if constexpr (F33 == 2 && NI <= 6) {   // (the image-pair instances, NI = 7, are at their register limit: extras)
            // replicate-row border term (p.border): the row this group belongs to has its MIRRORED tap dead (t - 6 at the
            // first row, t + 6 at the last) -- that tap's operand would be the clamped copy of the border row, i.e. the
            // pixel fragment of tap (0, dx) = t -+ 3, and its weights are the ones in registers right now
            if (t < 3 || t >= 6) {
              const int tm = t < 3 ? t + 6 : t - 6;
              if (p.border && ((dd[nf >> 1] >> tm) & 1u)) {
                const uint4 bx = b_base[bpix[nf] + COLS + t % 3];
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                  if (mf < MF - 1) MfmaAsm<T>::run(acc[0][mf][nf], a[t & 1][mf], bx);
                  else MfmaAsm<T>::run_pad(acc[0][mf][nf], a[t & 1][mf], bx);
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (SPREAD && nf == NF - 1 && more) {   // this tap's ninth of the next stage's loads
            issue_in(ncc * kchunk, t * NI / 9, (t + 1) * NI / 9);
            if (!wres && nchunks > 1) issue_w(ncc * kchunk, t * NW / 9, (t + 1) * NW / 9);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (t + 1 < 9) {   // this group's pixel fragment is consumed: re-read it, and a share of the weights, for tap t+1
            bb[nf] = b_base[bpix[nf] + ((t + 1) / 3) * COLS + (t + 1) % 3];
#pragma unroll
            for (int k = 0; k < APG; ++k) a[(t + 1) & 1][nf * APG + k] = a_base[(t + 1) * TO + (nf * APG + k) * 16];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else if (NC == 4 && p.s2d && !(CP_ABL & 2)) {
      // The stride-2 3x3 data gradient, taps straight-line: class (ph, pw) of the output parities sees the gy pixels
      // (dy, dx) in {0, 1}^2 with dy <= ph, dx <= pw -- 1 / 2 / 2 / 4 taps in the host's order below; every LDS offset
      // is an immediate.
      if constexpr (NC == 4) {
        constexpr int COLS = DTW + 1;
        constexpr int TDY[9] = {0, 0, 0, 1, 0, 1, 1, 0, 0}, TDX[9] = {0, 1, 0, 0, 0, 1, 0, 1, 0}, TCL[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          // no dead-tap branch: the rows below the image are staged as zeros in this (tap-list) form, so the three
          // taps the last gy row cannot see multiply zeros -- straight-line builtin MFMAs, hazards left to the compiler
          uint4 a[MF], bb[NF];
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) a[mf] = a_base[t * TO + mf * 16];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) bb[nf] = b_base[bpix[nf] + TDY[t] * COLS + TDX[t]];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
#if DGV2_S2D_ASM   // experiment builds: the in-place asm form; 1 = no drain in front of the epilogue, 2 = drained
              MfmaAsm<T>::run(acc[TCL[t]][mf][nf], a[mf], bb[nf]);
#else
              Mfma16<T>::run(acc[TCL[t]][mf][nf], a[mf], bb[nf]);
#endif
            }
        }
      }
    } else if (!(CP_ABL & 2)) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        for (int t = s_t0[c]; t < s_t0[c + 1]; ++t) {
          if ((dead_all >> t) & 1u) continue;
          uint4 a[MF], bb[NF];
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) a[mf] = a_base[t * TO + mf * 16];
          const int tapoff = s_tapoff[t];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) bb[nf] = b_base[bpix[nf] + tapoff];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            if (RW > 1 && ((dead[nf >> 1] >> t) & 1u)) continue;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) Mfma16<T>::run(acc[c][mf][nf], a[mf], bb[nf]);
          }
        }
      }
    }
    // border extras: only the wave rows that ARE the named output row take part (wave-uniform tests)
    for (int e = 0; e < p.nx; ++e) {
      const int xrow = s_xrow[e], xcls = s_xcls[e], xoff = s_xoff[e], xslot = s_xslot[e];
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        if ((p.hper ? (h0 + wave * RW + rr) % p.hper : h0 + wave * RW + rr) != xrow) continue;
        uint4 a[MF], bb[2];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) a[mf] = a_base[xslot * TO + mf * 16];
#pragma unroll
        for (int h = 0; h < 2; ++h) bb[h] = b_base[bpix[rr * 2 + h] + xoff];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (c != xcls) continue;
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            if constexpr (F33) {
              MfmaAsm<T>::run(acc[c][mf][rr * 2], a[mf], bb[0]);
              MfmaAsm<T>::run(acc[c][mf][rr * 2 + 1], a[mf], bb[1]);
            } else {
              Mfma16<T>::run(acc[c][mf][rr * 2], a[mf], bb[0]);
              Mfma16<T>::run(acc[c][mf][rr * 2 + 1], a[mf], bb[1]);
            }
          }
        }
      }
    }

    if (cc == nchunks - 1 && !(CP_ABL & 16)) {        // tile finished: epilogue, reset accumulators
      if constexpr (F33) mfma_drain();
#if DGV2_S2D_ASM == 2
      if constexpr (NC == 4) mfma_drain();
#endif
      const int w0 = (tw0 + tile) * DTW;
      // fast path (block-uniform): full channel tile, plain overwrite -- straight-line bias / lrelu / convert and,
      // for bf16, fragment pairs leaving as 16-byte stores; everything else takes the general store_frag
      const bool fast = !p.accumulate && o0 + TO <= p.O && (p.O & 7) == 0;
      // data gradients have no bias, no activation and no accumulator scale: their (output-heavy: K is one or two chunks
      // per tile) epilogue then is convert + exchange + store, without the LDS bias reads and four fma / max per value
      const bool plain = !p.bias && !p.acc_scale && p.act != 3;
      const TY* rbase = reinterpret_cast<const TY*>(p.resid);   // optional residual, same layout as y
      bool paired = false;
      if constexpr (NC == 4 && sizeof(TY) == 2 && MF >= 2) {
        // The stride-2 data gradient (p.s2d: class c writes pixel (2 gh + (c >> 1), 2 gw + (c & 1))): stored class by class,
        // an instruction writes 16 pixels of 64 bytes at a 128-byte pitch -- HALF of 16 cache lines, the other half one
        // instruction later (PMC: 330 MB written for a 268 MB tensor).  The column parities of a row are classes c, c + 1
        // of the SAME lane: lanes lr and lr ^ 1 trade one of their two packed quads (DPP quad_perm, no LDS), after which
        // instruction 1 carries the pixel pairs (2 j, 2 j + 1) of the even j and instruction 2 those of the odd j --
        // whole 128-byte lines each.
        paired = fast && plain && p.s2d && !rbase;
        if (paired) {
          const bool even = (lr & 1) == 0;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const int gh = h0 + wave * RW + (nf >> 1);
            const int gwl = w0 + (nf & 1) * 16 + lr;
            const int col1 = 2 * gwl + (even ? 0 : -1), col2 = col1 + 2;
            const bool rowlive = gh < p.Hg && !((CP_ABL & 1) && acc[0][0][nf][0] != 12345.678f);
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              TY* const rowp = y + (((int64_t)b * p.Hy + 2 * gh + ph) * p.Wy) * p.ldy + o0;
#pragma unroll
              for (int mf = 0; mf < MF; mf += 2) {
                float fa[2][2][4];
#pragma unroll
                for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                  for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) fa[pw][m][r] = acc[2 * ph + pw][mf + m][nf][r];
                uint4 pk0, pk1;
                const int co = pack_pair_bf16(fa[0][0], fa[0][1], lc, pk0);
                pack_pair_bf16(fa[1][0], fa[1][1], lc, pk1);
                const uint4 send = even ? pk1 : pk0;
                uint4 recv;   // the neighbour's (lr ^ 1) quad: quad_perm [1, 0, 3, 2]
                recv.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xF, 0xF, true);
                recv.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xF, 0xF, true);
                recv.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.z, 0xB1, 0xF, 0xF, true);
                recv.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.w, 0xB1, 0xF, 0xF, true);
                const uint4 s1 = even ? pk0 : recv, s2 = even ? recv : pk1;
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                TY* const q1 = rowp + (int64_t)col1 * p.ldy + mf * 16 + co;
                TY* const q2 = rowp + (int64_t)col2 * p.ldy + mf * 16 + co;
                if (rowlive && col1 < p.Wy) {
                  if (p.nt) __builtin_nontemporal_store((u32x4){s1.x, s1.y, s1.z, s1.w}, reinterpret_cast<u32x4*>(q1));
                  else *reinterpret_cast<uint4*>(q1) = s1;
                }
                if (rowlive && col2 < p.Wy) {
                  if (p.nt) __builtin_nontemporal_store((u32x4){s2.x, s2.y, s2.z, s2.w}, reinterpret_cast<u32x4*>(q2));
                  else *reinterpret_cast<uint4*>(q2) = s2;
                }
              }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
              for (int mf = 0; mf < MF; ++mf) acc[c][mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      // pixel fragment outermost: its 64-bit row address is formed once, the classes add (ooh * Wy + oow) * ldy
#pragma unroll
      for (int nf = 0; nf < (paired ? 0 : NF); ++nf) {
        const int gh = h0 + wave * RW + (nf >> 1);
        const int gw = w0 + (nf & 1) * 16 + lr;
        const bool live0 = gh < p.Hg && gw < p.Wg;
        TY* const row0 = y + (((int64_t)b * p.Hy + gh * p.out_stride) * p.Wy + gw * p.out_stride) * p.ldy;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const bool live = live0 && !((CP_ABL & 1) && acc[c][0][nf][0] != 12345.678f);
          TY* row = row0 + (p.cls_ooh[c] * p.Wy + p.cls_oow[c]) * p.ldy;
          if (fast) {
            float f[MF][4];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
              if (plain) {
#pragma unroll
                for (int r = 0; r < 4; ++r) f[mf][r] = acc[c][mf][nf][r];
                continue;
              }
              const float4 b4 = *reinterpret_cast<const float4*>(&s_bias[mf * 16 + lc * 4]);
              const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float t = fmaf(acc[c][mf][nf][r], ws, bq[r]);
                if (p.act == 3) t = fmaxf(t, t * p.alpha) * p.scale;   // leaky ReLU, 0 <= alpha <= 1
                f[mf][r] = t;
              }
            }
            if constexpr (sizeof(TY) == 2 && MF >= 2) {
#pragma unroll
              for (int mf = 0; mf < MF; mf += 2) {
                uint4 pk;
                const int co = pack_pair_bf16(f[mf], f[mf + 1], lc, pk);   // every lane takes part in the exchange
                if (live) {
                  TY* q = row + o0 + mf * 16 + co;
                  if (rbase) {   // residual added on the packed 8-channel run (one 16-byte read)
                    vec16<TY> a, r;
                    a.raw = pk;
                    r.load(rbase + (q - y));
#pragma unroll
                    for (int j = 0; j < 8; ++j) a.set(j, a.get(j) + r.get(j));
                    pk = a.raw;
                  }
                  if (p.nt) {
                    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                    const u32x4 v4 = {pk.x, pk.y, pk.z, pk.w};
                    __builtin_nontemporal_store(v4, reinterpret_cast<u32x4*>(q));
                  } else {
                    *reinterpret_cast<uint4*>(q) = pk;
                  }
                }
              }
            } else {
#pragma unroll
              for (int mf = 0; mf < MF; ++mf) {
                if (!live) continue;
                TY* q = row + o0 + mf * 16 + lc * 4;
                if (rbase) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) f[mf][r] += to_f32(rbase[(q - y) + r]);
                }
                if constexpr (sizeof(TY) == 4) {
                  *reinterpret_cast<float4*>(q) = make_float4(f[mf][0], f[mf][1], f[mf][2], f[mf][3]);
                } else {
                  union { uint2 u; bf16_t e[4]; } pk;
#pragma unroll
                  for (int r = 0; r < 4; ++r) pk.e[r] = (bf16_t)f[mf][r];
                  *reinterpret_cast<uint2*>(q) = pk.u;
                }
              }
            }
          } else if (live) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
              const int o = o0 + mf * 16 + lc * 4;
              if (o < p.O) store_frag<TY>(p, row, o, acc[c][mf][nf], &s_bias[mf * 16 + lc * 4], ws);
            }
          }
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc[c][mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    tile = ntile_i; cc = ncc;
  }
}

// ---------------------------------------------------------------------------------------------
// Synchronous fallback: 4 x 32 tile, stage -> barrier -> MFMA per chunk.
// ---------------------------------------------------------------------------------------------
template <typename T, int TO>
__global__ __launch_bounds__(256) void conv_direct_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                          const T* __restrict__ w, DConv p) {
  constexpr int CE = 16 / sizeof(T);
  constexpr int MF = TO / 16;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int npix = p.rows * p.cols;
  uint4* lds_in = smem;
  uint4* lds_w = smem + npix * 4;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lc = lane >> 4;
  const int tiles_h = (p.Hg + DTH - 1) / DTH;
  const int b = blockIdx.y / tiles_h;
  const int h0 = (blockIdx.y % tiles_h) * DTH;
  const int w0 = blockIdx.x * DTW;
  const int o0 = blockIdx.z * TO;
  const T* xb = x + (int64_t)b * p.Hin * p.Win * p.Cin;

  f32x4 acc[MF][2];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    acc[mf][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[mf][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int nchunks = p.Cin / (4 * CE);
  const int gh_base = h0 * p.in_stride + p.ioff_h + p.dymin;
  const int gw_base = w0 * p.in_stride + p.ioff_w + p.dxmin;
  for (int cc = 0; cc < nchunks; ++cc) {
    const int c0 = cc * 4 * CE;
    __syncthreads();
    for (int id = tid; id < npix * 4; id += 256) {
      const int pix = id >> 2, ch = id & 3;
      const int iy = pix / p.cols, ix = pix - iy * p.cols;
      int gh = gh_base + iy, gw = gw_base + ix;
      bool zero = false;
      if (p.hzero) zero = gh < 0 || gh >= p.Hin;
      gh = gh < 0 ? 0 : (gh >= p.Hin ? p.Hin - 1 : gh);
      gw = p.ring ? floormod(gw, p.Win) : (gw < 0 ? 0 : (gw >= p.Win ? p.Win - 1 : gw));
      uint4 v = make_uint4(0, 0, 0, 0);
      if (!zero) v = *reinterpret_cast<const uint4*>(xb + ((int64_t)gh * p.Win + gw) * p.Cin + c0 + ch * CE);
      lds_in[pix * 4 + (ch ^ ((pix >> 2) & 3))] = v;
    }
    for (int id = tid; id < TO * p.ntaps * 4; id += 256) {
      const int r = id / (p.ntaps * 4);
      const int rem = id - r * (p.ntaps * 4);
      const int t = rem >> 2, ch = rem & 3;
      const int o = o0 + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (o < p.O) v = *reinterpret_cast<const uint4*>(w + ((int64_t)o * p.wtaps + p.widx[t]) * p.Cin + c0 + ch * CE);
      lds_w[(r * p.ntaps + t) * 4 + (ch ^ ((r >> 2) & 3))] = v;
    }
    __syncthreads();
    for (int t = 0; t < p.ntaps; ++t) {
      uint4 a[MF], bb[2];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int r = mf * 16 + lr;
        a[mf] = lds_w[(r * p.ntaps + t) * 4 + (lc ^ ((r >> 2) & 3))];
      }
      const int prow = (wave * p.in_stride + p.dy[t] - p.dymin) * p.cols + p.dx[t] - p.dxmin;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int pix = prow + (nf * 16 + lr) * p.in_stride;
        bb[nf] = lds_in[pix * 4 + (lc ^ ((pix >> 2) & 3))];
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        Mfma16<T>::run(acc[mf][0], a[mf], bb[0]);
        Mfma16<T>::run(acc[mf][1], a[mf], bb[1]);
      }
    }
  }

  const int gh = h0 + wave;
  if (gh >= p.Hg) return;
  const int yh = gh * p.out_stride + p.ooff_h;
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    const int gw = w0 + nf * 16 + lr;
    if (gw >= p.Wg) continue;
    const int yw = gw * p.out_stride + p.ooff_w;
    T* row = y + (((int64_t)b * p.Hy + yh) * p.Wy + yw) * p.ldy;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int o = o0 + mf * 16 + lc * 4;
      if (o < p.O) store_frag<T>(p, row, o, acc[mf][nf]);
    }
  }
}

template <typename T, int TO>
int launch_direct(void* y, const void* x, const void* w, DConv p, hipStream_t st) {
  p.rows = (DTH - 1) * p.in_stride + p.rows;   // p.rows / p.cols arrive as the tap extents
  p.cols = (DTW - 1) * p.in_stride + p.cols;
  const size_t lds = sizeof(uint4) * ((size_t)p.rows * p.cols * 4 + (size_t)TO * p.ntaps * 4);
  if (lds > 160 * 1024) return DGV2_EINVAL;
  auto kern = conv_direct_kernel<T, TO>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid((p.Wg + DTW - 1) / DTW, ((p.Hg + DTH - 1) / DTH) * p.B, (p.O + TO - 1) / TO);
  kern<<<grid, 256, lds, st>>>((T*)y, (const T*)x, (const T*)w, p);
  return 0;
}

// returns -2 when the geometry does not fit this instantiation's register / LDS budget
template <typename T, typename TY, int TO, int RW, int NI, int NC, int F33 = 0>
int launch_pipe(void* y, const void* x, const void* w, DConv p, hipStream_t st) {
  constexpr int TH = 4 * RW;
  p.border = 0;
  if (F33 == 2 && NI <= 6 && p.border_ok) {   // see DConv::border
    p.border = 1;
    p.nx = 0;
  }
  constexpr int NW = (TO * 9 * 4 + 255) / 256;
  if (p.hper) {   // image pairs: two blocks of (hper - 1 + tap extent) halo rows (in_stride 1, host-checked)
    if (TH != 2 * p.hper) return -2;
    p.hrows = p.hper - 1 + p.rows;
    p.rows = 2 * p.hrows;
  } else {
    p.rows = (TH - 1) * p.in_stride + p.rows;
  }
  p.cols = (DTW - 1) * p.in_stride + p.cols;
  p.inv_cols = 1.0f / (float)p.cols;
  const int n_in = p.rows * p.cols * 4, n_w = TO * p.ntaps * 4;
  if (n_in > NI * 256 || n_w > NW * 256 || p.rows >= 32768 || p.cols >= 65536) return -2;
  if (F33 && (p.cols != (F33 == 3 ? 2 * (DTW - 1) + 3 : DTW + 2) || p.O % TO || (F33 != 2) != !p.hzero)) return -2;
  if ((F33 == 3) != (F33 && p.in_stride == 2)) return -2;
  size_t lds = sizeof(uint4) * 4 * 64 * (size_t)(NI + NW);   // four planes of NI*64 pixels and NW*64 weight rows
  static const bool no_wres = getenv("DGV2_NO_WRES") != nullptr;   // A/B switch for benchmarking
  p.wres = 0;
  if (!F33 && !no_wres && p.Cin == 8 * (int)(16 / sizeof(T)) && lds + sizeof(uint4) * 4 * 64 * (size_t)NW <= 64 * 1024) {
    p.wres = 1;                      // two K-chunks, at least two blocks per CU
    lds += sizeof(uint4) * 4 * 64 * (size_t)NW;
  }
  if (lds > 80 * 1024) return -2;   // two blocks per CU
  auto kern = conv_pipe_kernel<T, TY, TO, RW, NI, NC, F33>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int tiles_w = (p.Wg + DTW - 1) / DTW, tiles_h = (p.Hg + TH - 1) / TH, tiles_o = (p.O + TO - 1) / TO;
  // walk several tiles per block while the grid still holds >= ~4 blocks per CU slot
  const int64_t tiles = (int64_t)tiles_w * tiles_h * p.B * tiles_o;
  // (four-class launches: two blocks per CU and a heavy per-block prologue -> 1024 blocks = two full rounds; measured at
  // D block 0's stride-2 data gradient: 1 / 2 / 4 / 8 tiles per block = 171 / 151 / 134 / 129 us)
  int tpb = (int)(tiles / (NC == 4 ? 1024 : 2048));
  tpb = tpb < 1 ? 1 : (tpb > tiles_w ? tiles_w : tpb);
  tpb = tpb > 8 ? 8 : tpb;
  static const int tpb_env = getenv("DGV2_CONV_TPB") ? atoi(getenv("DGV2_CONV_TPB")) : 0;   // experiments
  if (tpb_env > 0) tpb = tpb_env > tiles_w ? tiles_w : tpb_env;
  p.tpb = tpb;
  p.nt = (!p.resid && !p.accumulate && nt_output((int64_t)p.B * p.Hy * p.Wy * p.ldy * sizeof(TY))) ? 1 : 0;
  dim3 grid((tiles_w + tpb - 1) / tpb, tiles_h * p.B, tiles_o);
  static const bool no_xcd = getenv("DGV2_CONV_NO_XCD") != nullptr;   // A/B switch for benchmarking
  p.xcd = 0;
  if (!no_xcd && tiles_o >= 2 && sizeof(T) < 4) {   // (the MFMA-bound fp32 instances gain nothing: 673 vs 681 us)
    p.xcd = 1;
    p.gx = grid.x; p.gy = grid.y; p.gz = grid.z;
    const int64_t npt = ((int64_t)grid.x * grid.y + 7) / 8 * 8;
    grid = dim3((unsigned)(npt * grid.z), 1, 1);
  }
  kern<<<grid, 256, lds, st>>>((TY*)y, (const T*)x, (const T*)w, p);
  return 0;
}

}  // namespace

// conv_strip.hip: streaming kernel for 3x3 stride-1 ring convs with 32 -> 32 channels (bf16); -2 = not covered
int dgv2_conv_strip_try(void* y, const void* x, const void* w, int B, int H, int W, int Cin, int O, int ntaps, int wtaps,
                        const int* taps4, int nextra, const int* extras5, int hzero, const float* bias,
                        const void* resid, int act, float alpha, float scale, hipStream_t st);

// conv8.hip: eight-wave form of the unrolled 3x3 forward conv (two groups of four waves sharing the halo tile or the
// weight slab); -2 = not covered
int dgv2_conv8_try(void* y, int ldy, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg, int Wg, int O,
                   int in_stride, int wtaps, int widx0, int wstep, const float* bias, const float* acc_scale,
                   const void* resid, int act, float alpha, float scale, int dtype, hipStream_t st, int wimg, int hzero);

namespace {

template <typename T, int TO, typename TY = T>
int dispatch_pipe(void* y, const void* x, const void* w, const DConv& p, bool f33, hipStream_t st) {
  if (p.ncls == 4) {
    if constexpr (TO <= 32) return p.in_stride == 1 ? launch_pipe<T, TY, TO, 1, 4, 4>(y, x, w, p, st) : -2;
    else return -2;
  }
  if (p.ncls != 1) return -2;
  if constexpr (TO == 64) {   // the full 3x3 grid with >= 64 output channels: unrolled taps
    if (f33 && p.O % TO == 0 && p.in_stride == 2) {   // stride-2 forward (the conv behind a blur): 4 x 32 tiles
      if (!p.hzero && !p.hper) return launch_pipe<T, TY, TO, 1, 10, 1, 3>(y, x, w, p, st);
    } else if (f33 && p.O % TO == 0) {
      if (p.hzero) {
        if (p.hper) return launch_pipe<T, TY, TO, 2, 7, 1, 2>(y, x, w, p, st);
        if (p.Hg >= 8) return launch_pipe<T, TY, TO, 2, 6, 1, 2>(y, x, w, p, st);
        return launch_pipe<T, TY, TO, 1, 4, 1, 2>(y, x, w, p, st);
      }
      if (p.hper) return launch_pipe<T, TY, TO, 2, 7, 1, 1>(y, x, w, p, st);
      if (p.Hg >= 8) return launch_pipe<T, TY, TO, 2, 6, 1, 1>(y, x, w, p, st);
      return launch_pipe<T, TY, TO, 1, 4, 1, 1>(y, x, w, p, st);
    }
  }
  if (p.hper) return launch_pipe<T, TY, TO, 2, 7, 1>(y, x, w, p, st);
  if (p.in_stride == 1 && p.Hg >= 8) return launch_pipe<T, TY, TO, 2, 6, 1>(y, x, w, p, st);
  if (p.in_stride == 1) return launch_pipe<T, TY, TO, 1, 4, 1>(y, x, w, p, st);
  return launch_pipe<T, TY, TO, 1, 10, 1>(y, x, w, p, st);
}

}  // namespace

extern "C" int dgv2_conv_taps_ld(void* y, int ldy, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg,
                                 int Wg, int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w,
                                 int out_stride, int ncls, const int* cls_host, int ntaps, int wtaps,
                                 const int* taps_host, int nextra, const int* extras_host, int hzero, int ring,
                                 int accumulate, const float* bias, const void* resid, int act, float alpha,
                                 float scale, int dtype, void* stream);

// y[b, gh*out_stride+ooff_h(c), gw*out_stride+ooff_w(c), o] (=|+=) act( sum_{t in class c} sum_ch
//     x[b, H(gh*in_stride+ioff_h+dy_t), W(gw*in_stride+ioff_w+dx_t), ch] * w[o, widx_t, ch] + resid + bias[o] )
// for gh < Hg, gw < Wg and every output class c.  taps_host: HOST pointer to ntaps quadruples
// (dy, dx, widx, cls), sorted by class, ntaps <= 9; cls_host: ncls pairs (ooff_h, ooff_w), ncls in {1, 4};
// extras_host: nextra <= 6 quintuples (dy, dx, widx, cls, row): an additional tap for output row gh == row only,
// widx must be one of the main taps' (its weights are already staged).
// H(): clamp (hzero = 0) or zero outside [0,Hin) (hzero = 1); W(): wrap (ring) or clamp.
// Cin must be a multiple of 32 (bf16) / 16 (fp32).  Classes / extras need the pipelined kernel; geometries it
// does not cover return DGV2_ENOTSUP (callers then issue one launch per class / border row).
extern "C" int dgv2_conv_taps_ex(void* y, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg,
                                 int Wg, int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w,
                                 int out_stride, int ncls, const int* cls_host, int ntaps, int wtaps,
                                 const int* taps_host, int nextra, const int* extras_host, int hzero, int ring,
                                 int accumulate, const float* bias, const void* resid, int act, float alpha,
                                 float scale, int dtype, void* stream) {
  return dgv2_conv_taps_ld(y, O, x, w, B, Hin, Win, Cin, Hg, Wg, O, Hy, Wy, in_stride, ioff_h, ioff_w, out_stride, ncls,
                           cls_host, ntaps, wtaps, taps_host, nextra, extras_host, hzero, ring, accumulate, bias, resid, act,
                           alpha, scale, dtype, stream);
}

// dgv2_conv_taps_ex writing the O output channels into rows of ldy >= O channels (y / resid point at the first of them):
// a launch may then produce a channel RANGE of a wider tensor -- the 528-channel data gradient of the discriminator's
// epilogue conv runs as 512 + 16 channels instead of nine 64-channel slabs of which the last is three quarters empty.
static int conv_taps_impl(void* y, int ldy, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg,
                          int Wg, int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w,
                          int out_stride, int ncls, const int* cls_host, int ntaps, int wtaps,
                          const int* taps_host, int nextra, const int* extras_host, int hzero, int ring,
                          int accumulate, const float* bias, const void* resid, int act, float alpha,
                          float scale, int dtype, const float* acc_scale, void* stream) {
  if (!y || !x || !w || !taps_host || !cls_host || ntaps < 1 || ntaps > 9 || wtaps < 1 || ldy < O) return DGV2_EINVAL;
  if (ldy != O && (ldy % (dtype == DGV2_F32 ? 4 : 8))) return DGV2_EINVAL;   // 16-byte stores stay aligned
  if (ncls != 1 && ncls != 4) return DGV2_EINVAL;
  if (nextra < 0 || nextra > 6 || (nextra > 0 && !extras_host)) return DGV2_EINVAL;
  if (B <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Hg <= 0 || Wg <= 0 || O <= 0 || in_stride < 1 || out_stride < 1)
    return DGV2_EINVAL;
  if (act != 0 && act != 3) return DGV2_EINVAL;
  const int kstep = dtype == DGV2_FP8 ? 64 : (dtype == DGV2_BF16 ? 32 : 16);   // one 64-byte K-chunk
  if (Cin % kstep || !aligned16(x) || !aligned16(w)) return DGV2_EINVAL;
  DConv p;
  p.acc_scale = acc_scale;
  p.nt = 0;
  p.hper = 0; p.hrows = 0;
  p.B = B; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Hg = Hg; p.Wg = Wg; p.O = O; p.Hy = Hy; p.Wy = Wy; p.ldy = ldy;
  p.in_stride = in_stride; p.ioff_h = ioff_h; p.ioff_w = ioff_w;
  p.out_stride = out_stride; p.ooff_h = cls_host[0]; p.ooff_w = cls_host[1];
  p.ntaps = ntaps; p.wtaps = wtaps;
  p.ncls = ncls;
  for (int c = 0; c < 4; ++c) {
    p.cls_ooh[c] = c < ncls ? cls_host[2 * c] : 0;
    p.cls_oow[c] = c < ncls ? cls_host[2 * c + 1] : 0;
  }
  // single-class launches: taps in (dy, dx) order (the sum does not depend on it; the full 3x3 grid then is tap
  // t = (dy - dymin) * 3 + (dx - dxmin), which the unrolled kernel variant relies on)
  int sorted[36];
  if (ncls == 1) {
    for (int t = 0; t < 4 * ntaps; ++t) sorted[t] = taps_host[t];
    for (int i = 1; i < ntaps; ++i)
      for (int j = i; j > 0 && (sorted[4 * j] < sorted[4 * j - 4] ||
                                (sorted[4 * j] == sorted[4 * j - 4] && sorted[4 * j + 1] < sorted[4 * j - 3])); --j)
        for (int k = 0; k < 4; ++k) { const int v = sorted[4 * j + k]; sorted[4 * j + k] = sorted[4 * j - 4 + k]; sorted[4 * j - 4 + k] = v; }
    taps_host = sorted;
  }
  int dymin = 1 << 30, dymax = -(1 << 30), dxmin = 1 << 30, dxmax = -(1 << 30);
  int prev_cls = 0;
  for (int c = 0; c < 5; ++c) p.cls_t0[c] = ntaps;
  p.cls_t0[0] = 0;
  for (int t = 0; t < 9; ++t) {
    p.dy[t] = p.dx[t] = p.widx[t] = 0;
    if (t < ntaps) {
      p.dy[t] = taps_host[4 * t]; p.dx[t] = taps_host[4 * t + 1]; p.widx[t] = taps_host[4 * t + 2];
      const int c = taps_host[4 * t + 3];
      if (p.widx[t] < 0 || p.widx[t] >= wtaps || c < prev_cls || c >= ncls) return DGV2_EINVAL;
      for (int k = prev_cls + 1; k <= c; ++k) p.cls_t0[k] = t;
      prev_cls = c;
      dymin = p.dy[t] < dymin ? p.dy[t] : dymin; dymax = p.dy[t] > dymax ? p.dy[t] : dymax;
      dxmin = p.dx[t] < dxmin ? p.dx[t] : dxmin; dxmax = p.dx[t] > dxmax ? p.dx[t] : dxmax;
    }
  }
  p.nx = nextra;
  for (int e = 0; e < 6; ++e) {
    p.x_dy[e] = p.x_dx[e] = p.x_slot[e] = p.x_cls[e] = 0;
    p.x_row[e] = -1;
    if (e < nextra) {
      const int* q = extras_host + 5 * e;
      p.x_dy[e] = q[0]; p.x_dx[e] = q[1]; p.x_cls[e] = q[3]; p.x_row[e] = q[4];
      int slot = -1;
      for (int t = 0; t < ntaps; ++t)
        if (p.widx[t] == q[2]) slot = t;
      if (slot < 0 || q[3] < 0 || q[3] >= ncls) return DGV2_EINVAL;
      p.x_slot[e] = slot;
      dymin = q[0] < dymin ? q[0] : dymin; dymax = q[0] > dymax ? q[0] : dymax;
      dxmin = q[1] < dxmin ? q[1] : dxmin; dxmax = q[1] > dxmax ? q[1] : dxmax;
    }
  }
  p.dymin = dymin; p.dxmin = dxmin;
  p.rows = dymax - dymin + 1;   // tap extents; the launchers add the tile extent
  p.cols = dxmax - dxmin + 1;
  p.hzero = hzero; p.ring = ring; p.accumulate = accumulate;
  p.tpb = 1; p.inv_cols = 0.f; p.wres = 0; p.s2d = 0; p.xcd = 0; p.gx = p.gy = p.gz = 1;
#ifdef DGV2_ABLATE
  p.ablate = getenv("DGV2_CP_ABLATE") ? atoi(getenv("DGV2_CP_ABLATE")) : 0;
#endif
  p.bias = bias; p.resid = resid; p.ybase = y; p.act = act; p.alpha = alpha; p.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  if (dtype == DGV2_BF16 && ldy == O && ncls == 1 && in_stride == 1 && out_stride == 1 && ring && !accumulate && Hin == Hg &&
      Win == Wg && Hy == Hg && Wy == Wg && ioff_h == 0 && ioff_w == 0 && cls_host[0] == 0 && cls_host[1] == 0 &&
      aligned16(y) && (!resid || aligned16(resid))) {
    // full-resolution 32 -> 32 channel layers: the strip-streaming kernel (conv_strip.hip)
    rc = dgv2_conv_strip_try(y, x, w, B, Hin, Win, Cin, O, ntaps, wtaps, taps_host, nextra, extras_host, hzero, bias,
                             resid, act, alpha, scale, st);
    if (rc != -2) {
      if (rc) return rc;
      DGV2_RETURN_LAST();
    }
    rc = 0;
  }
  // 4-row maps (the discriminator's last block and its epilogue): an 8-row tile over a PAIR of images stages the weight
  // slab once for both (fp32 epilogue conv forward 728 -> 632 us, bf16 121 -> 93 us at 128 x 4 x 32)
  static const bool no_pairs = getenv("DGV2_NO_PAIRS") != nullptr;   // A/B switch for benchmarking
  if (!no_pairs && ncls == 1 && in_stride == 1 && out_stride == 1 && Hin == 4 && Hg == 4 && Hy == 4 && ioff_h == 0 &&
      cls_host[0] == 0 && (B & 1) == 0 && O >= 64 &&
      (int64_t)(B / 2) * ((O + 63) / 64) * ((Wg + DTW - 1) / DTW) >= 512) {   // still two blocks per CU: fewer were slower
    p.hper = 4;
    p.B = B / 2; p.Hin = 8; p.Hg = 8; p.Hy = 8;
  }
  static const bool no_pipe = getenv("DGV2_NO_PIPE") != nullptr;   // A/B switch for benchmarking
  // the ring wrap of the pipelined kernel assumes -Win <= gw < 4*Win
  const bool wrap_ok = !ring || (ioff_w + dxmin >= -Win && in_stride * ((Wg + DTW - 1) / DTW * DTW) + ioff_w + dxmax < 4 * Win);
  const bool plain = ncls == 1 && nextra == 0;
  static const bool no_f33 = getenv("DGV2_NO_F33") != nullptr;   // A/B switch for benchmarking
  bool f33 = !no_f33 && ncls == 1 && (in_stride == 1 || (in_stride == 2 && out_stride == 1 && !hzero)) && ntaps == 9 &&
             p.rows == 3 && p.cols == 3;
  for (int t = 0; f33 && t < 9; ++t) f33 = p.dy[t] == dymin + t / 3 && p.dx[t] == dxmin + t % 3;
  for (int t = 2; f33 && t < 9; ++t) f33 = p.widx[t] - p.widx[t - 1] == p.widx[1] - p.widx[0];   // weight slots affine in t
  for (int e = 0; f33 && hzero && e < nextra; ++e)   // zero rows are not staged as zeros there: extras must read real rows
    f33 = (unsigned)(p.x_row[e] * in_stride + ioff_h + p.x_dy[e]) < (unsigned)Hin;
  {
    // border_ok: the extras are the replicate-row terms of the same-size stride-1 3x3 data gradient -- at output row 0
    // the taps (0, dx) with the weights of the taps (+1, dx), at the last row with those of (-1, dx)
    static const bool no_border = getenv("DGV2_NO_BORDER_REWEIGHT") != nullptr;   // A/B switch for benchmarking
    const int hrows_img = p.hper ? p.hper : p.Hg;
    bool ok = !no_border && f33 && hzero && nextra == 6 && in_stride == 1 && out_stride == 1 && ioff_h == 0 && dymin == -1 &&
              hrows_img >= 2 && (p.hper ? true : Hin == Hg);
    unsigned seen = 0;
    for (int e = 0; ok && e < 6; ++e) {
      const int top = p.x_row[e] == 0, bot = p.x_row[e] == hrows_img - 1, j = p.x_dx[e] - dxmin;
      ok = (top || bot) && p.x_dy[e] == 0 && p.x_cls[e] == 0 && j >= 0 && j < 3 && p.x_slot[e] == (top ? 6 + j : j);
      seen |= 1u << ((top ? 0 : 3) + (j & 3));
    }
    p.border_ok = (ok && seen == 63u) ? 1 : 0;
  }
  if ((dtype == DGV2_BF16 || dtype == DGV2_FP8) && f33 && ncls == 1 && nextra == 0 && !accumulate && !hzero && ring &&
      out_stride == 1 && ioff_h == 0 && ioff_w == 0 && cls_host[0] == 0 && cls_host[1] == 0 && dymin == -1 && dxmin == -1 &&
      Hy == Hg && Wy == Wg) {
    // forward 3x3 convs from 64 channels up: the eight-wave engine (conv8.hip) where it covers the geometry
    rc = dgv2_conv8_try(y, ldy, x, w, B, Hin, Win, Cin, Hg, Wg, O, in_stride, wtaps, p.widx[0], p.widx[1] - p.widx[0], bias,
                        acc_scale, resid, act, alpha, scale, dtype, st, 0, 0);
    if (rc != -2) {
      if (rc) return rc;
      DGV2_RETURN_LAST();
    }
    rc = 0;
  }
  if (dtype == DGV2_FP8) {
    // e4m3 operands, bf16 result / residual: the forward convs behind the FIRs (>= 64 output channels, one class)
    if (ncls != 1 || nextra || accumulate || O < 64 || !wrap_ok) return DGV2_ENOTSUP;
    rc = dispatch_pipe<fp8_t, 64, bf16_t>(y, x, w, p, f33, st);
    if (rc == -2) return DGV2_ENOTSUP;
    if (rc) return rc;
    DGV2_RETURN_LAST();
  }
  static const bool no_s2d = getenv("DGV2_NO_S2D") != nullptr;   // A/B switch for benchmarking
  if (!no_s2d && ncls == 4 && ntaps == 9 && in_stride == 1 && out_stride == 2 && hzero && dymin == 0 && dxmin == 0 &&
      p.rows == 2 && p.cols == 2) {
    static const int dy9[9] = {0, 0, 0, 1, 0, 1, 1, 0, 0}, dx9[9] = {0, 1, 0, 0, 0, 1, 0, 1, 0}, cl9[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
    bool ok = true;
    for (int t = 0; t < 9; ++t) ok = ok && p.dy[t] == dy9[t] && p.dx[t] == dx9[t] && taps_host[4 * t + 3] == cl9[t];
    for (int c = 0; c < 4; ++c) ok = ok && p.cls_ooh[c] == (c >> 1) && p.cls_oow[c] == (c & 1);
    p.s2d = ok ? 1 : 0;
  }
  DGV2_DISPATCH_DTYPE(dtype, {
    rc = -2;
    if ((!no_pipe || !plain) && wrap_ok) {
      // four output classes quadruple the accumulators: 32-channel tiles keep them in registers
      if (O <= 16) rc = dispatch_pipe<T, 16>(y, x, w, p, f33, st);
      else if (O <= 32 || ncls == 4) rc = dispatch_pipe<T, 32>(y, x, w, p, f33, st);
      else rc = dispatch_pipe<T, 64>(y, x, w, p, f33, st);
    }
    if (rc == -2) {
      if (!plain) return DGV2_ENOTSUP;
      if (O <= 16) rc = launch_direct<T, 16>(y, x, w, p, st);
      else if (O <= 32) rc = launch_direct<T, 32>(y, x, w, p, st);
      else rc = launch_direct<T, 64>(y, x, w, p, st);
    }
  });
  if (rc) return rc;
  DGV2_RETURN_LAST();
}

extern "C" int dgv2_conv_taps_ld(void* y, int ldy, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg,
                                 int Wg, int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w,
                                 int out_stride, int ncls, const int* cls_host, int ntaps, int wtaps,
                                 const int* taps_host, int nextra, const int* extras_host, int hzero, int ring,
                                 int accumulate, const float* bias, const void* resid, int act, float alpha,
                                 float scale, int dtype, void* stream) {
  if (dtype != DGV2_F32 && dtype != DGV2_BF16) return DGV2_EINVAL;
  return conv_taps_impl(y, ldy, x, w, B, Hin, Win, Cin, Hg, Wg, O, Hy, Wy, in_stride, ioff_h, ioff_w, out_stride, ncls,
                        cls_host, ntaps, wtaps, taps_host, nextra, extras_host, hzero, ring, accumulate, bias, resid, act,
                        alpha, scale, dtype, nullptr, stream);
}

// dgv2_conv_taps (single class, no extras, overwrite) on e4m3 operands: x8 [B,Hin,Win,Cin] and w8 [O,wtaps,Cin] are OCP
// e4m3 bytes (x at unit scale, w scaled by a per-tensor power of two: dgv2_fp8_quant_weights), contracted by
// v_mfma_f32_16x16x32_fp8_fp8 with fp32 accumulation;
//   y (bf16) = act( acc * acc_scale[0] + resid + bias ) * scale,   acc_scale: DEVICE scalar (EqualLR factor / weight scale).
// Cin % 64 == 0 (one 64-byte K-chunk = 64 channels), O >= 64; DGV2_ENOTSUP otherwise (callers then run the bf16 conv).
extern "C" int dgv2_conv_taps_fp8(void* y, const void* x8, const void* w8, const float* acc_scale, int B, int Hin,
                                  int Win, int Cin, int Hg, int Wg, int O, int Hy, int Wy, int in_stride, int ioff_h,
                                  int ioff_w, int ntaps, int wtaps, const int* taps_host, int ring, const float* bias,
                                  const void* resid, int act, float alpha, float scale, void* stream) {
  if (!taps_host || ntaps < 1 || ntaps > 9 || !acc_scale) return DGV2_EINVAL;
  if (Cin % 64) return DGV2_ENOTSUP;
  int taps4[36];
  for (int t = 0; t < ntaps; ++t) {
    taps4[4 * t] = taps_host[3 * t]; taps4[4 * t + 1] = taps_host[3 * t + 1]; taps4[4 * t + 2] = taps_host[3 * t + 2];
    taps4[4 * t + 3] = 0;
  }
  const int cls[2] = {0, 0};
  return conv_taps_impl(y, O, x8, w8, B, Hin, Win, Cin, Hg, Wg, O, Hy, Wy, in_stride, ioff_h, ioff_w, 1, 1, cls, ntaps,
                        wtaps, taps4, 0, nullptr, 0, ring, 0, bias, resid, act, alpha, scale, DGV2_FP8, acc_scale, stream);
}

// The single-class, no-extras form (taps_host: ntaps triples (dy, dx, widx)).
extern "C" int dgv2_conv_taps(void* y, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg, int Wg,
                              int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w, int out_stride,
                              int ooff_h, int ooff_w, int ntaps, int wtaps, const int* taps_host, int hzero, int ring,
                              int accumulate, const float* bias, const void* resid, int act, float alpha,
                              float scale, int dtype, void* stream) {
  if (!taps_host || ntaps < 1 || ntaps > 9) return DGV2_EINVAL;
  int taps4[36];
  for (int t = 0; t < ntaps; ++t) {
    taps4[4 * t] = taps_host[3 * t]; taps4[4 * t + 1] = taps_host[3 * t + 1]; taps4[4 * t + 2] = taps_host[3 * t + 2];
    taps4[4 * t + 3] = 0;
  }
  const int cls[2] = {ooff_h, ooff_w};
  return dgv2_conv_taps_ex(y, x, w, B, Hin, Win, Cin, Hg, Wg, O, Hy, Wy, in_stride, ioff_h, ioff_w, out_stride, 1, cls,
                           ntaps, wtaps, taps4, 0, nullptr, hzero, ring, accumulate, bias, resid, act, alpha, scale,
                           dtype, stream);
}
