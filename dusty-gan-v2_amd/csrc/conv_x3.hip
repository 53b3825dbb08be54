// fp32 3x3 ring convolutions on the bf16 matrix cores: the discriminator's fp32 epilogue conv
//   reference: ops.Conv2d(ch(4) + 1, ch(4), 3, 1, 1, ring) of Discriminator.epilogue (gans/models/dusty_v2.py:376-378) under
//   the fp32 island of Discriminator.forward (:394-395: h.to(torch.float32) in front of the epilogue).
//
// The island ran on v_mfma_f32_16x16x4_f32 (conv_pipe_kernel<float>, exact fp32 fma chains) at 0.73-0.78 of that
// instruction's 157 TFLOP/s -- and was 17 % of the train step's GPU time (five launches of 77 / 39 GFLOP per iteration).
// gfx950 runs bf16 MFMA at 16x the fp32 MFMA rate, so here every fp32 operand value is split into three bf16 planes
//     x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)      (|x - h - m - l| <= 2^-26 |x|)
// and a product a * b is taken as the six bf16 products
//     a_h b_l + a_h b_m + a_h b_h + a_m b_m + a_m b_h + a_l b_h
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- gemm_x3.hip's scheme (the dropped terms are below 2^-25 of the
// product, less than the rounding of an fp32 multiply-add chain itself; tests hold the result to the fp32 kernel's own
// distance from float64).  Six bf16 products cost 3/8 of one fp32 product.
//
// Structure: conv8.hip's (eight waves = two groups of four, operands in four 16-byte K planes, nine taps straight-line
// with immediate LDS offsets, in-place asm MFMAs), with the planes changing what is worth sharing:
//   * the WEIGHTS arrive pre-split from the weight bank (dgv2_conv_weight_bank_ex with dtype fp32: three staging images,
//     one per plane, each a contiguous 36 KB run per (64-channel slab, 32-channel chunk));
//   * the ACTIVATIONS are split while their halo tile is staged (two 16-byte loads -> three 16-byte LDS units);
//   * a block is two 4 x 32 pixel tiles on ONE 64-channel slab; per 32-channel chunk it runs three stages, one per
//     weight plane: stage 0 contracts w_h with x_l, x_m, x_h, stage 1 w_m with x_m, x_h, stage 2 w_l with x_h -- a
//     fragment read serves up to three MFMA groups, 2 MFMAs per ds_read_b128 against conv8's 1.3;
//   * the weight plane of stage s + 1 is written into the other of two LDS buffers at the start of stage s (its global
//     loads were issued one stage earlier and land under a whole stage of MFMAs): one barrier per stage.
// LDS: 2 tiles x 3 planes x 13 KB + 2 x 36 KB = 150 KB, one block per CU.
// The data gradient is the same kernel on the transposed images with zero rows and the replicate-row border terms
// (conv8.hip's HZ form); its input channels past the last whole 64-channel slab (the one minibatch-stddev channel of
// 512 + 1) come from a small exact-fp32 kernel.
#include <stdlib.h>

#include "gemm_core.h"

namespace {

struct CX3 {
  int B, H, W, Cx;        // x [B, H, W, Cx] fp32 (Cx % 8 == 0)
  int nchunks;            // 32-channel chunks of the weight images (= ceil(Cx / 32); channels >= Cx are staged as zeros)
  int O, ldy;             // output channels written (a multiple of 64) and the row pitch of y
  int xexact;             // the caller's promise: channels [0, xexact) of x hold bf16-representable values (activations of a
                          // bf16 trunk widened to fp32) -- their m and l planes are zero: not staged, their products not issued
  int tiles_h, tiles_w, ntiles, npairs, nslab, per_xcd, total;
  int* status;            // the caller's device status word (non-NULL whenever xexact > 0)
  const float* bias;
  const float* resid;
  int act;
  float alpha, scale;
#ifdef DGV2_ABLATE   // benchmarking builds only (make ABLATE=1): wrong results by design, never in the shipped library
  int ablate;        // DGV2_X3_ABLATE: 2 no MFMA loop, 4 no input loads / split, 8 no weight loads / writes, 64 no barriers inside the chunk loop
#define X3_ABL (p.ablate)
#else
#define X3_ABL 0
#endif
};

constexpr int X_ICOLS = 34, X_NPIX = 6 * X_ICOLS, X_PIN = 208, X_PW = 9 * 64, X_NI = 4, X_NW = 5;
constexpr int X_TILE = 3 * 4 * X_PIN;   // 16-byte units of one pixel tile: three split planes x four K planes
constexpr int X_WBUF = 4 * X_PW;
constexpr size_t X_LDS = sizeof(uint4) * (2 * X_TILE + 2 * X_WBUF + 512);   // + one dummy row per thread (w_store)
static_assert(X_LDS <= 160 * 1024 - 1024, "LDS image");

// eight fp32 values -> their three bf16 planes (round to nearest even each time; the residuals are exact in fp32)
__device__ __forceinline__ void split3x8(const float4& lo, const float4& hi, uint4& h, uint4& m, uint4& l) {
  const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  union { uint4 u; bf16_t e[8]; } ph, pm, pl;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
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

// the promise CX3::xexact is checked where it is used: a value with a residual raises bit 0 of the caller's status word
// (CX3::status, dgv2.h "status words"); the library keeps no flag of its own

// eight fp32 values that are promised to be bf16-representable -> plane h; true if one of them is not
__device__ __forceinline__ bool exact1x8(const float4& lo, const float4& hi, uint4& h) {
  const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  union { uint4 u; bf16_t e[8]; } ph;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    ph.e[i] = (bf16_t)f[i];
    bad |= (f[i] - (float)ph.e[i]) != 0.f;
  }
  h = ph.u;
  return bad;
}

// A group of four in-place bf16 MFMAs (one B fragment against four A fragments) as ONE asm statement: the compiler can place
// nothing between them, and inside the group the next MFMA's issue covers the previous one's operand reads.  NOP: eight
// wait states behind the last MFMA, so that nothing the compiler generates behind the group (branch conditions
// materialised through VGPRs, register shuffles at high pressure, the staging work) can write an operand register the
// MFMA is still reading (five wait states) or touch its result before it is written (eight) -- the hazard recogniser does
// not look inside asm (scripts/audit_asm_mfma.py).  At 256 registers the allocator split an
// accumulator's live range around the dead-tap branches and copied it with v_mov_b64 right behind a group that carried five
// wait states only: the copy took the upper half of the last fragment before the MFMA had written it (wrong values in one
// output row of the data gradient; found by tests/test_gpu_ops.py, then by the extended audit).
// NOP = false exists for experiments only (8 cycles per group = 5 % of the forward; unsafe, see x3_stage).
template <bool NOP>
__device__ __forceinline__ void mfma4_bf16(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, const uint4& a0, const uint4& a1,
                                           const uint4& a2, const uint4& a3, const uint4& b) {
  union U { uint4 u; bf16x8 v; };
  U ua0, ua1, ua2, ua3, ub;
  ua0.u = a0; ua1.u = a1; ua2.u = a2; ua3.u = a3; ub.u = b;
  if constexpr (NOP) {
    asm volatile(
        "v_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %2, %6, %8, %2\n\t"
        "v_mfma_f32_16x16x32_bf16 %3, %7, %8, %3\n\t"
        "s_nop 7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(ua0.v), "v"(ua1.v), "v"(ua2.v), "v"(ua3.v), "v"(ub.v));
  } else {
    asm volatile(
        "v_mfma_f32_16x16x32_bf16 %0, %4, %8, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %5, %8, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %2, %6, %8, %2\n\t"
        "v_mfma_f32_16x16x32_bf16 %3, %7, %8, %3"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(ua0.v), "v"(ua1.v), "v"(ua2.v), "v"(ua3.v), "v"(ub.v));
  }
}

// One stage: the nine taps of one weight plane (LDS buffer a_base, this lane's K plane and row) against the NX first
// split planes of the wave's pixel row (xb: split 0 of this lane's K plane; split q is q * 4 * X_PIN further), smallest
// plane first.  HZ: conv8.hip's data-gradient form -- `dead` = taps that read only zero rows for this wave's output row.
// `side(t)` runs once behind the MFMA groups of tap t: the staging work of the coming stages, issued under this stage's MFMAs.
#define X3_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef DGV2_ABLATE
__device__ int x3_abl_noreads;   // ablation builds: 1 = the in-loop fragment re-reads are skipped (timing of the pure MFMA issue)
#define X3_NOREADS (x3_abl_noreads != 0)
#else
#define X3_NOREADS false
#endif

template <int NX, int HZ, typename Side>
__device__ __forceinline__ void x3_stage(f32x4 (&acc)[4][2], const uint4* __restrict__ a_base, const uint4* __restrict__ xb,
                                         const int (&bpix)[2], unsigned dead, Side&& side) {
  constexpr int SLOTS = 2 * NX;
  const bool noreads = X3_NOREADS;
  uint4 a[2][4], bb[2][NX];
  unsigned dd = 0u;
  if constexpr (HZ != 0) {
    dd = dead;
    asm volatile("" : "+s"(dd));   // the bit tests stay in the loop (hoisted, their results live in SGPR pairs that spill)
  }
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) a[0][mf] = a_base[mf * 16];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int q = 0; q < NX; ++q) bb[nf][q] = xb[q * 4 * X_PIN + bpix[nf]];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
      for (int qi = 0; qi < NX; ++qi) {
        const int q = NX - 1 - qi;
        X3_SB();
        if (!(HZ && ((dd >> t) & 1u)))
        {
          // Wait states behind EVERY group.  Dropping them where only fragment reads and the next group follow measured 5 % on
          // the forward -- and the audit then found a v_mov_b64 of the allocator into a live B operand in the x_exact
          // instance: what the compiler puts between two groups is not under this file's control.
          mfma4_bf16<true>(acc[0][nf], acc[1][nf], acc[2][nf], acc[3][nf], a[t & 1][0], a[t & 1][1], a[t & 1][2], a[t & 1][3], bb[nf][q]);
        }
        if constexpr (HZ != 0) {
          // replicate-row border term (conv8.hip): the tap mirrored in dy is dead for this row -> the border row once
          // more (the pixel fragment of tap (dy = 0, dx)) through the weights in registers right now
          if (t < 3 || t >= 6) {
            const int tm = t < 3 ? t + 6 : t - 6;
            if ((dd >> tm) & 1u) {
              const uint4 bx = xb[q * 4 * X_PIN + bpix[nf] + X_ICOLS + t % 3];
              mfma4_bf16<true>(acc[0][nf], acc[1][nf], acc[2][nf], acc[3][nf], a[t & 1][0], a[t & 1][1], a[t & 1][2], a[t & 1][3], bx);
            }
          }
        }
        X3_SB();
        if (t + 1 < 9 && !noreads) {
          const int ky = (t + 1) / 3, kx = (t + 1) % 3;
          bb[nf][q] = xb[q * 4 * X_PIN + bpix[nf] + ky * X_ICOLS + kx];
          const int slot = nf * NX + qi;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k >= slot * 4 / SLOTS && k < (slot + 1) * 4 / SLOTS) a[(t + 1) & 1][k] = a_base[(t + 1) * 64 + k * 16];
        }
      }
    }
    X3_SB();
    side(t);
  }
  X3_SB();
}

// XE: the launch carries an x_exact promise (the exact-chunk code is compiled in); the data gradient and promise-free
// forwards run the instances without it -- 40 registers lighter
template <int HZ, bool XE>
__global__ __launch_bounds__(512, 2) void conv_x3_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                         const bf16_t* __restrict__ wimg, CX3 p) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  __shared__ __attribute__((aligned(16))) float s_bias[64];
  // blocks of one XCD (ids 8 apart) walk a contiguous range of (slab, tile pair) in slab-major order: the three weight
  // planes of a slab (1.9 MB at 544 channels) stay in that XCD's L2
  const int n = blockIdx.x, kk = n >> 3;
  const int item = (n & 7) * p.per_xcd + kk;
  if (kk >= p.per_xcd || item >= p.total) return;
  const int slab = item / p.npairs, pair = item - slab * p.npairs;

  const int tid = threadIdx.x;
  const int t256 = tid & 255, gn = tid >> 8;
  const int wave4 = (tid >> 6) & 3, lane = tid & 63;
  const int lr = lane & 15, lc = lane >> 4;
  uint4* const xs = smem + gn * X_TILE;
  uint4* const wb = smem + 2 * X_TILE;

  const int q_ = pair * 2 + gn;
  const bool tile_live = q_ < p.ntiles;
  const int qc = tile_live ? q_ : p.ntiles - 1;
  const int per_img = p.tiles_h * p.tiles_w;
  const int b = qc / per_img, rem = qc - b * per_img;
  const int h0 = (rem / p.tiles_w) * 4, w0 = (rem % p.tiles_w) * 32;
  const int o0 = slab * 64;
  const float* xb_g = x + (int64_t)b * p.H * p.W * p.Cx;

  // ---- staging slots ----
  const int in_plane = (t256 >> 3) & 3;
  int goff[X_NI], lrow[X_NI];
#pragma unroll
  for (int j = 0; j < X_NI; ++j) {
    const int u = t256 + 256 * j;
    const int pix = ((u >> 5) << 3) | (u & 7);
    const int pl = pix < X_NPIX ? pix : X_NPIX - 1;
    const int iy = pl / X_ICOLS, ix = pl - iy * X_ICOLS;
    int gh = h0 - 1 + iy;
    gh = gh < 0 ? 0 : (gh >= p.H ? p.H - 1 : gh);       // rows clamp (HZ: those rows are only read by dead taps)
    int gw = w0 - 1 + ix;
    gw = gw < 0 ? gw + p.W : (gw >= p.W ? gw - p.W : gw);
    goff[j] = (gh * p.W + gw) * p.Cx + in_plane * 8;
    lrow[j] = pix < X_NPIX ? pix : -1;
  }
  float4 rin[X_NI][2];
  bool rin_ok = true;
  auto issue_in = [&](int c0) {
    // whole 8-channel units (Cx % 8 == 0); a unit past Cx loads channel 0.. of its pixel instead and is staged as zeros
    // (a conditional load with a zero alternative crashes this compiler's machine copy propagation)
    rin_ok = c0 + in_plane * 8 < p.Cx;
    const int cc = rin_ok ? c0 : -in_plane * 8;
#pragma unroll
    for (int j = 0; j < X_NI; ++j) {
      const float4* src = reinterpret_cast<const float4*>(xb_g + cc + goff[j]);
      rin[j][0] = src[0];
      rin[j][1] = src[1];
    }
  };
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 rwt[X_NW];
  const size_t plane_units = (size_t)p.nslab * p.nchunks * (X_PW * 4);
  auto issue_w = [&](int st) {                               // stage st = chunk * 3 + plane
    const int c = st / 3, pl = st - c * 3;
    const u32x4* img = reinterpret_cast<const u32x4*>(wimg) + pl * plane_units + ((size_t)slab * p.nchunks + c) * (X_PW * 4) + tid;
#pragma unroll
    for (int j = 0; j < X_NW; ++j)
      if (j < X_NW - 1 || tid + 512 * j < X_PW * 4) rwt[j] = img[512 * j];
  };
  const int w_plane = (tid >> 3) & 3, w_row0 = ((tid >> 5) << 3) | (tid & 7);
  auto write_w = [&](int buf) {
    uint4* dst = wb + buf * X_WBUF + w_plane * X_PW + w_row0;
#pragma unroll
    for (int j = 0; j < X_NW; ++j)
      if (j < X_NW - 1 || tid + 512 * j < X_PW * 4) *reinterpret_cast<u32x4*>(dst + 128 * j) = rwt[j];
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int bpix[2];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) bpix[nf] = wave4 * X_ICOLS + nf * 16 + lr;
  if (tid < 64) s_bias[tid] = p.bias ? p.bias[o0 + tid] : 0.f;

  unsigned dead = 0u;
  if constexpr (HZ != 0) {
    const int orow = h0 + wave4;
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      if ((unsigned)(orow + t / 3 - 1) >= (unsigned)p.H) m |= 1u << t;
    dead = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
  }

  const int nst = p.nchunks * 3;
  // The weight plane of stage st + 1 (in registers since the stage before) goes into the OTHER buffer in the FIRST taps of
  // stage st, the registers are reloaded with stage st + 2's plane in its LAST taps.  Never a load in front of a store of
  // the same stage: the compiler waits with vmcnt(0) in front of every such store (it cannot count across the predicated
  // slots), i.e. for every load issued so far -- with store j, load j, store j + 1 interleaved each store waited out the
  // global load issued one tap before it, a round trip to L2 per slot (that, not the instruction count, was the 118 us of
  // "staging" of profiles/round4_mb_x3_ablate.txt).
  // ... and never a BRANCH around either: behind a conditional load the compiler's wait-count merge no longer knows how many
  // loads are younger than the one a store needs and falls back to vmcnt(0).  So: the slot of the plane that does not exist
  // (last slot, upper half of the block; the stages behind the last) is stored into a dummy LDS row and loaded from a valid
  // address; every thread issues every slot.
  uint4* const w_dummy = smem + 2 * X_TILE + 2 * X_WBUF + tid;
  auto w_store = [&](int st, int j) {
    if (X3_ABL & 8) return;
    const bool in = j < X_NW - 1 || tid + 512 * j < X_PW * 4;
    uint4* dst = wb + ((st + 1) & 1) * X_WBUF + w_plane * X_PW + w_row0 + 128 * j;
    dst = in ? dst : w_dummy;
    *reinterpret_cast<u32x4*>(dst) = rwt[j];
  };
  auto w_load = [&](int st, int j) {
    if (X3_ABL & 8) return;
    const bool in = j < X_NW - 1 || tid + 512 * j < X_PW * 4;
    const int s2 = min(st + 2, nst - 1);
    const int c = s2 / 3, pl = s2 - c * 3;
    const u32x4* img = reinterpret_cast<const u32x4*>(wimg) + pl * plane_units + ((size_t)slab * p.nchunks + c) * (X_PW * 4) + tid;
    rwt[j] = img[in ? 512 * j : 0];
  };
  // taps 0..4: stores; taps 5..8: the five loads (two in tap 5)
  auto w_side = [&](int st, int t) {
    if (t < X_NW) w_store(st, t);
    if (t == 5) { w_load(st, 0); w_load(st, 1); }
    if (t > 5) w_load(st, t - 4);
  };
  uint4 xh[X_NI];                            // plane h of the coming chunk's pixels between their split and the boundary
  // split slot j of the pixels in registers; planes m and l go to LDS at once, h waits in xh
  auto split_store = [&](int j, bool all, bool ex) {
    if (ex) {   // promised bf16-exact channels: plane h is the value, planes m and l are zero and never read for this chunk
      const bool bad = exact1x8(rin[j][0], rin[j][1], xh[j]);
      if (bad && rin_ok && lrow[j] >= 0) atomicOr(p.status, DGV2_STATUS_X_INEXACT);
      if (!rin_ok) xh[j] = make_uint4(0u, 0u, 0u, 0u);
      if (all && lrow[j] >= 0) xs[in_plane * X_PIN + lrow[j]] = xh[j];
      return;
    }
    uint4 m, l;
    split3x8(rin[j][0], rin[j][1], xh[j], m, l);
    if (!rin_ok) xh[j] = m = l = make_uint4(0u, 0u, 0u, 0u);
    if (lrow[j] >= 0) {
      uint4* d = xs + in_plane * X_PIN + lrow[j];
      if (all) d[0] = xh[j];
      d[4 * X_PIN] = m;
      d[8 * X_PIN] = l;
    }
  };

  issue_w(0);
  issue_in(0);
  write_w(0);
  if (nst > 1) issue_w(1);
#pragma unroll
  for (int j = 0; j < X_NI; ++j) split_store(j, true, XE && 32 <= p.xexact);
  __syncthreads();
  const uint4* const xlane = xs + lc * X_PIN;
  for (int c = 0; c < p.nchunks; ++c) {
    const bool more = c + 1 < p.nchunks && !(X3_ABL & 4);
    const int st = c * 3;
    // chunks of bf16-exact input channels (CX3::xexact): x = x_h, so each stage is its weight plane against x_h alone --
    // three products per multiply instead of six, the same sum (the other three are products with zero)
    const bool ex = XE && (c + 1) * 32 <= p.xexact, ex_next = XE && (min(c + 1, p.nchunks - 1) + 1) * 32 <= p.xexact;
    const uint4* const a0 = wb + (st & 1) * X_WBUF + lc * X_PW + lr;
    const uint4* const a1 = wb + ((st + 1) & 1) * X_WBUF + lc * X_PW + lr;
    // stage 0: w_h x (x_l, x_m, x_h); under it the weight stores / loads, and (behind the stores) the loads of the coming
    // chunk's pixels, which stage 2 splits
    auto side0 = [&](int t) {
      w_side(st, t);
      if (t == 5 && !(X3_ABL & 4)) issue_in(min(c + 1, p.nchunks - 1) * 32);   // (last chunk: its own pixels once more, unused)
    };
    if (X3_ABL & 2) { for (int t = 0; t < 9; ++t) side0(t); }
    else if (ex) x3_stage<1, HZ>(acc, a0, xlane, bpix, dead, side0);
    else x3_stage<3, HZ>(acc, a0, xlane, bpix, dead, side0);
    if (!(X3_ABL & 64)) __syncthreads();
    // stage 1: w_m x (x_m, x_h)
    auto side1 = [&](int t) { w_side(st + 1, t); };
    if (X3_ABL & 2) { for (int t = 0; t < 9; ++t) side1(t); }
    else if (ex) x3_stage<1, HZ>(acc, a1, xlane, bpix, dead, side1);
    else x3_stage<2, HZ>(acc, a1, xlane, bpix, dead, side1);
    if (!(X3_ABL & 64)) __syncthreads();
    // stage 2: w_l x x_h -- planes m and l of the pixel tiles are no longer read: the coming chunk's pixels (loaded in
    // stage 0) are split under its first taps, m and l stored at once
    auto side2 = [&](int t) {
      if (t < X_NI && !(X3_ABL & 4)) split_store(t, false, ex_next);
      w_side(st + 2, t);
    };
    if (X3_ABL & 2) { for (int t = 0; t < 9; ++t) side2(t); }
    else x3_stage<1, HZ>(acc, a0, xlane, bpix, dead, side2);
    if (!(X3_ABL & 64)) __syncthreads();                                         // chunk c read
    if (more) {
#pragma unroll
      for (int j = 0; j < X_NI; ++j)
        if (lrow[j] >= 0) xs[in_plane * X_PIN + lrow[j]] = xh[j];
      if (!(X3_ABL & 64)) __syncthreads();                                       // plane h of chunk c + 1 visible
    }
  }

  mfma_drain();
  const int gh = h0 + wave4;
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    const int gw = w0 + nf * 16 + lr;
    const bool live = tile_live && gh < p.H && gw < p.W;
    const int64_t row = (((int64_t)b * p.H + gh) * p.W + gw) * p.ldy;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
      const int o = o0 + mf * 16 + lc * 4;
      const float4 b4 = *reinterpret_cast<const float4*>(&s_bias[mf * 16 + lc * 4]);
      const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
      float f[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc[mf][nf][r] + bq[r];
        if (p.act == 3) t = fmaxf(t, t * p.alpha) * p.scale;   // leaky ReLU, 0 <= alpha <= 1
        f[r] = t;
      }
      if (live && o < p.O) {
        if (p.resid) {
          const float4 r4 = *reinterpret_cast<const float4*>(p.resid + row + o);
          f[0] += r4.x; f[1] += r4.y; f[2] += r4.z; f[3] += r4.w;
        }
        *reinterpret_cast<float4*>(y + row + o) = make_float4(f[0], f[1], f[2], f[3]);
      }
    }
  }
}

// The data gradient's channels past the last whole slab, exact fp32: gx[b, h, w, c] for c0 <= c < c0 + nc:
//   sum_{ky, kx, o} gy[b, Hz(h + 1 - ky), wrap(w + 1 - kx), o] * wt[c][ky * 3 + kx][o] + the replicate-row terms,
// and resid / zeros in c0 + nc <= c < cend.  A block is one image row, a wave a quarter of its pixels; per pixel the lanes
// split the O x 9 contraction (weights and the three gy rows come from L1 / L2: 18 KB + 3 rows per block).
__global__ __launch_bounds__(256) void x3_dgrad_tail_kernel(float* __restrict__ gx, const float* __restrict__ gy,
                                                            const float* __restrict__ wt, const float* __restrict__ resid,
                                                            int B, int H, int W, int ldx, int O, int c0, int nc, int cend) {
  const int row = blockIdx.x;                  // b * H + h
  const int b = row / H, h = row - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (W + 3) / 4;
  const float* gimg = gy + (int64_t)b * H * W * O;
  const bool border = h == 0 || h == H - 1;
  for (int ci = 0; ci < nc; ++ci) {
    const float* wc = wt + (size_t)(c0 + ci) * 9 * O;
    // eight pixels at a time: their loads are independent (one pixel after the other was a chain of L2 latencies)
    for (int w0 = wave * per; w0 < min(W, (wave + 1) * per); w0 += 8) {
      float s[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = 0.f;
      for (int ob = lane * 4; ob < O; ob += 256) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {          // gy row h + d - 1 through ky = 2 - d; rows outside the image are zero
          const int gh = h + d - 1;
          if (gh < 0 || gh >= H) continue;
          const float* grow = gimg + (int64_t)gh * W * O + ob;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float4 k4 = *reinterpret_cast<const float4*>(wc + ((2 - d) * 3 + kx) * O + ob);
            // replicate padding of the forward: its output row 0 read x row 0 through ky = 0 as well (and row H - 1
            // through ky = 2): the border gy row (d = 1) once more through that kernel row
            if (d == 1 && border) {
              const float4 e = *reinterpret_cast<const float4*>(wc + ((h == 0 ? 0 : 2) * 3 + kx) * O + ob);
              k4.x += e.x; k4.y += e.y; k4.z += e.z; k4.w += e.w;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              int j = w0 + k + 1 - kx;         // output pixel w sees gy pixel w + 1 - kx through tap kx
              j = j < 0 ? j + W : (j >= W ? j - W : j);
              j = j >= W ? 0 : j;              // (pixels past the wave's share: computed, not stored)
              const float4 g = *reinterpret_cast<const float4*>(grow + (int64_t)j * O);
              s[k] = fmaf(g.x, k4.x, fmaf(g.y, k4.y, fmaf(g.z, k4.z, fmaf(g.w, k4.w, s[k]))));
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = wave_sum(s[k]);
        const int w = w0 + k;
        if (lane == 0 && w < min(W, (wave + 1) * per)) {
          const int64_t at = ((int64_t)row * W + w) * ldx + c0 + ci;
          gx[at] = v + (resid ? resid[at] : 0.f);
        }
      }
    }
  }
  const int nz = cend - c0 - nc;
  for (int i = threadIdx.x; i < W * nz; i += 256) {
    const int64_t at = ((int64_t)row * W + i / nz) * ldx + c0 + nc + i % nz;
    gx[at] = resid ? resid[at] : 0.f;
  }
}

// The plane images from weight VALUES w [O][9][Cp] fp32 (the layout of the conv's own operand) -- for the passes that
// do not run on the weight bank (R1's double backward): one thread per 16-byte image unit.
__global__ __launch_bounds__(256) void x3_image_fwd_kernel(bf16_t* __restrict__ w3, const float* __restrict__ w, int O, int Cp,
                                                           int nch) {
  const int n = O * 9 * nch * 4;
  const size_t plane = (size_t)(O >> 6) * nch * (X_PW * 4);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int c8 = i % (nch * 4), r = i / (nch * 4);
    const int t = r % 9, o = r / 9;
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    if (c8 * 8 < Cp) {   // Cp % 8 == 0
      const float4* src = reinterpret_cast<const float4*>(w + ((size_t)o * 9 + t) * Cp + c8 * 8);
      lo = src[0];
      hi = src[1];
    }
    uint4 h, m, l;
    split3x8(lo, hi, h, m, l);
    const int row = t * 64 + (o & 63);
    const size_t unit = ((size_t)(o >> 6) * nch + (c8 >> 2)) * (X_PW * 4) + (row >> 3) * 32 + (c8 & 3) * 8 + (row & 7);
    uint4* img = reinterpret_cast<uint4*>(w3);
    img[unit] = h;
    img[plane + unit] = m;
    img[2 * plane + unit] = l;
  }
}

__global__ __launch_bounds__(256) void x3_image_bwd_kernel(bf16_t* __restrict__ w3t, const float* __restrict__ w, int O, int Cp,
                                                           int nslab) {
  const int CS = nslab * 64, nch = O / 32;
  const int n = CS * 9 * (O / 8);
  const size_t plane = (size_t)nslab * nch * (X_PW * 4);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int c = i % CS, r = i / CS;
    const int t = r % 9, o8 = r / 9;
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = w[((size_t)(o8 * 8 + k) * 9 + t) * Cp + c];
    uint4 h, m, l;
    split3x8(make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]), h, m, l);
    const int row = (8 - t) * 64 + (c & 63);      // the gradient's tap order
    const size_t unit = ((size_t)(c >> 6) * nch + (o8 >> 2)) * (X_PW * 4) + (row >> 3) * 32 + (o8 & 3) * 8 + (row & 7);
    uint4* img = reinterpret_cast<uint4*>(w3t);
    img[unit] = h;
    img[plane + unit] = m;
    img[2 * plane + unit] = l;
  }
}

template <int HZ, bool XE>
int launch_x3(float* y, const float* x, const bf16_t* wimg, CX3 p, hipStream_t st) {
  auto kern = conv_x3_kernel<HZ, XE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)X_LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  p.tiles_h = (p.H + 3) / 4;
  p.tiles_w = p.W / 32;
  p.ntiles = p.B * p.tiles_h * p.tiles_w;
  p.npairs = (p.ntiles + 1) / 2;
  p.total = p.npairs * p.nslab;
  p.per_xcd = (p.total + 7) / 8;
#ifdef DGV2_ABLATE
  p.ablate = getenv("DGV2_X3_ABLATE") ? atoi(getenv("DGV2_X3_ABLATE")) : 0;
  {
    const int nr = (p.ablate & 128) ? 1 : 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(x3_abl_noreads), &nr, sizeof(int));
  }
#endif
  kern<<<p.per_xcd * 8, 512, X_LDS, st>>>(y, x, wimg, p);
  return 0;
}

}  // namespace

// y [B, H, W, O] (fp32) = act( conv3x3_ring(x [B, H, W, Cx] fp32, w) + bias ) * scale + resid, the weights as the three
// plane images dgv2_conv_weight_bank_ex writes for dtype fp32 (w8: [3][O / 64][ceil(Cx / 32)][2304 units of 8 bf16]).
// fp32-equivalent (six bf16 products per multiply, fp32 accumulation).  Stride 1, pad 1, rows replicate, columns wrap.
// x_exact: the caller's promise that channels [0, x_exact) of x hold bf16-representable values (the activations of a bf16
// trunk widened to fp32: the discriminator's features in front of its fp32 epilogue) -- their m and l planes are zero, so
// their three products are not issued: the same sum at half the MFMAs.  A value that breaks the promise raises
// bit DGV2_STATUS_X_INEXACT of *status (the caller's device word; required when x_exact > 0).  0: no promise.
// DGV2_ENOTSUP where the kernel does not cover the geometry (callers then run dgv2_conv_taps in fp32).
extern "C" int dgv2_conv3x3_x3_fwd(void* y, const void* x, const void* w3, int B, int H, int W, int Cx, int x_exact, int O,
                                   const float* bias, const void* resid, int act, float alpha, float scale, int* status,
                                   void* stream) {
  if (!y || !x || !w3 || B < 1 || H < 1 || W < 1 || Cx < 1 || O < 1 || x_exact < 0 || x_exact > Cx) return DGV2_EINVAL;
  if (x_exact > 0 && !status) return DGV2_EINVAL;   // a promise without a word to report its breach to
  if (act != 0 && act != 3) return DGV2_EINVAL;
  static const bool off = getenv("DGV2_NO_CONV_X3") != nullptr;   // A/B switch for benchmarking
  if (off || W % 32 || Cx % 8 || O % 64 || Cx < 64) return DGV2_ENOTSUP;
  if (!aligned16(y) || !aligned16(x) || !aligned16(w3) || (resid && !aligned16(resid))) return DGV2_ENOTSUP;
  if ((int64_t)B * H * W * (Cx > O ? Cx : O) >= (1ll << 31)) return DGV2_ENOTSUP;
  CX3 p;
  p.B = B; p.H = H; p.W = W; p.Cx = Cx; p.nchunks = (Cx + 31) / 32; p.O = O; p.ldy = O; p.nslab = O / 64;
  p.xexact = x_exact;
  p.status = status;
  p.bias = bias; p.resid = (const float*)resid; p.act = act; p.alpha = alpha; p.scale = scale;
  const int rc = x_exact >= 32 ? launch_x3<0, true>((float*)y, (const float*)x, (const bf16_t*)w3, p, (hipStream_t)stream)
                               : launch_x3<0, false>((float*)y, (const float*)x, (const bf16_t*)w3, p, (hipStream_t)stream);
  if (rc) return rc;
  DGV2_RETURN_LAST();
}

// The data gradient of the same conv: gx [B, H, W, ldx] (fp32), channels [0, C) = the gradient (+ resid), [C, ldx) = resid
// or zero.  w3t: the transposed plane images ([3][ldx / 64][O / 32][2304 units], taps in the gradient's order) cover the
// channels of whole 64-channel slabs below C; wt [ldx, 9, O] fp32 (the bank's transposed layout) serves the up to four
// channels behind them in exact fp32 (up to 16, a launch pass each).  H >= 2 (a border row may not be its own mirror).
extern "C" int dgv2_conv3x3_x3_dgrad(void* gx, const void* gy, const void* w3t, const void* wt, int B, int H, int W, int C,
                                     int ldx, int O, const void* resid, void* stream) {
  if (!gx || !gy || !w3t || !wt || B < 1 || H < 1 || W < 1 || C < 1 || ldx < C || O < 1) return DGV2_EINVAL;
  static const bool off = getenv("DGV2_NO_CONV_X3") != nullptr;
  const int nslab = C / 64, ntail = C - nslab * 64;
  if (off || H < 2 || W % 32 || O % 32 || O < 64 || nslab < 1 || ntail > 16 || ldx % 4 || nslab != ldx / 64)
    return DGV2_ENOTSUP;
  if (!aligned16(gx) || !aligned16(gy) || !aligned16(w3t) || !aligned16(wt) || (resid && !aligned16(resid))) return DGV2_ENOTSUP;
  if ((int64_t)B * H * W * (ldx > O ? ldx : O) >= (1ll << 31)) return DGV2_ENOTSUP;
  CX3 p;
  p.B = B; p.H = H; p.W = W; p.Cx = O; p.nchunks = O / 32; p.O = nslab * 64; p.ldy = ldx; p.nslab = nslab;
  p.xexact = 0;
  p.status = nullptr;
  p.bias = nullptr; p.resid = (const float*)resid; p.act = 0; p.alpha = 0.2f; p.scale = 1.f;
  hipStream_t st = (hipStream_t)stream;
  const int rc = launch_x3<1, false>((float*)gx, (const float*)gy, (const bf16_t*)w3t, p, st);
  if (rc) return rc;
  if (ldx > nslab * 64)
    x3_dgrad_tail_kernel<<<B * H, 256, 0, st>>>((float*)gx, (const float*)gy, (const float*)wt, (const float*)resid, B, H, W,
                                               ldx, O, nslab * 64, ntail, ldx);
  DGV2_RETURN_LAST();
}

// The plane images of both entries above from weight VALUES w [O, 9, Cp] fp32 (what dgv2_conv_taps takes) -- for passes
// that do not run on the weight bank.  w3 [3][O / 64][ceil(Cp / 32)][2304 units], w3t [3][Cp / 64][O / 32][2304 units]; either
// may be NULL.  O % 64 == 0, Cp % 8 == 0, Cp >= 64.
extern "C" int dgv2_conv_x3_images(void* w3, void* w3t, const void* w, int O, int Cp, void* stream) {
  if ((!w3 && !w3t) || !w || O < 64 || O % 64 || Cp < 64 || Cp % 8) return DGV2_EINVAL;
  if ((w3 && !aligned16(w3)) || !aligned16(w) || (w3t && !aligned16(w3t))) return DGV2_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nch = (Cp + 31) / 32;
  if (w3)
    x3_image_fwd_kernel<<<grid_for((int64_t)O * 9 * nch * 4, 256), 256, 0, st>>>((bf16_t*)w3, (const float*)w, O, Cp, nch);
  if (w3t)
    x3_image_bwd_kernel<<<grid_for((int64_t)(Cp / 64) * 64 * 9 * (O / 8), 256), 256, 0, st>>>((bf16_t*)w3t, (const float*)w, O,
                                                                                                 Cp, Cp / 64);
  DGV2_RETURN_LAST();
}

