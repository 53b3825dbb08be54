// Streaming 3x3 convolution for the full-resolution, 32-channel layers of the discriminator (ResidualBlock 0 conv1:
// 64 x 512 pixels, 32 -> 32 channels, ring padding; reference: ops.Conv2d, gans/models/ops/common.py:187-210 at
// gans/models/dusty_v2.py:329) and its stride-1 data gradient.  These layers are HBM-bound (4.19 MB of activation
// traffic against 0.6 GFLOP per image); SQ counters of the generic tap-list engine (profiles/round2_sq_counters.txt)
// showed it VALU-issue bound there: 11 vector instructions per MFMA for staging addresses, LDS swizzles and the
// epilogue.  This kernel removes that work instead of tuning it:
//   * a block owns a 32-column STRIP of one image and walks DOWN it, 8 output rows per iteration; input rows stream
//     through a 24-row LDS ring by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, no
//     per-iteration address arithmetic beyond one row offset per slot), three 8-row groups in flight: one being
//     computed on (+ the two halo rows of its neighbours), one landed, one in flight.  Every input row is read once
//     per strip: the halo is the 2 ring-wrapped columns only (34/32 = 1.06x, was 1.33x with 8x32 tiles).
//   * the whole 3x3x32x32 weight set lives in REGISTERS as v_mfma_f32_32x32x16_bf16 A-fragments (72 VGPRs): no weight
//     traffic through LDS at all; O = 32 is exactly the M of that instruction.
//   * 32x32x16 MFMAs: half the instruction count and half the LDS bytes per FLOP of the 16x16x32 form, and 24 of their
//     32 cycles are free for other vector issue (MI355X_MICROARCH.md) -- the B-fragment reads are ONE ds_read_b128 per
//     MFMA at precomputed per-lane offsets + a wave-uniform row base.
//   * epilogue: bias + leaky ReLU + bf16, fragment halves exchanged with v_permlane32_swap so that every lane stores
//     16-byte runs; an optional residual (the sibling branch's gradient in the data-gradient call) is added there.
//   * blockIdx -> (image, strip) puts the strips of one image on ONE XCD, so the two halo columns a strip shares with
//     its neighbours are L2 hits.
// Ring padding: the W coordinate wraps in the DMA source address.  H: replicate rows are the clamped source rows
// (forward); for the data gradient (hzero) out-of-image rows are skipped and the replicate-padding terms are the
// three extra taps of output rows 0 and H-1.
#include "gemm_core.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct SGeom {
  int B, H, W;
  int widx[9];          // weight slot of tap (dy+1)*3 + (dx+1) in w[o][wtaps][32]
  int wtaps;
  int border;           // hzero only: add the replicate-padding terms of output rows 0 and H-1
  const float* bias;
  const bf16_t* resid;
  int act;
  float alpha, scale;
};

constexpr int S_C = 32;                  // channels in = out
constexpr int S_COLS = 34;               // strip columns incl. the two halo columns
constexpr int S_ROWSLOTS = S_COLS * 4;   // 16-byte slots per ring row
constexpr int S_GROUP = 8 * S_ROWSLOTS;  // slots per 8-row group (1088 = 17 wave pieces)
constexpr int S_RING = 24;

constexpr int NT_LD = 0;   // cache policy of the LDS-DMA loads: nontemporal (2) measured 5-10 % slower

template <bool HZERO, bool RESID>
__global__ __launch_bounds__(256, RESID ? 2 : 3) void conv3x3_strip_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ w, SGeom g) {
  extern __shared__ __attribute__((aligned(16))) uint4 ring[];   // [24 rows][34 cols][4 chunks], chunk ^= (col>>2)&3
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: everything derived from it is a scalar branch
  const int n = lane & 31, kg = lane >> 5;
  const int strips = g.W >> 5;
  const int nb = gridDim.x;
  // XCD-aware block order: hardware deals consecutive block ids round-robin over the 8 XCDs; give each XCD a
  // contiguous range of (image, strip) so that neighbouring strips share an L2
  const int bid = blockIdx.x;
  const int idx = (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid;
  const int b = idx / strips, w0 = (idx - b * strips) << 5;
  const bf16_t* xb = x + (int64_t)b * g.H * g.W * S_C;

  // ---- weights: A fragments (row m = lane % 32 = output channel, k = 8 * (lane / 32) + i) ----
  uint4 A[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      A[t][h] = *reinterpret_cast<const uint4*>(w + ((int64_t)n * g.wtaps + g.widx[t]) * S_C + h * 16 + kg * 8);

  // ---- DMA slots of this thread: piece j covers ring slots [j*256 + wave*64, +64) of a group ----
  int coloff[5], rowin[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int id = min(j * 256 + tid, S_GROUP - 1);
    const int rr = id / S_ROWSLOTS, rem = id - rr * S_ROWSLOTS;
    const int col = rem >> 2, chp = rem & 3;
    int gw = w0 - 1 + col;
    gw = gw < 0 ? gw + g.W : (gw >= g.W ? gw - g.W : gw);
    coloff[j] = gw * S_C + (chp ^ ((col >> 2) & 3)) * 8;
    rowin[j] = rr;
  }
  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef __attribute__((address_space(1))) const void gbl_void_t;
  auto dma_group = [&](int grp) {          // input rows 8*grp-7 .. 8*grp -> ring rows (grp % 3) * 8 ..
    uint4* base = ring + (grp % 3) * S_GROUP;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == 4 && wave != 0) continue;   // 17 pieces: the last one is wave 0's
      int r = 8 * grp - 7 + rowin[j];
      r = r < 0 ? 0 : (r >= g.H ? g.H - 1 : r);
      const bf16_t* src = xb + (int64_t)r * g.W * S_C + coloff[j];
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(base + j * 256 + wave * 64), 16, 0, NT_LD);
    }
  };

  // ---- per-lane B-fragment slot of (dx, K half): pixel column n + dx + 1 of the strip, logical chunk 2h + kg ----
  int bslot[3][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = n + d;
      bslot[d][h] = col * 4 + ((2 * h + kg) ^ ((col >> 2) & 3));
    }
  // bias in LDS behind the ring (a global read in the epilogue would make the compiler drain the DMA in flight with
  // vmcnt(0); 16 registers per lane are better spent elsewhere): this lane's channels are 8j + 4kg .. +3, j = 0..3
  float* s_bias = reinterpret_cast<float*>(ring + S_RING * S_ROWSLOTS);
  if (tid < S_C) s_bias[tid] = g.bias ? g.bias[tid] : 0.f;

  // ---- hand-issued LDS reads and residual loads --------------------------------------------------------------------
  // With an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front of every ds_read that follows it in program
  // order (it cannot prove the read does not alias the DMA's destination) and in front of the first use of any ordinary
  // global load: the loop below would wait for the group it has JUST requested, i.e. one group in flight per block
  // instead of two.  So the B-fragment reads, the bias reads and the residual loads are asm with counted waits:
  //   vmcnt  (in issue order per wave and iteration): [4 residual loads] [4-5 DMA pieces of group it+2] [4 stores]
  //          top of iteration it: all but the 4 youngest (the stores of it-1) done  =>  group it+1 has landed;
  //          before the epilogue: all but the DMA pieces done  =>  the residual values are in their registers
  //   lgkmcnt: a ring of RD fragment registers, PF reads in flight ahead of the MFMA that consumes them
  const unsigned lds_off = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ring;
  unsigned baddr[3][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int h = 0; h < 2; ++h) baddr[d][h] = lds_off + (unsigned)bslot[d][h] * 16u;
  const unsigned bias_addr = lds_off + (unsigned)(S_RING * S_ROWSLOTS) * 16u + (unsigned)kg * 16u;   // + 32 j bytes
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define STRIP_DS_READ(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define STRIP_DS_READ_OFF(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define STRIP_LGKM_WAIT(dst, cnt) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "n"(cnt))

  // this lane's position in the regrouped epilogue (see below): pixel (lane & 15) + 16 s of the row, 16-byte chunk
  const int rw = lane >> 4;
  const int chunk = (rw & 1) * 2 + (rw >> 1);
  const int64_t lane_off = (int64_t)(w0 + (lane & 15)) * S_C + chunk * 8;

  const int nit = g.H >> 3;
  dma_group(0);
  dma_group(1);
  for (int it = 0; it < nit; ++it) {
    // group it+1 was issued one iteration ago, BEFORE this wave's 4 stores of that iteration: in-order retirement
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's pieces landed; every wave is done with the group about to be refilled
    asm volatile("" ::: "memory");

    const int ro0 = 8 * it + 2 * wave;                       // this wave's two output rows ro0, ro0 + 1
    const int64_t row_off = ((int64_t)b * g.H + ro0) * g.W * S_C + lane_off;
    u32x4 rs[RESID ? 2 : 1][2];                              // residual operand of the 4 stores
    if (RESID) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const bf16_t* rp = g.resid + row_off + (int64_t)f * g.W * S_C + sidx * 16 * S_C;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rs[f][sidx]) : "v"(rp) : "memory");
        }
    }
    const bool issued = it + 2 <= nit;
    if (issued) dma_group(it + 2);

    // ---- 36 (read, MFMA) steps: step s = ((f * 3 + dyi) * 3 + d) * 2 + h ----
    // Rows outside the image (data gradient only): the DMA left a CLAMPED copy there, which is exactly the operand of
    // the replicate-padding term -- output row 0 sees gy row 0 through the ky = 0 weights (the taps of dy = +1), row
    // H-1 sees gy row H-1 through the ky = 2 weights (dy = -1): the dead tap row is not skipped but re-weighted, so a
    // border row costs what an interior row costs.  Without border terms (plain zero padding) it is skipped.
    int mode_top = 0, mode_bot = 0;                          // 0 plain, 1 border weights, 2 skip (wave-uniform)
    if (HZERO) {
      if (ro0 == 0) mode_top = g.border ? 1 : 2;
      if (ro0 + 1 == g.H - 1) mode_bot = g.border ? 1 : 2;
    }
    unsigned rb[2][3];                                       // ring row byte offsets of (f, dyi)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int dyi = 0; dyi < 3; ++dyi) rb[f][dyi] = (unsigned)((ro0 + f + dyi - 1 + 7) % S_RING) * (S_ROWSLOTS * 16u);

#pragma unroll
    for (int f = 0; f < 2; ++f) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      {
        constexpr int NS = 18, PF = RESID ? 3 : 4, RD = RESID ? 4 : 6;   // the residual operand needs 16 registers
        u32x4 bf[RD];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
          const unsigned addr = baddr[(q / 2) % 3][q % 2] + rb[f][q / 6];
          STRIP_DS_READ(bf[q % RD], addr);
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const int dyi = q / 6, d = (q / 2) % 3, h = q % 2;
          const int inflight = (q + PF <= NS ? PF : NS - q) - 1;
          switch (inflight) {
            case 3: STRIP_LGKM_WAIT(bf[q % RD], 3); break;
            case 2: STRIP_LGKM_WAIT(bf[q % RD], 2); break;
            case 1: STRIP_LGKM_WAIT(bf[q % RD], 1); break;
            default: STRIP_LGKM_WAIT(bf[q % RD], 0); break;
          }
          union { uint4 u; bf16x8 v; } ua;
          union { u32x4 u; bf16x8 v; } ub;
          ub.u = bf[q % RD];
          if (HZERO && f == 0 && dyi == 0 && mode_top) {
            if (mode_top == 1) {
              ua.u = A[6 + d][h];
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
            }
          } else if (HZERO && f == 1 && dyi == 2 && mode_bot) {
            if (mode_bot == 1) {
              ua.u = A[d][h];
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
            }
          } else {
            ua.u = A[dyi * 3 + d][h];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
          }
          if (q + PF < NS) {
            const int qn = q + PF;
            const unsigned addr = baddr[(qn / 2) % 3][qn % 2] + rb[f][qn / 6];
            STRIP_DS_READ(bf[qn % RD], addr);
          }
        }
      }

      // ---- epilogue: lane (n, kg) holds channels 8j + 4kg .. +3 (j = 0..3) of pixel (ro, w0 + n) ----
      if (RESID && f == 0) {   // all but this iteration's DMA pieces done: the residual loads (older) have landed
        if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (wave == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        // the registers become usable HERE: tied to the wait itself the compiler may set the operands up with copies
        // in front of it (it did: v_mov of registers whose data had not arrived), tied one by one behind it any
        // copy lands after the wait
#pragma unroll
        for (int ff = 0; ff < 2; ++ff)
#pragma unroll
          for (int sidx = 0; sidx < 2; ++sidx) asm volatile("" : "+v"(rs[ff][sidx]));
      }
      unsigned pk[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        union { uint2 u; bf16_t e[4]; } q;
        u32x4 bq;                                            // bias of channels 8j + 4kg .. +3
        switch (j) {
          case 0: STRIP_DS_READ_OFF(bq, bias_addr, 0); break;
          case 1: STRIP_DS_READ_OFF(bq, bias_addr, 32); break;
          case 2: STRIP_DS_READ_OFF(bq, bias_addr, 64); break;
          default: STRIP_DS_READ_OFF(bq, bias_addr, 96); break;
        }
        STRIP_LGKM_WAIT(bq, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = acc[4 * j + r] + __uint_as_float(bq[r]);
          if (g.act == 3) t = fmaxf(t, t * g.alpha) * g.scale;   // leaky ReLU, 0 <= alpha <= 1
          q.e[r] = (bf16_t)t;
        }
        pk[j][0] = q.u.x;
        pk[j][1] = q.u.y;
      }
      // groups (0,1) and (2,3): lanes n and n+32 swap one packed quad each, leaving 8 consecutive channels per lane:
      // kg = 0 -> [own j, partner j] = channels 16*(j/2) + 0..7; kg = 1 -> [partner j+1, own j+1] = 16*(j/2) + 8..15
      uint4 out[2];
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        auto s0 = __builtin_amdgcn_permlane32_swap(pk[j][0], pk[j + 1][0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(pk[j][1], pk[j + 1][1], false, false);
        out[j >> 1] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
      // out[0] / out[1] are the 16-byte chunks (kg, 2 + kg) of pixel n: stored as they stand, each instruction would
      // write HALF of every pixel's 64 bytes (32 segments of 32 B).  Exchanging the odd 16-lane rows of out[0] with the
      // even rows of out[1] (v_permlane16_swap) regroups them so that instruction s writes pixels 16 s .. 16 s + 15
      // COMPLETELY: one contiguous 1 KB run per instruction, whole 128-byte lines.
      {
        unsigned* a = reinterpret_cast<unsigned*>(&out[0]);
        unsigned* c = reinterpret_cast<unsigned*>(&out[1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          auto r = __builtin_amdgcn_permlane16_swap(a[e], c[e], false, false);
          a[e] = r[0];
          c[e] = r[1];
        }
      }
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        uint4 v = out[sidx];
        const int64_t o = row_off + (int64_t)f * g.W * S_C + sidx * 16 * S_C;
        if (RESID) {   // added on the packed run, as the generic engine's fast path does
          vec16<bf16_t> av, r;
          av.raw = v;
          r.raw = make_uint4(rs[f][sidx][0], rs[f][sidx][1], rs[f][sidx][2], rs[f][sidx][3]);
#pragma unroll
          for (int e = 0; e < 8; ++e) av.set(e, av.get(e) + r.get(e));
          v = av.raw;
        }
        // streaming store: the output (268 MB at B = 128) is far larger than L2 + MALL and is read next by another kernel
        const u32x4 q = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(q, reinterpret_cast<u32x4*>(y + o));
      }
    }
  }
#undef STRIP_DS_READ
#undef STRIP_DS_READ_OFF
#undef STRIP_LGKM_WAIT
}

}  // namespace

// Called by dgv2_conv_taps_ex (conv_direct.hip) for the geometries documented there.  Returns -2 when this kernel does
// not cover the call (the generic engine then runs it).
int dgv2_conv_strip_try(void* y, const void* x, const void* w, int B, int H, int W, int Cin, int O, int ntaps, int wtaps,
                        const int* taps4, int nextra, const int* extras5, int hzero, const float* bias,
                        const void* resid, int act, float alpha, float scale, hipStream_t st) {
  static const bool off = getenv("DGV2_NO_STRIP") != nullptr;   // A/B switch for benchmarking
  if (off || Cin != S_C || O != S_C || ntaps != 9 || wtaps < 9 || (W & 31) || (H & 7) || H < 16 || W < 32) return -2;
  if (B <= 0 || (int64_t)B * H * W * S_C >= (1LL << 40)) return -2;
  SGeom g;
  g.B = B; g.H = H; g.W = W; g.wtaps = wtaps;
  for (int t = 0; t < 9; ++t) g.widx[t] = -1;
  for (int t = 0; t < 9; ++t) {
    const int dy = taps4[4 * t], dx = taps4[4 * t + 1], wi = taps4[4 * t + 2];
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || taps4[4 * t + 3] != 0) return -2;
    g.widx[(dy + 1) * 3 + dx + 1] = wi;
  }
  for (int t = 0; t < 9; ++t)
    if (g.widx[t] < 0) return -2;
  g.border = 0;
  if (nextra) {
    // exactly the replicate-padding terms of the stride-1 data gradient: rows 0 / H-1, dy = 0, the weights of dy = +1 / -1
    if (!hzero || nextra != 6) return -2;
    int seen = 0;
    for (int e = 0; e < 6; ++e) {
      const int* q = extras5 + 5 * e;
      if (q[0] != 0 || q[1] < -1 || q[1] > 1 || q[3] != 0) return -2;
      if (q[4] == 0 && q[2] == g.widx[6 + q[1] + 1]) seen |= 1 << (q[1] + 1);
      else if (q[4] == H - 1 && q[2] == g.widx[q[1] + 1]) seen |= 8 << (q[1] + 1);
      else return -2;
    }
    if (seen != 63) return -2;
    g.border = 1;
  }
  g.bias = bias; g.resid = (const bf16_t*)resid; g.act = act; g.alpha = alpha; g.scale = scale;
  const size_t lds = sizeof(uint4) * S_RING * S_ROWSLOTS + sizeof(float) * S_C;   // 52,224 B ring + bias
  dim3 grid(B * (W >> 5));
  if (hzero && resid) conv3x3_strip_kernel<true, true><<<grid, 256, lds, st>>>((bf16_t*)y, (const bf16_t*)x, (const bf16_t*)w, g);
  else if (hzero) conv3x3_strip_kernel<true, false><<<grid, 256, lds, st>>>((bf16_t*)y, (const bf16_t*)x, (const bf16_t*)w, g);
  else if (resid) conv3x3_strip_kernel<false, true><<<grid, 256, lds, st>>>((bf16_t*)y, (const bf16_t*)x, (const bf16_t*)w, g);
  else conv3x3_strip_kernel<false, false><<<grid, 256, lds, st>>>((bf16_t*)y, (const bf16_t*)x, (const bf16_t*)w, g);
  return 0;
}
