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

template <bool HZERO>
__global__ __launch_bounds__(256, 3) void conv3x3_strip_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ w, SGeom g) {
  extern __shared__ __attribute__((aligned(16))) uint4 ring[];   // [24 rows][34 cols][4 chunks], chunk ^= (col>>2)&3
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
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

  const int nit = g.H >> 3;
  dma_group(0);
  dma_group(1);
  for (int it = 0; it < nit; ++it) {
    // group it+1 was issued one iteration ago, BEFORE this wave's 4 stores of that iteration: in-order retirement
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's pieces landed; every wave is done with the group about to be refilled
    asm volatile("" ::: "memory");
    if (it + 2 <= nit) dma_group(it + 2);

    f32x16 acc[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
      const int ro = 8 * it + 2 * wave + f;   // output row
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int ri = ro + dy;
        if (HZERO && (ri < 0 || ri >= g.H)) continue;   // wave-uniform
        const uint4* rowp = ring + ((ri + 7) % S_RING) * S_ROWSLOTS;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            union { uint4 u; bf16x8 v; } ua, ub;
            ua.u = A[(dy + 1) * 3 + d][h];
            ub.u = rowp[bslot[d][h]];
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[f], 0, 0, 0);
          }
      }
      if (HZERO && g.border && (ro == 0 || ro == g.H - 1)) {
        // replicate-padding terms of the data gradient: output row 0 also sees gy row 0 through the ky = 0 weights
        // (the taps of dy = +1), output row H-1 sees gy row H-1 through the ky = 2 weights (dy = -1)
        const uint4* rowp = ring + ((ro + 7) % S_RING) * S_ROWSLOTS;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            union { uint4 u; bf16x8 v; } ua, ub;
            ub.u = rowp[bslot[d][h]];
            if (ro == 0) {
              ua.u = A[6 + d][h];
              acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[f], 0, 0, 0);
            }
            if (ro == g.H - 1) {
              ua.u = A[d][h];
              acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[f], 0, 0, 0);
            }
          }
      }
    }

    // ---- epilogue: lane (n, kg) holds channels 8j + 4kg .. +3 (j = 0..3) of pixel (ro, w0 + n) ----
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int ro = 8 * it + 2 * wave + f;
      unsigned pk[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        union { uint2 u; bf16_t e[4]; } q;
        const float4 b4 = *reinterpret_cast<const float4*>(s_bias + 8 * j + 4 * kg);
        const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = acc[f][4 * j + r] + bq[r];
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
      const int rw = lane >> 4;
      const int chunk = (rw & 1) * 2 + (rw >> 1);
      const int64_t off = (((int64_t)b * g.H + ro) * g.W + w0 + (lane & 15)) * S_C + chunk * 8;
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        uint4 v = out[sidx];
        const int64_t o = off + sidx * 16 * S_C;
        if (g.resid) {   // added on the packed run, as the generic engine's fast path does
          vec16<bf16_t> av, r;
          av.raw = v;
          r.load(g.resid + o);
#pragma unroll
          for (int e = 0; e < 8; ++e) av.set(e, av.get(e) + r.get(e));
          v = av.raw;
          *reinterpret_cast<uint4*>(y + o) = v;
        } else {
          // streaming store: the output (268 MB at B = 128) is far larger than L2 + MALL and is read next by another
          // kernel; with a residual operand the same store was 25 % SLOWER (the resid lines just read leave with it)
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
          const u32x4 q = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(q, reinterpret_cast<u32x4*>(y + o));
        }
      }
    }
  }
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
  if (hzero) conv3x3_strip_kernel<true><<<grid, 256, lds, st>>>((bf16_t*)y, (const bf16_t*)x, (const bf16_t*)w, g);
  else conv3x3_strip_kernel<false><<<grid, 256, lds, st>>>((bf16_t*)y, (const bf16_t*)x, (const bf16_t*)w, g);
  return 0;
}
