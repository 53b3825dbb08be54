// Level-input modulated 1x1 conv of the generator with the up-sampling COMMUTED past the contraction:
//   reference (gans/models/dusty_v2.py:153-162, ops/style.py:105-118):
//       y = act( c * ( W_a . up2(h)  +  W_s . PE ) + bias )        W = [W_a | W_s] per sample, up2 = ring-aware FIR
//   a 1x1 conv acts per pixel and the FIR per channel, so  W_a . up2(h) == up2( W_a . h ):  the xa part of the
//   contraction runs at a QUARTER of the pixels (T = W_a . h at the previous level's resolution,
//   dgv2_modconv_up_t) and dgv2_modconv_up_fwd evaluates
//       y[b,p,:] = act( c * ( up2(T)[b,p,:]  +  sum_k W_s[b,:,k] PE[p,k] ) + bias ).
// Round 3: up2 is part of the MFMA chain.  up2(T)[:, p] = sum_j T[:, j] U[j, p] with U the (sparse, constant)
// interpolation matrix of the block's Resample(up=2): for the 32 output pixels of a wave (one row segment) only the 2
// low-resolution rows x 32 columns of an aligned window carry weight, so up2 is FOUR more K-steps of the same
// v_mfma_f32_32x32x16_bf16 chain: A = T[b] (kept CHANNEL-major [B, 32, Hin*Win], so a lane's fragment is one 16-byte
// run of 8 neighbouring low-resolution pixels of its channel), B = U built once per wave into registers next to the
// PE fragments.  The epilogue that used to unpack 8 tap vectors and run 64 FMAs per lane and sample (8.4 VALU
// instructions per MFMA, 45 % of the wave cycles waiting for issue) is gone: what is left per output value is one
// fma (c, bias with the activation gain folded in), one mul + max (leaky ReLU), the bf16 pack and the statistic.
// Structure (as modconv_pe.hip: a block owns 256 pixels and walks the samples; B fragments stay in registers):
//   * O = 32 = M: a wave owns 32 pixels (one N fragment), 32 PE + 4 up K-steps per sample;
//   * per-sample weights W_s[b] (32 x Ks) and the wave's window of T[b]: two LDS buffers each, filled one sample ahead
//     by LDS-DMA in the image [K step][K half][o][16 B] (a wave's A-fragment read = two contiguous 512-byte runs,
//     conflict-free); A fragments are read by hand-issued ds_read_b128 with counted lgkmcnt (hipcc would park every
//     ds_read behind the DMA in flight);
//   * epilogue: v_permlane32_swap on the fp32 accumulators gives each lane two runs of 8 consecutive channels ->
//     two 16-byte stores per lane; optional sum-of-squares partials of y.
#include <type_traits>

#include "gemm_core.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

struct MUGeom {
  int B, P, Wout, Hin, Win;   // output pixels P = Hout * Wout; T is [B, O, Hin * Win] in 8-pixel units
  int O;                      // output channels of the layer: blockIdx.z owns the 32-channel slab [32 z, 32 z + 32)
  int Ks;                     // w is the per-sample MFMA image [B][Ks/32][2 mt][4 kq][16 o][8 k] (dgv2_modconv_up_t)
  int samples_per_block;
  const int* idx_h;           // [Hout][2] low-res rows of the two taps, coef_h [Hout][2]
  const float* coef_h;
  const int* idx_w;           // [Wout][2]
  const float* coef_w;
  const float* bias;
  const float* in_scale;      // device scalar (or NULL): the layer's input-magnitude factor c, applied to the B operands
  int act;
  float alpha, scale;
  float* sumsq;
#ifdef DGV2_ABLATE   // benchmarking builds only (make ABLATE=1): wrong results by design, never in the shipped library
  int ablate;        // DGV2_MU_ABLATE: 1 skip the per-sample DMA, 2 skip the MFMA loop, 4 skip epilogue math + stores,
                     // 8 skip the barrier, 16 skip the T-window DMA only
#endif
};

constexpr int UP32 = 2;       // K-steps (of 32) of the up-sampling part: 2 low-res rows x 32 window columns

// v_mfma_f32_16x16x32_bf16 form (round 3, late): M = 2 tiles of 16 output channels, N = 2 tiles of 16 pixels per wave,
// K-steps of 32.  Same LDS traffic per FLOP as the 32x32x16 form (two A-fragment reads per 32 k), the same operand
// registers, but the 16x16 shape holds a higher clock under load on this part (guide: +12...15 % FLOP/s at equal cycles)
// and its C input may differ from its destination: the first MFMA of every chain takes the bias registers as C, so the
// bias costs nothing (no K-step, no accumulator fill).  Fragment maps (cdna_hip_programming.md, section 3): A lane l =
// row l % 16, k = 8 (l / 16) + j; B lane l = column l % 16, same k; C/D column l % 16, rows 4 (l / 16) + r.
template <int KS32, bool LRELU>   // Ks / 32; leaky ReLU or no activation
__global__ __launch_bounds__(512, 1) void modconv_up_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ tcm,
                                                            const bf16_t* __restrict__ xs, const bf16_t* __restrict__ w,
                                                            MUGeom g) {
  constexpr int KT = KS32 + UP32;                   // K-steps per sample
  constexpr int WBUF = KS32 * 128;                  // 16-byte slots of one sample's weights: [KS32][2 mt][4 kq][16 o]
  constexpr int TBUF = UP32 * 128;                  // ... of one wave's T window: [UP32][2 mt][4 kq][16 o]
  constexpr int NW = WBUF / 512;                    // weight DMA pieces per thread
  constexpr int NTP = TBUF / 64;                    // T-window pieces per wave (one per (row tap, M tile))
  constexpr int TOFF = 2 * WBUF;                    // T windows behind the two weight buffers: [2][8 waves][TBUF]
  extern __shared__ __attribute__((aligned(16))) uint4 lds_w[];
  const unsigned lds_off = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_w;   // LDS byte address

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: everything derived from it is wave-uniform
  const int n16 = lane & 15, kq = lane >> 4;
  // Levels with more than 32 output channels (round 4: O = 64 / 128 at levels 3 / 2) run as 32-channel SLABS, one per
  // blockIdx.z: T is [B][Hin][O / 16 tiles][Win / 8][16 o][8 px], the weight image [B][O / 32][Ks / 32][2][4][16][8] -- a
  // slab's per-sample weights and its window of T are the same contiguous 1 KB pieces as at O = 32.
  const int zs = blockIdx.z, MT = g.O >> 4, nslab = g.O >> 5;
  const int p0 = blockIdx.x * 256 + wave * 32;      // P % 32 == 0: a wave is all live or all dead
  const int b0 = blockIdx.y * g.samples_per_block;
  const int b1 = min(b0 + g.samples_per_block, g.B);
  const bool live = p0 < g.P;
  const int pw = live ? p0 : 0;                     // a dead wave computes pixel tile 0 and stores nothing
  const int Y = pw / g.Wout, X0 = pw - Y * g.Wout;  // Wout % 32 == 0: the wave's pixels are one row segment
  const int winbase = floormod((X0 >> 1) - 8, g.Win);

  const int iy[2] = {g.idx_h[2 * Y], g.idx_h[2 * Y + 1]};
  const float wy[2] = {g.coef_h[2 * Y], g.coef_h[2 * Y + 1]};
  // per-lane source offsets of the wave's T window: piece q = (row tap s, M tile mt), lane (kq, o16): the 8 low-res
  // pixels of window unit kq, channel 16 mt + o16.  T is [B][Hin][2 mt][Win/8][16 o][8 px]: the four units of a piece
  // are ONE contiguous 1 KB (two runs where the window wraps around the ring) -- as four 256-byte runs of a
  // [pixel unit][32 o] layout the T pieces cost 13 of 90 us, the equally large contiguous weight pieces nothing
  int toff[NTP];
#pragma unroll
  for (int q = 0; q < NTP; ++q) {
    int c = winbase + 8 * kq;
    c = c >= g.Win ? c - g.Win : c;
    toff[q] = (((iy[q >> 1] * MT + 2 * zs + (q & 1)) * (g.Win >> 3) + (c >> 3)) * 16 + n16) * 8;
  }

  // LDS slot L = ((s * 2 + mt) * 4 + kq) * 16 + o16  <-  image slot L of sample b: every piece is one contiguous 1 KB
  // (a piece gathered from 64 different lines costs the address path 8x the cycles; with eight such pieces per wave and
  // sample the waves queued at ISSUE and the transfer did not overlap the MFMA loop).  Piece q of a sample: q < NW
  // weights, then the wave's T window.
  auto dma_piece = [&](int q, int b, int buf) {
#ifdef DGV2_ABLATE
    if ((g.ablate & 1) || (q >= NW && (g.ablate & 16))) return;
#endif
    if (q < NW)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(w + (((int64_t)b * nslab + zs) * WBUF + tid + q * 512) * 8),
                                       (lds_void_t*)(lds_w + buf * WBUF + q * 512 + wave * 64), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(tcm + (int64_t)b * g.O * g.Hin * g.Win + toff[q - NW]),
                                       (lds_void_t*)(lds_w + TOFF + (buf * 8 + wave) * TBUF + (q - NW) * 64), 16, 0, 0);
  };
  constexpr int NQ = NW + NTP;

  // ---- B operands (column = pixel 16 nt + n16, k = 8 kq + j), registers for the whole walk: PE, then U ----
  uint4 bf[KT][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    // xs is the FRAGMENT image of the PE: [P/16][Ks/32][4 kq][16 pixels][8 k] -- a wave's load of one fragment is one
    // contiguous 1 KB (from the pixel-major [P, Ks] tensor every load touched 16 half lines: the prologue, 33.5 MB per
    // sample half, was 18 of the kernel's 77 us)
    const uint4* pe = reinterpret_cast<const uint4*>(xs) + (int64_t)((pw >> 4) + nt) * (KS32 * 64) + lane;
#pragma unroll
    for (int s = 0; s < KS32; ++s) bf[s][nt] = pe[s * 64];
    const int X = X0 + 16 * nt + n16;
    const int ix0 = g.idx_w[2 * X], ix1 = g.idx_w[2 * X + 1];
    const float wx0 = g.coef_w[2 * X], wx1 = g.coef_w[2 * X + 1];
#pragma unroll
    for (int s = 0; s < UP32; ++s) {   // K index of element j: low-res row tap s, window column 8 kq + j
      vec16<bf16_t> u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int c = winbase + 8 * kq + j;
        c = c >= g.Win ? c - g.Win : c;
        u.set(j, wy[s] * ((c == ix0 ? wx0 : 0.f) + (c == ix1 ? wx1 : 0.f)));
      }
      bf[KS32 + s][nt] = u.raw;
    }
  }
  if (b0 < b1) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) dma_piece(q, b0, 0);
  }
  // The layer's input-magnitude factor c (a scalar that exists only after THIS step's statistic has been folded into the
  // running mean, i.e. after the helper that wrote T and the weight image has read h) rides on the B operands: every
  // term of the chain is (A operand) x (PE or U), so c * acc costs 288 multiplies per lane ONCE per block instead of a
  // multiply per output value, and T / the image need not wait for the statistic.
  if (g.in_scale) {
    const float cin = *g.in_scale;
#pragma unroll
    for (int s = 0; s < KT; ++s)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        vec16<bf16_t> v;
        v.raw = bf[s][nt];
#pragma unroll
        for (int e = 0; e < 8; ++e) v.set(e, v.get(e) * cin);
        bf[s][nt] = v.raw;
      }
  }
  // ---- bias * gain as the C input of the first MFMA of each chain: this lane's rows 16 mt + 4 kq + r ----
  const float gain = LRELU ? g.scale * 0.5f * (1.f + g.alpha) : 1.f;   // see the epilogue
  f32x4 bias_c[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_c[mt][r] = g.bias ? g.bias[32 * zs + 16 * mt + 4 * kq + r] * gain : 0.f;

  float ss = 0.f;
  const float lr_k = (1.f - g.alpha) / (1.f + g.alpha);
  // The epilogue of sample b - 1 rides in the MFMA loop of sample b, a few instructions per K-step: an MFMA holds the
  // vector issue for a part of its cycles only, so the loop has room for them and the epilogue costs no time of its own.
  // Two accumulator sets alternate (the sample loop is unrolled by two), so nothing is copied between samples.
  // part nt: the two M tiles of pixel tile nt (this lane: pixel 16 nt + n16, channels 4 kq + r and 16 + 4 kq + r):
  // leaky ReLU as ONE fma -- with s = scale (1 + alpha) / 2 folded into T, the weights and the bias,
  // act(f) scale = max(f, alpha f) scale = f' + k |f'|, f' = s f, k = (1 - alpha) / (1 + alpha) -- bf16 pack, statistic
  // of the ROUNDED pairs (v_dot2c_f32_bf16), one v_permlane16_swap round (lanes kq and kq ^ 1 trade a packed quad:
  // an even kq then owns channels [4 kq, 4 kq + 8), an odd one [16 + 4 (kq - 1), + 8)) and ONE 16-byte store.
  // (Measured on one box: fmaxf() form 114 us, mul + fma 97, v_pk_mul + asm v_max 121 -- the asm operands cost copies.)
  const int px = pw + n16;
  const int co = (kq & 1) ? 16 + 4 * (kq - 1) : 4 * kq;
  auto epi_part = [&](f32x4 (&v)[2][2], int nt, int bprev) {
#ifdef DGV2_ABLATE
    if (g.ablate & 4) {
      if (nt == 0 && v[0][0][0] == 123.456f) y[0] = (bf16_t)v[1][1][3];   // keeps the accumulators alive
      return;
    }
#endif
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    union { unsigned u[2]; bf16x2 p[2]; } qa, qb;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float a0 = v[0][nt][2 * h], a1 = v[0][nt][2 * h + 1], c0 = v[1][nt][2 * h], c1 = v[1][nt][2 * h + 1];
      if (LRELU) {
        a0 = fmaf(fabsf(a0), lr_k, a0);
        a1 = fmaf(fabsf(a1), lr_k, a1);
        c0 = fmaf(fabsf(c0), lr_k, c0);
        c1 = fmaf(fabsf(c1), lr_k, c1);
      }
      qa.p[h] = bf16x2{(bf16_t)a0, (bf16_t)a1};
      qb.p[h] = bf16x2{(bf16_t)c0, (bf16_t)c1};
      ss = __builtin_amdgcn_fdot2_f32_bf16(qa.p[h], qa.p[h], ss, false);
      ss = __builtin_amdgcn_fdot2_f32_bf16(qb.p[h], qb.p[h], ss, false);
    }
    // odd 16-lane rows of qa <-> even rows of qb
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      auto r = __builtin_amdgcn_permlane16_swap(qa.u[h], qb.u[h], false, false);
      qa.u[h] = r[0];
      qb.u[h] = r[1];
    }
    // even kq: qa = own block-A quad, qb = the block-A quad of kq + 1;  odd kq: qa = block-B quad of kq - 1, qb = own
    if (live)
      *reinterpret_cast<uint4*>(y + ((int64_t)bprev * g.P + px + 16 * nt) * g.O + 32 * zs + co) = make_uint4(qa.u[0], qa.u[1], qb.u[0], qb.u[1]);
  };

  auto step = [&](auto has_prev, f32x4 (&acc)[2][2], f32x4 (&prev)[2][2], int buf, int b) {
    constexpr bool HP = decltype(has_prev)::value;
    // this sample's weights and T window were issued during the previous step, the last piece AFTER that step's two
    // stores: vector-memory operations retire in issue order, so a full drain waits for nothing but the DMA itself
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef DGV2_ABLATE
    if (!(g.ablate & 8))
#endif
    __builtin_amdgcn_s_barrier();   // every wave's pieces landed; every wave is done with the other buffer
    asm volatile("" ::: "memory");
    const int bn = min(b + 1, b1 - 1);

    // A fragments: row = lane % 16 (channel 16 mt + row), k = 8 (lane / 16) + j  <-  slot (s * 2 + mt) * 64 + lane.
    // The reads are issued as asm (ring of RD registers, PF reads in flight, explicit lgkmcnt): left to the compiler,
    // every ds_read that follows an LDS-DMA in program order is preceded by `s_waitcnt vmcnt(0)` (it cannot prove the
    // read does not alias the DMA's destination), which parks the whole MFMA loop behind the NEXT sample's transfer.
#ifdef DGV2_ABLATE
    if (g.ablate & 2) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = bias_c[mt];
#pragma unroll
      for (int q = 0; q < NQ; ++q) dma_piece(q, bn, buf ^ 1);
      return;
    }
#endif
    {
      constexpr int NR = 2 * KT;          // A-fragment reads per sample: (K-step, M tile)
      constexpr int PF = 4, RD = 6;
      const unsigned abase = lds_off + (unsigned)(buf * WBUF + lane) * 16u;
      const unsigned tbase = lds_off + (unsigned)(TOFF + (buf * 8 + wave) * TBUF + lane) * 16u;
      u32x4 a[RD];
#define DGV2_DS_READ(dst, i)                                                                                     \
  do {                                                                                                           \
    if ((i) < 2 * KS32) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(abase), "n"(((i) % (2 * KS32)) * 1024)); \
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(tbase), "n"((((i) >= 2 * KS32 ? (i) - 2 * KS32 : 0)) * 1024)); \
  } while (0)
#define DGV2_LGKM_WAIT(dst, cnt) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "n"(cnt))
#pragma unroll
      for (int i = 0; i < PF; ++i) DGV2_DS_READ(a[i % RD], i);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int s = i >> 1, mt = i & 1;
        // reads return in order: all but the (issued - i - 1) youngest are back
        const int inflight = (i + PF <= NR ? PF : NR - i) - 1;
        switch (inflight) {
          case 3: DGV2_LGKM_WAIT(a[i % RD], 3); break;
          case 2: DGV2_LGKM_WAIT(a[i % RD], 2); break;
          case 1: DGV2_LGKM_WAIT(a[i % RD], 1); break;
          default: DGV2_LGKM_WAIT(a[i % RD], 0); break;
        }
        union { u32x4 u; bf16x8 v; } ua;
        ua.u = a[i % RD];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          union { uint4 u; bf16x8 v; } ub;
          ub.u = bf[s][nt];
          // the chain starts from the bias (C input != destination): no zero fill, no bias K-step
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, s == 0 ? bias_c[mt] : acc[mt][nt], 0, 0, 0);
        }
        if (i + PF < NR) DGV2_DS_READ(a[(i + PF) % RD], i + PF);   // slot last read two groups ago
        // riders of this read group: the next sample's DMA pieces from the very first groups on (issued at groups
        // 20..34 the last ones had not landed at the next barrier: the T pieces alone cost 13 of 90 us), the previous
        // sample's epilogue in between
        if ((i & 1) == 0 && i / 2 < NQ) dma_piece(i / 2, bn, buf ^ 1);
        if constexpr (HP) {
          if (i == 17) epi_part(prev, 0, b - 1);
          if (i == 25) epi_part(prev, 1, b - 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef DGV2_DS_READ
#undef DGV2_LGKM_WAIT
    }
  };

  if (b0 < b1) {
    f32x4 accA[2][2], accB[2][2];
    step(std::false_type{}, accA, accB, 0, b0);
    int b = b0 + 1;
    for (; b + 1 < b1; b += 2) {
      step(std::true_type{}, accB, accA, (b - b0) & 1, b);
      step(std::true_type{}, accA, accB, (b + 1 - b0) & 1, b + 1);
    }
    if (b < b1) {
      step(std::true_type{}, accB, accA, (b - b0) & 1, b);
      epi_part(accB, 0, b1 - 1);
      epi_part(accB, 1, b1 - 1);
    } else {
      epi_part(accA, 0, b1 - 1);
      epi_part(accA, 1, b1 - 1);
    }
  }
  if (g.sumsq) {
    __shared__ float red[16];
    const float s = block_sum(live ? ss : 0.f, red);
    if (tid == 0) g.sumsq[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
  }
}

// T[b][o][p] = sum_c w[b][o][c] h[b][p][c] at the LOW resolution, stored as [B][Plow/8][32 o][8 px] (the A operand of
// the up-sampling K-steps above: a wave's fragment read of two 8-pixel units is one contiguous 1 KB).  A wave owns 32
// pixels: D[pixel][o] with A = h rows, B = w rows (both K-contiguous 16-byte fragments), so the result has its
// channel on the lane and 8 consecutive pixels per lane after one permlane32 exchange.  The first block of every
// sample also repacks the PE columns of the sample's weights into the MFMA image of dgv2_modconv_up_fwd.
template <int KA16>   // Ka / 16
__global__ __launch_bounds__(512) void modconv_up_t_kernel(bf16_t* __restrict__ tcm, bf16_t* __restrict__ wimg,
                                                           const bf16_t* __restrict__ h, const bf16_t* __restrict__ w,
                                                           const float* __restrict__ row_scale, float gain, int Plow,
                                                           int Wlow, int I, int koff, int Ks, int Otot) {
  constexpr int O = 32, Ka = KA16 * 16;               // O: the slab this block computes, [32 z, 32 z + 32) of Otot
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = lane & 31, kg = lane >> 5;
  const int b = blockIdx.y, zs = blockIdx.z, MT = Otot >> 4, nslab = Otot >> 5;
  w += (int64_t)zs * O * I;                             // this slab's rows of every sample's weights (sample stride Otot * I)
  if (row_scale) row_scale += zs * O;
  if (blockIdx.x == 0 && wimg) {
    // image slot L = ((s * 2 + mt) * 4 + kq) * 16 + o16  <-  w[b][16 mt + o16][koff + 32 s + 8 kq .. + 8]: the A fragments of
    // v_mfma_f32_16x16x32_bf16, one contiguous 1 KB per (K-step, M tile)
    const int slots = (Ks >> 5) * 128;
    for (int L = tid; L < slots; L += 512) {
      const int o = ((L >> 6) & 1) * 16 + (L & 15), kq4 = (L >> 4) & 3, s32 = L >> 7;
      vec16<bf16_t> v;
      v.raw = *reinterpret_cast<const uint4*>(w + ((int64_t)b * Otot + o) * I + koff + s32 * 32 + kq4 * 8);
      const float c = (row_scale ? row_scale[o] : 1.f) * gain;
      if (c != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v.set(e, v.get(e) * c);
      }
      *reinterpret_cast<uint4*>(wimg + (((int64_t)b * nslab + zs) * slots + L) * 8) = v.raw;
    }
  }
  const int p0 = blockIdx.x * 256 + wave * 32;
  if (p0 >= Plow) return;
  const bf16_t* hp = h + ((int64_t)b * Plow + p0 + n) * Ka + kg * 8;
  const bf16_t* wp = w + ((int64_t)b * Otot + n) * I + kg * 8;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  uint4 fa[KA16], fb[KA16];
#pragma unroll
  for (int s = 0; s < KA16; ++s) {
    fa[s] = *reinterpret_cast<const uint4*>(hp + s * 16);
    fb[s] = *reinterpret_cast<const uint4*>(wp + s * 16);
  }
#pragma unroll
  for (int s = 0; s < KA16; ++s) {
    union { uint4 u; bf16x8 v; } ua, ub;
    ua.u = fa[s];
    ub.u = fb[s];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
  }
  // lane (o = n, kg) holds pixels 8j + 4kg .. +3 (j = 0..3) -> two runs of 8 consecutive pixels (= two units);
  // layout [B][Hlow][Otot / 16 tiles][Wlow/8][16 o][8 px] (Wlow % 32 == 0: the wave's 32 pixels lie in one row)
  const int trow = p0 / Wlow, tcol = p0 - trow * Wlow;
  bf16_t* dst = tcm + (int64_t)b * Plow * Otot +
                ((((int64_t)trow * MT + 2 * zs + (n >> 4)) * (Wlow >> 3) + (tcol >> 3)) * 16 + (n & 15)) * 8;
  const float c = (row_scale ? row_scale[n] : 1.f) * gain;
#pragma unroll
  for (int j = 0; j < 4; j += 2) {
    vec16<bf16_t> o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[4 * j + r]), __float_as_uint(acc[4 * (j + 1) + r]),
                                                false, false);
      o.set(r, __uint_as_float(s[0]) * c);
      o.set(4 + r, __uint_as_float(s[1]) * c);
    }
    *reinterpret_cast<uint4*>(dst + (j + kg) * (16 * 8)) = o.raw;
  }
}

// dgv2_modconv_up_t and dgv2_up2_lag_sumsq as ONE pass over h (round 3, late): a wave owns a 32-column segment and walks
// RS rows down it.  Row i's fragments (lane = pixel n, 8 channels per K-step half) feed the T chain as above; the same
// registers, the fragments of the pixel one column to the right (their own loads: the lines are the neighbour lanes')
// and the two of row i + 1 (which become row i's on the next trip) give the five neighbourhood products of the
// quadratic form with v_dot2_f32_bf16 on the packed pairs.  h is read once (+ one halo row per RS) instead of twice.
template <int KA16, bool STAT>
__global__ __launch_bounds__(512) void modconv_up_tl_kernel(bf16_t* __restrict__ tcm, bf16_t* __restrict__ wimg,
                                                            const bf16_t* __restrict__ h, const bf16_t* __restrict__ w,
                                                            float gain, const float* __restrict__ ghd,
                                                            const float* __restrict__ gho, const float* __restrict__ gwd,
                                                            const float* __restrict__ gwo, int Hlow, int Wlow, int I,
                                                            int koff, int Ks, int RS, float* __restrict__ partial,
                                                            int Otot, int z0) {
  constexpr int O = 32, Ka = KA16 * 16;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = lane & 31, kg = lane >> 5;
  // slab z0 + blockIdx.z of the Otot output channels (the statistic instance runs slab 0 only: grid.z = 1)
  const int b = blockIdx.y, zs = z0 + blockIdx.z, MT = Otot >> 4, nslab = Otot >> 5;
  w += (int64_t)zs * O * I;
  if (blockIdx.x == 0 && wimg) {   // the PE columns of the sample's weights as the MFMA image (see modconv_up_t_kernel)
    const int slots = (Ks >> 5) * 128;
    for (int L = tid; L < slots; L += 512) {
      const int o = ((L >> 6) & 1) * 16 + (L & 15), kq4 = (L >> 4) & 3, s32 = L >> 7;
      vec16<bf16_t> v;
      v.raw = *reinterpret_cast<const uint4*>(w + ((int64_t)b * Otot + o) * I + koff + s32 * 32 + kq4 * 8);
      if (gain != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v.set(e, v.get(e) * gain);
      }
      *reinterpret_cast<uint4*>(wimg + (((int64_t)b * nslab + zs) * slots + L) * 8) = v.raw;
    }
  }
  const int segs = Wlow >> 5;
  const int wg = blockIdx.x * 8 + wave;
  const int rg = wg / segs, seg = wg - rg * segs;
  const int i0 = rg * RS;
  float st = 0.f;
  if (i0 < Hlow) {
    const int iend = min(i0 + RS, Hlow);
    const int col = seg * 32 + n, col1 = col + 1 == Wlow ? 0 : col + 1;
    const bf16_t* hb = h + (int64_t)b * Hlow * Wlow * Ka + kg * 8;
    const bf16_t* wp = w + ((int64_t)b * Otot + n) * I + kg * 8;
    uint4 fb[KA16], a[KA16], bq[KA16], c[KA16], d[KA16];
#pragma unroll
    for (int s = 0; s < KA16; ++s) {
      a[s] = *reinterpret_cast<const uint4*>(hb + ((int64_t)i0 * Wlow + col) * Ka + s * 16);
      if (STAT) bq[s] = *reinterpret_cast<const uint4*>(hb + ((int64_t)i0 * Wlow + col1) * Ka + s * 16);
    }
#pragma unroll
    for (int s = 0; s < KA16; ++s) fb[s] = *reinterpret_cast<const uint4*>(wp + s * 16);
    const float wd = STAT ? gwd[col] : 0.f, wo = STAT ? gwo[col] : 0.f;
    bf16_t* dst0 = tcm + (int64_t)b * Hlow * Wlow * Otot +
                   ((int64_t)((2 * zs + (n >> 4)) * (Wlow >> 3) + (seg * 4)) * 16 + (n & 15)) * 8;
    for (int i = i0; i < iend; ++i) {
      const int i1 = min(i + 1, Hlow - 1);
      if (STAT || i + 1 < iend) {
#pragma unroll
        for (int s = 0; s < KA16; ++s) {
          c[s] = *reinterpret_cast<const uint4*>(hb + ((int64_t)i1 * Wlow + col) * Ka + s * 16);
          if (STAT) d[s] = *reinterpret_cast<const uint4*>(hb + ((int64_t)i1 * Wlow + col1) * Ka + s * 16);
        }
      }
      f32x16 acc;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
      for (int s = 0; s < KA16; ++s) {
        union { uint4 u; bf16x8 v; } ua, ub;
        ua.u = a[s];
        ub.u = fb[s];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
      }
      // [B][Hlow][Otot / 16 tiles][Wlow/8][16 o][8 px]: row i, units 4 seg + (j + kg)
      bf16_t* dst = dst0 + (int64_t)i * MT * (Wlow >> 3) * 128;
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        vec16<bf16_t> o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[4 * j + r]), __float_as_uint(acc[4 * (j + 1) + r]),
                                                     false, false);
          o.set(r, __uint_as_float(sw[0]) * gain);
          o.set(4 + r, __uint_as_float(sw[1]) * gain);
        }
        *reinterpret_cast<uint4*>(dst + (j + kg) * (16 * 8)) = o.raw;
      }
      if (STAT) {
        float aa = 0.f, ab = 0.f, ac = 0.f, x = 0.f;
#pragma unroll
        for (int s = 0; s < KA16; ++s) {
          union { uint4 u; bf16x2 p[4]; } qa, qb, qc, qd;
          qa.u = a[s]; qb.u = bq[s]; qc.u = c[s]; qd.u = d[s];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            aa = __builtin_amdgcn_fdot2_f32_bf16(qa.p[k], qa.p[k], aa, false);
            ab = __builtin_amdgcn_fdot2_f32_bf16(qa.p[k], qb.p[k], ab, false);
            ac = __builtin_amdgcn_fdot2_f32_bf16(qa.p[k], qc.p[k], ac, false);
            x = __builtin_amdgcn_fdot2_f32_bf16(qa.p[k], qd.p[k], x, false);
            x = __builtin_amdgcn_fdot2_f32_bf16(qb.p[k], qc.p[k], x, false);
          }
        }
        st += ghd[i] * (wd * aa + 2.f * wo * ab) + 2.f * gho[i] * (wd * ac + wo * x);
      }
#pragma unroll
      for (int s = 0; s < KA16; ++s) {
        a[s] = c[s];
        if (STAT) bq[s] = d[s];
      }
    }
  }
  if (STAT) {
    __shared__ float red[16];
    const float tot = block_sum(st, red);
    if (tid == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = tot;
  }
}

// Sum of squares of up2(h) WITHOUT up-sampling: with U = Uh (x) Uw the up-2 operator, sum (U h)^2 = h^T (Gh (x) Gw) h,
// Gh = Uh^T Uh and Gw = Uw^T Uw tridiagonal (a 4-tap FIR at up = 2 has two taps per output and axis; ring: circulant).
// Per low-res pixel (i, j) and channel:
//   ghd[i] (gwd[j] a^2 + 2 gwo[j] a b) + 2 gho[i] (gwd[j] a c + gwo[j] (a d + b c)),
//   a = h[i][j], b = h[i][j+1], c = h[i+1][j], d = h[i+1][j+1]     (j + 1 wraps; gho[Hin-1] = 0)
// A thread owns one column x 8 channels and walks RS rows carrying (a, b) down.
template <int RS>
__global__ __launch_bounds__(256) void up2_lag_sumsq_kernel(const bf16_t* __restrict__ h, const float* __restrict__ ghd,
                                                            const float* __restrict__ gho, const float* __restrict__ gwd,
                                                            const float* __restrict__ gwo, int Hin, int Win, int C,
                                                            float* __restrict__ partial) {
  const int cg = C >> 3;                               // channel groups of 8 (256 % cg == 0)
  const int g8 = threadIdx.x % cg, jl = threadIdx.x / cg;
  const int j = blockIdx.x * (256 / cg) + jl;
  const int i0 = blockIdx.y * RS;
  const int b = blockIdx.z;
  float s = 0.f;
  if (j < Win) {
    const int j1 = j + 1 == Win ? 0 : j + 1;
    const float wd = gwd[j], wo = gwo[j];
    const bf16_t* hb = h + (int64_t)b * Hin * Win * C + g8 * 8;
    auto ld = [&](int i, int jj, float (&f)[8]) {
      const uint4 q = *reinterpret_cast<const uint4*>(hb + ((int64_t)i * Win + jj) * C);
      const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f[2 * k] = __uint_as_float(u[k] << 16);
        f[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
      }
    };
    float a[8], bq[8], c[8], d[8];
    ld(i0, j, a);
    ld(i0, j1, bq);
    const int iend = min(i0 + RS, Hin);
    for (int i = i0; i < iend; ++i) {
      const int i1 = min(i + 1, Hin - 1);
      ld(i1, j, c);
      ld(i1, j1, d);
      float aa = 0.f, ab = 0.f, ac = 0.f, x = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        aa = fmaf(a[k], a[k], aa);
        ab = fmaf(a[k], bq[k], ab);
        ac = fmaf(a[k], c[k], ac);
        x = fmaf(a[k], d[k], x);
        x = fmaf(bq[k], c[k], x);
      }
      s += ghd[i] * (wd * aa + 2.f * wo * ab) + 2.f * gho[i] * (wd * ac + wo * x);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a[k] = c[k];
        bq[k] = d[k];
      }
    }
  }
  __shared__ float red[16];
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
}

}  // namespace

// y[b,p,:O] = act( up2(T)[b,p,:] + sum_{k<Ks} xs[p,k] W_s[b,:,k] + bias )   (bf16 in / out; O in {32, 64, 128}: generator
//   levels 4 / 3 / 2, as 32-channel slabs of the same kernel), T and W_s carrying the
//   layer's input-magnitude factor row_scale[o] times gain = scale (1 + alpha) / 2 for act 3, 1 for act 0 (this entry
//   puts the same gain on the bias; the leaky ReLU is then one fma, see the kernel):
//   t [B,Hin,O/16,Win/8,16,8] = row_scale * gain * W_a . h at the previous level's resolution in 8-pixel units and
//   wimg [B,O/32,Ks/32,2,4,16,8] = the PE columns row_scale * gain * W_s of the prepared per-sample weights as the MFMA image,
//   both from dgv2_modconv_up_t; up2 by the
//   two-tap tables idx/coef [Hout][2], [Wout][2] (native.ResampleSpec.tables of the block's up-2 Resample, zero-padded to
//   two taps); xs: the batch-shared PE [Hout*Wout, Ks] as the fragment image [Hout*Wout/16][Ks/32][4][16][8]
//   (pixel tile of 16, K-step, k quarter, pixel, 8 k: element [p][k] of the PE at [p/16][k/32][(k%32)/8][p%16][k%8]).
//   Contract on the tables (the caller checks it once per table set, they live on the device): Wout % 32 == 0,
//   Win % 8 == 0, Win >= 32, and for every output column X both W taps lie in the aligned 32-column window
//   [(X & ~31) / 2 - 8, +32) mod Win (true for every up-2 FIR with at most 4 taps).  Ks in {512}.
//   sumsq: one partial per block (of the stored, bf16-rounded values).
extern "C" int dgv2_modconv_up_fwd(void* y, const void* t, const void* xs, const void* wimg, int B, int Hout, int Wout,
                                   int Hin, int Win, int Ks, int O, const int* idx_h, const float* coef_h,
                                   const int* idx_w, const float* coef_w, const float* bias, const float* in_scale,
                                   int act, float alpha, float scale, int dtype, float* sumsq, int sumsq_cap,
                                   int* sumsq_used, void* stream) {
  if (sumsq_used) *sumsq_used = 0;
  if (!y || !t || !xs || !wimg || !idx_h || !coef_h || !idx_w || !coef_w || B <= 0 || Hout <= 0 || Wout <= 0) return DGV2_EINVAL;
  if (dtype != DGV2_BF16 || (act != 0 && act != 3) || (O != 32 && O != 64 && O != 128) || Ks != 512 || (Wout & 31) ||
      (Win & 7) || Win < 32 || Hin <= 0 || (act == 3 && !(scale > 0.f)))
    return DGV2_ENOTSUP;
  if (!aligned16(y) || !aligned16(t) || !aligned16(xs) || !aligned16(wimg)) return DGV2_EINVAL;
  const int P = Hout * Wout;
  MUGeom g{B, P, Wout, Hin, Win, O, Ks, 1, idx_h, coef_h, idx_w, coef_w, bias, in_scale, act, alpha, scale, sumsq};
#ifdef DGV2_ABLATE
  g.ablate = getenv("DGV2_MU_ABLATE") ? atoi(getenv("DGV2_MU_ABLATE")) : 0;
#endif
  const int tiles = (P + 255) / 256, nslab = O / 32;
  int nsplit = (256 + tiles * nslab - 1) / (tiles * nslab);   // one resident block per CU (the B fragments fill the registers)
  nsplit = nsplit < 1 ? 1 : (nsplit > B ? B : nsplit);
  g.samples_per_block = (B + nsplit - 1) / nsplit;
  nsplit = (B + g.samples_per_block - 1) / g.samples_per_block;
  if (g.sumsq && sumsq_used && tiles * nsplit * nslab <= sumsq_cap) *sumsq_used = tiles * nsplit * nslab;
  else g.sumsq = nullptr;
  constexpr int KS32 = 16;
  const size_t lds = sizeof(uint4) * (2 * KS32 * 128 + 2 * 8 * UP32 * 128);
  auto kern = act == 3 ? modconv_up_kernel<KS32, true> : modconv_up_kernel<KS32, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid(tiles, nsplit, nslab);
  kern<<<grid, 512, lds, (hipStream_t)stream>>>((bf16_t*)y, (const bf16_t*)t, (const bf16_t*)xs, (const bf16_t*)wimg, g);
  DGV2_RETURN_LAST();
}

// The low-resolution part of the commuted level-input conv and the operand images of dgv2_modconv_up_fwd:
//   tcm [B,Hlow,2,Wlow/8,16,8]:  T[b][o][p] = f[o] sum_{c<Ka} w[b][o][c] h[b][p][c] in 8-pixel units, per low-res row the
//                         units of channels 0..15 then those of channels 16..31 (the four units a T piece of
//                         dgv2_modconv_up_fwd reads are then one contiguous 1 KB), f[o] = row_scale[o] (1 if NULL) * gain
//   wimg [B,Ks/32,2,4,16,8] (or NULL): f[o] w[b][16 mt + o16][koff + 32 s + 8 kq + j] at [b][s][mt][kq][o16][j]
// h [B,Hlow*Wlow,Ka], w [B,O,I] prepared per-sample weights (bf16); O in {32, 64, 128} (32-channel slabs: tcm then is
// [B,Hlow,O/16,Wlow/8,16,8], wimg [B,O/32,Ks/32,2,4,16,8]), Ka in {64, 128, 256}, Wlow % 32 == 0, Ks % 32 == 0.
// replaces: the xa columns of the ModConv2d contraction, gans/models/ops/style.py:105-118.
extern "C" int dgv2_modconv_up_t(void* tcm, void* wimg, const void* h, const void* w, const float* row_scale, float gain,
                                 int B, int Hlow, int Wlow, int Ka, int Ks, int O, int I, int koff, int dtype,
                                 void* stream) {
  const int Plow = Hlow * Wlow;
  if (!tcm || !h || !w || B <= 0 || Hlow <= 0 || Wlow <= 0) return DGV2_EINVAL;
  if (dtype != DGV2_BF16 || (O != 32 && O != 64 && O != 128) || (Ka != 64 && Ka != 128 && Ka != 256) || (Wlow & 31) ||
      (I & 7) || I < Ka || (wimg && ((Ks & 31) || (koff & 7) || koff + Ks > I)))
    return DGV2_ENOTSUP;
  if (!aligned16(tcm) || !aligned16(h) || !aligned16(w) || (wimg && !aligned16(wimg))) return DGV2_EINVAL;
  dim3 grid((Plow + 255) / 256, B, O / 32);
#define DGV2_T(KA16)                                                                                                  \
  modconv_up_t_kernel<KA16><<<grid, 512, 0, (hipStream_t)stream>>>((bf16_t*)tcm, (bf16_t*)wimg, (const bf16_t*)h,    \
                                                                   (const bf16_t*)w, row_scale, gain, Plow, Wlow, I, koff, Ks, O)
  if (Ka == 64) DGV2_T(4);
  else if (Ka == 128) DGV2_T(8);
  else DGV2_T(16);
#undef DGV2_T
  DGV2_RETURN_LAST();
}

// dgv2_modconv_up_t (with row_scale = NULL: T and the image carry `gain` only) and, when ghd != NULL, dgv2_up2_lag_sumsq
// of the same h in ONE pass: the layer's input statistic must be folded into its running mean BEFORE the factor c
// exists, so c cannot ride in T when T and the statistic share a pass -- dgv2_modconv_up_fwd takes it as `in_scale`.
// Contract of sumsq / cap / used as in dgv2_resample_tab_sq (one partial per block); Gram vectors as dgv2_up2_lag_sumsq.
// replaces: the xa columns of the ModConv2d contraction and its ema_var statistic, gans/models/ops/style.py:98-118.
extern "C" int dgv2_modconv_up_t_lag(void* tcm, void* wimg, const void* h, const void* w, float gain, const float* ghd,
                                     const float* gho, const float* gwd, const float* gwo, int B, int Hlow, int Wlow,
                                     int Ka, int Ks, int O, int I, int koff, int dtype, float* sumsq, int sumsq_cap,
                                     int* sumsq_used, void* stream) {
  if (sumsq_used) *sumsq_used = 0;
  if (!tcm || !h || !w || B <= 0 || Hlow <= 0 || Wlow <= 0) return DGV2_EINVAL;
  const bool stat = ghd != nullptr;
  if (stat && (!gho || !gwd || !gwo || !sumsq || !sumsq_used)) return DGV2_EINVAL;
  if (dtype != DGV2_BF16 || (O != 32 && O != 64 && O != 128) || (Ka != 64 && Ka != 128) || (Wlow & 31) || (I & 7) ||
      I < Ka || (wimg && ((Ks & 31) || (koff & 7) || koff + Ks > I)))
    return DGV2_ENOTSUP;   // (Ka = 256, level 2: five fragment sets of 16 do not fit the registers -> dgv2_modconv_up_t +
                           //  dgv2_up2_lag_sumsq as two launches)
  if (!aligned16(tcm) || !aligned16(h) || !aligned16(w) || (wimg && !aligned16(wimg))) return DGV2_EINVAL;
  const int nslab = O / 32;
  const char* rs_env = getenv("DGV2_TL_ROWS");   // rows a wave walks (A/B benchmarking)
  // 8 rows per wave where that still leaves a block per CU (B = 64 at 32 x 256: 1 / 2 / 4 / 8 / 16 rows measured
  // 51 / 35 / 28 / 23 / 37 us; the two separate passes 29 + 16)
  const int segs = Wlow >> 5;
  int RS = (int)((int64_t)Hlow * segs * B / 2048);
  RS = RS < 1 ? 1 : (RS > 8 ? 8 : RS);
  if (rs_env && atoi(rs_env) > 0) RS = atoi(rs_env);
  const int groups = (Hlow + RS - 1) / RS;
  dim3 grid((segs * groups + 7) / 8, B);
  if (stat) {
    if ((int64_t)grid.x * grid.y > sumsq_cap) return DGV2_ENOTSUP;
    *sumsq_used = (int)(grid.x * grid.y);
  }
  // the statistic is a property of h alone: slab 0's launch takes it, the other slabs run the plain instance
#define DGV2_TL(KA16, ST, GZ, Z0)                                                                                       \
  modconv_up_tl_kernel<KA16, ST><<<dim3(grid.x, grid.y, GZ), 512, 0, (hipStream_t)stream>>>(                            \
      (bf16_t*)tcm, (bf16_t*)wimg, (const bf16_t*)h, (const bf16_t*)w, gain, ghd, gho, gwd, gwo, Hlow, Wlow, I, koff, Ks, \
      RS, sumsq, O, Z0)
  if (Ka == 64) {
    if (stat) { DGV2_TL(4, true, 1, 0); if (nslab > 1) DGV2_TL(4, false, nslab - 1, 1); }
    else DGV2_TL(4, false, nslab, 0);
  } else {
    if (stat) { DGV2_TL(8, true, 1, 0); if (nslab > 1) DGV2_TL(8, false, nslab - 1, 1); }
    else DGV2_TL(8, false, nslab, 0);
  }
#undef DGV2_TL
  DGV2_RETURN_LAST();
}

// Per-block partial sums of  sum_{b,p,c} up2(h)[b,p,c]^2  from h at its own (low) resolution: the input statistic of the
// commuted level-input conv (ModConv2d's ema_var update, gans/models/ops/style.py:98-103) without a pass at the
// up-sampled size.  ghd / gho [Hin], gwd / gwo [Win]: diagonal and (i, i+1) / (j, j+1 mod Win) entries of the Gram
// matrices Uh^T Uh, Uw^T Uw of the up-2 operator's axis factors (tridiagonal; the caller builds and checks them).
// h [B,Hin,Win,C] bf16, C % 8 == 0, 256 % (C / 8) == 0.  Contract of sumsq / cap / used as in dgv2_resample_tab_sq.
extern "C" int dgv2_up2_lag_sumsq(const void* h, const float* ghd, const float* gho, const float* gwd,
                                  const float* gwo, int B, int Hin, int Win, int C, int dtype, float* sumsq,
                                  int sumsq_cap, int* sumsq_used, void* stream) {
  if (sumsq_used) *sumsq_used = 0;
  if (!h || !ghd || !gho || !gwd || !gwo || !sumsq || !sumsq_used || B <= 0 || Hin <= 0 || Win <= 0) return DGV2_EINVAL;
  if (dtype != DGV2_BF16 || (C & 7) || C <= 0 || 256 % (C >> 3) != 0) return DGV2_ENOTSUP;
  if (!aligned16(h)) return DGV2_EINVAL;
  constexpr int RS = 8;
  const int cols = 256 / (C >> 3);
  dim3 grid((Win + cols - 1) / cols, (Hin + RS - 1) / RS, B);
  const int64_t nb = (int64_t)grid.x * grid.y * grid.z;
  if (nb > sumsq_cap) return DGV2_ENOTSUP;
  *sumsq_used = (int)nb;
  up2_lag_sumsq_kernel<RS><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)h, ghd, gho, gwd, gwo, Hin, Win, C, sumsq);
  DGV2_RETURN_LAST();
}
