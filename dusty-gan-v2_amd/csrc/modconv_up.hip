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
  int B, P, Wout, Hin, Win;   // output pixels P = Hout * Wout; T is [B, 32, Hin * Win]
  int Ks;                     // w is the per-sample MFMA image [B][Ks/16][2][32 o][8 k] (dgv2_modconv_up_t)
  int samples_per_block;
  const int* idx_h;           // [Hout][2] low-res rows of the two taps, coef_h [Hout][2]
  const float* coef_h;
  const int* idx_w;           // [Wout][2]
  const float* coef_w;
  const float* bias;
  int act;
  float alpha, scale;
  float* sumsq;
#ifdef DGV2_ABLATE   // benchmarking builds only (make ABLATE=1): wrong results by design, never in the shipped library
  int ablate;        // DGV2_MU_ABLATE: 1 skip the per-sample DMA, 2 skip the MFMA loop, 4 skip epilogue math + stores,
                     // 8 skip the barrier, 16 skip the T-window DMA only
#endif
};

constexpr int UP16 = 4;       // K-steps of the up-sampling part: 2 low-res rows x 32 window columns

template <int KS16, bool LRELU>   // Ks / 16; leaky ReLU or no activation
__global__ __launch_bounds__(512, 1) void modconv_up_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ tcm,
                                                            const bf16_t* __restrict__ xs, const bf16_t* __restrict__ w,
                                                            MUGeom g) {
  constexpr int O = 32;
  constexpr int KT = KS16 + UP16;
  constexpr int WBUF = KS16 * 64;                   // 16-byte slots of one sample's weights: [KS16][2][32 o]
  constexpr int TBUF = UP16 * 64;                   // ... of one wave's T window: [UP16][2][32 o]
  constexpr int NW = WBUF / 512;                    // weight DMA pieces per thread (KS16 % 8 == 0)
  constexpr int TOFF = 2 * WBUF;                    // T windows behind the two weight buffers: [2][8 waves][TBUF]
  extern __shared__ __attribute__((aligned(16))) uint4 lds_w[];
  const unsigned lds_off = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_w;   // LDS byte address

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: everything derived from it is wave-uniform
  const int n = lane & 31, kg = lane >> 5;
  const int p0 = blockIdx.x * 256 + wave * 32;      // P % 32 == 0: a wave is all live or all dead
  const int b0 = blockIdx.y * g.samples_per_block;
  const int b1 = min(b0 + g.samples_per_block, g.B);
  const bool live = p0 < g.P;
  const int pw = live ? p0 : 0;                     // a dead wave computes pixel tile 0 and stores nothing
  const int px = pw + n;
  const int Y = pw / g.Wout, X0 = pw - Y * g.Wout;  // Wout % 32 == 0: the wave's pixels are one row segment
  const int winbase = floormod((X0 >> 1) - 8, g.Win);

  const int iy[2] = {g.idx_h[2 * Y], g.idx_h[2 * Y + 1]};
  const float wy[2] = {g.coef_h[2 * Y], g.coef_h[2 * Y + 1]};
  int ix0 = g.idx_w[2 * (X0 + n)], ix1 = g.idx_w[2 * (X0 + n) + 1];
  float wx0 = g.coef_w[2 * (X0 + n)], wx1 = g.coef_w[2 * (X0 + n) + 1];
  float bias_n = g.bias ? g.bias[n] : 0.f;
  // per-lane source offsets of the wave's T window: piece s, lane (kg, o = n): 8 low-res pixels of channel o
  int toff[UP16];
#pragma unroll
  for (int s = 0; s < UP16; ++s) {
    int c = winbase + 8 * (2 * (s & 1) + kg);
    c = c >= g.Win ? c - g.Win : c;
    toff[s] = (((iy[s >> 1] * g.Win + c) >> 3) * O + n) * 8;   // T is [B][Hin*Win/8][32 o][8 px]: 1 KB per piece
  }

  // LDS slot L = kc * 64 + half * 32 + o  <-  image slot L of sample b: every piece is one contiguous 1 KB (a piece
  // gathered from 64 different lines costs the address path 8x the cycles; with eight such pieces per wave and
  // sample the waves queued at ISSUE and the transfer did not overlap the MFMA loop).  Piece q of a sample:
  // q < NW weights, then the wave's T window.
  auto dma_piece = [&](int q, int b, int buf) {
#ifdef DGV2_ABLATE
    if ((g.ablate & 1) || (q >= NW && (g.ablate & 16))) return;
#endif
    if (q < NW)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(w + ((int64_t)b * WBUF + tid + q * 512) * 8),
                                       (lds_void_t*)(lds_w + buf * WBUF + q * 512 + wave * 64), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(tcm + (int64_t)b * O * g.Hin * g.Win + toff[q - NW]),
                                       (lds_void_t*)(lds_w + TOFF + (buf * 8 + wave) * TBUF + (q - NW) * 64), 16, 0, 0);
  };
  constexpr int NQ = NW + UP16;

  // (Measured on one box and dropped: the first sample's DMA issued ahead of the PE loads, +4 %; PE loads hand-issued
  // and awaited one by one in the first sample's K-steps, no gain.)
  uint4 bf[KT];
  // ---- B operands (column n = pixel, k = 8 * kg + i), registers for the whole walk: PE, then U ----
#pragma unroll
  for (int kc = 0; kc < KS16; ++kc)
    bf[kc] = *reinterpret_cast<const uint4*>(xs + (int64_t)px * g.Ks + kc * 16 + kg * 8);
  if (b0 < b1) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) dma_piece(q, b0, 0);
  }
  {
#pragma unroll
    for (int s = 0; s < UP16; ++s) {
      // K index of element j: low-res row tap r = s / 2, window column 8 * (2 * (s % 2) + kg) + j
      vec16<bf16_t> u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int c = winbase + 8 * (2 * (s & 1) + kg) + j;
        c = c >= g.Win ? c - g.Win : c;
        const float wx = (c == ix0 ? wx0 : 0.f) + (c == ix1 ? wx1 : 0.f);
        u.set(j, wy[s >> 1] * wx);
      }
      bf[KS16 + s] = u.raw;
    }
  }
  // ---- bias as one more K-step: A[o][k = 0, 1] = (hi, lo) halves of bias * gain, B[k = 0, 1][p] = 1 (the factor
  //      c * gain itself arrives inside T and the weight image, dgv2_modconv_up_t) ----
  const float gain = LRELU ? g.scale * 0.5f * (1.f + g.alpha) : 1.f;   // see the epilogue
  bf16x8 bias_a, ones_b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bias_a[e] = (bf16_t)0.f;
    ones_b[e] = (bf16_t)(e < 2 ? 1.f : 0.f);
  }
  if (kg == 0) {
    const float bv = bias_n * gain;
    bias_a[0] = (bf16_t)bv;
    bias_a[1] = (bf16_t)(bv - (float)bias_a[0]);
  }

  float ss = 0.f;
  const float lr_k = (1.f - g.alpha) / (1.f + g.alpha);
  // The epilogue of sample b - 1 (exchange, leaky ReLU, statistic, bf16 pack, two 16-byte stores: ~6 vector
  // instructions per output value) rides in the MFMA loop of sample b, a few instructions per K-step: an MFMA holds the
  // vector issue for 8 of its 32 cycles, so the loop has room for 6 per step and the epilogue costs no time of its own.
  // Two accumulator sets alternate (the sample loop is unrolled by two), so nothing is copied between samples.
  // part j = 0, 2: exchange accumulator groups j, j + 1 in place: this lane then owns channels [8 j + 8 kg, +8) in
  // registers 4 j .. 4 j + 7; part 1, 3: activation, statistic, pack, store of those 8 channels.
  auto epi_part = [&](f32x16& v, int part, int bprev) {
#ifdef DGV2_ABLATE
    if (g.ablate & 4) {
      if (part == 0 && v[0] == 123.456f) y[0] = (bf16_t)v[3];   // keeps the accumulators alive
      return;
    }
#endif
    const int j = part & ~1;
    if (!(part & 1)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[4 * j + r]), __float_as_uint(v[4 * (j + 1) + r]),
                                                   false, false);
        v[4 * j + r] = __uint_as_float(sw[0]);            // kg = 0: own group j      | kg = 1: partner's group j + 1
        v[4 * (j + 1) + r] = __uint_as_float(sw[1]);      // kg = 0: partner's group j | kg = 1: own group j + 1
      }
    } else {
      // per PAIR of values: two v_fma_f32 (the leaky ReLU, see below), v_cvt_pk_bf16_f32, v_dot2c_f32_bf16 (statistic
      // of the ROUNDED pair) = 2 instructions per value.  Leaky ReLU: with s = scale (1 + alpha) / 2 folded into T, the
      // weights and the bias, act(f) scale = max(f, alpha f) scale = f' + k |f'|, f' = s f, k = (1 - alpha) / (1 + alpha):
      // ONE fma with a free |.| modifier.  (Measured on one box: fmaxf() form 114 us, mul + fma 97, this 8x-leaner form
      // issued as v_pk_mul + asm v_max: 121 -- the asm operands cost copies.)
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
      union { uint4 raw; bf16x2 p[4]; } o;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        float f0 = v[4 * j + e], f1 = v[4 * j + e + 1];
        if (LRELU) {
          f0 = fmaf(fabsf(f0), lr_k, f0);
          f1 = fmaf(fabsf(f1), lr_k, f1);
        }
        o.p[e >> 1] = bf16x2{(bf16_t)f0, (bf16_t)f1};
        ss = __builtin_amdgcn_fdot2_f32_bf16(o.p[e >> 1], o.p[e >> 1], ss, false);
      }
      if (live) *reinterpret_cast<uint4*>(y + ((int64_t)bprev * g.P + px) * O + 8 * j + 8 * kg) = o.raw;
    }
  };

  auto step = [&](auto has_prev, f32x16& acc, f32x16& prev, int buf, int b) {
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

    // A fragments: row m = lane % 32 = output channel, k = 8 * (lane / 32) + i  <-  slot kc * 64 + kg * 32 + n.
    // The reads are issued as asm (ring of RD registers, PF reads in flight, explicit lgkmcnt): left to the compiler,
    // every ds_read that follows an LDS-DMA in program order is preceded by `s_waitcnt vmcnt(0)` (it cannot prove the
    // read does not alias the DMA's destination), which parks the whole MFMA loop behind the NEXT sample's transfer.
#ifdef DGV2_ABLATE
    if (g.ablate & 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) dma_piece(q, bn, buf ^ 1);
      return;
    }
#endif
    {
      constexpr int PF = 4, RD = 6;
      const unsigned abase = lds_off + (unsigned)(buf * WBUF + kg * 32 + n) * 16u;
      const unsigned tbase = lds_off + (unsigned)(TOFF + (buf * 8 + wave) * TBUF + kg * 32 + n) * 16u;
      u32x4 a[RD];
#define DGV2_DS_READ(dst, kc)                                                                                    \
  do {                                                                                                           \
    if ((kc) < KS16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(abase), "n"(((kc) % KS16) * 1024)); \
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(tbase), "n"((((kc) >= KS16 ? (kc) - KS16 : 0)) * 1024)); \
  } while (0)
#define DGV2_LGKM_WAIT(dst, cnt) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "n"(cnt))
#pragma unroll
      for (int kc = 0; kc < PF; ++kc) DGV2_DS_READ(a[kc % RD], kc);
      // the chain starts from the bias: no zero fill of the accumulators
      {
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bias_a, ones_b, z, 0, 0, 0);
      }
#pragma unroll
      for (int kc = 0; kc < KT; ++kc) {
        // reads return in order: all but the (issued - kc - 1) youngest are back
        const int inflight = (kc + PF <= KT ? PF : KT - kc) - 1;
        switch (inflight) {
          case 3: DGV2_LGKM_WAIT(a[kc % RD], 3); break;
          case 2: DGV2_LGKM_WAIT(a[kc % RD], 2); break;
          case 1: DGV2_LGKM_WAIT(a[kc % RD], 1); break;
          default: DGV2_LGKM_WAIT(a[kc % RD], 0); break;
        }
        union { u32x4 u; bf16x8 v; } ua;
        union { uint4 u; bf16x8 v; } ub;
        ua.u = a[kc % RD];
        ub.u = bf[kc];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
        if (kc + PF < KT) DGV2_DS_READ(a[(kc + PF) % RD], kc + PF);   // slot last read two MFMAs ago
        // riders of this K-step: the previous sample's epilogue (steps 1..12, stores at 6 and 12), then the next
        // sample's DMA pieces (steps 14, 16, ..: behind the stores in issue order)
        if constexpr (HP) {
          if (kc == 1) epi_part(prev, 0, b - 1);
          if (kc == 4) epi_part(prev, 1, b - 1);
          if (kc == 7) epi_part(prev, 2, b - 1);
          if (kc == 10) epi_part(prev, 3, b - 1);
        }
        if (kc >= 14 && ((kc - 14) & 1) == 0 && (kc - 14) / 2 < NQ) dma_piece((kc - 14) / 2, bn, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef DGV2_DS_READ
#undef DGV2_LGKM_WAIT
    }
  };

  if (b0 < b1) {
    f32x16 accA, accB;
    step(std::false_type{}, accA, accB, 0, b0);
    int b = b0 + 1;
    for (; b + 1 < b1; b += 2) {
      step(std::true_type{}, accB, accA, (b - b0) & 1, b);
      step(std::true_type{}, accA, accB, (b + 1 - b0) & 1, b + 1);
    }
    if (b < b1) {
      step(std::true_type{}, accB, accA, (b - b0) & 1, b);
#pragma unroll
      for (int part = 0; part < 4; ++part) epi_part(accB, part, b1 - 1);
    } else {
#pragma unroll
      for (int part = 0; part < 4; ++part) epi_part(accA, part, b1 - 1);
    }
  }
  if (g.sumsq) {
    __shared__ float red[16];
    const float s = block_sum(live ? ss : 0.f, red);
    if (tid == 0) g.sumsq[blockIdx.y * gridDim.x + blockIdx.x] = s;
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
                                                           int I, int koff, int Ks) {
  constexpr int O = 32, Ka = KA16 * 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = lane & 31, kg = lane >> 5;
  const int b = blockIdx.y;
  if (blockIdx.x == 0 && wimg) {
    // image slot L = kc * 64 + half * 32 + o  <-  w[b][o][koff + kc * 16 + half * 8 .. + 8]
    const int slots = (Ks >> 4) * 64;
    for (int L = tid; L < slots; L += 512) {
      const int o = L & 31, half = (L >> 5) & 1, kc = L >> 6;
      vec16<bf16_t> v;
      v.raw = *reinterpret_cast<const uint4*>(w + ((int64_t)b * O + o) * I + koff + kc * 16 + half * 8);
      const float c = (row_scale ? row_scale[o] : 1.f) * gain;
      if (c != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v.set(e, v.get(e) * c);
      }
      *reinterpret_cast<uint4*>(wimg + ((int64_t)b * slots + L) * 8) = v.raw;
    }
  }
  const int p0 = blockIdx.x * 256 + wave * 32;
  if (p0 >= Plow) return;
  const bf16_t* hp = h + ((int64_t)b * Plow + p0 + n) * Ka + kg * 8;
  const bf16_t* wp = w + ((int64_t)b * O + n) * I + kg * 8;
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
  // lane (o = n, kg) holds pixels 8j + 4kg .. +3 (j = 0..3) -> two runs of 8 consecutive pixels (= two units)
  bf16_t* dst = tcm + (((int64_t)b * Plow + p0) >> 3) * (O * 8) + n * 8;
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
    *reinterpret_cast<uint4*>(dst + (j + kg) * (O * 8)) = o.raw;
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

// y[b,p,:32] = act( up2(T)[b,p,:] + sum_{k<Ks} xs[p,k] W_s[b,:,k] + bias )   (bf16 in / out), T and W_s carrying the
//   layer's input-magnitude factor row_scale[o] times gain = scale (1 + alpha) / 2 for act 3, 1 for act 0 (this entry
//   puts the same gain on the bias; the leaky ReLU is then one fma, see the kernel):
//   t [B,Hin*Win/8,32,8] = row_scale * gain * W_a . h at the previous level's resolution in 8-pixel units and
//   wimg [B,Ks/16,2,32,8] = the PE columns row_scale * gain * W_s of the prepared per-sample weights as the MFMA image,
//   both from dgv2_modconv_up_t; up2 by the
//   two-tap tables idx/coef [Hout][2], [Wout][2] (native.ResampleSpec.tables of the block's up-2 Resample, zero-padded to
//   two taps); xs [Hout*Wout, Ks] batch-shared PE.
//   Contract on the tables (the caller checks it once per table set, they live on the device): Wout % 32 == 0,
//   Win % 8 == 0, Win >= 32, and for every output column X both W taps lie in the aligned 32-column window
//   [(X & ~31) / 2 - 8, +32) mod Win (true for every up-2 FIR with at most 4 taps).  Ks in {512}.
//   sumsq: one partial per block (of the stored, bf16-rounded values).
extern "C" int dgv2_modconv_up_fwd(void* y, const void* t, const void* xs, const void* wimg, int B, int Hout, int Wout,
                                   int Hin, int Win, int Ks, int O, const int* idx_h, const float* coef_h,
                                   const int* idx_w, const float* coef_w, const float* bias, int act, float alpha,
                                   float scale, int dtype, float* sumsq, int sumsq_cap, int* sumsq_used, void* stream) {
  if (sumsq_used) *sumsq_used = 0;
  if (!y || !t || !xs || !wimg || !idx_h || !coef_h || !idx_w || !coef_w || B <= 0 || Hout <= 0 || Wout <= 0) return DGV2_EINVAL;
  if (dtype != DGV2_BF16 || (act != 0 && act != 3) || O != 32 || Ks != 512 || (Wout & 31) || (Win & 7) || Win < 32 ||
      Hin <= 0 || (act == 3 && !(scale > 0.f)))
    return DGV2_ENOTSUP;
  if (!aligned16(y) || !aligned16(t) || !aligned16(xs) || !aligned16(wimg)) return DGV2_EINVAL;
  const int P = Hout * Wout;
  MUGeom g{B, P, Wout, Hin, Win, Ks, 1, idx_h, coef_h, idx_w, coef_w, bias, act, alpha, scale, sumsq};
#ifdef DGV2_ABLATE
  g.ablate = getenv("DGV2_MU_ABLATE") ? atoi(getenv("DGV2_MU_ABLATE")) : 0;
#endif
  const int tiles = (P + 255) / 256;
  int nsplit = (256 + tiles - 1) / tiles;          // one resident block per CU (the B fragments fill the registers)
  nsplit = nsplit < 1 ? 1 : (nsplit > B ? B : nsplit);
  g.samples_per_block = (B + nsplit - 1) / nsplit;
  nsplit = (B + g.samples_per_block - 1) / g.samples_per_block;
  if (g.sumsq && sumsq_used && tiles * nsplit <= sumsq_cap) *sumsq_used = tiles * nsplit;
  else g.sumsq = nullptr;
  constexpr int KS16 = 32;
  const size_t lds = sizeof(uint4) * (2 * KS16 * 64 + 2 * 8 * UP16 * 64);
  auto kern = act == 3 ? modconv_up_kernel<KS16, true> : modconv_up_kernel<KS16, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid(tiles, nsplit);
  kern<<<grid, 512, lds, (hipStream_t)stream>>>((bf16_t*)y, (const bf16_t*)t, (const bf16_t*)xs, (const bf16_t*)wimg, g);
  DGV2_RETURN_LAST();
}

// The low-resolution part of the commuted level-input conv and the operand images of dgv2_modconv_up_fwd:
//   tcm [B,Plow/8,32,8]:  T[b][o][p] = f[o] sum_{c<Ka} w[b][o][c] h[b][p][c] in 8-pixel units (unit u, channel o: pixels
//                         8u..8u+7), f[o] = row_scale[o] (1 if NULL) * gain
//   wimg [B,Ks/16,2,32,8] (or NULL): f[o] w[b][o][koff + 16 kc + 8 half + j] at [b][kc][half][o][j]
// h [B,Plow,Ka], w [B,O,I] prepared per-sample weights (bf16); O = 32, Ka in {64, 128}, Plow % 32 == 0, Ks % 16 == 0.
// replaces: the xa columns of the ModConv2d contraction, gans/models/ops/style.py:105-118.
extern "C" int dgv2_modconv_up_t(void* tcm, void* wimg, const void* h, const void* w, const float* row_scale, float gain,
                                 int B, int Plow, int Ka, int Ks, int O, int I, int koff, int dtype, void* stream) {
  if (!tcm || !h || !w || B <= 0 || Plow <= 0) return DGV2_EINVAL;
  if (dtype != DGV2_BF16 || O != 32 || (Ka != 64 && Ka != 128) || (Plow & 31) || (I & 7) || I < Ka ||
      (wimg && ((Ks & 15) || (koff & 7) || koff + Ks > I)))
    return DGV2_ENOTSUP;
  if (!aligned16(tcm) || !aligned16(h) || !aligned16(w) || (wimg && !aligned16(wimg))) return DGV2_EINVAL;
  dim3 grid((Plow + 255) / 256, B);
  if (Ka == 64)
    modconv_up_t_kernel<4><<<grid, 512, 0, (hipStream_t)stream>>>((bf16_t*)tcm, (bf16_t*)wimg, (const bf16_t*)h,
                                                                 (const bf16_t*)w, row_scale, gain, Plow, I, koff, Ks);
  else
    modconv_up_t_kernel<8><<<grid, 512, 0, (hipStream_t)stream>>>((bf16_t*)tcm, (bf16_t*)wimg, (const bf16_t*)h,
                                                                 (const bf16_t*)w, row_scale, gain, Plow, I, koff, Ks);
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
