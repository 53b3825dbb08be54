// Level-input modulated 1x1 conv of the generator on the batch-shared positional encoding -- the
// kernel the top pyramid levels spend their time in:
//   y[b,p,o] = act( sum_{k<Ka} xa[b,p,k] w[b,o,k] + sum_{k<Ks} xs[p,k] w[b,o,Ka+k] + bias[o] ) * scale
// reference: torch.cat([h, pe]) + ModConv2d contraction + FusedLeakyReLU, gans/models/dusty_v2.py:153-162,
// gans/models/ops/style.py:105-118 (same contract as dgv2_bmm_nn_cat, include/dgv2.h).
//
// At level 4 (32768 pixels, Ka = 64, Ks = 512, O = 32) the contraction is 77 GFLOP per 64-sample batch
// against 0.4 GB of compulsory HBM traffic (xa in, y out) -- provided the 33 MB PE operand is NOT re-read
// per sample.  The generic GEMM re-streams it through L2 for every sample (2.1 GB) and is bound by that.
// Here the roles are turned round: a block owns a tile of pixels and walks the SAMPLES.
//   * PE operand: each wave keeps the MFMA B-fragments of its 16*NFW pixels x all Ks channels in REGISTERS
//     for the whole walk (loaded once, straight from global in fragment shape: one 16-byte K-chunk per lane).
//   * xa operand: per sample, fragment-shaped global loads straight to registers, prefetched one sample ahead.
//   * per-sample weights w[b] (O x (Ka+Ks), the only operand all waves share): two LDS buffers in the
//     swizzled [kchunk][o][64 B] image of gemm_core.h, filled one sample ahead by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers -- the PE fragments own the register file -- with the
//     swizzle applied on the source address), one raw barrier + counted vmcnt per sample.
// LDS traffic is A-fragments only; HBM traffic is xa + y (+ PE once per block).
#include <type_traits>

#include "gemm_core.h"

namespace {

#ifdef DGV2_ABLATE
#define MP_ABL (g.ablate)
#else
#define MP_ABL 0
#endif

struct MPGeom {
  int B, P, Ka, Ks, O, I;
  int samples_per_block;
#ifdef DGV2_ABLATE   // benchmarking builds only (make ABLATE=1): wrong results by design, never in the shipped library
  int ablate;        // DGV2_MP_ABLATE: 1 skip stores, 2 skip MFMA loop, 4 skip weight staging, 8 skip xa loads
#endif
  const float* bias;
  int act;
  float alpha, scale;
  float* sumsq;   // optional: one partial sum of squares of the stored outputs per block
  const float* row_scale;   // optional fp32 [O]: y = act(acc * row_scale[o] + bias[o])
  // HEAD instances: the two 1-channel output heads of the level read this layer's output -- their contraction
  //   hd[b, p, j] = sum_o y[b, p, o] hw[b, j, o]   (j < 2, on the stored bf16 values, fp32 sum)
  // leaves from this epilogue instead of a pass of its own over y (reference: Head.forward, dusty_v2.py:30-57,171-178)
  const bf16_t* hw;   // [B, 2, O] the heads' prepared per-sample weights
  float* hd;          // [B, P, 2]
  // ACTBWD instances (this launch is a DATA GRADIENT whose result is the gradient of an upstream layer's activation
  // output): that layer's activation backward rides in the epilogue --
  //   t = (yref > 0 ? g : g * alpha) * scale on the bf16-rounded g,  gb[o] += bf16(t),  stored: bf16(t * up_scale[o])
  // bit for bit what dgv2_bias_act_bwd_rs makes of the g this launch would have stored (csrc/bias_act.hip)
  const bf16_t* yref;       // [B, P, O] the upstream layer's forward OUTPUT
  const float* up_scale;    // [O] its input-magnitude factor c[o]
  float* gbpart;            // [gridDim.y * tiles, O] per-block column sums of bf16(t): the upstream bias gradient's partials
};

typedef __attribute__((ext_vector_type(4))) unsigned mp_u32x4;

// xa fragments of one sample: fragment-shaped global loads straight to registers, issued as asm -- with an LDS-DMA in
// flight the first use of an ordinary load is preceded by `s_waitcnt vmcnt(0)` (see the kernel); the caller waits with a
// counted vmcnt before it touches `dst`.
template <int NFW, int KA>
__device__ __forceinline__ void mp_issue_xa(mp_u32x4 (&dst)[NFW][KA > 0 ? KA : 1], const bf16_t* __restrict__ xa, int b,
                                            int P, int Ka, int p0, int lr, int lc) {
#pragma unroll
  for (int nf = 0; nf < NFW; ++nf) {
    const int px = min(p0 + nf * 16 + lr, P - 1);
#pragma unroll
    for (int kc = 0; kc < KA; ++kc) {
      const bf16_t* src = xa + ((int64_t)b * P + px) * Ka + kc * 32 + lc * 8;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[nf][kc]) : "v"(src) : "memory");
    }
  }
}

// head weights of one sample: per lane the 8 channels it will hold of each fragment pair after pack_pair_bf16, both heads
template <int MF>
__device__ __forceinline__ void mp_issue_hw(mp_u32x4 (&dst)[MF / 2][2], const bf16_t* __restrict__ hw, int b, int O, int lc) {
  const int co = (lc & 1) ? 16 + 4 * (lc - 1) : 4 * lc;
#pragma unroll
  for (int pr = 0; pr < MF / 2; ++pr)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bf16_t* src = hw + ((int64_t)b * 2 + j) * O + pr * 32 + co;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[pr][j]) : "v"(src) : "memory");
    }
}

// (ACTBWD) the upstream layer's output at the 8 channels a lane stores per fragment pair, one sample
template <int NFW, int MF>
__device__ __forceinline__ void mp_issue_yref(mp_u32x4 (&dst)[NFW][MF / 2], const bf16_t* __restrict__ yref, int b, int P,
                                              int O, int p0, int lr, int ch0) {
#pragma unroll
  for (int nf = 0; nf < NFW; ++nf) {
    const int px = min(p0 + nf * 16 + lr, P - 1);
#pragma unroll
    for (int pr = 0; pr < MF / 2; ++pr) {
      const bf16_t* src = yref + ((int64_t)b * P + px) * O + ch0 + pr * 32;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[nf][pr]) : "v"(src) : "memory");
    }
  }
}

__device__ __forceinline__ float dot8_bf16(const uint4& a, const mp_u32x4& b) {
  const unsigned aw[4] = {a.x, a.y, a.z, a.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s = fmaf(__uint_as_float(aw[i] << 16), __uint_as_float(b[i] << 16), s);
    s = fmaf(__uint_as_float(aw[i] & 0xffff0000u), __uint_as_float(b[i] & 0xffff0000u), s);
  }
  return s;
}

// MF = O / 16, NFW = 16-pixel fragments per wave, KA = Ka / 32, KS = Ks / 32 (compile-time: register arrays)
template <int MF, int NFW, int KA, int KS, bool HEAD = false, bool ACTBWD = false>
__global__ __launch_bounds__(512, 2) void modconv_pe_fwd_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ xa,
                                                                const bf16_t* __restrict__ xs,
                                                                const bf16_t* __restrict__ w, MPGeom g) {
  constexpr int O = MF * 16, KC = KA + KS, I = KC * 32;
  constexpr int NSLOT = O * KC * 4;                 // 16-byte slots of one sample's weights
  constexpr int NW = (NSLOT + 511) / 512;
  constexpr int WBUF = NW * 512;                    // slots per weight buffer (padded to whole DMA pieces)
  constexpr int TP = 8 * 16 * NFW;                  // pixels per block
  extern __shared__ __attribute__((aligned(16))) uint4 lds_w[];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lc = lane >> 4;
  // slabs of one pixel tile are neighbouring blocks: they run together and share the tile's xa through L2
  const int slabs = g.O / O;
  const int p0 = (blockIdx.x / slabs) * TP + wave * 16 * NFW;
  const int o_base = (blockIdx.x % slabs) * O;   // output channels come in slabs of O = MF*16 (g.O = all of them)
  const int b0 = blockIdx.y * g.samples_per_block;
  const int b1 = min(b0 + g.samples_per_block, g.B);

  // ---- PE fragments: registers for the whole block ----
  constexpr int KSR = KS > 0 ? KS : 1;
  uint4 pe[NFW][KSR];
#pragma unroll
  for (int nf = 0; nf < NFW; ++nf) {
    const int px = min(p0 + nf * 16 + lr, g.P - 1);   // clamped: pixels past the end are computed, never stored
#pragma unroll
    for (int kc = 0; kc < KS; ++kc)
      pe[nf][kc] = *reinterpret_cast<const uint4*>(xs + (int64_t)px * g.Ks + kc * 32 + lc * 8);
  }

  float bias_r[MF][4];            // this lane's output channels: mf*16 + lc*4 + r
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_r[mf][r] = g.bias ? g.bias[o_base + mf * 16 + lc * 4 + r] : 0.f;
  float cs_r[MF][4];              // per-channel output scale (1 when absent)
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs_r[mf][r] = g.row_scale ? g.row_scale[o_base + mf * 16 + lc * 4 + r] : 1.f;
  constexpr int KAR = KA > 0 ? KA : 1;
  typedef mp_u32x4 u32x4;
  u32x4 xr[2][NFW][KAR];            // xa fragments of samples b, b+1
  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef __attribute__((address_space(1))) const void gbl_void_t;
  // LDS-DMA of sample b's weights into buffer `buf`: piece j of wave `wave` lands at slots [j*512 + wave*64, +64);
  // LDS slot L = (kc*O + r)*4 + p holds logical chunk p ^ ((r>>2)&3) of row r (swizzle on the source side)
  auto dma_w = [&](int b, int buf) {
    const bf16_t* wb = w + ((int64_t)b * g.O + o_base) * I;   // this block's slab of O output channels
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int L = min(tid + j * 512, NSLOT - 1);   // surplus lanes of the last piece land in the pad
      const int chp = (L & 3) ^ ((L >> 4) & 3), r = (L >> 2) % O, kc = (L >> 2) / O;
      const bf16_t* src = wb + r * I + kc * 32 + chp * 8;
      uint4* dst = lds_w + buf * WBUF + j * 512 + wave * 64;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
    }
  };
  u32x4 hwr[2][HEAD ? MF / 2 : 1][2];   // HEAD: head weights of samples b, b+1 (this lane's channels)
  auto issue_x = [&](auto slot, int b) {
    if constexpr (KA > 0)
      mp_issue_xa<NFW, KA>(xr[decltype(slot)::value], xa, min(b, b1 - 1), g.P, g.Ka, p0, lr, lc);   // tail: re-reads the last sample
    if constexpr (HEAD) mp_issue_hw<MF>(hwr[decltype(slot)::value], g.hw, min(b, b1 - 1), O, lc);
  };
  const int aswz = lc ^ ((lr >> 2) & 3);
  // ACTBWD: the upstream output at the 8 channels this lane stores per fragment pair (pack_pair_bf16's channel offset),
  // requested with the same hand-issued loads at the start of the sample's step, consumed in its epilogue
  const int co8 = (lc & 1) ? 16 + 4 * (lc - 1) : 4 * lc;
  u32x4 yr[2][ACTBWD ? NFW : 1][ACTBWD ? MF / 2 : 1];   // samples b, b + 1: prefetched one sample ahead like the xa fragments
  float gbs[ACTBWD ? MF / 2 : 1][8];
  __shared__ float s_up[ACTBWD ? MF * 16 : 1];      // up_scale of this block's slab
  __shared__ float s_gb[ACTBWD ? 8 * MF * 16 : 1];  // the waves' column sums
  if constexpr (ACTBWD) {
#pragma unroll
    for (int pr = 0; pr < MF / 2; ++pr)
#pragma unroll
      for (int j = 0; j < 8; ++j) gbs[pr][j] = 0.f;
    for (int c = tid; c < MF * 16; c += 512) s_up[c] = g.up_scale[o_base + c];
    __syncthreads();
  }
  auto issue_yref = [&](auto slot, int b) {
    if constexpr (ACTBWD)   // tail: re-reads the last sample (as issue_x does)
      mp_issue_yref<NFW, MF>(yr[decltype(slot)::value], g.yref, min(b, b1 - 1), g.P, g.O, p0, lr, o_base + co8);
  };

  float ss = 0.f;
  // Hand-issued LDS reads and counted waits.  With an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` before every
  // ds_read that follows it in program order (it cannot prove the read does not alias the DMA's destination) and before
  // the first use of an ordinary global load: left to the compiler, a sample's MFMA loop waits for the NEXT sample's
  // weights and xa fragments, i.e. nothing overlaps.  Per wave and sample the vector-memory operations are issued in
  // the order [NW weight pieces of b+1] [NFW*KA xa loads of b+1] [stores of b]; they retire in that order.
  const unsigned lds_off = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_w;
  const unsigned abase = lds_off + (unsigned)(lr * 4 + aswz) * 16u;
  // stores a full wave certainly issues per sample (fragments with at least one live pixel); a partial tile drains
  int nlive = 0;
#pragma unroll
  for (int nf = 0; nf < NFW; ++nf) nlive += (p0 + nf * 16 < g.P) ? 1 : 0;
  const bool full_wave = __builtin_amdgcn_readfirstlane(nlive) == NFW;
#define MP_DS_READ(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(abase_s), "n"(off))
#define MP_LGKM_WAIT(dst, cnt) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "n"(cnt))

  // one sample: weights of sample b are in LDS buffer S, its xa fragments in ring slot S
  auto step = [&](auto slot, int b) {
    constexpr int S = decltype(slot)::value;
    // everything but the stores of the previous sample (the youngest operations) has retired: the weights and the xa
    // fragments of THIS sample have landed (first sample / partial tile: full drain)
    if (b == b0 || !full_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NFW * (MF / 2) + (HEAD ? NFW : 0)) : "memory");   // (+ the hd stores)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
      for (int kc = 0; kc < KA; ++kc) asm volatile("" : "+v"(xr[S][nf][kc]));
    if constexpr (HEAD) {
#pragma unroll
      for (int pr = 0; pr < MF / 2; ++pr)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(hwr[S][pr][j]));
    }
    if constexpr (ACTBWD) {
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
        for (int pr = 0; pr < MF / 2; ++pr) asm volatile("" : "+v"(yr[S][nf][pr]));
    }
    __builtin_amdgcn_s_barrier();   // all pieces landed; every wave is done with the other buffer
    asm volatile("" ::: "memory");
    if (!(MP_ABL & 4)) dma_w(min(b + 1, b1 - 1), S ^ 1);
    if (!(MP_ABL & 8)) issue_x(std::integral_constant<int, S ^ 1>{}, b + 1);
    issue_yref(std::integral_constant<int, S ^ 1>{}, b + 1);   // (ACTBWD) with the next sample's fragments: landed by its step

    f32x4 acc[MF][NFW];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!(MP_ABL & 2)) {
      // A fragments: read q = kc * MF + mf  <-  slot (kc * O + mf * 16 + lr) * 4 + aswz of buffer S
      constexpr int NR = KC * MF, PF = NR < 4 ? NR : 4, RD = 6;
      const unsigned abase_s = abase + (unsigned)(S * WBUF * 16);   // the 16-bit offset field covers one buffer
      u32x4 a[RD];
#pragma unroll
      for (int q = 0; q < PF; ++q) MP_DS_READ(a[q % RD], ((q / MF) * O + (q % MF) * 16) * 64);
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const int kc = q / MF, mf = q % MF;
        const int inflight = (q + PF <= NR ? PF : NR - q) - 1;
        switch (inflight) {
          case 3: MP_LGKM_WAIT(a[q % RD], 3); break;
          case 2: MP_LGKM_WAIT(a[q % RD], 2); break;
          case 1: MP_LGKM_WAIT(a[q % RD], 1); break;
          default: MP_LGKM_WAIT(a[q % RD], 0); break;
        }
        const uint4 av = make_uint4(a[q % RD][0], a[q % RD][1], a[q % RD][2], a[q % RD][3]);
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) {
          uint4 bf;
          if (kc < KA) {
            const u32x4 t = xr[S][nf][kc < KA ? kc : 0];
            bf = make_uint4(t[0], t[1], t[2], t[3]);
          } else {
            bf = pe[nf][(KS > 0 && kc >= KA) ? kc - KA : 0];
          }
          Mfma16<bf16_t>::run(acc[mf][nf], av, bf);
        }
        if (q + PF < NR) {
          const int qn = q + PF;
          MP_DS_READ(a[qn % RD], ((qn / MF) * O + (qn % MF) * 16) * 64);   // slot last read two reads ago
        }
      }
    }

    // epilogue: lane holds o = mf*16 + lc*4 + r at pixel nf*16 + lr; fragment pairs leave as 16-byte stores
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      const int px = p0 + nf * 16 + lr;
      const bool live = px < g.P && !((MP_ABL & 1) && acc[0][0][0] != 12345.678f);
      bf16_t* row = y + ((int64_t)b * g.P + px) * g.O + o_base;
      float hd0 = 0.f, hd1 = 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; mf += 2) {
        float va[4], vb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float fa = acc[mf][nf][r], fb = acc[mf + 1][nf][r];
          fa = fmaf(fa, cs_r[mf][r], bias_r[mf][r]);
          fb = fmaf(fb, cs_r[mf + 1][r], bias_r[mf + 1][r]);
          if (g.act == 3) {
            fa = (fa > 0.f ? fa : fa * g.alpha) * g.scale;
            fb = (fb > 0.f ? fb : fb * g.alpha) * g.scale;
          }
          va[r] = fa;
          vb[r] = fb;
        }
        uint4 pk;
        const int co = pack_pair_bf16(va, vb, lc, pk);   // all lanes take part in the exchange
        if constexpr (ACTBWD) {
          const unsigned gw_[4] = {pk.x, pk.y, pk.z, pk.w};
          unsigned ow[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned yw = yr[S][nf][mf / 2][i];
            float t[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float gv = __uint_as_float(h ? (gw_[i] & 0xffff0000u) : (gw_[i] << 16));
              const float yv = __uint_as_float(h ? (yw & 0xffff0000u) : (yw << 16));
              const float v = (yv > 0.f ? gv : gv * g.alpha) * g.scale;
              const bf16_t vb16 = (bf16_t)v;
              if (live) gbs[mf / 2][2 * i + h] += (float)vb16;      // the sum of what the separate pass stores unscaled
              t[h] = v * s_up[mf * 16 + co + 2 * i + h];
            }
            union { bf16_t e[2]; unsigned u; } o2;
            o2.e[0] = (bf16_t)t[0];
            o2.e[1] = (bf16_t)t[1];
            ow[i] = o2.u;
          }
          pk = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
        if (live) {
          *reinterpret_cast<uint4*>(row + mf * 16 + co) = pk;
          if (g.sumsq) ss += sumsq_bf16x8(pk);
        }
        if constexpr (HEAD) {
          hd0 += dot8_bf16(pk, hwr[S][mf / 2][0]);
          hd1 += dot8_bf16(pk, hwr[S][mf / 2][1]);
        }
      }
      if constexpr (HEAD) {
        // the four lc lanes of a pixel hold disjoint channel sets: fold them (all lanes take part), lane lc == 0 stores
        hd0 += __shfl_xor(hd0, 16, 64);
        hd1 += __shfl_xor(hd1, 16, 64);
        hd0 += __shfl_xor(hd0, 32, 64);
        hd1 += __shfl_xor(hd1, 32, 64);
        if (live && lc == 0) *reinterpret_cast<float2*>(g.hd + ((int64_t)b * g.P + px) * 2) = make_float2(hd0, hd1);
      }
    }
  };

  if (b0 < b1) {
    dma_w(b0, 0);
    issue_x(std::integral_constant<int, 0>{}, b0);
    issue_yref(std::integral_constant<int, 0>{}, b0);
  }
  for (int b = b0; b < b1; b += 2) {
    step(std::integral_constant<int, 0>{}, b);
    if (b + 1 < b1) step(std::integral_constant<int, 1>{}, b + 1);
  }
#undef MP_DS_READ
#undef MP_LGKM_WAIT
  // The last step issued one more round of xa loads (a re-read of the last sample) that nothing consumes.  They are
  // asm-issued: the compiler takes their destination registers for dead past the loop and would hand them to the
  // reduction below while the loads are still in flight -- a late return then overwrites a partial sum.  Drain, and keep
  // the registers allocated up to the drain.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
      for (int kc = 0; kc < KAR; ++kc) asm volatile("" : "+v"(xr[s2][nf][kc]));
  if constexpr (HEAD) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int pr = 0; pr < MF / 2; ++pr)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(hwr[s2][pr][j]));
  }
  if constexpr (ACTBWD) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
        for (int pr = 0; pr < MF / 2; ++pr) asm volatile("" : "+v"(yr[s2][nf][pr]));
  }
  if (g.sumsq) {
    __shared__ float red[16];
    const float s = block_sum(ss, red);
    if (tid == 0) g.sumsq[blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
  if constexpr (ACTBWD) {
    // fold the 16 pixel lanes (lr) of a wave, then the 8 waves: one row of partials per (sample split, pixel tile)
#pragma unroll
    for (int pr = 0; pr < MF / 2; ++pr)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = gbs[pr][j];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (lr == 0) s_gb[wave * (MF * 16) + pr * 32 + co8 + j] = v;
      }
    __syncthreads();
    const int tile = blockIdx.x / slabs, tiles = gridDim.x / slabs;
    for (int c = tid; c < MF * 16; c += 512) {
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) v += s_gb[wv * (MF * 16) + c];
      g.gbpart[((int64_t)blockIdx.y * tiles + tile) * g.O + o_base + c] = v;
    }
  }
}

template <int MF, int NFW, int KA, int KS, bool HEAD = false, bool ACTBWD = false>
int mp_launch(void* y, const void* xa, const void* xs, const void* w, MPGeom g, hipStream_t st, int sumsq_cap,
              int* sumsq_used, int64_t* part_rows = nullptr, int64_t part_cap = 0) {
  constexpr int TP = 8 * 16 * NFW;
  const int tiles = (g.P + TP - 1) / TP;
  // one resident block per CU (the PE fragments fill the register file): one round of blocks, as few sample
  // splits (= PE reloads) as that allows
  static const int target = getenv("DGV2_MP_BLOCKS") ? atoi(getenv("DGV2_MP_BLOCKS")) : 256;
  int nsplit = (target + tiles - 1) / tiles;
  nsplit = nsplit < 1 ? 1 : (nsplit > g.B ? g.B : nsplit);
  g.samples_per_block = (g.B + nsplit - 1) / nsplit;
  nsplit = (g.B + g.samples_per_block - 1) / g.samples_per_block;
  const int slabs = g.O / (MF * 16);   // host-checked: a whole number
  // with several slabs fewer sample splits are needed to fill the chip
  if (slabs > 1) {
    nsplit = (target + tiles * slabs - 1) / (tiles * slabs);
    nsplit = nsplit < 1 ? 1 : (nsplit > g.B ? g.B : nsplit);
    g.samples_per_block = (g.B + nsplit - 1) / nsplit;
    nsplit = (g.B + g.samples_per_block - 1) / g.samples_per_block;
  }
  dim3 grid(tiles * slabs, nsplit);
  if (ACTBWD) {   // rows of column-sum partials this launch writes; a query (part_cap == 0) launches nothing
    if (part_rows) *part_rows = (int64_t)tiles * nsplit;
    if (part_cap < (int64_t)tiles * nsplit) return part_cap == 0 ? 0 : DGV2_EINVAL;
  }
  if (g.sumsq && sumsq_used && tiles * nsplit * slabs <= sumsq_cap) *sumsq_used = tiles * nsplit * slabs;
  else g.sumsq = nullptr;
  constexpr size_t lds = sizeof(uint4) * 2 * (size_t)(((MF * 16) * (KA + KS) * 4 + 511) / 512 * 512);
  auto kern = modconv_pe_fwd_kernel<MF, NFW, KA, KS, HEAD, ACTBWD>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  kern<<<grid, 512, lds, st>>>((bf16_t*)y, (const bf16_t*)xa, (const bf16_t*)xs,
                                                             (const bf16_t*)w, g);
  return 0;
}

// gb[c] = sum_r part[r][c]: one block per channel, 256 threads split the rows
__global__ __launch_bounds__(256) void mp_gb_reduce_kernel(float* __restrict__ gb, const float* __restrict__ part, int rows,
                                                           int C) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s = 0.f;
#pragma unroll 4
  for (int k = threadIdx.x; k < rows; k += 256) s += part[(int64_t)k * C + c];
  s = block_sum(s, red);
  if (threadIdx.x == 0) gb[c] = s;
}

}  // namespace

// Data gradient of a PE-free modulated 1x1 layer (conv2 of a generator level: K -> K channels) FUSED with the activation
// backward of the layer that produced its input (conv1 of the level):
//   g[b,p,k]    = sum_o gy[b,p,o] wt[b,k,o]                      (rounded to bf16: what dgv2_modconv_pe_fwd stores)
//   t           = (yref[b,p,k] > 0 ? g : g * alpha) * scale
//   gpre[b,p,k] = bf16(t * up_scale[k]),     gb[k] = sum_{b,p} bf16(t)
// -- bit for bit the pair dgv2_modconv_pe_fwd (Ks = 0) + dgv2_bias_act_bwd_rs it replaces (one pass over the gradient
// instead of three; reference: the autograd chain of ModConv2d -> FusedLeakyReLU, gans/models/ops/style.py:105-118,
// fused_act.py:46-59).  gy [B,P,K], wt [B,K,K] (conv2's per-sample weights, transposed), yref [B,P,K] bf16; up_scale, gb
// fp32 [K]; scratch fp32 [scratch_elems] >= *rows_needed * K (call with scratch == NULL to query rows_needed; nothing is
// launched).  K in {32, 64} (generator levels 4 / 3); DGV2_ENOTSUP otherwise.
extern "C" int dgv2_modconv_pe_dgrad_actbwd(void* gpre, float* gb, float* scratch, int64_t scratch_elems, int64_t* rows_needed,
                                            const void* gy, const void* wt, const void* yref, const float* up_scale,
                                            float alpha, float scale, int B, int P, int K, int dtype, void* stream) {
  if (rows_needed) *rows_needed = 0;
  // (K = 128 / 256, levels 2 / 1: their instances have no registers left for the prefetched upstream outputs and the
  // column sums -- 27 / 44 spills; they gained 6 us as a single-buffered form and keep the two launches)
  if (dtype != DGV2_BF16 || (K != 32 && K != 64)) return DGV2_ENOTSUP;
  if (B <= 0 || P <= 0) return DGV2_EINVAL;
  const bool query = scratch == nullptr;
  if (!query && (!gpre || !gb || !gy || !wt || !yref || !up_scale || !aligned16(gpre) || !aligned16(gy) || !aligned16(wt) ||
                 !aligned16(yref)))
    return DGV2_EINVAL;
#ifdef DGV2_ABLATE
  MPGeom g{B, P, K, 0, K, K, 1, 0, nullptr, 0, alpha, scale, nullptr, nullptr, nullptr, nullptr};
#else
  MPGeom g{B, P, K, 0, K, K, 1, nullptr, 0, alpha, scale, nullptr, nullptr, nullptr, nullptr};
#endif
  g.yref = (const bf16_t*)yref;
  g.up_scale = up_scale;
  g.gbpart = scratch;
  hipStream_t st = (hipStream_t)stream;
  int64_t rows = 0;
  const int64_t cap = query ? 0 : scratch_elems / K;
  if (!query && cap < 1) return DGV2_EINVAL;
  int rc;
  if (K == 32) rc = mp_launch<2, 2, 1, 0, false, true>(gpre, gy, gy, wt, g, st, 0, nullptr, &rows, cap);
  else rc = mp_launch<4, 2, 2, 0, false, true>(gpre, gy, gy, wt, g, st, 0, nullptr, &rows, cap);
  if (rows_needed) *rows_needed = rows;
  if (rc || query) return rc;
  mp_gb_reduce_kernel<<<K, 256, 0, st>>>(gb, scratch, (int)rows, K);
  DGV2_RETURN_LAST();
}

// Same contract as dgv2_bmm_nn_cat (bf16 in / bf16 out) for the shapes of the two top generator levels:
// (Ka, Ks, O) = (64, 512, 32) (level-4 conv1) and, with Ks = 0 (xs unused), the PE-free shapes
// (64,0,32), (32,0,64), (128,0,64), (64,0,128), (32,0,32), (64,0,64), (128,0,128), (256,0,256), (256,0,128), (128,0,256) = conv0 /
// conv2 of levels 4 ... 1 and the data gradients.  Returns DGV2_EINVAL for anything else: callers fall back to
// dgv2_bmm_nn_cat.
extern "C" int dgv2_modconv_pe_fwd(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka,
                                   int Ks, int O, const float* bias, int act, float alpha, float scale, int dtype,
                                   void* stream) {
  return dgv2_modconv_pe_fwd_head(y, xa, xs, w, B, P, Ka, Ks, O, nullptr, bias, act, alpha, scale, dtype, nullptr, 0, nullptr,
                                  nullptr, nullptr, stream);
}

extern "C" int dgv2_modconv_pe_fwd_sq(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka,
                                      int Ks, int O, const float* row_scale, const float* bias, int act, float alpha,
                                      float scale, int dtype, float* sumsq, int sumsq_cap, int* sumsq_used, void* stream) {
  return dgv2_modconv_pe_fwd_head(y, xa, xs, w, B, P, Ka, Ks, O, row_scale, bias, act, alpha, scale, dtype, sumsq, sumsq_cap,
                                  sumsq_used, nullptr, nullptr, stream);
}

extern "C" int dgv2_modconv_pe_fwd_head(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka,
                                        int Ks, int O, const float* row_scale, const float* bias, int act, float alpha,
                                        float scale, int dtype, float* sumsq, int sumsq_cap, int* sumsq_used,
                                        const void* head_w, float* head_out, void* stream) {
  if (sumsq_used) *sumsq_used = 0;
  if ((!head_w) != (!head_out)) return DGV2_EINVAL;
  // head_w [B, 2, O] / head_out fp32 [B, P, 2]: the contraction of the level's two output heads on this layer's output,
  // from this epilogue (PE-free one-slab shapes (32, 32) and (64, 64): conv2 of levels 4 / 3; DGV2_ENOTSUP otherwise)
  if (head_w && !(Ks == 0 && ((Ka == 32 && O == 32) || (Ka == 64 && O == 64)) && aligned16(head_w) && aligned16(head_out)))
    return DGV2_ENOTSUP;
  if (!y || !w || (Ks > 0 && !xs) || (Ka > 0 && !xa) || B <= 0 || P <= 0) return DGV2_EINVAL;
  if (dtype != DGV2_BF16 || (act != 0 && act != 3)) return DGV2_EINVAL;
  if (!aligned16(y) || !aligned16(xa) || !aligned16(xs) || !aligned16(w)) return DGV2_EINVAL;
  if (Ks == 0) xs = xa;   // never dereferenced (KS = 0), keeps the pointer arithmetic defined
#ifdef DGV2_ABLATE
  static const int abl = getenv("DGV2_MP_ABLATE") ? atoi(getenv("DGV2_MP_ABLATE")) : 0;
  MPGeom g{B, P, Ka, Ks, O, Ka + Ks, 1, abl, bias, act, alpha, scale, sumsq, row_scale, (const bf16_t*)head_w, head_out};
#else
  MPGeom g{B, P, Ka, Ks, O, Ka + Ks, 1, bias, act, alpha, scale, sumsq, row_scale, (const bf16_t*)head_w, head_out};
#endif
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (Ka == 64 && Ks == 512 && O == 32) rc = mp_launch<2, 2, 2, 16>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  // level-3 conv1: two slabs of 32 output channels (a 64-channel slab's weights, 2 x 82 KB, do not fit the LDS)
  else if (Ka == 128 && Ks == 512 && O == 64) rc = mp_launch<2, 1, 4, 16>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ka == 256 && Ks == 512 && O == 128) rc = mp_launch<2, 1, 8, 16>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  // the PE-free layers of the two top levels and their data gradients: same sample walk, weights by LDS-DMA
  else if (Ks == 0 && Ka == 64 && O == 32) rc = mp_launch<2, 2, 2, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ks == 0 && Ka == 32 && O == 64) rc = mp_launch<4, 2, 1, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ks == 0 && Ka == 128 && O == 64) rc = mp_launch<4, 2, 4, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ks == 0 && Ka == 64 && O == 128) rc = mp_launch<8, 2, 2, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  // conv2 of level 2 (and its data gradient): one 128-pixel tile per block keeps acc + two xa slots within 128 VGPRs
  else if (Ks == 0 && Ka == 128 && O == 128) rc = mp_launch<8, 1, 4, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  // conv2 of level 1: two slabs of 128 output channels (a slab's weights: 64 KB per buffer)
  else if (Ks == 0 && Ka == 256 && O == 256) rc = mp_launch<8, 1, 8, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ks == 0 && Ka == 256 && O == 128) rc = mp_launch<8, 1, 8, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ks == 0 && Ka == 128 && O == 256) rc = mp_launch<8, 1, 4, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ks == 0 && Ka == 32 && O == 32 && head_w) rc = mp_launch<2, 2, 1, 0, true>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ks == 0 && Ka == 64 && O == 64 && head_w) rc = mp_launch<4, 2, 2, 0, true>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ks == 0 && Ka == 32 && O == 32) rc = mp_launch<2, 2, 1, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else if (Ks == 0 && Ka == 64 && O == 64) rc = mp_launch<4, 2, 2, 0>(y, xa, xs, w, g, st, sumsq_cap, sumsq_used);
  else return DGV2_EINVAL;
  if (rc) return rc;
  DGV2_RETURN_LAST();
}
