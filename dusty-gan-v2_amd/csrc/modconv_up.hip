// Level-input modulated 1x1 conv of the generator with the up-sampling COMMUTED past the contraction:
//   reference (gans/models/dusty_v2.py:153-162, ops/style.py:105-118):
//       y = act( c * ( W_a . up2(h)  +  W_s . PE ) + bias )        W = [W_a | W_s] per sample, up2 = ring-aware FIR
//   a 1x1 conv acts per pixel and the FIR per channel, so  W_a . up2(h) == up2( W_a . h ):  the xa part of the
//   contraction runs at a QUARTER of the pixels (t = W_a . h, a small batched GEMM at the previous level's
//   resolution) and this kernel evaluates
//       y[b,p,:] = act( c * ( up2(t)[b,p,:]  +  sum_k W_s[b,:,k] PE[p,k] ) + bias )
//   with up2(t) taken in the epilogue from the four low-resolution neighbours.  Against dgv2_modconv_pe_fwd at level 4
//   (32768 px, Ka = 64, Ks = 512, O = 32) the per-sample HBM stream drops from 6.3 MB (xa in, y out; + 5.2 MB for
//   writing and re-reading up2(h) in the producer) to 2.1 MB (y out) + 0.5 MB of t from L2, and the full-resolution
//   MFMA work loses its Ka columns.
// Structure (as modconv_pe.hip: a block owns 256 pixels and walks the samples; PE fragments stay in registers):
//   * v_mfma_f32_32x32x16_bf16, O = 32 = M: a wave owns 32 pixels (one N fragment), 32 K-steps per sample;
//   * per-sample weights W_s[b] (32 x Ks): two LDS buffers filled one sample ahead by LDS-DMA in the image
//     [K step][K half][o][16 B], which a wave reads as two contiguous 512-byte runs per MFMA (conflict-free);
//   * the eight 16-byte tap loads of a sample (4 neighbours x 2 channel runs) are issued before its MFMA loop;
//   * epilogue: v_permlane32_swap on the fp32 accumulators gives each lane two runs of 8 consecutive channels, then
//     up2 + c + bias + leaky ReLU + bf16 and two 16-byte stores per lane; optional sum-of-squares partials of y.
#include "gemm_core.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct MUGeom {
  int B, P, Wout, Hin, Win;   // output pixels P = Hout * Wout; t is [B, Hin, Win, 32]
  int Ks, I, koff;            // w[b][o][koff + k], row stride I; I == 0: w is the per-sample MFMA image
                              // [B][Ks/16][2][32 o][8 k] (koff ignored): every DMA piece is one contiguous 1 KB
  int samples_per_block;
  const int* idx_h;           // [Hout][2] low-res rows of the two taps, coef_h [Hout][2]
  const float* coef_h;
  const int* idx_w;           // [Wout][2]
  const float* coef_w;
  const float* bias;
  const float* row_scale;
  int act;
  float alpha, scale;
  float* sumsq;
};

template <int KS16>   // Ks / 16
__global__ __launch_bounds__(512, 2) void modconv_up_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ t,
                                                            const bf16_t* __restrict__ xs, const bf16_t* __restrict__ w,
                                                            MUGeom g) {
  constexpr int O = 32;
  constexpr int WBUF = KS16 * 64;                   // 16-byte slots of one sample's weights: [KS16][32 o][2]
  constexpr int NW = WBUF / 512;                    // DMA pieces per thread (KS16 % 8 == 0)
  extern __shared__ __attribute__((aligned(16))) uint4 lds_w[];   // 2 x WBUF, then bias[32], cs[32]
  float* s_bias = reinterpret_cast<float*>(lds_w + 2 * WBUF);
  float* s_cs = s_bias + O;
  const unsigned lds_off = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_w;   // LDS byte address

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = lane & 31, kg = lane >> 5;
  const int p0 = blockIdx.x * 256 + wave * 32;
  const int b0 = blockIdx.y * g.samples_per_block;
  const int b1 = min(b0 + g.samples_per_block, g.B);
  const int px = min(p0 + n, g.P - 1);              // clamped: pixels past the end are computed, never stored
  if (tid < O) {
    s_bias[tid] = g.bias ? g.bias[tid] : 0.f;
    s_cs[tid] = g.row_scale ? g.row_scale[tid] : 1.f;
  }

  // ---- PE fragments (B operand: column n = pixel, k = 8 * kg + i): registers for the whole walk ----
  uint4 pe[KS16];
#pragma unroll
  for (int kc = 0; kc < KS16; ++kc)
    pe[kc] = *reinterpret_cast<const uint4*>(xs + (int64_t)px * g.Ks + kc * 16 + kg * 8);

  // ---- up2 taps of this lane's pixel: rows (wave-uniform when Wout % 32 == 0; kept per lane for generality) ----
  const int Y = px / g.Wout, X = px - Y * g.Wout;
  const int iy0 = g.idx_h[2 * Y], iy1 = g.idx_h[2 * Y + 1], ix0 = g.idx_w[2 * X], ix1 = g.idx_w[2 * X + 1];
  const float wy0 = g.coef_h[2 * Y], wy1 = g.coef_h[2 * Y + 1], wx0 = g.coef_w[2 * X], wx1 = g.coef_w[2 * X + 1];
  const float tw[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
  // after the permlane32 exchange this lane owns channels [8 kg, 8 kg + 8) and [16 + 8 kg, 16 + 8 kg + 8)
  const int toff[4] = {(iy0 * g.Win + ix0) * O + 8 * kg, (iy0 * g.Win + ix1) * O + 8 * kg,
                       (iy1 * g.Win + ix0) * O + 8 * kg, (iy1 * g.Win + ix1) * O + 8 * kg};

  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef __attribute__((address_space(1))) const void gbl_void_t;
  // LDS slot L = kc * 64 + half * 32 + o  <-  w[b][o][koff + kc*16 + half*8 .. +8]: a wave's A-fragment read is two
  // contiguous 512-byte runs (lanes 0-31 / 32-63), conflict-free for ds_read_b128's lane groups (a 32-byte row stride
  // is 2-way conflicted: MI355X_MICROARCH.md, LDS table)
  auto dma_w = [&](int b, int buf) {
    const bf16_t* wb = w + (int64_t)b * O * g.I + g.koff;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int L = tid + j * 512;
      const int o = L & 31, half = (L >> 5) & 1, kc = L >> 6;
      const bf16_t* src = g.I ? wb + o * g.I + kc * 16 + half * 8 : w + ((int64_t)b * WBUF + L) * 8;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(lds_w + buf * WBUF + j * 512 + wave * 64), 16, 0, 0);
    }
  };

  float ss = 0.f;
  auto step = [&](int buf, int b) {
    // this sample's weights were issued one sample ago, BEFORE that sample's 8 tap loads (consumed since) and its 2
    // stores: vector-memory operations retire in issue order, so all but the 2 youngest done means the DMA has landed
    // (first sample of the walk / a wave without live pixels issues no stores: full drain)
    if (b == b0 || p0 >= g.P) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's pieces landed; every wave is done with the other buffer
    asm volatile("" ::: "memory");
    // taps of THIS sample, in flight behind the MFMA loop.  Issued as asm BEFORE the DMA and awaited with a counted
    // vmcnt below: with a glds in flight hipcc drains everything (vmcnt(0)) at the first use of an ordinary load,
    // i.e. the epilogue would wait for the NEXT sample's weight transfer.
    const bf16_t* tb = t + (int64_t)b * g.Hin * g.Win * O;
    u32x4 tap[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16_t* tp = tb + toff[q];
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(tap[q][0]) : "v"(tp) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(tap[q][1]) : "v"(tp) : "memory");
    }
    dma_w(min(b + 1, b1 - 1), buf ^ 1);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // A fragments: row m = lane % 32 = output channel, k = 8 * (lane / 32) + i  <-  slot kc * 64 + kg * 32 + n.
    // The reads are issued as asm (ring of RD registers, PF reads in flight, explicit lgkmcnt): left to the compiler,
    // every ds_read that follows the LDS-DMA above in program order is preceded by `s_waitcnt vmcnt(0)` (it cannot
    // prove the read does not alias the DMA's destination), which parks the whole MFMA loop behind the NEXT sample's
    // weight transfer -- the stall both sample-walk kernels spent half their cycles in.
    {
      constexpr int PF = 4, RD = 6;
      const unsigned abase = lds_off + (unsigned)(buf * WBUF + kg * 32 + n) * 16u;
      u32x4 a[RD];
#define DGV2_DS_READ(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(abase), "n"(off))
#define DGV2_LGKM_WAIT(dst, cnt) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "n"(cnt))
#pragma unroll
      for (int kc = 0; kc < PF; ++kc) DGV2_DS_READ(a[kc % RD], kc * 1024);
#pragma unroll
      for (int kc = 0; kc < KS16; ++kc) {
        // reads return in order: all but the (issued - kc - 1) youngest are back
        constexpr int dummy = 0;
        (void)dummy;
        const int inflight = (kc + PF <= KS16 ? PF : KS16 - kc) - 1;
        switch (inflight) {
          case 3: DGV2_LGKM_WAIT(a[kc % RD], 3); break;
          case 2: DGV2_LGKM_WAIT(a[kc % RD], 2); break;
          case 1: DGV2_LGKM_WAIT(a[kc % RD], 1); break;
          default: DGV2_LGKM_WAIT(a[kc % RD], 0); break;
        }
        union { u32x4 u; bf16x8 v; } ua;
        union { uint4 u; bf16x8 v; } ub;
        ua.u = a[kc % RD];
        ub.u = pe[kc];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
        if (kc + PF < KS16) DGV2_DS_READ(a[(kc + PF) % RD], (kc + PF) * 1024);   // slot last read two MFMAs ago
      }
#undef DGV2_DS_READ
#undef DGV2_LGKM_WAIT
    }

    // lane (n, kg) holds channels 8j + 4kg .. +3 (j = 0..3); exchange so that it holds two runs of 8 channels
    float v[2][8];
#pragma unroll
    for (int j = 0; j < 4; j += 2)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[4 * j + r]), __float_as_uint(acc[4 * (j + 1) + r]),
                                                  false, false);
        v[j >> 1][r] = __uint_as_float(s[0]);       // kg = 0: own group j      | kg = 1: partner's group j + 1
        v[j >> 1][4 + r] = __uint_as_float(s[1]);   // kg = 0: partner's group j | kg = 1: own group j + 1
      }
    // the taps are older than the NW DMA pieces just issued: all but the NW youngest operations done = taps landed
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
    // tied one by one BEHIND the wait: tied to the wait itself, operand set-up copies could read the registers before
    // their data has arrived (seen in conv_strip.hip)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      asm volatile("" : "+v"(tap[q][0]));
      asm volatile("" : "+v"(tap[q][1]));
    }
    const bool live = p0 + n < g.P;
    bf16_t* row = y + ((int64_t)b * g.P + px) * O;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float up[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) up[e] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        vec16<bf16_t> tv;
        tv.raw = make_uint4(tap[q][h][0], tap[q][h][1], tap[q][h][2], tap[q][h][3]);
#pragma unroll
        for (int e = 0; e < 8; ++e) up[e] = fmaf(tw[q], tv.get(e), up[e]);
      }
      const int c0 = 16 * h + 8 * kg;
      vec16<bf16_t> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = fmaf(v[h][e] + up[e], s_cs[c0 + e], s_bias[c0 + e]);
        if (g.act == 3) f = fmaxf(f, f * g.alpha) * g.scale;   // leaky ReLU, 0 <= alpha <= 1
        o.set(e, f);
      }
      if (live) {
        *reinterpret_cast<uint4*>(row + c0) = o.raw;
        if (g.sumsq) ss += sumsq_bf16x8(o.raw);
      }
    }
  };

  if (b0 < b1) dma_w(b0, 0);
  for (int b = b0; b < b1; ++b) step((b - b0) & 1, b);
  if (g.sumsq) {
    __shared__ float red[16];
    const float s = block_sum(ss, red);
    if (tid == 0) g.sumsq[blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

}  // namespace

// y[b,p,:32] = act( row_scale * ( up2(t)[b,p,:] + sum_{k<Ks} xs[p,k] w[b,:,koff+k] ) + bias )   (bf16 in / out)
//   t [B,Hin,Win,32] = W_a . h at the previous level's resolution; up2 by the two-tap tables idx/coef [Hout][2], [Wout][2]
//   (native.ResampleSpec.tables of the block's up-2 Resample, zero-padded to two taps); xs [Hout*Wout, Ks] batch-shared
//   PE; w [B,32,I] prepared per-sample weights, PE columns start at koff.  Ks in {512}.  sumsq: one partial per block.
extern "C" int dgv2_modconv_up_fwd(void* y, const void* t, const void* xs, const void* w, int B, int Hout, int Wout,
                                   int Hin, int Win, int Ks, int O, int I, int koff, const int* idx_h,
                                   const float* coef_h, const int* idx_w, const float* coef_w, const float* row_scale,
                                   const float* bias, int act, float alpha, float scale, int dtype, float* sumsq,
                                   int sumsq_cap, int* sumsq_used, void* stream) {
  if (sumsq_used) *sumsq_used = 0;
  if (!y || !t || !xs || !w || !idx_h || !coef_h || !idx_w || !coef_w || B <= 0 || Hout <= 0 || Wout <= 0) return DGV2_EINVAL;
  if (dtype != DGV2_BF16 || (act != 0 && act != 3) || O != 32 || Ks != 512 || (koff & 7) || (I & 7) || (I && koff + Ks > I))
    return DGV2_ENOTSUP;
  if (!aligned16(y) || !aligned16(t) || !aligned16(xs) || !aligned16(w)) return DGV2_EINVAL;
  const int P = Hout * Wout;
  MUGeom g{B, P, Wout, Hin, Win, Ks, I, koff, 1, idx_h, coef_h, idx_w, coef_w, bias, row_scale, act, alpha, scale, sumsq};
  const int tiles = (P + 255) / 256;
  int nsplit = (256 + tiles - 1) / tiles;          // one resident block per CU (the PE fragments fill the registers)
  nsplit = nsplit < 1 ? 1 : (nsplit > B ? B : nsplit);
  g.samples_per_block = (B + nsplit - 1) / nsplit;
  nsplit = (B + g.samples_per_block - 1) / g.samples_per_block;
  if (g.sumsq && sumsq_used && tiles * nsplit <= sumsq_cap) *sumsq_used = tiles * nsplit;
  else g.sumsq = nullptr;
  constexpr int KS16 = 32;
  const size_t lds = sizeof(uint4) * 2 * KS16 * 64 + sizeof(float) * 64;
  auto kern = modconv_up_kernel<KS16>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid(tiles, nsplit);
  kern<<<grid, 512, lds, (hipStream_t)stream>>>((bf16_t*)y, (const bf16_t*)t, (const bf16_t*)xs, (const bf16_t*)w, g);
  DGV2_RETURN_LAST();
}
