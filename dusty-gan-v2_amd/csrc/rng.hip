// Every random number one step body consumes, from ONE launch (reference: the torch.randn / torch.rand calls scattered
// over a training iteration -- z in Trainer.sample_z, gans/trainer.py:206-208; the azimuth shift of
// SynthesisNetwork.forward, gans/models/dusty_v2.py:267-274; the uniforms of GumbelSigmoid / RelaxedBernoulli.rsample,
// gans/models/ops/gumbel.py:23-29; the ~20 draws of AdaptiveAugment.sample_affine / sample_color,
// gans/augment/adaptive_augment.py:386-470; the warm-up keep mask, gans/trainer.py:241-245 -- about fifteen generator
// launches plus their scale / clamp companions per iteration, and two philox-state fills per hipGraph replay).
//
// Philox4x32-10 (Salmon et al., SC'11; the generator torch.cuda uses), counter = stream offset + 4-value group index,
// key = seed.  The stream state lives in DEVICE memory the caller owns (state[0] = seed, state[1] = offset, state[2] =
// arrival ticket): every block reads the offset, the block that arrives last advances it -- so a launch captured into a
// hipGraph draws fresh numbers on every replay with no host involvement and no second launch.
#include "common.h"

namespace {

constexpr int RNG_MAX_SEG = 16;

struct RngSegs {
  float* out[RNG_MAX_SEG];
  long long begin[RNG_MAX_SEG + 1];   // in 4-value groups: segment s owns groups [begin[s], begin[s + 1])
  long long count[RNG_MAX_SEG];       // values
  int kind[RNG_MAX_SEG];              // 0: a + (b - a) u, u in [0, 1);  1: a + b n, n ~ N(0, 1);  2: clamp(u, a, b);  3: u < a ? 1 : 0
  float a[RNG_MAX_SEG], b[RNG_MAX_SEG];
  int nseg;
};

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint64_t key) {
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// The argument block is read where it lies (the kernarg segment): a by-value struct indexed with a per-thread segment number
// is otherwise copied to scratch memory first.
__global__ __launch_bounds__(256) void rng_fill_kernel(RngSegs s_by_value, unsigned long long* __restrict__ state) {
  const RngSegs& s = *(const RngSegs*)__builtin_amdgcn_kernarg_segment_ptr();
  const unsigned long long seed = state[0], offset = state[1];
  const int nseg = s.nseg;
  const long long total = s.begin[nseg];
  for (long long g = blockIdx.x * 256ll + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < RNG_MAX_SEG; ++j) k += (j < nseg && g >= s.begin[j]) ? 1 : 0;   // segments are ordered: count the starts passed
    const int kind = s.kind[k];
    const float a = s.a[k], b = s.b[k];
    const unsigned long long ctr = offset + (unsigned long long)g;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, seed);
    float v[4];
    if (kind == 1) {
      // Box-Muller on (0, 1] x [0, 1): two pairs per group
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float u1 = ((c[2 * h] >> 8) + 1u) * 5.9604645e-8f, u2 = (c[2 * h + 1] >> 8) * 5.9604645e-8f;
        const float r = sqrtf(-2.f * __logf(u1));
        float sn, cs;
        __sincosf(6.2831853071795865f * u2, &sn, &cs);
        v[2 * h] = a + b * r * cs;
        v[2 * h + 1] = a + b * r * sn;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = (c[j] >> 8) * 5.9604645e-8f;   // 24 bits, [0, 1) like torch.rand
        v[j] = kind == 0 ? fmaf(b - a, u, a) : (kind == 2 ? fminf(fmaxf(u, a), b) : (u < a ? 1.f : 0.f));
      }
    }
    const long long e = (g - s.begin[k]) * 4, cnt = s.count[k];
    float* o = s.out[k] + e;
    if (e + 4 <= cnt && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (e + j < cnt) o[j] = v[j];
    }
  }
  // The last block to arrive advances the stream and resets the ticket.  No fence is needed: a block takes its ticket
  // after all its threads have READ the offset (the barrier), so the block that draws the last ticket knows every block
  // has; the new offset only has to be visible to the NEXT launch on this state (a kernel boundary).  (__threadfence()
  // here cost 50 us per launch: on this part a device-scope release is an L2 write-back + invalidate, once per block.)
  __shared__ bool last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = __hip_atomic_fetch_add(&state[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = t == (unsigned long long)gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    state[1] = offset + (unsigned long long)total;
    state[2] = 0ull;
  }
}

}  // namespace

// nseg <= 16 output segments out[s] (fp32 device, count[s] values) filled from the caller's Philox stream `state`
// (device uint64[4]: seed, offset, ticket = 0, unused) and the stream advanced by sum ceil(count[s] / 4) -- ONE launch.
//   kind 0: uniform in [a, b) (a + (b - a) u);  kind 1: normal with mean a, standard deviation b;
//   kind 2: u in [0, 1) clamped to [a, b] (torch.distributions' clamp_probs form of the Gumbel / logistic uniforms);
//   kind 3: Bernoulli(a) as 0.0 / 1.0 (u < a: the valid-return mask of the synthetic scans).
// Launches on `state` must be stream-ordered (they are: one stream, or one graph replay at a time).
extern "C" int dgv2_rng_fill(float* const* out, const int64_t* count, const int* kind, const float* a, const float* b, int nseg,
                             uint64_t* state, void* stream) {
  if (!out || !count || !kind || !a || !b || !state || nseg < 1 || nseg > RNG_MAX_SEG) return DGV2_EINVAL;
  RngSegs s;
  s.nseg = nseg;
  long long g = 0;
  for (int k = 0; k < nseg; ++k) {
    if (!out[k] || count[k] < 1 || kind[k] < 0 || kind[k] > 3) return DGV2_EINVAL;
    s.out[k] = out[k];
    s.count[k] = count[k];
    s.kind[k] = kind[k];
    s.a[k] = a[k];
    s.b[k] = b[k];
    s.begin[k] = g;
    g += (count[k] + 3) / 4;
  }
  s.begin[nseg] = g;
  // (one block per CU at most: the arrival tickets are same-address atomics, ~15 ns each at the L2 -- 2048 blocks spent
  // 30 us queueing for theirs, 256 spend 4)
  rng_fill_kernel<<<grid_for(g, 256, 256), 256, 0, (hipStream_t)stream>>>(s, (unsigned long long*)state);
  DGV2_RETURN_LAST();
}
