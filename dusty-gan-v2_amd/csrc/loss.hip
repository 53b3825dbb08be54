// Non-saturating GAN objective and the discriminator statistics the trainer logs, as ONE launch forward and one
// multiply backward (reference: GANLoss "nsgan", gans/models/loss.py:37-41,66-69; Trainer.step scalars and
// AdaptiveAugment.cumulate, gans/trainer.py:293,400-406, gans/augment/adaptive_augment.py:368-370 -- a dozen
// elementwise / reduction launches on [B,1] logits per step otherwise).
//   loss = mean_r softplus(-y_r) + mean_f softplus(y_f)          (either part may be empty)
//   gy_r = -s sigmoid(-y_r) / n_r,   gy_f = s sigmoid(y_f) / n_f     (s * d loss / d y: the cotangent of y, s = the objective's weight)
//   stats = [loss, mean(y_r), mean(y_f), sum sign(y_r)]
#include "common.h"

namespace {

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch's threshold

__global__ __launch_bounds__(256) void nsgan_kernel(float* __restrict__ stats, float* __restrict__ gy,
                                                    const float* __restrict__ y, int n_real, int n_fake, float gs,
                                                    float* __restrict__ sign_cum, float* __restrict__ n_cum) {
  __shared__ float red[16];
  float l = 0.f, mr = 0.f, mf = 0.f, sg = 0.f;
  const float ir = n_real > 0 ? 1.f / n_real : 0.f, jf = n_fake > 0 ? 1.f / n_fake : 0.f;
  for (int i = threadIdx.x; i < n_real + n_fake; i += 256) {
    const float v = y[i];
    if (i < n_real) {
      l += softplus_f(-v) * ir;
      mr += v * ir;
      sg += (v > 0.f) - (v < 0.f);
      gy[i] = -gs * ir / (1.f + expf(v));     // -sigmoid(-v) / n_r
    } else {
      l += softplus_f(v) * jf;
      mf += v * jf;
      gy[i] = gs * jf / (1.f + expf(-v));     //  sigmoid(v) / n_f
    }
  }
  float out[4] = {l, mr, mf, sg};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float s = block_sum(out[k], red);
    if (threadIdx.x == 0) {
      stats[k] = s;
      if (k == 3 && sign_cum) {   // AdaptiveAugment.cumulate (adaptive_augment.py:368-370) in the same launch
        *sign_cum += s;
        *n_cum += (float)n_real;
      }
    }
    __syncthreads();
  }
}

}  // namespace

// y fp32 [n_real + n_fake] (reals first), stats fp32 [4], gy fp32 [n_real + n_fake] = gy_scale * d loss / d y.
// sign_cum / n_cum (both or neither; fp32 [1] each): ADA's running statistic, += sum sign(y_real) and += n_real.
extern "C" int dgv2_nsgan_loss(float* stats, float* gy, const float* y, int n_real, int n_fake, float gy_scale, float* sign_cum,
                               float* n_cum, void* stream) {
  if (!stats || !gy || !y || n_real < 0 || n_fake < 0 || n_real + n_fake <= 0 || (!sign_cum) != (!n_cum)) return DGV2_EINVAL;
  nsgan_kernel<<<1, 256, 0, (hipStream_t)stream>>>(stats, gy, y, n_real, n_fake, gy_scale, sign_cum, n_cum);
  DGV2_RETURN_LAST();
}
