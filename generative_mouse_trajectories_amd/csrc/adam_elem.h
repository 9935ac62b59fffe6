// One element of torch.optim.Adam (RecurrentGAN.py:864-873: lr 1e-4, betas (0.5, 0.9),
// eps 1e-8, no weight decay / amsgrad), in the operation order of torch's
// _single_tensor_adam so that fp32 results agree to the last bits.
#pragma once
#include <math.h>
#ifndef GMT_HD
#  ifdef __HIPCC__
#    define GMT_HD __host__ __device__ __forceinline__
#  else
#    define GMT_HD static inline
#  endif
#endif

namespace gmt_adam {
GMT_HD void adam_elem(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                      float step_size, float bc2_sqrt)
{
    const float w = 1.0f - b1;                     // exp_avg.lerp_(grad, 1 - beta1), ATen's two-sided form
    m = (w < 0.5f) ? m + w * (g - m) : g - (g - m) * (1.0f - w);
    v = v * b2 + (1.0f - b2) * g * g;              // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps; // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - step_size * (m / denom);               // param.addcdiv_(exp_avg, denom, value=-step_size)
}
}  // namespace gmt_adam
