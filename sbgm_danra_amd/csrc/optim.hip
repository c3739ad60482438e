// Optimizer step of the training loop (reference training.py:407 `optimizer.step()` on torch.optim.Adam / AdamW built by
// training_utils.get_optimizer, :50-59): one launch over every parameter tensor of the model.
//
// K40 adam_batched_kernel — descriptor table (param, grad, exp_avg, exp_avg_sq, numel) on the device, one workgroup per
//     1024 elements, the tensor found by binary search over the block prefix (as the batched weight pack does).  The update
//     is torch's (torch/optim/adam.py, single-tensor form; the same expressions as its fused CUDA kernel):
//         g'      = s * g + wd * p                  (Adam: L2 term)        |  p *= 1 - lr * wd   (AdamW: decoupled)
//                   s = grad_scale: 1, or 1 / world when g is the data-parallel SUM of the replicas' gradients (the division of
//                   reference-style averaging rides on this launch instead of a 76 MB read + write pass of its own)
//         m       = m + (1 - b1) * (g' - m)
//         v       = b2 * v + (1 - b2) * g' * g'
//         p      -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//     t is read from a device scalar, so a captured step replays correctly.
// HBM-bound: 28 B per parameter (4 reads + 3 writes); 19 M parameters = 532 MB per step.
#include <cmath>

#include "../../include/sbgm_hip.h"
#include "common.h"
#include "kernels.h"

namespace {

constexpr int ADAM_BLOCK_ELEMS = 1024;

__global__ __launch_bounds__(256) void adam_batched_kernel(const sbgm_adam_desc* __restrict__ desc, int n, const float* __restrict__ step,
                                                           float lr, float beta1, float beta2, float eps, float wd, int decoupled, float gscale) {
    int lo = 0, hi = n - 1;                                  // last descriptor whose first block is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const sbgm_adam_desc d = desc[lo];
    const float t = step[0];
    const float bc1 = 1.f - powf(beta1, t), bc2_sqrt = sqrtf(1.f - powf(beta2, t));
    const float step_size = lr / bc1;
    const int64_t i0 = (int64_t)(blockIdx.x - d.block_begin) * ADAM_BLOCK_ELEMS + threadIdx.x * 4;
    if (i0 >= d.numel) return;
    auto update = [&](float& p, float g, float& m, float& v) {
        g *= gscale;
        if (wd != 0.f) {
            if (decoupled) p *= 1.f - lr * wd; else g += wd * p;
        }
        m += (1.f - beta1) * (g - m);
        v = beta2 * v + (1.f - beta2) * g * g;
        p -= step_size * m / (sqrtf(v) / bc2_sqrt + eps);
    };
    const bool vec = i0 + 3 < d.numel && (((uintptr_t)d.p | (uintptr_t)d.g | (uintptr_t)d.m | (uintptr_t)d.v) & 15) == 0;
    if (vec) {
        f32x4 p = *reinterpret_cast<const f32x4*>(d.p + i0), m = *reinterpret_cast<const f32x4*>(d.m + i0);
        f32x4 v = *reinterpret_cast<const f32x4*>(d.v + i0);
        const f32x4 g = *reinterpret_cast<const f32x4*>(d.g + i0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = p[e], me = m[e], ve = v[e];
            update(pe, g[e], me, ve);
            p[e] = pe; m[e] = me; v[e] = ve;
        }
        *reinterpret_cast<f32x4*>(d.p + i0) = p;
        *reinterpret_cast<f32x4*>(d.m + i0) = m;
        *reinterpret_cast<f32x4*>(d.v + i0) = v;
    } else {
        for (int64_t i = i0; i < d.numel && i < i0 + 4; ++i) update(d.p[i], d.g[i], d.m[i], d.v[i]);
    }
}

}  // namespace

int sbgm_adam_blocks(int64_t numel) { return (int)((numel + ADAM_BLOCK_ELEMS - 1) / ADAM_BLOCK_ELEMS); }

int sbgm_launch_adam_batched(const sbgm_adam_desc* desc_dev, int n, int total_blocks, const float* step_dev, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int decoupled, float grad_scale, hipStream_t st) {
    SBGM_CHECK(desc_dev && step_dev && n >= 1 && total_blocks >= 1, "adam_step_batched: bad arguments");
    SBGM_CHECK(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "adam_step_batched: betas (%g, %g) / eps %g", beta1,
               beta2, eps);
    hipLaunchKernelGGL(adam_batched_kernel, dim3(total_blocks), dim3(256), 0, st, desc_dev, n, step_dev, lr, beta1, beta2, eps, weight_decay,
                       decoupled, grad_scale);
    SBGM_LAUNCH_CHECK();
    return 0;
}
