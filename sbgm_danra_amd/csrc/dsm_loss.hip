// K23 / K24: the denoising-score-matching loss around the network (reference sbgm/score_unet.py:936-985).
//   perturb : t_b = U(0,1)*(1-t_eps)+t_eps ; z ~ N(0,1) ; std_b = marginal_prob_std(t_b) ; x~ = x + std_b * z      (:957-963)
//   loss    : mean_b sum_{c,h,w} w * (score * std_b + z)^2 ,  w = sigmoid(sdf)*0.5 + 0.5  or 1                      (:974-984)
//   backward: dscore = dL * (2/B) * w * (score * std_b + z) * std_b
// All three are HBM-bound elementwise / reduction passes over [B][1][H][W] (12-16 B per element).  The draws are either the
// caller's (parity tests inject the reference's (t, z)) or in-kernel Philox keyed by a seed passed by value (eager calls) or by a
// DEVICE-resident (seed, offset) pair, so that a captured training step replays with fresh noise: the loss kernel advances the
// offset once per call.
#include "common.h"
#include "kernels.h"
#include "philox.h"

namespace {

__device__ __forceinline__ float ve_std(float t, float sigma) {            // score_unet.py:881-897, fp32 like the reference
    const float ls = logf(sigma);
    return fmaxf(sqrtf((expf((2.f * t) * ls) - 1.f) / (2.f * ls)), 1e-5f);
}

// grid (blocks over the quads of one sample, B)
__global__ __launch_bounds__(256) void dsm_perturb_kernel(const float* __restrict__ x, const float* __restrict__ z_in,
                                                          const float* __restrict__ t_in,
                                                          const unsigned long long* __restrict__ rng,
                                                          unsigned long long seed_val, float t_eps, float sigma,
                                                          float* __restrict__ xp, float* __restrict__ z_out,
                                                          float* __restrict__ t_out, float* __restrict__ std_out, size_t per4) {
    const int b = blockIdx.y;
    const unsigned long long seed = rng ? rng[0] : seed_val, off = rng ? rng[1] : 0ull;
    float t;
    if (t_in) t = t_in[b];
    else t = philox_uniform4(seed, 2ull * off, (unsigned long long)b)[0] * (1.f - t_eps) + t_eps;          // :957
    // sigma > 0: the VE schedule; sigma <= 0: std_out[b] holds the caller's marginal_prob_std(t_b) on entry (any schedule)
    const float sd = sigma > 0.f ? ve_std(t, sigma) : std_out[b];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t_out[b] = t;
        if (sigma > 0.f) std_out[b] = sd;
    }
    const size_t base = (size_t)b * per4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[base + i];
        const f32x4 zv = z_in ? reinterpret_cast<const f32x4*>(z_in)[base + i] : philox_normal4(seed, 2ull * off + 1ull, base + i);
        if (!z_in) reinterpret_cast<f32x4*>(z_out)[base + i] = zv;
        reinterpret_cast<f32x4*>(xp)[base + i] = xv + sd * zv;                                               // :963
    }
}

// grid (nblk, B): one fp64 partial per block, wavefront shuffles then the 4 waves through LDS in fixed order
__global__ __launch_bounds__(256) void dsm_loss_partial_kernel(const float* __restrict__ score, const float* __restrict__ z,
                                                               const float* __restrict__ std, const float* __restrict__ sdf,
                                                               double* __restrict__ partial, size_t per4) {
    __shared__ double wsum[4];
    const int b = blockIdx.y;
    const float sd = std[b];
    const size_t base = (size_t)b * per4;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 s = reinterpret_cast<const f32x4*>(score)[base + i];
        const f32x4 zv = reinterpret_cast<const f32x4*>(z)[base + i];
        f32x4 w = {1.f, 1.f, 1.f, 1.f};
        if (sdf) {
            const f32x4 d = reinterpret_cast<const f32x4*>(sdf)[base + i];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (1.f / (1.f + expf(-d[e]))) * 0.5f + 0.5f;                    // :977
        }
        const f32x4 r = s * sd + zv;
        const f32x4 q = w * (r * r);                                                                          // :984
        acc += (double)((q[0] + q[1]) + (q[2] + q[3]));
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// one block: per-sample sums in block order, mean over the batch, advance the RNG offset
__global__ __launch_bounds__(64) void dsm_loss_finish_kernel(const double* __restrict__ partial, int B, int nblk,
                                                             float* __restrict__ loss, unsigned long long* rng) {
    double tot = 0.0;
    for (int b = threadIdx.x; b < B; b += 64) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += partial[(size_t)b * nblk + k];
        tot += s;
    }
    tot = wave_sum_d(tot);
    if (threadIdx.x == 0) {
        loss[0] = (float)(tot / (double)B);
        if (rng) rng[1] += 1ull;
    }
}

__global__ __launch_bounds__(256) void dsm_loss_bwd_kernel(const float* __restrict__ score, const float* __restrict__ z,
                                                           const float* __restrict__ std, const float* __restrict__ sdf,
                                                           const float* __restrict__ dloss, float* __restrict__ dscore,
                                                           float inv_b2, size_t per4) {
    const int b = blockIdx.y;
    const float sd = std[b];
    const float k = dloss[0] * inv_b2 * sd;
    const size_t base = (size_t)b * per4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 s = reinterpret_cast<const f32x4*>(score)[base + i];
        const f32x4 zv = reinterpret_cast<const f32x4*>(z)[base + i];
        f32x4 w = {1.f, 1.f, 1.f, 1.f};
        if (sdf) {
            const f32x4 d = reinterpret_cast<const f32x4*>(sdf)[base + i];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (1.f / (1.f + expf(-d[e]))) * 0.5f + 0.5f;
        }
        reinterpret_cast<f32x4*>(dscore)[base + i] = k * (w * (s * sd + zv));
    }
}

inline int dsm_blocks(size_t per4) { return (int)std::min<size_t>(64, (per4 + 1023) / 1024); }   // >= 4 quads per thread

}  // namespace

int sbgm_dsm_nblk(int64_t per_sample) { return dsm_blocks((size_t)per_sample / 4); }

int sbgm_launch_dsm_perturb(const float* x, const float* z_in, const float* t_in, const unsigned long long* rng,
                            unsigned long long seed, float t_eps, float sigma, float* xp, float* z_out, float* t_out, float* std_out, int B, size_t per, hipStream_t st) {
    SBGM_CHECK(x && xp && t_out && std_out, "dsm_perturb: x, xp, t_out and std_out are required");
    SBGM_CHECK(B >= 1 && per % 4 == 0, "dsm_perturb: B=%d, per-sample size %zu must be a multiple of 4", B, per);
    SBGM_CHECK(per == 0 || z_in || z_out, "dsm_perturb: z_out is required when the noise is drawn in the kernel");
    hipLaunchKernelGGL(dsm_perturb_kernel, dim3(per ? dsm_blocks(per / 4) : 1, B), dim3(256), 0, st, x, z_in, t_in, rng, seed, t_eps, sigma, xp, z_out,
                       t_out, std_out, per / 4);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_dsm_loss_fwd(const float* score, const float* z, const float* std, const float* sdf, double* partial_ws, float* loss,
                             unsigned long long* rng_advance, int B, size_t per, hipStream_t st) {
    SBGM_CHECK(score && z && std && partial_ws && loss, "dsm_loss_fwd: null argument");
    SBGM_CHECK(B >= 1 && per >= 4 && per % 4 == 0, "dsm_loss_fwd: B=%d, per-sample size %zu must be a positive multiple of 4", B, per);
    const int nblk = dsm_blocks(per / 4);
    hipLaunchKernelGGL(dsm_loss_partial_kernel, dim3(nblk, B), dim3(256), 0, st, score, z, std, sdf, partial_ws, per / 4);
    SBGM_LAUNCH_CHECK();
    hipLaunchKernelGGL(dsm_loss_finish_kernel, dim3(1), dim3(64), 0, st, partial_ws, B, nblk, loss, rng_advance);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_dsm_loss_bwd(const float* score, const float* z, const float* std, const float* sdf, const float* dloss, float* dscore,
                             int B, size_t per, hipStream_t st) {
    SBGM_CHECK(score && z && std && dloss && dscore, "dsm_loss_bwd: null argument");
    SBGM_CHECK(B >= 1 && per >= 4 && per % 4 == 0, "dsm_loss_bwd: B=%d, per-sample size %zu must be a positive multiple of 4", B, per);
    hipLaunchKernelGGL(dsm_loss_bwd_kernel, dim3(dsm_blocks(per / 4), B), dim3(256), 0, st, score, z, std, sdf, dloss, dscore,
                       2.0f / (float)B, per / 4);
    SBGM_LAUNCH_CHECK();
    return 0;
}
