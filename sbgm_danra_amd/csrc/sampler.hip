// K25 / K26 / K27: per-step state updates of the reverse-SDE samplers, fused with the Gaussian draw.
// reference sbgm/score_sampling.py:124-125 (Euler-Maruyama), :200-204 (Langevin corrector), :224-227 (predictor),
// :55 (classifier-free-guidance combine).
//
// Noise: when `z` is null the kernels draw N(0,1) themselves (Philox4x32-10 counter RNG + Box-Muller, keyed by
// (seed, running offset, element index)), so the sampler loop never round-trips noise through HBM; when `z`
// is given (parity mode) the host-generated draw is used verbatim.
// All per-step scalars come from a device-resident table indexed by a device-side step counter, so one captured
// hipGraph replays for every step.
#include "common.h"
#include "kernels.h"
#include "philox.h"

namespace {

// Philox counter of quad i of a [B][H][W] batch.  Default: the quad index itself.  With a tile map (full-domain tiling,
// SURVEY.md 8f rank 3) the counter is the quad's position in the DOMAIN, so pixels that several overlapping tiles share
// receive the same draw in every tile and the tiles stay consistent where they are blended.
__device__ __forceinline__ unsigned long long noise_index(const NoiseMap& m, size_t i) {
    if (!m.origins) return i;
    const size_t per4 = (size_t)m.tile_h * m.tile_w4;
    const size_t b = i / per4, rem = i - b * per4;
    const size_t y = rem / m.tile_w4, x4 = rem - y * m.tile_w4;
    return ((unsigned long long)(m.origins[2 * b] + y)) * m.dom_w4 + (unsigned long long)(m.origins[2 * b + 1] >> 2) + x4;
}

__global__ void fill_kernel(float* t, float v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = v;
}

// x = scale * N(0,1)   (sampler start: randn * marginal_prob_std(1), score_sampling.py:94-95, :168)
__global__ __launch_bounds__(256) void init_noise_kernel(float* __restrict__ x, float scale, const float* __restrict__ z,
                                                         unsigned long long seed, const SamplerState* __restrict__ state,
                                                         unsigned long long off_val, size_t n4, NoiseMap nm) {
    const unsigned long long off = state ? state->rng_offset : off_val;
    if (state) seed = state->seed;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 n = z ? reinterpret_cast<const f32x4*>(z)[i] : philox_normal4(seed, off, noise_index(nm, i));
        reinterpret_cast<f32x4*>(x)[i] = n * scale;
    }
}

// x_mean = x + g^2 dt * score ;  x = x_mean + noise_coef * N(0,1)
__global__ __launch_bounds__(256) void em_update_kernel(float* __restrict__ x, float* __restrict__ x_mean,
                                                        const float* __restrict__ score, const float* __restrict__ z,
                                                        const StepScalars* __restrict__ table,
                                                        const SamplerState* __restrict__ state, StepScalars sc_val,
                                                        unsigned long long off_val, unsigned long long seed, size_t n4,
                                                        NoiseMap nm) {
    const StepScalars sc = state ? table[state->step] : sc_val;
    const unsigned long long off = state ? state->rng_offset : off_val;
    if (state) seed = state->seed;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 sv = reinterpret_cast<const f32x4*>(score)[i];
        const f32x4 n = z ? reinterpret_cast<const f32x4*>(z)[i] : philox_normal4(seed, off, noise_index(nm, i));
        const f32x4 mean = xv + (sc.g2 * sv) * sc.dt;   // association of score_sampling.py:124/:224
        reinterpret_cast<f32x4*>(x_mean)[i] = mean;
        reinterpret_cast<f32x4*>(x)[i] = mean + sc.noise * n;
    }
}

// runs after the update kernel of a step: advance the step counter / RNG offset, publish the next time
__global__ void advance_kernel(SamplerState* state, const StepScalars* table, float* t_dev, int B, int advance_step,
                               int n_steps) {
    const unsigned long long s = state->step;
    const unsigned long long ns = state->n_steps ? state->n_steps : (unsigned long long)n_steps;
    const int i = threadIdx.x;
    if (advance_step && t_dev && i < B) t_dev[i] = table[s].t_next;
    __syncthreads();
    if (i == 0) {
        state->rng_offset += 1;
        if (advance_step) state->step = (s + 1 < ns) ? s + 1 : s;
    }
}

// per-sample sum of squares of the score, fp64 atomics into sumsq[B] (zeroed by the launcher): ONE atomic per workgroup (the four
// wave sums meet in LDS first) and at most 16 workgroups per sample — same-address fp64 atomics retire at ~10 ns each, and one per wave
// of a 64 x B grid made this 4 MB reduction a 41 us kernel at B = 16, 256 x 256
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ score, double* __restrict__ sumsq,
                                                    size_t per_sample4) {
    __shared__ double part[4];
    const int b = blockIdx.y;
    const f32x4* s = reinterpret_cast<const f32x4*>(score) + (size_t)b * per_sample4;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = s[i];
        acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const double w = wave_sum_d((double)acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sumsq[b], (part[0] + part[1]) + (part[2] + part[3]));
}

// Langevin corrector: eps = 2 (snr*sqrt(CHW) / mean_b ||score_b||)^2 ;  x += eps*score + sqrt(2 eps) N(0,1)
// Tile mode (nm.origins set: the samples are tiles of one domain): the step size of a tile uses that tile's OWN score norm
// instead of the batch mean, so a tile's trajectory does not depend on which other tiles share its batch or its GPU
// (the reference has no tiler, its batch-mean rule :201 applies to batches of independent samples).
__global__ __launch_bounds__(256) void langevin_kernel(float* __restrict__ x, const float* __restrict__ score,
                                                       const float* __restrict__ z, float snr_noise_norm,
                                                       const double* __restrict__ sumsq,
                                                       const SamplerState* __restrict__ state,
                                                       unsigned long long off_val, unsigned long long seed, int B,
                                                       size_t n4, NoiseMap nm) {
    float gn = 0.f;
    for (int b = 0; b < B; ++b) gn += (float)sqrt(sumsq[b]);
    gn /= (float)B;
    const float r = snr_noise_norm / gn;
    float eps = 2.f * (r * r);
    float nz = sqrtf(2.f * eps);
    const bool per_tile = nm.origins != nullptr;
    const size_t per4 = n4 / (size_t)B;
    const unsigned long long off = state ? state->rng_offset : off_val;
    if (state) seed = state->seed;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (per_tile) {
            // a tile whose score is exactly zero (masked / constant tile) would get eps = inf and poison the stitched domain
            const float rt = snr_noise_norm / fmaxf((float)sqrt(sumsq[i / per4]), 1e-12f);
            eps = 2.f * (rt * rt);
            nz = sqrtf(2.f * eps);
        }
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 sv = reinterpret_cast<const f32x4*>(score)[i];
        const f32x4 n = z ? reinterpret_cast<const f32x4*>(z)[i] : philox_normal4(seed, off, noise_index(nm, i));
        reinterpret_cast<f32x4*>(x)[i] = xv + eps * sv + nz * n;
    }
}

__global__ __launch_bounds__(256) void cfg_combine_kernel(float* __restrict__ out, const float* __restrict__ sc,
                                                          const float* __restrict__ su, float w, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(sc)[i], b = reinterpret_cast<const f32x4*>(su)[i];
        reinterpret_cast<f32x4*>(out)[i] = (1.0f + w) * a - w * b;
    }
}

inline int stream_blocks(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 2048); }

}  // namespace

int sbgm_launch_fill_t(float* t, float value, int B, hipStream_t st) {
    hipLaunchKernelGGL(fill_kernel, dim3((B + 255) / 256), dim3(256), 0, st, t, value, B);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_init_noise(float* x, float scale, const float* z, unsigned long long seed, SamplerState* state,
                           unsigned long long draw_index, size_t n, hipStream_t st, NoiseMap nm) {
    SBGM_CHECK(n % 4 == 0, "init_noise: element count must be a multiple of 4");
    hipLaunchKernelGGL(init_noise_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, st, x, scale, z, seed, state, draw_index,
                       n / 4, nm);
    SBGM_LAUNCH_CHECK();
    if (state) {
        hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, st, state, (const StepScalars*)nullptr, (float*)nullptr, 0, 0, 0);
        SBGM_LAUNCH_CHECK();
    }
    return 0;
}

int sbgm_launch_em_update(float* x, float* x_mean, const float* score, const float* z, const StepScalars* table,
                          SamplerState* state, const StepScalars* sc_val, unsigned long long draw_index, float* t_dev,
                          unsigned long long seed, int B, size_t per_sample, int n_steps, hipStream_t st, int t_entries,
                          NoiseMap nm) {
    const size_t n = (size_t)B * per_sample;
    if (t_entries <= 0) t_entries = B;
    SBGM_CHECK(n % 4 == 0, "em_update: element count must be a multiple of 4");
    SBGM_CHECK(t_entries <= 1024, "em_update: %d time entries > 1024", t_entries);
    SBGM_CHECK(state != nullptr || sc_val != nullptr, "em_update: need a device table or explicit scalars");
    const StepScalars v = sc_val ? *sc_val : StepScalars{};
    hipLaunchKernelGGL(em_update_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, st, x, x_mean, score, z, table, state, v,
                       draw_index, seed, n / 4, nm);
    SBGM_LAUNCH_CHECK();
    if (state) {
        hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1024), 0, st, state, table, t_dev, t_entries, 1, n_steps);
        SBGM_LAUNCH_CHECK();
    }
    return 0;
}

int sbgm_launch_langevin(float* x, const float* score, const float* z, float snr_noise_norm, double* sumsq_ws,
                         SamplerState* state, unsigned long long draw_index, unsigned long long seed, int B,
                         size_t per_sample, hipStream_t st, NoiseMap nm) {
    SBGM_CHECK(per_sample % 4 == 0, "langevin: per-sample element count must be a multiple of 4");
    { if (sbgm_zero_async(sumsq_ws, sizeof(double) * B, st)) return 1; }
    const int bx = (int)std::min<size_t>((per_sample / 4 + 255) / 256, B >= 64 ? 4 : 16);
    hipLaunchKernelGGL(sumsq_kernel, dim3(bx, B), dim3(256), 0, st, score, sumsq_ws, per_sample / 4);
    SBGM_LAUNCH_CHECK();
    const size_t n4 = (size_t)B * per_sample / 4;
    hipLaunchKernelGGL(langevin_kernel, dim3(stream_blocks(n4)), dim3(256), 0, st, x, score, z, snr_noise_norm, sumsq_ws,
                       state, draw_index, seed, B, n4, nm);
    SBGM_LAUNCH_CHECK();
    if (state) {
        hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, st, state, (const StepScalars*)nullptr, (float*)nullptr, 0, 0, 0);
        SBGM_LAUNCH_CHECK();
    }
    return 0;
}

int sbgm_launch_cfg_combine(float* out, const float* s_cond, const float* s_uncond, float scale, size_t n, hipStream_t st) {
    SBGM_CHECK(n % 4 == 0, "cfg_combine: element count must be a multiple of 4");
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, st, out, s_cond, s_uncond, scale, n / 4);
    SBGM_LAUNCH_CHECK();
    return 0;
}
