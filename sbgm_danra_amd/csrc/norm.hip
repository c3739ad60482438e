// Normalisation kernels over NHWC fp32: GroupNorm/InstanceNorm (decoder), LayerNorm (attention tokens) and
// train-mode BatchNorm2d (encoder).  All reductions: per-thread fp32 partials over short pixel stripes,
// wavefront / LDS combination in fp64, one fp64 atomic per (workgroup, statistic).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NORM_THREADS = 256;

// ---- statistics: sum and sum of squares per (sample, channel-set) ------------------------------------------
// Grid (chunks, B).  Thread layout: channel quad fastest (coalesced 16 B loads along C), pixel stripes above.
// `per_channel` = 0: reduce to G groups of C/G channels (GroupNorm);  = 1: keep C channels (BatchNorm, where
// blockIdx.y slices pixels of the whole batch instead of one sample).
template <bool PER_CHANNEL>
__global__ __launch_bounds__(NORM_THREADS) void norm_stats_kernel(const float* __restrict__ x, double* __restrict__ stats,
                                                                  int HW, int C, int G, int px_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* ssum = reinterpret_cast<double*>(smem_raw);   // [C]
    double* ssq = ssum + C;                               // [C]
    const int cq = C >> 2;
    const int b = blockIdx.y;
    for (int c = threadIdx.x; c < 2 * C; c += NORM_THREADS) ssum[c] = 0.0;
    __syncthreads();

    const int q = threadIdx.x % cq;
    const int lanes_px = NORM_THREADS / cq;       // pixel stripes handled concurrently (cq <= 256)
    const int stripe = threadIdx.x / cq;
    const int p_begin = blockIdx.x * px_per_block;
    const int p_end = min(HW, p_begin + px_per_block);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (stripe < lanes_px) {
        const float* base = x + (size_t)b * HW * C + q * 4;
        for (int p = p_begin + stripe; p < p_end; p += lanes_px) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * C);
            s += v;
            s2 += v * v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&ssum[q * 4 + e], (double)s[e]);      // ds_add_f64
            atomicAdd(&ssq[q * 4 + e], (double)s2[e]);
        }
    }
    __syncthreads();
    if (PER_CHANNEL) {
        for (int c = threadIdx.x; c < C; c += NORM_THREADS) {
            atomicAdd(&stats[2 * c], ssum[c]);
            atomicAdd(&stats[2 * c + 1], ssq[c]);
        }
    } else {
        const int cpg = C / G;
        for (int g = threadIdx.x; g < G; g += NORM_THREADS) {
            double a = 0.0, a2 = 0.0;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += ssum[c]; a2 += ssq[c]; }
            atomicAdd(&stats[((size_t)b * G + g) * 2], a);
            atomicAdd(&stats[((size_t)b * G + g) * 2 + 1], a2);
        }
    }
}


// ---- GroupNorm pass 1: partial sums per (sample, chunk, group) -> stats[b][chunk][g][2] (doubles) -------------------
constexpr int GN_MAX_CHUNKS = 64;
constexpr int GN_APPLY_BLOCKS = 1024;   // blocks of the apply sweep over the whole batch; measured 128..16384: 1024 is best (each block re-reduces the chunk partials)
__global__ __launch_bounds__(NORM_THREADS) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ stats,
                                                                  int HW, int C, int G, int px_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* ssum = reinterpret_cast<double*>(smem_raw);   // [C]
    double* ssq = ssum + C;                               // [C]
    const int cq = C >> 2;
    const int b = blockIdx.y;
    for (int c = threadIdx.x; c < 2 * C; c += NORM_THREADS) ssum[c] = 0.0;
    __syncthreads();
    const int q = threadIdx.x % cq;
    const int lanes_px = NORM_THREADS / cq;
    const int stripe = threadIdx.x / cq;
    const int p_begin = blockIdx.x * px_per_block;
    const int p_end = min(HW, p_begin + px_per_block);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (stripe < lanes_px) {
        const float* base = x + (size_t)b * HW * C + q * 4;
        for (int p = p_begin + stripe; p < p_end; p += lanes_px) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * C);
            s += v;
            s2 += v * v;
        }
    }
    if ((cq & (cq - 1)) == 0 && cq <= NORM_THREADS) {
        // power-of-two channel count: combine the stripes that share a channel quad with wavefront shuffles, park one fp32
        // partial per (row, quad) in LDS with plain stores and finish in fp64 — no LDS atomics (they dominated this kernel)
        float* part = reinterpret_cast<float*>(ssq + C);                 // [rows][cq][8]
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int o = cq; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += __shfl_xor(s[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        }
        const int row = cq <= 64 ? wave : stripe;
        const int rows = cq <= 64 ? NORM_THREADS / 64 : NORM_THREADS / cq;
        if (cq > 64 || lane < cq) {
            float* o = part + ((size_t)row * cq + q) * 8;
            *reinterpret_cast<f32x4*>(o) = s;
            *reinterpret_cast<f32x4*>(o + 4) = s2;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += NORM_THREADS) {
            double a = 0.0, a2 = 0.0;
            for (int r = 0; r < rows; ++r) {
                const float* o = part + ((size_t)r * cq + (c >> 2)) * 8 + (c & 3);
                a += (double)o[0];
                a2 += (double)o[4];
            }
            ssum[c] = a;
            ssq[c] = a2;
        }
    } else if (stripe < lanes_px) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&ssum[q * 4 + e], (double)s[e]);      // ds_add_f64
            atomicAdd(&ssq[q * 4 + e], (double)s2[e]);
        }
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = threadIdx.x; g < G; g += NORM_THREADS) {
        double a = 0.0, a2 = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += ssum[c]; a2 += ssq[c]; }
        double* o = stats + (((size_t)b * gridDim.x + blockIdx.x) * G + g) * 2;
        o[0] = a;
        o[1] = a2;
    }
}

// ---- K18 + K21 + K12 + K20: y = act( GN(x) * gamma + beta  [+ skip] [+ tbias[b]] ) ---------------------------
// reference score_unet.py:585/592 (norms), :600 (skip add), :612 (time add), :615 (activation).  Grid (blocks, B).
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ skip,
                                                              const float* __restrict__ tbias, int act, int HW, int C,
                                                              int G, int chunks, float eps,
                                                              const double* __restrict__ stats, float* __restrict__ mr_out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* mr = reinterpret_cast<float*>(smem_raw);        // [G][2] mean, rstd
    const int b = blockIdx.y;
    const int cq = C >> 2, cpg = C / G;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        double a = 0.0, a2 = 0.0;
        const double* sp = stats + ((size_t)b * chunks * G + g) * 2;
        for (int c = 0; c < chunks; ++c) { a += sp[(size_t)c * G * 2]; a2 += sp[(size_t)c * G * 2 + 1]; }
        const double inv_n = 1.0 / ((double)HW * cpg);
        const double mean = a * inv_n;
        const double var = fmax(a2 * inv_n - mean * mean, 0.0);
        mr[2 * g] = (float)mean;
        mr[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
        if (mr_out != nullptr && blockIdx.x == 0) {          // saved for the backward pass
            mr_out[((size_t)b * G + g) * 2] = mr[2 * g];
            mr_out[((size_t)b * G + g) * 2 + 1] = mr[2 * g + 1];
        }
    }
    __syncthreads();
    const size_t per_sample = (size_t)HW * cq;
    const f32x4* xb = reinterpret_cast<const f32x4*>(x) + (size_t)b * per_sample;
    const f32x4* sb = skip ? reinterpret_cast<const f32x4*>(skip) + (size_t)b * per_sample : nullptr;
    f32x4* yb = reinterpret_cast<f32x4*>(y) + (size_t)b * per_sample;
    if (256 % cq == 0) {
        // fast path (C = 64 .. 1024 in powers of two): a thread keeps ONE channel quad for the whole sweep, so mean, rstd*gamma,
        // beta and the time bias live in registers and the loop body is load - sub - fma - (add) - act - store: no index
        // divisions, no LDS or parameter reads per element
        const int c = (threadIdx.x % cq) * 4;
        f32x4 mu, a, bt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (c + e) / cpg;
            mu[e] = mr[2 * g];
            a[e] = gamma ? mr[2 * g + 1] * gamma[c + e] : mr[2 * g + 1];
            bt[e] = gamma ? beta[c + e] : 0.f;
        }
        if (tbias) bt += *reinterpret_cast<const f32x4*>(tbias + (size_t)b * C + c);
        const size_t stride = (size_t)gridDim.x * blockDim.x;        // a multiple of cq, so the quad never changes
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + stride < per_sample; i += 2 * stride) {          // two independent quads in flight per thread
            f32x4 v0 = xb[i], v1 = xb[i + stride];
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            if (sb) { s0 = sb[i]; s1 = sb[i + stride]; }
            v0 = (v0 - mu) * a + s0 + bt;
            v1 = (v1 - mu) * a + s1 + bt;
            if (act != SBGM_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = sbgm_act(v0[e], act); v1[e] = sbgm_act(v1[e], act); }
            }
            yb[i] = v0;
            yb[i + stride] = v1;
        }
        if (i < per_sample) {
            f32x4 v = (xb[i] - mu) * a;
            if (sb) v += sb[i];
            v += bt;
            if (act != SBGM_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sbgm_act(v[e], act);
            }
            yb[i] = v;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        f32x4 v = xb[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (c + e) / cpg;
            float o = (v[e] - mr[2 * g]) * mr[2 * g + 1];
            if (gamma) o = o * gamma[c + e] + beta[c + e];
            v[e] = o;
        }
        if (sb) v += sb[i];
        if (tbias) v += *reinterpret_cast<const f32x4*>(tbias + (size_t)b * C + c);
        if (act != SBGM_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = sbgm_act(v[e], act);
        }
        yb[i] = v;
    }
}

// GroupNorm statistics -> per-(sample, channel) affine for the consumer convolution's load path (conv_lds.hip, input modes):
// out[b][c/4][0][c%4] = rstd*gamma, out[b][c/4][1][c%4] = beta - mean*rstd*gamma (+ tbias[b][c]).  Same statistics expressions as
// groupnorm_apply_kernel; one block per sample.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ stats, int chunks, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ tbias,
                                                          float* __restrict__ out, int HW, int C, int G, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* mr = reinterpret_cast<float*>(smem_raw);        // [G][2] mean, rstd
    const int b = blockIdx.x, cpg = C / G;
    // one wavefront-sized team per group sums the <= 64 chunk partials with shuffles (fixed order: deterministic); a serial loop
    // per group made this launch 7 us of pure latency
    const int lane = threadIdx.x & 63, team = threadIdx.x >> 6;
    for (int g = team; g < G; g += 4) {
        const double* sp = stats + ((size_t)b * chunks * G + g) * 2;
        double a = lane < chunks ? sp[(size_t)lane * G * 2] : 0.0, a2 = lane < chunks ? sp[(size_t)lane * G * 2 + 1] : 0.0;
        a = wave_sum_d(a);
        a2 = wave_sum_d(a2);
        if (lane == 0) {
            const double inv_n = 1.0 / ((double)HW * cpg);
            const double mean = a * inv_n;
            const double var = fmax(a2 * inv_n - mean * mean, 0.0);
            mr[2 * g] = (float)mean;
            mr[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float a = gamma ? mr[2 * g + 1] * gamma[c] : mr[2 * g + 1];
        float sh = (gamma ? beta[c] : 0.f) - mr[2 * g] * a;
        if (tbias) sh += tbias[(size_t)b * C + c];
        float* o = out + (((size_t)b * (C >> 2) + (c >> 2)) * 2) * 4 + (c & 3);
        o[0] = a;
        o[4] = sh;
    }
}

// ---- K13: LayerNorm over C, one wave per token, two-pass in registers (wavefront shuffle reductions) ----------
// reference score_unet.py:128-129, :141, :145 (eps 1e-5, affine)
template <int VPL>   // float4 vectors per lane: C = 256 * VPL  (VPL = 0 -> generic strided path)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= M) return;
    const float* xr = x + (size_t)row * C;
    float* yr = y + (size_t)row * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = wave_sum(s) / (float)C;
    float s2 = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c) - mean;
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float rstd = 1.f / sqrtf(wave_sum(s2) / (float)C + eps);
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = (*reinterpret_cast<const f32x4*>(xr + c) - mean) * rstd;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
        *reinterpret_cast<f32x4*>(yr + c) = v * g + be;
    }
}

// ---- K10 (train): BatchNorm2d with batch statistics ---------------------------------------------------------
// y = relu?( (x - mean_c) * rsqrt(var_c + eps) * gamma + beta [+ res] ) [+ tbias_after[b]];  running stats get
// momentum-weighted mean and UNBIASED variance, as torch.nn.BatchNorm2d does (used at score_unet.py:323 and in
// every BasicBlock).
// Training-mode apply pass with the statistics finalisation folded in: every block derives the per-channel (mean, rstd) pairs
// from the fp64 sums into LDS (same expressions as the former separate finalize launch, so forward values and running
// statistics are unchanged); block 0 also stores the pairs for the backward pass and updates the running statistics.
__global__ __launch_bounds__(256) void batchnorm_train_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ res, const float* __restrict__ tbias_after,
                                                                    int relu, int B, int HW, int C, const double* __restrict__ stats,
                                                                    float* __restrict__ mr, float* running_mean, float* running_var,
                                                                    double n, float eps, float momentum) {
    extern __shared__ float bn_ab[];                      // [C][2]: scale = rstd*gamma, shift = beta - mean*scale ... kept as (mean, rstd)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double inv_n = 1.0 / n;
        const double mean = stats[2 * c] * inv_n;
        const double var = fmax(stats[2 * c + 1] * inv_n - mean * mean, 0.0);
        const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
        bn_ab[2 * c] = mf;
        bn_ab[2 * c + 1] = rf;
        if (blockIdx.x == 0) {
            mr[2 * c] = mf;
            mr[2 * c + 1] = rf;
            if (running_mean != nullptr) {
                const double mean_r = stats[2 * c] / n;
                const double var_r = fmax(stats[2 * c + 1] / n - mean_r * mean_r, 0.0);
                const double unbiased = n > 1.0 ? var_r * n / (n - 1.0) : var_r;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean_r;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    __syncthreads();
    const int cq = C >> 2;
    const size_t total = (size_t)B * HW * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        const int b = (int)(i / ((size_t)cq * HW));
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (v[e] - bn_ab[2 * (c + e)]) * bn_ab[2 * (c + e) + 1] * gamma[c + e] + beta[c + e];
        }
        if (res) v += reinterpret_cast<const f32x4*>(res)[i];
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (tbias_after) v += *reinterpret_cast<const f32x4*>(tbias_after + (size_t)b * C + c);
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

inline int stream_blocks(size_t work_items) { return (int)std::min<size_t>((work_items + 255) / 256, 2048); }

}  // namespace

int sbgm_launch_groupnorm_apply(const float* x, float* y, const float* gamma, const float* beta, const float* skip, const float* tbias,
                                int act, int B, int HW, int C, int G, float eps, const double* stats, int chunks, hipStream_t st,
                                float* mr_out) {
    SBGM_CHECK(C % 4 == 0 && C <= 1024 && C % G == 0 && chunks >= 1 && chunks <= GN_MAX_CHUNKS, "groupnorm_apply: C=%d G=%d chunks=%d", C, G,
               chunks);
    const size_t per_sample = (size_t)HW * (C / 4);
    const int bx = (int)std::max<size_t>(1, std::min<size_t>((per_sample + 255) / 256, GN_APPLY_BLOCKS / std::max(1, B) + 1));
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(bx, B), dim3(256), 2 * G * sizeof(float), st, x, y, gamma, beta, skip, tbias,
                       act, HW, C, G, chunks, eps, stats, mr_out);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_gn_partial(const float* x, double* stats_ws, int B, int HW, int C, int G, int* chunks_out, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0 && C <= 1024 && C % G == 0, "groupnorm: C=%d G=%d unsupported", C, G);
    // per-(sample, pixel-chunk, group) partial sums, plain stores (no zeroing, no atomics, deterministic)
    const int lanes_px = std::max(1, NORM_THREADS / (C / 4));
    int chunks = std::max(1, std::min(GN_MAX_CHUNKS, HW / (lanes_px * 16)));
    const int ppb = (HW + chunks - 1) / chunks;
    chunks = (HW + ppb - 1) / ppb;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, B), dim3(NORM_THREADS), 2 * C * sizeof(double) + 8192, st, x, stats_ws,
                       HW, C, G, ppb);
    SBGM_LAUNCH_CHECK();
    *chunks_out = chunks;
    return 0;
}

int sbgm_launch_gn_finalize(const double* stats, int chunks, const float* gamma, const float* beta, const float* tbias, float* out,
                            int B, int HW, int C, int G, float eps, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0 && C <= 1024 && C % G == 0 && chunks >= 1 && chunks <= GN_MAX_CHUNKS, "gn_finalize: C=%d G=%d chunks=%d", C, G, chunks);
    SBGM_CHECK((gamma == nullptr) == (beta == nullptr), "gn_finalize: gamma and beta must both be set or both null");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 2 * G * sizeof(float), st, stats, chunks, gamma, beta, tbias, out, HW, C, G, eps);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_groupnorm(const float* x, float* y, const float* gamma, const float* beta, const float* skip,
                          const float* tbias, int act, int B, int HW, int C, int G, float eps, double* stats_ws,
                          hipStream_t st, float* mr_out) {
    SBGM_CHECK((gamma == nullptr) == (beta == nullptr), "groupnorm: gamma and beta must both be set or both null");
    // pass 1: chunk partials; pass 2: every apply block re-reduces its sample's <= GN_MAX_CHUNKS partials into LDS, then streams.
    int chunks = 0;
    if (sbgm_launch_gn_partial(x, stats_ws, B, HW, C, G, &chunks, st)) return 1;
    return sbgm_launch_groupnorm_apply(x, y, gamma, beta, skip, tbias, act, B, HW, C, G, eps, stats_ws, chunks, st, mr_out);
}

int sbgm_launch_layernorm(const float* x, float* y, const float* gamma, const float* beta, int M, int C, float eps,
                          hipStream_t st) {
    SBGM_CHECK(C % 4 == 0, "layernorm: C=%d must be a multiple of 4", C);
    hipLaunchKernelGGL(layernorm_kernel<0>, dim3((M + 3) / 4), dim3(256), 0, st, x, y, gamma, beta, M, C, eps);
    SBGM_LAUNCH_CHECK();
    return 0;
}

// first half of a train-mode BatchNorm: per-channel fp64 sums (x, x^2) of THIS process's batch into stats_ws[2C]
int sbgm_launch_batchnorm_stats(const float* x, int B, int HW, int C, double* stats_ws, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0 && C <= 1024, "batchnorm: C=%d unsupported", C);
    if (!sbgm_scratch_prezeroed) { if (sbgm_zero_async(stats_ws, sizeof(double) * 2 * (size_t)C, st)) return 1; }
    // treat the batch as one long pixel axis: [B*HW][C]
    const int n = B * HW;
    const int lanes_px = std::max(1, NORM_THREADS / (C / 4));
    int chunks = std::max(1, std::min(1024, n / (lanes_px * 8)));     // measured: 8 px per thread (32: 1.8 % slower per step)
    const int ppb = (n + chunks - 1) / chunks;
    chunks = (n + ppb - 1) / ppb;
    hipLaunchKernelGGL(norm_stats_kernel<true>, dim3(chunks, 1), dim3(NORM_THREADS), 2 * C * sizeof(double), st, x,
                       stats_ws, n, C, 1, ppb);
    SBGM_LAUNCH_CHECK();
    return 0;
}

// second half: finalise (mean, rstd) from the sums over n_total values per channel — the local B*HW, or the sum over all
// ranks after the caller all-reduced stats_ws (SyncBatchNorm) — update the running statistics and apply
int sbgm_launch_batchnorm_apply(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, const float* res, const float* tbias_after, int relu, int B, int HW, int C,
                                float eps, float momentum, double* stats_ws, double n_total, hipStream_t st, float* mr_out) {
    SBGM_CHECK(C % 4 == 0 && C <= 1024, "batchnorm: C=%d unsupported", C);
    SBGM_CHECK(n_total >= (double)B * HW, "batchnorm: n_total=%g is smaller than the local batch (%d x %d)", n_total, B, HW);
    const int n = B * HW;
    float* mr = mr_out ? mr_out : reinterpret_cast<float*>(stats_ws + 2 * (size_t)C);
    hipLaunchKernelGGL(batchnorm_train_apply_kernel, dim3(stream_blocks((size_t)n * (C / 4))), dim3(256), 2 * C * sizeof(float), st, x, y,
                       gamma, beta, res, tbias_after, relu, B, HW, C, stats_ws, mr, running_mean, running_var, n_total, eps, momentum);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_batchnorm_train(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, const float* res, const float* tbias_after, int relu, int B,
                                int HW, int C, float eps, float momentum, double* stats_ws, hipStream_t st, float* mr_out) {
    if (sbgm_launch_batchnorm_stats(x, B, HW, C, stats_ws, st)) return 1;
    return sbgm_launch_batchnorm_apply(x, y, gamma, beta, running_mean, running_var, res, tbias_after, relu, B, HW, C, eps, momentum,
                                       stats_ws, (double)B * HW, st, mr_out);
}
