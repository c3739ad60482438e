// Post-sampling row of SURVEY.md §8f: back-transform of generated fields to physical units, the extreme-value sentinel and
// the optional clamp, on the device (the reference bounces the sample through host memory for these:
// generation.py:71-107, training.py:697-748).
//
// K31 pointwise_chain_kernel   — a short program of scalar ops applied per element.  The reference's transform classes
//     (special_transforms.py:62-140 Scale/ScaleBackTransform, :143-233 ZScore*, :239-462 PrcpLog*) are chains of separately
//     rounded fp32 tensor-scalar ops; a program keeps their op order and rounding (no re-association, no FMA contraction),
//     so the affine transforms are bit-identical to the reference and only exp/log differ by the libm implementation.
// K32 sample_extremes_kernel   — per-sample max and linear-interpolated quantile (torch.quantile semantics,
//     utils.py:1647-1649) by MSB-first radix select on order-preserving keys: one workgroup per sample, 5 passes over a
//     sample that sits in L2 (64-256 KB), 2 floats per sample leave the device instead of the whole field.
// HBM-bound streaming kernels: 8 B/element (chain), 4 B/element read once + L2 re-reads (extremes).
#include <algorithm>
#include <cmath>

#include "../../include/sbgm_hip.h"
#include "common.h"
#include "kernels.h"

namespace {

struct ChainProgram {
    int n_ops;
    int op[SBGM_CHAIN_MAX_OPS];
    float c[SBGM_CHAIN_MAX_OPS];
};

__device__ __forceinline__ float chain_apply(float v, const ChainProgram& p) {
#pragma unroll
    for (int i = 0; i < SBGM_CHAIN_MAX_OPS; ++i) {
        if (i >= p.n_ops) break;
        const float c = p.c[i];
        switch (p.op[i]) {
            case SBGM_OP_ADD: v = __fadd_rn(v, c); break;
            case SBGM_OP_MUL: v = __fmul_rn(v, c); break;
            case SBGM_OP_DIV: v = __fdiv_rn(v, c); break;
            case SBGM_OP_CLAMP_MIN: v = v < c ? c : v; break;     // comparisons keep NaN, like torch.clamp
            case SBGM_OP_CLAMP_MAX: v = v > c ? c : v; break;
            case SBGM_OP_EXP: v = expf(v); break;
            case SBGM_OP_LOG: v = logf(v); break;
            default: break;
        }
    }
    return v;
}

__global__ __launch_bounds__(256) void pointwise_chain_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                              ChainProgram p) {
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = chain_apply(v[e], p);
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
    // ragged tail (n % 4 elements)
    const size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = chain_apply(x[t], p);
}

// ascending order-preserving key of an fp32 value (NaN sorts above +inf, as torch.sort places it last)
__device__ __forceinline__ unsigned int order_key(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// One workgroup per sample.  Selects the k0-th smallest key (0-based) with 4 passes of an 8-bit MSB-first radix histogram,
// then one pass finds the next order statistic (k0+1) = either the same value (duplicates) or the smallest larger key.
__global__ __launch_bounds__(1024) void sample_extremes_kernel(const float* __restrict__ x, size_t per, unsigned int k0,
                                                               int need_next, float weight, float* __restrict__ out_max,
                                                               float* __restrict__ out_q) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sh_prefix, sh_rank, sh_cnt_le, sh_min_gt, sh_max, sh_nan;
    const float* xs = x + (size_t)blockIdx.x * per;
    const int tid = threadIdx.x;
    if (tid == 0) { sh_prefix = 0; sh_rank = k0; sh_cnt_le = 0; sh_min_gt = 0xFFFFFFFFu; sh_max = 0; sh_nan = 0; }
    unsigned int mask = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned int prefix = sh_prefix;
        unsigned int local_max = 0, local_nan = 0;
        for (size_t i = tid; i < per; i += 1024) {
            const float f = xs[i];
            const unsigned int k = order_key(f);
            if (pass == 0) { local_max = k > local_max ? k : local_max; local_nan |= (f != f); }
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 0xFF], 1u);
        }
        if (pass == 0) {
            atomicMax(&sh_max, local_max);
            if (local_nan) sh_nan = 1;
        }
        __syncthreads();
        if (tid == 0) {                               // 256-bin scan by one lane: negligible next to the pass itself
            unsigned int r = sh_rank, b = 0;
            while (b < 255 && r >= hist[b]) { r -= hist[b]; ++b; }
            sh_rank = r;
            sh_prefix = prefix | (b << shift);
        }
        mask |= 0xFFu << shift;
        __syncthreads();
    }
    const unsigned int key0 = sh_prefix;
    if (need_next) {
        unsigned int cnt = 0, mn = 0xFFFFFFFFu;
        for (size_t i = tid; i < per; i += 1024) {
            const unsigned int k = order_key(xs[i]);
            cnt += (k <= key0);
            if (k > key0 && k < mn) mn = k;
        }
        atomicAdd(&sh_cnt_le, cnt);
        atomicMin(&sh_min_gt, mn);
        __syncthreads();
    }
    if (tid == 0) {
        const float v0 = key_value(key0);
        float q = v0;
        if (need_next) {
            const float v1 = (sh_cnt_le >= k0 + 2) ? v0 : key_value(sh_min_gt);
            const float diff = v1 - v0;                 // ATen lerp (native/Lerp.h): two-sided form
            q = weight < 0.5f ? v0 + weight * diff : v1 - diff * (1.f - weight);
        }
        const float nanv = __uint_as_float(0x7FC00000u);
        out_q[blockIdx.x] = sh_nan ? nanv : q;          // torch.quantile / torch.max propagate NaN
        out_max[blockIdx.x] = sh_nan ? nanv : key_value(sh_max);
    }
}

}  // namespace

int sbgm_launch_pointwise_chain(const float* x, float* y, size_t n, int n_ops, const int* ops, const float* consts, hipStream_t st) {
    SBGM_CHECK(n_ops >= 0 && n_ops <= SBGM_CHAIN_MAX_OPS, "pointwise_chain: %d ops (max %d)", n_ops, SBGM_CHAIN_MAX_OPS);
    SBGM_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "pointwise_chain: pointers must be 16-byte aligned");
    ChainProgram p{};
    p.n_ops = n_ops;
    for (int i = 0; i < n_ops; ++i) {
        SBGM_CHECK(ops[i] >= SBGM_OP_ADD && ops[i] <= SBGM_OP_LOG, "pointwise_chain: unknown op code %d", ops[i]);
        p.op[i] = ops[i];
        p.c[i] = consts[i];
    }
    if (n == 0) return 0;
    const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, 4096);
    hipLaunchKernelGGL(pointwise_chain_kernel, dim3(blocks), dim3(256), 0, st, x, y, n, p);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_sample_extremes(const float* x, int B, size_t per, float q, float* out_max, float* out_q, hipStream_t st) {
    SBGM_CHECK(B >= 1 && per >= 1 && per < (1ull << 31), "sample_extremes: B=%d per_sample=%zu", B, per);
    SBGM_CHECK(q >= 0.f && q <= 1.f, "sample_extremes: q=%g outside [0, 1]", (double)q);
    // torch.quantile: rank = q * (n - 1) evaluated in the tensor's dtype (fp32), then floor / ceil / lerp
    const float rank = q * (float)(per - 1);
    const float below = floorf(rank);
    const unsigned int k0 = (unsigned int)below;
    const float weight = rank - below;
    const int need_next = ceilf(rank) != below;
    hipLaunchKernelGGL(sample_extremes_kernel, dim3(B), dim3(1024), 0, st, x, per, k0, need_next, weight, out_max, out_q);
    SBGM_LAUNCH_CHECK();
    return 0;
}
