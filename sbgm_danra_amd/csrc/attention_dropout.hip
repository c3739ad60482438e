// Train-mode attention dropout: nn.MultiheadAttention(dropout = p) drops softmax probabilities (reference sbgm/score_unet.py:127, the
// `dropout` argument of ImageSelfAttention; nothing in the reference sets it, so this path is about API completeness, not speed).
//   A = softmax(q k^T / sqrt(d));  D_ij = keep_ij / (1 - p);  O = (A o D) V
//   dV = (A o D)^T dO;  dA = D o (dO V^T);  dS = A o (dA - rowsum(dA o A));  dQ = dS K / sqrt(d);  dK = dS^T Q / sqrt(d)
// keep_ij comes from the Philox stream (philox.h) keyed by (seed, offset, element index ((b heads + h) S + i) S + j): forward, backward and
// sbgm_mha_dropout_mask (the tests' window on the draw) see the same mask whatever their launch geometry.  One workgroup per
// (sample, head, block of 16 queries), 16 lanes per query, score rows in LDS — the layout of the scalar backward kernel in backward.hip.
#include "common.h"
#include "kernels.h"
#include "philox.h"

namespace {

struct DropArgs {
    unsigned long long seed, offset;
    float p, inv_keep;
};

__device__ __forceinline__ float drop_factor(const DropArgs& dr, int bh, int i, int j, int S) {
    const unsigned long long e = ((unsigned long long)bh * S + i) * S + j;
    const f32x4 u = philox_uniform4(dr.seed, dr.offset, e >> 2);
    return u[(int)(e & 3)] >= dr.p ? dr.inv_keep : 0.f;
}

// softmax rows of this workgroup's 16 queries into Pm [16][S]; returns nothing, every lane of a query's 16 ends with the row finished
__device__ __forceinline__ void softmax_rows(float* Pm, const float* base, const float* qrow, size_t rs, int C, int S, int d, float scale,
                                             int qi_l, int sub) {
    float mx = -INFINITY;
    for (int j = sub; j < S; j += 16) {
        const float* kr = base + (size_t)j * rs + C;
        float s = 0.f;
        for (int e = 0; e < d; ++e) s = fmaf(qrow[e], kr[e], s);
        s *= scale;
        Pm[qi_l * S + j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = sub; j < S; j += 16) { const float p = expf(Pm[qi_l * S + j] - mx); Pm[qi_l * S + j] = p; sum += p; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.f / sum;
    for (int j = sub; j < S; j += 16) Pm[qi_l * S + j] *= inv;
}

__global__ __launch_bounds__(256) void mha_core_dropout_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B, int S, int C,
                                                               int heads, float scale, DropArgs dr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Pm = reinterpret_cast<float*>(smem_raw);      // [16][S]
    const int d = C / heads, qblocks = (S + 15) / 16;
    int w = blockIdx.x;
    const int qb = w % qblocks; w /= qblocks;
    const int h = w % heads, b = w / heads;
    const int qi_l = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int qi = qb * 16 + qi_l;
    const bool q_ok = qi < S;
    const size_t rs = 3 * (size_t)C;
    const float* base = qkv + (size_t)b * S * rs + (size_t)h * d;
    const float* qrow = base + (size_t)(q_ok ? qi : 0) * rs;
    softmax_rows(Pm, base, qrow, rs, C, S, d, scale, qi_l, sub);
    for (int j = sub; j < S; j += 16) Pm[qi_l * S + j] *= drop_factor(dr, b * heads + h, q_ok ? qi : 0, j, S);
    __syncthreads();                                     // the product below reads the whole row, written by 16 different lanes
    if (!q_ok) return;
    for (int e = sub; e < d; e += 16) {
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc = fmaf(Pm[qi_l * S + j], base[(size_t)j * rs + 2 * C + e], acc);
        out[((size_t)b * S + qi) * C + (size_t)h * d + e] = acc;
    }
}

__global__ __launch_bounds__(256) void mha_core_dropout_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                   float* __restrict__ dqkv, int B, int S, int C, int heads, float scale,
                                                                   DropArgs dr) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Pm = reinterpret_cast<float*>(smem_raw);      // [16][S]: A, then A o D
    float* dSm = Pm + 16 * S;                            // [16][S]: dA, then dS
    const int d = C / heads, qblocks = (S + 15) / 16;
    int w = blockIdx.x;
    const int qb = w % qblocks; w /= qblocks;
    const int h = w % heads, b = w / heads;
    const int qi_l = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int qi = qb * 16 + qi_l;
    const bool q_ok = qi < S;
    const size_t rs = 3 * (size_t)C;
    const float* base = qkv + (size_t)b * S * rs + (size_t)h * d;
    float* dbase = dqkv + (size_t)b * S * rs + (size_t)h * d;
    const float* qrow = base + (size_t)(q_ok ? qi : 0) * rs;
    const float* dorow = dout + ((size_t)b * S + (q_ok ? qi : 0)) * C + (size_t)h * d;
    softmax_rows(Pm, base, qrow, rs, C, S, d, scale, qi_l, sub);
    float delta = 0.f;
    for (int j = sub; j < S; j += 16) {
        const float* vr = base + (size_t)j * rs + 2 * C;
        float dp = 0.f;
        for (int e = 0; e < d; ++e) dp = fmaf(dorow[e], vr[e], dp);
        const float da = dp * drop_factor(dr, b * heads + h, q_ok ? qi : 0, j, S);
        dSm[qi_l * S + j] = da;
        delta += Pm[qi_l * S + j] * da;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) delta += __shfl_xor(delta, o, 64);
    for (int j = sub; j < S; j += 16) {
        const float a = Pm[qi_l * S + j];
        dSm[qi_l * S + j] = q_ok ? a * (dSm[qi_l * S + j] - delta) * scale : 0.f;
        Pm[qi_l * S + j] = q_ok ? a * drop_factor(dr, b * heads + h, qi, j, S) : 0.f;
    }
    __syncthreads();
    if (q_ok) {
        for (int e = sub; e < d; e += 16) {
            float acc = 0.f;
            for (int j = 0; j < S; ++j) acc = fmaf(dSm[qi_l * S + j], base[(size_t)j * rs + C + e], acc);
            dbase[(size_t)qi * rs + e] = acc;
        }
    }
    const int nq = min(16, S - qb * 16);
    for (int idx = threadIdx.x; idx < S * d; idx += blockDim.x) {
        const int j = idx / d, e = idx - j * d;
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < nq; ++i) {
            const int qq = qb * 16 + i;
            ak = fmaf(dSm[i * S + j], base[(size_t)qq * rs + e], ak);
            av = fmaf(Pm[i * S + j], dout[((size_t)b * S + qq) * C + (size_t)h * d + e], av);
        }
        atomicAdd(dbase + (size_t)j * rs + C + e, ak);
        atomicAdd(dbase + (size_t)j * rs + 2 * C + e, av);
    }
}

__global__ __launch_bounds__(256) void mha_dropout_mask_kernel(float* __restrict__ mask, int BH, int S, DropArgs dr) {
    const size_t total = (size_t)BH * S * S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % S);
        const int q = (int)((i / S) % S);
        mask[i] = drop_factor(dr, (int)(i / ((size_t)S * S)), q, j, S);
    }
}

inline int drop_args(float p, unsigned long long seed, unsigned long long offset, DropArgs* dr) {
    SBGM_CHECK(p >= 0.f && p < 1.f, "mha dropout: p=%g must lie in [0, 1)", (double)p);
    *dr = DropArgs{seed, offset, p, 1.f / (1.f - p)};
    return 0;
}

}  // namespace

int sbgm_launch_mha_core_dropout(const float* qkv, const float* dout, float* out_or_dqkv, int B, int S, int C, int heads, float p,
                                 unsigned long long seed, unsigned long long offset, int backward, hipStream_t st) {
    SBGM_CHECK(heads > 0 && C % heads == 0, "mha dropout: C=%d heads=%d", C, heads);
    const size_t lds = (size_t)(backward ? 2 : 1) * 16 * S * 4;
    SBGM_CHECK(lds <= 150 * 1024, "mha dropout: S=%d too long for the LDS-resident score rows", S);
    DropArgs dr;
    if (drop_args(p, seed, offset, &dr)) return 1;
    const float scale = 1.f / sqrtf((float)(C / heads));
    const int blocks = B * heads * ((S + 15) / 16);
    if (backward) {
        if (!sbgm_scratch_prezeroed) { if (sbgm_zero_async(out_or_dqkv, (size_t)B * S * 3 * C * 4, st)) return 1; }
        static bool attr = false;
        if (!attr) { SBGM_HIP(hipFuncSetAttribute((const void*)mha_core_dropout_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
        hipLaunchKernelGGL(mha_core_dropout_bwd_kernel, dim3(blocks), dim3(256), lds, st, qkv, dout, out_or_dqkv, B, S, C, heads, scale, dr);
    } else {
        static bool attr = false;
        if (!attr) { SBGM_HIP(hipFuncSetAttribute((const void*)mha_core_dropout_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
        hipLaunchKernelGGL(mha_core_dropout_kernel, dim3(blocks), dim3(256), lds, st, qkv, out_or_dqkv, B, S, C, heads, scale, dr);
    }
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_mha_dropout_mask(float* mask, int B, int S, int heads, float p, unsigned long long seed, unsigned long long offset,
                                 hipStream_t st) {
    DropArgs dr;
    if (drop_args(p, seed, offset, &dr)) return 1;
    const size_t total = (size_t)B * heads * S * S;
    hipLaunchKernelGGL(mha_dropout_mask_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, st, mask, B * heads, S, dr);
    SBGM_LAUNCH_CHECK();
    return 0;
}
