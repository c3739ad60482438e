// Device helpers shared by the convolution kernels (conv_igemm.hip, conv_wino.hip).
#pragma once
#include "common.h"
#include "kernels.h"

namespace {

__device__ __noinline__ f32x4 gelu4(f32x4 v) {      // rare (attention FF only): keep the erf expansion out of line
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = sbgm_act(v[e], SBGM_ACT_GELU);
    return v;
}

// scale/bias (folded BN or conv bias) -> early time bias -> residual -> activation -> late time bias
__device__ __forceinline__ f32x4 conv_epilogue(f32x4 v, const ConvParams& p, int co, size_t m, int b) {
    if (p.scale) v *= *reinterpret_cast<const f32x4*>(p.scale + co);
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + co);
    if (p.tbias && !p.tbias_after_act) v += *reinterpret_cast<const f32x4*>(p.tbias + (size_t)b * p.Cout + co);
    if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + m * p.Cout + co);
    if (p.act == SBGM_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == SBGM_ACT_GELU) {
        v = gelu4(v);
    }
    if (p.tbias && p.tbias_after_act) v += *reinterpret_cast<const f32x4*>(p.tbias + (size_t)b * p.Cout + co);
    return v;
}


// workgroup id -> logical id such that every XCD (blocks b, b+8, ... share one) owns a contiguous range (bijective)
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace
