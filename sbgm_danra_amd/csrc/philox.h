// Philox4x32-10 counter RNG + Box-Muller, shared by the sampler updates (sampler.hip) and the loss perturbation (dsm_loss.hip).
// Keyed by (seed, stream offset, element index): the same triple always yields the same draw, whatever the launch geometry.
#pragma once
#include "common.h"

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

// 4 standard normals for (seed, stream offset, index)
__device__ __forceinline__ f32x4 philox_normal4(unsigned long long seed, unsigned long long offset, unsigned long long idx) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
    const float u1 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(c[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0);
    sincosf(6.283185307179586f * u3, &s1, &c1);
    return f32x4{r0 * c0, r0 * s0, r1 * c1, r1 * s1};
}

// 4 uniforms in (0,1) for (seed, stream offset, index) — the same counter block as philox_normal4, before Box-Muller
__device__ __forceinline__ f32x4 philox_uniform4(unsigned long long seed, unsigned long long offset, unsigned long long idx) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return f32x4{((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f),
                 ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), ((float)(c[3] >> 8) + 0.5f) * (1.0f / 16777216.0f)};
}
