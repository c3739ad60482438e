// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of the SBGM score-UNet hot path.
// Wavefront = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define SBGM_WAVE 64

// ---- error plumbing (thread-local message, C-ABI returns int status) ---------------------------
void sbgm_set_error(const char* fmt, ...);
#define SBGM_CHECK(cond, ...)                       \
    do {                                            \
        if (!(cond)) {                              \
            sbgm_set_error(__VA_ARGS__);            \
            return 1;                               \
        }                                           \
    } while (0)
#define SBGM_HIP(call)                                                                    \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            sbgm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)
#define SBGM_LAUNCH_CHECK() SBGM_HIP(hipGetLastError())

// Zero `bytes` (a multiple of 4) of device memory with a KERNEL on `st` (pointwise.hip).  Used instead of hipMemsetAsync in every
// launcher that can run under stream capture: a 196 KB memset node captured into a hipGraph (attention backward's dqkv) was
// observed to leave garbage from the second replay on (ROCm 7.2), while kernel nodes replay faithfully.
int sbgm_zero_async(void* p, size_t bytes, hipStream_t st);

// ---- activation codes (shared by epilogues) ----------------------------------------------------
enum { SBGM_ACT_NONE = 0, SBGM_ACT_RELU = 1, SBGM_ACT_SILU = 2, SBGM_ACT_GELU = 3 };

__device__ __forceinline__ float sbgm_act(float v, int act) {
    switch (act) {
        case SBGM_ACT_RELU: return v > 0.f ? v : 0.f;
        case SBGM_ACT_SILU: return v / (1.f + expf(-v));                    // x * sigmoid(x)
        case SBGM_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));  // exact erf GELU
        default: return v;
    }
}

// ---- wavefront reductions (64 lanes, butterfly over DPP/ds_swizzle via __shfl_xor) ---------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- buffer resource: 32-bit offsets + hardware bounds check (out-of-range lanes read 0) --------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), /*stride*/ 0, (int)bytes, 0x00020000);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x2, v);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
