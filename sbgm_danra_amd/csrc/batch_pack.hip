// Pre-network row of SURVEY.md §8f (rank 2): batch-level condition assembly on the device.
//
// K33 assemble_conditions_kernel — replaces, for a whole batch in one launch,
//   * the concatenation of the sorted `*_lr` fields along channels      (reference sbgm/utils.py:441-447),
//   * the classifier-free-guidance condition dropout of the dataset       (reference sbgm/data_modules.py:957-983):
//       dropped sample -> LR fields zeroed, class label -> NULL token 0,
//   * the value||mask assembly of the geo fields (lsm, topo)              (reference sbgm/data_modules.py:971-993):
//       [value, mask] with mask = 0 for a dropped sample and 1 otherwise; a field that already carries its mask
//       channel is copied through unchanged (the reference's `geo.shape[0] == 1` test).
// Pure data movement: HBM-bound, (read + write) 8 B per output element, 16-byte accesses along H*W.
#include "../../include/sbgm_hip.h"
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

struct AssembleSrc {
    const float* ptr[SBGM_ASSEMBLE_MAX_LR];
    int channels[SBGM_ASSEMBLE_MAX_LR];
    int n;
};

// grid.y enumerates output planes (b, c); a plane is HW contiguous floats
__global__ __launch_bounds__(256) void assemble_lr_kernel(AssembleSrc src, const unsigned char* __restrict__ dropped,
                                                          float* __restrict__ out, int c_total, size_t hw) {
    const int plane = blockIdx.y;
    const int b = plane / c_total;
    int c = plane - b * c_total, k = 0;
    while (c >= src.channels[k]) { c -= src.channels[k]; ++k; }           // k < src.n by construction of c_total
    const float* in = src.ptr[k] + ((size_t)b * src.channels[k] + c) * hw;
    float* o = out + (size_t)plane * hw;
    const bool drop = dropped && dropped[b];
    const size_t hw4 = hw >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(o)[i] = drop ? f32x4{0.f, 0.f, 0.f, 0.f} : reinterpret_cast<const f32x4*>(in)[i];
    const size_t t = (hw4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < hw) o[t] = drop ? 0.f : in[t];
}

// geo field [B][cin][HW] -> [B][2][HW]; cin == 1: append the mask plane, cin == 2: copy
__global__ __launch_bounds__(256) void assemble_geo_kernel(const float* __restrict__ in, int cin,
                                                           const unsigned char* __restrict__ dropped, float* __restrict__ out,
                                                           size_t hw) {
    const int plane = blockIdx.y;                    // b * 2 + c
    const int b = plane >> 1, c = plane & 1;
    float* o = out + (size_t)plane * hw;
    const bool copy = c == 0 || cin == 2;
    const float* src = in + ((size_t)b * cin + (cin == 2 ? c : 0)) * hw;
    const float fill = (dropped && dropped[b]) ? 0.f : 1.f;
    const size_t hw4 = hw >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(o)[i] = copy ? reinterpret_cast<const f32x4*>(src)[i] : f32x4{fill, fill, fill, fill};
    const size_t t = (hw4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < hw) o[t] = copy ? src[t] : fill;
}

__global__ void assemble_labels_kernel(const long long* __restrict__ y, const unsigned char* __restrict__ dropped,
                                       long long* __restrict__ out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = (dropped && dropped[b]) ? 0 : y[b];
}

}  // namespace

int sbgm_launch_assemble_conditions(const sbgm_assemble_args& a, hipStream_t st) {
    SBGM_CHECK(a.B >= 1 && a.HW >= 1, "assemble_conditions: B=%d HW=%lld", a.B, (long long)a.HW);
    SBGM_CHECK(a.HW % 4 == 0, "assemble_conditions: H*W=%lld must be a multiple of 4 (16-byte planes)", (long long)a.HW);
    SBGM_CHECK(a.n_lr >= 0 && a.n_lr <= SBGM_ASSEMBLE_MAX_LR, "assemble_conditions: %d LR fields (max %d)", a.n_lr,
               SBGM_ASSEMBLE_MAX_LR);
    const size_t hw = (size_t)a.HW;
    const int bx = (int)std::min<size_t>((hw / 4 + 255) / 256, 64);
    if (a.n_lr > 0) {
        SBGM_CHECK(a.lr_out, "assemble_conditions: lr_out is required with LR fields");
        AssembleSrc src{};
        int c_total = 0;
        for (int k = 0; k < a.n_lr; ++k) {
            SBGM_CHECK(a.lr[k] && a.lr_channels[k] >= 1, "assemble_conditions: LR field %d is null or has no channels", k);
            src.ptr[k] = a.lr[k];
            src.channels[k] = a.lr_channels[k];
            c_total += a.lr_channels[k];
        }
        src.n = a.n_lr;
        SBGM_CHECK((long long)a.B * c_total <= 65535, "assemble_conditions: B*C=%lld planes exceed the grid", (long long)a.B * c_total);
        hipLaunchKernelGGL(assemble_lr_kernel, dim3(bx, a.B * c_total), dim3(256), 0, st, src, a.dropped, a.lr_out, c_total, hw);
        SBGM_LAUNCH_CHECK();
    }
    const float* geo_in[2] = {a.lsm, a.topo};
    float* geo_out[2] = {a.lsm_out, a.topo_out};
    const int geo_c[2] = {a.lsm_channels, a.topo_channels};
    for (int g = 0; g < 2; ++g) {
        if (!geo_in[g]) continue;
        SBGM_CHECK(geo_out[g], "assemble_conditions: geo output %d is required", g);
        SBGM_CHECK(geo_c[g] == 1 || geo_c[g] == 2, "assemble_conditions: geo field %d has %d channels (1 = value, 2 = value||mask)", g,
                   geo_c[g]);
        hipLaunchKernelGGL(assemble_geo_kernel, dim3(bx, a.B * 2), dim3(256), 0, st, geo_in[g], geo_c[g], a.dropped, geo_out[g], hw);
        SBGM_LAUNCH_CHECK();
    }
    if (a.y) {
        SBGM_CHECK(a.y_out, "assemble_conditions: y_out is required with labels");
        hipLaunchKernelGGL(assemble_labels_kernel, dim3((a.B + 255) / 256), dim3(256), 0, st, (const long long*)a.y, a.dropped,
                           (long long*)a.y_out, a.B);
        SBGM_LAUNCH_CHECK();
    }
    return 0;
}
