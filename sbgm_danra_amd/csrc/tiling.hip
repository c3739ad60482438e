// Full-domain tiling row of SURVEY.md §8f (rank 3): BASELINE config 5 samples a 589x789 DANRA field as overlapping
// 256x256 tiles.  The reference holds only the domain dimensions (config/full_run_config_new.yaml:26,28) and no tiling,
// halo or stitching code, so this is new capability with its specification in DESIGN.md §9; there is no reference oracle
// beyond per-tile parity of the sampler itself.
//
// K34 extract_tiles_kernel  — gather: tiles[t][c][y][x] = domain[c][y0_t + y][x0_t + x]
// K35 stitch_tiles_kernel   — per output pixel, the normalised blend of every tile that covers it:
//         out = sum_t w_t * tile_t / sum_t w_t,   w_t = wy * wx,
//         w(i) = min(d_lo, d_hi, R) / R  with d_lo = i + 1, d_hi = L - i, except that an edge lying on the domain boundary
//         does not ramp (there is no neighbour to blend with).  Gather form: deterministic, no atomics.
// Both are HBM-bound copies (8 B per element moved).
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void extract_tiles_kernel(const float* __restrict__ dom, const int* __restrict__ origins,
                                                            float* __restrict__ tiles, int C, int Hd, int Wd, int th, int tw,
                                                            size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % tw);
        size_t r = i / tw;
        const int y = (int)(r % th); r /= th;
        const int c = (int)(r % C);
        const int t = (int)(r / C);
        tiles[i] = dom[((size_t)c * Hd + origins[2 * t] + y) * Wd + origins[2 * t + 1] + x];
    }
}

__device__ __forceinline__ float ramp(int i, int L, int origin, int dom_len, int R) {
    const int lo = origin == 0 ? R : i + 1;                    // distance to the tile's low edge (no ramp on the domain edge)
    const int hi = origin + L == dom_len ? R : L - i;
    return (float)min(min(lo, hi), R) / (float)R;
}

__global__ __launch_bounds__(256) void stitch_tiles_kernel(const float* __restrict__ tiles, const int* __restrict__ origins,
                                                           float* __restrict__ dom, int T, int C, int Hd, int Wd, int th, int tw,
                                                           int R, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % Wd);
        size_t r = i / Wd;
        const int Y = (int)(r % Hd);
        const int c = (int)(r / Hd);
        float acc = 0.f, wsum = 0.f;
        for (int t = 0; t < T; ++t) {
            const int y = Y - origins[2 * t], x = X - origins[2 * t + 1];
            if (y < 0 || y >= th || x < 0 || x >= tw) continue;
            const float w = ramp(y, th, origins[2 * t], Hd, R) * ramp(x, tw, origins[2 * t + 1], Wd, R);
            acc += w * tiles[(((size_t)t * C + c) * th + y) * tw + x];
            wsum += w;
        }
        dom[i] = wsum > 0.f ? acc / wsum : 0.f;
    }
}

}  // namespace

static int check_tiling(int T, int C, int Hd, int Wd, int th, int tw) {
    SBGM_CHECK(T >= 1 && C >= 1 && th >= 1 && tw >= 1 && th <= Hd && tw <= Wd, "tiling: T=%d C=%d tile %dx%d domain %dx%d", T, C, th,
               tw, Hd, Wd);
    return 0;
}

int sbgm_launch_extract_tiles(const float* dom, const int* origins, float* tiles, int T, int C, int Hd, int Wd, int th, int tw,
                              hipStream_t st) {
    if (check_tiling(T, C, Hd, Wd, th, tw)) return 1;
    const size_t total = (size_t)T * C * th * tw;
    hipLaunchKernelGGL(extract_tiles_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, st, dom, origins,
                       tiles, C, Hd, Wd, th, tw, total);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_stitch_tiles(const float* tiles, const int* origins, float* dom, int T, int C, int Hd, int Wd, int th, int tw,
                             int ramp_len, hipStream_t st) {
    if (check_tiling(T, C, Hd, Wd, th, tw)) return 1;
    SBGM_CHECK(ramp_len >= 1, "stitch_tiles: ramp length %d must be >= 1", ramp_len);
    const size_t total = (size_t)C * Hd * Wd;
    hipLaunchKernelGGL(stitch_tiles_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, st, tiles, origins,
                       dom, T, C, Hd, Wd, th, tw, ramp_len, total);
    SBGM_LAUNCH_CHECK();
    return 0;
}
