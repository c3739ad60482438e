// Microbenchmark (development tool): what bounds the Winograd sweep — LDS operand reads or the matrix pipe?
// A synthetic stage loop with the instruction mix of conv3x3_w2dp_kernel's sweep and nothing else (no staging, no transform,
// no epilogue): per stage G groups of {NA A-fragment ds_read_b128, NP/4 patch ds_read_b128, NM MFMAs}, NBAR workgroup barriers.
//   MODE 0: v_mfma_f32_16x16x4_f32, 16 groups x 8 MFMAs (FCO = 2: 128 accumulator registers)        -> the shipped structure
//   MODE 1: v_mfma_f32_32x32x2_f32,  8 groups x 8 MFMAs (8 products x 32 co x 32 blocks: 128 registers) -> half the operand reads
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/sweep_bench.hip -o tools/micro/bin/sweep_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NA, int NP, int NBAR, int MINW>
__global__ __launch_bounds__(256, MINW) void sweep(float* out, int stages) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* L = reinterpret_cast<f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3072; i += 256) L[i] = f32x4{(float)(i & 7) * 0.25f, 1.f, -0.5f, 0.125f * (float)(lane & 3)};
    __syncthreads();
    constexpr int G = MODE == 0 ? 16 : 8;
    f32x4 acc4[MODE == 0 ? 32 : 1];
    f32x16 acc16[MODE == 1 ? 8 : 1];
#pragma unroll
    for (int i = 0; i < (MODE == 0 ? 32 : 1); ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (MODE == 1 ? 8 : 1); ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;
    const f32x4* base = L + lane;                    // 64 consecutive quads per instruction: conflict-free
    for (int s = 0; s < stages; ++s) {
        const f32x4* st = base + (s & 1) * 64;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 V[4] = {st[0], st[64], st[128], st[192]};
#pragma unroll
            for (int gg = 0; gg < G / 2; ++gg) {
                const int g = half * (G / 2) + gg;
                f32x4 a[2];
                a[0] = st[(g * 2) * 64 + 256];
                a[1] = NA > 1 ? st[(g * 2 + 1) * 64 + 256] : a[0];
                if ((gg & 3) == 0) {                 // a new xi: NP patch reads folded into the 4 B operands
#pragma unroll
                    for (int u = 0; u < NP; ++u) V[u & 3] += st[((g + u) & 31) * 64 + 512];
                }
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc4[g * 2 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][k], V[gg & 3][k], acc4[g * 2 + i], 0, 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc16[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], V[(gg + i) & 3][k], acc16[g], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (half < NBAR) __syncthreads();
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < (MODE == 0 ? 32 : 1); ++i) r += acc4[i][0] + acc4[i][3];
#pragma unroll
    for (int i = 0; i < (MODE == 1 ? 8 : 1); ++i) r += acc16[i][0] + acc16[i][15];
    if (r == 123.456f) out[tid] = r;
}

template <int MODE, int NA, int NP, int NBAR, int MINW>
void run(float* out, int wgs_per_cu, int lds_kb) {
    const int stages = 4000;
    auto k = sweep<MODE, NA, NP, NBAR, MINW>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256 * wgs_per_cu), dim3(256), lds_kb * 1024, 0, out, 20);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256 * wgs_per_cu), dim3(256), lds_kb * 1024, 0, out, stages);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double mf = MODE == 0 ? 128 * 2048.0 : 64 * 4096.0;         // flops per wave and stage
    const double flops = 256.0 * wgs_per_cu * 4 * stages * mf;
    printf("mode %d  A reads/group %d  patch reads/xi %2d  barriers/stage %d  WGs/CU %d (minw %d): %7.2f ms  %6.1f TF-mfma\n", MODE, NA, NP, NBAR,
           wgs_per_cu, MINW, best, flops / best / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    // the shipped mix, then with operand reads removed, then without barriers
    run<0, 2, 8, 2, 2>(out, 2, 64);
    run<0, 2, 8, 0, 2>(out, 2, 64);
    run<0, 1, 8, 2, 2>(out, 2, 64);
    run<0, 2, 0, 2, 2>(out, 2, 64);
    run<0, 1, 0, 2, 2>(out, 2, 64);
    run<0, 1, 0, 0, 2>(out, 2, 64);
    // 32x32x2: 8 products per wave; 2 A reads per product, 12 patch reads per xi
    run<1, 2, 12, 2, 2>(out, 2, 64);
    run<1, 2, 12, 0, 2>(out, 2, 64);
    run<1, 2, 0, 2, 2>(out, 2, 64);
    run<1, 1, 0, 0, 2>(out, 2, 64);
    // one workgroup per CU
    run<0, 2, 8, 2, 1>(out, 1, 64);
    run<1, 2, 12, 2, 1>(out, 1, 64);
    return 0;
}
