// Backward kernels of the training path (reference: everything `loss.backward()` differentiates in
// sbgm/training.py:403-405, i.e. the autograd of the ops in sbgm/score_unet.py).  Data-gradients of the
// convolutions reuse conv_igemm.hip with transposed/flipped packed weights; this file holds the weight-gradient
// GEMM, the normalisation / attention / resampling backward passes and the small reductions.
// First version: correct and parity-tested; only the weight-gradient kernel is on the MFMA pipe.
#include <cstdlib>
#include <vector>
#include "common.h"
#include "kernels.h"

int sbgm_scratch_prezeroed = 0;

namespace {

inline int stream_blocks(size_t n, int cap = 2048) { return (int)std::min<size_t>((n + 255) / 256, (size_t)cap); }

// =====================================================================================================================
// Convolution weight gradient:  dW[tap][co][ci] += sum_p dy[p][co] * x[p @ tap][ci]        (fp32 MFMA 16x16x4)
// GEMM roles: rows = output channels, cols = input channels, K = pixels.  A lane's 16-byte dy load holds 4 CHANNELS of
// one pixel, so MFMA i of a group uses element i and owns the rows {4r+i}: 4 accumulator sets cover 64 channels, the
// K index of the instruction is the pixel (lane>>4).  x is read one dword per lane (16 consecutive channels of the
// tap-shifted pixel).  Workgroups split the pixel range; partial sums are combined with fp32 atomics.
// =====================================================================================================================
template <int FCI>   // 16*FCI input channels per wave
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         float* __restrict__ dwp, int B, int H, int W, int Cs, int OH,
                                                         int OW, int Cout, int KH, int KW, int S, int PAD, int px_per_split,
                                                         float* __restrict__ dbias) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int n_ci_t = (Cs + 16 * FCI - 1) / (16 * FCI), n_co_t = Cout / 64;
    int tile = blockIdx.x * 4 + wave;
    if (tile >= KH * KW * n_co_t * n_ci_t) return;
    const int ci_t = tile % n_ci_t; tile /= n_ci_t;
    const int co_t = tile % n_co_t; tile /= n_co_t;
    const int tap = tile;
    const int kh = tap / KW, kw = tap - kh * KW;
    const int co0 = co_t * 64, ci0 = ci_t * 16 * FCI;
    const int M = B * OH * OW;
    const int p_begin = blockIdx.y * px_per_split;
    const int p_end = min(M, p_begin + px_per_split);

    f32x4 acc[4][FCI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FCI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operands of the 4-pixel group starting at p0 (this lane's pixel p0 + kq is the MFMA k index); zeros past the range / image
    const int OHW = OH * OW;
    auto load = [&](int p0, f32x4& a, float (&bv)[FCI]) {
        const int p = p0 + kq;
        a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < FCI; ++j) bv[j] = 0.f;
        if (p < p_end) {
            a = *reinterpret_cast<const f32x4*>(dy + (size_t)p * Cout + co0 + 4 * r16);
            const int b = p / OHW;
            const int rr = p - b * OHW;
            const int oy = rr / OW, ox = rr - oy * OW;
            const int iy = oy * S - PAD + kh, ix = ox * S - PAD + kw;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                // (a 16-byte x load per lane was measured: the lane->channel map it forces makes the epilogue's atomics strided
                //  and the step 30 % slower; the dword loads keep every atomic instruction on 64 contiguous bytes per row)
                const float* xp = x + (((size_t)b * H + iy) * W + ix) * Cs + ci0 + r16;
#pragma unroll
                for (int j = 0; j < FCI; ++j)
                    if (ci0 + 16 * j + r16 < Cs) bv[j] = xp[16 * j];
            }
        }
    };
    // bias gradient db[co] = sum_p dy[p][co] as a by-product: the waves of tap 0 / input-channel tile 0 already stream every dy
    // row of their output-channel slice once (removes a separate column-sum launch and its memset per convolution)
    const bool do_bias = dbias != nullptr && tap == 0 && ci_t == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const f32x4& a, const float (&bv)[FCI]) {
        if (do_bias) bsum += a;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < FCI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[i][j], 0, 0, 0);
    };
    // two register sets: the loads of the next group are in flight while the matrix pipe works on the current one
    f32x4 a0, a1;
    float b0[FCI], b1[FCI];
    load(p_begin, a0, b0);
    for (int p0 = p_begin; p0 < p_end; p0 += 8) {
        load(p0 + 4, a1, b1);
        mma(a0, b0);
        load(p0 + 8, a0, b0);
        mma(a1, b1);
    }
    if (do_bias) {                                   // lanes kq = 0..3 hold the same channels 4*r16..+3 for different pixels
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bsum[e] += __shfl_xor(bsum[e], 16, 64);
            bsum[e] += __shfl_xor(bsum[e], 32, 64);
        }
        if (kq == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dbias + co0 + 4 * r16 + e, bsum[e]);
        }
    }
    // D_i[row][col]: row = 4*kq + reg -> channel co0 + 4*row + i ; col = r16 -> ci0 + 16j + r16
    float* base = dwp + ((size_t)tap * Cout) * Cs;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FCI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + 4 * (4 * kq + e) + i;
                if (ci0 + 16 * j + r16 < Cs) atomicAdd(base + (size_t)co * Cs + ci0 + 16 * j + r16, acc[i][j][e]);
            }
}

// =====================================================================================================================
// Weight gradient of a 3x3 / stride 1 / pad 1 convolution on maps whose sides are multiples of 16 (or 8 x 8), LDS-staged (the layers
// that dominate the backward pass: K = B*H*W pixels is long, Cout x Cin is small).  A workgroup of 8 waves owns ONE
// 32 x 32 (co x ci) block of all 9 taps and walks over 16 x 16 pixel tiles of its share of the images: per tile it stages
// dy[256 px][32 co] and the halo patch x[18 x 18 px][32 ci] in LDS once (16-byte global loads, zero padding from the
// buffer bounds check), then wave w accumulates rows 2w, 2w+1 of the tile into its own copy of the block with fp32 MFMAs
// whose K index is the pixel: 2 + 18 LDS dword reads feed 36 MFMAs per 4-pixel step.  The wave-level kernel above re-reads
// dy 9x and x 9x from global memory and spends most of its time on per-pixel index arithmetic.
// The 8 copies are folded through LDS tap by tap, so one workgroup sends 36 KB of fp32 atomics into the pre-zeroed
// [tap][Cout][Cs] slab, from which unpack_wgrad_kernel writes OIHW (a layer without a pixel split stores OIHW directly and
// needs neither).  That volume is what the epilogue costs (the atomic units take ~1.5 TB/s for 64-byte runs and longer;
// 4-byte scatter straight into OIHW costs twice as much): with 64 x 64 blocks split over 4 quarter-waves it was 4x larger
// and cost more than the matrix work at training batch sizes.
// Pixel stride in LDS = 48 floats: lanes (r16, kq) of a ds_read_b32 then fall on 64 distinct banks.
// =====================================================================================================================
constexpr int WG_PS = 48;                          // LDS floats per pixel
constexpr int WG_THREADS = 512;                    // 8 waves = 8 pixel groups
constexpr int WG_DY_FLOATS = 256 * WG_PS;
// TW = 16: tiles are 16 x 16 pixels of one image.  TW = 8 (8 x 8 maps): a tile is 4 whole images stacked, each with its own
// zero halo in the patch (4 x 10 rows of 10 pixels).
template <int TW> struct WgTile {
    static constexpr int PW = TW + 2, PH = TW == 16 ? 18 : 40, X_FLOATS = PH * PW * WG_PS;
};
// one launch's arguments; the batched form (every 3x3 weight gradient of a backward sweep in one launch per tile geometry) carries an array
struct HaloDesc {
    const float *dy, *x;
    float *dwp, *dbias, *dw_direct;
    int B, H, W, Cs, Cout, tiles_per_wg, Cin, gx, gy;
    uint32_t dy_bytes, x_bytes;
};

template <int TW>
__device__ __forceinline__ void halo_wgrad_body(const HaloDesc& a, const int bx, const int by) {
    const float* __restrict__ dy = a.dy;
    const float* __restrict__ x = a.x;
    float* __restrict__ dwp = a.dwp;
    float* __restrict__ dbias = a.dbias;
    float* __restrict__ dw_direct = a.dw_direct;
    const int B = a.B, H = a.H, W = a.W, Cs = a.Cs, Cout = a.Cout, tiles_per_wg = a.tiles_per_wg, Cin = a.Cin;
    const uint32_t dy_bytes = a.dy_bytes, x_bytes = a.x_bytes;
    constexpr int PW = WgTile<TW>::PW, PH = WgTile<TW>::PH;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* dys = reinterpret_cast<float*>(smem_raw);                   // [256 px][48]
    float* xs = dys + WG_DY_FLOATS;                                    // [PH][PW][48]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int n_ci = Cs / 32;
    const int co0 = (bx / n_ci) * 32, ci0 = (bx % n_ci) * 32;
    const int tiles_x = W / 16, tiles_img = tiles_x * (H / 16);        // TW = 16 only
    const int n_tiles = TW == 16 ? B * tiles_img : (B + 3) / 4;
    const int t_begin = by * tiles_per_wg, t_end = min(n_tiles, t_begin + tiles_per_wg);
    const __amdgpu_buffer_rsrc_t dr = make_rsrc(dy, dy_bytes);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, x_bytes);

    f32x4 acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = dbias != nullptr && ci0 == 0;
    float bsum[2] = {0.f, 0.f};

    // register-staged double buffering: the global loads of tile t+1 are in flight while the matrix pipe works on tile t
    constexpr int DY_QUADS = 256 * 8, X_QUADS = PH * PW * 8;
    constexpr int DYQ = DY_QUADS / WG_THREADS;                          // 4 dy quads per thread
    constexpr int XQ = (X_QUADS + WG_THREADS - 1) / WG_THREADS;         // 6 (7) x quads per thread
    f32x4 rdy[DYQ], rx[XQ];
    auto tile_load = [&](int t) {
        int b0, y0 = 0, x0 = 0;
        if (TW == 16) {
            b0 = t / tiles_img;
            const int r = t - b0 * tiles_img;
            y0 = (r / tiles_x) * 16;
            x0 = (r % tiles_x) * 16;
        } else {
            b0 = t * 4;
        }
#pragma unroll
        for (int u = 0; u < DYQ; ++u) {
            const int q = tid + WG_THREADS * u;
            const int px = q >> 3, c4 = (q & 7) * 4;
            int b, oy, ox;
            if (TW == 16) { b = b0; oy = y0 + (px >> 4); ox = x0 + (px & 15); }
            else { b = b0 + (px >> 6); oy = (px >> 3) & 7; ox = px & 7; }
            rdy[u] = buf_load4(dr, b < B ? (uint32_t)(((b * H + oy) * W + ox) * Cout + co0 + c4) * 4u : 0x80000000u);
        }
#pragma unroll
        for (int u = 0; u < XQ; ++u) {
            const int q = tid + WG_THREADS * u;
            const int pp = q >> 3, c4 = (q & 7) * 4;
            const int py = pp / PW, pxx = pp - py * PW;
            int b, iy;
            if (TW == 16) { b = b0; iy = y0 - 1 + py; }
            else { const int img = py / 10; b = b0 + img; iy = py - img * 10 - 1; }
            const int ix = x0 - 1 + pxx;
            const bool ok = (q < X_QUADS) & (b < B) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
            rx[u] = buf_load4(xr, ok ? (uint32_t)(((b * H + iy) * W + ix) * Cs + ci0 + c4) * 4u : 0x80000000u);
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int u = 0; u < DYQ; ++u) {
            const int q = tid + WG_THREADS * u;
            *reinterpret_cast<f32x4*>(dys + (q >> 3) * WG_PS + (q & 7) * 4) = rdy[u];
        }
#pragma unroll
        for (int u = 0; u < XQ; ++u) {
            const int q = tid + WG_THREADS * u;
            if (q < X_QUADS) *reinterpret_cast<f32x4*>(xs + (q >> 3) * WG_PS + (q & 7) * 4) = rx[u];
        }
    };
    if (t_begin < t_end) tile_load(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();                                              // the previous tile's reads are done
        tile_store();
        __syncthreads();
        if (t + 1 < t_end) tile_load(t + 1);
#pragma unroll 2
        for (int s = 0; s < 8; ++s) {                                  // this wave's 32 pixels: 8 steps of 4 consecutive pixels
            // this lane's pixel = the MFMA k index; prow = its row in the halo patch for kh = 0
            const int row = TW == 16 ? 2 * wave + (s >> 2) : 4 * wave + (s >> 1);
            const int col = (TW == 16 ? (s & 3) : (s & 1)) * 4 + kq;
            const int prow = TW == 16 ? row : row + 2 * (row >> 3);
            float a[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = dys[(row * TW + col) * WG_PS + 16 * i + r16];
            if (do_bias) { bsum[0] += a[0]; bsum[1] += a[1]; }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* xp = xs + ((prow + kh) * PW + col + kw) * WG_PS + r16;
                    bv[0] = xp[0];
                    bv[1] = xp[16];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[kh * 3 + kw][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[kh * 3 + kw][i][j], 0, 0, 0);
                }
        }
    }
    if (do_bias) {                                   // lane (r16, kq) summed channel 16i+r16 over its pixels: fold the 4 kq groups
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bsum[i] += __shfl_xor(bsum[i], 16, 64);
            bsum[i] += __shfl_xor(bsum[i], 32, 64);
            if (kq == 0) atomicAdd(dbias + co0 + 16 * i + r16, bsum[i]);
        }
    }
    // Fold the 8 copies through LDS tap by tap (the staging area is free now; two alternating 32 KB regions, one barrier per
    // tap); wave t % 8 sums tap t and sends its atomics.  D[row][col]: row = 4*kq + reg -> co, col = r16 -> ci
    f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                   // [2][8 waves][4 (i,j)][64 lanes] x f32x4
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        f32x4* region = red + (t & 1) * 8 * 4 * 64;
        if (wave != (t & 7)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) region[(wave * 4 + i * 2 + j) * 64 + lane] = acc[t][i][j];
        }
        __syncthreads();
        if (wave == (t & 7)) {
            float* base = dwp + ((size_t)t * Cout) * Cs;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 o = acc[t][i][j];
#pragma unroll
                    for (int w = 0; w < 8; ++w)
                        if (w != (t & 7)) {
                            const f32x4 v = region[(w * 4 + i * 2 + j) * 64 + lane];
                            o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
                        }
                    if (dw_direct) {                  // this workgroup is the only writer of its block: straight into OIHW
                        const int ci = ci0 + 16 * j + r16;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ci < Cin) dw_direct[((size_t)(co0 + 16 * i + 4 * kq + e) * Cin + ci) * 9 + t] = o[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            atomicAdd(base + (size_t)(co0 + 16 * i + 4 * kq + e) * Cs + ci0 + 16 * j + r16, o[e]);
                    }
                }
        }
    }
}

template <int TW>
__global__ __launch_bounds__(WG_THREADS) void conv3x3_wgrad_lds_kernel(const HaloDesc a) { halo_wgrad_body<TW>(a, blockIdx.x, blockIdx.y); }

// Batched form (see the per-tap kernel's batched form below for the rationale): at batch 8 most of a step's 19 3x3 layers own 2-8 tiles
// per workgroup before the 9-tap LDS fold and 36 KB of atomics; launched together they split their pixels over far fewer workgroups.
constexpr int HALO_MAX = 30;
struct HaloTable {
    HaloDesc d[HALO_MAX];
    int start[HALO_MAX + 1];
    int n;
};
template <int TW>
__global__ __launch_bounds__(WG_THREADS) void conv3x3_wgrad_batched_kernel(const HaloTable t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.start[k + 1]) ++k;
    k = __builtin_amdgcn_readfirstlane(k);
    const int rel = blockIdx.x - t.start[k];
    const int gx = t.d[k].gx;
    halo_wgrad_body<TW>(t.d[k], rel % gx, rel / gx);
}

// =====================================================================================================================
// Weight gradient as one split-K GEMM per kernel tap, LDS-staged: dW[tap][co][ci] = sum_m dy[m][co] * x[pixel(m, tap)][ci].
// Serves 1x1 convolutions and nn.Linear (one tap), and every other geometry the halo kernel above does not take (3x3
// stride 2, 3x3 on 4x4 maps, the 8x8 stride-2 stem convolutions): a tap only shifts the input pixel, out-of-range pixels
// load zeros.  A workgroup of 8 waves owns one (tap, 64 co, 64 ci) block; per tile of 128 output pixels it stages
// dy[128][64] and x[128][64] once, wave w takes pixels 16w..16w+15 (4 MFMA k-steps) and keeps its own 64 x 64 partial
// block (16 accumulator quads: 8 LDS dword reads feed 16 MFMAs, 128 staged bytes per MFMA).  The 8 partial blocks fold
// through LDS; one atomic (or, without a pixel split, one plain OIHW store) per element and workgroup.
// stem != 0 (padded Cin = 8, KW = 8): the 64 "channels" of a block are the (kw, ci) pairs of one kernel row kh, which are
// 64 contiguous floats of the NHWC input row, and the slab is [kh][Cout][kw][8].
// =====================================================================================================================
constexpr int W1_PX = 128, W1_PS = 80;             // pixels per tile, LDS floats per pixel (64 channels + pad: conflict-free)
struct TapGeom {
    int Mo, OH, OW, H, W, stride, pad, KW, taps, stem;
};
// one launch's arguments; the batched form (every small weight gradient of a backward sweep in ONE launch, below) carries an array of them
struct TapDesc {
    const float *dy, *x;
    float *dwp, *dbias, *dw_direct;
    TapGeom g;
    int Cs, Cout, tiles_per_wg, Cin, gx, gy;
    uint32_t dy_bytes, x_bytes;
    int big;               // 1: 128 x 128 blocks over 64-pixel tiles (tap_wgrad_body128), 0: 64 x 64 blocks over 128-pixel tiles
};

__device__ __forceinline__ void tap_wgrad_body(const TapDesc& a, const int bx, const int by) {
    const float* __restrict__ dy = a.dy;
    const float* __restrict__ x = a.x;
    float* __restrict__ dwp = a.dwp;
    float* __restrict__ dbias = a.dbias;
    float* __restrict__ dw_direct = a.dw_direct;
    const TapGeom g = a.g;
    const int Cs = a.Cs, Cout = a.Cout, tiles_per_wg = a.tiles_per_wg, Cin = a.Cin;
    const uint32_t dy_bytes = a.dy_bytes, x_bytes = a.x_bytes;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* dys = reinterpret_cast<float*>(smem_raw);                   // [128 px][80]
    float* xs = dys + W1_PX * W1_PS;                                   // [128 px][80]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int n_ci = g.stem ? 1 : Cs / 64, n_co = Cout / 64;
    int blk = bx;
    const int ci0 = (blk % n_ci) * 64; blk /= n_ci;
    const int co0 = (blk % n_co) * 64;
    const int tap = blk / n_co;
    const int kh = g.stem ? tap : tap / g.KW, kw0 = g.stem ? 0 : tap - kh * g.KW;
    const int slab_cs = g.stem ? 64 : Cs;                              // floats per (tap, co) row of the slab
    dwp += (size_t)tap * Cout * slab_cs;
    const int n_tiles = (g.Mo + W1_PX - 1) / W1_PX;
    const int t_begin = by * tiles_per_wg, t_end = min(n_tiles, t_begin + tiles_per_wg);
    const __amdgpu_buffer_rsrc_t dr = make_rsrc(dy, dy_bytes);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, x_bytes);
    const bool shifted = g.stride != 1 || g.taps != 1 || g.pad != 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = dbias != nullptr && ci0 == 0 && tap == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    constexpr int QPT = W1_PX * 16 / WG_THREADS;                       // 4 quads of each operand per thread
    f32x4 rdy[QPT], rx[QPT];
    auto tile_load = [&](int t) {
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            const int q = tid + WG_THREADS * u;
            const int p = t * W1_PX + (q >> 4), c4 = (q & 15) * 4;
            bool ok = p < g.Mo;
            uint32_t xoff = (uint32_t)(p * Cs + ci0 + c4);
            if (shifted) {
                const int b = p / (g.OH * g.OW), r = p - b * (g.OH * g.OW);
                const int oy = r / g.OW, ox = r - oy * g.OW;
                const int iy = oy * g.stride - g.pad + kh;
                const int ix = ox * g.stride - g.pad + (g.stem ? (c4 >> 3) : kw0);
                xoff = (uint32_t)(((b * g.H + iy) * g.W + ix) * Cs + (g.stem ? (c4 & 7) : ci0 + c4));
                rdy[u] = buf_load4(dr, ok ? (uint32_t)(p * Cout + co0 + c4) * 4u : 0x80000000u);
                ok = ok & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
            } else {
                rdy[u] = buf_load4(dr, ok ? (uint32_t)(p * Cout + co0 + c4) * 4u : 0x80000000u);
            }
            rx[u] = buf_load4(xr, ok ? xoff * 4u : 0x80000000u);
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            const int q = tid + WG_THREADS * u;
            *reinterpret_cast<f32x4*>(dys + (q >> 4) * W1_PS + (q & 15) * 4) = rdy[u];
            *reinterpret_cast<f32x4*>(xs + (q >> 4) * W1_PS + (q & 15) * 4) = rx[u];
        }
    };
    if (t_begin < t_end) tile_load(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();                                              // the previous tile's reads are done
        tile_store();
        __syncthreads();
        if (t + 1 < t_end) tile_load(t + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {                                  // this wave's 16 pixels: 4 steps of 4 pixels
            const int px = 16 * wave + 4 * s + kq;                      // this lane's pixel = the MFMA k index
            float a[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = dys[px * W1_PS + 16 * i + r16];
                bv[i] = xs[px * W1_PS + 16 * i + r16];
            }
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bsum[i] += a[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bsum[i] += __shfl_xor(bsum[i], 16, 64);
            bsum[i] += __shfl_xor(bsum[i], 32, 64);
            if (kq == 0) atomicAdd(dbias + co0 + 16 * i + r16, bsum[i]);
        }
    }
    // Fold the 8 partial blocks through LDS in two halves of 8 sub-blocks (64 KB each): every wave stores its copy of the
    // half, then wave w sums sub-block w of it over the 8 copies.  D[row][col]: row = 4*kq + reg -> co, col = r16 -> ci
    f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                   // [8 waves][8 sub-blocks][64 lanes] x f32x4
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                                              // staging reads / the previous half's reads are done
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(wave * 8 + i * 4 + j) * 64 + lane] = acc[2 * h + i][j];
        __syncthreads();
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const f32x4 v = red[(w * 8 + wave) * 64 + lane];
            o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
        }
        const int i = 2 * h + (wave >> 2), j = wave & 3;
        const int ci = ci0 + 16 * j + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co0 + 16 * i + 4 * kq + e;
            if (dw_direct) { if (ci < Cin) dw_direct[((size_t)co * Cin + ci) * g.taps + tap] = o[e]; }
            else atomicAdd(dwp + (size_t)co * slab_cs + ci, o[e]);
        }
    }
}

// The same GEMM on 128 (co) x 128 (ci) blocks: wave w owns the 64 x 64 sub-block (w & 1, (w >> 1) & 1) over one half (w >> 2) of a
// 64-pixel tile, so a staged byte feeds twice the MFMAs of the 64 x 64 form (32 instead of 16 flop per byte: that form re-reads dy once
// per input-channel block and runs at the L2's rate) and only TWO partial copies fold through LDS.  Cs and Cout multiples of 128.
constexpr int W2_PX = 64, W2_PS = 144;
__device__ __forceinline__ void tap_wgrad_body128(const TapDesc& a, const int bx, const int by) {
    const TapGeom g = a.g;
    const int Cs = a.Cs, Cout = a.Cout;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* dys = reinterpret_cast<float*>(smem_raw);                   // [64 px][144]
    float* xs = dys + W2_PX * W2_PS;                                   // [64 px][144]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int wi = wave & 1, wj = (wave >> 1) & 1, ph = wave >> 2;
    const int n_ci = Cs / 128, n_co = Cout / 128;
    int blk = bx;
    const int ci0 = (blk % n_ci) * 128; blk /= n_ci;
    const int co0 = (blk % n_co) * 128;
    const int tap = blk / n_co;
    const int kh = tap / g.KW, kw0 = tap - kh * g.KW;
    float* dwp = a.dwp + (size_t)tap * Cout * Cs;
    const int n_tiles = (g.Mo + W2_PX - 1) / W2_PX;
    const int t_begin = by * a.tiles_per_wg, t_end = min(n_tiles, t_begin + a.tiles_per_wg);
    const __amdgpu_buffer_rsrc_t dr = make_rsrc(a.dy, a.dy_bytes);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes);
    const bool shifted = g.stride != 1 || g.taps != 1 || g.pad != 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = a.dbias != nullptr && ci0 == 0 && tap == 0 && wj == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    constexpr int QPT = W2_PX * 32 / WG_THREADS;                       // 4 quads of each operand per thread
    f32x4 rdy[QPT], rx[QPT];
    auto tile_load = [&](int t) {
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            const int q = tid + WG_THREADS * u;
            const int p = t * W2_PX + (q >> 5), c4 = (q & 31) * 4;
            bool ok = p < g.Mo;
            uint32_t xoff = (uint32_t)(p * Cs + ci0 + c4);
            rdy[u] = buf_load4(dr, ok ? (uint32_t)(p * Cout + co0 + c4) * 4u : 0x80000000u);
            if (shifted) {
                const int b = p / (g.OH * g.OW), r = p - b * (g.OH * g.OW);
                const int oy = r / g.OW, ox = r - oy * g.OW;
                const int iy = oy * g.stride - g.pad + kh, ix = ox * g.stride - g.pad + kw0;
                xoff = (uint32_t)(((b * g.H + iy) * g.W + ix) * Cs + ci0 + c4);
                ok = ok & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
            }
            rx[u] = buf_load4(xr, ok ? xoff * 4u : 0x80000000u);
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            const int q = tid + WG_THREADS * u;
            *reinterpret_cast<f32x4*>(dys + (q >> 5) * W2_PS + (q & 31) * 4) = rdy[u];
            *reinterpret_cast<f32x4*>(xs + (q >> 5) * W2_PS + (q & 31) * 4) = rx[u];
        }
    };
    if (t_begin < t_end) tile_load(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        tile_store();
        __syncthreads();
        if (t + 1 < t_end) tile_load(t + 1);
#pragma unroll
        for (int s = 0; s < 8; ++s) {                                  // this wave's 32 pixels: 8 steps of 4 pixels
            const int px = 32 * ph + 4 * s + kq;
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = dys[px * W2_PS + 64 * wi + 16 * i + r16];
                bv[i] = xs[px * W2_PS + 64 * wj + 16 * i + r16];
            }
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bsum[i] += av[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bsum[i] += __shfl_xor(bsum[i], 16, 64);
            bsum[i] += __shfl_xor(bsum[i], 32, 64);
            if (kq == 0) atomicAdd(a.dbias + co0 + 64 * wi + 16 * i + r16, bsum[i]);
        }
    }
    // fold the two pixel halves: waves 4..7 hand their sub-block to waves 0..3 through LDS (16 KB each)
    f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                   // [4 sub-blocks][16 fragments][64 lanes]
    __syncthreads();
    if (ph == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[((wave & 3) * 16 + i * 4 + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (ph == 1) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 o = acc[i][j] + red[(wave * 16 + i * 4 + j) * 64 + lane];
            const int ci = ci0 + 64 * wj + 16 * j + r16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + 64 * wi + 16 * i + 4 * kq + e;
                if (a.dw_direct) { if (ci < a.Cin) a.dw_direct[((size_t)co * a.Cin + ci) * g.taps + tap] = o[e]; }
                else atomicAdd(dwp + (size_t)co * Cs + ci, o[e]);
            }
        }
}

__global__ __launch_bounds__(WG_THREADS) void conv_tap_wgrad_lds_kernel(const TapDesc a) {
    if (a.big) tap_wgrad_body128(a, blockIdx.x, blockIdx.y);
    else tap_wgrad_body(a, blockIdx.x, blockIdx.y);
}

// Batched form.  At training batch sizes a step has ~27 of these GEMMs (16 attention linears, the stride-2 and 1x1 shortcut convolutions,
// the 3x3 layers on 4x4 maps, the two stems), 0.1-4 GFLOP each: launched one by one each fills the chip with ~256 workgroups that own a
// tile or two and then pay the LDS fold and 4096 atomics, 13.8 us per launch on average.  Queued during the backward sweep and
// launched together, the layers fill the chip between them, so every layer splits its pixels over far fewer workgroups.
constexpr int TAP_MAX = 28;
struct TapTable {
    TapDesc d[TAP_MAX];
    int start[TAP_MAX + 1];
    int n;
};
__global__ __launch_bounds__(WG_THREADS) void conv_tap_wgrad_batched_kernel(const TapTable t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.start[k + 1]) ++k;
    k = __builtin_amdgcn_readfirstlane(k);
    const int rel = blockIdx.x - t.start[k];
    const int gx = t.d[k].gx;
    if (t.d[k].big) tap_wgrad_body128(t.d[k], rel % gx, rel / gx);
    else tap_wgrad_body(t.d[k], rel % gx, rel / gx);
}

// stem slab [kh][Cout][kw][8] -> OIHW [Cout][Cin][KH][8]
__global__ void unpack_wgrad_stem_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin, int KH) {
    const size_t total = (size_t)Cout * Cin * KH * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kw = (int)(i & 7);
        const int kh = (int)((i >> 3) % KH);
        const int ci = (int)((i / ((size_t)8 * KH)) % Cin);
        const int co = (int)(i / ((size_t)8 * KH * Cin));
        dw[i] = dwp[(((size_t)kh * Cout + co) * 8 + kw) * 8 + ci];
    }
}

// dwp [tap][Cout][Cs] -> OIHW [Cout][Cin][KH][KW]
__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin, int Cs, int taps) {
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const int ci = (int)((i / taps) % Cin);
        const int co = (int)(i / ((size_t)taps * Cin));
        dw[i] = dwp[((size_t)tap * Cout + co) * Cs + ci];
    }
}

// Deferred form: every slab -> OIHW pass of a backward sweep in ONE launch.  The descriptors travel by value in the kernel
// arguments (no device table to fill under stream capture); a block finds its slab by the block-start table.
constexpr int UNPACK_MAX = 48;
struct UnpackDesc { const float* src; float* dst; int Cout, Cin, Cs, taps; };      // taps < 0: stem slab with KH = -taps
struct UnpackTable {
    UnpackDesc d[UNPACK_MAX];
    int start[UNPACK_MAX + 1];
    int n;
};

__global__ __launch_bounds__(256) void unpack_wgrad_batched_kernel(const UnpackTable t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.start[k + 1]) ++k;
    const UnpackDesc d = t.d[k];
    const int nblk = t.start[k + 1] - t.start[k], blk = blockIdx.x - t.start[k];
    if (d.taps < 0) {
        const int KH = -d.taps;
        const size_t total = (size_t)d.Cout * d.Cin * KH * 8;
        for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < total; i += (size_t)nblk * blockDim.x) {
            const int kw = (int)(i & 7);
            const int kh = (int)((i >> 3) % KH);
            const int ci = (int)((i / ((size_t)8 * KH)) % d.Cin);
            const int co = (int)(i / ((size_t)8 * KH * d.Cin));
            d.dst[i] = d.src[(((size_t)kh * d.Cout + co) * 8 + kw) * 8 + ci];
        }
        return;
    }
    const size_t total = (size_t)d.Cout * d.Cin * d.taps;
    for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < total; i += (size_t)nblk * blockDim.x) {
        const int tap = (int)(i % d.taps);
        const int ci = (int)((i / d.taps) % d.Cin);
        const int co = (int)(i / ((size_t)d.taps * d.Cin));
        d.dst[i] = d.src[((size_t)tap * d.Cout + co) * d.Cs + ci];
    }
}

// =====================================================================================================================
// Small reductions.  colsum: out[c] (+)= sum_m x[m][c] (* y[m][c]);  samplesum: out[b][c] = sum_px x[b][px][c]
// =====================================================================================================================
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     float* __restrict__ out, int M, int C, int rows_per_block) {
    // thread -> channel (coalesced along C), loop over rows of this block's slab, one atomic per (block, channel)
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int m = r0; m < r1; ++m) s += y ? x[(size_t)m * C + c] * y[(size_t)m * C + c] : x[(size_t)m * C + c];
        atomicAdd(out + c, s);
    }
}

// =====================================================================================================================
// Normalisation backward.  Shared structure: pass 1 reduces, per (sample, channel), s1 = sum g and s2 = sum g*xhat over
// the pixels (g = upstream gradient after the activation / ReLU mask); pass 2 forms dx from group (or batch) means.
// =====================================================================================================================
struct NormBwdArgs {
    const float* x;        // [B][HW][C] input of the normalisation (conv output)
    const float* dy;       // [B][HW][C] upstream gradient
    const float* y;        // output of the fused op (for the ReLU mask of BatchNorm) or null
    const float* gamma;    // [C] or null
    const float* beta;     // [C] or null
    const float* skip;     // GroupNorm: residual added before the activation, or null
    const float* tbias;    // [B][C] time bias added before the activation (GroupNorm) / after ReLU (BatchNorm), or null
    const float* mr;       // statistics: GroupNorm [B][G][2] (mean, rstd); BatchNorm [C][2]
    float* dx;             // [B][HW][C]
    float* dres;           // gradient w.r.t. skip / residual, or null
    float* s12;            // workspace [B][C][2] (zeroed by the launcher)
    int B, HW, C, G, act, relu, has_res;
    float* dgamma;         // [C] or null  (parameter gradients: written by block (0,0) of the apply pass from s12)
    float* dbeta;          // [C] or null
    float* dtbias;         // [B][C] or null (GroupNorm: the time bias sits inside the activation)
    const float* sync_sums;// BatchNorm only, or null: [C][2] sums of (g, g*xhat) over ALL ranks (SyncBatchNorm); the means then
    float n_total;         // use these and n_total instead of the local s12 and B*HW (dgamma/dbeta stay local sums)
};

__device__ __forceinline__ float act_grad(float u, int act) {
    switch (act) {
        case SBGM_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case SBGM_ACT_SILU: { const float s = 1.f / (1.f + expf(-u)); return s * (1.f + u * (1.f - s)); }
        case SBGM_ACT_GELU: return 0.5f * (1.f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * expf(-0.5f * u * u);
        default: return 1.f;
    }
}

// GroupNorm: u = xhat*gamma + beta + skip + tbias ; y = act(u).  g = dy * act'(u).
template <bool BATCHNORM>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(NormBwdArgs a, int px_per_block) {
    const int b = blockIdx.y;
    const int cq = a.C >> 2, cpg = a.C / a.G;
    const int q = threadIdx.x % cq, stripe = threadIdx.x / cq, lanes_px = 256 / cq;
    __shared__ float red[256 * 8];                            // stripes fold here: one atomic per (block, channel, sum)
    const bool active = stripe < lanes_px;
    const int c = q * 4;
    const int p0 = blockIdx.x * px_per_block, p1 = min(a.HW, p0 + px_per_block);
    f32x4 mean, rstd, gam = {1.f, 1.f, 1.f, 1.f}, bet = {0.f, 0.f, 0.f, 0.f}, tb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int si = BATCHNORM ? (c + e) : (b * a.G + (c + e) / cpg);
        mean[e] = a.mr[2 * si];
        rstd[e] = a.mr[2 * si + 1];
    }
    if (a.gamma) { gam = *reinterpret_cast<const f32x4*>(a.gamma + c); bet = *reinterpret_cast<const f32x4*>(a.beta + c); }
    if (a.tbias) tb = *reinterpret_cast<const f32x4*>(a.tbias + (size_t)b * a.C + c);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const size_t base = (size_t)b * a.HW * a.C + c;
    for (int p = active ? p0 + stripe : p1; p < p1; p += lanes_px) {
        const size_t o = base + (size_t)p * a.C;
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(a.x + o) - mean) * rstd;
        f32x4 g = *reinterpret_cast<const f32x4*>(a.dy + o);
        if (BATCHNORM) {
            if (a.relu) {
                f32x4 yv = *reinterpret_cast<const f32x4*>(a.y + o);
                if (a.tbias) yv -= tb;                      // y = relu(.) + tbias  ->  relu output = y - tbias
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
            }
        } else if (a.act != SBGM_ACT_NONE) {
            f32x4 u = xh * gam + bet + tb;
            if (a.skip) u += *reinterpret_cast<const f32x4*>(a.skip + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] *= act_grad(u[e], a.act);
        }
        s1 += g;
        s2 += g * xh;
    }
    // red[stripe][c][2]
    if (active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(stripe * a.C + c + e) * 2] = s1[e];
            red[(stripe * a.C + c + e) * 2 + 1] = s2[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
        float t = 0.f;
        for (int sidx = 0; sidx < lanes_px; ++sidx) t += red[sidx * 2 * a.C + i];
        atomicAdd(a.s12 + (size_t)b * a.C * 2 + i, t);
    }
}

// pass 2: dx = rstd * (g*gamma - mean_set(g*gamma) - xhat * mean_set(g*gamma*xhat)); dres = g
template <bool BATCHNORM>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(NormBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* m12 = reinterpret_cast<float*>(smem_raw);       // per channel: mean over the set of g*gamma and g*gamma*xhat
    const int b = blockIdx.y;
    const int cq = a.C >> 2, cpg = a.C / a.G;
    if (BATCHNORM) {
        const float inv_n = a.sync_sums ? 1.f / a.n_total : 1.f / ((float)a.B * a.HW);
        for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
            float t1 = 0.f, t2 = 0.f;
            if (a.sync_sums) { t1 = a.sync_sums[2 * c]; t2 = a.sync_sums[2 * c + 1]; }
            else for (int bb = 0; bb < a.B; ++bb) { t1 += a.s12[((size_t)bb * a.C + c) * 2]; t2 += a.s12[((size_t)bb * a.C + c) * 2 + 1]; }
            m12[2 * c] = t1 * inv_n;                         // gamma factors out per channel for BatchNorm
            m12[2 * c + 1] = t2 * inv_n;
        }
    } else {
        const float inv_n = 1.f / ((float)a.HW * cpg);
        for (int g = threadIdx.x; g < a.G; g += blockDim.x) {
            float t1 = 0.f, t2 = 0.f;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                const float gm = a.gamma ? a.gamma[c] : 1.f;
                t1 += gm * a.s12[((size_t)b * a.C + c) * 2];
                t2 += gm * a.s12[((size_t)b * a.C + c) * 2 + 1];
            }
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { m12[2 * c] = t1 * inv_n; m12[2 * c + 1] = t2 * inv_n; }
        }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        // from s12 [B][C][2]: dgamma[c] = sum_b s2, dbeta[c] = sum_b s1, dtbias[b][c] = s1
        for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
            float t1 = 0.f, t2 = 0.f;
            for (int bb = 0; bb < a.B; ++bb) {
                const float v1 = a.s12[((size_t)bb * a.C + c) * 2];
                t1 += v1;
                t2 += a.s12[((size_t)bb * a.C + c) * 2 + 1];
                if (a.dtbias) a.dtbias[(size_t)bb * a.C + c] = v1;
            }
            if (a.dgamma) a.dgamma[c] = t2;
            if (a.dbeta) a.dbeta[c] = t1;
        }
    }
    __syncthreads();
    const size_t per_sample = (size_t)a.HW * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        const size_t o = (size_t)b * a.HW * a.C + i * 4;
        f32x4 g = *reinterpret_cast<const f32x4*>(a.dy + o);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + o);
        f32x4 dxv;
        f32x4 tb = {0.f, 0.f, 0.f, 0.f};
        if (a.tbias) tb = *reinterpret_cast<const f32x4*>(a.tbias + (size_t)b * a.C + c);
        f32x4 yv = {1.f, 1.f, 1.f, 1.f}, sk = {0.f, 0.f, 0.f, 0.f};
        if (BATCHNORM && a.relu) { yv = *reinterpret_cast<const f32x4*>(a.y + o); if (a.tbias) yv -= tb; }
        if (!BATCHNORM && a.skip) sk = *reinterpret_cast<const f32x4*>(a.skip + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int si = BATCHNORM ? (c + e) : (b * a.G + (c + e) / cpg);
            const float mean = a.mr[2 * si], rstd = a.mr[2 * si + 1];
            const float gm = a.gamma ? a.gamma[c + e] : 1.f;
            const float xh = (xv[e] - mean) * rstd;
            if (BATCHNORM) {
                if (a.relu && !(yv[e] > 0.f)) g[e] = 0.f;
                dxv[e] = gm * rstd * (g[e] - m12[2 * (c + e)] - xh * m12[2 * (c + e) + 1]);
            } else {
                if (a.act != SBGM_ACT_NONE) {
                    const float u = xh * gm + (a.beta ? a.beta[c + e] : 0.f) + sk[e] + tb[e];
                    g[e] *= act_grad(u, a.act);
                }
                dxv[e] = rstd * (g[e] * gm - m12[2 * (c + e)] - xh * m12[2 * (c + e) + 1]);
            }
        }
        *reinterpret_cast<f32x4*>(a.dx + o) = dxv;
        if (a.dres) *reinterpret_cast<f32x4*>(a.dres + o) = g;
    }
}

// out[b][c] = sum_px x[b][px][c]   (time-bias gradients that sit OUTSIDE an activation: conv1 and BatchNorm-late adds)
__global__ __launch_bounds__(256) void samplesum_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, int C,
                                                        int px_per_block) {
    // thread = (channel quad, pixel stripe): 16-byte loads along C, 256 / (C/4) stripes walk the block's pixels side by side, the stripes
    // fold through LDS and the block sends one atomic per channel (a thread per channel with 4-byte loads left 3 of 4 lanes idle at C = 64
    // and took 13 us per launch)
    __shared__ float red[256 * 4];
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * px_per_block, p1 = min(HW, p0 + px_per_block);
    if ((C & 3) == 0 && C <= 1024) {
        const int cq = C >> 2, lanes_px = 256 / cq > 0 ? 256 / cq : 1;
        for (int q0 = 0; q0 < cq; q0 += 256) {                   // one pass unless C > 1024 / 4 quads per block row
            const int q = q0 + (int)threadIdx.x % (cq < 256 ? cq : 256), stripe = (int)threadIdx.x / (cq < 256 ? cq : 256);
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (q < cq && stripe < lanes_px)
                for (int p = p0 + stripe; p < p1; p += lanes_px) s += *reinterpret_cast<const f32x4*>(x + ((size_t)b * HW + p) * C + 4 * q);
            *reinterpret_cast<f32x4*>(red + 4 * threadIdx.x) = s;
            __syncthreads();
            if (stripe == 0 && q < cq) {
                const int w = cq < 256 ? cq : 256;
                for (int k = 1; k < lanes_px; ++k) s += *reinterpret_cast<const f32x4*>(red + 4 * (threadIdx.x + k * w));
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(out + (size_t)b * C + 4 * q + e, s[e]);
            }
            __syncthreads();
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int p = p0; p < p1; ++p) s += x[((size_t)b * HW + p) * C + c];
        atomicAdd(out + (size_t)b * C + c, s);
    }
}

// LayerNorm backward, one wave per token: dx = rstd*(g*gamma - mean(g*gamma) - xhat*mean(g*gamma*xhat)).
// A wave walks over rows_per_wave tokens and keeps its lanes' dgamma/dbeta partial sums in registers (channel = lane + 64k);
// the 4 waves of a block fold through LDS and send ONE atomic per (block, channel) — per-token atomics on the same C
// addresses made this kernel 30 us per launch.
constexpr int LN_MAX_K = 16;     // C <= 1024
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C,
                                                            float eps, int rows_per_wave, const float* __restrict__ dx_add) {
    extern __shared__ float ln_red[];                   // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * rows_per_wave, row1 = min(M, row0 + rows_per_wave);
    float pg[LN_MAX_K], pb[LN_MAX_K];
#pragma unroll
    for (int k = 0; k < LN_MAX_K; ++k) pg[k] = pb[k] = 0.f;
    for (int row = row0; row < row1; ++row) {
        const float* xr = x + (size_t)row * C;
        const float* gr = dy + (size_t)row * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xr[c];
        const float mean = wave_sum(s) / (float)C;
        float s2 = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; s2 += d * d; }
        const float rstd = 1.f / sqrtf(wave_sum(s2) / (float)C + eps);
        float a1 = 0.f, a2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float xh = (xr[c] - mean) * rstd, gg = gr[c] * gamma[c];
            a1 += gg;
            a2 += gg * xh;
        }
        a1 = wave_sum(a1) / (float)C;
        a2 = wave_sum(a2) / (float)C;
#pragma unroll
        for (int k = 0; k < LN_MAX_K; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                const float xh = (xr[c] - mean) * rstd, g = gr[c];
                // dx_add: the gradient that reaches x along its other consumer (the residual connection around the attention
                // half-block), summed here instead of in a pass of its own
                dx[(size_t)row * C + c] = rstd * (g * gamma[c] - a1 - xh * a2) + (dx_add ? dx_add[(size_t)row * C + c] : 0.f);
                pg[k] += g * xh;
                pb[k] += g;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < LN_MAX_K; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
            ln_red[(wave * 2) * C + c] = pg[k];
            ln_red[(wave * 2 + 1) * C + c] = pb[k];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float g = 0.f, bsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            g += ln_red[(w * 2) * C + c];
            bsum += ln_red[(w * 2 + 1) * C + c];
        }
        atomicAdd(dgamma + c, g);
        atomicAdd(dbeta + c, bsum);
    }
}

// Register-resident form for C % 4 == 0: a lane owns the channel quads 4*(lane + 64k), k < KQ, reads x and dy ONCE with 16-byte loads
// and takes the three reductions (mean, variance about that mean, the two gradient means) from registers.  The scalar form above
// walks every row four times with 4-byte loads: 11.6 us per launch on the 0.5 - 1 MB token maps of a C3 step, most of it load latency.
template <int KQ>
__global__ __launch_bounds__(256) void layernorm_bwd_quad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ gamma, float* __restrict__ dx,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C,
                                                                 float eps, int rows_per_wave, const float* __restrict__ dx_add) {
    extern __shared__ float ln_red[];                   // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * rows_per_wave, row1 = min(M, row0 + rows_per_wave);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 pg[KQ], pb[KQ], gm[KQ];
    bool on[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int c = 4 * (lane + 64 * k);
        on[k] = c < C;
        pg[k] = pb[k] = zero;
        gm[k] = on[k] ? *reinterpret_cast<const f32x4*>(gamma + c) : zero;
    }
    const float inv_c = 1.f / (float)C;
    for (int row = row0; row < row1; ++row) {
        const size_t base = (size_t)row * C;
        f32x4 xv[KQ], gv[KQ];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int c = 4 * (lane + 64 * k);
            xv[k] = on[k] ? *reinterpret_cast<const f32x4*>(x + base + c) : zero;
            gv[k] = on[k] ? *reinterpret_cast<const f32x4*>(dy + base + c) : zero;
            s += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
        }
        const float mean = wave_sum(s) * inv_c;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            if (on[k]) {
                xv[k] -= mean;
                s2 += (xv[k][0] * xv[k][0] + xv[k][1] * xv[k][1]) + (xv[k][2] * xv[k][2] + xv[k][3] * xv[k][3]);
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(s2) * inv_c + eps);
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            xv[k] *= rstd;                               // xhat
            const f32x4 gg = gv[k] * gm[k];
            a1 += (gg[0] + gg[1]) + (gg[2] + gg[3]);
            a2 += (gg[0] * xv[k][0] + gg[1] * xv[k][1]) + (gg[2] * xv[k][2] + gg[3] * xv[k][3]);
        }
        a1 = wave_sum(a1) * inv_c;
        a2 = wave_sum(a2) * inv_c;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            if (on[k]) {
                const int c = 4 * (lane + 64 * k);
                f32x4 o = (gv[k] * gm[k] - a1 - xv[k] * a2) * rstd;
                if (dx_add) o += *reinterpret_cast<const f32x4*>(dx_add + base + c);
                *reinterpret_cast<f32x4*>(dx + base + c) = o;
                pg[k] += gv[k] * xv[k];
                pb[k] += gv[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int c = 4 * (lane + 64 * k);
        if (on[k]) {
            *reinterpret_cast<f32x4*>(ln_red + (wave * 2) * C + c) = pg[k];
            *reinterpret_cast<f32x4*>(ln_red + (wave * 2 + 1) * C + c) = pb[k];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float g = 0.f, bsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            g += ln_red[(w * 2) * C + c];
            bsum += ln_red[(w * 2 + 1) * C + c];
        }
        atomicAdd(dgamma + c, g);
        atomicAdd(dbeta + c, bsum);
    }
}

// =====================================================================================================================
// Attention core backward, LDS-staged variant (used whenever K, V of one (sample, head) fit): the arithmetic of the kernel
// below is unchanged, but K, V, the 16 query rows and their dO rows are first copied into LDS with coalesced 16-byte loads
// (rows padded to d+1 floats: conflict-free column walks) instead of being re-read from global memory with one cache line per
// lane — the scalar version spent almost all of its 145 us per launch on those uncoalesced reads.
// =====================================================================================================================
// LPQ = lanes per query row (16: 256 threads; 32: 512 threads, two waves per SIMD — the 256-token block is one workgroup per CU by its
// LDS footprint, and with one wave per SIMD its dependent LDS-read / FMA chains ran at 89 us per launch)
template <int LPQ>
__global__ __launch_bounds__(16 * LPQ) void mha_core_bwd_lds_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                    float* __restrict__ dqkv, int B, int S, int C, int heads, float scale) {
    constexpr int NT = 16 * LPQ;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int d = C / heads, dp = d + 1;
    float* Ks = reinterpret_cast<float*>(smem_raw);      // [S][d+1]
    float* Vs = Ks + (size_t)S * dp;                     // [S][d+1]
    float* Qs = Vs + (size_t)S * dp;                     // [16][d+1]
    float* Os = Qs + 16 * dp;                            // [16][d+1]   dO rows
    float* Pm = Os + 16 * dp;                            // [16][S]
    float* dSm = Pm + 16 * S;                            // [16][S]
    const int qblocks = (S + 15) / 16;
    int w = blockIdx.x;
    const int qb = w % qblocks; w /= qblocks;
    const int h = w % heads;
    const int b = w / heads;
    const size_t rs = 3 * (size_t)C;
    const float* base = qkv + (size_t)b * S * rs + (size_t)h * d;
    float* dbase = dqkv + (size_t)b * S * rs + (size_t)h * d;
    const int dq4 = d >> 2;
    for (int i = threadIdx.x; i < S * dq4; i += NT) {   // K and V rows of this (sample, head)
        const int j = i / dq4, e = (i - j * dq4) * 4;
        const f32x4 kv = *reinterpret_cast<const f32x4*>(base + (size_t)j * rs + C + e);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(base + (size_t)j * rs + 2 * C + e);
#pragma unroll
        for (int t = 0; t < 4; ++t) { Ks[j * dp + e + t] = kv[t]; Vs[j * dp + e + t] = vv[t]; }
    }
    for (int i = threadIdx.x; i < 16 * dq4; i += NT) {  // the 16 query rows and their output gradients
        const int r = i / dq4, e = (i - r * dq4) * 4, qi = qb * 16 + r;
        f32x4 qv = {0.f, 0.f, 0.f, 0.f}, ov = qv;
        if (qi < S) {
            qv = *reinterpret_cast<const f32x4*>(base + (size_t)qi * rs + e);
            ov = *reinterpret_cast<const f32x4*>(dout + ((size_t)b * S + qi) * C + (size_t)h * d + e);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { Qs[r * dp + e + t] = qv[t]; Os[r * dp + e + t] = ov[t]; }
    }
    __syncthreads();
    const int qi_l = threadIdx.x / LPQ, sub = threadIdx.x % LPQ;
    const int qi = qb * 16 + qi_l;
    const bool q_ok = qi < S;
    const float* qrow = Qs + qi_l * dp;
    const float* dorow = Os + qi_l * dp;
    float mx = -INFINITY;
    for (int j = sub; j < S; j += LPQ) {
        const float* kr = Ks + j * dp;
        float sc = 0.f;
        for (int e = 0; e < d; ++e) sc = fmaf(qrow[e], kr[e], sc);
        sc *= scale;
        Pm[qi_l * S + j] = sc;
        mx = fmaxf(mx, sc);
    }
#pragma unroll
    for (int o = LPQ / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = sub; j < S; j += LPQ) { const float pv = expf(Pm[qi_l * S + j] - mx); Pm[qi_l * S + j] = pv; sum += pv; }
#pragma unroll
    for (int o = LPQ / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.f / sum;
    float delta = 0.f;
    for (int j = sub; j < S; j += LPQ) {
        const float pv = Pm[qi_l * S + j] * inv;
        const float* vr = Vs + j * dp;
        float dpv = 0.f;
        for (int e = 0; e < d; ++e) dpv = fmaf(dorow[e], vr[e], dpv);
        Pm[qi_l * S + j] = pv;
        dSm[qi_l * S + j] = dpv;
        delta += pv * dpv;
    }
#pragma unroll
    for (int o = LPQ / 2; o > 0; o >>= 1) delta += __shfl_xor(delta, o, 64);
    for (int j = sub; j < S; j += LPQ) {
        const float ds = q_ok ? Pm[qi_l * S + j] * (dSm[qi_l * S + j] - delta) * scale : 0.f;
        dSm[qi_l * S + j] = ds;
        if (!q_ok) Pm[qi_l * S + j] = 0.f;
    }
    __syncthreads();
    if (q_ok) {                                          // dQ[qi][e] = sum_j dS[qi][j] K[j][e]
        for (int e = sub; e < d; e += LPQ) {
            float acc = 0.f;
            for (int j = 0; j < S; ++j) acc = fmaf(dSm[qi_l * S + j], Ks[j * dp + e], acc);
            dbase[(size_t)qi * rs + e] = acc;
        }
    }
    // dK[j][e] += sum_i dS[i][j] Q[i][e] ;  dV[j][e] += sum_i P[i][j] dO[i][e]   (rows i >= nq carry zeros)
    for (int idx = threadIdx.x; idx < S * d; idx += blockDim.x) {
        const int j = idx / d, e = idx - j * d;
        float ak = 0.f, av = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ak = fmaf(dSm[i * S + j], Qs[i * dp + e], ak);
            av = fmaf(Pm[i * S + j], Os[i * dp + e], av);
        }
        atomicAdd(dbase + (size_t)j * rs + C + e, ak);
        atomicAdd(dbase + (size_t)j * rs + 2 * C + e, av);
    }
}

// =====================================================================================================================
// Attention core backward on the fp32 matrix pipe (head dims that are multiples of 16; K, V of one (sample, head) in LDS).  Same
// decomposition as the kernels around it — one workgroup per (sample, head, block of 16 queries), dK / dV by atomics — but all five
// contractions are v_mfma_f32_16x16x4_f32 products on LDS-resident operands (rows padded by 4 floats: the (row = lane & 15,
// element = 4 step + (lane >> 4)) reads of the A / B fragments fall on 64 distinct banks):
//   S = Q K^T, dP = dO V^T      key blocks dealt round-robin to the 4 waves, d/4 MFMA steps each, both products share the loop
//   P = softmax(S * scale), dS = P * (dP - sum_j P dP) * scale        16 lanes per query row, in LDS
//   dQ = dS K                    one 16 x 16 output block per wave (d/16 blocks), S/4 steps
//   dK += dS^T Q, dV += P^T dO   per key block and 16-wide slice of d: 4 steps (the 16 queries), accumulators straight to the atomics
// The scalar kernels above / below spent 65 us on the 256-token block of a C3 training step (B = 8) where this form needs ~15.
// =====================================================================================================================
template <int D>
__global__ __launch_bounds__(256) void mha_core_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                float* __restrict__ dqkv, int B, int S, int C, int heads, float scale, int G) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int DP = D + 4, DQ4 = D / 4;
    constexpr int NBW = 128 / D;                         // key blocks per wave at the largest S the LDS holds (4, 2, 1 for d = 32, 64, 128)
    const int S16 = (S + 15) & ~15, SP = S16 + 4, NB = S16 >> 4;
    float* Ks = reinterpret_cast<float*>(smem_raw);      // [S16][DP]
    float* Vs = Ks + (size_t)S16 * DP;                   // [S16][DP]
    float* Qs = Vs + (size_t)S16 * DP;                   // [16][DP]
    float* Os = Qs + 16 * DP;                            // [16][DP]   dO rows
    float* Pm = Os + 16 * DP;                            // [16][SP]   scores, then P
    float* Dm = Pm + 16 * SP;                            // [16][SP]   dP, then dS * scale
    const int qblocks = (S + 15) / 16, qgroups = (qblocks + G - 1) / G;
    int w = blockIdx.x;
    const int qg = w % qgroups; w /= qgroups;
    const int h = w % heads;
    const int b = w / heads;
    const size_t rs = 3 * (size_t)C;
    const float* base = qkv + (size_t)b * S * rs + (size_t)h * D;
    float* dbase = dqkv + (size_t)b * S * rs + (size_t)h * D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    for (int i = tid; i < S16 * DQ4; i += 256) {         // K and V rows of this (sample, head), once for the G query blocks; rows past S are zero
        const int j = i / DQ4, e = (i - j * DQ4) * 4;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
        if (j < S) {
            kv = *reinterpret_cast<const f32x4*>(base + (size_t)j * rs + C + e);
            vv = *reinterpret_cast<const f32x4*>(base + (size_t)j * rs + 2 * C + e);
        }
        *reinterpret_cast<f32x4*>(Ks + j * DP + e) = kv;
        *reinterpret_cast<f32x4*>(Vs + j * DP + e) = vv;
    }
    // dK / dV of this wave's key blocks, accumulated over the workgroup's query blocks: [key block slot][16-wide slice of d]
    f32x4 ak[NBW][D / 16], av[NBW][D / 16];
#pragma unroll
    for (int u = 0; u < NBW; ++u)
#pragma unroll
        for (int eb = 0; eb < D / 16; ++eb) { ak[u][eb] = f32x4{0.f, 0.f, 0.f, 0.f}; av[u][eb] = ak[u][eb]; }

    for (int gq = 0; gq < G; ++gq) {
        const int qb = qg * G + gq;
        if (qb >= qblocks) break;                         // uniform
        for (int i = tid; i < 16 * DQ4; i += 256) {      // the 16 query rows and their output gradients
            const int rr = i / DQ4, e = (i - rr * DQ4) * 4, qi = qb * 16 + rr;
            f32x4 qv = {0.f, 0.f, 0.f, 0.f}, ov = qv;
            if (qi < S) {
                qv = *reinterpret_cast<const f32x4*>(base + (size_t)qi * rs + e);
                ov = *reinterpret_cast<const f32x4*>(dout + ((size_t)b * S + qi) * C + (size_t)h * D + e);
            }
            *reinterpret_cast<f32x4*>(Qs + rr * DP + e) = qv;
            *reinterpret_cast<f32x4*>(Os + rr * DP + e) = ov;
        }
        __syncthreads();
        // ---- S = Q K^T * scale and dP = dO V^T: lane ends up with rows 4 kq + reg (queries), column r (key) of a key block ---
        for (int jb = wave; jb < NB; jb += 4) {
            f32x4 as = {0.f, 0.f, 0.f, 0.f}, ap = as;
            const float* kr = Ks + (jb * 16 + r) * DP + kq;
            const float* vr = Vs + (jb * 16 + r) * DP + kq;
            const float* qr = Qs + r * DP + kq;
            const float* orow = Os + r * DP + kq;
#pragma unroll 4
            for (int st = 0; st < DQ4; ++st) {
                as = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[4 * st], kr[4 * st], as, 0, 0, 0);
                ap = __builtin_amdgcn_mfma_f32_16x16x4f32(orow[4 * st], vr[4 * st], ap, 0, 0, 0);
            }
            const bool key_ok = jb * 16 + r < S;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                Pm[(4 * kq + g) * SP + jb * 16 + r] = key_ok ? as[g] * scale : -INFINITY;
                Dm[(4 * kq + g) * SP + jb * 16 + r] = ap[g];
            }
        }
        __syncthreads();
        // ---- softmax rows and dS -----------------------------------------------------------------------------------------------
        {
            const int qi_l = tid >> 4, sub = tid & 15;
            const bool q_ok = qb * 16 + qi_l < S;
            float* prow = Pm + qi_l * SP;
            float* drow = Dm + qi_l * SP;
            float mx = -INFINITY;
            for (int j = sub; j < S16; j += 16) mx = fmaxf(mx, prow[j]);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float sum = 0.f;
            for (int j = sub; j < S16; j += 16) { const float pv = expf(prow[j] - mx); prow[j] = pv; sum += pv; }     // exp(-inf) = 0 past S
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            const float inv = 1.f / sum;
            float delta = 0.f;
            for (int j = sub; j < S16; j += 16) { const float pv = prow[j] * inv; prow[j] = pv; delta += pv * drow[j]; }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) delta += __shfl_xor(delta, o, 64);
            for (int j = sub; j < S16; j += 16) {
                const float pv = prow[j];
                drow[j] = q_ok ? pv * (drow[j] - delta) * scale : 0.f;
                if (!q_ok) prow[j] = 0.f;
            }
        }
        __syncthreads();
        // ---- dQ = dS K: one 16 x 16 block of (queries x head dim) per wave --------------------------------------------------------
        for (int eb = wave; eb < D / 16; eb += 4) {
            f32x4 aq = {0.f, 0.f, 0.f, 0.f};
            const float* dr = Dm + r * SP + kq;
            const float* kc = Ks + kq * DP + eb * 16 + r;
            for (int st = 0; st < S16 / 4; ++st) aq = __builtin_amdgcn_mfma_f32_16x16x4f32(dr[4 * st], kc[(size_t)4 * st * DP], aq, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qi = qb * 16 + 4 * kq + g;
                if (qi < S) dbase[(size_t)qi * rs + eb * 16 + r] = aq[g];
            }
        }
        // ---- dK += dS^T Q, dV += P^T dO: this wave's key blocks, 16-wide slices of the head dim, into the running accumulators -------
#pragma unroll
        for (int u = 0; u < NBW; ++u) {
            const int jb = wave + 4 * u;
            if (jb < NB) {                                // uniform
                f32x4 ds4, p4;                           // A fragments of the 4 steps (queries 4 st + kq), shared by every slice
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    ds4[st] = Dm[(4 * st + kq) * SP + jb * 16 + r];
                    p4[st] = Pm[(4 * st + kq) * SP + jb * 16 + r];
                }
#pragma unroll
                for (int eb = 0; eb < D / 16; ++eb)
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        ak[u][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds4[st], Qs[(4 * st + kq) * DP + eb * 16 + r], ak[u][eb], 0, 0, 0);
                        av[u][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(p4[st], Os[(4 * st + kq) * DP + eb * 16 + r], av[u][eb], 0, 0, 0);
                    }
            }
        }
        __syncthreads();                                  // Qs / Os / Pm / Dm are rewritten by the next query block
    }
#pragma unroll
    for (int u = 0; u < NBW; ++u) {
        const int jb = wave + 4 * u;
        if (jb >= NB) continue;
#pragma unroll
        for (int eb = 0; eb < D / 16; ++eb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key = jb * 16 + 4 * kq + g;
                if (key < S) {
                    atomicAdd(dbase + (size_t)key * rs + C + eb * 16 + r, ak[u][eb][g]);
                    atomicAdd(dbase + (size_t)key * rs + 2 * C + eb * 16 + r, av[u][eb][g]);
                }
            }
    }
}

// =====================================================================================================================
// Attention core backward.  One workgroup per (sample, head, block of 16 queries); 16 lanes per query.
//   P = softmax(q k^T * scale);  dV += P^T dO;  dP = dO V^T;  dS = P*(dP - sum_j P dP);  dQ = dS K * scale;  dK += dS^T Q * scale
// dK/dV are accumulated with fp32 atomics (every query block contributes to all keys).
// =====================================================================================================================
__global__ __launch_bounds__(256) void mha_core_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           float* __restrict__ dqkv, int B, int S, int C, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Pm = reinterpret_cast<float*>(smem_raw);      // [16][S]
    float* dSm = Pm + 16 * S;                            // [16][S]
    const int d = C / heads;
    const int qblocks = (S + 15) / 16;
    int w = blockIdx.x;
    const int qb = w % qblocks; w /= qblocks;
    const int h = w % heads;
    const int b = w / heads;
    const int qi_l = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int qi = qb * 16 + qi_l;
    const bool q_ok = qi < S;
    const size_t rs = 3 * (size_t)C;
    const float* base = qkv + (size_t)b * S * rs + (size_t)h * d;
    float* dbase = dqkv + (size_t)b * S * rs + (size_t)h * d;
    const float* qrow = base + (size_t)(q_ok ? qi : 0) * rs;
    const float* dorow = dout + ((size_t)b * S + (q_ok ? qi : 0)) * C + (size_t)h * d;
    // scores + softmax
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
    float delta = 0.f;
    for (int j = sub; j < S; j += 16) {
        const float p = Pm[qi_l * S + j] * inv;
        const float* vr = base + (size_t)j * rs + 2 * C;
        float dp = 0.f;
        for (int e = 0; e < d; ++e) dp = fmaf(dorow[e], vr[e], dp);
        Pm[qi_l * S + j] = p;
        dSm[qi_l * S + j] = dp;
        delta += p * dp;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) delta += __shfl_xor(delta, o, 64);
    for (int j = sub; j < S; j += 16) {
        const float ds = q_ok ? Pm[qi_l * S + j] * (dSm[qi_l * S + j] - delta) * scale : 0.f;
        dSm[qi_l * S + j] = ds;
        if (!q_ok) Pm[qi_l * S + j] = 0.f;
    }
    __syncthreads();
    // dQ[qi][e] = sum_j dS[qi][j] K[j][e]      (16 lanes of a query stride over e)
    if (q_ok) {
        for (int e = sub; e < d; e += 16) {
            float acc = 0.f;
            for (int j = 0; j < S; ++j) acc = fmaf(dSm[qi_l * S + j], base[(size_t)j * rs + C + e], acc);
            dbase[(size_t)qi * rs + e] = acc;
        }
    }
    // dK[j][e] += sum_i dS[i][j] Q[i][e] ;  dV[j][e] += sum_i P[i][j] dO[i][e]      (threads stride over (j, e))
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

// =====================================================================================================================
// Bilinear x2 (align_corners=False) backward as a gather: each input pixel collects its <= 4x4 output pixels.
// 1-D weights of input y from outputs 2y-1 (.25), 2y (.75, or 1 at y=0), 2y+1 (.75, or 1 at y=H-1), 2y+2 (.25).
// =====================================================================================================================
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H,
                                                             int W, int C) {
    const int cq = C >> 2, OH = 2 * H, OW = 2 * W;
    const size_t total = (size_t)B * H * W * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        size_t r = i / cq;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const float wy[4] = {y >= 1 ? 0.25f : 0.f, y == 0 ? 1.f : 0.75f, y == H - 1 ? 1.f : 0.75f, y <= H - 2 ? 0.25f : 0.f};
        const float wx[4] = {x >= 1 ? 0.25f : 0.f, x == 0 ? 1.f : 0.75f, x == W - 1 ? 1.f : 0.75f, x <= W - 2 ? 0.25f : 0.f};
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = 2 * y - 1 + a;
            if (wy[a] == 0.f || oy < 0 || oy >= OH) continue;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                const int ox = 2 * x - 1 + c2;
                if (wx[c2] == 0.f || ox < 0 || ox >= OW) continue;
                acc += (wy[a] * wx[c2]) * *reinterpret_cast<const f32x4*>(dy + (((size_t)b * OH + oy) * OW + ox) * C + q * 4);
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
    }
}

// =====================================================================================================================
// final_layer.conv (3x3, C -> 1) + /sigma(t) backward.  g[p] = dout[p] / sigma(t_b).
//   da[p][c] = sum_tap w[tap][c] * g[p - off(tap)] ;  dw[tap][c] = sum_p a[p + off(tap)][c] * g[p] ;  dbias = sum_p g[p]
// =====================================================================================================================
__device__ __forceinline__ float sigma_of(float t, float sigma) {
    const float ls = logf(sigma);
    return fmaxf(sqrtf((expf((2.f * t) * ls) - 1.f) / (2.f * ls)), 1e-5f);
}

__global__ __launch_bounds__(256) void cout1_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                             const float* __restrict__ t, float sigma, float* __restrict__ da,
                                                             int B, int H, int W, int C) {
    // thread = (channel quad q, 4 consecutive pixels of a row): the 9 weight quads of q live in registers for the whole loop (the grid
    // stride is a multiple of C/4, so q is fixed per thread) and the 3 x 6 upstream-gradient values around the 4 pixels are loaded once —
    // 4.5 scalar loads per output quad where the per-pixel form issued 9 + 9 weight loads (vector-memory issue bound: 22 us for 33 MB)
    const int cq = C >> 2, xg = (W + 3) >> 2;
    const size_t total = (size_t)B * H * xg * cq;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    const bool fixed_q = stride % cq == 0;
    f32x4 wq[9];
    if (fixed_q && i0 < total) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wq[k] = *reinterpret_cast<const f32x4*>(w + k * C + (int)(i0 % cq) * 4);
    }
    for (size_t i = i0; i < total; i += stride) {
        const int q = (int)(i % cq);
        size_t r = i / cq;
        const int x0 = (int)(r % xg) * 4; r /= xg;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const float inv = t ? 1.f / sigma_of(t[b], sigma) : 1.f;
        if (!fixed_q) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wq[k] = *reinterpret_cast<const f32x4*>(w + k * C + q * 4);
        }
        float g[3][6];                                   // g[kh][j] = dout at (y - (kh - 1), x0 - 1 + j) * inv, 0 outside the image
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int oy = y - (kh - 1);
            const bool rok = (unsigned)oy < (unsigned)H;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int ox = x0 - 1 + j;
                g[kh][j] = rok && (unsigned)ox < (unsigned)W ? dout[((size_t)b * H + oy) * W + ox] * inv : 0.f;
            }
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            if (x0 + px >= W) break;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc += g[kh][px + 1 - (kw - 1)] * wq[kh * 3 + kw];     // output (oy, ox = x - (kw - 1))
            *reinterpret_cast<f32x4*>(da + ((((size_t)b * H + y) * W + x0 + px) * cq + q) * 4) = acc;
        }
    }
}

// One block per slab of input rows.  Thread = (channel quad, pixel stripe): every activation quad is read ONCE and feeds the 9
// taps from registers (the 9 upstream-gradient values around it come from the cache); the stripes fold through LDS and the
// block sends one atomic per (tap, channel).  The first version walked the slab once per (tap, channel) thread: 154 us.
__global__ __launch_bounds__(256) void cout1_bwd_weight_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                               const float* __restrict__ t, float sigma, float* __restrict__ dw,
                                                               float* __restrict__ dbias, int B, int H, int W, int C,
                                                               int rows_per_block) {
    __shared__ float red[9 * 1024];                                  // [tap][stripe][C], stripes * C <= 1024
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* gs = reinterpret_cast<float*>(smem_raw);                   // upstream gradient rows r0 - 1 .. r1 of the [B*H][W] plane
    const int nrows = B * H;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    // the 9 values of g around a pixel used to be 9 scalar global loads per activation quad (a vector-memory instruction costs the wave
    // 100+ cycles of issue: 49 us for this 33 MB pass); they now come from LDS
    for (int i = threadIdx.x; i < (r1 - r0 + 2) * W; i += blockDim.x) {
        const int grow = r0 - 1 + i / W;
        gs[i] = (unsigned)grow < (unsigned)nrows ? dout[(size_t)grow * W + (i % W)] : 0.f;
    }
    __syncthreads();
    const int cq = C >> 2, q = threadIdx.x % cq, stripe = threadIdx.x / cq, lanes_px = 256 / cq;
    f32x4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (stripe < lanes_px) {
        for (int row = r0; row < r1; ++row) {
            const int b = row / H, iy = row - b * H;
            const float inv = t ? 1.f / sigma_of(t[b], sigma) : 1.f;
            const float* drow = gs + (size_t)(b * H - (r0 - 1)) * W;   // row oy of image b sits at global row b*H + oy
            for (int ix = stripe; ix < W; ix += lanes_px) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(a + (((size_t)b * H + iy) * W + ix) * C + q * 4) * inv;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int oy = iy - (kh - 1);                    // output pixel whose tap (kh, kw) reads input (iy, ix)
                    if ((unsigned)oy >= (unsigned)H) continue;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int ox = ix - (kw - 1);
                        if ((unsigned)ox >= (unsigned)W) continue;
                        acc[kh * 3 + kw] += av * drow[(size_t)oy * W + ox];
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(k * lanes_px + stripe) * C + q * 4 + e] = acc[k][e];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 9 * C; idx += blockDim.x) {
        const int tap = idx / C, c = idx - tap * C;
        float s = 0.f;
        for (int sidx = 0; sidx < lanes_px; ++sidx) s += red[(tap * lanes_px + sidx) * C + c];
        atomicAdd(dw + idx, s);
    }
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int row = r0; row < r1; ++row) {
            const float inv = t ? 1.f / sigma_of(t[row / H], sigma) : 1.f;
            for (int ox = threadIdx.x; ox < W; ox += 64) s += dout[(size_t)row * W + ox] * inv;
        }
        s = wave_sum(s);
        if (threadIdx.x == 0) atomicAdd(dbias, s);
    }
}

// =====================================================================================================================
// Time projection backward: out[b][c] = bias[c] + sum_d W[c][d]*semb[b][d], semb = silu(emb).
//   dW[c][d] = sum_b dout[b][c]*semb[b][d]; dbias[c] = sum_b dout[b][c]; demb[b][d] += silu'(emb[b][d]) * sum_c dout[b][c] W[c][d]
// =====================================================================================================================
__global__ __launch_bounds__(256) void time_proj_bwd_w_kernel(const float* __restrict__ dout, const float* __restrict__ semb,
                                                              float* __restrict__ dW, float* __restrict__ dbias, int B, int D, int ch) {
    const size_t total = (size_t)ch * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int dd = (int)(i % D), c = (int)(i / D);
        float s = 0.f, sb = 0.f;
        for (int b = 0; b < B; ++b) { const float g = dout[(size_t)b * ch + c]; s = fmaf(g, semb[(size_t)b * D + dd], s); sb += g; }
        dW[i] = s;
        if (dd == 0) dbias[c] = sb;
    }
}
// the same for up to 16 projections in ONE launch (a training step has 9: 42 us of 4.7 us launches): block -> projection by prefix
struct TimeProjBwdMulti {
    const float* dout[16];
    const float* semb[16];
    float* dW[16];
    float* dbias[16];
    int ch[16];
    int block_begin[17];
    int n, B, D;
};
__global__ __launch_bounds__(256) void time_proj_bwd_w_multi_kernel(const TimeProjBwdMulti a) {
    int k = 0;
    while (k + 1 < a.n && a.block_begin[k + 1] <= (int)blockIdx.x) ++k;
    const float* __restrict__ dout = a.dout[k];
    const float* __restrict__ semb = a.semb[k];
    const int ch = a.ch[k], D = a.D, B = a.B;
    const size_t total = (size_t)ch * D;
    const int nb = a.block_begin[k + 1] - a.block_begin[k];
    for (size_t i = (size_t)(blockIdx.x - a.block_begin[k]) * blockDim.x + threadIdx.x; i < total; i += (size_t)nb * blockDim.x) {
        const int dd = (int)(i % D), c = (int)(i / D);
        float s = 0.f, sb = 0.f;
        for (int b = 0; b < B; ++b) { const float g = dout[(size_t)b * ch + c]; s = fmaf(g, semb[(size_t)b * D + dd], s); sb += g; }
        a.dW[k][i] = s;
        if (dd == 0) a.dbias[k][c] = sb;
    }
}
// demb_pre[b][d] (+)= silu'(emb) * sum_c dout[b][c] W[c][d]   (emb recovered from `emb_raw` = pre-SiLU embedding)
__global__ __launch_bounds__(256) void time_proj_bwd_e_kernel(const float* __restrict__ dout, const float* __restrict__ Wt,
                                                              const float* __restrict__ emb_raw, float* __restrict__ demb, int B,
                                                              int D, int ch) {
    const int total = B * D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int dd = i % D, b = i / D;
        float s = 0.f;
        for (int c = 0; c < ch; ++c) s = fmaf(dout[(size_t)b * ch + c], Wt[(size_t)c * D + dd], s);
        demb[i] += s * act_grad(emb_raw[i], SBGM_ACT_SILU);
    }
}
// label embedding gradient: dtable[y[b]][d] += demb[b][d]
__global__ void label_emb_bwd_kernel(const float* __restrict__ demb, const int64_t* __restrict__ y, float* __restrict__ dtable, int B, int D) {
    const int total = B * D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
        atomicAdd(dtable + (size_t)y[i / D] * D + (i % D), demb[i]);
}

// dx = dy * act'(x)   (GELU between the attention FF linears)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                      size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * act_grad(x[i], act);
}

// y = act(x) out of place (training keeps the pre-activation)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = sbgm_act(x[i], act);
}

}  // namespace

// ---- launchers -----------------------------------------------------------------------------------------------------
// Deferred slab -> OIHW passes (sbgm_wgrad_defer / sbgm_wgrad_flush in the C ABI): while deferral is on, sbgm_launch_conv_wgrad
// queues its layout pass instead of launching it; the flush at the end of the backward sweep runs all of them as one kernel.
// The caller keeps every queued slab alive and untouched until the flush.  Process-wide, like the prezeroed switch.
// sbgm_wgrad_deferred bit 0: queue the layout passes; bit 1: queue the per-tap split-K GEMMs themselves (conv_tap_wgrad_lds_kernel) and run
// them as ONE batched launch at the flush — the caller then also keeps dy and x of every queued call alive until the flush.
int sbgm_wgrad_deferred = 0;
static std::vector<UnpackDesc> g_unpack_queue;
struct TapQueued { TapDesc d; int n_tiles; float* dw_oihw; int unpack_taps; bool aliased; };     // unpack_taps: as UnpackDesc.taps
static std::vector<TapQueued> g_tap_queue;
struct HaloQueued { HaloDesc d; int n_tiles; float* dw_oihw; };
static std::vector<HaloQueued> g_halo_queue[2];            // [0]: 16 x 16 tiles, [1]: 8 x 8 maps (4 images per tile)

static int unpack_or_queue(const float* src, float* dst, int Cout, int Cin, int Cs, int taps, hipStream_t st) {
    if (sbgm_wgrad_deferred & 1) {
        g_unpack_queue.push_back(UnpackDesc{src, dst, Cout, Cin, Cs, taps});
        return 0;
    }
    const size_t total = (size_t)Cout * Cin * (taps < 0 ? -taps * 8 : taps);
    if (taps < 0) hipLaunchKernelGGL(unpack_wgrad_stem_kernel, dim3(stream_blocks(total)), dim3(256), 0, st, src, dst, Cout, Cin, -taps);
    else hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(stream_blocks(total)), dim3(256), 0, st, src, dst, Cout, Cin, Cs, taps);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_wgrad_pending() { return (int)(g_unpack_queue.size() + g_tap_queue.size() + g_halo_queue[0].size() + g_halo_queue[1].size()); }
// forget the queued work without running it (a backward pass that raised: the tensors it points at may be gone)
void sbgm_wgrad_discard_queue() { g_unpack_queue.clear(); g_tap_queue.clear(); g_halo_queue[0].clear(); g_halo_queue[1].clear(); }

static int tap_set_attr() {
    static bool done = false;
    if (!done) {
        const int lds = 2 * W1_PX * W1_PS * 4;
        SBGM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tap_wgrad_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        SBGM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tap_wgrad_batched_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        done = true;
    }
    return 0;
}

// the queued per-tap GEMMs: split each layer's pixels so that all layers together make ~4 workgroups per CU, launch, queue the layout passes
static int tap_flush(hipStream_t st) {
    if (g_tap_queue.empty()) return 0;
    if (tap_set_attr()) return 1;
    static const int target = getenv("SBGM_TAP_BATCH_WGS") ? atoi(getenv("SBGM_TAP_BATCH_WGS")) : 1024;
    double work = 0.0;
    auto cost = [](const TapQueued& q) { return (double)q.d.gx * q.n_tiles * (q.d.big ? 2.0 : 1.0); };   // MFMAs per tile-step: 128 vs 64 per wave
    for (auto& q : g_tap_queue) work += cost(q);
    const size_t lds = (size_t)2 * W1_PX * W1_PS * 4;
    size_t i = 0;
    while (i < g_tap_queue.size()) {
        TapTable t{};
        int nb = 0;
        for (; i < g_tap_queue.size() && t.n < TAP_MAX; ++i) {
            TapQueued& q = g_tap_queue[i];
            TapDesc d = q.d;
            const double share = cost(q) / work;
            int gy = (int)std::lround(target * share / d.gx);
            gy = std::max(1, std::min(gy, q.n_tiles));
            d.tiles_per_wg = (q.n_tiles + gy - 1) / gy;
            d.gy = (q.n_tiles + d.tiles_per_wg - 1) / d.tiles_per_wg;
            // no pixel split: plain OIHW stores, no slab, no layout pass (not for the stem's slab layout)
            const bool direct = d.gy == 1 && !d.g.stem;
            d.dw_direct = direct ? q.dw_oihw : nullptr;
            if (q.d.g.stem) g_unpack_queue.push_back(UnpackDesc{d.dwp, q.dw_oihw, d.Cout, d.Cin, d.Cs, q.unpack_taps});
            else if (!direct && !q.aliased) g_unpack_queue.push_back(UnpackDesc{d.dwp, q.dw_oihw, d.Cout, d.Cin, d.Cs, q.unpack_taps});
            t.d[t.n] = d;
            t.start[t.n] = nb;
            nb += d.gx * d.gy;
            ++t.n;
        }
        t.start[t.n] = nb;
        hipLaunchKernelGGL(conv_tap_wgrad_batched_kernel, dim3(nb), dim3(WG_THREADS), lds, st, t);
        if (hipGetLastError() != hipSuccess) {
            g_tap_queue.clear();
            g_unpack_queue.clear();
            sbgm_set_error("wgrad_flush: batched weight-gradient launch failed");
            return 1;
        }
    }
    g_tap_queue.clear();
    return 0;
}

static int halo_set_attr() {
    static bool done = false;
    if (!done) {
        const int l16 = (WG_DY_FLOATS + WgTile<16>::X_FLOATS) * 4, l8 = (WG_DY_FLOATS + WgTile<8>::X_FLOATS) * 4;
        SBGM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_lds_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, l16));
        SBGM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_lds_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, l8));
        SBGM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_batched_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, l16));
        SBGM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_batched_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, l8));
        done = true;
    }
    return 0;
}

// the queued 3x3 weight gradients of one tile geometry: pixels split by each layer's share of the total work (one 8-wave workgroup per CU
// is resident: `target` workgroups = target / 256 rounds), one launch, layout passes queued behind it
static int halo_flush(int which, hipStream_t st) {
    std::vector<HaloQueued>& qv = g_halo_queue[which];
    if (qv.empty()) return 0;
    if (halo_set_attr()) return 1;
    static const int target = getenv("SBGM_HALO_BATCH_WGS") ? atoi(getenv("SBGM_HALO_BATCH_WGS")) : 768;
    double work = 0.0;
    for (auto& q : qv) work += (double)q.d.gx * q.n_tiles;
    const size_t lds = (size_t)(WG_DY_FLOATS + (which == 0 ? WgTile<16>::X_FLOATS : WgTile<8>::X_FLOATS)) * 4;
    size_t i = 0;
    while (i < qv.size()) {
        HaloTable t{};
        int nb = 0;
        for (; i < qv.size() && t.n < HALO_MAX; ++i) {
            HaloQueued& q = qv[i];
            HaloDesc d = q.d;
            int gy = (int)std::lround(target * ((double)d.gx * q.n_tiles / work) / d.gx);
            gy = std::max(1, std::min(gy, q.n_tiles));
            d.tiles_per_wg = (q.n_tiles + gy - 1) / gy;
            d.gy = (q.n_tiles + d.tiles_per_wg - 1) / d.tiles_per_wg;
            const bool direct = d.gy == 1;                   // no pixel split: plain OIHW stores, no slab, no layout pass
            d.dw_direct = direct ? q.dw_oihw : nullptr;
            if (!direct) g_unpack_queue.push_back(UnpackDesc{d.dwp, q.dw_oihw, d.Cout, d.Cin, d.Cs, 9});
            t.d[t.n] = d;
            t.start[t.n] = nb;
            nb += d.gx * d.gy;
            ++t.n;
        }
        t.start[t.n] = nb;
        if (which == 0) hipLaunchKernelGGL(conv3x3_wgrad_batched_kernel<16>, dim3(nb), dim3(WG_THREADS), lds, st, t);
        else hipLaunchKernelGGL(conv3x3_wgrad_batched_kernel<8>, dim3(nb), dim3(WG_THREADS), lds, st, t);
        if (hipGetLastError() != hipSuccess) {
            sbgm_wgrad_discard_queue();
            sbgm_set_error("wgrad_flush: batched 3x3 weight-gradient launch failed");
            return 1;
        }
    }
    qv.clear();
    return 0;
}

int sbgm_launch_wgrad_flush(hipStream_t st) {
    if (halo_flush(0, st) || halo_flush(1, st)) return 1;
    if (tap_flush(st)) return 1;
    size_t i = 0;
    const size_t n = g_unpack_queue.size();
    while (i < n) {
        UnpackTable t{};
        int nb = 0;
        t.n = 0;
        for (; i < n && t.n < UNPACK_MAX; ++i) {
            const UnpackDesc& d = g_unpack_queue[i];
            const size_t total = (size_t)d.Cout * d.Cin * (d.taps < 0 ? -d.taps * 8 : d.taps);
            t.d[t.n] = d;
            t.start[t.n] = nb;
            nb += (int)std::max<size_t>(1, std::min<size_t>((total + 1023) / 1024, 256));          // >= 4 elements per thread
            ++t.n;
        }
        t.start[t.n] = nb;
        hipLaunchKernelGGL(unpack_wgrad_batched_kernel, dim3(nb), dim3(256), 0, st, t);
        if (hipGetLastError() != hipSuccess) {
            g_unpack_queue.clear();
            sbgm_set_error("wgrad_flush: launch failed");
            return 1;
        }
    }
    g_unpack_queue.clear();
    return 0;
}

int sbgm_launch_conv_wgrad(const float* dy, const float* x, float* dw_oihw, float* dwp_ws, int B, int H, int W, int Cs, int Cin,
                           int Cout, int KH, int KW, int S, int PAD, hipStream_t st, float* dbias) {
    SBGM_CHECK(Cout % 64 == 0, "wgrad: Cout=%d must be a multiple of 64", Cout);
    SBGM_CHECK((Cs == 4 || Cs == 8 || Cs % 16 == 0) && Cin <= Cs, "wgrad: padded Cin %d unsupported (Cin=%d)", Cs, Cin);
    const int OH = (H + 2 * PAD - KH) / S + 1, OW = (W + 2 * PAD - KW) / S + 1, M = B * OH * OW;
    const int fci = Cs % 64 == 0 ? 4 : (Cs % 32 == 0 ? 2 : 1);
    const int tiles = KH * KW * (Cout / 64) * ((Cs + 16 * fci - 1) / (16 * fci));
    static const int wtarget = getenv("SBGM_WG_TARGET") ? atoi(getenv("SBGM_WG_TARGET")) : 4096;
    int splits = std::max(1, std::min((M + 63) / 64, (wtarget + tiles - 1) / tiles));   // measured: 1024 target waves is 14 % slower per step
    int pps = ((M + splits - 1) / splits + 3) / 4 * 4;
    splits = (M + pps - 1) / pps;
    const size_t n = (size_t)KH * KW * Cout * Cs;
    if (!sbgm_scratch_prezeroed && dbias) { if (sbgm_zero_async(dbias, (size_t)Cout * 4, st)) return 1; }
    const size_t dy_b = (size_t)M * Cout * 4, x_b = (size_t)B * H * W * Cs * 4;
    const bool lds16 = W % 16 == 0 && H % 16 == 0, lds8 = W == 8 && H == 8;
    if (KH == 3 && KW == 3 && S == 1 && PAD == 1 && Cs % 32 == 0 && (lds16 || lds8) && dy_b < (1ull << 31) && x_b < (1ull << 31) &&
        getenv("SBGM_NO_LDS_WGRAD") == nullptr) {
        const int blocks_x = (Cout / 32) * (Cs / 32);
        const int n_tiles = lds16 ? B * (H / 16) * (W / 16) : (B + 3) / 4;
        const int wgs_y = std::max(1, std::min(n_tiles, (256 + blocks_x - 1) / blocks_x));      // one 8-wave workgroup per CU
        const int tpw = (n_tiles + wgs_y - 1) / wgs_y;
        const dim3 grid_lds(blocks_x, (n_tiles + tpw - 1) / tpw);
        if (halo_set_attr()) return 1;
        HaloDesc d{dy, x, dwp_ws, dbias, nullptr, B, H, W, Cs, Cout, tpw, Cin, blocks_x, (int)grid_lds.y, (uint32_t)dy_b, (uint32_t)x_b};
        if (sbgm_wgrad_deferred & 2) {                      // queued for the sweep's batched launch (the split is decided at the flush)
            if (!sbgm_scratch_prezeroed) { if (sbgm_zero_async(dwp_ws, n * 4, st)) return 1; }
            g_halo_queue[lds16 ? 0 : 1].push_back(HaloQueued{d, n_tiles, dw_oihw});
            return 0;
        }
        float* direct = grid_lds.y == 1 ? dw_oihw : nullptr;          // no pixel split: no atomics, no slab, no unpack pass
        if (!direct && !sbgm_scratch_prezeroed) { if (sbgm_zero_async(dwp_ws, n * 4, st)) return 1; }
        d.dw_direct = direct;
        if (lds16) hipLaunchKernelGGL(conv3x3_wgrad_lds_kernel<16>, grid_lds, dim3(WG_THREADS), (size_t)(WG_DY_FLOATS + WgTile<16>::X_FLOATS) * 4, st, d);
        else hipLaunchKernelGGL(conv3x3_wgrad_lds_kernel<8>, grid_lds, dim3(WG_THREADS), (size_t)(WG_DY_FLOATS + WgTile<8>::X_FLOATS) * 4, st, d);
        SBGM_LAUNCH_CHECK();
        if (!direct) return unpack_or_queue(dwp_ws, dw_oihw, Cout, Cin, Cs, KH * KW, st);
        return 0;
    }
    const bool stem = Cs == 8 && KW == 8;
    if ((Cs % 64 == 0 || stem) && dy_b < (1ull << 31) && x_b < (1ull << 31) && getenv("SBGM_NO_LDS_WGRAD") == nullptr) {
        // one split-K GEMM per tap (1x1 / linear layers, and whatever the halo kernel above does not take)
        const int taps = stem ? KH : KH * KW;
        static const bool big_ok = getenv("SBGM_NO_TAP128") == nullptr;
        const int big = big_ok && !stem && Cs % 128 == 0 && Cout % 128 == 0 ? 1 : 0;
        const int blocks_x = big ? taps * (Cout / 128) * (Cs / 128) : taps * (Cout / 64) * (stem ? 1 : Cs / 64);
        const int n_tiles = big ? (M + W2_PX - 1) / W2_PX : (M + W1_PX - 1) / W1_PX;
        const int wgs_y = std::max(1, std::min(n_tiles, (256 + blocks_x - 1) / blocks_x));      // one 8-wave workgroup per CU
        const int tpw = (n_tiles + wgs_y - 1) / wgs_y;
        const dim3 grid_lds(blocks_x, (n_tiles + tpw - 1) / tpw);
        const size_t lds = (size_t)2 * W1_PX * W1_PS * 4;
        if (tap_set_attr()) return 1;
        // ws == dw_oihw: for a 1x1 kernel without channel padding the slab IS the OIHW gradient, so the caller may hand in the
        // (zeroed) gradient tensor as workspace and no unpack pass is needed
        const bool aliased = dwp_ws == dw_oihw;
        SBGM_CHECK(!aliased || (KH * KW == 1 && Cs == Cin), "wgrad: ws may alias dw only for 1x1 kernels with c_pad == Cin");
        const TapGeom tg{M, OH, OW, H, W, S, PAD, KW, KH * KW, stem ? 1 : 0};
        TapDesc d{dy, x, dwp_ws, dbias, nullptr, tg, Cs, Cout, tpw, Cin, blocks_x, (int)grid_lds.y, (uint32_t)dy_b, (uint32_t)x_b, big};
        if (sbgm_wgrad_deferred & 2) {
            // queued: the slab may still be needed (the split is decided at the flush), so it is zeroed now unless it arrives zeroed
            if (!sbgm_scratch_prezeroed) { if (sbgm_zero_async(dwp_ws, n * 4, st)) return 1; }
            g_tap_queue.push_back(TapQueued{d, n_tiles, dw_oihw, stem ? -KH : KH * KW, aliased});
            return 0;
        }
        float* direct = grid_lds.y == 1 && !stem ? dw_oihw : nullptr;
        if (!direct && !sbgm_scratch_prezeroed) { if (sbgm_zero_async(dwp_ws, n * 4, st)) return 1; }
        d.dw_direct = direct;
        hipLaunchKernelGGL(conv_tap_wgrad_lds_kernel, grid_lds, dim3(WG_THREADS), lds, st, d);
        SBGM_LAUNCH_CHECK();
        if (stem) return unpack_or_queue(dwp_ws, dw_oihw, Cout, Cin, Cs, -KH, st);
        if (!direct && !aliased) return unpack_or_queue(dwp_ws, dw_oihw, Cout, Cin, Cs, KH * KW, st);
        return 0;
    }
    if (!sbgm_scratch_prezeroed) { if (sbgm_zero_async(dwp_ws, n * 4, st)) return 1; }
    dim3 grid((tiles + 3) / 4, splits);
#define SBGM_WG(F) hipLaunchKernelGGL(conv_wgrad_kernel<F>, grid, dim3(256), 0, st, dy, x, dwp_ws, B, H, W, Cs, OH, OW, Cout, KH, KW, S, PAD, pps, dbias)
    if (fci == 4) SBGM_WG(4); else if (fci == 2) SBGM_WG(2); else SBGM_WG(1);
#undef SBGM_WG
    SBGM_LAUNCH_CHECK();
    return unpack_or_queue(dwp_ws, dw_oihw, Cout, Cin, Cs, KH * KW, st);
}

int sbgm_launch_colsum(const float* x, const float* y, float* out, int M, int C, hipStream_t st) {
    { if (sbgm_zero_async(out, (size_t)C * 4, st)) return 1; }
    const int slabs = std::max(1, std::min(512, M / 64));
    const int rpb = (M + slabs - 1) / slabs;
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, st, x, y, out, M, C, rpb);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_samplesum(const float* x, float* out, int B, int HW, int C, hipStream_t st) {
    if (!sbgm_scratch_prezeroed) { if (sbgm_zero_async(out, (size_t)B * C * 4, st)) return 1; }
    const int chunks = std::max(1, std::min(64, HW / 64));
    const int ppb = (HW + chunks - 1) / chunks;
    hipLaunchKernelGGL(samplesum_kernel, dim3((HW + ppb - 1) / ppb, B), dim3(256), 0, st, x, out, HW, C, ppb);
    SBGM_LAUNCH_CHECK();
    return 0;
}

static int norm_bwd_reduce(bool bn, const NormBwdArgs& a, hipStream_t st) {
    SBGM_CHECK(a.C % 4 == 0 && a.C <= 1024, "norm_bwd: C=%d unsupported", a.C);
    if (!sbgm_scratch_prezeroed) { if (sbgm_zero_async(a.s12, (size_t)a.B * a.C * 2 * 4, st)) return 1; }
    const int lanes_px = std::max(1, 256 / (a.C / 4));
    int chunks = std::max(1, std::min(256, a.HW / (lanes_px * 4)));      // measured: 4 px per thread; 16 was 7 % slower per step
    const int ppb = (a.HW + chunks - 1) / chunks;
    chunks = (a.HW + ppb - 1) / ppb;
    if (bn) hipLaunchKernelGGL(norm_bwd_reduce_kernel<true>, dim3(chunks, a.B), dim3(256), 0, st, a, ppb);
    else hipLaunchKernelGGL(norm_bwd_reduce_kernel<false>, dim3(chunks, a.B), dim3(256), 0, st, a, ppb);
    SBGM_LAUNCH_CHECK();
    return 0;
}

static int norm_bwd_apply(bool bn, const NormBwdArgs& a, hipStream_t st) {
    SBGM_CHECK(a.C % 4 == 0 && a.C <= 1024, "norm_bwd: C=%d unsupported", a.C);
    const size_t per_sample = (size_t)a.HW * (a.C / 4);
    const int bx = (int)std::max<size_t>(1, std::min<size_t>((per_sample + 255) / 256, 2048 / std::max(1, a.B) + 1));
    if (bn) hipLaunchKernelGGL(norm_bwd_apply_kernel<true>, dim3(bx, a.B), dim3(256), 2 * a.C * 4, st, a);
    else hipLaunchKernelGGL(norm_bwd_apply_kernel<false>, dim3(bx, a.B), dim3(256), 2 * a.C * 4, st, a);
    SBGM_LAUNCH_CHECK();
    return 0;
}

static int norm_bwd(bool bn, NormBwdArgs a, float* dgamma, float* dbeta, float* dtbias, hipStream_t st) {
    a.dgamma = dgamma;
    a.dbeta = dbeta;
    a.dtbias = dtbias;
    if (norm_bwd_reduce(bn, a, st)) return 1;
    return norm_bwd_apply(bn, a, st);
}

// SyncBatchNorm backward in two calls: the caller all-reduces the per-channel sums of s12 between them
int sbgm_launch_batchnorm_bwd_reduce(const float* x, const float* dy, const float* y, const float* tbias_after, const float* mr,
                                     int relu, float* s12_ws, int B, int HW, int C, hipStream_t st) {
    NormBwdArgs a{x, dy, y, nullptr, nullptr, nullptr, tbias_after, mr, nullptr, nullptr, s12_ws, B, HW, C, 1, SBGM_ACT_NONE, relu, 0};
    return norm_bwd_reduce(true, a, st);
}
int sbgm_launch_batchnorm_bwd_apply(const float* x, const float* dy, const float* y, const float* gamma, const float* tbias_after,
                                    const float* mr, int relu, float* dx, float* dres, float* dgamma, float* dbeta, const float* s12_ws,
                                    const float* sync_sums, double n_total, int B, int HW, int C, hipStream_t st) {
    NormBwdArgs a{x, dy, y, gamma, nullptr, nullptr, tbias_after, mr, dx, dres, const_cast<float*>(s12_ws), B, HW, C, 1, SBGM_ACT_NONE,
                  relu, dres != nullptr};
    a.dgamma = dgamma;
    a.dbeta = dbeta;
    a.sync_sums = sync_sums;
    a.n_total = (float)n_total;
    return norm_bwd_apply(true, a, st);
}

int sbgm_launch_groupnorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* skip,
                              const float* tbias, const float* mr, int act, float* dx, float* dskip, float* dgamma, float* dbeta,
                              float* dtbias, float* s12_ws, int B, int HW, int C, int G, hipStream_t st) {
    NormBwdArgs a{x, dy, nullptr, gamma, beta, skip, tbias, mr, dx, dskip, s12_ws, B, HW, C, G, act, 0, skip != nullptr};
    return norm_bwd(false, a, dgamma, dbeta, dtbias, st);
}

int sbgm_launch_batchnorm_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* tbias_after,
                              const float* mr, int relu, float* dx, float* dres, float* dgamma, float* dbeta, float* s12_ws, int B,
                              int HW, int C, hipStream_t st) {
    NormBwdArgs a{x, dy, y, gamma, nullptr, nullptr, tbias_after, mr, dx, dres, s12_ws, B, HW, C, 1, SBGM_ACT_NONE, relu, dres != nullptr};
    return norm_bwd(true, a, dgamma, dbeta, nullptr, st);
}

int sbgm_launch_layernorm_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta, int M, int C,
                              float eps, hipStream_t st, const float* dx_add) {
    SBGM_CHECK(C <= 64 * LN_MAX_K, "layernorm_bwd: C=%d > %d", C, 64 * LN_MAX_K);
    if (sbgm_scratch_prezeroed) {
    } else if (dbeta == dgamma + C) {                     // one [2][C] tensor: one memset
        { if (sbgm_zero_async(dgamma, (size_t)C * 8, st)) return 1; }
    } else {
        { if (sbgm_zero_async(dgamma, (size_t)C * 4, st)) return 1; }
        { if (sbgm_zero_async(dbeta, (size_t)C * 4, st)) return 1; }
    }
    const int rpw = std::max(1, M / 1024);               // ~256 blocks of 4 waves
    const dim3 grid((M + 4 * rpw - 1) / (4 * rpw));
    const size_t lds = (size_t)8 * C * sizeof(float);
    static const bool scalar_only = getenv("SBGM_LN_BWD_SCALAR") != nullptr;
    if (C % 4 == 0 && !scalar_only) {
        if (C <= 256) hipLaunchKernelGGL(layernorm_bwd_quad_kernel<1>, grid, dim3(256), lds, st, x, dy, gamma, dx, dgamma, dbeta, M, C, eps, rpw, dx_add);
        else if (C <= 512) hipLaunchKernelGGL(layernorm_bwd_quad_kernel<2>, grid, dim3(256), lds, st, x, dy, gamma, dx, dgamma, dbeta, M, C, eps, rpw, dx_add);
        else hipLaunchKernelGGL(layernorm_bwd_quad_kernel<4>, grid, dim3(256), lds, st, x, dy, gamma, dx, dgamma, dbeta, M, C, eps, rpw, dx_add);
    } else {
        hipLaunchKernelGGL(layernorm_bwd_kernel, grid, dim3(256), lds, st, x, dy, gamma, dx, dgamma, dbeta, M, C, eps, rpw, dx_add);
    }
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_mha_core_bwd(const float* qkv, const float* dout, float* dqkv, int B, int S, int C, int heads, hipStream_t st) {
    SBGM_CHECK(heads > 0 && C % heads == 0, "mha_bwd: C=%d heads=%d", C, heads);
    SBGM_CHECK((size_t)2 * 16 * S * 4 <= 150 * 1024, "mha_bwd: S=%d too long for the LDS-resident score rows", S);
    if (!sbgm_scratch_prezeroed) { if (sbgm_zero_async(dqkv, (size_t)B * S * 3 * C * 4, st)) return 1; }
    const int blocks = B * heads * ((S + 15) / 16);
    const int d = C / heads;
    {   // matrix-pipe form: head dim 32 / 64 / 128, K and V of a (sample, head) LDS-resident, <= 128 / d key blocks per wave
        const int S16 = (S + 15) & ~15;
        const size_t lds_mfma = ((size_t)(2 * S16 + 32) * (d + 4) + (size_t)32 * (S16 + 4)) * 4;
        static const bool mfma_ok = getenv("SBGM_NO_MHA_BWD_MFMA") == nullptr;
        if (mfma_ok && (d == 32 || d == 64 || d == 128) && lds_mfma <= 150 * 1024 && S16 / 16 <= 4 * (128 / d)) {
            const float scale = 1.0f / sqrtf((float)d);
            // G query blocks per workgroup share one staging of K / V and one round of dK / dV atomics: as many as still leave every CU a
            // workgroup (the 256-token block of a batch-8 step: 512 -> 256 workgroups of 2 query blocks)
            static const int gmax = getenv("SBGM_MHA_BWD_G") ? atoi(getenv("SBGM_MHA_BWD_G")) : 0;
            const int qblocks = (S + 15) / 16;
            int G = gmax > 0 ? gmax : std::max(1, blocks / 256);
            G = std::max(1, std::min(G, qblocks));
            const int grid = B * heads * ((qblocks + G - 1) / G);
#define SBGM_MB(DD)                                                                                                              \
    {                                                                                                                              \
        static bool attr = false;                                                                                                  \
        if (!attr) { SBGM_HIP(hipFuncSetAttribute((const void*)mha_core_bwd_mfma_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; } \
        hipLaunchKernelGGL(mha_core_bwd_mfma_kernel<DD>, dim3(grid), dim3(256), lds_mfma, st, qkv, dout, dqkv, B, S, C, heads, scale, G);  \
    }
            if (d == 32) SBGM_MB(32) else if (d == 64) SBGM_MB(64) else SBGM_MB(128)
#undef SBGM_MB
            SBGM_LAUNCH_CHECK();
            return 0;
        }
    }
    const size_t lds_staged = ((size_t)(2 * S + 32) * (d + 1) + (size_t)32 * S) * 4;
    if (d % 4 == 0 && lds_staged <= 150 * 1024) {
        if (lds_staged > 64 * 1024)
            SBGM_HIP(hipFuncSetAttribute((const void*)mha_core_bwd_lds_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_staged));
        if (lds_staged > 64 * 1024)
            SBGM_HIP(hipFuncSetAttribute((const void*)mha_core_bwd_lds_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_staged));
        static const int wide_from = getenv("SBGM_MHA_BWD_WIDE_S") ? atoi(getenv("SBGM_MHA_BWD_WIDE_S")) : 16;
        if (S >= wide_from)
            hipLaunchKernelGGL(mha_core_bwd_lds_kernel<32>, dim3(blocks), dim3(512), lds_staged, st, qkv, dout, dqkv, B, S, C, heads,
                               1.0f / sqrtf((float)d));
        else
            hipLaunchKernelGGL(mha_core_bwd_lds_kernel<16>, dim3(blocks), dim3(256), lds_staged, st, qkv, dout, dqkv, B, S, C, heads,
                               1.0f / sqrtf((float)d));
        SBGM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(mha_core_bwd_kernel, dim3(blocks), dim3(256), (size_t)2 * 16 * S * 4, st, qkv, dout, dqkv, B, S, C, heads,
                       1.0f / sqrtf((float)(C / heads)));
    SBGM_LAUNCH_CHECK();
    return 0;
}

// Data gradient of an 8x8 / stride-2 / pad-3 convolution without the zero-dilated input (75 % of its MACs multiply zeros).
// dx[2a+py][2b+px] only meets the taps kh = py+1 (mod 2), kw = px+1 (mod 2): per output PHASE it is a 4-tap correlation over dy,
//   py = 0: dy rows a-2 .. a+1 with kh = 7, 5, 3, 1        py = 1: dy rows a-1 .. a+2 with kh = 6, 4, 2, 0
// Both windows sit inside offsets -2 .. +2, so ONE 5x5 / stride-1 / pad-2 convolution over dy with 4*Cin phase-major output
// channels ((py, px, ci); the unused tap of each phase is zero) followed by the depth->space permutation gives dx: 25 taps per
// phase pixel instead of 64 per output pixel.  This kernel builds that operator in OIHW [4*Cin][Cout][5][5] from w [Cout][Cin][8][8].
namespace {
__global__ __launch_bounds__(256) void dgrad_phase_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
    const size_t total = (size_t)4 * Cin * Cout * 25;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % 5), u = (int)((i / 5) % 5);
        size_t r = i / 25;
        const int co = (int)(r % Cout); r /= Cout;
        const int ci = (int)(r % Cin);
        const int ph = (int)(r / Cin), py = ph >> 1, px = ph & 1;
        const int kh = py == 0 ? 7 - 2 * u : 8 - 2 * u, kw = px == 0 ? 7 - 2 * v : 8 - 2 * v;     // window offset u - 2 (v - 2)
        const bool ok = (py == 0 ? u < 4 : u > 0) && (px == 0 ? v < 4 : v > 0);
        out[i] = ok ? w[(((size_t)co * Cin + ci) * 8 + kh) * 8 + kw] : 0.f;
    }
}
}  // namespace

int sbgm_launch_dgrad_phase_weight(const float* w_oihw, float* out, int Cout, int Cin, hipStream_t st) {
    SBGM_CHECK(w_oihw && out && Cout > 0 && Cin > 0, "dgrad_phase_weight: bad arguments");
    hipLaunchKernelGGL(dgrad_phase_weight_kernel, dim3(stream_blocks((size_t)4 * Cin * Cout * 25)), dim3(256), 0, st, w_oihw, out, Cout, Cin);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0, "upsample2x_bwd: C=%d", C);
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(stream_blocks((size_t)B * H * W * (C / 4))), dim3(256), 0, st, dy, dx, B, H, W, C);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_cout1_bwd(const float* dout, const float* a, const float* w_tap_c, const float* t, float sigma, float* da,
                          float* dw_tap_c, float* dbias, int B, int H, int W, int C, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0 && C <= 1024, "cout1_bwd: C=%d", C);
    hipLaunchKernelGGL(cout1_bwd_data_kernel, dim3(stream_blocks((size_t)B * H * ((W + 3) / 4) * (C / 4))), dim3(256), 0, st, dout, w_tap_c, t, sigma,
                       da, B, H, W, C);
    SBGM_LAUNCH_CHECK();
    { if (sbgm_zero_async(dw_tap_c, (size_t)9 * C * 4, st)) return 1; }
    { if (sbgm_zero_async(dbias, 4, st)) return 1; }
    const int rows = B * H, rpb = std::max(1, rows / 256);
    hipLaunchKernelGGL(cout1_bwd_weight_kernel, dim3((rows + rpb - 1) / rpb), dim3(256), (size_t)(rpb + 2) * W * 4, st, dout, a, t, sigma, dw_tap_c, dbias, B, H,
                       W, C, rpb);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_time_proj_bwd(const float* dout, const float* weight, const float* semb, const float* emb_raw, float* dW, float* dbias,
                              float* demb_accum, int B, int D, int ch, hipStream_t st) {
    hipLaunchKernelGGL(time_proj_bwd_w_kernel, dim3(stream_blocks((size_t)ch * D)), dim3(256), 0, st, dout, semb, dW, dbias, B, D, ch);
    SBGM_LAUNCH_CHECK();
    if (demb_accum) {
        hipLaunchKernelGGL(time_proj_bwd_e_kernel, dim3((B * D + 255) / 256), dim3(256), 0, st, dout, weight, emb_raw, demb_accum, B, D, ch);
        SBGM_LAUNCH_CHECK();
    }
    return 0;
}

int sbgm_launch_time_proj_multi_bwd(const float* const* douts, const float* const* sembs, float* const* dWs, float* const* dbs, const int* chs,
                                    int n_proj, int B, int D, hipStream_t st) {
    SBGM_CHECK(n_proj >= 1 && n_proj <= 16, "time_proj_multi_bwd: n_proj=%d (1..16)", n_proj);
    TimeProjBwdMulti a{};
    a.n = n_proj; a.B = B; a.D = D;
    int nb = 0;
    for (int k = 0; k < n_proj; ++k) {
        SBGM_CHECK(douts[k] && sembs[k] && dWs[k] && dbs[k] && chs[k] > 0, "time_proj_multi_bwd: null tensor in projection %d", k);
        a.dout[k] = douts[k]; a.semb[k] = sembs[k]; a.dW[k] = dWs[k]; a.dbias[k] = dbs[k]; a.ch[k] = chs[k];
        a.block_begin[k] = nb;
        nb += (int)std::min<size_t>(((size_t)chs[k] * D + 255) / 256, 512);
    }
    a.block_begin[n_proj] = nb;
    hipLaunchKernelGGL(time_proj_bwd_w_multi_kernel, dim3(nb), dim3(256), 0, st, a);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_label_emb_bwd(const float* demb, const int64_t* y, float* dtable, int B, int D, hipStream_t st) {
    hipLaunchKernelGGL(label_emb_bwd_kernel, dim3((B * D + 255) / 256), dim3(256), 0, st, demb, y, dtable, B, D);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_act_bwd(const float* x, const float* dy, float* dx, size_t n, int act, hipStream_t st) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, st, x, dy, dx, n, act);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_act_fwd(const float* x, float* y, size_t n, int act, hipStream_t st) {
    hipLaunchKernelGGL(act_fwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, st, x, y, n, act);
    SBGM_LAUNCH_CHECK();
    return 0;
}
