// 3x3 / stride 1 / pad 1 convolution with a 1-D Winograd F(2,3) along the image rows, on the fp32 MFMA pipe.
//
// For every output PAIR (ox, ox+1) and filter row kh the three kw taps are replaced by four products
//     m_xi = (G g)_xi * (B^T d)_xi,   B^T d = [d0-d2, d1+d2, d2-d1, d1-d3],   G g = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2]
//     y0 = m0 + m1 + m2,   y1 = m1 - m2 - m3
// (d0..d3 = input pixels ox-1..ox+2 of row oy+kh-1).  The sums over kh and over the input channels stay in the Winograd
// domain, so the kernel is the implicit GEMM of conv_igemm.hip with 12 "taps" (kh, xi), FOUR accumulator sets (one per xi)
// and pixel pairs as GEMM columns: 4 MFMAs where the direct form needs 6 (1.5x fewer matrix instructions), the input
// transform is 4 float4 adds per loaded quad and the output transform is lane-local.  Transform constants are 1 and 1/2:
// the result differs from the direct fp32 convolution by ordinary rounding only (measured ~1e-6 relative).
//
// Same operand scheme as conv_igemm.hip: A = packed transformed weights U[kh][cb][xi][co][16] (buffer_load_dwordx4, 1 KiB
// fragments), B = input quads gathered straight from NHWC with the bounds check supplying the zero padding, wave-level
// tiles of (16*FCO) channels x (16*FPP) pairs, optional in-workgroup split-K over the (kh, cb) steps, XCD-contiguous
// tile order, and the shared epilogue (BN scale/shift or bias, time bias, residual, ReLU, fused tap projection).
#include "common.h"
#include "kernels.h"
#include "conv_common.h"

namespace {

template <int FCO, int FPP>
struct WFrags {
    f32x4 a[4][FCO];      // [xi][co fragment]
    f32x4 d[FPP][4];      // [pair fragment][input column 0..3]
};

// WS = 8 runs 8 waves (512 threads) on one tile: two waves per SIMD instead of one for the small-map layers, whose tile count
// cannot fill the chip any other way; the partials are combined in two LDS rounds (8 -> 4 -> 1) so the scratch stays <= 64 KB.
template <int FCO, int FPP, int WS>
__global__ __launch_bounds__(WS > 4 ? 512 : 256) void conv3x3_wino_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    constexpr int TPW = WS > 4 ? 1 : 4 / WS;
    const int tslot = wave / WS, kpart = wave - tslot * WS;
    const int tile_raw = lb * TPW + tslot;
    const bool tile_ok = tile_raw < p.n_px_tiles * p.n_co_tiles;
    if (WS == 1 && !tile_ok) return;
    const int tile = tile_ok ? tile_raw : 0;
    const int co_tile = tile / p.n_px_tiles;
    const int pp_tile = tile - co_tile * p.n_px_tiles;
    const int co0 = co_tile * (16 * FCO);
    const int q0 = pp_tile * (16 * FPP);              // first pair of this tile
    const int PW = p.W >> 1;                          // pairs per image row
    const int Mp = p.B * p.H * PW;                    // total pairs

    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.wp, p.w_bytes);

    // per-lane pair decode: pair -> (b, oy, first input column ox0 - 1)
    int pbase[FPP], py0[FPP], px0[FPP];
#pragma unroll
    for (int f = 0; f < FPP; ++f) {
        const int q = q0 + 16 * f + r16;
        if (q < Mp) {
            const int row = q / PW;                   // b*H + oy
            const int b = row / p.H;
            pbase[f] = b * p.H * p.W;
            py0[f] = row - b * p.H - 1;
            px0[f] = 2 * (q - row * PW) - 1;
        } else {
            pbase[f] = 0;
            py0[f] = -(1 << 28);
            px0[f] = 0;
        }
    }

    // K steps: s = kh * CB + cb
    const int CB = p.cb_per_tap;
    int s_begin = 0, s_end = 3 * CB;
    if (WS > 1) {
        const int per = (s_end + WS - 1) / WS;
        s_begin = kpart * per;
        s_end = min(s_end, s_begin + per);
        if (!tile_ok) s_end = s_begin;
    }
    const uint32_t w_lane_off = (uint32_t)((co0 + r16) * 16 + kq * 4) * 4u;
    const uint32_t xi_stride = (uint32_t)p.Cout * 64u;            // bytes between xi slabs of one step
    int st_s, st_kh, st_cb;
    auto seek = [&](int s) { st_s = s; st_kh = s / CB; st_cb = s - st_kh * CB; };
    auto load_next = [&](WFrags<FCO, FPP>& fr) {
        const uint32_t wo = (uint32_t)st_s * 4u * xi_stride + w_lane_off;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int i = 0; i < FCO; ++i) fr.a[xi][i] = buf_load4(wr, wo + (uint32_t)xi * xi_stride + (uint32_t)i * (16u * 64u));
        const int coff = st_cb * 16 + kq * 4;
#pragma unroll
        for (int f = 0; f < FPP; ++f) {
            const int iy = py0[f] + st_kh;
            const bool rok = (unsigned)iy < (unsigned)p.H;
            const int rowoff = pbase[f] + iy * p.W;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ix = px0[f] + c;
                const bool ok = rok & ((unsigned)ix < (unsigned)p.W);
                fr.d[f][c] = buf_load4(xr, ok ? (uint32_t)((rowoff + ix) * p.Cs + coff) * 4u : 0x80000000u);
            }
        }
        ++st_s;
        if (++st_cb == CB) { st_cb = 0; ++st_kh; }
    };

    f32x4 acc[4][FCO][FPP];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int i = 0; i < FCO; ++i)
#pragma unroll
            for (int j = 0; j < FPP; ++j) acc[xi][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const WFrags<FCO, FPP>& fr) {
        f32x4 v[4][FPP];
#pragma unroll
        for (int j = 0; j < FPP; ++j) {                 // input transform B^T d
            v[0][j] = fr.d[j][0] - fr.d[j][2];
            v[1][j] = fr.d[j][1] + fr.d[j][2];
            v[2][j] = fr.d[j][2] - fr.d[j][1];
            v[3][j] = fr.d[j][1] - fr.d[j][3];
        }
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < FCO; ++i)
#pragma unroll
                    for (int j = 0; j < FPP; ++j)
                        acc[xi][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.a[xi][i][k], v[xi][j][k], acc[xi][i][j], 0, 0, 0);
    };

    WFrags<FCO, FPP> f0, f1;
    int s = s_begin;
    if (s < s_end) {
        seek(s);
        load_next(f0);
        for (; s + 2 <= s_end; s += 2) {
            load_next(f1);
            compute(f0);
            load_next(f0);
            compute(f1);
        }
        if (s < s_end) compute(f0);
    }

    if (WS > 1) {                                       // in-workgroup split-K: combine the Winograd-domain partials
        f32x4* red = reinterpret_cast<f32x4*>(smem_raw);
        constexpr int NF = 4 * FCO * FPP;
        auto put = [&](int slot) {
            f32x4* dst = red + (size_t)slot * NF * 64 + lane;
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                for (int i = 0; i < FCO; ++i)
#pragma unroll
                    for (int j = 0; j < FPP; ++j) dst[((xi * FCO + i) * FPP + j) * 64] = acc[xi][i][j];
        };
        auto add = [&](int slot) {
            const f32x4* src = red + (size_t)slot * NF * 64 + lane;
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                for (int i = 0; i < FCO; ++i)
#pragma unroll
                    for (int j = 0; j < FPP; ++j) acc[xi][i][j] += src[((xi * FCO + i) * FPP + j) * 64];
        };
        if (WS == 8) {                                  // round 1: waves 4..7 hand over to waves 0..3
            if (kpart >= 4) put(kpart - 4);
            __syncthreads();
            if (kpart < 4) add(kpart);
            __syncthreads();                            // the slots are reused by round 2; every wave stays until the last barrier
        }
        constexpr int W2 = WS == 8 ? 4 : WS;
        if (kpart > 0 && kpart < W2) put(tslot * (W2 - 1) + (kpart - 1));
        __syncthreads();
        if (kpart > 0 || !tile_ok) return;
#pragma unroll
        for (int k = 0; k < W2 - 1; ++k) add(tslot * (W2 - 1) + k);
    }

    // ---- output transform + epilogue: lane owns channels co0+16i+4kq..+3 of the two pixels of pair q0+16j+r16 ----------
    const int hw = p.H * p.W;
#pragma unroll
    for (int j = 0; j < FPP; ++j) {
        const int q = q0 + 16 * j + r16;
        const bool ok = q < Mp;
        const int qq = ok ? q : 0;
        const int row = qq / PW;
        const int m_even = row * p.W + 2 * (qq - row * PW);      // linear NHWC pixel index of the pair's first pixel
        const int b = m_even / hw;
        f32x4 y0[FCO], y1[FCO];
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
            const int co = co0 + 16 * i + 4 * kq;
            y0[i] = conv_epilogue(acc[0][i][j] + acc[1][i][j] + acc[2][i][j], p, co, (size_t)m_even, b);
            y1[i] = conv_epilogue(acc[1][i][j] - acc[2][i][j] - acc[3][i][j], p, co, (size_t)m_even + 1, b);
        }
        if (p.proj_w != nullptr) {                      // fused tap projection of the final block (see conv_igemm.hip)
            const float* wl = p.proj_w + co0 + 4 * kq;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int i = 0; i < FCO; ++i) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + tap * p.Cout + 16 * i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s0 = fmaf(y0[i][e], w4[e], s0); s1 = fmaf(y1[i][e], w4[e], s1); }
                }
                s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
                s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                if (ok && kq == (tap & 3)) {
                    p.proj_out[(size_t)tap * p.M + m_even] = s0;
                    p.proj_out[(size_t)tap * p.M + m_even + 1] = s1;
                }
            }
        } else if (ok) {
#pragma unroll
            for (int i = 0; i < FCO; ++i) {
                const int co = co0 + 16 * i + 4 * kq;
                *reinterpret_cast<f32x4*>(p.out + (size_t)m_even * p.Cout + co) = y0[i];
                *reinterpret_cast<f32x4*>(p.out + ((size_t)m_even + 1) * p.Cout + co) = y1[i];
            }
        }
    }
}

// OIHW [Cout][Cin][3][3] -> U[kh][cb][xi][Cout][16],  U_xi = sum_kw G[xi][kw] * w[kh][kw]
__global__ void pack_wino_weight_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin, int cs) {
    const int CB = cs / 16;
    const size_t total = (size_t)3 * CB * 4 * Cout * 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c16 = (int)(i & 15);
        size_t r = i >> 4;
        const int co = (int)(r % Cout); r /= Cout;
        const int xi = (int)(r & 3); r >>= 2;
        const int cb = (int)(r % CB);
        const int kh = (int)(r / CB);
        const int c = cb * 16 + c16;
        float v = 0.f;
        if (c < Cin) {
            const float* g = w + (((size_t)co * Cin + c) * 3 + kh) * 3;
            const float g0 = g[0], g1 = g[1], g2 = g[2];
            v = xi == 0 ? g0 : xi == 1 ? 0.5f * ((g0 + g1) + g2) : xi == 2 ? 0.5f * ((g0 - g1) + g2) : g2;
        }
        up[i] = v;
    }
}

}  // namespace

size_t sbgm_wino_packed_floats(int Cout, int cs) { return (size_t)3 * (cs / 16) * 4 * Cout * 16; }

int sbgm_launch_pack_wino_weight(const float* w_oihw, float* up, int Cout, int Cin, int cs, hipStream_t st) {
    SBGM_CHECK(cs % 16 == 0 && Cin <= cs, "pack_wino: padded Cin %d must be a multiple of 16", cs);
    const size_t total = sbgm_wino_packed_floats(Cout, cs);
    hipLaunchKernelGGL(pack_wino_weight_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, st, w_oihw, up,
                       Cout, Cin, cs);
    SBGM_LAUNCH_CHECK();
    return 0;
}

// cfg.fpx counts PAIR fragments here (a wave tile is 16*fco channels x 32*fpx pixels); cfg.splits must be 1.
int sbgm_launch_conv_wino(ConvParams p, const ConvTile& cfg, hipStream_t st) {
    SBGM_CHECK(p.Cs % 16 == 0 && p.W % 2 == 0, "wino: needs Cin padded to 16 and an even width (Cs=%d W=%d)", p.Cs, p.W);
    SBGM_CHECK(p.Cout % (16 * cfg.fco) == 0, "wino: Cout=%d not a multiple of the %d-row tile", p.Cout, 16 * cfg.fco);
    SBGM_CHECK(cfg.splits <= 1 && (cfg.ws == 1 || cfg.ws == 2 || cfg.ws == 4 || cfg.ws == 8), "wino: no grid split-K; waves-per-tile 1, 2, 4 or 8");
    SBGM_CHECK(p.act == SBGM_ACT_NONE || p.act == SBGM_ACT_RELU || p.act == SBGM_ACT_GELU, "wino: act=%d does not fuse", p.act);
    SBGM_CHECK((size_t)p.B * p.H * p.W * p.Cs * 4 < (1ull << 31), "wino: input tensor exceeds 2 GiB buffer window");
    SBGM_CHECK(p.proj_w == nullptr || (p.Cout == 16 * cfg.fco && p.proj_out != nullptr), "wino: tap projection needs one co tile");
    p.OH = p.H; p.OW = p.W;
    p.M = p.B * p.H * p.W;
    p.cb_per_tap = p.Cs / 16;
    p.nsteps = 3 * p.cb_per_tap;
    const int Mp = p.M / 2;
    p.n_px_tiles = (Mp + 16 * cfg.fpx - 1) / (16 * cfg.fpx);
    p.n_co_tiles = p.Cout / (16 * cfg.fco);
    p.x_bytes = (uint32_t)((size_t)p.B * p.H * p.W * p.Cs * 4);
    p.w_bytes = (uint32_t)(sbgm_wino_packed_floats(p.Cout, p.Cs) * 4);
    const int ntiles = p.n_px_tiles * p.n_co_tiles, tpw = cfg.ws > 4 ? 1 : 4 / cfg.ws;
    dim3 grid((ntiles + tpw - 1) / tpw);
    int rc = 1;
#define SBGM_W(FC, FP, W_)                                                                                   \
    if (cfg.fco == FC && cfg.fpx == FP && cfg.ws == W_) {                                                     \
        const size_t lds = W_ == 8 ? (size_t)4 * 4 * FC * FP * 64 * 16                                         \
                                   : (W_ > 1 ? (size_t)(4 / W_) * (W_ - 1) * 4 * FC * FP * 64 * 16 : 0);       \
        if (lds > 64 * 1024)                                                                                  \
            SBGM_HIP(hipFuncSetAttribute((const void*)conv3x3_wino_kernel<FC, FP, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((conv3x3_wino_kernel<FC, FP, W_>), grid, dim3(W_ > 4 ? 512 : 256), lds, st, p);   \
        rc = 0;                                                                                              \
    }
#define SBGM_WT(W_) SBGM_W(4, 1, W_) SBGM_W(2, 2, W_) SBGM_W(2, 1, W_) SBGM_W(4, 2, W_)
    SBGM_WT(1) SBGM_WT(2) SBGM_WT(4) SBGM_WT(8)
#undef SBGM_WT
#undef SBGM_W
    SBGM_CHECK(rc == 0, "wino: no kernel for tile=%dx%d ws=%d", cfg.fco, cfg.fpx, cfg.ws);
    SBGM_LAUNCH_CHECK();
    return 0;
}
