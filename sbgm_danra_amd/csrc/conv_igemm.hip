// fp32 implicit-GEMM convolution on the CDNA4 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain,
// same peak as the fp32 vector ALU), NHWC activations, wave-level tiles, no LDS and no barriers.
//
// Replaces every nn.Conv2d / nn.Linear instance on the reference hot path
// (reference sbgm/score_unet.py:206-219 stem convs, torchvision BasicBlock convs used at :188,
//  :468/:489 decoder convs, :127-134 attention projections / FF).
//
//   D[co][px] = sum_{tap, c}  Wp[step(tap,c/16)][co][c%16] * X[b, oy*S-PAD+kh, ox*S-PAD+kw, c]
//
// GEMM roles: A operand = packed weights (rows = output channels), B operand = gathered input pixels
// (columns = output pixels), so each lane ends up owning 4 consecutive output channels of one pixel and
// the epilogue stores one float4 per 16x16 fragment straight into NHWC.
//
// One wave computes an (16*FCO) x (16*FPX) output tile.  Per K-step (16 input channels of one filter tap)
// a lane issues FCO + FPX 16-byte buffer loads (hardware bounds check supplies the zero padding) and
// 4*FCO*FPX MFMAs; the next step's operands are prefetched into a second register set.  Tiles are mapped to
// workgroups so that an XCD owns a contiguous range of (co-tile, px-tile) pairs (weights stay in its L2).
// Small-M layers are filled two ways: WS waves of a workgroup split the K range of ONE tile and combine their
// accumulators through LDS before the epilogue (in-workgroup split-K: more waves per SIMD to hide the operand
// latency, no extra launch, no global traffic), and, for the tiniest layers, split-K over gridDim.y followed by
// splitk_reduce_epilogue.
#include "../../include/sbgm_hip.h"
#include "common.h"
#include "kernels.h"
#include "conv_common.h"

namespace {

template <int FCO, int FPX>
struct Frags {
    f32x4 a[FCO];
    f32x4 b[FPX];
};

struct PixLane {      // per-lane decode of the output pixel this lane gathers for / stores to
    int base;         // b * H * W  (input pixel index of (b,0,0)); negative => lane inactive
    int y0, x0;       // oy*S - PAD, ox*S - PAD
};

template <int KH, int KW, int S, int PAD, int FCO, int FPX, int CMODE, int WS>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15;     // row (co) for the A fragment, column (pixel) for the B fragment
    const int kq = lane >> 4;      // which 4-channel quad of the 16-deep K step this lane supplies

    // ---- workgroup -> tile range, XCD-contiguous (blocks b, b+8, ... share an XCD) ---------------
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr int TPW = 4 / WS;                     // tiles per workgroup
    const int tslot = wave / WS, kpart = wave - tslot * WS;
    const int tile_raw = lb * TPW + tslot;
    const bool tile_ok = tile_raw < p.n_px_tiles * p.n_co_tiles;
    if (WS == 1 && !tile_ok) return;                // no barrier below in that configuration
    const int tile = tile_ok ? tile_raw : 0;
    const int co_tile = tile / p.n_px_tiles;
    const int px_tile = tile - co_tile * p.n_px_tiles;
    const int co0 = co_tile * (16 * FCO);
    const int m0 = px_tile * (16 * FPX);

    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.wp, p.w_bytes);

    // ---- per-lane pixel decode -------------------------------------------------------------------
    PixLane pl[FPX];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int f = 0; f < FPX; ++f) {
        const int m = m0 + 16 * f + r16;
        if (m < p.M) {
            const int b = m / ohw;
            const int rr = m - b * ohw;
            const int oy = rr / p.OW;
            const int ox = rr - oy * p.OW;
            pl[f].base = b * p.H * p.W;
            pl[f].y0 = oy * S - PAD;
            pl[f].x0 = ox * S - PAD;
        } else {
            pl[f].base = 0;
            pl[f].y0 = -(1 << 28);
            pl[f].x0 = 0;
        }
    }

    // ---- K range of this split ---------------------------------------------------------------------
    int s_begin = blockIdx.y * p.steps_per_split;
    int s_end = min(p.nsteps, s_begin + p.steps_per_split);
    if (WS > 1) {                                   // this wave's share of the split's K range
        const int len = s_end - s_begin, per = (len + WS - 1) / WS;
        s_begin = s_begin + kpart * per;
        s_end = min(s_end, s_begin + per);
        if (!tile_ok) s_end = s_begin;
    }

    const uint32_t w_lane_off = (uint32_t)((co0 + r16) * 16 + kq * 4) * 4u;
    const uint32_t w_step_stride = (uint32_t)p.Cout * 64u;   // bytes per K step: Cout rows x 16 floats

    // running K-step state (scalar registers): step -> (kh, kw0, cb); advanced once per load
    int st_s, st_kh, st_kw, st_cb;
    auto seek = [&](int s) {
        st_s = s;
        if (CMODE == 0) {
            const int tap = s / p.cb_per_tap;
            st_cb = s - tap * p.cb_per_tap;
            st_kh = tap / KW;
            st_kw = tap - st_kh * KW;
        } else {
            constexpr int SPR = KW / (16 / (CMODE ? CMODE : 16)) > 0 ? KW / (16 / (CMODE ? CMODE : 16)) : 1;   // steps per filter row
            st_kh = s / SPR;
            st_kw = (s - st_kh * SPR) * (16 / (CMODE ? CMODE : 16));
            st_cb = 0;
        }
    };
    auto load_next = [&](Frags<FCO, FPX>& fr) {   // loads step st_s, then advances the state
        int kw_lane, coff;
        if (CMODE == 0) { kw_lane = st_kw; coff = st_cb * 16 + kq * 4; }
        else if (CMODE == 4) { kw_lane = st_kw + kq; coff = 0; }
        else if (CMODE == 2) { kw_lane = st_kw + 2 * kq; coff = 0; }       // two taps x the 2 real channels of 4-slot pixels
        else { kw_lane = st_kw + (kq >> 1); coff = (kq & 1) * 4; }
        const uint32_t wo = (uint32_t)st_s * w_step_stride + w_lane_off;
#pragma unroll
        for (int f = 0; f < FCO; ++f) fr.a[f] = buf_load4(wr, wo + (uint32_t)f * (16u * 64u));
#pragma unroll
        for (int f = 0; f < FPX; ++f) {
            int iy = pl[f].y0 + st_kh, ix = pl[f].x0 + kw_lane;
            bool ok = true;
            if (p.in_dil == 2) {          // transposed (stride-2) convolution: the source is read through a zero-inserted grid
                ok = ((iy | ix) & 1) == 0;
                iy >>= 1;
                ix >>= 1;
            }
            if (CMODE == 2) {                                   // taps ix, ix+1: the first 8 bytes of two neighbouring pixels
                const bool oky = (unsigned)iy < (unsigned)p.H;
                const uint32_t off = (uint32_t)((pl[f].base + iy * p.W + ix) * p.Cs) * 4u;
                const f32x2 lo = buf_load2(xr, (oky & ((unsigned)ix < (unsigned)p.W)) ? off : 0x80000000u);
                const f32x2 hi = buf_load2(xr, (oky & ((unsigned)(ix + 1) < (unsigned)p.W)) ? off + (uint32_t)p.Cs * 4u : 0x80000000u);
                fr.b[f] = f32x4{lo[0], lo[1], hi[0], hi[1]};
                continue;
            }
            ok = ok & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const uint32_t off = (uint32_t)((pl[f].base + iy * p.W + ix) * p.Cs + coff) * 4u;
            fr.b[f] = buf_load4(xr, ok ? off : 0x80000000u);
        }
        ++st_s;
        if (CMODE == 0) {
            if (++st_cb == p.cb_per_tap) { st_cb = 0; if (++st_kw == KW) { st_kw = 0; ++st_kh; } }
        } else {
            st_kw += 16 / (CMODE ? CMODE : 16);
            if (st_kw == KW) { st_kw = 0; ++st_kh; }
        }
    };

    f32x4 acc[FCO][FPX];
#pragma unroll
    for (int i = 0; i < FCO; ++i)
#pragma unroll
        for (int j = 0; j < FPX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const Frags<FCO, FPX>& fr) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < FCO; ++i)
#pragma unroll
                for (int j = 0; j < FPX; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.a[i][k], fr.b[j][k], acc[i][j], 0, 0, 0);
    };

    // Software pipeline, two register sets.  The pair loop body is branch-free so the loads of step s+1
    // stay ahead of the MFMAs of step s (a `break` between them lets LLVM sink the loads to their use).
    // Loading one step past s_end is harmless: weights come back 0 from the bounds check or belong to the
    // next split, and the result is never multiplied in.
    Frags<FCO, FPX> f0, f1;
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

    // ---- in-workgroup split-K: waves 1..WS-1 of a tile hand their accumulators to wave 0 through LDS --------------
    if (WS > 1) {
        f32x4* red = reinterpret_cast<f32x4*>(smem_raw);      // [tslot][kpart-1][frag][lane]
        constexpr int NF = FCO * FPX;
        if (kpart > 0) {
            f32x4* dst = red + ((tslot * (WS - 1) + (kpart - 1)) * NF) * 64 + lane;
#pragma unroll
            for (int i = 0; i < FCO; ++i)
#pragma unroll
                for (int j = 0; j < FPX; ++j) dst[(i * FPX + j) * 64] = acc[i][j];
        }
        __syncthreads();
        if (kpart > 0 || !tile_ok) return;
#pragma unroll
        for (int k = 0; k < WS - 1; ++k) {
            const f32x4* src = red + ((tslot * (WS - 1) + k) * NF) * 64 + lane;
#pragma unroll
            for (int i = 0; i < FCO; ++i)
#pragma unroll
                for (int j = 0; j < FPX; ++j) acc[i][j] += src[(i * FPX + j) * 64];
        }
    }

    // ---- fused tap projection (final decoder block): the 3x3, Cout=1 convolution that follows is linear, so each
    // output pixel only needs the 9 per-tap dot products d[tap][m] = sum_co w[tap][co] * y[m][co] of THIS layer's
    // output; y itself is never written.  The wave tile spans all output channels (checked on the host).
    if (p.proj_w != nullptr) {
        bool ok[FPX];
        int mrow[FPX];
#pragma unroll
        for (int j = 0; j < FPX; ++j) {
            const int m = m0 + 16 * j + r16;
            ok[j] = m < p.M;
            mrow[j] = ok[j] ? m : 0;
            const int b = mrow[j] / ohw;
#pragma unroll
            for (int i = 0; i < FCO; ++i) acc[i][j] = conv_epilogue(acc[i][j], p, co0 + 16 * i + 4 * kq, (size_t)mrow[j], b);
        }
        const float* wl = p.proj_w + co0 + 4 * kq;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {          // fully unrolled: tap t+1's weights load under tap t's math
            f32x4 w4[FCO];
#pragma unroll
            for (int i = 0; i < FCO; ++i) w4[i] = *reinterpret_cast<const f32x4*>(wl + tap * p.Cout + 16 * i);
#pragma unroll
            for (int j = 0; j < FPX; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < FCO; ++i) {
                    s = fmaf(acc[i][j][0], w4[i][0], s);
                    s = fmaf(acc[i][j][1], w4[i][1], s);
                    s = fmaf(acc[i][j][2], w4[i][2], s);
                    s = fmaf(acc[i][j][3], w4[i][3], s);
                }
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (ok[j] && kq == (tap & 3)) p.proj_out[(size_t)tap * p.M + mrow[j]] = s;
            }
        }
        return;
    }

    // ---- epilogue: lane owns channels co0+16i+4kq..+3 of pixel m0+16j+r16 ------------------------------
    const bool partial = gridDim.y > 1;
    float* outp = p.out + (partial ? (size_t)blockIdx.y * (size_t)p.M * p.Cout : 0);
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const int m = m0 + 16 * j + r16;
        if (m >= p.M) continue;
        const int b = m / ohw;
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
            const int co = co0 + 16 * i + 4 * kq;
            f32x4 v = acc[i][j];
            if (!partial) v = conv_epilogue(v, p, co, (size_t)m, b);
            *reinterpret_cast<f32x4*>(outp + (size_t)m * p.Cout + co) = v;
        }
    }
}

// Sum split-K partials [nsplit][M][Cout] and apply the same epilogue.  One float4 per thread.
__global__ __launch_bounds__(256) void splitk_reduce_epilogue(const ConvParams p, const float* part, int nsplit) {
    const size_t n4 = (size_t)p.M * p.Cout / 4;
    const size_t stride = (size_t)p.M * p.Cout;
    const int ohw = p.OH * p.OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(part + e);
        for (int s = 1; s < nsplit; ++s) v += *reinterpret_cast<const f32x4*>(part + s * stride + e);
        const int m = (int)(e / p.Cout);
        const int co = (int)(e - (size_t)m * p.Cout);
        const int b = m / ohw;
        *reinterpret_cast<f32x4*>(p.out + e) = conv_epilogue(v, p, co, (size_t)m, b);
    }
}

// OIHW -> packed [step][Cout][16] (see header comment).  cs = padded input channel count.
// `transposed` != 0: pack the data-gradient operator instead: W'[o=ci][i=co][kh][kw] = W[co][ci][KH-1-kh][KW-1-kw]
// (Cout/Cin below are then the TRANSPOSED operator's sizes, i.e. Cout = forward Cin).
__device__ __forceinline__ float pack_conv_weight_value(const float* __restrict__ w, size_t i, int Cout, int Cin, int KH, int KW, int cs,
                                                        int transposed) {
    const int k16 = (int)(i & 15);
    const int co = (int)((i >> 4) % Cout);
    const int s = (int)((i >> 4) / Cout);
    int kh, kw, c;
    if (cs >= 16) {
        const int cb_per_tap = cs / 16;
        const int tap = s / cb_per_tap, cb = s - tap * cb_per_tap;
        kh = tap / KW; kw = tap - kh * KW; c = cb * 16 + k16;
    } else if (cs == 4) {
        kh = s / (KW / 4); kw = (s - kh * (KW / 4)) * 4 + (k16 >> 2); c = k16 & 3;
    } else if (cs == 2) {
        kh = s / (KW / 8); kw = (s - kh * (KW / 8)) * 8 + (k16 >> 1); c = k16 & 1;
    } else {  // cs == 8
        kh = s / (KW / 2); kw = (s - kh * (KW / 2)) * 2 + (k16 >> 3); c = k16 & 7;
    }
    if (c >= Cin) return 0.f;
    if (!transposed) return w[(((size_t)co * Cin + c) * KH + kh) * KW + kw];
    return w[(((size_t)c * Cout + co) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];   // forward layout [Cin'][Cout'] = [c][co]
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                        int KH, int KW, int cs, int nsteps, int transposed) {
    const size_t total = (size_t)nsteps * Cout * 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        wp[i] = pack_conv_weight_value(w, i, Cout, Cin, KH, KW, cs, transposed);
}

// Many weights in one launch (training: every conv weight changes every optimizer step and is needed twice, as the forward
// operator and as the data-gradient operator).  desc[k].block_begin is the exclusive prefix of 256-element blocks.
// Weights with >= 16 padded input channels and Cout % 16 == 0 are packed tile by tile through LDS: a block takes 16 output
// channels x 16 input channels x all T taps, which are 16 contiguous source runs of 16*T floats (either operator: forward
// rows are co with (ci, tap) contiguous, data-gradient rows are the forward co = packed c with (ci = packed co, tap)
// contiguous) and T contiguous destination planes of 256 floats.  The element-wise gather it replaces read 4 bytes per lane
// at a 36-byte (forward) or Cout*36-byte (data-gradient) stride: 145 us per training step for 228 MB.
// Other weights (the stem's 2/4/8-channel layouts, kernels with more than 16 taps) keep one block per 256 packed elements.
__host__ __device__ inline bool pack_tiled(int Cout, int cs, int taps) { return cs >= 16 && Cout % 16 == 0 && taps <= 16; }
__global__ __launch_bounds__(256) void pack_conv_weights_batched_kernel(const sbgm_pack_desc* __restrict__ desc, int n) {
    extern __shared__ float pk[];                            // [16 rows][16*Tp + 1]
    int lo = 0, hi = n - 1;                                  // last descriptor whose first block is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    sbgm_pack_desc d = desc[lo];
    const bool wino = (d.transposed & 2) != 0;               // validated on the host: 3x3, tiled
    const bool w2d = (d.transposed & 4) != 0;                // F(2x2,3x3) image U[cb][xi*4+eta][Cout][16] (conv_w2d.hip), same validation
    d.transposed &= 1;
    const int rel = blockIdx.x - d.block_begin;
    if (!pack_tiled(d.Cout, d.cs, d.KH * d.KW)) {
        const size_t total = (size_t)d.nsteps * d.Cout * 16;
        const size_t i = (size_t)rel * 256 + threadIdx.x;
        if (i < total) d.dst[i] = pack_conv_weight_value(d.src, i, d.Cout, d.Cin, d.KH, d.KW, d.cs, d.transposed);
        return;
    }
    const int T = d.KH * d.KW, Tp = T | 1, RS = 16 * Tp + 1, cbs = d.cs / 16;
    const int cb = rel % cbs, co0 = (rel / cbs) * 16, c0 = cb * 16;
    // source rows: forward w[co][c][tap] -> row = co, inner = c;  data-gradient w[c][co][tap] -> row = c, inner = co
    const int row0 = d.transposed ? c0 : co0, in0 = d.transposed ? co0 : c0;
    const int n_rows = d.transposed ? d.Cin : d.Cout, n_in = d.transposed ? d.Cout : d.Cin;     // source extents
    for (int e = threadIdx.x; e < 256 * T; e += 256) {
        const int r = e / (16 * T), m = e - r * 16 * T;      // m = inner * T + tap: contiguous in the source
        const int inner = m / T, tap = m - inner * T;
        float v = 0.f;
        if (row0 + r < n_rows && in0 + inner < n_in) v = d.src[((size_t)(row0 + r) * n_in + in0) * T + m];
        pk[r * RS + inner * Tp + tap] = v;
    }
    __syncthreads();
    const int k16 = threadIdx.x & 15, col = threadIdx.x >> 4;   // destination: c = c0 + k16, co = co0 + col
    if (w2d) {                                               // U = G g G^T: rows first, then columns (as pack_w2d_weight_kernel)
        float rowv[4][3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            float g[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int tap = kh * 3 + kw;
                g[kh] = d.transposed ? pk[k16 * RS + col * Tp + (T - 1 - tap)] : pk[col * RS + k16 * Tp + tap];
            }
            rowv[0][kw] = g[0]; rowv[1][kw] = 0.5f * ((g[0] + g[1]) + g[2]); rowv[2][kw] = 0.5f * ((g[0] - g[1]) + g[2]); rowv[3][kw] = g[2];
        }
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const float r0 = rowv[xi][0], r1 = rowv[xi][1], r2 = rowv[xi][2];
            const float u[4] = {r0, 0.5f * ((r0 + r1) + r2), 0.5f * ((r0 - r1) + r2), r2};
#pragma unroll
            for (int eta = 0; eta < 4; ++eta) d.dst[((size_t)(cb * 16 + xi * 4 + eta) * d.Cout + co0) * 16 + threadIdx.x] = u[eta];
        }
        return;
    }
    if (wino) {                                              // U[kh][cb][xi][Cout][16], U = G g along the filter row (conv_wino.hip)
        for (int kh = 0; kh < 3; ++kh) {
            float g[3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tap = kh * 3 + kw;
                g[kw] = d.transposed ? pk[k16 * RS + col * Tp + (T - 1 - tap)] : pk[col * RS + k16 * Tp + tap];
            }
            const float u[4] = {g[0], 0.5f * ((g[0] + g[1]) + g[2]), 0.5f * ((g[0] - g[1]) + g[2]), g[2]};
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) d.dst[((size_t)((kh * cbs + cb) * 4 + xi) * d.Cout + co0) * 16 + threadIdx.x] = u[xi];
        }
        return;
    }
    for (int tap = 0; tap < T; ++tap) {
        const float v = d.transposed ? pk[k16 * RS + col * Tp + (T - 1 - tap)] : pk[col * RS + k16 * Tp + tap];
        d.dst[((size_t)(tap * cbs + cb) * d.Cout + co0) * 16 + threadIdx.x] = v;
    }
}

template <int KH, int KW, int S, int PAD, int CMODE>
int launch_geom(const ConvParams& p, const ConvTile& t, dim3 grid, hipStream_t st) {
#define SBGM_TILE(FC, FP, W)                                                                                  \
    if (t.fco == FC && t.fpx == FP && t.ws == W) {                                                            \
        const size_t lds = W > 1 ? (size_t)(4 / W) * (W - 1) * FC * FP * 64 * 16 : 0;                          \
        hipLaunchKernelGGL((conv_igemm_kernel<KH, KW, S, PAD, FC, FP, CMODE, W>), grid, dim3(256), lds, st, p); \
        return 0;                                                                                             \
    }
#define SBGM_TILES(W) SBGM_TILE(4, 4, W) SBGM_TILE(4, 2, W) SBGM_TILE(4, 1, W) SBGM_TILE(2, 4, W) SBGM_TILE(2, 2, W) SBGM_TILE(2, 1, W)
    SBGM_TILES(1) SBGM_TILES(2) SBGM_TILES(4)
#undef SBGM_TILES
#undef SBGM_TILE
    return 1;
}

}  // namespace

int sbgm_conv_nsteps(int KH, int KW, int cs) {
    if (cs >= 16) return KH * KW * (cs / 16);
    return KH * (KW / (16 / cs));
}

int sbgm_launch_pack_conv_weight(const float* w_oihw, float* wp, int Cout, int Cin, int KH, int KW, int cs,
                                 hipStream_t st, int transposed) {
    SBGM_CHECK((cs == 2 && KW % 8 == 0) || cs == 4 || cs == 8 || (cs >= 16 && cs % 16 == 0), "pack_conv_weight: bad padded Cin %d", cs);
    SBGM_CHECK(cs >= 16 || KW % (16 / cs) == 0, "pack_conv_weight: KW=%d not divisible for cs=%d", KW, cs);
    SBGM_CHECK(Cin <= cs, "pack_conv_weight: Cin %d > padded %d", Cin, cs);
    const int nsteps = sbgm_conv_nsteps(KH, KW, cs);
    const size_t total = (size_t)nsteps * Cout * 16;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, st, w_oihw, wp, Cout, Cin, KH, KW, cs, nsteps,
                       transposed);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_conv_pack_blocks(int Cout, int KH, int KW, int cs) {
    if (pack_tiled(Cout, cs, KH * KW)) return (cs / 16) * (Cout / 16);
    return (int)(((size_t)sbgm_conv_nsteps(KH, KW, cs) * Cout * 16 + 255) / 256);
}

int sbgm_launch_pack_conv_weights_batched(const sbgm_pack_desc* desc_dev, int n, int total_blocks, hipStream_t st) {
    SBGM_CHECK(desc_dev && n >= 1 && total_blocks >= 1, "pack_weights_batched: bad arguments");
    constexpr int lds = 16 * (16 * 17 + 1) * 4;            // up to 16 taps
    hipLaunchKernelGGL(pack_conv_weights_batched_kernel, dim3(total_blocks), dim3(256), lds, st, desc_dev, n);
    SBGM_LAUNCH_CHECK();
    return 0;
}

// Launch one convolution.  `cfg` selects the wave tile and the split-K factor; `partial_ws` must hold
// splits*M*Cout floats when splits > 1.
int sbgm_launch_conv(const ConvGeom& g, ConvParams p, const ConvTile& cfg, float* partial_ws, hipStream_t st) {
    SBGM_CHECK(p.Cout % (16 * cfg.fco) == 0, "conv: Cout=%d not a multiple of the %d-row tile", p.Cout, 16 * cfg.fco);
    SBGM_CHECK(p.Cs == 4 || p.Cs == 8 || p.Cs % 16 == 0, "conv: padded Cin %d unsupported", p.Cs);
    SBGM_CHECK(p.act == SBGM_ACT_NONE || p.act == SBGM_ACT_RELU || p.act == SBGM_ACT_GELU, "conv: act=%d does not fuse into the epilogue", p.act);
    SBGM_CHECK(cfg.ws == 1 || cfg.ws == 2 || cfg.ws == 4, "conv: waves-per-tile %d must be 1, 2 or 4", cfg.ws);
    SBGM_CHECK(p.proj_w == nullptr || (p.Cout == 16 * cfg.fco && cfg.splits <= 1 && p.proj_out != nullptr),
               "conv: the fused tap projection needs one wave tile over all %d output channels and no grid split-K", p.Cout);
    SBGM_CHECK((size_t)p.B * p.H * p.W * p.Cs * 4 < (1ull << 31), "conv: input tensor exceeds 2 GiB buffer window");
    if (p.in_dil != 2) p.in_dil = 1;
    if (p.out_h > 0 && p.out_w > 0) {           // explicit output size (data-gradient of a strided convolution)
        p.OH = p.out_h;
        p.OW = p.out_w;
    } else {
        SBGM_CHECK(p.in_dil == 1, "conv: a dilated-input (transposed) convolution needs an explicit output size");
        p.OH = (p.H + 2 * g.pad - g.kh) / g.stride + 1;
        p.OW = (p.W + 2 * g.pad - g.kw) / g.stride + 1;
    }
    p.M = p.B * p.OH * p.OW;
    const bool two_of_four = p.c_real == 2 && p.Cs == 4;
    SBGM_CHECK(p.c_real == 0 || two_of_four, "conv: c_real=%d is only defined for 2 real channels in 4-slot pixels", p.c_real);
    p.cb_per_tap = p.Cs >= 16 ? p.Cs / 16 : 1;
    p.nsteps = sbgm_conv_nsteps(g.kh, g.kw, two_of_four ? 2 : p.Cs);
    const int splits = std::max(1, std::min(cfg.splits, p.nsteps));
    p.steps_per_split = (p.nsteps + splits - 1) / splits;
    const int real_splits = (p.nsteps + p.steps_per_split - 1) / p.steps_per_split;
    p.n_px_tiles = (p.M + 16 * cfg.fpx - 1) / (16 * cfg.fpx);
    p.n_co_tiles = p.Cout / (16 * cfg.fco);
    p.x_bytes = (uint32_t)((size_t)p.B * p.H * p.W * p.Cs * 4);
    p.w_bytes = (uint32_t)((size_t)p.nsteps * p.Cout * 64);
    const int ntiles = p.n_px_tiles * p.n_co_tiles;
    const int tpw = 4 / cfg.ws;
    dim3 grid((ntiles + tpw - 1) / tpw, real_splits);
    float* final_out = p.out;
    if (real_splits > 1) {
        SBGM_CHECK(partial_ws != nullptr, "conv: split-K needs a partial workspace");
        p.out = partial_ws;
    }
    const int cmode = two_of_four ? 2 : (p.Cs >= 16 ? 0 : p.Cs);
    int rc = 1;
#define SBGM_GEOM(KH_, KW_, S_, PAD_, CM_)                                                        \
    if (g.kh == KH_ && g.kw == KW_ && g.stride == S_ && g.pad == PAD_ && cmode == CM_)            \
        rc = launch_geom<KH_, KW_, S_, PAD_, CM_>(p, cfg, grid, st);
    SBGM_GEOM(8, 8, 2, 3, 0) SBGM_GEOM(8, 8, 2, 3, 4) SBGM_GEOM(8, 8, 2, 3, 8) SBGM_GEOM(8, 8, 2, 3, 2)
    SBGM_GEOM(3, 3, 1, 1, 0) SBGM_GEOM(3, 3, 2, 1, 0) SBGM_GEOM(1, 1, 2, 0, 0) SBGM_GEOM(1, 1, 1, 0, 0)
    SBGM_GEOM(8, 8, 1, 4, 0)                       // data-gradient of the 8x8/s2/p3 stem convolution (zero-dilated form)
    SBGM_GEOM(5, 5, 1, 2, 0)                       // ... and its phase-decomposed form (backward.hip: dgrad_phase_weight_kernel)
#undef SBGM_GEOM
    SBGM_CHECK(rc == 0, "conv: no kernel for k=%dx%d s=%d p=%d cs=%d tile=%dx%d ws=%d", g.kh, g.kw, g.stride, g.pad, p.Cs,
               cfg.fco, cfg.fpx, cfg.ws);
    SBGM_LAUNCH_CHECK();
    if (real_splits > 1) {
        p.out = final_out;
        const size_t n4 = (size_t)p.M * p.Cout / 4;
        const int blocks = (int)std::min<size_t>((n4 + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_reduce_epilogue, dim3(blocks), dim3(256), 0, st, p, partial_ws, real_splits);
        SBGM_LAUNCH_CHECK();
    }
    return 0;
}
