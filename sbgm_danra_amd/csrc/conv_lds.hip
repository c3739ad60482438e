// 3x3 / stride 1 / pad 1 convolution with LDS-staged input halo patch and weight slab (fp32 MFMA 16x16x4).
//
// The wave-level kernel (conv_igemm.hip) re-fetches every input pixel 9x (once per tap) and every weight fragment once
// per wave; measurements show it is bound by that operand traffic, not by the matrix pipe.  Here a workgroup owns a
// (4*FPX) x 16 pixel tile of ONE image and 16*FCO output channels.  Per 16-input-channel stage it loads, cooperatively
// and exactly once,
//     the halo patch   [(4*FPX+2) x 18 pixels][16 ch]      (NHWC quads, hardware bounds check = zero padding)
//     the weight slab  [9 taps][16*FCO channels][16 ch]    (1 KiB-contiguous fragments of the packed weights)
// into LDS, then every wave sweeps the 9 taps out of LDS: A fragments (weights) and B fragments (pixels of a tile row
// shifted by the tap) are 16-byte ds_read_b128 per lane, 4*FCO*FPX MFMAs per tap.  Global traffic per output drops ~4.5x.
// Occupancy (2-3 workgroups per CU) overlaps one workgroup's staging with another's MFMAs; one barrier pair per stage.
// With WINO the 3 kw taps of a filter row are replaced by the 4 products of Winograd F(2,3) (see conv_wino.hip): the
// patch is read as 4 input columns per output PAIR, transformed in registers, and the slab holds U = G g (12 "taps").
//
// Input modes (template IN) — the normalisation / resampling passes that used to run between two convolutions of a decoder
// block now happen while the patch is staged (reference score_unet.py:583-615):
//   IN = 1  affine on load: the patch holds x*scale[b,c] + shift[b,c] (GroupNorm of the producer's raw output, statistics
//           finalised into `in_affine` by gn_finalize_kernel); out-of-image pixels stay 0 (zero padding of the NORMALISED map)
//   IN = 2  bilinear x2 on load (nn.Upsample, align_corners=False): x is the LOW-resolution map [B][H/2][W/2][Cs].  Per stage
//           the (TH/2+2) x 10 low-res pixels under the patch are loaded (edge-clamped), optionally transformed
//           act(x*scale + shift + skip) (the pending GroupNorm + skip + time bias + activation of the previous block) and parked
//           in a small LDS region.  A second staging step turns them into the Winograd-transformed input V = B^T d directly: rows are
//           blended with the 0.25 / 0.75 taps and the column taps are folded into B^T — the 4 columns d0..d3 an output pair needs
//           are combinations of just 3 low-res neighbours,
//             d0 = .75 a + .25 b   d1 = .25 a + .75 b   d2 = .75 b + .25 c   d3 = .25 b + .75 c     (a, b, c = x[m-1], x[m], x[m+1])
//           so the 18-wide upsampled patch is never built and the sweep reads ready-made B fragments.  The upsampled tensor
//           (134 MB at the final block of a B=32, 128x128 evaluation) is never written or re-read.
#include "common.h"
#include "kernels.h"
#include "conv_common.h"

namespace {

constexpr int TW = 16;          // tile width = one MFMA fragment of pixels
constexpr int PWID = TW + 2;    // patch width (loaded columns)

// LDS layout.  A pixel (or an output channel of the weight slab) owns 4 consecutive 16-byte slots, one per 4-channel quad.
// ds_read_b128 is served in four fixed 16-lane groups that mix two k-quads and 8 + 8 fragment rows (MI355X_MICROARCH.md
// §LDS); with the plain order quad = kq the 16 lanes of a group fall on 4-8 distinct slots of the 256-byte bank row
// (2-way conflicts for the direct reads, 4-way for the Winograd column pairs; SQ_LDS_BANK_CONFLICT was 46 % of the LDS
// cycles).  Rotating the quad by (index >> 1), plus an odd patch stride for the Winograd variant, makes every group hit 16
// distinct slots (exhaustive check over all taps / column offsets).
__device__ __forceinline__ int swz(int idx, int quad) { return idx * 4 + ((quad + (idx >> 1)) & 3); }

// DB: two LDS stage buffers -> one barrier per stage instead of two, and a stage's LDS stores overlap the other waves' MFMAs.
template <int FCO, int FPX, bool WINO, bool DB, int IN>
__global__ __launch_bounds__(256) void conv3x3_lds_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TH = (WINO ? 8 : 4) * FPX;    // tile rows: 4 waves x FPX rows (Winograd: 2*FPX rows per wave)
    constexpr int PH = TH + 2;
    constexpr int NCO = 16 * FCO;
    constexpr int NTAP = WINO ? 12 : 9;
    constexpr int WQ = NTAP * NCO * 4;          // weight quads (16 B) per stage
    constexpr int PWS = WINO ? PWID + 1 : PWID; // patch row stride in LDS (odd for the Winograd column pairs)
    constexpr int PQ = PH * PWID * 4;           // patch quads loaded per stage
    constexpr int STAGE_QUADS = WQ + (IN == 2 ? PH * 32 : PH * PWS) * 4;   // IN == 2: V patch [PH][4 xi][8 pairs][4 quads]
    constexpr int LH = TH / 2 + 2, LW = TW / 2 + 2;             // IN == 2: low-resolution pixels under the patch
    constexpr int LQ = LH * LW * 4;
    f32x4* wl = reinterpret_cast<f32x4*>(smem_raw);            // [tap][co][4 quads]
    f32x4* pt = wl + WQ;                                        // [py][px (stride PWS)][4 quads, rotated]
    f32x4* const lr0 = reinterpret_cast<f32x4*>(smem_raw) + (DB ? 2 : 1) * STAGE_QUADS;   // IN == 2: [DB ? 2 : 1][LH][LW][4 quads]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;

    // block -> (image, tile row, tile col, co tile), co tile fastest: the output-channel slices of one pixel tile run back to
    // back on one XCD (XCD-contiguous ids), so the halo patch is fetched into that L2 once and neighbouring tiles share halos
    int t = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int tiles_x = p.W / TW, tiles_y = (p.H + TH - 1) / TH, n_co = p.Cout / NCO;
    const int co_tile = t % n_co; t /= n_co;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int co0 = co_tile * NCO, x0 = tx * TW, y0 = ty * TH;

    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.wp, p.w_bytes);
    const int CB = p.cb_per_tap;

    // WINO: a wave's FPX rows are processed as pairs: fragment column r16 = pair index 0..7 of row (2 rows per fragment)
    f32x4 acc[WINO ? 4 : 1][FCO][FPX];
#pragma unroll
    for (int xi = 0; xi < (WINO ? 4 : 1); ++xi)
#pragma unroll
        for (int i = 0; i < FCO; ++i)
#pragma unroll
            for (int j = 0; j < FPX; ++j) acc[xi][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Register-staged double buffering: the global loads of stage cb+1 are issued before the MFMAs of stage cb and land in
    // registers while the matrix pipe works; they are written to LDS after the barrier that retires stage cb's reads.
    constexpr int WPT = (WQ + 255) / 256;       // weight quads per thread per stage
    constexpr int PPT = (PQ + 255) / 256;       // patch quads per thread per stage
    constexpr int LPT = (LQ + 255) / 256;       // IN == 2: low-res quads per thread per stage
    const int hl = p.H >> 1, wlo = p.W >> 1;    // IN == 2: low-res extent
    f32x4 rw[WPT], rp[IN == 2 ? LPT : PPT], rs[IN == 2 ? LPT : 1], sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    // Loop-invariant parts of the stage addresses, computed once and explicitly (a stage then costs one add per load).
    // Weight slab: 256 is a multiple of the NCO*4 quads of a tap, so thread tid's u-th quad is tap u*RW + tl, row quad `rem`;
    // the u-dependent part of its offset is wavefront-uniform (scalar ALU), the per-thread part is one register.
    constexpr uint32_t OOB = 0x80000000u;       // any offset >= 2^31 fails the buffer bounds check and reads 0 (stays so after + cb*64)
    constexpr int RW = 256 / (NCO * 4);
    const int tl = tid / (NCO * 4), rem = tid - tl * (NCO * 4);
    const uint32_t wlane = (uint32_t)((WINO ? tl * p.Cout : tl * CB * p.Cout) * 16 + rem * 4) * 4u;
    const uint32_t wlane_last = (WPT - 1) * RW + tl < NTAP ? wlane : OOB;          // the slab may end inside the last 256-quad round
    uint32_t poff[IN == 2 ? LPT : PPT];
    if (IN != 2) {
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int q = tid + 256 * u;
            const int quad = q & 3, pix = q >> 2;
            const int py = pix / PWID, px = pix - py * PWID;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool ok = (q < PQ) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            poff[u] = ok ? (uint32_t)(((b * p.H + iy) * p.W + ix) * p.Cs + quad * 4) * 4u : OOB;
        }
    } else {
#pragma unroll
        for (int u = 0; u < LPT; ++u) {                      // low-res pixel (row r, col c) of the L region, edge-clamped
            const int q = tid + 256 * u;
            const int quad = q & 3, pix = q >> 2;
            const int r = pix / LW, c = pix - r * LW;
            const int ly = min(max((y0 >> 1) - 1 + r, 0), hl - 1), lx = min(max((x0 >> 1) - 1 + c, 0), wlo - 1);
            poff[u] = q < LQ ? (uint32_t)(((b * hl + ly) * wlo + lx) * p.Cs + quad * 4) * 4u : OOB;
        }
    }
    const float* const aff0 = IN != 0 && p.in_affine != nullptr ? p.in_affine + (((size_t)b * (p.Cs >> 2) + (tid & 3)) * 2) * 4 : nullptr;
    const __amdgpu_buffer_rsrc_t skr = make_rsrc(IN == 2 && p.in_skip != nullptr ? p.in_skip : p.x, p.x_bytes);
    auto stage_load = [&](int cb) {
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int t0 = u * RW;                           // tap of this round for tl = 0
            uint32_t su;                                     // wavefront-uniform
            if (!WINO) su = (uint32_t)(((t0 * CB + cb) * p.Cout + co0) * 16) * 4u;                                   // [tap][cb][Cout][16]
            else su = (uint32_t)(((((t0 >> 2) * CB + cb) * 4 + (t0 & 3)) * p.Cout + co0) * 16) * 4u;               // [kh][cb][xi][Cout][16]
            rw[u] = buf_load4(wr, (u == WPT - 1 ? wlane_last : wlane) + su);
        }
        if (IN != 0 && p.in_affine != nullptr) {             // a thread's quads all share (tid & 3): one scale / shift pair per stage
            const float* ap = aff0 + cb * 32;
            sc = *reinterpret_cast<const f32x4*>(ap);
            sh = *reinterpret_cast<const f32x4*>(ap + 4);
        }
        const uint32_t cbo = (uint32_t)cb * 64u;
#pragma unroll
        for (int u = 0; u < (IN == 2 ? LPT : PPT); ++u) {
            rp[u] = buf_load4(xr, poff[u] + cbo);
            if (IN == 2 && p.in_skip != nullptr) rs[u] = buf_load4(skr, poff[u] + cbo);
        }
    };
    f32x4* const wl0 = wl;
    f32x4* const pt0 = pt;
    auto stage_store_w = [&](int buf) {
        f32x4* wd = wl0 + buf * STAGE_QUADS;
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int q = tid + 256 * u;                       // [tap][co][quad]: rotate the quad by the fragment row (co & 15) >> 1
            if (q < WQ) wd[(q & ~3) + (((q & 3) + (((q >> 2) & 15) >> 1)) & 3)] = rw[u];
        }
    };
    auto stage_store_p = [&](int buf) {                        // IN != 2: the patch straight from the load registers
        f32x4* pd = pt0 + buf * STAGE_QUADS;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int q = tid + 256 * u;
            const int pix = q >> 2, py = pix / PWID, px = pix - py * PWID;
            f32x4 v = rp[IN == 2 ? 0 : u];
            if (IN == 1) {
                const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                v = ok ? v * sc + sh : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (q < PQ) pd[py * PWS * 4 + swz(px, q & 3)] = v;
        }
    };
    auto stage_store_l = [&](int lbuf) {                       // IN == 2: transformed low-res quads -> L region
        f32x4* ld = lr0 + lbuf * LQ;
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int q = tid + 256 * u;
            f32x4 v = rp[u];
            if (p.in_affine != nullptr) v = v * sc + sh;
            if (p.in_skip != nullptr) v += rs[u];
            if (p.in_act == SBGM_ACT_SILU) {                 // hardware exp2 / rcp (1 ulp each): the exact expf + division cost 7 % of the
#pragma unroll                                               // launch, recomputed for every halo; the budget is 1e-4
                for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[e]));
            } else if (p.in_act != SBGM_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sbgm_act(v[e], p.in_act);
            }
            if (q < LQ) ld[q] = v;
        }
    };
    // IN == 2: the staged "patch" holds the Winograd-transformed input V itself, [row][xi][pair][quad]: every (row, pair) is
    // transformed ONCE here instead of once per kh in the sweep, whose B fragments become 4 plain ds_read_b128 with no arithmetic.
    // Quad rotation (quad + 2*(row & 1)) & 3 with `pair` the fastest index: the 16 lanes of a ds_read_b128 group (4 consecutive
    // pairs x 2 quads x 2 row parities) hit 16 distinct slots.
    constexpr int PQ2 = PH * 8 * 4;                            // (row, pair, quad) items per stage
    constexpr int PPT2 = (PQ2 + 255) / 256;
    auto expand = [&](int lbuf, int buf) {
        static_assert(IN != 2 || WINO, "upsample-on-load is built on the Winograd sweep");
        const f32x4* ls = lr0 + lbuf * LQ;
        f32x4* pd = pt0 + buf * STAGE_QUADS;
#pragma unroll
        for (int u = 0; u < PPT2; ++u) {
            const int q = tid + 256 * u;
            const int quad = q & 3, pair = (q >> 2) & 7, py = q >> 5;
            const int iy = y0 - 1 + py;
            // output row iy blends L rows a = py >> 1 and a + 1 (y0 is even); weights as PyTorch: even row 0.25 / 0.75 (row 0: 0 / 1),
            // odd row 0.75 / 0.25; rows outside the image are the convolution's zero padding
            const int ra = py >> 1;
            const float wya = (py & 1) ? (iy == 0 ? 0.f : 0.25f) : 0.75f, wyb = 1.f - wya;
            if (q < PQ2) {
                const f32x4* la = ls + (ra * LW + pair) * 4 + quad;      // low-res neighbours a, b, c of this output pair (columns pair .. pair+2)
                const f32x4* lb = la + LW * 4;
                f32x4 xa = wya * la[0] + wyb * lb[0], xb = wya * la[4] + wyb * lb[4], xc = wya * la[8] + wyb * lb[8];
                if ((unsigned)iy >= (unsigned)p.H) xa = xb = xc = f32x4{0.f, 0.f, 0.f, 0.f};
                // d0..d3 (header comment) substituted into V = B^T d; at the image's left / right edge d0 / d3 is the convolution's zero
                // padding, which only changes the coefficients of v0 / v3
                const bool zl = x0 == 0 && pair == 0, zr = x0 + TW == p.W && pair == 7;
                const float a0 = zl ? 0.f : 0.75f, b0 = zl ? -0.75f : -0.5f, b3 = zr ? 0.75f : 0.5f, c3 = zr ? 0.f : -0.75f;
                f32x4* o = pd + ((py * 4) * 8 + pair) * 4 + ((quad + 2 * (py & 1)) & 3);
                o[0] = a0 * xa + b0 * xb - 0.25f * xc;
                o[32] = 0.25f * (xa + xc) + 1.5f * xb;
                o[64] = 0.25f * (xc - xa);
                o[96] = 0.25f * xa + b3 * xb + c3 * xc;
            }
        }
    };
    stage_load(0);
    stage_store_w(0);
    if (IN != 2) {
        stage_store_p(0);
        if (DB && CB > 1) stage_load(1);
    } else {
        stage_store_l(0);
        __syncthreads();
        expand(0, 0);
        if (DB && CB > 1) { stage_load(1); stage_store_l(1); }
    }
    for (int cb = 0; cb < CB; ++cb) {
        if (!DB) {
            if (cb + 1 < CB) stage_load(cb + 1);
            __syncthreads();
        } else {
            __syncthreads();                     // stage cb is visible; every wave has finished stage cb-1 (the other buffer)
            if (cb + 1 < CB) {
                stage_store_w((cb + 1) & 1);
                if (IN != 2) stage_store_p((cb + 1) & 1);
                else expand((cb + 1) & 1, (cb + 1) & 1);     // L[(cb+1)&1] was parked before the barrier
                if (cb + 2 < CB) stage_load(cb + 2);
            }
            wl = wl0 + (cb & 1) * STAGE_QUADS;
            pt = pt0 + (cb & 1) * STAGE_QUADS;
        }

        // ---- sweep the taps out of LDS ------------------------------------------------------------------------------------
        if (!WINO) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    f32x4 a[FCO], bb[FPX];
#pragma unroll
                    for (int i = 0; i < FCO; ++i) a[i] = wl[((kh * 3 + kw) * NCO + 16 * i) * 4 + swz(r16, kq)];
#pragma unroll
                    for (int j = 0; j < FPX; ++j) bb[j] = pt[(wave * FPX + j + kh) * PWS * 4 + swz(r16 + kw, kq)];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < FCO; ++i)
#pragma unroll
                            for (int j = 0; j < FPX; ++j)
                                acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][k], bb[j][k], acc[0][i][j], 0, 0, 0);
                }
        } else {
            // fragment j of a wave covers 2 tile rows x 8 pairs: lane r16 -> (row = r16 >> 3, pair = r16 & 7)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                f32x4 v[4][FPX];
#pragma unroll
                for (int j = 0; j < FPX; ++j) {
                    const int prow = wave * FPX * 2 + 2 * j + (r16 >> 3) + kh;     // patch row
                    if (IN != 2) {
                        const int pcol = 2 * (r16 & 7);                               // patch col of d0 (= ox0 - 1 - (x0 - 1))
                        const f32x4* src = pt + prow * PWS * 4;
                        const f32x4 d0 = src[swz(pcol, kq)], d1 = src[swz(pcol + 1, kq)], d2 = src[swz(pcol + 2, kq)],
                                    d3 = src[swz(pcol + 3, kq)];
                        v[0][j] = d0 - d2; v[1][j] = d1 + d2; v[2][j] = d2 - d1; v[3][j] = d1 - d3;
                    } else {
                        const f32x4* src = pt + ((prow * 4) * 8 + (r16 & 7)) * 4 + ((kq + 2 * (prow & 1)) & 3);
                        v[0][j] = src[0]; v[1][j] = src[32]; v[2][j] = src[64]; v[3][j] = src[96];
                    }
                }
                if (FCO * FPX == 1) {
                    // one accumulator per xi: sweep k outermost so consecutive MFMAs go to the 4 DIFFERENT xi accumulators (a lone
                    // accumulator chain pays the 40-cycle dependent latency instead of the 32-cycle issue rate)
                    f32x4 a4[4];
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) a4[xi] = wl[((kh * 4 + xi) * NCO) * 4 + swz(r16, kq)];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int xi = 0; xi < 4; ++xi)
                            acc[xi][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[xi][k], v[xi][0][k], acc[xi][0][0], 0, 0, 0);
                } else {
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) {
                        f32x4 a[FCO];
#pragma unroll
                        for (int i = 0; i < FCO; ++i) a[i] = wl[((kh * 4 + xi) * NCO + 16 * i) * 4 + swz(r16, kq)];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int i = 0; i < FCO; ++i)
#pragma unroll
                                for (int j = 0; j < FPX; ++j)
                                    acc[xi][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][k], v[xi][j][k], acc[xi][i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (!DB) {
            if (IN == 2 && cb + 1 < CB) stage_store_l(0);    // the L region was last read before this stage's first barrier
            __syncthreads();                     // every wave is done reading this stage
            if (cb + 1 < CB) {
                stage_store_w(0);
                if (IN != 2) stage_store_p(0);
                else expand(0, 0);
            }
        } else if (IN == 2 && cb + 2 < CB) {
            stage_store_l(cb & 1);               // stage cb+2's low-res quads; L[cb & 1] was expanded before this stage's barrier
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------------
    const int hw = p.H * p.W;
    const bool want_stats = p.gn_stats != nullptr;          // uniform
    f32x4 gs[FCO], gs2[FCO];                                // this lane's sum / sum of squares of its outputs, per channel
#pragma unroll
    for (int i = 0; i < FCO; ++i) { gs[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gs2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (!WINO) {
#pragma unroll
        for (int j = 0; j < FPX; ++j) {
            const int oy = y0 + wave * FPX + j, ox = x0 + r16;
            const bool ok = oy < p.H;                       // W is a multiple of 16
            const int m = (b * p.H + (ok ? oy : 0)) * p.W + ox;
            f32x4 y[FCO];
#pragma unroll
            for (int i = 0; i < FCO; ++i) y[i] = conv_epilogue(acc[0][i][j], p, co0 + 16 * i + 4 * kq, (size_t)m, b);
            if (p.proj_w != nullptr) {
                const float* wlp = p.proj_w + co0 + 4 * kq;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    float s0 = 0.f;
#pragma unroll
                    for (int i = 0; i < FCO; ++i) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wlp + tap * p.Cout + 16 * i);
#pragma unroll
                        for (int e = 0; e < 4; ++e) s0 = fmaf(y[i][e], w4[e], s0);
                    }
                    s0 += __shfl_xor(s0, 16, 64);
                    s0 += __shfl_xor(s0, 32, 64);
                    if (ok && kq == (tap & 3)) p.proj_out[(size_t)tap * p.M + m] = s0;
                }
            } else if (ok) {
#pragma unroll
                for (int i = 0; i < FCO; ++i) {
                    *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.Cout + co0 + 16 * i + 4 * kq) = y[i];
                    if (want_stats) { gs[i] += y[i]; gs2[i] += y[i] * y[i]; }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < FPX; ++j) {
            const int oy = y0 + wave * FPX * 2 + 2 * j + (r16 >> 3), ox = x0 + 2 * (r16 & 7);
            const bool ok = oy < p.H;
            const int m = (b * p.H + (ok ? oy : 0)) * p.W + ox;
            f32x4 y0v[FCO], y1v[FCO];
#pragma unroll
            for (int i = 0; i < FCO; ++i) {
                const int co = co0 + 16 * i + 4 * kq;
                y0v[i] = conv_epilogue(acc[0][i][j] + acc[1][i][j] + acc[2][i][j], p, co, (size_t)m, b);
                y1v[i] = conv_epilogue(acc[1][i][j] - acc[2][i][j] - acc[3][i][j], p, co, (size_t)m + 1, b);
            }
            if (p.proj_w != nullptr) {
                const float* wlp = p.proj_w + co0 + 4 * kq;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int i = 0; i < FCO; ++i) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wlp + tap * p.Cout + 16 * i);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s0 = fmaf(y0v[i][e], w4[e], s0); s1 = fmaf(y1v[i][e], w4[e], s1); }
                    }
                    s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
                    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                    if (ok && kq == (tap & 3)) {
                        p.proj_out[(size_t)tap * p.M + m] = s0;
                        p.proj_out[(size_t)tap * p.M + m + 1] = s1;
                    }
                }
            } else if (ok) {
#pragma unroll
                for (int i = 0; i < FCO; ++i) {
                    const int co = co0 + 16 * i + 4 * kq;
                    *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.Cout + co) = y0v[i];
                    *reinterpret_cast<f32x4*>(p.out + ((size_t)m + 1) * p.Cout + co) = y1v[i];
                    if (want_stats) { gs[i] += y0v[i] + y1v[i]; gs2[i] += y0v[i] * y0v[i] + y1v[i] * y1v[i]; }
                }
            }
        }
    }
    (void)hw;
    if (want_stats) {
        // GroupNorm statistics of this tile, deterministic: pixel lanes by shuffles, the 4 waves through LDS (fixed order), the
        // channels of a group in fp64; one plain store per (group, tile) in the layout groupnorm_apply_kernel reduces.
#pragma unroll
        for (int i = 0; i < FCO; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    gs[i][e] += __shfl_xor(gs[i][e], o, 64);
                    gs2[i][e] += __shfl_xor(gs2[i][e], o, 64);
                }
        __syncthreads();                                   // the stage buffers are free now
        float* red = reinterpret_cast<float*>(smem_raw);    // [wave][NCO][2]
        if (r16 == 0) {
#pragma unroll
            for (int i = 0; i < FCO; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    red[((wave * NCO) + 16 * i + 4 * kq + e) * 2] = gs[i][e];
                    red[((wave * NCO) + 16 * i + 4 * kq + e) * 2 + 1] = gs2[i][e];
                }
        }
        __syncthreads();
        const int G = p.gn_groups, cpg = p.Cout / G;
        const int sub = cpg > NCO ? cpg / NCO : 1;          // tiles' channel slices per group
        const int ngrp = cpg > NCO ? 1 : NCO / cpg;         // whole groups inside this slice
        const int span = cpg > NCO ? NCO : cpg;             // channels of one group inside this slice
        if (tid < ngrp) {
            double a = 0.0, a2 = 0.0;
            for (int c = tid * span; c < (tid + 1) * span; ++c)
                for (int w = 0; w < 4; ++w) {
                    a += (double)red[(w * NCO + c) * 2];
                    a2 += (double)red[(w * NCO + c) * 2 + 1];
                }
            const int chunks = tiles_x * tiles_y * sub;
            const int chunk = (ty * tiles_x + tx) * sub + (co_tile % sub);
            const int g = co0 / cpg + tid;
            double* o = p.gn_stats + (((size_t)b * chunks + chunk) * G + g) * 2;
            o[0] = a;
            o[1] = a2;
        }
    }
}

}  // namespace

int sbgm_conv_lds_gn_chunks(const ConvParams& p, const ConvTile& cfg) {
    if (!cfg.lds || p.gn_groups <= 0 || p.Cout % p.gn_groups || p.proj_w) return 0;
    const int nco = 16 * cfg.fco, cpg = p.Cout / p.gn_groups;
    if (cpg > nco ? cpg % nco != 0 : nco % cpg != 0) return 0;
    const int TH = 4 * (cfg.wino ? 2 * cfg.fpx : cfg.fpx);
    const int chunks = (p.W / 16) * ((p.H + TH - 1) / TH) * (cpg > nco ? cpg / nco : 1);
    return chunks <= 64 ? chunks : 0;
}

// dynamic LDS of one workgroup: the stage buffer(s) plus, with upsample-on-load, the low-resolution parking region(s)
size_t sbgm_conv_lds_bytes(const ConvTile& cfg, int in_mode) {
    const int TH = 4 * (cfg.wino ? 2 * cfg.fpx : cfg.fpx);
    const int nbuf = cfg.lds == 2 ? 2 : 1;
    const size_t patch = in_mode == 2 ? (size_t)(TH + 2) * 32 : (size_t)(TH + 2) * (cfg.wino ? 19 : 18);
    size_t quads = ((size_t)(cfg.wino ? 12 : 9) * 16 * cfg.fco * 4 + patch * 4) * nbuf;
    if (in_mode == 2) quads += (size_t)(TH / 2 + 2) * 10 * 4 * nbuf;
    return quads * 16;
}

// cfg: fco in {1,2,4}; fpx = tile rows per wave (direct: 1,2,4 -> tile 4/8/16 rows; Winograd: rows per wave = 2*fpx);
// cfg.wino selects the Winograd slab (p.wp must then be the Winograd pack).  W must be a multiple of 16.
int sbgm_launch_conv_lds(ConvParams p, const ConvTile& cfg, hipStream_t st) {
    SBGM_CHECK(p.Cs % 16 == 0 && p.W % 16 == 0, "conv_lds: needs Cin padded to 16 and W %% 16 == 0 (Cs=%d W=%d)", p.Cs, p.W);
    SBGM_CHECK(p.Cout % (16 * cfg.fco) == 0, "conv_lds: Cout=%d not a multiple of the %d-channel tile", p.Cout, 16 * cfg.fco);
    SBGM_CHECK(p.act == SBGM_ACT_NONE || p.act == SBGM_ACT_RELU || p.act == SBGM_ACT_GELU, "conv_lds: act=%d does not fuse", p.act);
    SBGM_CHECK((size_t)p.B * p.H * p.W * p.Cs * 4 < (1ull << 31), "conv_lds: input tensor exceeds 2 GiB buffer window");
    SBGM_CHECK(p.proj_w == nullptr || (p.Cout == 16 * cfg.fco && p.proj_out != nullptr), "conv_lds: tap projection needs one co tile");
    SBGM_CHECK(p.in_mode >= 0 && p.in_mode <= 2, "conv_lds: in_mode=%d", p.in_mode);
    SBGM_CHECK(p.in_mode == 0 || cfg.wino, "conv_lds: the fused input modes are instantiated for the Winograd tiles only");
    SBGM_CHECK(p.in_mode != 1 || p.in_affine != nullptr, "conv_lds: in_mode 1 needs in_affine");
    SBGM_CHECK(p.in_mode == 2 || (p.in_skip == nullptr && p.in_act == SBGM_ACT_NONE), "conv_lds: skip / activation on load need in_mode 2");
    SBGM_CHECK(p.in_mode != 2 || (p.H % 2 == 0 && p.W % 2 == 0), "conv_lds: upsample-on-load needs even H, W");
    if (p.gn_stats && sbgm_conv_lds_gn_chunks(p, cfg) == 0) p.gn_stats = nullptr;      // this tile cannot produce them
    p.OH = p.H; p.OW = p.W;
    p.M = p.B * p.H * p.W;
    p.cb_per_tap = p.Cs / 16;
    p.nsteps = 9 * p.cb_per_tap;
    p.x_bytes = (uint32_t)((size_t)p.B * p.H * p.W * p.Cs * 4 / (p.in_mode == 2 ? 4 : 1));
    p.w_bytes = (uint32_t)((cfg.wino ? sbgm_wino_packed_floats(p.Cout, p.Cs) : (size_t)9 * p.cb_per_tap * p.Cout * 16) * 4);
    const int rows_per_wave = cfg.wino ? 2 * cfg.fpx : cfg.fpx;
    const int TH = 4 * rows_per_wave;
    const int tiles = (p.W / 16) * ((p.H + TH - 1) / TH) * p.B * (p.Cout / (16 * cfg.fco));
    const int ntap = cfg.wino ? 12 : 9;
    const bool db = cfg.lds == 2;                       // double-buffered stages
    const size_t lds = sbgm_conv_lds_bytes(cfg, p.in_mode);
    SBGM_CHECK(lds <= 160 * 1024, "conv_lds: tile needs %zu bytes of LDS", lds);
    (void)ntap;
    int rc = 1;
#define SBGM_L3(FC, FP, WN, DBV, INV)                                                                        \
    if (cfg.fco == FC && cfg.fpx == FP && (cfg.wino != 0) == WN && db == DBV && p.in_mode == INV) {           \
        if (lds > 64 * 1024)                                                                                  \
            SBGM_HIP(hipFuncSetAttribute((const void*)conv3x3_lds_kernel<FC, FP, WN, DBV, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((conv3x3_lds_kernel<FC, FP, WN, DBV, INV>), dim3(tiles), dim3(256), lds, st, p); \
        rc = 0;                                                                                              \
    }
#define SBGM_L(FC, FP, WN) SBGM_L3(FC, FP, WN, false, 0) SBGM_L3(FC, FP, WN, true, 0)
#define SBGM_LF(FC, FP) SBGM_L3(FC, FP, true, false, 1) SBGM_L3(FC, FP, true, true, 1) SBGM_L3(FC, FP, true, false, 2) SBGM_L3(FC, FP, true, true, 2)
    SBGM_L(4, 1, false) SBGM_L(4, 2, false) SBGM_L(4, 4, false) SBGM_L(2, 2, false) SBGM_L(2, 4, false) SBGM_L(2, 1, false)
    SBGM_L(4, 1, true) SBGM_L(4, 2, true) SBGM_L(2, 1, true) SBGM_L(2, 2, true) SBGM_L(1, 1, true) SBGM_L(1, 2, true)
    SBGM_LF(4, 1) SBGM_LF(4, 2) SBGM_LF(2, 1) SBGM_LF(2, 2) SBGM_LF(1, 1) SBGM_LF(1, 2)
#undef SBGM_LF
#undef SBGM_L
#undef SBGM_L3
    SBGM_CHECK(rc == 0, "conv_lds: no kernel for tile fco=%d fpx=%d wino=%d in_mode=%d", cfg.fco, cfg.fpx, cfg.wino, p.in_mode);
    SBGM_LAUNCH_CHECK();
    return 0;
}
