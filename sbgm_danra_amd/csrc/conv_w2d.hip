// 3x3 / stride 1 / pad 1 convolution as a 2-D Winograd F(2x2, 3x3) on the fp32 MFMA pipe, LDS-staged (gfx950).
//
// conv_lds.hip transforms along image rows only: 12 products per output pair = 6 MFMA taps per output where the direct form
// needs 9.  Its matrix pipe is saturated inside the stage loop ("SQ stall split"), so the lever that is left is
// fewer MFMAs: the 2-D form needs 16 products per 2x2 output block = 4 per output,
//     V = B^T d B   (4x4 input tile d of a block)      U = G g G^T   (3x3 filter g, packed once per upload)
//     M[xi][eta] = sum_ci U[xi][eta][co][ci] * V[xi][eta][ci]         Y = A^T M A   (2x2 outputs)
// with the same benign constants as F(2,3) (B: 0, +-1; G: 1, 1/2; A: 0, +-1), so the result differs from the direct fp32
// convolution by ordinary rounding (measured ~1e-6 relative).
//
// Mapping.  A workgroup (4 waves) owns a 16 x 16 pixel tile of ONE image and NCO = 16*FCO output channels.  A wave owns 4 tile
// rows = 16 blocks (2 block rows x 8 block columns): fragment column r16 -> block (r16 >> 3, r16 & 7).  It keeps all 16
// (xi, eta) accumulator sets of its blocks: 16 * FCO f32x4 (128 registers at FCO = 2).  Per 16-input-channel stage the workgroup
// loads, cooperatively and once, the weight slab [16 (xi, eta)][NCO][16 ch] and the 18 x 18 halo patch into LDS; every wave then
// reads its 4 x 4 input quads per block, applies B^T . B in registers and issues 16 * FCO * 4 MFMAs.
//
// Input modes (template IN) as in conv_lds.hip: 1 = per-(sample, channel) affine on load (GroupNorm of the producer),
// 2 = bilinear x2 upsample on load from the LOW-resolution map (optionally act(x*scale + shift + skip) first).  In mode 2 a
// second staging step writes the COLUMN-transformed rows W[row][eta] = (d B)[row][eta] straight from the three low-resolution
// neighbours of a column pair (the interpolation's column taps fold into B, conv_lds.hip header); the sweep then only applies the
// row half V[xi] = B^T W, 16 vector operations per block instead of 32.
//
// Two kernels share the layouts and the arithmetic:
//   conv3x3_w2d_kernel   one tile per workgroup, register-staged slab and patch (single- or double-buffered stages)
//   conv3x3_w2dp_kernel  PERSISTENT workgroups (two per CU, each walking over a contiguous range of tiles so the stage pipeline runs
//                        across tile boundaries), weight slab by LDS-DMA in two rolling halves (no staging registers, no ds_write),
//                        double-buffered halo patch, A fragments and patch rows requested one MFMA group ahead
// The autotuner times both per layer (ConvTile.lds = 1 / 2: first kernel, 3: persistent kernel).  What the measurements say about
// them is in DESIGN.md section 3.1.
//
// LDS layouts (conflict-free for every ds_read_b128 lane group, exhaustive check in tools/lds_bank_check.py):
//   weight slab  [tap][co][4 quads], quad rotated by (co & 15) >> 1                          (as conv_lds.hip)
//   raw patch    [row (stride 74 quads)][px][4 quads], quad rotated by 2 * (px >> 2)          (modes 0, 1)
//   W patch      [row][eta][pair][4 quads], quad rotated by 2 * ((row >> 1) & 1)              (mode 2)
#include "common.h"
#include "kernels.h"
#include "conv_common.h"

namespace {

constexpr int TW = 16, TH = 16;         // tile = 16 x 16 pixels = 8 x 8 blocks of 2 x 2 outputs
constexpr int PH = TH + 2, PWID = TW + 2;
constexpr int SY = PWID * 4 + 2;        // raw patch: row stride in quads (2 spare quads put the two block rows of a fragment on
                                        // different halves of the bank row)

__device__ __forceinline__ int wslot(int row, int quad) { return row * 4 + ((quad + (row >> 1)) & 3); }
__device__ __forceinline__ int pslot(int px, int quad) { return px * 4 + ((quad + 2 * (px >> 2)) & 3); }

// MINW = waves per SIMD the register allocation is held to (2: two workgroups share a CU, at FCO = 2 at the price of a few spills)
template <int FCO, int MINW, bool DB, int IN>
__global__ __launch_bounds__(256, MINW) void conv3x3_w2d_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NCO = 16 * FCO;
    constexpr int WQ = 16 * NCO * 4;                            // weight quads per stage
    constexpr int PQ = PH * PWID * 4;                           // raw patch quads loaded per stage
    constexpr int PREG = IN == 2 ? PH * 4 * 8 * 4 : PH * SY;    // patch region of a stage buffer, quads
    constexpr int STAGE_QUADS = WQ + PREG;
    constexpr int LH = TH / 2 + 2, LW = TW / 2 + 2;             // IN == 2: low-resolution pixels under the patch
    constexpr int LQ = LH * LW * 4;
    f32x4* wl = reinterpret_cast<f32x4*>(smem_raw);            // [tap][co][4 quads]
    f32x4* pt = wl + WQ;
    f32x4* const lr0 = reinterpret_cast<f32x4*>(smem_raw) + (DB ? 2 : 1) * STAGE_QUADS;   // IN == 2: [DB ? 2 : 1][LH][LW][4 quads]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int br = r16 >> 3, bc = r16 & 7;                      // this lane's block inside the wave's 2 x 8 block fragment

    // block -> (image, tile row, tile col, co tile), co tile fastest (conv_lds.hip): XCD-contiguous ids keep the co slices of one
    // pixel tile and neighbouring tiles' halos in one L2
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

    f32x4 acc[16][FCO];
#pragma unroll
    for (int tp = 0; tp < 16; ++tp)
#pragma unroll
        for (int i = 0; i < FCO; ++i) acc[tp][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging (register-staged: the global loads of stage cb+1 fly during the MFMAs of stage cb) -------------------------
    constexpr int WPT = WQ / 256;               // weight quads per thread per stage (4 * FCO)
    constexpr int PPT = (PQ + 255) / 256;
    constexpr int LPT = (LQ + 255) / 256;
    constexpr int RW = 256 / (NCO * 4);         // taps per 256-quad round
    static_assert(16 % RW == 0 && WQ % 256 == 0, "slab rounds");
    const int hl = p.H >> 1, wlo = p.W >> 1;
    f32x4 rw[WPT], rp[IN == 2 ? LPT : PPT], rs[IN == 2 ? LPT : 1], sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    constexpr uint32_t OOB = 0x80000000u;
    const int tl = tid / (NCO * 4), rem = tid - tl * (NCO * 4);
    const uint32_t wlane = (uint32_t)(tl * p.Cout * 16 + rem * 4) * 4u;
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
        for (int u = 0; u < WPT; ++u) {                      // [cb][tap][Cout][16]: the round's first tap is wavefront-uniform
            const uint32_t su = (uint32_t)(((cb * 16 + u * RW) * p.Cout + co0) * 16) * 4u;
            rw[u] = buf_load4(wr, wlane + su);
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
            wd[(q & ~3) + (((q & 3) + (((q >> 2) & 15) >> 1)) & 3)] = rw[u];
        }
    };
    auto stage_store_p = [&](int buf) {                        // IN != 2: the raw patch straight from the load registers
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
            if (q < PQ) pd[py * SY + pslot(px, q & 3)] = v;
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
            if (p.in_act == SBGM_ACT_SILU) {                 // hardware exp2 / rcp (1 ulp each), as conv_lds.hip
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[e]));
            } else if (p.in_act != SBGM_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sbgm_act(v[e], p.in_act);
            }
            if (q < LQ) ld[q] = v;
        }
    };
    // IN == 2: W[row][eta][pair] = column transform of the upsampled row, straight from the low-res neighbours (see conv_lds.hip
    // for the coefficient derivation; identical arithmetic, other LDS rotation: the fragment's block rows are 2 patch rows apart)
    constexpr int PQ2 = PH * 8 * 4;
    constexpr int PPT2 = (PQ2 + 255) / 256;
    auto expand = [&](int lbuf, int buf) {
        const f32x4* ls = lr0 + lbuf * LQ;
        f32x4* pd = pt0 + buf * STAGE_QUADS;
#pragma unroll
        for (int u = 0; u < PPT2; ++u) {
            const int q = tid + 256 * u;
            const int quad = q & 3, pair = (q >> 2) & 7, py = q >> 5;
            const int iy = y0 - 1 + py;
            const int ra = py >> 1;
            const float wya = (py & 1) ? (iy == 0 ? 0.f : 0.25f) : 0.75f, wyb = 1.f - wya;
            if (q < PQ2) {
                const f32x4* la = ls + (ra * LW + pair) * 4 + quad;
                const f32x4* lb = la + LW * 4;
                f32x4 xa = wya * la[0] + wyb * lb[0], xb = wya * la[4] + wyb * lb[4], xc = wya * la[8] + wyb * lb[8];
                if ((unsigned)iy >= (unsigned)p.H) xa = xb = xc = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool zl = x0 == 0 && pair == 0, zr = x0 + TW == p.W && pair == 7;
                const float a0 = zl ? 0.f : 0.75f, b0 = zl ? -0.75f : -0.5f, b3 = zr ? 0.75f : 0.5f, c3 = zr ? 0.f : -0.75f;
                f32x4* o = pd + ((py * 4) * 8 + pair) * 4 + ((quad + 2 * ((py >> 1) & 1)) & 3);
                o[0] = a0 * xa + b0 * xb - 0.25f * xc;
                o[32] = 0.25f * (xa + xc) + 1.5f * xb;
                o[64] = 0.25f * (xc - xa);
                o[96] = 0.25f * xa + b3 * xb + c3 * xc;
            }
        }
    };

    // loop-invariant LDS read offsets of this lane (quads)
    const int aoff = wslot(r16, kq);
    const int r0 = wave * 4 + 2 * br;                         // first patch row of the lane's block
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        coff[c] = IN == 2 ? (c * 8 + bc) * 4 : pslot(2 * bc + c, kq);     // IN == 2: eta = c; the rotation depends on the row

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
                else expand((cb + 1) & 1, (cb + 1) & 1);
                if (cb + 2 < CB) stage_load(cb + 2);
            }
            wl = wl0 + (cb & 1) * STAGE_QUADS;
            pt = pt0 + (cb & 1) * STAGE_QUADS;
        }

        // ---- sweep: per xi the row combination of two patch rows, the column transform, then 4 eta x FCO x 4 MFMAs -----------
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};     // B^T rows: d0 - d2, d1 + d2, d2 - d1, d1 - d3
            const int rra = RA[xi], rrb = RB[xi];
            f32x4 T[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 da, db;
                if (IN == 2) {
                    da = pt[(r0 + rra) * 128 + coff[c] + ((kq + 2 * (((r0 + rra) >> 1) & 1)) & 3)];
                    db = pt[(r0 + rrb) * 128 + coff[c] + ((kq + 2 * (((r0 + rrb) >> 1) & 1)) & 3)];
                } else {
                    da = pt[(r0 + rra) * SY + coff[c]];
                    db = pt[(r0 + rrb) * SY + coff[c]];
                }
                T[c] = xi == 1 ? da + db : da - db;
            }
            f32x4 V[4];
            if (IN == 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) V[c] = T[c];
            } else {
                V[0] = T[0] - T[2]; V[1] = T[1] + T[2]; V[2] = T[2] - T[1]; V[3] = T[1] - T[3];
            }
            if (FCO == 1) {
                // one accumulator per (xi, eta): k outermost so consecutive MFMAs go to 4 different accumulators
                f32x4 a4[4];
#pragma unroll
                for (int eta = 0; eta < 4; ++eta) a4[eta] = wl[((xi * 4 + eta) * NCO) * 4 + aoff];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int eta = 0; eta < 4; ++eta)
                        acc[xi * 4 + eta][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[eta][k], V[eta][k], acc[xi * 4 + eta][0], 0, 0, 0);
            } else {
#pragma unroll
                for (int eta = 0; eta < 4; ++eta) {
                    f32x4 a[FCO];
#pragma unroll
                    for (int i = 0; i < FCO; ++i) a[i] = wl[((xi * 4 + eta) * NCO + 16 * i) * 4 + aoff];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < FCO; ++i)
                            acc[xi * 4 + eta][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][k], V[eta][k], acc[xi * 4 + eta][i], 0, 0, 0);
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
            stage_store_l(cb & 1);
        }
    }

    // ---- epilogue: Y = A^T M A, then the shared convolution epilogue on the block's 4 pixels ---------------------------------
    const bool want_stats = p.gn_stats != nullptr;          // uniform
    f32x4 gs[FCO], gs2[FCO];
#pragma unroll
    for (int i = 0; i < FCO; ++i) { gs[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gs2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int oy = y0 + wave * 4 + 2 * br, ox = x0 + 2 * bc;
    const bool ok = oy < p.H;                               // H is even: the block's second row is inside with the first
    const int m00 = (b * p.H + (ok ? oy : 0)) * p.W + ox;
    f32x4 y[4][FCO];                                        // [2 * row + col][co fragment]
#pragma unroll
    for (int i = 0; i < FCO; ++i) {
        f32x4 P0[4], P1[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            P0[xi] = acc[xi * 4][i] + acc[xi * 4 + 1][i] + acc[xi * 4 + 2][i];
            P1[xi] = acc[xi * 4 + 1][i] - acc[xi * 4 + 2][i] - acc[xi * 4 + 3][i];
        }
        const int co = co0 + 16 * i + 4 * kq;
        y[0][i] = conv_epilogue(P0[0] + P0[1] + P0[2], p, co, (size_t)m00, b);
        y[1][i] = conv_epilogue(P1[0] + P1[1] + P1[2], p, co, (size_t)m00 + 1, b);
        y[2][i] = conv_epilogue(P0[1] - P0[2] - P0[3], p, co, (size_t)m00 + p.W, b);
        y[3][i] = conv_epilogue(P1[1] - P1[2] - P1[3], p, co, (size_t)m00 + p.W + 1, b);
    }
    if (p.proj_w != nullptr) {
        // fused final block: project the NCO channels of each pixel onto the 9 taps of the following Cout = 1 convolution; with
        // several co tiles every tile writes its own partial plane [co_tile][tap][M] (tap_stencil sums them)
        const float* wlp = p.proj_w + co0 + 4 * kq;
        float* po = p.proj_out + (size_t)co_tile * 9 * p.M;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < FCO; ++i) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wlp + tap * p.Cout + 16 * i);
#pragma unroll
                for (int px = 0; px < 4; ++px)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[px] = fmaf(y[px][i][e], w4[e], s[px]);
            }
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                s[px] += __shfl_xor(s[px], 16, 64);
                s[px] += __shfl_xor(s[px], 32, 64);
            }
            if (ok && kq == (tap & 3)) {
                float* o = po + (size_t)tap * p.M + m00;
                o[0] = s[0]; o[1] = s[1]; o[p.W] = s[2]; o[p.W + 1] = s[3];
            }
        }
    } else if (ok) {
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
            float* o = p.out + (size_t)m00 * p.Cout + co0 + 16 * i + 4 * kq;
            *reinterpret_cast<f32x4*>(o) = y[0][i];
            *reinterpret_cast<f32x4*>(o + p.Cout) = y[1][i];
            *reinterpret_cast<f32x4*>(o + (size_t)p.W * p.Cout) = y[2][i];
            *reinterpret_cast<f32x4*>(o + (size_t)(p.W + 1) * p.Cout) = y[3][i];
            if (want_stats) {
                gs[i] += (y[0][i] + y[1][i]) + (y[2][i] + y[3][i]);
                gs2[i] += (y[0][i] * y[0][i] + y[1][i] * y[1][i]) + (y[2][i] * y[2][i] + y[3][i] * y[3][i]);
            }
        }
    }
    if (want_stats) {
        // GroupNorm statistics of this tile, deterministic (conv_lds.hip): lanes by shuffles, waves through LDS, groups in fp64
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
        const int sub = cpg > NCO ? cpg / NCO : 1;
        const int ngrp = cpg > NCO ? 1 : NCO / cpg;
        const int span = cpg > NCO ? NCO : cpg;
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

// ---- persistent form -----------------------------------------------------------------------------------------------------------
// Stage pipeline (one stage = 16 input channels).  Weight slab [16 taps][NCO][16 ch] in LDS, single copy, two halves: H0 = taps of
// xi 0,1 and H1 = taps of xi 2,3.  Halo patch, two copies (mode 2: one copy of the column-transformed rows + the low-res region).
//     E(s-1) | DMA H1(s) ; global loads of patch(s+1) -> registers | sweep xi 0,1 of stage s (reads H0(s), patch(s))
//     M(s)   | DMA H0(s+1)                                          | sweep xi 2,3 of stage s (reads H1(s), patch(s))
//            | patch(s+1) registers -> other patch copy ; last stage of a tile: output transform, epilogue, stores
//     E(s)   | ...
// E and M are workgroup barriers behind an explicit s_waitcnt vmcnt(0) of every wave (an LDS-DMA is a pending LDS write on that
// counter): a slab half is read only after the issuing waves' wait AND a barrier, and is overwritten only after a barrier every
// reader has passed.  Each DMA half has half a sweep to land.  Mode 2 needs a third barrier per stage (E | expand | X | sweep).
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int FCO, int IN, bool PROJ>
__global__ __launch_bounds__(256, 2) void conv3x3_w2dp_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NCO = 16 * FCO;
    constexpr int WQ = 16 * NCO * 4;                            // slab quads
    constexpr int HQ = WQ / 2;                                  // quads of one slab half
    constexpr int PQ = PH * PWID * 4;                           // raw patch quads loaded per stage
    constexpr int PREG = IN == 2 ? PH * 4 * 8 * 4 : PH * SY;    // one patch copy, quads
    constexpr int NPB = IN == 2 ? 1 : 2;                        // patch copies
    constexpr int LH = TH / 2 + 2, LW = TW / 2 + 2;             // IN == 2: low-resolution pixels under the patch
    constexpr int LQ = LH * LW * 4;
    f32x4* const wl = reinterpret_cast<f32x4*>(smem_raw);      // [tap][co][4 quads]
    f32x4* const pt0 = wl + WQ;                                 // patch copies
    f32x4* const lr0 = pt0 + NPB * PREG;                        // IN == 2: L region [LH][LW][4 quads]
    float* const red = reinterpret_cast<float*>(lr0 + (IN == 2 ? LQ : 0));   // GroupNorm statistics scratch [4 waves][NCO][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int br = r16 >> 3, bc = r16 & 7;

    const int tiles_x = p.W / TW, tiles_y = (p.H + TH - 1) / TH, n_co = p.Cout / NCO;
    const int n_tiles = tiles_x * tiles_y * p.B * n_co;
    const int CB = p.cb_per_tap;
    // this workgroup's tile range (XCD-contiguous logical id: one XCD's L2 sees neighbouring tiles and one weight slice)
    const int L = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int t_begin = (int)((long long)L * n_tiles / gridDim.x), t_end = (int)((long long)(L + 1) * n_tiles / gridDim.x);
    if (t_begin >= t_end) return;

    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t skr = make_rsrc(IN == 2 && p.in_skip != nullptr ? p.in_skip : p.x, p.x_bytes);
    const int hl = p.H >> 1, wlo = p.W >> 1;

    // ---- tile coordinates: `c*` = the tile being computed, `n*` = the tile whose stages are being loaded --------------------
    struct Tile { int co_tile, tx, ty, b; };
    auto decode = [&](int t) {
        Tile r;
        r.co_tile = t % n_co; t /= n_co;
        r.tx = t % tiles_x; t /= tiles_x;
        r.ty = t % tiles_y;
        r.b = t / tiles_y;
        return r;
    };
    auto advance = [&](Tile& r) {
        if (++r.co_tile == n_co) { r.co_tile = 0; if (++r.tx == tiles_x) { r.tx = 0; if (++r.ty == tiles_y) { r.ty = 0; ++r.b; } } }
    };
    Tile ct = decode(t_begin), nt = ct;
    int n_tile = t_begin, n_cb = 0;                             // loader position (tile, stage)
    bool n_ok = true;

    // per-thread patch load offsets of the loader's tile
    constexpr int PPT = (PQ + 255) / 256;
    constexpr int LPT = (LQ + 255) / 256;
    constexpr int NRP = IN == 2 ? LPT : PPT;
    constexpr uint32_t OOB = 0x80000000u;
    uint32_t poff[NRP];
    f32x4 rp[NRP], rs[IN == 2 ? LPT : 1], sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    auto set_poff = [&](const Tile& tl) {
        const int x0 = tl.tx * TW, y0 = tl.ty * TH;
        if (IN != 2) {
#pragma unroll
            for (int u = 0; u < PPT; ++u) {
                const int q = tid + 256 * u;
                const int quad = q & 3, pix = q >> 2;
                const int py = pix / PWID, px = pix - py * PWID;
                const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                const bool ok = (q < PQ) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                poff[u] = ok ? (uint32_t)(((tl.b * p.H + iy) * p.W + ix) * p.Cs + quad * 4) * 4u : OOB;
            }
        } else {
#pragma unroll
            for (int u = 0; u < LPT; ++u) {
                const int q = tid + 256 * u;
                const int quad = q & 3, pix = q >> 2;
                const int r = pix / LW, c = pix - r * LW;
                const int ly = min(max((y0 >> 1) - 1 + r, 0), hl - 1), lx = min(max((x0 >> 1) - 1 + c, 0), wlo - 1);
                poff[u] = q < LQ ? (uint32_t)(((tl.b * hl + ly) * wlo + lx) * p.Cs + quad * 4) * 4u : OOB;
            }
        }
    };
    // global loads of the loader's stage: patch quads (or low-res quads) -> registers
    auto patch_load = [&]() {
        if (IN != 0 && p.in_affine != nullptr) {
            const float* ap = p.in_affine + (((size_t)nt.b * (p.Cs >> 2) + (tid & 3)) * 2) * 4 + n_cb * 32;
            sc = *reinterpret_cast<const f32x4*>(ap);
            sh = *reinterpret_cast<const f32x4*>(ap + 4);
        }
        const uint32_t cbo = (uint32_t)n_cb * 64u;
#pragma unroll
        for (int u = 0; u < NRP; ++u) {
            rp[u] = buf_load4(xr, poff[u] + cbo);
            if (IN == 2 && p.in_skip != nullptr) rs[u] = buf_load4(skr, poff[u] + cbo);
        }
    };
    // LDS-DMA of one slab half of the loader's stage.  A wave instruction fills 64 consecutive quads = 16 output channels of one tap;
    // lane -> (co = lane >> 2, slot = lane & 3) and the slot holds source quad (slot - (co >> 1)) & 3 (the read-side rotation).
    constexpr int DPW = HQ / 64 / 4;                            // DMA instructions per wave per half (2 * FCO)
    const uint32_t dma_lane = (uint32_t)(((lane >> 2) * 16 + (((lane & 3) - ((lane >> 2) >> 1)) & 3) * 4) * 4);   // bytes inside a 1 KiB fragment
    auto slab_dma = [&](int half) {
        const char* src = reinterpret_cast<const char*>(p.wp);
#pragma unroll
        for (int u = 0; u < DPW; ++u) {
            const int frag = wave * DPW + u;                    // 16-channel fragment inside the half: [8 taps][FCO]
            const int tap = half * 8 + frag / FCO, cf = frag % FCO;
            const size_t off = ((size_t)((n_cb * 16 + tap) * p.Cout + nt.co_tile * NCO + cf * 16) * 16) * 4;     // wavefront-uniform
            __builtin_amdgcn_global_load_lds((gptr_t)(src + off + dma_lane), (lptr_t)(wl + half * HQ + frag * 64), 16, 0, 0);
        }
    };
    auto patch_store = [&](int buf) {                          // IN != 2: registers -> patch copy `buf`
        f32x4* pd = pt0 + buf * PREG;
        const int x0 = nt.tx * TW, y0 = nt.ty * TH;
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
            if (q < PQ) pd[py * SY + pslot(px, q & 3)] = v;
        }
    };
    auto l_store = [&]() {                                     // IN == 2: transformed low-res quads -> L region
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int q = tid + 256 * u;
            f32x4 v = rp[u];
            if (p.in_affine != nullptr) v = v * sc + sh;
            if (p.in_skip != nullptr) v += rs[u];
            if (p.in_act == SBGM_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[e]));
            } else if (p.in_act != SBGM_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sbgm_act(v[e], p.in_act);
            }
            if (q < LQ) lr0[q] = v;
        }
    };
    constexpr int PQ2 = PH * 8 * 4;
    constexpr int PPT2 = (PQ2 + 255) / 256;
    auto expand = [&](const Tile& tl) {                        // IN == 2: L region -> W patch (column-transformed upsampled rows)
        const int x0 = tl.tx * TW, y0 = tl.ty * TH;
#pragma unroll
        for (int u = 0; u < PPT2; ++u) {
            const int q = tid + 256 * u;
            const int quad = q & 3, pair = (q >> 2) & 7, py = q >> 5;
            const int iy = y0 - 1 + py;
            const int ra = py >> 1;
            const float wya = (py & 1) ? (iy == 0 ? 0.f : 0.25f) : 0.75f, wyb = 1.f - wya;
            if (q < PQ2) {
                const f32x4* la = lr0 + (ra * LW + pair) * 4 + quad;
                const f32x4* lb = la + LW * 4;
                f32x4 xa = wya * la[0] + wyb * lb[0], xb = wya * la[4] + wyb * lb[4], xc = wya * la[8] + wyb * lb[8];
                if ((unsigned)iy >= (unsigned)p.H) xa = xb = xc = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool zl = x0 == 0 && pair == 0, zr = x0 + TW == p.W && pair == 7;
                const float a0 = zl ? 0.f : 0.75f, b0 = zl ? -0.75f : -0.5f, b3 = zr ? 0.75f : 0.5f, c3 = zr ? 0.f : -0.75f;
                f32x4* o = pt0 + ((py * 4) * 8 + pair) * 4 + ((quad + 2 * ((py >> 1) & 1)) & 3);
                o[0] = a0 * xa + b0 * xb - 0.25f * xc;
                o[32] = 0.25f * (xa + xc) + 1.5f * xb;
                o[64] = 0.25f * (xc - xa);
                o[96] = 0.25f * xa + b3 * xb + c3 * xc;
            }
        }
    };
    auto loader_next = [&]() {                                 // loader -> next stage (possibly of the next tile)
        if (++n_cb == CB) {
            n_cb = 0;
            ++n_tile;
            n_ok = n_tile < t_end;
            if (n_ok) { advance(nt); set_poff(nt); }
        }
    };

    // ---- loop-invariant LDS read offsets of this lane (quads) -------------------------------------------------------------
    const int aoff = wslot(r16, kq);
    const int r0 = wave * 4 + 2 * br;
    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = IN == 2 ? (c * 8 + bc) * 4 : pslot(2 * bc + c, kq);

    f32x4 acc[16][FCO];
#pragma unroll
    for (int tp = 0; tp < 16; ++tp)
#pragma unroll
        for (int i = 0; i < FCO; ++i) acc[tp][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stage 0 of the first tile --------------------------------------------------------------------------------
    set_poff(nt);
    slab_dma(0);
    slab_dma(1);
    patch_load();
    if (IN != 2) patch_store(0);
    else l_store();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (IN == 2) { expand(nt); }
    loader_next();                                              // loader now at stage 1
    if (n_ok) patch_load();                                     // registers <- patch(1)
    if (IN == 2) __syncthreads();                               // W patch of stage 0 visible (its DMA halves were drained above)

    int pbuf = 0;                                               // patch copy of the stage being swept (IN != 2)
    for (int tile = t_begin; tile < t_end; ++tile) {
        for (int cb = 0; cb < CB; ++cb) {
            const f32x4* pt = pt0 + (IN == 2 ? 0 : pbuf * PREG);
            auto ldrow = [&](int rr, int c) -> f32x4 {
                if (IN == 2) return pt[(r0 + rr) * 128 + coff[c] + ((kq + 2 * (((r0 + rr) >> 1) & 1)) & 3)];
                return pt[(r0 + rr) * SY + coff[c]];
            };
            constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};     // B^T rows: d0 - d2, d1 + d2, d2 - d1, d1 - d3
            // one half of the sweep: xi = 2*half, 2*half + 1.  The A fragments of the next (xi, eta) group and the patch rows of the
            // next xi are requested one group ahead of the MFMAs that consume them.
            auto sweep_half = [&](int half) {
                f32x4 a_cur[FCO], a_nxt[FCO], V[4], Vn[4], da[4], db[4];
                const int g0 = half * 8;
#pragma unroll
                for (int i = 0; i < FCO; ++i) a_cur[i] = wl[(g0 * NCO + 16 * i) * 4 + aoff];
#pragma unroll
                for (int c = 0; c < 4; ++c) { da[c] = ldrow(RA[2 * half], c); db[c] = ldrow(RB[2 * half], c); }
#pragma unroll
                for (int c = 0; c < 4; ++c) da[c] = da[c] - db[c];         // xi = 0 and xi = 2 both subtract
                if (IN == 2) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) V[c] = da[c];
                } else {
                    V[0] = da[0] - da[2]; V[1] = da[1] + da[2]; V[2] = da[2] - da[1]; V[3] = da[1] - da[3];
                }
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) {
                    const int g = g0 + gg, xi = g >> 2, eta = g & 3;
                    const bool more = gg < 4;                               // a second xi follows inside this half
                    // requests first, pinned above this group's MFMAs (left alone the scheduler sinks them to one MFMA before use)
                    if (gg + 1 < 8) {
#pragma unroll
                        for (int i = 0; i < FCO; ++i) a_nxt[i] = wl[((g + 1) * NCO + 16 * i) * 4 + aoff];
                    }
                    if (eta == 0 && more) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) { da[c] = ldrow(RA[xi + 1], c); db[c] = ldrow(RB[xi + 1], c); }
                    }
                    if (eta == 2 && more) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) da[c] = xi + 1 == 1 ? da[c] + db[c] : da[c] - db[c];
                    }
                    if (eta == 3 && more) {
                        if (IN == 2) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) Vn[c] = da[c];
                        } else {
                            Vn[0] = da[0] - da[2]; Vn[1] = da[1] + da[2]; Vn[2] = da[2] - da[1]; Vn[3] = da[1] - da[3];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < FCO; ++i)
                            acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i][k], V[eta][k], acc[g][i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < FCO; ++i) a_cur[i] = a_nxt[i];
                    if (eta == 3 && more) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) V[c] = Vn[c];
                    }
                }
            };

            sweep_half(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's DMA of H1(s) (and the patch registers) have landed
            __syncthreads();                                    // M: H0 is free, H1 of this stage has landed
            if (n_ok) slab_dma(0);                              // H0 of the next stage
            if (IN == 2) { if (n_ok) l_store(); }               // L region: last read by this stage's expand, before E
            sweep_half(1);
            if (IN != 2 && n_ok) patch_store(pbuf ^ 1);

            // The DMA of H0(s+1) and the patch registers must have landed before E; waiting for them HERE, before the epilogue's stores
            // are issued, keeps those stores out of the wait (vmcnt retires in order: a vmcnt(0) behind 8 stores per lane waits for
            // their write acknowledgements, measured 12 k cycles per tile)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (cb == CB - 1) {
                // ---- per-tile epilogue: Y = A^T M A, the shared convolution epilogue on the block's 4 pixels --------------------
                const int co0 = ct.co_tile * NCO, x0 = ct.tx * TW, y0 = ct.ty * TH, b = ct.b;
                const bool want_stats = p.gn_stats != nullptr;
                f32x4 gs[FCO], gs2[FCO];
#pragma unroll
                for (int i = 0; i < FCO; ++i) { gs[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gs2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                const int oy = y0 + wave * 4 + 2 * br, ox = x0 + 2 * bc;
                const bool ok = oy < p.H;
                const int m00 = (b * p.H + (ok ? oy : 0)) * p.W + ox;
                f32x4 y[4][FCO];
#pragma unroll
                for (int i = 0; i < FCO; ++i) {
                    f32x4 P0[4], P1[4];
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) {
                        P0[xi] = acc[xi * 4][i] + acc[xi * 4 + 1][i] + acc[xi * 4 + 2][i];
                        P1[xi] = acc[xi * 4 + 1][i] - acc[xi * 4 + 2][i] - acc[xi * 4 + 3][i];
                    }
#pragma unroll
                    for (int tp = 0; tp < 16; ++tp) acc[tp][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int co = co0 + 16 * i + 4 * kq;
                    y[0][i] = conv_epilogue(P0[0] + P0[1] + P0[2], p, co, (size_t)m00, b);
                    y[1][i] = conv_epilogue(P1[0] + P1[1] + P1[2], p, co, (size_t)m00 + 1, b);
                    y[2][i] = conv_epilogue(P0[1] - P0[2] - P0[3], p, co, (size_t)m00 + p.W, b);
                    y[3][i] = conv_epilogue(P1[1] - P1[2] - P1[3], p, co, (size_t)m00 + p.W + 1, b);
                    if (!PROJ && ok) {
                        float* o = p.out + (size_t)m00 * p.Cout + co;
                        *reinterpret_cast<f32x4*>(o) = y[0][i];
                        *reinterpret_cast<f32x4*>(o + p.Cout) = y[1][i];
                        *reinterpret_cast<f32x4*>(o + (size_t)p.W * p.Cout) = y[2][i];
                        *reinterpret_cast<f32x4*>(o + (size_t)(p.W + 1) * p.Cout) = y[3][i];
                        if (want_stats) {
                            gs[i] += (y[0][i] + y[1][i]) + (y[2][i] + y[3][i]);
                            gs2[i] += (y[0][i] * y[0][i] + y[1][i] * y[1][i]) + (y[2][i] * y[2][i] + y[3][i] * y[3][i]);
                        }
                    }
                }
                if (PROJ) {
                    // d[tap][px] = sum over this tile's NCO channels of w[tap][co] * y[px][co]: a lane holds 4 FCO of them, the 4 kq lanes
                    // of a block the rest.  Reduce-SCATTER over kq (xor 32, then xor 16: 18 + 9 shuffles instead of 72 for an all-reduce):
                    // lane kq ends up with taps kq and kq + 4 (4 pixels each) and pixel kq of tap 8, and stores them with 4 two-pixel
                    // stores + 1 (a store instruction costs the wave 100+ cycles of issue; the all-reduce form needed 36)
                    const float* wlp = p.proj_w + co0 + 4 * kq;
                    float s[9][4];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                        for (int px = 0; px < 4; ++px) s[tap][px] = 0.f;
#pragma unroll
                        for (int i = 0; i < FCO; ++i) {
                            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wlp + tap * p.Cout + 16 * i);
#pragma unroll
                            for (int px = 0; px < 4; ++px)
#pragma unroll
                                for (int e = 0; e < 4; ++e) s[tap][px] = fmaf(y[px][i][e], w4[e], s[tap][px]);
                        }
                    }
                    const bool hi = (kq & 2) != 0, odd = (kq & 1) != 0;
                    float h[2][2][4], h8[2];                   // after xor 32: taps {0,1,4,5} (hi: {2,3,6,7}) as [t >> 2][t & 1][px]; tap 8 px {0,1} (hi: {2,3})
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int px = 0; px < 4; ++px) {
                                const float lo_v = s[4 * a + c][px], hi_v = s[4 * a + c + 2][px];
                                h[a][c][px] = (hi ? hi_v : lo_v) + __shfl_xor(hi ? lo_v : hi_v, 32, 64);
                            }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float lo_v = s[8][c], hi_v = s[8][c + 2];
                        h8[c] = (hi ? hi_v : lo_v) + __shfl_xor(hi ? lo_v : hi_v, 32, 64);
                    }
                    float f[2][4], f8;                          // after xor 16: taps kq, kq + 4; tap 8 pixel kq
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int px = 0; px < 4; ++px) f[a][px] = (odd ? h[a][1][px] : h[a][0][px]) + __shfl_xor(odd ? h[a][0][px] : h[a][1][px], 16, 64);
                    f8 = (odd ? h8[1] : h8[0]) + __shfl_xor(odd ? h8[0] : h8[1], 16, 64);
                    if (ok) {
                        float* po = p.proj_out + (size_t)ct.co_tile * 9 * p.M + m00;
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            float* o = po + (size_t)(4 * a + kq) * p.M;                 // m00, M and W are even: 8-byte aligned
                            *reinterpret_cast<f32x2*>(o) = f32x2{f[a][0], f[a][1]};
                            *reinterpret_cast<f32x2*>(o + p.W) = f32x2{f[a][2], f[a][3]};
                        }
                        po[(size_t)8 * p.M + (kq & 1) + (kq >> 1) * p.W] = f8;
                    }
                }
                if (want_stats) {
                    // GroupNorm statistics of this tile, deterministic: lanes by shuffles, waves through LDS (own scratch region: the
                    // stage buffers are live), groups in fp64; the two barriers are uniform (every wave runs the same stage)
#pragma unroll
                    for (int i = 0; i < FCO; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int o = 1; o < 16; o <<= 1) {
                                gs[i][e] += __shfl_xor(gs[i][e], o, 64);
                                gs2[i][e] += __shfl_xor(gs2[i][e], o, 64);
                            }
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
                    const int sub = cpg > NCO ? cpg / NCO : 1;
                    const int ngrp = cpg > NCO ? 1 : NCO / cpg;
                    const int span = cpg > NCO ? NCO : cpg;
                    if (tid < ngrp) {
                        double a = 0.0, a2 = 0.0;
                        for (int c = tid * span; c < (tid + 1) * span; ++c)
                            for (int w = 0; w < 4; ++w) {
                                a += (double)red[(w * NCO + c) * 2];
                                a2 += (double)red[(w * NCO + c) * 2 + 1];
                            }
                        const int chunks = tiles_x * tiles_y * sub;
                        const int chunk = (ct.ty * tiles_x + ct.tx) * sub + (ct.co_tile % sub);
                        const int g = co0 / cpg + tid;
                        double* o = p.gn_stats + (((size_t)b * chunks + chunk) * G + g) * 2;
                        o[0] = a;
                        o[1] = a2;
                    }
                    __syncthreads();                            // `red` is reused by the next tile
                }
                advance(ct);
            }
            __syncthreads();                                    // E: every wave is done with this stage; H0 of the next has landed
            if (n_ok) {
                slab_dma(1);                                    // H1 of the next stage
                if (IN == 2) {
                    expand(nt);                                 // L region (stored after M) -> W patch of the next stage
                }
            }
            const bool had = n_ok;
            loader_next();
            if (n_ok) patch_load();                             // registers <- patch of the stage after next
            if (IN == 2) { if (had) __syncthreads(); }          // X: W patch visible
            else pbuf ^= 1;
        }
    }
}


// OIHW [Cout][Cin][3][3] -> U[cb][xi*4 + eta][Cout][16],  U = G g G^T  (rows first, then columns; G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1])
__global__ void pack_w2d_weight_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin, int cs) {
    const int CB = cs / 16;
    const size_t total = (size_t)CB * 16 * Cout * 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c16 = (int)(i & 15);
        size_t r = i >> 4;
        const int co = (int)(r % Cout); r /= Cout;
        const int tap = (int)(r & 15);
        const int cb = (int)(r >> 4);
        const int xi = tap >> 2, eta = tap & 3;
        const int c = cb * 16 + c16;
        float v = 0.f;
        if (c < Cin) {
            const float* g = w + ((size_t)co * Cin + c) * 9;
            float rowv[3];                                   // (G g)[xi][kw]
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float g0 = g[kw], g1 = g[3 + kw], g2 = g[6 + kw];
                rowv[kw] = xi == 0 ? g0 : xi == 1 ? 0.5f * ((g0 + g1) + g2) : xi == 2 ? 0.5f * ((g0 - g1) + g2) : g2;
            }
            v = eta == 0 ? rowv[0] : eta == 1 ? 0.5f * ((rowv[0] + rowv[1]) + rowv[2]) : eta == 2 ? 0.5f * ((rowv[0] - rowv[1]) + rowv[2]) : rowv[2];
        }
        up[i] = v;
    }
}

}  // namespace

size_t sbgm_w2d_packed_floats(int Cout, int cs) { return (size_t)(cs / 16) * 16 * Cout * 16; }

int sbgm_launch_pack_w2d_weight(const float* w_oihw, float* up, int Cout, int Cin, int cs, hipStream_t st) {
    SBGM_CHECK(cs % 16 == 0 && Cin <= cs, "pack_w2d: padded Cin %d must be a multiple of 16", cs);
    const size_t total = sbgm_w2d_packed_floats(Cout, cs);
    hipLaunchKernelGGL(pack_w2d_weight_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, st, w_oihw, up,
                       Cout, Cin, cs);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_conv_w2d_gn_chunks(const ConvParams& p, const ConvTile& cfg) {
    if (p.gn_groups <= 0 || p.Cout % p.gn_groups || p.proj_w) return 0;
    const int nco = 16 * cfg.fco, cpg = p.Cout / p.gn_groups;
    if (cpg > nco ? cpg % nco != 0 : nco % cpg != 0) return 0;
    const int chunks = (p.W / TW) * ((p.H + TH - 1) / TH) * (cpg > nco ? cpg / nco : 1);
    return chunks <= 64 ? chunks : 0;
}

size_t sbgm_conv_w2d_bytes(const ConvTile& cfg, int in_mode) {
    if (cfg.lds == 3) {                          // persistent kernel: slab + two patch copies (mode 2: one + low-res region) + statistics scratch
        const size_t quads = (size_t)16 * 16 * cfg.fco * 4 +
                             (in_mode == 2 ? (size_t)PH * 128 + (size_t)(TH / 2 + 2) * (TW / 2 + 2) * 4 : (size_t)2 * PH * SY);
        return quads * 16 + (size_t)4 * 16 * cfg.fco * 2 * 4;
    }
    const int nbuf = cfg.lds == 2 ? 2 : 1;
    size_t quads = ((size_t)16 * 16 * cfg.fco * 4 + (in_mode == 2 ? (size_t)PH * 128 : (size_t)PH * SY)) * nbuf;
    if (in_mode == 2) quads += (size_t)(TH / 2 + 2) * (TW / 2 + 2) * 4 * nbuf;
    return quads * 16;
}

// number of co tiles a projection launch writes partial planes for (the tap_stencil launch sums them)
int sbgm_conv_w2d_proj_parts(const ConvParams& p, const ConvTile& cfg) { return p.Cout / (16 * cfg.fco); }

static int w2d_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// cfg.wino == 2; cfg.fco in {1, 2}; cfg.lds 1 (single stage buffer), 2 (double-buffered) or 3 (persistent kernel); p.wp = the
// F(2x2,3x3) weight image.
int sbgm_launch_conv_w2d(ConvParams p, const ConvTile& cfg, hipStream_t st) {
    SBGM_CHECK(p.Cs % 16 == 0 && p.W % 16 == 0 && p.H % 2 == 0, "conv_w2d: needs Cin padded to 16, W %% 16 == 0 and an even H (Cs=%d H=%d W=%d)", p.Cs, p.H, p.W);
    SBGM_CHECK(p.Cout % (16 * cfg.fco) == 0, "conv_w2d: Cout=%d not a multiple of the %d-channel tile", p.Cout, 16 * cfg.fco);
    SBGM_CHECK(p.act == SBGM_ACT_NONE || p.act == SBGM_ACT_RELU || p.act == SBGM_ACT_GELU, "conv_w2d: act=%d does not fuse", p.act);
    SBGM_CHECK((size_t)p.B * p.H * p.W * p.Cs * 4 < (1ull << 31), "conv_w2d: input tensor exceeds 2 GiB buffer window");
    SBGM_CHECK(p.proj_w == nullptr || p.proj_out != nullptr, "conv_w2d: tap projection needs proj_out");
    SBGM_CHECK(p.in_mode >= 0 && p.in_mode <= 2, "conv_w2d: in_mode=%d", p.in_mode);
    SBGM_CHECK(p.in_mode != 1 || p.in_affine != nullptr, "conv_w2d: in_mode 1 needs in_affine");
    SBGM_CHECK(p.in_mode == 2 || (p.in_skip == nullptr && p.in_act == SBGM_ACT_NONE), "conv_w2d: skip / activation on load need in_mode 2");
    if (p.gn_stats && sbgm_conv_w2d_gn_chunks(p, cfg) == 0) p.gn_stats = nullptr;
    p.OH = p.H; p.OW = p.W;
    p.M = p.B * p.H * p.W;
    p.cb_per_tap = p.Cs / 16;
    p.nsteps = 16 * p.cb_per_tap;
    p.x_bytes = (uint32_t)((size_t)p.B * p.H * p.W * p.Cs * 4 / (p.in_mode == 2 ? 4 : 1));
    p.w_bytes = (uint32_t)(sbgm_w2d_packed_floats(p.Cout, p.Cs) * 4);
    const int tiles = (p.W / TW) * ((p.H + TH - 1) / TH) * p.B * (p.Cout / (16 * cfg.fco));
    const bool db = cfg.lds == 2;
    const size_t lds = sbgm_conv_w2d_bytes(cfg, p.in_mode);
    SBGM_CHECK(lds <= 160 * 1024, "conv_w2d: tile needs %zu bytes of LDS", lds);
    int rc = 1;
    if (cfg.lds == 3) {                          // persistent workgroups: cfg.ws = workgroups per CU the grid is sized for (0 -> 2)
        const int per_cu = cfg.ws >= 1 && cfg.ws <= 8 ? cfg.ws : 2;
        const int grid = std::min(tiles, per_cu * w2d_cus());
        const bool proj = p.proj_w != nullptr;
#define SBGM_P3(FC, INV, PJ)                                                                                  \
    if (cfg.fco == FC && p.in_mode == INV && proj == PJ) {                                                    \
        if (lds > 64 * 1024)                                                                                  \
            SBGM_HIP(hipFuncSetAttribute((const void*)conv3x3_w2dp_kernel<FC, INV, PJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((conv3x3_w2dp_kernel<FC, INV, PJ>), dim3(grid), dim3(256), lds, st, p);           \
        rc = 0;                                                                                              \
    }
        SBGM_P3(1, 0, false) SBGM_P3(1, 1, false) SBGM_P3(1, 2, false) SBGM_P3(2, 0, false) SBGM_P3(2, 1, false) SBGM_P3(2, 2, false)
        SBGM_P3(1, 0, true) SBGM_P3(1, 1, true) SBGM_P3(1, 2, true) SBGM_P3(2, 0, true) SBGM_P3(2, 1, true) SBGM_P3(2, 2, true)
#undef SBGM_P3
        SBGM_CHECK(rc == 0, "conv_w2d: no persistent kernel for tile fco=%d in_mode=%d", cfg.fco, p.in_mode);
        SBGM_LAUNCH_CHECK();
        return 0;
    }
    const int minw = cfg.ws == 2 ? 2 : 1;
#define SBGM_L3(FC, MW, DBV, INV)                                                                             \
    if (cfg.fco == FC && minw == MW && db == DBV && p.in_mode == INV) {                                       \
        if (lds > 64 * 1024)                                                                                  \
            SBGM_HIP(hipFuncSetAttribute((const void*)conv3x3_w2d_kernel<FC, MW, DBV, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((conv3x3_w2d_kernel<FC, MW, DBV, INV>), dim3(tiles), dim3(256), lds, st, p);     \
        rc = 0;                                                                                              \
    }
#define SBGM_L(FC, MW) SBGM_L3(FC, MW, false, 0) SBGM_L3(FC, MW, true, 0) SBGM_L3(FC, MW, false, 1) SBGM_L3(FC, MW, true, 1) SBGM_L3(FC, MW, false, 2) SBGM_L3(FC, MW, true, 2)
    SBGM_L(1, 1) SBGM_L(2, 1) SBGM_L(2, 2)
#undef SBGM_L
#undef SBGM_L3
    SBGM_CHECK(rc == 0, "conv_w2d: no kernel for tile fco=%d ws=%d lds=%d in_mode=%d", cfg.fco, cfg.ws, cfg.lds, p.in_mode);
    SBGM_LAUNCH_CHECK();
    return 0;
}
