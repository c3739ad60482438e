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
#ifdef EXP_NO_RESTAGE
#define CBN 1
#else
#define CBN CB
#endif
#ifdef EXP_STAMP
    unsigned long long st_sweep = 0, st_bar1 = 0, st_store = 0, st_bar0 = 0, t_a = __builtin_amdgcn_s_memtime(), t_b;
    const unsigned long long t_start = t_a;
#define STAMP(acc_) { t_b = __builtin_amdgcn_s_memtime(); acc_ += t_b - t_a; t_a = t_b; }
#else
#define STAMP(acc_)
#endif
    for (int cb = 0; cb < CB; ++cb) {
        if (!DB) {
            if (cb + 1 < CBN) stage_load(cb + 1);
            __syncthreads();
            STAMP(st_bar0)
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

#ifdef EXP_PIPE
        // ---- software-pipelined sweep: the LDS reads (A fragments of the next (xi, eta) group, patch rows of the next xi) are issued
        // one step ahead of the MFMAs that consume them; V of the next xi is formed during the last two groups of the current xi
        {
            constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};
            auto ldrow = [&](int rr, int c) -> f32x4 {
                if (IN == 2) return pt[(r0 + rr) * 128 + coff[c] + ((kq + 2 * (((r0 + rr) >> 1) & 1)) & 3)];
                return pt[(r0 + rr) * SY + coff[c]];
            };
            f32x4 a_cur[FCO], a_nxt[FCO], V[4], Vn[4], da[4], db[4];
#pragma unroll
            for (int i = 0; i < FCO; ++i) a_cur[i] = wl[(16 * i) * 4 + aoff];
#pragma unroll
            for (int c = 0; c < 4; ++c) { da[c] = ldrow(RA[0], c); db[c] = ldrow(RB[0], c); }
#pragma unroll
            for (int c = 0; c < 4; ++c) da[c] = da[c] - db[c];
            if (IN == 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) V[c] = da[c];
            } else {
                V[0] = da[0] - da[2]; V[1] = da[1] + da[2]; V[2] = da[2] - da[1]; V[3] = da[1] - da[3];
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int xi = g >> 2, eta = g & 3;
                if (g + 1 < 16) {
#pragma unroll
                    for (int i = 0; i < FCO; ++i) a_nxt[i] = wl[((g + 1) * NCO + 16 * i) * 4 + aoff];
                }
                if (eta == 0 && xi < 3) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) { da[c] = ldrow(RA[xi + 1], c); db[c] = ldrow(RB[xi + 1], c); }
                }
                if (eta == 2 && xi < 3) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) da[c] = xi + 1 == 1 ? da[c] + db[c] : da[c] - db[c];
                }
                if (eta == 3 && xi < 3) {
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
#pragma unroll
                for (int i = 0; i < FCO; ++i) a_cur[i] = a_nxt[i];
                if (eta == 3 && xi < 3) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) V[c] = Vn[c];
                }
#ifdef EXP_SCHEDBAR
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
#else
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
#ifdef EXP_NO_XFORM
                T[c] = da; asm volatile("" :: "v"(db));
#else
                T[c] = xi == 1 ? da + db : da - db;
#endif
            }
            f32x4 V[4];
            if (IN == 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) V[c] = T[c];
            } else {
#ifdef EXP_NO_XFORM
                V[0] = T[0]; V[1] = T[1]; V[2] = T[2]; V[3] = T[3];
#else
                V[0] = T[0] - T[2]; V[1] = T[1] + T[2]; V[2] = T[2] - T[1]; V[3] = T[1] - T[3];
#endif
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
#endif
        if (!DB) {
            STAMP(st_sweep)
            if (IN == 2 && cb + 1 < CB) stage_store_l(0);    // the L region was last read before this stage's first barrier
            __syncthreads();                     // every wave is done reading this stage
            STAMP(st_bar1)
            if (cb + 1 < CBN) {
                stage_store_w(0);
                if (IN != 2) stage_store_p(0);
                else expand(0, 0);
            }
            STAMP(st_store)
        } else if (IN == 2 && cb + 2 < CB) {
            stage_store_l(cb & 1);
        }
    }

#ifdef EXP_STAMP
    if (p.proj_w == nullptr && p.proj_out != nullptr && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.proj_out) + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = st_bar0; o[1] = st_sweep; o[2] = st_bar1; o[3] = st_store; o[4] = t_a - t_start; o[5] = t_start;
    }
#endif
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
    const int nbuf = cfg.lds == 2 ? 2 : 1;
    size_t quads = ((size_t)16 * 16 * cfg.fco * 4 + (in_mode == 2 ? (size_t)PH * 128 : (size_t)PH * SY)) * nbuf;
    if (in_mode == 2) quads += (size_t)(TH / 2 + 2) * (TW / 2 + 2) * 4 * nbuf;
    return quads * 16;
}

// number of co tiles a projection launch writes partial planes for (the tap_stencil launch sums them)
int sbgm_conv_w2d_proj_parts(const ConvParams& p, const ConvTile& cfg) { return p.Cout / (16 * cfg.fco); }

// cfg.wino == 2; cfg.fco in {1, 2}; cfg.lds 1 (single stage buffer) or 2 (double-buffered); p.wp = the F(2x2,3x3) weight image.
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
    const int minw = cfg.ws;
#define SBGM_L3(FC, MW, DBV, INV)                                                                             \
    if (cfg.fco == FC && minw == MW && db == DBV && p.in_mode == INV) {                                       \
        if (lds > 64 * 1024)                                                                                  \
            SBGM_HIP(hipFuncSetAttribute((const void*)conv3x3_w2d_kernel<FC, MW, DBV, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((conv3x3_w2d_kernel<FC, MW, DBV, INV>), dim3(tiles), dim3(256), lds, st, p);     \
        rc = 0;                                                                                              \
    }
#define SBGM_L(FC, MW) SBGM_L3(FC, MW, false, 0) SBGM_L3(FC, MW, true, 0) SBGM_L3(FC, MW, false, 1) SBGM_L3(FC, MW, true, 1) SBGM_L3(FC, MW, false, 2) SBGM_L3(FC, MW, true, 2)
    SBGM_L(1, 1) SBGM_L(2, 1) SBGM_L(2, 2) SBGM_L(1, 3) SBGM_L(1, 2)
#undef SBGM_L
#undef SBGM_L3
    SBGM_CHECK(rc == 0, "conv_w2d: no kernel for tile fco=%d ws=%d lds=%d in_mode=%d", cfg.fco, cfg.ws, cfg.lds, p.in_mode);
    SBGM_LAUNCH_CHECK();
    return 0;
}
