// 3x3 / stride 1 / pad 1 convolution as a 2-D Winograd F(2x2, 3x3) on the fp32 MFMA pipe (gfx950): persistent workgroups,
// weight slab by LDS-DMA in two rolling halves, double-buffered halo patch, software-pipelined sweep.
//
//     V = B^T d B   (4x4 input tile d of a 2x2 output block)      U = G g G^T   (3x3 filter g, packed once per upload)
//     M[xi][eta] = sum_ci U[xi][eta][co][ci] * V[xi][eta][ci]      Y = A^T M A
// 16 products per 2x2 outputs = 4 MFMA taps per output (direct: 9, F(2,3) along rows: 6); constants 0, +-1, 1/2 only, so the result
// differs from the direct fp32 convolution by ordinary rounding (~1e-6 relative).
//
// Mapping.  A tile is 16 x 16 pixels of ONE image x NCO = 16*FCO output channels.  A wave owns 4 tile rows = 16 blocks (2 block
// rows x 8 block columns; fragment column r16 -> block (r16 >> 3, r16 & 7)) and keeps all 16 (xi, eta) accumulator sets of them
// (128 registers at FCO = 2).  Workgroups are PERSISTENT: the grid is two workgroups per CU and each walks over a contiguous range
// of tiles, so the stage pipeline below runs across tile boundaries — the first stage of the next tile is loaded during the last
// stage of the current one and only the output transform + stores remain as per-tile overhead (with 64 input channels a tile is
// just 4 stages; measured: the one-tile-per-workgroup form lost 28 % of its in-loop rate to prologue / epilogue).
//
// Stage pipeline (one stage = 16 input channels).  Weight slab [16 taps][NCO][16 ch] in LDS, single copy, two halves: H0 = taps of
// xi 0,1 and H1 = taps of xi 2,3.  Halo patch (18 x 18 pixels x 16 ch), two copies.
//     E(s-1) | DMA H1(s) ; global loads of patch(s+1) -> registers | sweep xi 0,1 of stage s (reads H0(s), patch(s))
//     M(s)   | DMA H0(s+1)                                          | sweep xi 2,3 of stage s (reads H1(s), patch(s))
//            | patch(s+1) registers -> other patch copy ; last stage of a tile: output transform, epilogue, stores
//     E(s)   | ...
// E and M are workgroup barriers; the compiler drains vmcnt before each (an LDS-DMA is a pending LDS write on that counter), which
// is exactly the hand-off rule: a half is read only after the issuing waves' wait AND a barrier, and is overwritten only after a
// barrier that every reader has passed.  The weights cost no registers and no ds_write; each DMA half has half a sweep to land.
//
// Input modes (template IN): 1 = per-(sample, channel) affine on load (GroupNorm of the producer), 2 = bilinear x2 upsample on load
// from the LOW-resolution map (optionally act(x*scale + shift + skip) first).  Mode 2 keeps ONE patch copy holding the
// column-transformed rows W[row][eta] = (d B)[row][eta], built from the low-resolution pixels parked in a small LDS region (the
// interpolation's column taps fold into B, conv_lds.hip header); it needs a third barrier per stage (E | expand | X | sweep).
//
// LDS layouts (conflict-free for every ds_read_b128 lane group, tools/lds_bank_check.py):
//   weight slab  [tap][co][4 quads], quad rotated by (co & 15) >> 1   (the DMA applies the rotation on its per-lane SOURCE address)
//   raw patch    [row (stride 74 quads)][px][4 quads], quad rotated by 2 * (px >> 2)          (modes 0, 1)
//   W patch      [row][eta][pair][4 quads], quad rotated by 2 * ((row >> 1) & 1)              (mode 2)
#include "common.h"
#include "kernels.h"
#include "conv_common.h"

namespace {

constexpr int TW = 16, TH = 16;
constexpr int PH = TH + 2, PWID = TW + 2;
constexpr int SY = PWID * 4 + 2;        // raw patch row stride in quads

__device__ __forceinline__ int wslot(int row, int quad) { return row * 4 + ((quad + (row >> 1)) & 3); }
__device__ __forceinline__ int pslot(int px, int quad) { return px * 4 + ((quad + 2 * (px >> 2)) & 3); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int FCO, int IN, bool PROJ>
__global__ __launch_bounds__(256, 2) void conv3x3_w2d_kernel(const ConvParams p) {
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

#ifdef EXP_STAGGER
    if ((blockIdx.x / EXP_STAGGER_DIV) & 1) {                  // de-phase the two workgroups of a CU by ~half a stage
        for (int i = 0; i < EXP_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
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

#ifdef EXP_STAMP
    unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_a = __builtin_amdgcn_s_memtime(), t_b;
    const unsigned long long t_start = t_a;
#define STAMP(i_) { t_b = __builtin_amdgcn_s_memtime(); st[i_] += t_b - t_a; t_a = t_b; }
#else
#define STAMP(i_)
#endif
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
#ifdef EXP_PIN
                    __builtin_amdgcn_sched_barrier(0);
#endif
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
            STAMP(0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's DMA of H1(s) (and the patch registers) have landed
            STAMP(1)
            __syncthreads();                                    // M: H0 is free, H1 of this stage has landed
            STAMP(2)
#ifndef ABL_NO_DMA
            if (n_ok) slab_dma(0);                              // H0 of the next stage
#endif
            if (IN == 2) { if (n_ok) l_store(); }               // L region: last read by this stage's expand, before E
            sweep_half(1);
            STAMP(3)
#ifndef ABL_NO_PSTORE
            if (IN != 2 && n_ok) patch_store(pbuf ^ 1);
#else
            if (IN != 2 && n_ok) { for (int u = 0; u < NRP; ++u) asm volatile("" :: "v"(rp[u])); }
#endif
            STAMP(4)

            // The DMA of H0(s+1) and the patch registers must have landed before E; waiting for them HERE, before the epilogue's stores
            // are issued, keeps those stores out of the wait (vmcnt retires in order: a vmcnt(0) behind 8 stores per lane waits for
            // their write acknowledgements, measured 12 k cycles per tile)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ABL_NO_EPI
            if (cb == CB - 1) { for (int tp = 0; tp < 16; ++tp) for (int i = 0; i < FCO; ++i) asm volatile("" :: "v"(acc[tp][i])); advance(ct); }
            if (false) {
#else
            if (cb == CB - 1) {
#endif
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
                    const float* wlp = p.proj_w + co0 + 4 * kq;
                    float* po = p.proj_out + (size_t)ct.co_tile * 9 * p.M;
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

            STAMP(5)
            __syncthreads();                                    // E: every wave is done with this stage; H0 of the next has landed
            STAMP(7)
            if (n_ok) {
#ifndef ABL_NO_DMA
                slab_dma(1);                                    // H1 of the next stage
#endif
                if (IN == 2) {
                    expand(nt);                                 // L region (stored after M) -> W patch of the next stage
                }
            }
            const bool had = n_ok;
            loader_next();
#ifndef ABL_NO_PLOAD
            if (n_ok) patch_load();                             // registers <- patch of the stage after next
#endif
            if (IN == 2) { if (had) __syncthreads(); }          // X: W patch visible
            else pbuf ^= 1;
            STAMP(8)
        }
    }
#ifdef EXP_STAMP
    if (p.proj_w == nullptr && p.proj_out != nullptr && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.proj_out) + ((size_t)blockIdx.x * 4 + wave) * 12;
        for (int i = 0; i < 9; ++i) o[i] = st[i];
        o[9] = t_a - t_start; o[10] = (unsigned long long)(t_end - t_begin) * CB;
    }
#endif
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
    size_t quads = (size_t)16 * 16 * cfg.fco * 4 + (in_mode == 2 ? (size_t)PH * 128 + (size_t)(TH / 2 + 2) * (TW / 2 + 2) * 4 : (size_t)2 * PH * SY);
    return quads * 16 + (size_t)4 * 16 * cfg.fco * 2 * 4;
}

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

// cfg.wino == 2; cfg.fco in {1, 2}; cfg.ws = workgroups per CU the grid is sized for (0 -> 2); p.wp = the F(2x2,3x3) weight image.
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
    const int per_cu = cfg.ws >= 1 && cfg.ws <= 8 ? cfg.ws : 2;
    const int grid = std::min(tiles, per_cu * w2d_cus());
    const size_t lds = sbgm_conv_w2d_bytes(cfg, p.in_mode);
    SBGM_CHECK(lds <= 160 * 1024, "conv_w2d: tile needs %zu bytes of LDS", lds);
    int rc = 1;
    const bool proj = p.proj_w != nullptr;
#define SBGM_L3(FC, INV, PJ)                                                                                  \
    if (cfg.fco == FC && p.in_mode == INV && proj == PJ) {                                                    \
        if (lds > 64 * 1024)                                                                                  \
            SBGM_HIP(hipFuncSetAttribute((const void*)conv3x3_w2d_kernel<FC, INV, PJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((conv3x3_w2d_kernel<FC, INV, PJ>), dim3(grid), dim3(256), lds, st, p);            \
        rc = 0;                                                                                              \
    }
    SBGM_L3(1, 0, false) SBGM_L3(1, 1, false) SBGM_L3(1, 2, false) SBGM_L3(2, 0, false) SBGM_L3(2, 1, false) SBGM_L3(2, 2, false)
    SBGM_L3(1, 0, true) SBGM_L3(1, 2, true) SBGM_L3(2, 0, true) SBGM_L3(2, 2, true)
#undef SBGM_L3
    SBGM_CHECK(rc == 0, "conv_w2d: no kernel for tile fco=%d in_mode=%d", cfg.fco, p.in_mode);
    SBGM_LAUNCH_CHECK();
    return 0;
}
