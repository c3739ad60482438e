// Experiment (development tool): the 2-D Winograd convolution tiled for v_mfma_f32_32x32x2_f32.
// Includes the shipped translation unit (layout helpers, packer, launcher) and adds conv3x3_w2x_kernel: one 16 x 16 tile per workgroup,
// 32 output channels, input mode 0 only.  Wave w: block group bg = w & 1 (4 block rows x 8 block columns = 32 blocks of 2 x 2 outputs =
// the MFMA's 32 columns), product half ph = w >> 1 (xi in {2 ph, 2 ph + 1}, all eta: 8 accumulator sets of 16 registers).
// Lane l: column j = l & 31 -> block (j >> 3, j & 7); k half kh = l >> 5: per 8-channel group u the lane owns channel quad 2 u + kh on
// both operands (A: weight row co = j, B: its block), so one ds_read_b128 feeds 4 MFMAs on either side.
// Per 16-channel stage and wave: 24 patch reads (3 rows x 4 columns x 2 quads) + 16 A reads for 64 MFMAs of 64 cycles —
// the shipped 16x16x4 mapping needs 32 + 32 for the same matrix work.
#include "../../sbgm_danra_amd/csrc/conv_w2d.hip"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SYX = 73;                                     // patch row stride in quads (conflict-free with the rotation below)
__device__ __forceinline__ int pslotx(int px, int quad) { return px * 4 + ((quad + (px >> 2) + 2 * (px >> 3)) & 3); }
__device__ __forceinline__ int wslotx(int co, int quad) { return co * 4 + ((quad + (co >> 2)) & 3); }

template <int MINW, bool DB>
__global__ __launch_bounds__(256, MINW) void conv3x3_w2x_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NCO = 32;
    constexpr int WQ = 16 * NCO * 4;                            // weight quads per stage
    constexpr int PQ = PH * PWID * 4;
    constexpr int PREG = PH * SYX;
    constexpr int STAGE_QUADS = WQ + PREG;
    f32x4* wl = reinterpret_cast<f32x4*>(smem_raw);
    f32x4* pt = wl + WQ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bg = wave & 1, ph = wave >> 1;
    const int j = lane & 31, kh = lane >> 5;
    const int brow = 4 * bg + (j >> 3), bcol = j & 7;           // block inside the tile

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

    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    constexpr int WPT = WQ / 256;               // 8
    constexpr int PPT = (PQ + 255) / 256;       // 6
    constexpr int RW = 256 / (NCO * 4);         // taps per 256-quad round (2)
    f32x4 rw[WPT], rp[PPT];
    constexpr uint32_t OOB = 0x80000000u;
    const int tl = tid / (NCO * 4), rem = tid - tl * (NCO * 4);
    const uint32_t wlane = (uint32_t)(tl * p.Cout * 16 + rem * 4) * 4u;
    uint32_t poff[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int q = tid + 256 * u;
        const int quad = q & 3, pix = q >> 2;
        const int py = pix / PWID, px = pix - py * PWID;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = (q < PQ) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        poff[u] = ok ? (uint32_t)(((b * p.H + iy) * p.W + ix) * p.Cs + quad * 4) * 4u : OOB;
    }
    auto stage_load = [&](int cb) {
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const uint32_t su = (uint32_t)(((cb * 16 + u * RW) * p.Cout + co0) * 16) * 4u;
            rw[u] = buf_load4(wr, wlane + su);
        }
        const uint32_t cbo = (uint32_t)cb * 64u;
#pragma unroll
        for (int u = 0; u < PPT; ++u) rp[u] = buf_load4(xr, poff[u] + cbo);
    };
    f32x4* const wl0 = wl;
    f32x4* const pt0 = pt;
    auto stage_store = [&](int buf) {
        f32x4* wd = wl0 + buf * STAGE_QUADS;
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int q = tid + 256 * u;                       // [tap][co][quad]
            const int co = (q >> 2) & 31;
            wd[(q & ~3) + (((q & 3) + (co >> 2)) & 3)] = rw[u];
        }
        f32x4* pd = pt0 + buf * STAGE_QUADS;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int q = tid + 256 * u;
            const int pix = q >> 2, py = pix / PWID, px = pix - py * PWID;
            if (q < PQ) pd[py * SYX + pslotx(px, q & 3)] = rp[u];
        }
    };

    const int prow0 = 2 * brow + ph;                            // first of the 3 patch rows this wave's xi pair reads
    stage_load(0);
    stage_store(0);
    if (DB && CB > 1) stage_load(1);
    for (int cb = 0; cb < CB; ++cb) {
        if (!DB) {
            if (cb + 1 < CB) stage_load(cb + 1);
            __syncthreads();
        } else {
            __syncthreads();
            if (cb + 1 < CB) {
                stage_store((cb + 1) & 1);
                if (cb + 2 < CB) stage_load(cb + 2);
            }
            wl = wl0 + (cb & 1) * STAGE_QUADS;
            pt = pt0 + (cb & 1) * STAGE_QUADS;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = 2 * u + kh;
            f32x4 e0[4], e1[4], e2[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int so = pslotx(2 * bcol + c, q);
                e0[c] = pt[(prow0 + 0) * SYX + so];
                e1[c] = pt[(prow0 + 1) * SYX + so];
                e2[c] = pt[(prow0 + 2) * SYX + so];
            }
            f32x4 Ta[4], Tb[4];
            if (ph == 0) {                                      // wave-uniform
#pragma unroll
                for (int c = 0; c < 4; ++c) { Ta[c] = e0[c] - e2[c]; Tb[c] = e1[c] + e2[c]; }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) { Ta[c] = e1[c] - e0[c]; Tb[c] = e0[c] - e2[c]; }
            }
            f32x4 V[2][4];
            V[0][0] = Ta[0] - Ta[2]; V[0][1] = Ta[1] + Ta[2]; V[0][2] = Ta[2] - Ta[1]; V[0][3] = Ta[1] - Ta[3];
            V[1][0] = Tb[0] - Tb[2]; V[1][1] = Tb[1] + Tb[2]; V[1][2] = Tb[2] - Tb[1]; V[1][3] = Tb[1] - Tb[3];
            const int ao = wslotx(j, q);
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                const int tap = (2 * ph + (pr >> 2)) * 4 + (pr & 3);
                const f32x4 a = wl[tap * NCO * 4 + ao];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[pr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], V[pr >> 2][pr & 3][e], acc[pr], 0, 0, 0);
            }
        }
        if (!DB) {
            __syncthreads();
            if (cb + 1 < CB) stage_store(0);
        }
    }

    // ---- epilogue: Z[xi][jj] per wave, exchange one Z with the partner wave (same bg, other ph), finish output row ii = ph ----------
    // accumulator element r of a lane: output channel co0 + 8 (r >> 2) + 4 kh + (r & 3), block j
    f32x16 Za[2], Zb[2];                                        // [jj] for xi = 2 ph and 2 ph + 1
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        Za[0][e] = acc[0][e] + acc[1][e] + acc[2][e];
        Za[1][e] = acc[1][e] - acc[2][e] - acc[3][e];
        Zb[0][e] = acc[4][e] + acc[5][e] + acc[6][e];
        Zb[1][e] = acc[5][e] - acc[6][e] - acc[7][e];
    }
    // Y[0] = Z0 + Z1 + Z2, Y[1] = Z1 - Z2 - Z3.  ph 0 keeps P = Z0 + Z1 and sends Z1; ph 1 keeps Q = Z2 + Z3 and sends Z2.
    __syncthreads();                                            // the stage buffers are free
    float* xch = reinterpret_cast<float*>(smem_raw);            // [wave][32 values][64 lanes]
    {
        float* mine = xch + wave * 32 * 64 + lane;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) mine[(jj * 16 + e) * 64] = ph == 0 ? Zb[jj][e] : Za[jj][e];
    }
    __syncthreads();
    f32x16 Y[2];
    {
        const float* other = xch + (wave ^ 2) * 32 * 64 + lane;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float o = other[(jj * 16 + e) * 64];
                Y[jj][e] = ph == 0 ? (Za[jj][e] + Zb[jj][e]) + o : o - (Za[jj][e] + Zb[jj][e]);
            }
    }
    const int oy = y0 + 2 * brow + ph, ox = x0 + 2 * bcol;
    if (oy < p.H) {
        const size_t m0 = ((size_t)b * p.H + oy) * p.W + ox;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = co0 + 8 * g + 4 * kh;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                f32x4 v = {Y[jj][4 * g], Y[jj][4 * g + 1], Y[jj][4 * g + 2], Y[jj][4 * g + 3]};
                v = conv_epilogue(v, p, co, m0 + jj, b);
                *reinterpret_cast<f32x4*>(p.out + (m0 + jj) * p.Cout + co) = v;
            }
        }
    }
}

}  // namespace

int launch_w2x(const ConvParams& p0, int minw, int db, hipStream_t st) {
    ConvParams p = p0;
    p.cb_per_tap = p.Cs / 16;
    p.M = p.B * p.H * p.W;
    p.x_bytes = (uint32_t)((size_t)p.B * p.H * p.W * p.Cs * 4);
    p.w_bytes = (uint32_t)(sbgm_w2d_packed_floats(p.Cout, p.Cs) * 4);
    const int tiles = (p.W / 16) * ((p.H + 15) / 16) * p.B * (p.Cout / 32);
    const size_t stage = (size_t)(16 * 32 * 4 + PH * SYX) * 16;
    const size_t lds = std::max((size_t)(db ? 2 : 1) * stage, (size_t)4 * 32 * 64 * 4);
#define W2X_LAUNCH(MW, D)                                                                                                        \
    {                                                                                                                              \
        static bool attr = false;                                                                                                  \
        if (!attr) { (void)hipFuncSetAttribute((const void*)conv3x3_w2x_kernel<MW, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((conv3x3_w2x_kernel<MW, D>), dim3(tiles), dim3(256), lds, st, p);                                        \
    }
    if (minw == 2) { if (db) W2X_LAUNCH(2, true) else W2X_LAUNCH(2, false) }
    else { if (db) W2X_LAUNCH(1, true) else W2X_LAUNCH(1, false) }
    return hipGetLastError() != hipSuccess;
}

// =================================================================================================================================
// Persistent form of the 32x32x2 tiling (the structure of conv3x3_w2dp_kernel: two workgroups per CU walk over tiles, weight slab by
// LDS-DMA in two rolling halves, double-buffered halo patch).  Slab halves are by 8-channel group u: H_u = [16 taps][32 co][quads 2u,
// 2u + 1], slot (tap * 32 + co) * 2 + ((kh + (co >> 3)) & 1).  Input modes 0 and 1.
// Tile end: the two waves of a block group exchange one Z each through LDS that is dead at that point (slab half H1 + the swept patch
// copy), two extra barriers per tile.
// =================================================================================================================================
namespace {

template <int IN>
__global__ __launch_bounds__(256, 2) void conv3x3_w2xp_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NCO = 32;
    constexpr int WQ = 16 * NCO * 4;
    constexpr int HQ = WQ / 2;
    constexpr int PQ = PH * PWID * 4;
    constexpr int PREG = PH * SYX;
    static_assert(IN != 2, "mode 2 not in this experiment");
    f32x4* const wl = reinterpret_cast<f32x4*>(smem_raw);      // [u][tap][co][2 slots]
    f32x4* const pt0 = wl + WQ;                                 // 2 patch copies

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bg = wave & 1, ph = wave >> 1;
    const int j = lane & 31, kh = lane >> 5;
    const int brow = 4 * bg + (j >> 3), bcol = j & 7;

    const int tiles_x = p.W / TW, tiles_y = (p.H + TH - 1) / TH, n_co = p.Cout / NCO;
    const int n_tiles = tiles_x * tiles_y * p.B * n_co;
    const int CB = p.cb_per_tap;
    const int L = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int t_begin = (int)((long long)L * n_tiles / gridDim.x), t_end = (int)((long long)(L + 1) * n_tiles / gridDim.x);
    if (t_begin >= t_end) return;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);

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
    int n_tile = t_begin, n_cb = 0;
    bool n_ok = true;

    constexpr int PPT = (PQ + 255) / 256;
    constexpr uint32_t OOB = 0x80000000u;
    uint32_t poff[PPT];
    f32x4 rp[PPT], sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    auto set_poff = [&](const Tile& tl) {
        const int x0 = tl.tx * TW, y0 = tl.ty * TH;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int q = tid + 256 * u;
            const int quad = q & 3, pix = q >> 2;
            const int py = pix / PWID, px = pix - py * PWID;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool ok = (q < PQ) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            poff[u] = ok ? (uint32_t)(((tl.b * p.H + iy) * p.W + ix) * p.Cs + quad * 4) * 4u : OOB;
        }
    };
    auto patch_load = [&]() {
        if (IN == 1) {
            const float* ap = p.in_affine + (((size_t)nt.b * (p.Cs >> 2) + (tid & 3)) * 2) * 4 + n_cb * 32;
            sc = *reinterpret_cast<const f32x4*>(ap);
            sh = *reinterpret_cast<const f32x4*>(ap + 4);
        }
        const uint32_t cbo = (uint32_t)n_cb * 64u;
#pragma unroll
        for (int u = 0; u < PPT; ++u) rp[u] = buf_load4(xr, poff[u] + cbo);
    };
    // LDS-DMA of slab half u of the loader's stage: a wave instruction fills 64 consecutive slots = all 32 co x 2 slots of one tap;
    // lane -> (co = lane >> 1, slot = lane & 1), the slot holds source quad 2u + ((slot - (co >> 3)) & 1)
    constexpr int DPW = HQ / 64 / 4;                            // 4 DMA instructions per wave and half
    const uint32_t dma_lane = (uint32_t)((lane >> 1) * 64 + ((((lane & 1) - ((lane >> 1) >> 3)) & 1)) * 16);
    auto slab_dma = [&](int half) {
        const char* src = reinterpret_cast<const char*>(p.wp);
#pragma unroll
        for (int u = 0; u < DPW; ++u) {
            const int tap = wave * DPW + u;
            const size_t off = ((size_t)((n_cb * 16 + tap) * p.Cout + nt.co_tile * NCO) * 16) * 4 + (size_t)half * 32;     // wavefront-uniform
            __builtin_amdgcn_global_load_lds((gptr_t)(src + off + dma_lane), (lptr_t)(wl + half * HQ + tap * 64), 16, 0, 0);
        }
    };
    auto patch_store = [&](int buf) {
        f32x4* pd = pt0 + buf * PREG;
        const int x0 = nt.tx * TW, y0 = nt.ty * TH;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int q = tid + 256 * u;
            const int pix = q >> 2, py = pix / PWID, px = pix - py * PWID;
            f32x4 v = rp[u];
            if (IN == 1) {
                const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                v = ok ? v * sc + sh : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (q < PQ) pd[py * SYX + pslotx(px, q & 3)] = v;
        }
    };
    auto loader_next = [&]() {
        if (++n_cb == CB) {
            n_cb = 0;
            ++n_tile;
            n_ok = n_tile < t_end;
            if (n_ok) { advance(nt); set_poff(nt); }
        }
    };

    const int prow0 = 2 * brow + ph;
    const int aslot = j * 2 + ((kh + (j >> 3)) & 1);
    int so[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) so[c] = pslotx(2 * bcol + c, kh);         // quad kh; quad 2 + kh is the same slot rotated by 2

    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    set_poff(nt);
    slab_dma(0);
    slab_dma(1);
    patch_load();
    patch_store(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    loader_next();
    if (n_ok) patch_load();

    int pbuf = 0;
    for (int tile = t_begin; tile < t_end; ++tile) {
        for (int cb = 0; cb < CB; ++cb) {
            const f32x4* pt = pt0 + pbuf * PREG;
            auto sweep_half = [&](int u) {
                f32x4 Ta[4], Tb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int s = (so[c] & ~3) | ((so[c] + 2 * u) & 3);
                    const f32x4 e0 = pt[(prow0 + 0) * SYX + s], e1 = pt[(prow0 + 1) * SYX + s], e2 = pt[(prow0 + 2) * SYX + s];
                    if (ph == 0) { Ta[c] = e0 - e2; Tb[c] = e1 + e2; }
                    else { Ta[c] = e1 - e0; Tb[c] = e0 - e2; }
                }
                f32x4 V[2][4];
                V[0][0] = Ta[0] - Ta[2]; V[0][1] = Ta[1] + Ta[2]; V[0][2] = Ta[2] - Ta[1]; V[0][3] = Ta[1] - Ta[3];
                V[1][0] = Tb[0] - Tb[2]; V[1][1] = Tb[1] + Tb[2]; V[1][2] = Tb[2] - Tb[1]; V[1][3] = Tb[1] - Tb[3];
                const f32x4* wh = wl + u * HQ + (8 * ph) * 64 + aslot;
                f32x4 a_cur = wh[0], a_nxt;
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) {
                    if (pr + 1 < 8) a_nxt = wh[(pr + 1) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[pr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[e], V[pr >> 2][pr & 3][e], acc[pr], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a_cur = a_nxt;
                }
            };

            sweep_half(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                    // M: H0 is free, H1 of this stage has landed
            if (n_ok) slab_dma(0);
            sweep_half(1);
            if (n_ok) patch_store(pbuf ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const bool last = cb == CB - 1;
            f32x16 K[2];
            if (last) {
                __syncthreads();                                // B1: every wave has finished the tile's sweeps: H1 and patch[pbuf] are dead
                float* x0r = reinterpret_cast<float*>(wl + HQ) + wave * 16 * 64 + lane;                      // jj = 0 -> H1 (16 KB)
                float* x1r = reinterpret_cast<float*>(pt0 + pbuf * PREG) + wave * 16 * 64 + lane;             // jj = 1 -> patch copy (16 of 21 KB)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float za0 = acc[0][e] + acc[1][e] + acc[2][e], za1 = acc[1][e] - acc[2][e] - acc[3][e];
                    const float zb0 = acc[4][e] + acc[5][e] + acc[6][e], zb1 = acc[5][e] - acc[6][e] - acc[7][e];
                    x0r[e * 64] = ph == 0 ? zb0 : za0;
                    x1r[e * 64] = ph == 0 ? zb1 : za1;
                    K[0][e] = za0 + zb0;
                    K[1][e] = za1 + zb1;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            }
            __syncthreads();                                    // E
            if (last) {
                const float* y0r = reinterpret_cast<const float*>(wl + HQ) + (wave ^ 2) * 16 * 64 + lane;
                const float* y1r = reinterpret_cast<const float*>(pt0 + pbuf * PREG) + (wave ^ 2) * 16 * 64 + lane;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float o0 = y0r[e * 64], o1 = y1r[e * 64];
                    K[0][e] = ph == 0 ? K[0][e] + o0 : o0 - K[0][e];
                    K[1][e] = ph == 0 ? K[1][e] + o1 : o1 - K[1][e];
                }
                __syncthreads();                                // B3: the exchange has been read; H1 may be overwritten
            }
            if (n_ok) slab_dma(1);
            const bool had = n_ok;
            (void)had;
            loader_next();
            if (n_ok) patch_load();
            pbuf ^= 1;
            if (last) {
                const int co0 = ct.co_tile * NCO, x0 = ct.tx * TW, y0 = ct.ty * TH, b = ct.b;
                const int oy = y0 + 2 * brow + ph, ox = x0 + 2 * bcol;
                if (oy < p.H) {
                    const size_t m0 = ((size_t)b * p.H + oy) * p.W + ox;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = co0 + 8 * g + 4 * kh;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            f32x4 v = {K[jj][4 * g], K[jj][4 * g + 1], K[jj][4 * g + 2], K[jj][4 * g + 3]};
                            v = conv_epilogue(v, p, co, m0 + jj, b);
                            *reinterpret_cast<f32x4*>(p.out + (m0 + jj) * p.Cout + co) = v;
                        }
                    }
                }
                advance(ct);
            }
        }
    }
}

}  // namespace

int launch_w2xp(const ConvParams& p0, int per_cu, hipStream_t st) {
    ConvParams p = p0;
    p.cb_per_tap = p.Cs / 16;
    p.M = p.B * p.H * p.W;
    p.x_bytes = (uint32_t)((size_t)p.B * p.H * p.W * p.Cs * 4);
    p.w_bytes = (uint32_t)(sbgm_w2d_packed_floats(p.Cout, p.Cs) * 4);
    const int tiles = (p.W / 16) * ((p.H + 15) / 16) * p.B * (p.Cout / 32);
    const size_t lds = (size_t)(16 * 32 * 4 + 2 * PH * SYX) * 16;
    const int grid = std::min(tiles, per_cu * 256);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv3x3_w2xp_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_w2xp_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    if (p.in_mode == 1) hipLaunchKernelGGL((conv3x3_w2xp_kernel<1>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv3x3_w2xp_kernel<0>), dim3(grid), dim3(256), lds, st, p);
    return hipGetLastError() != hipSuccess;
}
