// K25 / K26: the per-token halves of ImageSelfAttention (reference sbgm/score_unet.py:127-148) as two token-tile kernels, so an
// attention block is 3 launches (attn_in -> mha_core -> attn_out) instead of 7 (LN, in_proj, core, out_proj, LN, FF1, FF2):
//
//   attn_in :  qkv = LayerNorm1(x) . Win^T + bin                                                (:141-142, in_proj of nn.MultiheadAttention)
//   attn_out:  h   = x + att . Wo^T + bo ;  y = h + W2 . GELU(W1 . LayerNorm2(h) + b1) + b2     (:142-145)
//
// A workgroup owns T = 32 (or 16, when that is what it takes to give every CU a workgroup) tokens and all C channels.  The token tile lives in LDS ([T][C/4] quads, the quad index XOR-ed with
// (token & 15): the 16 lanes a ds_read_b128 serves together read 16 different tokens at one K position and land on 16 distinct
// 16-byte slots); the weights stream from L2 straight into the A fragments in the implicit-GEMM layout [k_step][Cout][16]
// (conv_igemm.hip), one 16-byte load per lane and fragment, whole groups of K steps prefetched.  Each of the 4 waves computes C/4 output
// channels of all T tokens (FCO = C/64 row fragments x T/16 token fragments), so a GEMM's result goes back to LDS without any cross-wave
// reduction and the chain LN -> GEMM -> GELU -> GEMM -> residual never touches HBM.  Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32
// FMA chains, k ascending as in the stand-alone kernels), LayerNorm two-pass in fp32 as layernorm_kernel.
#include "common.h"
#include "kernels.h"
#include "conv_common.h"

namespace {


// Weight fragments of one group of K steps (GD steps x FCO row fragments, <= 16 x 16-byte loads in flight per lane).  A workgroup
// runs one wave per SIMD, so nothing but the wave's own prefetch distance hides the L2 latency of the weight stream: whole groups
// are requested at once, two register sets, and the first group of the NEXT product is requested before the barrier / LayerNorm /
// epilogue that precedes it.
template <int FCO>
struct WGroup {
    static constexpr int NSTEP = 4 * FCO;                                   // K steps of 16 channels (C = 64 FCO)
    static constexpr int GD = 16 / FCO < NSTEP ? (16 / FCO < 1 ? 1 : 16 / FCO) : NSTEP;     // steps per group: 4, 8, 4, 2 for FCO = 1, 2, 4, 8
    static constexpr int NG = NSTEP / GD;                                   // 1, 1, 4, 16
    f32x4 a[GD][FCO];
    __device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t wr, uint32_t lane_off, uint32_t step_stride, int g) {
#pragma unroll
        for (int d = 0; d < GD; ++d)
#pragma unroll
            for (int i = 0; i < FCO; ++i) a[d][i] = buf_load4(wr, lane_off + (uint32_t)(g * GD + d) * step_stride + (uint32_t)i * 1024u);
    }
};

__device__ __forceinline__ uint32_t w_lane_off(int co_base, int r16, int kq) { return (uint32_t)((co_base + r16) * 16 + kq * 4) * 4u; }

// acc[i][j] = W[co_base + 16 i + r][:] . tile[16 j + c][:]   (A = weights from global, B = tokens from LDS); `g0` holds group 0
template <int FCO, int FPX>
__device__ __forceinline__ void tile_gemm(const f32x4* __restrict__ q, const __amdgpu_buffer_rsrc_t wr, int cout_total, int co_base,
                                          WGroup<FCO>& g0, f32x4 (&acc)[FCO][FPX], int r16, int kq) {
    using G = WGroup<FCO>;
    constexpr int QPT = 16 * FCO;            // quads per token
    const uint32_t lane_off = w_lane_off(co_base, r16, kq);
    const uint32_t step_stride = (uint32_t)cout_total * 64u;
#pragma unroll
    for (int i = 0; i < FCO; ++i)
#pragma unroll
        for (int j = 0; j < FPX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto run = [&](const G& w, int g) {
#pragma unroll
        for (int d = 0; d < G::GD; ++d) {
            f32x4 b[FPX];
#pragma unroll
            for (int j = 0; j < FPX; ++j) b[j] = q[(16 * j + r16) * QPT + (((g * G::GD + d) * 4 + kq) ^ r16)];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < FCO; ++i)
#pragma unroll
                    for (int j = 0; j < FPX; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.a[d][i][k], b[j][k], acc[i][j], 0, 0, 0);
        }
    };
    if (G::NG == 1) {
        run(g0, 0);
    } else {
        G g1;
#pragma unroll 1
        for (int g = 0; g < G::NG; g += 2) {     // NG is even; the group requested past the end is bounds-checked to 0 and unused
            g1.load(wr, lane_off, step_stride, g + 1);
            run(g0, g);
            g0.load(wr, lane_off, step_stride, g + 2);
            run(g1, g + 1);
        }
    }
}

// global [tokens m0 .. m0+TOK-1][C] -> swizzled LDS tile (rows past M are zero)
template <int FCO, int FPX>
__device__ __forceinline__ void tile_load(f32x4* __restrict__ dst, const float* __restrict__ src, int m0, int M) {
    constexpr int QPT = 16 * FCO, TOK = 16 * FPX;
#pragma unroll
    for (int u = 0; u < TOK * QPT / 256; ++u) {
        const int idx = threadIdx.x + 256 * u;
        const int tok = idx / QPT, quad = idx - tok * QPT;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + tok < M) v = reinterpret_cast<const f32x4*>(src)[(size_t)(m0 + tok) * QPT + quad];
        dst[tok * QPT + (quad ^ (tok & 15))] = v;
    }
}

struct LNParams {                       // this lane's gamma / beta quads, requested at kernel start
    f32x4 g[2], b[2];
    template <int FCO>
    __device__ __forceinline__ void load(const float* __restrict__ gamma, const float* __restrict__ beta, int lane) {
        constexpr int QPT = 16 * FCO, LPT = QPT < 64 ? QPT : 64, NQ = QPT / LPT;
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            g[n] = reinterpret_cast<const f32x4*>(gamma)[lane % LPT + n * LPT];
            b[n] = reinterpret_cast<const f32x4*>(beta)[lane % LPT + n * LPT];
        }
    }
};

// LayerNorm over the channels of every token of the tile: src -> dst (both swizzled), two-pass fp32 (norm.hip: layernorm_kernel)
template <int FCO, int FPX>
__device__ __forceinline__ void tile_layernorm(const f32x4* __restrict__ src, f32x4* __restrict__ dst, const LNParams& ln, float eps,
                                               int wave, int lane) {
    constexpr int QPT = 16 * FCO, C = 64 * FCO, TOK = 16 * FPX;
    constexpr int LPT = QPT < 64 ? QPT : 64;       // lanes per token
    constexpr int NQ = QPT / LPT;                  // quads per lane
    constexpr int PAR = 64 / LPT;                  // tokens a wave normalises at once
    const int l = lane % LPT;
#pragma unroll 1
    for (int t0 = wave * (TOK / 4); t0 < (wave + 1) * (TOK / 4); t0 += PAR) {
        const int tok = t0 + lane / LPT;
        f32x4 v[NQ];
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            v[n] = src[tok * QPT + ((l + n * LPT) ^ (tok & 15))];
            s += (v[n][0] + v[n][1]) + (v[n][2] + v[n][3]);
        }
#pragma unroll
        for (int off = LPT / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)C;
        float s2 = 0.f;
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            v[n] = v[n] - mean;
            s2 += (v[n][0] * v[n][0] + v[n][1] * v[n][1]) + (v[n][2] * v[n][2] + v[n][3] * v[n][3]);
        }
#pragma unroll
        for (int off = LPT / 2; off >= 1; off >>= 1) s2 += __shfl_xor(s2, off, 64);
        const float rstd = 1.f / sqrtf(s2 / (float)C + eps);
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const int quad = l + n * LPT;
            dst[tok * QPT + (quad ^ (tok & 15))] = (v[n] * rstd) * ln.g[n] + ln.b[n];
        }
    }
}

struct AttnInArgs {
    const float *x, *ln_g, *ln_b, *w, *bias;       // w: packed [C/16][3C][16], bias [3C]
    float* qkv;                                    // [M][3C]
    int M;
    float eps;
    unsigned w_bytes;
};

template <int FCO, int FPX>
__global__ __launch_bounds__(256) void attn_in_kernel(const AttnInArgs a) {
    constexpr int QPT = 16 * FCO, C = 64 * FCO, TOK = 16 * FPX;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* const P = reinterpret_cast<f32x4*>(smem_raw);
    f32x4* const Q = P + TOK * QPT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * TOK;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w, a.w_bytes);
    const uint32_t stride = (uint32_t)(3 * C) * 64u;
    WGroup<FCO> g0;
    g0.load(wr, w_lane_off(wave * (16 * FCO), r16, kq), stride, 0);         // q rows: in flight during the tile load and LayerNorm
    LNParams ln;
    ln.load<FCO>(a.ln_g, a.ln_b, lane);
    tile_load<FCO, FPX>(P, a.x, m0, a.M);
    f32x4 bias[3][FCO];
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int i = 0; i < FCO; ++i) bias[part][i] = *reinterpret_cast<const f32x4*>(a.bias + part * C + wave * (16 * FCO) + 16 * i + 4 * kq);
    __syncthreads();
    tile_layernorm<FCO, FPX>(P, Q, ln, a.eps, wave, lane);
    __syncthreads();
#pragma unroll
    for (int part = 0; part < 3; ++part) {                 // q, k, v: rows part*C .. part*C + C-1 of in_proj_weight
        const int co_base = part * C + wave * (16 * FCO);
        f32x4 acc[FCO][FPX];
        tile_gemm<FCO, FPX>(Q, wr, 3 * C, co_base, g0, acc, r16, kq);
        if (part < 2) g0.load(wr, w_lane_off(co_base + C, r16, kq), stride, 0);      // next part's first group under this part's stores
#pragma unroll
        for (int j = 0; j < FPX; ++j) {
            const int m = m0 + 16 * j + r16;
            if (m >= a.M) continue;
#pragma unroll
            for (int i = 0; i < FCO; ++i) {
                const int co = co_base + 16 * i + 4 * kq;
                *reinterpret_cast<f32x4*>(a.qkv + (size_t)m * (3 * C) + co) = acc[i][j] + bias[part][i];
            }
        }
    }
}

struct AttnOutArgs {
    const float *att, *x;                          // attention core output and the block input (residual), [M][C]
    const float *wo, *bo, *ln_g, *ln_b, *w1, *b1, *w2, *b2;      // packed [C/16][C][16] weights
    float* out;                                    // [M][C]; may alias x (a workgroup reads its rows of x before it writes them)
    int M;
    float eps;
    unsigned w_bytes;
};

template <int FCO, int FPX>
__global__ __launch_bounds__(256) void attn_out_kernel(const AttnOutArgs a) {
    constexpr int QPT = 16 * FCO, C = 64 * FCO, TOK = 16 * FPX;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* const P = reinterpret_cast<f32x4*>(smem_raw);       // h = x + out_proj(att): LayerNorm2's input and the last residual
    f32x4* const Q = P + TOK * QPT;                            // GEMM operand: att, then LayerNorm2(h), then GELU(FF1)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * TOK;
    const int co_base = wave * (16 * FCO);
    const uint32_t loff = w_lane_off(co_base, r16, kq), stride = (uint32_t)C * 64u;
    const __amdgpu_buffer_rsrc_t wo = make_rsrc(a.wo, a.w_bytes), w1 = make_rsrc(a.w1, a.w_bytes), w2 = make_rsrc(a.w2, a.w_bytes);
    WGroup<FCO> g0;
    g0.load(wo, loff, stride, 0);
    LNParams ln;
    ln.load<FCO>(a.ln_g, a.ln_b, lane);
    tile_load<FCO, FPX>(Q, a.att, m0, a.M);
    f32x4 bo[FCO], b1[FCO], b2[FCO];
#pragma unroll
    for (int i = 0; i < FCO; ++i) {
        bo[i] = *reinterpret_cast<const f32x4*>(a.bo + co_base + 16 * i + 4 * kq);
        b1[i] = *reinterpret_cast<const f32x4*>(a.b1 + co_base + 16 * i + 4 * kq);
        b2[i] = *reinterpret_cast<const f32x4*>(a.b2 + co_base + 16 * i + 4 * kq);
    }
    // residual rows of x: requested now, consumed after the first product
    f32x4 xres[FCO][FPX];
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const int m = m0 + 16 * j + r16;
#pragma unroll
        for (int i = 0; i < FCO; ++i)
            xres[i][j] = m < a.M ? *reinterpret_cast<const f32x4*>(a.x + (size_t)m * C + co_base + 16 * i + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    f32x4 acc[FCO][FPX];
    // ---- h = x + att . Wo^T + bo -> P -----------------------------------------------------------------------------------------
    tile_gemm<FCO, FPX>(Q, wo, C, co_base, g0, acc, r16, kq);
    g0.load(w1, loff, stride, 0);
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const int tok = 16 * j + r16;
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
            const int co = co_base + 16 * i + 4 * kq;
            P[tok * QPT + ((co >> 2) ^ (tok & 15))] = (acc[i][j] + bo[i]) + xres[i][j];
        }
    }
    __syncthreads();                                           // P complete; every wave is done reading att from Q
    tile_layernorm<FCO, FPX>(P, Q, ln, a.eps, wave, lane);
    __syncthreads();
    // ---- GELU(LN2(h) . W1^T + b1) -> Q ----------------------------------------------------------------------------------------
    tile_gemm<FCO, FPX>(Q, w1, C, co_base, g0, acc, r16, kq);
    g0.load(w2, loff, stride, 0);
    __syncthreads();                                           // every wave is done reading LN2(h)
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const int tok = 16 * j + r16;
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
            const int co = co_base + 16 * i + 4 * kq;
            Q[tok * QPT + ((co >> 2) ^ (tok & 15))] = gelu4(acc[i][j] + b1[i]);
        }
    }
    __syncthreads();
    // ---- y = h + FF1out . W2^T + b2 -------------------------------------------------------------------------------------------
    tile_gemm<FCO, FPX>(Q, w2, C, co_base, g0, acc, r16, kq);
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const int tok = 16 * j + r16, m = m0 + tok;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
            const int co = co_base + 16 * i + 4 * kq;
            *reinterpret_cast<f32x4*>(a.out + (size_t)m * C + co) =
                (acc[i][j] + b2[i]) + P[tok * QPT + ((co >> 2) ^ (tok & 15))];
        }
    }
}

inline size_t tile_lds_bytes(int C, int tok) { return (size_t)2 * tok * C * sizeof(float); }
// 32-token tiles when that still gives every CU a workgroup, 16-token tiles below
inline int tile_tokens(int M) { return M >= 256 * 64 ? 32 : 16; }

}  // namespace

int sbgm_attn_tokens_supported(int C) { return C == 64 || C == 128 || C == 256 || C == 512; }

#define SBGM_ATTN_LAUNCH(KERNEL, F, P_)                                                                                       \
    {                                                                                                                         \
        static bool attr = false;                                                                                             \
        if (!attr) { SBGM_HIP(hipFuncSetAttribute((const void*)KERNEL<F, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
        hipLaunchKernelGGL((KERNEL<F, P_>), grid, dim3(256), lds, st, a);                                                     \
    }
#define SBGM_ATTN_SWITCH(KERNEL)                                                                                              \
    switch ((C / 64) * 4 + tok / 16) {                                                                                        \
        case 1 * 4 + 1: SBGM_ATTN_LAUNCH(KERNEL, 1, 1) break;                                                                 \
        case 1 * 4 + 2: SBGM_ATTN_LAUNCH(KERNEL, 1, 2) break;                                                                 \
        case 2 * 4 + 1: SBGM_ATTN_LAUNCH(KERNEL, 2, 1) break;                                                                 \
        case 2 * 4 + 2: SBGM_ATTN_LAUNCH(KERNEL, 2, 2) break;                                                                 \
        case 4 * 4 + 1: SBGM_ATTN_LAUNCH(KERNEL, 4, 1) break;                                                                 \
        case 4 * 4 + 2: SBGM_ATTN_LAUNCH(KERNEL, 4, 2) break;                                                                 \
        case 8 * 4 + 1: SBGM_ATTN_LAUNCH(KERNEL, 8, 1) break;                                                                 \
        case 8 * 4 + 2: SBGM_ATTN_LAUNCH(KERNEL, 8, 2) break;                                                                 \
    }

int sbgm_launch_attn_in(const float* x, const float* ln_g, const float* ln_b, const float* w_packed, const float* bias, float* qkv,
                        int M, int C, float eps, hipStream_t st) {
    SBGM_CHECK(x && ln_g && ln_b && w_packed && bias && qkv, "attn_in: null argument");
    SBGM_CHECK(sbgm_attn_tokens_supported(C), "attn_in: C=%d (supported: 64, 128, 256, 512)", C);
    SBGM_CHECK(M >= 1, "attn_in: M=%d", M);
    AttnInArgs a{x, ln_g, ln_b, w_packed, bias, qkv, M, eps, (unsigned)((size_t)3 * C * C * sizeof(float))};
    const int tok = tile_tokens(M);
    const dim3 grid((M + tok - 1) / tok);
    const size_t lds = tile_lds_bytes(C, tok);
    SBGM_ATTN_SWITCH(attn_in_kernel)
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_attn_out(const float* att, const float* x, const float* wo, const float* bo, const float* ln_g, const float* ln_b,
                         const float* w1, const float* b1, const float* w2, const float* b2, float* out, int M, int C, float eps,
                         hipStream_t st) {
    SBGM_CHECK(att && x && wo && bo && ln_g && ln_b && w1 && b1 && w2 && b2 && out, "attn_out: null argument");
    SBGM_CHECK(sbgm_attn_tokens_supported(C), "attn_out: C=%d (supported: 64, 128, 256, 512)", C);
    SBGM_CHECK(M >= 1, "attn_out: M=%d", M);
    SBGM_CHECK(out != att, "attn_out: out must not alias att");
    AttnOutArgs a{att, x, wo, bo, ln_g, ln_b, w1, b1, w2, b2, out, M, eps, (unsigned)((size_t)C * C * sizeof(float))};
    const int tok = tile_tokens(M);
    const dim3 grid((M + tok - 1) / tok);
    const size_t lds = tile_lds_bytes(C, tok);
    SBGM_ATTN_SWITCH(attn_out_kernel)
    SBGM_LAUNCH_CHECK();
    return 0;
}
