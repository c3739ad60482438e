// The linear layers of ImageSelfAttention over tokens (reference sbgm/score_unet.py:127-134: mha.in_proj / out_proj, ff.0, ff.2;
// :141-145: the two LayerNorms, the two residual adds, the GELU), one launch per nn.Linear with the element-wise neighbours fused:
//     out = act( LayerNorm?(x) @ W^T + bias ) [+ res]
// The wave-level implicit-GEMM kernel (conv_igemm.hip) ran these 16 small products at 0.2-0.3 of the fp32 MFMA peak (every wave
// re-fetched its token fragments from L2, 5-8 us of prologue per launch) and the LayerNorms were separate 5 us launches.  Here a
// workgroup stages its 16 or 32 tokens ONCE in LDS (coalesced 16-byte loads, rows padded by 32 B: conflict-free ds_read_b128
// B fragments), normalises them in place when the layer starts with a LayerNorm (one wave per token, two-pass, wavefront
// shuffles — the same arithmetic as layernorm_kernel), and its 4 waves sweep K with the packed weight fragments streamed from
// L2 through a register double buffer (v_mfma_f32_16x16x4_f32, A = weights so a lane ends with 4 consecutive output channels
// of one token: float4 epilogue).  Every 64*FCO-channel slice of a token tile is its own workgroup; the slices of one tile run
// back to back on one XCD, so the tile is fetched into that L2 once.
#include "common.h"
#include "kernels.h"
#include "conv_common.h"

namespace {

template <int FCO, int FPX, bool LN>
__global__ __launch_bounds__(256) void token_gemm_kernel(const TokenGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TT = 16 * FPX;                 // tokens per workgroup
    constexpr int NACC = (FCO * FPX == 1) ? 2 : 1;   // a lone accumulator would serialise on the 40-cycle MFMA latency: split K in two
    float* xt = reinterpret_cast<float*>(smem_raw);
    const int KS = p.K + 8;                      // row stride (floats): slot = 2*token + quad (mod 16) -> 16 distinct banks per group
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    int t = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int n_co = p.N / (64 * FCO);
    const int co_tile = t % n_co, m_tile = t / n_co;
    const int m0 = m_tile * TT, co0 = co_tile * 64 * FCO + wave * 16 * FCO;

    // ---- stage the token tile ---------------------------------------------------------------------------------------------------
    const int kq4 = p.K >> 2;
    for (int i = tid; i < TT * kq4; i += 256) {
        const int r = i / kq4, c4 = i - r * kq4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + r < p.M) v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(m0 + r) * p.K + c4 * 4);
        *reinterpret_cast<f32x4*>(xt + r * KS + c4 * 4) = v;
    }
    __syncthreads();
    if (LN) {                                    // nn.LayerNorm(K), eps inside the square root, biased variance (score_unet.py:128-129)
        const float inv_k = 1.f / (float)p.K;
        for (int r = wave; r < TT; r += 4) {
            float* row = xt + r * KS;
            float s = 0.f;
            for (int c = lane; c < p.K; c += 64) s += row[c];
            const float mean = wave_sum(s) * inv_k;
            float v = 0.f;
            for (int c = lane; c < p.K; c += 64) { const float d = row[c] - mean; v += d * d; }
            const float rstd = 1.f / sqrtf(wave_sum(v) * inv_k + p.ln_eps);
            for (int c = lane; c < p.K; c += 64) row[c] = (row[c] - mean) * rstd * p.ln_g[c] + p.ln_b[c];
        }
        __syncthreads();
    }

    // ---- K sweep ------------------------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.wp, p.w_bytes);
    f32x4 acc[NACC][FCO][FPX];
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int i = 0; i < FCO; ++i)
#pragma unroll
            for (int j = 0; j < FPX; ++j) acc[n][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = p.K >> 4;
    f32x4 a[2][FCO];
    auto load_a = [&](int s, f32x4* dst) {
#pragma unroll
        for (int i = 0; i < FCO; ++i) dst[i] = buf_load4(wr, (uint32_t)(((s * p.N + co0 + 16 * i + r16) * 16 + 4 * kq) * 4));
    };
    load_a(0, a[0]);
    for (int s = 0; s < nsteps; s += 2) {        // K is a multiple of 32 on this path (C in {128, 256, 512, ...})
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ss = s + h;
            if (ss + 1 < nsteps) load_a(ss + 1, a[h ^ 1]);
            f32x4 b[FPX];
#pragma unroll
            for (int j = 0; j < FPX; ++j) b[j] = *reinterpret_cast<const f32x4*>(xt + (16 * j + r16) * KS + 16 * ss + 4 * kq);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < FCO; ++i)
#pragma unroll
                    for (int j = 0; j < FPX; ++j)
                        acc[NACC == 2 ? h : 0][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][i][k], b[j][k], acc[NACC == 2 ? h : 0][i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: bias -> activation -> residual ----------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const int m = m0 + 16 * j + r16;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
            const int co = co0 + 16 * i + 4 * kq;
            f32x4 v = acc[0][i][j];
            if (NACC == 2) v += acc[NACC - 1][i][j];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + co);
            if (p.act == SBGM_ACT_GELU) v = gelu4(v);
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.N + co);
            *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.N + co) = v;
        }
    }
}

}  // namespace

// 0 when the shape is outside this kernel's domain (the caller then uses the general implicit-GEMM path)
int sbgm_token_gemm_supported(int M, int K, int N) {
    return M >= 1 && K % 32 == 0 && K >= 32 && N % 64 == 0 && (size_t)32 * (K + 8) * 4 <= 150 * 1024;
}

int sbgm_launch_token_gemm(TokenGemmParams p, hipStream_t st) {
    SBGM_CHECK(sbgm_token_gemm_supported(p.M, p.K, p.N), "token_gemm: M=%d K=%d N=%d unsupported (K %% 32, N %% 64)", p.M, p.K, p.N);
    SBGM_CHECK(p.x && p.wp && p.out, "token_gemm: null tensor");
    SBGM_CHECK((p.ln_g == nullptr) == (p.ln_b == nullptr), "token_gemm: LayerNorm needs gamma and beta");
    SBGM_CHECK(p.act == SBGM_ACT_NONE || p.act == SBGM_ACT_GELU, "token_gemm: act must be none or gelu");
    SBGM_CHECK((size_t)p.K * p.N * 4 < (1ull << 31), "token_gemm: weights exceed the 2 GiB buffer window");
    p.w_bytes = (uint32_t)((size_t)p.K * p.N * 4);
    // largest tile that still gives the chip >= 256 workgroups; 64-channel x 16-token tiles otherwise
    int fco = 1, fpx = 1;
    const int cand[4][2] = {{2, 2}, {1, 2}, {2, 1}, {1, 1}};
    for (auto& c : cand) {
        if (p.N % (64 * c[0])) continue;
        const long wgs = (long)((p.M + 16 * c[1] - 1) / (16 * c[1])) * (p.N / (64 * c[0]));
        if (wgs >= 256 || (c[0] == 1 && c[1] == 1)) { fco = c[0]; fpx = c[1]; break; }
    }
    const int wgs = ((p.M + 16 * fpx - 1) / (16 * fpx)) * (p.N / (64 * fco));
    const size_t lds = (size_t)16 * fpx * (p.K + 8) * 4;
    const bool ln = p.ln_g != nullptr;
    int rc = 1;
#define SBGM_TG(FC, FP, LNV)                                                                                                   \
    if (fco == FC && fpx == FP && ln == LNV) {                                                                                 \
        if (lds > 64 * 1024)                                                                                                   \
            SBGM_HIP(hipFuncSetAttribute((const void*)token_gemm_kernel<FC, FP, LNV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((token_gemm_kernel<FC, FP, LNV>), dim3(wgs), dim3(256), lds, st, p);                              \
        rc = 0;                                                                                                                \
    }
    SBGM_TG(2, 2, false) SBGM_TG(2, 2, true) SBGM_TG(1, 2, false) SBGM_TG(1, 2, true) SBGM_TG(2, 1, false) SBGM_TG(2, 1, true)
    SBGM_TG(1, 1, false) SBGM_TG(1, 1, true)
#undef SBGM_TG
    SBGM_CHECK(rc == 0, "token_gemm: no kernel for tile %dx%d", fco, fpx);
    SBGM_LAUNCH_CHECK();
    return 0;
}
