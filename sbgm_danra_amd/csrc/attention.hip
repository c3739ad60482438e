// K14: multi-head self-attention core, softmax(Q K^T / sqrt(d)) V per (sample, head), fp32.
// reference sbgm/score_unet.py:127,142 (nn.MultiheadAttention, batch_first, dropout 0): this kernel is the
// part between in_proj and out_proj; the projections run on the implicit-GEMM kernel as 1x1 convolutions.
//
// One wave owns 16 query rows of one (sample, head) and walks the keys in blocks of 16 with an online softmax.
// Both contractions run on v_mfma_f32_16x16x4_f32:
//   S^T (keys x queries) = K . Q^T      A = K rows (16 B per lane, straight from the qkv buffer), B = Q rows (registers)
//   O^T (d    x queries) = V^T . P^T    A = V^T (dword loads), B = P^T = the S^T accumulator itself: the MFMA C/D map
//                                       (col = lane&15, row = 4*(lane>>4)+reg) is already the B-operand map for key
//                                       k = 4*(lane>>4)+reg, so P never leaves registers and V is indexed to match.
// Queries sit on lane&15 in both products, so the running max / sum / rescale are lane-local, and the softmax
// reductions over keys are 3 in-register ops + 2 wavefront shuffles (xor 16, xor 32).
// No LDS: K/V of one (sample, head) are at most S*d*8 bytes and stay in L1/L2 across the waves that share them.
#include "common.h"
#include "kernels.h"

namespace {

// KSPLIT = 1: one wave per 16-query block, all keys.  KSPLIT = 4: the 4 waves of a workgroup share one query block and take
// the key blocks round-robin (4x shorter dependent load->MFMA->softmax chains at S >= 64); their (max, sum, O^T) partials
// are merged through LDS with the usual online-softmax rescale.
template <int D16, int KSPLIT>   // head dim = 16 * D16
__global__ __launch_bounds__(256) void mha_core_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B,
                                                       int S, int C, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int qblocks = (S + 15) >> 4;
    int w = KSPLIT == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    if (w >= B * heads * qblocks) return;        // whole workgroup when KSPLIT > 1
    const int qb = w % qblocks; w /= qblocks;
    const int h = w % heads;
    const int b = w / heads;
    const int d = 16 * D16;
    const size_t row_stride = 3 * (size_t)C;
    const float* base = qkv + (size_t)b * S * row_stride + (size_t)h * d;   // q of token 0; k at +C, v at +2C

    // Q fragment (B operand): lane supplies Q[query r16][16*j + 4*kq .. +3], pre-scaled by 1/sqrt(d)
    const int qi = qb * 16 + r16;
    const bool q_ok = qi < S;
    f32x4 qf[D16];
#pragma unroll
    for (int j = 0; j < D16; ++j) {
        qf[j] = q_ok ? *reinterpret_cast<const f32x4*>(base + (size_t)qi * row_stride + 16 * j + 4 * kq)
                     : f32x4{0.f, 0.f, 0.f, 0.f};
        qf[j] *= scale;
    }

    f32x4 o[D16];      // O^T accumulators: rows = head-dim 16*j + 4*kq + reg, col = query r16
#pragma unroll
    for (int j = 0; j < D16; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = (KSPLIT == 1 ? 0 : wave * 16); k0 < S; k0 += 16 * KSPLIT) {
        // ---- S^T block: rows = keys k0..k0+15, cols = queries --------------------------------------------
        const int krow = k0 + r16;                       // A operand row this lane loads
        const bool k_ok = krow < S;
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < D16; ++j) {
            const f32x4 kf = k_ok ? *reinterpret_cast<const f32x4*>(base + (size_t)krow * row_stride + C + 16 * j + 4 * kq)
                                  : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[j][e], st, 0, 0, 0);
        }
        // lane now holds scores of query r16 against keys k0 + 4*kq + reg
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (k0 + 4 * kq + e >= S) st[e] = -INFINITY;
            mx = fmaxf(mx, st[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);            // finite: every block has at least one valid key
        const float alpha = expf(m_run - m_new);         // first block: exp(-inf) = 0
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            st[e] = expf(st[e] - m_new);                 // masked keys: exp(-inf) = 0
            ps += st[e];
        }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // ---- O^T = alpha * O^T + V^T . P^T ------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < D16; ++j) {
            o[j] *= alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // A operand: V^T[row = head-dim 16*j + r16][k = key k0 + 4*kq + e]
                const int key = k0 + 4 * kq + e;
                const float vv = key < S ? base[(size_t)key * row_stride + 2 * C + 16 * j + r16] : 0.f;
                o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, st[e], o[j], 0, 0, 0);
            }
        }
    }
    if (KSPLIT > 1) {
        // merge the waves' partial softmax states; a wave that saw no key block carries m = -inf, l = 0, O = 0
        float* ml = reinterpret_cast<float*>(smem_raw);                       // [wave][64 lanes][2]
        f32x4* ol = reinterpret_cast<f32x4*>(smem_raw + 4 * 64 * 2 * 4);      // [wave][D16][64 lanes]
        ml[(wave * 64 + lane) * 2] = m_run;
        ml[(wave * 64 + lane) * 2 + 1] = l_run;
#pragma unroll
        for (int j = 0; j < D16; ++j) ol[(wave * D16 + j) * 64 + lane] = o[j];
        __syncthreads();
        if (wave != 0) return;
        float m_all = m_run;
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) m_all = fmaxf(m_all, ml[(ww * 64 + lane) * 2]);
        float f0 = expf(m_run - m_all);
        l_run *= f0;
#pragma unroll
        for (int j = 0; j < D16; ++j) o[j] *= f0;
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const float fw = expf(ml[(ww * 64 + lane) * 2] - m_all);          // exp(-inf) = 0 for idle waves
            l_run += ml[(ww * 64 + lane) * 2 + 1] * fw;
#pragma unroll
            for (int j = 0; j < D16; ++j) o[j] += ol[(ww * D16 + j) * 64 + lane] * fw;
        }
    }
    if (q_ok) {
        const float inv = 1.f / l_run;
        float* op = out + ((size_t)b * S + qi) * C + (size_t)h * d;
#pragma unroll
        for (int j = 0; j < D16; ++j) *reinterpret_cast<f32x4*>(op + 16 * j + 4 * kq) = o[j] * inv;
    }
}

}  // namespace

int sbgm_launch_mha_core(const float* qkv, float* out, int B, int S, int C, int heads, hipStream_t st) {
    SBGM_CHECK(heads > 0 && C % heads == 0, "mha: C=%d not divisible by heads=%d", C, heads);
    const int d = C / heads;
    SBGM_CHECK(d % 16 == 0 && d <= 512, "mha: head dim %d must be a multiple of 16 (<= 512)", d);
    const int waves = B * heads * ((S + 15) / 16);
    const bool ksplit = S >= 64 && d <= 256;             // >= 4 key blocks: spread them over the 4 waves of a workgroup
    const dim3 grid(ksplit ? waves : (waves + 3) / 4), block(256);
    const size_t lds = ksplit ? (size_t)4 * 64 * 2 * 4 + (size_t)4 * (d / 16) * 64 * 16 : 0;
    const float scale = 1.0f / sqrtf((float)d);
    switch (d / 16) {
#define SBGM_MHA(N)                                                                                                      \
    case N:                                                                                                              \
        if (ksplit) hipLaunchKernelGGL((mha_core_kernel<N, 4>), grid, block, lds, st, qkv, out, B, S, C, heads, scale);    \
        else hipLaunchKernelGGL((mha_core_kernel<N, 1>), grid, block, 0, st, qkv, out, B, S, C, heads, scale);             \
        break;
        SBGM_MHA(1) SBGM_MHA(2) SBGM_MHA(3) SBGM_MHA(4) SBGM_MHA(6) SBGM_MHA(8) SBGM_MHA(12) SBGM_MHA(16)
#undef SBGM_MHA
        case 32: hipLaunchKernelGGL((mha_core_kernel<32, 1>), dim3((waves + 3) / 4), block, 0, st, qkv, out, B, S, C, heads, scale); break;
        default: SBGM_CHECK(false, "mha: head dim %d not instantiated", d);
    }
    SBGM_LAUNCH_CHECK();
    return 0;
}
