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
// No LDS in this kernel (short sequences: K/V of one (sample, head) are a few KB and stay in L1/L2 across the waves that share
// them); S >= 128 takes mha_core_lds_kernel below.
#include "common.h"
#include "kernels.h"

namespace {

// KSPLIT = 1: one wave per 16-query block, all keys.  KSPLIT = 4: the 4 waves of a workgroup share one query block and take
// the key blocks round-robin (4x shorter dependent load->MFMA->softmax chains at S >= 64); their (max, sum, O^T) partials
// are merged through LDS with the usual online-softmax rescale.
template <int D16, int KSPLIT>   // head dim d <= 16 * D16, d % 4 == 0 (missing quads of the last 16-block are read as zeros)
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
    const int d = C / heads;
    const size_t row_stride = 3 * (size_t)C;
    const float* base = qkv + (size_t)b * S * row_stride + (size_t)h * d;   // q of token 0; k at +C, v at +2C

    // Q fragment (B operand): lane supplies Q[query r16][16*j + 4*kq .. +3], pre-scaled by 1/sqrt(d)
    const int qi = qb * 16 + r16;
    const bool q_ok = qi < S;
    f32x4 qf[D16];
#pragma unroll
    for (int j = 0; j < D16; ++j) {
        qf[j] = (q_ok && 16 * j + 4 * kq < d) ? *reinterpret_cast<const f32x4*>(base + (size_t)qi * row_stride + 16 * j + 4 * kq)
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
            const f32x4 kf = (k_ok && 16 * j + 4 * kq < d) ? *reinterpret_cast<const f32x4*>(base + (size_t)krow * row_stride + C + 16 * j + 4 * kq)
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
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first block: 2^(-inf) = 0
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            st[e] = __builtin_amdgcn_exp2f(st[e] - m_new);   // masked keys: 2^(-inf) = 0
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
                const float vv = (key < S && 16 * j + r16 < d) ? base[(size_t)key * row_stride + 2 * C + 16 * j + r16] : 0.f;
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
        float f0 = __builtin_amdgcn_exp2f(m_run - m_all);
        l_run *= f0;
#pragma unroll
        for (int j = 0; j < D16; ++j) o[j] *= f0;
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const float fw = __builtin_amdgcn_exp2f(ml[(ww * 64 + lane) * 2] - m_all);   // 2^(-inf) = 0 for idle waves
            l_run += ml[(ww * 64 + lane) * 2 + 1] * fw;
#pragma unroll
            for (int j = 0; j < D16; ++j) o[j] += ol[(ww * D16 + j) * 64 + lane] * fw;
        }
    }
    if (q_ok) {
        const float inv = 1.f / l_run;
        float* op = out + ((size_t)b * S + qi) * C + (size_t)h * d;
#pragma unroll
        for (int j = 0; j < D16; ++j)
            if (16 * j + 4 * kq < d) *reinterpret_cast<f32x4*>(op + 16 * j + 4 * kq) = o[j] * inv;
    }
}

// ---- long sequences (S >= 128): K and V of one (sample, head) staged in LDS ----------------------------------------------------
// The register kernel above re-reads K and V once per 16-query block straight from L2 (S/16 times per head; V as strided
// dwords): at S = 256 it fetched 5.4x its qkv tensor.  Here a workgroup owns every `qsplit`-th query block of one (sample, head)
// — up to NQ blocks per wave, their online-softmax state in registers — and walks the keys in chunks of KCH = 128 staged in LDS
// with coalesced 16-byte loads, so K / V are fetched qsplit (2-8) times instead of S/16 times:
//   K chunk  [key][16-dim unit][4 quads], unit stride odd and the quad rotated by key >> 1: the A fragments of S^T = K Q^T are
//            conflict-free ds_read_b128;
//   V chunk  [key][d + 4]: the A operand of O^T = V^T P^T needs 4 keys x 1 dim per lane = 4 ds_read_b32, conflict-free with the
//            row stride = 4 (mod 8) floats (LDS time stays ~5 % of the MFMA time).
constexpr int MHA_KCH = 128, MHA_NQ_MAX = 4;
template <int D16, int MHA_NQ>        // MHA_NQ query blocks per wave, processed together (no branches between their chains)
__global__ __launch_bounds__(256) void mha_core_lds_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B, int S, int C,
                                                           int heads, float scale, int qsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int KU = D16 | 1;                               // 16-float units per key row (odd)
    const int d = C / heads, VS = 16 * D16 + 4;
    f32x4* Kl = reinterpret_cast<f32x4*>(smem_raw);           // [KCH][KU][4]
    float* Vl = reinterpret_cast<float*>(Kl + MHA_KCH * KU * 4);   // [KCH][VS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    int w = blockIdx.x;
    const int qs = w % qsplit; w /= qsplit;
    const int h = w % heads;
    const int b = w / heads;
    const size_t row_stride = 3 * (size_t)C;
    const float* base = qkv + (size_t)b * S * row_stride + (size_t)h * d;

    // this wave's query blocks: qb = qs + qsplit * (wave + 4 n)
    f32x4 qf[MHA_NQ][D16], o[MHA_NQ][D16];
    float m_run[MHA_NQ], l_run[MHA_NQ];
#pragma unroll
    for (int n = 0; n < MHA_NQ; ++n) {
        const int qi = (qs + qsplit * (wave + 4 * n)) * 16 + r16;
        m_run[n] = -INFINITY; l_run[n] = 0.f;
#pragma unroll
        for (int j = 0; j < D16; ++j) {
            o[n][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            qf[n][j] = (qi < S && 16 * j + 4 * kq < d) ? *reinterpret_cast<const f32x4*>(base + (size_t)qi * row_stride + 16 * j + 4 * kq)
                                                       : f32x4{0.f, 0.f, 0.f, 0.f};
            qf[n][j] *= scale;
        }
    }
    constexpr int uq = 4 * D16;                               // quads per (zero-padded) key row
    constexpr int NLD = MHA_KCH * uq / 256;                   // K (and V) quads a thread stages per chunk
    // register-staged chunks: the global loads of chunk c+1 are issued before the MFMAs of chunk c and land while the matrix
    // pipe works (a workgroup is alone on its CU at the usual grid sizes, so nothing else would hide that latency)
    f32x4 rk[NLD], rv[NLD];
    auto chunk_load = [&](int c0) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + 256 * u;
            const int key = i / uq, q4 = i - key * uq;
            const bool ok = c0 + key < S && 4 * q4 < d;
            const float* src = base + (size_t)(c0 + key) * row_stride + 4 * q4;
            rk[u] = ok ? *reinterpret_cast<const f32x4*>(src + C) : f32x4{0.f, 0.f, 0.f, 0.f};
            rv[u] = ok ? *reinterpret_cast<const f32x4*>(src + 2 * C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    chunk_load(0);
    for (int c0 = 0; c0 < S; c0 += MHA_KCH) {
        __syncthreads();                                      // every wave is done with the previous chunk
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + 256 * u;
            const int key = i / uq, q4 = i - key * uq;
            Kl[(key * KU + (q4 >> 2)) * 4 + (((q4 & 3) + (key >> 1)) & 3)] = rk[u];
            *reinterpret_cast<f32x4*>(Vl + key * VS + 4 * q4) = rv[u];
        }
        __syncthreads();
        if (c0 + MHA_KCH < S) chunk_load(c0 + MHA_KCH);
        const int kend = min(MHA_KCH, S - c0);
        // key blocks outermost, this wave's query blocks innermost: one K / V fragment read serves all of them and their
        // QK^T -> softmax -> PV chains are independent, so the latencies of one hide behind the MFMAs of the others
        for (int kl = 0; kl < kend; kl += 16) {
            f32x4 kf[D16];
#pragma unroll
            for (int j = 0; j < D16; ++j) kf[j] = Kl[((kl + r16) * KU + j) * 4 + ((kq + ((kl + r16) >> 1)) & 3)];
            float vv[D16][4];
#pragma unroll
            for (int j = 0; j < D16; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[j][e] = Vl[(kl + 4 * kq + e) * VS + 16 * j + r16];
#pragma unroll
            for (int n = 0; n < MHA_NQ; ++n) {                 // a slot beyond the last query block computes on zero queries (stores are masked)
                f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < D16; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][e], qf[n][j][e], st, 0, 0, 0);
                float mx = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c0 + kl + 4 * kq + e >= S) st[e] = -INFINITY;
                    mx = fmaxf(mx, st[e]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[n], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[n] - m_new);
                float ps = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[e] = __builtin_amdgcn_exp2f(st[e] - m_new);
                    ps += st[e];
                }
                ps += __shfl_xor(ps, 16, 64);
                ps += __shfl_xor(ps, 32, 64);
                l_run[n] = l_run[n] * alpha + ps;
                m_run[n] = m_new;
#pragma unroll
                for (int j = 0; j < D16; ++j) {
                    o[n][j] *= alpha;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[n][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[j][e], st[e], o[n][j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < MHA_NQ; ++n) {
        const int qi = (qs + qsplit * (wave + 4 * n)) * 16 + r16;
        if (qi < S) {
            const float inv = 1.f / l_run[n];
            float* op = out + ((size_t)b * S + qi) * C + (size_t)h * d;
#pragma unroll
            for (int j = 0; j < D16; ++j)
                if (16 * j + 4 * kq < d) *reinterpret_cast<f32x4*>(op + 16 * j + 4 * kq) = o[n][j] * inv;
        }
    }
}

}  // namespace

int sbgm_launch_mha_core(const float* qkv, float* out, int B, int S, int C, int heads, hipStream_t st) {
    SBGM_CHECK(heads > 0 && C % heads == 0, "mha: C=%d not divisible by heads=%d", C, heads);
    const int d = C / heads;
    SBGM_CHECK(d % 4 == 0 && d <= 512, "mha: head dim %d must be a multiple of 4 (<= 512)", d);
    const int d16 = (d + 15) / 16;
    // scores are kept in base-2 units (q pre-scaled by log2(e)/sqrt(d)): softmax = 2^(s - max) / sum, evaluated with the hardware
    // exp2 (1 ulp) — the full-range expf expansion cost as many VALU cycles per key block as its MFMAs
    const float scale = 1.4426950408889634f / sqrtf((float)d);
    static const bool lds_ok = getenv("SBGM_NO_LDS_ATTENTION") == nullptr;
    if (lds_ok && S >= 128 && d16 <= 4) {
        // query blocks of one (sample, head) are dealt to `qsplit` workgroups: one workgroup per CU if possible (every split stages
        // K / V again, so fewer is better for traffic), at most 16 blocks each
        const int qblocks = (S + 15) / 16, bh = B * heads;
        // one workgroup per CU; two from S = 1024 on, where a workgroup's 8 chunks are long enough for a second one to fill its
        // staging phases (measured 134 -> 117 us at B=16, S=1024; at S=256 twice the K/V staging costs more than it hides)
        const int wg_target = S >= 1024 ? 512 : 256;
        int qsplit = std::max((wg_target + bh - 1) / bh, (qblocks + 4 * MHA_NQ_MAX - 1) / (4 * MHA_NQ_MAX));
        qsplit = std::max(1, std::min(qsplit, qblocks));
        const int per_wave = (qblocks + 4 * qsplit - 1) / (4 * qsplit);          // query blocks per wave
        const int nq = per_wave <= 1 ? 1 : per_wave <= 2 ? 2 : 4;
        const size_t lds = (size_t)MHA_KCH * (d16 | 1) * 64 + (size_t)MHA_KCH * (16 * d16 + 4) * 4;
        const dim3 grid(bh * qsplit), block(256);
        int rc = 1;
#define SBGM_MHAL(DD, NN)                                                                                                              \
    if (d16 == DD && nq == NN) {                                                                                                       \
        hipLaunchKernelGGL((mha_core_lds_kernel<DD, NN>), grid, block, lds, st, qkv, out, B, S, C, heads, scale, qsplit);            \
        rc = 0;                                                                                                                        \
    }
        SBGM_MHAL(1, 1) SBGM_MHAL(1, 2) SBGM_MHAL(1, 4) SBGM_MHAL(2, 1) SBGM_MHAL(2, 2) SBGM_MHAL(2, 4)
        SBGM_MHAL(3, 1) SBGM_MHAL(3, 2) SBGM_MHAL(3, 4) SBGM_MHAL(4, 1) SBGM_MHAL(4, 2) SBGM_MHAL(4, 4)
#undef SBGM_MHAL
        SBGM_CHECK(rc == 0, "mha: no LDS kernel for head dim %d", d);
        SBGM_LAUNCH_CHECK();
        return 0;
    }
    const int waves = B * heads * ((S + 15) / 16);
    const bool ksplit = S >= 64 && d <= 256;             // >= 4 key blocks: spread them over the 4 waves of a workgroup
    const dim3 grid(ksplit ? waves : (waves + 3) / 4), block(256);
    const size_t lds = ksplit ? (size_t)4 * 64 * 2 * 4 + (size_t)4 * d16 * 64 * 16 : 0;
    switch (d16) {
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
