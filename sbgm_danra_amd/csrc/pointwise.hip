// Streaming (HBM-bound) kernels around the convolutions: layout packing, bilinear x2 upsample, BatchNorm
// folding, the Gaussian-Fourier time embedding + time projections, and the single-output-channel final conv.
#include "common.h"
#include "kernels.h"

namespace {

// ---- K4: channel concat + NCHW -> NHWC (zero-padded to Cs channels) --------------------------------------
// reference sbgm/score_unet.py:273-291 (torch.cat of x, lsm, topo, cond_img along C)
// One thread per (pixel, channel-quad); reads are coalesced along W per source plane, writes are 16 B.
__global__ __launch_bounds__(256) void pack_input_kernel(PackSrc src, float* __restrict__ dst, int B, int HW, int Cs) {
    const int cq = Cs >> 2;
    const size_t total = (size_t)B * HW * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        const size_t pix = i / cq;            // b*HW + p
        const int b = (int)(pix / HW);
        const int p = (int)(pix - (size_t)b * HW);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int c = q * 4 + e;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < src.n) {
                    if (c >= 0 && c < src.ch[s]) v[e] = src.ptr[s][((size_t)b * src.ch[s] + c) * HW + p];
                    c -= src.ch[s];
                }
            }
        }
        *reinterpret_cast<f32x4*>(dst + pix * Cs + q * 4) = v;
    }
}

// Tiled transposes between NCHW and NHWC through LDS (32 pixels x 32 channels per tile, +1 padding).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int HW, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? src[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (c < C && p < HW) dst[((size_t)b * HW + p) * C + c] = tile[tx][r];
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int HW, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (c < C && p < HW) ? src[((size_t)b * HW + p) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) dst[((size_t)b * C + c) * HW + p] = tile[tx][r];
    }
}

// ---- K16: bilinear x2, align_corners=False (nn.Upsample, reference score_unet.py:467,583) ------------
// out(2i)   = .25*in(i-1) + .75*in(i)   (i=0: in(0));   out(2i+1) = .75*in(i) + .25*in(i+1)  (clamped)
// evaluated as PyTorch does: w0h*(w0w*v00 + w1w*v01) + w1h*(w0w*v10 + w1w*v11).
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                         int H, int W, int C) {
    const int cq = C >> 2, OH = 2 * H, OW = 2 * W;
    const size_t total = (size_t)B * OH * OW * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        size_t r = i / cq;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        // source index = max(0, (o + 0.5) * 0.5 - 0.5)
        const float sy = fmaxf(0.f, (oy + 0.5f) * 0.5f - 0.5f), sx = fmaxf(0.f, (ox + 0.5f) * 0.5f - 0.5f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* base = x + (size_t)b * H * W * C + q * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0) * C);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x1) * C);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x0) * C);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x1) * C);
        const f32x4 o = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        *reinterpret_cast<f32x4*>(y + i * 4) = o;
    }
}

// ---- nn.Upsample(scale_factor = s, bilinear, align_corners=False) for any integer s >= 1: the DecoderBlock signature's
// `upsample_scale` (reference score_unet.py:420, :467); the Decoder itself only ever uses 2 (kernel above, and fused into the
// consumer convolution's load path on the sampling side).  source = max(0, (o + 0.5) / s - 0.5), as ATen's
// area_pixel_compute_source_index with the given scale factor.
__device__ __forceinline__ void bilinear_src(int o, float rs, int n, int& i0, int& i1, float& l) {
    const float sp = fmaxf(0.f, ((float)o + 0.5f) * rs - 0.5f);
    i0 = min((int)sp, n - 1);
    i1 = min(i0 + 1, n - 1);
    l = sp - (float)i0;
}
__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W,
                                                                int C, int s, float rs) {
    const int cq = C >> 2, OH = s * H, OW = s * W;
    const size_t total = (size_t)B * OH * OW * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        size_t r = i / cq;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(oy, rs, H, y0, y1, ly);
        bilinear_src(ox, rs, W, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* base = x + (size_t)b * H * W * C + q * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0) * C);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x1) * C);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x0) * C);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x1) * C);
        *reinterpret_cast<f32x4*>(y + i * 4) = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
}
// backward as a gather: an input pixel collects from the output rows / columns whose two taps include it (no atomics, fixed order)
__global__ __launch_bounds__(256) void upsample_bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H,
                                                                    int W, int C, int s, float rs) {
    const int cq = C >> 2, OH = s * H, OW = s * W;
    const size_t total = (size_t)B * H * W * cq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % cq);
        size_t r = i / cq;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* base = dy + (size_t)b * OH * OW * C + q * 4;
        for (int oy = max(0, s * (iy - 1)); oy < min(OH, s * (iy + 2)); ++oy) {
            int y0, y1;
            float ly;
            bilinear_src(oy, rs, H, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = max(0, s * (ix - 1)); ox < min(OW, s * (ix + 2)); ++ox) {
                int x0, x1;
                float lx;
                bilinear_src(ox, rs, W, x0, x1, lx);
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx != 0.f) acc += (wy * wx) * *reinterpret_cast<const f32x4*>(base + ((size_t)oy * OW + ox) * C);
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
    }
}

// ---- K10 (eval): fold BatchNorm2d running statistics into a per-channel scale / bias ------------------
__global__ void bn_fold_kernel(const float* g, const float* be, const float* mu, const float* var, float eps,
                               float* scale, float* bias, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float s = g[c] / sqrtf(var[c] + eps);
        scale[c] = s;
        bias[c] = be[c] - mu[c] * s;
    }
}

// ---- K1 + K3: Gaussian-Fourier features, optional label-embedding add, SiLU ----------------------------
// reference score_unet.py:41-45 (x*W*2pi in that association), :301-308, and the SiLU that opens every
// time-projection nn.Sequential (:377-381, :501-504).  sinf/cosf are the full-range-reduction versions
// (|2 pi t W| reaches several hundred): never compile this file with fast-math.
__global__ __launch_bounds__(256) void time_embed_kernel(TimeEmbedArgs a) {
    const int half = a.D >> 1;
    const int total = a.n_emb * a.B * a.D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int d = i % a.D;
        const int b = (i / a.D) % a.B;
        const int g = i / (a.D * a.B);
        const float w = a.freqs[g][d < half ? d : d - half];
        const float proj = (a.t[b] * w) * 6.283185307179586f;
        float e = d < half ? sinf(proj) : cosf(proj);
        if (g == 0 && a.y != nullptr) e += a.label_emb[(size_t)a.y[b] * a.D + d];
        if (a.emb_raw) a.emb_raw[i] = e;                 // kept for the backward pass (SiLU')
        a.emb_ws[i] = e / (1.f + expf(-e));
    }
}

// ---- K2: all time projections of one forward in one launch ---------------------------------------------------------
// One wave per (projection, output channel, group of 8 samples): the weight row sits in registers, the 8 dot
// products are accumulated with independent loads and reduced with interleaved wavefront butterflies.
constexpr int TP_BG = 8;
__global__ __launch_bounds__(256) void time_proj_kernel(TimeEmbedArgs a, int total_ch, int bgroups) {
    const int lane = threadIdx.x & 63;
    const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wid >= total_ch * bgroups) return;
    const int bg = wid % bgroups;
    int c = wid / bgroups, pi = 0;
    for (; pi < a.n_proj; ++pi) {
        if (c < a.proj[pi].ch) break;
        c -= a.proj[pi].ch;
    }
    const TimeProj& pr = a.proj[pi];
    const float* w = pr.weight + (size_t)c * a.D;
    float acc[TP_BG];
#pragma unroll
    for (int j = 0; j < TP_BG; ++j) acc[j] = 0.f;
    for (int d = lane; d < a.D; d += 64) {
        const float wv = w[d];
#pragma unroll
        for (int j = 0; j < TP_BG; ++j) {
            const int b = min(bg * TP_BG + j, a.B - 1);
            acc[j] = fmaf(wv, a.emb_ws[((size_t)pr.emb * a.B + b) * a.D + d], acc[j]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < TP_BG; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
    if (lane < TP_BG && bg * TP_BG + lane < a.B) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < TP_BG; ++j) v = (lane == j) ? acc[j] : v;
        pr.out[(size_t)(bg * TP_BG + lane) * pr.ch + c] = v + pr.bias[c];
    }
}

// ---- K19(final) + K22: 3x3, pad 1, C -> 1 output channel, then divide by sigma(t) ------------------------
// reference score_unet.py:489 (final_layer.conv), :876-877 and marginal_prob_std :881-897.
// HBM/L2-bound.  16 lanes share a strip of 8 consecutive output pixels of one row; each lane owns a float4 channel
// slice (C = 64 -> exactly one).  A loaded input column (3 rows x 1 pixel) feeds up to 3 outputs, so a strip needs
// 30 float4 loads per lane instead of 72; the 16-lane partial sums are combined with wavefront shuffles.
__global__ __launch_bounds__(256) void conv3x3_cout1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ t,
                                                            float sigma, float* __restrict__ out, int B, int H, int W,
                                                            int C) {
    constexpr int SW = 8;
    const int sub = threadIdx.x & 15;
    const int strips_per_row = (W + SW - 1) / SW;
    const size_t nstrips = (size_t)B * H * strips_per_row;
    const size_t gstride = (size_t)gridDim.x * (blockDim.x >> 4);
    for (size_t sidx = (size_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);; sidx += gstride) {
        const bool live = sidx < nstrips;
        if (__all(!live)) break;                     // whole waves leave together (shuffles below need all lanes)
        float acc[SW];
#pragma unroll
        for (int o = 0; o < SW; ++o) acc[o] = 0.f;
        int b = 0, oy = 0, ox0 = 0;
        if (live) {
            const int sr = (int)(sidx % strips_per_row);
            const size_t row = sidx / strips_per_row;
            oy = (int)(row % H);
            b = (int)(row / H);
            ox0 = sr * SW;
            for (int c = sub * 4; c < C; c += 64) {
                f32x4 wv[3][3];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) wv[kh][kw] = *reinterpret_cast<const f32x4*>(w + (kh * 3 + kw) * C + c);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int iy = oy + kh - 1;
                    if ((unsigned)iy >= (unsigned)H) continue;
                    const float* rowp = x + (((size_t)b * H + iy) * W) * C + c;
#pragma unroll
                    for (int j = 0; j < SW + 2; ++j) {       // input column ox0 - 1 + j
                        const int ix = ox0 - 1 + j;
                        if ((unsigned)ix >= (unsigned)W) continue;
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(rowp + (size_t)ix * C);
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {     // contributes to output o = j - kw
                            const int o = j - kw;
                            if (o < 0 || o >= SW) continue;
                            const f32x4 ww = wv[kh][kw];
                            acc[o] = fmaf(xv[0], ww[0], acc[o]);
                            acc[o] = fmaf(xv[1], ww[1], acc[o]);
                            acc[o] = fmaf(xv[2], ww[2], acc[o]);
                            acc[o] = fmaf(xv[3], ww[3], acc[o]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < SW; ++o) {
#pragma unroll
            for (int sft = 8; sft > 0; sft >>= 1) acc[o] += __shfl_xor(acc[o], sft, 64);
        }
        if (live && sub < SW && ox0 + sub < W) {
            float v = 0.f;
#pragma unroll
            for (int o = 0; o < SW; ++o) v = (sub == o) ? acc[o] : v;
            v += bias[0];
            if (t != nullptr) {
                const float ls = logf(sigma);
                const float var = (expf((2.f * t[b]) * ls) - 1.f) / (2.f * ls);
                v /= fmaxf(sqrtf(var), 1e-5f);
            }
            out[((size_t)b * H + oy) * W + ox0 + sub] = v;
        }
    }
}

// Finish of the fused final block: 9-point gather over the planar per-tap sums written by the conv_up epilogue.
__global__ __launch_bounds__(256) void tap_stencil_kernel(const float* __restrict__ d, const float* __restrict__ bias,
                                                          const float* __restrict__ t, float sigma, float* __restrict__ out,
                                                          int B, int H, int W, int parts) {
    const size_t M = (size_t)B * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int b = (int)(i / ((size_t)W * H));
        float v = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = y + kh - 1;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = x + kw - 1;
                if ((unsigned)ix >= (unsigned)W) continue;
                for (int pt = 0; pt < parts; ++pt) v += d[(size_t)(pt * 9 + kh * 3 + kw) * M + ((size_t)b * H + iy) * W + ix];
            }
        }
        v += bias[0];
        if (t != nullptr) {
            const float ls = logf(sigma);
            const float var = (expf((2.f * t[b]) * ls) - 1.f) / (2.f * ls);
            v /= fmaxf(sqrtf(var), 1e-5f);
        }
        out[i] = v;
    }
}

__global__ void pack_cout1_weight_kernel(const float* w, float* wp, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // wp[tap][c] = w[0][c][kh][kw]
    if (i < 9 * C) {
        const int tap = i / C, c = i - tap * C;
        wp[i] = w[c * 9 + tap];
    }
}

__global__ __launch_bounds__(256) void act_kernel(float* x, size_t n4, int act) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sbgm_act(v[e], act);
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

// grid cap measured 512..8192 on the sampling step: 2048 blocks is best
inline int stream_blocks(size_t work_items) { return (int)std::min<size_t>((work_items + 255) / 256, 2048); }

}  // namespace

namespace {
__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ p, size_t n_dwords) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dwords; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ __launch_bounds__(256) void zero4_kernel(i32x4* __restrict__ p, size_t n_quads) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += (size_t)gridDim.x * blockDim.x) p[i] = i32x4{0, 0, 0, 0};
}
}  // namespace

int sbgm_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    SBGM_CHECK(p != nullptr && bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0, "zero_async: %zu bytes at %p (need 4-byte granularity)", bytes, p);
    if (bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0)
        hipLaunchKernelGGL(zero4_kernel, dim3(stream_blocks(bytes / 16)), dim3(256), 0, st, static_cast<i32x4*>(p), bytes / 16);
    else
        hipLaunchKernelGGL(zero_kernel, dim3(stream_blocks(bytes / 4)), dim3(256), 0, st, static_cast<uint32_t*>(p), bytes / 4);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_pack_input(const PackSrc& src, float* dst, int B, int H, int W, int Cs, hipStream_t st) {
    int ctot = 0;
    for (int i = 0; i < src.n; ++i) ctot += src.ch[i];
    SBGM_CHECK(src.n >= 1 && src.n <= 4, "pack_input: %d sources (1..4 supported)", src.n);
    SBGM_CHECK(Cs % 4 == 0 && ctot <= Cs, "pack_input: %d channels do not fit padded width %d", ctot, Cs);
    hipLaunchKernelGGL(pack_input_kernel, dim3(stream_blocks((size_t)B * H * W * (Cs / 4))), dim3(256), 0, st, src, dst, B,
                       H * W, Cs);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_nchw_to_nhwc(const float* src, float* dst, int B, int H, int W, int C, hipStream_t st) {
    dim3 grid((H * W + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, st, src, dst, H * W, C);
    SBGM_LAUNCH_CHECK();
    return 0;
}
int sbgm_launch_nhwc_to_nchw(const float* src, float* dst, int B, int H, int W, int C, hipStream_t st) {
    dim3 grid((H * W + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, st, src, dst, H * W, C);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_upsample2x(const float* x, float* y, int B, int H, int W, int C, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0, "upsample2x: C=%d must be a multiple of 4", C);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(stream_blocks((size_t)B * H * W * C)), dim3(256), 0, st, x, y, B, H, W, C);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_upsample_bilinear(const float* x, float* y, int B, int H, int W, int C, int scale, int backward, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0 && scale >= 1 && scale <= 16, "upsample_bilinear: C=%d must be a multiple of 4, scale=%d in 1..16", C, scale);
    SBGM_CHECK((size_t)B * H * W * scale * scale * C < ((size_t)1 << 40), "upsample_bilinear: tensor too large");
    const float rs = (float)(1.0 / (double)scale);
    if (backward)      // x = dy [B][sH][sW][C], y = dx [B][H][W][C]
        hipLaunchKernelGGL(upsample_bilinear_bwd_kernel, dim3(stream_blocks((size_t)B * H * W * C)), dim3(256), 0, st, x, y, B, H, W, C, scale, rs);
    else
        hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(stream_blocks((size_t)B * H * W * scale * scale * C)), dim3(256), 0, st, x, y, B, H, W, C, scale, rs);
    SBGM_LAUNCH_CHECK();
    return 0;
}

// ---- ConvTranspose2d(k=2, s=2) support (decoder ablation path, reference score_unet.py:470-475, :589) -----------------------
// The transposed convolution is 4 independent 1x1 convolutions (one per output phase (dy,dx)); the engine runs them as ONE
// 1x1 implicit GEMM with 4C output channels ordered (dy, dx, co) and these two permutations move between that
// "depth" layout [B][H][W][4C] and the upsampled "space" layout [B][2H][2W][C].
namespace {
__global__ __launch_bounds__(256) void depth_space2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                                           int W, int C, int to_space, size_t total4) {
    const int c4n = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        // i enumerates float4 slots of the SPACE tensor [B][2H][2W][C/4]
        const int c4 = (int)(i % c4n);
        size_t r = i / c4n;
        const int X = (int)(r % (2 * W)); r /= 2 * W;
        const int Y = (int)(r % (2 * H));
        const int b = (int)(r / (2 * H));
        const size_t d = ((((size_t)b * H + (Y >> 1)) * W + (X >> 1)) * 4 + ((Y & 1) * 2 + (X & 1))) * c4n + c4;
        if (to_space) reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(in)[d];
        else reinterpret_cast<f32x4*>(out)[d] = reinterpret_cast<const f32x4*>(in)[i];
    }
}

// ConvTranspose2d weight [Cin][Cout][2][2] -> 1x1 conv weight OIHW [(dy*2+dx)*Cout + co][ci]
__global__ void tconv_weight_kernel(const float* __restrict__ w, float* __restrict__ o, int Cin, int Cout) {
    const int n = Cin * Cout * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int ci = i % Cin;
        const int row = i / Cin;                 // (dy*2+dx)*Cout + co
        const int co = row % Cout, ph = row / Cout;
        o[i] = w[((size_t)ci * Cout + co) * 4 + ph];
    }
}
// the same permutation for any stride s (ConvTranspose2d(k = s, stride = s) of a DecoderBlock(upsample_scale = s) called on its own):
// depth [B][H][W][s*s*C] with channels ordered (dy, dx, c)  <->  space [B][sH][sW][C]
__global__ __launch_bounds__(256) void depth_space_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C,
                                                          int s, int to_space, size_t total4) {
    const int c4n = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        size_t r = i / c4n;
        const int X = (int)(r % ((size_t)s * W)); r /= (size_t)s * W;
        const int Y = (int)(r % ((size_t)s * H));
        const int b = (int)(r / ((size_t)s * H));
        const size_t d = ((((size_t)b * H + Y / s) * W + X / s) * (s * s) + ((Y % s) * s + X % s)) * c4n + c4;
        if (to_space) reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(in)[d];
        else reinterpret_cast<f32x4*>(out)[d] = reinterpret_cast<const f32x4*>(in)[i];
    }
}
}  // namespace

int sbgm_launch_depth_space(const float* in, float* out, int B, int H, int W, int C, int s, int to_space, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0 && s >= 1 && s <= 16, "depth<->space: C=%d must be a multiple of 4, stride %d in 1..16", C, s);
    const size_t total4 = (size_t)B * s * s * H * W * (C / 4);
    hipLaunchKernelGGL(depth_space_kernel, dim3(stream_blocks(total4 * 4)), dim3(256), 0, st, in, out, B, H, W, C, s, to_space, total4);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_depth_space2(const float* in, float* out, int B, int H, int W, int C, int to_space, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0, "depth<->space: C=%d must be a multiple of 4", C);
    const size_t total4 = (size_t)B * 4 * H * W * (C / 4);
    hipLaunchKernelGGL(depth_space2_kernel, dim3(stream_blocks(total4 * 4)), dim3(256), 0, st, in, out, B, H, W, C, to_space, total4);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_tconv_weight(const float* w, float* oihw, int Cin, int Cout, hipStream_t st) {
    hipLaunchKernelGGL(tconv_weight_kernel, dim3(std::min((Cin * Cout * 4 + 255) / 256, 4096)), dim3(256), 0, st, w, oihw, Cin, Cout);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                        float* scale, float* bias, int C, hipStream_t st) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, mean, var, eps, scale, bias, C);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_time_embed(const TimeEmbedArgs& a, hipStream_t st) {
    SBGM_CHECK(a.n_emb >= 1 && a.n_emb <= 8 && a.n_proj >= 1 && a.n_proj <= 16, "time_embed: bad counts");
    SBGM_CHECK(a.D % 2 == 0, "time_embed: D=%d must be even", a.D);
    const int total = a.n_emb * a.B * a.D;
    hipLaunchKernelGGL(time_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a);
    SBGM_LAUNCH_CHECK();
    int total_ch = 0;
    for (int i = 0; i < a.n_proj; ++i) total_ch += a.proj[i].ch;
    const int bgroups = (a.B + TP_BG - 1) / TP_BG;
    hipLaunchKernelGGL(time_proj_kernel, dim3((total_ch * bgroups + 3) / 4), dim3(256), 0, st, a, total_ch, bgroups);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_conv3x3_cout1(const float* x, const float* w_tap_c, const float* bias, const float* t, float sigma,
                              float* out, int B, int H, int W, int C, hipStream_t st) {
    SBGM_CHECK(C % 4 == 0, "conv3x3_cout1: C=%d must be a multiple of 4", C);
    const size_t nstrips = (size_t)B * H * ((W + 7) / 8);
    hipLaunchKernelGGL(conv3x3_cout1_kernel, dim3((int)std::min<size_t>((nstrips + 15) / 16, 8192)), dim3(256), 0, st, x,
                       w_tap_c, bias, t, sigma, out, B, H, W, C);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_tap_stencil(const float* d, const float* bias, const float* t, float sigma, float* out, int B, int H, int W,
                            hipStream_t st, int parts) {
    hipLaunchKernelGGL(tap_stencil_kernel, dim3(stream_blocks((size_t)B * H * W)), dim3(256), 0, st, d, bias, t, sigma, out, B, H, W, parts);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_pack_cout1_weight(const float* w_oihw, float* w_tap_c, int C, hipStream_t st) {
    hipLaunchKernelGGL(pack_cout1_weight_kernel, dim3((9 * C + 255) / 256), dim3(256), 0, st, w_oihw, w_tap_c, C);
    SBGM_LAUNCH_CHECK();
    return 0;
}

int sbgm_launch_act(float* x, size_t n, int act, hipStream_t st) {
    SBGM_CHECK(n % 4 == 0, "act: n must be a multiple of 4");
    hipLaunchKernelGGL(act_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, st, x, n / 4, act);
    SBGM_LAUNCH_CHECK();
    return 0;
}
