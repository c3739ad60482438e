// Per-op C-ABI entry points declared in include/sbgm_hip.h (the model / sampler entry points live in engine.hip).
#include "../../include/sbgm_hip.h"
#include "common.h"
#include "kernels.h"

#define ST ((hipStream_t)stream)

extern "C" {

int sbgm_pack_input(const float* const* srcs, const int* src_channels, int n_src, float* dst_nhwc, int B, int H, int W,
                    int c_pad, void* stream) {
    SBGM_CHECK(n_src >= 1 && n_src <= 4, "pack_input: n_src=%d (1..4)", n_src);
    PackSrc s{};
    s.n = n_src;
    for (int i = 0; i < n_src; ++i) { s.ptr[i] = srcs[i]; s.ch[i] = src_channels[i]; }
    return sbgm_launch_pack_input(s, dst_nhwc, B, H, W, c_pad, ST);
}
int sbgm_nchw_to_nhwc(const float* src, float* dst, int B, int H, int W, int C, void* stream) {
    return sbgm_launch_nchw_to_nhwc(src, dst, B, H, W, C, ST);
}
int sbgm_nhwc_to_nchw(const float* src, float* dst, int B, int H, int W, int C, void* stream) {
    return sbgm_launch_nhwc_to_nchw(src, dst, B, H, W, C, ST);
}

int64_t sbgm_conv_packed_numel(int Cout, int KH, int KW, int c_pad) {
    return (int64_t)sbgm_conv_nsteps(KH, KW, c_pad) * Cout * 16;
}
int64_t sbgm_conv_wino_packed_numel(int Cout, int c_pad) { return (int64_t)sbgm_wino_packed_floats(Cout, c_pad); }
int sbgm_conv_wino_pack_weight(const float* w_oihw, float* packed, int Cout, int Cin, int c_pad, void* stream) {
    return sbgm_launch_pack_wino_weight(w_oihw, packed, Cout, Cin, c_pad, ST);
}
int64_t sbgm_conv_wino2d_packed_numel(int Cout, int c_pad) { return (int64_t)sbgm_w2d_packed_floats(Cout, c_pad); }
int sbgm_conv_wino2d_pack_weight(const float* w_oihw, float* packed, int Cout, int Cin, int c_pad, void* stream) {
    return sbgm_launch_pack_w2d_weight(w_oihw, packed, Cout, Cin, c_pad, ST);
}
int sbgm_conv_pack_weight(const float* w_oihw, float* packed, int Cout, int Cin, int KH, int KW, int c_pad, void* stream) {
    return sbgm_launch_pack_conv_weight(w_oihw, packed, Cout, Cin, KH, KW, c_pad, ST);
}

int sbgm_conv2d_fwd(const sbgm_conv_args* a, void* stream) {
    SBGM_CHECK(a && a->x && a->w_packed && a->out, "conv2d: null tensor");
    ConvParams p{};
    p.x = a->x; p.wp = a->w_packed; p.out = a->out; p.scale = a->scale; p.bias = a->bias; p.tbias = a->tbias;
    p.res = a->residual; p.B = a->B; p.H = a->H; p.W = a->W; p.Cs = a->c_pad; p.Cout = a->Cout;
    p.act = a->act;
    SBGM_CHECK(a->act == SBGM_NONE || a->act == SBGM_RELU || a->act == SBGM_GELU, "conv2d: act must be none, relu or gelu");
    p.tbias_after_act = a->tbias_after_act;
    p.in_dil = a->in_dil; p.out_h = a->out_h; p.out_w = a->out_w;
    p.in_mode = a->in_mode; p.in_affine = a->in_affine; p.in_skip = a->in_skip; p.in_act = a->in_act;
    SBGM_CHECK(a->in_mode == 0 || (a->winograd & 3) == 3 || (a->winograd & 8), "conv2d: in_mode %d needs an LDS-staged Winograd kernel (winograd bits 0 and 1, or bit 3)", a->in_mode);
    if (a->winograd & 8) {                   // 2-D Winograd F(2x2,3x3), LDS-staged (conv_w2d.hip): weights from sbgm_conv_wino2d_pack_weight
        SBGM_CHECK(a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->in_dil <= 1, "conv2d: the 2-D Winograd path is 3x3 stride 1 pad 1 only");
        SBGM_CHECK(a->w_wino2d || !(a->winograd & 3), "conv2d: winograd bit 3 beside bits 0/1 needs w_wino2d");
        if (a->w_wino2d) p.wp = a->w_wino2d;
        const ConvTile t2{a->tile_co ? a->tile_co : 2, 1, 1, (a->winograd & 16) ? 2 : (a->waves_per_tile == 2 ? 2 : 1), 2,
                          (a->winograd & 16) ? 3 : (a->winograd & 4) ? 2 : 1};
        return sbgm_launch_conv_w2d(p, t2, ST);
    }
    if ((a->winograd & 1) && a->w_wino) p.wp = a->w_wino;
    ConvTile t{a->tile_co ? a->tile_co : (a->Cout % 64 == 0 ? 4 : 2), a->tile_px ? a->tile_px : 2, a->splits ? a->splits : 1,
               a->waves_per_tile ? a->waves_per_tile : 1, a->winograd & 1, (a->winograd & 2) ? ((a->winograd & 4) ? 2 : 1) : 0};
    if (a->winograd & 2) {
        SBGM_CHECK(a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->in_dil <= 1, "conv2d: the LDS path is 3x3 stride 1 pad 1 only");
        if (!a->tile_px) t.fpx = 1;
        return sbgm_launch_conv_lds(p, t, ST);
    }
    if (a->winograd & 1) {
        SBGM_CHECK(a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->in_dil <= 1, "conv2d: Winograd path is 3x3 stride 1 pad 1 only");
        if (!a->tile_px) t.fpx = 1;
        return sbgm_launch_conv_wino(p, t, ST);
    }
    SBGM_CHECK(a->Cout % 32 == 0, "conv2d: Cout=%d must be a multiple of 32", a->Cout);
    if (t.splits > 1) {
        const int OH = a->out_h > 0 ? a->out_h : (a->H + 2 * a->pad - a->KH) / a->stride + 1;
        const int OW = a->out_w > 0 ? a->out_w : (a->W + 2 * a->pad - a->KW) / a->stride + 1;
        SBGM_CHECK(a->ws && a->ws_floats >= (int64_t)t.splits * a->B * OH * OW * a->Cout, "conv2d: split-K workspace too small");
    }
    return sbgm_launch_conv(ConvGeom{a->KH, a->KW, a->stride, a->pad}, p, t, a->ws, ST);
}

int sbgm_conv_pack_weights_batched_blocks(int Cout, int KH, int KW, int c_pad) { return sbgm_conv_pack_blocks(Cout, KH, KW, c_pad); }
int sbgm_conv_pack_weights_batched(const sbgm_pack_desc* desc_dev, int n, int total_blocks, void* stream) {
    return sbgm_launch_pack_conv_weights_batched(desc_dev, n, total_blocks, ST);   // descriptors live on the device: the caller
}                                                                                    // guarantees bit 1 only on 3x3, cs%16, Cout%16
int sbgm_adam_step_blocks(int64_t numel) { return sbgm_adam_blocks(numel); }
int sbgm_adam_step_batched(const sbgm_adam_desc* desc_dev, int n, int total_blocks, const float* step, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int decoupled, float grad_scale, void* stream) {
    return sbgm_launch_adam_batched(desc_dev, n, total_blocks, step, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale, ST);
}
int sbgm_batchnorm_train_stats(const float* x, int B, int HW, int C, void* stats_ws, void* stream) {
    return sbgm_launch_batchnorm_stats(x, B, HW, C, static_cast<double*>(stats_ws), ST);
}
int sbgm_batchnorm_train_apply(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, const float* residual, const float* tbias_after, int relu, int B, int HW, int C,
                               float eps, float momentum, void* stats_ws, double n_total, float* mean_rstd_out, void* stream) {
    return sbgm_launch_batchnorm_apply(x, y, gamma, beta, running_mean, running_var, residual, tbias_after, relu, B, HW, C, eps, momentum,
                                       static_cast<double*>(stats_ws), n_total, ST, mean_rstd_out);
}
int sbgm_batchnorm_bwd_reduce(const float* x, const float* dy, const float* y, const float* tbias_after, const float* mean_rstd,
                              int relu, float* ws, int B, int HW, int C, void* stream) {
    return sbgm_launch_batchnorm_bwd_reduce(x, dy, y, tbias_after, mean_rstd, relu, ws, B, HW, C, ST);
}
int sbgm_batchnorm_bwd_apply(const float* x, const float* dy, const float* y, const float* gamma, const float* tbias_after,
                             const float* mean_rstd, int relu, float* dx, float* dres, float* dgamma, float* dbeta, const float* ws,
                             const float* sync_sums, double n_total, int B, int HW, int C, void* stream) {
    return sbgm_launch_batchnorm_bwd_apply(x, dy, y, gamma, tbias_after, mean_rstd, relu, dx, dres, dgamma, dbeta, ws, sync_sums, n_total,
                                           B, HW, C, ST);
}
int sbgm_conv8x8s2_dgrad_phase_weight(const float* w_oihw, float* out_oihw, int Cout, int Cin, void* stream) {
    return sbgm_launch_dgrad_phase_weight(w_oihw, out_oihw, Cout, Cin, ST);
}
int sbgm_groupnorm_stats(const float* x, void* stats_ws, int B, int HW, int C, int G, int* chunks, void* stream) {
    SBGM_CHECK(chunks != nullptr, "groupnorm_stats: chunks is required");
    return sbgm_launch_gn_partial(x, static_cast<double*>(stats_ws), B, HW, C, G, chunks, ST);
}
int sbgm_groupnorm_finalize(const void* stats_ws, int chunks, const float* gamma, const float* beta, const float* tbias, float* out, int B,
                            int HW, int C, int G, float eps, void* stream) {
    return sbgm_launch_gn_finalize(static_cast<const double*>(stats_ws), chunks, gamma, beta, tbias, out, B, HW, C, G, eps, ST);
}
int sbgm_dsm_loss_blocks(int64_t per_sample) { return sbgm_dsm_nblk(per_sample); }
int sbgm_dsm_perturb(const float* x, const float* z, const float* t, const uint64_t* rng_state, uint64_t seed, float t_eps, float sigma,
                     float* x_perturbed, float* z_out, float* t_out, float* std_out, int B, int64_t per_sample, void* stream) {
    return sbgm_launch_dsm_perturb(x, z, t, reinterpret_cast<const unsigned long long*>(rng_state), (unsigned long long)seed, t_eps, sigma,
                                   x_perturbed, z_out,
                                   t_out, std_out, B, (size_t)per_sample, ST);
}
int sbgm_dsm_loss_fwd(const float* score, const float* z, const float* std, const float* sdf, void* partial_ws, float* loss,
                      uint64_t* rng_state, int B, int64_t per_sample, void* stream) {
    return sbgm_launch_dsm_loss_fwd(score, z, std, sdf, static_cast<double*>(partial_ws), loss,
                                    reinterpret_cast<unsigned long long*>(rng_state), B, (size_t)per_sample, ST);
}
int sbgm_dsm_loss_bwd(const float* score, const float* z, const float* std, const float* sdf, const float* dloss, float* dscore,
                      int B, int64_t per_sample, void* stream) {
    return sbgm_launch_dsm_loss_bwd(score, z, std, sdf, dloss, dscore, B, (size_t)per_sample, ST);
}
int sbgm_set_scratch_prezeroed(int on) {
    const int prev = sbgm_scratch_prezeroed;
    sbgm_scratch_prezeroed = on ? 1 : 0;
    return prev;
}
int sbgm_wgrad_defer(int on) {
    const int prev = sbgm_wgrad_deferred;
    sbgm_wgrad_deferred = on & 3;
    return prev;
}
int sbgm_wgrad_flush(void* stream) { return sbgm_launch_wgrad_flush(ST); }
int sbgm_wgrad_flush_pending(void) { return sbgm_wgrad_pending(); }
int sbgm_wgrad_discard(void) { const int n = sbgm_wgrad_pending(); sbgm_wgrad_discard_queue(); return n; }

int sbgm_conv2d_tune(const sbgm_conv_args* a, int* tile, void* stream) {
    SBGM_CHECK(a && tile && a->x && a->w_packed && a->out, "conv2d_tune: null argument");
    SBGM_CHECK(a->Cout % 32 == 0, "conv2d_tune: Cout=%d must be a multiple of 32", a->Cout);
    ConvParams p{};
    p.x = a->x; p.wp = a->w_packed; p.out = a->out; p.scale = a->scale; p.bias = a->bias; p.tbias = a->tbias;
    p.res = a->residual; p.B = a->B; p.H = a->H; p.W = a->W; p.Cs = a->c_pad; p.Cout = a->Cout;
    p.act = a->act; p.tbias_after_act = a->tbias_after_act;
    p.in_dil = a->in_dil; p.out_h = a->out_h; p.out_w = a->out_w;
    p.wp_wino = a->w_wino;
    p.wp_w2d = a->w_wino2d;
    ConvTile best{a->Cout % 64 == 0 ? 4 : 2, 2, 1, 1, 0, 0};
    if (sbgm_tune_conv(ConvGeom{a->KH, a->KW, a->stride, a->pad}, p, a->ws, a->ws ? (size_t)a->ws_floats : 0, ST, &best)) return 1;
    tile[0] = best.fco; tile[1] = best.fpx; tile[2] = best.splits; tile[3] = best.ws; tile[4] = best.wino; tile[5] = best.lds;
    return 0;
}

int sbgm_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    return sbgm_launch_upsample2x(x, y, B, H, W, C, ST);
}
int sbgm_groupnorm_fwd(const float* x, float* y, const float* gamma, const float* beta, const float* skip, const float* tbias,
                       int act, int B, int HW, int C, int G, float eps, void* stats_ws, float* mean_rstd_out, void* stream) {
    return sbgm_launch_groupnorm(x, y, gamma, beta, skip, tbias, act, B, HW, C, G, eps, (double*)stats_ws, ST, mean_rstd_out);
}
int sbgm_layernorm_fwd(const float* x, float* y, const float* gamma, const float* beta, int M, int C, float eps, void* stream) {
    return sbgm_launch_layernorm(x, y, gamma, beta, M, C, eps, ST);
}
int sbgm_batchnorm_train_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, const float* residual, const float* tbias_after, int relu, int B, int HW,
                             int C, float eps, float momentum, void* stats_ws, float* mean_rstd_out, void* stream) {
    return sbgm_launch_batchnorm_train(x, y, gamma, beta, running_mean, running_var, residual, tbias_after, relu, B, HW, C, eps,
                                       momentum, (double*)stats_ws, ST, mean_rstd_out);
}
int sbgm_mha_core_fwd(const float* qkv, float* out, int B, int S, int C, int heads, void* stream) {
    return sbgm_launch_mha_core(qkv, out, B, S, C, heads, ST);
}
int sbgm_mha_core_dropout_fwd(const float* qkv, float* out, int B, int S, int C, int heads, float p, uint64_t seed, uint64_t offset, void* stream) {
    return sbgm_launch_mha_core_dropout(qkv, nullptr, out, B, S, C, heads, p, seed, offset, 0, ST);
}
int sbgm_mha_core_dropout_bwd(const float* qkv, const float* dout, float* dqkv, int B, int S, int C, int heads, float p, uint64_t seed,
                              uint64_t offset, void* stream) {
    return sbgm_launch_mha_core_dropout(qkv, dout, dqkv, B, S, C, heads, p, seed, offset, 1, ST);
}
int sbgm_mha_dropout_mask(float* mask, int B, int S, int heads, float p, uint64_t seed, uint64_t offset, void* stream) {
    return sbgm_launch_mha_dropout_mask(mask, B, S, heads, p, seed, offset, ST);
}
int sbgm_attn_qkv_fwd(const float* x, const float* ln_gamma, const float* ln_beta, const float* w_in_packed, const float* b_in,
                      float* qkv, int M, int C, float eps, void* stream) {
    return sbgm_launch_attn_in(x, ln_gamma, ln_beta, w_in_packed, b_in, qkv, M, C, eps, ST);
}
int sbgm_attn_tail_fwd(const float* att, const float* x, const float* w_out_packed, const float* b_out, const float* ln_gamma,
                       const float* ln_beta, const float* w_ff1_packed, const float* b_ff1, const float* w_ff2_packed,
                       const float* b_ff2, float* out, int M, int C, float eps, void* stream) {
    return sbgm_launch_attn_out(att, x, w_out_packed, b_out, ln_gamma, ln_beta, w_ff1_packed, b_ff1, w_ff2_packed, b_ff2, out, M, C, eps, ST);
}
int sbgm_time_proj_fwd(const float* t, const int64_t* y, const float* label_emb, const float* freqs, const float* weight,
                       const float* bias, float* out, float* emb_ws, float* emb_raw, int B, int D, int ch, void* stream) {
    TimeEmbedArgs a{};
    a.emb_raw = emb_raw;
    a.t = t; a.y = y; a.label_emb = label_emb; a.freqs[0] = freqs; a.n_emb = 1; a.n_proj = 1;
    a.proj[0] = TimeProj{weight, bias, out, ch, 0};
    a.emb_ws = emb_ws; a.B = B; a.D = D;
    return sbgm_launch_time_embed(a, ST);
}
int sbgm_time_proj_multi_fwd(const float* t, const int64_t* y, const float* label_emb, const float* const* freqs, int n_emb,
                             const float* const* weights, const float* const* biases, float* const* outs, const int* chs,
                             const int* emb_index, int n_proj, float* emb_ws, float* emb_raw, int B, int D, void* stream) {
    SBGM_CHECK(t && freqs && weights && biases && outs && chs && emb_index && emb_ws, "time_proj_multi: null argument");
    SBGM_CHECK(n_emb >= 1 && n_emb <= 8 && n_proj >= 1 && n_proj <= 16, "time_proj_multi: n_emb=%d (1..8), n_proj=%d (1..16)", n_emb, n_proj);
    TimeEmbedArgs a{};
    a.t = t; a.y = y; a.label_emb = label_emb; a.n_emb = n_emb; a.n_proj = n_proj;
    for (int i = 0; i < n_emb; ++i) a.freqs[i] = freqs[i];
    for (int i = 0; i < n_proj; ++i) {
        SBGM_CHECK(emb_index[i] >= 0 && emb_index[i] < n_emb, "time_proj_multi: emb_index[%d]=%d", i, emb_index[i]);
        a.proj[i] = TimeProj{weights[i], biases[i], outs[i], chs[i], emb_index[i]};
    }
    a.emb_ws = emb_ws; a.emb_raw = emb_raw; a.B = B; a.D = D;
    return sbgm_launch_time_embed(a, ST);
}
int sbgm_cout1_pack_weight(const float* w_oihw, float* w_tap_c, int C, void* stream) {
    return sbgm_launch_pack_cout1_weight(w_oihw, w_tap_c, C, ST);
}
int sbgm_conv3x3_cout1_fwd(const float* x, const float* w_tap_c, const float* bias, const float* t, float sigma, float* out,
                           int B, int H, int W, int C, void* stream) {
    return sbgm_launch_conv3x3_cout1(x, w_tap_c, bias, t, sigma, out, B, H, W, C, ST);
}
int sbgm_act_inplace(float* x, int64_t n, int act, void* stream) { return sbgm_launch_act(x, (size_t)n, act, ST); }

int sbgm_pointwise_chain(const float* x, float* y, int64_t n, int n_ops, const int* ops, const float* consts, void* stream) {
    SBGM_CHECK(n >= 0 && (n_ops == 0 || (ops && consts)), "pointwise_chain: bad arguments");
    return sbgm_launch_pointwise_chain(x, y, (size_t)n, n_ops, ops, consts, ST);
}
int sbgm_sample_extremes(const float* x, int B, int64_t per_sample, float q, float* out_max, float* out_q, void* stream) {
    SBGM_CHECK(x && out_max && out_q && per_sample > 0, "sample_extremes: bad arguments");
    return sbgm_launch_sample_extremes(x, B, (size_t)per_sample, q, out_max, out_q, ST);
}

int sbgm_assemble_conditions(const sbgm_assemble_args* a, void* stream) {
    SBGM_CHECK(a, "assemble_conditions: null args");
    return sbgm_launch_assemble_conditions(*a, ST);
}

int sbgm_extract_tiles(const float* domain, const int* origins, float* tiles, int T, int C, int Hd, int Wd, int th, int tw,
                       void* stream) {
    SBGM_CHECK(domain && origins && tiles, "extract_tiles: null pointer");
    return sbgm_launch_extract_tiles(domain, origins, tiles, T, C, Hd, Wd, th, tw, ST);
}
int sbgm_stitch_tiles(const float* tiles, const int* origins, float* domain, int T, int C, int Hd, int Wd, int th, int tw,
                      int ramp_len, void* stream) {
    SBGM_CHECK(domain && origins && tiles, "stitch_tiles: null pointer");
    return sbgm_launch_stitch_tiles(tiles, origins, domain, T, C, Hd, Wd, th, tw, ramp_len, ST);
}

int sbgm_depth_to_space2(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    return sbgm_launch_depth_space2(x, y, B, H, W, C, 1, ST);
}
int sbgm_space_to_depth2(const float* y, float* x, int B, int H, int W, int C, void* stream) {
    return sbgm_launch_depth_space2(y, x, B, H, W, C, 0, ST);
}
int sbgm_depth_to_space(const float* x, float* y, int B, int H, int W, int C, int s, void* stream) {
    return sbgm_launch_depth_space(x, y, B, H, W, C, s, 1, ST);
}
int sbgm_space_to_depth(const float* y, float* x, int B, int H, int W, int C, int s, void* stream) {
    return sbgm_launch_depth_space(y, x, B, H, W, C, s, 0, ST);
}
int sbgm_tconv_weight_to_oihw(const float* w, float* oihw, int Cin, int Cout, void* stream) {
    return sbgm_launch_tconv_weight(w, oihw, Cin, Cout, ST);
}

// ---- training path: backward entry points ---------------------------------------------------------------------------
int sbgm_conv_pack_weight_dgrad(const float* w_oihw, float* packed, int Cout, int Cin, int KH, int KW, void* stream) {
    // operator of the data gradient: Cout' = Cin, Cin' = Cout (padded to 16), taps flipped
    return sbgm_launch_pack_conv_weight(w_oihw, packed, Cin, Cout, KH, KW, (Cout + 15) / 16 * 16, ST, 1);
}
int sbgm_conv2d_wgrad(const float* dy, const float* x, float* dw_oihw, float* ws, int B, int H, int W, int c_pad, int Cin, int Cout,
                      int KH, int KW, int stride, int pad, void* stream) {
    return sbgm_launch_conv_wgrad(dy, x, dw_oihw, ws, B, H, W, c_pad, Cin, Cout, KH, KW, stride, pad, ST);
}
int sbgm_conv2d_wgrad_bias(const float* dy, const float* x, float* dw_oihw, float* dbias, float* ws, int B, int H, int W, int c_pad, int Cin,
                           int Cout, int KH, int KW, int stride, int pad, void* stream) {
    SBGM_CHECK(dbias != nullptr, "conv2d_wgrad_bias: dbias is required");
    return sbgm_launch_conv_wgrad(dy, x, dw_oihw, ws, B, H, W, c_pad, Cin, Cout, KH, KW, stride, pad, ST, dbias);
}
int sbgm_colsum(const float* x, const float* y, float* out, int M, int C, void* stream) { return sbgm_launch_colsum(x, y, out, M, C, ST); }
int sbgm_samplesum(const float* x, float* out, int B, int HW, int C, void* stream) { return sbgm_launch_samplesum(x, out, B, HW, C, ST); }
int sbgm_groupnorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* skip, const float* tbias,
                       const float* mean_rstd, int act, float* dx, float* dskip, float* dgamma, float* dbeta, float* dtbias, float* ws,
                       int B, int HW, int C, int G, void* stream) {
    return sbgm_launch_groupnorm_bwd(x, dy, gamma, beta, skip, tbias, mean_rstd, act, dx, dskip, dgamma, dbeta, dtbias, ws, B, HW, C, G, ST);
}
int sbgm_batchnorm_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* tbias_after,
                       const float* mean_rstd, int relu, float* dx, float* dres, float* dgamma, float* dbeta, float* ws, int B, int HW,
                       int C, void* stream) {
    return sbgm_launch_batchnorm_bwd(x, dy, y, gamma, tbias_after, mean_rstd, relu, dx, dres, dgamma, dbeta, ws, B, HW, C, ST);
}
int sbgm_layernorm_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta, int M, int C,
                       float eps, const float* dx_add, void* stream) {
    return sbgm_launch_layernorm_bwd(x, dy, gamma, dx, dgamma, dbeta, M, C, eps, ST, dx_add);
}
int sbgm_fill_zero(void* p, int64_t bytes, void* stream) {
    SBGM_CHECK(p && bytes >= 0 && bytes % 4 == 0, "fill_zero: bytes=%lld must be a non-negative multiple of 4", (long long)bytes);
    return bytes ? sbgm_zero_async(p, (size_t)bytes, ST) : 0;
}
int sbgm_mha_core_bwd(const float* qkv, const float* dout, float* dqkv, int B, int S, int C, int heads, void* stream) {
    return sbgm_launch_mha_core_bwd(qkv, dout, dqkv, B, S, C, heads, ST);
}
int sbgm_upsample_bilinear_fwd(const float* x, float* y, int B, int H, int W, int C, int scale, void* stream) {
    return sbgm_launch_upsample_bilinear(x, y, B, H, W, C, scale, 0, ST);
}
int sbgm_upsample_bilinear_bwd(const float* dy, float* dx, int B, int H, int W, int C, int scale, void* stream) {
    return sbgm_launch_upsample_bilinear(dy, dx, B, H, W, C, scale, 1, ST);
}
int sbgm_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream) {
    return sbgm_launch_upsample2x_bwd(dy, dx, B, H, W, C, ST);
}
int sbgm_conv3x3_cout1_bwd(const float* dout, const float* a, const float* w_tap_c, const float* t, float sigma, float* da,
                           float* dw_tap_c, float* dbias, int B, int H, int W, int C, void* stream) {
    return sbgm_launch_cout1_bwd(dout, a, w_tap_c, t, sigma, da, dw_tap_c, dbias, B, H, W, C, ST);
}
int sbgm_time_proj_bwd(const float* dout, const float* weight, const float* semb, const float* emb_raw, float* dW, float* dbias,
                       float* demb_accum, int B, int D, int ch, void* stream) {
    return sbgm_launch_time_proj_bwd(dout, weight, semb, emb_raw, dW, dbias, demb_accum, B, D, ch, ST);
}
int sbgm_time_proj_multi_bwd(const float* const* douts, const float* const* sembs, float* const* dWs, float* const* dbiases, const int* chs,
                             int n_proj, int B, int D, void* stream) {
    SBGM_CHECK(douts && sembs && dWs && dbiases && chs, "time_proj_multi_bwd: null argument");
    return sbgm_launch_time_proj_multi_bwd(douts, sembs, dWs, dbiases, chs, n_proj, B, D, ST);
}
int sbgm_label_emb_bwd(const float* demb, const int64_t* y, float* dtable, int B, int D, void* stream) {
    return sbgm_launch_label_emb_bwd(demb, y, dtable, B, D, ST);
}
int sbgm_act_fwd(const float* x, float* y, int64_t n, int act, void* stream) { return sbgm_launch_act_fwd(x, y, (size_t)n, act, ST); }
int sbgm_act_bwd(const float* x, const float* dy, float* dx, int64_t n, int act, void* stream) {
    return sbgm_launch_act_bwd(x, dy, dx, (size_t)n, act, ST);
}

int sbgm_em_step(float* x, float* x_mean, const float* score, const float* z, float g2, float dt, float noise_coef,
                 uint64_t seed, uint64_t draw_index, int64_t n, void* stream) {
    const StepScalars sc{0.f, g2, dt, noise_coef, 0.f};
    return sbgm_launch_em_update(x, x_mean, score, z, nullptr, nullptr, &sc, draw_index, nullptr, seed, 1, (size_t)n, 1, ST);
}
int sbgm_langevin_step(float* x, const float* score, const float* z, float snr_noise_norm, void* sumsq_ws, uint64_t seed,
                       uint64_t draw_index, int B, int64_t per_sample, void* stream) {
    return sbgm_launch_langevin(x, score, z, snr_noise_norm, (double*)sumsq_ws, nullptr, draw_index, seed, B, (size_t)per_sample, ST);
}
int sbgm_cfg_combine(float* out, const float* s_cond, const float* s_uncond, float scale, int64_t n, void* stream) {
    return sbgm_launch_cfg_combine(out, s_cond, s_uncond, scale, (size_t)n, ST);
}
int sbgm_randn_scaled(float* x, float scale, uint64_t seed, uint64_t draw_index, int64_t n, void* stream) {
    return sbgm_launch_init_noise(x, scale, nullptr, seed, nullptr, draw_index, (size_t)n, ST);
}

}  // extern "C"
