// Internal launcher interface between the kernel translation units and the engine / C-ABI layer.
// Every launcher enqueues on `st`, never synchronises and never allocates (graph-capture safe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

struct ConvGeom {
    int kh, kw, stride, pad;
};
struct ConvTile {
    int fco, fpx;   // wave tile = (16*fco) output channels x (16*fpx) pixels
    int splits;     // split-K over gridDim.y (partials + reduce kernel)
    int ws;         // waves of a workgroup cooperating on one tile (in-workgroup split-K through LDS): 1, 2 or 4
    int wino;       // 1 = 3x3/s1 Winograd F(2,3) kernel (conv_wino.hip): fpx then counts PAIR fragments, weights = wino pack;
                    // 2 = 2-D Winograd F(2x2,3x3), LDS-staged only (conv_w2d.hip): 16x16-pixel tiles, fpx unused, weights = w2d pack
    int lds;        // 1 = LDS-staged 3x3/s1 kernel (conv_lds.hip): fpx = tile rows per wave (Winograd: 2*fpx rows)
};
struct ConvParams {
    const float* x;       // NHWC [B][H][W][Cs]
    const float* wp;      // packed weights [nsteps][Cout][16]
    float* out;           // NHWC [M][Cout]
    const float* scale;   // [Cout] or null   (folded BatchNorm gamma/sqrt(var+eps))
    const float* bias;    // [Cout] or null
    const float* tbias;   // [B][Cout] or null (time-projection bias, broadcast over pixels)
    const float* res;     // [M][Cout] or null (residual / skip)
    int B, H, W, Cs, Cout;
    int act, tbias_after_act;
    const float* proj_w;  // [9][Cout] or null: fuse the following 3x3 Cout=1 conv's per-tap channel dot products
    float* proj_out;      // [9][M] planar tap sums (then `out` is not written)
    const float* wp_wino; // host-side only: Winograd-packed copy of the weights (3x3 stride-1 layers), or null
    const float* wp_w2d;  // host-side only: F(2x2,3x3)-packed copy of the weights (3x3 stride-1 layers), or null
    double* gn_stats;     // conv_lds only, or null: per-workgroup GroupNorm partial sums of the OUTPUT (sum, sum of squares per
                          // group) in the [b][chunk][G][2] layout groupnorm_apply reads -> no separate statistics pass
    int gn_groups;        // G of that GroupNorm (channels per group must divide or be a multiple of the tile's channel slice)
    int c_real;           // 0, or 2 with Cs == 4: only 2 of the 4 stored channels are real (the 2-channel stem): a K step is then
                          // 8 taps x 2 channels (weights packed with cs = 2) instead of 4 taps x 4 slots, halving the MFMA work
    int in_dil;           // 1, or 2: read the input through a zero-inserted grid (data-gradient of a stride-2 conv)
    int out_h, out_w;     // explicit output size (required with in_dil == 2), else 0
    // conv_lds only — what happens to the input while the halo patch is staged (conv_lds.hip, "Input modes"):
    int in_mode;          // 0 plain; 1 affine on load (x*scale + shift per (sample, channel), zero padding kept); 2 bilinear x2 on
                          // load: x is the LOW-resolution map [B][H/2][W/2][Cs] (H, W stay the convolution's own size), optionally
                          // transformed act(x*scale + shift + skip) before the interpolation
    const float* in_affine;  // [B][Cs/4][2][4] (scale quad, shift quad) from sbgm_launch_gn_finalize, or null
    const float* in_skip;    // mode 2: [B][H/2][W/2][Cs] added before the activation, or null
    int in_act;              // mode 2: SBGM_ACT_* applied to the low-res value
    // filled by sbgm_launch_conv:
    int OH, OW, M, cb_per_tap, nsteps, steps_per_split, n_px_tiles, n_co_tiles;
    uint32_t x_bytes, w_bytes;
};

int sbgm_conv_nsteps(int KH, int KW, int cs);
int sbgm_conv_pack_blocks(int Cout, int KH, int KW, int cs);   // workgroups one weight takes in the batched pack launch
// transposed != 0 packs the data-gradient operator (swap Cout/Cin, flip taps); then Cout/Cin are the transposed sizes
int sbgm_launch_pack_conv_weight(const float* w_oihw, float* wp, int Cout, int Cin, int KH, int KW, int cs, hipStream_t st,
                                 int transposed = 0);
struct sbgm_pack_desc;
struct sbgm_adam_desc;
int sbgm_adam_blocks(int64_t numel);
int sbgm_launch_adam_batched(const sbgm_adam_desc* desc_dev, int n, int total_blocks, const float* step_dev, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int decoupled, float grad_scale, hipStream_t st);
int sbgm_launch_pack_conv_weights_batched(const sbgm_pack_desc* desc_dev, int n, int total_blocks, hipStream_t st);
int sbgm_launch_conv(const ConvGeom& g, ConvParams p, const ConvTile& cfg, float* partial_ws, hipStream_t st);
// When set, the launchers below trust that their atomically accumulated scratch (weight-gradient slab, norm-backward sums)
// arrives zeroed and skip their own memsets (the training path zeroes one pooled buffer per step instead of ~75 small ones).
extern int sbgm_scratch_prezeroed;
extern int sbgm_wgrad_deferred;                 // backward.hip: queue the slab -> OIHW passes for sbgm_launch_wgrad_flush
int sbgm_wgrad_pending();
void sbgm_wgrad_discard_queue();
int sbgm_launch_wgrad_flush(hipStream_t st);

// ---- conv_wino.hip: 3x3 stride-1 pad-1 convolution, 1-D Winograd F(2,3) along rows ------------------------------------
size_t sbgm_wino_packed_floats(int Cout, int cs);
int sbgm_launch_pack_wino_weight(const float* w_oihw, float* up, int Cout, int Cin, int cs, hipStream_t st);
int sbgm_launch_conv_wino(ConvParams p, const ConvTile& cfg, hipStream_t st);   // p.wp = Winograd-packed weights

// ---- conv_lds.hip: 3x3 stride-1 pad-1 convolution with LDS-staged halo patch + weight slab (direct or Winograd) -----------
int sbgm_launch_conv_lds(ConvParams p, const ConvTile& cfg, hipStream_t st);
size_t sbgm_conv_lds_bytes(const ConvTile& cfg, int in_mode);
// chunks per sample the launch above writes into p.gn_stats for this tile, or 0 if that tile cannot produce them
int sbgm_conv_lds_gn_chunks(const ConvParams& p, const ConvTile& cfg);

// ---- conv_w2d.hip: the same convolution as a 2-D Winograd F(2x2,3x3), LDS-staged (ConvTile.wino == 2) -----------------------
size_t sbgm_w2d_packed_floats(int Cout, int cs);
int sbgm_launch_pack_w2d_weight(const float* w_oihw, float* up, int Cout, int Cin, int cs, hipStream_t st);
int sbgm_launch_conv_w2d(ConvParams p, const ConvTile& cfg, hipStream_t st);    // p.wp = F(2x2,3x3)-packed weights
size_t sbgm_conv_w2d_bytes(const ConvTile& cfg, int in_mode);
int sbgm_conv_w2d_gn_chunks(const ConvParams& p, const ConvTile& cfg);
// co tiles of a tap-projection launch: each writes its own partial plane [tile][9][M], sbgm_launch_tap_stencil sums `parts` planes
int sbgm_conv_w2d_proj_parts(const ConvParams& p, const ConvTile& cfg);

// ---- pointwise.hip ---------------------------------------------------------------------------------
struct PackSrc {
    const float* ptr[4];   // NCHW sources, concatenated along C in this order
    int ch[4];
    int n;
};
int sbgm_launch_pack_input(const PackSrc& src, float* dst_nhwc, int B, int H, int W, int Cs, hipStream_t st);
int sbgm_launch_nhwc_to_nchw(const float* src, float* dst, int B, int H, int W, int C, hipStream_t st);
int sbgm_launch_nchw_to_nhwc(const float* src, float* dst, int B, int H, int W, int C, hipStream_t st);
int sbgm_launch_upsample2x(const float* x, float* y, int B, int H, int W, int C, hipStream_t st);
// nn.Upsample(scale_factor = scale, bilinear, align_corners=False), integer scale 1..16; backward != 0: x = dy (upsampled size), y = dx
int sbgm_launch_upsample_bilinear(const float* x, float* y, int B, int H, int W, int C, int scale, int backward, hipStream_t st);
// [B][H][W][4C] (phase-major channels) <-> [B][2H][2W][C]; to_space = 1: depth -> space
int sbgm_launch_depth_space2(const float* in, float* out, int B, int H, int W, int C, int to_space, hipStream_t st);
int sbgm_launch_depth_space(const float* in, float* out, int B, int H, int W, int C, int s, int to_space, hipStream_t st);   // any stride s
int sbgm_launch_tconv_weight(const float* w_cin_cout_2_2, float* oihw_4cout_cin, int Cin, int Cout, hipStream_t st);
int sbgm_launch_act(float* x, size_t n, int act, hipStream_t st);
int sbgm_launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                        float* scale, float* bias, int C, hipStream_t st);

struct TimeProj {          // out[b][c] = bias[c] + sum_d W[c][d] * silu(emb_g[b][d])
    const float* weight;   // [ch][D] (nn.Linear layout)
    const float* bias;     // [ch]
    float* out;            // [B][ch]
    int ch;
    int emb;               // which embedding (index into freqs[])
};
struct TimeEmbedArgs {
    const float* t;             // [B]
    const int64_t* y;           // [B] or null
    const float* label_emb;     // [ncls+1][D] or null (added to embedding 0 only)
    const float* freqs[8];      // Gaussian-Fourier W vectors [D/2]
    int n_emb;
    TimeProj proj[16];
    int n_proj;
    float* emb_ws;              // workspace [n_emb][B][D]: silu(embedding)
    float* emb_raw;             // optional [n_emb][B][D]: embedding before the SiLU (training)
    int B, D;
};
int sbgm_launch_time_embed(const TimeEmbedArgs& a, hipStream_t st);

// final 3x3 conv with a single output channel, fused with the division by sigma(t): out NCHW [B,1,H,W]
int sbgm_launch_conv3x3_cout1(const float* x, const float* w_tap_c, const float* bias, const float* t, float sigma,
                              float* out, int B, int H, int W, int C, hipStream_t st);
// out[b,y,x] = (bias + sum_taps d[tap][b, y+kh-1, x+kw-1]) / sigma(t_b): finishes the fused final conv
int sbgm_launch_tap_stencil(const float* d, const float* bias, const float* t, float sigma, float* out, int B, int H, int W,
                            hipStream_t st, int parts = 1);   // parts: partial planes [parts][9][M] summed on the way
int sbgm_launch_pack_cout1_weight(const float* w_oihw, float* w_tap_c, int C, hipStream_t st);

// ---- norm.hip ------------------------------------------------------------------------------------------
// GroupNorm / InstanceNorm over NHWC.  stats_ws: groupnorm needs 16*64*B*G bytes (partial sums per pixel chunk);
// batchnorm needs 24*C bytes (zeroed by the launcher).
// second half of sbgm_launch_groupnorm only: the statistics were already written (by a convolution epilogue) as `chunks` partials
int sbgm_launch_groupnorm_apply(const float* x, float* y, const float* gamma, const float* beta, const float* skip, const float* tbias,
                                int act, int B, int HW, int C, int G, float eps, const double* stats, int chunks, hipStream_t st,
                                float* mr_out = nullptr);
int sbgm_launch_groupnorm(const float* x, float* y, const float* gamma, const float* beta, const float* skip,
                          const float* tbias, int act, int B, int HW, int C, int G, float eps, double* stats_ws,
                          hipStream_t st, float* mr_out = nullptr);   // mr_out: [B][G][2] (mean, rstd) kept for backward
// GroupNorm statistics (chunk partials as written by gn_partial / the conv_lds epilogue) -> per-(sample, channel) affine
// out[b][c/4][0][4] = rstd*gamma, out[b][c/4][1][4] = beta - mean*rstd*gamma (+ tbias[b][c]): what conv_lds applies on load
int sbgm_launch_gn_finalize(const double* stats, int chunks, const float* gamma, const float* beta, const float* tbias, float* out,
                            int B, int HW, int C, int G, float eps, hipStream_t st);
// first half of sbgm_launch_groupnorm only: the chunk partials; returns the chunk count through *chunks
int sbgm_launch_gn_partial(const float* x, double* stats_ws, int B, int HW, int C, int G, int* chunks, hipStream_t st);
int sbgm_launch_layernorm(const float* x, float* y, const float* gamma, const float* beta, int M, int C, float eps,
                          hipStream_t st);
// train-mode BatchNorm2d: batch statistics over (B,H,W), running-stat update, optional residual + ReLU
int sbgm_launch_batchnorm_train(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, const float* res, const float* tbias_after, int relu, int B,
                                int HW, int C, float eps, float momentum, double* stats_ws, hipStream_t st,
                                float* mr_out = nullptr);

// SyncBatchNorm halves (the caller all-reduces stats_ws / the s12 channel sums over the ranks in between)
int sbgm_launch_batchnorm_stats(const float* x, int B, int HW, int C, double* stats_ws, hipStream_t st);
int sbgm_launch_batchnorm_apply(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, const float* res, const float* tbias_after, int relu, int B, int HW, int C,
                                float eps, float momentum, double* stats_ws, double n_total, hipStream_t st, float* mr_out = nullptr);
int sbgm_launch_batchnorm_bwd_reduce(const float* x, const float* dy, const float* y, const float* tbias_after, const float* mr,
                                     int relu, float* s12_ws, int B, int HW, int C, hipStream_t st);
int sbgm_launch_batchnorm_bwd_apply(const float* x, const float* dy, const float* y, const float* gamma, const float* tbias_after,
                                    const float* mr, int relu, float* dx, float* dres, float* dgamma, float* dbeta, const float* s12_ws,
                                    const float* sync_sums, double n_total, int B, int HW, int C, hipStream_t st);

// ---- attention.hip -------------------------------------------------------------------------------------
int sbgm_launch_mha_core(const float* qkv, float* out, int B, int S, int C, int heads, hipStream_t st);
// attn_tokens.hip: the per-token halves of an attention block, one launch each (C in {64, 128, 256, 512})
int sbgm_attn_tokens_supported(int C);
// ---- attention_dropout.hip: the attention core with train-mode dropout on the softmax probabilities (Philox mask keyed by seed / offset)
// forward: out_or_dqkv = out [B,S,C], dout unused;  backward != 0: out_or_dqkv = dqkv [B,S,3C] (zeroed unless sbgm_scratch_prezeroed)
int sbgm_launch_mha_core_dropout(const float* qkv, const float* dout, float* out_or_dqkv, int B, int S, int C, int heads, float p,
                                 unsigned long long seed, unsigned long long offset, int backward, hipStream_t st);
int sbgm_launch_mha_dropout_mask(float* mask, int B, int S, int heads, float p, unsigned long long seed, unsigned long long offset,
                                 hipStream_t st);
int sbgm_launch_attn_in(const float* x, const float* ln_g, const float* ln_b, const float* w_packed, const float* bias, float* qkv,
                        int M, int C, float eps, hipStream_t st);
int sbgm_launch_attn_out(const float* att, const float* x, const float* wo, const float* bo, const float* ln_g, const float* ln_b,
                         const float* w1, const float* b1, const float* w2, const float* b2, float* out, int M, int C, float eps,
                         hipStream_t st);

// ---- sampler.hip ---------------------------------------------------------------------------------------
struct StepScalars {       // one row of the device-side step table
    float t;               // time fed to the network at this step
    float g2;              // g(t)^2
    float dt;              // step size
    float noise;           // coefficient of the fresh N(0,1) draw in the predictor
    float t_next;          // time of the next step (written to the device time vector after the update)
};
struct SamplerState {      // device-resident, lets one captured graph serve every step
    unsigned long long step;        // advanced by the predictor kernel
    unsigned long long rng_offset;  // Philox counter base, advanced by every noise-drawing kernel
    unsigned long long seed;        // Philox key of the run: read from here when a state is given, so that a captured step is
                                    // reusable across runs with different seeds (the by-value seed argument serves eager calls)
    unsigned long long n_steps;     // length of the run (0: use the launch argument): the last step does not advance past it
};
// optional map from tile-local quads to domain-global Philox counters (null origins = plain element order)
struct NoiseMap {
    const int* origins;   // device [B][2] = (y0, x0) of each tile in the domain, x0 % 4 == 0
    int tile_h, tile_w4;  // tile rows, tile width / 4
    int dom_w4;           // ceil(domain width / 4)
};
int sbgm_launch_fill_t(float* t, float value, int B, hipStream_t st);
// `state` != null: scalars / RNG offset come from device memory (graph-replayable) and the state is advanced after
// the update; `state` == null: explicit by-value scalars and draw index.
int sbgm_launch_init_noise(float* x, float scale, const float* z, unsigned long long seed, SamplerState* state,
                           unsigned long long draw_index, size_t n, hipStream_t st, NoiseMap nm = NoiseMap{});
int sbgm_launch_em_update(float* x, float* x_mean, const float* score, const float* z, const StepScalars* table,
                          SamplerState* state, const StepScalars* sc_val, unsigned long long draw_index, float* t_dev,
                          unsigned long long seed, int B, size_t per_sample, int n_steps, hipStream_t st,
                          int t_entries = 0,    // entries of t_dev to refresh (0 -> B; 2B for the batched guidance pass)
                          NoiseMap nm = NoiseMap{});
int sbgm_launch_langevin(float* x, const float* score, const float* z, float snr_noise_norm, double* sumsq_ws,
                         SamplerState* state, unsigned long long draw_index, unsigned long long seed, int B,
                         size_t per_sample, hipStream_t st, NoiseMap nm = NoiseMap{});
int sbgm_launch_cfg_combine(float* out, const float* s_cond, const float* s_uncond, float scale, size_t n, hipStream_t st);

// ---- dsm_loss.hip (the loss around the network) -----------------------------------------------------------------------
int sbgm_dsm_nblk(int64_t per_sample);
int sbgm_launch_dsm_perturb(const float* x, const float* z_in, const float* t_in, const unsigned long long* rng,
                            unsigned long long seed, float t_eps, float sigma, float* xp, float* z_out, float* t_out, float* std_out, int B, size_t per, hipStream_t st);
int sbgm_launch_dsm_loss_fwd(const float* score, const float* z, const float* std, const float* sdf, double* partial_ws, float* loss,
                             unsigned long long* rng_advance, int B, size_t per, hipStream_t st);
int sbgm_launch_dsm_loss_bwd(const float* score, const float* z, const float* std, const float* sdf, const float* dloss, float* dscore,
                             int B, size_t per, hipStream_t st);

// ---- batch_pack.hip (before the network) ------------------------------------------------------------------------------
struct sbgm_assemble_args;
int sbgm_launch_assemble_conditions(const sbgm_assemble_args& a, hipStream_t st);

// ---- engine.hip: tile search for one convolution (used by the model's autotuner and by sbgm_conv2d_tune) -------------------
int sbgm_tune_conv(const ConvGeom& g, const ConvParams& p, float* partial, size_t partial_floats, hipStream_t st, ConvTile* best);

// ---- tiling.hip (full-domain tiles) -----------------------------------------------------------------------------------
int sbgm_launch_extract_tiles(const float* dom, const int* origins, float* tiles, int T, int C, int Hd, int Wd, int th, int tw,
                              hipStream_t st);
int sbgm_launch_stitch_tiles(const float* tiles, const int* origins, float* dom, int T, int C, int Hd, int Wd, int th, int tw,
                             int ramp_len, hipStream_t st);

// ---- postproc.hip (after the sampler) ---------------------------------------------------------------------------------
int sbgm_launch_pointwise_chain(const float* x, float* y, size_t n, int n_ops, const int* ops, const float* consts, hipStream_t st);
int sbgm_launch_sample_extremes(const float* x, int B, size_t per, float q, float* out_max, float* out_q, hipStream_t st);

// ---- backward.hip (training path) ------------------------------------------------------------------------------------
int sbgm_launch_conv_wgrad(const float* dy, const float* x, float* dw_oihw, float* dwp_ws, int B, int H, int W, int Cs, int Cin,
                           int Cout, int KH, int KW, int S, int PAD, hipStream_t st,
                           float* dbias = nullptr);   // optional bias gradient [Cout] (zeroed by the launcher unless pre-zeroed)
// OIHW operator [4*Cin][Cout][5][5] of the phase-decomposed data gradient of an 8x8/s2/p3 convolution (see backward.hip)
int sbgm_launch_dgrad_phase_weight(const float* w_oihw, float* out, int Cout, int Cin, hipStream_t st);
int sbgm_launch_colsum(const float* x, const float* y, float* out, int M, int C, hipStream_t st);
int sbgm_launch_samplesum(const float* x, float* out, int B, int HW, int C, hipStream_t st);
int sbgm_launch_groupnorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* skip,
                              const float* tbias, const float* mr, int act, float* dx, float* dskip, float* dgamma, float* dbeta,
                              float* dtbias, float* s12_ws, int B, int HW, int C, int G, hipStream_t st);
int sbgm_launch_batchnorm_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* tbias_after,
                              const float* mr, int relu, float* dx, float* dres, float* dgamma, float* dbeta, float* s12_ws, int B,
                              int HW, int C, hipStream_t st);
int sbgm_launch_layernorm_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta, int M, int C,
                              float eps, hipStream_t st, const float* dx_add = nullptr);   // dx_add [M, C] or null: summed into dx
int sbgm_launch_mha_core_bwd(const float* qkv, const float* dout, float* dqkv, int B, int S, int C, int heads, hipStream_t st);
int sbgm_launch_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, hipStream_t st);
int sbgm_launch_cout1_bwd(const float* dout, const float* a, const float* w_tap_c, const float* t, float sigma, float* da,
                          float* dw_tap_c, float* dbias, int B, int H, int W, int C, hipStream_t st);
int sbgm_launch_time_proj_bwd(const float* dout, const float* weight, const float* semb, const float* emb_raw, float* dW, float* dbias,
                              float* demb_accum, int B, int D, int ch, hipStream_t st);
int sbgm_launch_time_proj_multi_bwd(const float* const* douts, const float* const* sembs, float* const* dWs, float* const* dbs, const int* chs,
                                    int n_proj, int B, int D, hipStream_t st);
int sbgm_launch_label_emb_bwd(const float* demb, const int64_t* y, float* dtable, int B, int D, hipStream_t st);
int sbgm_launch_act_bwd(const float* x, const float* dy, float* dx, size_t n, int act, hipStream_t st);
int sbgm_launch_act_fwd(const float* x, float* y, size_t n, int act, hipStream_t st);
