/* sbgm_hip.h — C ABI of libsbgm_hip.so: the MI355X (gfx950) implementation of the SBGM_DANRA score-UNet forward
 * and reverse-SDE sampling hot path.
 *
 * Boundary: this library sits between the reference's Python layer L1 (sbgm/score_unet.py, sbgm/score_sampling.py)
 * and what used to be torch.nn ops.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless said
 * otherwise; every call enqueues on `stream` (a hipStream_t passed as void*) and returns without synchronising (the exceptions —
 * the tuning / profiling calls, a sampler's first call of a shape — say so where they are declared).
 * Return value: 0 = ok, non-zero = error (text via sbgm_last_error()).  The Python mirror raises RuntimeError.
 *
 * Tensor conventions at the boundary are the reference's: NCHW contiguous fp32, t float [B], y int64 [B] with
 * 0 = null class (reference score_unet.py:224-226).  Internally everything is NHWC; the *_nhwc per-op entry
 * points below expose that layout for unit parity tests.
 */
#ifndef SBGM_HIP_H
#define SBGM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sbgm_model sbgm_model;

const char* sbgm_last_error(void);
int sbgm_abi_version(void);

enum { SBGM_NONE = 0, SBGM_RELU = 1, SBGM_SILU = 2, SBGM_GELU = 3 };   /* activation codes */
enum { SBGM_NORM_INSTANCE = 0, SBGM_NORM_GROUP = 1 };
enum { SBGM_SAMPLER_EM = 0, SBGM_SAMPLER_PC = 1 };

/* ------------------------------------------------------------------------------------------------------------
 * Model handle.  Replaces: training_utils.get_model -> Encoder/Decoder/ScoreNet construction
 * (reference training_utils.py:645-666) + ScoreNet.forward (score_unet.py:829-879).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct sbgm_model_config {
    int struct_size;         /* = sizeof(sbgm_model_config) of the header the CALLER was compiled against; sbgm_model_create
                                rejects any other value (a binding generated from an older header would otherwise be read past
                                its end).  First member, so the check never reads beyond 4 bytes of a foreign struct. */
    int n_lsm_channels;      /* 0 or 2: lsm_cond value||mask          (cat order: x, lsm, topo, cond_img; :273-291) */
    int n_topo_channels;     /* 0 or 2 */
    int n_cond_channels;     /* LR condition channels */
    int time_embedding;      /* cfg.sampler.time_embedding */
    int block_layers[4];     /* cfg.sampler.block_layers */
    int n_heads;             /* cfg.sampler.num_heads */
    int num_classes;         /* 0 = no label embedding, else n_seasons (table has num_classes+1 rows) */
    int last_fmap_channels;  /* cfg.sampler.last_fmap_channels (512) */
    int decoder_norm;        /* SBGM_NORM_* (cfg.model.decoder_norm) */
    int gn_groups;           /* cfg.model.decoder_gn_groups */
    int decoder_activation;  /* SBGM_RELU / SBGM_SILU / SBGM_GELU (cfg.model.decoder_activation) */
    float sigma;             /* VE-SDE sigma (25.0, score_unet.py:932) */
    int decoder_transpose;   /* 0: bilinear x2 + conv_up (default); 1: ConvTranspose2d(k=2, s=2) upsampling, the reference's
                                ablation path (cfg.model.use_resize_conv = false, score_unet.py:470-475, :589) */
} sbgm_model_config;

/* Fails (non-zero, text in sbgm_last_error) when cfg->struct_size != sizeof(sbgm_model_config) of this library. */
int sbgm_model_create(const sbgm_model_config* cfg, sbgm_model** out);
/* sizeof(sbgm_model_config) as this library was compiled: lets a binding check its own struct before the first call. */
int sbgm_model_config_size(void);
void sbgm_model_destroy(sbgm_model* m);

/* Number of state_dict entries the model expects, and the i-th entry's name / element count (host strings). */
int sbgm_model_num_params(const sbgm_model* m);
const char* sbgm_model_param_name(const sbgm_model* m, int i);
int64_t sbgm_model_param_numel(const sbgm_model* m, int i);

/* Upload one state_dict tensor (device pointer, reference layout: OIHW convs, [out,in] linears, packed
 * mha.in_proj_weight [3C,C]).  The engine repacks into its own NHWC / K-major storage.  Replaces
 * nn.Module.load_state_dict for the path.  `num_batches_tracked` entries are accepted and ignored. */
int sbgm_model_set_param(sbgm_model* m, const char* name, const void* data, int64_t numel, void* stream);
/* Copy an engine-held tensor back in reference layout (used for BatchNorm running statistics after train-mode
 * forwards).  Only vector-shaped entries are supported. */
int sbgm_model_get_param(sbgm_model* m, const char* name, float* dst, int64_t numel, void* stream);
/* 0 when every expected entry has been uploaded; otherwise an error naming the first missing one. */
int sbgm_model_check_complete(const sbgm_model* m);
/* Bytes of activation workspace the handle currently owns.  The first evaluation of a shape runs on a generous bound; later calls
 * ask for the measured high-water mark and the surplus is returned (same addresses in the same order: bit-identical results). */
int64_t sbgm_model_workspace_bytes(const sbgm_model* m);

/* score = ScoreNet(x, t, y, cond_img, lsm_cond, topo_cond); out is NCHW [B,1,H,W].  Optional inputs may be NULL
 * when the model was configured with 0 channels for them.  bn_train != 0 uses batch statistics in the encoder's
 * BatchNorm layers and updates the engine-held running statistics (nn.Module.train() semantics).
 * fmaps (optional, may be NULL): 5 device pointers that receive the encoder feature maps in NHWC. */
int sbgm_model_forward(sbgm_model* m, const float* x, const float* t, const int64_t* y, const float* cond_img,
                       const float* lsm_cond, const float* topo_cond, float* out, float* const* fmaps, int B, int H,
                       int W, int bn_train, void* stream);

/* Whole reverse-SDE sampling loop on the device.  Replaces Euler_Maruyama_sampler / pc_sampler
 * (reference score_sampling.py:63-127, :136-230), including the classifier-free-guidance branch (guided_score_fn,
 * :10-56): with cfg_enabled the conditional and the unconditional evaluation of a step run as ONE batch of 2B samples
 * (rows B..2B-1: null class 0, zeroed cond_img, mask channel of the 2-channel geo fields zeroed) and are combined as
 * (1+w) s_c - w s_u.
 *   noise: NULL -> in-kernel Philox draws keyed by `seed`; else [n_draws][B][1][H][W] host-ordered N(0,1) draws
 *          consumed as the reference consumes torch.randn: init, then per step (corrector,) predictor.
 *   out:   x_mean after the last step, NCHW [B,1,H,W].
 *   use_graph: capture one SDE step into a hipGraph and replay it (ignored when noise != NULL). */
typedef struct sbgm_sampler_args {
    int kind;                /* SBGM_SAMPLER_* */
    int B, H, W;
    int num_steps;
    float eps;               /* smallest time (1e-3) */
    float snr;               /* PC only (0.16) */
    uint64_t seed;
    int use_graph;
    int bn_train;            /* literal reference launch_generation behaviour (generation.py:47) when != 0 */
    const int64_t* y;
    const float* cond_img;
    const float* lsm_cond;
    const float* topo_cond;
    const float* noise;
    float* out;
    int cfg_enabled;         /* classifier-free guidance on (cfg['classifier_free_guidance']['enabled']) */
    float cfg_scale;         /* guidance weight w of the predictor / Euler-Maruyama evaluation */
    float cfg_scale_corrector; /* w of the PC corrector evaluation (the reference clamps only this one to guidance_scale_max, :184-186) */
    /* Full-domain tiling (optional): the B samples are tiles of one domain.  tile_origins: device int32 [B][2] = (y0, x0),
     * x0 % 4 == 0; the in-kernel noise is then keyed by DOMAIN position, so overlapping tiles draw identical noise on the
     * pixels they share, and the Langevin step size of a tile uses that tile's own score norm (not the batch mean), so a tile's
     * result does not depend on which tiles share its batch or its GPU.  NULL = independent samples. */
    const int* tile_origins;
    int domain_w;
} sbgm_sampler_args;
/* Enqueues the whole reverse-SDE loop on `stream` and returns (with use_graph the step is captured once on a private stream and its
 * num_steps replays are launched on `stream`): the step table and the initial state are uploaded from pinned memory the handle owns,
 * so a steady-state call waits for nothing (two runs enqueued back to back simply execute in stream order).  It blocks the host only
 * in these cases: the first call of a (B, H, W) the handle has not seen (one measuring evaluation, then the workspace is sized), a call
 * that has to grow or trim the workspace or the step table (hipFree / hipMalloc), a change of the captured step's key while a replay of
 * the old graph is still executing, and guided runs (cfg_enabled), which free per-call condition copies at their end.  The handle's
 * workspace is the engine's: the caller passes no scratch memory. */
int sbgm_sampler_run(sbgm_model* m, const sbgm_sampler_args* a, void* stream);

/* Conv autotuning: time the tile / split-K candidates of every convolution of the (B,H,W) plan once and keep the
 * fastest.  Synchronises the stream.  Optional; without it a static heuristic is used. */
int sbgm_model_autotune(sbgm_model* m, int B, int H, int W, void* stream);
/* Persist / restore the tuned tile table (text file) so a service tunes once per problem size, not once per process.
 * Load merges into the current table; malformed lines are an error and leave the table untouched. */
int sbgm_model_tune_save(sbgm_model* m, const char* path);
int sbgm_model_tune_load(sbgm_model* m, const char* path);
/* Eager forward with each convolution launch bracketed by HIP events on `stream` (synchronises).  csv_path (host
 * string, may be NULL) receives one line per convolution. */
typedef struct sbgm_profile {
    float ms_total_with_events;  /* whole evaluation, including the event records */
    float ms_conv;               /* sum over implicit-GEMM convolution launches (+ their split-K reduce) */
    float ms_conv_max;           /* slowest single convolution */
    double flops_conv;           /* algorithmic FLOPs of those launches (2*M*Cout*K, unpadded K) */
    double flops_conv_max;
    int n_conv;
} sbgm_profile;
int sbgm_model_profile_forward(sbgm_model* m, const float* x, const float* t, const int64_t* y, const float* cond_img,
                               const float* lsm_cond, const float* topo_cond, float* out, int B, int H, int W,
                               sbgm_profile* summary, const char* csv_path, void* stream);
/* Device time of the last N forwards is not tracked here; bench.py brackets calls with HIP events it gets from: */
int sbgm_event_create(void** ev);
int sbgm_event_record(void* ev, void* stream);
int sbgm_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
int sbgm_event_destroy(void* ev);

/* ------------------------------------------------------------------------------------------------------------
 * Per-op entry points (NHWC fp32).  Each names the torch op instance it replaces.
 * ---------------------------------------------------------------------------------------------------------- */
/* torch.cat + .contiguous(NHWC): up to 4 NCHW sources -> [B,H,W,Cpad] (zero padded).  score_unet.py:273-291 */
int sbgm_pack_input(const float* const* srcs, const int* src_channels, int n_src, float* dst_nhwc, int B, int H,
                    int W, int c_pad, void* stream);
int sbgm_nchw_to_nhwc(const float* src, float* dst, int B, int H, int W, int C, void* stream);
int sbgm_nhwc_to_nchw(const float* src, float* dst, int B, int H, int W, int C, void* stream);

/* nn.Conv2d / nn.Linear weight repack: OIHW -> [k_step][Cout][16].  Returns packed float count via
 * sbgm_conv_packed_numel.  c_pad: padded input channels (4, 8 or a multiple of 16). */
int64_t sbgm_conv_packed_numel(int Cout, int KH, int KW, int c_pad);
int sbgm_conv_pack_weight(const float* w_oihw, float* packed, int Cout, int Cin, int KH, int KW, int c_pad, void* stream);
/* nn.Conv2d (+ folded BatchNorm scale/bias, + time bias, + residual, + ReLU) on the fp32 MFMA implicit-GEMM
 * kernel.  score_unet.py:206-219, torchvision BasicBlock convs, :468, :489; nn.Linear as 1x1 (:127-134).
 * tile_co/tile_px/splits/waves_per_tile = 0 -> defaults.  ws: split-K scratch (>= splits*M*Cout floats) or NULL if splits<=1. */
typedef struct sbgm_conv_args {
    const float* x;          /* [B,H,W,c_pad] */
    const float* w_packed;
    float* out;              /* [B,OH,OW,Cout] */
    const float* scale;      /* [Cout] or NULL */
    const float* bias;       /* [Cout] or NULL */
    const float* tbias;      /* [B,Cout] or NULL */
    const float* residual;   /* [B,OH,OW,Cout] or NULL */
    int B, H, W, c_pad, Cout;
    int KH, KW, stride, pad;
    int act;                 /* SBGM_NONE, SBGM_RELU or SBGM_GELU (exact erf), fused into the epilogue */
    int tbias_after_act;
    int tile_co, tile_px;    /* wave tile in 16-element fragments: {2,4} x {1,2,4}; 0 = default */
    int splits;              /* split-K over the grid (needs ws); 0/1 = off */
    int waves_per_tile;      /* in-workgroup split-K: 1, 2 or 4 waves share one tile (8 with winograd bit 0 alone); 0 = 1 */
    int winograd;            /* bit 0: Winograd F(2,3) weights/kernel (3x3/s1/p1; w_packed from sbgm_conv_wino_pack_weight;
                                tile_px counts 32-pixel fragments: {4,1} {2,2} {2,1} {4,2});
                                bit 1: LDS-staged kernel (W %% 16 == 0; tile_px = tile rows per wave, 2x that with bit 0);
                                bit 2 (with bit 1): two LDS stage buffers, one barrier per stage */
    int in_dil;              /* 0/1, or 2: read x through a zero-inserted grid (data gradient of a stride-2 conv) */
    int out_h, out_w;        /* explicit output size (required with in_dil = 2), else 0 */
    float* ws;
    int64_t ws_floats;
    /* LDS-staged kernel only (winograd bits 0+1): what happens to the input while the halo patch is staged, i.e. the
     * normalisation / resampling passes between the convolutions of a DecoderBlock (score_unet.py:583-615) without a pass of
     * their own.  0 = plain.  1 = affine on load: the convolution sees x*scale + shift per (sample, channel) (a GroupNorm of x,
     * in_affine from sbgm_groupnorm_finalize), zero padding kept.  2 = nn.Upsample(x2, bilinear, align_corners=False) on load:
     * x is the LOW-resolution map [B,H/2,W/2,c_pad] (H, W = the convolution's size), optionally act(x*scale + shift + in_skip)
     * first. */
    int in_mode;
    const float* in_affine;  /* [B][c_pad/4][2][4] scale quad, shift quad; required for mode 1, optional for mode 2 */
    const float* in_skip;    /* mode 2: [B,H/2,W/2,c_pad] or NULL */
    int in_act;              /* mode 2: SBGM_NONE / RELU / SILU / GELU applied to the low-resolution value */
    /* Optional second weight image in the Winograd layout (sbgm_conv_wino_pack_weight / a sbgm_pack_desc with transposed bit 1).
     * When given: sbgm_conv2d_tune also times the Winograd candidates, and sbgm_conv2d_fwd takes its weights from here when
     * winograd bit 0 is set (w_packed stays the implicit-GEMM image for every other kernel).  NULL: as before, a call with
     * winograd bit 0 reads the Winograd image from w_packed. */
    const float* w_wino;
    /* Optional third weight image for the 2-D Winograd F(2x2,3x3) LDS kernel (sbgm_conv_wino2d_pack_weight).  winograd bit 3 selects
     * that kernel: 16x16-pixel tiles, tile_co in {1, 2} (16 / 32 output channels per workgroup), tile_px / splits ignored,
     * waves_per_tile = 2 picks the build whose registers are held to two workgroups per CU, bit 2 = two LDS stage buffers, bit 4
     * (16) = the persistent form (two workgroups per CU walk over the tiles, weight slab by LDS-DMA); needs W %% 16 == 0 and an
     * even H; all three in_mode values.  Weights: w_wino2d when given, else w_packed must be that image.
     * When given, sbgm_conv2d_tune also times these candidates (tile[4] == 2 then means bit 3; tile[5] == 3 means bit 4). */
    const float* w_wino2d;
} sbgm_conv_args;
int sbgm_conv2d_fwd(const sbgm_conv_args* a, void* stream);
/* Times the kernel / tile / split candidates for exactly this call (same operands; launches are idempotent; synchronises)
 * and writes the fastest as tile[6] = {tile_co, tile_px, splits, waves_per_tile, winograd bit 0, LDS kernel: 0 off / 1 on (bit 1) / 2 double-buffered (bits 1+2)}, the values
 * to put into sbgm_conv_args.  Winograd candidates are timed when a->w_wino is given (they need the transformed weights);
 * split-K candidates are considered when a->ws is given.  Used by the training path, whose convolutions run op by op. */
int sbgm_conv2d_tune(const sbgm_conv_args* a, int* tile, void* stream);
/* Pack many convolution weights in one launch.  desc: DEVICE array of n descriptors; block_begin = exclusive prefix of
 * sbgm_conv_pack_weights_batched_blocks(Cout, KH, KW, cs) over the descriptors (the workgroups each weight needs),
 * total_blocks = its total; nsteps = sbgm_conv_packed_numel / (Cout*16).
 * transposed bit 0 packs the data-gradient operator: then Cout/Cin are the TRANSPOSED sizes (Cout = forward Cin, Cin = forward
 * Cout) and cs is the padded forward Cout, exactly as sbgm_conv_pack_weight_dgrad does for one weight.
 * transposed bit 1 (3x3 kernels, cs %% 16 == 0, Cout %% 16 == 0): dst is the Winograd image U[kh][cs/16][xi][Cout][16] of
 * sbgm_conv_wino_pack_weight (sbgm_conv_wino_packed_numel floats) of that operator instead; nsteps is ignored.
 * transposed bit 2 (same conditions): dst is the F(2x2,3x3) image of sbgm_conv_wino2d_pack_weight (sbgm_conv_wino2d_packed_numel). */
typedef struct sbgm_pack_desc {
    const float* src;      /* OIHW */
    float* dst;            /* packed [nsteps][Cout][16] */
    int Cout, Cin, KH, KW, cs, nsteps, transposed, block_begin;
} sbgm_pack_desc;
int sbgm_conv_pack_weights_batched_blocks(int Cout, int KH, int KW, int c_pad);
int sbgm_conv_pack_weights_batched(const sbgm_pack_desc* desc_dev, int n, int total_blocks, void* stream);
/* Process-wide switch for the backward launchers (sbgm_conv2d_wgrad[_bias], sbgm_groupnorm_bwd, sbgm_batchnorm_bwd,
 * sbgm_layernorm_bwd's dgamma/dbeta, sbgm_samplesum's output, sbgm_mha_core_bwd's dqkv, sbgm_batchnorm_train_fwd's sums): 1 = the caller
 * hands in already-zeroed scratch and the launchers skip their own memsets.  Returns the previous value. */
int sbgm_set_scratch_prezeroed(int on);
/* Deferred weight-gradient layout passes.  sbgm_conv2d_wgrad[_bias] accumulates most gradients in a [tap][Cout][c_pad] slab (ws)
 * and converts it to OIHW with a small layout launch.  With sbgm_wgrad_defer(1) in force that launch is queued instead, and
 * sbgm_wgrad_flush runs every queued conversion as ONE launch (21 per training step of the default model): the caller keeps the
 * queued ws / dw_oihw buffers alive and untouched until the flush, after which dw_oihw holds the gradients.  Process-wide, returns
 * the previous value; sbgm_wgrad_flush_pending() = number of queued conversions.
 * `on` is a bit mask: bit 0 = the layout passes as above; bit 1 (2) = also queue the weight-gradient GEMMs of the layers the per-tap
 * split-K kernel serves (1x1 / linear, 3x3 stride 2, 3x3 on 4x4 maps, the 8x8 stems) and run them as ONE batched launch at the flush —
 * the caller then keeps dy, x, ws and dw_oihw (and dbias) of every call made under the flag alive and untouched until the flush. */
int sbgm_wgrad_defer(int on);
int sbgm_wgrad_flush(void* stream);
int sbgm_wgrad_flush_pending(void);
/* Forget the queued conversions without running them (returns how many): after a backward pass that raised, the queued dw_oihw
 * destinations may already be freed, and the next backward recomputes every gradient anyway. */
int sbgm_wgrad_discard(void);
/* Winograd F(2,3)-along-rows weight transform for 3x3 kernels: OIHW -> U[kh][c/16][xi][Cout][16] */
int64_t sbgm_conv_wino_packed_numel(int Cout, int c_pad);
int sbgm_conv_wino_pack_weight(const float* w_oihw, float* packed, int Cout, int Cin, int c_pad, void* stream);

/* Winograd F(2x2,3x3) weight transform for 3x3 kernels: OIHW -> U[c/16][xi*4+eta][Cout][16], U = G g G^T */
int64_t sbgm_conv_wino2d_packed_numel(int Cout, int c_pad);
int sbgm_conv_wino2d_pack_weight(const float* w_oihw, float* packed, int Cout, int Cin, int c_pad, void* stream);

/* ConvTranspose2d(k=2,s=2) = one 1x1 convolution to 4C channels (weights from sbgm_tconv_weight_to_oihw, bias repeated
 * 4x) followed by depth->space; its backward is space->depth followed by the 1x1 convolution's backward. */
int sbgm_depth_to_space2(const float* x /* [B,H,W,4C] */, float* y /* [B,2H,2W,C] */, int B, int H, int W, int C, void* stream);
int sbgm_space_to_depth2(const float* y /* [B,2H,2W,C] */, float* x /* [B,H,W,4C] */, int B, int H, int W, int C, void* stream);
/* The same permutation for any stride s in 1..16 (ConvTranspose2d(k = s, stride = s): DecoderBlock(upsample_scale = s, use_resize_conv =
 * False) called on its own): depth [B,H,W,s*s*C] with channels ordered (dy, dx, c) <-> space [B,s*H,s*W,C]. */
int sbgm_depth_to_space(const float* x, float* y, int B, int H, int W, int C, int s, void* stream);
int sbgm_space_to_depth(const float* y, float* x, int B, int H, int W, int C, int s, void* stream);
int sbgm_tconv_weight_to_oihw(const float* w /* [Cin,Cout,2,2] */, float* oihw /* [4*Cout,Cin,1,1] */, int Cin, int Cout,
                              void* stream);
/* nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False).  score_unet.py:467 */
int sbgm_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
/* nn.GroupNorm / nn.InstanceNorm2d (+ skip add, + time bias, + activation).  score_unet.py:480-483, :585-615.
 * gamma/beta NULL = no affine (InstanceNorm2d default).  stats_ws: >= 1024*B*G bytes. */
int sbgm_groupnorm_fwd(const float* x, float* y, const float* gamma, const float* beta, const float* skip,
                       const float* tbias, int act, int B, int HW, int C, int G, float eps, void* stats_ws,
                       float* mean_rstd_out /* [B,G,2] or NULL */, void* stream);
/* GroupNorm statistics only (the chunk partials sbgm_groupnorm_fwd computes first): stats_ws as above; *chunks receives the
 * number of partials per (sample, group).  Then sbgm_groupnorm_finalize turns them into the per-(sample, channel) affine a
 * consumer convolution applies on load (sbgm_conv_args.in_affine): out[b][c/4][0][c%4] = rstd*gamma,
 * out[b][c/4][1][c%4] = beta - mean*rstd*gamma (+ tbias[b][c]).  gamma/beta NULL = no affine. */
int sbgm_groupnorm_stats(const float* x, void* stats_ws, int B, int HW, int C, int G, int* chunks, void* stream);
int sbgm_groupnorm_finalize(const void* stats_ws, int chunks, const float* gamma, const float* beta, const float* tbias,
                            float* out /* [B][C][2] */, int B, int HW, int C, int G, float eps, void* stream);
/* nn.LayerNorm(C).  score_unet.py:128-129 */
int sbgm_layernorm_fwd(const float* x, float* y, const float* gamma, const float* beta, int M, int C, float eps, void* stream);
/* nn.BatchNorm2d in training mode (+ residual, ReLU, time bias).  stats_ws: >= 24*C bytes; its first 16*C bytes are the
 * fp64 sums and must arrive zeroed when sbgm_set_scratch_prezeroed(1) is in force.  mean_rstd_out [C,2] (for
 * sbgm_batchnorm_bwd) or NULL: the pairs are then left behind the sums, at stats_ws + 16*C bytes. */
int sbgm_batchnorm_train_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, const float* residual, const float* tbias_after, int relu, int B, int HW,
                             int C, float eps, float momentum, void* stats_ws, float* mean_rstd_out, void* stream);
/* SyncBatchNorm (new capability: the reference is single-device, so its BatchNorm sees the whole batch, score_unet.py:323; with
 * the batch sharded over ranks the statistics must be summed over the ranks to reproduce that step).  The forward above in two
 * halves: _stats leaves the per-channel fp64 sums (x, x^2) of the local batch in stats_ws[2*C]; the caller sums stats_ws over
 * the ranks (one RCCL all-reduce of 2*C doubles); _apply finalises with n_total = sum over ranks of B*HW. */
int sbgm_batchnorm_train_stats(const float* x, int B, int HW, int C, void* stats_ws, void* stream);
int sbgm_batchnorm_train_apply(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, const float* residual, const float* tbias_after, int relu, int B, int HW, int C,
                               float eps, float momentum, void* stats_ws, double n_total, float* mean_rstd_out, void* stream);
/* Core of nn.MultiheadAttention between in_proj and out_proj: qkv [B,S,3C] -> [B,S,C].  score_unet.py:142 */
int sbgm_mha_core_fwd(const float* qkv, float* out, int B, int S, int C, int heads, void* stream);
/* The same core with TRAIN-MODE DROPOUT on the softmax probabilities — nn.MultiheadAttention(dropout = p), the `dropout` argument of
 * ImageSelfAttention (score_unet.py:118-127): out = (softmax(q k^T / sqrt(d)) o D) v with D_ij = keep_ij / (1 - p), keep_ij drawn from a
 * Philox stream keyed by (seed, offset, ((b heads + h) S + i) S + j).  _bwd needs the same (p, seed, offset) and the forward's qkv; dqkv
 * [B,S,3C] is accumulated with atomics (zeroed by the call unless sbgm_set_scratch_prezeroed(1)).  sbgm_mha_dropout_mask writes D
 * [B,heads,S,S] (how the tests compare against an explicit-mask evaluation).  0 <= p < 1; 32 * S bytes of LDS per workgroup. */
int sbgm_mha_core_dropout_fwd(const float* qkv, float* out, int B, int S, int C, int heads, float p, uint64_t seed, uint64_t offset,
                              void* stream);
int sbgm_mha_core_dropout_bwd(const float* qkv, const float* dout, float* dqkv, int B, int S, int C, int heads, float p, uint64_t seed,
                              uint64_t offset, void* stream);
int sbgm_mha_dropout_mask(float* mask, int B, int S, int heads, float p, uint64_t seed, uint64_t offset, void* stream);
/* The per-token halves of ImageSelfAttention (score_unet.py:141-145) as single launches over token tiles, C in {64,128,256,512}:
 *   sbgm_attn_qkv_fwd : qkv[M][3C] = LayerNorm(x; ln_gamma, ln_beta) . in_proj_weight^T + in_proj_bias          (self.ln1 + mha in_proj)
 *   sbgm_attn_tail_fwd: h = x + att . out_proj^T + b_out;  out = h + ff[2](GELU(ff[0](LayerNorm(h))))           (:142-145)
 * Weights are sbgm_conv_pack_weight images of the [Cout][C][1][1] matrices; `out` may alias `x`, not `att`. */
int sbgm_attn_qkv_fwd(const float* x, const float* ln_gamma, const float* ln_beta, const float* w_in_packed, const float* b_in,
                      float* qkv, int M, int C, float eps, void* stream);
int sbgm_attn_tail_fwd(const float* att, const float* x, const float* w_out_packed, const float* b_out, const float* ln_gamma,
                       const float* ln_beta, const float* w_ff1_packed, const float* b_ff1, const float* w_ff2_packed,
                       const float* b_ff2, float* out, int M, int C, float eps, void* stream);
/* SinusoidalEmbedding (+ label embedding) -> SiLU -> Linear, for one projection.  score_unet.py:41-45, :377-381 */
int sbgm_time_proj_fwd(const float* t, const int64_t* y, const float* label_emb, const float* freqs, const float* weight,
                       const float* bias, float* out, float* emb_ws /* [B,D] silu(emb) */, float* emb_raw /* [B,D] or NULL */,
                       int B, int D, int ch, void* stream);
/* Several projections in one launch pair (embeddings | projections): n_emb embeddings (label_emb is added to embedding 0), n_proj
 * projections, projection i reads embedding emb_index[i] and writes outs[i] [B, chs[i]].  emb_ws / emb_raw: [n_emb][B][D]
 * (silu(emb) / emb; emb_raw may be NULL).  n_emb <= 8, n_proj <= 16.  The encoder's 5 projections share one embedding
 * (score_unet.py:301-308); each decoder block has its own (:606-609). */
int sbgm_time_proj_multi_fwd(const float* t, const int64_t* y, const float* label_emb, const float* const* freqs, int n_emb,
                             const float* const* weights, const float* const* biases, float* const* outs, const int* chs,
                             const int* emb_index, int n_proj, float* emb_ws, float* emb_raw, int B, int D, void* stream);
/* final_layer.conv (3x3, C->1) + division by marginal_prob_std(t).  score_unet.py:489, :876-877.
 * w_tap_c from sbgm_cout1_pack_weight; t NULL = no division. */
int sbgm_cout1_pack_weight(const float* w_oihw, float* w_tap_c, int C, void* stream);
int sbgm_conv3x3_cout1_fwd(const float* x, const float* w_tap_c, const float* bias, const float* t, float sigma, float* out,
                           int B, int H, int W, int C, void* stream);
/* Elementwise activation in place (GELU between the attention FF linears, score_unet.py:132). */
int sbgm_act_inplace(float* x, int64_t n, int act, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Training path: what `loss.backward()` (reference training.py:403-405) needs from each op above.
 * Data gradients of convolutions / linears run sbgm_conv2d_fwd on weights packed by sbgm_conv_pack_weight_dgrad
 * (stride-2 layers with in_dil = 2 and an explicit output size).
 * ---------------------------------------------------------------------------------------------------------- */
int sbgm_conv_pack_weight_dgrad(const float* w_oihw, float* packed, int Cout, int Cin, int KH, int KW, void* stream);
/* Data gradient of an 8x8 / stride-2 / pad-3 convolution (the encoder's conv2, score_unet.py:214-219) by output phase instead of
 * through a zero-inserted dy: builds the OIHW operator [4*Cin][Cout][5][5] whose 5x5 / stride-1 / pad-2 convolution over dy
 * [B,OH,OW,Cout] yields the 4 phases (py, px, ci) of dx; sbgm_depth_to_space2 interleaves them into dx [B,2*OH,2*OW,Cin].
 * 2.56x fewer MACs than the dilated form.  Pack the result with sbgm_conv_pack_weight(.., 4*Cin, Cout, 5, 5, Cout). */
int sbgm_conv8x8s2_dgrad_phase_weight(const float* w_oihw, float* out_oihw, int Cout, int Cin, void* stream);
/* dW (OIHW) = sum_p dy[p,:] (x) x[p@tap,:]; ws: >= KH*KW*Cout*c_pad floats.  For a 1x1 kernel with c_pad == Cin (and c_pad % 64
 * == 0) ws may be dw_oihw itself: the partial sums then meet directly in the gradient and the layout pass is skipped. */
int sbgm_conv2d_wgrad(const float* dy, const float* x, float* dw_oihw, float* ws, int B, int H, int W, int c_pad, int Cin,
                      int Cout, int KH, int KW, int stride, int pad, void* stream);
/* the same, plus the bias gradient db[Cout] = sum_p dy[p,:] produced by the waves that stream dy anyway */
int sbgm_conv2d_wgrad_bias(const float* dy, const float* x, float* dw_oihw, float* dbias, float* ws, int B, int H, int W, int c_pad,
                           int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream);
/* `optimizer.step()` of torch.optim.Adam (decoupled = 0: weight decay as an L2 term on the gradient) or AdamW (decoupled = 1)
 * over every parameter tensor in one launch (reference training.py:407, training_utils.py:50-59).  desc: DEVICE array of n
 * descriptors, block_begin = exclusive prefix of sbgm_adam_step_blocks(numel), total_blocks = its total.  step: DEVICE
 * scalar holding the step count t of THIS update (the caller increments it before the call), used for the bias corrections
 * 1 - beta^t.  Update: m += (1-b1)(g'-m); v = b2 v + (1-b2) g'^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps). */
typedef struct sbgm_adam_desc {
    float* p;              /* parameter */
    const float* g;        /* gradient */
    float* m;              /* exp_avg */
    float* v;              /* exp_avg_sq */
    int64_t numel;
    int block_begin, reserved;
} sbgm_adam_desc;
int sbgm_adam_step_blocks(int64_t numel);
/* grad_scale multiplies every gradient as it is read (1, or 1/world when the gradients are the data-parallel SUM over the
 * replicas: the averaging of reference-style data parallelism without a pass of its own). */
int sbgm_adam_step_batched(const sbgm_adam_desc* desc_dev, int n, int total_blocks, const float* step, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int decoupled, float grad_scale, void* stream);
int sbgm_colsum(const float* x, const float* y /* NULL or multiplied elementwise */, float* out, int M, int C, void* stream);
int sbgm_samplesum(const float* x, float* out /* [B,C] */, int B, int HW, int C, void* stream);
/* ws: >= 8*B*C bytes */
int sbgm_groupnorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* skip,
                       const float* tbias, const float* mean_rstd, int act, float* dx, float* dskip, float* dgamma,
                       float* dbeta, float* dtbias, float* ws, int B, int HW, int C, int G, void* stream);
int sbgm_batchnorm_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* tbias_after,
                       const float* mean_rstd /* [C,2] */, int relu, float* dx, float* dres, float* dgamma, float* dbeta,
                       float* ws, int B, int HW, int C, void* stream);
/* SyncBatchNorm backward in two halves: _reduce leaves the local sums (g, g*xhat) per (sample, channel) in ws [B][C][2]; the caller
 * sums them over samples and ranks into sync_sums [C][2] (one all-reduce of 2*C floats); _apply uses sync_sums / n_total for the two
 * means (sync_sums NULL: the local ws and B*HW, i.e. exactly sbgm_batchnorm_bwd).  dgamma / dbeta stay LOCAL sums: they are
 * averaged over the ranks with every other parameter gradient. */
int sbgm_batchnorm_bwd_reduce(const float* x, const float* dy, const float* y, const float* tbias_after, const float* mean_rstd,
                              int relu, float* ws, int B, int HW, int C, void* stream);
int sbgm_batchnorm_bwd_apply(const float* x, const float* dy, const float* y, const float* gamma, const float* tbias_after,
                             const float* mean_rstd, int relu, float* dx, float* dres, float* dgamma, float* dbeta, const float* ws,
                             const float* sync_sums, double n_total, int B, int HW, int C, void* stream);
/* dx_add [M,C] or NULL: added to dx (the gradient reaching x along its other consumer, e.g. the residual around the block) */
int sbgm_layernorm_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta, int M,
                       int C, float eps, const float* dx_add, void* stream);
/* zero `bytes` (multiple of 4) of device memory with a kernel on `stream` (graph-capture safe; see common.h on hipMemsetAsync) */
int sbgm_fill_zero(void* p, int64_t bytes, void* stream);
int sbgm_mha_core_bwd(const float* qkv, const float* dout, float* dqkv, int B, int S, int C, int heads, void* stream);
int sbgm_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream);
/* nn.Upsample(scale_factor = scale, mode="bilinear", align_corners=False) over NHWC for any integer scale in 1..16 — DecoderBlock's
 * `upsample_scale` argument (score_unet.py:420, :467; the Decoder only ever passes 2, which has the kernels above).  x [B,H,W,C] ->
 * y [B,scale*H,scale*W,C]; _bwd: dy [B,scale*H,scale*W,C] -> dx [B,H,W,C] (a gather: no atomics, no zeroing needed). */
int sbgm_upsample_bilinear_fwd(const float* x, float* y, int B, int H, int W, int C, int scale, void* stream);
int sbgm_upsample_bilinear_bwd(const float* dy, float* dx, int B, int H, int W, int C, int scale, void* stream);
int sbgm_conv3x3_cout1_bwd(const float* dout, const float* a, const float* w_tap_c, const float* t, float sigma, float* da,
                           float* dw_tap_c, float* dbias, int B, int H, int W, int C, void* stream);
int sbgm_time_proj_bwd(const float* dout, const float* weight, const float* semb, const float* emb_raw, float* dW,
                       float* dbias, float* demb_accum /* NULL or [B,D], accumulated */, int B, int D, int ch, void* stream);
/* dW / dbias of up to 16 projections in one launch (HOST arrays of device pointers; projection i: douts[i] [B, chs[i]], sembs[i] [B, D]
 * = silu(embedding) it read, dWs[i] [chs[i], D], dbiases[i] [chs[i]]).  The embedding gradient (label embedding only) stays with
 * sbgm_time_proj_bwd. */
int sbgm_time_proj_multi_bwd(const float* const* douts, const float* const* sembs, float* const* dWs, float* const* dbiases,
                             const int* chs, int n_proj, int B, int D, void* stream);
int sbgm_label_emb_bwd(const float* demb, const int64_t* y, float* dtable, int B, int D, void* stream);
int sbgm_act_fwd(const float* x, float* y, int64_t n, int act, void* stream);
int sbgm_act_bwd(const float* x, const float* dy, float* dx, int64_t n, int act, void* stream);

/* ---- the loss around the network: loss_fn (reference score_unet.py:936-985) --------------------------------------------
 * perturb: t_b = U(0,1)*(1-t_eps)+t_eps, z ~ N(0,1), std_b = marginal_prob_std(t_b), x_perturbed = x + std_b*z  (:957-963).
 *   z / t given (device tensors [B,1,H,W] / [B]): used verbatim (parity runs inject the reference's draws; z_out is then not
 *   written and may be NULL).  Either NULL: drawn in the kernel with Philox keyed by `seed` (rng_state NULL; eager calls pass a
 *   fresh seed per call) or by rng_state = device uint64[2] {seed, offset}; sbgm_dsm_loss_fwd advances that offset by one, so a
 *   captured (hipGraph) step replays with fresh noise.
 * loss_fwd: loss[0] = mean_b sum_{chw} w*(score*std_b + z)^2, w = sigmoid(sdf)*0.5+0.5 or 1 (sdf NULL)           (:974-984);
 *   partial_ws: >= 8 * B * sbgm_dsm_loss_blocks(per_sample) bytes (one fp64 partial per workgroup, summed in fixed order).
 * loss_bwd: dscore = dloss[0] * (2/B) * w * (score*std_b + z) * std_b   (dloss: DEVICE scalar, autograd's grad_output). */
/* perturb with another schedule (loss_fn takes any marginal_prob_std callable, :936-985): call once with per_sample = 0 (only t_out is
 * drawn / copied), evaluate the schedule on t_out into std_out, call again with t = t_out and sigma <= 0: std_out is then READ. */
int sbgm_dsm_loss_blocks(int64_t per_sample);
int sbgm_dsm_perturb(const float* x, const float* z, const float* t, const uint64_t* rng_state, uint64_t seed, float t_eps, float sigma,
                     float* x_perturbed, float* z_out, float* t_out /* [B] */, float* std_out /* [B] */, int B,
                     int64_t per_sample, void* stream);
int sbgm_dsm_loss_fwd(const float* score, const float* z, const float* std, const float* sdf, void* partial_ws, float* loss,
                      uint64_t* rng_state /* advanced; may be NULL */, int B, int64_t per_sample, void* stream);
int sbgm_dsm_loss_bwd(const float* score, const float* z, const float* std, const float* sdf, const float* dloss, float* dscore,
                      int B, int64_t per_sample, void* stream);

/* Sampler updates with explicit scalars (the fused loop above uses a device-side table instead).
 * z NULL -> Philox draw keyed by (seed, draw_index).
 *   em:       x_mean = x + (g2*score)*dt ; x = x_mean + noise_coef*z            score_sampling.py:124-125, :224-227
 *   langevin: eps = 2*(snr_noise_norm / mean_b||score_b||)^2 ; x += eps*score + sqrt(2 eps)*z      :200-204
 *   cfg:      out = (1+w)*s_cond - w*s_uncond                                                        :55      */
int sbgm_em_step(float* x, float* x_mean, const float* score, const float* z, float g2, float dt, float noise_coef,
                 uint64_t seed, uint64_t draw_index, int64_t n, void* stream);
int sbgm_langevin_step(float* x, const float* score, const float* z, float snr_noise_norm, void* sumsq_ws /* >= 8*B B */,
                       uint64_t seed, uint64_t draw_index, int B, int64_t per_sample, void* stream);
int sbgm_cfg_combine(float* out, const float* s_cond, const float* s_uncond, float scale, int64_t n, void* stream);
int sbgm_randn_scaled(float* x, float scale, uint64_t seed, uint64_t draw_index, int64_t n, void* stream);

/* ---- after the sampler -------------------------------------------------------------------------
 * sbgm_pointwise_chain: y[i] = program(x[i]); the program is at most SBGM_CHAIN_MAX_OPS scalar steps, each rounded to fp32
 * on its own exactly like the reference's tensor-scalar arithmetic.  Replaces the __call__ bodies of Scale /
 * ScaleBackTransform (special_transforms.py:87-100, :125-139), ZScoreTransform / ZScoreBackTransform (:159-184, :202-233),
 * PrcpLogTransform / PrcpLogBackTransform (:288-355, :418-462) and the generation clamp (training.py:744-748).
 * x == y (in place) is allowed. */
enum { SBGM_OP_ADD = 0, SBGM_OP_MUL = 1, SBGM_OP_DIV = 2, SBGM_OP_CLAMP_MIN = 3, SBGM_OP_CLAMP_MAX = 4, SBGM_OP_EXP = 5,
       SBGM_OP_LOG = 6 };
#define SBGM_CHAIN_MAX_OPS 12
int sbgm_pointwise_chain(const float* x, float* y, int64_t n, int n_ops, const int* ops, const float* consts, void* stream);
/* Per-sample maximum and q-quantile (torch.quantile "linear" interpolation, NaN-propagating) of x[B][per_sample]:
 * the two statistics report_precip_extremes needs (utils.py:1647-1649).  out_max, out_q: device [B]. */
int sbgm_sample_extremes(const float* x, int B, int64_t per_sample, float q, float* out_max, float* out_q, void* stream);

/* ---- before the network ------------------------------------------------------------------------
 * Batch-level condition assembly: channel concatenation of the sorted *_lr fields (utils.py:441-447), classifier-free-
 * guidance condition dropout (data_modules.py:957-983: LR fields -> 0, class label -> NULL token 0) and the value||mask
 * layout of the geo fields (data_modules.py:971-993: mask 0 when dropped, 1 otherwise), one launch group per batch.
 * All tensors NCHW fp32 on the device; dropped: uint8 [B] (NULL = nothing dropped); any input group may be NULL/absent. */
#define SBGM_ASSEMBLE_MAX_LR 16
typedef struct sbgm_assemble_args {
    int B;
    int64_t HW;                               /* H*W, a multiple of 4 */
    int n_lr;                                 /* number of LR fields, already in sorted-key order */
    const float* lr[SBGM_ASSEMBLE_MAX_LR];    /* each [B][lr_channels[k]][H][W] */
    int lr_channels[SBGM_ASSEMBLE_MAX_LR];
    const float* lsm;  int lsm_channels;      /* [B][1 or 2][H][W] */
    const float* topo; int topo_channels;
    const int64_t* y;                         /* [B] class labels */
    const unsigned char* dropped;             /* [B] */
    float* lr_out;                            /* [B][sum lr_channels][H][W] */
    float* lsm_out;                           /* [B][2][H][W] */
    float* topo_out;                          /* [B][2][H][W] */
    int64_t* y_out;                           /* [B] */
} sbgm_assemble_args;
int sbgm_assemble_conditions(const sbgm_assemble_args* a, void* stream);

/* ---- full-domain tiling (no reference counterpart, specification in DESIGN.md) -----------------
 * origins: device int32 [T][2] = (y0, x0) of every tile; the host validates that each tile lies inside the domain.
 * extract: tiles[T][C][th][tw] <- domain[C][Hd][Wd].   stitch: domain <- normalised blend of the covering tiles with
 * linear ramps of `ramp_len` pixels on tile edges that are not domain edges. */
int sbgm_extract_tiles(const float* domain, const int* origins, float* tiles, int T, int C, int Hd, int Wd, int th, int tw,
                       void* stream);
int sbgm_stitch_tiles(const float* tiles, const int* origins, float* domain, int T, int C, int Hd, int Wd, int th, int tw,
                      int ramp_len, void* stream);

#ifdef __cplusplus
}
#endif
#endif
