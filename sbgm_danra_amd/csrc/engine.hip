// Host-side engine: owns the repacked weights and the activation workspace, builds the launch sequence of one
// ScoreNet evaluation for a given (B, H, W) and runs the reverse-SDE sampler loops (optionally as a replayed
// hipGraph).  Mirrors, at the launch-sequence level, reference sbgm/score_unet.py:247-364 (Encoder.forward),
// :559-627 (DecoderBlock.forward), :733-758 (Decoder.forward), :829-879 (ScoreNet.forward) and
// sbgm/score_sampling.py:63-127 / :136-230.
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/sbgm_hip.h"
#include "common.h"
#include "kernels.h"

namespace {

constexpr float BN_EPS = 1e-5f, BN_MOMENTUM = 0.1f, GN_EPS = 1e-5f, LN_EPS = 1e-5f;
const int FMAP_CH[5] = {64, 64, 128, 256, 512};   // score_unet.py:198

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int pad_channels(int c) { return c <= 4 ? 4 : c <= 8 ? 8 : (int)align_up(c, 16); }

// ---- parameters ------------------------------------------------------------------------------------------------
enum ParamKind { P_VEC, P_CONV, P_COUT1, P_IGNORE, P_TCONV, P_VEC4 };   // TCONV: ConvTranspose2d(2,2) weight; VEC4: vector stored 4x
struct Param {
    std::string name;
    ParamKind kind = P_VEC;
    int64_t numel = 0;          // element count in reference layout
    int cout = 0, cin = 0, kh = 1, kw = 1, cs = 0;   // P_CONV geometry (cs = padded Cin)
    float* dev = nullptr;       // engine storage (packed for P_CONV / P_COUT1)
    size_t dev_floats = 0;
    bool wino = false;          // 3x3 stride-1 conv: keep a Winograd F(2,3) packed copy as well
    float* dev_wino = nullptr;
    size_t wino_floats = 0;
    float* dev_w2d = nullptr;   // ... and a 2-D Winograd F(2x2,3x3) packed copy (conv_w2d.hip)
    size_t w2d_floats = 0;
    bool filled = false;
};

struct ConvW { Param* w = nullptr; Param* b = nullptr; };          // weight (+ optional bias)
struct BNW { Param *g, *b, *rm, *rv; float *scale, *bias; };       // + folded eval scale/bias
struct AttnW { Param *ln1g, *ln1b, *ln2g, *ln2b, *inw, *inb, *outw, *outb, *f1w, *f1b, *f2w, *f2b; int C; };
struct BlockW { ConvW c1, c2, ds; BNW bn1, bn2, dsbn; bool has_ds; int cin, cout, stride; };
struct DecW { ConvW up, conv; Param *n1g, *n1b, *n2g, *n2b, *freq, *tpw, *tpb; AttnW attn; bool has_attn; int cin, cout; };

// one launch of a tuned tile: picks the kernel family and the weight image it reads
int launch_tile(const ConvGeom& g, ConvParams p, const ConvTile& ct, float* partial, hipStream_t st) {
    if (ct.wino == 2) { p.wp = p.wp_w2d; return sbgm_launch_conv_w2d(p, ct, st); }
    if (!ct.wino && !ct.lds) return sbgm_launch_conv(g, p, ct, partial, st);
    if (ct.wino) p.wp = p.wp_wino;
    return ct.lds ? sbgm_launch_conv_lds(p, ct, st) : sbgm_launch_conv_wino(p, ct, st);
}
// GroupNorm statistics the launch above leaves in p.gn_stats (chunks per sample), 0 = none
int tile_gn_chunks(const ConvParams& p, const ConvTile& ct) {
    return ct.wino == 2 ? sbgm_conv_w2d_gn_chunks(p, ct) : sbgm_conv_lds_gn_chunks(p, ct);
}

struct ConvOpKey {
    int kh, kw, s, p, B, H, W, Cs, Cout, proj, in_mode;
    bool operator<(const ConvOpKey& o) const { return std::memcmp(this, &o, sizeof(*this)) < 0; }
};

}  // namespace

struct sbgm_model {
    sbgm_model_config cfg;
    int cin_total = 0, cs_in = 0, D = 0;
    std::vector<std::unique_ptr<Param>> params;
    std::map<std::string, Param*> by_name;
    float* arena = nullptr;
    size_t arena_floats = 0;
    // structure
    ConvW conv1, conv2;
    BNW bn1;
    std::vector<BlockW> layers[4];
    Param *enc_freq = nullptr, *label_emb = nullptr;
    Param *enc_tpw[5], *enc_tpb[5];
    AttnW enc_attn[5];
    bool enc_has_attn[5];
    DecW dec[4];
    ConvW fin_up, fin_conv;
    bool bn_dirty = true;
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0, ws_used = 0;
    // sampler state
    SamplerState* d_state = nullptr;
    StepScalars* d_table = nullptr;
    int table_cap = 0;
    // pinned host staging of a run's step table + initial state: the upload is a true asynchronous copy, so sbgm_sampler_run does not
    // have to wait for it (or for anything enqueued before it); ev_stage guards the buffer against the next call's rewrite
    char* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    hipEvent_t ev_stage = nullptr;
    bool stage_pending = false;
    std::map<ConvOpKey, ConvTile> tuned;
    ConvTile last_tile{};                   // tile of the most recent conv() (tells the caller whether GroupNorm statistics were fused)
    bool tuning = false;
    struct ConvRec { ConvGeom g; int B, H, W, Cs, Cout, M, nsteps; ConvTile t; double flops; hipEvent_t e0, e1; float ms; int c_real; int in_mode; int proj; };
    std::vector<ConvRec>* prof = nullptr;   // when set, conv() brackets every launch with events
    static constexpr int PROF_REPS = 4;
    hipStream_t graph_stream = nullptr;     // private capture stream (the caller's may be the legacy default stream)
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // The captured SDE step is kept across sampler calls: capture + instantiation of its ~75 kernel nodes cost ~3 ms, as much as
    // two steps.  Everything a step bakes in is in the key (shapes, sampler kind, caller tensors, scalar arguments, workspace,
    // tile-table generation); what changes between runs lives in device memory (step table, step counter, RNG offset AND seed).
    struct StepGraphKey {
        int B, H, W, kind, guided, bn_train, domain_w;
        const void *y, *cond, *lsm, *topo, *origins, *ws, *table;
        size_t ws_bytes;
        float cfg, cfg_corr, snr_nn;
        unsigned long long plan_gen;
    };
    StepGraphKey step_key{};
    hipGraph_t step_graph = nullptr;
    hipGraphExec_t step_exec = nullptr;
    unsigned long long plan_gen = 0;         // bumped whenever the tile table changes (the captured launches embed tile choices)
    void drop_step_graph() {
        if (step_exec && graph_stream) (void)hipStreamSynchronize(graph_stream);      // a replay of it may still be executing
        if (step_exec) (void)hipGraphExecDestroy(step_exec);
        if (step_graph) (void)hipGraphDestroy(step_graph);
        step_exec = nullptr;
        step_graph = nullptr;
    }

    ~sbgm_model() {
        drop_step_graph();
        if (arena) (void)hipFree(arena);
        if (ws) (void)hipFree(ws);
        if (d_state) (void)hipFree(d_state);
        if (d_table) (void)hipFree(d_table);
        if (h_stage) (void)hipHostFree(h_stage);
        if (ev_stage) (void)hipEventDestroy(ev_stage);
        if (graph_stream) (void)hipStreamDestroy(graph_stream);
        if (ev_in) (void)hipEventDestroy(ev_in);
        if (ev_out) (void)hipEventDestroy(ev_out);
    }

    Param* add(const std::string& name, ParamKind kind, int64_t numel) {
        params.emplace_back(new Param());
        Param* p = params.back().get();
        p->name = name; p->kind = kind; p->numel = numel;
        by_name[name] = p;
        return p;
    }
    Param* vec(const std::string& n, int64_t numel) { return add(n, P_VEC, numel); }
    Param* convw(const std::string& n, int cout, int cin, int kh, int kw, int cs = 0, bool wino = false) {
        Param* p = add(n, P_CONV, (int64_t)cout * cin * kh * kw);
        p->cout = cout; p->cin = cin; p->kh = kh; p->kw = kw;
        p->cs = cs ? cs : (int)align_up(cin, 16);
        p->wino = wino && kh == 3 && kw == 3 && p->cs % 16 == 0 && getenv("SBGM_NO_WINOGRAD") == nullptr;
        return p;
    }
    BNW bn(const std::string& pre, int c) {
        BNW b;
        b.g = vec(pre + ".weight", c); b.b = vec(pre + ".bias", c);
        b.rm = vec(pre + ".running_mean", c); b.rv = vec(pre + ".running_var", c);
        add(pre + ".num_batches_tracked", P_IGNORE, 1);
        b.scale = b.bias = nullptr;
        return b;
    }
    AttnW attn(const std::string& pre, int C) {
        AttnW a; a.C = C;
        a.inw = convw(pre + ".mha.in_proj_weight", 3 * C, C, 1, 1);
        a.inb = vec(pre + ".mha.in_proj_bias", 3 * C);
        a.outw = convw(pre + ".mha.out_proj.weight", C, C, 1, 1);
        a.outb = vec(pre + ".mha.out_proj.bias", C);
        a.ln1g = vec(pre + ".ln1.weight", C); a.ln1b = vec(pre + ".ln1.bias", C);
        a.ln2g = vec(pre + ".ln2.weight", C); a.ln2b = vec(pre + ".ln2.bias", C);
        a.f1w = convw(pre + ".ff.0.weight", C, C, 1, 1); a.f1b = vec(pre + ".ff.0.bias", C);
        a.f2w = convw(pre + ".ff.2.weight", C, C, 1, 1); a.f2b = vec(pre + ".ff.2.bias", C);
        return a;
    }

    int build(const sbgm_model_config& c);
    int ensure_ws(size_t bytes);
    int dummy_forward(int B, int H, int W, bool tune, hipStream_t st);
    int prepare_ws(int B, int H, int W, int bn_train, size_t factor, hipStream_t st);
    float* wsalloc(size_t floats) {     // bump allocator over the activation workspace; nullptr (+ error text) when full
        const size_t bytes = align_up(floats * 4, 256);
        if (ws_used + bytes > ws_bytes) {
            sbgm_set_error("workspace exhausted: need %zu more bytes at offset %zu of %zu", bytes, ws_used, ws_bytes);
            return nullptr;
        }
        float* p = reinterpret_cast<float*>(ws + ws_used);
        ws_used += bytes;
        return p;
    }
    float* partial = nullptr;           // split-K scratch shared by every convolution of a forward (stream-ordered reuse)
    static constexpr size_t PARTIAL_FLOATS = 16u << 20;   // 64 MiB
    // Workspace sizing.  The first evaluation of a (B, H, W, BatchNorm mode) runs on a generous bound (1 Ki floats per input pixel);
    // its bump-allocator high-water mark is recorded and every later call asks for exactly that (+ 8 MiB), and ensure_ws gives the
    // surplus back once every shape seen so far is measured (C2: 2.2 GB -> ~0.45 GB).  Addresses are assigned in the same order
    // either way, so results are bit-identical.
    std::map<std::array<int, 4>, size_t> ws_peak;
    size_t ws_target = 0;                   // largest measured total need (forward + sampler slabs) of any shape seen so far
    size_t fwd_need(int B, int H, int W, int bn_train = 0) const;
    size_t sampler_keep(int B, int H, int W) const {
        const size_t n = (size_t)B * H * W;
        return align_up(n * 4, 256) * 3 + align_up((size_t)B * 4, 256) + align_up((size_t)B * 8, 256);
    }
    size_t ws_need(int B, int H, int W, int bn_train = 0) const { return fwd_need(B, H, W, bn_train) + sampler_keep(B, H, W); }
    int fold_bn(hipStream_t st);
    ConvTile pick_tile(const ConvGeom& g, const ConvParams& p);
    int conv(const ConvGeom& g, ConvParams p, hipStream_t st);
    int launch_any(const ConvGeom& g, ConvParams p, const ConvTile& ct, hipStream_t st) { return launch_tile(g, p, ct, partial, st); }
    int attention(const AttnW& a, float* x, int B, int S, hipStream_t st);
    int forward_impl(const float* x, const float* t, const int64_t* y, const float* cond, const float* lsm, const float* topo,
                     float* out, float* const* fmaps_out, int B, int H, int W, int bn_train, hipStream_t st);
    int forward(const float* x, const float* t, const int64_t* y, const float* cond, const float* lsm, const float* topo,
                float* out, float* const* fmaps_out, int B, int H, int W, int bn_train, hipStream_t st) {
        const int rc = forward_impl(x, t, y, cond, lsm, topo, out, fmaps_out, B, H, W, bn_train, st);
        if (!rc && !tuning) {                                // the evaluation's high-water mark sizes every later call of this shape
            size_t& pk = ws_peak[std::array<int, 4>{B, H, W, bn_train != 0}];
            pk = std::max(pk, ws_used);
            ws_target = std::max(ws_target, pk + ((size_t)8 << 20) + sampler_keep(B, H, W));
        }
        return rc;
    }
    int sampler(const sbgm_sampler_args& a, hipStream_t st);
};

int sbgm_model::build(const sbgm_model_config& c) {
    cfg = c;
    SBGM_CHECK(c.time_embedding > 0 && c.time_embedding % 2 == 0, "time_embedding=%d must be even", c.time_embedding);
    SBGM_CHECK(c.last_fmap_channels == 512, "last_fmap_channels=%d: the encoder always emits 512 (score_unet.py:198)",
               c.last_fmap_channels);
    SBGM_CHECK(c.n_heads > 0, "n_heads must be positive");
    D = c.time_embedding;
    cin_total = 1 + c.n_lsm_channels + c.n_topo_channels + c.n_cond_channels;
    cs_in = pad_channels(cin_total);
    SBGM_CHECK(cs_in <= 16, "input channels %d > 16 unsupported", cin_total);
    // ---- encoder (registration mirrors the reference state_dict names) ------------------------------------------
    // 2 input channels (x + one condition, BASELINE config 2): pixels keep 4 slots but the weights are packed 8 taps x 2 channels
    // per K step, so no MFMA work is spent on the two padding slots
    conv1.w = convw("encoder.conv1.weight", 64, cin_total, 8, 8, cin_total == 2 ? 2 : cs_in);
    bn1 = bn("encoder.bn1", 64);
    int cin = 64;
    for (int li = 0; li < 4; ++li) {
        const int w = FMAP_CH[li + 1];
        SBGM_CHECK(c.block_layers[li] >= 1, "block_layers[%d] must be >= 1", li);
        for (int bi = 0; bi < c.block_layers[li]; ++bi) {
            const std::string pre = "encoder.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            BlockW b;
            b.cin = cin; b.cout = w; b.stride = (bi == 0 && li > 0) ? 2 : 1;
            b.c1.w = convw(pre + ".conv1.weight", w, cin, 3, 3, 0, b.stride == 1);
            b.bn1 = bn(pre + ".bn1", w);
            b.c2.w = convw(pre + ".conv2.weight", w, w, 3, 3, 0, true);
            b.bn2 = bn(pre + ".bn2", w);
            b.has_ds = (bi == 0) && (b.stride != 1 || cin != w);
            if (b.has_ds) {
                b.ds.w = convw(pre + ".downsample.0.weight", w, cin, 1, 1);
                b.dsbn = bn(pre + ".downsample.1", w);
            }
            layers[li].push_back(b);
            cin = w;
        }
    }
    enc_freq = vec("encoder.sinusoidal_embedding.W", D / 2);
    for (int i = 0; i < 5; ++i) {
        enc_tpw[i] = vec("encoder.time_projection_layers." + std::to_string(i) + ".1.weight", (int64_t)FMAP_CH[i] * D);
        enc_tpb[i] = vec("encoder.time_projection_layers." + std::to_string(i) + ".1.bias", FMAP_CH[i]);
    }
    for (int i = 0; i < 5; ++i) {
        enc_has_attn[i] = i >= 3;                                                   // score_unet.py:396
        if (enc_has_attn[i]) {
            SBGM_CHECK(FMAP_CH[i] % c.n_heads == 0, "channels %d not divisible by heads %d", FMAP_CH[i], c.n_heads);
            enc_attn[i] = attn("encoder.attention_layers." + std::to_string(i), FMAP_CH[i]);
        }
    }
    conv2.w = convw("encoder.conv2.weight", 64, 64, 8, 8);
    if (c.num_classes > 0) label_emb = vec("encoder.label_emb.weight", (int64_t)(c.num_classes + 1) * D);
    // ---- decoder ---------------------------------------------------------------------------------------------------
    int dc = c.last_fmap_channels;
    auto dec_block = [&](const std::string& pre, DecW& d, int ci, int co, bool with_norm, bool with_attn) {
        d.cin = ci; d.cout = co; d.has_attn = with_attn;
        if (c.decoder_transpose) {       // ablation path: ConvTranspose2d(ci, ci, 2, 2) as a 1x1 conv to 4*ci phase-major channels
            d.up.w = add(pre + ".transpose.weight", P_TCONV, (int64_t)ci * ci * 4);
            d.up.w->cout = 4 * ci; d.up.w->cin = ci; d.up.w->kh = d.up.w->kw = 1; d.up.w->cs = (int)align_up(ci, 16);
            d.up.b = add(pre + ".transpose.bias", P_VEC4, ci);
        } else {
            d.up.w = convw(pre + ".conv_up.weight", ci, ci, 3, 3, 0, true);
            d.up.b = vec(pre + ".conv_up.bias", ci);
        }
        const bool affine = with_norm && c.decoder_norm == SBGM_NORM_GROUP;
        d.n1g = affine ? vec(pre + ".norm1.weight", ci) : nullptr;
        d.n1b = affine ? vec(pre + ".norm1.bias", ci) : nullptr;
        if (co == 1) {
            d.conv.w = add(pre + ".conv.weight", P_COUT1, (int64_t)ci * 9);
            d.conv.w->cin = ci;
        } else {
            d.conv.w = convw(pre + ".conv.weight", co, ci, 3, 3, 0, true);
        }
        d.conv.b = vec(pre + ".conv.bias", co);
        d.n2g = affine ? vec(pre + ".norm2.weight", co) : nullptr;
        d.n2b = affine ? vec(pre + ".norm2.bias", co) : nullptr;
        d.freq = vec(pre + ".sinusoidal_embedding.W", D / 2);
        d.tpw = vec(pre + ".time_projection_layer.1.weight", (int64_t)co * D);
        d.tpb = vec(pre + ".time_projection_layer.1.bias", co);
        if (with_attn) {
            SBGM_CHECK(co % c.n_heads == 0, "channels %d not divisible by heads %d", co, c.n_heads);
            d.attn = attn(pre + ".attention", co);
        }
        return 0;
    };
    for (int i = 0; i < 4; ++i) {
        const int co = i != 3 ? dc / 2 : 64;
        if (dec_block("decoder.residual_layers." + std::to_string(i), dec[i], dc, co, true, i < 2)) return 1;
        dc = co;
    }
    DecW fin;
    if (dec_block("decoder.final_layer", fin, dec[3].cin, 1, false, false)) return 1;
    // final layer: its time-embedding tensors exist in the state_dict but are never used (score_unet.py:757)
    fin.freq->kind = fin.tpw->kind = fin.tpb->kind = P_IGNORE;
    fin_up = fin.up; fin_conv = fin.conv;

    // ---- storage ----------------------------------------------------------------------------------------------------
    size_t total = 0;
    for (auto& up : params) {
        Param* p = up.get();
        if (p->kind == P_CONV || p->kind == P_TCONV) p->dev_floats = (size_t)sbgm_conv_nsteps(p->kh, p->kw, p->cs) * p->cout * 16;
        else if (p->kind == P_VEC4) p->dev_floats = (size_t)p->numel * 4;
        else if (p->kind == P_IGNORE) p->dev_floats = 0;
        else p->dev_floats = (size_t)p->numel;
        if (p->wino) p->wino_floats = sbgm_wino_packed_floats(p->cout, p->cs);
        if (p->wino && getenv("SBGM_NO_WINOGRAD2D") == nullptr) p->w2d_floats = sbgm_w2d_packed_floats(p->cout, p->cs);
        total += align_up(p->dev_floats, 64) + align_up(p->wino_floats, 64) + align_up(p->w2d_floats, 64);
    }
    // folded BN scale/bias
    size_t bn_floats = 0;
    auto count_bn = [&](BNW& b, int c_) { bn_floats += 2 * align_up((size_t)c_, 64); (void)b; };
    count_bn(bn1, 64);
    for (int li = 0; li < 4; ++li)
        for (auto& b : layers[li]) { count_bn(b.bn1, b.cout); count_bn(b.bn2, b.cout); if (b.has_ds) count_bn(b.dsbn, b.cout); }
    arena_floats = total + bn_floats + 64;
    SBGM_HIP(hipMalloc(&arena, arena_floats * 4));
    SBGM_HIP(hipMemset(arena, 0, arena_floats * 4));
    size_t off = 0;
    for (auto& up : params) {
        Param* p = up.get();
        if (p->dev_floats) { p->dev = arena + off; off += align_up(p->dev_floats, 64); }
        if (p->wino_floats) { p->dev_wino = arena + off; off += align_up(p->wino_floats, 64); }
        if (p->w2d_floats) { p->dev_w2d = arena + off; off += align_up(p->w2d_floats, 64); }
        if (p->kind == P_IGNORE) p->filled = true;
    }
    auto place_bn = [&](BNW& b, int c_) {
        b.scale = arena + off; off += align_up((size_t)c_, 64);
        b.bias = arena + off; off += align_up((size_t)c_, 64);
    };
    place_bn(bn1, 64);
    for (int li = 0; li < 4; ++li)
        for (auto& b : layers[li]) { place_bn(b.bn1, b.cout); place_bn(b.bn2, b.cout); if (b.has_ds) place_bn(b.dsbn, b.cout); }
    SBGM_HIP(hipMalloc(&d_state, sizeof(SamplerState)));
    return 0;
}

int sbgm_model::ensure_ws(size_t bytes) {
    // enough, and not grossly oversized now that every shape in use has a measured need: keep it
    const bool trim = ws != nullptr && ws_target > 0 && bytes <= ws_bytes &&
                      ws_bytes > std::max(bytes, ws_target) + ((size_t)256 << 20);
    if (bytes <= ws_bytes && !trim) return 0;
    if (trim) bytes = std::max(bytes, ws_target);
    drop_step_graph();                       // its nodes point into the old workspace
    if (ws) SBGM_HIP(hipFree(ws));
    ws = nullptr; ws_bytes = 0;
    SBGM_HIP(hipMalloc(&ws, bytes));
    ws_bytes = bytes;
    return 0;
}

// measured high-water mark of this shape when there is one, else a generous upper bound
size_t sbgm_model::fwd_need(int B, int H, int W, int bn_train) const {
    auto it = ws_peak.find(std::array<int, 4>{B, H, W, bn_train != 0});
    if (it != ws_peak.end()) return it->second + ((size_t)8 << 20);
    const size_t px = (size_t)B * H * W;
    // NHWC floats per input pixel summed over all intermediates (encoder ~ 64/4*3 + ..., decoder dominated by the
    // final block's 3 x 64 channels at full resolution); 1024 floats/pixel is > 2x the true footprint.
    return px * 1024 * 4 + PARTIAL_FLOATS * 4 + (64u << 20);
}

int sbgm_model::fold_bn(hipStream_t st) {
    auto f = [&](BNW& b, int c) {
        return sbgm_launch_bn_fold(b.g->dev, b.b->dev, b.rm->dev, b.rv->dev, BN_EPS, b.scale, b.bias, c, st);
    };
    if (f(bn1, 64)) return 1;
    for (int li = 0; li < 4; ++li)
        for (auto& b : layers[li]) {
            if (f(b.bn1, b.cout) || f(b.bn2, b.cout)) return 1;
            if (b.has_ds && f(b.dsbn, b.cout)) return 1;
        }
    bn_dirty = false;
    return 0;
}

// Static choice for a convolution the autotuner has not timed.  It follows what the tuner picks on the BASELINE shapes and on
// small batches (profiles/r03_c2_tiles.txt, r03_c4_tiles.txt; B = 1, 2, 8 tables in DESIGN.md 3.2), so a sampler that never called
// sbgm_model_autotune runs within a few per cent of a tuned one instead of on the round-1 kernels:
//   3x3 stride 1, >= 512 tiles of 16x16 pixels x 16 channels (or the final projection): 2-D Winograd F(2x2,3x3) (conv_w2d.hip) — the
//     persistent 32-channel kernel once there are >= 512 such tiles (two per CU), 16-channel double-buffered workgroups below;
//   fewer tiles, or a fused input mode: the LDS-staged 1-D Winograd kernel on 16-channel slices (conv_lds.hip);
//   3x3 stride 1 with >= 2048 pixels of >= 128 channels (the 8x8 / 4x4 maps of a full batch): 1-D Winograd (conv_wino.hip), the largest
//     tile that still gives >= 256 workgroups, the K loop split over 4 or 8 waves;
//   small problems of any geometry (< 1024 tiles of 32 channels x 16 pixels): that smallest wave tile, K split over the 4 waves of
//     a workgroup and over up to 8 workgroups;
//   everything else (strided, 1x1, the stem): wave tiles that fill ~2 waves per SIMD.
ConvTile sbgm_model::pick_tile(const ConvGeom& g, const ConvParams& p) {
    const int OH = (p.H + 2 * g.pad - g.kh) / g.stride + 1, OW = (p.W + 2 * g.pad - g.kw) / g.stride + 1;
    ConvOpKey key{g.kh, g.kw, g.stride, g.pad, p.B, p.H, p.W, p.Cs, p.Cout, p.proj_w != nullptr, p.in_mode};
    auto it = tuned.find(key);
    if (it != tuned.end()) return it->second;
    static const bool round1 = getenv("SBGM_STATIC_ROUND1") != nullptr;      // the round-1 table (A/B of this function)
    static const bool lds_ok = getenv("SBGM_NO_LDS_CONV") == nullptr && !round1;
    const bool s1 = g.kh == 3 && g.kw == 3 && g.stride == 1 && g.pad == 1 && p.in_dil <= 1;
    const int M = p.B * OH * OW;
    const int nsteps = sbgm_conv_nsteps(g.kh, g.kw, p.c_real == 2 ? 2 : p.Cs);
    if (lds_ok && s1 && p.W % 16 == 0 && p.H % 2 == 0 && p.Cs % 16 == 0) {
        const long tiles16 = (long)p.B * (p.W / 16) * ((p.H + 15) / 16) * (p.Cout / 16);
        if (p.wp_w2d != nullptr && (tiles16 >= 512 || p.proj_w)) {
            const ConvTile big{2, 1, 1, 2, 2, 3}, mid{1, 1, 1, 2, 2, 3};
            if (p.Cout % 32 == 0 && tiles16 >= 1024 && sbgm_conv_w2d_bytes(big, p.in_mode) <= 160 * 1024) return big;
            if (p.proj_w && sbgm_conv_w2d_bytes(mid, p.in_mode) <= 160 * 1024) return mid;
            for (int lds : {2, 1}) {
                const ConvTile small{1, 1, 1, 1, 2, lds};
                if (!p.proj_w && sbgm_conv_w2d_bytes(small, p.in_mode) <= 160 * 1024) return small;
            }
        }
        if (p.wp_wino != nullptr && !p.proj_w && (p.in_mode != 0 || tiles16 >= 256 || (tiles16 >= 128 && p.W >= 32)))
            for (int lds : {2, 1}) {
                const ConvTile t{1, 1, 1, 1, 1, lds};
                if (sbgm_conv_lds_bytes(t, p.in_mode) <= 160 * 1024) return t;
            }
    }
    if (p.in_mode != 0) return ConvTile{p.Cout % 64 == 0 ? 4 : 2, 1, 1, 1, 1, 1};     // fused input modes: LDS-staged Winograd tiles only
    const bool wino_ok = p.wp_wino != nullptr && s1 && p.W % 2 == 0;
    if (wino_ok && p.proj_w) return ConvTile{p.Cout / 16, 1, 1, 1, 1, 0};
    if (wino_ok && !round1 && ((M >= 2048 && p.Cs >= 128) || (M >= 512 && p.Cs >= 512 && p.Cout >= 512))) {
        const int ns = 3 * (p.Cs / 16);
        const int wt[3][2] = {{4, 2}, {4, 1}, {2, 1}};
        for (auto& t : wt) {
            if (p.Cout % (16 * t[0])) continue;
            const long wgs = (long)((M + 32 * t[1] - 1) / (32 * t[1])) * (p.Cout / (16 * t[0]));
            if (wgs >= 256 || (t[0] == 2 && t[1] == 1)) {
                int ws = t[1] == 2 ? 4 : ((p.Cs >= 512 || wgs < 512) ? 8 : 4);
                while (ws > 1 && ns / ws < 2) ws >>= 1;
                return ConvTile{t[0], t[1], 1, ws, 1, 0};
            }
        }
    }
    if (wino_ok && round1) {                          // Winograd F(2,3): 1.5x fewer MFMAs; pick waves-per-tile to fill the chip
        const int Mp = p.B * OH * OW / 2, ns = 3 * (p.Cs / 16);
        const long tiles = (long)((Mp + 31) / 32) * (p.Cout / 32);          // (2,2) tiles: 32 channels x 64 pixels
        const int ws = tiles >= 2048 ? 1 : (tiles >= 1024 || ns < 8) ? 2 : 4;
        return ConvTile{2, 2, 1, ws, 1, 0};
    }
    if (p.proj_w) return ConvTile{p.Cout / 16, 2, 1, 1, 0, 0};
    const long t21 = (long)((M + 15) / 16) * (p.Cout / 32);
    if (!round1 && p.Cout % 32 == 0 && t21 < 1024) {
        const int ws = nsteps >= 8 ? 4 : nsteps >= 4 ? 2 : 1;
        int splits = 1;
        while (splits < 8 && t21 * splits * 2 <= 256 && nsteps / (splits * 2 * ws) >= 4) splits *= 2;
        return ConvTile{2, 1, splits, ws, 0, 0};
    }
    const int target = 2048;                 // ~2 waves per SIMD
    const int cand[3][2] = {{4, 4}, {4, 2}, {2, 2}};
    for (auto& c : cand) {
        if (p.Cout % (16 * c[0])) continue;
        const long tiles = (long)((M + 16 * c[1] - 1) / (16 * c[1])) * (p.Cout / (16 * c[0]));
        for (int ws : {1, 2, 4})
            if (tiles * ws >= target && nsteps / ws >= 2) return ConvTile{c[0], c[1], 1, ws, 0, 0};
    }
    // tiny problem: 64x32 (or 32x32) tiles, 4 waves per tile, plus split-K over the grid (>= 2 K-steps per wave)
    const int fco = p.Cout % 64 == 0 ? 4 : 2, fpx = 2;
    const long tiles = (long)((M + 16 * fpx - 1) / (16 * fpx)) * (p.Cout / (16 * fco));
    const int ws = nsteps >= 8 ? 4 : nsteps >= 4 ? 2 : 1;
    const int splits = (int)std::min<long>(std::max<long>(1, target / std::max<long>(1, tiles * ws)), std::max(1, nsteps / (2 * ws)));
    return ConvTile{fco, fpx, splits, ws, 0, 0};
}

// Times the tile candidates (template x tile x waves-per-tile x split-K) of ONE convolution on its real operands and returns
// the fastest in *best (in: the fallback).  A launch never reads what it writes, so repeating it is harmless.  Synchronises.
int sbgm_tune_conv(const ConvGeom& g, const ConvParams& p, float* partial, size_t partial_floats, hipStream_t st, ConvTile* best) {
    const int OH = p.out_h > 0 ? p.out_h : (p.H + 2 * g.pad - g.kh) / g.stride + 1;
    const int OW = p.out_w > 0 ? p.out_w : (p.W + 2 * g.pad - g.kw) / g.stride + 1;
    const size_t mc = (size_t)p.B * OH * OW * p.Cout;
    const int nsteps = sbgm_conv_nsteps(g.kh, g.kw, p.c_real == 2 ? 2 : p.Cs);
    auto launch = [&](const ConvTile& ct) -> int { return launch_tile(g, p, ct, partial, st); };
    std::vector<ConvTile> cands;
    const int tiles[6][2] = {{4, 4}, {4, 2}, {4, 1}, {2, 4}, {2, 2}, {2, 1}};
    for (auto& t : tiles) {
        if (p.in_mode != 0) break;                    // the fused input modes exist in the LDS-staged Winograd kernel only
        if (p.Cout % (16 * t[0])) continue;
        if (p.proj_w && 16 * t[0] != p.Cout) continue;
        const long ntile = (long)(((size_t)p.B * OH * OW + 16 * t[1] - 1) / (16 * t[1])) * (p.Cout / (16 * t[0]));
        for (int ws : {1, 2, 4}) {
            if (ws > 1 && (nsteps / ws < 2 || ntile * ws > 32768)) continue;
            for (int sp : {1, 2, 4, 8, 16}) {
                if (sp > 1 && (p.proj_w || !partial || nsteps / (sp * ws) < 2 || ntile * ws >= 4096)) continue;   // already enough waves
                cands.push_back(ConvTile{t[0], t[1], sp, ws, 0, 0});
            }
        }
    }
    const bool s1 = g.kh == 3 && g.kw == 3 && g.stride == 1 && g.pad == 1 && p.in_dil <= 1 &&
                    (p.out_h == 0 || (p.out_h == p.H && p.out_w == p.W));
    if (p.wp_wino != nullptr && s1 && p.W % 2 == 0 && p.in_mode == 0) {
        const int wt[4][2] = {{4, 1}, {2, 2}, {2, 1}, {4, 2}};
        const int nsw = 3 * (p.Cs / 16);
        for (auto& t : wt) {
            if (p.Cout % (16 * t[0])) continue;
            if (p.proj_w && 16 * t[0] != p.Cout) continue;
            for (int ws : {1, 2, 4, 8}) {
                if (ws > 1 && nsw / ws < 2) continue;
                cands.push_back(ConvTile{t[0], t[1], 1, ws, 1, 0});
            }
        }
    }
    if (s1 && p.W % 16 == 0 && p.Cs % 16 == 0 && getenv("SBGM_NO_LDS_CONV") == nullptr) {
        const int dt[6][2] = {{4, 1}, {4, 2}, {4, 4}, {2, 2}, {2, 4}, {2, 1}};
        auto lds_bytes = [](int fco, int rows_per_wave, bool wino) {
            return ((size_t)(wino ? 12 : 9) * 16 * fco * 4 + (size_t)(4 * rows_per_wave + 2) * (wino ? 19 : 18) * 4) * 16;
        };
        for (auto& t : dt) {
            if (p.in_mode != 0) break;
            if (p.Cout % (16 * t[0]) || (p.proj_w && 16 * t[0] != p.Cout)) continue;
            cands.push_back(ConvTile{t[0], t[1], 1, 1, 0, 1});
            if (2 * lds_bytes(t[0], t[1], false) <= 160 * 1024) cands.push_back(ConvTile{t[0], t[1], 1, 1, 0, 2});   // double-buffered
        }
        const int wt2[6][2] = {{4, 1}, {4, 2}, {2, 1}, {2, 2}, {1, 1}, {1, 2}};   // 16-channel slices double the workgroup count of small layers
        if (p.wp_wino)
            for (auto& t : wt2) {
                if (p.Cout % (16 * t[0]) || (p.proj_w && 16 * t[0] != p.Cout)) continue;
                if (sbgm_conv_lds_bytes(ConvTile{t[0], t[1], 1, 1, 1, 1}, p.in_mode) <= 160 * 1024) cands.push_back(ConvTile{t[0], t[1], 1, 1, 1, 1});
                if (sbgm_conv_lds_bytes(ConvTile{t[0], t[1], 1, 1, 1, 2}, p.in_mode) <= 160 * 1024) cands.push_back(ConvTile{t[0], t[1], 1, 1, 1, 2});
            }
    }
    // 2-D Winograd F(2x2,3x3), LDS-staged: 16x16-pixel tiles, 16 or 32 channels per workgroup; a tap projection may span several
    // channel tiles (partial planes).  ws = 2 selects the build that is held to two waves per SIMD.
    if (p.wp_w2d != nullptr && s1 && p.W % 16 == 0 && p.H % 2 == 0 && p.Cs % 16 == 0 && getenv("SBGM_NO_LDS_CONV") == nullptr)
        for (int fco : {2, 1}) {
            if (p.Cout % (16 * fco)) continue;
            for (int lds : {1, 2})
                for (int ws : {1, 2}) {
                    const ConvTile ct{fco, 1, 1, ws, 2, lds};
                    if (fco == 1 && ws == 2) continue;
                    if (sbgm_conv_w2d_bytes(ct, p.in_mode) <= 160 * 1024) cands.push_back(ct);
                }
            cands.push_back(ConvTile{fco, 1, 1, 2, 2, 3});       // persistent workgroups, LDS-DMA slab (two per CU)
        }
    hipEvent_t e0, e1;
    SBGM_HIP(hipEventCreate(&e0));
    SBGM_HIP(hipEventCreate(&e1));
    const bool cold = getenv("SBGM_TUNE_WARM") == nullptr;
    float best_ms = 1e30f;
    int rc = 0;
    for (int round = 0; round < 3 && !rc; ++round)          // three interleaved rounds, keep each candidate's best (DVFS / noise)
        for (auto& ct : cands) {
            if (ct.splits > 1 && mc * ct.splits > partial_floats) continue;
            constexpr int REPS = 6;
            float ms = 0.f;
            if (cold && partial) {
                // In the network a convolution finds its weights cold (the layers in between have streamed hundreds of MB
                // through L2 / Infinity Cache), so every timed launch is preceded by an untimed 48 MiB fill that evicts them:
                // ranking the candidates warm (back-to-back repeats) picked tiles that were 3 % slower per sampling step.
                for (int rep = 0; rep < 3 && !rc; ++rep) {
                    (void)hipMemsetAsync(partial, 0, std::min<size_t>(partial_floats * 4, (size_t)48 << 20), st);
                    (void)hipEventRecord(e0, st);
                    rc = launch(ct);
                    (void)hipEventRecord(e1, st);
                    (void)hipEventSynchronize(e1);
                    float m1 = 0.f;
                    (void)hipEventElapsedTime(&m1, e0, e1);
                    ms += m1;
                }
                if (rc) break;
            } else {
                for (int rep = 0; rep <= REPS && !rc; ++rep) {
                    if (rep == 1) (void)hipEventRecord(e0, st);
                    rc = launch(ct);
                }
                if (rc) break;
                (void)hipEventRecord(e1, st);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            if (ms < best_ms) { best_ms = ms; *best = ct; }
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int sbgm_model::conv(const ConvGeom& g, ConvParams p, hipStream_t st) {
    const int OH = (p.H + 2 * g.pad - g.kh) / g.stride + 1, OW = (p.W + 2 * g.pad - g.kw) / g.stride + 1;
    const size_t mc = (size_t)p.B * OH * OW * p.Cout;
    if (tuning) {
        // time every candidate on this op, keep the fastest
        ConvOpKey key{g.kh, g.kw, g.stride, g.pad, p.B, p.H, p.W, p.Cs, p.Cout, p.proj_w != nullptr, p.in_mode};
        if (tuned.find(key) == tuned.end()) {
            ConvTile best_t = pick_tile(g, p);
            if (sbgm_tune_conv(g, p, partial, PARTIAL_FLOATS, st, &best_t)) return 1;
            tuned[key] = best_t;
            ++plan_gen;
        }
    }
    ConvTile ct = pick_tile(g, p);
    if (ct.splits > 1 && mc * ct.splits > PARTIAL_FLOATS) ct.splits = (int)std::max<size_t>(1, PARTIAL_FLOATS / mc);
    last_tile = ct;
    if (!prof) return launch_any(g, p, ct, st);
    ConvRec r{g, p.B, p.H, p.W, p.Cs, p.Cout, p.B * OH * OW, sbgm_conv_nsteps(g.kh, g.kw, p.c_real == 2 ? 2 : p.Cs), ct, 0.0, nullptr, nullptr, 0.f, p.c_real, p.in_mode, p.proj_w != nullptr};
    // algorithmic FLOPs: 2 * M * Cout * (KH*KW*Cin_real); Cs may be padded (only the stem conv), count real K there
    const int cin_real = (g.kh == 8 && p.Cs <= 16) ? cin_total : p.Cs;
    r.flops = 2.0 * r.M * p.Cout * (double)(g.kh * g.kw * cin_real);
    SBGM_HIP(hipEventCreate(&r.e0));
    SBGM_HIP(hipEventCreate(&r.e1));
    // PROF_REPS back-to-back launches per event pair (a launch is idempotent: it never reads what it writes), so the interval is
    // dominated by execution time rather than by the event packets and the host's launch gaps
    SBGM_HIP(hipEventRecord(r.e0, st));
    int rc = 0;
    for (int rep = 0; rep < PROF_REPS && !rc; ++rep) rc = launch_any(g, p, ct, st);
    SBGM_HIP(hipEventRecord(r.e1, st));
    prof->push_back(r);
    return rc;
}

// y = h + FF(LN2(h)),  h = x + MHA(LN1(x))   over tokens [B*S, C]  (score_unet.py:136-148); in place on x
int sbgm_model::attention(const AttnW& a, float* x, int B, int S, hipStream_t st) {
    const int C = a.C, M = B * S;
    static const bool no_fused = getenv("SBGM_NO_FUSED_ATTENTION") != nullptr;
    // Token-tile kernels when there are enough 16-token tiles to give every CU one (3 launches: LN1 + in_proj | core | out_proj +
    // residual + LN2 + FF + residual).  Deep levels (few tokens, 256-512 channels) are bound by streaming 1-6 MB of weights: there
    // the separate GEMMs, which split the OUTPUT CHANNELS over the chip, stay faster (measured: 512 tokens x 512 channels 90 us
    // per fused kernel on 32 workgroups vs ~8 us per GEMM).
    static const int fused_min_m = getenv("SBGM_ATTN_FUSED_MIN_M") ? atoi(getenv("SBGM_ATTN_FUSED_MIN_M")) : 256 * 16;
    if (!no_fused && sbgm_attn_tokens_supported(C) && M >= fused_min_m) {
        float* qkv = wsalloc((size_t)M * 3 * C);
        if (!qkv) return 1;
        float* att = wsalloc((size_t)M * C);
        if (!att) return 1;
        if (sbgm_launch_attn_in(x, a.ln1g->dev, a.ln1b->dev, a.inw->dev, a.inb->dev, qkv, M, C, LN_EPS, st)) return 1;
        if (sbgm_launch_mha_core(qkv, att, B, S, C, cfg.n_heads, st)) return 1;
        return sbgm_launch_attn_out(att, x, a.outw->dev, a.outb->dev, a.ln2g->dev, a.ln2b->dev, a.f1w->dev, a.f1b->dev, a.f2w->dev,
                                    a.f2b->dev, x, M, C, LN_EPS, st);
    }
    float* n1 = wsalloc((size_t)M * C);
    if (!n1) return 1;
    float* qkv = wsalloc((size_t)M * 3 * C);
    if (!qkv) return 1;
    float* att = wsalloc((size_t)M * C);
    if (!att) return 1;
    float* h = wsalloc((size_t)M * C);
    if (!h) return 1;
    float* f1 = wsalloc((size_t)M * C);
    if (!f1) return 1;
    const ConvGeom lin{1, 1, 1, 0};
    if (sbgm_launch_layernorm(x, n1, a.ln1g->dev, a.ln1b->dev, M, C, LN_EPS, st)) return 1;
    ConvParams p{};
    p.B = 1; p.H = 1; p.W = M; p.Cs = C;
    p.x = n1; p.wp = a.inw->dev; p.out = qkv; p.bias = a.inb->dev; p.Cout = 3 * C;
    if (conv(lin, p, st)) return 1;
    if (sbgm_launch_mha_core(qkv, att, B, S, C, cfg.n_heads, st)) return 1;
    p.x = att; p.wp = a.outw->dev; p.out = h; p.bias = a.outb->dev; p.Cout = C; p.res = x;
    if (conv(lin, p, st)) return 1;
    if (sbgm_launch_layernorm(h, n1, a.ln2g->dev, a.ln2b->dev, M, C, LN_EPS, st)) return 1;
    p.x = n1; p.wp = a.f1w->dev; p.out = f1; p.bias = a.f1b->dev; p.res = nullptr; p.act = SBGM_ACT_GELU;   // :131-132
    if (conv(lin, p, st)) return 1;
    p.x = f1; p.wp = a.f2w->dev; p.out = x; p.bias = a.f2b->dev; p.res = h; p.act = SBGM_ACT_NONE;
    return conv(lin, p, st);
}

int sbgm_model::forward_impl(const float* x, const float* t, const int64_t* y, const float* cond, const float* lsm,
                             const float* topo, float* out, float* const* fmaps_out, int B, int H, int W, int bn_train,
                             hipStream_t st) {
    SBGM_CHECK(B >= 1 && H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0,
               "forward: H,W must be positive multiples of 32 (five stride-2 stages), got B=%d H=%d W=%d", B, H, W);
    SBGM_CHECK(x && t && out, "forward: x, t and out are required");
    SBGM_CHECK((cfg.n_cond_channels > 0) == (cond != nullptr), "forward: cond_img presence does not match the model (%d channels)", cfg.n_cond_channels);
    SBGM_CHECK((cfg.n_lsm_channels > 0) == (lsm != nullptr), "forward: lsm_cond presence does not match the model");
    SBGM_CHECK((cfg.n_topo_channels > 0) == (topo != nullptr), "forward: topo_cond presence does not match the model");
    SBGM_CHECK(!(y && !label_emb), "forward: y given but the model has no label embedding");
    if (sbgm_model_check_complete(this)) return 1;
    if (!tuning) {
        SBGM_CHECK(fwd_need(B, H, W, bn_train) <= ws_bytes, "forward: workspace not prepared for B=%d H=%d W=%d", B, H, W);
        ws_used = 0;
    }
    if (bn_dirty && fold_bn(st)) return 1;
    partial = wsalloc(PARTIAL_FLOATS);
    if (!partial) return 1;

    // ---- pack inputs, time embedding -------------------------------------------------------------------------------
    PackSrc src{};
    auto push = [&](const float* p, int c) { if (p && c) { src.ptr[src.n] = p; src.ch[src.n] = c; ++src.n; } };
    push(x, 1); push(lsm, cfg.n_lsm_channels); push(topo, cfg.n_topo_channels); push(cond, cfg.n_cond_channels);
    float* x0 = wsalloc((size_t)B * H * W * cs_in);
    if (!x0) return 1;
    if (sbgm_launch_pack_input(src, x0, B, H, W, cs_in, st)) return 1;

    TimeEmbedArgs te{};
    te.t = t; te.y = y; te.label_emb = (y && label_emb) ? label_emb->dev : nullptr;
    te.B = B; te.D = D;
    te.n_emb = 5;
    te.freqs[0] = enc_freq->dev;
    for (int i = 0; i < 4; ++i) te.freqs[1 + i] = dec[i].freq->dev;
    te.emb_ws = wsalloc((size_t)5 * B * D);
    if (!te.emb_ws) return 1;
    float* tb[9];
    te.n_proj = 9;
    for (int i = 0; i < 5; ++i) {
        tb[i] = wsalloc((size_t)B * FMAP_CH[i]);
        if (!tb[i]) return 1;
        te.proj[i] = TimeProj{enc_tpw[i]->dev, enc_tpb[i]->dev, tb[i], FMAP_CH[i], 0};
    }
    for (int i = 0; i < 4; ++i) {
        tb[5 + i] = wsalloc((size_t)B * dec[i].cout);
        if (!tb[5 + i]) return 1;
        te.proj[5 + i] = TimeProj{dec[i].tpw->dev, dec[i].tpb->dev, tb[5 + i], dec[i].cout, 1 + i};
    }
    if (sbgm_launch_time_embed(te, st)) return 1;

    // GroupNorm: 64 chunks x B x G x 2 doubles (G = C <= 512 for InstanceNorm); BatchNorm: 24 B x C
    const size_t n_groups = (size_t)B * (cfg.decoder_norm == SBGM_NORM_GROUP ? std::min(cfg.gn_groups, 512) : 512);
    double* stats = reinterpret_cast<double*>(wsalloc(std::max<size_t>(6 * 512, n_groups * 64 * 4)));
    if (!stats) return 1;

    // conv + BatchNorm (+res, relu, late time bias): eval folds BN into the conv epilogue, train runs it after
    auto conv_bn = [&](const ConvGeom& g, const float* in, int h, int w, int cs, const ConvW& cw, BNW& bnw, int cout,
                       const float* res, bool relu, const float* tb_after, float* o) -> int {
        ConvParams p{};
        p.x = in; p.wp = cw.w->dev; p.wp_wino = cw.w->dev_wino; p.wp_w2d = cw.w->dev_w2d; p.B = B; p.H = h; p.W = w; p.Cs = cs; p.Cout = cout;
        if (!bn_train) {
            p.out = o; p.scale = bnw.scale; p.bias = bnw.bias; p.res = res; p.act = relu ? SBGM_ACT_RELU : SBGM_ACT_NONE;
            p.tbias = tb_after; p.tbias_after_act = 1;
            return conv(g, p, st);
        }
        const int oh = (h + 2 * g.pad - g.kh) / g.stride + 1, ow = (w + 2 * g.pad - g.kw) / g.stride + 1;
        float* raw = wsalloc((size_t)B * oh * ow * cout);
        if (!raw) return 1;
        p.out = raw;
        if (conv(g, p, st)) return 1;
        return sbgm_launch_batchnorm_train(raw, o, bnw.g->dev, bnw.b->dev, bnw.rm->dev, bnw.rv->dev, res, tb_after, relu, B,
                                           oh * ow, cout, BN_EPS, BN_MOMENTUM, stats, st);
    };

    // ---- encoder ------------------------------------------------------------------------------------------------------
    float* fm[5];
    int fh[5], fw[5];
    fh[0] = H / 2; fw[0] = W / 2;
    fm[0] = wsalloc((size_t)B * fh[0] * fw[0] * 64);
    if (!fm[0]) return 1;
    {
        ConvParams p{};
        p.x = x0; p.wp = conv1.w->dev; p.out = fm[0]; p.tbias = tb[0]; p.B = B; p.H = H; p.W = W; p.Cs = cs_in; p.Cout = 64;
        p.c_real = cin_total == 2 ? 2 : 0;
        if (conv(ConvGeom{8, 8, 2, 3}, p, st)) return 1;                                        // score_unet.py:312-316
    }
    int ch = H / 4, cw_ = W / 4;
    float* cur = wsalloc((size_t)B * ch * cw_ * 64);
    if (!cur) return 1;
    if (conv_bn(ConvGeom{8, 8, 2, 3}, fm[0], fh[0], fw[0], 64, conv2, bn1, 64, nullptr, true, nullptr, cur)) return 1;   // :321-325
    int cc = 64;
    for (int li = 0; li < 4; ++li) {
        const int nb = (int)layers[li].size();
        for (int bi = 0; bi < nb; ++bi) {
            BlockW& b = layers[li][bi];
            const int oh = ch / b.stride, ow = cw_ / b.stride;
            float* y1 = wsalloc((size_t)B * oh * ow * b.cout);
            if (!y1) return 1;
            if (conv_bn(ConvGeom{3, 3, b.stride, 1}, cur, ch, cw_, cc, b.c1, b.bn1, b.cout, nullptr, true, nullptr, y1)) return 1;
            const float* idn = cur;
            if (b.has_ds) {
                float* d = wsalloc((size_t)B * oh * ow * b.cout);
                if (!d) return 1;
                if (conv_bn(ConvGeom{1, 1, b.stride, 0}, cur, ch, cw_, cc, b.ds, b.dsbn, b.cout, nullptr, false, nullptr, d)) return 1;
                idn = d;
            }
            float* y2 = wsalloc((size_t)B * oh * ow * b.cout);
            if (!y2) return 1;
            const float* tba = (bi == nb - 1) ? tb[li + 1] : nullptr;                        // fmap + t_emb (:332,341,350,359)
            if (conv_bn(ConvGeom{3, 3, 1, 1}, y1, oh, ow, b.cout, b.c2, b.bn2, b.cout, idn, true, tba, y2)) return 1;
            cur = y2; ch = oh; cw_ = ow; cc = b.cout;
        }
        if (enc_has_attn[li + 1] && attention(enc_attn[li + 1], cur, B, ch * cw_, st)) return 1;
        fm[li + 1] = cur; fh[li + 1] = ch; fw[li + 1] = cw_;
    }
    if (fmaps_out) {
        for (int i = 0; i < 5; ++i)
            if (fmaps_out[i])
                SBGM_HIP(hipMemcpyAsync(fmaps_out[i], fm[i], (size_t)B * fh[i] * fw[i] * FMAP_CH[i] * 4, hipMemcpyDeviceToDevice, st));
    }

    // ---- decoder ------------------------------------------------------------------------------------------------------
    // Fused path (default): no GroupNorm-apply or upsample pass between the convolutions of a block.  conv_up reads the
    // low-resolution map and interpolates while staging (in_mode 2, with the PREVIOUS block's pending GroupNorm + skip + time
    // bias + activation applied to the low-res pixels), `conv` reads conv_up's raw output through norm1's affine (in_mode 1);
    // the statistics come out of the producing convolution's epilogue and one tiny finalize launch turns them into the
    // per-(sample, channel) scale / shift.  A block whose output feeds attention keeps the separate apply pass, and so do maps
    // narrower than 32 pixels: the fused staging costs the (compute-bound) convolution 2-4 us more than the plain one (measured,
    // tools/bench_fused_conv.py), which only pays where the pass it replaces streams more than ~8 MB (64 / 128 channels at
    // 32x32 and up; at 16x16 x 256 channels the separate 4 us pass is cheaper).  SBGM_NO_FUSED_DECODER=1 forces the separate
    // passes everywhere.
    const int G_of = cfg.gn_groups;
    auto groups = [&](int c) { return cfg.decoder_norm == SBGM_NORM_GROUP ? std::max(1, std::min(G_of, c)) : c; };
    static const bool fused_ok = getenv("SBGM_NO_FUSED_DECODER") == nullptr && getenv("SBGM_NO_LDS_CONV") == nullptr && getenv("SBGM_NO_WINOGRAD") == nullptr;
    auto can_fuse = [&](const ConvW& cw, int c_in, int w_out) {
        return fused_ok && !cfg.decoder_transpose && w_out >= 32 && w_out % 16 == 0 && c_in % 16 == 0 && cw.w->dev_wino != nullptr;
    };
    struct Pending { const float* raw; const float* affine; const float* skip; int act; bool live; } pend{nullptr, nullptr, nullptr, SBGM_ACT_NONE, false};
    // statistics of `t` [B][hw][c] for its GroupNorm: from the convolution epilogue (chunks > 0) or a separate partial pass
    auto ensure_stats = [&](const float* t, int hw, int c, int& chunks) -> int {
        if (chunks > 0) return 0;
        return sbgm_launch_gn_partial(t, stats, B, hw, c, groups(c), &chunks, st);
    };
    // conv_up of a block: input `in` [B][ch][cw_][ci] (or the pending raw map), output raw [B][2ch][2cw_][ci] (+ bias)
    auto conv_up = [&](const ConvW& cw, const float* in, int ci, int oh, int ow, ConvParams& p, float* out_raw) -> int {
        p = ConvParams{};
        p.wp = cw.w->dev; p.wp_wino = cw.w->dev_wino; p.wp_w2d = cw.w->dev_w2d; p.out = out_raw; p.bias = cw.b->dev; p.B = B; p.H = oh; p.W = ow; p.Cs = ci; p.Cout = ci;
        if (can_fuse(cw, ci, ow)) {
            p.in_mode = 2;
            p.x = pend.live ? pend.raw : in;
            if (pend.live) { p.in_affine = pend.affine; p.in_skip = pend.skip; p.in_act = pend.act; }
            pend.live = false;
            return 0;
        }
        SBGM_CHECK(!pend.live, "decoder: a pending normalisation reached an unfused convolution");
        float* up = wsalloc((size_t)B * oh * ow * ci);
        if (!up) return 1;
        if (sbgm_launch_upsample2x(in, up, B, oh / 2, ow / 2, ci, st)) return 1;
        p.x = up;
        return 0;
    };
    cur = fm[4]; ch = fh[4]; cw_ = fw[4];
    for (int i = 0; i < 4; ++i) {
        DecW& d = dec[i];
        const int oh = 2 * ch, ow = 2 * cw_;
        SBGM_CHECK(oh == fh[3 - i] && ow == fw[3 - i] && d.cout == FMAP_CH[3 - i], "decoder/skip shape mismatch at block %d", i);
        float* a = wsalloc((size_t)B * oh * ow * d.cin);
        if (!a) return 1;
        ConvParams p{};
        int gn1_chunks = 0;
        if (cfg.decoder_transpose) {                 // ConvTranspose2d: 1x1 conv to 4*cin phase-major channels, then depth -> space
            float* up = wsalloc((size_t)B * oh * ow * d.cin);
            if (!up) return 1;
            p.x = cur; p.wp = d.up.w->dev; p.out = up; p.bias = d.up.b->dev; p.B = B; p.H = ch; p.W = cw_; p.Cs = d.cin; p.Cout = 4 * d.cin;
            if (conv(ConvGeom{1, 1, 1, 0}, p, st)) return 1;
            if (sbgm_launch_depth_space2(up, a, B, ch, cw_, d.cin, 1, st)) return 1;
        } else {
            if (conv_up(d.up, cur, d.cin, oh, ow, p, a)) return 1;
            p.gn_stats = stats; p.gn_groups = groups(d.cin);        // GroupNorm statistics in the epilogue when the LDS kernel runs
            if (conv(ConvGeom{3, 3, 1, 1}, p, st)) return 1;
            gn1_chunks = tile_gn_chunks(p, last_tile);
        }
        float* c2 = wsalloc((size_t)B * oh * ow * d.cout);
        if (!c2) return 1;
        if (ensure_stats(a, oh * ow, d.cin, gn1_chunks)) return 1;
        p = ConvParams{};
        p.B = B; p.H = oh; p.W = ow; p.Cs = d.cin;
        p.x = a; p.wp = d.conv.w->dev; p.wp_wino = d.conv.w->dev_wino; p.wp_w2d = d.conv.w->dev_w2d; p.out = c2; p.bias = d.conv.b->dev; p.Cout = d.cout;
        if (can_fuse(d.conv, d.cin, ow)) {           // norm1 applied while `conv` stages its patch
            float* aff1 = wsalloc((size_t)B * d.cin * 2);
            if (!aff1) return 1;
            if (sbgm_launch_gn_finalize(stats, gn1_chunks, d.n1g ? d.n1g->dev : nullptr, d.n1b ? d.n1b->dev : nullptr, nullptr, aff1, B, oh * ow,
                                        d.cin, groups(d.cin), GN_EPS, st)) return 1;
            p.in_mode = 1; p.in_affine = aff1;
        } else if (sbgm_launch_groupnorm_apply(a, a, d.n1g ? d.n1g->dev : nullptr, d.n1b ? d.n1b->dev : nullptr, nullptr, nullptr,
                                               SBGM_ACT_NONE, B, oh * ow, d.cin, groups(d.cin), GN_EPS, stats, gn1_chunks, st)) return 1;
        p.gn_stats = stats; p.gn_groups = groups(d.cout);
        if (conv(ConvGeom{3, 3, 1, 1}, p, st)) return 1;
        int gn2_chunks = tile_gn_chunks(p, last_tile);
        if (ensure_stats(c2, oh * ow, d.cout, gn2_chunks)) return 1;
        const ConvW& next_up = i < 3 ? dec[i + 1].up : fin_up;
        if (!d.has_attn && can_fuse(next_up, d.cout, 2 * ow)) {
            // norm2 + skip + time bias + activation stay pending: the next conv_up applies them to the low-res pixels it loads
            float* aff2 = wsalloc((size_t)B * d.cout * 2);
            if (!aff2) return 1;
            if (sbgm_launch_gn_finalize(stats, gn2_chunks, d.n2g ? d.n2g->dev : nullptr, d.n2b ? d.n2b->dev : nullptr, tb[5 + i], aff2, B, oh * ow,
                                        d.cout, groups(d.cout), GN_EPS, st)) return 1;
            pend = Pending{c2, aff2, fm[3 - i], cfg.decoder_activation, true};
        } else {
            if (sbgm_launch_groupnorm_apply(c2, c2, d.n2g ? d.n2g->dev : nullptr, d.n2b ? d.n2b->dev : nullptr, fm[3 - i], tb[5 + i],
                                            cfg.decoder_activation, B, oh * ow, d.cout, groups(d.cout), GN_EPS, stats, gn2_chunks, st)) return 1;
            if (d.has_attn && attention(d.attn, c2, B, oh * ow, st)) return 1;
        }
        cur = c2; ch = oh; cw_ = ow;
    }
    {   // final block: no norms, no skip, no time, identity activation (score_unet.py:726-730, :757)
        const int ci = dec[3].cout;
        ConvParams p{};
        if (cfg.decoder_transpose) {
            float* up = wsalloc((size_t)B * H * W * ci);
            if (!up) return 1;
            float* a = wsalloc((size_t)B * H * W * ci);
            if (!a) return 1;
            p.x = cur; p.wp = fin_up.w->dev; p.out = up; p.bias = fin_up.b->dev; p.B = B; p.H = ch; p.W = cw_; p.Cs = ci; p.Cout = 4 * ci;
            if (conv(ConvGeom{1, 1, 1, 0}, p, st)) return 1;
            if (sbgm_launch_depth_space2(up, a, B, ch, cw_, ci, 1, st)) return 1;
            return sbgm_launch_conv3x3_cout1(a, fin_conv.w->dev, fin_conv.b->dev, t, cfg.sigma, out, B, H, W, ci, st);
        }
        if (conv_up(fin_up, cur, ci, H, W, p, nullptr)) return 1;
        if (ci == 64) {
            // conv_up's 64-channel output feeds only the linear 3x3 Cout=1 conv: project onto its 9 taps in the epilogue
            // (9 floats per pixel instead of 64) and finish with a 9-point gather.
            p.proj_w = fin_conv.w->dev;
            float* d = wsalloc((size_t)9 * B * H * W * 4);      // up to 4 partial planes (2-D Winograd tiles of 16 channels)
            if (!d) return 1;
            p.out = d; p.proj_out = d;
            if (conv(ConvGeom{3, 3, 1, 1}, p, st)) return 1;
            const int parts = last_tile.wino == 2 ? sbgm_conv_w2d_proj_parts(p, last_tile) : 1;
            if (sbgm_launch_tap_stencil(d, fin_conv.b->dev, t, cfg.sigma, out, B, H, W, st, parts)) return 1;
        } else {
            float* a = wsalloc((size_t)B * H * W * ci);
            if (!a) return 1;
            p.out = a;
            if (conv(ConvGeom{3, 3, 1, 1}, p, st)) return 1;
            if (sbgm_launch_conv3x3_cout1(a, fin_conv.w->dev, fin_conv.b->dev, t, cfg.sigma, out, B, H, W, ci, st)) return 1;
        }
    }
    return 0;
}

// torch.linspace(start, end, n) in fp32, as ATen fills it (symmetric about the midpoint)
static std::vector<float> linspace_f32(float start, float end, int n) {
    std::vector<float> v(n);
    if (n == 1) { v[0] = start; return v; }
    const float step = (end - start) / (float)(n - 1);
    const int half = n / 2;
    for (int i = 0; i < n; ++i) v[i] = i < half ? start + step * (float)i : end - step * (float)(n - 1 - i);
    return v;
}

int sbgm_model::sampler(const sbgm_sampler_args& a, hipStream_t caller) {
    // Graph CAPTURE is illegal on the legacy default stream, so the step is captured on a private stream (capture records, it runs
    // nothing); the REPLAYS, the uploads and the final copy go to the caller's stream, so the run is ordinary stream-ordered work of the
    // caller.  (Round 2 also replayed on the private stream, fenced with events on both sides: measured 1.639 vs 1.589 ms per C2 step —
    // the same kernels dispatch 0.7 us apart closer on the caller's stream; SBGM_GRAPH_PRIVATE_STREAM=1 restores that form for A/B runs.)
    hipStream_t st = caller;
    const bool graphed = a.use_graph && !a.noise;
    static const bool private_replay = getenv("SBGM_GRAPH_PRIVATE_STREAM") != nullptr;
    const bool replay_on_caller = graphed && !private_replay;
    if (graphed) {
        if (!graph_stream) {
            SBGM_HIP(hipStreamCreateWithFlags(&graph_stream, hipStreamNonBlocking));
            SBGM_HIP(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
            SBGM_HIP(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
        }
        if (!replay_on_caller) {
            SBGM_HIP(hipEventRecord(ev_in, caller));
            SBGM_HIP(hipStreamWaitEvent(graph_stream, ev_in, 0));
            st = graph_stream;
        }
    }
    SBGM_CHECK(a.kind == SBGM_SAMPLER_EM || a.kind == SBGM_SAMPLER_PC, "sampler: unknown kind %d", a.kind);
    SBGM_CHECK(a.num_steps >= 2, "sampler: num_steps=%d must be >= 2 (step size = t0 - t1)", a.num_steps);
    SBGM_CHECK(a.out != nullptr, "sampler: out is required");
    const int B = a.B, H = a.H, W = a.W, N = a.num_steps;
    const bool guided = a.cfg_enabled != 0;
    const int BE = guided ? 2 * B : B;                     // samples per network evaluation
    const size_t per = (size_t)H * W, n = (size_t)B * per;
    SBGM_CHECK(ws_need(BE, H, W, a.bn_train) <= ws_bytes, "sampler: workspace not prepared for B=%d H=%d W=%d", BE, H, W);
    // ---- per-step scalars on the host, in the reference's precision, written into the pinned staging buffer ---------------------
    const size_t stage_need = sizeof(StepScalars) * (size_t)N + sizeof(SamplerState);
    if (stage_pending) {                                   // the previous call's upload still reads the buffer (normally long done)
        SBGM_HIP(hipEventSynchronize(ev_stage));
        stage_pending = false;
    }
    if (h_stage_bytes < stage_need) {
        if (h_stage) SBGM_HIP(hipHostFree(h_stage));
        h_stage = nullptr;
        h_stage_bytes = std::max(stage_need, sizeof(StepScalars) * (size_t)4096 + sizeof(SamplerState));
        SBGM_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_stage), h_stage_bytes, hipHostMallocDefault));
    }
    if (!ev_stage) SBGM_HIP(hipEventCreateWithFlags(&ev_stage, hipEventDisableTiming));
    StepScalars* tab = reinterpret_cast<StepScalars*>(h_stage);
    const float sig = cfg.sigma;
    if (a.kind == SBGM_SAMPLER_EM) {                       // score_sampling.py:96-97, :102-103, :124-125
        const std::vector<float> ts = linspace_f32(1.0f, a.eps, N);
        const float dt = ts[0] - ts[1];
        for (int i = 0; i < N; ++i) {
            const float g = powf(sig, ts[i]);
            tab[i] = StepScalars{ts[i], g * g, dt, sqrtf(dt) * g, ts[std::min(i + 1, N - 1)]};
        }
    } else {                                               // score_sampling.py:169-170, :176, :207, :224-227
        std::vector<double> ts(N);
        const double step = ((double)a.eps - 1.0) / (double)(N - 1);
        for (int i = 0; i < N; ++i) ts[i] = 1.0 + (double)i * step;
        ts[N - 1] = (double)a.eps;
        const float dt = (float)(ts[0] - ts[1]);
        for (int i = 0; i < N; ++i) {
            const float tf = (float)ts[i];
            const float g = powf(sig, tf);
            tab[i] = StepScalars{tf, g * g, dt, sqrtf((g * g) * dt), (float)ts[std::min(i + 1, N - 1)]};
        }
    }
    const float t_first = tab[0].t;
    if (table_cap < N) {
        if (d_table) SBGM_HIP(hipFree(d_table));
        d_table = nullptr;
        const int cap = std::max(N, 4096);                   // roomy: a new table address invalidates the cached step graph
        SBGM_HIP(hipMalloc(&d_table, sizeof(StepScalars) * cap));
        table_cap = cap;
    }
    SamplerState* state0 = reinterpret_cast<SamplerState*>(h_stage + sizeof(StepScalars) * (size_t)N);
    *state0 = SamplerState{0ull, 0ull, (unsigned long long)a.seed, (unsigned long long)N};
    SBGM_HIP(hipMemcpyAsync(d_table, tab, sizeof(StepScalars) * N, hipMemcpyHostToDevice, st));
    SBGM_HIP(hipMemcpyAsync(d_state, state0, sizeof(SamplerState), hipMemcpyHostToDevice, st));
    SBGM_HIP(hipEventRecord(ev_stage, st));               // no host wait: the copies read pinned memory this handle owns
    stage_pending = true;

    // persistent sampler buffers live at the top of the workspace, the forward uses the rest
    // layout (BE = B, or 2B with guidance): x [BE*per] (rows B.. mirror rows 0..B-1), score [BE*per], x_mean [B*per]
    const size_t keep = sampler_keep(BE, H, W);
    const size_t slab = align_up((size_t)BE * per * 4, 256);
    char* top = ws + ws_bytes - keep;
    float* xs = reinterpret_cast<float*>(top);
    float* score = reinterpret_cast<float*>(top + slab);
    float* xmean = reinterpret_cast<float*>(top + 2 * slab);
    float* t_dev = reinterpret_cast<float*>(top + 3 * slab);
    double* sumsq = reinterpret_cast<double*>(top + 3 * slab + align_up((size_t)BE * 4, 256));
    const size_t fwd_bytes = ws_bytes - keep;
    if (bn_dirty && fold_bn(st)) return 1;              // keep the fold out of the captured step

    // guidance: the unconditional half of the condition tensors is built once per run (guided_score_fn :27-43)
    const int64_t* y_e = a.y;
    const float *cond_e = a.cond_img, *lsm_e = a.lsm_cond, *topo_e = a.topo_cond;
    struct Scratch {                                       // freed on every exit path, after the stream has drained
        std::vector<void*> v;
        hipStream_t st;
        ~Scratch() {
            if (v.empty()) return;
            (void)hipStreamSynchronize(st);
            for (void* p : v) (void)hipFree(p);
        }
    } guided_bufs{{}, st};
    if (guided) {
        SBGM_CHECK(!a.bn_train, "sampler: guidance with train-mode BatchNorm would couple the two halves of the batch");
        auto dup = [&](const void* src, size_t bytes_half, int mode, int channels) -> void* {   // mode 0 zero, 1 copy, 2 strip mask
            void* p = nullptr;
            if (hipMalloc(&p, 2 * bytes_half) != hipSuccess) return nullptr;
            guided_bufs.v.push_back(p);
            (void)hipMemcpyAsync(p, src, bytes_half, hipMemcpyDeviceToDevice, st);
            char* lo = static_cast<char*>(p) + bytes_half;
            if (mode == 0) (void)hipMemsetAsync(lo, 0, bytes_half, st);
            else (void)hipMemcpyAsync(lo, src, bytes_half, hipMemcpyDeviceToDevice, st);
            if (mode == 2 && channels == 2)                  // NCHW [B][2][H][W]: zero channel 1 of every sample
                (void)hipMemset2DAsync(lo + per * 4, 2 * per * 4, 0, per * 4, B, st);
            return p;
        };
        bool ok = true;
        if (a.y) ok = ok && (y_e = static_cast<const int64_t*>(dup(a.y, (size_t)B * 8, 0, 0)));          // null token 0
        if (a.cond_img) ok = ok && (cond_e = static_cast<const float*>(dup(a.cond_img, n * 4 * cfg.n_cond_channels, 0, 0)));
        if (a.lsm_cond) ok = ok && (lsm_e = static_cast<const float*>(dup(a.lsm_cond, n * 4 * cfg.n_lsm_channels, 2, cfg.n_lsm_channels)));
        if (a.topo_cond) ok = ok && (topo_e = static_cast<const float*>(dup(a.topo_cond, n * 4 * cfg.n_topo_channels, 2, cfg.n_topo_channels)));
        SBGM_CHECK(ok && hipGetLastError() == hipSuccess, "sampler: could not allocate the unconditional condition tensors");
    }

    // x0 = randn * marginal_prob_std(1)
    const float ls = logf(sig);
    const float std1 = fmaxf(sqrtf((expf((2.f * 1.0f) * ls) - 1.f) / (2.f * ls)), 1e-5f);
    const float* z = a.noise;
    size_t draw = 0;
    auto next_z = [&]() -> const float* { const float* p = z ? z + draw * n : nullptr; ++draw; return p; };
    NoiseMap nm{};
    if (a.tile_origins) {
        SBGM_CHECK(W % 4 == 0 && a.domain_w >= W, "sampler: tiled noise needs W %% 4 == 0 and domain_w >= W (W=%d, domain_w=%d)", W,
                   a.domain_w);
        nm = NoiseMap{a.tile_origins, H, W / 4, (a.domain_w + 3) / 4};
    }
    if (sbgm_launch_init_noise(xs, std1, next_z(), a.seed, d_state, 0, n, st, nm)) return 1;
    if (sbgm_launch_fill_t(t_dev, t_first, BE, st)) return 1;
    const float snr_nn = (float)((double)a.snr * std::sqrt((double)per));     // snr * sqrt(prod(x.shape[1:])) (:202-203)

    // one (possibly guided) score evaluation of the current x into score[0 .. n)
    auto evaluate = [&](float w) -> int {
        if (guided) SBGM_HIP(hipMemcpyAsync(xs + n, xs, n * 4, hipMemcpyDeviceToDevice, st));
        if (forward(xs, t_dev, y_e, cond_e, lsm_e, topo_e, score, nullptr, BE, H, W, a.bn_train, st)) return 1;
        return guided ? sbgm_launch_cfg_combine(score, score, score + n, w, n, st) : 0;
    };
    auto one_step = [&](bool with_noise_ptrs) -> int {
        if (a.kind == SBGM_SAMPLER_PC) {
            if (evaluate(a.cfg_scale_corrector)) return 1;
            if (sbgm_launch_langevin(xs, score, with_noise_ptrs ? next_z() : nullptr, snr_nn, sumsq, d_state, 0, a.seed, B, per, st, nm)) return 1;
        }
        if (evaluate(a.cfg_scale)) return 1;
        return sbgm_launch_em_update(xs, xmean, score, with_noise_ptrs ? next_z() : nullptr, d_table, d_state, nullptr, 0, t_dev,
                                     a.seed, B, per, N, st, BE, nm);
    };

    const size_t saved_ws = ws_bytes;
    ws_bytes = fwd_bytes;            // forward() must not touch the sampler slabs
    int rc = 0;
    if (graphed) {
        StepGraphKey key{};                                  // (value-initialised: the padding bytes compare equal)
        key.B = B; key.H = H; key.W = W; key.kind = a.kind; key.guided = guided; key.bn_train = a.bn_train; key.table = d_table;
        key.domain_w = a.domain_w; key.y = y_e; key.cond = cond_e; key.lsm = lsm_e; key.topo = topo_e; key.origins = a.tile_origins;
        key.ws = ws; key.ws_bytes = saved_ws; key.cfg = a.cfg_scale; key.cfg_corr = a.cfg_scale_corrector; key.snr_nn = snr_nn;
        key.plan_gen = plan_gen;
        const bool reuse = step_exec != nullptr && !guided && std::memcmp(&key, &step_key, sizeof key) == 0;
        if (!reuse) {
            drop_step_graph();
            const hipStream_t run_st = st;
            if (replay_on_caller) st = graph_stream;          // capture (records, runs nothing) on the private stream; replay on the caller's
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                rc = one_step(false);
                hipError_t e2 = hipStreamEndCapture(st, &step_graph);
                st = run_st;
                if (rc == 0 && (e2 != hipSuccess || hipGraphInstantiate(&step_exec, step_graph, nullptr, nullptr, 0) != hipSuccess)) {
                    sbgm_set_error("hipGraph capture/instantiate failed: %s", hipGetErrorString(e2));
                    rc = 2;
                }
                if (rc) drop_step_graph();
                else step_key = key;
            } else {
                st = run_st;
                sbgm_set_error("hipStreamBeginCapture failed: %s", hipGetErrorString(e));
                rc = 2;
            }
        }
        for (int i = 0; i < N && rc == 0; ++i)
            if (hipGraphLaunch(step_exec, st) != hipSuccess) { sbgm_set_error("hipGraphLaunch failed at step %d", i); rc = 2; }
        if (guided) {                                        // its condition copies are freed when this call returns
            (void)hipStreamSynchronize(st);
            drop_step_graph();
        }
    } else {
        for (int i = 0; i < N && rc == 0; ++i) rc = one_step(z != nullptr);
    }
    ws_bytes = saved_ws;
    if (rc) return rc;
    SBGM_HIP(hipMemcpyAsync(a.out, xmean, n * 4, hipMemcpyDeviceToDevice, st));
    if (graphed && !replay_on_caller) {
        SBGM_HIP(hipEventRecord(ev_out, st));
        SBGM_HIP(hipStreamWaitEvent(caller, ev_out, 0));
    }
    return 0;
}

// =====================================================================================================================
// C ABI
// =====================================================================================================================
const char* sbgm_get_error();
extern "C" {

const char* sbgm_last_error(void) { return sbgm_get_error(); }
int sbgm_abi_version(void) { return 4; }
int sbgm_model_config_size(void) { return (int)sizeof(sbgm_model_config); }

int sbgm_model_create(const sbgm_model_config* cfg, sbgm_model** out) {
    SBGM_CHECK(cfg && out, "model_create: null argument");
    SBGM_CHECK(cfg->struct_size == (int)sizeof(sbgm_model_config),
               "model_create: sbgm_model_config.struct_size = %d but this library's struct has %d bytes (ABI version %d): the caller "
               "was built against a different include/sbgm_hip.h", cfg->struct_size, (int)sizeof(sbgm_model_config), sbgm_abi_version());
    std::unique_ptr<sbgm_model> m(new sbgm_model());
    if (m->build(*cfg)) return 1;
    *out = m.release();
    return 0;
}
void sbgm_model_destroy(sbgm_model* m) { delete m; }
int sbgm_model_num_params(const sbgm_model* m) { return (int)m->params.size(); }
const char* sbgm_model_param_name(const sbgm_model* m, int i) {
    return (i >= 0 && i < (int)m->params.size()) ? m->params[i]->name.c_str() : nullptr;
}
int64_t sbgm_model_param_numel(const sbgm_model* m, int i) {
    return (i >= 0 && i < (int)m->params.size()) ? m->params[i]->numel : -1;
}

int sbgm_model_set_param(sbgm_model* m, const char* name, const void* data, int64_t numel, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    auto it = m->by_name.find(name);
    SBGM_CHECK(it != m->by_name.end(), "set_param: unexpected key '%s'", name);
    Param* p = it->second;
    if (p->kind == P_IGNORE) return 0;
    SBGM_CHECK(numel == p->numel, "set_param: '%s' has %lld elements, expected %lld", name, (long long)numel, (long long)p->numel);
    const float* src = static_cast<const float*>(data);
    if (p->kind == P_VEC) {
        SBGM_HIP(hipMemcpyAsync(p->dev, src, (size_t)numel * 4, hipMemcpyDeviceToDevice, st));
    } else if (p->kind == P_CONV) {
        if (sbgm_launch_pack_conv_weight(src, p->dev, p->cout, p->cin, p->kh, p->kw, p->cs, st)) return 1;
        if (p->wino && sbgm_launch_pack_wino_weight(src, p->dev_wino, p->cout, p->cin, p->cs, st)) return 1;
        if (p->dev_w2d && sbgm_launch_pack_w2d_weight(src, p->dev_w2d, p->cout, p->cin, p->cs, st)) return 1;
    } else if (p->kind == P_TCONV) {             // [Cin][Cout][2][2] -> OIHW [4*Cout][Cin][1][1] (scratch) -> packed
        float* tmp = nullptr;
        SBGM_HIP(hipMalloc(&tmp, (size_t)numel * 4));
        int rc = sbgm_launch_tconv_weight(src, tmp, p->cin, p->cout / 4, st);
        if (!rc) rc = sbgm_launch_pack_conv_weight(tmp, p->dev, p->cout, p->cin, 1, 1, p->cs, st);
        (void)hipStreamSynchronize(st);
        (void)hipFree(tmp);
        if (rc) return rc;
    } else if (p->kind == P_VEC4) {
        for (int r = 0; r < 4; ++r)
            SBGM_HIP(hipMemcpyAsync(p->dev + (size_t)r * numel, src, (size_t)numel * 4, hipMemcpyDeviceToDevice, st));
    } else {
        if (sbgm_launch_pack_cout1_weight(src, p->dev, p->cin, st)) return 1;
    }
    p->filled = true;
    m->bn_dirty = true;
    return 0;
}

int sbgm_model_get_param(sbgm_model* m, const char* name, float* dst, int64_t numel, void* stream) {
    auto it = m->by_name.find(name);
    SBGM_CHECK(it != m->by_name.end(), "get_param: unknown key '%s'", name);
    Param* p = it->second;
    SBGM_CHECK(p->kind == P_VEC && numel == p->numel, "get_param: '%s' is not a plain vector of %lld elements", name, (long long)numel);
    SBGM_HIP(hipMemcpyAsync(dst, p->dev, (size_t)numel * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int64_t sbgm_model_workspace_bytes(const sbgm_model* m) { return (int64_t)m->ws_bytes; }

int sbgm_model_check_complete(const sbgm_model* m) {
    for (auto& p : m->params) SBGM_CHECK(p->filled, "model: state_dict entry '%s' was never uploaded", p->name.c_str());
    return 0;
}

int sbgm_model_forward(sbgm_model* m, const float* x, const float* t, const int64_t* y, const float* cond_img,
                       const float* lsm_cond, const float* topo_cond, float* out, float* const* fmaps, int B, int H, int W,
                       int bn_train, void* stream) {
    if (m->ensure_ws(m->ws_need(B, H, W, bn_train))) return 1;
    return m->forward(x, t, y, cond_img, lsm_cond, topo_cond, out, fmaps, B, H, W, bn_train, (hipStream_t)stream);
}

int sbgm_sampler_run(sbgm_model* m, const sbgm_sampler_args* a, void* stream) {
    SBGM_CHECK(a, "sampler_run: null args");
    if (m->prepare_ws(a->cfg_enabled ? 2 * a->B : a->B, a->H, a->W, a->bn_train, 1, (hipStream_t)stream)) return 1;
    return m->sampler(*a, (hipStream_t)stream);
}

// One evaluation of the (B, H, W) plan on zero inputs placed at the top of the workspace.  tune = true: every convolution times its tile
// candidates; tune = false: a plain evaluation whose only purpose is the bump allocator's high-water mark (ws_peak).
int sbgm_model::dummy_forward(int B, int H, int W, bool tune, hipStream_t st) {
    const size_t px = (size_t)B * H * W;
    const size_t in_floats = px * 16 + 1024;
    float* inp = reinterpret_cast<float*>(ws + ws_bytes - align_up(in_floats * 4, 256));
    SBGM_HIP(hipMemsetAsync(inp, 0, in_floats * 4, st));
    float* x = inp; float* t = inp + px; float* cond = t + 1024; float* lsm = cond + px * 8; float* topo = lsm + px * 2;
    float* out = topo + px * 2;
    if (sbgm_launch_fill_t(t, 0.5f, B, st)) return 1;
    const size_t saved = ws_bytes;
    ws_bytes -= align_up(in_floats * 4, 256);
    tuning = tune;
    ws_used = 0;
    const int rc = forward(x, t, nullptr, cfg.n_cond_channels ? cond : nullptr, cfg.n_lsm_channels ? lsm : nullptr,
                           cfg.n_topo_channels ? topo : nullptr, out, nullptr, B, H, W, 0, st);
    tuning = false;
    ws_bytes = saved;
    SBGM_HIP(hipStreamSynchronize(st));
    return rc;
}

// Workspace for an eval-mode call of this shape.  A shape seen for the first time is measured by one extra evaluation on zero inputs
// BEFORE the caller's work, so the slab has its final size (and address) from the first real call on: a sampler's captured step graph
// is then captured once, not again after a later call has trimmed the slab.  Train-mode BatchNorm shapes are not pre-measured (an
// evaluation would move the running statistics): they run on the generous bound first and are trimmed by a later call.
int sbgm_model::prepare_ws(int B, int H, int W, int bn_train, size_t factor, hipStream_t st) {
    if (!bn_train && ws_peak.find(std::array<int, 4>{B, H, W, 0}) == ws_peak.end() && sbgm_model_check_complete(this) == 0) {
        if (ensure_ws(ws_need(B, H, W) + ((size_t)B * H * W * 16 + 1024) * 4 + 4096)) return 1;
        if (bn_dirty && fold_bn(st)) return 1;
        if (dummy_forward(B, H, W, false, st)) return 1;
    }
    return ensure_ws(factor * ws_need(B, H, W, bn_train));
}

int sbgm_model_autotune(sbgm_model* m, int B, int H, int W, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (m->ensure_ws(2 * m->ws_need(B, H, W))) return 1;
    if (m->bn_dirty && m->fold_bn(st)) return 1;
    if (m->dummy_forward(B, H, W, true, st)) return 1;
    // the tuned plan's high-water mark, then the slab at its final size: what runs next (a sampler capturing its step) finds both settled
    if (m->dummy_forward(B, H, W, false, st)) return 1;
    return m->ensure_ws(m->ws_need(B, H, W));
}

extern "C++" {
static std::string conv_kernel_name(const sbgm_model::ConvRec& r) {
    char b[96];
    if (r.t.wino == 2 && r.t.lds == 3) snprintf(b, sizeof b, "conv3x3_w2dp_kernel<%d; %d; %s>", r.t.fco, r.in_mode, r.proj ? "true" : "false");
    else if (r.t.wino == 2) snprintf(b, sizeof b, "conv3x3_w2d_kernel<%d; %d; %s; %d>", r.t.fco, r.t.ws == 2 ? 2 : 1, r.t.lds == 2 ? "true" : "false", r.in_mode);
    else if (r.t.lds) snprintf(b, sizeof b, "conv3x3_lds_kernel<%d; %d; %s; %s; %d>", r.t.fco, r.t.fpx, r.t.wino ? "true" : "false", r.t.lds == 2 ? "true" : "false", r.in_mode);
    else if (r.t.wino) snprintf(b, sizeof b, "conv3x3_wino_kernel<%d; %d; %d>", r.t.fco, r.t.fpx, r.t.ws);
    else snprintf(b, sizeof b, "conv_igemm_kernel<%d; %d; %d; %d; %d; %d; %d; %d>", r.g.kh, r.g.kw, r.g.stride, r.g.pad, r.t.fco,
                  r.t.fpx, r.c_real == 2 ? 2 : (r.Cs >= 16 ? 0 : r.Cs), r.t.ws);
    return b;
}
}  // extern "C++"

// Tile table <-> text file: one line per tuned convolution, "kh kw stride pad B H W Cin_pad Cout proj in_mode | fco fpx splits ws wino lds".
int sbgm_model_tune_save(sbgm_model* m, const char* path) {
    SBGM_CHECK(path, "tune_save: null path");
    FILE* f = fopen(path, "w");
    SBGM_CHECK(f, "tune_save: cannot open %s", path);
    fprintf(f, "# sbgm conv tile table v2\n");
    for (auto& kv : m->tuned) {
        const ConvOpKey& k = kv.first;
        const ConvTile& t = kv.second;
        fprintf(f, "%d %d %d %d %d %d %d %d %d %d %d | %d %d %d %d %d %d\n", k.kh, k.kw, k.s, k.p, k.B, k.H, k.W, k.Cs, k.Cout, k.proj,
                k.in_mode, t.fco, t.fpx, t.splits, t.ws, t.wino, t.lds);
    }
    fclose(f);
    return 0;
}

int sbgm_model_tune_load(sbgm_model* m, const char* path) {
    SBGM_CHECK(path, "tune_load: null path");
    FILE* f = fopen(path, "r");
    SBGM_CHECK(f, "tune_load: cannot open %s", path);
    char line[256];
    std::map<ConvOpKey, ConvTile> table;
    int lineno = 0;
    while (fgets(line, sizeof line, f)) {
        ++lineno;
        if (line[0] == '#' || line[0] == '\n') continue;
        ConvOpKey k{};
        int t[6];
        const int n = sscanf(line, "%d %d %d %d %d %d %d %d %d %d %d | %d %d %d %d %d %d", &k.kh, &k.kw, &k.s, &k.p, &k.B, &k.H, &k.W,
                             &k.Cs, &k.Cout, &k.proj, &k.in_mode, &t[0], &t[1], &t[2], &t[3], &t[4], &t[5]);
        // the launchers reject tiles they do not instantiate; here only the ranges that index memory are checked
        const bool ok = n == 17 && k.in_mode >= 0 && k.in_mode <= 2 && (t[0] == 1 || t[0] == 2 || t[0] == 4) && (t[1] == 1 || t[1] == 2 || t[1] == 4) && t[2] >= 1 &&
                        t[2] <= 64 && (t[3] == 1 || t[3] == 2 || t[3] == 4 || t[3] == 8) && t[4] >= 0 && t[4] <= 2 && (t[4] != 2 || t[5] >= 1) && t[5] >= 0 && t[5] <= (t[4] == 2 ? 3 : 2) &&
                        k.Cout % (16 * t[0]) == 0;
        if (!ok) {
            fclose(f);
            SBGM_CHECK(false, "tune_load: %s line %d is malformed", path, lineno);
        }
        table[k] = ConvTile{t[0], t[1], t[2], t[3], t[4], t[5]};
    }
    fclose(f);
    for (auto& kv : table) m->tuned[kv.first] = kv.second;
    ++m->plan_gen;
    return 0;
}

// Eager forward with every convolution launch bracketed by HIP events on `stream`.  Fills the summary and, when
// csv_path is non-null, writes one line per convolution (geometry, tile, split-K, ms, TFLOP/s).
int sbgm_model_profile_forward(sbgm_model* m, const float* x, const float* t, const int64_t* y, const float* cond_img,
                               const float* lsm_cond, const float* topo_cond, float* out, int B, int H, int W,
                               sbgm_profile* summary, const char* csv_path, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (m->ensure_ws(m->ws_need(B, H, W))) return 1;
    std::vector<sbgm_model::ConvRec> recs;
    hipEvent_t t0, t1;
    SBGM_HIP(hipEventCreate(&t0));
    SBGM_HIP(hipEventCreate(&t1));
    m->prof = &recs;
    SBGM_HIP(hipEventRecord(t0, st));
    const int rc = m->forward(x, t, y, cond_img, lsm_cond, topo_cond, out, nullptr, B, H, W, 0, st);
    SBGM_HIP(hipEventRecord(t1, st));
    m->prof = nullptr;
    if (rc) return rc;
    SBGM_HIP(hipEventSynchronize(t1));
    sbgm_profile s{};
    SBGM_HIP(hipEventElapsedTime(&s.ms_total_with_events, t0, t1));
    FILE* f = csv_path ? fopen(csv_path, "w") : nullptr;
    if (f) fprintf(f, "idx,kh,kw,stride,B,H,W,Cin_pad,Cout,M,ksteps,tile_co,tile_px,splits,ws,gflop,ms,tflops,kernel\n");
    int i = 0;
    for (auto& r : recs) {
        SBGM_HIP(hipEventElapsedTime(&r.ms, r.e0, r.e1));
        r.ms /= sbgm_model::PROF_REPS;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
        s.ms_conv += r.ms;
        s.flops_conv += r.flops;
        s.n_conv += 1;
        if (r.ms > s.ms_conv_max) { s.ms_conv_max = r.ms; s.flops_conv_max = r.flops; }
        if (f) fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.4f,%.4f,%.2f,%s\n", i, r.g.kh, r.g.kw, r.g.stride, r.B, r.H, r.W,
                       r.Cs, r.Cout, r.M, r.nsteps, 16 * r.t.fco, r.t.wino == 2 ? 256 : r.t.lds ? 64 * r.t.fpx * (r.t.wino ? 2 : 1) : (r.t.wino ? 32 : 16) * r.t.fpx, r.t.splits, r.t.wino == 2 ? -40 : r.t.lds ? (r.t.wino ? -20 : 20) : (r.t.wino ? -r.t.ws : r.t.ws), r.flops * 1e-9, r.ms,
                       r.flops / (r.ms * 1e-3) * 1e-12, conv_kernel_name(r).c_str());
        ++i;
    }
    if (f) fclose(f);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    if (summary) *summary = s;
    return 0;
}

int sbgm_event_create(void** ev) { hipEvent_t e; SBGM_HIP(hipEventCreate(&e)); *ev = e; return 0; }
int sbgm_event_record(void* ev, void* stream) { SBGM_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return 0; }
int sbgm_event_elapsed_ms(void* start, void* stop, float* ms) {
    SBGM_HIP(hipEventSynchronize((hipEvent_t)stop));
    SBGM_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}
int sbgm_event_destroy(void* ev) { SBGM_HIP(hipEventDestroy((hipEvent_t)ev)); return 0; }

}  // extern "C"
