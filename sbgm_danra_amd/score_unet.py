"""Drop-in mirror of the reference's `sbgm/score_unet.py` public surface, with the bodies running on
libsbgm_hip.so (hand-written gfx950 kernels) instead of torch.nn ops.

Kept identical to the reference: class names, constructor signatures, `state_dict()` keys / shapes / layouts
(OIHW convs, packed `mha.in_proj_weight`, BatchNorm buffers, the six Gaussian-Fourier `W` buffers), the call
signature `ScoreNet.forward(x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None)`, NCHW fp32 tensors at
the boundary and the exceptions raised for malformed input (reference score_unet.py:274-280, :596-597, :965-967).

The torch.nn sub-modules below are parameter CONTAINERS only (so `load_state_dict`, `.to()`, `.parameters()`,
optimizers and checkpoints behave exactly as in the reference); their own `forward` is never called.  The
arithmetic happens in the native engine, which repacks the weights to NHWC / K-major on upload and is refreshed
whenever a parameter's version counter changes.  There is no CPU fallback: a CPU tensor or a missing
libsbgm_hip.so raises.
"""
from __future__ import annotations

import ctypes as C
import functools
import logging
from typing import Iterable, Optional

import torch
import torch.nn as nn

from . import _native as N

logger = logging.getLogger(__name__)

FMAP_CHANNELS = [64, 64, 128, 256, 512]          # reference score_unet.py:198


# --------------------------------------------------------------------------------------------------------------
# parameter containers (same attribute names as the reference / torchvision -> same state_dict keys)
# --------------------------------------------------------------------------------------------------------------
class _NativeOnly(nn.Module):
    """Sub-modules hold parameters; inside a ScoreNet the whole-network engine evaluates them.  Encoder, DecoderBlock and Decoder
    can also be called on their own like the reference's modules (NCHW tensors in and out): those calls run the same native
    kernels op by op (train_graph.py), with autograd in train() mode.  The leaf containers have no stand-alone forward."""

    def forward(self, *a, **k):
        raise NotImplementedError(
            f"{type(self).__name__} is evaluated by the native kernels through ScoreNet / Encoder / DecoderBlock / Decoder; "
            "it has no stand-alone forward")


class SinusoidalEmbedding(_NativeOnly):
    """Gaussian-Fourier time features; container for the fixed `W` buffer (reference score_unet.py:24-45)."""

    def __init__(self, embed_dim: int, scale: float = 30.0, device=None, dtype=torch.float32):
        super().__init__()
        if embed_dim % 2 != 0:
            raise ValueError(f"Embedding dimension must be even, got {embed_dim}.")
        self.register_buffer("W", torch.randn(embed_dim // 2, dtype=dtype, device=device) * scale, persistent=True)


class ImageSelfAttention(_NativeOnly):
    """Pre-LN residual MHA + FF over H*W tokens (reference score_unet.py:112-148)."""

    def __init__(self, input_channels: int, n_heads: int, dropout: float = 0.0):
        super().__init__()
        if input_channels % n_heads != 0:
            raise ValueError(f"Number of input channels ({input_channels}) must be divisible by number of heads ({n_heads}).")
        # dropout > 0 as in the reference's signature (:118-127): the identity in eval mode (all the sampling path needs); in train mode
        # the autograd path drops softmax probabilities inside the attention core (csrc/attention_dropout.hip, Philox mask seeded from
        # torch's generator).  The reference's own DecoderBlock / Encoder never pass a non-zero p.
        self.input_channels, self.n_heads, self.dropout = input_channels, n_heads, float(dropout)
        self.mha = nn.MultiheadAttention(embed_dim=input_channels, num_heads=n_heads, dropout=dropout, batch_first=True)
        self.ln1 = nn.LayerNorm(input_channels)
        self.ln2 = nn.LayerNorm(input_channels)
        self.ff = nn.Sequential(nn.Linear(input_channels, input_channels), nn.GELU(),
                                nn.Linear(input_channels, input_channels))


class BasicBlock(_NativeOnly):
    """torchvision ResNet basic block layout: conv1/bn1/relu/conv2/bn2/downsample."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class Encoder(_NativeOnly):
    """ResNet-18-style encoder: two 8x8/s2 stem convs, 4 stages, time-bias adds, attention on the two deepest maps
    (reference score_unet.py:151-404; stage layout from torchvision ResNet._make_layer)."""

    def __init__(self, input_channels: int, time_embedding: int, block=BasicBlock, block_layers: list = [2, 2, 2, 2],
                 n_heads: int = 4, num_classes: Optional[int] = None, cond_on_img=False, cond_img_dim=None, device=None):
        super().__init__()
        self.block, self.block_layers = block, list(block_layers)
        self.time_embedding = time_embedding
        self.input_channels = input_channels + 1          # + the noised HR field (reference :182)
        self.n_heads, self.num_classes = n_heads, num_classes
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if len(self.block_layers) != 4:
            raise ValueError("block_layers must have 4 entries (ResNet stages)")
        self.conv1 = nn.Conv2d(self.input_channels, 64, kernel_size=(8, 8), stride=(2, 2), padding=(3, 3), bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        inplanes = 64
        for li, (planes, nblocks) in enumerate(zip(FMAP_CHANNELS[1:], self.block_layers), start=1):
            stride = 1 if li == 1 else 2
            blocks = []
            for bi in range(nblocks):
                ds = None
                if bi == 0 and (stride != 1 or inplanes != planes):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
                blocks.append(BasicBlock(inplanes, planes, stride if bi == 0 else 1, ds))
                inplanes = planes
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        # torchvision init (Kaiming fan_out for convs, BN gamma=1 beta=0)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self.sinusoidal_embedding = SinusoidalEmbedding(time_embedding)
        self.time_projection_layers = self.make_time_projections(FMAP_CHANNELS)
        self.attention_layers = self.make_attention_layers(FMAP_CHANNELS)
        self.conv2 = nn.Conv2d(64, 64, kernel_size=(8, 8), stride=(2, 2), padding=(3, 3), bias=False)
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes + 1, time_embedding)
            with torch.no_grad():
                self.label_emb.weight[0].fill_(0.0)       # null class (reference :224-226)

    def forward(self, x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None):
        """-> (fmap1 .. fmap5), NCHW (reference score_unet.py:247-364)"""
        from .train_graph import encoder_call
        return encoder_call(self, x, t, y, cond_img, lsm_cond, topo_cond)

    def make_time_projections(self, fmap_channels: Iterable[int]):
        return nn.ModuleList([nn.Sequential(nn.SiLU(), nn.Linear(self.time_embedding, ch)) for ch in fmap_channels])

    def make_attention_layers(self, fmap_channels: Iterable[int]):
        fmap_channels = list(fmap_channels)
        return nn.ModuleList([ImageSelfAttention(ch, self.n_heads) if i >= len(fmap_channels) - 2 else nn.Identity()
                              for i, ch in enumerate(fmap_channels)])


class DecoderBlock(_NativeOnly):
    """upsample -> conv_up -> norm -> conv -> norm -> +skip -> +time -> act -> [attention]
    (reference score_unet.py:409-627)."""

    def __init__(self, input_channels: int, output_channels: int, time_embedding: int, upsample_scale: int = 2,
                 activation: type = nn.ReLU, compute_attn: bool = True, n_heads: int = 4, device=None, *,
                 use_resize_conv: bool = True, norm: str = "instance", gn_groups: int = 8):
        super().__init__()
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.input_channels, self.output_channels = input_channels, output_channels
        self.upsample_scale, self.time_embedding = upsample_scale, time_embedding
        self.compute_attn, self.n_heads = compute_attn, n_heads
        self.use_resize_conv, self.norm_kind, self.gn_groups = use_resize_conv, norm, gn_groups
        if int(upsample_scale) != upsample_scale or not 1 <= upsample_scale <= 16:
            raise NotImplementedError("upsample_scale must be an integer in 1..16")
        if use_resize_conv:
            self.upsample = nn.Upsample(scale_factor=upsample_scale, mode="bilinear", align_corners=False)
            self.conv_up = nn.Conv2d(input_channels, input_channels, kernel_size=3, padding=1, bias=True)
        else:
            # ablation path of the reference (score_unet.py:470-475): ConvTranspose2d(k=2, s=2), run natively as one 1x1
            # implicit GEMM to 4C phase-major channels + a depth->space permutation
            self.transpose = nn.ConvTranspose2d(input_channels, input_channels, kernel_size=upsample_scale,
                                                stride=upsample_scale)

        def make_norm(c):
            if self.norm_kind == "group":
                return nn.GroupNorm(num_groups=max(1, min(gn_groups, c)), num_channels=c)
            return nn.InstanceNorm2d(c)
        self.norm1 = make_norm(input_channels)
        self.conv = nn.Conv2d(input_channels, output_channels, kernel_size=3, padding=1)
        self.norm2 = make_norm(output_channels)
        self.activation = activation()
        self.sinusoidal_embedding = SinusoidalEmbedding(time_embedding)
        self.time_projection_layer = nn.Sequential(nn.SiLU(), nn.Linear(time_embedding, output_channels))
        self.attention = ImageSelfAttention(output_channels, n_heads) if compute_attn else nn.Identity()

    def forward(self, fmap, prev_fmap=None, t=None):
        """upsample -> conv_up -> norm1 -> conv -> norm2 -> +prev_fmap -> +time -> activation -> [attention]; `t` is the time
        vector [B] (reference score_unet.py:559-627)"""
        from .train_graph import decoder_block_call
        return decoder_block_call(self, fmap, prev_fmap, t)


class Decoder(_NativeOnly):
    """Four DecoderBlocks (attention on the first two) and a norm-free, activation-free final block
    (reference score_unet.py:662-789)."""

    def __init__(self, last_fmap_channels: int, output_channels: int, time_embedding: int, first_fmap_channels: int = 64,
                 n_heads: int = 4, device=None, *, use_resize_conv: bool = True, norm: str = "instance",
                 gn_groups: int = 8, activation: type = nn.ReLU):
        super().__init__()
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.last_fmap_channels, self.output_channels = last_fmap_channels, output_channels
        self.time_embedding, self.first_fmap_channels, self.n_heads = time_embedding, first_fmap_channels, n_heads
        self.use_resize_conv, self.norm, self.gn_groups, self.activation = use_resize_conv, norm, gn_groups, activation
        self.residual_layers = self.make_layers()
        self.final_layer = DecoderBlock(self.residual_layers[-1].input_channels, output_channels,
                                        time_embedding=time_embedding, activation=nn.Identity, compute_attn=False,
                                        n_heads=n_heads, device=self.device, use_resize_conv=use_resize_conv, norm=norm,
                                        gn_groups=gn_groups)
        self.final_layer.norm1 = nn.Identity()            # reference :726-730
        self.final_layer.norm2 = nn.Identity()
        self.final_layer.activation = nn.Identity()

    def forward(self, *fmaps, t=None):
        """fmaps = (fmap1 .. fmap5) in encoder order -> final_layer output [B,1,H,W], before the division by sigma(t)
        (reference score_unet.py:733-758)"""
        from .train_graph import decoder_call
        return decoder_call(self, *fmaps, t=t)

    def make_layers(self, n: int = 4):
        layers = []
        for i in range(n):
            in_ch = self.last_fmap_channels if i == 0 else layers[i - 1].output_channels
            out_ch = in_ch // 2 if i != (n - 1) else self.first_fmap_channels
            layers.append(DecoderBlock(in_ch, out_ch, time_embedding=self.time_embedding, compute_attn=(i < 2),
                                       n_heads=self.n_heads, device=self.device, use_resize_conv=self.use_resize_conv,
                                       norm=self.norm, gn_groups=self.gn_groups, activation=self.activation))
        return nn.ModuleList(layers)


_ACT_CODE = {nn.ReLU: N.RELU, nn.SiLU: N.SILU, nn.GELU: N.GELU, nn.Identity: N.NONE}


class _Engine:
    """One native model handle for a given split of the conditioning channels."""

    def __init__(self, net: "ScoreNet", n_lsm: int, n_topo: int, n_cond: int):
        enc, dec = net.encoder, net.decoder
        if 1 + n_lsm + n_topo + n_cond != enc.input_channels:
            raise ValueError(f"input channel mismatch: x(1)+lsm({n_lsm})+topo({n_topo})+cond_img({n_cond}) != "
                             f"encoder.conv1 in_channels ({enc.input_channels})")
        act = _ACT_CODE.get(dec.activation)
        if act is None:
            raise NotImplementedError(f"decoder activation {dec.activation} not implemented natively")
        odd = [i for i, b in enumerate(list(dec.residual_layers) + [dec.final_layer]) if getattr(b, "upsample_scale", 2) != 2]
        if odd:
            raise NotImplementedError(f"the whole-network engine runs the reference Decoder's x2 blocks; blocks {odd} have another "
                                      f"upsample_scale (such blocks run when called on their own)")
        cfg = N.ModelConfig(C.sizeof(N.ModelConfig), n_lsm, n_topo, n_cond, enc.time_embedding, (C.c_int * 4)(*enc.block_layers), enc.n_heads,
                            enc.num_classes or 0, dec.last_fmap_channels,
                            N.NORM_GROUP if dec.norm == "group" else N.NORM_INSTANCE, dec.gn_groups, act,
                            float(getattr(net, "sigma", 25.0)), 0 if dec.use_resize_conv else 1)
        self.lib = N.lib()
        h = C.c_void_p()
        N.check(self.lib.sbgm_model_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.version = None
        self._fast = None
        self._mods = None
        self.names = [self.lib.sbgm_model_param_name(self.h, i).decode()
                      for i in range(self.lib.sbgm_model_num_params(self.h))]

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.sbgm_model_destroy(self.h)
        except Exception:
            pass

    def _fast_version(self, net):
        """(identity, address, version) of every parameter / buffer over a cached module list: 0.1 ms instead of the 0.75 ms a
        state_dict() walk costs, paid by every forward / sampler call.  A changed module tree shows up as a mismatch of the full
        check below, which rebuilds the list."""
        mods = self._mods
        if mods is None:
            mods = self._mods = list(net.modules())
        out = [N.generation()]
        for m in mods:
            for c in m._modules.values():                # a replaced sub-module (net.decoder = ..., a swapped attention block)
                out.append(id(c))                        # changes the key, so the full check below rebuilds the list
            for v in m._parameters.values():
                if v is not None:
                    out.append((id(v), v.data_ptr(), v._version))
            for v in m._buffers.values():
                if v is not None:
                    out.append((id(v), v.data_ptr(), v._version))
        return tuple(out)

    def upload(self, net: "ScoreNet"):
        fast = self._fast_version(net)
        if fast == self._fast:
            return
        self._mods = None
        sd = net.state_dict(keep_vars=True)
        ver = (N.generation(),) + tuple((k, v.data_ptr(), v._version) for k, v in sd.items())
        if ver == self.version:
            self._fast = self._fast_version(net)
            return
        missing = [k for k in self.names if k not in sd]
        extra = [k for k in sd if k not in self.names]
        if missing or extra:
            raise N.NativeError(f"state_dict / engine key mismatch: missing {missing[:4]}, unexpected {extra[:4]}")
        st = N.stream()
        keep = []
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            N.require_device(v)
            t = N.f32c(v.detach())
            keep.append(t)
            N.check(self.lib.sbgm_model_set_param(self.h, k.encode(), t.data_ptr(), t.numel(), st))
        torch.cuda.current_stream().synchronize()       # `keep` may hold temporaries
        N.check(self.lib.sbgm_model_check_complete(self.h))
        self.version = ver
        self._fast = self._fast_version(net)

    def download_bn_stats(self, net: "ScoreNet", n_forwards: int = 1):
        """train-mode forwards update the engine's running statistics; mirror them into the module buffers
        (`num_batches_tracked` advances by the number of network evaluations that ran, as nn.BatchNorm2d counts them)"""
        st = N.stream()
        with torch.no_grad():
            for k, v in net.state_dict(keep_vars=True).items():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    N.check(self.lib.sbgm_model_get_param(self.h, k.encode(), v.data_ptr(), v.numel(), st))
                elif k.endswith("num_batches_tracked"):
                    v += n_forwards
        sd = net.state_dict(keep_vars=True)
        self.version = (N.generation(),) + tuple((k, v.data_ptr(), v._version) for k, v in sd.items())
        self._mods = None
        self._fast = self._fast_version(net)


class ScoreNet(nn.Module):
    """encoder -> decoder -> divide by sigma(t), evaluated by the native engine (reference score_unet.py:792-879)."""

    def __init__(self, marginal_prob_std, encoder: nn.Module, decoder: nn.Module, device=None,
                 debug_pre_sigma_div: bool = True):
        super().__init__()
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.marginal_prob_std = marginal_prob_std
        self.encoder, self.decoder = encoder, decoder
        self.debug_pre_sigma_div = debug_pre_sigma_div
        self.sigma = _sigma_of(marginal_prob_std)
        self._engines = {}
        self.to(self.device)

    # -- engine plumbing -----------------------------------------------------------------------------------
    def _engine(self, lsm_cond, topo_cond, cond_img) -> _Engine:
        key = tuple(0 if c is None else int(c.shape[1]) for c in (lsm_cond, topo_cond, cond_img))
        eng = self._engines.get(key)
        if eng is None:
            eng = self._engines[key] = _Engine(self, *key)
        eng.upload(self)
        return eng

    def _prep(self, x, t, y, cond_img, lsm_cond, topo_cond):
        N.require_device(x)
        dev = x.device
        for name, c in (("lsm_cond", lsm_cond), ("topo_cond", topo_cond)):
            if c is not None and c.shape[0] != x.shape[0]:
                raise ValueError(f"Batch mismatch: x= {x.shape[0]}, {name}={c.shape[0]}.")      # reference :275,:280
        x = N.f32c(x.to(dev))
        t = N.f32c(t.to(dev).view(-1))
        if t.numel() != x.shape[0]:
            raise ValueError(f"Batch mismatch: x= {x.shape[0]}, t={t.numel()}.")
        y = None if y is None else y.to(dev).long().contiguous()
        conds = [None if c is None else N.f32c(c.to(dev)) for c in (cond_img, lsm_cond, topo_cond)]
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"x must be [B,1,H,W], got {tuple(x.shape)}")
        return (x, t, y, *conds)

    def forward(self, x: torch.Tensor, t: torch.Tensor, y: Optional[torch.Tensor] = None,
                cond_img: Optional[torch.Tensor] = None, lsm_cond: Optional[torch.Tensor] = None,
                topo_cond: Optional[torch.Tensor] = None, *, _fmaps: Optional[list] = None):
        x, t, y, cond_img, lsm_cond, topo_cond = self._prep(x, t, y, cond_img, lsm_cond, topo_cond)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: op-by-op graph of native kernels with hand-written backward passes (train_graph.py)
            # (eval mode too: BatchNorm then uses its running statistics and stays differentiable, as in the reference)
            if _fmaps is not None:
                raise ValueError("_fmaps is an inference-only debugging hook")
            from .train_graph import forward_train
            if 1 + sum(0 if c is None else c.shape[1] for c in (lsm_cond, topo_cond, cond_img)) != self.encoder.input_channels:
                raise ValueError("input channel mismatch between the conditioning tensors and encoder.conv1")
            return forward_train(self, x, t, y, cond_img, lsm_cond, topo_cond)
        eng = self._engine(lsm_cond, topo_cond, cond_img)
        B, _, H, W = x.shape
        out = torch.empty_like(x)
        fm_ptrs = None
        if _fmaps is not None:
            hs = [(H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
            _fmaps[:] = [torch.empty(B, h, w, c, device=x.device) for (h, w), c in zip(hs, FMAP_CHANNELS)]
            fm_ptrs = (C.c_void_p * 5)(*[f.data_ptr() for f in _fmaps])
        train = self.training
        if train and any(isinstance(m, ImageSelfAttention) and m.dropout > 0 and m.training for m in self.modules()):
            raise NotImplementedError("train-mode attention dropout (p > 0) runs on the autograd path (grad enabled); the whole-network "
                                      "engine behind a no-grad train-mode evaluation / sampler has no dropout")
        N.check(eng.lib.sbgm_model_forward(eng.h, x.data_ptr(), t.data_ptr(), N.ptr(y), N.ptr(cond_img), N.ptr(lsm_cond),
                                           N.ptr(topo_cond), out.data_ptr(), fm_ptrs, B, H, W, int(train), N.stream()))
        if train:
            eng.download_bn_stats(self)
        if getattr(self, "debug_pre_sigma_div", False):
            with torch.no_grad():      # reference :866-872 (logging only): undo the division for the statistic
                pre = out * self.marginal_prob_std(t).view(-1, 1, 1, 1)
                s = self.marginal_prob_std(t)
                logger.info(f"[pre-σ-div] mean = {float(pre.mean()):.4g}, std = {float(pre.std()):.4g}, "
                            f"σ ∈ [{s.min():.4g}, {s.max():.4g}]")
        return out

    def autotune(self, batch: int, height: int, width: int, cond_channels=(0, 0, 1), cache: str = None):
        """time the conv tile candidates for this problem size once (optional).  `cache`: path of a tile-table file;
        when it exists it is loaded instead of tuning, otherwise the tuned table is written there."""
        import os
        shapes = [None if c == 0 else torch.empty(1, c, 1, 1) for c in cond_channels]
        eng = self._engine(*shapes)
        if cache and os.path.exists(cache):
            N.check(eng.lib.sbgm_model_tune_load(eng.h, os.fsencode(cache)))
            return
        N.check(eng.lib.sbgm_model_autotune(eng.h, batch, height, width, N.stream()))
        if cache:
            N.check(eng.lib.sbgm_model_tune_save(eng.h, os.fsencode(cache)))


def _sigma_of(fn) -> float:
    kw = getattr(fn, "keywords", None) or {}
    return float(kw.get("sigma", 25.0))


# --------------------------------------------------------------------------------------------------------------
# VE-SDE schedule (reference score_unet.py:881-934): [B]-sized host-orchestrated math, plain tensor ops
# --------------------------------------------------------------------------------------------------------------
_SIGMA_CACHE = {}


def _sigma_tensor(sigma: float, device) -> torch.Tensor:
    """`torch.tensor(sigma, device=...)` of the reference (:893), built once per (value, device): the host->device copy
    behind torch.tensor is not allowed while a stream is being captured into a graph"""
    key = (float(sigma), str(device))
    s = _SIGMA_CACHE.get(key)
    if s is None:
        s = _SIGMA_CACHE[key] = torch.tensor(float(sigma), dtype=torch.float32, device=device)
    return s


def marginal_prob_std(t: torch.Tensor, sigma: float, eps: float = 1e-5) -> torch.Tensor:
    t = t.to(dtype=torch.float32)
    s = _sigma_tensor(sigma, t.device)
    return torch.clamp(torch.sqrt((torch.exp((2.0 * t) * torch.log(s)) - 1.0) / (2.0 * torch.log(s))), min=eps)


def diffusion_coeff(t, sigma, device=None):
    return (sigma ** t).to(t.device)


sigma = 25.0
marginal_prob_std_fn = functools.partial(marginal_prob_std, sigma=sigma)
diffusion_coeff_fn = functools.partial(diffusion_coeff, sigma=sigma)


class _DSMLossFn(torch.autograd.Function):
    """mean_b sum_chw w * (score * std_b + z)^2 (reference score_unet.py:974-984) — csrc/dsm_loss.hip"""

    @staticmethod
    def forward(ctx, score, z, std, sdf, rng_state):
        B, per = score.shape[0], score[0].numel()
        lib = N.lib()
        loss = torch.empty((), device=score.device)
        partial = torch.empty(B * lib.sbgm_dsm_loss_blocks(per), dtype=torch.float64, device=score.device)
        N.check(lib.sbgm_dsm_loss_fwd(score.data_ptr(), z.data_ptr(), std.data_ptr(), N.ptr(sdf), partial.data_ptr(), loss.data_ptr(),
                                      N.ptr(rng_state), B, per, N.stream()))
        ctx.save_for_backward(score, z, std, sdf)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        score, z, std, sdf = ctx.saved_tensors
        B, per = score.shape[0], score[0].numel()
        dscore = torch.empty_like(score)
        dloss = N.f32c(dloss)
        N.check(N.lib().sbgm_dsm_loss_bwd(score.data_ptr(), z.data_ptr(), std.data_ptr(), N.ptr(sdf), dloss.data_ptr(),
                                          dscore.data_ptr(), B, per, N.stream()))
        return dscore, None, None, None, None


_LOSS_RNG = {}       # device -> int64[2] (Philox seed, offset) used by captured (hipGraph) steps; advanced on the device


def _loss_rng_state(dev):
    st = _LOSS_RNG.get(dev)
    if st is None:        # derived from torch's seed WITHOUT drawing from the generator (eager calls draw their own seed per call)
        seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x2545F4914F6CDD1D) & 0x7FFFFFFFFFFFFFFF
        st = _LOSS_RNG[dev] = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
    return st


def loss_fn(model, x, marginal_prob_std, t_eps=1e-3, device=None, y=None, cond_img=None, lsm_cond=None,
            topo_cond=None, sdf_cond=None, *, noise=None):
    """Denoising score-matching loss (reference score_unet.py:936-985), same argument list, same batch-size checks, same order
    of the two random draws (`t ~ U` then `z ~ N(0,1)`).  The perturbation x + sigma(t) z, the weighted squared-error reduction
    and its backward are HIP kernels (csrc/dsm_loss.hip); with grad enabled the network runs through
    train_graph.forward_train (native forward + backward kernels).

    Noise: drawn inside the perturbation kernel (Philox) from a seed taken from torch's CPU generator per call, so
    `torch.manual_seed` makes a run repeatable; while a hipGraph is being captured the seed / offset pair lives on the device
    and the loss kernel advances it, so every replay perturbs with fresh noise.  Keyword-only `noise=(t, z)` (not in the
    reference) injects the draws for parity runs.  `marginal_prob_std`: this module's VE schedule (evaluated inside the kernel) or
    any callable [B] -> [B] (evaluated on the drawn t with torch, between two launches of the perturbation kernel)."""
    N.require_device(x)
    for name, arr in (("cond_img", cond_img), ("lsm_cond", lsm_cond), ("topo_cond", topo_cond), ("y", y)):
        if arr is not None and arr.shape[0] != x.shape[0]:
            raise ValueError(f"Batch size mismatch: x={x.shape[0]}, {name}={arr.shape[0]}")
    # the module's own VE schedule is evaluated inside the perturbation kernel; any other callable (the reference accepts an
    # arbitrary marginal_prob_std, :936) is evaluated on the drawn t between two launches of the same kernel
    own_ve = getattr(marginal_prob_std, "func", marginal_prob_std) is globals()["marginal_prob_std"]
    sigma = _sigma_of(marginal_prob_std) if own_ve else 0.0
    xc = N.f32c(x)
    B, per = xc.shape[0], xc[0].numel()
    dev = xc.device
    lib = N.lib()
    t_in = z_in = None
    if noise is not None:
        t_in, z_in = N.f32c(noise[0].to(dev).view(-1)), N.f32c(noise[1].to(dev))
        if t_in.numel() != B or z_in.shape != xc.shape:
            raise ValueError(f"noise=(t, z) must have shapes [{B}] and {tuple(xc.shape)}")
    capturing = torch.cuda.is_current_stream_capturing()
    if noise is None and not capturing:
        _loss_rng_state(dev)                      # created eagerly (warm-up steps): a host->device copy is illegal during capture
    if noise is None and capturing and dev not in _LOSS_RNG:
        raise RuntimeError("loss_fn: run one eager (warm-up) step before capturing it into a graph")
    rng = _loss_rng_state(dev) if (noise is None and capturing) else None
    seed = 0
    if noise is None and not capturing:
        from .score_sampling import _fresh_seed
        seed = _fresh_seed()
    random_t, std = torch.empty(B, device=dev), torch.empty(B, device=dev)
    perturbed_x = torch.empty_like(xc)
    z = z_in if z_in is not None else torch.empty_like(xc)
    if not own_ve:
        N.check(lib.sbgm_dsm_perturb(xc.data_ptr(), None, N.ptr(t_in), N.ptr(rng), seed, float(t_eps), 1.0,
                                     perturbed_x.data_ptr(), None, random_t.data_ptr(), std.data_ptr(), B, 0, N.stream()))   # t only
        user_std = marginal_prob_std(random_t)
        std.copy_(torch.as_tensor(user_std, dtype=torch.float32, device=dev).reshape(-1).expand(B))
        t_in = random_t
    N.check(lib.sbgm_dsm_perturb(xc.data_ptr(), N.ptr(z_in), N.ptr(t_in), N.ptr(rng), seed, float(t_eps), sigma,
                                 perturbed_x.data_ptr(), z.data_ptr(), random_t.data_ptr(), std.data_ptr(), B, per, N.stream()))
    score = model(perturbed_x, random_t, y=y, cond_img=cond_img, lsm_cond=lsm_cond, topo_cond=topo_cond)
    N.require_device(score)
    sdf = None
    if sdf_cond is not None:
        sdf = N.f32c(sdf_cond.to(dev))
        if sdf.shape != xc.shape:
            sdf = sdf.expand_as(xc).contiguous()
    return _DSMLossFn.apply(N.f32c(score), z, std, sdf, rng)
