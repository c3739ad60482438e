"""ORACLE — test infrastructure only.  Never imported by the product path (sbgm_danra_amd/*).

CPU restatement, in plain PyTorch fp32, of the hot path of TheaQG/SBGM_DANRA: the score UNet
(sbgm/score_unet.py) and the reverse-SDE samplers (sbgm/score_sampling.py).  Each class/function cites
the reference lines it follows.  Parity status: PINNED — `oracle/make_goldens.py` instantiates the
reference's own classes in this container (with the build-owned torchvision ResNet/BasicBlock
restatement under oracle/_tv_standin, since torchvision is absent) and checks this file against them
bit-for-bit / to <=1e-6; the captured vectors live in tests/golden/.  The torchvision part
(BasicBlock/ResNet-18 layout) is a restatement of that library's published algorithm; the reference has
no tests that pin it, so at that boundary parity is pinned only by those goldens.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

State-dict keys and shapes are identical to the reference's ScoreNet so the same checkpoint loads in
the reference, in this oracle and in the product module.
"""
from __future__ import annotations

import functools
import math
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

SIGMA = 25.0  # reference score_unet.py:932


# ----------------------------------------------------------------------------------------------
# VE-SDE schedule (reference score_unet.py:881-897, 916-934)
# ----------------------------------------------------------------------------------------------
def marginal_prob_std(t: torch.Tensor, sigma: float = SIGMA, eps: float = 1e-5) -> torch.Tensor:
    """sqrt((sigma^(2t) - 1) / (2 ln sigma)), floored at eps — score_unet.py:881-897."""
    t = t.to(torch.float32)
    ls = torch.log(torch.tensor(sigma, dtype=t.dtype, device=t.device))
    return torch.clamp(torch.sqrt((torch.exp((2.0 * t) * ls) - 1.0) / (2.0 * ls)), min=eps)


def diffusion_coeff(t, sigma: float = SIGMA, device=None):
    """g(t) = sigma^t — score_unet.py:916-930."""
    return (sigma ** t).to(t.device)


marginal_prob_std_fn = functools.partial(marginal_prob_std, sigma=SIGMA)
diffusion_coeff_fn = functools.partial(diffusion_coeff, sigma=SIGMA)


# ----------------------------------------------------------------------------------------------
# Building blocks
# ----------------------------------------------------------------------------------------------
class SinusoidalEmbedding(nn.Module):
    """Gaussian-Fourier features [sin(2 pi t W), cos(2 pi t W)] — score_unet.py:24-45."""

    def __init__(self, embed_dim: int, scale: float = 30.0):
        super().__init__()
        if embed_dim % 2:
            raise ValueError(f"Embedding dimension must be even, got {embed_dim}.")
        self.register_buffer("W", torch.randn(embed_dim // 2) * scale, persistent=True)

    def forward(self, t):
        t = t.view(-1).to(self.W.dtype)
        p = t[:, None] * self.W[None, :] * (2.0 * torch.pi)   # same association as :44
        return torch.cat([p.sin(), p.cos()], dim=-1)


class ImageSelfAttention(nn.Module):
    """Pre-LN residual MHA + FF over H*W tokens — score_unet.py:112-148."""

    def __init__(self, input_channels: int, n_heads: int, dropout: float = 0.0):
        super().__init__()
        if input_channels % n_heads:
            raise ValueError(f"Number of input channels ({input_channels}) must be divisible by "
                             f"number of heads ({n_heads}).")
        self.input_channels, self.n_heads = input_channels, n_heads
        self.mha = nn.MultiheadAttention(input_channels, n_heads, dropout=dropout, batch_first=True)
        self.ln1 = nn.LayerNorm(input_channels)
        self.ln2 = nn.LayerNorm(input_channels)
        self.ff = nn.Sequential(nn.Linear(input_channels, input_channels), nn.GELU(),
                                nn.Linear(input_channels, input_channels))

    def forward(self, x):
        n, c, hh, ww = x.shape
        tok = x.reshape(n, c, hh * ww).transpose(1, 2)
        q = self.ln1(tok)
        h = tok + self.mha(q, q, q)[0]
        out = h + self.ff(self.ln2(h))
        return out.transpose(1, 2).reshape(n, c, hh, ww)


class BasicBlock(nn.Module):
    """torchvision ResNet basic block (published algorithm; used at score_unet.py:161,188)."""
    expansion = 1

    def __init__(self, cin, cout, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        return self.relu(self.bn2(self.conv2(y)) + skip)


def _time_proj(dim, ch):
    return nn.Sequential(nn.SiLU(), nn.Linear(dim, ch))


FMAP_CHANNELS = (64, 64, 128, 256, 512)   # score_unet.py:198


class Encoder(nn.Module):
    """ResNet-18-style encoder with two 8x8/s2 stem convs, time-bias adds and attention on the two
    deepest maps — score_unet.py:151-404 (torchvision ResNet.__init__/_make_layer for the stages)."""

    def __init__(self, input_channels: int, time_embedding: int, block=BasicBlock,
                 block_layers: Sequence[int] = (2, 2, 2, 2), n_heads: int = 4,
                 num_classes: Optional[int] = None, cond_on_img=False, cond_img_dim=None, device=None):
        super().__init__()
        self.input_channels = input_channels + 1      # +1: the noised HR field (:182)
        self.time_embedding, self.n_heads, self.num_classes = time_embedding, n_heads, num_classes
        self.block_layers = list(block_layers)
        # registration order mirrors the reference so state_dict ordering matches too
        self.conv1 = nn.Conv2d(self.input_channels, 64, 8, 2, 3, bias=False)          # :206
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        cin = 64
        for li, (w, nb) in enumerate(zip(FMAP_CHANNELS[1:], self.block_layers), start=1):
            stride = 1 if li == 1 else 2
            blocks = []
            for bi in range(nb):
                ds = None
                if bi == 0 and (stride != 1 or cin != w):
                    ds = nn.Sequential(nn.Conv2d(cin, w, 1, stride, bias=False), nn.BatchNorm2d(w))
                blocks.append(BasicBlock(cin, w, stride if bi == 0 else 1, ds))
                cin = w
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.sinusoidal_embedding = SinusoidalEmbedding(time_embedding)
        self.time_projection_layers = nn.ModuleList(_time_proj(time_embedding, c) for c in FMAP_CHANNELS)
        self.attention_layers = nn.ModuleList(
            ImageSelfAttention(c, n_heads) if i >= len(FMAP_CHANNELS) - 2 else nn.Identity()
            for i, c in enumerate(FMAP_CHANNELS))                                     # :394-397
        self.conv2 = nn.Conv2d(64, 64, 8, 2, 3, bias=False)                           # :214
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes + 1, time_embedding)
            with torch.no_grad():
                self.label_emb.weight[0].zero_()                                      # :224-226

    def forward(self, x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None):
        for name, c in (("lsm_cond", lsm_cond), ("topo_cond", topo_cond)):            # :273-282
            if c is not None:
                if c.shape[0] != x.shape[0]:
                    raise ValueError(f"Batch mismatch: x= {x.shape[0]}, {name}={c.shape[0]}.")
                x = torch.cat([x, c.to(x.device)], dim=1)
        if cond_img is not None:
            x = torch.cat([x, cond_img.to(x.device)], dim=1)                          # :291
        emb = self.sinusoidal_embedding(t.to(x.device).float().view(-1))              # :301-303
        if y is not None:
            emb = emb + self.label_emb(y.to(x.device))                                # :308
        tp = [proj(emb)[:, :, None, None] for proj in self.time_projection_layers]
        f1 = self.attention_layers[0](self.conv1(x) + tp[0])                          # :312-318
        h = self.relu(self.bn1(self.conv2(f1)))                                       # :321-325
        f2 = self.attention_layers[1](self.layer1(h) + tp[1])
        f3 = self.attention_layers[2](self.layer2(f2) + tp[2])
        f4 = self.attention_layers[3](self.layer3(f3) + tp[3])
        f5 = self.attention_layers[4](self.layer4(f4) + tp[4])
        return f1, f2, f3, f4, f5


class DecoderBlock(nn.Module):
    """upsample -> conv_up -> norm -> conv -> norm -> +skip -> +time -> act -> [attention]
    — score_unet.py:409-627."""

    def __init__(self, input_channels, output_channels, time_embedding, upsample_scale=2,
                 activation=nn.ReLU, compute_attn=True, n_heads=4, device=None, *,
                 use_resize_conv=True, norm="instance", gn_groups=8):
        super().__init__()
        self.input_channels, self.output_channels = input_channels, output_channels
        self.time_embedding, self.use_resize_conv = time_embedding, use_resize_conv
        if use_resize_conv:
            self.upsample = nn.Upsample(scale_factor=upsample_scale, mode="bilinear", align_corners=False)
            self.conv_up = nn.Conv2d(input_channels, input_channels, 3, padding=1, bias=True)
        else:
            self.transpose = nn.ConvTranspose2d(input_channels, input_channels, upsample_scale, upsample_scale)

        def mk(c):
            return (nn.GroupNorm(max(1, min(gn_groups, c)), c) if norm == "group" else nn.InstanceNorm2d(c))
        self.norm1 = mk(input_channels)
        self.conv = nn.Conv2d(input_channels, output_channels, 3, padding=1)
        self.norm2 = mk(output_channels)
        self.activation = activation()
        self.sinusoidal_embedding = SinusoidalEmbedding(time_embedding)
        self.time_projection_layer = _time_proj(time_embedding, output_channels)
        self.attention = ImageSelfAttention(output_channels, n_heads) if compute_attn else nn.Identity()

    def forward(self, fmap, prev_fmap=None, t=None):
        h = self.conv_up(self.upsample(fmap)) if self.use_resize_conv else self.transpose(fmap)
        h = self.norm2(self.conv(self.norm1(h)))
        if prev_fmap is not None and torch.is_tensor(prev_fmap):
            if prev_fmap.shape != h.shape:
                raise AssertionError(f"prev_fmap shape {prev_fmap.shape} must match output shape {tuple(h.shape)}")
            h = h + prev_fmap
        if t is not None:
            emb = self.sinusoidal_embedding(t.view(-1)) if (t.dim() == 1 or t.shape[-1] != self.time_embedding) else t
            h = h + self.time_projection_layer(emb)[:, :, None, None]
        return self.attention(self.activation(h))


class Decoder(nn.Module):
    """Four DecoderBlocks (attention on the first two) + a norm-free, activation-free final block
    called without skip or time — score_unet.py:662-789."""

    def __init__(self, last_fmap_channels, output_channels, time_embedding, first_fmap_channels=64,
                 n_heads=4, device=None, *, use_resize_conv=True, norm="instance", gn_groups=8,
                 activation=nn.ReLU):
        super().__init__()
        kw = dict(time_embedding=time_embedding, n_heads=n_heads, use_resize_conv=use_resize_conv,
                  norm=norm, gn_groups=gn_groups)
        blocks, cin = [], last_fmap_channels
        for i in range(4):
            cout = cin // 2 if i != 3 else first_fmap_channels
            blocks.append(DecoderBlock(cin, cout, compute_attn=(i < 2), activation=activation, **kw))
            cin = cout
        self.residual_layers = nn.ModuleList(blocks)
        self.final_layer = DecoderBlock(blocks[-1].input_channels, output_channels, compute_attn=False,
                                        activation=nn.Identity, **kw)
        self.final_layer.norm1 = nn.Identity()
        self.final_layer.norm2 = nn.Identity()
        self.final_layer.activation = nn.Identity()

    def forward(self, *fmaps, t=None):
        assert len(fmaps) == len(self.residual_layers) + 1
        f = fmaps[::-1]
        h = f[0]
        for i, blk in enumerate(self.residual_layers):
            h = blk(h, f[i + 1], t)
        return self.final_layer(h)


class ScoreNet(nn.Module):
    """encoder -> decoder -> divide by sigma(t) — score_unet.py:792-879 (debug statistic omitted: it
    only logs)."""

    def __init__(self, marginal_prob_std, encoder, decoder, device=None, debug_pre_sigma_div=False):
        super().__init__()
        self.marginal_prob_std, self.encoder, self.decoder = marginal_prob_std, encoder, decoder

    def forward(self, x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None):
        t = t.to(x.device).float()
        if y is not None:
            y = y.to(x.device).long()
        out = self.decoder(*self.encoder(x, t, y=y, cond_img=cond_img, lsm_cond=lsm_cond,
                                         topo_cond=topo_cond), t=t)
        return out / self.marginal_prob_std(t).view(-1, 1, 1, 1)


def build_scorenet(in_cond_channels=1, time_embedding=256, block_layers=(2, 2, 2, 2), n_heads=4,
                   num_classes=None, last_fmap_channels=512, norm="group", gn_groups=8,
                   activation=nn.SiLU, use_resize_conv=True):
    """What training_utils.get_model builds (reference training_utils.py:645-666)."""
    enc = Encoder(in_cond_channels, time_embedding, block_layers=block_layers, n_heads=n_heads,
                  num_classes=num_classes)
    dec = Decoder(last_fmap_channels, 1, time_embedding, n_heads=n_heads, use_resize_conv=use_resize_conv,
                  norm=norm, gn_groups=gn_groups, activation=activation)
    return ScoreNet(marginal_prob_std_fn, enc, dec)


# ----------------------------------------------------------------------------------------------
# Loss (reference score_unet.py:936-985).  `noise=(t, z)` lets tests inject the random draws.
# ----------------------------------------------------------------------------------------------
def loss_fn(model, x, marginal_prob_std, t_eps=1e-3, device=None, y=None, cond_img=None, lsm_cond=None,
            topo_cond=None, sdf_cond=None, noise=None):
    if noise is None:
        t = torch.rand(x.shape[0], device=x.device) * (1.0 - t_eps) + t_eps
        z = torch.randn_like(x)
    else:
        t, z = noise
    std = marginal_prob_std(t)
    for name, a in (("cond_img", cond_img), ("lsm_cond", lsm_cond), ("topo_cond", topo_cond), ("y", y)):
        if a is not None and a.shape[0] != x.shape[0]:
            raise ValueError(f"Batch size mismatch: x={x.shape[0]}, {name}={a.shape[0]}")
    score = model(x + std[:, None, None, None] * z, t, y=y, cond_img=cond_img, lsm_cond=lsm_cond,
                  topo_cond=topo_cond)
    w = torch.sigmoid(sdf_cond) * 0.5 + 0.5 if sdf_cond is not None else torch.ones_like(x)
    return torch.mean(torch.sum(w * (score * std[:, None, None, None] + z) ** 2, dim=(1, 2, 3)))


# ----------------------------------------------------------------------------------------------
# Samplers (reference score_sampling.py).  `noise` is an optional iterator of pre-drawn N(0,1)
# tensors consumed in the reference's RNG order (init, then per step) so device runs can be compared.
# ----------------------------------------------------------------------------------------------
def guided_score_fn(score_model, x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None,
                    null_token: int = 0, scale: float = 2.0):
    """CFG combine (1+w) s_c - w s_u with mask-channel strip — score_sampling.py:10-56."""
    def strip(c):
        if c is None or c.shape[1] != 2:
            return c
        c = c.clone()
        c[:, 1] = 0.0
        return c
    s_c = score_model(x, t, y, cond_img, lsm_cond, topo_cond)
    s_u = score_model(x, t, None if y is None else torch.full_like(y, null_token),
                      None if cond_img is None else torch.zeros_like(cond_img), strip(lsm_cond), strip(topo_cond))
    return (1.0 + scale) * s_c - scale * s_u


def _draw(noise, like):
    return torch.randn_like(like) if noise is None else next(noise).to(like)


def _score(score_model, cfg, x, t, y, cond_img, lsm_cond, topo_cond, clamp=False):
    g = (cfg or {}).get("classifier_free_guidance", {})
    if g.get("enabled", False):
        s = g.get("guidance_scale", 2.0)
        if clamp and g.get("guidance_scale_max") is not None:
            s = min(s, g["guidance_scale_max"])
        return guided_score_fn(score_model, x, t, y, cond_img, lsm_cond, topo_cond, scale=s)
    return score_model(x, t, y, cond_img, lsm_cond, topo_cond)


def Euler_Maruyama_sampler(score_model, marginal_prob_std, diffusion_coeff, batch_size=64, num_steps=500,
                           device="cpu", eps=1e-3, img_size=64, y=None, cond_img=None, lsm_cond=None,
                           topo_cond=None, cfg=None, noise=None, init_hw=32):
    """score_sampling.py:63-127.  The reference hard-codes a 32x32 start (:94); `init_hw` keeps that
    default and lets callers ask for another size."""
    ones = torch.ones(batch_size, device=device)
    x = _draw(noise, torch.empty(batch_size, 1, init_hw, init_hw, device=device)) \
        * marginal_prob_std(ones)[:, None, None, None]
    ts = torch.linspace(1.0, eps, num_steps, device=device)
    dt = ts[0] - ts[1]
    mean_x = x
    with torch.no_grad():
        for tt in ts:
            bt = ones * tt
            g = diffusion_coeff(bt)
            s = _score(score_model, cfg, x, bt, y, cond_img, lsm_cond, topo_cond)
            mean_x = x + (g ** 2)[:, None, None, None] * s * dt
            x = mean_x + torch.sqrt(dt) * g[:, None, None, None] * _draw(noise, x)
    return mean_x


def pc_sampler(score_model, marginal_prob_std, diffusion_coeff, batch_size=64, num_steps=800, snr=0.16,
               device="cpu", eps=1e-3, img_size=64, y=None, cond_img=None, lsm_cond=None, topo_cond=None,
               cfg=None, noise=None):
    """Langevin corrector + Euler-Maruyama predictor — score_sampling.py:136-230.  Note the fp64
    np.linspace scalars multiplying fp32 tensors (:169-176) and the batch-mean gradient norm (:201)."""
    ones = torch.ones(batch_size, device=device)
    x = _draw(noise, torch.empty(batch_size, 1, img_size, img_size, device=device)) \
        * marginal_prob_std(ones)[:, None, None, None]
    ts = np.linspace(1.0, eps, num_steps)
    dt = ts[0] - ts[1]
    x_mean = x
    with torch.no_grad():
        for tt in ts:
            bt = ones * tt
            grad = _score(score_model, cfg, x, bt, y, cond_img, lsm_cond, topo_cond, clamp=True)
            gnorm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
            lstep = 2 * (snr * np.sqrt(np.prod(x.shape[1:])) / gnorm) ** 2
            x = x + lstep * grad + torch.sqrt(2 * lstep) * _draw(noise, x)
            g = diffusion_coeff(bt)
            s = _score(score_model, cfg, x, bt, y, cond_img, lsm_cond, topo_cond)
            x_mean = x + (g ** 2)[:, None, None, None] * s * dt
            x = x_mean + torch.sqrt(g ** 2 * dt)[:, None, None, None] * _draw(noise, x)
    return x_mean


def ode_sampler(score_model, marginal_prob_std, diffusion_coeff, num_steps=100, batch_size=64, atol=1e-5, rtol=1e-5,
                device="cpu", z=None, eps=1e-3, img_size=64, y=None, cond_img=None, lsm_cond=None, topo_cond=None,
                cfg=None, return_nfev=False):
    """Probability-flow ODE with scipy's RK45 — score_sampling.py:239-300.  Like the reference it starts from
    32x32 unless `z` is given (:279-283), evaluates the network on (x, t) ONLY (no conditioning reaches it,
    :290), builds t as float64 ones * t cast to fp32 (:287-288), and squares g(t) = sigma^t in fp32 before the
    float64 product with the score (:296-297).  `return_nfev` (not in the reference, which only logs it)."""
    from scipy import integrate
    t = torch.ones(batch_size, device=device)
    init_x = torch.randn(batch_size, 1, 32, 32, device=device) * marginal_prob_std(t)[:, None, None, None] if z is None else z
    shape = init_x.shape

    def score_eval_wrapper(sample, time_steps):
        sample = torch.tensor(sample, device=device, dtype=torch.float32).reshape(shape)
        time_steps = torch.tensor(time_steps, device=device, dtype=torch.float32).reshape((sample.shape[0],))
        with torch.no_grad():
            score = score_model(sample, time_steps)
        return score.cpu().numpy().reshape((-1,)).astype(np.float64)

    def ode_func(tt, x):
        time_steps = np.ones((shape[0],)) * tt
        g = diffusion_coeff(torch.tensor(tt)).cpu().numpy()
        return -0.5 * (g ** 2) * score_eval_wrapper(x, time_steps)

    res = integrate.solve_ivp(ode_func, (1.0, eps), init_x.reshape(-1).cpu().numpy(), rtol=rtol, atol=atol, method="RK45")
    x = torch.tensor(res.y[:, -1], device=device).reshape(shape)
    return (x, res.nfev) if return_nfev else x


# ----------------------------------------------------------------------------------------------
# Deterministic, framework-independent weight generator: a counter-based hash of
# (tensor name, flat index) -> uniform value, so the 76 MB state_dict never has to be committed.
# ----------------------------------------------------------------------------------------------
def _hash_uniform(name: str, n: int) -> np.ndarray:
    """splitmix64 over (fnv1a(name) + index) -> float64 uniform in [0, 1)."""
    h = np.uint64(0xCBF29CE484222325)
    with np.errstate(over="ignore"):
        for ch in name.encode():
            h = (h ^ np.uint64(ch)) * np.uint64(0x100000001B3)
        z = h + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def synth_state_dict(model: nn.Module) -> dict:
    """Fill every state_dict entry from its name alone.  Conv/linear weights: uniform in the Xavier
    range; biases small; norm gammas around 1; BN running_var in [0.5, 1.5]; Fourier W ~ 30*N(0,1)
    via Box-Muller on the same hash; label_emb row 0 stays zero (score_unet.py:226)."""
    out = {}
    for k, v in model.state_dict().items():
        n = v.numel()
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
            continue
        u = _hash_uniform(k, max(n, 1))[:n]
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "W":
            u2 = _hash_uniform(k + "#2", n)
            val = 30.0 * np.sqrt(-2.0 * np.log(1.0 - u)) * np.cos(2 * np.pi * u2)
        elif leaf == "running_var":
            val = 0.5 + u
        elif leaf == "running_mean":
            val = 0.2 * (u - 0.5)
        elif v.dim() == 1 and leaf == "weight":          # norm gamma
            val = 0.75 + 0.5 * u
        elif v.dim() == 1 or leaf in ("bias", "in_proj_bias"):
            val = 0.1 * (u - 0.5)
        else:                                            # conv / linear / embedding matrices
            if v.dim() == 4:
                rf = v.shape[2] * v.shape[3]
                fan_in, fan_out = v.shape[1] * rf, v.shape[0] * rf
            else:
                fan_out, fan_in = v.shape[0], v.shape[1]
            a = math.sqrt(6.0 / (fan_in + fan_out))
            val = (2.0 * u - 1.0) * a
        t = torch.from_numpy(np.asarray(val, dtype=np.float64).reshape(v.shape)).to(v.dtype)
        if k.endswith("label_emb.weight"):
            t[0] = 0.0
        out[k] = t
    return out
