"""Drop-in mirror of the reference's `sbgm/score_sampling.py`: same function names, positional/keyword
signatures and return values; the loops run on the device through libsbgm_hip.so.

* `score_model` is a native `ScoreNet` and classifier-free guidance is off (the only configuration the
  reference CLI reaches): the whole loop — network evaluations, Langevin / Euler-Maruyama
  updates, in-kernel Philox noise — is enqueued by one C call (`sbgm_sampler_run`), optionally as a replayed
  hipGraph of one SDE step.
* otherwise (guidance on, or an arbitrary callable as `score_model`): a Python loop drives the same fused update
  kernels (`sbgm_em_step`, `sbgm_langevin_step`, `sbgm_cfg_combine`) around `score_model` calls.

Extra keyword-only arguments (not in the reference): `noise` — pre-drawn N(0,1) tensors consumed in the
reference's RNG order (init, then per step [corrector,] predictor) for parity runs; `use_graph`; `seed`.
Deviation kept explicit: the reference's Euler-Maruyama sampler ignores `img_size` and always starts from
32x32 (score_sampling.py:94); here `img_size` is honoured (pass 32 to reproduce the reference literally).
"""
from __future__ import annotations

import ctypes as C
import logging
import math

import numpy as np
import torch

from . import _native as N
from .score_unet import ScoreNet

logger = logging.getLogger(__name__)

signal_to_noise_ratio = 0.16      # reference score_sampling.py:132
error_tolerance = 1e-5            # reference score_sampling.py:238


def _fresh_seed() -> int:
    """64-bit seed drawn from torch's default CPU generator, so torch.manual_seed() makes runs repeatable"""
    return int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFFFFFF


def _cfg_enabled(cfg) -> bool:
    return bool((cfg or {}).get("classifier_free_guidance", {}).get("enabled", False))


def guided_score_fn(score_model, x, t, y=None, cond_img=None, lsm_cond=None, topo_cond=None, null_token: int = 0,
                    scale: float = 2.0):
    """Classifier-free guidance (reference score_sampling.py:10-56): (1+w)*s_cond - w*s_uncond, where the
    unconditional branch zeroes cond_img, zeroes only the MASK channel of 2-channel geo conditions and uses the
    null class token.  The two evaluations run through `score_model`; the combine is one fused kernel."""
    def strip_mask(c):
        if c is None or c.shape[1] != 2:
            return c
        c = c.clone()
        c[:, 1, :, :] = 0.0
        return c
    s_c = score_model(x, t, y, cond_img, lsm_cond, topo_cond)
    s_u = score_model(x, t, None if y is None else torch.full_like(y, null_token),
                      None if cond_img is None else torch.zeros_like(cond_img), strip_mask(lsm_cond), strip_mask(topo_cond))
    N.require_device(s_c, s_u)
    s_c, s_u = N.f32c(s_c), N.f32c(s_u)
    out = torch.empty_like(s_c)
    N.check(N.lib().sbgm_cfg_combine(out.data_ptr(), s_c.data_ptr(), s_u.data_ptr(), float(scale), out.numel(), N.stream()))
    return out


def _score(score_model, cfg, x, t, y, cond_img, lsm_cond, topo_cond, clamp=False):
    if _cfg_enabled(cfg):
        g = cfg["classifier_free_guidance"]
        scale = g.get("guidance_scale", 2.0)
        if clamp and g.get("guidance_scale_max") is not None and scale > g["guidance_scale_max"]:
            scale = g["guidance_scale_max"]                                            # reference :184-186
        return guided_score_fn(score_model, x, t, y, cond_img, lsm_cond, topo_cond, scale=scale)
    return score_model(x, t, y, cond_img, lsm_cond, topo_cond)


def _guidance(cfg):
    """(enabled, predictor scale, corrector scale): the reference clamps only the corrector's scale to
    guidance_scale_max (score_sampling.py:184-186 vs :213)."""
    if not _cfg_enabled(cfg):
        return 0, 0.0, 0.0
    g = cfg["classifier_free_guidance"]
    scale = float(g.get("guidance_scale", 2.0))
    corr = scale
    if g.get("guidance_scale_max") is not None and corr > g["guidance_scale_max"]:
        corr = float(g["guidance_scale_max"])
    return 1, scale, corr


def _native_run(kind, score_model: ScoreNet, batch_size, num_steps, snr, eps, hw, y, cond_img, lsm_cond, topo_cond,
                noise, use_graph, seed, device, cfg=None, tile_origins=None, domain_width=0):
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    if dev.type != "cuda":
        raise N.NativeError(f"the native samplers run on a ROCm device, got device={device!r}")
    x_dummy = torch.empty(batch_size, 1, hw, hw, device=dev)
    t_dummy = torch.empty(batch_size, device=dev)
    _, _, y, cond_img, lsm_cond, topo_cond = score_model._prep(x_dummy, t_dummy, y, cond_img, lsm_cond, topo_cond)
    eng = score_model._engine(lsm_cond, topo_cond, cond_img)
    out = torch.empty(batch_size, 1, hw, hw, device=dev)
    nz = None
    if noise is not None:
        nz = noise if torch.is_tensor(noise) else torch.stack(list(noise))
        need = 1 + num_steps * (2 if kind == N.SAMPLER_PC else 1)
        if nz.shape[0] < need:
            raise ValueError(f"noise holds {nz.shape[0]} draws, the sampler consumes {need}")
        nz = N.f32c(nz.to(dev))
    a = N.SamplerArgs(kind, batch_size, hw, hw, int(num_steps), float(eps), float(snr), int(seed), int(bool(use_graph)),
                      int(score_model.training), N.ptr(y), N.ptr(cond_img), N.ptr(lsm_cond), N.ptr(topo_cond), N.ptr(nz),
                      out.data_ptr(), *_guidance(cfg), N.ptr(tile_origins), int(domain_width or 0))
    if tile_origins is not None:
        N.require_device(tile_origins)
        if tile_origins.dtype != torch.int32 or tuple(tile_origins.shape) != (batch_size, 2) or not tile_origins.is_contiguous():
            raise ValueError("tile_origins must be a contiguous int32 [batch, 2] device tensor of (y0, x0)")
        if noise is not None:
            raise ValueError("tile_origins keys the in-kernel noise; it cannot be combined with injected noise")
    N.check(eng.lib.sbgm_sampler_run(eng.h, C.byref(a), N.stream()))
    if score_model.training:
        eng.download_bn_stats(score_model, n_forwards=int(num_steps) * (2 if kind == N.SAMPLER_PC else 1))
    return out


def Euler_Maruyama_sampler(score_model, marginal_prob_std, diffusion_coeff, batch_size=64, num_steps=500, device="cuda",
                           eps=1e-3, img_size=64, y=None, cond_img=None, lsm_cond=None, topo_cond=None, cfg=None, *,
                           noise=None, use_graph=True, seed=None, tile_origins=None, domain_width=0):
    """Euler-Maruyama reverse-SDE sampler (reference score_sampling.py:63-127).  Returns the last `mean_x`."""
    seed = _fresh_seed() if seed is None else seed
    if isinstance(score_model, ScoreNet) and not (_cfg_enabled(cfg) and score_model.training):
        return _native_run(N.SAMPLER_EM, score_model, batch_size, num_steps, 0.0, eps, img_size, y, cond_img, lsm_cond,
                           topo_cond, noise, use_graph, seed, device, cfg, tile_origins, domain_width)
    if tile_origins is not None:
        raise N.NativeError("domain-keyed noise (tile_origins) needs the native sampler loop (a ScoreNet in eval mode)")
    lib, st = N.lib(), N.stream
    noise = iter(noise) if noise is not None else None
    ones = torch.ones(batch_size, device=device)
    std1 = float(marginal_prob_std(ones)[0])
    x = torch.empty(batch_size, 1, img_size, img_size, device=device)
    draw = 0
    if noise is None:
        N.check(lib.sbgm_randn_scaled(x.data_ptr(), std1, seed, draw, x.numel(), st()))
    else:
        x.copy_(next(noise).to(x) * std1)
    time_steps = torch.linspace(1.0, eps, num_steps, device=device)
    step_size = float(time_steps[0] - time_steps[1])
    mean_x = torch.empty_like(x)
    with torch.no_grad():
        for ts in time_steps.tolist():
            bt = ones * ts
            g = float(diffusion_coeff(bt)[0])
            score = N.f32c(_score(score_model, cfg, x, bt, y, cond_img, lsm_cond, topo_cond))
            draw += 1
            z = None if noise is None else N.f32c(next(noise).to(x))
            N.check(lib.sbgm_em_step(x.data_ptr(), mean_x.data_ptr(), score.data_ptr(), N.ptr(z), g * g, step_size,
                                     math.sqrt(step_size) * g, seed, draw, x.numel(), st()))
    return mean_x


def pc_sampler(score_model, marginal_prob_std, diffusion_coeff, batch_size=64, num_steps=800, snr=signal_to_noise_ratio,
               device="cuda", eps=1e-3, img_size=64, y=None, cond_img=None, lsm_cond=None, topo_cond=None, cfg=None, *,
               noise=None, use_graph=True, seed=None, tile_origins=None, domain_width=0):
    """Predictor-corrector sampler: Langevin corrector with the batch-mean gradient norm, then an Euler-Maruyama
    predictor (reference score_sampling.py:136-230).  Returns the last `x_mean`."""
    seed = _fresh_seed() if seed is None else seed
    if isinstance(score_model, ScoreNet) and not (_cfg_enabled(cfg) and score_model.training):
        return _native_run(N.SAMPLER_PC, score_model, batch_size, num_steps, snr, eps, img_size, y, cond_img, lsm_cond,
                           topo_cond, noise, use_graph, seed, device, cfg, tile_origins, domain_width)
    if tile_origins is not None:
        raise N.NativeError("domain-keyed noise (tile_origins) needs the native sampler loop (a ScoreNet in eval mode)")
    lib, st = N.lib(), N.stream
    noise = iter(noise) if noise is not None else None
    ones = torch.ones(batch_size, device=device)
    std1 = float(marginal_prob_std(ones)[0])
    x = torch.empty(batch_size, 1, img_size, img_size, device=device)
    draw = 0
    if noise is None:
        N.check(lib.sbgm_randn_scaled(x.data_ptr(), std1, seed, draw, x.numel(), st()))
    else:
        x.copy_(next(noise).to(x) * std1)
    time_steps = np.linspace(1.0, eps, num_steps)
    step_size = float(time_steps[0] - time_steps[1])
    x_mean = torch.empty_like(x)
    sumsq = torch.empty(batch_size, dtype=torch.float64, device=device)
    per = x[0].numel()
    snr_nn = float(snr * np.sqrt(per))
    with torch.no_grad():
        for ts in time_steps:
            bt = ones * ts
            grad = N.f32c(_score(score_model, cfg, x, bt, y, cond_img, lsm_cond, topo_cond, clamp=True))
            draw += 1
            z = None if noise is None else N.f32c(next(noise).to(x))
            N.check(lib.sbgm_langevin_step(x.data_ptr(), grad.data_ptr(), N.ptr(z), snr_nn, sumsq.data_ptr(), seed, draw,
                                           batch_size, per, st()))
            g = float(diffusion_coeff(bt)[0])
            score = N.f32c(_score(score_model, cfg, x, bt, y, cond_img, lsm_cond, topo_cond))
            draw += 1
            z = None if noise is None else N.f32c(next(noise).to(x))
            N.check(lib.sbgm_em_step(x.data_ptr(), x_mean.data_ptr(), score.data_ptr(), N.ptr(z), g * g, step_size,
                                     math.sqrt(g * g * step_size), seed, draw, x.numel(), st()))
    return x_mean


def ode_sampler(score_model, marginal_prob_std, diffusion_coeff, num_steps=100, batch_size=64, atol=error_tolerance,
                rtol=error_tolerance, device="cuda", z=None, eps=1e-3, img_size=64, y=None, cond_img=None, lsm_cond=None,
                topo_cond=None, cfg=None, *, return_nfev=False):
    """Probability-flow ODE via scipy RK45 (reference score_sampling.py:239-300).  Host-driven by construction
    (the solver lives in scipy); every right-hand-side evaluation is one native network evaluation.  Like the
    reference it conditions on nothing but (x, t) (:290) and starts from 32x32 unless `z` is given (:279-283).  The
    right-hand side keeps the reference's precision: t goes to the network in fp32, g(t)^2 is squared in fp32 (:296) and
    multiplies the float64 score.  Keyword-only `return_nfev` (not in the reference, which only logs the count) also
    returns the number of right-hand-side evaluations."""
    from scipy import integrate
    ones = torch.ones(batch_size, device=device)
    if z is None:
        init_x = torch.randn(batch_size, 1, 32, 32, device=device) * marginal_prob_std(ones)[:, None, None, None]
    else:
        init_x = z
    shape = init_x.shape

    def rhs(t, xflat):
        xs = torch.tensor(xflat, device=device, dtype=torch.float32).reshape(shape)
        tt = torch.tensor(np.ones((shape[0],)) * t, device=device, dtype=torch.float32)
        with torch.no_grad():
            s = score_model(xs, tt)
        g = diffusion_coeff(torch.tensor(t)).cpu().numpy()          # fp32 0-d array, as in the reference
        return -0.5 * (g ** 2) * s.cpu().numpy().reshape(-1).astype(np.float64)

    res = integrate.solve_ivp(rhs, (1.0, eps), init_x.reshape(-1).cpu().numpy(), rtol=rtol, atol=atol, method="RK45")
    logger.info(f"Number of function evaluations: {res.nfev}")
    x = torch.tensor(res.y[:, -1], device=device).reshape(shape)
    return (x, res.nfev) if return_nfev else x


def edm_sigma_schedule(n_steps, sigma_min=0.002, sigma_max=80, rho=7.0, device="cuda"):
    """Karras sigma ladder (reference score_sampling.py:304-307; unused by the CLI)."""
    i = torch.linspace(0, 1, n_steps, device=device)
    return (sigma_max ** (1 / rho) + i * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
