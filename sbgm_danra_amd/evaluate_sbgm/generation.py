"""`SampleGenerator` — reference sbgm/evaluate_sbgm/generation.py:40-314 (its live part): draws a batch, runs
`pc_sampler` with the reference's kwargs, squeezes / moves to CPU exactly like `_run_sampler` (:56-83) and saves
`gen_samples_* / eval_samples_* / lsm_samples_* / seasons_*` npz files.  Plotting and the stats-file back-transforms
are out of scope.

Deliberate deviation: the reference writes `self.model.eval` without calling it, so its generation
runs BatchNorm in training mode.  Here `eval()` IS called; pass `literal_reference_bn=True` to reproduce the
reference's behaviour (the engine supports train-mode BatchNorm in the sampler)."""
from __future__ import annotations

import logging
import os

import numpy as np
import torch

from .. import parallel
from ..score_sampling import pc_sampler
from ..score_unet import diffusion_coeff_fn, marginal_prob_std_fn
from ..utils import extract_samples, get_model_string

logger = logging.getLogger(__name__)


def maybe_inverse_transform(k, arr, back_transforms):
    """reference generation.py:26-35"""
    if back_transforms and k in back_transforms:
        return back_transforms[k](arr)
    return arr


class SampleGenerator:
    def __init__(self, cfg, model, dataloader, back_transforms, device, literal_reference_bn: bool = False):
        self.cfg, self.model, self.dataloader, self.back_transforms, self.device = cfg, model, dataloader, back_transforms, device
        self.model.train() if literal_reference_bn else self.model.eval()
        self.model_name_str = get_model_string(cfg)
        self.output_dir = os.path.join(cfg["paths"]["sample_dir"], "generation", self.model_name_str)
        self.sample_path = os.path.join(self.output_dir, "generated_samples")
        os.makedirs(self.sample_path, exist_ok=True)

    def _sample_device(self, batch_size, y, cond_img, lsm_cond, topo_cond):
        """the sampler output as [B,H,W] still on the device (what _run_sampler returns after .cpu())"""
        gen = pc_sampler(score_model=self.model, marginal_prob_std=marginal_prob_std_fn, diffusion_coeff=diffusion_coeff_fn,
                         batch_size=batch_size, num_steps=self.cfg["sampler"]["n_timesteps"], device=self.device,
                         img_size=self.cfg["highres"]["data_size"][0], y=y, cond_img=cond_img, lsm_cond=lsm_cond,
                         topo_cond=topo_cond)
        gen = gen.squeeze().detach()
        if gen.ndim == 4:
            gen = gen.squeeze(1)
        elif gen.ndim == 2:
            gen = gen.unsqueeze(0)
        elif gen.ndim != 3:
            raise ValueError(f"Unknown generated sample shape: {gen.shape}")
        return gen

    def _run_sampler(self, batch_size, y, cond_img, lsm_cond, topo_cond):
        """reference generation.py:56-83: [B,H,W] on the host"""
        return self._sample_device(batch_size, y, cond_img, lsm_cond, topo_cond).cpu()

    def _apply_backtransforms(self, x, generated, cond_images, seasons=None):
        """reference generation.py:85-107, on the device: the transforms are elementwise, so one launch per key covers the
        whole batch (the reference loops over samples and stacks).  cond_images comes back as the reference's nested list
        [sample][variable] of [H,W] tensors."""
        hr_key = self.cfg["highres"]["variable"] + "_hr"
        if generated.ndim == 2:
            generated = generated.unsqueeze(0)
        x = maybe_inverse_transform(hr_key, x, self.back_transforms)
        generated = maybe_inverse_transform(hr_key, generated, self.back_transforms)
        if cond_images is not None:
            keys = self.cfg["lowres"]["condition_variables"] or []
            per_var = [maybe_inverse_transform(k + "_lr", cond_images[:, i], self.back_transforms) for i, k in enumerate(keys)]
            cond_images = [[v[b] for v in per_var] for b in range(cond_images.shape[0])]
        return x, generated, cond_images

    def _generate(self, x, seasons, cond, lsm, topo, suffix, batch=None):
        gen = self._sample_device(x.shape[0] if batch is None else batch, seasons, cond, lsm, topo)
        cond_out = cond
        if self.cfg["evaluation"].get("transform_back", False):
            x, gen, cond_out = self._apply_backtransforms(x, gen, cond, seasons)
        gen = gen.cpu()
        self._save_npz({"gen_samples": gen, "eval_samples": x, "lsm_samples": lsm, "seasons": seasons}, suffix)
        if cond_out is not None and isinstance(cond_out, list):
            for i, k in enumerate(self.cfg["lowres"]["condition_variables"] or []):
                self._save_npz({f"cond_samples_{k}": torch.stack([im[i] for im in cond_out])}, suffix)
        return gen

    def _save_npz(self, data, suffix):
        for k, v in data.items():
            if v is not None:
                np.savez_compressed(os.path.join(self.sample_path, f"{k}_{suffix}.npz"), v.cpu().numpy() if torch.is_tensor(v) else v)

    def _batch(self, first_only=False):
        x, seasons, cond, _lsm_hr, lsm, _sdf, topo, _hp, _lp = extract_samples(next(iter(self.dataloader)), self.device)
        if first_only:
            x, seasons, cond, lsm, topo = [None if t is None else t[:1] for t in (x, seasons, cond, lsm, topo)]
        return x, seasons, cond, lsm, topo

    @staticmethod
    def _rank_suffix():
        """with several ranks every rank owns whole batches and writes its own files (two ranks writing one path would race)"""
        rank, world = parallel.world()
        return f"_rank{rank}" if world > 1 else ""

    def generate_multiple(self):
        x, seasons, cond, lsm, topo = self._batch()
        return self._generate(x, seasons, cond, lsm, topo, f"multi_n_{x.shape[0]}" + self._rank_suffix())

    def generate_single(self):
        x, seasons, cond, lsm, topo = self._batch(first_only=True)
        return self._generate(x, seasons, cond, lsm, topo, "single" + self._rank_suffix())

    def generate_repeated(self):
        """cfg.evaluation.n_repeats samples from ONE conditioning sample, drawn as one batch (independent noise per
        row); with several ranks the repeats are independent units and are sharded, no collective."""
        x, seasons, cond, lsm, topo = self._batch(first_only=True)
        n = int(self.cfg["evaluation"]["n_repeats"])
        rank, world = parallel.world()
        mine = len(parallel.shard_range(n, rank, world))
        rep = lambda t: None if t is None else t.repeat(mine, *([1] * (t.dim() - 1)))   # noqa: E731
        if not mine:
            return None
        return self._generate(x, rep(seasons), rep(cond), rep(lsm), rep(topo),
                              f"repeated_n_{n}" + (f"_rank{rank}" if world > 1 else ""), batch=mine)
