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


class SampleGenerator:
    def __init__(self, cfg, model, dataloader, back_transforms, device, literal_reference_bn: bool = False):
        self.cfg, self.model, self.dataloader, self.back_transforms, self.device = cfg, model, dataloader, back_transforms, device
        self.model.train() if literal_reference_bn else self.model.eval()
        self.model_name_str = get_model_string(cfg)
        self.output_dir = os.path.join(cfg["paths"]["sample_dir"], "generation", self.model_name_str)
        self.sample_path = os.path.join(self.output_dir, "generated_samples")
        os.makedirs(self.sample_path, exist_ok=True)

    def _run_sampler(self, batch_size, y, cond_img, lsm_cond, topo_cond):
        gen = pc_sampler(score_model=self.model, marginal_prob_std=marginal_prob_std_fn, diffusion_coeff=diffusion_coeff_fn,
                         batch_size=batch_size, num_steps=self.cfg["sampler"]["n_timesteps"], device=self.device,
                         img_size=self.cfg["highres"]["data_size"][0], y=y, cond_img=cond_img, lsm_cond=lsm_cond,
                         topo_cond=topo_cond)
        gen = gen.squeeze().detach().cpu()
        if gen.ndim == 4:
            gen = gen.squeeze(1)
        elif gen.ndim == 2:
            gen = gen.unsqueeze(0)
        elif gen.ndim != 3:
            raise ValueError(f"Unknown generated sample shape: {gen.shape}")
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

    def generate_multiple(self):
        x, seasons, cond, lsm, topo = self._batch()
        gen = self._run_sampler(x.shape[0], seasons, cond, lsm, topo)
        self._save_npz({"gen_samples": gen, "eval_samples": x, "lsm_samples": lsm, "seasons": seasons}, f"multi_n_{x.shape[0]}")
        return gen

    def generate_single(self):
        x, seasons, cond, lsm, topo = self._batch(first_only=True)
        gen = self._run_sampler(1, seasons, cond, lsm, topo)
        self._save_npz({"gen_samples": gen, "eval_samples": x, "lsm_samples": lsm, "seasons": seasons}, "single")
        return gen

    def generate_repeated(self):
        """cfg.evaluation.n_repeats samples from ONE conditioning sample, drawn as one batch (independent noise per
        row); with several ranks the repeats are independent units and are sharded, no collective."""
        x, seasons, cond, lsm, topo = self._batch(first_only=True)
        n = int(self.cfg["evaluation"]["n_repeats"])
        rank, world = parallel.world()
        mine = len(parallel.shard_range(n, rank, world))
        rep = lambda t: None if t is None else t.repeat(mine, *([1] * (t.dim() - 1)))   # noqa: E731
        gen = self._run_sampler(mine, rep(seasons), rep(cond), rep(lsm), rep(topo)) if mine else None
        if gen is not None:
            self._save_npz({"gen_samples": gen, "eval_samples": x, "lsm_samples": lsm, "seasons": seasons},
                           f"repeated_n_{n}" + (f"_rank{rank}" if world > 1 else ""))
        return gen
