"""Model / optimizer / loader factories with the reference's names and config keys
(reference sbgm/training_utils.py: infer_in_channels :585-595, get_model :597-669, get_optimizer :672-696,
get_scheduler, get_dataloader, get_gen_dataloader, setup_logger :793-823).

Data: the reference reads zarr archives (out of scope); here `get_dataloader` serves synthetic
batches of the same dict layout unless a caller passes its own loaders to the pipeline."""
from __future__ import annotations

import logging
import os
import sys
import time

import torch.nn as nn
from torch.optim import SGD
from torch.optim import lr_scheduler as _sched

from .optim import Adam, AdamW            # torch's classes with a one-launch native step() for GPU parameters
from .score_unet import Decoder, Encoder, ScoreNet, marginal_prob_std_fn
from .synthetic_data import synthetic_loader
from .utils import get_model_string

logger = logging.getLogger(__name__)


def infer_in_channels(cfg) -> int:
    n_lr = len(cfg["lowres"]["condition_variables"]) if cfg["lowres"]["condition_variables"] is not None else 0
    geo = cfg["stationary_conditions"]["geographic_conditions"]
    return n_lr + (2 * len(geo["geo_variables"]) if geo["sample_w_geo"] else 0)


def get_model(cfg):
    """-> (ScoreNet, checkpoint_dir, checkpoint_name); same config keys and defaults as the reference."""
    mcfg = cfg.get("model", {}) or {}
    act = {"relu": nn.ReLU, "silu": nn.SiLU, "gelu": nn.GELU}.get(str(mcfg.get("decoder_activation", "SiLU")).lower(), nn.ReLU)
    sea = cfg["stationary_conditions"]["seasonal_conditions"]
    encoder = Encoder(input_channels=infer_in_channels(cfg), time_embedding=cfg["sampler"]["time_embedding"],
                      cond_on_img=cfg["lowres"]["condition_variables"] is not None,
                      block_layers=list(cfg["sampler"]["block_layers"]),
                      num_classes=sea["n_seasons"] if sea["sample_w_cond_season"] else None,
                      n_heads=cfg["sampler"]["num_heads"])
    decoder = Decoder(last_fmap_channels=cfg["sampler"]["last_fmap_channels"], output_channels=1,
                      time_embedding=cfg["sampler"]["time_embedding"], n_heads=cfg["sampler"]["num_heads"],
                      use_resize_conv=bool(mcfg.get("use_resize_conv", True)), norm=mcfg.get("decoder_norm", "group"),
                      gn_groups=int(mcfg.get("decoder_gn_groups", 8)), activation=act)
    model = ScoreNet(marginal_prob_std=marginal_prob_std_fn, encoder=encoder, decoder=decoder, debug_pre_sigma_div=False)
    ckpt_dir = os.path.join(cfg["paths"]["path_save"], cfg["paths"]["checkpoint_dir"])
    return model, ckpt_dir, get_model_string(cfg) + ".pth.tar"


def get_optimizer(cfg, model):
    t = cfg["training"]
    kind = t["optimizer"]
    if kind == "adam":
        return Adam(model.parameters(), lr=t["learning_rate"], weight_decay=t["weight_decay"])
    if kind == "adamw":
        return AdamW(model.parameters(), lr=t["learning_rate"], weight_decay=t["weight_decay"])
    if kind == "sgd":
        return SGD(model.parameters(), lr=t["learning_rate"], momentum=t["momentum"], weight_decay=t["weight_decay"])
    raise ValueError(f"Optimizer {kind} not recognized. Use 'adam', 'sgd', or 'adamw'.")


def get_scheduler(cfg, optimizer):
    """Built like the reference; note the reference never steps it and neither does the pipeline."""
    t = cfg["training"]
    kind, p = t.get("lr_scheduler"), dict(t.get("lr_scheduler_params", {}) or {})
    if kind is None:
        return None
    if kind == "ReduceLROnPlateau":
        return _sched.ReduceLROnPlateau(optimizer, mode="min", factor=p.get("factor", 0.5), patience=p.get("patience", 5),
                                        threshold=p.get("threshold", 0.01), min_lr=p.get("min_lr", t.get("min_lr", 1e-6)))
    if kind == "StepLR":
        return _sched.StepLR(optimizer, step_size=p.get("step_size", 10), gamma=p.get("gamma", 0.1))
    if kind == "CosineAnnealingLR":
        return _sched.CosineAnnealingLR(optimizer, T_max=p.get("T_max", 10), eta_min=p.get("eta_min", t.get("min_lr", 1e-6)))
    raise ValueError(f"Scheduler {kind} not recognized.")


def get_dataloader(cfg, shard=None):
    """-> (train, val, gen) loaders of synthetic batches (see module docstring).  `shard=(rank, world)` splits the training
    items over the data-parallel ranks (the reference's train loader drops the ragged last batch, data_modules / training_utils
    `drop_last=True`; kept here so every rank runs the same number of steps); validation and generation stay whole on every rank."""
    bs = cfg["training"]["batch_size"]
    n_gen = max(1, int(cfg["data_handling"].get("n_gen_samples", 1) or 1))
    world = shard[1] if shard else 1
    return (synthetic_loader(cfg, bs, n_items=4 * bs * world, seed=1, shard=shard, drop_last=True),
            synthetic_loader(cfg, bs, n_items=2 * bs, seed=2), synthetic_loader(cfg, n_gen, n_items=n_gen, seed=3))


def get_gen_dataloader(cfg, shard=None):
    """`shard=(rank, world)`: every rank draws its OWN evaluation batch (independent units, no collective: SURVEY.md 8e)"""
    bs = int(cfg["evaluation"]["batch_size"])
    world = shard[1] if shard is not None else 1
    return synthetic_loader(cfg, bs, n_items=bs * world, seed=4, shard=shard if world > 1 else None)


def setup_logger(log_dir, name="train_log"):
    os.makedirs(log_dir, exist_ok=True)
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    path = os.path.join(log_dir, f"{name}_{time.strftime('%Y%m%d_%H%M%S')}.log")
    if not any(isinstance(h, logging.FileHandler) and h.baseFilename == path for h in root.handlers):
        fmt = logging.Formatter("%(asctime)s %(levelname)s %(name)s: %(message)s")
        for h in (logging.FileHandler(path), logging.StreamHandler(sys.stdout)):
            h.setFormatter(fmt)
            root.addHandler(h)
    return logging.getLogger("sbgm")
