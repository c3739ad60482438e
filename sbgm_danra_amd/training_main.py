"""`train_main(cfg)` — reference sbgm/training_main.py:15-194 without the plotting / dataloader-timing side shows:
seed, device pick, loaders, model, optimizer, (unstepped) scheduler, pipeline, optional checkpoint load, train."""
from __future__ import annotations

import logging
import os

import numpy as np
import torch

from . import parallel
from .score_unet import diffusion_coeff_fn, loss_fn, marginal_prob_std_fn
from .training import TrainingPipeline_general
from .training_utils import get_dataloader, get_model, get_optimizer, get_scheduler, setup_logger


def train_main(cfg, loaders=None):
    log = setup_logger(os.path.join(cfg["paths"]["checkpoint_dir"], "logs"))
    log.info(f"=== Starting SBGM_SD Training Pipeline === experiment: {cfg['experiment']['name']}")
    rank, world, local = parallel.init_distributed()
    if cfg["training"]["device"] == "cuda" and torch.cuda.is_available():
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    train_dl, val_dl, gen_dl = loaders if loaders is not None else get_dataloader(cfg, shard=(rank, world))
    seed = cfg["training"]["seed"]
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    np.random.seed(seed)
    model, ckpt_dir, ckpt_name = get_model(cfg)
    model = model.to(device)
    optimizer = get_optimizer(cfg, model)
    scheduler = get_scheduler(cfg, optimizer) if cfg["training"].get("lr_scheduler") else None
    pipe = TrainingPipeline_general(model=model, loss_fn=loss_fn, marginal_prob_std_fn=marginal_prob_std_fn,
                                    diffusion_coeff_fn=diffusion_coeff_fn, optimizer=optimizer, device=device,
                                    lr_scheduler=scheduler, cfg=cfg)
    parallel.broadcast_parameters(model)
    if world > 1:                 # identical weights everywhere (above), but every replica draws its OWN loss noise (t, z) and
        torch.manual_seed(seed + rank)   # dropout flags: with one shared stream the replicas' gradients would be copies of each other
        torch.cuda.manual_seed(seed + rank)
    if cfg["training"].get("sync_batchnorm", False):
        from .train_graph import set_sync_batchnorm
        set_sync_batchnorm(True)
    ckpt = os.path.join(ckpt_dir, ckpt_name)
    if cfg["training"].get("load_checkpoint") and os.path.exists(ckpt):
        pipe.load_checkpoint(ckpt, load_ema=cfg["training"].get("load_ema", False))
    log.info(f"▸ trainable parameters: {sum(p.numel() for p in model.parameters() if p.requires_grad):,}  (rank {rank}/{world})")
    return pipe.train(train_dl, val_dl, gen_dl, cfg, epochs=cfg["training"]["epochs"], verbose=cfg["training"].get("verbose", True),
                      use_mixed_precision=cfg["training"].get("use_mixed_precision", False))


logging.getLogger(__name__)
