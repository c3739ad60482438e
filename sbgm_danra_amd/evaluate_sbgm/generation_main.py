"""`generation_main(cfg)` — reference sbgm/evaluate_sbgm/generation_main.py:47-181: seed, model, checkpoint
(`network_params`), generation loader, `SampleGenerator`, then every `cfg.evaluation.gen_type`."""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import parallel
from ..training_utils import get_gen_dataloader, get_model, setup_logger
from ..utils import get_model_string
from .generation import SampleGenerator


def generation_main(cfg, dataloader=None, back_transforms=None):
    seed = cfg["evaluation"]["seed"]
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    np.random.seed(seed)
    gen_dir = os.path.join(cfg["paths"]["sample_dir"], "generation", get_model_string(cfg))
    log = setup_logger(os.path.join(gen_dir, "logs"), name="gen_log")
    rank, world, local = parallel.init_distributed()
    if world > 1:      # one process per GPU: every rank generates from its own conditioning batch with its own noise and writes *_rank<r> files
        torch.manual_seed(seed + rank)
        torch.cuda.manual_seed(seed + rank)
        np.random.seed(seed + rank)
    dev = cfg["training"]["device"]
    device = torch.device("cuda", local) if dev == "cuda" and torch.cuda.is_available() else torch.device("cpu")
    model, ckpt_dir, ckpt_name = get_model(cfg)
    model = model.to(device)
    state = torch.load(os.path.join(ckpt_dir, ckpt_name), map_location=device, weights_only=True)["network_params"]
    model.load_state_dict(state)
    log.info(f"[INFO] Model checkpoint loaded from: {ckpt_dir}/{ckpt_name}")
    if back_transforms is None and cfg["evaluation"].get("transform_back", False):
        # reference generation_main.py:93-108: inverse transforms from the saved global statistics (device-side classes)
        from ..training import TrainingPipeline_general
        back_transforms = TrainingPipeline_general._build_back_transforms(cfg)
    gen = SampleGenerator(cfg, model, dataloader if dataloader is not None else get_gen_dataloader(cfg, shard=(rank, world)), back_transforms, device)
    out = {}
    for kind in cfg["evaluation"]["gen_type"]:
        if kind not in ("multiple", "single", "repeated"):
            raise ValueError(f"\nUnknown generation type: {kind}\n")
        log.info(f"[INFO] Running generation type: {kind}")
        out[kind] = getattr(gen, f"generate_{kind}")()
    return out
