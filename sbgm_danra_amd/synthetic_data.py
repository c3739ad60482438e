"""Synthetic ERA5/DANRA-shaped batches in the dict layout the reference's dataset yields and `extract_samples`
consumes (reference data_modules.py:957-993, utils.py:405-480).  The real zarr-backed dataset is out of scope
(SURVEY.md §2.1); shapes, key names, value ranges and the value||mask convention of the geo channels follow
SURVEY.md §8d."""
from __future__ import annotations

import torch
from torch.utils.data import DataLoader, Dataset


class SyntheticDownscalingDataset(Dataset):
    def __init__(self, cfg, n_items: int = 64, seed: int = 42, raw_geo: bool = False):
        self.raw_geo = raw_geo       # True: 1-channel geo fields; the mask channel is then assembled on the device
        self.hw = tuple(cfg["highres"]["data_size"])
        self.hr_var = cfg["highres"]["variable"]
        self.lr_vars = list(cfg["lowres"]["condition_variables"] or [])
        geo = cfg["stationary_conditions"]["geographic_conditions"]
        self.geo = list(geo["geo_variables"]) if geo["sample_w_geo"] else []
        self.sdf = bool(geo.get("sample_w_sdf", False))
        sea = cfg["stationary_conditions"]["seasonal_conditions"]
        self.n_seasons = sea["n_seasons"] if sea["sample_w_cond_season"] else 0
        self.n_items, self.seed = n_items, seed

    def __len__(self):
        return self.n_items

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        h, w = self.hw
        item = {f"{self.hr_var}_hr": torch.randn(1, h, w, generator=g)}
        for v in self.lr_vars:
            item[f"{v}_lr"] = torch.randn(1, h, w, generator=g)
        if "lsm" in self.geo:
            item["lsm"] = torch.cat([(torch.rand(1, h, w, generator=g) > 0.5).float(), torch.ones(1, h, w)], 0)
            item["lsm_hr"] = item["lsm"][:1]
        if "topo" in self.geo:
            item["topo"] = torch.cat([torch.rand(1, h, w, generator=g), torch.ones(1, h, w)], 0)
        if self.sdf:
            item["sdf"] = torch.rand(1, h, w, generator=g)
        if self.n_seasons:
            item["classifier"] = torch.randint(1, self.n_seasons + 1, (), generator=g)
        if self.raw_geo:
            for k in ("lsm", "topo"):
                if k in item:
                    item[k] = item[k][:1]
        return item


def synthetic_loader(cfg, batch_size, n_items=None, seed=42, shuffle=False, raw_geo=False, shard=None, drop_last=False) -> DataLoader:
    """`shard=(rank, world)`: every rank iterates over its own 1/world of the items (torch DistributedSampler), so the
    data-parallel replicas see different samples and the effective batch is batch_size * world."""
    n_items = n_items or 4 * batch_size
    ds = SyntheticDownscalingDataset(cfg, n_items, seed, raw_geo)
    if shard is not None and shard[1] > 1:
        from torch.utils.data.distributed import DistributedSampler
        smp = DistributedSampler(ds, num_replicas=shard[1], rank=shard[0], shuffle=shuffle, seed=seed, drop_last=drop_last)
        return DataLoader(ds, batch_size=batch_size, sampler=smp, num_workers=0, drop_last=drop_last)
    return DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=0, drop_last=drop_last)
