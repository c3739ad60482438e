"""CPU oracle of the pre-network row.  TEST INFRASTRUCTURE ONLY — nothing under sbgm_danra_amd/
may import this module.

Restates the tail of the reference dataset's `__getitem__` (classifier-free-guidance condition dropout and the value||mask
assembly of the geo fields, reference sbgm/data_modules.py:957-993) for ONE sample, and `extract_samples` (reference
sbgm/utils.py:405-480) for a collated batch.

Pinning: `sbgm/data_modules.py` and `sbgm/utils.py` import zarr / netCDF4 / omegaconf at module top, none of which is
installed here, so neither can be imported: PARITY UNPINNED by the reference for this row.  The restatement follows the
cited lines statement by statement (pure tensor bookkeeping: zeros_like, cat, fill_, full_like).
"""
import torch


def finish_sample(sample_dict, split, guidance_cfg, draw):
    """data_modules.py:957-993.  `draw` is the value of the reference's `torch.rand(())` for this sample.
    Note the reference looks the threshold up as `cfg_guidance.get(drop_prob, 0.1)` — keyed by the VALUE of drop_prob, so
    the effective probability is 0.1 unless the dict happens to hold such a key (:964); kept as is."""
    g = guidance_cfg or {}
    drop_prob = g.get("drop_prob", 0.1)
    dropped = False
    if split == "train" and g.get("enabled", False):
        if draw < g.get(drop_prob, 0.1):
            dropped = True
            for key, val in list(sample_dict.items()):
                if key.endswith("_lr") and val is not None:
                    sample_dict[key] = torch.zeros_like(val)
            for geo_key in ("lsm", "topo"):
                geo = sample_dict.get(geo_key)
                if geo is not None:
                    sample_dict[geo_key] = torch.cat([geo, torch.zeros_like(geo)], dim=0)
            if sample_dict.get("classifier") is not None:
                sample_dict["classifier"].fill_(0)
    for geo_key in ("lsm", "topo"):
        geo = sample_dict.get(geo_key)
        if geo is not None and geo.shape[0] == 1:
            sample_dict[geo_key] = torch.cat([geo, torch.full_like(geo, 0.0 if dropped else 1.0)], dim=0)
    return sample_dict, dropped


def extract_samples(samples, device="cpu"):
    """utils.py:405-480"""
    hr_keys = [k for k in samples if k.endswith("_hr") and not k.endswith("_original")]
    if "lsm_hr" in hr_keys:
        hr_keys.remove("lsm_hr")
    if not hr_keys:
        raise ValueError("No HR image found in samples dictionary.")
    hr = samples[hr_keys[0]].to(device).float()
    cls = samples.get("classifier")
    lr_keys = [k for k in samples if k.endswith("_lr") and not k.endswith("_original")]
    if not lr_keys:
        lr = None
    elif len(lr_keys) == 1:
        lr = samples[lr_keys[0]].to(device).float()
    else:
        lr = torch.cat([samples[k].to(device).float() for k in sorted(lr_keys)], dim=1)
    f = lambda k: None if samples.get(k) is None else samples[k].to(device).float()   # noqa: E731
    return hr, cls, lr, f("lsm_hr"), f("lsm"), f("sdf"), f("topo"), f("hr_point"), f("lr_point")
