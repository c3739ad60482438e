"""The four `sbgm/utils.py` helpers that touch the hot path: `extract_samples` (reference
utils.py:405-480), `get_model_string` (:88-128), `load_config` (:1626-1640), `report_precip_extremes` (:1642-1671).
Plotting, zarr/netCDF converters and the rest of that file are out of scope."""
from __future__ import annotations

import torch

from .config_loader import Config, load_config, to_config  # noqa: F401


def extract_samples(samples: dict, device=None):
    """Batch dict -> (hr, classifier, lr(cat over sorted *_lr keys), lsm_hr, lsm, sdf, topo, hr_point, lr_point), on
    `device`, images as fp32 — the tuple layout every reference caller unpacks."""
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

    def img(k):
        v = samples.get(k)
        return None if v is None else v.to(device, non_blocking=True).float()

    hr_keys = [k for k in samples if k.endswith("_hr") and not k.endswith("_original") and k != "lsm_hr"]
    if not hr_keys:
        raise ValueError("No HR image found in samples dictionary.")
    hr = img(hr_keys[0])
    cls = samples.get("classifier")
    if cls is not None:
        cls = cls.to(device, non_blocking=True)
    lr_keys = sorted(k for k in samples if k.endswith("_lr") and not k.endswith("_original"))
    lr = None if not lr_keys else (img(lr_keys[0]) if len(lr_keys) == 1 else torch.cat([img(k) for k in lr_keys], dim=1))
    pts = [None if samples.get(k) is None else samples[k].to(device).float() for k in ("hr_point", "lr_point")]
    return hr, cls, lr, img("lsm_hr"), img("lsm"), img("sdf"), img("topo"), pts[0], pts[1]


def draw_condition_dropout(batch_size: int, split: str, guidance_cfg) -> torch.Tensor | None:
    """Per-sample classifier-free-guidance dropout flags, drawn the way the reference dataset draws them: one
    `torch.rand(())` per sample from the default CPU generator, training split only (data_modules.py:960-965).  The
    threshold lookup is the reference's, including its quirk: `cfg.get(drop_prob, 0.1)` is keyed by the VALUE of drop_prob,
    so the effective probability is 0.1 unless the dict holds such a key.  Returns uint8 [B] on the host, or None."""
    g = guidance_cfg or {}
    if split != "train" or not g.get("enabled", False):
        return None
    thr = g.get(g.get("drop_prob", 0.1), 0.1)
    return torch.tensor([1 if float(torch.rand(())) < thr else 0 for _ in range(batch_size)], dtype=torch.uint8)


def extract_samples_device(samples: dict, device, dropped: torch.Tensor | None = None):
    """`extract_samples` for RAW batches (1-channel geo fields, separate *_lr fields), with the dataset-side condition
    handling moved onto the device and batched: one `sbgm_assemble_conditions` call concatenates
    the sorted *_lr fields, zeroes them / nulls the class label for dropped samples and writes the value||mask geo layout
    (reference utils.py:441-447 + data_modules.py:957-993).  Same 9-tuple as `extract_samples`."""
    import ctypes as C

    from . import _native as N
    dev = torch.device(device)
    if dev.type != "cuda":
        raise N.NativeError(f"extract_samples_device assembles the batch on a ROCm device, got device={device!r}")

    def img(k):
        v = samples.get(k)
        return None if v is None else N.f32c(v.to(dev, non_blocking=True))

    hr_keys = [k for k in samples if k.endswith("_hr") and not k.endswith("_original") and k != "lsm_hr"]
    if not hr_keys:
        raise ValueError("No HR image found in samples dictionary.")
    hr = img(hr_keys[0])
    B, hw = hr.shape[0], hr.shape[-2] * hr.shape[-1]
    lr_keys = sorted(k for k in samples if k.endswith("_lr") and not k.endswith("_original"))
    lrs = [img(k) for k in lr_keys]
    lsm, topo = img("lsm"), img("topo")
    cls = samples.get("classifier")
    cls = None if cls is None else cls.to(dev, non_blocking=True).to(torch.int64).contiguous()
    drop_d = None if dropped is None else dropped.to(dev).to(torch.uint8).contiguous()
    a = N.AssembleArgs()
    a.B, a.HW, a.n_lr = B, hw, len(lrs)
    for i, t in enumerate(lrs):
        a.lr[i], a.lr_channels[i] = t.data_ptr(), t.shape[1]
    c_lr = sum(t.shape[1] for t in lrs)
    lr_out = torch.empty(B, c_lr, *hr.shape[-2:], device=dev) if lrs else None
    lsm_out = None if lsm is None else torch.empty(B, 2, *hr.shape[-2:], device=dev)
    topo_out = None if topo is None else torch.empty(B, 2, *hr.shape[-2:], device=dev)
    y_out = None if cls is None else torch.empty_like(cls)
    a.lsm, a.lsm_channels = N.ptr(lsm), 0 if lsm is None else lsm.shape[1]
    a.topo, a.topo_channels = N.ptr(topo), 0 if topo is None else topo.shape[1]
    a.y, a.dropped = N.ptr(cls), N.ptr(drop_d)
    a.lr_out, a.lsm_out, a.topo_out, a.y_out = N.ptr(lr_out), N.ptr(lsm_out), N.ptr(topo_out), N.ptr(y_out)
    N.check(N.lib().sbgm_assemble_conditions(C.byref(a), N.stream()))
    pts = [None if samples.get(k) is None else samples[k].to(dev).float() for k in ("hr_point", "lr_point")]
    return hr, y_out, lr_out, img("lsm_hr"), lsm_out, img("sdf"), topo_out, pts[0], pts[1]


def get_model_string(cfg) -> str:
    """Checkpoint / output naming scheme of the reference (utils.py:88-128)."""
    hr = tuple(cfg["highres"]["data_size"]) if cfg["highres"].get("data_size") is not None else (128, 128)
    rf = cfg["lowres"].get("resize_factor", 1) or 1
    if rf > 1:
        hr = (hr[0] // rf, hr[1] // rf)
    lr_vars = "_".join(cfg["lowres"]["condition_variables"] or [])
    return (f"{cfg['experiment']['config_name']}__HR_{cfg['highres']['variable']}_{cfg['highres']['model']}__"
            f"SIZE_{hr[0]}x{hr[1]}__LR_{lr_vars}_{cfg['lowres']['model']}__LOSS_{cfg['training']['loss_type']}__"
            f"HEADS_{cfg['sampler']['num_heads']}__TIMESTEPS_{cfg['sampler']['n_timesteps']}")


def report_precip_extremes(x_bt: torch.Tensor, name: str, cap_mm_day: float = 500.0, logger=print) -> dict:
    """Per-sample sentinel on back-transformed precipitation (reference utils.py:1642-1671): a sample is flagged when its
    max exceeds max(5 x p99.9, cap) or is negative.  The two statistics (max, 99.9th percentile with torch.quantile's
    linear interpolation) are computed on the device by `sbgm_sample_extremes`; only 2*B floats come back to the host.
    Same messages and return dictionaries as the reference."""
    from .special_transforms import sample_extremes
    mx_t, p_t = sample_extremes(x_bt.flatten(1), 0.999)
    stats = torch.stack([p_t, mx_t]).cpu()
    n_ex, vals_ex, n_b0, vals_b0 = 0, [], 0, []
    for i, (p, m) in enumerate(zip(stats[0].tolist(), stats[1].tolist())):
        if m > max(5.0 * p, cap_mm_day):
            logger(f"{name} sample {i} has extreme precipitation: max={m:.1f} mm/day > max(5xp99.9={p:.1f} mm/day)")
            n_ex += 1
            vals_ex.append(m)
        if m < 0:
            logger(f"{name} sample {i} has negative precipitation: max={m:.1f} mm/day < 0")
            n_b0 += 1
            vals_b0.append(m)
    if n_b0 > 0 and n_ex > 0:
        return {"has_extreme": True, "n_extreme": n_ex, "extreme_values": vals_ex, "has_below_zero": True,
                "n_below_zero": n_b0, "below_zero_values": vals_b0}
    if n_ex > 0:
        return {"has_extreme": True, "n_extreme": n_ex, "extreme_values": vals_ex}
    if n_b0 > 0:
        return {"has_below_zero": True, "n_below_zero": n_b0, "below_zero_values": vals_b0}
    return {"has_extreme": False}
