"""Device-side mirror of the reference's transform classes (sbgm/special_transforms.py) — SURVEY.md §8f rank 1.

Same class names, constructor arguments, validation errors and call convention (`transform(sample) -> tensor`), but
`sample` is a fp32 tensor on the ROCm device and the arithmetic runs in ONE pass of `sbgm_pointwise_chain`
(csrc/postproc.hip): every class compiles itself into a short program of scalar steps that keeps the reference's op
order and per-op fp32 rounding, so the affine transforms are bit-identical to the reference and exp/log agree to libm
precision.  `fuse(...)` concatenates programs (e.g. back-transform + generation clamp, training.py:744-748) into one
launch.  There is no host path: a CPU tensor raises NativeError.

    Scale, ScaleBackTransform            reference :62-100, :103-139
    ZScoreTransform, ZScoreBackTransform reference :143-184, :187-233
    PrcpLogTransform, PrcpLogBackTransform  reference :239-355, :360-462
    build_back_transforms, load_global_stats, get_transforms_from_stats, get_backtransforms_from_stats,
    build_back_transforms_from_stats     reference :465-520, :576-592, :595-637, :639-684, :523-573
"""
import ctypes as C
import json
import logging
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _native as N

logger = logging.getLogger(__name__)

ADD, MUL, DIV, CLAMP_MIN, CLAMP_MAX, EXP, LOG = range(7)
MAX_OPS = 12


def _f32(v) -> float:
    """a Python/NumPy/torch scalar as the fp32 value torch's tensor-scalar kernels would use"""
    if isinstance(v, torch.Tensor):
        v = v.detach().reshape(-1)[0].item() if v.numel() == 1 else _not_scalar(v)
    return float(np.float32(v))


def _not_scalar(v):
    raise ValueError(f"per-element statistics of shape {tuple(v.shape)} are not supported on the device path (scalar mean/std only)")


def apply_chain(sample: torch.Tensor, program, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = program(sample), one launch.  program = [(op, const), ...]"""
    if not isinstance(sample, torch.Tensor):
        sample = torch.tensor(sample, dtype=torch.float32)
    N.require_device(sample)
    x = N.f32c(sample)
    if len(program) > MAX_OPS:
        raise ValueError(f"transform program has {len(program)} steps (max {MAX_OPS})")
    y = torch.empty_like(x) if out is None else out
    ops = (C.c_int * max(1, len(program)))(*[int(o) for o, _ in program])
    cs = (C.c_float * max(1, len(program)))(*[float(c) for _, c in program])
    N.check(N.lib().sbgm_pointwise_chain(x.data_ptr(), y.data_ptr(), x.numel(), len(program), ops, cs, N.stream()))
    return y


class _Chain:
    def program(self):
        raise NotImplementedError

    def __call__(self, sample):
        return apply_chain(sample, self.program())


def fuse(*stages):
    """one transform out of several (`_Chain` objects or raw programs), applied left to right in a single launch"""
    prog = []
    for s in stages:
        prog += list(s.program() if isinstance(s, _Chain) else s)

    class _Fused(_Chain):
        def program(self):
            return prog
    return _Fused()


def clamp_program(lo=None, hi=None):
    """torch.clamp(x, min=lo, max=hi) as program steps"""
    prog = []
    if lo is not None and lo != -float("inf"):
        prog.append((CLAMP_MIN, _f32(lo)))
    if hi is not None and hi != float("inf"):
        prog.append((CLAMP_MAX, _f32(hi)))
    return prog


class Scale(_Chain):
    """[data_min_in, data_max_in] -> [in_low, in_high]   (reference :62-100)"""
    def __init__(self, in_low, in_high, data_min_in=0, data_max_in=1):
        self.in_low, self.in_high, self.data_min_in, self.data_max_in = in_low, in_high, data_min_in, data_max_in

    def program(self):
        old, new = self.data_max_in - self.data_min_in, self.in_high - self.in_low
        return [(ADD, -_f32(self.data_min_in)), (MUL, _f32(new)), (DIV, _f32(old)), (ADD, _f32(self.in_low))]


class ScaleBackTransform(_Chain):
    """[in_low, in_high] -> [data_min_in, data_max_in]   (reference :103-139)"""
    def __init__(self, in_low=0, in_high=1, data_min_in=0, data_max_in=1):
        self.in_low, self.in_high, self.data_min_in, self.data_max_in = in_low, in_high, data_min_in, data_max_in

    def program(self):
        old, new = self.in_high - self.in_low, self.data_max_in - self.data_min_in
        return [(ADD, -_f32(self.in_low)), (MUL, _f32(new)), (DIV, _f32(old)), (ADD, _f32(self.data_min_in))]


def _std_eps(std) -> float:
    # the reference adds eps to an fp32 TENSOR (:182, :231): the sum is rounded to fp32
    return float(np.float32(np.float32(_f32(std)) + np.float32(1e-8)))


class ZScoreTransform(_Chain):
    """(x - mean) / (std + 1e-8)   (reference :143-184)"""
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def program(self):
        return [(ADD, -_f32(self.mean)), (DIV, _std_eps(self.std))]


class ZScoreBackTransform(_Chain):
    """x * (std + 1e-8) + mean   (reference :187-233)"""
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def program(self):
        return [(MUL, _std_eps(self.std)), (ADD, _f32(self.mean))]


_LOG_TYPES = ("log_zscore", "log_01", "log_minus1_1", "log")


def _check_log_args(scale_type, mean, std, lo, hi):
    if scale_type == "log_zscore" and (mean is None or std is None):
        raise ValueError("Global mean and standard deviation not provided. Using local statistics is not recommended.")
    if scale_type in ("log_01", "log_minus1_1") and (lo is None or hi is None):
        raise ValueError("Min and max log values not provided. Using global statistics is recommended.")
    if scale_type not in _LOG_TYPES:
        raise ValueError("Invalid scale type. Please choose from ['log_01', 'log_zscore', 'log_minus1_1', 'log'].")


class PrcpLogTransform(_Chain):
    """log(x + eps), then z-score / [0,1] / [-1,1] scaling in log space   (reference :239-355).
    The log range is widened by buffer_frac of the range on each side (:262-266)."""
    def __init__(self, eps=0.01, scale_type="log_zscore", glob_mean_log=None, glob_std_log=None, glob_min_log=None,
                 glob_max_log=None, buffer_frac=0.5):
        self.eps, self.scale_type = eps, scale_type
        self.glob_mean_log, self.glob_std_log = glob_mean_log, glob_std_log
        self.glob_min_log, self.glob_max_log, self.buffer_frac = glob_min_log, glob_max_log, buffer_frac
        if glob_min_log is not None and glob_max_log is not None:
            rng = glob_max_log - glob_min_log
            self.glob_min_log = glob_min_log - buffer_frac * rng
            self.glob_max_log = glob_max_log + buffer_frac * rng
        _check_log_args(scale_type, glob_mean_log, glob_std_log, glob_min_log, glob_max_log)

    def program(self):
        prog = [(ADD, _f32(self.eps)), (LOG, 0.0)]
        if self.scale_type == "log_01":
            denom = self.glob_max_log - self.glob_min_log
            if denom == 0:
                raise ValueError("The log-range of data is zero. Cannot scale to [0, 1]. Please check the data.")
            prog += [(ADD, -_f32(self.glob_min_log)), (DIV, _f32(denom))]
        elif self.scale_type == "log_zscore":
            prog += [(ADD, -_f32(self.glob_mean_log)), (DIV, _f32(self.glob_std_log + 1e-8))]
        elif self.scale_type == "log_minus1_1":
            prog += [(ADD, -_f32(self.glob_min_log)), (DIV, _f32(self.glob_max_log - self.glob_min_log)), (MUL, 2.0), (ADD, -1.0)]
        return prog


class PrcpLogBackTransform(_Chain):
    """inverse of PrcpLogTransform with an optional clamp in log space before exp   (reference :360-462).
    Here the log range is widened by buffer_frac/2 per side (:393-399) — kept as in the reference."""
    def __init__(self, scale_type="log_zscore", glob_mean_log=None, glob_std_log=None, glob_min_log=None, glob_max_log=None,
                 buffer_frac=0.5, clamp_log_min=None, clamp_log_max=None):
        self.scale_type = scale_type
        self.glob_mean_log, self.glob_std_log = glob_mean_log, glob_std_log
        self.glob_min_log, self.glob_max_log, self.buffer_frac = glob_min_log, glob_max_log, buffer_frac
        self.clamp_log_min, self.clamp_log_max = clamp_log_min, clamp_log_max
        self.hi = float("inf") if clamp_log_max is None else float(clamp_log_max)
        self.lo = -float("inf") if clamp_log_min is None else float(clamp_log_min)
        if glob_min_log is not None and glob_max_log is not None:
            logger.info(f"Extended log range from [{glob_min_log}, {glob_max_log}]")
            rng = glob_max_log - glob_min_log
            self.glob_min_log = glob_min_log - (buffer_frac / 2) * rng
            self.glob_max_log = glob_max_log + (buffer_frac / 2) * rng
            logger.info(f"to [{self.glob_min_log}, {self.glob_max_log}]\n")
        _check_log_args(scale_type, glob_mean_log, glob_std_log, glob_min_log, glob_max_log)

    def program(self):
        if self.scale_type == "log_01":
            prog = [(MUL, _f32(self.glob_max_log - self.glob_min_log)), (ADD, _f32(self.glob_min_log))]
        elif self.scale_type == "log_zscore":
            prog = [(MUL, _f32(self.glob_std_log + 1e-8)), (ADD, _f32(self.glob_mean_log))]
        elif self.scale_type == "log_minus1_1":      # 0.5 * (x + 1) * range + min, evaluated left to right
            prog = [(ADD, 1.0), (MUL, 0.5), (MUL, _f32(self.glob_max_log - self.glob_min_log)), (ADD, _f32(self.glob_min_log))]
        else:
            prog = []
        return prog + clamp_program(self.lo, self.hi) + [(EXP, 0.0)]


# ---- factories (host logic, reference :465-684) ------------------------------------------------------------------------
def _log_back(mth, prm):
    return PrcpLogBackTransform(scale_type=mth, glob_mean_log=prm["glob_mean_log"], glob_std_log=prm["glob_std_log"],
                                glob_min_log=prm["glob_min_log"], glob_max_log=prm["glob_max_log"], buffer_frac=prm["buffer_frac"],
                                clamp_log_min=prm.get("clamp_log_min", None), clamp_log_max=prm.get("clamp_log_max", None))


def build_back_transforms(hr_var, hr_scaling_method, hr_scaling_params, lr_vars, lr_scaling_methods, lr_scaling_params):
    """plot-key -> inverse transform ('<hr_var>_hr', 'generated', '<cond>_lr')   (reference :465-520)"""
    def one(mth, prm, what):
        if mth in _LOG_TYPES:
            return _log_back(mth, prm)
        if mth == "zscore":
            return ZScoreBackTransform(prm["glob_mean"], prm["glob_std"])
        if mth == "01":
            return ScaleBackTransform(0, 1, prm["glob_min"], prm["glob_max"])
        raise ValueError(f"Unknown {what} scaling method: {mth}")
    bt = {}
    inv = one(hr_scaling_method, hr_scaling_params, "HR")
    bt[f"{hr_var}_hr"] = inv
    bt["generated"] = inv
    for cond, mth, prm in zip(lr_vars, lr_scaling_methods, lr_scaling_params):
        bt[f"{cond}_lr"] = one(mth, prm, "LR")
    return bt


def load_global_stats(variable, model, domain_str, crop_region_str, split, dir_load):
    """reference :576-592"""
    path = os.path.join(dir_load, model, variable, split,
                        f"global_stats__{model}__{domain_str}__crop__{crop_region_str}__{variable}__{split}.json")
    if not os.path.exists(path):
        logger.warning(f"Stats file not found: {path}")
        return None
    logger.info(f"Loading stats from {path}")
    with open(path, "r") as f:
        return json.load(f)


def _resolve_stats(variable, model, domain_str, crop_region_str, split, stats, stats_file_path):
    if stats and stats_file_path:
        stats_file_path = ""
    if stats is None and stats_file_path:
        if not os.path.exists(stats_file_path):
            raise ValueError(f"Stats file not found: {stats_file_path}")
        stats = load_global_stats(variable, model, domain_str, crop_region_str, split, stats_file_path)
    if stats is None:
        raise ValueError(f"Failed to load stats from {stats_file_path}")
    return stats


def get_transforms_from_stats(variable: str, model: str, domain_str: str, crop_region_str: str, split: str, transform_type: str,
                              buffer_frac: float, stats: Optional[dict] = None, stats_file_path: str = ""):
    """reference :595-637"""
    stats = _resolve_stats(variable, model, domain_str, crop_region_str, split, stats, stats_file_path)
    if transform_type == "zscore":
        return ZScoreTransform(mean=stats["mean"], std=stats["std"])
    if transform_type == "scale01":
        return Scale(0, 1, data_min_in=stats["min"], data_max_in=stats["max"])
    if transform_type == "scale_minus1_1":
        return Scale(-1, 1, data_min_in=stats["min"], data_max_in=stats["max"])
    if transform_type in _LOG_TYPES:
        return PrcpLogTransform(scale_type=transform_type, glob_mean_log=stats["log_mean"], glob_std_log=stats["log_std"],
                                glob_min_log=stats["log_min"], glob_max_log=stats["log_max"], buffer_frac=buffer_frac)
    raise ValueError(f"Unknown transform type: {transform_type}")


def get_backtransforms_from_stats(variable: str, model: str, domain_str: str, crop_region_str: str, split: str,
                                  transform_type: str, buffer_frac: float, stats: Optional[dict] = None, stats_file_path: str = ""):
    """reference :639-684 (the log back-transform clamps to the observed log-min / log-max)"""
    stats = _resolve_stats(variable, model, domain_str, crop_region_str, split, stats, stats_file_path)
    if transform_type == "zscore":
        return ZScoreBackTransform(mean=stats["mean"], std=stats["std"])
    if transform_type == "scale01":
        return ScaleBackTransform(0, 1, data_min_in=stats["min"], data_max_in=stats["max"])
    if transform_type == "scale_minus1_1":
        return ScaleBackTransform(-1, 1, data_min_in=stats["min"], data_max_in=stats["max"])
    if transform_type in _LOG_TYPES:
        return PrcpLogBackTransform(scale_type=transform_type, glob_mean_log=stats["log_mean"], glob_std_log=stats["log_std"],
                                    glob_min_log=stats["log_min"], glob_max_log=stats["log_max"], buffer_frac=buffer_frac,
                                    clamp_log_min=stats["log_min"], clamp_log_max=stats["log_max"])
    raise ValueError(f"Unknown transform type: {transform_type}")


def build_back_transforms_from_stats(hr_var: str, hr_model: str, domain_str_hr: str, crop_region_str_hr: str, hr_scaling_method: str,
                                     hr_buffer_frac: float, lr_vars: List[str], lr_model: str, crop_region_str_lr: str,
                                     domain_str_lr: str, lr_scaling_methods: List[str], lr_buffer_frac: float, split: str,
                                     stats_dir_root: str) -> Dict[str, object]:
    """reference :523-573"""
    bt = {}
    inv = get_backtransforms_from_stats(variable=hr_var, model=hr_model, domain_str=domain_str_hr, crop_region_str=crop_region_str_hr,
                                        split=split, transform_type=hr_scaling_method, buffer_frac=hr_buffer_frac,
                                        stats_file_path=stats_dir_root)
    bt[f"{hr_var}_hr"] = inv
    bt["generated"] = inv
    for cond, mth in zip(lr_vars, lr_scaling_methods):
        bt[f"{cond}_lr"] = get_backtransforms_from_stats(variable=cond, model=lr_model, domain_str=domain_str_lr,
                                                         crop_region_str=crop_region_str_lr, split=split, transform_type=mth,
                                                         buffer_frac=lr_buffer_frac, stats_file_path=stats_dir_root)
    return bt


def sample_extremes(x: torch.Tensor, q: float = 0.999):
    """(max, q-quantile) per sample of x[B, ...] on the device (csrc/postproc.hip K32); two [B] device tensors"""
    N.require_device(x)
    x = N.f32c(x)
    B = x.shape[0]
    mx, qq = torch.empty(B, device=x.device), torch.empty(B, device=x.device)
    N.check(N.lib().sbgm_sample_extremes(x.data_ptr(), B, x[0].numel(), float(q), mx.data_ptr(), qq.data_ptr(), N.stream()))
    return mx, qq
