"""CPU oracle of the post-sampling row: data transforms, back-transforms and the
precipitation sentinel.  TEST INFRASTRUCTURE ONLY — nothing under sbgm_danra_amd/ may import this module.

Restates, in plain PyTorch-CPU fp32:
  Scale / ScaleBackTransform            reference sbgm/special_transforms.py:62-100, :103-139
  ZScoreTransform / ZScoreBackTransform reference sbgm/special_transforms.py:143-184, :187-233
  PrcpLogTransform / PrcpLogBackTransform  reference sbgm/special_transforms.py:239-355, :360-462
  report_precip_extremes                reference sbgm/utils.py:1642-1671

Pinning: `oracle/make_goldens.py` imports the reference's own `sbgm.special_transforms` in this container and asserts
max-rel 0.0 between these classes and the reference's on the committed fixture `tests/golden/transforms.npz`.
`report_precip_extremes` lives in `sbgm/utils.py`, which cannot be imported here (omegaconf / zarr / netCDF4 absent):
for that one function parity is UNPINNED by the reference; it is three torch calls (flatten, quantile(0.999), max) and a
host loop, restated below line by line.
"""
import torch


def _as_tensor(sample):
    return sample if isinstance(sample, torch.Tensor) else torch.tensor(sample, dtype=torch.float32)


class Scale:                                              # special_transforms.py:62-100
    def __init__(self, in_low, in_high, data_min_in=0, data_max_in=1):
        self.in_low, self.in_high, self.data_min_in, self.data_max_in = in_low, in_high, data_min_in, data_max_in

    def __call__(self, sample):
        old, new = self.data_max_in - self.data_min_in, self.in_high - self.in_low
        return (((sample - self.data_min_in) * new) / old) + self.in_low


class ScaleBackTransform:                                 # :103-139
    def __init__(self, in_low=0, in_high=1, data_min_in=0, data_max_in=1):
        self.in_low, self.in_high, self.data_min_in, self.data_max_in = in_low, in_high, data_min_in, data_max_in

    def __call__(self, sample):
        old, new = self.in_high - self.in_low, self.data_max_in - self.data_min_in
        return (((sample - self.in_low) * new) / old) + self.data_min_in


class _ZBase:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def _stats(self, sample):                             # :166-178 / :209-226: fp32 0-dim tensors broadcast to the sample
        mean = self.mean if isinstance(self.mean, torch.Tensor) else torch.tensor(self.mean, dtype=torch.float32)
        std = self.std if isinstance(self.std, torch.Tensor) else torch.tensor(self.std, dtype=torch.float32)
        while mean.dim() < sample.dim():
            mean, std = mean.unsqueeze(0), std.unsqueeze(0)
        return mean.to(sample.device), std.to(sample.device)


class ZScoreTransform(_ZBase):                            # :143-184
    def __call__(self, sample):
        sample = _as_tensor(sample)
        mean, std = self._stats(sample)
        return (sample - mean) / (std + 1e-8)


class ZScoreBackTransform(_ZBase):                        # :187-233
    def __call__(self, sample):
        sample = _as_tensor(sample)
        mean, std = self._stats(sample)
        return (sample * (std + 1e-8)) + mean


_LOG_TYPES = ("log_zscore", "log_01", "log_minus1_1", "log")


def _check_log_args(scale_type, mean, std, lo, hi):
    if scale_type == "log_zscore" and (mean is None or std is None):
        raise ValueError("Global mean and standard deviation not provided. Using local statistics is not recommended.")
    if scale_type in ("log_01", "log_minus1_1") and (lo is None or hi is None):
        raise ValueError("Min and max log values not provided. Using global statistics is recommended.")
    if scale_type not in _LOG_TYPES:
        raise ValueError("Invalid scale type. Please choose from ['log_01', 'log_zscore', 'log_minus1_1', 'log'].")


class PrcpLogTransform:                                   # :239-355 (log range widened by buffer_frac on EACH side, :262-266)
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

    def __call__(self, sample):
        v = torch.log(_as_tensor(sample) + self.eps)
        if self.scale_type == "log_01":
            denom = self.glob_max_log - self.glob_min_log
            if denom == 0:
                raise ValueError("The log-range of data is zero. Cannot scale to [0, 1]. Please check the data.")
            return (v - self.glob_min_log) / denom
        if self.scale_type == "log_zscore":
            return (v - self.glob_mean_log) / (self.glob_std_log + 1e-8)
        if self.scale_type == "log_minus1_1":
            return 2 * ((v - self.glob_min_log) / (self.glob_max_log - self.glob_min_log)) - 1
        return v


class PrcpLogBackTransform:                               # :360-462 (range widened by buffer_frac/2 per side, :393-399)
    def __init__(self, scale_type="log_zscore", glob_mean_log=None, glob_std_log=None, glob_min_log=None, glob_max_log=None,
                 buffer_frac=0.5, clamp_log_min=None, clamp_log_max=None):
        self.scale_type = scale_type
        self.glob_mean_log, self.glob_std_log = glob_mean_log, glob_std_log
        self.glob_min_log, self.glob_max_log, self.buffer_frac = glob_min_log, glob_max_log, buffer_frac
        self.hi = float("inf") if clamp_log_max is None else float(clamp_log_max)
        self.lo = -float("inf") if clamp_log_min is None else float(clamp_log_min)
        if glob_min_log is not None and glob_max_log is not None:
            rng = glob_max_log - glob_min_log
            self.glob_min_log = glob_min_log - (buffer_frac / 2) * rng
            self.glob_max_log = glob_max_log + (buffer_frac / 2) * rng
        _check_log_args(scale_type, glob_mean_log, glob_std_log, glob_min_log, glob_max_log)

    def __call__(self, sample):
        s = _as_tensor(sample)
        if self.scale_type == "log_01":
            v = s * (self.glob_max_log - self.glob_min_log) + self.glob_min_log
        elif self.scale_type == "log_zscore":
            v = (s * (self.glob_std_log + 1e-8)) + self.glob_mean_log
        elif self.scale_type == "log_minus1_1":
            v = 0.5 * (s + 1) * (self.glob_max_log - self.glob_min_log) + self.glob_min_log
        else:
            v = s
        return torch.exp(torch.clamp(v, self.lo, self.hi))


def report_precip_extremes(x_bt, name, cap_mm_day=500.0, logger=print):     # utils.py:1642-1671
    flat = x_bt.flatten(1)
    p999 = torch.quantile(flat, 0.999, dim=1)
    mx = torch.max(flat, dim=1).values
    n_ex, vals_ex, n_b0, vals_b0 = 0, [], 0, []
    for i, (p, m) in enumerate(zip(p999.tolist(), mx.tolist())):
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


def sample_extremes(x_bt, q=0.999):
    """the two per-sample statistics the sentinel consumes (what the device kernel K32 returns)"""
    flat = x_bt.flatten(1)
    return torch.max(flat, dim=1).values, torch.quantile(flat, q, dim=1)

