"""The transform configurations behind tests/golden/transforms.npz (same table as oracle/make_goldens.py)."""
LOGP = dict(glob_mean_log=-1.2345, glob_std_log=2.0321, glob_min_log=-4.60517, glob_max_log=5.7038, buffer_frac=0.5)


def cases():
    """name -> (factory(module) -> transform, input key in the fixture, input scale)"""
    c = {
        "zscore_back": (lambda M: M.ZScoreBackTransform(8.7012, 6.1923), "z", 1),
        "zscore_fwd": (lambda M: M.ZScoreTransform(8.7012, 6.1923), "phys", 1),
        "scale_back": (lambda M: M.ScaleBackTransform(0, 1, -23.5, 41.25), "z", 1),
        "scale_back_m11": (lambda M: M.ScaleBackTransform(-1, 1, 0.0, 155.3), "z", 1),
        "scale_fwd": (lambda M: M.Scale(-1, 1, -23.5, 41.25), "phys", 1),
        "log_fwd": (lambda M: M.PrcpLogTransform(scale_type="log", **LOGP), "phys", 1),
        "log_back": (lambda M: M.PrcpLogBackTransform(scale_type="log", clamp_log_max=3.0, **LOGP), "z", 1),
    }
    for st in ("log_zscore", "log_01", "log_minus1_1"):
        c[st + "_fwd"] = (lambda M, st=st: M.PrcpLogTransform(scale_type=st, **LOGP), "phys", 1)
        c[st + "_back"] = (lambda M, st=st: M.PrcpLogBackTransform(scale_type=st, **LOGP), "z", 1)
        c[st + "_back_clamped"] = (lambda M, st=st: M.PrcpLogBackTransform(scale_type=st, clamp_log_min=-4.60517,
                                                                          clamp_log_max=5.7038, **LOGP), "z", 3)
    return c


EXACT = {"zscore_back", "zscore_fwd", "scale_back", "scale_back_m11", "scale_fwd"}   # no exp / log: bit-identical
