"""CPU checks of the post-sampling row:
* the oracle restatement (oracle/transforms_ref.py) reproduces the fixtures generated from the reference's own
  sbgm/special_transforms.py bit for bit;
* the programs the device-side mirror (sbgm_danra_amd/special_transforms.py) compiles itself into, interpreted here with
  NumPy fp32 scalar ops in program order, reproduce the same fixtures (bit-exact for the affine transforms; exp/log within
  2 ulp-ish 1e-6) — i.e. the host logic is right without needing a GPU;
* constructor validation errors match the reference's.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from transform_cases import EXACT, cases  # noqa: E402
from util_models import load_golden, maxrel  # noqa: E402

from oracle import transforms_ref as OT  # noqa: E402
from sbgm_danra_amd import special_transforms as ST  # noqa: E402


def interpret(program, x):
    v = x.astype(np.float32)
    for op, c in program:
        c = np.float32(c)
        if op == ST.ADD:
            v = (v + c).astype(np.float32)
        elif op == ST.MUL:
            v = (v * c).astype(np.float32)
        elif op == ST.DIV:
            v = (v / c).astype(np.float32)
        elif op == ST.CLAMP_MIN:
            v = np.where(v < c, c, v)
        elif op == ST.CLAMP_MAX:
            v = np.where(v > c, c, v)
        elif op == ST.EXP:
            v = np.exp(v.astype(np.float64)).astype(np.float32)
        elif op == ST.LOG:
            v = np.log(v.astype(np.float64)).astype(np.float32)
    return v


@pytest.mark.parametrize("name", sorted(cases()))
def test_oracle_and_programs_match_reference_fixture(golden_dir, name):
    g = load_golden(os.path.join(golden_dir, "transforms.npz"))
    mk, key, scale = cases()[name]
    x = g[key] * scale
    assert torch.equal(mk(OT)(x.clone()), g[name])
    got = torch.from_numpy(interpret(mk(ST).program(), x.numpy()))
    if name in EXACT:
        assert torch.equal(got, g[name])
    else:
        assert maxrel(got, g[name]) <= 1e-6


def test_constructor_errors_match_reference():
    for M in (OT, ST):
        with pytest.raises(ValueError, match="Global mean and standard deviation not provided"):
            M.PrcpLogBackTransform(scale_type="log_zscore")
        with pytest.raises(ValueError, match="Min and max log values not provided"):
            M.PrcpLogTransform(scale_type="log_01", glob_mean_log=0.0, glob_std_log=1.0)
        with pytest.raises(ValueError, match="Invalid scale type"):
            M.PrcpLogBackTransform(scale_type="sqrt", glob_mean_log=0.0, glob_std_log=1.0)


def test_back_transform_factories(tmp_path):
    prm = dict(glob_mean_log=-1.0, glob_std_log=2.0, glob_min_log=-4.0, glob_max_log=5.0, buffer_frac=0.5, clamp_log_max=5.0)
    bt = ST.build_back_transforms("prcp", "log_zscore", prm, ["temp", "prcp"], ["zscore", "01"],
                                  [dict(glob_mean=280.0, glob_std=9.0), dict(glob_min=0.0, glob_max=120.0)])
    assert set(bt) == {"prcp_hr", "generated", "temp_lr", "prcp_lr"} and bt["generated"] is bt["prcp_hr"]
    assert bt["generated"].program()[-1][0] == ST.EXP and bt["generated"].program()[-2] == (ST.CLAMP_MAX, 5.0)
    with pytest.raises(ValueError, match="Unknown HR scaling method"):
        ST.build_back_transforms("prcp", "sqrt", prm, [], [], [])
    # stats-file route: file layout of load_global_stats (special_transforms.py:576-592)
    d = tmp_path / "DANRA" / "prcp" / "train"
    d.mkdir(parents=True)
    stats = dict(mean=1.0, std=2.0, min=0.0, max=9.0, log_mean=-1.0, log_std=2.0, log_min=-4.0, log_max=5.0)
    (d / "global_stats__DANRA__589x789__crop__full__prcp__train.json").write_text(__import__("json").dumps(stats))
    t = ST.get_backtransforms_from_stats("prcp", "DANRA", "589x789", "full", "train", "log_zscore", 0.5,
                                         stats_file_path=str(tmp_path))
    assert isinstance(t, ST.PrcpLogBackTransform) and t.lo == -4.0 and t.hi == 5.0
    f = ST.get_transforms_from_stats("prcp", "DANRA", "589x789", "full", "train", "scale_minus1_1", 0.5, stats=stats)
    assert isinstance(f, ST.Scale) and (f.in_low, f.in_high, f.data_max_in) == (-1, 1, 9.0)
    with pytest.raises(ValueError, match="Failed to load stats"):
        ST.get_backtransforms_from_stats("prcp", "DANRA", "x", "y", "train", "zscore", 0.5)
    fused = ST.fuse(t, ST.clamp_program(0.0, 500.0))
    assert fused.program()[-2:] == [(ST.CLAMP_MIN, 0.0), (ST.CLAMP_MAX, 500.0)]


def test_device_only():
    from sbgm_danra_amd._native import NativeError
    with pytest.raises(NativeError):
        ST.ZScoreBackTransform(0.0, 1.0)(torch.zeros(4))


def test_oracle_sentinel_dict_shapes():
    x = torch.rand(3, 1, 64, 64)
    x[1, 0, 0, 0] = 900.0
    x[2] = -x[2] - 1
    out = OT.report_precip_extremes(x, "t", 500.0, logger=lambda *_: None)
    assert out["has_extreme"] and out["n_extreme"] == 1 and out["n_below_zero"] == 1
    assert OT.report_precip_extremes(x[:1], "t", 500.0, logger=lambda *_: None) == {"has_extreme": False}
    only_neg = OT.report_precip_extremes(x[2:], "t", 500.0, logger=lambda *_: None)
    assert "has_extreme" not in only_neg and only_neg["has_below_zero"]
