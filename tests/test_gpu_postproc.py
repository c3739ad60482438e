"""GPU parity of the post-sampling row through the C ABI: transform programs against the fixtures
generated from the reference's sbgm/special_transforms.py (bit-exact for the affine transforms, 1e-6 max-rel where exp/log
are involved: device libm vs the CPU's), the per-sample max / quantile kernel against torch.max / torch.quantile on the
CPU (the oracle of utils.py:1647-1649), and the fused monitor block of the training preview."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from transform_cases import EXACT, cases  # noqa: E402
from util_models import load_golden, maxrel  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(cases()))
def test_transforms_match_reference_fixture(golden_dir, name):
    from sbgm_danra_amd import special_transforms as ST
    g = load_golden(os.path.join(golden_dir, "transforms.npz"))
    mk, key, scale = cases()[name]
    got = mk(ST)((g[key] * scale).cuda()).cpu()
    if name in EXACT:
        assert torch.equal(got, g[name])
    else:
        assert maxrel(got, g[name]) <= 1e-6


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 4099, 1 << 20])
def test_chain_ragged_sizes_and_in_place(n):
    from sbgm_danra_amd import special_transforms as ST
    x = torch.randn(n)
    want = (x * 3.0 + 1.5).clamp(-2.0, 4.0)
    prog = [(ST.MUL, 3.0), (ST.ADD, 1.5)] + ST.clamp_program(-2.0, 4.0)
    xd = x.cuda()
    assert torch.equal(ST.apply_chain(xd, prog).cpu(), want)
    ST.apply_chain(xd, prog, out=xd)
    assert torch.equal(xd.cpu(), want)


def test_chain_keeps_nan_and_rejects_bad_programs():
    from sbgm_danra_amd import special_transforms as ST
    from sbgm_danra_amd._native import NativeError
    x = torch.tensor([float("nan"), -1.0, 7.0, 2.0]).cuda()
    got = ST.apply_chain(x, ST.clamp_program(0.0, 5.0)).cpu()
    assert torch.isnan(got[0]) and got[1:].tolist() == [0.0, 5.0, 2.0]          # torch.clamp semantics
    with pytest.raises(NativeError):
        ST.apply_chain(x, [(99, 0.0)])
    with pytest.raises(ValueError):
        ST.apply_chain(x, [(ST.ADD, 1.0)] * 13)


@pytest.mark.parametrize("B,per,q", [(4, 128 * 128, 0.999), (2, 256 * 256, 0.999), (3, 1000, 0.5), (2, 17, 0.0), (2, 17, 1.0),
                                      (5, 4097, 0.25), (1, 1, 0.999)])
def test_sample_extremes_match_torch(B, per, q):
    from sbgm_danra_amd.special_transforms import sample_extremes
    g = torch.Generator().manual_seed(per + B)
    x = torch.randn(B, per, generator=g) ** 3 * 40.0
    x[0, : per // 3] = x[0, 0]                                   # heavy duplicates
    if per > 100:
        x[-1, 5] = 1e4                                          # a spike
    mx, qq = sample_extremes(x.cuda(), q)
    assert torch.equal(mx.cpu(), x.max(dim=1).values)
    want = torch.quantile(x, q, dim=1)
    assert maxrel(qq.cpu(), want) <= 1e-6


def test_sample_extremes_special_values():
    from sbgm_danra_amd.special_transforms import sample_extremes
    x = torch.randn(3, 4096)
    x[0, 7] = float("nan")
    x[1, :] = -0.0
    x[1, 3] = 0.0
    x[2, 11] = float("inf")
    mx, qq = sample_extremes(x.cuda(), 0.999)
    assert torch.isnan(mx[0]) and torch.isnan(qq[0])              # torch.max / torch.quantile propagate NaN
    assert mx[1].item() == 0.0 and qq[1].item() == 0.0
    assert mx[2].item() == float("inf")
    assert maxrel(qq[2:].cpu(), torch.quantile(x[2:], 0.999, dim=1)) <= 1e-6


def test_report_precip_extremes_matches_oracle():
    from oracle import transforms_ref as OT
    from sbgm_danra_amd.utils import report_precip_extremes
    x = torch.rand(4, 1, 128, 128) * 30
    x[1, 0, 0, 0] = 900.0
    x[2] = -x[2] - 1
    x[3, 0, 5, 5] = 700.0
    x[3] = x[3] - 800.0
    msgs_a, msgs_b = [], []
    a = report_precip_extremes(x.cuda(), "gen", 500.0, logger=msgs_a.append)
    b = OT.report_precip_extremes(x, "gen", 500.0, logger=msgs_b.append)
    assert a == b and msgs_a == msgs_b and a["n_extreme"] == 1 and a["n_below_zero"] == 2


def test_monitor_block_back_transform_sentinel_clamp():
    """training.py:697-748 on the device: back-transform -> sentinel -> clamp fused with the back-transform"""
    from oracle import transforms_ref as OT
    from sbgm_danra_amd import special_transforms as ST
    from sbgm_danra_amd.training import TrainingPipeline_general
    prm = dict(glob_mean_log=-1.0, glob_std_log=2.0, glob_min_log=-4.0, glob_max_log=5.0, buffer_frac=0.5)
    p = TrainingPipeline_general.__new__(TrainingPipeline_general)
    p.extreme_enabled, p.extreme_threshold_mm, p.extreme_clamp_in_gen = True, 500.0, True
    p.back_transforms = {"hr": ST.PrcpLogBackTransform(scale_type="log_zscore", **prm)}
    gen = torch.randn(4, 1, 64, 64) * 0.3
    gen[2, 0, 3, 3] = 5.5                                        # exp(5.5*2 - 1) ~ 22026 mm/day
    mon = {"monitoring": {"extreme_prcp": {"enabled": True, "threshold_mm": 500.0, "clamp_max_mm": 450.0}}}
    out, chk = p.monitor_generated(gen.cuda(), mon)
    want_bt = OT.PrcpLogBackTransform(scale_type="log_zscore", **prm)(gen)
    assert chk["has_extreme"] and chk["n_extreme"] == 1
    assert maxrel(out.cpu(), want_bt.clamp(0.0, 450.0)) <= 1e-6 and out.max().item() == 450.0
    # nothing extreme -> the raw samples come back untouched (the reference only replaces them when it clamps)
    out2, chk2 = p.monitor_generated(gen[:2].cuda(), mon)
    assert chk2 == {"has_extreme": False} and torch.equal(out2.cpu(), gen[:2])


@pytest.mark.parametrize("B,hw,lr_channels,geo,labels,drop", [
    (5, (32, 32), [1, 1, 1], (1, 1), True, [0, 1, 0, 1, 1]),
    (3, (20, 12), [2, 1], (2, 1), True, None),                 # lsm already carries its mask channel: copied through
    (3, (20, 12), [2, 1], (1, 1), True, [1, 0, 0]),            # a 2-channel LR field
    (4, (16, 16), [1], (0, 1), False, None),                   # no dropout (validation split), no lsm, no labels
    (2, (64, 64), [], (1, 0), True, [1, 1]),
])
def test_condition_assembly_matches_oracle(B, hw, lr_channels, geo, labels, drop):
    """device batch assembly == the reference dataset's per-sample dropout / mask logic followed by
    collation and extract_samples (oracle/data_ref.py), bit for bit."""
    from oracle import data_ref as OD
    from sbgm_danra_amd.utils import extract_samples_device
    g = torch.Generator().manual_seed(B * 31 + hw[0])
    names = ["temp", "prcp", "ewvf", "nwvf"]
    items = []
    for b in range(B):
        it = {"prcp_hr": torch.randn(1, *hw, generator=g)}
        for k, c in enumerate(lr_channels):
            it[f"{names[k]}_lr"] = torch.randn(c, *hw, generator=g)
        if geo[0]:
            it["lsm"] = (torch.rand(geo[0], *hw, generator=g) > 0.5).float()
        if geo[1]:
            it["topo"] = torch.rand(geo[1], *hw, generator=g)
        if labels:
            it["classifier"] = torch.randint(1, 5, (), generator=g)
        items.append(it)
    raw = {k: torch.stack([it[k] for it in items]) for k in items[0]}
    guidance = {"enabled": drop is not None}
    done = [OD.finish_sample({k: v.clone() for k, v in it.items()}, "train", guidance, 0.0 if (drop and drop[b]) else 0.9)[0]
            for b, it in enumerate(items)]
    want = OD.extract_samples({k: torch.stack([d[k] for d in done]) for k in done[0]})
    got = extract_samples_device(raw, "cuda", None if drop is None else torch.tensor(drop, dtype=torch.uint8))
    for w, h in zip(want, got):
        assert (w is None) == (h is None)
        if w is not None:
            assert w.shape == h.shape and torch.equal(w, h.cpu())


def test_dropout_draw_follows_reference_rng_order():
    from sbgm_danra_amd.utils import draw_condition_dropout
    torch.manual_seed(123)
    want = [1 if float(torch.rand(())) < 0.1 else 0 for _ in range(64)]
    torch.manual_seed(123)
    got = draw_condition_dropout(64, "train", {"enabled": True, "drop_prob": 0.5})     # the reference ignores drop_prob (:964)
    assert got.tolist() == want and sum(want) > 0
    assert draw_condition_dropout(8, "valid", {"enabled": True}) is None
    assert draw_condition_dropout(8, "train", {"enabled": False}) is None
