"""Test helpers: build the oracle (checker) and the native model with identical hash-generated weights."""
import numpy as np
import torch
import torch.nn as nn

from oracle import torch_ref as O


def build_pair(n_cond, num_classes=None, device="cuda", heads=4, temb=256, layers=(2, 2, 2, 2), norm="group", resize=True):
    import sbgm_danra_amd as S
    ora = O.build_scorenet(n_cond, num_classes=num_classes, time_embedding=temb, n_heads=heads, block_layers=layers, norm=norm,
                           use_resize_conv=resize)
    sd = O.synth_state_dict(ora)
    ora.load_state_dict(sd)
    enc = S.Encoder(n_cond, temb, block_layers=list(layers), n_heads=heads, num_classes=num_classes)
    dec = S.Decoder(512, 1, temb, n_heads=heads, norm=norm, gn_groups=8, activation=nn.SiLU, use_resize_conv=resize)
    net = S.ScoreNet(S.marginal_prob_std_fn, enc, dec, device=torch.device(device), debug_pre_sigma_div=False)
    net.load_state_dict(sd)
    return ora, net, sd


def load_golden(path):
    z = np.load(path)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def maxrel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elemrel(a, b):
    """element-wise relative error with a floor of 1e-3 of the tensor's scale: max |a - b| / (|b| + 1e-3 max|b|).  Beside the global
    norm above it bounds a LOCALISED error on small-magnitude outputs (a score where |s| << max|s|), which max|a-b| / max|b| hides."""
    b = b.to(a.dtype)
    return float(((a - b).abs() / (b.abs() + 1e-3 * b.abs().max().clamp_min(1e-30))).max())


_MEASURED = {}


def check_parity(got, want, tol, label, elem_tol=None):
    """global-norm max-rel <= tol AND element-wise (floored) relative error <= elem_tol (default 1e-2: an output 1000 x smaller than
    the largest is still right to 1 %; fp32 accumulation noise is uniform in ABSOLUTE size, ~2e-6 of the scale, so the measured
    element-wise figure sits near 2e-6 / 1e-3 = 2e-3); both printed and kept in gpurun_out/measured_parity.json"""
    import json
    import os
    g, e = maxrel(got, want), elemrel(got, want)
    print(f"parity[{label}]: global max-rel {g:.2e} (tol {tol:.0e}), element-wise {e:.2e}")
    _MEASURED[label] = {"global": g, "elementwise": e}
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        json.dump(_MEASURED, open(os.path.join(d, "measured_parity.json"), "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    assert g <= tol, (label, g)
    assert e <= (elem_tol if elem_tol is not None else 1e-2), (label, e)
