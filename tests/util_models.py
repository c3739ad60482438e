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
