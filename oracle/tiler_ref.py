"""CPU restatement (NumPy) of the full-domain tiling specification, DESIGN.md §9 / sbgm_danra_amd/csrc/tiling.hip.
TEST INFRASTRUCTURE ONLY — nothing under sbgm_danra_amd/ may import this module.

There is NO reference counterpart: the reference repository contains only the full-domain dimensions
(config/full_run_config_new.yaml:26,28), no tiling / halo / stitching code.  PARITY UNPINNED by construction; this file
pins the device kernels to the written specification, and the sampler inside each tile is covered by the ordinary
per-tile parity tests.
"""
import numpy as np


def extract(domain, origins, tile):
    """domain [C,Hd,Wd], origins [(y0,x0)] -> [T,C,tile,tile]"""
    return np.stack([domain[:, y:y + tile, x:x + tile] for y, x in origins]).astype(np.float32)


def ramp(L, origin, dom_len, R):
    i = np.arange(L)
    lo = np.full(L, R) if origin == 0 else i + 1
    hi = np.full(L, R) if origin + L == dom_len else L - i
    return (np.minimum(np.minimum(lo, hi), R).astype(np.float32) / np.float32(R)).astype(np.float32)


def stitch(tiles, origins, Hd, Wd, R):
    """normalised linear-ramp blend, fp32 accumulation in tile order (the kernel's order)"""
    T, C, th, tw = tiles.shape
    acc = np.zeros((C, Hd, Wd), np.float32)
    wsum = np.zeros((Hd, Wd), np.float32)
    for t, (y, x) in enumerate(origins):
        w = (ramp(th, y, Hd, R)[:, None] * ramp(tw, x, Wd, R)[None, :]).astype(np.float32)
        acc[:, y:y + th, x:x + tw] += w[None] * tiles[t]
        wsum[y:y + th, x:x + tw] += w
    return np.where(wsum > 0, acc / np.where(wsum > 0, wsum, 1), 0).astype(np.float32)
