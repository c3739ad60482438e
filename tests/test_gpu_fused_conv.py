"""The decoder's fused convolution input modes (csrc/conv_lds.hip) through the C ABI, against plain PyTorch-CPU ops:

  mode 1   conv3x3(GroupNorm(x))                                   = DecoderBlock.conv(norm1(.))       reference score_unet.py:585-589
  mode 2   conv3x3(Upsample_x2_bilinear(act(GroupNorm(x) + skip + tbias)))
           = next block's conv_up(upsample(.)) over the previous block's norm2 / skip / time / activation   :583-584, :592-615

for every LDS-staged Winograd tile (single- and double-buffered), ragged tile rows, C = 64 / 128 / 256, and the whole engine with
the fused decoder against the separate-pass decoder (SBGM_NO_FUSED_DECODER) — the oracle parity of the fused engine itself is
what every test in test_gpu_model.py now exercises.  Tolerance 2e-5 max-rel per convolution (fp32, Winograd F(2,3))."""
import ctypes as C
import math
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from sbgm_danra_amd import _native as N  # noqa: E402

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed + sum(shape))) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max())


def gn_affine(x_nhwc_dev, gamma, beta, tbias, G):
    """stats + finalize through the C ABI -> [B][C][2] table"""
    lib = N.lib()
    B, H, W, Cc = x_nhwc_dev.shape
    ws = torch.empty(1024 * B * G, dtype=torch.uint8, device=DEV)
    chunks = C.c_int(0)
    N.check(lib.sbgm_groupnorm_stats(x_nhwc_dev.data_ptr(), ws.data_ptr(), B, H * W, Cc, G, C.byref(chunks), N.stream()))
    out = torch.empty(B * Cc * 2, device=DEV)
    N.check(lib.sbgm_groupnorm_finalize(ws.data_ptr(), chunks.value, N.ptr(gamma), N.ptr(beta), N.ptr(tbias), out.data_ptr(), B, H * W, Cc, G,
                                        1e-5, N.stream()))
    return out


def fused_conv(x_nhwc_dev, w, bias, H, W, tile, db, in_mode, affine=None, skip=None, act=N.NONE):
    """tile = (co fragments, rows per wave): the 1-D Winograd LDS kernel; tile = ("2d", co fragments, min waves per SIMD): conv_w2d.hip"""
    lib = N.lib()
    B, Cin = x_nhwc_dev.shape[0], x_nhwc_dev.shape[3]
    Cout = w.shape[0]
    wd = w.contiguous().to(DEV)
    w2d = tile[0] == "2d"
    packed = torch.empty((lib.sbgm_conv_wino2d_packed_numel if w2d else lib.sbgm_conv_wino_packed_numel)(Cout, Cin), device=DEV)
    N.check((lib.sbgm_conv_wino2d_pack_weight if w2d else lib.sbgm_conv_wino_pack_weight)(wd.data_ptr(), packed.data_ptr(), Cout, Cin, Cin, N.stream()))
    out = torch.empty(B, H, W, Cout, device=DEV)
    bd = bias.to(DEV)
    tco, tpx, wpt, bits = (tile[1], 0, 0 if tile[2] == "p" else tile[2], 8 | (16 if tile[2] == "p" else 0)) if w2d else (tile[0], tile[1], 0, 3)
    if w2d and tile[2] == "p":
        db = False
    a = N.ConvArgs(x_nhwc_dev.data_ptr(), packed.data_ptr(), out.data_ptr(), None, bd.data_ptr(), None, None, B, H, W, Cin, Cout, 3, 3, 1, 1,
                   N.NONE, 0, tco, tpx, 0, wpt, bits | (4 if db else 0), 0, 0, 0, None, 0, in_mode, N.ptr(affine), N.ptr(skip), act)
    N.check(lib.sbgm_conv2d_fwd(C.byref(a), N.stream()))
    torch.cuda.synchronize()
    return nchw(out.cpu())


TILES = [(4, 1), (4, 2), (2, 1), (2, 2), (1, 1), (1, 2), ("2d", 1, 1), ("2d", 2, 1), ("2d", 2, 2), ("2d", 1, "p"), ("2d", 2, "p")]
ACT = {N.NONE: lambda v: v, N.SILU: F.silu, N.RELU: F.relu, N.GELU: F.gelu}


@pytest.mark.parametrize("shape", [(2, 64, 32, 32, 64, 8), (1, 128, 24, 16, 64, 8), (2, 256, 16, 16, 128, 8), (1, 64, 8, 48, 64, 64)])
@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("db", [False, True])
def test_groupnorm_affine_on_load(shape, tile, db):
    B, Cin, H, W, Cout, G = shape
    if Cout % (16 * (tile[1] if tile[0] == "2d" else tile[0])):
        pytest.skip("tile wider than Cout")
    affine_params = G != Cin                       # G == C: InstanceNorm2d, no affine (reference default norm)
    x, w, b = rnd(B, Cin, H, W) * 1.7 + 0.4, rnd(Cout, Cin, 3, 3, seed=1, scale=1.0 / math.sqrt(Cin * 9)), rnd(Cout, seed=2)
    gamma, beta = (rnd(Cin, seed=3) * 0.3 + 1.0, rnd(Cin, seed=4) * 0.2) if affine_params else (None, None)
    want = F.conv2d(F.group_norm(x, G, gamma, beta, 1e-5), w, b, padding=1)
    xd = nhwc(x).to(DEV)
    aff = gn_affine(xd, None if gamma is None else gamma.to(DEV), None if beta is None else beta.to(DEV), None, G)
    got = fused_conv(xd, w, b, H, W, tile, db, 1, affine=aff)
    assert relerr(got, want) < 2e-5


@pytest.mark.parametrize("shape", [(2, 64, 32, 32, 64), (1, 128, 24, 16, 64), (2, 256, 16, 16, 128), (1, 64, 8, 48, 64), (3, 64, 40, 16, 64)])
@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("db", [False, True])
@pytest.mark.parametrize("pre", ["plain", "norm+skip+silu"])
def test_bilinear_upsample_on_load(shape, tile, db, pre):
    B, Cin, H, W, Cout = shape                      # H, W: the convolution's (high-resolution) size
    if Cout % (16 * (tile[1] if tile[0] == "2d" else tile[0])):
        pytest.skip("tile wider than Cout")
    h, w_ = H // 2, W // 2
    x, w, b = rnd(B, Cin, h, w_) * 1.3 - 0.2, rnd(Cout, Cin, 3, 3, seed=1, scale=1.0 / math.sqrt(Cin * 9)), rnd(Cout, seed=2)
    xd = nhwc(x).to(DEV)
    if pre == "plain":
        low, aff, skip, act = x, None, None, N.NONE
    else:
        gamma, beta, tb, sk = rnd(Cin, seed=3) * 0.3 + 1.0, rnd(Cin, seed=4) * 0.2, rnd(B, Cin, seed=5) * 0.5, rnd(B, Cin, h, w_, seed=6)
        low = F.silu(F.group_norm(x, 8, gamma, beta, 1e-5) + sk + tb[:, :, None, None])
        aff = gn_affine(xd, gamma.to(DEV), beta.to(DEV), tb.to(DEV), 8)
        skip, act = nhwc(sk).to(DEV), N.SILU
    want = F.conv2d(F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=False), w, b, padding=1)
    try:
        got = fused_conv(xd, w, b, H, W, tile, db, 2, affine=aff, skip=skip, act=act)
    except N.NativeError as e:
        if "bytes of LDS" in str(e) and db and tuple(tile) == (4, 2):       # the widest tile, double-buffered: 2 x 84 KB of stages
            pytest.skip(str(e))
        raise
    assert relerr(got, want) < 2e-5


def test_fused_mode_argument_checks():
    x = torch.zeros(1, 16, 16, 64, device=DEV)
    w, b = torch.zeros(64, 64, 3, 3), torch.zeros(64)
    with pytest.raises(N.NativeError):                          # mode 1 without the affine table
        fused_conv(x, w, b, 16, 16, (2, 1), False, 1)
    with pytest.raises(N.NativeError):                          # skip without upsample mode
        fused_conv(x, w, b, 16, 16, (2, 1), False, 1, affine=torch.zeros(128, device=DEV), skip=x)


_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from util_models import build_pair
import sbgm_danra_amd as S
_, net, _ = build_pair(1)
net.eval()
g = torch.Generator().manual_seed(3)
x, c, t = torch.randn(2, 1, 128, 128, generator=g) * 8, torch.randn(2, 1, 128, 128, generator=g), torch.tensor([0.2, 0.9])
with torch.no_grad():
    y = net(x.cuda(), t.cuda(), cond_img=c.cuda()).cpu()
    s = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=2, num_steps=3, device="cuda", img_size=128, cond_img=c.cuda(), seed=5).cpu()
torch.save({"y": y, "s": s}, sys.argv[2])
"""


def test_fused_decoder_equals_separate_passes(tmp_path):
    """same network, same inputs, two processes: default (fused decoder) vs SBGM_NO_FUSED_DECODER=1 (GroupNorm-apply and upsample as
    passes of their own, the round-1 engine).  They differ only in rounding (x*scale + shift vs (x - mean)*rstd*gamma + beta)."""
    outs = {}
    for tag, env in (("fused", {}), ("separate", {"SBGM_NO_FUSED_DECODER": "1"})):
        path = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, "-c", _CHILD, ROOT, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = torch.load(path, weights_only=True)
    assert relerr(outs["fused"]["y"], outs["separate"]["y"]) < 1e-5
    assert relerr(outs["fused"]["s"], outs["separate"]["s"]) < 1e-4
