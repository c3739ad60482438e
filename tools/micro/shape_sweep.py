"""Consistency sweep of the static kernel choice over batch sizes and (non-square) map sizes: row k of a batched evaluation against the
same sample evaluated alone (the two take different kernels: 2-D / LDS Winograd tiles vs small wave tiles), finite outputs, no faults.
python tools/micro/shape_sweep.py"""
import sys, os, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import build_model
dev = torch.device("cuda")
worst = 0.0
for n_cond in (1, 4):
    net = build_model(dev, n_cond=n_cond)
    g = torch.Generator().manual_seed(5)
    for (H, W), B in itertools.product([(32, 32), (64, 64), (64, 96), (96, 160), (128, 128), (160, 96), (192, 256), (256, 256), (224, 128)],
                                       [1, 2, 3, 5, 8, 13, 16, 24, 32, 48]):
        if B * H * W > 32 * 192 * 256:
            continue
        x = torch.randn(B, 1, H, W, generator=g).to(dev) * 3
        c = torch.randn(B, n_cond, H, W, generator=g).to(dev)
        t = (torch.rand(B, generator=g) * 0.9 + 0.05).to(dev)
        with torch.no_grad():
            full = net(x, t, cond_img=c)
            k = B // 2
            solo = net(x[k:k + 1], t[k:k + 1], cond_img=c[k:k + 1])
        torch.cuda.synchronize()
        err = float((full[k:k + 1] - solo).abs().max() / solo.abs().max())
        ok = bool(torch.isfinite(full).all()) and err <= 5e-5
        worst = max(worst, err)
        print(f"C_in={1 + n_cond} B={B:3d} {H}x{W}: batch-vs-solo {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        assert ok
print("worst", worst)
