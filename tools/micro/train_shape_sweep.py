"""Fault / finiteness sweep of the training path (loss_fn + backward, eager) over odd batch sizes and non-square maps: every gradient
finite, the loss repeatable with injected noise.  python tools/micro/train_shape_sweep.py"""
import sys, os, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sbgm_danra_amd as S
from bench import build_model
dev = torch.device("cuda")
for n_cond in (1, 4):
    net = build_model(dev, n_cond=n_cond)
    net.train()
    g = torch.Generator().manual_seed(6)
    for (H, W), B in itertools.product([(64, 64), (96, 96), (64, 96), (96, 128), (160, 96), (128, 128)], [1, 2, 3, 5, 8]):
        x = torch.randn(B, 1, H, W, generator=g).to(dev)
        c = torch.randn(B, n_cond, H, W, generator=g).to(dev)
        t = (torch.rand(B, generator=g) * 0.9 + 0.05).to(dev)
        z = torch.randn(B, 1, H, W, generator=g).to(dev)
        vals = []
        for rep in range(2):
            for p in net.parameters():
                p.grad = None
            loss = S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=c, noise=(t, z))
            loss.backward()
            torch.cuda.synchronize()
            vals.append(float(loss))
            assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in net.parameters()), (B, H, W)
        rel = abs(vals[0] - vals[1]) / max(abs(vals[0]), 1e-30)
        print(f"C_in={1 + n_cond} B={B} {H}x{W}: loss {vals[0]:.6g}, repeat rel diff {rel:.1e}", flush=True)
        assert rel < 1e-2, vals          # (train-mode BatchNorm moves its running statistics between the two calls; the loss does not use them)
print("done")
