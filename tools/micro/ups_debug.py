import sys, os, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn
import sbgm_danra_amd as S
from sbgm_danra_amd import train_graph as T
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 4
stage = sys.argv[2] if len(sys.argv) > 2 else "fwd"
use_skip = (sys.argv[3] if len(sys.argv) > 3 else "skip") == "skip"
def say(*a):
    print(*a, flush=True)
bn = S.DecoderBlock(128, 64, 128, upsample_scale=scale, activation=nn.SiLU, compute_attn=False, norm="group", gn_groups=8).cuda()
g = torch.Generator().manual_seed(7)
B, h = 3, 6
x, t = torch.randn(B, 128, h, h + 2, generator=g), torch.rand(B, generator=g) * 0.9 + 0.05
skip = torch.randn(B, 64, scale * h, scale * (h + 2), generator=g)
xn = x.cuda().requires_grad_(True)
say("upsample op alone")
xx = torch.randn(B, h, h + 2, 128, device="cuda", requires_grad=True)
y = T.UpsampleFn.apply(xx, scale); torch.cuda.synchronize(); say(" fwd ok", tuple(y.shape))
y.sum().backward(); torch.cuda.synchronize(); say(" bwd ok")
say("block forward")
yn = bn(xn, skip.cuda() if use_skip else None, t.cuda()); torch.cuda.synchronize(); say(" ok", tuple(yn.shape), float(yn.abs().max()))
if stage == "bwd":
    say("block backward")
    yn.square().mean().backward(); torch.cuda.synchronize(); say(" ok", float(xn.grad.abs().max()))
