"""C2 at its full length: 1000 Euler-Maruyama steps at B=32, 128x128, graph replay against eager launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sbgm_danra_amd as S
from bench import build_model        # the bench's model: reference training initialisation, no oracle involved
net = build_model(torch.device('cuda'))
net.eval()
g = torch.Generator().manual_seed(2)
c = torch.randn(32, 1, 128, 128, generator=g).cuda()
kw = dict(batch_size=32, num_steps=1000, device="cuda", img_size=128, cond_img=c, seed=9)
for ug in (True, False, True):
    torch.cuda.synchronize(); t0 = time.time()
    a = S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, use_graph=ug, **kw)
    torch.cuda.synchronize(); print("graph" if ug else "eager", time.time() - t0, "s", "finite", bool(torch.isfinite(a).all()), float(a.abs().max()), flush=True)
    if ug: ga = a
    else: ea = a
print("bit-equal", torch.equal(ga.view(torch.int32), ea.view(torch.int32)))
