"""ms per network evaluation through the PUBLIC samplers (no autotune, as launch_generation runs them), next to bench.py's tuned figure:
C2's shape, then a full-featured model (4 low-res conditions + lsm + topo + 4 season classes)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sbgm_danra_amd as S
import torch.nn as nn


def build(n_in, classes=None):
    """a model with the reference's training initialisation (as bench.build_model; no oracle involved)"""
    enc = S.Encoder(n_in, 256, block_layers=[2, 2, 2, 2], n_heads=4, num_classes=classes)
    dec = S.Decoder(512, 1, 256, n_heads=4, norm="group", gn_groups=8, activation=nn.SiLU)
    net = S.ScoreNet(S.marginal_prob_std_fn, enc, dec, device=torch.device("cuda"), debug_pre_sigma_div=False)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.fill_(0.01)
    return net.eval()


def run(label, net, fn, evals_per_step, steps, **kw):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=steps, device="cuda", **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{label} run {rep}: {dt:.3f} s = {dt / (steps * evals_per_step) * 1e3:.3f} ms per evaluation, finite {bool(torch.isfinite(out).all())}", flush=True)


g = torch.Generator().manual_seed(1)
net = build(1)
c = torch.randn(32, 1, 128, 128, generator=g).cuda()
run("C2 em", net, S.Euler_Maruyama_sampler, 1, 500, batch_size=32, img_size=128, cond_img=c, seed=1)
run("C2 pc", net, S.pc_sampler, 2, 250, batch_size=32, img_size=128, cond_img=c, seed=1)
full = build(8, 4)
if True:
    full.eval()
    c4 = torch.randn(32, 4, 128, 128, generator=g).cuda()
    lsm = torch.cat([(torch.rand(32, 1, 128, 128, generator=g) > 0.5).float(), torch.ones(32, 1, 128, 128)], 1).cuda()
    topo = torch.cat([torch.rand(32, 1, 128, 128, generator=g), torch.ones(32, 1, 128, 128)], 1).cuda()
    y = torch.randint(1, 5, (32,), generator=g).cuda()
    run("full pc", full, S.pc_sampler, 2, 250, batch_size=32, img_size=128, cond_img=c4, lsm_cond=lsm, topo_cond=topo, y=y, seed=1)
