"""The loss around the network on the GPU: csrc/dsm_loss.hip (perturbation, weighted squared-error reduction, backward)
against the same expressions in PyTorch-CPU, `S.loss_fn` ITSELF (not a re-implementation) against the reference's golden loss
and gradients with the golden's (t, z) injected, and the in-kernel noise: repeatable under torch.manual_seed, in range, N(0,1)
moments, fresh on every replay of a captured step.  Reference: sbgm/score_unet.py:936-985."""
import ctypes as C
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from sbgm_danra_amd import _native as N  # noqa: E402
from util_models import build_pair, load_golden, maxrel  # noqa: E402


def _std(t, sigma=25.0):
    ls = math.log(sigma)
    return torch.sqrt((torch.exp(2 * t * ls) - 1) / (2 * ls)).clamp_min(1e-5)


@pytest.mark.parametrize("B,H,W,with_sdf", [(3, 32, 32, True), (2, 64, 32, False), (8, 128, 128, True), (1, 2, 2, False)])
def test_dsm_kernels_match_the_reference_expressions(B, H, W, with_sdf):
    """perturb (:963), loss (:974-984) and d loss / d score, on injected (t, z)"""
    import sbgm_danra_amd.score_unet as SU
    g = torch.Generator().manual_seed(B * H + W)
    x, z, score = (torch.randn(B, 1, H, W, generator=g) for _ in range(3))
    t = torch.rand(B, generator=g) * 0.999 + 1e-3
    sdf = torch.rand(B, 1, H, W, generator=g) * 4 - 2 if with_sdf else None
    std = _std(t)
    lib, per = N.lib(), H * W
    xd, zd, td = x.cuda(), z.cuda(), t.cuda()
    xp, t_out, std_out = torch.empty_like(xd), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    N.check(lib.sbgm_dsm_perturb(xd.data_ptr(), zd.data_ptr(), td.data_ptr(), None, 0, 1e-3, 25.0, xp.data_ptr(), None, t_out.data_ptr(),
                                 std_out.data_ptr(), B, per, N.stream()))
    assert torch.equal(t_out.cpu(), t) and maxrel(std_out.cpu(), std) < 1e-6
    assert maxrel(xp.cpu(), x + std[:, None, None, None] * z) < 1e-6
    # loss + backward through the autograd Function loss_fn uses
    sr = score.clone().requires_grad_(True)
    w = torch.sigmoid(sdf) * 0.5 + 0.5 if with_sdf else torch.ones_like(x)
    want = torch.mean(torch.sum(w * (sr * std[:, None, None, None] + z) ** 2, dim=(1, 2, 3)))
    (want * 0.7).backward()
    sd_ = score.cuda().requires_grad_(True)
    got = SU._DSMLossFn.apply(sd_, zd, std_out, None if sdf is None else sdf.cuda(), None)
    (got * 0.7).backward()
    assert abs(float(got) / float(want) - 1) < 1e-6
    assert maxrel(sd_.grad.cpu(), sr.grad) < 1e-6


def test_loss_fn_itself_matches_reference_goldens(golden_dir):
    """S.loss_fn fed the golden's (t, z) through noise=: loss value (sdf-weighted) and the 7 probe gradients recorded from the
    reference's loss_fn + backward (tests/golden/loss_b2_64.npz)"""
    import sbgm_danra_amd as S
    g = load_golden(os.path.join(golden_dir, "loss_b2_64.npz"))
    _, net, _ = build_pair(1)
    net.train()
    x, cond, sdf, t, z = (g[k].cuda() for k in ("x", "cond_img", "sdf", "t", "z"))
    loss = S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=cond, sdf_cond=sdf, noise=(t, z))
    loss.backward()
    assert abs(float(loss) / float(g["loss"]) - 1) < 1e-5
    params = dict(net.named_parameters())
    worst = {}
    for k in [k for k in g if k.startswith("grad::")]:
        gr = params[k[6:]].grad.reshape(-1).cpu()
        worst[k] = maxrel(gr[:: max(1, gr.numel() // 2048)][:2048], g[k])
    print("loss_fn probe-gradient max-rel:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert max(worst.values()) < 1e-4, worst
    # without the sdf weight the loss differs (the weight is live), and the no-grad path (validation) gives the same value
    with torch.no_grad():
        net.eval()                              # running-stat BatchNorm: a different network function, only finiteness + determinism
        a = S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=cond, sdf_cond=sdf, noise=(t, z))
        b = S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=cond, sdf_cond=sdf, noise=(t, z))
        c = S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=cond, noise=(t, z))
    assert torch.isfinite(a) and float(a) == float(b) and float(c) != float(a)


def test_loss_fn_argument_checks():
    import functools
    import sbgm_danra_amd as S
    _, net, _ = build_pair(1)
    x = torch.randn(2, 1, 32, 32).cuda()
    with pytest.raises(ValueError):
        S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=torch.randn(3, 1, 32, 32).cuda())        # reference :965-967
    with pytest.raises(N.NativeError):
        S.loss_fn(net, x.cpu(), S.marginal_prob_std_fn, cond_img=torch.randn(2, 1, 32, 32))
    # another sigma is honoured (perturbation and loss use the schedule's own sigma)
    fn10 = functools.partial(S.marginal_prob_std, sigma=10.0)
    t, z = torch.tensor([0.3, 0.9]).cuda(), torch.randn(2, 1, 32, 32).cuda()
    seen = {}

    def model(xp, tt, **kw):
        seen["xp"] = xp
        return torch.zeros_like(xp)
    S.loss_fn(model, x, fn10, noise=(t, z))
    assert maxrel(seen["xp"].cpu(), (x + fn10(t)[:, None, None, None] * z).cpu()) < 1e-6


def test_loss_fn_takes_any_marginal_prob_std_callable():
    """reference loss_fn accepts an arbitrary marginal_prob_std (score_unet.py:936-985): a VP-like schedule here.  Perturbation, loss
    value and d loss / d score against the reference expressions in PyTorch, with injected (t, z) and with in-kernel draws (the
    t that was drawn is the t the callable saw)."""
    import sbgm_danra_amd as S
    sched = lambda t: torch.sqrt(1.0 - torch.exp(-0.1 * t - 9.95 * t * t))          # noqa: E731
    g = torch.Generator().manual_seed(3)
    x, z = torch.randn(3, 1, 32, 32, generator=g).cuda(), torch.randn(3, 1, 32, 32, generator=g).cuda()
    t = (torch.rand(3, generator=g) * 0.9 + 0.05).cuda()
    seen = {}

    def model(xp, tt, **kw):
        seen["xp"], seen["t"] = xp, tt
        s = (xp * 0.3).requires_grad_(True)
        seen["s"] = s
        return s
    loss = S.loss_fn(model, x, sched, noise=(t, z))
    loss.backward()
    std = sched(t)[:, None, None, None]
    assert torch.equal(seen["t"], t) and maxrel(seen["xp"].cpu(), (x + std * z).cpu()) < 1e-6
    sr = (seen["xp"].detach() * 0.3).requires_grad_(True)
    want = torch.mean(torch.sum((sr * std + z) ** 2, dim=(1, 2, 3)))
    want.backward()
    assert abs(float(loss) / float(want) - 1) < 1e-6 and maxrel(seen["s"].grad.cpu(), sr.grad.cpu()) < 1e-6
    torch.manual_seed(11)
    S.loss_fn(model, x, sched)                                                       # draws inside the kernel
    t2, xp2 = seen["t"].clone(), seen["xp"].clone()
    assert float(t2.min()) >= 1e-3 and float(t2.max()) <= 1.0
    zz = (xp2 - x) / sched(t2)[:, None, None, None]
    assert abs(float(zz.mean())) < 0.1 and abs(float(zz.std()) - 1) < 0.1
    torch.manual_seed(11)
    S.loss_fn(model, x, sched)
    assert torch.equal(seen["t"], t2) and torch.equal(seen["xp"], xp2)               # repeatable under torch.manual_seed


def test_gradients_through_an_eval_mode_network(golden_dir):
    """reference ScoreNet.forward is differentiable in any mode (score_unet.py:829-879): in eval() BatchNorm uses its running statistics
    (dx = gamma * rstd * g).  Loss and every parameter gradient against autograd through the CPU oracle in eval mode."""
    import sbgm_danra_amd as S
    from oracle import torch_ref as O
    ora, net, _ = build_pair(1)
    ora.eval()
    net.eval()
    g = torch.Generator().manual_seed(9)
    x, c = torch.randn(2, 1, 64, 64, generator=g), torch.randn(2, 1, 64, 64, generator=g)
    t, z = torch.rand(2, generator=g) * 0.8 + 0.1, torch.randn(2, 1, 64, 64, generator=g)
    want = O.loss_fn(ora, x, O.marginal_prob_std_fn, cond_img=c, noise=(t, z))
    want.backward()
    loss = S.loss_fn(net, x.cuda(), S.marginal_prob_std_fn, cond_img=c.cuda(), noise=(t.cuda(), z.cuda()))
    loss.backward()
    assert abs(float(loss) / float(want) - 1) < 1e-5
    ref = dict(ora.named_parameters())
    errs = {k: maxrel(p.grad.cpu(), ref[k].grad) for k, p in net.named_parameters() if ref[k].grad is not None}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print("eval-mode gradients, worst max-rel:", worst)
    assert len(errs) >= 160 and worst[0][1] < 1e-4, worst
    assert int(net.encoder.bn1.num_batches_tracked) == int(ora.encoder.bn1.num_batches_tracked)       # running statistics untouched


def test_in_kernel_noise_is_seeded_in_range_and_normal():
    import sbgm_danra_amd as S
    lib = N.lib()
    B, H = 64, 64
    x = torch.zeros(B, 1, H, H, device="cuda")
    outs = []
    for seed in (7, 7, 8):
        xp, z = torch.empty_like(x), torch.empty_like(x)
        t, std = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
        N.check(lib.sbgm_dsm_perturb(x.data_ptr(), None, None, None, seed, 1e-3, 25.0, xp.data_ptr(), z.data_ptr(), t.data_ptr(),
                                     std.data_ptr(), B, H * H, N.stream()))
        outs.append((t.cpu(), z.cpu(), xp.cpu(), std.cpu()))
    (t0, z0, xp0, std0), (t1, z1, _, _), (t2, z2, _, _) = outs
    assert torch.equal(t0, t1) and torch.equal(z0, z1) and not torch.equal(t0, t2) and not torch.equal(z0, z2)
    assert float(t0.min()) >= 1e-3 and float(t0.max()) <= 1.0 and 0.3 < float(t0.mean()) < 0.7
    assert abs(float(z0.mean())) < 0.01 and abs(float(z0.var()) - 1) < 0.02 and abs(float((z0 ** 4).mean()) - 3) < 0.15
    assert maxrel(xp0, std0[:, None, None, None] * z0) < 1e-6
    # loss_fn: same torch seed -> same loss; the draw order is t then z as in the reference (:957-959)
    _, net, _ = build_pair(1)
    net.eval()
    xx, cond = torch.randn(2, 1, 32, 32).cuda(), torch.randn(2, 1, 32, 32).cuda()
    vals = []
    with torch.no_grad():
        for seed in (3, 3, 4):
            torch.manual_seed(seed)
            vals.append(float(S.loss_fn(net, xx, S.marginal_prob_std_fn, cond_img=cond)))
    assert vals[0] == vals[1] and vals[0] != vals[2]


def test_captured_loss_draws_fresh_noise_on_every_replay():
    import sbgm_danra_amd as S
    import sbgm_danra_amd.score_unet as SU
    x = torch.randn(2, 1, 32, 32).cuda()
    seen = []

    def model(xp, tt, **kw):
        seen.append((xp, tt))
        return torch.zeros_like(xp)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        S.loss_fn(model, x, S.marginal_prob_std_fn)
    torch.cuda.current_stream().wait_stream(side)
    seen.clear()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = S.loss_fn(model, x, S.marginal_prob_std_fn)
    st = SU._loss_rng_state(x.device)
    off0 = int(st[1])
    res = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        res.append((float(loss), seen[0][0].clone(), seen[0][1].clone()))
    assert int(st[1]) == off0 + 3
    assert len({r[0] for r in res}) == 3 and not torch.equal(res[0][1], res[1][1]) and not torch.equal(res[0][2], res[1][2])
