"""hipGraph replay of the training path must be idempotent: every native op (forward + backward) and the whole loss_fn step
give the SAME gradients on the first, second and third replay as an eager run, with eager steps in between.

Why this file exists: a `hipMemsetAsync` of 196 KB captured into a graph (attention backward zeroing dqkv) left garbage from the
second replay on (ROCm 7.2) while the first replay — the only one the older tests looked at — was right; graph-mode training
was silently wrong.  All launchers now zero their scratch with a kernel (sbgm_zero_async)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from sbgm_danra_amd import _native as N  # noqa: E402
from sbgm_danra_amd import train_graph as T  # noqa: E402
from util_models import build_pair  # noqa: E402


def rnd(*s, seed=0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed + sum(s))).cuda()


def leaf(t):
    return t.clone().requires_grad_(True)


B, H, Cc = 2, 16, 64
_T2 = None


def _t2():
    global _T2
    if _T2 is None:
        _T2 = torch.tensor([0.1, 0.7], device="cuda")      # created outside any capture (host -> device copy)
    return _T2


OPS = {
    "conv3x3+bias+res+tbias": lambda: ([leaf(rnd(B, H, H, Cc)), leaf(rnd(Cc, Cc, 3, 3, seed=1) * 0.05), leaf(rnd(Cc, seed=2)),
                                        leaf(rnd(B, H, H, Cc, seed=3)), leaf(rnd(B, Cc, seed=4))],
                                       lambda x, w, b, r, tb: T.ConvFn.apply(x, w, b, r, tb, 1, 1)),
    "conv3x3 stride 2": lambda: ([leaf(rnd(B, H, H, Cc)), leaf(rnd(128, Cc, 3, 3, seed=1) * 0.05)],
                                 lambda x, w: T.ConvFn.apply(x, w, None, None, None, 2, 1)),
    "linear": lambda: ([leaf(rnd(1, 1, 512, 128)), leaf(rnd(128, 128, 1, 1, seed=1) * 0.05), leaf(rnd(128, seed=2))],
                       lambda x, w, b: T.ConvFn.apply(x, w, b, None, None, 1, 0)),
    "stem conv 8x8 stride 2": lambda: ([leaf(rnd(B, 32, 32, 64)), leaf(rnd(64, 64, 8, 8, seed=1) * 0.02)],
                                       lambda x, w: T.ConvFn.apply(x, w, None, None, None, 2, 3)),
    "batchnorm": lambda: ([leaf(rnd(B, H, H, Cc)), leaf(rnd(Cc, seed=1)), leaf(rnd(Cc, seed=2)), leaf(rnd(B, H, H, Cc, seed=3)),
                           leaf(rnd(B, Cc, seed=4))],
                          lambda x, g, b, r, tb: T.BNTrainFn.apply(x, g, b, torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda"), r, tb,
                                                                   True, 1e-5, 0.1)),
    "groupnorm": lambda: ([leaf(rnd(B, H, H, Cc)), leaf(rnd(Cc, seed=1)), leaf(rnd(Cc, seed=2)), leaf(rnd(B, H, H, Cc, seed=3)),
                           leaf(rnd(B, Cc, seed=4))], lambda x, g, b, s, tb: T.GroupNormFn.apply(x, g, b, s, tb, N.SILU, 8, 1e-5)),
    "layernorm": lambda: ([leaf(rnd(512, 128)), leaf(rnd(128, seed=1)), leaf(rnd(128, seed=2))], lambda x, g, b: T.LayerNormFn.apply(x, g, b, 1e-5)),
    "attention core S=64": lambda: ([leaf(rnd(2 * 64, 3 * 128))], lambda q: T.MHACoreFn.apply(q, 2, 64, 128, 4)),
    "attention core S=256": lambda: ([leaf(rnd(2 * 256, 3 * 128))], lambda q: T.MHACoreFn.apply(q, 2, 256, 128, 4)),
    "upsample": lambda: ([leaf(rnd(B, H, H, Cc))], lambda x: T.UpsampleFn.apply(x)),
    "gelu": lambda: ([leaf(rnd(512, 128))], lambda x: T.ActFn.apply(x, N.GELU)),
    "final conv": lambda: ([leaf(rnd(B, H, H, Cc)), leaf(rnd(1, Cc, 3, 3, seed=1) * 0.05), leaf(rnd(1, seed=2))],
                           lambda a, w, b: T.Cout1Fn.apply(a, w, b, _t2(), 25.0)),
}


def _capture(step, leaves):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = [l.grad.clone() for l in leaves]
    for l in leaves:
        l.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(zero=False)
    return g, ref, [l.grad for l in leaves]


@pytest.mark.parametrize("name", list(OPS))
def test_every_training_op_replays_identically(name):
    leaves, fn = OPS[name]()
    dev = leaves[0].device
    _t2()

    def step(zero=True):
        if zero:
            for l in leaves:
                l.grad = None
        T._zero_reset(dev)
        out = fn(*leaves)
        out.backward(torch.ones_like(out))
    g, ref, static = _capture(step, leaves)
    for r in range(3):
        g.replay()
        torch.cuda.synchronize()
        err = max(float((s - q).abs().max() / q.abs().max().clamp_min(1e-30)) for s, q in zip(static, ref))
        assert err < 1e-5, (name, r, err)        # fp32 atomics reorder sums: not bit-exact, but never garbage
        if r == 1:
            step()                               # an eager step between replays must not disturb the graph


def test_whole_training_step_replays_identically():
    import sbgm_danra_amd as S
    _, net, _ = build_pair(1)
    net.train()
    gen = torch.Generator().manual_seed(1)
    x, cond = torch.randn(2, 1, 64, 64, generator=gen).cuda(), torch.randn(2, 1, 64, 64, generator=gen).cuda()
    t, z = (torch.rand(2, generator=gen) * 0.9 + 0.05).cuda(), torch.randn(2, 1, 64, 64, generator=gen).cuda()
    params = list(net.parameters())

    def step(zero=True):
        if zero:
            for p in params:
                p.grad = None
        S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=cond, noise=(t, z)).backward()
    step()
    torch.cuda.synchronize()
    used = [p for p in params if p.grad is not None]
    g, ref, static = _capture(step, used)
    names = {id(p): k for k, p in net.named_parameters()}
    for r in range(4):
        g.replay()
        torch.cuda.synchronize()
        bad = [(names[id(p)], float((s - q).abs().max() / q.abs().max())) for p, s, q in zip(used, static, ref)
               if not float((s - q).abs().max() / q.abs().max()) < 1e-4]
        assert not bad, (r, bad[:4])
        if r == 1:
            step()
