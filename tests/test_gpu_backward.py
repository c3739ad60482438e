"""Training path on the GPU: every native backward kernel against PyTorch-CPU autograd of the same op, the ConvTranspose2d
ablation decoder against the reference's goldens, every parameter gradient of a 7-channel model against CPU autograd of the
oracle and a two-step SGD trajectory.  (loss_fn itself vs the reference's golden loss / gradients: test_gpu_loss.py; the
C3-shaped step: test_gpu_configs.py.)  Tolerance: 1e-4 max-rel on gradients (fp32 atomics reorder sums), where
max-rel = max|a - b| / max|b| over the whole tensor."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from sbgm_danra_amd import _native as N  # noqa: E402
from sbgm_danra_amd import train_graph as T  # noqa: E402
from util_models import build_pair, load_golden, maxrel  # noqa: E402

GT = 1e-4


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def leaf(t, dev=False):
    t = t.clone().cuda() if dev else t.clone()
    return t.requires_grad_(True)


@pytest.mark.parametrize("B,Cin,H,Cout,k,s,p", [(2, 64, 8, 64, 3, 1, 1), (2, 64, 8, 128, 3, 2, 1), (2, 64, 8, 128, 1, 2, 0),
                                                 (1, 64, 16, 64, 8, 2, 3), (2, 128, 4, 64, 3, 1, 1), (1, 256, 2, 512, 3, 2, 1),
                                                 (2, 64, 16, 64, 3, 1, 1), (1, 128, 32, 64, 3, 1, 1), (3, 64, 16, 128, 3, 1, 1),    # LDS-staged wgrad
                                                 (5, 128, 8, 64, 3, 1, 1), (8, 64, 8, 64, 3, 1, 1), (1, 64, 8, 64, 3, 1, 1),      # ... 8 x 8 maps: 4 images per tile
                                                 (9, 64, 16, 64, 3, 1, 1),                                                       # several tiles per workgroup
                                                 (3, 128, 16, 64, 1, 1, 0), (1, 64, 6, 128, 1, 1, 0), (2, 192, 10, 64, 1, 2, 0),  # LDS-staged 1x1 (pixel split / ragged tile / stride 2)
                                                 (8, 64, 32, 64, 1, 1, 0)])
def test_conv_backward(B, Cin, H, Cout, k, s, p):
    x, w, b = rnd(B, Cin, H, H), rnd(Cout, Cin, k, k, seed=1, scale=1 / math.sqrt(Cin * k * k)), rnd(Cout, seed=2)
    res_shape = F.conv2d(x, w, b, s, p).shape
    res, tb, go = rnd(*res_shape, seed=3), rnd(B, Cout, seed=4), rnd(*res_shape, seed=5)
    xr, wr, br, rr, tr = leaf(x), leaf(w), leaf(b), leaf(res), leaf(tb)
    (F.conv2d(xr, wr, br, s, p) + rr + tr[:, :, None, None]).backward(go)
    xd, wd, bd, rd, td = leaf(nhwc(x), True), leaf(w, True), leaf(b, True), leaf(nhwc(res), True), leaf(tb, True)
    y = T.ConvFn.apply(xd, wd, bd, rd, td, s, p)
    y.backward(nhwc(go).cuda())
    assert maxrel(nchw(xd.grad.cpu()), xr.grad) < GT
    assert maxrel(wd.grad.cpu(), wr.grad) < GT and maxrel(bd.grad.cpu(), br.grad) < GT
    assert maxrel(nchw(rd.grad.cpu()), rr.grad) < GT and maxrel(td.grad.cpu(), tr.grad) < GT


@pytest.mark.parametrize("cin", [2, 7, 13])
def test_stem_conv_weight_gradient_with_padded_channels(cin):
    B, H = 2, 32
    x, w, go = rnd(B, cin, H, H), rnd(64, cin, 8, 8, seed=1, scale=0.05), rnd(B, 64, H // 2, H // 2, seed=2)
    wr = leaf(w)
    F.conv2d(x, wr, None, 2, 3).backward(go)
    cs = T._pad_c(cin)
    xp = torch.zeros(B, H, H, cs)
    xp[..., :cin] = nhwc(x)
    wd = leaf(w, True)
    T.ConvFn.apply(xp.cuda(), wd, None, None, None, 2, 3).backward(nhwc(go).cuda())
    assert maxrel(wd.grad.cpu(), wr.grad) < GT


@pytest.mark.parametrize("with_res", [False, True])
def test_batchnorm_train_backward(with_res):
    B, Cc, H = 3, 64, 6
    x, g, b = rnd(B, Cc, H, H) * 2 + 0.3, rnd(Cc, seed=1), rnd(Cc, seed=2)
    res, tb, go = rnd(B, Cc, H, H, seed=3), rnd(B, Cc, seed=4), rnd(B, Cc, H, H, seed=5)
    xr, gr, br, rr, tr = leaf(x), leaf(g), leaf(b), leaf(res), leaf(tb)
    y = F.batch_norm(xr, torch.zeros(Cc), torch.ones(Cc), gr, br, True, 0.1, 1e-5)
    y = (F.relu(y + rr) + tr[:, :, None, None]) if with_res else F.relu(y)
    y.backward(go)
    xd, gd, bd = leaf(nhwc(x), True), leaf(g, True), leaf(b, True)
    rd, td = (leaf(nhwc(res), True), leaf(tb, True)) if with_res else (None, None)
    yd = T.BNTrainFn.apply(xd, gd, bd, torch.zeros(Cc).cuda(), torch.ones(Cc).cuda(), rd, td, True, 1e-5, 0.1)
    assert maxrel(nchw(yd.detach().cpu()), y.detach()) < 1e-5
    yd.backward(nhwc(go).cuda())
    assert maxrel(nchw(xd.grad.cpu()), xr.grad) < GT and maxrel(gd.grad.cpu(), gr.grad) < GT and maxrel(bd.grad.cpu(), br.grad) < GT
    if with_res:
        assert maxrel(nchw(rd.grad.cpu()), rr.grad) < GT and maxrel(td.grad.cpu(), tr.grad) < GT


@pytest.mark.parametrize("Cc,G,act", [(64, 8, N.SILU), (128, 8, N.NONE), (64, 64, N.RELU), (256, 8, N.GELU)])
def test_groupnorm_backward(Cc, G, act):
    B, H = 2, 8
    affine = G != Cc
    x, go = rnd(B, Cc, H, H) * 1.5 + 0.2, rnd(B, Cc, H, H, seed=5)
    g, b = (rnd(Cc, seed=1), rnd(Cc, seed=2)) if affine else (None, None)
    skip, tb = rnd(B, Cc, H, H, seed=3), rnd(B, Cc, seed=4)
    fn = {N.SILU: F.silu, N.RELU: F.relu, N.GELU: F.gelu, N.NONE: lambda v: v}[act]
    xr, sr, tr = leaf(x), leaf(skip), leaf(tb)
    gr, br = (leaf(g), leaf(b)) if affine else (None, None)
    fn(F.group_norm(xr, G, gr, br, 1e-5) + sr + tr[:, :, None, None]).backward(go)
    xd, sd, td = leaf(nhwc(x), True), leaf(nhwc(skip), True), leaf(tb, True)
    gd, bd = (leaf(g, True), leaf(b, True)) if affine else (None, None)
    T.GroupNormFn.apply(xd, gd, bd, sd, td, act, G, 1e-5).backward(nhwc(go).cuda())
    assert maxrel(nchw(xd.grad.cpu()), xr.grad) < GT and maxrel(nchw(sd.grad.cpu()), sr.grad) < GT and maxrel(td.grad.cpu(), tr.grad) < GT
    if affine:
        assert maxrel(gd.grad.cpu(), gr.grad) < GT and maxrel(bd.grad.cpu(), br.grad) < GT


@pytest.mark.parametrize("M,Cc", [(40, 256), (2051, 128), (4096, 512), (7, 96), (3, 1024)])   # 1..4 rows per wave, ragged tails
def test_layernorm_backward(M, Cc):
    x, g, b, go = rnd(M, Cc) * 2 + 1, rnd(Cc, seed=1), rnd(Cc, seed=2), rnd(M, Cc, seed=3)
    xr, gr, br = leaf(x), leaf(g), leaf(b)
    F.layer_norm(xr, (Cc,), gr, br, 1e-5).backward(go)
    xd, gd, bd = leaf(x, True), leaf(g, True), leaf(b, True)
    T.LayerNormFn.apply(xd, gd, bd, 1e-5).backward(go.cuda())
    assert maxrel(xd.grad.cpu(), xr.grad) < GT and maxrel(gd.grad.cpu(), gr.grad) < GT and maxrel(bd.grad.cpu(), br.grad) < GT


@pytest.mark.parametrize("B,S,Cc,heads,p", [(2, 64, 64, 4, 0.25), (1, 50, 96, 2, 0.1), (3, 16, 256, 4, 0.5), (1, 256, 128, 4, 0.2), (2, 33, 32, 1, 0.0)])
def test_attention_core_with_dropout(B, S, Cc, heads, p):
    """train-mode nn.MultiheadAttention(dropout=p): out = (softmax(q k^T / sqrt(d)) o D) v with D = keep / (1 - p).  The kernels' mask is
    read back through sbgm_mha_dropout_mask and the same expression is evaluated (forward and autograd) in torch on the CPU; the mask
    itself must keep ~(1 - p) of the probabilities, differ between seeds and be the same for forward and backward."""
    d, seed = Cc // heads, 1234567 + S
    qkv = rnd(B, S, 3 * Cc, seed=S) * 0.7
    go = rnd(B, S, Cc, seed=S + 1)
    mask = torch.empty(B, heads, S, S, device="cuda")
    N.check(T._L().sbgm_mha_dropout_mask(mask.data_ptr(), B, S, heads, p, seed, 0, N.stream()))
    m = mask.cpu()
    keep = float((m > 0).float().mean())
    assert set(torch.unique(m).tolist()) <= {0.0, float(torch.tensor(1.0 / (1.0 - p), dtype=torch.float32))}
    assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / m.numel()) ** 0.5 + 1e-9
    if p > 0:
        other = torch.empty_like(mask)
        N.check(T._L().sbgm_mha_dropout_mask(other.data_ptr(), B, S, heads, p, seed + 1, 0, N.stream()))
        assert not torch.equal(other, mask)
    qr = leaf(qkv)
    q, k, v = [t.view(B, S, heads, d).transpose(1, 2) for t in qr.split(Cc, dim=-1)]
    a = torch.softmax(q @ k.transpose(-1, -2) / d ** 0.5, dim=-1)
    want = ((a * m) @ v).transpose(1, 2).reshape(B, S, Cc)
    want.backward(go)
    qd = leaf(qkv.view(B * S, 3 * Cc), True)
    got = T.MHACoreDropoutFn.apply(qd, B, S, Cc, heads, p, seed)
    got.backward(go.view(B * S, Cc).cuda())
    assert maxrel(got.detach().cpu().view(B, S, Cc), want.detach()) < 1e-5
    assert maxrel(qd.grad.cpu().view(B, S, 3 * Cc), qr.grad) < GT
    if p == 0.0:                                               # no dropout: the same numbers as the production core
        plain = T.MHACoreFn.apply(leaf(qkv.view(B * S, 3 * Cc), True), B, S, Cc, heads)
        assert maxrel(plain.detach().cpu(), got.detach().cpu()) < 1e-5


def test_attention_block_trains_with_dropout():
    """ImageSelfAttention(dropout=p).train() through the autograd path: runs, is repeatable under torch.manual_seed, differs between
    calls, and equals the eval-mode block when every probability is kept (p -> 0 limit: p = 0)"""
    import sbgm_danra_amd as S
    torch.manual_seed(3)
    blk = S.ImageSelfAttention(64, 4, dropout=0.3).cuda()
    x = (torch.randn(2, 8, 8, 64) * 0.5).cuda().requires_grad_(True)
    blk.train()
    torch.manual_seed(11); a = T._attention(blk, x)
    torch.manual_seed(11); b = T._attention(blk, x)
    c = T._attention(blk, x)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(a).all()
    a.square().mean().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and all(p.grad is not None for p in blk.parameters())
    blk.eval()
    e = T._attention(blk, x)
    blk0 = S.ImageSelfAttention(64, 4, dropout=0.0).cuda()
    blk0.load_state_dict(blk.state_dict())
    blk0.train()
    assert maxrel(T._attention(blk0, x).detach().cpu(), e.detach().cpu()) < 1e-5


def test_act_upsample_backward():
    M, Cc = 40, 256
    x, go = rnd(M, Cc) * 2 + 1, rnd(M, Cc, seed=3)
    xr = leaf(x)
    F.gelu(xr).backward(go)
    xd = leaf(x, True)
    T.ActFn.apply(xd, N.GELU).backward(go.cuda())
    assert maxrel(xd.grad.cpu(), xr.grad) < 1e-5
    for shape in [(2, 64, 4, 4), (1, 64, 1, 1), (1, 128, 3, 5)]:
        u = rnd(*shape)
        gu = rnd(shape[0], shape[1], 2 * shape[2], 2 * shape[3], seed=7)
        ur = leaf(u)
        F.interpolate(ur, scale_factor=2, mode="bilinear", align_corners=False).backward(gu)
        ud = leaf(nhwc(u), True)
        T.UpsampleFn.apply(ud).backward(nhwc(gu).cuda())
        assert maxrel(nchw(ud.grad.cpu()), ur.grad) < 1e-5


@pytest.mark.parametrize("B,S,Cc,heads", [(2, 64, 256, 4), (1, 16, 512, 4), (1, 256, 128, 4), (2, 4, 256, 2), (1, 40, 128, 1),
                                          (1, 1024, 128, 4),       # exceeds the LDS-staged variant: scalar fallback
                                          (2, 64, 128, 16), (1, 144, 96, 8)])      # head dims 8 and 12 (num_heads = 16 of the reference's sweep)
def test_mha_core_backward(B, S, Cc, heads):
    qkv, go = rnd(B * S, 3 * Cc), rnd(B * S, Cc, seed=1)
    d = Cc // heads
    qr = leaf(qkv)
    q, k, v = [t.view(B, S, heads, d).transpose(1, 2) for t in qr.view(B, S, 3 * Cc).split(Cc, dim=-1)]
    att = torch.softmax((q / math.sqrt(d)) @ k.transpose(-1, -2), -1) @ v
    att.transpose(1, 2).reshape(B * S, Cc).backward(go)
    qd = leaf(qkv, True)
    T.MHACoreFn.apply(qd, B, S, Cc, heads).backward(go.cuda())
    assert maxrel(qd.grad.cpu(), qr.grad) < GT


@pytest.mark.parametrize("with_y", [False, True])
def test_time_projection_backward(with_y):
    B, D, ch = 4, 256, 128
    t = torch.tensor([1e-3, 0.2, 0.7, 1.0])
    freqs, w, b, table, go = rnd(D // 2) * 30, rnd(ch, D, seed=1) * 0.05, rnd(ch, seed=2), rnd(5, D, seed=3), rnd(B, ch, seed=4)
    y = torch.tensor([1, 4, 0, 2]) if with_y else None
    wr, br, tabr = leaf(w), leaf(b), leaf(table)
    pr = t[:, None] * freqs[None, :] * (2 * torch.pi)
    emb = torch.cat([pr.sin(), pr.cos()], -1)
    if with_y:
        emb = emb + tabr[y]
    F.linear(F.silu(emb), wr, br).backward(go)
    wd, bd, tabd = leaf(w, True), leaf(b, True), leaf(table, True)
    T.TimeProjFn.apply(t.cuda(), y.cuda() if with_y else None, tabd if with_y else None, freqs.cuda(), wd, bd).backward(go.cuda())
    assert maxrel(wd.grad.cpu(), wr.grad) < GT and maxrel(bd.grad.cpu(), br.grad) < GT
    if with_y:
        assert maxrel(tabd.grad.cpu(), tabr.grad) < GT


def test_time_projection_multi_equals_single_projections():
    """TimeProjMultiFn (one launch pair for several projections) == one TimeProjFn per projection: outputs and every gradient,
    for the encoder's pattern (5 heads on one embedding + label embedding) and the decoder's (4 embeddings, one head each)"""
    B, D = 4, 256
    t, y = torch.tensor([1e-3, 0.2, 0.7, 1.0]).cuda(), torch.tensor([1, 4, 0, 2]).cuda()
    chs = [64, 64, 128, 256, 512]
    for emb_index, n_emb, with_y in (((0,) * 5, 1, True), ((0, 1, 2, 3), 4, False)):
        n = len(emb_index)
        freqs = [rnd(D // 2, seed=10 + i).cuda() * 30 for i in range(n_emb)]
        ws = [rnd(chs[i], D, seed=20 + i) * 0.05 for i in range(n)]
        bs = [rnd(chs[i], seed=30 + i) for i in range(n)]
        gos = [rnd(B, chs[i], seed=40 + i).cuda() for i in range(n)]
        table = rnd(5, D, seed=3)
        w1, b1, tab1 = [leaf(w, True) for w in ws], [leaf(b, True) for b in bs], leaf(table, True)
        single = [T.TimeProjFn.apply(t, y if with_y else None, tab1 if with_y else None, freqs[emb_index[i]], w1[i], b1[i]) for i in range(n)]
        torch.autograd.backward(single, gos)
        w2, b2, tab2 = [leaf(w, True) for w in ws], [leaf(b, True) for b in bs], leaf(table, True)
        multi = T.TimeProjMultiFn.apply(t, y if with_y else None, tab2 if with_y else None, emb_index, n_emb, *freqs, *w2, *b2)
        torch.autograd.backward(multi, gos)
        for i in range(n):
            assert torch.equal(multi[i], single[i])
            assert torch.equal(w2[i].grad, w1[i].grad) and torch.equal(b2[i].grad, b1[i].grad)
        if with_y:
            assert maxrel(tab2.grad.cpu(), tab1.grad.cpu()) < 1e-6           # atomics: the accumulation order may differ


def test_deferred_weight_gradient_layout_passes():
    """sbgm_wgrad_defer / sbgm_wgrad_flush: queued slab -> OIHW conversions (3x3, 1x1 with channel padding, the 8-channel stem),
    flushed as one launch, give exactly the gradients of the immediate form"""
    lib = N.lib()
    cases = [(2, 64, 16, 64, 3, 1, 1, 64), (3, 128, 8, 64, 3, 1, 1, 128), (2, 64, 16, 128, 3, 2, 1, 64), (1, 5, 32, 64, 8, 2, 3, 8),
             (2, 64, 8, 64, 3, 1, 1, 64), (4, 64, 32, 64, 3, 1, 1, 64), (4, 64, 32, 128, 3, 2, 1, 64)]
    keep, now, later = [], [], []
    for defer in (0, 1):
        prev = lib.sbgm_wgrad_defer(defer)
        for i, (B, cin, H, cout, k, s, p, cs) in enumerate(cases):
            oh = (H + 2 * p - k) // s + 1
            x = torch.zeros(B, H, H, cs)
            x[..., :cin] = rnd(B, H, H, cin, seed=i)
            dy = rnd(B, oh, oh, cout, seed=50 + i)
            xd, dyd = x.cuda(), dy.cuda()
            dw, ws = torch.full((cout, cin, k, k), float("nan"), device="cuda"), torch.zeros(k * k * cout * cs, device="cuda")
            N.check(lib.sbgm_conv2d_wgrad(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, H, H, cs, cin, cout, k, k, s, p,
                                          N.stream()))
            keep += [xd, dyd, ws]
            (later if defer else now).append(dw)
        lib.sbgm_wgrad_defer(prev)
    assert 2 <= lib.sbgm_wgrad_flush_pending() <= len(cases)                 # layers without a pixel split store OIHW directly
    assert any(torch.isnan(d).any() for d in later)                          # ... the queued ones are not written yet
    N.check(lib.sbgm_wgrad_flush(N.stream()))
    assert lib.sbgm_wgrad_flush_pending() == 0
    for a, b in zip(now, later):
        assert maxrel(b.cpu(), a.cpu()) < 1e-6 and not torch.isnan(b).any()  # atomically accumulated slabs: order may differ in the last bit


def test_deferred_weight_gradient_gemms_run_as_batched_launches():
    """sbgm_wgrad_defer bit 1: the weight-gradient GEMMs themselves are queued (3x3 halo kernel on 16x16 tiles and on 8x8 maps, the per-tap
    kernel in its 64x64 and 128x128 block forms, a linear whose slab IS the gradient, the 8-channel stem) and run at the flush as one
    launch per kernel family with pixel splits chosen across the batch; results equal the immediate launches (atomics: last-bit order)
    and torch's conv2d weight gradients"""
    lib = N.lib()
    #        B  cin   H  cout k  s  p   cs
    cases = [(2, 64, 32, 64, 3, 1, 1, 64), (2, 128, 16, 128, 3, 1, 1, 128), (4, 256, 8, 256, 3, 1, 1, 256), (2, 64, 16, 128, 3, 2, 1, 64),
             (2, 128, 16, 256, 3, 2, 1, 128), (2, 512, 4, 512, 3, 1, 1, 512), (1, 5, 32, 64, 8, 2, 3, 8), (2, 64, 16, 64, 8, 2, 3, 64),
             (2, 128, 16, 256, 1, 2, 0, 128)]
    linears = [(300, 256, 768), (128, 512, 512), (2048, 128, 128)]            # (tokens, Cin, Cout): ws aliases dw
    keep, res = [], {0: [], 3: []}
    for defer in (0, 3):
        prev = lib.sbgm_wgrad_defer(defer)
        for i, (B, cin, H, cout, k, s, p, cs) in enumerate(cases):
            oh = (H + 2 * p - k) // s + 1
            x = torch.zeros(B, H, H, cs)
            x[..., :cin] = rnd(B, H, H, cin, seed=i)
            dy = rnd(B, oh, oh, cout, seed=50 + i)
            xd, dyd = x.cuda(), dy.cuda()
            dw, ws = torch.zeros(cout, cin, k, k, device="cuda"), torch.zeros(k * k * cout * cs, device="cuda")
            db = torch.zeros(cout, device="cuda")
            N.check(lib.sbgm_conv2d_wgrad_bias(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, H, H, cs, cin,
                                               cout, k, k, s, p, N.stream()))
            keep += [xd, dyd, ws]
            res[defer].append((dw, db, x[..., :cin], dy, s, p))
        for i, (M, cin, cout) in enumerate(linears):
            x, dy = rnd(M, cin, seed=80 + i), rnd(M, cout, seed=90 + i)
            xd, dyd = x.cuda(), dy.cuda()
            dw = torch.zeros(cout, cin, 1, 1, device="cuda")
            N.check(lib.sbgm_conv2d_wgrad(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), dw.data_ptr(), 1, 1, M, cin, cin, cout, 1, 1, 1, 0,
                                          N.stream()))
            keep += [xd, dyd]
            res[defer].append((dw, None, x.view(1, 1, M, cin), dy.view(1, 1, M, cout), 1, 0))
        lib.sbgm_wgrad_defer(prev)
    assert lib.sbgm_wgrad_flush_pending() >= len(cases) + len(linears) - 1    # (a layer the general kernel serves is not queued)
    torch.cuda.synchronize()
    assert all(float(d[0].abs().max()) == 0.0 for d in res[3])               # nothing has run yet
    N.check(lib.sbgm_wgrad_flush(N.stream()))
    assert lib.sbgm_wgrad_flush_pending() == 0
    for (a, ab, x, dy, s, p), (b, bb, _, _, _, _) in zip(res[0], res[3]):
        assert maxrel(b.cpu(), a.cpu()) < 1e-5
        if ab is not None:
            assert maxrel(bb.cpu(), ab.cpu()) < 1e-5
        xr = nchw(x).clone().requires_grad_(False)
        wr = torch.zeros(b.shape, requires_grad=True)
        F.conv2d(xr, wr, None, s, p).backward(nchw(dy))
        assert maxrel(b.cpu(), wr.grad) < GT


def test_final_conv_backward():
    B, Cc, H, W = 2, 64, 10, 12
    a, w, b, t, go = rnd(B, Cc, H, W), rnd(1, Cc, 3, 3, seed=1) * 0.05, rnd(1, seed=2), torch.tensor([0.05, 0.8]), rnd(B, 1, H, W, seed=3)
    ar, wr, br = leaf(a), leaf(w), leaf(b)
    ls = math.log(25.0)
    std = torch.sqrt((torch.exp(2 * t * ls) - 1) / (2 * ls))
    (F.conv2d(ar, wr, br, 1, 1) / std.view(-1, 1, 1, 1)).backward(go)
    ad, wd, bd = leaf(nhwc(a), True), leaf(w, True), leaf(b, True)
    T.Cout1Fn.apply(ad, wd, bd, t.cuda(), 25.0).backward(go.cuda())
    assert maxrel(nchw(ad.grad.cpu()), ar.grad) < GT and maxrel(wd.grad.cpu(), wr.grad) < GT and maxrel(bd.grad.cpu(), br.grad) < GT


def test_transpose_decoder_matches_reference_goldens(golden_dir):
    """the ConvTranspose2d ablation decoder (model.use_resize_conv = false): forward, loss and gradients vs the reference"""
    import sbgm_danra_amd as S
    g = load_golden(os.path.join(golden_dir, "transpose_decoder_b2_64.npz"))
    ora, net, _ = build_pair(1, resize=False)
    assert "decoder.residual_layers.0.transpose.weight" in net.state_dict() and "decoder.final_layer.conv_up.weight" not in net.state_dict()
    net.eval()
    x, cond, t = g["x"].cuda(), g["cond_img"].cuda(), g["t"].cuda()
    with torch.no_grad():
        assert maxrel(net(x, t, cond_img=cond).cpu(), g["score_eval"]) <= 1e-4
        got = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=2, num_steps=2, device="cuda", img_size=64,
                           cond_img=cond, seed=3)
    assert torch.isfinite(got).all()
    net.train()
    loss = S.loss_fn(net, x, S.marginal_prob_std_fn, cond_img=cond, noise=(g["t_used"].cuda(), g["z_used"].cuda()))
    loss.backward()
    assert abs(float(loss) / float(g["loss"]) - 1) < 1e-5
    params = dict(net.named_parameters())
    worst = {}
    for k in [k for k in g if k.startswith("grad_sub::")]:
        gr = params[k[10:]].grad.reshape(-1).cpu()
        worst[k] = maxrel(gr[:: (257 if gr.numel() > 4096 else 1)][:4096], g[k])
    print("transpose decoder probe-gradient max-rel:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert max(worst.values()) < GT, worst


def _batch(gen, B=3, hw=64):
    x, cond = torch.randn(B, 1, hw, hw, generator=gen), torch.randn(B, 1, hw, hw, generator=gen)
    lsm = torch.cat([(torch.rand(B, 1, hw, hw, generator=gen) > 0.5).float(), torch.ones(B, 1, hw, hw)], 1)
    topo = torch.cat([torch.rand(B, 1, hw, hw, generator=gen), torch.ones(B, 1, hw, hw)], 1)
    y = torch.randint(0, 5, (B,), generator=gen)
    t, z = torch.rand(B, generator=gen) * 0.999 + 1e-3, torch.randn(B, 1, hw, hw, generator=gen)
    return x, cond, lsm, topo, y, t, z


def _native_loss(net, S, x, cond, lsm, topo, y, t, z):
    return S.loss_fn(net, x.cuda(), S.marginal_prob_std_fn, y=y.cuda(), cond_img=cond.cuda(), lsm_cond=lsm.cuda(), topo_cond=topo.cuda(),
                     noise=(t.cuda(), z.cuda()))


def test_every_parameter_gradient_matches_the_oracle():
    """7-channel input, seasons, geo conditions: all 166 parameter gradients vs CPU autograd of the oracle"""
    import sbgm_danra_amd as S
    from oracle import torch_ref as O
    ora, net, _ = build_pair(5, 4)
    ora.train(), net.train()
    x, cond, lsm, topo, y, t, z = _batch(torch.Generator().manual_seed(11))
    lo = O.loss_fn(ora, x, O.marginal_prob_std_fn, y=y, cond_img=cond, lsm_cond=lsm, topo_cond=topo, noise=(t, z))
    lo.backward()
    ln = _native_loss(net, S, x, cond, lsm, topo, y, t, z)
    ln.backward()
    assert abs(float(ln.detach()) / float(lo.detach()) - 1) < 1e-5
    po, pn = dict(ora.named_parameters()), dict(net.named_parameters())
    unused = ("decoder.final_layer.time_projection_layer", "decoder.final_layer.sinusoidal")
    errs = {}
    for k, p in po.items():
        if k.startswith(unused):
            assert pn[k].grad is None and p.grad is None          # never used by the forward (score_unet.py:757)
            continue
        errs[k] = maxrel(pn[k].grad.cpu(), p.grad)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("all-parameter gradients vs oracle (B=3, 64x64, 7 channels), worst max-rel:", [(k, f"{v:.2e}") for k, v in worst])
    assert worst[0][1] < GT, worst
    # BatchNorm running statistics moved identically
    so, sn = ora.state_dict(), net.state_dict()
    for k in so:
        if "running_" in k:
            assert maxrel(sn[k].cpu(), so[k]) < 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(sn[k]) == int(so[k])


def test_two_sgd_steps_track_the_oracle():
    """same data and injected (t, z), lr small enough for a stable trajectory (|grad| ~ 5e3 at these sigmas): the parameter
    UPDATES after 2 SGD steps agree with the CPU oracle; the inference engine then
    sees the updated weights (version-counter re-upload)"""
    import sbgm_danra_amd as S
    from oracle import torch_ref as O
    ora, net, sd0 = build_pair(5, 4)
    ora.train(), net.train()
    opt_o, opt_n = torch.optim.SGD(ora.parameters(), lr=1e-6), torch.optim.SGD(net.parameters(), lr=1e-6)
    gen = torch.Generator().manual_seed(12)
    for _ in range(2):
        b = _batch(gen, B=2)
        x, cond, lsm, topo, y, t, z = b
        opt_o.zero_grad()
        lo = O.loss_fn(ora, x, O.marginal_prob_std_fn, y=y, cond_img=cond, lsm_cond=lsm, topo_cond=topo, noise=(t, z))
        lo.backward()
        opt_o.step()
        opt_n.zero_grad()
        ln = _native_loss(net, S, *b)
        ln.backward()
        opt_n.step()
        assert abs(float(ln.detach()) / float(lo.detach()) - 1) < 1e-4
    so, sn = ora.state_dict(), net.state_dict()
    worst = max(maxrel(sn[k].cpu().float(), so[k].float()) for k in so if so[k].dtype.is_floating_point)
    assert worst < 1e-3, worst            # updates are O(|w|) for the norm biases here; gradient error itself is <= 2e-4
    assert max(float((so[k] - sd0[k]).abs().max()) for k in so if so[k].dtype.is_floating_point) > 1e-4   # weights did move
    ora.eval(), net.eval()
    x, cond, lsm, topo, y, t, z = _batch(gen, B=2)
    with torch.no_grad():
        want = ora(x, t, y, cond, lsm, topo)
        got = net(x.cuda(), t.cuda(), y.cuda(), cond.cuda(), lsm.cuda(), topo.cuda()).cpu()
    assert maxrel(got, want) < 1e-3         # the two models' weights now differ by up to ~5e-4 (above)


@pytest.mark.parametrize("B,Cin,Cout,H", [(2, 64, 64, 32), (8, 64, 64, 64), (1, 32, 64, 16), (3, 64, 128, 24)])
def test_stem_conv_phase_decomposed_data_gradient(B, Cin, Cout, H):
    """8x8 / stride 2 / pad 3 (encoder.conv2): dx through the 5x5 phase operator + depth->space instead of the zero-dilated 8x8"""
    x, w = rnd(B, Cin, H, H), rnd(Cout, Cin, 8, 8, seed=1, scale=1 / math.sqrt(Cin * 64))
    go = rnd(B, Cout, H // 2, H // 2, seed=5)
    xr, wr = leaf(x), leaf(w)
    F.conv2d(xr, wr, None, 2, 3).backward(go)
    xd, wd = leaf(nhwc(x), True), leaf(w, True)
    T.ConvFn.apply(xd, wd, None, None, None, 2, 3).backward(nhwc(go).cuda())
    assert maxrel(nchw(xd.grad.cpu()), xr.grad) < GT and maxrel(wd.grad.cpu(), wr.grad) < GT
    # the operator itself: only the taps of the right parity survive
    lib = N.lib()
    ph = torch.empty(4 * Cin, Cout, 5, 5, device="cuda")
    N.check(lib.sbgm_conv8x8s2_dgrad_phase_weight(wd.data_ptr(), ph.data_ptr(), Cout, Cin, N.stream()))
    ph = ph.cpu().view(2, 2, Cin, Cout, 5, 5)
    assert torch.equal(ph[0, 0, :, :, 0, 0], w[:, :, 7, 7].t()) and torch.equal(ph[1, 1, :, :, 4, 4], w[:, :, 0, 0].t())
    assert float(ph[0, :, :, :, 4, :].abs().max()) == 0.0 and float(ph[1, :, :, :, 0, :].abs().max()) == 0.0
