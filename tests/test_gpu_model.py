"""Whole-path parity on the GPU: native ScoreNet / samplers (through libsbgm_hip.so) vs the golden vectors captured
from the reference, and vs the CPU oracle on fresh seeded inputs.  Tolerance: <= 1e-4 max-rel per network
evaluation (north_star); short sampler horizons are held to 1e-3 (chained evaluations)."""
import os

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from util_models import build_pair, check_parity, load_golden, maxrel  # noqa: E402

TOL = 1e-4
CASES = {"fwd_b2_64_c2": (1, None), "fwd_b1_128_c7_y": (6, 4), "fwd_b2_32_c1": (0, None)}


def _dev(d, k):
    return d[k].cuda() if k in d else None


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_forward_matches_reference_goldens(golden_dir, name, mode):
    n_cond, classes = CASES[name]
    g = load_golden(os.path.join(golden_dir, name + ".npz"))
    _, net, _ = build_pair(n_cond, classes)
    net.train(mode == "train")
    # 32x32 input in train mode: layer4 is 1x1, so BatchNorm's batch statistics are taken over B*H*W = 2 values per
    # channel and (x - mean) * rstd amplifies fp32 reassociation noise of the conv by |x| / |x_0 - x_1|; the reference
    # itself is ill-conditioned there, so this one case is held to 5e-3 instead of 1e-4.
    tol = 5e-3 if (name == "fwd_b2_32_c1" and mode == "train") else TOL
    fm = []
    with torch.no_grad():
        out = net(g["x"].cuda(), g["t"].cuda(), _dev(g, "y"), _dev(g, "cond_img"), _dev(g, "lsm_cond"), _dev(g, "topo_cond"), _fmaps=fm)
    # (the ill-conditioned 2-value BatchNorm case is reported only: its element-wise figure is not a property of the kernels)
    check_parity(out.cpu(), g[f"score_{mode}"], tol, f"forward golden {name} {mode}", elem_tol=1.0 if tol > TOL else None)
    for i, f in enumerate(fm):          # encoder feature maps: strided NHWC subsample + abs-mean recorded from the reference
        flat = f.reshape(-1).cpu()
        sub = flat[:: max(1, flat.numel() // 4096)][:4096]
        if tol > TOL and i == 4:
            continue                    # the degenerate 2-value BatchNorm output itself (see above)
        assert maxrel(sub, g[f"fmap{i + 1}_{mode}_sub"]) <= tol, f"fmap{i + 1}"
        assert abs(float(f.abs().mean()) / float(g[f"fmap{i + 1}_{mode}_absmean"]) - 1) <= tol


def test_train_mode_updates_running_stats_like_torch():
    ora, net, sd = build_pair(1)
    x, t, c = torch.randn(4, 1, 64, 64), torch.rand(4) * 0.9 + 0.05, torch.randn(4, 1, 64, 64)
    ora.train(), net.train()
    with torch.no_grad():
        ora(x, t, cond_img=c)
        net(x.cuda(), t.cuda(), cond_img=c.cuda())
    so, sn = ora.state_dict(), net.state_dict()
    for k in so:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert maxrel(sn[k].cpu(), so[k]) <= 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(sn[k]) == int(so[k]) == 1


@pytest.mark.parametrize("B,hw,n_lr,geo,classes", [(3, 64, 4, True, 4), (1, 32, 1, False, None), (2, 96, 2, False, None)])
def test_forward_matches_oracle_fresh_inputs(B, hw, n_lr, geo, classes):
    n_cond = n_lr + (4 if geo else 0)
    ora, net, _ = build_pair(n_cond, classes)
    ora.eval(), net.eval()
    g = torch.Generator().manual_seed(hw + B)
    x, t = torch.randn(B, 1, hw, hw, generator=g), torch.rand(B, generator=g) * 0.999 + 1e-3
    cond = torch.randn(B, n_lr, hw, hw, generator=g)
    lsm = torch.cat([(torch.rand(B, 1, hw, hw, generator=g) > 0.5).float(), torch.ones(B, 1, hw, hw)], 1) if geo else None
    topo = torch.cat([torch.rand(B, 1, hw, hw, generator=g), torch.ones(B, 1, hw, hw)], 1) if geo else None
    y = torch.randint(0, classes + 1, (B,), generator=g) if classes else None
    cu = lambda v: None if v is None else v.cuda()  # noqa: E731
    with torch.no_grad():
        want = ora(x, t, y, cond, lsm, topo)
        got = net(cu(x), cu(t), cu(y), cu(cond), cu(lsm), cu(topo)).cpu()
    check_parity(got, want, TOL, f"forward fresh B{B} {hw}x{hw} c{n_cond}")


def test_instance_norm_decoder_and_other_shapes():
    ora, net, _ = build_pair(1, norm="instance", heads=8, temb=128, layers=(1, 2, 1, 1))
    ora.eval(), net.eval()
    x, t, c = torch.randn(2, 1, 64, 64), torch.tensor([0.3, 0.9]), torch.randn(2, 1, 64, 64)
    with torch.no_grad():
        assert maxrel(net(x.cuda(), t.cuda(), cond_img=c.cuda()).cpu(), ora(x, t, cond_img=c)) <= TOL


def test_pc_sampler_matches_reference_golden(golden_dir):
    import sbgm_danra_amd as S
    g = load_golden(os.path.join(golden_dir, "pc_b2_64_3steps.npz"))
    _, net, _ = build_pair(1)
    net.eval()
    got = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=2, num_steps=3, device="cuda", img_size=64,
                       cond_img=g["cond_img"].cuda(), noise=g["noise"])
    check_parity(got.cpu(), g["x_mean"], 1e-3, "pc sampler golden 3 steps")


def test_em_sampler_matches_reference_golden(golden_dir):
    import sbgm_danra_amd as S
    g = load_golden(os.path.join(golden_dir, "em_b2_32_5steps.npz"))
    _, net, _ = build_pair(1)
    net.eval()
    got = S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=2, num_steps=5, device="cuda",
                                   img_size=32, cond_img=g["cond_img"].cuda(), noise=g["noise"])
    check_parity(got.cpu(), g["mean_x"], 1e-3, "em sampler golden 5 steps")


def test_cfg_guided_score_matches_reference_golden(golden_dir):
    import sbgm_danra_amd as S
    g = load_golden(os.path.join(golden_dir, "cfg_b2_32_c6_y.npz"))
    _, net, _ = build_pair(5, 4)
    net.eval()
    with torch.no_grad():
        got = S.guided_score_fn(net, g["x"].cuda(), g["t"].cuda(), g["y"].cuda(), g["cond_img"].cuda(), g["lsm_cond"].cuda(),
                                g["topo_cond"].cuda(), scale=1.5)
    check_parity(got.cpu(), g["guided"], TOL, "guided score golden")


GUIDED = {"classifier_free_guidance": {"enabled": True, "guidance_scale": 2.5, "guidance_scale_max": 1.5}}


def test_guided_samplers_match_reference_goldens(golden_dir):
    """live classifier-free guidance in the native loop: conditional + unconditional rows evaluated as one 2B batch"""
    import sbgm_danra_amd as S
    _, net, _ = build_pair(5, 4)
    net.eval()
    g = gp = load_golden(os.path.join(golden_dir, "pc_cfg_b2_32_2steps.npz"))
    kw = dict(y=g["y"].cuda(), cond_img=g["cond_img"].cuda(), lsm_cond=g["lsm_cond"].cuda(), topo_cond=g["topo_cond"].cuda(),
              cfg=GUIDED, device="cuda", batch_size=2, img_size=32)
    got = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=2, noise=g["noise"], **kw)
    check_parity(got.cpu(), g["x_mean"], 1e-3, "guided pc sampler golden")
    g = load_golden(os.path.join(golden_dir, "em_cfg_b2_32_3steps.npz"))
    got = S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=3, noise=g["noise"], **kw)
    check_parity(got.cpu(), g["mean_x"], 1e-3, "guided em sampler golden")
    # graph replay == eager launches, and the generic-callable Python loop (two evaluations + combine kernel) agrees
    a = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=3, seed=11, use_graph=True, **kw)
    b = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=3, seed=11, use_graph=False, **kw)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    f = lambda x, t, y=None, c=None, l=None, tp=None: net(x, t, y, c, l, tp)  # noqa: E731
    c = S.pc_sampler(f, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=2, noise=gp["noise"], **kw)
    d = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=2, noise=gp["noise"], **kw)
    assert maxrel(c.cpu(), d.cpu()) <= 1e-4


def test_python_loop_sampler_equals_native_loop():
    """generic-callable path (Python loop + fused update kernels) vs the single-call native loop, same noise"""
    import sbgm_danra_amd as S
    _, net, _ = build_pair(1)
    net.eval()
    g = torch.Generator().manual_seed(3)
    cond = torch.randn(2, 1, 32, 32, generator=g).cuda()
    noise = torch.randn(9, 2, 1, 32, 32, generator=g)
    f = lambda x, t, y=None, c=None, l=None, tp=None: net(x, t, y, c, l, tp)  # noqa: E731  (a plain callable, not a ScoreNet)
    a = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=2, num_steps=4, device="cuda", img_size=32,
                     cond_img=cond, noise=noise)
    b = S.pc_sampler(f, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=2, num_steps=4, device="cuda", img_size=32,
                     cond_img=cond, noise=noise)
    assert maxrel(a.cpu(), b.cpu()) <= 1e-4


def test_graph_replay_equals_eager_and_seed_reproducible():
    import sbgm_danra_amd as S
    _, net, _ = build_pair(1)
    net.eval()
    cond = torch.randn(2, 1, 64, 64).cuda()
    kw = dict(batch_size=2, num_steps=6, device="cuda", img_size=64, cond_img=cond, seed=77)
    a = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, use_graph=True, **kw)
    b = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, use_graph=False, **kw)
    c = S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, use_graph=True, **kw)
    d = S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, use_graph=False, **kw)
    assert torch.equal(a, b) and torch.equal(c, d)
    assert torch.isfinite(a).all() and not torch.equal(a, c)


def test_tile_table_save_load_roundtrip(tmp_path):
    """autotune -> save; a second model loads the table (no tuning) and evaluates bit-identically; a corrupt file is an error"""
    from sbgm_danra_amd._native import NativeError
    _, a, _ = build_pair(1)
    _, b, _ = build_pair(1)
    a.eval(), b.eval()
    path = str(tmp_path / "tiles.txt")
    a.autotune(2, 64, 64, cache=path)
    assert os.path.getsize(path) > 200
    b.autotune(2, 64, 64, cache=path)                  # loads
    x, c, t = torch.randn(2, 1, 64, 64).cuda(), torch.randn(2, 1, 64, 64).cuda(), torch.tensor([0.2, 0.7]).cuda()
    with torch.no_grad():
        assert torch.equal(a(x, t, cond_img=c), b(x, t, cond_img=c))
    bad = str(tmp_path / "bad.txt")
    open(bad, "w").write("3 3 1 1 2 64 64 64 64 0 | 7 1 1 1 0 0\n")
    with pytest.raises(NativeError):
        b.autotune(2, 64, 64, cache=bad)


def test_static_plan_uses_the_current_kernels_and_agrees_with_the_tuned_plan(tmp_path):
    """A sampler or forward call that never ran the autotuner must not fall back to the first-generation kernels: at the C2 shape
    the static choice (engine.hip pick_tile) puts the large 3x3 layers on the 2-D Winograd kernels and the 8x8 / 4x4 maps on the
    1-D Winograd kernel, and its output equals the tuned plan's up to the kernels' rounding."""
    import csv
    import ctypes as C
    from sbgm_danra_amd import _native as N
    _, net, _ = build_pair(1)
    net.eval()
    g = torch.Generator().manual_seed(3)
    B, HW = 32, 128
    x, c = torch.randn(B, 1, HW, HW, generator=g).cuda(), torch.randn(B, 1, HW, HW, generator=g).cuda()
    t = (torch.rand(B, generator=g) * 0.999 + 1e-3).cuda()
    with torch.no_grad():
        static = net(x, t, cond_img=c)
    eng = net._engine(None, None, c)
    prof, o, path = N.Profile(), torch.empty_like(x), str(tmp_path / "convs.csv")
    N.check(N.lib().sbgm_model_profile_forward(eng.h, x.data_ptr(), t.data_ptr(), None, c.data_ptr(), None, None, o.data_ptr(), B, HW, HW,
                                               C.byref(prof), path.encode(), N.stream()))
    rows = list(csv.DictReader(open(path)))
    big = [r for r in rows if r["kh"] == "3" and r["stride"] == "1" and int(r["W"]) >= 32 and int(r["Cin_pad"]) >= 64]
    small = [r for r in rows if r["kh"] == "3" and r["stride"] == "1" and int(r["W"]) <= 8]
    assert big and all("w2d" in r["kernel"] for r in big), [r["kernel"] for r in big]
    assert small and all("wino" in r["kernel"] for r in small), [r["kernel"] for r in small]
    net.autotune(B, HW, HW)
    with torch.no_grad():
        tuned = net(x, t, cond_img=c)
    assert maxrel(static.cpu(), tuned.cpu()) <= 2e-5


@pytest.mark.parametrize("B,H,W", [(8, 96, 128), (40, 96, 128)])
def test_non_square_maps_with_ragged_tile_rows(B, H, W):
    """96 x 128 inputs put 24-row maps (one and a half 16-row tiles) on the LDS-staged kernels: at B = 8 the static plan takes the
    row-Winograd tiles there, at B = 40 the 2-D Winograd ones.  One sample against the oracle, the batch against that sample alone."""
    ora, net, _ = build_pair(1)
    ora.eval(), net.eval()
    g = torch.Generator().manual_seed(H + W + B)
    x, c = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
    t = torch.rand(B, generator=g) * 0.999 + 1e-3
    k = B // 2
    with torch.no_grad():
        full = net(x.cuda(), t.cuda(), cond_img=c.cuda())
        solo = net(x[k:k + 1].cuda(), t[k:k + 1].cuda(), cond_img=c[k:k + 1].cuda())
        want = ora(x[k:k + 1], t[k:k + 1], cond_img=c[k:k + 1])
    assert torch.isfinite(full).all()
    check_parity(solo.cpu(), want, TOL, f"non-square {H}x{W} solo")
    assert maxrel(full[k:k + 1].cpu(), solo.cpu()) <= 2e-5


def test_full_size_properties_b32_128():
    """BASELINE config 2 shape (B=32, 128x128, 1 condition): size-independent properties instead of an oracle run:
    samples are independent in eval mode (row i of a batched evaluation == the same row evaluated alone),
    evaluation is deterministic, and the output is finite with the 1/sigma(t) scaling applied per sample."""
    ora, net, _ = build_pair(1)
    net.eval()
    g = torch.Generator().manual_seed(42)
    x, c = torch.randn(32, 1, 128, 128, generator=g).cuda(), torch.randn(32, 1, 128, 128, generator=g).cuda()
    t = (torch.rand(32, generator=g) * 0.999 + 1e-3).cuda()
    with torch.no_grad():
        full = net(x, t, cond_img=c)
        again = net(x, t, cond_img=c)
        solo = net(x[5:7], t[5:7], cond_img=c[5:7])
        want = ora.eval()(x[5:6].cpu(), t[5:6].cpu(), cond_img=c[5:6].cpu())
    assert torch.equal(full, again)
    assert torch.isfinite(full).all()
    assert maxrel(full[5:7].cpu(), solo.cpu()) <= 2e-5          # tile/split choices differ with B; values must not
    assert maxrel(full[5:6].cpu(), want) <= TOL


def test_config4_256_pc_b16():
    """BASELINE config 4 shape (256x256, attention over 1024 / 256 / 64 tokens, predictor-corrector, batch 16):
    one sample against the oracle, then the full batch through size-independent properties (per-sample independence,
    graph replay == eager launches bit for bit, finite output)."""
    import sbgm_danra_amd as S
    ora, net, _ = build_pair(1)
    net.eval()
    g = torch.Generator().manual_seed(256)
    x, c = torch.randn(16, 1, 256, 256, generator=g).cuda(), torch.randn(16, 1, 256, 256, generator=g).cuda()
    t = (torch.rand(16, generator=g) * 0.999 + 1e-3).cuda()
    with torch.no_grad():
        full = net(x, t, cond_img=c)
        solo = net(x[3:4], t[3:4], cond_img=c[3:4])
        want = ora.eval()(x[3:4].cpu(), t[3:4].cpu(), cond_img=c[3:4].cpu())
    assert torch.isfinite(full).all()
    assert maxrel(solo.cpu(), want) <= TOL
    assert maxrel(full[3:4].cpu(), solo.cpu()) <= 2e-5
    kw = dict(batch_size=16, num_steps=2, device="cuda", img_size=256, cond_img=c, seed=5)
    a = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, use_graph=True, **kw)
    b = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, use_graph=False, **kw)
    assert torch.equal(a, b) and torch.isfinite(a).all() and a.shape == (16, 1, 256, 256)


def test_errors_are_loud():
    import sbgm_danra_amd as S
    from sbgm_danra_amd._native import NativeError
    _, net, _ = build_pair(1)
    net.eval()
    with torch.no_grad():
        with pytest.raises(NativeError):
            net(torch.randn(1, 1, 32, 32), torch.rand(1), cond_img=torch.randn(1, 1, 32, 32))        # CPU tensors: no fallback
        with pytest.raises(ValueError):
            net(torch.randn(2, 1, 32, 32).cuda(), torch.rand(2).cuda(), cond_img=torch.randn(2, 1, 32, 32).cuda(),
                lsm_cond=torch.randn(3, 2, 32, 32).cuda())                                             # batch mismatch (reference :275)
        with pytest.raises(ValueError):
            net(torch.randn(2, 1, 32, 32).cuda(), torch.rand(2).cuda())                               # missing cond channels
        with pytest.raises(NativeError):
            net(torch.randn(1, 1, 40, 40).cuda(), torch.rand(1).cuda(), cond_img=torch.randn(1, 1, 40, 40).cuda())   # not /32
    out = net(torch.randn(1, 1, 32, 32).cuda(), torch.rand(1).cuda(), cond_img=torch.randn(1, 1, 32, 32).cuda())
    assert out.requires_grad          # eval-mode forward with grad enabled builds a graph, like the reference (test_gpu_loss.py checks the values)
    net.train()
    loss = S.loss_fn(net, torch.randn(2, 1, 32, 32).cuda(), S.marginal_prob_std_fn, cond_img=torch.randn(2, 1, 32, 32).cuda())
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all()
                                        for k, p in net.named_parameters() if not k.startswith("decoder.final_layer.time_"))


def test_ode_sampler_matches_reference_golden(golden_dir):
    """ode_sampler (reference score_sampling.py:239-300): scipy RK45 around native network evaluations, start z injected.
    The adaptive solver amplifies fp32-level differences of the right-hand side up to its own tolerance, so the endpoint is held to
    20 x the solver tolerance (measured: 1.6e-5 at rtol = atol = 1e-5, 5e-4 .. 4e-3 at 1e-3, depending on the convolution tiles'
    summation order) and the evaluation count to 2 % (golden: 974 evaluations at 1e-5, 224 at 1e-3)."""
    import sbgm_danra_amd as S
    g = load_golden(os.path.join(golden_dir, "ode_b2_32.npz"))
    _, net, _ = build_pair(0)
    net.eval()
    for tol, tag in ((1e-3, "tol1e-3"), (1e-5, "tol1e-5")):
        got, nfev = S.ode_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=2, device="cuda", z=g["z"].cuda(),
                                  atol=tol, rtol=tol, return_nfev=True)
        want, nref = g[f"x_{tag}"], int(g[f"nfev_{tag}"])
        err = maxrel(got.cpu(), want)
        print(f"ode_sampler {tag}: max-rel {err:.2e}, nfev {nfev} (reference {nref})")
        assert got.dtype == torch.float64 and got.shape == (2, 1, 32, 32)       # res.y is float64 in the reference too (:297)
        assert err <= 20 * tol and abs(nfev - nref) <= max(6, 0.02 * nref)
    # default start: 32x32 draws scaled by marginal_prob_std(1), like the reference (:279-281)
    torch.manual_seed(3)
    out = S.ode_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=1, device="cuda", atol=1e-2, rtol=1e-2)
    assert out.shape == (1, 1, 32, 32) and torch.isfinite(out).all()


def test_pc_sampler_with_train_mode_batchnorm_matches_reference_golden(golden_dir):
    """literal launch_generation behaviour (evaluate_sbgm/generation.py:47 never calls .eval()): BatchNorm batch statistics
    inside the sampler, B = 4 at 64x64 (well-conditioned statistics, so the usual sampler tolerance applies)"""
    import sbgm_danra_amd as S
    g = load_golden(os.path.join(golden_dir, "pc_trainbn_b4_64_2steps.npz"))
    _, net, _ = build_pair(1)
    net.train()
    with torch.no_grad():
        got = S.pc_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=4, num_steps=2, device="cuda", img_size=64,
                           cond_img=g["cond_img"].cuda(), noise=g["noise"])
    err = maxrel(got.cpu(), g["x_mean"])
    print(f"train-mode BatchNorm PC sampler (B=4, 64x64, 2 steps): max-rel {err:.2e}")
    assert err <= 1e-3
    sd = net.state_dict()
    assert maxrel(sd["encoder.bn1.running_mean"].cpu(), g["bn1_running_mean"]) <= 1e-4
    assert maxrel(sd["encoder.bn1.running_var"].cpu(), g["bn1_running_var"]) <= 1e-4
    assert maxrel(sd["encoder.layer4.1.bn2.running_var"].cpu(), g["l4_running_var"]) <= 1e-3
    assert int(sd["encoder.bn1.num_batches_tracked"]) == int(g["num_batches_tracked"]) == 4      # 2 steps x 2 evaluations


def test_sixteen_heads_from_the_reference_sweep_space():
    """num_heads = 16 (reference sweep/run_optuna.py:124-131) gives head dims 8 / 16 / 32 on the 128- / 256- / 512-channel attention
    blocks: forward vs the oracle and the attention parameters' gradients (head dim 8 goes through the generic-dim paths)"""
    import sbgm_danra_amd as S
    from oracle import torch_ref as O
    ora, net, _ = build_pair(1, heads=16)
    g = torch.Generator().manual_seed(16)
    x, c, t = torch.randn(2, 1, 64, 64, generator=g) * 5, torch.randn(2, 1, 64, 64, generator=g), torch.tensor([0.3, 0.8])
    ora.eval(), net.eval()
    with torch.no_grad():
        assert maxrel(net(x.cuda(), t.cuda(), cond_img=c.cuda()).cpu(), ora(x, t, cond_img=c)) <= TOL
    ora.train(), net.train()
    z = torch.randn(2, 1, 64, 64, generator=g)
    O.loss_fn(ora, x, O.marginal_prob_std_fn, cond_img=c, noise=(t, z)).backward()
    S.loss_fn(net, x.cuda(), S.marginal_prob_std_fn, cond_img=c.cuda(), noise=(t.cuda(), z.cuda())).backward()
    po, pn = dict(ora.named_parameters()), dict(net.named_parameters())
    worst = max(maxrel(pn[k].grad.cpu(), po[k].grad) for k in po if ".attention" in k or ".mha." in k)
    assert worst < 1e-4, worst


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_standalone_submodule_calls_match_the_oracle(mode):
    """Encoder.forward / Decoder.forward / DecoderBlock.forward called on their own, NCHW in and out like the reference's modules
    (score_unet.py:247-364, :559-627, :733-758), in eval and train mode; in train mode gradients flow through them"""
    ora, net, _ = build_pair(5, 4)
    g = torch.Generator().manual_seed(44)
    B = 4
    x, cond = torch.randn(B, 1, 64, 64, generator=g) * 3, torch.randn(B, 1, 64, 64, generator=g)
    lsm = torch.cat([(torch.rand(B, 1, 64, 64, generator=g) > 0.5).float(), torch.ones(B, 1, 64, 64)], 1)
    topo = torch.cat([torch.rand(B, 1, 64, 64, generator=g), torch.ones(B, 1, 64, 64)], 1)
    y, t = torch.randint(0, 5, (B,), generator=g), torch.rand(B, generator=g) * 0.9 + 0.05
    ora.train(mode == "train"), net.train(mode == "train")
    ctx = torch.enable_grad() if mode == "train" else torch.no_grad()
    with ctx:
        fo = ora.encoder(x, t, y=y, cond_img=cond, lsm_cond=lsm, topo_cond=topo)
        fn = net.encoder(x.cuda(), t.cuda(), y=y.cuda(), cond_img=cond.cuda(), lsm_cond=lsm.cuda(), topo_cond=topo.cuda())
        assert len(fn) == 5
        for a, b in zip(fn, fo):
            assert a.shape == b.shape and maxrel(a.detach().cpu(), b.detach()) <= TOL
        do = ora.decoder(*fo, t=t)
        dn = net.decoder(*fn, t=t.cuda())
        assert dn.shape == (B, 1, 64, 64) and maxrel(dn.detach().cpu(), do.detach()) <= TOL
        blk_o, blk_n = ora.decoder.residual_layers[2], net.decoder.residual_layers[2]
        bo = blk_o(fo[2], fo[1], t)
        bn = blk_n(fn[2], fn[1], t.cuda())
        assert maxrel(bn.detach().cpu(), bo.detach()) <= TOL
        # a PRECOMPUTED time embedding [B, time_dim] instead of the time vector (reference :604-609): the block's own sinusoidal
        # embedding is skipped, the value goes straight through SiLU -> Linear
        emb = torch.randn(B, blk_o.time_embedding, generator=g)
        emb_n = emb.cuda().requires_grad_(mode == "train")
        bo2 = blk_o(fo[2], fo[1], emb)
        bn2 = blk_n(fn[2], fn[1], emb_n)
        assert maxrel(bn2.detach().cpu(), bo2.detach()) <= TOL
        if mode == "train":                                  # ... and it is differentiable w.r.t. the embedding
            emb_o = emb.clone().requires_grad_(True)
            go, = torch.autograd.grad(blk_o(fo[2].detach(), fo[1].detach(), emb_o).square().mean(), emb_o)
            gn, = torch.autograd.grad(blk_n(fn[2].detach(), fn[1].detach(), emb_n).square().mean(), emb_n)
            assert maxrel(gn.cpu(), go) < 1e-4
        assert maxrel(net.decoder.final_layer(bn.new_zeros(B, 64, 32, 32) + 0.5).cpu(), ora.decoder.final_layer(torch.zeros(B, 64, 32, 32) + 0.5).detach()) <= TOL
    if mode == "train":
        do.square().mean().backward()
        dn.square().mean().backward()
        po, pn = dict(ora.named_parameters()), dict(net.named_parameters())
        # train-mode BatchNorm over 16 values per channel (layer4 is 2x2 here) makes the encoder's gradients ill-conditioned in fp32
        # (test_gpu_configs.py measures that against a float64 oracle); here the point is that gradients FLOW through the calls
        for k, tol in (("decoder.final_layer.conv.weight", 1e-4), ("decoder.residual_layers.1.attention.mha.in_proj_weight", 1e-3),
                       ("encoder.conv1.weight", 5e-2), ("encoder.layer3.0.conv1.weight", 5e-2), ("encoder.label_emb.weight", 5e-2)):
            assert maxrel(pn[k].grad.cpu(), po[k].grad) < tol, k
    with pytest.raises(AssertionError):                      # reference :596-597
        net.decoder.residual_layers[2](fn[2].detach(), fn[2].detach(), t.cuda())
    # a block whose norms were replaced by Identity (what Decoder does to its final layer, :726-730) somewhere else: skip + time
    # projection + activation (+ attention) without the norms
    for i in (2, 1):                                         # block 1 has attention
        bo_, bn_ = ora.decoder.residual_layers[i], net.decoder.residual_layers[i]
        bo_.norm1 = bo_.norm2 = torch.nn.Identity()
        bn_.norm1 = bn_.norm2 = torch.nn.Identity()
        with ctx:
            xo = fo[4 - i].detach().clone().requires_grad_(mode == "train")
            xn = fn[4 - i].detach().clone().requires_grad_(mode == "train")
            yo = bo_(xo, fo[3 - i].detach(), t)
            yn = bn_(xn, fn[3 - i].detach(), t.cuda())
            assert maxrel(yn.detach().cpu(), yo.detach()) <= TOL, i
            if mode == "train":
                go, = torch.autograd.grad(yo.square().mean(), xo)
                gn, = torch.autograd.grad(yn.square().mean(), xn)
                assert maxrel(gn.cpu(), go) < 1e-4, i


@pytest.mark.parametrize("scale,with_skip,resize", [(4, True, True), (1, False, True), (3, True, True), (4, True, False), (3, False, False)])
def test_decoder_block_with_another_upsample_scale(scale, with_skip, resize):
    """DecoderBlock(upsample_scale=s) (reference score_unet.py:420, :467: nn.Upsample(scale_factor=s, bilinear)) called on its own,
    forward and gradients against the oracle's block with the same weights.  The whole-network engine refuses such a block."""
    import sbgm_danra_amd as S
    from oracle import torch_ref as O
    torch.manual_seed(scale)
    kw = dict(upsample_scale=scale, activation=nn.SiLU, compute_attn=False, norm="group", gn_groups=8, use_resize_conv=resize)
    bo = O.DecoderBlock(128, 64, 128, **kw)
    bn = S.DecoderBlock(128, 64, 128, **kw).cuda()
    bn.load_state_dict(bo.state_dict())
    g = torch.Generator().manual_seed(7)
    B, h = 3, 6
    x, t = torch.randn(B, 128, h, h + 2, generator=g), torch.rand(B, generator=g) * 0.9 + 0.05
    skip = torch.randn(B, 64, scale * h, scale * (h + 2), generator=g) if with_skip else None
    xo, xn = x.clone().requires_grad_(True), x.cuda().requires_grad_(True)
    yo = bo(xo, skip, t)
    yn = bn(xn, None if skip is None else skip.cuda(), t.cuda())
    assert yn.shape == yo.shape == (B, 64, scale * h, scale * (h + 2))
    assert maxrel(yn.detach().cpu(), yo.detach()) <= TOL
    yo.square().mean().backward()
    yn.square().mean().backward()
    assert maxrel(xn.grad.cpu(), xo.grad) < 1e-4
    po, pn = dict(bo.named_parameters()), dict(bn.named_parameters())
    for k in ("conv_up.weight" if resize else "transpose.weight", "conv.weight", "norm1.weight", "time_projection_layer.1.weight"):
        assert maxrel(pn[k].grad.cpu(), po[k].grad) < 1e-4, k
    with pytest.raises(NotImplementedError):
        S.DecoderBlock(64, 32, 128, upsample_scale=0)
    if scale == 4 and resize:
        _, net, _ = build_pair(1)
        net.decoder.residual_layers[3] = S.DecoderBlock(64, 64, 256, upsample_scale=4, activation=nn.SiLU, compute_attn=False,
                                                        norm="group", gn_groups=8).cuda()
        args = (torch.zeros(1, 1, 64, 64).cuda(), torch.tensor([0.5]).cuda())
        with pytest.raises(NotImplementedError), torch.no_grad():       # the whole-network engine (sampling path) runs x2 blocks only
            net.eval()(*args, cond_img=torch.zeros(1, 1, 64, 64).cuda())
        with pytest.raises(AssertionError):                             # the autograd path runs the block; its skip no longer fits (:596-597)
            net.eval()(*args, cond_img=torch.zeros(1, 1, 64, 64).cuda())


def test_cached_step_graph_is_reused_across_runs_with_other_seeds_and_lengths():
    """the captured SDE step is kept in the model handle: a later run with another seed / num_steps / cond CONTENT (same tensors)
    replays it — its result must equal eager launches of that run bit for bit; a run with other tensors re-captures"""
    import sbgm_danra_amd as S
    _, net, _ = build_pair(1)
    net.eval()
    g = torch.Generator().manual_seed(21)
    cond = torch.randn(2, 1, 64, 64, generator=g).cuda()
    for sampler in (S.Euler_Maruyama_sampler, S.pc_sampler):
        kw = dict(batch_size=2, device="cuda", img_size=64, cond_img=cond)
        sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=3, seed=9, use_graph=True, **kw)        # captures
        cond.copy_(torch.randn(2, 1, 64, 64, generator=g))                                                            # same tensor, new content
        a = sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=7, seed=1, use_graph=True, **kw)    # replays the cached step
        b = sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=7, seed=1, use_graph=False, **kw)
        assert torch.isfinite(a).all() and torch.equal(a, b)
        cond2 = cond.clone()
        c = sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=7, seed=1, use_graph=True, **dict(kw, cond_img=cond2))
        assert torch.equal(c, a)
        d = sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, num_steps=7, seed=2, use_graph=True, **kw)
        assert not torch.equal(d, a)


def test_replaced_submodule_is_seen_by_the_engine():
    """The engine keeps a repacked copy of the weights and checks a cached module list for changes on every call (0.1 ms).  A
    sub-module replaced AFTER the first forward (same parameter shapes, other values) must be picked up: the key includes the
    identity of every module's children."""
    import copy
    _, net, _ = build_pair(1)
    net.eval()
    g = torch.Generator().manual_seed(5)
    x, c, t = torch.randn(1, 1, 64, 64, generator=g).cuda(), torch.randn(1, 1, 64, 64, generator=g).cuda(), torch.tensor([0.4]).cuda()
    with torch.no_grad():
        y0 = net(x, t, cond_img=c).clone()
        new_block = copy.deepcopy(net.decoder.residual_layers[3])
        for p in new_block.parameters():
            p.mul_(1.5)
        net.decoder.residual_layers[3] = new_block           # a different module object with different weights
        y1 = net(x, t, cond_img=c)
    assert maxrel(y1.cpu(), y0.cpu()) > 1e-3


def test_workspace_is_trimmed_to_the_measured_high_water_mark():
    """C2 shape (B = 32, 128x128): the first evaluation runs on the generous bound (1 Ki floats per input pixel = 2.2 GB), later calls
    on the measured high-water mark; the handle must end below 0.6 GB and the outputs must not change by a bit"""
    from sbgm_danra_amd import _native as N
    _, net, _ = build_pair(1)
    net.eval()
    g = torch.Generator().manual_seed(8)
    x, c = torch.randn(32, 1, 128, 128, generator=g).cuda(), torch.randn(32, 1, 128, 128, generator=g).cuda()
    t = (torch.rand(32, generator=g) * 0.9 + 0.05).cuda()
    sizes, outs = [], []
    with torch.no_grad():
        for _ in range(3):
            outs.append(net(x, t, cond_img=c).clone())
            sizes.append(N.lib().sbgm_model_workspace_bytes(net._engine(None, None, c).h))
    print("workspace bytes after calls 1..3:", sizes)
    assert sizes[0] > 2e9 and sizes[-1] < 0.6e9
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # a sampler run at the same shape still fits (its persistent slabs sit on top of the evaluation's share)
    import sbgm_danra_amd as S
    s = S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=32, num_steps=3, device="cuda", img_size=128,
                                 cond_img=c, seed=3)
    assert torch.isfinite(s).all() and N.lib().sbgm_model_workspace_bytes(net._engine(None, None, c).h) < 0.6e9


def test_sampler_settles_its_workspace_before_the_first_capture():
    """A shape the handle has not seen is measured by one extra evaluation BEFORE the first sampler call captures its step graph, so
    the workspace has its final size from call 1 on (a later trim would free the slab the captured graph points into and force a second
    capture: the slow second call of a fresh process); results of consecutive calls with one seed are bit-identical"""
    from sbgm_danra_amd import _native as N
    import sbgm_danra_amd as S
    _, net, _ = build_pair(1)
    net.eval()
    g = torch.Generator().manual_seed(9)
    c = torch.randn(8, 1, 64, 64, generator=g).cuda()
    sizes, outs = [], []
    for _ in range(3):
        outs.append(S.Euler_Maruyama_sampler(net, S.marginal_prob_std_fn, S.diffusion_coeff_fn, batch_size=8, num_steps=3, device="cuda",
                                             img_size=64, cond_img=c, seed=5).clone())
        sizes.append(N.lib().sbgm_model_workspace_bytes(net._engine(None, None, c).h))
    print("workspace bytes after sampler calls 1..3:", sizes)
    assert sizes[0] == sizes[1] == sizes[2]          # (a surplus below 256 MB is not worth a reallocation: small shapes keep the bound)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
