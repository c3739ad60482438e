"""ORACLE tooling — runs ONLY in the build container (needs /root/reference); never on the GPU box.

1. Imports the reference's own `sbgm.score_unet` / `sbgm.score_sampling` on PyTorch-CPU.  torchvision
   is absent from the image, so the build-owned restatement of its two public classes
   (oracle/_tv_standin) is put on sys.path for this script only.
2. Checks oracle/torch_ref.py (the CPU restatement that travels to the GPU box) against the reference on
   every case below — forward (eval + train BN), loss + gradients, EM and PC samplers, schedule fns.
3. Writes the golden vectors under tests/golden/ (inputs + expected outputs only; no reference text).

Weights are not committed: both models are filled by torch_ref.synth_state_dict (a hash of the tensor
name), and a per-tensor checksum list is stored instead.

Usage:  python oracle/make_goldens.py            (rewrites tests/golden/*.npz, *.json)
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)                                   # for `oracle`
sys.path.insert(0, os.path.join(HERE, "_tv_standin"))
sys.path.insert(0, REF)                                    # FIRST: `sbgm` must resolve to the reference, not the repo's alias

import torch.nn as nn                                                   # noqa: E402
import sbgm.score_unet as R                                             # noqa: E402  (the reference)
import sbgm.score_sampling as RS                                        # noqa: E402
from oracle import torch_ref as O                                        # noqa: E402

assert R.__file__.startswith(REF) and RS.__file__.startswith(REF)
GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(8)
CPU = torch.device("cpu")


def build_ref(n_cond, num_classes, heads=4, temb=256, layers=(2, 2, 2, 2), resize=True):
    enc = R.Encoder(input_channels=n_cond, time_embedding=temb, block_layers=list(layers),
                    num_classes=num_classes, n_heads=heads)
    dec = R.Decoder(last_fmap_channels=512, output_channels=1, time_embedding=temb, n_heads=heads,
                    use_resize_conv=resize, norm="group", gn_groups=8, activation=nn.SiLU)
    return R.ScoreNet(R.marginal_prob_std_fn, enc, dec, device=CPU, debug_pre_sigma_div=False)


def pair(n_cond, num_classes, **kw):
    ref = build_ref(n_cond, num_classes, **kw)
    ora = O.build_scorenet(n_cond, num_classes=num_classes,
                           time_embedding=kw.get("temb", 256), n_heads=kw.get("heads", 4),
                           block_layers=kw.get("layers", (2, 2, 2, 2)), use_resize_conv=kw.get("resize", True))
    rk = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    ok = {k: tuple(v.shape) for k, v in ora.state_dict().items()}
    assert rk == ok, set(rk.items()) ^ set(ok.items())
    sd = O.synth_state_dict(ora)
    ref.load_state_dict(sd)
    ora.load_state_dict(sd)
    return ref, ora, sd


def inputs(seed, b, hw, n_lr, geo, classes):
    g = torch.Generator().manual_seed(seed)
    d = dict(x=torch.randn(b, 1, hw, hw, generator=g), t=torch.rand(b, generator=g) * 0.999 + 1e-3)
    d["cond_img"] = torch.randn(b, n_lr, hw, hw, generator=g) if n_lr else None
    if geo:
        lsm = (torch.rand(b, 1, hw, hw, generator=g) > 0.5).float()
        topo = torch.rand(b, 1, hw, hw, generator=g)
        d["lsm_cond"] = torch.cat([lsm, torch.ones_like(lsm)], 1)
        d["topo_cond"] = torch.cat([topo, torch.ones_like(topo)], 1)
    else:
        d["lsm_cond"] = d["topo_cond"] = None
    d["y"] = torch.randint(1, classes + 1, (b,), generator=g) if classes else None
    return d


def maxrel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def npz(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()
            if v is not None}


report = {}

# ---------------------------------------------------------------------------------- schedule fns
tg = torch.tensor([1e-5, 1e-3, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0])
assert torch.equal(R.marginal_prob_std_fn(tg), O.marginal_prob_std_fn(tg))
assert torch.equal(R.diffusion_coeff_fn(tg), O.diffusion_coeff_fn(tg))
np.savez_compressed(os.path.join(GOLD, "schedule.npz"), t=tg.numpy(),
                    std=R.marginal_prob_std_fn(tg).numpy(), g=R.diffusion_coeff_fn(tg).numpy())

# ---------------------------------------------------------------------------------- forward cases
CASES = {
    # name: (B, HW, n_lr, geo, classes)
    "fwd_b2_64_c2": (2, 64, 1, False, 0),
    "fwd_b1_128_c7_y": (1, 128, 2, True, 4),
    "fwd_b2_32_c1": (2, 32, 0, False, 0),
}
state_manifest = {}
for name, (b, hw, n_lr, geo, classes) in CASES.items():
    n_cond = n_lr + (4 if geo else 0)
    ref, ora, sd = pair(n_cond, classes or None)
    state_manifest[f"ncond{n_cond}_cls{classes}"] = {
        k: [list(v.shape), hashlib.sha256(v.numpy().tobytes()).hexdigest()[:16]] for k, v in sd.items()}
    inp = inputs(1234 + hw, b, hw, n_lr, geo, classes)
    out = {}
    for mode in ("eval", "train"):
        for m in (ref, ora):
            m.load_state_dict(sd)             # reset BN running stats
            m.train(mode == "train")
        with torch.no_grad():
            fr = ref.encoder(inp["x"], inp["t"], y=inp["y"], cond_img=inp["cond_img"],
                             lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"])
        ref.load_state_dict(sd)
        ref.train(mode == "train")
        with torch.no_grad():
            yr = ref(inp["x"], inp["t"], inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
            yo = ora(inp["x"], inp["t"], inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"])
        report[f"{name}/{mode}"] = maxrel(yo, yr)
        assert maxrel(yo, yr) <= 1e-6, (name, mode, maxrel(yo, yr))
        out[f"score_{mode}"] = yr
        for i, f in enumerate(fr):
            flat = f.permute(0, 2, 3, 1).reshape(-1)          # NHWC order, strided subsample
            out[f"fmap{i + 1}_{mode}_sub"] = flat[:: max(1, flat.numel() // 4096)][:4096].clone()
            out[f"fmap{i + 1}_{mode}_absmean"] = f.abs().mean()
    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), **npz({**inp, **out}))

# ---------------------------------------------------------------------------------- loss + grads
ref, ora, sd = pair(1, None)
inp = inputs(77, 2, 64, 1, False, 0)
g = torch.Generator().manual_seed(5)
sdf = torch.rand(2, 1, 64, 64, generator=g)
PROBE = ["encoder.conv1.weight", "encoder.layer2.0.downsample.0.weight", "encoder.bn1.weight",
         "encoder.attention_layers.4.mha.in_proj_weight", "decoder.residual_layers.1.norm2.bias",
         "decoder.final_layer.conv.weight", "decoder.residual_layers.3.time_projection_layer.1.weight"]
res = {}
for m, mod in ((ref, R), (ora, O)):
    m.load_state_dict(sd)
    m.train()
    m.zero_grad()
    torch.manual_seed(99)
    L = mod.loss_fn(m, inp["x"], mod.marginal_prob_std_fn, cond_img=inp["cond_img"], sdf_cond=sdf)
    L.backward()
    p = dict(m.named_parameters())
    res[mod.__name__] = (L.detach(), {k: p[k].grad.clone() for k in PROBE})
(Lr, gr), (Lo, go) = res[R.__name__], res[O.__name__]
report["loss"] = abs(float(Lr - Lo) / float(Lr))
assert report["loss"] <= 1e-6
for k in PROBE:
    report[f"grad/{k}"] = maxrel(go[k], gr[k])
    assert report[f"grad/{k}"] <= 1e-5, (k, report[f"grad/{k}"])
torch.manual_seed(99)
t_used = torch.rand(2) * (1.0 - 1e-3) + 1e-3
z_used = torch.randn_like(inp["x"])
gsub = {f"grad::{k}": gr[k].reshape(-1)[:: max(1, gr[k].numel() // 2048)][:2048].clone() for k in PROBE}
np.savez_compressed(os.path.join(GOLD, "loss_b2_64.npz"),
                    **npz(dict(x=inp["x"], cond_img=inp["cond_img"], sdf=sdf, t=t_used, z=z_used, loss=Lr, **gsub)))

# ---------------------------------------------------------------------------------- samplers
for m in (ref, ora):
    m.load_state_dict(sd)
    m.eval()
cond = inputs(31, 2, 64, 1, False, 0)["cond_img"]
torch.manual_seed(7)
xr = RS.pc_sampler(ref, R.marginal_prob_std_fn, R.diffusion_coeff_fn, batch_size=2, num_steps=3, device="cpu",
                   img_size=64, cond_img=cond)
torch.manual_seed(7)
xo = O.pc_sampler(ora, O.marginal_prob_std_fn, O.diffusion_coeff_fn, batch_size=2, num_steps=3, device="cpu",
                  img_size=64, cond_img=cond)
report["pc_sampler"] = maxrel(xo, xr)
assert report["pc_sampler"] <= 1e-6
torch.manual_seed(7)
noise = [torch.randn(2, 1, 64, 64)] + [torch.randn(2, 1, 64, 64) for _ in range(6)]
np.savez_compressed(os.path.join(GOLD, "pc_b2_64_3steps.npz"),
                    **npz(dict(cond_img=cond, noise=torch.stack(noise), x_mean=xr)))

cond32 = inputs(32, 2, 32, 1, False, 0)["cond_img"]
torch.manual_seed(8)
xr = RS.Euler_Maruyama_sampler(ref, R.marginal_prob_std_fn, R.diffusion_coeff_fn, batch_size=2, num_steps=5,
                               device="cpu", cond_img=cond32)
torch.manual_seed(8)
xo = O.Euler_Maruyama_sampler(ora, O.marginal_prob_std_fn, O.diffusion_coeff_fn, batch_size=2, num_steps=5,
                              device="cpu", cond_img=cond32)
report["em_sampler"] = maxrel(xo, xr)
assert report["em_sampler"] <= 1e-6
torch.manual_seed(8)
noise = [torch.randn(2, 1, 32, 32) for _ in range(6)]
np.savez_compressed(os.path.join(GOLD, "em_b2_32_5steps.npz"),
                    **npz(dict(cond_img=cond32, noise=torch.stack(noise), mean_x=xr)))

# CFG combine
inp = inputs(55, 2, 32, 1, True, 4)
ref, ora, sd = pair(5, 4)
ref.eval(), ora.eval()
with torch.no_grad():
    a = RS.guided_score_fn(ref, inp["x"], inp["t"], inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"], scale=1.5)
    b_ = O.guided_score_fn(ora, inp["x"], inp["t"], inp["y"], inp["cond_img"], inp["lsm_cond"], inp["topo_cond"], scale=1.5)
report["cfg"] = maxrel(b_, a)
assert report["cfg"] <= 1e-6
np.savez_compressed(os.path.join(GOLD, "cfg_b2_32_c6_y.npz"), **npz({**inp, "guided": a}))

# samplers with live classifier-free guidance (score_sampling.py:105-118, :179-222): 6 conditioning channels + class labels
gcfg = {"classifier_free_guidance": {"enabled": True, "guidance_scale": 2.5, "guidance_scale_max": 1.5}}
kw = dict(y=inp["y"], cond_img=inp["cond_img"], lsm_cond=inp["lsm_cond"], topo_cond=inp["topo_cond"], cfg=gcfg)
torch.manual_seed(9)
xr = RS.pc_sampler(ref, R.marginal_prob_std_fn, R.diffusion_coeff_fn, batch_size=2, num_steps=2, device="cpu", img_size=32, **kw)
torch.manual_seed(9)
xo = O.pc_sampler(ora, O.marginal_prob_std_fn, O.diffusion_coeff_fn, batch_size=2, num_steps=2, device="cpu", img_size=32, **kw)
report["pc_sampler_cfg"] = maxrel(xo, xr)
assert report["pc_sampler_cfg"] <= 1e-6
torch.manual_seed(9)
noise = [torch.randn(2, 1, 32, 32) for _ in range(5)]
np.savez_compressed(os.path.join(GOLD, "pc_cfg_b2_32_2steps.npz"),
                    **npz({**{k: inp[k] for k in ("y", "cond_img", "lsm_cond", "topo_cond")}, "noise": torch.stack(noise), "x_mean": xr}))
torch.manual_seed(10)
xr = RS.Euler_Maruyama_sampler(ref, R.marginal_prob_std_fn, R.diffusion_coeff_fn, batch_size=2, num_steps=3, device="cpu", **kw)
torch.manual_seed(10)
xo = O.Euler_Maruyama_sampler(ora, O.marginal_prob_std_fn, O.diffusion_coeff_fn, batch_size=2, num_steps=3, device="cpu", **kw)
report["em_sampler_cfg"] = maxrel(xo, xr)
assert report["em_sampler_cfg"] <= 1e-6
torch.manual_seed(10)
noise = [torch.randn(2, 1, 32, 32) for _ in range(4)]
np.savez_compressed(os.path.join(GOLD, "em_cfg_b2_32_3steps.npz"),
                    **npz({**{k: inp[k] for k in ("y", "cond_img", "lsm_cond", "topo_cond")}, "noise": torch.stack(noise), "mean_x": xr}))

# ---- ode_sampler (score_sampling.py:239-300): scipy RK45 over the flattened state, unconditional 1-channel model, injected start z
ref0, ora0, sd0 = pair(0, None)
ref0.eval(), ora0.eval()
gz = torch.Generator().manual_seed(123)
z_ode = torch.randn(2, 1, 32, 32, generator=gz) * R.marginal_prob_std_fn(torch.ones(2))[:, None, None, None]
ode_out = {"z": z_ode}
for tol, tag in ((1e-5, "tol1e-5"), (1e-3, "tol1e-3")):
    xr = RS.ode_sampler(ref0, R.marginal_prob_std_fn, R.diffusion_coeff_fn, batch_size=2, device="cpu", z=z_ode, atol=tol, rtol=tol)
    xo, nfev = O.ode_sampler(ora0, O.marginal_prob_std_fn, O.diffusion_coeff_fn, batch_size=2, device="cpu", z=z_ode, atol=tol, rtol=tol,
                             return_nfev=True)
    report[f"ode_sampler/{tag}"] = maxrel(xo, xr)
    assert torch.equal(xo, xr) and xr.dtype == torch.float64, tag
    ode_out[f"x_{tag}"] = xr
    ode_out[f"nfev_{tag}"] = np.int64(nfev)
np.savez_compressed(os.path.join(GOLD, "ode_b2_32.npz"), **npz(ode_out))

# ---- literal launch_generation behaviour (evaluate_sbgm/generation.py:47 never calls .eval()): PC sampling with BatchNorm in
# TRAIN mode at a batch large enough for well-conditioned batch statistics (B = 4 at 64x64: >= 16 values per channel) ----------
ref, ora, sd = pair(1, None)
for m in (ref, ora):
    m.load_state_dict(sd)
    m.train()
cond4 = inputs(64, 4, 64, 1, False, 0)["cond_img"]
torch.manual_seed(17)
xr = RS.pc_sampler(ref, R.marginal_prob_std_fn, R.diffusion_coeff_fn, batch_size=4, num_steps=2, device="cpu", img_size=64, cond_img=cond4)
torch.manual_seed(17)
xo = O.pc_sampler(ora, O.marginal_prob_std_fn, O.diffusion_coeff_fn, batch_size=4, num_steps=2, device="cpu", img_size=64, cond_img=cond4)
report["pc_sampler_train_bn"] = maxrel(xo, xr)
assert report["pc_sampler_train_bn"] <= 1e-6
torch.manual_seed(17)
noise = [torch.randn(4, 1, 64, 64) for _ in range(5)]
rs = ref.state_dict()
np.savez_compressed(os.path.join(GOLD, "pc_trainbn_b4_64_2steps.npz"),
                    **npz(dict(cond_img=cond4, noise=torch.stack(noise), x_mean=xr,
                               bn1_running_mean=rs["encoder.bn1.running_mean"], bn1_running_var=rs["encoder.bn1.running_var"],
                               l4_running_var=rs["encoder.layer4.1.bn2.running_var"],
                               num_batches_tracked=rs["encoder.bn1.num_batches_tracked"])))

# ---- ConvTranspose2d decoder (model.use_resize_conv = false, score_unet.py:470-475): forward, loss and two gradients -------
ref, ora, sd = pair(1, None, resize=False)
tinp = inputs(4242, 2, 64, 1, False, 0)
tout = {}
for m in (ref, ora):
    m.load_state_dict(sd)
    m.eval()
with torch.no_grad():
    yr = ref(tinp["x"], tinp["t"], None, tinp["cond_img"])
    yo = ora(tinp["x"], tinp["t"], None, tinp["cond_img"])
report["transpose_decoder/eval"] = maxrel(yo, yr)
assert report["transpose_decoder/eval"] <= 1e-6
TPROBE = ["decoder.residual_layers.0.transpose.weight", "decoder.final_layer.transpose.bias", "encoder.conv2.weight"]
tres = {}
for m, mod in ((ref, R), (ora, O)):
    m.load_state_dict(sd)
    m.train()
    m.zero_grad()
    torch.manual_seed(41)
    L = mod.loss_fn(m, tinp["x"], mod.marginal_prob_std_fn, cond_img=tinp["cond_img"])
    L.backward()
    pp = dict(m.named_parameters())
    tres[mod.__name__] = (L.detach(), {k: pp[k].grad.clone() for k in TPROBE})
report["transpose_decoder/loss"] = abs(float(tres[R.__name__][0] - tres[O.__name__][0]) / float(tres[R.__name__][0]))
assert report["transpose_decoder/loss"] <= 1e-6
for k in TPROBE:
    report[f"transpose_decoder/grad/{k}"] = maxrel(tres[O.__name__][1][k], tres[R.__name__][1][k])
    assert report[f"transpose_decoder/grad/{k}"] <= 1e-5
torch.manual_seed(41)
tt = torch.rand(2) * (1.0 - 1e-3) + 1e-3
tz = torch.randn_like(tinp["x"])
np.savez_compressed(os.path.join(GOLD, "transpose_decoder_b2_64.npz"),
                    **npz({"x": tinp["x"], "t": tinp["t"], "cond_img": tinp["cond_img"], "score_eval": yr, "loss": tres[R.__name__][0],
                           "t_used": tt, "z_used": tz,
                           **{f"grad_sub::{k}": tres[R.__name__][1][k].reshape(-1)[:: (257 if tres[R.__name__][1][k].numel() > 4096 else 1)][:4096].clone()
                              for k in TPROBE}}))
state_manifest["ncond1_cls0_transpose"] = {k: [list(v.shape), hashlib.sha256(v.numpy().tobytes()).hexdigest()[:16]] for k, v in sd.items()}

# ---- transforms / back-transforms: reference sbgm/special_transforms.py, imported as is --------------
import sbgm.special_transforms as RT                                     # noqa: E402
from oracle import transforms_ref as OT                                  # noqa: E402
assert RT.__file__.startswith(REF)
tg = torch.Generator().manual_seed(1234)
z = torch.randn(3, 1, 24, 20, generator=tg) * 1.7                       # model-space field (z-scored / scaled)
phys = torch.rand(3, 1, 24, 20, generator=tg) ** 4 * 80.0               # physical precipitation-like field, >= 0
LOGP = dict(glob_mean_log=-1.2345, glob_std_log=2.0321, glob_min_log=-4.60517, glob_max_log=5.7038, buffer_frac=0.5)
TR_CASES = {
    "zscore_back": (lambda M: M.ZScoreBackTransform(8.7012, 6.1923), z),
    "zscore_fwd": (lambda M: M.ZScoreTransform(8.7012, 6.1923), phys),
    "scale_back": (lambda M: M.ScaleBackTransform(0, 1, -23.5, 41.25), z),
    "scale_back_m11": (lambda M: M.ScaleBackTransform(-1, 1, 0.0, 155.3), z),
    "scale_fwd": (lambda M: M.Scale(-1, 1, -23.5, 41.25), phys),
    "log_fwd": (lambda M: M.PrcpLogTransform(scale_type="log", **LOGP), phys),
    "log_back": (lambda M: M.PrcpLogBackTransform(scale_type="log", clamp_log_max=3.0, **LOGP), z),
}
for st in ("log_zscore", "log_01", "log_minus1_1"):
    TR_CASES[st + "_fwd"] = (lambda M, st=st: M.PrcpLogTransform(scale_type=st, **LOGP), phys)
    TR_CASES[st + "_back"] = (lambda M, st=st: M.PrcpLogBackTransform(scale_type=st, **LOGP), z)
    TR_CASES[st + "_back_clamped"] = (lambda M, st=st: M.PrcpLogBackTransform(scale_type=st, clamp_log_min=-4.60517,
                                                                             clamp_log_max=5.7038, **LOGP), z * 3)
tr_out = {"z": z, "phys": phys}
for name, (mk, inp_t) in TR_CASES.items():
    a, b_ = mk(RT)(inp_t.clone()), mk(OT)(inp_t.clone())
    report["transform/" + name] = maxrel(b_, a)
    assert torch.equal(a, b_), name                                      # bit-identical on the same CPU
    tr_out[name] = a
np.savez_compressed(os.path.join(GOLD, "transforms.npz"), **npz(tr_out))

with open(os.path.join(GOLD, "state_manifest.json"), "w") as f:
    json.dump(state_manifest, f, indent=0, sort_keys=True)
with open(os.path.join(GOLD, "oracle_vs_reference.json"), "w") as f:
    json.dump({"torch": torch.__version__, "max_rel_err_oracle_vs_reference": report}, f, indent=1, sort_keys=True)
print(json.dumps(report, indent=1))
