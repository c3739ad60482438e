"""C3-shape gradient diagnostics (GPU): where does the native-vs-float64 gradient error of the encoder live?
(a) synthetic hash weights (tests/util_models.build_pair): per OUTPUT CHANNEL error of encoder.layer{2,3}.* conv weight gradients against
    that channel's BatchNorm batch variance / post-ReLU active fraction (float64 oracle forward hooks);
(b) the reference's training initialisation (torch default init + xavier_uniform on Conv2d, bias 0.01, seed 42; reference
    training.py:188-201): every gradient, native vs fp32 oracle and vs float64."""
import copy, json, os, sys
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # .../tests (this probe is test infrastructure: it uses the oracle)
sys.path.insert(0, os.path.dirname(_TESTS))
sys.path.insert(0, _TESTS)
import torch, torch.nn as nn
from util_models import build_pair, maxrel
import sbgm_danra_amd as S
from oracle import torch_ref as O

torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
g = torch.Generator().manual_seed(333)
B = 8
x, cond = torch.randn(B, 1, 128, 128, generator=g), torch.randn(B, 4, 128, 128, generator=g)
t, z = torch.rand(B, generator=g) * 0.999 + 1e-3, torch.randn(B, 1, 128, 128, generator=g)

def run(ora, net, tag):
    ora.train(); net.train()
    ora64 = copy.deepcopy(ora).double()
    stats = {}
    def hook(name):
        def f(mod, inp, out):
            v = inp[0].detach()
            stats[name] = (v.var(dim=(0, 2, 3), unbiased=False), (torch.relu(out.detach()) > 0).double().mean(dim=(0, 2, 3)))
        return f
    for n, m in ora64.named_modules():
        if isinstance(m, nn.BatchNorm2d):
            m.register_forward_hook(hook(n))
    lo = O.loss_fn(ora, x, O.marginal_prob_std_fn, cond_img=cond, noise=(t, z)); lo.backward()
    l64 = O.loss_fn(ora64, x.double(), O.marginal_prob_std_fn, cond_img=cond.double(), noise=(t, z.double())); l64.backward()
    ln = S.loss_fn(net, x.cuda(), S.marginal_prob_std_fn, cond_img=cond.cuda(), noise=(t.cuda(), z.cuda())); ln.backward()
    po, p64, pn = dict(ora.named_parameters()), dict(ora64.named_parameters()), dict(net.named_parameters())
    out = {"loss_rel": abs(float(ln) / float(l64) - 1)}
    errs = {k: (maxrel(pn[k].grad.cpu().double(), p.grad), maxrel(po[k].grad.double(), p.grad), maxrel(pn[k].grad.cpu(), po[k].grad))
            for k, p in p64.items() if p.grad is not None}
    out["n"] = len(errs)
    out["n_native_over_1e-4_vs_f64"] = sum(1 for v in errs.values() if v[0] > 1e-4)
    out["n_fp32oracle_over_1e-4_vs_f64"] = sum(1 for v in errs.values() if v[1] > 1e-4)
    out["n_native_over_1e-4_vs_fp32oracle"] = sum(1 for v in errs.values() if v[2] > 1e-4)
    out["worst_native_vs_f64"] = sorted(((v[0], k) for k, v in errs.items()), reverse=True)[:6]
    out["worst_native_vs_fp32"] = sorted(((v[2], k) for k, v in errs.items()), reverse=True)[:6]
    print(f"== {tag}: loss rel {out['loss_rel']:.2e}; {out['n']} gradients; native>1e-4 vs f64: {out['n_native_over_1e-4_vs_f64']}, fp32 oracle>1e-4 vs f64: "
          f"{out['n_fp32oracle_over_1e-4_vs_f64']}, native>1e-4 vs fp32 oracle: {out['n_native_over_1e-4_vs_fp32oracle']}")
    print("   worst native vs f64:", [(f"{e:.1e}", k) for e, k in out["worst_native_vs_f64"]])
    print("   worst native vs fp32 oracle:", [(f"{e:.1e}", k) for e, k in out["worst_native_vs_fp32"]])
    # per-channel localisation: conv weight gradients of BasicBlocks, per output channel, vs the variance of the BN that follows
    chan = {}
    for k, p in p64.items():
        if p.grad is None or p.dim() != 4 or not k.startswith("encoder.layer"):
            continue
        bn = k.replace("conv1.weight", "bn1").replace("conv2.weight", "bn2").replace("downsample.0.weight", "downsample.1")
        if bn not in stats:
            continue
        var, act = stats[bn]
        gn, g6 = pn[k].grad.cpu().double(), p.grad
        scale = g6.abs().max()
        e_c = (gn - g6).abs().flatten(1).max(1).values / scale
        eo_c = (po[k].grad.double() - g6).abs().flatten(1).max(1).values / scale
        small = var <= 1e-6 * var.mean()
        chan[k] = dict(n_small=int(small.sum()), n=int(var.numel()), err_small=float(e_c[small].max()) if small.any() else 0.0,
                       err_rest=float(e_c[~small].max()), oracle_err_small=float(eo_c[small].max()) if small.any() else 0.0,
                       oracle_err_rest=float(eo_c[~small].max()), min_var=float(var.min()), mean_var=float(var.mean()),
                       min_active=float(act.min()), dead=int((act == 0).sum()))
    for k, v in chan.items():
        if v["err_rest"] > 1e-5 or v["err_small"] > 1e-5:
            print(f"   {k}: channels with var<=1e-6*mean: {v['n_small']}/{v['n']}  native err there {v['err_small']:.1e} / elsewhere {v['err_rest']:.1e}; "
                  f"fp32 oracle {v['oracle_err_small']:.1e} / {v['oracle_err_rest']:.1e}; min var {v['min_var']:.1e} (mean {v['mean_var']:.1e}); dead channels {v['dead']}")
    out["channels"] = chan
    return out

res = {}
ora, net, _ = build_pair(4)
res["synthetic"] = run(ora, net, "synthetic hash weights")
torch.manual_seed(42)
ora2 = O.build_scorenet(4)
def xavier(m):
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            m.bias.data.fill_(0.01)
ora2.apply(xavier)
_, net2, _ = build_pair(4)
net2.load_state_dict(ora2.state_dict())
res["training_init"] = run(ora2, net2, "reference training initialisation (seed 42)")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/c3_grad_probe.json", "w"), indent=1, default=str)
