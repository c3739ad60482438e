import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import sbgm_danra_amd as S
from oracle import torch_ref as O
from util_models import build_pair, maxrel
from test_gpu_backward import _batch, _native_loss

def grads(net, b):
    net.zero_grad(set_to_none=True)
    _native_loss(net, S, *b).backward()
    return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}

ora, net, sd = build_pair(5, 4)
net.train()
b = _batch(torch.Generator().manual_seed(12), B=2)
g1 = grads(net, b)
net.load_state_dict(sd)
g2 = grads(net, b)
w = sorted(((maxrel(g2[k], g1[k]), k) for k in g1), reverse=True)[:5]
print("run-to-run nondeterminism:", w)
# 2 sgd steps, repeated 3 times
for rep in range(3):
    ora, net, sd = build_pair(5, 4)
    ora.train(), net.train()
    oo, on = torch.optim.SGD(ora.parameters(), lr=1e-3), torch.optim.SGD(net.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(12)
    for step in range(2):
        b = _batch(gen, B=2)
        x, cond, lsm, topo, y, t, z = b
        oo.zero_grad(); lo = O.loss_fn(ora, x, O.marginal_prob_std_fn, y=y, cond_img=cond, lsm_cond=lsm, topo_cond=topo, noise=(t, z)); lo.backward()
        on.zero_grad(); ln = _native_loss(net, S, *b); ln.backward()
        po, pn = dict(ora.named_parameters()), dict(net.named_parameters())
        ge = sorted(((maxrel(pn[k].grad.cpu(), po[k].grad), k) for k in po if po[k].grad is not None), reverse=True)[:3]
        gmax = max(float(p.grad.abs().max()) for p in po.values() if p.grad is not None)
        print(f"rep {rep} step {step}: loss o={float(lo):.6g} n={float(ln):.6g} t={t.tolist()} gradmax={gmax:.3g} worst grad err {ge}")
        oo.step(); on.step()
    so, sn = ora.state_dict(), net.state_dict()
    pe = sorted(((maxrel(sn[k].cpu().float(), so[k].float()), k) for k in so if so[k].dtype.is_floating_point), reverse=True)[:4]
    print("   params:", pe)
