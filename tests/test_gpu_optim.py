"""GPU suite: the one-launch Adam / AdamW step (csrc/optim.hip behind sbgm_danra_amd.optim) against torch's own
single-tensor step on the CPU — the optimizer the reference builds in training_utils.get_optimizer (:50-59) and steps at
training.py:407.  Tolerance: 2e-6 relative on parameters after 6 steps (fp32 pow/sqrt/division differ in the last bit)."""
import copy

import pytest
import torch

from sbgm_danra_amd import optim as O

pytestmark = pytest.mark.gpu
SHAPES = [(1,), (7,), (33, 5), (64, 64, 3, 3), (4097,), (3, 1, 1, 1), (512, 256)]


def _params(dev):
    g = torch.Generator().manual_seed(5)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in SHAPES]


def _grads(step):
    g = torch.Generator().manual_seed(100 + step)
    return [torch.randn(*s, generator=g) * (0.1 + step) for s in SHAPES]


def maxrel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("cls,ref,wd", [(O.Adam, torch.optim.Adam, 0.0), (O.Adam, torch.optim.Adam, 1e-2), (O.AdamW, torch.optim.AdamW, 5e-2)])
def test_native_step_matches_torch(cls, ref, wd):
    pc, pd = _params("cpu"), _params("cuda")
    oc = ref(pc, lr=3e-3, betas=(0.9, 0.995), eps=1e-8, weight_decay=wd, foreach=False)
    od = cls(pd, lr=3e-3, betas=(0.9, 0.995), eps=1e-8, weight_decay=wd)
    for step in range(6):
        for p, q, g in zip(pc, pd, _grads(step)):
            p.grad, q.grad = g.clone(), g.clone().cuda()
        if step == 3:                                   # a learning-rate schedule changes the host-side value between steps
            oc.param_groups[0]["lr"] = od.param_groups[0]["lr"] = 1e-3
        oc.step()
        od.step()
    for p, q in zip(pc, pd):
        assert maxrel(q.detach().cpu(), p.detach()) < 2e-6
        assert maxrel(od.state[q]["exp_avg"].cpu(), oc.state[p]["exp_avg"]) < 2e-6
        assert maxrel(od.state[q]["exp_avg_sq"].cpu(), oc.state[p]["exp_avg_sq"]) < 2e-6
    assert float(od.state[pd[0]]["step"]) == 6.0


def test_missing_gradients_and_state_dict_round_trip():
    pd = _params("cuda")
    od = O.Adam(pd, lr=1e-3, weight_decay=1e-6)
    for step in range(2):
        for i, (q, g) in enumerate(zip(pd, _grads(step))):
            q.grad = None if i == 2 else g.cuda()       # a parameter without a gradient is skipped, as torch does
        od.step()
    assert 2 not in [i for i, q in enumerate(pd) if q in od.state and "exp_avg" in od.state[q]]
    sd = copy.deepcopy(od.state_dict())
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}       # the layout torch.optim.Adam checkpoints have
    p2 = [torch.nn.Parameter(q.detach().clone()) for q in pd]
    o2 = O.Adam(p2, lr=1e-3, weight_decay=1e-6)
    o2.load_state_dict(sd)
    for q, r, g in zip(pd, p2, _grads(7)):
        q.grad, r.grad = g.cuda(), g.cuda()
    od.step()
    o2.step()
    for q, r in zip(pd, p2):
        assert torch.equal(q, r)                         # resumed == uninterrupted, bit for bit
    # a checkpoint of torch's own Adam (host-side step counters) resumes too
    pc = [torch.nn.Parameter(q.detach().cpu().clone()) for q in pd]
    oc = torch.optim.Adam(pc, lr=1e-3, weight_decay=1e-6, foreach=False)
    for p, g in zip(pc, _grads(9)):
        p.grad = g.clone()
    oc.step()
    p3 = [torch.nn.Parameter(p.detach().clone().cuda()) for p in pc]
    o3 = O.Adam(p3, lr=1e-3, weight_decay=1e-6)
    o3.load_state_dict(copy.deepcopy(oc.state_dict()))   # as torch.load would hand it over (no tensors shared with `oc`)
    for p, r, g in zip(pc, p3, _grads(10)):
        p.grad, r.grad = g.clone(), g.clone().cuda()
    oc.step()
    o3.step()
    for p, r in zip(pc, p3):
        assert maxrel(r.detach().cpu(), p.detach()) < 2e-6
