"""`torch.optim.Adam` / `torch.optim.AdamW` whose `step()` is ONE native launch over every parameter of the model
(csrc/optim.hip) when the parameters live on the GPU; the update rule, hyper-parameters, `state_dict()` layout
(`step`, `exp_avg`, `exp_avg_sq` per parameter) and every other method are torch's, so checkpoints written by the reference
(training.py:407, :545 saves `optimizer.state_dict()`) load and resume unchanged.  Options the launch does not implement
(amsgrad, maximize, sparse gradients, tensor learning rates, CPU parameters) take torch's own step."""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N


class _NativeStep:
    """mix-in: the per-group descriptor table and the launch.

    `grad_scale` (default 1.0) multiplies every gradient as the step reads it.  The data-parallel pipeline sets it to 1 / world
    and leaves the all-reduced SUM in `.grad`: the division of gradient averaging then costs nothing (a `flat.div_(world)` is a
    76 MB read + write pass per step).  Groups that fall back to torch's own step get their gradients scaled in place first."""
    _decoupled = 0
    grad_scale = 1.0

    def _native_ok(self, group, params):
        return (params and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse
                               and p.grad.dtype == torch.float32 for p in params)
                and not group.get("amsgrad") and not group.get("maximize") and not group.get("differentiable")
                and not isinstance(group["lr"], torch.Tensor) and len({p.device for p in params}) == 1)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        leftovers = False
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if not self._native_ok(group, params):
                leftovers = True
                continue
            self._step_group(gi, group, params)
        if leftovers:                                  # torch's own step for the groups the launch does not cover
            saved = [(g, g["params"]) for g in self.param_groups]
            try:
                for g, ps in saved:
                    keep = [p for p in ps if p.grad is not None]
                    g["params"] = [] if (keep and self._native_ok(g, keep)) else ps
                    if g["params"] and self.grad_scale != 1.0:
                        for p in keep:
                            p.grad.mul_(self.grad_scale)
                super().step()
            finally:
                for g, ps in saved:
                    g["params"] = ps
        return loss

    def state_dict(self):
        """torch's layout; the shared device step counter is written out as one CLONE per parameter, so a stock
        torch.optim.Adam that loads the checkpoint owns independent counters (it would otherwise advance the one shared
        storage once per parameter and per step)."""
        sd = super().state_dict()
        sd["state"] = {k: ({**v, "step": v["step"].clone()} if torch.is_tensor(v.get("step")) else v) for k, v in sd["state"].items()}
        return sd

    def _step_group(self, gi, group, params):
        dev = params[0].device
        first = self.state[params[0]]
        master = first.get("step")
        if master is None or not master.is_cuda:       # fresh state, or a state_dict loaded with host-side step counters
            master = torch.full((), float(master) if master is not None else 0.0, dtype=torch.float32, device=dev)
        for p in params:
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] = master                        # one device counter per group; every parameter's entry refers to it
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in params]
        key = tuple((p.data_ptr(), g.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel())
                    for p, g in zip(params, grads))
        cache = self.__dict__.setdefault("_native_tables", {})
        tab = cache.get(gi)
        if tab is None or tab[0] != key:               # gradients are new tensors after every eager backward; static under a graph
            lib, blk, descs = N.lib(), 0, []
            for ptr_p, ptr_g, ptr_m, ptr_v, n in key:
                descs.append(N.AdamDesc(ptr_p, ptr_g, ptr_m, ptr_v, n, blk, 0))
                blk += lib.sbgm_adam_step_blocks(n)
            raw = (N.AdamDesc * len(descs))(*descs)
            host = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8)
            tab = cache[gi] = (key, host.to(dev), len(descs), blk)
        master += 1
        b1, b2 = group["betas"]
        N.check(N.lib().sbgm_adam_step_batched(tab[1].data_ptr(), tab[2], tab[3], master.data_ptr(), float(group["lr"]), float(b1), float(b2),
                                               float(group["eps"]), float(group["weight_decay"]), self._decoupled, float(self.grad_scale), N.stream()))
        N.bump_generation()                            # parameters were written through raw pointers (no version-counter bump)


class Adam(_NativeStep, torch.optim.Adam):
    _decoupled = 0


class AdamW(_NativeStep, torch.optim.AdamW):
    _decoupled = 1
