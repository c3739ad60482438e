"""`TrainingPipeline_general` — the reference's training hot loop and its two satellites (reference sbgm/training.py:
__init__ :41-117, xavier_init_weights :188-201, load_checkpoint :203-222, save_model :224-244, train_batches :246-422,
train :424-508, validate_batches :510-609, generate_and_plot_samples :611-786).

Same constructor signature, same per-batch order (zero_grad -> loss_fn -> backward -> step -> .item()), same checkpoint
dict (`network_params`, `optimizer_params`), best-validation checkpointing, per-epoch pickled losses.  Kept out:
matplotlib plotting (`generate_and_plot_samples` only generates and returns the samples) and the precipitation
back-transform of the sentinel.  New relative to the reference: when a process group exists
the gradients are all-reduced (one flattened bucket, RCCL) between backward and step.
"""
from __future__ import annotations

import logging
import os
import pickle

import torch
import torch.nn as nn

from . import parallel
from .score_sampling import Euler_Maruyama_sampler, ode_sampler, pc_sampler
from .utils import (draw_condition_dropout, extract_samples, extract_samples_device, get_model_string,
                    report_precip_extremes)

logger = logging.getLogger(__name__)
_SAMPLERS = {"pc_sampler": pc_sampler, "Euler_Maruyama_sampler": Euler_Maruyama_sampler, "ode_sampler": ode_sampler}


class TrainingPipeline_general:
    def __init__(self, model, loss_fn, marginal_prob_std_fn, diffusion_coeff_fn, optimizer, device, lr_scheduler, cfg):
        self.model, self.loss_fn, self.optimizer, self.lr_scheduler, self.cfg = model, loss_fn, optimizer, lr_scheduler, cfg
        self.marginal_prob_std_fn, self.diffusion_coeff_fn = marginal_prob_std_fn, diffusion_coeff_fn
        self.model.debug_pre_sigma_div = cfg["training"].get("debug_pre_sigma_div", True)
        self.device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        t = cfg["training"]
        self.weight_init, self.custom_weight_initializer = t["weight_init"], t.get("custom_weight_initializer")
        self.sdf_weighted_loss, self.with_ema = t.get("sdf_weighted_loss", False), t.get("with_ema", False)
        if self.weight_init:
            self.model.apply(self.custom_weight_initializer or self.xavier_init_weights)
        mon = (cfg.get("monitoring", {}) or {}).get("extreme_prcp", {}) or {}      # training.py:151-160 (same defaults)
        self.extreme_enabled = bool(mon.get("enabled", True))
        self.extreme_every_step = int(mon.get("every_steps", 50))
        self.extreme_threshold_mm = float(mon.get("threshold_mm", 500.0))
        self.extreme_back_transform = bool(mon.get("back_transform", True))
        self.extreme_log_first_n = int(mon.get("log_first_n", 5))
        self.extreme_clamp_in_gen = bool(mon.get("clamp_in_generation", True))
        self.back_transforms = self._build_back_transforms(cfg)        # device-side transforms; None when no stats are reachable
        self.last_generation_check = None
        self.model_string = get_model_string(cfg)
        self.checkpoint_dir = cfg["paths"]["checkpoint_dir"]
        self.checkpoint_name = self.model_string + ".pth.tar"
        self.checkpoint_path = os.path.join(self.checkpoint_dir, self.checkpoint_name)
        self.path_losses = os.path.join(cfg["paths"]["path_save"], "samples", self.model_string, "losses")
        for d in (self.checkpoint_dir, self.path_losses):
            os.makedirs(d, exist_ok=True)
        self._bucket = None

    @staticmethod
    def _build_back_transforms(cfg):
        """training.py:162-185: inverse transforms from the saved global statistics, or None (the sentinel then runs on
        model-space values, as in the reference when the stats cannot be loaded)."""
        from .special_transforms import build_back_transforms_from_stats
        try:
            hr, lr = cfg["highres"], cfg["lowres"]
            dims = lambda d: f"{d[0]}x{d[1]}" if d is not None else "full_domain"                 # noqa: E731
            crop = lambda c: "_".join(map(str, c)) if c is not None else "no_crop"               # noqa: E731
            return build_back_transforms_from_stats(
                hr_var=hr["variable"], hr_model=hr["model"], domain_str_hr=dims(hr.get("full_domain_dims")),
                crop_region_str_hr=crop(hr.get("cutout_domains")), hr_scaling_method=hr["scaling_method"],
                hr_buffer_frac=hr.get("buffer_frac", 0.0), lr_vars=lr["condition_variables"], lr_model=lr["model"],
                lr_scaling_methods=lr["scaling_methods"], domain_str_lr=dims(lr.get("full_domain_dims")),
                crop_region_str_lr=crop(lr.get("cutout_domains")), lr_buffer_frac=lr.get("buffer_frac", 0.0), split="all",
                stats_dir_root=cfg["paths"]["stats_load_dir"])
        except Exception as e:                       # noqa: BLE001  (the reference swallows this too, :186-187)
            logger.warning(f"[monitor] Could not build back transforms for sentinel; will skip back_transform in training. Error: {e}")
            return None

    @staticmethod
    def xavier_init_weights(m):
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None and torch.is_tensor(m.bias):
                m.bias.data.fill_(0.01)

    def load_checkpoint(self, checkpoint_path, load_ema=False, device=None):
        state = torch.load(checkpoint_path, map_location=device or self.device, weights_only=True)["network_params"]
        self.model.load_state_dict(state)

    def save_model(self, dirname="./model_params", filename="SBGM.pth"):
        os.makedirs(dirname, exist_ok=True)
        return torch.save({"network_params": self.model.state_dict(), "optimizer_params": self.optimizer.state_dict()},
                          os.path.join(dirname, filename))

    def _extract(self, samples, split):
        """Batch dict -> tensors on the device.  Raw batches (1-channel geo fields) and batches that need the
        classifier-free-guidance condition dropout are assembled by the device-side packer (one launch group per batch,
        reference data_modules.py:957-993); batches that already carry value||mask geo fields take the plain path."""
        guidance = (self.cfg.get("classifier_free_guidance", {}) or {})
        raw_geo = any(samples.get(k) is not None and samples[k].shape[1] == 1 for k in ("lsm", "topo"))
        dropped = draw_condition_dropout(next(iter(samples.values())).shape[0], split, guidance)
        if raw_geo or dropped is not None:
            return extract_samples_device(samples, self.device, dropped)
        return extract_samples(samples, self.device)

    def _loss(self, samples, split="train"):
        x, seasons, cond, _lsm_hr, lsm, sdf, topo, _hp, _lp = self._extract(samples, split)
        return x, self.loss_fn(self.model, x, self.marginal_prob_std_fn, y=seasons, cond_img=cond, lsm_cond=lsm,
                               topo_cond=topo, sdf_cond=sdf if self.sdf_weighted_loss else None)

    def _graph_step(self, samples, soft=False):
        """`training.use_hip_graph: true` — loss_fn + backward of one step replayed as a hipGraph (torch.cuda.graphs over the
        C-ABI launches): the ~700 launches of a step cost one host call.  The batch is copied into static input tensors; the
        first batch of a new shape runs 2 eager warm-up steps' worth of launches (tile tuning, weight-pack plan) and captures.
        Gradients land in the parameters' (static) .grad tensors exactly as after loss.backward()."""
        x, seasons, cond, _lsm_hr, lsm, sdf, topo, _hp, _lp = self._extract(samples, "train")
        live = [x, seasons, cond, lsm, topo, sdf if self.sdf_weighted_loss else None]
        key = tuple(None if t is None else (tuple(t.shape), t.dtype) for t in live)
        g = getattr(self, "_graphs", None)
        if g is None:
            g = self._graphs = {}
        ent = g.get(key)
        if ent is None:
            static = [None if t is None else t.clone() for t in live]

            def fwd_bwd():
                loss = self.loss_fn(self.model, static[0], self.marginal_prob_std_fn, y=static[1], cond_img=static[2],
                                    lsm_cond=static[3], topo_cond=static[4], sdf_cond=static[5])
                loss.backward()
                return loss
            saved = {k: v.detach().clone() for k, v in self.model.state_dict().items()}      # warm-up must not train
            entry_stream = torch.cuda.current_stream()
            graph = loss = None
            for attempt in range(2 if soft else 1):          # auto mode: one retry (a capture can be invalidated by something transient)
                try:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(3 if attempt == 0 else 1):   # tile tuning, then the weight-pack plan settles (one more than it needs)
                            self.optimizer.zero_grad(set_to_none=True)
                            fwd_bwd()
                    torch.cuda.current_stream().wait_stream(side)
                    self.model.load_state_dict(saved)                                          # BatchNorm running statistics back
                    graph = torch.cuda.CUDAGraph()
                    self.optimizer.zero_grad(set_to_none=True)
                    with torch.cuda.graph(graph):
                        loss = fwd_bwd()
                    break
                except Exception as e:                                                         # noqa: BLE001
                    # torch.cuda.graph.__exit__ ends the capture FIRST and restores the thread's stream after it: when ending an
                    # invalidated capture raises, the thread is left on the dead capture stream and every later launch fails.  Put it back.
                    torch.cuda.set_stream(entry_stream)
                    if not soft:
                        raise
                    # `auto` mode: something in this step cannot be captured (a host synchronisation inside a user-supplied loss /
                    # model hook, ...).  Put the model back as it was; after the second failure the caller runs eager steps from here on.
                    torch.cuda.synchronize()
                    self.model.load_state_dict(saved)
                    self.optimizer.zero_grad(set_to_none=True)
                    graph = loss = None
                    if attempt == 1:
                        self._graph_failed = True
                        logger.warning(f"training.use_hip_graph=auto: the step could not be captured ({type(e).__name__}: {e}); "
                                       f"continuing with eager steps")
                        return None
            # each capture leaves ITS gradient tensors in .grad (graph-pool memory, or the model's arena): keep them with the
            # graph, so that replaying an older graph after a newer capture hands the optimizer the tensors that replay wrote
            params = [p for p in self.model.parameters() if p.grad is not None]
            ent = g[key] = (graph, static, loss, params, [p.grad for p in params])
            self._graph_live = key
            # the capture itself did not execute: the replay below is this batch's step
        graph, static, loss, params, grads = ent
        for s_, t in zip(static, live):
            if s_ is not None:
                s_.copy_(t)
        graph.replay()
        if getattr(self, "_graph_live", None) != key or (params and params[0].grad is not grads[0]):
            for p in self.model.parameters():
                p.grad = None
            for p, gr in zip(params, grads):
                p.grad = gr
            self._graph_live = key
        from . import _native
        _native.bump_generation()       # the replay moved BatchNorm running statistics / num_batches_tracked without a version bump
        return x, loss

    class _LossMeter:
        """Sum of the per-batch losses without a host synchronisation per step.  The reference reads `loss.item()` after every
        optimizer step (training.py:410), which parks the device while the host collates the next batch; here each loss is copied
        into a slot of a device buffer and the buffer is read back every `cap` steps and at the end — the same fp32 losses summed
        in the same order in a Python float, so the reported epoch mean is unchanged while the loader overlaps the running step."""

        def __init__(self, device, cap=64):
            self.device, self.cap, self.buf = torch.device(device), cap, None
            self.n, self.total = 0, 0.0

        def add(self, loss):
            if self.device.type != "cuda":
                self.total += loss.item()
                return
            if self.buf is None:
                self.buf = torch.empty(self.cap, device=loss.device)
            self.buf[self.n].copy_(loss.detach())
            self.n += 1
            if self.n == self.buf.numel():
                self.flush()

        def flush(self):
            if self.buf is not None and self.n:
                for v in self.buf[: self.n].tolist():
                    self.total += v
                self.n = 0
            return self.total

    def train_batches(self, dataloader, epochs=10, current_epoch=1, verbose=True, use_mixed_precision=False):
        if use_mixed_precision:
            raise NotImplementedError("fp32 only: the reference's autocast branch is commented out (training.py:325-343)")
        self.model.train()
        world = parallel.world()[1]
        if self._bucket is None and world > 1:
            self._bucket = parallel.GradientBucket(self.model)
        # 1 / world rides on the optimizer launch when the optimizer can take it (optim.Adam / AdamW), else the bucket divides
        fold = self._bucket is not None and hasattr(self.optimizer, "grad_scale")
        if fold:
            self.optimizer.grad_scale = 1.0 / world
        meter = self._LossMeter(self.device)
        from . import train_graph
        # training.use_hip_graph: true / false / auto (default, also when the key is absent as in the reference's YAML files): auto
        # captures the step on a ROCm device unless something known not to be capturable is switched on (SyncBatchNorm collectives,
        # the model's host-side debug statistic), and falls back to eager steps if the capture itself fails
        want = self.cfg["training"].get("use_hip_graph", "auto")
        auto = isinstance(want, str) and want.strip().lower() == "auto"
        on_gpu = torch.device(self.device).type == "cuda"
        if auto:
            use_graph = (on_gpu and not getattr(self, "_graph_failed", False) and train_graph._sync_world() is None
                         and not getattr(self.model, "debug_pre_sigma_div", False)
                         and not any(getattr(m, "dropout", 0.0) > 0 for m in self.model.modules() if type(m).__name__ == "ImageSelfAttention"))
        else:
            use_graph = (want if isinstance(want, bool) else str(want).strip().lower() in ("1", "true", "yes", "on")) and on_gpu
        if use_graph and train_graph._sync_world() is not None:
            raise ValueError("training.sync_batchnorm needs host-driven collectives between kernel halves and cannot run inside a captured "
                             "step: set training.use_hip_graph: false (or sync_batchnorm: false)")
        prev_overlap = train_graph.set_overlap_bucket(self._bucket if not use_graph else None)
        for idx, samples in enumerate(dataloader):
            step = self._graph_step(samples, soft=auto) if use_graph else None
            if step is not None:
                x, batch_loss = step
            else:
                use_graph = False                           # (auto mode: a failed capture switches the rest of the run to eager steps)
                self.optimizer.zero_grad()
                x, batch_loss = self._loss(samples)
                batch_loss.backward()
            if self.extreme_enabled and idx % self.extreme_every_step == 0:
                self._check_ground_truth(x)
            if self._bucket is not None:
                self._bucket.all_reduce_(average=not fold)   # the exchange step of the path (its decoder part started inside backward)
            self.optimizer.step()
            meter.add(batch_loss)
        train_graph.set_overlap_bucket(prev_overlap)
        avg = meter.flush() / max(1, len(dataloader))
        if verbose:
            logger.info(f"→ Epoch {current_epoch}/{epochs} completed: Avg. training Loss: {avg:.4f}")
        return avg

    def validate_batches(self, dataloader, verbose=True):
        self.model.eval()
        meter = self._LossMeter(self.device)
        with torch.inference_mode():
            for samples in dataloader:
                meter.add(self._loss(samples, "valid")[1])
        avg = meter.flush() / max(1, len(dataloader))
        if verbose:
            logger.info(f"→ Validation Loss: {avg:.4f}")
        return avg

    def train(self, train_dataloader, val_dataloader, gen_dataloader, cfg, epochs=1, verbose=True, use_mixed_precision=False):
        train_losses, val_losses = [], []
        train_loss = val_loss = best = float("inf")
        for epoch in range(1, epochs + 1):
            self.epoch = epoch
            train_loss = self.train_batches(train_dataloader, epochs, epoch, verbose, use_mixed_precision)
            val_loss = self.validate_batches(val_dataloader, verbose)
            train_losses.append(train_loss)
            val_losses.append(val_loss)
            if val_loss < best:
                best = val_loss
                if parallel.world()[0] == 0:
                    self.save_model(self.checkpoint_dir, self.checkpoint_name)
                    logger.info(f"→ Best model saved with validation loss: {best:.4f} at epoch {epoch}.")
            parallel.barrier()            # the other ranks read rank 0's checkpoint in generate_and_plot_samples
            with open(os.path.join(self.path_losses, f"losses_{self.model_string}.pkl"), "wb") as f:
                pickle.dump({"train_losses": train_losses, "val_losses": val_losses}, f)
            if cfg["visualization"].get("create_figs") and cfg["data_handling"].get("n_gen_samples", 0) > 0 and gen_dataloader is not None:
                self.generate_and_plot_samples(gen_dataloader, cfg=cfg, epoch=epoch)
        return train_loss, val_loss

    def generate_and_plot_samples(self, gen_dataloader, cfg, epoch, **_):
        """Per-epoch preview: first generation batch through cfg.sampler.sampler_type with the reference's kwargs
        (training.py:683-695).  Returns the generated tensor [B,1,H,W] (plotting is out of scope)."""
        if os.path.exists(self.checkpoint_path):
            self.load_checkpoint(self.checkpoint_path)
        self.model.eval()
        sampler = _SAMPLERS.get(cfg["sampler"]["sampler_type"])
        if sampler is None:
            raise ValueError(f"Sampler type {cfg['sampler']['sampler_type']} not recognized.")
        samples = next(iter(gen_dataloader))
        x, seasons, cond, _lsm_hr, lsm, _sdf, topo, _hp, _lp = extract_samples(samples, self.device)
        kw = dict(score_model=self.model, marginal_prob_std=self.marginal_prob_std_fn, diffusion_coeff=self.diffusion_coeff_fn,
                  batch_size=x.shape[0], num_steps=cfg["sampler"]["n_timesteps"], device=self.device,
                  img_size=cfg["highres"]["data_size"][0])
        if sampler is not ode_sampler:
            kw.update(y=seasons, cond_img=cond, lsm_cond=lsm, topo_cond=topo)
        gen = sampler(**kw)
        gen, self.last_generation_check = self.monitor_generated(gen, cfg)
        return gen

    def _check_ground_truth(self, x):
        """training.py:364-395: back-transform the HR batch (when configured) and run the sentinel, on the device"""
        x_bt = x.detach()
        bt = (self.back_transforms or {}).get("hr") if self.extreme_back_transform else None
        if callable(bt):
            x_bt = bt(x_bt)
        return report_precip_extremes(x_bt, "ground_truth_hr", self.extreme_threshold_mm, logger=logger.warning)

    def monitor_generated(self, gen, cfg):
        """Post-sampler block of the reference's preview (training.py:697-748) without the host round trip:
        back-transform -> sentinel -> optional clamp to [0, clamp_max_mm], all on the device; when the clamp fires it is
        fused with the back-transform into one pass over the raw samples.  Returns (tensor to plot/save, sentinel dict)."""
        mon = (cfg.get("monitoring", {}) or {}).get("extreme_prcp", {}) or {}
        if not mon.get("enabled", self.extreme_enabled):
            return gen, {"has_extreme": False}
        from .special_transforms import _Chain, apply_chain, clamp_program, fuse
        bt = (self.back_transforms or {}).get("hr") if mon.get("back_transform", True) else None
        gen_bt = bt(gen) if callable(bt) else gen
        thr = float(mon.get("threshold_mm", self.extreme_threshold_mm))
        chk = report_precip_extremes(gen_bt.detach(), "generated_hr", cap_mm_day=thr, logger=logger.warning)
        if chk.get("has_extreme", False):
            vals = chk.get("extreme_values", [])
            logger.warning("[monitor][gen] Extreme precipitation detected in generated samples:")
            logger.warning(f"               max={max(vals):.1f} mm/day, count={len(vals)}, threshold={thr} mm/day")
            if mon.get("clamp_in_generation", self.extreme_clamp_in_gen):
                clamp_max = float(mon.get("clamp_max_mm", thr))
                clamp = clamp_program(0.0, clamp_max)
                gen = fuse(bt, clamp)(gen) if isinstance(bt, _Chain) else apply_chain(gen_bt, clamp)
                logger.warning(f"[monitor][gen] Clamped generated samples to max {clamp_max} mm/day.")
        return gen, chk
