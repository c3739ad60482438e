"""One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference is single-device; this is the data-parallel layer around its two loops:

* sampling — whole independent batches per rank, NO data-path collective (the Langevin corrector's batch-mean
  gradient norm, reference score_sampling.py:201, stays inside one rank's batch, so every rank is bit-comparable
  with a single-GPU run of the same batch); results are gathered only on request.
* training — batch-sharded data parallel with exactly one exchange step: a sum all-reduce of one flattened fp32
  gradient bucket (76.3 MB for the default model) between `backward()` and `optimizer.step()`
  (reference training.py:405 -> :407), divided by the world size.  BatchNorm keeps per-replica statistics
  (PyTorch-DDP default).
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """Join the process group described by RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun contract).
    Returns (rank, world, local device index); world == 1 without those variables (no group is created).
    The backend is RCCL ("nccl") on GPUs; SBGM_DIST_BACKEND=gloo overrides it — that is how the multi-rank code paths are rehearsed
    on a one-GPU box, where all ranks share device 0 (local index = LOCAL_RANK modulo the visible device count)."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank, local = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    if torch.cuda.is_available():
        local %= max(1, torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or os.environ.get("SBGM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def world() -> tuple[int, int]:
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def barrier() -> None:
    if world()[1] > 1:
        dist.barrier()


def shard_range(n_items: int, rank: int, world_size: int) -> range:
    """Contiguous, balanced split of n_items independent units (batches / tiles): first n%world ranks get one more."""
    q, r = divmod(n_items, world_size)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


class GradientBucket:
    """One flat fp32 gradient tensor + one sum all-reduce, averaged over ranks.

    `GradientBucket(model)` adopts the model's gradient ARENA (train_graph.GradArena): the native backward kernels write every
    parameter gradient straight into its slice and `p.grad` is a view of it, so the exchange is the all-reduce alone — no
    gather / scatter copies (the native Adam step reads the same memory through `p.grad`).  A gradient that did not land in the
    arena (live `.grad` tensors from gradient accumulation, the ConvTranspose2d ablation weights, a CPU model) is copied in and
    `p.grad` re-pointed at its slice, which keeps the result identical.
    `GradientBucket(iterable of parameters)` (no arena) keeps the same interface with a private flat tensor."""

    def __init__(self, model_or_params):
        self.model = model_or_params if isinstance(model_or_params, torch.nn.Module) else None
        params = self.model.parameters() if self.model is not None else model_or_params
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradientBucket needs at least one trainable parameter")
        self._own = None
        self.copies = 0                                 # gradients that had to be copied into the flat tensor (diagnostic)

    def _layout(self):
        """(flat tensor, [view per parameter])"""
        if self.model is not None and self.params[0].is_cuda:
            from .train_graph import arena_for
            arena = arena_for(self.model)
            if getattr(self, "_views_for", None) is not arena:
                self._views = [arena.grad_of(p) for p in self.params]
                self._views_for = arena
            return arena.flat, self._views
        if self._own is None:
            dev = self.params[0].device
            flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
            views, off = [], 0
            for p in self.params:
                views.append(flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self._own = (flat, views)
        return self._own

    def all_reduce_(self) -> None:
        rank, ws = world()
        if ws == 1:
            return
        flat, views = self._layout()
        for p, v in zip(self.params, views):
            g = p.grad
            if g is None:
                v.zero_()                               # unused parameter: contributes 0, stays without a gradient
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g)
                p.grad = v
                self.copies += 1
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(ws)


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every replica start from rank `src`'s weights and buffers."""
    if world()[1] == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def sample_sharded(sampler, n_batches: int, make_kwargs, gather: bool = False):
    """Run `sampler(**make_kwargs(i))` for this rank's share of `n_batches` independent batches.
    Returns {batch index: sample tensor}; with gather=True rank 0 receives every rank's results (CPU tensors)."""
    rank, ws = world()
    mine = {i: sampler(**make_kwargs(i)) for i in shard_range(n_batches, rank, ws)}
    if not gather or ws == 1:
        return mine
    payload = {i: v.detach().cpu() for i, v in mine.items()}
    out = [None] * ws if rank == 0 else None
    dist.gather_object(payload, out, dst=0)
    if rank != 0:
        return mine
    merged = {}
    for d in out:
        merged.update(d)
    return merged
