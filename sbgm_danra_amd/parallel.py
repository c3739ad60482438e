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
    """One flat fp32 gradient tensor, summed over the ranks in up to two all-reduces, the first overlapped with backward.

    `GradientBucket(model)` adopts the model's gradient ARENA (train_graph.GradArena): the native backward kernels write every
    parameter gradient straight into its slice and `p.grad` is a view of it, so the exchange is the all-reduce alone — no
    gather / scatter copies (the native Adam step reads the same memory through `p.grad`).  A gradient that did not land in the
    arena (live `.grad` tensors from gradient accumulation, the ConvTranspose2d ablation weights, a CPU model) is copied in and
    `p.grad` re-pointed at its slice, which keeps the result identical.
    `GradientBucket(iterable of parameters)` (no arena) keeps the same interface with a private flat tensor.

    Overlap.  The arena is in parameter order: encoder | decoder.  Backward runs the decoder first, so when the gradient reaches the
    encoder/decoder boundary (train_graph._BucketBoundary on the bottleneck feature map) every decoder gradient is final:
    `begin_early(params)` then starts an ASYNCHRONOUS all-reduce of the decoder slice (5.4 M of the 19 M parameters) that runs
    beside the encoder's backward; `all_reduce_()` after backward waits for it and reduces the rest.  Summation is per element
    in both forms, so the result is bit-identical to the single all-reduce (world size 2: a + b either way).

    average=True divides by the world size in place (a read + write pass over the bucket); the training pipeline passes False
    and sets `optimizer.grad_scale = 1 / world` instead (optim.Adam folds the factor into its launch)."""

    def __init__(self, model_or_params):
        self.model = model_or_params if isinstance(model_or_params, torch.nn.Module) else None
        params = self.model.parameters() if self.model is not None else model_or_params
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradientBucket needs at least one trainable parameter")
        self._own = None
        self.copies = 0                                 # gradients that had to be copied into the flat tensor (diagnostic)
        self.early = 0                                  # asynchronous early all-reduces launched (diagnostic)
        self._pending = None                            # (work handle, first element, end element) of the early slice
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._absent = {}                               # parameter index -> True when no rank ever produces its gradient

    def _layout(self):
        """(flat tensor, [view per parameter], [offset per parameter])"""
        if self.model is not None and self.params[0].is_cuda:
            from .train_graph import arena_for
            arena = arena_for(self.model)
            if getattr(self, "_views_for", None) is not arena:
                self._views = [arena.grad_of(p) for p in self.params]
                self._offs = [arena.index[p.data_ptr()][0] for p in self.params]
                self._views_for = arena
            return arena.flat, self._views, self._offs
        if self._own is None:
            dev = self.params[0].device
            flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
            views, offs, off = [], [], 0
            for p in self.params:
                views.append(flat[off:off + p.numel()].view(p.shape))
                offs.append(off)
                off += p.numel()
            self._own = (flat, views, offs)
        return self._own

    def _adopt(self, idxs, views, zero_missing=True):
        for i in idxs:
            p, v = self.params[i], views[i]
            g = p.grad
            if g is None:
                if zero_missing:
                    v.zero_()                           # unused on this rank: contributes 0 (the reduced value is attached below)
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g)
                p.grad = v
                self.copies += 1

    def begin_early(self, params: Iterable[torch.nn.Parameter]) -> bool:
        """Start the asynchronous all-reduce of the CONTIGUOUS tail slice holding `params` (their gradients must be final).
        Returns False (and does nothing) with one rank, an early slice already in flight, or a slice that is not the tail."""
        rank, ws = world()
        if ws == 1 or self._pending is not None:
            return False
        idxs = sorted(self._index[id(p)] for p in params if id(p) in self._index)
        if not idxs or idxs[-1] != len(self.params) - 1 or idxs != list(range(idxs[0], idxs[-1] + 1)):
            return False
        flat, views, offs = self._layout()
        # mid-backward: a slice without a .grad yet is left alone (the arena was zeroed at the start of the step and the kernels write
        # into it directly; zeroing here could wipe a gradient whose AccumulateGrad has not run)
        self._adopt(idxs, views, zero_missing=self._own is not None)
        a = offs[idxs[0]]
        work = dist.all_reduce(flat[a:], op=dist.ReduceOp.SUM, async_op=True)
        self._pending = (work, a, idxs[0])
        self.early += 1
        return True

    def all_reduce_(self, average: bool = True) -> None:
        rank, ws = world()
        if ws == 1:
            return
        flat, views, offs = self._layout()
        end, n_rest = flat.numel(), len(self.params)
        if self._pending is not None:
            work, end, n_rest = self._pending
            self._pending = None
        self._adopt(range(n_rest), views)
        missing = [i for i, p in enumerate(self.params) if p.grad is None]
        if end > 0:
            dist.all_reduce(flat[:end], op=dist.ReduceOp.SUM)
        if n_rest < len(self.params):
            work.wait()
        # No local gradient.  If another rank produced one, this rank must step the parameter with the same averaged gradient or
        # the replicas diverge; if no rank did (the final block's unused time projection), it must stay without a gradient, as in
        # the single-device reference (a zero gradient would still apply weight decay).  Which of the two holds is decided once
        # per parameter, on the first step it is seen missing (one device->host check), and remembered.
        for i in missing:
            known = self._absent.get(i)
            if known is None:
                known = self._absent[i] = not bool(views[i].any())
            if not known:
                self.params[i].grad = views[i]
        if average:
            flat.div_(ws)


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every replica start from rank `src`'s weights and buffers."""
    if world()[1] == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
    if any(t.is_cuda for t in module.parameters()):
        from . import _native                           # written through .data: no version bump an engine copy could see
        _native.bump_generation()


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
