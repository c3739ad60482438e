"""Full-domain sampling by overlapping tiles (BASELINE config 5: a 589x789 DANRA field as 256x256
tiles with halo).  The reference holds only the domain dimensions (config/full_run_config_new.yaml:26,28) — no tiling,
halo or stitching code exists there, so this module has no reference counterpart; its specification is DESIGN.md §9:

* tile origins: per axis n = ceil((L - overlap) / (tile - overlap)) tiles, first at 0, last flush with the domain edge,
  the rest evenly spread (x origins rounded down to multiples of 4: the in-kernel noise is drawn in quads);
* conditioning fields are cut into tiles on the device (`sbgm_extract_tiles`), the tiles are one sampler batch;
* the sampler's in-kernel noise is keyed by DOMAIN position (`tile_origins` of `sbgm_sampler_run`), so the pixels two tiles
  share see identical noise in both and differ only through their receptive-field context;
* `sbgm_stitch_tiles` blends the tiles with linear ramps over the overlap, normalised per pixel (a partition of unity).

Multi-GPU: tiles are independent units -> rank r samples tiles r::world with no collective during sampling; one
all-reduce of the finished tiles (every tile is written by exactly one rank, the others hold zeros) precedes the stitch.
The result does NOT depend on the world size or on `tiles_per_batch`: with `tile_origins` the Langevin step size of a
tile uses that tile's own score norm (csrc/sampler.hip; the reference's batch-mean rule, score_sampling.py:201, is for
batches of independent samples), the noise is keyed by domain position and everything else in the network is per
sample in eval mode.
"""
from __future__ import annotations

import math

import torch

from . import _native as N
from . import parallel


def axis_origins(length: int, tile: int, overlap: int, align: int = 1):
    if tile >= length:
        if tile > length:
            raise ValueError(f"tile {tile} larger than the domain extent {length}")
        return [0]
    if not 0 <= overlap < tile:
        raise ValueError(f"overlap {overlap} must be in [0, tile)")
    n = max(2, math.ceil((length - overlap) / (tile - overlap)))
    last = length - tile
    out = []
    for i in range(n):
        o = round(i * last / (n - 1))
        if 0 < i < n - 1 or align > 1:
            o = (o // align) * align
        out.append(o)
    if align > 1 and out[-1] != last:          # keep the domain covered: add a final tile flush with the edge when rounding fell short
        if last % align == 0:
            out[-1] = last
        else:
            raise ValueError(f"domain extent {length} - tile {tile} = {last} is not a multiple of {align}; pad the domain")
    return out


class FullDomainTiler:
    """Tile table + device gather/blend kernels for one (domain, tile, halo) geometry."""

    def __init__(self, domain_hw, tile: int = 256, halo: int = 32, device="cuda"):
        self.Hd, self.Wd = int(domain_hw[0]), int(domain_hw[1])
        self.tile, self.halo, self.overlap = int(tile), int(halo), 2 * int(halo)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.NativeError(f"FullDomainTiler runs on a ROCm device, got device={device!r}")
        ys = axis_origins(self.Hd, self.tile, self.overlap)
        xs = self._x_origins()
        self.origins = [(y, x) for y in ys for x in xs]
        self.origins_dev = torch.tensor(self.origins, dtype=torch.int32, device=self.device).contiguous()

    def _x_origins(self):
        # x origins must be multiples of 4 (noise quads).  The last tile has to end at the domain edge, so when
        # Wd - tile is not a multiple of 4 the tile table works on a domain padded on the right to the next multiple.
        self.Wd_pad = self.Wd + (-(self.Wd - self.tile)) % 4
        return axis_origins(self.Wd_pad, self.tile, self.overlap, align=4)

    def __len__(self):
        return len(self.origins)

    def _pad(self, dom):
        """[C, Hd, Wd] -> [C, Hd, Wd_pad], right edge replicated (only when Wd - tile is not a multiple of 4)"""
        if self.Wd_pad == self.Wd:
            return dom
        return torch.nn.functional.pad(dom, (0, self.Wd_pad - self.Wd), mode="replicate")

    def extract(self, domain: torch.Tensor, which=None) -> torch.Tensor:
        """domain [C, Hd, Wd] (device) -> tiles [T, C, tile, tile]; `which` = tile indices (default all)"""
        N.require_device(domain)
        if domain.dim() != 3 or domain.shape[1] != self.Hd or domain.shape[2] != self.Wd:
            raise ValueError(f"domain tensor {tuple(domain.shape)} does not match the tiler's [C, {self.Hd}, {self.Wd}]")
        dom = N.f32c(self._pad(domain.float()))
        C = dom.shape[0]
        org = self.origins_dev if which is None else self.origins_dev[list(which)].contiguous()
        T = org.shape[0]
        tiles = torch.empty(T, C, self.tile, self.tile, device=dom.device)
        N.check(N.lib().sbgm_extract_tiles(dom.data_ptr(), org.data_ptr(), tiles.data_ptr(), T, C, self.Hd, self.Wd_pad, self.tile,
                                           self.tile, N.stream()))
        return tiles

    def stitch(self, tiles: torch.Tensor) -> torch.Tensor:
        """tiles [T, C, tile, tile] (all tiles, table order) -> domain [C, Hd, Wd]"""
        N.require_device(tiles)
        tiles = N.f32c(tiles)
        T, C = tiles.shape[:2]
        if T != len(self.origins) or tiles.shape[2] != self.tile or tiles.shape[3] != self.tile:
            raise ValueError(f"expected {len(self.origins)} tiles of {self.tile}x{self.tile}, got {tuple(tiles.shape)}")
        dom = torch.empty(C, self.Hd, self.Wd_pad, device=tiles.device)
        N.check(N.lib().sbgm_stitch_tiles(tiles.data_ptr(), self.origins_dev.data_ptr(), dom.data_ptr(), T, C, self.Hd, self.Wd_pad,
                                          self.tile, self.tile, max(1, self.overlap), N.stream()))
        return dom[:, :, : self.Wd].contiguous() if self.Wd_pad != self.Wd else dom

    def sample(self, score_model, sampler, marginal_prob_std, diffusion_coeff, num_steps, cond_img=None, lsm_cond=None,
               topo_cond=None, y=None, seed=None, tiles_per_batch=None, **sampler_kw) -> torch.Tensor:
        """Sample the whole domain: cond_img [C,Hd,Wd] / lsm_cond, topo_cond [2,Hd,Wd] / y scalar class -> [1,Hd,Wd].
        Tiles are sharded over the ranks of the process group (if any) and, per rank, run in batches of
        `tiles_per_batch`; every batch shares `seed`, so the domain-keyed noise is identical wherever tiles overlap."""
        from .score_sampling import _fresh_seed
        _, world = parallel.world()
        seed = _fresh_seed() if seed is None else seed
        if world > 1:                                  # one seed for the whole domain
            import torch.distributed as dist
            s = torch.tensor([seed], dtype=torch.int64, device=self.device)
            dist.broadcast(s, 0)
            seed = int(s.item())
        def run_batch(idx):
            cut = lambda f: None if f is None else self.extract(f, idx)   # noqa: E731
            yb = None if y is None else torch.full((len(idx),), int(y), dtype=torch.int64, device=self.device)
            return sampler(score_model, marginal_prob_std, diffusion_coeff, batch_size=len(idx), num_steps=num_steps,
                           device=self.device, img_size=self.tile, y=yb, cond_img=cut(cond_img), lsm_cond=cut(lsm_cond),
                           topo_cond=cut(topo_cond), seed=seed, tile_origins=self.origins_dev[idx].contiguous(),
                           domain_width=self.Wd_pad, **sampler_kw)
        out = sample_tiles_sharded(len(self), run_batch, (1, self.tile, self.tile), self.device, tiles_per_batch)
        return self.stitch(out)


def sample_tiles_sharded(n_tiles: int, run_batch, tile_shape, device, tiles_per_batch=None) -> torch.Tensor:
    """Deal `n_tiles` independent tiles round-robin over the ranks of the process group (rank r owns tiles r::world), run this
    rank's tiles through `run_batch(list of tile indices) -> [len, *tile_shape]` in batches of `tiles_per_batch`, and merge:
    every tile is written by exactly one rank, the others hold zeros, so ONE sum all-reduce leaves all tiles on every rank.
    No collective runs while the tiles are being sampled."""
    rank, world = parallel.world()
    mine = list(range(n_tiles))[rank::world]
    per = tiles_per_batch or max(1, len(mine))
    out = torch.zeros(n_tiles, *tile_shape, device=device)
    for i in range(0, len(mine), per):
        idx = mine[i:i + per]
        out[idx] = run_batch(idx)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(out)
    return out
