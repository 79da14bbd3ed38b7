"""Tile-parallel inference across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

Super-resolution tiles are independent (first row): the batch of tiles is cut into contiguous,
balanced slices, every rank runs the HIP forward on its slice, and the HR tiles are collected with ONE all_gather of
equally padded slices.  There is no collective on the data path itself.  (Sharding a single huge image with
per-layer halo exchange -- BASELINE config 4 -- lives in studiosr_amd/strips.py.)
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def tile_partition(n_tiles: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous balanced [start, stop) slice of every rank; the first n % world ranks get one extra tile."""
    base, extra = divmod(n_tiles, world_size)
    out, start = [], 0
    for r in range(world_size):
        stop = start + base + (1 if r < extra else 0)
        out.append((start, stop))
        start = stop
    return out


class TileParallel:
    """Run `fn` (e.g. model.forward) on this rank's slice of a tile batch and gather all HR tiles on every rank."""

    def __init__(self, fn: Callable[[Tensor], Tensor], group: Optional[dist.ProcessGroup] = None) -> None:
        self.fn = fn
        self.group = group

    def __call__(self, tiles: Tensor) -> Tensor:
        if not dist.is_available() or not dist.is_initialized():
            return self.fn(tiles)
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        parts = tile_partition(tiles.shape[0], world)
        lo, hi = parts[rank]
        n_max = max(b - a for a, b in parts)
        if hi > lo:
            mine = self.fn(tiles[lo:hi].contiguous())
        else:  # more ranks than tiles: run one tile to learn the output shape, contribute nothing
            mine = self.fn(tiles[:1].contiguous())[:0]
        out_dev = mine.device
        if mine.is_cuda and dist.get_backend(self.group) != "nccl":  # only RCCL is ordered on the device stream: stage through the host
            torch.cuda.current_stream(out_dev).synchronize()
            mine = mine.cpu()
        pad = torch.zeros((n_max,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        pad[: hi - lo] = mine
        gathered = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(gathered, pad, group=self.group)
        return torch.cat([g[: b - a] for g, (a, b) in zip(gathered, parts)], dim=0).to(out_dev)
