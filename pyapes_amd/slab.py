"""Slab decomposition along axis 0 (new).  Host logic only."""
from __future__ import annotations


def slab_extent(n0: int, rank: int, world: int) -> tuple[int, int]:
    """-> (first global plane, number of planes) of ``rank``; remainders go to the low ranks."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"slab_extent: bad rank/world {rank}/{world}")
    if n0 < 3 * world:
        raise ValueError(f"slab_extent: {n0} planes are too few for {world} slabs (>= 3 planes each)")
    base, rem = divmod(n0, world)
    off = rank * base + min(rank, rem)
    return off, base + (1 if rank < rem else 0)
