"""Map sharding for the multi-GPU scan-to-map path (one process per GPU).

The only coupling between scan points in the reference's registration is the sum that
forms the 6x6 / 6x1 normal equations (reference PCR/src/LoamRegister.cpp:153-188).
So the sub-map is cut into spatial tiles (slabs along its longest axis, balanced by
point count); each rank indexes its tile plus a halo of the k-NN gate radius (1 m:
LoamRegister.cpp:59 gates the SQUARED 5th-neighbour distance at 1.0) and processes only
the scan points whose transformed position falls inside its un-haloed tile -- every scan
point is handled by exactly one rank with its complete neighbourhood, and one RCCL
all-reduce of 28 doubles per linearisation (pcr_comm_init) rebuilds the full system.
No point is ever exchanged.
"""
from dataclasses import dataclass

import numpy as np

BIG = 1.0e30


@dataclass
class Tile:
    points: np.ndarray   # float32 (n, stride): tile + halo
    lo: np.ndarray       # 3, inclusive lower bound of the query tile (map frame, metres)
    hi: np.ndarray       # 3, exclusive upper bound
    axis: int
    n_core: int          # points inside the tile proper


def split_bounds(points, world_size, align=1.0):
    """Cut positions along the longest axis: world_size-1 interior cuts at point-count
    quantiles, snapped to multiples of `align` (so cuts coincide with index cells)."""
    xyz = np.asarray(points)[:, :3]
    finite = np.isfinite(xyz).all(1)
    xyz = xyz[finite]
    ext = xyz.max(0) - xyz.min(0) if xyz.shape[0] else np.zeros(3)
    axis = int(np.argmax(ext))
    if world_size == 1 or xyz.shape[0] == 0:
        return axis, []
    qs = np.quantile(xyz[:, axis].astype(np.float64), np.arange(1, world_size) / world_size)
    cuts = np.round(qs / align) * align
    # keep cuts strictly increasing even for degenerate clouds
    for i in range(1, len(cuts)):
        if cuts[i] <= cuts[i - 1]:
            cuts[i] = cuts[i - 1] + align
    return axis, [float(c) for c in cuts]


def tile_for_rank(points, rank, world_size, halo=1.0, align=1.0):
    """The tile (with halo) of `rank`.  The outer tiles extend to +-BIG so that a scan point
    outside the map's bounding box still belongs to exactly one rank."""
    points = np.asarray(points)
    axis, cuts = split_bounds(points, world_size, align)
    edges = [-BIG] + cuts + [BIG]
    lo = np.full(3, -BIG)
    hi = np.full(3, BIG)
    lo[axis], hi[axis] = edges[rank], edges[rank + 1]
    c = points[:, axis].astype(np.float64)
    core = (c >= lo[axis]) & (c < hi[axis])
    keep = (c >= lo[axis] - halo) & (c < hi[axis] + halo)
    return Tile(np.ascontiguousarray(points[keep]), lo, hi, axis, int(core.sum()))


def unique_id():
    """128-byte RCCL unique id (rank 0 creates it, the caller shares it, e.g. with
    torch.distributed.broadcast_object_list)."""
    from .pcr import comm_unique_id
    return comm_unique_id()
