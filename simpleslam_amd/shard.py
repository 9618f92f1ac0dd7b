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
    halo: float = 1.0    # the cloud holds every map point within this distance of the tile along `axis`


def split_bounds(points, world_size, align=1.0, shift=0.0):
    """Cut positions along the longest axis: world_size-1 interior cuts at point-count
    quantiles, snapped to the lattice (k + shift) * align (so cuts coincide with index cells / voxel faces)."""
    xyz = np.asarray(points)[:, :3]
    finite = np.isfinite(xyz).all(1)
    xyz = xyz[finite]
    ext = xyz.max(0) - xyz.min(0) if xyz.shape[0] else np.zeros(3)
    axis = int(np.argmax(ext))
    if world_size == 1 or xyz.shape[0] == 0:
        return axis, []
    qs = np.quantile(xyz[:, axis].astype(np.float64), np.arange(1, world_size) / world_size)
    k = np.round(qs / align - shift)
    # keep cuts strictly increasing even for degenerate clouds
    for i in range(1, len(k)):
        if k[i] <= k[i - 1]:
            k[i] = k[i - 1] + 1
    return axis, [float((v + shift) * align) for v in k]


def tile_for_rank(points, rank, world_size, halo=1.0, align=1.0, shift=0.0):
    """The tile (with halo) of `rank`.  The outer tiles extend to +-BIG so that a scan point
    outside the map's bounding box still belongs to exactly one rank."""
    points = np.asarray(points)
    axis, cuts = split_bounds(points, world_size, align, shift)
    edges = [-BIG] + cuts + [BIG]
    lo = np.full(3, -BIG)
    hi = np.full(3, BIG)
    lo[axis], hi[axis] = edges[rank], edges[rank + 1]
    c = points[:, axis].astype(np.float64)
    core = (c >= lo[axis]) & (c < hi[axis])
    keep = (c >= lo[axis] - halo) & (c < hi[axis] + halo)
    return Tile(np.ascontiguousarray(points[keep]), lo, hi, axis, int(core.sum()), float(halo))


def tile_for_method(points, rank, world_size, method, resolution=1.0, halo=None):
    """Tile + halo as pcr_set_shard wants them for `method` (include/pcr_hip.h):
    loam  cuts on the 1 m index cells, halo = the k-NN gate radius (1 m);
    ndt   cuts on the voxel faces k * resolution, halo = one voxel (two unless the resolution is a power of two);
    vgicp cuts on the voxel faces (k + 0.5) * resolution, halo wide enough for every tile point's 20 nearest
          neighbours (default max(4 m, 8 voxels); the library checks it on the device and refuses a halo that is too small)."""
    if method == "loam":
        return tile_for_rank(points, rank, world_size, halo=1.0 if halo is None else halo, align=1.0)
    if method == "ndt":
        res = float(np.float32(resolution))
        m, _ = np.frexp(res)
        return tile_for_rank(points, rank, world_size, halo=(res if m == 0.5 else 2 * res) if halo is None else halo, align=res)
    if method == "vgicp":
        return tile_for_rank(points, rank, world_size, halo=max(4.0, 8 * resolution) if halo is None else halo, align=resolution, shift=0.5)
    raise ValueError(method)


class ThreadCollective:
    """All-reduce among the threads of ONE process, for pcr_comm_init_host: `n` handles, one host thread each, exchange their
    sums through shared memory.  Sums are formed in rank order, so every rank gets the same bits.  This is how a one-GPU box
    runs the sharded path with 2 ... 8 ranks end to end (the ranks' kernels share the card); between GPUs the exchange is
    RCCL (pcr_comm_init)."""

    def __init__(self, n, timeout=120.0):
        import threading
        self.n = n
        self.buf = np.zeros((n, 64))
        self.barrier = threading.Barrier(n)
        self.timeout = timeout
        self.calls = [0] * n

    def fn(self, rank):
        def allreduce(ptr, count, op, _user):
            try:
                a = np.ctypeslib.as_array(ptr, shape=(count,))
                self.buf[rank, :count] = a
                self.barrier.wait(self.timeout)
                if op == 0:
                    acc = self.buf[0, :count].copy()
                    for r in range(1, self.n):
                        acc += self.buf[r, :count]
                else:
                    acc = self.buf[:, :count].max(0)
                self.barrier.wait(self.timeout)      # every rank has read the slots before any of them writes again
                a[:] = acc
                self.calls[rank] += 1
                return 0
            except Exception:
                return 1
        return allreduce


def gloo_collective():
    """The same exchange over torch.distributed (any initialised backend that reduces CPU tensors, e.g. gloo): one process
    per rank.  Returns the callable for PointCloudRegister.comm_init_host."""
    import torch
    import torch.distributed as dist

    def allreduce(ptr, count, op, _user):
        try:
            t = torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(count,)))      # shares the caller's memory
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
            return 0
        except Exception:
            return 1
    return allreduce


def unique_id():
    """128-byte RCCL unique id (rank 0 creates it, the caller shares it, e.g. with
    torch.distributed.broadcast_object_list)."""
    from .pcr import comm_unique_id
    return comm_unique_id()
