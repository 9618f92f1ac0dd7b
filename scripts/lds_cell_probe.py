"""Driver of scripts/probe/lds_cell_probe.hip: the LOAM candidate search of the bench workload, per-lane stream from
global memory against cell tiles staged in LDS, for the queries that sit in crowded target cells (>= 64 queries per 1 m cell).

  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/probe/liblds_cell_probe.so scripts/probe/lds_cell_probe.hip    (here, offline)
  python scripts/lds_cell_probe.py                                                                                          (on the GPU box)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simpleslam_amd import synth  # noqa: E402

S = 20261003 + 2
world, m = synth.make_map(1_000_000, seed=S)
scan, T = synth.make_scan(world, 0, seed=S)
T0 = synth.perturb(T, S)
q = (scan[:, :3].astype(np.float64) @ T0[:3, :3].T + T0[:3, 3]).astype(np.float32)

cell = 1.0
lo = np.floor(m[:, :3].min(0)).astype(np.int64) - 2
hi = np.floor(m[:, :3].max(0)).astype(np.int64) + 3
dims = (hi - lo + 1).astype(np.int64)
n_cells = int(dims.prod())


def keys(p):
    c = np.floor(p.astype(np.float64)).astype(np.int64) - lo
    return c, (c[:, 2] * dims[1] + c[:, 1]) * dims[0] + c[:, 0]


_, km = keys(m[:, :3])
order = np.argsort(km, kind="stable")
pts = np.zeros((m.shape[0] + 16, 4), np.float32)
pts[:m.shape[0], :3] = m[order, :3]
pts[:m.shape[0], 3] = order.astype(np.uint32).view(np.float32)
cell_start = np.searchsorted(km[order], np.arange(n_cells + 2)).astype(np.uint32)

cq, kq = keys(q)
inside = np.all((cq >= 1) & (cq <= dims - 2), axis=1)
q, kq = q[inside], kq[inside]
cnt = np.bincount(kq, minlength=n_cells)
crowded = cnt[kq] >= 64
print(f"map {m.shape[0]} points in {n_cells} cells; {q.shape[0]} queries inside, {np.count_nonzero(np.unique(kq).size)} cells hold queries ({np.unique(kq).size}); "
      f"{crowded.mean() * 100:.1f} % of the queries sit in the {np.count_nonzero(cnt >= 64)} cells with >= 64 queries")
cand = cell_start[1:] - cell_start[:-1]


def block_candidates(k):      # points in the 3 x 3 x 3 block of cell k
    d0, d1 = int(dims[0]), int(dims[1])
    tot = np.zeros(k.shape[0], np.int64)
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            row = k + dz * d1 * d0 + dy * d0
            tot += cell_start[row + 2].astype(np.int64) - cell_start[row - 1].astype(np.int64)
    return tot


bc = block_candidates(kq)
print(f"candidates per query: mean {bc.mean():.0f}, crowded {bc[crowded].mean():.0f}, sparse {bc[~crowded].mean():.0f}, max {bc.max()}")

lib = C.CDLL(os.path.join(ROOT, "scripts", "probe", "liblds_cell_probe.so"))
lib.probe_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int), C.c_void_p, C.c_uint,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
d_pts = torch.from_numpy(pts).cuda()
d_cs = torch.from_numpy(cell_start.view(np.int32)).cuda()
org = (C.c_double * 3)(*[float(v) for v in lo])
dm = (C.c_int * 3)(*[int(v) for v in dims])


def q4(a):
    out = np.zeros((a.shape[0], 4), np.float32)
    out[:, :3] = a
    return torch.from_numpy(out).cuda()


def run(mode, dq, nq, blocks=None, reps=200):
    out = torch.zeros(max(nq, 1), dtype=torch.int32, device="cuda")
    us = C.c_float(0)
    bf = bc_ = bk = None
    nb = 0
    if blocks is not None:
        bf, bc_, bk = (torch.from_numpy(np.asarray(b, np.uint32).view(np.int32)).cuda() for b in blocks)
        nb = bf.shape[0]
    rc = lib.probe_run(mode, d_pts.data_ptr(), d_cs.data_ptr(), org, cell, dm, dq.data_ptr(), nq,
                       bf.data_ptr() if nb else None, bc_.data_ptr() if nb else None, bk.data_ptr() if nb else None, nb, out.data_ptr(), reps, C.byref(us))
    assert rc == 0, rc
    return us.value, out.cpu().numpy().view(np.uint32)


# every query, the scan's own order (what loam_iterate_kernel sees)
t_all, ref_all = run(0, q4(q), q.shape[0])
# round 4: four / two adjacent lanes per query, lists merged in registers
t_q4, got4 = run(2, q4(q), q.shape[0])
t_q2, got2 = run(3, q4(q), q.shape[0])
assert np.array_equal(got4, ref_all) and np.array_equal(got2, ref_all), "the multi-lane searches disagree with the one-lane stream"
print(f"one lane per query {t_all:7.2f} us | two lanes per query {t_q2:7.2f} us | four lanes per query {t_q4:7.2f} us   (all {q.shape[0]} queries, scan order, same result words)")
# the crowded half: scan order, cell order, staged
qc, kc = q[crowded], kq[crowded]
t_c, ref = run(0, q4(qc), qc.shape[0])
o = np.argsort(kc, kind="stable")
qs, ks = qc[o], kc[o]
t_cs, ref_s = run(0, q4(qs), qs.shape[0])
first, count, bcell = [], [], []
u, start, n_in = np.unique(ks, return_index=True, return_counts=True)
for k, s0, n in zip(u, start, n_in):
    for off in range(0, n, 256):
        first.append(s0 + off); count.append(min(256, n - off)); bcell.append(k)
t_b, got = run(1, q4(qs), qs.shape[0], (first, count, bcell))
assert np.array_equal(got, ref_s), "staged and streamed searches disagree"
assert np.array_equal(np.sort(ref), np.sort(ref_s))
# the sparse rest, scan order
t_s, _ = run(0, q4(q[~crowded]), int((~crowded).sum()))
print(f"per-lane stream, all {q.shape[0]} queries, scan order        : {t_all:7.2f} us")
print(f"per-lane stream, {qc.shape[0]} crowded queries, scan order    : {t_c:7.2f} us")
print(f"per-lane stream, crowded queries sorted by target cell   : {t_cs:7.2f} us")
print(f"LDS-staged, crowded queries, {len(first)} blocks of one cell     : {t_b:7.2f} us   (mean {np.mean(count):.0f} queries per block)")
print(f"per-lane stream, the {int((~crowded).sum())} sparse queries, scan order  : {t_s:7.2f} us")
print(f"=> staged crowded + streamed sparse, run one after the other: {t_b + t_s:7.2f} us against {t_all:.2f} us for the one stream -- before the cost of "
      f"sorting the queries by cell on the device (a 65 k-point index build: 35-60 us, profiles/r03_notes.md)")
