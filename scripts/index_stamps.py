"""Development aid (DEV library): where a block of the bin / tile kernel of the index build spends its time (s_memrealtime stamps)."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simpleslam_amd
from simpleslam_amd import LoamRegister, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
method = sys.argv[2] if len(sys.argv) > 2 else "loam"      # ndt: BASELINE configs[4]'s shapes (128 beams, 0.22 m map), region-only index from the second call on
S = 20261003 + 2
if method == "ndt":
    from simpleslam_amd import NdtRegister
    w, m = synth.make_map(n, seed=S, spacing=0.22)
    scan, T = synth.make_scan(w, 0, seed=S, beams=128, azimuths=1024)
    T0 = synth.perturb(T, S, trans=0.1, rot_deg=0.5)
    reg = NdtRegister()
else:
    w, m = synth.make_map(n, seed=S)
    scan, T = synth.make_scan(w, 0, seed=S)
    T0 = synth.perturb(T, S)
    reg = LoamRegister(loam_iters=1, loam_early_exit=0)
ds, dm = torch.from_numpy(scan).cuda(), torch.from_numpy(m).cuda()
L = simpleslam_amd.load_library()
for i in range(6):
    pose = T0.copy(); reg.scan2Map(ds, dm, pose)
L.pcr_dev_read_stamps(None, 0)      # arm + clear
pose = T0.copy(); reg.scan2Map(ds, dm, pose)
buf = np.zeros(2 * 8192 * 8, np.uint64)
L.pcr_dev_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf.reshape(2, 8192, 8).astype(np.int64)
b0 = st[1][0]
if b0[0] > 0 and b0[7] > 0:
    print(f"tile kernel, block 0 (the planner): counts + tile words in LDS after {(b0[5] - b0[0]) / 100:.1f} us, decided + scanned after {(b0[6] - b0[0]) / 100:.1f} us, layout written after {(b0[7] - b0[0]) / 100:.1f} us")
for k, name, labels in ((0, "bin", ["start", "keys+lds done", "barrier", "claims back", "chunk loop done", "ticket", "last block done"]), (1, "tile", ["start", "loaded+hist", "scanned", "stored", "end"])):
    a = st[k][1:] if k == 1 else st[k]; used = a[:, 0] > 0; a = a[used][:, :5] if k == 1 else a[used]
    if not len(a): continue
    t0 = a[:, 0].min()
    print(f"{name} kernel: {len(a)} blocks, span {(a.max() - t0) / 100:.1f} us; block starts: min 0 median {np.median(a[:, 0] - t0) / 100:.1f} max {(a[:, 0].max() - t0) / 100:.1f} us")
    for j in range(1, len(labels)):
        ok = a[:, j] > 0
        if ok.any(): print(f"   {labels[j]:18s} since block start: median {np.median(a[ok, j] - a[ok, 0]) / 100:6.1f} p90 {np.percentile(a[ok, j] - a[ok, 0], 90) / 100:6.1f} max {(a[ok, j] - a[ok, 0]).max() / 100:6.1f} us   (latest at {(a[ok, j].max() - t0) / 100:.1f} us of the kernel)   n={ok.sum()}")
