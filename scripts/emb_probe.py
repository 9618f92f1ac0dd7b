import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench, numpy as np, torch
args = bench.parse()
def vg(tag):
    r = bench.secondary(args, method="vgicp", steps=40, warmup=5, embedded=True)
    print(tag, round(r["ms_per_step"], 4), round(r["roofline"]["target_prep_ms"], 4), flush=True)
vg("A fresh process")
from simpleslam_amd import LoamRegister, synth, pcr as _pcr
world, map_np = synth.make_map(1_000_000, seed=1)
s, T = synth.make_scan(world, 0, seed=1)
dev = torch.device("cuda", 0)
d_map, d_scan = torch.from_numpy(map_np).to(dev), torch.from_numpy(s).to(dev)
reg = LoamRegister(device=0, loam_iters=10, loam_early_exit=0)
for i in range(30):
    p = T.copy(); reg.scan2Map(d_scan, d_map, p)
vg("B after 30 LOAM scans (handle alive)")
reg.set_profile(2)
for i in range(5):
    p = T.copy(); reg.scan2Map(d_scan, d_map, p)
reg.set_profile(0)
vg("C after profile level 2")
mp = np.array(map_np, copy=True); _pcr.host_pin(mp)
for i in range(5):
    p = T.copy(); reg.scan2Map(s, mp, p)
_pcr.host_unpin(mp)
vg("D after host-buffer calls with a pinned map")
reg.setTarget(d_map)
for i in range(10):
    p = T.copy(); reg.align(d_scan, p)
reg.invalidateTarget()
vg("E after setTarget / align / invalidateTarget")
reg_nh = LoamRegister(device=0, loam_iters=10, loam_early_exit=0, index_no_hints=1)
for i in range(10):
    p = T.copy(); reg_nh.scan2Map(d_scan, d_map, p)
del reg_nh
vg("F after a second LOAM handle (no hints) created and deleted")
for i in range(5):
    p = T.copy(); reg.scan2Map(s, map_np, p)
vg("G after host-buffer calls from pageable memory")
x = torch.zeros(1, device=dev); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e0.record(); torch.cuda.synchronize()
vg("H after torch events")
del d_map, d_scan
vg("I after freeing the LOAM clouds")
