#!/usr/bin/env python3
"""Per-wave record of vgicp_cov_kernel<true> on the bench's VGICP scan: duration of every wave against the work of its lanes' searches.
Development aid, kept for the record of profiles/r03_notes.md: it reads a table that an INSTRUMENTED build of vgicp.hip fills (s_memrealtime
at the wave's begin and end, per-lane counters of runs / steps / insertions folded over the wave, exported as pcr_cov_debug_dump) -- that
instrumentation is not in the tree; the notes describe it and what it found."""
import ctypes as C, os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simpleslam_amd import VgicpRegister, synth
from simpleslam_amd import pcr

SEED = int(re.search(r"^SEED\s*=\s*(\d+)", open(os.path.join(ROOT, "bench.py")).read(), re.M).group(1))
world, map_np = synth.make_map(1_000_000, seed=SEED + 3)
s, T = synth.make_scan(world, 0, seed=SEED + 3)
init = synth.perturb(T, SEED + 3)
dev = torch.device("cuda", 0)
d_map, d_scan = torch.from_numpy(map_np).to(dev), torch.from_numpy(s).to(dev)
reg = VgicpRegister(device=0, vgicp_resolution=0.5)
for _ in range(3):
    pose = init.copy()
    reg.scan2Map(d_scan, d_map, pose)
lib = reg._lib
if not hasattr(lib, "pcr_cov_debug_dump"):
    raise SystemExit("this library has no pcr_cov_debug_dump: see the docstring")
buf = (C.c_ulonglong * (4096 * 10))()
lib.pcr_cov_debug_dump.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.pcr_cov_debug_dump(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 10).astype(np.float64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
start, dur = (a[:, 0] - t0) / 100.0, a[:, 1] / 100.0      # s_memrealtime: 100 MHz -> us
print("waves", len(a), "kernel span us", (start + dur).max(), " wave duration us: mean", dur.mean(), "p50", np.percentile(dur, 50), "p90", np.percentile(dur, 90), "max", dur.max())
print("start offsets us: max", start.max())
names = ["steps_sum", "steps_max", "ins_sum", "ins_max", "runs_sum", "runs_max"]
for k, nm in enumerate(names):
    print(f"corr(duration, {nm}) = {np.corrcoef(dur, a[:, 3 + k])[0, 1]:.3f}   mean {a[:, 3 + k].mean():.1f} max {a[:, 3 + k].max():.0f}")
order = np.argsort(-dur)
print("slowest waves: dur, first_idx, steps_sum, steps_max, ins_sum, ins_max, runs_sum, runs_max")
for i in order[:12]:
    print(f"{dur[i]:8.1f} {int(a[i, 2]):7d} " + " ".join(f"{int(a[i, 3 + k]):8d}" for k in range(6)))
print("median waves:")
for i in order[len(order) // 2: len(order) // 2 + 6]:
    print(f"{dur[i]:8.1f} {int(a[i, 2]):7d} " + " ".join(f"{int(a[i, 3 + k]):8d}" for k in range(6)))
# least squares: duration ~ a * steps_max + b * ins_max + c * runs_max + d
X = np.stack([a[:, 4], a[:, 6], a[:, 8], np.ones(len(a))], 1)
coef, *_ = np.linalg.lstsq(X, dur, rcond=None)
print("fit dur ~ %.4f*steps_max + %.4f*ins_max + %.4f*runs_max + %.2f" % tuple(coef), " residual rms", np.sqrt(np.mean((X @ coef - dur) ** 2)))
