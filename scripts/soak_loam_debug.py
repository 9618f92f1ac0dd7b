"""Debug aid for scripts/soak_parity.py: replay the listed LOAM cases and print the per-iteration normal equations of both
sides (condition number, step, difference).   usage: soak_loam_debug.py 46,193 [seed]"""
import sys
import numpy as np
sys.path.insert(0, '.')
want = set(int(c) for c in sys.argv[1].split(","))
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
import torch  # noqa: F401
import oracle
from simpleslam_amd import LoamRegister, synth

ns = {"__name__": "soak_head"}
sys.argv = [sys.argv[0], "none", "0", str(seed0)]
exec(compile(open("scripts/soak_parity.py").read().split("total_bad = 0")[0], "soak_head", "exec"), ns)
rng = np.random.default_rng(seed0 * 1000 + len("loam"))
np.set_printoptions(precision=6, linewidth=200)
for case in range(max(want) + 1):
    n_map = int(rng.choice([300, 2_000, 20_000, 80_000, 200_000]))
    beams, az = int(rng.choice([4, 16, 32, 64])), int(rng.choice([64, 256, 512, 1024]))
    world, m = synth.make_map(n_map, seed=seed0 * 100000 + case)
    scan, T = synth.make_scan(world, int(rng.integers(0, 6)), seed=seed0 * 100000 + case, beams=beams, azimuths=az)
    init = synth.perturb(T, 7 * case + seed0, trans=float(rng.choice([0.0, 0.05, 0.3, 1.0, 3.0])), rot_deg=float(rng.choice([0.0, 0.5, 2.0, 10.0])))
    scan, m, tag = ns["nasty"](rng, scan, m)
    kw = dict(iters=int(rng.integers(1, 14)), early_exit=int(rng.integers(0, 2)), knn_max_sq=float(rng.choice([0.25, 1.0, 4.0])),
              plane_thresh=float(rng.uniform(0.05, 0.4)), point_thresh=float(rng.uniform(0.0, 0.4)))
    if case not in want:
        continue
    print(f"case {case}: map {m.shape[0]} scan {scan.shape[0]} {tag} {kw}")
    reg = LoamRegister(loam_iters=kw["iters"], loam_early_exit=kw["early_exit"], loam_knn_max_sq=kw["knn_max_sq"],
                       loam_plane_thresh=kw["plane_thresh"], loam_point_thresh=kw["point_thresh"], record_trace=1)
    pose = init.copy()
    conv = reg.scan2Map(scan, m, pose)
    fin = ns["finite"]
    po, co, info = oracle.loam_scan2map(fin(scan), fin(m), init, oracle.loam_params(threads=16, **kw), trace=True)
    tr = reg.trace()
    print("  converged", conv, co, "iterations", tr["iters_run"], info["iters_run"], "pose error", synth.pose_error(pose, po))
    for k in range(min(tr["iters_run"], info["iters_run"])):
        A, Ao = tr["JtJ"][k], info["JtJ"][k]
        w = np.linalg.eigvalsh((A + A.T) / 2)
        print(f"  it {k}: n {tr['n'][k]} vs {info['n'][k]}  JtJ rel diff {np.abs(A - Ao).max() / max(1e-300, np.abs(Ao).max()):.2e}  JtE rel diff "
              f"{np.abs(tr['JtE'][k] - info['JtE'][k]).max() / max(1e-300, np.abs(info['JtE'][k]).max()):.2e}  eig min/max {w[0]:.3e} {w[-1]:.3e}  diag {np.diag(A)}")
        print(f"        x gpu {tr['x'][k]}\n        x ora {info['x'][k]}")
