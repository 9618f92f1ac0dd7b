"""Debug aid for scripts/soak_parity.py: replay the listed NDT cases and compare the derivatives at the initial pose
(a mismatch there is a kernel/oracle difference; agreement there with different final poses is amplification by the
line search).   usage: soak_ndt_debug.py 0,6,31 [seed]"""
import sys
import numpy as np
from scipy.spatial.transform import Rotation as Rot
sys.path.insert(0, '.')
want = set(int(c) for c in sys.argv[1].split(","))
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
import torch  # noqa: F401
import oracle
from simpleslam_amd import NdtRegister, synth

ns = {"__name__": "soak_head"}
sys.argv = [sys.argv[0], "none", "0", str(seed0)]
exec(compile(open("scripts/soak_parity.py").read().split("total_bad = 0")[0], "soak_head", "exec"), ns)
rng = np.random.default_rng(seed0 * 1000 + len("ndt"))
for case in range(max(want) + 1):
    n_map = int(rng.choice([20_000, 100_000, 250_000])); spacing = float(rng.choice([0.2, 0.4]))
    world, m = synth.make_map(n_map, seed=seed0 * 100000 + 70000 + case, spacing=spacing)
    scan, T = synth.make_scan(world, int(rng.integers(0, 4)), seed=seed0 * 100000 + 70000 + case, beams=int(rng.choice([8, 16, 32])), azimuths=int(rng.choice([128, 256])))
    tr, rd = float(rng.choice([0.0, 0.05, 0.2, 1.0])), float(rng.choice([0.0, 0.5, 3.0]))
    init = synth.perturb(T, 13 * case + seed0, trans=tr, rot_deg=rd)
    scan, m, tag = ns["nasty"](rng, scan, m)
    res = float(rng.choice([0.5, 1.0, 2.0]))
    if case not in want:
        continue
    print(f"case {case}: map {m.shape[0]} scan {scan.shape[0]} {tag} res {res} init error {tr} m {rd} deg")
    reg = NdtRegister(ndt_resolution=res)
    reg.setTarget(m)
    p = np.concatenate([init[:3, 3], Rot.from_matrix(init[:3, :3]).as_euler("XYZ")])
    g = reg.derivatives(scan, p, double_hessian=True)
    o = oracle.ndt_derivatives(scan, m, p, oracle.ndt_params(resolution=res), double_hessian=True)
    gs, hs = np.abs(o["grad"]).max(), np.abs(o["hess"]).max()
    print(f"   score {g['score']:.9g} vs {o['score']:.9g}  rel {abs(g['score'] - o['score']) / max(1e-30, abs(o['score'])):.2e}")
    print(f"   grad rel {np.abs(g['grad'] - o['grad']).max() / gs:.2e}  hess rel {np.abs(g['hess'] - o['hess']).max() / hs:.2e}  hess_d rel {np.abs(g['hess_d'] - o['hess_d']).max() / hs:.2e}")
    pose = init.copy(); conv = reg.scan2Map(scan, m, pose)
    po, co, info = oracle.ndt_scan2map(scan, m, init, oracle.ndt_params(resolution=res))
    print(f"   iterations {reg.stats()['iterations']} vs {info['iterations']}  conv {conv} vs {co}  gpu-oracle {synth.pose_error(pose, po)}  to truth: gpu {synth.pose_error(pose, T)} oracle {synth.pose_error(po, T)}")
