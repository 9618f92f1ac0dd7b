#!/usr/bin/env python3
"""bench.py -- scans/s of the LOAM scan-to-map hot path on MI355X.

Workload (BASELINE.json configs[1]): pcr=loam, 65 536-point scan vs 1 M-point sub-map,
10 Gauss-Newton iterations with early exit off, target index rebuilt on every call (as the
reference does, PCR/src/LoamRegister.cpp:110), inputs resident in HBM.  One "step" = one
scan2Map call through the C ABI (pcr_scan2map_device).

  python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: one rank per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this
process IS a rank; started plainly (`python bench.py --gpus N`) it is the LAUNCHER: before anything
touches the GPU it starts N fresh child processes (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_* set), waits
for them and relays rank 0's JSON line (launch_ranks).  `--dry-run` makes every rank report its
environment instead of measuring (no GPU needed: the CPU suite drives the launcher that way).
Every rank holds a full map replica and registers its own scans (scans are independent objects: no
data-path collective) -> weak scaling, value = all ranks' scans / max-over-ranks time.
`--shard-map` instead shards the map tiles across ranks with an RCCL all-reduce of the normal
equations per iteration (BASELINE.json configs[3]: 10 M-point map by default); it is not the default
line.  The line says how many ranks really ran: n_gpus = WORLD_SIZE, rccl_ranks = what the
communicator reports.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SCAN = 65_536
SEED = 20261003
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def host_cores():
    """CPU cores this process may actually use: the cgroup quota when there is one (a GPU box grants a share
    of its host), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def timed_windows(step, barrier, steps, windows):
    """`windows` back-to-back timed windows of exactly `steps` steps, each bracketed by barrier + synchronize -> seconds per window"""
    out = []
    for w in range(windows):
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        barrier()
        out.append(time.perf_counter() - t0)
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--map-points", type=int, default=None, help="default 1 000 000 (configs[1]); 10 000 000 with --shard-map (configs[3])")
    ap.add_argument("--dry-run", action="store_true", help="every rank prints the environment it was started with and leaves: no GPU, no measurement")
    ap.add_argument("--master-port", type=int, default=0, help="launcher only: rendezvous port of the ranks it starts (0 = pick a free one)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--scans", type=int, default=8, help="distinct synthetic scans cycled through")
    ap.add_argument("--shard-map", action="store_true", help="shard map tiles across ranks + RCCL all-reduce (config 4)")
    ap.add_argument("--transport", choices=["rccl", "peer"], default="rccl",
                    help="--shard-map, loam: the exchange of the 32 sums -- rccl (reduce kernel + ncclAllReduce) or peer (pcr_comm_init_peer: the ranks push into "
                         "each other's receive buffers mapped with hipIpc, one launch per linearisation; prototype)")
    ap.add_argument("--method", choices=["loam", "vgicp", "ndt"], default="loam",
                    help="loam = the headline line (BASELINE configs[1]); vgicp / ndt = configs[2] / configs[4], extra lines")
    ap.add_argument("--streams", type=int, default=1,
                    help="> 1: additionally time S independent handles (one HIP stream and one host thread each) registering "
                         "scans concurrently on the same GPU; reported as \"concurrent\", never as \"value\"")
    ap.add_argument("--full-target", action="store_true", help="vgicp / ndt: prepare the whole target on every call (pcr_params.full_target = 1) instead of the scan's region, for A/B runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=8.0, help="wall-clock bound of EACH CPU leg (per thread count / index flavour / method)")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; value = the median window")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[2] (vgicp) and configs[4] (ndt) lines embedded as \"extra\"")
    ap.add_argument("--extra-steps", type=int, default=40)
    ap.add_argument("--secondary-map-points", type=int, default=0, help="--method vgicp|ndt: map size instead of the configuration's (rehearsals and tests)")
    ap.add_argument("--no-twin", action="store_true", help="skip the stateless twins (pcr_params.index_no_hints = 1) and the host-buffer legs: counter passes "
                                                        "(scripts/profile_round.sh) then see the dispatches of ONE leg, the one the line's figures are of")
    ap.add_argument("--sequence-scans", type=int, default=64, help="scans of the drive behind extra.sequence (the caller's workload: key frames, sub-map assembly, "
                                                                    "pcr_scan2map_submap with init = previous pose); 0 skips it")
    args = ap.parse_args()
    if args.map_points is None:
        args.map_points = 10_000_000 if args.shard_map else 1_000_000
    return args


def rank_env():
    return {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N ranks of this script, one per GPU, the way
    torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), wait for them and print rank 0's
    JSON line.  This process never imports torch and never touches the GPU: the ranks are fresh children (a process that has
    initialised the GPU must not be replaced or forked).  Returns the exit code."""
    import socket
    import subprocess
    n = args.gpus
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PCR_BENCH_LAUNCHER="bench.py")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0 = ""
    try:
        # rank 0's stdout carries the line; the others' goes to stderr.  A rank that dies takes the others with it (they would wait
        # for it in a collective for ever): poll, and end exactly the processes started here.
        import threading
        box = {}
        t = threading.Thread(target=lambda: box.setdefault("out", procs[0].stdout.read()), daemon=True)
        t.start()
        rc = 0
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(0.05)
        if rc:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
        t.join(10)
        out0 = box.get("out", "") or ""
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{"):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if rc:
        print(f"bench.py launcher: a rank exited with code {rc}", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py launcher: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def dry_run(args):
    """--dry-run: what this rank was started with.  Rank 0 gathers every rank's report over gloo (which also proves the
    rendezvous the real run would use) and prints one line."""
    me = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
          "world_size": int(os.environ.get("WORLD_SIZE", "1")), "pid": os.getpid(), "env": rank_env(),
          "launcher": os.environ.get("PCR_BENCH_LAUNCHER", "external (torch.distributed.run)" if "WORLD_SIZE" in os.environ else "none")}
    ws = me["world_size"]
    if os.environ.get("PCR_BENCH_DRYRUN_FAIL_RANK") == str(me["rank"]):      # test hook: a rank that dies before the rendezvous
        return 3
    reports = [me]
    if ws > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=me["rank"], world_size=ws)
        reports = [None] * ws
        dist.all_gather_object(reports, me)
        dist.barrier()
        dist.destroy_process_group()
    if me["rank"] == 0:
        print(json.dumps({"dry_run": True, "n_gpus": ws, "gpus_arg": args.gpus, "shard_map": bool(args.shard_map),
                          "map_points": args.map_points, "ranks": reports}), flush=True)
    return 0


def join_ranks(torch, dist, rank, local_rank, world_size, rehearse):
    """Bind this rank to its GPU and join the process group.  -> (device, rccl_ranks): rccl_ranks is the number of ranks that took
    part in an RCCL all-reduce issued right here (a sum of ones over the "nccl" group -- measured, not read from argv); 1 for a
    single rank, 0 in a rehearsal (several ranks on ONE card exchange over gloo: RCCL refuses one device twice)."""
    n_dev = torch.cuda.device_count()
    if rehearse:
        local_rank = local_rank % max(1, n_dev)
    elif local_rank >= n_dev:
        raise SystemExit(f"bench.py: rank {rank} is to use GPU {local_rank} but this node shows {n_dev}; "
                         "PCR_BENCH_REHEARSE=1 lets the ranks share a card (a rehearsal of the launch path, not a measurement)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size == 1:
        return dev, 1
    if rehearse:
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
        return dev, 0
    dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(ones)
    return dev, int(round(float(ones.item())))


def gather_ranks(dist, world_size, info):
    """every rank's `info` dict, in rank order, on every rank"""
    if world_size == 1:
        return [info]
    out = [None] * world_size
    dist.all_gather_object(out, info)
    return out


def launcher_name(world_size):
    return os.environ.get("PCR_BENCH_LAUNCHER", "torch.distributed.run (external)" if world_size > 1 else "none")


def secondary(args, method=None, steps=None, warmup=None, embedded=False):
    """configs[2] (pcr=vgicp, 0.5 m voxels, 65 536 x 1 M) and configs[4] (pcr=ndt, 1 m cells, 131 072 x 5 M) on one GPU:
    same timing contract, target rebuilt on every call like the reference (fast_vgicp_impl.hpp:66-67; ndt_omp.h:276-283).
    Printed as the line of `--method vgicp|ndt`, or returned to be embedded as "extra" in the LOAM line (embedded=True: one GPU,
    no process group of its own)."""
    method = method or args.method
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    import torch
    import oracle
    from simpleslam_amd import VgicpRegister, NdtRegister, synth
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # replicas, as in main(): one process per GPU, each with the whole map and its own scans, no data-path collective
    rank, local_rank, world_size = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if embedded:
        rank, world_size = 0, 1
    rehearse = os.environ.get("PCR_BENCH_REHEARSE") == "1" and world_size > 1
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if embedded:
        torch.cuda.set_device(local_rank)
        dev, rccl_ranks = torch.device("cuda", local_rank), 1
    else:
        dev, rccl_ranks = join_ranks(torch, dist, rank, local_rank, world_size, rehearse)
        local_rank = dev.index
    if method == "vgicp":
        cfg, n_map, kw, mk = 3, 1_000_000, {}, {}
        reg = VgicpRegister(device=local_rank, vgicp_resolution=0.5, full_target=int(args.full_target))
        pert = {}
        cores = host_cores()
        ref = lambda s, m, T, th: oracle.vgicp_scan2map(s, m, T, oracle.vgicp_params(resolution=0.5, threads=th))[0]
        # one-off per target >= N_m (16 + 20*16 + 128) for the covariances + N_m (16 + 128) for the voxel map
        alg = lambda n_s, n_m: 608 * n_m
        what = "target preparation (index levels + covariance + voxel kernels), 608 B per map point"
        workload = ("pcr=vgicp, 0.5 m voxels, 65536-pt scan vs 1000000-pt submap, target rebuilt per call (covariances and voxels for the scan's region only; "
                    "the whole target with --full-target), inputs in HBM")
    else:
        cfg, n_map, kw, mk = 5, 5_000_000, dict(beams=128, azimuths=1024), dict(spacing=0.22)
        reg = NdtRegister(device=local_rank, full_target=int(args.full_target))
        pert = dict(trans=0.1, rot_deg=0.5)
        # computeDerivatives on `cores` threads as the reference runs it (ndt_omp_impl.hpp:206, NdtRegister.cpp:18); the voxel grid,
        # computeHessian and the line-search bookkeeping are serial there and here
        ref = lambda s, m, T, th: oracle.ndt_scan2map(s, m, T, oracle.ndt_params(threads=th))[0]
        cores = host_cores()
        alg = lambda n_s, n_m: 16 * n_m          # voxel build reads every map point once (+104 B per voxel written)
        what = "target preparation (index + voxel Gaussians), 16 B per map point"
        workload = ("pcr=ndt, 1.0 m cells, 131072-pt 128-beam scan vs 5000000-pt submap, target rebuilt per call (voxel Gaussians for the scan's region only, "
                    "from a handle's second call on the index holds the region's points only), inputs in HBM")
    if args.secondary_map_points and not embedded:
        n_map = int(args.secondary_map_points)
        workload = workload.replace("1000000-pt submap", f"{n_map}-pt submap").replace("5000000-pt submap", f"{n_map}-pt submap")      # (the line names the map it ran on)
    world, map_np = synth.make_map(n_map, seed=SEED + cfg, **mk)
    scans, inits = [], []
    for j in range(args.scans):
        k = rank * args.scans + j
        s, T = synth.make_scan(world, k, seed=SEED + cfg, **kw)
        scans.append(s); inits.append(synth.perturb(T, SEED + cfg + k, **pert))
    d_map = torch.from_numpy(map_np).to(dev)
    d_scans = [torch.from_numpy(s).to(dev) for s in scans]
    d_map_full = d_map
    scaling = "weak"
    parallelism = (f"replicas x{world_size}: THE scaling mode of this path -- one process per GPU, every rank the whole map and its own scans, no data-path collective "
                   "(weak scaling); --shard-map is the other mode, one call's map cut over the ranks, and is latency-bound (DESIGN.md 5)") if world_size > 1 else "single GPU"
    rank_info = {"rank": rank, "device": local_rank, "map_points": int(map_np.shape[0])}
    sharded = bool(args.shard_map) and not embedded
    if sharded:
        # BASELINE configs[4] ("1 -> 8 GPU scaling curve") / configs[2] sharded: the map cut into tiles on the method's voxel lattice, every rank
        # prepares its tile + halo and handles the scan points that land in it, one all-reduce of the 43 sums per evaluation pass
        # (reference loops being sharded: ndt_omp_impl.hpp:206-285, fast_vgicp_impl.hpp:135-177)
        from simpleslam_amd import shard
        res = 0.5 if method == "vgicp" else 1.0
        tile = shard.tile_for_method(map_np, rank, world_size, method, resolution=res)
        d_map = torch.from_numpy(tile.points).to(dev)
        reg.set_shard(tile.lo, tile.hi, tile.halo)
        if args.transport == "peer":      # (the device-resident loops over the peer exchange: one more launch per pass, no host round trip; works in a rehearsal too)
            handles = [None] * world_size
            if world_size > 1:
                dist.all_gather_object(handles, reg.comm_peer_export())
            else:
                handles = [reg.comm_peer_export()]
            reg.comm_init_peer(handles, rank, world_size)
        elif rehearse:      # all ranks on one card: RCCL refuses a communicator with one device twice -> the exchange goes through gloo
            reg.comm_init_host(shard.gloo_collective(), rank, world_size)
        else:
            uid = [shard.unique_id() if rank == 0 else None]
            if world_size > 1:
                dist.broadcast_object_list(uid, src=0)
            reg.comm_init(uid[0], rank, world_size)
        # one scan is split over the tiles: the SAME scans and guesses on every rank (rank 0's)
        scans, inits = [], []
        for j in range(args.scans):
            sc, T = synth.make_scan(world, j, seed=SEED + cfg, **kw)
            scans.append(sc); inits.append(synth.perturb(T, SEED + cfg + j, **pert))
        d_scans = [torch.from_numpy(sc).to(dev) for sc in scans]
        scaling = "strong"
        ci = reg.comm_info()
        rccl_ranks = ci["nranks"] if ci["transport"] == "rccl" else 0
        parallelism = f"map tiles x{world_size} + all-reduce of the 43 sums per evaluation pass over {ci['transport']}"
        rank_info.update(tile_points=int(tile.points.shape[0]), tile_core_points=int(tile.n_core), tile_axis=int(tile.axis),
                         tile_lo=float(tile.lo[tile.axis]), tile_hi=float(tile.hi[tile.axis]), halo=float(tile.halo), transport=ci["transport"])

    def step(i):
        pose = inits[i % args.scans].copy()
        reg.scan2Map(d_scans[i % args.scans], d_map, pose)
        return pose

    reg.set_profile(0)
    for i in range(warmup):
        step(i)

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wins = timed_windows(step, barrier, steps, max(1, args.windows))
    if world_size > 1:
        t = torch.tensor(wins, dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wins = [float(v) for v in t.tolist()]
    elapsed = float(np.median(wins))
    if sharded:
        # every rank has to take part in every further call (each one is a chain of collectives): the per-rank report and, on rank 0 alone,
        # the N = 1 origin of the curve (the SAME scans against the WHOLE map, an unsharded handle) while the others wait at a barrier
        reg.set_profile(1)
        prep = aln = 0.0
        for i in range(4):
            step(i); st = reg.stats(); prep += st["index_ms"] / 4; aln += st["solve_ms"] / 4
        reg.set_profile(0)
        rank_info.update(target_prep_ms=prep, align_ms=aln)
        all_ranks = gather_ranks(dist, world_size, rank_info)
        n1 = None
        if world_size >= 1:      # (one rank too: the same call with and without the exchange on one card is what the exchange costs)
            if rank == 0:
                cls = VgicpRegister if method == "vgicp" else NdtRegister
                reg1 = cls(device=local_rank, **(dict(vgicp_resolution=0.5) if method == "vgicp" else {}))
                reg1.set_profile(0)

                def step1(i):
                    pose = inits[i % args.scans].copy(); reg1.scan2Map(d_scans[i % args.scans], d_map_full, pose)
                for i in range(5):
                    step1(i)
                w1 = timed_windows(step1, lambda: torch.cuda.synchronize(), steps, 3)
                e1 = float(np.median(w1))
                n1 = {"value": steps / e1, "unit": "scans/s", "ms_per_step": 1e3 * e1 / steps,
                      "note": "same scans, whole map, one GPU, unsharded handle (the N = 1 point of this strong-scaling curve)"}
                if world_size == 1:
                    # what the exchange costs before any link is crossed: the ALIGNMENT of the sharded call against the alignment of the unsharded one (a sharded
                    # target is always prepared in full, the unsharded one for the scan's region: the preparations are not comparable), per evaluation pass
                    passes = max(1, int(reg.stats().get("attempts", 0)))
                    reg1.set_profile(1)
                    aln1 = 0.0
                    for i in range(8):
                        step1(i); aln1 += reg1.stats()["solve_ms"] / 8
                    reg1.set_profile(0)
                    n1["align_ms"] = aln1
                    n1["exchange_us_per_pass"] = 1e3 * (aln - aln1) / passes
                    n1["passes_last_call"] = passes
                    n1["exchange_note"] = ("(alignment of the sharded call - alignment of the unsharded call) / evaluation passes of the last call, one rank on one card: "
                                           "the sharded loop's extra launch per pass and its exchange")
                del reg1
            if world_size > 1:
                dist.barrier()
    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return
    out = {"metric": f"scans/s ({workload.split(',')[0]}, BASELINE configs[{cfg - 1}])", "value": steps * (1 if sharded else world_size) / elapsed, "unit": "scans/s",
           "n_gpus": world_size, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "windows_ms": [1e3 * v for v in wins],
           "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64" if method == "vgicp" else "f32",
           "data": "synthetic" + (" (REHEARSAL: all ranks on one card)" if rehearse else ""),
           "config": {"workload": workload, "scans_cycled": args.scans, "parallelism": parallelism},
           "rccl_ranks": rccl_ranks, "gpus_arg": args.gpus, "launcher": launcher_name(world_size)}
    if sharded:
        out["ranks"] = all_ranks
        if n1:
            out["n1"] = n1
        print(json.dumps(out), flush=True)
        if world_size > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    reg.set_profile(1)
    idx_ms = sol_ms = 0.0
    for i in range(8):
        step(i); st = reg.stats(); idx_ms += st["index_ms"] / 8; sol_ms += st["solve_ms"] / 8
    reg.set_profile(0)
    # calls so far whose pose left the region the target had been prepared for and were repeated on the whole target (pcr_stats.region_repeats)
    out["region_repeats"] = int(reg.stats().get("region_repeats", 0))
    out["region_index"] = int(reg.stats().get("region_index", 0))      # NDT: the last call indexed only the target points of the scan's region
    # ---- rooflines computed from the work DONE: a profiling pass (pcr_set_profile 2) puts events at the kernels' own begin and end and has
    #      the kernels count what they processed (pcr_stats: region_points / region_voxels, pairs_grad / pairs_hess) ----
    reg.set_profile(2)
    prof = {"kernel_ms": 0.0, "aux_kernel_ms": 0.0, "region_points": 0.0, "region_voxels": 0.0, "pairs_grad": 0.0, "pairs_hess": 0.0, "kernel_launches": 0.0, "attempts": 0.0}
    n_prof = 8
    for i in range(n_prof):
        step(i); st = reg.stats()
        for k in prof:
            prof[k] += st[k] / n_prof
    reg.set_profile(0)
    # HBM traffic and SQ counters are NOT measured in this run (counters need passes of their own under rocprofv3): they are read from the summary the
    # newest committed profile round left for this method -- the line says which file, and that file says which leg it profiled
    pmc, pmc_src = {}, None
    for rnd in ("r05", "r04"):
        pj = os.path.join(ROOT, "profiles", f"{rnd}_{method}_pmc.json")
        if os.path.exists(pj):
            try:
                pmc = json.load(open(pj)); pmc_src = f"profiles/{rnd}_{method}_pmc.json ({pmc.get('leg', 'hinted and stateless calls mixed: the pass ran both legs')}); not measured in this run"
                break
            except Exception:
                pmc = {}
    n_s = scans[0].shape[0]
    if method == "vgicp":
        # covariance of one point 16 + 20 x 16 + 128 = 464 B, voxel map 16 + 128 = 144 B per point, an index build 32 B per point
        k_s = prof["kernel_ms"] * 1e-3
        cov_bytes = 464.0 * prof["region_points"]
        ach = cov_bytes / k_s / 1e9 if k_s > 0 else 0.0
        prep_bytes = 2 * 32.0 * n_map + 464.0 * prof["region_points"] + 144.0 * prof["region_points"]
        scan_bytes = 464.0 * n_s + 2 * 32.0 * n_s
        out["roofline"] = {
            "bound": "hbm", "kernel": "vgicp_cov_kernel<false> (the target's covariances inside the scan's region: the critical path of the preparation)",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": cov_bytes, "avg_launch_us": prof["kernel_ms"] * 1e3,
            "traffic": pmc.get("hbm_bytes_per_launch_cov_target"),
            "region_points": prof["region_points"], "region_voxels": prof["region_voxels"], "map_points": n_map,
            "preparation": {"what": "2 index builds x 32 B x map points + (464 + 144) B x region points (covariances + voxel map), over target_prep_ms",
                            "algorithmic_bytes": prep_bytes, "achieved": prep_bytes / (idx_ms * 1e-3) / 1e9, "frac": prep_bytes / (idx_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "traffic": pmc.get("hbm_bytes_per_scan_preparation")},
            "scan_search": {"what": "the scan's own covariances (csrc/cov_search.hip: three kernels on the side stream), 464 B x scan points",
                            "kernels_us": prof["aux_kernel_ms"] * 1e3, "algorithmic_bytes": 464.0 * n_s,
                            "achieved": 464.0 * n_s / (prof["aux_kernel_ms"] * 1e-3) / 1e9 if prof["aux_kernel_ms"] > 0 else None,
                            "frac": 464.0 * n_s / (prof["aux_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if prof["aux_kernel_ms"] > 0 else None},
            "target_prep_ms": idx_ms, "align_ms": sol_ms}
    else:
        # ndt_pass_pro_kernel is bound by arithmetic: flop per (point, voxel) pair counted from ndt.hip: ndt_derivatives_body -- 130 for a pass
        # that accumulates score + gradient, 425 with the float Hessian -- against the 157.3 TFLOP/s f32 vector peak
        flop = 130.0 * prof["pairs_grad"] + 425.0 * prof["pairs_hess"]
        k_s = prof["kernel_ms"] * 1e-3
        ach = flop / k_s / 1e12 if k_s > 0 else 0.0
        prep_bytes = alg(n_s, n_map)
        out["roofline"] = {
            "bound": "valu_f32", "kernel": "ndt_pass_pro_kernel (computeDerivatives / computeHessian passes of the device-resident Newton + More-Thuente loop)",
            "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3,
            "flop_per_pair": {"gradient_pass": 130, "hessian_pass": 425}, "pairs_per_scan": {"gradient_passes": prof["pairs_grad"], "hessian_passes": prof["pairs_hess"]},
            "launches_per_scan": prof["kernel_launches"], "avg_launch_us": prof["kernel_ms"] * 1e3 / max(1.0, prof["kernel_launches"]), "kernel_us_per_scan": prof["kernel_ms"] * 1e3,
            "valu_active": pmc.get("valu_active_frac"), "traffic": None,
            "preparation": {"what": what, "algorithmic_bytes": prep_bytes, "achieved": prep_bytes / (idx_ms * 1e-3) / 1e9, "unit": "GB/s",
                            "frac": prep_bytes / (idx_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc.get("hbm_bytes_per_scan_preparation")},
            "target_prep_ms": idx_ms, "align_ms": sol_ms}
    out["roofline"]["traffic_source"] = pmc_src
    if args.full_target and method == "vgicp":
        out["roofline"]["note_full_target"] = "--full-target: no region exists, the kernels' counters of region points / voxels do not run; the covariance roofline above is not meaningful here"
    out["roofline"]["note"] = ("target_prep_ms / align_ms come from a separate 8-scan pass with phase events (pcr_set_profile 1); kernel durations and the "
                               "counts of what was processed from another with events at the kernels' own begin and end (pcr_set_profile 2)")
    # ---- the same scans with nothing carried across calls (pcr_params.index_no_hints = 1): the stateless twin of `value` ----
    if world_size == 1 and not args.no_twin:
        cls = VgicpRegister if method == "vgicp" else NdtRegister
        kw_nh = dict(vgicp_resolution=0.5) if method == "vgicp" else {}
        reg_nh = cls(device=local_rank, index_no_hints=1, full_target=int(args.full_target), **kw_nh)
        reg_nh.set_profile(0)

        def step_nh(i):
            pose = inits[i % args.scans].copy(); reg_nh.scan2Map(d_scans[i % args.scans], d_map, pose)
        for i in range(5):
            step_nh(i)
        n_nh = max(10, steps // 2)
        w_nh = timed_windows(step_nh, lambda: torch.cuda.synchronize(), n_nh, 3)
        e_nh = float(np.median(w_nh)) / n_nh
        out["no_index_hints"] = {"value": 1.0 / e_nh, "unit": "scans/s", "ms_per_step": 1e3 * e_nh,
                                 "note": "pcr_params.index_no_hints = 1: fresh bounding boxes and build launches on every call, no tile layout, no region-only index; nothing carried across calls"}
        del reg_nh
    if not args.no_cpu_baseline and world_size == 1:      # the CPU leg is timed at N = 1 only
        def cpu_leg(threads, budget, parity):
            n_done, t_cpu, et, er, nan_both, nan_one = 0, 0.0, [], [], 0, 0
            while n_done < 2 or (t_cpu < budget and n_done < args.scans):
                j = n_done % args.scans
                c0 = time.perf_counter(); pref = ref(scans[j], map_np, inits[j], threads); t_cpu += time.perf_counter() - c0
                n_done += 1
                if not parity:
                    continue
                pg = step(j)
                if not (np.isfinite(pg).all() and np.isfinite(pref).all()):
                    nan_both += int(not np.isfinite(pg).all() and not np.isfinite(pref).all())
                    nan_one += int(np.isfinite(pg).all() != np.isfinite(pref).all())
                    continue
                dt, dr = synth.pose_error(pg, pref)
                et.append(dt); er.append(dr)
            return n_done, t_cpu, et, er, nan_both, nan_one

        by_cores = {}
        for th in sorted({1, min(4, cores)} - {cores}):      # 1 thread and the reference's default `cores` = 4 (config/params.json:5)
            n_d, t_c, *_ = cpu_leg(th, args.cpu_budget_s / 2, False)
            by_cores[str(th)] = {"value": n_d / t_c, "scans": n_d, "wall_s": t_c}
        n_done, t_cpu, et, er, nan_both, nan_one = cpu_leg(cores, args.cpu_budget_s, True)
        by_cores[str(cores)] = {"value": n_done / t_cpu, "scans": n_done, "wall_s": t_cpu}
        out["cpu_baseline"] = {"value": n_done / t_cpu, "unit": "scans/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
                               "sample": f"{n_done} of the same scans through the oracle, {t_cpu:.1f} s wall, OpenMP threads = {cores}",
                               "by_cores": by_cores}
        out["pose_rmse_vs_cpu"] = {"trans_m": float(np.sqrt(np.mean(np.square(et)))) if et else None,
                                   "rot_rad": float(np.sqrt(np.mean(np.square(er)))) if er else None, "scans": len(et),
                                   # pclomp's line search can return NaN (ndt_omp_impl.hpp:773-932); both sides then agree on it
                                   "non_finite_on_both_sides": nan_both, "non_finite_on_one_side": nan_one}
    if embedded:
        del reg, d_map, d_scans
        torch.cuda.empty_cache()
        return out
    print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


def loam_10m(args, local_rank):
    """BASELINE configs[3] at N = 1 -- pcr=loam, 65 536-pt scan vs a 10 M-point sub-map, 10 iterations, one MI355X, unsharded: the origin of
    config 4's strong-scaling curve, driver-timed.  The index build dominates there (roofline.kernel)."""
    import torch
    import oracle
    from simpleslam_amd import LoamRegister, synth
    n_map, n_scans = 10_000_000, 4
    world, map_np = synth.make_map(n_map, seed=SEED + 4)
    scans, inits = [], []
    for j in range(n_scans):
        sc, T = synth.make_scan(world, j, seed=SEED + 4)
        scans.append(sc); inits.append(synth.perturb(T, SEED + 4 + j))
    dev = torch.device("cuda", local_rank)
    d_map = torch.from_numpy(map_np).to(dev)
    d_scans = [torch.from_numpy(sc).to(dev) for sc in scans]
    reg = LoamRegister(device=local_rank, loam_iters=args.iters, loam_early_exit=0)
    reg.set_profile(0)

    def step(i):
        pose = inits[i % n_scans].copy(); reg.scan2Map(d_scans[i % n_scans], d_map, pose)
        return pose
    for i in range(6):
        step(i)
    steps = max(10, args.extra_steps // 2)
    wins = timed_windows(step, lambda: torch.cuda.synchronize(), steps, 3)
    elapsed = float(np.median(wins))
    reg.set_profile(2)
    k_ms, k_n, idx_ms = 0.0, 0, 0.0
    for i in range(8):
        step(i); st = reg.stats(); k_ms += st["kernel_ms"]; k_n += st["kernel_launches"]; idx_ms += st["index_ms"] / 8
    reg.set_profile(0)
    ach = 32.0 * n_map / (idx_ms * 1e-3) / 1e9
    out = {"metric": "scans/s (pcr=loam, 65 536-pt scan vs 10 M-pt submap, 10 GN iters, BASELINE configs[3] at N = 1)", "value": steps / elapsed, "unit": "scans/s",
           "n_gpus": 1, "steps": steps, "ms_per_step": 1e3 * elapsed / steps, "windows_ms": [1e3 * w for w in wins], "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"pcr=loam, {N_SCAN}-pt scan vs {n_map}-pt submap, {args.iters} GN iters, early exit off, index rebuilt per call "
                                  "(previous call's box and tile layout as checked hints), inputs in HBM, one GPU, unsharded", "scans_cycled": n_scans},
           "roofline": {"bound": "hbm", "kernel": "index build (grid_bin_kernel + grid_tile_kernel: the dominant cost at this map size)",
                        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_build": 32.0 * n_map,
                        "index_build_us": 1e3 * idx_ms, "iterate_avg_launch_us": 1e3 * k_ms / max(1, k_n), "traffic": None}}
    if not args.no_cpu_baseline:
        cores = host_cores()
        prm = oracle.loam_params(iters=args.iters, early_exit=0, threads=cores)
        t_cpu, et, er, n_done = 0.0, [], [], 0
        while n_done < 2 or (t_cpu < args.cpu_budget_s and n_done < n_scans):
            c0 = time.perf_counter(); ref, _, _ = oracle.loam_scan2map(scans[n_done], map_np, inits[n_done], prm); t_cpu += time.perf_counter() - c0
            dt, dr = synth.pose_error(step(n_done), ref)
            et.append(dt); er.append(dr); n_done += 1
        out["cpu_baseline"] = {"value": n_done / t_cpu, "unit": "scans/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
                               "sample": f"{n_done} of the same scans through the oracle (kd-tree of the 10 M-point map rebuilt per call), {t_cpu:.1f} s wall, OpenMP threads = {cores}"}
        out["pose_rmse_vs_cpu"] = {"trans_m": float(np.sqrt(np.mean(np.square(et)))), "rot_rad": float(np.sqrt(np.mean(np.square(er)))),
                                   "max_trans_m": float(max(et)), "max_rot_rad": float(max(er)), "scans": len(et), "tolerance": "1e-4 m / 1e-4 rad"}
    del reg, d_map, d_scans
    torch.cuda.empty_cache()
    return out


def sequence_leg(args, local_rank):
    """The caller's workload (reference frontend/src/LidarOdometry.cpp:160-200, frontend/src/MapManager.cpp:109-201): a drive of
    `--sequence-scans` poses, 0.5 m apart, through the box world.  Per scan: pcr_voxel_filter (0.5 m) -> pcr_scan2map_submap against the sub-map the
    key frames so far make (init = the previous result o the commanded motion, the reference's own defaults: LOAM 8 iterations with early exit, NDT and
    VGICP as configured) -> a key frame when none lies within 1 m -> the sub-map assembled again (pcr_map_update: radius 8 m, leaf 0.5 m) once the pose has
    moved 1 m.  Timed: the whole loop, scans already in HBM; one line per method, the CPU oracle driven through the SAME loop beside it."""
    import torch
    import oracle
    from simpleslam_amd import make_register, sequence, synth
    n = args.sequence_scans
    scans, truth, cmds = sequence.make_drive(n, SEED + 7)
    d_scans = [torch.from_numpy(sc).cuda() for sc in scans]
    cores = host_cores()
    res = {"workload": f"drive of {n} scans (64 beams x 1024, 0.5 m apart) through the box world; per scan: voxel filter 0.5 m -> scan2map against the sub-map of "
                       "the key frames within 8 m (voxel-filtered at 0.5 m, kept in HBM: pcr_map_*, pcr_scan2map_submap) with init = previous result o commanded "
                       "motion (odometry error <= 2 cm / 0.2 deg); key frame every 1 m, sub-map assembled again every 1 m; reference defaults of every method",
           "scans": n, "unit": "scans/s"}
    for mth in ("loam", "ndt", "vgicp"):
        try:
            # timed pass
            reg = make_register(mth, device=local_rank)
            reg.set_profile(0)
            sequence.drive(sequence.GpuFront(reg), d_scans[:8], cmds[:8], truth[0])       # warm-up: allocations, first builds
            del reg
            reg = make_register(mth, device=local_rank)
            reg.set_profile(0)
            front = sequence.GpuFront(reg)
            torch.cuda.synchronize()
            # a session's FIRST drive: the key-frame store, the concatenation and the voxel filter's index grow to size as it goes (hipMalloc / copy / hipFree,
            # each a device-wide stop) -- reported beside the line as `first_drive`; then the same drive again on the same handles, key frames forgotten
            # (pcr_map_clear), memory kept: the steady state a caller that runs for thousands of scans is in.  The poses of the two are compared.
            r_first = sequence.drive(front, d_scans, cmds, truth[0])
            passes = []
            for _w in range(3):      # (a drive is ~15 ms: three passes, the median one is the line's; all three are listed)
                front.reset()
                torch.cuda.synchronize()
                passes.append(sequence.drive(front, d_scans, cmds, truth[0]))
            r = sorted(passes, key=lambda x: x["seconds"])[1]
            same_as_first = all(np.array_equal(a, b) for q in passes for a, b in zip(q["poses"], r_first["poses"]))
            # untimed pass for what the timed one must not pay for: which hints held, the neighbour cache's hit rate (LOAM trace)
            kw = dict(record_trace=1) if mth == "loam" else {}
            reg2 = make_register(mth, device=local_rank, **kw)
            diag = {"builds": 0, "box_hint": 0, "layout_hint": 0, "region_index": 0, "hits": 0, "searches": 0}

            class Diag(sequence.GpuFront):
                def scan2map(self, ds, pose):
                    b0 = self.reg.stats()["target_builds"]
                    out = super().scan2map(ds, pose)
                    st = self.reg.stats()
                    if st["target_builds"] != b0:
                        diag["builds"] += 1; diag["box_hint"] += st["index_box_hint"]; diag["layout_hint"] += st["index_layout_hint"]
                    diag["region_index"] += st["region_index"]
                    if mth == "loam":
                        tr = self.reg.trace()
                        diag["hits"] += int(sum(tr["cache_hits"])); diag["searches"] += int(sum(tr["searches"]))
                    return out
            dfront = Diag(reg2, record=True)
            r2 = sequence.drive(dfront, d_scans, cmds, truth[0])
            same = all(np.array_equal(a, b) for a, b in zip(r["poses"], r2["poses"]))
            st = reg2.stats()
            # the CPU oracle through the same loop (bounded: as many scans of the drive as fit the budget), its poses against the GPU's
            prm = {"loam": lambda: oracle.loam_params(threads=cores), "ndt": lambda: oracle.ndt_params(threads=cores), "vgicp": lambda: oracle.vgicp_params(threads=cores)}[mth]()
            n_cpu = n
            t0 = time.perf_counter()
            rc = sequence.drive(oracle.SequenceFront(mth, prm), scans[:8], cmds[:8], truth[0])
            per = (time.perf_counter() - t0) / 8
            n_cpu = int(max(8, min(n, 3.0 * args.cpu_budget_s / max(per, 1e-6))))
            if n_cpu > 8:
                rc = sequence.drive(oracle.SequenceFront(mth, prm), scans[:n_cpu], cmds[:n_cpu], truth[0])
            # parity: calls of the drive repeated by the oracle on exactly what the HIP path was given (filtered scan, sub-map from HBM, initial pose)
            fn = {"loam": oracle.loam_scan2map, "ndt": oracle.ndt_scan2map, "vgicp": oracle.vgicp_scan2map}[mth]
            et, er = [], []
            t0 = time.perf_counter()
            for (ds, sub, init, out_pose, _c) in dfront.calls[::max(1, len(dfront.calls) // 16)]:
                dt, dr = synth.pose_error(out_pose, fn(ds, sub, init, prm)[0])
                et.append(dt); er.append(dr)
                if time.perf_counter() - t0 > 2.0 * args.cpu_budget_s and len(et) >= 4:
                    break
            drift = max(synth.pose_error(a, b)[0] for a, b in zip(r["poses"][:len(rc["poses"])], rc["poses"]))
            gt = [synth.pose_error(a, b)[0] for a, b in zip(r["poses"], truth)]
            line = {"value": n / r["seconds"], "ms_per_scan": 1e3 * r["seconds"] / n, "scan2map_ms_per_scan": 1e3 * r["scan2map_seconds"] / max(1, n - 1),
                    "ms_per_scan_by_step": {k: round(1e3 * v / n, 4) for k, v in r["step_seconds"].items()},
                    "timed": "the drive again over the same handles (key frames forgotten by pcr_map_clear, device memory kept): steady state; the median of three passes",
                    "passes_scans_per_s": [round(n / q["seconds"], 1) for q in passes],
                    "first_drive": {"value": n / r_first["seconds"], "ms_per_scan_by_step": {k: round(1e3 * v / n, 4) for k, v in r_first["step_seconds"].items()},
                                    "same_poses_as_timed_drive": bool(same_as_first),
                                    "note": "the same drive on fresh handles: the store, the concatenation and the filter's index grow as it goes (allocations, copies, frees)"},
                    "mean_iterations": float(np.mean(r["iterations"][1:])), "converged": int(sum(r["converged"])), "keyframes": r["keyframes"], "submap_assemblies": r["updates"],
                    "submap_points_last": int(r["submap_points"][-1]), "target_builds": int(st["target_builds"]),
                    "hints": {"index_builds": diag["builds"], "box_hint_held": diag["box_hint"], "tile_layout_held": diag["layout_hint"],
                              "region_only_index_calls": diag["region_index"], "region_repeats": int(st["region_repeats"]),
                              "note": "of the index builds of the drive (one per sub-map generation: the index is kept while the sub-map does not change), how many reused the previous "
                                      "target's bounding box / tile layout; kept targets (pcr_scan2map_submap) are always prepared in full, so region-only preparation does not apply"},
                    "repeatable": bool(same), "max_error_vs_truth_m": float(max(gt)),
                    "pose_rmse_vs_cpu": {"trans_m": float(np.sqrt(np.mean(np.square(et)))), "rot_rad": float(np.sqrt(np.mean(np.square(er)))), "max_trans_m": float(max(et)),
                                         "max_rot_rad": float(max(er)), "scans": len(et), "tolerance": "1e-4 m / 1e-4 rad",
                                         "how": "calls of the drive repeated by the oracle on the inputs the HIP path was given (device-filtered scan, sub-map from HBM, initial pose)"},
                    "max_distance_between_gpu_and_cpu_drives_m": float(drift),
                    "cpu_baseline": {"value": len(rc["poses"]) / rc["seconds"], "unit": "scans/s", "cores": cores, "kind": "port",
                                     "sample": f"the first {len(rc['poses'])} scans of the same drive through the same loop (oracle voxel filter, sub-map assembly and scan2map), "
                                               f"{rc['seconds']:.1f} s wall, OpenMP threads = {cores}"}}
            if mth == "loam" and diag["hits"] + diag["searches"]:
                line["neighbour_cache_hit_rate"] = diag["hits"] / (diag["hits"] + diag["searches"])
            res[mth] = line
            del reg, reg2
        except Exception as e:      # noqa: BLE001 -- the headline line must survive a failure here
            res[mth] = {"error": repr(e)}
    return res


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    if args.dry_run:
        return dry_run(args)
    if args.method != "loam":
        return secondary(args)
    import torch
    import torch.distributed as dist
    from simpleslam_amd import LoamRegister, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # PCR_BENCH_REHEARSE=1: several ranks on ONE card (gloo for the barrier and the max over ranks) -- only to exercise the
    # launch path of the multi-GPU run on a one-GPU box; the line it prints is not a measurement of N GPUs.
    rehearse = os.environ.get("PCR_BENCH_REHEARSE") == "1" and world_size > 1
    dev, rccl_ranks = join_ranks(torch, dist, rank, local_rank, world_size, rehearse)
    local_rank = dev.index

    # ---- synthetic workload (same map on every rank; scans differ per rank) ----
    world, map_np = synth.make_map(args.map_points, seed=SEED + 2)
    scans, inits, truths = [], [], []
    for j in range(args.scans):
        k = rank * args.scans + j
        s, T = synth.make_scan(world, k, seed=SEED + 2)
        assert s.shape[0] == N_SCAN
        scans.append(s)
        truths.append(T)
        inits.append(synth.perturb(T, SEED + 2 + k))

    reg = LoamRegister(device=local_rank, loam_iters=args.iters, loam_early_exit=0)
    d_map_full = torch.from_numpy(map_np).to(dev)
    d_scans = [torch.from_numpy(s).to(dev) for s in scans]
    d_map = d_map_full
    scaling = "weak"
    parallelism = (f"replicas x{world_size}: THE scaling mode of this path -- one process per GPU, every rank the whole map and its own scans, no data-path collective "
                   "(weak scaling); --shard-map is the other mode, one call's map cut over the ranks, and is latency-bound (DESIGN.md 5)") if world_size > 1 else "single GPU"
    rank_info = {"rank": rank, "device": local_rank, "map_points": int(map_np.shape[0])}
    if args.shard_map:
        from simpleslam_amd import shard
        tile = shard.tile_for_method(map_np, rank, world_size, "loam")
        d_map = torch.from_numpy(tile.points).to(dev)
        reg.set_shard(tile.lo, tile.hi, tile.halo)
        if args.transport == "peer":      # (works in a rehearsal too: two processes may map each other's buffers on one card)
            handles = [None] * world_size
            if world_size > 1:
                dist.all_gather_object(handles, reg.comm_peer_export())
            else:
                handles = [reg.comm_peer_export()]
            reg.comm_init_peer(handles, rank, world_size)
        elif rehearse:      # all ranks on one card: RCCL refuses a communicator with one device twice -> the exchange goes through gloo
            reg.comm_init_host(shard.gloo_collective(), rank, world_size)
        else:
            uid = [shard.unique_id() if rank == 0 else None]
            if world_size > 1:
                dist.broadcast_object_list(uid, src=0)
            reg.comm_init(uid[0], rank, world_size)
        # the scans are the SAME on every rank (one scan is split over the tiles), so rank 0's set
        scans = [synth.make_scan(world, j, seed=SEED + 2)[0] for j in range(args.scans)]
        d_scans = [torch.from_numpy(s).to(dev) for s in scans]
        inits = [synth.perturb(synth.scan_pose(world, j, SEED + 2), SEED + 2 + j) for j in range(args.scans)]
        scaling = "strong"
        ci = reg.comm_info()
        rccl_ranks = ci["nranks"] if ci["transport"] == "rccl" else 0      # read back from the communicator the handle reduces over
        parallelism = f"map tiles x{world_size} + all-reduce of JtJ/JtE per iteration over {ci['transport']}"
        rank_info.update(tile_points=int(tile.points.shape[0]), tile_core_points=int(tile.n_core), tile_axis=int(tile.axis),
                         tile_lo=float(tile.lo[tile.axis]), tile_hi=float(tile.hi[tile.axis]), transport=ci["transport"])

    def step(i):
        pose = inits[i % args.scans].copy()
        reg.scan2Map(d_scans[i % args.scans], d_map, pose)
        return pose

    reg.set_profile(0)
    for i in range(args.warmup):
        step(i)

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # EXACTLY args.steps steps per timed window, bracketed by barrier + synchronize; several windows back to back, the MAX over
    # ranks of each, and the MEDIAN window is the one reported (a 200-step window is 50 ms: one stray host hiccup moves it by percent)
    wins = timed_windows(step, barrier, args.steps, max(1, args.windows))
    if world_size > 1:
        t = torch.tensor(wins, dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wins = [float(v) for v in t.tolist()]
    elapsed = float(np.median(wins))
    units = args.steps * (1 if args.shard_map else world_size)
    value = units / elapsed

    out = {
        "metric": "scans/s (65 k-pt scan vs 1 M-pt submap, LOAM 10 iters) + pose RMSE vs CPU ref",
        "value": value, "unit": "scans/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "windows_ms": [1e3 * v for v in wins], "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic" + (" (REHEARSAL: all ranks on one card)" if rehearse else ""),
        "config": {"workload": f"pcr=loam, {N_SCAN}-pt 64-beam scan vs {args.map_points}-pt submap, {args.iters} GN iters, "
                               "early exit off, index rebuilt per call (previous call's bounding box and tile layout reused as checked hints), inputs in HBM",
                   "parallelism": parallelism, "scans_cycled": args.scans},
        # how many ranks REALLY ran: n_gpus is WORLD_SIZE of the process group, rccl_ranks the size of the RCCL group that carried a
        # collective (replicas: the barrier / max-over-ranks group; --shard-map: the handle's own communicator); gpus_arg is argv
        "rccl_ranks": rccl_ranks, "gpus_arg": args.gpus, "launcher": launcher_name(world_size),
    }

    if rank == 0 and args.streams > 1 and not args.shard_map:
        # ---- S independent registrations in flight (multi-robot / loop-closure candidates): handles are independent
        #      objects with their own stream ("distinct handles are concurrent-safe") ----
        import threading
        from simpleslam_amd import pcr as _pcr
        prm = _pcr.default_params(device=local_rank, loam_iters=args.iters, loam_early_exit=0)
        prm.loam_coresident = 1          # the two-waves-per-SIMD variant of the iterate kernel: blocks of different handles share the CUs
        regs = [LoamRegister(params=prm) for _ in range(args.streams)]
        per = max(1, args.steps // args.streams)
        for r in regs:
            for i in range(3):
                p = inits[i % args.scans].copy(); r.scan2Map(d_scans[i % args.scans], d_map, p)
        start = threading.Barrier(args.streams + 1)

        def worker(r, off):
            start.wait()
            for i in range(per):
                p = inits[(i + off) % args.scans].copy()
                r.scan2Map(d_scans[(i + off) % args.scans], d_map, p)

        th = [threading.Thread(target=worker, args=(r, j)) for j, r in enumerate(regs)]
        for t in th: t.start()
        torch.cuda.synchronize()
        start.wait(); c0 = time.perf_counter()
        for t in th: t.join()
        torch.cuda.synchronize()
        ct = time.perf_counter() - c0
        out["concurrent"] = {"streams": args.streams, "value": per * args.streams / ct, "unit": "scans/s",
                             "scans": per * args.streams, "note": "independent handles (pcr_params.loam_coresident = 1), one stream and one host thread each, same GPU"}
    # ---- roofline of the dominant kernel (loam_iterate_kernel), live, HIP events on its stream.  Rank 0 reports it; with a sharded
    #      map every rank has to take part in the calls (each one is a chain of collectives) ----
    k_ms, k_n, idx_ms, tot_ms = 0.0, 0, 0.0, 0.0
    reps = 16
    reg.set_profile(2)
    for i in range(reps):
        step(i)
        st = reg.stats()
        k_ms += st["kernel_ms"]; k_n += st["kernel_launches"]; idx_ms += st["index_ms"]; tot_ms += st["total_ms"]
    reg.set_profile(0)
    avg_s = (k_ms / max(1, k_n)) * 1e-3
    alg_bytes = 96 * N_SCAN + 216          # per linearisation launch (sharded: the scan is split, the launch is not shorter)
    achieved = alg_bytes / avg_s / 1e9
    rank_info.update(avg_launch_us=avg_s * 1e6, hbm_frac=achieved / HBM_PEAK_GBS, index_build_us=1e3 * idx_ms / reps,
                     index_build_GBs=32 * d_map.shape[0] / (idx_ms / reps * 1e-3) / 1e9 if idx_ms > 0 else None)
    out["ranks"] = gather_ranks(dist, world_size, rank_info)
    if args.shard_map and world_size > 1:
        # the origin of the curve: the SAME scans against the WHOLE map on one GPU (rank 0, an unsharded handle), while the others wait
        if rank == 0:
            reg1 = LoamRegister(device=local_rank, loam_iters=args.iters, loam_early_exit=0)
            reg1.set_profile(0)

            def step1(i):
                pose = inits[i % args.scans].copy()
                reg1.scan2Map(d_scans[i % args.scans], d_map_full, pose)
            for i in range(min(args.warmup, 10)):
                step1(i)
            w1 = timed_windows(step1, lambda: torch.cuda.synchronize(), args.steps, 3)
            e1 = float(np.median(w1))
            out["n1"] = {"value": args.steps / e1, "unit": "scans/s", "ms_per_step": 1e3 * e1 / args.steps,
                         "note": "same scans, whole map, one GPU, unsharded handle (the N = 1 point of this strong-scaling curve)"}
            del reg1
        dist.barrier()
    if rank == 0:
        # (HBM traffic is NOT measured in this run -- counters need passes of their own under rocprofv3: it is read from the summary the newest committed
        #  profile round left, and the line says which)
        traffic, traffic_src = None, None
        for name in ("r05_loam_iterate_pmc.json", "loam_iterate_pmc.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch"); traffic_src = f"profiles/{name} (FETCH_SIZE / WRITE_SIZE passes of the headline command); not measured in this run"
                    break
                except Exception:
                    traffic = None
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                           "kernel": "loam_iterate_kernel", "avg_launch_us": avg_s * 1e6,
                           "algorithmic_bytes_per_launch": alg_bytes,
                           "index_build_us": 1e3 * idx_ms / reps,
                           "index_build_GBs": 32 * d_map.shape[0] / (idx_ms / reps * 1e-3) / 1e9 if idx_ms > 0 else None,
                           # (this pass runs with an event pair around every launch -- pcr_set_profile(2) -- which stretches the call:
                           #  it is NOT comparable with ms_per_step, which is timed without any event)
                           "device_ms_per_scan_profiled": tot_ms / reps}

        # ---- the same scans against a target index that is kept while the sub-map does not change (pcr_set_target + pcr_align: the
        #      path pcr_scan2map_submap takes for every scan after the first of a pcr_map generation).  Reported separately: `value`
        #      stays the reference's rebuild-per-call semantics. ----
        if not args.shard_map:
            reg.setTarget(d_map)
            for i in range(10):
                p = inits[i % args.scans].copy(); reg.align(d_scans[i % args.scans], p)
            n_keep = max(20, args.steps // 2)

            def kept(i):
                p = inits[i % args.scans].copy(); reg.align(d_scans[i % args.scans], p)
            kw = timed_windows(kept, lambda: torch.cuda.synchronize(), n_keep, 3)
            dt = sorted(kw)[len(kw) // 2] / n_keep
            out["index_kept"] = {"value": 1.0 / dt, "unit": "scans/s", "ms_per_step": dt * 1e3, "scans": n_keep, "windows_ms": [w * 1e3 for w in kw],
                                 "note": "target index built once per sub-map generation (pcr_scan2map_submap / pcr_set_target + pcr_align), not per call; never reported as value"}
            reg.invalidateTarget()

        # ---- what the headline carries, said in the line ----
        # (a) index_hints: `value` is measured with the target index rebuilt on every call, but the build reuses the bounding box and the
        #     tile layout of the PREVIOUS call's target as hints (checked on the device; a sub-map changes by a key frame at a time).  The
        #     same scans with pcr_params.index_no_hints = 1 -- fresh box, no layout, no state across calls -- are timed beside it.
        # (b) host_buffers: the drop-in adapter hands HOST clouds to pcr_scan2map; that path pays the PCIe copy of the
        #     map on every call.  Timed from pageable memory and from a range the caller page-locked with pcr_host_pin.  Never `value`.
        if not args.shard_map and not args.no_twin:
            from simpleslam_amd import pcr as _pcr
            out["index_hints"] = True
            reg_nh = LoamRegister(device=local_rank, loam_iters=args.iters, loam_early_exit=0, index_no_hints=1)
            reg_nh.set_profile(0)

            def step_nh(i):
                pose = inits[i % args.scans].copy(); reg_nh.scan2Map(d_scans[i % args.scans], d_map, pose)
            for i in range(10):
                step_nh(i)
            n_nh = max(20, args.steps // 2)
            w_nh = timed_windows(step_nh, lambda: torch.cuda.synchronize(), n_nh, 3)
            e_nh = float(np.median(w_nh)) / n_nh
            reg_nh.set_profile(1)
            idx_nh = 0.0
            for i in range(8):
                step_nh(i); idx_nh += reg_nh.stats()["index_ms"] / 8
            out["no_index_hints"] = {"value": 1.0 / e_nh, "unit": "scans/s", "ms_per_step": 1e3 * e_nh, "index_build_us": 1e3 * idx_nh,
                                     "note": "pcr_params.index_no_hints = 1: bounding box pass + three build launches on every call, nothing carried across calls"}
            del reg_nh
            host = {}
            n_hb = max(10, args.steps // 4)
            map_pinned = np.array(map_np, copy=True)
            scans_host = [np.ascontiguousarray(s) for s in scans]
            for name, m_host in (("pageable", map_np), ("pinned", map_pinned)):
                if name == "pinned":
                    _pcr.host_pin(map_pinned)
                    for s in scans_host:
                        _pcr.host_pin(s)

                def step_h(i):
                    pose = inits[i % args.scans].copy(); reg.scan2Map(scans_host[i % args.scans], m_host, pose)
                    return pose
                for i in range(5):
                    step_h(i)
                w_h = timed_windows(step_h, lambda: torch.cuda.synchronize(), n_hb, 3)
                e_h = float(np.median(w_h)) / n_hb
                host[name] = {"value": 1.0 / e_h, "unit": "scans/s", "ms_per_step": 1e3 * e_h,
                              "bytes_uploaded_per_scan": int(m_host.nbytes + scans_host[0].nbytes)}
            _pcr.host_unpin(map_pinned)
            for s in scans_host:
                _pcr.host_unpin(s)
            host["note"] = ("pcr_scan2map with HOST clouds (16-byte points): map + scan cross PCIe on every call; pinned = ranges page-locked by the "
                            "caller with pcr_host_pin.  PCIe-bound: bytes / ~55 GB/s is the floor under the copy")
            # pcl32: exactly what INTEGRATION.md's adapter hands over -- 32-byte pcl::PointXYZI records (x y z 1 | intensity pad pad pad,
            # common/types/basic.hpp:16), uploaded verbatim (34 MB per call); `_xyz_only` = pcr_params.host_copy_xyz = 1, a pitched copy of the
            # first 16 bytes of every record (half the bytes, and several times slower on this runtime: measured, not the default).
            def xyzi32(a):
                o = np.zeros((a.shape[0], 8), np.float32)
                o[:, :3] = a[:, :3]; o[:, 3] = 1.0; o[:, 4] = a[:, 3]
                return o
            map32 = xyzi32(map_np)
            scans32 = [xyzi32(sc) for sc in scans]
            for name, pin, whole in (("pcl32", False, 0), ("pcl32_pinned", True, 0), ("pcl32_xyz_only", False, 1), ("pcl32_xyz_only_pinned", True, 1)):
                reg_h = LoamRegister(device=local_rank, loam_iters=args.iters, loam_early_exit=0, host_copy_xyz=whole)
                reg_h.set_profile(0)
                if pin:
                    _pcr.host_pin(map32)
                    for sc in scans32:
                        _pcr.host_pin(sc)

                def step_p(i):
                    pose = inits[i % args.scans].copy(); reg_h.scan2Map(scans32[i % args.scans], map32, pose)
                    return pose
                for i in range(5):
                    step_p(i)
                w_h = timed_windows(step_p, lambda: torch.cuda.synchronize(), n_hb, 3)
                e_h = float(np.median(w_h)) / n_hb
                p_dev, p_host = step(0), step_p(0)
                host[name] = {"value": 1.0 / e_h, "unit": "scans/s", "ms_per_step": 1e3 * e_h,
                              "bytes_crossing_pcie_per_scan": int((map32.shape[0] + scans32[0].shape[0]) * (16 if whole else 32)),
                              "same_pose_as_device_buffers": bool(np.array_equal(p_dev, p_host))}
                if pin:
                    _pcr.host_unpin(map32)
                    for sc in scans32:
                        _pcr.host_unpin(sc)
                del reg_h
            out["host_buffers"] = host
            # loc_static: test/loc.cpp's flow (map loaded once from a PCD, MapManager.cpp:52-78; every scan localised against it):
            # pcr_set_target once, pcr_align per scan with a HOST scan of 32-byte records -- only the scan crosses PCIe
            reg_s = LoamRegister(device=local_rank, loam_iters=args.iters, loam_early_exit=0)
            reg_s.set_profile(0)
            reg_s.setTarget(map32)

            def step_s(i):
                pose = inits[i % args.scans].copy(); reg_s.align(scans32[i % args.scans], pose)
                return pose
            for i in range(10):
                step_s(i)
            n_ls = max(20, args.steps // 2)
            w_s = timed_windows(step_s, lambda: torch.cuda.synchronize(), n_ls, 3)
            e_s = float(np.median(w_s)) / n_ls
            out["loc_static"] = {"value": 1.0 / e_s, "unit": "scans/s", "ms_per_step": 1e3 * e_s, "scans": n_ls,
                                 "same_pose_as_scan2map": bool(np.array_equal(step_s(0), step(0))),
                                 "note": "pcr_set_target(map) once + pcr_align(host scan, 32-byte records) per scan: the static-map localisation of test/loc.cpp; "
                                         "only the scan (2 MB) crosses PCIe per call; never reported as value"}
            del reg_s, map32, scans32

        # ---- pose parity + CPU baseline: the oracle (a port of the reference's loop, rebuilt kd-tree per call) on this host's
        #      cores, at 1 thread, at the reference's default `cores` = 4 (config/params.json:5) and at all cores, plus the same
        #      with the REFERENCE's own vendored nanoflann as the index when oracle/_ref was built ----
        if not args.no_cpu_baseline and not args.shard_map and world_size == 1:      # the CPU leg is timed at N = 1 only
            import oracle
            cores = host_cores()

            def cpu_leg(threads, budget, parity=False):
                prm = oracle.loam_params(iters=args.iters, early_exit=0, threads=threads)
                n_done, t_cpu, et, er = 0, 0.0, [], []
                while n_done < 4 * args.scans and (n_done < (args.scans if parity else 2) or t_cpu < budget):
                    j = n_done % args.scans
                    c0 = time.perf_counter()
                    ref, _, _ = oracle.loam_scan2map(scans[j], map_np, inits[j], prm)
                    t_cpu += time.perf_counter() - c0
                    if parity and n_done < args.scans:          # pose parity on every distinct scan
                        dt, dr = synth.pose_error(step(j), ref)
                        et.append(dt); er.append(dr)
                    n_done += 1
                return n_done, t_cpu, et, er

            by_cores = {}
            for th in sorted({1, min(4, cores), cores}):
                if th == cores:
                    continue
                n_done, t_cpu, _, _ = cpu_leg(th, args.cpu_budget_s)
                by_cores[str(th)] = {"value": n_done / t_cpu, "scans": n_done, "wall_s": t_cpu}
            n_done, t_cpu, et, er = cpu_leg(cores, args.cpu_budget_s, parity=True)
            by_cores[str(cores)] = {"value": n_done / t_cpu, "scans": n_done, "wall_s": t_cpu}
            out["cpu_baseline"] = {"value": n_done / t_cpu, "unit": "scans/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
                                   "index": "the port's own kd-tree (leaf 10, rebuilt per call)", "build": "gcc -O3 -fopenmp",
                                   "sample": f"{n_done} of the same scans (kd-tree rebuilt per call + {args.iters} iterations), "
                                             f"{t_cpu:.1f} s wall, OpenMP threads = {cores}",
                                   "by_cores": by_cores}
            if oracle.ref_available():
                try:
                    oracle.use_reference_nanoflann(True)
                    n2, t2, _, _ = cpu_leg(cores, args.cpu_budget_s)
                    out["cpu_baseline"]["nanoflann_ref"] = {"value": n2 / t2, "scans": n2, "wall_s": t2, "cores": cores,
                                                            "index": "nanoflann(_ref): the reference's vendored nanoflann.hpp, built per call (serial) + exact 5-NN"}
                finally:
                    oracle.use_reference_nanoflann(False)
            out["pose_rmse_vs_cpu"] = {"trans_m": float(np.sqrt(np.mean(np.square(et)))), "rot_rad": float(np.sqrt(np.mean(np.square(er)))),
                                       "max_trans_m": float(max(et)), "max_rot_rad": float(max(er)), "scans": len(et),
                                       "tolerance": "1e-4 m / 1e-4 rad"}
        # ---- BASELINE configs[2] and configs[4] on the same card, embedded so that the driver's line carries them ----
        if world_size == 1 and not args.shard_map and not args.no_extra:
            del d_map_full, d_scans, d_map
            out["extra"] = {}
            for mth in ("vgicp", "ndt"):
                try:
                    out["extra"][mth] = secondary(args, method=mth, steps=args.extra_steps, warmup=5, embedded=True)
                except Exception as e:      # noqa: BLE001 -- the headline line must survive a failure of a secondary one
                    out["extra"][mth] = {"error": repr(e)}
            try:
                out["extra"]["loam_10m"] = loam_10m(args, local_rank)
            except Exception as e:      # noqa: BLE001
                out["extra"]["loam_10m"] = {"error": repr(e)}
            if args.sequence_scans > 0:
                try:
                    out["extra"]["sequence"] = sequence_leg(args, local_rank)
                except Exception as e:      # noqa: BLE001
                    out["extra"]["sequence"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
