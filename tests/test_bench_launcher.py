"""bench.py as the driver starts it: `python bench.py --gpus N` must really run N ranks (missing #1).

No GPU here: `--dry-run` makes every rank report the environment it was started with (gathered over the same 127.0.0.1
rendezvous the measuring run uses) instead of measuring."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PCR_BENCH_LAUNCHER")}
    return env


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 4])
def test_gpus_n_starts_n_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = _line(r.stdout)
    assert d["dry_run"] and d["n_gpus"] == n and d["gpus_arg"] == n
    ranks = d["ranks"]
    assert [x["rank"] for x in ranks] == list(range(n))
    assert [x["local_rank"] for x in ranks] == list(range(n))
    assert len({x["pid"] for x in ranks}) == n and os.getpid() not in {x["pid"] for x in ranks}
    ports = {x["env"]["MASTER_PORT"] for x in ranks}
    assert len(ports) == 1 and int(ports.pop()) > 0
    for x in ranks:
        assert x["world_size"] == n and x["env"]["WORLD_SIZE"] == str(n)
        assert x["env"]["MASTER_ADDR"] == "127.0.0.1"
        assert x["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert x["launcher"] == "bench.py"


def test_shard_map_defaults_to_config4_map():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--shard-map", "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = _line(r.stdout)
    assert d["shard_map"] and d["map_points"] == 10_000_000 and d["n_gpus"] == 2


def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["ranks"][0]["launcher"] == "none" and d["map_points"] == 1_000_000


def test_under_torch_distributed_run_the_process_is_a_rank():
    """the driver's N > 1 command: torch.distributed.run starts the ranks, bench.py must not start more"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", BENCH, "--gpus", "2", "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and len(d["ranks"]) == 2
    assert all(x["launcher"].startswith("external") for x in d["ranks"])


def test_a_failing_rank_fails_the_launch():
    """rank 1 dies before the rendezvous: the launcher ends rank 0 (which would wait for it for ever), exits nonzero, prints no line"""
    env = _clean_env()
    env["PCR_BENCH_DRYRUN_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
@pytest.mark.parametrize("shard", [False, True])
def test_two_ranks_rehearsed_on_one_card(shard):
    """The rank path behind the launcher, on the one GPU a test box has: PCR_BENCH_REHEARSE=1 lets the two ranks share the card and
    exchange over gloo (RCCL refuses one device twice).  Not a measurement -- the line says so -- but everything else is what an
    8-GPU node would run: fresh rank processes, tiles + halo per rank, the all-reduce per linearisation, the per-rank report, the
    N = 1 origin of the sharded curve."""
    env = _clean_env()
    env["PCR_BENCH_REHEARSE"] = "1"
    cmd = [sys.executable, BENCH, "--gpus", "2", "--steps", "6", "--warmup", "2", "--windows", "1", "--map-points", "200000", "--scans", "2",
           "--no-cpu-baseline", "--no-extra"] + (["--shard-map"] if shard else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["gpus_arg"] == 2 and d["launcher"] == "bench.py" and d["rccl_ranks"] == 0      # gloo in a rehearsal
    assert "REHEARSAL" in d["data"] and d["value"] > 0
    assert [x["rank"] for x in d["ranks"]] == [0, 1]
    if shard:
        assert d["scaling"] == "strong" and d["n1"]["value"] > 0
        assert sum(x["tile_core_points"] for x in d["ranks"]) == 200000          # every map point belongs to exactly one tile
        assert all(x["tile_points"] >= x["tile_core_points"] and x["transport"] == "host" for x in d["ranks"])
        assert d["ranks"][0]["tile_hi"] == d["ranks"][1]["tile_lo"]
    else:
        assert d["scaling"] == "weak" and "n1" not in d


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["ndt", "vgicp"])
def test_two_ranks_rehearsed_sharded_ndt_and_vgicp(method):
    """`bench.py --method ndt|vgicp --shard-map --gpus 2` (BASELINE configs[4]'s "1 -> 8 GPU scaling curve", configs[2] sharded), rehearsed
    on one card: tiles on the method's voxel lattice with its halo, the 43 sums of every evaluation pass all-reduced (gloo here, RCCL
    between GPUs), the per-rank report and the N = 1 origin of the curve in the line."""
    env = _clean_env()
    env["PCR_BENCH_REHEARSE"] = "1"
    cmd = [sys.executable, BENCH, "--gpus", "2", "--method", method, "--shard-map", "--steps", "4", "--warmup", "2", "--windows", "1", "--scans", "2",
           "--secondary-map-points", "300000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 0 and d["scaling"] == "strong" and "REHEARSAL" in d["data"]
    assert d["value"] > 0 and d["n1"]["value"] > 0
    assert [x["rank"] for x in d["ranks"]] == [0, 1]
    assert sum(x["tile_core_points"] for x in d["ranks"]) == 300000
    assert all(x["tile_points"] > x["tile_core_points"] and x["transport"] == "host" and x["halo"] > 0 for x in d["ranks"])
    assert d["ranks"][0]["tile_hi"] == d["ranks"][1]["tile_lo"]
