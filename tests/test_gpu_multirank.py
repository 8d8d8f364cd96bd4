"""bench.py's N > 1 path on the GPU box: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2`).  A gpurun box has one GPU, so
both ranks share it (bench.py maps LOCAL_RANK modulo the device count) and the closing all-gather runs on gloo
(BTBA_DIST_BACKEND) -- RCCL refuses two ranks on one device; on a real node the same code runs one rank per GPU on nccl.
Checks the contract: one JSON line from rank 0, whole-job value = all ranks' iterations / slowest rank's time."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_through_torchrun():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, BTBA_DIST_BACKEND="gloo", BTBA_BENCH_NPROC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--instances", "4", "--distinct", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                           # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert len(d["per_rank"]) == 2
    total = sum(r["gn_iters"] for r in d["per_rank"])
    slowest = max(r["seconds"] for r in d["per_rank"])
    assert total == 2 * 4 * 7 * 3                                    # ranks x instances x GN iterations x steps
    assert abs(d["value"] - total / slowest) <= 1e-3 * d["value"]
    assert d["config"]["instances_per_gpu"] == 4


def test_one_rank_on_rccl():
    """The RCCL branch itself, executed: one rank under torchrun with backend nccl (= RCCL on ROCm) -- communicator creation,
    the device barrier on both sides of the timed region and the closing device all-gather are the calls an 8-GPU run makes,
    only with one participant (a gpurun box has one GPU)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, BTBA_DIST_BACKEND="nccl", BTBA_BENCH_NPROC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--instances", "4", "--distinct", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["collective"]["backend"] == "nccl" and d["collective"]["rccl_ranks"] == 1
    assert d["n_gpus"] == 1 and len(d["per_rank"]) == 1 and d["per_rank"][0]["gn_iters"] == 4 * 7 * 3
    assert d["per_rank"][0]["pose_checksum"] > 0


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher: bench.py re-runs itself under torch.distributed.run (one rank per GPU, RCCL on a real
    node).  On a one-GPU box the two ranks share the device and the gather runs on gloo (BTBA_DIST_BACKEND); without that override the
    same command must refuse instead of printing an n_gpus: 1 line.  With --same-instances both ranks solve identical seeds: identical pose checksums."""
    env = dict(os.environ, BTBA_BENCH_NPROC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--instances", "4", "--distinct", "2", "--no-cpu-baseline",
           "--same-instances"]
    import torch
    if torch.cuda.device_count() < 2:
        refused = subprocess.run(cmd, env=dict(env, BTBA_DIST_BACKEND=""), cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert refused.returncode != 0 and "only 1 GPU" in (refused.stderr + refused.stdout) and not [l for l in refused.stdout.splitlines() if l.startswith("{")]
        env["BTBA_DIST_BACKEND"] = "gloo"
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["per_rank"]) == 2 and d["scaling"] == "weak"
    assert sum(r["gn_iters"] for r in d["per_rank"]) == 2 * 4 * 7 * 3
    # --same-instances: both ranks solved the same seeds -> the same poses, bit for bit
    assert d["per_rank"][0]["pose_checksum"] > 0 and d["per_rank"][0]["pose_checksum"] == d["per_rank"][1]["pose_checksum"]


def test_eight_ranks_first_run():
    """What the driver's 8-GPU box runs first, as far as a one-GPU box can show it: `bench.py --gpus 8` spawning EIGHT ranks (sharing the one device, gather
    on gloo), each with its own worker pool sized from the CPUs the container really has (sharding.usable_cpus), the same two instances on every rank:
    eight equal pose checksums, eight per-rank records, the whole-job value = all iterations / the slowest rank."""
    env = dict(os.environ, BTBA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BTBA_BENCH_NPROC"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--instances", "2", "--distinct", "2", "--no-cpu-baseline",
           "--no-tracker-call", "--same-instances", "--settle-ms", "0", "--baseline-n1", "1000"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and len(d["per_rank"]) == 8 and d["scaling"] == "weak"
    assert sum(r["gn_iters"] for r in d["per_rank"]) == 8 * 2 * 7 * 2
    sums = {r["pose_checksum"] for r in d["per_rank"]}
    assert len(sums) == 1 and next(iter(sums)) > 0, d["per_rank"]
    assert d["rank_spread"]["ms_per_step_min"] <= d["rank_spread"]["ms_per_step_max"]
    assert abs(d["efficiency_vs_n1"]["value"] - d["value"] / 8000.0) < 1e-3 * max(1.0, d["value"] / 8000.0)
    assert d["collective"]["backend"] == "gloo" and d["collective"]["rccl_ranks"] == 0          # (RCCL refuses eight ranks on one device; a real node reports 8)
