"""Multi-GPU scaling of the hot path: independent tracking instances sharded over ranks.

One bundle-adjustment problem is tiny (<= 75 MB input, <= 174 unknowns) and its PCG is
latency-bound, so it is never split.  Instances share nothing: instance n runs on
rank n mod G, every rank keeps its instances resident in its own HBM and calls btba_solve_batch;
the ONLY collective is an all-gather of {seconds, gn_iterations} per rank after the timed region
(RCCL over xGMI on GPUs via backend "nccl"; gloo in the CPU tests).  The reference has no
distributed code at all ("Parallelism strategies").
"""
from __future__ import annotations

import os


def instances_for_rank(n_instances: int, rank: int, world_size: int) -> list[int]:
    """Instance n -> rank n mod G."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return list(range(rank, n_instances, world_size))


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).
    Returns (rank, world_size, local_rank); world_size 1 needs no process group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # a process group also at world size 1 when launched by torchrun (or BTBA_DIST_FORCE=1): the same RCCL calls -- communicator
    # set-up, device barrier, device all-gather -- then run on a single-GPU box
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("BTBA_DIST_FORCE"):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            if backend is None:
                backend = os.environ.get("BTBA_DIST_BACKEND", "") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier(device=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if device is not None and str(device).startswith("cuda") and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[int(str(device).split(":")[1])] if ":" in str(device) else None)
        else:
            dist.barrier()


def gather_throughput(seconds: float, gn_iters: float, device="cpu", checksum: float = 0.0):
    """All-gather {seconds, gn_iters, pose checksum} of every rank (24 bytes per rank -- latency only; the checksum is SURVEY.md 8(e)'s
    optional consistency check of the sharded run).  Returns a list of (seconds, gn_iters) indexed by rank; the checksums are kept in
    gather_throughput.checksums."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(seconds), float(gn_iters), float(checksum)], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        gather_throughput.checksums = [float(mine[2])]
        return [(float(mine[0]), float(mine[1]))]
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    gather_throughput.checksums = [float(t[2]) for t in out]
    return [(float(t[0]), float(t[1])) for t in out]


gather_throughput.checksums = []


def backend_name():
    import torch.distributed as dist
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None


def aggregate(per_rank):
    """Whole-job throughput: total GN iterations / slowest rank's time."""
    total = sum(g for _, g in per_rank)
    slowest = max(s for s, _ in per_rank)
    return total / slowest if slowest > 0 else float("nan"), slowest


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (/sys/fs/cgroup/cpu.max, v2; cpu.cfs_quota_us, v1).
    os.cpu_count() reports the machine (256 on a gpurun box whose container is allowed 16): sizing worker pools from it oversubscribes."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            quota = None
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def bind_rank_to_cpus(local_rank: int, local_world: int, gpu_index: int | None = None):
    """Give every rank of a node its own CPUs, preferably on the NUMA node of its GPU: each rank is one launcher thread issuing ~10 k kernel
    launches per second, and eight of them sharing one CPU quota (or bouncing across sockets) would make the host the bottleneck of an
    N = 8 run that has none on the device.  Best effort -- returns a description of what was done (recorded in the bench line)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"bound": False, "why": "no sched_getaffinity"}
    if local_world <= 1 or len(allowed) < 2 * local_world:
        return {"bound": False, "cpus_allowed": len(allowed), "why": "single rank or too few CPUs to split"}
    numa_cpus = None
    numa_node = None
    if gpu_index is not None:
        try:                                        # the GPU's NUMA node, where the driver exposes it
            import glob
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/numa_node"), key=lambda p: int("".join(ch for ch in p.split("/")[4] if ch.isdigit())))
            cards = [c for c in cards if os.path.exists(os.path.join(os.path.dirname(c), "mem_info_vram_total"))]
            if gpu_index < len(cards):
                numa_node = int(open(cards[gpu_index]).read().strip())
                if numa_node >= 0:
                    cpus = set()
                    for part in open(f"/sys/devices/system/node/node{numa_node}/cpulist").read().strip().split(","):
                        lo, _, hi = part.partition("-")
                        cpus.update(range(int(lo), int(hi or lo) + 1))
                    numa_cpus = sorted(cpus & set(allowed))
        except Exception:
            numa_cpus = None
    per = len(allowed) // local_world
    mine = allowed[local_rank * per:(local_rank + 1) * per]
    if numa_cpus and len(numa_cpus) >= 2:
        # ranks whose GPUs share a NUMA node split that node's CPUs among themselves by local rank
        share = max(2, len(numa_cpus) // local_world)
        k = (local_rank * share) % max(1, len(numa_cpus) - share + 1)
        mine = numa_cpus[k:k + share]
    try:
        os.sched_setaffinity(0, mine)
    except OSError as e:
        return {"bound": False, "why": str(e)[:80]}
    return {"bound": True, "cpus": len(mine), "first_cpu": mine[0], "numa_node": numa_node}
