"""Multi-GPU plumbing: AECM streams are independent, so the hot path shards by static contiguous
ranges with NO data-path collective.  torch.distributed (RCCL on GPUs, gloo in CPU tests) is used
only to line the ranks up for timing and to gather a handful of throughput counters."""
from __future__ import annotations

import os
import socket
import sys

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int):
    """Contiguous [first, first+count) of `total` streams owned by `rank` (remainder to low ranks)."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(script: str, argv, n_ranks: int):
    """`python script --gpus N` started by hand (no WORLD_SIZE in the environment): replace this process by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... script argv` -- one rank per GPU of
    this node, rendezvous on 127.0.0.1.  Does not return."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(sys.executable, cmd, env)


def init(backend: str, device_index=None, timeout_s: int = 300):
    """Form the process group (RCCL when backend == "nccl").  `device_index`: the HIP device this rank already selected --
    handed to init_process_group so the communicator is bound to it eagerly and barriers need no device guess."""
    import datetime
    rank, local_rank, world = env_rank_world()
    # AECM_FORCE_DIST=1 initialises the process group even for one rank (exercises the RCCL path on a 1-GPU box)
    if (world > 1 or os.environ.get("AECM_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)
        kw = {}
        if backend == "nccl" and device_index is not None:
            kw["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, local_rank, world


def barrier(device_index=None):
    """Line the ranks up.  With the NCCL/RCCL backend the barrier runs on `device_index`."""
    if dist.is_initialized():
        if device_index is not None and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device_index])
        else:
            dist.barrier()


def gather_counters(frames: int, seconds: float, kernel_ms: float, device, name: str = ""):
    """Whole-job counters over the ranks (the only collective traffic of a run):
    returns dict(frames = sum, seconds = max, kernel_ms = max, ranks_seen = all-reduce of 1 per rank,
    per_rank = [(frames, seconds, kernel_ms)] in rank order)."""
    if not dist.is_initialized():
        return dict(frames=int(frames), seconds=float(seconds), kernel_ms=float(kernel_ms), ranks_seen=1,
                    per_rank=[(int(frames), float(seconds), float(kernel_ms))], backend=None, names=[name])
    world = dist.get_world_size()
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)                       # proves every rank took part in the collective
    mine = torch.tensor([float(frames), float(seconds), float(kernel_ms)], dtype=torch.float64, device=device)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    per_rank = [(int(round(r[0].item())), float(r[1].item()), float(r[2].item())) for r in rows]
    nb = name.encode()[:96].ljust(96, b"\0")                           # device names as fixed-size byte rows
    mine_n = torch.tensor(list(nb), dtype=torch.uint8, device=device)
    rows_n = [torch.zeros_like(mine_n) for _ in range(world)]
    dist.all_gather(rows_n, mine_n)
    names = [bytes(r.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for r in rows_n]
    return dict(names=names, frames=sum(p[0] for p in per_rank), seconds=max(p[1] for p in per_rank),
                kernel_ms=max(p[2] for p in per_rank), ranks_seen=int(one.item()), per_rank=per_rank,
                backend=dist.get_backend())


def all_ok(ok: bool, device) -> int:
    """How many ranks report ok (an all-reduce of 0/1 per rank; == world size when every rank's check passed)."""
    if not dist.is_initialized():
        return int(bool(ok))
    t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
