"""Multi-GPU plumbing: AECM streams are independent, so the hot path shards by static contiguous
ranges with NO data-path collective.  torch.distributed (RCCL on GPUs, gloo in CPU tests) is used
only to line the ranks up for timing and to gather a handful of throughput counters."""
from __future__ import annotations

import os

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


def init(backend: str):
    rank, local_rank, world = env_rank_world()
    # AECM_FORCE_DIST=1 initialises the process group even for one rank (exercises the RCCL path on a 1-GPU box)
    if (world > 1 or os.environ.get("AECM_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier(device_index=None):
    """Line the ranks up.  With the NCCL/RCCL backend the barrier runs on `device_index`."""
    if dist.is_initialized():
        if device_index is not None and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device_index])
        else:
            dist.barrier()


def gather_counters(frames: int, seconds: float, kernel_ms: float, device):
    """Whole-job counters: (sum of frames, max of wall seconds, max of kernel ms) over ranks."""
    if not dist.is_initialized():
        return int(frames), float(seconds), float(kernel_ms)
    s = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    m = torch.tensor([float(seconds), float(kernel_ms)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(round(s.item())), float(m[0].item()), float(m[1].item())
