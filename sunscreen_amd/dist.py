"""Multi-GPU plumbing: batch sharding, max-over-ranks timing and the optional result gather.

The path shards by independent ciphertexts: rank r owns a contiguous block of the batch,
context tables and keys are replicated, and there is no collective on the data path.  The only
collectives are barriers, a MAX all-reduce of elapsed time and (optionally) an all_gather of results.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
import time
from typing import Callable

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of `total` items owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment. Returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, local_rank, world


def barrier_sync() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_steps(step: Callable[[], None], steps: int, warmup: int, device: str | torch.device = "cpu") -> float:
    """W untimed warm-up steps, then exactly `steps` timed steps bracketed by barrier+synchronize on both
    sides; returns the MAX elapsed seconds over all ranks."""
    for _ in range(warmup):
        step()
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier_sync()
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def gather_results(local: torch.Tensor, total: int) -> torch.Tensor | None:
    """Gather per-rank result blocks (shard_range order) on every rank: int64[total, ...].
    Blocks may differ in length by one item, so they are padded to a common length for all_gather."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    longest = max(shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world))
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out, dim=0)


def broadcast_bytes(data: bytes | None, src: int = 0, device: str | torch.device = "cpu") -> bytes:
    """One-time replication of a serialised object (keys are broadcast from their owner, 2.6 MB ... 486 MB):
    rank `src` passes the bytes, every other rank passes None; all ranks return the same bytes.  Two collectives (length,
    payload); over RCCL the payload travels as one uint8 tensor on `device`."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert data is not None
        return data
    rank = dist.get_rank()
    n = torch.tensor([len(data) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    if rank == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty((int(n.item()),), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


def replicate_keys(ctx, key, cls, src: int = 0, device: str | torch.device = "cpu"):
    """The key owner (rank `src`) holds `key` (a sunscreen_amd.seal RelinearizationKeys / GaloisKeys / PublicKey object);
    every rank returns a device-resident copy, made from the SEAL wire format (uncompressed: the bytes cross xGMI once,
    zstd would cost more host time than the transfer).  Evaluation then needs no further communication."""
    blob = broadcast_bytes(key.as_bytes(compression=0) if (not dist.is_initialized() or dist.get_rank() == src) else None, src, device)
    if dist.is_initialized() and dist.get_world_size() > 1 and dist.get_rank() != src:
        return cls.from_bytes(ctx, blob)
    return key
