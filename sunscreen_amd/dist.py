"""Multi-GPU plumbing: batch sharding, max-over-ranks timing, the optional result gather and the one
data-path exchange the scope has (the cross-GPU sum of examples/pir's row-sharded database).

The path shards by independent ciphertexts: rank r owns a contiguous block of the batch,
context tables and keys are replicated, and there is no collective on the data path.  The only
collectives are barriers, a MAX all-reduce of elapsed time and (optionally) a gather of results to a root.
Exception (SURVEY 8e "Exception", config 5a): a database sharded by row leaves one partial ciphertext per
GPU; `reduce_ciphertexts` gathers them on the root (world x 2*K*N*8 bytes, point-to-point over xGMI) and adds.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests and by the one-device validation runs.
"""
from __future__ import annotations

import datetime
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
    if (world > 1 or os.environ.get("HIPBFV_DIST_FORCE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world, timeout=init_timeout())
    return rank, local_rank, world


def init_timeout() -> datetime.timedelta:
    """Explicit rendezvous / collective timeout (HIPBFV_DIST_TIMEOUT_S, default 600 s): a rank that never arrives fails
    the job with a message instead of leaving the others in the default 30-minute wait."""
    return datetime.timedelta(seconds=int(os.environ.get("HIPBFV_DIST_TIMEOUT_S", "600")))


def is_nccl() -> bool:
    return dist.is_initialized() and dist.get_backend() == "nccl"


def solo() -> bool:
    """True when no collective needs to be issued: no process group, or a group of ONE rank.  HIPBFV_DIST_FORCE=1 makes a
    single-rank group issue every collective anyway -- the way a 1-GPU box executes the RCCL code path of this module
    (tests/test_gpu_dist.py; a world of one is a complete, if trivial, communicator: broadcasts, gathers and all-reduces go
    through librccl and complete on the device)."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and os.environ.get("HIPBFV_DIST_FORCE") != "1"


def barrier() -> None:
    """dist.barrier on the device this rank drives: under RCCL the barrier is an all-reduce on a device tensor, and
    without `device_ids` torch guesses the device from the global rank (wrong whenever LOCAL_RANK != RANK % devices)."""
    if solo():
        return
    if is_nccl():
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def barrier_sync() -> None:
    barrier()
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
    if not solo():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def _via(t: torch.Tensor) -> torch.Tensor:
    """The tensor a collective is issued on: device tensors travel as they are under RCCL; gloo gets a host copy."""
    return t if (is_nccl() or not t.is_cuda) else t.cpu()


def gather_results(local: torch.Tensor, total: int, root: int = 0) -> torch.Tensor | None:
    """Gather per-rank result blocks (shard_range order) on `root`: int64[total, ...] there, None elsewhere (SURVEY 8e:
    a `gather` to the consumer's device, point-to-point over the 7 xGMI links at once -- not an all_gather, nobody else
    needs the copy).  Blocks may differ in length by one item, so they are padded to a common length."""
    if solo():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    longest = max(shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world))
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    send = _via(pad)
    parts = [torch.empty_like(send) for _ in range(world)] if rank == root else None
    dist.gather(send, parts, dst=root)
    if rank != root:
        return None
    out = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        out.append(parts[r][: hi - lo].to(local.device))
    return torch.cat(out, dim=0)


def reduce_ciphertexts(local: torch.Tensor, add: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], root: int = 0) -> torch.Tensor | None:
    """Sum one ciphertext batch per rank on `root` (SURVEY 8e "Exception": examples/pir with the database sharded by row --
    every GPU holds the partial sum over its rows, the answer is their sum).  local: int64[count, size, K, N] on every
    rank; `add(a, b)` is the evaluator's ciphertext addition (BatchEvaluator.add on the GPU).  Gather + add rather than an
    all-reduce: the sum is modulo a different prime per residue row, which RCCL's reductions cannot express, and only the
    root needs it (world x 2 MiB at n = 16384, point-to-point).  Returns the sum on `root`, None elsewhere; modular
    addition of canonical residues is associative and commutative, so the bits do not depend on the rank order."""
    if solo():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    send = _via(local.contiguous())
    parts = [torch.empty_like(send) for _ in range(world)] if rank == root else None
    dist.gather(send, parts, dst=root)
    if rank != root:
        return None
    acc = parts[0].to(local.device)
    for r in range(1, world):
        acc = add(acc, parts[r].to(local.device))
    return acc


def broadcast_tensor(t: torch.Tensor | None, shape, dtype, device, src: int = 0) -> torch.Tensor:
    """One tensor from `src` to every rank (the client's query ciphertexts reaching every database shard)."""
    if solo():
        assert t is not None
        return t
    if dist.get_rank() != src:
        t = torch.empty(tuple(shape), dtype=dtype, device=device)
    buf = _via(t.contiguous())
    # RCCL moves one message per call; keep each below 1 GiB (int32 element counts in some transports)
    flat = buf.view(-1)
    step = (1 << 30) // flat.element_size()
    for o in range(0, flat.numel(), step):
        dist.broadcast(flat[o : o + step], src=src)
    return buf.to(device) if buf.device != torch.device(device) else buf


def broadcast_bytes(data: bytes | None, src: int = 0, device: str | torch.device = "cpu") -> bytes:
    """One-time replication of a serialised object (keys are broadcast from their owner, 2.6 MB ... 486 MB):
    rank `src` passes the bytes, every other rank passes None; all ranks return the same bytes.  Two collectives (length,
    payload); over RCCL the payload travels as uint8 tensors on `device`, in messages of at most 256 MiB (the 486 MB Galois
    key set of n = 16384 is two of them)."""
    if solo():
        assert data is not None
        return data
    rank = dist.get_rank()
    n = torch.tensor([len(data) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    if rank == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty((int(n.item()),), dtype=torch.uint8, device=device)
    step = 256 << 20
    for o in range(0, buf.numel(), step):
        dist.broadcast(buf[o : o + step], src=src)
    return bytes(buf.cpu().numpy().tobytes())


def replicate_keys(ctx, key, cls, src: int = 0, device: str | torch.device = "cpu"):
    """The key owner (rank `src`) holds `key` (a sunscreen_amd.seal RelinearizationKeys / GaloisKeys / PublicKey object);
    every rank returns a device-resident copy, made from the SEAL wire format (uncompressed: the bytes cross xGMI once,
    zstd would cost more host time than the transfer).  Evaluation then needs no further communication."""
    if solo():
        return key
    blob = broadcast_bytes(key.as_bytes(compression=0) if dist.get_rank() == src else None, src, device)
    if dist.get_rank() != src:
        return cls.from_bytes(ctx, blob)
    return key
