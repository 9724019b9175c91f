"""Prompt-parallel multi-GPU execution: one process per GPU, independent (prompt, seed) units
sharded across ranks, and exactly one collective -- an RCCL broadcast of the weight arenas from
rank 0 over xGMI at start-up.  No per-step communication.

The reference has no multi-device code at all (batch hard-wired to 1, no mx.distributed:
SURVEY.md 2.3); every (prompt, seed) is a complete independent trajectory
(pipelines/distilled.py:302), so the path shards as independent units ("weak" scaling).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    """RANK, WORLD_SIZE, local device index.  LTX2_LOCAL_DEVICE overrides LOCAL_RANK as the device index (several
    ranks on one GPU: the single-GPU rehearsal of the multi-process path, together with LTX2_DIST_BACKEND=gloo)."""
    local = int(os.environ.get("LTX2_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if "LTX2_LOCAL_DEVICE" not in os.environ and torch.cuda.is_available():
        n = torch.cuda.device_count()
        if n and local >= n:      # the launcher narrowed device visibility per rank (HIP_VISIBLE_DEVICES): index within what is visible
            local %= n
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), local


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  backend 'nccl' is RCCL on ROCm; 'gloo' for CPU tests."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("LTX2_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():       # whatever the backend: every rank works on ITS device (LOCAL_RANK / LTX2_LOCAL_DEVICE)
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of independent (prompt, seed) units: rank r takes r, r+G, r+2G, ..."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_units, world))


def _bcast_flat(flat: torch.Tensor, src: int, mode: str) -> None:
    """One flat buffer from rank `src` to everyone.  "ring": dist.broadcast (RCCL's ring / tree: every byte crosses ONE xGMI link per hop,
    ~153 GB/s).  "scatter": the root deals the buffer out in world_size chunks (chunk r to rank r: the root's 7 links carry different bytes in
    parallel) and an all-gather completes every rank's copy over all links at once -- the two-phase broadcast of SURVEY 5 for a point-to-point
    fabric.  Same result; which one is faster on the node is for the first 8-GPU run to say (bench.py reports the GB/s of the one it used)."""
    world = dist.get_world_size()
    if mode == "ring" or world == 1 or flat.numel() < world:
        dist.broadcast(flat, src=src)
        return
    # scatter + all-gather straight into the bucket: the only extra memory is this rank's 1 / world chunk (round 4 padded a copy of the bucket and
    # gathered into a second one: three buckets in flight); the < world elements that do not divide go by one small broadcast
    n = flat.numel()
    per = n // world
    main = per * world
    body = flat[:main]
    mine = torch.empty(per, dtype=flat.dtype, device=flat.device)
    chunks = list(body.view(world, per).unbind(0)) if dist.get_rank() == src else None
    dist.scatter(mine, chunks, src=src)
    dist.all_gather_into_tensor(body, mine)
    if main < n:
        dist.broadcast(flat[main:], src=src)


def broadcast_tensors(tensors: Dict[str, torch.Tensor], src: int = 0, bucket_bytes: int = 256 << 20, mode: Optional[str] = None) -> int:
    """Broadcast every tensor of `tensors` from rank `src`, in place, as large flat per-dtype
    buckets (few, large collectives: xGMI is point-to-point, ~153 GB/s per link, so ring traffic
    is per-link bound and small messages waste it).  All ranks must hold identically shaped
    tensors under identical names.  mode: "ring" (dist.broadcast) or "scatter" (scatter + all-gather, _bcast_flat); default: the
    LTX2_BCAST environment variable, else "ring".  Returns the number of buckets sent."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    mode = mode or os.environ.get("LTX2_BCAST", "ring")
    if mode not in ("ring", "scatter"):
        raise ValueError(f"broadcast mode {mode!r}: ring or scatter")
    names = sorted(tensors.keys())
    by_dtype: Dict[torch.dtype, List[str]] = {}
    for n in names:
        by_dtype.setdefault(tensors[n].dtype, []).append(n)
    n_coll = 0
    for dtype, group in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
        esz = tensors[group[0]].element_size()
        cap = max(1, bucket_bytes // esz)
        i = 0
        while i < len(group):
            # greedily pack whole tensors; a tensor larger than the bucket goes alone, unflattened
            t0 = tensors[group[i]]
            if not t0.is_contiguous():
                raise ValueError(f"broadcast_tensors: {group[i]} is not contiguous")
            if t0.numel() >= cap:
                _bcast_flat(t0.view(-1), src, mode)
                n_coll += 1
                i += 1
                continue
            j, total = i, 0
            while j < len(group) and total + tensors[group[j]].numel() <= cap:
                total += tensors[group[j]].numel()
                j += 1
            flat = torch.empty(total, dtype=dtype, device=t0.device)
            off = 0
            if dist.get_rank() == src:
                for n in group[i:j]:
                    k = tensors[n].numel()
                    flat[off:off + k].copy_(tensors[n].reshape(-1))
                    off += k
            _bcast_flat(flat, src, mode)
            n_coll += 1
            off = 0
            if dist.get_rank() != src:
                for n in group[i:j]:
                    k = tensors[n].numel()
                    tensors[n].view(-1).copy_(flat[off:off + k])      # view: a non-contiguous tensor must fail, not copy into a temporary
                    off += k
            i = j
    return n_coll


def tensors_checksum(tensors: Dict[str, torch.Tensor]) -> int:
    """Order-independent-free 63-bit checksum of the BYTES of every tensor (names in sorted order, each tensor's bytes summed as int64 words with a
    position weight): equal on two ranks iff -- up to a 2^-63 accident -- the replicas hold identical weights."""
    acc = 0
    for i, n in enumerate(sorted(tensors.keys())):
        t = tensors[n].contiguous().view(torch.uint8).reshape(-1)
        pad = (-t.numel()) % 8
        if pad:
            t = torch.cat([t, t.new_zeros(pad)])
        w = t.view(torch.int64)
        # position-weighted sum (wraps mod 2^64): a permutation of words inside a tensor changes it
        k = torch.arange(1, w.numel() + 1, device=w.device, dtype=torch.int64)
        acc = (acc * 1000003 + int((w * k).sum().item()) + (i + 1) * 7919) & 0x7FFFFFFFFFFFFFFF
    return acc


def replicas_identical(tensors: Dict[str, torch.Tensor], device: Optional[torch.device] = None) -> bool:
    """All ranks hold byte-identical `tensors`: the MIN and the MAX over ranks of tensors_checksum() agree (two all-reduces of one int64)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    c = tensors_checksum(tensors)
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    lo, hi = torch.tensor([c], dtype=torch.int64, device=dev), torch.tensor([c], dtype=torch.int64, device=dev)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return int(lo.item()) == int(hi.item())


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def count_ranks(device: Optional[torch.device] = None) -> int:
    """Number of ranks that actually joined the process group, counted THROUGH it: an all-reduce (sum) of a one on `device`
    (RCCL when the backend is nccl) -- not a copy of WORLD_SIZE."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 1
    t = torch.ones(1, dtype=torch.float32, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def gather_floats(values: Sequence[float], device: Optional[torch.device] = None) -> List[List[float]]:
    """Every rank's list of floats, on every rank (all_gather of a small tensor): per-rank timings / power for the bench line."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(values)]
    # gloo gathers host tensors only (its all_gather has no device path); RCCL gathers on the device
    t = torch.tensor(list(values), dtype=torch.float64, device=(device or "cuda") if dist.get_backend() == "nccl" else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o.tolist()] for o in out]


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
