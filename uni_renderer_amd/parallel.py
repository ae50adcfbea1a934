"""Batch-sharded data parallelism over one 8-GPU MI355X node: one process per GPU, ``torch.distributed`` with
backend "nccl" (= RCCL over xGMI on ROCm); "gloo" on CPU for tests.

The denoise path shards over the batch with NO data-path collective (every op is per-sample: GroupNorm,
LayerNorm and attention never mix samples -- SURVEY.md §8e): inference = replicate weights, split the batch,
optionally gather the results.  The only collective the reference ever issues is the gradient all-reduce of its
three DDP wrappers (train/train.py:1140-1142, 1421); ``GradientBuckets`` restates it as ONE flat bucket list
over enc + unet + dec (reverse registration order ~ backward order), with large buckets sized for xGMI
(7 direct links per GPU: prefer few, large messages) and optional bf16 transport.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None) -> tuple:
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns
    (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local) if device is None else device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_bounds(n: int, rank: int, world: int) -> tuple:
    """Contiguous, balanced [lo, hi) of a batch of n over `world` ranks (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(tensors: Sequence[torch.Tensor], rank: int, world: int) -> List[torch.Tensor]:
    out = []
    for t in tensors:
        if t.shape[0] == 1:  # broadcast entries such as a single prompt embedding
            out.append(t)
        else:
            lo, hi = shard_bounds(t.shape[0], rank, world)
            out.append(t[lo:hi])
    return out


def gather_batch(local: torch.Tensor, total: int, rank: int, world: int) -> Optional[torch.Tensor]:
    """Reassemble per-rank shards on every rank (all_gather of padded shards; shards may be ragged)."""
    if world == 1:
        return local
    q = (total + world - 1) // world
    pad = torch.zeros((q,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        parts.append(bufs[r][: hi - lo])
    return torch.cat(parts, 0)


def max_over_ranks(seconds: float, device=None) -> float:
    """bench.py's timing rule: the step time of the job is the slowest rank's."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class GradientBuckets:
    """Flat gradient buckets over several modules, all-reduced (mean) in reverse parameter order.

    The reference wraps enc, dec and unet in three DistributedDataParallel instances (train.py:1140-1142): three
    independent 25 MB bucket streams.  Here all parameters form one list, bucketed at ``bucket_mb`` (default
    256 MB: on xGMI every GPU has 7 point-to-point links, so a few large messages beat many small ones), and each
    bucket is reduced with one collective.  Ranks may take different data-dependent branches (``compute_t`` draws
    ``random`` per rank, train.py:445): parameters without a gradient contribute zeros, so every rank issues the
    same collectives in the same order.
    """

    def __init__(self, modules: Iterable[torch.nn.Module], bucket_mb: float = 256.0, comm_dtype=None):
        params = [p for m in modules for p in m.parameters() if p.requires_grad]
        self.params = list(reversed(params))
        self.comm_dtype = comm_dtype
        cap = int(bucket_mb * (1 << 20))
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in self.params:
            nbytes = p.numel() * (2 if comm_dtype in (torch.bfloat16, torch.float16) else 4)
            if cur and size + nbytes > cap:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)

    @torch.no_grad()
    def all_reduce_mean(self):
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        for bucket in self.buckets:
            dt = self.comm_dtype or torch.float32
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt) for p in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(world)
            off = 0
            for p in bucket:
                n = p.numel()
                g = flat[off:off + n].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
