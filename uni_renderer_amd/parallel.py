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


_FORWARD_ORDER = {"AttributeEncoderModel": 0, "UNet2DConditionModel": 1, "AttributeDecoderModel": 2}


@torch.no_grad()
def broadcast_parameters(modules: Iterable[torch.nn.Module], src: int = 0, group=None, chunk_mb: float = 256.0) -> int:
    """Make every rank start from rank ``src``'s parameters AND buffers -- what the constructor of each of the reference's
    three ``DistributedDataParallel`` wrappers does (train/train.py:1140-1142; torch DDP ``_sync_module_states``).  Without
    it ranks agree only if they were seeded identically; a rank that resumed from a different checkpoint would diverge
    silently.  Tensors are packed per dtype into flat chunks of ``chunk_mb`` (xGMI: few large messages) and copied back in
    place.  Returns the number of elements sent; a no-op without an initialised multi-rank group."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    seen, tensors = set(), []
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            if id(t) not in seen:
                seen.add(id(t))
                tensors.append(t.data)
    sent = 0
    by_kind = {}
    for t in tensors:
        by_kind.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, dev), ts in by_kind.items():
        if not (dtype.is_floating_point or dtype in (torch.int64, torch.int32, torch.uint8, torch.bool)):
            continue
        cap = max(1, int(chunk_mb * (1 << 20)) // max(1, torch.empty((), dtype=dtype).element_size()))
        i = 0
        while i < len(ts):
            part, n = [], 0
            while i < len(ts) and (not part or n + ts[i].numel() <= cap):
                part.append(ts[i])
                n += ts[i].numel()
                i += 1
            wire = torch.uint8 if dtype == torch.bool else dtype
            flat = torch.empty(n, dtype=wire, device=dev)
            off = 0
            for t in part:
                flat[off:off + t.numel()].copy_(t.reshape(-1))
                off += t.numel()
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in part:
                t.copy_(flat[off:off + t.numel()].view(t.shape))
                off += t.numel()
            sent += n
    return sent


def rs_ag_shard(buf: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank ``rank``'s slice of a flat bucket for reduce-scatter + all-gather: the bucket is ``world`` equal contiguous
    slices in rank order (``reduce_scatter_tensor`` / ``all_gather_into_tensor`` semantics), returned as a VIEW so the
    reduce-scatter lands in place and the all-gather reads it in place."""
    if buf.dim() != 1 or buf.numel() % world:
        raise ValueError(f"bucket of {tuple(buf.shape)} elements does not split over {world} ranks")
    return buf.view(world, -1)[rank]


class GradientBuckets:
    """Flat gradient buckets over several modules, all-reduced (mean) WHILE the backward pass is still running.

    The reference wraps enc, dec and unet in three DistributedDataParallel instances (train.py:1140-1142): three
    independent 25 MB bucket streams, synchronised inside ``accelerator.backward(loss)`` (1421).  Here:

      * ONE bucket list over all parameters, in reverse forward order (dec -> unet -> enc, each module's parameters
        reversed): the order in which the backward pass finishes them.  Buckets are large (default 256 MB: on xGMI every
        GPU has 7 point-to-point links, a few large messages beat many small ones).
      * Gradients LIVE in the buckets: every ``p.grad`` is a view into its bucket's flat fp32 buffer (autograd accumulates
        in place), so there is no gather copy before the collective and no scatter after it.  Use ``zero_grad()`` of
        this object (zeroes the flat buffers) instead of ``optimizer.zero_grad(set_to_none=True)``.
      * Overlap: a post-accumulate hook per parameter counts arrivals; when a bucket is complete -- and every bucket
        before it has been launched, so that all ranks issue the collectives in the SAME order -- its collective is
        enqueued asynchronously on the process group's stream while autograd keeps running the remaining backward
        kernels.  ``finish()`` (after ``backward()``) launches what is left, waits, and writes the means back.
      * Ranks may take different data-dependent branches (``compute_t`` draws ``random`` per rank, train.py:445): a
        parameter without a gradient on this rank simply leaves its (zeroed) slice of the bucket untouched; the bucket is
        then launched from ``finish()``, still in index order -- no deadlock, no mismatched collectives.
      * Transport: ``comm_dtype`` (e.g. bf16: 3.5 GB instead of 7 GB per step) and ``algorithm``: "all_reduce", or
        "rs_ag" = reduce-scatter + all-gather of the flat bucket, the form that uses all seven xGMI links of a GPU at
        once (~11 ms vs ~80 ms for a ring over 6.98 GB of fp32).
    The collectives are issued from Python hooks during ``backward()``.  Eagerly that is DDP's overlap; inside
    ``train_step.GraphedTrainStep(capture_collectives=True)`` (RCCL) the same hooks fire during the capture and every
    collective becomes a parallel branch of the ONE step graph.
    """

    def __init__(self, modules: Iterable[torch.nn.Module], bucket_mb: float = 256.0, comm_dtype=None,
                 algorithm: str = "all_reduce", overlap: bool = True, process_group=None, force_collectives: bool = False,
                 broadcast: bool = True):
        """``force_collectives``: issue the collectives even in a one-rank group (tests: drives RCCL's
        all-reduce / reduce-scatter / all-gather and their stream ordering on a single GPU).
        ``broadcast``: start every rank from rank 0's parameters and buffers, as each DDP constructor of the reference
        does (train.py:1140-1142); see ``broadcast_parameters``."""
        mods = list(modules)
        mods.sort(key=lambda m: _FORWARD_ORDER.get(getattr(m, "module", m).__class__.__name__, 1))  # stable for others
        self.broadcast_elements = broadcast_parameters(mods, 0, process_group) if broadcast else 0
        params = [p for m in mods for p in m.parameters() if p.requires_grad]
        self.params = list(reversed(params))
        self.comm_dtype = comm_dtype
        self.algorithm = algorithm
        self.overlap = overlap
        self.group = process_group
        self.force = bool(force_collectives)
        if algorithm not in ("all_reduce", "rs_ag"):
            raise ValueError(algorithm)
        cap = int(bucket_mb * (1 << 20))
        esz = 2 if comm_dtype in (torch.bfloat16, torch.float16) else 4
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in self.params:
            nbytes = p.numel() * esz
            if cur and size + nbytes > cap:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        # flat fp32 storage, gradients as views (padded to a multiple of 8 ranks x 8 elements for reduce-scatter)
        self.flat: List[torch.Tensor] = []
        self._bucket_of, self._offset = {}, {}
        for bi, bucket in enumerate(self.buckets):
            # every gradient view starts on a 16-byte boundary (optim.FusedAdamW's float4 path; RCCL likes it too)
            n = sum((p.numel() + 3) // 4 * 4 for p in bucket)
            npad = (n + 63) // 64 * 64
            dev = bucket[0].device
            flat = torch.zeros(npad, dtype=torch.float32, device=dev)
            off = 0
            for p in bucket:
                if p.dtype != torch.float32:
                    raise RuntimeError("GradientBuckets holds fp32 gradients of fp32 master parameters (train.py:1082-1089)")
                g = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g
                self._bucket_of[p] = bi
                self._offset[p] = off
                off += (p.numel() + 3) // 4 * 4
            self.flat.append(flat)
        self._by_id = {id(p): p for p in self.params}
        # gradients are written straight into the bucket slices by the kernels that produce them (backward.GradSink) instead of
        # being added into zeroed buckets by autograd; False restores the round-5 protocol (zero + accumulate) for A/B runs
        self.direct_write = True
        self.sumsq_parts: Optional[List[torch.Tensor]] = None  # sums of squares collected by finish()'s copy-back pass
        self._comm: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._arrived = [0] * len(self.buckets)
        self._launched = 0
        self._work: List = []
        self._in_backward = True
        self._sync = True  # False inside no_sync(): gradients accumulate locally, no collective is issued
        self.launched_from_hooks = 0  # diagnostics: buckets whose collective was enqueued during backward()
        self._hooks = []
        if overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ---- helpers -------------------------------------------------------------------------------------------------
    def _world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _single(self) -> bool:
        return self._world() == 1 and not (self.force and dist.is_initialized())

    def no_sync(self):
        """Context manager for gradient accumulation -- DDP's ``no_sync()``, which ``accelerator.accumulate(controlnet,
        controldec, unet)`` enters on every micro-step but the last (train/train.py:1236, flag --gradient_accumulation_steps
        at 621).  Inside it ``backward()`` only accumulates into the flat buffers (the ``p.grad`` views): no hook counts an
        arrival, ``finish()`` issues nothing.  The first ``backward()`` + ``finish()`` outside it reduces the ACCUMULATED
        gradients.  Do not call ``zero_grad()`` between the micro-steps of one optimisation step."""
        buckets = self

        class _NoSync:
            def __enter__(self_inner):
                self_inner.prev = buckets._sync
                buckets._sync = False
                return buckets

            def __exit__(self_inner, *exc):
                buckets._sync = self_inner.prev
                return False

        return _NoSync()

    def _on_grad(self, p):
        if self._single() or not self._sync:
            return
        bi = self._bucket_of[p]
        if p.grad is not None and p.grad.data_ptr() != self._view_ptr(p):  # someone re-created .grad: pull it back in
            self._adopt(p)
        self._arrived[bi] += 1
        while self._launched < len(self.buckets) and self._arrived[self._launched] >= len(self.buckets[self._launched]):
            self._launch(self._launched)
            self.launched_from_hooks += 1

    def _view_ptr(self, p) -> int:
        return self.flat[self._bucket_of[p]].data_ptr() + 4 * self._offset[p]

    def _view(self, p) -> torch.Tensor:
        off = self._offset[p]
        return self.flat[self._bucket_of[p]][off:off + p.numel()].view_as(p)

    def _adopt(self, p):
        view = self._view(p)
        view.copy_(p.grad)
        p.grad = view

    def _launch(self, bi: int):
        """Enqueue the collective of bucket ``bi`` (buckets are launched strictly in index order on every rank)."""
        assert bi == self._launched
        world = self._world()
        flat = self.flat[bi]
        if self.comm_dtype in (torch.bfloat16, torch.float16):
            buf = self._comm[bi]
            if buf is None:
                buf = self._comm[bi] = torch.empty(flat.numel(), dtype=self.comm_dtype, device=flat.device)
            torch.mul(flat, 1.0 / world, out=buf)  # mean folded in before the cast; ONE kernel per bucket
        else:
            buf = flat
            buf.mul_(1.0 / world)
        if self.algorithm == "rs_ag" and buf.numel() % world == 0 and dist.get_backend(self.group) != "gloo":
            # reduce-scatter into this rank's slice of the bucket, then all-gather IN PLACE (the slice is the input): both
            # are enqueued here, back to back on the process group's stream -- RCCL orders them -- so the all-gather also
            # runs under the rest of the backward instead of inside finish().  (gloo has no reduce_scatter_tensor: the CPU
            # test transport takes the all-reduce branch.)
            shard = rs_ag_shard(buf, dist.get_rank(self.group), world)
            dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w2 = dist.all_gather_into_tensor(buf, shard, group=self.group, async_op=True)
            self._work.append((bi, w2, "rs_ag"))
        else:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._work.append((bi, w, "ar"))
        self._launched += 1

    # ---- the step protocol ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def finish(self):
        """After ``backward()``: launch the buckets the hooks could not (missing gradients on this rank), wait for all
        collectives, and leave the mean gradients in the flat buffers (= in every ``p.grad``)."""
        if self._single():
            from . import backward as bw
            bw.grad_sink.end()
            self.sumsq_parts = None
            self._reset()
            return
        if not self._sync:  # accumulation micro-step: keep the local sums, adopt gradients re-created outside the buckets
            for p in self.params:
                if p.grad is not None and p.grad.data_ptr() != self._view_ptr(p):
                    self._adopt(p)
            self._reset()
            return
        for p in self.params:  # a .grad re-created outside (set_to_none + a fresh backward): adopt it
            if p.grad is None:
                self._adopt_none(p)
            elif p.grad.data_ptr() != self._view_ptr(p):
                self._adopt(p)
        while self._launched < len(self.buckets):
            self._launch(self._launched)
        from . import backward as bw
        parts = []
        for bi, w, kind in self._work:
            w.wait()
            buf = self._comm[bi] if self._comm[bi] is not None else self.flat[bi]
            if buf is not self.flat[bi]:
                if buf.is_cuda and bw.FUSED_GRADNORM:
                    # reduced half-precision means -> the fp32 gradients, and the clipping norm's sums of squares from the same
                    # pass (ur_cast_multi_sumsq) instead of another sweep over the 7 GB (train.py:1422)
                    _, part = bw.cast_many([buf], torch.float32, sumsq=True, outs=[self.flat[bi]])
                    parts.append(part)
                else:
                    self.flat[bi].copy_(buf)
        self.sumsq_parts = parts if parts and len(parts) == len(self.flat) else None
        bw.grad_sink.end()
        self._reset()

    @torch.no_grad()
    def adopt_all(self) -> int:
        """After ``backward()``: every gradient that did NOT arrive through the sink or a hook's adoption (a parameter outside the
        batched cast / pack / barrier nodes: autograd then keeps its own tensor as ``p.grad``) is copied into its bucket slice
        with ONE multi-tensor launch.  train_step calls this right behind ``backward()`` -- INSIDE the captured forward + backward
        graph -- because those tensors live in the graph's pool: on a replay nothing on the host would notice that they changed
        (``finish()`` compares ``p.grad`` objects, which a replay does not touch).  Returns how many were copied."""
        src, dst, ps = [], [], []
        for p in self.params:
            if p.grad is not None and p.grad.data_ptr() != self._view_ptr(p):
                src.append(p.grad)
                dst.append(self._view(p))
                ps.append(p)
        if dst:
            torch._foreach_copy_(dst, src)
            for p, v in zip(ps, dst):
                p.grad = v
        self.adopted_last = len(dst)
        return len(dst)

    def _adopt_none(self, p):
        v = self._view(p)
        if self.direct_write:
            v.zero_()  # nobody wrote this slice in this step (it still holds the previous step's gradient)
        p.grad = v  # zeros

    def _reset(self):
        self._arrived = [0] * len(self.buckets)
        self._launched = 0
        self._work = []

    # kept name of the round-1 API (all-reduce after the backward): now the tail of the overlapped protocol
    all_reduce_mean = finish

    def _sink_view(self, pid):
        p = self._by_id.get(pid)
        return None if p is None else self._view(p)  # a NEW tensor object: AccumulateGrad adopts it only if nobody else holds it

    @torch.no_grad()
    def zero_grad(self):
        """Start an optimisation step.  ``direct_write`` (default): nothing is zeroed -- every ``p.grad`` is dropped and the
        backward's kernels STORE the gradients into the bucket slices (backward.GradSink; autograd adopts the views as
        ``p.grad``); a parameter the backward does not reach gets its slice zeroed in ``finish()``.  Otherwise (round-5
        protocol): zero the flat buffers, the gradients stay views into them and autograd adds into them."""
        from . import backward as bw
        if self.direct_write:
            for p in self.params:
                p.grad = None
            bw.grad_sink.begin(self._sink_view)
        else:
            bw.grad_sink.end()
            for p in self.params:
                if p.grad is None or p.grad.data_ptr() != self._view_ptr(p):
                    p.grad = self._view(p)
            for f in self.flat:
                f.zero_()
        self.sumsq_parts = None
        self._reset()

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """train.py:1422-1424 over the flat buffers: one norm per bucket instead of one per parameter."""
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(list(self.flat))))
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        for f in self.flat:
            f.mul_(coef)
        return total
