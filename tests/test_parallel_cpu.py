"""world_size-2 gloo tests (CPU) of the N > 1 path: batch sharding / gathering without a data-path collective,
the max-over-ranks timing rule, and the single flat gradient all-reduce over the three networks, including the
case where ranks take different branches (a rank has no gradient for some parameters)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from uni_renderer_amd import parallel

    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    # --- batch sharding: 5 samples over 2 ranks -> 3 + 2, no collective on the data path
    x = torch.arange(5 * 3, dtype=torch.float32).view(5, 3)
    prompt = torch.ones(1, 4)
    xs, ps = parallel.shard_batch([x, prompt], rank, world)
    assert xs.shape[0] == (3 if rank == 0 else 2) and ps.shape[0] == 1
    y = xs * 2 + rank * 0  # "denoise" the local shard
    full = parallel.gather_batch(y, 5, rank, world)
    assert torch.equal(full, x * 2)
    # --- timing rule
    t = parallel.max_over_ranks(1.0 + rank)
    assert t == 2.0
    # --- one flat gradient all-reduce over three modules; rank 1 has no grad for module c ("other branch")
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(4, 3), torch.nn.Linear(3, 2), torch.nn.Linear(2, 2)
    for m in (a, b) + ((c,) if rank == 0 else ()):
        for p in m.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
    gb = parallel.GradientBuckets([a, b, c], bucket_mb=1e-5)  # force several buckets
    assert len(gb.buckets) > 1
    gb.all_reduce_mean()
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for m in (a, b) for p in m.parameters())
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, 0.5)) for p in c.parameters())
    gb16 = parallel.GradientBuckets([a], comm_dtype=torch.bfloat16)
    gb16.all_reduce_mean()
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in a.parameters())
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_shard_bounds_cover_everything():
    from uni_renderer_amd.parallel import shard_bounds

    for n in (1, 4, 5, 32, 33):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
