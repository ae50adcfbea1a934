#!/usr/bin/env python3
"""Training-step timing (cfg 4 of BASELINE.json: 512x512, per-GPU batch 4, bf16 compute, fp32 master parameters, AdamW,
batch-sharded data parallel with one flat gradient all-reduce over RCCL).  NOT the headline metric (bench.py is);
this reports where the first functional training path stands.

    python tools/train_bench.py [--batch 4] [--latent 64] [--steps 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd.parallel import GradientBuckets  # noqa: E402
from uni_renderer_amd.train_step import train_step  # noqa: E402


def _queue_stats():
    """Deferred weight gradients keep (dy, x) of every layer of a network alive until its barrier (backward.WgradQueue):
    the peak of those bytes, the cap (UR_WGRAD_PENDING_MB) and how often it forced an early flush."""
    from uni_renderer_amd import backward as B

    q = B.wgrad_queue
    return dict(deferred=B.WGRAD_DEFER, peak_pending_gb=round(q.peak_pending / 2**30, 2), cap_gb=round(q.cap / 2**30, 1),
                early_flushes=q.early_flushes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--algorithm", default="all_reduce", choices=["all_reduce", "rs_ag"],
                    help="gradient collective per bucket: one all-reduce, or reduce-scatter + all-gather (all 7 xGMI links)")
    ap.add_argument("--comm-dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-overlap", action="store_true", help="reduce all buckets after the backward instead of during it")
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW(fused=True) instead of optim.FusedAdamW")
    ap.add_argument("--gpus", type=int, default=0, help="without a launcher: start this many ranks (one per GPU) ourselves")
    ap.add_argument("--graph", action="store_true", help="time HIP-graph replays (train_step.GraphedTrainStep): one GPU = the whole step "
                    "as one graph; several = ONE graph with the bucket collectives captured as parallel branches (overlapped "
                    "with the backward), or with --no-overlap: forward + backward graph, eager collectives, update graph")
    ap.add_argument("--force-collectives", action="store_true", help="one rank: still build the buckets and issue the RCCL "
                    "collectives (world size 1), to time / profile the multi-GPU code path on one GPU")
    ap.add_argument("--gradient-checkpointing", action="store_true", help="enable_gradient_checkpointing() on the three networks "
                    "(train/train.py:1073-1074): the resnets of the flagged blocks are recomputed in the backward")
    ap.add_argument("--accumulate-into-buckets", action="store_true", help="round-5 gradient protocol (zero the flat buckets, autograd "
                    "adds every gradient into them) instead of round 6's direct writes (backward.GradSink): A/B runs")
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        bench._spawn_ranks(args.gpus, script=__file__)  # does not return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_collectives:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        torch.distributed.init_process_group("nccl", device_id=dev)
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    nets = bench.build_models(dev, torch.float32)  # fp32 master parameters
    for m in nets:
        m.train()
        m.requires_grad_(True)
        if args.gradient_checkpointing:
            m.enable_gradient_checkpointing()
    B, L = args.batch, args.latent
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    batch = dict(x_t=mk(B, 4, L, L), cond=mk(B, 28, L, L), ehs=mk(B, 77, 768) * 0.5,
                 t_img=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 t_attr=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 target_img=mk(B, 4, L, L), target_attr=mk(B, 28, L, L))
    params = [p for m in nets for p in m.parameters()]
    if args.torch_adamw:
        opt = torch.optim.AdamW(params, lr=1e-5, fused=True, capturable=args.graph)
    else:
        from uni_renderer_amd.optim import FusedAdamW
        opt = FusedAdamW(params, lr=1e-5)
    buckets = GradientBuckets(nets, comm_dtype=(torch.bfloat16 if args.comm_dtype == "bf16" else None), algorithm=args.algorithm,
                              overlap=not args.no_overlap, force_collectives=args.force_collectives) \
        if (world > 1 or args.force_collectives) else None
    if buckets is not None and args.accumulate_into_buckets:
        buckets.direct_write = False
    if args.graph:
        # one GPU: the whole step is one graph; several: forward + backward graph, eager bucket collectives and update
        from uni_renderer_amd.train_step import GraphedTrainStep
        gstep = GraphedTrainStep(nets, batch, opt, buckets=buckets, dtype=dt)
        gstats = gstep.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gstats = gstep.step()
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / args.steps
        stats = {k: float(v) for k, v in gstats.items()}
        phases = {k: round(v, 2) for k, v in gstep.phase_times().items()}
        phases["collectives_captured_in_graph"] = gstep.capture_collectives
        if gstep.capture_collectives:
            phases["collectives_forked_during_backward"] = gstep.collectives_from_hooks
    else:
        stats = train_step(nets, batch, optimizer=opt, buckets=buckets, dtype=dt)  # warm-up (packs nothing: weights change)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            stats = train_step(nets, batch, optimizer=opt, buckets=buckets, dtype=dt)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / args.steps
    if world > 1:
        t = torch.tensor([dtm], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dtm = float(t.item())
    if rank == 0:
        print(json.dumps(dict(metric="train-steps/sec (dual-UNet, 512^2, per-GPU batch %d)" % B, value=round(world / dtm, 4),
                              ms_per_step=round(dtm * 1e3, 1), n_gpus=world, dtype=args.dtype, loss=stats["loss"],
                              grad_norm=stats.get("grad_norm"), graph=bool(args.graph),
                              grad_sync=(dict(algorithm=args.algorithm, comm_dtype=args.comm_dtype, overlap=not args.no_overlap,
                                              buckets=len(buckets.buckets), launched_during_backward=buckets.launched_from_hooks)
                                         if buckets is not None else None), peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 1),
                              wgrad_queue=_queue_stats(), phase_ms=(phases if args.graph else None))))
    if world > 1 or args.force_collectives:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
