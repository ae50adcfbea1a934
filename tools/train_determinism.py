#!/usr/bin/env python3
"""Two trainings from the same initial state on the same batches must produce bit-identical parameters (every
reduction of the backward is fixed-order; no atomics): 3 eager steps + 3 graphed steps, twice, compared tensor by tensor.

    python tools/train_determinism.py [--steps 3]
"""
import argparse
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd.optim import FusedAdamW  # noqa: E402
from uni_renderer_amd.train_step import GraphedTrainStep, train_step  # noqa: E402


def run(nets0, batch, steps, graph):
    nets = [copy.deepcopy(m) for m in nets0]
    for m in nets:
        m.train()
        m.requires_grad_(True)
    opt = FusedAdamW([p for m in nets for p in m.parameters()], lr=1e-4)
    losses = []
    if graph:
        g = GraphedTrainStep(nets, batch, opt, dtype=torch.bfloat16)
        for _ in range(steps):
            losses.append(float(g.step()["loss"]))
    else:
        for _ in range(steps):
            losses.append(float(train_step(nets, batch, optimizer=opt, dtype=torch.bfloat16)["loss"]))
    torch.cuda.synchronize()
    return [p.detach().clone() for m in nets for p in m.parameters()], losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    nets0 = bench.build_models(dev, torch.float32)
    B, L = a.batch, a.latent
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    batch = dict(x_t=mk(B, 4, L, L), cond=mk(B, 28, L, L), ehs=mk(B, 77, 768) * 0.5,
                 t_img=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 t_attr=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 target_img=mk(B, 4, L, L), target_attr=mk(B, 28, L, L))
    for graph in (False, True):
        pa, la = run(nets0, batch, a.steps, graph)
        pb, lb = run(nets0, batch, a.steps, graph)
        diff = sum(0 if torch.equal(x, y) else 1 for x, y in zip(pa, pb))
        print(f"{'graph' if graph else 'eager'}: losses {la} vs {lb}; {diff} of {len(pa)} parameter tensors differ after {a.steps} steps", flush=True)
        del pa, pb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
