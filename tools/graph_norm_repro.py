#!/usr/bin/env python3
"""A captured ``torch.linalg.vector_norm`` over a large tensor returns wrong values on replay (torch 2.10 + ROCm 7.x on
MI355X) -- found while building train_step.GraphedTrainStep: the clipping norm over the 32 MB flat
gradient buckets, captured into a second graph, disagreed with the eager norm at the same point.  Pure torch, no
library code: graph A = zero_ + add_ on four 32 MB buffers, graph B = the norms of the four buffers, eager norms in
between.  From the second replay on graph B's ``vector_norm`` values are wrong for buffers 1..3 while eager norms before
and after it are right; ``torch._foreach_norm`` in the same place (``--foreach``) is right.  ``--no-round-trip`` leaves
out the D2H / H2D copies (they turned out to be irrelevant).
-> gpurun_out/graph_norm_repro.json"""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
bufs = [torch.zeros(8 << 20, device=dev) for _ in range(4)]
src = [torch.randn(8 << 20, device=dev) for _ in range(4)]
host = [torch.empty(8 << 20, pin_memory=True) for _ in range(4)]
side = torch.cuda.Stream()

ga = torch.cuda.CUDAGraph()
with torch.cuda.graph(ga):
    for b, s in zip(bufs, src):
        b.zero_()
        b.add_(s)
FOREACH = "--foreach" in __import__("sys").argv
gb = torch.cuda.CUDAGraph()
with torch.cuda.graph(gb):
    out = torch.stack(torch._foreach_norm(bufs)) if FOREACH else torch.stack([torch.linalg.vector_norm(b) for b in bufs])

import sys
ROUND_TRIP = "--no-round-trip" not in sys.argv
rows = []
for it in range(4):
    ga.replay()
    torch.cuda.current_stream().synchronize()
    with torch.cuda.stream(side if ROUND_TRIP else torch.cuda.current_stream()):  # the host round trip of a CPU-side collective, on its own stream
        if ROUND_TRIP:
            for b, h in zip(bufs, host):
                h.copy_(b, non_blocking=True)
            side.synchronize()
            for h in host:
                h.mul_(0.5)
            for b, h in zip(bufs, host):
                b.copy_(h, non_blocking=True)
            side.synchronize()
    eager = [float(torch.linalg.vector_norm(b)) for b in bufs] if it % 2 else None
    gb.replay()
    torch.cuda.synchronize()
    graph = out.tolist()
    after = [float(torch.linalg.vector_norm(b)) for b in bufs]
    rows.append(dict(iteration=it, eager_norms_before_graph=eager, graph_norms=graph, eager_norms_after=after,
                     stale=[abs(g - a) > 1e-3 * a for g, a in zip(graph, after)]))
    print(rows[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "graph_norm_repro.json"), "w"), indent=1)
