#!/usr/bin/env python3
"""Captured torch reductions (F.mse_loss / vector_norm, multi-block sizes) vs the same reductions run eagerly after the
replay; with and without eager reductions interleaved between replays.  See tools/graph_norm_repro.py."""
import sys

import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
interleave = "--no-interleave" not in sys.argv
for n in (458752, 65536, 1835008):
    a = [torch.randn(n, device=dev) for _ in range(2)]
    b = [torch.randn(n, device=dev) for _ in range(2)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = torch.stack([F.mse_loss(x, y) for x, y in zip(a, b)])
    for it in range(4):
        for x in a:
            x.normal_()
        if interleave and it % 2:
            _ = [float(F.mse_loss(x, y)) for x, y in zip(a, b)]
        g.replay()
        torch.cuda.synchronize()
        ref = torch.stack([F.mse_loss(x, y) for x, y in zip(a, b)])
        print(dict(n=n, it=it, interleave=interleave, graph=[round(v, 5) for v in out.tolist()], eager=[round(v, 5) for v in ref.tolist()]), flush=True)
