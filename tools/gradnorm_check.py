#!/usr/bin/env python3
"""The clipping norm from the per-launch sums of squares (backward.GradSquares) against torch's norm over the same
gradients, after one backward of the cfg-4 step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd import backward as B  # noqa: E402
from uni_renderer_amd import train_step as TS  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    nets = bench.build_models(dev, torch.float32)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    Bn, L = 4, 64
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    batch = dict(x_t=mk(Bn, 4, L, L), cond=mk(Bn, 28, L, L), ehs=mk(Bn, 77, 768) * 0.5,
                 t_img=torch.randint(0, 1000, (Bn,), device=dev, generator=g),
                 t_attr=torch.randint(0, 1000, (Bn,), device=dev, generator=g),
                 target_img=mk(Bn, 4, L, L), target_attr=mk(Bn, 28, L, L))
    TS._forward_backward(nets, batch, None, None, torch.bfloat16, None)
    params = [p for n in nets for p in n.parameters() if p.grad is not None]
    ref = torch.linalg.vector_norm(torch.stack(torch._foreach_norm([p.grad for p in params])))
    ref64 = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params))
    gs = B.grad_squares
    cov = gs.count
    rest = [p.grad for p in params if id(p) not in cov]
    sq = torch.cat(gs.parts).sum() + (torch.stack(torch._foreach_norm(rest)).square().sum() if rest else 0.0)
    sq64 = torch.cat(gs.parts).double().sum() + sum((t.double() ** 2).sum() for t in rest)
    covered_ref = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params if id(p) in cov))
    print(f"usable {gs.usable()}, covered {len(cov)} of {len(params)} tensors, "
          f"{sum(p.numel() for p in params if id(p) in cov) / sum(p.numel() for p in params):.4f} of the elements")
    print(f"torch norm {float(ref):.8f}  fp64 {float(ref64):.8f}  fused {float(sq.sqrt()):.8f}  fused(fp64 sum of the partials) {float(sq64.sqrt()):.8f}")
    print(f"covered part: fp64 over p.grad {float(covered_ref):.8f}  vs sqrt(sum partials) {float(torch.cat(gs.parts).double().sum().sqrt()):.8f}")


if __name__ == "__main__":
    main()
