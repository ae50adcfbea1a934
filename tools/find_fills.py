#!/usr/bin/env python3
"""Which Python lines of one eager training step launch the small elementwise kernels (fill / zero / add / copy): torch.profiler
with stacks over train_step on the cfg 4 shapes, grouped by the innermost frame of this repository."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd.train_step import train_step  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    nets = bench.build_models(dev, torch.float32)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    B, L = 4, 64
    batch = dict(x_t=mk(B, 4, L, L), cond=mk(B, 28, L, L), ehs=mk(B, 77, 768) * 0.5,
                 t_img=torch.randint(0, 1000, (B,), device=dev, generator=g), t_attr=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 target_img=mk(B, 4, L, L), target_attr=mk(B, 28, L, L))
    from uni_renderer_amd.optim import FusedAdamW
    opt = FusedAdamW([p for m in nets for p in m.parameters()], lr=1e-5)
    train_step(nets, batch, optimizer=opt, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode

    by = collections.Counter()

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if any(k in name for k in ("fill", "zero", "aten.add", "aten.copy", "aten.cat", "aten.mul", "aten.sum", "aten.clone", "_to_copy", "aten.div")):
                frames = [f for f in traceback.extract_stack()
                          if ("/uni_renderer_amd/" in f.filename or "/tools/" in f.filename) and not f.filename.endswith("find_fills.py")]
                fr = frames[-1] if frames else None
                by[(name, f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line}"[:120] if fr else "?")] += 1
            return func(*args, **(kwargs or {}))

    with Spy():
        train_step(nets, batch, optimizer=opt, dtype=torch.bfloat16)
        torch.cuda.synchronize()
    for (name, frame), n in by.most_common(50):
        print(f"{n:5d}  {name:28s} {frame}")


if __name__ == "__main__":
    main()
