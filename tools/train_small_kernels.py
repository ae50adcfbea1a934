#!/usr/bin/env python3
"""Attribution of the small kernels of one eager training step (cfg 4) to their Python call sites: torch.profiler with
stacks, every CPU op that launches a kernel is charged to the innermost frame inside uni_renderer_amd/.

    python tools/train_small_kernels.py > gpurun_out/train_small_kernels.txt
"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd.optim import FusedAdamW  # noqa: E402
from uni_renderer_amd.train_step import train_step  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    nets = bench.build_models(dev, torch.float32)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    B, L = 4, 64
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    batch = dict(x_t=mk(B, 4, L, L), cond=mk(B, 28, L, L), ehs=mk(B, 77, 768) * 0.5,
                 t_img=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 t_attr=torch.randint(0, 1000, (B,), device=dev, generator=g),
                 target_img=mk(B, 4, L, L), target_attr=mk(B, 28, L, L))
    opt = FusedAdamW([p for m in nets for p in m.parameters()], lr=1e-5)
    for _ in range(2):
        train_step(nets, batch, optimizer=opt, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        train_step(nets, batch, optimizer=opt, dtype=torch.bfloat16)
        torch.cuda.synchronize()
    by_site = collections.defaultdict(lambda: [0, 0.0])
    by_kern = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        top, par = ev, ev.cpu_parent
        chain = [ev.name]
        while par is not None:
            chain.append(par.name)
            top, par = par, par.cpu_parent
        owner = next((c for c in chain if not c.startswith("aten::") and not c.startswith("hip")), chain[-1])
        owner = owner.replace("autograd::engine::evaluate_function: ", "")
        for k in ev.kernels:
            kn = k.name.split("(")[0][-60:]
            by_kern[(owner, kn)][0] += 1
            by_kern[(owner, kn)][1] += k.duration
            by_site[(owner, ev.name)][0] += 1
            by_site[(owner, ev.name)][1] += k.duration
    print("== by (owning autograd node / op, launching aten op): kernels, device us")
    for k, (n, t) in sorted(by_site.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{n:6d} {t:10.0f}  {k[0]:44s} {k[1]}")
    print("== by (owner, kernel)")
    for k, (n, t) in sorted(by_kern.items(), key=lambda kv: -kv[1][1])[:90]:
        print(f"{n:6d} {t:10.0f}  {k[0]:44s} {k[1]}")


if __name__ == "__main__":
    main()
