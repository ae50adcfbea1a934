#!/usr/bin/env python3
"""cProfile of the eager training step (host-bound): where the Python time goes.  -> gpurun_out/host_profile.txt"""
import cProfile, io, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from uni_renderer_amd.optim import FusedAdamW
from uni_renderer_amd.train_step import train_step

dev = torch.device("cuda:0")
nets = bench.build_models(dev, torch.float32)
for m in nets:
    m.train(); m.requires_grad_(True)
g = torch.Generator(device=dev).manual_seed(7)
mk = lambda *s: torch.randn(*s, device=dev, generator=g)
B, L = 4, 64
batch = dict(x_t=mk(B, 4, L, L), cond=mk(B, 28, L, L), ehs=mk(B, 77, 768) * 0.5, t_img=torch.randint(0, 1000, (B,), device=dev, generator=g),
             t_attr=torch.randint(0, 1000, (B,), device=dev, generator=g), target_img=mk(B, 4, L, L), target_attr=mk(B, 28, L, L))
opt = FusedAdamW([p for m in nets for p in m.parameters()], lr=1e-5)
for _ in range(2):
    train_step(nets, batch, optimizer=opt, dtype=torch.bfloat16)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    train_step(nets, batch, optimizer=opt, dtype=torch.bfloat16)
torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(40)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "host_profile.txt"), "w").write(out.getvalue())
