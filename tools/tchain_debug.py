#!/usr/bin/env python3
"""Diagnostics for ur_tchain (run on the GPU box): isolate I/O arrangement, bias, weight images, hand-off."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni_renderer_amd import ops, tchain

dev = torch.device("cuda:0")
dt = torch.float16
C, M = 320, 128
torch.manual_seed(0)
z = lambda *s: torch.zeros(*s, device=dev)
eye = torch.eye(C, device=dev)

def run_q(wo, bo, g, b, wq, ao, res):
    ws, cs = tchain.pack_chain_q(wo, bo, g, b, wq, 1.0, dt)
    y, q = tchain.chain_q(ao, res, ws, cs, 1e-5, streams=1, hilo=False)
    torch.cuda.synchronize()
    return y.float(), q.float()

def show(name, got, want):
    err = (got - want).abs().max().item()
    print(f"{name}: max abs err {err:.4g}  (|want| max {want.abs().max().item():.3g})")
    if err > 1e-2:
        bad = (got - want).abs() > 1e-2
        rows = bad.any(1).nonzero().flatten()[:6].tolist()
        cols = bad.any(0).nonzero().flatten()[:24].tolist()
        print("   bad rows", rows, "... bad cols", cols, " frac bad", bad.float().mean().item())
        r0 = rows[0] if rows else 0
        print("   got [r0,:24]", [round(v, 2) for v in got[r0, :24].tolist()])
        print("   want[r0,:24]", [round(v, 2) for v in want[r0, :24].tolist()])

chan = torch.arange(C, device=dev).float()
rowv = torch.arange(M, device=dev).float()
# 1. W = 0, bias = 0: y must equal res (I/O arrangement in and out)
res = (rowv[:, None] * 0.01 + chan[None, :] * 1.0).to(dt)   # distinct value per (row, channel) at fp16 resolution? use small ints
res = ((rowv[:, None] % 8) * 400 + chan[None, :]).to(dt)
ao = z(M, C).to(dt)
y, q = run_q(z(C, C), z(C), torch.ones(C, device=dev), z(C), z(C, C), ao, res)
show("1 y=res (I/O)", y, res.float())
# 2. bias only
y, q = run_q(z(C, C), chan.clone(), torch.ones(C, device=dev), z(C), z(C, C), ao, z(M, C).to(dt))
show("2 y=bias", y, chan[None, :].expand(M, C))
# 3. W = I: y = ao
ao = ((rowv[:, None] % 4) * 400 + chan[None, :]).to(dt)
y, q = run_q(eye, z(C), torch.ones(C, device=dev), z(C), z(C, C), ao, z(M, C).to(dt))
show("3 y=ao (identity W0)", y, ao.float())
# 4. random W0
W0 = torch.randn(C, C, device=dev) * C ** -0.5
ao = torch.randn(M, C, device=dev).to(dt)
y, q = run_q(W0, z(C), torch.ones(C, device=dev), z(C), z(C, C), ao, z(M, C).to(dt))
show("4 y=ao W0^T", y, ao.float() @ W0.to(dt).float().t())
# 5. LN + identity Wq: q = LN(y) (hand-off + KPERM)
y, q = run_q(W0, z(C), torch.ones(C, device=dev), z(C), eye, ao, z(M, C).to(dt))
yr = ao.float() @ W0.to(dt).float().t()
show("5 q=LN(y) (identity Wq)", q, torch.nn.functional.layer_norm(yr, (C,)))
# 6. gamma/beta
gam, bet = 1 + 0.01 * chan, 0.1 * chan
y, q = run_q(W0, z(C), gam, bet, eye, ao, z(M, C).to(dt))
show("6 q=LN affine", q, torch.nn.functional.layer_norm(yr, (C,), gam, bet))
