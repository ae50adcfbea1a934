#!/usr/bin/env python3
"""Find the first op of the grouped step whose output depends on uninitialised memory: every torch.empty of the package
is filled with a byte pattern, the outputs of the op wrappers are checksummed in call order, two patterns are compared."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd import ops, tchain, fused  # noqa: E402

PATTERN = [0]
_empty = torch.empty


def poisoned_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.numel():
        t.view(torch.uint8).fill_(PATTERN[0]) if t.is_contiguous() else None
    return t


LOG = []


def csum(t):
    if isinstance(t, (tuple, list)):
        return tuple(csum(u) for u in t)
    if not torch.is_tensor(t):
        return None
    v = t.float()
    lo = ops.lo_of(t)
    return (float(v.double().abs().sum()), float(ops.lo_float(lo).double().abs().sum()) if lo is not None else None)


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        r = f(*a, **k)
        LOG.append((f"{mod.__name__.split('.')[-1]}.{name}", tuple(getattr(x, 'shape', None) and tuple(x.shape) for x in a if torch.is_tensor(x))[:2], csum(r)))
        return r
    setattr(mod, name, g)


def main():
    dev = torch.device("cuda", 0)
    unet, enc, dec = bench.build_models(dev, torch.float16)
    B, L = 4, 64
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B, 4, L, L, device=dev, generator=g).half()
    c = torch.randn(B, 28, L, L, device=dev, generator=g).half()
    ehs = (torch.randn(B, 77, 768, device=dev, generator=g) * 0.5).half()
    ti = torch.randint(0, 1000, (B,), device=dev, generator=g)
    ta = torch.randint(0, 1000, (B,), device=dev, generator=g)
    step = fused.GroupedDualStreamStep(unet, enc, dec)
    with torch.no_grad():
        step(x, c, ehs, ti, ta)  # builds the packs with the real torch.empty
    torch.empty = poisoned_empty
    for m, names in ((ops, ["conv3x3", "linear", "groupnorm", "layernorm", "attention", "vt_proj", "add", "igemm"]),
                     (tchain, ["chain_pre", "chain_q", "chain_ff"])):
        for n in names:
            if hasattr(m, n):
                wrap(m, n)
    logs = []
    for pat in (0x00, 0x7B):
        PATTERN[0] = pat
        LOG.clear()
        with torch.no_grad():
            out = step(x, c, ehs, ti, ta)
        logs.append(list(LOG) + [("final", None, csum((out["img_pred"], out["attr_pred"])))])
    a, b = logs
    print(len(a), len(b))
    n = 0
    for i, (ea, eb) in enumerate(zip(a, b)):
        if ea[2] != eb[2]:
            print("DIFF", i, ea[0], ea[1], ea[2], eb[2])
            n += 1
            if n > 12:
                break
    if n == 0:
        print("no op output differs")


if __name__ == "__main__":
    main()
