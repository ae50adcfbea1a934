#!/usr/bin/env python3
"""Where the time of ur_wgrad's main loop goes: the kernel with the LDS-DMA copies, the MFMAs, the fragment reads removed,
or the transpose reads replaced by plain 8-byte reads (a library built with `make WGRAD_ABL=1`; UR_WGRAD_ABLATE selects).
--one: a single launch set per variant (for a rocprofv3 --pmc pass)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni_renderer_amd import backward as B_  # noqa: E402
from tools.wgrad_bench import timeit  # noqa: E402

NAMES = {0: "full", 1: "no copies", 2: "no MFMA", 3: "no fragment reads", 4: "plain b64 reads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--one", action="store_true")
    ap.add_argument("--tile", type=int, default=1)
    args = ap.parse_args()
    dt = torch.bfloat16
    mk = lambda *s: torch.randn(*s, device="cuda").to(dt)
    probs = [("linear P1024 N10240 K1280", mk(1024, 10240), mk(1024, 1280), None, 1),
             ("linear P16384 N2560 K320", mk(16384, 2560), mk(16384, 320), None, 8),
             ("conv 4x32x32 C1280 N640", mk(4096, 640), mk(4, 32, 32, 1280), (32, 32, 1), 1)]
    for name, dy, x, conv, sp in probs:
        line = f"{name:28s}"
        for abl in (0, 1, 2, 3, 4):
            os.environ["UR_WGRAD_ABLATE"] = str(abl)
            fn = lambda: B_.wgrad(dy, x, False, conv=conv, tile=args.tile, splits=sp)
            if args.one:
                fn()
                torch.cuda.synchronize()
                continue
            line += f" | {NAMES[abl]} {timeit(fn, 20):7.1f}"
        print(line + " us", flush=True)


if __name__ == "__main__":
    main()
