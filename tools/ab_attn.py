#!/usr/bin/env python3
"""Time ur_attention at the headline self-/cross-attention shapes (HIP events, median of 5 x 20 launches).
Run once per library build (UR_LIB_PATH=...) for a same-box A/B."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uni_renderer_amd import ops  # noqa: E402


def timeit(fn, rounds=5, iters=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


def main():
    dev = torch.device("cuda:0")
    out = []
    only = os.environ.get("AB_ATTN_ONLY")  # e.g. "8,8,4096,4096,40" : one fp16 problem (for PMC runs)
    for dt in ((torch.float16,) if only else (torch.float16, torch.bfloat16)):
        for (B, H, T, Tk, d) in [tuple(int(v) for v in only.split(","))] if only else [(8, 8, 4096, 4096, 40), (8, 8, 1024, 1024, 80), (8, 8, 256, 256, 160),
                                 (8, 8, 4096, 77, 40), (1, 8, 16384, 16384, 40)]:
            C = H * d
            g = torch.Generator(device="cpu").manual_seed(1)
            q = torch.randn(B, T, C, generator=g).to(dev).to(dt)
            k = torch.randn(B, Tk, C, generator=g).to(dev).to(dt)
            v = torch.randn(B, Tk, C, generator=g).to(dev).to(dt)
            Tkp = (Tk + 63) // 64 * 64
            vt = torch.zeros(B, C, Tkp, device=dev, dtype=dt)
            vt[:, :, :Tk] = v.transpose(1, 2)
            kw = {}
            if os.environ.get("AB_ATTN_SCALE0"):  # the step's form: q.k pre-scaled to log2 units by the projections (SLOT kernel)
                f = (d ** -0.5 * 1.4426950408889634) ** 0.5
                q, k = (q.float() * f).to(dt), (k.float() * f).to(dt)
                kw = dict(scale=0.0)
            if os.environ.get("AB_ATTN_HEAD_MAJOR") and d <= 64:  # q / k as [B][H][T][d] images (what the chain kernels write, round 6)
                hm = lambda t, n: t.view(B, n, H, d).permute(0, 2, 1, 3).contiguous().view(B, n, C)
                tm = lambda t, n: t.view(B, H, n, d).permute(0, 2, 1, 3).contiguous().view(B, n, C)
                o_tm = ops.attention(q, k, vt, B=B, H=H, Tq=T, Tk=Tk, d=d, ldq=C, ldk=C, **kw)
                q, k = hm(q, T), hm(k, Tk)
                kw = dict(kw, q_hstride=T * d, k_hstride=Tk * d)
                assert torch.equal(o_tm, ops.attention(q, k, vt, B=B, H=H, Tq=T, Tk=Tk, d=d, ldq=C, ldk=C, **kw))
            o = ops.attention(q, k, vt, B=B, H=H, Tq=T, Tk=Tk, d=d, ldq=C, ldk=C, **kw)
            if "q_hstride" in kw:
                q, k = tm(q, T), tm(k, Tk)  # the reference and the pre-scaled leg below read token matrices
            ref = torch.nn.functional.scaled_dot_product_attention(
                q.view(B, T, H, d).transpose(1, 2).float(), k.view(B, Tk, H, d).transpose(1, 2).float(),
                v.view(B, Tk, H, d).transpose(1, 2).float()).transpose(1, 2).reshape(B, T, C)
            err = float((o.float() - ref).norm() / ref.norm())
            qa, ka = (hm(q, T), hm(k, Tk)) if "q_hstride" in kw else (q, k)
            us = timeit(lambda: ops.attention(qa, ka, vt, B=B, H=H, Tq=T, Tk=Tk, d=d, ldq=C, ldk=C, **kw))
            # the modules' path: scale * log2(e) folded into q by the projection epilogue, kernel called with scale 0
            cs = d ** -0.5 * 1.4426950408889634
            qs = (q.float() * cs).to(dt)
            k0, kw0 = k, {}
            if "q_hstride" in kw:
                qs, k0, kw0 = hm(qs, T), hm(k, Tk), dict(q_hstride=T * d, k_hstride=Tk * d)
            o0 = ops.attention(qs, k0, vt, B=B, H=H, Tq=T, Tk=Tk, d=d, ldq=C, ldk=C, scale=0.0, **kw0)
            err0 = float((o0.float() - ref).norm() / ref.norm())
            us0 = timeit(lambda: ops.attention(qs, k0, vt, B=B, H=H, Tq=T, Tk=Tk, d=d, ldq=C, ldk=C, scale=0.0, **kw0))
            out.append(dict(dtype=str(dt), head_major=bool("q_hstride" in kw), B=B, H=H, T=T, Tk=Tk, d=d, us=round(us, 1), rel_l2=err,
                            tflops=round(4.0 * B * H * T * Tk * d / us / 1e6, 1), us_prescaled=round(us0, 1),
                            rel_l2_prescaled=err0, tflops_prescaled=round(4.0 * B * H * T * Tk * d / us0 / 1e6, 1)))
            print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
