#!/usr/bin/env python3
"""DETERMINISM of the row-local chains: every mode N times on the same inputs, outputs compared bitwise with the first run.
(derived from tools/tchain_bench.py) -- Time the row-local chains (ur_tchain) against the launches they replace, at the 64x64 level of the headline step
(M = 4 x 4096 rows per stream, 2 streams grouped): HIP events around graph replays of each variant.

    python tools/tchain_bench.py [--iters 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--M", type=int, default=16384)
    ap.add_argument("--S", type=int, default=2)
    ap.add_argument("--dtype", default="fp16")
    args = ap.parse_args()
    from uni_renderer_amd import ops, tchain
    from uni_renderer_amd.layers import f32, geglu_perm, pack_matrix

    EPS = 1e-5
    dev = torch.device("cuda:0")
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    C, M, S = 320, args.M, args.S
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    W = [dict(wo=r(C, C, sc=C ** -0.5), bo=r(C, sc=0.1), g=1 + r(C, sc=0.1), b=r(C, sc=0.1), wq=r(C, C, sc=C ** -0.5),
              w1=r(8 * C, C, sc=C ** -0.5), b1=r(8 * C, sc=0.1), w2=r(C, 4 * C, sc=(4 * C) ** -0.5), b2=r(C, sc=0.1),
              wpo=r(C, C, sc=C ** -0.5), bpo=r(C, sc=0.1)) for _ in range(S)]

    def stream(hilo):
        v = r(S, M, C, sc=1.5)
        t = v.to(dt)
        if hilo:
            t.lo = ops.lo_encode(v - t.float(), dt)
        return t

    ao, res, blk = stream(False), stream(True), stream(True)
    pq = [tchain.pack_chain_q(w["wo"], w["bo"], w["g"], w["b"], w["wq"], 0.228, dt) for w in W]
    pf = [tchain.pack_chain_ff(w["wo"], w["bo"], w["g"], w["b"], w["w1"], w["b1"], w["w2"], w["b2"], w["wpo"], w["bpo"], dt) for w in W]
    wsq, csq = torch.stack([p[0] for p in pq]).contiguous(), torch.stack([p[1] for p in pq]).contiguous()
    wsf, csf = torch.stack([p[0] for p in pf]).contiguous(), torch.stack([p[1] for p in pf]).contiguous()
    stk = lambda k, fn: torch.stack([fn(w[k]) for w in W]).contiguous()
    wo, bo, wq = stk("wo", lambda t: pack_matrix(t, dt)), stk("bo", f32), stk("wq", lambda t: pack_matrix(t * 0.228, dt))
    gm, bt = stk("g", f32), stk("b", f32)
    perm = geglu_perm(4 * C, dev)
    w1, b1 = stk("w1", lambda t: pack_matrix(t, dt)[perm]), stk("b1", lambda t: f32(t)[perm])
    w2, b2 = stk("w2", lambda t: pack_matrix(t, dt)), stk("b2", f32)
    wpo, bpo = stk("wpo", lambda t: pack_matrix(t, dt)), stk("bpo", f32)

    def unfused_q():
        y = ops.linear(ao, wo, bo, res=res, streams=S, hilo=True)
        return ops.linear(ops.layernorm(y, gm, bt, 1e-5, streams=S), wq, streams=S)

    def unfused_ff():
        y = ops.linear(ao, wo, bo, res=res, streams=S, hilo=True)
        xn = ops.layernorm(y, gm, bt, 1e-5, streams=S)
        gg = ops.linear(xn, w1, b1, act=ops.ACT_GEGLU, streams=S)
        y3 = ops.linear(gg, w2, b2, res=y, streams=S, hilo=True)
        return ops.linear(y3, wpo, bpo, res=blk, streams=S, hilo=True)

    def fused_q():
        return tchain.chain_q(ao.view(S * M, C), ops.view_hilo(res, S * M, C), wsq, csq, EPS, streams=S)

    def fused_ff():
        return tchain.chain_ff(ao.view(S * M, C), ops.view_hilo(res, S * M, C), ops.view_hilo(blk, S * M, C), wsf, csf, EPS, streams=S)

    out = {}
    Tn = 4096 if M % 4096 == 0 else M
    wk_, wv_ = [r(C, C, sc=C ** -0.5) for _ in range(S)], [r(C, C, sc=C ** -0.5) for _ in range(S)]
    pp = [tchain.pack_chain_pre(w["wo"].view(C, C, 1, 1), w["bo"], w["g"], w["b"], w["wq"], wk_[i], wv_[i], 0.4777, dt) for i, w in enumerate(W)]
    wsp, csp = torch.stack([p[0] for p in pp]).contiguous(), torch.stack([p[1] for p in pp]).contiguous()
    wqk = torch.stack([torch.cat([pack_matrix(w["wq"], dt), pack_matrix(wk_[i], dt)], 0) for i, w in enumerate(W)]).contiguous()
    wvv = torch.stack([pack_matrix(t, dt) for t in wv_]).contiguous()

    def unfused_pre():
        y = ops.linear(ao, wo, bo, streams=S, hilo=True)
        xn = ops.layernorm(y, gm, bt, 1e-5, streams=S)
        vt = ops.vt_proj(xn.view(S * M // Tn, Tn, C), wvv, streams=S)
        return ops.linear(xn, wqk, streams=S, out_scale=0.4777), vt

    def fused_pre():
        return tchain.chain_pre(ao.view(S * M, C), wsp, csp, 1e-5, tokens_per_sample=Tn, streams=S)


    def flat(o):
        ts = o if isinstance(o, (tuple, list)) else [o]
        res_ = []
        for t in ts:
            res_.append(t)
            if ops.lo_of(t) is not None:
                res_.append(t.lo)
        return res_

    for name, fn in (("fused_pre", fused_pre), ("fused_q", fused_q), ("fused_ff", fused_ff)):
        ref = [t.clone() for t in flat(fn())]
        bad = 0
        first = None
        for i in range(args.iters):
            cur = flat(fn())
            diff = [j for j, (x_, y_) in enumerate(zip(cur, ref)) if not torch.equal(x_, y_)]
            if diff:
                bad += 1
                if first is None:
                    j = diff[0]
                    d = (cur[j].float() - ref[j].float()).reshape(-1, cur[j].shape[-1]) if cur[j].dim() >= 2 else None
                    rows = torch.nonzero(d.abs().sum(1)).flatten() if d is not None else None
                    cols = torch.nonzero(d.abs().sum(0)).flatten() if d is not None else None
                    first = dict(run=i, outputs=diff, rows=(rows[:8].tolist(), int(rows.numel())) if rows is not None else None,
                                 cols=(cols[:8].tolist(), int(cols.numel())) if cols is not None else None,
                                 shape=list(cur[j].shape))
        print(name, f"{bad} of {args.iters} runs differ", first, flush=True)


if __name__ == "__main__":
    main()
