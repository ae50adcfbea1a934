"""ur_tchain (csrc/tchain.hip): the row-local chains after the two attentions of a BasicTransformerBlock at the
320-channel level, one launch each, against a plain fp32 PyTorch restatement of the same operations on the same
(rounded) inputs and against the unfused product path (ops.linear / ops.layernorm) they replace."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

C = 320


def _weights(dev, seed, S):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    W = []
    for _ in range(S):
        W.append(dict(wo=r(C, C, sc=C ** -0.5), bo=r(C, sc=0.1), g=1 + r(C, sc=0.1), b=r(C, sc=0.1), wq=r(C, C, sc=C ** -0.5),
                      w1=r(8 * C, C, sc=C ** -0.5), b1=r(8 * C, sc=0.1), w2=r(C, 4 * C, sc=(4 * C) ** -0.5), b2=r(C, sc=0.1),
                      wpo=r(C, C, sc=C ** -0.5), bpo=r(C, sc=0.1)))
    return W


def _stream(dev, dtype, M, S, seed, hilo):
    from uni_renderer_amd import ops

    g = torch.Generator(device=dev).manual_seed(seed)
    v = torch.randn(S * M, C, device=dev, generator=g) * 1.5
    t = v.to(dtype)
    if hilo:
        t.lo = ops.lo_encode(v - t.float(), dtype)
    return t


def _f(t):
    from uni_renderer_amd import ops

    lo = ops.lo_of(t)
    return t.float() + (ops.lo_float(lo) if lo is not None else 0.0)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1.5e-3), (torch.bfloat16, 1.2e-2)])
@pytest.mark.parametrize("M,S", [(256, 1), (4096, 2), (200, 2)])
def test_chain_q_vs_fp32_and_unfused(dev, dtype, tol, M, S):
    from uni_renderer_amd import ops, tchain
    from uni_renderer_amd.layers import f32, pack_matrix

    W = _weights(dev, 3, S)
    ao = _stream(dev, dtype, M, S, 10, False)
    res = _stream(dev, dtype, M, S, 11, True)
    scale = 40 ** -0.5 * 1.4426950408889634
    packs = [tchain.pack_chain_q(w["wo"], w["bo"], w["g"], w["b"], w["wq"], scale, dtype) for w in W]
    ws = torch.stack([p[0] for p in packs]).contiguous()
    cs = torch.stack([p[1] for p in packs]).contiguous()
    if S == 1:
        ws, cs = ws[0], cs[0]
    y, q = tchain.chain_q(ao, res, ws, cs, 1e-5, streams=S)
    torch.cuda.synchronize()
    for s in range(S):
        w = W[s]
        sl = slice(s * M, (s + 1) * M)
        wo_r, wq_r = w["wo"].to(dtype).float(), (w["wq"] * scale).to(dtype).float()
        y_ref = ao[sl].float() @ wo_r.t() + w["bo"] + _f(res)[sl]
        xn = F.layer_norm(y_ref, (C,), w["g"], w["b"], 1e-5).to(dtype).float()
        q_ref = xn @ wq_r.t()
        ey, eq = rel_l2(_f(y)[sl], y_ref), rel_l2(q[sl], q_ref)
        print(f"tchain_q {dtype} M={M} z={s}: y {ey:.2e} q {eq:.2e}")
        assert ey < (2e-4 if dtype == torch.float16 else 2e-3) and eq < tol, (ey, eq)
    # the unfused product path on the same operands
    wo = torch.stack([pack_matrix(w["wo"], dtype) for w in W])
    bo = torch.stack([f32(w["bo"]) for w in W])
    wq = torch.stack([pack_matrix(w["wq"] * scale, dtype) for w in W])
    gm, bt = torch.stack([f32(w["g"]) for w in W]), torch.stack([f32(w["b"]) for w in W])
    y2 = ops.linear(ao.view(S, M, C), wo, bo, res=ops.view_hilo(res, S, M, C), streams=S, hilo=True)
    q2 = ops.linear(ops.layernorm(y2, gm, bt, 1e-5, streams=S), wq, streams=S)
    assert rel_l2(_f(y), _f(y2).view(S * M, C)) < (1e-4 if dtype == torch.float16 else 1e-3)
    assert rel_l2(q, q2.view(S * M, C)) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1.5e-3), (torch.bfloat16, 1.2e-2)])
@pytest.mark.parametrize("M,S", [(256, 1), (4096, 2), (200, 2)])
def test_chain_ff_vs_fp32(dev, dtype, tol, M, S):
    from uni_renderer_amd import ops, tchain

    W = _weights(dev, 5, S)
    ao = _stream(dev, dtype, M, S, 20, False)
    res = _stream(dev, dtype, M, S, 21, True)
    blk = _stream(dev, dtype, M, S, 22, True)
    packs = [tchain.pack_chain_ff(w["wo"], w["bo"], w["g"], w["b"], w["w1"], w["b1"], w["w2"], w["b2"], w["wpo"], w["bpo"], dtype)
             for w in W]
    ws = torch.stack([p[0] for p in packs]).contiguous()
    cs = torch.stack([p[1] for p in packs]).contiguous()
    if S == 1:
        ws, cs = ws[0], cs[0]
    out = tchain.chain_ff(ao, res, blk, ws, cs, 1e-5, streams=S)
    torch.cuda.synchronize()
    for s in range(S):
        w = W[s]
        sl = slice(s * M, (s + 1) * M)
        r = lambda t: t.to(dtype).float()
        y = ao[sl].float() @ r(w["wo"]).t() + w["bo"] + _f(res)[sl]
        xn = r(F.layer_norm(y, (C,), w["g"], w["b"], 1e-5))
        hcat = xn @ r(w["w1"]).t() + w["b1"]
        h = r(hcat[:, : 4 * C] * F.gelu(hcat[:, 4 * C:]))
        y3 = y + h @ r(w["w2"]).t() + w["b2"]
        ref = r(y3) @ r(w["wpo"]).t() + w["bpo"] + _f(blk)[sl]
        e = rel_l2(_f(out)[sl], ref)
        print(f"tchain_ff {dtype} M={M} z={s}: out {e:.2e}")
        assert e < tol, e
        assert rel_l2(out[sl], ref) < tol  # the hi part alone is the ordinary rounded tensor


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1.5e-3), (torch.bfloat16, 1.2e-2)])
@pytest.mark.parametrize("B,T,S", [(2, 128, 1), (2, 1024, 2), (3, 96, 2)])
def test_chain_pre_vs_fp32(dev, dtype, tol, B, T, S):
    """proj_in + LayerNorm1 + q / k / V^T projections in one launch (UR_TCHAIN_PRE)."""
    from uni_renderer_amd import tchain

    W = _weights(dev, 7, S)
    g = torch.Generator(device=dev).manual_seed(30)
    for w in W:
        w["wk"] = torch.randn(C, C, device=dev, generator=g) * C ** -0.5
        w["wv"] = torch.randn(C, C, device=dev, generator=g) * C ** -0.5
    M = B * T
    h0 = _stream(dev, dtype, M, S, 31, False)
    sc = math.sqrt(40 ** -0.5 * 1.4426950408889634)
    packs = [tchain.pack_chain_pre(w["wo"].view(C, C, 1, 1), w["bo"], w["g"], w["b"], w["wq"], w["wk"], w["wv"], sc, dtype) for w in W]
    ws = torch.stack([p[0] for p in packs]).contiguous()
    cs = torch.stack([p[1] for p in packs]).contiguous()
    if S == 1:
        ws, cs = ws[0], cs[0]
    y, q, k, vt = tchain.chain_pre(h0, ws, cs, 1e-5, tokens_per_sample=T, streams=S)
    torch.cuda.synchronize()
    Tpad = (T + 63) // 64 * 64
    assert vt.shape == (S * B, C, Tpad)
    r = lambda t: t.to(dtype).float()
    for s in range(S):
        w = W[s]
        sl = slice(s * M, (s + 1) * M)
        y_ref = h0[sl].float() @ r(w["wo"]).t() + w["bo"]
        xn = r(F.layer_norm(y_ref, (C,), w["g"], w["b"], 1e-5))
        q_ref, k_ref, v_ref = xn @ r(w["wq"] * sc).t(), xn @ r(w["wk"] * sc).t(), xn @ r(w["wv"]).t()
        vt_ref = v_ref.view(B, T, C).transpose(1, 2)
        errs = dict(y=rel_l2(_f(y)[sl], y_ref), q=rel_l2(q[sl], q_ref), k=rel_l2(k[sl], k_ref),
                    vt=rel_l2(vt[s * B:(s + 1) * B, :, :T], vt_ref))
        print(f"tchain_pre {dtype} B={B} T={T} z={s}: {errs}")
        assert errs["y"] < (2e-4 if dtype == torch.float16 else 2e-3) and max(errs["q"], errs["k"], errs["vt"]) < tol, errs
        if Tpad != T:
            assert float(vt[s * B:(s + 1) * B, :, T:].abs().max()) == 0.0


def test_chain_pre_is_reproducible_at_the_benchmarked_size(dev):
    """Regression: the V^T staging of UR_TCHAIN_PRE used to start while slower waves were still reading the last weight
    stage out of the same LDS bytes (ring slot 1 = the staging slices of waves 0 / 1): ~1 launch in 300 stored a few wrong
    V^T channels for one wave, i.e. one grouped step in seven differed from the next.  400 launches at the benchmarked size
    (2 streams x 4 x 4096 rows) must be bit-identical."""
    from uni_renderer_amd import ops, tchain

    dtype, S, B, T = torch.float16, 2, 4, 4096
    W = _weights(dev, 7, S)
    g = torch.Generator(device=dev).manual_seed(30)
    for w in W:
        w["wk"] = torch.randn(C, C, device=dev, generator=g) * C ** -0.5
        w["wv"] = torch.randn(C, C, device=dev, generator=g) * C ** -0.5
    h0 = _stream(dev, dtype, B * T, S, 31, False)
    packs = [tchain.pack_chain_pre(w["wo"].view(C, C, 1, 1), w["bo"], w["g"], w["b"], w["wq"], w["wk"], w["wv"], 0.4777, dtype) for w in W]
    ws = torch.stack([p[0] for p in packs]).contiguous()
    cs = torch.stack([p[1] for p in packs]).contiguous()

    def run():
        y, q, k, vt = tchain.chain_pre(h0, ws, cs, 1e-5, tokens_per_sample=T, streams=S)
        return [y, ops.lo_of(y), q, k, vt]

    ref = [t.clone() for t in run()]
    for i in range(400):
        cur = run()
        for name, a, b in zip(("y", "y.lo", "q", "k", "vt"), cur, ref):
            assert torch.equal(a, b), f"launch {i}: {name} differs from the first launch"


def test_chain_rejects_other_widths(dev):
    from uni_renderer_amd import tchain

    x = torch.zeros(128, 640, device=dev, dtype=torch.float16)
    assert not tchain.supported(x)
    with pytest.raises(RuntimeError):
        tchain.chain_q(x, x, torch.zeros(10 * 20480, device=dev, dtype=torch.float16), torch.zeros(960, device=dev), 1e-5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,T,S", [(2, 128, 1), (2, 1024, 2), (3, 96, 2)])
def test_head_major_q_k_are_the_token_matrices_permuted_and_attention_reads_them_bit_for_bit(dev, dtype, B, T, S):
    """Round 6 (ur_tchain_desc.qk_heads, ur_attn_desc.q_hstride / k_hstride): the chains hand q / k to the d = 40 attention as
    [sample][head][token][40] images.  Same values, other addresses: the images equal the token matrices permuted, and the
    attention over them returns the same bits as over the token matrices (self-attention with both operands head-major; the
    cross-attention form with only q head-major and ragged 77 keys)."""
    from uni_renderer_amd import ops, tchain

    H, d = 8, 40
    W = _weights(dev, 7, S)
    g = torch.Generator(device=dev).manual_seed(30)
    for w in W:
        w["wk"] = torch.randn(C, C, device=dev, generator=g) * C ** -0.5
        w["wv"] = torch.randn(C, C, device=dev, generator=g) * C ** -0.5
    M = B * T
    h0 = _stream(dev, dtype, M, S, 31, False)
    sc = math.sqrt(40 ** -0.5 * 1.4426950408889634)
    packs = [tchain.pack_chain_pre(w["wo"].view(C, C, 1, 1), w["bo"], w["g"], w["b"], w["wq"], w["wk"], w["wv"], sc, dtype) for w in W]
    ws = torch.stack([p[0] for p in packs]).contiguous()
    cs = torch.stack([p[1] for p in packs]).contiguous()
    if S == 1:
        ws, cs = ws[0], cs[0]
    y0, q0, k0, vt0 = tchain.chain_pre(h0, ws, cs, 1e-5, tokens_per_sample=T, streams=S)
    y1, q1, k1, vt1 = tchain.chain_pre(h0, ws, cs, 1e-5, tokens_per_sample=T, streams=S, head_major=True)
    hm = lambda t: t.view(S * B, T, H, d).permute(0, 2, 1, 3).contiguous().view(S * M, C)
    assert torch.equal(y0, y1) and torch.equal(ops.lo_of(y0), ops.lo_of(y1)) and torch.equal(vt0, vt1)
    assert torch.equal(q1, hm(q0)) and torch.equal(k1, hm(k0))
    kw = dict(B=S * B, H=H, Tq=T, Tk=T, d=d, ldq=C, ldk=C, scale=0.0)
    o0 = ops.attention(q0, k0, vt0, **kw)
    o1 = ops.attention(q1, k1, vt1, q_hstride=T * d, k_hstride=T * d, **kw)
    assert torch.equal(o0, o1)
    # chain_q: the cross-attention query
    packs = [tchain.pack_chain_q(w["wo"], w["bo"], w["g"], w["b"], w["wq"], sc * sc, dtype) for w in W]
    wsq = torch.stack([p[0] for p in packs]).contiguous()
    csq = torch.stack([p[1] for p in packs]).contiguous()
    if S == 1:
        wsq, csq = wsq[0], csq[0]
    res = _stream(dev, dtype, M, S, 11, True)
    ya, qa = tchain.chain_q(o0.view(S * M, C), res, wsq, csq, 1e-5, streams=S)
    yb, qb = tchain.chain_q(o0.view(S * M, C), res, wsq, csq, 1e-5, streams=S, head_major_tokens=T)
    assert torch.equal(ya, yb) and torch.equal(ops.lo_of(ya), ops.lo_of(yb)) and torch.equal(qb, hm(qa))
    Tk = 77
    kc = (torch.randn(S * B, Tk, C, device=dev, generator=g) * 0.5).to(dtype)
    vtc = torch.zeros(S * B, C, 128, device=dev, dtype=dtype)
    vtc[:, :, :Tk] = torch.randn(S * B, C, Tk, device=dev, generator=g).to(dtype)
    kx = dict(B=S * B, H=H, Tq=T, Tk=Tk, d=d, ldq=C, ldk=C, scale=0.0)
    assert torch.equal(ops.attention(qa, kc, vtc, **kx), ops.attention(qb, kc, vtc, q_hstride=T * d, **kx))


def test_head_major_images_are_refused_where_they_are_not_built(dev):
    from uni_renderer_amd import ops, tchain

    B, T, H, d = 1, 128, 8, 80  # the d = 80 kernel reads token matrices only
    q = torch.zeros(B, T, H * d, device=dev, dtype=torch.float16)
    vt = torch.zeros(B, H * d, T, device=dev, dtype=torch.float16)
    with pytest.raises(RuntimeError):
        ops.attention(q, q, vt, B=B, H=H, Tq=T, Tk=T, d=d, ldq=H * d, ldk=H * d, q_hstride=T * d, k_hstride=T * d)
    x = torch.zeros(2 * 100, C, device=dev, dtype=torch.float16)  # 100 tokens per sample: a wave would straddle two samples
    with pytest.raises(RuntimeError):
        tchain.chain_q(x, x, torch.zeros(10 * 20480, device=dev, dtype=torch.float16), torch.zeros(960, device=dev), 1e-5,
                       head_major_tokens=100)
