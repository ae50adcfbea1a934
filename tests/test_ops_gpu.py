"""GPU parity of every C-ABI device op against a plain PyTorch fp32 reference of the same op evaluated on
the CPU from the same (fp16/bf16-rounded) inputs.  Tolerances (rel-L2): fp16 2e-3, bf16 1.2e-2 -- one
storage rounding of the output (2^-11 / 2^-8 relative) on top of fp32 accumulation."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 2e-3, torch.bfloat16: 1.2e-2}
DTYPES = [torch.float16, torch.bfloat16]
PP_TILES = list(range(49, 56))  # 8-wave ping-pong builds (csrc/igemm_pp.hip)
ALL_TILES = [t for t in range(1, 47) if t != 39] + PP_TILES + list(range(56, 62))  # 56-61: round-6 big-wave-tile builds


def _need(tile):
    """Skip the cases of an opt-in kernel build (csrc/Makefile: PP=1) the loaded library does not contain."""
    from uni_renderer_amd import ops
    if tile in PP_TILES and not ops.pp_built():
        pytest.skip("ping-pong tiles: opt-in build (make PP=1)")



def _rand(shape, dtype, dev, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", ALL_TILES)
@pytest.mark.parametrize("shape", [(256, 320, 320), (300, 64, 128), (1000, 448, 640), (4, 1280, 320)])
def test_linear_bias_res(dev, dtype, tile, shape):
    from uni_renderer_amd import ops
    _need(tile)
    M, N, K = shape
    x = _rand((M, K), dtype, dev, seed=1)
    w = _rand((N, K), dtype, dev, 1 / math.sqrt(K), seed=2)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3)).to(dev)
    r = _rand((M, N), dtype, dev, seed=4)
    y = ops.linear(x, w, b, res=r, tile=tile, splitk=1)
    ref = x.float().cpu() @ w.float().cpu().t() + b.cpu() + r.float().cpu()
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("splitk", [2, 4])
def test_linear_splitk_silu_scale(dev, dtype, splitk):
    from uni_renderer_amd import ops
    M, N, K = 200, 320, 2048
    x = _rand((M, K), dtype, dev, seed=1)
    w = _rand((N, K), dtype, dev, 1 / math.sqrt(K), seed=2)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3)).to(dev)
    y = ops.linear(x, w, b, act=ops.ACT_SILU, out_scale=0.5, tile=3, splitk=splitk)
    ref = F.silu(x.float().cpu() @ w.float().cpu().t() + b.cpu()) * 0.5
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_two_sources(dev, dtype):
    """1x1 shortcut over cat(hidden, skip): K walks two tensors."""
    from uni_renderer_amd import ops
    M, N = 520, 128
    x0, x1 = _rand((M, 128), dtype, dev, seed=1), _rand((M, 64), dtype, dev, seed=2)
    w = _rand((N, 192), dtype, dev, 0.1, seed=3)
    y = ops.linear(x0, w, None, x1=x1)
    ref = torch.cat([x0, x1], -1).float().cpu() @ w.float().cpu().t()
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", ALL_TILES)
def test_geglu(dev, dtype, tile):
    from uni_renderer_amd import ops
    _need(tile)
    from uni_renderer_amd.layers import geglu_perm
    M, K, NH = 300, 128, 512
    x = _rand((M, K), dtype, dev, seed=1)
    w = _rand((2 * NH, K), dtype, dev, 0.1, seed=2)
    b = torch.randn(2 * NH, generator=torch.Generator().manual_seed(3)).to(dev)
    perm = geglu_perm(NH, dev)
    y = ops.linear(x, w[perm].contiguous(), b[perm].contiguous(), act=ops.ACT_GEGLU, tile=tile, splitk=1)
    full = x.float().cpu() @ w.float().cpu().t() + b.cpu()
    ref = full[:, :NH] * F.gelu(full[:, NH:])
    assert y.shape == (M, NH)
    assert rel_l2(y, ref) < TOL[dtype]


def _conv_ref(x_nhwc, w_oihw, b, stride=1, ups=False):
    x = x_nhwc.float().cpu().permute(0, 3, 1, 2)
    if ups:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    y = F.conv2d(x, w_oihw.float().cpu(), None if b is None else b.cpu(), stride=stride, padding=1)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", ALL_TILES)
@pytest.mark.parametrize("mode", ["s1", "s2", "ups"])
def test_conv3x3(dev, dtype, tile, mode):
    from uni_renderer_amd import ops
    _need(tile)
    from uni_renderer_amd.layers import pack_conv3x3
    B, H, W, Ci, Co = 2, 12, 10, 128, 192
    x = _rand((B, H, W, Ci), dtype, dev, seed=1)
    w = _rand((Co, Ci, 3, 3), dtype, dev, 1 / math.sqrt(9 * Ci), seed=2)
    b = torch.randn(Co, generator=torch.Generator().manual_seed(3)).to(dev)
    stride, ups = (2, False) if mode == "s2" else (1, mode == "ups")
    y = ops.conv3x3(x, pack_conv3x3(w, dtype), b, stride=stride, ups=ups, tile=tile, splitk=1)
    ref = _conv_ref(x, w, b, stride, ups)
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("splitk", [1, 3])
def test_conv3x3_concat_temb_res(dev, dtype, splitk):
    """The up-path resnet conv: two sources, +bias, + per-sample time embedding, + residual, scale."""
    from uni_renderer_amd import ops
    from uni_renderer_amd.layers import pack_conv3x3
    B, H, W, C0, C1, Co = 3, 8, 8, 128, 64, 64
    x0, x1 = _rand((B, H, W, C0), dtype, dev, seed=1), _rand((B, H, W, C1), dtype, dev, seed=2)
    w = _rand((Co, C0 + C1, 3, 3), dtype, dev, 0.03, seed=3)
    b = torch.randn(Co, generator=torch.Generator().manual_seed(4)).to(dev)
    temb_all = _rand((B, 256), dtype, dev, seed=5)
    res = _rand((B, H, W, Co), dtype, dev, seed=6)
    y = ops.conv3x3(x0, pack_conv3x3(w, dtype), b, x1=x1, rowadd=temb_all[:, 64:128], res=res, out_scale=0.5,
                    tile=3, splitk=splitk)
    ref = _conv_ref(torch.cat([x0, x1], -1), w, b) + temb_all[:, 64:128].float().cpu()[:, None, None, :]
    ref = (ref + res.float().cpu()) * 0.5
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout", [(4, 64), (28, 128), (64, 4), (128, 28)])
def test_conv_in_out_odd_channels(dev, dtype, cin, cout):
    from uni_renderer_amd import ops
    from uni_renderer_amd.layers import pack_conv3x3
    B, H, W = 2, 16, 16
    w = _rand((cout, cin, 3, 3), dtype, dev, 0.1, seed=2)
    b = torch.randn(cout, generator=torch.Generator().manual_seed(3)).to(dev)
    if cin < 64:
        x_nchw = torch.randn(B, cin, H, W, generator=torch.Generator().manual_seed(1)).to(dev)  # fp32 NCHW input
        x = ops.to_nhwc(x_nchw, dtype, 64)
        assert x.shape == (B, H, W, 64) and float(x[..., cin:].abs().max()) == 0.0
        y = ops.conv3x3(x, pack_conv3x3(w, dtype, 64), b)
        ref = _conv_ref(x[..., :cin], w, b)
    else:
        x = _rand((B, H, W, cin), dtype, dev, seed=1)
        y = ops.conv3x3(x, pack_conv3x3(w, dtype), b, n_out=cout)
        ref = _conv_ref(x, w, b)
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", [4, 77, 256, 1000])
def test_vt_proj(dev, dtype, T):
    from uni_renderer_amd import ops
    B, Cin, Cout = 2, 128, 192
    x = _rand((B, T, Cin), dtype, dev, seed=1)
    wv = _rand((Cout, Cin), dtype, dev, 0.1, seed=2)
    vt = ops.vt_proj(x, wv)
    Tpad = (T + 63) // 64 * 64
    assert vt.shape == (B, Cout, Tpad)
    ref = torch.einsum("oc,btc->bot", wv.float().cpu(), x.float().cpu())
    assert rel_l2(vt[:, :, :T], ref) < TOL[dtype]
    if Tpad > T:
        assert float(vt[:, :, T:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(64, 0, 16), (320, 0, 64), (128, 64, 64), (1280, 1280, 16), (640, 320, 100)])
@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("fused", [False, True])
def test_groupnorm(dev, dtype, cfg, silu, fused):
    """fused = the one-launch kernel of the 16x16 / 8x8 levels (ur_groupnorm_fused), else stats + apply."""
    from uni_renderer_amd import ops
    c0, c1, rows = cfg
    B = 3
    x0 = _rand((B, rows, 1, c0), dtype, dev, seed=1) * 2 + 0.5
    x1 = (_rand((B, rows, 1, c1), dtype, dev, seed=2) - 1.0) if c1 else None
    C = c0 + c1
    g = torch.randn(C, generator=torch.Generator().manual_seed(3)).to(dev)
    b = torch.randn(C, generator=torch.Generator().manual_seed(4)).to(dev)
    y = ops.groupnorm(x0, g, b, 1e-5, x1=x1, groups=32, silu=silu, fused=fused)
    xc = torch.cat([x0, x1], -1) if c1 else x0
    ref = F.group_norm(xc.float().cpu().permute(0, 3, 1, 2), 32, g.cpu(), b.cpu(), 1e-5)
    if silu:
        ref = F.silu(ref)
    assert rel_l2(y, ref.permute(0, 2, 3, 1)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(1280, 0, 256, 2), (1280, 1280, 64, 2), (640, 0, 1024, 1), (320, 0, 1024, 2), (1920, 0, 256, 1),
                                 (1280, 640, 256, 2), (1280, 0, 1024, 1), (64, 0, 16, 1)])
@pytest.mark.parametrize("hilo", [False, True])
def test_groupnorm_register_resident_equals_two_sweeps_bitwise(dev, dtype, cfg, hilo):
    """Round 6: the one-launch GroupNorm loads small strips ONCE and keeps them in registers across the block reduction
    (gn_resident_kernel) instead of sweeping them twice.  Same statistics order, same arithmetic: the output must equal the
    two-sweep kernel's bit for bit -- grouped streams with per-stream affine parameters, two sources, (hi, lo) inputs, strips of
    1 .. 8 pieces per thread (the last two cases exceed the resident limit / are tiny: both take their usual path) -- and
    match fp32 group_norm."""
    from uni_renderer_amd import ops
    c0, c1, rows, S = cfg
    B = 2 * S
    x0 = _rand((B, rows, 1, c0), dtype, dev, seed=1) * 2 + 0.5
    x1 = (_rand((B, rows, 1, c1), dtype, dev, seed=2) - 1.0) if c1 else None
    if hilo:
        x0.lo = ops.lo_encode(torch.randn(B, rows, 1, c0, generator=torch.Generator().manual_seed(5)).to(dev) * 2e-4, dtype)
    C = c0 + c1
    g = torch.randn(S * C, generator=torch.Generator().manual_seed(3)).to(dev)
    b = torch.randn(S * C, generator=torch.Generator().manual_seed(4)).to(dev)
    kw = dict(x1=x1, groups=32, silu=True, fused=True, streams=S)
    a = ops.groupnorm(x0, g, b, 1e-5, resident=True, **kw)
    t = ops.groupnorm(x0, g, b, 1e-5, resident=False, **kw)
    assert torch.equal(a, t)
    xc = (torch.cat([x0, x1], -1) if c1 else x0).float().cpu()
    if hilo:
        xc[..., :c0] += ops.lo_float(x0.lo).cpu()
    for s_ in range(S):
        sl = slice(s_ * 2, s_ * 2 + 2)
        ref = F.silu(F.group_norm(xc[sl].permute(0, 3, 1, 2), 32, g[s_ * C:(s_ + 1) * C].cpu(), b[s_ * C:(s_ + 1) * C].cpu(), 1e-5))
        assert rel_l2(a[sl], ref.permute(0, 2, 3, 1)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [64, 320, 640, 1280])
def test_layernorm(dev, dtype, C):
    from uni_renderer_amd import ops
    x = _rand((2, 333, C), dtype, dev, seed=1) * 3 + 1
    g = torch.randn(C, generator=torch.Generator().manual_seed(3)).to(dev)
    b = torch.randn(C, generator=torch.Generator().manual_seed(4)).to(dev)
    y = ops.layernorm(x, g, b, 1e-5)
    ref = F.layer_norm(x.float().cpu(), (C,), g.cpu(), b.cpu(), 1e-5)
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [64, 512, 1024, 2048, 320])
def test_layernorm_many_rows_and_streams(dev, dtype, C):
    """rows >= 8192 selects the several-rows-per-wave launch; streams=2 the per-stream affine parameters."""
    from uni_renderer_amd import ops
    x = _rand((2, 4101, C), dtype, dev, seed=2) * 2 - 0.5
    g = torch.randn(2, C, generator=torch.Generator().manual_seed(5)).to(dev)
    b = torch.randn(2, C, generator=torch.Generator().manual_seed(6)).to(dev)
    y = ops.layernorm(x, g, b, 1e-5, streams=2)
    for s in range(2):
        ref = F.layer_norm(x[s].float().cpu(), (C,), g[s].cpu(), b[s].cpu(), 1e-5)
        assert rel_l2(y[s], ref) < TOL[dtype]


def _attn_ref(q, k, v, H):
    B, Tq, C = q.shape
    d = C // H
    qh = q.float().cpu().view(B, Tq, H, d).transpose(1, 2)
    kh = k.float().cpu().view(B, -1, H, d).transpose(1, 2)
    vh = v.float().cpu().view(B, -1, H, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(B, Tq, C)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d", [32, 40, 64, 80, 160])
@pytest.mark.parametrize("T", [(64, 64), (200, 77), (256, 256), (16, 16), (130, 333)])
def test_attention(dev, dtype, d, T):
    from uni_renderer_amd import ops
    Tq, Tk = T
    B, H = 2, 2
    C = H * d
    q = _rand((B, Tq, C), dtype, dev, seed=1)
    k = _rand((B, Tk, C), dtype, dev, seed=2)
    v = _rand((B, Tk, C), dtype, dev, seed=3)
    Tpad = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, C, Tpad, dtype=dtype, device=dev)
    vt[:, :, :Tk] = v.transpose(1, 2)
    o = ops.attention(q, k, vt, B=B, H=H, Tq=Tq, Tk=Tk, d=d, ldq=C, ldk=C)
    assert rel_l2(o, _attn_ref(q, k, v, H)) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_fused_qk_layout_and_peaky_softmax(dev, dtype):
    """q|k packed in one matrix (self-attention layout), long sequence, and a spiked key per query so the
    running-max rescale path of the online softmax is exercised in every tile."""
    from uni_renderer_amd import ops
    B, H, d, T = 1, 8, 40, 1024
    C = H * d
    qk = _rand((B, T, 2 * C), dtype, dev, seed=1)
    idx = torch.arange(T, device=dev)
    qk[0, idx, C:] += 4.0 * qk[0, (idx * 7 + 3) % T, :C]  # key (7i+3)%T aligned with query ... large scores
    v = _rand((B, T, C), dtype, dev, seed=3)
    vt = v.transpose(1, 2).contiguous()
    o = ops.attention(qk, qk, vt, B=B, H=H, Tq=T, Tk=T, d=d, ldq=2 * C, ldk=2 * C, q_off=0, k_off=C)
    ref = _attn_ref(qk[..., :C], qk[..., C:], v, H)
    assert rel_l2(o, ref) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d", [32, 40, 64, 80, 160])
@pytest.mark.parametrize("T", [(256, 256), (200, 77), (128, 40), (1024, 1000)])
def test_attention_prescaled_log2_scores(dev, dtype, d, T):
    """scale=0: q.k already carries d^-1/2 * log2(e) (folded into the projections by the modules).  For d = 40
    this runs the kernel variant that keeps the softmax reference in the zero-padded k column of Q."""
    from uni_renderer_amd import ops
    Tq, Tk = T
    B, H = 2, 2
    C = H * d
    q = _rand((B, Tq, C), dtype, dev, seed=1)
    k = _rand((B, Tk, C), dtype, dev, seed=2)
    v = _rand((B, Tk, C), dtype, dev, seed=3)
    Tpad = (Tk + 63) // 64 * 64
    vt = torch.zeros(B, C, Tpad, dtype=dtype, device=dev)
    vt[:, :, :Tk] = v.transpose(1, 2)
    cs = d ** -0.5 * 1.4426950408889634
    qs = (q.float() * cs).to(dtype)  # what the q projection epilogue stores
    o = ops.attention(qs, k, vt, B=B, H=H, Tq=Tq, Tk=Tk, d=d, ldq=C, ldk=C, scale=0.0)
    ref = _attn_ref((qs.float() / cs), k, v, H)  # reference on the SAME rounded operands
    assert rel_l2(o, ref) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shift", [-60.0, 0.0, 60.0])
def test_attention_prescaled_reference_tracking(dev, dtype, shift):
    """Scores far below / above zero and a spiked key per query: the first tile must SET the reference (also
    downwards), later tiles must raise it when a score outgrows it by 2^8 (lazy rescale), in both kernel variants."""
    from uni_renderer_amd import ops
    B, H, d, T = 1, 4, 40, 512
    C = H * d
    cs = d ** -0.5 * 1.4426950408889634
    q = _rand((B, T, C), dtype, dev, seed=4)
    k = _rand((B, T, C), dtype, dev, seed=5)
    idx = torch.arange(T, device=dev)
    k[0, idx] += 3.0 * q[0, (idx * 5 + 1) % T]  # every query has one dominant key somewhere along the sequence
    # a constant offset of all scores of a head: one extra-large shared component in q and k
    q[..., 0::d] = 8.0
    k[..., 0::d] = shift / 8.0 / cs
    v = _rand((B, T, C), dtype, dev, seed=6)
    vt = v.transpose(1, 2).contiguous()
    qs = (q.float() * cs).to(dtype)
    o = ops.attention(qs, k, vt, B=B, H=H, Tq=T, Tk=T, d=d, ldq=C, ldk=C, scale=0.0)
    ref = _attn_ref(qs.float() / cs, k, v, H)
    assert bool(torch.isfinite(o.float()).all())
    assert rel_l2(o, ref) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", DTYPES)
def test_add_and_timestep_and_layout(dev, dtype):
    from uni_renderer_amd import ops
    a, b = _rand((3, 5, 7, 64), dtype, dev, seed=1), _rand((3, 5, 7, 64), dtype, dev, seed=2)
    assert rel_l2(ops.add(a, b, 0.5), a.float().cpu() + 0.5 * b.float().cpu()) < TOL[dtype]
    # timestep embedding vs the diffusers formula (flip_sin_to_cos=True, shift 0)
    t = torch.tensor([0.0, 1.0, 500.0, 999.0], device=dev)
    e = ops.timestep_embedding(t, 4, 320, True, 0.0, dtype)
    half = 160
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.cpu()[:, None] * freqs[None]
    ref = torch.cat([torch.cos(arg), torch.sin(arg)], -1)
    assert float((e.float().cpu() - ref).abs().max()) < (2e-3 if dtype == torch.float16 else 1e-2)
    assert torch.equal(e[0].float().cpu(), torch.cat([torch.ones(half), torch.zeros(half)]))  # known answer t=0
    e1 = ops.timestep_embedding(t[2:3], 4, 320, True, 0.0, dtype)  # broadcast of a single timestep
    assert torch.equal(e1[0], e[2]) and torch.equal(e1[3], e[2])
    # layout glue round trip
    x = torch.randn(2, 28, 6, 5, generator=torch.Generator().manual_seed(5)).to(dev)
    n = ops.to_nhwc(x, dtype, 64)
    back = ops.to_nchw(n[..., :28].contiguous(), torch.float32)
    assert torch.equal(back.cpu(), x.to(dtype).float().cpu())
    assert ops.to_nhwc(ops.as_nchw_view(a), dtype).data_ptr() == a.data_ptr()  # zero-copy for channels-last views


def _pair(shape, dtype, dev, seed):
    """An fp32 tensor and its (hi, lo) representation in `dtype`."""
    from uni_renderer_amd import ops
    v = torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * 2 + 0.3
    hi = v.to(dtype)
    lo = ops.lo_encode(v - hi.float(), dtype)  # e5m2 byte (fp16 streams) / bf16
    h = hi.to(dev)
    h.lo = lo.to(dev)
    return hi.float() + ops.lo_float(lo), h  # the value the pair represents


@pytest.mark.parametrize("dtype", DTYPES)
def test_hilo_residual_stream_ops(dev, dtype):
    """(hi, lo) residual stream (include/ur_kernels.h): the pair reproduces the fp32 value to ~2^-14 (fp16, one e5m2
    byte of low part) / 2^-15 (bf16) instead of 2^-11 / 2^-8, through the GEMM / conv epilogues, the norms and the add."""
    from uni_renderer_amd import ops
    tol = 6e-5 if dtype == torch.float16 else 4e-4  # accumulation order + one lo rounding; plain storage: 3e-4 / 2e-3
    LF = ops.lo_float
    # linear: out + out.lo == x @ w^T + b + (res + res.lo)
    x = _rand((2, 200, 128), dtype, dev, seed=1)
    w = (_rand((192, 128), dtype, dev, seed=2) * 0.1).to(dtype)
    b = torch.randn(192, generator=torch.Generator().manual_seed(3)).to(dev)
    rv, r = _pair((2, 200, 192), dtype, dev, 4)
    y = ops.linear(x, w, b, res=r, hilo=True)
    ref = x.float().cpu() @ w.float().cpu().t() + b.cpu() + rv
    assert y.lo is not None and y.lo.shape == y.shape
    assert rel_l2(y.float().cpu() + LF(y.lo).cpu(), ref) < tol
    assert rel_l2(y, ref) < TOL[dtype]  # hi alone is the ordinary rounded result
    assert torch.equal(y.float().cpu() + LF(y.lo).cpu(), (y.float() + LF(y.lo)).cpu())
    # split-K path (epilogue in the reduce kernel)
    y2 = ops.linear(x, w, b, res=r, hilo=True, splitk=2, tile=3)
    assert rel_l2(y2.float().cpu() + LF(y2.lo).cpu(), ref) < tol
    # conv3x3 with a (hi, lo) residual
    xi = _rand((2, 12, 10, 64), dtype, dev, seed=5)
    wc = (_rand((64, 9 * 64), dtype, dev, seed=6) * 0.05).to(dtype)
    rv2, r2 = _pair((2, 12, 10, 64), dtype, dev, 7)
    yc = ops.conv3x3(xi, wc, None, res=r2, hilo=True)
    wt = wc.float().cpu().view(64, 3, 3, 64).permute(0, 3, 1, 2)
    refc = F.conv2d(xi.float().cpu().permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1) + rv2
    assert rel_l2(yc.float().cpu() + LF(yc.lo).cpu(), refc) < tol
    # GroupNorm / LayerNorm read hi + lo
    gv, gx = _pair((2, 9, 7, 320), dtype, dev, 8)
    gam = torch.randn(320, generator=torch.Generator().manual_seed(9)).to(dev)
    bet = torch.randn(320, generator=torch.Generator().manual_seed(10)).to(dev)
    yg = ops.groupnorm(gx, gam, bet, 1e-5, groups=32, silu=True)
    refg = F.silu(F.group_norm(gv.permute(0, 3, 1, 2), 32, gam.cpu(), bet.cpu(), 1e-5)).permute(0, 2, 3, 1)
    plain = gx.clone()  # without the low part the input itself is only good to 2^-11
    yg_plain = ops.groupnorm(plain, gam, bet, 1e-5, groups=32, silu=True)
    assert rel_l2(yg, refg) < TOL[dtype] and rel_l2(yg, refg) <= rel_l2(yg_plain, refg) * 1.05
    lv, lx = _pair((3, 50, 640), dtype, dev, 11)
    g2 = torch.randn(640, generator=torch.Generator().manual_seed(12)).to(dev)
    b2 = torch.randn(640, generator=torch.Generator().manual_seed(13)).to(dev)
    yl = ops.layernorm(lx, g2, b2, 1e-5)
    assert rel_l2(yl, F.layer_norm(lv, (640,), g2.cpu(), b2.cpu(), 1e-5)) < TOL[dtype]
    # add over pairs
    av, a = _pair((4, 8, 64), dtype, dev, 14)
    bv, bb = _pair((4, 8, 64), dtype, dev, 15)
    s_ = ops.add(a, bb, 0.5, hilo=True)
    assert rel_l2(s_.float().cpu() + LF(s_.lo).cpu(), av + 0.5 * bv) < tol


def test_no_cpu_fallback():
    from uni_renderer_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.add(torch.zeros(8, dtype=torch.float16), torch.zeros(8, dtype=torch.float16))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(640, 320, 1, False, None), (960, 320, 1, False, 4), (640, 640, 2, False, None), (1280, 640, 1, True, 2)])
def test_conv3x3_block_outer_k_order(dev, dtype, cfg):
    """cblock > 0: K walks (block of 320 channels, tap) with weights packed to match -- same result as the tap-outer
    order (bit-exact without split-K: the per-output sums run in a different order only across MFMA k-steps, so
    compare within rounding), incl. stride 2, nearest-2x and split-K launches starting inside a block."""
    from uni_renderer_amd import ops
    from uni_renderer_amd.layers import pack_conv3x3
    cin, cout, stride, ups, sk = cfg
    x = _rand((2, 12, 10, cin), dtype, dev, seed=1)
    w = _rand((cout, cin, 3, 3), torch.float32, dev, seed=2) * (cin * 9) ** -0.5
    b = _rand((cout,), torch.float32, dev, seed=3)
    cb = ops.conv_cblock(cin)
    assert cb == 320
    y0 = ops.conv3x3(x, pack_conv3x3(w, dtype), b, stride=stride, ups=ups, splitk=sk, tile=(None if sk is None else 2))
    y1 = ops.conv3x3(x, pack_conv3x3(w, dtype, cblock=cb), b, stride=stride, ups=ups, cblock=cb, splitk=sk,
                     tile=(None if sk is None else 2))
    xin = x.float().cpu().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.to(dtype).float().cpu(), b.cpu(), stride=stride, padding=1).permute(0, 2, 3, 1)
    assert rel_l2(y1, ref) < TOL[dtype] and rel_l2(y1, y0.float().cpu()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(320, 320, 640, 320, None, 0), (640, 320, 640, 0, None, 320), (1280, 640, 1280, 640, 4, 320),
                                 (320, 64, 64, 0, 8, 0)])
def test_conv3x3_with_1x1_tail(dev, dtype, cfg):
    """conv3x3(h) + conv1x1(cat(t0, t1)) in ONE launch (ur_igemm_desc.t0/t1): how a resnet's conv_shortcut rides in the
    K loop of conv2.  Both K orders, split-K slices that start inside the tail, grouped (z = 2) launches."""
    from uni_renderer_amd import ops
    from uni_renderer_amd.layers import pack_conv3x3
    cin, cout, ca, cb, sk, cblock = cfg
    for S in (1, 2):
        h = _rand((S * 2, 10, 12, cin), dtype, dev, seed=1)
        ta = _rand((S * 2, 10, 12, ca), dtype, dev, seed=2)
        tb = _rand((S * 2, 10, 12, cb), dtype, dev, seed=3) if cb else None
        w3 = [_rand((cout, cin, 3, 3), torch.float32, dev, seed=4 + s_) * (9 * cin) ** -0.5 for s_ in range(S)]
        w1 = [_rand((cout, ca + cb), torch.float32, dev, seed=8 + s_) * (ca + cb) ** -0.5 for s_ in range(S)]
        b = [_rand((cout,), torch.float32, dev, seed=12 + s_) for s_ in range(S)]
        wp = torch.stack([torch.cat([pack_conv3x3(w3[s_], dtype, cblock=cblock), w1[s_].to(dtype)], 1) for s_ in range(S)])
        bp = torch.stack(b)
        if S == 1:
            wp, bp = wp[0], bp[0]
        # tile None: the planner's choice; 2: 128x64 on the 16x16x32 MFMA; 24 / 22 / 26: 128x64 / 128x320 / 64x64 on 32x32x16
        for tile in (None, 24, 22, 26, 31, 35, 36) + ((49, 50, 51, 54) if ops.pp_built() else ()):
            y = ops.conv3x3(h, wp, bp, tail=(ta, tb), cblock=cblock, streams=S, hilo=True,
                            splitk=(sk if sk is not None else (None if tile is None else 1)),
                            tile=(tile if tile is not None else (None if sk is None else 2)))
            for s_ in range(S):
                sl = slice(2 * s_, 2 * s_ + 2)
                tcat = torch.cat([ta[sl], tb[sl]], -1) if cb else ta[sl]
                ref = F.conv2d(h[sl].float().cpu().permute(0, 3, 1, 2), w3[s_].to(dtype).float().cpu(), b[s_].cpu(), padding=1)
                ref = ref + F.conv2d(tcat.float().cpu().permute(0, 3, 1, 2), w1[s_].to(dtype).float().cpu()[:, :, None, None])
                assert rel_l2(y[sl], ref.permute(0, 2, 3, 1)) < TOL[dtype], (cfg, S, s_, tile)
                full = y[sl].float() + ops.lo_float(y.lo[sl])  # (hi, lo) pair written by the same epilogue
                assert rel_l2(full, ref.permute(0, 2, 3, 1)) < TOL[dtype], (cfg, S, s_, tile)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", PP_TILES)
def test_pingpong_tiles_long_k_splitk_concat_and_repeatability(dev, dtype, tile):
    """The ping-pong main loop (csrc/igemm_pp.hip) where its ring protocol matters: K long enough to wrap the 4- / 5-slot
    ring many times, split-K slices of uneven length (slices that END early exercise the tail of the counted waits), a
    two-source concat, a ragged M and N tile, stride 2 -- against fp32 conv2d; then 200 launches of one problem must be
    BIT-identical (a ring race shows as a rare differing launch, not as a tolerance failure: DESIGN.md section 5)."""
    from uni_renderer_amd import ops
    from uni_renderer_amd.layers import pack_conv3x3
    _need(tile)
    for (B, H, W, C0, C1, Co, stride, sk) in [(2, 20, 18, 640, 0, 320, 1, 1), (2, 20, 18, 320, 320, 200, 1, 3),
                                              (1, 16, 16, 1280, 0, 640, 2, 4), (3, 9, 7, 64, 0, 96, 1, 1),
                                              (1, 8, 8, 1920, 640, 1280, 1, 7)]:
        x0 = _rand((B, H, W, C0), dtype, dev, seed=1)
        x1 = _rand((B, H, W, C1), dtype, dev, seed=2) if C1 else None
        w = _rand((Co, C0 + C1, 3, 3), dtype, dev, 1 / math.sqrt(9 * (C0 + C1)), seed=3)
        b = torch.randn(Co, generator=torch.Generator().manual_seed(4)).to(dev)
        y = ops.conv3x3(x0, pack_conv3x3(w, dtype), b, x1=x1, stride=stride, tile=tile, splitk=sk)
        xin = x0 if x1 is None else torch.cat([x0, x1], -1)
        ref = _conv_ref(xin, w, b, stride)
        assert y.shape == ref.shape
        assert rel_l2(y, ref) < TOL[dtype], (B, H, W, C0, C1, Co, stride, sk)
    # a GEMM with K = 5120 (160 stages), ragged M
    M, N, K = 1000, 640, 5120
    x = _rand((M, K), dtype, dev, seed=5)
    wl = _rand((N, K), dtype, dev, 1 / math.sqrt(K), seed=6)
    r = _rand((M, N), dtype, dev, seed=7)
    y = ops.linear(x, wl, None, res=r, tile=tile, splitk=1)
    assert rel_l2(y, x.float().cpu() @ wl.float().cpu().t() + r.float().cpu()) < TOL[dtype]
    # bitwise repeatability of the level-0 conv shape (M = 8192 rows: 64 workgroups, every CU of an XCD busy)
    x = _rand((2, 64, 64, 320), dtype, dev, seed=8)
    w = pack_conv3x3(_rand((320, 320, 3, 3), dtype, dev, 1 / math.sqrt(2880), seed=9), dtype)
    first = ops.conv3x3(x, w, None, tile=tile, splitk=1).clone()
    diff = sum(int(not torch.equal(ops.conv3x3(x, w, None, tile=tile, splitk=1), first)) for _ in range(200))
    assert diff == 0, f"{diff} of 200 launches differ"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile,splitk", [(None, None), (2, 1), (9, 1), (3, 2), (24, 1), (49, 1), (54, 2)])
@pytest.mark.parametrize("S", [1, 2])
def test_qkv_projection_with_transposed_value_output(dev, dtype, tile, splitk, S):
    """ur_igemm_desc.out_vt (ABI 8): q | k | v = x @ [Wq; Wk; Wv]^T in ONE launch, q | k row-major (scaled by out_scale),
    the value columns transposed to V^T[sample][channel][token] -- against the two-launch form it replaces (ops.linear +
    ops.vt_proj) and against fp32; grouped (two streams), split-K (the reduce pass runs the same epilogue), 16x16x32 /
    32x32x16 / ping-pong tiles."""
    from uni_renderer_amd import ops
    _need(tile)
    B, T, C = 3, 128, 320
    x = _rand((S * B, T, C), dtype, dev, seed=1)
    w = _rand((S, 3 * C, C), dtype, dev, 1 / math.sqrt(C), seed=2)
    ws = w if S > 1 else w[0]
    qk, vt = ops.linear(x, ws, streams=S, out_scale=0.5, vt_cols=C, vt_tokens=T, tile=tile, splitk=splitk)
    assert qk.shape == (S * B, T, 2 * C) and vt.shape == (S * B, C, T)
    for s_ in range(S):
        xs = x[s_ * B:(s_ + 1) * B].float().cpu()
        full = xs @ w[s_].float().cpu().t()
        assert rel_l2(qk[s_ * B:(s_ + 1) * B], 0.5 * full[..., :2 * C]) < TOL[dtype]
        assert rel_l2(vt[s_ * B:(s_ + 1) * B], full[..., 2 * C:].transpose(1, 2)) < TOL[dtype]
    # bit-identical to the pair of launches it replaces when both use the same tile and no split-K (same MFMA order)
    if tile == 2:
        qk2 = ops.linear(x, (w[:, :2 * C].contiguous() if S > 1 else w[0, :2 * C].contiguous()), streams=S, out_scale=0.5, tile=2, splitk=1)
        assert torch.equal(qk2, qk)
    vt2 = ops.vt_proj(x, (w[:, 2 * C:].contiguous() if S > 1 else w[0, 2 * C:].contiguous()), streams=S)
    assert rel_l2(vt, vt2.float().cpu()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [2, 44, 9])  # 16x16x32 tile, a 32x32x16 tile (its own accumulator layout), the 128x320 tile (ragged N)
@pytest.mark.parametrize("cfg", [(16, 640, 1280, 4, 2, True), (8, 1280, 1280, 8, 2, True), (16, 320, 640, 2, 1, False),
                                 (8, 640, 1280, 16, 1, True), (32, 320, 640, 2, 2, True)])
def test_conv_groupnorm_as_the_splitk_second_pass(dev, dtype, cfg, tile, monkeypatch):
    """ur_igemm_splitk_gn: split-K conv3x3 (+bias, + per-sample time-embedding row) whose second pass IS the GroupNorm
    (+ SiLU) of its output -- against fp32 conv2d -> storage rounding -> group_norm -> silu, and against the unfused
    product path (split-K reduce, then the one-launch GroupNorm), which rounds at the same point: both streams of a grouped
    launch, split-K 2 .. 16; the last case (32x32 map: 1024 rows) is outside the kernel's strip limit and must fall back
    to conv + groupnorm with the same result."""
    from uni_renderer_amd import ops
    from uni_renderer_amd.layers import pack_conv3x3
    monkeypatch.setattr(ops, "SPLITK_GN", True)  # off by default (measured slower in the step): exercised here
    L, Ci, Co, sk, S, silu = cfg
    B = 2
    x = _rand((S * B, L, L, Ci), dtype, dev, seed=1)
    wt = [_rand((Co, Ci, 3, 3), torch.float32, dev, seed=2 + s_) * (9 * Ci) ** -0.5 for s_ in range(S)]
    w = torch.stack([pack_conv3x3(t, dtype) for t in wt])
    bias = torch.stack([_rand((Co,), torch.float32, dev, seed=5 + s_) for s_ in range(S)])
    gam = torch.stack([1.0 + 0.3 * _rand((Co,), torch.float32, dev, seed=7 + s_) for s_ in range(S)])
    bet = torch.stack([0.2 * _rand((Co,), torch.float32, dev, seed=9 + s_) for s_ in range(S)])
    temb = _rand((S * B, 2 * Co), dtype, dev, seed=11)
    if S == 1:
        w, bias, gam, bet = w[0], bias[0], gam[0], bet[0]
    kw = dict(rowadd=temb[:, Co // 2: Co // 2 + Co], streams=S, splitk=sk, tile=tile)
    fused = ops.conv3x3(x, w, bias, gn=(gam, bet, 1e-5, 32, silu), **kw)
    h = ops.conv3x3(x, w, bias, **kw)
    unfused = ops.groupnorm(h, gam, bet, 1e-5, groups=32, silu=silu, streams=S)
    assert ops.splitk_gn_ok(L * L, Co, 32) == (L * L * (Co // 32) <= 16384)
    for s_ in range(S):
        sl = slice(s_ * B, (s_ + 1) * B)
        ref = F.conv2d(x[sl].float().cpu().permute(0, 3, 1, 2), wt[s_].to(dtype).float().cpu(),
                       (bias[s_] if S > 1 else bias).cpu(), padding=1)
        ref = ref + temb[sl, Co // 2: Co // 2 + Co].float().cpu()[:, :, None, None]
        ref = ref.to(dtype).float()  # the storage rounding of the conv output
        ref = F.group_norm(ref, 32, (gam[s_] if S > 1 else gam).cpu(), (bet[s_] if S > 1 else bet).cpu(), 1e-5)
        ref = (F.silu(ref) if silu else ref).permute(0, 2, 3, 1)
        assert rel_l2(fused[sl], ref) < TOL[dtype], cfg
    assert rel_l2(fused, unfused.float().cpu()) < 0.3 * TOL[dtype]  # same rounding points, different summation order


DXS_TILES = [9, 5, 2, 7, 3, 1, 8]  # tiles csrc/igemm_dxs.hip instantiates (128x320, 128x64, 64x64, 128x128, 256x128)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", DXS_TILES)
def test_conv3x3_with_shared_dx_taps(dev, dtype, tile, monkeypatch):
    """csrc/igemm_dxs.hip: the three dx taps of a (channel block, dy, chunk) multiply out of ONE staged pixel block with a
    zero halo per image row.  Every shape class the step has -- all four latent widths (64 .. 8: 2 .. 16 image rows per
    128-pixel tile, tiles that span two samples), the nearest-2x fused upsample, the channel-block-outer K order, the 1x1
    tail over one and two sources, split-K slices of uneven length incl. slices that start inside the tail, grouped
    launches, ragged N, bias + time-embedding row + (hi, lo) residual -- against fp32 conv2d; the launch must really have
    taken the kernel (ur_igemm_uses_dxs)."""
    from uni_renderer_amd import ops
    from uni_renderer_amd.layers import pack_conv3x3
    if not ops.dxs_active():
        pytest.skip("the dx-tap-sharing conv kernel is an opt-in build (make DXS=1, UR_DXS=1): measured 4-8 % slower in the "
                    "step than the lock-step kernels (DESIGN.md section 4, round 4), not part of the product library")
    trace = []
    monkeypatch.setattr(ops, "DXS_TRACE", trace)
    cases = [  # (B, H=W, Cin, Cout, ups, cblock, tail channels (a, b), splitk, streams)
        (1, 64, 64, 96, False, 0, (0, 0), 1, 1), (2, 32, 128, 320, False, 0, (0, 0), 1, 2), (3, 16, 640, 200, False, 320, (0, 0), 3, 1),
        (4, 8, 640, 320, False, 320, (0, 0), 4, 2), (2, 8, 128, 64, True, 0, (0, 0), 1, 1), (1, 16, 320, 128, True, 0, (0, 0), 2, 2),
        (2, 16, 128, 192, False, 0, (64, 0), 1, 1), (2, 32, 640, 320, False, 320, (128, 64), 7, 2), (5, 8, 192, 64, False, 64, (64, 64), 10, 1)]
    for (B, L, Ci, Co, ups, cb, (ca, cbt), sk, S) in cases:
        Lo = 2 * L if ups else L
        x = _rand((S * B, L, L, Ci), dtype, dev, seed=1)
        wt = [_rand((Co, Ci, 3, 3), torch.float32, dev, seed=2 + s_) * (9 * Ci) ** -0.5 for s_ in range(S)]
        ta = _rand((S * B, Lo, Lo, ca), dtype, dev, seed=20) if ca else None
        tb = _rand((S * B, Lo, Lo, cbt), dtype, dev, seed=21) if cbt else None
        w1 = [_rand((Co, ca + cbt), torch.float32, dev, seed=30 + s_) * max(ca + cbt, 1) ** -0.5 for s_ in range(S)] if ca else None
        w = torch.stack([torch.cat([pack_conv3x3(wt[s_], dtype, cblock=cb)] + ([w1[s_].to(dtype)] if ca else []), 1) for s_ in range(S)])
        bias = torch.stack([_rand((Co,), torch.float32, dev, seed=40 + s_) for s_ in range(S)])
        temb = _rand((S * B, Co), dtype, dev, seed=50)
        res = _rand((S * B, Lo, Lo, Co), dtype, dev, seed=51)
        if S == 1:
            w, bias = w[0], bias[0]
        del trace[:]
        y = ops.conv3x3(x, w, bias, ups=ups, rowadd=temb, res=res, out_scale=0.5, tile=tile, splitk=sk, streams=S, hilo=True,
                        cblock=cb, tail=((ta, tb) if ca else None))
        assert trace == [1], (B, L, Ci, Co, ups, cb, ca, cbt, sk, S, trace)
        for s_ in range(S):
            sl = slice(s_ * B, (s_ + 1) * B)
            ref = _conv_ref(x[sl], wt[s_].to(dtype), (bias[s_] if S > 1 else bias), 1, ups)
            if ca:
                tcat = torch.cat([ta[sl], tb[sl]], -1) if cbt else ta[sl]
                ref = ref + (tcat.float().cpu() @ w1[s_].to(dtype).float().cpu().t())
            ref = (ref + temb[sl].float().cpu()[:, None, None, :] + res[sl].float().cpu()) * 0.5
            full = y[sl].float() + ops.lo_float(y.lo[sl])
            assert rel_l2(y[sl], ref) < TOL[dtype], (B, L, Ci, Co, ups, cb, ca, cbt, sk, S, s_)
            assert rel_l2(full, ref) < TOL[dtype]
    # ineligible shapes stay on the lock-step kernel: odd widths, stride 2, two sources
    del trace[:]
    x = _rand((2, 12, 10, 128), dtype, dev, seed=1)
    wt_ = _rand((64, 128, 3, 3), dtype, dev, 1 / math.sqrt(9 * 128), seed=2)
    ops.conv3x3(x, pack_conv3x3(wt_, dtype), None, tile=tile, splitk=1)
    x = _rand((2, 16, 16, 128), dtype, dev, seed=1)
    ops.conv3x3(x, pack_conv3x3(wt_, dtype), None, stride=2, tile=tile, splitk=1)
    assert trace == [0, 0]
    # 200 launches of the level-0 shape are bit-identical (the ring protocol of the pixel / weight buffers)
    x = _rand((2, 64, 64, 320), dtype, dev, seed=8)
    w = pack_conv3x3(_rand((320, 320, 3, 3), dtype, dev, 1 / math.sqrt(2880), seed=9), dtype)
    first = ops.conv3x3(x, w, None, tile=tile, splitk=1).clone()
    assert sum(int(not torch.equal(ops.conv3x3(x, w, None, tile=tile, splitk=1), first)) for _ in range(200)) == 0
