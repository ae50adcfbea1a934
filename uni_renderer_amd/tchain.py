"""Host side of the row-local transformer chains (``ur_tchain``, csrc/tchain.hip; C ABI in include/ur_kernels.h).

One BasicTransformerBlock of the reference (diffusers 0.24, instantiated by models/unet_2d_blocks.py:1115-1126) at the
320-channel level is, after each attention, a chain of row-local operations:

    x  = attn1_out . Wo1^T + bo1 + x ;  q = LN2(x) . Wq2^T                          -> ``chain_q``   (UR_TCHAIN_Q)
    x  = attn2_out . Wo2^T + bo2 + x ;  x = x + FF(LN3(x)) ;  out = proj_out(x) + block input  -> ``chain_ff``  (UR_TCHAIN_FF)

The kernel keeps the activations of 128 rows in registers and streams only weights; this module builds what it
streams, once per (dtype, parameter version):

* **stage images**: 40960-byte blocks in exactly the byte order of the LDS ring slot they are copied into.  An image
  holds rows of 64 k-values (128 bytes); inside a row the eight 16-byte chunks are stored XOR-swizzled by
  ``(row >> 1) & 7`` so that the ``ds_read_b128`` A-fragment reads of v_mfma_f32_32x32x16 (32 rows x two k halves per
  instruction) are bank-conflict free (same key as csrc/igemm.hip's 32x32 build, tools/lds_bank_check.py).
* **KPERM**: a matrix whose operand comes out of an MFMA accumulator (everything except the leading GEMM, whose operand
  is loaded from memory) has its k columns permuted per group of 16 by [0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15]: lane
  (pixel, half h) of an accumulator block owns rows 8 q + 4 h + r, so the 8 values it can hand to the next MFMA as
  k = 8 h .. 8 h + 7 are the channels {4 h + r} of two consecutive 8-row blocks.
* **const block**: the fp32 vectors the kernel reads from LDS: leading bias | LayerNorm gamma | beta
  [| FF bias value half | gate half | FF-out bias | proj_out bias].
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib, ops
from ._lib import TChainDesc, check

MODE_Q, MODE_FF, MODE_PRE = 0, 1, 2
CH = 320           # the level the kernel is built for
FF_HIDDEN = 4 * CH  # GEGLU hidden width the feed-forward chain is built for (TC_FF in csrc/tchain.hip)
HEADS = 8          # heads of the head-major q / k images (ur_tchain_desc.qk_heads)
STAGE = 40960      # bytes per stage image
KPERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


def kperm(w: torch.Tensor) -> torch.Tensor:
    """Columns of ``w`` [N, K] reordered so that packed column 16 g + kk holds original column 16 g + KPERM16[kk]."""
    N, K = w.shape
    idx = torch.tensor(KPERM16, device=w.device)
    return w.reshape(N, K // 16, 16)[:, :, idx].reshape(N, K)


def _swizzle_rows(img: torch.Tensor) -> torch.Tensor:
    """img [R, 64] (k-values of one row) -> [R, 64] with the eight 8-element chunks of row r stored at chunk position
    c ^ ((r >> 1) & 7)  (position c' holds logical chunk c' ^ key: XOR is its own inverse)."""
    R = img.shape[0]
    r = torch.arange(R, device=img.device)
    key = (r >> 1) & 7
    pos = torch.arange(8, device=img.device)[None, :] ^ key[:, None]          # [R, 8]: logical chunk stored at position c'
    return torch.gather(img.reshape(R, 8, 8), 1, pos[:, :, None].expand(R, 8, 8)).reshape(R, 64)


def gemm_images(w: torch.Tensor, permute_k: bool) -> torch.Tensor:
    """w [320, K] (K % 64 == 0) -> [K // 64, 320 * 64]: one stage image per 64-wide k chunk."""
    N, K = w.shape
    assert N == CH and K % 64 == 0
    wp = kperm(w) if permute_k else w
    return torch.stack([_swizzle_rows(wp[:, 64 * c: 64 * c + 64]).reshape(-1) for c in range(K // 64)], 0)


def ff_images(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """GEGLU feed-forward as the kernel walks it.  Per 64 hidden units j there are three images
         A0(j) / A1(j): for hidden 64 j + 32 half .. + 31: five [64 rows, 64 k] sub-images (k chunk c of the 320 inputs),
                  rows 0..31 = value rows of w1 (``proj`` rows hid), rows 32..63 = gate rows (rows 1280 + hid);
         B(j):    0.5 * w2[:, 64 j .. 64 j + 63] as [320 rows, 64 k] -- the kernel's GEGLU program produces
                  2 * value * gelu(gate) (it leaves the 0.5 of 0.5 x (1 + erf) out; a power of two, so folding it into
                  the weights changes no rounding),
       in the software-pipelined consumption order  A0(0) A1(0) | A0(j) A1(j) B(j-1) for j = 1 .. | B(last)
       (csrc/tchain.hip: B lags one step so that the GEGLU of a half-chunk runs under the next stage's MFMAs).
       Both operands come out of accumulators -> KPERM on the k axis of both."""
    nh = w2.shape[1]
    assert w1.shape == (2 * nh, CH) and w2.shape == (CH, nh) and nh % 64 == 0
    w1p, w2p = kperm(w1), kperm(w2 * 0.5)

    def a_img(j, half):
        hid = 64 * j + 32 * half
        rows = torch.cat([w1p[hid: hid + 32], w1p[nh + hid: nh + hid + 32]], 0)          # [64, 320]
        return torch.cat([_swizzle_rows(rows[:, 64 * c: 64 * c + 64]).reshape(-1) for c in range(CH // 64)], 0)

    def b_img(j):
        return _swizzle_rows(w2p[:, 64 * j: 64 * j + 64]).reshape(-1)

    nj = nh // 64
    out = [a_img(0, 0), a_img(0, 1)]
    for j in range(1, nj):
        out += [a_img(j, 0), a_img(j, 1), b_img(j - 1)]
    out.append(b_img(nj - 1))
    return torch.stack(out, 0)


def pack_chain_q(wo: torch.Tensor, bo: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, wq: torch.Tensor,
                 q_scale: float, dtype):
    """(wstream [10 * 20480] dtype elements, consts [960] fp32) of one UR_TCHAIN_Q chain.  ``q_scale`` is folded into the
    query weights in fp32 (the attention kernels take q.k in log2 units, layers.Attention)."""
    w0 = wo.detach().float().reshape(CH, CH)
    wq_ = wq.detach().float().reshape(CH, CH) * q_scale
    stream = torch.cat([gemm_images(w0, False), gemm_images(wq_, True)], 0).to(dtype).reshape(-1).contiguous()
    consts = torch.cat([bo.detach().float(), gamma.detach().float(), beta.detach().float()]).contiguous()
    return stream, consts


def pack_chain_pre(wpi: torch.Tensor, bpi: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, wq: torch.Tensor,
                   wk: torch.Tensor, wv: torch.Tensor, qk_scale: float, dtype):
    """(wstream [20 stages], consts [960]) of one UR_TCHAIN_PRE chain: proj_in (1x1 conv [320, 320, 1, 1]) | LayerNorm1 | q, k,
    v projections of the self-attention.  ``qk_scale`` = sqrt(d^-0.5 * log2(e)) multiplies BOTH q and k weights (the
    attention kernel takes q.k in log2 units, layers.Attention)."""
    f = lambda t: t.detach().float().reshape(CH, CH)
    stream = torch.cat([gemm_images(f(wpi), False), gemm_images(f(wq) * qk_scale, True), gemm_images(f(wk) * qk_scale, True),
                        gemm_images(f(wv), True)], 0).to(dtype).reshape(-1).contiguous()
    consts = torch.cat([bpi.detach().float(), gamma.detach().float(), beta.detach().float()]).contiguous()
    return stream, consts


def pack_chain_ff(wo: torch.Tensor, bo: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w1: torch.Tensor,
                  b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, wpo: torch.Tensor, bpo: torch.Tensor, dtype):
    """(wstream, consts [3840] fp32) of one UR_TCHAIN_FF chain.  ``w1`` / ``b1`` are diffusers' GEGLU ``proj`` ([2 * 1280,
    320]: value rows first, gate rows second), ``w2`` the FeedForward output Linear, ``wpo`` / ``bpo`` proj_out."""
    f = lambda t, *s: t.detach().float().reshape(*s)
    nh = w2.shape[1]
    stream = torch.cat([gemm_images(f(wo, CH, CH), False), ff_images(f(w1, 2 * nh, CH), f(w2, CH, nh)),
                        gemm_images(f(wpo, CH, CH), True)], 0).to(dtype).reshape(-1).contiguous()
    b1f = b1.detach().float()
    consts = torch.cat([f(bo, CH), f(gamma, CH), f(beta, CH), b1f[:nh], b1f[nh:], f(b2, CH), f(bpo, CH)]).contiguous()
    return stream, consts


def _launch(mode, a0, res, wstream, consts, eps, *, blk=None, y_out=None, out=None, streams=1, profile=None, out2=None,
            out3=None, rows_per_b=0, qk_heads=0):
    ops._require_gpu(a0)
    lib = _lib.load()
    Cn = a0.shape[-1]
    M = a0.numel() // Cn // streams
    d = TChainDesc()
    d.a0, d.out = a0.data_ptr(), out.data_ptr()
    if res is not None:
        d.res, d.res_lo = res.data_ptr(), ops._ptr(ops.lo_of(res))
    if out2 is not None:
        d.out2, d.out3, d.rows_per_b, d.ld_vt = out2.data_ptr(), out3.data_ptr(), rows_per_b, out3.shape[-1]
    if qk_heads:
        d.qk_heads, d.rows_per_b = int(qk_heads), int(rows_per_b)
    d.out_lo = ops._ptr(ops.lo_of(out))
    if blk is not None:
        d.blk, d.blk_lo = blk.data_ptr(), ops._ptr(ops.lo_of(blk))
    if y_out is not None:
        d.y_out, d.y_out_lo = y_out.data_ptr(), ops._ptr(ops.lo_of(y_out))
    d.wstream, d.consts = wstream.data_ptr(), consts.data_ptr()
    d.z_wstream = wstream.stride(0) * wstream.element_size() if streams > 1 else 0
    d.z_consts = consts.stride(0) if streams > 1 else 0
    if streams == 1:  # the C side still checks the per-z sizes
        d.z_wstream, d.z_consts = wstream.numel() * wstream.element_size(), consts.numel()
    d.M, d.zbatch, d.mode, d.dtype, d.channels, d.eps = M, streams, mode, ops.DT[a0.dtype], Cn, float(eps)
    if profile is not None:
        d.profile = profile.data_ptr()
    e0 = ops._prof_begin()
    check(lib.ur_tchain(C.byref(d), ops._stream()), "ur_tchain")
    if e0 is not None:
        el = a0.element_size()
        rows = M * streams
        if mode == MODE_PRE:
            fl = 2.0 * rows * CH * CH * 4
            by = rows * CH * (el * 5 + (1 if a0.dtype == torch.float16 else el)) + wstream.numel() * el
        elif mode == MODE_Q:
            fl = 2.0 * rows * CH * CH * 2
            by = rows * CH * (el * 4 + 2 * (1 if a0.dtype == torch.float16 else el)) + wstream.numel() * el
        else:
            fl = 2.0 * rows * CH * (CH + 8 * CH + 4 * CH + CH)
            by = rows * CH * (el * 4 + 3 * (1 if a0.dtype == torch.float16 else el)) + wstream.numel() * el
        ops._prof_end(e0, {MODE_Q: "tchain_q", MODE_FF: "tchain_ff", MODE_PRE: "tchain_pre"}[mode], fl, by)


def supported(x: torch.Tensor) -> bool:
    return x.is_cuda and x.shape[-1] == CH and x.dtype in (torch.float16, torch.bfloat16)


def chain_pre(x_norm, wstream, consts, eps, *, tokens_per_sample, streams=1, hilo=True, profile=None, head_major=False):
    """x_norm [S*B*T, 320] (the GroupNorm output).  Returns (y, q, k, vt): y = proj_in(x_norm) as a (hi, lo) tensor, q / k
    [S*B*T, 320] already carrying sqrt(scale * log2 e) each, vt [S*B, 320, Tpad] (columns >= T zero).
    ``head_major``: q / k leave as [S*B, 8, T, 40] images (same storage; ``ops.attention(..., q_hstride=T * 40,
    k_hstride=T * 40)`` reads them): a head's keys are contiguous instead of 80-byte slices of 640-byte token rows."""
    T = tokens_per_sample
    rows = x_norm.shape[0]
    Tpad = (T + 63) // 64 * 64
    y = ops._with_lo(torch.empty_like(x_norm), hilo)
    q, k = torch.empty_like(x_norm), torch.empty_like(x_norm)
    vt = (torch.zeros if Tpad != T else torch.empty)(rows // T, CH, Tpad, dtype=x_norm.dtype, device=x_norm.device)
    _launch(MODE_PRE, x_norm, None, wstream, consts, eps, y_out=y, out=q, out2=k, out3=vt, rows_per_b=T, streams=streams,
            profile=profile, qk_heads=HEADS if head_major else 0)
    return y, q, k, vt


def chain_q(attn_out, residual, wstream, consts, eps, *, streams=1, hilo=True, profile=None, head_major_tokens=0):
    """Returns (y, q): y = attn_out Wo^T + bo + residual as a (hi, lo) tensor (``y.lo`` when ``hilo``), q = LN(y) Wq^T.
    ``head_major_tokens`` = T > 0: q leaves as a [S*B, 8, T, 40] image (see ``chain_pre``)."""
    y = ops._with_lo(torch.empty_like(attn_out), hilo)
    q = torch.empty_like(attn_out)
    _launch(MODE_Q, attn_out, residual, wstream, consts, eps, y_out=y, out=q, streams=streams, profile=profile,
            rows_per_b=head_major_tokens, qk_heads=HEADS if head_major_tokens else 0)
    return y, q


def chain_ff(attn_out, residual, block_input, wstream, consts, eps, *, streams=1, hilo=True, profile=None):
    """out = proj_out(x + FF(LN3(x))) + block_input with x = attn_out Wo^T + bo + residual."""
    out = ops._with_lo(torch.empty_like(attn_out), hilo)
    _launch(MODE_FF, attn_out, residual, wstream, consts, eps, blk=block_input, out=out, streams=streams, profile=profile)
    return out
