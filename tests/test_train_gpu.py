"""The training step on the GPU (uni_renderer_amd/train_step.py; SURVEY section 8a device op 11, cfg 4) against the
CPU oracle's autograd: same parameters, same inputs, same x0-prediction MSE losses -> loss value and the gradient of
every parameter of the three networks; then a few optimisation steps must bring the loss down, and two ranks' worth
of gradient buckets must equal the single-rank gradients (gloo collective on CPU tensors is covered in
test_parallel_cpu.py; here the bucket list is built over the GPU modules)."""
import pytest
import torch

from conftest import rel_l2
from util_models import O, build_product_from_oracle

pytestmark = pytest.mark.gpu


def _oracle_loss(models, x, c, ehs, ti, ta, tgt_img, tgt_attr):
    """the call pattern of O.dual_stream_step (which runs under no_grad) with autograd on"""
    unet, enc, dec = models
    res, mid, raw_enc, raw_mid_enc = enc(x, ta, ehs, controlnet_cond=c)
    img_pred, raw_unet, raw_mid_unet, _ = unet(x, ti, ehs, down_block_additional_residuals=res,
                                               mid_block_additional_residual=mid)
    attr_pred = dec(raw_mid_enc, raw_enc, ta, ehs, down_block_additional_residuals=raw_unet,
                    mid_block_additional_residual=raw_mid_unet)
    return torch.mean((img_pred - tgt_img) ** 2) + torch.mean((attr_pred - tgt_attr) ** 2)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-2), (torch.bfloat16, 1.2e-1)])
def test_train_step_loss_and_gradients_match_oracle_autograd(dev, dtype, tol):
    from uni_renderer_amd.train_step import dual_stream_forward, mse_losses

    oracle = O.build_triplet(O.TINY_CONFIG, seed=31)
    x, c, ehs, ti, ta = O.make_inputs(2, 16, 64, seed=12)
    g = torch.Generator().manual_seed(13)
    tgt_img, tgt_attr = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 28, 16, 16, generator=g)
    for m in oracle:
        m.requires_grad_(True)
    loss_o = _oracle_loss(oracle, x, c, ehs, ti, ta, tgt_img, tgt_attr)
    loss_o.backward()

    unet, enc, dec = build_product_from_oracle(*oracle, torch.float32, dev)  # fp32 master parameters
    for m in (unet, enc, dec):
        m.train()
        m.requires_grad_(True)
    out = dual_stream_forward(unet, enc, dec, x.to(dev), c.to(dev), ehs.to(dev), ti.to(dev), ta.to(dev), dtype=dtype)
    loss = mse_losses(out, tgt_img.to(dev), tgt_attr.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o.detach())) / float(loss_o.detach()) < (3e-3 if dtype == torch.float16 else 2e-2)

    worst, flat_p, flat_o, rows = ("", 0.0), [], [], []
    for mo, mp in zip(oracle, (unet, enc, dec)):
        po, pp = dict(mo.named_parameters()), dict(mp.named_parameters())
        assert set(po) == set(pp)
        for name, p in pp.items():
            assert p.grad is not None, name
            go = po[name].grad
            if go is None or float(go.abs().max()) == 0.0:  # e.g. the encoder's unused branches
                continue
            e = rel_l2(p.grad, go)
            flat_p.append(p.grad.float().cpu().reshape(-1))
            flat_o.append(go.reshape(-1))
            rows.append((name, e, float(go.norm()) / go.numel() ** 0.5,
                         float((p.grad.float().cpu() - go).norm()) / go.numel() ** 0.5))
    total = rel_l2(torch.cat(flat_p), torch.cat(flat_o))
    rms_all = float(torch.cat(flat_o).norm()) / sum(t.numel() for t in flat_o) ** 0.5
    # a parameter passes if its gradient is right relative to its own size, or -- for gradients that nearly cancel
    # (e.g. the LayerNorm bias in front of a cross-attention query) -- if the absolute error is small against the
    # typical gradient entry of the network
    for name, e, rms_o, rms_err in rows:
        if e > worst[1] and rms_err > 0.02 * rms_all:
            worst = (name, e)
    print({"dtype": str(dtype), "loss": float(loss), "loss_oracle": float(loss_o), "grad_rel_l2_all": total,
           "worst_param": worst})
    # measured (r02): fp16 3.2e-3 total, bf16 2.6e-2 total / 5.4e-2 worst parameter: bounds ~2x that
    assert total < tol / 2 and worst[1] < (tol * 3 if dtype == torch.float16 else tol)


def test_few_optimizer_steps_reduce_the_loss(dev):
    from uni_renderer_amd.parallel import GradientBuckets
    from uni_renderer_amd.train_step import train_step

    oracle = O.build_triplet(O.TINY_CONFIG, seed=32)
    nets = build_product_from_oracle(*oracle, torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    x, c, ehs, ti, ta = [t.to(dev) for t in O.make_inputs(2, 16, 64, seed=14)]
    g = torch.Generator().manual_seed(15)
    batch = dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=torch.randn(2, 4, 16, 16, generator=g).to(dev),
                 target_attr=torch.randn(2, 28, 16, 16, generator=g).to(dev))
    opt = torch.optim.AdamW([p for m in nets for p in m.parameters()], lr=2e-4)
    buckets = GradientBuckets(nets)  # single rank: all_reduce_mean is a no-op, the bucket list must cover everything
    assert sum(p.numel() for b in buckets.buckets for p in b) == sum(p.numel() for m in nets for p in m.parameters())
    losses = [train_step(nets, batch, optimizer=opt, buckets=buckets, dtype=torch.bfloat16)["loss"] for _ in range(6)]
    print({"losses": losses})
    assert losses[-1] < losses[0] * 0.9 and all(l == l for l in losses)


def test_train_step_gradients_sd_size(dev):
    """SD-1.x-size networks (1.74 G parameters), 128x128 image = 16x16 latent, batch 1, bf16 compute with fp32 master
    parameters (cfg 4's arithmetic): loss and the gradient of all parameters against the CPU oracle's autograd."""
    from uni_renderer_amd.train_step import dual_stream_forward, mse_losses

    oracle = O.build_triplet(O.SD15_CONFIG, seed=41)
    x, c, ehs, ti, ta = O.make_inputs(1, 16, 768, seed=16)
    g = torch.Generator().manual_seed(17)
    tgt_img, tgt_attr = torch.randn(1, 4, 16, 16, generator=g), torch.randn(1, 28, 16, 16, generator=g)
    for m in oracle:
        m.requires_grad_(True)
    loss_o = _oracle_loss(oracle, x, c, ehs, ti, ta, tgt_img, tgt_attr)
    loss_o.backward()
    nets = build_product_from_oracle(*oracle, torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    out = dual_stream_forward(*nets, x.to(dev), c.to(dev), ehs.to(dev), ti.to(dev), ta.to(dev), dtype=torch.bfloat16)
    loss = mse_losses(out, tgt_img.to(dev), tgt_attr.to(dev))
    loss.backward()
    num = den = 0.0
    for mo, mp in zip(oracle, nets):
        po = dict(mo.named_parameters())
        for name, p in mp.named_parameters():
            go = po[name].grad
            if go is None:
                continue
            d = p.grad.float().cpu() - go
            num += float((d * d).sum())
            den += float((go * go).sum())
    err = (num / den) ** 0.5
    print({"sd_size_bf16": True, "loss": float(loss.detach()), "loss_oracle": float(loss_o.detach()), "grad_rel_l2_all": err})
    assert abs(float(loss.detach()) - float(loss_o.detach())) / float(loss_o.detach()) < 1e-2
    assert err < 4.5e-2  # measured 2.2e-2 (r02)


@pytest.mark.parametrize("inverse", [True, False])
def test_reference_losses_match_oracle_autograd(dev, inverse):
    """The objectives of train/train.py:1356-1413 -- inverse rendering with the cycle-consistency pass (the decoder's
    prediction is fed back, differentiably, as the encoder's condition) and rendering with the contrastive term --
    loss value and all parameter gradients against the same computation on the CPU oracle."""
    from uni_renderer_amd.train_step import reference_losses

    F = torch.nn.functional
    oracle = O.build_triplet(O.TINY_CONFIG, seed=33)
    unet_o, enc_o, dec_o = oracle
    x, c, ehs, ti, ta = O.make_inputs(2, 16, 64, seed=18)
    g = torch.Generator().manual_seed(19)
    tgt_img, tgt_attr = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 24, 16, 16, generator=g)
    x_c, ti_c = torch.randn(2, 4, 16, 16, generator=g), torch.randint(0, 1000, (2,), generator=g)
    for m in oracle:
        m.requires_grad_(True)
    # ---- oracle (NCHW, fp32), literally the reference's step body
    res, mid, raw_enc, raw_mid_enc = enc_o(x, ta, ehs, controlnet_cond=c)
    img_pred, raw_unet, raw_mid_unet, _ = unet_o(x, ti, ehs, down_block_additional_residuals=res,
                                                 mid_block_additional_residual=mid)
    mask_pred = dec_o(raw_mid_enc, raw_enc, ta, ehs, down_block_additional_residuals=raw_unet,
                      mid_block_additional_residual=raw_mid_unet)[:, 4:]
    loss_img, loss_mask = F.mse_loss(img_pred, tgt_img), F.mse_loss(mask_pred, tgt_attr)
    if inverse:
        cond_c = torch.cat((c[:, :4], mask_pred), dim=1)
        res, mid, _, _ = enc_o(x_c, torch.zeros(2).long(), ehs, controlnet_cond=cond_c)
        img_c = unet_o(x_c, ti_c, ehs, down_block_additional_residuals=res, mid_block_additional_residual=mid)[0]
        loss_o = loss_img + loss_mask + 0.8 * F.mse_loss(img_c, tgt_img)
    else:
        cos = lambda a: F.cosine_similarity(a[0].reshape(-1), a[1].reshape(-1), dim=0) / 0.1
        pos = torch.exp(cos(mask_pred[:, 8:12]))
        neg = pos + torch.exp(cos(mask_pred[:, :4])) + torch.exp(cos(mask_pred[:, 12:16]))
        loss_o = loss_img + loss_mask * 10.0 - torch.log(pos / neg) * 0.01
    loss_o.backward()
    # ---- product
    nets = build_product_from_oracle(*oracle, torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    batch = dict(x_t=x.to(dev), cond=c.to(dev), ehs=ehs.to(dev), t_img=ti.to(dev), t_attr=ta.to(dev),
                 target_img=tgt_img.to(dev), target_attr=tgt_attr.to(dev), x_t_c=x_c.to(dev), t_img_c=ti_c.to(dev))
    loss = reference_losses(nets, batch, dtype=torch.float16, inverse=inverse)["loss"]
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o.detach())) / abs(float(loss_o.detach())) < 5e-3
    num = den = 0.0
    for mo, mp in zip(oracle, nets):
        po = dict(mo.named_parameters())
        for name, p in mp.named_parameters():
            go = po[name].grad
            if go is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name  # e.g. the decoder in no branch
                continue
            assert p.grad is not None, name
            d = p.grad.float().cpu() - go
            num += float((d * d).sum())
            den += float((go * go).sum())
    err = (num / den) ** 0.5
    print({"inverse": inverse, "loss": float(loss.detach()), "loss_oracle": float(loss_o.detach()), "grad_rel_l2_all": err})
    assert err < 1.5e-2


def test_training_step_captured_into_a_hip_graph(dev):
    """The whole optimisation step -- forward, losses, backward, clipping, fused AdamW -- captured once into a HIP graph
    (train_step(as_tensors=True): no host synchronisation inside) and replayed must walk the same parameter trajectory
    as the eager step."""
    from uni_renderer_amd.train_step import train_step

    def setup():
        oracle = O.build_triplet(O.TINY_CONFIG, seed=34)
        nets = build_product_from_oracle(*oracle, torch.float32, dev)
        for m in nets:
            m.train()
            m.requires_grad_(True)
        opt = torch.optim.AdamW([p for m in nets for p in m.parameters()], lr=2e-4, fused=True, capturable=True)
        return nets, opt

    x, c, ehs, ti, ta = [t.to(dev) for t in O.make_inputs(2, 16, 64, seed=20)]
    g = torch.Generator().manual_seed(21)
    batch = dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=torch.randn(2, 4, 16, 16, generator=g).to(dev),
                 target_attr=torch.randn(2, 28, 16, 16, generator=g).to(dev))
    nets_e, opt_e = setup()
    eager = [train_step(nets_e, batch, optimizer=opt_e, dtype=torch.bfloat16)["loss"] for _ in range(5)]

    nets_g, opt_g = setup()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    losses = []
    with torch.cuda.stream(side):
        for _ in range(2):  # the usual warm-up iterations on the capture stream: they are real steps 1 and 2
            losses.append(train_step(nets_g, batch, optimizer=opt_g, dtype=torch.bfloat16, as_tensors=True)["loss"].clone())
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = train_step(nets_g, batch, optimizer=opt_g, dtype=torch.bfloat16, as_tensors=True)
    # NOTE: capture does not execute; replays are steps 3, 4, 5
    for _ in range(3):
        graph.replay()
        losses.append(st["loss"].clone())
    torch.cuda.synchronize()
    losses = [float(l) for l in losses]
    print({"eager": eager, "graphed": losses})
    assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(eager, losses)), (eager, losses)
    num = sum(float(((pe.detach() - pg.detach()) ** 2).sum()) for me, mg in zip(nets_e, nets_g)
              for pe, pg in zip(me.parameters(), mg.parameters()))
    den = sum(float((pe.detach() ** 2).sum()) for me in nets_e for pe in me.parameters())
    assert (num / den) ** 0.5 < 1e-4


def _reference_step_losses(controlnet, unet, controldec, b, inverse, F=torch.nn.functional):
    """The step body of train/train.py:1324-1416 written ONCE over the reference's module call surface (keyword for
    keyword), so the same function drives the product modules on the GPU and the oracle on the CPU.  ``b``: dict of NCHW
    tensors.  (The oracle's forwards take the same keywords minus ``return_dict``.)"""
    kw = {} if b.get("oracle") else {"return_dict": False}
    down_block_res_samples, mid_block_res_sample, raw_down_ctl, raw_mid_ctl = controlnet(
        b["noisy_latents_img"], b["timesteps_attribute"], encoder_hidden_states=b["ehs"],
        controlnet_cond=b["noisy_latents_attr"], **kw)
    wd = b["weight_dtype"]
    img_pred, raw_down_unet, raw_mid_unet, _ = unet(
        b["noisy_latents_img"], b["timesteps_img"], encoder_hidden_states=b["ehs"],
        down_block_additional_residuals=[s.to(dtype=wd) for s in down_block_res_samples],
        mid_block_additional_residual=mid_block_res_sample.to(dtype=wd), **kw)
    mask_pred = controldec(
        sample=raw_mid_ctl, down_block_res_samples=raw_down_ctl, timestep=b["timesteps_attribute"],
        encoder_hidden_states=b["ehs"], down_block_additional_residuals=[s.to(dtype=wd) for s in raw_down_unet],
        mid_block_additional_residual=raw_mid_unet.to(dtype=wd), **kw)
    mask_pred = mask_pred[:, 4:, :, :]
    material_pred, albedo_pred, spec_pred = mask_pred[:, :4], mask_pred[:, 8:12], mask_pred[:, 12:16]
    temperature = 0.1
    cos = lambda a: F.cosine_similarity(a[0].reshape(-1).float(), a[1].reshape(-1).float(), dim=0) / temperature
    pos = torch.exp(cos(albedo_pred))
    neg = pos + torch.exp(cos(material_pred)) + torch.exp(cos(spec_pred))
    contrastive_loss = -torch.log(pos / neg)
    loss_img = F.mse_loss(img_pred.float(), b["latents_img"].float(), reduction="mean")
    loss_mask = F.mse_loss(mask_pred.float(), b["latents_attr"].float(), reduction="mean")
    loss = loss_img + loss_mask * 10.0 + contrastive_loss * 0.01
    if inverse:
        noisy_latents_attr_c = torch.cat((b["latents_mask"].to(mask_pred.dtype), mask_pred), dim=1)
        res_c, mid_c, _, _ = controlnet(b["noisy_latents_img_c"], b["timesteps_attribute_c"], encoder_hidden_states=b["ehs"],
                                        controlnet_cond=noisy_latents_attr_c, **kw)
        img_pred_c = unet(b["noisy_latents_img_c"], b["timesteps_img_c"], encoder_hidden_states=b["ehs"],
                          down_block_additional_residuals=[s.to(dtype=wd) for s in res_c],
                          mid_block_additional_residual=mid_c.to(dtype=wd), **kw)[0]
        loss_c = F.mse_loss(img_pred_c.float(), b["latents_img"].float(), reduction="mean")
        loss = loss_img + loss_mask + 0.8 * loss_c
    return loss


def test_deferred_grouped_weight_gradients_are_the_immediate_ones(dev, monkeypatch):
    """Linear weight / bias gradients deferred to the CastParams / ParamBarrier nodes and computed in grouped launches
    (backward.WgradQueue) against the same step with every Linear computing its own at once: every parameter gradient of the
    three networks, both objectives of the step (the inverse branch runs enc + unet twice: two CastParams nodes per network)."""
    from uni_renderer_amd import backward as B_
    from uni_renderer_amd.train_step import train_step

    oracle = O.build_triplet(O.TINY_CONFIG, seed=41)
    x, c, ehs, ti, ta = [t.to(dev) for t in O.make_inputs(2, 16, 64, seed=17)]
    g = torch.Generator().manual_seed(18)
    batch = dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=torch.randn(2, 4, 16, 16, generator=g).to(dev),
                 target_attr=torch.randn(2, 28, 16, 16, generator=g).to(dev))
    grads = {}
    for flag in (True, False):
        monkeypatch.setattr(B_, "WGRAD_DEFER", flag)
        nets = build_product_from_oracle(*oracle, torch.float32, dev)
        for m in nets:
            m.train()
            m.requires_grad_(True)
        B_.wgrad_queue.trace = {} if flag else None
        stats = train_step(nets, batch, optimizer=None, dtype=torch.bfloat16, max_grad_norm=None)
        grads[flag] = ([p.grad.clone() for m in nets for p in m.parameters()], stats["loss"])
        if flag:
            groups, B_.wgrad_queue.trace = B_.wgrad_queue.trace, None
            assert groups and max(int(k.split("@")[1]) for k in groups) > 1, groups   # something was actually grouped
            assert not B_.wgrad_queue.items
    assert grads[True][1] == grads[False][1]
    worst = 0.0
    for a, b in zip(grads[True][0], grads[False][0]):
        # same kernels on the same operands; only the slice count (and so the fp32 summation order) may differ
        worst = max(worst, (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6))
    print({"deferred_vs_immediate_max_rel": worst})
    assert worst <= 1e-2


def _train_batch(B, L, cross, seed):
    x, c, ehs, ti, ta = O.make_inputs(B, L, cross, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    return dict(noisy_latents_img=x, noisy_latents_attr=c, ehs=ehs, timesteps_img=ti, timesteps_attribute=ta,
                latents_img=torch.randn(B, 4, L, L, generator=g), latents_attr=torch.randn(B, 24, L, L, generator=g),
                latents_mask=c[:, :4].clone(), noisy_latents_img_c=torch.randn(B, 4, L, L, generator=g),
                timesteps_img_c=torch.randint(0, 1000, (B,), generator=g), timesteps_attribute_c=torch.zeros(B).long())


def _grad_error(oracle, nets, sample_every=1):
    num = den = 0.0
    for mo, mp in zip(oracle, nets):
        po = dict(mo.named_parameters())
        for i, (name, p) in enumerate(mp.named_parameters()):
            go = po[name].grad
            if go is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
                continue
            assert p.grad is not None, name
            if i % sample_every:
                continue
            d = p.grad.float().cpu() - go
            num += float((d * d).sum())
            den += float((go * go).sum())
    return (num / den) ** 0.5


@pytest.mark.parametrize("inverse", [True, False])
def test_reference_train_step_on_the_module_surface_under_autocast(dev, inverse):
    """train/train.py:1324-1416 run LITERALLY on the product modules -- the three ``forward``s with the
    reference's keywords, fp32 master parameters under ``torch.autocast`` (train.py:882-887, 1082-1089), the
    ``.to(dtype=weight_dtype)`` casts, the losses, then ``loss.backward()`` -- must give the oracle's loss and
    parameter gradients.  (The module forwards route to the autograd path whenever grad is recording.)"""
    oracle = O.build_triplet(O.TINY_CONFIG, seed=35)
    b = _train_batch(2, 16, 64, seed=22)
    for m in oracle:
        m.requires_grad_(True)
    unet_o, enc_o, dec_o = oracle
    loss_o = _reference_step_losses(enc_o, unet_o, dec_o, dict(b, oracle=True, weight_dtype=torch.float32), inverse)
    loss_o.backward()
    nets = build_product_from_oracle(*oracle, torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    unet, enc, dec = nets
    bg = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    with torch.autocast("cuda", dtype=torch.float16):
        loss = _reference_step_losses(enc, unet, dec, dict(bg, weight_dtype=torch.float16), inverse)
    assert loss.requires_grad
    loss.backward()
    err = _grad_error(oracle, nets)
    print({"module_surface": True, "inverse": inverse, "loss": float(loss.detach()), "loss_oracle": float(loss_o.detach()),
           "grad_rel_l2_all": err})
    assert abs(float(loss.detach()) - float(loss_o.detach())) / abs(float(loss_o.detach())) < 5e-3
    assert err < 1.5e-2
    # and with autograd off the same modules take the fused inference path (no graph is recorded)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        r = enc(bg["noisy_latents_img"], bg["timesteps_attribute"], encoder_hidden_states=bg["ehs"],
                controlnet_cond=bg["noisy_latents_attr"], return_dict=False)
    assert not r[1].requires_grad and r[1].grad_fn is None


def test_cfg4_per_gpu_training_step_vs_oracle(dev):
    """cfg 4's per-GPU workload: SD-1.x-size networks, batch 4, 512x512 image = 64x64 latent, bf16 autocast over fp32
    master parameters, the reference's rendering-branch objective (mse + 10 mse + 0.01 contrastive on samples 0 / 1,
    train.py:1356-1378) through the module call surface.  Loss and parameter gradients (every 7th parameter tensor of
    each network, ~250 M elements) against the CPU oracle's autograd.  The oracle runs the batch as two half batches
    (samples are independent except for the contrastive pair, which lives in the first half) to bound host memory:
    loss = (L_A + L_B) / 2 with the contrastive term only in L_A at weight 2 x 0.01 / 2."""
    F = torch.nn.functional
    oracle = O.build_triplet(O.SD15_CONFIG, seed=42)
    b = _train_batch(4, 64, 768, seed=24)
    for m in oracle:
        m.requires_grad_(True)
    unet_o, enc_o, dec_o = oracle
    loss_o = 0.0
    for h in range(2):
        sl = slice(2 * h, 2 * h + 2)
        bh = {k: (v[sl] if torch.is_tensor(v) else v) for k, v in b.items()}
        res, mid, raw_enc, raw_mid_enc = enc_o(bh["noisy_latents_img"], bh["timesteps_attribute"], bh["ehs"],
                                               controlnet_cond=bh["noisy_latents_attr"])
        img_pred, raw_unet, raw_mid_unet, _ = unet_o(bh["noisy_latents_img"], bh["timesteps_img"], bh["ehs"],
                                                     down_block_additional_residuals=res, mid_block_additional_residual=mid)
        mask_pred = dec_o(raw_mid_enc, raw_enc, bh["timesteps_attribute"], bh["ehs"], down_block_additional_residuals=raw_unet,
                          mid_block_additional_residual=raw_mid_unet)[:, 4:]
        lh = 0.5 * (F.mse_loss(img_pred, bh["latents_img"]) + 10.0 * F.mse_loss(mask_pred, bh["latents_attr"]))
        if h == 0:
            cos = lambda a: F.cosine_similarity(a[0].reshape(-1), a[1].reshape(-1), dim=0) / 0.1
            pos = torch.exp(cos(mask_pred[:, 8:12]))
            lh = lh - 0.01 * torch.log(pos / (pos + torch.exp(cos(mask_pred[:, :4])) + torch.exp(cos(mask_pred[:, 12:16]))))
        lh.backward()
        loss_o += float(lh.detach())
        del res, mid, raw_enc, raw_mid_enc, img_pred, raw_unet, raw_mid_unet, mask_pred, lh
    nets = build_product_from_oracle(*oracle, torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    unet, enc, dec = nets
    bg = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = _reference_step_losses(enc, unet, dec, dict(bg, weight_dtype=torch.bfloat16), inverse=False)
    loss.backward()
    err = _grad_error(oracle, nets, sample_every=7)
    print({"cfg4_per_gpu_shape": "B=4, 64x64 latent, bf16 autocast, SD size", "loss": float(loss.detach()), "loss_oracle": loss_o,
           "grad_rel_l2_sampled": err})
    assert abs(float(loss.detach()) - loss_o) / abs(loss_o) < 2e-3  # measured 1e-5 (11.1493 vs 11.1494)
    assert err < 2e-2  # measured 0.9e-2 (r02)


def test_cfg4_inverse_branch_at_sd_size_vs_oracle(dev):
    """cfg 4's OTHER branch at SD size (train/train.py:1388-1416): the cycle-consistency pass -- a second encoder + UNet
    forward conditioned on [mask | the decoder's own prediction] with the gradient flowing through the prediction --
    SD-1.x-size networks, batch 2, 64x64 latent, bf16 autocast over fp32 master parameters, through the module call
    surface.  Loss and sampled parameter gradients against the CPU oracle's autograd of the same function."""
    oracle = O.build_triplet(O.SD15_CONFIG, seed=43)
    b = _train_batch(2, 64, 768, seed=27)
    for m in oracle:
        m.requires_grad_(True)
    unet_o, enc_o, dec_o = oracle
    loss_o = _reference_step_losses(enc_o, unet_o, dec_o, dict(b, oracle=True, weight_dtype=torch.float32), inverse=True)
    loss_o.backward()
    loss_o = float(loss_o.detach())
    nets = build_product_from_oracle(*oracle, torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    unet, enc, dec = nets
    bg = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = _reference_step_losses(enc, unet, dec, dict(bg, weight_dtype=torch.bfloat16), inverse=True)
    loss.backward()
    err = _grad_error(oracle, nets, sample_every=7)
    print({"cfg4_inverse_branch": "B=2, 64x64 latent, bf16 autocast, SD size", "loss": float(loss.detach()), "loss_oracle": loss_o,
           "grad_rel_l2_sampled": err})
    assert abs(float(loss.detach()) - loss_o) / abs(loss_o) < 5e-3
    assert err < 3e-2


def _ddp_worker(rank, world, port, q):
    """one rank of the two-process training test below (both ranks share cuda:0; gloo carries the collectives, RCCL
    refuses two ranks on one device)"""
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist

    from uni_renderer_amd.parallel import GradientBuckets
    from uni_renderer_amd.train_step import train_step

    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        oracle = O.build_triplet(O.TINY_CONFIG, seed=36)
        nets = build_product_from_oracle(*oracle, torch.float32, dev)
        for m in nets:
            m.train()
            m.requires_grad_(True)
        opt = torch.optim.AdamW([p for m in nets for p in m.parameters()], lr=2e-4)
        buckets = GradientBuckets(nets, bucket_mb=0.5, comm_dtype=None)
        losses = []
        for it in range(2):
            x, c, ehs, ti, ta = [t.to(dev) for t in O.make_inputs(2, 16, 64, seed=40 + 2 * it + rank)]
            g = torch.Generator().manual_seed(50 + 2 * it + rank)
            batch = dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=torch.randn(2, 4, 16, 16, generator=g).to(dev),
                         target_attr=torch.randn(2, 24, 16, 16, generator=g).to(dev),
                         x_t_c=torch.randn(2, 4, 16, 16, generator=g).to(dev), t_img_c=torch.randint(0, 1000, (2,), generator=g).to(dev))
            # divergent branches (compute_t, train.py:445): the ranks disagree about inverse rendering in both iterations
            losses.append(train_step(nets, batch, optimizer=opt, buckets=buckets, dtype=torch.bfloat16, inverse=bool((rank + it) % 2))["loss"])
        flat = torch.cat([p.detach().reshape(-1) for m in nets for p in m.parameters()]).cpu()
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        q.put((rank, bool(torch.equal(both[0], both[1])), buckets.launched_from_hooks, len(buckets.buckets), losses))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # report instead of hanging the parent
        q.put((rank, repr(e), 0, 0, []))


def test_two_rank_training_on_the_real_modules(dev):
    """Multi-rank path of cfg 4 on the real networks and HIP backward kernels (tiny config): two processes, each its own
    micro-batch and its own branch of the objective, gradients in the flat buckets, bucket collectives enqueued from
    autograd hooks while the backward is running, identical parameters on both ranks after two optimizer steps."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    print({"two_rank_training": res})
    if any(isinstance(r[1], str) and "gloo" in r[1].lower() and "cuda" in r[1].lower() for r in res):
        pytest.skip("this torch build's gloo backend does not take device tensors")
    assert [(r[0], r[1]) for r in res] == [(0, True), (1, True)], res
    assert all(r[3] > 4 for r in res)

@pytest.mark.gpu
def test_fused_adamw_matches_torch(dev):
    """optim.FusedAdamW (ur_adamw_multi) against torch.optim.AdamW over five steps: > 64 tensors (several launches), odd
    sizes, a gradient that is an unaligned view of a flat buffer (scalar path), clipping folded through ``grad_scale``;
    state_dict round trip into torch's optimizer."""
    from uni_renderer_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(320, 320), (1280,), (7,), (33, 5), (640, 3, 3, 3), (1,)] * 12  # 72 tensors
    ref_p = [torch.randn(*s, generator=g).to(dev).requires_grad_() for s in shapes]
    our_p = [p.detach().clone().requires_grad_() for p in ref_p]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    ref, our = torch.optim.AdamW(ref_p, **kw), FusedAdamW(our_p, **kw)
    flat = torch.zeros(sum(p.numel() for p in our_p) + 1, device=dev)
    for it in range(5):
        off = 1  # views at odd element offsets: not 16-byte aligned
        scale = torch.tensor(1.0 + 0.5 * it, device=dev)
        for a, b in zip(ref_p, our_p):
            gr = torch.randn(a.shape, generator=g).to(dev)
            a.grad = gr / scale
            view = flat[off:off + b.numel()].view_as(b)
            view.copy_(gr)
            b.grad = view if it % 2 else gr.clone()
            off += b.numel()
        our.grad_scale, our.found_inf = scale, torch.zeros((), device=dev)
        ref.step()
        our.step()
    worst = max(float((a - b).abs().max()) for a, b in zip(ref_p, our_p))  # parameters are O(1), updates O(lr) = 1e-2
    print({"fused_adamw_vs_torch_abs_max_after_5_steps": worst})
    assert worst < 5e-6
    del our.grad_scale, our.found_inf
    import copy
    sd = copy.deepcopy(our.state_dict())  # load_state_dict keeps same-dtype tensors uncopied: no sharing between optimizers
    ref2 = torch.optim.AdamW([p.detach().clone().requires_grad_() for p in our_p], **kw)
    ref2.load_state_dict(sd)
    st = ref2.state[ref2.param_groups[0]["params"][0]]
    assert float(st["step"]) == 5.0 and torch.equal(st["exp_avg"], our.state[our_p[0]]["exp_avg"])
    our2 = FusedAdamW([p.detach().clone().requires_grad_() for p in our_p], **kw)
    our2.load_state_dict(copy.deepcopy(ref2.state_dict()))
    for a, b in zip(our2.param_groups[0]["params"], ref2.param_groups[0]["params"]):
        a.grad = torch.ones_like(a)
        b.grad = torch.ones_like(b)
    our2.step()
    ref2.step()
    worst = max(float((a - b).abs().max()) for a, b in zip(our2.param_groups[0]["params"], ref2.param_groups[0]["params"]))
    assert worst < 1e-5, (worst, float(our2.state[our2.param_groups[0]['params'][0]]['step']), float(ref2.state[ref2.param_groups[0]['params'][0]]['step']))

def _graphed_worker(rank, world, port, q):
    """one rank of the two-process GraphedTrainStep test (both ranks on cuda:0, gloo collectives between the graphs)"""
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist

    from uni_renderer_amd.optim import FusedAdamW
    from uni_renderer_amd.parallel import GradientBuckets
    from uni_renderer_amd.train_step import GraphedTrainStep, train_step

    try:
        import time
        t0, marks = time.time(), []
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")

        def setup():
            oracle = O.build_triplet(O.TINY_CONFIG, seed=36)
            nets = build_product_from_oracle(*oracle, torch.float32, dev)
            for m in nets:
                m.train()
                m.requires_grad_(True)
            return nets, FusedAdamW([p for m in nets for p in m.parameters()], lr=2e-4)

        def make_batch(it):
            x, c, ehs, ti, ta = [t.to(dev) for t in O.make_inputs(2, 16, 64, seed=40 + 2 * it + rank)]
            g = torch.Generator().manual_seed(50 + 2 * it + rank)
            return dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=torch.randn(2, 4, 16, 16, generator=g).to(dev),
                        target_attr=torch.randn(2, 28, 16, 16, generator=g).to(dev))

        nets_e, opt_e = setup()
        b_e = GradientBuckets(nets_e, bucket_mb=32.0, comm_dtype=None, overlap=False)
        for it in range(3):
            train_step(nets_e, make_batch(it), optimizer=opt_e, buckets=b_e, dtype=torch.bfloat16)
        marks.append(round(time.time() - t0, 1))
        nets_g, opt_g = setup()
        b_g = GradientBuckets(nets_g, bucket_mb=32.0, comm_dtype=None, overlap=False)
        step = GraphedTrainStep(nets_g, make_batch(0), opt_g, buckets=b_g, dtype=torch.bfloat16, warmup=0)
        for it in range(3):
            st = step.step(make_batch(it))
        marks.append(round(time.time() - t0, 1))
        flat_e = torch.cat([p.detach().reshape(-1) for m in nets_e for p in m.parameters()])
        flat_g = torch.cat([p.detach().reshape(-1) for m in nets_g for p in m.parameters()])
        same = bool(torch.equal(flat_e, flat_g))
        worst = float((flat_e - flat_g).abs().max())
        both = [torch.empty_like(flat_g.cpu()) for _ in range(world)]
        dist.all_gather(both, flat_g.cpu())
        marks.append(round(time.time() - t0, 1))
        q.put((rank, same, worst, bool(torch.equal(both[0], both[1])), float(st["loss"]), float(st["grad_norm"]), marks))
        dist.barrier()
        torch.cuda.synchronize()
        q.close()
        q.join_thread()
        os._exit(0)  # tearing down gloo + the interpreter with live HIP graphs was seen to stall for minutes: just leave
    except Exception as e:  # report instead of hanging the parent
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()[-600:], 0.0, False, 0.0, 0.0, []))


@pytest.mark.gpu
def test_graphed_train_step_two_ranks(dev):
    """train_step.GraphedTrainStep on two ranks: forward + backward as one graph writing into the flat buckets, eager
    collectives and clipping + FusedAdamW between the replays -- same parameters after three steps as the eager
    bucketed step, identical on both ranks."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_graphed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(30)
        if p.is_alive():
            p.terminate()
    print({"graphed_two_rank_training": res})
    if any(isinstance(r[1], str) and "gloo" in r[1].lower() and "cuda" in r[1].lower() for r in res):
        pytest.skip("this torch build's gloo backend does not take device tensors")
    for r in res:
        assert not isinstance(r[1], str), r
        assert r[3], res          # both ranks hold the same parameters
        assert r[2] < 1e-6, res   # and they are the eager trajectory (bit-identical in practice)


@pytest.mark.gpu
def test_graphed_train_step_single_gpu(dev):
    """GraphedTrainStep without buckets = the whole step as one graph; new batches are copied into the captured inputs."""
    from uni_renderer_amd.optim import FusedAdamW
    from uni_renderer_amd.train_step import GraphedTrainStep, train_step

    def setup():
        oracle = O.build_triplet(O.TINY_CONFIG, seed=34)
        nets = build_product_from_oracle(*oracle, torch.float32, dev)
        for m in nets:
            m.train()
            m.requires_grad_(True)
        return nets, FusedAdamW([p for m in nets for p in m.parameters()], lr=2e-4)

    def make_batch(it):
        x, c, ehs, ti, ta = [t.to(dev) for t in O.make_inputs(2, 16, 64, seed=20 + it)]
        g = torch.Generator().manual_seed(21 + it)
        return dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=torch.randn(2, 4, 16, 16, generator=g).to(dev),
                    target_attr=torch.randn(2, 28, 16, 16, generator=g).to(dev))

    nets_e, opt_e = setup()
    eager = [train_step(nets_e, make_batch(it), optimizer=opt_e, dtype=torch.bfloat16)["loss"] for it in range(4)]
    nets_g, opt_g = setup()
    step = GraphedTrainStep(nets_g, make_batch(0), opt_g, dtype=torch.bfloat16, warmup=0)
    graphed = [float(step.step(make_batch(it))["loss"]) for it in range(4)]
    print({"eager": eager, "graphed": graphed})
    assert eager == graphed
    for a, b in zip((p for m in nets_e for p in m.parameters()), (p for m in nets_g for p in m.parameters())):
        assert torch.equal(a, b)


def test_graphed_train_step_follows_an_lr_schedule_without_recapture(dev):
    """with a non-constant --lr_scheduler (train.py offers linear / cosine / *_with_warmup) the captured step
    used to be re-captured on EVERY step (lr was a by-value kernel argument).  ``ur_adamw_multi`` now reads lr / weight
    decay from a device pair that FusedAdamW.sync_hyper refreshes before the replay: a LambdaLR schedule over the graphed
    step gives bit-identical parameters to the eager step, with ONE capture."""
    from uni_renderer_amd.optim import FusedAdamW
    from uni_renderer_amd.train_step import GraphedTrainStep, train_step

    def setup():
        oracle = O.build_triplet(O.TINY_CONFIG, seed=38)
        nets = build_product_from_oracle(*oracle, torch.float32, dev)
        for m in nets:
            m.train()
            m.requires_grad_(True)
        opt = FusedAdamW([p for m in nets for p in m.parameters()], lr=4e-4, weight_decay=1e-2)
        return nets, opt, torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0 / (1.0 + it))

    def make_batch(it):
        x, c, ehs, ti, ta = [t.to(dev) for t in O.make_inputs(2, 16, 64, seed=70 + it)]
        g = torch.Generator().manual_seed(71 + it)
        return dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=torch.randn(2, 4, 16, 16, generator=g).to(dev),
                    target_attr=torch.randn(2, 28, 16, 16, generator=g).to(dev))

    nets_e, opt_e, sch_e = setup()
    eager = []
    for it in range(4):
        eager.append(train_step(nets_e, make_batch(it), optimizer=opt_e, dtype=torch.bfloat16)["loss"])
        sch_e.step()
    nets_g, opt_g, sch_g = setup()
    step = GraphedTrainStep(nets_g, make_batch(0), opt_g, dtype=torch.bfloat16, warmup=0)
    first_graph = step.g_fb
    graphed = []
    for it in range(4):
        graphed.append(float(step.step(make_batch(it))["loss"]))
        sch_g.step()
    assert step.g_fb is first_graph, "the step was re-captured although only lr changed"
    assert opt_g.param_groups[0]["lr"] == opt_e.param_groups[0]["lr"] != 4e-4
    assert eager == graphed
    for a, b in zip((p for m in nets_e for p in m.parameters()), (p for m in nets_g for p in m.parameters())):
        assert torch.equal(a, b)


def test_reference_fp16_amp_recipe_at_sd_size_and_an_overflow_step(dev):
    """VERDICT r3 'missing' 3: the reference's ACTUAL training precision is fp16 AMP with a GradScaler (train/train.sh:21
    ``--mixed_precision="fp16"``; train.py:882-887; accelerate: ``scaler.scale(loss).backward()``, ``unscale_`` inside
    ``clip_grad_norm_``, ``scaler.step``, ``scaler.update``).  SD-1.x-size networks, batch 2, 64x64 latent, the
    rendering-branch objective through the module call surface under ``torch.autocast(float16)`` with fp32 master
    parameters: (a) loss and unscaled sampled gradients against the CPU oracle's autograd; (b) one real optimisation step
    with ``optim.FusedAdamW``; (c) a step whose targets hold an inf: the scaler finds the overflow, parameters, moments
    and the step counter stay bit-identical, the scale is halved."""
    from uni_renderer_amd.optim import FusedAdamW

    oracle = O.build_triplet(O.SD15_CONFIG, seed=44)
    b = _train_batch(2, 64, 768, seed=29)
    for m in oracle:
        m.requires_grad_(True)
    unet_o, enc_o, dec_o = oracle
    loss_o = _reference_step_losses(enc_o, unet_o, dec_o, dict(b, oracle=True, weight_dtype=torch.float32), inverse=False)
    loss_o.backward()
    loss_o = float(loss_o.detach())
    nets = build_product_from_oracle(*oracle, torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    unet, enc, dec = nets
    params = [p for m in nets for p in m.parameters()]
    opt = FusedAdamW(params, lr=5e-6, weight_decay=1e-2)  # train.sh:36 learning rate
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=2000)
    bg = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    with torch.autocast("cuda", dtype=torch.float16):
        loss = _reference_step_losses(enc, unet, dec, dict(bg, weight_dtype=torch.float16), inverse=False)
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    err = _grad_error(oracle, nets, sample_every=7)
    found = sum(float(v) for v in scaler._per_optimizer_states[id(opt)]["found_inf_per_device"].values())
    print({"fp16_amp_sd_size": "B=2, 64x64 latent, fp16 autocast + GradScaler(1024)", "loss": float(loss.detach()),
           "loss_oracle": loss_o, "grad_rel_l2_sampled_after_unscale": err, "found_inf": found})
    assert found == 0.0
    assert abs(float(loss.detach()) - loss_o) / abs(loss_o) < 2e-3
    assert err < 1.5e-2
    before = [p.detach().clone() for p in params[:40]]
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    scaler.step(opt)
    scaler.update()
    step_t = opt.state[params[0]]["step"]
    assert float(step_t) == 1.0 and float(scaler.get_scale()) == 1024.0
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, params[:40]))
    # ---- (c) the overflow step
    snap = [p.detach().clone() for p in params]
    mom = [opt.state[p]["exp_avg"].clone() for p in params[:40]]
    bad = dict(bg)
    bad["latents_img"] = bg["latents_img"].clone()
    bad["latents_img"][0, 0, 0, 0] = float("inf")
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        loss2 = _reference_step_losses(enc, unet, dec, dict(bad, weight_dtype=torch.float16), inverse=False)
    scaler.scale(loss2).backward()
    scaler.unscale_(opt)
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    scaler.step(opt)
    scaler.update()
    assert not torch.isfinite(loss2.detach())
    assert all(torch.equal(a, p.detach()) for a, p in zip(snap, params))             # parameters untouched
    assert all(torch.equal(a, opt.state[p]["exp_avg"]) for a, p in zip(mom, params[:40]))  # moments untouched
    assert float(step_t) == 1.0                                                       # bias-correction counter untouched
    assert float(scaler.get_scale()) == 512.0


def test_train_step_with_grad_scaler_skips_on_overflow_and_accumulates(dev):
    """The same two features through the package's own step function (train_step.train_step): ``scaler=`` runs the
    reference's unscale -> clip -> scaler.step -> update order; ``grad_accum=(k, n)`` zeroes at k = 0, divides the loss by
    n and updates at k = n - 1 (accelerator.accumulate, train.py:1236) -- two micro-steps of batch 2 must leave the
    gradients of one step over the mean of the two losses."""
    from uni_renderer_amd.optim import FusedAdamW
    from uni_renderer_amd.train_step import train_step

    oracle = O.build_triplet(O.TINY_CONFIG, seed=37)
    nets = build_product_from_oracle(*oracle, torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    params = [p for m in nets for p in m.parameters()]

    def batch(seed, poison=False):
        x, c, ehs, ti, ta = [t.to(dev) for t in O.make_inputs(2, 16, 64, seed=seed)]
        g = torch.Generator().manual_seed(seed + 1)
        tgt = torch.randn(2, 4, 16, 16, generator=g).to(dev)
        if poison:
            tgt[0, 0, 0, 0] = float("inf")
        return dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=tgt,
                    target_attr=torch.randn(2, 28, 16, 16, generator=g).to(dev))

    # accumulation: grads after (0, 2) + (1, 2) == grads of 0.5 * (loss_a + loss_b), no optimizer involved.  In bf16:
    # halving the loss is exact through every bf16 gradient tensor (no underflow), so the identity holds to fp32 rounding;
    # an UNSCALED fp16 backward of loss / 2 loses small gradients to fp16's subnormals (measured 4.6e-3 on this test),
    # which is what the GradScaler of the second half is for
    train_step(nets, batch(60), optimizer=None, dtype=torch.bfloat16, max_grad_norm=None, grad_accum=(0, 2))
    g_first = [p.grad.detach().clone() for p in params]
    train_step(nets, batch(62), optimizer=None, dtype=torch.bfloat16, max_grad_norm=None, grad_accum=(1, 2))
    acc = [p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    train_step(nets, batch(62), optimizer=None, dtype=torch.bfloat16, max_grad_norm=None)
    second = [p.grad.detach().clone() for p in params]
    num = sum(float(((a - (f + 0.5 * s)) ** 2).sum()) for a, f, s in zip(acc, g_first, second))
    den = sum(float((a ** 2).sum()) for a in acc)
    print({"grad_accum_rel_l2": (num / den) ** 0.5})
    assert (num / den) ** 0.5 < 1e-6
    # GradScaler through train_step: a clean step updates, a poisoned step is skipped
    for p in params:
        p.grad = None
    opt = FusedAdamW(params, lr=1e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=256.0)
    train_step(nets, batch(64), optimizer=opt, dtype=torch.float16, scaler=scaler)
    step_t = opt.state[params[0]]["step"]
    assert float(step_t) == 1.0
    snap = [p.detach().clone() for p in params]
    train_step(nets, batch(66, poison=True), optimizer=opt, dtype=torch.float16, scaler=scaler)
    assert all(torch.equal(a, p.detach()) for a, p in zip(snap, params))
    assert float(step_t) == 1.0 and float(scaler.get_scale()) == 128.0


def _torch_ddp_worker(rank, world, port, q):
    """one rank of the DistributedDataParallel test below (both ranks share cuda:0, gloo carries the collectives)"""
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        # DIFFERENT seeds per rank: the DDP constructors broadcast rank 0's parameters (train.py:1140-1142)
        oracle = O.build_triplet(O.TINY_CONFIG, seed=70 + rank)
        nets = build_product_from_oracle(*oracle, torch.float32, dev)
        for m in nets:
            m.train()
            m.requires_grad_(True)
        unet, enc, dec = nets
        shards = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in _train_batch(2, 16, 64, seed=80 + r).items()}
                  for r in range(world)]
        enc_d, dec_d, unet_d = DDP(enc, device_ids=[0]), DDP(dec, device_ids=[0]), DDP(unet, device_ids=[0])
        params = [p for m in nets for p in m.parameters()]
        # expected: mean over the ranks' shards of the single-process gradients (after the broadcast: same parameters)
        want = [torch.zeros_like(p) for p in params]
        for r in range(world):
            for p in params:
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                l = _reference_step_losses(enc, unet, dec, dict(shards[r], weight_dtype=torch.bfloat16), inverse=bool(r % 2))
            l.backward()
            for w, p in zip(want, params):
                if p.grad is not None:
                    w += p.grad / world
        for p in params:
            p.grad = None
        # train.py:1324-1427 on the WRAPPED modules; the ranks take different branches of the objective (compute_t, 445)
        opt = torch.optim.AdamW(params, lr=1e-4)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = _reference_step_losses(enc_d, unet_d, dec_d, dict(shards[rank], weight_dtype=torch.bfloat16), inverse=bool(rank % 2))
        loss.backward()
        num = sum(float(((p.grad - w) ** 2).sum()) for p, w in zip(params, want) if p.grad is not None)
        den = sum(float((w ** 2).sum()) for w in want)
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        flat = torch.cat([p.detach().reshape(-1) for p in params]).cpu()
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        q.put((rank, bool(torch.equal(both[0], both[1])), (num / den) ** 0.5, float(loss.detach())))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # report instead of hanging the parent
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()[-1500:], 1.0, 0.0))


def test_product_modules_wrapped_in_torch_ddp(dev):
    """SURVEY 8b 'must be DDP-wrappable' / VERDICT r3 'missing' 1: the reference's only multi-GPU mechanism is three
    ``torch.nn.parallel.DistributedDataParallel`` wrappers (train/train.py:1140-1142).  Two processes wrap the PRODUCT
    modules exactly like that (different seeds per rank: DDP's constructor broadcast must make them equal), run
    train.py:1324-1427 on the wrappers under autocast with DIVERGENT objective branches per rank, and must end with
    (i) gradients equal to the mean of the single-process gradients of the two shards and (ii) bit-identical parameters
    after clip + AdamW on both ranks."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_torch_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    print({"torch_ddp_wrapped": [(r[0], r[1] if r[1] is True else str(r[1])[:300], r[2], r[3]) for r in res]})
    if any(isinstance(r[1], str) and "gloo" in r[1].lower() and "cuda" in r[1].lower() for r in res):
        pytest.skip("this torch build's gloo backend does not take device tensors")
    assert [(r[0], r[1]) for r in res] == [(0, True), (1, True)], res
    assert all(r[2] < 1e-5 for r in res), res


@pytest.mark.parametrize("inverse", [None, True])
def test_gradient_checkpointing_recomputes_the_resnets_bit_for_bit(dev, inverse):
    """Round 6 (VERDICT r5 'missing' 4; reference: train/train.py:1073-1074 -> controlnet.py:745-747 -> unet_2d_blocks.py:1172-1197):
    ``enable_gradient_checkpointing()`` makes the training forward run every ResnetBlock2D of the flagged blocks under
    ``torch.utils.checkpoint`` (use_reentrant=False): its activations are dropped and recomputed by the same HIP kernels in the
    backward.  Loss and EVERY parameter gradient must be bit-identical to the plain step -- the plain MSE objective and the
    reference's inverse (cycle-consistency) branch, which runs enc + unet twice -- and the resnets must really be recomputed
    (more conv launches in the checkpointed step)."""
    from uni_renderer_amd import ops
    from uni_renderer_amd.train_step import _forward_backward

    oracle = O.build_triplet(O.TINY_CONFIG, seed=41)
    x, c, ehs, ti, ta = O.make_inputs(2, 16, 64, seed=42)
    g = torch.Generator().manual_seed(43)
    batch = dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=torch.randn(2, 4, 16, 16, generator=g),
                 target_attr=torch.randn(2, 28 if inverse is None else 24, 16, 16, generator=g),
                 x_t_c=torch.randn(2, 4, 16, 16, generator=g), t_img_c=torch.randint(0, 1000, (2,), generator=g))
    batch = {k: v.to(dev) for k, v in batch.items()}
    nets = build_product_from_oracle(*oracle, torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)

    def run(ckpt):
        for m in nets:
            (m.enable_gradient_checkpointing if ckpt else m.disable_gradient_checkpointing)()
            for p in m.parameters():
                p.grad = None
        calls = []
        orig = ops.igemm
        ops.igemm = lambda **kw: (calls.append(kw.get("taps", 1)), orig(**kw))[1]
        try:
            loss = _forward_backward(nets, batch, None, None, torch.bfloat16, inverse)
        finally:
            ops.igemm = orig
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.clone() for m in nets for n, p in m.named_parameters() if p.grad is not None}, sum(t == 9 for t in calls)

    l0, g0, convs0 = run(False)
    l1, g1, convs1 = run(True)
    assert all(m.is_gradient_checkpointing for m in nets)
    assert convs1 > convs0, (convs0, convs1)  # the checkpointed resnets' convs ran again in the backward
    assert l0 == l1 and set(g0) == set(g1) and len(g0) > 100
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    for m in nets:
        m.disable_gradient_checkpointing()
