"""SURVEY 8f rank 4 on the HIP path: ``checkpoint-N`` save / resume-latest (reference train/train.py:1002-1045 hooks,
1191-1218 resume, 1434-1457 save) around GPU-resident modules, ``optim.FusedAdamW`` state and the CAPTURED training step.
Interrupted-and-resumed training must be bit-identical to uninterrupted training: parameters, both moments, the step
counter and the loss of the next step -- through a fresh set of objects (a new process would build them) and through the
live objects whose captured graph keeps replaying (the optimizer state is loaded in place, optim.FusedAdamW.load_state_dict)."""
import pytest
import torch

from util_models import O, build_product_from_oracle

pytestmark = pytest.mark.gpu
LR, BETAS = 3e-4, (0.9, 0.99)


def _setup(dev, seed):
    from uni_renderer_amd.optim import FusedAdamW

    nets = build_product_from_oracle(*O.build_triplet(O.TINY_CONFIG, seed=seed), torch.float32, dev)
    for m in nets:
        m.train()
        m.requires_grad_(True)
    return nets, FusedAdamW([p for m in nets for p in m.parameters()], lr=LR, betas=BETAS, weight_decay=1e-2)


def _batch(dev, it):
    x, c, ehs, ti, ta = [t.to(dev) for t in O.make_inputs(2, 16, 64, seed=90 + it)]
    g = torch.Generator().manual_seed(91 + it)
    return dict(x_t=x, cond=c, ehs=ehs, t_img=ti, t_attr=ta, target_img=torch.randn(2, 4, 16, 16, generator=g).to(dev),
                target_attr=torch.randn(2, 28, 16, 16, generator=g).to(dev))


def _snapshot(nets, opt):
    ps = [p for m in nets for p in m.parameters()]
    return ([p.detach().clone() for p in ps], [opt.state[p]["exp_avg"].clone() for p in ps],
            [opt.state[p]["exp_avg_sq"].clone() for p in ps], float(opt.state[ps[0]]["step"]))


def _same(a, b):
    assert a[3] == b[3]
    for xs, ys in zip(a[:3], b[:3]):
        assert len(xs) == len(ys) and all(torch.equal(x, y) for x, y in zip(xs, ys))


def test_graphed_training_resumes_bit_identically_from_a_checkpoint(dev, tmp_path):
    from uni_renderer_amd import checkpointing as C
    from uni_renderer_amd.train_step import GraphedTrainStep

    out = str(tmp_path / "run")
    # uninterrupted: three graphed steps
    nets_a, opt_a = _setup(dev, 60)
    step_a = GraphedTrainStep(nets_a, _batch(dev, 0), opt_a, dtype=torch.bfloat16, warmup=0)
    loss_a = [float(step_a.step(_batch(dev, it))["loss"]) for it in range(3)]
    want = _snapshot(nets_a, opt_a)
    assert want[3] == 3.0

    # interrupted after two steps
    nets_b, opt_b = _setup(dev, 60)
    step_b = GraphedTrainStep(nets_b, _batch(dev, 0), opt_b, dtype=torch.bfloat16, warmup=0)
    loss_b = [float(step_b.step(_batch(dev, it))["loss"]) for it in range(2)]
    assert loss_b == loss_a[:2]
    path = C.save_state(nets_b, out, 2, optimizer=opt_b, checkpoints_total_limit=2)
    assert path.endswith("checkpoint-2")

    # (i) a fresh run: other initial weights, fresh optimizer with other lr / betas; resume "latest" restores everything and
    # the CLI values overwrite lr / betas (train.py:1209-1211); the third step equals the uninterrupted one bit for bit
    nets_c, opt_c = _setup(dev, 61)
    for g in opt_c.param_groups:
        g["lr"], g["betas"] = 1.0, (0.5, 0.5)
    assert C.resume_from_checkpoint(nets_c, out, "latest", opt_c, learning_rate=LR, betas=BETAS) == 2
    assert all(p.is_cuda for m in nets_c for p in m.parameters())
    assert all(opt_c.state[p]["exp_avg"].is_cuda for m in nets_c for p in m.parameters())
    step_c = GraphedTrainStep(nets_c, _batch(dev, 0), opt_c, dtype=torch.bfloat16, warmup=0)
    assert float(step_c.step(_batch(dev, 2))["loss"]) == loss_a[2]
    _same(_snapshot(nets_c, opt_c), want)

    # (ii) the live objects: training runs on (two more steps at another lr), then the SAME modules / optimizer / captured
    # graph are rolled back to the checkpoint.  The optimizer state is loaded in place -- the replayed ur_adamw_multi keeps
    # its addresses -- so no re-capture is needed and the next step is again the uninterrupted third step
    for g in opt_b.param_groups:
        g["lr"] = 5e-3
    for it in (5, 6):
        step_b.step(_batch(dev, it))
    graph_before, gen = step_b.g_fb, opt_b.generation
    moments = [opt_b.state[p]["exp_avg"].data_ptr() for m in nets_b for p in m.parameters()]
    assert C.resume_from_checkpoint(nets_b, out, "latest", opt_b, learning_rate=LR, betas=BETAS) == 2
    assert opt_b.generation == gen and moments == [opt_b.state[p]["exp_avg"].data_ptr() for m in nets_b for p in m.parameters()]
    assert float(step_b.step(_batch(dev, 2))["loss"]) == loss_a[2]
    assert step_b.g_fb is graph_before, "resume forced a re-capture although no address changed"
    _same(_snapshot(nets_b, opt_b), want)

    # rotation on the GPU path: a third and fourth save with limit 2 keep the two newest
    C.save_state(nets_b, out, 3, optimizer=opt_b, checkpoints_total_limit=2)
    C.save_state(nets_b, out, 4, optimizer=opt_b, checkpoints_total_limit=2)
    assert C.list_checkpoints(out) == ["checkpoint-3", "checkpoint-4"]


def test_eager_training_with_torch_adamw_resumes_from_the_same_checkpoint_layout(dev, tmp_path):
    """``optimizer.bin`` is torch.optim.AdamW's layout: a run that saved with FusedAdamW resumes into torch's fused AdamW (and
    the eager train_step) with the same parameters / moments."""
    from uni_renderer_amd import checkpointing as C
    from uni_renderer_amd.train_step import train_step

    out = str(tmp_path / "run")
    nets, opt = _setup(dev, 62)
    for it in range(2):
        train_step(nets, _batch(dev, it), optimizer=opt, dtype=torch.bfloat16)
    C.save_state(nets, out, 2, optimizer=opt)
    nets2, _ = _setup(dev, 63)
    opt2 = torch.optim.AdamW([p for m in nets2 for p in m.parameters()], lr=LR, betas=BETAS, weight_decay=1e-2, fused=True)
    assert C.resume_from_checkpoint(nets2, out, "checkpoint-2", opt2) == 2
    for a, b in zip((p for m in nets for p in m.parameters()), (p for m in nets2 for p in m.parameters())):
        assert torch.equal(a, b)
        assert torch.equal(opt.state[a]["exp_avg"], opt2.state[b]["exp_avg"].to(a.device))
        assert float(opt2.state[b]["step"]) == 2.0
    l1 = train_step(nets, _batch(dev, 2), optimizer=opt, dtype=torch.bfloat16)["loss"]
    l2 = train_step(nets2, _batch(dev, 2), optimizer=opt2, dtype=torch.bfloat16)["loss"]
    assert l1 == l2  # same parameters -> same forward
