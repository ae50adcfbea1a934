"""CPU tests of the training-state checkpoints (uni_renderer_amd/checkpointing.py; reference train/train.py:1002-1045
save / load hooks, 1191-1218 resume-latest, 1434-1457 rotation): on-disk layout, rotation arithmetic, resume of the
latest ``checkpoint-N`` including the channel-surgery shapes and the lr / betas overwrite."""
import json
import os

import torch

from util_models import O, build_product_from_oracle


def _nets(seed):
    return build_product_from_oracle(*O.build_triplet(O.TINY_CONFIG, seed=seed), torch.float32)


def test_layout_rotation_and_resume_latest(tmp_path):
    from uni_renderer_amd import checkpointing as C

    out = str(tmp_path / "run")
    nets = _nets(50)
    opt = torch.optim.AdamW([p for m in nets for p in m.parameters()], lr=5e-6, betas=(0.9, 0.999))
    # give the optimizer some state so optimizer.bin is not trivial
    for p in list(nets[0].parameters())[:3]:
        p.grad = torch.ones_like(p)
    opt.step()
    assert C.resume_from_checkpoint(nets, out, "latest", opt) == 0  # nothing there: fresh run (train.py:1202-1206)
    saved = []
    for step in (5000, 10000, 15000, 20000):
        with torch.no_grad():
            nets[0].conv_in.bias.fill_(float(step))  # a value that identifies the checkpoint
        saved.append(C.save_state(nets, out, step, optimizer=opt, checkpoints_total_limit=3))
        # train.py:1436-1453: before saving keep at most limit-1 old ones -> never more than `limit` on disk
        assert C.list_checkpoints(out) == [f"checkpoint-{s}" for s in (5000, 10000, 15000, 20000) if s <= step][-3:]
    ck = os.path.join(out, "checkpoint-20000")
    assert sorted(os.listdir(ck)) == ["controldec", "controlnet", "optimizer.bin", "unet"]  # sub-folders by class name
    for sub, cls, chans in (("controlnet", "AttributeEncoderModel", ("in_channels", 28)),
                            ("controldec", "AttributeDecoderModel", ("out_channels", 28)),
                            ("unet", "UNet2DConditionModel", ("in_channels", 4))):
        assert sorted(os.listdir(os.path.join(ck, sub))) == ["config.json", "diffusion_pytorch_model.safetensors"]
        cfg = json.load(open(os.path.join(ck, sub, "config.json")))
        assert cfg["_class_name"] == cls and cfg[chans[0]] == chans[1]
    assert C.save_state(nets, out, 25000, is_main_process=False) is None and len(C.list_checkpoints(out)) == 3

    # a different run resumes "latest": weights, surgery shapes, optimizer state; lr / betas come from the caller
    nets2 = _nets(51)
    opt2 = torch.optim.AdamW([p for m in nets2 for p in m.parameters()], lr=1.0, betas=(0.5, 0.5))
    ids = [id(m) for m in nets2]
    step = C.resume_from_checkpoint(nets2, out, "latest", opt2, learning_rate=5e-6, betas=(0.9, 0.999))
    assert step == 20000 and [id(m) for m in nets2] == ids  # loaded in place
    for a, b in zip(nets, nets2):
        sa, sb = a.state_dict(), b.state_dict()
        assert sa.keys() == sb.keys() and all(torch.equal(sa[k], sb[k]) for k in sa)
    assert float(nets2[0].conv_in.bias[0]) == 20000.0
    assert all(g["lr"] == 5e-6 and g["betas"] == (0.9, 0.999) for g in opt2.param_groups)
    assert len(opt2.state_dict()["state"]) == len(opt.state_dict()["state"]) > 0
    # by name / path: the basename is looked up in output_dir (train.py:1193)
    assert C.resume_from_checkpoint(nets2, out, "/somewhere/else/checkpoint-10000", None) == 10000
    assert float(nets2[0].conv_in.bias[0]) == 10000.0
    assert C.resume_from_checkpoint(nets2, out, "checkpoint-5000", None) == 0  # rotated away -> fresh run


def test_rotation_without_limit_keeps_everything(tmp_path):
    from uni_renderer_amd import checkpointing as C

    out = str(tmp_path / "run")
    nets = _nets(52)
    for step in (1, 2, 3, 10):
        C.save_state(nets[:1], out, step)
    assert C.list_checkpoints(out) == ["checkpoint-1", "checkpoint-2", "checkpoint-3", "checkpoint-10"]  # numeric order
    assert C.rotate_checkpoints(out, 2) == ["checkpoint-1", "checkpoint-2", "checkpoint-3"]
