#!/usr/bin/env python3
"""End-to-end sampling loops (latents in, latents out; VAE / CLIP outside): wall time of the 50-step
inverse-rendering loop (UniRendererPipeline.real_image2mask_3mod_albedo) and the rendering loop
(mask2image_3mod_albedo) against 50 x the bare graph-replayed denoise step."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uni_renderer_amd.pipeline import UniRendererPipeline  # noqa: E402


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    B, L, steps = int(os.environ.get("B", 4)), 64, 50
    unet, enc, dec = bench.build_models(dev, dt)
    pipe = UniRendererPipeline(unet=unet, controlnet=enc, controldec=dec)
    g = torch.Generator(device=dev).manual_seed(3)
    img = torch.randn(B, 4, L, L, device=dev, generator=g).to(dt)
    msk = torch.randn(B, 4, L, L, device=dev, generator=g).to(dt)
    ehs = (torch.randn(B, 77, 768, device=dev, generator=g) * 0.5).to(dt)
    attr = torch.randn(B, 28, L, L, device=dev, generator=g).to(dt)
    res = {}
    for hoist in (True, False):  # False: every network on every step (the round-4 loops)
        pipe.hoist_invariants = hoist
        sfx = "" if hoist else "_all_networks_every_step"
        for name, fn in (
            ("inverse_50", lambda: pipe.real_image2mask_3mod_albedo(prompt_embeds=ehs, image_latents=img, mask_latents=msk,
                                                                   num_inference_steps=steps, guidance_scale=0.0,
                                                                   output_type="latent")),
            ("render_50", lambda: pipe.mask2image_3mod_albedo(prompt_embeds=ehs, attr_latents=attr,
                                                              num_inference_steps=steps, guidance_scale=0.0,
                                                              output_type="latent")),
        ):
            fn()  # capture + warm
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res[name + sfx] = dict(ms_total=round(min(ts) * 1e3, 2), ms_per_step=round(min(ts) * 1e3 / steps, 3))
    pipe.hoist_invariants = True
    # round 6: the same hoisted loops WITHOUT the per-call time-projection tables (every step recomputes its time embedding)
    pipe.precompute_time_tables = False
    for name, fn in (
        ("inverse_50_no_time_tables", lambda: pipe.real_image2mask_3mod_albedo(prompt_embeds=ehs, image_latents=img, mask_latents=msk,
                                                                              num_inference_steps=steps, guidance_scale=0.0,
                                                                              output_type="latent")),
        ("render_50_no_time_tables", lambda: pipe.mask2image_3mod_albedo(prompt_embeds=ehs, attr_latents=attr,
                                                                         num_inference_steps=steps, guidance_scale=0.0,
                                                                         output_type="latent")),
    ):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[name] = dict(ms_total=round(min(ts) * 1e3, 2), ms_per_step=round(min(ts) * 1e3 / steps, 3))
    pipe.precompute_time_tables = True
    # the reference's live eval protocol (eval/test_real.py:485-492, 547-564): UniPC, 20 steps, guidance 0, the same image
    # 5 times (compute_times) -- as five calls at batch 1 and folded into ONE batch of 5 (num_images_per_prompt=5)
    from uni_renderer_amd.pipeline import SCHEDULER_NAMES
    from uni_renderer_amd.schedulers import UniPCMultistepScheduler

    for n in SCHEDULER_NAMES:
        setattr(pipe, f"scheduler_{n}", UniPCMultistepScheduler())
    kw = dict(prompt_embeds=ehs[:1], image_latents=img[:1], mask_latents=msk[:1], num_inference_steps=20, guidance_scale=0.0,
              output_type="latent")
    for name, fn in (("unipc20_five_calls_b1", lambda: [pipe.real_image2mask_3mod_albedo(**kw) for _ in range(5)]),
                     ("unipc20_folded_b5", lambda: pipe.real_image2mask_3mod_albedo(num_images_per_prompt=5, **kw))):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[name] = dict(ms_total=round(min(ts) * 1e3, 2))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
