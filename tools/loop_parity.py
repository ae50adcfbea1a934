#!/usr/bin/env python3
"""cfg 3 as the reference runs it -- the FULL 50-step DDIM inverse-rendering loop (models/pipeline.py:2629-2730:
enc -> unet -> dec per step, the 24 attribute latent channels fed back through the scheduler, mask channels and image
latents fixed, t_img = 0) -- at SD-1.x size on the HIP path (captured default executor, fp16) against the SAME loop run
by the CPU fp32 oracle networks and the independent numpy-float64 DDIM restatement (oracle/schedulers_oracle.py).
Runs all 50 steps on both sides (batch 1: ~4-5 min of host time) and records how the per-step error compounds -- the
per-step TRAJECTORY; since round 5 the END of the loop at the benchmarked batch 4 is a 15-second test against committed oracle
outputs (tests/test_golden_sd_gpu.py, tests/golden/sd_cfg3_b4.safetensors: final latents 0.50e-3 / 0.90e-3).

    python tools/loop_parity.py [--batch 1] [--steps 50] [--out profiles/r04_loop_parity.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_loop_parity.json"))
    a = ap.parse_args()
    from util_models import O, OracleScheduler, build_product_from_oracle  # the oracle is the checker here, never the product

    from uni_renderer_amd.graph import GraphedDualStreamStep
    from uni_renderer_amd.schedulers import DDIMScheduler

    dev = torch.device("cuda:0")
    B, L = a.batch, a.latent
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    oracle = O.build_triplet(O.SD15_CONFIG, seed=1234)
    unet, enc, dec = build_product_from_oracle(*oracle, torch.float16, dev)
    x, c, ehs, ti, ta = O.make_inputs(B, L, 768, seed=28, t_img=0)
    so = OracleScheduler("ddim", a.steps)
    sp = DDIMScheduler()
    sp.set_timesteps(a.steps)
    assert so.timesteps.tolist() == sp.timesteps.tolist()
    runner = GraphedDualStreamStep(unet, enc, dec, batch=B, latent_hw=L, cross_dim=768, dtype=torch.float16, device=dev)
    lat_o = c.clone()
    lat = c.to(dev).float()
    xg, eg, tig = x.to(dev).half(), ehs.to(dev).half(), ti.to(dev)
    traj, t_cpu, t_gpu = [], 0.0, 0.0
    for k, t in enumerate(so.timesteps):
        t0 = time.perf_counter()
        out_o = O.dual_stream_step(*oracle, x, lat_o, ehs, ti, t.expand(B))
        lat_o = torch.cat([lat_o[:, :4], so.step(out_o["attr_pred"][:, 4:], t, lat_o[:, 4:])[0]], 1)
        t_cpu += time.perf_counter() - t0
        t0 = time.perf_counter()
        out = runner.step(xg, lat.half(), eg, tig, t.expand(B).to(dev))
        lat = torch.cat([lat[:, :4], sp.step(out["attr_pred"][:, 4:].float(), t, lat[:, 4:])[0]], 1)
        torch.cuda.synchronize()
        t_gpu += time.perf_counter() - t0
        row = dict(step=k + 1, t=int(t), rel_l2_attr_pred=rel(out["attr_pred"], out_o["attr_pred"]),
                   rel_l2_img_pred=rel(out["img_pred"], out_o["img_pred"]), rel_l2_latents=rel(lat[:, 4:], lat_o[:, 4:]))
        traj.append(row)
        if (k + 1) % 5 == 0 or k == 0:
            print(json.dumps(row), flush=True)
    res = dict(what="cfg 3: full DDIM loop, inverse direction, SD-1.x-size networks (1.74 G parameters, random init), fp16, "
                    "captured default executor vs the CPU fp32 oracle loop (fp32 weights) + numpy-float64 DDIM oracle",
               batch=B, latent=L, steps=a.steps,
               final_rel_l2_latents=traj[-1]["rel_l2_latents"],
               final_rel_l2_of_the_update=rel(lat[:, 4:].cpu() - c[:, 4:], lat_o[:, 4:] - c[:, 4:]),
               max_rel_l2_attr_pred_over_steps=max(r["rel_l2_attr_pred"] for r in traj),
               max_rel_l2_img_pred_over_steps=max(r["rel_l2_img_pred"] for r in traj),
               oracle_seconds=round(t_cpu, 1), hip_path_seconds_incl_host_scheduler=round(t_gpu, 2), trajectory=traj)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "trajectory"}))


if __name__ == "__main__":
    main()
