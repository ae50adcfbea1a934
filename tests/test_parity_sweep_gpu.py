"""The parity claim with a WORST CASE instead of one seed: single enc -> unet -> dec steps at SD size
(64x64 latent, batch 1, fp16) over 8 input seeds x timesteps {0, 500, 999, drawn}, through both executors (the grouped
step bench.py times, and the hoisted step the sampling loops replay), against COMMITTED outputs of the CPU fp32 oracle
(tests/golden/sd_cfg3_sweep.safetensors, generator tests/golden/make_golden_sd.py --sweep, run in the build container).

img_pred is compared in full; attr_pred on the fixed random quarter of its elements the golden stores (``attr_index``; the
sampling error of a rel-L2 over 28 672 elements is ~0.5 % of its value).  Asserted on the MAXIMUM over the sweep:
  * vs the oracle holding the same fp16-rounded parameters: north_star's 1e-3;
  * vs the fp32-parameter oracle (adds the checkpoint's fp16 quantisation, which no kernel can undo): 1.3e-3.
Like every golden here these pin the oracle, not the reference (parity unpinned)."""
import json
import os
import sys

import pytest
import torch

from conftest import rel_l2
from util_models import O, build_product_from_oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "sd_cfg3_sweep.safetensors")
BOUND_SAME, BOUND_FP32W = 1e-3, 1.3e-3


@pytest.fixture(scope="module")
def gold():
    from safetensors.torch import load_file

    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/sd_cfg3_sweep.safetensors not generated (tests/golden/make_golden_sd.py --sweep)")
    return load_file(GOLD)


@pytest.fixture(scope="module")
def nets(dev, gold):
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden_sd import weights_probe

    oracle = O.build_triplet(O.SD15_CONFIG, seed=1234)
    assert torch.allclose(weights_probe(oracle), gold["weights_probe"], rtol=1e-10, atol=0), "seeded weights differ from the golden's (RNG drift)"
    prod = build_product_from_oracle(*oracle, torch.float16, dev)
    del oracle
    return prod


def sweep_errors(dev, gold, nets):
    """{executor: {case: {img/attr x fp16w/fp32w}}} over the whole sweep; shared with bench-side tools."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden_sd import SWEEP_SEEDS, SWEEP_T, sweep_inputs

    from uni_renderer_amd.graph import GraphedDualStreamStep, GraphedHoistedStep

    unet, enc, dec = nets
    kw = dict(batch=1, latent_hw=64, cross_dim=768, dtype=torch.float16, device=dev)
    grouped = GraphedDualStreamStep(unet, enc, dec, **kw)
    hoisted = GraphedHoistedStep(unet, enc, dec, **kw)
    idx = gold["attr_index"].to(dev)
    res = {"grouped": {}, "hoisted": {}}
    for seed in SWEEP_SEEDS:
        for tname in SWEEP_T:
            x, c, ehs, ti, ta = [t.to(dev) for t in sweep_inputs(seed, tname)]
            key = f"s{seed}.t{tname}"
            out = grouped.step(x.half(), c.half(), ehs.half(), ti, ta)
            img, attr = out["img_pred"].float().clone(), out["attr_pred"].float().flatten()[idx].clone()
            hst = hoisted.step(x.half(), c.half(), ehs.half(), ti, ta)["attr_pred"].float().flatten()[idx].clone()
            e = {}
            for tag in ("fp16w", "fp32w"):
                gi, ga = gold[f"{key}.img_pred.{tag}"], gold[f"{key}.attr_sample.{tag}"]
                e[tag] = dict(img=rel_l2(img, gi), attr=rel_l2(attr, ga), hoisted_attr=rel_l2(hst, ga))
            res["grouped"][key] = {t: dict(img=e[t]["img"], attr=e[t]["attr"]) for t in e}
            res["hoisted"][key] = {t: dict(attr=e[t]["hoisted_attr"]) for t in e}
    return res


def summarise(res):
    out = {}
    for ex, cases in res.items():
        for tag in ("fp16w", "fp32w"):
            for what in ("img", "attr"):
                vals = {k: v[tag][what] for k, v in cases.items() if what in v[tag]}
                if vals:
                    worst = max(vals, key=vals.get)
                    out[f"{ex}.{what}.{tag}"] = dict(max=vals[worst], at=worst, mean=sum(vals.values()) / len(vals), min=min(vals.values()))
    return out


def test_worst_case_over_seeds_and_edge_timesteps_at_sd_size(dev, gold, nets):
    res = sweep_errors(dev, gold, nets)
    summ = summarise(res)
    print(json.dumps(dict(sweep="8 seeds x t in {0, 500, 999, drawn}, SD size, 64x64 latent, batch 1, fp16", summary=summ)))
    by_t = {}
    for ex, cases in res.items():
        for k, v in cases.items():
            t = k.split(".t")[1]
            for what, val in v["fp16w"].items():
                by_t.setdefault(f"{ex}.{what}.t{t}", []).append(val)
    print(json.dumps({k: round(max(v), 7) for k, v in sorted(by_t.items())}))
    for k, v in summ.items():
        bound = BOUND_SAME if k.endswith("fp16w") else BOUND_FP32W
        assert v["max"] < bound, (k, v)
