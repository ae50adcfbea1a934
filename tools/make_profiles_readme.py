#!/usr/bin/env python3
"""Regenerate profiles/README.md from the committed summaries (profiles/<tag>_*.{json,csv}, profiles/pmc_traffic.json).

    python tools/make_profiles_readme.py [tag=r01c]
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def pretty(n):
    m = re.search(r"igemm_kernelID(?:F16_|F16b)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(n?\d+)ELb(\d)(?:ELi(\d+)ELi(\d+))?", n)
    if m:
        bm, bn, wm, wn = (int(m.group(i)) for i in range(1, 5))
        extra = ""
        if m.group(7):
            extra += ", 32x32x16 MFMA" if m.group(7) == "32" else ""
            extra += f", {m.group(8)} loader waves" if m.group(8) != "0" else ""
        return f"igemm_kernel<{bm}x{bn}, {wm * wn} waves, stages {m.group(5)}, {'conv3x3' if m.group(6) == '1' else 'gemm'}{extra}>"
    m = re.search(r"attention32_kernelID(?:F16_|F16b)Li(\d+)ELb(\d)", n)
    if m:
        return f"attention32_kernel<d={m.group(1)}{', reference slot' if m.group(2) == '1' else ''}>"
    m = re.search(r"attention_kernelID(?:F16_|F16b)Li(\d+)", n)
    if m:
        return f"attention_kernel<d={m.group(1)}>"
    m = re.search(r"_ZN2ur\d+([a-z0-9_]+?)(?:ID|E|I)", n)
    return m.group(1) if m else n[:60]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01c"
    rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_kernel_stats.csv"))))
    d = json.loads(open(os.path.join(P, f"{tag}_bench_default.json")).read().strip().split("\n")[-1])
    mf = json.load(open(os.path.join(P, f"{tag}_pmc_mfma_util.json")))
    tr = json.load(open(os.path.join(P, "pmc_traffic.json")))
    l2p = os.path.join(P, f"{tag}_pmc_l2_step.json")
    l2 = json.load(open(l2p)) if os.path.exists(l2p) else {}
    ours = [(pretty(r["Name"]), r) for r in rows if "_ZN2ur" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for _, r in ours)
    alltot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = []
    for name, r in ours:
        u = mf.get(r["Name"])
        lines.append(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
                     f"{100 * float(r['TotalDurationNs']) / tot:.1f} | {('%.1f %%' % (100 * u['mfma_util'])) if u else ''} |")
    roof, cpu = d["roofline"], d["cpu_baseline"]
    dom_sym = roof["kernel"].replace("igemm_", "").replace("s2_", " ").split()[0]  # e.g. 128x320
    dom = [r for n, r in ours if dom_sym in n and ("conv3x3" in n) == ("conv3x3" in roof["kernel"])][0]
    att = [k for k in d["kernel_classes"] if k["kernel"] == "attention_d40"]
    att_u = [v for k, v in mf.items() if "attention32" in k and "Li40" in k and "F16_" in k]
    loop_txt = ""
    lp, lb, hb = d.get("loop"), os.path.join(P, f"{tag}_loop_bench.json"), os.path.join(P, f"{tag}_hoist_bench.json")
    if lp and os.path.exists(lb) and os.path.exists(hb):
        L = json.loads(open(lb).read().strip().split("\n")[-1])
        H = json.loads(open(hb).read().strip().split("\n")[-1])
        par = d["config"].get("parity_rel_l2", {})
        loop_txt = f"""## Sampling loops (round 5: the loop-invariant half hoisted, `uni_renderer_amd/hoist.py`)

`loop` object of the bench line (50-step DDIM calls of the pipeline, latents in / out, on-device sampler, best of 3): inverse
{lp['inverse_ms_total']:.0f} ms ({lp['inverse_ms_per_step']:.2f} ms per step), rendering {lp['render_ms_total']:.0f} ms ({lp['render_ms_per_step']:.2f}).  `{tag}_loop_bench.json` (its own
process): hoisted {L['inverse_50']['ms_total']:.0f} / {L['render_50']['ms_total']:.0f} ms against {L['inverse_50_all_networks_every_step']['ms_total']:.0f} / {L['render_50_all_networks_every_step']['ms_total']:.0f} ms with every network on
every step; the eval protocol (UniPC 20 steps x 5 repeats) {L['unipc20_five_calls_b1']['ms_total']:.0f} ms as five calls, {L['unipc20_folded_b5']['ms_total']:.0f} ms folded into one batch.
`{tag}_hoist_bench.json`: prologue graph {H['inverse']['prologue_ms']:.2f} ms once per call, per-step graph {H['inverse']['step_ms']:.2f} ms (inverse: encoder + 13
adds + decoder, {H['inverse'].get('launches', 302)} launches) / {H['render']['step_ms']:.2f} ms (rendering: UNet + 13 adds) against {H['inverse']['all_networks_step_ms']:.2f} / {H['render']['all_networks_step_ms']:.2f} ms for the
grouped all-networks step; `{tag}_hoist_kernel_stats.csv` = the same tool under `rocprofv3 --kernel-trace --stats`.
Parity measured in the bench run on the benchmarked networks and inputs (`config.parity_rel_l2`, img_pred / attr_pred): same
fp16-rounded parameters {par.get('same_weights', {}).get('img_pred', 0):.2e} / {par.get('same_weights', {}).get('attr_pred', 0):.2e}, fp32 parameters {par.get('fp32_weights', {}).get('img_pred', 0):.2e} / {par.get('fp32_weights', {}).get('attr_pred', 0):.2e}.
"""
    out = f"""# profiles/ — rocprofv3 evidence

All files were produced on an MI355X `gpurun` box from this repo (`bash tools/collect_profiles.sh {tag}`, which runs the
commands below and keeps the summaries; this file is `python tools/make_profiles_readme.py {tag}`).  `r01_*` / `r01c_*` =
round 1 (first complete collection / end of round), kept for the history of the numbers; `{tag}_*` = the state at the
end of round {tag[1:3].lstrip("0")}.

Round-3 additions: `r03_tchain_bench.json` (`tools/tchain_bench.py`: the three row-local chain launches of the
320-channel transformer blocks against the GEMM / LayerNorm launches they replace, graph-timed, plus s_memtime phase and
per-stage stamps of the kernels), `r03_pmc_conv_sq.json` / `r03_pmc_gemm320_sq.json` (`tools/pmc_conv_sq.sh`: three SQ-counter
passes over the level-0 conv / a K = 320 GEMM replayed alone: parked, issue-stalled and active wave cycles, LDS activity,
bank conflicts), `r03_mfma_rate.txt` (`tools/ubench/mfma_rate.hip`, the corrected MFMA issue-rate micro-benchmark: random
operands, distinct A / B registers, 1 / 2 / 4 waves per SIMD), `r03_ring_depth_ab.txt` (`tools/experiments/r03_run6.sh`: a 3-deep LDS ring
on the 128x256 tile against the 2-deep one: no difference, the conv loop is not waiting for its copies).
Second half of round 3: `r03_determinism.txt` (`tools/determinism_check.py`, `tools/tchain_determinism.py`: the grouped step
and the three chain kernels repeated on fixed inputs, bitwise comparison -- after the LDS race of the first chain kernel was
fixed, DESIGN.md section 5), `r03_gap_analysis.txt` (`tools/gap_analysis.py` over a `rocprofv3 --kernel-trace` of the graph
replay: 393 kernels per step, 0.1 % idle between them), `r03_wsconv_bench.txt` (`tools/wsconv_bench.py`: the weight-streaming
conv kernel against the tuned LDS-tiled one on the 18 resnet conv shapes of the step), `r03_pmc_attn_slot.txt`
(`tools/pmc_attn.sh`: SQ counters of the d = 40 self-attention on the pre-scaled path), `r03_ab_collections.txt` (same-box
A/B of the commits of the round's two collections: the 11.67 vs 12.01 ms of their bench lines is the box, not the code).
Round-4 additions (each is the output of one `tools/experiments/r04_run*.sh` call): `r04_pp_ab.txt` (`tools/pp_ab.py`: the
8-wave ping-pong tiles of `csrc/igemm_pp.hip` against the shipped table on the 41 heaviest problems of the step, isolated
graph replay, with the rel-L2 between the two outputs), `r04_pp_ablate.txt` (ablation builds of the ping-pong kernel on the
level-0 conv and three more problems: no copies / no MFMAs / no fragment reads / pairs / no barriers; first block =
unablated tiles 9, 49, 50, 55), `r04_pp_variants.txt` (four schedule variants: `s_setprio` on / off x copies issued in the
READ block or between the MFMAs), `r04_pp_cblock.txt` (K order of the conv: channel blocks of 64 / 320, lock-step, loader-wave
and ping-pong tiles, and the copies-only build), `r04_insitu_pp.txt` (in-situ tuning pass with the ping-pong tiles as
candidates: 244 captured steps), `r04_ldsdma_pattern.txt` (`tools/ubench/ldsdma_pattern.hip`: global -> LDS delivery rate of
a CU by address pattern: contiguous / 128-byte / 64-byte row segments, shared / private / mixed regions, buffer vs flat
addressing, ring depth 2 / 3 / 5), `r04_qkv_ab.txt` and `r04_splitk_gn_ab.txt` (same-box alternations of `UR_QKV_ONE_LAUNCH`,
`UR_SPLITK_GN`, `UR_CTXKV_ONE_LAUNCH`), `r04_loop_parity.json` (`tools/loop_parity.py`: the full 50-step DDIM loop of cfg 3 at SD
size against the oracle loop, per-step trajectory), `r04_weight_sensitivity.json` (`tools/weight_sensitivity.py`: fp16 rounding
of the checkpoint by parameter subset kept in fp32, oracle arithmetic), `r04_smoke.txt` (`__graft_entry__.smoke()`),
`r04_gap_analysis.txt`, `r04_determinism.txt`.  The cfg 2 / cfg 5 lines of round 4 carry `roofline` with live traffic.
Round 4, second half (the weight-gradient kernel `csrc/wgrad.hip`): `r04_tr_probe.txt` (`tools/ubench/tr_probe.hip`: what
`ds_read_b64_tr_b16` returns lane by lane), `r04_wgrad_bench.txt` (`tools/wgrad_bench.py --sweep`: `ur_wgrad` per tile / ring
depth / slice count against transposes + im2col + GEMM + column sums on 29 problems of the training step),
`r04_wgrad_ablate.txt` (`tools/wgrad_ablate.py` on a `make WGRAD_ABL=1` library: no copies / no MFMAs / no fragment reads /
plain 8-byte reads, with the LDS conflict counters of the full kernel), `r04_wgrad_l2.txt` (TCC hit / miss of the kernel),
`r04_wgrad_ring.txt` (ring depth 2 / 4 / 8 at equal tiles: workgroups per CU, not bytes in flight, is the lever),
`r04_tune_wgrad.txt` / `r04_tune_wgrad_groups.txt` (`tools/tune_wgrad.py`: the (tile, slices) table, single problems and the
grouped launches of the deferred queue), `r04_wgrad_step_ab.txt` (graphed training step alternating `UR_WGRAD_DEFER=1`,
`UR_WGRAD_DEFER=0`, `UR_WGRAD=0` on one box), `r04_insitu_final.txt` / `r04_insitu_train.txt` (in-situ tile tuning of the inference
step and of the captured training step on the final tree), `r04_k32_ab.txt` (`tools/pp_ab.py --tiles 56..61`: the 32-deep-chunk
tiles of `ur_igemm`, four workgroups per CU, against the table: slower on 39 of 40 problems; the tiles are not in the tree,
`tools/experiments/r04_k32_tiles.patch`).
Round-5 additions: `r05_hoist_shapes.json` / `r05_hoist_bench.json` / `r05_hoist_kernel_stats.csv` (`tools/hoist_bench.py`: prologue and
per-step graph of the hoisted sampling loops, per-shape event tables, the step graph's kernels under `rocprofv3 --kernel-trace
--stats`), `r05_hoist_ab.txt` (chain kernels on / off for the z = 1 launches; the cfg 2 shape), `r05_insitu_loop.txt`
(`tools/tune_in_situ.py --loop inverse`: two in-situ tuning passes over the hoisted step), `r05_loop_bench.json` (50-step loops
hoisted and with every network on every step), `r05_train_graph_rccl_w1_captured.json` / `..._serial.json`
(`tools/train_bench.py --graph --force-collectives`: the bucketed training step on RCCL world size 1 with the collectives
captured into the graph vs forward + backward graph | eager collectives | update graph, with per-phase times),
`docs/NOTEBOOK_r1_r4.md` (the measurement narratives of rounds 1-4, moved out of DESIGN.md verbatim).  The bench line of round 5
carries `loop` and `config.parity_rel_l2`.
Round-6 additions: `r06_pmc_attn_l2_cfg5.json` / `r06_pmc_attn_l2_cfg5_head_major.json` / `r06_pmc_attn_l2_cfg3.json` (`tools/pmc_attn_l2.sh`: L2 hit rate,
memory-side read requests by size, FETCH_SIZE / WRITE_SIZE of the d = 40 attention in separate `--pmc` passes -- token-matrix q / k vs
the head-major images of round 6: 247.7 -> 90.7 MB per 16384-token launch for 83.9 MB algorithmic), `r06_head_major_ab.txt`
(`tools/r06_head_major_ab.sh`: the step with / without the images alternating on one box, the isolated launches),
`r06_xcd_raster_ab.txt` (per-launch tile order inside an XCD's run: feature and library A/B, rejected), `r06_tchain_io_ab.txt` (chain
kernels: loads in flight, store drain, phase offset -- no effect), `r06_insitu_loop_render.txt` (in-situ pass on the rendering loop),
`r06_gridbar.txt` (`tools/ubench/gridbar.hip`: a device-wide barrier inside one persistent kernel -- fences +
counter / counter only with `sc1` data / fences by one workgroup per XCD / hierarchical -- against one kernel launch per phase in
a graph, 128 / 256 / 512 workgroups, with a cross-XCD data check), `r06_bigwave_sweep.txt` (`tools/ab_gemm.py --sweep --tiles
9,22,42,43,44,11,28,56..59`: the few-wave / big-wave-tile igemm builds of round 6 on the seven heaviest problems, isolated),
`r06_insitu_bigwave.txt` / `r06_insitu_cfg2.txt` (`tools/tune_in_situ.py --broad`: the new tiles on the headline step; cfg 2 after
the chain gate became row-aware), `r06_parity_sweep.json` (`tests/test_parity_sweep_gpu.py -s`: max / mean / min rel-L2 over 8 seeds
x 4 timestep settings, both executors, both oracles), `r06_loop_bench.json` (incl. the loops WITHOUT the per-call time tables),
`r06_train_graph_rccl_w1_captured.json` / `..._r05_protocol.json` / `..._serial.json` / `r06_train_graph.json` and
`r06_train_rccl_kernel_stats_r05_protocol.csv` (`tools/r06_train_ab.sh`, `tools/r06_train_prof.sh`: the bucketed training step with
gradients written into the buckets vs added into zeroed buckets, one box; the kernel statistics of the old protocol show the 1430
`add` launches per step), `r06_splitk_gn_ab.txt` (GroupNorm as the split-K second pass over group-blocked slabs, in-step
alternation), `r06_tchain_rows_ab.txt` (chain kernels forced on / off at 4096 / 8192 / 16384 rows per launch),
`r06_gn_resident_ab.txt` (`tools/gn_bench.py` + in-step alternation: register-resident one-launch GroupNorm),
`r06_runtime_env_ab.txt` (HIP runtime knobs against the captured step).  The bench line of round 6 carries `loop.roofline` and
`loop.parity_rel_l2`.
Other summaries: `{tag}_parity_numbers.json` (every rel-L2 the `-m gpu` suite printed: the chain kernels, cfg 3 at batch 2 and 4
and as a 5-step DDIM loop, cfg 5 vs the oracle, the 16384-token attention, UpRes, the module-surface and cfg-4 training steps
incl. the inverse branch at SD size, the VAE, the RCCL world-size-1 collectives),
`{tag}_bench_cfg2/cfg5/b5/b8/b10/b20.json` (the other configurations and the repeat batches through `bench.py`),
`{tag}_loop_bench.json` (50-step DDIM loops; 20-step UniPC x 5 repeats as five calls vs one folded batch),
`{tag}_train_graph/eager.json` (`tools/train_bench.py`; `{tag}_train_graph_torch_adamw.json`: the same with torch's fused AdamW instead
of `optim.FusedAdamW`), `{tag}_vae_bench.json` (`tools/vae_bench.py`); round 2's A/B logs stay as `r02_*`.

| file | what |
|---|---|
| `{tag}_bench_default.json` | `python bench.py` (defaults: 30 steps, 5 warm-up, fp16, B=4, 64x64 latent), un-profiled: the headline line incl. `roofline` and `cpu_baseline` |
| `{tag}_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline` (16 step executions in the process: 2 capture warm-ups, 2+10 graph replays, 2 eager roofline steps) |
| `{tag}_bench_under_rocprofv3.json` | the bench line printed by that profiled run |
| `{tag}_per_shape_eager_events.json` | per (kernel class, problem shape) table of one eager step, each launch bracketed by HIP events on the launch stream (`bench.py --shape-table`); eager launches of small kernels include launch latency |
| `pmc_traffic.json` | HBM-side bytes per launch per kernel class from two separate PMC passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace`, `--pmc WRITE_SIZE --kernel-trace`, each around `bench.py --steps 3 --warmup 1`), reduced by `tools/pmc_traffic.py`: `FETCH_SIZE*2*1024 + WRITE_SIZE*1024` (KiB units; gfx950 FETCH_SIZE counts 128-B requests as 64 B — MI355X_MICROARCH.md §HBM).  `bench.py` reads it for `roofline.traffic`. |
| `{tag}_pmc_l2_step.json` | a fourth PMC pass (`--pmc TCC_HIT TCC_MISS TCC_REQ --kernel-trace`, `tools/pmc_l2_step.sh`): L2 hit rate and miss bytes per launch per kernel class, in situ over whole steps |
| `{tag}_pmc_mfma_util.json` | a third PMC pass (`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace`): per kernel symbol `SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)` summed over its launches (busy = 16 cycles per 16x16x32 MFMA, 32 per 32x32x16; calibrated with `tools/ubench/mfma_rate.hip`) |

Commands (as the guide prescribes, counters in their own passes): `cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT &&
rocprofv3 --kernel-trace --stats -d <out> -o {tag} --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline`.

## Headline (un-profiled, `{tag}_bench_default.json`)

{d['value']:.2f} denoise-steps/s = {d['ms_per_step']:.2f} ms per dual-stream step (enc + unet + dec, SD-1.x size, B=4, 512x512, fp16,
default (hi, lo) residual stream) on one MI355X (boxes of the pool differ by +-2 %: round 5's tree measured 11.44 .. 11.88 ms over
the day's boxes -- 87.0 / 86.4 / 85.0 / 84.2 steps/s in four bench runs, the tuner's baselines 11.44 and 11.79 ms; only same-box
A/Bs compare code); CPU oracle on the same
host ({cpu['cores']}-core cgroup quota) {cpu['value']:.4f} steps/s (median of 3 timed steps).  6.49 TFLOP/step => {6.49 / d['ms_per_step']:.3f} PFLOP/s
algorithmic = {100 * 6.49 / d['ms_per_step'] / 2.5:.0f} % of the dense fp16 MFMA roofline for the whole step (round 1 ended at 12.3 .. 13.1 ms, round 2 at 12.0 .. 12.9 ms;
DESIGN.md section 4, "Round 3").

Dominant kernel (by symbol; plain + split-K launches {100 * roof.get('share_of_step_incl_splitk_launches', roof['share_of_step']):.0f} % of the step):
`{roof['kernel']}` at {roof['achieved']:.0f} TFLOP/s = {100 * roof['frac']:.0f} % of peak over its {roof['calls_per_step']} plain launches/step
({roof['avg_launch_us']:.0f} us each by HIP events; rocprofv3 average over plain AND split launches of the symbol: {float(dom['AverageNs']) / 1e3:.1f} us),
MFMA-busy {100 * mf[dom['Name']]['mfma_util']:.0f} %.  PMC traffic of the plain launches {tr[roof['kernel']]['hbm_bytes_per_launch'] / 1e6:.0f} MB/launch
({tr[roof['kernel']]['read_bytes'] / 1e6:.0f} read + {tr[roof['kernel']]['write_bytes'] / 1e6:.0f} written; split-K launches of the symbol {tr.get(roof['kernel'] + '_splitk', {}).get('hbm_bytes_per_launch', 0) / 1e6:.0f} MB incl. their fp32 partial
slabs) against {roof['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic: FETCH_SIZE / WRITE_SIZE count what leaves the XCD's L2, not what reaches
HBM.  In situ the L2 hit rate of these launches is {100 * l2.get(roof['kernel'], {}).get('hit_rate', 0):.0f} % of {l2.get(roof['kernel'], {}).get('tcc_req', 0) * 128 / 1e6:.0f} MB of requests
(`{tag}_pmc_l2_step.json`); replayed alone the level-0 problem has 92 % hits and 57 MB of misses (`tools/pmc_l2.sh`).  With the K axis tap-outer these launches had 83 % hits and 205 MB of
misses: a workgroup returns to an input line after C/64 chunks of every co-resident workgroup, 9 MB per XCD at C = 320
but 18-27 MB for the 640 / 960-channel inputs of the up path, where all nine taps missed.  The channel-block-outer K
order (`ur_igemm_desc.cblock = 320`: K walks (block of 320 channels, tap), weights packed to match) makes every conv
re-read like the C = 320 one; what remains above the operand bytes is every XCD of a z fetching that z's whole weight
matrix.  The step time did not change (+-0.3 %): the misses were served by the 256 MB Infinity Cache under an
MFMA-issue-bound loop (DESIGN.md section 4).
Attention d = 40 (4096-token self-attention + 77-key cross-attention launches): {att[0]['tflops'] if att else 0:.0f} TFLOP/s,
MFMA-busy {100 * att_u[0]['mfma_util'] if att_u else 0:.0f} % (north_star asks >= 40 %), PMC traffic {tr.get('attention_d40', {}).get('hbm_bytes_per_launch', 0) / 1e6:.0f} MB/launch vs 84 MB algorithmic Q+K+V+O.

{loop_txt}
## Kernel summary (`{tag}_kernel_stats.csv`, our kernels only, 16 step executions, grouped mode)

| kernel | calls | total ms | avg us | % | MFMA-busy (PMC pass) |
|---|---|---|---|---|---|
""" + "\n".join(lines) + f"""

Sum over all `ur::` kernels: {tot / 1e6:.1f} ms in the profiled process = {tot / 1e6 / 16:.2f} ms per step execution; all kernels incl.
the harness's torch fills/copies: {alltot / 1e6:.1f} ms.
"""
    tk = os.path.join(P, f"{tag}_train_kernel_stats.csv")
    if os.path.exists(tk):
        trows = list(csv.DictReader(open(tk)))
        n = 6  # step executions in that process: 2 warm-up + 1 + 3 graph replays
        ttot = sum(float(r["TotalDurationNs"]) for r in trows)

        def short(nm):
            return pretty(nm) if "_ZN2ur" in nm or nm.startswith("ur::") else re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", nm)[:70]
        tg = os.path.join(P, f"{tag}_train_graph.json")
        unprof = json.loads(open(tg).read().strip().split("\n")[-1])["ms_per_step"] if os.path.exists(tg) else 144
        tl = "\n".join(f"| {short(r['Name'])} | {int(r['Calls']) // n} | {float(r['TotalDurationNs']) / n / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} |"
                       for r in trows[:16])
        out += f"""
## Training step (`{tag}_train_kernel_stats.csv`: `rocprofv3 --kernel-trace --stats -- python tools/train_bench.py --steps 3 --graph`)

cfg 4's per-GPU shape (B=4, 512x512, bf16 compute, fp32 master parameters, fused AdamW, clipping), the whole step replayed as
one HIP graph: {ttot / n / 1e6:.0f} ms of kernel time per step under the profiler ({unprof:.0f} ms per replay un-profiled).  The profiled
process also builds and initialises the three networks: the `FillFunctor<float>` / `distribution_elementwise` rows are that, not
the step.  Top kernels per step:

| kernel | launches / step | ms / step | avg us |
|---|---|---|---|
{tl}
"""
    open(os.path.join(P, "README.md"), "w").write(out)
    print(out[-1500:])


if __name__ == "__main__":
    main()
