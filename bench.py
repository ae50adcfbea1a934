#!/usr/bin/env python3
"""Headline benchmark: denoise-steps/sec of the dual-stream step (enc + unet + dec, SD-1.x size, 512x512 image
=> 64x64 latent, batch 4 per GPU, fp16) on N MI355X -- BASELINE.json's metric on config[2] ("inverse-rendering
direction, 512x512, bs=4, fp16"), the configuration the metric is quoted on.

    python bench.py                                  # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W     # N GPUs, one rank per GPU (RCCL only for the timing barrier)

A "step" is one full pass enc -> unet -> dec over one batch of synthetic latents already resident in HBM,
replayed as one HIP graph.  Batch-sharded data parallel inference has no data-path collective:
every rank denoises its own batch of 4, scaling is "weak", value = N * K / max-over-ranks(time).

Rank 0 additionally reports
  roofline     -- the dominant kernel class of the step, measured live with HIP events around every launch of one
                  eager step (events recorded on the launch stream): algorithmic FLOP (or bytes) / summed duration
  cpu_baseline -- the CPU fp32 oracle (oracle/unirenderer_oracle.py, "port") timed on this host's cores on one
                  step of the same shape.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_TFLOPS = 2500.0  # dense fp16/bf16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(L, B=1, direction="inverse", cross_tokens=77, cross_dim=768):
    """Algorithmic FLOPs (2 x MAC) of one dual-stream step at latent side L, per-class formulae of SURVEY.md section 8d:
    conv `2 B Cin Cout k^2 Ho Wo`, linear `2 B T Cin Cout`, self-attention `4 B heads T^2 d` (quadratic in the token
    count: cfg 5 is NOT cfg 3 x 4), cross-attention `4 B heads T 77 d`.  SD-1.x layout:
    cfg 3 = 6.49, cfg 5 = 9.41, cfg 2 = 0.479 TFLOP (tests/test_host_cpu.py pins the three)."""
    ch = [320, 640, 1280, 1280]

    def conv(ci, co, H): return 2 * ci * co * 9 * H * H
    def lin(T, ci, co): return 2 * T * ci * co

    def resnet(ci, co, H):
        return conv(ci, co, H) + conv(co, co, H) + 2 * 1280 * co + (lin(H * H, ci, co) if ci != co else 0)

    def transformer(C, H):
        T = H * H
        f = 2 * lin(T, C, C)                                   # proj_in / proj_out
        f += 4 * lin(T, C, C) + 4 * T * T * C                  # self: q, k, v, out + QK^T / PV
        f += 2 * lin(T, C, C) + 2 * lin(cross_tokens, cross_dim, C) + 4 * T * cross_tokens * C  # cross
        return f + lin(T, C, 8 * C) + lin(T, 4 * C, C)         # GEGLU feed-forward

    def net(part, cin=4, cout=4, exchange=True):
        f, H, prev = 0, L, 320
        if part in ("unet", "enc"):
            f += conv(cin, 320, L) + 2 * 320 * 1280 + 2 * 1280 * 1280
            for i, c in enumerate(ch):
                for _ in range(2):
                    f += resnet(prev, c, H) + (transformer(c, H) if i < 3 else 0)
                    prev = c
                if i < 3:
                    H //= 2
                    f += conv(c, c, H)
            f += 2 * resnet(1280, 1280, H) + transformer(1280, H)
        if part in ("unet", "dec"):
            H, prev = L // 8, 1280
            skips = [320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280]
            for i, c in enumerate([1280, 1280, 640, 320]):
                for _ in range(3):
                    f += resnet(prev + skips.pop(), c, H) + (transformer(c, H) if i > 0 else 0)
                    prev = c
                if i < 3:
                    H *= 2
                    f += conv(c, c, H)
            f += conv(320, cout, L) + (2 * 320 * 1280 + 2 * 1280 * 1280 if part == "dec" else 0)
        if part in ("enc", "dec") and exchange:  # the 12 + 1 exchange 1x1 convs
            for h, c in zip([L] * 3 + [L // 2] * 3 + [L // 4] * 3 + [L // 8] * 4, [320] * 4 + [640] * 3 + [1280] * 6):
                f += lin(h * h, c, c)
        return f

    if direction == "hoisted_inverse":
        # per step of the hoisted inverse loop (uni_renderer_amd/hoist.py): encoder conv_in + down + mid (its 13 exchange convs are
        # dead there) and decoder up + conv_out (its 13 exchange products are computed once per call)
        total = net("enc", cin=28, exchange=False) + net("dec", cout=28, exchange=False)
    elif direction == "hoisted_render":  # per step: the UNet (the encoder runs once per call)
        total = net("unet")
    else:
        total = net("unet") + net("enc", cin=28) + (net("dec", cout=28) if direction == "inverse" else 0)
    return float(B) * total


_FLOP_DIRECTIONS = ("inverse", "render", "hoisted_inverse", "hoisted_render")


def _spawn_ranks(n, script=None):
    """`python bench.py --gpus N` with no launcher: start the N ranks ourselves, exactly the way the driver's own
    command does (one process per GPU through torch.distributed.run, rendezvous on 127.0.0.1), and hand its exit code
    back.  Rank 0 of the children prints the JSON line."""
    import socket
    import subprocess

    ngpu = torch.cuda.device_count()
    if ngpu < n:
        raise SystemExit(f"bench.py --gpus {n}: only {ngpu} GPU(s) visible")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script or __file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def build_models(dev, dtype, seed=1234, fp32_state=None):
    """``fp32_state``: a list that receives the three networks' fp32 state dicts (CPU) before the cast to ``dtype`` -- the
    parameters the CPU oracle of the ``cpu_baseline`` / parity leg loads."""
    import uni_renderer_amd as U

    torch.manual_seed(seed)
    with torch.device(dev):
        unet = U.UNet2DConditionModel(in_channels=4, out_channels=4, cross_attention_dim=768, sample_size=64)
        enc = U.AttributeEncoderModel.from_unet(unet)
        dec = U.AttributeDecoderModel.from_unet(unet)
    # 4 -> 28 channel surgery of train/train.py:976-996
    enc.conv_in.weight = torch.nn.Parameter(enc.conv_in.weight.repeat(1, 7, 1, 1) * 0.142)
    enc.register_to_config(in_channels=28)
    dec.conv_out.weight = torch.nn.Parameter(dec.conv_out.weight.repeat(7, 1, 1, 1) * 0.142)
    dec.conv_out.bias = torch.nn.Parameter(dec.conv_out.bias.repeat(7) * 0.142)
    dec.register_to_config(out_channels=28)
    g = torch.Generator(device=dev).manual_seed(4321)
    for z in list(enc.controlnet_down_blocks) + [enc.controlnet_mid_block] + list(dec.control_down_blocks) + [dec.control_mid_block]:
        z.weight.data.normal_(0, 0.02, generator=g)  # exchange convs live
        z.bias.data.normal_(0, 0.02, generator=g)
    if fp32_state is not None:
        fp32_state.extend({k: v.detach().float().cpu() for k, v in m.state_dict().items()} for m in (unet, enc, dec))
    return [m.to(dev).to(dtype).eval() for m in (unet, enc, dec)]


def make_inputs(B, L, dev, dtype, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x_t = torch.randn(B, 4, L, L, device=dev, generator=g).to(dtype)
    cond = torch.randn(B, 28, L, L, device=dev, generator=g).to(dtype)
    ehs = (torch.randn(B, 77, 768, device=dev, generator=g) * 0.5).to(dtype)
    t_img = torch.randint(0, 1000, (B,), device=dev, generator=g)
    t_attr = torch.randint(0, 1000, (B,), device=dev, generator=g)
    return x_t, cond, ehs, t_img, t_attr


def measure_roofline(step_fn, by_shape=False, live_traffic=False, extra_args=()):
    """One eager step (same executor as the timed mode, launched serially on one stream) with every launch
    bracketed by HIP events on the launch stream; per kernel class sums."""
    from uni_renderer_amd import ops

    rec = []
    with torch.no_grad():
        step_fn()  # untimed, warm
        torch.cuda.synchronize()
        ops.profile_into(rec, by_shape)
        step_fn()
        torch.cuda.synchronize()
        ops.profile_into(None)
    agg = {}
    for key, fl, by, e0, e1 in rec:
        a = agg.setdefault(key, dict(calls=0, ms=0.0, flop=0.0, bytes=0.0))
        a["calls"] += 1
        a["ms"] += e0.elapsed_time(e1)
        a["flop"] += fl
        a["bytes"] += by
    total_ms = sum(a["ms"] for a in agg.values())
    table = []
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        row = dict(kernel=key, calls=a["calls"], ms=round(a["ms"], 4), share=round(a["ms"] / total_ms, 4),
                   avg_us=round(1e3 * a["ms"] / a["calls"], 2))
        row["alg_bytes_per_launch"] = round(a["bytes"] / a["calls"])
        if a["flop"] > 0:
            row["tflops"] = round(a["flop"] / (a["ms"] * 1e-3) / 1e12, 2)
            row["alg_flop_per_launch"] = round(a["flop"] / a["calls"])
            if key == "attention_d40":
                # the head dim is zero-padded for the MFMA: Q.K^T contracts over 48, P.V produces 64 rows; `tflops`
                # counts the USEFUL 40 + 40, the matrix pipe also executes the padding (MFMA-busy counts both)
                row["useful_frac_of_mfma_flops"] = round((40 + 40) / (48 + 64), 3)
        else:
            row["gbs"] = round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1)
        table.append(row)
    # Dominant kernel = the kernel SYMBOL with the largest share: split-K and plain launches of one igemm tile are the
    # same device kernel (rocprofv3 lists them under one name), so their shares are added when ranking; the figures
    # reported are those of the plain launches (a split launch's event bracket also covers its reduce kernel).
    sym_share = {}
    for row in table:
        sym_share[row["kernel"].replace("_splitk", "")] = sym_share.get(row["kernel"].replace("_splitk", ""), 0.0) + row["share"]
    dom_sym = max(sym_share, key=sym_share.get)
    dom = next((r for r in table if r["kernel"] == dom_sym), None) or next(r for r in table if r["kernel"].startswith(dom_sym))
    if "tflops" in dom:
        roof = dict(bound="mfma", kernel=dom["kernel"], achieved=dom["tflops"], peak=PEAK_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(dom["tflops"] / PEAK_MFMA_TFLOPS, 4))
    else:
        roof = dict(bound="hbm", kernel=dom["kernel"], achieved=dom["gbs"], peak=PEAK_HBM_GBS, unit="GB/s",
                    frac=round(dom["gbs"] / PEAK_HBM_GBS, 4))
    if "tflops" in dom:
        # the same figure over ALL launches of the symbol (plain + split-K; a split launch's bracket also contains its
        # reduce pass, so this is a lower bound on the main kernel's own rate) -- VERDICT r3 weak 9
        rows = [r for r in table if r["kernel"].replace("_splitk", "") == dom_sym and "tflops" in r]
        fl_all = sum(r["alg_flop_per_launch"] * r["calls"] for r in rows)
        ms_all = sum(r["ms"] for r in rows)
        roof.update(achieved_all_launches=round(fl_all / (ms_all * 1e-3) / 1e12, 2),
                    frac_all_launches=round(fl_all / (ms_all * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS, 4),
                    calls_per_step_all_launches=sum(r["calls"] for r in rows))
    roof.update(calls_per_step=dom["calls"], avg_launch_us=dom["avg_us"], share_of_step=dom["share"],
                algorithmic_bytes_per_launch=dom["alg_bytes_per_launch"],
                algorithmic_flop_per_launch=dom.get("alg_flop_per_launch"),
                share_of_step_incl_splitk_launches=round(sym_share[dom_sym], 4), traffic=None)
    live = measure_traffic_live(dom["kernel"].replace("_splitk", ""), extra_args) if live_traffic else None
    if live is not None:
        roof["traffic"], roof["traffic_source"] = live
        return roof, table, total_ms
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # HBM bytes/launch from a separate rocprofv3 --pmc pass
    if os.path.exists(tf):
        try:
            ent = json.load(open(tf)).get(dom["kernel"].replace("_splitk", ""))
            if ent:
                roof["traffic"] = ent["hbm_bytes_per_launch"]  # PMC: FETCH_SIZE*2 + WRITE_SIZE, bytes per launch
                roof["traffic_source"] = ("STATIC: read from profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / "
                                          "WRITE_SIZE passes of this command, tools/collect_profiles.sh), not measured in this run")
        except Exception:
            pass
    return roof, table, total_ms


def measure_traffic_live(klass, extra_args, steps=2, timeout=150):
    """HBM bytes per launch of kernel class ``klass`` measured NOW: two child runs of this script under
    ``rocprofv3 --pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` (separate passes: the two counters do not fit the TCC's four
    slots together; only --kernel-trace beside them), reduced exactly as tools/pmc_traffic.py does (FETCH_SIZE in KiB and
    doubled on gfx950 for 16-B-per-lane reads, WRITE_SIZE in KiB as reported: MI355X_MICROARCH.md, HBM section).
    Returns (bytes per launch, description) or None when rocprofv3 is missing or a pass fails."""
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None  # this process is itself being profiled (tools/collect_profiles.sh): no nested profiler
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_traffic
    except Exception:
        return None
    tmp = tempfile.mkdtemp(prefix="ur_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", out, "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline",
                   "--no-roofline", "--no-loop"] + list(extra_args)
            proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)  # the process group this call started (rocprofv3 + its python child)
                proc.wait()
                return None
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None
            agg = pmc_traffic.per_class(files[0], counter)
            if klass not in agg or agg[klass][1] == 0:
                return None
            got[counter] = agg[klass][0] / agg[klass][1], agg[klass][1]
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd, wr = 2.0 * 1024.0 * got["FETCH_SIZE"][0], 1024.0 * got["WRITE_SIZE"][0]
    return round(rd + wr), (f"LIVE: two rocprofv3 passes of this command (--pmc FETCH_SIZE, --pmc WRITE_SIZE; --kernel-trace only beside "
                            f"them) right after the timed region, {got['FETCH_SIZE'][1]} launches of the class averaged; "
                            f"FETCH_SIZE x 2 x 1024 = {round(rd)} B read + WRITE_SIZE x 1024 = {round(wr)} B written per launch")


def cpu_baseline(B, L, check=None):
    """The oracle ("port") timed on this host: the same enc+unet+dec step, fp32, all cores.
    ``check`` = (fp32 state dicts of [unet, enc, dec], the GPU step's inputs on the CPU, its outputs on the CPU, low dtype):
    the timed CPU steps then run on THE BENCHMARKED networks and inputs, and their outputs double as the parity check of
    the number above them -- rel-L2 of img_pred / attr_pred against the oracle holding the fp32 parameters and against the
    oracle holding the parameters rounded to the product's dtype (``parity_rel_l2``; north_star: <= 1e-3 in fp16)."""
    from oracle import unirenderer_oracle as O

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # honour a cgroup CPU quota (a 256-thread pool on a quota of a few cores thrashes)
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(p))))
    except Exception:
        pass
    cores = min(cores, 64)  # one socket's worth of physical cores; oneDNN does not scale past that here
    torch.set_num_threads(cores)
    with torch.device("meta"):
        unet = O.UNet2DConditionModel(**O.SD15_CONFIG)
        enc = O.AttributeEncoderModel(**dict(O.SD15_CONFIG, in_channels=28))
        dec = O.AttributeDecoderModel(**dict(O.SD15_CONFIG, out_channels=28))
    mods = []
    g = torch.Generator().manual_seed(1)
    for i, m in enumerate((unet, enc, dec)):
        m = m.to_empty(device="cpu")
        if check is not None:
            m.load_state_dict(check[0][i])  # the benchmarked networks' fp32 parameters
        else:
            for p in m.parameters():
                p.data.normal_(0, 0.02, generator=g)
        mods.append(m.eval())
    # Bounded sample: one untimed batch-1 step (allocator / oneDNN primitive warm-up; also sizes the
    # budget), then THREE timed full steps at the benchmarked batch -- min and median reported, ``value`` = 1 / median
    # -- unless a full step would blow the ~60 s budget, in which case batch-1 steps are timed and scaled (samples are
    # independent on the CPU).  Plus cfg 1 exactly: the UNet alone, batch 1, 2 steps.
    xw = O.make_inputs(1, L, 768, seed=3)
    t0 = time.perf_counter()
    O.dual_stream_step(*mods, *xw)
    warm = time.perf_counter() - t0
    if warm > 45.0:
        return dict(value=round(1.0 / (warm * B), 6), unit="denoise-steps/sec", cores=cores, kind="port",
                    sample=f"ONE untimed-warm batch-1 step took {warm:.1f} s (over budget): value = 1/({warm:.1f} s x {B}); "
                           f"{L}x{L} latent, fp32, torch {torch.__version__} CPU ops")
    full = warm * B <= 25.0
    nb = B if full else 1
    if check is not None:
        x = tuple(t[:nb].float() if t.is_floating_point() else t[:nb] for t in check[1])
    else:
        x = O.make_inputs(nb, L, 768, seed=4)
    times, parity = [], None
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    for k in range(3):
        if check is not None and k == 2:  # the third timed step runs on the parameters ROUNDED like the product's
            for m in mods:
                for p in m.parameters():
                    p.data = p.data.to(check[3]).to(torch.float32)
        t0 = time.perf_counter()
        out = O.dual_stream_step(*mods, *x)
        times.append((time.perf_counter() - t0) * (1 if full else B))
        if check is not None and k in (0, 2):
            parity = parity or {}
            parity["fp32_weights" if k == 0 else "same_weights"] = dict(
                img_pred=round(rel(check[2]["img_pred"][:nb], out["img_pred"]), 7),
                attr_pred=round(rel(check[2]["attr_pred"][:nb], out["attr_pred"]), 7))
    times.sort()
    med = times[1]
    what = (f"1 untimed batch-1 warm-up + 3 timed full steps at batch {B}" if full else
            f"1 untimed batch-1 warm-up + 3 timed batch-1 steps x {B} (samples are independent)")
    out = dict(value=round(1.0 / med, 5), unit="denoise-steps/sec", cores=cores, kind="port",
               step_seconds=dict(min=round(times[0], 3), median=round(med, 3), max=round(times[2], 3)),
               sample=f"{what}: min {times[0]:.2f} s / median {med:.2f} s; same enc+unet+dec step"
                      f"{' on the benchmarked networks and inputs' if check is not None else ''}, {L}x{L} latent, fp32, "
                      f"torch {torch.__version__} CPU ops (oneDNN {'on' if torch.backends.mkldnn.is_available() else 'off'}), "
                      f"{cores} threads")
    if parity is not None:
        parity["samples_compared"] = nb
        parity["note"] = ("rel-L2 of the timed GPU step's outputs against the CPU fp32 oracle (parity unpinned: restatement of the "
                          "reference, SURVEY 8c) on the SAME networks and inputs: fp32_weights = oracle holds the fp32 parameters "
                          "(includes the checkpoint's quantisation to the product's dtype), same_weights = oracle holds the "
                          "parameters rounded like the product's (kernel arithmetic only; north_star's <= 1e-3 is claimed for this one)")
        out["parity_rel_l2"] = parity
    # cfg 1 (BASELINE.json configs[0]): single-stream UNet forward, 64x64 latent, bs 1, 2 denoise steps, CPU fp32
    x1 = O.make_inputs(1, 64, 768, seed=5)
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in range(2):
            mods[0](x1[0], x1[3], x1[2])
        c1 = time.perf_counter() - t0
    out["cfg1_unet_only_bs1_2steps"] = dict(seconds=round(c1, 3), steps_per_sec=round(2.0 / c1, 4))
    return out


def sampling_loops(models, B, L, dev, dtype, steps=50):
    """cfg 3's actual workload: the 50-step DDIM sampling loops of the pipeline (latents in, latents out; VAE / CLIP outside),
    inverse (real_image2mask_3mod_albedo) and rendering (mask2image_3mod_albedo) direction, with the loop-invariant half
    hoisted out of the loop (uni_renderer_amd/hoist.py) -- wall time of a whole call, best of 3."""
    from uni_renderer_amd.pipeline import UniRendererPipeline

    pipe = UniRendererPipeline(unet=models[0], controlnet=models[1], controldec=models[2])
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator(device=dev).manual_seed(3)
    img = torch.randn(B, 4, L, L, device=dev, generator=g).to(dtype)
    msk = torch.randn(B, 4, L, L, device=dev, generator=g).to(dtype)
    ehs = (torch.randn(B, 77, 768, device=dev, generator=g) * 0.5).to(dtype)
    attr = torch.randn(B, 28, L, L, device=dev, generator=g).to(dtype)
    res = dict(steps=steps, scheduler="DDIM (x0 prediction), on-device update", guidance_scale=0.0,
               executor="hoisted: inverse = UNet down+mid + decoder exchange convs once per call, per step enc + 13 adds + dec; "
                        "render = encoder once per call, per step UNet + 13 adds")
    for name, fn in (
        ("inverse", lambda: pipe.real_image2mask_3mod_albedo(prompt_embeds=ehs, image_latents=img, mask_latents=msk,
                                                             num_inference_steps=steps, guidance_scale=0.0, output_type="latent")),
        ("render", lambda: pipe.mask2image_3mod_albedo(prompt_embeds=ehs, attr_latents=attr, num_inference_steps=steps,
                                                       guidance_scale=0.0, output_type="latent")),
    ):
        fn()  # capture + warm
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[name + "_ms_total"] = round(min(ts) * 1e3, 2)
        res[name + "_ms_per_step"] = round(min(ts) * 1e3 / steps, 3)
    # the reference's live eval protocol (eval/test_real.py:485-492, 547-564): UniPC, 20 steps, guidance 0, the same image five
    # times (compute_times) -- folded into ONE batch of 5 (num_images_per_prompt = 5)
    from uni_renderer_amd.pipeline import SCHEDULER_NAMES
    from uni_renderer_amd.schedulers import UniPCMultistepScheduler

    for n in SCHEDULER_NAMES:
        setattr(pipe, f"scheduler_{n}", UniPCMultistepScheduler())
    fn = lambda: pipe.real_image2mask_3mod_albedo(prompt_embeds=ehs[:1], image_latents=img[:1], mask_latents=msk[:1],
                                                  num_inference_steps=20, guidance_scale=0.0, num_images_per_prompt=5,
                                                  output_type="latent")
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    res["eval_protocol_unipc20_x5_folded_ms_total"] = round(min(ts) * 1e3, 2)
    # roofline of the per-step half of the hoisted inverse loop (what a sampling call replays 50 times): one eager run with every
    # launch bracketed by HIP events, dominant kernel by symbol as for the full step
    try:
        from uni_renderer_amd.graph import GraphedHoistedStep

        gh = next(g for g in pipe._graphs.values() if isinstance(g, GraphedHoistedStep) and g.run_decoder and g.x_t.shape[0] == B)
        gh.begin()
        roof, table, total_ms = measure_roofline(gh._run)
        # per-step FLOP of the hoisted inverse loop: encoder conv_in + down + mid and decoder up + conv_out
        fl = (algorithmic_flops(L, B, "hoisted_inverse") if "hoisted_inverse" in _FLOP_DIRECTIONS else None)
        res["roofline"] = dict(roof, launches_per_step=sum(r["calls"] for r in table), sum_kernel_ms_eager_step=round(total_ms, 3),
                               traffic_source="not measured for the loop step (the full step's line carries the PMC traffic of the same kernel)")
        if fl:
            res["algorithmic_tflop_per_step"] = round(fl / 1e12, 3)
            res["inverse_step_frac_of_mfma_peak"] = round(fl / 1e12 / res["inverse_ms_per_step"] / PEAK_MFMA_TFLOPS * 1e3, 4)
    except Exception as e:
        res["roofline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return res


def loop_parity(dev, dtype):
    """CHECKER leg (with ``cpu_baseline``): the pipeline's hoisted 50-step DDIM inverse loop -- the loop ``loop.inverse_ms_total``
    times -- on the SD-size networks the oracle builds from seed 1234, against the final latents of the same loop on the CPU
    fp32 oracle, COMMITTED as tests/golden/sd_cfg3_b4.safetensors (tests/golden/make_golden_sd.py: ~50 min of CPU in the build
    container).  rel-L2 of the [4, 24, 64, 64] attribute latents after 50 steps against the oracle holding the same fp16-rounded
    parameters and the fp32 parameters.  fp16, batch 4, 64x64 only (what the golden holds)."""
    gold_path = os.path.join(ROOT, "tests", "golden", "sd_cfg3_b4.safetensors")
    if dtype != torch.float16 or not os.path.exists(gold_path):
        return None
    from safetensors.torch import load_file

    from oracle import unirenderer_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util_models import build_product_from_oracle
    from uni_renderer_amd.pipeline import UniRendererPipeline

    gold = load_file(gold_path)
    t0 = time.perf_counter()
    oracle = O.build_triplet(O.SD15_CONFIG, seed=1234)
    nets = build_product_from_oracle(*oracle, dtype, dev)
    del oracle
    pipe = UniRendererPipeline(unet=nets[0], controlnet=nets[1], controldec=nets[2])
    pipe.set_progress_bar_config(disable=True)
    x, c, ehs, _, _ = [t.to(dev) for t in O.make_inputs(4, 64, 768, seed=28, t_img=0)]
    sched = pipe.scheduler_attr
    sched.set_timesteps(50)
    fin = pipe._fused_loop(x.to(dtype), c, ehs.to(dtype), sched.timesteps, sched, run_decoder=True, lat_dtype=torch.float32).float().cpu()
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    return dict(final_latents_same_weights=round(rel(fin, gold["loop.final_latents.fp16w"]), 7),
                final_latents_fp32_weights=round(rel(fin, gold["loop.final_latents.fp32w"]), 7),
                oracle_fp16w_vs_fp32w=round(rel(gold["loop.final_latents.fp16w"], gold["loop.final_latents.fp32w"]), 7),
                seconds=round(time.perf_counter() - t0, 1),
                note="hoisted on-device 50-step DDIM inverse loop (B = 4, 64x64, fp16) vs the committed CPU-oracle loop "
                     "(tests/golden/sd_cfg3_b4.safetensors); networks rebuilt from the golden's seed")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch (headline: 4)")
    ap.add_argument("--latent", type=int, default=64, help="latent side (headline: 64 = 512x512 image)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--direction", default="inverse", choices=["inverse", "render"],
                    help="inverse: enc+unet+dec (headline, cfg 3/5); render: enc+unet only (cfg 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-loop", action="store_true", help="skip the 50-step sampling-loop timings (the `loop` object)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from profiles/pmc_traffic.json instead of two rocprofv3 --pmc child runs (~1 min)")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of the HIP graph")
    ap.add_argument("--no-concurrent", action="store_true", help="capture the three networks serially on one stream")
    ap.add_argument("--mode", default=None, choices=["grouped", "concurrent", "serial"],
                    help="step execution: grouped launches of the two streams (default), two concurrent graph branches, or serial")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel-class table to stderr")
    ap.add_argument("--shape-table", default="", help="write a per-(kernel, problem shape) table (JSON) to this path")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            _spawn_ranks(args.gpus)  # no launcher around us: become one (does not return)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_on = world > 1 or "RANK" in os.environ  # under a launcher even one rank times through RCCL's barrier / max-reduce
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16

    from uni_renderer_amd import _lib
    from uni_renderer_amd.graph import GraphedDualStreamStep, dual_stream_step

    _lib.load()  # no HIP library -> fail loudly
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    fp32_state = [] if want_cpu else None
    models = build_models(dev, dtype, fp32_state=fp32_state)
    inputs = make_inputs(args.batch, args.latent, dev, dtype, seed=100 + rank)
    runner = GraphedDualStreamStep(*models, batch=args.batch, latent_hw=args.latent, cross_dim=768, dtype=dtype,
                                   device=dev, run_decoder=(args.direction == "inverse"),
                                   concurrent=not args.no_concurrent, mode=args.mode)
    runner.load_inputs(*inputs)
    if args.eager:
        def one():
            with torch.no_grad():
                dual_stream_step(*models, runner.x_t, runner.cond, runner.ehs, runner.t_img, runner.t_attr,
                                 run_decoder=(args.direction == "inverse"))
        one()
    else:
        runner.capture()
        one = runner.replay

    def barrier():
        if dist_on:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        one()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        out = {
            "metric": "denoise-steps/sec (dual-UNet, 512^2 bs=4)",
            "value": round(world * args.steps / dt, 4),
            "unit": "denoise-steps/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if dtype == torch.float16 else "bf16",
            "data": "synthetic",
            "config": {
                "workload": ("inverse-rendering denoise step: AttributeEncoderModel + UNet2DConditionModel + "
                             "AttributeDecoderModel forward (SD-1.x size, 1.74 G params, random init), "
                             if args.direction == "inverse" else
                             "rendering denoise step: AttributeEncoderModel + UNet2DConditionModel forward (SD-1.x size, "
                             "random init), ") +
                            f"{args.latent * 8}x{args.latent * 8} image = {args.latent}x{args.latent} latent, 28-channel "
                            f"attribute latent, 77x768 prompt embedding, batch {args.batch} per GPU",
                "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                "launch": "eager" if args.eager else {"grouped": "hipGraph replay; enc||unet.down and unet.up||dec issued as grouped (zbatch=2) launches",
                                                      "concurrent": "hipGraph replay, 2 concurrent branches (enc || unet.down, dec || unet.up)",
                                                      "serial": "hipGraph replay, serial"}[runner.mode],
                "algorithmic_tflop_per_step": round(algorithmic_flops(args.latent, args.batch, args.direction) / 1e12, 3),
                "residual_stream": ("(hi, lo) pairs (parity <= 1e-3, DESIGN.md section 5)"
                                    if os.environ.get("UR_PRECISE_RESIDUAL", "1") != "0" else "plain (UR_PRECISE_RESIDUAL=0)"),
            },
        }
        if not args.no_roofline:  # rank 0 only (this block), at every N: the kernels are the same on every rank
            side, runner.side = runner.side, None  # serial launches for per-kernel timing
            # HBM traffic of the dominant kernel: measured live (one GPU, headline-style run) by two PMC child runs of the
            # same workload; the committed profile is the fallback
            child = ["--batch", str(args.batch), "--latent", str(args.latent), "--dtype", args.dtype, "--direction", args.direction]
            roof, table, total_ms = measure_roofline(runner._run, live_traffic=(world == 1 and not args.no_live_traffic
                                                                                and not args.eager), extra_args=child)
            out["roofline"] = roof
            # the whole step against the MFMA roofline: algorithmic FLOP of the step / measured step time / dense peak
            out["step_frac_of_mfma_peak"] = round(out["config"]["algorithmic_tflop_per_step"] / out["ms_per_step"]
                                                  / PEAK_MFMA_TFLOPS * 1e3, 4)
            out["kernel_classes"] = table[:8]
            out["config"]["sum_kernel_ms_eager_step"] = round(total_ms, 3)
            if args.kernel_table:
                for r in table:
                    print(json.dumps(r), file=sys.stderr)
            if args.shape_table:
                _, stable, _ = measure_roofline(runner._run, by_shape=True)
                with open(args.shape_table, "w") as f:
                    json.dump(stable, f)
        profiled = (any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ)
                    or "rocprofiler" in os.environ.get("LD_PRELOAD", ""))  # a counter pass over 400 more graph replays takes minutes
        if world == 1 and not args.no_loop and not args.eager and not profiled:
            try:
                out["loop"] = sampling_loops(models, args.batch, args.latent, dev, dtype)
            except Exception as e:  # the headline line must not depend on this leg
                out["loop"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if want_cpu:
            check = None
            if not args.eager and args.direction == "inverse":
                one()  # the timed step's outputs on the benchmarked inputs
                torch.cuda.synchronize()
                check = (fp32_state, [t.cpu() for t in inputs], {k: v.float().cpu() for k, v in runner.out.items()}, dtype)
            del runner
            out["cpu_baseline"] = cpu_baseline(args.batch, args.latent, check)
            if "parity_rel_l2" in out["cpu_baseline"]:
                out["config"]["parity_rel_l2"] = out["cpu_baseline"].pop("parity_rel_l2")
            if isinstance(out.get("loop"), dict) and "error" not in out["loop"] and args.batch == 4 and args.latent == 64:
                try:  # the checker of the loop timings above: its final latents against the committed oracle loop
                    del models
                    torch.cuda.empty_cache()
                    lp = loop_parity(dev, dtype)
                    if lp is not None:
                        out["loop"]["parity_rel_l2"] = lp
                except Exception as e:
                    out["loop"]["parity_rel_l2"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.barrier()  # rank 0 may still have been in its roofline leg
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
