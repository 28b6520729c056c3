#!/usr/bin/env python3
"""bench.py — DDIM denoise steps/sec on MuDG's MDM1024 configuration (576x1024x16f) on N MI355X GPUs.

One "step" = one p_sample_ddim of the reference (lvdm/models/samplers/ddim.py:205-279): two full UNet forwards
(conditional + unconditional, classifier-free guidance 7.5), guidance rescale 0.7, v-prediction, dynamic rescale and
the x_{t-1} update with eta = 1 — for ONE clip (B = 1 latents (1, 4, 16, 72, 128), 12-channel hybrid input,
context (1, 333, 1024)).  Weights and inputs are synthetic (seeded), resident in HBM before the timed region.

Multi-GPU (`torchrun --nproc-per-node N bench.py --gpus N`): the path shards by independent clips — every rank
denoises its own clip with a full weight replica; there is no collective inside the step.  RCCL is used only for the
start/stop barriers and the max-over-ranks reduction of the elapsed time.  value = N * steps / max-rank time (weak
scaling: per-GPU work is fixed).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel family, measured live with hipEvents on the launch
stream) and `cpu_baseline` (the CPU oracle timed on a bounded sample, N = 1 only).
"""
import argparse
import contextlib
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--resolution", default="1024", choices=["1024", "512"])
    ap.add_argument("--batch", type=int, default=1, help="clips per GPU per step (reference driver uses 3 modalities)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full"],
                    help="sample: one oracle forward at MDM512's spatial size on a 4-frame clip (3.15 TFLOP, tens of seconds); "
                         "full: one whole MDM512 forward (12.6 TFLOP, minutes)")
    ap.add_argument("--no-profile", action="store_true", help="skip hipEvent bracketing of kernel families")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--profile-steps", type=int, default=2, help="eager steps re-run with hipEvents for the roofline")
    ap.add_argument("--no-decode", action="store_true", help="skip the (untimed) 16-frame VAE decode used for clips/min")
    ap.add_argument("--no-at-tolerance", action="store_true",
                    help="skip the bf16x3 child run (the operand mode that meets the 1e-3 decoded-frame tolerance) whose "
                         "steps/s is reported as `at_tolerance` next to the bf16 line")
    ap.add_argument("--no-children", action="store_true", help="no child runs at all (set by the children themselves)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the child runs of BASELINE configs[1] (MDM512) and configs[4] (MX-fp8 attention scores)")
    ap.add_argument("--no-training", action="store_true",
                    help="skip the training-step child run (SURVEY §8 f4: p_losses -> backward -> AdamW of the same UNet at the same "
                         "resolution) whose seconds per step are reported as `training_step` next to the inference line")
    ap.add_argument("--operand", default=os.environ.get("MUDG_OPERAND", "bf16"), choices=["bf16", "fp16", "bf16x3", "bf16x6"],
                    help="MFMA operand type: bf16 (the BASELINE dtype), fp16 (the reference's autocast dtype), or the "
                         "split-operand precision modes bf16x3 / bf16x6 (2 / 3 bf16 pieces per value, 3 / 6 MFMAs per product "
                         "tile: the modes that meet the 1e-3 decoded-frame tolerance; their cost is what this flag measures)")
    return ap.parse_args()


PEAK_TFLOPS_BF16 = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
MEASURED_MFMA_PEAK_TFLOPS = 1747.5    # tools/ubench/mfma_peak.hip: v_mfma_f32_32x32x16_bf16, random operands, best of 1-8 waves/SIMD (zeros: 2393.8)
MFMA_FAMS = ("gemm", "conv3x3", "tconv3", "attention")


def cpu_baseline(model, inputs, resolution, mode):
    """The CPU oracle (torch fp32 eager, the same op graph as the reference's einsum attention path; pinned to vectors
    captured from the reference, tests/test_oracle_golden.py) timed on this host's cores with this run's weights.

    mode "sample" (default; bounded to tens of seconds as the bench contract asks): ONE UNet forward at MDM512's spatial
    size (40 x 64 latents, the real 1.44 B-parameter topology, context 1024) on a 4-frame clip = 3.15 TFLOP.
    mode "full": ONE full MDM512 forward (BASELINE configs[0]'s model and shape: 16 frames, 12.6 TFLOP; minutes).
    Either way steps/s at the benchmarked configuration = measured TFLOP/s / algorithmic TFLOP per CFG step (MDM1024 is
    4.15x MDM512 by FLOPs); the einsum path's 27 GB score tensors rule out a direct MDM1024 run.
    A recorded full-size measurement (profiles/r*/cpu_baseline_*.json), if committed, is quoted next to the live one; `value` is
    always the live one."""
    from mudg_amd import configs
    from oracle import unet as o_unet
    threads = torch.get_num_threads()
    t, h, w = (16, 40, 64) if mode == "full" else (4, 40, 64)
    name = "MDM512, 16 frames (BASELINE configs[0] shape)" if mode == "full" else "MDM512 spatial size, 4-frame clip"
    cfg = dict(configs.UNET_MDM, temporal_length=t)
    sd = {k: v.detach().float().cpu() for k, v in model.model.diffusion_model.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 12, t, h, w, generator=g)
    ctx = torch.randn(1, 77 + 16 * t, 1024, generator=g)
    ts, lab, fs = torch.full((1,), 499), torch.zeros(1, dtype=torch.long), torch.full((1,), 10)
    from torch.utils.flop_counter import FlopCounterMode
    t0 = time.perf_counter()
    with FlopCounterMode(display=False) as fc:
        o_unet.unet_forward(sd, cfg, x, ts, lab, ctx, fs, head_chunk=8)
    dt = time.perf_counter() - t0
    flops = fc.get_total_flops() / 1e12
    full = 2 * configs.UNET_TFLOP[resolution]
    out = {"value": (flops / dt) / full, "unit": "steps/s", "cores": threads, "kind": "port",
           "sample": f"one CPU-oracle UNet forward (fp32, {threads} threads) at {name}: latent {t}x{h}x{w} = "
                     f"{flops:.2f} TFLOP in {dt:.1f} s -> {flops / dt:.3f} TFLOP/s; steps/s = that / "
                     f"{full:.2f} TFLOP per CFG step at MDM{resolution}",
           "tflops": flops / dt}
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cpu_baseline_full.json")))
    if recs and mode != "full":
        try:
            with open(recs[-1]) as f:
                rec = json.load(f)
            out["recorded_full_mdm512_forward"] = {"file": os.path.relpath(recs[-1], ROOT), "tflops": rec["tflops"],
                                                   "cores": rec["cores"], "seconds": rec["seconds"],
                                                   "steps_per_s_at_this_config": rec["tflops"] / full}
        except Exception:
            pass
    rec1024 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cpu_baseline_mdm1024.json")))
    if rec1024 and resolution == "1024":
        try:
            with open(rec1024[-1]) as f:
                rec = json.load(f)
            out["recorded_full_mdm1024_forward"] = {"file": os.path.relpath(rec1024[-1], ROOT), "tflops": rec["tflops"], "cores": rec["cores"],
                                                    "seconds": rec["seconds"], "steps_per_s_at_this_config": rec["tflops"] / full}
        except Exception:
            pass
    out["seconds"] = dt
    # `value` is the sample timed on THIS box in THIS run (north_star: "timed on the same box's host cores in the same run").  A
    # recorded whole forward at the benchmarked size on the same host class, when one is committed, rides along as an
    # annotation: short clips use the cores worse, the two differ by 1.2-1.6x.
    key = "recorded_full_mdm1024_forward" if resolution == "1024" else "recorded_full_mdm512_forward"
    if key in out:
        out["sample"] += (f"; for reference, the recorded full {key[14:-8].upper()} oracle forward ({out[key]['file']}: "
                          f"{out[key]['tflops']:.3f} TFLOP/s on {out[key]['cores']} threads) = "
                          f"{out[key]['steps_per_s_at_this_config']:.5f} steps/s")
    return out


def pmc_traffic(family, args, launches_per_step=None):
    """HBM-side bytes per launch of `family` from the committed rocprofv3 PMC passes (profiles/rN/traffic.json: FETCH_SIZE
    and WRITE_SIZE collected in separate passes of this same command, FETCH doubled per the gfx950 correction).  PMC
    counters cannot be read from inside the process, so this is the recorded figure for the default workload only."""
    if args.resolution != "1024" or args.batch != 1 or args.operand != "bf16":
        return {}            # the counters were collected on the default bf16 workload only
    rounds = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*", "traffic.json")))
    if not rounds:
        return {}
    try:
        with open(rounds[-1]) as f:
            rec = json.load(f)
        fam = rec["families"][family]
        per_launch = fam["bytes_per_step"] / launches_per_step if (launches_per_step and "bytes_per_step" in fam) else fam["bytes_per_launch"]
        return {"traffic": round(per_launch),
                "traffic_source": f"{os.path.relpath(rounds[-1], os.path.dirname(os.path.abspath(__file__)))}: {rec['method']}; these are the L2's "
                                  "fabric-side requests — Infinity-Cache hits included (MI355X_MICROARCH.md HBM section), so operand panels "
                                  "re-read once the 4-MB L2 of an XCD has lost them count although they never reach HBM"}
    except Exception:
        return {}


def child_bench(args, what, flags=(), env_extra=None, steps=4, warmup=1):
    """The same bench in a child process with a different operand mode / resolution / switch (the operand type is a property of
    the loaded library and the switches are read at import: one configuration per process).  A few steps after one warm-up; the
    child's own roofline record rides along.  Children run no children, no CPU baseline and no decode."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup), "--batch", str(args.batch),
           "--no-cpu-baseline", "--no-decode", "--no-children", "--profile-steps", "1", *flags]
    if args.no_graph:
        cmd.append("--no-graph")
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return {"config": what, "value": None, "error": f"{type(e).__name__}: {e}"}
    return {"config": what, "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "steps": rec["steps"],
            "warmup": rec["warmup"], "dtype": rec["dtype"], "output_finite": rec.get("output_finite"), "roofline": rec.get("roofline"),
            "kernels": [{k: f[k] for k in ("family", "ms_per_step", "achieved", "unit", "frac")} for f in rec.get("kernels", [])]}


def at_tolerance(args):
    """The same workload in the operand mode that meets north_star's tolerance (decoded frames within 1e-3 rel-L2 of the
    reference): bf16x3.  The parity figures quoted are the last measured ones of that mode (profiles/rN/parity_modes.json,
    written from the `-m gpu` run of tests/test_pipeline_gpu.py and tests/test_fullsize_gpu.py, which assert the literal 1e-3
    in that mode)."""
    out = child_bench(args, "BASELINE configs[2] in the bf16x3 operand mode", ["--operand", "bf16x3", "--resolution", args.resolution],
                      {"MUDG_OPERAND": "bf16x3"})
    out["operand"] = "bf16x3"
    out["tolerance"] = "decoded frames within 1e-3 rel-L2 of the reference (BASELINE.json north_star)"
    par = _parity_of("bf16x3")
    if par:
        out["parity_last_measured"] = par
    return out


def _parity_of(mode):
    """The last measured parity figures of an operand mode / switch (profiles/rN/parity_modes.json, written by the `-m gpu` suite)."""
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "parity_modes.json")))
    if not recs:
        return None
    try:
        with open(recs[-1]) as f:
            return dict(json.load(f).get(mode, {}), file=os.path.relpath(recs[-1], ROOT))
    except Exception:
        return None


def other_configs(args):
    """The other single-GPU configurations, on the driver's clock: BASELINE configs[1] (MDM512, bf16), configs[4] (MDM1024 with MX-fp8
    attention scores; GroupNorm-SiLU is fused and the decode frame-batched in every configuration) with its measured parity next to
    it, and the reference's OWN operating point — fp16 operands (it runs under torch.autocast(fp16),
    virtual_render/virtual_pose_render.py:218) and its 3-modality batch (colour / depth / semantic clips denoised together, :90-100,
    206-213; `value` there is CLIP-steps/s: 3 clips advance per step)."""
    out = {"mdm512": child_bench(args, "BASELINE configs[1]: MDM512 320x512x16f, 50 DDIM steps, bf16", ["--operand", "bf16", "--resolution", "512"],
                                 {"MUDG_OPERAND": "bf16"}, steps=8, warmup=2),
           "fp8_scores": child_bench(args, "BASELINE configs[4]: MDM1024 with MX-fp8 scores in the long self-attention (MUDG_ATTN_FP8=1)",
                                     ["--operand", "bf16", "--resolution", "1024"], {"MUDG_OPERAND": "bf16", "MUDG_ATTN_FP8": "1"}),
           "fp16": child_bench(args, "MDM1024 with fp16 MFMA operands (the reference's autocast dtype), fp32 residual stream",
                               ["--operand", "fp16", "--resolution", "1024"], {"MUDG_OPERAND": "fp16"}),
           "b3": child_bench(args, "MDM1024, the reference driver's 3-modality batch (B = 3 clips per step), bf16; value = clip-steps/s",
                             ["--operand", "bf16", "--resolution", "1024", "--batch", "3"], {"MUDG_OPERAND": "bf16"}, steps=3, warmup=1)}
    for key, mode in (("fp8_scores", "bf16+fp8scores"), ("fp16", "fp16")):
        par = _parity_of(mode)
        if par:
            out[key]["parity_last_measured"] = par
    return out


def training_step_record(args):
    """One optimisation step of the same UNet at the same resolution, in a child process so that its 111 GiB of
    activations never share the allocator with the inference replica: tools/train_bench.py, two timed steps after a warm-up."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), args.resolution, "2", "json"]
    env = dict(os.environ, MUDG_OPERAND="bf16")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return {"s_per_step": None, "error": f"{type(e).__name__}: {e}"}


def setup(args, local):
    """Device and collective backend of this rank (the plumbing test swaps in cpu / gloo)."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    return torch.device("cuda", local), "nccl"


def sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize()


def make_workload(args, device, rank, dist):
    """Build the library (rank 0 only, the others wait), the synthetic model replica and this rank's clip; returns the pieces the
    timed loop needs.  Rank r denoises clip r: its own seeded inputs, a full weight replica, no collective inside a step."""
    from mudg_amd import build as mbuild, factory, hip, parallel
    if rank == 0:
        mbuild.build(verbose=False)          # no-op when the in-tree .so files are current; one rank only (no races)
    parallel.barrier(dist)
    hip.set_operand(args.operand)
    hip.lib()
    from lvdm.models.samplers.ddim import DDIMSampler
    torch.manual_seed(parallel.clip_seed(123, rank) % (2 ** 31))   # rank r denoises clip r (one clip per GPU per step)
    with contextlib.redirect_stdout(sys.stderr):       # the boundary modules print like the reference's; keep stdout = ONE JSON line
        model = factory.build_synthetic_model(args.resolution, device, seed=123)
    inp = factory.synthetic_inputs(model, args.resolution, args.batch, device, seed=parallel.clip_seed(123, rank) % (2 ** 31))
    sampler = DDIMSampler(model)
    S = 50
    sampler.make_schedule(S, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    kw = dict(unconditional_guidance_scale=7.5, unconditional_conditioning=inp["uc"], guidance_rescale=0.7,
              fs=inp["fs"], sparse_x=inp["sparse_x"], class_label=inp["class_label"], cfg_img=None,
              unconditional_conditioning_img_nonetext=None)

    def run(n, x, start_index):
        for i in range(n):
            index = (start_index - i) % S
            ts = torch.full((args.batch,), int(sampler.ddim_timesteps[index]), device=device, dtype=torch.long)
            x, _ = sampler.p_sample_ddim(x, inp["cond"], ts, index=index, **kw)
        return x

    model.model.diffusion_model.use_hip_graph = not args.no_graph
    return {"run": run, "x0": inp["x_T"], "S": S, "model": model, "inp": inp, "hip": hip}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torchrun with --nproc-per-node {args.gpus}")
    device, backend = setup(args, local)
    from mudg_amd import parallel
    dist = parallel.init_from_env(backend)[3]
    wl = make_workload(args, device, rank, dist)
    run, S, model, hip = wl["run"], wl["S"], wl["model"], wl["hip"]
    use_graph = not args.no_graph
    x = run(args.warmup, wl["x0"], S - 1)
    profile = not args.no_profile and hip is not None
    if profile and not use_graph:            # eager mode: the timed region itself is bracketed with hipEvents
        hip.prof_reset()
        hip.prof_enable((1 << len(hip.FAM_NAMES)) - 1)
    sync(device)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    x = run(args.steps, x, S - 1 - args.warmup)
    sync(device)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, dist, device)
    finite = bool(torch.isfinite(x).all().item())

    # One clip = 50 such steps + one AutoencoderKL decode of its 16 frames (outside the timed region; reported so that
    # clips/min is a measured figure, not steps/s divided by 50).
    decode_ms = None
    if not args.no_decode and model is not None:
        z = x[:1].contiguous()
        model.decode_first_stage(z)                      # warm-up: weight packing, allocator
        sync(device)
        td = time.perf_counter()
        frames = model.decode_first_stage(z)
        sync(device)
        decode_ms = 1000.0 * (time.perf_counter() - td)
        finite = finite and bool(torch.isfinite(frames).all().item())
        del frames

    fams, prof_steps = [], args.steps
    if profile and use_graph:
        # a hipGraph replay cannot carry per-kernel events: re-run a few of the same steps eagerly (identical launches,
        # same stream) with every kernel family bracketed by hipEvents
        prof_steps = max(1, args.profile_steps)
        hip.prof_reset()
        hip.prof_enable((1 << len(hip.FAM_NAMES)) - 1)
        run(prof_steps, x, S - 1 - args.warmup)
        sync(device)
    if profile:
        fams = [hip.prof_collect(i) for i in range(len(hip.FAM_NAMES))]
        hip.prof_enable(0)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    from mudg_amd import configs
    steps_per_s = world * args.batch * args.steps / elapsed
    stream_name = "fp16"
    if hip is not None:
        from mudg_amd import ops as _ops
        stream_name = {torch.float16: "fp16", torch.float32: "fp32"}[_ops.STREAM()]
    step_tflop = 2 * configs.UNET_TFLOP[args.resolution]
    out = {
        "metric": "DDIM denoise steps/sec, MDM1024 576x1024x16f (CFG: 2 UNet forwards + fused update per step, per clip)"
        if args.resolution == "1024" else "DDIM denoise steps/sec, MDM512 320x512x16f",
        "value": round(steps_per_s, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.operand, "data": "synthetic",
        "config": {"workload": f"MDM{args.resolution} latents (B={args.batch},4,16,"
                               f"{configs.LATENT_SHAPE[args.resolution][2]},{configs.LATENT_SHAPE[args.resolution][3]}) "
                               f"+ c_concat 8ch, context (B,333,1024), 50-step uniform_trailing DDIM schedule, "
                               f"cfg 7.5, guidance_rescale 0.7, eta 1.0, v-prediction, dynamic rescale",
                   "clips_per_gpu": args.batch, "parallelism": f"clip-DP x{world} (no in-step collective)",
                   "launch": "hipGraph replay of each UNet pass (cond+uncond batched)" if use_graph else "eager launches",
                   "weights": f"seeded N(0,0.02^2) incl. zero-init tensors, fp32 params -> {args.operand} MFMA operands, "
                              f"fp32 accumulate / norms / softmax, residual stream {stream_name}"},
        "algorithmic_tflop_per_step": step_tflop,
        # the REFERENCE's FLOPs per step over the measured time: the build issues fewer (the guidance passes share the
        # context-free prefix, 3.4 T; the sub-pixel upsample convs issue 4/9 of theirs), so this is a reference-equivalent rate
        "reference_equivalent_tflops_per_gpu": round(step_tflop * args.batch * args.steps / elapsed, 2),
        "reference_equivalent_frac_of_bf16_mfma_peak": round(step_tflop * args.batch * args.steps / elapsed / PEAK_TFLOPS_BF16, 4),
        "clips_per_min": round(60.0 * world * args.batch / (50.0 * elapsed / args.steps + (decode_ms or 0.0) / 1000.0), 4),
        "vae_decode_ms_per_clip": None if decode_ms is None else round(decode_ms, 2),
        "output_finite": finite,
    }
    if fams:
        total_ms = sum(f["ms"] for f in fams) or 1.0
        kernels = []
        for f in fams:
            if not f["launches"]:
                continue
            sec = f["ms"] / 1e3
            k = {"family": f["family"], "launches_per_step": f["launches"] / prof_steps,
                 "ms_per_step": round(f["ms"] / prof_steps, 3), "share_of_kernel_time": round(f["ms"] / total_ms, 4),
                 "avg_launch_us": round(1e3 * f["ms"] / f["launches"], 2),
                 "algorithmic_bytes_per_launch": round(f["bytes"] / f["launches"])}
            if f["family"] in MFMA_FAMS:
                k.update(bound="mfma", achieved=round(f["flops"] / sec / 1e12, 2), peak=PEAK_TFLOPS_BF16, unit="TFLOP/s")
            else:
                k.update(bound="hbm", achieved=round(f["bytes"] / sec / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s")
            k["frac"] = round(k["achieved"] / k["peak"], 4)
            kernels.append(k)
        kernels.sort(key=lambda k: -k["ms_per_step"])
        dom = kernels[0]
        out["roofline"] = {"kernel": dom["family"], "bound": dom["bound"], "achieved": dom["achieved"],
                           "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"], "traffic": None,
                           "avg_launch_us": dom["avg_launch_us"], "launches_per_step": dom["launches_per_step"],
                           "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"]}
        out["roofline"]["measured"] = ("hipEvents on the launch stream over the timed region" if not use_graph else
                                       f"hipEvents on the launch stream over {prof_steps} eager steps re-run right after "
                                       "the timed region (the timed region replays the same launches as a hipGraph)")
        out["roofline"].update(pmc_traffic(dom["family"], args, dom["launches_per_step"]))
        if dom["bound"] == "mfma":
            # next to the nominal peak, the rate this chip sustains on register-resident MFMAs with random
            # bf16 operands (tools/ubench/mfma_peak.hip, measured on the same pool: the chip clocks to its power budget)
            out["roofline"]["measured_mfma_peak"] = MEASURED_MFMA_PEAK_TFLOPS
            out["roofline"]["frac_of_measured_peak"] = round(dom["achieved"] / MEASURED_MFMA_PEAK_TFLOPS, 4)
        out["kernels"] = kernels
        out["kernel_time_ms_per_step"] = round(total_ms / prof_steps, 3)
    children = world == 1 and args.operand == "bf16" and not args.no_children and not os.environ.get("MUDG_ATTN_FP8")
    if children and not args.no_at_tolerance:
        out["at_tolerance"] = at_tolerance(args)      # child process; this one idles meanwhile (288 GB hold both replicas)
        out["value_at_tolerance"] = out["at_tolerance"].get("value")      # steps/s of the mode that meets the stated 1e-3 tolerance
    if children and not args.no_other_configs and args.resolution == "1024" and args.batch == 1:
        out["other_configs"] = other_configs(args)
    if children and not args.no_training and args.batch == 1:
        out["training_step"] = training_step_record(args)
    if world == 1 and not args.no_cpu_baseline and model is not None:
        try:
            out["cpu_baseline"] = cpu_baseline(model, wl["inp"], args.resolution, args.cpu_baseline)
        except Exception as e:          # the baseline is a report, never a reason to lose the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
