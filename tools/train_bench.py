#!/usr/bin/env python3
"""One training step (p_losses -> backward -> AdamW) of the full 1.44 B-parameter MDM UNet on ONE MI355X: seconds per step, peak
memory.  `python tools/train_bench.py [512|1024] [steps] [ckpt] [stage2] [b4] [acc2] [json]` — `ckpt` turns activation checkpointing on
(use_checkpoint); `stage2` applies what the reference's stage-2 training config adds to a step (configs/stage2-1024_mdm_waymo/
config.yaml): the stages' temporal transformers frozen (temporal_frozen), the gradient 2-norm clipped to 0.5; `b4` = the
reference's per-GPU batch of 4 clips (config.yaml:113-135), `acc2` = its accumulate_grad_batches 2 (two micro-batches per
optimiser step).  All parameters trainable and no clipping otherwise (the heavier step).  FLOP accounting: 3 x the forward per
clip, whatever is frozen.  After the timed steps one more step runs with the contraction families bracketed by hipEvents
(forward GEMMs / convs / attention and the input-gradient GEMMs that run on the same kernels): the `roofline` of the json."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import configs, factory

res = sys.argv[1] if len(sys.argv) > 1 else "512"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ckpt = "ckpt" in sys.argv[3:]
stage2 = "stage2" in sys.argv[3:]
B = 4 if "b4" in sys.argv[3:] else 1
ACC = 2 if "acc2" in sys.argv[3:] else 1
dev = torch.device("cuda:0")
model = factory.build_synthetic_model(res, dev, seed=123).train()
unet = model.model.diffusion_model
for m in unet.modules():
    if hasattr(m, "use_checkpoint"):
        m.use_checkpoint = ckpt
    if isinstance(getattr(m, "checkpoint", None), bool):
        m.checkpoint = ckpt
if stage2:
    for m in unet.modules():
        if type(m).__name__ == "TemporalTransformer" and m is not unet.init_attn[0]:
            m._frozen_model()
inp = factory.synthetic_inputs(model, res, B, dev, seed=123)
model.learning_rate = 1e-5
opt = model.configure_optimizers()
from mudg_amd.train import step
clip = step.GradientClipper([p for g in opt.param_groups for p in g["params"]], 0.5) if stage2 else None
batch = dict(x_start=inp["x_T"], cond=inp["cond"], t=torch.tensor([500, 120, 870, 333][:B], device=dev), class_label=inp["class_label"], fs=inp["fs"])
torch.cuda.reset_peak_memory_stats()
times, losses = [], []
def one_step():
    opt.zero_grad(set_to_none=False)         # multi-tensor fill; gradient tensors (and the pointer tables built on them) persist
    for _ in range(ACC):
        loss = model.training_step(batch)
        (loss / ACC).backward()
    norm = clip() if clip is not None else None
    opt.step()
    return loss, norm


for i in range(steps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, norm = one_step()
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    losses.append(float(loss.detach()))
    print(f"step {i}: loss {losses[-1]:.5f}  {times[-1]:.2f} s" + (f"  grad norm {float(norm[0]):.3f}" if norm is not None else ""), flush=True)
from mudg_amd import hip
hip.prof_reset(); hip.prof_enable((1 << len(hip.FAM_NAMES)) - 1)
one_step()
torch.cuda.synchronize()
fams = [hip.prof_collect(i) for i in range(len(hip.FAM_NAMES))]
hip.prof_enable(0)
mf = [f for f in fams if f["family"] in ("gemm", "conv3x3", "tconv3", "attention") and f["launches"]]
dom = max(mf, key=lambda f: f["ms"]) if mf else None
roof = None if dom is None else {"kernel": dom["family"], "bound": "mfma", "achieved": round(dom["flops"] / (dom["ms"] / 1e3) / 1e12, 1), "peak": 2500.0,
                                 "unit": "TFLOP/s", "frac": round(dom["flops"] / (dom["ms"] / 1e3) / 1e12 / 2500.0, 4), "ms_per_step": round(dom["ms"], 2),
                                 "launches_per_step": dom["launches"], "measured": "hipEvents on the launch stream over one more step; the family's "
                                 "forward and input-gradient launches (weight gradients and the attention backward have kernels of their own)"}
fl = 3 * configs.UNET_TFLOP[res] * B * ACC
best = min(times[1:])
if "json" in sys.argv[3:]:
    import json
    print(json.dumps({"workload": f"MDM{res} training step: p_losses -> backward -> AdamW, full 1.44 B-parameter UNet, B = {B}, 16 frames"
                                  + (f", {ACC} micro-batches per optimiser step" if ACC > 1 else ""),
                      "s_per_step": round(best, 4), "steps": steps, "tflops_per_s": round(fl / best, 1), "tflop_per_step": round(fl, 1),
                      "flop_accounting": "3 x the forward per clip", "batch": B, "accumulate": ACC, "roofline": roof, "checkpointing": ckpt, "stage2_settings": stage2,
                      "peak_memory_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1), "loss_first_last": [round(losses[0], 5), round(losses[-1], 5)]}))
    sys.exit(0)
print(f"MDM{res} training step (B = {B}" + (f" x {ACC} micro-batches" if ACC > 1 else "") + f", 16 frames, checkpointing {'on' if ckpt else 'off'}{', stage-2 settings' if stage2 else ''}): {best:.2f} s = {fl / best:.1f} TFLOP/s of the "
      f"{fl:.1f} TFLOP a forward + backward costs (3 x forward); peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
