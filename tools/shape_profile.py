#!/usr/bin/env python3
"""Per-shape time accounting of one eager DDIM step at MDM1024 (cond+uncond batched): wraps the ops-level launchers
with stream events and aggregates by (op, M, N, K, epilogue flags).  Run on the GPU box:
    python tools/shape_profile.py [resolution] > gpurun_out/shapes.md"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mudg_amd import factory, hip, ops

records = []


def wrap(name, keyfn, flopfn):
    orig = getattr(ops, name)

    def f(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **kw)
        e1.record()
        records.append((name, keyfn(r, *a, **kw), flopfn(r, *a, **kw), e0, e1))
        return r

    setattr(ops, name, f)


def gemm_key(r, x, w, **kw):
    M = kw.get("M") or x.shape[0]
    N = kw.get("N") or w.shape[0]
    K = kw.get("K") or w.shape[1]
    fl = []
    if kw.get("batch", 1) > 1:
        fl.append(f"batch{kw['batch']}")
    for k in ("bias", "gbias", "residual", "x2"):
        if kw.get(k) is not None:
            fl.append(k + ("32" if k == "residual" and kw[k].dtype == torch.float32 else ""))
    for k in ("geglu", "gelu", "out_fp32"):
        if kw.get(k):
            fl.append(k)
    if kw.get("out") is not None and kw["out"].dtype == torch.float32 and "out_fp32" not in fl:
        fl.append("out_fp32")
    return (M, N, K, " ".join(fl))


def gemm_flops(r, x, w, **kw):
    M, N, K, _ = gemm_key(r, x, w, **kw)
    return 2.0 * M * N * K * kw.get("batch", 1)


def conv_key(r, x, w, **kw):
    M, N = r.shape[0], w.shape[0]
    fl = [f"{kw['hin']}x{kw['win']}", f"s{kw.get('stride', 1)}"]
    if kw.get("upsample"):
        fl.append("up2")
    for k in ("bias", "gbias", "residual", "x2"):
        if kw.get(k) is not None:
            fl.append(k)
    if r.dtype == torch.float32:
        fl.append("out_fp32")
    return (M, N, 9 * kw["cin"], " ".join(fl))


def tconv_key(r, x, w, **kw):
    fl = []
    for k in ("bias", "residual"):
        if kw.get(k) is not None:
            fl.append(k)
    if r.dtype == torch.float32:
        fl.append("out_fp32")
    return (r.shape[0], w.shape[0], 3 * kw["cin"], " ".join(fl))


def mnk_flops(r, x, w, **kw):
    return 2.0 * r.shape[0] * w.shape[0] * w.shape[1]


def attn_key(r, q, k, vt, out, **kw):
    return (kw["frames"] * kw["nq"], kw["nk"], kw["heads"], "acc" if kw.get("accumulate") else "")


def attn_flops(r, q, k, vt, out, **kw):
    return 4.0 * kw["frames"] * kw["heads"] * kw["nq"] * kw["nk"] * 64


def gn_key(r, x, *a, **kw):
    return (x.shape[0], x.shape[1], 0, str(x.dtype).replace("torch.", "") + (" x2" if kw.get("x2") is not None else ""))


def zero(*a, **kw):
    return 0.0


def main():
    res = sys.argv[1] if len(sys.argv) > 1 else "1024"
    dev = torch.device("cuda")
    hip.lib()
    model = factory.build_synthetic_model(res, dev, seed=123)
    inp = factory.synthetic_inputs(model, res, 1, dev, seed=5)
    from lvdm.models.samplers.ddim import DDIMSampler
    sampler = DDIMSampler(model)
    sampler.make_schedule(50, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    kw = dict(unconditional_guidance_scale=7.5, unconditional_conditioning=inp["uc"], guidance_rescale=0.7,
              fs=inp["fs"], sparse_x=inp["sparse_x"], class_label=inp["class_label"], cfg_img=None,
              unconditional_conditioning_img_nonetext=None)

    def step(x, index):
        ts = torch.full((1,), int(sampler.ddim_timesteps[index]), device=dev, dtype=torch.long)
        return sampler.p_sample_ddim(x, inp["cond"], ts, index=index, **kw)[0]

    x = inp["x_T"]
    for i in range(2):
        x = step(x, 49 - i)
    torch.cuda.synchronize()
    wrap("gemm", gemm_key, gemm_flops)
    wrap("conv3x3", conv_key, mnk_flops)
    wrap("tconv3", tconv_key, mnk_flops)
    wrap("attention", attn_key, attn_flops)
    wrap("groupnorm", gn_key, zero)
    wrap("layernorm", gn_key, zero)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    x = step(x, 47)
    b.record()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, key, fl, e0, e1 in records:
        d = agg.setdefault((name,) + key, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += fl
    tot = sum(d[1] for d in agg.values())
    print(f"# per-shape profile of one eager step, MDM{res}: {a.elapsed_time(b):.1f} ms wall, {tot:.1f} ms in wrapped ops\n")
    print("| op | M | N | K | flags | calls | total ms | avg us | TFLOP/s |")
    print("|---|---:|---:|---:|---|---:|---:|---:|---:|")
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tf = d[2] / (d[1] * 1e-3) / 1e12 if d[1] > 0 else 0
        print(f"| {k[0]} | {k[1]} | {k[2]} | {k[3]} | {k[4]} | {d[0]} | {d[1]:.2f} | {1e3 * d[1] / d[0]:.1f} | {tf:.0f} |")


if __name__ == "__main__":
    main()
