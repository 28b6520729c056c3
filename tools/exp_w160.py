#!/usr/bin/env python3
"""The 160-row tile kernel (csrc/wgemm.hip: w160_kernel) against the 128 x 128 kernels on the same inputs: parity of results and GroupNorm
partials on small problems of every mode / epilogue, then per-shape timings of the MDM512 shapes (BASELINE configs[1]) with either kernel.
    MUDG_DEBUG_VARIANTS=1 python tools/exp_w160.py [parity|time|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MUDG_DEBUG_VARIANTS", "1")
import torch
import torch.nn.functional as F
from mudg_amd import hip, ops
from tools.kernel_bench import timeit

assert hip.planes() == 1, "the 160-row tile exists in the 16-bit builds only"


def rn(*shape):
    return (torch.randn(*shape, device="cuda") * 0.5).to(ops.H16())


def rs(*shape):
    return (torch.randn(*shape, device="cuda") * 0.5).to(ops.STREAM())


what = sys.argv[1] if len(sys.argv) > 1 else "all"
fails = 0


def both(fn):
    """fn() on the 128 x 128 kernels, then on the 160-row tile (forced for every eligible problem)."""
    out = []
    for v in ("0", "2"):
        os.environ["MUDG_GEMM_W160"] = v
        os.environ["MUDG_GEMM_W288"] = "0"
        y = fn()
        torch.cuda.synchronize()
        out.append((y, getattr(y, ops.GN_ATTR, None), getattr(y, ops.GN_ATTR + "_rows", 128)))
    os.environ["MUDG_GEMM_W160"] = "1"
    os.environ["MUDG_GEMM_W288"] = "1"
    return out


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def report(name, res, samples_rows=None, exact=True):
    """exact: no residual — the bits of the 128 x 128 kernels; with a residual the tile adds it first (1e-3-class differences of a 16-bit result)."""
    global fails
    (y0, p0, r0), (y1, p1, r1) = res
    same = torch.equal(y0, y1)
    line = f"{name}: result rel-L2 {rel(y1, y0):.2e} bit-identical {same}"
    bad = (exact and not same) or rel(y1, y0) > 5e-3 or not bool(torch.isfinite(y1.double()).all())
    if p0 is not None:
        if not (r1 == 160 and r0 == 128):
            bad = True
            line += f" [partial blocks {r0} / {r1}: the tile did not run]"
        else:
            n = samples_rows
            s1 = p1.reshape(-1, n // 160, *p1.shape[1:]).double().sum(1)
            ref = torch.stack([y1.double().reshape(-1, n, y1.shape[1]).sum(1), (y1.double() ** 2).reshape(-1, n, y1.shape[1]).sum(1)], -1)
            e = rel(s1, ref)
            line += f"; partials vs sums of the stored result {e:.2e}"
            bad = bad or e > 1e-5
    if bad:
        fails += 1
        line += "   <-- FAIL"
    print(line, flush=True)


if what in ("parity", "all"):
    torch.manual_seed(0)
    for M, N, K in ((160 * 5, 640, 320), (160 * 3 + 100, 320, 1280), (160 * 9, 960, 64), (160 * 2, 320, 128), (160 * 7 + 16, 320, 192)):
        x, w = rn(M, K), rn(N, K)
        b = torch.randn(N, device="cuda")
        r = rs(M, N)
        report(f"gemm {M}x{N}x{K} bias+residual, stream out", both(lambda: ops.gemm(x, w, bias=b, residual=r, out_stream=True, frame_rows=160)), exact=False)
        report(f"gemm {M}x{N}x{K} fp32 out", both(lambda: ops.gemm(x, w, bias=b, out_fp32=True, frame_rows=160)))
        report(f"gemm {M}x{N}x{K} operand out, alpha", both(lambda: ops.gemm(x, w, bias=b, alpha=0.37, frame_rows=160)))
    M, N, K = 160 * 8, 320, 640
    x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
    r32 = torch.randn(M, N, device="cuda")
    report("gemm fp32 residual, operand out, stats", both(lambda: ops.gemm(x, w, bias=b, residual=r32, stats=True, frame_rows=320)), 320, exact=False)
    report("gemm stream out, stats", both(lambda: ops.gemm(x, w, bias=b, stats=True, out_stream=True, frame_rows=640)), 640)
    x2 = rn(M, 192)
    w2 = rn(N, K + 192)
    report("gemm two sources, stats", both(lambda: ops.gemm(x, w2, x2=x2, bias=b, stats=True, out_stream=True, frame_rows=160)), 160)
    for korder in (0, 1):
        f, h, wd, cin, cout = 3, 20, 32, 128, 320
        x, w, b = rn(f * h * wd, cin), rn(cout, 9 * cin), torch.randn(cout, device="cuda")
        emb = torch.randn(f, cout, device="cuda")
        r = rs(f * h * wd, cout)
        report(f"conv korder {korder} bias+gbias+stats", both(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b, gbias=emb,
                                                                                 rows_per_group=h * wd, stats=True)), h * wd)
        report(f"conv korder {korder} residual, stream out, stats", both(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b,
                                                                                            residual=r, out_stream=True, stats=True)), h * wd, exact=False)
        xa, xb = rn(f * h * wd, 64), rn(f * h * wd, 64)
        report(f"conv korder {korder} two sources fp32 out", both(lambda: ops.conv3x3(xa, w, x2=xb, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b,
                                                                                     out_fp32=True)))
    f, h, wd, cin, cout = 2, 10, 16, 64, 640                       # one tile per frame, every row at an image border or next to one
    x, w, b = rn(f * h * wd, cin), rn(cout, 9 * cin), torch.randn(cout, device="cuda")
    report("conv 10x16 frames (one tile each)", both(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=1, bias=b, stats=True)), h * wd)
    for M, N, K in ((160 * 4 + 40, 512, 320), (160 * 2, 2560, 128), (160 * 6, 1024, 64)):
        x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
        report(f"geglu {M}x{N}x{K}", both(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=160)))
        report(f"geglu {M}x{N}x{K} fp32 out", both(lambda: ops.gemm(x, w, bias=b, geglu=True, out_fp32=True, frame_rows=160)))
    clips, t, hw, c, co = 2, 4, 160, 128, 320
    x, w, b = rn(clips * t * hw, c), rn(co, 3 * c), torch.randn(co, device="cuda")
    r = rs(clips * t * hw, co)
    report("tconv bias+stats", both(lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, bias=b, stats=True)), t * hw)
    report("tconv residual, stream out", both(lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, bias=b, residual=r, out_stream=True)), exact=False)
    # against fp64 references
    f, h, wd, cin, cout = 2, 20, 24, 64, 320
    x, w = rn(f * h * wd, cin), rn(cout, 9 * cin)
    os.environ["MUDG_GEMM_W160"], os.environ["MUDG_GEMM_W288"] = "2", "0"
    y = ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=0, out_fp32=True)
    M, N, K = 160 * 11, 640, 1280
    xg, wg = rn(M, K), rn(N, K)
    yg = ops.gemm(xg, wg, out_fp32=True, frame_rows=160)
    os.environ["MUDG_GEMM_W160"], os.environ["MUDG_GEMM_W288"] = "1", "1"
    xi = x.double().reshape(f, h, wd, cin).permute(0, 3, 1, 2)
    wi = w.double().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xi, wi, padding=1).permute(0, 2, 3, 1).reshape(f * h * wd, cout)
    e1, e2 = rel(y, ref), rel(yg, xg.double() @ wg.double().t())
    print(f"conv vs fp64 conv2d: {e1:.2e}; gemm vs fp64 matmul: {e2:.2e}", flush=True)
    if e1 > 1e-5 or e2 > 1e-5:
        fails += 1
    # bit-reproducibility of repeated launches (the ring's counted waits: a race shows up as run-to-run differences)
    os.environ["MUDG_GEMM_W160"], os.environ["MUDG_GEMM_W288"] = "2", "0"
    M, N, K = 160 * 512, 320, 320
    xg, wg, bg, rg = rn(M, K), rn(N, K), torch.randn(N, device="cuda"), rs(M, N)
    y0 = ops.gemm(xg, wg, bias=bg, residual=rg, out_stream=True, frame_rows=2560).clone()
    same = all(torch.equal(y0, ops.gemm(xg, wg, bias=bg, residual=rg, out_stream=True, frame_rows=2560)) for _ in range(20))
    xc, wc = rn(32 * 40 * 64, 320), rn(320, 9 * 320)
    c0 = ops.conv3x3(xc, wc, frames=32, hin=40, win=64, cin=320, korder=1, stats=True).clone()
    same_c = all(torch.equal(c0, ops.conv3x3(xc, wc, frames=32, hin=40, win=64, cin=320, korder=1, stats=True)) for _ in range(20))
    os.environ["MUDG_GEMM_W160"], os.environ["MUDG_GEMM_W288"] = "1", "1"
    print(f"20 repeats bit-identical: gemm {same}, conv {same_c}", flush=True)
    if not (same and same_c):
        fails += 1
    # the deferred residual (the default) against the residual as seeds: the same values up to the order of one fp32 addition
    os.environ["MUDG_GEMM_W288"] = "0"
    for M, N, K in ((160 * 5 + 3, 640, 320), (160 * 3, 320, 64), (160 * 4, 320, 128), (160 * 6, 960, 1280)):
        x, w, b, r = rn(M, K), rn(N, K), torch.randn(N, device="cuda"), rs(M, N)
        ys = []
        for dv in ("1", "0"):
            os.environ["MUDG_GEMM_W160"], os.environ["MUDG_GEMM_W160DEFER"] = "2", dv
            ys.append(ops.gemm(x, w, bias=b, residual=r, out_stream=True, stats=True, frame_rows=160))
        e = rel(ys[0], ys[1])
        ok = e < 2e-3 and bool(torch.isfinite(ys[0].float()).all())
        fails += not ok
        print(f"gemm {M}x{N}x{K} residual deferred vs seeded: rel-L2 {e:.2e}" + ("" if ok else "   <-- FAIL"), flush=True)
    os.environ["MUDG_GEMM_W160"], os.environ["MUDG_GEMM_W288"] = "1", "1"
    os.environ.pop("MUDG_GEMM_W160DEFER", None)
    print(f"PARITY {'OK' if not fails else 'FAILED: %d' % fails}", flush=True)

if what in ("time", "all"):
    # MDM512: frames of 40 x 64 / 20 x 32 / 10 x 16 latent pixels, a guidance batch of one 16-frame clip = 32 frames
    def t2(fn):
        ts = []
        for v in ("0", "2"):
            os.environ["MUDG_GEMM_W160"] = v
            ts.append(timeit(fn, iters=10))
        os.environ["MUDG_GEMM_W160"] = "1"
        return ts
    G = [(81920, 320, 320, 2560), (81920, 320, 1280, 2560), (81920, 960, 320, 2560), (81920, 640, 320, 2560), (20480, 640, 640, 640), (20480, 640, 2560, 640),
         (20480, 1920, 640, 640), (20480, 1280, 640, 640), (5120, 1280, 5120, 160), (5120, 3840, 1280, 160), (5120, 1280, 1280, 160), (5120, 2560, 1280, 160)]
    for resid in (1, 0):
        for (M, N, K, hw) in G:
            x, w = rn(M, K), rn(N, K)
            b = torch.randn(N, device="cuda")
            r = rs(M, N) if resid else None
            ts = t2(lambda: ops.gemm(x, w, bias=b, residual=r, out_stream=bool(resid), frame_rows=hw))
            extra = ""
            if resid:               # the residual seeding the accumulators (GEMM_W160DEFER = 0) instead of deferred to the epilogue
                os.environ["MUDG_GEMM_W160"], os.environ["MUDG_GEMM_W160DEFER"] = "2", "0"
                tsd = timeit(lambda: ops.gemm(x, w, bias=b, residual=r, out_stream=True, frame_rows=hw), iters=10)
                os.environ["MUDG_GEMM_W160"] = "1"
                os.environ.pop("MUDG_GEMM_W160DEFER", None)
                extra = f" | residual as seeds {tsd*1e6:8.1f} us (deferral x{tsd/ts[1]:.3f})"
            print(f"gemm {M} {N} {K} residual={resid}: 128x128 {ts[0]*1e6:8.1f} us {2.0*M*N*K/ts[0]/1e12:7.1f} TF | 160x320 {ts[1]*1e6:8.1f} us {2.0*M*N*K/ts[1]/1e12:7.1f} TF"
                  f"  x{ts[0]/ts[1]:.3f}{extra}", flush=True)
    for (M, N, K, hw) in [(81920, 2560, 320, 2560), (20480, 5120, 640, 640), (5120, 10240, 1280, 160)]:
        x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
        ts = t2(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw))
        print(f"geglu {M} {N} {K}: 128x128 {ts[0]*1e6:8.1f} us {2.0*M*N*K/ts[0]/1e12:7.1f} TF | 160x256 {ts[1]*1e6:8.1f} us {2.0*M*N*K/ts[1]/1e12:7.1f} TF  x{ts[0]/ts[1]:.3f}", flush=True)
    for (clips, t, hw, c) in [(2, 16, 2560, 320), (2, 16, 640, 640), (2, 16, 160, 1280)]:
        x, w = rn(clips * t * hw, c), rn(c, 3 * c)
        M = clips * t * hw
        ts = []
        for v, ko in (("0", 1), ("0", 0), ("2", 0)):
            os.environ["MUDG_GEMM_W160"] = v
            ts.append(timeit(lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, stats=True, korder=ko), iters=10))
        os.environ["MUDG_GEMM_W160"] = "1"
        print(f"tconv {M} {c} {3*c}: 128x128 slab {ts[0]*1e6:8.1f} us, plain {ts[1]*1e6:8.1f} us | 160x320 {ts[2]*1e6:8.1f} us {2.0*M*c*3*c/ts[2]/1e12:7.1f} TF"
              f"  x{min(ts[0], ts[1])/ts[2]:.3f}", flush=True)
    C = [(32, 40, 64, 320, 320), (32, 40, 64, 640, 320), (32, 40, 64, 960, 320), (16, 40, 64, 320, 320), (32, 20, 32, 640, 640), (32, 20, 32, 1280, 640),
         (32, 20, 32, 1920, 640), (32, 10, 16, 1280, 1280), (32, 10, 16, 2560, 1280), (32, 10, 16, 1920, 1280)]
    for (f, h, w_, cin, cout) in C:
        x, w = rn(f * h * w_, cin), rn(cout, 9 * cin)
        M = f * h * w_
        ts = t2(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=w_, cin=cin, korder=1, stats=True))
        fl = 2.0 * M * cout * 9 * cin
        print(f"conv {M} {cout} {9*cin}: 128x128 {ts[0]*1e6:8.1f} us {fl/ts[0]/1e12:7.1f} TF | 160x320 {ts[1]*1e6:8.1f} us {fl/ts[1]/1e12:7.1f} TF  x{ts[0]/ts[1]:.3f}", flush=True)
sys.exit(1 if fails else 0)
