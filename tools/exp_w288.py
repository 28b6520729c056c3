#!/usr/bin/env python3
"""The 288 x 320-tile kernel (csrc/wgemm.hip) against the 128 x 128 kernels on the same inputs: parity of results and GroupNorm
partials on small problems of every mode / epilogue, then per-shape timings of the MDM1024 shapes with either kernel.
    MUDG_DEBUG_VARIANTS=1 [MUDG_OPERAND=bf16x3] python tools/exp_w288.py [parity|time|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MUDG_DEBUG_VARIANTS", "1")
import torch
import torch.nn.functional as F
from mudg_amd import hip, ops
from tools.kernel_bench import timeit


def rn(*shape):
    """A random operand matrix of the loaded build (MUDG_OPERAND=bf16x3: both pieces, through mudg_cast_rows)."""
    x = torch.randn(*shape, device="cuda") * 0.5
    return ops.cast_bf16(x) if hip.planes() > 1 else x.to(ops.H16())


def val(t):
    """The values an operand matrix stands for, in fp64 (bf16x3: the sum of its pieces)."""
    if hip.planes() > 1:
        c = t.shape[1]
        return sum(t._base[:, p * c:(p + 1) * c].double() for p in range(hip.planes()))
    return t.double()


def rs(*shape):
    """A random residual-stream matrix."""
    return (torch.randn(*shape, device="cuda") * 0.5).to(ops.STREAM())


what = sys.argv[1] if len(sys.argv) > 1 else "all"


def both(fn):
    out = []
    for v in ("0", "2"):
        os.environ["MUDG_GEMM_W288"] = v
        y = fn()
        torch.cuda.synchronize()
        out.append((y, getattr(y, ops.GN_ATTR, None), getattr(y, ops.GN_ATTR + "_rows", 128)))
    os.environ["MUDG_GEMM_W288"] = "1"
    return out


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def report(name, res, samples_rows=None):
    (y0, p0, r0), (y1, p1, r1) = res
    if y0.dtype == ops.H16():
        y0, y1 = val(y0), val(y1)
    line = f"{name}: result rel-L2 {rel(y1, y0):.2e}"
    if p0 is not None:
        assert r1 == 288 and r0 == 128, (r0, r1)
        n = samples_rows
        s0 = p0.reshape(-1, n // 128, *p0.shape[1:]).double().sum(1) if n % 128 == 0 else None
        s1 = p1.reshape(-1, n // 288, *p1.shape[1:]).double().sum(1)
        ref = torch.stack([y1.double().reshape(-1, n, y1.shape[1]).sum(1), (y1.double() ** 2).reshape(-1, n, y1.shape[1]).sum(1)], -1)
        line += f"; partials vs sums of the stored result {rel(s1, ref):.2e}" + (f", 128-row kernels' {rel(s0, ref):.2e}" if s0 is not None else "")
    print(line, flush=True)


if what in ("parity", "all"):
    torch.manual_seed(0)
    for M, N, K in ((288 * 5, 640, 320), (288 * 3 + 100, 320, 1280), (288 * 9, 960, 64)):
        x, w = rn(M, K), rn(N, K)
        b = torch.randn(N, device="cuda")
        r = rs(M, N)
        report(f"gemm {M}x{N}x{K} bias+residual, stream out", both(lambda: ops.gemm(x, w, bias=b, residual=r, out_stream=True, frame_rows=288)))
        report(f"gemm {M}x{N}x{K} fp32 out", both(lambda: ops.gemm(x, w, bias=b, out_fp32=True, frame_rows=288)))
    M, N, K = 288 * 8, 320, 640
    x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
    r32 = torch.randn(M, N, device="cuda")
    report("gemm fp32 residual, operand out, stats", both(lambda: ops.gemm(x, w, bias=b, residual=r32, stats=True, frame_rows=576)), 576)
    x2 = rn(M, 192)
    w2 = rn(N, K + 192)
    report("gemm two sources, stats", both(lambda: ops.gemm(x, w2, x2=x2, bias=b, stats=True, out_stream=True, frame_rows=288)), 288)
    for korder in (0, 1):
        f, h, wd, cin, cout = 3, 24, 36, 128, 320
        x, w, b = rn(f * h * wd, cin), rn(cout, 9 * cin), torch.randn(cout, device="cuda")
        emb = torch.randn(f, cout, device="cuda")
        r = rs(f * h * wd, cout)
        report(f"conv korder {korder} bias+gbias+stats", both(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b, gbias=emb,
                                                                                 rows_per_group=h * wd, stats=True)), h * wd)
        report(f"conv korder {korder} residual, stream out, stats", both(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b,
                                                                                            residual=r, out_stream=True, stats=True)), h * wd)
        xa, xb = rn(f * h * wd, 64), rn(f * h * wd, 64)
        report(f"conv korder {korder} two sources fp32 out", both(lambda: ops.conv3x3(xa, w, x2=xb, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b,
                                                                                     out_fp32=True)))
    for M, N, K in ((288 * 4 + 40, 512, 320), (288 * 2, 2560, 128)):
        x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
        report(f"geglu {M}x{N}x{K}", both(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=288)))
    clips, t, hw, c, co = 2, 4, 288, 128, 320
    x, w, b = rn(clips * t * hw, c), rn(co, 3 * c), torch.randn(co, device="cuda")
    r = rs(clips * t * hw, co)
    report("tconv bias+stats", both(lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, bias=b, stats=True)), t * hw)
    report("tconv residual, stream out", both(lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, bias=b, residual=r, out_stream=True)))
    # against fp64 references (the kernels' own operand rounding in both)
    f, h, wd, cin, cout = 2, 24, 24, 64, 320
    x, w = rn(f * h * wd, cin), rn(cout, 9 * cin)
    os.environ["MUDG_GEMM_W288"] = "2"
    y = ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=0, out_fp32=True)
    os.environ["MUDG_GEMM_W288"] = "1"
    xi = val(x).reshape(f, h, wd, cin).permute(0, 3, 1, 2)
    wi = val(w).reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xi, wi, padding=1).permute(0, 2, 3, 1).reshape(f * h * wd, cout)
    print(f"conv vs fp64 conv2d: {rel(y, ref):.2e}", flush=True)

if what in ("pparity", "all") and hip.planes() == 1:
    # the persistent form (more tiles than CUs) against the one-tile form of the same kernel and against the 128 x 128 kernels: the same bits
    torch.manual_seed(1)
    def three(fn):
        outs = []
        for w288, pers in (("0", "1"), ("2", "0"), ("2", "2")):
            os.environ["MUDG_GEMM_W288"], os.environ["MUDG_GEMM_W288P"] = w288, pers
            y = fn()
            torch.cuda.synchronize()
            outs.append((y, getattr(y, ops.GN_ATTR, None)))
        os.environ["MUDG_GEMM_W288"] = "1"
        os.environ.pop("MUDG_GEMM_W288P", None)
        return outs
    for name, fn in (
        ("gemm 201600x320x320 bias+residual stream", lambda: ops.gemm(xs[0], ws[0], bias=bs[0], residual=rs_[0], out_stream=True, frame_rows=288)),
        ("gemm 201600x320x320 fp32 out, stats", lambda: ops.gemm(xs[0], ws[0], bias=bs[0], out_fp32=True, stats=True, frame_rows=288)),
        ("gemm 86400x960x640 operand out", lambda: ops.gemm(xs[1], ws[1], frame_rows=288)),
        ("gemm 86400x640x128 (two K-tiles) fp32 residual", lambda: ops.gemm(xs[2], ws[2], bias=bs[2], residual=r32, frame_rows=288)),
        ("geglu 86400x2560x320", lambda: ops.gemm(xs[3], ws[3], bias=bs[3], geglu=True, frame_rows=288)),
        ("geglu 86400x2560x320 fp32 out", lambda: ops.gemm(xs[3], ws[3], bias=bs[3], geglu=True, out_fp32=True, frame_rows=288)),
    ):
        if name.startswith("gemm 201600x320x320 bias"):
            xs = [rn(288 * 700, 320), rn(288 * 300, 640), rn(288 * 300, 128), rn(288 * 300, 320)]
            ws = [rn(320, 320), rn(960, 640), rn(640, 128), rn(2560, 320)]
            bs = [torch.randn(320, device="cuda"), None, torch.randn(640, device="cuda"), torch.randn(2560, device="cuda")]
            rs_ = [rs(288 * 700, 320)]
            r32 = torch.randn(288 * 300, 640, device="cuda")
        (y0, p0), (y1, p1), (y2, p2) = three(fn)
        same = torch.equal(y1, y2) and (p1 is None or torch.equal(p1, p2))
        print(f"{name}: persistent == one-tile 288 x 320: {same}; == 128 x 128 kernels: {torch.equal(y0, y2)} (rel-L2 {rel(y2, y0):.1e}; a residual seeds the tile's accumulators)", flush=True)

if what in ("time", "all"):
    G = [(294912, 320, 320, 9216), (294912, 320, 1280, 9216), (294912, 960, 320, 9216), (294912, 640, 320, 9216), (73728, 640, 640, 2304), (73728, 640, 2560, 2304),
         (73728, 1920, 640, 2304), (73728, 1280, 640, 2304), (18432, 1280, 5120, 576), (18432, 3840, 1280, 576), (18432, 1280, 1280, 576), (18432, 2560, 1280, 576)]
    for resid in (1, 0):
        for (M, N, K, hw) in G:
            x, w = rn(M, K), rn(N, K)
            b = torch.randn(N, device="cuda")
            r = rs(M, N) if resid else None
            ts = []
            for v in ("0", "2"):
                os.environ["MUDG_GEMM_W288"] = v
                ts.append(timeit(lambda: ops.gemm(x, w, bias=b, residual=r, out_stream=bool(resid), frame_rows=hw), iters=10))
            print(f"gemm {M} {N} {K} residual={resid}: 128x128 {ts[0]*1e6:8.1f} us {2.0*M*N*K/ts[0]/1e12:7.1f} TF | 288x320 {ts[1]*1e6:8.1f} us {2.0*M*N*K/ts[1]/1e12:7.1f} TF"
                  f"  x{ts[0]/ts[1]:.3f}", flush=True)
    for (M, N, K, hw) in [(294912, 2560, 320, 9216), (73728, 5120, 640, 2304), (18432, 10240, 1280, 576), (147456, 2560, 320, 9216)]:
        x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
        ts = []
        for v in ("0", "2"):
            os.environ["MUDG_GEMM_W288"] = v
            ts.append(timeit(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw), iters=10))
        print(f"geglu {M} {N} {K}: 128x128 {ts[0]*1e6:8.1f} us {2.0*M*N*K/ts[0]/1e12:7.1f} TF | 288x256 {ts[1]*1e6:8.1f} us {2.0*M*N*K/ts[1]/1e12:7.1f} TF  x{ts[0]/ts[1]:.3f}", flush=True)
    T = [(2, 16, 9216, 320), (2, 16, 2304, 640), (2, 16, 576, 1280)]
    for (clips, t, hw, c) in T:
        x, w = rn(clips * t * hw, c), rn(c, 3 * c)
        M = clips * t * hw
        ts = []
        for v, ko in (("0", 1), ("0", 0), ("2", 0)):
            os.environ["MUDG_GEMM_W288"] = v
            ts.append(timeit(lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, stats=True, korder=ko), iters=10))
        print(f"tconv {M} {c} {3*c}: 128x128 slab {ts[0]*1e6:8.1f} us, plain {ts[1]*1e6:8.1f} us | 288x320 {ts[2]*1e6:8.1f} us {2.0*M*c*3*c/ts[2]/1e12:7.1f} TF"
              f"  x{min(ts[0], ts[1])/ts[2]:.3f}", flush=True)
    C = [(32, 72, 128, 320, 320), (32, 72, 128, 640, 320), (32, 72, 128, 960, 320), (16, 72, 128, 320, 320), (32, 36, 64, 640, 640), (32, 36, 64, 1280, 640),
         (32, 36, 64, 1920, 640), (32, 18, 32, 1280, 1280), (32, 18, 32, 2560, 1280), (32, 18, 32, 1920, 1280)]
    for (f, h, w_, cin, cout) in C:
        x, w = rn(f * h * w_, cin), rn(cout, 9 * cin)
        M = f * h * w_
        ts = []
        for v in ("0", "2"):
            os.environ["MUDG_GEMM_W288"] = v
            ts.append(timeit(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=w_, cin=cin, korder=1, stats=True), iters=10))
        fl = 2.0 * M * cout * 9 * cin
        print(f"conv {M} {cout} {9*cin}: 128x128 {ts[0]*1e6:8.1f} us {fl/ts[0]/1e12:7.1f} TF | 288x320 {ts[1]*1e6:8.1f} us {fl/ts[1]/1e12:7.1f} TF  x{ts[0]/ts[1]:.3f}", flush=True)
    os.environ["MUDG_GEMM_W288"] = "1"

if what in ("q", "qparity", "qtime") and hip.planes() == 1:
    # Round 6: the 288-row tile on the loop of the 160-row tile (wq_kernel<..., 9>: one barrier per k half, fragments refreshed in place
    # between the MFMAs, branch-free steady state) against wgemm_kernel's six-phase loop — MUDG_GEMM_W288Q = 0 / 2, both under MUDG_GEMM_W288 = 2.
    os.environ["MUDG_GEMM_W288"] = "2"

    def qboth(fn):
        out = []
        for v in ("0", "2"):
            os.environ["MUDG_GEMM_W288Q"] = v
            y = fn()
            torch.cuda.synchronize()
            out.append((y, getattr(y, ops.GN_ATTR, None)))
        os.environ["MUDG_GEMM_W288Q"] = "0"
        return out

    bad = 0
    if what in ("q", "qparity"):
        torch.manual_seed(2)
        cases = []
        for M, N, K in ((288 * 5, 640, 320), (288 * 3 + 100, 320, 1280), (288 * 9, 960, 64), (288 * 2 + 1, 320, 128), (288 * 4, 320, 192), (288 * 300, 320, 320)):
            x, w, b, r = rn(M, K), rn(N, K), torch.randn(N, device="cuda"), rs(M, N)
            cases.append((f"gemm {M}x{N}x{K} residual stream stats", lambda x=x, w=w, b=b, r=r: ops.gemm(x, w, bias=b, residual=r, out_stream=True, stats=True, frame_rows=288)))
            cases.append((f"gemm {M}x{N}x{K} fp32", lambda x=x, w=w, b=b: ops.gemm(x, w, bias=b, out_fp32=True, frame_rows=288)))
        for korder in (0, 1):
            f, h, wd, cin, cout = 3, 24, 36, 128, 320
            x, w, b = rn(f * h * wd, cin), rn(cout, 9 * cin), torch.randn(cout, device="cuda")
            emb, r = torch.randn(f, cout, device="cuda"), rs(f * h * wd, cout)
            xa, xb = rn(f * h * wd, 64), rn(f * h * wd, 64)
            cases.append((f"conv korder {korder} gbias stats", lambda x=x, w=w, b=b, emb=emb, korder=korder: ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b, gbias=emb, rows_per_group=h * wd, stats=True)))
            cases.append((f"conv korder {korder} residual", lambda x=x, w=w, b=b, r=r, korder=korder: ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b, residual=r, out_stream=True, stats=True)))
            cases.append((f"conv korder {korder} two sources", lambda xa=xa, xb=xb, w=w, b=b, korder=korder: ops.conv3x3(xa, w, x2=xb, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b, out_fp32=True)))
        for M, N, K in ((288 * 4 + 40, 512, 320), (288 * 2, 2560, 128), (288 * 6, 1024, 64)):
            x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
            cases.append((f"geglu {M}x{N}x{K}", lambda x=x, w=w, b=b: ops.gemm(x, w, bias=b, geglu=True, frame_rows=288)))
        clips, t, hw, c, co = 2, 4, 288, 128, 320
        x, w, b, r = rn(clips * t * hw, c), rn(co, 3 * c), torch.randn(co, device="cuda"), rs(clips * t * hw, co)
        cases.append(("tconv stats", lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, bias=b, stats=True)))
        cases.append(("tconv residual", lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, bias=b, residual=r, out_stream=True)))
        for name, fn in cases:
            (y0, p0), (y1, p1) = qboth(fn)
            (y2, p2) = qboth(fn)[1]
            same = torch.equal(y0, y1) and (p0 is None or torch.equal(p0, p1)) and torch.equal(y1, y2) and bool(torch.isfinite(y1.float()).all())
            bad += not same
            print(f"{name}: new loop == six-phase loop (bits, partials, repeat): {same}" + ("" if same else f"   <-- FAIL rel-L2 {rel(y1, y0):.2e}"), flush=True)
        print(f"QPARITY {'OK' if not bad else 'FAILED: %d' % bad}", flush=True)
    if what in ("q", "qtime"):
        def tq(fn, pers=None):
            ts = []
            for v in ("0", "2"):
                os.environ["MUDG_GEMM_W288Q"] = v
                ts.append(timeit(fn, iters=10))
            os.environ["MUDG_GEMM_W288Q"] = "0"
            return ts
        G = [(294912, 320, 320, 9216), (294912, 320, 1280, 9216), (294912, 960, 320, 9216), (294912, 640, 320, 9216), (73728, 640, 640, 2304), (73728, 640, 2560, 2304),
             (73728, 1920, 640, 2304), (73728, 1280, 640, 2304), (18432, 1280, 5120, 576), (18432, 3840, 1280, 576), (18432, 1280, 1280, 576), (18432, 2560, 1280, 576)]
        for resid in (1, 0):
            for (M, N, K, hw) in G:
                x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
                r = rs(M, N) if resid else None
                ts = tq(lambda: ops.gemm(x, w, bias=b, residual=r, out_stream=bool(resid), frame_rows=hw))
                print(f"gemm {M} {N} {K} residual={resid}: six-phase {ts[0]*1e6:8.1f} us {2.0*M*N*K/ts[0]/1e12:7.1f} TF | new loop {ts[1]*1e6:8.1f} us {2.0*M*N*K/ts[1]/1e12:7.1f} TF  x{ts[0]/ts[1]:.3f}", flush=True)
        for (M, N, K, hw) in [(294912, 2560, 320, 9216), (73728, 5120, 640, 2304), (18432, 10240, 1280, 576), (147456, 2560, 320, 9216)]:
            x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
            ts = tq(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw))
            print(f"geglu {M} {N} {K}: six-phase (persistent where it applies) {ts[0]*1e6:8.1f} us {2.0*M*N*K/ts[0]/1e12:7.1f} TF | new loop, one tile {ts[1]*1e6:8.1f} us {2.0*M*N*K/ts[1]/1e12:7.1f} TF  x{ts[0]/ts[1]:.3f}", flush=True)
        for (clips, t, hw, c) in [(2, 16, 9216, 320), (2, 16, 2304, 640), (2, 16, 576, 1280)]:
            x, w = rn(clips * t * hw, c), rn(c, 3 * c)
            M = clips * t * hw
            ts = tq(lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, stats=True, korder=0))
            print(f"tconv {M} {c} {3*c}: six-phase {ts[0]*1e6:8.1f} us | new loop {ts[1]*1e6:8.1f} us {2.0*M*c*3*c/ts[1]/1e12:7.1f} TF  x{ts[0]/ts[1]:.3f}", flush=True)
        C = [(32, 72, 128, 320, 320), (32, 72, 128, 640, 320), (32, 72, 128, 960, 320), (16, 72, 128, 320, 320), (32, 36, 64, 640, 640), (32, 36, 64, 1280, 640),
             (32, 36, 64, 1920, 640), (32, 18, 32, 1280, 1280), (32, 18, 32, 2560, 1280), (32, 18, 32, 1920, 1280)]
        for (f, h, w_, cin, cout) in C:
            x, w = rn(f * h * w_, cin), rn(cout, 9 * cin)
            M = f * h * w_
            ts = tq(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=w_, cin=cin, korder=1, stats=True))
            fl = 2.0 * M * cout * 9 * cin
            print(f"conv {M} {cout} {9*cin}: six-phase {ts[0]*1e6:8.1f} us {fl/ts[0]/1e12:7.1f} TF | new loop {ts[1]*1e6:8.1f} us {fl/ts[1]/1e12:7.1f} TF  x{ts[0]/ts[1]:.3f}", flush=True)
    os.environ["MUDG_GEMM_W288"] = "1"
    sys.exit(1 if bad else 0)
