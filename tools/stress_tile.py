#!/usr/bin/env python3
"""Randomised stress of the 288 x 320 tile and (--tile 160, round 6) of the 160 x 320 tile (csrc/wgemm.hip: hand-counted s_waitcnt vmcnt rings, inline-asm MFMAs, buffer descriptors
based before the tile's first row) — the tool the round-5 review asked for after an unexplained abort in 2 of ~100 suite runs.

Every case is a random problem the tile can run (plain GEMM / GEGLU / same-size 3x3 conv / temporal 3-tap conv; ragged M, K = 64 ...
1536 in odd and even numbers of K-tiles, one or two activation sources, every residual / result storage kind, group bias, GroupNorm
partials, ragged row strides), run

  * once on the 128 x 128 kernels   (MUDG_GEMM_W288=0: the reference bits),
  * REPEAT times on the tile        (MUDG_GEMM_W288=2: forced for every eligible problem), each time into a FRESH canary-filled buffer,
  * and, where the persistent form can run it, REPEAT times on that (MUDG_GEMM_W288P=2),

with the LDS of every CU overwritten by NaN patterns between launches (a fragment read that overtakes its DMA then multiplies NaNs
instead of the previous launch's identical, correct bytes), and checked:

  1. every tile run is BIT-IDENTICAL to every other tile run of the case (a race shows as a run-to-run difference),
  2. without a residual the tile's bits equal the 128 x 128 kernels' in the 16-bit builds (with one: rel-L2 within the storage rounding —
     the residual seeds the accumulators there; bf16x3: fp32-rounding apart, the three piece products are summed in another order),
  3. the canary rows before / after the result and the canary columns between N and the row stride are untouched,
  4. GroupNorm partials equal the column sums of the stored result.

    MUDG_DEBUG_VARIANTS=1 [MUDG_OPERAND=bf16x3] python tools/stress_tile.py [--launches 2000] [--seed 0] [--repeat 3] [--no-poison]
tools/stress_tile.sh runs it in both builds under the allocator / serialisation variants and keeps the logs."""
import argparse
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MUDG_DEBUG_VARIANTS", "1")
import torch  # noqa: E402

from mudg_amd import hip, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=2000, help="tile launches to reach (reference launches are not counted)")
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--no-poison", action="store_true")
ap.add_argument("--tile", type=int, default=288, choices=(288, 160), help="288: wgemm_kernel / wgemm_pkernel; 160: w160_kernel (16-bit builds)")
args = ap.parse_args()

CANARY = 0x5A5B
TILE = args.tile
SWITCH = "MUDG_GEMM_W288" if TILE == 288 else "MUDG_GEMM_W160"      # the variant switch that forces the stressed tile
G = 16                                  # canary rows before and after every result


def build_poison():
    """tools/ubench/lds_poison.hip -> a tiny library with one entry point (hipcc is on the GPU box: same image)."""
    if args.no_poison:
        return None
    src = os.path.join(ROOT, "tools", "ubench", "lds_poison.hip")
    out = os.path.join(ROOT, "gpurun_out", "lds_poison.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    try:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", out], check=True, capture_output=True, timeout=300)
        lib = ctypes.CDLL(out)
        lib.lds_poison.restype = ctypes.c_int
        lib.lds_poison.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
        return lib
    except Exception as e:                  # the stress still runs, without the poison
        print(f"[stress] LDS poison unavailable: {e}", flush=True)
        return None


POISON = build_poison()

# Which kernel ran a launch: the library's own answer for the descriptor ops.* built (mudg_gemm_stats_rows = 288 <=> the tile runs it).
LIB = hip.lib()
_real_gemm = LIB.mudg_gemm
LAST = {"rows": 0}


def _spy(dref, stream):
    LAST["rows"] = LIB.mudg_gemm_stats_rows(dref)
    return _real_gemm(dref, stream)


LIB.mudg_gemm = _spy
FAILURES = []


def fail(msg):
    FAILURES.append(msg)
    print("[stress] FAIL " + msg, flush=True)


def poison(i):
    if POISON is not None:
        rc = POISON.lds_poison(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 1024, 0x7FC07FC0 if i % 2 else 0xFFC1FFC1)
        assert rc == 0, rc


def rn(g, rows, cols, scale=0.5):
    x = torch.randn(rows, cols, generator=g, device="cuda") * scale
    return ops.cast_bf16(x) if hip.planes() > 1 else x.to(ops.H16())


def guarded(rows, cols, dtype, pad_cols):
    """(whole buffer, [rows, cols] view G rows in) — canary everywhere; operand matrices of the split builds keep their plane layout."""
    planes = hip.planes() if dtype == ops.H16() else 1
    width = planes * (cols + pad_cols)
    whole = torch.empty((rows + 2 * G, width), dtype=dtype, device="cuda")
    whole.view(torch.int16).fill_(CANARY)
    return whole, whole[G:G + rows, :cols]


def canaries_intact(whole, rows, cols):
    raw = whole.view(torch.int16)
    per = raw.shape[1] // whole.shape[1]               # int16 words per element
    planes = hip.planes() if whole.dtype == ops.H16() else 1
    pw = raw.shape[1] // planes                        # words per plane
    ok = bool((raw[:G] == CANARY).all()) and bool((raw[G + rows:] == CANARY).all())
    for p in range(planes):
        ok = ok and bool((raw[G:G + rows, p * pw + cols * per:(p + 1) * pw] == CANARY).all())
    return ok


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def values(t):
    if t.dtype == ops.H16() and hip.planes() > 1:
        c, base = t.shape[1], t._base
        pw = base.shape[1] // hip.planes()
        return sum(base[G:G + t.shape[0], p * pw:p * pw + c].double() for p in range(hip.planes()))
    return t.double()


def pick(rng, seq):
    return seq[int(torch.randint(len(seq), (1,), generator=rng))]


def ri(rng, lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=rng))


def make_case(rng, g):
    """-> (description, run(out) -> out, M, nout, out dtype, has_residual, stats sample rows or 0, persistent-eligible)"""
    kind = pick(rng, ["gemm"] * 5 + ["geglu"] * 2 + ["conv"] * 2 + ["tconv"] * 2)
    big = ri(rng, 0, 9) >= 7                             # three in ten cases fill more than one round of the 256 CUs
    out_kind = pick(rng, ["operand", "stream", "fp32"])
    odt = {"operand": ops.H16(), "stream": ops.STREAM(), "fp32": torch.float32}[out_kind]
    res_kind = pick(rng, [None, None, "stream", "fp32", "operand"])
    bias_on = ri(rng, 0, 3) > 0
    if kind in ("gemm", "geglu"):
        geglu = kind == "geglu"
        N = 256 * ri(rng, 1, 6) if geglu else 320 * ri(rng, 1, 4)
        K = 64 * ri(rng, 1, 24 if not big else 8)
        tiles_m = ri(rng, 60, 300) if big else ri(rng, 1, 12)
        whole = ri(rng, 0, 2) > 0
        M = TILE * tiles_m - (0 if whole else ri(rng, 1, TILE - 1))
        two = (not geglu) and K >= 128 and ri(rng, 0, 3) == 0
        c1 = 64 * ri(rng, 1, K // 64 - 1) if two else K
        x = rn(g, M, c1)
        x2 = rn(g, M, K - c1) if two else None
        w = rn(g, N, K, 0.1)
        b = torch.randn(N, generator=g, device="cuda") if bias_on else None
        nout = N // 2 if geglu else N
        if geglu:
            res_kind = None
            if out_kind == "stream":
                odt = ops.H16()
        r = None
        if res_kind:
            rdt = {"operand": ops.H16(), "stream": ops.STREAM(), "fp32": torch.float32}[res_kind]
            r = rn(g, M, nout) if rdt == ops.H16() else (torch.randn(M, nout, generator=g, device="cuda") * 0.5).to(rdt)
        gb, rpg, stats = None, 0, 0
        if not geglu and whole and ri(rng, 0, 2) == 0:
            per = pick(rng, [d for d in (1, 2, 3, 4, 6) if tiles_m % d == 0])
            rpg = TILE * per
            gb = torch.randn(M // rpg, N, generator=g, device="cuda")
        if not geglu and ri(rng, 0, 1):
            stats = TILE * pick(rng, [d for d in (1, 2, 3, 4) if tiles_m % d == 0]) if whole else 0
        desc = f"{kind} M={M} N={N} K={K}" + (f" split {c1}" if two else "") + f" out={out_kind} res={res_kind} bias={int(bias_on)} gbias={rpg} stats={stats}"

        def run(out):
            return ops.gemm(x, w, out=out, x2=x2, bias=b, gbias=gb, rows_per_group=rpg, residual=r, geglu=geglu, stats=bool(stats), frame_rows=TILE)
        pers = TILE == 288 and hip.planes() == 1 and whole and K >= 128 and r is None
        return desc, run, M, nout, odt, r is not None, stats, pers
    if kind == "conv":
        f = ri(rng, 1, 3) if not big else ri(rng, 8, 16)
        h, wd = ri(rng, 3, 40), ri(rng, 3, 40)
        if TILE == 160 and ri(rng, 0, 1):                    # half the cases: frames of whole 160-row tiles (partials per tile)
            h, wd = pick(rng, [(10, 16), (16, 10), (20, 32), (32, 20), (8, 20), (20, 8), (5, 32), (32, 5), (25, 32), (15, 32), (30, 16), (40, 24)])
        cin, cout = 64 * ri(rng, 1, 6), 320 * ri(rng, 1, 2)
        two = cin >= 128 and ri(rng, 0, 3) == 0
        c1 = 64 * ri(rng, 1, cin // 64 - 1) if two else cin
        korder = ri(rng, 0, 1)
        M = f * h * wd
        x = rn(g, M, c1)
        x2 = rn(g, M, cin - c1) if two else None
        w = rn(g, cout, 9 * cin, 0.05)
        b = torch.randn(cout, generator=g, device="cuda") if bias_on else None
        r = None
        if res_kind:
            rdt = {"operand": ops.H16(), "stream": ops.STREAM(), "fp32": torch.float32}[res_kind]
            r = rn(g, M, cout) if rdt == ops.H16() else (torch.randn(M, cout, generator=g, device="cuda") * 0.5).to(rdt)
        stats = h * wd if ri(rng, 0, 1) and (h * wd) % TILE == 0 else 0
        desc = f"conv f={f} {h}x{wd} cin={cin} cout={cout}" + (f" split {c1}" if two else "") + f" korder={korder} out={out_kind} res={res_kind} bias={int(bias_on)} stats={stats}"

        def run(out):
            return ops.conv3x3(x, w, out=out, x2=x2, frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b, residual=r, stats=bool(stats))
        return desc, run, M, cout, odt, r is not None, stats, False
    clips = ri(rng, 1, 2) if not big else ri(rng, 2, 4)
    t = pick(rng, [2, 3, 4, 8, 16])
    hw = ri(rng, 5, 700) if not big else ri(rng, 1000, 2500)
    cin, cout = 64 * ri(rng, 1, 8), 320 * ri(rng, 1, 2)
    M = clips * t * hw
    x, w = rn(g, M, cin), rn(g, cout, 3 * cin, 0.1)
    b = torch.randn(cout, generator=g, device="cuda") if bias_on else None
    r = None
    if res_kind:
        rdt = {"operand": ops.H16(), "stream": ops.STREAM(), "fp32": torch.float32}[res_kind]
        r = rn(g, M, cout) if rdt == ops.H16() else (torch.randn(M, cout, generator=g, device="cuda") * 0.5).to(rdt)
    desc = f"tconv clips={clips} t={t} hw={hw} cin={cin} cout={cout} out={out_kind} res={res_kind} bias={int(bias_on)}"

    def run(out):
        return ops.tconv3(x, w, out=out, clips=clips, t=t, hw=hw, cin=cin, bias=b, residual=r, korder=0)
    return desc, run, M, cout, odt, r is not None, 0, False


def main():
    rng = torch.Generator().manual_seed(args.seed)
    g = torch.Generator(device="cuda").manual_seed(args.seed + 1)
    launches = cases = skipped = 0
    env = {k: os.environ.get(k) for k in ("MUDG_OPERAND", "PYTORCH_NO_CUDA_MEMORY_CACHING", "AMD_SERIALIZE_KERNEL", "HSA_XNACK")}
    print(f"[stress] tile {TILE} build {hip.operand_name()} planes {hip.planes()} seed {args.seed} target {args.launches} tile launches, repeat {args.repeat}, "
          f"LDS poison {'on' if POISON else 'off'}, env {env}", flush=True)
    t0 = time.time()
    while launches < args.launches:
        desc, run, M, nout, odt, has_res, stats_rows, pers = make_case(rng, g)
        pad = pick(rng, [0, 0, 8, 24, 64])
        os.environ["MUDG_GEMM_W288"], os.environ["MUDG_GEMM_W288P"], os.environ["MUDG_GEMM_W160"] = "0", "0", "0"
        ref_whole, ref_view = guarded(M, nout, odt, pad)
        ref = run(ref_view)
        assert LAST["rows"] == 128, "MUDG_GEMM_W288=0 must not select the tile"
        forms = [("tile", "2", "0")] + ([("persistent", "2", "2")] if pers else [])
        first = None
        for form, v, pv in forms:
            os.environ[SWITCH], os.environ["MUDG_GEMM_W288P"] = v, pv
            for i in range(args.repeat):
                poison(launches)
                whole, view = guarded(M, nout, odt, pad)
                y = run(view)
                if LAST["rows"] != TILE:                 # the library declined the tile for this descriptor: nothing to stress
                    break
                launches += 1
                torch.cuda.synchronize()
                if not canaries_intact(whole, M, nout):
                    fail(f"CANARY overwritten: {desc} [{form} run {i}]")
                if first is None:
                    first = (whole, y)
                    if has_res:
                        e = rel(values(y), values(ref))
                        tol = 3e-3 if odt != torch.float32 else 2e-5
                        if not e < tol:
                            fail(f"tile vs 128x128 rel-L2 {e:.2e}: {desc}")
                    elif hip.planes() > 1:
                        # bf16x3: the tile sums x1 w0 + x0 w0 + x0 w1 per k half, the fused-piece 128 x 128 kernels per K-tile — fp32-rounding apart
                        e = rel(values(y), values(ref))
                        if not e < 1e-5:
                            fail(f"tile vs 128x128 rel-L2 {e:.2e}: {desc}")
                    elif not torch.equal(whole.view(torch.int16), ref_whole.view(torch.int16)):
                        fail(f"tile bits != 128x128 bits (rel-L2 {rel(values(y), values(ref)):.2e}): {desc} [{form}]")
                    if stats_rows:
                        p = getattr(y, ops.GN_ATTR)
                        s = p.reshape(-1, stats_rows // TILE, nout, 2).double().sum(1)
                        v64 = values(y).reshape(-1, stats_rows, nout)
                        want = torch.stack([v64.sum(1), (v64 ** 2).sum(1)], -1)
                        e = rel(s, want)
                        if not e < 1e-5:
                            fail(f"GroupNorm partials vs column sums {e:.2e}: {desc}")
                else:
                    if not torch.equal(whole.view(torch.int16), first[0].view(torch.int16)):
                        fail(f"RUN-TO-RUN difference: {desc} [{form} run {i}]")
                    if stats_rows and not torch.equal(getattr(y, ops.GN_ATTR), getattr(first[1], ops.GN_ATTR)):
                        fail(f"RUN-TO-RUN difference in partials: {desc}")
        if first is None:
            skipped += 1
        cases += 1
        if cases % 100 == 0:
            print(f"[stress] {cases} cases, {launches} tile launches clean, {skipped} declined, {time.time() - t0:.0f} s; last: {desc}", flush=True)
    os.environ["MUDG_GEMM_W288"] = "1"
    os.environ["MUDG_GEMM_W160"] = "1"
    os.environ.pop("MUDG_GEMM_W288P", None)
    verdict = "all bit-reproducible, canaries intact" if not FAILURES else f"{len(FAILURES)} FAILURES"
    print(f"[stress] DONE: {cases} cases, {launches} tile launches (one-tile and persistent forms), {verdict}, "
          f"{skipped} cases declined by the library; {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if FAILURES else 0)


if __name__ == "__main__":
    main()
