#!/usr/bin/env python3
"""Where a GEGLU tile's time goes: s_memtime stamps of wave 0 of every workgroup at the phase boundaries of the persistent 288 x 256
kernel (wgemm_pkernel<4, true>) and of the two-workgroup 144 x 256 kernel (hgeglu_kernel), debug-variants build.
    MUDG_DEBUG_VARIANTS=1 python tools/exp_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MUDG_DEBUG_VARIANTS"] = "1"
import torch
from mudg_amd import hip, ops

lib = hip.lib()
lib.mudg_debug_set_stamps.restype = ctypes.c_int
lib.mudg_debug_set_stamps.argtypes = [ctypes.c_void_p]
rn = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(ops.H16())


def run(M, N, K, hw, w288, h144, nwg):
    x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
    os.environ["MUDG_GEMM_W288"], os.environ["MUDG_GEMM_H144"] = w288, h144
    for _ in range(3):
        ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw)
    buf = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
    assert lib.mudg_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw)
    e1.record()
    torch.cuda.synchronize()
    assert lib.mudg_debug_set_stamps(ctypes.c_void_p(0)) == 0
    us = e0.elapsed_time(e1) * 1e3
    st = buf.cpu().reshape(nwg, 64)
    used = st[:, 0] > 0
    st = st[used]
    # s_memtime counters of different XCDs are not aligned: only differences inside one workgroup mean anything.  The tick is calibrated on
    # the persistent kernel, whose workgroups live for (nearly) the whole launch.
    last = st.max(dim=1).values
    tick_ns = float((us * 1e3 / (last - st[:, 0]).double()).median())
    return st, us, tick_ns


for (M, N, K, hw) in [(294912, 2560, 320, 9216), (73728, 5120, 640, 2304)]:
    tiles = (M // 288) * (N // 256)
    st, us, tick = run(M, N, K, hw, "1", "0", 256)
    print(f"geglu {M}x{N}x{K}  persistent 288 x 256 (256 workgroups, {tiles / 256:.0f} tiles each): {us:.0f} us by events; stamp span -> {tick:.2f} ns per tick")
    nt = int(((st[0, 1:] > 0).sum()) // 2)
    ep0, ep1 = st[:, 1:1 + 2 * nt:2].double(), st[:, 2:2 + 2 * nt:2].double()
    first = (ep0[:, 0] - st[:, 0].double()) * tick / 1e3
    epi = (ep1 - ep0) * tick / 1e3
    loop = (ep0[:, 1:] - ep1[:, :-1]) * tick / 1e3
    print(f"   start -> first epilogue {first.mean():.2f} us (first-fetch latency + one K loop); K loop of a later tile {loop.mean():.2f} us (min {loop.min():.2f}, max {loop.max():.2f}); "
          f"epilogue {epi.mean():.2f} us (min {epi.min():.2f}, max {epi.max():.2f}); {nt} tiles stamped per workgroup")
    tiles2 = ((M + 143) // 144) * (N // 256)
    st, us, _ = run(M, N, K, hw, "2", "2", tiles2)
    d = st.double()
    pro, loop, epi = (d[:, 1] - d[:, 0]) * tick / 1e3, (d[:, 2] - d[:, 1]) * tick / 1e3, (d[:, 3] - d[:, 2]) * tick / 1e3
    life = (d[:, 3] - d[:, 0]) * tick / 1e3
    print(f"   two-workgroup 144 x 256 ({tiles2} workgroups): {us:.0f} us by events, {tick:.2f} ns per tick; per workgroup: first k half landed after {pro.mean():.2f} us "
          f"(min {pro.min():.2f}, max {pro.max():.2f}), K loop {loop.mean():.2f} (min {loop.min():.2f}, max {loop.max():.2f}), epilogue {epi.mean():.2f} "
          f"(min {epi.min():.2f}, max {epi.max():.2f}), stamped lifetime {life.mean():.2f}; slots x lifetime = {512 * us / tiles2:.2f} us per workgroup "
          f"=> {512 * us / tiles2 - life.mean():.2f} us per workgroup outside the stamps (launch, table copy, store drain)")
os.environ["MUDG_GEMM_W288"], os.environ["MUDG_GEMM_H144"] = "1", "1"

# Round 6, second half: where a one-tile workgroup of the plain contraction tiles spends its life, and what a CU spends BETWEEN two
# workgroups (launch, descriptor set-up, the drain of the stores): slots x launch time / tiles - stamped lifetime.
rs = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(ops.STREAM())


def run_plain(fn, nwg):
    for _ in range(3):
        fn()
    buf = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
    assert lib.mudg_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    assert lib.mudg_debug_set_stamps(ctypes.c_void_p(0)) == 0
    return buf.cpu().reshape(nwg, 64).double(), e0.elapsed_time(e1) * 1e3


def report(name, fn, tiles, tick):
    d, us = run_plain(fn, tiles)
    d = d[d[:, 0] > 0]
    pro, loop, epi = (d[:, 1] - d[:, 0]) * tick / 1e3, (d[:, 2] - d[:, 1]) * tick / 1e3, (d[:, 3] - d[:, 2]) * tick / 1e3
    life = (d[:, 3] - d[:, 0]) * tick / 1e3
    slot = 256 * us / tiles if tiles >= 256 else us
    print(f"{name}: {us:.0f} us, {tiles} one-tile workgroups = {slot:.2f} us of a CU each; stamped: first k half (+ residual) landed after {pro.mean():.2f} us "
          f"(min {pro.min():.2f}, max {pro.max():.2f}), K loop {loop.mean():.2f} (min {loop.min():.2f}, max {loop.max():.2f}), epilogue {epi.mean():.2f} "
          f"(min {epi.min():.2f}, max {epi.max():.2f}), lifetime {life.mean():.2f} => {slot - life.mean():.2f} us per workgroup outside the stamps", flush=True)


tick = 0.6          # ns per s_memtime tick, re-calibrated on a long persistent launch below
st, us, tick = run(294912, 2560, 320, 9216, "1", "0", 256)
print(f"tick {tick:.3f} ns")
os.environ["MUDG_GEMM_W288"] = "2"
for (M, N, K, hw, resid) in [(294912, 320, 320, 9216, 1), (294912, 320, 320, 9216, 0), (294912, 320, 1280, 9216, 1), (294912, 960, 320, 9216, 0), (73728, 640, 640, 2304, 1),
                              (73728, 640, 2560, 2304, 1), (18432, 1280, 5120, 576, 1)]:
    x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
    r = rs(M, N) if resid else None
    report(f"gemm {M}x{N}x{K} residual={resid} on the 288 x 320 tile", lambda: ops.gemm(x, w, bias=b, residual=r, out_stream=bool(resid), frame_rows=hw), (M // 288) * (N // 320), tick)
for (f, h, w_, cin, cout) in [(32, 72, 128, 320, 320), (32, 36, 64, 1280, 640)]:
    x, w = rn(f * h * w_, cin), rn(cout, 9 * cin)
    report(f"conv {f * h * w_}x{cout}x{9 * cin} on the 288 x 320 tile", lambda: ops.conv3x3(x, w, frames=f, hin=h, win=w_, cin=cin, korder=1, stats=True), (f * h * w_ // 288) * (cout // 320), tick)
os.environ["MUDG_GEMM_W288"], os.environ["MUDG_GEMM_W160"] = "0", "2"
for (M, N, K, hw, resid) in [(81920, 320, 320, 2560, 1), (81920, 320, 1280, 2560, 1), (20480, 640, 2560, 640, 1)]:
    x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
    r = rs(M, N) if resid else None
    report(f"gemm {M}x{N}x{K} residual={resid} on the 160 x 320 tile", lambda: ops.gemm(x, w, bias=b, residual=r, out_stream=bool(resid), frame_rows=hw), (M // 160) * (N // 320), tick)
for (f, h, w_, cin, cout) in [(32, 40, 64, 320, 320), (32, 20, 32, 1280, 640)]:
    x, w = rn(f * h * w_, cin), rn(cout, 9 * cin)
    report(f"conv {f * h * w_}x{cout}x{9 * cin} on the 160 x 320 tile", lambda: ops.conv3x3(x, w, frames=f, hin=h, win=w_, cin=cin, korder=1, stats=True), (f * h * w_ // 160) * (cout // 320), tick)
os.environ["MUDG_GEMM_W288"], os.environ["MUDG_GEMM_W160"] = "1", "1"
