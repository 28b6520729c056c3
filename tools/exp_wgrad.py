#!/usr/bin/env python3
"""The row-contracting weight-gradient kernel (mudg_wgrad) alone on the UNet's shapes at MDM1024 (B = 1): microseconds and TFLOP/s,
next to the K-sliced GEMM over already-transposed operands that it replaces (the transposes themselves not counted).
`python tools/exp_wgrad.py` on an MI355X; ONLY=linear|conv|tconv to restrict."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from mudg_amd.train import kernels as K, functions as Fn
from tools.kernel_bench import timeit, rn

only = os.environ.get("ONLY", "")
L = [(147456, 320, 320), (147456, 2560, 320), (147456, 320, 1280), (36864, 640, 640), (36864, 5120, 640), (9216, 1280, 1280), (9216, 10240, 1280)]
if not only or "linear" in only:
    for (P, m, c) in L:
        a, b = rn(P, m), rn(P, c)
        sec = timeit(lambda: K.wgrad(a, b, positions=P, m=m, c=c), iters=5)
        at, bt = Fn._zeros_operand(m, Fn._width(m, c, P), a.device), Fn._zeros_operand(c, Fn._width(m, c, P), a.device)
        ref = timeit(lambda: Fn.wgrad_gemm(at, bt, m, c, P), iters=5)
        print(f"linear dW[{m}][{c}] over {P}: wgrad {sec*1e6:8.1f} us {2.0*P*m*c/sec/1e12:6.1f} TF | gemm on transposed {ref*1e6:8.1f} us {2.0*P*m*c/ref/1e12:6.1f} TF", flush=True)
C = [(16, 72, 128, 320, 320), (16, 72, 128, 640, 320), (16, 36, 64, 640, 640), (16, 18, 32, 1280, 1280), (16, 9, 16, 1280, 1280)]
if not only or "conv" in only:
    for (f, h, w, ci, co) in C:
        P = f * h * w
        a, b = rn(P, co), rn(P, ci)
        sec = timeit(lambda: K.wgrad(a, b, positions=P, m=co, c=ci, taps=9, mode=1, geo=dict(Hin=h, Win=w, Hout=h, Wout=w, stride=1, pad=1)), iters=5)
        n = 9 * ci
        at, bt = Fn._zeros_operand(co, Fn._width(co, n, P), a.device), Fn._zeros_operand(n, Fn._width(co, n, P), a.device)
        ref = timeit(lambda: Fn.wgrad_gemm(at, bt, co, n, P), iters=5)
        print(f"conv dW[{co}][9 x {ci}] over {P}: wgrad {sec*1e6:8.1f} us {2.0*P*co*n/sec/1e12:6.1f} TF | gemm on transposed {ref*1e6:8.1f} us {2.0*P*co*n/ref/1e12:6.1f} TF", flush=True)
T = [(1, 16, 9216, 320), (1, 16, 2304, 640), (1, 16, 576, 1280)]
if not only or "tconv" in only:
    for (clips, t, hw, c) in T:
        P = clips * t * hw
        a, b = rn(P, c), rn(P, c)
        sec = timeit(lambda: K.wgrad(a, b, positions=P, m=c, c=c, taps=3, mode=2, geo=dict(T=t, HW=hw)), iters=5)
        print(f"tconv dW[{c}][3 x {c}] over {P}: wgrad {sec*1e6:8.1f} us {2.0*P*c*3*c/sec/1e12:6.1f} TF", flush=True)
