#!/usr/bin/env python3
"""AutoencoderKL decode of one 16-frame MDM1024 clip (the second term of clips/min)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import factory
from tools.kernel_bench import timeit

model = factory.build_synthetic_model("1024", "cuda", seed=3)
z = torch.randn(1, 4, 16, 72, 128, device="cuda") * 0.5
with torch.no_grad():
    sec = timeit(lambda: model.decode_first_stage(z), iters=3, warm=1)
print(f"VAE decode 16 frames 576x1024: {sec*1e3:.1f} ms", flush=True)
