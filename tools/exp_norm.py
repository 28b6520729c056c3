"""GroupNorm / LayerNorm kernels alone on the UNet's shapes (MDM1024, guidance batch 2): microseconds and HBM GB/s of the bytes the pass
must move.  `python tools/exp_norm.py` on an MI355X."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mudg_amd import ops  # noqa: E402

dev = torch.device("cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


SILU = os.environ.get("SILU", "1") == "1"        # SILU=0: GroupNorm alone (the transformers' entry norm)


def main():
    frames = 32
    for hw, c in ((9216, 320), (9216, 640), (9216, 960), (2304, 640), (2304, 1280), (2304, 1920), (576, 1280), (576, 2560), (144, 1280), (144, 2560)):
        rows = frames * hw
        x = (torch.randn(rows, c, device=dev) * 2).to(ops.STREAM())
        g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        out = ops.empty_rows(rows, c, ops.H16(), dev)
        xb = x.element_size()
        us = timed(lambda: ops.groupnorm(x, g, b, samples=frames, rows=hw, eps=1e-5, silu=SILU, out=out))
        by = rows * c * (2 * xb + 2)
        print(f"groupnorm+silu stats+apply  [{frames} x {hw}][{c}]: {us:8.1f} us  {by / us / 1e3:7.0f} GB/s of {by / 1e6:.0f} MB")
        if hw % 128 == 0:
            # the common case in the UNet: the producer's epilogue left per-(128-row block, channel) partial sums, GroupNorm = fold + apply
            xf = x.float().view(-1, 128, c)
            part = torch.stack([xf.sum(1), (xf * xf).sum(1)], -1).contiguous()
            setattr(x, ops.GN_ATTR, part)
            setattr(x, ops.GN_ATTR + "_version", ops._version(x))
            us = timed(lambda: ops.groupnorm(x, g, b, samples=frames, rows=hw, eps=1e-5, silu=SILU, out=out))
            by = rows * c * (xb + 2)
            print(f"groupnorm+silu fold+apply   [{frames} x {hw}][{c}]: {us:8.1f} us  {by / us / 1e3:7.0f} GB/s of {by / 1e6:.0f} MB")
            setattr(x, ops.GN_ATTR, None)
            del xf, part
        if hw % 8 == 0 and c <= 1280:
            us = timed(lambda: ops.layernorm(x, g, b, out=out))
            by = rows * c * (xb + 2)
            print(f"layernorm                   [{rows}][{c}]: {us:8.1f} us  {by / us / 1e3:7.0f} GB/s of {by / 1e6:.0f} MB")


if __name__ == "__main__":
    main()
