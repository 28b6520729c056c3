import os, sys, torch
sys.path.insert(0, os.getcwd())
from mudg_amd import ops
torch.manual_seed(0)
BF = torch.bfloat16
def rel(a, b): return ((a.float() - b.float()).norm() / b.float().norm()).item()
for (M, N, K, g, od) in [(65536, 512, 320, 0, None), (40000 + 37, 960, 320, 0, None), (36864, 2560, 320, 1, None), (131072, 320, 320, 0, "stream"),
                         (131072 + 5, 320, 320, 0, "f32"), (65536, 1280, 320, 1, "stream"), (36864 + 77, 1024, 320, 0, "stream")]:
    x = (torch.randn(M, K) * 1.0).to(BF).cuda(); w = (torch.randn(N, K) * 0.05).to(BF).cuda(); b = torch.randn(N).cuda()
    y = ops.gemm(x, w, bias=b, geglu=bool(g), out_stream=od == "stream", out_fp32=od == "f32")
    acc = x.float() @ w.float().t() + b
    if g:
        a = acc.view(M, N // 64, 2, 32)
        ref = (a[:, :, 0] * torch.nn.functional.gelu(a[:, :, 1])).reshape(M, N // 2)
    else:
        ref = acc
    print(M, N, K, g, od, y.dtype, tuple(y.shape), "rel", rel(y, ref), "maxabs", (y.float() - ref).abs().max().item(), flush=True)
