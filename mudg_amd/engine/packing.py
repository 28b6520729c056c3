"""Weight caches for the kernels: bf16, GEMM-ready layouts derived from the nn.Parameters the boundary modules own.

The parameters stay the single source of truth (fp32 nn.Parameter, reference names).  A packed copy is rebuilt
whenever the parameter object or its in-place version counter changes — load_state_dict, optimizer steps and module
surgery all invalidate it — so the caches never need manual flushing.  Packing is one-off layout work (torch ops on
the device), not part of the per-step path.
"""
import torch

from .. import hip



def _version(p):
    try:
        return p._version
    except RuntimeError:            # tensors created under torch.inference_mode() do not track versions
        return 0


def _key(params):
    return tuple((p.data_ptr(), _version(p), p.dtype, tuple(p.shape)) for p in params)


def cached(module, name, params, build):
    store = module.__dict__.setdefault("_mudg_packed", {})
    key = _key(params)
    hit = store.get(name)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        val = build()
    store[name] = (key, val)
    return val


def _need_cuda(p, what):
    if not p.is_cuda:
        raise RuntimeError(f"{what}: parameters must live on the GPU (got {p.device}); the MI355X path has no CPU "
                           "fallback — move the model with .cuda() first")


def f32(module, name):
    """A parameter as contiguous fp32 (biases, norm scales): usually the parameter's own storage."""
    p = getattr(module, name)
    if p is None:
        return None
    _need_cuda(p, type(module).__name__)
    if p.dtype == torch.float32 and p.is_contiguous():
        return p.detach()
    return cached(module, "f32:" + name, (p,), lambda: p.detach().float().contiguous())


def operand(t2d):
    """fp32 [N, K] -> the MFMA operand matrix the kernels read: a 16-bit cast, or in the split-operand builds the
    pieces hi = bf16(w), mid = bf16(w - hi), ... side by side ([N, planes * K] storage, returned as the [N, K] view of
    piece 0; see ops.empty_rows).  One-off layout work on the device, redone only when the parameter changes."""
    planes, dt = hip.planes(), hip.operand_dtype()
    t2d = t2d.detach()
    if planes == 1:
        return t2d.to(dt).contiguous()
    rest = t2d.float()
    pieces = []
    for _ in range(planes):
        piece = rest.to(dt)
        pieces.append(piece)
        rest = rest - piece.float()
    return torch.cat(pieces, dim=1).contiguous()[:, :t2d.shape[1]]


def linear(mod):
    """[N, K] bf16 from nn.Linear / 1x1 Conv2d / k=1 Conv1d weights."""
    w = mod.weight
    _need_cuda(w, type(mod).__name__)
    return cached(mod, "w", (w,), lambda: operand(w.detach().reshape(w.shape[0], -1)))


def linear_cat(owner, name, mods):
    """Rows of several bias-free Linears stacked into one [sum N, K] bf16 matrix (fused q/k or q/k/v projection)."""
    ws = tuple(m.weight for m in mods)
    for w in ws:
        _need_cuda(w, name)
    return cached(owner, name, ws, lambda: operand(torch.cat([w.detach().float() for w in ws], 0)))


def qk_prescaled(owner, to_q, to_k, scale):
    """[q | k] projection rows with the attention scale AND log2(e) folded into the q rows (in fp32, before the one cast
    to the operand type — no extra rounding): Q K^T then is the base-2 exponent of the softmax directly
    (MudgAttnDesc.q_prescaled)."""
    import math
    ws = (to_q.weight, to_k.weight)
    for w in ws:
        _need_cuda(w, "qk")
    c = float(scale) * math.log2(math.e)
    return cached(owner, f"qk_prescaled:{c!r}", ws,
                  lambda: operand(torch.cat([to_q.weight.detach().float() * c, to_k.weight.detach().float()], 0)))


def conv3x3(mod):
    """(Cout, Cin, 3, 3) -> bf16 [Cout][K]; returns (matrix, padded Cin, korder).  K is ordered [Cin/64][tap][64]
    when Cin is a multiple of 64 (korder 1: the nine taps of a channel slab are adjacent K tiles, which keeps the
    shifted input re-reads in L2), else [tap][Cin padded to a multiple of 8] (korder 0)."""
    w = mod.weight
    _need_cuda(w, type(mod).__name__)
    cout, cin = w.shape[0], w.shape[1]
    cpad = (cin + 7) // 8 * 8
    korder = 1 if cin % 64 == 0 else 0

    def build():
        t = w.detach().permute(0, 2, 3, 1)                       # Cout, ky, kx, Cin
        if korder:
            t = t.reshape(cout, 9, cin // 64, 64).permute(0, 2, 1, 3)
        elif cpad != cin:
            t = torch.nn.functional.pad(t, (0, cpad - cin))
        return operand(t.reshape(cout, 9 * cpad))

    return cached(mod, "w3x3", (w,), build), cpad, korder


def conv3x3_subpixel(mod):
    """(Cout, Cin, 3, 3) -> the four [Cout][4 Cin] matrices of the sub-pixel form of "nearest-2x upsample, then this conv"
    (MudgGemmDesc.subpixel), stacked [4 Cout][4 Cin]; None when Cin is not a multiple of 64 (the caller then runs the
    upsampling loader with the 3x3 weights).  An output pixel of parity (py, px) sees the low-resolution pixels
    (oy - 1 + py + a, ox - 1 + px + b), a, b in {0, 1}; the 3x3 taps that land on the same one are summed (in fp32, before
    the operand rounding): rows {0 | 1, 2} for py = 0, {0, 1 | 2} for py = 1, likewise the columns."""
    w = mod.weight
    _need_cuda(w, type(mod).__name__)
    cout, cin = w.shape[0], w.shape[1]
    if cin % 64:
        return None

    def build():
        t = w.detach().float()                                   # Cout, Cin, ky, kx
        sets = (((0,), (1, 2)), ((0, 1), (2,)))
        mats = []
        for py in range(2):
            for px in range(2):
                taps = [sum(t[:, :, dy, dx] for dy in sets[py][a] for dx in sets[px][b]) for a in range(2) for b in range(2)]
                m = torch.stack(taps, 1)                         # Cout, tap (2a + b), Cin
                mats.append(m.reshape(cout, 4, cin // 64, 64).permute(0, 2, 1, 3).reshape(cout, 4 * cin))
        return operand(torch.cat(mats, 0))

    return cached(mod, "w3x3sub", (w,), build)


def tconv(mod, slab=False):
    """(Cout, Cin, 3, 1, 1) -> [Cout][tap][Cin] bf16, or (slab: ops.tconv3's korder 1) [Cout][Cin/64][tap][64]."""
    w = mod.weight
    _need_cuda(w, type(mod).__name__)
    if slab:
        cout, cin = w.shape[0], w.shape[1]
        return cached(mod, "wt_slab", (w,), lambda: operand(w.detach()[:, :, :, 0, 0].permute(0, 2, 1).reshape(cout, 3, cin // 64, 64)
                                                                 .permute(0, 2, 1, 3).reshape(cout, -1)))
    return cached(mod, "wt", (w,), lambda: operand(w.detach()[:, :, :, 0, 0].permute(0, 2, 1).reshape(w.shape[0], -1)))


def geglu(mod):
    """GEGLU projection [2I, K] (+bias) reordered into blocks of 32 value rows followed by their 32 gate rows,
    the order mudg_gemm's fused GEGLU epilogue expects.  Returns (weight bf16, bias fp32)."""
    w, b = mod.weight, mod.bias
    _need_cuda(w, "GEGLU")
    inner = w.shape[0] // 2
    if inner % 32:
        raise RuntimeError(f"GEGLU inner width {inner} must be a multiple of 32")

    def build():
        idx = torch.arange(inner, device=w.device).reshape(-1, 32)
        order = torch.cat([idx, idx + inner], dim=1).reshape(-1)
        return (operand(w.detach()[order]), b.detach()[order].float().contiguous())

    return cached(mod, "geglu", (w, b), build)
