"""hipGraph replay of a whole UNet forward.

A forward at MDM1024 is ~1100 kernel launches issued from Python; at the coarse levels the kernels are shorter than
the time it takes to issue them, which leaves ~6 % of the step as gaps.  Capturing the launch sequence once per input
signature into a hipGraph (through torch.cuda.CUDAGraph, which also pins every intermediate buffer in a private pool,
so all addresses are stable) and replaying it removes the host from the loop.

Opt-in per model: `model.use_hip_graph = True` on a boundary UNetModel.  Inputs are copied into static buffers before
each replay and the result is copied out, so callers keep ordinary tensor semantics.  A graph is dropped and re-captured
whenever any parameter of the model changed (load_state_dict, optimiser step, module surgery): the packed bf16 weight
copies it points at would be stale otherwise.
"""
import torch

from . import unet as U
from .packing import _version


def _params_signature(model):
    sig = 0
    for p in model.parameters():
        sig = (sig * 1000003 + _version(p) + (p.data_ptr() & 0xFFFFF)) & 0xFFFFFFFFFFFF
    return sig


class UNetGraphs:
    def __init__(self, model):
        self.model = model
        self.entries = {}
        self.signature = None

    def _key(self, parts, timesteps, c_label, context, fs):
        ctx_key = ("prepared", context.shape, context.T) if isinstance(context, U.PreparedContext) else (tuple(context.shape), context.dtype)
        return (tuple((tuple(p.shape), p.dtype) for p in parts), tuple(timesteps.shape),
                None if c_label is None else tuple(c_label.shape), ctx_key, None if fs is None else tuple(fs.shape))

    def __call__(self, parts, timesteps, c_label, context, fs):
        sig = _params_signature(self.model)
        if sig != self.signature:
            self.entries.clear()
            self.signature = sig
        timesteps = torch.as_tensor(timesteps, device=parts[0].device)
        # A prepared context whose projections were made with other parameter values (load_state_dict / LoRA swap between two
        # runs that reuse the same cond tensors) is re-projected HERE, at the source: the graph's own copy is only ever
        # filled from a context whose signature is the current one, never from stale K / V^T.
        refreshed = False
        if isinstance(context, U.PreparedContext) and context.signature != sig:
            context.project(self.model)
            refreshed = True
        key = self._key(parts, timesteps, c_label, context, fs)
        entry = self.entries.get(key)
        if entry is None:
            static = {
                "parts": [p.clone() for p in parts], "t": timesteps.clone(),
                "label": None if c_label is None else torch.as_tensor(c_label, device=parts[0].device).clone(),
                # a PreparedContext is cloned into buffers the graph owns (token rows + per-layer K / V^T); a replay with a
                # DIFFERENT prepared context copies that one's buffers over them first — once per sampling run, not per step
                "ctx": context.clone(), "ctx_src": context,
                "fs": None if fs is None else torch.as_tensor(fs, device=parts[0].device).clone(),
            }
            run = lambda: U.forward(self.model, static["parts"], static["t"], c_label=static["label"],
                                    context=static["ctx"], fs=static["fs"])
            run()                                    # eager warm-up: lazy library state, weight packing, allocator
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static["out"] = run()
            entry = (graph, static)
            self.entries[key] = entry
        graph, static = entry
        for dst, src in zip(static["parts"], parts):
            dst.copy_(src)
        static["t"].copy_(timesteps)
        if c_label is not None:
            static["label"].copy_(torch.as_tensor(c_label, device=static["label"].device))
        if isinstance(context, U.PreparedContext):
            if static["ctx_src"] is not context or refreshed or static["ctx"].signature != context.signature:
                static["ctx"].copy_from(context)
                static["ctx_src"] = context
        else:
            static["ctx"].copy_(context)
        if fs is not None:
            static["fs"].copy_(torch.as_tensor(fs, device=static["fs"].device))
        graph.replay()
        return static["out"].clone()
