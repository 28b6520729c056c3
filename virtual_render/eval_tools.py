"""Tensor-level part of MuDG's result post-processing (reference: virtual_render/eval_tools.py) on the MI355X path: what
`save_virtual_{color,depth,semantic}_results` compute before they hand pixels to the PNG / NPY writers.  The writers, the
matplotlib depth colour map and the mp4 export are host I/O outside the path.

  frames_to_uint8     eval_tools.py:22-27, 59-63, 109-113   clamp, (x + 1) / 2 * 255, truncate, (b c t h w) -> (b t h w c)
  depth_prediction    eval_tools.py:71                       channel mean of the uint8 frame / 255  -> (1, h, w) in [0, 1]
  visualize_semantic  eval_tools.py:309-347                  nearest of the 19 class colours; same signature and return
"""
import torch

from mudg_amd import ops


def frames_to_uint8(video):
    """(b, c, t, h, w) samples -> (b, t, h, w, c) uint8 like `grid` in the reference's save functions."""
    return ops.frames_to_uint8(video)


def depth_prediction(grid_frame):
    """(h, w, 3) uint8 frame of the depth stream -> (1, h, w) fp32 depth in [0, 1] (`result_pred`, eval_tools.py:71)."""
    return ops.depth_from_uint8(grid_frame)


def visualize_semantic(semantic, return_pt=False):
    """(3, H, W) uint8 -> (recoloured (H, W, 3) [or (3, H, W) tensor with return_pt], labels (H, W)) as eval_tools.py:309-347:
    numpy arrays by default, torch tensors with return_pt=True."""
    if not torch.is_tensor(semantic):
        semantic = torch.as_tensor(semantic)
    vis, lab = ops.semantic_nearest(semantic.to(torch.uint8).cuda() if not semantic.is_cuda else semantic.to(torch.uint8))
    if return_pt:
        return vis, lab
    return vis.permute(1, 2, 0).cpu().numpy(), lab.cpu().numpy()
