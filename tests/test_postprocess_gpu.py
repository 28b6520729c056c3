"""Per-modality post-processing of decoded frames (virtual_render/eval_tools.py) on the HIP path: byte / integer work,
so the bar is bit-equality — against the fixtures captured from the reference's own function and expressions, and against
the numpy oracle on larger seeded inputs, ties, saturated values and odd sizes."""
import numpy as np
import pytest
import torch

from helpers import golden

pytestmark = pytest.mark.gpu


def test_postprocess_matches_reference_fixtures_bit_for_bit(cuda):
    from mudg_amd import ops
    g = golden("postprocess.pt")
    u8 = ops.frames_to_uint8(g["video"].to(cuda))
    assert u8.dtype == torch.uint8 and torch.equal(u8.cpu(), g["u8"])
    depth = ops.depth_from_uint8(g["u8"].to(cuda))
    assert torch.equal(depth.cpu(), g["depth"])
    vis, lab = ops.semantic_nearest(g["semantic_in"].to(cuda))
    assert torch.equal(lab.cpu(), g["semantic_labels"]) and torch.equal(vis.cpu(), g["semantic_vis"])


def test_postprocess_matches_oracle_on_frame_sized_inputs(cuda):
    from mudg_amd import ops
    from oracle import postprocess as pp
    gen = torch.Generator().manual_seed(5)
    video = torch.randn(3, 3, 5, 73, 131, generator=gen)                       # odd sizes, values beyond [-1, 1]
    video[0, 0, 0, 0, :6] = torch.tensor([float("inf"), -float("inf"), 1.0, -1.0, 0.99999994, -0.99999994])
    u8 = ops.frames_to_uint8(video.to(cuda))
    want = pp.frames_to_uint8(video.numpy())
    assert np.array_equal(u8.cpu().numpy(), want)
    assert np.array_equal(ops.depth_from_uint8(u8).cpu().numpy(), pp.depth_from_uint8(want))
    img = torch.randint(0, 256, (3, 576, 1024), generator=gen, dtype=torch.uint8)      # one full MDM1024 frame
    vis, lab = ops.semantic_nearest(img.to(cuda))
    wv, wl = pp.visualize_semantic(img.numpy())
    assert np.array_equal(lab.cpu().numpy(), wl) and np.array_equal(vis.cpu().numpy(), wv)
    assert int(lab.min()) >= 0 and int(lab.max()) <= 18


def test_eval_tools_mirror_keeps_the_reference_signature(cuda):
    from virtual_render import eval_tools
    g = golden("postprocess.pt")
    vis_np, lab_np = eval_tools.visualize_semantic(g["semantic_in"])                    # numpy (H, W, 3), (H, W) like the reference
    assert isinstance(vis_np, np.ndarray) and vis_np.shape == (24, 32, 3) and lab_np.shape == (24, 32)
    assert np.array_equal(lab_np, g["semantic_labels"].numpy())
    vis_pt, lab_pt = eval_tools.visualize_semantic(g["semantic_in"], return_pt=True)    # tensors (3, H, W), (H, W)
    assert torch.equal(vis_pt.cpu(), g["semantic_vis"]) and torch.equal(lab_pt.cpu(), g["semantic_labels"])
    grid = eval_tools.frames_to_uint8(g["video"].to(cuda))
    assert torch.equal(eval_tools.depth_prediction(grid[0, 1]).cpu(), g["depth"][0, 1])
