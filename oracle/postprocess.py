"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the per-modality post-processing MuDG's driver applies to decoded
frames.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.

Reference: virtual_render/eval_tools.py
  * 22-27 / 59-63 / 109-113   clamp to [-1, 1], (x + 1) / 2 * 255, `.to(torch.uint8)` (truncation), (c,t,h,w) -> (t,h,w,c)
  * 71                        depth prediction = mean over the three uint8 channels / 255
  * 309-347 visualize_semantic  nearest of 19 palette colours (np.linalg.norm over the channel axis, np.argmin: first
                              minimum wins), recoloured with the palette
Pinned by tests/golden/postprocess.pt, captured from the reference's own function / expressions."""
import numpy as np

PALETTE = np.array([[255, 120, 50], [255, 192, 203], [255, 255, 0], [0, 150, 245], [0, 255, 255], [255, 127, 0], [255, 0, 0],
                    [255, 240, 150], [135, 60, 0], [160, 32, 240], [255, 0, 255], [139, 137, 137], [75, 0, 75], [150, 240, 80],
                    [230, 230, 250], [0, 175, 0], [0, 255, 127], [222, 155, 161], [140, 62, 69]], dtype=np.int64)   # eval_tools.py:312-332


def frames_to_uint8(video):
    """(b, c, t, h, w) float -> (b, t, h, w, c) uint8, eval_tools.py:22-27."""
    v = np.clip(np.asarray(video, dtype=np.float32), np.float32(-1.0), np.float32(1.0))
    g = (v + np.float32(1.0)) / np.float32(2.0) * np.float32(255.0)
    return np.transpose(g.astype(np.uint8), (0, 2, 3, 4, 1))          # truncation toward zero like Tensor.to(uint8)


def depth_from_uint8(frames):
    """(..., h, w, 3) uint8 -> (..., 1, h, w) float32 in [0, 1], eval_tools.py:71 (fp32 mean of three, then / 255)."""
    f = np.asarray(frames).astype(np.float32)
    s = (f[..., 0] + f[..., 1]) + f[..., 2]
    return ((s / np.float32(3.0)) / np.float32(255.0))[..., None, :, :]


def visualize_semantic(img):
    """(3, h, w) uint8 -> ((3, h, w) uint8 recoloured, (h, w) int64 labels), eval_tools.py:309-347."""
    x = np.transpose(np.asarray(img).astype(np.int64), (1, 2, 0))     # h, w, 3
    d2 = ((x[:, :, None, :] - PALETTE[None, None]) ** 2).sum(-1)      # squared distances are integers: argmin is exact
    lab = np.argmin(d2, axis=2)
    vis = PALETTE[lab].astype(np.uint8)
    return np.transpose(vis, (2, 0, 1)), lab.astype(np.int64)
