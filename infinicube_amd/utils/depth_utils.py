"""`vis_depth` — the depth visualisation stage 2 writes as `depth_vis_video_<res>_front.mp4`
[R infinicube/utils/depth_utils.py:20-69; infinicube/inference/guidance_buffer_generation.py:679-683,726].  Same name,
arguments and result.  A per-frame percentile + 256-entry colormap lookup for a DEBUG video nothing downstream reads:
host-side numpy / matplotlib like the reference (no kernel), pinned against the reference's own output
(tests/golden/vis_depth_cases.npz, generator tests/golden/make_vis_depth_golden.py)."""
from __future__ import annotations

import numpy as np
import torch


def vis_depth(depth, minmax=None, valid_farthest=300):
    """depth [H, W] (ndarray or tensor; 0 / huge = no value) -> uint8 [H, W, 3] 'magma_r' colouring between the 0.5 and
    99.5 percentiles (of the depths below ``valid_farthest`` for the upper one), or between ``minmax``."""
    import matplotlib as mpl
    from matplotlib import cm
    is_tensor = isinstance(depth, torch.Tensor)
    if is_tensor:
        device = depth.device
        depth = depth.detach().cpu().numpy()
    depth = np.nan_to_num(depth)
    if minmax is None:
        constant_max = np.percentile(depth[depth < valid_farthest], 99.5)
        p_lo = np.percentile(depth, 0.5)
        constant_min = p_lo if p_lo < constant_max else 0
    else:
        constant_min, constant_max = minmax
    mapper = cm.ScalarMappable(norm=mpl.colors.Normalize(vmin=constant_min, vmax=constant_max), cmap="magma_r")
    colored = (mapper.to_rgba(depth)[:, :, :3] * 255).astype(np.uint8)
    return torch.from_numpy(colored).to(device) if is_tensor else colored
