"""`infinicube.utils.semantic_utils` names served by the MI355X-native implementation."""
from infinicube_amd.utils.semantic_utils import (WAYMO_CATEGORY_NAMES, WAYMO_MAPPING, WAYMO_PALETTE,
                                                 generate_rgb_semantic_buffer, semantic_to_color)

__all__ = ["WAYMO_CATEGORY_NAMES", "WAYMO_MAPPING", "WAYMO_PALETTE", "generate_rgb_semantic_buffer", "semantic_to_color"]

from . import overlay_reference as _overlay

_overlay(__name__, globals(), __all__)
