"""`infinicube.utils.buffer_utils` names served by the MI355X-native implementation."""
from infinicube_amd.utils.buffer_utils import generate_coordinate_buffer_from_memory_global_norm

__all__ = ["generate_coordinate_buffer_from_memory_global_norm"]

from . import overlay_reference as _overlay

_overlay(__name__, globals(), __all__)
