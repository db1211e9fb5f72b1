"""`diffsynth` compatibility surface (boundary B-inner): exactly the names the
reference's wrapper imports [R infinicube/videogen/inference.py:25-26], backed by infinicube_amd, so a
developer checkout of the reference's own inference.py runs unmodified on MI355X."""
from infinicube_amd.videogen.io import load_state_dict, save_video

__all__ = ["load_state_dict", "save_video"]
