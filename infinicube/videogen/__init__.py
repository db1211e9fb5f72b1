"""`infinicube.videogen` served by the MI355X-native implementation (boundary B-outer)."""
from infinicube_amd.videogen.inference import WanVideoGenerator

__all__ = ["WanVideoGenerator"]
