"""MI355X-native ``infinicube.videogen``: ``from infinicube_amd.videogen import WanVideoGenerator``
(or, unchanged for the reference's caller, ``from infinicube.videogen import WanVideoGenerator``)."""

from .inference import WanVideoGenerator

__all__ = ["WanVideoGenerator"]
