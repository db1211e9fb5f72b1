"""infinicube_amd — MI355X-native implementation of InfiniCube's video-generation hot path.

Scope: the guidance-buffer-conditioned Wan2.1 DiT denoising loop behind
``infinicube.videogen.WanVideoGenerator`` [R infinicube/videogen/inference.py:30-240].
Python host code here drives a C ABI (``include/icvideo.h``) into hand-written HIP kernels
for gfx950 (``infinicube_amd/csrc``).  Nothing in this package imports ``oracle/``.
"""

__version__ = "0.1.0"
