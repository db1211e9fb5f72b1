# Drop-in namespace shim: lets InfiniCube's stage-2 script keep its import line
# `from infinicube.videogen import WanVideoGenerator`
# [R infinicube/inference/guidance_buffer_generation.py:742] when this repo is on sys.path in place of
# (or ahead of) the reference package.  Only the video-generation hot path is provided.
