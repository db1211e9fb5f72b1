"""Drop-in namespace shim: lets InfiniCube's stage-2 script keep its import lines
[R infinicube/inference/guidance_buffer_generation.py:56-77,742] when this repo is on sys.path AHEAD of the
reference checkout.  Only the video-generation hot path (``infinicube.videogen``) and the two buffer
producers (``infinicube.utils.{buffer_utils,semantic_utils}``) are served from here; every other
``infinicube.*`` module (camera, data_process, inference, voxelgen, the rest of utils) still resolves to the
reference checkout further down sys.path, because the package search path is extended with it."""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)


def __getattr__(name):
    # the reference package re-exports one helper at the top level [R infinicube/__init__.py:3];
    # resolved lazily (it needs webdataset) from the reference's own utils.wds_utils
    if name == "get_sample":
        from .utils.wds_utils import get_sample
        return get_sample
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
